out=gpurun_out/r3o; mkdir -p $out
MZ_GTREE_WAVE=0 timeout 300 python bench.py --workload c4 --steps 4 --warmup 1 --no-cpu-baseline --no-sustained > $out/c4_old.json 2>/dev/null
MZ_GTREE_WAVE=1 timeout 300 python bench.py --workload c4 --steps 4 --warmup 1 --no-cpu-baseline --no-sustained > $out/c4_new.json 2>/dev/null
python - <<PY
import json
for n in ("c4_old","c4_new"):
    d=json.load(open("$out/%s.json"%n)); print(n, "%.1f k sims/s"%(d["value"]/1e3), "ms %.2f"%d["ms_per_step"], "frac_step %.4f"%d["roofline"]["frac_step"])
PY
