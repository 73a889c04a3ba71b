out=gpurun_out/r3o; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_api_errors.py -x -q -m gpu 2>&1 | tail -3
MZ_GTREE_WAVE=0 timeout 300 python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-sustained > $out/c5_old.json 2>/dev/null
timeout 300 python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-sustained > $out/c5_new.json 2>/dev/null
python - <<PY
import json
for n in ("c5_old","c5_new"):
    d=json.load(open("$out/%s.json"%n)); print(n, "%.1f k sims/s"%(d["value"]/1e3), "ms %.2f"%d["ms_per_step"], "frac_step %.4f"%d["roofline"]["frac_step"])
PY
