out=gpurun_out/r3a; mkdir -p $out
python bench.py --no-cpu-baseline --no-sustained --no-e2e --no-configs > $out/c2.json 2> $out/c2.err
MZ_FUSE_ENV=1 python bench.py --no-cpu-baseline --no-sustained --no-e2e --no-configs > $out/c2_fuse.json 2> $out/c2_fuse.err
python bench.py --workload c3 --no-cpu-baseline --no-sustained > $out/c3.json 2> $out/c3.err
python - <<PY
import json
for n in ("c2","c2_fuse","c3"):
    try:
        d = json.load(open("$out/%s.json" % n)); print(n, "%.1f M sims/s" % (d["value"]/1e6), "%.4f ms" % d["ms_per_step"], "frac %.4f" % d["roofline"]["frac"], "k_ms %.4f" % d["roofline"].get("avg_launch_ms", d["roofline"].get("avg_move_ms")))
    except Exception as e: print(n, "ERR", e)
PY
