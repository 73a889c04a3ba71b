out=gpurun_out/r3b; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/tests.txt 2>&1; tail -5 $out/tests.txt
timeout 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err; tail -3 $out/bench_default.err
python - <<PY
import json
d = json.load(open("$out/bench_default.json"))
print("c2 %.1f M sims/s %.4f ms frac %.4f frac_step %.4f" % (d["value"]/1e6, d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_step"]))
for k, v in (d.get("configs") or {}).items(): print(k, "%.1f k sims/s" % (v["value"]/1e3), "ms %.3f" % v["ms_per_step"], "frac %.4f step %.4f" % (v["roofline"]["frac"], v["roofline"]["frac_step"]))
print(d["e2e"]); print(d["sustained"]); print(d["cpu_baseline"])
PY
