out=gpurun_out/r3j; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/tests.txt 2>&1; tail -3 $out/tests.txt
timeout 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err; tail -2 $out/bench_default.err
python - <<PY
import json
d = json.load(open("$out/bench_default.json"))
print("c2 %.1f M sims/s %.4f ms frac %.4f frac_step %.4f" % (d["value"]/1e6, d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_step"]))
for k, v in (d.get("configs") or {}).items(): print(k, "%.1f k sims/s" % (v["value"]/1e3), "ms %.3f" % v["ms_per_step"], "frac %.4f step %.4f" % (v["roofline"]["frac"], v["roofline"]["frac_step"]))
print(d["e2e"]["fraction_of_planner_rate"], d["sustained"]["frac"], d["cpu_baseline"]["value"])
PY
bash tools/profile_all.sh c2 c3 c4 c5 > $out/profile_all.log 2>&1; tail -3 $out/profile_all.log
