#!/bin/bash
# Round checkpoint on the GPU box (run through gpurun): the whole GPU test suite, the default bench line, rocprofv3 profiles of all
# four workloads (tools/profile_all.sh -> gpurun_out/profiles/, then tools/stamp_profiles.py here) and the phase stamps of C2 / C3.
out=gpurun_out/checkpoint; mkdir -p $out
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
python tools/phase_profile.py cartpole 2>&1 | grep -v amdgpu.ids > $out/phase_c2.txt; tail -14 $out/phase_c2.txt
python tools/phase_profile.py tictactoe 2>&1 | grep -v amdgpu.ids > $out/phase_c3.txt; tail -14 $out/phase_c3.txt
