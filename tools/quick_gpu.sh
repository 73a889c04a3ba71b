#!/bin/bash
# quick GPU iteration: parity subset, C2/C3 bench lines (no CPU leg), phase stamps.  Usage: tools/quick_gpu.sh <tag>
# (build muzero_amd/lib/*.so locally first: python -m muzero_amd.build && python -m muzero_amd.build --stamps)
tag=${1:-q}
out=gpurun_out/$tag
mkdir -p $out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $out/parity.txt 2>&1; tail -2 $out/parity.txt
python bench.py --no-cpu-baseline --no-sustained --no-e2e --no-configs > $out/bench_c2.json 2> $out/bench_c2.err
python bench.py --workload c3 --no-cpu-baseline --no-sustained --no-e2e --no-configs > $out/bench_c3.json 2> $out/bench_c3.err
python - <<PY
import json
for n in ("c2", "c3"):
    try:
        d = json.load(open("$out/bench_%s.json" % n)); print(n, "%.1f M sims/s" % (d["value"] / 1e6), "%.4f ms" % d["ms_per_step"], "frac %.4f" % d["roofline"]["frac"])
    except Exception as e: print(n, "ERR", e)
PY
python tools/phase_profile.py cartpole 2>&1 | grep -v amdgpu.ids > $out/phase_c2.txt; cat $out/phase_c2.txt
python tools/phase_profile.py tictactoe 2>&1 | grep -v amdgpu.ids > $out/phase_c3.txt; cat $out/phase_c3.txt
