# every tool once, briefly: nothing under tools/ may be broken by the round's refactors
run() { name=$1; shift; timeout 300 "$@" > gpurun_out/tool_$name.log 2>&1; echo "$name rc=$? $(grep -v amdgpu.ids gpurun_out/tool_$name.log | tail -1 | cut -c1-160)"; }
run learner_bench python tools/learner_bench.py --batches 128 --iters 20
run conv_learner_bench python tools/conv_learner_bench.py --board 9 --planes 32 --blocks 2 --iters 5
run infer_bench python tools/infer_bench.py
run tree_bench python tools/tree_bench.py
run phase_profile python tools/phase_profile.py cartpole
run idle_clock python tools/idle_clock_probe.py
run host_bound python tools/dev/host_bound.py 128
