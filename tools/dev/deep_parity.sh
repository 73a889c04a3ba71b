#!/bin/bash
# The deep parity checks of a round's FINAL build, run on the GPU box through gpurun; log tails are copied to profiles/<round>/deep_parity/.
#   1. MZ_SLOW_TESTS=1: C5 (Gomoku 15x15, A = 226) through all 200 simulations against the oracle  2. MZ_SOAK_SEEDS=16  3. MZ_FUZZ_CASES=300
OUT=$GRAFT_REPO_ROOT/gpurun_out/deep
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
FP=$(python3 -c "from muzero_amd import build as b; print('planner sources', b.source_fingerprint(), '| learner sources', b.learner_fingerprint())")
#   4. the learner fuzz generators at 300 (MLP) / 300 (board conv, random + kink-free weights) / 100 (Atari) cases
# JOBS="slow soak fuzz learnfuzz" (default: all four)
for job in "slow:MZ_SLOW_TESTS=1:tests/test_gpu_conv.py -k full_size" "soak:MZ_SOAK_SEEDS=${SOAK:-16}:tests/test_gpu_soak.py" "fuzz:MZ_FUZZ_CASES=${FUZZ:-300}:tests/test_gpu_fuzz.py" \
           "learnfuzz:MZ_FUZZ_LEARN_CASES=300 MZ_FUZZ_CONV_LEARN_CASES=300 MZ_FUZZ_ATARI_LEARN_CASES=100:tests/test_gpu_fuzz.py -k learner"; do
  name=${job%%:*}; rest=${job#*:}; envs=${rest%%:*}; args=${rest#*:}
  case " ${JOBS:-slow soak fuzz learnfuzz} " in *" $name "*) ;; *) continue ;; esac
  { echo "== $name: $envs python -m pytest $args -q -m gpu"; echo "== build: $FP"; date -u; } > $OUT/$name.log
  ( export $envs; timeout ${JOB_TIMEOUT:-1500} python3 -m pytest $args -q -m gpu -x --durations=5 2>&1 | grep -v "amdgpu.ids" | tail -40 ) >> $OUT/$name.log
  { date -u; } >> $OUT/$name.log
  tail -4 $OUT/$name.log
done
