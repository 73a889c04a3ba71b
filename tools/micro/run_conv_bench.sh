#!/bin/bash
# builds the diagnostic variants of tools/micro/conv_bench.hip and runs them (GPU box)
set -e
cd $(dirname $0)
OUT=${GRAFT_REPO_ROOT:-../..}/gpurun_out/conv_bench
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -Wno-unused-function"
for v in BASE NO_FETCH NO_XS NO_STORE NO_EPI "NO_FETCH -DMZC_NO_STORE" "NO_FETCH -DMZC_NO_STORE -DMZC_NO_EPI" "NO_FETCH -DMZC_NO_STORE -DMZC_NO_EPI -DMZC_NO_XS"; do
  n=$(echo $v | tr -d ' ' | tr -d '-')
  hipcc $FLAGS -DMZC_$v conv_bench.hip -o $OUT/cb_$n &
done
wait
for b in $OUT/cb_*; do
  echo "== $b"
  for args in "$@"; do $b $args; done
done
