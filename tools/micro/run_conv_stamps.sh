#!/bin/bash
# per-phase cycle stamps of k_conv3x3 (diagnostic build) on the GPU box; extra -D flags via $MZC_FLAGS
set -e
cd $(dirname $0)
OUT=${GRAFT_REPO_ROOT:-../..}/gpurun_out/conv_bench
mkdir -p $OUT
for v in "" "-DMZC_NO_EPI" "-DMZC_NO_FETCH"; do
  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -Wno-unused-function -DMZC_STAMPS $v conv_bench.hip -o $OUT/cb_stamps
  echo "== variant [$v]"
  for args in "$@"; do $OUT/cb_stamps $args; done
done
