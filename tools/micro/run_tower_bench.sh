#!/bin/bash
# fused residual tower vs separate conv launches, with cycle stamps (GPU box)
set -e
cd $(dirname $0)
OUT=${GRAFT_REPO_ROOT:-../..}/gpurun_out/conv_bench
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -Wno-unused-function -DMZC_STAMPS conv_bench.hip -o $OUT/cb_tower
for args in "$@"; do $OUT/cb_tower tower $args; done
