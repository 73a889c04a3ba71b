#!/bin/bash
# geometry variants of the product kernel (MZ_CONV_G / MZ_CONV_NCT overrides) on the GPU box
set -e
cd $(dirname $0)
OUT=${GRAFT_REPO_ROOT:-../..}/gpurun_out/conv_bench
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -Wno-unused-function conv_bench.hip -o $OUT/cb
for env in "" "MZ_CONV_NCT=2" "MZ_CONV_G=2" "MZ_CONV_G=2 MZ_CONV_NCT=2" "MZ_CONV_G=3" "MZ_CONV_G=1" "MZ_CONV_G=8"; do
  echo "== $env"
  for args in "$@"; do env $env $OUT/cb $args; done
done
