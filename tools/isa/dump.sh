#!/bin/bash
# usage: tools/isa/dump.sh [outdir]   -> <outdir>/hot.s + resource summary + memory-op skeleton per kernel
set -e
OUT=${1:-/tmp/isa}
mkdir -p $OUT
HERE=$(cd $(dirname $0) && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -I $HERE/../../include --cuda-device-only -S -o $OUT/hot.s $HERE/hot_kernels.hip 2>&1 | grep -v "hip-link" || true
python3 $HERE/summ.py $OUT/hot.s
