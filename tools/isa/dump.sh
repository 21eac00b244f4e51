#!/bin/bash
# usage: tools/isa/dump.sh [outdir] [unit]   -> <outdir>/<unit>.s + resource summary + memory-op skeleton per kernel
#   unit = hot (default: tools/isa/hot_kernels.hip, the 256-thread kernels of the default topology) | wide (wide_kernels.hip) | graph
# then: python tools/isa/hist.py <outdir>/<unit>.s [name filter]   (instruction-class histogram of every kernel's tile loop)
set -e
OUT=${1:-/tmp/isa}
UNIT=${2:-hot}
mkdir -p $OUT
HERE=$(cd $(dirname $0) && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -I $HERE/../../include --cuda-device-only -S -o $OUT/$UNIT.s $HERE/${UNIT}_kernels.hip 2>&1 | grep -v "hip-link" || true
python3 $HERE/summ.py $OUT/$UNIT.s
