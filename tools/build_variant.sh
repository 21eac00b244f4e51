#!/bin/bash
# kernel-tuning variant of the library: tools/build_variant.sh <name> [-DFLAG ...]  ->  microwakeword_amd/libmww_<name>.so
# (-DMWW_SLIM: default-topology kernels only - FULL=1 in the environment builds every shape, e.g. for --model inception;
# add -DMWW_PROFILE for the ablation / phase-clock switches)
R=$(cd $(dirname $0)/.. && pwd)
N=$1; shift
SLIM=-DMWW_SLIM; [ -n "$FULL" ] && SLIM=
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -shared -fPIC -pthread $SLIM "$@" -I $R/include $R/microwakeword_amd/csrc/mww_lib.hip $R/microwakeword_amd/csrc/sampler.cpp -o $R/microwakeword_amd/libmww_$N.so -ldl 2>&1 | grep -E "error|Error" ; ls -la $R/microwakeword_amd/libmww_$N.so
