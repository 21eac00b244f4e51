#!/bin/bash
# profiling variant of the library (phase ablation + phase clocks compiled in): microwakeword_amd/libmww_hip_prof.so
R=$(cd $(dirname $0)/.. && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -shared -fPIC -pthread -DMWW_PROFILE -I $R/include $R/microwakeword_amd/csrc/mww_lib.hip $R/microwakeword_amd/csrc/sampler.cpp -o $R/microwakeword_amd/libmww_hip_prof.so
