// ISA inspection unit: the wide (512-thread) block backward kernels of the default topology (tools/isa/dump.sh wide).
#include "../../microwakeword_amd/csrc/kernels_bwdw.hip.h"
namespace mww {
template __global__ void bwd_blockw_kernel<48, 48, 9, false, 512>(BwdBlockArgs);
template __global__ void bwd_blockw_kernel<48, 48, 13, false, 512>(BwdBlockArgs);
template __global__ void bwd_blockw_kernel<48, 48, 21, true, 512>(BwdBlockArgs);
}
namespace mww {
template __global__ void bwd_firstw_kernel<3, 32, 48, 5, 1, 512, true>(BwdFirstArgs);
}
