// ISA inspection unit: instantiates only the default-topology block kernels so that
// `tools/isa/dump.sh` compiles in seconds (the library itself instantiates ~160 kernels).
#include "../../microwakeword_amd/csrc/kernels_bwd.hip.h"
#include "../../microwakeword_amd/csrc/kernels_head.hip.h"
#include "../../microwakeword_amd/csrc/kernels_tail.hip.h"
namespace mww {
template __global__ void fwd_first_kernel<3, 32, 48, 5, 1, false>(FwdFirstArgs);
template __global__ void fwd_first_kernel<3, 32, 48, 5, 1, false, false, true>(FwdFirstArgs);
template __global__ void fwd_block_kernel<48, 48, 9, false>(FwdBlockArgs);
template __global__ void fwd_block_kernel<48, 48, 13, false>(FwdBlockArgs);
template __global__ void fwd_block_kernel<48, 48, 21, false>(FwdBlockArgs);
template __global__ void bwd_block_kernel<48, 48, 9, false, false>(BwdBlockArgs);
template __global__ void bwd_block_kernel<48, 48, 13, false, false>(BwdBlockArgs);
template __global__ void bwd_block_kernel<48, 48, 21, true, false>(BwdBlockArgs);
template __global__ void bwd_first_kernel<3, 32, 48, 5, 1, false>(BwdFirstArgs);
template __global__ void bwd_first_kernel<3, 32, 48, 5, 1, false, false, true>(BwdFirstArgs);
template __global__ void head_kernel<48, 8>(HeadArgs);
}
