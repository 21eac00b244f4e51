// ISA inspection unit for the conv/BN graph kernels (Inception shapes)
#include "../../microwakeword_amd/csrc/kernels_graph.hip.h"
namespace mww {
template __global__ void gconv_kernel<24, 0>(GConvArgs);
template __global__ void gconv_kernel<10, 0>(GConvArgs);
template __global__ void gconv_bwd_kernel<16, 48>(GWgradArgs, GConvArgs, int);
}
