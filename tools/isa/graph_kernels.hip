// ISA inspection unit for the conv/BN graph kernels (Inception shapes)
#include "../../microwakeword_amd/csrc/kernels_graph.hip.h"
namespace mww {
template __global__ void gconv_kernel<24, 0>(GConvArgs);
template __global__ void gconv_kernel<10, 0>(GConvArgs);
template __global__ void gconv_bwd_kernel<16, 48>(GWgradArgs, GConvArgs, int, int);
template __global__ void gconv_bwd2_kernel<10, 10>(GBwd2Args, int, int);
template __global__ void gdw_kernel<0>(GDwArgs);
template __global__ void gdw_kernel<1>(GDwArgs);
}
