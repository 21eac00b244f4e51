// ISA inspection unit for the conv/BN graph kernels (Inception shapes): run-time instantiations next to the static-shape ones
#include "../../microwakeword_amd/csrc/kernels_graph.hip.h"
namespace mww {
template __global__ void gconv_kernel<24, 0>(GConvArgs);
template __global__ void gconv_kernel<10, 0>(GConvArgs);
template __global__ void gconv_bwd_kernel<16, 48>(GWgradArgs, GConvArgs, int, int);
template __global__ void gconv_bwd2_kernel<10, 10>(GBwd2Args, int, int);
template __global__ void gdw_kernel<0>(GDwArgs);
template __global__ void gdw_kernel<1>(GDwArgs);
typedef GShape<5, 1, 40, 40> GShStem;
typedef GShape<5, 1, 10, 10> GSh10k5;
typedef GShape<1, 3, 10, 10, 10, 10, 10, 10> GShCat10;   // the shapes of the default Inception as the library launches them
typedef GShape<1, 3, 16, 16, 16, 16, 16, 16> GShCat16;   // (profiles/round5_rocprofv3_kernel_stats_inception.csv)
template __global__ void gconv_kernel<24, 0, GShStem>(GConvArgs);
template __global__ void gconv_wgrad_kernel<24, GShStem>(GWgradArgs);
template __global__ void gconv_xg_kernel<24, GShStem>(GConvArgs, XGather);         // the stem gathering its input from the stores
template __global__ void gconv_wgrad_xg_kernel<24, GShStem>(GWgradArgs, XGather);
template __global__ void gconv_kernel<10, 0, GSh10k5>(GConvArgs);
template __global__ void gconv_bwd_kernel<16, 48, GShCat16>(GWgradArgs, GConvArgs, int, int);
template __global__ void gconv_bwd_kernel<10, 30, GShCat10>(GWgradArgs, GConvArgs, int, int);
template __global__ void gconv_bwd2_kernel<10, 10, GSh10k5>(GBwd2Args, int, int);
}
