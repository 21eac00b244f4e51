// Translation unit of the 256-thread backward block kernels with 64 input channels (tu_bwd_block.inc).
#define MWW_TU_CIN 64
#include "tu_bwd_block.inc"
