// Translation unit of the 256-thread backward block kernels with 32 input channels (tu_bwd_block.inc).
#define MWW_TU_CIN 32
#include "tu_bwd_block.inc"
