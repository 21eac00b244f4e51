// Translation unit of the 256-thread backward block kernels with 48 input channels (tu_bwd_block.inc).
#define MWW_TU_CIN 48
#include "tu_bwd_block.inc"
