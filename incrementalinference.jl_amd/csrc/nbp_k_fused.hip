// the fused variable-update kernels (nbp_fused.h)
// hipcc-flags: -mllvm -disable-machine-licm
// (machine LICM hoists the materialisation of the double constants of log / sincos / the polynomial kernels out of the
//  proposal loop and keeps ~40 VGPRs of them live across every phase: 144 VGPRs + spills with it, 118-121 and none without)
#define NBP_TU 64
#include "nbp_fused.h"
