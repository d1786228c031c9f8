// bandwidth fits + KD builds at five waves per SIMD (workgroups of 4k + 1 waves: N = 257 .. 320)
#define NBP_TU 256
#include "nbp_kernels.h"
