// product kernels, latency geometries (x16, l8)
#define NBP_TU 8
#include "nbp_kernels.h"
