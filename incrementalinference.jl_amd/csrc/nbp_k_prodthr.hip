// product kernels, throughput geometries, any mix of manifolds (m4, t2)
#define NBP_TU 16
#include "nbp_kernels.h"
