// product kernels, throughput geometries, one manifold per instance
#define NBP_TU 32
#include "nbp_kernels.h"
