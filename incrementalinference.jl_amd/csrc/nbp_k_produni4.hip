// product kernels, throughput geometries, one manifold per instance, four helper lanes per sample
#define NBP_TU 128
#include "nbp_kernels.h"
