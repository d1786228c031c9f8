// product kernels, throughput geometries, one manifold per instance, two helper lanes per sample
#define NBP_TU 32
#include "nbp_kernels.h"
