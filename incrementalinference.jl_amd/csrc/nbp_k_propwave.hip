// proposal kernels, one wave per proposal (see nbp_kernels.h, "Translation units")
#define NBP_TU 512
#include "nbp_kernels.h"
