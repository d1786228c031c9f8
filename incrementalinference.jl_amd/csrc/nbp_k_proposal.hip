// proposal / deconv kernels and the small copy / reseed / resample kernels (see nbp_kernels.h, "Translation units")
#define NBP_TU 1
#include "nbp_kernels.h"
