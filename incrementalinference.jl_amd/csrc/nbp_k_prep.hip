// bandwidth fits + KD builds, sequential golden-section search
#define NBP_TU 2
#include "nbp_kernels.h"
