// bandwidth fits + KD builds with the speculative search (launches that cannot fill the chip)
#define NBP_TU 4
#include "nbp_kernels.h"
