// Instantiation unit of the halo-resident 3x3 convolution kernel (igemm_halo_kernel.h): the twelve-wave folded-upsample form (UPS = 1, round 6).
#include "igemm_halo_kernel.h"
LADI_HALO_INSTANTIATE(u320x192_w6, 5, 1, 1, 2, 6, 48, 0, 0, 1)
