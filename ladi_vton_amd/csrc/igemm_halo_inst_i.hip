// Instantiation unit of the halo-resident 3x3 convolution kernel (igemm_halo_kernel.h): the folded-upsample forms (UPS = 1, round 6).
#include "igemm_halo_kernel.h"
LADI_HALO_INSTANTIATE(u128x192, 2, 3, 1, 2, 2, 48, 0, 0, 1)
LADI_HALO_INSTANTIATE(u128x128, 2, 2, 1, 2, 2, 48, 0, 0, 1)
