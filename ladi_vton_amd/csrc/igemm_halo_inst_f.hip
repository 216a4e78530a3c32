// Instantiation unit of the halo-resident 3x3 convolution kernel (igemm_halo_kernel.h): two of its sixteen forms.
#include "igemm_halo_kernel.h"
LADI_HALO_INSTANTIATE(f320x192_w6, 5, 1, 1, 2, 6)
LADI_HALO_INSTANTIATE(g128x256, 2, 2, 1, 3, 4, 48, 0, 1)
