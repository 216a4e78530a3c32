// Instantiation unit of the halo-resident 3x3 convolution kernel (igemm_halo_kernel.h): two of its sixteen forms.
#include "igemm_halo_kernel.h"
LADI_HALO_INSTANTIATE(f320x192_one, 5, 3, 1, 2, 2, 48, 1)
LADI_HALO_INSTANTIATE(g128x128_w2, 2, 2, 1, 2, 2, 48, 0, 1)
