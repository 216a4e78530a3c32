// Instantiation unit of the halo-resident 3x3 convolution kernel (igemm_halo_kernel.h): two of its sixteen forms.
#include "igemm_halo_kernel.h"
LADI_HALO_INSTANTIATE(f320x256, 5, 2, 1, 2, 4)
LADI_HALO_INSTANTIATE(f128x128_w2n, 2, 2, 1, 3, 2, 24)
