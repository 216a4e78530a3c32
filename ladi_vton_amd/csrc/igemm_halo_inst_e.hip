// Instantiation unit of the halo-resident 3x3 convolution kernel (igemm_halo_kernel.h): two of its sixteen forms.
#include "igemm_halo_kernel.h"
LADI_HALO_INSTANTIATE(f128x192_w2n, 2, 3, 1, 3, 2, 24)
LADI_HALO_INSTANTIATE(f128x256_w2n, 2, 4, 1, 2, 2, 24)
