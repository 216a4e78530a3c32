// Instantiation unit of the halo-resident 3x3 convolution kernel (igemm_halo_kernel.h): two of its sixteen forms.
#include "igemm_halo_kernel.h"
LADI_HALO_INSTANTIATE(f128x128_d, 2, 1, 2, 4, 4)
LADI_HALO_INSTANTIATE(f128x192_w2, 2, 3, 1, 2, 2)
