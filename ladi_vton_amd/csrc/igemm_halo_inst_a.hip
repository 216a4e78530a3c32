// Instantiation unit of the halo-resident 3x3 convolution kernel (igemm_halo_kernel.h): two of its sixteen forms.
#include "igemm_halo_kernel.h"
LADI_HALO_INSTANTIATE(f128x256_d, 2, 2, 2, 3, 4)
LADI_HALO_INSTANTIATE(f256x128, 4, 1, 1, 3, 4)
