// Instantiation unit of the halo-resident 3x3 convolution kernel (igemm_halo_kernel.h): two of its sixteen forms.
#include "igemm_halo_kernel.h"
LADI_HALO_INSTANTIATE(g256x256, 4, 2, 1, 3, 4, 48, 0, 1)
LADI_HALO_INSTANTIATE(g320x256, 5, 2, 1, 2, 4, 48, 0, 1)
