#include <hip/hip_runtime.h>
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef __fp16 f16x4v __attribute__((__vector_size__(4 * sizeof(__fp16))));
__global__ void k(_Float16* out) {
    __shared__ _Float16 s[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) s[i] = (_Float16)i;
    __syncthreads();
    auto p = (__attribute__((address_space(3))) f16x4v*)(s + threadIdx.x * 4);
    f16x4v v = __builtin_amdgcn_ds_read_tr16_b64_v4f16(p);
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
    _Float16* d; hipMalloc((void**)&d, 512);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    _Float16 h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %4d", (int)(float)h[l*4+j]); printf("\n"); }
}
