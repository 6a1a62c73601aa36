// Where does global_load_lds_dwordx4 put the data of a HALF-active wave?  Lanes 0..31 and lanes 32..63 issue separately.
// hipcc --offload-arch=gfx950 -O3 tools/micro/dma_half.hip -o /tmp/dma_half && /tmp/dma_half
#include <hip/hip_runtime.h>
#include <cstdio>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void k(const float* in, float* out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) lds[i] = -1.f;
    __syncthreads();
    const float* src = in + lane * 4;
    if (lane < 32) {
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds), 16, 0, 0);
    } else {
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + 512 - 128), 16, 0, 0);  // expect lane 32 at float 512
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 1024; i += 64) out[i] = lds[i];
}
int main() {
    float *in, *out, h[1024];
    hipMalloc(&in, 4096); hipMalloc(&out, 4096);
    for (int i = 0; i < 1024; ++i) h[i] = (float)i;
    hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, in, out);
    hipMemcpy(h, out, 4096, hipMemcpyDeviceToHost);
    for (int i = 0; i < 1024; i += 4) if (h[i] >= 0.f) printf("lds[%d..] = %g %g %g %g\n", i, h[i], h[i+1], h[i+2], h[i+3]);
    return 0;
}
