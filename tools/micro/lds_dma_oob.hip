// Round 5: what does an out-of-range lane of buffer_load_dwordx4 ... lds leave in LDS - zeros, or the old contents?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void k(const float* src, float* out, int nrec) {
    __shared__ __attribute__((aligned(16))) float lds[256];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = -7.f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, nrec, 0x00020000);
    const unsigned vo = (threadIdx.x & 1) ? 0x80000000u : threadIdx.x * 16;  // odd lanes out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)lds, 16, vo, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}
int main() {
    float *src, *out, h[256];
    hipMalloc(&src, 4096); hipMalloc(&out, 1024);
    for (int i = 0; i < 256; ++i) h[i] = (float)(i + 1);
    hipMemcpy(src, h, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, out, 1024);
    hipMemcpy(h, out, 1024, hipMemcpyDeviceToHost);
    printf("lane 0 (in range): %g %g %g %g   lane 1 (out of range): %g %g %g %g   lane 2: %g lane 3: %g\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[12]);
    return 0;
}
