// LDS-DMA (global_load_lds_dwordx4) cost per CU: (a) alone, from an L2-resident 64-KiB region, 8 waves x 8 pieces of
// 1 KiB per round as in gemm_glds.hip; (b) ds_read_b128 fragment-pattern reads alone (12 per wave per round);
// (c) both in the same loop.  Prints clocks per round; one round = one 16-wide half of a 256x256x32 k-tile for (b),
// one whole k-tile for (a).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
template <int MODE>
__global__ __launch_bounds__(512) void k(const char* src, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 131072 / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 0.f;
    __syncthreads();
    const char* p = src + (size_t)blockIdx.x % 4 * 65536 + wave * 8192 + lane * 16;
    const int off = ((wave * 32 + (lane & 31)) & 255) * 128 + ((((lane >> 5) * 2) ^ ((lane >> 1) & 7)) << 4);
    f4 acc = {0, 0, 0, 0};
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE & 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_amdgcn_global_load_lds((gptr_t)(p + i * 1024), (lptr_t)(lds + (it & 1) * 65536 + (wave * 8 + i) * 1024), 16, 0, 0);
        }
        if (MODE & 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f4 v[12];
#pragma unroll
                for (int u = 0; u < 12; ++u) {
                    const unsigned addr = (unsigned)((off + u * 4096 + h * 64) & 65535) + ((it + 1) & 1) * 65536;
                    asm volatile("ds_read_b128 %0, %1" : "=v"(v[u]) : "v"(addr) : "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int u = 0; u < 12; ++u) acc += v[u];
            }
        }
        if (MODE & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    long long t1 = clock64();
    if (acc[0] == 123.456f) out[0] = acc[1];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = (float)(t1 - t0);
}
int main() {
    float* out; hipMalloc(&out, 16);
    char* src; hipMalloc(&src, 4 * 65536); hipMemset(src, 0, 4 * 65536);
    const int iters = 2000;
    hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void*)k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    for (int mode : {1, 2, 3}) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 131072, 0, src, out, iters);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 131072, 0, src, out, iters);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 131072, 0, src, out, iters);
            hipDeviceSynchronize();
        }
        float h[2]; hipMemcpy(h, out, 8, hipMemcpyDeviceToHost);
        printf("%s: %.0f clocks per round (DMA 64 KiB%s)\n", mode == 1 ? "DMA only           " : mode == 2 ? "fragment reads only" : "DMA + reads        ",
               h[1] / iters, mode == 1 ? "" : ", reads 192 KiB");
    }
    return 0;
}
