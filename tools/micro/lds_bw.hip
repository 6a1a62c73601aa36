// LDS read bandwidth per CU for ds_read_b128 with the fragment access pattern of gemm_glds.hip (row pitch 128 B, 16-byte
// chunk XOR-swizzled by row & 7) and for plain linear reads; prints bytes / clock / CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(1024) void lds_read(float* out, int iters) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = (float)i;
    __syncthreads();
    int off;
    if (MODE == 0) off = lane * 16 + wave * 1024;                                        // linear 1 KiB per wave
    else if (MODE == 1) off = ((wave * 32 + (lane & 31)) & 255) * 128 + ((((lane >> 5) * 2) ^ (lane & 7)) << 4);  // fragment pattern, key = row & 7
    else off = ((wave * 32 + (lane & 31)) & 255) * 128 + ((((lane >> 5) * 2) ^ ((lane >> 1) & 7)) << 4);          // key = (row >> 1) & 7
    f4 acc = {0, 0, 0, 0};
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        f4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned addr = (unsigned)((off + u * 4096 + it * 128) & 65535);
            asm volatile("ds_read_b128 %0, %1" : "=v"(v[u]) : "v"(addr) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    long long t1 = clock64();
    if (acc[0] == 123.456f) out[0] = acc[1];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = (float)(t1 - t0);
}
int main() {
    float* out; hipMalloc(&out, 16);
    for (int waves : {4, 8, 16}) for (int mode : {0, 1, 2}) {
        const int iters = 2000;
        hipFuncSetAttribute((const void*)lds_read<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipFuncSetAttribute((const void*)lds_read<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipFuncSetAttribute((const void*)lds_read<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(lds_read<0>, dim3(256), dim3(waves * 64), 65536, 0, out, iters);
            else if (mode == 1) hipLaunchKernelGGL(lds_read<1>, dim3(256), dim3(waves * 64), 65536, 0, out, iters);
            else hipLaunchKernelGGL(lds_read<2>, dim3(256), dim3(waves * 64), 65536, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        float h[2]; hipMemcpy(h, out, 8, hipMemcpyDeviceToHost);
        const double bytes_per_cu = (double)waves * 64 * 16 * 8 * iters;
        printf("waves/CU %d mode %s: %.1f us, %.1f B/clk/CU by clock64 (%.0f clocks), %.2f TB/s aggregate\n", waves, mode == 0 ? "linear" : mode == 1 ? "fragment key=row&7" : "fragment key=(row>>1)&7",
               ms * 1e3, bytes_per_cu / h[1], h[1], bytes_per_cu * 256 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
