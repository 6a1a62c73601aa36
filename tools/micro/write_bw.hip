// Store throughput of one CU and of the chip for the two store shapes a GEMM epilogue can use: 16 lanes x 16 B = one
// 256-byte row segment per quarter wave (what gemm_glds.hip's LDS-transposed epilogue issues; row pitch 4 KiB) and
// a fully contiguous 1 KiB per wave instruction.  Each block of 512 threads writes `iters` tiles of 256 KiB, the
// tile size of the 256x256 kernel, to distinct addresses.  Prints GB/s per CU and aggregate for 1..256 active blocks.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void store_tiles(float* out, int iters, long long tile_stride_f, int pitch_f) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f4 v = {(float)lane, 1.f, 2.f, 3.f};
    for (int it = 0; it < iters; ++it) {
        float* base = out + ((long long)it * gridDim.x + blockIdx.x) * tile_stride_f;
        if (MODE == 0) {
            // wave = (wr, wc) of a 2 x 4 grid of 128 x 64 wave tiles; per pass 4 rows x 256 B
            const int wr = wave >> 2, wc = wave & 3;
#pragma unroll 4
            for (int p = 0; p < 32; ++p) {
                const int row = wr * 128 + p * 4 + (lane >> 4);
                *reinterpret_cast<f4*>(base + (long long)row * pitch_f + wc * 64 + (lane & 15) * 4) = v;
            }
        } else {
#pragma unroll 4
            for (int p = 0; p < 32; ++p) *reinterpret_cast<f4*>(base + ((p * 8 + wave) * 64 + lane) * 4) = v;
        }
    }
}
int main() {
    const int iters = 16;
    const size_t bytes = (size_t)iters * 256 * 256 * 1024;  // 1 GiB
    float* out; hipMalloc(&out, bytes); hipMemset(out, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode : {0, 1}) for (int blocks : {1, 8, 32, 64, 128, 256}) {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(store_tiles<0>, dim3(blocks), dim3(512), 0, 0, out, iters, 65536LL, 256);
            else hipLaunchKernelGGL(store_tiles<1>, dim3(blocks), dim3(512), 0, 0, out, iters, 65536LL, 256);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        const double b = (double)blocks * iters * 262144.0;
        printf("%s blocks %3d: %8.1f us  %7.1f GB/s per CU  %6.2f TB/s aggregate  (%.1f us per 256 KiB tile)\n",
               mode == 0 ? "row-segment stores" : "contiguous stores ", blocks, best * 1e3, b / blocks / (best * 1e-3) / 1e9,
               b / (best * 1e-3) / 1e12, best * 1e3 / iters);
    }
    return 0;
}
