// What does one global_store_dwordx4 cost the issuing wave?  Every active CU runs one block of eight waves that stores a 256 KB tile the way
// the persistent GEMM's epilogue does (32 stores of 1 KB per wave, back to back); s_memtime (shader clock cycles) after the last ISSUE and again
// after vmcnt(0).
//   hipcc --offload-arch=gfx950 -O2 -o store_issue store_issue.hip && ./store_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// pattern 0: the persistent GEMM's epilogue - block b owns rows [256 b, 256 b + 256) x 256 columns of a [65536][1024] f32 matrix, eight waves of
// 128 x 64, 32 stores per wave of 4 rows x 256 B.  pattern 1: the same bytes as whole 1-KB rows (a wave stores 32 rows of the tile's 256 columns).
__global__ __launch_bounds__(512) void k(float* out, long long* t_issue, long long* t_done, int pattern, int n_cu_active, int ct) {
    if ((int)blockIdx.x >= n_cu_active) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 2, wc = wave & 3, rsub = lane >> 4, c4 = lane & 15;
    const long long ldc = 1024;
    float* tile = out + (long long)blockIdx.x * 256 * ldc + ct * 256;
    f32x4 v = {1.f, 2.f, 3.f, (float)lane};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < 32; ++i) {
        float* p;
        if (pattern == 0) p = tile + (long long)(wr * 128 + i * 4 + rsub) * ldc + wc * 64 + c4 * 4;
        else p = tile + (long long)(wave * 32 + i) * ldc + lane * 4;
        asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    }
    const long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t2 = __builtin_readcyclecounter();
    if (lane == 0) { t_issue[blockIdx.x * 8 + wave] = t1 - t0; t_done[blockIdx.x * 8 + wave] = t2 - t0; }
}
int main() {
    const int n_cu = 256;
    const size_t bytes = (size_t)65536 * 1024 * 4;
    float* out; long long *ti, *td;
    hipMalloc(&out, bytes); hipMalloc(&ti, n_cu * 8 * 8); hipMalloc(&td, n_cu * 8 * 8);
    hipMemset(out, 0, bytes);
    std::vector<long long> hi(n_cu * 8), hd(n_cu * 8);
    const char* names[] = {"4 rows x 256 B (epilogue)", "whole 1-KB rows"};
    for (int active : {1, 32, 256})
        for (int pattern = 0; pattern < 2; ++pattern) {
            for (int rep = 0; rep < 4; ++rep) hipLaunchKernelGGL(k, dim3(n_cu), dim3(512), 0, 0, out, ti, td, pattern, active, rep & 3);
            hipMemcpy(hi.data(), ti, n_cu * 8 * 8, hipMemcpyDeviceToHost); hipMemcpy(hd.data(), td, n_cu * 8 * 8, hipMemcpyDeviceToHost);
            double si = 0, sd = 0;
            for (int i = 0; i < active * 8; ++i) { si += hi[i]; sd += hd[i]; }
            si /= active * 8; sd /= active * 8;  // shader clock cycles per wave for its 32 stores
            printf("%3d CUs storing a 256 KB tile each, %-26s: %6.1f cycles per store instruction to issue, tile acknowledged after %7.0f cycles (%5.1f B/clk per CU)\n",
                   active, names[pattern], si / 32, sd, 262144.0 / sd);
        }
    return 0;
}
