// Round 5: what costs a v_mfma_f32_32x32x2_f32 stream its duty cycle?  W waves per SIMD, 4 accumulators per wave, 16 MFMAs per "step";
// per step optionally V independent VALU instructions (v_fma_f32 or v_pk_add_f32) and R ds_read_b128.  Prints the fraction of the f32 MFMA peak.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int V, int PK, int R, int NACC>
__global__ void loop(float* out, int iters, float scale) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 0.001f * (float)(i & 255);
    __syncthreads();
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    f32x4 a0 = {scale * lane, scale, 1.f, 2.f}, b0 = {scale, 0.5f * scale, 3.f, 1.f};
    float x[32];
    f32x2 px[16];
    for (int i = 0; i < 32; ++i) x[i] = scale * i;
    for (int i = 0; i < 16; ++i) px[i] = f32x2{scale * i, 1.f};
    f32x4 rd[4] = {a0, b0, a0, b0};
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds + lane * 16;
    for (int it = 0; it < iters; ++it) {
        if (R) {
#pragma unroll
            for (int u = 0; u < R; ++u) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rd[u]) : "v"(base), "n"(u * 1024) : "memory");
        }
#pragma unroll
        for (int s = 0; s < 16 / NACC; ++s)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s & 3], b0[j & 3], acc[j], 0, 0, 0);
        if (V && !PK) {
#pragma unroll
            for (int i = 0; i < V; ++i) x[i & 31] = __builtin_fmaf(x[i & 31], 1.0001f, scale);
        }
        if (V && PK) {
#pragma unroll
            for (int i = 0; i < V; ++i) px[i & 15] = px[i & 15] + f32x2{scale, scale};
        }
        // spread: one VALU behind every MFMA as far as they go
        if (V) {
#define SGB(g) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, (V + 15 - g) / 16, 0);
            SGB(0) SGB(1) SGB(2) SGB(3) SGB(4) SGB(5) SGB(6) SGB(7) SGB(8) SGB(9) SGB(10) SGB(11) SGB(12) SGB(13) SGB(14) SGB(15)
#undef SGB
        }
        __builtin_amdgcn_sched_barrier(0);
        if (R) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rd[0]), "+v"(rd[1]), "+v"(rd[2]), "+v"(rd[3])::"memory");
    }
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) s += acc[j][0];
    for (int i = 0; i < 32; ++i) s += x[i];
    for (int i = 0; i < 16; ++i) s += px[i][0];
    for (int u = 0; u < 4; ++u) s += rd[u][0];
    if (s == 123.456f) out[0] = s;
}
template <int V, int PK, int R, int NACC>
static void run(float* out, int waves, const char* what) {
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((loop<V, PK, R, NACC>), dim3(256), dim3(64 * 4 * waves), 65536, 0, out, iters, 0.01f);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * 4 * waves * iters * 16 * 4096.0;
    printf("%-44s waves/SIMD %d: %8.1f us  %.3f of 157.3 TFLOP/s\n", what, waves, ms * 1e3, flops / (ms * 1e-3) / 157.3e12);
}
int main() {
    float* out; (void)hipMalloc(&out, 16);
    hipFuncSetAttribute((const void*)loop<0, 0, 0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int w : {1, 2, 4}) {
        run<0, 0, 0, 4>(out, w, "pure MFMA, 4 accumulators");
        run<0, 0, 0, 2>(out, w, "pure MFMA, 2 accumulators");
        run<0, 0, 0, 1>(out, w, "pure MFMA, 1 accumulator (dependent chain)");
        run<8, 0, 0, 4>(out, w, "+ 8 v_fma per 16 MFMAs");
        run<32, 0, 0, 4>(out, w, "+ 32 v_fma per 16 MFMAs");
        run<64, 0, 0, 4>(out, w, "+ 64 v_fma per 16 MFMAs");
        run<16, 1, 0, 4>(out, w, "+ 16 v_pk_add_f32 per 16 MFMAs");
        run<32, 1, 0, 4>(out, w, "+ 32 v_pk_add_f32 per 16 MFMAs");
        run<0, 0, 4, 4>(out, w, "+ 4 ds_read_b128 per 16 MFMAs");
        run<8, 0, 4, 4>(out, w, "+ 4 ds_read_b128 + 8 v_fma per 16 MFMAs");
    }
    return 0;
}
