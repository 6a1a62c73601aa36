// Packed-f32 ops while the SIMD's other wave issues MFMAs: block = 8 waves (2 per SIMD), waves 0..3 run a long MFMA loop,
// waves 4..7 repeat a short VALU sequence and count iterations whose result differs from the plain-VALU evaluation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VAR>
__global__ __launch_bounds__(512) void k(const float* in, unsigned* bad, float* sink, int iters, int with_mfma) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 4) {
        if (!with_mfma) return;
        f32x16 acc = {};
        half8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.01f + i); b[i] = (_Float16)(0.5f - i * 0.1f); }
        for (int it = 0; it < iters * 6; ++it) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc, 0, 0, 0);
        }
        if (acc[0] == 12345.f) sink[threadIdx.x] = acc[1];
        return;
    }
    const int i = blockIdx.x * 256 + (threadIdx.x - 256);
    float x = in[i * 3], y = in[i * 3 + 1], z = in[i * 3 + 2];
    unsigned wrong = 0;
    for (int it = 0; it < iters; ++it) {
        float r0, r1;
        // v[22:23] = (x, y); v[20:21] = (y, x) with the high register overwritten by z inside the sequence; result v[24:25]
#define LOADS "v_mov_b32 v22, %2\n\tv_mov_b32 v23, %3\n\tv_mov_b32 v20, %3\n\tv_mov_b32 v21, %2\n\ts_nop 2\n\t"
#define OUTS "\n\ts_nop 4\n\tv_mov_b32 %0, v24\n\tv_mov_b32 %1, v25"
#define EXEC_OFF_ON "s_mov_b64 s[10:11], exec\n\ts_mov_b64 exec, 0\n\ts_nop 1\n\ts_or_b64 exec, exec, s[10:11]\n\t"
#define PKSUB "v_pk_add_f32 v[24:25], v[22:23], v[20:21] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]"
#define CLOBS "v20", "v21", "v22", "v23", "v24", "v25", "s10", "s11"
        if (VAR == 0) asm volatile(LOADS EXEC_OFF_ON "v_mov_b32 v21, %4\n\t" PKSUB OUTS : "=&v"(r0), "=&v"(r1) : "v"(x), "v"(y), "v"(z) : CLOBS);
        else if (VAR == 1) asm volatile(LOADS "v_mov_b32 v21, %4\n\t" PKSUB OUTS : "=&v"(r0), "=&v"(r1) : "v"(x), "v"(y), "v"(z) : CLOBS);
        else if (VAR == 2) asm volatile(LOADS EXEC_OFF_ON "v_mov_b32 v21, %4\n\tv_sub_f32 v24, v22, v21\n\tv_sub_f32 v25, v23, v21" OUTS : "=&v"(r0), "=&v"(r1) : "v"(x), "v"(y), "v"(z) : CLOBS);
        else asm volatile(LOADS "v_readfirstlane_b32 s10, %4\n\ts_nop 1\n\tv_pk_add_f32 v[24:25], v[22:23], s[10:11] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" OUTS
                          : "=&v"(r0), "=&v"(r1) : "v"(x), "v"(y), "v"(z) : CLOBS);
        const float r[2] = {r0, r1};
        float e0, e1;
        if (VAR == 3) { const float k0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, z))); e0 = x - k0; e1 = y - k0; }
        else { e0 = x - z; e1 = y - z; }
        wrong += (r[0] != e0) + ((r[1] != e1) << 16);
        x += 0.25f; z += 0.5f;
    }
    if (wrong) { atomicAdd(&bad[0], wrong & 0xffff); atomicAdd(&bad[1], wrong >> 16); }
}
int main() {
    const int blocks = 256 * 4, n = blocks * 256, iters = 2000;
    std::vector<float> h(n * 3);
    for (int i = 0; i < n * 3; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 256.f;
    float *d, *sink; unsigned* bad;
    hipMalloc(&d, n * 3 * 4); hipMalloc(&sink, 4096); hipMalloc(&bad, 8);
    hipMemcpy(d, h.data(), n * 3 * 4, hipMemcpyHostToDevice);
    const char* names[] = {"exec off/on ; v_mov ; pk_add op_sel", "v_mov ; pk_add op_sel", "exec off/on ; v_mov ; v_sub x2 (control)", "readfirstlane ; pk_add s[n:n+1]"};
    for (int with_mfma = 0; with_mfma < 2; ++with_mfma)
        for (int var = 0; var < 4; ++var) {
            hipMemset(bad, 0, 8);
            switch (var) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 0, 0, d, bad, sink, iters, with_mfma); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 0, 0, d, bad, sink, iters, with_mfma); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(512), 0, 0, d, bad, sink, iters, with_mfma); break;
                default: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(512), 0, 0, d, bad, sink, iters, with_mfma); break;
            }
            unsigned hb[2];
            hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost);
            printf("mfma partner %d | %-42s: wrong lo %u hi %u (of %ld lane-iterations)\n", with_mfma, names[var], hb[0], hb[1], (long)n * iters);
        }
    return 0;
}
