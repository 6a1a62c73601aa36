// Round 4: the fused conv + GroupNorm epilogue's second fault (garbage rows when the SLP vectoriser forms packed-f32 code) - what do
// v_pk_*_f32 instructions do with an SGPR-PAIR source?  The failing build's epilogue holds (ISA of the GNT kernels):
//     v_readfirstlane_b32 s12, v150
//     ...
//     v_pk_add_f32 v[188:189], v[152:153], s[12:13] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]     ; (a - k, b - k), k = s12, s13 = whatever
// Variants, every wave of a busy grid, wrong lanes counted against the host's model:
//   sg_lo      : s12 = k, s13 = POISON, pk_add with op_sel_hi:[1,0]  -> expected (x - k, y - k).  If the HIGH result shows y - POISON, the
//                hardware (or the assembler's encoding) does not honour op_sel_hi on an SGPR source: the high lane reads s13.
//   sg_pair    : s12 = k, s13 = k2, plain pk_add                      -> expected (x - k, y - k2): the documented pair semantics.
//   rfl_N      : v_readfirstlane_b32 s12 <- lane 0's z, N wait states, pk_add as in sg_lo -> expected (x - z0, y - z0): a missing
//                VALU-writes-SGPR -> packed-VALU-reads-SGPR interlock would show as wrong LOW results at small N.
//   rfl_sc     : the same through a plain (non-packed) v_sub_f32 pair: control.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>

#define CLOB "v10", "v11", "v12", "v14", "v15", "s12", "s13"
#define KERNEL(NAME, BODY)                                                                                 \
    __global__ void NAME(const float* in, float* out, float k, float k2) {                                 \
        const int i = blockIdx.x * blockDim.x + threadIdx.x;                                               \
        const float x = in[i * 3], y = in[i * 3 + 1], z = in[i * 3 + 2];                                   \
        float r0, r1;                                                                                      \
        asm volatile("v_mov_b32 v10, %2\n\tv_mov_b32 v11, %3\n\tv_mov_b32 v12, %4\n\tv_mov_b32 v14, 0\n\tv_mov_b32 v15, 0\n\t"     \
                     "s_mov_b32 s12, %5\n\ts_mov_b32 s13, %6\n\ts_nop 4\n\t" BODY                          \
                     "\n\ts_nop 7\n\tv_mov_b32 %0, v14\n\tv_mov_b32 %1, v15"                              \
                     : "=&v"(r0), "=&v"(r1) : "v"(x), "v"(y), "v"(z), "s"(k), "s"(k2) : CLOB);            \
        out[i * 2] = r0; out[i * 2 + 1] = r1;                                                              \
    }
#define PK_LO "v_pk_add_f32 v[14:15], v[10:11], s[12:13] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]"
KERNEL(sg_lo, PK_LO)
KERNEL(sg_pair, "v_pk_add_f32 v[14:15], v[10:11], s[12:13] neg_lo:[0,1] neg_hi:[0,1]")
KERNEL(sg_mul, "v_pk_mul_f32 v[14:15], s[12:13], v[10:11]")
KERNEL(rfl_0, "v_readfirstlane_b32 s12, v12\n\t" PK_LO)
KERNEL(rfl_1, "v_readfirstlane_b32 s12, v12\n\ts_nop 0\n\t" PK_LO)
KERNEL(rfl_2, "v_readfirstlane_b32 s12, v12\n\ts_nop 1\n\t" PK_LO)
KERNEL(rfl_4, "v_readfirstlane_b32 s12, v12\n\ts_nop 3\n\t" PK_LO)
KERNEL(rfl_v2, "v_readfirstlane_b32 s12, v12\n\tv_fma_f32 v14, v10, v10, v10\n\tv_fma_f32 v15, v11, v11, v11\n\t" PK_LO)  // the build's own spacing: two VALU between
KERNEL(rfl_sc, "v_readfirstlane_b32 s12, v12\n\tv_sub_f32 v14, v10, s12\n\tv_sub_f32 v15, v11, s12")

enum { LO_K, PAIR, MUL, RFL };
int main() {
    const int blocks = 4096, threads = 256, n = blocks * threads;
    std::vector<float> h(n * 3), o(n * 2);
    for (int i = 0; i < n * 3; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 256.f;
    float *d, *dout;
    hipMalloc(&d, n * 3 * 4); hipMalloc(&dout, n * 2 * 4);
    hipMemcpy(d, h.data(), n * 3 * 4, hipMemcpyHostToDevice);
    const float k = 3.25f, poison = 1000.5f;
    struct V { const char* name; void (*fn)(const float*, float*, float, float); int kind; } vs[] = {
        {"s12 = k, s13 = poison ; pk_add op_sel_hi:[1,0]       ", sg_lo, LO_K}, {"s[12:13] = (k, k2) ; pk_add (pair semantics)          ", sg_pair, PAIR},
        {"s[12:13] = (k, k2) ; pk_mul s, v                        ", sg_mul, MUL},
        {"readfirstlane s12 ; pk_add op_sel_hi:[1,0]            ", rfl_0, RFL}, {"readfirstlane s12 ; s_nop 0 ; pk_add                   ", rfl_1, RFL},
        {"readfirstlane s12 ; s_nop 1 ; pk_add                   ", rfl_2, RFL}, {"readfirstlane s12 ; s_nop 3 ; pk_add                   ", rfl_4, RFL},
        {"readfirstlane s12 ; 2 x v_fma ; pk_add                 ", rfl_v2, RFL}, {"readfirstlane s12 ; v_sub, v_sub (control)             ", rfl_sc, RFL}};
    for (auto& v : vs) {
        long bad[2] = {0, 0}, hi_is_poison = 0;
        for (int rep = 0; rep < 5; ++rep) {
            hipMemset(dout, 0, n * 2 * 4);
            hipLaunchKernelGGL(v.fn, dim3(blocks), dim3(threads), 0, 0, d, dout, k, poison);
            hipMemcpy(o.data(), dout, n * 2 * 4, hipMemcpyDeviceToHost);
            for (int w = 0; w < n / 64; ++w)
                for (int l = 0; l < 64; ++l) {
                    const float* p = &h[(w * 64 + l) * 3];
                    float e0, e1;
                    if (v.kind == LO_K) { e0 = p[0] - k; e1 = p[1] - k; }
                    else if (v.kind == PAIR) { e0 = p[0] - k; e1 = p[1] - poison; }
                    else if (v.kind == MUL) { e0 = p[0] * k; e1 = p[1] * poison; }
                    else { const float z0 = h[(w * 64) * 3 + 2]; e0 = p[0] - z0; e1 = p[1] - z0; }
                    const float g0 = o[(w * 64 + l) * 2], g1 = o[(w * 64 + l) * 2 + 1];
                    bad[0] += g0 != e0; bad[1] += g1 != e1;
                    hi_is_poison += g1 == p[1] - poison;
                }
        }
        printf("%s : wrong lanes  lo %ld  hi %ld  (of %ld)   hi == y - s13: %ld\n", v.name, bad[0], bad[1], 5L * n, hi_is_poison);
    }
    return 0;
}
