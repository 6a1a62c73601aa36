// The compiler's interleaving of two wave_sum_dpp chains (fused GroupNorm epilogue), replayed as one asm block.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define DPP(d, s, ctl) "v_add_f32_dpp " d ", " s ", " s " " ctl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define CLOB "v145", "v152", "v154", "v160", "v161", "v174", "v175"
// in: %2 = chain-1 lane value p, %3 = chain-2 lane value q.  out: %0 = sum(q) (v160), %1 = sum(p) (v161)
#define SEQ(PAD_A, PAD_B)                                                                                               \
    "v_mov_b32 v145, %2\n\tv_mov_b32 v152, %3\n\ts_nop 4\n\t"                                                            \
    DPP("v145", "v145", "quad_perm:[1,0,3,2]") "s_nop 1\n\t" DPP("v145", "v145", "quad_perm:[2,3,0,1]") "s_nop 1\n\t"   \
    DPP("v145", "v145", "row_half_mirror") "s_nop 1\n\t" DPP("v145", "v145", "row_mirror")                              \
    "v_mov_b32 v154, v145\n\ts_nop 1\n\tv_permlane16_swap_b32 v154, v145\n\ts_nop 0\n\t"                                 \
    "v_add_f32 v161, v154, v145\n\t" PAD_A                                                                              \
    DPP("v145", "v152", "quad_perm:[1,0,3,2]")                                                                           \
    "v_mov_b32 v175, v161\n\ts_nop 1\n\tv_permlane32_swap_b32 v175, v161\n\ts_nop 0\n\t" PAD_B                           \
    DPP("v145", "v145", "quad_perm:[2,3,0,1]") "s_nop 1\n\t" DPP("v145", "v145", "row_half_mirror") "s_nop 1\n\t"        \
    DPP("v145", "v145", "row_mirror")                                                                                   \
    "v_mov_b32 v152, v145\n\ts_nop 1\n\tv_permlane16_swap_b32 v152, v145\n\ts_nop 0\n\t"                                 \
    "v_add_f32 v160, v152, v145\n\tv_mov_b32 v174, v160\n\ts_nop 1\n\tv_permlane32_swap_b32 v174, v160\n\ts_nop 0\n\t"   \
    "v_pk_add_f32 v[160:161], v[174:175], v[160:161]\n\ts_nop 4\n\tv_mov_b32 %0, v160\n\tv_mov_b32 %1, v161"
#define KERNEL(NAME, A, B)                                                                       \
    __global__ void NAME(const float* in, float* out) {                                          \
        const int i = blockIdx.x * blockDim.x + threadIdx.x;                                     \
        const float p = in[i * 2], q = in[i * 2 + 1];                                            \
        float r0, r1;                                                                            \
        asm volatile(SEQ(A, B) : "=&v"(r0), "=&v"(r1) : "v"(p), "v"(q) : CLOB);                  \
        out[i * 2] = r0; out[i * 2 + 1] = r1;                                                    \
    }
KERNEL(as_compiled, "", "")
KERNEL(pad_a, "s_nop 3\n\t", "")
KERNEL(pad_b, "", "s_nop 3\n\t")
int main() {
    const int blocks = 4096, threads = 256, n = blocks * threads;
    std::vector<float> h(n * 2), o(n * 2);
    for (int i = 0; i < n * 2; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xff);  // small integers: sums are exact
    float *d, *dout;
    hipMalloc(&d, n * 2 * 4); hipMalloc(&dout, n * 2 * 4);
    hipMemcpy(d, h.data(), n * 2 * 4, hipMemcpyHostToDevice);
    struct V { const char* name; void (*k)(const float*, float*); } vs[] = {{"as compiled", as_compiled}, {"pad after add ", pad_a}, {"pad after swap32", pad_b}};
    for (auto& v : vs) {
        long bad[2] = {0, 0};
        for (int rep = 0; rep < 5; ++rep) {
            hipLaunchKernelGGL(v.k, dim3(blocks), dim3(threads), 0, 0, d, dout);
            hipMemcpy(o.data(), dout, n * 2 * 4, hipMemcpyDeviceToHost);
            for (int w = 0; w < n / 64; ++w) {
                float sp = 0, sq = 0;
                for (int l = 0; l < 64; ++l) { sp += h[(w * 64 + l) * 2]; sq += h[(w * 64 + l) * 2 + 1]; }
                for (int l = 0; l < 64; ++l) { bad[0] += o[(w * 64 + l) * 2] != sq; bad[1] += o[(w * 64 + l) * 2 + 1] != sp; }
            }
        }
        printf("%-18s: wrong lanes  sum(q) %ld  sum(p) %ld  (of %ld)\n", v.name, bad[0], bad[1], 5L * n);
    }
    return 0;
}
