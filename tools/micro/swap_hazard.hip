// Which instruction pairs around v_permlane{16,32}_swap_b32 need wait states on gfx950?  Each variant runs a fixed asm
// sequence in every wave of a busy grid and counts lanes whose results differ from the host's model.
//   hipcc --offload-arch=gfx950 -O2 -o swap_hazard swap_hazard.hip && ./swap_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define SEQ_HEAD "v_mov_b32 %0, %3\n\tv_mov_b32 %1, %4\n\ts_nop 4\n\t"
// outputs: %0 = a, %1 = b, %2 = c ; inputs %3 = x, %4 = y, %5 = z
#define KERNEL(NAME, BODY)                                                                                      \
    __global__ void NAME(const float* in, float* out) {                                                         \
        const int i = blockIdx.x * blockDim.x + threadIdx.x;                                                    \
        const float x = in[i * 3], y = in[i * 3 + 1], z = in[i * 3 + 2];                                        \
        float a, b, c;                                                                                          \
        asm volatile(SEQ_HEAD BODY "\n\ts_nop 7" : "=&v"(a), "=&v"(b), "=&v"(c) : "v"(x), "v"(y), "v"(z));     \
        out[i * 3] = a; out[i * 3 + 1] = b; out[i * 3 + 2] = c;                                                 \
    }

#define DPP_C "v_add_f32_dpp %2, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1"
// swap, then a DPP op on unrelated registers, 0..2 wait states between
KERNEL(s32_dpp_0, "v_permlane32_swap_b32 %0, %1\n\t" DPP_C)
KERNEL(s32_dpp_1, "v_permlane32_swap_b32 %0, %1\n\ts_nop 0\n\t" DPP_C)
KERNEL(s32_dpp_2, "v_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\t" DPP_C)
KERNEL(s16_dpp_0, "v_permlane16_swap_b32 %0, %1\n\t" DPP_C)
KERNEL(s16_dpp_1, "v_permlane16_swap_b32 %0, %1\n\ts_nop 0\n\t" DPP_C)
// DPP op, then the swap on unrelated registers
KERNEL(dpp_s32_0, DPP_C "\n\tv_permlane32_swap_b32 %0, %1")
KERNEL(dpp_s16_0, DPP_C "\n\tv_permlane16_swap_b32 %0, %1")
// swap, plain VALU read of its results (c = a + b), then a DPP op that overwrites b  (the compiler's interleaving of two reductions)
KERNEL(s16_use_dppw, "v_permlane16_swap_b32 %0, %1\n\ts_nop 0\n\tv_add_f32 %2, %0, %1\n\tv_add_f32_dpp %1, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1")
KERNEL(s32_use_dppw, "v_permlane32_swap_b32 %0, %1\n\ts_nop 0\n\tv_add_f32 %2, %0, %1\n\tv_add_f32_dpp %1, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1")
// swap results read by a plain VALU op with 0 / 1 wait states
KERNEL(s32_use_0, "v_permlane32_swap_b32 %0, %1\n\tv_add_f32 %2, %0, %1")
KERNEL(s16_use_0, "v_permlane16_swap_b32 %0, %1\n\tv_add_f32 %2, %0, %1")
// two swaps back to back on different registers (c starts as a copy of z; swapped with a afterwards)
KERNEL(s16_s32, "v_mov_b32 %2, %5\n\ts_nop 4\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %1")

struct Model { int kind; };  // how to compute the expectation
enum { SWAP32_DPP, SWAP16_DPP, S16_USE_DPPW, S32_USE_DPPW, S32_USE, S16_USE, S16_S32 };

static void swap16(float* a, float* b) {  // odd rows of a <-> even rows of b (rows of 16 lanes)
    for (int l = 0; l < 16; ++l) { float t = a[16 + l]; a[16 + l] = b[l]; b[l] = t; t = a[48 + l]; a[48 + l] = b[32 + l]; b[32 + l] = t; }
}
static void swap32(float* a, float* b) {  // upper half of a <-> lower half of b
    for (int l = 0; l < 32; ++l) { float t = a[32 + l]; a[32 + l] = b[l]; b[l] = t; }
}

int main() {
    const int blocks = 4096, threads = 256, n = blocks * threads;
    std::vector<float> h(n * 3), o(n * 3);
    for (int i = 0; i < n * 3; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 256.f;
    float *d, *dout;
    hipMalloc(&d, n * 3 * 4); hipMalloc(&dout, n * 3 * 4);
    hipMemcpy(d, h.data(), n * 3 * 4, hipMemcpyHostToDevice);
    struct V { const char* name; void (*k)(const float*, float*); int kind; } vs[] = {
        {"swap32 ; dpp(other)          ", s32_dpp_0, SWAP32_DPP}, {"swap32 ; s_nop 0 ; dpp(other)", s32_dpp_1, SWAP32_DPP},
        {"swap32 ; s_nop 1 ; dpp(other)", s32_dpp_2, SWAP32_DPP}, {"swap16 ; dpp(other)          ", s16_dpp_0, SWAP16_DPP},
        {"swap16 ; s_nop 0 ; dpp(other)", s16_dpp_1, SWAP16_DPP}, {"dpp(other) ; swap32          ", dpp_s32_0, SWAP32_DPP},
        {"dpp(other) ; swap16          ", dpp_s16_0, SWAP16_DPP}, {"swap16 ; nop ; add ; dpp -> b", s16_use_dppw, S16_USE_DPPW},
        {"swap32 ; nop ; add ; dpp -> b", s32_use_dppw, S32_USE_DPPW}, {"swap32 ; add(a, b)           ", s32_use_0, S32_USE},
        {"swap16 ; add(a, b)           ", s16_use_0, S16_USE}, {"swap16(a,b) ; swap32(c,b)    ", s16_s32, S16_S32}};
    for (auto& v : vs) {
        long bad[3] = {0, 0, 0};
        for (int rep = 0; rep < 5; ++rep) {
            hipMemset(dout, 0, n * 3 * 4);
            hipLaunchKernelGGL(v.k, dim3(blocks), dim3(threads), 0, 0, d, dout);
            hipMemcpy(o.data(), dout, n * 3 * 4, hipMemcpyDeviceToHost);
            for (int w = 0; w < n / 64; ++w) {
                float a[64], b[64], c[64], z[64], dppz[64];
                for (int l = 0; l < 64; ++l) { a[l] = h[(w * 64 + l) * 3]; b[l] = h[(w * 64 + l) * 3 + 1]; z[l] = h[(w * 64 + l) * 3 + 2]; }
                for (int l = 0; l < 64; ++l) dppz[l] = z[l] + z[l ^ 1];
                switch (v.kind) {
                    case SWAP32_DPP: swap32(a, b); for (int l = 0; l < 64; ++l) c[l] = dppz[l]; break;
                    case SWAP16_DPP: swap16(a, b); for (int l = 0; l < 64; ++l) c[l] = dppz[l]; break;
                    case S16_USE_DPPW: swap16(a, b); for (int l = 0; l < 64; ++l) { c[l] = a[l] + b[l]; b[l] = dppz[l]; } break;
                    case S32_USE_DPPW: swap32(a, b); for (int l = 0; l < 64; ++l) { c[l] = a[l] + b[l]; b[l] = dppz[l]; } break;
                    case S32_USE: swap32(a, b); for (int l = 0; l < 64; ++l) c[l] = a[l] + b[l]; break;
                    case S16_USE: swap16(a, b); for (int l = 0; l < 64; ++l) c[l] = a[l] + b[l]; break;
                    case S16_S32: for (int l = 0; l < 64; ++l) c[l] = z[l]; swap16(a, b); swap32(c, b); break;
                }
                for (int l = 0; l < 64; ++l) {
                    const float* g = &o[(w * 64 + l) * 3];
                    bad[0] += g[0] != a[l]; bad[1] += g[1] != b[l]; bad[2] += g[2] != c[l];
                }
            }
        }
        printf("%s : wrong lanes a %ld  b %ld  c %ld  (of %ld)\n", v.name, bad[0], bad[1], bad[2], 5L * n);
    }
    return 0;
}
