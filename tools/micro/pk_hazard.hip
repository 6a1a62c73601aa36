// Packed-f32 ops behind (a) a permlane swap, (b) a plain v_mov of their op_sel'ed operand, (c) an exec restore - do they need
// wait states the compiler / an asm author must add on gfx950?  Same harness as swap_hazard.hip: wrong lanes out of a busy grid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define LOAD "v_mov_b32 v10, %2\n\tv_mov_b32 v11, %3\n\tv_mov_b32 v12, %4\n\tv_mov_b32 v13, %3\n\tv_mov_b32 v14, 0\n\tv_mov_b32 v15, 0\n\ts_nop 4\n\t"
#define CLOB "v10", "v11", "v12", "v13", "v14", "v15", "s10", "s11", "s12", "s13"
#define KERNEL(NAME, BODY)                                                                                 \
    __global__ void NAME(const float* in, float* out) {                                                    \
        const int i = blockIdx.x * blockDim.x + threadIdx.x;                                               \
        const float x = in[i * 3], y = in[i * 3 + 1], z = in[i * 3 + 2];                                   \
        float r0, r1;                                                                                      \
        asm volatile(LOAD BODY "\n\ts_nop 7\n\tv_mov_b32 %0, v14\n\tv_mov_b32 %1, v15"                     \
                     : "=&v"(r0), "=&v"(r1) : "v"(x), "v"(y), "v"(z) : CLOB);                              \
        out[i * 2] = r0; out[i * 2 + 1] = r1;                                                              \
    }
// v10 = x, v11 = y, v12 = z, v13 = y
// (a) swap32(v10, v12) then pk_add (v10 + v12, v11 + v13) with n wait states
KERNEL(swap_pk_0, "v_permlane32_swap_b32 v10, v12\n\tv_pk_add_f32 v[14:15], v[10:11], v[12:13]")
KERNEL(swap_pk_1, "v_permlane32_swap_b32 v10, v12\n\ts_nop 0\n\tv_pk_add_f32 v[14:15], v[10:11], v[12:13]")
KERNEL(swap_pk_2, "v_permlane32_swap_b32 v10, v12\n\ts_nop 1\n\tv_pk_add_f32 v[14:15], v[10:11], v[12:13]")
KERNEL(swap_add_1, "v_permlane32_swap_b32 v10, v12\n\ts_nop 0\n\tv_add_f32 v14, v10, v12\n\tv_add_f32 v15, v11, v13")
// (b) v_mov v13 <- z, then pk: (v10 - v13, v11 - v13) through op_sel (low result reads the HIGH register of the pair)
#define PK_OPSEL "v_pk_add_f32 v[14:15], v[10:11], v[12:13] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]"
KERNEL(mov_pk_0, "v_mov_b32 v13, v12\n\t" PK_OPSEL)
KERNEL(mov_pk_1, "v_mov_b32 v13, v12\n\ts_nop 0\n\t" PK_OPSEL)
// (c) exec cleared and restored right in front
#define EXEC_OFF_ON "s_mov_b64 s[10:11], exec\n\ts_mov_b64 exec, 0\n\ts_nop 3\n\ts_or_b64 exec, exec, s[10:11]\n\t"
KERNEL(exec_pk_0, EXEC_OFF_ON PK_OPSEL)
KERNEL(exec_mov_pk_0, EXEC_OFF_ON "v_mov_b32 v13, v12\n\t" PK_OPSEL)
KERNEL(exec_mov_pk_1, EXEC_OFF_ON "v_mov_b32 v13, v12\n\ts_nop 0\n\t" PK_OPSEL)
KERNEL(exec_mov_add_0, EXEC_OFF_ON "v_mov_b32 v13, v12\n\tv_sub_f32 v14, v10, v13\n\tv_sub_f32 v15, v11, v13")
// (c') exec restored by the compiler's idiom: saveexec of an empty mask, branch over nothing, or-restore
#define SAVEEXEC "s_mov_b64 s[12:13], 0\n\ts_and_saveexec_b64 s[10:11], s[12:13]\n\ts_cbranch_execz 0\n\ts_or_b64 exec, exec, s[10:11]\n\t"
KERNEL(save_mov_pk_0, SAVEEXEC "v_mov_b32 v13, v12\n\t" PK_OPSEL)

enum { A_SWAP, B_SUB_Z, B_SUB_Y };
int main() {
    const int blocks = 4096, threads = 256, n = blocks * threads;
    std::vector<float> h(n * 3), o(n * 2);
    for (int i = 0; i < n * 3; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 256.f;
    float *d, *dout;
    hipMalloc(&d, n * 3 * 4); hipMalloc(&dout, n * 2 * 4);
    hipMemcpy(d, h.data(), n * 3 * 4, hipMemcpyHostToDevice);
    struct V { const char* name; void (*k)(const float*, float*); int kind; } vs[] = {
        {"swap32 ; pk_add                      ", swap_pk_0, A_SWAP}, {"swap32 ; s_nop 0 ; pk_add            ", swap_pk_1, A_SWAP},
        {"swap32 ; s_nop 1 ; pk_add            ", swap_pk_2, A_SWAP}, {"swap32 ; s_nop 0 ; add, add          ", swap_add_1, A_SWAP},
        {"v_mov hi ; pk_add op_sel             ", mov_pk_0, B_SUB_Z}, {"v_mov hi ; s_nop 0 ; pk_add op_sel   ", mov_pk_1, B_SUB_Z},
        {"exec 0 -> on ; pk_add op_sel         ", exec_pk_0, B_SUB_Y}, {"exec 0 -> on ; v_mov ; pk_add op_sel ", exec_mov_pk_0, B_SUB_Z},
        {"exec 0 -> on ; v_mov ; nop ; pk_add  ", exec_mov_pk_1, B_SUB_Z}, {"exec 0 -> on ; v_mov ; sub, sub      ", exec_mov_add_0, B_SUB_Z},
        {"saveexec/execz/or ; v_mov ; pk_add   ", save_mov_pk_0, B_SUB_Z}};
    for (auto& v : vs) {
        long bad[2] = {0, 0};
        for (int rep = 0; rep < 5; ++rep) {
            hipMemset(dout, 0, n * 2 * 4);
            hipLaunchKernelGGL(v.k, dim3(blocks), dim3(threads), 0, 0, d, dout);
            hipMemcpy(o.data(), dout, n * 2 * 4, hipMemcpyDeviceToHost);
            for (int w = 0; w < n / 64; ++w)
                for (int l = 0; l < 64; ++l) {
                    const float* p = &h[(w * 64 + l) * 3];
                    float e0, e1;
                    if (v.kind == A_SWAP) {  // a = x, b = z: upper half of a <-> lower half of b; a + b per lane
                        const float* q = &h[(w * 64 + (l ^ 32)) * 3];
                        const float a2 = l < 32 ? p[0] : q[2], b2 = l < 32 ? q[0] : p[2];
                        e0 = a2 + b2; e1 = p[1] + p[1];
                    } else {
                        const float m = v.kind == B_SUB_Z ? p[2] : p[1];
                        e0 = p[0] - m; e1 = p[1] - m;
                    }
                    bad[0] += o[(w * 64 + l) * 2] != e0; bad[1] += o[(w * 64 + l) * 2 + 1] != e1;
                }
        }
        printf("%s : wrong lanes  lo %ld  hi %ld  (of %ld)\n", v.name, bad[0], bad[1], 5L * n);
    }
    return 0;
}
