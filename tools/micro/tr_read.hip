// ds_read_b64_tr_b16 semantics probe (gfx950): which values does lane j of a 16-lane group receive when lane i supplies the address of
// piece (row i >> 2, 8-byte quad i & 3)?  Expected (attn_f16.hip's P V operand): lane j <- rows 0..3 of column 8 (j >> 2) + (j & 3) when
// quad q sits at column 8 q.   hipcc --offload-arch=gfx950 -O3 tools/micro/tr_read.hip -o /tmp/tr_read && /tmp/tr_read
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short short4v __attribute__((ext_vector_type(4)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
__global__ void k(_Float16* out) {
    __shared__ __attribute__((aligned(16))) _Float16 s[64 * 136];
    for (int i = threadIdx.x; i < 64 * 136; i += 64) s[i] = (_Float16)(float)((i / 136) * 32 + ((i % 136) & 31));
    __syncthreads();
    const int lane = threadIdx.x, i16 = lane & 15, g4 = lane >> 4;
    const _Float16* p = &s[(4 * g4 + (i16 >> 2)) * 136 + 8 * (i16 & 3)];
    short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)p);
    half4v h = __builtin_bit_cast(half4v, v);
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = h[j];
}
int main() {
    _Float16* d; hipMalloc(&d, 256 * 2);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    _Float16 h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) for (int r = 0; r < 4; ++r) {
        const int j = lane & 15, g4 = lane >> 4;
        const float want = (4 * g4 + r) * 32 + 8 * (j >> 2) + (j & 3), got = (float)h[lane * 4 + r];
        if (want != got) { if (bad < 8) printf("lane %d r %d: got %g (row %d col %d) want %g\n", lane, r, got, (int)got / 32, (int)got % 32, want); ++bad; }
    }
    printf("tr_read: %d mismatches\n", bad);
    return bad != 0;
}
