// The descending xor butterfly of a 64-lane sum / max on the vector ALU (v_permlane32_swap, v_permlane16_swap, DPP row rotations and quad
// permutations) against the __shfl_xor form (ds_bpermute through the LDS crossbar): every lane, every step, bit for bit.
//   hipcc -O3 --offload-arch=gfx950 -I sola_amd/csrc tools/micro/wave_reduce_check.hip -o tools/micro/wave_reduce_check && tools/micro/wave_reduce_check
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"

__global__ void check_kernel(const float* in, float* out_ref, float* out_new, float* out_refmax, float* out_newmax, float* steps_ref, float* steps_new) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float x = in[i];
    float v = x;
    for (int o = 32, s = 0; o > 0; o >>= 1, ++s) {
        v += __shfl_xor(v, o, 64);
        steps_ref[(size_t)i * 6 + s] = v;
    }
    out_ref[i] = v;
    float m = x;
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    out_refmax[i] = m;
    out_new[i] = wave_sum(x);
    out_newmax[i] = wave_max(x);
    float w = x;
    w = sum_xor32(w); steps_new[(size_t)i * 6 + 0] = w;
    w = sum_xor16(w); steps_new[(size_t)i * 6 + 1] = w;
    w = sum_xor8(w); steps_new[(size_t)i * 6 + 2] = w;
    w = sum_xor4(w); steps_new[(size_t)i * 6 + 3] = w;
    w = sum_xor2(w); steps_new[(size_t)i * 6 + 4] = w;
    w = sum_xor1(w); steps_new[(size_t)i * 6 + 5] = w;
}

int main() {
    const int n = 64 * 4096;
    std::vector<float> h(n);
    srand(1);
    for (int i = 0; i < n; ++i) h[i] = ((rand() % 20001) - 10000) / 37.0f * ((i % 7) ? 1.f : 1e-3f);
    float *in, *a, *b, *c, *d, *sr, *sn;
    hipMalloc(&in, n * 4); hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&c, n * 4); hipMalloc(&d, n * 4);
    hipMalloc(&sr, (size_t)n * 24); hipMalloc(&sn, (size_t)n * 24);
    hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(check_kernel, dim3(n / 256), dim3(256), 0, 0, in, a, b, c, d, sr, sn);
    std::vector<float> ha(n), hb(n), hc(n), hd(n), hsr((size_t)n * 6), hsn((size_t)n * 6);
    hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hc.data(), c, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hd.data(), d, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hsr.data(), sr, (size_t)n * 24, hipMemcpyDeviceToHost); hipMemcpy(hsn.data(), sn, (size_t)n * 24, hipMemcpyDeviceToHost);
    long bad_sum = 0, bad_max = 0, bad_step[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        bad_sum += memcmp(&ha[i], &hb[i], 4) != 0;
        bad_max += memcmp(&hc[i], &hd[i], 4) != 0;
        for (int s = 0; s < 6; ++s) bad_step[s] += memcmp(&hsr[(size_t)i * 6 + s], &hsn[(size_t)i * 6 + s], 4) != 0;
    }
    printf("lanes checked %d: wave_sum mismatches %ld, wave_max mismatches %ld, per step (xor 32,16,8,4,2,1): %ld %ld %ld %ld %ld %ld\n", n, bad_sum, bad_max,
           bad_step[0], bad_step[1], bad_step[2], bad_step[3], bad_step[4], bad_step[5]);
    return (bad_sum || bad_max) ? 1 : 0;
}
