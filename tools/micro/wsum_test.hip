// wave_sum_dpp (common.h) against the shuffle reduction, in the call patterns of the fused GroupNorm epilogue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "../../sola_amd/csrc/common.h"
void sola_set_error(const char*, ...) {}
SolaProfScope::SolaProfScope(int, hipStream_t, double, double) {}
SolaProfScope::~SolaProfScope() {}
__global__ void k(const float* in, float* out, int tl) {
    const int l = threadIdx.x;
    float lp[4];
    for (int p = 0; p < 4; ++p) lp[p] = in[l * 4 + p];
    float r[4];
    if (tl == 16) { const float m = wave_sum_dpp((lp[0] + lp[1]) + (lp[2] + lp[3])); r[0] = r[1] = r[2] = r[3] = m; }
    else if (tl == 8) { r[0] = r[1] = wave_sum_dpp(lp[0] + lp[1]); r[2] = r[3] = wave_sum_dpp(lp[2] + lp[3]); }
    else { for (int p = 0; p < 4; ++p) r[p] = wave_sum_dpp(lp[p]); }
    float q[4];
    if (tl == 16) { const float m = wave_sum_dpp(((lp[0] - r[0]) + (lp[1] - r[1])) + ((lp[2] - r[2]) + (lp[3] - r[3]))); q[0] = q[1] = q[2] = q[3] = m; }
    else if (tl == 8) { q[0] = q[1] = wave_sum_dpp((lp[0] - r[0]) + (lp[1] - r[1])); q[2] = q[3] = wave_sum_dpp((lp[2] - r[2]) + (lp[3] - r[3])); }
    else { for (int p = 0; p < 4; ++p) q[p] = wave_sum_dpp(lp[p] - r[p]); }
    for (int p = 0; p < 4; ++p) { out[(l * 4 + p) * 2] = r[p]; out[(l * 4 + p) * 2 + 1] = q[p]; }
}
int main() {
    float h[256], *d, *o, ho[512];
    for (int i = 0; i < 256; ++i) h[i] = sinf(i * 0.37f) + 0.01f * i;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    for (int tl : {16, 8, 4}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, tl);
        hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
        double worst = 0;
        for (int p = 0; p < 4; ++p) {
            double ref = 0;
            if (tl == 16) for (int i = 0; i < 256; ++i) ref += h[i];
            else if (tl == 8) for (int l = 0; l < 64; ++l) ref += h[l * 4 + (p & ~1)] + h[l * 4 + (p | 1)];
            else for (int l = 0; l < 64; ++l) ref += h[l * 4 + p];
            for (int l = 0; l < 64; ++l) worst = fmax(worst, fabs(ho[(l * 4 + p) * 2] - ref));
        }
        printf("tl %2d: worst |sum - ref| over lanes %.3e\n", tl, worst);
    }
    return 0;
}
