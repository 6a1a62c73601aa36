// What HBM delivers for the access pattern of the attention core: a (group, head) unit reads 512-byte pieces (one 128-float
// head slice) of rows that are 4 KiB (or, for the inter-object attention, T' x 4 KiB) apart, and writes the same shape.
// Each block of 256 threads copies `rows` row pieces of `piece` bytes (16 B per lane) from in to out; units are distributed like
// the kernel's: consecutive blocks take consecutive head slices of the same rows.  Prints read+write TB/s for piece sizes
// from 512 B (one head) to 4 KiB (whole rows = all eight heads by one block) and row strides 1 / 4 rows.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
// unit u: group = u / slices, slice = u % slices; rows group*rows_per_unit*row_stride .. ; piece at slice*piece_f4
__global__ __launch_bounds__(256) void copy_units(const f4* __restrict__ in, f4* __restrict__ out, long long n_units, int slices, int rows_per_unit,
                                                  int row_stride, int piece_f4, int row_f4) {
    for (long long u = blockIdx.x; u < n_units; u += gridDim.x) {
        const long long grp = u / slices;
        const int sl = (int)(u - grp * slices);
        const long long row0 = grp / row_stride * ((long long)rows_per_unit * row_stride) + grp % row_stride;  // interleaved groups when row_stride > 1
        const int total = rows_per_unit * piece_f4;
#pragma unroll 4
        for (int i = threadIdx.x; i < total; i += 256) {
            const int r = i / piece_f4, c = i - r * piece_f4;
            const long long off = (row0 + (long long)r * row_stride) * row_f4 + sl * piece_f4 + c;
            out[off] = in[off];
        }
    }
}
// the attention core's mix: three input streams (q, k, v) of the same shape, one output stream
__global__ __launch_bounds__(256) void add3_units(const f4* __restrict__ a, const f4* __restrict__ b, const f4* __restrict__ c, f4* __restrict__ out,
                                                  long long n_units, int slices, int rows_per_unit, int row_stride, int piece_f4, int row_f4) {
    for (long long u = blockIdx.x; u < n_units; u += gridDim.x) {
        const long long grp = u / slices;
        const int sl = (int)(u - grp * slices);
        const long long row0 = grp / row_stride * ((long long)rows_per_unit * row_stride) + grp % row_stride;
        const int total = rows_per_unit * piece_f4;
#pragma unroll 4
        for (int i = threadIdx.x; i < total; i += 256) {
            const int r = i / piece_f4, cc = i - r * piece_f4;
            const long long off = (row0 + (long long)r * row_stride) * row_f4 + sl * piece_f4 + cc;
            out[off] = a[off] + b[off] + c[off];
        }
    }
}
// the same 512-byte pieces, but with the MFMA operand layout of the attention kernels: one load / store instruction of a wave
// covers SIXTEEN rows x 64 bytes (lane l: row l & 15, 16-byte chunk 4 i + (l >> 4) of the piece) instead of two rows x 512 bytes
__global__ __launch_bounds__(256) void copy_units_mfma(const f4* __restrict__ in, f4* __restrict__ out, long long n_units, int slices, int rows_per_unit,
                                                       int row_stride, int row_f4) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long long u = blockIdx.x; u < n_units; u += gridDim.x) {
        const long long grp = u / slices;
        const int sl = (int)(u - grp * slices);
        const long long row0 = grp / row_stride * ((long long)rows_per_unit * row_stride) + grp % row_stride;
        for (int t = wave; t < rows_per_unit / 16; t += 4) {  // 16-row tiles of the unit, one per wave
            const long long r = row0 + (long long)(t * 16 + (lane & 15)) * row_stride;
            f4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = in[r * row_f4 + sl * 32 + 4 * i + (lane >> 4)];
#pragma unroll
            for (int i = 0; i < 8; ++i) out[r * row_f4 + sl * 32 + 4 * i + (lane >> 4)] = v[i];
        }
    }
}
int main() {
    const long long rows = 65536LL * 4;  // 1 GiB of 4 KiB rows
    const int row_f4 = 256;
    f4 *in, *out;
    hipMalloc(&in, rows * row_f4 * 16); hipMalloc(&out, rows * row_f4 * 16);
    hipMemset(in, 1, rows * row_f4 * 16); hipMemset(out, 0, rows * row_f4 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int row_stride : {1, 4})
        for (int piece : {512, 1024, 2048, 4096})
            for (int blocks : {512, 1024, 2048}) {
                const int piece_f4 = piece / 16, slices = 4096 / piece, rpu = 64;
                const long long n_units = rows / rpu * slices;
                float best = 1e9;
                for (int rep = 0; rep < 3; ++rep) {
                    hipEventRecord(e0);
                    hipLaunchKernelGGL(copy_units, dim3(blocks), dim3(256), 0, 0, in, out, n_units, slices, rpu, row_stride, piece_f4, row_f4);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
                }
                const double bytes = 2.0 * rows * row_f4 * 16;
                printf("row stride %d rows, piece %4d B, %4d blocks: %7.1f us  %5.2f TB/s (read + write)\n", row_stride, piece, blocks, best * 1e3,
                       bytes / (best * 1e-3) / 1e12);
            }
    for (int row_stride : {1, 4})
        for (int blocks : {2048, 8192, 32768}) {
            const int slices = 8, rpu = 64;
            const long long n_units = rows / rpu * slices;
            float best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(copy_units_mfma, dim3(blocks), dim3(256), 0, 0, in, out, n_units, slices, rpu, row_stride, row_f4);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
            }
            const double bytes = 2.0 * rows * row_f4 * 16;
            printf("MFMA-layout copy (16 rows x 64 B per instruction), row stride %d, %5d blocks: %7.1f us  %5.2f TB/s (read + write)\n", row_stride, blocks,
                   best * 1e3, bytes / (best * 1e-3) / 1e12);
        }
    {   // 3 reads : 1 write, 512-byte head slices, row stride 4 (the inter-object attention at T' = 4)
        const long long rows3 = 65536LL;  // 256 MiB per tensor, 1 GiB moved: the headline launch
        f4 *b, *c;
        hipMalloc(&b, rows3 * row_f4 * 16); hipMalloc(&c, rows3 * row_f4 * 16);
        hipMemset(b, 1, rows3 * row_f4 * 16); hipMemset(c, 1, rows3 * row_f4 * 16);
        for (int row_stride : {1, 4})
            for (int blocks : {1024, 2048, 8192}) {
                const int piece_f4 = 32, slices = 8, rpu = 64;
                const long long n_units = rows3 / rpu * slices;
                float best = 1e9;
                for (int rep = 0; rep < 3; ++rep) {
                    hipEventRecord(e0);
                    hipLaunchKernelGGL(add3_units, dim3(blocks), dim3(256), 0, 0, in, b, c, out, n_units, slices, rpu, row_stride, piece_f4, row_f4);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
                }
                const double bytes = 4.0 * rows3 * row_f4 * 16;
                printf("3 in + 1 out, row stride %d, 512 B pieces, %4d blocks: %7.1f us  %5.2f TB/s\n", row_stride, blocks, best * 1e3, bytes / (best * 1e-3) / 1e12);
            }
    }
    return 0;
}
