// Weight gradients dW[n][k] = sum_m dY[m][n] X[m][k] on the split-f16 MFMA path (training, precision 1).
// The reduction runs over the ROWS of both operands, so each is first written transposed in the split-f16 format
// ([N][Mp] and [K][Mp], Mp = M rounded up and zero-filled) - one streaming pass that replaces the f32 kernel's strided
// staging - and the product is then the ordinary NT GEMM of gemm_glds.hip with the reduction dim = Mp.  dW has only
// (N/256) x (K/256) output tiles (16 for the 1024 x 1024 projections), so the reduction is cut into `ksplit` ranges,
// one persistent-kernel work item each, and gemm.hip's splitk_reduce_kernel folds the partial sums in a fixed order.
// dY is a gradient (entries ~1e-4..1e-9): it is scaled by a data-dependent power of two found on the device
// (cast.hip's amax pass; all problems of a launch share the scale), undone by the GEMM's out_scale_dev.
// Second route (round 3, the default for 16-bit operands - GemmTnSplitDesc::pure - when N and K are multiples of 256): NO transposed
// copies.  The operands are cast ROW-MAJOR (dY's cast is the dX GEMM's operand anyway, X's comes from the training forward's kept
// casts or one plain cast; convs: one cast of the conv input) and gemm_glds.hip's gemm_tn_tr_kernel transposes them in its LDS reads.
#include <algorithm>

#include "kernels.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float auto_scale_t(unsigned amax_bits) {  // as cast.hip: max|x| -> [2^13, 2^14)
    if (amax_bits == 0u) return 1.f;
    const int e = (int)(amax_bits >> 23) - 127;
    const int k = min(max(13 - e, -100), 100);
    return __uint_as_float((unsigned)(k + 127) << 23);
}

// in [rows][cols] f32 (pitch ld_in) -> out [cols][ld_out] split-f16, out[c][8b..8b+7] = scale * in[8b..8b+7][c]; rows beyond
// `rows` up to ld_out are written as zeros.  Tile = 128 rows x 64 columns through LDS.
// With rowmap != null (ragged batches) row m reads source row rowmap[m].x + tap, zeros where bit `tap` of rowmap[m].y is clear.
// With conv_T_out > 0 the input is the implicit im2col of a channels-last conv for ONE tap: row m = (r, to) reads source row
// r * conv_T_in + to * conv_stride + conv_toff (tap - pad), zeros outside [0, conv_T_in).
// PURE: the output is plain _Float16 ([cols][ld_out halfs]) for the one-MFMA-per-product GEMM (GemmDesc::arith 2); bf != 0: the
// 16-bit values are bfloat16 bit patterns instead.
__device__ __forceinline__ _Float16 cvt16(float v, int bf) {
    if (bf) return __builtin_bit_cast(_Float16, (__bf16)v);
    return (_Float16)v;
}
// RMP: the format of the optional row-major copy (PURE = as the transposed output; false with PURE = true: the weight-gradient
// product takes plain f16 operands while the dX GEMM of the same gradient matrix keeps split-f16 ones - sola_tune "train_dw_f16")
template <bool PURE, bool RMP = PURE>
__global__ __launch_bounds__(256) void cast_sp16_t_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols,
                                                          int ld_in, long long ld_out, float* __restrict__ scal, int conv_T_in,
                                                          int conv_T_out, int conv_stride, int conv_toff, float* __restrict__ out_rm,
                                                          long long ld_rm, const int2* __restrict__ rowmap, int tap, int bf) {
    __shared__ float tile[128][65];
    const int t = threadIdx.x;
    const int r0 = blockIdx.x * 128, c0 = blockIdx.y * 64;
    float scale = 1.f;
    if (scal) {
        scale = auto_scale_t(reinterpret_cast<const unsigned*>(scal)[0]);
        if (blockIdx.x == 0 && blockIdx.y == 0 && t == 0) scal[1] = 1.f / scale;
    }
    {   // 16-byte loads when the pitch allows; LDS pitch 65: bank = (row + col) % 64, conflict-free for both phases
        const int c4 = (t & 15) * 4, rq = t >> 4;
        const bool vec = (ld_in & 3) == 0 && c0 + 64 <= cols;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = i * 16 + rq;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            long long src = r0 + r;
            bool ok = r0 + r < rows;
            if (rowmap && ok) {  // ragged batches: tap `tap` of output row m lives at source row rowmap[m].x + tap (GemmDesc::rowmap)
                const int2 rm = rowmap[r0 + r];
                ok = (rm.y >> tap) & 1;
                src = (long long)rm.x + tap;
            } else if (conv_T_out > 0 && ok) {
                const int m = r0 + r, rr = m / conv_T_out, tt = (m - rr * conv_T_out) * conv_stride + conv_toff;
                ok = (unsigned)tt < (unsigned)conv_T_in;
                src = (long long)rr * conv_T_in + tt;
            }
            if (ok) {
                const float* p = in + src * ld_in + c0 + c4;
                if (vec) {
                    const float4 q = *reinterpret_cast<const float4*>(p);
                    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (c0 + c4 + e < cols) v[e] = p[e];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[r][c4 + e] = v[e] * scale;
            if (out_rm && r0 + r < rows) {  // the row-major cast of the same values (vec path guaranteed by the launcher)
                const long long col = c0 + c4;
                if (RMP) {
                    typedef _Float16 half4t __attribute__((ext_vector_type(4)));
                    half4t h;
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[e] = cvt16(v[e] * scale, bf);
                    *reinterpret_cast<half4t*>(reinterpret_cast<_Float16*>(out_rm) + (long long)(r0 + r) * ld_rm + col) = h;
                } else {  // 8-value blocks [hi8 | lo8]: this thread owns half a block
                    typedef _Float16 half4t __attribute__((ext_vector_type(4)));
                    half4t hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        _Float16 h1, l1;
                        split_f16(v[e] * scale, h1, l1);
                        hi[e] = h1; lo[e] = l1;
                    }
                    char* dst = reinterpret_cast<char*>(out_rm + (long long)(r0 + r) * ld_rm + (col & ~7ll)) + 8 * ((col >> 2) & 1);
                    *reinterpret_cast<half4t*>(dst) = hi;
                    *reinterpret_cast<half4t*>(dst + 16) = lo;
                }
            }
        }
    }
    __syncthreads();
    // a lane owns (8-row block b, column cc) of an 8 x 8 patch: banks 8 b + cc + e are all different; 8 lanes write the 256
    // contiguous bytes of one output row
    const int lane = t & 63, wave = t >> 6;
    const int b8 = lane & 7, cc = lane >> 3;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = wave * 16 + cc + 8 * (j & 1), b = b8 + 8 * (j >> 1);
        if (c0 + c >= cols) continue;
        half8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = tile[8 * b + e][c];
            _Float16 h1, l1;
            if (PURE) { h1 = cvt16(v, bf); l1 = (_Float16)0.f; }
            else split_f16(v, h1, l1);
            hi[e] = h1; lo[e] = l1;
        }
        if (PURE) {
            *reinterpret_cast<half8*>(reinterpret_cast<_Float16*>(out) + (long long)(c0 + c) * ld_out + r0 + 8 * b) = hi;
        } else {
            half8* o = reinterpret_cast<half8*>(out + (long long)(c0 + c) * ld_out + r0 + 8 * b);
            o[0] = hi;
            o[1] = lo;
        }
    }
}

// Transposed-conv scatter as a gather: z [R*T_out][k*cin] holds, for every output step, its contribution to each of the k
// input steps it touches (z = dY W, one GEMM); dx[(r, ti)][ci] = sum over the taps kk with (ti + pad - kk) = to * stride,
// 0 <= to < T_out, of z[(r, to)][kk*cin + ci].  One thread per float4 of dx; fixed tap order.
// ZB (round 6, bf16 steps): z is a bfloat16 matrix of the same shape (the GEMM that produced it wrote bfloat16 rows); the taps are summed in f32
__device__ __forceinline__ float4 c2i_load(const float* z, long long off, bool zb) {
    if (zb) {
        const uint2 w = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(z) + off);
        return make_float4(__builtin_bit_cast(float, w.x << 16), __builtin_bit_cast(float, w.x & 0xffff0000u),
                           __builtin_bit_cast(float, w.y << 16), __builtin_bit_cast(float, w.y & 0xffff0000u));
    }
    return *reinterpret_cast<const float4*>(z + off);
}
template <bool ZB>
__global__ __launch_bounds__(256) void col2im_kernel(const float* __restrict__ z, float* __restrict__ dx, long long n4, int T_in, int T_out,
                                                     int cin4, int k, int stride, int pad) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int c4 = (int)(i % cin4);
    const long long rt = i / cin4;
    const int ti = (int)(rt % T_in);
    const long long r = rt / T_in;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int kk = 0; kk < k; ++kk) {
        const int u = ti + pad - kk;
        if (u < 0 || u % stride) continue;
        const int to = u / stride;
        if (to >= T_out) continue;
        const float4 v = c2i_load(z, ((r * T_out + to) * k + kk) * (long long)(cin4 * 4) + c4 * 4, ZB);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4*>(dx + i * 4) = acc;
}

// ragged batches: input row i of the conv has imap[i] = (first output row of its sequence, T_out, step ti, -)
template <bool ZB>
__global__ __launch_bounds__(256) void col2im_ragged_kernel(const float* __restrict__ z, float* __restrict__ dx, long long n4,
                                                            const int4* __restrict__ imap, int cin4, int k, int stride, int pad) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int c4 = (int)(i % cin4);
    const int4 im = imap[i / cin4];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int kk = 0; kk < k; ++kk) {
        const int u = im.z + pad - kk;
        if (u < 0 || u % stride) continue;
        const int to = u / stride;
        if (to >= im.y) continue;
        const float4 v = c2i_load(z, (((long long)im.x + to) * k + kk) * (long long)(cin4 * 4) + c4 * 4, ZB);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4*>(dx + i * 4) = acc;
}

int cast_t(int pure, const float* in, int ld_in, float* out, long long ld_out, int rows, int cols, float* scal, hipStream_t s, int conv_T_in = 0,
           int conv_T_out = 0, int conv_stride = 1, int conv_toff = 0, float* out_rm = nullptr, long long ld_rm = 0,
           const int2* rowmap = nullptr, int tap = 0, int rm_split = 0) {
    const int bf = pure == 2 ? 1 : 0;
    const bool rmp = pure && !rm_split;
    SolaProfScope prof(SOLA_PROF_MISC, s, 0, ((pure ? 6.0 : 8.0) + (out_rm ? (rmp ? 2.0 : 4.0) : 0.0)) * rows * cols);
    const dim3 grid((unsigned)(ld_out / 128), (unsigned)((cols + 63) / 64));
    if (pure && rm_split) hipLaunchKernelGGL((cast_sp16_t_kernel<true, false>), grid, dim3(256), 0, s, in, out, rows, cols, ld_in, ld_out, scal, conv_T_in, conv_T_out, conv_stride, conv_toff, out_rm, ld_rm, rowmap, tap, bf);
    else if (pure) hipLaunchKernelGGL(cast_sp16_t_kernel<true>, grid, dim3(256), 0, s, in, out, rows, cols, ld_in, ld_out, scal, conv_T_in, conv_T_out, conv_stride, conv_toff, out_rm, ld_rm, rowmap, tap, bf);
    else hipLaunchKernelGGL(cast_sp16_t_kernel<false>, grid, dim3(256), 0, s, in, out, rows, cols, ld_in, ld_out, scal, conv_T_in, conv_T_out, conv_stride, conv_toff, out_rm, ld_rm, rowmap, tap, bf);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

void geometry(int M, int& ksplit, long long& Mp, bool pure = false) {
    ksplit = M >= 8192 ? 16 : (M >= 2048 ? 8 : (M >= 512 ? 4 : 2));
    // every range at least two k-tiles (32 pairs or 64 halfs wide); a multiple of the 128-row cast tile
    const long long q = (pure ? 128LL : 64LL) * ksplit;
    Mp = (M + q - 1) / q * q;
}

}  // namespace

int launch_cast_sp16_t(const float* in, int ld_in, float* out, long long ld_out, int rows, int cols, float* scal, hipStream_t s) {
    SOLA_ARG(in && out && rows > 0 && cols > 0 && ld_out % 128 == 0 && ld_out >= rows, "cast_sp16_t: rows=%d ld_out=%lld", rows, ld_out);
    return cast_t(0, in, ld_in, out, ld_out, rows, cols, scal, s);
}

int launch_cast_f16_t(const float* in, int ld_in, void* out, long long ld_out, int rows, int cols, float* scal, hipStream_t s, int bf16) {
    SOLA_ARG(in && out && rows > 0 && cols > 0 && ld_out % 128 == 0 && ld_out >= rows, "cast_f16_t: rows=%d ld_out=%lld", rows, ld_out);
    return cast_t(bf16 ? 2 : 1, in, ld_in, static_cast<float*>(out), ld_out, rows, cols, scal, s);
}

int launch_col2im(const float* z, float* dx, long long R, int T_in, int T_out, int cin, int k, int stride, int pad, hipStream_t s, int z_bf16) {
    SOLA_ARG(z && dx && cin % 4 == 0 && k >= 1 && stride >= 1, "col2im: cin=%d k=%d stride=%d", cin, k, stride);
    const long long n4 = R * T_in * (cin / 4);
    SolaProfScope prof(SOLA_PROF_MISC, s, 0, (z_bf16 ? 2.0 : 4.0) * R * T_out * (double)k * cin + 4.0 * R * T_in * (double)cin);
    if (z_bf16) hipLaunchKernelGGL(col2im_kernel<true>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, z, dx, n4, T_in, T_out, cin / 4, k, stride, pad);
    else hipLaunchKernelGGL(col2im_kernel<false>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, z, dx, n4, T_in, T_out, cin / 4, k, stride, pad);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

int launch_col2im_ragged(const float* z, float* dx, long long rows_in, const int4* imap, int cin, int k, int stride, int pad, hipStream_t s, int z_bf16) {
    SOLA_ARG(z && dx && imap && cin % 4 == 0 && k >= 1 && stride >= 1 && rows_in > 0, "col2im_ragged: cin=%d k=%d stride=%d", cin, k, stride);
    const long long n4 = rows_in * (cin / 4);
    SolaProfScope prof(SOLA_PROF_MISC, s, 0, (z_bf16 ? 2.0 : 4.0) * rows_in * (double)k * cin / stride + 4.0 * rows_in * (double)cin);
    if (z_bf16) hipLaunchKernelGGL(col2im_ragged_kernel<true>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, z, dx, n4, imap, cin / 4, k, stride, pad);
    else hipLaunchKernelGGL(col2im_ragged_kernel<false>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, z, dx, n4, imap, cin / 4, k, stride, pad);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

bool gemm_tn_split_supported(int M, int N, int K) { return N % 8 == 0 && K % 8 == 0 && M >= 64; }

// the row-major copy is written by the 16-byte-load path of the cast only: whole 64-column tiles, aligned slices
bool gemm_tn_split_writes_rm(const GemmTnSplitDesc& d) {
    if (!d.a_rm || d.N % 64 != 0 || d.lda % 4 != 0 || d.a_rm_ld % 8 != 0) return false;
    for (int j = 0; j < d.nprob; ++j) {
        const long long off = d.A[j] - d.A[0];
        if (off < 0 || off % 8 != 0 || off + d.N > d.a_rm_ld) return false;
    }
    return true;
}

size_t gemm_tn_split_scratch_bytes(int M, int N, int K, int nprob) {
    int ks; long long Mp;
    geometry(M, ks, Mp, true);  // the f16-operand layout pads the rows further; sized for either
    const size_t el = (size_t)nprob * N * Mp + (size_t)nprob * K * Mp + (size_t)nprob * ks * N * K + 64;
    return el * sizeof(float);
}

int launch_gemm_tn_split(const GemmTnSplitDesc& d, hipStream_t s) {
    SOLA_ARG(d.nprob >= 1 && d.nprob <= 3 && d.scratch, "gemm_tn_split: nprob %d", d.nprob);
    SOLA_ARG(gemm_tn_split_supported(d.M, d.N, d.K) && d.lda % 4 == 0, "gemm_tn_split: M=%d N=%d K=%d lda=%d", d.M, d.N, d.K, d.lda);
    SOLA_ARG(!d.conv || (d.Cin > 0 && d.K % d.Cin == 0 && (d.rowmap || (d.T_out > 0 && d.M % d.T_out == 0))), "gemm_tn_split: conv geometry K=%d Cin=%d", d.K, d.Cin);
    SOLA_ARG(d.scratch_bytes >= gemm_tn_split_scratch_bytes(d.M, d.N, d.K, d.nprob), "gemm_tn_split: scratch too small");
    int ks; long long Mp;
    const bool pure = d.pure != 0;
    const int pure_fmt = d.pure;  // 0 split-f16, 1 f16, 2 bf16
    geometry(d.M, ks, Mp, pure);
    // transposed operands: rows of Mp split-f16 pairs (4 bytes per element) or Mp halfs (2 bytes); the buffers are sized for pairs
    const long long rowf = pure ? Mp / 2 : Mp;  // floats per transposed row
    float* at = d.scratch;
    float* xt = at + (size_t)d.nprob * d.N * Mp;
    float* slabs = xt + (size_t)d.nprob * d.K * Mp;
    float* scal = d.scal ? d.scal : slabs + (size_t)d.nprob * ks * d.N * d.K;
    if (!d.scal) {  // one scale for all problems: the amax passes accumulate into the same slot
        SOLA_HIP(hipMemsetAsync(scal, 0, 2 * sizeof(float), s));
        for (int j = 0; j < d.nprob; ++j) SOLA_TRY(launch_amax_accumulate(d.A[j], d.lda, d.M, d.N, scal, s));
    }
    const bool rm = gemm_tn_split_writes_rm(d);
    const long long conv_rows_in = !d.conv ? 0 : (d.rowmap ? d.B_rows : (long long)d.M / d.T_out * d.T_in);
    if (pure && d.ldb % 4 == 0 && gemm_tn_tr_supported(d.M, d.N, d.K, 8, 8) && (!d.conv || (d.Cin % 256 == 0 && d.ldb == d.Cin && conv_rows_in > 0))) {
        // 16-bit operands, linear layers: NO transposed copies.  gemm_glds.hip's gemm_tn_tr_kernel takes the ROW-MAJOR casts - dY's is the
        // dX GEMM's operand anyway - and transposes between LDS and the matrix pipe (ds_read_b64_tr_b16).
        const int bf = pure_fmt == 2 ? 1 : 0;
        GemmTnTrDesc t{};
        t.nprob = d.nprob; t.M = d.M; t.N = d.N; t.K = d.K; t.bf16 = bf;
        _Float16* a16 = reinterpret_cast<_Float16*>(at);
        _Float16* x16 = reinterpret_cast<_Float16*>(xt);
        const bool rm16 = rm && !d.rm_split;  // the caller's row-major copy has the operand's format: cast once, into it
        // ... or it is the split-f16 copy (the dX GEMM of a split-f16 step): its hi halves ARE the plain f16 cast, the kernel fetches them
        const bool rmsp = rm && d.rm_split && !bf;
        int n_x = 0;
        bool b_kept = !d.scal_b && !(d.b16_split && bf);  // every problem's X comes from the forward's kept operand casts
        for (int j = 0; j < d.nprob; ++j) b_kept = b_kept && d.B16[j] != nullptr;
        for (int j = 0; j < d.nprob; ++j) {
            const long long off = d.A[j] - d.A[0];
            if (rm16) {
                _Float16* dst = reinterpret_cast<_Float16*>(d.a_rm) + off;
                if (!d.a_rm_ready) SOLA_TRY(launch_cast_f16_scaled(d.A[j], d.lda, dst, d.a_rm_ld, d.M, d.N, scal, s, bf));
                t.A[j] = dst;
            } else if (rmsp) {
                SOLA_TRY(launch_cast_sp16_scaled(d.A[j], d.lda, d.a_rm + off, d.a_rm_ld, d.M, d.N, scal, s));
                t.A[j] = d.a_rm + off;
            } else {
                _Float16* dst = a16 + (size_t)j * d.M * d.N;
                SOLA_TRY(launch_cast_f16_scaled(d.A[j], d.lda, dst, d.N, d.M, d.N, scal, s, bf));
                t.A[j] = dst;
                if (rm) SOLA_TRY(launch_cast_sp16_scaled(d.A[j], d.lda, d.a_rm + off, d.a_rm_ld, d.M, d.N, scal, s));  // the dX GEMM keeps split-f16 operands
            }
            t.B[j] = nullptr;
            for (int e = 0; e < j; ++e)
                if (d.B[e] == d.B[j]) t.B[j] = t.B[e];
            if (!t.B[j] && b_kept) t.B[j] = d.B16[j];  // the forward's operand cast of the same activation
            if (!t.B[j]) {
                // conv: ONE row-major cast of the conv input; the kernel gathers the taps (implicit im2col) in its DMA addresses
                const long long b_rows = d.conv ? conv_rows_in : d.M;
                const int b_cols = d.conv ? d.Cin : d.K;
                _Float16* dst = x16 + (size_t)n_x++ * b_rows * b_cols;
                if (d.scal_b) SOLA_TRY(launch_cast_f16_scaled(d.B[j], d.ldb, dst, b_cols, b_rows, b_cols, d.scal_b, s, bf));
                else SOLA_TRY(launch_cast_f16(d.B[j], d.ldb, dst, b_cols, b_rows, b_cols, 1.f, nullptr, s, 13, nullptr, bf));
                t.B[j] = dst;
            }
        }
        t.lda = (rm16 || rmsp) ? d.a_rm_ld : d.N;
        t.a_split = rmsp ? 1 : 0;
        t.ldb = d.conv ? d.Cin : d.K;
        t.b_split = b_kept && d.b16_split ? 1 : 0;
        t.conv = d.conv; t.Cin = d.Cin; t.T_in = d.T_in; t.T_out = d.T_out; t.stride = d.stride; t.pad = d.pad; t.rowmap = d.rowmap;
        gemm_tn_tr_geometry(d.M, d.N, d.K, d.nprob, ks, t.ksplit, t.kper);
        t.part = slabs;
        // bf16 steps (train_bf16_store 3): the partial sums of the k ranges leave as bfloat16 slabs (half the bytes written by the 256 blocks
        // at once and read back by the fold); sixteen of them are summed in f32 - autocast's linear backward rounds the whole dW to bfloat16 once
        extern int g_train_bf16_store;
        t.part_bf16 = (bf && g_train_bf16_store >= 3 && d.K % 4 == 0) ? 1 : 0;
        SOLA_TRY(launch_gemm_tn_tr(t, s));
        if (t.part_bf16) return launch_splitk_reduce_bf16(slabs, t.ksplit, d.nprob, d.C, d.N, d.K, d.K, scal + 1, d.scal_b ? d.scal_b + 1 : nullptr, s);
        return launch_splitk_reduce(slabs, t.ksplit, d.nprob, d.C, d.N, d.K, d.K, scal + 1, d.scal_b ? d.scal_b + 1 : nullptr, s);
    }
    const float* xt_of[3] = {nullptr, nullptr, nullptr};
    int n_xt = 0;
    for (int j = 0; j < d.nprob; ++j) {
        if (rm && !(d.a_rm_ready && pure && !d.rm_split)) {  // + the row-major cast of the same gradient matrix (the dX GEMM's operand): one read of dY for both
            const long long off = d.A[j] - d.A[0];
            float* dst = (pure && !d.rm_split) ? reinterpret_cast<float*>(reinterpret_cast<_Float16*>(d.a_rm) + off) : d.a_rm + off;
            SOLA_TRY(cast_t(pure_fmt, d.A[j], d.lda, at + (size_t)j * d.N * rowf, Mp, d.M, d.N, scal, s, 0, 0, 1, 0, dst, d.a_rm_ld, nullptr, 0, d.rm_split));
        } else {
            SOLA_TRY(cast_t(pure_fmt, d.A[j], d.lda, at + (size_t)j * d.N * rowf, Mp, d.M, d.N, scal, s));
        }
        for (int e = 0; e < j; ++e)
            if (d.B[e] == d.B[j]) xt_of[j] = xt_of[e];
        if (!xt_of[j]) {
            float* dst = xt + (size_t)n_xt++ * d.K * rowf;
            if (d.conv) {  // rows [kk*Cin, (kk+1)*Cin) of X^T = tap kk of the implicit im2col
                for (int kk = 0; kk < d.K / d.Cin; ++kk)
                    SOLA_TRY(cast_t(pure_fmt, d.B[j], d.ldb, dst + (size_t)kk * d.Cin * rowf, Mp, d.M, d.Cin, d.scal_b, s, d.T_in, d.T_out, d.stride, kk - d.pad,
                                    nullptr, 0, d.rowmap, kk));
            } else {
                SOLA_TRY(cast_t(pure_fmt, d.B[j], d.ldb, dst, Mp, d.M, d.K, d.scal_b, s));
            }
            xt_of[j] = dst;
        }
    }
    GemmDesc g{};
    g.nprob = d.nprob;
    for (int j = 0; j < d.nprob; ++j) g.p[j] = GemmProblem{at + (size_t)j * d.N * rowf, xt_of[j], nullptr, nullptr, d.C[j]};
    g.M = d.N; g.N = d.K; g.K = (int)Mp; g.lda = (int)Mp; g.ldr = 0; g.ldc = d.K;
    g.arith = pure ? 2 : 1; g.bf16 = pure_fmt == 2 ? 1 : 0; g.out_scale = 1.f; g.out_scale_dev = scal + 1;
    if (d.scal_b)  // the activations' own scale (the caller's tokens): every problem shares it
        for (int j = 0; j < d.nprob; ++j) g.p[j].scale_dev = d.scal_b + 1;
    g.ksplit = ks; g.splitk_ws = slabs; g.splitk_bytes = (size_t)d.nprob * ks * d.N * d.K * sizeof(float);
    return launch_gemm(g, s);
}
