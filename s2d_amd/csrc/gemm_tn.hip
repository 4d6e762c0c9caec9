// Weight-gradient contraction C[n][k] = sum_m A[m][n] * B[m][k] (dW = dY^T . X) in the split-fp16 x3 arithmetic of the forward
// kernels, WITHOUT transposed copies of the operands: the row-major tiles (32 contraction rows x 128 / 64 columns) are
// loaded with coalesced 16-B loads and transposed on their way into LDS -- two consecutive contraction rows are packed
// into one fp16 pair, so every LDS write is a 32-bit store into the [column][k] image the MFMA fragments are read from
// (the image of gemm_f16x3_hi_kernel: rows of [16 words hi | 16 words lo | 4 pad]); the columns a thread owns are spread
// over LDS rows 32 apart (a permutation of the output rows, undone in the epilogue) and a half-wave covers 8 column groups x
// 4 row pairs, so that its 32 stores hit 32 different banks (adjacent rows would give 2 banks x 16).
// The contraction (3e5 .. 4e6 rows) is cut into slices, one per blockIdx.y; slice s writes its partial [Mo][No] tile to
// C + s * slice_stride and s2d_reduce_slices_f32 adds the slices in a fixed order (reproducible; the taps of a convolution
// interleave their partial tiles, slice_stride = taps * Mo * No, so that ONE reduction finishes all of them).  B may start `shift` rows later
// than A (a convolution tap on the zero-padded grid): rows past its end read as zero.
#include "common.h"
#include <stdlib.h>

namespace {

typedef __fp16 h16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int TBK = 32, TROWW = 36;
constexpr unsigned int TOOB = 0xFFFFFFF0u;

struct TnParams {
    const float *A, *B;
    float *C;
    int Mo, No;                 // output rows (columns of A), output columns (columns of B)
    long rowsA, rowsB;          // contraction rows available in A / B (B may be shorter: shifted view)
    long lda, ldb, chunk;
    long sC;                    // elements between the partial tiles of consecutive slices (>= Mo * No)
    float *colsum;              // optional [slices][Mo]: per-slice column sums of A (the bias gradient of the same dY), or NULL
    // CONV form (s2d_conv_wgrad_tn_f32): A = dY [N * Ho * Wo][Co], B = X [N * H * W][Ci]; workgroup (tile, tap): row m = (n, yo, xo) of A meets
    // row (n, yo * stride + ky - pad, xo * stride + kx - pad) of B, or zeros when that pixel lies outside the image
    int taps, KW, stride, pad, H, W, Ho, Wo;
    unsigned int mgWo, shWo, mgHo, shHo;   // round-up magic numbers of the divisions by Wo and Ho (exact for every 32-bit dividend)
    long ldc;                   // CONV: elements between output rows of C (taps * No: the partial tile of a tap is a column block of [Mo][taps][No])
};

// q = m / d for the (magic, shift) pair magic_u32() makes of d
__device__ __forceinline__ unsigned int div_magic(unsigned int m, unsigned int magic, unsigned int shift)
{
    const unsigned int t = __umulhi(m, magic);
    return (t + ((m - t) >> 1)) >> shift;
}

// hi / lo fp16 pairs of two values that are consecutive along the contraction
__device__ __forceinline__ void split_pair(float a, float b, unsigned int &hi, unsigned int &lo)
{
    const h16x2 h = __builtin_amdgcn_cvt_pkrtz(a, b);
    const f32x2 f = __builtin_convertvector(h, f32x2);
    const h16x2 l = __builtin_amdgcn_cvt_pkrtz((a - f[0]) * 2048.f, (b - f[1]) * 2048.f);
    hi = __builtin_bit_cast(unsigned int, h);
    lo = __builtin_bit_cast(unsigned int, l);
}

// BNs = 128 (outputs at least 128 columns wide): a wave owns 64 x 64 of a 128 x 128 tile -- per 32 contraction rows 24 MFMAs against 8
// loads, 16 splits and 32 LDS stores per thread, where the 128 x 64 tile has 12 MFMAs against 6 loads, 12 splits and 24 stores: the
// transposing / splitting work per MFMA is what bounds this kernel, not the MFMAs.
template <int BNs, bool CONV = false, bool MF16 = false>
__global__ __launch_bounds__(256, BNs == 128 ? 2 : 4) void gemm_tn_f16x3_kernel(TnParams p)
{
    constexpr int BM = 128, NB = BNs / 64;                  // NB: 32-column MFMA tiles per wave along the output columns
    __shared__ __attribute__((aligned(16))) unsigned int As[BM * TROWW];
    __shared__ __attribute__((aligned(16))) unsigned int Bs[BNs * TROWW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l32 = lane & 31, h = lane >> 5;
    const int tiles_n = (p.No + BNs - 1) / BNs;
    // CONV: the taps of one tile are neighbours in the grid, so the A rows they all read and the B rows they share meet in L2
    const int tap = CONV ? (int)(blockIdx.x % (unsigned int)p.taps) : 0;
    const int tile = CONV ? (int)(blockIdx.x / (unsigned int)p.taps) : (int)blockIdx.x;
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BNs;
    const int tap_dy = CONV ? tap / p.KW - p.pad : 0, tap_dx = CONV ? tap % p.KW - p.pad : 0;
    const long r_lo = (long)blockIdx.y * p.chunk;
    const long r_hi = r_lo + p.chunk < p.rowsA ? r_lo + p.chunk : p.rowsA;
    const int nk = (int)((r_hi - r_lo + TBK - 1) / TBK);

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.A), 0, (int)(p.rowsA * p.lda * 4L), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.B), 0, (int)(p.rowsB * p.ldb * 4L), 0x00020000);
    // A tile: 32 rows x 128 columns; a thread takes 4 columns of the row pairs (2 ra, 2 ra + 1) and (2 ra + 16, 2 ra + 17)
    // (a half-wave = 8 column groups x 4 row pairs: its 32 transposing stores land in 32 different banks, its loads are 128-B runs)
    const int ca = (tid & 7) | (((tid >> 5) & 3) << 3), ra_ = ((tid >> 3) & 3) | ((tid >> 7) << 2);
    // B tile: 32 rows x 64 columns; a thread takes 4 columns of the row pair (2 rb, 2 rb + 1)
    // (BNs = 128: the A tile's mapping)
    const int cb = BNs == 128 ? ca : (tid & 7) | (((tid >> 5) & 1) << 3), rb_ = BNs == 128 ? ra_ : ((tid >> 3) & 3) | ((tid >> 6) << 2);
    const unsigned int a_col = (unsigned int)(m0 + 4 * ca), b_col = (unsigned int)(n0 + 4 * cb);
    const unsigned int a_bad = a_col < (unsigned int)p.Mo ? 0u : TOOB, b_bad = b_col < (unsigned int)p.No ? 0u : TOOB;
    f32x4 va[4], vb[2 * NB];
    // column sums of A ride along in the workgroups of the first column tile: dY is in registers here anyway (a separate pass would
    // read it again: 1.27 GB for the encoder's linear1)
    const bool do_cs = !CONV && p.colsum != nullptr && (tile % tiles_n) == 0;
    f32x4 cs = {0.f, 0.f, 0.f, 0.f};
    auto load_tile = [&](int kt) {
        const long r = r_lo + (long)kt * TBK;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const long m = r + 2 * ra_ + t + 16 * i;
                const unsigned int off = (unsigned int)((m * p.lda + a_col) * 4L) | a_bad | (m < r_hi ? 0u : TOOB);
                va[2 * i + t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)off, 0, 0));
            }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if constexpr (CONV) {
                // (n, yo, xo) of the pair's even row by two magic divisions, the odd row by carry
                const unsigned int me = (unsigned int)(r + 2 * rb_ + 16 * i);
                unsigned int ry = div_magic(me, p.mgWo, p.shWo);            // n * Ho + yo
                int xo = (int)(me - ry * (unsigned int)p.Wo);
                unsigned int nn = div_magic(ry, p.mgHo, p.shHo);
                int yo = (int)(ry - nn * (unsigned int)p.Ho);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int yi = yo * p.stride + tap_dy, xi = xo * p.stride + tap_dx;
                    const bool ok = (long)me + t < r_hi && (unsigned int)yi < (unsigned int)p.H && (unsigned int)xi < (unsigned int)p.W;
                    const unsigned int rowb = (nn * (unsigned int)p.H + (unsigned int)yi) * (unsigned int)p.W + (unsigned int)xi;
                    const unsigned int off = ((rowb * (unsigned int)p.ldb + b_col) * 4u) | b_bad | (ok ? 0u : TOOB);
                    vb[2 * i + t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)off, 0, 0));
                    if (++xo == p.Wo) { xo = 0; if (++yo == p.Ho) { yo = 0; ++nn; } }
                }
            } else {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const long m = r + 2 * rb_ + t + 16 * i;
                    const unsigned int off = (unsigned int)((m * p.ldb + b_col) * 4L) | b_bad | (m < r_hi && m < p.rowsB ? 0u : TOOB);
                    vb[2 * i + t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)off, 0, 0));
                }
            }
        }
    };
    auto store_tile = [&]() {
        if (do_cs) cs += (va[0] + va[1]) + (va[2] + va[3]);          // rows past the slice read as zero
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned int hi, lo;
                split_pair(va[2 * i][j], va[2 * i + 1][j], hi, lo);
                unsigned int *row = &As[(32 * j + ca) * TROWW + ra_ + 8 * i];       // column 4 ca + j lives in LDS row 32 j + ca
                row[0] = hi; row[16] = lo;
            }
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned int hi, lo;
                split_pair(vb[2 * i][j], vb[2 * i + 1][j], hi, lo);
                unsigned int *row = &Bs[((BNs / 4) * j + cb) * TROWW + rb_ + 8 * i];    // column 4 cb + j lives in LDS row (BNs / 4) j + cb
                row[0] = hi; row[16] = lo;
            }
    };
    // MF16: the same products on v_mfma_f32_16x16x32_f16 (one k-step of 32 per 16 x 16 tile instead of two of 16 per 32 x 32 tile: equal cycles
    // per FLOP, but the chip holds a higher clock under that shape -- MI355X_MICROARCH.md, "clock under load", item 7)
    f32x16 accm[2][NB], accx[2][NB];
    f32x4 acm[MF16 ? 4 : 1][MF16 ? 2 * NB : 1], acx[MF16 ? 4 : 1][MF16 ? 2 * NB : 1];
    if constexpr (MF16) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < 2 * NB; ++t) { acm[i][t] = f32x4(0.f); acx[i][t] = f32x4(0.f); }
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int t = 0; t < NB; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) { accm[i][t][r] = 0.f; accx[i][t][r] = 0.f; }
    }
    const int l16 = lane & 15, kq = lane >> 4;
    if (nk > 0) load_tile(0);
    for (int kt = 0; kt < nk; ++kt) {
        store_tile();
        __syncthreads();
        if (kt + 1 < nk) load_tile(kt + 1);
        if constexpr (MF16) {
            // lane (row l16, k quarter kq) of a 16-row tile reads 8 consecutive k: words 4 kq .. 4 kq + 3 of the row's hi half, + 16 for lo
            const unsigned int *as = &As[(wm * 64 + l16) * TROWW + 4 * kq];
            const unsigned int *bs = &Bs[(wn * 32 * NB + l16) * TROWW + 4 * kq];
            f16x8 bh[2 * NB], bl[2 * NB];
#pragma unroll
            for (int t = 0; t < 2 * NB; ++t) {
                bh[t] = *reinterpret_cast<const f16x8 *>(bs + t * 16 * TROWW);
                bl[t] = *reinterpret_cast<const f16x8 *>(bs + t * 16 * TROWW + 16);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f16x8 ah = *reinterpret_cast<const f16x8 *>(as + i * 16 * TROWW);
                const f16x8 al = *reinterpret_cast<const f16x8 *>(as + i * 16 * TROWW + 16);
#pragma unroll
                for (int t = 0; t < 2 * NB; ++t) {
                    acx[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[t], acx[i][t], 0, 0, 0);
                    acx[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[t], acx[i][t], 0, 0, 0);
                    acm[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[t], acm[i][t], 0, 0, 0);
                }
            }
        } else {
        const unsigned int *as = &As[(wm * 64 + l32) * TROWW + 4 * h];
        const unsigned int *bs = &Bs[(wn * 32 * NB + l32) * TROWW + 4 * h];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f16x8 bh[NB], bl[NB];
#pragma unroll
            for (int t = 0; t < NB; ++t) {
                bh[t] = *reinterpret_cast<const f16x8 *>(bs + t * 32 * TROWW + 8 * s);
                bl[t] = *reinterpret_cast<const f16x8 *>(bs + t * 32 * TROWW + 16 + 8 * s);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f16x8 ah = *reinterpret_cast<const f16x8 *>(as + i * 32 * TROWW + 8 * s);
                const f16x8 al = *reinterpret_cast<const f16x8 *>(as + i * 32 * TROWW + 16 + 8 * s);
#pragma unroll
                for (int t = 0; t < NB; ++t) {
                    accx[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[t], accx[i][t], 0, 0, 0);
                    accx[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[t], accx[i][t], 0, 0, 0);
                    accm[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[t], accm[i][t], 0, 0, 0);
                }
            }
        }
        }
        __syncthreads();
    }
    if (do_cs) {                                             // the 8 threads that share 4 columns, added in a fixed order (As is free: the loop ended on a barrier)
        float *red = reinterpret_cast<float *>(As);              // [8][128]
        *reinterpret_cast<f32x4 *>(red + ra_ * 128 + 4 * ca) = cs;
        __syncthreads();
        if (tid < 128 && m0 + tid < p.Mo) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) t += red[r * 128 + tid];
            p.colsum[(long)blockIdx.y * p.Mo + m0 + tid] = t;
        }
    }
    float *C = p.C + (long)blockIdx.y * p.sC + (CONV ? (long)tap * p.No : 0L);
    const long ldc = CONV ? p.ldc : (long)p.No;
    // LDS row r of the A image holds output row 4 (r % 32) + r / 32, LDS row c of the B image output column 4 (c % 16) + c / 16
    // (the permutation that spreads the transposing stores over the banks); undo it here
    if constexpr (MF16) {
        // 16 x 16 tile (i, t) of the wave: D[row 4 kq + r][col l16]; A-image LDS row ra = 64 wm + 16 i + 4 kq + r, B-image LDS row cb = 32 NB wn + 16 t + l16
#pragma unroll
        for (int t = 0; t < 2 * NB; ++t) {
            const int cb = 32 * NB * wn + 16 * t + l16;
            const int col = BNs == 128 ? n0 + 4 * (cb & 31) + (cb >> 5) : n0 + 4 * (cb & 15) + (cb >> 4);
            if (col >= p.No) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ra = 64 * wm + 16 * i + 4 * kq + r;
                    const int row = m0 + 4 * (ra & 31) + (ra >> 5);
                    if (row < p.Mo) C[(long)row * ldc + col] = acm[i][t][r] + acx[i][t][r] * (1.0f / 2048.0f);
                }
        }
        return;
    }
#pragma unroll
    for (int tn = 0; tn < NB; ++tn) {
        // BNs = 128: LDS row c = 64 wn + 32 tn + l32 of the B image <-> output column 4 (c % 32) + c / 32
        const int col = BNs == 128 ? n0 + 4 * l32 + 2 * wn + tn : n0 + 4 * (l32 & 15) + 2 * wn + (l32 >> 4);
        if (col >= p.No) continue;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2) + 4 * h;
                const int row = m0 + 4 * rr + 2 * wm + tm;
                if (row < p.Mo) C[(long)row * ldc + col] = accm[tm][tn][r] + accx[tm][tn][r] * (1.0f / 2048.0f);
            }
    }
}

}  // namespace

// S2D_TN_MFMA16 (read per call; A/B runs in one process): 1 = the 16x16x32 form of the matrix instructions, 0 = 32x32x16
static bool tn_mfma16()
{
    const char *e = getenv("S2D_TN_MFMA16");
    return e ? atoi(e) != 0 : false;
}

extern "C" int s2d_gemm_tn_f32(const float *A, const float *B, float *C_slices, int Mo, int No, long rowsA, long rowsB, long lda, long ldb,
                               long chunk, long slice_stride, float *colsum_slices, hipStream_t stream)
{
    if (slice_stride == 0) slice_stride = (long)Mo * No;
    if (slice_stride < (long)Mo * No) return S2D_ERR_ARG;
    if (Mo <= 0 || No <= 0 || rowsA <= 0 || rowsB <= 0 || chunk <= 0 || (chunk & 31) || (Mo & 3) || (No & 3) || (lda & 3) || (ldb & 3) ||
        lda < Mo || ldb < No)
        return S2D_ERR_ARG;
    if (rowsA * lda * 4L > 0xFFFFFF00L || rowsB * ldb * 4L > 0xFFFFFF00L) return S2D_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) return S2D_ERR_ARG;
    const long S = (rowsA + chunk - 1) / chunk;
    if (S > 65535) return S2D_ERR_ARG;
    TnParams p{};
    p.A = A; p.B = B; p.C = C_slices; p.Mo = Mo; p.No = No; p.rowsA = rowsA; p.rowsB = rowsB; p.lda = lda; p.ldb = ldb; p.chunk = chunk;
    p.sC = slice_stride; p.colsum = colsum_slices;
    static int wide = -1;                                   // S2D_TN_WIDE=0: the 128 x 64 tile for every shape (A/B runs)
    if (wide < 0) { const char *e = getenv("S2D_TN_WIDE"); wide = e ? atoi(e) : 1; }
    const bool mf16 = tn_mfma16();
    if (wide && No >= 128) {
        if (mf16) hipLaunchKernelGGL((gemm_tn_f16x3_kernel<128, false, true>), dim3(cdiv(Mo, 128) * cdiv(No, 128), (int)S), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL(gemm_tn_f16x3_kernel<128>, dim3(cdiv(Mo, 128) * cdiv(No, 128), (int)S), dim3(256), 0, stream, p);
    } else {
        if (mf16) hipLaunchKernelGGL((gemm_tn_f16x3_kernel<64, false, true>), dim3(cdiv(Mo, 128) * cdiv(No, 64), (int)S), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL(gemm_tn_f16x3_kernel<64>, dim3(cdiv(Mo, 128) * cdiv(No, 64), (int)S), dim3(256), 0, stream, p);
    }
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

// round-up magic number of an unsigned 32-bit division (Granlund & Montgomery): q = (t + ((m - t) >> 1)) >> shift, t = mulhi(m, magic)
static void magic_u32(unsigned int d, unsigned int *magic, unsigned int *shift)
{
    unsigned int l = 0;
    while ((1ull << l) < d) ++l;                              // ceil(log2 d)
    *magic = (unsigned int)(((1ull << 32) * ((1ull << l) - d)) / d + 1ull);
    *shift = l - 1;                                           // d >= 2 (the entry point rejects one-pixel outputs)
}

/* Weight gradient of y = conv2d_nhwc(x, w [Cout][KH][KW][Cin], stride, pad) (detectron2's Conv2d in the R50 trunk and the pixel decoder,
 * e.g. mask2former/modeling/pixel_decoder/msdeformattn.py:289-303; autograd's conv weight gradient in the reference), in the TN kernel's
 * arithmetic with the input pixel of every (output position, tap) addressed in place: no zero-padded copy of x, no copy of dY scattered
 * onto the input grid (round 5; a stride-2 convolution walked four times its output positions that way).  One launch: workgroup = (output
 * tile, tap, slice of the positions); slice s leaves its partial gradients in part[s] laid out [Cout][KH][KW][Cin];
 * s2d_reduce_slices_f32(part, S, Cout*KH*KW*Cin, ...) finishes all taps.  chunk: positions per slice, a multiple of 32. */
extern "C" int s2d_conv_wgrad_tn_f32(const float *dy, const float *x, int N, int H, int W, int Ci, int Ho, int Wo, int Co, int KH, int KW,
                                     int stride, int pad, long chunk, float *part, hipStream_t stream)
{
    const long rowsA = (long)N * Ho * Wo, rowsB = (long)N * H * W;
    if (N <= 0 || H <= 0 || W <= 0 || Ho <= 1 || Wo <= 1 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0 || chunk <= 0 || (chunk & 31) || (Co & 3) || (Ci & 3) ||
        Co <= 0 || Ci <= 0)
        return S2D_ERR_ARG;
    if ((long)(Ho - 1) * stride + KH - pad > H + pad || (long)(Wo - 1) * stride + KW - pad > W + pad) return S2D_ERR_ARG;      // not the output size of this convolution
    if (rowsA * Co * 4L > 0xFFFFFF00L || rowsB * Ci * 4L > 0xFFFFFF00L || rowsA >= (1L << 31)) return S2D_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x)) & 15) return S2D_ERR_ARG;
    const long S = (rowsA + chunk - 1) / chunk;
    const int taps = KH * KW;
    if (S > 65535) return S2D_ERR_ARG;
    TnParams p{};
    p.A = dy; p.B = x; p.C = part; p.Mo = Co; p.No = Ci; p.rowsA = rowsA; p.rowsB = rowsB; p.lda = Co; p.ldb = Ci; p.chunk = chunk;
    p.sC = (long)Co * taps * Ci; p.colsum = nullptr;
    p.taps = taps; p.KW = KW; p.stride = stride; p.pad = pad; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.ldc = (long)taps * Ci;
    magic_u32((unsigned int)Wo, &p.mgWo, &p.shWo);
    magic_u32((unsigned int)Ho, &p.mgHo, &p.shHo);
    const long tiles = (long)cdiv(Co, 128) * cdiv(Ci, Ci >= 128 ? 128 : 64);
    if (tiles * taps > 0x7fffffffL) return S2D_ERR_ARG;
    const bool mf16 = tn_mfma16();
    if (Ci >= 128) {
        if (mf16) hipLaunchKernelGGL((gemm_tn_f16x3_kernel<128, true, true>), dim3((unsigned int)(tiles * taps), (int)S), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((gemm_tn_f16x3_kernel<128, true>), dim3((unsigned int)(tiles * taps), (int)S), dim3(256), 0, stream, p);
    } else {
        if (mf16) hipLaunchKernelGGL((gemm_tn_f16x3_kernel<64, true, true>), dim3((unsigned int)(tiles * taps), (int)S), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((gemm_tn_f16x3_kernel<64, true>), dim3((unsigned int)(tiles * taps), (int)S), dim3(256), 0, stream, p);
    }
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}
