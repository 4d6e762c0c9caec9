// Dense contractions at fp32-class accuracy on the bf16 matrix cores ("split-bf16 x3").
//
// Every f32 operand x is split on the fly into hi = bf16(x) and lo = bf16(x - hi) while it is staged into LDS, and
//     A.B^T  ~=  Ah.Bh^T + Ah.Bl^T + Al.Bh^T          (the dropped Al.Bl^T term is ~2^-16 relative)
// is accumulated in f32 by three v_mfma_f32_32x32x16_bf16 per tile and k-step.  gfx950 has no TF32/xf32 path and its
// fp32-input MFMA runs at 1/16 of the bf16 rate, so three bf16 MFMAs deliver fp32-class results (observed relative
// error ~1e-6..1e-5, well inside the path's 1e-3 parity budget) at ~5x the fp32-MFMA throughput.
//
// Same NT-GEMM / implicit-GEMM-conv contract and epilogue as gemm.hip (C = act(A.B^T * scale + bias + res)).
// Tile: (64*WM) x 128 x 32 per workgroup, WM*2 waves, each wave 64x64 = 2x2 MFMA tiles.  WM = 4 (256x128, 512
// threads, 1 workgroup/CU = 2 waves/SIMD) for large M, WM = 2 (128x128) otherwise.  Staging: global f32 (16-B loads)
// -> registers -> split (v_cvt_pk_bf16_f32, v_pk_add_f32) -> LDS rows of [16 words hi | 16 words lo | 4 pad]
// (stride 36 words: conflict-free for ds_read_b128 fragment reads and the ds_write_b64 stores), double buffered.
#include "common.h"
#include "gemm_params.h"
#include "dropout.h"
#include <stdlib.h>

namespace {

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int BN = 128, BK = 32, ROWW = 36;  // LDS row = 36 words (144 B)

__device__ __forceinline__ void split4(const f32x4 v, u32x2 &hi, u32x2 &lo)
{
    const f32x2 a = {v[0], v[1]}, b = {v[2], v[3]};
    const bf16x2 ha = __builtin_convertvector(a, bf16x2), hb = __builtin_convertvector(b, bf16x2);
    const unsigned int ua = __builtin_bit_cast(unsigned int, ha), ub = __builtin_bit_cast(unsigned int, hb);
    const f32x2 fa = {__uint_as_float(ua << 16), __uint_as_float(ua & 0xFFFF0000u)};
    const f32x2 fb = {__uint_as_float(ub << 16), __uint_as_float(ub & 0xFFFF0000u)};
    const bf16x2 la = __builtin_convertvector(a - fa, bf16x2), lb = __builtin_convertvector(b - fb, bf16x2);
    hi[0] = ua; hi[1] = ub;
    lo[0] = __builtin_bit_cast(unsigned int, la); lo[1] = __builtin_bit_cast(unsigned int, lb);
}

template <int WM, bool CONV>
__global__ __launch_bounds__(128 * WM, 1) void gemm_bf16x3_kernel(GemmParams p)
{
    constexpr int BM = 64 * WM, NT = 128 * WM, RPT = NT / 8;     // RPT: rows covered per staging pass
    constexpr int A_IT = BM / RPT, B_IT = BN / RPT;              // float4 loads per thread per k-tile (4 / 4 or 2)
    extern __shared__ __attribute__((aligned(16))) unsigned int lds[];
    unsigned int *As = lds;                         // [2][BM][ROWW]
    unsigned int *Bs = lds + 2 * BM * ROWW;         // [2][BN][ROWW]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l32 = lane & 31, h = lane >> 5;

    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {   // XCD-aware order: consecutive tile ids (sharing an A row panel) stay on one XCD (bijective remap)
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, within = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    }
    const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int bz = blockIdx.y;
    const float *A = p.A + (long)bz * p.sA;
    const float *B = p.B + (long)bz * p.sB;
    float *C = p.C + (long)bz * p.sC;

    // Staging coordinates.  Loads are branch-free buffer loads: an out-of-range row / k / convolution tap gets an
    // offset beyond the descriptor's size and the hardware returns zeros (no exec-mask branches, so the compiler
    // can pipeline loads, splits and MFMAs in one basic block).  Rows of consecutive 8-lane groups are 4 apart
    // (mod 8): their ds_write_b64 land 16 banks apart instead of overlapping (2-way conflict with adjacent rows).
    const int c4 = tid & 7, g = tid >> 3;
    const int r0 = (g & ~7) | ((g & 1) << 2) | ((g >> 1) & 3);
    constexpr unsigned int OOB = 0xFFFFFFF0u;      // any 16-B load at this offset is out of range -> returns zeros
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A), 0, (int)p.bytesA, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(B), 0, (int)p.bytesB, 0x00020000);
    unsigned int a_off[A_IT], a_bad[A_IT];      // dense: byte offset of the row; conv: byte offset of the image; a_bad = OOB bits for rows >= M
    int a_iy0[A_IT], a_ix0[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + r0 + RPT * i;
        const bool ok = m < p.M;
        if (CONV) {
            const int mm = ok ? m : 0;
            const int ox = mm % p.Wout, t = mm / p.Wout, oy = t % p.Hout, n = t / p.Hout;
            a_iy0[i] = ok ? oy * p.stride - p.pad : -(1 << 20);     // invalid row: every tap is out of the image
            a_ix0[i] = ox * p.stride - p.pad;
            a_off[i] = (unsigned int)((long)n * p.Hin * p.Win * p.Cin * 4L);
        } else {
            a_off[i] = ok ? (unsigned int)((long)m * p.lda * 4L) : 0u;
            a_iy0[i] = a_ix0[i] = 0;
        }
        a_bad[i] = ok ? 0u : OOB;
    }
    unsigned int b_off[B_IT], b_bad[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int n = n0 + r0 + RPT * i;
        b_off[i] = n < p.N ? (unsigned int)((long)n * p.ldb * 4L) : 0u;
        b_bad[i] = n < p.N ? 0u : OOB;
    }

    f32x4 ra0[A_IT], rb0[B_IT], ra1[A_IT], rb1[B_IT];
    auto load_tile = [&](int kt, f32x4 (&ra)[A_IT], f32x4 (&rb)[B_IT]) {
        const int k = kt * BK + c4 * 4;
        const unsigned int kmask = (unsigned int)((p.K - 1 - k) >> 31) & OOB;   // all-OOB bits when k >= K (no select: keeps loads unconditional)
        int kh = 0, kw = 0, ci = 0;
        if (CONV) {
            const int tap = k / p.Cin;
            ci = k - tap * p.Cin;
            kh = tap / p.KW;
            kw = tap - kh * p.KW;
        }
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            unsigned int off;
            if (CONV) {
                const int iy = a_iy0[i] + kh, ix = a_ix0[i] + kw;
                const unsigned int tmask = (unsigned int)(((iy | ix | (p.Hin - 1 - iy) | (p.Win - 1 - ix)) >> 31)) & OOB;   // tap outside the image
                off = (a_off[i] + (unsigned int)(((iy * p.Win + ix) * p.Cin + ci) * 4)) | tmask | kmask;
            } else {
                off = (a_off[i] + (unsigned int)(k * 4)) | a_bad[i] | kmask;
            }
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)off, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i)
            rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                rsB, (int)((b_off[i] + (unsigned int)(k * 4)) | b_bad[i] | kmask), 0, 0));
    };
    auto store_tile = [&](int buf, f32x4 (&ra)[A_IT], f32x4 (&rb)[B_IT]) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            u32x2 hi, lo;
            split4(ra[i], hi, lo);
            unsigned int *row = &As[(buf * BM + r0 + RPT * i) * ROWW];
            *reinterpret_cast<u32x2 *>(row + c4 * 2) = hi;
            *reinterpret_cast<u32x2 *>(row + 16 + c4 * 2) = lo;
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            u32x2 hi, lo;
            split4(rb[i], hi, lo);
            unsigned int *row = &Bs[(buf * BN + r0 + RPT * i) * ROWW];
            *reinterpret_cast<u32x2 *>(row + c4 * 2) = hi;
            *reinterpret_cast<u32x2 *>(row + 16 + c4 * 2) = lo;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int cur) {
        // fragment of k-step s: 8 bf16 (16 B) at word 8*s + 4*h (hi) / 16 + 8*s + 4*h (lo) of the lane's row
        const unsigned int *as = &As[(cur * BM + wm * 64 + l32) * ROWW + 4 * h];
        const unsigned int *bs = &Bs[(cur * BN + wn * 64 + l32) * ROWW + 4 * h];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                ah[t] = *reinterpret_cast<const bf16x8 *>(as + t * 32 * ROWW + 8 * s);
                al[t] = *reinterpret_cast<const bf16x8 *>(as + t * 32 * ROWW + 16 + 8 * s);
                bh[t] = *reinterpret_cast<const bf16x8 *>(bs + t * 32 * ROWW + 8 * s);
                bl[t] = *reinterpret_cast<const bf16x8 *>(bs + t * 32 * ROWW + 16 + 8 * s);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    // small terms first, the dominant hi.hi last
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
    };

    const int nk = (p.K + BK - 1) / BK;
    load_tile(0, ra0, rb0);
    if (nk > 1) load_tile(1, ra1, rb1);
    store_tile(0, ra0, rb0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        // even phase: MFMAs on LDS buffer 0 (tile kt); set 0 is free -> fetch tile kt+2; then stage tile kt+1 (set 1)
        if (kt + 2 < nk) load_tile(kt + 2, ra0, rb0);
        compute(0);
        if (kt + 1 < nk) store_tile(1, ra1, rb1);
        __syncthreads();
        if (kt + 1 >= nk) break;
        // odd phase: MFMAs on buffer 1 (tile kt+1); fetch tile kt+3 into set 1; stage tile kt+2 (set 0)
        if (kt + 3 < nk) load_tile(kt + 3, ra1, rb1);
        compute(1);
        if (kt + 2 < nk) store_tile(0, ra0, rb0);
        __syncthreads();
    }

    const float *res = p.res ? p.res + (long)bz * p.sR : nullptr;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const int col = n0 + wn * 64 + tn * 32 + l32;
        if (col >= p.N) continue;
        const float sc = p.scale ? p.scale[col] : 1.f;
        const float bi = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row >= p.M) continue;
                float v = acc[tm][tn][r] * sc + bi;
                if (res && col < p.res_cols) v += res[(long)(p.res_rows ? row % p.res_rows : row) * p.ldr + col];
                if (p.relu) v = fmaxf(v, 0.f);
                C[(long)row * p.ldc + col] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Split-fp16 x3 with a scaled low part ("f16x3"): fp32-class accuracy (~3*2^-22) at the same MFMA count.
//   x = h + l * 2^-11,   h = fp16_rtz(x),   l = fp16_rtz((x - h) * 2^11)          (22+ significant bits, l never
//   A.B^T = [Ah.Bh^T] + 2^-11 * [Ah.Bl^T + Al.Bh^T]                                 underflows relative to h)
// The two brackets are accumulated in separate f32 accumulators (main / cross) and combined in the epilogue.
// Operands must satisfy |x| < 65504 (fp16 range); the S2D activations and weights are O(1e-3..1e3).
typedef __fp16 h16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split4_f16(const f32x4 v, u32x2 &hi, u32x2 &lo)
{
    const h16x2 ha = __builtin_amdgcn_cvt_pkrtz(v[0], v[1]), hb = __builtin_amdgcn_cvt_pkrtz(v[2], v[3]);
    const f32x2 a = {v[0], v[1]}, b = {v[2], v[3]};
    const f32x2 ra = (a - __builtin_convertvector(ha, f32x2)) * 2048.f, rb = (b - __builtin_convertvector(hb, f32x2)) * 2048.f;
    const h16x2 la = __builtin_amdgcn_cvt_pkrtz(ra[0], ra[1]), lb = __builtin_amdgcn_cvt_pkrtz(rb[0], rb[1]);
    hi[0] = __builtin_bit_cast(unsigned int, ha); hi[1] = __builtin_bit_cast(unsigned int, hb);
    lo[0] = __builtin_bit_cast(unsigned int, la); lo[1] = __builtin_bit_cast(unsigned int, lb);
}

template <bool CONV, bool PIPE, bool BSPLIT, bool DROP = false>
__global__ __launch_bounds__(256, 2) void gemm_f16x3_kernel(GemmParams p)
{
    constexpr int BM = 128, RPT = 32;
    extern __shared__ __attribute__((aligned(16))) unsigned int lds[];
    unsigned int *As = lds;                         // [2][BM][ROWW]
    unsigned int *Bs = lds + 2 * BM * ROWW;         // [2][BN][ROWW]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l32 = lane & 31, h = lane >> 5;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, within = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    }
    const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int bz = blockIdx.y;
    const float *A = p.A + (long)bz * p.sA;
    const float *B = p.B + (long)bz * p.sB;
    float *C = p.C + (long)bz * p.sC;

    const int c4 = tid & 7, g = tid >> 3;
    const int r0 = (g & ~7) | ((g & 1) << 2) | ((g >> 1) & 3);
    constexpr unsigned int OOB = 0xFFFFFFF0u;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A), 0, (int)p.bytesA, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = BSPLIT
        ? __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned int *>(p.Bsplit), 0, (int)((long)p.N * p.kblocks * 128L), 0x00020000)
        : __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(B), 0, (int)p.bytesB, 0x00020000);
    const int wsel = (c4 & 3) * 4 + (c4 >> 2) * 16;     // BSPLIT: this thread's 4 words of a pre-split row block
    unsigned int a_off[4], b_off[4], a_bad[4], b_bad[4];
    int a_iy0[4], a_ix0[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + r0 + RPT * i;
        const bool ok = m < p.M;
        if (CONV) {
            const int mm = ok ? m : 0;
            const int ox = mm % p.Wout, t = mm / p.Wout, oy = t % p.Hout, n = t / p.Hout;
            a_iy0[i] = ok ? oy * p.stride - p.pad : -(1 << 20);
            a_ix0[i] = ox * p.stride - p.pad;
            a_off[i] = (unsigned int)((long)n * p.Hin * p.Win * p.Cin * 4L);
        } else {
            a_off[i] = ok ? (unsigned int)((long)m * p.lda * 4L) : 0u;
            a_iy0[i] = a_ix0[i] = 0;
        }
        const int n = n0 + r0 + RPT * i;
        b_off[i] = n < p.N ? (unsigned int)(BSPLIT ? (long)n * p.kblocks * 128L + wsel * 4 : (long)n * p.ldb * 4L) : 0u;
        a_bad[i] = ok ? 0u : OOB;
        b_bad[i] = n < p.N ? 0u : OOB;
    }
    f32x4 ra[4], rb[4];
    auto load_tile = [&](int kt) {
        const int k = kt * BK + c4 * 4;
        const unsigned int kmask = (unsigned int)((p.K - 1 - k) >> 31) & OOB;   // all-OOB bits when k >= K (no select: keeps loads unconditional)
        int kh = 0, kw = 0, ci = 0;
        if (CONV) {
            const int tap = k / p.Cin;
            ci = k - tap * p.Cin;
            kh = tap / p.KW;
            kw = tap - kh * p.KW;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned int off;
            if (CONV) {
                const int iy = a_iy0[i] + kh, ix = a_ix0[i] + kw;
                const unsigned int tmask = (unsigned int)(((iy | ix | (p.Hin - 1 - iy) | (p.Win - 1 - ix)) >> 31)) & OOB;   // tap outside the image
                off = (a_off[i] + (unsigned int)(((iy * p.Win + ix) * p.Cin + ci) * 4)) | tmask | kmask;
            } else {
                off = (a_off[i] + (unsigned int)(k * 4)) | a_bad[i] | kmask;
            }
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)off, 0, 0));
            rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                rsB, (int)((b_off[i] + (unsigned int)(BSPLIT ? kt * 128 : k * 4)) | b_bad[i] | (BSPLIT ? 0u : kmask)), 0, 0));
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u32x2 hi, lo;
            split4_f16(ra[i], hi, lo);
            unsigned int *row = &As[(buf * BM + r0 + RPT * i) * ROWW];
            *reinterpret_cast<u32x2 *>(row + c4 * 2) = hi;
            *reinterpret_cast<u32x2 *>(row + 16 + c4 * 2) = lo;
            row = &Bs[(buf * BN + r0 + RPT * i) * ROWW];
            if (BSPLIT) { *reinterpret_cast<f32x4 *>(row + wsel) = rb[i]; continue; }
            split4_f16(rb[i], hi, lo);
            *reinterpret_cast<u32x2 *>(row + c4 * 2) = hi;
            *reinterpret_cast<u32x2 *>(row + 16 + c4 * 2) = lo;
        }
    };
    f32x16 accm[2][2], accx[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { accm[i][j][r] = 0.f; accx[i][j][r] = 0.f; }

    const int nk = (p.K + BK - 1) / BK;
    if constexpr (PIPE) {
        // Software pipeline inside one wave (no branches in the loop body, so it is one scheduling region):
        // the 24 MFMAs of k-tile kt are interleaved with the fp16 split + LDS stores of k-tile kt+1 (whose global loads
        // were issued one iteration earlier, group by group, into the same registers) and with the loads of k-tile
        // kt+2; operand fragments are read one group ahead.  Per MFMA (32 cycles of the matrix pipe) the wave issues
        // ~5 VALU + LDS/VMEM instructions in its shadow instead of running them as a separate phase after the MFMAs.
        // k-tile being loaded: this thread's k, its (kh, kw, ci) decode for the implicit GEMM and the past-K mask.  The
        // decode advances incrementally (one wrap per 32-wide step when Cin >= 32), so the loop body has no division.
        int lk = c4 * 4, lkh = 0, lkw = 0, lci = 0;
        unsigned int lkmask = 0u;
        auto decode_set = [&](int kt) {
            lk = kt * BK + c4 * 4;
            lkmask = (unsigned int)((p.K - 1 - lk) >> 31) & OOB;
            if (CONV) {
                const int tap = lk / p.Cin;
                lci = lk - tap * p.Cin;
                lkh = tap / p.KW;
                lkw = tap - lkh * p.KW;
            }
        };
        auto decode_next = [&]() {
            lk += BK;
            lkmask = (unsigned int)((p.K - 1 - lk) >> 31) & OOB;
            if (CONV) {
                lci += BK;
                const int wrap = lci >= p.Cin ? 1 : 0;
                lci -= wrap ? p.Cin : 0;
                lkw += wrap;
                const int w2 = lkw == p.KW ? 1 : 0;
                lkw = w2 ? 0 : lkw;
                lkh += w2;
            }
        };
        auto load_slot = [&](int i) {       // row slot i of the k-tile described by (lk, lkh, lkw, lci); zeros past K
            unsigned int off;
            if (CONV) {
                const int iy = a_iy0[i] + lkh, ix = a_ix0[i] + lkw;
                const unsigned int tmask = (unsigned int)(((iy | ix | (p.Hin - 1 - iy) | (p.Win - 1 - ix)) >> 31)) & OOB;
                off = (a_off[i] + (unsigned int)(((iy * p.Win + ix) * p.Cin + lci) * 4)) | tmask | lkmask;
            } else {
                off = (a_off[i] + (unsigned int)(lk * 4)) | a_bad[i] | lkmask;
            }
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)off, 0, 0));
            rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                rsB, (int)((b_off[i] + (unsigned int)(BSPLIT ? (lk >> 5) * 128 : lk * 4)) | b_bad[i] | (BSPLIT ? ((unsigned int)((p.kblocks * 32 - 1 - lk) >> 31) & OOB) : lkmask)), 0, 0));
        };
        auto store_slot = [&](int buf, int i) {
            u32x2 hi, lo;
            split4_f16(ra[i], hi, lo);
            unsigned int *row = &As[(buf * BM + r0 + RPT * i) * ROWW];
            *reinterpret_cast<u32x2 *>(row + c4 * 2) = hi;
            *reinterpret_cast<u32x2 *>(row + 16 + c4 * 2) = lo;
            row = &Bs[(buf * BN + r0 + RPT * i) * ROWW];
            if (BSPLIT) { *reinterpret_cast<f32x4 *>(row + wsel) = rb[i]; return; }
            split4_f16(rb[i], hi, lo);
            *reinterpret_cast<u32x2 *>(row + c4 * 2) = hi;
            *reinterpret_cast<u32x2 *>(row + 16 + c4 * 2) = lo;
        };
        const bool incr = !CONV || p.Cin >= BK;      // one wrap per step at most (the Cin = 4 stem decodes by division)
        decode_set(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) load_slot(i);
#pragma unroll
        for (int i = 0; i < 4; ++i) store_slot(0, i);
        decode_set(1);
#pragma unroll
        for (int i = 0; i < 4; ++i) load_slot(i);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (incr) decode_next(); else decode_set(kt + 2);
            const unsigned int *as = &As[(cur * BM + wm * 64 + l32) * ROWW + 4 * h];
            const unsigned int *bs = &Bs[(cur * BN + wn * 64 + l32) * ROWW + 4 * h];
            f16x8 bh[2][2], bl[2][2], ah[2], al[2];      // [s & 1] / [group parity]: fragments one group ahead
            auto read_b = [&](int s) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    bh[s][t] = *reinterpret_cast<const f16x8 *>(bs + t * 32 * ROWW + 8 * s);
                    bl[s][t] = *reinterpret_cast<const f16x8 *>(bs + t * 32 * ROWW + 16 + 8 * s);
                }
            };
            auto read_a = [&](int s, int i, int slot) {
                ah[slot] = *reinterpret_cast<const f16x8 *>(as + i * 32 * ROWW + 8 * s);
                al[slot] = *reinterpret_cast<const f16x8 *>(as + i * 32 * ROWW + 16 + 8 * s);
            };
            read_b(0); read_a(0, 0, 0);
            // One chunk = 1 MFMA + ~5 VALU (a third of one float4's fp16 split) [+ its LDS stores / the slot's loads];
            // sched_barrier(0) pins the chunk order, so every MFMA has independent work issuing in its shadow.
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int s = g >> 1, i = g & 1, fs = g & 1;
                h16x2 ha0, ha1, hb0, hb1;
                f32x2 fa0, fa1, fb0, fb1;
                unsigned int *rowA = &As[((cur ^ 1) * BM + r0 + RPT * g) * ROWW];
                unsigned int *rowB = &Bs[((cur ^ 1) * BN + r0 + RPT * g) * ROWW];
                const f32x4 va = ra[g], vb = rb[g];
                // chunk 0
                accx[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[fs], bh[s][0], accx[i][0], 0, 0, 0);
                if (g == 0) read_a(0, 1, 1);
                if (g == 1) { read_b(1); read_a(1, 0, 0); }
                if (g == 2) read_a(1, 1, 1);
                ha0 = __builtin_amdgcn_cvt_pkrtz(va[0], va[1]); ha1 = __builtin_amdgcn_cvt_pkrtz(va[2], va[3]);
                __builtin_amdgcn_sched_barrier(0);
                // chunk 1   (dependent MFMAs on one accumulator are kept two chunks apart)
                accx[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[fs], bh[s][1], accx[i][1], 0, 0, 0);
                fa0 = __builtin_convertvector(ha0, f32x2); fa1 = __builtin_convertvector(ha1, f32x2);
                __builtin_amdgcn_sched_barrier(0);
                // chunk 2
                accm[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[fs], bh[s][0], accm[i][0], 0, 0, 0);
                {
                    const f32x2 a0 = {va[0], va[1]}, a1 = {va[2], va[3]};
                    const f32x2 r0_ = (a0 - fa0) * 2048.f, r1_ = (a1 - fa1) * 2048.f;
                    const h16x2 l0 = __builtin_amdgcn_cvt_pkrtz(r0_[0], r0_[1]), l1 = __builtin_amdgcn_cvt_pkrtz(r1_[0], r1_[1]);
                    const u32x2 hi = {__builtin_bit_cast(unsigned int, ha0), __builtin_bit_cast(unsigned int, ha1)};
                    const u32x2 lo = {__builtin_bit_cast(unsigned int, l0), __builtin_bit_cast(unsigned int, l1)};
                    *reinterpret_cast<u32x2 *>(rowA + c4 * 2) = hi;
                    *reinterpret_cast<u32x2 *>(rowA + 16 + c4 * 2) = lo;
                }
                __builtin_amdgcn_sched_barrier(0);
                // chunk 3
                accx[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[fs], bl[s][0], accx[i][0], 0, 0, 0);
                if (BSPLIT) *reinterpret_cast<f32x4 *>(rowB + wsel) = vb;
                else {
                    hb0 = __builtin_amdgcn_cvt_pkrtz(vb[0], vb[1]); hb1 = __builtin_amdgcn_cvt_pkrtz(vb[2], vb[3]);
                    fb0 = __builtin_convertvector(hb0, f32x2); fb1 = __builtin_convertvector(hb1, f32x2);
                }
                __builtin_amdgcn_sched_barrier(0);
                // chunk 4
                accx[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[fs], bl[s][1], accx[i][1], 0, 0, 0);
                if (!BSPLIT) {
                    const f32x2 b0 = {vb[0], vb[1]}, b1 = {vb[2], vb[3]};
                    const f32x2 r0_ = (b0 - fb0) * 2048.f, r1_ = (b1 - fb1) * 2048.f;
                    const h16x2 l0 = __builtin_amdgcn_cvt_pkrtz(r0_[0], r0_[1]), l1 = __builtin_amdgcn_cvt_pkrtz(r1_[0], r1_[1]);
                    const u32x2 hi = {__builtin_bit_cast(unsigned int, hb0), __builtin_bit_cast(unsigned int, hb1)};
                    const u32x2 lo = {__builtin_bit_cast(unsigned int, l0), __builtin_bit_cast(unsigned int, l1)};
                    *reinterpret_cast<u32x2 *>(rowB + c4 * 2) = hi;
                    *reinterpret_cast<u32x2 *>(rowB + 16 + c4 * 2) = lo;
                }
                __builtin_amdgcn_sched_barrier(0);
                // chunk 5
                accm[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[fs], bh[s][1], accm[i][1], 0, 0, 0);
                load_slot(g);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        }
    } else {
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const unsigned int *as = &As[(cur * BM + wm * 64 + l32) * ROWW + 4 * h];
        const unsigned int *bs = &Bs[(cur * BN + wn * 64 + l32) * ROWW + 4 * h];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f16x8 bh[2], bl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                bh[t] = *reinterpret_cast<const f16x8 *>(bs + t * 32 * ROWW + 8 * s);
                bl[t] = *reinterpret_cast<const f16x8 *>(bs + t * 32 * ROWW + 16 + 8 * s);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f16x8 ah = *reinterpret_cast<const f16x8 *>(as + i * 32 * ROWW + 8 * s);
                const f16x8 al = *reinterpret_cast<const f16x8 *>(as + i * 32 * ROWW + 16 + 8 * s);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[j], accx[i][j], 0, 0, 0);
                    accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[j], accx[i][j], 0, 0, 0);
                    accm[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[j], accm[i][j], 0, 0, 0);
                }
            }
        }
        if (kt + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
    }
    }
    const float *res = p.res ? p.res + (long)bz * p.sR : nullptr;
    if (((p.N | p.ldc | p.ldr | p.res_cols) & 3) == 0) {
        // Vector epilogue: every wave parks its 64x64 tile in LDS (the operand ring is dead now) and streams it out
        // row-wise, 16 B per lane: 4 rows x 256 B per wave instruction instead of 64 scalar stores per lane, with
        // bias / scale read once and the residual read as float4.
        __syncthreads();
        float *ep = reinterpret_cast<float *>(lds) + wave * 64 * 68;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ep[(tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 68 + tn * 32 + l32] =
                        accm[tm][tn][r] + accx[tm][tn][r] * (1.0f / 2048.0f);
        if constexpr (DROP) {
            // dropout epilogue: a lane owns one 8-column mask block of a row (one Philox call), 8 lanes x 32 B = a 256-B row
            // segment per instruction, 8 rows per pass.  act(drop(acc * scale + bias) + res): the mask multiplies by 0 or 1 / (1 - p).
            const int c8 = lane & 7, rr8 = lane >> 3;
            const int col = n0 + wn * 64 + c8 * 8;
            if (col < p.N) {
                f32x4 sc0 = {1.f, 1.f, 1.f, 1.f}, sc1 = sc0, bi0 = {0.f, 0.f, 0.f, 0.f}, bi1 = bi0;
                if (p.scale) { sc0 = *reinterpret_cast<const f32x4 *>(p.scale + col); sc1 = *reinterpret_cast<const f32x4 *>(p.scale + col + 4); }
                if (p.bias) { bi0 = *reinterpret_cast<const f32x4 *>(p.bias + col); bi1 = *reinterpret_cast<const f32x4 *>(p.bias + col + 4); }
#pragma unroll 2
                for (int it = 0; it < 8; ++it) {
                    const int row = m0 + wm * 64 + it * 8 + rr8;
                    if (row >= p.M) break;
                    f32x4 v0 = *reinterpret_cast<const f32x4 *>(&ep[(it * 8 + rr8) * 68 + c8 * 8]);
                    f32x4 v1 = *reinterpret_cast<const f32x4 *>(&ep[(it * 8 + rr8) * 68 + c8 * 8 + 4]);
                    v0 = v0 * sc0 + bi0; v1 = v1 * sc1 + bi1;
                    float m[8];
                    s2d_dropout8((uint32_t)row + p.drop_row0, (uint32_t)(col >> 3), p.drop_stream, p.drop_k0, p.drop_k1, p.drop_thresh, p.drop_scale, m);
                    v0[0] *= m[0]; v0[1] *= m[1]; v0[2] *= m[2]; v0[3] *= m[3];
                    v1[0] *= m[4]; v1[1] *= m[5]; v1[2] *= m[6]; v1[3] *= m[7];
                    if (res) {
                        const float *rp = res + (long)(p.res_rows ? row % p.res_rows : row) * p.ldr + col;
                        if (col < p.res_cols) v0 += *reinterpret_cast<const f32x4 *>(rp);
                        if (col + 4 < p.res_cols) v1 += *reinterpret_cast<const f32x4 *>(rp + 4);
                    }
                    if (p.relu) {
                        v0[0] = fmaxf(v0[0], 0.f); v0[1] = fmaxf(v0[1], 0.f); v0[2] = fmaxf(v0[2], 0.f); v0[3] = fmaxf(v0[3], 0.f);
                        v1[0] = fmaxf(v1[0], 0.f); v1[1] = fmaxf(v1[1], 0.f); v1[2] = fmaxf(v1[2], 0.f); v1[3] = fmaxf(v1[3], 0.f);
                    }
                    *reinterpret_cast<f32x4 *>(C + (long)row * p.ldc + col) = v0;
                    *reinterpret_cast<f32x4 *>(C + (long)row * p.ldc + col + 4) = v1;
                }
            }
            return;
        }
        const int c4 = lane & 15, rr = lane >> 4;
        const int col = n0 + wn * 64 + c4 * 4;
        if (col < p.N) {
            f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
            if (p.scale) sc = *reinterpret_cast<const f32x4 *>(p.scale + col);
            if (p.bias) bi = *reinterpret_cast<const f32x4 *>(p.bias + col);
            const int rbase = m0 + wm * 64 + rr;
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
                const int row = rbase + it * 4;
                if (row >= p.M) break;
                f32x4 v = *reinterpret_cast<const f32x4 *>(&ep[(it * 4 + rr) * 68 + c4 * 4]);
                v = v * sc + bi;
                if (res && col < p.res_cols) v += *reinterpret_cast<const f32x4 *>(res + (long)(p.res_rows ? row % p.res_rows : row) * p.ldr + col);
                if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                *reinterpret_cast<f32x4 *>(C + (long)row * p.ldc + col) = v;
            }
        }
        return;
    }
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const int col = n0 + wn * 64 + tn * 32 + l32;
        if (col >= p.N) continue;
        const float sc = p.scale ? p.scale[col] : 1.f;
        const float bi = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row >= p.M) continue;
                float v = (accm[tm][tn][r] + accx[tm][tn][r] * (1.0f / 2048.0f)) * sc + bi;
                if (res && col < p.res_cols) v += res[(long)(p.res_rows ? row % p.res_rows : row) * p.ldr + col];
                if (p.relu) v = fmaxf(v, 0.f);
                C[(long)row * p.ldc + col] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// High-occupancy split-fp16 x3 variant: 128 x 64 x 32 tile, 4 waves of 64 x 32 (two f32 accumulator sets = 64
// registers), single LDS buffer (27.6 KB operands, 36.9 KB with the epilogue staging).  ~4 workgroups = 16 waves per
// CU: while one workgroup splits / stores / waits at its barriers, three others keep the matrix pipe busy (the
// 128 x 128 kernel above is limited to 2 waves/SIMD by its 128 accumulator registers and leaves the pipe ~2/3 idle).
template <bool CONV, bool BSPLIT>
__global__ __launch_bounds__(256, 4) void gemm_f16x3_hi_kernel(GemmParams p)
{
    constexpr int BM = 128, BNs = 64, RPT = 32;
    extern __shared__ __attribute__((aligned(16))) unsigned int lds[];
    unsigned int *As = lds;                     // [BM][ROWW]
    unsigned int *Bs = lds + BM * ROWW;         // [BNs][ROWW]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l32 = lane & 31, h = lane >> 5;
    const int tiles_n = (p.N + BNs - 1) / BNs, tiles_m = (p.M + BM - 1) / BM;
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, within = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    }
    const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BNs;
    const int bz = blockIdx.y;
    const float *A = p.A + (long)bz * p.sA;
    const float *B = p.B + (long)bz * p.sB;
    float *C = p.C + (long)bz * p.sC;

    const int c4 = tid & 7, g = tid >> 3;
    const int r0 = (g & ~7) | ((g & 1) << 2) | ((g >> 1) & 3);
    constexpr unsigned int OOB = 0xFFFFFFF0u;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A), 0, (int)p.bytesA, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = BSPLIT
        ? __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned int *>(p.Bsplit), 0, (int)((long)p.N * p.kblocks * 128L), 0x00020000)
        : __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(B), 0, (int)p.bytesB, 0x00020000);
    const int wsel = (c4 & 3) * 4 + (c4 >> 2) * 16;     // BSPLIT: this thread's 4 words of a pre-split row block
    unsigned int a_off[4], a_bad[4], b_off[2], b_bad[2];
    int a_iy0[4], a_ix0[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + r0 + RPT * i;
        const bool ok = m < p.M;
        if (CONV) {
            const int mm = ok ? m : 0;
            const int ox = mm % p.Wout, t = mm / p.Wout, oy = t % p.Hout, n = t / p.Hout;
            a_iy0[i] = ok ? oy * p.stride - p.pad : -(1 << 20);
            a_ix0[i] = ox * p.stride - p.pad;
            a_off[i] = (unsigned int)((long)n * p.Hin * p.Win * p.Cin * 4L);
        } else {
            a_off[i] = ok ? (unsigned int)((long)m * p.lda * 4L) : 0u;
            a_iy0[i] = a_ix0[i] = 0;
        }
        a_bad[i] = ok ? 0u : OOB;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int n = n0 + r0 + RPT * i;
        b_off[i] = n < p.N ? (unsigned int)(BSPLIT ? (long)n * p.kblocks * 128L + wsel * 4 : (long)n * p.ldb * 4L) : 0u;
        b_bad[i] = n < p.N ? 0u : OOB;
    }
    f32x4 ra[4], rb[2];
    auto load_tile = [&](int kt) {
        const int k = kt * BK + c4 * 4;
        const unsigned int kmask = (unsigned int)((p.K - 1 - k) >> 31) & OOB;
        int kh = 0, kw = 0, ci = 0;
        if (CONV) {
            const int tap = k / p.Cin;
            ci = k - tap * p.Cin;
            kh = tap / p.KW;
            kw = tap - kh * p.KW;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned int off;
            if (CONV) {
                const int iy = a_iy0[i] + kh, ix = a_ix0[i] + kw;
                const unsigned int tmask = (unsigned int)(((iy | ix | (p.Hin - 1 - iy) | (p.Win - 1 - ix)) >> 31)) & OOB;
                off = (a_off[i] + (unsigned int)(((iy * p.Win + ix) * p.Cin + ci) * 4)) | tmask | kmask;
            } else {
                off = (a_off[i] + (unsigned int)(k * 4)) | a_bad[i] | kmask;
            }
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)off, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
            rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                rsB, (int)((b_off[i] + (unsigned int)(BSPLIT ? kt * 128 : k * 4)) | b_bad[i] | (BSPLIT ? 0u : kmask)), 0, 0));
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u32x2 hi, lo;
            split4_f16(ra[i], hi, lo);
            unsigned int *row = &As[(r0 + RPT * i) * ROWW];
            *reinterpret_cast<u32x2 *>(row + c4 * 2) = hi;
            *reinterpret_cast<u32x2 *>(row + 16 + c4 * 2) = lo;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            u32x2 hi, lo;
            unsigned int *row = &Bs[(r0 + RPT * i) * ROWW];
            if (BSPLIT) { *reinterpret_cast<f32x4 *>(row + wsel) = rb[i]; continue; }
            split4_f16(rb[i], hi, lo);
            *reinterpret_cast<u32x2 *>(row + c4 * 2) = hi;
            *reinterpret_cast<u32x2 *>(row + 16 + c4 * 2) = lo;
        }
    };
    f32x16 accm[2], accx[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { accm[i][r] = 0.f; accx[i][r] = 0.f; }

    const int nk = (p.K + BK - 1) / BK;
    load_tile(0);
    for (int kt = 0; kt < nk; ++kt) {
        store_tile();                                  // tile kt: registers -> (split) -> LDS
        __syncthreads();
        if (kt + 1 < nk) load_tile(kt + 1);            // in flight during the MFMAs
        const unsigned int *as = &As[(wm * 64 + l32) * ROWW + 4 * h];
        const unsigned int *bs = &Bs[(wn * 32 + l32) * ROWW + 4 * h];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const f16x8 bh = *reinterpret_cast<const f16x8 *>(bs + 8 * s);
            const f16x8 bl = *reinterpret_cast<const f16x8 *>(bs + 16 + 8 * s);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f16x8 ah = *reinterpret_cast<const f16x8 *>(as + i * 32 * ROWW + 8 * s);
                const f16x8 al = *reinterpret_cast<const f16x8 *>(as + i * 32 * ROWW + 16 + 8 * s);
                accx[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accx[i], 0, 0, 0);
                accx[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, accx[i], 0, 0, 0);
                accm[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, accm[i], 0, 0, 0);
            }
        }
        __syncthreads();                               // everyone done reading before the next store_tile
    }
    const float *res = p.res ? p.res + (long)bz * p.sR : nullptr;
    if (((p.N | p.ldc | p.ldr | p.res_cols) & 3) == 0) {
        float *ep = reinterpret_cast<float *>(lds) + wave * 64 * 36;     // 64 x 32 tile, row stride 36
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                ep[(tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 36 + l32] = accm[tm][r] + accx[tm][r] * (1.0f / 2048.0f);
        const int c4e = lane & 7, rr = lane >> 3;                         // 8 lanes x 16 B = one 128-B row segment
        const int col = n0 + wn * 32 + c4e * 4;
        if (col < p.N) {
            f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
            if (p.scale) sc = *reinterpret_cast<const f32x4 *>(p.scale + col);
            if (p.bias) bi = *reinterpret_cast<const f32x4 *>(p.bias + col);
            const int rbase = m0 + wm * 64 + rr;
#pragma unroll 4
            for (int it = 0; it < 8; ++it) {
                const int row = rbase + it * 8;
                if (row >= p.M) break;
                f32x4 v = *reinterpret_cast<const f32x4 *>(&ep[(it * 8 + rr) * 36 + c4e * 4]);
                v = v * sc + bi;
                if (res && col < p.res_cols) v += *reinterpret_cast<const f32x4 *>(res + (long)(p.res_rows ? row % p.res_rows : row) * p.ldr + col);
                if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                *reinterpret_cast<f32x4 *>(C + (long)row * p.ldc + col) = v;
            }
        }
        return;
    }
    const int col = n0 + wn * 32 + l32;
    if (col >= p.N) return;
    const float sc = p.scale ? p.scale[col] : 1.f;
    const float bi = p.bias ? p.bias[col] : 0.f;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row >= p.M) continue;
            float v = (accm[tm][r] + accx[tm][r] * (1.0f / 2048.0f)) * sc + bi;
            if (res && col < p.res_cols) v += res[(long)(p.res_rows ? row % p.res_rows : row) * p.ldr + col];
            if (p.relu) v = fmaxf(v, 0.f);
            C[(long)row * p.ldc + col] = v;
        }
}

template <bool CONV>
int launch_f16_hi(const GemmParams &p, int batch, hipStream_t st)
{
    const size_t lds = sizeof(float) * 4 * 64 * 36;   // 36.9 KB: epilogue staging >= operand tiles (27.6 KB)
    const int nwg = cdiv(p.M, 128) * cdiv(p.N, 64);
    if (p.Bsplit) hipLaunchKernelGGL((gemm_f16x3_hi_kernel<CONV, true>), dim3(nwg, batch), dim3(256), lds, st, p);
    else hipLaunchKernelGGL((gemm_f16x3_hi_kernel<CONV, false>), dim3(nwg, batch), dim3(256), lds, st, p);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

template <bool CONV, bool PIPE, bool BSPLIT, bool DROP = false>
int launch_f16_v(const GemmParams &p, int batch, hipStream_t st)
{
    const size_t lds = sizeof(unsigned int) * 2 * (128 + BN) * ROWW;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_f16x3_kernel<CONV, PIPE, BSPLIT, DROP>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return S2D_ERR_LAUNCH;
        attr_set = true;
    }
    const int nwg = cdiv(p.M, 128) * cdiv(p.N, BN);
    hipLaunchKernelGGL((gemm_f16x3_kernel<CONV, PIPE, BSPLIT, DROP>), dim3(nwg, batch), dim3(256), lds, st, p);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

template <bool CONV, bool PIPE>
int launch_f16(const GemmParams &p, int batch, hipStream_t st)
{
    return p.Bsplit ? launch_f16_v<CONV, PIPE, true>(p, batch, st) : launch_f16_v<CONV, PIPE, false>(p, batch, st);
}

template <int WM, bool CONV>
int launch_t(const GemmParams &p, int batch, hipStream_t st)
{
    constexpr int BM = 64 * WM;
    const size_t lds = sizeof(unsigned int) * 2 * (BM + BN) * ROWW;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_bf16x3_kernel<WM, CONV>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return S2D_ERR_LAUNCH;
        attr_set = true;
    }
    const int nwg = cdiv(p.M, BM) * cdiv(p.N, BN);
    hipLaunchKernelGGL((gemm_bf16x3_kernel<WM, CONV>), dim3(nwg, batch), dim3(128 * WM), lds, st, p);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

// static weights -> [N][kblocks][16 words hi | 16 words lo]: the LDS row image of the split-fp16 kernels, zero padded past K
__global__ __launch_bounds__(256) void split_weights_kernel(const float *__restrict__ W, int N, int K, long ldw, int kblocks,
                                                            unsigned int *__restrict__ out)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)N * kblocks * 8) return;
    const int c4 = (int)(i & 7);
    const long nb = i >> 3;
    const int kb = (int)(nb % kblocks);
    const long n = nb / kblocks;
    const int k = kb * 32 + c4 * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (k + j < K) v[j] = W[n * ldw + k + j];
    u32x2 hi, lo;
    split4_f16(v, hi, lo);
    unsigned int *o = out + nb * 32;
    *reinterpret_cast<u32x2 *>(o + c4 * 2) = hi;
    *reinterpret_cast<u32x2 *>(o + 16 + c4 * 2) = lo;
}

}  // namespace

int s2d_split_weights_launch(const float *W, int N, int K, long ldw, unsigned int *out, hipStream_t st)
{
    const int kblocks = (K + 31) / 32;
    const long n = (long)N * kblocks * 8;
    if (n == 0) return S2D_OK;
    hipLaunchKernelGGL(split_weights_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, W, N, K, ldw, kblocks, out);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

int s2d_launch_gemm_bf16x3(const GemmParams &pin, bool conv, int batch, hipStream_t st, int f16)
{
    GemmParams p = pin;
    const long bA = conv ? (long)(p.M / ((long)p.Hout * p.Wout)) * p.Hin * p.Win * p.Cin * 4L : ((long)(p.M - 1) * p.lda + p.K) * 4L;
    const long bB = ((long)(p.N - 1) * p.ldb + p.K) * 4L;
    if (bA > 0xFFFFFF00L || bB > 0xFFFFFF00L) return S2D_ERR_ARG;   // 32-bit buffer offsets
    p.bytesA = (unsigned int)bA; p.bytesB = (unsigned int)bB;
    if (p.Bsplit) {
        if (!f16 || (batch > 1 && p.sB != 0) || (long)p.N * ((p.K + 31) / 32) * 128L > 0xFFFFFF00L) return S2D_ERR_ARG;
        p.kblocks = (p.K + 31) / 32;
    }
    if (p.drop_thresh) {
        // fused dropout lives in the vector epilogue of the pipelined 128x128 split-fp16 kernel (the three encoder-layer
        // GEMMs that carry it all dispatch there): 8-column mask blocks, 16-B aligned rows
        if (!f16 || conv || ((p.N | p.ldc) & 7) || (p.res && (((p.ldr | p.res_cols) & 7)))) return S2D_ERR_ARG;
        return p.Bsplit ? launch_f16_v<false, true, true, true>(p, batch, st) : launch_f16_v<false, true, false, true>(p, batch, st);
    }
    // 256-row tiles only when they still fill the chip
    if (f16) {
        // measured (scripts/mb_shapes.py, mb_pipe.py): the pipelined 128x128 kernel wins on the spatial convolutions and, once
        // the weights arrive pre-split, on the GEMMs with K >= 512 (+9..23 %) and on K = 256 when 128-wide tiles waste no
        // more columns than 64-wide ones; the 16-waves/CU 128x64 kernel keeps the short-K (<= 128: two to four k-steps,
        // all prologue/epilogue), narrow (N <= 64) and dynamic-B launches
        static int hi = -1;
        if (hi < 0) { const char *e = getenv("S2D_GEMM_HI"); hi = e ? atoi(e) : 2; }
        bool use_hi;
        if (hi != 2) use_hi = hi == 1;
        else if (conv && p.KH > 1) use_hi = p.N <= 64;     // N <= 64: half a 128-wide tile would idle
        else if (p.N <= 64 || p.K <= 128 || !p.Bsplit) use_hi = true;
        else if (p.K >= 512) use_hi = false;
        else {
            const float w128 = (float)(cdiv(p.N, 128) * 128) / p.N, w64 = (float)(cdiv(p.N, 64) * 64) / p.N;
            use_hi = w128 > w64 + 0.1f;
        }
        if (use_hi) return conv ? launch_f16_hi<true>(p, batch, st) : launch_f16_hi<false>(p, batch, st);
        static int pipe = -1;
        if (pipe < 0) { const char *e = getenv("S2D_GEMM_PIPE"); pipe = e ? atoi(e) : 1; }   // measured +6..12 %
        if (pipe) return conv ? launch_f16<true, true>(p, batch, st) : launch_f16<false, true>(p, batch, st);
        return conv ? launch_f16<true, false>(p, batch, st) : launch_f16<false, false>(p, batch, st);
    }
    static int force = -1;
    if (force < 0) { const char *e = getenv("S2D_GEMM_WM"); force = e ? atoi(e) : 0; }
    bool big = (long)cdiv(p.M, 256) * cdiv(p.N, BN) * batch >= 256;
    if (force == 2) big = false;
    if (force == 4) big = true;
    if (big) return conv ? launch_t<4, true>(p, batch, st) : launch_t<4, false>(p, batch, st);
    return conv ? launch_t<2, true>(p, batch, st) : launch_t<2, false>(p, batch, st);
}
