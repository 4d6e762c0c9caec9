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
#include <type_traits>

namespace {

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_t __attribute__((vector_size(16)));

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
// Split-bf16 x3 with 128 x 64 per wave ("w128"): bf16 needs no scaled low part, so the three products of a tile land in ONE
// accumulator set -- 128 accumulator registers now cover 128 x 64 per wave instead of 64 x 64.  Per k-step a wave reads 12
// fragments for 24 MFMAs (0.5 reads per MFMA against 0.67 with 64 x 64 tiles) and a 256 x 128 x 32 tile costs ~3 staging
// instructions per MFMA instead of ~5.  Workgroup = 4 waves (2 x 2), tile 256 x 128 x 32, ONE LDS operand buffer (55 KB;
// 69.6 KB with the epilogue staging) so that two workgroups share a CU: one's split / store / barrier phases run under the
// other's 48-MFMA blocks.  Operand loads for k-tile kt+1 are issued right after the barrier that publishes k-tile kt and
// land behind its MFMAs.
template <bool CONV, bool BSPLIT>
__global__ __launch_bounds__(256, 2) void gemm_bf16x3_w128_kernel(GemmParams p)
{
    constexpr int BM = 256, RPT = 32, A_IT = BM / RPT, B_IT = BN / RPT;      // 8 / 4 float4 per thread per k-tile
    extern __shared__ __attribute__((aligned(16))) unsigned int lds[];
    unsigned int *As = lds;                         // [BM][ROWW]
    unsigned int *Bs = lds + BM * ROWW;             // [BN][ROWW]
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
    unsigned int a_off[A_IT], a_bad[A_IT];
    int a_iy0[A_IT], a_ix0[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
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
    unsigned int b_off[B_IT], b_bad[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int n = n0 + r0 + RPT * i;
        b_off[i] = n < p.N ? (unsigned int)(BSPLIT ? (long)n * p.kblocks * 128L + wsel * 4 : (long)n * p.ldb * 4L) : 0u;
        b_bad[i] = n < p.N ? 0u : OOB;
    }
    // every wait on these loads is a full drain (vmcnt(0) before the split), so masked out-of-range offsets are safe here
    f32x4 ra[A_IT], rb[B_IT];
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
        for (int i = 0; i < A_IT; ++i) {
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
        for (int i = 0; i < B_IT; ++i)
            rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                rsB, (int)((b_off[i] + (unsigned int)(BSPLIT ? kt * 128 : k * 4)) | b_bad[i] | (BSPLIT ? 0u : kmask)), 0, 0));
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            u32x2 hi, lo;
            split4(ra[i], hi, lo);
            unsigned int *row = &As[(r0 + RPT * i) * ROWW];
            *reinterpret_cast<u32x2 *>(row + c4 * 2) = hi;
            *reinterpret_cast<u32x2 *>(row + 16 + c4 * 2) = lo;
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            u32x2 hi, lo;
            unsigned int *row = &Bs[(r0 + RPT * i) * ROWW];
            if (BSPLIT) { *reinterpret_cast<f32x4 *>(row + wsel) = rb[i]; continue; }
            split4(rb[i], hi, lo);
            *reinterpret_cast<u32x2 *>(row + c4 * 2) = hi;
            *reinterpret_cast<u32x2 *>(row + 16 + c4 * 2) = lo;
        }
    };
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const unsigned int *as = &As[(wm * 128 + l32) * ROWW + 4 * h];
    const unsigned int *bs = &Bs[(wn * 64 + l32) * ROWW + 4 * h];
    // Fragment reads run one row block ahead of the MFMAs that use them (two A fragment sets, the next k-step's B fragments
    // fetched during the last row block): left to itself the compiler reads every fragment into the same registers right in
    // front of its MFMAs and the matrix pipe waits out an LDS round trip per row block.
    auto compute = [&]() {
        bf16x8 bh[2][2], bl[2][2], ah[2], al[2];             // [k-step parity][column block] / [row-block parity]
        auto read_b = [&](int s) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                bh[s & 1][t] = *reinterpret_cast<const bf16x8 *>(bs + t * 32 * ROWW + 8 * s);
                bl[s & 1][t] = *reinterpret_cast<const bf16x8 *>(bs + t * 32 * ROWW + 16 + 8 * s);
            }
        };
        auto read_a = [&](int s, int i) {
            ah[i & 1] = *reinterpret_cast<const bf16x8 *>(as + i * 32 * ROWW + 8 * s);
            al[i & 1] = *reinterpret_cast<const bf16x8 *>(as + i * 32 * ROWW + 16 + 8 * s);
        };
        read_b(0); read_a(0, 0);
#pragma unroll
        for (int g = 0; g < 8; ++g) {                         // g = k-step * 4 + row block
            const int s = g >> 2, i = g & 3;
            if (g + 1 < 8) read_a((g + 1) >> 2, (g + 1) & 3);
            if (i == 3 && s == 0) read_b(1);
            __builtin_amdgcn_sched_barrier(0);
            // small terms first, the dominant hi.hi last; the two column blocks alternate so that dependent MFMAs are one apart
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i & 1], bh[s][0], acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i & 1], bh[s][1], acc[i][1], 0, 0, 0);
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i & 1], bl[s][0], acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i & 1], bl[s][1], acc[i][1], 0, 0, 0);
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i & 1], bh[s][0], acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i & 1], bh[s][1], acc[i][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const int nk = (p.K + BK - 1) / BK;
    load_tile(0);
    for (int kt = 0; kt < nk; ++kt) {
        store_tile();                                         // k-tile kt: split into the (free) operand buffer
        __syncthreads();
        if (kt + 1 < nk) load_tile(kt + 1);                   // lands behind the MFMAs below
        compute();
        __syncthreads();                                      // everyone is done reading before the next store
    }

    // epilogue: two halves of 64 rows per wave through LDS (the operand buffer is dead), 16-B row stores when the shapes allow
    const float *res = p.res ? p.res + (long)bz * p.sR : nullptr;
    const bool vec = ((p.N | p.ldc | p.ldr | p.res_cols) & 3) == 0;
    float *ep = reinterpret_cast<float *>(lds) + wave * 64 * 68;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (vec) {
            if (half) __syncthreads();                        // (uniform) the previous half has been streamed out
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        ep[(tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 68 + tn * 32 + l32] = acc[half * 2 + tm][tn][r];
            const int c4e = lane & 15, rr = lane >> 4;
            const int col = n0 + wn * 64 + c4e * 4;
            if (col < p.N) {
                f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
                if (p.scale) sc = *reinterpret_cast<const f32x4 *>(p.scale + col);
                if (p.bias) bi = *reinterpret_cast<const f32x4 *>(p.bias + col);
                const int rbase = m0 + wm * 128 + half * 64 + rr;
#pragma unroll 4
                for (int it = 0; it < 16; ++it) {
                    const int row = rbase + it * 4;
                    if (row >= p.M) break;
                    f32x4 v = *reinterpret_cast<const f32x4 *>(&ep[(it * 4 + rr) * 68 + c4e * 4]);
                    v = v * sc + bi;
                    if (res && col < p.res_cols) v += *reinterpret_cast<const f32x4 *>(res + (long)(p.res_rows ? row % p.res_rows : row) * p.ldr + col);
                    if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                    *reinterpret_cast<f32x4 *>(C + (long)row * p.ldc + col) = v;
                }
            }
        } else {
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
                        const int row = m0 + wm * 128 + half * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        if (row >= p.M) continue;
                        float v = acc[half * 2 + tm][tn][r] * sc + bi;
                        if (res && col < p.res_cols) v += res[(long)(p.res_rows ? row % p.res_rows : row) * p.ldr + col];
                        if (p.relu) v = fmaxf(v, 0.f);
                        C[(long)row * p.ldc + col] = v;
                    }
                }
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
                if (p.gate) {
                    const f32x4 g = *reinterpret_cast<const f32x4 *>(p.gate + (long)row * p.ldg + col);
                    v[0] = g[0] > 0.f ? v[0] * p.gate_scale : 0.f; v[1] = g[1] > 0.f ? v[1] * p.gate_scale : 0.f;
                    v[2] = g[2] > 0.f ? v[2] * p.gate_scale : 0.f; v[3] = g[3] > 0.f ? v[3] * p.gate_scale : 0.f;
                }
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
// 3 x 3 / stride 1 / pad 1 convolution with the input staged ONCE per channel block ("halo"): the implicit-GEMM kernel above
// re-loads and re-splits the 128 input rows of a tile for each of the nine taps (9 x 128 rows per 32 channels); here a
// workgroup owns an 8 x 16 patch of output pixels, stages the 10 x 18 input halo of a 32-channel block (180 rows) and all
// nine taps read their A fragments from it at shifted rows.  Per tap a thread then only copies its four 16-B pieces of the
// pre-split weight block: ~15 staging instructions per 24 MFMAs instead of ~100.  Same arithmetic per product; the k order
// is (channel block, tap) instead of (tap, channel block), so sums differ from the implicit-GEMM kernel in the last bits.
//   LDS: halo 180 rows (25.9 KB, single: replaced between channel blocks) + two weight buffers (36.9 KB); 69.6 KB with the
//   epilogue staging -> two workgroups per CU.
// PH x 16 output pixels and BNT = 64 WNW output channels per workgroup, a wave = 64 pixels x 64 channels either way:
//   <8, 2>: 8 x 16 patch x 128 channels (Cout > 64);  <16, 1>: 16 x 16 patch x 64 channels (Cout <= 64: the res2 bottlenecks'
//   3 x 3 convolutions, which the implicit-GEMM 128 x 64 kernel ran at 234 TFLOP/s with 9 x the input traffic through L2).
constexpr int HT_H = 8, HT_W = 16, HALO_W = HT_W + 2;
template <int PH, int WNW, bool PIPE>
__global__ __launch_bounds__(256, 2) void conv3x3_f16x3_halo_kernel(GemmParams p)
{
    constexpr int HALO_ROWS = (PH + 2) * HALO_W, BNT = 64 * WNW, WROWS = BNT / 32;      // WROWS: weight rows a thread copies per step
    static_assert(PH * HT_W == 64 * (4 / WNW), "four waves of 64 pixels x 64 channels");
    extern __shared__ __attribute__((aligned(16))) unsigned int lds[];
    unsigned int *Ah = lds;                              // [HALO_ROWS][ROWW]
    unsigned int *Bs = lds + HALO_ROWS * ROWW;           // [2][BNT][ROWW]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WNW, wn = wave % WNW;
    const int l32 = lane & 31, h = lane >> 5;
    const int H = p.Hin, W = p.Win, Cin = p.Cin;
    const int tiles_x = (W + HT_W - 1) / HT_W, tiles_y = (H + PH - 1) / PH;
    const int tiles_n = (p.N + BNT - 1) / BNT, tiles_m = (p.M / (H * W)) * tiles_y * tiles_x;
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, within = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    }
    const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
    const int n0 = tile_n * BNT;
    const int img = tile_m / (tiles_y * tiles_x), trem = tile_m % (tiles_y * tiles_x);
    const int y0 = (trem / tiles_x) * PH, x0 = (trem % tiles_x) * HT_W;

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.A), 0, (int)p.bytesA, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned int *>(p.Bsplit), 0, (int)((long)p.N * p.kblocks * 128L), 0x00020000);
    // every load is in range (see the wave-specialised kernel): pixels outside the image read offset 0 and are zeroed when the
    // halo is stored; weight rows past Cout read the last row (their columns are never stored)
    // halo staging: item = (halo row, 16-B piece); 1440 items over 256 threads
    constexpr int A_IT = (HALO_ROWS * 8 + 255) / 256;     // 6 (PH = 8), 11 (PH = 16)
    unsigned int ha_off[A_IT];
    int ha_dst[A_IT];
    bool ha_zero[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int item = tid + 256 * i;
        const int row = item >> 3, c4 = item & 7;
        const int hy = row / HALO_W, hx = row - hy * HALO_W;
        const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        const bool ok = row < HALO_ROWS && iy >= 0 && iy < H && ix >= 0 && ix < W;
        ha_off[i] = ok ? (unsigned int)((((long)img * H + iy) * W + ix) * Cin * 4L + c4 * 16) : 0u;
        ha_zero[i] = !ok;
        ha_dst[i] = row < HALO_ROWS ? row * ROWW + c4 * 2 : -1;
    }
    // weight staging: as the implicit-GEMM kernel (row r0 + 32 i of the tile, 16-B piece wsel of its pre-split block)
    const int c4 = tid & 7, g = tid >> 3;
    const int r0 = (g & ~7) | ((g & 1) << 2) | ((g >> 1) & 3);
    const int wsel = (c4 & 3) * 4 + (c4 >> 2) * 16;
    unsigned int b_off[WROWS];
#pragma unroll
    for (int i = 0; i < WROWS; ++i) {
        const int n = n0 + r0 + 32 * i;
        b_off[i] = (unsigned int)((long)(n < p.N ? n : p.N - 1) * p.kblocks * 128L + wsel * 4);
    }
    const int cblocks = Cin / 32;
    f32x4 ra[A_IT], rb[WROWS];
    auto load_halo = [&](int cb) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i)
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)(ha_off[i] + (ha_zero[i] ? 0u : (unsigned int)(cb * 128))), 0, 0));   // per-lane choice: in the vector offset (a divergent scalar offset compiles to a waterfall loop)
    };
    auto store_halo = [&]() {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            if (ha_dst[i] < 0) continue;
            u32x2 hi, lo;
            split4_f16(ra[i], hi, lo);
            if (ha_zero[i]) { hi = u32x2{0u, 0u}; lo = u32x2{0u, 0u}; }
            *reinterpret_cast<u32x2 *>(Ah + ha_dst[i]) = hi;
            *reinterpret_cast<u32x2 *>(Ah + ha_dst[i] + 16) = lo;
        }
    };
    auto load_w = [&](int tap, int cb) {                   // k block of (tap, channel block) in the [Cout][3][3][Cin] weights
        const unsigned int kb = (unsigned int)(tap * cblocks + cb) * 128u;
#pragma unroll
        for (int i = 0; i < WROWS; ++i)
            rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)b_off[i], (int)kb, 0));
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int i = 0; i < WROWS; ++i)
            *reinterpret_cast<f32x4 *>(Bs + (buf * BNT + r0 + 32 * i) * ROWW + wsel) = rb[i];
    };
    // fragment rows: output pixel (py, px) of the patch reads halo row (py + dy) * 18 + px + dx for tap (dy, dx)
    int a_row[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int pix = wm * 64 + i * 32 + l32;
        a_row[i] = ((pix / HT_W) * HALO_W + (pix % HT_W)) * ROWW + 4 * h;
    }
    f32x16 accm[2][2], accx[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { accm[i][j][r] = 0.f; accx[i][j][r] = 0.f; }
    auto compute = [&](int tap, int buf) {
        const int toff = ((tap / 3) * HALO_W + (tap % 3)) * ROWW;
        const unsigned int *bs = Bs + (buf * BNT + wn * 64 + l32) * ROWW + 4 * h;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f16x8 bh[2], bl[2];
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                bh[t2] = *reinterpret_cast<const f16x8 *>(bs + t2 * 32 * ROWW + 8 * s);
                bl[t2] = *reinterpret_cast<const f16x8 *>(bs + t2 * 32 * ROWW + 16 + 8 * s);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f16x8 ah = *reinterpret_cast<const f16x8 *>(Ah + a_row[i] + toff + 8 * s);
                const f16x8 al = *reinterpret_cast<const f16x8 *>(Ah + a_row[i] + toff + 16 + 8 * s);
                accx[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[0], accx[i][0], 0, 0, 0);
                accx[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[1], accx[i][1], 0, 0, 0);
                accx[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[0], accx[i][0], 0, 0, 0);
                accx[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[1], accx[i][1], 0, 0, 0);
                accm[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[0], accm[i][0], 0, 0, 0);
                accm[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[1], accm[i][1], 0, 0, 0);
            }
        }
    };
    // Round 5 (S2D_CONV_HALO_PIPE=1, opt-in: measured, no net gain): the same 24 MFMAs of a tap with their fragment reads ONE GROUP AHEAD.  In the form above every
    // tap opens with its reads and the first MFMAs wait for them one by one (the schedule of its ISA: r..rrrrr|M|Mrrrrrr.|M|MMMrr|MM..: four to
    // six exposed LDS round trips per 768 MFMA cycles).  Here a group = one (k-step, pixel tile) = 6 MFMAs; while it runs, the A pair of the next
    // group is read, the B fragments of the tap's second k-step during its first group, and the A pair of the NEXT TAP's first group during the
    // tap's last group (the halo does not change inside a channel block; a new block's halo is read behind its barrier).  Only the next tap's
    // four B fragments are read behind the tap's barrier (the weights become visible there).  Same products, same order: same bits.
    f16x8 pa[2][2];                     // [buffer][hi / lo] A pair of a group
    f16x8 pb[2][2][2];                  // [k-step][n tile][hi / lo]
    auto read_a = [&](int bufi, int tap, int s, int i) {
        const int toff = ((tap / 3) * HALO_W + (tap % 3)) * ROWW;
        pa[bufi][0] = *reinterpret_cast<const f16x8 *>(Ah + a_row[i] + toff + 8 * s);
        pa[bufi][1] = *reinterpret_cast<const f16x8 *>(Ah + a_row[i] + toff + 16 + 8 * s);
    };
    auto read_b = [&](int buf, int s) {
        const unsigned int *bs = Bs + (buf * BNT + wn * 64 + l32) * ROWW + 4 * h;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            pb[s][t2][0] = *reinterpret_cast<const f16x8 *>(bs + t2 * 32 * ROWW + 8 * s);
            pb[s][t2][1] = *reinterpret_cast<const f16x8 *>(bs + t2 * 32 * ROWW + 16 + 8 * s);
        }
    };
    auto group = [&](int bufi, int s, int i) {
        accx[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa[bufi][1], pb[s][0][0], accx[i][0], 0, 0, 0);
        accx[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa[bufi][1], pb[s][1][0], accx[i][1], 0, 0, 0);
        accx[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa[bufi][0], pb[s][0][1], accx[i][0], 0, 0, 0);
        accx[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa[bufi][0], pb[s][1][1], accx[i][1], 0, 0, 0);
        accm[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa[bufi][0], pb[s][0][0], accm[i][0], 0, 0, 0);
        accm[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa[bufi][0], pb[s][1][0], accm[i][1], 0, 0, 0);
    };
    // tap `tap` on weight buffer `buf`; its first A pair (buffer 0) and its k-step-0 B fragments are already in registers.  nexta: the next
    // tap reads the same halo (prefetch its first A pair during the last group)
    auto compute_pipe = [&](int tap, int buf, int ntap, bool nexta) {
        read_a(1, tap, 0, 1); read_b(buf, 1);
        __builtin_amdgcn_sched_barrier(0);
        group(0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_a(0, tap, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        group(1, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        read_a(1, tap, 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        group(0, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (nexta) read_a(0, ntap, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        group(1, 1, 1);
        __builtin_amdgcn_sched_barrier(0);
    };
    // (channel block, tap) stream: q = cb * 9 + tap.  Weights of step q+1 are loaded during step q and stored into the other
    // buffer before the barrier; the halo of block cb+1 is loaded during tap 6 of block cb and stored after the barrier that
    // ends tap 8 (when nobody reads the old halo any more), followed by one more barrier.
    constexpr bool pipe = PIPE;
    load_halo(0); load_w(0, 0);
    store_halo(); store_w(0);
    __syncthreads();
    const int Q = cblocks * 9;
    int tap = 0, cb = 0;
    if (pipe) { read_a(0, 0, 0, 0); read_b(0, 0); }
    for (int q = 0; q < Q; ++q) {
        const int ntap = tap == 8 ? 0 : tap + 1, ncb = tap == 8 ? cb + 1 : cb;
        if (q + 1 < Q) load_w(ntap, ncb);
        if (tap == 6 && cb + 1 < cblocks) load_halo(cb + 1);
        const bool newhalo = tap == 8 && cb + 1 < cblocks;
        if (pipe) compute_pipe(tap, q & 1, ntap, q + 1 < Q && !newhalo);
        else compute(tap, q & 1);
        if (q + 1 < Q) store_w((q + 1) & 1);
        __syncthreads();
        if (newhalo) { store_halo(); __syncthreads(); }
        if (pipe && q + 1 < Q) {
            if (newhalo) read_a(0, ntap, 0, 0);
            read_b((q + 1) & 1, 0);
        }
        tap = ntap; cb = ncb;
    }

    // epilogue (the 16-B row form; host-checked: Cout % 4 == 0): the wave's 64 pixels x 64 channels through LDS
    float *ep = reinterpret_cast<float *>(lds) + wave * 64 * 68;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                ep[(tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 68 + tn * 32 + l32] = accm[tm][tn][r] + accx[tm][tn][r] * (1.0f / 2048.0f);
    const int c4e = lane & 15, rr = lane >> 4;
    const int col = n0 + wn * 64 + c4e * 4;
    if (col < p.N) {
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
        if (p.scale) sc = *reinterpret_cast<const f32x4 *>(p.scale + col);
        if (p.bias) bi = *reinterpret_cast<const f32x4 *>(p.bias + col);
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int pix = wm * 64 + it * 4 + rr;
            const int oy = y0 + pix / HT_W, ox = x0 + pix % HT_W;
            if (oy >= H || ox >= W) continue;
            const long row = ((long)img * H + oy) * W + ox;
            f32x4 v = *reinterpret_cast<const f32x4 *>(&ep[(it * 4 + rr) * 68 + c4e * 4]);
            v = v * sc + bi;
            if (p.res) v += *reinterpret_cast<const f32x4 *>(p.res + row * p.ldr + col);
            if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
            *reinterpret_cast<f32x4 *>(p.C + row * p.ldc + col) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// The R50 stem: 7 x 7 / stride 2 / pad 3 on the 4-channel NHWC input (3 colours + a zero), 64 output channels.  As an implicit
// GEMM its A operand is 49 taps x 4 channels gathered 16 B at a time with a division per tap and a border test per load, for a
// K of 196 that is all staging (139 TFLOP/s).  Here a workgroup owns a 16 x 16 patch of output pixels, stages the 37 x 37 input
// halo ONCE (22 KB as fp16 hi / lo planes, pixels outside the image as zeros), and every tap pair of the MFMA's k range is two
// 8-B LDS reads at a constant offset from the lane's pixel; the weights keep the implicit GEMM's image ([64][7 k-blocks] of
// k = (kh 7 + kw) 4 + c, zero past 196) and stream through two LDS buffers.  Same products, same k order within a k-block.
// PH: patch height.  16: a wave = 64 pixels x 64 channels (128 accumulator registers, two workgroups per CU); 8: a wave = 32 pixels
// x 64 channels (64 accumulator registers, four workgroups per CU -- the kernel is a sequence halo load -> MFMAs -> store per
// workgroup, so more workgroups in flight is what it lacks).
constexpr int ST_P = 16, ST_HW = 2 * ST_P + 5;          // patch width, halo width (37)
template <int PH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PH == 16 ? 2 : 4, PH == 16 ? 2 : 4))) void conv7x7s2_c4_halo_kernel(GemmParams p)
{
    constexpr int ST_HH = 2 * PH + 5, ST_PLANE = ST_HH * ST_HW * 2, RT = PH / 8, WPIX = 32 * RT;   // halo height, words of one fp16 plane, row tiles and pixels per wave
    extern __shared__ __attribute__((aligned(16))) unsigned int lds[];
    unsigned int *Ah = lds;                              // [37][37][2 words]: 4 channels as fp16
    unsigned int *Al = lds + ST_PLANE;                   // the scaled low parts
    unsigned int *Bs = lds + 2 * ST_PLANE;               // [2][64][ROWW]   (whole 16-B units: rows stay 16-B aligned)
    static_assert((2 * ST_PLANE) % 4 == 0, "weight rows are copied 16 B at a time");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l32 = lane & 31, h = lane >> 5;
    const int H = p.Hin, W = p.Win, Ho = p.Hout, Wo = p.Wout;
    const int tiles_x = (Wo + ST_P - 1) / ST_P, tiles_y = (Ho + PH - 1) / PH;
    const int nwg = (p.M / (Ho * Wo)) * tiles_y * tiles_x;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, within = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    }
    const int img = bid / (tiles_y * tiles_x), trem = bid % (tiles_y * tiles_x);
    const int y0 = (trem / tiles_x) * PH, x0 = (trem % tiles_x) * ST_P;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.A), 0, (int)p.bytesA, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned int *>(p.Bsplit), 0, (int)((long)p.N * p.kblocks * 128L), 0x00020000);
    // halo: pixel (hy, hx) of the 37 x 37 window = input pixel (2 y0 - 3 + hy, 2 x0 - 3 + hx): one 16-B load, two 8-B LDS stores
    constexpr int A_IT = (ST_HH * ST_HW + 255) / 256;     // 6 / 4
    f32x4 ra[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int item = tid + 256 * i;
        const int hy = item / ST_HW, hx = item - hy * ST_HW;
        const int iy = 2 * y0 - 3 + hy, ix = 2 * x0 - 3 + hx;
        const bool ok = item < ST_HH * ST_HW && iy >= 0 && iy < H && ix >= 0 && ix < W;
        ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, ok ? (int)((((long)img * H + iy) * W + ix) * 16L) : (int)0x80000000, 0, 0));
    }
    // weights: row r0 + 32 i of the 64 channels, 16-B piece wsel of its k-block (as the other split-fp16 kernels copy them)
    const int c4 = tid & 7, g = tid >> 3;
    const int r0 = (g & ~7) | ((g & 1) << 2) | ((g >> 1) & 3);
    const int wsel = (c4 & 3) * 4 + (c4 >> 2) * 16;
    unsigned int b_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) b_off[i] = (unsigned int)((long)min(r0 + 32 * i, p.N - 1) * p.kblocks * 128L + wsel * 4);
    // all seven k-blocks of the weights are requested up front, with the halo (one memory round trip per workgroup), and go to LDS
    // one block ahead of their use.  (Measured equal to a one-block lookahead: at two 256-register workgroups per CU the kernel is
    // bound by the sequence halo load -> 156 MFMAs per wave -> 64 KB of output per workgroup, matrix pipe 31 % busy, LDS 30 %.)
    constexpr int WSETS = PH == 16 ? 7 : 2;               // PH = 8 (128 registers): one block ahead, two register sets
    f32x4 rb[WSETS][2];
    auto load_w = [&](int kb) {
#pragma unroll
        for (int i = 0; i < 2; ++i) rb[kb % WSETS][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)b_off[i], kb * 128, 0));
    };
#pragma unroll
    for (int kb = 0; kb < (WSETS == 7 ? 7 : 1); ++kb) load_w(kb);
    auto store_w = [&](int kb) {
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4 *>(Bs + ((kb & 1) * 64 + r0 + 32 * i) * ROWW + wsel) = rb[kb % WSETS][i];
    };
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int item = tid + 256 * i;
        if (item >= ST_HH * ST_HW) continue;
        u32x2 hi, lo;
        split4_f16(ra[i], hi, lo);
        *reinterpret_cast<u32x2 *>(Ah + item * 2) = hi;
        *reinterpret_cast<u32x2 *>(Al + item * 2) = lo;
    }
    store_w(0);
    __syncthreads();
    // the lane's two output pixels (row tiles i = 0, 1 of the wave's 64) and, per k16 step, its two taps t = 4 step + 2 h, + 1
    int a_pix[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int pix = wave * WPIX + i * 32 + l32;
        a_pix[i] = ((2 * (pix / ST_P)) * ST_HW + 2 * (pix % ST_P)) * 2;
    }
    f32x16 accm[RT][2], accx[RT][2];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { accm[i][j][r] = 0.f; accx[i][j][r] = 0.f; }
    constexpr int NSTEP = 13;                              // 49 taps in steps of 4; taps 49 .. 51 meet zero weights
#pragma unroll
    for (int kb = 0; kb < 7; ++kb) {
        if (WSETS != 7 && kb + 1 < 7) load_w(kb + 1);
        const unsigned int *bs = Bs + ((kb & 1) * 64 + l32) * ROWW + 4 * h;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int step = 2 * kb + s;
            if (step >= NSTEP) break;
            const int ta = min(4 * step + 2 * h, 48), tb = min(4 * step + 2 * h + 1, 48);
            const int oa = ((ta / 7) * ST_HW + ta % 7) * 2, ob = ((tb / 7) * ST_HW + tb % 7) * 2;
            f16x8 bh[2], bl[2];
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                bh[t2] = *reinterpret_cast<const f16x8 *>(bs + t2 * 32 * ROWW + 8 * s);
                bl[t2] = *reinterpret_cast<const f16x8 *>(bs + t2 * 32 * ROWW + 16 + 8 * s);
            }
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
                const u32x2 h0 = *reinterpret_cast<const u32x2 *>(Ah + a_pix[i] + oa), h1 = *reinterpret_cast<const u32x2 *>(Ah + a_pix[i] + ob);
                const u32x2 l0 = *reinterpret_cast<const u32x2 *>(Al + a_pix[i] + oa), l1 = *reinterpret_cast<const u32x2 *>(Al + a_pix[i] + ob);
                const f16x8 ah = __builtin_bit_cast(f16x8, u32x4v{h0[0], h0[1], h1[0], h1[1]});
                const f16x8 al = __builtin_bit_cast(f16x8, u32x4v{l0[0], l0[1], l1[0], l1[1]});
                accx[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[0], accx[i][0], 0, 0, 0);
                accx[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[1], accx[i][1], 0, 0, 0);
                accx[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[0], accx[i][0], 0, 0, 0);
                accx[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[1], accx[i][1], 0, 0, 0);
                accm[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[0], accm[i][0], 0, 0, 0);
                accm[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[1], accm[i][1], 0, 0, 0);
            }
        }
        if (kb + 1 < 7) store_w(kb + 1);
        __syncthreads();
    }
    // epilogue: the wave's 64 pixels x 64 channels through LDS, 16-B rows (Cout % 4 == 0 host-checked)
    float *ep = reinterpret_cast<float *>(lds) + wave * WPIX * 68;
#pragma unroll
    for (int tm = 0; tm < RT; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                ep[(tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 68 + tn * 32 + l32] = accm[tm][tn][r] + accx[tm][tn][r] * (1.0f / 2048.0f);
    const int c4e = lane & 15, rr = lane >> 4;
    const int col = c4e * 4;
    if (col < p.N) {
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
        if (p.scale) sc = *reinterpret_cast<const f32x4 *>(p.scale + col);
        if (p.bias) bi = *reinterpret_cast<const f32x4 *>(p.bias + col);
#pragma unroll 4
        for (int it = 0; it < 8 * RT; ++it) {
            const int pix = wave * WPIX + it * 4 + rr;
            const int oy = y0 + pix / ST_P, ox = x0 + pix % ST_P;
            if (oy >= Ho || ox >= Wo) continue;
            const long row = ((long)img * Ho + oy) * Wo + ox;
            f32x4 v = *reinterpret_cast<const f32x4 *>(&ep[(it * 4 + rr) * 68 + c4e * 4]);
            v = v * sc + bi;
            if (p.res) v += *reinterpret_cast<const f32x4 *>(p.res + row * p.ldr + col);
            if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
            *reinterpret_cast<f32x4 *>(p.C + row * p.ldc + col) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// High-occupancy split-fp16 x3 variant: 128 x 64 x 32 tile, 4 waves of 64 x 32 (two f32 accumulator sets = 64
// registers), single LDS buffer (27.6 KB operands, 36.9 KB with the epilogue staging).  ~4 workgroups = 16 waves per
// CU: while one workgroup splits / stores / waits at its barriers, three others keep the matrix pipe busy (the
// 128 x 128 kernel above is limited to 2 waves/SIMD by its 128 accumulator registers and leaves the pipe ~2/3 idle).
// PRE (short K with a residual: two to four k-steps, where the launch is a stream of tiles and the epilogue's residual read was a
// third dependent memory round trip per tile): the residual tile is fetched into registers before the first k-tile.
template <bool CONV, bool BSPLIT, bool PRE = false>
__global__ __launch_bounds__(256, PRE ? 3 : 4) void gemm_f16x3_hi_kernel(GemmParams p)
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
    f32x4 rres[8];
    if constexpr (PRE) {
        // the epilogue's mapping: lane (c4e, rr) owns columns n0 + wn * 32 + 4 c4e .. + 3 of rows m0 + wm * 64 + 8 it + rr
        const float *resp = p.res + (long)bz * p.sR;
        const int col = n0 + wn * 32 + (lane & 7) * 4, rbase = m0 + wm * 64 + (lane >> 3);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = rbase + it * 8;
            rres[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row < p.M && col < p.N && col < p.res_cols)
                rres[it] = *reinterpret_cast<const f32x4 *>(resp + (long)(p.res_rows ? row % p.res_rows : row) * p.ldr + col);
        }
    }
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
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = rbase + it * 8;
                if (row >= p.M) break;
                f32x4 v = *reinterpret_cast<const f32x4 *>(&ep[(it * 8 + rr) * 36 + c4e * 4]);
                v = v * sc + bi;
                if constexpr (PRE) v += rres[it];
                else if (res && col < p.res_cols) v += *reinterpret_cast<const f32x4 *>(res + (long)(p.res_rows ? row % p.res_rows : row) * p.ldr + col);
                if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                if (p.gate) {
                    const f32x4 g = *reinterpret_cast<const f32x4 *>(p.gate + (long)row * p.ldg + col);
                    v[0] = g[0] > 0.f ? v[0] * p.gate_scale : 0.f; v[1] = g[1] > 0.f ? v[1] * p.gate_scale : 0.f;
                    v[2] = g[2] > 0.f ? v[2] * p.gate_scale : 0.f; v[3] = g[3] > 0.f ? v[3] * p.gate_scale : 0.f;
                }
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

// ------------------------------------------------------------------------------------------------------------
// Wave-specialised persistent split-fp16 x3 kernel ("ws"): 128 x 128 x 32 tiles, 8 waves per workgroup, one workgroup
// per CU looping over its tiles.
//   waves 0-3 (one per SIMD)  consumers: 64 x 64 of the tile each; their instruction stream is ds_read_b128 + MFMA only
//                             (fragments of half a k-tile are read one half ahead of the MFMAs that use them), so
//                             the matrix pipe never waits behind a global load or the fp16 split of the same wave
//   waves 4-7 (one per SIMD)  producers: global loads D k-tiles ahead (D register sets), fp16 split, LDS stores of
//                             k-tile q+1 while the consumers work on k-tile q, and - spread over the k-loop of the
//                             NEXT tile - the epilogue of the previous tile (scale / bias / dropout / residual / ReLU,
//                             16-B row stores) out of an LDS staging area the consumers park their accumulators in
// The k-tile stream runs across tile boundaries, so prologue, epilogue and the C write-back of a tile overlap the
// MFMAs of its neighbours.  One s_barrier per k-tile couples the two roles (LDS: 2 operand slots of 36.9 KB + 69.6 KB
// of C staging = 143.4 KB).
template <bool CONV, bool BSPLIT, bool DROP, bool RES, bool ASPLIT = false>
__global__ __launch_bounds__(512, 1) void gemm_f16x3_ws_kernel(GemmParams p)
{
    constexpr int BM = 128, RPT = 32, D = 3, SLOT = (BM + BN) * ROWW;
    constexpr int EPC = DROP ? 8 : 16;                       // epilogue chunks per tile and producer wave
    constexpr int CPI = DROP ? 2 : 3;                        // chunks per k-tile: (nk - 1) * CPI >= EPC needs nk >= 7
    extern __shared__ __attribute__((aligned(16))) unsigned int lds[];
    unsigned int *ring = lds;                                             // [2][BM + BN][ROWW]
    float *cst = reinterpret_cast<float *>(lds + 2 * SLOT);               // [4][64][68]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int nwg = tiles_m * tiles_n;
    const int nk = (p.K + BK - 1) / BK;
    const int G = gridDim.x;
    const int my_tiles = (nwg - (int)blockIdx.x + G - 1) / G;
    const int Q = my_tiles * nk;                             // k-tiles of this workgroup, all its tiles back to back
    auto tile_origin = [&](int j, int &m0, int &n0) {
        int bid = blockIdx.x + j * G;
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, within = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
        m0 = (bid / tiles_n) * BM;
        n0 = (bid % tiles_n) * BN;
    };

    if (wave < 4) {
        // ---------------------------------------------------------------- consumers
        const int wm = wave >> 1, wn = wave & 1, l32 = lane & 31, h = lane >> 5;
        __builtin_amdgcn_s_setprio(2);
        f32x16 accm[2][2], accx[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) { accm[i][j][r] = 0.f; accx[i][j][r] = 0.f; }
        f16x8 ah[2][2], al[2][2], bh[2][2], bl[2][2];       // [half of the k-tile][row block / column block]
        const unsigned int *abase = ring + (wm * 64 + l32) * ROWW + 4 * h;
        const unsigned int *bbase = ring + (BM + wn * 64 + l32) * ROWW + 4 * h;
        auto rd = [&](int slot, auto S) {
            constexpr int s = decltype(S)::value;
            const unsigned int *as = abase + slot * SLOT, *bs = bbase + slot * SLOT;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                bh[s][t] = *reinterpret_cast<const f16x8 *>(bs + t * 32 * ROWW + 8 * s);
                bl[s][t] = *reinterpret_cast<const f16x8 *>(bs + t * 32 * ROWW + 16 + 8 * s);
                ah[s][t] = *reinterpret_cast<const f16x8 *>(as + t * 32 * ROWW + 8 * s);
                al[s][t] = *reinterpret_cast<const f16x8 *>(as + t * 32 * ROWW + 16 + 8 * s);
            }
        };
        auto mm = [&](auto S) {        // 12 MFMAs; the two updates of one cross accumulator are four MFMAs apart
            constexpr int s = decltype(S)::value;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s][i], bh[s][j], accx[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s][i], bl[s][j], accx[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) accm[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s][i], bh[s][j], accm[i][j], 0, 0, 0);
        };
        using H0 = std::integral_constant<int, 0>;
        using H1 = std::integral_constant<int, 1>;
        float *ep = cst + wave * 64 * 68;
        int kt = 0;
        __syncthreads();                                      // barrier 0: k-tile 0 is in slot 0
        rd(0, H0{});
        for (int q = 0; q < Q; ++q) {
            const int slot = q & 1;
            rd(slot, H1{});
            __builtin_amdgcn_sched_barrier(0);
            mm(H0{});
            __builtin_amdgcn_sched_barrier(0);                // the MFMAs cover the latency of the reads before the barrier
            __syncthreads();                                  // k-tile q+1 stored; every read of slot q has landed
            if (q + 1 < Q) rd(slot ^ 1, H0{});
            __builtin_amdgcn_sched_barrier(0);
            mm(H1{});
            __builtin_amdgcn_sched_barrier(0);
            if (++kt == nk) {
                kt = 0;
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            ep[(tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 68 + tn * 32 + l32] =
                                accm[tm][tn][r] + accx[tm][tn][r] * (1.0f / 2048.0f);
                            accm[tm][tn][r] = 0.f; accx[tm][tn][r] = 0.f;
                        }
            }
        }
        __syncthreads();                                      // the last tile is parked
        return;
    }

    // -------------------------------------------------------------------- producers
    // The loop bodies have no branches around loads, so every s_waitcnt vmcnt(N) the compiler places is the exact count of
    // younger operations and a k-tile's loads stay in flight for D iterations.  Every load is IN RANGE: an out-of-range
    // buffer load (the usual way to get zeros for masked rows / taps) returns without a memory access and ahead of older
    // in-range loads, so a counted vmcnt wait behind one is satisfied before the data it guards has landed (measured:
    // sporadic stale first rows of a register set whenever a masked load sat behind it; profiles/r2_ws_gemm/README.md).
    // Masked rows / columns read a clamped address instead (their products are never stored), dead k-tiles past the end of
    // the stream re-read offset 0, and padding taps / the K tail of a convolution are zeroed by a select after the load.
    const int pw = wave - 4, pt = tid - 256;
    const int c4 = pt & 7, g = pt >> 3;
    const int r0 = (g & ~7) | ((g & 1) << 2) | ((g >> 1) & 3);
    const __amdgpu_buffer_rsrc_t rsA = ASPLIT
        ? __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned int *>(p.Asplit), 0, (int)((long)p.M * p.kblocks * 128L), 0x00020000)
        : __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.A), 0, (int)p.bytesA, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = BSPLIT
        ? __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned int *>(p.Bsplit), 0, (int)((long)p.N * p.kblocks * 128L), 0x00020000)
        : __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.B), 0, (int)p.bytesB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)(((long)(p.M - 1) * p.ldc + p.N) * 4L), 0x00020000);
    const int res_h = p.res_rows ? p.res_rows : p.M;
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(RES ? p.res : p.A), 0, RES ? (int)(((long)(res_h - 1) * p.ldr + (p.res_cols < p.N ? p.res_cols : p.N)) * 4L) : 32, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.scale ? p.scale : p.ones), 0, p.N * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsI = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.bias ? p.bias : p.zeros), 0, p.N * 4, 0x00020000);
    const float relu_lo = p.relu ? 0.f : -__builtin_inff();
    const int wsel = (c4 & 3) * 4 + (c4 >> 2) * 16;
    // load stream: tile lj, k-tile lkt of it.  Past the end of the stream it keeps re-reading the last tile's k-tiles
    // (in range, never stored to the ring's live slot).
    int lj = 0, lkt = 0;
    unsigned int a_off[4], b_off[4];
    int a_iy0[4], a_ix0[4];
    auto set_tile = [&](int j) {
        int m0, n0;
        tile_origin(j, m0, n0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + r0 + RPT * i;
            const int mc = m < p.M ? m : p.M - 1;              // rows past M: a valid row whose products nobody stores
            if (CONV) {
                const int ox = mc % p.Wout, t = mc / p.Wout, oy = t % p.Hout, n = t / p.Hout;
                a_iy0[i] = oy * p.stride - p.pad;
                a_ix0[i] = ox * p.stride - p.pad;
                a_off[i] = (unsigned int)((long)n * p.Hin * p.Win * p.Cin * 4L);
            } else {
                a_off[i] = ASPLIT ? (unsigned int)((long)mc * p.kblocks * 128L + wsel * 4) : (unsigned int)((long)mc * p.lda * 4L) + (unsigned int)(c4 * 16);
                a_iy0[i] = a_ix0[i] = 0;
            }
            const int n = n0 + r0 + RPT * i, nc = n < p.N ? n : p.N - 1;
            b_off[i] = (unsigned int)(BSPLIT ? (long)nc * p.kblocks * 128L + wsel * 4 : (long)nc * p.ldb * 4L + c4 * 16);
        }
    };
    auto load_set = [&](f32x4 (&ra)[4], f32x4 (&rb)[4], unsigned int &zm) {   // next k-tile of the stream -> one register set; zm: rows to zero when the set is stored
        const int kb = lkt * (BK * 4);                          // uniform byte offset of the k-tile (K % 32 == 0 off the conv path)
        zm = 0u;
        int kh = 0, kw = 0, ci = 0;
        bool kok = true;
        if (CONV) {
            const int k = lkt * BK + c4 * 4;
            kok = k < p.K;                                     // K % 4 == 0 (host-checked): a float4 is inside or outside
            const int kc = kok ? k : 0;
            const int tap = kc / p.Cin;
            ci = kc - tap * p.Cin;
            kh = tap / p.KW;
            kw = tap - kh * p.KW;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned int off;
            if (CONV) {
                const int iy = a_iy0[i] + kh, ix = a_ix0[i] + kw;
                const bool zero = ((iy | ix | (p.Hin - 1 - iy) | (p.Win - 1 - ix)) < 0) || !kok;           // padding tap or the K tail
                const int iyc = iy < 0 ? 0 : (iy >= p.Hin ? p.Hin - 1 : iy), ixc = ix < 0 ? 0 : (ix >= p.Win ? p.Win - 1 : ix);
                off = a_off[i] + (unsigned int)(((iyc * p.Win + ixc) * p.Cin + ci) * 4);
                if (zero) zm |= 1u << i;
            } else {
                off = a_off[i];
            }
            // the k-tile's byte offset rides in the instruction's scalar offset: no per-lane address arithmetic
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)off, CONV ? 0 : kb, 0));
            // B: pre-split images are zero padded past K; a dynamic B has K % 32 == 0 here
            rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)b_off[i], kb, 0));
        }
        if (++lkt == nk) {
            lkt = 0;
            if (++lj < my_tiles) set_tile(lj);
        }
    };
    auto store_set = [&](const f32x4 (&ra)[4], const f32x4 (&rb)[4], unsigned int zm, int slot) {
        unsigned int *As = ring + slot * SLOT, *Bs = As + BM * ROWW;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u32x2 hi, lo;
            unsigned int *row = &As[(r0 + RPT * i) * ROWW];
            if (ASPLIT) *reinterpret_cast<f32x4 *>(row + wsel) = ra[i];
            else {
                split4_f16(ra[i], hi, lo);
                if (CONV && ((zm >> i) & 1u)) { hi = u32x2{0u, 0u}; lo = u32x2{0u, 0u}; }
                *reinterpret_cast<u32x2 *>(row + c4 * 2) = hi;
                *reinterpret_cast<u32x2 *>(row + 16 + c4 * 2) = lo;
            }
            row = &Bs[(r0 + RPT * i) * ROWW];
            if (BSPLIT) { *reinterpret_cast<f32x4 *>(row + wsel) = rb[i]; continue; }
            split4_f16(rb[i], hi, lo);
            *reinterpret_cast<u32x2 *>(row + c4 * 2) = hi;
            *reinterpret_cast<u32x2 *>(row + 16 + c4 * 2) = lo;
        }
    };
    // epilogue stream: producer wave pw writes back the 64 x 64 block consumer wave pw parked, CPI row chunks per k-tile
    // (a chunk = 4 rows x 256 B, or 8 rows with dropout where a lane owns an 8-column mask block).  The loads a chunk needs
    // (residual rows, per-column scale / bias) are issued D iterations before the chunk is written, in the same register
    // set and right behind the operand loads that will be consumed in that iteration, so waiting for them never drains
    // a younger operand set.  Per tile the stream keeps row / column / byte offsets of this lane's first chunk; a chunk adds
    // a uniform multiple, so an iteration costs a handful of VALU instructions besides the arithmetic itself.
    constexpr int EW = DROP ? 8 : 4;                          // columns per lane
    constexpr int ER = DROP ? 8 : 4;                          // rows per chunk
    constexpr int EV = EW / 4;
    const int ewm = pw >> 1, ewn = pw & 1;
    const int ecl = (DROP ? (lane & 7) : (lane & 15)) * EW, erl = DROP ? (lane >> 3) : (lane >> 4);
    const float *epb = cst + pw * 64 * 68 + erl * 68 + ecl;  // this lane's float4 of chunk 0
    struct EpSet {                                            // what one iteration's chunks need, per register set
        f32x4 sc[EV], bi[EV], rs[CPI][EV];
        float rmask[EV];                                      // 1: this lane's columns take the residual, 0: they do not
        int row, col;                                         // this lane's row of chunk 0 / first column, of the tile the chunks belong to
        unsigned int coff;                                    // byte offset of (row, col) in C
        int it0;                                              // first chunk (uniform); < 0 or >= EPC: nothing to do
    };
    // position of the iteration the loads are issued FOR (x = q + D): x = xtile * nk + xkt
    int xkt = 0, xtile = 0, x_row = ewm * 64 + erl, x_col = ewn * 64 + ecl, x_rr = 0;
    unsigned int x_coff = 0u, x_cb = 0u;
    float x_rmask[EV] = {};
    const int ldc4 = (int)p.ldc * 4 * ER;
    auto ep_issue = [&](EpSet &s, bool live) {
        // the tile that ended at k-tile x - xkt - 1 is parked after barrier x - xkt and visible from xkt = 1 on; it must be
        // drained before the next one is parked: nk - 1 iterations x CPI chunks >= EPC (host-checked)
        if (xkt == 1 && xtile > 0) {
            int m0, n0;
            tile_origin(xtile - 1, m0, n0);
            x_row = m0 + ewm * 64 + erl;
            x_col = n0 + ewn * 64 + ecl;
            x_coff = (unsigned int)((long)x_row * p.ldc * 4L) + (unsigned int)(x_col * 4);
            x_cb = x_col < p.N ? (unsigned int)(x_col * 4) : 0u;
            if (RES) {
                x_rr = p.res_rows ? x_row % p.res_rows : x_row;
#pragma unroll
                for (int w = 0; w < EV; ++w) x_rmask[w] = x_col + 4 * w < p.res_cols ? 1.f : 0.f;
            }
        }
        if (CONV) {
#pragma unroll
            for (int w = 0; w < EV; ++w) s.sc[w] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsS, (int)x_cb, 16 * w, 0));
        }
#pragma unroll
        for (int w = 0; w < EV; ++w) s.bi[w] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsI, (int)x_cb, 16 * w, 0));
        s.row = x_row; s.col = x_col; s.coff = x_coff;
        s.it0 = (live && xtile > 0) ? (xkt - 1) * CPI : -CPI;
        if (RES) {
            const int itb = s.it0 < 0 ? 0 : s.it0;
#pragma unroll
            for (int w = 0; w < EV; ++w) s.rmask[w] = x_rmask[w];
#pragma unroll
            for (int c = 0; c < CPI; ++c) {
                const int itc = itb + c < EPC ? itb + c : EPC - 1;      // dead chunks re-read a live row
                int rr = x_rr + itc * ER;                      // res_rows >= 64 (host-checked): one conditional subtraction
                if (p.res_rows) rr = rr >= p.res_rows ? rr - p.res_rows : rr;
                else rr = rr < p.M ? rr : p.M - 1;
                // columns without a residual (rmask 0) read the row's first columns: finite values, multiplied by 0
                const unsigned int ro = (unsigned int)(rr * (int)p.ldr) * 4u + (x_rmask[EV - 1] != 0.f ? x_cb : 0u);
#pragma unroll
                for (int w = 0; w < EV; ++w)
                    s.rs[c][w] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, (int)ro, 16 * w, 0));
            }
        }
        if (++xkt == nk) { xkt = 0; ++xtile; }
    };
    // the chunks' LDS reads are issued at the top of the iteration and consumed after the operand split, so their latency
    // is covered by that arithmetic
    auto ep_read = [&](const EpSet &s, f32x4 (&v)[CPI][EV]) {
        if (s.it0 < 0 || s.it0 >= EPC) return;               // uniform
#pragma unroll
        for (int c = 0; c < CPI; ++c) {
            const int it = s.it0 + c < EPC ? s.it0 + c : EPC - 1;
#pragma unroll
            for (int w = 0; w < EV; ++w) v[c][w] = *reinterpret_cast<const f32x4 *>(epb + it * (ER * 68) + 4 * w);
        }
    };
    auto ep_write = [&](const EpSet &s, f32x4 (&v)[CPI][EV]) {
        if (s.it0 < 0 || s.it0 >= EPC) return;               // uniform
        const bool cok = s.col < p.N;
#pragma unroll
        for (int c = 0; c < CPI; ++c) {
            const int it = s.it0 + c;
            if (it >= EPC) break;                              // uniform
            const int row = s.row + it * ER;
#pragma unroll
            for (int w = 0; w < EV; ++w) {
                if (CONV) v[c][w] *= s.sc[w];
                v[c][w] += s.bi[w];
            }
            if constexpr (DROP) {
                float m[8];
                s2d_dropout8((uint32_t)row + p.drop_row0, (uint32_t)(s.col >> 3), p.drop_stream, p.drop_k0, p.drop_k1, p.drop_thresh, p.drop_scale, m);
                v[c][0][0] *= m[0]; v[c][0][1] *= m[1]; v[c][0][2] *= m[2]; v[c][0][3] *= m[3];
                v[c][EV - 1][0] *= m[4]; v[c][EV - 1][1] *= m[5]; v[c][EV - 1][2] *= m[6]; v[c][EV - 1][3] *= m[7];
            }
#pragma unroll
            for (int w = 0; w < EV; ++w) {
                if (RES) {                                     // element-wise on purpose: no packed op as the first reader of a loaded register (scripts/isa_lint.py)
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[c][w][j] = __builtin_fmaf(s.rs[c][w][j], s.rmask[w], v[c][w][j]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) v[c][w][j] = __builtin_amdgcn_fmed3f(v[c][w][j], relu_lo, __builtin_inff());
            }
            // the store is predicated, not masked out of range: an out-of-range store retires early just like a load
            if (cok && row < p.M) {
#pragma unroll
                for (int w = 0; w < EV; ++w)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v[c][w]), rsC, (int)s.coff, it * ldc4 + 16 * w, 0);
            }
        }
    };

    f32x4 ra[D][4], rb[D][4];
    unsigned int zm[D];
    EpSet es[D];
    set_tile(0);
    // prologue: the same operation pattern per set as a loop iteration.  Iterations 0 .. D-1 have nothing to write back
    // (nk > D), so their epilogue sets are dead; the stream position starts at x = D.
#pragma unroll
    for (int d = 0; d < D; ++d) {                             // Q >= nk > D
        load_set(ra[d], rb[d], zm[d]);
        ep_issue(es[d], false);
        __builtin_amdgcn_sched_barrier(0);                    // the loop's issue order, so that its waits can be counted from here
    }
    store_set(ra[0], rb[0], zm[0], 0);
    __builtin_amdgcn_sched_barrier(0);
    load_set(ra[0], rb[0], zm[0]);
    ep_issue(es[0], false);                                   // rides with k-tile D = iteration D - 1: still the first tile
    __builtin_amdgcn_sched_barrier(0);
    xkt = D; xtile = 0;
    __syncthreads();                                          // barrier 0
    auto body = [&](int q, auto U) {
        constexpr int s = (decltype(U)::value + 1) % D;       // q = q0 + u with q0 % D == 0: k-tile q+1 lives in set (u+1) % D
        // k-tile q+1 -> the slot the consumers left at the last barrier (past the end: a re-read nobody uses), this
        // iteration's chunks of the parked tile, then the register set takes k-tile q+1+D and the chunk loads of
        // iteration q+D (where that k-tile is stored).  sched_barrier: issue order = program order, the vmcnt arithmetic
        // counts on it.
        f32x4 ev[CPI][EV];
        ep_read(es[s], ev);
        __builtin_amdgcn_sched_barrier(0);
        store_set(ra[s], rb[s], zm[s], (q + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        ep_write(es[s], ev);
        __builtin_amdgcn_sched_barrier(0);
        load_set(ra[s], rb[s], zm[s]);
        ep_issue(es[s], q + D < Q);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    };
    using U0 = std::integral_constant<int, 0>;
    using U1 = std::integral_constant<int, 1>;
    using U2 = std::integral_constant<int, 2>;
    static_assert(D == 3, "the unrolled bodies below are written for three register sets");
    // whole groups of D iterations without a branch between them (a guarded body would make the compiler count only the
    // loads of the guaranteed bodies as younger and wait for a set one iteration after it was issued), then the remainder
    int q0 = 0;
    for (; q0 + D <= Q; q0 += D) { body(q0, U0{}); body(q0 + 1, U1{}); body(q0 + 2, U2{}); }
    if (q0 < Q) body(q0, U0{});
    if (q0 + 1 < Q) body(q0 + 1, U1{});
    __syncthreads();                                          // the last tile is parked
    xtile = my_tiles;                                         // its chunks, CPI at a time
    for (int it0 = 0; it0 < EPC; it0 += CPI) {
        xkt = it0 / CPI + 1;
        ep_issue(es[0], true);
        xtile = my_tiles;
        f32x4 ev[CPI][EV];
        ep_read(es[0], ev);
        ep_write(es[0], ev);
    }
}

template <bool CONV>
int launch_f16_hi(const GemmParams &p, int batch, hipStream_t st)
{
    const size_t lds = sizeof(float) * 4 * 64 * 36;   // 36.9 KB: epilogue staging >= operand tiles (27.6 KB)
    const int nwg = cdiv(p.M, 128) * cdiv(p.N, 64);
    static int pre = -1;
    if (pre < 0) { const char *e = getenv("S2D_GEMM_HI_PRE"); pre = e ? atoi(e) : 1; }
    // residual prefetch: short K (the 1 x 1 convolutions of res2 / res3 that close a bottleneck), vector epilogue, no gate
    if (pre && p.Bsplit && p.res && !p.gate && p.K <= 4 * BK && ((p.N | p.ldc | p.ldr | p.res_cols) & 3) == 0)
        hipLaunchKernelGGL((gemm_f16x3_hi_kernel<CONV, true, true>), dim3(nwg, batch), dim3(256), lds, st, p);
    else if (p.Bsplit) hipLaunchKernelGGL((gemm_f16x3_hi_kernel<CONV, true>), dim3(nwg, batch), dim3(256), lds, st, p);
    else hipLaunchKernelGGL((gemm_f16x3_hi_kernel<CONV, false>), dim3(nwg, batch), dim3(256), lds, st, p);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

template <bool CONV, bool PIPE, bool BSPLIT, bool DROP = false>
int launch_f16_v(const GemmParams &p, int batch, hipStream_t st)
{
    const size_t lds = sizeof(unsigned int) * 2 * (128 + BN) * ROWW;
    static S2dDevOnce attr_set;
    if (!attr_set.done()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_f16x3_kernel<CONV, PIPE, BSPLIT, DROP>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return S2D_ERR_LAUNCH;
        attr_set.mark();
    }
    const int nwg = cdiv(p.M, 128) * cdiv(p.N, BN);
    hipLaunchKernelGGL((gemm_f16x3_kernel<CONV, PIPE, BSPLIT, DROP>), dim3(nwg, batch), dim3(256), lds, st, p);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

constexpr int WS_CONST_N = 16384;
__global__ void ws_constants_kernel(float *z, float *o)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    z[i] = 0.f; o[i] = 1.f;
}

// per device: WS_CONST_N zeros and ones (an absent bias / scale vector of the wave-specialised kernel)
int s2d_ws_constants(const float **zeros, const float **ones, hipStream_t st)
{
    static float *buf[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return S2D_ERR_LAUNCH;
    if (!buf[dev]) {
        float *b = nullptr;
        if (hipMalloc(&b, sizeof(float) * 2 * WS_CONST_N) != hipSuccess) return S2D_ERR_LAUNCH;
        hipLaunchKernelGGL(ws_constants_kernel, dim3(WS_CONST_N / 256), dim3(256), 0, st, b, b + WS_CONST_N);
        if (hipStreamSynchronize(st) != hipSuccess) return S2D_ERR_LAUNCH;   // other streams may launch next
        buf[dev] = b;
    }
    *zeros = buf[dev]; *ones = buf[dev] + WS_CONST_N;
    return S2D_OK;
}

template <bool CONV, bool BSPLIT, bool DROP, bool RES, bool ASPLIT = false>
int launch_f16_ws_v(const GemmParams &p, hipStream_t st)
{
    const size_t lds = sizeof(unsigned int) * 2 * (128 + BN) * ROWW + sizeof(float) * 4 * 64 * 68;
    static S2dDevOnce attr_set;
    static int cus = 0;
    if (!attr_set.done()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_f16x3_ws_kernel<CONV, BSPLIT, DROP, RES, ASPLIT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return S2D_ERR_LAUNCH;
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return S2D_ERR_LAUNCH;
        cus = prop.multiProcessorCount;
        attr_set.mark();
    }
    const int nwg = cdiv(p.M, 128) * cdiv(p.N, BN);
    GemmParams q = p;
    if (s2d_ws_constants(&q.zeros, &q.ones, st) != S2D_OK) return S2D_ERR_LAUNCH;
    hipLaunchKernelGGL((gemm_f16x3_ws_kernel<CONV, BSPLIT, DROP, RES, ASPLIT>), dim3(nwg < cus ? nwg : cus), dim3(512), lds, st, q);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

template <bool CONV, bool BSPLIT, bool DROP>
int launch_f16_ws(const GemmParams &p, hipStream_t st)
{
    return p.res ? launch_f16_ws_v<CONV, BSPLIT, DROP, true>(p, st) : launch_f16_ws_v<CONV, BSPLIT, DROP, false>(p, st);
}

template <bool CONV, bool PIPE>
int launch_f16(const GemmParams &p, int batch, hipStream_t st)
{
    return p.Bsplit ? launch_f16_v<CONV, PIPE, true>(p, batch, st) : launch_f16_v<CONV, PIPE, false>(p, batch, st);
}

int launch_conv7x7s2_stem(const GemmParams &p, hipStream_t st)
{
    static int ph = -1;
    if (ph < 0) { const char *e = getenv("S2D_CONV_STEM_PH"); ph = e ? atoi(e) : 8; }
    const size_t lds16 = sizeof(float) * 4 * 64 * 68, lds8 = sizeof(float) * 4 * 32 * 68;       // epilogue staging >= halo planes + two weight buffers
    static S2dDevOnce attr_set;
    if (!attr_set.done()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv7x7s2_c4_halo_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16) != hipSuccess)
            return S2D_ERR_LAUNCH;
        attr_set.mark();
    }
    const int imgs = p.M / (p.Hout * p.Wout);
    if (ph == 16) hipLaunchKernelGGL(conv7x7s2_c4_halo_kernel<16>, dim3(imgs * cdiv(p.Hout, 16) * cdiv(p.Wout, ST_P)), dim3(256), lds16, st, p);
    else hipLaunchKernelGGL(conv7x7s2_c4_halo_kernel<8>, dim3(imgs * cdiv(p.Hout, 8) * cdiv(p.Wout, ST_P)), dim3(256), lds8, st, p);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

int launch_conv3x3_halo(const GemmParams &pin, hipStream_t st)
{
    const GemmParams &p = pin;
    // opt-in (S2D_CONV_HALO_PIPE=1, read per call): measured +2.6 % on 942 080 x 256 x 2 304, -1..-2 % on the smaller 3 x 3 shapes, -9 % on the
    // 64-channel form (scripts/mb_conv3_pipe.py, profiles/r5_experiments/not_adopted.txt): at two waves per SIMD the fragment latency the
    // round-4 form exposes is already covered by the other wave
    int pipe = 0;
    if (const char *e = getenv("S2D_CONV_HALO_PIPE")) pipe = atoi(e);
    const size_t lds = sizeof(float) * 4 * 64 * 68;           // epilogue staging (69.6 KB) >= halo + two weight buffers (62.8 / 65.1 KB)
    static S2dDevOnce attr_set;
    if (!attr_set.done()) {
        for (const void *fn : {reinterpret_cast<const void *>(conv3x3_f16x3_halo_kernel<8, 2, true>), reinterpret_cast<const void *>(conv3x3_f16x3_halo_kernel<16, 1, true>),
                               reinterpret_cast<const void *>(conv3x3_f16x3_halo_kernel<8, 2, false>), reinterpret_cast<const void *>(conv3x3_f16x3_halo_kernel<16, 1, false>)})
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return S2D_ERR_LAUNCH;
        attr_set.mark();
    }
    const int imgs = p.M / (p.Hin * p.Win);
    if (p.N <= 64) {
        const int nwg = imgs * cdiv(p.Hin, 16) * cdiv(p.Win, HT_W);
        if (pipe) hipLaunchKernelGGL((conv3x3_f16x3_halo_kernel<16, 1, true>), dim3(nwg), dim3(256), lds, st, p);
        else hipLaunchKernelGGL((conv3x3_f16x3_halo_kernel<16, 1, false>), dim3(nwg), dim3(256), lds, st, p);
    } else {
        const int nwg = imgs * cdiv(p.Hin, HT_H) * cdiv(p.Win, HT_W) * cdiv(p.N, BN);
        if (pipe) hipLaunchKernelGGL((conv3x3_f16x3_halo_kernel<8, 2, true>), dim3(nwg), dim3(256), lds, st, p);
        else hipLaunchKernelGGL((conv3x3_f16x3_halo_kernel<8, 2, false>), dim3(nwg), dim3(256), lds, st, p);
    }
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

template <bool CONV, bool BSPLIT>
int launch_w128_v(const GemmParams &p, int batch, hipStream_t st)
{
    const size_t lds = sizeof(float) * 4 * 64 * 68;           // epilogue staging (69.6 KB) >= operand buffer (55.3 KB)
    static S2dDevOnce attr_set;
    if (!attr_set.done()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_bf16x3_w128_kernel<CONV, BSPLIT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return S2D_ERR_LAUNCH;
        attr_set.mark();
    }
    const int nwg = cdiv(p.M, 256) * cdiv(p.N, BN);
    hipLaunchKernelGGL((gemm_bf16x3_w128_kernel<CONV, BSPLIT>), dim3(nwg, batch), dim3(256), lds, st, p);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

template <bool CONV>
int launch_w128(const GemmParams &p, int batch, hipStream_t st)
{
    return p.Bsplit ? launch_w128_v<CONV, true>(p, batch, st) : launch_w128_v<CONV, false>(p, batch, st);
}

template <int WM, bool CONV>
int launch_t(const GemmParams &p, int batch, hipStream_t st)
{
    constexpr int BM = 64 * WM;
    const size_t lds = sizeof(unsigned int) * 2 * (BM + BN) * ROWW;
    static S2dDevOnce attr_set;
    if (!attr_set.done()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_bf16x3_kernel<WM, CONV>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return S2D_ERR_LAUNCH;
        attr_set.mark();
    }
    const int nwg = cdiv(p.M, BM) * cdiv(p.N, BN);
    hipLaunchKernelGGL((gemm_bf16x3_kernel<WM, CONV>), dim3(nwg, batch), dim3(128 * WM), lds, st, p);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

// static weights -> [N][kblocks][16 words hi | 16 words lo]: the LDS row image of the split-fp16 kernels, zero padded past K
template <bool BF16>
__global__ __launch_bounds__(256) void split_weights_kernel(const float *__restrict__ W, int N, int K, long ldw, int kblocks,
                                                            unsigned int *__restrict__ out, int vec4)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)N * kblocks * 8) return;
    const int c4 = (int)(i & 7);
    const long nb = i >> 3;
    const int kb = (int)(nb % kblocks);
    const long n = nb / kblocks;
    const int k = kb * 32 + c4 * 4;
    // one 16-B load when the row layout allows it (every GEMM operand of the library: K, ldw multiples of 4, 16-B aligned base):
    // the conversions below are packed ops, which must not be the first readers of single-dword loads (scripts/isa_lint.py)
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (vec4) { if (k < K) v = *reinterpret_cast<const f32x4 *>(W + n * ldw + k); }
    else {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (k + j < K) t[j] = W[n * ldw + k + j];
        // ragged rows: every value passes through a plain move first (an opaque v_mov_b32 the optimiser cannot fold away)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float r;
            asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(t[j]));
            v[j] = r;
        }
    }
    u32x2 hi, lo;
    if (BF16) split4(v, hi, lo);            // split-bf16 mode: unscaled low part (gemm_bf16x3_w128_kernel)
    else split4_f16(v, hi, lo);
    unsigned int *o = out + nb * 32;
    *reinterpret_cast<u32x2 *>(o + c4 * 2) = hi;
    *reinterpret_cast<u32x2 *>(o + 16 + c4 * 2) = lo;
}

}  // namespace

int s2d_split_weights_launch(const float *W, int N, int K, long ldw, unsigned int *out, hipStream_t st, int bf16)
{
    const int kblocks = (K + 31) / 32;
    const long n = (long)N * kblocks * 8;
    if (n == 0) return S2D_OK;
    const int vec4 = ((K | ldw) & 3) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0;
    if (bf16) hipLaunchKernelGGL(split_weights_kernel<true>, dim3(cdiv(n, 256)), dim3(256), 0, st, W, N, K, ldw, kblocks, out, vec4);
    else hipLaunchKernelGGL(split_weights_kernel<false>, dim3(cdiv(n, 256)), dim3(256), 0, st, W, N, K, ldw, kblocks, out, vec4);
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
        if ((batch > 1 && p.sB != 0) || (long)p.N * ((p.K + 31) / 32) * 128L > 0xFFFFFF00L) return S2D_ERR_ARG;
        p.kblocks = (p.K + 31) / 32;
    }
    if (!f16 && (p.Asplit || p.gate || p.drop_thresh)) return S2D_ERR_ARG;
    if (s2d_gemm_small_m_ok(p, conv, batch, f16)) return s2d_launch_gemm_small_m(p, st);      // the video decoder's query side (M = 200)
    static int ws = -1;
    if (ws < 0) { const char *e = getenv("S2D_GEMM_WS"); ws = e ? atoi(e) : 0; }

    // wave-specialised persistent kernel: one batch slice, >= 7 k-tiles (the previous tile's write-back is spread over the
    // k-loop), 16-B rows for the row-wise epilogue, 32-bit buffer offsets into C and the residual
    const long bC = ((long)(p.M - 1) * p.ldc + p.N) * 4L, bR = p.res ? ((long)((p.res_rows ? p.res_rows : p.M) - 1) * p.ldr + p.N) * 4L : 0;
    const bool ws_ok = ws && f16 && batch == 1 && !p.gate && p.N <= WS_CONST_N && (conv || !p.scale) && p.K >= 7 * BK - (BK - 1) && ((p.N | p.ldc | p.ldr | p.res_cols) & 3) == 0 &&
                       (conv ? (p.Cin % 4 == 0 && (p.Bsplit || p.K % BK == 0)) : p.K % BK == 0) &&
                       bC <= 0xFFFFFF00L && bR <= 0xFFFFFF00L && (!p.res_rows || p.res_rows >= 64) &&
                       (!p.drop_thresh || (((p.N | p.ldc) & 7) == 0 && (!p.res || ((p.ldr | p.res_cols) & 7) == 0)));
    if (p.Asplit) {
        // pre-split A: only the wave-specialised kernel reads it (its producers copy); the caller guarantees a static B image
        const bool ok = f16 && !conv && batch == 1 && p.Bsplit && !p.scale && p.N <= WS_CONST_N && p.K >= 7 * BK - (BK - 1) && p.K % BK == 0 &&
                        ((p.N | p.ldc | p.ldr | p.res_cols) & 3) == 0 && bC <= 0xFFFFFF00L && bR <= 0xFFFFFF00L &&
                        (long)p.M * p.kblocks * 128L <= 0xFFFFFF00L && (!p.res_rows || p.res_rows >= 64) &&
                        (!p.drop_thresh || (((p.N | p.ldc) & 7) == 0 && (!p.res || ((p.ldr | p.res_cols) & 7) == 0)));
        if (!ok) return S2D_ERR_ARG;
        if (p.drop_thresh) return p.res ? launch_f16_ws_v<false, true, true, true, true>(p, st) : launch_f16_ws_v<false, true, true, false, true>(p, st);
        return p.res ? launch_f16_ws_v<false, true, false, true, true>(p, st) : launch_f16_ws_v<false, true, false, false, true>(p, st);
    }
    if (ws_ok && !conv && (ws == 2 || (p.Bsplit && (long)cdiv(p.M, 128) * cdiv(p.N, BN) >= 512))) {
        if (p.drop_thresh) return p.Bsplit ? launch_f16_ws<false, true, true>(p, st) : launch_f16_ws<false, false, true>(p, st);
        return p.Bsplit ? launch_f16_ws<false, true, false>(p, st) : launch_f16_ws<false, false, false>(p, st);
    }
    if (ws_ok && conv && !p.drop_thresh && (ws == 2 || (long)cdiv(p.M, 128) * cdiv(p.N, BN) >= 512)) {
        return p.Bsplit ? launch_f16_ws<true, true, false>(p, st) : launch_f16_ws<true, false, false>(p, st);
    }
    if (p.gate && (!f16 || batch != 1 || p.drop_thresh || ((p.N | p.ldc | p.ldr | p.res_cols | p.ldg) & 3))) return S2D_ERR_ARG;
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
        static int halo = -1;
        if (halo < 0) { const char *e = getenv("S2D_CONV_HALO"); halo = e ? atoi(e) : 1; }
        // 3 x 3 / stride 1 / pad 1 on whole 32-channel blocks with static weights: the input-halo kernel
        // ... when its 8 x 16 patches cover the image without much overhang (23 x 40 -> 24 x 48 wastes 20 %: implicit GEMM wins)
        static int stem = -1;
        if (stem < 0) { const char *e = getenv("S2D_CONV_STEM"); stem = e ? atoi(e) : 1; }
        // the R50 stem geometry: its own halo kernel
        if (stem && conv && p.KH == 7 && p.KW == 7 && p.stride == 2 && p.pad == 3 && p.Cin == 4 && p.N <= 64 && p.N > 32 && p.Bsplit && batch == 1 &&
            ((p.N | p.ldc | p.ldr) & 3) == 0 && !p.res_rows && p.res_cols == p.N && !p.gate && (long)p.Hin * p.Win * 16L * (p.M / ((long)p.Hout * p.Wout)) < 0x7fffffffL)
            return launch_conv7x7s2_stem(p, st);
        static int halo64 = -1;
        if (halo64 < 0) { const char *e = getenv("S2D_CONV_HALO64"); halo64 = e ? atoi(e) : 1; }
        const int ph = p.N <= 64 ? 16 : HT_H;                 // Cout <= 64: 16 x 16 patches x 64 channels
        if (halo && conv && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && p.Cin % 32 == 0 && p.Bsplit && batch == 1 && (p.N > 64 || (halo64 && p.N > 32)) &&
            ((p.N | p.ldc | p.ldr) & 3) == 0 && !p.res_rows && p.res_cols == p.N && !p.gate &&
            (halo == 2 || (long)p.Hin * p.Win * 100 >= (long)cdiv(p.Hin, ph) * ph * cdiv(p.Win, HT_W) * HT_W * 88))
            return launch_conv3x3_halo(p, st);
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
    static int w128 = -1;
    if (w128 < 0) { const char *e = getenv("S2D_GEMM_W128"); w128 = e ? atoi(e) : 1; }
    // 128 x 64 per wave: GEMMs with enough tiles for two workgroups per CU (the implicit-GEMM form needs more address registers
    // than the 256-register budget leaves and spills: convolutions stay on the 64 x 64 kernels)
    if (w128 && !conv && (long)cdiv(p.M, 256) * cdiv(p.N, BN) * batch >= (w128 == 2 ? 1 : 512)) return launch_w128<false>(p, batch, st);
    static int force = -1;
    if (force < 0) { const char *e = getenv("S2D_GEMM_WM"); force = e ? atoi(e) : 0; }
    bool big = (long)cdiv(p.M, 256) * cdiv(p.N, BN) * batch >= 256;
    if (force == 2) big = false;
    if (force == 4) big = true;
    if (big) return conv ? launch_t<4, true>(p, batch, st) : launch_t<4, false>(p, batch, st);
    return conv ? launch_t<2, true>(p, batch, st) : launch_t<2, false>(p, batch, st);
}
