// Dense contractions with fp16 OPERANDS and f32 accumulate / f32 OUTPUT ("autocast-like"; SOLVER.AMP.ENABLED True in every shipped
// yaml; engine/train_loop.py:709 `with autocast():`): both operands rounded to fp16 (round-to-nearest-even, as `.half()`), products
// accumulated in f32 by ONE v_mfma_f32_32x32x16_f16 per tile and k-step, f32 out.  This is NOT bit-for-bit what torch.autocast
// computes: real autocast also rounds every OUTPUT to fp16 (so FrozenBN, residual adds and ReLUs of the trunk run on fp16 tensors)
// and runs the attention bmm's in fp16; here activations stay f32 between layers.  The AMP parity tests therefore pin this mode
// against the oracle's restatement of THIS definition (oracle_np.AMP), not against a recording of torch.autocast.  Opt-in, for the modules
// autocast runs in fp16 -- the R50 trunk, the video decoder's linear layers, the mask-logit einsum; the pixel decoder and the
// matcher force fp32 in the reference (msdeformattn.py:314, matcher.py:266-268) and stay on the split-fp16 x3 kernels -- reported
// separately by bench.py (`amp`): the headline metric is fp32-class.
//
// Same NT-GEMM / implicit-GEMM-conv contract and epilogue as gemm.hip (C = act(A.B^T * scale + bias + res)).  Tile 128 x 128 x 64,
// four waves of 64 x 64 (2 x 2 MFMA tiles x 4 k-steps), two workgroups per CU.  Staging: global f32 (16-B loads, hardware
// bounds) -> registers (one k-tile ahead) -> v_cvt (RTN) -> LDS rows of [32 words = 64 halves | 4 pad] (the 36-word row of
// gemm_bf16.hip: conflict-free ds_read_b128 fragment reads), double buffered, one barrier per k-tile.
#include "common.h"
#include "gemm_params.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int BM = 128, BN = 128, BK = 64, ROWW = 36, SLOTS = 8;   // 256 threads: 16 float4 columns x 16 rows per pass, 8 passes

__device__ __forceinline__ u32x2 cvt4_rtn(const f32x4 v)
{
    const f16x2v a = {(_Float16)v[0], (_Float16)v[1]}, b = {(_Float16)v[2], (_Float16)v[3]};      // v_cvt_f16_f32: round to nearest even
    u32x2 r;
    r[0] = __builtin_bit_cast(unsigned int, a);
    r[1] = __builtin_bit_cast(unsigned int, b);
    return r;
}

template <bool CONV>
__global__ __launch_bounds__(256, 2) void gemm_f16_amp_kernel(GemmParams p)
{
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
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, within = bid / 8;       // an XCD sweeps a contiguous band of tiles
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    }
    const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int bz = blockIdx.y;
    const float *A = p.A + (long)bz * p.sA;
    const float *B = p.B + (long)bz * p.sB;
    float *C = p.C + (long)bz * p.sC;

    const int c16 = tid & 15, g = tid >> 4;
    constexpr unsigned int OOB = 0xFFFFFFF0u;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A), 0, (int)p.bytesA, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(B), 0, (int)p.bytesB, 0x00020000);
    unsigned int a_off[SLOTS], b_off[SLOTS], a_bad[SLOTS], b_bad[SLOTS];
    int a_iy0[SLOTS], a_ix0[SLOTS];
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
        const int m = m0 + g + 16 * i;
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
        const int n = n0 + g + 16 * i;
        b_off[i] = n < p.N ? (unsigned int)((long)n * p.ldb * 4L) : 0u;
        a_bad[i] = ok ? 0u : OOB;
        b_bad[i] = n < p.N ? 0u : OOB;
    }
    f32x4 ra[SLOTS], rb[SLOTS];
    auto load_tile = [&](int kt) {
        const int k = kt * BK + c16 * 4;
        const unsigned int kmask = (unsigned int)((p.K - 1 - k) >> 31) & OOB;   // all-OOB bits when k >= K: the hardware drops the load (zeros)
        int kh = 0, kw = 0, ci = 0;
        if (CONV) {
            const int tap = k / p.Cin;
            ci = k - tap * p.Cin;
            kh = tap / p.KW;
            kw = tap - kh * p.KW;
        }
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            unsigned int off;
            if (CONV) {
                const int iy = a_iy0[i] + kh, ix = a_ix0[i] + kw;
                const unsigned int tmask = (unsigned int)(((iy | ix | (p.Hin - 1 - iy) | (p.Win - 1 - ix)) >> 31)) & OOB;   // tap outside the image
                off = (a_off[i] + (unsigned int)(((iy * p.Win + ix) * p.Cin + ci) * 4)) | tmask | kmask;
            } else {
                off = (a_off[i] + (unsigned int)(k * 4)) | a_bad[i] | kmask;
            }
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)off, 0, 0));
            rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)((b_off[i] + (unsigned int)(k * 4)) | b_bad[i] | kmask), 0, 0));
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            *reinterpret_cast<u32x2 *>(&As[(buf * BM + g + 16 * i) * ROWW + c16 * 2]) = cvt4_rtn(ra[i]);
            *reinterpret_cast<u32x2 *>(&Bs[(buf * BN + g + 16 * i) * ROWW + c16 * 2]) = cvt4_rtn(rb[i]);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (p.K + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const unsigned int *as = &As[(cur * BM + wm * 64 + l32) * ROWW + 4 * h];
        const unsigned int *bs = &Bs[(cur * BN + wn * 64 + l32) * ROWW + 4 * h];
#pragma unroll
        for (int s = 0; s < 4; ++s) {                     // k-step s: halves 16 s .. 16 s + 15 of the row, lane half h takes 8 of them
            f16x8 bh[2], ah[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                bh[t] = *reinterpret_cast<const f16x8 *>(bs + t * 32 * ROWW + 8 * s);
                ah[t] = *reinterpret_cast<const f16x8 *>(as + t * 32 * ROWW + 8 * s);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
    }
    const float *res = p.res ? p.res + (long)bz * p.sR : nullptr;
    if (((p.N | p.ldc | p.ldr | p.res_cols) & 3) == 0) {
        // vector epilogue: every wave parks its 64 x 64 tile in LDS (the operand ring is dead) and streams it out row-wise, 16 B per lane
        float *ep = reinterpret_cast<float *>(lds) + wave * 64 * 68;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) ep[(tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 68 + tn * 32 + l32] = acc[tm][tn][r];
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
                float v = acc[tm][tn][r] * sc + bi;
                if (res && col < p.res_cols) v += res[(long)(p.res_rows ? row % p.res_rows : row) * p.ldr + col];
                if (p.relu) v = fmaxf(v, 0.f);
                C[(long)row * p.ldc + col] = v;
            }
        }
    }
}

template <bool CONV>
int launch_amp(GemmParams p, int batch, hipStream_t st)
{
    const long bA = CONV ? (long)(p.M / ((long)p.Hout * p.Wout)) * p.Hin * p.Win * p.Cin * 4L : ((long)(p.M - 1) * p.lda + p.K) * 4L;
    const long bB = ((long)(p.N - 1) * p.ldb + p.K) * 4L;
    if (bA > 0xFFFFFF00L || bB > 0xFFFFFF00L) return S2D_ERR_ARG;   // 32-bit buffer offsets
    p.bytesA = (unsigned int)bA; p.bytesB = (unsigned int)bB;
    const size_t lds = sizeof(unsigned int) * 2 * (BM + BN) * ROWW;   // 73 728 B: also holds the four 64 x 68 epilogue tiles (69 632 B)
    static S2dDevOnce attr_set;
    if (!attr_set.done()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_f16_amp_kernel<CONV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return S2D_ERR_LAUNCH;
        attr_set.mark();
    }
    const int nwg = cdiv(p.M, BM) * cdiv(p.N, BN);
    hipLaunchKernelGGL((gemm_f16_amp_kernel<CONV>), dim3(nwg, batch), dim3(256), lds, st, p);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

}  // namespace

extern "C" {

int s2d_gemm_nt_amp_f32(const float *A, const float *B, float *C, int M, int N, int K, long lda, long ldb, long ldc, int batch,
                        long strideA, long strideB, long strideC, const float *scale, const float *bias, const float *res, long ldr,
                        long strideR, int res_rows, int res_cols, int relu, hipStream_t stream)
{
    if (M <= 0 || N <= 0 || batch <= 0) return S2D_OK;
    if (K <= 0 || (K & 3) || (lda & 3) || (ldb & 3) || res_rows < 0 || res_cols < 0 || res_cols > N || (res_cols & 3)) return S2D_ERR_ARG;
    GemmParams p{};
    p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.sA = strideA; p.sB = strideB; p.sC = strideC;
    p.scale = scale; p.bias = bias; p.res = res; p.ldr = res ? ldr : N; p.sR = strideR; p.relu = relu;
    p.res_rows = res_rows; p.res_cols = res_cols > 0 ? res_cols : N;
    return launch_amp<false>(p, batch, stream);
}

int s2d_conv2d_nhwc_amp_f32(const float *x, const float *w, float *y, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                            int pad, const float *scale, const float *bias, const float *res, int relu, hipStream_t stream)
{
    if (N <= 0) return S2D_OK;
    if ((Cin & 3) || stride <= 0 || pad < 0) return S2D_ERR_ARG;
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return S2D_ERR_ARG;
    GemmParams p{};
    p.A = x; p.B = w; p.C = y;
    p.M = N * Ho * Wo; p.N = Cout; p.K = KH * KW * Cin;
    p.lda = 0; p.ldb = p.K; p.ldc = Cout;
    p.scale = scale; p.bias = bias; p.res = res; p.ldr = Cout; p.relu = relu;
    p.res_rows = 0; p.res_cols = Cout;
    p.Hin = H; p.Win = W; p.Cin = Cin; p.Hout = Ho; p.Wout = Wo; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad;
    if ((long)p.M * Cout * 4L > 0x7FFFFFFF00L) return S2D_ERR_ARG;
    return launch_amp<true>(p, 1, stream);
}

}  // extern "C"
