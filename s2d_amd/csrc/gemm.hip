// Dense contractions of the S2D forward on the fp32-input MFMA (exact f32 FMA chain):
//   C[M,N] = epilogue( A[M,K] * B[N,K]^T )            ("NT" GEMM, both operands K-contiguous)
// used for every nn.Linear, every 1x1 conv in NHWC, the mask-logit contraction
// (video_mask2former_transformer_decoder.py:455) and, with the implicit-im2col A loader, the
// 3x3 / 7x7 convolutions of the R50 trunk and the FPN (NHWC activations, [Cout][kh][kw][Cin] weights).
//
// Tile: 128x128x32 per 256-thread workgroup; 4 waves as 2x2, each wave 64x64 = 2x2 MFMA 32x32x2 tiles
// (64 accumulator VGPRs).  Operands are staged global -> registers -> LDS ([row][k], row stride 36 floats:
// 16-byte aligned and conflict-free for ds_read_b128 / ds_write_b128), double buffered so the next tile's
// global loads fly under the current tile's 64 MFMAs per wave.  Each lane feeds four consecutive MFMAs
// from one ds_read_b128: lane half h supplies k = 8g+4h+j for MFMA j of group g (A and B use the same
// map, so every k is visited once).
#include "common.h"
#include "gemm_params.h"
#include "dropout.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32, LDS_STRIDE = BK + 4;

template <bool CONV>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmParams p)
{
    __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * LDS_STRIDE];
    float *As = lds;                          // [2][BM][LDS_STRIDE]
    float *Bs = lds + 2 * BM * LDS_STRIDE;    // [2][BN][LDS_STRIDE]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l32 = lane & 31, h = lane >> 5;

    // XCD-aware tile order: consecutive tile ids (which share an A row panel) stay on one XCD.
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

    // per-thread staging coordinates: 4 float4 of A and 4 of B per k-tile
    const int c4 = tid & 7;       // which float4 along k
    const int r0 = tid >> 3;      // rows r0 + 32*i
    const float *a_row[4];
    int a_iy0[4], a_ix0[4];
    bool a_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + r0 + 32 * i;
        a_ok[i] = m < p.M;
        if (CONV) {
            const int mm = a_ok[i] ? m : 0;
            const int ox = mm % p.Wout, t = mm / p.Wout, oy = t % p.Hout, n = t / p.Hout;
            a_iy0[i] = oy * p.stride - p.pad;
            a_ix0[i] = ox * p.stride - p.pad;
            a_row[i] = A + (long)n * p.Hin * p.Win * p.Cin;
        } else {
            a_row[i] = A + (long)(a_ok[i] ? m : 0) * p.lda;
            a_iy0[i] = a_ix0[i] = 0;
        }
    }
    const float *b_row[4];
    bool b_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + r0 + 32 * i;
        b_ok[i] = n < p.N;
        b_row[i] = B + (long)(b_ok[i] ? n : 0) * p.ldb;
    }

    f32x4 ra[4], rb[4];
    auto load_tile = [&](int kt) {
        const int k = kt * BK + c4 * 4;
        const bool kok = k < p.K;
        int kh = 0, kw = 0, ci = 0;
        if (CONV) {
            const int tap = k / p.Cin;
            ci = k - tap * p.Cin;
            kh = tap / p.KW;
            kw = tap - kh * p.KW;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (CONV) {
                const int iy = a_iy0[i] + kh, ix = a_ix0[i] + kw;
                if (a_ok[i] && kok && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win)
                    v = *reinterpret_cast<const f32x4 *>(a_row[i] + ((long)iy * p.Win + ix) * p.Cin + ci);
            } else {
                if (a_ok[i] && kok) v = *reinterpret_cast<const f32x4 *>(a_row[i] + k);
            }
            ra[i] = v;
            f32x4 w = {0.f, 0.f, 0.f, 0.f};
            if (b_ok[i] && kok) w = *reinterpret_cast<const f32x4 *>(b_row[i] + k);
            rb[i] = w;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f32x4 *>(&As[(buf * BM + r0 + 32 * i) * LDS_STRIDE + c4 * 4]) = ra[i];
            *reinterpret_cast<f32x4 *>(&Bs[(buf * BN + r0 + 32 * i) * LDS_STRIDE + c4 * 4]) = rb[i];
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
        const float *as = &As[(cur * BM + wm * 64 + l32) * LDS_STRIDE + 4 * h];
        const float *bs = &Bs[(cur * BN + wn * 64 + l32) * LDS_STRIDE + 4 * h];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 a0 = *reinterpret_cast<const f32x4 *>(as + 8 * g);
            const f32x4 a1 = *reinterpret_cast<const f32x4 *>(as + 32 * LDS_STRIDE + 8 * g);
            const f32x4 b0 = *reinterpret_cast<const f32x4 *>(bs + 8 * g);
            const f32x4 b1 = *reinterpret_cast<const f32x4 *>(bs + 32 * LDS_STRIDE + 8 * g);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[1][1], 0, 0, 0);
            }
        }
        if (kt + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
    }

    // epilogue: acc[tm][tn][r] is C[row = (r&3) + 8*(r>>2) + 4*h][col = l32] of its 32x32 tile
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

int g_dense_mode = 2;   // 0: fp32-input MFMA (exact f32 FMA chain); 1: split-bf16 x3; 2: split-fp16 x3 (gemm_bf16.hip)

int launch(const GemmParams &p, bool conv, int batch, hipStream_t st)
{
    if (p.M <= 0 || p.N <= 0 || batch <= 0) return S2D_OK;
    if (p.K <= 0 || (p.K & 3) || (p.lda & 3) || (p.ldb & 3)) return S2D_ERR_ARG;
    if (p.gate && g_dense_mode != 2) return S2D_ERR_ARG;    // the gate lives in the split-fp16 kernels' vector epilogues
    if (g_dense_mode >= 1) {
        GemmParams q = p;      // the pre-split image of a static B was made for the mode in force (s2d_split_weights_f16)
        return s2d_launch_gemm_bf16x3(q, conv, batch, st, g_dense_mode == 2);
    }
    const int nwg = cdiv(p.M, BM) * cdiv(p.N, BN);
    dim3 grid(nwg, batch);
    if (conv)
        hipLaunchKernelGGL(gemm_nt_kernel<true>, grid, dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL(gemm_nt_kernel<false>, grid, dim3(256), 0, st, p);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

}  // namespace

extern "C" {

// C[b][M,N] = act( (A[b][M,K] * B[b][N,K]^T) * scale[N] + bias[N] + res[b][M,N] )
int s2d_gemm_nt_f32(const float *A, const float *B, float *C, int M, int N, int K, long lda, long ldb, long ldc,
                    int batch, long strideA, long strideB, long strideC, const float *scale, const float *bias,
                    const float *res, long ldr, long strideR, int res_rows, int res_cols, int relu, const void *B_split,
                    hipStream_t stream)
{
    if (res_rows < 0 || res_cols < 0 || res_cols > N) return S2D_ERR_ARG;
    GemmParams p{};
    p.Bsplit = reinterpret_cast<const unsigned int *>(B_split);
    p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.sA = strideA; p.sB = strideB; p.sC = strideC;
    p.scale = scale; p.bias = bias; p.res = res; p.ldr = ldr; p.sR = strideR; p.relu = relu;
    p.res_rows = res_rows; p.res_cols = res_cols > 0 ? res_cols : N;
    return launch(p, false, batch, stream);
}

// C[M,N] = gate > 0 ? (A . B^T + res) * gate_scale : 0: a dgrad GEMM with the ReLU (and dropout) gate of the layer it
// differentiates folded into its epilogue (backward.py input_grad)
int s2d_gemm_nt_gate_f32(const float *A, const float *B, float *C, int M, int N, int K, long lda, long ldb, long ldc, const float *scale,
                         const float *res, long ldr, const float *gate, long ldg, float gate_scale, const void *B_split, hipStream_t stream)
{
    if (!gate) return S2D_ERR_ARG;
    GemmParams p{};
    p.Bsplit = reinterpret_cast<const unsigned int *>(B_split);
    p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.scale = scale; p.res = res; p.ldr = res ? ldr : N; p.res_cols = N;
    p.gate = gate; p.ldg = ldg; p.gate_scale = gate_scale;
    return launch(p, false, 1, stream);
}

// the same contraction with A handed over pre-split (s2d_split_weights_f16 layout over A's rows) next to static pre-split weights
int s2d_gemm_nt_presplit_f32(const void *A_split, const float *B, float *C, int M, int N, int K, long ldb, long ldc, const float *bias,
                             const float *res, long ldr, int res_rows, int res_cols, int relu, const void *B_split, hipStream_t stream)
{
    if (!A_split || !B_split || res_rows < 0 || res_cols < 0 || res_cols > N || g_dense_mode != 2) return S2D_ERR_ARG;
    GemmParams p{};
    p.Asplit = reinterpret_cast<const unsigned int *>(A_split);
    p.Bsplit = reinterpret_cast<const unsigned int *>(B_split);
    p.A = reinterpret_cast<const float *>(A_split); p.B = B; p.C = C; p.M = M; p.N = N; p.K = K;
    p.lda = K; p.ldb = ldb; p.ldc = ldc;
    p.bias = bias; p.res = res; p.ldr = res ? ldr : N; p.relu = relu;
    p.res_rows = res_rows; p.res_cols = res_cols > 0 ? res_cols : N;
    return launch(p, false, 1, stream);
}

// the same contraction with dropout fused into the epilogue (see include/s2d_hip.h; dropout.h for the mask definition)
int s2d_gemm_nt_dropout_f32(const float *A, const float *B, float *C, int M, int N, int K, long lda, long ldb, long ldc,
                            const float *bias, const float *res, long ldr, int relu, const void *B_split, float p,
                            uint64_t seed, unsigned site, unsigned row0, hipStream_t stream)
{
    if (!(p >= 0.f) || p >= 1.f) return S2D_ERR_ARG;
    GemmParams q{};
    q.Bsplit = reinterpret_cast<const unsigned int *>(B_split);
    q.A = A; q.B = B; q.C = C; q.M = M; q.N = N; q.K = K;
    q.lda = lda; q.ldb = ldb; q.ldc = ldc;
    q.bias = bias; q.res = res; q.ldr = res ? ldr : N; q.relu = relu; q.res_cols = N;
    const unsigned thresh = s2d_dropout_thresh(p);
    if (thresh == 0) return launch(q, false, 1, stream);        // p rounds to zero: the plain contraction
    if (g_dense_mode != 2) return S2D_ERR_ARG;
    q.drop_thresh = thresh;
    q.drop_scale = s2d_dropout_scale(thresh);
    q.drop_k0 = (unsigned)(seed & 0xFFFFFFFFull); q.drop_k1 = (unsigned)(seed >> 32); q.drop_stream = site; q.drop_row0 = row0;
    return launch(q, false, 1, stream);
}

// NHWC convolution as implicit GEMM. x [N,H,W,Cin] (Cin % 4 == 0), w [Cout][KH][KW][Cin],
// y [N,Ho,Wo,Cout] = act( conv(x,w) * scale[Cout] + bias[Cout] + res[N,Ho,Wo,Cout] )
int s2d_conv2d_nhwc_f32(const float *x, const float *w, float *y, int N, int H, int W, int Cin, int Cout, int KH,
                        int KW, int stride, int pad, const float *scale, const float *bias, const float *res,
                        int relu, const void *w_split, hipStream_t stream)
{
    if (Cin & 3) return S2D_ERR_ARG;
    GemmParams p{};
    p.Bsplit = reinterpret_cast<const unsigned int *>(w_split);
    p.Hin = H; p.Win = W; p.Cin = Cin; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad;
    p.Hout = (H + 2 * pad - KH) / stride + 1;
    p.Wout = (W + 2 * pad - KW) / stride + 1;
    p.A = x; p.B = w; p.C = y;
    p.M = N * p.Hout * p.Wout; p.N = Cout; p.K = KH * KW * Cin;
    p.lda = 4; p.ldb = p.K; p.ldc = Cout;
    p.scale = scale; p.bias = bias; p.res = res; p.ldr = Cout; p.relu = relu; p.res_cols = Cout;
    return launch(p, true, 1, stream);
}

// the convolution with the gate epilogue of s2d_gemm_nt_gate_f32 (dgrad of a 3x3 convolution whose input came out of a ReLU):
// y = gate > 0 ? conv(x, w) * scale : 0
int s2d_conv2d_nhwc_gate_f32(const float *x, const float *w, float *y, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                             int pad, const float *scale, const float *gate, float gate_scale, const void *w_split, hipStream_t stream)
{
    if ((Cin & 3) || !gate) return S2D_ERR_ARG;
    GemmParams p{};
    p.Bsplit = reinterpret_cast<const unsigned int *>(w_split);
    p.Hin = H; p.Win = W; p.Cin = Cin; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad;
    p.Hout = (H + 2 * pad - KH) / stride + 1;
    p.Wout = (W + 2 * pad - KW) / stride + 1;
    p.A = x; p.B = w; p.C = y;
    p.M = N * p.Hout * p.Wout; p.N = Cout; p.K = KH * KW * Cin;
    p.lda = 4; p.ldb = p.K; p.ldc = Cout;
    p.scale = scale; p.ldr = Cout; p.res_cols = Cout;
    p.gate = gate; p.ldg = Cout; p.gate_scale = gate_scale;
    return launch(p, true, 1, stream);
}

}  // extern "C"

extern "C" long s2d_split_weights_words(int N, int K) { return (long)N * ((K + 31) / 32) * 32; }

extern "C" int s2d_split_weights_f16(const float *W, int N, int K, long ldw, void *out, hipStream_t stream)
{
    if (N < 0 || K <= 0 || ldw < K) return S2D_ERR_ARG;
    // the image follows the dense mode in force: fp16 hi / scaled lo (mode 2) or bf16 hi / lo (mode 1); callers key their
    // caches by the mode (ops._static_split)
    return s2d_split_weights_launch(W, N, K, ldw, reinterpret_cast<unsigned int *>(out), stream, g_dense_mode == 1);
}

extern "C" int s2d_abi_version(void) { return 11; }

extern "C" int s2d_set_dense_mode(int mode)
{
    if (mode < 0 || mode > 2) return S2D_ERR_ARG;
    g_dense_mode = mode;
    return S2D_OK;
}
