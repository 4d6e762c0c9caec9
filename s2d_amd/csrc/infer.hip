// Eval-side step after the path: what KDVideoMaskFormer.forward does with the network outputs when
// not training (model_training/mask2former_video/kd_video_maskformer_model.py:327-356) and inference_video (:530-610;
// video_maskformer_model.py:298-360 is the same code for the non-KD model):
//
//   scores = softmax(class logits)[:, :-1]; sorted top-K over the flattened [Q*C] scores;           (:532-538)
//   masks  = bilinear(mask logits -> padded input size) [:341-346], crop to the unpadded size [:545],
//            bilinear -> output size [:546-548], > 0 [:550];
//   optional greedy same-label mask-NMS with IoU from pairwise sum(a & b) / sum(a | b)               (:552-583)
//
// The reference materialises the [Q,T,Hp,Wp] fp32 upsample of ALL queries (0.94 GB per 10 frames at 720p, Q = 100), then
// the [K,T,oh,ow] fp32 resize, and runs the NMS as O(K^2) pairs of full-tensor reductions with a device->host sync each.
// Here: one selection kernel; the K selected queries' logits are gathered from the pixel-major maps into small planes;
// one kernel evaluates both bilinear stages per output pixel (16 low-resolution taps, the same fp32 expression tree as
// two consecutive F.interpolate calls) and writes the boolean mask once, as bytes (the tensor the caller gets) and as
// bit words; a tiled popcount kernel turns the bit words into the K x K intersection counts, from which the host runs
// the greedy loop without touching the masks again.
#include "common.h"

namespace {

constexpr int SEL_THREADS = 1024;

// one workgroup: softmax per query, then rank every flat score (ties -> lower flat index comes first)
__global__ __launch_bounds__(SEL_THREADS) void infer_select_kernel(const float *__restrict__ cls, int Q, int C, int K,
                                                                   float *__restrict__ scores, int *__restrict__ query,
                                                                   int *__restrict__ label)
{
    extern __shared__ float sc[];                       // [Q*C]
    const int n = Q * C;
    for (int q = threadIdx.x; q < Q; q += SEL_THREADS) {
        const float *l = cls + (long)q * (C + 1);
        float mx = l[0];
        for (int c = 1; c <= C; ++c) mx = fmaxf(mx, l[c]);
        float sum = 0.f;
        for (int c = 0; c <= C; ++c) sum += expf(l[c] - mx);
        for (int c = 0; c < C; ++c) sc[q * C + c] = expf(l[c] - mx) / sum;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += SEL_THREADS) {
        const float s = sc[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const float o = sc[j];
            rank += (o > s) || (o == s && j < i);
        }
        if (rank < K) { scores[rank] = s; query[rank] = i / C; label[rank] = i % C; }
    }
}

// planes[k][pix] = ml[pix][query[k]]   (pixel-major [T*hm*wm, ldq] -> K query-major planes), transposed through LDS: a
// workgroup takes 64 pixels; lanes run over k within a pixel's 512-B row when reading and over pixels when writing.
constexpr int GP = 64, GK = 128;
__global__ __launch_bounds__(256) void infer_gather_kernel(const float *__restrict__ ml, int ldq, long npix,
                                                           const int *__restrict__ query, int K, float *__restrict__ planes)
{
    __shared__ float tile[GK][GP + 1];
    __shared__ int qs[GK];
    const long pix0 = (long)blockIdx.x * GP;
    const int np = npix - pix0 < GP ? (int)(npix - pix0) : GP;
    for (int k0 = 0; k0 < K; k0 += GK) {
        const int kc = K - k0 < GK ? K - k0 : GK;
        __syncthreads();
        if ((int)threadIdx.x < kc) qs[threadIdx.x] = query[k0 + threadIdx.x];
        __syncthreads();
        for (int idx = threadIdx.x; idx < np * kc; idx += 256) {
            const int pp = idx / kc, k = idx - pp * kc;
            tile[k][pp] = ml[(pix0 + pp) * ldq + qs[k]];
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < kc * GP; idx += 256) {
            const int k = idx >> 6, pp = idx & 63;
            if (pp < np) planes[(long)(k0 + k) * npix + pix0 + pp] = tile[k][pp];
        }
    }
}

struct ResizeParams {
    const float *planes;       // [K][T][hm][wm]
    int T, hm, wm, Hp, Wp, ih, iw, oh, ow;
    long N;                    // T*oh*ow, elements of one mask
    long words;                // bit words per mask = ceil(N / 32)
    float s1y, s1x, s2y, s2x;  // hm/Hp, wm/Wp, ih/oh, iw/ow as float (area_pixel_compute_scale)
    int same;                  // output size == unpadded size: the second resize is the identity
    uint8_t *masks;            // [K][N]
    uint32_t *bits;            // [K][words] or null
};

// source index / weights of F.interpolate(mode="bilinear", align_corners=False)
__device__ __forceinline__ void src_tap(float scale, int dst, int in_size, int &i0, int &i1, float &l0, float &l1)
{
    float s = scale * (dst + 0.5f) - 0.5f;
    if (s < 0.f) s = 0.f;
    i0 = (int)s;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = s - i0;
    l0 = 1.f - l1;
}

// value of the padded-size upsample (:341-346) from its two low-resolution rows and their weights
__device__ __forceinline__ float stage1(const float *__restrict__ r0, const float *__restrict__ r1, float hy, float ly, int x0,
                                        int x1, float hx, float lx)
{
    return hy * (hx * r0[x0] + lx * r0[x1]) + ly * (hx * r1[x0] + lx * r1[x1]);
}

// A thread owns 4 consecutive elements of one mask (flat index over [T][oh][ow]); 8 lanes make one 32-bit word.  The
// vertical taps are formed once per thread and again only when its 4 elements wrap into the next row.
template <bool SAME>
__global__ __launch_bounds__(256) void infer_resize_kernel(ResizeParams p)
{
    const int k = blockIdx.y;
    const long i0 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    unsigned int nib = 0u;
    if (i0 < p.N) {
        int t, y, x;
        if (p.N < (1L << 31)) {                          // 32-bit index arithmetic (the usual case)
            const unsigned int frame = (unsigned int)p.oh * (unsigned int)p.ow, i = (unsigned int)i0;
            const unsigned int tt = i / frame, r = i - tt * frame, yy = r / (unsigned int)p.ow;
            t = (int)tt; y = (int)yy; x = (int)(r - yy * (unsigned int)p.ow);
        } else {
            const long frame = (long)p.oh * p.ow;
            t = (int)(i0 / frame);
            const long r = i0 - (long)t * frame;
            y = (int)(r / p.ow); x = (int)(r - (long)y * p.ow);
        }
        // stage-1 rows (a: under stage-2 row Y0, b: under Y1; SAME uses a only) and the stage-2 vertical weights
        const float *a0 = nullptr, *a1 = nullptr, *b0 = nullptr, *b1 = nullptr;
        float hya = 0.f, lya = 0.f, hyb = 0.f, lyb = 0.f, HY = 1.f, LY = 0.f;
#define S2D_SET_ROW()                                                                           \
        do {                                                                                    \
            const float *pl = p.planes + ((long)k * p.T + t) * p.hm * p.wm;                     \
            int q0, q1;                                                                         \
            if (SAME) {                                                                         \
                src_tap(p.s1y, y, p.hm, q0, q1, hya, lya);                                      \
                a0 = pl + q0 * p.wm; a1 = pl + q1 * p.wm;                                       \
            } else {                                                                            \
                int Y0, Y1;                                                                     \
                src_tap(p.s2y, y, p.ih, Y0, Y1, HY, LY);                                        \
                src_tap(p.s1y, Y0, p.hm, q0, q1, hya, lya);                                     \
                a0 = pl + q0 * p.wm; a1 = pl + q1 * p.wm;                                       \
                src_tap(p.s1y, Y1, p.hm, q0, q1, hyb, lyb);                                     \
                b0 = pl + q0 * p.wm; b1 = pl + q1 * p.wm;                                       \
            }                                                                                   \
        } while (0)
        S2D_SET_ROW();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (i0 + j < p.N) {
                float v;
                int xa0, xa1;
                float ha, la;
                if (SAME) {
                    src_tap(p.s1x, x, p.wm, xa0, xa1, ha, la);
                    v = stage1(a0, a1, hya, lya, xa0, xa1, ha, la);
                } else {
                    int X0, X1, xb0, xb1;
                    float HX, LX, hb, lb;
                    src_tap(p.s2x, x, p.iw, X0, X1, HX, LX);
                    src_tap(p.s1x, X0, p.wm, xa0, xa1, ha, la);
                    src_tap(p.s1x, X1, p.wm, xb0, xb1, hb, lb);
                    v = HY * (HX * stage1(a0, a1, hya, lya, xa0, xa1, ha, la) + LX * stage1(a0, a1, hya, lya, xb0, xb1, hb, lb)) +
                        LY * (HX * stage1(b0, b1, hyb, lyb, xa0, xa1, ha, la) + LX * stage1(b0, b1, hyb, lyb, xb0, xb1, hb, lb));
                }
                nib |= (v > 0.f ? 1u : 0u) << j;
            }
            if (++x == p.ow) {
                x = 0;
                if (++y == p.oh) { y = 0; ++t; }
                if (j < 3 && i0 + j + 1 < p.N) S2D_SET_ROW();
            }
        }
#undef S2D_SET_ROW
        uint8_t *mo = p.masks + (long)k * p.N + i0;
        if (i0 + 4 <= p.N && (reinterpret_cast<uintptr_t>(mo) & 3) == 0)
            *reinterpret_cast<uint32_t *>(mo) = (nib & 1u) | ((nib & 2u) << 7) | ((nib & 4u) << 14) | ((nib & 8u) << 21);
        else
            for (int j = 0; j < 4 && i0 + j < p.N; ++j) mo[j] = (nib >> j) & 1u;
    }
    if (p.bits) {                                       // whole waves reach this point: lanes past N carry 0
        unsigned int w = nib << (4 * (threadIdx.x & 7));
        w |= __shfl_xor(w, 1, 64);
        w |= __shfl_xor(w, 2, 64);
        w |= __shfl_xor(w, 4, 64);
        const long wi = i0 >> 5;
        if ((threadIdx.x & 7) == 0 && wi < p.words) p.bits[(long)k * p.words + wi] = w;
    }
}

// inter[i][j] += sum over a chunk of words of popcount(bits[i] & bits[j]) for an 8 x 8 tile of mask pairs (tile row <=
// tile column; the diagonal holds the areas).  Integer atomics: the result does not depend on the order.
constexpr int PT = 8;
__global__ __launch_bounds__(256) void pair_count_kernel(const uint32_t *__restrict__ bits, int K, long words, int ntile,
                                                         unsigned long long *__restrict__ inter)
{
    __shared__ unsigned int red[4][PT * PT];
    // blockIdx.y enumerates the tile pairs (ti <= tj)
    int ti = 0, rem = blockIdx.y;
    while (rem >= ntile - ti) { rem -= ntile - ti; ++ti; }
    const int tj = ti + rem;
    unsigned int acc[PT][PT];
#pragma unroll
    for (int a = 0; a < PT; ++a)
#pragma unroll
        for (int b = 0; b < PT; ++b) acc[a][b] = 0u;
    const long per = (words + gridDim.x - 1) / gridDim.x;
    const long w0 = (long)blockIdx.x * per, w1 = w0 + per < words ? w0 + per : words;
    for (long w = w0 + threadIdx.x; w < w1; w += 256) {
        unsigned int va[PT], vb[PT];
#pragma unroll
        for (int a = 0; a < PT; ++a) {
            const int i = ti * PT + a, j = tj * PT + a;
            va[a] = i < K ? bits[(long)i * words + w] : 0u;
            vb[a] = j < K ? bits[(long)j * words + w] : 0u;
        }
#pragma unroll
        for (int a = 0; a < PT; ++a)
#pragma unroll
            for (int b = 0; b < PT; ++b) acc[a][b] += __popc(va[a] & vb[b]);
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < PT; ++a)
#pragma unroll
        for (int b = 0; b < PT; ++b) {
            unsigned int v = acc[a][b];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) red[wv][a * PT + b] = v;
        }
    __syncthreads();
    if (threadIdx.x < PT * PT) {
        const int a = threadIdx.x / PT, b = threadIdx.x % PT;
        const int i = ti * PT + a, j = tj * PT + b;
        const unsigned long long v = (unsigned long long)red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (i < K && j < K && v) {
            atomicAdd(&inter[(long)i * K + j], v);
            if (ti != tj) atomicAdd(&inter[(long)j * K + i], v);
        }
    }
}


// u8 mask planes [K][n] (0 / non-0) -> bit words [K][ceil(n / 32)] (flat index i -> word i / 32, bit i % 32; tail bits zero):
// the input format of pair_count_kernel, for masks that exist as byte planes (the KD pseudo targets, DISTILLATION_NMS)
__global__ __launch_bounds__(256) void pack_bits_kernel(const uint8_t *__restrict__ masks, long n, long words, uint32_t *__restrict__ bits)
{
    const int k = blockIdx.y;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;                  // element index; a wave covers two words
    const bool v = i < n && masks[(long)k * n + i] != 0;
    const unsigned long long b = __ballot(v);
    const int lane = threadIdx.x & 63;
    const long w = i >> 5;
    if ((lane & 31) == 0 && w < words) bits[(long)k * words + w] = (uint32_t)(lane ? (b >> 32) : (b & 0xFFFFFFFFull));
}

}  // namespace

extern "C" {

int s2d_infer_select_f32(const float *cls_logits, int Q, int C, int K, float *scores, int *query, int *label, hipStream_t stream)
{
    if (Q < 1 || C < 1 || K < 1 || K > Q * C) return S2D_ERR_ARG;
    const size_t lds = sizeof(float) * (size_t)Q * C;
    if (lds > 64 * 1024) return S2D_ERR_ARG;
    hipLaunchKernelGGL(infer_select_kernel, dim3(1), dim3(SEL_THREADS), lds, stream, cls_logits, Q, C, K, scores, query, label);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

long s2d_infer_workspace_floats(int K, int T, int hm, int wm) { return (long)K * T * hm * wm; }

long s2d_mask_bit_words(int T, int oh, int ow) { return ((long)T * oh * ow + 31) / 32; }

int s2d_infer_masks_u8(const float *mask_logits, int ldq, int T, int hm, int wm, int Hp, int Wp, int ih, int iw, int oh, int ow,
                       const int *query, int K, float *workspace, uint8_t *masks, uint32_t *bits, hipStream_t stream)
{
    if (K < 0 || T < 1 || hm < 1 || wm < 1 || ih < 1 || iw < 1 || ih > Hp || iw > Wp || oh < 1 || ow < 1) return S2D_ERR_ARG;
    if (K == 0) return S2D_OK;
    const long npix = (long)T * hm * wm;
    hipLaunchKernelGGL(infer_gather_kernel, dim3(cdiv(npix, GP)), dim3(256), 0, stream, mask_logits, ldq, npix, query, K, workspace);
    ResizeParams p;
    p.planes = workspace;
    p.T = T; p.hm = hm; p.wm = wm; p.Hp = Hp; p.Wp = Wp; p.ih = ih; p.iw = iw; p.oh = oh; p.ow = ow;
    p.N = (long)T * oh * ow;
    p.words = (p.N + 31) / 32;
    p.s1y = (float)hm / Hp; p.s1x = (float)wm / Wp; p.s2y = (float)ih / oh; p.s2x = (float)iw / ow;
    p.same = (ih == oh && iw == ow) ? 1 : 0;
    p.masks = masks; p.bits = bits;
    const long nthreads = (p.words * 32 + 3) / 4;       // whole words: the tail lanes write the zero padding bits
    if (p.same) hipLaunchKernelGGL(infer_resize_kernel<true>, dim3(cdiv(nthreads, 256), K), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(infer_resize_kernel<false>, dim3(cdiv(nthreads, 256), K), dim3(256), 0, stream, p);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

int s2d_pack_mask_bits_u8(const uint8_t *masks, int K, long n, uint32_t *bits, hipStream_t stream)
{
    if (K < 0 || n <= 0) return S2D_ERR_ARG;
    if (K == 0) return S2D_OK;
    const long words = (n + 31) / 32;
    if ((words * 32 + 255) / 256 >= (1L << 31)) return S2D_ERR_ARG;
    hipLaunchKernelGGL(pack_bits_kernel, dim3(cdiv(words * 32, 256), K), dim3(256), 0, stream, masks, n, words, bits);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

int s2d_mask_pair_counts_u64(const uint32_t *bits, int K, long words, unsigned long long *inter, hipStream_t stream)
{
    if (K < 0 || words < 0) return S2D_ERR_ARG;
    if (K == 0) return S2D_OK;
    if (s2d_zero_async(inter, sizeof(unsigned long long) * (size_t)K * K, stream) != S2D_OK) return S2D_ERR_LAUNCH;
    if (words == 0) return S2D_OK;
    const int ntile = (K + PT - 1) / PT;
    const int npairs = ntile * (ntile + 1) / 2;
    int chunks = cdiv(words, 256 * 16);                 // >= 16 words per thread
    const int want = cdiv(2048, npairs);                // enough workgroups to fill 256 CUs
    if (chunks > want) chunks = want;
    if (chunks < 1) chunks = 1;
    hipLaunchKernelGGL(pair_count_kernel, dim3(chunks, npairs), dim3(256), 0, stream, bits, K, words, ntile, inter);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

}  // extern "C"
