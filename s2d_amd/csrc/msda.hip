// Multi-scale deformable attention sampling for gfx950.
//
// Replaces ms_deformable_im2col_gpu_kernel (ops/src/cuda/ms_deform_im2col_cuda.cuh:242-304, bilinear fetch
// :38-89) and, in the fused form, also the softmax / sampling-location arithmetic of
// MSDeformAttn.forward (ops/modules/ms_deform_attn.py:101-109).
//
// Mapping (wave64-native): one work item = (query, head, 4-channel group); a head's D channels are one
// contiguous 4*D-byte run of `value`, so the D/4 lanes of a head read each bilinear corner as one coalesced
// 128-B row (D = 32) with 16-B loads.  With M*D = 256 one wavefront is exactly one query: its 8 heads x 8
// lanes.  The 12 (level, point) samples of a (query, head) are read once per lane as float4s (the 8 lanes
// of a head hit the same address: one broadcast fetch), all 48 corner gathers are independent and stay in
// flight together.  Bound: HBM/L2 gather bandwidth, no MFMA.  Blocks are renumbered so each XCD sweeps a
// contiguous band of queries: a band's sampling footprint (~1/8 of a 20 MB value map) then stays in that
// XCD's 4 MB L2.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int MAX_L = 8;

struct Levels {
    int H[MAX_L], W[MAX_L];
    long start[MAX_L];
};

// Bilinear-cell index space of the atomic-free backward (see below).
struct Cells {
    int base[MAX_L];      // first cell index of level l inside one (frame, head)
    int tot;              // cells per (frame, head): sum over levels of (H + 1) * (W + 1)  (floor(y) in [-1, H-1], floor(x) in [-1, W-1])
};

// Geometry of one call.  The host-shape entry points pass it by value (kernarg memory); the *_dev entry points, whose spatial
// shapes / level starts are DEVICE tensors as in the reference op (ms_deform_attn_cuda.cu:60-75), have msda_geom_kernel write it
// into the caller's workspace once per call and every kernel reads it through a pointer (uniform scalar loads either way).
struct Geom {
    Levels lv;
    Cells cl;
    unsigned int kmax;    // N * M * cl.tot: the key of a sample outside every map
    int err;              // dev form: the shapes failed fill_levels' checks; kernels then write zeros and touch nothing else
};
struct GeomVal {
    Geom g;
    __device__ __forceinline__ const Levels &levels() const { return g.lv; }
    __device__ __forceinline__ const Cells &cells() const { return g.cl; }
    __device__ __forceinline__ unsigned int kmax() const { return g.kmax; }
    __device__ __forceinline__ bool bad() const { return false; }
};
struct GeomPtr {
    const Geom *g;
    __device__ __forceinline__ const Levels &levels() const { return g->lv; }
    __device__ __forceinline__ const Cells &cells() const { return g->cl; }
    __device__ __forceinline__ unsigned int kmax() const { return g->kmax; }
    __device__ __forceinline__ bool bad() const { return g->err != 0; }
};

__device__ __forceinline__ int xcd_band(int bid, int nblk)
{
    const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, within = bid / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
}

template <int V>
struct Vec;
template <>
struct Vec<4> {
    typedef f32x4 T;
};
template <>
struct Vec<1> {
    typedef float T;
};

template <int V>
__device__ __forceinline__ void sample_accum(typename Vec<V>::T &acc, const float *__restrict__ vbase, long rowstride,
                                             int H, int W, float h_im, float w_im, float aw)
{
    typedef typename Vec<V>::T VT;
    if (!(h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W)) return;  // cuh:293
    const int h0 = (int)floorf(h_im), w0 = (int)floorf(w_im), h1 = h0 + 1, w1 = w0 + 1;
    const float lh = h_im - h0, lw = w_im - w0, hh = 1.f - lh, hw = 1.f - lw;
    VT v1 = VT(0.f), v2 = VT(0.f), v3 = VT(0.f), v4 = VT(0.f);
    if (h0 >= 0 && w0 >= 0) v1 = *reinterpret_cast<const VT *>(vbase + ((long)h0 * W + w0) * rowstride);
    if (h0 >= 0 && w1 <= W - 1) v2 = *reinterpret_cast<const VT *>(vbase + ((long)h0 * W + w1) * rowstride);
    if (h1 <= H - 1 && w0 >= 0) v3 = *reinterpret_cast<const VT *>(vbase + ((long)h1 * W + w0) * rowstride);
    if (h1 <= H - 1 && w1 <= W - 1) v4 = *reinterpret_cast<const VT *>(vbase + ((long)h1 * W + w1) * rowstride);
    const float c1 = hh * hw, c2 = hh * lw, c3 = lh * hw, c4 = lh * lw;
    acc += (c1 * v1 + c2 * v2 + c3 * v3 + c4 * v4) * aw;  // cuh:85-88, :299
}

// Drop-in form: sampling locations and attention weights are inputs (the reference op's signature).
template <int V, class G>
__global__ __launch_bounds__(256) void msda_fwd_kernel(const float *__restrict__ value, G geo,
                                                       const float *__restrict__ loc, const float *__restrict__ aw,
                                                       int S, int M, int D, int L, int Lq, int P, int blk_per_n,
                                                       float *__restrict__ out)
{
    typedef typename Vec<V>::T VT;
    const Levels &lv = geo.levels();      // a bad dev geometry has H = W = 0 on every level: every sample is outside, the output 0
    const int n = blockIdx.y;
    const int bid = xcd_band(blockIdx.x, blk_per_n);
    const int dv = D / V;
    const long item = (long)bid * 256 + threadIdx.x;
    if (item >= (long)Lq * M * dv) return;
    const int c = (int)(item % dv);
    const int m = (int)((item / dv) % M);
    const int q = (int)(item / ((long)dv * M));
    const long qm = ((long)n * Lq + q) * M + m;
    const float *lp = loc + qm * L * P * 2;
    const float *wp = aw + qm * L * P;
    const long rowstride = (long)M * D;
    VT acc = VT(0.f);
    if (V == 4 && dv == 8 && L * P <= 16) {
        // D = 32: a head is 8 lanes of 4 channels; they divide the per-sample set-up as in the fused kernel below (lane j: samples
        // j and j + 8: location / weight loads, floor, bilinear weights, border tests), shared by 8-lane shuffles.  Bit-identical.
        const int LP = L * P, j = c;
        float s1[2], s2[2], s3[2], s4[2], sa[2];
        int spix[2], smask[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int i = j + 8 * r;
            const int ic = i < LP ? i : LP - 1;
            const int l = ic / P;
            const int H = lv.H[l], W = lv.W[l];
            const float2 xy = *reinterpret_cast<const float2 *>(lp + 2 * ic);
            sa[r] = wp[ic];
            const float h_im = xy.y * H - 0.5f, w_im = xy.x * W - 0.5f;
            const bool in = i < LP && h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;  // cuh:293
            const int h0 = (int)floorf(h_im), w0 = (int)floorf(w_im), h1 = h0 + 1, w1 = w0 + 1;
            const float lh = h_im - h0, lw = w_im - w0, hh = 1.f - lh, hw = 1.f - lw;
            s1[r] = hh * hw; s2[r] = hh * lw; s3[r] = lh * hw; s4[r] = lh * lw;
            spix[r] = h0 * W + w0;
            smask[r] = !in ? 0 : ((h0 >= 0 && w0 >= 0) ? 1 : 0) | ((h0 >= 0 && w1 <= W - 1) ? 2 : 0) | ((h1 <= H - 1 && w0 >= 0) ? 4 : 0) |
                                 ((h1 <= H - 1 && w1 <= W - 1) ? 8 : 0);
        }
        for (int i = 0; i < LP; ++i) {
            const int r = i >> 3, src = i & 7;
            const int l = i / P;
            const int mask = __shfl(r ? smask[1] : smask[0], src, 8);
            if (mask == 0) continue;                           // uniform over the head's 8 lanes
            const int W = lv.W[l];
            const int pix = __shfl(r ? spix[1] : spix[0], src, 8);
            const float c1 = __shfl(r ? s1[1] : s1[0], src, 8), c2 = __shfl(r ? s2[1] : s2[0], src, 8);
            const float c3 = __shfl(r ? s3[1] : s3[0], src, 8), c4 = __shfl(r ? s4[1] : s4[0], src, 8);
            const float a = __shfl(r ? sa[1] : sa[0], src, 8);
            const float *p1 = value + ((long)n * S + lv.start[l] + pix) * rowstride + m * D + c * V;
            VT v1 = VT(0.f), v2 = VT(0.f), v3 = VT(0.f), v4 = VT(0.f);
            if (mask & 1) v1 = *reinterpret_cast<const VT *>(p1);
            if (mask & 2) v2 = *reinterpret_cast<const VT *>(p1 + rowstride);
            if (mask & 4) v3 = *reinterpret_cast<const VT *>(p1 + (long)W * rowstride);
            if (mask & 8) v4 = *reinterpret_cast<const VT *>(p1 + (long)(W + 1) * rowstride);
            acc += (c1 * v1 + c2 * v2 + c3 * v3 + c4 * v4) * a;  // cuh:85-88, :299
        }
        *reinterpret_cast<VT *>(out + qm * D + c * V) = acc;
        return;
    }
    for (int l = 0; l < L; ++l) {
        const int H = lv.H[l], W = lv.W[l];
        const float *vbase = value + ((long)n * S + lv.start[l]) * rowstride + m * D + c * V;
        for (int p = 0; p < P; ++p) {
            const float lx = lp[(l * P + p) * 2], ly = lp[(l * P + p) * 2 + 1];
            sample_accum<V>(acc, vbase, rowstride, H, W, ly * H - 0.5f, lx * W - 0.5f, wp[l * P + p]);
        }
    }
    *reinterpret_cast<VT *>(out + qm * D + c * V) = acc;
}

// Fused form for the pixel decoder (self-attention over the flattened pyramid: Lq == S, the query's own
// normalised pixel centre is its reference point on every level, msdeformattn.py:141-153 with valid
// ratios == 1).  `oa` [N,S,ldoa] holds, per query, the raw sampling offsets [M][L][P][2] followed by the raw
// attention logits [M][L*P] (one GEMM output).  Softmax over L*P and loc = ref + off/(W_l,H_l) happen here.
// TILED: a 1024-thread workgroup = the 16 queries of a 4 x 4 pixel patch of one level (wave w: pixel (w / 4, w % 4) of the patch), the
// patches of a frame numbered level by level, row-major, and dealt to the XCDs in contiguous bands.  The 16 waves of a workgroup
// are co-resident on one CU by construction, so the corner rows neighbouring queries share (their sampling offsets differ by
// less than the patch when the offsets field is smooth) are fetched into that CU's L1 once; with 256-thread workgroups of 4
// consecutive queries the dispatcher deals a CU workgroups that lie 128 queries apart.
// timing experiments only (scripts/build_msda_dbg.sh; results wrong by construction): 1 no value gathers, 2 one corner line per sample instead of
// four (profiles/r4_experiments/msda_dbg.txt)
#ifndef S2D_MSDA_DBG
#define S2D_MSDA_DBG 0
#endif
template <int LP_, bool HM = false, bool SHARE = false, bool TILED = false>
__global__ __launch_bounds__(TILED ? 1024 : 256) void msda_fused_kernel(const float *__restrict__ value, int ldv, Levels lv,
                                                         const float *__restrict__ oa, int ldoa, int S, int M, int L,
                                                         int P, int blk_per_n, float *__restrict__ out, int skip = -1)
{
    constexpr int D = 32, V = 4, dv = D / V;
    const int n = blockIdx.y;
    const int bid = xcd_band(blockIdx.x, blk_per_n);
    int c, m, q, lq = 0, qy, qx;
    if constexpr (TILED) {
        // `skip`: a level whose queries another launch takes (msda_fused_win_kernel); its patches are not numbered
        int tb = bid, ntx = 1;
        for (; lq < L; ++lq) {
            if (lq == skip) continue;
            ntx = (lv.W[lq] + 3) >> 2;
            const int nt = ntx * ((lv.H[lq] + 3) >> 2);
            if (tb < nt) break;
            tb -= nt;
        }
        if (lq >= L) return;
        const int w = threadIdx.x >> 6;
        qy = (tb / ntx) * 4 + (w >> 2);
        qx = (tb % ntx) * 4 + (w & 3);
        if (qy >= lv.H[lq] || qx >= lv.W[lq]) return;         // whole waves leave (patches overhanging the level's border)
        q = (int)lv.start[lq] + qy * lv.W[lq] + qx;
        c = threadIdx.x & 7;
        m = (threadIdx.x >> 3) & 7;
    } else {
        const long item = (long)bid * 256 + threadIdx.x;
        if (item >= (long)S * M * dv) return;
        c = (int)(item % dv);
        m = (int)((item / dv) % M);
        q = (int)(item / ((long)dv * M));
        // which level does query q live on, and where
        while (lq + 1 < L && q >= lv.start[lq + 1]) ++lq;
        const int qi = q - (int)lv.start[lq];
        qy = qi / lv.W[lq]; qx = qi - qy * lv.W[lq];
    }
    const float ref_x = ((float)qx + 0.5f) / (float)lv.W[lq];
    const float ref_y = ((float)qy + 0.5f) / (float)lv.H[lq];

    const float *row = oa + ((long)n * S + q) * ldoa;
    const float *offp_g = row + m * (LP_ * 2);
    const float *lgp = row + M * LP_ * 2 + m * LP_;
    float offp[LP_ * 2];
    float lg[LP_];
    float inv;
    // SHARE: the 8 lanes of a head divide its samples (lane j: samples j and j + 8).  Each lane loads the offsets and the logit of
    // its own samples, evaluates their exponentials and their bilinear set-up (floor, weights, border tests, first pixel), and the
    // group reads everything back with 8-lane shuffles (ds_bpermute: the LDS crossbar, idle in this kernel).  Same formulas; the
    // softmax denominator is added in index order in every lane, so the results are bit-identical to the all-in-one-lane form.
    const int sj = threadIdx.x & 7;
    float own_ox[2] = {0.f, 0.f}, own_oy[2] = {0.f, 0.f}, own_e[2] = {0.f, 0.f};
    if constexpr (SHARE && LP_ <= 16) {
        float own_lg[2];
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int i = sj + 8 * r;
            const bool live = i < LP_;
            const int ic = live ? i : LP_ - 1;
            const float2 o = *reinterpret_cast<const float2 *>(offp_g + 2 * ic);
            own_ox[r] = o.x; own_oy[r] = o.y;
            own_lg[r] = live ? lgp[ic] : -INFINITY;
            mx = fmaxf(mx, own_lg[r]);
        }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 8));
#pragma unroll
        for (int r = 0; r < 2; ++r) own_e[r] = expf(own_lg[r] - mx);
        float den = 0.f;
#pragma unroll
        for (int i = 0; i < LP_; ++i) den += __shfl(own_e[i >> 3], i & 7, 8);
        inv = 1.f / den;
    } else {
    // offsets (2 * LP floats) and logits (LP floats) of this (query, head) as 16-B loads: with scalar loads they were a
    // quarter of the kernel's L1 accesses (36 instructions x 8 heads' lines per wave vs 9 now)
    if constexpr (LP_ % 4 == 0) {
#pragma unroll
        for (int i = 0; i < LP_ * 2; i += 4) {
            const f32x4 t = *reinterpret_cast<const f32x4 *>(offp_g + i);
            offp[i] = t[0]; offp[i + 1] = t[1]; offp[i + 2] = t[2]; offp[i + 3] = t[3];
        }
#pragma unroll
        for (int i = 0; i < LP_; i += 4) {
            const f32x4 t = *reinterpret_cast<const f32x4 *>(lgp + i);
            lg[i] = t[0]; lg[i + 1] = t[1]; lg[i + 2] = t[2]; lg[i + 3] = t[3];
        }
    } else {
#pragma unroll
        for (int i = 0; i < LP_ * 2; ++i) offp[i] = offp_g[i];
#pragma unroll
        for (int i = 0; i < LP_; ++i) lg[i] = lgp[i];
    }
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < LP_; ++i) mx = fmaxf(mx, lg[i]);
    float den = 0.f;
#pragma unroll
    for (int i = 0; i < LP_; ++i) { lg[i] = expf(lg[i] - mx); den += lg[i]; }
    inv = 1.f / den;
    }
    // HM (experiment): value stored head-major, [N][M][S][32], a pixel's 128 B of one head next to its neighbours'
    const long rowstride = HM ? D : ldv;
    f32x4 acc = f32x4(0.f);
    if constexpr (SHARE && LP_ <= 16) {
        // The bilinear set-up of a sample (floor, weights, border tests, first pixel) is the same for the 8 lanes of a head.
        // Lane j of the group prepares samples j and j + 8 and the group reads them back with 8-lane shuffles (ds_bpermute: the
        // LDS crossbar, idle in this kernel): 2 set-ups per wave instruction stream instead of LP_.  Same formulas, same values.
        const int j = sj;
        float sw1[2], sw2[2], sw3[2], sw4[2], saw[2];
        int spix[2], smask[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int i = j + 8 * r;
            const int ic = i < LP_ ? i : LP_ - 1;
            const float ox = own_ox[r], oy = own_oy[r], lgi = own_e[r];
            const int l = ic / P;
            const int H = lv.H[l], W = lv.W[l];
            const float lx = ref_x + ox / (float)W, ly = ref_y + oy / (float)H;
            const float h_im = ly * H - 0.5f, w_im = lx * W - 0.5f;
            const bool in = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;   // cuh:293
            const int h0 = (int)floorf(h_im), w0 = (int)floorf(w_im), h1 = h0 + 1, w1 = w0 + 1;
            const float lh = h_im - h0, lw = w_im - w0, hh = 1.f - lh, hw = 1.f - lw;
            sw1[r] = hh * hw; sw2[r] = hh * lw; sw3[r] = lh * hw; sw4[r] = lh * lw;
            saw[r] = lgi * inv;
            spix[r] = h0 * W + w0;
            smask[r] = !in ? 0 : ((h0 >= 0 && w0 >= 0) ? 1 : 0) | ((h0 >= 0 && w1 <= W - 1) ? 2 : 0) | ((h1 <= H - 1 && w0 >= 0) ? 4 : 0) |
                                 ((h1 <= H - 1 && w1 <= W - 1) ? 8 : 0);
        }
#pragma unroll
        for (int i = 0; i < LP_; ++i) {
            const int r = i >> 3, src = i & 7;
            const int l = i / P;
            const int W = lv.W[l];
            const int mask = __shfl(smask[r], src, 8);
            if (mask == 0) continue;                           // uniform over the head's 8 lanes
            const int pix = __shfl(spix[r], src, 8);
            const float c1 = __shfl(sw1[r], src, 8), c2 = __shfl(sw2[r], src, 8), c3 = __shfl(sw3[r], src, 8), c4 = __shfl(sw4[r], src, 8);
            const float aw = __shfl(saw[r], src, 8);
            const float *vbase = HM ? value + (((long)n * M + m) * S + lv.start[l]) * D + c * V
                                    : value + ((long)n * S + lv.start[l]) * rowstride + m * D + c * V;
            const float *p1 = vbase + (long)pix * rowstride;
            f32x4 v1 = f32x4(0.f), v2 = f32x4(0.f), v3 = f32x4(0.f), v4 = f32x4(0.f);
            if (S2D_MSDA_DBG & 1) {                                // timing experiment: no gathers
                v1 = f32x4(c1); v2 = f32x4(c2); v3 = f32x4(aw); v4 = f32x4((float)pix);
            } else if (S2D_MSDA_DBG & 2) {                         // timing experiment: one corner line instead of four
                v1 = *reinterpret_cast<const f32x4 *>(p1 + ((mask & 1) ? 0 : (long)(W + 1) * rowstride));
                v2 = v1 * c2; v3 = v1 * c3; v4 = v1 * c4;
            } else
            if (__builtin_amdgcn_readfirstlane(mask) == 15 && __all(mask == 15)) {
                // interior sample in every head of the wave (the common case): four loads, no per-corner exec juggling
                v1 = *reinterpret_cast<const f32x4 *>(p1);
                v2 = *reinterpret_cast<const f32x4 *>(p1 + rowstride);
                v3 = *reinterpret_cast<const f32x4 *>(p1 + (long)W * rowstride);
                v4 = *reinterpret_cast<const f32x4 *>(p1 + (long)(W + 1) * rowstride);
            } else {
                if (mask & 1) v1 = *reinterpret_cast<const f32x4 *>(p1);
                if (mask & 2) v2 = *reinterpret_cast<const f32x4 *>(p1 + rowstride);
                if (mask & 4) v3 = *reinterpret_cast<const f32x4 *>(p1 + (long)W * rowstride);
                if (mask & 8) v4 = *reinterpret_cast<const f32x4 *>(p1 + (long)(W + 1) * rowstride);
            }
            acc += (c1 * v1 + c2 * v2 + c3 * v3 + c4 * v4) * aw;  // cuh:85-88, :299
        }
    } else {
#pragma unroll
    for (int i = 0; i < LP_; ++i) {
        const int l = i / P;
        const int H = lv.H[l], W = lv.W[l];
        const float *vbase = HM ? value + (((long)n * M + m) * S + lv.start[l]) * D + c * V
                                : value + ((long)n * S + lv.start[l]) * rowstride + m * D + c * V;
        const float lx = ref_x + offp[2 * i] / (float)W;       // ms_deform_attn.py:106-109
        const float ly = ref_y + offp[2 * i + 1] / (float)H;
        sample_accum<4>(acc, vbase, rowstride, H, W, ly * H - 0.5f, lx * W - 0.5f, lg[i] * inv);
    }
    }
    *reinterpret_cast<f32x4 *>(out + (((long)n * S + q) * M + m) * D + c * V) = acc;
}

// Record form of the fused kernel (round 5; the default for the S2D geometry L = 3, P = 4, M = 8, D = 32).
// The TILED form above spends half its time outside the gathers (profiles/r4_experiments/msda_dbg.txt: 0.32 of 0.67 ms with no gather
// at all), and its ISA shows why: per sample 7 ds_bpermute + a wait, two scalar loads of the level's geometry (the level index is i / P
// with a runtime P) + a wait, ~25 scalar instructions of exec juggling around the border cases, then four loads and `s_waitcnt vmcnt(0)`
// before the next sample starts -- ~600 scalar instructions per query on the CU's one scalar unit and twelve serialised
// shuffle -> load -> use chains per wave.  Here the set-up is split off completely:
//   phase 1  lane (head m, j) prepares samples j and j + 8 of its head as in the SHARE form (same formulas, same order) and writes one
//            RECORD per sample into the wave's own LDS area: the four corners' BYTE OFFSETS into the frame's value slice (0x80000000 for a
//            corner outside the map: the buffer load's bounds check then returns zeros, which is the value the reference gives a missing
//            corner, cuh:61-83), the four bilinear weights and the attention weight;
//   phase 2  twelve branch-free steps: the 8 lanes of a head read their record (two ds_read_b128 + one ds_read_b32, broadcast within the
//            head, conflict-free across heads), issue four buffer_load_dwordx4 with 32-bit offsets (no 64-bit address arithmetic) and
//            accumulate; nothing in a step depends on the previous one but the accumulator, so the compiler keeps several samples' loads
//            in flight.
// L and P are template constants (level of sample i known at compile time; H, W, start live in SGPRs for the whole kernel).
// Arithmetic and accumulation order are those of msda_fused_kernel: results are bit-identical (tests/test_gpu_msda_glue.py).
// Record row of sample i: 8 heads x 48 B + 16 B of skew (400 B: the 8 lanes of a head WRITE samples j = 0..7, 400 B apart = 4 banks
// apart: conflict-free ds_write_b128; the heads of one sample READ 48 B apart: 12 banks).
constexpr int REC_HEAD = 48;
constexpr int REC_ROW = 8 * REC_HEAD + 16;
constexpr unsigned int REC_OOB = 0x80000000u;

template <int L_, int P_, int WAVES_EU>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(WAVES_EU, WAVES_EU)))
void msda_fused_rec_kernel(const float *__restrict__ value, int ldv, Levels lv, const float *__restrict__ oa, int ldoa, int S,
                           int blk_per_n, float *__restrict__ out, unsigned int frame_bytes)
{
    constexpr int LP = L_ * P_, M = 8, D = 32;
    constexpr int REC_WAVE = LP * REC_ROW;
    static_assert(LP <= 16 && LP > 8, "two samples per lane");
    extern __shared__ __attribute__((aligned(16))) unsigned char rec_lds[];
    const int n = blockIdx.y;
    int tb = xcd_band(blockIdx.x, blk_per_n), lq = 0, ntx = 1;
#pragma unroll
    for (; lq < L_; ++lq) {
        ntx = (lv.W[lq] + 3) >> 2;
        const int nt = ntx * ((lv.H[lq] + 3) >> 2);
        if (tb < nt) break;
        tb -= nt;
    }
    if (lq >= L_) return;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int qy = (tb / ntx) * 4 + (w >> 2), qx = (tb % ntx) * 4 + (w & 3);
    if (qy >= lv.H[lq] || qx >= lv.W[lq]) return;             // whole waves leave (patches overhanging the level's border); no workgroup barrier below
    const int q = (int)lv.start[lq] + qy * lv.W[lq] + qx;
    const int m = lane >> 3, j = lane & 7;
    const float ref_x = ((float)qx + 0.5f) / (float)lv.W[lq];
    const float ref_y = ((float)qy + 0.5f) / (float)lv.H[lq];
    unsigned char *rec = rec_lds + w * REC_WAVE;

    // ---- phase 1: softmax over the head's L * P logits, bilinear set-up of this lane's two samples, records -> LDS ----
    const float *row = oa + ((long)n * S + q) * ldoa;
    const float *offp_g = row + m * (LP * 2);
    const float *lgp = row + M * LP * 2 + m * LP;
    float own_ox[2], own_oy[2], own_lg[2], own_e[2];
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i = j + 8 * r;
        const bool live = i < LP;
        const int ic = live ? i : LP - 1;
        const float2 o = *reinterpret_cast<const float2 *>(offp_g + 2 * ic);
        own_ox[r] = o.x; own_oy[r] = o.y;
        own_lg[r] = live ? lgp[ic] : -INFINITY;
        mx = fmaxf(mx, own_lg[r]);
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 8));
#pragma unroll
    for (int r = 0; r < 2; ++r) own_e[r] = expf(own_lg[r] - mx);
    float den = 0.f;
#pragma unroll
    for (int i = 0; i < LP; ++i) den += __shfl(own_e[i >> 3], i & 7, 8);      // index order in every lane, as the other forms
    const float inv = 1.f / den;
    const unsigned int ldv4 = (unsigned int)ldv * 4u;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i = j + 8 * r;
        if (i < LP) {
            const int l = i / P_;                                             // lane-dependent: selects over the (scalar) level table
            int H = lv.H[0], W = lv.W[0], st = (int)lv.start[0];
#pragma unroll
            for (int k = 1; k < L_; ++k)
                if (l == k) { H = lv.H[k]; W = lv.W[k]; st = (int)lv.start[k]; }
            const float lx = ref_x + own_ox[r] / (float)W, ly = ref_y + own_oy[r] / (float)H;      // ms_deform_attn.py:106-109
            const float h_im = ly * H - 0.5f, w_im = lx * W - 0.5f;
            const bool in = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;   // cuh:293
            const int h0 = (int)floorf(h_im), w0 = (int)floorf(w_im), h1 = h0 + 1, w1 = w0 + 1;
            const float lh = h_im - h0, lw = w_im - w0, hh = 1.f - lh, hw = 1.f - lw;
            const f32x4 cw = {hh * hw, hh * lw, lh * hw, lh * lw};
            const bool t = in && h0 >= 0, b = in && h1 <= H - 1, lft = w0 >= 0, rgt = w1 <= W - 1;
            const unsigned int p00 = (unsigned int)(st + h0 * W + w0) * ldv4;  // wraps for h0 / w0 = -1; only used where the corner exists
            typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
            const u32x4_t off = {t && lft ? p00 : REC_OOB, t && rgt ? p00 + ldv4 : REC_OOB,
                                 b && lft ? p00 + (unsigned int)W * ldv4 : REC_OOB, b && rgt ? p00 + (unsigned int)(W + 1) * ldv4 : REC_OOB};
            unsigned char *d = rec + i * REC_ROW + m * REC_HEAD;
            *reinterpret_cast<u32x4_t *>(d) = off;
            *reinterpret_cast<f32x4 *>(d + 16) = cw;
            *reinterpret_cast<float *>(d + 32) = own_e[r] * inv;
        }
    }
    // the records cross lanes inside the wave only: LDS instructions of one wave execute in order; keep the compiler from moving the
    // reads above the writes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- phase 2: 12 x (record, four 128-B corner rows of the head, accumulate) ----
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(value + (long)n * S * ldv), 0, (int)frame_bytes, 0x00020000);
    const unsigned int lanecol = (unsigned int)(m * D + j * 4) * 4u;
    const unsigned char *rd = rec + m * REC_HEAD;
    f32x4 acc = f32x4(0.f);
    // explicit software pipeline, DEPTH samples' loads (4 x 16 B per lane each) in flight: the slot a step has just consumed is
    // refilled with the sample DEPTH steps ahead; the scheduling fence after every step keeps the compiler from hoisting more
    // loads than the register budget of WAVES_EU waves per SIMD holds
    constexpr int DEPTH = WAVES_EU >= 8 ? 2 : 4;
    f32x4 pv[DEPTH][4], pc[DEPTH];
    float pa[DEPTH];
    auto issue = [&](int i, int sl) {
        typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
        const u32x4_t off = *reinterpret_cast<const u32x4_t *>(rd + i * REC_ROW);
        pc[sl] = *reinterpret_cast<const f32x4 *>(rd + i * REC_ROW + 16);
        pa[sl] = *reinterpret_cast<const float *>(rd + i * REC_ROW + 32);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            pv[sl][k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsV, (int)(off[k] + lanecol), 0, 0));
    };
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) issue(i, i);
#pragma unroll
    for (int i = 0; i < LP; ++i) {
        const int sl = i % DEPTH;
        acc += (pc[sl][0] * pv[sl][0] + pc[sl][1] * pv[sl][1] + pc[sl][2] * pv[sl][2] + pc[sl][3] * pv[sl][3]) * pa[sl];     // cuh:85-88, :299
        if (i + DEPTH < LP) issue(i + DEPTH, sl);
        __builtin_amdgcn_sched_barrier(0);
    }
    *reinterpret_cast<f32x4 *>(out + (((long)n * S + q) * M + m) * D + j * 4) = acc;
}

// Record form of the query-owned half of the fused backward (round 5): d(loss)/d(offsets, logits) of one (query, head) from grad_out and
// the value rows, in the geometry of msda_fused_rec_kernel (a wave = one query, 8 lanes x 4 channels per head, per-wave records in LDS,
// twelve branch-free steps of four bounds-checked 16-B loads) -- it replaces msda_bwd_loc_kernel + msda_fused_chain_kernel for the S2D
// geometry and reads the raw projection rows, so grad_loc / grad_attn never exist in memory:
//   d_k      = <grad_out[q, m, :], value[corner k]>                              (4 channels per lane, summed over the head's 8 lanes)
//   d a_i    = hh hw d1 + hh lw d2 + lh hw d3 + lh lw d4                         (cuh:150-155)
//   d off_x  = a_i (hh (d2 - d1) + lh (d4 - d3)),  d off_y = a_i (hw (d3 - d1) + lw (d4 - d2))
//              -- the W of d/d(loc.x) (cuh:146-149) and the 1 / W of loc = ref + off / W (ms_deform_attn.py:106-109) cancel
//   d logit_i = a_i (d a_i - sum_j a_j d a_j)                                     (softmax backward, j in index order)
// The 8-lane sums are three DPP adds (quad_perm xor 1, xor 2, row_half_mirror): every lane of a head ends with the same bits.
// Record of a sample: four corner byte offsets (REC_OOB outside the map), lh, lw, a; lh = lw = 0 for a sample outside every map, whose
// corners all read zero: its gradients are exactly zero, as the reference skips it (cuh:367).
__device__ __forceinline__ float head_sum8(float x)
{
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));     // quad_perm [1,0,3,2]
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));     // quad_perm [2,3,0,1]
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));    // row_half_mirror
    return x;
}

template <int L_, int P_, int WAVES_EU>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(WAVES_EU, WAVES_EU)))
void msda_fused_bwd_rec_kernel(const float *__restrict__ value, int ldv, Levels lv, const float *__restrict__ oa, int ldoa,
                               const float *__restrict__ gout, int S, int blk_per_n, float *__restrict__ doa, int ldd,
                               unsigned int frame_bytes)
{
    constexpr int LP = L_ * P_, M = 8, D = 32;
    constexpr int REC_WAVE = LP * REC_ROW;
    static_assert(LP <= 16 && LP > 8, "two samples per lane");
    extern __shared__ __attribute__((aligned(16))) unsigned char rec_lds[];
    const int n = blockIdx.y;
    int tb = xcd_band(blockIdx.x, blk_per_n), lq = 0, ntx = 1;
#pragma unroll
    for (; lq < L_; ++lq) {
        ntx = (lv.W[lq] + 3) >> 2;
        const int nt = ntx * ((lv.H[lq] + 3) >> 2);
        if (tb < nt) break;
        tb -= nt;
    }
    if (lq >= L_) return;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int qy = (tb / ntx) * 4 + (w >> 2), qx = (tb % ntx) * 4 + (w & 3);
    if (qy >= lv.H[lq] || qx >= lv.W[lq]) return;             // whole waves leave; no workgroup barrier below
    const int q = (int)lv.start[lq] + qy * lv.W[lq] + qx;
    const int m = lane >> 3, j = lane & 7;
    const float ref_x = ((float)qx + 0.5f) / (float)lv.W[lq];
    const float ref_y = ((float)qy + 0.5f) / (float)lv.H[lq];
    unsigned char *rec = rec_lds + w * REC_WAVE;

    // ---- phase 1: as the forward's (same formulas, same order), the record holds lh, lw instead of the four products ----
    const float *row = oa + ((long)n * S + q) * ldoa;
    const float *offp_g = row + m * (LP * 2);
    const float *lgp = row + M * LP * 2 + m * LP;
    float own_ox[2], own_oy[2], own_lg[2], own_e[2];
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i = j + 8 * r;
        const bool live = i < LP;
        const int ic = live ? i : LP - 1;
        const float2 o = *reinterpret_cast<const float2 *>(offp_g + 2 * ic);
        own_ox[r] = o.x; own_oy[r] = o.y;
        own_lg[r] = live ? lgp[ic] : -INFINITY;
        mx = fmaxf(mx, own_lg[r]);
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 8));
#pragma unroll
    for (int r = 0; r < 2; ++r) own_e[r] = expf(own_lg[r] - mx);
    float den = 0.f;
#pragma unroll
    for (int i = 0; i < LP; ++i) den += __shfl(own_e[i >> 3], i & 7, 8);
    const float inv = 1.f / den;
    const unsigned int ldv4 = (unsigned int)ldv * 4u;
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i = j + 8 * r;
        if (i < LP) {
            const int l = i / P_;
            int H = lv.H[0], W = lv.W[0], st = (int)lv.start[0];
#pragma unroll
            for (int k = 1; k < L_; ++k)
                if (l == k) { H = lv.H[k]; W = lv.W[k]; st = (int)lv.start[k]; }
            const float lx = ref_x + own_ox[r] / (float)W, ly = ref_y + own_oy[r] / (float)H;
            const float h_im = ly * H - 0.5f, w_im = lx * W - 0.5f;
            const bool in = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
            const int h0 = (int)floorf(h_im), w0 = (int)floorf(w_im), h1 = h0 + 1, w1 = w0 + 1;
            const float lh = in ? h_im - h0 : 0.f, lw = in ? w_im - w0 : 0.f;
            const bool t = in && h0 >= 0, b = in && h1 <= H - 1, lft = w0 >= 0, rgt = w1 <= W - 1;
            const unsigned int p00 = (unsigned int)(st + h0 * W + w0) * ldv4;
            const u32x4_t off = {t && lft ? p00 : REC_OOB, t && rgt ? p00 + ldv4 : REC_OOB,
                                 b && lft ? p00 + (unsigned int)W * ldv4 : REC_OOB, b && rgt ? p00 + (unsigned int)(W + 1) * ldv4 : REC_OOB};
            unsigned char *d = rec + i * REC_ROW + m * REC_HEAD;
            *reinterpret_cast<u32x4_t *>(d) = off;
            own_e[r] *= inv;                                                    // a_i from here on
            *reinterpret_cast<f32x4 *>(d + 16) = f32x4{lh, lw, own_e[r], 0.f};
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- phase 2 ----
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(value + (long)n * S * ldv), 0, (int)frame_bytes, 0x00020000);
    const unsigned int lanecol = (unsigned int)(m * D + j * 4) * 4u;
    const unsigned char *rd = rec + m * REC_HEAD;
    const f32x4 tg = *reinterpret_cast<const f32x4 *>(gout + (((long)n * S + q) * M + m) * D + j * 4);
    constexpr int DEPTH = WAVES_EU >= 8 ? 2 : 4;
    f32x4 pv[DEPTH][4];
    float2 pl[DEPTH];
    float pa[DEPTH];
    auto issue = [&](int i, int sl) {
        const u32x4_t off = *reinterpret_cast<const u32x4_t *>(rd + i * REC_ROW);
        pl[sl] = *reinterpret_cast<const float2 *>(rd + i * REC_ROW + 16);
        pa[sl] = *reinterpret_cast<const float *>(rd + i * REC_ROW + 24);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            pv[sl][k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsV, (int)(off[k] + lanecol), 0, 0));
    };
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) issue(i, i);
    float kw[2] = {0.f, 0.f}, kx[2] = {0.f, 0.f}, ky[2] = {0.f, 0.f};
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < LP; ++i) {
        const int sl = i % DEPTH;
        float d[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) d[k] = tg[0] * pv[sl][k][0] + tg[1] * pv[sl][k][1] + tg[2] * pv[sl][k][2] + tg[3] * pv[sl][k][3];
        const float lh = pl[sl].x, lw = pl[sl].y, a = pa[sl];
        const float hh = 1.f - lh, hw = 1.f - lw;
        float gw = hh * hw * d[0] + hh * lw * d[1] + lh * hw * d[2] + lh * lw * d[3];
        float gx = a * (hh * (d[1] - d[0]) + lh * (d[3] - d[2]));
        float gy = a * (hw * (d[2] - d[0]) + lw * (d[3] - d[1]));
        if (i + DEPTH < LP) issue(i + DEPTH, sl);
        gw = head_sum8(gw); gx = head_sum8(gx); gy = head_sum8(gy);
        dot += a * gw;
        if ((i & 7) == j) { kw[i >> 3] = gw; kx[i >> 3] = gx; ky[i >> 3] = gy; }
        __builtin_amdgcn_sched_barrier(0);
    }
    float *orow = doa + ((long)n * S + q) * ldd;
    float *od = orow + m * (LP * 2), *ol = orow + M * LP * 2 + m * LP;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i = j + 8 * r;
        if (i < LP) {
            *reinterpret_cast<float2 *>(od + 2 * i) = float2{kx[r], ky[r]};
            ol[i] = own_e[r] * (kw[r] - dot);
        }
    }
}

// One HEAD per workgroup, the coarsest level of that head in LDS.  The gather is bound by the line rate of the vector L1: every bilinear
// tap of a (query, head) is one 128-B line, 48 per (query, head) at L x P = 12.  A head's 32 channels of the coarsest level are
// H x W x 128 B -- 115 KB at the S2D geometry (23 x 40) -- so a workgroup that works on ONE head copies that plane into LDS once and
// serves the P samples of that level from it with ds_read_b128 (256 B/clk/CU against the L1's 64): 32 lines per (query, head) instead
// of 48, for any offsets (nothing is windowed: every tap of the level is in the plane).  Workgroup = (frame, head, chunk of the
// frame's query patches); patch = 16 x 8 queries of one level, wave w = 8 x-adjacent queries of patch row w / 2 (8 lanes x float4 =
// the head's 32 channels of a query), so the lines neighbouring queries share stay in the CU's L1 as in the TILED form.  blockIdx.x
// % 8 is the head, so an XCD's L2 only ever sees the lines of one head.  Same formulas and accumulation order as msda_fused_kernel
// (SHARE form): bit-identical results.
template <int LP_>
__global__ __launch_bounds__(1024) void msda_fused_head_kernel(const float *__restrict__ value, int ldv, Levels lv, const float *__restrict__ oa,
                                                               int ldoa, int S, int M, int L, int P, int ls, int npatch, int chunks,
                                                               float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float plane[];     // [H_ls * W_ls][32]
    constexpr int D = 32;
    const int n = blockIdx.y, m = blockIdx.x % M, chunk = blockIdx.x / M;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 7;
    {
        const int npx = lv.H[ls] * lv.W[ls];
        const float *src = value + ((long)n * S + lv.start[ls]) * ldv + m * D;
        for (int i = threadIdx.x; i < npx * 8; i += 1024)
            *reinterpret_cast<f32x4 *>(plane + (long)(i >> 3) * D + (i & 7) * 4) = *reinterpret_cast<const f32x4 *>(src + (long)(i >> 3) * ldv + (i & 7) * 4);
    }
    __syncthreads();
    const int p_lo = (int)((long)chunk * npatch / chunks), p_hi = (int)((long)(chunk + 1) * npatch / chunks);
    for (int pi = p_lo; pi < p_hi; ++pi) {
        int lq = 0, tb = pi, ntx = 1;
        for (; lq < L; ++lq) {
            ntx = (lv.W[lq] + 15) >> 4;
            const int nt = ntx * ((lv.H[lq] + 7) >> 3);
            if (tb < nt) break;
            tb -= nt;
        }
        const int Hq = lv.H[lq], Wq = lv.W[lq];
        const int qy_ = (tb / ntx) * 8 + (w >> 1), qx_ = (tb % ntx) * 16 + (w & 1) * 8 + (lane >> 3);
        const bool live_q = qy_ < Hq && qx_ < Wq;                  // a patch overhanging the level's border: computed on a clamped query, not stored
        const int qy = min(qy_, Hq - 1), qx = min(qx_, Wq - 1);
        const int q = (int)lv.start[lq] + qy * Wq + qx;
        const float ref_x = ((float)qx + 0.5f) / (float)Wq;
        const float ref_y = ((float)qy + 0.5f) / (float)Hq;
        const float *row = oa + ((long)n * S + q) * ldoa;
        const float *offp_g = row + m * (LP_ * 2);
        const float *lgp = row + M * LP_ * 2 + m * LP_;
        float own_ox[2], own_oy[2], own_e[2], own_lg[2];
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int i = c + 8 * r;
            const bool live = i < LP_;
            const int ic = live ? i : LP_ - 1;
            const float2 o = *reinterpret_cast<const float2 *>(offp_g + 2 * ic);
            own_ox[r] = o.x; own_oy[r] = o.y;
            own_lg[r] = live ? lgp[ic] : -INFINITY;
            mx = fmaxf(mx, own_lg[r]);
        }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 8));
#pragma unroll
        for (int r = 0; r < 2; ++r) own_e[r] = expf(own_lg[r] - mx);
        float den = 0.f;
#pragma unroll
        for (int i = 0; i < LP_; ++i) den += __shfl(own_e[i >> 3], i & 7, 8);
        const float inv = 1.f / den;
        float sw1[2], sw2[2], sw3[2], sw4[2], saw[2];
        int spix[2], smask[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int i = c + 8 * r;
            const int ic = i < LP_ ? i : LP_ - 1;
            const int l = ic / P;
            const int H = lv.H[l], W = lv.W[l];
            const float lx = ref_x + own_ox[r] / (float)W, ly = ref_y + own_oy[r] / (float)H;
            const float h_im = ly * H - 0.5f, w_im = lx * W - 0.5f;
            const bool in = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;   // cuh:293
            const int h0 = (int)floorf(h_im), w0 = (int)floorf(w_im), h1 = h0 + 1, w1 = w0 + 1;
            const float lh = h_im - h0, lw = w_im - w0, hh = 1.f - lh, hw = 1.f - lw;
            sw1[r] = hh * hw; sw2[r] = hh * lw; sw3[r] = lh * hw; sw4[r] = lh * lw;
            saw[r] = own_e[r] * inv;
            spix[r] = h0 * W + w0;
            smask[r] = !in ? 0 : ((h0 >= 0 && w0 >= 0) ? 1 : 0) | ((h0 >= 0 && w1 <= W - 1) ? 2 : 0) | ((h1 <= H - 1 && w0 >= 0) ? 4 : 0) |
                                 ((h1 <= H - 1 && w1 <= W - 1) ? 8 : 0);
        }
        f32x4 acc = f32x4(0.f);
#pragma unroll
        for (int i = 0; i < LP_; ++i) {
            const int r = i >> 3, src = i & 7;
            const int l = i / P;
            const int W = lv.W[l];
            const int mask = __shfl(smask[r], src, 8);
            if (mask == 0) continue;                               // uniform over the query's 8 lanes
            const int pix = __shfl(spix[r], src, 8);
            const float c1 = __shfl(sw1[r], src, 8), c2 = __shfl(sw2[r], src, 8), c3 = __shfl(sw3[r], src, 8), c4 = __shfl(sw4[r], src, 8);
            const float aw = __shfl(saw[r], src, 8);
            f32x4 v1 = f32x4(0.f), v2 = f32x4(0.f), v3 = f32x4(0.f), v4 = f32x4(0.f);
            if (l == ls) {                                         // workgroup-uniform: this level's taps come from the plane in LDS
                const float *p1 = plane + (long)pix * D + c * 4;
                if (mask & 1) v1 = *reinterpret_cast<const f32x4 *>(p1);
                if (mask & 2) v2 = *reinterpret_cast<const f32x4 *>(p1 + D);
                if (mask & 4) v3 = *reinterpret_cast<const f32x4 *>(p1 + (long)W * D);
                if (mask & 8) v4 = *reinterpret_cast<const f32x4 *>(p1 + (long)(W + 1) * D);
            } else {
                const float *p1 = value + ((long)n * S + lv.start[l] + pix) * ldv + m * D + c * 4;
                if (__builtin_amdgcn_readfirstlane(mask) == 15 && __all(mask == 15)) {
                    v1 = *reinterpret_cast<const f32x4 *>(p1);
                    v2 = *reinterpret_cast<const f32x4 *>(p1 + ldv);
                    v3 = *reinterpret_cast<const f32x4 *>(p1 + (long)W * ldv);
                    v4 = *reinterpret_cast<const f32x4 *>(p1 + (long)(W + 1) * ldv);
                } else {
                    if (mask & 1) v1 = *reinterpret_cast<const f32x4 *>(p1);
                    if (mask & 2) v2 = *reinterpret_cast<const f32x4 *>(p1 + ldv);
                    if (mask & 4) v3 = *reinterpret_cast<const f32x4 *>(p1 + (long)W * ldv);
                    if (mask & 8) v4 = *reinterpret_cast<const f32x4 *>(p1 + (long)(W + 1) * ldv);
                }
            }
            acc += (c1 * v1 + c2 * v2 + c3 * v3 + c4 * v4) * aw;  // cuh:85-88, :299
        }
        if (live_q) *reinterpret_cast<f32x4 *>(out + (((long)n * S + q) * M + m) * D + c * 4) = acc;
    }
}

// Windowed form of the fused kernel for the queries of the LAST level when it is the finest (three quarters of the pyramid's
// queries at the S2D geometry).  The gather above is bound by the line rate of the vector L1 (every bilinear tap of a head is one
// 128-B line; 48 taps per (query, head)), not by HBM.  Deformable offsets are local -- at the reference's initialisation
// (ms_deform_attn.py:66-80) they are the constant grid (+-1 .. +-4 px on every level), trained ones stay within a few pixels -- so
// the taps of an 8 x 8 patch of queries fall, per target level, into the patch's footprint on that level grown by a margin R.  A
// 512-thread workgroup = one patch (wave = a patch row, 8 lanes x float4 = a query's 32 channels of the current head) walks the
// heads; per head it brings the three windows (pixels outside the map as zeros) into LDS -- 629 lines for 64 x 48 = 3 072 taps at
// R = 4 -- and takes every tap whose 2 x 2 footprint lies inside its window from there (ds_read_b128: 128 B/clk/CU against the
// L1's 64).  A sample whose footprint leaves the window (offsets beyond R) takes the original global path, so any input gives the
// same result as msda_fused_kernel: same formulas, same order of accumulation.
// Windows go global -> LDS directly (buffer_load ... lds: no staging registers) into two regions that are refilled half a head
// apart: X = the coarser levels (samples 0 .. 7), Y = the last level (samples 8 .. 11).  While a head's X samples run, its Y
// window lands; while its Y samples and the next head's sample set-up run, the next head's X windows land.
struct WinGeo {
    int lq, R;
    int nx, ny, basey;             // pixels of region X / Y, first LDS pixel of region Y (a multiple of 8)
    int ww[4], wh[4], base[4];     // window extent and first LDS pixel of level l
    float sx[4], sy[4];            // W_l / W_lq, H_l / H_lq
};

template <int L_, int P_>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void msda_fused_win_kernel(
    const float *__restrict__ value, int ldv, Levels lv, WinGeo wg, const float *__restrict__ oa, int ldoa, int S, int M, int blk_per_n,
    float *__restrict__ out)
{
    constexpr int LP_ = L_ * P_, D = 32, MAXX = 6, MAXY = 5;
    static_assert(L_ == 3 && P_ == 4, "three levels of four points: lane j of a query owns samples j (levels 0 / 1) and 8 + j (level 2)");
    extern __shared__ __attribute__((aligned(16))) f32x4 win[];          // [pixel][8]
    const int n = blockIdx.y, bid = xcd_band(blockIdx.x, blk_per_n);
    constexpr int lq = L_ - 1;
    const int Wq = lv.W[lq], Hq = lv.H[lq];
    const int ntx = (Wq + 7) >> 3;
    const int qx0 = (bid % ntx) * 8, qy0 = (bid / ntx) * 8;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, c = lane & 7;
    const int qy = qy0 + wv, qx = qx0 + (lane >> 3);
    const bool valid = qy < Hq && qx < Wq;                               // overhanging queries compute on a clamped query, store nothing
    const int qyc = min(qy, Hq - 1), qxc = min(qx, Wq - 1);
    const int q = (int)lv.start[lq] + qyc * Wq + qxc;
    const float ref_x = ((float)qxc + 0.5f) / (float)Wq;
    const float ref_y = ((float)qyc + 0.5f) / (float)Hq;
    int ox[L_], oy[L_];
#pragma unroll
    for (int l = 0; l < L_; ++l) {
        ox[l] = (int)floorf(((float)qx0 + 0.5f) * wg.sx[l] - 0.5f) - wg.R;
        oy[l] = (int)floorf(((float)qy0 + 0.5f) * wg.sy[l] - 0.5f) - wg.R;
    }
    // this thread's window pixels (the same for every head): LDS pixel (region base) + it * 64 + slot <- byte offset of its 16 B
    // in the frame; a pixel outside the map or past the region: an offset beyond the buffer (the load brings zeros or nothing:
    // the regions are zeroed once)
    const int slot = tid >> 3;
    auto pixel_off = [&](int l, int loc) -> unsigned int {
        const int wwl = l == 0 ? wg.ww[0] : (l == 1 ? wg.ww[1] : wg.ww[2]);
        const int oxl = l == 0 ? ox[0] : (l == 1 ? ox[1] : ox[2]), oyl = l == 0 ? oy[0] : (l == 1 ? oy[1] : oy[2]);
        const int Hl = l == 0 ? lv.H[0] : (l == 1 ? lv.H[1] : lv.H[2]), Wl = l == 0 ? lv.W[0] : (l == 1 ? lv.W[1] : lv.W[2]);
        const int stl = (int)(l == 0 ? lv.start[0] : (l == 1 ? lv.start[1] : lv.start[2]));
        const int wy = loc / wwl, wx = loc - wy * wwl;
        const int gy = oyl + wy, gx = oxl + wx;
        const bool ok = gy >= 0 && gy < Hl && gx >= 0 && gx < Wl;
        return ok ? (unsigned int)(((stl + gy * Wl + gx) * ldv + c * 4) * 4) : 0x80000000u;
    };
    unsigned int soffx[MAXX], soffy[MAXY];
#pragma unroll
    for (int it = 0; it < MAXX; ++it) {
        const int pi = it * 64 + slot;
        const int l = pi >= wg.base[1] ? 1 : 0;
        soffx[it] = pi < wg.nx ? pixel_off(l, pi - (l ? wg.base[1] : 0)) : 0x80000000u;
    }
#pragma unroll
    for (int it = 0; it < MAXY; ++it) {
        const int pi = it * 64 + slot;
        soffy[it] = pi < wg.ny ? pixel_off(2, pi) : 0x80000000u;
    }
    const float *vn = value + (long)n * S * ldv;
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(vn), 0, (int)((long)S * ldv * 4), 0x00020000);
    auto issue_x = [&](int m) {
#pragma unroll
        for (int it = 0; it < MAXX; ++it) {
            if (it * 64 + wv * 8 >= wg.nx) break;          // wave-uniform
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (__attribute__((address_space(3))) void *)(win + (it * 64 + wv * 8) * 8), 16, (int)soffx[it],
                                                     m * (D * 4), 0, 0);
        }
    };
    auto issue_y = [&](int m) {
#pragma unroll
        for (int it = 0; it < MAXY; ++it) {
            if (it * 64 + wv * 8 >= wg.ny) break;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (__attribute__((address_space(3))) void *)(win + (wg.basey + it * 64 + wv * 8) * 8), 16,
                                                     (int)soffy[it], m * (D * 4), 0, 0);
        }
    };
    const float *row = oa + ((long)n * S + q) * ldoa;
    const int sj = c;
    // the lane's own two samples: i0 = sj on level l0 = sj >> 2 (0 or 1), i1 = 8 + sj on level 2 (lanes 0..3; lanes 4..7 repeat sample 11)
    const bool lvl1 = sj >= 4, live1 = sj < 4;
    const int i1 = live1 ? 8 + sj : LP_ - 1;
    const int H0_ = lvl1 ? lv.H[1] : lv.H[0], W0_ = lvl1 ? lv.W[1] : lv.W[0];
    const int gH[2] = {H0_, lv.H[2]}, gW[2] = {W0_, lv.W[2]};
    const int gww[2] = {lvl1 ? wg.ww[1] : wg.ww[0], wg.ww[2]}, gwh[2] = {lvl1 ? wg.wh[1] : wg.wh[0], wg.wh[2]};
    const int gox[2] = {lvl1 ? ox[1] : ox[0], ox[2]}, goy[2] = {lvl1 ? oy[1] : oy[0], oy[2]};
    const int gbase[2] = {lvl1 ? wg.base[1] : wg.base[0], wg.base[2]};
    // offsets and logits of the lane's own two samples, one head ahead
    float2 no0, no1;
    float nl0, nl1;
    auto issue_own = [&](int m) {
        const float *offp_g = row + m * (LP_ * 2);
        const float *lgp = row + M * LP_ * 2 + m * LP_;
        no0 = *reinterpret_cast<const float2 *>(offp_g + 2 * sj);
        no1 = *reinterpret_cast<const float2 *>(offp_g + 2 * i1);
        nl0 = lgp[sj];
        nl1 = lgp[i1];
    };
    float sw1[2], sw2[2], sw3[2], sw4[2], saw[2];
    int spix[2], smask[2], swin[2];
    bool fast;
    auto setup = [&]() {
        const float own_ox[2] = {no0.x, no1.x}, own_oy[2] = {no0.y, no1.y};
        const float own_lg[2] = {nl0, live1 ? nl1 : -INFINITY};
        float mx = fmaxf(own_lg[0], own_lg[1]);
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 8));
        float own_e[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) own_e[r] = expf(own_lg[r] - mx);
        float den = 0.f;
#pragma unroll
        for (int i = 0; i < LP_; ++i) den += __shfl(own_e[i >> 3], i & 7, 8);
        const float inv = 1.f / den;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int H = gH[r], W = gW[r];
            const float lx = ref_x + own_ox[r] / (float)W, ly = ref_y + own_oy[r] / (float)H;
            const float h_im = ly * H - 0.5f, w_im = lx * W - 0.5f;
            const bool in = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;   // cuh:293
            const int h0 = (int)floorf(h_im), w0 = (int)floorf(w_im), h1 = h0 + 1, w1 = w0 + 1;
            const float lh = h_im - h0, lw = w_im - w0, hh = 1.f - lh, hw = 1.f - lw;
            sw1[r] = hh * hw; sw2[r] = hh * lw; sw3[r] = lh * hw; sw4[r] = lh * lw;
            saw[r] = own_e[r] * inv;
            spix[r] = h0 * W + w0;
            smask[r] = !in ? 0 : ((h0 >= 0 && w0 >= 0) ? 1 : 0) | ((h0 >= 0 && w1 <= W - 1) ? 2 : 0) | ((h1 <= H - 1 && w0 >= 0) ? 4 : 0) |
                                 ((h1 <= H - 1 && w1 <= W - 1) ? 8 : 0);
            const int wx = w0 - gox[r], wy = h0 - goy[r];
            const bool inw = wx >= 0 && wx + 1 < gww[r] && wy >= 0 && wy + 1 < gwh[r];
            swin[r] = inw ? (gbase[r] + wy * gww[r] + wx) * 8 + c : -1;      // float4 index of the upper left tap, this lane's channel column
        }
        // Branch-free form when every sample of the wave has its 2 x 2 footprint inside the windows.  A sample outside the map
        // (`in` false, skipped by the general path) then reads zeros (staged for pixels outside the map) or, at h_im == -1 /
        // w_im == -1 exactly, map pixels under a weight of exactly 0: it adds +0 either way.
        fast = __all((swin[0] >= 0) && (!live1 || swin[1] >= 0));
    };
    f32x4 acc;
    auto samples = [&](int m, auto FIRST, auto LAST) {
        constexpr int i_lo = decltype(FIRST)::value, i_hi = decltype(LAST)::value;
        if (fast) {
#pragma unroll
            for (int i = i_lo; i < i_hi; ++i) {
                const int r = i >> 3, src = i & 7, l = i / P_;
                const int wi = __shfl(swin[r], src, 8) - src + c;       // the owner's index is for its own channel column
                const float c1 = __shfl(sw1[r], src, 8), c2 = __shfl(sw2[r], src, 8), c3 = __shfl(sw3[r], src, 8), c4 = __shfl(sw4[r], src, 8);
                const float aw = __shfl(saw[r], src, 8);
                const f32x4 *pw = win + wi;
                const f32x4 *pl = pw + wg.ww[l] * 8;
                const f32x4 v1 = pw[0], v2 = pw[8], v3 = pl[0], v4 = pl[8];
                acc += (c1 * v1 + c2 * v2 + c3 * v3 + c4 * v4) * aw;  // cuh:85-88, :299
            }
        } else {
#pragma unroll
            for (int i = i_lo; i < i_hi; ++i) {
                const int r = i >> 3, src = i & 7, l = i / P_;
                const int W = lv.W[l];
                const int mask = __shfl(smask[r], src, 8);
                if (mask == 0) continue;                           // uniform over the query's 8 lanes
                const int wi = __shfl(swin[r], src, 8);
                const float c1 = __shfl(sw1[r], src, 8), c2 = __shfl(sw2[r], src, 8), c3 = __shfl(sw3[r], src, 8), c4 = __shfl(sw4[r], src, 8);
                const float aw = __shfl(saw[r], src, 8);
                f32x4 v1 = f32x4(0.f), v2 = f32x4(0.f), v3 = f32x4(0.f), v4 = f32x4(0.f);
                if (wi >= 0) {
                    const f32x4 *pw = win + (wi - src + c);
                    const int up = wg.ww[l] * 8;
                    v1 = pw[0]; v2 = pw[8]; v3 = pw[up]; v4 = pw[up + 8];
                } else {
                    const int pix = __shfl(spix[r], src, 8);
                    const float *p1 = vn + ((long)lv.start[l] + pix) * ldv + m * D + c * 4;
                    if (mask & 1) v1 = *reinterpret_cast<const f32x4 *>(p1);
                    if (mask & 2) v2 = *reinterpret_cast<const f32x4 *>(p1 + ldv);
                    if (mask & 4) v3 = *reinterpret_cast<const f32x4 *>(p1 + (long)W * ldv);
                    if (mask & 8) v4 = *reinterpret_cast<const f32x4 *>(p1 + (long)(W + 1) * ldv);
                }
                acc += (c1 * v1 + c2 * v2 + c3 * v3 + c4 * v4) * aw;  // cuh:85-88, :299
            }
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I8 = std::integral_constant<int, 2 * P_>;
    using I12 = std::integral_constant<int, LP_>;

    for (int i = tid; i < (wg.basey + ((wg.ny + 7) & ~7)) * 8; i += 512) win[i] = f32x4(0.f);
    issue_own(0);
    __syncthreads();                                        // zeros written
    issue_x(0);
    setup();
    for (int m = 0; m < M; ++m) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's X lines of head m have landed
        __syncthreads();                                    // X(m) complete; every wave is done with Y(m - 1)
        issue_y(m);
        if (m + 1 < M) issue_own(m + 1);
        __builtin_amdgcn_sched_barrier(0);
        acc = f32x4(0.f);
        samples(m, I0{}, I8{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                    // Y(m) complete; every wave is done with X(m)
        if (m + 1 < M) issue_x(m + 1);
        __builtin_amdgcn_sched_barrier(0);
        samples(m, I8{}, I12{});
        if (valid) *reinterpret_cast<f32x4 *>(out + (((long)n * S + q) * M + m) * D + c * 4) = acc;
        if (m + 1 < M) setup();
    }
}

// Backward (ms_deform_im2col_cuda.cuh:92-164 formulas; the reference launches 32-thread blocks with a serial
// shared-memory sum, :306-408).  Here: one lane per (query, head, channel), D == 32 lanes = half a wavefront
// per (query, head); grad_sampling_loc / grad_attn_weight are reduced across the 32 channel lanes with
// DPP/shuffle butterflies (no LDS, no serial loop); grad_value uses float atomics, issued as 128-B row
// segments (two per wave-instruction).
template <class G>
__global__ __launch_bounds__(256) void msda_bwd_kernel(const float *__restrict__ value, G geo,
                                                       const float *__restrict__ loc, const float *__restrict__ aw,
                                                       const float *__restrict__ gout, int S, int M, int L, int Lq,
                                                       int P, float *__restrict__ gvalue, float *__restrict__ gloc,
                                                       float *__restrict__ gaw)
{
    constexpr int D = 32;
    const Levels &lv = geo.levels();
    const int n = blockIdx.y;
    const long item = (long)blockIdx.x * 256 + threadIdx.x;
    const bool live = item < (long)Lq * M * D;
    const long it = live ? item : 0;
    const int d = (int)(it % D);
    const int m = (int)((it / D) % M);
    const int q = (int)(it / ((long)D * M));
    const long qm = ((long)n * Lq + q) * M + m;
    const float tg = live ? gout[qm * D + d] : 0.f;
    const long rowstride = (long)M * D;
    for (int l = 0; l < L; ++l) {
        const int H = lv.H[l], W = lv.W[l];
        const long vb = ((long)n * S + lv.start[l]) * rowstride + m * D + d;
        for (int p = 0; p < P; ++p) {
            const long wi = (qm * L + l) * P + p;
            const float lx = loc[2 * wi], ly = loc[2 * wi + 1], a = aw[wi];
            const float h_im = ly * H - 0.5f, w_im = lx * W - 0.5f;
            float gw = 0.f, gx = 0.f, gy = 0.f;
            if (live && h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
                const int h0 = (int)floorf(h_im), w0 = (int)floorf(w_im), h1 = h0 + 1, w1 = w0 + 1;
                const float lh = h_im - h0, lw = w_im - w0, hh = 1.f - lh, hw = 1.f - lw;
                const float tgv = tg * a;
                float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f, ghw = 0.f, gww = 0.f;
                if (h0 >= 0 && w0 >= 0) {
                    const long ix = vb + ((long)h0 * W + w0) * rowstride;
                    v1 = value[ix]; ghw -= hw * v1; gww -= hh * v1; atomicAdd(gvalue + ix, hh * hw * tgv);
                }
                if (h0 >= 0 && w1 <= W - 1) {
                    const long ix = vb + ((long)h0 * W + w1) * rowstride;
                    v2 = value[ix]; ghw -= lw * v2; gww += hh * v2; atomicAdd(gvalue + ix, hh * lw * tgv);
                }
                if (h1 <= H - 1 && w0 >= 0) {
                    const long ix = vb + ((long)h1 * W + w0) * rowstride;
                    v3 = value[ix]; ghw += hw * v3; gww -= lh * v3; atomicAdd(gvalue + ix, lh * hw * tgv);
                }
                if (h1 <= H - 1 && w1 <= W - 1) {
                    const long ix = vb + ((long)h1 * W + w1) * rowstride;
                    v4 = value[ix]; ghw += lw * v4; gww += lh * v4; atomicAdd(gvalue + ix, lh * lw * tgv);
                }
                gw = tg * (hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4);
                gx = (float)W * gww * tgv;
                gy = (float)H * ghw * tgv;
            }
            // reduce over the 32 channel lanes of this (query, head)
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) {
                gw += __shfl_xor(gw, o, 64);
                gx += __shfl_xor(gx, o, 64);
                gy += __shfl_xor(gy, o, 64);
            }
            if (live && d == 0) {
                gaw[wi] = gw;
                gloc[2 * wi] = gx;
                gloc[2 * wi + 1] = gy;
            }
        }
    }
}


// ---- atomic-free backward (semantics of ms_deform_im2col_cuda.cuh:306-408) ------------------------
// grad_value[n, s, m, :] = sum over the samples (q, l, p) whose bilinear cell has pixel s as one of its four corners of
// corner_weight * attn * grad_out[n, q, m, :].  The reference (and s2d_msda_backward_f32) scatters these with float atomics:
// 4 corners x 32 channels x 30 M samples = 3.8 G atomics per encoder layer at config c4, the chip's atomic ceiling (10.8 ms),
// and the sum order -- hence the bits -- changes from run to run.  Here the sampling graph is inverted once per call instead:
//   1. every sample is keyed by the bilinear cell (head, level, floor(y), floor(x)) it falls in; a histogram of the keys
//      (integer atomics: order-free) and its prefix sum give every cell's segment;
//   2. a stable radix sort (rocPRIM) of (cell key, sample id) pairs lists each cell's samples in ascending sample id;
//   3. a gather kernel owns one grad_value row (n, s, m, 32 channels) per half-wave, walks the four cells that have pixel s
//      as a corner, and adds weight * grad_out rows in that fixed order -- one 128-B store per row, no atomics, bitwise
//      reproducible.
// grad_loc / grad_attn stay query-owned (msda_bwd_loc_kernel): no scatter there.
template <class G>
__global__ __launch_bounds__(256) void msda_cell_key_kernel(const float *__restrict__ loc, G geo, unsigned int per_n, int M, int L,
                                                            int P, unsigned int *__restrict__ keys,
                                                            unsigned int *__restrict__ vals)
{
    const Levels &lv = geo.levels();
    const Cells &cl = geo.cells();
    const unsigned int kmax = geo.kmax();
    const unsigned int n = blockIdx.y;
    const unsigned int w = blockIdx.x * 256u + threadIdx.x;          // sample index inside frame n: (q * M + m) * LP + lp
    if (w >= per_n) return;
    const unsigned int LP = (unsigned int)(L * P);
    const unsigned int lp = w % LP, qm = w / LP, m = qm % (unsigned int)M;
    const int l = (int)(lp / (unsigned int)P);
    const int H = lv.H[l], W = lv.W[l];
    const unsigned int v = n * per_n + w;
    const float2 xy = *reinterpret_cast<const float2 *>(loc + 2L * v);
    const float h_im = xy.y * H - 0.5f, w_im = xy.x * W - 0.5f;
    unsigned int key = kmax;
    if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {      // cuh:293 / :367
        const int h0 = (int)floorf(h_im), w0 = (int)floorf(w_im);
        key = (n * (unsigned int)M + m) * (unsigned int)cl.tot + (unsigned int)(cl.base[l] + (h0 + 1) * (W + 1) + (w0 + 1));
    }
    keys[v] = key;
    vals[v] = v;
}

// After the sort: segment starts of every cell (off[k] = first sorted position whose key >= k, filled at the key changes -- no
// histogram, no scattered atomics) and one 16-B record per sorted sample {loc.x, loc.y, attention weight, grad_out row}, so
// that the gather reads a sample with one coalesced load instead of three dependent ones.
struct __attribute__((aligned(16))) SampleRec {
    float lx, ly, a;
    unsigned int row;
};

template <class G>
__global__ __launch_bounds__(256) void msda_cell_bounds_kernel(const unsigned int *__restrict__ skeys, const unsigned int *__restrict__ svals,
                                                               const float *__restrict__ loc, const float *__restrict__ aw, unsigned int nsamp,
                                                               G geo, unsigned int LP, int *__restrict__ off,
                                                               SampleRec *__restrict__ rec)
{
    const unsigned int kmax = geo.kmax();
    const unsigned int i = blockIdx.x * 256u + threadIdx.x;
    if (i > nsamp) return;
    const long kprev = i ? (long)skeys[i - 1] : -1L;
    const long k = i < nsamp ? (long)skeys[i] : (long)kmax + 1;       // position nsamp closes every remaining segment (off[.. kmax + 1])
    for (long kk = kprev + 1; kk <= k && kk <= (long)kmax + 1; ++kk) off[kk] = (int)i;
    if (i < nsamp) {
        const unsigned int v = svals[i];
        const float2 xy = *reinterpret_cast<const float2 *>(loc + 2L * v);
        SampleRec r;
        r.lx = xy.x; r.ly = xy.y; r.a = aw[v]; r.row = v / LP;
        rec[i] = r;
    }
}

// one grad_value row (n, s, m, 32 channels) per 8-lane group (16 B per lane): a wave = the 8 heads of one pixel.  A batch of
// 8 records is prepared by the group's lanes, then its 8 grad_out rows are loaded together (all in flight) and added in record
// order; slots beyond the segment carry weight 0 (row 0), which leaves the sum's bits unchanged.
template <class G>
__global__ __launch_bounds__(256) void msda_bwd_value_kernel(const float *__restrict__ gout, const SampleRec *__restrict__ rec,
                                                             const int *__restrict__ off, G geo, long ntgt, int S, int M, int L,
                                                             float *__restrict__ gvalue, long ldg)
{
    constexpr int D = 32;
    const Levels &lv = geo.levels();
    const Cells &cl = geo.cells();
    const long item = (long)blockIdx.x * 256 + threadIdx.x;
    const long t = item >> 3;                                       // target row (n, s, m)
    const int c = (int)(item & 7);
    if (t >= ntgt) return;                                          // whole 8-lane groups leave together
    float *gdst = gvalue + (t / M) * ldg + (t % M) * D + c * 4;       // row (n, s) of grad_value: M * D floats at stride ldg
    if (geo.bad()) {                                                // dev form with rejected shapes: zeros, no indexing by them
        *reinterpret_cast<f32x4 *>(gdst) = f32x4{0.f, 0.f, 0.f, 0.f};
        return;
    }
    const int m = (int)(t % M);
    const int s = (int)((t / M) % S);
    const long n = t / M / S;
    int l = 0;
    while (l + 1 < L && s >= lv.start[l + 1]) ++l;
    const int H = lv.H[l], W = lv.W[l];
    const int pix = s - (int)lv.start[l];
    const int y = pix / W, x = pix - y * W;
    const long kbase = (n * M + m) * (long)cl.tot + cl.base[l];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int dy = 1 - (k >> 1), dx = 1 - (k & 1);             // pixel s is corner (h0 + dy, w0 + dx) of cell (h0, w0) = (y - dy, x - dx)
        const int h0 = y - dy, w0 = x - dx;
        const long key = kbase + (long)(h0 + 1) * (W + 1) + (w0 + 1);
        const int beg = off[key], end = off[key + 1];
        for (int i = beg; i < end; i += 8) {
            const int idx = i + c;
            float wgt = 0.f;
            int grow = 0;
            if (idx < end) {                                        // lane c prepares record i + c
                const SampleRec r = rec[idx];
                const float lh = (r.ly * H - 0.5f) - (float)h0, lw = (r.lx * W - 0.5f) - (float)w0;
                wgt = (dy ? lh : 1.f - lh) * (dx ? lw : 1.f - lw) * r.a;
                grow = (int)r.row;
            }
            float wj[8];
            f32x4 g[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                wj[j] = __shfl(wgt, j, 8);
                const unsigned int rj = (unsigned int)__shfl(grow, j, 8);
                g[j] = *reinterpret_cast<const f32x4 *>(gout + (long)rj * D + c * 4);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += wj[j] * g[j];        // fixed order: ascending sample id inside the cell
        }
    }
    *reinterpret_cast<f32x4 *>(gdst) = acc;
}

// grad_sampling_loc / grad_attn_weight of the backward, query-owned (cuh:119-163 without the grad_value scatter): the forward's
// geometry -- a wave = one query's 8 heads x 8 lanes x 4 channels, every corner a 16-B load -- with the channel sums reduced
// over the 8 lanes of a head
template <class G>
__global__ __launch_bounds__(256) void msda_bwd_loc_kernel(const float *__restrict__ value, G geo, const float *__restrict__ loc,
                                                           const float *__restrict__ aw, const float *__restrict__ gout, int S, int M, int L,
                                                           int Lq, int P, int blk_per_n, float *__restrict__ gloc, float *__restrict__ gaw, long ldv)
{
    constexpr int D = 32;
    const Levels &lv = geo.levels();
    const int n = blockIdx.y;
    const int bid = xcd_band(blockIdx.x, blk_per_n);
    const long item = (long)bid * 256 + threadIdx.x;
    if (item >= (long)Lq * M * 8) return;                           // whole 8-lane groups leave together
    const int c = (int)(item & 7);
    const int m = (int)((item >> 3) % M);
    const int q = (int)((item >> 3) / M);
    const long qm = ((long)n * Lq + q) * M + m;
    const f32x4 tg = *reinterpret_cast<const f32x4 *>(gout + qm * D + c * 4);
    const long rowstride = ldv;
    const int LP = L * P;
    if (LP <= 16) {
        // as in the fused forward: lane j of the head's 8 prepares samples j and j + 8 (location / weight loads, floor, bilinear
        // weights, border tests) and the group reads them back with 8-lane shuffles
        const int j = c;
        float s_lh[2], s_lw[2], s_a[2];
        int s_pix[2], s_mask[2], s_H[2], s_W[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int i = j + 8 * r;
            const int ic = i < LP ? i : LP - 1;
            const int l = ic / P;
            const int H = lv.H[l], W = lv.W[l];
            const long wi = qm * LP + ic;
            const float2 xy = *reinterpret_cast<const float2 *>(loc + 2 * wi);
            s_a[r] = aw[wi];
            const float h_im = xy.y * H - 0.5f, w_im = xy.x * W - 0.5f;
            const bool in = i < LP && h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
            const int h0 = (int)floorf(h_im), w0 = (int)floorf(w_im), h1 = h0 + 1, w1 = w0 + 1;
            s_lh[r] = h_im - h0; s_lw[r] = w_im - w0;
            s_pix[r] = h0 * W + w0; s_H[r] = H; s_W[r] = W;
            s_mask[r] = !in ? 0 : ((h0 >= 0 && w0 >= 0) ? 1 : 0) | ((h0 >= 0 && w1 <= W - 1) ? 2 : 0) | ((h1 <= H - 1 && w0 >= 0) ? 4 : 0) |
                                  ((h1 <= H - 1 && w1 <= W - 1) ? 8 : 0);
        }
        for (int i = 0; i < LP; ++i) {
            const int r = i >> 3, src = i & 7;
            const int l = i / P;
            const int mask = __shfl(r ? s_mask[1] : s_mask[0], src, 8);
            float gw = 0.f, gx = 0.f, gy = 0.f;
            if (mask) {                                         // uniform over the head's 8 lanes
                const int W = lv.W[l], H = lv.H[l];
                const int pix = __shfl(r ? s_pix[1] : s_pix[0], src, 8);
                const float lh = __shfl(r ? s_lh[1] : s_lh[0], src, 8), lw = __shfl(r ? s_lw[1] : s_lw[0], src, 8);
                const float a = __shfl(r ? s_a[1] : s_a[0], src, 8);
                const float hh = 1.f - lh, hw = 1.f - lw;
                const float *vb = value + ((long)n * S + lv.start[l]) * rowstride + m * D + c * 4 + (long)pix * rowstride;
                f32x4 v1 = {0.f, 0.f, 0.f, 0.f}, v2 = v1, v3 = v1, v4 = v1;
                if (mask & 1) v1 = *reinterpret_cast<const f32x4 *>(vb);
                if (mask & 2) v2 = *reinterpret_cast<const f32x4 *>(vb + rowstride);
                if (mask & 4) v3 = *reinterpret_cast<const f32x4 *>(vb + (long)W * rowstride);
                if (mask & 8) v4 = *reinterpret_cast<const f32x4 *>(vb + (long)(W + 1) * rowstride);
                const float d1 = tg[0] * v1[0] + tg[1] * v1[1] + tg[2] * v1[2] + tg[3] * v1[3];
                const float d2 = tg[0] * v2[0] + tg[1] * v2[1] + tg[2] * v2[2] + tg[3] * v2[3];
                const float d3 = tg[0] * v3[0] + tg[1] * v3[1] + tg[2] * v3[2] + tg[3] * v3[3];
                const float d4 = tg[0] * v4[0] + tg[1] * v4[1] + tg[2] * v4[2] + tg[3] * v4[3];
                gw = hh * hw * d1 + hh * lw * d2 + lh * hw * d3 + lh * lw * d4;          // cuh:150-155 summed over this lane's channels
                gx = (float)W * a * (hh * (d2 - d1) + lh * (d4 - d3));                     // d/dx: W * sum_d tg_d * a * grad_w_weight
                gy = (float)H * a * (hw * (d3 - d1) + lw * (d4 - d2));
            }
#pragma unroll
            for (int o = 4; o > 0; o >>= 1) {                       // the head's 8 lanes
                gw += __shfl_xor(gw, o, 8);
                gx += __shfl_xor(gx, o, 8);
                gy += __shfl_xor(gy, o, 8);
            }
            if (c == 0) {
                const long wi = qm * LP + i;
                gaw[wi] = gw;
                gloc[2 * wi] = gx;
                gloc[2 * wi + 1] = gy;
            }
        }
        return;
    }
    for (int l = 0; l < L; ++l) {
        const int H = lv.H[l], W = lv.W[l];
        const float *vb = value + ((long)n * S + lv.start[l]) * rowstride + m * D + c * 4;
        for (int p = 0; p < P; ++p) {
            const long wi = (qm * L + l) * P + p;
            const float2 xy = *reinterpret_cast<const float2 *>(loc + 2 * wi);
            const float a = aw[wi];
            const float h_im = xy.y * H - 0.5f, w_im = xy.x * W - 0.5f;
            float gw = 0.f, gx = 0.f, gy = 0.f;
            if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
                const int h0 = (int)floorf(h_im), w0 = (int)floorf(w_im), h1 = h0 + 1, w1 = w0 + 1;
                const float lh = h_im - h0, lw = w_im - w0, hh = 1.f - lh, hw = 1.f - lw;
                f32x4 v1 = {0.f, 0.f, 0.f, 0.f}, v2 = v1, v3 = v1, v4 = v1;
                if (h0 >= 0 && w0 >= 0) v1 = *reinterpret_cast<const f32x4 *>(vb + ((long)h0 * W + w0) * rowstride);
                if (h0 >= 0 && w1 <= W - 1) v2 = *reinterpret_cast<const f32x4 *>(vb + ((long)h0 * W + w1) * rowstride);
                if (h1 <= H - 1 && w0 >= 0) v3 = *reinterpret_cast<const f32x4 *>(vb + ((long)h1 * W + w0) * rowstride);
                if (h1 <= H - 1 && w1 <= W - 1) v4 = *reinterpret_cast<const f32x4 *>(vb + ((long)h1 * W + w1) * rowstride);
                const float d1 = tg[0] * v1[0] + tg[1] * v1[1] + tg[2] * v1[2] + tg[3] * v1[3];
                const float d2 = tg[0] * v2[0] + tg[1] * v2[1] + tg[2] * v2[2] + tg[3] * v2[3];
                const float d3 = tg[0] * v3[0] + tg[1] * v3[1] + tg[2] * v3[2] + tg[3] * v3[3];
                const float d4 = tg[0] * v4[0] + tg[1] * v4[1] + tg[2] * v4[2] + tg[3] * v4[3];
                gw = hh * hw * d1 + hh * lw * d2 + lh * hw * d3 + lh * lw * d4;          // cuh:150-155 summed over this lane's channels
                gx = (float)W * a * (hh * (d2 - d1) + lh * (d4 - d3));                     // d/dx: W * sum_d tg_d * a * grad_w_weight
                gy = (float)H * a * (hw * (d3 - d1) + lw * (d4 - d2));
            }
#pragma unroll
            for (int o = 4; o > 0; o >>= 1) {                       // the head's 8 lanes
                gw += __shfl_xor(gw, o, 8);
                gx += __shfl_xor(gx, o, 8);
                gy += __shfl_xor(gy, o, 8);
            }
            if (c == 0) {
                gaw[wi] = gw;
                gloc[2 * wi] = gx;
                gloc[2 * wi + 1] = gy;
            }
        }
    }
}

int fill_levels(Levels &lv, const int64_t *shapes, const int64_t *lsi, int L, long S)
{
    if (L < 1 || L > MAX_L) return S2D_ERR_ARG;
    long tot = 0;
    for (int l = 0; l < L; ++l) {
        lv.H[l] = (int)shapes[2 * l];
        lv.W[l] = (int)shapes[2 * l + 1];
        lv.start[l] = lsi ? (long)lsi[l] : tot;
        if (lv.H[l] <= 0 || lv.W[l] <= 0 || lv.start[l] < 0 || lv.start[l] + (long)lv.H[l] * lv.W[l] > S) return S2D_ERR_ARG;
        tot += (long)lv.H[l] * lv.W[l];
    }
    return S2D_OK;
}

// ---- backward of the fused form: the drop-in backward kernel works on explicit sampling locations
// and softmaxed weights, so the fused projection output is first expanded (prep) and the gradients it returns are chained
// back to the raw offsets / logits (chain): d_off = d_loc / (W_l, H_l); d_logit = a * (d_a - sum_j a_j d_a_j).
template <int LP_>
__global__ __launch_bounds__(256) void msda_fused_prep_kernel(const float *__restrict__ oa, int ldoa, Levels lv, int S, int M, int L, int P,
                                                              float *__restrict__ loc, float *__restrict__ attn)
{
    // one thread per (query, head); its 2 * LP offsets and LP logits are contiguous and 16-B aligned (LP % 4 == 0, ldoa % 4 == 0):
    // 16-B loads and stores (the scalar form ran at 0.6 TB/s: 72 strided 4-B accesses per thread)
    const int n = blockIdx.y;
    const long item = (long)blockIdx.x * 256 + threadIdx.x;
    if (item >= (long)S * M) return;
    const int m = (int)(item % M), q = (int)(item / M);
    int lq = 0;
    while (lq + 1 < L && q >= lv.start[lq + 1]) ++lq;
    const int qi = q - (int)lv.start[lq];
    const int qy = qi / lv.W[lq], qx = qi - qy * lv.W[lq];
    const float ref_x = ((float)qx + 0.5f) / (float)lv.W[lq], ref_y = ((float)qy + 0.5f) / (float)lv.H[lq];
    const float *row = oa + ((long)n * S + q) * ldoa;
    const float *offp = row + m * (LP_ * 2), *lgp = row + M * LP_ * 2 + m * LP_;
    float *lo = loc + (((long)n * S + q) * M + m) * LP_ * 2, *ao = attn + (((long)n * S + q) * M + m) * LP_;
    float off[LP_ * 2], lg[LP_];
#pragma unroll
    for (int i = 0; i < LP_ * 2; i += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(offp + i);
        off[i] = v[0]; off[i + 1] = v[1]; off[i + 2] = v[2]; off[i + 3] = v[3];
    }
#pragma unroll
    for (int i = 0; i < LP_; i += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(lgp + i);
        lg[i] = v[0]; lg[i + 1] = v[1]; lg[i + 2] = v[2]; lg[i + 3] = v[3];
    }
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < LP_; ++i) mx = fmaxf(mx, lg[i]);
    float den = 0.f;
#pragma unroll
    for (int i = 0; i < LP_; ++i) den += expf(lg[i] - mx);
    const float inv = 1.f / den;
#pragma unroll
    for (int i = 0; i < LP_; ++i) {
        const int l = i / P;
        lg[i] = expf(lg[i] - mx) * inv;
        off[2 * i] = ref_x + off[2 * i] / (float)lv.W[l];
        off[2 * i + 1] = ref_y + off[2 * i + 1] / (float)lv.H[l];
    }
#pragma unroll
    for (int i = 0; i < LP_ * 2; i += 4) *reinterpret_cast<f32x4 *>(lo + i) = f32x4{off[i], off[i + 1], off[i + 2], off[i + 3]};
#pragma unroll
    for (int i = 0; i < LP_; i += 4) *reinterpret_cast<f32x4 *>(ao + i) = f32x4{lg[i], lg[i + 1], lg[i + 2], lg[i + 3]};
}

template <int LP_>
__global__ __launch_bounds__(256) void msda_fused_chain_kernel(const float *__restrict__ attn, const float *__restrict__ gloc,
                                                               const float *__restrict__ gattn, Levels lv, int S, int M, int L, int P,
                                                               float *__restrict__ doa, int ldd)
{
    const int n = blockIdx.y;
    const long item = (long)blockIdx.x * 256 + threadIdx.x;
    if (item >= (long)S * M) return;
    const int m = (int)(item % M), q = (int)(item / M);
    const long base = (((long)n * S + q) * M + m) * LP_;
    float *row = doa + ((long)n * S + q) * ldd;
    float a[LP_], ga[LP_], gl[LP_ * 2];
#pragma unroll
    for (int i = 0; i < LP_; i += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(attn + base + i), w = *reinterpret_cast<const f32x4 *>(gattn + base + i);
        a[i] = v[0]; a[i + 1] = v[1]; a[i + 2] = v[2]; a[i + 3] = v[3];
        ga[i] = w[0]; ga[i + 1] = w[1]; ga[i + 2] = w[2]; ga[i + 3] = w[3];
    }
#pragma unroll
    for (int i = 0; i < LP_ * 2; i += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(gloc + base * 2 + i);
        gl[i] = v[0]; gl[i + 1] = v[1]; gl[i + 2] = v[2]; gl[i + 3] = v[3];
    }
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < LP_; ++i) dot += a[i] * ga[i];
#pragma unroll
    for (int i = 0; i < LP_; ++i) {
        const int l = i / P;
        gl[2 * i] = gl[2 * i] / (float)lv.W[l];
        gl[2 * i + 1] = gl[2 * i + 1] / (float)lv.H[l];
        a[i] = a[i] * (ga[i] - dot);
    }
    float *od = row + m * (LP_ * 2), *ol = row + M * LP_ * 2 + m * LP_;
#pragma unroll
    for (int i = 0; i < LP_ * 2; i += 4) *reinterpret_cast<f32x4 *>(od + i) = f32x4{gl[i], gl[i + 1], gl[i + 2], gl[i + 3]};
#pragma unroll
    for (int i = 0; i < LP_; i += 4) *reinterpret_cast<f32x4 *>(ol + i) = f32x4{a[i], a[i + 1], a[i + 2], a[i + 3]};
}

// ---- drop-in op, host-shape and device-shape forms share these launchers ---------------------------------------------------
static int fill_cells(Cells &cl, const Levels &lv, int L)
{
    long tot = 0;
    for (int l = 0; l < L; ++l) {
        cl.base[l] = (int)tot;
        tot += (long)(lv.H[l] + 1) * (lv.W[l] + 1);
    }
    if (tot >= (1L << 31)) return S2D_ERR_ARG;
    cl.tot = (int)tot;
    return S2D_OK;
}

static int host_geom(GeomVal &gv, const int64_t *shapes, const int64_t *lsi, int L, long S, int N, int M)
{
    if (int e = fill_levels(gv.g.lv, shapes, lsi, L, S)) return e;
    if (int e = fill_cells(gv.g.cl, gv.g.lv, L)) return e;
    const long ncell = (long)N * M * gv.g.cl.tot;
    if (ncell >= (1L << 32) - 1) return S2D_ERR_ARG;
    gv.g.kmax = (unsigned int)ncell;
    gv.g.err = 0;
    return S2D_OK;
}

template <class G>
static int launch_forward(const G &geo, const float *value, const float *loc, const float *attn_w, int N, int S, int M, int D, int L, int Lq,
                          int P, float *out, hipStream_t stream)
{
    if ((D & 3) == 0) {
        const long items = (long)Lq * M * (D / 4);
        const int nb = cdiv(items, 256);
        hipLaunchKernelGGL((msda_fwd_kernel<4, G>), dim3(nb, N), dim3(256), 0, stream, value, geo, loc, attn_w, S, M, D, L, Lq, P, nb, out);
    } else {
        const long items = (long)Lq * M * D;
        const int nb = cdiv(items, 256);
        hipLaunchKernelGGL((msda_fwd_kernel<1, G>), dim3(nb, N), dim3(256), 0, stream, value, geo, loc, attn_w, S, M, D, L, Lq, P, nb, out);
    }
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

template <class G>
static int launch_backward_atomic(const G &geo, const float *value, const float *loc, const float *attn_w, const float *grad_out, int N,
                                  int S, int M, int L, int Lq, int P, float *grad_value, float *grad_loc, float *grad_attn_w,
                                  hipStream_t stream)
{
    if (s2d_zero_async(grad_value, sizeof(float) * (size_t)N * S * M * 32, stream) != S2D_OK) return S2D_ERR_LAUNCH;
    if (s2d_zero_async(grad_loc, sizeof(float) * (size_t)N * Lq * M * L * P * 2, stream) != S2D_OK) return S2D_ERR_LAUNCH;
    if (s2d_zero_async(grad_attn_w, sizeof(float) * (size_t)N * Lq * M * L * P, stream) != S2D_OK) return S2D_ERR_LAUNCH;
    const long items = (long)Lq * M * 32;
    hipLaunchKernelGGL((msda_bwd_kernel<G>), dim3(cdiv(items, 256), N), dim3(256), 0, stream, value, geo, loc, attn_w, grad_out, S, M, L, Lq,
                       P, grad_value, grad_loc, grad_attn_w);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

// Workspace of the atomic-free backward: four u32 arrays of one entry per sample (sort ping-pong), the 16-B sample records, the
// cell offsets, and rocPRIM's temporaries.
static size_t sorted_ws_layout(long nsamp, long ncell, size_t *o_keys, size_t *o_vals, size_t *o_rec, size_t *o_off, size_t *o_tmp, size_t *tmp_bytes)
{
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t o = 0;
    o_keys[0] = o; o = al(o + (size_t)nsamp * 4);
    o_keys[1] = o; o = al(o + (size_t)nsamp * 4);
    o_vals[0] = o; o = al(o + (size_t)nsamp * 4);
    o_vals[1] = o; o = al(o + (size_t)nsamp * 4);
    *o_rec = o; o = al(o + (size_t)nsamp * 16);
    *o_off = o; o = al(o + (size_t)(ncell + 2) * 4);
    *o_tmp = o;
    *tmp_bytes = (size_t)(4 * nsamp + (1L << 20)) * 4;
    return o + *tmp_bytes;
}

// ncell_cap: the number of cells the workspace was sized for (exact for host shapes, the bound of s2d_msda_dev_* for device shapes;
// it only fixes the offsets array's extent and the number of key bits sorted)
template <class G>
static int launch_backward_sorted(const G &geo, long ncell_cap, const float *value, const float *loc, const float *attn_w,
                                  const float *grad_out, int N, int S, int M, int L, int Lq, int P, float *grad_value, float *grad_loc,
                                  float *grad_attn_w, char *ws, long workspace_bytes, hipStream_t stream, long ldv = 0, long ldg = 0)
{
    if (ldv == 0) ldv = (long)M * 32;                        // rows of value / grad_value: M * D floats, contiguous unless a stride is given
    if (ldg == 0) ldg = (long)M * 32;
    const long nsamp = (long)N * Lq * M * L * P;
    if (nsamp >= (1L << 31) || ncell_cap >= (1L << 32) - 1) return S2D_ERR_ARG;      // segment offsets are int
    size_t ok[2], ov[2], orec, oo, ot, tb;
    if ((long)sorted_ws_layout(nsamp, ncell_cap, ok, ov, &orec, &oo, &ot, &tb) > workspace_bytes) return S2D_ERR_ARG;
    unsigned int *keys_in = (unsigned int *)(ws + ok[0]), *keys_out = (unsigned int *)(ws + ok[1]);
    unsigned int *vals_in = (unsigned int *)(ws + ov[0]), *vals_out = (unsigned int *)(ws + ov[1]);
    SampleRec *rec = (SampleRec *)(ws + orec);
    int *off = (int *)(ws + oo);
    const unsigned int per_n = (unsigned int)((long)Lq * M * L * P);
    hipLaunchKernelGGL((msda_cell_key_kernel<G>), dim3(cdiv(per_n, 256), N), dim3(256), 0, stream, loc, geo, per_n, M, L, P, keys_in, vals_in);
    S2D_CHECK_LAUNCH();
    int bits = 1;
    while ((1L << bits) <= ncell_cap) ++bits;                // keys 0 .. kmax (kmax = "outside every map"), kmax <= ncell_cap
    if (int e = s2d_radix_sort_pairs_u32(keys_in, keys_out, vals_in, vals_out, (size_t)nsamp, bits, ws + ot, tb, stream)) return e;
    hipLaunchKernelGGL((msda_cell_bounds_kernel<G>), dim3(cdiv(nsamp + 1, 256)), dim3(256), 0, stream, keys_out, vals_out, loc, attn_w,
                       (unsigned int)nsamp, geo, (unsigned int)(L * P), off, rec);
    S2D_CHECK_LAUNCH();
    const long ntgt = (long)N * S * M;
    hipLaunchKernelGGL((msda_bwd_value_kernel<G>), dim3(cdiv(ntgt * 8, 256)), dim3(256), 0, stream, grad_out, rec, off, geo, ntgt, S, M, L,
                       grad_value, ldg);
    S2D_CHECK_LAUNCH();
    if (!grad_loc && !grad_attn_w) return S2D_OK;             // grad_value only (the fused backward takes the query-owned half in its own launch)
    if (!grad_loc || !grad_attn_w) return S2D_ERR_ARG;
    const int nb = cdiv((long)Lq * M * 8, 256);
    hipLaunchKernelGGL((msda_bwd_loc_kernel<G>), dim3(nb, N), dim3(256), 0, stream, value, geo, loc, attn_w, grad_out, S, M, L, Lq, P, nb,
                       grad_loc, grad_attn_w, ldv);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

// Device-shape form: one thread turns the reference op's two int64 DEVICE tensors into the Geom the kernels read (the checks of
// fill_levels, on the device: a geometry that fails them becomes "every level empty" + err = 1, so no kernel indexes by it).
__global__ void msda_geom_kernel(const int64_t *__restrict__ shapes, const int64_t *__restrict__ lsi, int L, long S, int N, int M,
                                 Geom *__restrict__ g, int *sticky)
{
    if (threadIdx.x | blockIdx.x) return;
    bool ok = true;
    long tot = 0, cells = 0;
    for (int l = 0; l < MAX_L; ++l) {
        long H = 0, W = 0, st = 0;
        if (l < L) {
            H = shapes[2 * l]; W = shapes[2 * l + 1]; st = lsi ? lsi[l] : tot;
            ok = ok && H > 0 && W > 0 && H < (1L << 30) && W < (1L << 30) && st >= 0 && H * W <= S && st + H * W <= S;
        }
        g->lv.H[l] = (int)H; g->lv.W[l] = (int)W; g->lv.start[l] = st;
        g->cl.base[l] = (int)cells;
        if (l < L) { tot += H * W; cells += (H + 1) * (W + 1); }
    }
    // overlapping levels (sum of H*W > S, e.g. every level_start_index 0) pass the per-level test above but make more bilinear cells
    // than the workspace and the sort-key width were sized for (dev_ncell_cap = N*M*(2S + 2L) assumes sum(H*W) <= S; every H, W >= 1
    // gives (H+1)(W+1) <= 2HW + 2): such geometry takes the rejected path (zeros + flag)
    ok = ok && tot <= S && cells <= 2 * S + 2 * (long)L;
    ok = ok && (long)N * M * cells < (1L << 32) - 1;
    if (!ok) {
        cells = 0;
        for (int l = 0; l < MAX_L; ++l) {
            g->lv.H[l] = 0; g->lv.W[l] = 0; g->lv.start[l] = 0;
            g->cl.base[l] = (int)cells;
            if (l < L) cells += 1;
        }
    }
    g->cl.tot = (int)cells;
    g->kmax = (unsigned int)((long)N * M * cells);
    g->err = ok ? 0 : 1;
    // the process's error word (s2d_msda_dev_error_word; pinned host memory in the drop-in module): set, never cleared, by a rejected
    // call, so that the host learns of it without synchronising -- by its next look at the word
    if (!ok && sticky) __hip_atomic_store(sticky, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

static long dev_ncell_cap(int N, int S, int M, int L)
{
    // sum over levels of (H + 1)(W + 1) with sum H W <= S and H, W >= 1:  (H + 1)(W + 1) <= 2 H W + 2
    return (long)N * M * (2L * S + 2L * L);
}

static std::atomic<int *> g_error_word{nullptr};     // s2d_msda_dev_error_word

static int launch_geom(const int64_t *shapes_dev, const int64_t *lsi_dev, int L, long S, int N, int M, void *workspace, hipStream_t stream)
{
    if (L < 1 || L > MAX_L || !shapes_dev || !workspace || (reinterpret_cast<uintptr_t>(workspace) & 15)) return S2D_ERR_ARG;
    hipLaunchKernelGGL(msda_geom_kernel, dim3(1), dim3(64), 0, stream, shapes_dev, lsi_dev, L, S, N, M, reinterpret_cast<Geom *>(workspace),
                       g_error_word.load(std::memory_order_relaxed));
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

constexpr long GEOM_BYTES = 256;
static_assert(sizeof(Geom) <= GEOM_BYTES, "Geom outgrew its workspace slot");

// ---- float64 instantiation of the drop-in op (ops/src/cuda/ms_deform_attn_cuda.cu:69, :137: AT_DISPATCH_FLOATING_TYPES; the reference's
// ops/test.py gradchecks the op in double).  Not a hot path: one thread per (n, q, m, channel), shapes read from the DEVICE tensors, a
// level that does not lie inside S contributes nothing; the backward scatters grad_value with double atomics (like the reference's
// col2im kernels) and adds the channel sums of grad_sampling_loc / grad_attn_weight with double atomics too.
__device__ __forceinline__ bool level_ok_f64(const int64_t *shapes, const int64_t *lsi, int l, long S, int &H, int &W, long &st)
{
    const int64_t h = shapes[2 * l], w = shapes[2 * l + 1], s0 = lsi[l];
    H = (int)h; W = (int)w; st = (long)s0;
    return h > 0 && w > 0 && h < (1 << 20) && w < (1 << 20) && s0 >= 0 && s0 + h * w <= S;
}

template <bool BWD>
__global__ __launch_bounds__(256) void msda_f64_kernel(const double *__restrict__ value, const int64_t *__restrict__ shapes,
                                                       const int64_t *__restrict__ lsi, const double *__restrict__ loc,
                                                       const double *__restrict__ aw, const double *__restrict__ gout, long S, int M, int D,
                                                       int L, long Lq, int P, long total, double *__restrict__ out,
                                                       double *__restrict__ gvalue, double *__restrict__ gloc, double *__restrict__ gaw,
                                                       int *__restrict__ err_word)
{
    const long item = (long)blockIdx.x * 256 + threadIdx.x;
    if (item >= total) return;
    if (item == 0 && err_word) {                              // the same sticky process word the f32 geometry check sets (s2d_msda_dev_error_word):
        for (int l = 0; l < L; ++l) {                          // a rejected level must not pass as zeros (a gradcheck in double would "pass")
            int H, W; long st;
            if (!level_ok_f64(shapes, lsi, l, S, H, W, st)) { *err_word = 1; break; }
        }
    }
    const int d = (int)(item % D);
    const int m = (int)((item / D) % M);
    const long q = (item / ((long)D * M)) % Lq;
    const long n = item / ((long)D * M * Lq);
    const long qm = (n * Lq + q) * M + m;
    const double go = BWD ? gout[qm * D + d] : 0.0;
    double acc = 0.0;
    for (int l = 0; l < L; ++l) {
        int H, W; long st;
        if (!level_ok_f64(shapes, lsi, l, S, H, W, st)) continue;
        const double *vb = value + ((n * S + st) * M + m) * D + d;
        double *gb = BWD ? gvalue + ((n * S + st) * M + m) * D + d : nullptr;
        const long rs = (long)M * D;
        for (int p = 0; p < P; ++p) {
            const long wi = (qm * L + l) * P + p;
            const double lx = loc[2 * wi], ly = loc[2 * wi + 1], a = aw[wi];
            const double h_im = ly * H - 0.5, w_im = lx * W - 0.5;                    // cuh:262-263
            if (!(h_im > -1 && w_im > -1 && h_im < H && w_im < W)) continue;           // cuh:265
            const int h0 = (int)floor(h_im), w0 = (int)floor(w_im), h1 = h0 + 1, w1 = w0 + 1;
            const double lh = h_im - h0, lw = w_im - w0, hh = 1 - lh, hw = 1 - lw;
            const bool o1 = h0 >= 0 && w0 >= 0, o2 = h0 >= 0 && w1 <= W - 1, o3 = h1 <= H - 1 && w0 >= 0, o4 = h1 <= H - 1 && w1 <= W - 1;
            const double v1 = o1 ? vb[((long)h0 * W + w0) * rs] : 0.0, v2 = o2 ? vb[((long)h0 * W + w1) * rs] : 0.0;
            const double v3 = o3 ? vb[((long)h1 * W + w0) * rs] : 0.0, v4 = o4 ? vb[((long)h1 * W + w1) * rs] : 0.0;
            const double w1_ = hh * hw, w2_ = hh * lw, w3_ = lh * hw, w4_ = lh * lw;
            const double val = w1_ * v1 + w2_ * v2 + w3_ * v3 + w4_ * v4;             // cuh:19-63
            if (!BWD) { acc += val * a; continue; }
            const double tg = go * a;                                                  // cuh:66-137
            if (o1) atomicAdd(gb + ((long)h0 * W + w0) * rs, w1_ * tg);
            if (o2) atomicAdd(gb + ((long)h0 * W + w1) * rs, w2_ * tg);
            if (o3) atomicAdd(gb + ((long)h1 * W + w0) * rs, w3_ * tg);
            if (o4) atomicAdd(gb + ((long)h1 * W + w1) * rs, w4_ * tg);
            const double gh = -hw * v1 - lw * v2 + hw * v3 + lw * v4, gw = -hh * v1 + hh * v2 - lh * v3 + lh * v4;
            atomicAdd(gaw + wi, go * val);
            atomicAdd(gloc + 2 * wi, W * gw * tg);
            atomicAdd(gloc + 2 * wi + 1, H * gh * tg);
        }
    }
    if (!BWD) out[qm * D + d] = acc;
}

}  // namespace

extern "C" {

int s2d_msda_forward_f32(const float *value, const int64_t *shapes_host, const int64_t *level_start_host,
                         const float *loc, const float *attn_w, int N, int S, int M, int D, int L, int Lq, int P,
                         float *out, hipStream_t stream)
{
    GeomVal gv;
    if (int e = host_geom(gv, shapes_host, level_start_host, L, S, N > 0 ? N : 1, M)) return e;
    if (N <= 0 || Lq <= 0) return S2D_OK;
    return launch_forward(gv, value, loc, attn_w, N, S, M, D, L, Lq, P, out, stream);
}

int s2d_msda_backward_f32(const float *value, const int64_t *shapes_host, const int64_t *level_start_host,
                          const float *loc, const float *attn_w, const float *grad_out, int N, int S, int M, int D,
                          int L, int Lq, int P, float *grad_value, float *grad_loc, float *grad_attn_w,
                          hipStream_t stream)
{
    GeomVal gv;
    if (int e = host_geom(gv, shapes_host, level_start_host, L, S, N > 0 ? N : 1, M)) return e;
    if (D != 32) return S2D_ERR_ARG;
    if (N <= 0 || Lq <= 0) return S2D_OK;
    return launch_backward_atomic(gv, value, loc, attn_w, grad_out, N, S, M, L, Lq, P, grad_value, grad_loc, grad_attn_w, stream);
}

long s2d_msda_backward_workspace_bytes(const int64_t *shapes_host, int N, int M, int L, int Lq, int P)
{
    long tot = 0;
    for (int l = 0; l < L; ++l) tot += (shapes_host[2 * l] + 1) * (shapes_host[2 * l + 1] + 1);
    size_t a[2], b[2], c, d, e, tb;
    return (long)sorted_ws_layout((long)N * Lq * M * L * P, (long)N * M * tot, a, b, &c, &d, &e, &tb);
}

int s2d_msda_backward_sorted_f32(const float *value, const int64_t *shapes_host, const int64_t *level_start_host,
                                 const float *loc, const float *attn_w, const float *grad_out, int N, int S, int M, int D,
                                 int L, int Lq, int P, float *grad_value, float *grad_loc, float *grad_attn_w, void *workspace,
                                 long workspace_bytes, hipStream_t stream)
{
    GeomVal gv;
    if (int e = host_geom(gv, shapes_host, level_start_host, L, S, N > 0 ? N : 1, M)) return e;
    if (D != 32 || !workspace) return S2D_ERR_ARG;
    if (N <= 0 || Lq <= 0) return S2D_OK;
    return launch_backward_sorted(gv, (long)gv.g.kmax, value, loc, attn_w, grad_out, N, S, M, L, Lq, P, grad_value, grad_loc, grad_attn_w,
                                  reinterpret_cast<char *>(workspace), workspace_bytes, stream);
}

int s2d_msda_backward_sorted_strided_f32(const float *value, long ldv, const int64_t *shapes_host, const int64_t *level_start_host,
                                         const float *loc, const float *attn_w, const float *grad_out, int N, int S, int M, int D,
                                         int L, int Lq, int P, float *grad_value, long ldg, float *grad_loc, float *grad_attn_w,
                                         void *workspace, long workspace_bytes, hipStream_t stream)
{
    GeomVal gv;
    if (int e = host_geom(gv, shapes_host, level_start_host, L, S, N > 0 ? N : 1, M)) return e;
    if (D != 32 || !workspace || ldv < (long)M * D || ldg < (long)M * D || (ldv & 3) || (ldg & 3)) return S2D_ERR_ARG;
    if (N <= 0 || Lq <= 0) return S2D_OK;
    return launch_backward_sorted(gv, (long)gv.g.kmax, value, loc, attn_w, grad_out, N, S, M, L, Lq, P, grad_value, grad_loc, grad_attn_w,
                                  reinterpret_cast<char *>(workspace), workspace_bytes, stream, ldv, ldg);
}

int s2d_msda_forward_dev_f64(const double *value, const int64_t *shapes_dev, const int64_t *level_start_dev, const double *loc,
                             const double *attn_w, int N, int S, int M, int D, int L, int Lq, int P, double *out, hipStream_t stream)
{
    if (N < 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Lq < 0 || P <= 0) return S2D_ERR_ARG;
    const long total = (long)N * Lq * M * D;
    if (total == 0) return S2D_OK;
    hipLaunchKernelGGL((msda_f64_kernel<false>), dim3(cdiv(total, 256)), dim3(256), 0, stream, value, shapes_dev, level_start_dev, loc, attn_w,
                       (const double *)nullptr, (long)S, M, D, L, (long)Lq, P, total, out, (double *)nullptr, (double *)nullptr, (double *)nullptr,
                       g_error_word.load(std::memory_order_relaxed));
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

int s2d_msda_backward_dev_f64(const double *value, const int64_t *shapes_dev, const int64_t *level_start_dev, const double *loc,
                              const double *attn_w, const double *grad_out, int N, int S, int M, int D, int L, int Lq, int P,
                              double *grad_value, double *grad_loc, double *grad_attn_w, hipStream_t stream)
{
    if (N < 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Lq < 0 || P <= 0) return S2D_ERR_ARG;
    if (s2d_zero_async(grad_value, sizeof(double) * (size_t)N * S * M * D, stream) != S2D_OK ||
        s2d_zero_async(grad_loc, sizeof(double) * (size_t)N * Lq * M * L * P * 2, stream) != S2D_OK ||
        s2d_zero_async(grad_attn_w, sizeof(double) * (size_t)N * Lq * M * L * P, stream) != S2D_OK)
        return S2D_ERR_LAUNCH;
    const long total = (long)N * Lq * M * D;
    if (total == 0) return S2D_OK;
    hipLaunchKernelGGL((msda_f64_kernel<true>), dim3(cdiv(total, 256)), dim3(256), 0, stream, value, shapes_dev, level_start_dev, loc, attn_w,
                       grad_out, (long)S, M, D, L, (long)Lq, P, total, (double *)nullptr, grad_value, grad_loc, grad_attn_w,
                       g_error_word.load(std::memory_order_relaxed));
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

long s2d_msda_dev_forward_workspace_bytes(void) { return GEOM_BYTES; }

int s2d_msda_forward_dev_f32(const float *value, const int64_t *shapes_dev, const int64_t *level_start_dev, const float *loc,
                             const float *attn_w, int N, int S, int M, int D, int L, int Lq, int P, float *out, void *workspace,
                             hipStream_t stream)
{
    if (N <= 0 || Lq <= 0) return S2D_OK;
    if (int e = launch_geom(shapes_dev, level_start_dev, L, S, N, M, workspace, stream)) return e;
    GeomPtr gp{reinterpret_cast<const Geom *>(workspace)};
    return launch_forward(gp, value, loc, attn_w, N, S, M, D, L, Lq, P, out, stream);
}

long s2d_msda_dev_backward_workspace_bytes(int N, int S, int M, int L, int Lq, int P)
{
    size_t a[2], b[2], c, d, e, tb;
    return GEOM_BYTES + (long)sorted_ws_layout((long)N * Lq * M * L * P, dev_ncell_cap(N, S, M, L), a, b, &c, &d, &e, &tb);
}

int s2d_msda_backward_dev_f32(const float *value, const int64_t *shapes_dev, const int64_t *level_start_dev, const float *loc,
                              const float *attn_w, const float *grad_out, int N, int S, int M, int D, int L, int Lq, int P,
                              float *grad_value, float *grad_loc, float *grad_attn_w, void *workspace, long workspace_bytes,
                              hipStream_t stream)
{
    if (D != 32 || !workspace || workspace_bytes < GEOM_BYTES) return S2D_ERR_ARG;
    if (N <= 0 || Lq <= 0) return S2D_OK;
    if (int e = launch_geom(shapes_dev, level_start_dev, L, S, N, M, workspace, stream)) return e;
    GeomPtr gp{reinterpret_cast<const Geom *>(workspace)};
    return launch_backward_sorted(gp, dev_ncell_cap(N, S, M, L), value, loc, attn_w, grad_out, N, S, M, L, Lq, P, grad_value, grad_loc,
                                  grad_attn_w, reinterpret_cast<char *>(workspace) + GEOM_BYTES, workspace_bytes - GEOM_BYTES, stream);
}

int s2d_msda_dev_error_word(int *word)
{
    g_error_word.store(word, std::memory_order_relaxed);
    return S2D_OK;
}

int s2d_msda_dev_status(const void *workspace, int *err_host, hipStream_t stream)
{
    // reads back Geom.err of a finished (or, after this call's own stream sync, the last enqueued) *_dev call: the one place the
    // device-shape form can report rejected shapes.  Synchronises `stream`: a debugging aid, not part of the data path.
    if (!workspace || !err_host) return S2D_ERR_ARG;
    Geom g;
    if (hipMemcpyAsync(&g, workspace, sizeof(Geom), hipMemcpyDeviceToHost, stream) != hipSuccess) return S2D_ERR_LAUNCH;
    if (hipStreamSynchronize(stream) != hipSuccess) return S2D_ERR_LAUNCH;
    *err_host = g.err;
    return S2D_OK;
}

int s2d_msda_fused_prep_f32(const float *offs_logits, int ldoa, const int64_t *shapes_host, int N, int S, int M, int L, int P, float *loc,
                            float *attn, hipStream_t stream)
{
    Levels lv;
    if (int e = fill_levels(lv, shapes_host, nullptr, L, S)) return e;
    if (ldoa < M * L * P * 3 || (ldoa & 3) || (reinterpret_cast<uintptr_t>(offs_logits) & 15)) return S2D_ERR_ARG;   // 16-B rows
    if (N <= 0) return S2D_OK;
    const dim3 grid(cdiv((long)S * M, 256), N);
    switch (L * P) {                                         // samples per (query, head): 12 in the S2D geometry
    case 4: hipLaunchKernelGGL(msda_fused_prep_kernel<4>, grid, dim3(256), 0, stream, offs_logits, ldoa, lv, S, M, L, P, loc, attn); break;
    case 8: hipLaunchKernelGGL(msda_fused_prep_kernel<8>, grid, dim3(256), 0, stream, offs_logits, ldoa, lv, S, M, L, P, loc, attn); break;
    case 12: hipLaunchKernelGGL(msda_fused_prep_kernel<12>, grid, dim3(256), 0, stream, offs_logits, ldoa, lv, S, M, L, P, loc, attn); break;
    case 16: hipLaunchKernelGGL(msda_fused_prep_kernel<16>, grid, dim3(256), 0, stream, offs_logits, ldoa, lv, S, M, L, P, loc, attn); break;
    default: return S2D_ERR_ARG;
    }
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

int s2d_msda_fused_chain_f32(const float *attn, const float *grad_loc, const float *grad_attn, const int64_t *shapes_host, int N, int S,
                             int M, int L, int P, float *d_offs_logits, int ldd, hipStream_t stream)
{
    Levels lv;
    if (int e = fill_levels(lv, shapes_host, nullptr, L, S)) return e;
    if (ldd < M * L * P * 3 || (ldd & 3) || (reinterpret_cast<uintptr_t>(d_offs_logits) & 15)) return S2D_ERR_ARG;
    if (N <= 0) return S2D_OK;
    const dim3 grid(cdiv((long)S * M, 256), N);
    switch (L * P) {
    case 4: hipLaunchKernelGGL(msda_fused_chain_kernel<4>, grid, dim3(256), 0, stream, attn, grad_loc, grad_attn, lv, S, M, L, P, d_offs_logits, ldd); break;
    case 8: hipLaunchKernelGGL(msda_fused_chain_kernel<8>, grid, dim3(256), 0, stream, attn, grad_loc, grad_attn, lv, S, M, L, P, d_offs_logits, ldd); break;
    case 12: hipLaunchKernelGGL(msda_fused_chain_kernel<12>, grid, dim3(256), 0, stream, attn, grad_loc, grad_attn, lv, S, M, L, P, d_offs_logits, ldd); break;
    case 16: hipLaunchKernelGGL(msda_fused_chain_kernel<16>, grid, dim3(256), 0, stream, attn, grad_loc, grad_attn, lv, S, M, L, P, d_offs_logits, ldd); break;
    default: return S2D_ERR_ARG;
    }
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

int s2d_msda_fused_forward_f32(const float *value, int ldv, const int64_t *shapes_host, const float *offs_logits, int ldoa,
                               int N, int S, int M, int D, int L, int P, float *out, hipStream_t stream)
{
    Levels lv;
    if (int e = fill_levels(lv, shapes_host, nullptr, L, S)) return e;
    if (D != 32 || L * P != 12 || ldoa < M * L * P * 3 || (ldoa & 3) || (ldv != -1 && (ldv < M * D || (ldv & 3)))) return S2D_ERR_ARG;  // the S2D geometry (msdeformattn.py:232-239)
    if (N <= 0) return S2D_OK;
    const long items = (long)S * M * 8;
    const int nb = cdiv(items, 256);
    if (ldv < 0)      // experiment switch of scripts/mb_msda.py: ldv = -1 reads a head-major value tensor [N][M][S][32]
        hipLaunchKernelGGL((msda_fused_kernel<12, true>), dim3(nb, N), dim3(256), 0, stream, value, 32, lv, offs_logits, ldoa, S, M, L, P, nb, out);
    else {
        static int share = -1, tiled = -1;
        if (share < 0) { const char *e = getenv("S2D_MSDA_SHARE"); share = e ? atoi(e) : 1; }
        if (tiled < 0) { const char *e = getenv("S2D_MSDA_TILED"); tiled = e ? atoi(e) : 1; }
        // opt-in (S2D_MSDA_WIN=1; read per call: tests and scripts/mb_msda_win.py switch it inside one process): faster only while
        // the offsets are as regular as at initialisation, see profiles/r3_experiments/not_adopted.txt
        int winmode = 0, winR = 4;
        if (const char *e = getenv("S2D_MSDA_WIN")) winmode = atoi(e);
        if (const char *r = getenv("S2D_MSDA_WIN_R")) winR = min(max(atoi(r), 1), 8);
        int lq = 0;
        for (int l = 1; l < L; ++l) if ((long)lv.H[l] * lv.W[l] > (long)lv.H[lq] * lv.W[lq]) lq = l;
        // opt-in (S2D_MSDA_HEAD=1, read per call): one head per workgroup with the coarsest level's plane of that head in LDS
        // (msda_fused_head_kernel), when that plane fits.  A third fewer L1 lines per (query, head), bit-identical, and SLOWER at c4
        // (0.77-0.80 vs 0.66-0.72 ms): profiles/r4_experiments/not_adopted.txt
        int ls = 0;
        for (int l = 1; l < L; ++l) if ((long)lv.H[l] * lv.W[l] < (long)lv.H[ls] * lv.W[ls]) ls = l;
        bool head_ok = false;
        if (const char *e = getenv("S2D_MSDA_HEAD")) head_ok = atoi(e) != 0 && L > 1 && (long)lv.H[ls] * lv.W[ls] * 128 <= 150 * 1024 && (ldv & 3) == 0;
        WinGeo wg;
        // the windowed kernel takes the last level's queries when that level is the finest and holds most of the pyramid
        bool win_ok = winmode && tiled && M == 8 && L == 3 && P == 4 && lq == L - 1 && 2L * lv.H[lq] * lv.W[lq] > S && (long)S * ldv * 4 < 0x7fffffffL;
        if (win_ok) {
            // the largest margin R <= winR whose windows fit the two staging regions (640 pixels = 80 KB: two workgroups per CU)
            int R = winR;
            for (; R >= 1; --R) {
                wg.lq = lq; wg.R = R; wg.nx = 0;
                for (int l = 0; l < 4; ++l) { wg.ww[l] = wg.wh[l] = 1; wg.base[l] = 0; wg.sx[l] = wg.sy[l] = 1.f; }
                for (int l = 0; l < L; ++l) {
                    wg.sx[l] = (float)lv.W[l] / (float)lv.W[lq];
                    wg.sy[l] = (float)lv.H[l] / (float)lv.H[lq];
                    wg.ww[l] = (int)ceilf(7.f * wg.sx[l]) + 2 * R + 2;      // floor((x0 + 0.5) s - 0.5) .. floor((x0 + 7.5) s - 0.5), +-R, + the right tap
                    wg.wh[l] = (int)ceilf(7.f * wg.sy[l]) + 2 * R + 2;
                    if (l != lq) { wg.base[l] = wg.nx; wg.nx += wg.ww[l] * wg.wh[l]; }
                }
                wg.basey = (wg.nx + 7) & ~7;
                wg.base[lq] = wg.basey;
                wg.ny = wg.ww[lq] * wg.wh[lq];
                if (wg.nx <= 384 && wg.ny <= 320 && wg.basey + ((wg.ny + 7) & ~7) <= 640) break;
            }
            win_ok = R >= 1;
        }
        if (win_ok) {
            static S2dDevOnce attr;
            if (!attr.done()) {
                if (hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_fused_win_kernel<3, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 640 * 128) != hipSuccess)
                    return S2D_ERR_LAUNCH;
                attr.mark();
            }
            const int npatch = ((lv.W[lq] + 7) / 8) * ((lv.H[lq] + 7) / 8);
            hipLaunchKernelGGL((msda_fused_win_kernel<3, 4>), dim3(npatch, N), dim3(512), (size_t)(wg.basey + ((wg.ny + 7) & ~7)) * 128, stream, value, ldv, lv, wg, offs_logits,
                               ldoa, S, M, npatch, out);
            S2D_CHECK_LAUNCH();
            int ntile = 0;
            for (int l = 0; l < L; ++l) if (l != lq) ntile += ((lv.W[l] + 3) / 4) * ((lv.H[l] + 3) / 4);
            if (ntile > 0)
                hipLaunchKernelGGL((msda_fused_kernel<12, false, true, true>), dim3(ntile, N), dim3(1024), 0, stream, value, ldv, lv, offs_logits, ldoa, S, M, L, P,
                                   ntile, out, lq);
        } else if (tiled && M == 8 && head_ok) {
            static S2dDevOnce attr;
            if (!attr.done()) {
                if (hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_fused_head_kernel<12>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
                    return S2D_ERR_LAUNCH;
                attr.mark();
            }
            int npatch = 0;
            for (int l = 0; l < L; ++l) npatch += ((lv.W[l] + 15) / 16) * ((lv.H[l] + 7) / 8);
            // ~4 workgroups per CU in total (one resident per CU: the plane takes most of its LDS)
            const int chunks = max(1, min(npatch, (int)cdiv(1024L, (long)N * M)));
            hipLaunchKernelGGL((msda_fused_head_kernel<12>), dim3(chunks * M, N), dim3(1024), (size_t)lv.H[ls] * lv.W[ls] * 128, stream, value, ldv, lv, offs_logits, ldoa,
                               S, M, L, P, ls, npatch, chunks, out);
        } else if (tiled && M == 8) {
            int ntile = 0;
            for (int l = 0; l < L; ++l) ntile += ((lv.W[l] + 3) / 4) * ((lv.H[l] + 3) / 4);
            // record form (default; S2D_MSDA_REC=0: the TILED form, 2: the record form at 4 waves per SIMD / 128 registers), read per call
            int recmode = 1;
            if (const char *e = getenv("S2D_MSDA_REC")) recmode = atoi(e);
            const long fb = ((long)(S - 1) * ldv + (long)M * D) * 4;           // bytes of one frame's value slice as the kernel addresses it
            if (recmode && L == 3 && P == 4 && fb < 0x7fffffffL && (long)lv.start[L - 1] * ldv * 4 < 0x7fffffffL) {
                static S2dDevOnce attr;
                if (!attr.done()) {
                    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_fused_rec_kernel<3, 4, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 12 * REC_ROW) != hipSuccess ||
                        hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_fused_rec_kernel<3, 4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 12 * REC_ROW) != hipSuccess)
                        return S2D_ERR_LAUNCH;
                    attr.mark();
                }
                if (recmode == 2)
                    hipLaunchKernelGGL((msda_fused_rec_kernel<3, 4, 4>), dim3(ntile, N), dim3(1024), 16 * 12 * REC_ROW, stream, value, ldv, lv, offs_logits, ldoa, S, ntile, out, (unsigned int)fb);
                else
                    hipLaunchKernelGGL((msda_fused_rec_kernel<3, 4, 8>), dim3(ntile, N), dim3(1024), 16 * 12 * REC_ROW, stream, value, ldv, lv, offs_logits, ldoa, S, ntile, out, (unsigned int)fb);
            } else
            hipLaunchKernelGGL((msda_fused_kernel<12, false, true, true>), dim3(ntile, N), dim3(1024), 0, stream, value, ldv, lv, offs_logits, ldoa, S, M, L, P, ntile, out);
        } else if (share) hipLaunchKernelGGL((msda_fused_kernel<12, false, true>), dim3(nb, N), dim3(256), 0, stream, value, ldv, lv, offs_logits, ldoa, S, M, L, P, nb, out);
        else hipLaunchKernelGGL((msda_fused_kernel<12, false, false>), dim3(nb, N), dim3(256), 0, stream, value, ldv, lv, offs_logits, ldoa, S, M, L, P, nb, out);
    }
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

/* The query-owned half of the fused backward in one launch: d(loss)/d(offs_logits) [N][S][ldd] (2 * M * L * P offset gradients, then
 * M * L * P logit gradients per row, the layout of offs_logits) from grad_out [N][S][M][32], the value tensor and the raw projection
 * rows; what s2d_msda_backward_sorted*_f32's grad_loc / grad_attn followed by s2d_msda_fused_chain_f32 give, without those two
 * tensors.  S2D geometry only (M = 8, D = 32, L = 3, P = 4); other geometries: S2D_ERR_ARG, use the two-step form. */
int s2d_msda_fused_backward_query_f32(const float *value, int ldv, const int64_t *shapes_host, const float *offs_logits, int ldoa,
                                      const float *grad_out, int N, int S, int M, int D, int L, int P, float *d_offs_logits, int ldd,
                                      hipStream_t stream)
{
    Levels lv;
    if (int e = fill_levels(lv, shapes_host, nullptr, L, S)) return e;
    if (M != 8 || D != 32 || L != 3 || P != 4 || ldoa < M * L * P * 3 || (ldoa & 3) || ldv < M * D || (ldv & 3) || ldd < M * L * P * 3 || (ldd & 3) ||
        (reinterpret_cast<uintptr_t>(offs_logits) & 15) || (reinterpret_cast<uintptr_t>(d_offs_logits) & 7) || (reinterpret_cast<uintptr_t>(grad_out) & 15) ||
        (reinterpret_cast<uintptr_t>(value) & 15))
        return S2D_ERR_ARG;
    if (N <= 0) return S2D_OK;
    const long fb = ((long)(S - 1) * ldv + (long)M * D) * 4;
    if (fb >= 0x7fffffffL) return S2D_ERR_ARG;
    int ntile = 0;
    for (int l = 0; l < L; ++l) ntile += ((lv.W[l] + 3) / 4) * ((lv.H[l] + 3) / 4);
    static S2dDevOnce attr;
    if (!attr.done()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_fused_bwd_rec_kernel<3, 4, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 12 * REC_ROW) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_fused_bwd_rec_kernel<3, 4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 12 * REC_ROW) != hipSuccess)
            return S2D_ERR_LAUNCH;
        attr.mark();
    }
    // 4 waves per SIMD, four samples' loads in flight: 3.98 ms per encoder layer's whole backward at c4 against 4.56 with the loc + chain
    // pair.  The 8-wave form (S2D_MSDA_BWD_WAVES=8, read per call; the forward's choice) needs 6 spilled registers and 28 B of scratch
    // per lane at its 64-register budget and runs at 11.8 ms (profiles/r5_experiments/not_adopted.txt).
    int waves = 4;
    if (const char *e = getenv("S2D_MSDA_BWD_WAVES")) waves = atoi(e);
    if (waves != 8)
        hipLaunchKernelGGL((msda_fused_bwd_rec_kernel<3, 4, 4>), dim3(ntile, N), dim3(1024), 16 * 12 * REC_ROW, stream, value, ldv, lv, offs_logits, ldoa, grad_out, S, ntile,
                           d_offs_logits, ldd, (unsigned int)fb);
    else
        hipLaunchKernelGGL((msda_fused_bwd_rec_kernel<3, 4, 8>), dim3(ntile, N), dim3(1024), 16 * 12 * REC_ROW, stream, value, ldv, lv, offs_logits, ldoa, grad_out, S, ntile,
                           d_offs_logits, ldd, (unsigned int)fb);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

}  // extern "C"
