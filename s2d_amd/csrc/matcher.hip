// VideoHungarianMatcher on the device: fused point-sampling + cost contractions, and the rectangular LSAP.
//
// Replaces model_training/mask2former_video/modeling/matcher.py:225-294 for ALL prediction layers and clips
// of one criterion pass in two launches + one LSAP launch, with no device->host copy (the reference does
// C.cpu() + scipy per layer per clip, matcher.py:287-289).
//
// Cost algebra (exact rewrites of matcher.py:15-30, 38-62; x = sampled logit, t = sampled target in [0,1]):
//   cost_mask[q,n] = ( sum_p softplus(-x)*t + softplus(x)*(1-t) ) / TP = ( sum_p softplus(x_qp) - sum_p x_qp*t_np ) / TP
//   cost_dice[q,n] = 1 - (2*sum_p sigmoid(x_qp)*t_np + 1) / (sum_p sigmoid(x_qp) + sum_p t_np + 1)
// so one pass over the T*P sample points accumulates two [Q x N] contractions (x.t^T and sigmoid(x).t^T, on the
// fp32 MFMA) and three vectors.  Partials per sample chunk are written to a workspace and reduced in a fixed
// order in double, so the cost matrix -- and therefore the assignment -- is run-to-run deterministic.
//
// Mask logits are read PIXEL-MAJOR ([T*hm*wm][ldq], the row-major output of the mask-logit GEMM): the four
// bilinear corners of a sample point are four contiguous Q-float rows, one coalesced 128-B segment per wave.
#include "common.h"
#include <type_traits>

namespace {

constexpr int SB = 32;   // samples per batch (the MFMA K extent)
constexpr int QP = 128;  // padded queries
constexpr int NP = 128;  // padded targets

__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// two uniforms in [0,1) with 24 random bits each (torch.rand's float construction), keyed by (seed, stream, i)
__device__ __forceinline__ void rand2(uint64_t seed, uint64_t stream, uint64_t i, float &u, float &v)
{
    const uint64_t r = mix64(seed ^ mix64(stream * 0xD1342543DE82EF95ull + i));
    u = (float)(uint32_t)(r & 0xFFFFFFu) * (1.0f / 16777216.0f);
    v = (float)(uint32_t)((r >> 32) & 0xFFFFFFu) * (1.0f / 16777216.0f);
}

struct Bil {
    int i00, i01, i10, i11;   // pixel indices (y*W+x), -1 = outside (zero padding)
    float w00, w01, w10, w11;
};
// F.grid_sample(bilinear, zeros, align_corners=False) at normalised (u,v) in [0,1]  (point_features.py:19-42)
__device__ __forceinline__ Bil bil_setup(float u, float v, int H, int W)
{
    const float gx = 2.f * u - 1.f, gy = 2.f * v - 1.f;
    const float x = ((gx + 1.f) * W - 1.f) * 0.5f, y = ((gy + 1.f) * H - 1.f) * 0.5f;
    const int x0 = (int)floorf(x), y0 = (int)floorf(y), x1 = x0 + 1, y1 = y0 + 1;
    const float fx = x - x0, fy = y - y0;
    Bil b;
    const bool xa = x0 >= 0 && x0 < W, xb = x1 >= 0 && x1 < W, ya = y0 >= 0 && y0 < H, yb = y1 >= 0 && y1 < H;
    b.i00 = (ya && xa) ? y0 * W + x0 : -1;
    b.i01 = (ya && xb) ? y0 * W + x1 : -1;
    b.i10 = (yb && xa) ? y1 * W + x0 : -1;
    b.i11 = (yb && xb) ? y1 * W + x1 : -1;
    b.w00 = (1.f - fx) * (1.f - fy); b.w01 = fx * (1.f - fy); b.w10 = (1.f - fx) * fy; b.w11 = fx * fy;
    return b;
}

struct CostParams {
    const float *ml;         // [NL][B][T*hm*wm][ldq]
    const unsigned int *tbits;   // [B][T][H][W]: bit n = target n of the clip is set at that pixel (n < min(count, 32)); the f16 kernels' target side
    const uint8_t *tgt;      // [B][Nmax][T][H][W]
    const int *tgt_count;    // [B]
    const float *coords;     // [NL][B][P][2]  (band-sorted copy made by sort_points)
    int NL, B, Q, ldq, T, hm, wm, H, W, Nmax, P, chunks;   // chunks = T * CHM
    float *wsA, *wsD;        // [prob][chunk][QP][NP]
    float *wsV;              // [prob][chunk][3][128] : softplus sums[q], sigmoid sums[q], target sums[n]
};

constexpr int CHM = 16;      // blocks per (problem, frame) pair; they walk the sorted points together (interleaved batches)

// ---- points: generation (RNG mode) and a stable sort by logit-map cell -----------------------------------------
// The P points of a problem are i.i.d. uniform; every cost term is a SUM over points, so the order is free.
// They are sorted by the cell of the mask-logit map their upper-left tap falls in (stable LSD radix sort of
// (problem * cells + cell, index) pairs, sort.hip: deterministic).  The CHM blocks of one (problem, frame) pair run on one
// XCD and walk this list together, so the few logit rows and target rows they are sampling at any moment are fetched from
// HBM once and shared through that XCD's L2 (measured before any sorting: 12 % L2 hit rate, 47 GB fetched per launch),
// and each 32-point batch touches one short run of pixels that the f16 kernel stages in LDS.
__global__ void gen_points_kernel(float *__restrict__ out, uint64_t seed, int P)
{
    const int prob = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    float u, v;
    rand2(seed, (uint64_t)prob, (uint64_t)i, u, v);
    out[((long)prob * P + i) * 2] = u;
    out[((long)prob * P + i) * 2 + 1] = v;
}

// sort key of a point: the logit-map cell of its upper-left bilinear tap (clamped into the map), problem-major
__global__ __launch_bounds__(256) void cell_key_kernel(const float *__restrict__ coords, int P, int hm, int wm, unsigned int *__restrict__ keys,
                                                       unsigned int *__restrict__ vals)
{
    const int prob = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const long g = (long)prob * P + i;
    const float u = coords[2 * g], v = coords[2 * g + 1];
    const float fx = ((2.f * u - 1.f + 1.f) * wm - 1.f) * 0.5f, fy = ((2.f * v - 1.f + 1.f) * hm - 1.f) * 0.5f;   // as bil_setup
    const int cx = min(max((int)floorf(fx), 0), wm - 1), cy = min(max((int)floorf(fy), 0), hm - 1);
    keys[g] = (unsigned int)prob * (unsigned int)(hm * wm) + (unsigned int)(cy * wm + cx);
    vals[g] = (unsigned int)g;
}

__global__ __launch_bounds__(256) void gather_points_kernel(const float *__restrict__ coords, const unsigned int *__restrict__ vals, long n,
                                                            float *__restrict__ out)
{
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const unsigned int src = vals[j];
    out[2 * j] = coords[2L * src];
    out[2 * j + 1] = coords[2L * src + 1];
}

template <int NT>
__global__ __launch_bounds__(256) void matcher_cost_kernel(CostParams p)
{
    constexpr int TN = 32 * NT, SLOTS = 256 / TN, SPT = SB / SLOTS;
    __shared__ float Ts[SB][TN];
    __shared__ int bqi[4][SB], bti[4][SB];
    __shared__ float bqw[4][SB], btw[4][SB];
    __shared__ float tpart[SLOTS][TN];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l32 = lane & 31, h = lane >> 5;
    // XCD-affine decode: blocks with equal blockIdx % 8 share an XCD (speed only).  The CHM blocks of a
    // (problem, frame) pair get consecutive slots on ONE XCD, i.e. they are dispatched together and share its L2.
    const int xcd = blockIdx.x & 7, bslot = blockIdx.x >> 3;
    const int pair = (bslot / CHM) * 8 + xcd, c = bslot % CHM;
    if (pair >= p.NL * p.B * p.T) return;
    const int prob = pair / p.T, t = pair % p.T;
    const int b = prob % p.B;
    const int N = min(p.tgt_count[b], p.Nmax);
    if (N == 0 || (NT == 1 ? N > 32 : N <= 32)) return;      // the other instantiation owns this problem
    const int ntl = (N + 31) / 32;
    const int q = wv * 32 + l32;
    const bool qok = q < p.Q;
    const float *ml = p.ml + ((long)prob * p.T + t) * p.hm * p.wm * p.ldq + (qok ? q : 0);
    const uint8_t *tg = p.tgt + ((long)b * p.Nmax * p.T + t) * p.H * p.W;
    const long tplane = (long)p.T * p.H * p.W;
    const float *cr = p.coords + (long)prob * p.P * 2;
    const int tn = tid % TN, slot = tid / TN;

    f32x16 aA[NT], aD[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) { aA[k][r] = 0.f; aD[k][r] = 0.f; }
    float spsum = 0.f, sgsum = 0.f, tsum = 0.f;

    for (int base = c * SB; base < p.P; base += CHM * SB) {     // batches c, c+CHM, ... of the row-sorted points
        const int nvalid = min(SB, p.P - base);
        __syncthreads();  // previous batch fully consumed
        if (tid < SB) {
            float u = 0.f, v = 0.f;
            if (tid < nvalid) { u = cr[2 * (base + tid)]; v = cr[2 * (base + tid) + 1]; }
            const Bil a = bil_setup(u, v, p.hm, p.wm), d = bil_setup(u, v, p.H, p.W);
            // out-of-image corners (zero padding): clamp the offset to a valid element and zero the weight, so the
            // gathers below are unconditional (no exec-mask branches around 80 loads per batch)
            const bool tail = tid >= nvalid;
            bqi[0][tid] = a.i00 < 0 ? 0 : a.i00 * p.ldq; bqi[1][tid] = a.i01 < 0 ? 0 : a.i01 * p.ldq;
            bqi[2][tid] = a.i10 < 0 ? 0 : a.i10 * p.ldq; bqi[3][tid] = a.i11 < 0 ? 0 : a.i11 * p.ldq;
            bqw[0][tid] = (a.i00 < 0 || tail) ? 0.f : a.w00; bqw[1][tid] = (a.i01 < 0 || tail) ? 0.f : a.w01;
            bqw[2][tid] = (a.i10 < 0 || tail) ? 0.f : a.w10; bqw[3][tid] = (a.i11 < 0 || tail) ? 0.f : a.w11;
            bti[0][tid] = max(d.i00, 0); bti[1][tid] = max(d.i01, 0); bti[2][tid] = max(d.i10, 0); bti[3][tid] = max(d.i11, 0);
            btw[0][tid] = (d.i00 < 0 || tail) ? 0.f : d.w00; btw[1][tid] = (d.i01 < 0 || tail) ? 0.f : d.w01;
            btw[2][tid] = (d.i10 < 0 || tail) ? 0.f : d.w10; btw[3][tid] = (d.i11 < 0 || tail) ? 0.f : d.w11;
        }
        __syncthreads();
        // target tile Ts[k][n]: thread (tn, slot) samples target tn at points slot, slot+SLOTS, ...
        {
            const uint8_t *pl = tg + (long)(tn < N ? tn : 0) * tplane;
            const float live = tn < N ? 1.f : 0.f;
#pragma unroll
            for (int j = 0; j < SPT; ++j) {
                const int k = slot + SLOTS * j;
                float val = 0.f;
#pragma unroll
                for (int cnr = 0; cnr < 4; ++cnr) val += (float)pl[bti[cnr][k]] * btw[cnr][k];
                val *= live;
                Ts[k][tn] = val;
                tsum += val;
            }
        }
        // query side: lane (q, h) samples its query at points 2s+h
        float xs[16], sg[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int k = 2 * s + h;
            float x = 0.f;
#pragma unroll
            for (int cnr = 0; cnr < 4; ++cnr) x += ml[bqi[cnr][k]] * bqw[cnr][k];
            const float liveq = (qok && k < nvalid) ? 1.f : 0.f;
            const float e = __expf(-fabsf(x));
            const float inv = __builtin_amdgcn_rcpf(1.f + e);
            const float sgm = (x >= 0.f ? inv : e * inv) * liveq;          // sigmoid(x)
            spsum += (fmaxf(x, 0.f) + __logf(1.f + e)) * liveq;            // softplus(x) = BCE-with-logits vs 0 (matcher.py:54-56)
            sgsum += sgm;
            xs[s] = x * liveq; sg[s] = sgm;
        }
        __syncthreads();  // Ts complete
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (nt < ntl) {
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const float tb = Ts[2 * s + h][nt * 32 + l32];
                    aA[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[s], tb, aA[nt], 0, 0, 0);
                    aD[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(sg[s], tb, aD[nt], 0, 0, 0);
                }
            }
        }
    }
    // partials: acc[nt][r] = M[q = wv*32 + (r&3)+8(r>>2)+4h][n = nt*32 + l32]
    const long pc = (long)prob * p.chunks + (long)t * CHM + c;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        if (nt < ntl) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qq = wv * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                p.wsA[(pc * QP + qq) * NP + nt * 32 + l32] = aA[nt][r];
                p.wsD[(pc * QP + qq) * NP + nt * 32 + l32] = aD[nt][r];
            }
        }
    }
    spsum += __shfl_xor(spsum, 32, 64);
    sgsum += __shfl_xor(sgsum, 32, 64);
    if (h == 0) {
        p.wsV[(pc * 3 + 0) * 128 + q] = spsum;
        p.wsV[(pc * 3 + 1) * 128 + q] = sgsum;
    }
    tpart[slot][tn] = tsum;
    __syncthreads();
    if (tid < TN) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) s += tpart[k][tid];   // fixed order: deterministic
        p.wsV[(pc * 3 + 2) * 128 + tid] = s;
    }
}

// The target side of the f16 kernels reads ONE word per (point, tap) for all <= 32 targets of the clip: the binary target planes
// [B][Nmax][T][H][W] (bytes, nonzero = set; the reference's bool gt_masks, matcher.py:246) are interleaved once per call into
// [B][T][H][W] words.  A 32-point batch then costs 128 dword loads instead of 4 096 byte gathers over up to 32 planes -- measured
// 1.36 ms of the 5.9 ms call at c4 (profiles/r4_experiments/matcher_dbg.txt).  Four pixels per thread.
__global__ __launch_bounds__(256) void target_bits_kernel(const uint8_t *__restrict__ tgt, const int *__restrict__ tgt_count, int Nmax, int T,
                                                           long HW, unsigned int *__restrict__ out)
{
    const int bt = blockIdx.y, b = bt / T, t = bt % T;
    const long i4 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= HW) return;
    const int N = min(min(tgt_count[b], Nmax), 32);
    const uint8_t *src = tgt + ((long)b * Nmax * T + t) * HW + i4;
    unsigned int w[4] = {0u, 0u, 0u, 0u};
    const bool full = i4 + 4 <= HW && (HW & 3) == 0;
    for (int n = 0; n < N; ++n) {
        const uint8_t *pl = src + (long)n * T * HW;
        if (full) {
            const unsigned int v = *reinterpret_cast<const unsigned int *>(pl);
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] |= (((v >> (8 * e)) & 0xFFu) ? 1u : 0u) << n;
        } else {
            for (int e = 0; e < 4 && i4 + e < HW; ++e) w[e] |= (pl[e] ? 1u : 0u) << n;
        }
    }
    unsigned int *dst = out + (long)bt * HW + i4;
    for (int e = 0; e < 4 && i4 + e < HW; ++e) dst[e] = w[e];
}

// N <= 32 targets: the two [Q x N] contractions on the f16 matrix cores with the split-fp16 x3 scheme of
// gemm_bf16.hip (x = h + l*2^-11, main and cross accumulators): 12 MFMAs of 32 cycles per 32-sample batch instead of
// 32 fp32-input MFMAs of 64 cycles, at fp32-class accuracy (~3*2^-22 relative).  |logit| < 65504 is required.
typedef __fp16 h16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split_pair(float a, float b, unsigned int &hi, unsigned int &lo)
{
    const h16x2 h = __builtin_amdgcn_cvt_pkrtz(a, b);
    // (a - h) * 2048 == fma(h, -2048, a * 2048) exactly (a - h is exact, the factor a power of two): one multiply + one
    // v_fma_mix_f32 (fp16 operand converted inside the fma) per value instead of convert, subtract, multiply
    const h16x2 l = __builtin_amdgcn_cvt_pkrtz(__builtin_fmaf((float)h[0], -2048.f, a * 2048.f), __builtin_fmaf((float)h[1], -2048.f, b * 2048.f));
    hi = __builtin_bit_cast(unsigned int, h);
    lo = __builtin_bit_cast(unsigned int, l);
}

// timing experiments only (scripts/build_matcher_dbg.sh): 1 no target gathers, 2 no staged-row DMA, 4 no tap-table setup, 8 no query sampling.
// Results of such builds are wrong by construction.
#ifndef S2D_MATCHER_DBG
#define S2D_MATCHER_DBG 0
#endif
constexpr int SPANMAX = 24;     // staged logit rows per tap row (upper / lower): a 32-point batch spans ~12 cells at S2D density
constexpr int ROWS = 2 * SPANMAX + 3;   // LDS rows of one staged block: slack, upper run, slack, lower run, slack
// Q16: 16-query row tiles (v_mfma_f32_16x16x32_f16), one per wave, ceil(Q / 16) waves (at least the four the target side needs):
// Q = 100 pads to 112 rows instead of 128 -- the sigmoid / softplus work per (query, point) is what bounds the kernel -- and a
// workgroup brings 7 waves instead of 4 to hide its LDS latency.  Lane (query l & 15, point group l >> 4) evaluates points
// 8 (l >> 4) .. + 7 of a 32-point batch: the A fragment of one MFMA over the whole batch; targets in one or two 16-column tiles.
// MODE 2 (96 < Q <= 112, the shipped Q = 100): waves 0..2 take 32-query tiles, wave 3 the queries 96..111 as ONE 16-row tile -- half
// the per-lane sampling work of a 32-row tile whose rows 100..127 are padding -- and, having time to spare, also both tap-table setups
// of the batch two ahead (lanes 0..31 the logit-map side, lanes 32..63 the target side), which in the other modes make waves 0 and 1
// the ones every barrier waits for.  Per batch and lane: waves 0..2 16 samples + the target tile, wave 3 8 samples + target tile +
// setups.  Same sums in the same order per (query, target): a query's row of the contraction does not depend on its tile shape.
template <int MODE>
__device__ __forceinline__ void matcher_cost_f16_body(const CostParams &p)
{
    constexpr bool Q16 = MODE == 1;
    constexpr int TN = 32, SLOTS = 8, SPT = SB / SLOTS;     // 4 samples per thread on the target side
    constexpr int TROW = 20;                               // words per target row: 16 data (32 fp16) + 4 pad
    // The points arrive sorted by the logit-map cell they fall in (cell_key_kernel + radix sort), so the 32 points of a
    // batch touch a short run of consecutive pixels [cmin, cmax+1] in the row of their upper taps and the same run one
    // row below.  Those ~2 x 14 pixel rows (Q logits each) are copied into LDS once per batch -- coalesced 16-B loads
    // issued one batch ahead -- and the 64 taps a lane needs are LDS reads: a batch costs ~28 row fetches from L2 instead of
    // 128, and no lane waits on a gather.  A batch whose run is longer than SPANMAX (sparse or injected points) takes the
    // direct-gather path.  One barrier per batch: tap tables in a ring of 3 (batch i in use, i+1 target gathers and row
    // loads in flight, i+2 being set up), staged rows and target tile in rings of 2.
    // rowbuf [2][ROWS][128], ROWS = 2 * SPANMAX + 3: [slack][upper run: span rows][slack][lower run: span rows][slack].  The two
    // x-adjacent taps of a tap row are LDS rows r and r + 1 (512 B apart): one ds_read2_b32 per tap row from one address.  A tap
    // outside the map has weight 0 and reads whatever finite number lies there: the slack rows (x0 = -1 reads the row before a
    // run, x1 = wm the row after; y0 = -1 / y1 = hm read rows 0 / 1) -- the buffer is zeroed once and only ever receives logits.
    extern __shared__ __attribute__((aligned(16))) float rowbuf[];
    __shared__ __attribute__((aligned(16))) unsigned int Th[2][TN][TROW], Tl[2][TN][TROW];
    __shared__ __attribute__((aligned(16))) int bqi[3][SB][4], bti[3][SB][4];      // per sample: 4 tap offsets / weights,
    __shared__ __attribute__((aligned(16))) float bqw[3][SB][4], btw[3][SB][4];   // read back as one 16-B LDS load each
    __shared__ int bmeta[3][4];                                                  // cmin, span, staged
    __shared__ __attribute__((aligned(16))) unsigned int twl[2][SB][4];          // per sample: the 4 taps' target words (bit n = target n)
    __shared__ float tpart[SLOTS][TN];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l32 = lane & 31, h = lane >> 5;
    const int l16 = lane & 15, g16 = lane >> 4;
    const int xcd = blockIdx.x & 7, bslot = blockIdx.x >> 3;
    const int pair = (bslot / CHM) * 8 + xcd, c = bslot % CHM;
    if (pair >= p.NL * p.B * p.T) return;
    const int prob = pair / p.T, t = pair % p.T;
    const int b = prob % p.B;
    const int N = min(p.tgt_count[b], p.Nmax);
    if (N == 0 || N > 32) return;                           // N > 32: matcher_cost_kernel<4>
    const bool w16 = Q16 || (MODE == 2 && wv == 3);        // this wave works on a 16-query tile (wave-uniform)
    const int q = Q16 ? wv * 16 + l16 : (w16 ? 96 + l16 : wv * 32 + l32);
    // Rows q >= Q and target columns >= N are computed on clamped (valid) data and never read by the finalize kernel:
    // a row of the contraction depends on its own query only, a column on its own target only.  All global accesses are
    // buffer loads with 32-bit byte offsets.
    const long mapf = (long)p.hm * p.wm * p.ldq;
    const int npix = p.hm * p.wm;
    const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.ml + ((long)prob * p.T + t) * mapf), 0, (int)(mapf * 4), 0x00020000);
    const int qr = q < p.Q ? q : 0;
    const unsigned int q4 = (unsigned int)qr * 4u;
    const __amdgpu_buffer_rsrc_t rsT = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned int *>(p.tbits + ((long)b * p.T + t) * p.H * p.W), 0, (int)((long)p.H * p.W * 4), 0x00020000);
    const float *cr = p.coords + (long)prob * p.P * 2;
    const int tn = tid % TN, slot = tid / TN;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    // tap setup: lanes 0..31 of wave 0 do the logit-map side of sample l32, lanes 0..31 of wave 1 the target side
    constexpr int SETW = MODE == 2 ? 3 : 0;
    const bool setq = wv == SETW && lane < SB, sett = MODE == 2 ? (wv == 3 && lane >= 32) : (tid >= 64 && tid < 64 + SB);
    const int l4 = p.ldq >> 2;

    f32x16 aAm, aAx, aDm, aDx;
#pragma unroll
    for (int r = 0; r < 16; ++r) { aAm[r] = 0.f; aAx[r] = 0.f; aDm[r] = 0.f; aDx[r] = 0.f; }
    f32x4 bAm[2], bAx[2], bDm[2], bDx[2];                  // Q16: [target tile]
#pragma unroll
    for (int t = 0; t < 2; ++t) { bAm[t] = f32x4{0.f, 0.f, 0.f, 0.f}; bAx[t] = bAm[t]; bDm[t] = bAm[t]; bDx[t] = bAm[t]; }
    const bool two = N > 16;
    float relusum = 0.f, lg2sum = 0.f, sgsum = 0.f, tsum = 0.f;

    // out-of-image corners (zero padding): the offset is clamped to a valid element and the weight zeroed, so every
    // access is unconditional
    auto setup = [&](int buf, float u, float v, bool tail) {
        if (wv == SETW) {                                   // the setup wave: lanes 0..31 hold a sample each, 32..63 are neutral
            const Bil a = bil_setup(u, v, p.hm, p.wm);
            const float fx = ((2.f * u - 1.f + 1.f) * p.wm - 1.f) * 0.5f, fy = ((2.f * v - 1.f + 1.f) * p.hm - 1.f) * 0.5f;
            const int cx = min(max((int)floorf(fx), 0), p.wm - 1), cy = min(max((int)floorf(fy), 0), p.hm - 1);
            const bool mine = setq && !tail;
            int cmin = mine ? cy * p.wm + cx : 0x7fffffff, cmax = mine ? cy * p.wm + cx : -1;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) { cmin = min(cmin, __shfl_xor(cmin, o, 64)); cmax = max(cmax, __shfl_xor(cmax, o, 64)); }
            if (cmax < 0) { cmin = 0; cmax = 0; }            // all-tail batch
            const int span = (cmax - cmin + 3) & ~1;         // cells cmin .. cmax + 1, rounded up to whole row pairs of the DMA
            const bool staged = span <= SPANMAX;
            if (setq) {
                const int ii[4] = {a.i00, a.i01, a.i10, a.i11};
                const float ww[4] = {a.w00, a.w01, a.w10, a.w11};
                i32x4 o; f32x4 w;
                if (staged) {
                    // o[0] / o[1]: float offset of the LEFT tap of the tap row y0 / y0 + 1 in the staged block; its right tap is the
                    // next LDS row.  The block holds pixels [cmin, cmin + span) at LDS rows 1 .. span and [cmin + wm, cmin + wm + span)
                    // at rows span + 2 .. 2 span + 1; a left tap one pixel before a run (x0 = -1) is the slack row in front of it, a
                    // right tap one past it the slack row behind.  A tap row outside the map reads rows 0 / 1 with zero weights.
                    const int x0 = (int)floorf(fx), y0 = (int)floorf(fy);
                    auto locate = [&](int y, bool &ok) {
                        ok = y >= 0 && y < p.hm;
                        const int d = y * p.wm + x0 - cmin, d2 = d - p.wm;
                        if (ok && d >= -1 && d < span) return (1 + d) * 128;
                        if (ok && d2 >= -1 && d2 < span) return (span + 2 + d2) * 128;
                        ok = false;                          // cannot happen for a row inside the map; keep it harmless
                        return 0;
                    };
                    bool uok, lok;
                    o[0] = locate(y0, uok);
                    o[1] = locate(y0 + 1, lok);
                    o[2] = 0; o[3] = 0;
                    w[0] = (uok && ii[0] >= 0 && !tail) ? ww[0] : 0.f; w[1] = (uok && ii[1] >= 0 && !tail) ? ww[1] : 0.f;
                    w[2] = (lok && ii[2] >= 0 && !tail) ? ww[2] : 0.f; w[3] = (lok && ii[3] >= 0 && !tail) ? ww[3] : 0.f;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool ok = ii[e] >= 0 && !tail;
                        o[e] = ii[e] < 0 ? 0 : ii[e] * (p.ldq * 4);      // byte offset into the (problem, frame) logit map
                        w[e] = ok ? ww[e] : 0.f;
                    }
                }
                *reinterpret_cast<i32x4 *>(bqi[buf][l32]) = o;
                *reinterpret_cast<f32x4 *>(bqw[buf][l32]) = w;
                if (lane == 0) { bmeta[buf][0] = cmin; bmeta[buf][1] = span; bmeta[buf][2] = staged ? 1 : 0; }
            }
        }
        if (sett) {
            const Bil d = bil_setup(u, v, p.H, p.W);
            const i32x4 o = {max(d.i00, 0), max(d.i01, 0), max(d.i10, 0), max(d.i11, 0)};
            const f32x4 w = {(d.i00 < 0 || tail) ? 0.f : d.w00, (d.i01 < 0 || tail) ? 0.f : d.w01,
                             (d.i10 < 0 || tail) ? 0.f : d.w10, (d.i11 < 0 || tail) ? 0.f : d.w11};
            *reinterpret_cast<i32x4 *>(bti[buf][l32]) = o;
            *reinterpret_cast<f32x4 *>(btw[buf][l32]) = w;
        }
    };
    // staged logit rows of a batch: global -> LDS directly (buffer_load ... lds, 16 B per lane, no registers): one wave
    // instruction fills two 512-B LDS rows (lanes 0..31 / 32..63; ldq / 4 <= 32 float4s of each are real data)
    auto rows_dma = [&](int tb, int buf) {
        const int cmin = bmeta[tb][0], span = bmeta[tb][1];
        if (!bmeta[tb][2] || wv >= 4 || (S2D_MATCHER_DBG & 2)) return;
        // MODE 2: the row's last 16 floats (padding: Q <= 112) receive a second copy of queries 96..111 -- the 16-query wave's odd
        // point groups read that copy, so the two groups of a 32-lane LDS access fall on banks 0..15 and 16..31
        const int c4 = (MODE == 2 && (lane & 31) >= 28) ? (lane & 31) - 4 : (lane & 31);
#pragma unroll
        for (int r = 0; r < SPANMAX / 4; ++r) {
            const int rp = wv * (SPANMAX / 4) + r;           // row pair 0 .. SPANMAX-1
            if (2 * rp >= 2 * span) break;                   // wave-uniform
            const int row = 2 * rp + (lane >> 5);
            const int pix = cmin + row + (row >= span ? p.wm - span : 0);
            const bool ok = row < 2 * span && pix < npix && c4 < l4;
            // span is even, so a row pair lies in one run: LDS rows 1 + row (upper run) or 2 + row (lower run, behind its slack row)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rsM, (__attribute__((address_space(3))) void *)(rowbuf + ((long)buf * ROWS + 2 * rp + (2 * rp >= span ? 2 : 1)) * 128), 16,
                ok ? (pix * p.ldq + c4 * 4) * 4 : (int)0xFFFFFFF0u, 0, 0, 0);
        }
    };
    // target tile: thread (tn, slot) samples target tn at the 4 consecutive points 4*slot .. 4*slot+3 and stores them
    // as two fp16 pairs (hi / scaled lo) of row tn
    // target side: thread (sample tid >> 2, tap tid & 3) of the first two waves loads the word of its tap -- all targets of the clip at
    // once -- a batch ahead, parks it in LDS before the batch's barrier; thread (tn, slot) then interpolates target tn at the 4
    // consecutive points 4*slot .. 4*slot+3 from its bit of those words and stores them as two fp16 pairs (hi / scaled lo) of row tn
    unsigned int tword = 0u;
    auto target_gather = [&](int tb) {
        if (tid >= 4 * SB) return;
        if (S2D_MATCHER_DBG & 1) { tword = (unsigned int)tid * 0x9E3779B9u; return; }
        tword = __builtin_amdgcn_raw_buffer_load_b32(rsT, bti[tb][tid >> 2][tid & 3] * 4, 0, 0);
    };
    auto target_stash = [&](int buf) {
        if (tid < 4 * SB) twl[buf][tid >> 2][tid & 3] = tword;
    };
    auto target_tile = [&](int tb, int buf) {
        if (tid >= 256) return;
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        float val[SPT];
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            const f32x4 tw = *reinterpret_cast<const f32x4 *>(btw[tb][slot * SPT + j]);
            const u32x4 wd = *reinterpret_cast<const u32x4 *>(twl[buf][slot * SPT + j]);
            val[j] = fmaf((float)((wd[3] >> tn) & 1u), tw[3], fmaf((float)((wd[2] >> tn) & 1u), tw[2],
                          fmaf((float)((wd[1] >> tn) & 1u), tw[1], (float)((wd[0] >> tn) & 1u) * tw[0])));
            tsum += val[j];
        }
        unsigned int h0, l0, h1, l1;
        split_pair(val[0], val[1], h0, l0);
        split_pair(val[2], val[3], h1, l1);
        Th[buf][tn][slot * 2] = h0; Th[buf][tn][slot * 2 + 1] = h1;
        Tl[buf][tn][slot * 2] = l0; Tl[buf][tn][slot * 2 + 1] = l1;
    };
    auto load_uv = [&](int base, float &u, float &v, bool &tail) {
        u = 0.f; v = 0.f;
        tail = base + l32 >= p.P;
        if ((setq || sett) && !tail) { u = cr[2 * (base + l32)]; v = cr[2 * (base + l32) + 1]; }
    };

    for (int i = tid; i < 2 * ROWS * 128 / 4; i += blockDim.x) reinterpret_cast<f32x4 *>(rowbuf)[i] = f32x4{0.f, 0.f, 0.f, 0.f};   // finite slack rows
    __syncthreads();
    const int base0 = c * SB, bstep = CHM * SB;
    if (base0 < p.P) {
        float u, v; bool tail;
        load_uv(base0, u, v, tail);
        setup(0, u, v, tail);
        load_uv(base0 + bstep, u, v, tail);                // all-tail (zero weights, no load) past the end
        setup(1, u, v, tail);
        __syncthreads();
        target_gather(0);
        rows_dma(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        target_stash(0);
        __syncthreads();
        target_tile(0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                        // rows [0] visible
    int r0 = 0, cur = 0;                                    // r0 = i % 3, cur = i & 1
    for (int base = base0; base < p.P; base += bstep, cur ^= 1, r0 = (r0 == 2 ? 0 : r0 + 1)) {
        const int r1 = r0 == 2 ? 0 : r0 + 1, r2 = r1 == 2 ? 0 : r1 + 1;
        const int nvalid = min(SB, p.P - base);
        float un, vn; bool tailn;
        load_uv(base + 2 * bstep, un, vn, tailn);
        target_gather(r1);                                  // batch i+1 (all-tail tables past the end: offsets 0)
        rows_dma(r1, cur ^ 1);                              // batch i+1's logit rows -> rows [cur^1] (last read before the previous barrier)
        // query side: lane (q, h) samples its query at points 16*st + 8*h + j  (the lane's A-fragment k range)
        const bool staged_now = bmeta[r0][2] != 0;
        const float *rb = rowbuf + (long)cur * ROWS * 128 + ((MODE == 2 && w16) ? 96 + l16 + 16 * (g16 & 1) : qr);
        // softplus(x) = max(x,0) + ln2 * log2(1 + 2^(-|x| log2 e)): the two sums are kept apart and ln2 is applied once.
        // A tail sample (zero tap weights) has x == 0 exactly; only the one partial batch of a chunk pays for the masks.
        unsigned int xh[2][4], xl[2][4], gh[2][4], gl[2][4];
        auto query_half = [&](int st, auto masked, auto staged) {
            float dprod = 1.f;
            float m[8][4];                                   // direct-gather path: this half's 32 taps in flight together
            if constexpr (!decltype(staged)::value) {
#pragma unroll
                for (int s8 = 0; s8 < 8; ++s8) {
                    const i32x4 qi = *reinterpret_cast<const i32x4 *>(bqi[r0][(w16 ? 8 * g16 : 16 * st + 8 * h) + s8]);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        m[s8][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsM, (int)((unsigned int)qi[e] + q4), 0, 0));
                }
            }
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) {
                float xv[2], sv[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int s8 = 2 * jp + e, k = (w16 ? 8 * g16 : 16 * st + 8 * h) + s8;
                    const f32x4 qw = *reinterpret_cast<const f32x4 *>(bqw[r0][k]);
                    float m0, m1, m2, m3;
                    if constexpr (decltype(staged)::value) {
                        typedef int i32x2 __attribute__((ext_vector_type(2)));
                        const i32x2 qi = *reinterpret_cast<const i32x2 *>(bqi[r0][k]);     // left taps of the two tap rows
                        m0 = rb[qi[0]]; m1 = rb[qi[0] + 128]; m2 = rb[qi[1]]; m3 = rb[qi[1] + 128];    // one ds_read2_b32 per tap row
                    } else { m0 = m[s8][0]; m1 = m[s8][1]; m2 = m[s8][2]; m3 = m[s8][3]; }
                    const float x = fmaf(m3, qw[3], fmaf(m2, qw[2], fmaf(m1, qw[1], m0 * qw[0])));
                    const float ex = __builtin_amdgcn_exp2f(fabsf(x) * -1.44269504f);
                    const float den = 1.f + ex;
                    const float inv = __builtin_amdgcn_rcpf(den);
                    float sgm = x >= 0.f ? inv : ex * inv;
                    float dl = den;
                    if constexpr (decltype(masked)::value) {
                        const bool liveq = k < nvalid;
                        sgm = liveq ? sgm : 0.f; dl = liveq ? den : 1.f;
                    }
                    relusum += fmaxf(x, 0.f);
                    dprod *= dl;                               // sum of log2(1 + e) over the half's 8 samples = log2 of the product (<= 2^8)
                    sgsum += sgm;
                    xv[e] = x; sv[e] = sgm;
                }
                split_pair(xv[0], xv[1], xh[st][jp], xl[st][jp]);
                split_pair(sv[0], sv[1], gh[st][jp], gl[st][jp]);
            }
            lg2sum += __builtin_amdgcn_logf(dprod);
        };
#pragma unroll
        for (int st = 0; st < (Q16 ? 1 : 2); ++st) {
            if (st == 1 && w16) break;                       // wave-uniform
            if (S2D_MATCHER_DBG & 8) {
#pragma unroll
                for (int jp = 0; jp < 4; ++jp) { xh[st][jp] = 0x3c003c00u + lane; xl[st][jp] = lane; gh[st][jp] = 0x38003800u; gl[st][jp] = tid; }
                continue;
            }
            if (staged_now) {
                if (nvalid == SB) query_half(st, std::false_type{}, std::true_type{}); else query_half(st, std::true_type{}, std::true_type{});
            } else {
                if (nvalid == SB) query_half(st, std::false_type{}, std::false_type{}); else query_half(st, std::true_type{}, std::false_type{});
            }
        }
        if (!(S2D_MATCHER_DBG & 4)) setup(r2, un, vn, tailn);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's row DMAs (and target words) have landed
        target_stash(cur ^ 1);
        __syncthreads();  // target tile [cur] (written last iteration), rows [cur^1], target words [cur^1] and taps [r2] complete
        if (!w16) {
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const f16x8 th = *reinterpret_cast<const f16x8 *>(&Th[cur][l32][8 * st + 4 * h]);
            const f16x8 tl = *reinterpret_cast<const f16x8 *>(&Tl[cur][l32][8 * st + 4 * h]);
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 a0 = {xh[st][0], xh[st][1], xh[st][2], xh[st][3]}, a1 = {xl[st][0], xl[st][1], xl[st][2], xl[st][3]};
            const u32x4 g0 = {gh[st][0], gh[st][1], gh[st][2], gh[st][3]}, g1 = {gl[st][0], gl[st][1], gl[st][2], gl[st][3]};
            const f16x8 xhv = __builtin_bit_cast(f16x8, a0), xlv = __builtin_bit_cast(f16x8, a1);
            const f16x8 ghv = __builtin_bit_cast(f16x8, g0), glv = __builtin_bit_cast(f16x8, g1);
            aAx = __builtin_amdgcn_mfma_f32_32x32x16_f16(xlv, th, aAx, 0, 0, 0);
            aAx = __builtin_amdgcn_mfma_f32_32x32x16_f16(xhv, tl, aAx, 0, 0, 0);
            aAm = __builtin_amdgcn_mfma_f32_32x32x16_f16(xhv, th, aAm, 0, 0, 0);
            aDx = __builtin_amdgcn_mfma_f32_32x32x16_f16(glv, th, aDx, 0, 0, 0);
            aDx = __builtin_amdgcn_mfma_f32_32x32x16_f16(ghv, tl, aDx, 0, 0, 0);
            aDm = __builtin_amdgcn_mfma_f32_32x32x16_f16(ghv, th, aDm, 0, 0, 0);
        }
        } else {
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 a0 = {xh[0][0], xh[0][1], xh[0][2], xh[0][3]}, a1 = {xl[0][0], xl[0][1], xl[0][2], xl[0][3]};
            const u32x4 g0 = {gh[0][0], gh[0][1], gh[0][2], gh[0][3]}, g1 = {gl[0][0], gl[0][1], gl[0][2], gl[0][3]};
            const f16x8 xhv = __builtin_bit_cast(f16x8, a0), xlv = __builtin_bit_cast(f16x8, a1);
            const f16x8 ghv = __builtin_bit_cast(f16x8, g0), glv = __builtin_bit_cast(f16x8, g1);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (t == 1 && !two) break;                 // block-uniform: at most 16 targets
                const f16x8 th = *reinterpret_cast<const f16x8 *>(&Th[cur][16 * t + l16][4 * g16]);
                const f16x8 tl = *reinterpret_cast<const f16x8 *>(&Tl[cur][16 * t + l16][4 * g16]);
                bAx[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xlv, th, bAx[t], 0, 0, 0);
                bAx[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xhv, tl, bAx[t], 0, 0, 0);
                bAm[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xhv, th, bAm[t], 0, 0, 0);
                bDx[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(glv, th, bDx[t], 0, 0, 0);
                bDx[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ghv, tl, bDx[t], 0, 0, 0);
                bDm[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ghv, th, bDm[t], 0, 0, 0);
            }
        }
        target_tile(r1, cur ^ 1);
    }
    const long pc = (long)prob * p.chunks + (long)t * CHM + c;
    float spsum = relusum + 0.693147181f * lg2sum;
    if (!w16) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qq = wv * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            p.wsA[(pc * QP + qq) * NP + l32] = aAm[r] + aAx[r] * (1.0f / 2048.0f);
            p.wsD[(pc * QP + qq) * NP + l32] = aDm[r] + aDx[r] * (1.0f / 2048.0f);
        }
        spsum += __shfl_xor(spsum, 32, 64);
        sgsum += __shfl_xor(sgsum, 32, 64);
        if (h == 0) {
            p.wsV[(pc * 3 + 0) * 128 + q] = spsum;
            p.wsV[(pc * 3 + 1) * 128 + q] = sgsum;
        }
    } else {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (t == 1 && !two) break;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qq = (Q16 ? wv * 16 : 96) + 4 * g16 + r;
                p.wsA[(pc * QP + qq) * NP + 16 * t + l16] = bAm[t][r] + bAx[t][r] * (1.0f / 2048.0f);
                p.wsD[(pc * QP + qq) * NP + 16 * t + l16] = bDm[t][r] + bDx[t][r] * (1.0f / 2048.0f);
            }
        }
        spsum += __shfl_xor(spsum, 16, 64); spsum += __shfl_xor(spsum, 32, 64);
        sgsum += __shfl_xor(sgsum, 16, 64); sgsum += __shfl_xor(sgsum, 32, 64);
        if (g16 == 0) {
            p.wsV[(pc * 3 + 0) * 128 + q] = spsum;
            p.wsV[(pc * 3 + 1) * 128 + q] = sgsum;
        }
    }
    if (tid < 256) tpart[slot][tn] = tsum;
    __syncthreads();
    if (tid < TN) {
        float sacc = 0.f;
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) sacc += tpart[k][tid];
        p.wsV[(pc * 3 + 2) * 128 + tid] = sacc;
    }
}

__global__ __launch_bounds__(256, 2) void matcher_cost_f16_kernel(CostParams p) { matcher_cost_f16_body<0>(p); }
__global__ __launch_bounds__(256, 2) void matcher_cost_f16_mix_kernel(CostParams p) { matcher_cost_f16_body<2>(p); }
// two 7-wave workgroups per CU: 128 registers
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void matcher_cost_f16_q16_kernel(CostParams p) { matcher_cost_f16_body<1>(p); }

// C[prob][q][n] = w_mask*cost_mask + w_class*(-softmax(logits)[q][0]) + w_dice*cost_dice   (matcher.py:280-287)
__global__ void matcher_finalize_kernel(CostParams p, const float *__restrict__ cls, float wc, float wm_, float wd,
                                        float *__restrict__ C)
{
    const int prob = blockIdx.y;
    const int b = prob % p.B;
    const int N = min(p.tgt_count[b], p.Nmax);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.Q * p.Nmax) return;
    const int q = i / p.Nmax, n = i % p.Nmax;
    float out = 0.f;
    if (n < N) {
        double A = 0., D = 0., SP = 0., SG = 0., ST = 0.;
        for (int c = 0; c < p.chunks; ++c) {
            const long pc = (long)prob * p.chunks + c;
            A += (double)p.wsA[(pc * QP + q) * NP + n];
            D += (double)p.wsD[(pc * QP + q) * NP + n];
            SP += (double)p.wsV[(pc * 3 + 0) * 128 + q];
            SG += (double)p.wsV[(pc * 3 + 1) * 128 + q];
            ST += (double)p.wsV[(pc * 3 + 2) * 128 + n];
        }
        const double TP = (double)p.T * p.P;
        const double cost_mask = (SP - A) / TP;
        const double cost_dice = 1.0 - (2.0 * D + 1.0) / (SG + ST + 1.0);
        const float l0 = cls[((long)prob * p.Q + q) * 2], l1 = cls[((long)prob * p.Q + q) * 2 + 1];
        const float mx = fmaxf(l0, l1);
        const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
        const double prob0 = (double)(e0 / (e0 + e1));
        out = (float)((double)wm_ * cost_mask + (double)wc * (-prob0) + (double)wd * cost_dice);
    }
    C[((long)prob * p.Q + q) * p.Nmax + n] = out;
}

// ------------------------------------------------------------------------------------------------
// Rectangular linear sum assignment, one wavefront per problem.  Same algorithm and scan order as
// scipy.optimize.linear_sum_assignment (Crouse 2016 shortest augmenting paths; called at matcher.py:289), so
// ties resolve identically: rows <= cols (the [Q,N] cost is used transposed when N < Q), the column scan runs
// over the `remaining` list (filled in reverse, swap-removed), and among equal minima the LAST unassigned
// column in scan order wins, else the FIRST column.  All arithmetic in double on the fp32 costs, like scipy.
constexpr int LMAX = 128;
constexpr int COSTMAX = 12800;  // floats of LDS for the cost matrix: min(Q,N) * max(Q,N) <= 100 * 128
__global__ __launch_bounds__(64) void lsap_kernel(const float *__restrict__ Call, const int *__restrict__ tgt_count, int B,
                                                  int Q, int Nmax, int maxm, int *__restrict__ idx_q,
                                                  int *__restrict__ idx_t, int *__restrict__ n_match)
{
    __shared__ float cost[COSTMAX];   // [nr][nc]
    __shared__ double u[LMAX], v[LMAX], spc[LMAX];
    __shared__ int path[LMAX], row4col[LMAX], col4row[LMAX], remaining[LMAX];
    __shared__ unsigned char SR[LMAX], SC[LMAX];
    __shared__ int s_i, s_sink, s_nrem;
    __shared__ double s_min;
    const int prob = blockIdx.x, lane = threadIdx.x;
    const int b = prob % B;
    const int N = min(tgt_count[b], Nmax);
    const float *C = Call + (long)prob * Q * Nmax;
    const bool transpose = N < Q;
    const int nr = transpose ? N : Q, nc = transpose ? Q : N;
    if (nr == 0) { if (lane == 0) n_match[prob] = 0; return; }
    for (int e = lane; e < nr * nc; e += 64) {
        const int r = e / nc, c = e % nc;
        cost[e] = transpose ? C[(long)c * Nmax + r] : C[(long)r * Nmax + c];
    }
    for (int e = lane; e < LMAX; e += 64) { u[e] = 0.; v[e] = 0.; col4row[e] = -1; row4col[e] = -1; path[e] = -1; }
    __syncthreads();
    for (int cur = 0; cur < nr; ++cur) {
        for (int e = lane; e < nc; e += 64) { remaining[e] = nc - e - 1; spc[e] = INFINITY; SC[e] = 0; }
        for (int e = lane; e < nr; e += 64) SR[e] = 0;
        if (lane == 0) { s_i = cur; s_sink = -1; s_nrem = nc; s_min = 0.; }
        __syncthreads();
        while (true) {
            const int i = s_i, nrem = s_nrem;
            const double minVal = s_min;
            if (lane == 0) SR[i] = 1;
            // candidate of this lane over its strided share of `remaining`, in ascending `it`
            double best = INFINITY; int best_it_first = 0x7fffffff, best_it_unas = -1;
            const double ui = u[i];
            for (int it = lane; it < nrem; it += 64) {
                const int j = remaining[it];
                const double r = minVal + (double)cost[i * nc + j] - ui - v[j];
                if (r < spc[j]) { path[j] = i; spc[j] = r; }
                const double sj = spc[j];
                if (sj < best) { best = sj; best_it_first = it; best_it_unas = (row4col[j] == -1) ? it : -1; }
                else if (sj == best) { if (row4col[j] == -1) best_it_unas = it; }
            }
            // wave reduction: global min; among lanes at the min: max unassigned `it`, else min `it`
            double gmin = best;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) gmin = fmin(gmin, __shfl_xor(gmin, o, 64));
            int fi = (best == gmin) ? best_it_first : 0x7fffffff;
            int ua = (best == gmin) ? best_it_unas : -1;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { fi = min(fi, __shfl_xor(fi, o, 64)); ua = max(ua, __shfl_xor(ua, o, 64)); }
            __syncthreads();
            if (gmin == INFINITY) { if (lane == 0) n_match[prob] = -1; return; }  // infeasible (cannot happen: finite costs)
            const int index = ua >= 0 ? ua : fi;
            if (lane == 0) {
                const int j = remaining[index];
                s_min = gmin;
                if (row4col[j] == -1) s_sink = j; else s_i = row4col[j];
                SC[j] = 1;
                remaining[index] = remaining[nrem - 1];
                s_nrem = nrem - 1;
            }
            __syncthreads();
            if (s_sink != -1) break;
        }
        const double minVal = s_min;
        // dual updates
        for (int r = lane; r < nr; r += 64) {
            if (r == cur) u[r] += minVal;
            else if (SR[r]) u[r] += minVal - spc[col4row[r]];
        }
        for (int j = lane; j < nc; j += 64)
            if (SC[j]) v[j] -= minVal - spc[j];
        __syncthreads();
        if (lane == 0) {
            int j = s_sink;
            while (true) {
                const int r = path[j];
                row4col[j] = r;
                const int t = col4row[r];
                col4row[r] = j;
                j = t;
                if (r == cur) break;
            }
        }
        __syncthreads();
    }
    // emit in scipy's order (first index ascending)
    int *oq = idx_q + (long)prob * maxm, *ot = idx_t + (long)prob * maxm;
    if (!transpose) {
        for (int r = lane; r < nr; r += 64) { oq[r] = r; ot[r] = col4row[r]; }
    } else {
        int basec = 0;
        for (int j0 = 0; j0 < nc; j0 += 64) {
            const int j = j0 + lane;
            const bool has = j < nc && row4col[j] != -1;
            const unsigned long long m = __ballot(has);
            if (has) {
                const int pos = basec + __popcll(m & ((1ull << lane) - 1ull));
                oq[pos] = j; ot[pos] = row4col[j];
            }
            basec += __popcll(m);
        }
    }
    if (lane == 0) n_match[prob] = nr;
}

}  // namespace

extern "C" {



long s2d_matcher_workspace_floats(int NL, int B, int T, int P, int H, int W)
{
    const long nprob = (long)NL * B, ch = (long)T * CHM, n = nprob * P;
    // partial sums + raw / sorted coordinates + 4 key/value arrays + radix-sort scratch (checked against rocPRIM's need at launch)
    // + the interleaved target words [B][T][H][W]
    return nprob * ch * (2L * QP * NP + 3 * 128) + 4 * n + 4 * n + (4 * n + (1L << 20)) + 64 + (long)B * T * H * W;
}

int s2d_matcher_cost_f32(const float *mask_logits, const float *class_logits, const uint8_t *tgt, const int *tgt_count,
                         const float *coords, uint64_t seed, int NL, int B, int Q, int ldq, int T, int hm, int wm, int H,
                         int W, int Nmax, int P, float w_class, float w_mask, float w_dice, float *workspace, float *C,
                         hipStream_t stream)
{
    if (Q > QP || Nmax > NP || Q <= 0 || Nmax <= 0 || ldq < Q) return S2D_ERR_ARG;
    const int nprob = NL * B;
    if (nprob == 0) return S2D_OK;
    CostParams p;
    p.ml = mask_logits; p.tgt = tgt; p.tgt_count = tgt_count;
    p.NL = NL; p.B = B; p.Q = Q; p.ldq = ldq; p.T = T; p.hm = hm; p.wm = wm; p.H = H; p.W = W; p.Nmax = Nmax; p.P = P;
    p.chunks = T * CHM;
    p.wsA = workspace;
    p.wsD = p.wsA + (long)nprob * p.chunks * QP * NP;
    p.wsV = p.wsD + (long)nprob * p.chunks * QP * NP;
    const long n = (long)nprob * P;
    float *raw = p.wsV + (long)nprob * p.chunks * 3 * 128;
    float *sorted = raw + 2 * n;
    unsigned int *keys_in = (unsigned int *)(sorted + 2 * n), *keys_out = keys_in + n, *vals_in = keys_out + n, *vals_out = vals_in + n;
    void *tmp = vals_out + n;
    const size_t tmp_bytes = (size_t)(4 * n + (1L << 20)) * 4;
    unsigned int *tbits = reinterpret_cast<unsigned int *>(tmp) + (4 * n + (1L << 20)) + 64;
    if ((long)H * W * 4 > 0x7FFFFFFFL) return S2D_ERR_ARG;
    hipLaunchKernelGGL(target_bits_kernel, dim3(cdiv(cdiv((long)H * W, 4), 256), B * T), dim3(256), 0, stream, tgt, tgt_count, Nmax, T, (long)H * W, tbits);
    p.tbits = tbits;
    if (!coords) {
        hipLaunchKernelGGL(gen_points_kernel, dim3(cdiv(P, 256), nprob), dim3(256), 0, stream, raw, seed, P);
        coords = raw;
    }
    // Points are i.i.d. and every cost term is a SUM over points, so their order is free: a stable radix sort by logit-map
    // cell makes each 32-point batch of the cost kernel touch one short run of pixels (see matcher_cost_f16_kernel).
    const long nkeys = (long)nprob * hm * wm;
    if (nkeys >= (1L << 32) || n >= (1L << 31)) return S2D_ERR_ARG;
    int bits = 1;
    while ((1L << bits) < nkeys) ++bits;
    hipLaunchKernelGGL(cell_key_kernel, dim3(cdiv(P, 256), nprob), dim3(256), 0, stream, coords, P, hm, wm, keys_in, vals_in);
    if (int e = s2d_radix_sort_pairs_u32(keys_in, keys_out, vals_in, vals_out, (size_t)n, bits, tmp, tmp_bytes, stream)) return e;
    hipLaunchKernelGGL(gather_points_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, coords, vals_out, n, sorted);
    p.coords = sorted;
    const int npairs = nprob * T;
    const int grid = ((npairs + 7) / 8) * 8 * CHM;
    const size_t lds_rows = sizeof(float) * 2 * ROWS * 128;
    static S2dDevOnce attr_set;
    if (!attr_set.done()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(matcher_cost_f16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_rows) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(matcher_cost_f16_mix_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_rows) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(matcher_cost_f16_q16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_rows) != hipSuccess)
            return S2D_ERR_LAUNCH;
        attr_set.mark();
    }
    // opt-in (read per call): measured slower, profiles/r3_experiments/not_adopted.txt -- four point groups per wave read four staged
    // logit rows at once and every row starts at bank 0
    int q16 = 0;
    if (const char *e = getenv("S2D_MATCHER_Q16")) q16 = atoi(e);
    int mix = Q > 96 && Q <= 112;                           // three 32-query tiles + one 16-query tile (S2D_MATCHER_MIX=0: four 32-query tiles)
    if (const char *e = getenv("S2D_MATCHER_MIX")) mix = mix && atoi(e);
    if (q16) hipLaunchKernelGGL(matcher_cost_f16_q16_kernel, dim3(grid), dim3(64 * max(4, cdiv(Q, 16))), lds_rows, stream, p);
    else if (mix) hipLaunchKernelGGL(matcher_cost_f16_mix_kernel, dim3(grid), dim3(256), lds_rows, stream, p);
    else hipLaunchKernelGGL(matcher_cost_f16_kernel, dim3(grid), dim3(256), lds_rows, stream, p);
    if (Nmax > 32) hipLaunchKernelGGL(matcher_cost_kernel<4>, dim3(grid), dim3(256), 0, stream, p);
    hipLaunchKernelGGL(matcher_finalize_kernel, dim3(cdiv((long)Q * Nmax, 256), nprob), dim3(256), 0, stream, p,
                       class_logits, w_class, w_mask, w_dice, C);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

int s2d_lsap_f32(const float *C, const int *tgt_count, int nprob, int B, int Q, int Nmax, int *idx_q, int *idx_t,
                 int *n_match, hipStream_t stream)
{
    if (Q > LMAX || Nmax > LMAX || (long)Q * Nmax > COSTMAX) return S2D_ERR_ARG;
    if (nprob == 0) return S2D_OK;
    const int maxm = Q < Nmax ? Q : Nmax;
    hipLaunchKernelGGL(lsap_kernel, dim3(nprob), dim3(64), 0, stream, C, tgt_count, B, Q, Nmax, maxm, idx_q, idx_t, n_match);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

}  // extern "C"
