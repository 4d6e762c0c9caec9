// Bandwidth-bound glue kernels of the S2D forward (all HBM-bound, 16-B accesses, no MFMA):
// input normalise+pad, max-pool, GroupNorm (NHWC), LayerNorm(+residual), broadcast adds, sine position
// encodings, bilinear resize (+add).
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------------
// (x - mean) / std, zero-pad to (Hp, Wp), NCHW uint8 -> NHWC float with C padded 3 -> 4
// kd_video_maskformer_model.py:263-269 (+ detectron2 ImageList.from_tensors zero padding)
// (the element index is decomposed in the index type I: unsigned 32-bit whenever the tensor allows -- a 64-bit division
// chain per element costs more than the element's memory traffic)
template <typename I>
__global__ void normalize_pad_kernel(const uint8_t *__restrict__ in, int F, int H0, int W0, int Hp, int Wp, f32x4 mean,
                                     f32x4 stdv, float *__restrict__ out)
{
    const I i = (I)blockIdx.x * blockDim.x + threadIdx.x;
    const I total = (I)F * Hp * Wp;
    if (i >= total) return;
    const int x = (int)(i % (I)Wp);
    const I r = i / (I)Wp;
    const int y = (int)(r % (I)Hp);
    const int f = (int)(r / (I)Hp);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (y < H0 && x < W0) {
        const uint8_t *p = in + ((long)f * 3 * H0 + y) * W0 + x;
        const long cs = (long)H0 * W0;
        v[0] = ((float)p[0] - mean[0]) / stdv[0];
        v[1] = ((float)p[cs] - mean[1]) / stdv[1];
        v[2] = ((float)p[2 * cs] - mean[2]) / stdv[2];
    }
    *reinterpret_cast<f32x4 *>(out + (long)i * 4) = v;
}

// 3x3 / stride 2 / pad 1 max pool, NHWC, C % 4 == 0 (detectron2 BasicStem)
// IDX: also the arg-max tap (ky * 3 + kx, the FIRST maximum in scan order: where torch's max_pool2d backward routes the gradient) of every
// output element, one byte each -- the training step's backward then reads 4 bytes + one dy row per window instead of recomputing the
// arg-max of up to four windows from 36 input rows per input pixel (round 5)
template <typename I, bool IDX = false>
__global__ void maxpool_kernel(const float *__restrict__ in, int N, int H, int W, int C, int Ho, int Wo,
                               float *__restrict__ out, unsigned int *__restrict__ idx = nullptr)
{
    const int c4n = C / 4;
    const I i = (I)blockIdx.x * blockDim.x + threadIdx.x;
    const I total = (I)N * Ho * Wo * c4n;
    if (i >= total) return;
    const int c = (int)(i % (I)c4n);
    I t = i / (I)c4n;
    const int ox = (int)(t % (I)Wo); t /= (I)Wo;
    const int oy = (int)(t % (I)Ho);
    const int n = (int)(t / (I)Ho);
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    unsigned int am = 0u;
    bool seen = false;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int iy = oy * 2 - 1 + dy;
        if (iy < 0 || iy >= H) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int ix = ox * 2 - 1 + dx;
            if (ix < 0 || ix >= W) continue;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(in + (((long)n * H + iy) * W + ix) * C + c * 4);
            if constexpr (IDX) {
                const unsigned int tap = (unsigned int)(dy * 3 + dx);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (v[j] > m[j] || !seen) { m[j] = v[j]; am = (am & ~(0xFFu << (8 * j))) | (tap << (8 * j)); }
                seen = true;
            } else {
                m[0] = fmaxf(m[0], v[0]); m[1] = fmaxf(m[1], v[1]); m[2] = fmaxf(m[2], v[2]); m[3] = fmaxf(m[3], v[3]);
            }
        }
    }
    *reinterpret_cast<f32x4 *>(out + (long)i * 4) = m;
    if constexpr (IDX) idx[i] = am;
}

// ---------------------------------------------------------------------------------------------------
// GroupNorm over NHWC tokens x [N, HW, C]: per-block partial (sum, sumsq) in double, reduced in a fixed order by
// gn_reduce_kernel (no atomics: the statistics, and with them every downstream value, are bitwise reproducible).
// 256 threads: thread t owns channel quad c4 = t % (C/4) for rows r = t / (C/4) + k * (256/(C/4)).
__global__ __launch_bounds__(256) void gn_stats_kernel(const float *__restrict__ x, int HW, int C, int G, int rows_per_blk,
                                                       double *__restrict__ part)
{
    __shared__ double sh[2][256];   // [sum | sumsq][rslot * q + c4]
    const int n = blockIdx.y;
    const int q = C / 4, tpr = 256 / q;  // threads per row-slot
    const int c4 = threadIdx.x % q, rslot = threadIdx.x / q;
    const long r0 = (long)blockIdx.x * rows_per_blk;
    const long r1 = min((long)HW, r0 + rows_per_blk);
    double s = 0., ss = 0.;
    if (rslot < tpr) {
        const float *base = x + ((long)n * HW) * C + c4 * 4;
        long r = r0 + rslot;
        for (; r + 3L * tpr < r1; r += 4L * tpr) {           // four rows in flight per thread; added in row order, as the tail below
            f32x4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const f32x4 *>(base + (r + (long)k * tpr) * C);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s += (double)v[k][0] + (double)v[k][1] + (double)v[k][2] + (double)v[k][3];
                ss += (double)v[k][0] * v[k][0] + (double)v[k][1] * v[k][1] + (double)v[k][2] * v[k][2] + (double)v[k][3] * v[k][3];
            }
        }
        for (; r < r1; r += tpr) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(base + r * C);
            s += (double)v[0] + (double)v[1] + (double)v[2] + (double)v[3];
            ss += (double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2] + (double)v[3] * v[3];
        }
        sh[0][threadIdx.x] = s;
        sh[1][threadIdx.x] = ss;
    }
    __syncthreads();
    const int cpg4 = (C / G) / 4;  // float4s per group
    if (threadIdx.x < G) {
        double a = 0., b = 0.;
        for (int k = 0; k < cpg4; ++k)
            for (int rs = 0; rs < tpr; ++rs) { a += sh[0][rs * q + threadIdx.x * cpg4 + k]; b += sh[1][rs * q + threadIdx.x * cpg4 + k]; }
        double *o = part + (((long)n * gridDim.x + blockIdx.x) * G + threadIdx.x) * 2;
        o[0] = a; o[1] = b;
    }
}

// one wavefront per (n, g): lane l sums partials l, l + 64, ... in that order, then a fixed butterfly -- deterministic
__global__ __launch_bounds__(64) void gn_reduce_kernel(const double *__restrict__ part, int nblk, int G, double *__restrict__ stats)
{
    const int n = blockIdx.x / G, g = blockIdx.x % G;
    double a = 0., b = 0.;
    for (int k = threadIdx.x; k < nblk; k += 64) {
        a += part[(((long)n * nblk + k) * G + g) * 2];
        b += part[(((long)n * nblk + k) * G + g) * 2 + 1];
    }
    a = wave_sum_d(a); b = wave_sum_d(b);
    if (threadIdx.x == 0) {
        stats[((long)n * G + g) * 2] = a;
        stats[((long)n * G + g) * 2 + 1] = b;
    }
}

// ---- GroupNorm backward.  y = xhat * gamma + beta, xhat = (x - mean_g) * rstd_g per (sample, group):
//   g = dy * gamma,  dx = rstd * (g - mean_g(g) - xhat * mean_g(g * xhat)),  dgamma = sum dy * xhat,  dbeta = sum dy.
// gnb_sums_kernel has the thread layout of gn_stats_kernel and leaves, per (sample, row block), the group sums of g and
// g * xhat (double) and the per-channel sums of dy * xhat and dy (float, row slots added in a fixed order); both are finished
// by fixed-order reductions, so the gradients are bitwise reproducible.
__device__ __forceinline__ void gn_mean_rstd(const double *__restrict__ stats, int n, int g, int G, double cnt, float eps, float &mean, float &rstd)
{
    const double mu = stats[((long)n * G + g) * 2] / cnt;
    const double var = stats[((long)n * G + g) * 2 + 1] / cnt - mu * mu;
    rstd = (float)(1.0 / sqrt(var + (double)eps));
    mean = (float)mu;
}

__global__ __launch_bounds__(256) void gnb_sums_kernel(const float *__restrict__ x, const float *__restrict__ dy, const float *__restrict__ gamma,
                                                       const double *__restrict__ stats, int HW, int C, int G, float eps, int rows_per_blk,
                                                       double *__restrict__ part_g, float *__restrict__ part_c)
{
    __shared__ double sh[2][256];
    __shared__ float shc[2][256][4];
    const int n = blockIdx.y;
    const int q = C / 4, tpr = 256 / q;
    const int c4 = threadIdx.x % q, rslot = threadIdx.x / q;
    const long r0 = (long)blockIdx.x * rows_per_blk;
    const long r1 = min((long)HW, r0 + rows_per_blk);
    double s1 = 0., s2 = 0.;
    f32x4 dgam = f32x4(0.f), dbet = f32x4(0.f);
    if (rslot < tpr) {
        float mean, rstd;
        gn_mean_rstd(stats, n, (c4 * 4) / (C / G), G, (double)HW * (C / G), eps, mean, rstd);
        const f32x4 ga = *reinterpret_cast<const f32x4 *>(gamma + c4 * 4);
        const long base = ((long)n * HW) * C + c4 * 4;
        for (long r = r0 + rslot; r < r1; r += tpr) {
            const f32x4 xh = (*reinterpret_cast<const f32x4 *>(x + base + r * C) - mean) * rstd;
            const f32x4 d = *reinterpret_cast<const f32x4 *>(dy + base + r * C);
            const f32x4 g = d * ga;
            s1 += (double)g[0] + (double)g[1] + (double)g[2] + (double)g[3];
            s2 += (double)g[0] * xh[0] + (double)g[1] * xh[1] + (double)g[2] * xh[2] + (double)g[3] * xh[3];
            dgam += d * xh;
            dbet += d;
        }
        sh[0][threadIdx.x] = s1;
        sh[1][threadIdx.x] = s2;
#pragma unroll
        for (int j = 0; j < 4; ++j) { shc[0][threadIdx.x][j] = dgam[j]; shc[1][threadIdx.x][j] = dbet[j]; }
    }
    __syncthreads();
    const int cpg4 = (C / G) / 4;
    if (threadIdx.x < G) {
        double a = 0., b = 0.;
        for (int k = 0; k < cpg4; ++k)
            for (int rs = 0; rs < tpr; ++rs) { a += sh[0][rs * q + threadIdx.x * cpg4 + k]; b += sh[1][rs * q + threadIdx.x * cpg4 + k]; }
        double *o = part_g + (((long)n * gridDim.x + blockIdx.x) * G + threadIdx.x) * 2;
        o[0] = a; o[1] = b;
    }
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        const int w = i / C, c = i - w * C;
        float a = 0.f;
        for (int rs = 0; rs < tpr; ++rs) a += shc[w][rs * q + (c >> 2)][c & 3];
        part_c[(((long)n * gridDim.x + blockIdx.x) * 2 + w) * C + c] = a;
    }
}

template <typename I>
__global__ void gnb_apply_kernel(const float *__restrict__ x, const float *__restrict__ dy, const float *__restrict__ gamma,
                                 const double *__restrict__ stats, const double *__restrict__ stats2, int N, int HW, int C, int G, float eps,
                                 float *__restrict__ dx)
{
    const int q = C / 4;
    const I i = (I)blockIdx.x * blockDim.x + threadIdx.x;
    const I total = (I)N * HW * q;
    if (i >= total) return;
    const int c4 = (int)(i % (I)q);
    const int n = (int)((i / (I)q) / (I)HW);
    const int g = (c4 * 4) / (C / G);
    const double cnt = (double)HW * (C / G);
    float mean, rstd;
    gn_mean_rstd(stats, n, g, G, cnt, eps, mean, rstd);
    const float m1 = (float)(stats2[((long)n * G + g) * 2] / cnt), m2 = (float)(stats2[((long)n * G + g) * 2 + 1] / cnt);
    const f32x4 xh = (*reinterpret_cast<const f32x4 *>(x + (long)i * 4) - mean) * rstd;
    const f32x4 gg = *reinterpret_cast<const f32x4 *>(dy + (long)i * 4) * *reinterpret_cast<const f32x4 *>(gamma + c4 * 4);
    *reinterpret_cast<f32x4 *>(dx + (long)i * 4) = (gg - m1 - xh * m2) * rstd;
}

// y = GN(x) * gamma + beta  [+ bilinear_resize(up)[N,hu,wu,C] -> (H,W)]  [relu]
// PX consecutive pixels of one frame per thread (H * W % PX == 0): the group's mean / rstd (double divisions and a square root) are
// formed once for them and PX 16-B loads are in flight per lane
template <typename I, int PX>
__global__ void gn_apply_px_kernel(const float *__restrict__ x, const double *__restrict__ stats, const float *__restrict__ gamma,
                                   const float *__restrict__ beta, int N, int H, int W, int C, int G, float eps,
                                   const float *__restrict__ up, int hu, int wu, int relu, float *__restrict__ y)
{
    const int q = C / 4;
    const I i = (I)blockIdx.x * blockDim.x + threadIdx.x;
    const I total = (I)N * H * W / PX * q;
    if (i >= total) return;
    const int c4 = (int)(i % (I)q);
    const I pix0 = (i / (I)q) * PX;
    const int n = (int)(pix0 / ((I)H * W));
    const int g = (c4 * 4) / (C / G);
    const double cnt = (double)H * W * (C / G);
    const double mu = stats[((long)n * G + g) * 2] / cnt;
    const double var = stats[((long)n * G + g) * 2 + 1] / cnt - mu * mu;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float mean = (float)mu;
    const f32x4 ga = *reinterpret_cast<const f32x4 *>(gamma + c4 * 4);
    const f32x4 be = *reinterpret_cast<const f32x4 *>(beta + c4 * 4);
    f32x4 v[PX];
#pragma unroll
    for (int k = 0; k < PX; ++k) v[k] = *reinterpret_cast<const f32x4 *>(x + ((long)pix0 + k) * C + c4 * 4);
#pragma unroll
    for (int k = 0; k < PX; ++k) {
        f32x4 o = (v[k] - mean) * rstd * ga + be;
        if (up) {
            // F.interpolate(bilinear, align_corners=False) source index rule (msdeformattn.py:349)
            const I pix = pix0 + k;
            const int px = (int)(pix % (I)W), py = (int)((pix / (I)W) % (I)H);
            float sy = ((float)hu / H) * (py + 0.5f) - 0.5f; if (sy < 0.f) sy = 0.f;
            float sx = ((float)wu / W) * (px + 0.5f) - 0.5f; if (sx < 0.f) sx = 0.f;
            const int y0 = (int)sy, x0 = (int)sx, y1 = y0 + (y0 < hu - 1 ? 1 : 0), x1 = x0 + (x0 < wu - 1 ? 1 : 0);
            const float ly = sy - y0, lx = sx - x0, hy = 1.f - ly, hx = 1.f - lx;
            const float *ub = up + (long)n * hu * wu * C + c4 * 4;
            const f32x4 a = *reinterpret_cast<const f32x4 *>(ub + ((long)y0 * wu + x0) * C);
            const f32x4 b = *reinterpret_cast<const f32x4 *>(ub + ((long)y0 * wu + x1) * C);
            const f32x4 c = *reinterpret_cast<const f32x4 *>(ub + ((long)y1 * wu + x0) * C);
            const f32x4 d = *reinterpret_cast<const f32x4 *>(ub + ((long)y1 * wu + x1) * C);
            o += hy * (hx * a + lx * b) + ly * (hx * c + lx * d);
        }
        if (relu) { o[0] = fmaxf(o[0], 0.f); o[1] = fmaxf(o[1], 0.f); o[2] = fmaxf(o[2], 0.f); o[3] = fmaxf(o[3], 0.f); }
        *reinterpret_cast<f32x4 *>(y + ((long)pix0 + k) * C + c4 * 4) = o;
    }
}

template <typename I>
__global__ void gn_apply_kernel(const float *__restrict__ x, const double *__restrict__ stats, const float *__restrict__ gamma,
                                const float *__restrict__ beta, int N, int H, int W, int C, int G, float eps,
                                const float *__restrict__ up, int hu, int wu, int relu, float *__restrict__ y)
{
    const int q = C / 4;
    const I i = (I)blockIdx.x * blockDim.x + threadIdx.x;
    const I total = (I)N * H * W * q;
    if (i >= total) return;
    const int c4 = (int)(i % (I)q);
    const I pix = i / (I)q;
    const int n = (int)(pix / ((I)H * W));
    const int g = (c4 * 4) / (C / G);
    const double cnt = (double)H * W * (C / G);
    const double mu = stats[((long)n * G + g) * 2] / cnt;
    const double var = stats[((long)n * G + g) * 2 + 1] / cnt - mu * mu;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float mean = (float)mu;
    f32x4 v = *reinterpret_cast<const f32x4 *>(x + (long)i * 4);
    const f32x4 ga = *reinterpret_cast<const f32x4 *>(gamma + c4 * 4);
    const f32x4 be = *reinterpret_cast<const f32x4 *>(beta + c4 * 4);
    v = (v - mean) * rstd * ga + be;
    if (up) {
        // F.interpolate(bilinear, align_corners=False) source index rule (msdeformattn.py:349)
        const int px = (int)(pix % (I)W), py = (int)((pix / (I)W) % (I)H);
        float sy = ((float)hu / H) * (py + 0.5f) - 0.5f; if (sy < 0.f) sy = 0.f;
        float sx = ((float)wu / W) * (px + 0.5f) - 0.5f; if (sx < 0.f) sx = 0.f;
        const int y0 = (int)sy, x0 = (int)sx, y1 = y0 + (y0 < hu - 1 ? 1 : 0), x1 = x0 + (x0 < wu - 1 ? 1 : 0);
        const float ly = sy - y0, lx = sx - x0, hy = 1.f - ly, hx = 1.f - lx;
        const float *ub = up + (long)n * hu * wu * C + c4 * 4;
        const f32x4 a = *reinterpret_cast<const f32x4 *>(ub + ((long)y0 * wu + x0) * C);
        const f32x4 b = *reinterpret_cast<const f32x4 *>(ub + ((long)y0 * wu + x1) * C);
        const f32x4 c = *reinterpret_cast<const f32x4 *>(ub + ((long)y1 * wu + x0) * C);
        const f32x4 d = *reinterpret_cast<const f32x4 *>(ub + ((long)y1 * wu + x1) * C);
        v += hy * (hx * a + lx * b) + ly * (hx * c + lx * d);
    }
    if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
    *reinterpret_cast<f32x4 *>(y + (long)i * 4) = v;
}

// ---------------------------------------------------------------------------------------------------
// y = LayerNorm(x + res) over the last dim C (C % 4 == 0, C <= 1024): one wavefront per row.
__global__ __launch_bounds__(256) void layernorm_kernel(const float *__restrict__ x, const float *__restrict__ res,
                                                        const float *__restrict__ gamma, const float *__restrict__ beta,
                                                        long rows, int C, float eps, float *__restrict__ y)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int q = C / 4;
    f32x4 v[4];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c4 = lane + 64 * k;
        v[k] = f32x4(0.f);
        if (c4 < q) {
            v[k] = *reinterpret_cast<const f32x4 *>(x + row * C + c4 * 4);
            if (res) v[k] += *reinterpret_cast<const f32x4 *>(res + row * C + c4 * 4);
            s += v[k][0] + v[k][1] + v[k][2] + v[k][3];
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c4 = lane + 64 * k;
        if (c4 < q) {
            const f32x4 d = v[k] - mean;
            ss += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
        }
    }
    const float rstd = 1.f / sqrtf(wave_sum(ss) / (float)C + eps);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c4 = lane + 64 * k;
        if (c4 < q) {
            const f32x4 ga = *reinterpret_cast<const f32x4 *>(gamma + c4 * 4);
            const f32x4 be = *reinterpret_cast<const f32x4 *>(beta + c4 * 4);
            *reinterpret_cast<f32x4 *>(y + row * C + c4 * 4) = (v[k] - mean) * rstd * ga + be;
        }
    }
}

// C == 256 (every LayerNorm of the path): a wave takes RW consecutive rows, one 16-B vector per lane and row, all RW loads in
// flight before the first reduction -- with one 1-KB row per wave the launch had ~8 MB in flight on the whole chip and ran at
// 4.9 TB/s of its two passes.  Same arithmetic per row as layernorm_kernel (same bits).
template <int RW>
__global__ __launch_bounds__(256) void layernorm256_kernel(const float *__restrict__ x, const float *__restrict__ res,
                                                           const float *__restrict__ gamma, const float *__restrict__ beta,
                                                           long rows, float eps, float *__restrict__ y)
{
    const int lane = threadIdx.x & 63;
    const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RW;
    if (row0 >= rows) return;
    f32x4 v[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        const long row = row0 + i < rows ? row0 + i : rows - 1;              // tail rows are recomputed, not stored
        v[i] = *reinterpret_cast<const f32x4 *>(x + row * 256 + lane * 4);
    }
    if (res) {
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            const long row = row0 + i < rows ? row0 + i : rows - 1;
            v[i] += *reinterpret_cast<const f32x4 *>(res + row * 256 + lane * 4);
        }
    }
    const f32x4 ga = *reinterpret_cast<const f32x4 *>(gamma + lane * 4);
    const f32x4 be = *reinterpret_cast<const f32x4 *>(beta + lane * 4);
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        float s = 0.f;
        s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        const float mean = wave_sum(s) / 256.f;
        const f32x4 d = v[i] - mean;
        float ss = 0.f;
        ss += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
        const float rstd = 1.f / sqrtf(wave_sum(ss) / 256.f + eps);
        if (row0 + i < rows) *reinterpret_cast<f32x4 *>(y + (row0 + i) * 256 + lane * 4) = (v[i] - mean) * rstd * ga + be;
    }
}

// y[n][r][:] = x[n][r][:] + b[r % brows][:]   (broadcast add of a [brows, C] table over the batch)
template <typename I>
__global__ void add_bcast_kernel(const float *__restrict__ x, const float *__restrict__ b, long n4, long b4,
                                 float *__restrict__ y)
{
    const I i = (I)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (I)n4) return;
    *reinterpret_cast<f32x4 *>(y + (long)i * 4) =
        *reinterpret_cast<const f32x4 *>(x + (long)i * 4) + *reinterpret_cast<const f32x4 *>(b + (long)(i % (I)b4) * 4);
}

// ---------------------------------------------------------------------------------------------------
// Sine position encodings, written token-major [T*H*W, C] (== NHWC), plus an optional per-channel addend
// (the level embedding).  2-D: mask2former/modeling/transformer_decoder/position_encoding.py:29-52
// (normalize=True, scale 2*pi, temperature 1e4); 3-D: mask2former_video/.../position_encoding.py:29-57
// (z term over frames is ADDED to cat(pos_y, pos_x), :44-56).  T == 0 selects the 2-D form.
__global__ void pe_sine_kernel(int T, int H, int W, int F, const float *__restrict__ addc, float *__restrict__ out)
{
    const int C = 2 * F;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long tt = T > 0 ? T : 1;
    const long total = tt * H * W * C;
    if (i >= total) return;
    const int c = (int)(i % C);
    long t = i / C;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H);
    const int z = (int)(t / H);
    const float scale = 6.283185307179586f, eps = 1e-6f;
    const bool is_y = c < F;
    const int j = is_y ? c : c - F;
    const float e = is_y ? (float)(y + 1) / ((float)H + eps) * scale : (float)(x + 1) / ((float)W + eps) * scale;
    const float dim_t = powf(10000.f, (float)(2 * (j / 2)) / (float)F);
    const float a = e / dim_t;
    float v = (j & 1) ? cosf(a) : sinf(a);
    if (T > 0) {
        const float ez = (float)(z + 1) / ((float)T + eps) * scale;
        const float dz = powf(10000.f, (float)(2 * (c / 2)) / (float)C);
        const float az = ez / dz;
        v += (c & 1) ? cosf(az) : sinf(az);
    }
    if (addc) v += addc[c];
    out[i] = v;
}

}  // namespace

namespace {
__global__ __launch_bounds__(256) void zero_kernel(uint32_t *__restrict__ p, size_t n, uint8_t *__restrict__ tail, int ntail)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0u;
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
}
}  // namespace

int s2d_zero_async(void *p, size_t bytes, hipStream_t stream)
{
    if (bytes == 0) return S2D_OK;
    uint8_t *b = reinterpret_cast<uint8_t *>(p);
    const int head = (int)((4 - ((uintptr_t)b & 3)) & 3);           // unaligned leading bytes (views)
    if ((size_t)head >= bytes) {
        hipLaunchKernelGGL(zero_kernel, dim3(1), dim3(256), 0, stream, nullptr, (size_t)0, b, (int)bytes);
        S2D_CHECK_LAUNCH();
        return S2D_OK;
    }
    if (head) hipLaunchKernelGGL(zero_kernel, dim3(1), dim3(256), 0, stream, nullptr, (size_t)0, b, head);
    const size_t n = (bytes - head) / 4;
    const int ntail = (int)((bytes - head) & 3);
    const int nb = (int)(n / 256 + 1 < 4096 ? n / 256 + 1 : 4096);
    hipLaunchKernelGGL(zero_kernel, dim3(nb), dim3(256), 0, stream, reinterpret_cast<uint32_t *>(b + head), n, b + head + 4 * n,
                       ntail);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

extern "C" {

int s2d_normalize_pad_nhwc4_f32(const uint8_t *frames, int F, int H0, int W0, int Hp, int Wp, const float *mean3_host,
                                const float *std3_host, float *out, hipStream_t stream)
{
    if (Hp < H0 || Wp < W0) return S2D_ERR_ARG;
    const long total = (long)F * Hp * Wp;
    if (total == 0) return S2D_OK;
    f32x4 m = {mean3_host[0], mean3_host[1], mean3_host[2], 0.f}, s = {std3_host[0], std3_host[1], std3_host[2], 1.f};
    if (total < (1L << 31))
        hipLaunchKernelGGL(normalize_pad_kernel<unsigned int>, dim3(cdiv(total, 256)), dim3(256), 0, stream, frames, F, H0, W0, Hp, Wp,
                           m, s, out);
    else
        hipLaunchKernelGGL(normalize_pad_kernel<long>, dim3(cdiv(total, 256)), dim3(256), 0, stream, frames, F, H0, W0, Hp, Wp, m, s,
                           out);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

int s2d_maxpool3x3s2_nhwc_f32(const float *x, int N, int H, int W, int C, float *y, hipStream_t stream)
{
    if (C & 3) return S2D_ERR_ARG;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long total = (long)N * Ho * Wo * (C / 4);
    if (total == 0) return S2D_OK;
    if (total < (1L << 31))
        hipLaunchKernelGGL(maxpool_kernel<unsigned int>, dim3(cdiv(total, 256)), dim3(256), 0, stream, x, N, H, W, C, Ho, Wo, y);
    else
        hipLaunchKernelGGL(maxpool_kernel<long>, dim3(cdiv(total, 256)), dim3(256), 0, stream, x, N, H, W, C, Ho, Wo, y);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

int s2d_maxpool3x3s2_nhwc_idx_f32(const float *x, int N, int H, int W, int C, float *y, unsigned char *argmax, hipStream_t stream)
{
    if ((C & 3) || !argmax || (reinterpret_cast<uintptr_t>(argmax) & 3)) return S2D_ERR_ARG;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long total = (long)N * Ho * Wo * (C / 4);
    if (total == 0) return S2D_OK;
    if (total >= (1L << 31)) return S2D_ERR_ARG;
    hipLaunchKernelGGL((maxpool_kernel<unsigned int, true>), dim3(cdiv(total, 256)), dim3(256), 0, stream, x, N, H, W, C, Ho, Wo, y,
                       reinterpret_cast<unsigned int *>(argmax));
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

static int gn_rows_per_blk(long HW) { return (int)max(256L, (long)cdiv(HW, 240L)); }

long s2d_groupnorm_workspace_doubles(int N, int H, int W, int G)
{
    const long HW = (long)H * W;
    if (N <= 0 || HW <= 0) return 0;
    return 2L * N * G * (1 + cdiv(HW, (long)gn_rows_per_blk(HW)));
}

int s2d_groupnorm_nhwc_f32(const float *x, int N, int H, int W, int C, int G, const float *gamma, const float *beta,
                           float eps, const float *up, int hu, int wu, int relu, double *stats_ws, float *y,
                           hipStream_t stream)
{
    if ((C & 3) || C / 4 > 256 || 256 % (C / 4) || C % G || (C / G) & 3 || G > 256) return S2D_ERR_ARG;
    const long HW = (long)H * W;
    if (N == 0 || HW == 0) return S2D_OK;
    const int rows_per_blk = gn_rows_per_blk(HW);
    const int nblk = (int)cdiv(HW, (long)rows_per_blk);
    double *part = stats_ws + 2L * N * G;
    hipLaunchKernelGGL(gn_stats_kernel, dim3(nblk, N), dim3(256), 0, stream, x, (int)HW, C, G, rows_per_blk, part);
    hipLaunchKernelGGL(gn_reduce_kernel, dim3(N * G), dim3(64), 0, stream, part, nblk, G, stats_ws);
    const long total = (long)N * HW * (C / 4);
    if (HW % 4 == 0 && total >= (1L << 20) && total < (1L << 31))
        hipLaunchKernelGGL((gn_apply_px_kernel<unsigned int, 4>), dim3(cdiv(total / 4, 256)), dim3(256), 0, stream, x, stats_ws, gamma, beta, N, H,
                           W, C, G, eps, up, hu, wu, relu, y);
    else if (total < (1L << 31))
        hipLaunchKernelGGL(gn_apply_kernel<unsigned int>, dim3(cdiv(total, 256)), dim3(256), 0, stream, x, stats_ws, gamma, beta, N, H, W,
                           C, G, eps, up, hu, wu, relu, y);
    else
        hipLaunchKernelGGL(gn_apply_kernel<long>, dim3(cdiv(total, 256)), dim3(256), 0, stream, x, stats_ws, gamma, beta, N, H, W, C, G,
                           eps, up, hu, wu, relu, y);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

/* part_c rows (float [rows][2][C]) and workspace doubles of s2d_groupnorm_backward_f32 */
long s2d_groupnorm_backward_blocks(int N, int H, int W)
{
    const long HW = (long)H * W;
    if (N <= 0 || HW <= 0) return 0;
    return (long)N * cdiv(HW, (long)gn_rows_per_blk(HW));
}

int s2d_groupnorm_backward_f32(const float *x, const float *dy, const float *gamma, int N, int H, int W, int C, int G, float eps,
                               double *ws, float *dx, float *part_c, hipStream_t stream)
{
    if ((C & 3) || C / 4 > 256 || 256 % (C / 4) || C % G || (C / G) & 3 || G > 256) return S2D_ERR_ARG;
    const long HW = (long)H * W;
    if (N == 0 || HW == 0) return S2D_OK;
    const int rows_per_blk = gn_rows_per_blk(HW);
    const int nblk = (int)cdiv(HW, (long)rows_per_blk);
    const long half = 2L * N * G * (1 + nblk);                       // = s2d_groupnorm_workspace_doubles
    double *stats = ws, *part = ws + 2L * N * G, *stats2 = ws + half, *part2 = ws + half + 2L * N * G;
    hipLaunchKernelGGL(gn_stats_kernel, dim3(nblk, N), dim3(256), 0, stream, x, (int)HW, C, G, rows_per_blk, part);
    hipLaunchKernelGGL(gn_reduce_kernel, dim3(N * G), dim3(64), 0, stream, part, nblk, G, stats);
    hipLaunchKernelGGL(gnb_sums_kernel, dim3(nblk, N), dim3(256), 0, stream, x, dy, gamma, stats, (int)HW, C, G, eps, rows_per_blk, part2, part_c);
    hipLaunchKernelGGL(gn_reduce_kernel, dim3(N * G), dim3(64), 0, stream, part2, nblk, G, stats2);
    const long total = (long)N * HW * (C / 4);
    if (total < (1L << 31))
        hipLaunchKernelGGL(gnb_apply_kernel<unsigned int>, dim3(cdiv(total, 256)), dim3(256), 0, stream, x, dy, gamma, stats, stats2, N, (int)HW, C,
                           G, eps, dx);
    else
        hipLaunchKernelGGL(gnb_apply_kernel<long>, dim3(cdiv(total, 256)), dim3(256), 0, stream, x, dy, gamma, stats, stats2, N, (int)HW, C, G, eps,
                           dx);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

int s2d_layernorm_f32(const float *x, const float *res, const float *gamma, const float *beta, long rows, int C,
                      float eps, float *y, hipStream_t stream)
{
    if ((C & 3) || C > 1024) return S2D_ERR_ARG;
    if (rows == 0) return S2D_OK;
    if (C == 256 && rows >= 4096)
        hipLaunchKernelGGL(layernorm256_kernel<4>, dim3(cdiv(rows, 16)), dim3(256), 0, stream, x, res, gamma, beta, rows, eps, y);
    else
        hipLaunchKernelGGL(layernorm_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, stream, x, res, gamma, beta, rows, C, eps, y);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

int s2d_add_bcast_f32(const float *x, const float *b, long n, long bn, float *y, hipStream_t stream)
{
    if ((n & 3) || (bn & 3) || bn == 0 || n % bn) return S2D_ERR_ARG;
    if (n == 0) return S2D_OK;
    if (n / 4 < (1L << 31))
        hipLaunchKernelGGL(add_bcast_kernel<unsigned int>, dim3(cdiv(n / 4, 256)), dim3(256), 0, stream, x, b, n / 4, bn / 4, y);
    else
        hipLaunchKernelGGL(add_bcast_kernel<long>, dim3(cdiv(n / 4, 256)), dim3(256), 0, stream, x, b, n / 4, bn / 4, y);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

int s2d_pe_sine_f32(int T, int H, int W, int num_pos_feats, const float *add_c, float *out, hipStream_t stream)
{
    const long total = (long)(T > 0 ? T : 1) * H * W * 2 * num_pos_feats;
    if (total == 0) return S2D_OK;
    hipLaunchKernelGGL(pe_sine_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, T, H, W, num_pos_feats, add_c, out);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

/* ---- timing helpers (bench.py roofline): HIP events without the system-scope fence of the default flags, so that
 * bracketing every dense launch does not flush the caches the next kernel wants warm ------------------------------- */
long s2d_prof_event_create(void)
{
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) return 0;
    return (long)(intptr_t)e;
}

int s2d_prof_event_record(long ev, hipStream_t stream)
{
    return hipEventRecord((hipEvent_t)(intptr_t)ev, stream) == hipSuccess ? S2D_OK : S2D_ERR_LAUNCH;
}

/* milliseconds between two recorded events as a double stored through `out_ms` (host pointer); synchronises on `b` */
int s2d_prof_event_elapsed(long a, long b, double *out_ms)
{
    float ms = 0.f;
    if (hipEventSynchronize((hipEvent_t)(intptr_t)b) != hipSuccess) return S2D_ERR_LAUNCH;
    if (hipEventElapsedTime(&ms, (hipEvent_t)(intptr_t)a, (hipEvent_t)(intptr_t)b) != hipSuccess) return S2D_ERR_LAUNCH;
    *out_ms = (double)ms;
    return S2D_OK;
}

int s2d_prof_event_destroy(long ev)
{
    return hipEventDestroy((hipEvent_t)(intptr_t)ev) == hipSuccess ? S2D_OK : S2D_ERR_LAUNCH;
}

}  // extern "C"
