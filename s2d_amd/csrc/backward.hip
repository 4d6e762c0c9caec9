// Training-step callers before the optimizer: gradients of the dense layers, built on the forward
// NT-GEMM / implicit-GEMM kernels (same split-fp16 x3 arithmetic, fp32-class accuracy):
//   dX = dY . W            -> an NT GEMM against W^T (weights are small: transposed once per step);
//   dW = dY^T . X          -> a contraction over the M rows (3e5 .. 9e5 of them): both operands are transposed once
//                             (HBM-bound 64 x 64 LDS-tiled transpose), the contraction is cut into S slices that run as
//                             the batch dimension of one NT GEMM, and the S partial results are added in a fixed order
//                             (reproducible; no atomics);
//   db = column sums of dY -> per-slice partial sums + the same fixed-order reduction.
// This file holds the three HBM-bound helpers; the GEMMs themselves are s2d_gemm_nt_f32 launches (s2d_amd/backward.py).
#include "common.h"

namespace {

// out[c][r] = in[r][c]; in [R][ldi] (C <= ldi), out [C][ldo] (R <= ldo).  64 x 64 tile through LDS, 16-B global accesses on
// both sides when the shapes allow.
__global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ in, long R, long C, long ldi, float *__restrict__ out, long ldo)
{
    __shared__ float tile[64][65];
    const long r0 = (long)blockIdx.x * 64, c0 = (long)blockIdx.y * 64;          // rows (millions of positions) on grid.x
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;          // 16 x 16 threads, each 4 columns x 4 rows
    const bool vin = ((ldi & 3) == 0) && ((reinterpret_cast<uintptr_t>(in) & 15) == 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long r = r0 + ty + 16 * i, c = c0 + 4 * tx;
        if (r < R) {
            if (vin && c + 3 < C) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(in + r * ldi + c);
                tile[ty + 16 * i][4 * tx] = v[0]; tile[ty + 16 * i][4 * tx + 1] = v[1];
                tile[ty + 16 * i][4 * tx + 2] = v[2]; tile[ty + 16 * i][4 * tx + 3] = v[3];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c + j < C) tile[ty + 16 * i][4 * tx + j] = in[r * ldi + c + j];
            }
        }
    }
    __syncthreads();
    const bool vout = ((ldo & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long c = c0 + ty + 16 * i, r = r0 + 4 * tx;            // output row = input column
        if (c < C) {
            if (vout && r + 3 < R) {
                const f32x4 v = {tile[4 * tx][ty + 16 * i], tile[4 * tx + 1][ty + 16 * i], tile[4 * tx + 2][ty + 16 * i], tile[4 * tx + 3][ty + 16 * i]};
                *reinterpret_cast<f32x4 *>(out + c * ldo + r) = v;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (r + j < R) out[c * ldo + r + j] = tile[4 * tx + j][ty + 16 * i];
            }
        }
    }
}

// out[i] = beta * out[i] + sum_s part[s][i] in a fixed order: eight interleaved chains (slice s goes to chain s % 8, ascending
// within a chain), combined pairwise at the end.  One chain made every thread wait out S dependent HBM round trips (36 us for
// the 16.8 MB of a 256 x 256 weight gradient); eight loads in flight per thread bring the launch to its bandwidth.
__global__ __launch_bounds__(256) void reduce_slices_kernel(const float *__restrict__ part, int S, long n, long stride, float beta,
                                                            float *__restrict__ out)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float *p = part + i;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int s = 0;
    for (; s + 8 <= S; s += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[(long)(s + j) * stride];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += v[j];
    }
    for (int j = 0; s < S; ++s, ++j) a[j] += p[(long)s * stride];
    const float sum = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    out[i] = beta != 0.f ? beta * out[i] + sum : sum;
}

// part[s][c] = sum over the rows of slice s of in[r][c]  (rows_per_slice rows each; sequential in r: fixed order)
__global__ __launch_bounds__(256) void colsum_slices_kernel(const float *__restrict__ in, long R, long C, long ldi, long rows_per_slice,
                                                            float *__restrict__ part)
{
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    const int s = blockIdx.y;
    if (c >= C) return;
    const long r0 = (long)s * rows_per_slice, r1 = r0 + rows_per_slice < R ? r0 + rows_per_slice : R;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};        // eight interleaved chains (row r goes to chain (r - r0) % 8), combined in a fixed order
    long r = r0;
    for (; r + 7 < r1; r += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = in[(r + j) * ldi + c];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += v[j];
    }
    for (int j = 0; r < r1; ++r, ++j) a[j] += in[r * ldi + c];
    part[(long)s * C + c] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}

// LayerNorm backward over the last dim C (C % 4 == 0, C <= 1024), one wavefront per row, the input row recomputed into
// registers as the forward does (x + res; statistics in the forward's order):
//   xhat = (x - mean) * rstd,  g = dy * gamma,  dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),
//   dgamma = sum_rows dy * xhat,  dbeta = sum_rows dy: each workgroup walks a fixed set of rows in a fixed order and
//   leaves one partial row per quantity (part [nblocks][2][C]), finished by s2d_reduce_slices_f32 (reproducible).
// Rows per wavefront (a workgroup = 4 wavefronts): 64 for the long tensors, fewer for short ones so that the rows spread over the
// chip -- the video decoder's 200-row LayerNorms ran as ONE workgroup whose waves walked 64 rows one memory round trip at a time
// (108 us per call, 37 calls per backward; round 5).  A function of `rows` alone: the partial sums' order stays fixed per shape.
constexpr int LNB_ROWS = 64;
static int lnb_rows_per_wave(long rows)
{
    int r = 1;
    while (r < LNB_ROWS && (long)r * 2048 < rows) r *= 2;
    return r;
}
__global__ __launch_bounds__(256) void layernorm_backward_kernel(const float *__restrict__ x, const float *__restrict__ res,
                                                                 const float *__restrict__ dy, const float *__restrict__ gamma, long rows,
                                                                 int C, float eps, float *__restrict__ dx, float *__restrict__ part, int rpw)
{
    __shared__ float red[4][2][1024];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int q = C / 4;
    f32x4 ga[4], dg[4], db[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c4 = lane + 64 * k;
        ga[k] = c4 < q ? *reinterpret_cast<const f32x4 *>(gamma + c4 * 4) : f32x4(0.f);
        dg[k] = f32x4(0.f); db[k] = f32x4(0.f);
    }
    const long r0 = ((long)blockIdx.x * 4 + wv) * rpw;
    for (long row = r0; row < r0 + rpw && row < rows; ++row) {
        f32x4 v[4], d[4];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c4 = lane + 64 * k;
            v[k] = f32x4(0.f); d[k] = f32x4(0.f);
            if (c4 < q) {
                v[k] = *reinterpret_cast<const f32x4 *>(x + row * C + c4 * 4);
                if (res) v[k] += *reinterpret_cast<const f32x4 *>(res + row * C + c4 * 4);
                d[k] = *reinterpret_cast<const f32x4 *>(dy + row * C + c4 * 4);
                s += v[k][0] + v[k][1] + v[k][2] + v[k][3];
            }
        }
        const float mean = wave_sum(s) / (float)C;
        float ss = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c4 = lane + 64 * k;
            if (c4 < q) {
                v[k] = v[k] - mean;
                ss += v[k][0] * v[k][0] + v[k][1] * v[k][1] + v[k][2] * v[k][2] + v[k][3] * v[k][3];
            }
        }
        const float rstd = 1.f / sqrtf(wave_sum(ss) / (float)C + eps);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] = v[k] * rstd;                               // xhat (0 in the lanes beyond C)
            const f32x4 g = d[k] * ga[k];
            sg += g[0] + g[1] + g[2] + g[3];
            sgx += g[0] * v[k][0] + g[1] * v[k][1] + g[2] * v[k][2] + g[3] * v[k][3];
            dg[k] += d[k] * v[k];
            db[k] += d[k];
        }
        const float mg = wave_sum(sg) / (float)C, mgx = wave_sum(sgx) / (float)C;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c4 = lane + 64 * k;
            if (c4 < q) *reinterpret_cast<f32x4 *>(dx + row * C + c4 * 4) = (d[k] * ga[k] - mg - v[k] * mgx) * rstd;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c4 = lane + 64 * k;
        if (c4 < q) {
            *reinterpret_cast<f32x4 *>(&red[wv][0][c4 * 4]) = dg[k];
            *reinterpret_cast<f32x4 *>(&red[wv][1][c4 * 4]) = db[k];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        const int w = i / C, c = i - w * C;
        part[((long)blockIdx.x * 2 + w) * C + c] = (red[0][w][c] + red[1][w][c]) + (red[2][w][c] + red[3][w][c]);
    }
}

// C == 256: a row is one float4 per lane, so RW rows' loads (x, residual, dy: up to 3 RW 16-B loads per lane) are issued before the first
// reduction -- the general form above walks its rows one memory round trip at a time (0.57 ms on the encoder's 309 120 rows against
// 0.15 ms of traffic).  Same lane-to-column map, same row order, same arithmetic: the same bits.
template <int RW>
__global__ __launch_bounds__(256) void layernorm256_backward_kernel(const float *__restrict__ x, const float *__restrict__ res,
                                                                    const float *__restrict__ dy, const float *__restrict__ gamma, long rows,
                                                                    float eps, float *__restrict__ dx, float *__restrict__ part, int rpw)
{
    constexpr int C = 256;                                                    // rpw % RW == 0
    __shared__ float red[4][2][C];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const f32x4 ga = *reinterpret_cast<const f32x4 *>(gamma + lane * 4);
    f32x4 dg = f32x4(0.f), db = f32x4(0.f);
    const long r0 = ((long)blockIdx.x * 4 + wv) * rpw;
    for (long rb = r0; rb < r0 + rpw && rb < rows; rb += RW) {
        f32x4 v[RW], d[RW];
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            const long row = rb + i < rows ? rb + i : rows - 1;               // tail rows are re-read, not used
            v[i] = *reinterpret_cast<const f32x4 *>(x + row * C + lane * 4);
            d[i] = *reinterpret_cast<const f32x4 *>(dy + row * C + lane * 4);
        }
        if (res) {
#pragma unroll
            for (int i = 0; i < RW; ++i) {
                const long row = rb + i < rows ? rb + i : rows - 1;
                v[i] += *reinterpret_cast<const f32x4 *>(res + row * C + lane * 4);
            }
        }
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            if (rb + i >= rows) break;                                        // wave-uniform
            float s = 0.f;
            s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
            const float mean = wave_sum(s) / (float)C;
            f32x4 xc = v[i] - mean;
            float ss = 0.f;
            ss += xc[0] * xc[0] + xc[1] * xc[1] + xc[2] * xc[2] + xc[3] * xc[3];
            const float rstd = 1.f / sqrtf(wave_sum(ss) / (float)C + eps);
            xc = xc * rstd;                                                   // xhat
            const f32x4 g = d[i] * ga;
            float sg = 0.f, sgx = 0.f;
            sg += g[0] + g[1] + g[2] + g[3];
            sgx += g[0] * xc[0] + g[1] * xc[1] + g[2] * xc[2] + g[3] * xc[3];
            dg += d[i] * xc;
            db += d[i];
            const float mg = wave_sum(sg) / (float)C, mgx = wave_sum(sgx) / (float)C;
            *reinterpret_cast<f32x4 *>(dx + (rb + i) * C + lane * 4) = (d[i] * ga - mg - xc * mgx) * rstd;
        }
    }
    *reinterpret_cast<f32x4 *>(&red[wv][0][lane * 4]) = dg;
    *reinterpret_cast<f32x4 *>(&red[wv][1][lane * 4]) = db;
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        const int w = i / C, c = i - w * C;
        part[((long)blockIdx.x * 2 + w) * C + c] = (red[0][w][c] + red[1][w][c]) + (red[2][w][c] + red[3][w][c]);
    }
}

// dz = dy * (y > 0) * scale[c]: gradient through y = relu(z * scale + bias) (the conv -> FrozenBN -> ReLU epilogue);
// scale NULL = 1, y NULL = no ReLU.  n elements, C channels innermost, 16-B accesses.
__global__ __launch_bounds__(256) void relu_scale_backward_kernel(const float *__restrict__ dy, const float *__restrict__ y,
                                                                  const float *__restrict__ scale, long n4, int C4, float *__restrict__ dz,
                                                                  float *__restrict__ dres)
{
    // four 16-B pieces per thread, a workgroup's 4 x 256 pieces contiguous per step: all loads of a thread in flight together
    constexpr int U = 4;
    const long base = (long)blockIdx.x * (256 * U) + threadIdx.x;
    f32x4 g[U], v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long i = base + u * 256;
        g[u] = i < n4 ? reinterpret_cast<const f32x4 *>(dy)[i] : f32x4(0.f);
        v[u] = (y && i < n4) ? reinterpret_cast<const f32x4 *>(y)[i] : f32x4(1.f);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long i = base + u * 256;
        if (i >= n4) continue;
        f32x4 t = g[u];
        if (y) { t[0] = v[u][0] > 0.f ? t[0] : 0.f; t[1] = v[u][1] > 0.f ? t[1] : 0.f; t[2] = v[u][2] > 0.f ? t[2] : 0.f; t[3] = v[u][3] > 0.f ? t[3] : 0.f; }
        if (dres) reinterpret_cast<f32x4 *>(dres)[i] = t;                  // the residual branch sees the ReLU mask only
        if (scale) t = t * reinterpret_cast<const f32x4 *>(scale)[i % C4];
        reinterpret_cast<f32x4 *>(dz)[i] = t;
    }
}

// out = a + (y > 0 ? g : 0): a stage output's own gradient g (from the pixel decoder) joins the gradient a that arrives gated from the next
// block, in one pass (the gate pass + the add were two passes over up to 1 GB each)
__global__ __launch_bounds__(256) void relu_gate_add_kernel(const float *__restrict__ a, const float *__restrict__ g, const float *__restrict__ y,
                                                            long n4, float *__restrict__ out)
{
    constexpr int U = 4;
    const long base = (long)blockIdx.x * (256 * U) + threadIdx.x;
    f32x4 va[U], vg[U], vy[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long i = base + u * 256;
        va[u] = i < n4 ? reinterpret_cast<const f32x4 *>(a)[i] : f32x4(0.f);
        vg[u] = i < n4 ? reinterpret_cast<const f32x4 *>(g)[i] : f32x4(0.f);
        vy[u] = i < n4 ? reinterpret_cast<const f32x4 *>(y)[i] : f32x4(0.f);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long i = base + u * 256;
        if (i >= n4) continue;
        f32x4 t = va[u];
        t[0] += vy[u][0] > 0.f ? vg[u][0] : 0.f; t[1] += vy[u][1] > 0.f ? vg[u][1] : 0.f;
        t[2] += vy[u][2] > 0.f ? vg[u][2] : 0.f; t[3] += vy[u][3] > 0.f ? vg[u][3] : 0.f;
        reinterpret_cast<f32x4 *>(out)[i] = t;
    }
}

// Adjoint of F.interpolate(up [N,hu,wu,C] -> (H,W), bilinear, align_corners=False) (the top-down add of the pixel decoder,
// msdeformattn.py:349), as a gather so that it is reproducible: low-resolution pixel (yl, xl) collects w_y * w_x * dy from
// every high-resolution pixel whose source taps include it (the forward's own source rule decides the weights).
__device__ __forceinline__ float tap_weight(float scale, int dst, int in_size, int want)
{
    float s = scale * (dst + 0.5f) - 0.5f;
    if (s < 0.f) s = 0.f;
    const int i0 = (int)s, i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    const float l1 = s - i0, l0 = 1.f - l1;
    return (i0 == want ? l0 : 0.f) + (i1 == want ? l1 : 0.f);
}
__global__ __launch_bounds__(256) void resize_backward_kernel(const float *__restrict__ dy, int N, int H, int W, int C, int hu, int wu,
                                                              float *__restrict__ dup)
{
    const int q = C / 4;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)N * hu * wu * q) return;
    const int c4 = (int)(i % q);
    const long pix = i / q;
    const int xl = (int)(pix % wu), yl = (int)((pix / wu) % hu), n = (int)(pix / ((long)wu * hu));
    const float sy = (float)hu / H, sx = (float)wu / W, ry = (float)H / hu, rx = (float)W / wu;
    // high-resolution rows / columns whose source coordinate can fall in (yl - 1, yl + 1)
    int Y0 = (int)floorf((yl - 0.5f) * ry - 0.5f) - 1, Y1 = (int)ceilf((yl + 1.5f) * ry - 0.5f) + 1;
    int X0 = (int)floorf((xl - 0.5f) * rx - 0.5f) - 1, X1 = (int)ceilf((xl + 1.5f) * rx - 0.5f) + 1;
    Y0 = max(Y0, 0); X0 = max(X0, 0); Y1 = min(Y1, H - 1); X1 = min(X1, W - 1);
    if (yl == hu - 1) Y1 = H - 1;                       // clamped taps at the bottom / right edge all land here
    if (xl == wu - 1) X1 = W - 1;
    f32x4 acc = f32x4(0.f);
    for (int Y = Y0; Y <= Y1; ++Y) {
        const float wy = tap_weight(sy, Y, hu, yl);
        if (wy == 0.f) continue;
        for (int X = X0; X <= X1; ++X) {
            const float wx = tap_weight(sx, X, wu, xl);
            if (wx == 0.f) continue;
            acc += *reinterpret_cast<const f32x4 *>(dy + (((long)n * H + Y) * W + X) * C + c4 * 4) * (wy * wx);
        }
    }
    *reinterpret_cast<f32x4 *>(dup + i * 4) = acc;
}

// 3x3 / stride 2 / pad 1 max-pool backward (detectron2 BasicStem), NHWC, as a gather: an input pixel belongs to at most
// 2 x 2 windows; for each, the window's arg-max is recomputed with the forward's rule (scan in (ky, kx) order, the first
// maximum wins, which is also where torch's max_pool2d backward routes the gradient) and dy is taken if it is this pixel.
__global__ __launch_bounds__(256) void maxpool_backward_kernel(const float *__restrict__ x, const float *__restrict__ dy, int N, int H, int W,
                                                               int C, int Ho, int Wo, float *__restrict__ dx)
{
    const int q = C / 4;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)N * H * W * q) return;
    const int c4 = (int)(i % q);
    const long pix = i / q;
    const int px = (int)(pix % W), py = (int)((pix / W) % H), n = (int)(pix / ((long)W * H));
    const float *xb = x + (long)n * H * W * C + c4 * 4;
    f32x4 acc = f32x4(0.f);
    for (int oy = (py + 0) / 2; oy <= (py + 1) / 2; ++oy) {           // windows with 2*oy - 1 <= py <= 2*oy + 1
        if (oy >= Ho) continue;
        for (int ox = (px + 0) / 2; ox <= (px + 1) / 2; ++ox) {
            if (ox >= Wo) continue;
            f32x4 best = f32x4(-INFINITY);
            int by[4] = {-1, -1, -1, -1}, bx[4] = {-1, -1, -1, -1};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = 2 * oy - 1 + ky;
                if (iy < 0 || iy >= H) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int ix = 2 * ox - 1 + kx;
                    if (ix < 0 || ix >= W) continue;
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(xb + ((long)iy * W + ix) * C);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (v[j] > best[j] || by[j] < 0) { best[j] = v[j]; by[j] = iy; bx[j] = ix; }
                }
            }
            const f32x4 g = *reinterpret_cast<const f32x4 *>(dy + (((long)n * Ho + oy) * Wo + ox) * C + c4 * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (by[j] == py && bx[j] == px) acc[j] += g[j];
        }
    }
    *reinterpret_cast<f32x4 *>(dx + i * 4) = acc;
}

// dx[n][y][x][c] = G[n][y / 2 + 1][x / 2 + 1][((y & 1) * 2 + (x & 1)) * C + c] (* scale[c]) (-> 0 where gate <= 0): the depth-to-space step of a
// stride-2 3 x 3 convolution's input gradient computed as ONE stride-1 2 x 2 convolution of dY with 4 C output channels (one block of C per
// parity class of the input pixel; s2d_amd/backward.py conv_input_grad), 16-B accesses, one thread per four channels of an input pixel.
__global__ __launch_bounds__(256) void pixel_shuffle2_kernel(const float *__restrict__ G, int N, int Hg, int Wg, int C, int H, int W,
                                                             const float *__restrict__ scale, const float *__restrict__ gate,
                                                             float *__restrict__ dx)
{
    const int q = C / 4;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)N * H * W * q) return;
    const int c4 = (int)(i % q);
    const long pix = i / q;
    const int x = (int)(pix % W), y = (int)((pix / W) % H), n = (int)(pix / ((long)W * H));
    const long g = (((long)n * Hg + (y >> 1) + 1) * Wg + (x >> 1) + 1) * (4L * C) + (long)(((y & 1) * 2 + (x & 1)) * C + c4 * 4);
    f32x4 v = *reinterpret_cast<const f32x4 *>(G + g);
    if (scale) v = v * *reinterpret_cast<const f32x4 *>(scale + c4 * 4);
    if (gate) {
        const f32x4 t = *reinterpret_cast<const f32x4 *>(gate + i * 4);
        v[0] = t[0] > 0.f ? v[0] : 0.f; v[1] = t[1] > 0.f ? v[1] : 0.f; v[2] = t[2] > 0.f ? v[2] : 0.f; v[3] = t[3] > 0.f ? v[3] : 0.f;
    }
    *reinterpret_cast<f32x4 *>(dx + i * 4) = v;
}

// The same from the forward's stored arg-max taps (s2d_maxpool3x3s2_nhwc_idx_f32): per window one 4-byte index word and, where this pixel is a
// channel's arg-max, the dy row.
__global__ __launch_bounds__(256) void maxpool_backward_idx_kernel(const unsigned int *__restrict__ idx, const float *__restrict__ dy, int N, int H,
                                                                   int W, int C, int Ho, int Wo, float *__restrict__ dx)
{
    const int q = C / 4;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)N * H * W * q) return;
    const int c4 = (int)(i % q);
    const long pix = i / q;
    const int px = (int)(pix % W), py = (int)((pix / W) % H), n = (int)(pix / ((long)W * H));
    f32x4 acc = f32x4(0.f);
    for (int oy = py / 2; oy <= (py + 1) / 2; ++oy) {
        if (oy >= Ho) continue;
        for (int ox = px / 2; ox <= (px + 1) / 2; ++ox) {
            if (ox >= Wo) continue;
            const unsigned int tap = (unsigned int)((py - (2 * oy - 1)) * 3 + (px - (2 * ox - 1)));
            const long w = (((long)n * Ho + oy) * Wo + ox) * q + c4;
            const unsigned int am = idx[w];
            const unsigned int hit = ((am & 0xFFu) == tap ? 1u : 0u) | (((am >> 8) & 0xFFu) == tap ? 2u : 0u) | (((am >> 16) & 0xFFu) == tap ? 4u : 0u) |
                                     ((am >> 24) == tap ? 8u : 0u);
            if (hit) {
                const f32x4 g = *reinterpret_cast<const f32x4 *>(dy + w * 4);
                acc[0] += (hit & 1u) ? g[0] : 0.f; acc[1] += (hit & 2u) ? g[1] : 0.f; acc[2] += (hit & 4u) ? g[2] : 0.f; acc[3] += (hit & 8u) ? g[3] : 0.f;
            }
        }
    }
    *reinterpret_cast<f32x4 *>(dx + i * 4) = acc;
}

// col[p][(ky*KW + kx)*C + c] = x[n][oy*stride - pad + ky][ox*stride - pad + kx][c] (0 outside), p = (n*Ho + oy)*Wo + ox: the
// explicit im2col of a convolution with few input channels (the 7x7 stem: C = 4, 49 taps), whose weight gradient is then
// ONE sliced contraction dY^T . col instead of 49 of them with a 4-wide output.
__global__ __launch_bounds__(256) void im2col_kernel(const float *__restrict__ x, int N, int H, int W, int C, int KH, int KW, int stride,
                                                     int pad, int Ho, int Wo, float *__restrict__ col)
{
    const int q = C / 4, taps = KH * KW;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)N * Ho * Wo * taps * q) return;
    const int c4 = (int)(i % q);
    const int tap = (int)((i / q) % taps);
    const long p = i / ((long)q * taps);
    const int ox = (int)(p % Wo), oy = (int)((p / Wo) % Ho), n = (int)(p / ((long)Wo * Ho));
    const int iy = oy * stride - pad + tap / KW, ix = ox * stride - pad + tap % KW;
    f32x4 v = f32x4(0.f);
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *reinterpret_cast<const f32x4 *>(x + (((long)n * H + iy) * W + ix) * C + c4 * 4);
    *reinterpret_cast<f32x4 *>(col + i * 4) = v;
}

// two slice reductions of the same slice count in one launch (a weight gradient and the bias gradient that left the TN kernel with it)
__global__ __launch_bounds__(256) void reduce_slices_pair_kernel(const float *__restrict__ partA, long nA, long strideA, float betaA, float *__restrict__ outA,
                                                                 const float *__restrict__ partB, long nB, long strideB, float betaB, float *__restrict__ outB,
                                                                 int S)
{
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    const bool second = i >= nA;
    if (second) i -= nA;
    if (second && i >= nB) return;
    const float *p = (second ? partB : partA) + i;
    const long stride = second ? strideB : strideA;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};                   // the same eight chains and the same final tree as reduce_slices_kernel: same bits
    int s = 0;
    for (; s + 8 <= S; s += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[(long)(s + j) * stride];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += v[j];
    }
    for (int j = 0; s < S; ++s, ++j) a[j] += p[(long)s * stride];
    const float sum = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    float *o = (second ? outB : outA) + i;
    const float beta = second ? betaB : betaA;
    *o = beta == 0.f ? sum : beta * *o + sum;
}

// dst_i += src_i for a list of tensors in ONE launch: table [n][3] = (src pointer, dst pointer, element count) as 64-bit words, a block
// takes one chunk of one tensor (chunk_tensor / chunk_off name it).  The ~350 "param.grad += g" of a training iteration's glue.
__global__ __launch_bounds__(256) void multi_add_kernel(const long *__restrict__ table, const int *__restrict__ chunk_tensor,
                                                        const long *__restrict__ chunk_off, int chunk)
{
    const int t = chunk_tensor[blockIdx.x];
    const float *src = reinterpret_cast<const float *>(table[3 * t]);
    float *dst = reinterpret_cast<float *>(table[3 * t + 1]);
    const long n = table[3 * t + 2], off = chunk_off[blockIdx.x];
    const long end = off + chunk < n ? off + chunk : n;
    if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0 && (off & 3) == 0) {
        const long e4 = off + ((end - off) & ~3L);
        for (long i = off + 4L * threadIdx.x; i < e4; i += 1024) {
            f32x4 a = *reinterpret_cast<const f32x4 *>(src + i), b = *reinterpret_cast<const f32x4 *>(dst + i);
            *reinterpret_cast<f32x4 *>(dst + i) = b + a;
        }
        for (long i = e4 + threadIdx.x; i < end; i += 256) dst[i] += src[i];
    } else {
        for (long i = off + threadIdx.x; i < end; i += 256) dst[i] += src[i];
    }
}

}  // namespace

extern "C" {

int s2d_multi_add_f32(const long *table, const int *chunk_tensor, const long *chunk_off, int nchunks, int chunk, hipStream_t stream)
{
    if (nchunks < 0 || chunk <= 0 || (chunk & 3)) return S2D_ERR_ARG;
    if (nchunks == 0) return S2D_OK;
    if (!table || !chunk_tensor || !chunk_off) return S2D_ERR_ARG;
    hipLaunchKernelGGL(multi_add_kernel, dim3(nchunks), dim3(256), 0, stream, table, chunk_tensor, chunk_off, chunk);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}


int s2d_im2col_nhwc_f32(const float *x, int N, int H, int W, int C, int KH, int KW, int stride, int pad, float *col, hipStream_t stream)
{
    if ((C & 3) || N < 0 || H < 1 || W < 1 || KH < 1 || KW < 1 || stride < 1 || pad < 0) return S2D_ERR_ARG;
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    const long total = (long)N * Ho * Wo * KH * KW * (C / 4);
    if (total == 0) return S2D_OK;
    hipLaunchKernelGGL(im2col_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, x, N, H, W, C, KH, KW, stride, pad, Ho, Wo, col);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

int s2d_maxpool3x3s2_backward_nhwc_f32(const float *x, const float *dy, int N, int H, int W, int C, float *dx, hipStream_t stream)
{
    if ((C & 3) || N < 0 || H < 1 || W < 1) return S2D_ERR_ARG;
    const long total = (long)N * H * W * (C / 4);
    if (total == 0) return S2D_OK;
    hipLaunchKernelGGL(maxpool_backward_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, x, dy, N, H, W, C, (H + 1) / 2, (W + 1) / 2, dx);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

int s2d_pixel_shuffle2_gate_f32(const float *G, int N, int Hg, int Wg, int C, int H, int W, const float *scale, const float *gate, float *dx,
                                hipStream_t stream)
{
    if ((C & 3) || N < 0 || H < 1 || W < 1 || Hg != (H - 1) / 2 + 2 || Wg != (W - 1) / 2 + 2) return S2D_ERR_ARG;
    const long total = (long)N * H * W * (C / 4);
    if (total == 0) return S2D_OK;
    hipLaunchKernelGGL(pixel_shuffle2_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, G, N, Hg, Wg, C, H, W, scale, gate, dx);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

int s2d_relu_gate_add_f32(const float *a, const float *g, const float *y, long n, float *out, hipStream_t stream)
{
    if ((n & 3) || n < 0) return S2D_ERR_ARG;
    if (n == 0) return S2D_OK;
    hipLaunchKernelGGL(relu_gate_add_kernel, dim3(cdiv(n / 4, 1024)), dim3(256), 0, stream, a, g, y, n / 4, out);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

int s2d_maxpool3x3s2_backward_idx_nhwc_f32(const unsigned char *argmax, const float *dy, int N, int H, int W, int C, float *dx, hipStream_t stream)
{
    if ((C & 3) || N < 0 || H < 1 || W < 1 || !argmax || (reinterpret_cast<uintptr_t>(argmax) & 3)) return S2D_ERR_ARG;
    const long total = (long)N * H * W * (C / 4);
    if (total == 0) return S2D_OK;
    hipLaunchKernelGGL(maxpool_backward_idx_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, reinterpret_cast<const unsigned int *>(argmax), dy, N, H,
                       W, C, (H + 1) / 2, (W + 1) / 2, dx);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

int s2d_resize_bilinear_backward_nhwc_f32(const float *dy, int N, int H, int W, int C, int hu, int wu, float *dup, hipStream_t stream)
{
    if ((C & 3) || N < 0 || H < 1 || W < 1 || hu < 1 || wu < 1) return S2D_ERR_ARG;
    const long total = (long)N * hu * wu * (C / 4);
    if (total == 0) return S2D_OK;
    hipLaunchKernelGGL(resize_backward_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, dy, N, H, W, C, hu, wu, dup);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

long s2d_layernorm_backward_blocks(long rows)
{
    const long per = 4L * lnb_rows_per_wave(rows);
    return (rows + per - 1) / per;
}

int s2d_layernorm_backward_f32(const float *x, const float *res, const float *dy, const float *gamma, long rows, int C, float eps,
                               float *dx, float *part, hipStream_t stream)
{
    if ((C & 3) || C > 1024 || rows < 0) return S2D_ERR_ARG;
    if (rows == 0) return S2D_OK;
    const int rpw = lnb_rows_per_wave(rows);
    if (C == 256 && rpw >= 4)
        hipLaunchKernelGGL(layernorm256_backward_kernel<4>, dim3((unsigned int)s2d_layernorm_backward_blocks(rows)), dim3(256), 0, stream, x, res, dy,
                           gamma, rows, eps, dx, part, rpw);
    else
        hipLaunchKernelGGL(layernorm_backward_kernel, dim3((unsigned int)s2d_layernorm_backward_blocks(rows)), dim3(256), 0, stream, x, res, dy, gamma,
                           rows, C, eps, dx, part, rpw);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

int s2d_relu_scale_backward_f32(const float *dy, const float *y, const float *scale, long n, int C, float *dz, float *dres,
                                hipStream_t stream)
{
    if ((n & 3) || (C & 3) || C <= 0 || n % C) return S2D_ERR_ARG;
    if (n == 0) return S2D_OK;
    hipLaunchKernelGGL(relu_scale_backward_kernel, dim3(cdiv(n / 4, 1024)), dim3(256), 0, stream, dy, y, scale, n / 4, C / 4, dz, dres);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

int s2d_transpose_f32(const float *in, long R, long C, long ldi, float *out, long ldo, hipStream_t stream)
{
    if (R < 0 || C < 0 || ldi < C || ldo < R) return S2D_ERR_ARG;
    if (R == 0 || C == 0) return S2D_OK;
    if ((C + 63) / 64 > 65535 || (R + 63) / 64 >= (1L << 31)) return S2D_ERR_ARG;
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(R, 64), cdiv(C, 64)), dim3(256), 0, stream, in, R, C, ldi, out, ldo);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

int s2d_reduce_slices_f32(const float *part, int S, long n, long stride, float beta, float *out, hipStream_t stream)
{
    if (S < 0 || n < 0 || stride < n) return S2D_ERR_ARG;
    if (n == 0) return S2D_OK;
    hipLaunchKernelGGL(reduce_slices_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, part, S, n, stride, beta, out);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

int s2d_reduce_slices_pair_f32(const float *partA, long nA, long strideA, float betaA, float *outA, const float *partB, long nB, long strideB,
                               float betaB, float *outB, int S, hipStream_t stream)
{
    if (S < 0 || nA < 0 || nB < 0 || strideA < nA || strideB < nB) return S2D_ERR_ARG;
    if (nA + nB == 0) return S2D_OK;
    hipLaunchKernelGGL(reduce_slices_pair_kernel, dim3(cdiv(nA + nB, 256)), dim3(256), 0, stream, partA, nA, strideA, betaA, outA, partB, nB, strideB,
                       betaB, outB, S);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

int s2d_colsum_slices_f32(const float *in, long R, long C, long ldi, long rows_per_slice, float *part, hipStream_t stream)
{
    if (R <= 0 || C <= 0 || ldi < C || rows_per_slice <= 0) return S2D_ERR_ARG;
    const long S = (R + rows_per_slice - 1) / rows_per_slice;
    if (S > 65535) return S2D_ERR_ARG;
    hipLaunchKernelGGL(colsum_slices_kernel, dim3(cdiv(C, 256), (int)S), dim3(256), 0, stream, in, R, C, ldi, rows_per_slice, part);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

}  // extern "C"
