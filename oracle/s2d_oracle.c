/*
 * oracle/s2d_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the loops of the S2D hot path that are too slow in
 * numpy.  The loops whose outputs are independent (MSDeformAttn forward, point
 * sampling, bilinear resize) carry OpenMP pragmas so that bench.py's all-core
 * CPU baseline really uses the host's cores (orc_set_threads; results do not
 * depend on the thread count: every output element is computed by one thread
 * in the same order of operations).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; s2d_amd never does.
 *
 * Every function cites the reference lines it restates (paths relative to
 * /root/reference).  Pinned against the golden vectors in tests/golden/ that
 * were produced by the reference's own Python (tests/golden/make_golden.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* threads of the parallel loops below (<= 0: the OpenMP default); returns the count now in force */
int orc_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

/* ------------------------------------------------------------------------- *
 * MSDeformAttn core, forward.
 * model_training/mask2former/modeling/pixel_decoder/ops/functions/ms_deform_attn_func.py:52-72
 * == ops/src/cuda/ms_deform_im2col_cuda.cuh:38-89 (bilinear), :242-304 (kernel).
 * value [N,S,M,D], shapes [L,2]=(H,W), lsi [L], loc [N,Lq,M,L,P,2]=(x,y), w [N,Lq,M,L,P]
 * out [N,Lq,M*D]
 * ------------------------------------------------------------------------- */
#define DEF_MSDA_FWD(NAME, T)                                                                      \
    void NAME(const T *value, const int64_t *shapes, const int64_t *lsi, const T *loc, const T *w, \
              int N, int S, int M, int D, int L, int Lq, int P, T *out)                            \
    {                                                                                              \
        _Pragma("omp parallel for collapse(2) schedule(static)")                                   \
        for (int n = 0; n < N; ++n)                                                                \
            for (int q = 0; q < Lq; ++q)                                                           \
                for (int m = 0; m < M; ++m) {                                                      \
                    T *o = out + (((size_t)n * Lq + q) * M + m) * D;                               \
                    for (int d = 0; d < D; ++d) o[d] = 0;                                          \
                    for (int l = 0; l < L; ++l) {                                                  \
                        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];              \
                        const T *vl = value + ((size_t)n * S + lsi[l]) * M * D;                    \
                        for (int p = 0; p < P; ++p) {                                              \
                            const size_t wi = ((((size_t)n * Lq + q) * M + m) * L + l) * P + p;    \
                            const T lx = loc[2 * wi], ly = loc[2 * wi + 1], aw = w[wi];            \
                            const T h_im = ly * H - (T)0.5, w_im = lx * W - (T)0.5; /* cuh:290 */  \
                            if (!(h_im > -1 && w_im > -1 && h_im < H && w_im < W)) continue;       \
                            const int h0 = (int)floor((double)h_im), w0 = (int)floor((double)w_im);\
                            const int h1 = h0 + 1, w1 = w0 + 1;                                    \
                            const T lh = h_im - h0, lw = w_im - w0, hh = 1 - lh, hw = 1 - lw;      \
                            const T w1_ = hh * hw, w2_ = hh * lw, w3_ = lh * hw, w4_ = lh * lw;    \
                            for (int d = 0; d < D; ++d) {                                          \
                                T v1 = 0, v2 = 0, v3 = 0, v4 = 0;                                  \
                                if (h0 >= 0 && w0 >= 0) v1 = vl[((size_t)h0 * W + w0) * M * D + m * D + d];          \
                                if (h0 >= 0 && w1 <= W - 1) v2 = vl[((size_t)h0 * W + w1) * M * D + m * D + d];      \
                                if (h1 <= H - 1 && w0 >= 0) v3 = vl[((size_t)h1 * W + w0) * M * D + m * D + d];      \
                                if (h1 <= H - 1 && w1 <= W - 1) v4 = vl[((size_t)h1 * W + w1) * M * D + m * D + d];  \
                                o[d] += (w1_ * v1 + w2_ * v2 + w3_ * v3 + w4_ * v4) * aw;          \
                            }                                                                      \
                        }                                                                          \
                    }                                                                              \
                }                                                                                  \
    }
DEF_MSDA_FWD(orc_msda_forward_f32, float)
DEF_MSDA_FWD(orc_msda_forward_f64, double)

/* MSDeformAttn core, backward (ms_deform_im2col_cuda.cuh:92-164, formulas :119-163).
 * grad_value/grad_loc/grad_w must be zero-initialised by the caller. */
void orc_msda_backward_f32(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc,
                           const float *w, const float *grad_out, int N, int S, int M, int D, int L, int Lq,
                           int P, float *grad_value, float *grad_loc, float *grad_w)
{
    for (int n = 0; n < N; ++n)
        for (int q = 0; q < Lq; ++q)
            for (int m = 0; m < M; ++m) {
                const float *go = grad_out + (((size_t)n * Lq + q) * M + m) * D;
                for (int l = 0; l < L; ++l) {
                    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
                    const size_t base = ((size_t)n * S + lsi[l]) * M * D;
                    for (int p = 0; p < P; ++p) {
                        const size_t wi = ((((size_t)n * Lq + q) * M + m) * L + l) * P + p;
                        const float lx = loc[2 * wi], ly = loc[2 * wi + 1], aw = w[wi];
                        const float h_im = ly * H - 0.5f, w_im = lx * W - 0.5f;
                        if (!(h_im > -1 && w_im > -1 && h_im < H && w_im < W)) continue;
                        const int h0 = (int)floorf(h_im), w0 = (int)floorf(w_im), h1 = h0 + 1, w1 = w0 + 1;
                        const float lh = h_im - h0, lw = w_im - w0, hh = 1 - lh, hw = 1 - lw;
                        const float c1 = hh * hw, c2 = hh * lw, c3 = lh * hw, c4 = lh * lw;
                        double gw_acc = 0, gx = 0, gy = 0;
                        for (int d = 0; d < D; ++d) {
                            const float tg = go[d];
                            const float tgv = tg * aw;
                            float v1 = 0, v2 = 0, v3 = 0, v4 = 0;
                            float ghw = 0, gww = 0; /* d(val)/d(h_im), d(val)/d(w_im) */
                            if (h0 >= 0 && w0 >= 0) {
                                size_t ix = base + ((size_t)h0 * W + w0) * M * D + m * D + d;
                                v1 = value[ix]; ghw -= hw * v1; gww -= hh * v1; grad_value[ix] += c1 * tgv;
                            }
                            if (h0 >= 0 && w1 <= W - 1) {
                                size_t ix = base + ((size_t)h0 * W + w1) * M * D + m * D + d;
                                v2 = value[ix]; ghw -= lw * v2; gww += hh * v2; grad_value[ix] += c2 * tgv;
                            }
                            if (h1 <= H - 1 && w0 >= 0) {
                                size_t ix = base + ((size_t)h1 * W + w0) * M * D + m * D + d;
                                v3 = value[ix]; ghw += hw * v3; gww -= lh * v3; grad_value[ix] += c3 * tgv;
                            }
                            if (h1 <= H - 1 && w1 <= W - 1) {
                                size_t ix = base + ((size_t)h1 * W + w1) * M * D + m * D + d;
                                v4 = value[ix]; ghw += lw * v4; gww += lh * v4; grad_value[ix] += c4 * tgv;
                            }
                            const float val = c1 * v1 + c2 * v2 + c3 * v3 + c4 * v4;
                            gw_acc += (double)tg * val;
                            gx += (double)W * gww * tgv;
                            gy += (double)H * ghw * tgv;
                        }
                        grad_w[wi] = (float)gw_acc;
                        grad_loc[2 * wi] = (float)gx;
                        grad_loc[2 * wi + 1] = (float)gy;
                    }
                }
            }
}

/* ------------------------------------------------------------------------- *
 * point_sample == F.grid_sample(input, 2*coords-1, bilinear, zeros, align_corners=False)
 * model_training/mask2former_video/modeling/point_features.py:19-42.
 * input [R,C,H,W]; coords [R,P,2] (x,y in [0,1]); out [R,C,P].
 * ------------------------------------------------------------------------- */
void orc_point_sample_f32(const float *in, const float *coords, int R, int C, int H, int W, int P, float *out)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int r = 0; r < R; ++r)
        for (int p = 0; p < P; ++p) {
            const float u = coords[((size_t)r * P + p) * 2], v = coords[((size_t)r * P + p) * 2 + 1];
            /* grid = 2u-1 ; unnormalise (align_corners=False): ((g+1)*W-1)/2 */
            const float gx = 2.0f * u - 1.0f, gy = 2.0f * v - 1.0f;
            const float x = ((gx + 1.0f) * W - 1.0f) * 0.5f, y = ((gy + 1.0f) * H - 1.0f) * 0.5f;
            const int x0 = (int)floorf(x), y0 = (int)floorf(y), x1 = x0 + 1, y1 = y0 + 1;
            const float fx = x - x0, fy = y - y0;
            const float wnw = (1 - fx) * (1 - fy), wne = fx * (1 - fy), wsw = (1 - fx) * fy, wse = fx * fy;
            for (int c = 0; c < C; ++c) {
                const float *im = in + ((size_t)r * C + c) * H * W;
                float acc = 0;
                if (y0 >= 0 && y0 < H && x0 >= 0 && x0 < W) acc += im[(size_t)y0 * W + x0] * wnw;
                if (y0 >= 0 && y0 < H && x1 >= 0 && x1 < W) acc += im[(size_t)y0 * W + x1] * wne;
                if (y1 >= 0 && y1 < H && x0 >= 0 && x0 < W) acc += im[(size_t)y1 * W + x0] * wsw;
                if (y1 >= 0 && y1 < H && x1 >= 0 && x1 < W) acc += im[(size_t)y1 * W + x1] * wse;
                out[((size_t)r * C + c) * P + p] = acc;
            }
        }
}

/* same, input uint8 {0,1} masks (targets are stored as bytes by the product path) */
void orc_point_sample_u8(const uint8_t *in, const float *coords, int R, int C, int H, int W, int P, float *out)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int r = 0; r < R; ++r)
        for (int p = 0; p < P; ++p) {
            const float u = coords[((size_t)r * P + p) * 2], v = coords[((size_t)r * P + p) * 2 + 1];
            const float gx = 2.0f * u - 1.0f, gy = 2.0f * v - 1.0f;
            const float x = ((gx + 1.0f) * W - 1.0f) * 0.5f, y = ((gy + 1.0f) * H - 1.0f) * 0.5f;
            const int x0 = (int)floorf(x), y0 = (int)floorf(y), x1 = x0 + 1, y1 = y0 + 1;
            const float fx = x - x0, fy = y - y0;
            const float wnw = (1 - fx) * (1 - fy), wne = fx * (1 - fy), wsw = (1 - fx) * fy, wse = fx * fy;
            for (int c = 0; c < C; ++c) {
                const uint8_t *im = in + ((size_t)r * C + c) * H * W;
                float acc = 0;
                if (y0 >= 0 && y0 < H && x0 >= 0 && x0 < W) acc += im[(size_t)y0 * W + x0] * wnw;
                if (y0 >= 0 && y0 < H && x1 >= 0 && x1 < W) acc += im[(size_t)y0 * W + x1] * wne;
                if (y1 >= 0 && y1 < H && x0 >= 0 && x0 < W) acc += im[(size_t)y1 * W + x0] * wsw;
                if (y1 >= 0 && y1 < H && x1 >= 0 && x1 < W) acc += im[(size_t)y1 * W + x1] * wse;
                out[((size_t)r * C + c) * P + p] = acc;
            }
        }
}

/* ------------------------------------------------------------------------- *
 * F.interpolate(mode="bilinear", align_corners=False) on [R,H,W] -> [R,OH,OW]
 * call sites: video_mask2former_transformer_decoder.py:460 (attention-mask resize),
 * msdeformattn.py:349 (FPN upsample), kd_video_maskformer_model.py:466 (KD targets).
 * Source index rule (ATen area_pixel_compute_source_index): src = scale*(dst+0.5)-0.5,
 * clamped at 0; i1 = min(i0+1, in-1).
 * ------------------------------------------------------------------------- */
void orc_resize_bilinear_f32(const float *in, int R, int H, int W, int OH, int OW, float *out)
{
    const float sh = (float)H / OH, sw = (float)W / OW;
#pragma omp parallel for collapse(2) schedule(static)
    for (int r = 0; r < R; ++r)
        for (int oy = 0; oy < OH; ++oy) {
            float sy = sh * (oy + 0.5f) - 0.5f; if (sy < 0) sy = 0;
            const int y0 = (int)sy, y1 = y0 + (y0 < H - 1 ? 1 : 0);
            const float ly = sy - y0, hy = 1.0f - ly;
            for (int ox = 0; ox < OW; ++ox) {
                float sx = sw * (ox + 0.5f) - 0.5f; if (sx < 0) sx = 0;
                const int x0 = (int)sx, x1 = x0 + (x0 < W - 1 ? 1 : 0);
                const float lx = sx - x0, hx = 1.0f - lx;
                const float *im = in + (size_t)r * H * W;
                out[((size_t)r * OH + oy) * OW + ox] =
                    hy * (hx * im[(size_t)y0 * W + x0] + lx * im[(size_t)y0 * W + x1]) +
                    ly * (hx * im[(size_t)y1 * W + x0] + lx * im[(size_t)y1 * W + x1]);
            }
        }
}

/* ------------------------------------------------------------------------- *
 * Rectangular linear sum assignment, as called at matcher.py:289
 * (scipy.optimize.linear_sum_assignment; scipy is a third-party dependency that
 * is not under /root/reference -- scipy 1.15.3 in this image).  Restates the
 * published algorithm scipy implements: D. F. Crouse, "On implementing 2D
 * rectangular assignment algorithms", IEEE TAES 52(4), 2016 -- shortest
 * augmenting paths with dual variables, rows <= columns (transposed otherwise),
 * including its scan order so ties resolve identically.  Pinned in the tests by
 * calling scipy itself on the same matrices.
 * cost [nr,nc] row-major double. Output: a[min(nr,nc)], b[min(nr,nc)] with a ascending.
 * returns 0, or -1 infeasible / -2 invalid (nan/-inf) entries.
 * ------------------------------------------------------------------------- */
static int lsap_core(int nr, int nc, const double *cost, int64_t *col4row, int64_t *row4col)
{
    double *u = calloc(nr, sizeof(double)), *v = calloc(nc, sizeof(double));
    double *spc = malloc(nc * sizeof(double));
    int64_t *path = malloc(nc * sizeof(int64_t)), *remaining = malloc(nc * sizeof(int64_t));
    char *SR = malloc(nr), *SC = malloc(nc);
    for (int i = 0; i < nr; ++i) col4row[i] = -1;
    for (int j = 0; j < nc; ++j) { row4col[j] = -1; path[j] = -1; }
    int rc = 0;
    for (int cur = 0; cur < nr; ++cur) {
        double minVal = 0;
        int i = cur, num_remaining = nc;
        for (int it = 0; it < nc; ++it) remaining[it] = nc - it - 1;
        memset(SR, 0, nr); memset(SC, 0, nc);
        for (int j = 0; j < nc; ++j) spc[j] = INFINITY;
        int64_t sink = -1;
        while (sink == -1) {
            int index = -1;
            double lowest = INFINITY;
            SR[i] = 1;
            for (int it = 0; it < num_remaining; ++it) {
                const int64_t j = remaining[it];
                const double r = minVal + cost[(size_t)i * nc + j] - u[i] - v[j];
                if (r < spc[j]) { path[j] = i; spc[j] = r; }
                if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) { lowest = spc[j]; index = it; }
            }
            minVal = lowest;
            if (minVal == INFINITY) { rc = -1; goto done; }
            const int64_t j = remaining[index];
            if (row4col[j] == -1) sink = j; else i = (int)row4col[j];
            SC[j] = 1;
            remaining[index] = remaining[--num_remaining];
        }
        u[cur] += minVal;
        for (int r = 0; r < nr; ++r)
            if (SR[r] && r != cur) u[r] += minVal - spc[col4row[r]];
        for (int j = 0; j < nc; ++j)
            if (SC[j]) v[j] -= minVal - spc[j];
        int64_t j = sink;
        while (1) {
            const int64_t r = path[j];
            row4col[j] = r;
            const int64_t t = col4row[r];
            col4row[r] = j;
            j = t;
            if (r == cur) break;
        }
    }
done:
    free(u); free(v); free(spc); free(path); free(remaining); free(SR); free(SC);
    return rc;
}

int orc_lsap_f64(const double *cost, int nr, int nc, int64_t *a, int64_t *b)
{
    if (nr == 0 || nc == 0) return 0;
    for (size_t k = 0; k < (size_t)nr * nc; ++k)
        if (isnan(cost[k]) || cost[k] == -INFINITY) return -2;
    const int transpose = nc < nr;
    int R = nr, C = nc;
    double *ct = NULL;
    const double *c = cost;
    if (transpose) {
        ct = malloc((size_t)nr * nc * sizeof(double));
        for (int i = 0; i < nr; ++i)
            for (int j = 0; j < nc; ++j) ct[(size_t)j * nr + i] = cost[(size_t)i * nc + j];
        c = ct; R = nc; C = nr;
    }
    int64_t *col4row = malloc(R * sizeof(int64_t)), *row4col = malloc(C * sizeof(int64_t));
    int rc = lsap_core(R, C, c, col4row, row4col);
    if (rc == 0) {
        if (transpose) {
            /* original rows are the C side: emit pairs sorted by original row */
            int k = 0;
            for (int j = 0; j < C; ++j)
                if (row4col[j] != -1) { a[k] = j; b[k] = row4col[j]; ++k; }
        } else {
            for (int i = 0; i < R; ++i) { a[i] = i; b[i] = col4row[i]; }
        }
    }
    free(col4row); free(row4col); free(ct);
    return rc;
}

/* ------------------------------------------------------------------------- *
 * Keymask propagation (keymask_ident/cotracker_matching.py).
 * K3 pred_tracks_to_binary_masks(return_mask=False) :453-503 : round half-to-even,
 *    keep 0<=x<W, 0<=y<H, scatter 1.   tracks [T,Np,2]=(x,y) -> masks [T,H,W] u8
 * ------------------------------------------------------------------------- */
void orc_tracks_to_masks(const float *tracks, int T, int Np, int H, int W, uint8_t *masks)
{
    memset(masks, 0, (size_t)T * H * W);
    for (int t = 0; t < T; ++t)
        for (int p = 0; p < Np; ++p) {
            const long x = lrintf(tracks[((size_t)t * Np + p) * 2]);     /* FE_TONEAREST = half-to-even */
            const long y = lrintf(tracks[((size_t)t * Np + p) * 2 + 1]);
            if (x >= 0 && x < W && y >= 0 && y < H) masks[((size_t)t * H + y) * W + x] = 1;
        }
}

/* K4+K5 for one (frame, object): get_segmentation_mask :176-209, nearest resize :687-689
 * (src = floor(dst * in/out), computed in float like ATen nearest), and
 * compute_point_mask_intersection :640-662: #(points & obj) / #points  (0.0 when no points).
 * ids [Hi,Wi] int64 id map of the frame; pm [H,W] u8 point mask. */
double orc_point_mask_iou(const int64_t *ids, int Hi, int Wi, int64_t oid, const uint8_t *pm, int H, int W)
{
    const float sh = (float)Hi / H, sw = (float)Wi / W;
    long inter = 0, uni = 0;
    for (int y = 0; y < H; ++y) {
        int sy = (int)floorf(y * sh); if (sy > Hi - 1) sy = Hi - 1;
        for (int x = 0; x < W; ++x) {
            if (!pm[(size_t)y * W + x]) continue;
            int sx = (int)floorf(x * sw); if (sx > Wi - 1) sx = Wi - 1;
            ++uni;
            if (ids[(size_t)sy * Wi + sx] == oid) ++inter;
        }
    }
    return uni == 0 ? 0.0 : (double)inter / (double)uni;
}
