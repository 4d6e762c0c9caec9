// Data-side step before the hot path, on the device: the clip augmentation of the dataset mapper
// (model_training/mask2former_video/data_video/dataset_mapper.py:306-404 with the augmentation list of
// augmentation.py:116-168: crop, resize-shortest-edge, flip, brightness, contrast, rotation) as ONE resampling pass per
// frame, and the video copy-paste of the trainer (engine/train_loop.py:377-590).  HBM-bound byte kernels, no MFMA.
//
// The reference runs these on the host through detectron2 / fvcore / PIL / cv2 transforms, one image pass per transform;
// detectron2 is not in the reference tree, so the geometric + photometric chain here follows the published semantics of those
// transforms composed into a single inverse affine map per frame (parity unpinned for that part).  The copy-paste
// arithmetic in the reference is plain torch (F.interpolate + compositing) and is followed exactly.
#include "common.h"

namespace {

// One frame's augmentation: output pixel centre (x + 0.5, y + 0.5) -> source point (sx, sy) = A . (x + 0.5, y + 0.5, 1).
// The source is read inside the crop rectangle [cx, cx + cw) x [cy, cy + ch) only: points outside it give 0 (rotation
// fill), taps are clamped to it (edge replication, as a resize of the cropped image would).
struct AugFrame {
    float a11, a12, a13, a21, a22, a23;
    float cx, cy, cw, ch;
    float bright;      // RandomBrightness: img * w                      (1 = off)
    float contrast;    // RandomContrast:  (1 - w) * mean + w * img      (1 = off)
    float cmean;       // the mean the contrast blend pulls towards (filled by aug_crop_mean_kernel when < 0)
    float pad0, pad1, pad2;
};

__global__ __launch_bounds__(256) void aug_crop_mean_kernel(const uint8_t *__restrict__ src, int H0, int W0, AugFrame *__restrict__ fr)
{
    // mean over the 3 channels of the crop rectangle of frame blockIdx.x, times the brightness weight: what image.mean() is
    // when RandomContrast draws its transform (after crop, resize, flip and brightness)
    __shared__ double part[256];
    const int t = blockIdx.x;
    AugFrame f = fr[t];
    const int x0 = (int)f.cx, y0 = (int)f.cy, w = (int)f.cw, h = (int)f.ch;
    double s = 0.0;
    for (int c = 0; c < 3; ++c) {
        const uint8_t *p = src + ((long)t * 3 + c) * H0 * W0;
        for (long i = threadIdx.x; i < (long)w * h; i += 256) s += p[(long)(y0 + i / w) * W0 + x0 + i % w];
    }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0 && f.cmean < 0.f) fr[t].cmean = (float)(part[0] / (3.0 * w * h)) * f.bright;
}

__device__ __forceinline__ float trunc_u8(float v) { return truncf(fminf(fmaxf(v, 0.f), 255.f)); }

__global__ __launch_bounds__(256) void aug_warp_frames_kernel(const uint8_t *__restrict__ src, int H0, int W0, const AugFrame *__restrict__ fr,
                                                              int H1, int W1, uint8_t *__restrict__ out)
{
    const int t = blockIdx.z, y = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W1) return;
    const AugFrame f = fr[t];
    const float px = x + 0.5f, py = y + 0.5f;
    const float sx = f.a11 * px + f.a12 * py + f.a13, sy = f.a21 * px + f.a22 * py + f.a23;
    uint8_t r[3] = {0, 0, 0};
    if (sx >= f.cx && sy >= f.cy && sx < f.cx + f.cw && sy < f.cy + f.ch) {
        const float fx = sx - 0.5f, fy = sy - 0.5f;
        const float x0f = floorf(fx), y0f = floorf(fy);
        const float lx = fx - x0f, ly = fy - y0f;
        const int xl = (int)f.cx, xh = (int)(f.cx + f.cw) - 1, yl = (int)f.cy, yh = (int)(f.cy + f.ch) - 1;
        const int x0 = min(max((int)x0f, xl), xh), x1 = min(max((int)x0f + 1, xl), xh);
        const int y0 = min(max((int)y0f, yl), yh), y1 = min(max((int)y0f + 1, yl), yh);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const uint8_t *p = src + ((long)t * 3 + c) * H0 * W0;
            const float v = (1.f - ly) * ((1.f - lx) * p[(long)y0 * W0 + x0] + lx * p[(long)y0 * W0 + x1]) +
                            ly * ((1.f - lx) * p[(long)y1 * W0 + x0] + lx * p[(long)y1 * W0 + x1]);
            float q = rintf(v);                                               // the resampled uint8 image
            if (f.bright != 1.f) q = trunc_u8(f.bright * q);                  // BlendTransform: clip, astype(uint8)
            if (f.contrast != 1.f) q = trunc_u8((1.f - f.contrast) * f.cmean + f.contrast * q);
            r[c] = (uint8_t)q;
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) out[(((long)t * 3 + c) * H1 + y) * W1 + x] = r[c];
}

// masks [N][T][H0][W0] (0 / non-0) -> [N][T][H1][W1] (0 / 1): nearest source pixel under the same map (apply_segmentation)
__global__ __launch_bounds__(256) void aug_warp_masks_kernel(const uint8_t *__restrict__ src, int T, int H0, int W0,
                                                             const AugFrame *__restrict__ fr, int H1, int W1, uint8_t *__restrict__ out)
{
    const int nt = blockIdx.z, t = nt % T, y = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W1) return;
    const AugFrame f = fr[t];
    const float px = x + 0.5f, py = y + 0.5f;
    const float sx = f.a11 * px + f.a12 * py + f.a13, sy = f.a21 * px + f.a22 * py + f.a23;
    uint8_t r = 0;
    if (sx >= f.cx && sy >= f.cy && sx < f.cx + f.cw && sy < f.cy + f.ch)
        r = src[((long)nt * H0 + (int)floorf(sy)) * W0 + (int)floorf(sx)] != 0;
    out[((long)nt * H1 + y) * W1 + x] = r;
}

// ---- video copy-paste (engine/train_loop.py:441-560) ---------------------------------------------------------------
// Per target frame f: the K copied source masks and the source frame are resized to (h_new, w_new) with
// F.interpolate(bilinear, align_corners=False) -- image `.byte()` (truncation), masks `.bool()` (non-zero) -- and placed at
// (h_shift, w_shift) of an empty canvas (:470-500); alpha = any kept copied mask (:540); the composite takes the source
// pixel under alpha (:546); every target mask loses alpha (:541); the copied masks become new instances (:548).
struct PasteFrame { int h_new, w_new, h_shift, w_shift; };

__device__ __forceinline__ void bilin_taps(int dst, int in_size, int out_size, int &i0, int &i1, float &l1)
{
    float s = ((float)in_size / out_size) * (dst + 0.5f) - 0.5f;         // ATen area_pixel_compute_source_index, align_corners=False
    if (s < 0.f) s = 0.f;
    i0 = (int)s;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = s - i0;
}

__global__ __launch_bounds__(256) void copy_paste_kernel(const uint8_t *__restrict__ tgt_frames, const uint8_t *__restrict__ tgt_masks, int N, int T,
                                                         int H, int W, const uint8_t *__restrict__ src_frame, const uint8_t *__restrict__ src_masks,
                                                         int K, int Hs, int Ws, const PasteFrame *__restrict__ pf, const uint8_t *__restrict__ keep,
                                                         uint8_t *__restrict__ out_frames, uint8_t *__restrict__ out_masks)
{
    const int t = blockIdx.z, y = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    const PasteFrame p = pf[t];
    const int yy = y - p.h_shift, xx = x - p.w_shift;
    const bool in = yy >= 0 && xx >= 0 && yy < p.h_new && xx < p.w_new;
    int y0 = 0, y1 = 0, x0 = 0, x1 = 0;
    float ly = 0.f, lx = 0.f;
    if (in) {
        bilin_taps(yy, Hs, p.h_new, y0, y1, ly);
        bilin_taps(xx, Ws, p.w_new, x0, x1, lx);
    }
    bool alpha = false;
    for (int k = 0; k < K; ++k) {
        bool m = false;
        if (in) {
            const uint8_t *s = src_masks + (long)k * Hs * Ws;
            const float v = (1.f - ly) * ((1.f - lx) * (s[(long)y0 * Ws + x0] != 0) + lx * (s[(long)y0 * Ws + x1] != 0)) +
                            ly * ((1.f - lx) * (s[(long)y1 * Ws + x0] != 0) + lx * (s[(long)y1 * Ws + x1] != 0));
            m = v != 0.f;                                                    // .bool()
        }
        out_masks[(((long)(N + k) * T + t) * H + y) * W + x] = m && keep[k];
        alpha = alpha || (m && keep[k]);
    }
    for (int n = 0; n < N; ++n) {
        const long i = (((long)n * T + t) * H + y) * W + x;
        out_masks[i] = alpha ? 0 : (tgt_masks[i] != 0);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const long i = (((long)t * 3 + c) * H + y) * W + x;
        uint8_t v = tgt_frames[i];
        if (alpha) {
            const uint8_t *s = src_frame + (long)c * Hs * Ws;
            const float f = (1.f - ly) * ((1.f - lx) * s[(long)y0 * Ws + x0] + lx * s[(long)y0 * Ws + x1]) +
                            ly * ((1.f - lx) * s[(long)y1 * Ws + x0] + lx * s[(long)y1 * Ws + x1]);
            v = (uint8_t)f;                                                  // .byte(): truncation
        }
        out_frames[i] = v;
    }
}

// counts[k][n] = #(pasted copy k of frame 0 AND target mask n of frame 0), area[n] = #target mask n: the "ioy" test of :515-527
__global__ __launch_bounds__(256) void copy_paste_overlap_kernel(const uint8_t *__restrict__ tgt_masks, int N, int T, int H, int W,
                                                                 const uint8_t *__restrict__ src_masks, int K, int Hs, int Ws, PasteFrame p,
                                                                 int *__restrict__ counts, int *__restrict__ area)
{
    const int k = blockIdx.x, n = blockIdx.y;
    const uint8_t *tm = tgt_masks + ((long)n * T) * H * W;                   // frame 0
    const uint8_t *s = src_masks + (long)k * Hs * Ws;
    int c = 0, a = 0;
    for (long i = threadIdx.x; i < (long)H * W; i += 256) {
        const int y = (int)(i / W), x = (int)(i - (long)y * W);
        const bool tv = tm[i] != 0;
        a += tv;
        const int yy = y - p.h_shift, xx = x - p.w_shift;
        if (tv && yy >= 0 && xx >= 0 && yy < p.h_new && xx < p.w_new) {
            int y0, y1, x0, x1;
            float ly, lx;
            bilin_taps(yy, Hs, p.h_new, y0, y1, ly);
            bilin_taps(xx, Ws, p.w_new, x0, x1, lx);
            const float v = (1.f - ly) * ((1.f - lx) * (s[(long)y0 * Ws + x0] != 0) + lx * (s[(long)y0 * Ws + x1] != 0)) +
                            ly * ((1.f - lx) * (s[(long)y1 * Ws + x0] != 0) + lx * (s[(long)y1 * Ws + x1] != 0));
            c += v != 0.f;
        }
    }
    __shared__ int sc[256], sa[256];
    sc[threadIdx.x] = c; sa[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { sc[threadIdx.x] += sc[threadIdx.x + o]; sa[threadIdx.x] += sa[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { counts[k * N + n] = sc[0]; if (k == 0) area[n] = sa[0]; }
}

// One frame of the reference's loop as it is written (engine/train_loop.py:445-570), including what looks like an accident but is
// the shipped behaviour: `copied_instances.gt_masks` is REASSIGNED to the pasted canvas at the end of every frame (:512-514), so
// frame f + 1 deep-copies, resizes and shifts the canvas of frame f -- the copied masks are transformed cumulatively -- while the
// image patch is resized from the source frame afresh each time (:470).  cur_masks [K][Hc][Wc] is therefore the source masks at
// the first frame and the previous frame's canvas afterwards.  Outputs: the new canvas [K][H][W] (0 / 1), the composite frame,
// the target masks minus alpha, and the integers the host's decisions need: inter[k][n] = |canvas_k AND target_n| and
// tarea[n] = |target_n| (both before alpha is removed: the "ioy" matrix of :517-521), alive[n] = |target_n AND NOT alpha| (:543).
__global__ __launch_bounds__(256) void copy_paste_frame_kernel(const uint8_t *__restrict__ src_frame, int Hs, int Ws,
                                                               const uint8_t *__restrict__ cur_masks, int K, int Hc, int Wc,
                                                               const uint8_t *__restrict__ tgt_frame, const uint8_t *__restrict__ tgt_masks, int N,
                                                               int H, int W, PasteFrame p, uint8_t *__restrict__ canvas,
                                                               uint8_t *__restrict__ out_frame, uint8_t *__restrict__ out_tgt,
                                                               int *__restrict__ inter, int *__restrict__ tarea, int *__restrict__ alive)
{
    const int y = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
    const bool live = x < W;
    const int yy = y - p.h_shift, xx = x - p.w_shift;
    const bool in = live && yy >= 0 && xx >= 0 && yy < p.h_new && xx < p.w_new;
    int my0 = 0, my1 = 0, mx0 = 0, mx1 = 0, iy0 = 0, iy1 = 0, ix0 = 0, ix1 = 0;
    float mly = 0.f, mlx = 0.f, ily = 0.f, ilx = 0.f;
    if (in) {
        bilin_taps(yy, Hc, p.h_new, my0, my1, mly); bilin_taps(xx, Wc, p.w_new, mx0, mx1, mlx);      // masks: from the current canvas
        bilin_taps(yy, Hs, p.h_new, iy0, iy1, ily); bilin_taps(xx, Ws, p.w_new, ix0, ix1, ilx);      // image: from the source frame
    }
    unsigned long long mbits = 0ull;                                          // bit k: copy k covers this pixel (K <= 64)
    for (int k = 0; k < K; ++k) {
        bool m = false;
        if (in) {
            const uint8_t *c = cur_masks + (long)k * Hc * Wc;
            const float v = (1.f - mly) * ((1.f - mlx) * (c[(long)my0 * Wc + mx0] != 0) + mlx * (c[(long)my0 * Wc + mx1] != 0)) +
                            mly * ((1.f - mlx) * (c[(long)my1 * Wc + mx0] != 0) + mlx * (c[(long)my1 * Wc + mx1] != 0));
            m = v != 0.f;                                                     // .bool()
        }
        if (live) canvas[((long)k * H + y) * W + x] = m;
        mbits |= (unsigned long long)m << k;
    }
    const bool alpha = mbits != 0ull;
    const int lane = threadIdx.x & 63;
    for (int n = 0; n < N; ++n) {
        const bool tv = live && tgt_masks[((long)n * H + y) * W + x] != 0;
        if (live) out_tgt[((long)n * H + y) * W + x] = tv && !alpha;
        const unsigned long long bt = __ballot(tv), ba = __ballot(tv && !alpha);
        if (lane == 0) {
            if (bt) atomicAdd(tarea + n, (int)__popcll(bt));
            if (ba) atomicAdd(alive + n, (int)__popcll(ba));
        }
        if (bt) {                                                             // wave-uniform
            for (int k = 0; k < K; ++k) {
                const unsigned long long bi = __ballot(tv && ((mbits >> k) & 1ull));
                if (lane == 0 && bi) atomicAdd(inter + k * N + n, (int)__popcll(bi));
            }
        }
    }
    if (!live) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const long i = ((long)c * H + y) * W + x;
        uint8_t v = tgt_frame[i];
        if (alpha) {
            const uint8_t *sp = src_frame + (long)c * Hs * Ws;
            const float f = (1.f - ily) * ((1.f - ilx) * sp[(long)iy0 * Ws + ix0] + ilx * sp[(long)iy0 * Ws + ix1]) +
                            ily * ((1.f - ilx) * sp[(long)iy1 * Ws + ix0] + ilx * sp[(long)iy1 * Ws + ix1]);
            v = (uint8_t)f;                                                   // .byte(): truncation
        }
        out_frame[i] = v;
    }
}

// ---- sparse-mask densification (engine/train_loop.py:30-156): every output plane of a clip in ONE launch ------------------------------
// plan row j = {address of the source plane (bool / u8 [H, W]), dx, dy}: out[j][y][x] = src[y + dy][x + dx] inside the frame, else 0.
// Kept instances are rows with dx = dy = 0 (the reference concatenates them in front of the synthesised ones), filled instances read
// the plane of the frame the id was last seen in.  16 output bytes per thread; a shifted row is read with unaligned byte loads only on
// the planes that are shifted (dx != 0).
struct ShiftPlan { unsigned long long src; int dx, dy; };

__global__ __launch_bounds__(256) void shift_planes_kernel(const ShiftPlan *__restrict__ plan, int H, int W, uint8_t *__restrict__ out)
{
    const ShiftPlan pl = plan[blockIdx.y];
    const uint8_t *src = reinterpret_cast<const uint8_t *>(pl.src);
    uint8_t *dst = out + (size_t)blockIdx.y * H * W;
    const long npix = (long)H * W;
    for (long i0 = ((long)blockIdx.x * 256 + threadIdx.x) * 16; i0 < npix; i0 += (long)gridDim.x * 256 * 16) {
        const bool fast = pl.dx == 0 && pl.dy == 0 && i0 + 16 <= npix && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
        if (fast) {
            *reinterpret_cast<uint4 *>(dst + i0) = *reinterpret_cast<const uint4 *>(src + i0);
            continue;
        }
        int y = (int)(i0 / W), x = (int)(i0 - (long)y * W);
        for (int e = 0; e < 16 && i0 + e < npix; ++e) {
            const int sy = y + pl.dy, sx = x + pl.dx;
            dst[i0 + e] = (sy >= 0 && sy < H && sx >= 0 && sx < W) ? (src[(long)sy * W + sx] != 0) : 0;
            if (++x == W) { x = 0; ++y; }
        }
    }
}

}  // namespace

extern "C" {

int s2d_copy_paste_frame_u8(const uint8_t *src_frame, int Hs, int Ws, const uint8_t *cur_masks, int K, int Hc, int Wc,
                            const uint8_t *tgt_frame, const uint8_t *tgt_masks, int N, int H, int W, int h_new, int w_new, int h_shift,
                            int w_shift, uint8_t *canvas, uint8_t *out_frame, uint8_t *out_tgt, int *inter, int *tarea, int *alive,
                            hipStream_t stream)
{
    if (K <= 0 || K > 64 || N < 0 || H <= 0 || W <= 0 || H > 65535 || h_new <= 0 || w_new <= 0) return S2D_ERR_ARG;
    if (N > 0) {
        if (s2d_zero_async(inter, sizeof(int) * (size_t)K * N, stream) != S2D_OK || s2d_zero_async(tarea, sizeof(int) * (size_t)N, stream) != S2D_OK ||
            s2d_zero_async(alive, sizeof(int) * (size_t)N, stream) != S2D_OK)
            return S2D_ERR_LAUNCH;
    }
    PasteFrame p{h_new, w_new, h_shift, w_shift};
    hipLaunchKernelGGL(copy_paste_frame_kernel, dim3(cdiv(W, 256), H), dim3(256), 0, stream, src_frame, Hs, Ws, cur_masks, K, Hc, Wc, tgt_frame,
                       tgt_masks, N, H, W, p, canvas, out_frame, out_tgt, inter, tarea, alive);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

int s2d_aug_warp_frames_u8(const uint8_t *frames, int T, int H0, int W0, void *aug_frames_dev, int H1, int W1, uint8_t *out,
                           hipStream_t stream)
{
    if (T <= 0 || H0 <= 0 || W0 <= 0 || H1 <= 0 || W1 <= 0 || H1 > 65535) return S2D_ERR_ARG;
    AugFrame *fr = reinterpret_cast<AugFrame *>(aug_frames_dev);
    hipLaunchKernelGGL(aug_crop_mean_kernel, dim3(T), dim3(256), 0, stream, frames, H0, W0, fr);
    hipLaunchKernelGGL(aug_warp_frames_kernel, dim3(cdiv(W1, 256), H1, T), dim3(256), 0, stream, frames, H0, W0, fr, H1, W1, out);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

int s2d_aug_warp_masks_u8(const uint8_t *masks, int N, int T, int H0, int W0, const void *aug_frames_dev, int H1, int W1, uint8_t *out,
                          hipStream_t stream)
{
    if (N < 0 || T <= 0 || H0 <= 0 || W0 <= 0 || H1 <= 0 || W1 <= 0 || H1 > 65535 || (long)N * T > 65535) return S2D_ERR_ARG;
    if (N == 0) return S2D_OK;
    hipLaunchKernelGGL(aug_warp_masks_kernel, dim3(cdiv(W1, 256), H1, N * T), dim3(256), 0, stream, masks, T, H0, W0,
                       reinterpret_cast<const AugFrame *>(aug_frames_dev), H1, W1, out);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

int s2d_copy_paste_overlap(const uint8_t *tgt_masks, int N, int T, int H, int W, const uint8_t *src_masks, int K, int Hs, int Ws, int h_new,
                           int w_new, int h_shift, int w_shift, int *counts, int *area, hipStream_t stream)
{
    if (N <= 0 || K <= 0 || T <= 0 || h_new <= 0 || w_new <= 0) return S2D_ERR_ARG;
    PasteFrame p{h_new, w_new, h_shift, w_shift};
    hipLaunchKernelGGL(copy_paste_overlap_kernel, dim3(K, N), dim3(256), 0, stream, tgt_masks, N, T, H, W, src_masks, K, Hs, Ws, p, counts, area);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

int s2d_copy_paste_u8(const uint8_t *tgt_frames, const uint8_t *tgt_masks, int N, int T, int H, int W, const uint8_t *src_frame,
                      const uint8_t *src_masks, int K, int Hs, int Ws, const int *paste_frames_dev, const uint8_t *keep_dev,
                      uint8_t *out_frames, uint8_t *out_masks, hipStream_t stream)
{
    if (N < 0 || K <= 0 || T <= 0 || H <= 0 || W <= 0 || H > 65535) return S2D_ERR_ARG;
    hipLaunchKernelGGL(copy_paste_kernel, dim3(cdiv(W, 256), H, T), dim3(256), 0, stream, tgt_frames, tgt_masks, N, T, H, W, src_frame, src_masks, K,
                       Hs, Ws, reinterpret_cast<const PasteFrame *>(paste_frames_dev), keep_dev, out_frames, out_masks);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

int s2d_shift_planes_u8(const void *plan_dev, int n_planes, int H, int W, uint8_t *out, hipStream_t stream)
{
    if (n_planes < 0 || H < 0 || W < 0 || (n_planes && (!plan_dev || !out)) || n_planes > 65535) return S2D_ERR_ARG;
    if (n_planes == 0 || (long)H * W == 0) return S2D_OK;
    const long npix = (long)H * W;
    int gx = cdiv(npix, 256 * 16);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(shift_planes_kernel, dim3(gx, n_planes), dim3(256), 0, stream, reinterpret_cast<const ShiftPlan *>(plan_dev), H, W, out);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

}  // extern "C"
