// Keymask discovery, device side (bandwidth-bound; coalesced HBM, wavefront reductions, no MFMA).
// Reference: /root/reference/keymask_ident/cotracker_matching.py and cotracker_occlusions.py.
//
//  K2 visibility curve      cotracker_occlusions.py:359       mean over points of pred_visibility
//  K3 tracks -> point masks cotracker_matching.py:453-503     round-half-even, bounds filter, scatter 1
//  K4 id == oid, nearest    :176-209, :687-689                never materialised: the id map is read through the
//                                                             nearest-resize index map at the point pixels only
//  K5 point/mask "IoU"      :640-662                          #(points & obj) / #points  -> integer counts per (frame, id)
//  K6 match loop            :665-719                          one launch for all frames x all object ids of a tracked mask
//  K1 local correlation     co-tracker (third party, not in the reference tree): self-defined restatement of its local
//                           4-D correlation: per (frame, track point) the (2r+1)^2 bilinear-sampled neighbourhood
//                           features dotted with the track's (2r+1)^2 support features.  PARITY UNPINNED.
//
// The reference runs K4+K5 as O(frames x objects) full-frame launches with two .item() syncs per pair; here the
// point mask is visited once per frame and a per-frame histogram over object ids yields every pair's count at once.
#include "common.h"

namespace {

__global__ void scatter_tracks_kernel(const float *__restrict__ tracks, int T, int Np, int H, int W, uint8_t *__restrict__ masks)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)T * Np) return;
    const int t = (int)(i / Np);
    const long x = (long)rintf(tracks[2 * i]);       // torch.round: half to even, then .long()
    const long y = (long)rintf(tracks[2 * i + 1]);
    if (x >= 0 && x < W && y >= 0 && y < H) masks[((long)t * H + y) * W + x] = 1;
}

// counts[t][id] = #{point pixels of frame t whose nearest-resized id-map value is id}; total[t] = #point pixels
__global__ __launch_bounds__(256) void point_id_hist_kernel(const uint8_t *__restrict__ pm, const int64_t *__restrict__ ids,
                                                            int H, int W, int Hi, int Wi, int max_id,
                                                            int *__restrict__ counts, int *__restrict__ total)
{
    extern __shared__ int hist[];   // [max_id + 2]: ids 0..max_id, last = total
    const int t = blockIdx.y;
    for (int i = threadIdx.x; i < max_id + 2; i += 256) hist[i] = 0;
    __syncthreads();
    const float sh = (float)Hi / H, sw = (float)Wi / W;      // F.interpolate(mode="nearest"): src = floor(dst * in/out)
    const long per = ((long)H * W + gridDim.x - 1) / gridDim.x;
    const long i0 = (long)blockIdx.x * per, i1 = min((long)H * W, i0 + per);
    const uint8_t *p = pm + (long)t * H * W;
    const int64_t *idf = ids + (long)t * Hi * Wi;
    for (long i = i0 + threadIdx.x; i < i1; i += 256) {
        if (!p[i]) continue;
        const int y = (int)(i / W), x = (int)(i % W);
        int sy = (int)floorf(y * sh), sx = (int)floorf(x * sw);
        sy = min(sy, Hi - 1); sx = min(sx, Wi - 1);
        const int64_t id = idf[(long)sy * Wi + sx];
        atomicAdd(&hist[max_id + 1], 1);
        if (id >= 0 && id <= max_id) atomicAdd(&hist[(int)id], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i <= max_id; i += 256)
        if (hist[i]) atomicAdd(&counts[(long)t * (max_id + 1) + i], hist[i]);
    if (threadIdx.x == 0 && hist[max_id + 1]) atomicAdd(&total[t], hist[max_id + 1]);
}

// presence[t][id] = id occurs in frame t of the id map (torch.unique of :680)
__global__ __launch_bounds__(256) void id_presence_kernel(const int64_t *__restrict__ ids, long HW, int max_id,
                                                          uint8_t *__restrict__ presence)
{
    extern __shared__ int seen[];
    const int t = blockIdx.y;
    for (int i = threadIdx.x; i <= max_id; i += 256) seen[i] = 0;
    __syncthreads();
    const long per = (HW + gridDim.x - 1) / gridDim.x;
    const long i0 = (long)blockIdx.x * per, i1 = min(HW, i0 + per);
    const int64_t *p = ids + (long)t * HW;
    for (long i = i0 + threadIdx.x; i < i1; i += 256) {
        const int64_t id = p[i];
        if (id >= 0 && id <= max_id && !seen[(int)id]) seen[(int)id] = 1;   // benign race: all writers store 1
    }
    __syncthreads();
    for (int i = threadIdx.x; i <= max_id; i += 256)
        if (seen[i]) presence[(long)t * (max_id + 1) + i] = 1;
}

// vis [T][Np] (bytes, nonzero = visible) -> curve[t] = mean  (wavefront reduction)
__global__ __launch_bounds__(256) void visibility_kernel(const uint8_t *__restrict__ vis, int Np, float *__restrict__ curve)
{
    __shared__ int red[4];
    const int t = blockIdx.x;
    int c = 0;
    for (int i = threadIdx.x; i < Np; i += 256) c += vis[(long)t * Np + i] != 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) curve[t] = (float)(red[0] + red[1] + red[2] + red[3]) / (float)Np;
}

// K1: corr[t][n][i][j] = sum_c nb[t][n][i][c] * support[n][j][c], nb = bilinear samples of fmap[t] (NHWC) on the
// (2r+1)^2 integer-offset grid around coords[t][n] (zero padding outside).  One workgroup per (n, t).
// VALU kernel (north_star: no MFMA in the keymask set).  Both S x C operands are staged in LDS (16-B aligned rows, the
// neighbourhood sampled with 16-B tap loads along c); the S x S products are register-tiled: a thread (ti, tj) of a TB x TB grid
// owns outputs (ti + TB a, tj + TB b), a, b < 4, and per 4 channels reads 4 + 4 LDS vectors for 64 multiply-adds -- the untiled
// form read two LDS words per multiply-add and ran at 0.59 TB/s of the kernel's algorithmic bytes.  The sum over c runs in
// channel order in one accumulator per output, with fused multiply-adds.
template <int TB>
__global__ __launch_bounds__(256) void local_corr_kernel(const float *__restrict__ fmap, const float *__restrict__ coords,
                                                         const float *__restrict__ support, int T, int Np, int H, int W,
                                                         int C, int r, float *__restrict__ corr)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int D = 2 * r + 1, S = D * D;
    const int SR = S + 1;                     // staged rows: the S real ones and one row of zeros that every padded (i, j) >= S reads
    const int ldc = C + 4;                    // 16-B aligned rows; consecutive rows 4 banks apart
    float *nb = sm, *sp = sm + SR * ldc;
    const int n = blockIdx.x, t = blockIdx.y;
    const float cx = coords[((long)t * Np + n) * 2], cy = coords[((long)t * Np + n) * 2 + 1];
    const float *fm = fmap + (long)t * H * W * C;
    const int C4 = C >> 2;
    for (int e = threadIdx.x; e < SR * C4; e += 256) {
        const int i = e / C4, c = (e - i * C4) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f}, w = v;
        if (i < S) {
            const int dy = i / D - r, dx = i % D - r;
            const float x = cx + dx, y = cy + dy;
            const int x0 = (int)floorf(x), y0 = (int)floorf(y);
            const float fx = x - x0, fy = y - y0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int xx = x0 + (k & 1), yy = y0 + (k >> 1);
                if (xx >= 0 && xx < W && yy >= 0 && yy < H)
                    v += *reinterpret_cast<const f32x4 *>(fm + ((long)yy * W + xx) * C + c) * ((k & 1 ? fx : 1.f - fx) * (k >> 1 ? fy : 1.f - fy));
            }
            w = *reinterpret_cast<const f32x4 *>(support + ((long)n * S + i) * C + c);
        }
        *reinterpret_cast<f32x4 *>(nb + i * ldc + c) = v;
        *reinterpret_cast<f32x4 *>(sp + i * ldc + c) = w;
    }
    __syncthreads();
    if (threadIdx.x >= TB * TB) return;
    const int ti = threadIdx.x / TB, tj = threadIdx.x - ti * TB;
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
    int ra[4], rb[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        ra[a] = min(ti + TB * a, S) * ldc;
        rb[a] = min(tj + TB * a, S) * ldc;
    }
    for (int c = 0; c < C; c += 4) {
        f32x4 va[4], vb[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            va[a] = *reinterpret_cast<const f32x4 *>(nb + ra[a] + c);
            vb[a] = *reinterpret_cast<const f32x4 *>(sp + rb[a] + c);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_fmaf(va[a][k], vb[b][k], acc[a][b]);   // one v_fma per product (channel order kept)
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int i = ti + TB * a;
        if (i >= S) continue;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int j = tj + TB * b;
            if (j < S) corr[(((long)t * Np + n) * S + i) * S + j] = acc[a][b];
        }
    }
}

// ---- colour masks -> id maps (load_masks, cotracker_matching.py:22-84) ------------------------------------------------
// Per frame: the distinct non-black colours, sorted lexicographically by (R,G,B) (= numerically as R<<16|G<<8|B), get ids
// 1..n; black is 0.  The reference loops over the colours with a full-frame compare each (O(#colours * H * W) numpy); here:
// (1) distinct keys of a frame go into a small open-addressing table -- a lane skips the atomic when its left neighbour in
// the wave holds the same key, which is almost always (masks are piecewise constant); (2) one workgroup per frame sorts
// the table (bitonic, LDS); (3) every pixel looks its key up by binary search in the sorted list held in LDS.
constexpr int IDSLOTS = 8192, IDMAX = 4096;           // table slots / most distinct colours per frame
constexpr unsigned int IDEMPTY = 0u;                  // black is never inserted, so 0 marks a free slot (table zero-filled)

__device__ __forceinline__ unsigned int rgb_key(const uint8_t *__restrict__ px)
{
    return ((unsigned int)px[0] << 16) | ((unsigned int)px[1] << 8) | (unsigned int)px[2];
}

__global__ __launch_bounds__(256) void idmap_collect_kernel(const uint8_t *__restrict__ rgb, long HW, unsigned int *__restrict__ table,
                                                            int *__restrict__ overflow)
{
    const int t = blockIdx.y;
    unsigned int *tab = table + (long)t * IDSLOTS;
    const uint8_t *fr = rgb + (long)t * HW * 3;
    const long per = (HW + gridDim.x - 1) / gridDim.x;
    const long p0 = (long)blockIdx.x * per, p1 = p0 + per < HW ? p0 + per : HW;
    const int lane = threadIdx.x & 63;
    for (long base = p0; base < p1; base += 256) {          // uniform trip count per block: shuffles need whole waves
        const long i = base + threadIdx.x;
        const unsigned int key = i < p1 ? rgb_key(fr + i * 3) : 0u;
        const unsigned int left = __shfl_up(key, 1, 64);
        if (key != 0u && (lane == 0 || left != key)) {
            unsigned int h = (key * 2654435761u) >> 19;     // 13 bits
            int probes = 0;
            while (true) {
                const unsigned int old = atomicCAS(&tab[h], IDEMPTY, key);
                if (old == IDEMPTY || old == key) break;
                h = (h + 1) & (IDSLOTS - 1);
                if (++probes >= IDSLOTS) { atomicExch(overflow, 1); break; }
            }
        }
    }
}

__global__ __launch_bounds__(1024) void idmap_sort_kernel(unsigned int *__restrict__ table, int *__restrict__ n_ids, int *__restrict__ overflow)
{
    __shared__ unsigned int s[IDSLOTS];
    __shared__ int cnt;
    unsigned int *tab = table + (long)blockIdx.x * IDSLOTS;
    if (threadIdx.x == 0) cnt = 0;
    for (int i = threadIdx.x; i < IDSLOTS; i += 1024) s[i] = tab[i];
    __syncthreads();
    for (int k = 2; k <= IDSLOTS; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < IDSLOTS; i += 1024) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned int a = s[i], b = s[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { s[i] = b; s[ixj] = a; }
                }
            }
            __syncthreads();
        }
    int mine = 0;
    for (int i = threadIdx.x; i < IDSLOTS; i += 1024) { tab[i] = s[i]; mine += s[i] != IDEMPTY; }
    atomicAdd(&cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0) {
        n_ids[blockIdx.x] = cnt;
        if (cnt > IDMAX) atomicExch(overflow, 1);
    }
}

__global__ __launch_bounds__(256) void idmap_apply_kernel(const uint8_t *__restrict__ rgb, long HW, const unsigned int *__restrict__ table,
                                                          const int *__restrict__ n_ids, int64_t *__restrict__ ids)
{
    __shared__ unsigned int s[IDMAX];
    const int t = blockIdx.y;
    const int n = min(n_ids[t], IDMAX);
    for (int i = threadIdx.x; i < n; i += 256) s[i] = table[(long)t * IDSLOTS + (IDSLOTS - n) + i];   // empties (0) sort first
    __syncthreads();
    const uint8_t *fr = rgb + (long)t * HW * 3;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < HW; i += (long)gridDim.x * 256) {
        const unsigned int key = rgb_key(fr + i * 3);
        int lo = 0, hi = n;                              // first index with s[idx] >= key
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s[mid] < key) lo = mid + 1; else hi = mid;
        }
        ids[(long)t * HW + i] = key == 0u ? 0 : (int64_t)(lo + 1);
    }
}


// K4 for a whole list of (frame, object) candidates at once (get_segmentation_mask, keymask_utils.py:37-67 /
// cotracker_matching.py:176-209): out[k] = (idmap[frame_k] == obj_k) * 255, obj_k == -1 selects every non-background id.
// One launch and one device-to-host copy per video instead of one of each per written PNG.
__global__ __launch_bounds__(256) void select_masks_kernel(const int64_t *__restrict__ idmap, long HW, const int *__restrict__ frames,
                                                           const int *__restrict__ objs, uint8_t *__restrict__ out)
{
    const int k = blockIdx.y;
    const int64_t *src = idmap + (long)frames[k] * HW;
    const long o = objs[k];
    uint8_t *dst = out + (long)k * HW;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8; i < HW; i += (long)gridDim.x * 256 * 8) {
        if (i + 8 <= HW && (HW & 7) == 0) {
            uint64_t w = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int64_t v = src[i + j];
                w |= (uint64_t)((o < 0 ? v != 0 : v == o) ? 0xFFu : 0u) << (8 * j);
            }
            *reinterpret_cast<uint64_t *>(dst + i) = w;
        } else {
            for (long j = i; j < HW && j < i + 8; ++j) {
                const int64_t v = src[j];
                dst[j] = (o < 0 ? v != 0 : v == o) ? 255 : 0;
            }
        }
    }
}

}  // namespace

extern "C" {

long s2d_color_ids_workspace_words(int T) { return (long)T * IDSLOTS; }

int s2d_color_masks_to_ids(const uint8_t *rgb, int T, int H, int W, unsigned int *workspace, int *n_ids, int64_t *ids, int *overflow,
                           hipStream_t stream)
{
    if (T < 0 || H < 1 || W < 1) return S2D_ERR_ARG;
    if (s2d_zero_async(overflow, sizeof(int), stream) != S2D_OK) return S2D_ERR_LAUNCH;
    if (T == 0) return S2D_OK;
    const long HW = (long)H * W;
    if (s2d_zero_async(workspace, sizeof(unsigned int) * (size_t)T * IDSLOTS, stream) != S2D_OK) return S2D_ERR_LAUNCH;
    const int bx = (int)(HW / 4096 > 0 ? (HW / 4096 < 256 ? HW / 4096 : 256) : 1);
    hipLaunchKernelGGL(idmap_collect_kernel, dim3(bx, T), dim3(256), 0, stream, rgb, HW, workspace, overflow);
    hipLaunchKernelGGL(idmap_sort_kernel, dim3(T), dim3(1024), 0, stream, workspace, n_ids, overflow);
    hipLaunchKernelGGL(idmap_apply_kernel, dim3(bx, T), dim3(256), 0, stream, rgb, HW, workspace, n_ids, ids);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

int s2d_tracks_to_masks_u8(const float *tracks, int T, int Np, int H, int W, uint8_t *masks, hipStream_t stream)
{
    if (s2d_zero_async(masks, (size_t)T * H * W, stream) != S2D_OK) return S2D_ERR_LAUNCH;
    const long n = (long)T * Np;
    if (n == 0) return S2D_OK;
    hipLaunchKernelGGL(scatter_tracks_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, tracks, T, Np, H, W, masks);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

int s2d_point_id_counts(const uint8_t *point_masks, const int64_t *idmap, int T, int H, int W, int Hi, int Wi, int max_id,
                        int *counts, int *total, hipStream_t stream)
{
    if (max_id < 0 || max_id > 8190) return S2D_ERR_ARG;
    if (T == 0) return S2D_OK;
    if (s2d_zero_async(counts, sizeof(int) * (size_t)T * (max_id + 1), stream) != S2D_OK) return S2D_ERR_LAUNCH;
    if (s2d_zero_async(total, sizeof(int) * (size_t)T, stream) != S2D_OK) return S2D_ERR_LAUNCH;
    hipLaunchKernelGGL(point_id_hist_kernel, dim3(32, T), dim3(256), sizeof(int) * (max_id + 2), stream, point_masks, idmap, H, W,
                       Hi, Wi, max_id, counts, total);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

int s2d_idmap_presence_u8(const int64_t *idmap, int T, int Hi, int Wi, int max_id, uint8_t *presence, hipStream_t stream)
{
    if (max_id < 0 || max_id > 8190) return S2D_ERR_ARG;
    if (T == 0) return S2D_OK;
    if (s2d_zero_async(presence, (size_t)T * (max_id + 1), stream) != S2D_OK) return S2D_ERR_LAUNCH;
    hipLaunchKernelGGL(id_presence_kernel, dim3(32, T), dim3(256), sizeof(int) * (max_id + 1), stream, idmap, (long)Hi * Wi, max_id,
                       presence);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

int s2d_idmap_select_masks_u8(const int64_t *idmap, int T, int H, int W, const int *frames, const int *objs, int K, uint8_t *out,
                              hipStream_t stream)
{
    if (T <= 0 || H <= 0 || W <= 0 || K < 0) return S2D_ERR_ARG;
    if (K == 0) return S2D_OK;
    const long HW = (long)H * W;
    int gx = cdiv(HW, 256 * 8);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(select_masks_kernel, dim3(gx, K), dim3(256), 0, stream, idmap, HW, frames, objs, out);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

int s2d_visibility_curve_f32(const uint8_t *visibility, int T, int Np, float *curve, hipStream_t stream)
{
    if (T == 0 || Np <= 0) return S2D_OK;
    hipLaunchKernelGGL(visibility_kernel, dim3(T), dim3(256), 0, stream, visibility, Np, curve);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

int s2d_local_corr_f32(const float *fmap_nhwc, const float *coords, const float *support, int T, int Np, int H, int W, int C,
                       int r, float *corr, hipStream_t stream)
{
    const int S = (2 * r + 1) * (2 * r + 1);
    if (r < 0 || (C & 3) || S > 64) return S2D_ERR_ARG;                   // 16-B channel vectors; at most 16 x 16 threads x 4 x 4 outputs
    if (T == 0 || Np == 0) return S2D_OK;
    // thread grid TB x TB with 4 TB >= S: r = 3 (S = 49) -> 13 x 13 threads, 52 staged rows
    const int TB = (S + 3) / 4;
    const size_t lds = sizeof(float) * 2 * (S + 1) * (C + 4);
    if (lds > 96 * 1024 || TB > 16) return S2D_ERR_ARG;
#define S2D_LC(tb)                                                                                                              \
    case tb: {                                                                                                                  \
        static S2dDevOnce set;                                                                                                \
        if (!set.done() && lds > 48 * 1024) {                                                                                          \
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(local_corr_kernel<tb>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess) \
                return S2D_ERR_LAUNCH;                                                                                          \
            set.mark();                                                                                                         \
        }                                                                                                                       \
        hipLaunchKernelGGL(local_corr_kernel<tb>, dim3(Np, T), dim3(256), lds, stream, fmap_nhwc, coords, support, T, Np, H, W, C, r, corr); \
    } break;
    switch (TB) {
        S2D_LC(1) S2D_LC(3) S2D_LC(7) S2D_LC(13)                              // r = 0, 1, 2, 3
    default: return S2D_ERR_ARG;                                           // S = (2r+1)^2 <= 64 gives TB in {1, 3, 7, 13} only
    }
#undef S2D_LC
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

}  // extern "C"
