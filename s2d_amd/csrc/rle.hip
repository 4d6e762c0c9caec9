// COCO run-length encoding of binary masks on the device: the wire format either side of the path --
// `mask_util.encode(np.array(mask[:, :, None], order="F"))` in the evaluator (mask2former_video/data_video/ytvis_eval.py:
// 345-350) and in the keymask annotation writer (keymask_ident/annotations.py:100-106, with `area` and `toBbox`).
// pycocotools is a third-party dependency absent from the reference tree; this restates its published algorithm
// (maskApi.c rleEncode: runs of the COLUMN-major flattened mask, alternating 0s / 1s, starting with the zero run).
//
// A boolean [T,H,W] prediction is 0.9 MB per 720p frame as bytes and a few KB as runs, so encoding where the mask was
// made removes the device->host copy that dominates inference_video (eval row).  Two passes over a frame, one
// workgroup per frame, a thread owning 4 adjacent columns and walking down the rows (row-major memory is read 4 bytes per
// lane, coalesced along x; the run order is column-major, so the ownership is by column):
//   pass 1 counts the run boundaries of every column (a boundary = pixel != its column-major predecessor), scans the
//          counts over the columns (exclusive), and reduces area and the tight box;
//   pass 2 walks the same way and writes each boundary's column-major position x*H + y at its slot.
// The host turns positions into run lengths (differences) and into the LEB-like ASCII string (s2d_amd/rle.py).
#include "common.h"

namespace {

constexpr int RT = 1024;   // threads per frame workgroup: 4096 columns per sweep

// value of pixel (y, x) of the column-major predecessor of (0, x): (H-1, x-1), or 0 before the first pixel
__device__ __forceinline__ unsigned int col_pred(const uint8_t *__restrict__ fr, int H, int W, int x)
{
    return x > 0 ? (fr[(long)(H - 1) * W + x - 1] != 0) : 0u;
}

__global__ __launch_bounds__(RT) void rle_count_kernel(const uint8_t *__restrict__ masks, int H, int W, int *__restrict__ col_off,
                                                       int *__restrict__ nbound, int *__restrict__ area, int *__restrict__ bbox)
{
    __shared__ int scan[RT];
    __shared__ int carry;
    const int f = blockIdx.x;
    const uint8_t *fr = masks + (long)f * H * W;
    const bool vec = (W & 3) == 0 && (reinterpret_cast<uintptr_t>(fr) & 3) == 0;
    int t_area = 0, xs = W, xe = -1, ys = H, ye = -1;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int x0 = 0; x0 < W; x0 += 4 * RT) {
        const int x = x0 + 4 * threadIdx.x;
        int cnt[4] = {0, 0, 0, 0};
        if (x < W) {
            unsigned int prev[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) prev[j] = x + j < W ? col_pred(fr, H, W, x + j) : 0u;
            for (int y = 0; y < H; ++y) {
                unsigned int v[4];
                if (vec) {
                    const unsigned int w4 = *reinterpret_cast<const unsigned int *>(fr + (long)y * W + x);
                    v[0] = (w4 & 0xFFu) != 0; v[1] = (w4 & 0xFF00u) != 0; v[2] = (w4 & 0xFF0000u) != 0; v[3] = (w4 & 0xFF000000u) != 0;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = x + j < W ? (fr[(long)y * W + x + j] != 0) : 0u;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    cnt[j] += (int)(v[j] != prev[j]);
                    prev[j] = v[j];
                    if (v[j]) { ++t_area; xs = min(xs, x + j); xe = max(xe, x + j); }
                }
                if (v[0] | v[1] | v[2] | v[3]) { ys = min(ys, y); ye = max(ye, y); }
            }
        }
        // exclusive scan of the 4*RT column counts of this sweep (+ carry from earlier sweeps)
        const int mine = cnt[0] + cnt[1] + cnt[2] + cnt[3];
        scan[threadIdx.x] = mine;
        __syncthreads();
        for (int o = 1; o < RT; o <<= 1) {
            const int add = threadIdx.x >= o ? scan[threadIdx.x - o] : 0;
            __syncthreads();
            scan[threadIdx.x] += add;
            __syncthreads();
        }
        int base = carry + scan[threadIdx.x] - mine;
        if (x < W) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (x + j < W) { col_off[(long)f * W + x + j] = base; base += cnt[j]; }
        }
        __syncthreads();
        if (threadIdx.x == RT - 1) carry += scan[RT - 1];
        __syncthreads();
    }
    // reductions: area (sum), box (min / max)
    auto wave_red = [&](int v, int op) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int u = __shfl_xor(v, o, 64);
            v = op == 0 ? v + u : (op == 1 ? min(v, u) : max(v, u));
        }
        return v;
    };
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int vals[5] = {wave_red(t_area, 0), wave_red(xs, 1), wave_red(ys, 1), wave_red(xe, 2), wave_red(ye, 2)};
    __shared__ int r5[5][RT / 64];
    if (lane == 0)
        for (int i = 0; i < 5; ++i) r5[i][wv] = vals[i];
    __syncthreads();
    if (threadIdx.x == 0) {
        int a = 0, bxs = W, bys = H, bxe = -1, bye = -1;
        for (int w = 0; w < RT / 64; ++w) { a += r5[0][w]; bxs = min(bxs, r5[1][w]); bys = min(bys, r5[2][w]); bxe = max(bxe, r5[3][w]); bye = max(bye, r5[4][w]); }
        nbound[f] = carry;
        area[f] = a;
        if (a == 0) { bbox[4 * f] = bbox[4 * f + 1] = bbox[4 * f + 2] = bbox[4 * f + 3] = 0; }   // rleToBbox of an empty mask
        else { bbox[4 * f] = bxs; bbox[4 * f + 1] = bys; bbox[4 * f + 2] = bxe - bxs + 1; bbox[4 * f + 3] = bye - bys + 1; }
    }
}

__global__ __launch_bounds__(RT) void rle_positions_kernel(const uint8_t *__restrict__ masks, int H, int W, const int *__restrict__ col_off,
                                                           const long *__restrict__ frame_off, int *__restrict__ positions)
{
    const int f = blockIdx.x;
    const uint8_t *fr = masks + (long)f * H * W;
    const bool vec = (W & 3) == 0 && (reinterpret_cast<uintptr_t>(fr) & 3) == 0;
    int *out = positions + frame_off[f];
    for (int x0 = 0; x0 < W; x0 += 4 * RT) {
        const int x = x0 + 4 * threadIdx.x;
        if (x >= W) continue;
        unsigned int prev[4];
        int slot[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            prev[j] = x + j < W ? col_pred(fr, H, W, x + j) : 0u;
            slot[j] = x + j < W ? col_off[(long)f * W + x + j] : 0;
        }
        for (int y = 0; y < H; ++y) {
            unsigned int v[4];
            if (vec) {
                const unsigned int w4 = *reinterpret_cast<const unsigned int *>(fr + (long)y * W + x);
                v[0] = (w4 & 0xFFu) != 0; v[1] = (w4 & 0xFF00u) != 0; v[2] = (w4 & 0xFF0000u) != 0; v[3] = (w4 & 0xFF000000u) != 0;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = x + j < W ? (fr[(long)y * W + x + j] != 0) : 0u;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (v[j] != prev[j]) out[slot[j]++] = (x + j) * H + y;
                prev[j] = v[j];
            }
        }
    }
}

// ---- run lengths -> the ASCII string of maskApi.c rleToString, on the device --------------------------------------------
// count c of mask f (local index i, n boundaries): cnt(i) = pos(i) - pos(i-1) with pos(-1) = 0, pos(n) = H*W; the value
// written is x = cnt(i) - cnt(i-2) for i > 2; chars: 5 bits each, bit 5 = "more", + 48.
struct RleStr {
    const int *pos;
    const long *frame_off;     // [F+1]
    int F;
    long hw, ncounts;
};
__device__ __forceinline__ long rle_value(const RleStr &p, long c, int &f_out, long &i_out)
{
    int lo = 0, hi = p.F;                                  // largest f with frame_off[f] + f <= c
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (p.frame_off[mid] + mid <= c) lo = mid; else hi = mid;
    }
    const int f = lo;
    const long b = p.frame_off[f], n = p.frame_off[f + 1] - b, i = c - (b + f);
    auto at = [&](long k) -> long { return k < 0 ? 0 : (k < n ? (long)p.pos[b + k] : p.hw); };
    long x = at(i) - at(i - 1);
    if (i > 2) x -= at(i - 2) - at(i - 3);
    f_out = f; i_out = i;
    return x;
}
__global__ __launch_bounds__(256) void rle_len_kernel(RleStr p, int *__restrict__ len)
{
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c >= p.ncounts) return;
    int f; long i;
    long x = rle_value(p, c, f, i);
    int k = 0;
    bool more = true;
    while (more) {
        const int ch = (int)(x & 0x1f);
        x >>= 5;
        more = (ch & 0x10) ? x != -1 : x != 0;
        ++k;
    }
    len[c] = k;
}
__global__ __launch_bounds__(256) void rle_chars_kernel(RleStr p, const int *__restrict__ len, const int *__restrict__ off,
                                                        uint8_t *__restrict__ chars, long *__restrict__ str_off)
{
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c >= p.ncounts) return;
    int f; long i;
    long x = rle_value(p, c, f, i);
    uint8_t *o = chars + off[c];
    if (i == 0) str_off[f] = off[c];
    if (c == p.ncounts - 1) str_off[p.F] = (long)off[c] + len[c];
    bool more = true;
    while (more) {
        int ch = (int)(x & 0x1f);
        x >>= 5;
        more = (ch & 0x10) ? x != -1 : x != 0;
        if (more) ch |= 0x20;
        *o++ = (uint8_t)(ch + 48);
    }
}

}  // namespace

extern "C" {

long s2d_rle_string_workspace_bytes(long ncounts)
{
    if (ncounts <= 0) return 0;
    return 2 * ((ncounts * 4 + 255) / 256 * 256) + (long)s2d_exclusive_scan_i32_temp_bytes((size_t)ncounts) + 256;
}

int s2d_rle_strings_u8(const int *positions, const long *frame_off, int F, long hw, long ncounts, void *workspace, long workspace_bytes,
                       uint8_t *chars, long *str_off, hipStream_t stream)
{
    if (F < 0 || ncounts < F || hw < 1 || hw >= (1L << 31) || ncounts * 7 >= (1L << 31)) return S2D_ERR_ARG;
    if (F == 0) return S2D_OK;
    if (workspace_bytes < s2d_rle_string_workspace_bytes(ncounts)) return S2D_ERR_ARG;
    const long slab = (ncounts * 4 + 255) / 256 * 256;
    int *len = reinterpret_cast<int *>(workspace);
    int *off = reinterpret_cast<int *>(reinterpret_cast<char *>(workspace) + slab);
    void *tmp = reinterpret_cast<char *>(workspace) + 2 * slab;
    RleStr p{positions, frame_off, F, hw, ncounts};
    hipLaunchKernelGGL(rle_len_kernel, dim3(cdiv(ncounts, 256)), dim3(256), 0, stream, p, len);
    if (int e = s2d_exclusive_scan_i32(len, off, (size_t)ncounts, tmp, (size_t)(workspace_bytes - 2 * slab), stream)) return e;
    hipLaunchKernelGGL(rle_chars_kernel, dim3(cdiv(ncounts, 256)), dim3(256), 0, stream, p, len, off, chars, str_off);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

int s2d_rle_count_u8(const uint8_t *masks, int F, int H, int W, int *col_off, int *nbound, int *area, int *bbox, hipStream_t stream)
{
    if (F < 0 || H < 1 || W < 1 || (long)H * W >= (1L << 31)) return S2D_ERR_ARG;
    if (F == 0) return S2D_OK;
    hipLaunchKernelGGL(rle_count_kernel, dim3(F), dim3(RT), 0, stream, masks, H, W, col_off, nbound, area, bbox);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

int s2d_rle_positions_u8(const uint8_t *masks, int F, int H, int W, const int *col_off, const long *frame_off, int *positions,
                         hipStream_t stream)
{
    if (F < 0 || H < 1 || W < 1 || (long)H * W >= (1L << 31)) return S2D_ERR_ARG;
    if (F == 0) return S2D_OK;
    hipLaunchKernelGGL(rle_positions_kernel, dim3(F), dim3(RT), 0, stream, masks, H, W, col_off, frame_off, positions);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

}  // extern "C"
