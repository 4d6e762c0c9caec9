// Elementwise form of the counter-based dropout mask (dropout.h): y = x * mask / (1 - p).  The forward applies the mask
// in the producing GEMM's epilogue (gemm_bf16.hip); this kernel is what the backward of the three encoder-layer dropout
// sites runs on the incoming gradient (msdeformattn.py:101-125), regenerating the mask from (seed, site, row, column).
#include "common.h"
#include "dropout.h"

namespace {

__global__ __launch_bounds__(256) void dropout_kernel(const float *__restrict__ x, long M, int N, uint32_t thresh, float scale,
                                                      uint32_t k0, uint32_t k1, uint32_t site, uint32_t row0, float *__restrict__ y)
{
    const int nb = N >> 3;                                   // 8-column mask blocks per row
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * nb) return;
    const long row = i / nb;
    const int cb = (int)(i - row * nb);
    const f32x4 a = *reinterpret_cast<const f32x4 *>(x + row * N + cb * 8);
    const f32x4 b = *reinterpret_cast<const f32x4 *>(x + row * N + cb * 8 + 4);
    float m[8];
    s2d_dropout8((uint32_t)row + row0, (uint32_t)cb, site, k0, k1, thresh, scale, m);
    const f32x4 ya = {a[0] * m[0], a[1] * m[1], a[2] * m[2], a[3] * m[3]};
    const f32x4 yb = {b[0] * m[4], b[1] * m[5], b[2] * m[6], b[3] * m[7]};
    *reinterpret_cast<f32x4 *>(y + row * N + cb * 8) = ya;
    *reinterpret_cast<f32x4 *>(y + row * N + cb * 8 + 4) = yb;
}

}  // namespace

extern "C" int s2d_dropout_f32(const float *x, long M, int N, float p, uint64_t seed, unsigned site, unsigned row0, float *y, hipStream_t stream)
{
    if (!(p >= 0.f) || p >= 1.f || M < 0 || N <= 0 || (N & 7) || M >= (1L << 32)) return S2D_ERR_ARG;
    if (M == 0) return S2D_OK;
    const unsigned thresh = s2d_dropout_thresh(p);
    const float scale = thresh ? s2d_dropout_scale(thresh) : 1.0f;
    const long n = M * (N >> 3);
    hipLaunchKernelGGL(dropout_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, x, M, N, thresh, scale,
                       (uint32_t)(seed & 0xFFFFFFFFull), (uint32_t)(seed >> 32), site, row0, y);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}
