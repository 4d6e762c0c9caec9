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
    const long r0 = (long)blockIdx.y * 64, c0 = (long)blockIdx.x * 64;
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

// out[i] = beta * out[i] + sum_s part[s][i], s ascending (fixed order)
__global__ __launch_bounds__(256) void reduce_slices_kernel(const float *__restrict__ part, int S, long n, long stride, float beta,
                                                            float *__restrict__ out)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float acc = beta != 0.f ? beta * out[i] : 0.f;
    for (int s = 0; s < S; ++s) acc += part[(long)s * stride + i];
    out[i] = acc;
}

// part[s][c] = sum over the rows of slice s of in[r][c]  (rows_per_slice rows each; sequential in r: fixed order)
__global__ __launch_bounds__(256) void colsum_slices_kernel(const float *__restrict__ in, long R, long C, long ldi, long rows_per_slice,
                                                            float *__restrict__ part)
{
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    const int s = blockIdx.y;
    if (c >= C) return;
    const long r0 = (long)s * rows_per_slice, r1 = r0 + rows_per_slice < R ? r0 + rows_per_slice : R;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;                   // four interleaved chains, combined in a fixed order
    long r = r0;
    for (; r + 3 < r1; r += 4) {
        a0 += in[r * ldi + c]; a1 += in[(r + 1) * ldi + c]; a2 += in[(r + 2) * ldi + c]; a3 += in[(r + 3) * ldi + c];
    }
    for (; r < r1; ++r) a0 += in[r * ldi + c];
    part[(long)s * C + c] = (a0 + a1) + (a2 + a3);
}

}  // namespace

extern "C" {

int s2d_transpose_f32(const float *in, long R, long C, long ldi, float *out, long ldo, hipStream_t stream)
{
    if (R < 0 || C < 0 || ldi < C || ldo < R) return S2D_ERR_ARG;
    if (R == 0 || C == 0) return S2D_OK;
    if ((R + 63) / 64 > 65535) return S2D_ERR_ARG;
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(C, 64), cdiv(R, 64)), dim3(256), 0, stream, in, R, C, ldi, out, ldo);
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
