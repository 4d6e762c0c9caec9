// Training-step callers after the loss, the part that does not need backward kernels: what
// `grad_scaler.step(optimizer)` + the EMA teacher update do per iteration in the reference --
//   GradScaler.unscale_ (grads *= 1/scale, inf/nan check; engine/train_loop.py:709-726),
//   torch.nn.utils.clip_grad_norm_ over ALL parameters (FullModelGradientClippingOptimizer, train_net_video.py:188-203),
//   torch.optim.AdamW.step (per-group lr / weight decay; :205-213),
//   teacher = m * teacher + (1 - m) * student (engine/train_loop.py:754-764)
// -- as two launches over a table of tensors instead of ~6 foreach passes + a python loop over 400 parameters:
//   1. grad_sqnorm_kernel + grad_norm_finalize_kernel: sum of squares of the unscaled gradients in fixed slots (double),
//      reduced in a fixed order -> total norm, clip coefficient, found-inf flag, all left on the device (no host sync);
//   2. adamw_ema_kernel: one pass that reads g, p, m, v, teacher and writes p, m, v, teacher (36 B per parameter: the HBM
//      roofline of the whole optimizer + EMA step is 44 M x 36 B / 8 TB/s = 0.2 ms).
// Parameters stay where torch allocated them: a device table holds (param, grad, exp_avg, exp_avg_sq, ema) pointers per
// tensor and (tensor, offset) per chunk of elements (16 K in s2d_amd/optim.py); a workgroup owns one chunk.
// The arithmetic is torch's single-tensor AdamW in its operation order (fp32), so the result matches torch.optim.AdamW
// to rounding.
#include "common.h"

namespace {

struct Table {
    const void *const *ptrs;   // [ntensors][5]: param, grad (may be null), exp_avg, exp_avg_sq, ema target (may be null)
    const long *numel;         // [ntensors]
    const int *chunk_tensor;   // [nchunks]
    const long *chunk_off;     // [nchunks]
    int chunk;                 // elements per chunk
};

__device__ __forceinline__ double block_sum_fixed(double v, double *red)
{
    v = wave_sum_d(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) red[wv] = v;
    __syncthreads();
    double s = 0.;
    if (threadIdx.x == 0)
        for (unsigned int w = 0; w < blockDim.x / 64; ++w) s += red[w];
    return s;   // valid in thread 0
}

__global__ __launch_bounds__(256) void grad_sqnorm_kernel(Table tb, float inv_scale, double *__restrict__ partial)
{
    __shared__ double red[4];
    const int c = blockIdx.x, ti = tb.chunk_tensor[c];
    const long off = tb.chunk_off[c];
    const float *g = static_cast<const float *>(tb.ptrs[ti * 5 + 1]);
    double acc = 0.;
    if (g) {
        const long n = tb.numel[ti] - off < tb.chunk ? tb.numel[ti] - off : tb.chunk;
        g += off;
        // squares and their sum in double: the sum is finite exactly when every (unscaled) gradient is finite
        auto sq = [&](float x) { const double d = (double)(x * inv_scale); acc += d * d; };
        if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
            const long n4 = n >> 2;
            for (long i = threadIdx.x; i < n4; i += 256) {
                const f32x4 x = reinterpret_cast<const f32x4 *>(g)[i];
                sq(x[0]); sq(x[1]); sq(x[2]); sq(x[3]);
            }
            for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) sq(g[i]);
        } else {
            for (long i = threadIdx.x; i < n; i += 256) sq(g[i]);
        }
    }
    const double s = block_sum_fixed(acc, red);
    if (threadIdx.x == 0) partial[c] = s;
}

// normbuf[0] = total norm, [1] = clip coefficient (clip_grad_norm_: max_norm / (total + 1e-6) clamped to 1), [2] = found_inf
__global__ __launch_bounds__(256) void grad_norm_finalize_kernel(const double *__restrict__ partial, int nchunks, float max_norm,
                                                                 float *__restrict__ normbuf)
{
    __shared__ double red[4];
    double acc = 0.;
    for (int i = threadIdx.x; i < nchunks; i += 256) acc += partial[i];
    const double s = block_sum_fixed(acc, red);
    if (threadIdx.x == 0) {
        const float total = (float)sqrt(s);
        float clip = 1.f;
        if (max_norm > 0.f) {
            clip = max_norm / (total + 1e-6f);
            if (clip > 1.f) clip = 1.f;
        }
        normbuf[0] = total;
        normbuf[1] = clip;
        normbuf[2] = (isfinite(s) ? 0.f : 1.f);
    }
}

struct StepConst {
    float beta2, w1, omb2, eps, bc2s, inv_scale, ema_m, ema_om;
    double lr_factor, bc1;
    int do_ema;
};

__device__ __forceinline__ void adamw_one(float g, float &p, float &m, float &v, const StepConst &k, float clip, float decay, float nss)
{
    g = g * k.inv_scale;                         // GradScaler.unscale_
    g = g * clip;                                // clip_grad_norm_
    p = p * decay;                               // param.mul_(1 - lr * weight_decay)
    // torch's elementwise kernels evaluate a + b * c as one fused multiply-add (nvcc / hipcc contraction on the device,
    // vec::fmadd on AVX2 hosts); the three such expressions are written as explicit fmaf so the rounding matches
    m = fmaf(k.w1, g - m, m);                    // exp_avg.lerp_(grad, 1 - beta1):  self + weight * (end - self)
    v = v * k.beta2;                             // exp_avg_sq.mul_(beta2)
    v = fmaf(k.omb2 * g, g, v);                  //           .addcmul_(grad, grad, value = 1 - beta2)
    const float denom = sqrtf(v) / k.bc2s + k.eps;
    p = fmaf(nss, m / denom, p);                 // param.addcdiv_(exp_avg, denom, value = -step_size)
}

__global__ __launch_bounds__(256) void adamw_ema_kernel(Table tb, const double *__restrict__ hyper, StepConst k,
                                                        const float *__restrict__ normbuf)
{
    const int c = blockIdx.x, ti = tb.chunk_tensor[c];
    const long off = tb.chunk_off[c];
    const long n = tb.numel[ti] - off < tb.chunk ? tb.numel[ti] - off : tb.chunk;
    float *p = static_cast<float *>(const_cast<void *>(tb.ptrs[ti * 5 + 0])) + off;
    const float *g = static_cast<const float *>(tb.ptrs[ti * 5 + 1]);
    float *m = static_cast<float *>(const_cast<void *>(tb.ptrs[ti * 5 + 2])) + off;
    float *v = static_cast<float *>(const_cast<void *>(tb.ptrs[ti * 5 + 3])) + off;
    float *e = static_cast<float *>(const_cast<void *>(tb.ptrs[ti * 5 + 4]));
    const float clip = normbuf ? normbuf[1] : 1.f;
    const bool skip = normbuf && normbuf[2] != 0.f;                 // GradScaler: inf/nan gradients -> no optimizer step
    const bool upd = g != nullptr && !skip;
    const bool ema = k.do_ema && e != nullptr;
    if (!upd && !ema) return;
    if (g) g += off;
    if (e) e += off;
    const double lr = hyper[ti * 2] * k.lr_factor, wd = hyper[ti * 2 + 1];
    const float decay = (float)(1.0 - lr * wd), nss = (float)(-(lr / k.bc1));
    const uintptr_t al = reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v) |
                         (g ? reinterpret_cast<uintptr_t>(g) : 0) | (e ? reinterpret_cast<uintptr_t>(e) : 0);
    long done = 0;
    if ((al & 15) == 0) {
        const long n4 = n >> 2;
        for (long i = threadIdx.x; i < n4; i += 256) {
            f32x4 pv = reinterpret_cast<f32x4 *>(p)[i];
            if (upd) {
                const f32x4 gv = reinterpret_cast<const f32x4 *>(g)[i];
                f32x4 mv = reinterpret_cast<f32x4 *>(m)[i], vv = reinterpret_cast<f32x4 *>(v)[i];
#pragma unroll
                for (int j = 0; j < 4; ++j) { float pj = pv[j], mj = mv[j], vj = vv[j]; adamw_one(gv[j], pj, mj, vj, k, clip, decay, nss); pv[j] = pj; mv[j] = mj; vv[j] = vj; }
                reinterpret_cast<f32x4 *>(p)[i] = pv;
                reinterpret_cast<f32x4 *>(m)[i] = mv;
                reinterpret_cast<f32x4 *>(v)[i] = vv;
            }
            if (ema) {
                f32x4 tv = reinterpret_cast<f32x4 *>(e)[i];
                tv = tv * k.ema_m;                                   // teacher.mul_(m)
                tv = tv + k.ema_om * pv;                             //        .add_((1 - m) * student)
                reinterpret_cast<f32x4 *>(e)[i] = tv;
            }
        }
        done = n4 << 2;
    }
    for (long i = done + threadIdx.x; i < n; i += 256) {
        float pj = p[i];
        if (upd) {
            float mj = m[i], vj = v[i];
            adamw_one(g[i], pj, mj, vj, k, clip, decay, nss);
            p[i] = pj; m[i] = mj; v[i] = vj;
        }
        if (ema) {
            float t = e[i] * k.ema_m;
            e[i] = t + k.ema_om * pj;
        }
    }
}

}  // namespace

extern "C" {

int s2d_optim_grad_norm_f32(const void *const *ptrs, const long *numel, const int *chunk_tensor, const long *chunk_off, int nchunks,
                            int chunk, float inv_scale, float max_norm, double *partial, float *normbuf, hipStream_t stream)
{
    if (nchunks < 0 || chunk < 4 || (chunk & 3)) return S2D_ERR_ARG;
    Table tb{ptrs, numel, chunk_tensor, chunk_off, chunk};
    if (nchunks) hipLaunchKernelGGL(grad_sqnorm_kernel, dim3(nchunks), dim3(256), 0, stream, tb, inv_scale, partial);
    hipLaunchKernelGGL(grad_norm_finalize_kernel, dim3(1), dim3(256), 0, stream, partial, nchunks, max_norm, normbuf);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

int s2d_optim_adamw_ema_f32(const void *const *ptrs, const long *numel, const double *hyper, const int *chunk_tensor,
                            const long *chunk_off, int nchunks, int chunk, double lr_factor, double beta1, double beta2, double eps,
                            double bias_correction1, double bias_correction2_sqrt, float inv_scale, double ema_m,
                            const float *normbuf, hipStream_t stream)
{
    if (nchunks < 0 || chunk < 4 || (chunk & 3)) return S2D_ERR_ARG;
    if (nchunks == 0) return S2D_OK;
    Table tb{ptrs, numel, chunk_tensor, chunk_off, chunk};
    StepConst k;
    k.beta2 = (float)beta2; k.w1 = (float)(1.0 - beta1); k.omb2 = (float)(1.0 - beta2); k.eps = (float)eps;
    k.bc2s = (float)bias_correction2_sqrt; k.inv_scale = inv_scale;
    k.do_ema = ema_m >= 0.0 ? 1 : 0;
    k.ema_m = (float)ema_m; k.ema_om = (float)(1.0 - ema_m);
    k.lr_factor = lr_factor; k.bc1 = bias_correction1;
    hipLaunchKernelGGL(adamw_ema_kernel, dim3(nchunks), dim3(256), 0, stream, tb, hyper, k, normbuf);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

}  // extern "C"
