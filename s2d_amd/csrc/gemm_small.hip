// Split-fp16 x3 NT GEMM for a handful of rows (M <= 256: the video decoder's query side, 100 queries x 2 clips -- 11 linear
// layers per decoder layer, video_mask2former_transformer_decoder.py:36-160), static pre-split weights.
//
// The tiled kernels of gemm_bf16.hip are throughput kernels: a workgroup walks its k-tiles one memory round trip at a time, and
// at M = 200 a launch has 8 workgroups that do nothing else (8 us at K = 256, 53 us at K = 2048).  Here a launch is one round
// trip: a workgroup owns a 32 x 64 tile of C, its NW waves split K among themselves (64 .. 256 each), every lane loads its MFMA
// fragments straight from global memory -- A rows as fp32 (split to fp16 hi / scaled lo in registers), B from the weight image
// [n][k / 32][16 w hi | 16 w lo] -- with all loads of a 64-wide chunk in flight together, and the NW partial tiles are added in
// wave order through LDS (fixed order: reproducible) before the usual epilogue (scale, bias, residual, ReLU).
#include <hip/hip_runtime.h>

#include "common.h"
#include "gemm_params.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __fp16 h16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int TM = 32, TN = 64, RS = 72;      // RS: LDS row stride of a partial tile (rows 4 apart land 32 banks apart)

// x = h + l * 2^-11, h = fp16_rtz(x), l = fp16_rtz((x - h) * 2^11): the same split as gemm_bf16.hip's split4_f16
__device__ __forceinline__ void split8(const f32x4 a, const f32x4 b, f16x8 &hi, f16x8 &lo)
{
    u32x4 h, l;
    const f32x4 v[2] = {a, b};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const h16x2 h0 = __builtin_amdgcn_cvt_pkrtz(v[i][0], v[i][1]), h1 = __builtin_amdgcn_cvt_pkrtz(v[i][2], v[i][3]);
        const f32x2 x0 = {v[i][0], v[i][1]}, x1 = {v[i][2], v[i][3]};
        const f32x2 r0 = (x0 - __builtin_convertvector(h0, f32x2)) * 2048.f, r1 = (x1 - __builtin_convertvector(h1, f32x2)) * 2048.f;
        const h16x2 l0 = __builtin_amdgcn_cvt_pkrtz(r0[0], r0[1]), l1 = __builtin_amdgcn_cvt_pkrtz(r1[0], r1[1]);
        h[2 * i] = __builtin_bit_cast(unsigned int, h0); h[2 * i + 1] = __builtin_bit_cast(unsigned int, h1);
        l[2 * i] = __builtin_bit_cast(unsigned int, l0); l[2 * i + 1] = __builtin_bit_cast(unsigned int, l1);
    }
    hi = __builtin_bit_cast(f16x8, h);
    lo = __builtin_bit_cast(f16x8, l);
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void gemm_small_m_kernel(GemmParams p, int kper)
{
    extern __shared__ __attribute__((aligned(16))) float red[];      // [NW][TM][RS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, h = lane >> 5;
    const int tiles_n = (p.N + TN - 1) / TN;
    const int m0 = (blockIdx.x / tiles_n) * TM, n0 = (blockIdx.x % tiles_n) * TN;
    // rows / columns past the edge compute on the last valid one and are not stored
    const int row = min(m0 + l32, p.M - 1);
    const float *Ap = p.A + (long)row * p.lda + wave * kper + 8 * h;
    const unsigned int *Bp[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) Bp[t] = p.Bsplit + ((long)min(n0 + 32 * t + l32, p.N - 1) * p.kblocks + (wave * kper) / 32) * 32 + 4 * h;
    // epilogue operands of this thread's outputs (column n0 + lane, rows m0 + wave + j NW), fetched before anything else: the
    // epilogue then has no memory latency of its own
    constexpr int PER = TM / NW;
    const int gc = n0 + lane, gcc = min(gc, p.N - 1);
    const float sc = p.scale ? p.scale[gcc] : 1.f, bi = p.bias ? p.bias[gcc] : 0.f;
    float rs[PER];
    const bool has_res = p.res && gc < p.res_cols;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int gr = min(m0 + wave + j * NW, p.M - 1);
        rs[j] = has_res ? p.res[(long)(p.res_rows ? gr % p.res_rows : gr) * p.ldr + gcc] : 0.f;
    }
    f32x16 accm[2], accx[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { accm[t][r] = 0.f; accx[t][r] = 0.f; }
    for (int kc = 0; kc < kper; kc += 64) {
        // the chunk's 8 + 16 fragment loads, all in flight together: k16-step st covers k = kc + 16 st + 8 h .. + 7
        f32x4 a[4][2];
        u32x4 bh[2][4], bl[2][4];
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            a[st][0] = *reinterpret_cast<const f32x4 *>(Ap + kc + 16 * st);
            a[st][1] = *reinterpret_cast<const f32x4 *>(Ap + kc + 16 * st + 4);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const unsigned int *b = Bp[t] + (kc / 32 + (st >> 1)) * 32 + 8 * (st & 1);
                bh[t][st] = *reinterpret_cast<const u32x4 *>(b);
                bl[t][st] = *reinterpret_cast<const u32x4 *>(b + 16);
            }
        __builtin_amdgcn_sched_barrier(0);                 // keep the 24 loads ahead of their first use (the scheduler would sink each to it)
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            f16x8 ah, al;
            split8(a[st][0], a[st][1], ah, al);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const f16x8 bhv = __builtin_bit_cast(f16x8, bh[t][st]), blv = __builtin_bit_cast(f16x8, bl[t][st]);
                accx[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bhv, accx[t], 0, 0, 0);
                accx[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, blv, accx[t], 0, 0, 0);
                accm[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bhv, accm[t], 0, 0, 0);
            }
        }
    }
    float *mine = red + wave * TM * RS;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            mine[((r & 3) + 8 * (r >> 2) + 4 * h) * RS + 32 * t + l32] = accm[t][r] + accx[t][r] * (1.0f / 2048.0f);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int r = wave + j * NW;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[(w * TM + r) * RS + lane];       // wave order: the sum does not depend on timing
        const int gr = m0 + r;
        v = v * sc + bi;
        v += rs[j];
        if (p.relu) v = fmaxf(v, 0.f);
        if (gr < p.M && gc < p.N) p.C[(long)gr * p.ldc + gc] = v;
    }
}

template <int NW>
int launch_small(const GemmParams &p, hipStream_t st)
{
    const size_t lds = (size_t)NW * TM * RS * sizeof(float);
    static S2dDevOnce attr;
    if (!attr.done() && lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_small_m_kernel<NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return S2D_ERR_LAUNCH;
        attr.mark();
    }
    const int nwg = ((p.M + TM - 1) / TM) * ((p.N + TN - 1) / TN);
    hipLaunchKernelGGL((gemm_small_m_kernel<NW>), dim3(nwg), dim3(NW * 64), lds, st, p, p.K / NW);
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

}  // namespace

// true when the launch qualifies (the dispatcher of gemm_bf16.hip asks first): few rows, static pre-split B, K in whole 64-wide
// chunks per wave, a plain epilogue, 16-B aligned A rows
bool s2d_gemm_small_m_ok(const GemmParams &p, bool conv, int batch, int f16)
{
    static int on = -1;
    if (on < 0) { const char *e = getenv("S2D_GEMM_SMALL"); on = e ? atoi(e) : 1; }
    return on && f16 && !conv && batch == 1 && p.M >= 1 && p.M <= 256 && p.Bsplit && !p.Asplit && !p.gate && !p.drop_thresh && p.K >= 64 && p.K % 64 == 0 &&
           (p.lda & 3) == 0 && ((reinterpret_cast<uintptr_t>(p.A) & 15) == 0);
}

int s2d_launch_gemm_small_m(const GemmParams &p, hipStream_t st)
{
    const int chunks = p.K / 64;
    if (chunks % 8 == 0) return launch_small<8>(p, st);
    if (chunks % 4 == 0) return launch_small<4>(p, st);
    if (chunks % 2 == 0) return launch_small<2>(p, st);
    return launch_small<1>(p, st);
}
