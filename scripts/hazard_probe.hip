// Probe for the stale-VGPR read found behind the round-1 "two-stream stale read" (Streams):
//     global_load_dword x4 ; s_waitcnt vmcnt(0) ; <first consumer of the LAST load's destination register>
// A victim kernel runs that sequence (pinned with inline asm on fixed registers) in a loop and compares the consumer's
// result with the value computed from a re-load of the same address; an aggressor kernel on a second HIP stream keeps the
// matrix pipe, LDS and the vector register file of the same CUs busy.  Reported per consumer form: mismatching
// (wave, iteration, lane) triples, by quarter-wave.
//
//   hipcc --offload-arch=gfx950 -O3 -o scripts/hazard_probe.bin scripts/hazard_probe.hip ; ./scripts/hazard_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// MODE 0: v_pk_mul_f32 straight behind the waitcnt (the failing shape)   1: v_mul_f32 x2 straight behind it
// MODE 2: s_nop 0 between waitcnt and v_pk_mul_f32                       3: v_mov_b32 of the last register, then v_pk_mul_f32
// MODE 4: as 0 but every lane active (no exec mask on the loads)         5: the four loads as ONE global_load_dwordx4, v_pk_mul_f32 behind it
template <int MODE>
__global__ __launch_bounds__(256) void victim(const float *__restrict__ src, int rows, int ld, int iters, unsigned *__restrict__ bad /*[4 quarters + total]*/)
{
    const int lane = threadIdx.x & 63;
    const int g = threadIdx.x & 31;
    const bool act = MODE == 4 || g < 25;                                   // lanes 25..31 / 57..63 idle, as in attn_mask_kernel
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    unsigned nb = 0;
    for (int it = 0; it < iters; ++it) {
        const long key = (wave * 2 + (lane >> 5) + (long)it * 7919) % (rows - 400);
        const int gg = g < 25 ? g : 24;
        const float *p0 = src + key * ld + 4 * gg + (it & 3), *p1 = p0 + ld, *p2 = p0 + 320L * ld, *p3 = p2 + ld;
        float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
        const f32x2 w = {0.5f, 0.25f};
        if (act) {
            if (MODE == 5) {
                asm volatile("global_load_dwordx4 v[20:23], %[p0], off\n"
                             "s_waitcnt vmcnt(0)\n"
                             "v_pk_mul_f32 v[22:23], %[w], v[22:23]\n"
                             "v_pk_mul_f32 v[20:21], %[w], v[20:21]\n"
                             "s_nop 1\n"
                             "v_mov_b32 %[o0], v20\n v_mov_b32 %[o1], v21\n v_mov_b32 %[o2], v22\n v_mov_b32 %[o3], v23\n"
                             : [o0] "=v"(o0), [o1] "=v"(o1), [o2] "=v"(o2), [o3] "=v"(o3)
                             : [p0] "v"(p0), [w] "v"(w) : "v20", "v21", "v22", "v23", "memory");
            } else {
#define LOADS "global_load_dword v20, %[p0], off\n global_load_dword v23, %[p1], off\n global_load_dword v22, %[p2], off\n global_load_dword v21, %[p3], off\n"
#define OUTS "s_nop 1\n v_mov_b32 %[o0], v20\n v_mov_b32 %[o1], v21\n v_mov_b32 %[o2], v22\n v_mov_b32 %[o3], v23\n"
#define OPS : [o0] "=v"(o0), [o1] "=v"(o1), [o2] "=v"(o2), [o3] "=v"(o3) : [p0] "v"(p0), [p1] "v"(p1), [p2] "v"(p2), [p3] "v"(p3), [w] "v"(w) : "v20", "v21", "v22", "v23", "memory"
                if (MODE == 0 || MODE == 4)
                    asm volatile(LOADS "s_waitcnt vmcnt(1)\n v_pk_mul_f32 v[22:23], %[w], v[22:23]\n s_waitcnt vmcnt(0)\n v_pk_mul_f32 v[20:21], %[w], v[20:21]\n" OUTS OPS);
                else if (MODE == 1)
                    asm volatile(LOADS "s_waitcnt vmcnt(1)\n v_pk_mul_f32 v[22:23], %[w], v[22:23]\n s_waitcnt vmcnt(0)\n v_mul_f32 v21, 0.25, v21\n v_mul_f32 v20, 0.5, v20\n" OUTS OPS);
                else if (MODE == 2)
                    asm volatile(LOADS "s_waitcnt vmcnt(1)\n v_pk_mul_f32 v[22:23], %[w], v[22:23]\n s_waitcnt vmcnt(0)\n s_nop 0\n v_pk_mul_f32 v[20:21], %[w], v[20:21]\n" OUTS OPS);
                else
                    asm volatile(LOADS "s_waitcnt vmcnt(1)\n v_pk_mul_f32 v[22:23], %[w], v[22:23]\n s_waitcnt vmcnt(0)\n v_mov_b32 v21, v21\n v_pk_mul_f32 v[20:21], %[w], v[20:21]\n" OUTS OPS);
            }
            // reference: plain re-loads, compiler-scheduled, single multiplies
            float e0, e1, e2, e3;
            if (MODE == 5) { e0 = 0.5f * p0[0]; e1 = 0.25f * p0[1]; e2 = 0.5f * p0[2]; e3 = 0.25f * p0[3]; }
            else { e0 = 0.5f * __builtin_nontemporal_load(p0); e1 = 0.25f * __builtin_nontemporal_load(p3);
                   e2 = 0.5f * __builtin_nontemporal_load(p2); e3 = 0.25f * __builtin_nontemporal_load(p1); }
            if (o1 != e1) { nb += 1; atomicAdd(&bad[lane >> 4], 1u); }                  // the LAST load's register
            if (o0 != e0 || o2 != e2 || o3 != e3) atomicAdd(&bad[5], 1u);                // any other register
        }
    }
    if (nb) atomicAdd(&bad[4], nb);
}

// aggressor: MFMA + LDS + global traffic, 2 workgroups per CU, runs until told to stop by iteration count
__global__ __launch_bounds__(256, 2) void aggressor(const float *__restrict__ a, float *__restrict__ out, int iters, int n)
{
    __shared__ float lds[4096];
    f32x16 acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
    const int t = threadIdx.x;
    long idx = ((long)blockIdx.x * 256 + t) * 4 % n;
    for (int it = 0; it < iters; ++it) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(a + idx);
        idx = (idx + 256L * 4 * 977) % n;
        lds[(t * 4 + it) & 4095] = v[0];
        __syncthreads();
        f16x8 x, y;
#pragma unroll
        for (int i = 0; i < 8; ++i) { x[i] = (_Float16)(v[i & 3] + lds[(t + i) & 4095]); y[i] = (_Float16)v[(i + 1) & 3]; }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, x, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, y, acc3, 0, 0, 0);
        }
        __syncthreads();
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + acc2[i] + acc3[i];
    out[(long)blockIdx.x * 256 + t] = s;
}

template <int MODE>
void run(const char *name, const float *src, int rows, int ld, unsigned *bad, hipStream_t s1, hipStream_t s2, const float *agg_in, float *agg_out,
         int n_agg, bool with_aggressor)
{
    CK(hipMemsetAsync(bad, 0, 8 * sizeof(unsigned), s1));
    CK(hipStreamSynchronize(s1));
    if (with_aggressor)
        for (int k = 0; k < 12; ++k) hipLaunchKernelGGL(aggressor, dim3(2048), dim3(256), 0, s2, agg_in, agg_out, 3000, n_agg);
    for (int k = 0; k < 24; ++k) hipLaunchKernelGGL(victim<MODE>, dim3(29440), dim3(256), 0, s1, src, rows, ld, 8, bad);
    CK(hipDeviceSynchronize());
    unsigned h[8];
    CK(hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost));
    printf("%-58s aggressor %d: last-load register wrong %u (quarter-waves 0..3: %u %u %u %u), other registers wrong %u  of %.1f M lane-checks\n",
           name, (int)with_aggressor, h[4], h[0], h[1], h[2], h[3], h[5], 24.0 * 29440 * 256 * 8 / 1e6);
    fflush(stdout);
}

int main()
{
    const int rows = 471040, ld = 100;
    std::vector<float> h((size_t)rows * ld);
    unsigned s = 12345u;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = 0.5f + (float)(s >> 8) * (1.0f / 16777216.0f); }   // never 0
    float *src, *agg_in, *agg_out; unsigned *bad;
    const int n_agg = 64 << 20;
    CK(hipMalloc(&src, h.size() * 4)); CK(hipMalloc(&agg_in, (size_t)n_agg * 4)); CK(hipMalloc(&agg_out, 2048 * 256 * 4)); CK(hipMalloc(&bad, 64));
    CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(agg_in, 0x3c, (size_t)n_agg * 4));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    for (int agg = 0; agg < 2; ++agg) {
        run<0>("dword x4; waitcnt; v_pk_mul_f32 (failing shape)", src, rows, ld, bad, s1, s2, agg_in, agg_out, n_agg, agg);
        run<1>("dword x4; waitcnt; v_mul_f32 x2", src, rows, ld, bad, s1, s2, agg_in, agg_out, n_agg, agg);
        run<2>("dword x4; waitcnt; s_nop 0; v_pk_mul_f32", src, rows, ld, bad, s1, s2, agg_in, agg_out, n_agg, agg);
        run<3>("dword x4; waitcnt; v_mov_b32 v21,v21; v_pk_mul_f32", src, rows, ld, bad, s1, s2, agg_in, agg_out, n_agg, agg);
        run<4>("as the first, every lane active", src, rows, ld, bad, s1, s2, agg_in, agg_out, n_agg, agg);
        run<5>("dwordx4 x1; waitcnt; v_pk_mul_f32 x2", src, rows, ld, bad, s1, s2, agg_in, agg_out, n_agg, agg);
    }
    return 0;
}
