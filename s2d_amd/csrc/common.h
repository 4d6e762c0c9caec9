// Internal helpers shared by the gfx950 kernels of libs2d_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define S2D_OK 0
#define S2D_ERR_ARG (-1)
#define S2D_ERR_LAUNCH (-2)

#define S2D_CHECK_LAUNCH()                                  \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return S2D_ERR_LAUNCH;       \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies per DEVICE: a process that drives several GPUs must set it on each, so the
// launchers' "already set" state is a bit per device of the calling thread's current device (atomic: host threads may race to set it)
#include <atomic>
struct S2dDevOnce {
    std::atomic<unsigned long long> bits[4];
    bool done() const
    {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 256) return false;     // unknown device: set the attribute again (cheap)
        return (bits[d >> 6].load(std::memory_order_relaxed) >> (d & 63)) & 1ull;
    }
    void mark()
    {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 256) return;
        bits[d >> 6].fetch_or(1ull << (d & 63), std::memory_order_relaxed);
    }
};

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Zero `bytes` bytes with a kernel on `stream`.  Used instead of hipMemsetAsync: the
// runtime may run a memset on a different engine than the kernels around it, and with two streams sharing the GPU the
// kernel that followed such a memset was observed reading its predecessor's output before it was complete.  A fill
// kernel is an ordinary dispatch in the stream's queue.  (elem.hip)
int s2d_zero_async(void *p, size_t bytes, hipStream_t stream);

// stable LSD radix sort of (key, value) pairs on bits [0, end_bit) (sort.hip)
int s2d_radix_sort_pairs_u32(const unsigned int *keys_in, unsigned int *keys_out, const unsigned int *vals_in, unsigned int *vals_out,
                             size_t n, int end_bit, void *temp, size_t temp_bytes, hipStream_t stream);

// exclusive prefix sum of n ints (sort.hip, rocPRIM)
int s2d_exclusive_scan_i32(const int *in, int *out, size_t n, void *temp, size_t temp_bytes, hipStream_t stream);
size_t s2d_exclusive_scan_i32_temp_bytes(size_t n);

// 64-lane butterfly reductions
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
