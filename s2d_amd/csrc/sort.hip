// rocPRIM device primitives used as they are (AMD's own header-only library), kept in one translation unit because the
// header is heavy: an exclusive prefix sum (rle.hip: string offsets) and the
// stable radix sort of (u32 key, u32 value) pairs: the matcher orders its sample points by the logit-map cell they fall
// in (matcher.hip).  rocPRIM's device radix sort is used as is (AMD's own primitive library, header-only): LSD radix,
// stable, deterministic.  Kept in its own translation unit because the header is heavy.
#include <cstring>
#include <rocprim/rocprim.hpp>
#include "common.h"

int s2d_radix_sort_pairs_u32(const unsigned int *keys_in, unsigned int *keys_out, const unsigned int *vals_in, unsigned int *vals_out,
                             size_t n, int end_bit, void *temp, size_t temp_bytes, hipStream_t stream)
{
    if (n == 0) return S2D_OK;
    size_t need = 0;
    if (rocprim::radix_sort_pairs(nullptr, need, keys_in, keys_out, vals_in, vals_out, n, 0, (unsigned int)end_bit, stream) != hipSuccess)
        return S2D_ERR_LAUNCH;
    if (need > temp_bytes) return S2D_ERR_ARG;
    if (rocprim::radix_sort_pairs(temp, need, keys_in, keys_out, vals_in, vals_out, n, 0, (unsigned int)end_bit, stream) != hipSuccess)
        return S2D_ERR_LAUNCH;
    return S2D_OK;
}

int s2d_exclusive_scan_i32(const int *in, int *out, size_t n, void *temp, size_t temp_bytes, hipStream_t stream)
{
    if (n == 0) return S2D_OK;
    size_t need = 0;
    if (rocprim::exclusive_scan(nullptr, need, in, out, 0, n, rocprim::plus<int>(), stream) != hipSuccess) return S2D_ERR_LAUNCH;
    if (need > temp_bytes) return S2D_ERR_ARG;
    if (rocprim::exclusive_scan(temp, need, in, out, 0, n, rocprim::plus<int>(), stream) != hipSuccess) return S2D_ERR_LAUNCH;
    return S2D_OK;
}

size_t s2d_exclusive_scan_i32_temp_bytes(size_t n)
{
    size_t need = 0;
    if (n == 0) return 0;
    if (rocprim::exclusive_scan(nullptr, need, (const int *)nullptr, (int *)nullptr, 0, n, rocprim::plus<int>(), (hipStream_t)0) != hipSuccess) return 0;
    return need;
}
