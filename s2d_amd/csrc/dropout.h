// Counter-based dropout masks (Philox4x32-10, Salmon et al. 2011; the generator torch's nn.Dropout draws from too).
// The mask of an [M, N] activation is a pure function of (seed, site, row, column): the forward applies it in the
// producing GEMM's epilogue and the backward regenerates it instead of storing it.
//   block (row, col / 8)  ->  Philox4x32-10(counter = (row, col / 8, site, 0), key = (seed_lo, seed_hi))  ->  128 bits
//   element col % 8 = e   ->  16 bits: half (e & 1) of word (e >> 1);  kept iff  bits >= round(p * 65536)
// so P(keep) = 1 - round(p * 65536) / 65536 (p quantised to 1.5e-5); kept elements are multiplied by 1 / (1 - p) as
// nn.Dropout does (reference sites: model_training/mask2former/modeling/pixel_decoder/msdeformattn.py:101-125).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ void s2d_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                                  uint32_t (&out)[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// multipliers (0 or scale) of the 8 elements of block (row, col8)
__device__ __forceinline__ void s2d_dropout8(uint32_t row, uint32_t col8, uint32_t site, uint32_t k0, uint32_t k1, uint32_t thresh,
                                             float scale, float (&m)[8])
{
    uint32_t r[4];
    s2d_philox4x32_10(row, col8, site, 0u, k0, k1, r);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const uint32_t bits = (e & 1) ? (r[e >> 1] >> 16) : (r[e >> 1] & 0xFFFFu);
        m[e] = bits >= thresh ? scale : 0.f;
    }
}
