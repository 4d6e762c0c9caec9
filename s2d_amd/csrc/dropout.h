// Counter-based dropout masks (Philox4x32-10, Salmon et al. 2011; the generator torch's nn.Dropout draws from too).
// The mask of an [M, N] activation is a pure function of (seed, site, row, column): the forward applies it in the
// producing GEMM's epilogue and the backward regenerates it instead of storing it.
//   block (row, col / 16)  ->  Philox4x32-10(counter = (row, col / 16, site, 0), key = (seed_lo, seed_hi))  ->  128 bits
//   element col % 16 = e   ->  8 bits: byte (e & 3) of word (e >> 2);  kept iff  byte >= T,  T = round(p * 256)
// so P(drop) = T / 256 exactly (p quantised to 1 / 256: 77 / 256 = 0.3008 for the shipped p = 0.3) and kept elements are multiplied
// by 256 / (256 - T) = 1 / P(keep): inverted dropout, unbiased for the realised keep probability, as nn.Dropout is for its p
// (reference sites: model_training/mask2former/modeling/pixel_decoder/msdeformattn.py:101-125).
// Round 5: 8 bits per element instead of 16 -- one Philox call per 16 elements.  v_mad_u64_u32 is a quarter-rate instruction and the
// 20 of them per call were the largest single item of the one-launch FFN's vector issue (csrc/ffn.hip); the mask stream is this
// library's own definition (torch's is an implementation detail of its kernels), oracle/oracle_np.py:dropout_multipliers restates it.
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

__host__ __device__ __forceinline__ uint32_t s2d_dropout_thresh(float p)
{
    const float t = p * 256.0f + 0.5f;
    return t <= 0.f ? 0u : (t >= 255.f ? 255u : (uint32_t)t);
}
__host__ __device__ __forceinline__ float s2d_dropout_scale(uint32_t thresh) { return 256.0f / (float)(256u - thresh); }

// multipliers (0 or scale) of the 8 elements col8 * 8 .. col8 * 8 + 7 of a row: half (col8 & 1) of block (row, col8 >> 1)
__device__ __forceinline__ void s2d_dropout8(uint32_t row, uint32_t col8, uint32_t site, uint32_t k0, uint32_t k1, uint32_t thresh,
                                             float scale, float (&m)[8])
{
    uint32_t r[4];
    s2d_philox4x32_10(row, col8 >> 1, site, 0u, k0, k1, r);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const uint32_t w = (col8 & 1u) ? r[2 + (e >> 2)] : r[e >> 2];
        m[e] = ((w >> (8 * (e & 3))) & 0xFFu) >= thresh ? scale : 0.f;
    }
}
