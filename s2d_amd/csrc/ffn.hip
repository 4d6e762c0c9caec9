// The encoder layer's feed-forward block in ONE launch (split-fp16 x3 arithmetic of gemm_bf16.hip, fp32 in / fp32 out):
//
//     y = [LayerNorm2]( x + dropout3( W2 . dropout2( relu( W1 . x + b1 ) ) + b2 ) )        x = [LayerNorm1]( xin )
//
// (reference: model_training/mask2former/modeling/pixel_decoder/msdeformattn.py:116-131, forward_ffn + norm2; with LN1 = norm1 of the
// same layer folded into the prologue).  As two GEMM launches the 1024-wide hidden activation of 309 120 rows is 1.27 GB written and
// read back per layer (12 times per step); here it never leaves the registers of the wave that computes it.
//
// Work split.  A workgroup = 4 waves (one per SIMD, 512 registers each) = 128 rows; a wave owns 32 rows for the whole launch:
//   * the wave's 32 x 256 input tile lives in registers as the B operand of GEMM 1 (16 k-steps x fp16 hi / lo fragments = 128 VGPRs);
//   * the hidden layer is walked in chunks of 32 units.  GEMM 1 computes the chunk TRANSPOSED, H^T[32 units, 32 rows] = W1c . X^T
//     (A = weight fragments from LDS, B = the resident X fragments): the 32 x 32 accumulator then has a row of the activation on the
//     lane and 16 hidden units in its registers, which is exactly the B-operand layout of v_mfma_f32_32x32x16_f16 -- bias (accumulator
//     init), ReLU, the Philox mask and the fp16 hi / lo split happen in place and the result feeds GEMM 2 with no data movement;
//   * GEMM 2 accumulates Y^T[256, 32 rows] += W2c . H^T in 8 tiles x (main, cross) accumulators = 256 AGPRs;
//   * the epilogue (dropout3, residual, LayerNorm over the 256 outputs of a row = 128 values on the lane + 128 on lane ^ 32) runs on
//     the accumulators and stores 64 contiguous bytes per lane and tile.
// The row -> hidden-unit / output-column permutation inside a 32-row MFMA tile is chosen so that a lane's 16 accumulator registers are
// 16 CONSECUTIVE units / columns (one whole 16-column Philox block, 64 contiguous bytes): MFMA row 8g + 4h + i <-> 16h + 4g + i.
//
// Weights: s2d_ffn_pack_f16 writes both matrices once as an image of MFMA A-fragments in the order the kernel consumes them
// (per chunk: 16 k-steps x (hi, lo) of W1 | k-step 0: 8 tiles x (hi, lo) of W2 | k-step 1: the same; 1 KB = 64 lanes x 16 B per
// fragment), so a chunk goes global -> LDS by `buffer_load_dwordx4 ... lds` as 64 linear 1-KB pieces and every fragment read is one
// conflict-free ds_read_b128.  LDS: two 32-KB W1 buffers + two 16-KB buffers for each k-step of W2 + b1 (4 KB).
//
// Schedule of chunk p (one barrier per chunk, weights double-buffered): DMA of the next pieces; GEMM 1 of chunk p (48 MFMAs, the Philox
// rounds of its mask in their shadow); GEMM 2 k-step 1 of chunk p-1 (24 MFMAs, ReLU / mask / split of chunk p's k-step 0 in their
// shadow); GEMM 2 k-step 0 of chunk p (24 MFMAs, the same for its k-step 1).
#include "common.h"
#include "dropout.h"
#include <stdlib.h>
#include <type_traits>

#ifndef S2D_FFN_DBG
#define S2D_FFN_DBG 0
#endif
// where the epilogue takes the residual x (the FFN's input) from: 0 = reloaded from memory tile by tile (round 4), 1 = rebuilt from the
// resident fp16 hi / lo fragments (x = hi + lo / 2048, no memory traffic; the out_proj form then stores Xn only for callers that keep
// it), 2 = all 32 loads issued before the trailing pass.  Measured in profiles/r5_experiments/ffn_phases.txt.
#if !(S2D_FFN_DBG & 16)
#define FFN_STAMP(slot) do { } while (0)
#endif
#ifndef S2D_FFN_EPI
#define S2D_FFN_EPI 1
#endif
#ifndef S2D_FFN_POSTV
#define S2D_FFN_POSTV 0
#endif

#if S2D_FFN_DBG & 16
// diagnostic build only (scripts/mb_ffn_clock.py): s_memtime / s_memrealtime around the chunk loop of wave 0 of every workgroup, into
// a buffer of their own that nothing else reads -- the in-kernel clock is d(memtime) / d(memrealtime) x 100 MHz
// (MI355X_MICROARCH.md, DVFS give-back (6))
__device__ unsigned long long g_ffn_stamps[16 * 4096];      // per workgroup: entry, loop start (time, realtime), loop end (time, realtime), epilogue end, kernel end
// fine stamps (slots 7..15): 7 input + residual tiles landed (fragments formed), 8 out_proj phase done, 9 LN1 + fragments done (= just before the
// loop's barrier), 10 trailing pass done, 11 epilogue tile loop done, 13 / 14 / 15 projection part 4 / 8 / 12 done
#define FFN_STAMP(slot) do { if (tid == 0 && blockIdx.x < 4096) g_ffn_stamps[16 * blockIdx.x + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int s2d_ffn_dbg_stamps(unsigned long long *host_out)
{
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_ffn_stamps), sizeof(g_ffn_stamps)) == hipSuccess ? 0 : -1;
}
#endif

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int FC = 256;                     // model width (the register layout is built on it)
constexpr int FRAG = 1024;                  // bytes of one MFMA operand fragment (64 lanes x 16 B)
constexpr int PART = 32 * FRAG;             // 32 KB: 16 k-steps x (hi, lo)  |  8 tiles x 2 k-steps x (hi, lo)
constexpr int CHUNKB = 2 * PART;            // image bytes per chunk of 32 hidden units
constexpr int LDS_B1 = 2 * CHUNKB;          // byte offset of the bias copy (behind the two chunk buffers)
constexpr int FMAX = 2048;                  // hidden width limit of the bias copy
constexpr int LDS_PB = LDS_B1 + FMAX * 4;   // PRE: the output projection's bias (256 floats) behind it
constexpr int LDS_LN = LDS_PB + FC * 4;     // LayerNorm parameters: gamma1 | beta1 | gamma2 | beta2 (256 floats each).  Read from global they
                                            // were 16-B loads in the middle of the epilogue's store stream: every use waited `vmcnt(0)`, i.e.
                                            // for every store issued before it too (in-order counter) -- from LDS they wait on lgkmcnt only

// MFMA row rho of a 32-row tile <-> unit / column 16 h + 4 g + i   (rho = 8 g + 4 h + i): a lane half's 16 accumulator registers
// (reg = 4 g + i at rows 8 g + 4 h + i) are then the 16 consecutive units 16 h + reg
__host__ __device__ __forceinline__ int perm_row(int rho) { return 16 * ((rho >> 2) & 1) + 4 * (rho >> 3) + (rho & 3); }

__device__ __forceinline__ unsigned int pk_hi(float a, float b) { return __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_pkrtz(a, b)); }
__device__ __forceinline__ unsigned int pk_lo(float a, float b, unsigned int hi)
{
    // (a - h) * 2048 == fma(h, -2048, a * 2048) exactly (a - h is exact, the factor a power of two): one multiply and one
    // v_fma_mix_f32 (the fp16 operand converted inside the fma) per value instead of convert, subtract, multiply
    const h16x2 h = __builtin_bit_cast(h16x2, hi);
    return __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_pkrtz(__builtin_fmaf((float)h[0], -2048.f, a * 2048.f), __builtin_fmaf((float)h[1], -2048.f, b * 2048.f)));
}

// Register classes.  The 256 output accumulators must live in the accumulator half of the register file and everything else in the
// vector half (X fragments 128, chunk accumulators 32, activation / weight fragments 48, Philox state ...).  With every MFMA a builtin
// the allocator gives the chunk accumulators AGPRs and shuttles an output tile between the halves around every chunk (~100
// v_accvgpr moves per 96 MFMAs) or spills the X fragments.  What works: GEMM 1 (the chunk accumulators) as asm statements with a "+v"
// accumulator -- that pins am / ax to VGPRs -- and GEMM 2 on the builtin, whose accumulators then take exactly the 256 AGPRs.
// GEMM 2 must NOT be asm: an asm form with "+a" accumulators gave wrong low-order bits (~1e-4 of the output) in some output tiles for
// one of two orders of the same three products, deterministic per build, tiles changing with unrelated code motion; with the builtin
// (hipcc sees an MFMA and keeps its hazards) both orders are exact.  The cause inside the asm form was not isolated; the rules the asm
// form of GEMM 1 is written to, beyond what the compiler does for any asm operand (s_waitcnt for the LDS reads that feed it):
//   * an MFMA reading the accumulator the preceding MFMA wrote needs no wait states (same-size back-to-back accumulation);
//   * WAR on A / B: every MFMA lists the previous MFMA's operands as unused inputs, so their registers stay allocated until the next
//     MFMA has issued (the allocator otherwise hands a dead fragment's registers to the very next VALU instruction);
//   * the VALU readers of am / ax (the activation quarters) sit >= 4 MFMAs behind GEMM 1's last MFMA.
// tests/test_gpu_ffn.py checks every variant against a float64 oracle at 2e-5 of the output scale (a lost low-order product is 1e-4).
// lane * 16 recomputed on the spot (volatile: never hoisted, no live range); a free __device__ function: an asm with a "v" constraint in a
// lambda of the kernel body is checked by the HOST pass too, which then silently drops the kernel's host stub
__device__ __forceinline__ int lane16_asm()
{
    int v;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshlrev_b32 %0, 4, %0" : "=v"(v));
    return v;
}
struct MfmaPrev { f16x8 a, b; };
__device__ __forceinline__ void mfma_a(f32x16 &c, const f16x8 a, const f16x8 b) { c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ void mfma_v(f32x16 &c, const f16x8 a, const f16x8 b, MfmaPrev &pv)
{
    asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b), "v"(pv.a), "v"(pv.b));
    pv.a = a; pv.b = b;
}
__device__ __forceinline__ void mfma_v0(f32x16 &c, const f16x8 a, const f16x8 b, MfmaPrev &pv)       // the accumulator's first product: C = 0
{
    asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b), "v"(pv.a), "v"(pv.b));
    pv.a = a; pv.b = b;
}

// The 256 input columns of a "part" (16 k-steps x (hi, lo) A-fragments of 32 weight rows) are taken in the order the B operand holds
// them when it comes out of an accumulator tile set: k-step ks = 2 t + s, lane half h, element j  <->  column 32 t + 16 h + 8 s + j
// (a lane's accumulator registers are columns 32 t + 16 h + 0..15; the input tile is loaded from memory in that layout too).
__host__ __device__ __forceinline__ int kcol(int ks, int h) { return 32 * (ks >> 1) + 16 * h + 8 * (ks & 1); }

// one thread = one lane's 16 bytes of one fragment.  Image: [F / 32 chunks x 64 fragments: W1 part | W2 k-step 0 | W2 k-step 1]
// [npost / 32 parts x 32 fragments: the rows 32 j + perm_row(r) of Wpost, as a W1 part][npre / 32 parts: Wpre likewise]
__global__ __launch_bounds__(256) void ffn_pack_kernel(const float *__restrict__ W1, const float *__restrict__ W2, int F,
                                                       const float *__restrict__ Wpost, int npost, const float *__restrict__ Wpre, int npre,
                                                       u32x4 *__restrict__ out)
{
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const long nfrag = (long)(F / 32) * 64 + (long)((npost + npre) / 32) * 32;
    if (gid >= nfrag * 64) return;
    const int lane = (int)(gid & 63);
    const long fidx = gid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const float *src;
    int lo;
    if (fidx < (long)(F / 32) * 64) {
        const int frag = (int)(fidx & 63);
        const long c = fidx >> 6;
        lo = frag & 1;
        if (frag < 32) src = W1 + (32 * c + perm_row(r)) * FC + kcol(frag >> 1, h);                            // k-step frag >> 1 of GEMM 1
        else {
            const int f2 = frag - 32, s = f2 >> 4, t = (f2 >> 1) & 7;
            src = W2 + (long)(32 * t + perm_row(r)) * F + 32 * c + 16 * h + 8 * s;                               // k-step s, tile t of GEMM 2
        }
    } else {
        const long f2 = fidx - (long)(F / 32) * 64;
        const int frag = (int)(f2 & 31);
        lo = frag & 1;
        const long part = f2 >> 5;
        src = (part < npost / 32 ? Wpost + (32 * part + perm_row(r)) * FC : Wpre + (32 * (part - npost / 32) + perm_row(r)) * FC) + kcol(frag >> 1, h);
    }
    u32x4 w;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned int hi = pk_hi(src[2 * q], src[2 * q + 1]);
        w[q] = lo ? pk_lo(src[2 * q], src[2 * q + 1], hi) : hi;
    }
    out[gid] = w;
}

struct FfnParams {
    const float *X;             // [M, 256] input rows (LayerNorm1 applied on the fly when g1 != NULL)
    float *Y;                   // [M, 256]
    float *Xn;                  // optional [M, 256]: the normalised input (g1 != NULL), for callers that keep it
    int M, nchunks;
    const unsigned int *pack;
    const float *b1, *b2;
    const float *g1, *be1;      // LayerNorm on the input (NULL: none)
    const float *g2, *be2;      // LayerNorm on the output (NULL: none)
    float eps;
    unsigned int thresh;        // dropout: element kept iff its 8 bits >= thresh (csrc/dropout.h); 0 = no dropout
    float dscale;
    unsigned int k0, k1, site_h, site_o, row0;
    // POST: the next encoder layer's merged projection of the output row, out_post[row][n] = y . Wpost[n]^T + post_bias[n]
    // (+ post_pos[row % post_S][n] for n < post_npos: the row-periodic (pos . W^T + b) term of the sampling offsets / attention logits)
    const float *post_bias, *post_pos;
    const float *pre_bias, *pre_res;     // PRE: out_proj bias [256]; the layer input rows [M, 256] (the attention's residual)
    unsigned int site_pre;
    float *post_out;
    int post_parts, post_S, post_npos, post_ld, post_ldpos;
};

template <bool DROP, bool LN1, bool LN2, bool POST, bool PRE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void ffn_f16x3_kernel(FfnParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];      // [W1 buf 0 | W1 buf 1 | W2 buf 0 | W2 buf 1 | b1]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#if S2D_FFN_DBG & 16
    const unsigned long long st_te = __builtin_amdgcn_s_memtime();
#endif
    const int tok = lane & 31, h = lane >> 5;
    const long row = (long)blockIdx.x * 128 + wave * 32 + tok;
    const bool rowok = row < p.M;
    const long rowc = rowok ? row : p.M - 1;
    const unsigned int mrow = p.row0 + (unsigned int)rowc;                   // mask row (lanes past M replicate row M - 1 exactly: their stores of
                                                                             // the projection phase are unconditional and must carry its values)

    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned int *>(p.pack), 0, p.nchunks * CHUNKB + (p.post_parts + (PRE ? 8 : 0)) * PART, 0x00020000);
    // The LDS-DMA's per-lane source offset, lane * 16.  Held in one register for the whole launch it is the value the allocator spills
    // (the chunk loop fills all 512 registers), and a spilled operand of a DMA instruction is reloaded in front of EVERY piece with
    // `scratch_load; s_waitcnt vmcnt(0)` -- vmcnt(0) drains all pieces in flight, one memory round trip per piece, in exactly the phases
    // (out_proj, projection) that have no other work to hide it.  So outside the chunk loop it is recomputed where it is used (3 VALU,
    // volatile: not hoisted, no live range; lane16_asm, a free __device__ function -- called through a lambda of the kernel body the host pass
    // silently dropped the kernel's stub); the chunk loop keeps its own copy made just in front of it.
    // a wave copies pieces wave * n .. wave * n + n - 1 of a part of 4 n pieces (n = 8: 32 KB; n = 4: one k-step of W2, 16 KB)
    auto dma_part = [&](int src_byte, int dst_byte, auto n_) {
        constexpr int n = decltype(n_)::value;
        const int voff = lane16_asm();
#pragma unroll
        for (int i = 0; i < n; ++i) {
            const int piece = wave * n + i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void *)(lds + dst_byte + piece * FRAG), 16,
                                                     voff, src_byte + piece * FRAG, 0, 0);
        }
    };
    // ---- helpers shared by the phases of the launch ----
    uint32_t rb[4], rt = 0u;            // Philox state / result of the chunk's 16-unit mask block (one byte per unit); rt: the c0 a half-done round holds back
    f16x8 fr[4][2];                     // weight fragment ring [slot][hi / lo]: a chunk is 32 MFMA groups (16 + 8 + 8), group g uses slot g & 3 and
                                        // requests group g + 2's pair first thing (one group of lead exposed ~30 cycles of LDS latency per group)
    const unsigned char *lane_lds = lds + lane * 16;
    constexpr int dbg = S2D_FFN_DBG;    // compile-time timing experiments (results are wrong with any of bits 1-8 set): 1 no DMA, 2 no barrier, 4 no activation work, 8 no fragment reads; 16: clock stamps (results unchanged); 32 no projection-phase stores, 64 no residual loads in the epilogue, 128 no Y / Xn stores, 256 no residual loads in the out_proj phase, 512 no input tile loads

    // Philox4x32-10 of the lane's mask block, round r, in two halves (idx = 2 r + half):
    //   half 0: M1 * c2 -> rt = hi ^ c1 ^ k0, c1 = lo;   half 1: M0 * c0 -> c2 = hi ^ c3 ^ k1, c3 = lo, c0 = rt
    auto philox_half = [&](int idx) {
        const int r = idx >> 1;
        if (!(idx & 1)) {
            const uint64_t pr = (uint64_t)0xCD9E8D57u * rb[2];
            rt = (uint32_t)(pr >> 32) ^ rb[1] ^ (p.k0 + 0x9E3779B9u * (uint32_t)r);
            rb[1] = (uint32_t)pr;
        } else {
            const uint64_t pr = (uint64_t)0xD2511F53u * rb[0];
            rb[2] = (uint32_t)(pr >> 32) ^ rb[3] ^ (p.k1 + 0xBB67AE85u * (uint32_t)r);
            rb[3] = (uint32_t)pr;
            rb[0] = rt;
        }
    };
    // the 20 half-rounds of a part / chunk go into every other MFMA gap of its first 40 (gap g = 3 ks + 0..2)
    auto philox_gap = [&](int g) {
        if (g < 40 && !(g & 1)) philox_half(g >> 1);
    };
    auto dma_piece = [&](int src_byte, int dst_byte) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void *)(lds + dst_byte), 16, lane16_asm(), src_byte, 0, 0);
    };
    // piece `sub` (0..3) of a run of four consecutive 1-KB pieces: the instruction's immediate offset advances the source and the LDS
    // address alike, so the four share one M0 value and one scalar offset (two scalar instructions saved per piece)
    int voff_loop = 0;                  // the chunk loop's copy of lane * 16 (set just in front of the loop)
    auto dma_piece4 = [&](int src_byte, int dst_byte, auto sub_, bool in_loop = false) {
        constexpr int SUB = decltype(sub_)::value;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void *)(lds + dst_byte), 16, in_loop ? voff_loop : lane16_asm(), src_byte, SUB * FRAG, 0);
    };
    auto frag_read = [&](int slot, const unsigned char *at) {
        if (dbg & 8) return;
        fr[slot][0] = *reinterpret_cast<const f16x8 *>(at);
        fr[slot][1] = *reinterpret_cast<const f16x8 *>(at + FRAG);
    };

    // phase 0's weights first (they are the longest wait of the prologue), then the bias copy and the input tile
    const int pre_base = p.nchunks * CHUNKB + p.post_parts * PART;           // the image: [chunks][projection parts][out_proj parts]
    auto chunk_loop_prologue = [&](auto w1_) {
        if (decltype(w1_)::value) dma_part(0, 0, std::integral_constant<int, 8>{});                       // W1 of chunk 0
        dma_part(PART, 2 * PART, std::integral_constant<int, 4>{});              // W2 k-step 0 of chunk 0
        if (decltype(w1_)::value) dma_part((p.nchunks > 1 ? 1 : 0) * CHUNKB, PART, std::integral_constant<int, 8>{});    // W1 of chunk 1
        for (int i = tid; i < PART / 2 / 16; i += 256)            // W2 k-step 1 buffer 1: what chunk 0's (empty) pass A multiplies by zero
            reinterpret_cast<u32x4 *>(lds + 3 * PART + PART / 2)[i] = u32x4{0u, 0u, 0u, 0u};
    };
    if (PRE) {                                                               // the out_proj phase walks the whole weight area first
        dma_part(pre_base, 0, std::integral_constant<int, 8>{});
        dma_part(pre_base + PART, PART, std::integral_constant<int, 8>{});
    } else chunk_loop_prologue(std::true_type{});
    // The small parameter vectors (b1, the out_proj bias, the LayerNorm parameters) go to LDS by LDS-DMA as well, 1 KB per instruction,
    // spread over the waves: loaded through registers each needed its own wait in front of its ds_write -- memory round trips in series IN
    // FRONT of the input tile's loads (the launch's longest wait).  They are visible behind the first `vmcnt(0)` + barrier, like the weights.
    {
        auto dma_vec = [&](const float *src, int nbytes, int dst_byte, int first) {          // piece k of the vector by wave (first + k) % 4
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, nbytes, 0x00020000);
            for (int k = 0; k * 1024 < nbytes; ++k)
                if (((first + k) & 3) == wave)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(lds + dst_byte + k * 1024), 16, lane16_asm(), k * 1024, 0, 0);
        };
        dma_vec(p.b1, p.nchunks * 32 * 4, LDS_B1, 0);
        if (PRE) dma_vec(p.pre_bias, FC * 4, LDS_PB, 1);
        if (LN1) { dma_vec(p.g1, FC * 4, LDS_LN, 2); dma_vec(p.be1, FC * 4, LDS_LN + FC * 4, 3); }
        if (LN2) { dma_vec(p.g2, FC * 4, LDS_LN + 2 * FC * 4, 0); dma_vec(p.be2, FC * 4, LDS_LN + 3 * FC * 4, 1); }
    }

    // ---- input tile -> fp16 hi / lo B fragments of GEMM 1.  The row is loaded in the accumulator layout (this lane: columns
    // 32 t + 16 h + 0..15 for t = 0..7, lane ^ 32 the other halves) and parked in the output accumulators, which are idle until the
    // chunk loop: LayerNorm1 needs the whole row before any fragment can be formed, and 128 parked values + 128 fragment registers do
    // not fit the vector half.  Fragment (k-step 2 t + s) = columns 32 t + 16 h + 8 s + 0..7: kcol(), the order the weight image uses.
    f32x16 ym[8], yx[8];
    f16x8 xh[16], xl[16];
    float mean1 = 0.f, rstd1 = 1.f;
    auto tile_to_frags = [&](const f32x16 (&src)[8]) {        // a row tile (fp32, accumulator layout) -> xh / xl
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                u32x4 hi, lo;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float a = src[t][8 * s2 + 2 * q], b = src[t][8 * s2 + 2 * q + 1];
                    hi[q] = pk_hi(a, b);
                    lo[q] = pk_lo(a, b, hi[q]);
                }
                xh[2 * t + s2] = __builtin_bit_cast(f16x8, hi);
                xl[2 * t + s2] = __builtin_bit_cast(f16x8, lo);
            }
    };
    {
        const float *xr = p.X + rowc * FC + 16 * h;
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 a = (dbg & 512) ? f32x4{(float)lane, 1.f, (float)t, (float)q} : *reinterpret_cast<const f32x4 *>(xr + 32 * t + 4 * q);
                f32x16 &dst = PRE ? yx[t] : ym[t];      // PRE: the sampled values park in yx, the residual tile goes to ym, where x1 is formed in place
                dst[4 * q] = a[0]; dst[4 * q + 1] = a[1]; dst[4 * q + 2] = a[2]; dst[4 * q + 3] = a[3];
            }
        if constexpr (PRE) {
            // ---- the attention's output projection in front of everything: x1 = res + dropout1( Wo . samp + bo ), tile by tile into the
            // parked row (ms_deform_attn.py:124 output_proj, msdeformattn.py:125 dropout1 + residual).  Same part machinery as the
            // projection phase at the end (four-buffer ring, pieces two parts ahead, builtin MFMAs on two accumulator pairs, the
            // previous tile's epilogue in the gaps), with the part index a compile-time constant: a tile's values go into ym[tile].
            // The residual's WHOLE row tile is loaded here, in flight together with the sampled values' tile (parked in the cross
            // accumulators yx, dead once the fragments are formed), into ym, where tile T's x1 = res + drop(...) is then formed IN PLACE:
            // one exposed memory round trip for both tiles instead of one per part -- loaded "a part ahead" each part still waited ~2 us for
            // its tile, 16 of the phase's 56 us (profiles/r5_experiments/ffn_phases.txt; one wave per SIMD: nothing hides it)
            const float *resr = p.pre_res + rowc * FC + 16 * h;
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 a = (dbg & 256) ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4 *>(resr + 32 * t + 4 * q);
                    ym[t][4 * q] = a[0]; ym[t][4 * q + 1] = a[1]; ym[t][4 * q + 2] = a[2]; ym[t][4 * q + 3] = a[3];
                }
            __builtin_amdgcn_sched_barrier(0);                      // both tiles' loads are issued before the first is waited for
            tile_to_frags(yx);                                      // the sampled values as B fragments
            FFN_STAMP(7);
            const unsigned char *pbo = lds + LDS_PB + 64 * h;      // the projection's bias, copied to LDS by the prologue
            f32x16 qm[2], qx[2], zero16, binit;
#pragma unroll
            for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
            uint32_t mk[4];                                         // the finished mask words of the previous tile
            auto bias_init = [&](int t) {                           // a lane's accumulator registers are 16 consecutive columns: C = bias
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(pbo + 128 * t + 16 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) binit[4 * q + e] = v[e];
                }
            };
            auto pre_out = [&](const f32x16 &m, const f32x16 &x, auto t_, int q) {     // quarter q of tile T -> ym[T][4 q ..]
                constexpr int T = decltype(t_)::value;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = __builtin_fmaf(x[4 * q + e], 1.0f / 2048.0f, m[4 * q + e]);
                    if (DROP) v *= ((mk[q] >> (8 * e)) & 0xFFu) >= p.thresh ? p.dscale : 0.f;     // element 4 q + e of the tile's 16: byte e of word q
                    ym[T][4 * q + e] += v;
                }
            };
            auto pre_part = [&](auto j_) {
                constexpr int J = decltype(j_)::value, CUR = J & 1;
                const unsigned char *w = lane_lds + (J & 3) * PART, *wn = lane_lds + ((J + 1) & 3) * PART;
                // two parts ahead; behind the last part come the chunk loop's first weights (W1 of chunks 0 and 1 live in ring buffers 0, 1)
                const int src2 = J + 2 < 8 ? pre_base + (J + 2) * PART : (J == 6 || p.nchunks < 2 ? 0 : CHUNKB), dst2 = ((J + 2) & 3) * PART;
                if (DROP) { rb[0] = mrow; rb[1] = (uint32_t)(2 * J + h); rb[2] = p.site_pre; rb[3] = 0u; }      // columns 32 J + 16 h + 0..15
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) {
                    const int sl = ks & 3, sn = (ks + 2) & 3;
                    if (ks == 14) {
                        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");      // this part's 8 pieces may stay in flight
                        __builtin_amdgcn_s_barrier();
                        __builtin_amdgcn_sched_barrier(0);
                        if (DROP) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) mk[c] = rb[c];
                        }
                        if (J < 7) bias_init(J + 1);
                    }
                    frag_read(sn, ks < 14 ? w + (2 * ks + 4) * FRAG : wn + (2 * (ks - 14)) * FRAG);
                    if (ks == 0) qm[CUR] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[sl][0], xh[ks], binit, 0, 0, 0); else mfma_a(qm[CUR], fr[sl][0], xh[ks]);
                    if (DROP) philox_gap(3 * ks);
                    __builtin_amdgcn_sched_barrier(0);
                    if (ks == 0) qx[CUR] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[sl][0], xl[ks], zero16, 0, 0, 0); else mfma_a(qx[CUR], fr[sl][0], xl[ks]);
                    if (ks < 8) {
                        const int s4 = src2 + (wave * 8 + (ks & 4)) * FRAG, d4 = dst2 + (wave * 8 + (ks & 4)) * FRAG;
                        switch (ks & 3) {
                        case 0: dma_piece4(s4, d4, std::integral_constant<int, 0>{}); break;
                        case 1: dma_piece4(s4, d4, std::integral_constant<int, 1>{}); break;
                        case 2: dma_piece4(s4, d4, std::integral_constant<int, 2>{}); break;
                        default: dma_piece4(s4, d4, std::integral_constant<int, 3>{}); break;
                        }
                    }
                    if (DROP) philox_gap(3 * ks + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_a(qx[CUR], fr[sl][1], xh[ks]);
                    if (J > 0 && ks >= 6 && ks < 14 && !(ks & 1)) pre_out(qm[CUR ^ 1], qx[CUR ^ 1], std::integral_constant<int, (J > 0 ? J - 1 : 0)>{}, (ks - 6) >> 1);
                    if (DROP) philox_gap(3 * ks + 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // parts 0, 1 (and the sampled tile's loads)
            __syncthreads();
            frag_read(0, lane_lds);
            frag_read(1, lane_lds + 2 * FRAG);
            bias_init(0);
            pre_part(std::integral_constant<int, 0>{}); pre_part(std::integral_constant<int, 1>{});
            pre_part(std::integral_constant<int, 2>{}); pre_part(std::integral_constant<int, 3>{});
            pre_part(std::integral_constant<int, 4>{}); pre_part(std::integral_constant<int, 5>{});
            pre_part(std::integral_constant<int, 6>{}); pre_part(std::integral_constant<int, 7>{});
#pragma unroll
            for (int q = 0; q < 4; ++q) pre_out(qm[1], qx[1], std::integral_constant<int, 7>{}, q);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();                                        // every wave is done with the ring: the rest of the chunk loop's start
            FFN_STAMP(8);
            chunk_loop_prologue(std::false_type{});
        }
        if (LN1) {
            if (!PRE) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
            float s = 0.f;
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += ym[t][r];
            s += __shfl_xor(s, 32, 64);
            mean1 = s / (float)FC;
            float ss = 0.f;
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float d = ym[t][r] - mean1; ss += d * d; }
            ss += __shfl_xor(ss, 32, 64);
            rstd1 = 1.f / sqrtf(ss / (float)FC + p.eps);
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float *lnp = reinterpret_cast<const float *>(lds + LDS_LN) + 32 * t + 16 * h + 4 * q;
                    const f32x4 ga = *reinterpret_cast<const f32x4 *>(lnp), be = *reinterpret_cast<const f32x4 *>(lnp + FC);
                    f32x4 v = {ym[t][4 * q], ym[t][4 * q + 1], ym[t][4 * q + 2], ym[t][4 * q + 3]};
                    v = (v - mean1) * rstd1 * ga + be;
                    ym[t][4 * q] = v[0]; ym[t][4 * q + 1] = v[1]; ym[t][4 * q + 2] = v[2]; ym[t][4 * q + 3] = v[3];
                    if (((PRE && S2D_FFN_EPI != 1) || (p.Xn && rowok)) && !(dbg & 128)) *reinterpret_cast<f32x4 *>(p.Xn + rowc * FC + 32 * t + 16 * h + 4 * q) = v;     // PRE, EPI != 1: the epilogue's residual
                }
        }
        tile_to_frags(ym);
        FFN_STAMP(9);
    }

    // ---- output accumulators: tile t, register reg <-> column 32 t + 16 h + reg of the wave's rows; bias b2 as the initial value ----
#pragma unroll
    for (int t = 0; t < 8; ++t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b = *reinterpret_cast<const f32x4 *>(p.b2 + 32 * t + 16 * h + 4 * q);
            ym[t][4 * q] = b[0]; ym[t][4 * q + 1] = b[1]; ym[t][4 * q + 2] = b[2]; ym[t][4 * q + 3] = b[3];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) yx[t][r] = 0.f;
    }

    f32x16 am, ax;                      // GEMM 1 accumulators of the current chunk
    u32x4 hw[2], lw[2];                 // fp16 hi / lo B fragments (k-steps 0, 1) of the chunk's activation, as packed words
    float hv0 = 0.f, hv1 = 0.f;         // the activation word being formed
    MfmaPrev pv;                        // the previous asm MFMA's operands (see mfma_v)
    pv.a = xh[0]; pv.b = xh[0];
    unsigned int hhi = 0u;

    // ---- filler work, cut into units of at most ~6 vector instructions: one unit per MFMA gap (a lone wave hides ~5 beside an MFMA) ----
    // activation word q = idx >> 2 of k-step s (units 16 h + 8 s + 2 q, + 1) in four quarters: combine + ReLU, mask, fp16 hi, scaled fp16 lo
    auto hquarter = [&](int s, int idx) {
        const int q = idx >> 2, part = idx & 3, r0 = 8 * s + 2 * q;
        if (part == 0) {
            hv0 = fmaxf(__builtin_fmaf(ax[r0], 1.0f / 2048.0f, am[r0]), 0.f);
            hv1 = fmaxf(__builtin_fmaf(ax[r0 + 1], 1.0f / 2048.0f, am[r0 + 1]), 0.f);
        } else if (part == 1) {
            if (DROP) {
                const uint32_t w = rb[2 * s + (q >> 1)];                // units 8 s + 2 q, + 1 of the lane's 16: bytes 2 (q & 1), + 1 of word 2 s + (q >> 1)
                hv0 *= ((w >> (16 * (q & 1))) & 0xFFu) >= p.thresh ? p.dscale : 0.f;      // a multiplier (select of two constants), not a select of the products:
                hv1 *= ((w >> (16 * (q & 1) + 8)) & 0xFFu) >= p.thresh ? p.dscale : 0.f;  // the latter compiles to exec-masked blocks that cut the schedule
            }
        } else if (part == 2) {
            hhi = pk_hi(hv0, hv1);
            hw[s][q] = hhi;
        } else {
            lw[s][q] = pk_lo(hv0, hv1, hhi);
        }
    };
    // the 16 pieces a wave copies per chunk, all behind the chunk's barrier (which frees their destinations):
    //   k 0..7  W1 of chunk pc + 2 -> W1 buffer pc & 1 (GEMM 1 of chunk pc has read it);  k 8..11  W2 k-step 0 of chunk pc + 1 -> S0 buffer
    //   (pc + 1) & 1;  k 12..15  W2 k-step 1 of chunk pc -> S1 buffer pc & 1 (read by the next chunk's pass A).  Past the last chunk the
    //   source is clamped (a copy nobody reads) so that the instruction stream has no branches.
    auto dma_k = [&](int k, int pc) {
        const int cur = pc & 1, last = p.nchunks - 1;
        if (dbg & 1) return;
        auto go = [&](int src, int dst) {
            switch (k & 3) {
            case 0: dma_piece4(src, dst, std::integral_constant<int, 0>{}, true); break;
            case 1: dma_piece4(src, dst, std::integral_constant<int, 1>{}, true); break;
            case 2: dma_piece4(src, dst, std::integral_constant<int, 2>{}, true); break;
            default: dma_piece4(src, dst, std::integral_constant<int, 3>{}, true); break;
            }
        };
        if (k < 8) go(min(pc + 2, last) * CHUNKB + (wave * 8 + (k & 4)) * FRAG, cur * PART + (wave * 8 + (k & 4)) * FRAG);
        else if (k < 12) go(min(pc + 1, last) * CHUNKB + PART + wave * 4 * FRAG, 2 * PART + (cur ^ 1) * (PART / 2) + wave * 4 * FRAG);
        else go(pc * CHUNKB + PART + PART / 2 + wave * 4 * FRAG, 3 * PART + cur * (PART / 2) + wave * 4 * FRAG);
    };
    auto bias_read = [&](int pc) {      // am := b1 of chunk pc's units 16 h + 0..15 (the accumulator's initial value)
        const float *bp = reinterpret_cast<const float *>(lds + LDS_B1) + 32 * pc + 16 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b = *reinterpret_cast<const f32x4 *>(bp + 4 * q);
            am[4 * q] = b[0]; am[4 * q + 1] = b[1]; am[4 * q + 2] = b[2]; am[4 * q + 3] = b[3];
        }
    };
    // post-barrier gap j (0..53: GEMM 1's last two k-steps, pass A, pass B): DMA piece k at j = 3 k + 1
    auto post_gap = [&](int j, int pc) {
        if (j % 3 == 1 && j / 3 < 16) dma_k(j / 3, pc);
    };
    // GEMM 2, k-step S of a chunk over the 8 output tiles (fragment (tile t, hi / lo) at wf + (t * 2 + hi/lo) * FRAG; tile 0's pair is
    // already in ring slot 0), with the activation quarters of the OTHER k-step of the current chunk as filler: pass A (S = 1, previous
    // chunk) forms hw[0] / lw[0], pass B (S = 0, current chunk) reads them and forms hw[1] / lw[1], which the next chunk's pass A
    // reads.  `next`: where ring slot 0 is refilled from behind tile 7 (the next MFMA group's first fragment pair).
    auto gemm2_pass = [&](auto s_, const unsigned char *wf, const unsigned char *next, int j0, int pc, int first_q, int bias_pc) {
        constexpr int S = decltype(s_)::value;
        const f16x8 bh = __builtin_bit_cast(f16x8, hw[S]), bl = __builtin_bit_cast(f16x8, lw[S]);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int sl = t & 3, sn = (t + 2) & 3;                                 // both passes start at a group index that is a multiple of 4
            frag_read(sn, t + 2 < 8 ? wf + (2 * t + 4) * FRAG : next + (2 * (t + 2 - 8)) * FRAG);
            mfma_a(ym[t], fr[sl][0], bh);
            post_gap(j0 + 3 * t, pc);
            if (3 * t >= first_q && 3 * t < first_q + 16 && !(dbg & 4)) hquarter(S ^ 1, 3 * t - first_q);
            __builtin_amdgcn_sched_barrier(0);
            mfma_a(yx[t], fr[sl][0], bl);
            post_gap(j0 + 3 * t + 1, pc);
            if (3 * t + 1 >= first_q && 3 * t + 1 < first_q + 16 && !(dbg & 4)) hquarter(S ^ 1, 3 * t + 1 - first_q);
            __builtin_amdgcn_sched_barrier(0);
            mfma_a(yx[t], fr[sl][1], bh);
            post_gap(j0 + 3 * t + 2, pc);
            if (3 * t + 2 >= first_q && 3 * t + 2 < first_q + 16 && !(dbg & 4)) hquarter(S ^ 1, 3 * t + 2 - first_q);
            if (S == 0 && t == 6) bias_read(bias_pc);          // am is dead behind the last activation quarter (gap 17): the next chunk's initial value
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // Chunk pc:  GEMM 1 k-steps 0..13 (42 MFMAs; the 20 Philox half-rounds of its mask in every other gap)  ->  vmcnt(0), BARRIER (the pieces
    // issued a chunk ago are visible, the buffers of chunk pc - 1 and W1 of chunk pc are free)  ->  GEMM 1 k-steps 14, 15 (their fragments were
    // read before the barrier: 6 MFMAs that cover the latency of pass A's first fragment reads)  ->  pass A: GEMM 2 k-step 1 of chunk pc - 1 (24
    // MFMAs; chunk pc's first activation fragment formed in their gaps)  ->  pass B: GEMM 2 k-step 0 of chunk pc (24 MFMAs; its second
    // activation fragment).  The 16 DMA pieces of the chunk go out one per three MFMAs behind the barrier.  Deferring a chunk's second
    // k-step by one chunk is what hides the activation work: every MFMA of GEMM 2 needs the activation.
    // Live beside the 256 output accumulators: X fragments 128, am / ax 32, activation fragments 16, weight fragments 32.
    hw[1] = u32x4{0u, 0u, 0u, 0u}; lw[1] = hw[1];                // chunk 0's pass A: zero activation against the zeroed S1 buffer
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the prologue's DMA pieces (and the input tile's loads)
    __syncthreads();
    frag_read(0, lane_lds);
    frag_read(1, lane_lds + 2 * FRAG);
    bias_read(0);
    voff_loop = lane16_asm();
#if S2D_FFN_DBG & 16
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (int pc = 0; pc < p.nchunks; ++pc) {
        const int cur = pc & 1;
        const unsigned char *w1 = lane_lds + cur * PART;                          // W1 of chunk pc
        const unsigned char *w2a = lane_lds + 3 * PART + (cur ^ 1) * (PART / 2);   // W2 k-step 1 of chunk pc - 1
        const unsigned char *w2b = lane_lds + 2 * PART + cur * (PART / 2);         // W2 k-step 0 of chunk pc
        if (DROP) { rb[0] = mrow; rb[1] = (uint32_t)(2 * pc + h); rb[2] = p.site_h; rb[3] = 0u; }      // hidden units 32 pc + 16 h + 0..15
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const int sl = ks & 3, sn = (ks + 2) & 3;
            if (ks == 14) {
                // pass A's fragments (requested from here on) sit in a buffer this wave's pieces of a chunk ago went to: they must have
                // landed, in every wave.  GEMM 1's last two k-steps already have their fragments: 6 MFMAs to cover the first read's latency.
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                if (!(dbg & 2)) __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
            frag_read(sn, ks < 14 ? w1 + (2 * ks + 4) * FRAG : w2a + (2 * (ks - 14)) * FRAG);
            mfma_v(am, fr[sl][0], xh[ks], pv);
            if (ks >= 14) post_gap(3 * (ks - 14), pc);
            if (DROP) philox_gap(3 * ks);
            __builtin_amdgcn_sched_barrier(0);
            if (ks == 0) mfma_v0(ax, fr[sl][0], xl[ks], pv); else mfma_v(ax, fr[sl][0], xl[ks], pv);
            if (ks >= 14) post_gap(3 * (ks - 14) + 1, pc);
            if (DROP) philox_gap(3 * ks + 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_v(ax, fr[sl][1], xh[ks], pv);
            if (ks >= 14) post_gap(3 * (ks - 14) + 2, pc);
            if (DROP) philox_gap(3 * ks + 2);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ring slots 0, 1 hold pass A's first two fragment pairs (requested behind the barrier)
        gemm2_pass(std::integral_constant<int, 1>{}, w2a, w2b, 6, pc, 6, -1);
        gemm2_pass(std::integral_constant<int, 0>{}, w2b, lane_lds + (cur ^ 1) * PART, 30, pc, 2, pc + 1 < p.nchunks ? pc + 1 : pc);
    }
#if S2D_FFN_DBG & 16
    {
        const unsigned long long st_t1 = __builtin_amdgcn_s_memtime(), st_r1 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0 && blockIdx.x < 4096) {
            g_ffn_stamps[16 * blockIdx.x] = st_te;
            g_ffn_stamps[16 * blockIdx.x + 1] = st_t0; g_ffn_stamps[16 * blockIdx.x + 2] = st_r0;
            g_ffn_stamps[16 * blockIdx.x + 3] = st_t1; g_ffn_stamps[16 * blockIdx.x + 4] = st_r1;
        }
    }
#endif
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();                                            // the last chunk's W2 k-step 1 pieces, copied by all four waves
    const int post_base = p.nchunks * CHUNKB;
    if (POST) {
        // both W1 buffers are free: the projection's parts 0 and 1 go out now and land behind the last pass and the epilogue
#pragma unroll
        for (int k = 0; k < 8; ++k) dma_piece(post_base + (wave * 8 + k) * FRAG, (wave * 8 + k) * FRAG);
#pragma unroll
        for (int k = 0; k < 8; ++k) dma_piece(post_base + (p.post_parts > 1 ? PART : 0) + (wave * 8 + k) * FRAG, PART + (wave * 8 + k) * FRAG);
    }
    // The epilogue's residual row tile (the FFN's input x: this lane's 128 values), ALL 32 loads issued here, into the registers the
    // input fragments have just left: one memory round trip, covered by the trailing pass below.  Round 4 loaded it tile by tile inside
    // the epilogue (register fear: the fragments are dead by then) -- eight dependent round trips with nothing to hide them at one wave
    // per SIMD, and each wait also drained the stores issued before it (vmcnt is in order).
    f32x4 resv[8][4];
    const float *rsrc = (PRE ? p.Xn : p.X) + rowc * FC + 16 * h;        // PRE: the normalised x1 this lane stored behind the out_proj phase
    if (S2D_FFN_EPI == 2) {
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) resv[t][q] = (dbg & 64) ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4 *>(rsrc + 32 * t + 4 * q);
        __builtin_amdgcn_sched_barrier(0);
    }
    frag_read(0, lane_lds + 3 * PART + ((p.nchunks - 1) & 1) * (PART / 2));
    frag_read(1, lane_lds + 3 * PART + ((p.nchunks - 1) & 1) * (PART / 2) + 2 * FRAG);
    gemm2_pass(std::integral_constant<int, 1>{}, lane_lds + 3 * PART + ((p.nchunks - 1) & 1) * (PART / 2), lane_lds, 1000, 0, 1000, -1);   // no fillers

    FFN_STAMP(10);
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs' results are read by the VALU below (see mfma_a)
    // ---- epilogue: y = [LN2]( x + dropout3(acc) ),  this lane: columns 32 t + 16 h + 0..15 of its row ----
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        uint32_t m0[4];
        if (DROP) s2d_philox4x32_10(mrow, (uint32_t)(2 * t + h), p.site_o, 0u, p.k0, p.k1, m0);      // columns 32 t + 16 h + 0..15
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 x;
            if (S2D_FFN_EPI == 2) x = resv[t][q];
            else if (S2D_FFN_EPI == 0) x = (dbg & 64) ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4 *>(rsrc + 32 * t + 4 * q);
            else {
                // fragment (k-step 2 t + s, element j) IS column 32 t + 16 h + 8 s + j of the lane's row: the accumulator layout
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * q + e;
                    x[e] = (dbg & 64) ? 0.f : __builtin_fmaf((float)xl[2 * t + (r >> 3)][r & 7], 1.0f / 2048.0f, (float)xh[2 * t + (r >> 3)][r & 7]);
                }
            }
            if (LN1 && !PRE && S2D_FFN_EPI != 1) {
                const float *lnp = reinterpret_cast<const float *>(lds + LDS_LN) + 32 * t + 16 * h + 4 * q;
                const f32x4 ga = *reinterpret_cast<const f32x4 *>(lnp), be = *reinterpret_cast<const f32x4 *>(lnp + FC);
                x = (x - mean1) * rstd1 * ga + be;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * q + e;
                float v = ym[t][r] + yx[t][r] * (1.0f / 2048.0f);
                if (DROP) v = ((m0[q] >> (8 * e)) & 0xFFu) >= p.thresh ? v * p.dscale : 0.f;      // r = 4 q + e: byte e of word q
                v += x[e];
                ym[t][r] = v;
                s += v;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    FFN_STAMP(11);
    float mean = 0.f, rstd = 1.f;
    if (LN2) {
        s += __shfl_xor(s, 32, 64);
        mean = s / (float)FC;
        float ss = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float d = ym[t][r] - mean; ss += d * d; }
        ss += __shfl_xor(ss, 32, 64);
        rstd = 1.f / sqrtf(ss / (float)FC + p.eps);
    }
    float *yr = p.Y + rowc * FC + 16 * h;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v = {ym[t][4 * q], ym[t][4 * q + 1], ym[t][4 * q + 2], ym[t][4 * q + 3]};
            if (LN2) {
                const float *lnp = reinterpret_cast<const float *>(lds + LDS_LN) + 2 * FC + 32 * t + 16 * h + 4 * q;
                const f32x4 ga = *reinterpret_cast<const f32x4 *>(lnp), be = *reinterpret_cast<const f32x4 *>(lnp + FC);
                v = (v - mean) * rstd * ga + be;
            }
            if (rowok && !(dbg & 128)) *reinterpret_cast<f32x4 *>(yr + 32 * t + 4 * q) = v;
            if (POST) { ym[t][4 * q] = v[0]; ym[t][4 * q + 1] = v[1]; ym[t][4 * q + 2] = v[2]; ym[t][4 * q + 3] = v[3]; }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#if S2D_FFN_DBG & 16
    if (tid == 0 && blockIdx.x < 4096) g_ffn_stamps[16 * blockIdx.x + 5] = __builtin_amdgcn_s_memtime();
#endif
    if constexpr (POST) {
        // ---- the next layer's merged projection of the row just written: out_post[row][32 j + 16 h + reg] for the post_parts tiles j,
        // one weight part (32 KB, W1 buffers alternately) per tile, the row as the B fragments (same layout as the input tile's).
        // Accumulators: builtin MFMAs on two (main, cross) pairs, so that tile j - 1's epilogue -- 16 values per lane: combine, the
        // row-periodic pos term for the first post_npos columns, one 64-byte store -- runs in the gaps of tile j's MFMAs.
        tile_to_frags(ym);
        const long prow = rowc % p.post_S;
        const float *posr = p.post_pos + prow * p.post_ldpos + 16 * h;
        float *outr = p.post_out + rowc * p.post_ld + 16 * h;
        const float *pbr = p.post_bias + 16 * h;
        f32x16 pm[2], px[2], zero16;
#pragma unroll
        for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
        f32x4 pos4[4];
        // tile j's additive term: the row-periodic pos . W^T + b for the offsets / logits columns (their bias vector is zero), the bias
        // for the value columns.  Loaded a quarter part ahead of its use: as the accumulators' initial value it stalled every part's
        // first MFMA on an L2 round trip.
        auto add_load = [&](int j, bool haspos) {
#pragma unroll
            for (int q = 0; q < 4; ++q) pos4[q] = *reinterpret_cast<const f32x4 *>((haspos ? posr : pbr) + 32 * j + 4 * q);
        };
        auto tile_out = [&](const f32x16 &m, const f32x16 &x, int j, int q) {
            f32x4 v = {m[4 * q] + x[4 * q] * (1.0f / 2048.0f), m[4 * q + 1] + x[4 * q + 1] * (1.0f / 2048.0f),
                       m[4 * q + 2] + x[4 * q + 2] * (1.0f / 2048.0f), m[4 * q + 3] + x[4 * q + 3] * (1.0f / 2048.0f)};
            v += pos4[q];
            if (!(dbg & 32)) *reinterpret_cast<f32x4 *>(outr + 32 * j + 4 * q) = v;        // unconditional (rows past M rewrite row M - 1 with its own values): the wait below counts it
        };
        // part j (accumulator pair CUR = j & 1; PREV: 0 = no previous tile, 1 = the previous tile takes the pos term, 2 = it does not) reads
        // buffer j & 3 of a ring of FOUR 32-KB buffers (the whole weight area: the chunk loop is over).  While it runs, the 8 pieces of
        // part j + 2 go to buffer (j + 2) & 3 -- free since barrier j - 1, behind which every wave is done with part j - 2 -- so a piece has
        // more than a whole part to land before barrier j + 1 publishes it (with two buffers and one part of lead, every part waited for
        // its successor's pieces: 4 400 cycles per part instead of 1 536)
        auto post_part = [&](auto cur_, auto prev_, int j) {
            constexpr int CUR = decltype(cur_)::value, PREV = decltype(prev_)::value;
            const unsigned char *w = lane_lds + (j & 3) * PART, *wn = lane_lds + ((j + 1) & 3) * PART;
            const int last = p.post_parts - 1;
            const int src2 = post_base + min(j + 2, last) * PART, dst2 = ((j + 2) & 3) * PART;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const int sl = ks & 3, sn = (ks + 2) & 3;
                if (ks == 14) {
                    // part j + 1's pieces (issued during part j - 1) must have landed; what this part issued -- its 8 pieces, the previous
                    // tile's four additive-term loads and four stores -- are the youngest vector-memory operations and need not: vmcnt
                    // retires in order
                    // S2D_FFN_POSTV = 1: the part's pieces all go out in k-steps 0..3 (two per k-step), in front of every store of the part, so
                    // that the previous part's four stores are younger than the pieces waited for here as well: vmcnt(20)
                    if (PREV) { if (S2D_FFN_POSTV) asm volatile("s_waitcnt vmcnt(20) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory"); }
                    else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
                frag_read(sn, ks < 14 ? w + (2 * ks + 4) * FRAG : wn + (2 * (ks - 14)) * FRAG);
                if (ks == 0) pm[CUR] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[sl][0], xh[ks], zero16, 0, 0, 0); else mfma_a(pm[CUR], fr[sl][0], xh[ks]);
                if (ks == 0 && PREV) add_load(j - 1, PREV == 1);
                __builtin_amdgcn_sched_barrier(0);
                if (ks == 0) px[CUR] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[sl][0], xl[ks], zero16, 0, 0, 0); else mfma_a(px[CUR], fr[sl][0], xl[ks]);
                auto piece = [&](int k) {                                   // piece k (0..7) of part j + 2
                    const int s4 = src2 + (wave * 8 + (k & 4)) * FRAG, d4 = dst2 + (wave * 8 + (k & 4)) * FRAG;
                    switch (k & 3) {
                    case 0: dma_piece4(s4, d4, std::integral_constant<int, 0>{}); break;
                    case 1: dma_piece4(s4, d4, std::integral_constant<int, 1>{}); break;
                    case 2: dma_piece4(s4, d4, std::integral_constant<int, 2>{}); break;
                    default: dma_piece4(s4, d4, std::integral_constant<int, 3>{}); break;
                    }
                };
                if (S2D_FFN_POSTV) { if (ks < 4) piece(2 * ks); }
                else if (ks < 8) piece(ks);
                __builtin_amdgcn_sched_barrier(0);
                mfma_a(px[CUR], fr[sl][1], xh[ks]);
                if (S2D_FFN_POSTV && ks < 4) piece(2 * ks + 1);
                if (PREV && ks >= 6 && ks < 14 && !(ks & 1)) tile_out(pm[CUR ^ 1], px[CUR ^ 1], j - 1, (ks - 6) >> 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        typedef std::integral_constant<int, 0> I0;
        typedef std::integral_constant<int, 1> I1;
        typedef std::integral_constant<int, 2> I2;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // part 0 (and the head of part 1) have landed ...
        __syncthreads();                                        // ... in every wave
        frag_read(0, lane_lds);
        frag_read(1, lane_lds + 2 * FRAG);
        const int npt = p.post_npos >> 5;                       // tiles 0 .. npt - 1 take the pos term
        post_part(I0{}, I0{}, 0);
        for (int j = 1; j < p.post_parts; ++j) {
            if (j == 4) FFN_STAMP(13);
            if (j == 8) FFN_STAMP(14);
            if (j == 12) FFN_STAMP(15);
            const bool hp = j - 1 < npt;                        // wave-uniform: one branch per part, outside its schedule
            if (j & 1) { if (hp) post_part(I1{}, I1{}, j); else post_part(I1{}, I2{}, j); }
            else { if (hp) post_part(I0{}, I1{}, j); else post_part(I0{}, I2{}, j); }
        }
        {
            const int j = p.post_parts - 1;
            add_load(j, j < npt);
#pragma unroll
            for (int q = 0; q < 4; ++q) { if (j & 1) tile_out(pm[1], px[1], j, q); else tile_out(pm[0], px[0], j, q); }
        }
    }
#if S2D_FFN_DBG & 16
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the stores have left: the kernel-end stamp counts them
    if (tid == 0 && blockIdx.x < 4096) g_ffn_stamps[16 * blockIdx.x + 6] = __builtin_amdgcn_s_memtime();
#endif
}

}  // namespace

extern "C" {

long s2d_ffn_pack_words(int C, int F, int Npost, int pre)
{
    if (C != FC || F <= 0 || F % 32 || F > FMAX || Npost < 0 || Npost % 32 || Npost > 4096 || (pre != 0 && pre != 1)) return -1;
    return (long)(F / 32) * (CHUNKB / 4) + (long)(Npost / 32 + (pre ? FC / 32 : 0)) * (PART / 4);
}

int s2d_ffn_pack_f16(const float *W1, const float *W2, int C, int F, const float *Wpost, int Npost, const float *Wpre, void *out,
                     hipStream_t stream)
{
    if (s2d_ffn_pack_words(C, F, Npost, Wpre != nullptr) < 0 || !W1 || !W2 || !out || (Npost > 0 && !Wpost)) return S2D_ERR_ARG;
    const int npre = Wpre ? FC : 0;
    const long n = ((long)(F / 32) * 64 + (long)((Npost + npre) / 32) * 32) * 64;
    hipLaunchKernelGGL(ffn_pack_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, W1, W2, F, Wpost, Npost, Wpre, npre,
                       reinterpret_cast<u32x4 *>(out));
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

int s2d_ffn_fused_f32(const float *x, long M, int C, int F, const void *pack, const float *b1, const float *b2, const float *ln1_gamma,
                      const float *ln1_beta, const float *ln2_gamma, const float *ln2_beta, float eps, float p, uint64_t seed,
                      unsigned site_hidden, unsigned site_out, unsigned row0, float *xn, float *y, int Npost, const float *post_bias,
                      const float *post_pos, int post_S, int post_npos, long post_ldpos, float *post_out, long post_ld,
                      const float *pre_bias, const float *pre_res, unsigned site_pre, hipStream_t stream)
{
    if (s2d_ffn_pack_words(C, F, Npost, pre_bias != nullptr) < 0 || !x || !pack || !b1 || !b2 || !y || M <= 0 || M > 0x7FFFFF00L) return S2D_ERR_ARG;
    if ((ln1_gamma == nullptr) != (ln1_beta == nullptr) || (ln2_gamma == nullptr) != (ln2_beta == nullptr)) return S2D_ERR_ARG;
    for (const void *q16 : {(const void *)b1, (const void *)ln1_gamma, (const void *)ln1_beta, (const void *)ln2_gamma, (const void *)ln2_beta, (const void *)pre_bias})
        if (reinterpret_cast<uintptr_t>(q16) & 15) return S2D_ERR_ARG;      // the parameter vectors travel by 16-B LDS-DMA
    if (xn && !ln1_gamma) return S2D_ERR_ARG;
    if (!(p >= 0.f && p < 1.f)) return S2D_ERR_ARG;
    if (pre_bias && (!pre_res || (!xn && S2D_FFN_EPI != 1) || !ln1_gamma || !ln2_gamma)) return S2D_ERR_ARG;     // the out_proj phase exists for the encoder layer's form only
    if (Npost > 0 && (!post_bias || !post_out || post_ld < Npost || (post_ld & 3) || post_npos < 0 || post_npos > Npost || (post_npos & 31) ||
                      (post_npos > 0 && (!post_pos || post_S <= 0 || post_ldpos < post_npos || (post_ldpos & 3))) || M * post_ld > 0x7FFFFFFFL * 4))
        return S2D_ERR_ARG;
    FfnParams q;
    q.X = x; q.Y = y; q.Xn = xn; q.M = (int)M; q.nchunks = F / 32;
    q.pack = reinterpret_cast<const unsigned int *>(pack);
    q.b1 = b1; q.b2 = b2; q.g1 = ln1_gamma; q.be1 = ln1_beta; q.g2 = ln2_gamma; q.be2 = ln2_beta; q.eps = eps;
    q.thresh = s2d_dropout_thresh(p);
    q.dscale = q.thresh ? s2d_dropout_scale(q.thresh) : 1.f;
    q.k0 = (unsigned int)seed; q.k1 = (unsigned int)(seed >> 32); q.site_h = site_hidden; q.site_o = site_out; q.row0 = row0;
    q.post_bias = post_bias; q.post_pos = post_npos > 0 ? post_pos : post_bias; q.post_out = post_out; q.post_parts = Npost / 32;
    q.pre_bias = pre_bias; q.pre_res = pre_res; q.site_pre = site_pre;
    q.post_S = post_npos > 0 ? post_S : 1; q.post_npos = post_npos; q.post_ld = (int)post_ld; q.post_ldpos = (int)post_ldpos;
    const int smem = LDS_LN + 4 * FC * 4;
    const dim3 grid(cdiv(M, 128)), block(256);
    static S2dDevOnce attr[32];
    const bool drop = q.thresh != 0, ln1 = ln1_gamma != nullptr, ln2 = ln2_gamma != nullptr, post = Npost > 0, pre = pre_bias != nullptr;
    const void *fn = nullptr;
#define S2D_FFN_CASE(D, A, B, P, R) if (drop == D && ln1 == A && ln2 == B && post == P && pre == R) fn = (const void *)ffn_f16x3_kernel<D, A, B, P, R>;
    S2D_FFN_CASE(false, false, false, false, false) S2D_FFN_CASE(false, false, true, false, false) S2D_FFN_CASE(false, true, true, false, false)
    S2D_FFN_CASE(true, false, false, false, false) S2D_FFN_CASE(true, false, true, false, false) S2D_FFN_CASE(true, true, true, false, false)
    S2D_FFN_CASE(false, true, true, true, false) S2D_FFN_CASE(true, true, true, true, false)
    S2D_FFN_CASE(false, true, true, false, true) S2D_FFN_CASE(true, true, true, false, true)
    S2D_FFN_CASE(false, true, true, true, true) S2D_FFN_CASE(true, true, true, true, true)
#undef S2D_FFN_CASE
    if (!fn) return S2D_ERR_ARG;          // combinations without a caller are not instantiated
    const int slot = (pre ? 16 : 0) + (post ? 8 : 0) + (drop ? 4 : 0) + (ln1 ? 2 : 0) + (ln2 ? 1 : 0);
    if (!attr[slot].done()) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) return S2D_ERR_LAUNCH;
        attr[slot].mark();
    }
    void *args[] = {&q};
    if (hipLaunchKernel(fn, grid, block, args, smem, stream) != hipSuccess) return S2D_ERR_LAUNCH;
    S2D_CHECK_LAUNCH();
    return S2D_OK;
}

}  // extern "C"
