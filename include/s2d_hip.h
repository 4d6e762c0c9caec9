/*
 * include/s2d_hip.h -- C ABI of libs2d_hip.so, the MI355X (gfx950) implementation of the S2D hot path.
 *
 * The reference (leonsick/s2d) exposes this path through Python/detectron2 registries and ONE native
 * FFI, the pybind module `MultiScaleDeformableAttention`
 * (model_training/mask2former/modeling/pixel_decoder/ops/src/vision.cpp:18-21,
 *  .../src/ms_deform_attn.h:25-66).  Every entry point below replaces the reference code cited next to it.
 *
 * Conventions: plain pointers to DEVICE memory and sizes, no framework types; the caller owns every
 * buffer; work is enqueued on `stream` and the call returns immediately; thread-safe per stream;
 * return 0 on success, a negative S2D_ERR_* code otherwise (never throws, never synchronises).
 * Layouts are stated per function.  "NHWC" = channels-last feature maps, which is also the reference's
 * [N, S, C] token layout once H*W is flattened.
 */
#ifndef S2D_HIP_H
#define S2D_HIP_H
#include <stdint.h>

#ifndef HIP_INCLUDE_HIP_HIP_RUNTIME_API_H
typedef struct ihipStream_t *hipStream_t;
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define S2D_OK 0
#define S2D_ERR_ARG (-1)
#define S2D_ERR_LAUNCH (-2)

int s2d_abi_version(void);

/* Arithmetic of the dense contractions below (process-wide), all fp32 in / fp32 out:
 * 2 (default) = split-fp16 x3 on the f16 MFMA (x = h + l*2^-11; A.B^T ~= Ah.Bh^T + 2^-11 (Ah.Bl^T + Al.Bh^T), f32
 *     accumulation: ~5e-7 relative; operands must satisfy |x| < 65504);
 * 1 = split-bf16 x3 on the bf16 MFMA (~5e-6 relative, no range limit);
 * 0 = fp32-input MFMA (an exact f32 FMA chain, 1/16 of the 16-bit rate). */
int s2d_set_dense_mode(int mode);

/* ---- dense contractions (fp32-input MFMA) ------------------------------------------------------ */

/* C[b][M,N] = act((A[b][M,K] * B[b][N,K]^T) * scale[N] + bias[N] + res[b][M,N]); scale/bias/res may be NULL.
 * res_rows > 0: the residual is row-periodic, res[b][row % res_rows, :] (a per-position term shared by all frames, e.g.
 * pos . W^T of "(src + pos) . W^T", ms_deform_attn.py:107-108 via msdeformattn.py:68); res_cols > 0: only columns
 * < res_cols receive the residual (multiple of 4).
 * Replaces every nn.Linear / 1x1 Conv2d on the path (e.g. ms_deform_attn.py:98-104,124; msdeformattn.py:122-131;
 * video_mask2former_transformer_decoder.py:99-111,164-168,193-205) and the mask-logit einsum
 * "bqc,btchw->bqthw" (video_mask2former_transformer_decoder.py:455).  K, lda, ldb multiples of 4. */
int s2d_gemm_nt_f32(const float *A, const float *B, float *C, int M, int N, int K, long lda, long ldb, long ldc,
                    int batch, long strideA, long strideB, long strideC, const float *scale, const float *bias,
                    const float *res, long ldr, long strideR, int res_rows, int res_cols, int relu, const void *B_split,
                    hipStream_t stream);

/* The same contraction with nn.Dropout fused into the epilogue: C = act(dropout_p(A.B^T + bias) + res) -- the three dropout
 * sites of the pixel decoder's encoder layers in training mode (msdeformattn.py:101-125: dropout1 on the attention output,
 * dropout2 after the FFN activation (ReLU and a non-negative mask commute), dropout3 on the FFN output; each before its
 * residual add).  The mask is counter-based (ABI 10: 8 bits per element, one generator call per 16 elements): element (row, col)
 * of the [M,N] output is kept iff byte (col % 4) of word ((col % 16) / 4) of
 * Philox4x32-10(counter = (row0 + row, col / 16, site, 0), key = seed) is >= T = round(p * 256) (row0: the mask row of output
 * row 0, for a launch over a row range of a larger activation): P(drop) = T / 256, p quantised to 1 / 256; kept elements are
 * multiplied by 256 / (256 - T) = 1 / P(keep) (inverted dropout, unbiased for the realised keep probability); the backward
 * regenerates it (s2d_dropout_f32) instead of storing it.  Unbatched; N, ldc, ldr multiples of 8; dense mode 2 only
 * (S2D_ERR_ARG otherwise). */
int s2d_gemm_nt_dropout_f32(const float *A, const float *B, float *C, int M, int N, int K, long lda, long ldb, long ldc,
                            const float *bias, const float *res, long ldr, int relu, const void *B_split, float p,
                            uint64_t seed, unsigned site, unsigned row0, hipStream_t stream);

/* y[M,N] = x * mask / P(keep) with the mask of s2d_gemm_nt_dropout_f32 for the same (p, seed, site): the gradient of a
 * dropout site (and the mask itself, from x = 1).  N multiple of 8; y may alias x. */
int s2d_dropout_f32(const float *x, long M, int N, float p, uint64_t seed, unsigned site, unsigned row0, float *y, hipStream_t stream);

/* Static weights can be split into their fp16 hi/lo image once (dense mode 2) instead of in every launch that reads
 * them: out = s2d_split_weights_words(N,K) 32-bit words, laid out [N][ceil(K/32)][16 words hi | 16 words lo] (the LDS
 * row image of the kernels, zero padded past K).  Pass it as B_split / w_split together with the fp32 weights (the
 * other dense modes read those); NULL = split on the fly.  Results are bit-identical either way.  Unbatched B only. */
/* dgrad GEMM with the gate of the differentiated layer in its epilogue: C = gate[row][col] > 0 ? (A.B^T + res) * gate_scale : 0.
 * Replaces `grad_input = grad_output @ W` followed by the threshold_backward / dropout-scale pass autograd runs for
 * F.relu / nn.Dropout in the reference's FFNs (mask2former/modeling/pixel_decoder/msdeformattn.py:121-125,
 * mask2former_video/modeling/transformer_decoder/video_mask2former_transformer_decoder.py FFNLayer).  split-fp16 mode, N, ldc, ldr,
 * ldg multiples of 4; otherwise S2D_ERR_ARG. */
int s2d_gemm_nt_gate_f32(const float *A, const float *B, float *C, int M, int N, int K, long lda, long ldb, long ldc, const float *scale,
                         const float *res, long ldr, const float *gate, long ldg, float gate_scale, const void *B_split, hipStream_t stream);
/* ... and the convolution form (dgrad of a k x k convolution whose input came out of a ReLU; scale: the per-channel FrozenBatchNorm
 * factor of the layer that produced that input, detectron2 FrozenBatchNorm2d folded as in s2d_conv2d_nhwc_f32; gate [N,Ho,Wo,Cout]) */
int s2d_conv2d_nhwc_gate_f32(const float *x, const float *w, float *y, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                             int pad, const float *scale, const float *gate, float gate_scale, const void *w_split, hipStream_t stream);
/* The GEMM with A pre-split as well: A_split is the s2d_split_weights_f16 image of A's M rows (an activation whose producer wrote it
 * in that layout, or a one-off conversion); B_split is required.  Same result bits as s2d_gemm_nt_f32 on the fp32 A.  Shapes the
 * wave-specialised kernel does not take (K < 224, K % 32, N % 4) return S2D_ERR_ARG. */
int s2d_gemm_nt_presplit_f32(const void *A_split, const float *B, float *C, int M, int N, int K, long ldb, long ldc, const float *bias,
                             const float *res, long ldr, int res_rows, int res_cols, int relu, const void *B_split, hipStream_t stream);
long s2d_split_weights_words(int N, int K);
int s2d_split_weights_f16(const float *W, int N, int K, long ldw, void *out, hipStream_t stream);

/* The encoder layer's feed-forward block in one launch (split-fp16 x3 arithmetic, dense mode 2's):
 *     xn = ln1_gamma ? LayerNorm(x; ln1_gamma, ln1_beta, eps) : x
 *     y  = LN2?( xn + dropout_p( W2 . dropout_p( relu( W1 . xn + b1 ) ) + b2 ) )      LN2 applied when ln2_gamma != NULL
 * i.e. `src = norm2(src + dropout3(linear2(dropout2(relu(linear1(src))))))` of MSDeformAttnTransformerEncoderLayer
 * (mask2former/modeling/pixel_decoder/msdeformattn.py:116-131), optionally with the layer's norm1 (:125) folded into the input side.
 * The F-wide hidden activation stays in registers.  x, y [M, C] fp32 (C = 256; F a multiple of 32, <= 2048); pack = the image
 * s2d_ffn_pack_f16 wrote from W1 [F, C] and W2 [C, F] (s2d_ffn_pack_words(C, F, Npost, pre) 32-bit words; -1 = unsupported sizes).
 * Dropout: p = 0 -> none; otherwise the counter-based masks of s2d_gemm_nt_dropout_f32 for (seed, site_hidden) on the [M, F] hidden
 * activation and (seed, site_out) on the [M, C] output, mask row of row 0 = row0 -- the same bits the two-launch form applies.
 * xn (optional, requires ln1_gamma): receives the normalised input.
 * Npost > 0 (a multiple of 32; requires both LayerNorms): the same launch also applies the NEXT encoder layer's merged projection to
 * every output row while it is still in registers -- post_out[row][n] = y[row] . Wpost[n]^T + post_pos[row % post_S][n] for
 * n < post_npos (a multiple of 32: the row-periodic `pos . W^T + b` term of the sampling offsets / attention logits, which carries
 * their bias -- post_bias is NOT read for these columns; post_ldpos its row stride) and y[row] . Wpost[n]^T + post_bias[n] for
 * post_npos <= n < Npost, row stride post_ld -- i.e. [sampling_offsets | attention_weights | value_proj] of
 * ops/modules/ms_deform_attn.py:98-104 applied to the layer output, which is the next layer's `src`.  Wpost [Npost, C] is packed behind
 * the FFN weights by s2d_ffn_pack_f16.
 * pre_bias != NULL (requires both LayerNorms and xn): the launch also takes over what precedes norm1 in the layer -- x is then the
 * deformable attention's sampled values [M, C] and the FFN's input row becomes
 *     x1 = pre_res + dropout_p( Wpre . x + pre_bias )        mask (seed, site_pre)
 * i.e. `src = src + dropout1(output_proj(samp))` (ms_deform_attn.py:124, msdeformattn.py:124-125) with pre_res the layer input [M, C];
 * Wpre [C, C] is packed behind Wpost (s2d_ffn_pack_f16's Wpre, s2d_ffn_pack_words' pre = 1).  xn receives norm1(x1) as before. */
long s2d_ffn_pack_words(int C, int F, int Npost, int pre);
int s2d_ffn_pack_f16(const float *W1, const float *W2, int C, int F, const float *Wpost, int Npost, const float *Wpre, void *out,
                     hipStream_t stream);
int s2d_ffn_fused_f32(const float *x, long M, int C, int F, const void *pack, const float *b1, const float *b2, const float *ln1_gamma,
                      const float *ln1_beta, const float *ln2_gamma, const float *ln2_beta, float eps, float p, uint64_t seed,
                      unsigned site_hidden, unsigned site_out, unsigned row0, float *xn, float *y, int Npost, const float *post_bias,
                      const float *post_pos, int post_S, int post_npos, long post_ldpos, float *post_out, long post_ld,
                      const float *pre_bias, const float *pre_res, unsigned site_pre, hipStream_t stream);

/* NHWC convolution as implicit GEMM: x [N,H,W,Cin] (Cin % 4 == 0), w [Cout][KH][KW][Cin],
 * y [N,Ho,Wo,Cout] = act(conv(x,w) * scale[Cout] + bias[Cout] + res).  Replaces detectron2 Conv2d+FrozenBN+ReLU
 * of the R50 trunk (build_resnet_backbone, call site kd_video_maskformer_model.py:132,135) and the FPN
 * 3x3 output conv (msdeformattn.py:264-281). */
int s2d_conv2d_nhwc_f32(const float *x, const float *w, float *y, int N, int H, int W, int Cin, int Cout, int KH,
                        int KW, int stride, int pad, const float *scale, const float *bias, const float *res,
                        int relu, const void *w_split, hipStream_t stream);

/* The two contractions with fp16 operands / f32 accumulate / f32 output -- autocast-LIKE (engine/train_loop.py:709 `with autocast():`,
 * SOLVER.AMP.ENABLED True in every shipped yaml; real autocast additionally rounds every output to fp16): operands rounded to fp16
 * (nearest-even), f32 accumulation by ONE fp16 MFMA per product, f32 out.  Opt-in, for the modules autocast runs in fp16 (the R50 trunk, the video decoder's linear layers, the
 * mask-logit einsum); the pixel decoder and the matcher force fp32 in the reference (msdeformattn.py:314, matcher.py:266-268) and
 * never take these.  Same arguments as s2d_gemm_nt_f32 / s2d_conv2d_nhwc_f32 without the pre-split image. */
int s2d_gemm_nt_amp_f32(const float *A, const float *B, float *C, int M, int N, int K, long lda, long ldb, long ldc, int batch,
                        long strideA, long strideB, long strideC, const float *scale, const float *bias, const float *res, long ldr,
                        long strideR, int res_rows, int res_cols, int relu, hipStream_t stream);
int s2d_conv2d_nhwc_amp_f32(const float *x, const float *w, float *y, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                            int pad, const float *scale, const float *bias, const float *res, int relu, hipStream_t stream);

/* ---- multi-scale deformable attention (HBM/L2-bound gather, no MFMA) --------------------------------- */

/* Drop-in for MSDA.ms_deform_attn_forward (ops/src/ms_deform_attn.h:25-44, ops/src/vision.cpp:19; kernel
 * ops/src/cuda/ms_deform_im2col_cuda.cuh:242-304): value [N,S,M,D], sampling_loc [N,Lq,M,L,P,2] (x,y in
 * [0,1]), attn_w [N,Lq,M,L,P] -> out [N,Lq,M*D].  spatial shapes [L,2]=(H,W) and level starts [L] are HOST
 * int64 arrays (they are tiny and known on the host; the reference reads them from device memory).
 * No im2col_step chunking is needed. */
int s2d_msda_forward_f32(const float *value, const int64_t *shapes_host, const int64_t *level_start_host,
                         const float *loc, const float *attn_w, int N, int S, int M, int D, int L, int Lq, int P,
                         float *out, hipStream_t stream);

/* Drop-in for MSDA.ms_deform_attn_backward (ops/src/ms_deform_attn.h:46-66; kernels cuh:306-408, formulas
 * :119-163).  grad buffers are zero-initialised here, as the reference does (ms_deform_attn_cuda.cu:119-121). */
int s2d_msda_backward_f32(const float *value, const int64_t *shapes_host, const int64_t *level_start_host,
                          const float *loc, const float *attn_w, const float *grad_out, int N, int S, int M, int D,
                          int L, int Lq, int P, float *grad_value, float *grad_loc, float *grad_attn_w,
                          hipStream_t stream);

/* The same gradients without float atomics, bitwise reproducible (what the training step uses): the sampling graph is
 * inverted once per call -- samples keyed by their bilinear cell, histogram + prefix sum, stable radix sort -- and every
 * grad_value row is then written once by a gather over the four cells that touch its pixel; grad_loc / grad_attn are
 * query-owned.  workspace: s2d_msda_backward_workspace_bytes(...) bytes of device memory.  Lq, M, L, P as above; D == 32. */
long s2d_msda_backward_workspace_bytes(const int64_t *shapes_host, int N, int M, int L, int Lq, int P);
int s2d_msda_backward_sorted_f32(const float *value, const int64_t *shapes_host, const int64_t *level_start_host,
                                 const float *loc, const float *attn_w, const float *grad_out, int N, int S, int M, int D,
                                 int L, int Lq, int P, float *grad_value, float *grad_loc, float *grad_attn_w, void *workspace,
                                 long workspace_bytes, hipStream_t stream);
/* The same with row strides: value rows (n, s) are M * D floats at stride ldv, grad_value rows at stride ldg (column slices of
 * the merged projection output / of its gradient buffer: no contiguous copy of value, no concatenation of the gradients).
 * grad_loc == grad_attn_w == NULL: grad_value only (the caller takes the query-owned half from s2d_msda_fused_backward_query_f32). */
int s2d_msda_backward_sorted_strided_f32(const float *value, long ldv, const int64_t *shapes_host, const int64_t *level_start_host,
                                         const float *loc, const float *attn_w, const float *grad_out, int N, int S, int M, int D,
                                         int L, int Lq, int P, float *grad_value, long ldg, float *grad_loc, float *grad_attn_w,
                                         void *workspace, long workspace_bytes, hipStream_t stream);

/* The drop-in op with the reference's own argument kinds: spatial shapes [L,2] and level starts [L] as int64 DEVICE tensors,
 * read inside the kernels exactly as ms_deformable_im2col_gpu_kernel does (ops/src/cuda/ms_deform_attn_cuda.cu:60-75 passes
 * spatial_shapes.data<int64_t>() / level_start_index.data<int64_t>() straight to the kernels; the module builds both tensors
 * fresh on every forward, msdeformattn.py:82-83).  No host copy, no cache, no synchronisation: a one-thread kernel turns the two
 * tensors into the geometry record at the head of `workspace` and every kernel of the call reads it from there.
 * Shapes that fail the host form's argument checks (H, W <= 0; a level past S), or whose levels overlap (sum of H*W > S: the
 * reference would read the shared rows twice; the sorted backward's workspace is sized for disjoint levels), cannot be reported through the return code
 * without a sync: the call then produces zeros and sets the record's error flag, which s2d_msda_dev_status reads back.
 * workspace: s2d_msda_dev_forward_workspace_bytes() / s2d_msda_dev_backward_workspace_bytes(...) bytes of device memory, 16-B
 * aligned, private to the call until it has finished.  The backward is the atomic-free sorted form. */
long s2d_msda_dev_forward_workspace_bytes(void);
int s2d_msda_forward_dev_f32(const float *value, const int64_t *shapes_dev, const int64_t *level_start_dev, const float *loc,
                             const float *attn_w, int N, int S, int M, int D, int L, int Lq, int P, float *out, void *workspace,
                             hipStream_t stream);
long s2d_msda_dev_backward_workspace_bytes(int N, int S, int M, int L, int Lq, int P);
int s2d_msda_backward_dev_f32(const float *value, const int64_t *shapes_dev, const int64_t *level_start_dev, const float *loc,
                              const float *attn_w, const float *grad_out, int N, int S, int M, int D, int L, int Lq, int P,
                              float *grad_value, float *grad_loc, float *grad_attn_w, void *workspace, long workspace_bytes,
                              hipStream_t stream);
/* Failing loudly without a sync: `word` is an int in memory that both the GPU and the host can address (pinned host memory, e.g.
 * hipHostMalloc / torch's pin_memory; NULL: none), registered once for the process.  A *_dev call whose shapes are rejected stores 1
 * into it (system scope, never cleared by the library): the caller tests the word with a plain host read before / after its calls and
 * raises -- at the latest one call after the offending one has executed -- instead of training on zeros.  The drop-in module
 * (s2d_amd/compat/MultiScaleDeformableAttention.py) does exactly that. */
/* float64 instantiation of the drop-in op (the reference extension dispatches on floating types, ms_deform_attn_cuda.cu:69, :137; its
 * ops/test.py gradchecks in double): same argument kinds as the *_dev_f32 entries, no workspace.  Not a hot path -- one thread per
 * output channel, grad_value / grad_sampling_loc / grad_attn_weight accumulated with double atomics (as the reference's col2im kernels
 * do), so the gradients are reproducible to rounding, not bitwise.  A level that does not lie inside S contributes nothing. */
int s2d_msda_forward_dev_f64(const double *value, const int64_t *shapes_dev, const int64_t *level_start_dev, const double *loc,
                             const double *attn_w, int N, int S, int M, int D, int L, int Lq, int P, double *out, hipStream_t stream);
int s2d_msda_backward_dev_f64(const double *value, const int64_t *shapes_dev, const int64_t *level_start_dev, const double *loc,
                              const double *attn_w, const double *grad_out, int N, int S, int M, int D, int L, int Lq, int P,
                              double *grad_value, double *grad_loc, double *grad_attn_w, hipStream_t stream);

int s2d_msda_dev_error_word(int *word);
/* *err_host = the error flag of the *_dev call that used `workspace` (0 = shapes accepted).  Synchronises `stream`. */
int s2d_msda_dev_status(const void *workspace, int *err_host, hipStream_t stream);

/* Fused pixel-decoder form: also does softmax over L*P and loc = ref + off/(W_l,H_l)
 * (ops/modules/ms_deform_attn.py:101-109) with the query's own pixel centre as reference point
 * (msdeformattn.py:141-153).  offs_logits [N,S,ldoa]: per query M*L*P*2 raw offsets then M*L*P raw logits.
 * Requires D == 32, L*P == 12 (the shipped geometry, msdeformattn.py:232-239). */
int s2d_msda_fused_forward_f32(const float *value, int ldv, const int64_t *shapes_host, const float *offs_logits, int ldoa,
                               int N, int S, int M, int D, int L, int P, float *out, hipStream_t stream);

/* Backward of the fused form = prep (raw offsets / logits -> loc [N,S,M,L,P,2], softmaxed attn [N,S,M,L,P] with the
 * forward's reference points), s2d_msda_backward_f32 on those, chain (d_loc, d_attn -> d_offs_logits [N,S,ldd], the
 * gradient of the merged projection output: d_off = d_loc / (W_l, H_l), d_logit = a (d_a - sum a d_a)). */
int s2d_msda_fused_prep_f32(const float *offs_logits, int ldoa, const int64_t *shapes_host, int N, int S, int M, int L, int P,
                            float *loc, float *attn, hipStream_t stream);
int s2d_msda_fused_chain_f32(const float *attn, const float *grad_loc, const float *grad_attn, const int64_t *shapes_host, int N,
                             int S, int M, int L, int P, float *d_offs_logits, int ldd, hipStream_t stream);
/* The query-owned half of that backward in ONE launch for the S2D geometry (M = 8, D = 32, L = 3, P = 4; anything else: S2D_ERR_ARG):
 * d_offs_logits [N][S][ldd] straight from grad_out [N][S][M][32], the value rows (stride ldv) and the raw projection rows -- what
 * grad_loc / grad_attn of s2d_msda_backward_sorted*_f32 followed by s2d_msda_fused_chain_f32 give (reference: the grad_sampling_loc /
 * grad_attn_weight half of ms_deformable_col2im_gpu_kernel_*, ops/src/cuda/ms_deform_im2col_cuda.cuh:119-163, and autograd's chain
 * through ms_deform_attn.py:103-113), without those two tensors in memory.  16-B aligned value / offs_logits / grad_out rows. */
int s2d_msda_fused_backward_query_f32(const float *value, int ldv, const int64_t *shapes_host, const float *offs_logits, int ldoa,
                                      const float *grad_out, int N, int S, int M, int D, int L, int P, float *d_offs_logits, int ldd,
                                      hipStream_t stream);

/* ---- bandwidth-bound glue (HBM-bound, 16-B accesses) ----------------------------------------------- */

/* (x - mean)/std per frame + zero pad bottom/right to (Hp,Wp): kd_video_maskformer_model.py:263-269 and
 * detectron2 ImageList.from_tensors.  frames u8 [F,3,H0,W0] (the mapper's layout, dataset_mapper.py:306-404)
 * -> out f32 [F,Hp,Wp,4] (NHWC, 4th channel zero so the 7x7 stem runs as a Cin=4 implicit GEMM).
 * mean3/std3 are HOST arrays. */
int s2d_normalize_pad_nhwc4_f32(const uint8_t *frames, int F, int H0, int W0, int Hp, int Wp, const float *mean3_host,
                                const float *std3_host, float *out, hipStream_t stream);

/* 3x3/2 pad 1 max pool, NHWC (detectron2 BasicStem). y [N,(H+1)/2,(W+1)/2,C]. */
int s2d_maxpool3x3s2_nhwc_f32(const float *x, int N, int H, int W, int C, float *y, hipStream_t stream);
/* The same, also writing the arg-max tap (ky * 3 + kx, the first maximum in scan order) of every output element: argmax [N][Ho][Wo][C] bytes,
 * 4-byte aligned.  The training step keeps it for s2d_maxpool3x3s2_backward_idx_nhwc_f32. */
int s2d_maxpool3x3s2_nhwc_idx_f32(const float *x, int N, int H, int W, int C, float *y, unsigned char *argmax, hipStream_t stream);

/* y = GroupNorm_G(x)*gamma+beta [+ bilinear_resize(up [N,hu,wu,C] -> (H,W), align_corners=False)] [relu]; x NHWC.
 * nn.GroupNorm(32,256) at msdeformattn.py:213-226 and detectron2 get_norm("GN") at :261-281; the fused
 * upsample-add is msdeformattn.py:349.  stats_ws: s2d_groupnorm_workspace_doubles(N,H,W,G) doubles (statistics + the
 * per-block partials they are reduced from in a fixed order: no atomics, bitwise reproducible). */
long s2d_groupnorm_workspace_doubles(int N, int H, int W, int G);
int s2d_groupnorm_nhwc_f32(const float *x, int N, int H, int W, int C, int G, const float *gamma, const float *beta,
                           float eps, const float *up, int hu, int wu, int relu, double *stats_ws, float *y,
                           hipStream_t stream);

/* y = LayerNorm(x + res) over the last dim (res may be NULL): the post-norm residual blocks of
 * msdeformattn.py:116-131 and video_mask2former_transformer_decoder.py:41-51,99-111,164-168. */
int s2d_layernorm_f32(const float *x, const float *res, const float *gamma, const float *beta, long rows, int C,
                      float eps, float *y, hipStream_t stream);

/* y[i] = x[i] + b[i % bn] (n, bn multiples of 4): with_pos_embed / level-embed adds. */
int s2d_add_bcast_f32(const float *x, const float *b, long n, long bn, float *y, hipStream_t stream);

/* Sine position encoding written token-major [T*H*W, 2F] (+ add_c[2F] if not NULL).  T == 0: 2-D form
 * (mask2former/modeling/transformer_decoder/position_encoding.py:29-52); T > 0: 3-D form
 * (mask2former_video/modeling/transformer_decoder/position_encoding.py:29-57). */
int s2d_pe_sine_f32(int T, int H, int W, int num_pos_feats, const float *add_c, float *out, hipStream_t stream);

/* ---- masked cross-attention of the video decoder (fp32 MFMA QK^T / AV, streamed over keys) ----------- */

/* Attention-mask builder: bilinear-resize the pixel-major mask logits [B][T*hm*wm][ldq] to the level size
 * (hl,wl), threshold sigmoid<0.5 (== logit<0) and pack to bits [B][K=T*hl*wl][4] (bit q set = query q must NOT
 * attend key); unmasked [B][4] gets bit q set iff query q has at least one attendable key.
 * video_mask2former_transformer_decoder.py:460-465 (the x8 head repeat is implicit: heads share the bits).
 * compact != 0: mask_logits holds only the four bilinear source pixels of every key, [B][K][4][ldq] in the order
 * (y0,x0) (y0,x1) (y1,x0) (y1,x1) of the ATen align_corners=False source rule -- for a network whose intermediate mask
 * predictions feed nothing but the next layer's attention mask (the frozen teacher). */
int s2d_attn_mask_bits(const float *mask_logits, int ldq, int B, int Q, int T, int hm, int wm, int hl, int wl, int compact,
                       uint32_t *bits, uint32_t *unmasked, hipStream_t stream);

/* floats of workspace s2d_masked_attn_f32 needs */
long s2d_attn_workspace_floats(int B, int H, int K);

/* out[b][q][:] = concat_h softmax_k(q_h k_h^T / sqrt(32) + mask) v_h with q [B][Q][C], k/v [B][K] rows of C floats
 * at row strides ldk / ldv >= C (a column slice of a wider projection output; multiples of 4), already projected,
 * C = 32*H, Q <= 128.  bits/unmasked from s2d_attn_mask_bits (NULL = no mask: the self-attention
 * of :41-51); a query with no attendable key attends everywhere (the fix at :413).  lse (may be NULL) [B][H][128]
 * receives the base-2 log-sum-exp of the scaled scores, which the backward needs.  Replaces the core of
 * nn.MultiheadAttention at :99-111 without materialising the [B*8,Q,K] mask or the scores. */
int s2d_masked_attn_f32(const float *q, const float *k, const float *v, long ldk, long ldv, const uint32_t *bits,
                        const uint32_t *unmasked, int B, int Q, int K, int C, int H, float *workspace, float *out, float *lse,
                        hipStream_t stream);

/* Backward of s2d_masked_attn_f32.  lse [B][H][128]: the base-2 log-sum-exp the forward leaves when
 * its `lse` argument is non-NULL; out = the forward output; dout [B][Q][C].  dq [B][Q][C], dk / dv [B][K][C] (dense).
 * Probabilities are recomputed from lse; dQ partials over key ranges are added in a fixed order.  workspace:
 * s2d_attn_backward_workspace_floats(B, H, K) floats. */
long s2d_attn_backward_workspace_floats(int B, int H, int K);
int s2d_masked_attn_backward_f32(const float *q, const float *k, const float *v, long ldk, long ldv, const uint32_t *bits,
                                 const uint32_t *unmasked, const float *out, const float *lse, const float *dout, int B, int Q,
                                 int K, int C, int H, float *workspace, float *dq, float *dk, float *dv, hipStream_t stream);
/* The same with row strides on dk / dv (lddk, lddv >= C, multiples of 4, 16-B aligned bases): the gradients land in column slices of a wider
 * buffer -- the video decoder collects the key / value gradients of the three layers that share a memory level side by side for ONE
 * projection backward (video_mask2former_transformer_decoder.py:99-111 x 9 layers), without a copy per layer. */
int s2d_masked_attn_backward_strided_f32(const float *q, const float *k, const float *v, long ldk, long ldv, const uint32_t *bits,
                                         const uint32_t *unmasked, const float *out, const float *lse, const float *dout, int B, int Q,
                                         int K, int C, int H, float *workspace, float *dq, float *dk, long lddk, float *dv, long lddv,
                                         hipStream_t stream);

/* ---- VideoHungarianMatcher on the device -------------------------------------------------------------- */
/* A criterion pass handles NL prediction layers x B clips = NL*B independent "problems" (problem = layer*B +
 * clip) in one call.  Targets of the pass: tgt u8 [B][Nmax][T][H][W] -- binary masks, nonzero = set (the reference's
 * bool gt_masks, matcher.py:246; a clip with <= 32 targets is read through bit-interleaved words) -- with the per-clip count in DEVICE
 * memory (tgt_count[B]) so that distillation targets, whose number depends on teacher scores, need no host sync.
 * mask_logits are pixel-major [NL][B][T*hm*wm][ldq]; class_logits [NL][B][Q][2]. */

long s2d_matcher_workspace_floats(int NL, int B, int T, int P, int H, int W);

/* C[problem][Q][Nmax] (columns >= tgt_count[clip] are zero) = w_mask*cost_mask + w_class*cost_class +
 * w_dice*cost_dice at P shared random points per problem: matcher.py:236-287 (batch_sigmoid_ce_loss :38-62,
 * batch_dice_loss :15-30, point_sample).  coords [NL][B][P][2] injects the torch.rand(1,P,2) draws of :252
 * (parity mode); NULL = counter-based device RNG keyed by (seed, problem, point). */
int s2d_matcher_cost_f32(const float *mask_logits, const float *class_logits, const uint8_t *tgt, const int *tgt_count,
                         const float *coords, uint64_t seed, int NL, int B, int Q, int ldq, int T, int hm, int wm, int H,
                         int W, int Nmax, int P, float w_class, float w_mask, float w_dice, float *workspace, float *C,
                         hipStream_t stream);

/* scipy.optimize.linear_sum_assignment (matcher.py:289) for nprob cost matrices C[problem][Q][Nmax] using the
 * first tgt_count[problem % B] columns.  idx_q/idx_t [nprob][min(Q,Nmax)] receive (query, target) pairs in scipy's
 * order (queries ascending), n_match[nprob] their number = min(Q, N).  Q <= 128, Nmax <= 128, Q*Nmax <= 12800. */
int s2d_lsap_f32(const float *C, const int *tgt_count, int nprob, int B, int Q, int Nmax, int *idx_q, int *idx_t,
                 int *n_match, hipStream_t stream);

/* ---- VideoSetCriterion on the device + distillation targets ------------------------------------------ */

/* prepare_distillation_targets (kd_video_maskformer_model.py:436-468, nms off): per clip keep the queries whose
 * teacher score softmax(logits)[q][0] is among the top `topk` and >= score_thr, in ascending query order, and
 * write masks = bilinear(teacher mask logits [B][T*hm*wm][ldq] -> (H,W), align_corners=False) > 0 as u8 planes
 * tgt [B][Nmax][T][H][W]; count[B], kept_q[B][Nmax], nonempty[B][Nmax][T] (DropLoss predicate) on the device. */
int s2d_kd_targets_u8(const float *t_class_logits, const float *t_mask_logits, float score_thr, int topk, int B, int Q,
                      int ldq, int T, int hm, int wm, int H, int W, int Nmax, uint8_t *tgt, int *count, int *kept_q,
                      int *nonempty, hipStream_t stream);

/* nonempty[b][n][t] = any(tgt[b][n][t]) for ground-truth targets (criterion.py:310-313). H*W % 16 == 0. */
int s2d_target_nonempty(const uint8_t *tgt, const int *count, int B, int Nmax, int T, int H, int W, int *nonempty,
                        hipStream_t stream);

/* H, W: the padded frame size of the targets (frames beyond 1.15 M pixels keep a bit-packed target plane per workgroup behind the
 * sample buffer: their plane does not fit LDS) */
long s2d_point_loss_workspace_bytes(int NL, int B, int Q, int Nmax, int T, int hm, int wm, int num_points,
                                    float oversample_ratio, float importance_ratio, int H, int W);

/* loss_masks for NL layers at once (criterion.py:292-356, point_features.py:63-116): losses[layer][0] = loss_mask,
 * [1] = loss_dice, both already divided by num_masks = max(sum_b tgt_count[b] / world_size, 1) (:404-409).
 * idx_q/idx_t/n_match from s2d_lsap_f32.  coords_over [NL][B*maxm*T][int(P*oversample)][2] and
 * coords_rand [NL][B*maxm*T][P - int(importance*P)][2] inject the two torch.rand draws of
 * point_features.py:90,112, indexed by the row's rank among the kept rows of its layer (parity mode);
 * NULL = device RNG.  drop_empty = temporal DropLoss ("masks-only" strategy, criterion.py:307-322). */
int s2d_point_loss_f32(const float *mask_logits, const uint8_t *tgt, const int *tgt_count, const int *nonempty,
                       const int *idx_q, const int *idx_t, const int *n_match, const float *coords_over,
                       const float *coords_rand, uint64_t seed, int NL, int B, int Q, int ldq, int T, int hm, int wm, int H,
                       int W, int Nmax, int num_points, float oversample_ratio, float importance_ratio, int drop_empty,
                       float world_size, void *workspace, float *losses, hipStream_t stream);

/* loss_labels (criterion.py:227-251) for one layer: class_logits [B][Q][2], idx_q [B][maxm], n_match [B]. */
int s2d_class_loss_f32(const float *class_logits, const int *idx_q, const int *n_match, int B, int Q, int maxm,
                       float eos_coef, float *loss_ce, hipStream_t stream);

/* ---- eval-side step after the path: inference_video -------------------------------- */

/* kd_video_maskformer_model.py:532-538 (= video_maskformer_model.py:300-306): scores = softmax(class_logits
 * [Q][C+1])[:, :-1] flattened to [Q*C]; sorted top-K (descending; ties -> lower flat index); scores[K], query[K] =
 * flat // C, label[K] = flat % C on the device.  Q*C <= 16384, 1 <= K <= Q*C. */
int s2d_infer_select_f32(const float *class_logits, int Q, int C, int K, float *scores, int *query, int *label,
                         hipStream_t stream);

/* floats of workspace s2d_infer_masks_u8 needs (the K gathered low-resolution planes) */
long s2d_infer_workspace_floats(int K, int T, int hm, int wm);

/* 32-bit words of one bit-packed mask [T][oh][ow] (flat index i -> word i/32, bit i%32; tail bits zero) */
long s2d_mask_bit_words(int T, int oh, int ow);

/* masks[k] = bilinear( bilinear(mask_logits[query[k]] -> (Hp,Wp)) [:, :ih, :iw] -> (oh,ow) ) > 0, both resizes
 * align_corners=False, fp32, the expression tree of two F.interpolate calls (kd_video_maskformer_model.py:341-346,
 * :545-550).  mask_logits pixel-major [T*hm*wm][ldq]; masks u8 [K][T][oh][ow]; bits (may be NULL) [K][words] with
 * words = s2d_mask_bit_words(T,oh,ow).  Nothing of size [Q][T][Hp][Wp] is materialised. */
int s2d_infer_masks_u8(const float *mask_logits, int ldq, int T, int hm, int wm, int Hp, int Wp, int ih, int iw, int oh,
                       int ow, const int *query, int K, float *workspace, uint8_t *masks, uint32_t *bits,
                       hipStream_t stream);

/* byte mask planes u8 [K][n] (0 / non-0) -> bit words [K][ceil(n/32)] in the layout above: feeds s2d_mask_pair_counts_u64 for
 * masks that exist as bytes -- the distillation pseudo targets under MODEL.MASK_FORMER.DISTILLATION_NMS
 * (kd_video_maskformer_model.py:484-520). */
int s2d_pack_mask_bits_u8(const uint8_t *masks, int K, long n, uint32_t *bits, hipStream_t stream);

/* inter[i][j] = sum(mask_i & mask_j) for all pairs of K bit-packed masks (diagonal = areas; sum(mask_i | mask_j) =
 * inter[i][i] + inter[j][j] - inter[i][j]): everything the greedy mask-NMS of :552-583 reads, in one launch instead
 * of two full-tensor reductions and a host sync per pair.  inter [K][K] (zeroed here). */
int s2d_mask_pair_counts_u64(const uint32_t *bits, int K, long words, unsigned long long *inter, hipStream_t stream);

/* ---- COCO run-length encoding of masks (wire format after inference_video / of the keymask annotations) ------- */

/* pycocotools rleEncode (third party, restated) of F binary masks u8 [F][H][W] (non-zero = set), column-major runs.
 * Pass 1: col_off [F][W] = exclusive count of run boundaries before column x (a boundary = pixel != its column-major
 * predecessor, the pixel before (0,0) counting as 0); nbound[F] = boundaries per mask (runs = nbound + 1); area[F] =
 * set pixels (mask_util.area); bbox [F][4] = x, y, w, h of the tight box, 0,0,0,0 if empty (mask_util.toBbox).
 * ytvis_eval.py:345-350, keymask_ident/annotations.py:100-106. */
int s2d_rle_count_u8(const uint8_t *masks, int F, int H, int W, int *col_off, int *nbound, int *area, int *bbox,
                     hipStream_t stream);

/* Pass 2: positions[frame_off[f] + i] = column-major index x*H + y of the i-th boundary of mask f (ascending);
 * frame_off[F] = exclusive prefix of nbound.  Run lengths are the differences (first run = positions[0], last =
 * H*W - positions[last]). */
int s2d_rle_positions_u8(const uint8_t *masks, int F, int H, int W, const int *col_off, const long *frame_off,
                         int *positions, hipStream_t stream);

/* maskApi.c rleToString on the device: the run lengths implied by `positions` (frame_off [F+1] = exclusive prefix of
 * nbound, ncounts = frame_off[F] + F runs in all) become the ASCII strings, concatenated in chars (capacity 7 * ncounts
 * bytes); str_off [F+1] = byte offset of every mask's string (str_off[F] = total).  hw = H*W.
 * workspace: s2d_rle_string_workspace_bytes(ncounts) bytes. */
long s2d_rle_string_workspace_bytes(long ncounts);
int s2d_rle_strings_u8(const int *positions, const long *frame_off, int F, long hw, long ncounts, void *workspace,
                       long workspace_bytes, uint8_t *chars, long *str_off, hipStream_t stream);

/* ---- gradients of the dense layers: HBM-bound helpers around s2d_gemm_nt_f32 ----------- */

/* Weight-gradient contraction without transposed copies: C_slices[s][n][k] = sum over the rows m of slice s of
 * A[m][n] * B[m][k]  (A [rowsA][lda] = dY, B [rowsB][ldb] = X, both row-major; Mo = columns of A used, No = columns of B
 * used; slices of `chunk` rows (multiple of 32), ceil(rowsA / chunk) of them; rows m >= rowsB of B read as zero, so B may be
 * a view that starts some rows later, e.g. a convolution tap on the padded grid).  Same split-fp16 x3 arithmetic as
 * s2d_gemm_nt_f32 mode 2.  Slice s writes its [Mo][No] tile at C_slices + s * slice_stride (0 = Mo * No; larger: several
 * launches -- the taps of a convolution -- interleave their tiles so that one s2d_reduce_slices_f32 finishes them all).
 * Mo, No, lda, ldb multiples of 4; A, B 16-B aligned; each operand < 4 GiB.
 * colsum_slices (optional, [slices][Mo]): slice s also leaves the column sums of its rows of A -- the bias gradient of the same dY
 * (s2d_reduce_slices_f32 adds the slices) -- so dY is not read a second time for it. */
int s2d_gemm_tn_f32(const float *A, const float *B, float *C_slices, int Mo, int No, long rowsA, long rowsB, long lda, long ldb,
                    long chunk, long slice_stride, float *colsum_slices, hipStream_t stream);

/* Weight gradient of y = conv2d_nhwc(x [N][H][W][Cin], w [Cout][KH][KW][Cin], stride, pad), dy [N][Ho][Wo][Cout] (detectron2's Conv2d in the
 * R50 trunk and the pixel decoder, e.g. mask2former/modeling/pixel_decoder/msdeformattn.py:289-303; the reference leaves it to autograd's
 * convolution backward), on the TN kernel with the input pixel of every (output position, tap) addressed in place (pixels outside the image
 * read as zero): no padded copy of x, no copy of dy on the input grid.  ONE launch for all taps: slice s of the positions (chunk each, a
 * multiple of 32; S = ceil(N * Ho * Wo / chunk) <= 65535) leaves its partial gradients in part[s], laid out [Cout][KH][KW][Cin] like w;
 * s2d_reduce_slices_f32(part, S, Cout*KH*KW*Cin, Cout*KH*KW*Cin, beta, dw) finishes them in a fixed order.  Cin, Cout % 4 == 0, Ho, Wo > 1. */
int s2d_conv_wgrad_tn_f32(const float *dy, const float *x, int N, int H, int W, int Ci, int Ho, int Wo, int Co, int KH, int KW, int stride,
                          int pad, long chunk, float *part, hipStream_t stream);

/* out[c][r] = in[r][c]; in [R][ldi], out [C][ldo].  dW = dY^T . X runs as an NT GEMM on the transposed operands. */
int s2d_transpose_f32(const float *in, long R, long C, long ldi, float *out, long ldo, hipStream_t stream);

/* dst_t[i] += src_t[i] for a list of tensors in one launch (the "param.grad += g" glue of a training iteration): table [n][3] =
 * (src pointer, dst pointer, element count) as 64-bit words in DEVICE memory; block b adds elements [chunk_off[b], chunk_off[b] +
 * chunk) of tensor chunk_tensor[b].  No two entries may share a destination. */
int s2d_multi_add_f32(const long *table, const int *chunk_tensor, const long *chunk_off, int nchunks, int chunk, hipStream_t stream);

/* out[i] = beta * out[i] + sum_{s < S} part[s * stride + i], s ascending (fixed order: reproducible split-K) */
int s2d_reduce_slices_f32(const float *part, int S, long n, long stride, float beta, float *out, hipStream_t stream);
/* two such reductions with the same slice count in one launch (a weight gradient and its bias gradient): same bits as two calls */
int s2d_reduce_slices_pair_f32(const float *partA, long nA, long strideA, float betaA, float *outA, const float *partB, long nB, long strideB,
                               float betaB, float *outB, int S, hipStream_t stream);

/* part[s][c] = sum of in[r][c] over the rows of slice s (rows_per_slice rows each, ceil(R / rows_per_slice) slices):
 * bias gradients, finished by s2d_reduce_slices_f32 */
int s2d_colsum_slices_f32(const float *in, long R, long C, long ldi, long rows_per_slice, float *part, hipStream_t stream);

/* LayerNorm backward for y = LN(x + res) (res may be NULL) over the last dim C (C % 4 == 0, C <= 1024): dx [rows][C]
 * (= the gradient of x and of res), and per-workgroup partials part [blocks][2][C] (dgamma rows, then dbeta rows) with
 * blocks = s2d_layernorm_backward_blocks(rows); finish with s2d_reduce_slices_f32(part, blocks, 2*C, 2*C, ...). */
long s2d_layernorm_backward_blocks(long rows);
int s2d_layernorm_backward_f32(const float *x, const float *res, const float *dy, const float *gamma, long rows, int C,
                               float eps, float *dx, float *part, hipStream_t stream);

/* GroupNorm backward for y = GN_G(x) * gamma + beta, x / dy NHWC: dx, and per-block partials part_c
 * [s2d_groupnorm_backward_blocks(N,H,W)][2][C] (dgamma rows, dbeta rows), finished by s2d_reduce_slices_f32.  ws:
 * 2 * s2d_groupnorm_workspace_doubles(N,H,W,G) doubles.  The ReLU / upsample-add branches of the forward are separate
 * gradient steps (s2d_relu_scale_backward_f32, s2d_resize_bilinear_backward_nhwc_f32). */
long s2d_groupnorm_backward_blocks(int N, int H, int W);
int s2d_groupnorm_backward_f32(const float *x, const float *dy, const float *gamma, int N, int H, int W, int C, int G, float eps,
                               double *ws, float *dx, float *part_c, hipStream_t stream);

/* adjoint of bilinear_resize(up [N,hu,wu,C] -> (H,W), align_corners=False): dup [N,hu,wu,C] from dy [N,H,W,C] (gather
 * form, reproducible); the gradient of the upsample-add branch of s2d_groupnorm_nhwc_f32 (msdeformattn.py:349). */
int s2d_resize_bilinear_backward_nhwc_f32(const float *dy, int N, int H, int W, int C, int hu, int wu, float *dup,
                                          hipStream_t stream);

/* backward of s2d_maxpool3x3s2_nhwc_f32: dx [N,H,W,C] from x and dy [N,(H+1)/2,(W+1)/2,C]; ties go to the first maximum
 * in window scan order (torch's rule).  Gather form: reproducible. */
int s2d_maxpool3x3s2_backward_nhwc_f32(const float *x, const float *dy, int N, int H, int W, int C, float *dx,
                                       hipStream_t stream);
/* ... from the arg-max taps the forward stored (s2d_maxpool3x3s2_nhwc_idx_f32) instead of x: 4 bytes + (where the pixel is an arg-max) one dy row
 * per window, against 9 input rows per window for the recomputing form. */
int s2d_maxpool3x3s2_backward_idx_nhwc_f32(const unsigned char *argmax, const float *dy, int N, int H, int W, int C, float *dx,
                                           hipStream_t stream);

/* explicit im2col for convolutions with few input channels (the 7x7 stem): col [N*Ho*Wo][KH*KW*C] from x [N,H,W,C], zero
 * outside the image; the weight gradient is then one contraction dY^T . col (s2d_amd/backward.py:conv_weight_grad) */
int s2d_im2col_nhwc_f32(const float *x, int N, int H, int W, int C, int KH, int KW, int stride, int pad, float *col,
                        hipStream_t stream);

/* Depth-to-space step of a stride-2 3 x 3 convolution's input gradient (the R50 trunk's res3.0 / res4.0 / res5.0 conv2, STRIDE_IN_1X1 False):
 * the gradient is ONE stride-1 2 x 2 convolution of dY with 4 C output channels (a block of C per parity class (y & 1, x & 1) of the input
 * pixel; four ninths of the products of the zero-dilated form), G [N][Hg][Wg][4 C] with Hg = (H - 1) / 2 + 2, Wg = (W - 1) / 2 + 2, and
 * dx[n][y][x][c] = G[n][y / 2 + 1][x / 2 + 1][((y & 1) * 2 + (x & 1)) * C + c] * scale[c] (scale may be NULL), zeroed where gate [N][H][W][C]
 * (may be NULL) is <= 0.  C % 4 == 0. */
int s2d_pixel_shuffle2_gate_f32(const float *G, int N, int Hg, int Wg, int C, int H, int W, const float *scale, const float *gate, float *dx,
                                hipStream_t stream);

/* dz = dy * (y > 0) * scale[channel]: the gradient through y = relu(z * scale + bias), the conv / linear epilogue
 * (FrozenBN scale; scale NULL = 1; y NULL = no ReLU); dres (may be NULL) = dy * (y > 0), the gradient of a residual
 * added before the ReLU.  n elements, C innermost. */
int s2d_relu_scale_backward_f32(const float *dy, const float *y, const float *scale, long n, int C, float *dz, float *dres,
                                hipStream_t stream);

/* out = a + (y > 0 ? g : 0) over n elements (n % 4 == 0; out may alias a): a ResNet stage output's own gradient g joins the gradient a that the
 * next block returned already gated by that output's ReLU (y = the stage output), in one pass. */
int s2d_relu_gate_add_f32(const float *a, const float *g, const float *y, long n, float *out, hipStream_t stream);

/* ---- training-step callers after the loss: optimizer + EMA ------------------------------ */

/* Tensor table shared by the two entry points (all arrays on the device): ptrs [ntensors][5] = {param, grad or NULL,
 * exp_avg, exp_avg_sq, ema target or NULL} (float32 each, numel[i] elements); the work list (chunk_tensor[c],
 * chunk_off[c]) names a tensor and an element offset, `chunk` elements (multiple of 4) per entry, one workgroup each.
 *
 * GradScaler.unscale_ + clip_grad_norm_ over all parameters (train_net_video.py:188-203, engine/train_loop.py:709-726)
 * without touching the gradients and without a host sync: normbuf[0] = || grads * inv_scale ||_2, normbuf[1] =
 * min(1, max_norm / (norm + 1e-6)) (1 if max_norm <= 0), normbuf[2] = 1 if any gradient is inf/nan else 0.
 * partial: double[nchunks] scratch (fixed slots, fixed-order reduction: reproducible). */
int s2d_optim_grad_norm_f32(const void *const *ptrs, const long *numel, const int *chunk_tensor, const long *chunk_off,
                            int nchunks, int chunk, float inv_scale, float max_norm, double *partial, float *normbuf,
                            hipStream_t stream);

/* torch.optim.AdamW.step (train_net_video.py:205-213; single-tensor formula, fp32, its operation order) on
 * g * inv_scale * normbuf[1], fused with the EMA teacher update ema = ema_m * ema + (1 - ema_m) * param
 * (engine/train_loop.py:754-764; ema_m < 0: no EMA).  hyper [ntensors][2] = {lr, weight_decay} as double; lr is
 * multiplied by lr_factor (the scheduler's factor).  bias_correction1 = 1 - beta1^step, bias_correction2_sqrt =
 * sqrt(1 - beta2^step), formed by the caller in double as torch does.  normbuf NULL: no clipping, no inf check;
 * normbuf[2] != 0: parameters and moments are left untouched (GradScaler skips the step), the EMA still runs.
 * Tensors whose grad pointer is NULL get the EMA only.  The (unscaled, clipped) gradients are not written back. */
int s2d_optim_adamw_ema_f32(const void *const *ptrs, const long *numel, const double *hyper, const int *chunk_tensor,
                            const long *chunk_off, int nchunks, int chunk, double lr_factor, double beta1, double beta2,
                            double eps, double bias_correction1, double bias_correction2_sqrt, float inv_scale, double ema_m,
                            const float *normbuf, hipStream_t stream);

/* Backward of s2d_point_loss_f32: same arguments, the workspace the forward call left behind
 * (thresholds, tie state, stored point samples, per-row sums), plus the loss weights.  grad_rows
 * [NL*B*min(Q,Nmax)*T][hm*wm] = d(w_mask * loss_mask + w_dice * loss_dice, all layers) / d(logit map of row
 * ((layer*B + b)*maxm + slot)*T + t); rows of unmatched slots and dropped frames are zero.  The selected points are
 * exactly the forward's (same threshold, same tie rule); their gradients are scattered through the bilinear taps into an
 * LDS tile (one map part at a time) and written out once.  bit_scratch: unused (may be NULL) -- the forward leaves every target
 * plane of the pass bit-packed in the workspace.  Only for passes whose active rows all took the stored-sample path. */
int s2d_point_loss_backward_f32(const float *mask_logits, const uint8_t *tgt, const int *tgt_count, const int *nonempty,
                                const int *idx_q, const int *idx_t, const int *n_match, const float *coords_over,
                                const float *coords_rand, uint64_t seed, int NL, int B, int Q, int ldq, int T, int hm, int wm,
                                int H, int W, int Nmax, int num_points, float oversample_ratio, float importance_ratio,
                                int drop_empty, float world_size, void *workspace, float w_mask, float w_dice, float *grad_rows,
                                unsigned int *bit_scratch, hipStream_t stream);

/* Test hook for the RNG (timing) mode of s2d_point_loss_f32: the oversampled points of rows [0, nrows) for `seed` on an
 * (hm, wm) logit map, uv [nrows][n_over][2], and every row's strata bounds [nrows][9], from the same device functions the loss
 * kernels use.  The generator is this library's own (the reference draws torch.rand, point_features.py:89-93; parity tests inject
 * those draws): i.i.d. uniform points generated per map part -- an exact multinomial split over the parts' v bands, then uniform
 * inside each band -- which is the law of i.i.d. uniform points.  scratch_list: nrows + 1 ints.  row0 must be 0. */
int s2d_point_loss_rng_points(uint64_t seed, int hm, int wm, int row0, int nrows, int n_over, float *uv, int *bounds, int *scratch_list,
                              hipStream_t stream);

/* d(w_ce * loss_labels)/d(class_logits) for one layer (same arguments as s2d_class_loss_f32) -> [B][Q][2] */
int s2d_class_loss_backward_f32(const float *class_logits, const int *idx_q, const int *n_match, int B, int Q, int maxm,
                                float eos_coef, float w_ce, float *d_class_logits, hipStream_t stream);

/* ---- keymask discovery (paths relative to /root/reference/keymask_ident) ------------------------------- */

/* pred_tracks_to_binary_masks(return_mask=False), cotracker_matching.py:453-503: tracks [T][Np][2] (x,y px) ->
 * masks u8 [T][H][W] (zeroed here); torch.round (half-to-even), keep 0<=x<W, 0<=y<H. */
int s2d_tracks_to_masks_u8(const float *tracks, int T, int Np, int H, int W, uint8_t *masks, hipStream_t stream);

/* extract_mask_matches inner loops (:665-719) for ALL frames and object ids at once: counts[t][id] =
 * #(point pixels of frame t whose nearest-resized (:687-689) id-map value == id), total[t] = #point pixels.
 * compute_point_mask_intersection (:640-662) is then counts[t][oid] / total[t] (0.0 if total == 0), formed on the
 * host in double exactly as the reference's python float division.  idmap int64 [T][Hi][Wi] (:176-209). */
int s2d_point_id_counts(const uint8_t *point_masks, const int64_t *idmap, int T, int H, int W, int Hi, int Wi, int max_id,
                        int *counts, int *total, hipStream_t stream);

/* get_segmentation_mask (keymask_ident/keymask_utils.py:37-67 == cotracker_matching.py:176-209) for a list of K (frame, object)
 * candidates: out u8 [K][H][W] = (idmap[frames[k]] == objs[k]) * 255, objs[k] == -1 selecting every non-background id.  frames /
 * objs are DEVICE int32 arrays (frame indices are not range-checked here: the caller built them from the same id map).  The
 * masks the cluster / group PNG trees hold (keymask_utils.py:100-126, cotracker_matching.py:402-431). */
int s2d_idmap_select_masks_u8(const int64_t *idmap, int T, int H, int W, const int *frames, const int *objs, int K, uint8_t *out,
                              hipStream_t stream);

/* presence[t][id] = id occurs in frame t (torch.unique at :680); u8 [T][max_id+1]. */
int s2d_idmap_presence_u8(const int64_t *idmap, int T, int Hi, int Wi, int max_id, uint8_t *presence, hipStream_t stream);

/* load_masks (cotracker_matching.py:22-84) after the PNG decode: rgb u8 [T][H][W][3] -> ids int64 [T][H][W], per frame
 * black = 0 and the other distinct colours 1..n in lexicographic (R,G,B) order; n_ids[T] = n.  At most 4096 distinct
 * colours per frame: *overflow is set to 1 otherwise (ids are then invalid).  workspace: s2d_color_ids_workspace_words(T)
 * 32-bit words. */
long s2d_color_ids_workspace_words(int T);
int s2d_color_masks_to_ids(const uint8_t *rgb, int T, int H, int W, unsigned int *workspace, int *n_ids, int64_t *ids,
                           int *overflow, hipStream_t stream);

/* cotracker_occlusions.py:359: curve[t] = mean_n(visibility[t][n] != 0). */
int s2d_visibility_curve_f32(const uint8_t *visibility, int T, int Np, float *curve, hipStream_t stream);

/* K1 (co-tracker is not in the reference tree; self-defined restatement, parity unpinned): local 4-D correlation
 * corr[t][n][i][j] = <bilinear(fmap[t], coords[t][n] + offset_i), support[n][j]> over C channels, offsets on the
 * (2r+1)^2 integer grid, zero padding.  fmap NHWC [T][H][W][C], coords [T][Np][2] in pixels of this level,
 * support [Np][(2r+1)^2][C], corr [T][Np][(2r+1)^2][(2r+1)^2].
 * Contract: C a multiple of 4 (16-B channel vectors), 0 <= r <= 3 (co-tracker's correlation radius is 3; a thread grid of
 * ceil(S/4)^2 <= 256 threads covers the S x S outputs), (S + 1)(C + 4) * 8 bytes of LDS <= 96 KB; anything else returns S2D_ERR_ARG. */
int s2d_local_corr_f32(const float *fmap_nhwc, const float *coords, const float *support, int T, int Np, int H, int W, int C,
                       int r, float *corr, hipStream_t stream);

/* ---- timing helpers for the benchmark's per-launch roofline (not on the data path) ------------------------ */

/* HIP events created with hipEventDisableSystemFence (handles are opaque integers, 0 = failure). */
long s2d_prof_event_create(void);
int s2d_prof_event_record(long ev, hipStream_t stream);
int s2d_prof_event_elapsed(long ev_start, long ev_end, double *out_ms_host);
int s2d_prof_event_destroy(long ev);

/* ---- data-side step before the hot path: clip augmentation and video copy-paste on the device ---------- */

/* The dataset mapper's augmentation chain (data_video/dataset_mapper.py:306-404 with augmentation.py:116-168: crop,
 * resize-shortest-edge, flip, brightness, contrast, rotation) as one resampling pass per frame.  frames u8 [T][3][H0][W0] ->
 * out u8 [T][3][H1][W1].  aug_frames_dev: DEVICE array of T records of 16 floats
 *   {a11, a12, a13, a21, a22, a23,  cx, cy, cw, ch,  bright, contrast, cmean, 0, 0, 0}:
 * output pixel centre (x + .5, y + .5) -> source point A . (x + .5, y + .5, 1); read only inside the crop rectangle
 * (outside: 0, the rotation fill; taps clamped to it); bilinear -> round -> brightness (img * w, clip, truncate) ->
 * contrast ((1 - w) * cmean + w * img, clip, truncate), the BlendTransform arithmetic.  cmean < 0: filled in by the call with
 * the crop's mean x bright (what RandomContrast measures).  detectron2 (absent from the reference tree) runs one image pass
 * per transform; the single-pass composition is this library's definition: parity unpinned. */
int s2d_aug_warp_frames_u8(const uint8_t *frames, int T, int H0, int W0, void *aug_frames_dev, int H1, int W1, uint8_t *out,
                           hipStream_t stream);
/* instance masks u8 [N][T][H0][W0] (0 / non-0) -> u8 [N][T][H1][W1] (0 / 1) under the same per-frame maps, nearest source
 * pixel (apply_segmentation). */
int s2d_aug_warp_masks_u8(const uint8_t *masks, int N, int T, int H0, int W0, const void *aug_frames_dev, int H1, int W1, uint8_t *out,
                          hipStream_t stream);

/* Sparse-mask densification of the trainer (`propagate_sparse_masks`, mask2former_video/engine/train_loop.py:30-156): all output
 * planes of a clip in one launch.  plan_dev: n_planes records {uint64 address of a source plane (bool / u8 [H, W], device memory),
 * int32 dx, int32 dy} (16 bytes each); out [n_planes][H][W] u8: out[j][y][x] = src_j[y + dy][x + dx] != 0 inside the frame, else 0 --
 * the reference's `_translate` (:58-68) for the instances an id's last sighting fills in, dx = dy = 0 for the instances a frame keeps.
 * Which planes, and the (dx, dy) draws in the reference's `random.randint` order, are the host's plan (s2d_amd/data/copy_paste.py). */
int s2d_shift_planes_u8(const void *plan_dev, int n_planes, int H, int W, uint8_t *out, hipStream_t stream);

/* One target frame of the trainer's video copy-paste loop exactly as written (engine/train_loop.py:445-570): the K copied masks
 * cur_masks [K][Hc][Wc] -- the source masks at the first frame, the PREVIOUS frame's canvas afterwards (the reference reassigns
 * copied_instances.gt_masks to the pasted canvas every frame, :512-514, so the masks are transformed cumulatively) -- and the source
 * frame [3][Hs][Ws] are resized to (h_new, w_new) with F.interpolate(bilinear, align_corners=False) (masks .bool(), image .byte()),
 * placed at (h_shift, w_shift): canvas u8 [K][H][W]; out_frame = alpha ? pasted source : target; out_tgt [N][H][W] = target masks
 * minus alpha; inter [K][N] / tarea [N] (before alpha: the ioy matrix of :517-527) and alive [N] (target areas after, :543) as
 * int32 counts.  K <= 64.  Used by s2d_amd/data/copy_paste.py, which keeps every decision of the loop on these integers. */
int s2d_copy_paste_frame_u8(const uint8_t *src_frame, int Hs, int Ws, const uint8_t *cur_masks, int K, int Hc, int Wc,
                            const uint8_t *tgt_frame, const uint8_t *tgt_masks, int N, int H, int W, int h_new, int w_new, int h_shift,
                            int w_shift, uint8_t *canvas, uint8_t *out_frame, uint8_t *out_tgt, int *inter, int *tarea, int *alive,
                            hipStream_t stream);

/* Video copy-paste, engine/train_loop.py:441-560: K source masks [K][Hs][Ws] and their frame [3][Hs][Ws] are resized to
 * (h_new, w_new) per target frame with F.interpolate(bilinear, align_corners=False) (image .byte(), masks .bool()), placed at
 * (h_shift, w_shift) and composited over the target clip: out_frames u8 [T][3][H][W]; out_masks u8 [N+K][T][H][W] = the N
 * target masks minus the pasted area, then the K pasted masks (all zero for copies with keep[k] == 0).  paste_frames_dev:
 * DEVICE int32 [T][4] = {h_new, w_new, h_shift, w_shift}; keep_dev: DEVICE u8 [K].
 * s2d_copy_paste_overlap: counts[k][n] = |pasted copy k AND target n| and area[n] = |target n| on frame 0, the integers behind
 * the "intersection over target area < 0.5" rule that decides keep (:515-527). */
int s2d_copy_paste_overlap(const uint8_t *tgt_masks, int N, int T, int H, int W, const uint8_t *src_masks, int K, int Hs, int Ws, int h_new,
                           int w_new, int h_shift, int w_shift, int *counts, int *area, hipStream_t stream);
int s2d_copy_paste_u8(const uint8_t *tgt_frames, const uint8_t *tgt_masks, int N, int T, int H, int W, const uint8_t *src_frame,
                      const uint8_t *src_masks, int K, int Hs, int Ws, const int *paste_frames_dev, const uint8_t *keep_dev,
                      uint8_t *out_frames, uint8_t *out_masks, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* S2D_HIP_H */
