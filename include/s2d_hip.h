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

/* ---- dense contractions (fp32-input MFMA) ------------------------------------------------------ */

/* C[b][M,N] = act((A[b][M,K] * B[b][N,K]^T) * scale[N] + bias[N] + res[b][M,N]); scale/bias/res may be NULL.
 * Replaces every nn.Linear / 1x1 Conv2d on the path (e.g. ms_deform_attn.py:98-104,124; msdeformattn.py:122-131;
 * video_mask2former_transformer_decoder.py:99-111,164-168,193-205) and the mask-logit einsum
 * "bqc,btchw->bqthw" (video_mask2former_transformer_decoder.py:455).  K, lda, ldb multiples of 4. */
int s2d_gemm_nt_f32(const float *A, const float *B, float *C, int M, int N, int K, long lda, long ldb, long ldc,
                    int batch, long strideA, long strideB, long strideC, const float *scale, const float *bias,
                    const float *res, long ldr, long strideR, int relu, hipStream_t stream);

/* NHWC convolution as implicit GEMM: x [N,H,W,Cin] (Cin % 4 == 0), w [Cout][KH][KW][Cin],
 * y [N,Ho,Wo,Cout] = act(conv(x,w) * scale[Cout] + bias[Cout] + res).  Replaces detectron2 Conv2d+FrozenBN+ReLU
 * of the R50 trunk (build_resnet_backbone, call site kd_video_maskformer_model.py:132,135) and the FPN
 * 3x3 output conv (msdeformattn.py:264-281). */
int s2d_conv2d_nhwc_f32(const float *x, const float *w, float *y, int N, int H, int W, int Cin, int Cout, int KH,
                        int KW, int stride, int pad, const float *scale, const float *bias, const float *res,
                        int relu, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* S2D_HIP_H */
