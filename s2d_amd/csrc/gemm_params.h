// Parameter block shared by the dense-contraction kernels (gemm.hip: fp32-input MFMA, gemm_bf16.hip: split-bf16).
#pragma once
#include <hip/hip_runtime.h>

struct GemmParams {
    const float *A, *B;
    float *C;
    int M, N, K;
    long lda, ldb, ldc;
    long sA, sB, sC;  // batch strides (elements)
    const float *scale, *bias, *res;
    long ldr, sR;
    int res_rows, res_cols;   // residual row = row % res_rows (0: row); columns >= res_cols get no residual
    int relu;
    // optional gate of the epilogue value (the ReLU / dropout gate of a backward pass folded into the dgrad GEMM):
    // C = gate[row][col] > 0 ? value * gate_scale : 0, applied last; 16-B rows (vector epilogues of the split-fp16 kernels only)
    const float *gate;
    long ldg;
    float gate_scale;
    // implicit-GEMM convolution (A = NHWC input)
    int Hin, Win, Cin, Hout, Wout, KH, KW, stride, pad;
    // byte extents of one batch slice of A and B (buffer-descriptor bounds of the split-bf16 kernel)
    unsigned int bytesA, bytesB;
    // optional pre-split B (static weights): [N][kblocks][16 words fp16 hi | 16 words fp16 lo] per 32-wide k block, zero
    // padded past K; NULL = split on the fly
    const unsigned int *Bsplit;
    int kblocks;
    // optional pre-split A (an activation an upstream kernel already wrote in the same row-image layout, [M][kblocks][32 words]):
    // wave-specialised kernel only; its producers then copy instead of converting
    const unsigned int *Asplit;
    // fused dropout of the epilogue value (after scale / bias, before the residual; commutes with the ReLU): the three
    // nn.Dropout sites of the pixel decoder's encoder layers (msdeformattn.py:101-125).  drop_thresh 0 = off.
    unsigned int drop_thresh;   // element kept iff its 8 random bits >= drop_thresh (= round(p * 256), csrc/dropout.h)
    float drop_scale;           // 256 / (256 - drop_thresh)
    unsigned int drop_k0, drop_k1, drop_stream;   // Philox key (the call's seed) and the stream id of this dropout site
    unsigned int drop_row0;     // mask row of output row 0 (a launch over rows [r0, r1) of a larger activation passes r0)
    // wave-specialised kernel: 16 K floats of 0 and of 1 that stand in for an absent bias / scale vector (branch-free epilogue)
    const float *zeros, *ones;
};


int s2d_split_weights_launch(const float *W, int N, int K, long ldw, unsigned int *out, hipStream_t st, int bf16);
int s2d_launch_gemm_bf16x3(const GemmParams &p, bool conv, int batch, hipStream_t st, int f16);
// gemm_small.hip: one-round-trip kernel for launches of a handful of rows (M <= 256) against static pre-split weights
bool s2d_gemm_small_m_ok(const GemmParams &p, bool conv, int batch, int f16);
int s2d_launch_gemm_small_m(const GemmParams &p, hipStream_t st);
