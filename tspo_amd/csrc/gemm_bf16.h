// Internal interface between the bf16 MFMA GEMM kernels (gemm_bf16.hip) and the CLIP encoder (clip_vit.hip).
#pragma once
#include "common.h"

typedef uint16_t bf16_t;

// GE_BIAS_LN / GE_GELU_LN: the A operand is the RAW residual stream x and W carries the LayerNorm scale (W' = gamma o W);
// the epilogue applies the per-row statistics: y = rstd_m * acc - (mu_m * rstd_m) * lnc[n] + bias[n] with lnc[n] = sum_k W'[n,k]
// and bias = the folded bias (b + W beta).  GE_RESID_ST: GE_RESID that also writes, per row and 64-column slice, the (mean, centred sum
// of squares) of the bf16 values it stores: spart[m][N/64][2]  (LayerNorm folded into the GEMMs around it).
// GE_BIAS_LN_HM ("head-major"): GE_BIAS_LN whose bf16 output is stored by 64-column blocks, C[N/64][M][64] instead of C[M][N] - the
// q/k/v projection of the encoder: head h of part p (q, k, v) is block p*heads + h, so the 257 rows of a (frame, head) item are ONE
// contiguous 33 KB block for the attention kernel instead of 257 pieces of 128 B at a 6 KB pitch.  Same values, no extra arithmetic
// in the epilogue (block base + m * 64 in place of m * N).  N % 64 == 0.
enum { GE_BIAS = 0, GE_GELU = 1, GE_RESID = 2, GE_F32 = 3, GE_PATCH = 4, GE_BIAS_LN = 5, GE_GELU_LN = 6, GE_RESID_ST = 7,
       GE_BIAS_LN_HM = 8 };

struct GemmArgs {
  const bf16_t* A; const bf16_t* W; const float* bias; const bf16_t* R; void* C;
  const float* pos;  // GE_PATCH: pos_emb [S, N]
  const float* lnc;     // GE_*_LN: column sums of the folded weight [N]
  const float* rstats;  // GE_*_LN: (rstd, -mean*rstd) per row of A [M][2]
  float* spart;         // GE_RESID_ST: per-slice (mean, M2) row statistics out [M][N/64][2]
  int M, N, K, tilesN, nwg, P;  // P: patches per frame (GE_PATCH row remap)
  int variant;                  // 0 = auto; 1 = 128x128 small-problem kernel; 77 = 256x256 LDS-DMA operands; 83 = 77 without its remainder phase
  int ngrp;                     // 0 = auto N-group count per XCD
};

namespace tspo {
// C = A * W^T with epilogue `epi` (GE_*), enqueued on `st`; returns a TSPO_* code.
int gemm_bf16(int epi, const GemmArgs& g, hipStream_t st);
// true when gemm_bf16 would run this shape on the persistent 256x256 kernel (the only one with the *_LN / *_ST epilogues)
bool gemm_bf16_is_big(long M, int N, int K);
// gemm_dma.hip: the persistent 256x256 four-wave kernel, 256 accumulators per lane in AGPRs, LDS-DMA operands (variant 77; 83 = the
// same kernel with the remainder phase off: a partial last round of whole tiles, the round-4 behaviour; 67-76 = schedule A/Bs and
// probes of a --dev build, 82 = the register-staged kernel of rounds 2-3, also --dev only)
int gemm_bf16_dma(int epi, const GemmArgs& g, hipStream_t st);
}
