// Internal interface between the bf16 MFMA GEMM kernels (gemm_bf16.hip) and the CLIP encoder (clip_vit.hip).
#pragma once
#include "common.h"

typedef uint16_t bf16_t;

enum { GE_BIAS = 0, GE_GELU = 1, GE_RESID = 2, GE_F32 = 3, GE_PATCH = 4 };

struct GemmArgs {
  const bf16_t* A; const bf16_t* W; const float* bias; const bf16_t* R; void* C;
  const float* pos;  // GE_PATCH: pos_emb [S, N]
  int M, N, K, tilesN, nwg, P;  // P: patches per frame (GE_PATCH row remap); P < 0 = ablation hooks (tests only)
  int variant;                  // 0 = auto; 1 = 128x128 2-stage; 2 = persistent 256x128 ring; 6 = persistent 256x256
  int ngrp;                     // 0 = auto N-group count per XCD
};

namespace tspo {
// C = A * W^T with epilogue `epi` (GE_*), enqueued on `st`; returns a TSPO_* code.
int gemm_bf16(int epi, const GemmArgs& g, hipStream_t st);
}
