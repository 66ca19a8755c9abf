// bf16 MFMA GEMMs of the CLIP encoder for gfx950 (MI355X): C = A * W^T with fused epilogues.
// This file: the small-problem kernel (128x128x64, 2-stage LDS-DMA ring; projections, class-token rows, tiny models) and the
// dispatcher.  Large problems (M*N >= 256^3, K >= 128) run on the persistent 256x256x64 four-wave kernel with AGPR accumulators and
// LDS-DMA operands of gemm_dma.hip (production since round 4).  The register-staged kernel of rounds 2-3 lives in dev/gemm_agpr.hip
// (variant 82 of a --dev build); the 8-wave ring kernel of rounds 1-2 and the round-1..3 A/B kernels are gone from the tree (their
// measurements: DESIGN 4.1-4.3, profiles/r1_* .. r3_*; their code: git history up to 99eb127).
#include "gemm_epilogue.h"

// defined only by csrc/dev/gemm_agpr.hip (`python -m tspo_amd.build --dev`)
extern "C" __attribute__((weak)) int tspo_lab_gemm_agpr(int epi, const GemmArgs* g, hipStream_t st);

namespace {


// ===========================================================================
// GEMM  C[M,N] = A[M,K] * W[N,K]^T  (both operands K-contiguous, bf16)
// 128x128x64 workgroup tile, 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16x32.
// LDS image per operand tile: [128 rows][128 B], 16-B chunk c of row r stored
// at chunk (c ^ (r & 7)) - the XOR is applied on the global SOURCE address of
// the LDS-DMA (destination must stay lane-linear) and again on the ds_read.
// The MFMA is issued "swapped" (A-operand = W fragment, B-operand = A fragment)
// so that each lane ends up with 4 consecutive N for one M -> 8-byte stores.
// ===========================================================================
#define GT_STAGE_BYTES (2 * 128 * 128)  // A tile + W tile, 16 KB each

__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ G, int rows_total, int row0, int K, int kt,
                                           char* lds_tile, int wid, int lane) {
  // 16 pieces of 1 KB (8 rows x 128 B); wave `wid` issues pieces wid*4 .. wid*4+3
  const int rin = lane >> 3, slot = lane & 7;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int piece = wid * 4 + p;
    const int r = piece * 8 + rin;
    int gr = row0 + r;
    gr = gr < rows_total ? gr : rows_total - 1;
    const int c = slot ^ (r & 7);
    const bf16_t* src = G + (size_t)gr * K + (size_t)kt * GT_BK + c * 8;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds_tile + piece * 1024), 16, 0, 0);
  }
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) char lds[2 * GT_STAGE_BYTES];  // 2 stages x (A 16K | W 16K); the ONLY LDS object
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, q4 = lane >> 4;
  // XCD-aware, bijective remap: consecutive workgroups of one XCD walk the N tiles of one M row-panel
  const int bid = blockIdx.x;
  const int xcd = bid & 7, qq = g.nwg >> 3, rr = g.nwg & 7;
  const int wg = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  const int tm = wg / g.tilesN, tn = wg - tm * g.tilesN;
  const int m0 = tm * GT_BM, n0 = tn * GT_BN;
  const int wm = wid >> 1, wn = wid & 1;
  const int nk = g.K / GT_BK;

  f32x4 acc[4][4];  // [ni][mi]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  stage_tile(g.A, g.M, m0, g.K, 0, lds, wid, lane);
  stage_tile(g.W, g.N, n0, g.K, 0, lds + 16384, wid, lane);

  // per-lane read offsets (row-dependent swizzle is loop invariant)
  int offA[4], offW[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ra = wm * 64 + i * 16 + l15;
    const int rw = wn * 64 + i * 16 + l15;
    offA[i] = ra * 128;
    offW[i] = 16384 + rw * 128;
  }
  const int sw = l15 & 7;  // (row & 7) == (l15 & 7) because every row base is a multiple of 16

  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();  // stage kt landed (vmcnt(0) folded in by the compiler) and compute(kt-1) is done everywhere
    char* cur = lds + (kt & 1) * GT_STAGE_BYTES;
    if (kt + 1 < nk) {
      char* nxt = lds + ((kt + 1) & 1) * GT_STAGE_BYTES;
      stage_tile(g.A, g.M, m0, g.K, kt + 1, nxt, wid, lane);
      stage_tile(g.W, g.N, n0, g.K, kt + 1, nxt + 16384, wid, lane);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int co = (((kk * 4 + q4) ^ sw) << 4);
      bf16x8 fa[4], fw[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i] = *reinterpret_cast<const bf16x8*>(cur + offA[i] + co);
        fw[i] = *reinterpret_cast<const bf16x8*>(cur + offW[i] + co);
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ni], fa[mi], acc[ni][mi], 0, 0, 0);
    }
  }

  // epilogue: lane holds n = nb + q4*4 + r (r = 0..3), m = mb + l15
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int m = m0 + wm * 64 + mi * 16 + l15;
    if (m >= g.M) continue;
    size_t orow = (size_t)m;
    int prow = 0;
    if (EPI == GE_PATCH) {
      const int f = m / g.P;
      prow = 1 + (m - f * g.P);
      orow = (size_t)f * (g.P + 1) + prow;
    }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = n0 + wn * 64 + ni * 16 + q4 * 4;
      if (n >= g.N) continue;
      f32x4 v = acc[ni][mi];
      if (EPI == GE_BIAS || EPI == GE_GELU || EPI == GE_RESID) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(g.bias + n);
        v += bv;
      }
      if (EPI == GE_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = quick_gelu_f(v[r]);
      }
      if (EPI == GE_PATCH) {
        const f32x4 pv = *reinterpret_cast<const f32x4*>(g.pos + (size_t)prow * g.N + n);
        v += pv;
      }
      const size_t o = orow * g.N + n;
      if (EPI == GE_RESID) {
        const uint2 rv = *reinterpret_cast<const uint2*>(g.R + o);
        v[0] += bf16_to_f32((uint16_t)(rv.x & 0xffff)); v[1] += bf16_to_f32((uint16_t)(rv.x >> 16));
        v[2] += bf16_to_f32((uint16_t)(rv.y & 0xffff)); v[3] += bf16_to_f32((uint16_t)(rv.y >> 16));
      }
      if (EPI == GE_F32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.C) + o) = v;
      } else {
        uint2 pk;
        pk.x = pack_bf16x2(v[0], v[1]);
        pk.y = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(g.C) + o) = pk;
      }
    }
  }
}



template <int EPI>
int launch_gemm_v1(GemmArgs g, hipStream_t st) {
  const int tilesM = (g.M + GT_BM - 1) / GT_BM;
  g.tilesN = (g.N + GT_BN - 1) / GT_BN;
  g.nwg = tilesM * g.tilesN;
  hipLaunchKernelGGL((gemm_bf16_kernel<EPI>), dim3(g.nwg), dim3(256), 0, st, g);
  return tspo::check_launch("gemm_bf16");
}


// Kernel for "big" problems (gemm_bf16_is_big; K is a multiple of 64 at the ABI and at least 128 here): the four-wave AGPR
// kernel with LDS-DMA operands of gemm_dma.hip (variant 77; 83 = without its remainder phase, for A/B runs).  Variant 1 = the
// small-problem kernel above; 82 = the register-staged kernel of rounds 2-3, present only in a --dev build (weak symbol, null
// in the shipped library).  The shipped library reads no environment variables.
template <int EPI>
int launch_big(GemmArgs g, hipStream_t st) {
  if (g.variant == 0) g.variant = 77;
  if ((g.variant >= 67 && g.variant < 78) || g.variant == 83) return tspo::gemm_bf16_dma(EPI, g, st);
  if (g.variant == 82 && tspo_lab_gemm_agpr) return tspo_lab_gemm_agpr(EPI, &g, st);
  return tspo::set_err(TSPO_EINVAL, "gemm: kernel variant %d is not part of this build", g.variant);
}

template <int EPI>
int launch_gemm(GemmArgs g, hipStream_t st) {
  const bool big = tspo::gemm_bf16_is_big(g.M, g.N, g.K);
  if (g.variant == 1 || (g.variant == 0 && !big)) return launch_gemm_v1<EPI>(g, st);
  return launch_big<EPI>(g, st);
}

// LayerNorm-folded epilogues exist only in the persistent 256x256 kernels
template <int EPI>
int launch_gemm_ln(GemmArgs g, hipStream_t st) {
  if (!tspo::gemm_bf16_is_big(g.M, g.N, g.K) || g.N % 64)
    return tspo::set_err(TSPO_EINVAL, "gemm: LayerNorm-folded epilogue %d needs the 256x256 kernel (M=%d N=%d K=%d)", EPI, g.M, g.N, g.K);
  if (EPI == GE_RESID_ST ? !g.spart : !(g.lnc && g.rstats))
    return tspo::set_err(TSPO_EINVAL, "gemm: epilogue %d without its statistics pointers", EPI);
  return launch_big<EPI>(g, st);
}
}  // namespace

bool tspo::gemm_bf16_is_big(long M, int N, int K) {
  return M * N >= (long)256 * 256 * 256 && K >= 128 && N <= 4096;
}

int tspo::gemm_bf16(int epi, const GemmArgs& g, hipStream_t st) {
  switch (epi) {
    case GE_BIAS_LN: return launch_gemm_ln<GE_BIAS_LN>(g, st);
    case GE_GELU_LN: return launch_gemm_ln<GE_GELU_LN>(g, st);
    case GE_BIAS_LN_HM: return launch_gemm_ln<GE_BIAS_LN_HM>(g, st);
    case GE_RESID_ST: return launch_gemm_ln<GE_RESID_ST>(g, st);
    case GE_BIAS: return launch_gemm<GE_BIAS>(g, st);
    case GE_GELU: return launch_gemm<GE_GELU>(g, st);
    case GE_RESID: return launch_gemm<GE_RESID>(g, st);
    case GE_F32: return launch_gemm<GE_F32>(g, st);
    case GE_PATCH: return launch_gemm<GE_PATCH>(g, st);
  }
  return tspo::set_err(TSPO_EINVAL, "gemm: bad epilogue %d", epi);
}
