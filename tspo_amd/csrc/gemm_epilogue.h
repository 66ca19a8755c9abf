// Internal: epilogue pieces shared by the bf16 GEMM kernels (gemm_bf16.hip, gemm_agpr.hip).
#pragma once
#include "gemm_bf16.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace {
__host__ __device__ constexpr bool epi_has_bias(int epi) {
  return epi == GE_BIAS || epi == GE_GELU || epi == GE_RESID || epi == GE_BIAS_LN || epi == GE_GELU_LN || epi == GE_RESID_ST ||
         epi == GE_BIAS_LN_HM;
}
__host__ __device__ constexpr bool epi_is_ln(int epi) { return epi == GE_BIAS_LN || epi == GE_GELU_LN || epi == GE_BIAS_LN_HM; }

#define GT_BM 128
#define GT_BN 128
#define GT_BK 64
#define G3_BM 256
#define G3_BN 256
#define G3_STAGE (G3_BM * 128 + G3_BN * 128)  // 65536 B: one K = 64 stage of a 256x256 tile (A rows | W rows, 128 B each)

// x * sigmoid(1.702 x) = x / (1 + 2^(-1.702 log2(e) x)): one v_exp_f32 + one v_rcp_f32 (1 ulp; the result is rounded
// to bf16 anyway) instead of a full-precision division sequence
__device__ __forceinline__ float quick_gelu_f(float x) {
  return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.4554669595930157f * x));
}

// Epilogue of one 256x256 tile, shared by the ring kernels.  A wave's columns come in 64-column slices (4 MFMA tiles
// of 16): g3_epi_cols holds what a slice needs once per tile, g3_epi_row finishes ONE 16-row block of a slice (this
// lane: row m, 4 x 4 consecutive columns) and g3_epilogue_t walks MI row blocks of accumulators held in VGPRs.
struct EpiCols { f32x4 lc[4]; f32x4 bias[4]; };   // bias[] only for the PRE (register-prefetched) form of g3_epi_row
template <int EPI>
__device__ __forceinline__ void g3_epi_cols(const GemmArgs& g, int n0, int wn, int q4, EpiCols& ec) {
  if (epi_is_ln(EPI)) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = n0 + wn * 64 + ni * 16 + q4 * 4;
      ec.lc[ni] = *reinterpret_cast<const f32x4*>(g.lnc + (n < g.N ? n : 0));
    }
  }
}
__device__ __forceinline__ float2 g3_epi_rowstat(const GemmArgs& g, int m) {   // (rstd, -mean*rstd) of row m
  return *reinterpret_cast<const float2*>(g.rstats + 2 * (size_t)(m < g.M ? m : g.M - 1));
}

// v[ni] = the 4 accumulators of column tile ni for row m (lane l15 of the 16-row block), rst = g3_epi_rowstat(m)
// PRE: bias / folded bias comes from ec.bias and the residual from rpre[ni] (this lane's 4 bf16 of column tile ni, MFMA
// layout) - both fetched by the caller ahead of time - instead of being loaded here.
// FULL: the caller guarantees that the whole 256x256 tile lies inside the matrix (no per-element range predicates).
template <int EPI, bool PRE = false, bool FULL = false>
__device__ __forceinline__ void g3_epi_row(const GemmArgs& g, f32x4 (&vv)[4], const EpiCols& ec, float2 rst, int m, int n0,
                                           int wn, int q4, const float* lbias, const uint2* rpre = nullptr) {
  constexpr bool LN = epi_is_ln(EPI);
  constexpr bool RES = EPI == GE_RESID || EPI == GE_RESID_ST;
  size_t orow = (size_t)m;
  int prow = 0;
  if (EPI == GE_PATCH) {
    const int f = m / g.P;
    prow = 1 + (m - f * g.P);
    orow = (size_t)f * (g.P + 1) + prow;
  }
  const float rs = rst.x, mu = rst.y;   // rstd, -mean * rstd
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  f32x2_t st_lo[4], st_hi[4];   // GE_RESID_ST: the 16 values of this lane as stored (bf16-rounded), kept for the second pass
  uint2 pk[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int n = n0 + wn * 64 + ni * 16 + q4 * 4;
    f32x4 v = vv[ni];
    const bool ok = FULL || (m < g.M && n < g.N);
    if (LN) {   // rstats holds (rstd, -mean*rstd): y = acc*rstd + (-mean*rstd)*c[n] + d[n], two FMAs per value
      const f32x4 dv = PRE ? ec.bias[ni] : *reinterpret_cast<const f32x4*>(lbias + (FULL || n < g.N ? n : 0));
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = fmaf(v[r], rs, fmaf(mu, ec.lc[ni][r], dv[r]));
    } else if (epi_has_bias(EPI)) {
      v += PRE ? ec.bias[ni] : *reinterpret_cast<const f32x4*>(lbias + (FULL || n < g.N ? n : 0));
    }
    if (EPI == GE_GELU || EPI == GE_GELU_LN) {   // quick_gelu_f on 4 values with the multiplies / adds packed
      typedef float f2 __attribute__((ext_vector_type(2)));
      const f2 kk = {-2.4554669595930157f, -2.4554669595930157f}, one = {1.f, 1.f};
      const f2 lo = {v[0], v[1]}, hi = {v[2], v[3]};
      const f2 tl = lo * kk, th = hi * kk;
      const f2 dl = f2{__builtin_amdgcn_exp2f(tl[0]), __builtin_amdgcn_exp2f(tl[1])} + one;
      const f2 dh = f2{__builtin_amdgcn_exp2f(th[0]), __builtin_amdgcn_exp2f(th[1])} + one;
      const f2 rl = lo * f2{__builtin_amdgcn_rcpf(dl[0]), __builtin_amdgcn_rcpf(dl[1])};
      const f2 rh = hi * f2{__builtin_amdgcn_rcpf(dh[0]), __builtin_amdgcn_rcpf(dh[1])};
      v = f32x4{rl[0], rl[1], rh[0], rh[1]};
    }
    const size_t o = orow * g.N + n;
    if (EPI == GE_PATCH && ok) v += *reinterpret_cast<const f32x4*>(g.pos + (size_t)prow * g.N + n);
    if (RES && (PRE || ok)) {
      const uint2 rv = PRE ? rpre[ni] : *reinterpret_cast<const uint2*>(g.R + o);
      v[0] += bf16_to_f32((uint16_t)(rv.x & 0xffff)); v[1] += bf16_to_f32((uint16_t)(rv.x >> 16));
      v[2] += bf16_to_f32((uint16_t)(rv.y & 0xffff)); v[3] += bf16_to_f32((uint16_t)(rv.y >> 16));
    }
    if (EPI == GE_F32) {
      if (ok) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.C) + o) = v;
    } else {
      pk[ni].x = pack_bf16x2(v[0], v[1]);
      pk[ni].y = pack_bf16x2(v[2], v[3]);
      if (EPI == GE_RESID_ST) {   // statistics of the values as stored (bf16-rounded): what the next GEMM reads
        const float keep = ok ? 1.f : 0.f;   // (FULL: constant 1)
        st_lo[ni] = f32x2_t{__uint_as_float(pk[ni].x << 16), __uint_as_float(pk[ni].x & 0xffff0000u)} * keep;
        st_hi[ni] = f32x2_t{__uint_as_float(pk[ni].y << 16), __uint_as_float(pk[ni].y & 0xffff0000u)} * keep;
      }
    }
  }
  if (EPI == GE_RESID_ST) {   // the row's 64 columns of this slice live in the 4 lanes that share l15
    // per-slice (mean, centred sum of squares): two passes over the 16 stored values of this lane (packed adds / fmas),
    // so the later combination of the N/64 slices (Chan et al.) is as robust as a two-pass LayerNorm.  The 4-lane sums
    // use v_permlane16/32_swap: no LDS round trip (ds_bpermute) in the middle of the epilogue.
    // Contraction is OFF in this block and every fused multiply-add is spelled out: this code is instantiated several times
    // per kernel (two slices of the whole-tile epilogue, the 64x64 remainder sub-tiles) and hipcc's choice of what to fuse differed
    // between the copies - 1 ulp in the sum of squares of odd slices (round 5, found by the frame-permutation test) - which
    // made a frame's features depend on WHERE in the batch it sat.  Now every copy performs the same operations.
#pragma clang fp contract(off)
    auto sum4 = [](float x) {
      const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
      x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
      const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
      return __uint_as_float(b[0]) + __uint_as_float(b[1]);
    };
    f32x2_t s2 = (st_lo[0] + st_hi[0]) + (st_lo[1] + st_hi[1]);
    s2 = s2 + ((st_lo[2] + st_hi[2]) + (st_lo[3] + st_hi[3]));
    const float smean = sum4(s2[0] + s2[1]) * (1.0f / 64.0f);
    const f32x2_t mm = {smean, smean};
    f32x2_t q2 = {0.f, 0.f};
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const f32x2_t d0 = st_lo[ni] - mm, d1 = st_hi[ni] - mm;
      q2 = __builtin_elementwise_fma(d0, d0, q2);
      q2 = __builtin_elementwise_fma(d1, d1, q2);
    }
    const float ssq = sum4(q2[0] + q2[1]);
    const int cslice = (n0 >> 6) + wn;
    if (q4 == 0 && (FULL || (m < g.M && cslice * 64 < g.N)))
      *reinterpret_cast<float2*>(g.spart + ((size_t)m * (g.N >> 6) + cslice) * 2) = make_float2(smean, ssq);
  }
  if (EPI != GE_F32) {
    // widen the stores: v_permlane16_swap exchanges the odd 16-lane rows of tile a with the even rows of tile
    // b, after which row q4 holds 16 contiguous bytes of tile (q4 & 1 ? b : a) at column (q4 >> 1) * 8
    // -> 16 instead of 32 store instructions per wave and tile (the epilogue is store-issue bound)
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const auto w0 = __builtin_amdgcn_permlane16_swap(pk[2 * pr].x, pk[2 * pr + 1].x, false, false);
      const auto w1 = __builtin_amdgcn_permlane16_swap(pk[2 * pr].y, pk[2 * pr + 1].y, false, false);
      const int n = n0 + wn * 64 + (2 * pr + (q4 & 1)) * 16 + (q4 >> 1) * 8;
      if (FULL || (m < g.M && n < g.N)) {
        uint4 st;
        st.x = w0[0]; st.y = w1[0]; st.z = w0[1]; st.w = w1[1];
        // (head-major form: the 64-column slice (n0 >> 6) + wn of the output is its own [M][64] matrix)
        const size_t o16 = EPI == GE_BIAS_LN_HM ? (((size_t)((n0 >> 6) + wn) * g.M + orow) << 6) + (n & 63) : orow * g.N + n;
        *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(g.C) + o16) = st;
      }
    }
  }
}

}  // namespace
