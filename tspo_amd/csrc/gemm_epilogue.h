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
// PRE: bias / folded bias comes from ec.bias (fetched by the caller ahead of time) instead of being loaded here.
// FULL: the caller guarantees that the whole 256x256 tile lies inside the matrix (no per-element range predicates).
// Residual forms (round 5): the caller has ALREADY added the residual rows to the accumulators - on the matrix pipe
// (epi_resid_mfma below), which idles in the epilogue while the vector pipe was its limit: the unpack / add / un-swap of the
// residual and the two-pass row statistics were 1 400 of the 2 330 vector instructions of a tile's residual + statistics epilogue.
typedef __attribute__((ext_vector_type(4))) int epi_i32x4;
// a[i][k] = 1 where k-slot k of the residual fragment holds column 16 t + i of its 32-column pair, else 0 (bf16).  The residual
// fragment is the 16-byte load of the store mapping: lane (row l15, q4) holds columns (q4 & 1) * 16 + (q4 >> 1) * 8 + e, e = 0..7.
__device__ __forceinline__ epi_i32x4 epi_resid_sel(int t, int l15, int q4) {
  const bool on = (q4 & 1) == t && (q4 >> 1) == (l15 >> 3);
  const int e = l15 & 7;
  epi_i32x4 s;
#pragma unroll
  for (int d = 0; d < 4; ++d) s[d] = !on ? 0 : (e == 2 * d ? 0x00003f80 : (e == 2 * d + 1 ? 0x3f800000 : 0));
  return s;
}
// acc += the residual values of one 16-row x 16-column tile: D[i][j] = C[i][j] + sum_k sel[i][k] R[j][k] (one exact product per output)
__device__ __forceinline__ void epi_resid_mfma(f32x4& acc, const epi_i32x4 sel, const epi_i32x4 rfrag) {
  // (s_nop: hipcc may assemble an operand tuple with v_movs right in front of an asm statement, and nothing interlocks a VALU
  // write with the MFMA's operand read - it does not see the MFMA inside the asm; measured: garbage without it)
  asm volatile("s_nop 3\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(sel), "v"(rfrag));
}
template <int EPI, bool PRE = false, bool FULL = false>
__device__ __forceinline__ void g3_epi_row(const GemmArgs& g, f32x4 (&vv)[4], const EpiCols& ec, float2 rst, int m, int n0,
                                           int wn, int q4, const float* lbias) {
  constexpr bool LN = epi_is_ln(EPI);
  size_t orow = (size_t)m;
  int prow = 0;
  if (EPI == GE_PATCH) {
    const int f = m / g.P;
    prow = 1 + (m - f * g.P);
    orow = (size_t)f * (g.P + 1) + prow;
  }
  const float rs = rst.x, mu = rst.y;   // rstd, -mean * rstd
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
    if (EPI == GE_F32) {
      if (ok) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.C) + o) = v;
    } else {
      pk[ni].x = pack_bf16x2(v[0], v[1]);
      pk[ni].y = pack_bf16x2(v[2], v[3]);
    }
  }
  if (EPI == GE_RESID_ST) {
    // per-slice (mean, centred sum of squares) of the row's 64 values AS STORED (bf16-rounded: what the next GEMM reads), on the
    // matrix pipe: the lane's 16 packed values are two 16x16x32 operands (8 values each; which k-slot holds which column does not
    // matter for a sum); ones x V gives every lane its row's sum over the 4 lanes that share l15 (no cross-lane step), V x V the
    // row's sum of squares on the diagonal (lane q4 == l15 >> 2, register l15 & 3) - products of bf16 values are exact in fp32 and
    // the accumulation is fp32.  M2 = sum x^2 - sum x * mean: the cancellation is bounded by eps * sum x^2 / M2 per slice, and a
    // slice whose 64 columns are that constant contributes nothing to the row's variance (the slices are merged with Chan's
    // formula by stats_finalize, whose between-slice term is exact).  Every copy of this block (two slices of the whole-tile
    // epilogue, the 64x64 remainder sub-tiles) feeds the same operands in the same k-slots: bitwise the same statistics.
    const epi_i32x4 v01 = {(int)pk[0].x, (int)pk[0].y, (int)pk[1].x, (int)pk[1].y};
    const epi_i32x4 v23 = {(int)pk[2].x, (int)pk[2].y, (int)pk[3].x, (int)pk[3].y};
    const epi_i32x4 ones = {0x3f803f80, 0x3f803f80, 0x3f803f80, 0x3f803f80};
    f32x4 sm, sq;
    // (s_nop in front of each: the packs / the compiler's operand-tuple moves have just written the inputs, see epi_resid_mfma)
    asm volatile("s_nop 3\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(sm) : "v"(ones), "v"(v01));
    asm volatile("s_nop 3\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(sq) : "v"(v01), "v"(v01));
    asm volatile("s_nop 3\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(sm) : "v"(ones), "v"(v23));
    asm volatile("s_nop 3\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(sq) : "v"(v23), "v"(v23));
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(sm), "+v"(sq));   // MFMA result -> VALU read (the hazard recogniser does not see asm MFMAs)
    {
#pragma clang fp contract(off)
      const int l15 = threadIdx.x & 15;
      const float rsum = sm[0];
      const float smean = rsum * (1.0f / 64.0f);
      const float x2 = (l15 & 2) ? ((l15 & 1) ? sq[3] : sq[2]) : ((l15 & 1) ? sq[1] : sq[0]);
      const float ssq = fmaxf(__builtin_fmaf(-rsum, smean, x2), 0.f);
      const int cslice = (n0 >> 6) + wn;
      if (q4 == (l15 >> 2) && (FULL || (m < g.M && cslice * 64 < g.N)))
        *reinterpret_cast<float2*>(g.spart + ((size_t)m * (g.N >> 6) + cslice) * 2) = make_float2(smean, ssq);
    }
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
