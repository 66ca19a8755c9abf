// Internal: what the 4-wave kernels that keep their 256 accumulators per lane in AGPRs share (gemm_agpr.hip: register-staged
// operands; gemm_dma.hip: LDS-DMA operands): literal-AGPR MFMA statements, the compiler fence, and the epilogue out of a[0:255].
#pragma once
#include "gemm_epilogue.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

namespace {
template <int I, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sfor<I + 1, N>(static_cast<F&&>(f));
  }
}

#define A4_ALL_AGPRS                                                                                                     \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18",  \
      "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35",   \
      "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52",   \
      "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69",   \
      "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86",   \
      "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102",       \
      "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117",  \
      "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132",  \
      "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147",  \
      "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162",  \
      "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177",  \
      "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192",  \
      "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207",  \
      "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222",  \
      "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237",  \
      "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252",  \
      "a253", "a254", "a255"

#define A4_FENCE() asm volatile("" ::: A4_ALL_AGPRS)
// accumulator tile (nn, mi) = a[(nn*8 + mi)*4 .. +3]; nn = column tile 0..7 (16 columns each), mi = row tile 0..7
#define A4_MFMA(NN, MI, WF, AF)                                                                              \
  asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(WF), "v"(AF), "i"(((NN)*8 + (MI)) * 4), \
               "i"(((NN)*8 + (MI)) * 4 + 3))
#define A4_MFMA_Z(NN, MI, WF, AF)   /* first K-half of a tile: C = 0, no zeroing pass over the accumulators */     \
  asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, 0" ::"v"(WF), "v"(AF), "i"(((NN)*8 + (MI)) * 4), \
               "i"(((NN)*8 + (MI)) * 4 + 3))
// the same behind a short wait: for operands the compiler may have (re)assembled with VALU moves right in front of the statement
#define A4_MFMA_W(NN, MI, WF, AF)                                                                            \
  asm volatile("s_nop 3\n\tv_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(WF), "v"(AF), "i"(((NN)*8 + (MI)) * 4), \
               "i"(((NN)*8 + (MI)) * 4 + 3))
template <int IDX>
__device__ __forceinline__ float a4_acc_read() {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(x) : "i"(IDX));
  return x;
}

// What a 64-column slice of the wave's tile needs from memory besides the residual: bias / folded bias and LayerNorm
// column sums.  Slice 0's copy is requested one K-step BEFORE the epilogue (behind the MFMAs of the tile's last K-step),
// slice 1's at the start of the epilogue - no load latency is exposed for them.  EpiRows: what the lane's 8 ROW blocks need, the
// same for both slices - the (rstd, -mean*rstd) pairs of the LayerNorm-folded forms - requested with slice 0's columns, i.e.
// also one K-step ahead (round 5: they used to be the first thing the epilogue waited for - an L2 round trip per tile exposed).
struct EpiPre { EpiCols ec; };
struct EpiRows { float2 rst[8]; };
template <int EPI>
__device__ __forceinline__ void epi_prefetch(const GemmArgs& g, int n0, int ws, int q4, EpiPre& p) {
  g3_epi_cols<EPI>(g, n0, ws, q4, p.ec);
  if (epi_has_bias(EPI)) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = n0 + ws * 64 + ni * 16 + q4 * 4;
      p.ec.bias[ni] = *reinterpret_cast<const f32x4*>(g.bias + (n < g.N ? n : 0));
    }
  }
}
template <int EPI>
__device__ __forceinline__ void epi_prefetch_rows(const GemmArgs& g, int m0, int wm, int l15, EpiRows& pr) {
#pragma unroll
  for (int mi = 0; mi < 8; ++mi)
    pr.rst[mi] = epi_is_ln(EPI) ? g3_epi_rowstat(g, m0 + wm * 128 + mi * 16 + l15) : make_float2(1.f, 0.f);
}

// The residual tile of the wave comes in 16-byte loads in the STORE mapping (4 lanes cover 64 contiguous bytes of a row) through
// a ring of 8 x 2 registers.  Slice 0's eight row blocks are requested from INSIDE the tile's last K-step (round 5; epi_rload is
// called from the gaps behind its last barrier, one 16-byte load per gap that holds no DMA): the K-half-0 fragment registers are dead there, so the 64 ring registers cost
// nothing, and the first-touch latency of the residual rows (HBM: the residual stream of a whole video fits no cache) runs behind
// the last 40 MFMAs instead of in front of the epilogue.  Buffer loads: one descriptor per tile (origin = the tile's first
// element, range = to the end of the matrix, so rows past M read 0 without a predicate), one per-lane byte offset for the whole
// kernel, the (row block, column pair, slice) part in the scalar offset - no vector arithmetic and no branch inside the K-step.
// (Columns past N - only when N is not a multiple of 256 - read the next row's values; those elements are never stored and
// belong to no stored statistics slice.)
typedef __attribute__((ext_vector_type(4))) unsigned int epi_u32x4;
struct EpiRes { uint4 r[8][2]; __amdgpu_buffer_rsrc_t rs; unsigned lane; };
__device__ __forceinline__ void epi_res_setup(const GemmArgs& g, int m0, int n0, int wm, int wn, int l15, int q4, EpiRes& rr) {
  const long left = ((long)(g.M - m0) * g.N - n0) * 2;
  rr.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(g.R + (size_t)m0 * g.N + n0), 0, (int)(left < 0x40000000L ? left : 0x40000000L),
                                            0x00020000);
  rr.lane = ((unsigned)(wm * 128 + l15) * (unsigned)g.N + (unsigned)(wn * 128 + (q4 & 1) * 16 + (q4 >> 1) * 8)) * 2u;
}
__device__ __forceinline__ void epi_rload(const GemmArgs& g, int slice, int mi, int pr, EpiRes& rr) {   // slice: 0 / 1 of the wave
  const epi_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rr.rs, rr.lane, (mi * 16 * g.N + slice * 64 + pr * 32) * 2, 0);
  rr.r[mi][pr] = make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void epi_rload_rb(const GemmArgs& g, int slice, int mi, EpiRes& rr) {
  epi_rload(g, slice, mi, 0, rr);
  epi_rload(g, slice, mi, 1, rr);
}

// Epilogue of the wave's 128x128 tile held in a[0:255].  Residual forms: rres holds slice 0's residual rows on entry (see above).
// They are added to the accumulators ON THE MATRIX PIPE, in the load's own layout - D = sel x R + C with the 0/1 selection
// fragments of epi_resid_sel, two MFMAs per 16-byte load - one row block AHEAD of the v_accvgpr_reads of the epilogue proper, so
// the vector pipe neither unpacks nor adds nor un-swaps residual values (round 5: 448 of its instructions per tile; the loads of
// round 4's "residual on the matrix pipe" sat in the K-loop and cost more there than they saved here - these do not).  As soon as
// a row block's pair has gone into its MFMAs the registers are re-requested for slice 1.
template <int EPI, bool FULL>
__device__ __forceinline__ void agpr_epilogue(const GemmArgs& g, int m0, int n0, int wm, int wn, int l15, int q4,
                                              const EpiPre& p0, const EpiRows& prow, EpiRes& rres) {
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");   // last MFMA's result -> first v_accvgpr_read
  constexpr bool RES = EPI == GE_RESID || EPI == GE_RESID_ST;
  EpiPre p1;
  epi_prefetch<EPI>(g, n0, wn * 2 + 1, q4, p1);
  i32x4 sel0 = {0, 0, 0, 0}, sel1 = {0, 0, 0, 0};
  if (RES) { sel0 = epi_resid_sel(0, l15, q4); sel1 = epi_resid_sel(1, l15, q4); }
  auto radd = [&](auto nhs_, auto mi_) {   // accumulator tiles (nhs * 4 + 0..3, mi) += the residual rows of row block mi, slice nhs
    constexpr int nhs = decltype(nhs_)::value, mi = decltype(mi_)::value;
    sfor<0, 2>([&](auto pr_) {
      constexpr int pr = decltype(pr_)::value;
      const i32x4 rf = {(int)rres.r[mi][pr].x, (int)rres.r[mi][pr].y, (int)rres.r[mi][pr].z, (int)rres.r[mi][pr].w};
      const i32x4 s0 = sel0, s1 = sel1;   // (named copies: clang does not capture a variable that only an asm operand uses)
      A4_MFMA_W(nhs * 4 + 2 * pr, mi, s0, rf);
      A4_MFMA_W(nhs * 4 + 2 * pr + 1, mi, s1, rf);
    });
    A4_FENCE();
    if (nhs == 0) epi_rload_rb(g, 1, mi, rres);
  };
  if (RES) radd(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  sfor<0, 2>([&](auto nh_) {                          // the wave's two 64-column slices
    constexpr int nhs = decltype(nh_)::value;
    const int ws = wn * 2 + nhs;
    const EpiPre& p = nhs == 0 ? p0 : p1;
    sfor<0, 8>([&](auto mi_) {
      constexpr int mi = decltype(mi_)::value;
      if (RES) {   // the NEXT row block's residual goes onto its accumulators; this block's MFMAs are one row block old
        if constexpr (mi < 7) radd(std::integral_constant<int, nhs>{}, std::integral_constant<int, mi + 1>{});
        else if constexpr (nhs == 0) radd(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
      }
      f32x4 vv[4];
      sfor<0, 4>([&](auto ni_) {
        constexpr int ni = decltype(ni_)::value;
        constexpr int base = ((nhs * 4 + ni) * 8 + mi) * 4;
        vv[ni][0] = a4_acc_read<base>(); vv[ni][1] = a4_acc_read<base + 1>();
        vv[ni][2] = a4_acc_read<base + 2>(); vv[ni][3] = a4_acc_read<base + 3>();
      });
      A4_FENCE();
      g3_epi_row<EPI, true, FULL>(g, vv, p.ec, prow.rst[mi], m0 + wm * 128 + mi * 16 + l15, n0, ws, q4, nullptr);
    });
  });
}
// The epilogue with everything requested at its start (the form of rounds 2-4; the laboratory units of csrc/dev/ call it).
template <int EPI, bool FULL>
__device__ __forceinline__ void agpr_epilogue(const GemmArgs& g, int m0, int n0, int wm, int wn, int l15, int q4,
                                              const EpiPre& p0) {
  EpiRows prow;
  epi_prefetch_rows<EPI>(g, m0, wm, l15, prow);
  EpiRes rres;
  if (EPI == GE_RESID || EPI == GE_RESID_ST) {
    epi_res_setup(g, m0, n0, wm, wn, l15, q4, rres);
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) epi_rload_rb(g, 0, mi, rres);
  }
  agpr_epilogue<EPI, FULL>(g, m0, n0, wm, wn, l15, q4, p0, prow, rres);
}
}  // namespace
