// --dev builds only (round 5: production never launches it - the LDS-DMA kernel of gemm_dma.hip replaced it in round 4 - so it left
// the shipped library; `python -m tspo_amd.build --dev` links it back in as kernel variant 82 for A/B runs of tools/bench_gemm.py).
// bf16 MFMA GEMM for gfx950 (MI355X) with ONE wave per SIMD: persistent 256x256x64 tile, four waves (2x2), each wave owning
// 128x128 outputs = 8x8 MFMA 16x16x32 tiles = 256 fp32 accumulators per lane, held in the accumulator half of the unified
// register file under LITERAL names a[0:255].  hipcc cannot allocate a 256-accumulator kernel itself (it spills, also with
// "+a" inline-asm operands - DESIGN 4.1), so every MFMA, accumulator read and zeroing here is an inline-asm statement that
// names its AGPRs, and hipcc allocates only the architectural VGPRs around them.  Two guards keep it out of a[0:255]:
// A4_FENCE() statements (empty asm that "clobbers" all 256 AGPRs) between the MFMA groups, so nothing of the compiler's
// can live in an AGPR across them, and an audit of the generated code (tests/test_abi.py: no scratch access and no
// AGPR reference outside the asm statements).
//
// Operand path ("a7"): buffer_load_dwordx4 -> 64 staging VGPRs per lane -> ds_write_b128 into a 2 x 64 KB LDS ring, NOT
// LDS-DMA.  Measured (profiles/r2_*): the 8-wave LDS-DMA kernel of gemm_bf16.hip, a 4-wave LDS-DMA kernel with this wave
// layout (-33 % LDS reads per MFMA) and one with five 32 KB half-stages (96 KB in flight) all run a K-step in ~3600 cycles
// for ~2200 cycles of MFMA; what they share is 64 LDS-DMA wave-instructions per CU and K-step - an LDS-DMA piece holds up
// the issuing wave for 60-180 cycles, and the stream tops out near 25 B/clk/CU.  Ordinary vector loads do not pay that, and
// with the accumulators in AGPRs there are VGPRs to keep a whole K-step share (16 x 16 B per lane) in flight.
//   piece j (8 rows x 128 B; 0-7 = A, 8-15 = W):  S_j(x) = ds_write_b128 of piece j's registers into stage x's buffer,
//                                                 G_j(x) = buffer_load_dwordx4 of stage x's rows into the same registers
//   K-step it:  [A] 64 MFMAs on K-half 0 | fragments of K-half 1 | S_j(it+1), G_j(it+2) for the W pieces | lgkmcnt(0) s_barrier
//               [B] 64 MFMAs on K-half 1 | fragments of K-half 0 of stage it+1 | S_j(it+2), G_j(it+3) for the A pieces
// Every load has a full K-step to land before its ds_write; every fragment read half a K-step before its MFMAs; the only
// synchronisation is one barrier per K-step between four waves running the same in-order stream on separate SIMDs.
// LDS image as in gemm_bf16.hip: 128-byte rows, 16-byte chunk c of row r at chunk c ^ (r & 7) (here applied by the
// ds_write address: the 8 lanes of a row cover all 32 banks) -> conflict-free ds_read_b128 fragment reads.
#include "../gemm_agpr_common.h"

namespace {

// (The A/B forms of round 2-3 - A staged two K-steps ahead, relaxed waits, deferred stores, staggered starts, the s_memtime
// probes - were measured and are gone from the tree: profiles/r2_g_*, r3_a_gemm_path_probes.txt; code: git history up to 99eb127.)
template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm_bf16_a7_kernel(GemmArgs g, int tilesM, int ngrp) {
  __shared__ __attribute__((aligned(16))) char lds[2 * G3_STAGE];  // the ONLY LDS object
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q4 = lane >> 4;
  const int wm = wid >> 1, wn = wid & 1;
  const int nk = g.K / GT_BK;   // even, >= 2 (the dispatcher sends other shapes to the 8-wave kernel)
  const int xcd = blockIdx.x & 7, wl = blockIdx.x >> 3, nwl = gridDim.x >> 3;
  const int grp = xcd % ngrp, pset = xcd / ngrp, npset = 8 / ngrp, n_per = g.tilesN / ngrp;
  const int panels = (tilesM - pset + npset - 1) / npset;
  const int ntile_x = panels * n_per;
  const int my_tiles = wl < ntile_x ? (ntile_x - wl + nwl - 1) / nwl : 0;
  if (my_tiles == 0) return;
  A4_FENCE();   // claims a[0:255] for this kernel

  // ---- staging registers: this wave's 8 W pieces and 8 A pieces of a stage, each loaded one K-step before its ds_write
  //      (a second A set, two K-steps ahead, was measured in round 2: no gain, 32 more VGPRs) ----
  const int rin = lane >> 3, slot = lane & 7;
  const unsigned lane_goff = ((unsigned)rin * (unsigned)g.K + (unsigned)(slot << 3)) * 2u;   // plain 128-byte rows
  const int lane_woff = rin * 128 + ((slot ^ rin) << 4);                                      // swizzled LDS image
  const unsigned piece_stride = 8u * (unsigned)g.K * 2u;
  u32x4 sa0[8], sw_[8];
  // the stage the A registers / the W registers are loaded for NEXT: K-step inside the tile, tile, resource descriptor
  // (rows past the matrix end - and whole tiles past the last one - fall outside num_records and read as zeros)
  int a_kt = 0, a_s = wl, w_kt = 0, w_s = wl;
  auto rsrc_a = [&](int s_) {
    const int m0 = ((s_ / n_per) * npset + pset) * G3_BM;
    const long r = m0 < g.M ? ((long)(g.M - m0) * g.K * 2) : 0;
    return __builtin_amdgcn_make_buffer_rsrc((void*)(g.A + (size_t)(m0 < g.M ? m0 : 0) * g.K), 0,
                                             (int)(r < 0x40000000L ? r : 0x40000000L), 0x00020000);
  };
  auto rsrc_w = [&](int s_) {
    const int n0 = (grp * n_per + s_ % n_per) * G3_BN;
    const long r = n0 < g.N ? ((long)(g.N - n0) * g.K * 2) : 0;
    return __builtin_amdgcn_make_buffer_rsrc((void*)(g.W + (size_t)(n0 < g.N ? n0 : 0) * g.K), 0,
                                             (int)(r < 0x40000000L ? r : 0x40000000L), 0x00020000);
  };
  __amdgpu_buffer_rsrc_t a_rs = rsrc_a(a_s), w_rs = rsrc_w(w_s);
  auto adv_a = [&]() { if (++a_kt == nk) { a_kt = 0; a_s += nwl; a_rs = rsrc_a(a_s); } };
  auto adv_w = [&]() { if (++w_kt == nk) { w_kt = 0; w_s += nwl; w_rs = rsrc_w(w_s); } };
  auto gload_a = [&](auto q_, u32x4 (&sa)[8]) {
    constexpr int q = decltype(q_)::value;
    sa[q] = __builtin_amdgcn_raw_buffer_load_b128(a_rs, lane_goff + (unsigned)a_kt * (GT_BK * 2u),
                                                  (unsigned)(wid * 8 + q) * piece_stride, 0);
  };
  auto gload_w = [&](auto q_) {
    constexpr int q = decltype(q_)::value;
    sw_[q] = __builtin_amdgcn_raw_buffer_load_b128(w_rs, lane_goff + (unsigned)w_kt * (GT_BK * 2u),
                                                   (unsigned)(wid * 8 + q) * piece_stride, 0);
  };
  auto swrite_a = [&](auto q_, const u32x4 (&sa)[8], int buf) {
    constexpr int q = decltype(q_)::value;
    *reinterpret_cast<u32x4*>(lds + buf * G3_STAGE + (wid * 8 + q) * 1024 + lane_woff) = sa[q];
  };
  auto swrite_w = [&](auto q_, int buf) {
    constexpr int q = decltype(q_)::value;
    *reinterpret_cast<u32x4*>(lds + buf * G3_STAGE + G3_BM * 128 + (wid * 8 + q) * 1024 + lane_woff) = sw_[q];
  };

  // ---- prologue: stage 0 complete in buffer 0; A of stage 1 in buffer 1, W of stage 1 in the W registers; A of stage 2
  //      in the A registers ----
  sfor<0, 8>([&](auto q_) { gload_a(q_, sa0); gload_w(q_); });
  sfor<0, 8>([&](auto q_) { swrite_a(q_, sa0, 0); swrite_w(q_, 0); });
  adv_a(); adv_w();
  sfor<0, 8>([&](auto q_) { gload_a(q_, sa0); gload_w(q_); });
  sfor<0, 8>([&](auto q_) { swrite_a(q_, sa0, 1); });
  adv_a(); adv_w();   // w cursor -> stage 2 (loaded in [A] of K-step 0, after stage 1's W went to LDS)
  sfor<0, 8>([&](auto q_) { gload_a(q_, sa0); });
  adv_a();

  // ---- fragments: A double-buffered per K-half (2 x 32 VGPRs); W in ONE set of 8 x 4 VGPRs refilled in place: the
  //      fragment of column tile nn is dead after its 8 MFMAs, so the next K-half's fragment nn is read right behind them ----
  const int sw = l15 & 7;
  const int fbaseA = (wm * 128 + l15) * 128, fbaseW = G3_BM * 128 + (wn * 128 + l15) * 128;
  i32x4 fa0[8], fa1[8], fw[9];   // fw[8]: spare slot of the last column tile (see khalf)
  auto co_of = [&](int kk) { return ((kk * 4 + q4) ^ sw) << 4; };
  auto read_a = [&](const char* buf, int kk, i32x4 (&fa)[8]) {
    const int co = co_of(kk);
#pragma unroll
    for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const i32x4*>(buf + fbaseA + i * 2048 + co);
  };
  auto read_w_all = [&](const char* buf, int kk) {
    const int co = co_of(kk);
#pragma unroll
    for (int i = 0; i < 8; ++i) fw[i] = *reinterpret_cast<const i32x4*>(buf + fbaseW + i * 2048 + co);
  };
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0xC07F);
  read_a(lds, 0, fa0);
  read_w_all(lds, 0);
  __builtin_amdgcn_s_waitcnt(0xC07F);

  // one K-half: 64 MFMAs on (fa, fw); behind column tile nn's MFMAs its W fragment is refilled from (nbuf, nkk) = the next
  // K-half; na <- the next K-half's A fragments behind the first 16 MFMAs; one staging piece behind every 6th MFMA from
  // the 17th on: its registers go to LDS (ring buffer sbuf) and are reloaded for the cursor's stage
  // The W fragment of the LAST column tile cannot be refilled behind its own MFMAs - the read would be the last thing
  // before the barrier's lgkmcnt(0) and expose a full LDS latency every K-half - so it alternates between slots 7 and 8
  // (PAR = parity of the K-half: [A] computes from slot 7 and fills slot 8 early, [B] the other way round).
  auto khalf = [&](auto zero_, auto par_, i32x4 (&fa)[8], i32x4 (&na)[8], const char* nbuf, int nkk, auto&& piece) {
    constexpr bool ZERO = decltype(zero_)::value;
    constexpr int PAR = decltype(par_)::value;
    const int wco = co_of(nkk);
    sfor<0, 8>([&](auto nn_) {
      constexpr int nn = decltype(nn_)::value;
      constexpr int slot = nn == 7 ? 7 + PAR : nn;
      sfor<0, 8>([&](auto mi_) {
        constexpr int mi = decltype(mi_)::value;
        // the next K-half's A fragments: one read per column tile (fragment 7 rides with tile 3), so that every column
        // tile carries the same four memory instructions - A read, staging write, staging reload, W refill
        if (nn < 7 && mi == 1) na[nn] = *reinterpret_cast<const i32x4*>(nbuf + fbaseA + nn * 2048 + wco);
        if (nn == 3 && mi == 5) na[7] = *reinterpret_cast<const i32x4*>(nbuf + fbaseA + 7 * 2048 + wco);
        if (mi == 3) piece(nn_, std::integral_constant<int, 0>{});   // ds_write of staging piece nn ...
        if (mi == 6) piece(nn_, std::integral_constant<int, 1>{});   // ... and its reload, half a column tile later
        const i32x4 wf = fw[slot], af = fa[mi];
        if (ZERO) A4_MFMA_Z(nn, mi, wf, af); else A4_MFMA(nn, mi, wf, af);
      });
      if (nn < 7) fw[nn] = *reinterpret_cast<const i32x4*>(nbuf + fbaseW + nn * 2048 + wco);
      if (nn == 4) fw[8 - PAR] = *reinterpret_cast<const i32x4*>(nbuf + fbaseW + 7 * 2048 + wco);
      A4_FENCE();
    });
  };

  int it = 0, c_s = wl;
  auto kstep = [&](auto zero_, u32x4 (&sa)[8]) {
    const int cb = it & 1, nb = cb ^ 1;
    const char* cur = lds + cb * G3_STAGE;
    const char* nxt = lds + nb * G3_STAGE;
    // [A]: K-half 0; fragments of K-half 1 of this stage; W pieces: S_j(it+1) -> buffer nb, then G_j(it+2)
    khalf(zero_, std::integral_constant<int, 0>{}, fa0, fa1, cur, 1, [&](auto j_, auto ph_) {
      if (decltype(ph_)::value == 0) swrite_w(j_, nb); else gload_w(j_);
    });
    adv_w();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // stage it+1 complete in LDS; buffer cb fully read
    __builtin_amdgcn_s_waitcnt(0xC07F);   // (the same wait as a builtin: free at run time, keeps hipcc's wait model exact)
    // [B]: K-half 1; fragments of K-half 0 of stage it+1; A pieces of set `sa`: S_j(it+2) -> buffer cb, then G_j(it+4)
    khalf(std::false_type{}, std::integral_constant<int, 1>{}, fa1, fa0, nxt, 0, [&](auto j_, auto ph_) {
      if (decltype(ph_)::value == 0) swrite_a(j_, sa, cb); else gload_a(j_, sa);
    });
    adv_a();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    ++it;
  };
  for (int t = 0; t < my_tiles; ++t) {
    int m0 = ((c_s / n_per) * npset + pset) * G3_BM, n0 = (grp * n_per + c_s % n_per) * G3_BN;
    kstep(std::true_type{}, sa0);                              // first K-step: its K-half 0 starts the accumulators (C = 0)
    for (int kt = 1; kt < nk - 1; ++kt) kstep(std::false_type{}, sa0);
    EpiPre p0;                                                 // slice 0's epilogue inputs arrive behind the last K-step
    epi_prefetch<EPI>(g, n0, wn * 2, q4, p0);
    kstep(std::false_type{}, sa0);
    if (m0 + G3_BM <= g.M && n0 + G3_BN <= g.N) agpr_epilogue<EPI, true>(g, m0, n0, wm, wn, l15, q4, p0);
    else agpr_epilogue<EPI, false>(g, m0, n0, wm, wn, l15, q4, p0);
    c_s += nwl;
    // the next tile's first fragments again, AFTER the epilogue: the copies read in the last [B] are dead here, so
    // nothing but the staging registers stays live across the epilogue (stale data after the last tile, never used)
    const char* nbuf = lds + (it & 1) * G3_STAGE;
    read_a(nbuf, 0, fa0);
    read_w_all(nbuf, 0);
    __builtin_amdgcn_s_waitcnt(0xC07F);
  }
}


template <int EPI>
int launch_gemm_a7(GemmArgs g, hipStream_t st) {
  const int tilesM = (g.M + G3_BM - 1) / G3_BM;
  g.tilesN = (g.N + G3_BN - 1) / G3_BN;
  g.nwg = tilesM * g.tilesN;
  int ngrp = ((double)g.N * g.K * 2.0 > 4.0e6 && g.tilesN >= 8 && g.tilesN % 2 == 0) ? 2 : 1;
  if (g.ngrp > 0 && g.tilesN % g.ngrp == 0) ngrp = g.ngrp;
  hipLaunchKernelGGL((gemm_bf16_a7_kernel<EPI>), dim3(256), dim3(256), 0, st, g, tilesM, ngrp);
  return tspo::check_launch("gemm_bf16_a7");
}

template <int EPI>
int launch_a7_variant(const GemmArgs& g, hipStream_t st) {
  if ((g.K / GT_BK) % 2 != 0 || g.K < 2 * GT_BK)
    return tspo::set_err(TSPO_EINVAL, "gemm_agpr: K=%d must be a multiple of 128", g.K);
  return launch_gemm_a7<EPI>(g, st);
}
}  // namespace

static int gemm_bf16_agpr(int epi, const GemmArgs& g, hipStream_t st) {
  switch (epi) {
    case GE_BIAS: return launch_a7_variant<GE_BIAS>(g, st);
    case GE_GELU: return launch_a7_variant<GE_GELU>(g, st);
    case GE_RESID: return launch_a7_variant<GE_RESID>(g, st);
    case GE_F32: return launch_a7_variant<GE_F32>(g, st);
    case GE_PATCH: return launch_a7_variant<GE_PATCH>(g, st);
    case GE_BIAS_LN: return launch_a7_variant<GE_BIAS_LN>(g, st);
    case GE_GELU_LN: return launch_a7_variant<GE_GELU_LN>(g, st);
    case GE_BIAS_LN_HM: return launch_a7_variant<GE_BIAS_LN_HM>(g, st);
    case GE_RESID_ST: return launch_a7_variant<GE_RESID_ST>(g, st);
  }
  return tspo::set_err(TSPO_EINVAL, "gemm_agpr: bad epilogue %d", epi);
}

// reached from gemm_bf16.hip through this weak symbol (null in the shipped library)
extern "C" int tspo_lab_gemm_agpr(int epi, const GemmArgs* g, hipStream_t st) { return gemm_bf16_agpr(epi, *g, st); }
