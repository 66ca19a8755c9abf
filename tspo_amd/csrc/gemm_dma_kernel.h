// Internal: the LDS-DMA operand-path GEMM kernel ("a9") as a template over its K-step schedule - instantiated by gemm_dma.hip
// (production schedule) and, in --dev builds, by dev/gemm_dma_lab.hip (schedule A/Bs; the instrumented copy of the kernel - no-DMA
// ablation, s_memtime probe - lives in dev/gemm_dma_lab_kernel.h, not here).
#pragma once
#include "gemm_agpr_common.h"

namespace {
// ===========================================================================
// "a9": the same tile, wave layout, AGPR accumulators and epilogue as a7, but the operands reach LDS by LDS-DMA
// (buffer_load_dwordx4 ... lds) in the schedule of the vendor library's hand-written 256x256x64 kernel for this chip
// (hipBLASLt `Custom_Cijk_Alik_Bljk_BBS_..._MT256x256x64_MI16x16x1`, read from its disassembly as a specification;
// DESIGN 4.4): the WHOLE K-step's fragments live in registers (4 x 32 VGPRs), so a ring buffer can be refilled two stages
// ahead as soon as its second K-half has been read into registers - no staging VGPRs, no ds_write, 16 instead of 32 staging
// instructions per wave and K-step:
//   K-step it (buffer cb = it & 1 holds stage it, nb stage it+1 in flight, fa0/fw0 = K-half 0 of stage it):
//     K-half 0 MFMAs | all 16 fragments of K-half 1 (A, W) <- cb | B1: cb may be refilled | DMA A(it+2), W(it+2) -> cb, one every
//                    6 gaps, on through K-half 1
//     K-half 1 MFMAs | ... DMA | vmcnt(n): stage it+1 landed, B3 | fragments of K-half 0 of stage it+1 <- nb
//   (the vendor kernel frees the A and the W region of cb with separate barriers - three per K-step; measured here: two are
//    0.5-2 % faster, ONE - everything behind the middle of the K-step - 5-8 % slower because it bunches the memory instructions)
// The XOR swizzle of the LDS image (chunk c of row r at c ^ (r & 7)) is applied on the global SOURCE address (an LDS-DMA
// destination is lane-linear).  DMA instructions are inline asm: hipcc's wait-count pass does not see them, so its own
// vmcnt waits (epilogue loads / stores) can only over-wait, and the waits that order DMA against the fragment reads are
// written here explicitly.
//
// WHAT goes into WHICH of the 128 gaps between a K-step's MFMAs is a compile-time table (A9Sched): a wave alone on its SIMD
// hides about three issue slots per 16-cycle MFMA, and the measured cost of a K-step follows the densest stretch of its
// memory instructions, not their number (a one-barrier schedule with 32 memory instructions behind 32 consecutive MFMAs ran
// 5-8 % slower than the three-barrier one; s_memtime probe: a barrier costs 20-40 cycles, the vmcnt wait 0).
// Measured (profiles/r5_a_gemm_tail_ab.txt, same box, interleaved): a partial last round of whole tiles costs 0.18-0.48 tile-times
// (not 1: its lone workgroups run on an otherwise idle chip), ONE sub-tile per workgroup 0.10-0.12 (out-proj 16.48 -> 16.12, fc2
// 16.36 -> 16.10 tile-times), THREE (QKV) 0.49 against 0.18, FOUR (fc1) 0.60 against 0.46 - so the remainder phase runs only
// when the left-over tiles make at most one sub-tile per workgroup.
#define A9_TAIL_MAX_SUBTILES 1

enum : int {
  OP_NONE = 0,
  OP_RA1 = 1,    // +j: A fragment j of K-half 1 <- current buffer
  OP_RW1 = 9,    // +j: W fragment j of K-half 1 <- current buffer
  OP_RA0 = 17,   // +j: A fragment j of K-half 0 of the NEXT stage <- other buffer
  OP_RW0 = 25,
  OP_MA = 33,    // +j: m0 <- LDS address of A piece j
  OP_DA = 41,    // +j: DMA of A piece j (m0 set by the OP_MA before it)
  OP_MW = 49,
  OP_DW = 57,
  OP_MDA = 65,   // +j: m0 and DMA in the same gap
  OP_MDW = 73,
  OP_B1 = 81, OP_B2 = 82, OP_B3 = 83
};
struct A9Sched {
  signed char op[128];   // gap after MFMA i of the K-step (0-63: K-half 0, 64-127: K-half 1)
  int vm;                // DMAs of this K-step issued before B3 (-> s_waitcnt vmcnt(vm) waits for the previous K-step's only)
};
// filled by a schedule's make(): put(gap, op); count_vm() after the last put
struct A9Builder {
  A9Sched t{};
  constexpr void put(int gap, int op) { t.op[gap] = (signed char)op; }
  constexpr A9Sched done() {
    for (int i = 0; i < 128 && t.op[i] != OP_B3; ++i)
      if ((t.op[i] >= OP_DA && t.op[i] < OP_MW) || (t.op[i] >= OP_DW && t.op[i] < OP_MDA) || (t.op[i] >= OP_MDA && t.op[i] < OP_B1)) ++t.vm;
    return t;
  }
};

// The production schedule (round 4; every alternative measured against it lives in dev/gemm_dma_lab.hip, results in
// profiles/r4_a_gemm_dma_schedules_and_probes.txt): the 16 K-half-1 fragments in the first 16 gaps, ONE barrier at gap 24 frees
// both regions of the buffer, one DMA every 6 gaps from gap 25 (A pieces, then W pieces; around B3 and the next stage's reads),
// B3 at gap 85, the next stage's K-half-0 fragments in every second gap behind it.
struct A9ScheduleProduction {
  static constexpr A9Sched make() {
    A9Builder b;
    for (int j = 0; j < 8; ++j) { b.put(j, OP_RA1 + j); b.put(8 + j, OP_RW1 + j); }
    b.put(24, OP_B1);
    const int d[16] = {25, 31, 37, 43, 49, 55, 61, 67, 73, 79, 91, 97, 103, 109, 115, 121};
    for (int j = 0; j < 8; ++j) { b.put(d[j], OP_MDA + j); b.put(d[8 + j], OP_MDW + j); }
    b.put(85, OP_B3);
    for (int j = 0; j < 8; ++j) { b.put(86 + 2 * j, OP_RA0 + j); b.put(102 + 2 * j, OP_RW0 + j); }
    return b.done();
  }
};

// Where the last K-step of a tile issues residual load i (0-15) of the residual forms: behind B3 (gap 85), in gaps without a DMA.
constexpr int a9_resid_slot(int gap) {
  constexpr int at[16] = {86, 87, 89, 93, 95, 99, 101, 105, 107, 111, 113, 117, 119, 123, 125, 127};
  for (int i = 0; i < 16; ++i)
    if (at[i] == gap) return i;
  return -1;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Remainder phase (round 5).  Whole 256x256 tiles are handed out in whole ROUNDS only: XCD set x owns ntile_x tiles, its 32
// workgroups run floor(ntile_x / 32) of them each.  What is left over (out-proj / fc2 of the encoder: 4 112 tiles = 16.06 rounds,
// i.e. 16 tiles for which the persistent loop used to run a 17th round at 1/16 occupancy) is cut into 64x64 SUB-tiles - 16 per
// tile - and spread over ALL workgroups of the launch: no cross-workgroup reduction, every output element keeps its K order
// (bitwise the results of the whole-tile path).  A sub-tile: 4 waves x (16 rows x 64 columns) - the shape g3_epi_row finishes, so
// every epilogue form incl. the per-64-column-slice row statistics works unchanged - operands by LDS-DMA into an 8-deep ring of
// 16 KB stages (the whole 128 KB), 7 stages in flight, one counted vmcnt + one barrier per K-step.
// ---------------------------------------------------------------------------------------------------------------------------
#define A9T_STAGE 16384        // 64 A rows | 64 W rows, 128 B each
#define A9T_DEPTH 8

struct A9Remainder { int total; int pre[9]; int rounds[8]; };   // tiles left over per XCD set (prefix sums), whole rounds per set
__host__ __device__ inline A9Remainder a9_remainder(int tilesM, int tilesN, int ngrp, int nwl) {
  A9Remainder r{};
  const int npset = 8 / ngrp, n_per = tilesN / ngrp;
  for (int x = 0; x < 8; ++x) {
    const int pset = x / ngrp;
    const int panels = tilesM > pset ? (tilesM - pset + npset - 1) / npset : 0;
    const int nt = panels * n_per;
    r.rounds[x] = nt / nwl;
    r.pre[x + 1] = r.pre[x] + nt % nwl;
  }
  r.total = r.pre[8];
  return r;
}

template <int EPI>
__device__ __forceinline__ void a9_tail_subtile(const GemmArgs& g, char* lds, int m_s, int n_s, int wid, int lane) {
  const int l15 = lane & 15, q4 = lane >> 4, rin = lane >> 3, slot = lane & 7;
  const int nk = g.K / GT_BK;
  // waves 0, 1 bring the A rows (pieces 0-7 of a stage), waves 2, 3 the W rows (pieces 8-15): ONE descriptor per wave
  const bool isw = wid >= 2;
  const bf16_t* base = isw ? g.W + (size_t)n_s * g.K : g.A + (size_t)m_s * g.K;
  const long left = (long)(isw ? g.N - n_s : g.M - m_s) * g.K * 2;
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(left < 0x40000000L ? left : 0x40000000L), 0x00020000);
  const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0, 0x00020000);   // past the last K-step
  const unsigned lane_goff = ((unsigned)rin * (unsigned)g.K + (unsigned)((slot ^ rin) << 3)) * 2u;
  const unsigned piece_stride = 8u * (unsigned)g.K * 2u;
  const unsigned so0 = (unsigned)(wid & 1) * 4u * piece_stride;
  const unsigned lds_w = (unsigned)(size_t)lds + (unsigned)wid * 4096u;
  auto issue = [&](int st) {   // this wave's 4 pieces of stage st -> ring slot st % DEPTH
    const unsigned bb = lds_w + (unsigned)(st & (A9T_DEPTH - 1)) * A9T_STAGE;
    const unsigned vo = lane_goff + (unsigned)st * (GT_BK * 2u);
    const __amdgpu_buffer_rsrc_t r = st < nk ? rs : rz;
#define A9T_PIECE(Q)                                                                                  \
    asm volatile("s_add_u32 m0, %0, %1" ::"s"(bb), "i"((Q) * 1024) : "scc");                             \
    asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(vo), "s"(r), "s"(so0 + (Q) * piece_stride) : "memory")
    A9T_PIECE(0); A9T_PIECE(1); A9T_PIECE(2); A9T_PIECE(3);
#undef A9T_PIECE
  };
  // the previous user of the LDS (main loop / previous sub-tile) is done everywhere, and nothing of this wave is in flight
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
  for (int st = 0; st < A9T_DEPTH - 1; ++st) issue(st);
  f32x4 acc[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) acc[ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int sw = l15 & 7;
  const int offA = (wid * 16 + l15) * 128, offW = 8192 + l15 * 128;
  for (int kt = 0; kt < nk; ++kt) {
    // stage kt landed (6 younger stages x 4 DMAs may fly) - for every wave; and everybody is done with ring slot (kt - 1) % DEPTH
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"(4 * (A9T_DEPTH - 2)) : "memory");
    issue(kt + A9T_DEPTH - 1);
    const char* cur = lds + (kt & (A9T_DEPTH - 1)) * A9T_STAGE;
    bf16x8 fa[2], fw[2][4];      // both K-halves first (one LDS latency per K-step), then the 8 MFMAs
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      const int co = ((kh * 4 + q4) ^ sw) << 4;
      fa[kh] = *reinterpret_cast<const bf16x8*>(cur + offA + co);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) fw[kh][ni] = *reinterpret_cast<const bf16x8*>(cur + offW + ni * 2048 + co);
    }
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)   // inline asm with VGPR accumulators: hipcc must not pick AGPRs in this kernel (it would for the builtin)
        // (s_nop: see epi_resid_mfma - an operand tuple the compiler assembles with v_movs in front of the statement is not interlocked)
        asm volatile("s_nop 3\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[ni]) : "v"(fw[kh][ni]), "v"(fa[kh]));
  }
  const int m = m_s + wid * 16 + l15;
  if (EPI == GE_RESID || EPI == GE_RESID_ST) {   // the residual rows onto the accumulators, exactly as the whole-tile epilogue does it
    const i32x4 sel0 = epi_resid_sel(0, l15, q4), sel1 = epi_resid_sel(1, l15, q4);
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const int n = n_s + (2 * pr + (q4 & 1)) * 16 + (q4 >> 1) * 8;
      const uint4 rv = (m < g.M && n < g.N) ? *reinterpret_cast<const uint4*>(g.R + (size_t)m * g.N + n) : make_uint4(0u, 0u, 0u, 0u);
      const i32x4 rf = {(int)rv.x, (int)rv.y, (int)rv.z, (int)rv.w};
      epi_resid_mfma(acc[2 * pr], sel0, rf);
      epi_resid_mfma(acc[2 * pr + 1], sel1, rf);
    }
  }
  // last MFMA's result -> first VALU read (the hazard recogniser does not see asm MFMAs); the accumulators are operands of the
  // wait so that no read of them can be scheduled in front of it
  asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])::"memory");
  EpiCols ec;
  g3_epi_cols<EPI>(g, n_s, 0, q4, ec);
  const float2 rst = epi_is_ln(EPI) ? g3_epi_rowstat(g, m) : make_float2(1.f, 0.f);
  g3_epi_row<EPI, false, false>(g, acc, ec, rst, m, n_s, 0, q4, g.bias);
}

// tail: 0 = every tile of an XCD set is a whole tile (the round-4 behaviour: a partial last round); 1 = whole rounds + the
// remainder phase above.  launch_gemm_a9 chooses (g.variant 83 forces 0).
template <int EPI, class SCHED>
__global__ __launch_bounds__(256, 1) void gemm_bf16_a9_kernel(GemmArgs g, int tilesM, int ngrp, int tail) {
  __shared__ __attribute__((aligned(16))) char lds[2 * G3_STAGE];  // the ONLY LDS object
  constexpr A9Sched SC = SCHED::make();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q4 = lane >> 4;
  const int wm = wid >> 1, wn = wid & 1;
  const int nk = g.K / GT_BK;   // >= 2
  const int xcd = blockIdx.x & 7, wl = blockIdx.x >> 3, nwl = gridDim.x >> 3;
  const int grp = xcd % ngrp, pset = xcd / ngrp, npset = 8 / ngrp, n_per = g.tilesN / ngrp;
  const int panels = (tilesM - pset + npset - 1) / npset;
  const int ntile_x = panels * n_per;
  const int lim = tail ? (ntile_x / nwl) * nwl : ntile_x;   // tiles of this XCD set that are run as whole tiles
  const int my_tiles = wl < lim ? (lim - wl + nwl - 1) / nwl : 0;
  if (my_tiles == 0 && !tail) return;
  A4_FENCE();   // claims a[0:255] for this kernel
  // (Measured in round 5 and not kept - profiles/r5_e_gemm_epilogue_experiments.txt: workgroup b started b * w / 256 cycles late, w = 8k
  // ... 65k cycles, so that the 256 epilogues of a round of tiles - 33-67 MB of stores and residual loads - do not reach the memory
  // system in the same few microseconds: no shape gets faster at any window, every shape pays the window.  The epilogue is a
  // per-CU limit, not a chip-level burst.)

  if (my_tiles > 0) {
  // ---- LDS-DMA: piece P = wid*8 + q of a region = rows 8P..8P+7 (1 KB); lane (rin, slot) brings global chunk slot ^ rin.
  //      Address split: voffset = lane part + K-step (ONE v_add per K-step), soffset = piece (loop-invariant SGPRs, the same
  //      for A and W), m0 = LDS destination (one s_add with a literal per piece) ----
  const int rin = lane >> 3, slot = lane & 7;
  unsigned lane_goff = ((unsigned)rin * (unsigned)g.K + (unsigned)((slot ^ rin) << 3)) * 2u;
  unsigned piece_stride = 8u * (unsigned)g.K * 2u;
  unsigned lds0 = (unsigned)(size_t)lds + (unsigned)wid * 8192u;
  unsigned soff0 = (unsigned)wid * 8u * piece_stride;
  int d_kt = 0, d_s = wl;   // the stage the NEXT K-step's DMA brings: K-step inside the tile, tile
  auto rsrc_a = [&](int s_) {   // (tiles at or past `lim` - the two stages requested past the last whole tile - get zero-length resources)
    const int m0 = ((s_ / n_per) * npset + pset) * G3_BM;
    const bool in = s_ < lim && m0 < g.M;
    const long r = in ? ((long)(g.M - m0) * g.K * 2) : 0;
    return __builtin_amdgcn_make_buffer_rsrc((void*)(g.A + (size_t)(in ? m0 : 0) * g.K), 0,
                                             (int)(r < 0x40000000L ? r : 0x40000000L), 0x00020000);
  };
  auto rsrc_w = [&](int s_) {
    const int n0 = (grp * n_per + s_ % n_per) * G3_BN;
    const bool in = s_ < lim && n0 < g.N;
    const long r = in ? ((long)(g.N - n0) * g.K * 2) : 0;
    return __builtin_amdgcn_make_buffer_rsrc((void*)(g.W + (size_t)(in ? n0 : 0) * g.K), 0,
                                             (int)(r < 0x40000000L ? r : 0x40000000L), 0x00020000);
  };
  __amdgpu_buffer_rsrc_t a_rs = rsrc_a(d_s), w_rs = rsrc_w(d_s);
  auto adv_d = [&]() {
    if (++d_kt == nk) {
      asm volatile("" ::: "memory");   // keeps the tile switch (two divisions, two descriptors) a BRANCH: if-converted it runs every K-step
      d_kt = 0; d_s += nwl; a_rs = rsrc_a(d_s); w_rs = rsrc_w(d_s);
    }
  };
  // (named copies inside the lambdas: clang does not capture a variable that only an asm operand uses)
  auto set_m0 = [&](auto q_, auto isw_, unsigned bufbase) {
    constexpr int off = decltype(q_)::value * 1024 + (decltype(isw_)::value ? G3_BM * 128 : 0);
    const unsigned b = bufbase;
    asm volatile("s_add_u32 m0, %0, %1" ::"s"(b), "i"(off) : "scc");
  };
  auto dma = [&](auto q_, auto isw_, unsigned voff) {
    constexpr int q = decltype(q_)::value;
    constexpr bool isw = decltype(isw_)::value;
    const unsigned vo = voff, so = soff0 + q * piece_stride;
    const __amdgpu_buffer_rsrc_t rs = isw ? w_rs : a_rs;
    // default cache policy on purpose: `nt` on the A or the W stream cuts the QKV form's L2-side fetch by a third (3.7 -> 2.5 GB per
    // launch) and is 2-6 % SLOWER on every shape; sc1 / sc0 sc1 change nothing (profiles/r4_g_fetch_calibration_and_cache_policy.txt)
    asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(vo), "s"(rs), "s"(so) : "memory");
  };

  // ---- fragments: both K-halves of a stage, A and W: 4 x 8 x 4 VGPRs ----
  const int sw = l15 & 7;
  const int fbaseA = (wm * 128 + l15) * 128, fbaseW = G3_BM * 128 + (wn * 128 + l15) * 128;
  const int co0 = (q4 ^ sw) << 4, co1 = ((4 + q4) ^ sw) << 4;
  i32x4 fa0[8], fa1[8], fw0[8], fw1[8];
  auto ldfrag = [&](const char* p) { return *reinterpret_cast<const i32x4*>(p); };

  // ---- prologue: stages 0 and 1 in flight, stage 0 landed, its K-half 0 in registers ----
  sfor<0, 2>([&](auto b_) {
    const unsigned vo = lane_goff + (unsigned)d_kt * (GT_BK * 2u), bb = lds0 + decltype(b_)::value * G3_STAGE;
    sfor<0, 8>([&](auto q_) {
      set_m0(q_, std::false_type{}, bb); dma(q_, std::false_type{}, vo);
      set_m0(q_, std::true_type{}, bb); dma(q_, std::true_type{}, vo);
    });
    adv_d();
  });
  asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
#pragma unroll
  for (int i = 0; i < 8; ++i) { fa0[i] = ldfrag(lds + fbaseA + i * 2048 + co0); fw0[i] = ldfrag(lds + fbaseW + i * 2048 + co0); }

  int it = 0, c_s = wl;
  constexpr bool RES = EPI == GE_RESID || EPI == GE_RESID_ST;
  EpiRes rres;   // (the residual forms: slice 0's rows, requested from the last K-step of the tile)
  auto kstep = [&](auto zero_, auto last_) {
    constexpr bool ZERO = decltype(zero_)::value, LAST = decltype(last_)::value;
    const int cb = it & 1;
    const char* cur = lds + cb * G3_STAGE;
    const char* nxt = lds + (cb ^ 1) * G3_STAGE;
    const unsigned vo = lane_goff + (unsigned)d_kt * (GT_BK * 2u), bb = lds0 + (unsigned)cb * G3_STAGE;
    sfor<0, 16>([&](auto grp8_) {   // 16 groups of 8 MFMAs: group nn = grp8 & 7 of K-half grp8 >> 3 (see below for what a group walks)
      constexpr int kh = decltype(grp8_)::value >> 3, nn = decltype(grp8_)::value & 7;
      sfor<0, 8>([&](auto mi_) {
        constexpr int mi = decltype(mi_)::value;
        constexpr int gap = kh * 64 + nn * 8 + mi;
        constexpr int op = SC.op[gap];
        {
          // Issue order of a K-half's 64 MFMAs (round 5): group g keeps ONE A-row fragment and walks the eight W fragments, up in
          // even groups and down in odd ones, so that every MFMA shares an operand register quad with the one before it - also across
          // the group boundary, where the straight order (W fragment fixed, A fragments 0..7, then both change) switched both.  Same
          // accumulation order per output, bitwise the same results; +0.4 % frames/s in 9 of 9 alternating same-box rounds against
          // the straight order, three zig-zag forms alike (profiles/r5_e_gemm_epilogue_experiments.txt #6) - the chip is limited by
          // power under this kernel, and operand traffic is power.
          constexpr int wi = (nn & 1) ? 7 - mi : mi;
          const i32x4 wf = kh ? fw1[wi] : fw0[wi], af = kh ? fa1[nn] : fa0[nn];
          if (ZERO && kh == 0) A4_MFMA_Z(wi, nn, wf, af); else A4_MFMA(wi, nn, wf, af);
        }
        if constexpr (op >= OP_RA1 && op < OP_RA1 + 8) fa1[op - OP_RA1] = ldfrag(cur + fbaseA + (op - OP_RA1) * 2048 + co1);
        if constexpr (op >= OP_RW1 && op < OP_RW1 + 8) fw1[op - OP_RW1] = ldfrag(cur + fbaseW + (op - OP_RW1) * 2048 + co1);
        if constexpr (!LAST && op >= OP_RA0 && op < OP_RA0 + 8) fa0[op - OP_RA0] = ldfrag(nxt + fbaseA + (op - OP_RA0) * 2048 + co0);
        if constexpr (!LAST && op >= OP_RW0 && op < OP_RW0 + 8) fw0[op - OP_RW0] = ldfrag(nxt + fbaseW + (op - OP_RW0) * 2048 + co0);
        // last K-step of a tile: no next-stage fragments (the epilogue comes first) - the residual forms request the 16 residual
        // loads of slice 0 there instead (row block by row block, the order the epilogue consumes them), into the registers K-half
        // 0's fragments have left: one load per gap that holds no DMA, from the barrier to the end of the K-step
        if constexpr (LAST && RES) {
          constexpr int ri = a9_resid_slot(gap);
          if constexpr (ri >= 0) epi_rload(g, 0, ri >> 1, ri & 1, rres);
        }
        if constexpr (op >= OP_MA && op < OP_MA + 8) set_m0(std::integral_constant<int, op - OP_MA>{}, std::false_type{}, bb);
        if constexpr (op >= OP_DA && op < OP_DA + 8) dma(std::integral_constant<int, op - OP_DA>{}, std::false_type{}, vo);
        if constexpr (op >= OP_MW && op < OP_MW + 8) set_m0(std::integral_constant<int, op - OP_MW>{}, std::true_type{}, bb);
        if constexpr (op >= OP_DW && op < OP_DW + 8) dma(std::integral_constant<int, op - OP_DW>{}, std::true_type{}, vo);
        if constexpr (op >= OP_MDA && op < OP_MDA + 8) {
          set_m0(std::integral_constant<int, op - OP_MDA>{}, std::false_type{}, bb);
          dma(std::integral_constant<int, op - OP_MDA>{}, std::false_type{}, vo);
        }
        if constexpr (op >= OP_MDW && op < OP_MDW + 8) {
          set_m0(std::integral_constant<int, op - OP_MDW>{}, std::true_type{}, bb);
          dma(std::integral_constant<int, op - OP_MDW>{}, std::true_type{}, vo);
        }
        if constexpr (op == OP_B1 || op == OP_B2) {   // every wave holds its A (B1) / W (B2) fragments of this stage -> region may be refilled
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
          __builtin_amdgcn_s_waitcnt(0xC07F);
        }
        if constexpr (op == OP_B3)   // this wave's pieces of stage it+1 have landed (SC.vm younger DMAs may fly) -> everybody's
          asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"(SC.vm) : "memory");
      });
      A4_FENCE();
    });
    adv_d();
    ++it;
  };

  for (int t = 0; t < my_tiles; ++t) {
    const int m0 = ((c_s / n_per) * npset + pset) * G3_BM, n0 = (grp * n_per + c_s % n_per) * G3_BN;
    kstep(std::true_type{}, std::false_type{});
    for (int kt = 1; kt < nk - 1; ++kt) kstep(std::false_type{}, std::false_type{});
    EpiPre p0;                                                 // slice 0's epilogue inputs arrive behind the last K-step
    EpiRows prow;                                              // and so do the row statistics of the LayerNorm-folded forms
    epi_prefetch<EPI>(g, n0, wn * 2, q4, p0);
    epi_prefetch_rows<EPI>(g, m0, wm, l15, prow);
    if (RES) epi_res_setup(g, m0, n0, wm, wn, l15, q4, rres);
    kstep(std::false_type{}, std::true_type{});
    if (m0 + G3_BM <= g.M && n0 + G3_BN <= g.N) agpr_epilogue<EPI, true>(g, m0, n0, wm, wn, l15, q4, p0, prow, rres);
    else agpr_epilogue<EPI, false>(g, m0, n0, wm, wn, l15, q4, p0, prow, rres);
    c_s += nwl;
    const char* nbuf = lds + (it & 1) * G3_STAGE;              // the next tile's first fragments, behind the epilogue
#pragma unroll
    for (int i = 0; i < 8; ++i) { fa0[i] = ldfrag(nbuf + fbaseA + i * 2048 + co0); fw0[i] = ldfrag(nbuf + fbaseW + i * 2048 + co0); }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the two stages requested past the last tile (zero-length resources)
  }

  if (tail) {   // ---- remainder phase: the left-over tiles of all XCD sets as 64x64 sub-tiles over all workgroups ----
    const A9Remainder rm = a9_remainder(tilesM, g.tilesN, ngrp, nwl);
    for (int j = blockIdx.x; j < rm.total * 16; j += gridDim.x) {
      const int tt = j >> 4, sub = j & 15;
      int x = 0;
#pragma unroll
      for (int i = 1; i < 8; ++i) x += tt >= rm.pre[i] ? 1 : 0;
      int rx = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) rx = i == x ? rm.rounds[i] : rx;
      int px = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) px = i == x ? rm.pre[i] : px;
      const int s = rx * nwl + (tt - px);
      const int m0 = ((s / n_per) * npset + x / ngrp) * G3_BM, n0 = ((x % ngrp) * n_per + s % n_per) * G3_BN;
      const int m_s = m0 + (sub >> 2) * 64, n_s = n0 + (sub & 3) * 64;
      if (m_s < g.M && n_s < g.N) a9_tail_subtile<EPI>(g, lds, m_s, n_s, wid, lane);   // (uniform per workgroup)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the zero-length stages requested past the last K-step
  }
}

template <int EPI, class SCHED>
int launch_gemm_a9(GemmArgs g, hipStream_t st) {
  const int tilesM = (g.M + G3_BM - 1) / G3_BM;
  g.tilesN = (g.N + G3_BN - 1) / G3_BN;
  g.nwg = tilesM * g.tilesN;
  int ngrp = ((double)g.N * g.K * 2.0 > 4.0e6 && g.tilesN >= 8 && g.tilesN % 2 == 0) ? 2 : 1;
  if (g.ngrp > 0 && g.tilesN % g.ngrp == 0) ngrp = g.ngrp;
  // remainder phase when a partial last round exists, every XCD set has whole rounds to run, and the left-over tiles make at
  // most A9_TAIL_MAX_SUBTILES 64x64 sub-tiles per workgroup (beyond that the partial round of whole tiles is cheaper: see above)
  const A9Remainder rm = a9_remainder(tilesM, g.tilesN, ngrp, 32);
  int min_rounds = rm.rounds[0];
  for (int x = 1; x < 8; ++x) min_rounds = rm.rounds[x] < min_rounds ? rm.rounds[x] : min_rounds;
  const int tail = (g.variant != 83 && rm.total > 0 && min_rounds >= 1 && rm.total * 16 <= A9_TAIL_MAX_SUBTILES * 256) ? 1 : 0;
  hipLaunchKernelGGL((gemm_bf16_a9_kernel<EPI, SCHED>), dim3(256), dim3(256), 0, st, g, tilesM, ngrp, tail);
  return tspo::check_launch("gemm_bf16_a9");
}


}  // namespace
