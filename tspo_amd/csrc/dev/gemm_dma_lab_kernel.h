// --dev builds only: the round-4 form of the LDS-DMA GEMM kernel WITH its laboratory instrumentation - PROBE (s_memtime cycles per
// K-step and inside each synchronisation point, summed per wave; tools/probe_gemm_dma.py) and NODMA (the same instruction stream
// without its DMA instructions: timing only, wrong results).  A derived copy of csrc/gemm_dma_kernel.h as of round 4 (whole tiles
// only, no remainder phase); the product kernel carries none of this.  Enums / schedule tables come from the product header.
#pragma once
#include "../gemm_dma_kernel.h"

namespace {
// NODMA (--dev builds, timing only, wrong results): the same stream without its DMA instructions.
template <int EPI, class SCHED, bool PROBE = false, bool NODMA = false>
__global__ __launch_bounds__(256, 1) void gemm_bf16_a9lab_kernel(GemmArgs g, int tilesM, int ngrp) {
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
  const int my_tiles = wl < ntile_x ? (ntile_x - wl + nwl - 1) / nwl : 0;
  if (my_tiles == 0) return;
  A4_FENCE();   // claims a[0:255] for this kernel

  // ---- LDS-DMA: piece P = wid*8 + q of a region = rows 8P..8P+7 (1 KB); lane (rin, slot) brings global chunk slot ^ rin.
  //      Address split: voffset = lane part + K-step (ONE v_add per K-step), soffset = piece (loop-invariant SGPRs, the same
  //      for A and W), m0 = LDS destination (one s_add with a literal per piece) ----
  const int rin = lane >> 3, slot = lane & 7;
  unsigned lane_goff = ((unsigned)rin * (unsigned)g.K + (unsigned)((slot ^ rin) << 3)) * 2u;
  unsigned piece_stride = 8u * (unsigned)g.K * 2u;
  unsigned lds0 = (unsigned)(size_t)lds + (unsigned)wid * 8192u;
  unsigned soff0 = (unsigned)wid * 8u * piece_stride;
  int d_kt = 0, d_s = wl;   // the stage the NEXT K-step's DMA brings: K-step inside the tile, tile
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
    if (!NODMA) asm volatile("s_add_u32 m0, %0, %1" ::"s"(b), "i"(off) : "scc");
  };
  auto dma = [&](auto q_, auto isw_, unsigned voff) {
    constexpr int q = decltype(q_)::value;
    constexpr bool isw = decltype(isw_)::value;
    const unsigned vo = voff, so = soff0 + q * piece_stride;
    const __amdgpu_buffer_rsrc_t rs = isw ? w_rs : a_rs;
    // default cache policy on purpose: `nt` on the A or the W stream cuts the QKV form's L2-side fetch by a third (3.7 -> 2.5 GB per
    // launch) and is 2-6 % SLOWER on every shape; sc1 / sc0 sc1 change nothing (profiles/r4_g_fetch_calibration_and_cache_policy.txt)
    if (!NODMA) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(vo), "s"(rs), "s"(so) : "memory");
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
  // PROBE (--dev builds, csrc/dev/gemm_dma_lab.hip): s_memtime cycles per K-step and inside each synchronisation point, summed per wave
  unsigned long long pr_ks = 0, pr_n = 0, pr_b1 = 0, pr_b2 = 0, pr_vm = 0, pr_b3 = 0, pr_vmz = 0, pr_nz = 0, pr_vm1 = 0, pr_epi = 0, pr_tile = 0;
  int pr_kt = 0;
  auto stamp = [&]() {
    unsigned long long t = 0;
    if (PROBE) { t = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    return t;
  };
  auto kstep = [&](auto zero_, auto last_) {
    constexpr bool ZERO = decltype(zero_)::value, LAST = decltype(last_)::value;
    const int cb = it & 1;
    const unsigned long long pr_t0 = stamp();
    const char* cur = lds + cb * G3_STAGE;
    const char* nxt = lds + (cb ^ 1) * G3_STAGE;
    const unsigned vo = lane_goff + (unsigned)d_kt * (GT_BK * 2u), bb = lds0 + (unsigned)cb * G3_STAGE;
    sfor<0, 16>([&](auto grp8_) {   // 16 groups of 8 MFMAs: column tile nn = grp8 & 7 of K-half grp8 >> 3
      constexpr int kh = decltype(grp8_)::value >> 3, nn = decltype(grp8_)::value & 7;
      sfor<0, 8>([&](auto mi_) {
        constexpr int mi = decltype(mi_)::value;
        constexpr int gap = kh * 64 + nn * 8 + mi;
        constexpr int op = SC.op[gap];
        {
          const i32x4 wf = kh ? fw1[nn] : fw0[nn], af = kh ? fa1[mi] : fa0[mi];
          if (ZERO && kh == 0) A4_MFMA_Z(nn, mi, wf, af); else A4_MFMA(nn, mi, wf, af);
        }
        if constexpr (op >= OP_RA1 && op < OP_RA1 + 8) fa1[op - OP_RA1] = ldfrag(cur + fbaseA + (op - OP_RA1) * 2048 + co1);
        if constexpr (op >= OP_RW1 && op < OP_RW1 + 8) fw1[op - OP_RW1] = ldfrag(cur + fbaseW + (op - OP_RW1) * 2048 + co1);
        if constexpr (!LAST && op >= OP_RA0 && op < OP_RA0 + 8) fa0[op - OP_RA0] = ldfrag(nxt + fbaseA + (op - OP_RA0) * 2048 + co0);
        if constexpr (!LAST && op >= OP_RW0 && op < OP_RW0 + 8) fw0[op - OP_RW0] = ldfrag(nxt + fbaseW + (op - OP_RW0) * 2048 + co0);
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
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          const unsigned long long ta = stamp();
          asm volatile("s_barrier" ::: "memory");
          __builtin_amdgcn_s_waitcnt(0xC07F);
          if (PROBE) { if (op == OP_B1) pr_b1 += stamp() - ta; else pr_b2 += stamp() - ta; }
        }
        if constexpr (op == OP_B3) {   // this wave's pieces of stage it+1 have landed (SC.vm younger DMAs may fly) -> everybody's
          const unsigned long long ta = stamp();
          asm volatile("s_waitcnt vmcnt(%0)" ::"i"(SC.vm) : "memory");
          const unsigned long long tb = stamp();
          asm volatile("s_barrier" ::: "memory");
          if (PROBE) {
            const unsigned long long tc = stamp(); pr_vm += tb - ta; pr_b3 += tc - tb;
            if (ZERO) { pr_vmz += tb - ta; ++pr_nz; }
            if (pr_kt == 1) pr_vm1 += tb - ta;
          }
        }
      });
      A4_FENCE();
    });
    adv_d();
    ++it;
    if (PROBE) { pr_ks += stamp() - pr_t0; ++pr_n; pr_kt = LAST ? 0 : pr_kt + 1; }
  };

  for (int t = 0; t < my_tiles; ++t) {
    const int m0 = ((c_s / n_per) * npset + pset) * G3_BM, n0 = (grp * n_per + c_s % n_per) * G3_BN;
    const unsigned long long pr_tt0 = stamp();
    kstep(std::true_type{}, std::false_type{});
    for (int kt = 1; kt < nk - 1; ++kt) kstep(std::false_type{}, std::false_type{});
    EpiPre p0;                                                 // slice 0's epilogue inputs arrive behind the last K-step
    epi_prefetch<EPI>(g, n0, wn * 2, q4, p0);
    kstep(std::false_type{}, std::true_type{});
    const unsigned long long pr_te0 = stamp();
    if (m0 + G3_BM <= g.M && n0 + G3_BN <= g.N) agpr_epilogue<EPI, true>(g, m0, n0, wm, wn, l15, q4, p0);
    else agpr_epilogue<EPI, false>(g, m0, n0, wm, wn, l15, q4, p0);
    if (PROBE) pr_epi += stamp() - pr_te0;
    c_s += nwl;
    const char* nbuf = lds + (it & 1) * G3_STAGE;              // the next tile's first fragments, behind the epilogue
#pragma unroll
    for (int i = 0; i < 8; ++i) { fa0[i] = ldfrag(nbuf + fbaseA + i * 2048 + co0); fw0[i] = ldfrag(nbuf + fbaseW + i * 2048 + co0); }
    if (PROBE) pr_tile += stamp() - pr_tt0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the two stages requested past the last tile (zero-length resources)
  if (PROBE && lane == 0 && g.pos) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(const_cast<float*>(g.pos)) + ((size_t)blockIdx.x * 4 + wid) * 12;
    o[9] = pr_epi; o[10] = pr_tile; o[11] = (unsigned long long)my_tiles;
    o[0] = pr_ks; o[1] = pr_n; o[2] = pr_b1; o[3] = pr_b2; o[4] = pr_vm; o[5] = pr_b3; o[6] = pr_vmz; o[7] = pr_nz; o[8] = pr_vm1;
  }
}

template <int EPI, class SCHED, bool PROBE = false, bool NODMA = false>
int launch_gemm_a9lab(GemmArgs g, hipStream_t st) {
  const int tilesM = (g.M + G3_BM - 1) / G3_BM;
  g.tilesN = (g.N + G3_BN - 1) / G3_BN;
  g.nwg = tilesM * g.tilesN;
  int ngrp = ((double)g.N * g.K * 2.0 > 4.0e6 && g.tilesN >= 8 && g.tilesN % 2 == 0) ? 2 : 1;
  if (g.ngrp > 0 && g.tilesN % g.ngrp == 0) ngrp = g.ngrp;
  hipLaunchKernelGGL((gemm_bf16_a9lab_kernel<EPI, SCHED, PROBE, NODMA>), dim3(256), dim3(256), 0, st, g, tilesM, ngrp);
  return tspo::check_launch("gemm_bf16_a9lab");
}


}  // namespace
