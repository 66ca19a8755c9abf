// --dev builds only (python -m tspo_amd.build --dev compiles csrc/dev/*.hip into the library; the shipped build does not): the
// laboratory of the LDS-DMA GEMM - the schedules the production one was measured against, the no-DMA ablation (timing only, wrong
// results), N-group overrides and the s_memtime probe of the K-step (tools/probe_gemm_dma.py).  Reached from gemm_dma.hip through
// the weak symbol tspo_lab_gemm_dma.  Results: profiles/r4_a_gemm_dma_schedules_and_probes.txt, DESIGN 4.4.
#include "gemm_dma_lab_kernel.h"

static void* g_dma_debug = nullptr;
extern "C" void tspo_dma_set_debug(void* p) { g_dma_debug = p; }

namespace {
// the vendor kernel's positions (three barriers; m0 and DMA in one gap)
struct LabVendorPositions {
  static constexpr A9Sched make() {
    A9Builder b;
    for (int j = 0; j < 8; ++j) b.put(2 * j, OP_RA1 + j);
    b.put(21, OP_B1);
    for (int j = 0; j < 5; ++j) { b.put(22 + 2 * j, OP_MDA + j); b.put(23 + 2 * j, OP_RW1 + j); }
    b.put(33, OP_RW1 + 5); b.put(35, OP_RW1 + 6); b.put(37, OP_RW1 + 7);
    b.put(47, OP_B2);
    b.put(48, OP_MDA + 5); b.put(51, OP_MDA + 6); b.put(54, OP_MDA + 7); b.put(57, OP_MDW + 0); b.put(60, OP_MDW + 1);
    b.put(64 + 2, OP_MDW + 2); b.put(64 + 6, OP_MDW + 3); b.put(64 + 10, OP_MDW + 4);
    b.put(64 + 20, OP_B3);
    b.put(64 + 24, OP_MDW + 5); b.put(64 + 28, OP_MDW + 6); b.put(64 + 55, OP_MDW + 7);
    const int ra[8] = {21, 22, 23, 25, 26, 29, 31, 32}, rw[8] = {33, 34, 37, 40, 42, 45, 48, 51};
    for (int j = 0; j < 8; ++j) { b.put(64 + ra[j], OP_RA0 + j); b.put(64 + rw[j], OP_RW0 + j); }
    return b.done();
  }
};
// three barriers, DMA spread evenly (one every 6 gaps from B1): the production schedule before the two-barrier one
struct LabThreeBarriers {
  static constexpr A9Sched make() {
    A9Builder b;
    for (int j = 0; j < 8; ++j) b.put(2 * j, OP_RA1 + j);
    b.put(21, OP_B1);
    for (int j = 0; j < 8; ++j) b.put(23 + 2 * j, OP_RW1 + j);
    b.put(22, OP_MDA + 0); b.put(28, OP_MDA + 1); b.put(34, OP_MDA + 2); b.put(40, OP_MDA + 3);
    b.put(47, OP_B2);
    b.put(48, OP_MDA + 4); b.put(54, OP_MDW + 0); b.put(60, OP_MDA + 5); b.put(66, OP_MDW + 1); b.put(72, OP_MDA + 6); b.put(78, OP_MDW + 2);
    b.put(84, OP_MDA + 7);
    b.put(85, OP_B3);
    for (int j = 0; j < 8; ++j) { b.put(86 + 2 * j, OP_RA0 + j); b.put(102 + 2 * j, OP_RW0 + j); }
    b.put(91, OP_MDW + 3); b.put(97, OP_MDW + 4); b.put(103, OP_MDW + 5); b.put(109, OP_MDW + 6); b.put(115, OP_MDW + 7);
    return b.done();
  }
};

template <int EPI>
int lab_variant(const GemmArgs& g, hipStream_t st) {
  if (g.variant == 76) return launch_gemm_a9<EPI, LabThreeBarriers>(g, st);
  if (g.variant == 67) return launch_gemm_a9<EPI, LabVendorPositions>(g, st);
  if (g.variant == 75) return launch_gemm_a9lab<EPI, A9ScheduleProduction, false, true>(g, st);   // production stream without its DMA instructions
  if (g.variant == 73 || g.variant == 72 || g.variant == 71) {     // production schedule, N groups per XCD set forced to 1 / 4 / 8 (auto: 2 for wide N)
    GemmArgs h = g;
    h.ngrp = g.variant == 73 ? 1 : (g.variant == 72 ? 4 : 8);
    return launch_gemm_a9<EPI, A9ScheduleProduction>(h, st);
  }
  if (g.variant == 74) {                                         // production schedule with the s_memtime probe
    GemmArgs h = g;
    h.pos = reinterpret_cast<const float*>(g_dma_debug);
    if (!h.pos) return tspo::set_err(TSPO_EINVAL, "gemm_dma: probe variant without a debug buffer (tspo_dma_set_debug)");
    return launch_gemm_a9lab<EPI, A9ScheduleProduction, true>(h, st);
  }
  return tspo::set_err(TSPO_EINVAL, "gemm: kernel variant %d is not part of this build", g.variant);
}
}  // namespace

extern "C" int tspo_lab_gemm_dma(int epi, const GemmArgs* g, hipStream_t st) {
  switch (epi) {   // the micro-benchmark's epilogues
    case GE_BIAS: return lab_variant<GE_BIAS>(*g, st);
    case GE_GELU: return lab_variant<GE_GELU>(*g, st);
    case GE_RESID: return lab_variant<GE_RESID>(*g, st);
  }
  return tspo::set_err(TSPO_EINVAL, "gemm: lab variant %d is built for the bias / gelu / residual epilogues only", g->variant);
}
