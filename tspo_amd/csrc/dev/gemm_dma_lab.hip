// --dev builds only (python -m tspo_amd.build --dev compiles csrc/dev/*.hip into the library; the shipped build does not): the
// laboratory of the LDS-DMA GEMM - schedule A/B variants, the no-DMA ablation (timing only, wrong results) and the s_memtime probe
// of the K-step (tools/probe_gemm_dma.py).  Reached from gemm_dma.hip through the weak symbol tspo_lab_gemm_dma.
#include "../gemm_dma_kernel.h"

static void* g_dma_debug = nullptr;
extern "C" void tspo_dma_set_debug(void* p) { g_dma_debug = p; }

namespace {
template <int EPI>
int lab_variant(const GemmArgs& g, hipStream_t st) {
  if (g.variant == 76) return launch_gemm_a9<EPI, 4>(g, st);     // three barriers, DMA spread evenly (production until the two-barrier schedule)
  if (g.variant == 67) return launch_gemm_a9<EPI, 0>(g, st);     // the vendor kernel's positions
  if (g.variant == 75) return launch_gemm_a9<EPI, 109>(g, st);   // production schedule without its DMA instructions (timing only)
  if (g.variant == 68) return launch_gemm_a9<EPI, 10>(g, st);    // two barriers, B3 ten gaps later
  if (g.variant == 69) return launch_gemm_a9<EPI, 8>(g, st);     // two barriers per K-step
  if (g.variant == 73 || g.variant == 72 || g.variant == 71) {     // production schedule, N groups per XCD set forced to 1 / 4 / 8 (auto: 2 for wide N)
    GemmArgs h = g;
    h.ngrp = g.variant == 73 ? 1 : (g.variant == 72 ? 4 : 8);
    return launch_gemm_a9<EPI, 9>(h, st);
  }
  if (g.variant == 74) {                                         // production schedule with the s_memtime probe
    GemmArgs h = g;
    h.pos = reinterpret_cast<const float*>(g_dma_debug);
    if (!h.pos) return tspo::set_err(TSPO_EINVAL, "gemm_dma: probe variant without a debug buffer (tspo_dma_set_debug)");
    return launch_gemm_a9<EPI, 9, true>(h, st);
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
