// bf16 MFMA GEMM for gfx950 (MI355X), LDS-DMA operand path ("a9", production since round 4): the kernel is the template of
// gemm_dma_kernel.h (whole 256x256 tiles in whole rounds + the 64x64 remainder phase of round 5); literal-AGPR MFMA statements and
// the epilogue out of a[0:255]: gemm_agpr_common.h.
#include "gemm_dma_kernel.h"

// Defined only by the extra translation unit of a `python -m tspo_amd.build --dev` build (csrc/dev/gemm_dma_lab.hip): schedule
// A/B variants 72-76.  The shipped library leaves it unresolved (null) and rejects those variant numbers.
extern "C" __attribute__((weak)) int tspo_lab_gemm_dma(int epi, const GemmArgs* g, hipStream_t st);

namespace {
template <int EPI>
int launch_a9_variant(const GemmArgs& g, hipStream_t st) {
  if (g.K < 2 * GT_BK) return tspo::set_err(TSPO_EINVAL, "gemm_dma: K=%d too small for the DMA kernel", g.K);
  if (g.variant == 77 || g.variant == 83) return launch_gemm_a9<EPI, A9ScheduleProduction>(g, st);   // 83: remainder phase off
  if (tspo_lab_gemm_dma) return tspo_lab_gemm_dma(EPI, &g, st);
  return tspo::set_err(TSPO_EINVAL, "gemm: kernel variant %d (epilogue %d) is not part of this build", g.variant, EPI);
}
}  // namespace

int tspo::gemm_bf16_dma(int epi, const GemmArgs& g, hipStream_t st) {
  switch (epi) {
    case GE_BIAS: return launch_a9_variant<GE_BIAS>(g, st);
    case GE_GELU: return launch_a9_variant<GE_GELU>(g, st);
    case GE_RESID: return launch_a9_variant<GE_RESID>(g, st);
    case GE_F32: return launch_a9_variant<GE_F32>(g, st);
    case GE_PATCH: return launch_a9_variant<GE_PATCH>(g, st);
    case GE_BIAS_LN: return launch_a9_variant<GE_BIAS_LN>(g, st);
    case GE_GELU_LN: return launch_a9_variant<GE_GELU_LN>(g, st);
    case GE_BIAS_LN_HM: return launch_a9_variant<GE_BIAS_LN_HM>(g, st);
    case GE_RESID_ST: return launch_a9_variant<GE_RESID_ST>(g, st);
  }
  return tspo::set_err(TSPO_EINVAL, "gemm_dma: bad epilogue %d", epi);
}
