"""tspo_amd - MI355X-native temporal sampling agent and TSPO policy step.

Hot path (hand-written HIP for gfx950, C ABI in include/tspo_hip.h):
CLIP-L frame encode -> temporal scoring head -> top-k / bin-max / Gumbel-top-k frame sampler ->
group-relative advantage + policy-gradient reduction -> selector backward -> AdamW.

    from tspo_amd.temporal_agent import TSPOModel, MultiModal_Align, positional_encoding   # model/temporal_agent.py
    from tspo_amd.utils import gumbel_softmax, AKS_sampling                                # model/utils.py
    from tspo_amd.policy import TemporalPolicy                                             # llava_qwen.py:131-176
    from tspo_amd.pipeline import FrameScorer, PolicyTrainer                               # fused pipelines
"""
__version__ = "0.1.0"


def install_reference_aliases() -> None:
    """Make `from model.temporal_agent import ...` / `from model.utils import ...` (the reference's import
    paths) resolve to this package, so existing callers run unchanged."""
    import sys
    import types
    from . import temporal_agent, utils
    pkg = types.ModuleType("model")
    pkg.__path__ = []  # mark as package
    pkg.temporal_agent, pkg.utils = temporal_agent, utils
    sys.modules["model"] = pkg
    sys.modules["model.temporal_agent"] = temporal_agent
    sys.modules["model.utils"] = utils
