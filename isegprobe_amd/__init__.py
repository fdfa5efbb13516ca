"""isegprobe_amd: MI355X-native implementation of iSegProbe's per-click dense-feature path.

Hot kernels are hand-written HIP for gfx950 behind a C-ABI shared library
(include/isegprobe_hip.h, isegprobe_amd/csrc); ``isegprobe_amd.core`` mirrors the
reference's plugin API (featurizers / upsamplers / heads / ModelBuilder / iSegProbeModel /
BasePredictor) on top of it.
"""
__version__ = "0.1.0"


def install_as_core():
    """Alias ``isegprobe_amd.core`` as top-level ``core`` (and its sub-modules on import) so
    reference checkpoints / scripts that name ``core.model...`` resolve to this package."""
    import importlib
    import sys

    pkg = importlib.import_module("isegprobe_amd.core")
    sys.modules.setdefault("core", pkg)
    for name in ("model", "model.ops", "model.iseg_base_model", "model.iseg_probe_model", "model.featurizers",
                 "model.upsamplers", "model.heads", "utils", "utils.model_builder", "utils.serialization",
                 "utils.log"):
        mod = importlib.import_module("isegprobe_amd.core." + name)
        sys.modules.setdefault("core." + name, mod)
    return pkg
