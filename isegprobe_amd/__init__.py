"""isegprobe_amd: MI355X-native implementation of iSegProbe's per-click dense-feature path.

Hot kernels are hand-written HIP for gfx950 behind a C-ABI shared library
(include/isegprobe_hip.h, isegprobe_amd/csrc); ``isegprobe_amd.core`` mirrors the
reference's plugin API (featurizers / upsamplers / heads / ModelBuilder / iSegProbeModel /
BasePredictor) on top of it.
"""
__version__ = "0.1.0"
