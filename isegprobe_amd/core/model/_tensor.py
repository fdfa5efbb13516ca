"""Layout glue between the plugin API (NCHW-shaped tensors, as in the reference) and the
kernels (NHWC bf16).  Modules hand each other *NCHW-shaped views of NHWC bf16 storage* --
the reference's own featurizer also returns a permuted view (DINOv2.py:545) -- so chaining
our modules costs no copies, while an fp32 NCHW tensor from foreign code is converted once
by a HIP kernel."""
import torch

from ... import hip_ops as ops
from ..._lib import IspError

BF16 = torch.bfloat16


def to_nhwc_bf16(x, keep_f16=False):
    """x: [B,C,H,W]-shaped GPU tensor (bf16 channels-last view, or fp32 in any strides)
    -> contiguous [B,H,W,C] bf16 (an IEEE-half channels-last view stays half with ``keep_f16``)."""
    if x.dim() != 4:
        raise IspError(f"expected a 4-D [B,C,H,W] tensor, got {tuple(x.shape)}")
    if not x.is_cuda:
        raise IspError("the HIP path needs GPU tensors (no CPU fallback)")
    if x.dtype == BF16:
        y = x.permute(0, 2, 3, 1)
        return y if y.is_contiguous() else y.contiguous()
    if x.dtype == torch.float16 and x.permute(0, 2, 3, 1).is_contiguous():
        if keep_f16:  # IEEE-half NHWC maps (FeatUp-JBU / LoftUp inference output) go to consumers that take them
            return x.permute(0, 2, 3, 1)
        return x.permute(0, 2, 3, 1).to(BF16)
    if x.dtype != torch.float32:
        x = x.float()
    return ops.nchw_f32_to_nhwc_bf16(x)


def nchw_view(x_nhwc):
    return x_nhwc.permute(0, 3, 1, 2)


def to_nchw_f32(x):
    """Materialise a plugin tensor as contiguous fp32 NCHW (for host callers / dumps)."""
    if x.dtype == BF16 and x.permute(0, 2, 3, 1).is_contiguous():
        return ops.nhwc_bf16_to_nchw_f32(x.permute(0, 2, 3, 1))
    return x.float().contiguous()


_pack_serial = [0]


def pack_serial(P):
    """Identity of a packed-weight value for cache keys: the serial number its build was given (monotonically increasing,
    never reused -- ``id(P)`` of a rebuilt dict can be)."""
    return P["_serial"] if isinstance(P, dict) and "_serial" in P else id(P)


class PackedCache:
    """Kernel-layout (bf16, permuted, padded) copies of a module's fp32 parameters,
    rebuilt whenever a parameter is replaced or modified in place."""

    def __init__(self):
        self._key = None
        self._val = None

    def tensors_of(self, enumerate_tensors):
        """The tensor list to watch, enumerated once: Parameter / buffer objects of a module tree are stable
        (load_state_dict, .to(), optimizers update them in place), walking the tree per forward is not free."""
        if getattr(self, "_watch", None) is None:
            self._watch = list(enumerate_tensors())
        return self._watch

    def get(self, tensors, build):
        key = tuple((t.data_ptr(), t._version, t.device) for t in tensors)
        if key != self._key:
            with torch.no_grad():
                self._val = build()
            if isinstance(self._val, dict):
                _pack_serial[0] += 1
                self._val["_serial"] = _pack_serial[0]
            self._key = key
        return self._val
