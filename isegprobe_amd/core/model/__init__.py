from .iseg_base_model import iSegBaseModel
from .iseg_probe_model import iSegProbeModel

__all__ = ["iSegBaseModel", "iSegProbeModel"]
