"""Placeholder so the registry imports; replaced below."""
from . import BaseUpsampler


class LiFTUpsampler(BaseUpsampler):
    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError("LiFTUpsampler: HIP path not built yet")

    def forward(self, source, guidance):
        raise NotImplementedError
