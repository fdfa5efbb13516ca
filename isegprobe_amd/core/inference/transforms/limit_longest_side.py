"""Limit the longest image side (reference core/inference/transforms/limit_longest_side.py:12-32): a ZoomIn whose window is
always the whole image -- taken whenever the image's longest side exceeds ``max_size``, from the first call on, and never
re-chosen from a prediction."""
from .zoom_in import ZoomIn


class LimitLongestSide(ZoomIn):
    def __init__(self, max_size: int = 800):
        super().__init__(target_size=max_size, skip_clicks=-1)  # (no click count gates it)
        self.skip_clicks = 0                                    # the attribute value the reference's instance carries

    def _transform(self, image_nd, clicks_lists):
        gate, self.skip_clicks = self.skip_clicks, -1
        try:
            return super()._transform(image_nd, clicks_lists)
        finally:
            self.skip_clicks = gate

    def _propose(self, image_nd, clicks):
        height, width = image_nd.shape[2:4]
        return (0, height - 1, 0, width - 1) if max(height, width) > self.target_size else None

    def _must_replace(self, proposed, clicks) -> bool:
        return True

    def check_possible_recalculation(self) -> bool:
        return False
