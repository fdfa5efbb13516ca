"""Normalized focal loss (reference core/training/losses.py:11-109).  SURVEY.md section 2 keeps the
loss / optimizer as PyTorch ops (tiny elementwise work outside the dense-feature path); the two
host syncs of the reference's statistics tracking (:67-83) are dropped -- they feed logging only."""
import torch
import torch.nn as nn


class NormalizedFocalLossSigmoid(nn.Module):
    def __init__(self, axis=-1, alpha=0.25, gamma=2, max_mult=-1, eps=1e-12, from_sigmoid=False,
                 detach_delimeter=True, batch_axis=0, weight=None, size_average=True, ignore_label=-1):
        super().__init__()
        self._alpha, self._gamma, self._ignore_label = alpha, gamma, ignore_label
        self._weight = weight if weight is not None else 1.0
        self._from_logits, self._eps = from_sigmoid, eps
        self._size_average, self._detach_delimeter, self._max_mult = size_average, detach_delimeter, max_mult

    def forward(self, pred, label):
        one_hot = label > 0.5
        sample_weight = label != self._ignore_label
        if not self._from_logits:
            pred = torch.sigmoid(pred)
        alpha = torch.where(one_hot, self._alpha * sample_weight, (1 - self._alpha) * sample_weight)
        pt = torch.where(sample_weight, 1.0 - torch.abs(label - pred), torch.ones_like(pred))
        beta = (1 - pt) ** self._gamma
        sw_sum = torch.sum(sample_weight, dim=(-2, -1), keepdim=True)
        beta_sum = torch.sum(beta, dim=(-2, -1), keepdim=True)
        mult = sw_sum / (beta_sum + self._eps)
        if self._detach_delimeter:
            mult = mult.detach()
        beta = beta * mult
        if self._max_mult > 0:
            beta = torch.clamp_max(beta, self._max_mult)
        loss = -alpha * beta * torch.log(torch.clamp_max(pt + self._eps, 1.0))
        loss = self._weight * (loss * sample_weight)
        dims = tuple(range(1, loss.dim()))
        if self._size_average:
            return torch.sum(loss, dim=dims) / (torch.sum(sample_weight, dim=dims) + self._eps)
        return torch.sum(loss, dim=dims)
