"""Normalized focal loss on sigmoid outputs (behaviour of reference core/training/losses.py:11-109, pinned by
tests/golden/train_step.npz: loss values and the gradients they induce).

Per sample, with p = sigmoid(logit), v = [label != ignore_label] and q = 1 - |label - p| (the probability
assigned to the true class; 1 on ignored pixels):

    focal weight   f = (1 - q)^gamma
    normaliser     m = sum(v) / (sum(f) + eps)         over the sample's H x W plane, a constant for autograd
    pixel loss     l = -a * min(f * m, max_mult) * log(min(q + eps, 1)) * v * weight,  a = alpha on label > 0.5, else 1 - alpha
    sample loss    sum(l) / (sum(v) + eps)             (or sum(l) when size_average is off)

SURVEY.md section 2 keeps the loss as PyTorch ops (a few elementwise passes over [B,1,H,W], off the dense-feature
path).  The reference additionally tracks two running statistics for TensorBoard with two device->host copies per
call; they do not influence the loss and are not kept."""
import torch
import torch.nn as nn


def _true_class_prob(prob, label, valid):
    return torch.where(valid, 1.0 - (label - prob).abs(), torch.ones_like(prob))


class NormalizedFocalLossSigmoid(nn.Module):
    def __init__(self, axis=-1, alpha=0.25, gamma=2, max_mult=-1, eps=1e-12, from_sigmoid=False,
                 detach_delimeter=True, batch_axis=0, weight=None, size_average=True, ignore_label=-1):
        super().__init__()
        self.alpha, self.gamma, self.max_mult, self.eps = alpha, gamma, max_mult, eps
        self.inputs_are_probabilities = from_sigmoid
        self.constant_normaliser = detach_delimeter
        self.weight = 1.0 if weight is None else weight
        self.size_average, self.ignore_label = size_average, ignore_label

    def forward(self, pred, label):
        prob = pred if self.inputs_are_probabilities else torch.sigmoid(pred)
        valid = label != self.ignore_label
        q = _true_class_prob(prob, label, valid)
        focal = (1.0 - q).pow(self.gamma)
        plane = (-2, -1)
        n_valid_plane = valid.sum(dim=plane, keepdim=True)
        normaliser = n_valid_plane / (focal.sum(dim=plane, keepdim=True) + self.eps)
        if self.constant_normaliser:
            normaliser = normaliser.detach()
        focal = focal * normaliser
        if self.max_mult > 0:
            focal = focal.clamp(max=self.max_mult)
        class_weight = torch.where(label > 0.5, self.alpha, 1.0 - self.alpha) * valid
        pixel = -(class_weight * focal * torch.log((q + self.eps).clamp(max=1.0))) * valid * self.weight
        per_sample = pixel.flatten(1).sum(1)
        if self.size_average:
            per_sample = per_sample / (valid.flatten(1).sum(1) + self.eps)
        return per_sample
