"""Straight-through mask layers on the wavelet coefficients: counterpart of model/Straight_Through_Dropout.py
(STEFunction :10-17, Straight_Through_Dropout :20-43, MaskedWavelet_Straight_Through_Dropout :47-78).

Both layers hand the fused HIP decode a per-coefficient factor (``drop_factor()``):
* Straight_Through_Dropout: the binary mask ``rand < mask_values`` (a comparison result: like in the reference no
  gradient reaches ``mask_values`` through it, only the L1 penalty trains them);
* MaskedWavelet_Straight_Through_Dropout: ``sigmoid(mask_values)`` plus the threshold -- the kernel forms the reference's
  ``(x*hard - x*soft).detach() + x*soft`` value op for op and returns the soft-mask gradients.
"""
from __future__ import annotations

import torch
from torch.nn import functional as F

from .. import _lib, ops
from .Dropout_Layer import DropFactor, DropoutLayer


class STEFunction(torch.autograd.Function):
    """Binary threshold with a hard-tanh pass-through gradient (reference :10-17; kept for callers that import it)."""

    @staticmethod
    def forward(ctx, input, thresh):
        return input < thresh

    @staticmethod
    def backward(ctx, grad_output):
        return F.hardtanh(grad_output), None


class Straight_Through_Dropout(DropoutLayer):

    def __init__(self, size=(1, 1, 1), probability=0.5, threshold=0.5):
        super().__init__(size, probability, threshold)
        self.mask_values = torch.nn.Parameter(torch.ones(size), requires_grad=True)

    def _draw(self):
        return torch.rand(self.c, device=self.mask_values.device)

    def drop_factor(self):
        if not self.training:
            return None
        with torch.no_grad():
            return DropFactor((self._draw() < self.mask_values).to(torch.float32))

    def l1_loss(self):
        return ops.penalty_sums([_lib.PENALTY_L1], [self.mask_values])[0]

    def calculate_pruning_mask(self, device):
        return self.mask_values > self.threshold

    def multiply_values_with_dropout(self, input, device):
        with torch.no_grad():
            return input * self.calculate_pruning_mask(device)
    # no size_layer(): the reference has none either, so save_dropvalues_on_grid raises TypeError


class MaskedWavelet_Straight_Through_Dropout(DropoutLayer):

    def __init__(self, size=(1, 1, 1), probability=0.5, threshold=0.5):
        super().__init__(size, probability, threshold)
        self.mask_values = torch.nn.Parameter(torch.ones(size), requires_grad=True)
        self.d_mask = None

    def drop_factor(self):
        if not self.training:
            return None
        if self.d_mask is not None:
            return DropFactor(self.d_mask.to(self.mask_values.device, torch.float32))
        return DropFactor(torch.sigmoid(self.mask_values), float(self.threshold))

    def l1_loss(self):
        return ops.penalty_sums([_lib.PENALTY_L1], [self.mask_values])[0]

    def calculate_pruning_mask(self, device):
        mask = torch.sigmoid(self.mask_values)
        self.d_mask = (mask >= self.threshold).to(device)
        return mask                                        # the SOFT mask, as in the reference (:67-70)

    def multiply_values_with_dropout(self, input, device):
        with torch.no_grad():
            mask = self.calculate_pruning_mask(device)
            return (input * (mask >= self.threshold) - input * mask) + (input * mask)

    def size_layer(self):
        return self.mask_values.numel()
