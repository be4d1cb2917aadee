"""Straight-through mask layers on the wavelet coefficients: counterpart of model/Straight_Through_Dropout.py
(STEFunction :10-17, Straight_Through_Dropout :20-43, MaskedWavelet_Straight_Through_Dropout :47-78); same class names,
parameter (``mask_values``), thresholds and methods.

Neither layer multiplies coefficients itself: ``drop_factor()`` names the per-coefficient factor and the fused HIP decode
applies it (forward and backward inside the wavelet kernels):
* ``Straight_Through_Dropout``: the binary mask ``rand < mask_values``.  A comparison result carries no gradient, so --
  exactly like in the reference, whose STEFunction.backward never runs for this reason -- only the L1 penalty trains
  ``mask_values``;
* ``MaskedWavelet_Straight_Through_Dropout``: ``sigmoid(mask_values)`` together with the threshold; the kernel forms the
  reference's ``(x*hard - x*soft).detach() + x*soft`` op for op (value of the hard mask, gradients of the soft one).
"""
from __future__ import annotations

import torch
from torch.nn import functional as F

from .. import _lib, ops
from .Dropout_Layer import DropFactor, DropoutLayer


class STEFunction(torch.autograd.Function):
    """``input < thresh`` with a hard-tanh pass-through gradient (reference :10-17); kept for code that imports it."""

    @staticmethod
    def forward(ctx, input, thresh):
        return input < thresh

    @staticmethod
    def backward(ctx, grad_output):
        return F.hardtanh(grad_output), None


class _MaskValueLayer(DropoutLayer):
    """Shared part of the two layers: one learnable ``mask_values`` entry per coefficient position, initialised to 1,
    penalised by its L1 norm."""

    def __init__(self, size, probability, threshold):
        super().__init__(size, probability, threshold)
        self.mask_values = torch.nn.Parameter(torch.ones(size), requires_grad=True)

    def l1_loss(self):
        return ops.penalty_sums([_lib.PENALTY_L1], [self.mask_values])[0]


class Straight_Through_Dropout(_MaskValueLayer):

    def __init__(self, size=(1, 1, 1), probability=0.5, threshold=0.5):
        super().__init__(size, probability, threshold)

    def _draw(self):
        """U(0,1) of the layer's size (torch's generator; the parity tests substitute recorded draws)."""
        return torch.rand(self.c, device=self.mask_values.device)

    def drop_factor(self):
        if not self.training:
            return None
        with torch.no_grad():
            return DropFactor((self._draw() < self.mask_values).to(torch.float32))

    def calculate_pruning_mask(self, device):
        return self.mask_values > self.threshold

    def multiply_values_with_dropout(self, input, device):
        with torch.no_grad():
            return input * self.calculate_pruning_mask(device)
    # size_layer() is deliberately absent, as in the reference: save_dropvalues_on_grid then raises TypeError after having
    # replaced the coefficients by their masked version (pinned by tests/golden/drop_straight_through.npz)


class MaskedWavelet_Straight_Through_Dropout(_MaskValueLayer):

    def __init__(self, size=(1, 1, 1), probability=0.5, threshold=0.5):
        super().__init__(size, probability, threshold)
        self.d_mask = None

    def drop_factor(self):
        if not self.training:
            return None
        if self.d_mask is not None:                     # after pruning: the stored hard mask only
            return DropFactor(self.d_mask.to(self.mask_values.device, torch.float32))
        return DropFactor(torch.sigmoid(self.mask_values), float(self.threshold))

    def calculate_pruning_mask(self, device):
        soft = torch.sigmoid(self.mask_values)
        self.d_mask = (soft >= self.threshold).to(device)
        return soft                                       # the reference returns the SOFT mask here (:67-70)

    def multiply_values_with_dropout(self, input, device):
        with torch.no_grad():
            soft = self.calculate_pruning_mask(device)
            return (input * (soft >= self.threshold) - input * soft) + (input * soft)

    def size_layer(self):
        return self.mask_values.numel()
