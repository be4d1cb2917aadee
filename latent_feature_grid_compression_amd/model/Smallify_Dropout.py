"""Smallify pruning of the wavelet coefficients: counterpart of model/Smallify_Dropout.py (SmallifyLoss :11-40,
SmallifyDropout :43-78, SmallifySignVarianceTracker :81-118).  Same class names, constructor arguments, parameter
name (``betas``), methods and attributes, so the reference's training driver and checkpoints keep working.

MI355X-first differences:
* the beta multiply (and its gradient sum over channels) runs inside the HIP inverse-wavelet kernels through
  ``drop_factor()``; nothing of coefficient size is materialised for it;
* the sign tracker's EMA state lives on the GPU and is advanced by one small kernel in the reference's fp32 operation
  order (the reference copies beta to the CPU on every training forward: a device sync per layer per step);
* SmallifyLoss evaluates every L1 / L2 term of the model in ONE reduction launch (+ one launch for all gradients).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import _lib, ops
from .Dropout_Layer import DropFactor, DropoutLayer
from .Straight_Through_Dropout import MaskedWavelet_Straight_Through_Dropout, Straight_Through_Dropout


def _l1_parameter(module):
    """The tensor whose L1 norm the reference penalises for this module, or None (Smallify_Dropout.py:23-28)."""
    if isinstance(module, SmallifyDropout):
        return module.betas
    if isinstance(module, (MaskedWavelet_Straight_Through_Dropout, Straight_Through_Dropout)):
        return module.mask_values
    return None


class SmallifyLoss(nn.Module):
    """weight_l1 * sum_layers |beta|_1 + weight_l2 * sum_tensors |coefficients|_2^2 (reference :32-40)."""

    def __init__(self, weight_l1: float = 1., weight_l2: float = 1.):
        super().__init__()
        self.weight_l1 = float(weight_l1)
        self.weight_l2 = float(weight_l2)

    def forward(self, model: nn.Module) -> torch.Tensor:
        from .Feature_Grid_Model import Feature_Grid_Model
        l1, l2 = [], []                    # tensors still to be reduced here
        pre_l1, pre_l2 = [], []            # terms the model's last decode already holds (their gradients ride in its backward)
        covered = set()
        for m in model.modules():
            if isinstance(m, Feature_Grid_Model):
                cached = m.cached_penalties()
                if cached is not None:
                    pre_l2.append(cached[0])
                    for i, term in cached[1].items():
                        if _l1_parameter(m.drop[i]) is not None:
                            pre_l1.append(term)
                            covered.add(id(m.drop[i]))
                else:
                    l2.extend(m.feature_grid)
        for m in model.modules():
            p = _l1_parameter(m)
            if p is not None and id(m) not in covered:
                l1.append(p)
        use_l1, use_l2 = self.weight_l1 > 0., self.weight_l2 > 0.
        tensors = (l1 if use_l1 else []) + (l2 if use_l2 else [])
        n1 = len(l1) if use_l1 else 0
        sums = None
        if tensors:
            kinds = [_lib.PENALTY_L1] * n1 + [_lib.PENALTY_L2] * (len(tensors) - n1)
            sums = ops.penalty_sums(kinds, tensors)
        loss = 0.
        if use_l1 and (n1 or pre_l1):
            total = sums[:n1].sum() if n1 else 0.
            for t in pre_l1:
                total = total + t
            loss = loss + self.weight_l1 * total
        if use_l2 and (len(tensors) > n1 or pre_l2):
            total = sums[n1:].sum() if len(tensors) > n1 else 0.
            for t in pre_l2:
                total = total + t.sum()
            loss = loss + self.weight_l2 * total
        return loss


class SmallifyDropout(DropoutLayer):

    def __init__(self, size=(1, 1, 1), sign_variance_momentum=0.025, threshold=0.75):
        super().__init__(size, sign_variance_momentum, threshold)
        self.betas = nn.Parameter(torch.empty(size).normal_(0, 1), requires_grad=True)
        self.tracker = SmallifySignVarianceTracker(self.c, sign_variance_momentum, self.threshold, self.betas)
        self.d_mask = None
        self._tracker_stepped = False

    @classmethod
    def _prepare_group(cls, group) -> None:
        """One launch advances the sign trackers of all active layers that share a momentum."""
        by_mom = {}
        for l in group:
            if l.training and l.d_mask is None and l.betas.is_cuda:
                by_mom.setdefault(float(l.tracker.sign_variance_momentum), []).append(l)
        for mom, ls in by_mom.items():
            if len(ls) < 2 or len(ls) > _lib.PENALTY_MAX_TERMS:
                continue
            for l in ls:
                l.tracker._follow(l.betas)
            ops.sign_variance_update_multi([l.betas for l in ls], [l.tracker.EMA for l in ls],
                                           [l.tracker.EMAVar for l in ls], mom)
            for l in ls:
                l._tracker_stepped = True

    def drop_factor(self):
        if not self.training:
            return None                                   # reference :55: eval is the identity
        if self.d_mask is not None:
            return DropFactor(self.d_mask.to(self.betas.device, torch.float32))
        if self._tracker_stepped:
            self._tracker_stepped = False                 # done for this forward by _prepare_group
        else:
            self.tracker.sign_variance_pruning_onlyVar(self.betas)
        return DropFactor(self.betas, None, True)

    def l1_loss(self):
        return ops.penalty_sums([_lib.PENALTY_L1], [self.betas])[0]

    def calculate_pruning_mask(self, device):
        mask = self.tracker.calculate_pruning_mask(device)
        self.d_mask = mask
        return mask

    def multiply_values_with_dropout(self, input, device):
        with torch.no_grad():
            mask = self.calculate_pruning_mask(device) * self.betas.unsqueeze(0)
            return input * mask

    def size_layer(self):
        return self.betas.numel()


class SmallifySignVarianceTracker:
    """EMA of sign(beta) and of its variance; a coefficient whose sign keeps flipping (variance above the threshold) is
    pruned.  State tensors ``EMA`` / ``EMAVar`` follow the device of the betas they are fed."""

    def __init__(self, c, sign_variance_momentum, threshold, betas):
        self.c = c
        self.sign_variance_momentum = sign_variance_momentum
        self.EMA, self.EMAVar = self.init_variance_data(betas)
        self.threshold = threshold

    def init_variance_data(self, betas):
        with torch.no_grad():
            return torch.sign(betas.detach()).clone(), torch.zeros(self.c, device=betas.device)

    def _follow(self, betas):
        if self.EMA.device != betas.device:
            self.EMA = self.EMA.to(betas.device).contiguous()
            self.EMAVar = self.EMAVar.to(betas.device).contiguous()

    def _step(self, betas):
        self._follow(betas)
        ops.sign_variance_update(betas, self.EMA, self.EMAVar, self.sign_variance_momentum)

    def sign_variance_pruning_onlyVar(self, betas):
        self._step(betas)

    def sign_variance_pruning(self, device, betas):
        self._step(betas)
        return self.calculate_pruning_mask(device)

    def calculate_pruning_mask(self, device):
        with torch.no_grad():
            return torch.where(self.EMAVar < self.threshold, 1.0, 0.0).to(device)
