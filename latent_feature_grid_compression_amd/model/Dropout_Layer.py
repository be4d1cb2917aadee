"""Base class of the pruning ("drop") layers: counterpart of model/Dropout_Layer.py (class DropoutLayer :4-39).

Same constructor, attributes (``c``, ``p``, ``threshold``) and class-level threshold list as the reference, so
``drop_layer.create_instance(size, drop_layer.p, drop_layer.threshold)`` in Feature_Grid_Model.__init__
(model/Feature_Grid_Model.py:30-32) and ``DropoutLayer.set_threshold_list`` keep working.

What is new is the protocol the MI355X decode uses: every reference drop layer scales a coefficient tensor ``(C, ...)``
by a per-coefficient factor of shape ``(...)`` shared by the channels, so a layer here only says WHICH factor
(``drop_factor()``) and the multiply itself -- forward and backward -- runs inside the HIP inverse-wavelet kernels
(``lfgc_idwt_level_drop_f32``); ``forward(x)`` on its own is the same arithmetic as one HIP elementwise kernel.
"""
from __future__ import annotations

from typing import NamedTuple, Optional

import torch

from .. import ops


class DropFactor(NamedTuple):
    """``mul``: tensor of the layer's ``size`` (autograd-connected to the layer's parameters where the reference's is).
    ``threshold``: None -> value = x * mul; a float -> masked straight-through rule: the value of x * (mul >= threshold)
    with the gradient of x * mul."""
    mul: torch.Tensor
    threshold: Optional[float] = None
    l1_target: bool = False      # ``mul`` IS the parameter the layer's L1 penalty is taken of (Smallify betas): its penalty
                                 # gradient can then ride in the decode's adjoint kernels


class DropoutLayer(torch.nn.Module):
    # the reference hands thresholds out by construction order through class state (model/Dropout_Layer.py:6-19):
    # the i-th layer constructed after set_threshold_list() takes entry i-1 (the prototype built first keeps its own)
    i = 0
    theshold_list = None

    def __init__(self, size=0, p: float = 0.5, threshold: float = 0.9):
        super().__init__()
        self.c = size
        self.p = p
        self.threshold = threshold
        cls = DropoutLayer
        if cls.theshold_list is not None and cls.i != 0:
            self.threshold = cls.theshold_list[cls.i - 1]
        cls.i += 1

    # ---- protocol of the fused decode -------------------------------------------------------------------
    def drop_factor(self) -> Optional[DropFactor]:
        """Factor this layer applies in its current mode (None = identity).  Called exactly once per forward: layers
        with per-forward side effects (sign tracker step, random draw) perform them here."""
        return None

    @staticmethod
    def prepare(layers) -> None:
        """Called by the model once per decode with all its drop layers, before their drop_factor(): lets a layer type
        batch its per-forward side effects over the layers (one launch instead of one per layer)."""
        groups = {}
        for layer in layers:
            if isinstance(layer, DropoutLayer):
                groups.setdefault(type(layer), []).append(layer)
        for cls, group in groups.items():
            cls._prepare_group(group)

    @classmethod
    def _prepare_group(cls, group) -> None:
        return None

    def forward(self, x):
        f = self.drop_factor()
        if f is None:
            return x
        return ops.DropApplyFn.apply(x, f.mul, f.threshold)

    # ---- reference interface (model/Dropout_Layer.py:21-30) ------------------------------------------------
    def calculate_pruning_mask(self, device):
        return None

    def multiply_values_with_dropout(self, input, device):
        return None

    def size_layer(self):
        return None

    @classmethod
    def set_threshold_list(cls, list):
        # as in the reference the state lands on the class the method is CALLED on, while __init__ reads the base
        # class: only DropoutLayer.set_threshold_list(...) has an effect (probed against the reference)
        cls.i = 0
        cls.theshold_list = list

    @classmethod
    def create_instance(cls, size, sign_variance_momentum=0.02, threshold=0.9):
        return cls(size, sign_variance_momentum, threshold)
