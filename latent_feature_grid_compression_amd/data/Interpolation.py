"""Ground-truth sampler: mirror of data/Interpolation.py::trilinear_f_interpolation (:8-44), same
signature, computed by a HIP kernel that follows the reference's operation order bit for bit."""
from __future__ import annotations

import torch

from .. import ops


def trilinear_f_interpolation(p, f, min_bb, max_bb, res):
    """p (N,3) raw lattice positions, f (X,Y,Z) volume, min_bb/max_bb/res (3,) float tensors -> (N,)."""
    return ops.gt_interp(p, f, min_bb.detach().cpu(), max_bb.detach().cpu(), res.detach().cpu())


def trilinear_mse_loss(pred, p, f, min_bb, max_bb, res):
    """``torch.nn.MSELoss()(pred, trilinear_f_interpolation(p, f, min_bb, max_bb, res))`` (training/training.py:107-109,
    :127) as one fused HIP pass: same ground truth bit for bit, fp64-accumulated mean, analytic gradient to ``pred``."""
    host = lambda v: v.detach().cpu().tolist() if torch.is_tensor(v) else list(v)
    return ops.gt_mse_loss(pred, p, f, host(min_bb), host(max_bb), host(res))
