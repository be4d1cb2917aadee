"""Ground-truth sampler: mirror of data/Interpolation.py::trilinear_f_interpolation (:8-44), same
signature, computed by a HIP kernel that follows the reference's operation order bit for bit."""
from __future__ import annotations

import torch

from .. import ops


def trilinear_f_interpolation(p, f, min_bb, max_bb, res):
    """p (N,3) raw lattice positions, f (X,Y,Z) volume, min_bb/max_bb/res (3,) float tensors -> (N,)."""
    return ops.gt_interp(p, f, min_bb.detach().cpu(), max_bb.detach().cpu(), res.detach().cpu())
