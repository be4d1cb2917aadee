"""ORACLE (test infrastructure, NOT product code) -- pure-PyTorch CPU restatement of the pruning ("drop") layers that
act on the wavelet coefficients inside decode_volume, and of the penalty terms of their losses (SURVEY.md section 8,
row f3).  Only ``tests/`` may import this module.

Functional restatement: every function takes the layer's tensors explicitly and performs the reference's ATen ops in
the reference's order (so CPU results are bit-identical to the reference's modules) and cites the lines it follows.
Random draws (``torch.rand`` / ``torch.randn_like`` in the reference) are ARGUMENTS here.

Pinning: fixtures captured from the reference's own modules by ``tools/make_goldens_drop.py``
(``tests/golden/drop_*.npz``); ``tests/test_oracle_golden.py`` checks every function below against them.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch
import torch.nn.functional as F

from . import ref_torch

K1, K2, K3 = 0.63576, 1.87320, 1.48695          # model/Variational_Dropout_Layer.py:74-77 (Molchanov et al.)


# ---- Smallify (model/Smallify_Dropout.py) ---------------------------------------------------------------

def smallify_apply(x, betas, d_mask=None, training=True):
    """SmallifyDropout.forward (:54-61): train -> x * betas (or x * d_mask once pruned); eval -> x."""
    if not training:
        return x
    if d_mask is None:
        return x.mul(betas.unsqueeze(0))
    return x * d_mask.unsqueeze(0)


def sign_variance_init(betas):
    """SmallifySignVarianceTracker.init_variance_data (:87-91)."""
    return torch.sign(betas).detach(), torch.zeros(betas.shape)


def sign_variance_update(ema, emavar, betas, momentum: float):
    """sign_variance_pruning_onlyVar (:106-112): one EMA / EMA-variance step on sign(betas)."""
    with torch.no_grad():
        new_val = torch.sign(betas)
        phi = new_val - ema
        ema = ema + (momentum * phi)
        emavar = (torch.ones(betas.shape) - momentum) * (emavar + (momentum * (phi ** 2)))
    return ema, emavar


def sign_variance_mask(emavar, threshold: float):
    """calculate_pruning_mask (:114-118)."""
    return torch.where(emavar < threshold, 1.0, 0.0)


def smallify_fold(x, betas, mask):
    """multiply_values_with_dropout (:71-75): coefficients <- coefficients * (mask * betas)."""
    return x * (mask * betas.unsqueeze(0))


def l1_penalty(values):
    """l1_loss (:63-64 and model/Straight_Through_Dropout.py:33-34, :64-65)."""
    return torch.abs(values).sum()


def grid_l2_penalty(coeffs: Sequence[torch.Tensor]):
    """SmallifyLoss._collect_penalties for the model (:29-30) == VariationalDropoutLoss (:52-53)."""
    return sum([torch.sum(torch.abs(f) ** 2) for f in coeffs])


def smallify_loss(l1_terms, coeffs, weight_l1: float, weight_l2: float):
    """SmallifyLoss.forward (:32-40)."""
    loss = 0.
    if weight_l1 > 0.:
        loss = loss + weight_l1 * sum(l1_terms)
    if weight_l2 > 0.:
        loss = loss + weight_l2 * sum([grid_l2_penalty(coeffs)])
    return loss


# ---- straight-through masks (model/Straight_Through_Dropout.py) -----------------------------------------

def ste_apply(x, mask_values, u, training=True):
    """Straight_Through_Dropout.forward (:26-30): x * (u < mask_values); the comparison result is a bool tensor, so no
    gradient reaches mask_values through it (STEFunction.backward :16-17 never runs)."""
    if not training:
        return x
    return x.mul((u < mask_values).unsqueeze(0))


def ste_prune_mask(mask_values, threshold: float):
    """calculate_pruning_mask (:36-37)."""
    return mask_values > threshold


def masked_ste_apply(x, mask_values, threshold: float, d_mask=None, training=True):
    """MaskedWavelet_Straight_Through_Dropout.forward (:54-62): value x*(sigmoid >= thr), gradient of x*sigmoid."""
    if not training:
        return x
    mask = torch.sigmoid(mask_values)
    if d_mask is None:
        return (x * (mask >= threshold) - x * mask).detach() + (x * mask)
    return x * d_mask


def masked_ste_fold(x, mask_values, threshold: float):
    """multiply_values_with_dropout (:72-76); also returns the stored d_mask (:67-70)."""
    mask = torch.sigmoid(mask_values)
    d_mask = (mask >= threshold)
    return (x * (mask >= threshold) - x * mask) + (x * mask), d_mask


# ---- variational dropout (model/Variational_Dropout_Layer.py) -------------------------------------------

def variational_apply(x, log_thetas, log_var, xi, d_mask=None):
    """VariationalDropout.forward (:101-112): w = exp(log_theta) + exp(log_var / 2) * xi, applied in train AND eval."""
    thetas = torch.exp(log_thetas)
    w = thetas + torch.exp(log_var / 2.0) * xi
    if d_mask is None:
        return x * w
    return x * d_mask


def variational_log_alpha(log_thetas, log_var):
    return log_var - 2.0 * log_thetas


def variational_dropout_rates(log_thetas, log_var):
    """alphas / dropout_rates properties (:89-95)."""
    alphas = torch.exp(log_var - 2.0 * log_thetas)
    return alphas / (1.0 + alphas)


def variational_dkl(log_thetas, log_var):
    """calculate_Dkl (:115-122)."""
    log_alphas = log_var - 2.0 * log_thetas
    t1 = K1 * torch.sigmoid(K2 + K3 * log_alphas)
    t2 = 0.5 * F.softplus(-log_alphas, beta=1.)
    return torch.sum(- t1 + t2 + K1)


def variational_entropy(log_thetas, log_var):
    """calculate_Dropout_Entropy (:124-127)."""
    r = variational_dropout_rates(log_thetas, log_var)
    return torch.sum(r * torch.log(r) + (1.0 - r) * torch.log(1 - r))


def variational_prune_mask(log_thetas, log_var, threshold: float):
    """calculate_pruning_mask (:138-148)."""
    with torch.no_grad():
        rates = variational_dropout_rates(log_thetas, log_var)
        mask = torch.where(rates < threshold, 1.0, 0.0)
        if mask.numel() - torch.count_nonzero(mask) == 0:
            mask.data[0] = 1.0
        return mask


def variational_fold(x, log_thetas, mask):
    """multiply_values_with_dropout (:150-154)."""
    return x * (mask * torch.exp(log_thetas))


def log_likelihood_variance(pred, gt, log_sigma):
    """calculate_Log_Likelihood_variance (:27-33)."""
    x_mu = (gt - pred) ** 2
    sigma = torch.exp(log_sigma)
    a = 1 / (2 * (sigma ** 2))
    b = - (math.log(2 * math.pi) + (2 * log_sigma)) / 2
    return a * (-x_mu) + b, x_mu


def variational_loss(dkl_terms, coeffs, pred, gt, log_sigma, size_volume: float, batch_size: float, weight_dkl: float,
                     weight_weights: float, weight_dkl_multiplier: float, weight_dkl_max: float = 30.0):
    """VariationalDropoutLoss.forward (:55-71).  Returns (loss, log-likelihood, mse, dkl term, weight term, the
    annealed weight_dkl that the module keeps for the next call)."""
    batch_scale = size_volume / batch_size
    if weight_dkl < weight_dkl_max:
        weight_dkl = weight_dkl * (1.0 + weight_dkl_multiplier)
    ll, mse = log_likelihood_variance(pred, gt, log_sigma)
    mse = mse.sum() * (1 / pred.shape[0])
    ll = ll.sum() * batch_scale
    dkl_sum = weight_dkl * sum(dkl_terms) * batch_scale
    weight_sum = weight_weights * sum([grid_l2_penalty(coeffs)]) * batch_scale
    loss = -(ll - dkl_sum - weight_sum)
    return loss, ll, mse, dkl_sum, weight_sum, weight_dkl


# ---- decode_volume with the hooks in place (model/Feature_Grid_Model.py:102-108) -------------------------

def decode_volume_dropped(dropped: Sequence[torch.Tensor], shape_array, filter_rev) -> torch.Tensor:
    """decode_volume (:102-108) given the coefficient tensors AFTER their drop layers (``drop[i](feature_grid[i])``)."""
    return ref_torch.decode_volume(list(dropped), shape_array, filter_rev)


def pruned_count(folded: Sequence[torch.Tensor], layer_sizes: Sequence[int]):
    """save_dropvalues_on_grid's return value (:110-128): zeros - (sum of layer sizes) / 32."""
    zeros = 0
    for g in folded:
        zeros += (g.numel() - torch.count_nonzero(g))
    mask_floats = torch.tensor(0, dtype=torch.float32)
    for s in layer_sizes:
        mask_floats += s
    return zeros - mask_floats / 32.0
