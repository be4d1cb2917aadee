"""Sparse variational dropout on the wavelet coefficients: counterpart of model/Variational_Dropout_Layer.py
(likelihood helpers :11-33, VariationalDropoutLoss :36-71, VariationalDropout :74-159, Variance_Model :162-175); same
public names, constructor arguments, parameters (``log_thetas``, ``log_var``) and return values.

MI355X-first: the layer does not touch the coefficients itself.  Its noisy weight
``w = exp(log_theta) + exp(log_var / 2) * xi`` has the layer's size (1/C of the coefficient tensor); ``drop_factor()``
hands it to the fused HIP decode, whose kernels apply it while staging coefficients and return ``d_w`` (sum over the
channels), and autograd carries that to ``log_thetas`` / ``log_var`` through the few small ops below.  Every D_KL term and
the coefficient L2 term of the loss come out of ONE reduction launch (``ops.penalty_sums``).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib, ops
from .Dropout_Layer import DropFactor, DropoutLayer

_LOG_2PI = math.log(2 * np.pi)


# ---- Gaussian likelihood of the prediction (reference :11-33) ---------------------------------------------------

def inference_variational_model(mu, sigma):
    """One sample of N(mu, sigma) per element."""
    return torch.normal(mu, sigma)


def _gaussian_log_density(neg_sq_err, inv_two_var, log_sigma):
    return inv_two_var * neg_sq_err + (-(_LOG_2PI + (2 * log_sigma)) / 2)


def calculate_Log_Likelihood(loss_criterion, predicted_volume, ground_truth_volume, log_sigma):
    """Scalar ``log_sigma`` (a Python float); returns (log-likelihood built on the criterion's value, that value)."""
    fit = loss_criterion(predicted_volume, ground_truth_volume)
    variance = math.exp(log_sigma) ** 2
    return _gaussian_log_density(-fit, 1 / (2 * variance), log_sigma), fit


def calculate_Log_Likelihood_variance(predicted_volume, ground_truth_volume, variance):
    """Per-sample ``variance`` = log sigma (a tensor); returns (per-sample log-likelihood, per-sample squared error)."""
    sq_err = (ground_truth_volume - predicted_volume) ** 2
    sigma = torch.exp(variance)
    return _gaussian_log_density(-sq_err, 1 / (2 * (sigma ** 2)), variance), sq_err


class VariationalDropoutLoss(nn.Module):
    """``-(LL - weight_dkl * D_KL - weight_weights * |coefficients|^2)``, every term scaled from the batch to the whole
    volume; ``weight_dkl`` anneals upwards by ``(1 + multiplier)`` per call until it passes ``weight_dkl_max``."""

    def __init__(self, size_volume: float, batch_size: float, weight_dkl: float = 1., weight_weights: float = 1.,
                 weight_dkl_max=30.0):
        super().__init__()
        self.batch_scale = (size_volume / batch_size)
        self.weight_dkl = float(weight_dkl)
        self.weight_dkl_max = weight_dkl_max
        self.weight_weights = float(weight_weights)

    def _penalties(self, model):
        """(sum of the layers' D_KL, sum of squared coefficients) -- one launch for all of them."""
        from .Feature_Grid_Model import Feature_Grid_Model
        layers = [m for m in model.modules() if isinstance(m, VariationalDropout)]
        grids, pre_sq = [], 0
        for m in model.modules():
            if isinstance(m, Feature_Grid_Model):
                cached = m.cached_penalties()          # sums taken inside the last decode: gradients ride in its backward
                if cached is not None:
                    pre_sq = pre_sq + cached[0].sum()
                else:
                    grids += list(m.feature_grid)
        kinds = [_lib.PENALTY_DKL] * len(layers) + [_lib.PENALTY_L2] * len(grids)
        if not kinds:
            return 0, pre_sq
        tensors = [t for l in layers for t in (l.log_thetas, l.log_var)] + grids
        sums = ops.penalty_sums(kinds, tensors)
        return (sums[:len(layers)].sum() if layers else 0), (sums[len(layers):].sum() + pre_sq if grids else pre_sq)

    def forward(self, model: nn.Module, predicted_volume, ground_truth_volume, log_sigma, weight_dkl_multiplier):
        dkl, sq_weights = self._penalties(model)
        if self.weight_dkl < self.weight_dkl_max:
            self.weight_dkl = self.weight_dkl * (1.0 + weight_dkl_multiplier)
        per_sample_ll, sq_err = calculate_Log_Likelihood_variance(predicted_volume, ground_truth_volume, log_sigma)
        mse = sq_err.sum() * (1 / predicted_volume.shape[0])
        log_likelihood = per_sample_ll.sum() * self.batch_scale
        dkl_term = self.weight_dkl * dkl * self.batch_scale
        weight_term = self.weight_weights * sq_weights * self.batch_scale
        return -(log_likelihood - dkl_term - weight_term), log_likelihood, mse, dkl_term, weight_term


class VariationalDropout(DropoutLayer):
    # D_KL approximation constants of Molchanov et al. 2017 (reference :75-78); the penalty kernel uses the same values
    k1 = 0.63576
    k2 = 1.87320
    k3 = 1.48695
    C = -k1

    def __init__(self, size=(1, 1, 1), init_dropout=0.5, threshold=0.9):
        super().__init__(size, init_dropout, threshold)
        log_alpha0 = math.log(init_dropout / (1 - init_dropout))
        self.log_thetas = nn.Parameter(torch.zeros(size), requires_grad=True)
        self.log_var = nn.Parameter(torch.full(size, log_alpha0), requires_grad=True)      # log sigma^2 = log(theta^2 alpha)
        self.d_mask = None

    # ---- derived quantities ------------------------------------------------------------------------------------
    @property
    def alphas(self):
        return torch.exp(self.log_var - 2.0 * self.log_thetas)

    @property
    def sigma(self):
        return torch.exp(self.log_var / 2.0)

    @property
    def dropout_rates(self):
        a = self.alphas
        return a / (1.0 + a)

    # ---- the layer's factor for the fused decode ---------------------------------------------------------------
    def _draw(self):
        """xi ~ N(0, 1) of the layer's size (torch's generator; the parity tests substitute recorded draws)."""
        return torch.randn_like(self.log_thetas)

    def drop_factor(self):
        # as in the reference the noise is injected in train AND eval mode; after pruning only the mask
        if self.d_mask is not None:
            return DropFactor(self.d_mask.to(self.log_thetas.device, torch.float32))
        return DropFactor(torch.exp(self.log_thetas) + self.sigma * self._draw())

    # ---- diagnostics / loss terms -------------------------------------------------------------------------------
    def calculate_Dkl(self):
        return ops.penalty_sums([_lib.PENALTY_DKL], [self.log_thetas, self.log_var])[0]

    def calculate_Dropout_Entropy(self):
        p = self.dropout_rates
        return torch.sum(p * torch.log(p) + (1.0 - p) * torch.log(1 - p))

    def get_valid_fraction(self):
        rates = self.dropout_rates
        return torch.mean((rates < self.threshold).to(torch.float)).item(), rates

    # ---- pruning --------------------------------------------------------------------------------------------------
    def calculate_pruning_mask(self, device):
        with torch.no_grad():
            keep = torch.where(self.dropout_rates < self.threshold, 1.0, 0.0)
            if keep.numel() - torch.count_nonzero(keep) == 0:
                keep.data[0] = 1.0                     # reference quirk (:144-145), kept
            self.d_mask = keep.to(device)
            return keep.to(device)

    def multiply_values_with_dropout(self, input, device):
        with torch.no_grad():
            return input * (self.calculate_pruning_mask(device) * torch.exp(self.log_thetas))

    def size_layer(self):
        return self.log_thetas.numel()


class Variance_Model(nn.Module):
    """Small ReLU MLP predicting a per-sample log-sigma ('dynamic' variational mode, reference :162-175).  Plain torch:
    it is not part of the feature-grid path."""

    def __init__(self, input_ch=3, output_ch=1, n_layers=4, size_layers=32):
        super().__init__()
        widths = [input_ch] + [size_layers] * n_layers
        self.net_layers = nn.ModuleList([nn.Linear(a, b) for a, b in zip(widths[:-1], widths[1:])])
        self.final_layer = nn.Linear(size_layers, output_ch)

    def forward(self, input):
        out = input
        for layer in self.net_layers:
            out = F.relu(layer(out))
        return self.final_layer(out)
