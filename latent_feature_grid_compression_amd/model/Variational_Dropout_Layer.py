"""Sparse variational dropout on the wavelet coefficients: counterpart of model/Variational_Dropout_Layer.py
(likelihood helpers :11-33, VariationalDropoutLoss :36-71, VariationalDropout :74-159, Variance_Model :162-175).

The noisy weight ``w = exp(log_theta) + exp(log_var / 2) * xi`` is a tensor of the layer's size (1/C of the
coefficients); it is handed to the fused HIP decode as the layer's factor, the kernels return ``d_w`` and autograd
carries it to ``log_thetas`` / ``log_var``.  All D_KL terms and the coefficient L2 term of the loss are evaluated in one
reduction launch.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib, ops
from .Dropout_Layer import DropFactor, DropoutLayer


def inference_variational_model(mu, sigma):
    return torch.normal(mu, sigma)


def calculate_Log_Likelihood(loss_criterion, predicted_volume, ground_truth_volume, log_sigma):
    """Gaussian log-likelihood with one scalar log-sigma (reference :16-23)."""
    x_mu_loss = loss_criterion(predicted_volume, ground_truth_volume)
    sigma = math.exp(log_sigma)
    a = 1 / (2 * (sigma ** 2))
    b = - (math.log(2 * math.pi) + (2 * log_sigma)) / 2
    return a * (-x_mu_loss) + b, x_mu_loss


def calculate_Log_Likelihood_variance(predicted_volume, ground_truth_volume, variance):
    """Per-sample Gaussian log-likelihood, ``variance`` = log-sigma per sample (reference :26-33)."""
    x_mu_loss = (ground_truth_volume - predicted_volume) ** 2
    sigma = torch.exp(variance)
    a = 1 / (2 * (sigma ** 2))
    b = - (math.log(2 * np.pi) + (2 * variance)) / 2
    return a * (-x_mu_loss) + b, x_mu_loss


class VariationalDropoutLoss(nn.Module):
    """-(log-likelihood - weight_dkl * D_KL - weight_weights * |coefficients|^2), each scaled to the whole volume;
    ``weight_dkl`` is annealed upwards on every call until ``weight_dkl_max`` (reference :55-71)."""

    def __init__(self, size_volume: float, batch_size: float, weight_dkl: float = 1., weight_weights: float = 1.,
                 weight_dkl_max=30.0):
        super().__init__()
        self.batch_scale = (size_volume / batch_size)
        self.weight_dkl = float(weight_dkl)
        self.weight_dkl_max = weight_dkl_max
        self.weight_weights = float(weight_weights)

    def forward(self, model: nn.Module, predicted_volume, ground_truth_volume, log_sigma, weight_dkl_multiplier):
        from .Feature_Grid_Model import Feature_Grid_Model
        kinds, tensors, n_dkl = [], [], 0
        for m in model.modules():
            if isinstance(m, VariationalDropout):
                kinds.append(_lib.PENALTY_DKL)
                tensors += [m.log_thetas, m.log_var]
                n_dkl += 1
        for m in model.modules():
            if isinstance(m, Feature_Grid_Model):
                kinds += [_lib.PENALTY_L2] * len(m.feature_grid)
                tensors += list(m.feature_grid)
        sums = ops.penalty_sums(kinds, tensors) if kinds else None
        dkl = sums[:n_dkl].sum() if n_dkl else 0
        weights = sums[n_dkl:].sum() if len(kinds) > n_dkl else 0

        if self.weight_dkl < self.weight_dkl_max:
            self.weight_dkl = self.weight_dkl * (1.0 + weight_dkl_multiplier)

        Log_Likelyhood, mse = calculate_Log_Likelihood_variance(predicted_volume, ground_truth_volume, log_sigma)
        mse = mse.sum() * (1 / predicted_volume.shape[0])
        Log_Likelyhood = Log_Likelyhood.sum() * self.batch_scale
        Dkl_sum = self.weight_dkl * dkl * self.batch_scale
        weight_sum = self.weight_weights * weights * self.batch_scale
        loss = -(Log_Likelyhood - Dkl_sum - weight_sum)
        return loss, Log_Likelyhood, mse, Dkl_sum, weight_sum


class VariationalDropout(DropoutLayer):
    # constants of the D_KL approximation of Molchanov et al. (reference :75-78)
    k1 = 0.63576
    k2 = 1.87320
    k3 = 1.48695
    C = -k1

    def __init__(self, size=(1, 1, 1), init_dropout=0.5, threshold=0.9):
        super().__init__(size, init_dropout, threshold)
        self.log_thetas = nn.Parameter(torch.zeros(size), requires_grad=True)
        log_alphas = math.log(init_dropout / (1 - init_dropout))
        self.log_var = nn.Parameter(torch.empty(size).fill_(log_alphas), requires_grad=True)   # log sigma^2
        self.d_mask = None

    @property
    def alphas(self):
        return torch.exp(self.log_var - 2.0 * self.log_thetas)

    @property
    def dropout_rates(self):
        return self.alphas / (1.0 + self.alphas)

    @property
    def sigma(self):
        return torch.exp(self.log_var / 2.0)

    def _draw(self):
        return torch.randn_like(self.log_thetas)

    def drop_factor(self):
        # noise is injected in train AND eval mode, as in the reference
        if self.d_mask is not None:
            return DropFactor(self.d_mask.to(self.log_thetas.device, torch.float32))
        thetas = torch.exp(self.log_thetas)
        xi = self._draw()
        return DropFactor(thetas + self.sigma * xi)

    def calculate_Dkl(self):
        return ops.penalty_sums([_lib.PENALTY_DKL], [self.log_thetas, self.log_var])[0]

    def calculate_Dropout_Entropy(self):
        drop_rate = self.dropout_rates
        h = drop_rate * torch.log(drop_rate) + (1.0 - drop_rate) * torch.log(1 - drop_rate)
        return torch.sum(h)

    def get_valid_fraction(self):
        rates = self.dropout_rates
        not_dropped = torch.mean((rates < self.threshold).to(torch.float)).item()
        return not_dropped, rates

    def calculate_pruning_mask(self, device):
        with torch.no_grad():
            prune_mask = torch.where(self.dropout_rates < self.threshold, 1.0, 0.0)
            if prune_mask.numel() - torch.count_nonzero(prune_mask) == 0:
                prune_mask.data[0] = 1.0
            self.d_mask = prune_mask.to(device)
            return prune_mask.to(device)

    def multiply_values_with_dropout(self, input, device):
        with torch.no_grad():
            mask = self.calculate_pruning_mask(device) * torch.exp(self.log_thetas)
            return input * mask

    def size_layer(self):
        return self.log_thetas.numel()


class Variance_Model(nn.Module):
    """Small ReLU MLP predicting a per-sample log-sigma ('dynamic' variational mode, reference :162-175).  Plain
    torch: it is not part of the feature-grid path."""

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
