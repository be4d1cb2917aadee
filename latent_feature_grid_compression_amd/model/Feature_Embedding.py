"""Fourier positional embedding: counterpart of model/Feature_Embedding.py (Embedder :5-16, FourierEmbedding :20-34).

Inside ``Feature_Grid_Model.forward`` the embedding is computed by the fused HIP kernel; this object carries the
constructor contract the model needs (``out_dim``, ``n_freqs``) and keeps ``embed`` usable on its own (plain torch
ops on any device) for code that calls it directly.  Term order and frequencies are the reference's:
``[sin(f_0 p), cos(f_0 p), sin(f_1 p), cos(f_1 p), ...]`` with ``f_k = fp32(2^k) * 2 * pi`` formed in fp32.
"""
from __future__ import annotations

from typing import Callable, List

import numpy as np
import torch


class Embedder:
    """A list of maps ``p -> features``; ``embed`` concatenates their outputs along the last axis."""

    def __init__(self):
        self.embed_functions: List[Callable] = []
        self.out_dim: int = 0

    def create_embedding_function(self, *args, **kwargs):
        return None

    def embed(self, inputs):
        return torch.cat([term(inputs) for term in self.embed_functions], dim=-1)


class _PeriodicTerm:
    """``p -> fn(p * freq)`` with ``freq`` a 0-d fp32 tensor (the product is formed in fp32 like the reference's)."""

    __slots__ = ('fn', 'freq')

    def __init__(self, fn, freq):
        self.fn, self.freq = fn, freq

    def __call__(self, p):
        return self.fn(p * self.freq)


class FourierEmbedding(Embedder):
    def __init__(self, n_freqs, input_dim):
        super().__init__()
        self.n_freqs, self.input_dim = int(n_freqs), int(input_dim)
        self.periodic_functions = [torch.sin, torch.cos]
        self.create_embedding_function(self.n_freqs, self.input_dim)

    def create_embedding_function(self, n_freqs, input_dim):
        octaves = torch.linspace(0., n_freqs - 1, steps=n_freqs)
        self.freq_bands = (2. ** octaves) * 2. * np.pi              # fp32: (2^k * 2) * pi
        self.embed_functions = [_PeriodicTerm(fn, f) for f in self.freq_bands for fn in self.periodic_functions]
        self.out_dim = input_dim * len(self.embed_functions)
