"""Host-side mirror of model/Feature_Embedding.py (Embedder :5-16, FourierEmbedding :20-34).

Inside Feature_Grid_Model.forward the embedding is fused into the HIP kernel; this object carries the
constructor contract (``out_dim``, ``n_freqs``) and keeps ``embed`` callable for code that uses it on
its own (plain torch ops, any device)."""
from __future__ import annotations

import numpy as np
import torch


class Embedder:
    def __init__(self):
        self.embed_functions = []
        self.out_dim = 0

    def create_embedding_function(self):
        pass

    def embed(self, inputs):
        return torch.cat([fn(inputs) for fn in self.embed_functions], -1)


class FourierEmbedding(Embedder):
    def __init__(self, n_freqs, input_dim):
        super().__init__()
        self.n_freqs = int(n_freqs)
        self.input_dim = int(input_dim)
        self.periodic_functions = [torch.sin, torch.cos]
        self.create_embedding_function(self.n_freqs, self.input_dim)

    def create_embedding_function(self, n_freqs, input_dim):
        freq_bands = (2. ** torch.linspace(0., n_freqs - 1, steps=n_freqs)) * 2. * np.pi
        for freq in freq_bands:
            for p_fn in self.periodic_functions:
                self.embed_functions.append(lambda x, p_fn=p_fn, freq=freq: p_fn(x * freq))
                self.out_dim += input_dim
