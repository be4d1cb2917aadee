"""Constructor call site of the hot path: mirror of setup_model (model/model_utils.py:23-59).
The binary codec in the same reference file (store_model_parameters / restore_model) is offline host
code and out of scope."""
from __future__ import annotations

import torch

from ..wavelet_transform.Torch_Wavelet_Transform import WaveletFilter3d
from .Feature_Embedding import FourierEmbedding
from .Feature_Grid_Model import Feature_Grid_Model
from .Smallify_Dropout import SmallifyDropout
from .Straight_Through_Dropout import MaskedWavelet_Straight_Through_Dropout, Straight_Through_Dropout
from .Variational_Dropout_Layer import VariationalDropout


def setup_model(input_channel, hidden_channel, out_channel, num_layer, embedding_type, n_embedding_freq, drop_type,
                drop_momentum, drop_threshold, wavelet_filter, grid_features, grid_size, checkpoint_path,
                drop_layer=None, num_levels=None):
    """Same positional arguments as the reference (``drop_type``: '' | 'smallify' | 'straight_through' |
    'masked_straight_through' | anything containing 'variational', reference :35-45).  Extras: ``drop_layer`` = a
    ready prototype object with the DropoutLayer interface (takes precedence), ``num_levels`` = wavelet depth."""
    size_tensor = (grid_features, grid_size, grid_size, grid_size)
    feature_grid = torch.empty(size_tensor).uniform_(0, 1)
    wavelet = WaveletFilter3d(wavelet_filter)
    if drop_type and drop_layer is None:
        size = feature_grid.shape[1:]
        if drop_type == 'smallify':
            drop_layer = SmallifyDropout(size, drop_momentum, drop_threshold)
        if drop_type == 'straight_through':
            drop_layer = Straight_Through_Dropout(size, drop_momentum, drop_threshold)
        if drop_type == 'masked_straight_through':
            drop_layer = MaskedWavelet_Straight_Through_Dropout(size, drop_momentum, drop_threshold)
        if 'variational' in drop_type:
            drop_layer = VariationalDropout(size, drop_momentum, drop_threshold)
        if drop_layer is None:
            raise ValueError('unknown drop_type %r' % (drop_type,))
    embedder = FourierEmbedding(n_freqs=n_embedding_freq, input_dim=input_channel)
    model = Feature_Grid_Model(embedder, feature_grid, drop_layer, wavelet, input_channel_data=input_channel,
                               hidden_channel=hidden_channel, out_channel=out_channel, num_layer=num_layer,
                               num_levels=num_levels)
    if checkpoint_path:
        model.load_state_dict(torch.load(checkpoint_path, weights_only=True))
    return model
