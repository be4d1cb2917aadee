"""Constructor call site of the hot path: mirror of setup_model (model/model_utils.py:23-59).
The binary codec in the same reference file (store_model_parameters / restore_model) is offline host
code and out of scope."""
from __future__ import annotations

import torch

from ..wavelet_transform.Torch_Wavelet_Transform import WaveletFilter3d
from .Feature_Embedding import FourierEmbedding
from .Feature_Grid_Model import Feature_Grid_Model


def setup_model(input_channel, hidden_channel, out_channel, num_layer, embedding_type, n_embedding_freq, drop_type,
                drop_momentum, drop_threshold, wavelet_filter, grid_features, grid_size, checkpoint_path,
                drop_layer=None, num_levels=None):
    """Same positional arguments as the reference.  ``drop_type`` other than ''/None needs the reference's
    pruning layers (model/*Dropout*.py, out of scope here): pass a ready ``drop_layer`` prototype object
    with the reference's DropoutLayer interface instead and it is plugged in unchanged."""
    size_tensor = (grid_features, grid_size, grid_size, grid_size)
    feature_grid = torch.empty(size_tensor).uniform_(0, 1)
    wavelet = WaveletFilter3d(wavelet_filter)
    if drop_type and drop_layer is None:
        raise NotImplementedError("drop_type=%r: the pruning layers are outside this package; construct the "
                                  "reference's drop layer and pass it as drop_layer=" % (drop_type,))
    embedder = FourierEmbedding(n_freqs=n_embedding_freq, input_dim=input_channel)
    model = Feature_Grid_Model(embedder, feature_grid, drop_layer, wavelet, input_channel_data=input_channel,
                               hidden_channel=hidden_channel, out_channel=out_channel, num_layer=num_layer,
                               num_levels=num_levels)
    if checkpoint_path:
        model.load_state_dict(torch.load(checkpoint_path, weights_only=True))
    return model
