"""Counterpart of model/model_utils.py: ``setup_model`` (:23-59, the constructor call site of the hot path) and the
binary checkpoint codec ``store_model_parameters`` / ``restore_model`` (:120-332) with the same signatures and the SAME
FILE FORMAT, so files written by either implementation are read by the other.

The reference's codec does its per-coefficient work in Python (mask string concatenation :207-208, one ``np.insert``
per pruned coefficient :302-305, scikit-learn k-means on the host) and is unusable beyond toy sizes; here that work runs
on the GPU through the C-ABI (``lfgc_codec_*``: bit mask, order-preserving compaction / expansion, 1-D k-means, label
dequantisation) and only the file layout itself (header, layer fields, byte streams) is host code below.
"""
from __future__ import annotations

import math
import os
import re
import struct

import numpy as np
import torch

from .. import ops

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


def write_dict(dictionary, filename, experiment_path=''):
    with open(os.path.join(experiment_path, filename), 'w') as f:
        for key, value in dictionary.items():
            f.write('%s = %s\n' % (key, value))


def get_net_weights_biases(net):
    """Parameters whose names end in .weight / .bias, in named_parameters() order (reference :62-70)."""
    weights = [p.data for name, p in net.named_parameters() if name.endswith('.weight')]
    biases = [p.data for name, p in net.named_parameters() if name.endswith('.bias')]
    return weights, biases


def kmeans_quantization(w, q):
    """(labels, centres) as Python lists like the reference helper (:65-70); the clustering runs on the GPU."""
    t = torch.as_tensor(np.asarray(w, dtype=np.float32).reshape(-1))
    centres, labels = ops.codec_kmeans(t.cuda(), int(q))
    return labels.cpu().tolist(), centres.cpu().tolist()


_BIT_PRECISION = 8        # the reference hard-codes 8 (:141); the reader below accepts any width the header names


def _device_flat(t: torch.Tensor) -> torch.Tensor:
    if not torch.cuda.is_available():
        raise ops._lib.LfgcError('the checkpoint codec runs its per-coefficient work on the MI355X: no GPU is visible. '
                                 'There is no CPU fallback.')
    return t.detach().to('cuda', torch.float32).reshape(-1)


def _f32_bytes(t) -> bytes:
    return np.ascontiguousarray(t.detach().cpu().numpy().reshape(-1).astype('<f4')).tobytes()


def _quantised_block(values: torch.Tensor) -> bytes:
    """[2^8 fp32 centres][one label byte per value] (reference write_tensor_quantized :170-182)."""
    centres, labels = ops.codec_kmeans(values, 1 << _BIT_PRECISION)
    return _f32_bytes(centres) + labels.cpu().numpy().tobytes()


def store_model_parameters(model, filename):
    """Write ``filename`` (header, first / final layer in fp32, hidden layers and the non-zero wavelet coefficients as
    8-bit codebook indices) and ``filename + "_mask.bnr"`` (bit mask of the non-zero coefficients): reference :120-223."""
    if len(model.shape_array) == 0:
        raise ValueError('a model without wavelet levels has no grid_size to put in the header (reference :131)')
    grids = [_device_flat(g) for g in model.feature_grid]
    nonzero = [ops.codec_compact(g) for g in grids]
    header = struct.pack('9B', model.num_layer, model.hidden_width, model.input_channel, model.d_in, model.output_channel,
                         _BIT_PRECISION, int(model.shape_array[-1][0]), len(grids), int(model.feature_grid[0].shape[0]))
    weights, biases = get_net_weights_biases(model)
    with open(filename, 'wb') as f:
        f.write(header)
        for g, nz in zip(grids, nonzero):
            f.write(struct.pack('<I', nz.numel()))
        for g, nz in zip(grids, nonzero):
            f.write(struct.pack('<I', g.numel() - nz.numel()))
        f.write(_f32_bytes(weights[0]))
        f.write(_f32_bytes(biases[0]))
        for w, b in zip(weights[1:-1], biases[1:-1]):
            f.write(_quantised_block(_device_flat(w)))
            f.write(_f32_bytes(b))
        f.write(_f32_bytes(weights[-1]))
        f.write(_f32_bytes(biases[-1]))
        for nz in nonzero:
            if nz.numel() == 0:
                raise ValueError('a coefficient tensor without any non-zero entry cannot be quantised (the reference '
                                 'fails in KMeans here as well)')
            f.write(_quantised_block(nz))
    mask = ops.codec_mask(torch.cat(grids))           # one bit stream over all tensors, like the reference's mask_string
    with open(filename + '_mask.bnr', 'wb') as f:
        f.write(mask.cpu().numpy().tobytes())


def restore_model(filename):
    """Rebuild the model a parameter file + mask file describe (reference :226-332).  Like the reference the network is
    re-created with 'fourier' embedding (2 frequencies), db2 and no drop layers; unlike it the model is returned on the GPU
    (it cannot run anywhere else)."""
    with open(filename, 'rb') as f:
        raw = f.read()
    with open(filename + '_mask.bnr', 'rb') as f:
        mask_raw = f.read()
    pos = 0

    def take(fmt):
        nonlocal pos
        size = struct.calcsize(fmt)
        if pos + size > len(raw):
            raise ValueError('%s: truncated parameter file' % filename)
        vals = struct.unpack(fmt, raw[pos:pos + size])
        pos += size
        return vals

    n_layers, layer_width, input_dim, input_channel, output_dim, bits, grid_size, n_grids, feature_size = take('9B')
    if not 1 <= bits <= 16:
        raise ValueError('%s: bit precision %d not supported' % (filename, bits))
    n_clusters = int(math.pow(2, bits))
    grid_sizes = [take('<I')[0] for _ in range(n_grids)]
    zeros = [take('<I')[0] for _ in range(n_grids)]
    dev = torch.device('cuda')

    def floats(n):
        nonlocal pos
        if pos + 4 * n > len(raw):
            raise ValueError('%s: truncated parameter file' % filename)
        a = np.frombuffer(raw, dtype='<f4', count=n, offset=pos)
        pos += 4 * n
        return torch.from_numpy(a.astype(np.float32))

    def quantised(n):
        nonlocal pos
        centres = floats(n_clusters).to(dev)
        nbytes = (n * bits) // 8 + (1 if (n * bits) % 8 else 0)
        if pos + nbytes > len(raw):
            raise ValueError('%s: truncated parameter file' % filename)
        packed = torch.from_numpy(np.frombuffer(raw, dtype=np.uint8, count=nbytes, offset=pos).copy()).to(dev)
        pos += nbytes
        out = ops.codec_dequant(packed, bits, n, centres) if n else torch.empty(0, device=dev)
        if bits % 8 != 0:                       # the writer repeats the last label as a uint32 (:184-186, :265-267)
            last = take('<I')[0]
            if n:
                out[-1] = centres[last]
        return out

    net_weights = [floats(input_dim * layer_width).to(dev)]
    net_biases = [floats(layer_width).to(dev)]
    for _ in range(n_layers - 1):
        net_weights.append(quantised(layer_width * layer_width))
        net_biases.append(floats(layer_width).to(dev))
    net_weights.append(floats(output_dim * layer_width).to(dev))
    net_biases.append(floats(output_dim).to(dev))

    total = sum(grid_sizes) + sum(zeros)
    if len(mask_raw) * 8 < total:
        raise ValueError('%s_mask.bnr: %d bits needed, %d present' % (filename, total, len(mask_raw) * 8))
    mask = torch.from_numpy(np.frombuffer(mask_raw, dtype=np.uint8).copy()).to(dev)
    grid_params, bit_offset = [], 0
    for nz, z in zip(grid_sizes, zeros):
        values = quantised(nz)
        grid_params.append(ops.codec_expand(mask, bit_offset, nz + z, values))
        bit_offset += nz + z

    model = setup_model(input_channel=input_channel, hidden_channel=layer_width, out_channel=output_dim,
                        num_layer=n_layers, embedding_type='fourier', n_embedding_freq=2, drop_type='',
                        drop_momentum=0.025, drop_threshold=0.75, wavelet_filter='db2', grid_features=feature_size,
                        grid_size=grid_size, checkpoint_path='').to(dev)
    wdx = bdx = gdx = 0
    for name, p in model.named_parameters():        # same name tests, same order as the reference (:316-328)
        if re.match(r'.*grid.*', name, re.I):
            p.data = grid_params[gdx].view(p.data.shape)
            gdx += 1
        if re.match(r'.*.weight', name, re.I):
            p.data = net_weights[wdx].view(p.data.shape)
            wdx += 1
        if re.match(r'.*.bias', name, re.I):
            p.data = net_biases[bdx].view(p.data.shape)
            bdx += 1
    return model
