"""CPU: pin oracle/ref_codec.py (binary checkpoint format) against the files the reference's
own store_model_parameters wrote and the state its restore_model rebuilt (tools/make_goldens_codec.py)."""
import os

import numpy as np

from oracle import ref_codec as K


def test_parse_matches_reference_restore(golden_dir):
    g = np.load(os.path.join(golden_dir, 'codec_small.npz'))
    C, G, H, L, nf = [int(v) for v in g['meta']]
    p = K.parse(g['param_file'].tobytes(), g['mask_file'].tobytes())
    h = p['header']
    assert (h['n_layers'], h['layer_width'], h['input_dim'], h['input_channel'], h['output_dim']) == (L, H, 15 + C, 3, 1)
    assert (h['bit_precision'], h['grid_size'], h['n_grids'], h['feature_size']) == (8, G, len(g['shape_array']) + 1, C)
    for i in range(h['n_grids']):
        src = g['sd.feature_grid.%d' % i]
        assert h['grid_sizes'][i] == np.count_nonzero(src) and h['zeros'][i] == src.size - np.count_nonzero(src)
        assert np.array_equal(p['grids'][i].reshape(src.shape), g['restored.feature_grid.%d' % i])
        assert np.array_equal(p['grids'][i] == 0, src.reshape(-1) == 0)            # the mask is the zero pattern
    for i in range(L):
        assert np.array_equal(p['weights'][i].reshape(g['restored.net_layers.%d.weight' % i].shape),
                              g['restored.net_layers.%d.weight' % i])
        assert np.array_equal(p['biases'][i], g['restored.net_layers.%d.bias' % i])
    assert np.array_equal(p['weights'][L].reshape(1, H), g['restored.final_layer.weight'])
    assert np.array_equal(p['biases'][L], g['restored.final_layer.bias'])
    # unquantised parts are the stored model's exactly; quantised ones are one of the 256 centres each
    assert np.array_equal(p['weights'][0].reshape(H, 15 + C), g['sd.net_layers.0.weight'])
    assert np.array_equal(p['weights'][L].reshape(1, H), g['sd.final_layer.weight'])
    for b in p['blocks']:
        assert b['centres'].shape == (256,) and b['labels'].min() >= 0 and b['labels'].max() < 256


def test_serialize_reproduces_reference_bytes(golden_dir):
    g = np.load(os.path.join(golden_dir, 'codec_small.npz'))
    p = K.parse(g['param_file'].tobytes(), g['mask_file'].tobytes())
    param, mask = K.serialize(p['header'], p['weights'], p['biases'], p['blocks'], p['mask'])
    assert param == g['param_file'].tobytes()
    assert mask == g['mask_file'].tobytes()


def test_label_bit_packing_forms():
    rng = np.random.default_rng(1)
    for bits in (8, 4, 2, 1, 16):                 # widths that tile bytes: exact round trip
        lab = rng.integers(0, 2 ** bits, 37 if bits >= 8 else 64)
        assert np.array_equal(K.unpack_labels(K.pack_labels(lab, bits), lab.size, bits), lab)
    # a width that does not tile bytes: the writer right-aligns the leftover chunk (reference quirk), so only labels that
    # end before the last partial byte survive the reader's MSB-first slicing
    lab = np.asarray([5, 2, 7, 1, 6])
    raw = K.pack_labels(lab, 3)
    assert len(raw) == 2 and raw[1] == int('1001110'[0:7], 2)
    assert np.array_equal(K.unpack_labels(raw, 5, 3)[:2], lab[:2])
