"""CPU: pin the oracle (oracle/ref_torch.py, oracle/ref_explicit.py) against the fixtures captured
from the reference's own modules (tools/make_goldens.py).  The op-for-op torch restatement must be
BIT-IDENTICAL on the same torch build; the explicit fp64 restatement must agree to fp32 rounding."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import ref_torch as R
from oracle import ref_explicit as E

torch.set_num_threads(4)


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def model_from_sd(g, prefix='sd.'):
    L = 0
    while prefix + 'net_layers.%d.weight' % L in g:
        L += 1
    n = 0
    while prefix + 'feature_grid.%d' % n in g:
        n += 1
    coeffs = [torch.from_numpy(g[prefix + 'feature_grid.%d' % i]) for i in range(n)]
    weights = [torch.from_numpy(g[prefix + 'net_layers.%d.weight' % i]) for i in range(L)] + \
              [torch.from_numpy(g[prefix + 'final_layer.weight'])]
    biases = [torch.from_numpy(g[prefix + 'net_layers.%d.bias' % i]) for i in range(L)] + \
             [torch.from_numpy(g[prefix + 'final_layer.bias'])]
    frev = torch.from_numpy(g[prefix + 'filter.filter_rev'])
    return coeffs, weights, biases, frev


FWD = ['fwd_cfg1_c16g16h32l2.npz', 'fwd_c4g15h16l3.npz', 'fwd_c6g17h32l4.npz', 'fwd_c2g32h64l4.npz']


def test_pywt_constants_and_filters(golden_dir):
    with open(os.path.join(golden_dir, 'pywt_db2.json')) as f:
        pw = json.load(f)
    assert pw['filter_bank'] == [R.DB2_DEC_LO, R.DB2_DEC_HI, R.DB2_REC_LO, R.DB2_REC_HI]
    for n, lv in pw['dwt_max_level_flen4'].items():
        assert R.dwt_max_level(int(n), 4) == lv, n
    g = load(golden_dir, 'db2_filters.npz')
    ffwd, frev = R.build_filters(3)
    assert np.array_equal(ffwd.numpy(), g['filter_fwd'])
    assert np.array_equal(frev.numpy(), g['filter_rev'])


def test_levels_table(golden_dir):
    with open(os.path.join(golden_dir, 'levels_table.json')) as f:
        table = json.load(f)
    ffwd, _ = R.build_filters(3)
    for key, row in table.items():
        G = int(key.split('_')[0])
        nl = 3 if key.endswith('levels3') else None
        if G > 64:          # shapes only: run on one channel of zeros (cheap)
            pass
        coeffs, shapes = R.encode_volume(torch.zeros(1, G, G, G), ffwd, num_levels=nl)
        assert len(shapes) == row['num_levels']
        assert np.asarray(shapes).tolist() == row['shape_array']
        assert [list(c.shape) for c in coeffs] == row['coeff_shapes']


@pytest.mark.parametrize('G', [15, 16, 17])
def test_dwt_roundtrip(golden_dir, G):
    g = load(golden_dir, 'dwt_roundtrip_%d.npz' % G)
    ffwd, frev = R.build_filters(3)
    grid = torch.from_numpy(g['input'])
    coeffs, shapes = R.encode_volume(grid, ffwd)
    assert np.array_equal(np.asarray(shapes), g['shape_array'])
    for i, c in enumerate(coeffs):
        assert np.array_equal(c.numpy(), g['coeff%d' % i]), i
    dec = R.decode_volume(coeffs, shapes, frev)
    assert np.array_equal(dec.numpy(), g['decoded'])
    # explicit-index restatement agrees to fp32 rounding, and the transform inverts inside the crop
    dec64 = E.decode_volume([c.numpy() for c in coeffs], shapes, frev.numpy())
    assert np.abs(dec64 - g['decoded']).max() < 5e-6
    assert np.abs(dec64 - g['input']).max() < 5e-6
    lvl = E.dwt_level(g['input'], ffwd.numpy())
    ref1, _ = R.wavelet_encode(grid.unsqueeze(0), ffwd)
    assert np.abs(lvl - ref1[0].numpy()).max() < 5e-6


def test_dwt_noncubic(golden_dir):
    g = load(golden_dir, 'dwt_noncubic.npz')
    ffwd, frev = R.build_filters(3)
    coeffs, shape = R.wavelet_encode(torch.from_numpy(g['input']), ffwd)
    assert np.array_equal(coeffs.numpy(), g['coeffs'])
    assert np.array_equal(shape, g['shape'])
    dec = R.wavelet_decode(coeffs, shape, frev)
    assert np.array_equal(dec.numpy(), g['decoded'])
    lvl = E.dwt_level(g['input'][0], ffwd.numpy())
    assert lvl.shape == g['coeffs'][0].shape
    assert np.abs(lvl - g['coeffs'][0]).max() < 5e-6
    d64 = E.idwt_level(g['coeffs'][0][:, 0], g['coeffs'][0][:, 1:], frev.numpy(), g['shape'])
    assert np.abs(d64 - g['decoded'][0]).max() < 5e-6


@pytest.mark.parametrize('name', FWD)
def test_forward_matches_reference(golden_dir, name):
    g = load(golden_dir, name)
    C, G, H, L, nf = [int(v) for v in g['meta']]
    coeffs, weights, biases, frev = model_from_sd(g)
    pos = torch.from_numpy(g['pos'])
    dense = R.decode_volume(coeffs, g['shape_array'], frev)
    assert np.array_equal(dense.numpy(), g['decoded'])
    y, parts = R.forward_from_grid(dense, weights, biases, pos, nf, return_parts=True)
    x0 = torch.cat([pos, parts['emb'], parts['feat']], -1)
    assert np.array_equal(x0.numpy(), g['x0'])
    for i in range(L):
        assert np.array_equal(parts['pre'][i].numpy(), g['pre%d' % i]), i
    assert np.array_equal(y.numpy(), g['y'])
    y2 = R.forward(coeffs, g['shape_array'], frev, weights, biases, pos, nf, training=True)
    assert np.array_equal(y2.numpy(), g['y'])
    # eval branch (intended semantics)
    ev = torch.from_numpy(g['eval_pos'])
    yev = R.forward(coeffs, g['shape_array'], frev, weights, biases, ev, nf, training=False)
    assert yev.shape == (1, 1, 8, 9, 10, 1)
    assert np.array_equal(yev.numpy(), g['eval_y'])
    # explicit fp64 restatement: same function up to fp32 rounding of the reference
    y64 = E.forward_from_grid(E.decode_volume([c.numpy() for c in coeffs], g['shape_array'], frev.numpy()),
                              [w.numpy() for w in weights], [b.numpy() for b in biases], g['pos'], nf)
    scale = np.abs(g['y']).max()
    assert np.abs(y64 - g['y']).max() / scale < 1e-5
    feat64 = E.sample_grid(g['decoded'], g['pos'])
    assert np.abs(feat64 - g['x0'][:, 15:]).max() < 2e-6
    emb64 = E.fourier_embed(g['pos'], nf)
    assert np.abs(emb64 - g['x0'][:, 3:15]).max() < 2e-6


@pytest.mark.parametrize('name', ['fwd_c4g15h16l3.npz', 'fwd_c6g17h32l4.npz'])
def test_backward_matches_reference(golden_dir, name):
    g = load(golden_dir, name)
    C, G, H, L, nf = [int(v) for v in g['meta']]
    coeffs, weights, biases, frev = model_from_sd(g)
    leaves = [t.requires_grad_(True) for t in coeffs + weights + biases]
    pos = torch.from_numpy(g['pos']).requires_grad_(True)
    y = R.forward(coeffs, g['shape_array'], frev, weights, biases, pos, nf, training=True)
    loss = torch.nn.functional.mse_loss(y.squeeze(-1), torch.from_numpy(g['target']))
    assert loss.item() == float(g['loss'])
    loss.backward()
    for i, c in enumerate(coeffs):
        assert np.array_equal(c.grad.numpy(), g['grad.feature_grid.%d' % i]), i
    for i in range(L):
        assert np.array_equal(weights[i].grad.numpy(), g['grad.net_layers.%d.weight' % i])
        assert np.array_equal(biases[i].grad.numpy(), g['grad.net_layers.%d.bias' % i])
    assert np.array_equal(weights[L].grad.numpy(), g['grad.final_layer.weight'])
    assert np.array_equal(biases[L].grad.numpy(), g['grad.final_layer.bias'])
    assert np.array_equal(pos.grad.numpy(), g['grad_pos'])


def test_gt_interpolation(golden_dir):
    g = load(golden_dir, 'gt_interp.npz')
    for tag in ('a', 'b'):
        vol = torch.from_numpy(g['vol_' + tag])
        ds = R.VolumeIndexing(vol.shape)
        for kind in ('lat', 'frac'):
            p = torch.from_numpy(g[kind + '_' + tag])
            out = R.trilinear_f_interpolation(p, vol, ds.min_idx, ds.max_idx, ds.vol_res)
            assert np.array_equal(out.numpy(), g['gt_%s_%s' % (kind, tag)])
        # lattice points degenerate to a gather (exact on the cube, <=2e-6 otherwise)
        gat = E.gt_gather_lattice(g['vol_' + tag], g['lat_' + tag])
        assert np.abs(gat - g['gt_lat_' + tag]).max() < 2e-6
        # IndexDataset.__getitem__ arithmetic
        raw = torch.from_numpy(g['item_raw_' + tag])
        raw2, norm2 = ds.training_positions(raw.to(torch.long))
        assert np.array_equal(raw2.numpy(), g['item_raw_' + tag])
        assert np.array_equal(norm2.numpy(), g['item_norm_' + tag])


def test_tiles_and_stats(golden_dir):
    g = load(golden_dir, 'tiles_70x40x33.npz')
    ds = R.VolumeIndexing(g['volume'].shape)
    assert np.array_equal(ds.scales.numpy(), g['scales'])
    bounds = list(R.tile_iter(ds.vol_res_touple, 32))
    assert len(bounds) == int(g['n_calls']) == 12
    for i, b in enumerate(bounds):
        tp = R.tile_positions(ds, b).unsqueeze(0)           # CPU branch: (1,x,y,z,3)
        assert np.array_equal(tp.numpy(), g['tile%d' % i]), i
    gt = load(golden_dir, 'fwd_c4g15h16l3.npz')
    coeffs, weights, biases, frev = model_from_sd(gt)
    net = lambda t: R.forward(coeffs, gt['shape_array'], frev, weights, biases, t, 2, training=False)
    full = R.field_from_net(ds, net, 32)
    assert np.array_equal(full.numpy(), g['full_vol'])
    stats = R.deviation_statistics(full, torch.from_numpy(g['volume']))
    assert np.allclose(stats, g['stats'], rtol=0, atol=0)


def test_trainstep(golden_dir):
    g = load(golden_dir, 'trainstep_small.npz')
    coeffs, weights, biases, frev = model_from_sd(g, 'before.')
    L = len(weights) - 1
    vol = torch.from_numpy(g['volume'])
    ds = R.VolumeIndexing(vol.shape)
    idx = torch.from_numpy(g['idx'])
    lat = torch.stack([idx // (24 * 24), (idx // 24) % 24, idx % 24], 1)
    raw, norm = ds.training_positions(lat)
    assert np.array_equal(raw.numpy(), g['raw'])
    assert np.array_equal(norm.numpy(), g['norm'])
    params = [t.requires_grad_(True) for t in coeffs] + \
             [p.requires_grad_(True) for pair in zip(weights, biases) for p in pair]
    opt = torch.optim.Adam(params, lr=0.008)
    pred = R.forward(coeffs, g['shape_array'], frev, weights, biases, norm, 2, training=True).squeeze(-1)
    gtv = R.trilinear_f_interpolation(raw, vol, ds.min_idx, ds.max_idx, ds.vol_res)
    assert np.array_equal(gtv.numpy(), g['gt'])
    assert np.array_equal(pred.detach().numpy(), g['pred'])
    loss = torch.nn.MSELoss()(pred, gtv)
    assert loss.item() == float(g['loss'])
    loss.backward()
    opt.step()
    for i, c in enumerate(coeffs):
        assert np.array_equal(c.detach().numpy(), g['after.feature_grid.%d' % i])
    for i in range(L):
        assert np.array_equal(weights[i].detach().numpy(), g['after.net_layers.%d.weight' % i])
        assert np.array_equal(biases[i].detach().numpy(), g['after.net_layers.%d.bias' % i])
    assert np.array_equal(weights[L].detach().numpy(), g['after.final_layer.weight'])


def test_finite_difference_gradient(golden_dir):
    g = load(golden_dir, 'gt_fd_grad.npz')
    p, vol = torch.from_numpy(g['p']), torch.from_numpy(g['vol'])
    mn, mx, rs = (torch.from_numpy(g[k]) for k in ('min_bb', 'max_bb', 'res'))
    assert np.array_equal(R.finite_difference_trilinear_grad(p, vol, mn, mx, rs).numpy(), g['grad'])
    assert np.array_equal(R.finite_difference_trilinear_grad(p, vol, mn, mx, rs, scale=torch.from_numpy(g['scale'])).numpy(),
                          g['grad_scaled'])
