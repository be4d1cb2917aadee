"""Generate tests/golden/*.npz from the REFERENCE's own Python modules (build container only).

    python tools/make_goldens.py

The reference (/root/reference, read-only) is imported, never copied.  Two third-party modules it
imports at module top are not installed for this interpreter (ordinary ModuleNotFoundError):
  * ``pywt``   -- used for 16 constants (db2 filter bank) and ``dwt_max_level``; replaced here by a
                  15-line stand-in fed from tests/golden/pywt_db2.json, which was dumped from a real
                  PyWavelets 1.1.1 by tools/dump_pywt_constants.py.
  * ``pyevtk`` -- only ``imageToVTK`` (VTK file dump, never called here); replaced by a no-op.
Everything else (model, embedding, wavelet filter, interpolation, dataset, tile driver) is the
reference's own code.  The fixtures are DATA (inputs + outputs); no reference source is stored.

The reference's eval-mode forward raises TypeError on torch >= 2 (model/Feature_Grid_Model.py:78,
``x.view(orig_shape[0:-1],1)``); eval fixtures are therefore "train-mode forward on the flattened
positions, reshaped, clamped" = the intended semantics.  The script records
whether the TypeError still reproduces.
"""
import json
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.normpath(os.path.join(HERE, '..', 'tests', 'golden'))

if not os.path.isdir(REF):
    sys.exit('tools/make_goldens.py needs the reference checkout at /root/reference (build container only)')

# ---- stand-ins for the two absent third-party modules -------------------------------------------
with open(os.path.join(GOLD, 'pywt_db2.json')) as f:
    _PYWT = json.load(f)


class _Wavelet:
    def __init__(self, name):
        assert name == 'db2', name
        self.name = name
        self.filter_bank = tuple(list(x) for x in _PYWT['filter_bank'])
        self.dec_len = _PYWT['dec_len']


def _dwt_max_level(data_len, filter_len):
    flen = filter_len.dec_len if isinstance(filter_len, _Wavelet) else int(filter_len)
    assert flen == 4
    return _PYWT['dwt_max_level_flen4'][str(int(data_len))]


pywt_mod = types.ModuleType('pywt')
pywt_mod.Wavelet = _Wavelet
pywt_mod.dwt_max_level = _dwt_max_level
sys.modules['pywt'] = pywt_mod
pyevtk_mod = types.ModuleType('pyevtk')
pyevtk_hl = types.ModuleType('pyevtk.hl')
pyevtk_hl.imageToVTK = lambda *a, **k: None
pyevtk_mod.hl = pyevtk_hl
sys.modules['pyevtk'] = pyevtk_mod
sys.modules['pyevtk.hl'] = pyevtk_hl

sys.path.insert(0, REF)
from model.model_utils import setup_model                       # noqa: E402
from model.Feature_Grid_Model import Feature_Grid_Model         # noqa: E402
from model.Feature_Embedding import FourierEmbedding            # noqa: E402
from wavelet_transform.Torch_Wavelet_Transform import WaveletFilter3d   # noqa: E402
from data.Interpolation import trilinear_f_interpolation        # noqa: E402
from data.IndexDataset import IndexDataset, normalize_volume    # noqa: E402
from visualization.OutputToVTK import field_from_net, calculate_deviation_statistics   # noqa: E402

torch.set_num_threads(4)


def rng_for(seed):
    return np.random.Generator(np.random.PCG64(seed))


def build_model(C, G, H, L, seed, n_freqs=2):
    """Reference setup_model (model/model_utils.py:23-59), then overwrite every parameter from a
    seeded numpy stream so the fixture does not depend on torch's global RNG."""
    model = setup_model(3, H, 1, L, 'fourier', n_freqs, '', 0.1, 0.9, 'db2', C, G, '')
    rng = rng_for(seed)
    grid = torch.from_numpy(rng.random((C, G, G, G), dtype=np.float32))
    feats, shapes = model.encode_volume(grid)
    with torch.no_grad():
        for p, f in zip(model.feature_grid, feats):
            p.copy_(f)
        for lin in list(model.net_layers) + [model.final_layer]:
            bound = 1.0 / np.sqrt(lin.in_features)
            lin.weight.copy_(torch.from_numpy(rng.uniform(-bound, bound, lin.weight.shape).astype(np.float32)))
            lin.bias.copy_(torch.from_numpy(rng.uniform(-bound, bound, lin.bias.shape).astype(np.float32)))
    assert np.array_equal(np.asarray(shapes), model.shape_array)
    return model, grid


def special_positions(rng, n_random, G):
    """Cube corners, face/edge points at exactly +-1, cell-boundary points, random interior."""
    pts = [[sx, sy, sz] for sx in (-1.0, 1.0) for sy in (-1.0, 1.0) for sz in (-1.0, 1.0)]
    for a in range(3):
        for s in (-1.0, 1.0):
            for _ in range(6):
                p = rng.uniform(-1, 1, 3)
                p[a] = s
                pts.append(p.tolist())
    # exactly on cell centres / cell boundaries of the grid: ix = ((p+1)G-1)/2 integer or half-integer
    for k in (0, 1, G // 2, G - 2, G - 1):
        c = (2.0 * k + 1.0) / G - 1.0
        pts.append([c, c, c])
        pts.append([c, -c, 0.123])
        b = 2.0 * k / G - 1.0
        pts.append([b, 0.3, -0.7])
    pts.append([0.0, 0.0, 0.0])
    pts = np.asarray(pts, np.float32)
    rnd = rng.uniform(-1, 1, (n_random, 3)).astype(np.float32)
    return torch.from_numpy(np.concatenate([pts, rnd], 0))


def state_arrays(model):
    return {'sd.' + k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def gen_filters_and_levels():
    filt = WaveletFilter3d('db2')
    np.savez_compressed(os.path.join(GOLD, 'db2_filters.npz'),
                        filter_fwd=filt.filter_fwd.numpy(), filter_rev=filt.filter_rev.numpy())
    table = {}
    emb = FourierEmbedding(2, 3)
    for G in (8, 15, 16, 17, 32, 33, 64, 128):
        m = Feature_Grid_Model(emb, torch.zeros(1, G, G, G), None, filt, hidden_channel=4, num_layer=1)
        table[str(G)] = {'num_levels': int(len(m.shape_array)),
                         'shape_array': np.asarray(m.shape_array).tolist(),
                         'coeff_shapes': [list(p.shape) for p in m.feature_grid]}
    # explicit num_levels (encode_volume(num_levels=), model/Feature_Grid_Model.py:83): BASELINE cfg 5 asks 3 on 128^3
    m = Feature_Grid_Model(emb, torch.zeros(1, 16, 16, 16), None, filt, hidden_channel=4, num_layer=1)
    feats, shapes = m.encode_volume(torch.zeros(1, 128, 128, 128), num_levels=3)
    table['128_levels3'] = {'num_levels': 3, 'shape_array': np.asarray(shapes).tolist(),
                            'coeff_shapes': [list(f.shape) for f in feats]}
    with open(os.path.join(GOLD, 'levels_table.json'), 'w') as f:
        json.dump(table, f, indent=1)


def gen_dwt_roundtrip():
    filt = WaveletFilter3d('db2')
    emb = FourierEmbedding(2, 3)
    for G in (15, 16, 17):
        rng = rng_for(100 + G)
        grid = torch.from_numpy(rng.random((3, G, G, G), dtype=np.float32))
        m = Feature_Grid_Model(emb, grid, None, filt, hidden_channel=4, num_layer=1)
        out = {'input': grid.numpy(), 'shape_array': np.asarray(m.shape_array),
               'decoded': m.decode_volume().detach().numpy()}
        for i, p in enumerate(m.feature_grid):
            out['coeff%d' % i] = p.detach().numpy()
        np.savez_compressed(os.path.join(GOLD, 'dwt_roundtrip_%d.npz' % G), **out)
    # non-cubic single level through the filter directly (odd/even mix; pins the pad-slot quirk)
    rng = rng_for(140)
    data = torch.from_numpy(rng.random((1, 2, 9, 12, 7), dtype=np.float32))
    coeffs, shape = filt.encode(data)
    dec = filt.decode(coeffs, shape)
    np.savez_compressed(os.path.join(GOLD, 'dwt_noncubic.npz'), input=data.numpy(), coeffs=coeffs.numpy(),
                        shape=np.asarray(shape), decoded=dec.numpy())


FWD_CASES = {                     # name: (C, G, H, L, seed, n_random, with_backward)
    'cfg1_c16g16h32l2': (16, 16, 32, 2, 2001, 400, False),
    'c4g15h16l3': (4, 15, 16, 3, 2011, 400, True),
    'c6g17h32l4': (6, 17, 32, 4, 2012, 400, True),
    'c2g32h64l4': (2, 32, 64, 4, 2013, 400, False),
}


def gen_forward_backward():
    eval_typeerror = None
    for name, (C, G, H, L, seed, n_random, with_bwd) in FWD_CASES.items():
        model, grid = build_model(C, G, H, L, seed)
        model.train()
        rng = rng_for(seed + 7)
        pos = special_positions(rng, n_random, G)
        captured = {}
        h0 = model.net_layers[0].register_forward_pre_hook(lambda mod, inp: captured.__setitem__('x0', inp[0].detach().clone()))
        hooks = [h0]
        for i, lin in enumerate(model.net_layers):
            hooks.append(lin.register_forward_hook(
                lambda mod, inp, out, i=i: captured.__setitem__('pre%d' % i, out.detach().clone())))
        pos_req = pos.clone().requires_grad_(True)
        y = model(pos_req)
        for h in hooks:
            h.remove()
        out = dict(state_arrays(model))
        out.update(meta=np.asarray([C, G, H, L, 2]), shape_array=np.asarray(model.shape_array), pos=pos.numpy(),
                   y=y.detach().numpy(), x0=captured['x0'].numpy(),
                   decoded=model.decode_volume().detach().numpy())
        for i in range(L):
            out['pre%d' % i] = captured['pre%d' % i].numpy()
        # eval-shaped case: (1,1,8,9,10,3) tile-like input (intended semantics, see module docstring)
        ev = torch.from_numpy(rng.uniform(-1, 1, (1, 1, 8, 9, 10, 3)).astype(np.float32))
        with torch.no_grad():
            yev = model(ev.reshape(-1, 3)).view(1, 1, 8, 9, 10, 1).clamp(-1, 1)
        out.update(eval_pos=ev.numpy(), eval_y=yev.numpy())
        if eval_typeerror is None:
            model.eval()
            try:
                with torch.no_grad():
                    model(ev)
                eval_typeerror = False
            except TypeError:
                eval_typeerror = True
            model.train()
        if with_bwd:
            target = torch.from_numpy(rng.uniform(-1, 1, (pos.shape[0],)).astype(np.float32))
            loss = torch.nn.functional.mse_loss(y.squeeze(-1), target)
            model.zero_grad()
            loss.backward()
            out.update(target=target.numpy(), loss=np.asarray(loss.item(), np.float64), grad_pos=pos_req.grad.numpy())
            for k, p in model.named_parameters():
                out['grad.' + k] = p.grad.numpy()
        np.savez_compressed(os.path.join(GOLD, 'fwd_%s.npz' % name), **out)
    return eval_typeerror


def gen_gt_interp():
    out = {}
    for tag, shape, seed in (('a', (20, 21, 22), 301), ('b', (31, 31, 31), 302)):
        rng = rng_for(seed)
        vol = torch.from_numpy(rng.uniform(-1, 1, shape).astype(np.float32))
        ds = IndexDataset(vol, 16)
        lat = torch.stack([torch.from_numpy(rng.integers(0, s, 400)) for s in shape], 1).to(torch.float)
        # include the far corner and the origin
        lat[0] = torch.tensor([0., 0., 0.])
        lat[1] = torch.tensor([shape[0] - 1., shape[1] - 1., shape[2] - 1.])
        frac = torch.from_numpy(np.stack([rng.uniform(0, s - 1, 400) for s in shape], 1).astype(np.float32))
        out['vol_' + tag] = vol.numpy()
        out['lat_' + tag] = lat.numpy()
        out['frac_' + tag] = frac.numpy()
        out['gt_lat_' + tag] = trilinear_f_interpolation(lat, vol, ds.min_idx, ds.max_idx, ds.vol_res).numpy()
        out['gt_frac_' + tag] = trilinear_f_interpolation(frac, vol, ds.min_idx, ds.max_idx, ds.vol_res).numpy()
        raw, norm = None, None
        torch.manual_seed(seed)
        raw, norm = ds[0]
        out['item_raw_' + tag] = raw.numpy()
        out['item_norm_' + tag] = norm.numpy()
    np.savez_compressed(os.path.join(GOLD, 'gt_interp.npz'), **out)


def gen_tiles():
    shape = (70, 40, 33)
    rng = rng_for(401)
    vol = torch.from_numpy(rng.uniform(-3, 5, shape).astype(np.float32))
    vol = normalize_volume(vol, vol.min(), vol.max(), -1.0, 1.0)
    ds = IndexDataset(vol, 16)
    model, _ = build_model(4, 15, 16, 3, 2011)
    calls = []

    class Rec(torch.nn.Module):
        def forward(self, t):
            calls.append(t.detach().clone())
            flat = t.reshape(-1, 3)
            model.train()
            return model(flat).view(*t.shape[:-1], 1).clamp(-1, 1)

    full = field_from_net(ds, Rec(), is_cuda=False, tiled_res=32)
    psnr, l1, mse, rmse = calculate_deviation_statistics(full, vol)
    out = {'volume': vol.numpy(), 'full_vol': full.numpy(), 'stats': np.asarray([psnr, l1, mse, rmse], np.float64),
           'n_calls': np.asarray(len(calls)), 'scales': ds.scales.numpy()}
    for i, c in enumerate(calls):
        out['tile%d' % i] = c.numpy()
    np.savez_compressed(os.path.join(GOLD, 'tiles_70x40x33.npz'), **out)


def gen_trainstep():
    """One step of the reference loop body (training/training.py:95-138, :199-201) with the reference's
    own model / dataset / GT sampler: forward -> GT -> MSELoss -> backward -> Adam(lr=0.008)."""
    rng = rng_for(501)
    shape = (24, 24, 24)
    vol = torch.from_numpy(rng.uniform(-2, 2, shape).astype(np.float32))
    vol = normalize_volume(vol, vol.min(), vol.max(), -1.0, 1.0)
    ds = IndexDataset(vol, 16)
    model, _ = build_model(4, 15, 16, 3, 2011)
    model.train()
    before = state_arrays(model)
    idx = torch.from_numpy(rng.integers(0, ds.n_voxels, 512))
    raw = ds.volume_indices[idx]
    norm = ds.scales.unsqueeze(0) * normalize_volume(raw, ds.min_idx.unsqueeze(0), ds.max_idx.unsqueeze(0), -1.0, 1.0)
    opt = torch.optim.Adam(model.parameters(), lr=0.008)
    norm.requires_grad = True
    opt.zero_grad()
    pred = model(norm).squeeze(-1)
    gt = trilinear_f_interpolation(raw, vol, ds.min_idx, ds.max_idx, ds.vol_res)
    loss = torch.nn.MSELoss()(pred, gt)
    loss.backward()
    opt.step()
    out = {'volume': vol.numpy(), 'idx': idx.numpy(), 'raw': raw.numpy(), 'norm': norm.detach().numpy(),
           'pred': pred.detach().numpy(), 'gt': gt.numpy(), 'loss': np.asarray(loss.item(), np.float64),
           'shape_array': np.asarray(model.shape_array)}
    out.update({'before.' + k[3:]: v for k, v in before.items()})
    out.update({'after.' + k[3:]: v for k, v in state_arrays(model).items()})
    np.savez_compressed(os.path.join(GOLD, 'trainstep_small.npz'), **out)


if __name__ == '__main__':
    gen_filters_and_levels()
    gen_dwt_roundtrip()
    err = gen_forward_backward()
    gen_gt_interp()
    gen_tiles()
    gen_trainstep()
    with open(os.path.join(GOLD, 'README.json'), 'w') as f:
        json.dump({'generator': 'tools/make_goldens.py', 'torch': torch.__version__, 'numpy': np.__version__,
                   'reference': 'Bussler/Latent_Feature_Grid_Compression @ 2024_10_08 (/root/reference)',
                   'pywt_constants_from': 'PyWavelets ' + _PYWT['pywt_version'] + ' (tools/dump_pywt_constants.py)',
                   'reference_eval_forward_raises_TypeError': bool(err)}, f, indent=1)
    tot = sum(os.path.getsize(os.path.join(GOLD, f)) for f in os.listdir(GOLD))
    print('goldens written to', GOLD, 'total bytes', tot)
