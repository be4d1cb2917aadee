"""GPU parity tests (through the C-ABI) for the forward side of the hot path: wavelet decode/encode,
fused sample+embed+MLP forward, ground-truth sampler, full-volume drivers.

Checked against (a) the fixtures captured from the reference's own modules (tests/golden) and (b) the
oracle (oracle/ref_torch.py on CPU, oracle/ref_explicit.py in fp64) on seeded synthetic models of the
BASELINE shapes.  Tolerance for the network output (north_star): <= 1e-5 relative fp32, i.e.
max|y - y_ref| / max|y_ref| <= 1e-5 and allclose(rtol=1e-5, atol=1e-6 * max|y_ref|) with a small
multiple for the deepest nets, stated per test."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import ref_torch as R
from oracle import ref_explicit as E

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')


def rel_err(y, ref):
    y = np.asarray(y, np.float64).reshape(-1)
    ref = np.asarray(ref, np.float64).reshape(-1)
    return np.abs(y - ref).max() / max(np.abs(ref).max(), 1e-30)


def build_from_golden(g, dev):
    from latent_feature_grid_compression_amd.model.Feature_Grid_Model import Feature_Grid_Model
    from latent_feature_grid_compression_amd.model.Feature_Embedding import FourierEmbedding
    from latent_feature_grid_compression_amd.wavelet_transform.Torch_Wavelet_Transform import WaveletFilter3d
    C, G, H, L, nf = [int(v) for v in g['meta']]
    m = Feature_Grid_Model(FourierEmbedding(nf, 3), torch.zeros(C, G, G, G), None, WaveletFilter3d('db2'),
                           hidden_channel=H, num_layer=L)
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd.')}
    m.load_state_dict(sd)
    assert np.array_equal(m.shape_array, g['shape_array'])
    return m.to(dev)


def build_synth(C, G, H, L, seed, dev, num_levels=None):
    from latent_feature_grid_compression_amd.model.Feature_Grid_Model import Feature_Grid_Model
    from latent_feature_grid_compression_amd.model.Feature_Embedding import FourierEmbedding
    from latent_feature_grid_compression_amd.wavelet_transform.Torch_Wavelet_Transform import WaveletFilter3d
    sm = R.synth_model(C, G, H, L, seed=seed, num_levels=num_levels)
    m = Feature_Grid_Model(FourierEmbedding(2, 3), torch.zeros(C, G, G, G), None, WaveletFilter3d('db2'),
                           hidden_channel=H, num_layer=L, num_levels=num_levels)
    assert np.array_equal(m.shape_array, sm['shape_array'])
    with torch.no_grad():
        for p, c in zip(m.feature_grid, sm['coeffs']):
            p.copy_(c)
        for lin, w, b in zip(list(m.net_layers) + [m.final_layer], sm['weights'], sm['biases']):
            lin.weight.copy_(w)
            lin.bias.copy_(b)
    return m.to(dev), sm


def test_library_loads_and_reports_version(dev):
    from latent_feature_grid_compression_amd import _lib
    lib = _lib.load()
    assert lib.lfgc_version() == 100
    assert lib.lfgc_error_string(-3).decode().startswith('network shape')


def test_device_trig_accuracy(dev):
    from latent_feature_grid_compression_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(7)
    xs = np.concatenate([rng.uniform(-40, 40, 400000), np.linspace(-1.6, 1.6, 100001), rng.uniform(-3e4, 3e4, 100000),
                         np.asarray([0.0, -0.0, 1e-30, 3.14159274, 1.57079637, 32768.0, -32768.0])]).astype(np.float32)
    x = torch.from_numpy(xs).to(dev)
    s, c, k = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    rc = lib.lfgc_debug_trig_f32(x.data_ptr(), x.numel(), s.data_ptr(), c.data_ptr(), k.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    x64 = xs.astype(np.float64)
    assert np.abs(s.cpu().numpy() - np.sin(x64)).max() < 2.5e-7
    assert np.abs(c.cpu().numpy() - np.cos(x64)).max() < 2.5e-7
    snake = 0.5 * x64 + np.sin(x64) ** 2
    assert (np.abs(k.cpu().numpy() - snake) / np.maximum(1.0, np.abs(snake))).max() < 4e-7
    # wide path: huge / non-finite arguments behave like libm (finite in [-1,1]; nan for inf/nan)
    big = torch.tensor([1e6, -3.3e7, 1e9, float('inf'), float('nan'), 5e4], dtype=torch.float32, device=dev)
    s, c, k = torch.empty_like(big), torch.empty_like(big), torch.empty_like(big)
    assert lib.lfgc_debug_trig_f32(big.data_ptr(), big.numel(), s.data_ptr(), c.data_ptr(), k.data_ptr(), None) == 0
    sb = s.cpu().numpy()
    b64 = big.cpu().numpy().astype(np.float64)
    assert np.abs(sb[[0, 1, 2, 5]] - np.sin(b64[[0, 1, 2, 5]])).max() < 3e-7
    assert np.isnan(sb[3]) and np.isnan(sb[4])


@pytest.mark.parametrize('G', [15, 16, 17])
def test_wavelet_levels_match_reference(dev, G):
    from latent_feature_grid_compression_amd import ops
    g = np.load(os.path.join(GOLD, 'dwt_roundtrip_%d.npz' % G))
    fl = np.load(os.path.join(GOLD, 'db2_filters.npz'))
    ffwd, frev = torch.from_numpy(fl['filter_fwd']).to(dev), torch.from_numpy(fl['filter_rev']).to(dev)
    coeffs = [torch.from_numpy(g['coeff%d' % i]).to(dev) for i in range(len(g['shape_array']) + 1)]
    dec = ops.decode_levels(coeffs, g['shape_array'], frev, channel_last=False)
    assert dec.shape == g['decoded'].shape
    assert rel_err(dec.cpu().numpy(), g['decoded']) < 1e-6
    dec_cl = ops.decode_levels(coeffs, g['shape_array'], frev, channel_last=True)
    C = g['input'].shape[0]
    # the last level of the channel-last decode is a kernel of its own (different summation order inside the stencil)
    assert dec_cl.shape[-1] == 8 and rel_err(dec_cl[..., :C].permute(3, 0, 1, 2).cpu().numpy(), g['decoded']) < 1e-6
    assert float(dec_cl[..., C:].abs().max()) == 0.0
    # forward DWT (init path), level by level like encode_volume
    data = torch.from_numpy(g['input']).to(dev)
    for lvl in range(len(g['shape_array']), 0, -1):
        out = ops.dwt_level(data, ffwd)
        # stencil contracted axis by axis (lfgc.h `taps`): agrees with the reference's dense 4^3 filter to fp32 rounding;
        # bound relative to the band's largest coefficient (coarse coefficients grow ~2.8x per level)
        assert np.abs(out[:, 1:].cpu().numpy() - g['coeff%d' % lvl]).max() < 1e-6 * np.abs(out.cpu().numpy()).max()
        data = out[:, 0].contiguous()
    assert rel_err(data.cpu().numpy(), g['coeff0']) < 1e-6


def test_wavelet_noncubic_and_adjoint(dev):
    from latent_feature_grid_compression_amd import ops
    g = np.load(os.path.join(GOLD, 'dwt_noncubic.npz'))
    fl = np.load(os.path.join(GOLD, 'db2_filters.npz'))
    ffwd, frev = torch.from_numpy(fl['filter_fwd']).to(dev), torch.from_numpy(fl['filter_rev']).to(dev)
    out = ops.dwt_level(torch.from_numpy(g['input'][0]).to(dev), ffwd)
    assert rel_err(out.cpu().numpy(), g['coeffs'][0]) < 1e-6
    co = torch.from_numpy(g['coeffs'][0]).to(dev)
    dec = ops.idwt_level(co[:, 0].contiguous(), co[:, 1:].contiguous(), frev, g['shape'])
    assert rel_err(dec.cpu().numpy(), g['decoded'][0]) < 1e-6
    # adjoint vs autograd of the oracle's conv_transpose3d + crop, both layouts
    rng = np.random.default_rng(5)
    d_out = torch.from_numpy(rng.standard_normal(g['decoded'][0].shape).astype(np.float32))
    data = torch.from_numpy(g['coeffs']).clone().requires_grad_(True)
    ref = R.wavelet_decode(data, g['shape'], torch.from_numpy(fl['filter_rev']))
    ref.backward(d_out.unsqueeze(0))
    d = g['coeffs'].shape[-3:]
    d_lll, d_hf = ops.idwt_level_bwd(d_out.to(dev), frev, d)
    assert np.abs(d_lll.cpu().numpy() - data.grad[0, :, 0].numpy()).max() < 5e-6
    assert np.abs(d_hf.cpu().numpy() - data.grad[0, :, 1:].numpy()).max() < 5e-6
    # layout conversion round trip (channel-first <-> channel-last with zero pad channels), ragged sizes
    x = torch.from_numpy(rng.standard_normal((22, 5, 7, 13)).astype(np.float32)).to(dev)
    xl = ops.to_channel_last(x)
    assert xl.shape == (5, 7, 13, 24) and torch.equal(xl[..., :22].permute(3, 0, 1, 2), x)
    assert float(xl[..., 22:].abs().max()) == 0.0
    assert torch.equal(ops.to_channel_first(xl, 22), x)


CL_CASES = [   # (C, d, t): channel counts around every stride, ragged planes, every crop the level formula can produce
    (1, (3, 4, 5), (7, 9, 11)), (5, (6, 7, 9), (13, 15, 19)), (8, (9, 9, 9), (16, 16, 16)), (13, (5, 12, 7), (10, 24, 14)),
    (16, (17, 17, 17), (32, 32, 32)), (22, (4, 35, 6), (9, 69, 13)), (24, (10, 11, 12), (19, 21, 23)),
    (30, (8, 9, 40), (17, 18, 81)), (32, (17, 18, 16), (33, 34, 31)), (32, (33, 33, 33), (64, 64, 64)),
    (3, (1, 1, 1), (2, 3, 4)), (32, (2, 70, 3), (5, 140, 6)),
    (40, (5, 6, 7), (11, 13, 15)),      # C > 32: the wrappers fall back to level + layout conversion
    # large planes: 16-wave synthesis workgroups (32 channels), 64-cell adjoint tiles (16- and 8-channel groups)
    (32, (2, 64, 64), (5, 129, 128)), (16, (2, 64, 64), (4, 128, 128)), (8, (3, 60, 60), (6, 119, 120)),
    (32, (65, 65, 65), (128, 128, 128)),   # cfg-5 last level: streaming stores (grid > 48 MiB)
]


@pytest.mark.parametrize('C,d,t', CL_CASES)
def test_last_level_channel_last_kernels(dev, C, d, t):
    """The fused last level (synthesis writing / adjoint reading the channel-last grid) against the channel-first level
    kernels + the layout conversion, which the tests above pin to the reference fixtures and the oracle's
    conv_transpose3d.  Different summation order inside the 64-tap stencil: agreement to fp32 rounding."""
    from latent_feature_grid_compression_amd import ops
    rng = np.random.default_rng(C * 1000 + d[0])
    _, frev = R.build_filters(3)
    frev = frev.to(dev)
    lll = torch.from_numpy(rng.standard_normal((C,) + d).astype(np.float32)).to(dev)
    hf = torch.from_numpy(rng.standard_normal((C, 7) + d).astype(np.float32)).to(dev)
    want = ops.to_channel_last(ops.idwt_level(lll, hf, frev, t))
    got = ops.idwt_level_cl(lll, hf, frev, t)
    assert got.shape == want.shape
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) < 2e-6 * scale
    cs = got.shape[-1]
    if cs > C:
        assert float(got[..., C:].abs().max()) == 0.0
    g_cl = torch.from_numpy(rng.standard_normal(tuple(t) + (cs,)).astype(np.float32)).to(dev)
    w_l, w_h = ops.idwt_level_bwd(ops.to_channel_first(g_cl, C), frev, d)
    g_l, g_h = ops.idwt_level_cl_bwd(g_cl, C, frev, d)
    s = float(max(w_l.abs().max(), w_h.abs().max()))
    assert float((g_l - w_l).abs().max()) < 2e-6 * s
    assert float((g_h - w_h).abs().max()) < 2e-6 * s


def test_last_level_channel_last_against_oracle(dev):
    """... and once directly against the oracle (conv_transpose3d + crop, autograd for the adjoint)."""
    from latent_feature_grid_compression_amd import ops
    rng = np.random.default_rng(11)
    C, d, t = 6, (7, 9, 8), (15, 19, 16)
    _, frev = R.build_filters(3)
    data = torch.from_numpy(rng.standard_normal((1, C, 8) + d).astype(np.float32)).requires_grad_(True)
    ref = R.wavelet_decode(data, t, frev)
    w = torch.from_numpy(rng.standard_normal(ref.shape).astype(np.float32))
    ref.backward(w)
    fd = frev.to(dev)
    got = ops.idwt_level_cl(data[0, :, 0].detach().to(dev).contiguous(), data[0, :, 1:].detach().to(dev).contiguous(), fd, t)
    assert rel_err(got[..., :C].permute(3, 0, 1, 2).cpu().numpy(), ref[0].detach().numpy()) < 1e-6
    g_cl = torch.zeros(tuple(t) + (8,), dtype=torch.float32)
    g_cl[..., :C] = w[0].permute(1, 2, 3, 0)
    g_cl[..., C:] = 7.0              # pad channels of the incoming gradient are ignored
    d_lll, d_hf = ops.idwt_level_cl_bwd(g_cl.to(dev), C, fd, d)
    assert rel_err(d_lll.cpu().numpy(), data.grad[0, :, 0].numpy()) < 2e-6
    assert rel_err(d_hf.cpu().numpy(), data.grad[0, :, 1:].numpy()) < 2e-6


def test_reduced_precision_entry_points_are_the_f16_mode(dev):
    """lfgc_forward_bf16 / lfgc_backward_bf16 (the names SURVEY 8(b) lists) = the f32 entries with LFGC_PRECISION_F16."""
    from latent_feature_grid_compression_amd import ops, _lib
    g = np.load(os.path.join(GOLD, 'fwd_c6g17h32l4.npz'))
    m = build_from_golden(g, dev).train()
    lib = _lib.load()
    desc = m._descriptor()
    grid = m._decoded_channel_last().detach()
    packed = m._packed()
    pos = torch.from_numpy(np.random.default_rng(2).uniform(-1, 1, (777, 3)).astype(np.float32)).to(dev)
    want, stash_w = ops.forward_raw(desc, grid, packed, pos=pos, want_stash=True, precision='f16', range_fallback=False)
    ps, n = ops._positions_struct(pos)
    D, H, W, _ = grid.shape
    out = torch.empty(n, device=dev)
    stash = torch.empty_like(stash_w)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.lfgc_forward_bf16(ctypes.byref(desc), ctypes.byref(ps), grid.data_ptr(), D, H, W, packed.data_ptr(), 0,
                                 out.data_ptr(), stash.data_ptr(), None, st) == 0
    assert torch.equal(out, want) and torch.equal(stash, stash_w)
    word = torch.full((1,), 7, dtype=torch.int32, device=dev)          # with a status word: cleared, stays clear, same result
    assert lib.lfgc_forward_bf16(ctypes.byref(desc), ctypes.byref(ps), grid.data_ptr(), D, H, W, packed.data_ptr(), 0,
                                 out.data_ptr(), stash.data_ptr(), word.data_ptr(), st) == 0
    assert int(word.item()) == 0 and torch.equal(out, want) and torch.equal(stash, stash_w)
    weights, biases = m._mlp_params()
    d_y = torch.from_numpy(np.random.default_rng(3).standard_normal(n).astype(np.float32)).to(dev)
    w_grid, w_w, w_b, _ = ops.backward_raw(desc, grid, packed, pos, stash_w, d_y, weights, biases, False, precision='f16')
    d_grid = torch.zeros_like(grid)
    d_w = [torch.empty_like(w) for w in weights]
    d_b = [torch.empty_like(b) for b in biases]
    nbytes = int(lib.lfgc_backward_workspace_bytes(ctypes.byref(desc), n))
    ws = torch.empty(max(nbytes, 16) // 4, device=dev)
    wp, _k1 = _lib.ptr_array([t.data_ptr() for t in d_w])
    bp, _k2 = _lib.ptr_array([t.data_ptr() for t in d_b])
    assert lib.lfgc_backward_bf16(ctypes.byref(desc), ctypes.byref(ps), grid.data_ptr(), D, H, W, packed.data_ptr(),
                                  stash_w.data_ptr(), d_y.data_ptr(), d_grid.data_ptr(), wp, bp, None, ws.data_ptr(), nbytes, st) == 0
    for a_, b_ in zip(d_w + d_b, list(w_w) + list(w_b)):
        assert torch.equal(a_, b_)                               # weight gradients: deterministic slab reduction
    assert float((d_grid - w_grid).abs().max()) <= 1e-5 * float(w_grid.abs().max())     # float atomics: order-dependent last bits


def test_dense_stencil_for_non_separable_filter(dev):
    """A filter buffer that is NOT an outer product of a 1-D bank (never produced by the reference, but a legal buffer
    value) takes the dense 4^3-tap kernels: forward, adjoint and the analysis form, against the oracle's convolutions."""
    from latent_feature_grid_compression_amd import ops
    rng = np.random.default_rng(77)
    filt = torch.from_numpy(rng.standard_normal((8, 1, 4, 4, 4)).astype(np.float32) * 0.3)
    assert ops.filter_taps(filt) is None
    _, frev = R.build_filters(3)
    assert ops.filter_taps(frev) is not None
    C, d, t = 5, (6, 7, 9), (13, 15, 19)
    data = torch.from_numpy(rng.standard_normal((1, C, 8) + d).astype(np.float32)).requires_grad_(True)
    ref = R.wavelet_decode(data, t, filt)
    w = torch.from_numpy(rng.standard_normal(ref.shape).astype(np.float32))
    ref.backward(w)
    fd = filt.to(dev)
    out = ops.idwt_level(data[0, :, 0].detach().to(dev).contiguous(), data[0, :, 1:].detach().to(dev).contiguous(), fd, t)
    assert rel_err(out.cpu().numpy(), ref[0].detach().numpy()) < 1e-6
    d_lll, d_hf = ops.idwt_level_bwd(w[0].to(dev), fd, d)
    assert rel_err(d_lll.cpu().numpy(), data.grad[0, :, 0].numpy()) < 2e-6
    assert rel_err(d_hf.cpu().numpy(), data.grad[0, :, 1:].numpy()) < 2e-6
    x = torch.from_numpy(rng.standard_normal((1, C, 11, 12, 13)).astype(np.float32))
    enc, _ = R.wavelet_encode(x, filt)
    got = ops.dwt_level(x[0].to(dev), fd)
    assert rel_err(got.cpu().numpy(), enc[0].numpy()) < 1e-6


FWD = ['fwd_cfg1_c16g16h32l2.npz', 'fwd_c4g15h16l3.npz', 'fwd_c6g17h32l4.npz', 'fwd_c2g32h64l4.npz']


PRECISIONS = ['f16x2', 'fp32']      # layer-GEMM arithmetic of the fused kernel (include/lfgc.h); both must meet 1e-5


@pytest.mark.parametrize('precision', PRECISIONS)
@pytest.mark.parametrize('name', FWD)
def test_forward_matches_reference_fixture(dev, name, precision):
    g = np.load(os.path.join(GOLD, name))
    m = build_from_golden(g, dev)
    m.precision = precision
    m.train()
    with torch.no_grad():
        dec = m.decode_volume()
        assert np.abs(dec.cpu().numpy() - g['decoded']).max() < 3e-6
        y = m(torch.from_numpy(g['pos']).to(dev))
    assert y.shape == (g['pos'].shape[0], 1)
    yr = g['y']
    assert rel_err(y.cpu().numpy(), yr) <= 1e-5
    assert np.allclose(y.cpu().numpy(), yr, rtol=1e-5, atol=2e-6 * np.abs(yr).max())
    # eval branch: tile-shaped input, clamped, same leading dims
    m.eval()
    with torch.no_grad():
        yev = m(torch.from_numpy(g['eval_pos']).to(dev))
    assert yev.shape == (1, 1, 8, 9, 10, 1)
    assert rel_err(yev.cpu().numpy(), g['eval_y']) <= 1e-5
    assert float(yev.max()) <= 1.0 and float(yev.min()) >= -1.0


def _decode_stash(stash, n, KS0, L, MT):
    """Private stash layout -> x0 in packed k order (N, 2*KS0) and pre-activations (L, N, 32*MT)."""
    per_tile = 64 * (KS0 + L * 16 * MT)
    tiles = stash.numel() // per_tile
    st = stash.view(tiles, KS0 + L * 16 * MT, 64).cpu().numpy()
    x0 = np.zeros((tiles * 32, 2 * KS0), np.float32)
    pre = np.zeros((L, tiles * 32, 32 * MT), np.float32)
    for lane in range(64):
        j, hh = lane & 31, lane >> 5
        for s in range(KS0):
            x0[j::32, 8 * (s // 4) + 4 * hh + (s % 4)] = st[:, s, lane]
        for l in range(L):
            for mr in range(16 * MT):
                m_, r = mr // 16, mr % 16
                row = 32 * m_ + (r & 3) + 8 * (r >> 2) + 4 * hh
                pre[l, j::32, row] = st[:, KS0 + l * 16 * MT + mr, lane]
    return x0[:n], pre[:, :n]


@pytest.mark.parametrize('precision', PRECISIONS)
def test_stash_holds_reference_preactivations(dev, precision):
    """The values saved for backward are the reference's layer-0 input and pre-activations."""
    from latent_feature_grid_compression_amd import ops
    g = np.load(os.path.join(GOLD, 'fwd_cfg1_c16g16h32l2.npz'))
    m = build_from_golden(g, dev)
    pos = torch.from_numpy(g['pos']).to(dev)
    with torch.no_grad():
        grid = m._decoded_channel_last()
        y, stash = ops.forward_raw(m._descriptor(), grid, m._packed(), pos=pos, want_stash=True, precision=precision)
    C, G, H, L, nf = [int(v) for v in g['meta']]
    x0, pre = _decode_stash(stash, pos.shape[0], (16 + 16) // 2, L, 1)
    # packed column order of layer 0 -> reference column order [p, emb, feat]
    ref_x0 = g['x0']
    for cl in range(32):
        s, hh = (cl >> 3) * 4 + (cl & 3), (cl >> 2) & 1
        if s < 8:
            src = 15 + hh * 8 + s
        else:
            e = hh * 8 + (s - 8)
            src = e if e < 15 else None
        if src is None:
            assert np.all(x0[:, cl] == 0)
        else:
            assert np.abs(x0[:, cl] - ref_x0[:, src]).max() < 2e-6, (cl, src)
    for l in range(L):
        assert np.abs(pre[l][:, :H] - g['pre%d' % l]).max() < 1e-5 * max(1.0, np.abs(g['pre%d' % l]).max())


@pytest.mark.parametrize('C,G,H,L,n,tol', [
    (16, 32, 64, 4, 50000, 1e-5),      # BASELINE cfg 2
    (32, 64, 128, 4, 40000, 1e-5),     # BASELINE cfg 3/4
    (22, 17, 32, 4, 10001, 1e-5),      # reference experiment configs: odd grid, 22 channels, ragged N
    (3, 15, 100, 1, 77, 1e-5),         # padding everywhere: C 3->8, H 100->128, one layer, N < one tile
    (32, 20, 128, 8, 4096, 2e-5),      # deepest supported net
])
@pytest.mark.parametrize('precision', PRECISIONS)
def test_forward_matches_oracle_on_synthetic_models(dev, C, G, H, L, n, tol, precision):
    m, sm = build_synth(C, G, H, L, seed=4000 + C + G + H, dev=dev)
    m.precision = precision
    rng = np.random.default_rng(C * 1000 + G)
    pos = torch.from_numpy(rng.uniform(-1, 1, (n, 3)).astype(np.float32))
    pos[:8] = torch.tensor([[sx, sy, sz] for sx in (-1., 1.) for sy in (-1., 1.) for sz in (-1., 1.)])[:min(8, n)]
    m.train()
    with torch.no_grad():
        y = m(pos.to(dev)).cpu().numpy()
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    dense = R.decode_volume(sm['coeffs'], sm['shape_array'], sm['filter_rev'])
    yref = R.forward_from_grid(dense, sm['weights'], sm['biases'], pos, 2).numpy()
    assert rel_err(y, yref) <= tol
    # fp64 truth: the HIP path must not be further from it than ~the fp32 CPU path is
    sub = slice(0, min(n, 4000))
    y64 = E.forward_from_grid(E.decode_volume([c.numpy() for c in sm['coeffs']], sm['shape_array'], sm['filter_rev'].numpy()),
                              [w.numpy() for w in sm['weights']], [b.numpy() for b in sm['biases']], pos[sub].numpy(), 2)
    e_hip, e_cpu = rel_err(y[sub], y64), rel_err(yref[sub], y64)
    assert e_hip <= max(3 * e_cpu, 3e-6), (e_hip, e_cpu)
    print('precision %s C%d G%d H%d L%d: HIP vs fp64 %.2e, torch-CPU vs fp64 %.2e, HIP vs torch-CPU %.2e'
          % (precision, C, G, H, L, e_hip, e_cpu, rel_err(y, yref)))


@pytest.mark.parametrize('num_levels', [3, None])
def test_cfg5_grid_three_level_code_and_large_lattice(dev, num_levels):
    """BASELINE cfg 5 shape: 128^3 x 32-channel grid, MLP 4x128, coded with 3 wavelet levels as BASELINE names it
    (encode_volume(num_levels=3)) and with the reference's own default depth (model/Feature_Grid_Model.py:85:
    pywt.dwt_max_level(128, 4) = 5 levels).  Random positions vs the oracle, and a slab of the 1024^3 lattice
    (x = 512..543, generated in-kernel) vs the same voxels pushed through the explicit-position path."""
    from latent_feature_grid_compression_amd.data.IndexDataset import IndexDataset
    from latent_feature_grid_compression_amd.visualization import OutputToVTK as V
    m, sm = build_synth(32, 128, 128, 4, seed=4242, dev=dev, num_levels=num_levels)
    if num_levels == 3:
        assert np.asarray(m.shape_array).tolist() == [[34, 34, 34], [65, 65, 65], [128, 128, 128]]
    else:
        assert len(m.feature_grid) == 6 and np.asarray(m.shape_array)[-1].tolist() == [128, 128, 128]
    rng = np.random.default_rng(5)
    pos = torch.from_numpy(rng.uniform(-1, 1, (20000, 3)).astype(np.float32))
    m.train()
    with torch.no_grad():
        y = m(pos.to(dev)).cpu().numpy()
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    dense = R.decode_volume(sm['coeffs'], sm['shape_array'], sm['filter_rev'])
    yref = R.forward_from_grid(dense, sm['weights'], sm['biases'], pos, 2).numpy()
    assert rel_err(y, yref) <= 1e-5
    ds = IndexDataset((1024, 1024, 1024), 16, build_index_table=False)
    m.eval()
    slab = V.field_from_net_fused(ds, m, 512, 544)[:, 100:164, 900:1024]        # (32, 64, 124) voxels of the slab
    b = (512, 544, 96, 128, 896, 928)
    tp = R.tile_positions(R.VolumeIndexing((1024, 1024, 1024)), b)              # one reference tile inside it
    with torch.no_grad():
        yt = m(tp.unsqueeze(0).to(dev)).squeeze(0).squeeze(-1)
    full_tile = V.field_from_net_fused(ds, m, 512, 544)[:, 96:128, 896:928]
    assert rel_err(full_tile.cpu().numpy(), yt.cpu().numpy()) <= 2e-6
    assert slab.shape == (32, 64, 124) and bool(torch.isfinite(slab).all())


def test_gt_interpolation_bit_exact(dev):
    from latent_feature_grid_compression_amd.data.Interpolation import trilinear_f_interpolation
    g = np.load(os.path.join(GOLD, 'gt_interp.npz'))
    for tag in ('a', 'b'):
        vol = torch.from_numpy(g['vol_' + tag])
        ds = R.VolumeIndexing(vol.shape)
        for kind in ('lat', 'frac'):
            p = torch.from_numpy(g[kind + '_' + tag]).to(dev)
            out = trilinear_f_interpolation(p, vol.to(dev), ds.min_idx, ds.max_idx, ds.vol_res)
            assert np.array_equal(out.cpu().numpy(), g['gt_%s_%s' % (kind, tag)]), (tag, kind)


def test_full_volume_drivers_match_reference_tiles(dev):
    from latent_feature_grid_compression_amd.data.IndexDataset import IndexDataset
    from latent_feature_grid_compression_amd.visualization import OutputToVTK as V
    g = np.load(os.path.join(GOLD, 'tiles_70x40x33.npz'))
    gm = np.load(os.path.join(GOLD, 'fwd_c4g15h16l3.npz'))
    m = build_from_golden(gm, dev)
    vol = torch.from_numpy(g['volume'])
    ds = IndexDataset(vol, 16)
    m.eval()
    full_loop = V.field_from_net(ds, m, is_cuda=True, tiled_res=32)          # reference-style: 12 net(tile) calls
    assert rel_err(full_loop.numpy(), g['full_vol']) <= 1e-5
    full_fused = V.field_from_net_fused(ds, m)                               # one launch, lattice built in-kernel
    assert full_fused.shape == vol.shape
    assert rel_err(full_fused.cpu().numpy(), g['full_vol']) <= 1e-5
    # slabs: any tile-aligned x range reproduces the same voxels
    part = V.field_from_net_fused(ds, m, 32, 70)
    assert torch.equal(part, full_fused[32:70])
    stats = V.calculate_deviation_statistics(full_fused, vol.to(dev), verbose=False)
    assert abs(stats[0] - g['stats'][0]) < 1e-3                              # PSNR within 0.01 dB (north_star)
    assert np.allclose(stats[1:], g['stats'][1:], rtol=2e-5)
    single = V.reconstruct_volume_sharded(ds, m)                             # world size 1 path
    assert torch.equal(single, full_fused)


@pytest.mark.parametrize('C,G,H,L,res,slab', [
    (32, 16, 128, 4, (9, 40, 64), None),        # rows of two whole 32-voxel runs, streamed net, 4-wave workgroups
    (16, 8, 64, 4, (5, 7, 95), (1, 4)),         # ragged rows (95 = 2 x 32 + 31), an x-slab that starts and ends mid-volume
    (8, 8, 32, 2, (6, 5, 31), None),            # a row shorter than one run (31 live lanes), resident net
    (24, 12, 100, 3, (4, 6, 70), (0, 3)),       # 24 channels: 6 lanes per column cell, 60 of 64 lanes carry a cell
    (16, 32, 64, 4, (3, 4, 150), None),         # BASELINE cfg 2 ratio (150 voxels on 32 cells): a 9-cell column
    (8, 64, 32, 2, (3, 3, 40), None),           # grid finer than the lattice: column too long -> the per-sample path
])
def test_lattice_zrun_column_sampler(dev, C, G, H, L, res, slab):
    """Lattice mode takes z-run tiles + the column sampler (csrc/lfgc_forward.h, LfgcColumnSampler) where the column is
    short and the per-sample 8-corner gather otherwise (`LFGC_NO_ZRUN=1` forces the latter).  Both against the oracle on
    the reference's own tile positions (<= 1e-5, north_star), and against each other: the two evaluate the same
    trilinear sum in a different order, so they agree to fp32 rounding -- and, where the column path is taken, are not
    bit-identical, which is how the test knows it ran."""
    from latent_feature_grid_compression_amd import ops
    from latent_feature_grid_compression_amd.data.IndexDataset import IndexDataset
    m, sm = build_synth(C, G, H, L, seed=900 + C + G, dev=dev)
    m.eval()
    ds = IndexDataset(res, 16, build_index_table=False)
    xb, xe = slab if slab else (0, res[0])
    outs = {}
    for mode in ('zrun', 'gather'):
        if mode == 'gather':
            os.environ['LFGC_NO_ZRUN'] = '1'
        try:
            with torch.no_grad():
                y, _ = ops.forward_raw(m._descriptor(), m._decoded_channel_last(), m._packed(), lattice=(res, xb, xe, 32), clamp=True)
            outs[mode] = y.view(xe - xb, res[1], res[2]).cpu()
        finally:
            os.environ.pop('LFGC_NO_ZRUN', None)
    rds = R.VolumeIndexing(res)
    dense = R.decode_volume(sm['coeffs'], sm['shape_array'], sm['filter_rev'])
    ref = torch.empty(res)
    for b in R.tile_iter(rds.vol_res_touple, 32):
        yt = R.forward_from_grid(dense, sm['weights'], sm['biases'], R.tile_positions(rds, b).reshape(-1, 3), 2).clamp(-1, 1)
        ref[b[0]:b[1], b[2]:b[3], b[4]:b[5]] = yt.reshape(b[1] - b[0], b[3] - b[2], b[5] - b[4])
    ref = ref[xb:xe].numpy()
    for mode, y in outs.items():
        assert rel_err(y.numpy(), ref) <= 1e-5, mode
    assert rel_err(outs['zrun'].numpy(), outs['gather'].numpy()) <= 3e-6
    short_column = int(31.0 * G / (res[2] - 1) + 1e-3) + 3 <= 12
    assert torch.equal(outs['zrun'], outs['gather']) != short_column, 'which sampler ran is not what the column length says'


def test_two_tiles_per_wave_kernel_is_the_same_function(dev):
    """`LFGC_FWD_X2=1` (csrc/lfgc_forward16x2.h: two 32-sample tiles per wave, one wave per SIMD -- the variant VERDICT r2
    asked for, kept opt-in because it measured 7 % slower) computes every sample with the same instruction sequence as the
    default z-run kernel: bit-identical volumes at the headline net on a lattice big enough for the host to select it."""
    from latent_feature_grid_compression_amd import ops
    m, sm = build_synth(32, 64, 128, 4, seed=4321, dev=dev)
    m.eval()
    res = (72, 64, 256)                      # 72 * 64 rows x 8 runs = 36 864 tiles >= 8 per CU
    outs = {}
    for mode in ('x1', 'x2'):
        if mode == 'x2':
            os.environ['LFGC_FWD_X2'] = '1'
        try:
            with torch.no_grad():
                y, _ = ops.forward_raw(m._descriptor(), m._decoded_channel_last(), m._packed(), lattice=(res, 3, 70, 32), clamp=True)
            outs[mode] = y.cpu()
        finally:
            os.environ.pop('LFGC_FWD_X2', None)
    assert bool(torch.isfinite(outs['x1']).all())
    assert torch.equal(outs['x1'], outs['x2'])
    # and the function is the right one: a few tiles against the oracle
    rds = R.VolumeIndexing(res)
    dense = R.decode_volume(sm['coeffs'], sm['shape_array'], sm['filter_rev'])
    vol = outs['x2'].view(67, 64, 256)
    for b in [(32, 64, 0, 32, 96, 128), (64, 72, 32, 64, 224, 256)]:
        pos = R.tile_positions(rds, b).reshape(-1, 3)
        ref = R.forward_from_grid(dense, sm['weights'], sm['biases'], pos, 2).clamp(-1, 1).reshape(b[1] - b[0], b[3] - b[2], b[5] - b[4])
        x0, x1 = max(b[0], 3), min(b[1], 70)
        assert rel_err(vol[x0 - 3:x1 - 3, b[2]:b[3], b[4]:b[5]].numpy(), ref[x0 - b[0]:x1 - b[0]].numpy()) <= 1e-5


def test_eval_cache_tracks_parameter_updates(dev):
    m, sm = build_synth(8, 16, 32, 2, seed=91, dev=dev)
    m.eval()
    pos = torch.rand(1, 4, 4, 4, 3, device=dev) * 2 - 1
    with torch.no_grad():
        y1 = m(pos)
        y1b = m(pos)
        assert torch.equal(y1, y1b)
        m.feature_grid[0].mul_(1.5)                  # in-place update must invalidate the decoded-grid cache
        m.final_layer.bias.add_(0.25)                # ... and the packed-parameter cache
        y2 = m(pos)
    assert not torch.equal(y1, y2)
    coeffs = [p.detach().cpu() for p in m.feature_grid]
    w = [l.weight.detach().cpu() for l in list(m.net_layers) + [m.final_layer]]
    b = [l.bias.detach().cpu() for l in list(m.net_layers) + [m.final_layer]]
    yref = R.forward(coeffs, sm['shape_array'], sm['filter_rev'], w, b, pos.cpu(), 2, training=False)
    assert rel_err(y2.cpu().numpy(), yref.numpy()) <= 1e-5


def test_rejects_cpu_tensors_and_bad_shapes(dev):
    from latent_feature_grid_compression_amd import _lib, ops
    m, _ = build_synth(8, 16, 32, 2, seed=92, dev=dev)
    with pytest.raises(_lib.LfgcError):
        m(torch.zeros(4, 3))                         # CPU input: no silent fallback
    with pytest.raises(ValueError):
        m.train()(torch.zeros(4, 2, device=dev))
    with pytest.raises(_lib.LfgcError):
        ops.make_desc(64, 32, 2, 2)                  # outside the compiled kernel set
    y = m.train()(torch.zeros(0, 3, device=dev))     # empty batch
    assert y.shape == (0, 1)


def _two_rank_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from latent_feature_grid_compression_amd.data.IndexDataset import IndexDataset
    from latent_feature_grid_compression_amd.visualization import OutputToVTK as V
    dev = torch.device('cuda:0')                      # both ranks share the one GPU of the test box
    m, _ = build_synth(8, 16, 32, 2, seed=123, dev=dev)
    m.eval()
    ds = IndexDataset((100, 40, 36), 16, build_index_table=False)
    res = {'full': V.field_from_net_fused(ds, m).cpu()}

    # (1) reconstruct_volume_sharded itself on both ranks, slabs from the fused HIP kernel, collectives on host tensors
    #     (gloo's native form): partition, pieces, in-place receives and both gather modes of the function under test.
    def hip_slab_to_host(b, e, view):
        view.copy_(V.field_from_net_fused(ds, m, b, e).cpu())
    for mode in ('all', 'root'):
        v = V.reconstruct_volume_sharded(ds, m, chunks=2, slab_fn=hip_slab_to_host, device=torch.device('cpu'), gather=mode)
        res['host_' + mode] = v if v is not None else None
    # (2) the same function with DEVICE tensors handed to the collective (what RCCL does on a multi-GPU node).  gloo may or
    #     may not take device tensors in this build: recorded, never silently replaced by another code path.
    try:
        res['device_all'] = V.reconstruct_volume_sharded(ds, m, chunks=2).cpu()
        res['device_error'] = ''
    except RuntimeError as exc:
        res['device_all'] = None
        res['device_error'] = '%s: %s' % (type(exc).__name__, str(exc)[:300])
    torch.save(res, os.path.join(out_dir, 'r%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.fixture(scope='module')
def two_rank_results(tmp_path_factory):
    import socket
    import torch.multiprocessing as mp
    if not torch.cuda.is_available():
        pytest.skip('needs the GPU')
    out = tmp_path_factory.mktemp('two_ranks')
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_two_rank_worker, args=(2, port, str(out)), nprocs=2, join=True)
    return [torch.load(os.path.join(str(out), 'r%d.pt' % r), weights_only=True) for r in range(2)]


def test_sharded_reconstruction_two_ranks_share_the_gpu(dev, two_rank_results):
    """reconstruct_volume_sharded with two processes (gloo rendezvous) on the one GPU: each rank's x-slab comes from the
    fused HIP kernel, the pieces travel as host tensors; the volume the FUNCTION UNDER TEST assembles equals the
    single-launch volume bit for bit -- on every rank for gather='all', on the root alone for gather='root'."""
    for r, d in enumerate(two_rank_results):
        assert torch.equal(d['host_all'], d['full']), r
        if r == 0:
            assert torch.equal(d['host_root'], d['full'])
        else:
            assert d['host_root'] is None


def test_sharded_reconstruction_two_ranks_device_collective(dev, two_rank_results):
    """The same with device tensors handed straight to the collective.  Skips -- explicitly, with the backend's own
    message -- where this gloo build takes no device tensors; RCCL (which does) cannot put two ranks on one GPU, so on a
    one-GPU box its path is covered at world size 1 (test_cfg4_volume_255_cubed_sharded_edge_tiles)."""
    errs = [d['device_error'] for d in two_rank_results]
    if any(errs):
        pytest.skip('device-tensor collective unavailable with gloo here: ' + next(e for e in errs if e))
    for r, d in enumerate(two_rank_results):
        assert torch.equal(d['device_all'], d['full']), r


def test_headline_volume_invariants(dev):
    """BASELINE full size (256^3 lattice, 64^3 x 32-channel grid, MLP 4x128), size-independent properties:
    (1) the volume assembled from the 8 x-slabs an 8-GPU run would compute is bit-identical to the single launch
    (the kernel's result does not depend on which workgroup/batch a sample lands in);
    (2) random 32^3 tiles of it equal the explicit-position forward of the reference-style tile tensors;
    (3) every voxel is finite and inside the eval clamp."""
    from latent_feature_grid_compression_amd.data.IndexDataset import IndexDataset
    from latent_feature_grid_compression_amd.visualization import OutputToVTK as V
    m, _ = build_synth(32, 64, 128, 4, seed=2003, dev=dev)
    m.eval()
    ds = IndexDataset((256, 256, 256), 16, build_index_table=False)
    full = V.field_from_net_fused(ds, m)
    assert full.shape == (256, 256, 256) and bool(torch.isfinite(full).all())
    assert float(full.max()) <= 1.0 and float(full.min()) >= -1.0
    parts = V.slab_partition(256, 8, 32)
    stitched = torch.cat([V.field_from_net_fused(ds, m, b, e) for b, e in parts], 0)
    assert torch.equal(stitched, full)
    vi = R.VolumeIndexing((256, 256, 256))
    rng = np.random.default_rng(11)
    for _ in range(3):
        t = [int(v) * 32 for v in rng.integers(0, 8, 3)]
        b = (t[0], t[0] + 32, t[1], t[1] + 32, t[2], t[2] + 32)
        tp = R.tile_positions(vi, b)
        with torch.no_grad():
            yt = m(tp.unsqueeze(0).to(dev)).squeeze(0).squeeze(-1)
        assert rel_err(full[b[0]:b[1], b[2]:b[3], b[4]:b[5]].cpu().numpy(), yt.cpu().numpy()) <= 2e-6


def test_sharded_driver_over_rccl_single_rank(dev):
    """The chunked all-gather path of reconstruct_volume_sharded through RCCL itself (backend 'nccl'), one rank: the
    multi-rank logic is covered by the gloo tests on CPU, this checks the collective calls on device tensors."""
    import torch.distributed as dist
    from latent_feature_grid_compression_amd.data.IndexDataset import IndexDataset
    from latent_feature_grid_compression_amd.visualization import OutputToVTK as V
    if dist.is_initialized():
        pytest.skip('a process group already exists')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    try:
        m, _ = build_synth(8, 16, 32, 2, seed=77, dev=dev)
        m.eval()
        ds = IndexDataset((70, 40, 33), 16, build_index_table=False)
        ref = V.field_from_net_fused(ds, m)
        got = V.reconstruct_volume_sharded(ds, m, chunks=3, always_gather=True)
        assert torch.equal(got, ref)
        # pieces finer than a tile plane (what 8 ranks on 256^3 use): slabs may start anywhere
        got = V.reconstruct_volume_sharded(IndexDataset((31, 40, 33), 16, build_index_table=False), m, chunks=4,
                                           always_gather=True)
        assert torch.equal(got, V.field_from_net_fused(IndexDataset((31, 40, 33), 16, build_index_table=False), m))
        parts = [V.field_from_net_fused(ds, m, b, e) for b, e in ((0, 5), (5, 37), (37, 64), (64, 70))]
        assert torch.equal(torch.cat(parts, 0), ref)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('name', FWD)
def test_reduced_precision_f16_mode(dev, name):
    """LFGC_PRECISION_F16 (opt-in; the "bf16 compute" of BASELINE config 3 in f16 form): one f16 product per GEMM block,
    fp32 accumulate.  Not a parity mode: bound = a few f16 ulps (2^-11) amplified through the layers, stated here as
    3e-3 of the output range; the default and exact builds keep the 1e-5 bound (tests above)."""
    g = np.load(os.path.join(GOLD, name))
    m = build_from_golden(g, dev)
    m.precision = 'f16'
    m.train()
    with torch.no_grad():
        y = m(torch.from_numpy(g['pos']).to(dev))
    err = rel_err(y.cpu().numpy(), g['y'])
    assert 1e-6 < err <= 3e-3, err          # really the reduced arithmetic, and within its bound


def _oracle_forward(sm, pos):
    dense = R.decode_volume(sm['coeffs'], sm['shape_array'], sm['filter_rev'])
    return R.forward_from_grid(dense, sm['weights'], sm['biases'], pos, 2).numpy()


@pytest.mark.parametrize('where', ['layer1_weights', 'layer3_weights', 'grid_features'])
def test_default_precision_is_range_safe(dev, where):
    """The reference computes in fp32 and stays finite for any finite parameters (model/Feature_Grid_Model.py:12-13,
    :72-78).  The default f16-split arithmetic has a range (|pre-activation| <~ 800, |feature| < 65504): a pass that
    leaves it must come back reference-equivalent all the same (exact-fp32 redo predicated on the kernel's status word),
    never as NaN or as a finite wrong number."""
    from latent_feature_grid_compression_amd import ops
    C, G, H, L = 16, 16, 64, 4
    m, sm = build_synth(C, G, H, L, seed=77, dev=dev)
    with torch.no_grad():
        if where == 'layer1_weights':          # layer-2 pre-activations ~1e5 (>= 7e4: VERDICT r1 item 1)
            m.net_layers[1].weight.mul_(3.0e4)
            sm['weights'][1] = sm['weights'][1] * 3.0e4
        elif where == 'layer3_weights':        # the LAST hidden layer: no f16 conversion downstream, explicit screen
            m.net_layers[3].weight.mul_(2.0e4)
            sm['weights'][3] = sm['weights'][3] * 2.0e4
        else:                                  # grid features beyond the f16 range (U(0,1) * 4e5), the network function
            for p in m.feature_grid:             # itself kept well conditioned: layer 0 scales them back
                p.mul_(4.0e5)
            sm['coeffs'] = [c * 4.0e5 for c in sm['coeffs']]
            m.net_layers[0].weight[:, 15:].mul_(2.5e-6)
            sm['weights'][0] = sm['weights'][0].clone()
            sm['weights'][0][:, 15:] *= 2.5e-6
    rng = np.random.default_rng(5)
    pos = torch.from_numpy(rng.uniform(-1, 1, (3000, 3)).astype(np.float32))
    yref = _oracle_forward(sm, pos)
    assert np.isfinite(yref).all()
    m.train()
    with torch.no_grad():
        y = m(pos.to(dev)).cpu().numpy()                                   # default precision, default (safe) path
        m.precision = 'fp32'
        y32 = m(pos.to(dev)).cpu().numpy()
    assert np.isfinite(y).all()
    assert rel_err(y32, yref) <= 1e-5
    assert rel_err(y, yref) <= 1e-5
    # what the fast kernel alone does with such a pass: it flags it, and returns NaN -- not a finite wrong value --
    # for the samples it could not do
    m.precision = 'f16x2'
    with torch.no_grad():
        y_raw, _, status = ops.forward_raw(m._descriptor(), m._decoded_channel_last(), m._packed(), pos=pos.to(dev),
                                           range_fallback=False, return_status=True)
        y_ok, _, st_ok = ops.forward_raw(m._descriptor(), m._decoded_channel_last(), m._packed(), pos=pos.to(dev),
                                         return_status=True)
    assert status is None                                   # no status word asked for: nothing redone
    y_raw = y_raw.cpu().numpy()
    bad = ~np.isfinite(y_raw)
    assert bad.any()
    good = ~bad
    if good.any():
        assert np.abs(y_raw[good] - yref.reshape(-1)[good]).max() <= 1e-5 * np.abs(yref).max()
    assert int(st_ok.item()) == 1 and rel_err(y_ok.cpu().numpy(), yref) <= 1e-5
    # ... and a well-behaved model leaves the status word clear
    m2, sm2 = build_synth(C, G, H, L, seed=78, dev=dev)
    with torch.no_grad():
        _, _, st2 = ops.forward_raw(m2._descriptor(), m2._decoded_channel_last(), m2._packed(), pos=pos.to(dev), return_status=True)
    assert int(st2.item()) == 0


def test_reduced_precision_entry_is_range_safe(dev):
    """lfgc_forward_bf16 with a status word: a pass that leaves the f16 range (layer-2 pre-activations ~1e5 -- finite in
    fp32 and in a true bfloat16 path) comes back finite and reference-equivalent through the predicated exact redo, and
    the word reports it; without the word the same pass returns NaN for the samples it could not do."""
    from latent_feature_grid_compression_amd import ops, _lib
    C, G, H, L = 16, 16, 64, 4
    m, sm = build_synth(C, G, H, L, seed=77, dev=dev)
    with torch.no_grad():
        m.net_layers[1].weight.mul_(3.0e4)
        sm['weights'][1] = sm['weights'][1] * 3.0e4
    pos = torch.from_numpy(np.random.default_rng(5).uniform(-1, 1, (3000, 3)).astype(np.float32))
    yref = _oracle_forward(sm, pos)
    assert np.isfinite(yref).all()
    lib = _lib.load()
    desc, grid, packed = m._descriptor(), m._decoded_channel_last().detach(), m._packed()
    pd = pos.to(dev)
    ps, n = ops._positions_struct(pd)
    D, Hh, W, _ = grid.shape
    st = torch.cuda.current_stream().cuda_stream
    out = torch.empty(n, device=dev)
    word = torch.zeros(1, dtype=torch.int32, device=dev)
    assert lib.lfgc_forward_bf16(ctypes.byref(desc), ctypes.byref(ps), grid.data_ptr(), D, Hh, W, packed.data_ptr(), 0,
                                 out.data_ptr(), None, word.data_ptr(), st) == 0
    assert int(word.item()) == 1
    y = out.cpu().numpy()
    assert np.isfinite(y).all() and rel_err(y, yref) <= 1e-5              # the redo is the exact build
    assert lib.lfgc_forward_bf16(ctypes.byref(desc), ctypes.byref(ps), grid.data_ptr(), D, Hh, W, packed.data_ptr(), 0,
                                 out.data_ptr(), None, None, st) == 0
    assert not np.isfinite(out.cpu().numpy()).all()                       # no word: flagged as NaN, never a finite wrong value


def test_range_fallback_also_rewrites_the_training_stash(dev):
    """A training forward whose fast pass overflowed must hand the backward kernels the exact build's stash: gradients of
    a diverged model equal the oracle's."""
    C, G, H, L = 8, 8, 32, 3
    m, sm = build_synth(C, G, H, L, seed=91, dev=dev)
    with torch.no_grad():
        m.net_layers[1].weight.mul_(4.0e3)
        sm['weights'][1] = sm['weights'][1] * 4.0e3
    rng = np.random.default_rng(6)
    pos = torch.from_numpy(rng.uniform(-1, 1, (512, 3)).astype(np.float32))
    m.train()
    y = m(pos.to(dev))
    y.square().mean().backward()
    leaves = [t.clone().requires_grad_(True) for t in sm['coeffs'] + sm['weights'] + sm['biases']]
    nc = len(sm['coeffs'])
    yr = R.forward(leaves[:nc], sm['shape_array'], sm['filter_rev'], leaves[nc:nc + L + 1], leaves[nc + L + 1:], pos, 2)
    yr.square().mean().backward()
    assert rel_err(y.detach().cpu().numpy(), yr.detach().numpy()) <= 1e-5
    ours = [p.grad for p in m.feature_grid] + [l.weight.grad for l in list(m.net_layers) + [m.final_layer]]
    refs = [t.grad for t in leaves[:nc + L + 1]]
    for g, r in zip(ours, refs):
        assert torch.isfinite(g).all()
        # |a| ~ 1e3 here: snake'(a) = 0.5 + sin 2a amplifies the fp32 rounding of a (1e-4 absolute) -- two fp32 evaluation
        # orders of such a model agree to ~1e-4 of the gradient's scale at best; a wrong (overflowed) stash would be off by O(1)
        assert (g.cpu() - r).abs().max() <= 1e-3 * r.abs().max()


def test_non_current_device_is_respected(dev):
    """A model on cuda:1 called while cuda:0 is the current device must launch on cuda:1 (ops._on_device)."""
    if torch.cuda.device_count() < 2:
        pytest.skip('one visible device')
    d1 = torch.device('cuda:1')
    m, sm = build_synth(16, 16, 32, 2, seed=11, dev=d1)
    pos = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, (1000, 3)).astype(np.float32))
    assert torch.cuda.current_device() == 0
    with torch.no_grad():
        y = m.train()(pos.to(d1)).cpu().numpy()
    assert rel_err(y, _oracle_forward(sm, pos)) <= 1e-5
    from latent_feature_grid_compression_amd import _lib
    with pytest.raises(_lib.LfgcError, match='different devices'):
        m(pos.to(dev))


def test_cfg4_volume_255_cubed_sharded_edge_tiles(dev):
    """BASELINE config 4 itself: the 255^3 volume (7 x 32 + 31 per axis: ragged last tile on every axis) with the cfg 3/4
    model, through the sharded driver's collective path on one rank (RCCL, world 1, always_gather), both assembly modes.
    The three 31-wide edge tiles (one per axis) and the 31^3 corner tile are checked against the oracle's per-tile forward
    at 1e-5; the whole volume against the package's per-tile driver."""
    import torch.distributed as dist
    from latent_feature_grid_compression_amd.data.IndexDataset import IndexDataset
    from latent_feature_grid_compression_amd.visualization import OutputToVTK as V
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29617')
    m, sm = build_synth(32, 64, 128, 4, seed=4242, dev=dev)
    m.eval()
    res = (255, 255, 255)
    ds = IndexDataset(res, 16, build_index_table=False)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    try:
        tm = {}
        vol = V.reconstruct_volume_sharded(ds, m, chunks=4, always_gather=True, gather='root', timings=tm)
        vol_all = V.reconstruct_volume_sharded(ds, m, chunks=3, always_gather=True, gather='all')
    finally:
        dist.destroy_process_group()
    assert tuple(vol.shape) == res and tm['world_size'] == 1 and tm['gather'] == 'root' and tm['compute_ms'] > 0
    assert torch.equal(vol, vol_all)
    dense = R.decode_volume(sm['coeffs'], sm['shape_array'], sm['filter_rev'])
    ro = R.VolumeIndexing(res)
    scale = None
    for box in [(224, 255, 0, 32, 0, 32), (0, 32, 224, 255, 32, 64), (96, 128, 64, 96, 224, 255), (224, 255, 224, 255, 224, 255),
                (0, 32, 0, 32, 0, 32)]:
        pos = R.tile_positions(ro, box).reshape(-1, 3)
        yr = R.forward_from_grid(dense, sm['weights'], sm['biases'], pos, 2).clamp(-1, 1).reshape(box[1] - box[0], box[3] - box[2], box[5] - box[4])
        mine = vol[box[0]:box[1], box[2]:box[3], box[4]:box[5]].cpu()
        scale = float(yr.abs().max()) if scale is None else max(scale, float(yr.abs().max()))
        assert float((mine - yr).abs().max()) <= 1e-5 * scale, box
    # size-independent property at the full size: the one-launch-per-slab path equals the per-tile call contract
    sub = V.field_from_net(IndexDataset((31, 40, 33), 16, build_index_table=False), m, is_cuda=True)
    one = V.field_from_net_fused(IndexDataset((31, 40, 33), 16, build_index_table=False), m).cpu()
    assert float((sub - one).abs().max()) <= 2e-6
