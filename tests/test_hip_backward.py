"""GPU parity tests (through the C-ABI) for the backward side of the hot path: gradients into the
wavelet coefficients, the MLP parameters and the input positions, and one complete train step.

Reference = fixtures captured from the reference's own autograd (tests/golden) and the oracle's autograd on
seeded synthetic models.  Gradients are sums over thousands of samples accumulated in a different order
(MFMA k-order, float atomics), so they are compared relative to the largest gradient entry of each tensor:
max|g - g_ref| / max|g_ref| <= 2e-5 (fp32), stated per test."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_torch as R
from test_hip_forward import build_from_golden, build_synth, rel_err, GOLD, dev  # noqa: F401

pytestmark = pytest.mark.gpu


def _grads_vs(model, ref_named, tol, what=''):
    worst = 0.0
    for name, p in model.named_parameters():
        g = p.grad
        assert g is not None, name
        ref = ref_named[name]
        assert tuple(g.shape) == tuple(ref.shape), name
        e = rel_err(g.cpu().numpy(), ref)
        worst = max(worst, e)
        assert e <= tol, '%s %s: rel err %.3e' % (what, name, e)
    return worst


@pytest.mark.parametrize('precision', ['f16x2', 'fp32'])
@pytest.mark.parametrize('name', ['fwd_c4g15h16l3.npz', 'fwd_c6g17h32l4.npz'])
def test_backward_matches_reference_fixture(dev, name, precision):
    g = np.load(os.path.join(GOLD, name))
    m = build_from_golden(g, dev).train()
    m.precision = precision
    pos = torch.from_numpy(g['pos']).to(dev).requires_grad_(True)      # training.py:99 sets requires_grad on positions
    y = m(pos)
    loss = torch.nn.functional.mse_loss(y.squeeze(-1), torch.from_numpy(g['target']).to(dev))
    assert abs(loss.item() - float(g['loss'])) <= 1e-5 * abs(float(g['loss']))
    loss.backward()
    ref = {k[5:]: g[k] for k in g.files if k.startswith('grad.')}
    _grads_vs(m, ref, 2e-5, name)
    assert rel_err(pos.grad.cpu().numpy(), g['grad_pos']) <= 2e-5


@pytest.mark.parametrize('C,G,H,L,n', [
    (16, 32, 64, 4, 16384),      # cfg 2 shape, test_vol train batch (turbulence_basic.txt: 1024 x 16)
    (32, 64, 128, 4, 32768),     # cfg 3: mhd train step, 2048 x 16 samples
    (22, 17, 32, 4, 5000),       # reference experiment config shape, ragged N
    (3, 15, 100, 1, 77),         # padding everywhere, single hidden layer, N < one tile
    (5, 4, 4, 2, 300),           # NAS lower bounds: 4^3 grid (no wavelet level at all), hidden 4
    (24, 7, 20, 3, 1000),        # one wavelet level, odd grid
])
@pytest.mark.parametrize('precision', ['f16x2', 'fp32', 'f16'])
def test_backward_matches_oracle_on_synthetic_models(dev, C, G, H, L, n, precision):
    """Gradients of every parameter and of the positions against the oracle's autograd, every arithmetic build at every
    shape: the two full-precision builds at 2e-5 of each tensor's largest entry, the opt-in reduced build ('f16': one f16
    product per block, the form BASELINE config 3's "bf16 train step" takes here -- DESIGN.md section 7) at its stated
    bounds: loss 1e-2, gradients 3e-2."""
    reduced = precision == 'f16'
    if reduced and (C, G, H, L) not in ((16, 32, 64, 4), (32, 64, 128, 4)):
        pytest.skip('reduced build: exercised at the BASELINE train-step shapes')
    tol_loss, tol_grad = (1e-2, 3e-2) if reduced else (1e-5, 2e-5)
    m, sm = build_synth(C, G, H, L, seed=5000 + C + G + H, dev=dev)
    m.train()
    m.precision = precision
    rng = np.random.default_rng(C * 77 + G)
    # training samples are voxel-lattice points (data/IndexDataset.py:90-96): use a 255^3 lattice
    ds = R.VolumeIndexing((255, 255, 255))
    idx = torch.from_numpy(rng.integers(0, 255, (n, 3)))
    _, pos = ds.training_positions(idx)
    target = torch.from_numpy(rng.uniform(-1, 1, (n,)).astype(np.float32))
    pos_d = pos.to(dev).requires_grad_(True)
    y = m(pos_d)
    loss = torch.nn.functional.mse_loss(y.squeeze(-1), target.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    # oracle autograd
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    coeffs = [c.clone().requires_grad_(True) for c in sm['coeffs']]
    ws = [w.clone().requires_grad_(True) for w in sm['weights']]
    bs = [b.clone().requires_grad_(True) for b in sm['biases']]
    pos_r = pos.clone().requires_grad_(True)
    yr = R.forward(coeffs, sm['shape_array'], sm['filter_rev'], ws, bs, pos_r, 2, training=True)
    lr = torch.nn.functional.mse_loss(yr.squeeze(-1), target)
    lr.backward()
    assert abs(loss.item() - lr.item()) <= tol_loss * abs(lr.item())
    ref = {}
    for i, c in enumerate(coeffs):
        ref['feature_grid.%d' % i] = c.grad.numpy()
    for i in range(L):
        ref['net_layers.%d.weight' % i] = ws[i].grad.numpy()
        ref['net_layers.%d.bias' % i] = bs[i].grad.numpy()
    ref['final_layer.weight'] = ws[L].grad.numpy()
    ref['final_layer.bias'] = bs[L].grad.numpy()
    worst = _grads_vs(m, ref, tol_grad, '%s C%d G%d H%d L%d' % (precision, C, G, H, L))
    assert rel_err(pos_d.grad.cpu().numpy(), pos_r.grad.numpy()) <= tol_grad
    if reduced:
        assert worst > 1e-5                   # really the reduced arithmetic


def test_backward_without_input_grad_and_repeatability(dev):
    """positions that do not require grad skip the d_pos work; parameter grads are unchanged.  Weight/bias
    gradients are bitwise repeatable (slab reduction); grid gradients use float atomics (order-dependent)."""
    m, _ = build_synth(16, 16, 32, 2, seed=77, dev=dev)
    m.train()
    pos = (torch.rand(3000, 3, device=dev) * 2 - 1)
    outs = []
    for req in (True, False, False):
        m.zero_grad()
        p = pos.clone().requires_grad_(req)
        m(p).square().mean().backward()
        outs.append({k: v.grad.clone() for k, v in m.named_parameters()})
    for k in outs[0]:
        if k.startswith('feature_grid'):
            assert rel_err(outs[1][k].cpu().numpy(), outs[0][k].cpu().numpy()) <= 1e-5
        else:
            assert torch.equal(outs[0][k], outs[1][k]) and torch.equal(outs[1][k], outs[2][k]), k


def test_train_step_matches_reference(dev):
    """training/training.py:95-138 with the reference's Adam(lr=0.008): loss, gradients and updated parameters."""
    from latent_feature_grid_compression_amd.data.Interpolation import trilinear_f_interpolation
    from latent_feature_grid_compression_amd.data.IndexDataset import IndexDataset
    g = np.load(os.path.join(GOLD, 'trainstep_small.npz'))
    gm = np.load(os.path.join(GOLD, 'fwd_c4g15h16l3.npz'))
    m = build_from_golden(gm, dev).train()          # same seeded model as the fixture's "before" state
    for k, p in m.state_dict().items():
        assert np.array_equal(p.cpu().numpy(), g['before.' + k]), k
    vol = torch.from_numpy(g['volume']).to(dev)
    ds = IndexDataset(vol.cpu(), 16)
    raw = torch.from_numpy(g['raw']).to(dev)
    _, norm = ds.positions_for(raw)
    assert np.array_equal(norm.cpu().numpy(), g['norm'])
    opt = torch.optim.Adam(m.parameters(), lr=0.008)
    norm.requires_grad = True
    opt.zero_grad()
    pred = m(norm).squeeze(-1)
    gt = trilinear_f_interpolation(raw, vol, ds.min_idx, ds.max_idx, ds.vol_res)
    assert np.array_equal(gt.cpu().numpy(), g['gt'])
    assert rel_err(pred.detach().cpu().numpy(), g['pred']) <= 1e-5
    loss = torch.nn.MSELoss()(pred, gt)
    assert abs(loss.item() - float(g['loss'])) <= 1e-5 * float(g['loss'])
    loss.backward()
    before = {k: p.detach().clone() for k, p in m.named_parameters()}
    opt.step()
    # Reference gradients of the same batch (the fixture stores parameters before / after, not gradients): the oracle's
    # autograd, itself pinned bit-for-bit to the reference's by tests/test_oracle_golden.py.
    nc = len(m.feature_grid)
    L = m.num_layer
    sd = {k[7:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('before.')}
    leaves = ([sd['feature_grid.%d' % i].clone().requires_grad_(True) for i in range(nc)] +
              [sd['net_layers.%d.weight' % i].clone().requires_grad_(True) for i in range(L)] + [sd['final_layer.weight'].clone().requires_grad_(True)] +
              [sd['net_layers.%d.bias' % i].clone().requires_grad_(True) for i in range(L)] + [sd['final_layer.bias'].clone().requires_grad_(True)])
    yr = R.forward(leaves[:nc], g['shape_array'], sd['filter.filter_rev'], leaves[nc:nc + L + 1], leaves[nc + L + 1:],
                   torch.from_numpy(g['norm']), 2, training=True)
    torch.nn.functional.mse_loss(yr.squeeze(-1), torch.from_numpy(g['gt'])).backward()
    names = (['feature_grid.%d' % i for i in range(nc)] + ['net_layers.%d.weight' % i for i in range(L)] + ['final_layer.weight'] +
             ['net_layers.%d.bias' % i for i in range(L)] + ['final_layer.bias'])
    gref = {n: t.grad.numpy() for n, t in zip(names, leaves)}
    lr, eps = 0.008, 1e-8
    checked = 0
    for k, p in m.named_parameters():
        upd = (p.detach() - before[k]).cpu().numpy().astype(np.float64)
        upd_ref = (g['after.' + k].astype(np.float64) - g['before.' + k].astype(np.float64))
        gr = np.abs(gref[k])
        # Adam's first step moves an entry by -lr g / (|g| + eps).  Entries whose reference gradient is neither tiny
        # against its tensor (> 1e-6 of the largest: their sign cannot flip under the 2e-5 gradient tolerance ... if it is
        # also > 1e-4 of it) nor comparable to eps are compared at 1e-3 RELATIVE; the rest -- sign-flip-prone, listed
        # here, not hidden under a blanket tolerance -- only have to stay inside the step's bound lr.
        solid = (gr > 1e-4 * gr.max()) & (gr > 1e3 * eps)
        # (a small batch leaves most coefficient gradients exactly zero: no per-tensor quota, the total is asserted below)
        if solid.any():
            assert np.abs(upd[solid] - upd_ref[solid]).max() <= 1e-3 * np.abs(upd_ref[solid]).min(), k
        assert np.abs(upd).max() <= lr * (1 + 1e-3) and np.abs(upd - upd_ref)[~solid].max(initial=0.0) <= 2 * lr, k
        checked += int(solid.sum())
    assert checked > 1000


def test_device_lattice_sampler_matches_index_dataset(dev):
    """Row f2: the on-device sampler draws lattice points and normalises them exactly like IndexDataset.__getitem__."""
    from latent_feature_grid_compression_amd.data.IndexDataset import DeviceLatticeSampler, IndexDataset
    shape = (20, 21, 22)
    smp = DeviceLatticeSampler(shape, dev)
    raw, norm = smp.sample(5000)
    assert raw.is_cuda and raw.shape == (5000, 3) and norm.shape == (5000, 3)
    r = raw.cpu()
    assert torch.equal(r, r.round()) and float(r.min()) >= 0 and all(float(r[:, a].max()) <= shape[a] - 1 for a in range(3))
    ds = IndexDataset(torch.zeros(shape), 16)
    flat = (r[:, 0] * shape[1] * shape[2] + r[:, 1] * shape[2] + r[:, 2]).long()
    _, ref_norm = ds.positions_for(ds.volume_indices[flat])
    assert torch.equal(norm.cpu(), ref_norm)
    # the fused position kernel at the cfg-3 volume size, against the torch arithmetic on the same flat indices
    big = IndexDataset((255, 255, 255), 16, build_index_table=False)
    flat = torch.randint(0, big.n_voxels, (40000,), device=dev)
    raw_k, norm_k = big.positions_from_flat(flat)
    raw_t, norm_t = big.positions_for(big.lattice_from_flat(flat.cpu()))
    assert torch.equal(raw_k.cpu(), raw_t) and torch.equal(norm_k.cpu(), norm_t)


from philox_ref import philox4x32_10 as _philox4x32_10      # numpy restatement, pinned by known-answer vectors on the CPU


def test_fused_lattice_sampler(dev):
    """Row f2, one-kernel form: the indices the kernel draws are the documented Philox stream (checked against the numpy
    restatement in tests/philox_ref.py, itself pinned by the published known-answer vectors in test_host_logic.py), the
    positions are IndexDataset's for those indices, the draw counter advances on the device -- also across replays of a
    captured launch."""
    from latent_feature_grid_compression_amd import ops
    from latent_feature_grid_compression_amd.data.IndexDataset import DeviceLatticeSampler, IndexDataset
    shape = (255, 255, 255)
    ds = IndexDataset(shape, 16, build_index_table=False)
    nvox = 255 ** 3
    n, seed = 40000, 0x1234_5678_9ABC_DEF1
    state = torch.zeros(2, dtype=torch.int64, device=dev)
    mn, mx, sc = ds.min_idx.tolist(), ds.max_idx.tolist(), ds.scales.tolist()
    for step in range(3):
        raw, norm, flat = ops.lattice_sample(state, n, seed, shape, mn, mx, sc, want_flat=True)
        assert state.cpu().tolist() == [step + 1, 0]
        ctr = np.zeros((n, 4), np.uint32)
        ctr[:, 0] = np.arange(n)
        ctr[:, 2] = step
        r = _philox4x32_10(ctr, np.array([seed & 0xFFFFFFFF, seed >> 32], np.uint32))
        u = [(int(a) << 32) | int(b) for a, b in zip(r[:, 0], r[:, 1])]
        want = np.array([(x * nvox) >> 64 for x in u], np.int64)
        assert np.array_equal(flat.cpu().numpy(), want)
        raw_t, norm_t = ds.positions_for(ds.lattice_from_flat(flat.cpu()))
        assert torch.equal(raw.cpu(), raw_t) and torch.equal(norm.cpu(), norm_t)
    # uniform over the volume: 64 equal slabs of the flat index range, 40 000 draws -> chi^2 with 63 degrees of freedom
    counts = np.bincount((flat.cpu().numpy() * 64 // nvox), minlength=64)
    chi2 = float(((counts - n / 64) ** 2 / (n / 64)).sum())
    assert chi2 < 120.0, chi2                          # P(chi2_63 > 120) ~ 2e-5
    # captured launch: every replay draws the next batch
    smp = DeviceLatticeSampler((20, 21, 22), dev)
    smp.sample_fused(256, seed=7)                      # creates the device state outside the capture
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        smp.sample_fused(256, seed=7)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        raw_g, _ = smp.sample_fused(256, seed=7)
    seen = []
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        seen.append(raw_g.cpu().clone())
    assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2])
    assert smp.ds._sample_state.cpu().tolist() == [5, 0]
    r = seen[2]
    assert torch.equal(r, r.round()) and float(r.min()) >= 0 and all(float(r[:, a].max()) <= (20, 21, 22)[a] - 1 for a in range(3))


def test_unit_grad_seed_skips_nothing_but_the_multiply(dev):
    """loss.backward(mse_unit_grad(dev)) gives the gradients of loss.backward() bit for bit."""
    from latent_feature_grid_compression_amd.data.Interpolation import trilinear_mse_loss, mse_unit_grad
    rng = np.random.default_rng(3)
    vol = torch.from_numpy(rng.uniform(-1, 1, (9, 10, 11)).astype(np.float32)).to(dev)
    p = torch.from_numpy(rng.uniform(0, 8, (500, 3)).astype(np.float32)).to(dev)
    mn, mx, rs = [0.0, 0.0, 0.0], [8.0, 9.0, 10.0], [9.0, 10.0, 11.0]
    grads = []
    for seed in (None, mse_unit_grad(dev), torch.full((), 1.0, device=dev)):
        pred = torch.from_numpy(rng.standard_normal(500).astype(np.float32)).to(dev).requires_grad_(True) if not grads \
            else grads[0][1].detach().clone().requires_grad_(True)
        loss = trilinear_mse_loss(pred, p, vol, mn, mx, rs)
        loss.backward() if seed is None else loss.backward(seed)
        grads.append((pred.grad.clone(), pred))
    assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][0], grads[2][0])
    pred = grads[0][1].detach().clone().requires_grad_(True)
    trilinear_mse_loss(pred, p, vol, mn, mx, rs).backward(torch.full((), 3.0, device=dev))
    assert torch.allclose(pred.grad, 3.0 * grads[0][0], rtol=1e-6, atol=0)


class _ScaleDrop(torch.nn.Module):
    """Minimal pruning layer with the reference's DropoutLayer interface (model/Dropout_Layer.py:4-39): one learnable
    multiplicative factor per coefficient position, broadcast over channels -- what SmallifyDropout.forward does
    (model/Smallify_Dropout.py:54-61) without its sign-variance bookkeeping."""

    def __init__(self, size, p=0.5, threshold=0.9):
        super().__init__()
        self.p, self.threshold = p, threshold
        self.betas = torch.nn.Parameter(torch.full(tuple(size), 0.75))

    def forward(self, x):
        return x * self.betas

    @classmethod
    def create_instance(cls, size, p, threshold):
        return cls(size, p, threshold)


def test_pluggable_drop_layers_get_gradients(dev):
    """drop layers are applied at the reference's points inside decode_volume (model/Feature_Grid_Model.py
    :103,:105) and gradients reach both the coefficients and the drop layer's own parameters."""
    from latent_feature_grid_compression_amd.model.Feature_Grid_Model import Feature_Grid_Model
    from latent_feature_grid_compression_amd.model.Feature_Embedding import FourierEmbedding
    from latent_feature_grid_compression_amd.wavelet_transform.Torch_Wavelet_Transform import WaveletFilter3d
    sm = R.synth_model(8, 16, 32, 2, seed=9)
    m = Feature_Grid_Model(FourierEmbedding(2, 3), sm['grid'], _ScaleDrop((1,)), WaveletFilter3d('db2'),
                           hidden_channel=32, num_layer=2)
    with torch.no_grad():
        for lin, w, b in zip(list(m.net_layers) + [m.final_layer], sm['weights'], sm['biases']):
            lin.weight.copy_(w)
            lin.bias.copy_(b)
    m = m.to(dev).train()
    assert [tuple(d.betas.shape) for d in m.drop] == [(6, 6, 6), (7, 6, 6, 6), (7, 9, 9, 9)]
    pos = torch.rand(2000, 3, device=dev) * 2 - 1
    m(pos).square().mean().backward()
    # oracle: the same model with the masks folded into the coefficients
    coeffs = [(c * 0.75).clone().requires_grad_(True) for c in sm['coeffs']]
    ws = [w.clone().requires_grad_(True) for w in sm['weights']]
    bs = [b.clone().requires_grad_(True) for b in sm['biases']]
    R.forward(coeffs, sm['shape_array'], sm['filter_rev'], ws, bs, pos.cpu(), 2).square().mean().backward()
    for i, (p, d) in enumerate(zip(m.feature_grid, m.drop)):
        gc = coeffs[i].grad                                        # d loss / d (coeff * beta)
        assert rel_err(p.grad.cpu().numpy(), (gc * 0.75).numpy()) <= 2e-5
        gb = (gc * sm['coeffs'][i]).sum(0)                          # beta is broadcast over channels
        assert rel_err(d.betas.grad.cpu().numpy(), gb.numpy()) <= 5e-5


def test_fused_gt_mse_loss_matches_torch(dev):
    """Row f2: ground truth + MSELoss in one pass == trilinear_f_interpolation followed by torch.nn.MSELoss."""
    from latent_feature_grid_compression_amd.data.Interpolation import trilinear_f_interpolation, trilinear_mse_loss
    rng = np.random.default_rng(12)
    vol = torch.from_numpy(rng.uniform(-1, 1, (20, 21, 22)).astype(np.float32)).to(dev)
    n = 5003
    p = torch.from_numpy(np.stack([rng.uniform(0, 19, n), rng.uniform(0, 20, n), rng.uniform(0, 21, n)], 1).astype(np.float32))
    p[:1000] = p[:1000].round()
    p = p.to(dev)
    mn, mx, rs = torch.zeros(3), torch.tensor([19.0, 20.0, 21.0]), torch.tensor([20.0, 21.0, 22.0])
    pred = torch.from_numpy(rng.uniform(-1, 1, n).astype(np.float32)).to(dev).requires_grad_(True)
    gt = trilinear_f_interpolation(p, vol, mn, mx, rs)
    ref = torch.nn.MSELoss()(pred, gt)
    ref.backward()
    ref_grad = pred.grad.clone()
    pred.grad = None
    loss = trilinear_mse_loss(pred, p, vol, mn, mx, rs)
    assert abs(loss.item() - ref.item()) <= 1e-6 * abs(ref.item())
    (3.0 * loss).backward()
    assert rel_err(pred.grad.cpu().numpy(), 3.0 * ref_grad.cpu().numpy()) <= 1e-6


def test_reduced_precision_f16_backward(dev):
    """LFGC_PRECISION_F16 backward: data-gradient chain with single f16 products (per-tile power-of-two scaling keeps the
    gradients inside the f16 range), weight gradients exact fp32.  Bound: 2e-2 of each tensor's largest entry."""
    g = np.load(os.path.join(GOLD, 'fwd_c6g17h32l4.npz'))
    m = build_from_golden(g, dev).train()
    m.precision = 'f16'
    pos = torch.from_numpy(g['pos']).to(dev).requires_grad_(True)
    y = m(pos)
    loss = torch.nn.functional.mse_loss(y.squeeze(-1), torch.from_numpy(g['target']).to(dev))
    assert abs(loss.item() - float(g['loss'])) <= 1e-2 * abs(float(g['loss']))
    loss.backward()
    ref = {k[5:]: g[k] for k in g.files if k.startswith('grad.')}
    worst = _grads_vs(m, ref, 2e-2, 'f16')
    assert worst > 1e-5                       # not accidentally the full-precision path
    assert rel_err(pos.grad.cpu().numpy(), g['grad_pos']) <= 2e-2


@pytest.mark.parametrize('fused', [True, False])
def test_forward_follows_optimizer_updates(dev, fused):
    """torch's fused Adam updates parameters without bumping their version counters: the forward must still see the new
    weights (nothing derived from the parameters may be cached in training mode), and an eval() after training must
    re-decode / re-pack."""
    g = np.load(os.path.join(GOLD, 'fwd_c4g15h16l3.npz'))
    m = build_from_golden(g, dev).train()
    pos = torch.from_numpy(g['pos']).to(dev)
    opt = torch.optim.Adam(m.parameters(), lr=0.05, fused=fused)
    m.eval()
    with torch.no_grad():
        y_eval0 = m(pos.view(1, -1, 1, 1, 3)).reshape(-1).clone()
    m.train()
    for _ in range(3):
        opt.zero_grad()
        m(pos).square().mean().backward()
        opt.step()
    with torch.no_grad():
        y_train = m(pos).reshape(-1)
    fresh = build_from_golden(g, dev)
    fresh.load_state_dict(m.state_dict())
    fresh.train()
    with torch.no_grad():
        y_ref = fresh(pos).reshape(-1)
    assert torch.equal(y_train, y_ref)
    assert (y_train - y_eval0.clamp(-1, 1)).abs().max() > 1e-3            # the steps really changed the function
    m.eval()
    with torch.no_grad():
        y_eval1 = m(pos.view(1, -1, 1, 1, 3)).reshape(-1)
    assert torch.equal(y_eval1, y_ref.clamp(-1, 1))


def test_finite_difference_gradient_of_the_gt_sampler(dev):
    """data/Interpolation.py::finite_difference_trilinear_grad (imported by training/training.py:7): same values as the
    reference's (same fp32 operations around the bit-exact sampler)."""
    from latent_feature_grid_compression_amd.data.Interpolation import finite_difference_trilinear_grad
    g = np.load(os.path.join(GOLD, 'gt_fd_grad.npz'))
    p, vol = torch.from_numpy(g['p']).to(dev), torch.from_numpy(g['vol']).to(dev)
    mn, mx, rs = (torch.from_numpy(g[k]) for k in ('min_bb', 'max_bb', 'res'))
    got = finite_difference_trilinear_grad(p, vol, mn, mx, rs)
    assert np.array_equal(got.cpu().numpy(), g['grad'])
    got = finite_difference_trilinear_grad(p, vol, mn, mx, rs, scale=torch.from_numpy(g['scale']))
    assert np.array_equal(got.cpu().numpy(), g['grad_scaled'])


def test_graph_replayed_train_step_matches_eager_steps(dev):
    """The cfg-3 train step bench.py times is ONE captured HIP graph replayed (bench.capture_train_step).  Replays must
    do what eager steps do: same Philox batches (the draw counter lives on the device and advances per replay), same
    forward / loss / backward / fused Adam.  Run A = 3 eager warm-up steps + capture + K replays (exactly bench.py's
    sequence), run B = 3 + K eager steps of an identically built model; compared after the last step:
      * MLP weights and biases: max|A - B| <= 1e-5 of each tensor's largest entry (their gradients are bitwise
        repeatable given the same inputs; the inputs differ in the last bits through the coefficients),
      * wavelet coefficients: grid gradients are float-atomic sums (order-dependent in the last bits) and Adam's
        first steps move an entry by ~lr g/|g|, so an entry whose gradient is rounding noise can move either way:
        <= 1e-4 of the tensor's largest entry for all but 1e-4 of the entries, and never by more than 2 lr K;
      * the per-step losses of B against the losses A's replays wrote (same batches): 1e-5 relative;
      * step 1's gradients (run B) against the oracle's autograd on the same batch and ground truth: 2e-5."""
    import bench
    K, warm, lr = 5, 3, 0.008
    ctx_a = bench.cfg3_train_setup(dev, 'f16x2')
    graph, loss_a, eager_a = bench.capture_train_step(ctx_a, eager_warmup=warm)
    replay_losses = []
    for _ in range(K):
        graph.replay()
        replay_losses.append(float(loss_a.detach()))          # .item() synchronises: the replay is done
    torch.cuda.synchronize()
    assert ctx_a['ds']._sample_state.cpu().tolist() == [warm + K, 0]          # capture itself drew nothing

    ctx_b = bench.cfg3_train_setup(dev, 'f16x2')
    mb = ctx_b['model']
    for (ka, pa), (kb, pb) in zip(ctx_a['model'].named_parameters(), mb.named_parameters()):
        assert ka == kb and pa.shape == pb.shape
    start = {k: p.detach().clone() for k, p in mb.named_parameters()}
    losses_b, step1 = [], None
    for i in range(warm + K):
        losses_b.append(float(ctx_b['step']().detach()))
        if i == 0:
            raw1, norm1 = (t.detach().clone() for t in ctx_b['last_batch'])
            step1 = {k: p.grad.detach().clone() for k, p in mb.named_parameters()}
    torch.cuda.synchronize()
    assert ctx_b['ds']._sample_state.cpu().tolist() == [warm + K, 0]
    for i in range(warm):
        assert abs(float(eager_a[i]) - losses_b[i]) <= 1e-5 * abs(losses_b[i]), (i, float(eager_a[i]), losses_b[i])
    for i in range(K):
        assert abs(replay_losses[i] - losses_b[warm + i]) <= 1e-5 * abs(losses_b[warm + i]), (i, replay_losses[i], losses_b[warm + i])
    moved = 0.0
    for (k, pa), (_, pb) in zip(ctx_a['model'].named_parameters(), mb.named_parameters()):
        a, b = pa.detach().cpu().numpy().astype(np.float64), pb.detach().cpu().numpy().astype(np.float64)
        moved = max(moved, float(np.abs(b - start[k].cpu().numpy()).max()))
        d, top = np.abs(a - b), np.abs(b).max()
        if k.startswith('feature_grid'):
            assert d.max() <= 2 * lr * (warm + K) * (1 + 1e-3), k
            assert float((d > 1e-4 * top).mean()) <= 1e-4, (k, float((d > 1e-4 * top).mean()), d.max(), top)
        else:
            assert d.max() <= 1e-5 * top, (k, d.max(), top)
    assert moved > lr                                      # parameters did move
    # step 1 of run B against the oracle's autograd (same batch, ground truth from the oracle's own sampler)
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    sd = {k: v.cpu() for k, v in start.items()}
    nc, L = len(mb.feature_grid), mb.num_layer
    coeffs = [sd['feature_grid.%d' % i].clone().requires_grad_(True) for i in range(nc)]
    ws = [sd['net_layers.%d.weight' % i].clone().requires_grad_(True) for i in range(L)] + [sd['final_layer.weight'].clone().requires_grad_(True)]
    bs = [sd['net_layers.%d.bias' % i].clone().requires_grad_(True) for i in range(L)] + [sd['final_layer.bias'].clone().requires_grad_(True)]
    mn, mx, rs = (torch.tensor(v, dtype=torch.float32) for v in ctx_b['bounds'])
    gt = R.trilinear_f_interpolation(raw1.cpu(), ctx_b['vol'].cpu(), mn, mx, rs)
    yr = R.forward(coeffs, mb.shape_array, mb.filter.filter_rev.detach().cpu(), ws, bs, norm1.cpu(), 2, training=True)
    lr_ = torch.nn.functional.mse_loss(yr.squeeze(-1), gt)
    lr_.backward()
    assert abs(losses_b[0] - lr_.item()) <= 1e-5 * abs(lr_.item())
    ref = {'feature_grid.%d' % i: c.grad.numpy() for i, c in enumerate(coeffs)}
    for i in range(L):
        ref['net_layers.%d.weight' % i], ref['net_layers.%d.bias' % i] = ws[i].grad.numpy(), bs[i].grad.numpy()
    ref['final_layer.weight'], ref['final_layer.bias'] = ws[L].grad.numpy(), bs[L].grad.numpy()
    for k, g in step1.items():
        assert rel_err(g.cpu().numpy(), ref[k]) <= 2e-5, k
