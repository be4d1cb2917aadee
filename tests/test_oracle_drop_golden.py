"""CPU: pin oracle/ref_drop.py (pruning layers inside decode_volume) against fixtures captured
from the reference's own modules (tools/make_goldens_drop.py).  Op-for-op restatement => bit-identical on this torch."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_drop as D
from oracle import ref_torch as R

torch.set_num_threads(4)
TYPES = ['smallify', 'straight_through', 'masked_straight_through', 'variational']


def t(a):
    return torch.from_numpy(np.asarray(a))


class OracleDropModel:
    """State of one fixture model + the reference's decode / forward with the drop hooks, from oracle functions."""

    def __init__(self, g, kind, prefix='sd.'):
        self.kind = kind
        self.momentum, self.threshold = [float(v) for v in g['momentum_threshold']]
        n = 0
        while prefix + 'feature_grid.%d' % n in g:
            n += 1
        L = 0
        while prefix + 'net_layers.%d.weight' % L in g:
            L += 1
        self.coeffs = [t(g[prefix + 'feature_grid.%d' % i]).clone().requires_grad_(True) for i in range(n)]
        self.weights = [t(g[prefix + 'net_layers.%d.weight' % i]).clone().requires_grad_(True) for i in range(L)] + \
                       [t(g[prefix + 'final_layer.weight']).clone().requires_grad_(True)]
        self.biases = [t(g[prefix + 'net_layers.%d.bias' % i]).clone().requires_grad_(True) for i in range(L)] + \
                      [t(g[prefix + 'final_layer.bias']).clone().requires_grad_(True)]
        self.frev = t(g[prefix + 'filter.filter_rev'])
        self.shape_array = g['shape_array']
        names = {'smallify': ['betas'], 'straight_through': ['mask_values'], 'masked_straight_through': ['mask_values'],
                 'variational': ['log_thetas', 'log_var']}[kind]
        self.drop = [{k: t(g[prefix + 'drop.%d.%s' % (i, k)]).clone().requires_grad_(True) for k in names}
                     for i in range(n)]
        self.d_mask = [None] * n

    def dropped(self, noise, training=True):
        out = []
        for i, (c, p) in enumerate(zip(self.coeffs, self.drop)):
            if self.kind == 'smallify':
                out.append(D.smallify_apply(c, p['betas'], self.d_mask[i], training))
            elif self.kind == 'straight_through':
                out.append(D.ste_apply(c, p['mask_values'], noise[i], training))
            elif self.kind == 'masked_straight_through':
                out.append(D.masked_ste_apply(c, p['mask_values'], self.threshold, self.d_mask[i], training))
            else:
                out.append(D.variational_apply(c, p['log_thetas'], p['log_var'], noise[i], self.d_mask[i]))
        return out

    def decode(self, noise, training=True):
        return D.decode_volume_dropped(self.dropped(noise, training), self.shape_array, self.frev)

    def forward(self, pos, noise):
        dense = self.decode(noise, True)
        return R.forward_from_grid(dense, self.weights, self.biases, pos, 2)

    def named(self):
        out = {}
        for i, c in enumerate(self.coeffs):
            out['feature_grid.%d' % i] = c
        for i, p in enumerate(self.drop):
            for k, v in p.items():
                out['drop.%d.%s' % (i, k)] = v
        L = len(self.weights) - 1
        for i in range(L):
            out['net_layers.%d.weight' % i] = self.weights[i]
            out['net_layers.%d.bias' % i] = self.biases[i]
        out['final_layer.weight'] = self.weights[L]
        out['final_layer.bias'] = self.biases[L]
        return out


def noise_of(g, tag, n=3):
    return [t(g['%s.%d' % (tag, i)]) if '%s.%d' % (tag, i) in g else None for i in range(n)]


@pytest.mark.parametrize('kind', TYPES)
def test_decode_with_drop_layers(golden_dir, kind):
    g = np.load(os.path.join(golden_dir, 'drop_%s.npz' % kind))
    m = OracleDropModel(g, kind)
    with torch.no_grad():
        assert np.array_equal(m.decode(noise_of(g, 'noise_decode_train'), True).numpy(), g['decoded_train'])
        assert np.array_equal(m.decode(noise_of(g, 'noise_decode_eval'), False).numpy(), g['decoded_eval'])


@pytest.mark.parametrize('kind', TYPES)
def test_loss_and_gradients(golden_dir, kind):
    g = np.load(os.path.join(golden_dir, 'drop_%s.npz' % kind))
    m = OracleDropModel(g, kind)
    pos = t(g['pos']).clone().requires_grad_(True)
    target = t(g['target'])
    pred = m.forward(pos, noise_of(g, 'noise_step')).squeeze(-1)
    assert np.array_equal(pred.detach().numpy(), g['pred'])
    if kind == 'variational':
        dkl = [D.variational_dkl(p['log_thetas'], p['log_var']) for p in m.drop]
        assert np.array_equal(np.asarray([d.item() for d in dkl]), g['dkl_per_layer'])
        ent = [D.variational_entropy(p['log_thetas'], p['log_var']).item() for p in m.drop]
        assert np.array_equal(np.asarray(ent), g['entropy_per_layer'])
        vf = [torch.mean((D.variational_dropout_rates(p['log_thetas'], p['log_var']) < m.threshold).to(torch.float)).item()
              for p in m.drop]
        assert np.array_equal(np.asarray(vf), g['valid_fraction'])
        log_sigma = torch.ones_like(pred).fill_(-2.0)
        loss, ll, mse, dkl_sum, wsum, wd = D.variational_loss(dkl, m.coeffs, pred, target, log_sigma, float(24 ** 3),
                                                              float(pos.shape[0]), 1e-3, 1e-6, 0.01)
        assert ll.item() == float(g['loss_ll']) and mse.item() == float(g['loss_mse'])
        assert dkl_sum.item() == float(g['loss_dkl']) and wsum.item() == float(g['loss_weight'])
        assert wd == float(g['weight_dkl_after'])
    else:
        vol_loss = torch.nn.MSELoss()(pred, target)
        key = 'betas' if kind == 'smallify' else 'mask_values'
        l1 = [D.l1_penalty(p[key]) for p in m.drop]
        assert np.array_equal(np.asarray([v.item() for v in l1]), g['l1_per_layer'])
        d_loss = D.smallify_loss(l1, m.coeffs, 1e-3, 1e-5)
        assert vol_loss.item() == float(g['loss_vol']) and d_loss.item() == float(g['loss_drop'])
        loss = vol_loss + d_loss
    assert loss.item() == float(g['loss'])
    loss.backward()
    assert np.array_equal(pos.grad.numpy(), g['grad_pos'])
    for k, p in m.named().items():
        got = p.grad.numpy() if p.grad is not None else np.zeros(p.shape, np.float32)
        assert np.array_equal(got, g['grad.' + k]), k


def test_sign_variance_tracker(golden_dir):
    g = np.load(os.path.join(golden_dir, 'drop_smallify.npz'))
    mom = float(g['momentum_threshold'][0])
    for i in range(3):
        betas = t(g['sd.drop.%d.betas' % i])
        ema, var = t(g['ema0.%d' % i]), t(g['emavar0.%d' % i])
        assert var.abs().max() == 0
        for step in (1, 2):
            ema, var = D.sign_variance_update(ema, var, betas, mom)
            assert np.array_equal(ema.numpy(), g['ema%d.%d' % (step, i)])
            assert np.array_equal(var.numpy(), g['emavar%d.%d' % (step, i)])
    # six more updates with sign flips (replayed from the generator's seeded stream)
    rng = np.random.Generator(np.random.PCG64(7001 + 5))
    betas = [t(g['sd.drop.%d.betas' % i]).clone() for i in range(3)]
    state = [(t(g['ema2.%d' % i]), t(g['emavar2.%d' % i])) for i in range(3)]
    for _ in range(3, 9):
        for i in range(3):
            flip = t(np.where(rng.random(tuple(betas[i].shape)) < 0.4, -1.0, 1.0).astype(np.float32))
            betas[i] = betas[i] * flip
            state[i] = D.sign_variance_update(state[i][0], state[i][1], betas[i], mom)
    for i in range(3):
        assert np.array_equal(betas[i].numpy(), g['betas8.%d' % i])
        assert np.array_equal(state[i][0].numpy(), g['ema8.%d' % i])
        assert np.array_equal(state[i][1].numpy(), g['emavar8.%d' % i])
        mask = D.sign_variance_mask(state[i][1], float(g['thresholds8'][i]))
        assert np.array_equal(mask.numpy(), g['d_mask.%d' % i])


@pytest.mark.parametrize('kind', TYPES)
def test_pruning_fold_and_removal(golden_dir, kind):
    g = np.load(os.path.join(golden_dir, 'drop_%s.npz' % kind))
    m = OracleDropModel(g, kind)
    n = len(m.coeffs)
    with torch.no_grad():
        if kind == 'smallify':
            for i in range(n):
                m.drop[i]['betas'] = t(g['betas8.%d' % i])           # the tracker rounds flipped the signs in place
            masks = [D.sign_variance_mask(t(g['emavar8.%d' % i]), float(g['thresholds8'][i])) for i in range(n)]
            folded = [D.smallify_fold(c, p['betas'], mk) for c, p, mk in zip(m.coeffs, m.drop, masks)]
            m.d_mask = masks
            removal = masks
        elif kind == 'straight_through':
            # save_dropvalues_on_grid raises TypeError at the size_layer() sum, AFTER it has already
            # replaced feature_grid by grid * (mask_values > threshold) (model/Feature_Grid_Model.py:116-117)
            assert bool(g['save_raises'])
            removal = [D.ste_prune_mask(p['mask_values'], m.threshold) for p in m.drop]
            folded = [c * mk for c, mk in zip(m.coeffs, removal)]
        elif kind == 'masked_straight_through':
            pairs = [D.masked_ste_fold(c, p['mask_values'], m.threshold) for c, p in zip(m.coeffs, m.drop)]
            folded = [a for a, _ in pairs]
            m.d_mask = [b for _, b in pairs]
            removal = [torch.sigmoid(p['mask_values']) for p in m.drop]   # calculate_pruning_mask returns the SOFT mask
        else:
            masks = [D.variational_prune_mask(p['log_thetas'], p['log_var'], m.threshold) for p in m.drop]
            folded = [D.variational_fold(c, p['log_thetas'], mk) for c, p, mk in zip(m.coeffs, m.drop, masks)]
            m.d_mask = masks
            removal = masks
        if folded is not None:
            for i in range(n):
                assert np.array_equal(folded[i].numpy(), g['saved.feature_grid.%d' % i]), i
                if m.d_mask[i] is not None:
                    assert np.array_equal(m.d_mask[i].float().numpy(), g['d_mask.%d' % i])
            if 'zeros' in g:
                sizes = [int(np.prod(c.shape[1:])) for c in m.coeffs]
                assert float(D.pruned_count(folded, sizes)) == float(g['zeros'])
            m.coeffs = [f.clone() for f in folded]
    # masked forward after pruning (drop layers now multiply by the stored mask) + gradients
    for c in m.coeffs:
        c.requires_grad_(True)
    pos, target = t(g['pos']), t(g['target'])
    pred = m.forward(pos, noise_of(g, 'noise_pruned')).squeeze(-1)
    assert np.array_equal(pred.detach().numpy(), g['pred_pruned'])
    torch.nn.MSELoss()(pred, target).backward()
    for k, p in m.named().items():
        got = p.grad.numpy() if p.grad is not None else np.zeros(p.shape, np.float32)
        assert np.array_equal(got, g['grad_pruned.' + k]), k
    with torch.no_grad():
        removed = [c * mk for c, mk in zip(m.coeffs, removal)]
        for i in range(n):
            assert np.array_equal(removed[i].numpy(), g['removed.feature_grid.%d' % i])
        dense = R.decode_volume(removed, m.shape_array, m.frev)
        y = R.forward_from_grid(dense, m.weights, m.biases, pos, 2).squeeze(-1)
        assert np.array_equal(y.numpy(), g['pred_removed'])
