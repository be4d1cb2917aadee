"""Fixtures for the pruning ("drop") layers acting inside decode_volume, generated from
the REFERENCE's own modules (model/Smallify_Dropout.py, model/Straight_Through_Dropout.py,
model/Variational_Dropout_Layer.py, model/Feature_Grid_Model.py) in the build container:

    python tools/make_goldens_drop.py        -> tests/golden/drop_<type>.npz

Per drop type: a small model with every parameter overwritten from a seeded numpy stream; the random draws the
layers make (torch.rand / torch.randn_like, CPU generator) recorded in call order; decoded volume in train and eval
mode; train-mode forward, the reference's loss (MSE + SmallifyLoss, or VariationalDropoutLoss) and the gradients of
every parameter; the sign-variance tracker state over three forwards; the pruning masks, save_dropvalues_on_grid(),
the masked forward after pruning and remove_drop_layers().  Data only; no reference source is stored.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _ref_standins                                            # noqa: E402

_ref_standins.install()
GOLD = _ref_standins.GOLD

from model.model_utils import setup_model                       # noqa: E402
from model.Dropout_Layer import DropoutLayer                    # noqa: E402
from model.Smallify_Dropout import SmallifyLoss, SmallifyDropout          # noqa: E402
from model.Variational_Dropout_Layer import VariationalDropoutLoss, VariationalDropout   # noqa: E402

torch.set_num_threads(4)
C, G, H, L = 4, 15, 16, 3
TYPES = {'smallify': (0.025, 0.75), 'straight_through': (0.5, 0.5), 'masked_straight_through': (0.5, 0.5),
         'variational': (0.5, 0.9)}


def rng_for(seed):
    return np.random.Generator(np.random.PCG64(seed))


def f32(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float32))


def build(drop_type, seed):
    momentum, threshold = TYPES[drop_type]
    DropoutLayer.set_threshold_list(None)
    DropoutLayer.i = 0
    model = setup_model(3, H, 1, L, 'fourier', 2, drop_type, momentum, threshold, 'db2', C, G, '')
    rng = rng_for(seed)
    grid = f32(rng.random((C, G, G, G), dtype=np.float32))
    feats, _ = model.encode_volume(grid)
    with torch.no_grad():
        for p, f in zip(model.feature_grid, feats):
            p.copy_(f)
        for lin in list(model.net_layers) + [model.final_layer]:
            bound = 1.0 / np.sqrt(lin.in_features)
            lin.weight.copy_(f32(rng.uniform(-bound, bound, lin.weight.shape)))
            lin.bias.copy_(f32(rng.uniform(-bound, bound, lin.bias.shape)))
        for d in model.drop:
            if drop_type == 'smallify':
                d.betas.copy_(f32(rng.normal(0, 1, d.betas.shape)))
            elif drop_type == 'straight_through':
                d.mask_values.copy_(f32(rng.uniform(0.0, 1.3, d.mask_values.shape)))
            elif drop_type == 'masked_straight_through':
                d.mask_values.copy_(f32(rng.uniform(-2.0, 2.0, d.mask_values.shape)))
            else:
                d.log_thetas.copy_(f32(rng.normal(0, 0.3, d.log_thetas.shape)))
                d.log_var.copy_(f32(rng.normal(0.0, 2.0, d.log_var.shape)))
    return model, rng


def draws_for(model, drop_type, seed):
    """The random tensors the layers will draw in decode_volume() order after torch.manual_seed(seed)."""
    torch.manual_seed(seed)
    out = []
    for d in model.drop:
        if drop_type == 'straight_through':
            out.append(torch.rand(d.c))
        elif drop_type == 'variational':
            out.append(torch.randn_like(torch.exp(d.log_thetas)))
    torch.manual_seed(seed)
    return out


def state(model, prefix):
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def gen(drop_type, seed):
    model, rng = build(drop_type, seed)
    out = {'meta': np.asarray([C, G, H, L, 2]), 'shape_array': np.asarray(model.shape_array),
           'momentum_threshold': np.asarray(TYPES[drop_type], np.float64)}
    out.update(state(model, 'sd.'))
    if drop_type == 'smallify':
        for i, d in enumerate(model.drop):
            out['ema0.%d' % i] = d.tracker.EMA.detach().numpy().copy()
            out['emavar0.%d' % i] = d.tracker.EMAVar.detach().numpy().copy()
    n = 300
    pos = f32(rng.uniform(-1, 1, (n, 3)))
    target = f32(rng.uniform(-1, 1, (n,)))
    out.update(pos=pos.numpy(), target=target.numpy())

    # ---- decode in train and eval mode --------------------------------------------------------------
    model.train()
    noise = draws_for(model, drop_type, 11)
    with torch.no_grad():
        out['decoded_train'] = model.decode_volume().numpy().copy()
    for i, z in enumerate(noise):
        out['noise_decode_train.%d' % i] = z.numpy().copy()
    model.eval()
    noise = draws_for(model, drop_type, 12)
    with torch.no_grad():
        out['decoded_eval'] = model.decode_volume().numpy().copy()
    for i, z in enumerate(noise):
        out['noise_decode_eval.%d' % i] = z.numpy().copy()
    model.train()
    if drop_type == 'smallify':           # the train-mode decode above advanced the tracker once
        for i, d in enumerate(model.drop):
            out['ema1.%d' % i] = d.tracker.EMA.detach().numpy().copy()
            out['emavar1.%d' % i] = d.tracker.EMAVar.detach().numpy().copy()

    # ---- one loss evaluation the way training/training.py:103-137 forms it ----------------------------
    noise = draws_for(model, drop_type, 13)
    for i, z in enumerate(noise):
        out['noise_step.%d' % i] = z.numpy().copy()
    pos_req = pos.clone().requires_grad_(True)
    model.zero_grad()
    pred = model(pos_req).squeeze(-1)
    if drop_type == 'variational':
        crit = VariationalDropoutLoss(size_volume=float(24 ** 3), batch_size=float(n), weight_dkl=1e-3, weight_weights=1e-6)
        log_sigma = torch.ones_like(pred).fill_(-2.0)
        loss, ll, mse, dkl, wsum = crit(model, pred, target, log_sigma, 0.01)
        out.update(loss_ll=np.asarray(ll.item(), np.float64), loss_mse=np.asarray(mse.item(), np.float64),
                   loss_dkl=np.asarray(dkl.item(), np.float64), loss_weight=np.asarray(wsum.item(), np.float64),
                   weight_dkl_after=np.asarray(crit.weight_dkl, np.float64),
                   dkl_per_layer=np.asarray([d.calculate_Dkl().item() for d in model.drop], np.float64),
                   entropy_per_layer=np.asarray([d.calculate_Dropout_Entropy().item() for d in model.drop], np.float64),
                   valid_fraction=np.asarray([d.get_valid_fraction()[0] for d in model.drop], np.float64))
    else:
        vol_loss = torch.nn.MSELoss()(pred, target)
        crit = SmallifyLoss(weight_l1=1e-3, weight_l2=1e-5)
        d_loss = crit(model)
        loss = vol_loss + d_loss
        out.update(loss_vol=np.asarray(vol_loss.item(), np.float64), loss_drop=np.asarray(d_loss.item(), np.float64),
                   l1_per_layer=np.asarray([d.l1_loss().item() for d in model.drop], np.float64))
    loss.backward()
    out.update(pred=pred.detach().numpy().copy(), loss=np.asarray(loss.item(), np.float64),
               grad_pos=pos_req.grad.numpy().copy())
    for k, p in model.named_parameters():
        out['grad.' + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()
        out['hasgrad.' + k] = np.asarray(p.grad is not None)

    # ---- tracker: two more forwards with sign flips in between -----------------------------------------
    if drop_type == 'smallify':
        for i, d in enumerate(model.drop):
            out['ema2.%d' % i] = d.tracker.EMA.detach().numpy().copy()
            out['emavar2.%d' % i] = d.tracker.EMAVar.detach().numpy().copy()
        flip_rng = rng_for(seed + 5)
        for rnd in range(3, 9):
            with torch.no_grad():
                for d in model.drop:
                    flip = f32(np.where(flip_rng.random(d.betas.shape) < 0.4, -1.0, 1.0))
                    d.betas.mul_(flip)
                model.decode_volume()
        for i, d in enumerate(model.drop):
            out['betas8.%d' % i] = d.betas.detach().numpy().copy()
            out['ema8.%d' % i] = d.tracker.EMA.detach().numpy().copy()
            out['emavar8.%d' % i] = d.tracker.EMAVar.detach().numpy().copy()
        # make the variance threshold bite on part of the entries
        for d in model.drop:
            d.tracker.threshold = float(np.median(d.tracker.EMAVar.numpy()))
        out['thresholds8'] = np.asarray([d.tracker.threshold for d in model.drop], np.float64)

    # ---- pruning: masks, save_dropvalues_on_grid, masked forward, remove_drop_layers --------------------
    noise = draws_for(model, drop_type, 14)
    try:
        zeros = model.save_dropvalues_on_grid('cpu')
        out['save_raises'] = np.asarray(False)
        out['zeros'] = np.asarray(float(zeros), np.float64)
    except TypeError:
        out['save_raises'] = np.asarray(True)         # Straight_Through_Dropout has no size_layer()
    out.update(state(model, 'saved.'))
    for i, d in enumerate(model.drop):
        m = getattr(d, 'd_mask', None)
        if m is not None:
            out['d_mask.%d' % i] = m.detach().numpy().astype(np.float32)
    noise = draws_for(model, drop_type, 15)
    for i, z in enumerate(noise):
        out['noise_pruned.%d' % i] = z.numpy().copy()
    model.zero_grad()
    pred2 = model(pos).squeeze(-1)
    torch.nn.MSELoss()(pred2, target).backward()
    out['pred_pruned'] = pred2.detach().numpy().copy()
    for k, p in model.named_parameters():
        out['grad_pruned.' + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()
    model.remove_drop_layers('cpu')
    out.update(state(model, 'removed.'))
    with torch.no_grad():
        out['pred_removed'] = model(pos).squeeze(-1).numpy().copy()
    np.savez_compressed(os.path.join(GOLD, 'drop_%s.npz' % drop_type), **out)
    return sum(v.nbytes for v in out.values())


if __name__ == '__main__':
    for k, (t, s) in enumerate(zip(TYPES, (7001, 7002, 7003, 7004))):
        print(t, gen(t, s), 'bytes (raw)')
    # the class-level threshold list quirk (model/Dropout_Layer.py:8-19): thresholds handed out by construction order
    DropoutLayer.set_threshold_list([0.1, 0.2, 0.3, 0.4, 0.5, 0.6])
    model = setup_model(3, H, 1, L, 'fourier', 2, 'smallify', 0.025, 0.75, 'db2', C, G, '')
    np.savez(os.path.join(GOLD, 'drop_threshold_list.npz'),
             thresholds=np.asarray([d.threshold for d in model.drop], np.float64),
             counter=np.asarray(DropoutLayer.i))
    DropoutLayer.set_threshold_list(None)
