"""Secondary measurement: BASELINE config 3 train step (64^3 x 32ch grid, MLP 4x128, 2048x16 = 32768 lattice
samples per step, fp32): forward + GT gather + MSE + backward + Adam, all on one MI355X.

    python tools/bench_trainstep.py [--steps 50] [--warmup 10]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--no-input-grad', action='store_true')
    ap.add_argument('--foreach-adam', action='store_true', help="torch's default (foreach) Adam instead of fused=True")
    ap.add_argument('--graph', action='store_true', help='capture one train step in a HIP graph and replay it')
    ap.add_argument('--precision', default='f16x2', choices=['f16x2', 'fp32', 'f16'])
    ap.add_argument('--torch-sampler', action='store_true', help='torch.randint + the position kernel and a plain loss.backward() (the round-1 step)')
    ap.add_argument('--torch-loss', action='store_true', help='GT gather + torch MSELoss instead of the fused GT+MSE kernel')
    ap.add_argument('--drop-type', default='', choices=['', 'smallify', 'masked_straight_through', 'variational'],
                    help='pruning layers on the coefficients + their loss (the reference CLI default is smallify)')
    ap.add_argument('--unfused-drop', action='store_true',
                    help='comparison: apply the drop factors and penalties with torch ops instead of the fused HIP kernels')
    args = ap.parse_args()
    from latent_feature_grid_compression_amd.data.Interpolation import trilinear_f_interpolation, trilinear_mse_loss
    from latent_feature_grid_compression_amd.data.IndexDataset import IndexDataset
    dev = torch.device('cuda:0')
    w = bench.WORKLOADS['headline']
    model = bench.build_model(w, seed=2003, device=dev).train()
    model.precision = args.precision
    drop_loss = None
    if args.drop_type:
        import torch.nn as nn
        from latent_feature_grid_compression_amd.model.Smallify_Dropout import SmallifyDropout, SmallifyLoss
        from latent_feature_grid_compression_amd.model.Straight_Through_Dropout import MaskedWavelet_Straight_Through_Dropout
        from latent_feature_grid_compression_amd.model.Variational_Dropout_Layer import VariationalDropout, VariationalDropoutLoss
        cls, a, b = {'smallify': (SmallifyDropout, 0.025, 0.75),
                     'masked_straight_through': (MaskedWavelet_Straight_Through_Dropout, 0.5, 0.5),
                     'variational': (VariationalDropout, 0.5, 0.9)}[args.drop_type]
        layers = [cls(tuple(f.shape[1:]), a, b).to(dev) for f in model.feature_grid]

        class TorchDrop(nn.Module):
            """The same layer applied the reference's way: a coefficient-sized torch multiply + autograd."""
            def __init__(self, inner):
                super().__init__()
                self.inner = inner

            def forward(self, x):
                f = self.inner.drop_factor()
                if f.threshold is None:
                    return x * f.mul.unsqueeze(0)
                return (x * (f.mul >= f.threshold) - x * f.mul).detach() + x * f.mul

        model.drop = nn.ModuleList([TorchDrop(l) for l in layers] if args.unfused_drop else layers)
        if args.drop_type == 'variational':
            crit = VariationalDropoutLoss(size_volume=255.0 ** 3, batch_size=32768.0, weight_dkl=1e-6, weight_weights=1e-8)
            if args.unfused_drop:
                def drop_loss(pred, gt):
                    dkl = sum(l.inner_dkl() for l in layers)
                    raise SystemExit('unfused variational loss not wired; use smallify for the comparison')
            else:
                drop_loss = lambda pred, gt: crit(model, pred, gt, torch.full_like(pred, -2.0), 0.0)[0]
        elif args.unfused_drop:
            key = 'betas' if args.drop_type == 'smallify' else 'mask_values'
            drop_loss = lambda pred, gt: loss_fn(pred, gt) + 1e-6 * sum(torch.abs(getattr(l, key)).sum() for l in layers) + \
                1e-8 * sum(torch.sum(torch.abs(f) ** 2) for f in model.feature_grid)
        else:
            sl = SmallifyLoss(1e-6, 1e-8)
            drop_loss = lambda pred, gt: loss_fn(pred, gt) + sl(model)
    rng = np.random.Generator(np.random.PCG64(1003))
    vol = torch.from_numpy(rng.uniform(-1, 1, (255, 255, 255)).astype(np.float32)).to(dev)
    ds = IndexDataset((255, 255, 255), 16, build_index_table=False)
    # torch's single-kernel fused Adam by default: the foreach form the reference's torch.optim.Adam(lr) defaults to makes
    # ~10 passes over the 38 MB of coefficients (0.65 ms per step, more than everything else together); --foreach-adam
    # selects it.  Same algorithm; the optimiser is outside the scope of this package either way.
    opt = torch.optim.Adam(model.parameters(), lr=0.008, capturable=args.graph, fused=not args.foreach_adam)
    loss_fn = torch.nn.MSELoss()
    n = 2048 * 16
    mn, mx, rs = ds.min_idx.clone(), ds.max_idx.clone(), ds.vol_res.clone()      # host copies for the GT sampler's bounds
    mn_h, mx_h, rs_h = mn.tolist(), mx.tolist(), rs.tolist()
    ds.min_idx, ds.max_idx, ds.scales = ds.min_idx.to(dev), ds.max_idx.to(dev), ds.scales.to(dev)   # no H2D inside the step

    from latent_feature_grid_compression_amd.data.Interpolation import mse_unit_grad
    unit = mse_unit_grad(dev)                          # outside any capture
    fused_seed = drop_loss is None and not args.torch_loss     # the loss node that recognises the unit seed

    def step(i, backward=True):
        if args.torch_sampler:
            if args.graph:
                flat = torch.randint(0, ds.n_voxels, (n,), device=dev)    # default generator: graph-safe philox state
            else:
                g = torch.Generator(device=dev)
                g.manual_seed(3003 + i)
                flat = torch.randint(0, ds.n_voxels, (n,), device=dev, generator=g)
            raw, norm = ds.positions_from_flat(flat)
        else:
            raw, norm = ds.sample_positions(n, dev, seed=3003)            # draw + positions in one kernel
        norm.requires_grad = not args.no_input_grad
        opt.zero_grad()
        pred = model(norm).squeeze(-1)
        if drop_loss is None and not args.torch_loss:
            loss = trilinear_mse_loss(pred, raw, vol, mn_h, mx_h, rs_h)
        else:
            gt = trilinear_f_interpolation(raw, vol, mn, mx, rs)
            loss = loss_fn(pred, gt) if drop_loss is None else drop_loss(pred, gt)
        if backward:
            loss.backward(unit) if fused_seed and not args.torch_sampler else loss.backward()
            opt.step()
        return loss

    out = {}
    if args.graph:
        # whole step (sampler -> forward -> GT -> MSE -> backward -> Adam) captured once, replayed: removes the ~60 host
        # launches per step; the C-ABI entry points neither allocate nor synchronise, so they capture like torch's own ops
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for i in range(3):
                step(i)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(graph):
            static_loss = step(0)
        for _ in range(args.warmup):
            graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            graph.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        out['train_step_graph'] = {'ms_per_step': dt * 1e3, 'Msamples_per_s': n / dt / 1e6}
        out['final_loss'] = float(static_loss)
        out['config'] = 'cfg3 train step replayed from one HIP graph (same work as train_step); drop %s%s' % (
            args.drop_type or 'none', ' (torch ops)' if args.unfused_drop else '')
        print(json.dumps(out))
        return
    for name, bw in (('fwd_only', False), ('train_step', True)):
        for i in range(args.warmup):
            step(i, bw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            loss = step(i, bw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        out[name] = {'ms_per_step': dt * 1e3, 'Msamples_per_s': n / dt / 1e6}
    out['final_loss'] = float(loss)
    out['config'] = 'cfg3: 64^3x32ch grid, MLP 4x128, 32768 lattice samples/step, fp32, Adam; input grad %s; drop %s%s' % (
        'off' if args.no_input_grad else 'on (reference sets requires_grad on positions)', args.drop_type or 'none',
        ' (torch ops)' if args.unfused_drop else '')
    print(json.dumps(out))


if __name__ == '__main__':
    main()
