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
    args = ap.parse_args()
    from latent_feature_grid_compression_amd.data.Interpolation import trilinear_f_interpolation
    from latent_feature_grid_compression_amd.data.IndexDataset import IndexDataset
    dev = torch.device('cuda:0')
    w = bench.WORKLOADS['headline']
    model = bench.build_model(w, seed=2003, device=dev).train()
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
    ds.min_idx, ds.max_idx, ds.scales = ds.min_idx.to(dev), ds.max_idx.to(dev), ds.scales.to(dev)   # no H2D inside the step

    def step(i, backward=True):
        if args.graph:
            flat = torch.randint(0, ds.n_voxels, (n,), device=dev)        # default generator: graph-safe philox state
        else:
            g = torch.Generator(device=dev)
            g.manual_seed(3003 + i)
            flat = torch.randint(0, ds.n_voxels, (n,), device=dev, generator=g)
        raw = ds.lattice_from_flat(flat)
        _, norm = ds.positions_for(raw)
        norm.requires_grad = not args.no_input_grad
        opt.zero_grad()
        pred = model(norm).squeeze(-1)
        gt = trilinear_f_interpolation(raw, vol, mn, mx, rs)
        loss = loss_fn(pred, gt)
        if backward:
            loss.backward()
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
        out['config'] = 'cfg3 train step replayed from one HIP graph (same work as train_step)'
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
    out['config'] = 'cfg3: 64^3x32ch grid, MLP 4x128, 32768 lattice samples/step, fp32, Adam; input grad %s' % (
        'off' if args.no_input_grad else 'on (reference sets requires_grad on positions)')
    print(json.dumps(out))


if __name__ == '__main__':
    main()
