"""Randomised sweep of the lattice-mode forward (hand-run on the GPU box: python tests/fuzz_zrun.py [n] [seed]).

For random nets, grids, volumes and x-slabs the fused lattice launch is run twice -- z-run tiles + column sampler where the
host selects them, and the per-sample gather (LFGC_NO_ZRUN=1) -- and both are compared with each other (<= 3e-6 of the
largest output: the two evaluate the same trilinear sum in a different order) and, on a random subset of voxels, with the
position-list entry fed the reference-style tile positions (the path the oracle tests pin).  Not collected by pytest."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import ref_torch as R                          # noqa: E402
from test_hip_forward import build_synth, rel_err         # noqa: E402
from latent_feature_grid_compression_amd import ops       # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    dev = torch.device('cuda:0')
    worst, taken = 0.0, 0
    for case in range(n_cases):
        C = int(rng.choice([3, 8, 16, 22, 24, 32]))
        G = int(rng.choice([6, 8, 12, 16, 20, 32]))
        H = int(rng.choice([16, 32, 64, 100, 128]))
        L = int(rng.integers(1, 6))
        res = (int(rng.integers(2, 12)), int(rng.integers(2, 40)), int(rng.choice([2, 5, 31, 32, 33, 64, 95, 150, 256, 300])))
        xb = int(rng.integers(0, res[0]))
        xe = int(rng.integers(xb + 1, res[0] + 1))
        m, sm = build_synth(C, G, H, L, seed=int(rng.integers(1, 10 ** 6)), dev=dev)
        m.eval()
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
        e = rel_err(outs['zrun'].numpy(), outs['gather'].numpy())
        short = int(31.0 * G / (res[2] - 1) + 1e-3) + 3 <= 12
        same = bool(torch.equal(outs['zrun'], outs['gather']))
        taken += int(not same)
        # a few tiles through the position-list entry with the reference-style tile positions
        rds = R.VolumeIndexing(res)
        tiles = [b for b in R.tile_iter(rds.vol_res_touple, 32) if b[0] < xe and b[1] > xb]
        e2 = 0.0
        for b in [tiles[i] for i in rng.choice(len(tiles), size=min(3, len(tiles)), replace=False)]:
            pos = R.tile_positions(rds, b).reshape(-1, 3).to(dev)
            with torch.no_grad():
                yp, _ = ops.forward_raw(m._descriptor(), m._decoded_channel_last(), m._packed(), pos=pos, clamp=True)
            yp = yp.view(b[1] - b[0], b[3] - b[2], b[5] - b[4]).cpu()
            x0, x1 = max(b[0], xb), min(b[1], xe)
            e2 = max(e2, rel_err(outs['zrun'][x0 - xb:x1 - xb, b[2]:b[3], b[4]:b[5]].numpy(), yp[x0 - b[0]:x1 - b[0]].numpy()))
        worst = max(worst, e, e2)
        ok = e <= 3e-6 and e2 <= 3e-6 and np.isfinite(outs['zrun'].numpy()).all() and (same or short)
        print('%s case %2d C%-2d G%-2d H%-3d L%d res %-14s slab [%d,%d): zrun vs gather %.1e, vs position list %.1e%s'
              % ('ok  ' if ok else 'FAIL', case, C, G, H, L, res, xb, xe, e, e2, '' if not same else '  (same path)'), flush=True)
        if not ok:
            sys.exit(1)
    print('all %d cases ok, worst %.2e, column sampler taken in %d' % (n_cases, worst, taken))


if __name__ == '__main__':
    main()
