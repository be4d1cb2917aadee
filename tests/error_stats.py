"""Error of the fused forward per arithmetic build on the reference fixtures and on synthetic BASELINE shapes:
max|y - y_ref| / max|y_ref| (the north_star bound, <= 1e-5) and the worst element of the allclose test
(|y - y_ref| - rtol |y_ref|) / max|y_ref| (<= 2e-6 in tests/test_hip_forward.py).  GPU box only.  Test infrastructure (it checks the HIP path against the oracle):
lives under tests/, not collected by pytest.
    python tests/error_stats.py
"""
import glob
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))      # build_from_golden / build_synth of the parity tests
from test_hip_forward import build_from_golden, build_synth, GOLD  # noqa: E402
from oracle import ref_torch as R  # noqa: E402
from oracle import ref_explicit as E  # noqa: E402


def stats(y, yr):
    y, yr = np.asarray(y, np.float64).reshape(-1), np.asarray(yr, np.float64).reshape(-1)
    mx = np.abs(yr).max()
    return np.abs(y - yr).max() / mx, (np.abs(y - yr) - 1e-5 * np.abs(yr)).max() / mx


def main():
    dev = torch.device('cuda:0')
    for f in sorted(glob.glob(os.path.join(GOLD, 'fwd_*.npz'))):
        g = np.load(f)
        m = build_from_golden(g, dev).train()
        pos = torch.from_numpy(g['pos']).to(dev)
        row = [os.path.basename(f)]
        for prec in ('fp32', 'f16x2', 'f16'):
            m.precision = prec
            with torch.no_grad():
                y = m(pos).cpu().numpy()
            row.append('%s rel %.2e allclose-margin %.2e' % ((prec,) + stats(y, g['y'])))
        print(' | '.join(row), flush=True)
    for (C, G, H, L, n) in [(16, 32, 64, 4, 50000), (32, 64, 128, 4, 40000), (22, 17, 32, 4, 10001), (32, 20, 128, 8, 4096)]:
        m, sm = build_synth(C, G, H, L, seed=4000 + C + G + H, dev=dev)
        rng = np.random.default_rng(C * 1000 + G)
        pos = torch.from_numpy(rng.uniform(-1, 1, (n, 3)).astype(np.float32))
        sub = slice(0, 4000)
        y64 = E.forward_from_grid(E.decode_volume([c.numpy() for c in sm['coeffs']], sm['shape_array'], sm['filter_rev'].numpy()),
                                  [w.numpy() for w in sm['weights']], [b.numpy() for b in sm['biases']], pos[sub].numpy(), 2)
        dense = R.decode_volume(sm['coeffs'], sm['shape_array'], sm['filter_rev'])
        yref = R.forward_from_grid(dense, sm['weights'], sm['biases'], pos[sub], 2).numpy()
        row = ['synthetic C%d G%d H%d L%d: torch-CPU vs fp64 %.2e' % (C, G, H, L, stats(yref, y64)[0])]
        m.train()
        for prec in ('fp32', 'f16x2', 'f16'):
            m.precision = prec
            with torch.no_grad():
                y = m(pos[sub].to(dev)).cpu().numpy()
            row.append('%s vs fp64 %.2e vs CPU %.2e' % (prec, stats(y, y64)[0], stats(y, yref)[0]))
        print(' | '.join(row), flush=True)


if __name__ == '__main__':
    main()
