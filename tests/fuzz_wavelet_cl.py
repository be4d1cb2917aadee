"""Randomised shape sweep of the channel-last level kernels against the channel-first kernels + layout conversion
(not collected by pytest: run by hand on a GPU box, `python tests/fuzz_wavelet_cl.py [cases] [seed]`)."""
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latent_feature_grid_compression_amd import ops
from latent_feature_grid_compression_amd.wavelet_transform.Torch_Wavelet_Transform import WaveletFilter3d

dev = torch.device('cuda:0')
frev = WaveletFilter3d('db2').filter_rev.to(dev)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for i in range(cases):
    C = int(rng.integers(1, 33))
    big = rng.random() < 0.15
    d = [int(rng.integers(1, 72 if (big and a > 0) else 14)) for a in range(3)]
    t = [int(rng.integers(max(1, 2 * v - 1), 2 * v + 3)) for v in d]
    lll = torch.from_numpy(rng.standard_normal([C] + d).astype(np.float32)).to(dev)
    hf = torch.from_numpy(rng.standard_normal([C, 7] + d).astype(np.float32)).to(dev)
    want = ops.to_channel_last(ops.idwt_level(lll, hf, frev, t))
    got = ops.idwt_level_cl(lll, hf, frev, t)
    e1 = float((got - want).abs().max()) / max(float(want.abs().max()), 1e-30)
    cs = got.shape[-1]
    pad_ok = cs == C or float(got[..., C:].abs().max()) == 0.0
    g = torch.from_numpy(rng.standard_normal(t + [cs]).astype(np.float32)).to(dev)
    w_l, w_h = ops.idwt_level_bwd(ops.to_channel_first(g, C), frev, d)
    g_l, g_h = ops.idwt_level_cl_bwd(g, C, frev, d)
    s = max(float(w_l.abs().max()), float(w_h.abs().max()), 1e-30)
    e2 = max(float((g_l - w_l).abs().max()), float((g_h - w_h).abs().max())) / s
    worst = max(worst, e1, e2)
    if not (e1 < 3e-6 and e2 < 3e-6 and pad_ok):
        print('FAIL', i, C, d, t, e1, e2, pad_ok, flush=True)
        sys.exit(1)
print('%d cases, worst relative difference %.2e' % (cases, worst))
