"""Fixture for data/Interpolation.py::finite_difference_trilinear_grad of the REFERENCE (imported, not copied):
python tools/make_goldens_fdgrad.py -> tests/golden/gt_fd_grad.npz"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _ref_standins
_ref_standins.install()
from data.Interpolation import finite_difference_trilinear_grad          # noqa: E402

rng = np.random.Generator(np.random.PCG64(4242))
vol = torch.from_numpy(rng.uniform(-1, 1, (20, 21, 22)).astype(np.float32))
n = 3000
p = np.stack([rng.uniform(0, 19, n), rng.uniform(0, 20, n), rng.uniform(0, 21, n)], 1).astype(np.float32)
p[:500] = np.round(p[:500])                      # lattice points
p[500:520, 0] = 0.0; p[520:540, 1] = 20.0; p[540:560, 2] = 21.0; p[560:570] = 0.0      # on the faces: clamped steps
p = torch.from_numpy(p)
mn, mx, rs = torch.zeros(3), torch.tensor([19.0, 20.0, 21.0]), torch.tensor([20.0, 21.0, 22.0])
sc = torch.tensor([19.0 / 21.0, 20.0 / 21.0, 1.0])
out = {'vol': vol.numpy(), 'p': p.numpy(), 'min_bb': mn.numpy(), 'max_bb': mx.numpy(), 'res': rs.numpy(), 'scale': sc.numpy(),
       'grad': finite_difference_trilinear_grad(p, vol, mn, mx, rs).numpy(),
       'grad_scaled': finite_difference_trilinear_grad(p, vol, mn, mx, rs, scale=sc).numpy()}
np.savez_compressed(os.path.join(_ref_standins.GOLD, 'gt_fd_grad.npz'), **out)
print({k: v.shape for k, v in out.items()})
