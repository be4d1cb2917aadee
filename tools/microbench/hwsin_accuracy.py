import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from latent_feature_grid_compression_amd import _lib
lib = _lib.load()
rng = np.random.default_rng(1)
for rngmax in (1.6, 4.0, 12.6, 40.0):
    xs = rng.uniform(-rngmax, rngmax, 2000000).astype(np.float32)
    x = torch.from_numpy(xs).cuda(); o = torch.empty_like(x)
    assert lib.lfgc_debug_hwsin_f32(x.data_ptr(), x.numel(), o.data_ptr(), None) == 0
    s, c, k = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    lib.lfgc_debug_trig_f32(x.data_ptr(), x.numel(), s.data_ptr(), c.data_ptr(), k.data_ptr(), None)
    torch.cuda.synchronize()
    ref = np.sin(xs.astype(np.float64))
    print('|x|<=%5.1f  hw v_sin max abs err %.3e   polynomial path %.3e' % (rngmax, np.abs(o.cpu().numpy() - ref).max(), np.abs(s.cpu().numpy() - ref).max()))
