"""Synthesis / adjoint wavelet kernels in isolation at the last level of cfg 3 (d=33 -> 64) and cfg 5 (d=65 -> 128), C=32."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from latent_feature_grid_compression_amd import ops
from latent_feature_grid_compression_amd.wavelet_transform.Torch_Wavelet_Transform import WaveletFilter3d
dev = torch.device('cuda:0')
frev = WaveletFilter3d('db2').filter_rev.to(dev)
for C, d, t in ((32, 33, 64), (32, 65, 128)):
    lll = torch.randn(C, d, d, d, device=dev); hf = torch.randn(C, 7, d, d, d, device=dev); g = torch.randn(C, t, t, t, device=dev)
    mb = 4 * C * (8 * d ** 3 + t ** 3) / 1e6
    for name, fn in (('synthesis', lambda: ops.idwt_level(lll, hf, frev, (t, t, t))), ('adjoint', lambda: ops.idwt_level_bwd(g, frev, (d, d, d)))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print('d=%d %-9s %8.1f us  %6.1f MB  %5.2f TB/s' % (d, name, us, mb, mb / us))
