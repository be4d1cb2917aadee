"""Last wavelet level in isolation at cfg 3 (d=33 -> 64) and cfg 5 (d=65 -> 128), C=32: the channel-first level kernels,
the layout conversions, and the fused channel-last level kernels that replace each pair.  MB = coefficient + grid bytes
(each touched once); the two-kernel form moves the grid three times."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from latent_feature_grid_compression_amd import ops
from latent_feature_grid_compression_amd.wavelet_transform.Torch_Wavelet_Transform import WaveletFilter3d
dev = torch.device('cuda:0')
frev = WaveletFilter3d('db2').filter_rev.to(dev)


def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for C, d, t in ((32, 33, 64), (32, 65, 128)):
    lll = torch.randn(C, d, d, d, device=dev); hf = torch.randn(C, 7, d, d, d, device=dev); g = torch.randn(C, t, t, t, device=dev)
    g_cl = ops.to_channel_last(g)
    mb = 4 * C * (8 * d ** 3 + t ** 3) / 1e6
    rows = (('synthesis (channel-first)', lambda: ops.idwt_level(lll, hf, frev, (t, t, t))),
            ('to_channel_last', lambda: ops.to_channel_last(g)),
            ('synthesis channel-last', lambda: ops.idwt_level_cl(lll, hf, frev, (t, t, t))),
            ('adjoint (channel-first)', lambda: ops.idwt_level_bwd(g, frev, (d, d, d))),
            ('to_channel_first', lambda: ops.to_channel_first(g_cl, C)),
            ('adjoint channel-last', lambda: ops.idwt_level_cl_bwd(g_cl, C, frev, (d, d, d))))
    for name, fn in rows:
        us = timed(fn)
        print('d=%d %-26s %8.1f us  %6.1f MB  %5.2f TB/s' % (d, name, us, mb, mb / us))
