"""One shape of the channel-last level kernels, a few launches each: the program tools/prof_idwt_cl.sh profiles."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from latent_feature_grid_compression_amd import ops
from latent_feature_grid_compression_amd.wavelet_transform.Torch_Wavelet_Transform import WaveletFilter3d
dev = torch.device('cuda:0')
frev = WaveletFilter3d('db2').filter_rev.to(dev)
C, d, t = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 65, None
t = 2 * d - 2
lll = torch.randn(C, d, d, d, device=dev); hf = torch.randn(C, 7, d, d, d, device=dev)
g_cl = torch.randn(t, t, t, C, device=dev)
for _ in range(5):
    ops.idwt_level_cl(lll, hf, frev, (t, t, t))
    ops.idwt_level_cl_bwd(g_cl, C, frev, (d, d, d))
    ops.idwt_level(lll, hf, frev, (t, t, t))
torch.cuda.synchronize()
