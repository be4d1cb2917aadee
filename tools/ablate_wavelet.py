"""Where the synthesis (IDWT) kernel's time goes at the cfg-3 last level (C=32, d=33 -> t=64): diagnostics builds of the
library with parts of the kernel removed (LFGC_WAVELET_ABLATE: 1 no output stores, 2 no arithmetic, 4 no staging loads).

    python tools/ablate_wavelet.py            (builds the variants here, CPU)   then on the GPU box:
    python tools/ablate_wavelet.py --run
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'tools', 'ablate_out')      # git-ignored; travels to the GPU box with the snapshot
VARIANTS = {'full': 0, 'no_store': 1, 'no_math': 2, 'no_load': 4, 'no_store_no_math': 3, 'no_load_no_math': 6, 'shell': 7}

if '--run' not in sys.argv:
    from latent_feature_grid_compression_amd.build import build_variant
    os.makedirs(OUT, exist_ok=True)
    for name, code in VARIANTS.items():
        build_variant(os.path.join(OUT, 'liblfgc_w_%s.so' % name), ['LFGC_WAVELET_ABLATE=%d' % code])
    sys.exit(0)

if '--one' in sys.argv:
    import torch
    from latent_feature_grid_compression_amd import ops
    from latent_feature_grid_compression_amd.wavelet_transform.Torch_Wavelet_Transform import WaveletFilter3d
    dev = torch.device('cuda:0')
    frev = WaveletFilter3d('db2').filter_rev.to(dev)
    C, d, t = 32, 33, 64
    lll = torch.randn(C, d, d, d, device=dev); hf = torch.randn(C, 7, d, d, d, device=dev)
    for _ in range(5):
        ops.idwt_level(lll, hf, frev, (t, t, t))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.idwt_level(lll, hf, frev, (t, t, t))
    e1.record(); torch.cuda.synchronize()
    print('%.1f us' % (e0.elapsed_time(e1) / 50 * 1e3))
    sys.exit(0)

for name in VARIANTS:
    env = dict(os.environ, LFGC_LIB_PATH=os.path.join(OUT, 'liblfgc_w_%s.so' % name))
    r = subprocess.run([sys.executable, os.path.abspath(__file__), '--run', '--one'], env=env, capture_output=True, text=True)
    print('%-18s %s' % (name, r.stdout.strip().split('\n')[-1] if r.stdout.strip() else r.stderr[-300:]))
