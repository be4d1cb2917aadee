"""Diagnostics: where does the forward kernel's time go?  Builds ablated copies of the library
(-DLFGC_ABLATE=mask, see csrc/lfgc_forward.h) in the build container and times the headline launch with each on
the GPU box.  Outputs of ablated builds are wrong by construction; only the timings matter.

    python tools/ablate_forward.py build      # here (hipcc, no GPU needed)
    python tools/ablate_forward.py run        # on the GPU box (via gpurun)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'tools', 'microbench', 'ablate')
MASKS = {'full': 0, 'no_gather_loads': 1, 'no_stream_no_barrier': 8, 'neither': 9}
FLAG_VARIANTS = {}

if sys.argv[1] == 'build':
    from latent_feature_grid_compression_amd.build import build_variant
    os.makedirs(OUT, exist_ok=True)
    for name, mask in MASKS.items():
        build_variant(os.path.join(OUT, 'liblfgc_%s.so' % name), ['LFGC_ABLATE=%d' % mask])
    for name, flags in FLAG_VARIANTS.items():
        build_variant(os.path.join(OUT, 'liblfgc_%s.so' % name), ['LFGC_ABLATE=0'], flags=flags)
elif sys.argv[1] == 'run':
    res = {}
    for name in list(MASKS) + list(FLAG_VARIANTS):
        env = dict(os.environ, LFGC_LIB_PATH=os.path.join(OUT, 'liblfgc_%s.so' % name))
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '5', '--warmup', '2',
                            '--no-cpu-baseline', '--no-check'], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith('{')]
        res[name] = json.loads(line[-1])['roofline']['kernel_ms'] if line else r.stderr[-300:]
    print(json.dumps(res))
