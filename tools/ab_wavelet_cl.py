"""A/B timing of channel-last level kernel variants on ONE box: builds the library once per -D set, then runs
tools/microbench/idwt_sizes.py with each.
    python tools/ab_wavelet_cl.py build name1=DEF1,DEF2 name2= ...      # here (hipcc)
    python tools/ab_wavelet_cl.py run                                    # on the GPU box
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'tools', 'microbench', 'ablate')

if sys.argv[1] == 'build':
    from latent_feature_grid_compression_amd.build import build_variant
    os.makedirs(OUT, exist_ok=True)
    for f in os.listdir(OUT):
        if f.startswith('liblfgc_ab_') and f.endswith('.so'):
            os.remove(os.path.join(OUT, f))
    for spec in sys.argv[2:]:
        name, _, defs = spec.partition('=')
        build_variant(os.path.join(OUT, 'liblfgc_ab_%s.so' % name), [d for d in defs.split(',') if d])
else:
    names = sorted(f[len('liblfgc_ab_'):-3] for f in os.listdir(OUT) if f.startswith('liblfgc_ab_') and f.endswith('.so'))
    for n in names:
        env = dict(os.environ, LFGC_LIB_PATH=os.path.join(OUT, 'liblfgc_ab_%s.so' % n))
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'microbench', 'idwt_sizes.py')], env=env, capture_output=True, text=True)
        print('== ' + n, flush=True)
        print('\n'.join(l for l in r.stdout.splitlines() if 'channel-last' in l) or r.stderr[-400:], flush=True)
