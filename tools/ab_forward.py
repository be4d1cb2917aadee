"""A/B timing of forward-kernel variants on ONE box (devices differ by several per cent, so variants are only ever
compared inside one gpurun call): builds the library once per -D set, then times the headline launch with each,
interleaved over a few rounds.

    python tools/ab_forward.py build name1=DEF1,DEF2 name2=DEF3 ...     # here (hipcc); "name=" = no defines
    python tools/ab_forward.py run [--precision f16x2] [--workload headline] [--rounds 3]   # on the GPU box
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.environ.get('LFGC_AB_DIR') or os.path.join(ROOT, 'tools', 'microbench', 'ablate')

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
    args = sys.argv[2:]
    rounds = int(args[args.index('--rounds') + 1]) if '--rounds' in args else 3
    extra = []
    for k in ('--precision', '--workload'):
        if k in args:
            extra += [k, args[args.index(k) + 1]]
    names = sorted(f[len('liblfgc_ab_'):-3] for f in os.listdir(OUT) if f.startswith('liblfgc_ab_') and f.endswith('.so'))
    res = {n: [] for n in names}
    for _ in range(rounds):
        for n in names:
            env = dict(os.environ, LFGC_LIB_PATH=os.path.join(OUT, 'liblfgc_ab_%s.so' % n))
            r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '10', '--warmup', '3',
                                '--no-cpu-baseline', '--no-check'] + extra, env=env, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith('{')]
            res[n].append(json.loads(line[-1])['roofline']['kernel_ms'] if line else float('nan'))
            if not line:
                print(n, r.stderr[-400:])
    for n in names:
        print('%-28s kernel_ms min %.3f  all %s' % (n, min(res[n]), ' '.join('%.3f' % v for v in res[n])), flush=True)
