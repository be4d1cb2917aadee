"""A/B timing of library variants on the cfg-3 train step (graph replay), one box:
    python tools/ab_forward.py build name=DEFS ...      # builds tools/microbench/ablate/liblfgc_ab_<name>.so
    python tools/ab_trainstep.py                       # on the GPU box
"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.environ.get('LFGC_AB_DIR') or os.path.join(ROOT, 'tools', 'microbench', 'ablate')
names = sorted(f[len('liblfgc_ab_'):-3] for f in os.listdir(OUT) if f.startswith('liblfgc_ab_') and f.endswith('.so'))
res = {n: [] for n in names}
for _ in range(3):
    for n in names:
        env = dict(os.environ, LFGC_LIB_PATH=os.path.join(OUT, 'liblfgc_ab_%s.so' % n))
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'bench_trainstep.py'), '--graph', '--steps', '200', '--warmup', '20'] + sys.argv[1:],
                           env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith('{')]
        res[n].append(json.loads(line[-1])['train_step_graph']['ms_per_step'] if line else float('nan'))
for n in names:
    print('%-24s ms/step min %.4f  all %s' % (n, min(res[n]), ' '.join('%.4f' % v for v in res[n])), flush=True)
