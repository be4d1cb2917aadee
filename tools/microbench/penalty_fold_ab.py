"""A/B of the penalty fold (gradients of the coefficient L2 / Smallify L1 sums riding in the decode's adjoint kernels) against
separate penalty launches, on the cfg-3 train step replayed from a HIP graph.  usage: penalty_fold_ab.py <drop_type> <0|1>"""
import subprocess, sys, os
drop, fold = sys.argv[1], sys.argv[2]
env = dict(os.environ, LFGC_NO_PENALTY_FOLD='' if fold == '1' else '1')
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'bench_trainstep.py'), '--graph', '--drop-type', drop],
                     env=env, capture_output=True, text=True).stdout.strip().split('\n')[-1]
print(drop, 'fold' if fold == '1' else 'separate', out[:90])
