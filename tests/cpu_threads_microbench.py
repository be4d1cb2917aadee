import os, sys, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from oracle import ref_torch as R
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
sm = R.synth_model(32, 64, 128, 4, seed=1)
dense = R.decode_volume(sm['coeffs'], sm['shape_array'], sm['filter_rev'])
pos = torch.rand(32768, 3) * 2 - 1
for t in (4, 8, 16, 32, 64, 128):
    torch.set_num_threads(t)
    with torch.no_grad():
        R.forward_from_grid(dense, sm['weights'], sm['biases'], pos, 2)
        t0 = time.perf_counter()
        for _ in range(3): R.forward_from_grid(dense, sm['weights'], sm['biases'], pos, 2)
        dt = (time.perf_counter() - t0) / 3
    print('threads %3d: %.1f ms per 32768 samples = %.3f Msamples/s' % (t, dt * 1e3, 32768 / dt / 1e6))
