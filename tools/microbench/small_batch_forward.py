"""Forward kernel at the train-step batch (32 768 samples = one 32-sample tile per SIMD): with and without the stash."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from latent_feature_grid_compression_amd import ops
dev = torch.device('cuda:0')
m = bench.build_model(bench.WORKLOADS['headline'], 2003, dev)
grid, packed, desc = m._decoded_channel_last(), m._packed(), m._descriptor()
for n in (32768, 65536, 262144, 1048576):
    pos = torch.rand(n, 3, device=dev) * 2 - 1
    for stash in (False, True):
        for prec in ('f16x2', 'f16'):
            for _ in range(5):
                ops.forward_raw(desc, grid, packed, pos=pos, want_stash=stash, precision=prec)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.forward_raw(desc, grid, packed, pos=pos, want_stash=stash, precision=prec)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            print('n=%8d stash=%-5s %-5s %8.1f us  %7.1f Msamples/s' % (n, stash, prec, us, n / us))
