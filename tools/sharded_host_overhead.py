"""Host overhead of the sharded full-volume driver on the slab an 8-rank run gives one rank.

At 8 ranks on 256^3 a rank owns ONE tile plane: 32 x-rows = 2 097 152 samples ~ 0.7 ms of kernel, issued as `chunks`
fused launches each followed by its own asynchronous collective, all from Python.  Is the host visible against that?
Measured through RCCL with one rank (world size 1, always_gather: the collective path runs for real), the dataset cut down
to the 32-row slab so that this rank's share is exactly an 8-rank share; gather='root' (what bench.py times).

    python tools/sharded_host_overhead.py          # on the GPU box
Prints, per piece count: wall time per step (host clock around the call + a device sync), the summed device time of the
fused launches (HIP events), and the host's share.
"""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                             # noqa: E402
from latent_feature_grid_compression_amd.data.IndexDataset import IndexDataset          # noqa: E402
from latent_feature_grid_compression_amd.visualization import OutputToVTK as V          # noqa: E402
from latent_feature_grid_compression_amd import ops                       # noqa: E402


class SlabOf256(IndexDataset):
    """A (32, 256, 256) output volume whose positions are those of x-rows [96, 128) of the 256^3 lattice."""
    pass


def main():
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    w = bench.WORKLOADS['headline']
    model = bench.build_model(w, 2003, dev)
    full = IndexDataset((256, 256, 256), 16, build_index_table=False)
    res = full.vol_res_touple
    slab = IndexDataset((32, 256, 256), 16, build_index_table=False)     # the driver only needs its shape for the cuts
    x0 = 96

    def slab_fn(b, e, out_view):
        with torch.no_grad():
            ops.forward_raw(model._descriptor(), model._decoded_channel_last(), model._packed(), pos=None,
                            lattice=(res, x0 + b, x0 + e, 32), clamp=True, out=out_view.view(-1), precision=model.precision)

    def step(chunks, tm=None):
        model._grid_cache = None
        model._pack_cache = None
        return V.reconstruct_volume_sharded(slab, model, 32, slab_fn=slab_fn, device=dev, gather='root', chunks=chunks,
                                            always_gather=True, timings=tm)

    print('slab of one of 8 ranks on 256^3: 32 x-rows = %d samples; RCCL world size %d' % (32 * 256 * 256, dist.get_world_size()))
    step(1)                                  # first collective of the process: communicator set-up
    for chunks in (4, 2, 1, None):          # None = the driver's default (pieces of whole tile planes: 1 here)
        for _ in range(5):
            step(chunks)
        torch.cuda.synchronize()
        n = 50
        t0 = time.perf_counter()
        for _ in range(n):
            step(chunks)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / n * 1e3
        # device time of the launches alone (events inside the driver; the extra synchronisation of `timings` is why this is
        # a separate loop)
        comp = []
        for _ in range(20):
            tm = {}
            step(chunks, tm)
            comp.append(tm['compute_ms'])
        comp = sum(comp) / len(comp)
        # decode + pack (repeated every step like bench.py does) is inside the first slab_fn call, hence inside compute_ms
        print('chunks=%s: wall %.3f ms/step, fused launches + decode/pack (device, events) %.3f ms, host-visible remainder %.3f ms = %.0f %% of wall'
              % ('default' if chunks is None else chunks, wall, comp, wall - comp, 100 * (wall - comp) / wall), flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
