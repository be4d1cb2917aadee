"""Diagnostics: per-phase wave cycles of the f16-split forward kernel on the headline launch, from in-kernel s_memtime
stamps (a separate library built with -DLFGC_STAMPS; the shipped library executes no stamp).

    python tools/phase_stamps.py build      # here (hipcc, no GPU needed)
    python tools/phase_stamps.py run        # on the GPU box (via gpurun)
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.environ.get('LFGC_AB_DIR') or os.path.join(ROOT, 'tools', 'microbench', 'ablate')
LIB = os.path.join(OUT, 'liblfgc_stamps.so')

if sys.argv[1] == 'build':
    from latent_feature_grid_compression_amd.build import build_variant
    os.makedirs(OUT, exist_ok=True)
    build_variant(LIB, ['LFGC_STAMPS=1'])
else:
    os.environ['LFGC_LIB_PATH'] = LIB
    import numpy as np
    import torch
    import bench
    from latent_feature_grid_compression_amd import _lib, ops
    dev = torch.device('cuda:0')
    w = bench.WORKLOADS[sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != 'bwd' else 'headline']
    model = bench.build_model(w, 1234, dev)
    lib = _lib.load()
    if len(sys.argv) > 2 and sys.argv[2] == 'bwd':
        # backward data kernel at the cfg-3 train-step batch
        n = 32768
        nslots = 256 * 8
        buf = torch.zeros(nslots * 20, dtype=torch.int64, device=dev)
        lib.lfgc_debug_set_bwd_stamp_buffer.argtypes = [ctypes.c_void_p]
        lib.lfgc_debug_set_bwd_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
        model = bench.build_model(bench.WORKLOADS['headline'], 1234, dev).train()
        for _ in range(3):
            pos = (torch.rand(n, 3, device=dev) * 2 - 1).requires_grad_(True)
            model.zero_grad()
            model(pos).square().mean().backward()
        torch.cuda.synchronize()
        s = buf.cpu().numpy().reshape(nslots, 20).astype(np.float64)
        s = s[s[:, 16] > 0]
        names = {0: "head dH + loop", 1: "snake' (stash loads), dstash stores, scale, split", 2: 'wait for stores / weight DMA + barrier',
                 3: 'MFMAs + scale-back', 4: 'geometry, staging, atomic scatter', 5: 'd_pos'}
        tot = s[:, 16].mean()
        print('bwd data kernel: waves %d, wave lifetime %.0f cycles' % (len(s), tot))
        for k in sorted(names):
            print('  %-52s %9.0f cycles  %5.1f %%' % (names[k], s[:, k].mean(), 100 * s[:, k].mean() / tot))
        sys.exit(0)
    nslots = 256 * 8
    buf = torch.zeros(nslots * 20, dtype=torch.int64, device=dev)
    lib.lfgc_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
    lib.lfgc_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
    train = len(sys.argv) > 3 and sys.argv[3] == 'train'      # `run headline train`: the cfg-3 train-step batch, stash written
    with torch.no_grad():
        grid, packed = model._decoded_channel_last(), model._packed()
        res = (w['vol'],) * 3
        for _ in range(3):
            if train:
                pos = torch.rand(32768, 3, device=dev) * 2 - 1
                y, _ = ops.forward_raw(model._descriptor(), grid, packed, pos=pos, clamp=False, want_stash=True)
            else:
                y, _ = ops.forward_raw(model._descriptor(), grid, packed, lattice=(res, 0, w['vol'], 32), clamp=True)
        torch.cuda.synchronize()
    s = buf.cpu().numpy().reshape(nslots, 20).astype(np.float64)
    s = s[s[:, 16] > 0]
    ntiles = (32768 if train else w['vol'] ** 3) / 32 / len(s)
    names = {0: 'loop/tail+store', 1: 'wait for own weight-DMA pieces', 2: 'inputs issue (positions, corner loads)',
             3: 'barrier before layer 0', 4: 'layer 0 (+ embed, interpolate, split)', 5: 'barrier before layer 1', 6: 'layer 1',
             7: 'barrier before layer 2', 8: 'layer 2', 9: 'barrier before layer 3', 10: 'layer 3', 11: 'barrier before layer 4',
             12: 'layer 4', 13: 'barrier before layers 5+', 14: 'last layer + head'}
    tot = s[:, 16].mean()
    print('waves %d, tiles per wave %.1f, wave lifetime %.0f cycles, in-kernel clock %.3f GHz' %
          (len(s), ntiles, tot, (s[:, 16] / (s[:, 17] * 10e-9)).mean() * 1e-9))
    for k in sorted(names):
        v = s[:, k].mean()
        if v > 0:
            print('  %-32s %9.0f cycles per tile  %5.1f %%' % (names[k], v / ntiles, 100 * v / tot))
    print('  %-32s %9.0f cycles per tile' % ('sum', s[:, :16].sum(axis=1).mean() / ntiles))
