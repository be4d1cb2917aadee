#!/usr/bin/env python3
"""bench.py -- headline benchmark of the latent-feature-grid sample/decode hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one complete full-volume reconstruction pass of the hot path (BASELINE.json north_star /
metric: "Msamples/s (grid-interp+embed+MLP fwd) on 256^3 volume"): inverse-wavelet decode of the latent
grid (once per pass, as an inference call does), parameter packing, then the fused trilinear-sample +
Fourier-embed + 4-layer SnakeAlt MLP forward over every voxel of a 256^3 lattice (512 tiles of 32^3;
positions generated per tile on device), fp32.  Model = BASELINE configs[2]/[3] shape (64^3 x 32-channel
grid, 4-level db2 wavelet code, MLP 4 x 128), synthetic random-init parameters (no datasets offline).
With N > 1 ranks the tile lattice is cut into contiguous x-slabs (strong scaling: the volume is fixed)
and the output volume is assembled on rank 0 with an RCCL gather over xGMI (`--gather all`: on every rank);
the other assembly mode, per-rank kernel / decode+pack / gather-wait times and the world size RCCL saw are
reported in `multi_gpu`.

Rank 0 prints one JSON line.  `roofline` is for the dominant kernel (lfgc_fwd16_kernel): algorithmic fp32
MLP FLOPs per launch / its average launch duration (HIP events on the launch stream) against the dense
peak of the matrix pipe the build runs on (2.5 PFLOP/s f16 for the default build; the MFMA work actually
executed -- three f16 products per fp32 product -- is `executed_mfma`); the HBM-side figure on the
algorithmic gather bytes is `hbm_algorithmic`, the measured HBM traffic per launch `traffic`.
`cpu_baseline` times the oracle's op-for-op PyTorch restatement of the reference on the host cores for a
bounded sample of the same workload (rank 0, N = 1 only).  `extra` (N = 1) puts the other BASELINE
configs under the same clock: cfg 2 and cfg 5 forward + decode, the cfg-3 train step (default and reduced).
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, dense f32-input MFMA
F16_MFMA_PEAK_TFLOPS = 2500.0      # same guide, dense BF16/F16 MFMA
HBM_PEAK_GBS = 8000.0              # same guide, HBM3E spec


def traffic_bytes(workload, precision):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (separate --pmc runs for
    FETCH_SIZE and WRITE_SIZE; bytes = 2 x FETCH_SIZE KB (gfx950 correction for 16-B/lane reads) + WRITE_SIZE KB), or
    None when no profile of this (workload, build) is committed.  Newest round first."""
    for rnd in ('r3', 'r2', 'r1'):
        path = os.path.join(ROOT, 'profiles', rnd, 'bench_headline_pmc.json')
        try:
            with open(path) as fh:
                hbm = json.load(fh).get('hbm', {})
        except (OSError, ValueError):
            continue
        if workload == 'headline' and precision in hbm:
            return hbm[precision].get('hbm_bytes_per_launch')
    return None

def profiled_counters(workload, precision):
    """Counter-derived figures of the dominant kernel from the committed PMC passes (None without a profile of this
    (workload, build)): matrix-pipe occupancy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), and the
    instruction mix per 32-sample tile."""
    if workload != 'headline':
        return None
    for rnd in ('r3', 'r2'):
        try:
            with open(os.path.join(ROOT, 'profiles', rnd, 'bench_headline_pmc.json')) as fh:
                c = json.load(fh).get('sq_counters', {}).get(precision, {})
        except (OSError, ValueError):
            continue
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'GRBM_GUI_ACTIVE' in c:
            tiles = 16777216 / 32
            return {'source': 'profiles/%s/bench_headline_pmc.json' % rnd,
                    'mfma_pipe_busy_frac_of_simd_cycles': c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['GRBM_GUI_ACTIVE'] / 8 * 1024),
                    'per_tile': {k: c[n] / tiles for k, n in (('valu', 'SQ_INSTS_VALU'), ('mfma', 'SQ_INSTS_MFMA'),
                                                             ('lds', 'SQ_INSTS_LDS'), ('salu', 'SQ_INSTS_SALU')) if n in c}}
    return None


WORKLOADS = {
    # name: (volume edge, grid channels C, grid edge G, hidden H, layers L)
    'headline': dict(vol=256, C=32, G=64, H=128, L=4,
                     desc='256^3 full-volume reconstruction, 64^3x32ch grid (4-level db2), MLP 4x128, fp32'),
    'cfg5': dict(vol=1024, C=32, G=128, H=128, L=4, levels=3,
                 desc='1024^3 full-volume reconstruction, 128^3x32ch grid (3-level db2), MLP 4x128, fp32'),
    'cfg5_l5': dict(vol=1024, C=32, G=128, H=128, L=4, levels=5,
                    desc='1024^3 full-volume reconstruction, 128^3x32ch grid (5-level db2 = the reference default), MLP 4x128, fp32'),
    'cfg2': dict(vol=150, C=16, G=32, H=64, L=4,
                 desc='150^3 full-volume reconstruction, 32^3x16ch grid (3-level db2), MLP 4x64, fp32'),
}


def build_model(w, seed, device):
    """Random-init model of the named architecture: dense grid U(0,1) (model/model_utils.py:27-28) DWT-encoded
    by the constructor, Linear layers U(+-1/sqrt(fan_in)); numpy PCG64 so every rank builds identical weights."""
    from latent_feature_grid_compression_amd.model.Feature_Grid_Model import Feature_Grid_Model
    from latent_feature_grid_compression_amd.model.Feature_Embedding import FourierEmbedding
    from latent_feature_grid_compression_amd.wavelet_transform.Torch_Wavelet_Transform import WaveletFilter3d
    rng = np.random.Generator(np.random.PCG64(seed))
    grid = torch.from_numpy(rng.random((w['C'], w['G'], w['G'], w['G']), dtype=np.float32)).to(device)
    model = Feature_Grid_Model(FourierEmbedding(2, 3), grid, None, WaveletFilter3d('db2').to(device),
                               hidden_channel=w['H'], num_layer=w['L'], num_levels=w.get('levels'))
    with torch.no_grad():
        for lin in list(model.net_layers) + [model.final_layer]:
            bound = 1.0 / math.sqrt(lin.in_features)
            lin.weight.copy_(torch.from_numpy(rng.uniform(-bound, bound, tuple(lin.weight.shape)).astype(np.float32)))
            lin.bias.copy_(torch.from_numpy(rng.uniform(-bound, bound, tuple(lin.bias.shape)).astype(np.float32)))
    return model.to(device).eval()


def cpu_baseline(model, w, budget_s=20.0, hip_volume=None):
    """Oracle (op-for-op torch restatement of the reference) on the host cores, bounded sample of the workload.  With
    ``hip_volume`` (the volume the timed HIP path produced) the oracle's tiles also serve as the in-run parity check."""
    from oracle import ref_torch as R
    # the GPU box gives one-GPU jobs a 16-CPU share of its 256 hardware threads; torch oversubscribed to 256 threads
    # runs this path 70x slower (tests/cpu_threads_microbench.py: 4..32 threads all give ~0.5 Msamples/s)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    coeffs = [p.detach().cpu() for p in model.feature_grid]
    layers = list(model.net_layers) + [model.final_layer]
    weights = [l.weight.detach().cpu() for l in layers]
    biases = [l.bias.detach().cpu() for l in layers]
    frev = model.filter.filter_rev.detach().cpu()
    ds = R.VolumeIndexing((w['vol'],) * 3)
    tiles = list(R.tile_iter(ds.vol_res_touple, 32))
    with torch.no_grad():
        t0 = time.perf_counter()
        dense = R.decode_volume(coeffs, model.shape_array, frev)
        t_decode = time.perf_counter() - t0
        # warm-up tile
        R.forward_from_grid(dense, weights, biases, R.tile_positions(ds, tiles[0]).reshape(-1, 3), 2)
        n_samples, n_tiles, t_fwd = 0, 0, 0.0
        max_err, max_ref, sq_err, gt_min, gt_max = 0.0, 0.0, 0.0, float('inf'), float('-inf')
        for b in tiles:
            pos = R.tile_positions(ds, b).reshape(-1, 3)
            t0 = time.perf_counter()
            y = R.forward_from_grid(dense, weights, biases, pos, 2).clamp(-1, 1)
            t_fwd += time.perf_counter() - t0
            n_samples += pos.shape[0]
            n_tiles += 1
            if hip_volume is not None:
                x0, x1, y0, y1, z0, z1 = b
                mine = hip_volume[x0:x1, y0:y1, z0:z1].cpu().reshape(-1).double()
                ref = y.reshape(-1).double()
                max_err = max(max_err, float((mine - ref).abs().max()))
                max_ref = max(max_ref, float(ref.abs().max()))
                sq_err += float(((mine - ref) ** 2).sum())
                gt_min, gt_max = min(gt_min, float(ref.min())), max(gt_max, float(ref.max()))
            if t_fwd > budget_s or n_tiles >= 64:
                break
    parity = None
    if hip_volume is not None:
        mse = sq_err / n_samples
        parity = {'max_rel_err_vs_oracle': max_err / max_ref, 'tolerance': 1e-5,
                  'psnr_of_hip_vs_oracle_dB': (10.0 * math.log10((gt_max - gt_min) ** 2 / mse)) if mse > 0 else float('inf'),
                  'samples': n_samples}
    cpu_model = 'unknown'
    try:
        with open('/proc/cpuinfo') as fh:
            for line in fh:
                if line.startswith('model name'):
                    cpu_model = line.split(':', 1)[1].strip()
                    break
    except OSError:
        pass
    return {
        'parity': parity, 'cpu_model': cpu_model, 'host_threads_visible': os.cpu_count(),
        'value': n_samples / t_fwd / 1e6, 'unit': 'Msamples/s', 'cores': torch.get_num_threads(), 'kind': 'port',
        # the reference decodes the whole grid again for every 32^3 tile (model/Feature_Grid_Model.py:54): its actual rate
        'value_with_per_tile_decode': n_samples / (t_fwd + n_tiles * t_decode) / 1e6,
        'decode_s': t_decode,
        'sample': '%d tiles of 32^3 (%d samples) of the same lattice; value = grid decoded once (%.3f s, not included), '
                  'value_with_per_tile_decode = the reference\'s behaviour (decode per tile)' % (n_tiles, n_samples, t_decode),
    }


def _event_ms(fn, reps):
    """Average device time of fn() over reps calls, HIP events on the current stream."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def oracle_parts(model):
    """Host copies of everything the oracle's functional restatement takes."""
    layers = list(model.net_layers) + [model.final_layer]
    return ([p.detach().cpu() for p in model.feature_grid], [l.weight.detach().cpu() for l in layers],
            [l.bias.detach().cpu() for l in layers], model.filter.filter_rev.detach().cpu())


def parity_stats(mine, ref):
    """north_star's figures for one block of outputs: max|y - y_ref| / max|y_ref| and the PSNR of y against y_ref."""
    mine, ref = mine.double().reshape(-1), ref.double().reshape(-1)
    mse = float(((mine - ref) ** 2).mean())
    rng_ = float(ref.max() - ref.min())
    return {'max_rel_err_vs_oracle': float((mine - ref).abs().max() / ref.abs().max()), 'tolerance': 1e-5,
            'psnr_of_hip_vs_oracle_dB': (10.0 * math.log10(rng_ ** 2 / mse)) if mse > 0 else float('inf'),
            'samples': int(ref.numel())}


def forward_extra(name, device, precision='f16x2', reps=3, check_tiles=0, random_batch=0):
    """One more BASELINE shape under the same clock: full-volume fused forward of workload `name` (+ its decode).
    check_tiles > 0: in-run parity + PSNR of that many 32^3 tiles of the 256^3 sub-block at the origin against the oracle
    (cfg 5).  random_batch > 0: also a uniform-random U(-1,1)^3 batch of that many samples through the
    position-list entry (cfg 2), timed and checked against the oracle on its first 65 536 samples."""
    from latent_feature_grid_compression_amd import ops
    w = WORKLOADS[name]
    model = build_model(w, seed=2003, device=device)
    model.precision = precision
    res = (w['vol'],) * 3
    n = w['vol'] ** 3
    out = torch.empty(n, dtype=torch.float32, device=device)
    with torch.no_grad():
        def decode():
            model._grid_cache = None
            return model._decoded_channel_last()
        grid = decode()
        packed = model._packed()
        fwd = lambda: ops.forward_raw(model._descriptor(), grid, packed, lattice=(res, 0, w['vol'], 32), clamp=True, out=out,
                                      precision=precision)
        fwd()
        torch.cuda.synchronize()
        k_ms = _event_ms(fwd, reps)
        d_ms = _event_ms(decode, reps)
    assert bool(torch.isfinite(out[::4097]).all())
    bps = 12 + 4 + 8 * w['C'] * 4
    K0 = 3 + 12 + w['C']
    fl = 2 * (K0 * w['H'] + (w['L'] - 1) * w['H'] ** 2 + w['H'])
    r = {'workload': w['desc'], 'precision': precision, 'samples': n, 'kernel_ms': k_ms, 'decode_ms': d_ms,
         'wavelet_levels': len(model.feature_grid) - 1,
         'value': n / (k_ms * 1e-3) / 1e6, 'unit': 'Msamples/s (forward launch only; decode_ms beside it)',
         'algorithmic_TFLOPs': fl * n / (k_ms * 1e-3) / 1e12,
         'hbm_algorithmic': {'bytes_per_sample': bps, 'achieved_GBs': bps * n / (k_ms * 1e-3) / 1e9,
                             'frac': bps * n / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}}
    if check_tiles or random_batch:
        from oracle import ref_torch as R
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        coeffs, weights, biases, frev = oracle_parts(model)
        with torch.no_grad():
            dense = R.decode_volume(coeffs, model.shape_array, frev)
            if check_tiles:
                ds = R.VolumeIndexing(res)
                sub = min(256, w['vol'])
                tiles = [b for b in R.tile_iter(ds.vol_res_touple, 32) if b[1] <= sub and b[3] <= sub and b[5] <= sub]
                step = max(1, len(tiles) // check_tiles)
                vol = out.view(res)
                mine, ref = [], []
                for b in tiles[::step][:check_tiles]:
                    ref.append(R.forward_from_grid(dense, weights, biases, R.tile_positions(ds, b).reshape(-1, 3), 2).clamp(-1, 1).reshape(-1))
                    mine.append(vol[b[0]:b[1], b[2]:b[3], b[4]:b[5]].reshape(-1).cpu())
                r['parity_256_subblock'] = dict(parity_stats(torch.cat(mine), torch.cat(ref)), tiles=len(ref),
                                                block='32^3 tiles spread over the 256^3 sub-block at the origin')
            if random_batch:
                rng = np.random.Generator(np.random.PCG64(3002))
                pos = torch.from_numpy(rng.uniform(-1, 1, (random_batch, 3)).astype(np.float32)).to(device)
                yb = torch.empty(random_batch, dtype=torch.float32, device=device)
                fb = lambda: ops.forward_raw(model._descriptor(), grid, packed, pos=pos, clamp=False, out=yb, precision=precision)
                fb()
                torch.cuda.synchronize()
                b_ms = _event_ms(fb, reps)
                nchk = min(65536, random_batch)
                yr = R.forward_from_grid(dense, weights, biases, pos[:nchk].cpu(), 2).reshape(-1)
                r['random_batch'] = dict(parity_stats(yb[:nchk].cpu(), yr), batch=random_batch, kernel_ms=b_ms,
                                         value=random_batch / (b_ms * 1e-3) / 1e6, unit='Msamples/s',
                                         positions='U(-1,1)^3, numpy PCG64(3002); parity on the first %d samples' % nchk)
                del pos, yb
    del out, grid, model
    torch.cuda.empty_cache()
    return r


def smooth_volume(shape, seed, device):
    """SURVEY 8(d), cfg 3: a smooth synthetic volume -- 8 random low-frequency sinusoids + 0.05 x uniform noise, min-max
    normalised to [-1, 1] like data/IndexDataset.py:15-17 -- so that loss and PSNR of a train run mean something."""
    rng = np.random.Generator(np.random.PCG64(seed))
    ax = [torch.linspace(0.0, 1.0, n, device=device, dtype=torch.float32) for n in shape]
    X, Y, Z = torch.meshgrid(*ax, indexing='ij')
    vol = torch.zeros(shape, dtype=torch.float32, device=device)
    for _ in range(8):
        f = rng.uniform(0.5, 3.0, 3) * rng.choice([-1.0, 1.0], 3)
        ph, amp = rng.uniform(0, 2 * math.pi), rng.uniform(0.5, 1.0)
        vol += float(amp) * torch.sin(2 * math.pi * (float(f[0]) * X + float(f[1]) * Y + float(f[2]) * Z) + float(ph))
    vol += 0.05 * (torch.from_numpy(rng.random(shape, dtype=np.float32)).to(device) * 2 - 1)
    lo, hi = vol.min(), vol.max()
    return (vol - lo) / (hi - lo) * 2 - 1


def cfg3_train_setup(device, precision='f16x2', seed=2003, volume='smooth', n=2048 * 16, vol_shape=(255, 255, 255), workload=None):
    """The BASELINE config-3 train step (training/training.py:95-138) as bench.py times it and tests/ replays it: model,
    volume, on-device lattice sampler (Philox stream `1003`, counter on the device), torch fused capturable Adam(lr 0.008)
    and `step()` = draw -> forward -> fused ground truth + MSE -> backward -> Adam.  Every launch of `step` is
    stream-ordered and allocation-free after the first call, so it can be captured in one HIP graph."""
    from latent_feature_grid_compression_amd.data.Interpolation import trilinear_mse_loss, mse_unit_grad
    from latent_feature_grid_compression_amd.data.IndexDataset import IndexDataset
    w = workload or WORKLOADS['headline']
    model = build_model(w, seed=seed, device=device).train()
    model.precision = precision
    if volume == 'smooth':
        vol = smooth_volume(vol_shape, 1003, device)
    else:
        rng = np.random.Generator(np.random.PCG64(1003))
        vol = torch.from_numpy(rng.uniform(-1, 1, vol_shape).astype(np.float32)).to(device)
    ds = IndexDataset(vol_shape, 16, build_index_table=False)
    opt = torch.optim.Adam(model.parameters(), lr=0.008, capturable=True, fused=True)
    mn_h, mx_h, rs_h = ds.min_idx.tolist(), ds.max_idx.tolist(), ds.vol_res.tolist()
    ds.min_idx, ds.max_idx, ds.scales = ds.min_idx.to(device), ds.max_idx.to(device), ds.scales.to(device)
    unit = mse_unit_grad(device)                       # created outside any capture
    ctx = {'model': model, 'vol': vol, 'ds': ds, 'opt': opt, 'n': n, 'bounds': (mn_h, mx_h, rs_h), 'seed': 1003}

    def step():
        raw, norm = ds.sample_positions(n, device, seed=1003)      # draw + positions: one kernel, counter on the device
        norm.requires_grad = True                      # the reference sets requires_grad on positions (training.py:99)
        opt.zero_grad()
        loss = trilinear_mse_loss(model(norm).squeeze(-1), raw, vol, mn_h, mx_h, rs_h)
        loss.backward(unit)
        opt.step()
        ctx['last_batch'] = (raw, norm)
        return loss

    ctx['step'] = step
    return ctx


def capture_train_step(ctx, eager_warmup=3):
    """`eager_warmup` eager steps on a side stream (allocator + Adam state warm), then ONE step captured in a HIP graph.
    Returns (graph, loss tensor the replays overwrite, [losses of the eager steps])."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    eager = []
    with torch.cuda.stream(side):
        for _ in range(eager_warmup):
            eager.append(ctx['step']().detach().clone())
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    ctx['opt'].zero_grad(set_to_none=True)
    with torch.cuda.graph(graph):
        loss = ctx['step']()
    return graph, loss, eager


def train_step_extra(device, precision='f16x2', steps=100, warmup=10):
    """BASELINE config 3 train step: 32 768 lattice samples of a smooth synthetic 255^3 volume -> forward -> ground truth +
    MSE -> backward -> Adam(lr 0.008, torch fused), the whole step captured in one HIP graph and replayed.  Reports the
    loss of the first step (untrained model) and of the last one, and the PSNR of the full reconstructed volume after the
    run: the optimizer has to have learnt something for these to move."""
    from latent_feature_grid_compression_amd.visualization import OutputToVTK as V
    ctx = cfg3_train_setup(device, precision)
    graph, loss, eager = capture_train_step(ctx)
    for _ in range(warmup):
        graph.replay()
    torch.cuda.synchronize()
    ms = _event_ms(graph.replay, steps)
    loss_first, loss_last = float(eager[0]), float(loss.detach())
    model, vol, n = ctx['model'], ctx['vol'], ctx['n']
    model.eval()
    with torch.no_grad():
        rec = V.field_from_net_fused(ctx['ds'], model)
        psnr, l1, mse, rmse = V.calculate_deviation_statistics(rec, vol, verbose=False)
    total = len(eager) + warmup + steps
    del graph, ctx, model, vol, rec
    torch.cuda.empty_cache()
    return {'workload': 'cfg3 train step: 64^3x32ch grid (4-level db2), MLP 4x128, 32768 lattice samples of a smooth '
                        'synthetic 255^3 volume (8 low-frequency sinusoids + 0.05 noise), fwd + GT + MSE + bwd + fused '
                        'Adam, one HIP graph', 'precision': precision, 'samples': n,
            'ms_per_step': ms, 'value': n / (ms * 1e-3) / 1e6, 'unit': 'Msamples/s', 'steps': steps,
            'optimizer_steps_total': total, 'loss_first': loss_first, 'loss_last': loss_last,
            'psnr_dB_after_run': float(psnr), 'rmse_after_run': float(rmse)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default='headline', choices=sorted(WORKLOADS))
    ap.add_argument('--precision', default='f16x2', choices=['f16x2', 'fp32', 'f16'],
                    help="layer-GEMM arithmetic: 'f16x2' (default, f16 hi+lo split on the f16 matrix pipe) or 'fp32' (exact f32 MFMA)")
    ap.add_argument('--gather', default='root', choices=['root', 'all'],
                    help="N > 1: how the volume is assembled for `value` (the other mode is timed beside it)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra', action='store_true', help='skip the other BASELINE configs (cfg 2, cfg 5, cfg-3 train step)')
    ap.add_argument('--no-check', action='store_true', help='skip the finite-output check (ablation builds)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit('launch with torch.distributed.run --nproc-per-node %d for --gpus %d' % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        sys.exit('bench.py needs an MI355X (no CPU fallback for the HIP path)')
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=device)

    from latent_feature_grid_compression_amd.data.IndexDataset import IndexDataset
    from latent_feature_grid_compression_amd.visualization import OutputToVTK as V
    from latent_feature_grid_compression_amd import ops

    w = WORKLOADS[args.workload]
    model = build_model(w, seed=2003, device=device)
    model.precision = args.precision
    ds = IndexDataset((w['vol'],) * 3, 16, build_index_table=False)
    res = ds.vol_res_touple

    ev = []                                   # (start, end, samples) per timed launch of the dominant kernel
    ev_prep = []                              # (start, end) per timed decode + pack
    state = {'timed': False}

    def slab_fn(b, e, out_view):
        # the fused forward of this rank's slab; events bracket exactly the dominant kernel's launch (and, apart from
        # it, the wavelet decode + parameter pack that every pass repeats)
        with torch.no_grad():
            timed = state['timed']
            if timed:
                p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                p0.record()
            grid_cl = model._decoded_channel_last()
            packed = model._packed()
            if timed:
                p1.record()
                ev_prep.append((p0, p1))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            ops.forward_raw(model._descriptor(), grid_cl, packed, pos=None, lattice=(res, b, e, 32), clamp=True,
                            out=out_view.view(-1), precision=model.precision)
            if timed:
                e1.record()
                ev.append((e0, e1, (e - b) * res[1] * res[2]))

    def one_step(gather, tm=None):
        model._grid_cache = None          # every pass decodes the wavelet-coded grid and re-packs: no cached outputs
        model._pack_cache = None
        return V.reconstruct_volume_sharded(ds, model, 32, slab_fn=slab_fn, device=device, gather=gather, timings=tm)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_loop(gather, steps, collect):
        for _ in range(args.warmup):
            vol = one_step(gather)
        sync()
        state['timed'] = collect
        waits = []
        t0 = time.perf_counter()
        for _ in range(steps):
            tm = {} if (collect and world > 1) else None
            vol = one_step(gather, tm)
            if tm:
                waits.append(tm['gather_wait_ms'])
        sync()
        elapsed = time.perf_counter() - t0
        state['timed'] = False
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.item(), vol, waits

    elapsed, vol, waits = timed_loop(args.gather, args.steps, True)
    total_samples = res[0] * res[1] * res[2]
    value = total_samples * args.steps / elapsed / 1e6
    if vol is not None:
        assert tuple(vol.shape) == tuple(res) and (args.no_check or bool(torch.isfinite(vol).all()))

    kern_total_ms = float(sum(a.elapsed_time(b) for a, b, _ in ev))
    kern_samples = int(sum(n for _, _, n in ev))
    kern_ms = kern_total_ms / max(len(ev), 1)
    prep_ms = float(sum(a.elapsed_time(b) for a, b in ev_prep)) / max(len(ev_prep), 1)

    multi = None
    if world > 1:
        # per-rank attribution (events on each rank's own stream) + the other assembly mode under the same clock
        mine = torch.tensor([kern_ms * len(ev) / max(args.steps, 1), prep_ms, float(np.mean(waits)) if waits else 0.0],
                            dtype=torch.float64, device=device)
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        other = 'all' if args.gather == 'root' else 'root'
        o_steps = max(3, args.steps // 2)
        o_elapsed, _, _ = timed_loop(other, o_steps, False)
        multi = {'world_size': dist.get_world_size(), 'backend': dist.get_backend(), 'gather': args.gather,
                 'per_rank': {'kernel_ms_per_step': [float(t[0]) for t in allr], 'decode_pack_ms': [float(t[1]) for t in allr],
                              'gather_wait_ms': [float(t[2]) for t in allr]},
                 'bytes_received_by_root_per_step': 4 * total_samples * (world - 1) // world,
                 'other_mode': {'gather': other, 'value': total_samples * o_steps / o_elapsed / 1e6,
                                'ms_per_step': o_elapsed / o_steps * 1e3, 'steps': o_steps}}

    if rank == 0:
        K0 = 3 + 12 + w['C']
        flop_per_sample = 2 * (K0 * w['H'] + (w['L'] - 1) * w['H'] ** 2 + w['H'])
        bytes_per_sample = 12 + 4 + 8 * w['C'] * 4
        achieved_tflops = flop_per_sample * kern_samples / (kern_total_ms * 1e-3) / 1e12
        hbm_alg_gbs = bytes_per_sample * kern_samples / (kern_total_ms * 1e-3) / 1e9
        split = model.precision == 'f16x2'
        half = model.precision == 'f16'          # reduced-precision opt-in: NOT the headline configuration
        # the matrix pipe the build runs on: dense f16 MFMA (2.5 PFLOP/s) for both f16 builds, f32-input MFMA for 'fp32'
        peak = FP32_MFMA_PEAK_TFLOPS if model.precision == 'fp32' else F16_MFMA_PEAK_TFLOPS
        executed = achieved_tflops * (3.0 if split else 1.0)
        out = {
            'metric': 'Msamples/s (grid-interp+embed+MLP fwd) on 256^3 volume' if args.workload == 'headline'
                      else 'Msamples/s (grid-interp+embed+MLP fwd)',
            'value': value, 'unit': 'Msamples/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'strong', 'vs_baseline': None,
            'dtype': ('f32 (layer GEMMs as f16 hi+lo split, f32 accumulate)' if split else
                      'f16 GEMM inputs, f32 accumulate (REDUCED precision, not the headline configuration)' if half else 'f32'),
            'data': 'synthetic',
            'config': {'workload': w['desc'], 'samples_per_step': total_samples, 'tiles': 'x-slabs of 32^3 tiles',
                       'parallelism': ('tile-slab x%d, pieces gathered to %s (RCCL) under the next piece\'s compute'
                                       % (world, 'rank 0' if args.gather == 'root' else 'every rank')) if world > 1 else 'single GPU',
                       'step_includes': 'wavelet decode + param pack + fused forward' + (' + RCCL gather' if world > 1 else '')},
            # bound: the MLP contractions (110 592 algorithmic FLOP per sample) on the f16 matrix pipe; `achieved` counts
            # ALGORITHMIC fp32 FLOPs, `peak` is the dense peak of the pipe the build executes on (no credit for the 3x
            # emulation work: that is `executed_mfma`).  The HBM-side figure on algorithmic bytes is beside it.
            'roofline': {'bound': 'mfma', 'achieved': achieved_tflops, 'peak': peak, 'unit': 'TFLOP/s',
                         'frac': achieved_tflops / peak, 'traffic': traffic_bytes(args.workload, model.precision),
                         'kernel': 'lfgc_fwd16_kernel' if (split or half) else 'lfgc_fwd_kernel', 'kernel_ms': kern_ms,
                         'launches_per_step': len(ev) // max(args.steps, 1),
                         'samples_per_launch': kern_samples // max(len(ev), 1), 'flop_per_sample': flop_per_sample,
                         'executed_mfma': {'TFLOPs': executed, 'frac_of_peak': executed / peak,
                                           'note': ('three f16 MFMAs per fp32 product block (hi/lo split)' if split else
                                                    'one MFMA per product block')},
                         'decode_pack_ms': prep_ms,
                         'profiled': profiled_counters(args.workload, model.precision),
                         'hbm_algorithmic': {'bytes_per_sample': bytes_per_sample, 'achieved_GBs': hbm_alg_gbs,
                                             'peak_GBs': HBM_PEAK_GBS, 'frac': hbm_alg_gbs / HBM_PEAK_GBS}},
        }
        if multi is not None:
            out['multi_gpu'] = multi
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(model, w, hip_volume=vol)
            if split:
                # the same pass with the exact build (v_mfma_f32_32x32x2_f32, bitwise an fp32 fmaf chain): its rate and
                # how far the default build's volume is from it
                model.precision = 'fp32'
                for _ in range(2):
                    v32 = one_step('all')
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    v32 = one_step('all')
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 5
                out['exact_fp32_build'] = {'value': total_samples / dt / 1e6, 'unit': 'Msamples/s', 'ms_per_step': dt * 1e3,
                                           'max_rel_diff_of_default_build': float((vol - v32).abs().max() / v32.abs().max())}
                model.precision = 'f16x2'
                del v32
        else:
            out['cpu_baseline'] = None
        if world == 1 and not args.no_extra and args.workload == 'headline':
            # every other BASELINE config under the same clock (HIP events; sample counts given so value x time can be
            # cross-checked): cfg 2 and cfg 5 forward + decode, the cfg-3 train step in the default and the reduced build
            del vol
            torch.cuda.empty_cache()
            extra = {}
            for key, fn in (('cfg2_forward', lambda: forward_extra('cfg2', device, args.precision, reps=10, random_batch=4 * 1024 * 1024)),
                            ('cfg5_forward', lambda: forward_extra('cfg5', device, args.precision, reps=2, check_tiles=8)),
                            ('cfg5_forward_5level', lambda: forward_extra('cfg5_l5', device, args.precision, reps=2, check_tiles=8)),
                            ('cfg3_train_step', lambda: train_step_extra(device, args.precision)),
                            ('cfg3_train_step_reduced_f16', lambda: train_step_extra(device, 'f16'))):
                try:
                    extra[key] = fn()
                except Exception as exc:                      # noqa: BLE001 -- an extra must not take the headline down
                    extra[key] = {'error': '%s: %s' % (type(exc).__name__, exc)}
            out['extra'] = extra
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
