"""Full-volume reconstruction drivers: counterpart of visualization/OutputToVTK.py
(field_from_net :7-47, calculate_deviation_statistics :53-60, tiled_net_out :64-82).

* ``field_from_net`` keeps the reference's call contract (one ``net(tile)`` per 32^3 tile) so existing
  callers work unchanged.
* ``field_from_net_fused`` evaluates an x-slab of the volume in ONE kernel launch: the per-tile lattice
  (same fp32 operation order as the reference's host code) is generated inside the HIP kernel, so no
  positions are uploaded and nothing is copied back per tile.
* ``reconstruct_volume_sharded`` cuts the tile lattice into contiguous x-slabs, one per rank
  (torch.distributed, one process per GPU), and assembles the volume with a gather to the root rank or an
  all-gather (RCCL over xGMI on MI355X; gloo in the CPU tests), received straight into views of the output.
  Tiles are independent, the decoded grid is replicated: there is no other communication.
VTK file output (pyevtk) is out of scope.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

from .. import ops


def iter_tiles(res, tiled_res: int = 32):
    """(x0, x1, y0, y1, z0, z1) of every tile of the volume lattice, x-major -- the order slabs and pieces are cut in."""
    for x0 in range(0, res[0], tiled_res):
        for y0 in range(0, res[1], tiled_res):
            for z0 in range(0, res[2], tiled_res):
                yield (x0, min(x0 + tiled_res, res[0]), y0, min(y0 + tiled_res, res[1]), z0, min(z0 + tiled_res, res[2]))


def field_from_net(dataset, net, is_cuda, tiled_res=32, verbose=False):
    """One ``net(tile)`` call per tile with the call contract of the reference's driver (visualization/OutputToVTK.py:7-47):
    a (1, [1,] x, y, z, 3) position tensor in, a clamped (1, [1,] x, y, z, 1) tensor out, result assembled on the host.
    For callers that hand in their own ``net``; the package's own path is ``field_from_net_fused`` (one launch per slab,
    positions formed inside the kernel).  Tile positions come from ``IndexDataset.tile_positions``."""
    res = dataset.vol_res_touple
    full_vol = torch.zeros(res)
    with torch.no_grad():
        for (x0, x1, y0, y1, z0, z1) in iter_tiles(res, tiled_res):
            pos = dataset.tile_positions((x0, y0, z0), (x1, y1, z1)).unsqueeze(0)
            if is_cuda:
                pos = pos.unsqueeze(0).cuda()
            full_vol[x0:x1, y0:y1, z0:z1] = net(pos).reshape(x1 - x0, y1 - y0, z1 - z0).cpu()
    return full_vol


def field_from_net_fused(dataset, net, x_begin: int = 0, x_end: Optional[int] = None, tiled_res: int = 32,
                         out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Eval-mode forward of ``net`` (a Feature_Grid_Model on the GPU) over the x-slab [x_begin, x_end) of the
    volume lattice, one launch, result (x_end - x_begin, Y, Z) on the device, clamped like the eval branch."""
    res = dataset.vol_res_touple
    x_end = res[0] if x_end is None else int(x_end)
    with torch.no_grad():
        grid_cl = net._decoded_channel_last()
        flat = None if out is None else out.view(-1)
        y, _ = ops.forward_raw(net._descriptor(), grid_cl, net._packed(), pos=None,
                               lattice=(res, x_begin, x_end, tiled_res), clamp=True, out=flat,
                               precision=getattr(net, 'precision', 'f16x2'))
    return y.view(x_end - x_begin, res[1], res[2])


def calculate_deviation_statistics(prediction, ground_truth, verbose: bool = True):
    """PSNR / L1 / MSE / RMSE as the reference defines them (:53-60).  GPU tensors are reduced by the HIP
    kernel with fp64 accumulators; CPU tensors (e.g. the reference-style tile loop's output) by torch."""
    if prediction.is_cuda and ground_truth.is_cuda:
        acc = ops.deviation_partial(prediction, ground_truth).cpu()
        n = prediction.numel()
        mse = acc[0].item() / n
        l1 = acc[1].item() / n
        sqd = (acc[3].item() - acc[2].item()) ** 2
        psnr = 10.0 * float(np.log10(sqd / mse))
        rmse = float(np.sqrt(mse))
    else:
        diff_vol = ground_truth - prediction
        sqd_max_diff = (torch.max(ground_truth) - torch.min(ground_truth)) ** 2
        l1 = torch.mean(torch.abs(diff_vol)).item()
        mse_t = torch.mean(torch.pow(diff_vol, 2.0))
        psnr = (10 * torch.log10(sqd_max_diff / mse_t)).item()
        mse, rmse = mse_t.item(), torch.sqrt(mse_t).item()
    if verbose:
        print('PSNR:', psnr, 'l1:', l1, 'mse:', mse, 'rmse:', rmse)
    return psnr, l1, mse, rmse


def tiled_net_out(dataset, net, is_cuda, gt_vol=None, evaluate=True, write_vols=False, filename='vol', fused=True):
    """Counterpart of tiled_net_out (:64-82): eval(), reconstruct, statistics, back to train()."""
    if write_vols:
        raise NotImplementedError('VTK output (pyevtk) is outside the scope of this package')
    if is_cuda:
        net = net.cuda()
    net.eval()
    if fused and is_cuda:
        full_vol = field_from_net_fused(dataset, net)
        if gt_vol is not None:
            gt_vol = gt_vol.to(full_vol.device)
    else:
        full_vol = field_from_net(dataset, net, is_cuda, tiled_res=32)
    psnr = l1_diff = mse = rmse = 0
    if evaluate and gt_vol is not None:
        psnr, l1_diff, mse, rmse = calculate_deviation_statistics(full_vol, gt_vol)
    net.train()
    return psnr, l1_diff, mse, rmse


# ---- multi-GPU: x-slabs of tiles, one all-gather ---------------------------------------------------------

def slab_partition(res_x: int, world_size: int, tiled_res: int = 32) -> List[Tuple[int, int]]:
    """Contiguous x ranges [begin, end) per rank, cut on tile boundaries, balanced to within one tile."""
    n_tiles = (res_x + tiled_res - 1) // tiled_res
    bounds = []
    for r in range(world_size):
        t0 = (r * n_tiles) // world_size
        t1 = ((r + 1) * n_tiles) // world_size
        bounds.append((min(t0 * tiled_res, res_x), min(t1 * tiled_res, res_x)))
    return bounds


def reconstruct_volume_sharded(dataset, net=None, tiled_res: int = 32, group=None,
                               slab_fn: Optional[Callable[[int, int, torch.Tensor], None]] = None,
                               device: Optional[torch.device] = None, chunks: Optional[int] = None,
                               always_gather: bool = False, gather: str = 'all', root: int = 0,
                               timings: Optional[dict] = None) -> Optional[torch.Tensor]:
    """Every rank evaluates its x-slab of tiles; the slabs are assembled over the process group (RCCL over xGMI on
    MI355X, gloo in the CPU tests).

    ``gather='root'``: a gather to rank ``root`` -- the only rank that assembles (and returns) the (X,Y,Z) volume, the
    others return None; each rank's slab crosses the fabric once (what a writer of the volume needs).
    ``gather='all'``: an all-gather -- every rank returns the full volume (world-1 times the bytes of 'root').

    The slab is cut into ``chunks`` pieces of x-rows (default: up to 4 whole tile planes, never less than a plane per
    piece; an explicit ``chunks`` cuts on multiples of 8 rows when there are fewer planes than pieces); each piece's
    collective is issued
    asynchronously right after the piece's kernel launch, so the transfer of piece c runs under the compute of piece
    c+1.  Pieces are received straight into views of the output volume (x is the slowest axis, so a piece of a slab is a
    contiguous range of it); only a piece that is shorter than the common piece length (ragged volumes, e.g. 255^3 =
    7 x 32 + 31) goes through a scratch buffer for its padding rows.  ``chunks=1`` = one collective.

    ``slab_fn(x_begin, x_end, out_view)`` fills ``out_view`` ((x_end-x_begin, Y, Z)); default = the fused HIP
    forward of ``net``.  (The CPU/gloo tests inject a stub here: the HIP path itself has no CPU form.)
    ``timings``: a dict that receives 'compute_ms' (slab_fn calls, device time) and 'gather_wait_ms' (time this rank
    spent waiting for the collectives after its last launch), measured with events on the current stream."""
    if gather not in ('all', 'root'):
        raise ValueError("gather must be 'all' or 'root'")
    res = dataset.vol_res_touple
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    parts = slab_partition(res[0], world, tiled_res)
    max_x = max(e - b for b, e in parts)
    if device is None:
        device = next(net.parameters()).device if net is not None else torch.device('cpu')
    if slab_fn is None:
        def slab_fn(b, e, out_view):
            field_from_net_fused(dataset, net, b, e, tiled_res, out=out_view)
    on_gpu = torch.device(device).type == 'cuda'
    ev = []

    def timed_slab(xb, xe, view):
        if timings is not None and on_gpu:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            slab_fn(xb, xe, view)
            e1.record()
            ev.append((e0, e1))
        else:
            slab_fn(xb, xe, view)

    b, e = parts[rank]
    if world == 1 and not always_gather:     # always_gather: run the collective path even alone (RCCL smoke test)
        out = torch.empty(res, dtype=torch.float32, device=device)
        if e > b:
            timed_slab(b, e, out[b:e])
        if timings is not None:
            if on_gpu:
                torch.cuda.synchronize(device)
            timings.update(compute_ms=float(sum(a.elapsed_time(c) for a, c in ev)), gather_wait_ms=0.0, world_size=1)
        return out
    # piece boundaries relative to the slab start, on tile planes, identical on every rank (from max_x).  Default: up to 4
    # pieces of WHOLE tile planes -- a piece below one plane (~0.7 ms of kernel on 256^3) costs more host time to issue
    # than it hides: at 8 ranks on 256^3 (one plane per rank) 4 pieces of 8 rows took 2.05 ms of wall per step against
    # 0.76 ms for one piece (profiles/r3/sharded_host_overhead.log).  An explicit `chunks` still cuts below a plane.
    planes = (max_x + tiled_res - 1) // tiled_res
    want = min(4, max(1, planes)) if chunks is None else max(1, int(chunks))
    unit = tiled_res if planes >= want else 8
    units = (max_x + unit - 1) // unit
    n_chunks = max(1, min(units, want))
    cuts = [min(((c * units) // n_chunks) * unit, max_x) for c in range(n_chunks + 1)]
    cuts[-1] = max_x
    assembles = gather == 'all' or rank == root
    out = torch.empty(res, dtype=torch.float32, device=device) if assembles else None
    works, fixups = [], []
    for c in range(n_chunks):
        lo, hi = cuts[c], cuts[c + 1]
        span = hi - lo
        xb, xe = min(b + lo, e), min(b + hi, e)
        full_mine = xe - xb == span
        # my piece: computed in place in the output volume when I assemble and the piece is whole
        if assembles and full_mine:
            mine = out[xb:xe]
        else:
            mine = torch.empty((span, res[1], res[2]), dtype=torch.float32, device=device)
            if not full_mine:
                mine[xe - xb:].zero_()               # rows past this rank's (shorter) slab: padding of the equal-size collective
        if xe > xb:
            timed_slab(xb, xe, mine[:xe - xb])
        if assembles:
            dests = []
            for r, (rb, re_) in enumerate(parts):
                take = max(0, min(re_ - rb - lo, span))
                if r == rank and full_mine:
                    dests.append(mine)
                elif take == span:
                    dests.append(out[rb + lo:rb + lo + span])
                else:                               # short (or empty) piece: its padding rows need somewhere to land
                    scratch = torch.empty((span, res[1], res[2]), dtype=torch.float32, device=device)
                    dests.append(scratch)
                    if take > 0:
                        fixups.append((out[rb + lo:rb + lo + take], scratch, take))
        if gather == 'all':
            works.append(dist.all_gather(dests, mine, group=group, async_op=True))
        else:
            works.append(dist.gather(mine, gather_list=dests if rank == root else None, dst=root, group=group, async_op=True))
    if timings is not None and on_gpu:
        g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g0.record()
    for w in works:
        w.wait()
    for dst, scratch, take in fixups:
        dst.copy_(scratch[:take])
    if timings is not None:
        if on_gpu:
            g1.record()
            torch.cuda.synchronize(device)
            timings.update(compute_ms=float(sum(a.elapsed_time(c) for a, c in ev)), gather_wait_ms=float(g0.elapsed_time(g1)))
        timings.update(world_size=world, chunks=n_chunks, gather=gather)
    return out
