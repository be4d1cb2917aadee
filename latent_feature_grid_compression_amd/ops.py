"""Tensor-level wrappers over the C-ABI (device memory and streams come from PyTorch-ROCm; all
arithmetic happens in liblfgc.so) and the two autograd Functions of the hot path."""
from __future__ import annotations

import ctypes
import functools
import os
import weakref
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch.autograd.function import once_differentiable

from . import _lib
from ._lib import MlpDesc, Positions, check


_DEBUG_STATUS = bool(os.environ.get('LFGC_DEBUG_STATUS'))


def _require_hip(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _lib.LfgcError('the latent-feature-grid hot path runs on the MI355X only: got a %s tensor. '
                                 'Move the module and its inputs to the GPU (model.cuda()); there is no CPU fallback.'
                                 % t.device)


def _stream(t: torch.Tensor) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _cuda_tensors(objs):
    for o in objs:
        if torch.is_tensor(o):
            if o.is_cuda:
                yield o
        elif isinstance(o, (list, tuple)):
            yield from _cuda_tensors(o)


def _on_device(fn):
    """Run a wrapper with the device of its tensor arguments current.  The library launches on the calling thread's
    CURRENT device (and keeps its per-device caches by it), while the stream handle passed down belongs to the tensors'
    device: a model on cuda:1 called while cuda:0 is current would otherwise launch on the wrong device.  Tensors living
    on different devices are refused."""
    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        dev = None
        for t in _cuda_tensors(list(args) + list(kwargs.values())):
            if dev is None:
                dev = t.device
            elif t.device != dev:
                raise _lib.LfgcError('tensors on different devices (%s and %s) in one call of %s' % (dev, t.device, fn.__name__))
        if dev is None or dev.index == torch.cuda.current_device():
            return fn(*args, **kwargs)
        with torch.cuda.device(dev):
            return fn(*args, **kwargs)
    return wrapped


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def make_desc(C: int, H: int, L: int, n_freqs: int, d_in: int = 3, d_out: int = 1) -> MlpDesc:
    d = MlpDesc(int(C), int(H), int(L), int(n_freqs), int(d_in), int(d_out))
    if not _lib.load().lfgc_mlp_supported(ctypes.byref(d)):
        raise _lib.LfgcError('network shape not covered by the compiled HIP kernels: grid_channels=%d (<=32), hidden=%d '
                             '(<=128), num_layer=%d (<=8), n_freqs=%d (==2), d_in=%d (==3), d_out=%d (==1)'
                             % (C, H, L, n_freqs, d_in, d_out))
    return d


def grid_channel_stride(C: int) -> int:
    return int(_lib.load().lfgc_grid_channel_stride(int(C)))


# ---- wavelet levels ------------------------------------------------------------------------------------

_TAPS_CACHE = {}      # id(filter tensor) -> (weakref, version, ctypes float[8] or None)


def _outer_bank(a: np.ndarray) -> np.ndarray:
    """(2,4) fp32 taps -> (8,4,4,4) fp32 the way the reference forms it: a[sz][tz] * (a[sy][ty] * a[sx][tx]), s = 4sz+2sy+sx
    (wavelet_transform/Torch_Wavelet_Transform.py:44-53)."""
    a = a.astype(np.float32)
    yx = (a[:, None, :, None] * a[None, :, None, :]).astype(np.float32)               # [sy][sx][ty][tx]
    out = (a[:, None, None, :, None, None] * yx[None, :, :, None, :, :]).astype(np.float32)   # [sz][sy][sx][tz][ty][tx]
    return out.reshape(8, 4, 4, 4)


def _factor_bank(f3d: np.ndarray) -> Optional[np.ndarray]:
    """1-D bank (2,4) whose outer product IS the given (8,4,4,4) filter, or None if the filter is not separable."""
    f3d = f3d.astype(np.float32).reshape(8, 4, 4, 4)
    c = float(f3d[0, 0, 0, 0])
    if c == 0.0 or not np.isfinite(f3d).all():
        return None
    a0 = np.cbrt(np.float64(c))
    a = (f3d[[0, 1], 0, 0, :].astype(np.float64) / (a0 * a0)).astype(np.float32)      # bands (0,0,sx), taps along x
    # exact fp32 taps are within an ulp or two of this estimate: search the neighbours for a bit-exact reconstruction
    best, best_err = None, np.inf
    for d_lo in (0, -1, 1, -2, 2):
        for d_hi in (0, -1, 1, -2, 2):
            cand = a.copy()
            for row, d in ((0, d_lo), (1, d_hi)):
                for _ in range(abs(d)):
                    cand[row] = np.nextafter(cand[row], np.float32(np.inf) if d > 0 else np.float32(-np.inf))
            err = float(np.abs(_outer_bank(cand) - f3d).max())
            if err < best_err:
                best, best_err = cand, err
            if err == 0.0:
                return cand
    scale = float(np.abs(f3d).max())
    return best if best_err <= 4e-7 * scale else None


def filter_taps(filt: torch.Tensor):
    """ctypes float[8] = the 1-D bank of a wavelet filter buffer (or None: not separable -> dense stencil).  Costs
    one 2 KB device-to-host copy the first time a buffer is seen; cached per tensor object and in-place version."""
    ent = _TAPS_CACHE.get(id(filt))
    if ent is not None and ent[0]() is filt and ent[1] == filt._version:
        return ent[2]
    bank = _factor_bank(filt.detach().float().cpu().numpy())
    taps = None if bank is None else (ctypes.c_float * 8)(*[float(v) for v in bank.reshape(-1)])
    if len(_TAPS_CACHE) > 256:
        for k in [k for k, e in _TAPS_CACHE.items() if e[0]() is None]:
            del _TAPS_CACHE[k]
    _TAPS_CACHE[id(filt)] = (weakref.ref(filt), filt._version, taps)
    return taps


@_on_device
def idwt_level(lll: torch.Tensor, hf: torch.Tensor, filter_rev: torch.Tensor, target: Sequence[int]) -> torch.Tensor:
    """lll (C,d0,d1,d2), hf (C,7,d0,d1,d2) -> (C,t0,t1,t2)."""
    _require_hip(lll, hf, filter_rev)
    taps = filter_taps(filter_rev)
    lll, hf, filter_rev = _f32c(lll), _f32c(hf), _f32c(filter_rev)
    C, d0, d1, d2 = lll.shape
    if tuple(hf.shape) != (C, 7, d0, d1, d2):
        raise ValueError('detail bands %s do not match low band %s' % (tuple(hf.shape), tuple(lll.shape)))
    t = [int(v) for v in target]
    out = torch.empty((C, t[0], t[1], t[2]), dtype=torch.float32, device=lll.device)
    check(_lib.load().lfgc_idwt_level_f32(lll.data_ptr(), hf.data_ptr(), filter_rev.data_ptr(), taps, out.data_ptr(),
                                          C, d0, d1, d2, t[0], t[1], t[2], _stream(lll)), 'lfgc_idwt_level_f32')
    return out


@_on_device
def idwt_level_bwd(d_out: torch.Tensor, filter_rev: torch.Tensor, d: Sequence[int]) -> Tuple[torch.Tensor, torch.Tensor]:
    """d_out (C,t0,t1,t2) -> (d_lll (C,d0,d1,d2), d_hf (C,7,d0,d1,d2))."""
    _require_hip(d_out, filter_rev)
    taps = filter_taps(filter_rev)
    d_out, filter_rev = _f32c(d_out), _f32c(filter_rev)
    C, t0, t1, t2 = d_out.shape
    d = [int(v) for v in d]
    d_lll = torch.empty((C, d[0], d[1], d[2]), dtype=torch.float32, device=d_out.device)
    d_hf = torch.empty((C, 7, d[0], d[1], d[2]), dtype=torch.float32, device=d_out.device)
    check(_lib.load().lfgc_idwt_level_bwd_f32(d_out.data_ptr(), filter_rev.data_ptr(), taps, d_lll.data_ptr(), d_hf.data_ptr(),
                                              C, d[0], d[1], d[2], t0, t1, t2, _stream(d_out)), 'lfgc_idwt_level_bwd_f32')
    return d_lll, d_hf


@_on_device
def to_channel_last(grid_cf: torch.Tensor) -> torch.Tensor:
    """(C,D,H,W) -> (D,H,W,Cs), Cs = C rounded up to 8, pad channels zero."""
    _require_hip(grid_cf)
    grid_cf = _f32c(grid_cf)
    C, D, H, W = grid_cf.shape
    cs = grid_channel_stride(C)
    out = torch.empty((D, H, W, cs), dtype=torch.float32, device=grid_cf.device)
    check(_lib.load().lfgc_grid_layout_f32(grid_cf.data_ptr(), out.data_ptr(), C, D * H * W, cs, 1, _stream(grid_cf)),
          'lfgc_grid_layout_f32')
    return out


@_on_device
def to_channel_first(grid_cl: torch.Tensor, C: int) -> torch.Tensor:
    """(D,H,W,Cs) -> (C,D,H,W)."""
    _require_hip(grid_cl)
    grid_cl = _f32c(grid_cl)
    D, H, W, cs = grid_cl.shape
    out = torch.empty((C, D, H, W), dtype=torch.float32, device=grid_cl.device)
    check(_lib.load().lfgc_grid_layout_f32(grid_cl.data_ptr(), out.data_ptr(), C, D * H * W, cs, 0, _stream(grid_cl)),
          'lfgc_grid_layout_f32')
    return out


def dwt_out_shape(n: Sequence[int]) -> List[int]:
    n = [int(v) for v in n]
    hi = [2 + (n[2] & 1), 2 + (n[1] & 1), 2 + (n[0] & 1)]     # reference pad-slot quirk, see lfgc.h
    return [(n[a] + 2 + hi[a] - 4) // 2 + 1 for a in range(3)]


@_on_device
def dwt_level(data: torch.Tensor, filter_fwd: torch.Tensor) -> torch.Tensor:
    """data (C,n0,n1,n2) -> (C,8,d0,d1,d2)."""
    _require_hip(data, filter_fwd)
    taps = filter_taps(filter_fwd)
    data, filter_fwd = _f32c(data), _f32c(filter_fwd)
    C, n0, n1, n2 = data.shape
    d = dwt_out_shape((n0, n1, n2))
    out = torch.empty((C, 8, d[0], d[1], d[2]), dtype=torch.float32, device=data.device)
    check(_lib.load().lfgc_dwt_level_f32(data.data_ptr(), filter_fwd.data_ptr(), taps, out.data_ptr(), C, n0, n1, n2,
                                         _stream(data)), 'lfgc_dwt_level_f32')
    return out


_E_UNSUPPORTED = -3


def _cl_level_ok(C: int, d: Sequence[int], t: Sequence[int], taps) -> bool:
    """Shapes the channel-last last-level kernels take (lfgc.h: lfgc_idwt_level_cl_f32): separable bank, C <= 32,
    arrays below 2^30 bytes.  Anything else composes the channel-first level with the layout conversion."""
    if taps is None or C > 32 or os.environ.get('LFGC_CL_LEVEL', '1') == '0':
        return False
    cs = grid_channel_stride(C)
    return t[0] * t[1] * t[2] * cs * 4 < (1 << 30) and d[0] * d[1] * d[2] * 7 * C * 4 < (1 << 30)


@_on_device
def idwt_level_cl(lll: torch.Tensor, hf: torch.Tensor, filter_rev: torch.Tensor, target: Sequence[int]) -> torch.Tensor:
    """lll (C,d0,d1,d2), hf (C,7,d0,d1,d2) -> (t0,t1,t2,Cs) channel-last, pad channels zero: the last level of the
    decode and the layout conversion in one kernel (falls back to the two-kernel form for shapes it does not take)."""
    _require_hip(lll, hf, filter_rev)
    taps = filter_taps(filter_rev)
    C, d0, d1, d2 = lll.shape
    t = [int(v) for v in target]
    if not _cl_level_ok(C, (d0, d1, d2), t, taps):
        return to_channel_last(idwt_level(lll, hf, filter_rev, target))
    lll, hf = _f32c(lll), _f32c(hf)
    if tuple(hf.shape) != (C, 7, d0, d1, d2):
        raise ValueError('detail bands %s do not match low band %s' % (tuple(hf.shape), tuple(lll.shape)))
    cs = grid_channel_stride(C)
    out = torch.empty((t[0], t[1], t[2], cs), dtype=torch.float32, device=lll.device)
    check(_lib.load().lfgc_idwt_level_cl_f32(lll.data_ptr(), hf.data_ptr(), taps, out.data_ptr(), C, cs, d0, d1, d2,
                                             t[0], t[1], t[2], _stream(lll)), 'lfgc_idwt_level_cl_f32')
    return out


@_on_device
def idwt_level_cl_bwd(d_out_cl: torch.Tensor, C: int, filter_rev: torch.Tensor,
                      d: Sequence[int]) -> Tuple[torch.Tensor, torch.Tensor]:
    """d_out_cl (t0,t1,t2,Cs) -> (d_lll (C,d0,d1,d2), d_hf (C,7,d0,d1,d2)): adjoint of idwt_level_cl."""
    _require_hip(d_out_cl, filter_rev)
    taps = filter_taps(filter_rev)
    d = [int(v) for v in d]
    t0, t1, t2, cs = d_out_cl.shape
    if not _cl_level_ok(C, d, (t0, t1, t2), taps):
        return idwt_level_bwd(to_channel_first(d_out_cl, C), filter_rev, d)
    d_out_cl = _f32c(d_out_cl)
    d_lll = torch.empty((C, d[0], d[1], d[2]), dtype=torch.float32, device=d_out_cl.device)
    d_hf = torch.empty((C, 7, d[0], d[1], d[2]), dtype=torch.float32, device=d_out_cl.device)
    check(_lib.load().lfgc_idwt_level_cl_bwd_f32(d_out_cl.data_ptr(), taps, d_lll.data_ptr(), d_hf.data_ptr(), C, cs,
                                                 d[0], d[1], d[2], t0, t1, t2, _stream(d_out_cl)),
          'lfgc_idwt_level_cl_bwd_f32')
    return d_lll, d_hf


def decode_levels(coeffs: Sequence[torch.Tensor], shape_array, filter_rev: torch.Tensor,
                  channel_last: bool) -> torch.Tensor:
    """All IDWT levels (model/Feature_Grid_Model.py:102-108, drop layers already applied by the caller);
    channel_last: the last level writes the sampler's layout directly."""
    restored = coeffs[0]          # grids smaller than 6 voxels have no wavelet level at all (dwt_max_level = 0)
    n = len(coeffs) - 1
    for k, (hf, shape) in enumerate(zip(coeffs[1:], shape_array)):
        if channel_last and k == n - 1:
            return idwt_level_cl(restored, hf, filter_rev, shape)
        restored = idwt_level(restored, hf, filter_rev, shape)
    return to_channel_last(restored) if channel_last else restored


class DecodeVolumeFn(torch.autograd.Function):
    """decode_volume() as one autograd node: IDWT chain forward, adjoint chain backward."""

    @staticmethod
    def forward(ctx, filter_rev, shape_array, channel_last, *coeffs):
        ctx.filter_rev = filter_rev
        ctx.shape_array = [tuple(int(v) for v in s) for s in shape_array]
        ctx.channel_last = bool(channel_last)
        ctx.dims = [tuple(c.shape) for c in coeffs]
        with torch.no_grad():
            out = decode_levels([c.detach() for c in coeffs], ctx.shape_array, filter_rev, ctx.channel_last)
            # zero levels and channel-first: the output would alias the parameter; hand autograd a fresh tensor
            return out.clone() if out.data_ptr() == coeffs[0].data_ptr() else out

    @staticmethod
    @once_differentiable
    def backward(ctx, d_out):
        C = ctx.dims[0][0]
        n_levels = len(ctx.dims) - 1
        grads = [None] * len(ctx.dims)
        if ctx.channel_last and n_levels == 0:
            d_out = to_channel_first(d_out, C)
        g = d_out
        for lvl in range(n_levels, 0, -1):
            if ctx.channel_last and lvl == n_levels:
                g, d_hf = idwt_level_cl_bwd(g, C, ctx.filter_rev, ctx.dims[lvl][2:])
            else:
                g, d_hf = idwt_level_bwd(g, ctx.filter_rev, ctx.dims[lvl][2:])
            grads[lvl] = d_hf
        grads[0] = g
        return (None, None, None) + tuple(grads)


# ---- pruning ("drop") layers fused into the decode --------------------------------

_NAN = float('nan')


def _thr(v) -> float:
    return _NAN if v is None else float(v)


@_on_device
def idwt_level_drop(lll, hf, mul_l, thr_l, mul_h, thr_h, filter_rev, target) -> torch.Tensor:
    """One IDWT level with the drop factors of its inputs folded in: mul_l (d0,d1,d2) / mul_h (7,d0,d1,d2) or None;
    thr None = plain product, a float = masked straight-through rule (see include/lfgc.h)."""
    _require_hip(lll, hf, filter_rev, mul_l, mul_h)
    taps = filter_taps(filter_rev)
    lll, hf, filter_rev = _f32c(lll), _f32c(hf), _f32c(filter_rev)
    C, d0, d1, d2 = lll.shape
    if tuple(hf.shape) != (C, 7, d0, d1, d2):
        raise ValueError('detail bands %s do not match low band %s' % (tuple(hf.shape), tuple(lll.shape)))
    if mul_l is not None:
        mul_l = _f32c(mul_l)
        if tuple(mul_l.shape) != (d0, d1, d2):
            raise ValueError('low-band drop factor %s does not match %s' % (tuple(mul_l.shape), (d0, d1, d2)))
    if mul_h is not None:
        mul_h = _f32c(mul_h)
        if tuple(mul_h.shape) != (7, d0, d1, d2):
            raise ValueError('detail drop factor %s does not match %s' % (tuple(mul_h.shape), (7, d0, d1, d2)))
    t = [int(v) for v in target]
    out = torch.empty((C, t[0], t[1], t[2]), dtype=torch.float32, device=lll.device)
    check(_lib.load().lfgc_idwt_level_drop_f32(
        lll.data_ptr(), hf.data_ptr(), mul_l.data_ptr() if mul_l is not None else None, _thr(thr_l),
        mul_h.data_ptr() if mul_h is not None else None, _thr(thr_h), filter_rev.data_ptr(), taps, out.data_ptr(),
        C, d0, d1, d2, t[0], t[1], t[2], _stream(lll)), 'lfgc_idwt_level_drop_f32')
    return out


@_on_device
def idwt_level_drop_bwd(d_out, filter_rev, lll, hf, mul_l, mul_h, want_dml, want_dmh, d, penalty_ptrs=None):
    """Adjoint of idwt_level_drop -> (d_lll, d_hf, d_mul_l or None, d_mul_h or None).  want_dml / want_dmh: False, True
    (a zero tensor is allocated) or a ZERO-FILLED tensor of the factor's shape to accumulate into.  penalty_ptrs: None or
    4 device addresses (0 = none) of the upstream gradients of [sum lll^2, sum hf^2, sum |mul_l|, sum |mul_h|] whose own
    gradients the kernel folds in (include/lfgc.h)."""
    _require_hip(d_out, filter_rev)
    taps = filter_taps(filter_rev)
    d_out, filter_rev = _f32c(d_out), _f32c(filter_rev)
    C, t0, t1, t2 = d_out.shape
    d = [int(v) for v in d]
    dev = d_out.device
    d_lll = torch.empty((C, d[0], d[1], d[2]), dtype=torch.float32, device=dev)
    d_hf = torch.empty((C, 7, d[0], d[1], d[2]), dtype=torch.float32, device=dev)
    d_ml = want_dml if torch.is_tensor(want_dml) else (
        torch.zeros((d[0], d[1], d[2]), dtype=torch.float32, device=dev) if want_dml else None)
    d_mh = want_dmh if torch.is_tensor(want_dmh) else (
        torch.zeros((7, d[0], d[1], d[2]), dtype=torch.float32, device=dev) if want_dmh else None)
    ptr = lambda t: t.data_ptr() if t is not None else None
    pen, _keep = (None, None) if penalty_ptrs is None else _lib.ptr_array([int(v) for v in penalty_ptrs])
    check(_lib.load().lfgc_idwt_level_drop_bwd_f32(
        d_out.data_ptr(), filter_rev.data_ptr(), taps, ptr(lll), ptr(hf), ptr(mul_l), ptr(mul_h), d_lll.data_ptr(),
        d_hf.data_ptr(), ptr(d_ml), ptr(d_mh), pen, C, d[0], d[1], d[2], t0, t1, t2, _stream(d_out)),
        'lfgc_idwt_level_drop_bwd_f32')
    return d_lll, d_hf, d_ml, d_mh


@_on_device
def drop_apply(x: torch.Tensor, mul: torch.Tensor, thr=None) -> torch.Tensor:
    """x (C, ...) * mul (...) with the value rule of include/lfgc.h (one drop layer outside the decode)."""
    _require_hip(x, mul)
    x, mul = _f32c(x), _f32c(mul)
    if tuple(x.shape[1:]) != tuple(mul.shape):
        raise ValueError('drop factor %s does not match coefficients %s' % (tuple(mul.shape), tuple(x.shape)))
    out = torch.empty_like(x)
    check(_lib.load().lfgc_drop_apply_f32(x.data_ptr(), mul.data_ptr(), _thr(thr), out.data_ptr(), x.shape[0],
                                          mul.numel(), _stream(x)), 'lfgc_drop_apply_f32')
    return out


class DropApplyFn(torch.autograd.Function):
    """A drop layer's own forward(x): value by the layer's rule, gradients d_x = g*m, d_m = sum_c g*x."""

    @staticmethod
    @_on_device
    def forward(ctx, x, mul, thr):
        ctx.save_for_backward(x.detach(), mul.detach())
        ctx.need_dm = mul.requires_grad
        return drop_apply(x.detach(), mul.detach(), thr)

    @staticmethod
    @once_differentiable
    @_on_device
    def backward(ctx, g):
        x, mul = ctx.saved_tensors
        g, x, mul = _f32c(g), _f32c(x), _f32c(mul)
        d_x = torch.empty_like(x)
        d_m = torch.empty_like(mul) if ctx.need_dm else None
        check(_lib.load().lfgc_drop_apply_bwd_f32(g.data_ptr(), x.data_ptr(), mul.data_ptr(), d_x.data_ptr(),
                                                  d_m.data_ptr() if d_m is not None else None, x.shape[0], mul.numel(),
                                                  _stream(g)), 'lfgc_drop_apply_bwd_f32')
        return d_x, d_m, None


def decode_levels_drop(coeffs, factors, thresholds, shape_array, filter_rev, channel_last: bool) -> torch.Tensor:
    """decode_volume() with drop factors (model/Feature_Grid_Model.py:102-108): factors[i] / thresholds[i] belong
    to coeffs[i]; None = that tensor passes unchanged."""
    if len(coeffs) == 1:
        restored = coeffs[0] if factors[0] is None else drop_apply(coeffs[0], factors[0], thresholds[0])
    else:
        restored, mul_l, thr_l = coeffs[0], factors[0], thresholds[0]
        for k in range(1, len(coeffs)):
            restored = idwt_level_drop(restored, coeffs[k], mul_l, thr_l, factors[k], thresholds[k], filter_rev,
                                       shape_array[k - 1])
            mul_l, thr_l = None, None
    return to_channel_last(restored) if channel_last else restored


class DecodeVolumeDropFn(torch.autograd.Function):
    """decode_volume() with the drop layers folded into the IDWT kernels, as one autograd node.
    apply(filter_rev, shape_array, channel_last, thresholds, n, *coeffs, *factors) -- n coefficient tensors, then n
    factors (tensor or None); needs at least one wavelet level."""

    @staticmethod
    def forward(ctx, filter_rev, shape_array, channel_last, thresholds, n, *tensors):
        coeffs, factors = tensors[:n], tensors[n:]
        ctx.filter_rev = filter_rev
        ctx.shape_array = [tuple(int(v) for v in s) for s in shape_array]
        ctx.channel_last = bool(channel_last)
        ctx.n = n
        ctx.dims = [tuple(c.shape) for c in coeffs]
        ctx.want = [f is not None and f.requires_grad for f in factors]
        ctx.has = [f is not None for f in factors]
        ctx.l1_flags = [False] * n
        det = [c.detach() for c in coeffs]
        fdet = [f.detach() if f is not None else None for f in factors]
        ctx.save_for_backward(*det, *[f for f in fdet if f is not None])
        with torch.no_grad():
            return decode_levels_drop(det, fdet, list(thresholds), ctx.shape_array, filter_rev, ctx.channel_last)

    @staticmethod
    @once_differentiable
    def backward(ctx, d_out):
        d_coef, d_fac = _decode_drop_backward(ctx, d_out, None)
        return (None, None, None, None, None) + tuple(d_coef) + tuple(d_fac)


def _decode_drop_backward(ctx, d_out, d_pen):
    """Shared backward of the two decode-with-factors nodes; d_pen fp32 or None = upstream gradients of the penalty sums
    (layout of DecodeVolumePenaltyFn) folded into the adjoint kernels."""
    n = ctx.n
    saved = list(ctx.saved_tensors)
    coeffs = saved[:n]
    it = iter(saved[n:])
    factors = [next(it) if h else None for h in ctx.has]
    C = ctx.dims[0][0]
    g = to_channel_first(d_out, C) if ctx.channel_last else d_out
    d_coef, d_fac = [None] * n, [None] * n
    # the factor gradients are accumulated with atomics: one zero fill for all of them
    sizes = [int(np.prod(ctx.dims[i][1:])) if ctx.want[i] else 0 for i in range(n)]
    flat = torch.zeros(sum(sizes), dtype=torch.float32, device=g.device) if sum(sizes) else None
    zeroed, at = [False] * n, 0
    for i in range(n):
        if sizes[i]:
            zeroed[i] = flat[at:at + sizes[i]].view(ctx.dims[i][1:])
            at += sizes[i]
    if d_pen is not None:
        d_pen = _f32c(d_pen)
        base = d_pen.data_ptr()
        l2 = lambda i: base + 4 * i
        l1 = lambda i: (base + 4 * (n + ctx.l1_pos[i])) if (ctx.l1_flags[i] and zeroed[i] is not False) else 0
    for lvl in range(n - 1, 0, -1):
        first = lvl == 1
        ml = factors[0] if first else None
        pens = None
        if d_pen is not None:
            pens = [l2(0) if first else 0, l2(lvl), l1(0) if first else 0, l1(lvl)]
        g, d_hf, d_ml, d_mh = idwt_level_drop_bwd(
            g, ctx.filter_rev, coeffs[0] if (first and (ctx.has[0] or d_pen is not None)) else None, coeffs[lvl], ml,
            factors[lvl], zeroed[0] if first else False, zeroed[lvl], ctx.dims[lvl][2:], pens)
        d_coef[lvl], d_fac[lvl] = d_hf, d_mh
        if first:
            d_fac[0] = d_ml
    d_coef[0] = g
    return d_coef, d_fac


class DecodeVolumePenaltyFn(torch.autograd.Function):
    """DecodeVolumeDropFn that also returns the penalty sums of its inputs, so that their gradients ride in the adjoint
    kernels instead of costing passes (and autograd accumulation adds) of their own:
    apply(filter_rev, shape_array, channel_last, thresholds, n, l1_flags, *coeffs, *factors) -> (grid, pen) with
    pen fp32 = [sum coeffs[i]^2 for i in range(n)] followed by [sum |factors[i]| for the i with l1_flags[i]] (in order):
    factors that ARE the penalised parameter, e.g. Smallify betas."""

    @staticmethod
    @_on_device
    def forward(ctx, filter_rev, shape_array, channel_last, thresholds, n, l1_flags, *tensors):
        coeffs, factors = tensors[:n], tensors[n:]
        ctx.set_materialize_grads(False)
        ctx.filter_rev = filter_rev
        ctx.shape_array = [tuple(int(v) for v in s) for s in shape_array]
        ctx.channel_last = bool(channel_last)
        ctx.n = n
        ctx.l1_flags = [bool(f) and factors[i] is not None for i, f in enumerate(l1_flags)]
        ctx.dims = [tuple(c.shape) for c in coeffs]
        ctx.want = [f is not None and f.requires_grad for f in factors]
        ctx.has = [f is not None for f in factors]
        det = [c.detach() for c in coeffs]
        fdet = [f.detach() if f is not None else None for f in factors]
        ctx.save_for_backward(*det, *[f for f in fdet if f is not None])
        with torch.no_grad():
            grid = decode_levels_drop(det, fdet, list(thresholds), ctx.shape_array, filter_rev, ctx.channel_last)
            l1_idx = [i for i in range(n) if ctx.l1_flags[i]]
            kinds = [_lib.PENALTY_L2] * n + [_lib.PENALTY_L1] * len(l1_idx)
            terms, keep = _penalty_terms(kinds, det + [fdet[i] for i in l1_idx])
            sums = torch.empty(len(kinds) * (1 + _lib.PENALTY_BLOCKS), dtype=torch.float64, device=grid.device)
            check(_lib.load().lfgc_penalty_sums_f32(terms, len(kinds), sums.data_ptr(), _stream(sums)), 'lfgc_penalty_sums_f32')
            pen = sums[:len(kinds)].float()
        ctx.l1_pos = {i: j for j, i in enumerate(l1_idx)}
        return grid, pen

    @staticmethod
    @once_differentiable
    def backward(ctx, d_out, d_pen):
        if d_out is None:
            raise RuntimeError('the decoded grid took no part in the loss (only its penalties did): unsupported')
        d_coef, d_fac = _decode_drop_backward(ctx, d_out, d_pen)
        return (None, None, None, None, None, None) + tuple(d_coef) + tuple(d_fac)


@_on_device
def sign_variance_update(betas: torch.Tensor, ema: torch.Tensor, emavar: torch.Tensor, momentum: float) -> None:
    """In-place EMA / EMA-variance step of the Smallify sign tracker (model/Smallify_Dropout.py:106-112)."""
    _require_hip(betas, ema, emavar)
    if not (ema.is_contiguous() and emavar.is_contiguous() and ema.dtype == emavar.dtype == torch.float32):
        raise ValueError('tracker state must be contiguous fp32')
    b = _f32c(betas.detach())
    check(_lib.load().lfgc_sign_variance_update_f32(b.data_ptr(), ema.data_ptr(), emavar.data_ptr(), float(momentum),
                                                    b.numel(), _stream(b)), 'lfgc_sign_variance_update_f32')


@_on_device
def sign_variance_update_multi(betas, emas, emavars, momentum: float) -> None:
    """sign_variance_update for all drop layers of a model in one launch."""
    bs = [_f32c(b.detach()) for b in betas]
    _require_hip(*bs, *emas, *emavars)
    for e, v in zip(emas, emavars):
        if not (e.is_contiguous() and v.is_contiguous() and e.dtype == v.dtype == torch.float32):
            raise ValueError('tracker state must be contiguous fp32')
    pb, _k1 = _lib.ptr_array([b.data_ptr() for b in bs])
    pe, _k2 = _lib.ptr_array([e.data_ptr() for e in emas])
    pv, _k3 = _lib.ptr_array([v.data_ptr() for v in emavars])
    ns = (ctypes.c_int64 * len(bs))(*[b.numel() for b in bs])
    check(_lib.load().lfgc_sign_variance_update_multi_f32(pb, pe, pv, ns, len(bs), float(momentum), _stream(bs[0])),
          'lfgc_sign_variance_update_multi_f32')


def _penalty_terms(kinds, tensors):
    if len(kinds) > _lib.PENALTY_MAX_TERMS:
        raise ValueError('at most %d penalty terms per launch' % _lib.PENALTY_MAX_TERMS)
    terms = (_lib.PenaltyTerm * len(kinds))()
    keep, it = [], iter(tensors)
    for t, kind in enumerate(kinds):
        a = _f32c(next(it).detach())
        b = _f32c(next(it).detach()) if kind == _lib.PENALTY_DKL else None
        _require_hip(a, b)
        if b is not None and b.shape != a.shape:
            raise ValueError('log_thetas / log_var shapes differ')
        keep.append((a, b))
        terms[t].a, terms[t].b, terms[t].n, terms[t].kind = a.data_ptr(), (b.data_ptr() if b is not None else None), a.numel(), kind
    return terms, keep


class PenaltyFn(torch.autograd.Function):
    """All penalty terms of a pruning loss in one reduction launch (+ one gradient launch).
    apply(kinds, *tensors) -> fp32 (len(kinds),); a DKL term consumes two tensors (log_thetas, log_var)."""

    @staticmethod
    @_on_device
    def forward(ctx, kinds, *tensors):
        kinds = [int(k) for k in kinds]
        terms, keep = _penalty_terms(kinds, tensors)
        dev = keep[0][0].device
        sums = torch.empty(len(kinds) * (1 + _lib.PENALTY_BLOCKS), dtype=torch.float64, device=dev)   # results + scratch
        check(_lib.load().lfgc_penalty_sums_f32(terms, len(kinds), sums.data_ptr(), _stream(sums)), 'lfgc_penalty_sums_f32')
        ctx.kinds = kinds
        ctx.save_for_backward(*[t.detach() for t in tensors])
        return sums[:len(kinds)].float()

    @staticmethod
    @once_differentiable
    @_on_device
    def backward(ctx, d_sums):
        tensors = ctx.saved_tensors
        terms, keep = _penalty_terms(ctx.kinds, tensors)
        d_sums = _f32c(d_sums)
        ga = [torch.empty_like(a) for a, _ in keep]
        gb = [torch.empty_like(b) if b is not None else None for _, b in keep]
        pa, _k1 = _lib.ptr_array([g.data_ptr() for g in ga])
        pb, _k2 = _lib.ptr_array([g.data_ptr() if g is not None else 0 for g in gb])
        check(_lib.load().lfgc_penalty_grads_f32(terms, len(ctx.kinds), d_sums.data_ptr(), pa, pb, _stream(d_sums)),
              'lfgc_penalty_grads_f32')
        flat = []
        for g_a, g_b in zip(ga, gb):
            flat.append(g_a)
            if g_b is not None:
                flat.append(g_b)
        return (None,) + tuple(g.view(t.shape) for g, t in zip(flat, tensors))


def penalty_sums(kinds, tensors) -> torch.Tensor:
    return PenaltyFn.apply(list(kinds), *tensors)


# ---- fused sample + embed + MLP ------------------------------------------------------------------------

@_on_device
def pack_mlp(desc: MlpDesc, weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor]) -> torch.Tensor:
    lib = _lib.load()
    ws = [_f32c(w.detach()) for w in weights]
    bs = [_f32c(b.detach()) for b in biases]
    _require_hip(*ws, *bs)
    nbytes = int(lib.lfgc_packed_bytes(ctypes.byref(desc)))
    packed = torch.empty(nbytes // 4, dtype=torch.float32, device=ws[0].device)
    wp, _k1 = _lib.ptr_array([w.data_ptr() for w in ws])
    bp, _k2 = _lib.ptr_array([b.data_ptr() for b in bs])
    check(lib.lfgc_pack_mlp_f32(ctypes.byref(desc), wp, bp, packed.data_ptr(), _stream(packed)), 'lfgc_pack_mlp_f32')
    return packed


def _positions_struct(pos: Optional[torch.Tensor], lattice=None) -> Tuple[Positions, int]:
    ps = Positions()
    if pos is not None:
        ps.pos = pos.data_ptr()
        ps.n = pos.shape[0]
        ps.res[0] = ps.res[1] = ps.res[2] = 2
        ps.x_begin, ps.x_end, ps.tile = 0, 0, 32
        return ps, int(pos.shape[0])
    res, x_begin, x_end, tile = lattice
    ps.pos = None
    ps.n = 0
    for a in range(3):
        ps.res[a] = int(res[a])
    ps.x_begin, ps.x_end, ps.tile = int(x_begin), int(x_end), int(tile)
    return ps, (int(x_end) - int(x_begin)) * int(res[1]) * int(res[2])


@_on_device
def forward_raw(desc: MlpDesc, grid_cl: torch.Tensor, packed: torch.Tensor, pos: Optional[torch.Tensor] = None,
                lattice=None, clamp: bool = False, want_stash: bool = False, out: Optional[torch.Tensor] = None,
                precision: str = 'f16x2', range_fallback: bool = True, return_status: bool = False):
    """lfgc_forward_f32.  pos (N,3) or lattice=(res, x_begin, x_end, tile).  Returns (y (N,), stash or None)
    [+ the device status word with return_status].  range_fallback=False returns out-of-range samples of the f16
    builds as NaN instead of redoing the pass on the exact build (diagnostics)."""
    lib = _lib.load()
    _require_hip(grid_cl, packed, pos)
    if pos is not None:
        pos = _f32c(pos)
    ps, n = _positions_struct(pos, lattice)
    D, H, W, cs = grid_cl.shape
    if cs != grid_channel_stride(desc.grid_channels):
        raise ValueError('channel-last grid has stride %d, expected %d' % (cs, grid_channel_stride(desc.grid_channels)))
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=grid_cl.device)
    if n == 0:
        return out, (torch.empty(0, dtype=torch.float32, device=grid_cl.device) if want_stash else None)
    stash = None
    if want_stash:
        stash = torch.empty(int(lib.lfgc_stash_bytes(ctypes.byref(desc), n)) // 4, dtype=torch.float32, device=grid_cl.device)
    # range status word of the f16 builds: the library clears it, the kernel sets it, and the exact-fp32 redo the library
    # enqueues behind the kernel is predicated on it -- all in stream order, nothing is read back here
    status = torch.empty(1, dtype=torch.int32, device=grid_cl.device) if (range_fallback and precision != 'fp32') else None
    check(lib.lfgc_forward_f32(ctypes.byref(desc), ctypes.byref(ps), grid_cl.data_ptr(), D, H, W, packed.data_ptr(),
                               _lib.PRECISION[precision], int(clamp), out.data_ptr(),
                               stash.data_ptr() if stash is not None else None,
                               status.data_ptr() if status is not None else None, _stream(grid_cl)), 'lfgc_forward_f32')
    if _DEBUG_STATUS and status is not None:        # diagnostics: LFGC_DEBUG_STATUS=1 (synchronises)
        print('[lfgc] forward n=%d precision=%s status=%d' % (n, precision, int(status.item())), flush=True)
    if return_status:
        return out, stash, status
    return out, stash


@_on_device
def backward_raw(desc: MlpDesc, grid_cl, packed, pos, stash, d_out, weights, biases, need_d_pos: bool,
                 precision: str = 'f16x2'):
    lib = _lib.load()
    pos = _f32c(pos)
    d_out = _f32c(d_out)
    ps, n = _positions_struct(pos)
    D, H, W, cs = grid_cl.shape
    dev = grid_cl.device
    d_grid = torch.zeros_like(grid_cl)
    d_w = [torch.empty_like(w, dtype=torch.float32, memory_format=torch.contiguous_format) for w in weights]
    d_b = [torch.empty_like(b, dtype=torch.float32, memory_format=torch.contiguous_format) for b in biases]
    d_pos = torch.empty((n, 3), dtype=torch.float32, device=dev) if need_d_pos else None
    ws_bytes = int(lib.lfgc_backward_workspace_bytes(ctypes.byref(desc), n))
    ws = torch.empty(max(ws_bytes, 16) // 4, dtype=torch.float32, device=dev)
    wp, _k1 = _lib.ptr_array([w.data_ptr() for w in d_w])
    bp, _k2 = _lib.ptr_array([b.data_ptr() for b in d_b])
    check(lib.lfgc_backward_f32(ctypes.byref(desc), ctypes.byref(ps), grid_cl.data_ptr(), D, H, W, packed.data_ptr(),
                                _lib.PRECISION[precision], stash.data_ptr(), d_out.data_ptr(), d_grid.data_ptr(), wp, bp,
                                d_pos.data_ptr() if d_pos is not None else None, ws.data_ptr(), ws_bytes,
                                _stream(grid_cl)), 'lfgc_backward_f32')
    return d_grid, d_w, d_b, d_pos


class SampleDecodeFn(torch.autograd.Function):
    """model/Feature_Grid_Model.py:62-75 as one autograd node (HIP forward + HIP backward)."""

    @staticmethod
    def forward(ctx, desc, pos, grid_cl, packed, n_layers, precision, *params):
        weights, biases = params[:n_layers + 1], params[n_layers + 1:]
        need_grad = any(t.requires_grad for t in (pos, grid_cl) + tuple(params))
        y, stash = forward_raw(desc, grid_cl.detach(), packed, pos=pos.detach(), clamp=False, want_stash=need_grad,
                               precision=precision)
        if need_grad:
            ctx.desc = desc
            ctx.precision = precision
            ctx.n_layers = n_layers
            ctx.need_d_pos = pos.requires_grad
            ctx.save_for_backward(pos.detach(), grid_cl.detach(), packed, stash, *[p.detach() for p in params])
        return y.view(-1, 1)

    @staticmethod
    @once_differentiable
    def backward(ctx, d_y):
        pos, grid_cl, packed, stash = ctx.saved_tensors[:4]
        params = ctx.saved_tensors[4:]
        L = ctx.n_layers
        weights, biases = params[:L + 1], params[L + 1:]
        d_grid, d_w, d_b, d_pos = backward_raw(ctx.desc, grid_cl, packed, pos, stash, d_y.reshape(-1), weights, biases,
                                               ctx.need_d_pos, precision=ctx.precision)
        return (None, d_pos, d_grid, None, None, None) + tuple(d_w) + tuple(d_b)


# ---- binary checkpoint codec, device side -------------------------------------------

def _flat_f32(x: torch.Tensor) -> torch.Tensor:
    _require_hip(x)
    return _f32c(x.detach()).reshape(-1)


def _select_workspace(n: int, device) -> torch.Tensor:
    nbytes = int(_lib.load().lfgc_codec_select_workspace_bytes(int(n)))
    return torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=device)


@_on_device
def codec_mask(x: torch.Tensor) -> torch.Tensor:
    """uint8 (ceil(n/8),): bit i (MSB first) = x.flat[i] != 0."""
    x = _flat_f32(x)
    mask = torch.empty((x.numel() + 7) // 8, dtype=torch.uint8, device=x.device)
    check(_lib.load().lfgc_codec_mask_f32(x.data_ptr(), x.numel(), mask.data_ptr(), _stream(x)), 'lfgc_codec_mask_f32')
    return mask


@_on_device
def codec_compact(x: torch.Tensor) -> torch.Tensor:
    """The non-zero values of x in order (1-D).  Synchronises once to learn their number."""
    x = _flat_f32(x)
    out = torch.empty_like(x)
    count = torch.zeros(1, dtype=torch.int64, device=x.device)
    ws = _select_workspace(x.numel(), x.device)
    check(_lib.load().lfgc_codec_compact_f32(x.data_ptr(), x.numel(), out.data_ptr(), count.data_ptr(), ws.data_ptr(),
                                             ws.numel() * 8, _stream(x)), 'lfgc_codec_compact_f32')
    return out[:int(count.item())]


@_on_device
def codec_expand(mask: torch.Tensor, bit_offset: int, n: int, values: torch.Tensor) -> torch.Tensor:
    """(n,) fp32: values scattered to the set bits [bit_offset, bit_offset + n) of the MSB-first mask, zeros elsewhere."""
    _require_hip(mask, values)
    if mask.dtype != torch.uint8 or (int(bit_offset) + int(n) + 7) // 8 > mask.numel():
        raise ValueError('mask too short for %d bits at offset %d' % (n, bit_offset))
    values = _f32c(values).reshape(-1)
    if values.numel() == 0:
        values = torch.zeros(1, dtype=torch.float32, device=mask.device)
    out = torch.empty(int(n), dtype=torch.float32, device=mask.device)
    ws = _select_workspace(n, mask.device)
    check(_lib.load().lfgc_codec_expand_f32(mask.contiguous().data_ptr(), int(bit_offset), int(n), values.data_ptr(),
                                            out.data_ptr(), ws.data_ptr(), ws.numel() * 8, _stream(mask)), 'lfgc_codec_expand_f32')
    return out


@_on_device
def codec_kmeans(x: torch.Tensor, k: int = 256, iterations: int = 40) -> Tuple[torch.Tensor, torch.Tensor]:
    """(centres (k,) fp32 sorted, labels (n,) uint8) of the 1-D value set x.  Initial centres: Ward merging (host C++,
    lfgc_codec_ward_init_host) of a sorted strided sample of at most 2^16 values; then Lloyd iterations over all values
    on the GPU."""
    x = _flat_f32(x)
    n = x.numel()
    if n < 1 or not 1 <= k <= 256:
        raise ValueError('k-means needs at least one value and 1 <= k <= 256')
    lib = _lib.load()
    stride = max(1, -(-n // (1 << 16)))
    sample = np.ascontiguousarray(torch.sort(x[::stride])[0].cpu().numpy())
    init = np.empty(k, dtype=np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    check(lib.lfgc_codec_ward_init_host(sample.ctypes.data_as(fp), sample.size, int(k), init.ctypes.data_as(fp)),
          'lfgc_codec_ward_init_host')
    centres = torch.from_numpy(init).to(x.device)
    labels = torch.empty(n, dtype=torch.uint8, device=x.device)
    nbytes = int(lib.lfgc_codec_kmeans_workspace_bytes(int(k)))
    ws = torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=x.device)
    check(lib.lfgc_codec_kmeans1d_f32(x.data_ptr(), n, int(k), centres.data_ptr(), labels.data_ptr(), int(iterations),
                                      ws.data_ptr(), ws.numel() * 8, _stream(x)), 'lfgc_codec_kmeans1d_f32')
    return centres, labels


@_on_device
def codec_dequant(packed: torch.Tensor, bits: int, n: int, centres: torch.Tensor) -> torch.Tensor:
    """(n,) fp32 = centres[label_i], labels `bits` wide, MSB first, in the uint8 stream `packed`."""
    _require_hip(packed, centres)
    if packed.dtype != torch.uint8:
        raise ValueError('packed labels must be uint8')
    centres = _f32c(centres)
    if centres.numel() < (1 << int(bits)):
        raise ValueError('codebook smaller than 2^bits')
    out = torch.empty(int(n), dtype=torch.float32, device=packed.device)
    check(_lib.load().lfgc_codec_dequant_f32(packed.contiguous().data_ptr(), packed.numel(), int(bits), int(n),
                                             centres.data_ptr(), out.data_ptr(), _stream(packed)), 'lfgc_codec_dequant_f32')
    return out


# ---- ground truth / statistics -------------------------------------------------------------------------

@_on_device
def gt_interp(p: torch.Tensor, f: torch.Tensor, min_bb, max_bb, res) -> torch.Tensor:
    lib = _lib.load()
    _require_hip(p, f)
    p, f = _f32c(p), _f32c(f)
    arr = lambda v: (ctypes.c_float * 3)(*[float(x) for x in (v.tolist() if hasattr(v, 'tolist') else v)])
    out = torch.empty(p.shape[0], dtype=torch.float32, device=p.device)
    X, Y, Z = f.shape
    check(lib.lfgc_gt_interp_f32(p.data_ptr(), f.data_ptr(), arr(min_bb), arr(max_bb), arr(res), p.shape[0], X, Y, Z,
                                 out.data_ptr(), _stream(p)), 'lfgc_gt_interp_f32')
    return out


class GtMseLossFn(torch.autograd.Function):
    """loss = MSELoss()(pred, trilinear_f_interpolation(p, f, ...)) as one node: ground truth, squared error and the
    gradient of the mean in one kernel (+ a one-workgroup fold).  apply(pred (N,), p (N,3), f (X,Y,Z), min_bb, max_bb, res)
    with the three bounds as host sequences of 3 floats."""

    @staticmethod
    @_on_device
    def forward(ctx, pred, p, f, min_bb, max_bb, res):
        lib = _lib.load()
        _require_hip(pred, p, f)
        pred_c, p, f = _f32c(pred.detach()).reshape(-1), _f32c(p.detach()), _f32c(f.detach())
        n = pred_c.numel()
        if p.shape != (n, 3):
            raise ValueError('positions %s do not match %d predictions' % (tuple(p.shape), n))
        arr = lambda v: (ctypes.c_float * 3)(*[float(x) for x in (v.tolist() if hasattr(v, 'tolist') else v)])
        d_pred = torch.empty_like(pred_c)
        loss = torch.empty((), dtype=torch.float32, device=pred_c.device)
        ws = torch.empty(int(lib.lfgc_gt_mse_workspace_bytes(n)) // 8, dtype=torch.float64, device=pred_c.device)
        X, Y, Z = f.shape
        check(lib.lfgc_gt_mse_f32(p.data_ptr(), f.data_ptr(), arr(min_bb), arr(max_bb), arr(res), n, X, Y, Z,
                                  pred_c.data_ptr(), None, d_pred.data_ptr(), loss.data_ptr(), ws.data_ptr(),
                                  ws.numel() * 8, _stream(pred_c)), 'lfgc_gt_mse_f32')
        ctx.save_for_backward(d_pred)
        ctx.shape = pred.shape
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (d_pred,) = ctx.saved_tensors
        unit = _UNIT_GRADS.get(g.device)
        if unit is not None and g.data_ptr() == unit.data_ptr():      # unit_grad(): known to be 1.0
            return d_pred.view(ctx.shape), None, None, None, None, None
        return (d_pred * g).view(ctx.shape), None, None, None, None, None


def gt_mse_loss(pred, p, f, min_bb, max_bb, res) -> torch.Tensor:
    return GtMseLossFn.apply(pred, p, f, min_bb, max_bb, res)


@_on_device
def lattice_positions(flat: torch.Tensor, res, min_idx, max_idx, scales) -> Tuple[torch.Tensor, torch.Tensor]:
    """(raw (N,3), norm (N,3)) for flat voxel indices on the device: lfgc_lattice_positions_f32."""
    _require_hip(flat)
    flat = flat.to(torch.int64).contiguous()
    n = flat.numel()
    raw = torch.empty((n, 3), dtype=torch.float32, device=flat.device)
    norm = torch.empty((n, 3), dtype=torch.float32, device=flat.device)
    f3 = lambda v: (ctypes.c_float * 3)(*[float(x) for x in (v.tolist() if hasattr(v, 'tolist') else v)])
    r3 = (ctypes.c_int32 * 3)(*[int(x) for x in res])
    check(_lib.load().lfgc_lattice_positions_f32(flat.data_ptr(), n, r3, f3(min_idx), f3(max_idx), f3(scales),
                                                 raw.data_ptr(), norm.data_ptr(), _stream(flat)), 'lfgc_lattice_positions_f32')
    return raw, norm


@_on_device
def lattice_sample(state: torch.Tensor, n: int, seed: int, res, min_idx, max_idx, scales,
                   want_flat: bool = False):
    """(raw (N,3), norm (N,3)[, flat (N) int64]) for N voxel indices drawn uniformly (with replacement) by the kernel
    itself: lfgc_lattice_sample_f32.  `state`: device int64[2], zeroed once by the caller and then left alone -- the
    draw counter lives in it and advances on the device, so a captured call draws a new batch on every graph replay."""
    _require_hip(state)
    if state.dtype != torch.int64 or state.numel() != 2 or not state.is_contiguous():
        raise ValueError('state must be a contiguous int64 tensor of 2 elements')
    n = int(n)
    raw = torch.empty((n, 3), dtype=torch.float32, device=state.device)
    norm = torch.empty((n, 3), dtype=torch.float32, device=state.device)
    flat = torch.empty(n, dtype=torch.int64, device=state.device) if want_flat else None
    f3 = lambda v: (ctypes.c_float * 3)(*[float(x) for x in (v.tolist() if hasattr(v, 'tolist') else v)])
    r3 = (ctypes.c_int32 * 3)(*[int(x) for x in res])
    check(_lib.load().lfgc_lattice_sample_f32(int(seed) & 0xFFFFFFFFFFFFFFFF, state.data_ptr(), n, r3, f3(min_idx), f3(max_idx),
                                              f3(scales), raw.data_ptr(), norm.data_ptr(),
                                              flat.data_ptr() if want_flat else None, _stream(state)),
          'lfgc_lattice_sample_f32')
    return (raw, norm, flat) if want_flat else (raw, norm)


# Gradient seed of a scalar loss that GtMseLossFn.backward recognises by identity and does not multiply by: one fill
# and one elementwise launch less per train step than `loss.backward()` (autograd's ones_like + `d_pred * g`).  Read-only.
_UNIT_GRADS = {}


def unit_grad(device) -> torch.Tensor:
    """0-d fp32 tensor holding 1.0 on `device` (cached; create it before capturing a graph): `loss.backward(unit_grad(dev))`."""
    device = torch.device(device)
    if device.type == 'cuda' and device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    t = _UNIT_GRADS.get(device)
    if t is None:
        t = _UNIT_GRADS[device] = torch.ones((), dtype=torch.float32, device=device)
    return t


@_on_device
def deviation_partial(pred: torch.Tensor, gt: torch.Tensor, acc: Optional[torch.Tensor] = None) -> torch.Tensor:
    """acc = [sum sq, sum abs, min gt, max gt] (fp64, device)."""
    lib = _lib.load()
    _require_hip(pred, gt)
    pred, gt = _f32c(pred).reshape(-1), _f32c(gt).reshape(-1)
    if acc is None:
        acc = torch.tensor([0.0, 0.0, float('inf'), float('-inf')], dtype=torch.float64, device=pred.device)
    check(lib.lfgc_deviation_partial_f32(pred.data_ptr(), gt.data_ptr(), pred.numel(), acc.data_ptr(), _stream(pred)),
          'lfgc_deviation_partial_f32')
    return acc
