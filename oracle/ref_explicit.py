"""ORACLE (test infrastructure, NOT product code) -- explicit-index numpy restatement (fp64 by
default) of the arithmetic that the reference delegates to ATen ops.

``ref_torch.py`` restates the reference op-for-op with the same ATen calls; this file writes out
what those calls compute as index formulas, which is exactly what the HIP
kernels implement.  It exists (a) to prove the index formulas against ``ref_torch`` on CPU before a
kernel is written to them and (b) to give the GPU parity tests an fp64 "truth" so that the HIP
kernel's rounding error can be compared with the fp32 CPU path's own rounding error.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import it.
"""
from __future__ import annotations

import numpy as np


def idwt_level(lll: np.ndarray, hf: np.ndarray, filter_rev: np.ndarray, target) -> np.ndarray:
    """One inverse-DWT level = grouped conv_transpose3d(stride 2, 4 taps) + crop
    (wavelet_transform/Torch_Wavelet_Transform.py:91-104, :69-73).

    lll (C, d0,d1,d2), hf (C, 7, d0,d1,d2), filter_rev (8,1,4,4,4), target (3,) -> (C, *target).
    out_full[c, o] = sum_s sum_t in[c, s, i] * F_s[t] with o = 2 i + t per axis (edge 2 d + 2);
    crop axis a to [floor(delta/2), floor(delta/2) + target), delta = 2 d + 2 - target.
    """
    dt = np.float64
    C = lll.shape[0]
    d = lll.shape[1:]
    sub = np.concatenate([lll[:, None], hf], axis=1).astype(dt)          # (C, 8, d0,d1,d2)
    full = np.zeros((C, 2 * d[0] + 2, 2 * d[1] + 2, 2 * d[2] + 2), dt)
    F = filter_rev.astype(dt)[:, 0]                                      # (8,4,4,4)
    for tz in range(4):
        for ty in range(4):
            for tx in range(4):
                contrib = np.einsum('cszyx,s->czyx', sub, F[:, tz, ty, tx])
                full[:, tz:tz + 2 * d[0]:2, ty:ty + 2 * d[1]:2, tx:tx + 2 * d[2]:2] += contrib
    off = [(2 * d[a] + 2 - int(target[a])) // 2 for a in range(3)]
    return full[:, off[0]:off[0] + int(target[0]), off[1]:off[1] + int(target[1]), off[2]:off[2] + int(target[2])]


def decode_volume(coeffs, shape_array, filter_rev) -> np.ndarray:
    """model/Feature_Grid_Model.py:102-108 with Identity drop layers."""
    restored = np.asarray(coeffs[0], np.float64)
    for hf, shape in zip(coeffs[1:], shape_array):
        restored = idwt_level(restored, np.asarray(hf, np.float64), np.asarray(filter_rev), shape)
    return restored


def dwt_level(data: np.ndarray, filter_fwd: np.ndarray) -> np.ndarray:
    """One forward-DWT level = zero-pad (2, 2 + odd) + grouped conv3d stride 2
    (Torch_Wavelet_Transform.py:59-67, :75-89).  data (C, n0,n1,n2) -> (C, 8, d0,d1,d2)."""
    dt = np.float64
    C = data.shape[0]
    n = data.shape[1:]
    odd = [n[a] % 2 for a in range(3)]
    # reference quirk: F.pad slot order is (last axis first); odd bit of axis a lands on axis 2-a.
    hi = [2 + odd[2 - a] for a in range(3)]
    padded = np.zeros((C, n[0] + 2 + hi[0], n[1] + 2 + hi[1], n[2] + 2 + hi[2]), dt)
    padded[:, 2:2 + n[0], 2:2 + n[1], 2:2 + n[2]] = data
    d = [(padded.shape[1 + a] - 4) // 2 + 1 for a in range(3)]
    F = filter_fwd.astype(dt)[:, 0]
    out = np.zeros((C, 8, d[0], d[1], d[2]), dt)
    for tz in range(4):
        for ty in range(4):
            for tx in range(4):
                win = padded[:, tz:tz + 2 * d[0]:2, ty:ty + 2 * d[1]:2, tx:tx + 2 * d[2]:2]
                out += win[:, None] * F[None, :, tz, ty, tx, None, None, None]
    return out


def sample_grid(dense: np.ndarray, pos: np.ndarray, dtype=np.float64) -> np.ndarray:
    """F.grid_sample(bilinear, align_corners=False, padding zeros) written out.
    dense (C, D, H, W), pos (N,3) with pos[:,0]->W, [:,1]->H, [:,2]->D.  -> (N, C)."""
    C, D, H, W = dense.shape
    p = pos.astype(dtype)
    ix = ((p[:, 0] + 1) * W - 1) / 2
    iy = ((p[:, 1] + 1) * H - 1) / 2
    iz = ((p[:, 2] + 1) * D - 1) / 2
    x0, y0, z0 = np.floor(ix), np.floor(iy), np.floor(iz)
    fx, fy, fz = ix - x0, iy - y0, iz - z0
    x0, y0, z0 = x0.astype(np.int64), y0.astype(np.int64), z0.astype(np.int64)
    out = np.zeros((pos.shape[0], C), dtype)
    g = dense.astype(dtype)
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                w = (fx if dx else 1 - fx) * (fy if dy else 1 - fy) * (fz if dz else 1 - fz)
                xi, yi, zi = x0 + dx, y0 + dy, z0 + dz
                valid = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H) & (zi >= 0) & (zi < D)
                v = g[:, np.clip(zi, 0, D - 1), np.clip(yi, 0, H - 1), np.clip(xi, 0, W - 1)]   # (C, N)
                out += np.where(valid, w, 0.0)[:, None] * v.T
    return out


def fourier_embed(pos: np.ndarray, n_freqs: int, dtype=np.float64) -> np.ndarray:
    """model/Feature_Embedding.py:27-34.  The frequencies are fp32 values and the product p*f is
    formed in fp32 in the reference; here the fp32 product is reproduced and only sin/cos run in
    ``dtype``."""
    outs = []
    for k in range(n_freqs):
        f = np.float32(np.float32(2.0 ** k) * np.float32(2.0) * np.float32(np.pi))
        arg = (pos.astype(np.float32) * f).astype(dtype)
        outs += [np.sin(arg), np.cos(arg)]
    if not outs:
        return np.zeros((pos.shape[0], 0), dtype)
    return np.concatenate(outs, -1)


def mlp_forward(x: np.ndarray, weights, biases, dtype=np.float64):
    """model/Feature_Grid_Model.py:72-75.  Returns (y, [pre-activations])."""
    x = x.astype(dtype)
    pre = []
    for w, b in zip(weights[:-1], biases[:-1]):
        a = x @ np.asarray(w, dtype).T + np.asarray(b, dtype)
        pre.append(a)
        x = 0.5 * a + np.sin(a) ** 2
    y = x @ np.asarray(weights[-1], dtype).T + np.asarray(biases[-1], dtype)
    return y, pre


def forward_from_grid(dense, weights, biases, pos, n_freqs, dtype=np.float64):
    feat = sample_grid(np.asarray(dense), np.asarray(pos), dtype)
    emb = fourier_embed(np.asarray(pos), n_freqs, dtype)
    x = np.concatenate([np.asarray(pos, dtype), emb, feat], -1)
    return mlp_forward(x, weights, biases, dtype)[0]


def gt_gather_lattice(volume: np.ndarray, raw_pos: np.ndarray) -> np.ndarray:
    """What data/Interpolation.py:8-44 degenerates to for lattice inputs on cubic volumes."""
    idx = np.asarray(raw_pos).astype(np.int64)
    return np.asarray(volume)[idx[:, 0], idx[:, 1], idx[:, 2]]
