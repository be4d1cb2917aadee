"""ORACLE (test infrastructure, NOT product code) -- pure-PyTorch CPU restatement of the
latent-feature-grid sample/decode path of Bussler/Latent_Feature_Grid_Compression.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product package (``latent_feature_grid_compression_amd``) never does.

Every function restates one reference function op-for-op (the same ATen ops in the same order,
so that on CPU the result is bit-identical to the reference's own code) and cites it.  Citations are
relative to the reference checkout (``/root/reference`` in the build container).

Pinning: the reference's own tests hold no golden vector for this path, so
this restatement is pinned by fixtures captured from the reference's own modules imported in the
build container (``tools/make_goldens.py`` -> ``tests/golden/*.npz``; see DESIGN.md "Oracle").
``tests/test_oracle_golden.py`` checks every function below against those fixtures.
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------------
# db2 filter bank.  The reference reads it from PyWavelets (``pywt.Wavelet('db2').filter_bank``,
# wavelet_transform/Torch_Wavelet_Transform.py:14,41; Env.txt:178 pins PyWavelets==1.4.1).  The 16
# constants below are the published Daubechies-2 taps as PyWavelets returns them (read from a real
# pywt 1.1.1 in the build container by tools/dump_pywt_constants.py -> tests/golden/pywt_db2.json).
# --------------------------------------------------------------------------------------------
DB2_DEC_LO = [-0.12940952255126037, 0.2241438680420134, 0.8365163037378079, 0.48296291314453416]
DB2_DEC_HI = [-0.48296291314453416, 0.8365163037378079, -0.2241438680420134, -0.12940952255126037]
DB2_REC_LO = [0.48296291314453416, 0.8365163037378079, 0.2241438680420134, -0.12940952255126037]
DB2_REC_HI = [-0.12940952255126037, -0.2241438680420134, 0.8365163037378079, -0.48296291314453416]


def dwt_max_level(data_len: int, filter_len: int) -> int:
    """pywt.dwt_max_level (used at model/Feature_Grid_Model.py:85): floor(log2(n / (flen - 1)))."""
    if filter_len < 2 or data_len < filter_len - 1:
        return 0
    return max(0, int(math.floor(math.log2(data_len / (filter_len - 1.0)))))


def build_filters(dim: int = 3) -> Tuple[torch.Tensor, torch.Tensor]:
    """_WaveletFilterNd._register_filters (wavelet_transform/Torch_Wavelet_Transform.py:39-57).

    Returns (filter_fwd, filter_rev), each (2**dim, 1, 4, ..., 4) float32.  Sub-band index
    s = 4a + 2b + c with a/b/c the filter (0 = low, 1 = high) along the first/second/third spatial
    axis; 1-D taps are cast to fp32 first (torch.tensor of python floats, :41) and the outer products
    are formed in fp32 (:44-53).
    """
    fwd_low, fwd_high, rev_low, rev_high = (torch.tensor(x) for x in
                                            (DB2_DEC_LO, DB2_DEC_HI, DB2_REC_LO, DB2_REC_HI))

    def build_ndim_filter(filters_1d: List[torch.Tensor]) -> torch.Tensor:
        out = [f.unsqueeze(0) for f in filters_1d]
        f1d = filters_1d
        for _ in range(1, dim):
            f1d = [f.unsqueeze(-1) for f in f1d]
            out = [f1 * fn for f1 in f1d for fn in out]
        return torch.stack(out, dim=0).unsqueeze(1)

    filter_fwd = build_ndim_filter([fwd_low.flip(-1), fwd_high.flip(-1)])
    filter_rev = build_ndim_filter([rev_low, rev_high])
    return filter_fwd, filter_rev


def padding_size(shape: np.ndarray, filter_length: int = 4) -> Tuple[int, ...]:
    """_get_padding_size (Torch_Wavelet_Transform.py:59-63).  NOTE: F.pad pads the LAST dim first,
    and the reference adds the odd-bit of shape[i] to pad slot 2i+1, i.e. the odd bit of the FIRST
    spatial axis goes to the high side of the LAST axis.  Irrelevant for cubes; kept verbatim."""
    is_odd = np.asarray(shape) % 2 == 1
    out = np.full(6, (2 * filter_length - 3) // 2, dtype=int)
    out[1::2] += is_odd.astype(int)
    return tuple(int(v) for v in out)


def wavelet_encode(data: torch.Tensor, filter_fwd: torch.Tensor):
    """_WaveletFilterNd.encode (Torch_Wavelet_Transform.py:75-89): zero-pad, grouped conv3d stride 2.
    data (B, C, D, H, W) -> coeffs (B, C, 8, d, h, w), shape (pre-pad spatial shape)."""
    shape = np.asarray(data.shape[-3:])
    padded = F.pad(data, padding_size(shape, filter_fwd.shape[-1]), mode='constant')
    b, c = padded.shape[:2]
    weight = filter_fwd.repeat(c, 1, 1, 1, 1)
    result = F.conv3d(padded, weight, stride=2, groups=c)
    coeffs = result.reshape(b, c, 8, *result.shape[2:])
    return coeffs, shape


def wavelet_decode(data: torch.Tensor, shape: Sequence[int], filter_rev: torch.Tensor) -> torch.Tensor:
    """_WaveletFilterNd.decode + _unpad_for_reverse (Torch_Wavelet_Transform.py:91-104, :69-73):
    grouped conv_transpose3d stride 2 on (B, C*8, d,h,w), then crop each axis to
    [floor(diff/2), -ceil(diff/2) or None)."""
    b, c = data.shape[:2]
    weight = filter_rev.repeat(c, 1, 1, 1, 1)
    result = F.conv_transpose3d(torch.flatten(data, start_dim=1, end_dim=2), weight, groups=c, stride=2)
    difference = np.asarray(result.shape[-3:]) - np.asarray(shape)
    slices = tuple(slice(int(np.floor(d / 2)), -int(np.ceil(d / 2)) or None, None) for d in difference)
    return result[(slice(None), slice(None)) + slices]


def encode_volume(feature_volume: torch.Tensor, filter_fwd: torch.Tensor, num_levels=None):
    """Feature_Grid_Model.encode_volume (model/Feature_Grid_Model.py:83-99).
    (C, G, G, G) -> ([coarse (C,d0,d0,d0), hf_1 (C,7,d,d,d), ...], shape_array (levels, 3) int)."""
    if num_levels is None:
        num_levels = min(dwt_max_level(s, filter_fwd.shape[-1]) for s in feature_volume.shape[-3:])
    features, shapes = [], []
    data = feature_volume.detach().unsqueeze(0)
    for _ in range(num_levels):
        filtered, shape = wavelet_encode(data, filter_fwd)
        features.append(filtered[0, :, 1:])
        shapes.append(shape)
        data = filtered[:, :, 0]
    features = [data[0]] + [*reversed(features)]
    shape_array = np.asarray(shapes[::-1], dtype=int)
    return features, shape_array


def decode_volume(coeffs: Sequence[torch.Tensor], shape_array: np.ndarray, filter_rev: torch.Tensor) -> torch.Tensor:
    """Feature_Grid_Model.decode_volume (model/Feature_Grid_Model.py:102-108) with Identity drop
    layers (every BASELINE config: drop_type '').  Returns the dense grid (C, G, G, G)."""
    restored = coeffs[0].unsqueeze(0)
    for high_freq, shape in zip(coeffs[1:], shape_array):
        data = torch.cat([restored.unsqueeze(2), high_freq.unsqueeze(0)], dim=2)
        restored = wavelet_decode(data, shape, filter_rev)
    return restored[0]


def fourier_freqs(n_freqs: int) -> torch.Tensor:
    """FourierEmbedding.create_embedding_function (model/Feature_Embedding.py:27-29):
    2**linspace(0, n-1, n) * 2. * np.pi, as a float32 tensor (so f_k = fp32(fp32(2**k)*2)*fp32(pi)...
    evaluated by torch in fp32)."""
    freq_bands = 2. ** torch.linspace(0., n_freqs - 1, steps=n_freqs)
    return freq_bands * 2. * np.pi


def fourier_embed(inputs: torch.Tensor, n_freqs: int) -> torch.Tensor:
    """Embedder.embed (model/Feature_Embedding.py:14-16, :31-34): cat_k [sin(x*f_k), cos(x*f_k)]."""
    outs = []
    for freq in fourier_freqs(n_freqs):
        for p_fn in (torch.sin, torch.cos):
            outs.append(p_fn(inputs * freq))
    if not outs:
        return inputs.new_zeros(*inputs.shape[:-1], 0)
    return torch.cat(outs, -1)


def snake_alt(x: torch.Tensor) -> torch.Tensor:
    """SnakeAlt (model/Feature_Grid_Model.py:12-13)."""
    return 0.5 * x + torch.sin(x) ** 2


def sample_grid(dense_grid: torch.Tensor, pos: torch.Tensor) -> torch.Tensor:
    """model/Feature_Grid_Model.py:62-64: F.grid_sample(bilinear, align_corners=False, zeros padding)
    of the (C,D,H,W) grid at (N,3) positions -> (N, C).  pos[:,0] indexes W (last axis)."""
    grid = pos.view(1, 1, 1, *pos.shape)
    out = F.grid_sample(dense_grid.unsqueeze(0), grid, mode='bilinear', align_corners=False)
    return out.reshape(dense_grid.shape[0], pos.shape[0]).transpose(0, 1)


def mlp_forward(x: torch.Tensor, weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor],
                return_preacts: bool = False):
    """model/Feature_Grid_Model.py:72-75: L x (Linear + SnakeAlt), then final Linear.
    weights/biases hold L hidden layers followed by the final layer, nn.Linear layout (out, in)."""
    pre = []
    for w, b in zip(weights[:-1], biases[:-1]):
        a = F.linear(x, w, b)
        if return_preacts:
            pre.append(a)
        x = snake_alt(a)
    y = F.linear(x, weights[-1], biases[-1])
    return (y, pre) if return_preacts else y


def forward_from_grid(dense_grid: torch.Tensor, weights, biases, pos: torch.Tensor, n_freqs: int,
                      return_parts: bool = False):
    """Feature_Grid_Model.forward, train branch, after decode (model/Feature_Grid_Model.py:62-75).
    pos (N,3) -> (N,1)."""
    feat = sample_grid(dense_grid, pos)
    emb = fourier_embed(pos, n_freqs)
    x = torch.cat([pos, emb, feat], -1)
    if return_parts:
        y, pre = mlp_forward(x, weights, biases, return_preacts=True)
        return y, dict(feat=feat, emb=emb, pre=pre)
    return mlp_forward(x, weights, biases)


def forward(coeffs, shape_array, filter_rev, weights, biases, pos: torch.Tensor, n_freqs: int,
            training: bool = True) -> torch.Tensor:
    """Feature_Grid_Model.forward (model/Feature_Grid_Model.py:50-80).

    Eval branch = the INTENDED semantics of :57-60/:77-78 (flatten leading dims, run, restore
    ``input.shape[:-1] + (1,)``, clamp to [-1,1]); as written the reference's eval branch raises
    TypeError on torch >= 2."""
    dense = decode_volume(coeffs, shape_array, filter_rev)
    if training:
        return forward_from_grid(dense, weights, biases, pos, n_freqs)
    orig_shape = pos.shape
    y = forward_from_grid(dense, weights, biases, pos.reshape(-1, orig_shape[-1]), n_freqs)
    return y.view(*orig_shape[:-1], 1).clamp(-1, 1)


def trilinear_f_interpolation(p, f, min_bb, max_bb, res):
    """data/Interpolation.py:8-44, verbatim arithmetic (fp32 lattice coords, fp64 alpha, fp32 lerps
    x then y then z)."""
    normalized_p = ((p - min_bb.unsqueeze(0)) / ((max_bb - min_bb).unsqueeze(0))) * (res.unsqueeze(0) - 1)
    lo = torch.floor(normalized_p).to(torch.long)
    hi = torch.ceil(normalized_p).to(torch.long)
    min_ref = 1e-12 * torch.ones_like(normalized_p[:1, 0])
    the_diff = torch.max((hi - lo).to(torch.double), min_ref.to(torch.double))
    alpha = (normalized_p.to(torch.double) - lo.to(torch.double)) / the_diff
    alpha = alpha.to(torch.float)
    one_alpha = 1.0 - alpha
    x_y0z0 = one_alpha[:, 0] * f[lo[:, 0], lo[:, 1], lo[:, 2]] + alpha[:, 0] * f[hi[:, 0], lo[:, 1], lo[:, 2]]
    x_y1z0 = one_alpha[:, 0] * f[lo[:, 0], hi[:, 1], lo[:, 2]] + alpha[:, 0] * f[hi[:, 0], hi[:, 1], lo[:, 2]]
    x_y0z1 = one_alpha[:, 0] * f[lo[:, 0], lo[:, 1], hi[:, 2]] + alpha[:, 0] * f[hi[:, 0], lo[:, 1], hi[:, 2]]
    x_y1z1 = one_alpha[:, 0] * f[lo[:, 0], hi[:, 1], hi[:, 2]] + alpha[:, 0] * f[hi[:, 0], hi[:, 1], hi[:, 2]]
    y_z0 = one_alpha[:, 1] * x_y0z0 + alpha[:, 1] * x_y1z0
    y_z1 = one_alpha[:, 1] * x_y0z1 + alpha[:, 1] * x_y1z1
    return one_alpha[:, 2] * y_z0 + alpha[:, 2] * y_z1


# ---- caller-side arithmetic that defines the inputs (harness counterpart) ----------

def finite_difference_trilinear_grad(p, f, min_bb, max_bb, res, scale=None):
    """data/Interpolation.py:47-84: central differences of trilinear_f_interpolation with one lattice step per axis,
    steps clamped to the bounding box; optional per-axis scale of the step length.  (N,3)."""
    step = (max_bb - min_bb) / (res - 1)
    cols = []
    for a in range(3):
        lo, hi = p.clone(), p.clone()
        lo[:, a] = p[:, a] - step[a]
        hi[:, a] = p[:, a] + step[a]
        lo[lo[:, a] < min_bb[a], a] = min_bb[a]
        hi[hi[:, a] > max_bb[a], a] = max_bb[a]
        if scale is None:
            diff = 2 * (hi[:, a] - lo[:, a]) / (max_bb[a] - min_bb[a])
        else:
            diff = 2 * scale[a] * (hi[:, a] - lo[:, a]) / (max_bb[a] - min_bb[a])
        cols.append((trilinear_f_interpolation(hi, f, min_bb, max_bb, res) -
                     trilinear_f_interpolation(lo, f, min_bb, max_bb, res)) / diff)
    return torch.cat([c.unsqueeze(1) for c in cols], 1)


def normalize_volume(volume, min_v, max_v, min_n, max_n):
    """data/IndexDataset.py:7-8."""
    return (max_n - min_n) * ((volume - min_v) / (max_v - min_v)) + min_n


def generate_indices(start, end, res) -> torch.Tensor:
    """IndexDataset.generate_indices (data/IndexDataset.py:69-76)."""
    r = [int(v) for v in res]
    out = torch.zeros(r[0], r[1], r[2], 3)
    out[:, :, :, 0] = torch.linspace(float(start[0]), float(end[0]), r[0], dtype=torch.float).view(r[0], 1, 1)
    out[:, :, :, 1] = torch.linspace(float(start[1]), float(end[1]), r[1], dtype=torch.float).view(1, r[1], 1)
    out[:, :, :, 2] = torch.linspace(float(start[2]), float(end[2]), r[2], dtype=torch.float).view(1, 1, r[2])
    return out


class VolumeIndexing:
    """The fields of IndexDataset the drivers read (data/IndexDataset.py:51-65)."""

    def __init__(self, vol_shape):
        self.vol_res = torch.tensor(vol_shape, dtype=torch.float)
        self.vol_res_touple = tuple(int(v) for v in vol_shape)
        self.n_voxels = int(np.prod(self.vol_res_touple))
        self.min_idx = torch.tensor([0.0, 0.0, 0.0], dtype=torch.float)
        self.max_idx = torch.tensor([self.vol_res[0] - 1, self.vol_res[1] - 1, self.vol_res[2] - 1],
                                    dtype=torch.float)
        self.max_dim = torch.max(self.max_idx)
        self.scales = self.max_idx / self.max_dim

    def training_positions(self, idx: torch.Tensor):
        """IndexDataset.__getitem__ (data/IndexDataset.py:90-96) for explicit lattice indices
        idx (N,3) int -> (raw_positions float (N,3), normalized positions (N,3))."""
        raw = idx.to(torch.float)
        norm = normalize_volume(raw, self.min_idx.unsqueeze(0), self.max_idx.unsqueeze(0), -1.0, 1.0)
        return raw, self.scales.unsqueeze(0) * norm


def tile_iter(target_res, tiled_res: int = 32):
    """Tile loop bounds of field_from_net (visualization/OutputToVTK.py:11-21)."""
    for xb in range(0, target_res[0], tiled_res):
        xe = min(xb + tiled_res, target_res[0])
        for yb in range(0, target_res[1], tiled_res):
            ye = min(yb + tiled_res, target_res[1])
            for zb in range(0, target_res[2], tiled_res):
                ze = min(zb + tiled_res, target_res[2])
                yield (xb, xe, yb, ye, zb, ze)


def tile_positions(ds: VolumeIndexing, bounds) -> torch.Tensor:
    """Per-tile lattice of field_from_net (visualization/OutputToVTK.py:23-37): (x,y,z,3) fp32."""
    xb, xe, yb, ye, zb, ze = bounds
    tr = ds.vol_res_touple
    tile_resolution = torch.tensor([xe - xb, ye - yb, ze - zb], dtype=torch.int)
    min_alpha = torch.tensor([xb / (tr[0] - 1), yb / (tr[1] - 1), zb / (tr[2] - 1)], dtype=torch.float)
    max_alpha = torch.tensor([(xe - 1) / (tr[0] - 1), (ye - 1) / (tr[1] - 1), (ze - 1) / (tr[2] - 1)],
                             dtype=torch.float)
    min_bounds = ds.min_idx + min_alpha * (ds.max_idx - ds.min_idx)
    max_bounds = ds.min_idx + max_alpha * (ds.max_idx - ds.min_idx)
    start = min_bounds / (ds.max_idx - ds.min_idx)
    end = max_bounds / (ds.max_idx - ds.min_idx)
    norm_indices = generate_indices(start, end, tile_resolution)
    norm_indices = 2.0 * norm_indices - 1.0
    return ds.scales.view(1, 1, 1, 3) * norm_indices


def field_from_net(ds: VolumeIndexing, net_fn, tiled_res: int = 32) -> torch.Tensor:
    """field_from_net (visualization/OutputToVTK.py:7-47); ``net_fn`` maps a (1,x,y,z,3) tensor to
    (1,x,y,z,1) (the eval-mode forward)."""
    full_vol = torch.zeros(ds.vol_res_touple)
    for b in tile_iter(ds.vol_res_touple, tiled_res):
        tp = tile_positions(ds, b)
        with torch.no_grad():
            tile_vol = net_fn(tp.unsqueeze(0)).squeeze(0).squeeze(-1)
        full_vol[b[0]:b[1], b[2]:b[3], b[4]:b[5]] = tile_vol
    return full_vol


def deviation_statistics(prediction: torch.Tensor, ground_truth: torch.Tensor):
    """calculate_deviation_statistics (visualization/OutputToVTK.py:53-60) -> (psnr, l1, mse, rmse)."""
    diff_vol = ground_truth - prediction
    sqd_max_diff = (torch.max(ground_truth) - torch.min(ground_truth)) ** 2
    l1_diff = torch.mean(torch.abs(diff_vol))
    mse = torch.mean(torch.pow(diff_vol, 2.0))
    psnr = 10 * torch.log10(sqd_max_diff / mse)
    return psnr.item(), l1_diff.item(), mse.item(), torch.sqrt(mse).item()


# ---- synthetic model construction shared by tests / smoke / bench (seeded, box-independent) -------

def synth_model(C: int, G: int, H: int, L: int, n_freqs: int = 2, seed: int = 0, num_levels=None,
                d_in: int = 3, d_out: int = 1):
    """Random-init parameters shaped like model_utils.setup_model builds them (model/model_utils.py:
    27-28 grid U(0,1); nn.Linear default init U(+-1/sqrt(fan_in))) from a numpy PCG64 stream so the
    build container and the GPU box agree bit-for-bit.  Returns a dict of fp32 CPU tensors."""
    rng = np.random.Generator(np.random.PCG64(seed))
    grid = torch.from_numpy(rng.random((C, G, G, G), dtype=np.float32))
    filter_fwd, filter_rev = build_filters(3)
    coeffs, shape_array = encode_volume(grid, filter_fwd, num_levels=num_levels)
    k0 = d_in + 2 * n_freqs * d_in + C
    dims = [k0] + [H] * L + [d_out]
    weights, biases = [], []
    for fan_in, fan_out in zip(dims[:-1], dims[1:]):
        bound = 1.0 / math.sqrt(fan_in)
        weights.append(torch.from_numpy(rng.uniform(-bound, bound, (fan_out, fan_in)).astype(np.float32)))
        biases.append(torch.from_numpy(rng.uniform(-bound, bound, (fan_out,)).astype(np.float32)))
    return dict(grid=grid, coeffs=[c.contiguous() for c in coeffs], shape_array=shape_array,
                filter_fwd=filter_fwd, filter_rev=filter_rev, weights=weights, biases=biases,
                n_freqs=n_freqs, C=C, G=G, H=H, L=L)
