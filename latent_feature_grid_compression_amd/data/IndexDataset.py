"""Host-side lattice bookkeeping the drivers need: mirror of the arithmetic in data/IndexDataset.py
(normalize_volume :7-8, IndexDataset.__init__ :51-65, generate_indices :69-76, __getitem__ :90-96).
File loaders (.npy/.h5/.cvol) are out of scope; volumes are handed over as tensors."""
from __future__ import annotations

import torch

from .. import ops


def normalize_volume(volume, minV, maxV, minN, maxN):
    return (maxN - minN) * ((volume - minV) / (maxV - minV)) + minN


def normalize_to_unit_range(volume: torch.Tensor) -> torch.Tensor:
    """get_tensor_from_numpy's normalisation (:15-17): min-max to [-1, 1]."""
    return normalize_volume(volume, torch.min(volume), torch.max(volume), -1.0, 1.0)


class IndexDataset(torch.utils.data.Dataset):
    def __init__(self, volume, sampleSize=16, build_index_table=True):
        shape = tuple(int(v) for v in (volume.shape if hasattr(volume, 'shape') else volume))
        self.vol_res = torch.tensor(shape, dtype=torch.float)
        self.vol_res_touple = shape
        self.n_voxels = int(shape[0]) * int(shape[1]) * int(shape[2])
        self.min_idx = torch.tensor([0.0, 0.0, 0.0], dtype=torch.float)
        self.max_idx = torch.tensor([self.vol_res[0] - 1, self.vol_res[1] - 1, self.vol_res[2] - 1], dtype=torch.float)
        # the reference materialises an (n_voxels, 3) table (12.9 GB at 1024^3); optional here
        self.volume_indices = (self.generate_indices(self.min_idx, self.max_idx, self.vol_res.int()).view(-1, 3)
                               if build_index_table else None)
        self.sample_size = sampleSize
        self.max_dim = torch.max(self.max_idx)
        self.scales = self.max_idx / self.max_dim

    def generate_indices(self, start, end, res):
        r = [int(v) for v in res]
        out = torch.zeros(r[0], r[1], r[2], 3)
        out[:, :, :, 0] = torch.linspace(float(start[0]), float(end[0]), r[0], dtype=torch.float).view(r[0], 1, 1)
        out[:, :, :, 1] = torch.linspace(float(start[1]), float(end[1]), r[1], dtype=torch.float).view(1, r[1], 1)
        out[:, :, :, 2] = torch.linspace(float(start[2]), float(end[2]), r[2], dtype=torch.float).view(1, 1, r[2])
        return out

    def tile_positions(self, begin, end) -> torch.Tensor:
        """(x, y, z, 3) fp32 normalised positions of the voxels [begin, end) of the volume lattice, formed per tile with
        the arithmetic of the reference's driver (visualization/OutputToVTK.py:23-37): the tile's corner voxels as
        fractions of the index range -> linspace between them -> [-1, 1] -> axis scales.  (A global lattice differs from
        this per-tile form by up to 6e-8; the HIP kernel's in-kernel lattice follows the same per-tile form.)"""
        span = self.max_idx - self.min_idx
        first = torch.tensor([b / (r - 1) for b, r in zip(begin, self.vol_res_touple)], dtype=torch.float)
        last = torch.tensor([(e - 1) / (r - 1) for e, r in zip(end, self.vol_res_touple)], dtype=torch.float)
        lo = (self.min_idx + first * span) / span
        hi = (self.min_idx + last * span) / span
        counts = [int(e) - int(b) for b, e in zip(begin, end)]
        return self.scales.view(1, 1, 1, 3) * (2.0 * self.generate_indices(lo, hi, counts) - 1.0)

    def lattice_from_flat(self, flat_idx: torch.Tensor) -> torch.Tensor:
        """Rows of the index table without materialising it."""
        _, Y, Z = self.vol_res_touple
        return torch.stack([flat_idx // (Y * Z), (flat_idx // Z) % Y, flat_idx % Z], -1).to(torch.float)

    def positions_for(self, raw: torch.Tensor):
        norm = normalize_volume(raw, self.min_idx.to(raw.device).unsqueeze(0), self.max_idx.to(raw.device).unsqueeze(0),
                                -1.0, 1.0)
        return raw, self.scales.to(raw.device).unsqueeze(0) * norm

    def positions_from_flat(self, flat_idx: torch.Tensor):
        """(raw, normalised) positions of flat voxel indices: one HIP kernel for device indices (bit-identical to the
        two methods above, which are ~10 elementwise torch launches), the torch route for host indices."""
        if flat_idx.is_cuda:
            if getattr(self, '_host_bounds', None) is None:        # host copies: no device read inside a train step
                self._host_bounds = (self.min_idx.cpu().tolist(), self.max_idx.cpu().tolist(), self.scales.cpu().tolist())
            mn, mx, sc = self._host_bounds
            return ops.lattice_positions(flat_idx, self.vol_res_touple, mn, mx, sc)
        return self.positions_for(self.lattice_from_flat(flat_idx))

    def sample_positions(self, n: int, device, seed: int = 0):
        """(raw, normalised) positions of `n` voxels drawn uniformly with replacement ON the device, draw and positions in
        one HIP kernel (ops.lattice_sample).  The draw counter is device state of this dataset: successive calls -- and
        successive replays of a captured call -- give successive batches of the stream selected by `seed`."""
        device = torch.device(device)
        if device.type == 'cuda' and device.index is None:
            device = torch.device('cuda', torch.cuda.current_device())
        st = getattr(self, '_sample_state', None)
        if st is None or st.device != device:
            st = self._sample_state = torch.zeros(2, dtype=torch.int64, device=device)
        if getattr(self, '_host_bounds', None) is None:
            self._host_bounds = (self.min_idx.cpu().tolist(), self.max_idx.cpu().tolist(), self.scales.cpu().tolist())
        mn, mx, sc = self._host_bounds
        return ops.lattice_sample(st, n, seed, self.vol_res_touple, mn, mx, sc)

    def __len__(self):
        return self.n_voxels

    def __getitem__(self, index):
        flat = torch.randint(0, self.n_voxels, (self.sample_size,))
        raw = self.volume_indices[flat] if self.volume_indices is not None else self.lattice_from_flat(flat)
        return self.positions_for(raw)


class DeviceLatticeSampler:
    """On-device counterpart of DataLoader(IndexDataset) (training/training.py:188-189, data/IndexDataset.py:90-96):
    draws ``n`` random voxel-lattice points per call directly on the GPU (no (n_voxels, 3) table, no worker processes,
    no H2D copies) and returns ``(raw_positions (n,3), normalized_positions (n,3))`` with the reference's arithmetic.
    Everything it launches is stream-ordered, so a train step using it can be captured in a HIP graph."""

    def __init__(self, volume_shape, device):
        self.ds = IndexDataset(tuple(int(v) for v in volume_shape), 1, build_index_table=False)
        self.device = torch.device(device)
        self.min_idx = self.ds.min_idx.to(self.device)
        self.max_idx = self.ds.max_idx.to(self.device)
        self.scales = self.ds.scales.to(self.device)
        self.n_voxels = self.ds.n_voxels

    def sample(self, n: int, generator=None):
        flat = torch.randint(0, self.n_voxels, (int(n),), device=self.device, generator=generator)
        return self.ds.positions_from_flat(flat)

    def sample_fused(self, n: int, seed: int = 0):
        """The same in one kernel: the indices are drawn by a counter-based generator inside the kernel that forms the
        positions (IndexDataset.sample_positions); two launches and torch's per-replay generator bookkeeping less."""
        return self.ds.sample_positions(n, self.device, seed)
