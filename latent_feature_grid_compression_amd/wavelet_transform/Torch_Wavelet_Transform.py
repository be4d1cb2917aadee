"""Host-side mirror of the reference's wavelet filter module
(wavelet_transform/Torch_Wavelet_Transform.py: _WaveletFilterNd :21-104, WaveletFilter3d :110-113).

Same constructor, buffers (``filter_fwd``, ``filter_rev`` -> state_dict keys ``filter.filter_fwd`` /
``filter.filter_rev``), ``encode`` / ``decode`` signatures and return shapes.  ``decode`` (on the hot
path: it runs inside every forward) and its autograd adjoint are HIP kernels behind the C-ABI;
``encode`` (init only) is a HIP kernel for GPU tensors and a torch grouped conv for CPU tensors so
that a model can still be constructed on the host, as the reference's setup_model does, before it is
moved to the GPU.
"""
from __future__ import annotations

import math
from typing import Sequence, Union

import numpy as np
import torch
from torch import Tensor, nn
from torch.nn import functional as F

from .. import ops

# PyWavelets' db2 filter bank (dec_lo, dec_hi, rec_lo, rec_hi); pywt is used when importable, this table
# otherwise (the reference only ever uses 'db2': experiment-config-files/*.txt, model/model_utils.py:312).
_DB2 = (
    (-0.12940952255126037, 0.2241438680420134, 0.8365163037378079, 0.48296291314453416),
    (-0.48296291314453416, 0.8365163037378079, -0.2241438680420134, -0.12940952255126037),
    (0.48296291314453416, 0.8365163037378079, 0.2241438680420134, -0.12940952255126037),
    (-0.12940952255126037, -0.2241438680420134, 0.8365163037378079, -0.48296291314453416),
)


def _filter_bank(wavelet) -> Sequence[Sequence[float]]:
    if hasattr(wavelet, 'filter_bank'):
        return wavelet.filter_bank
    name = str(wavelet)
    if name == 'db2':
        return _DB2
    try:
        import pywt
    except ImportError as exc:
        raise ValueError("wavelet %r needs PyWavelets; only 'db2' is built in" % name) from exc
    return pywt.Wavelet(name).filter_bank


def dwt_max_level(data_len: int, filter_len: int) -> int:
    """pywt.dwt_max_level: floor(log2(data_len / (filter_len - 1))), clamped at 0."""
    if filter_len < 2 or data_len < filter_len - 1:
        return 0
    return max(0, int(math.floor(math.log2(data_len / (filter_len - 1.0)))))


class _WaveletFilterNd(nn.Module):
    def __init__(self, wavelet, dim: int, padding: str = 'constant'):
        super().__init__()
        if dim != 3:
            raise NotImplementedError('the HIP wavelet kernels are 3-D')
        if padding != 'constant':
            raise NotImplementedError("only zero ('constant') padding is implemented, as used by the reference")
        self.dim = dim
        self.padding = padding
        self._register_filters(_filter_bank(wavelet), dim)
        if self.filter_length != 4:
            raise NotImplementedError('the HIP wavelet kernels are written for 4-tap filters (db2)')

    @property
    def filter_length(self) -> int:
        return self.filter_fwd.shape[-1]

    def _register_filters(self, bank, dim: int):
        # 1-D taps cast to fp32 first, outer products in fp32; sub-band s = 4a + 2b + c (first/second/third axis)
        fwd_low, fwd_high, rev_low, rev_high = (torch.tensor([float(v) for v in x], dtype=torch.float32) for x in bank)

        def nd(f1d):
            out = list(f1d)
            for _ in range(1, dim):
                out = [a.reshape(-1, *([1] * o.dim())) * o.unsqueeze(0) for a in f1d for o in out]
            return torch.stack(out, dim=0).unsqueeze(1)

        self.register_buffer('filter_fwd', nd([fwd_low.flip(-1), fwd_high.flip(-1)]))
        self.register_buffer('filter_rev', nd([rev_low, rev_high]))

    def _get_padding_size(self, shape: np.ndarray):
        is_odd = np.asarray(shape) % 2 == 1
        out = np.full(2 * self.dim, (2 * self.filter_length - 3) // 2, dtype=int)
        out[1::2] += is_odd.astype(int)
        return tuple(int(v) for v in out)

    def encode(self, data: Tensor):
        """(batch, C, D, H, W) -> coeffs (batch, C, 8, d, h, w), pre-pad spatial shape (ndarray)."""
        assert data.dim() == 2 + self.dim, 'encode expects (batch, channel, *spatial)'
        shape = np.asarray(data.shape[-self.dim:])
        if data.is_cuda:
            coeffs = torch.stack([ops.dwt_level(d, self.filter_fwd.to(data.device)) for d in data], 0)
        else:   # host-side construction path (init only, never inside forward)
            padded = F.pad(data, self._get_padding_size(shape), mode=self.padding)
            c = padded.shape[1]
            result = F.conv3d(padded, self.filter_fwd.to(padded).repeat(c, 1, 1, 1, 1), stride=2, groups=c)
            coeffs = result.reshape(padded.shape[0], c, 2 ** self.dim, *result.shape[2:])
        return coeffs, shape

    def decode(self, data: Tensor, shape) -> Tensor:
        """(batch, C, 8, d, h, w) -> (batch, C, *shape).  GPU only (HIP kernel); differentiable."""
        assert data.dim() == 3 + self.dim, 'decode expects (batch, channel, 8, *spatial)'
        outs = []
        for d in data:
            lll, hf = d[:, 0], d[:, 1:]
            if d.requires_grad:
                # one-level instance of the decode autograd node: coeffs = (coarse, detail)
                outs.append(ops.DecodeVolumeFn.apply(self.filter_rev, [tuple(int(v) for v in shape)], False,
                                                     lll.contiguous(), hf.contiguous()))
            else:
                outs.append(ops.idwt_level(lll, hf, self.filter_rev, shape))
        return torch.stack(outs, 0)

    def forward(self, data: Tensor):
        return self.encode(data)[0]


class WaveletFilter3d(_WaveletFilterNd):
    def __init__(self, wavelet: Union[str, object], padding: str = 'constant'):
        super().__init__(wavelet, 3, padding=padding)
