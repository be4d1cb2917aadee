"""ctypes binding of liblfgc.so (include/lfgc.h).  No torch types cross this boundary: only raw device
pointers, extents and the HIP stream handle.  Loading fails loudly -- there is no CPU fallback."""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint64, c_void_p

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('LFGC_LIB_PATH') or os.path.join(PKG_DIR, 'liblfgc.so')   # override: diagnostics builds only
LFGC_MAX_LAYERS = 8
PRECISION = {'fp32': 0, 'f16x2': 1, 'f16': 2}


class LfgcError(RuntimeError):
    pass


class MlpDesc(Structure):
    _fields_ = [('grid_channels', c_int32), ('hidden', c_int32), ('num_layers', c_int32),
                ('n_freqs', c_int32), ('d_in', c_int32), ('d_out', c_int32)]


class Positions(Structure):
    _fields_ = [('pos', c_void_p), ('n', c_int64), ('res', c_int32 * 3),
                ('x_begin', c_int32), ('x_end', c_int32), ('tile', c_int32)]


class PenaltyTerm(Structure):
    _fields_ = [('a', c_void_p), ('b', c_void_p), ('n', c_int64), ('kind', c_int32)]


PENALTY_L1, PENALTY_L2, PENALTY_DKL = 0, 1, 2
PENALTY_MAX_TERMS = 16
PENALTY_BLOCKS = 1024             # LFGC_PENALTY_SUMS_DOUBLES(n) = n * (1 + PENALTY_BLOCKS)
_PP = POINTER(c_void_p)
_TAPS = POINTER(c_float)          # host float[8] (1-D filter bank) or None

# name -> (restype, argtypes); mirrors include/lfgc.h one to one
SIGNATURES = {
    'lfgc_version': (c_int, []),
    'lfgc_error_string': (c_char_p, [c_int]),
    'lfgc_idwt_level_f32': (c_int, [c_void_p, c_void_p, c_void_p, _TAPS, c_void_p] + [c_int] * 7 + [c_void_p]),
    'lfgc_idwt_level_bwd_f32': (c_int, [c_void_p, c_void_p, _TAPS, c_void_p, c_void_p] + [c_int] * 7 + [c_void_p]),
    'lfgc_idwt_level_cl_f32': (c_int, [c_void_p, c_void_p, _TAPS, c_void_p] + [c_int] * 8 + [c_void_p]),
    'lfgc_idwt_level_cl_bwd_f32': (c_int, [c_void_p, _TAPS, c_void_p, c_void_p] + [c_int] * 8 + [c_void_p]),
    'lfgc_grid_layout_f32': (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_void_p]),
    'lfgc_dwt_level_f32': (c_int, [c_void_p, c_void_p, _TAPS, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'lfgc_idwt_level_drop_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_float, c_void_p, _TAPS, c_void_p] +
                                 [c_int] * 7 + [c_void_p]),
    'lfgc_idwt_level_drop_bwd_f32': (c_int, [c_void_p, c_void_p, _TAPS] + [c_void_p] * 8 + [_PP] + [c_int] * 7 + [c_void_p]),
    'lfgc_drop_apply_f32': (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_int, c_int64, c_void_p]),
    'lfgc_drop_apply_bwd_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p]),
    'lfgc_sign_variance_update_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_int64, c_void_p]),
    'lfgc_sign_variance_update_multi_f32': (c_int, [_PP, _PP, _PP, POINTER(c_int64), c_int, c_float, c_void_p]),
    'lfgc_penalty_sums_f32': (c_int, [POINTER(PenaltyTerm), c_int, c_void_p, c_void_p]),
    'lfgc_penalty_grads_f32': (c_int, [POINTER(PenaltyTerm), c_int, c_void_p, _PP, _PP, c_void_p]),
    'lfgc_codec_mask_f32': (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    'lfgc_codec_select_workspace_bytes': (c_int64, [c_int64]),
    'lfgc_codec_compact_f32': (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    'lfgc_codec_expand_f32': (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    'lfgc_codec_ward_init_host': (c_int, [POINTER(c_float), c_int64, c_int, POINTER(c_float)]),
    'lfgc_codec_kmeans_workspace_bytes': (c_int64, [c_int]),
    'lfgc_codec_kmeans1d_f32': (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_void_p]),
    'lfgc_codec_dequant_f32': (c_int, [c_void_p, c_int64, c_int, c_int64, c_void_p, c_void_p, c_void_p]),
    'lfgc_mlp_supported': (c_int, [POINTER(MlpDesc)]),
    'lfgc_grid_channel_stride': (c_int, [c_int]),
    'lfgc_packed_bytes': (c_int64, [POINTER(MlpDesc)]),
    'lfgc_stash_bytes': (c_int64, [POINTER(MlpDesc), c_int64]),
    'lfgc_pack_mlp_f32': (c_int, [POINTER(MlpDesc), _PP, _PP, c_void_p, c_void_p]),
    'lfgc_forward_f32': (c_int, [POINTER(MlpDesc), POINTER(Positions), c_void_p, c_int, c_int, c_int,
                                 c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'lfgc_backward_workspace_bytes': (c_int64, [POINTER(MlpDesc), c_int64]),
    'lfgc_backward_f32': (c_int, [POINTER(MlpDesc), POINTER(Positions), c_void_p, c_int, c_int, c_int,
                                  c_void_p, c_int, c_void_p, c_void_p, c_void_p, _PP, _PP, c_void_p,
                                  c_void_p, c_int64, c_void_p]),
    'lfgc_forward_bf16': (c_int, [POINTER(MlpDesc), POINTER(Positions), c_void_p, c_int, c_int, c_int,
                                  c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'lfgc_backward_bf16': (c_int, [POINTER(MlpDesc), POINTER(Positions), c_void_p, c_int, c_int, c_int,
                                   c_void_p, c_void_p, c_void_p, c_void_p, _PP, _PP, c_void_p,
                                   c_void_p, c_int64, c_void_p]),
    'lfgc_gt_interp_f32': (c_int, [c_void_p, c_void_p, POINTER(c_float), POINTER(c_float), POINTER(c_float),
                                   c_int64, c_int, c_int, c_int, c_void_p, c_void_p]),
    'lfgc_gt_mse_workspace_bytes': (c_int64, [c_int64]),
    'lfgc_gt_mse_f32': (c_int, [c_void_p, c_void_p, POINTER(c_float), POINTER(c_float), POINTER(c_float), c_int64, c_int, c_int,
                                c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    'lfgc_lattice_positions_f32': (c_int, [c_void_p, c_int64, POINTER(c_int32), POINTER(c_float), POINTER(c_float),
                                           POINTER(c_float), c_void_p, c_void_p, c_void_p]),
    'lfgc_lattice_sample_f32': (c_int, [c_uint64, c_void_p, c_int64, POINTER(c_int32), POINTER(c_float), POINTER(c_float),
                                        POINTER(c_float), c_void_p, c_void_p, c_void_p, c_void_p]),
    'lfgc_deviation_partial_f32': (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    'lfgc_debug_trig_f32': (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    'lfgc_debug_hwsin_f32': (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
}

_lib = None


def load() -> ctypes.CDLL:
    """Load liblfgc.so (built by ``python -m latent_feature_grid_compression_amd.build``)."""
    global _lib
    if _lib is not None:
        return _lib
    build_error = None
    if not os.path.exists(LIB_PATH) and 'LFGC_LIB_PATH' not in os.environ:
        try:                                   # fresh checkout: build in-tree (hipcc, gfx950); still no CPU fallback
            from .build import build
            build(verbose=False)
        except Exception as e:                 # noqa: BLE001 -- reported below, chained into the LfgcError
            build_error = e
    if not os.path.exists(LIB_PATH):
        raise LfgcError('liblfgc.so not found at %s: build it with `python -m latent_feature_grid_compression_amd.build` '
                        '(hipcc, gfx950). There is no CPU fallback for the HIP path.%s'
                        % (LIB_PATH, '' if build_error is None else ' The in-tree build failed: %s' % build_error)
                        ) from build_error
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header / library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code: int, what: str) -> None:
    if code != 0:
        msg = load().lfgc_error_string(int(code))
        raise LfgcError('%s failed: %s (code %d)' % (what, msg.decode() if msg else '?', code))


def ptr_array(ptrs):
    arr = (c_void_p * len(ptrs))(*[c_void_p(p) for p in ptrs])
    return ctypes.cast(arr, _PP), arr
