"""ORACLE (test infrastructure, NOT product code) -- numpy restatement of the reference's binary checkpoint format
(model/model_utils.py: store_model_parameters :120-223, restore_model :226-332, bit helpers :73-117; SURVEY.md section 8
row f4).  Only ``tests/`` may import this module.

File layout (all scalars in the writer's native = little-endian order, no padding: every field is its own struct.pack):
  parameter file
    9 x uint8    n_layers, layer_width, input_dim (= 3 + 12 + C), input_channel (d_in), output_dim, bit_precision,
                 grid_size (shape_array[-1][0]), n_grids, feature_size (C)                                  (:145-155)
    n_grids x uint32  non-zero count per coefficient tensor;   n_grids x uint32  zero count per tensor      (:156-159)
    first layer: weight (layer_width x input_dim) fp32, bias fp32                                           (:164-168)
    hidden layers 1..n_layers-1: [2^bits fp32 centres][labels, `bits` bits each, MSB first, last byte's tail
                 filled with whatever int(bin_string, 2) of the short string gives = right-aligned][if bits % 8:
                 uint32 of the last label] then bias fp32                                                   (:170-191)
    final layer: weight, bias fp32                                                                          (:193-197)
    per coefficient tensor: quantised NON-ZERO values only, same [centres][labels] block                    (:202-215)
  mask file (name + "_mask.bnr"): one bit per coefficient over all tensors concatenated, 1 = non-zero, MSB first, last
                 byte padded with zero bits on the right                                                    (:94-107, :217)

Pinning: ``tests/golden/codec_small.npz`` holds the two files the reference wrote for a small pruned model and the state
its restore_model rebuilt; tests/test_oracle_codec_golden.py checks parse() against that state bit for bit and that
serialize(parse(file)) reproduces the reference's bytes exactly.
"""
from __future__ import annotations

import struct
from typing import Dict, List

import numpy as np


def unpack_labels(buf: bytes, n: int, bits: int) -> np.ndarray:
    """read_in_data_quantized's bit-string slicing (:262-263): label i = bits [bits*i, bits*(i+1)) of the byte stream,
    MSB first.  (For a trailing partial byte the writer's int(bin_string, 2) right-aligns the leftover bits, so the last
    label of a non-multiple-of-8 stream is garbage here and is overridden by the explicit uint32 that follows, :265-267.)"""
    bitarr = np.unpackbits(np.frombuffer(buf, dtype=np.uint8))
    need = n * bits
    if bitarr.size < need:
        bitarr = np.concatenate([bitarr, np.zeros(need - bitarr.size, np.uint8)])
    w = (1 << np.arange(bits - 1, -1, -1)).astype(np.int64)
    return (bitarr[:need].reshape(n, bits).astype(np.int64) * w).sum(1)


def pack_labels(labels: np.ndarray, bits: int) -> bytes:
    """ints_to_bits_to_bytes (:73-86): concatenate `bits`-wide binary strings, cut into bytes; a short last chunk is
    converted with int(chunk, 2), i.e. RIGHT-aligned in its byte."""
    labels = np.asarray(labels, dtype=np.int64)
    w = (labels[:, None] >> np.arange(bits - 1, -1, -1)[None, :]) & 1
    bitarr = w.reshape(-1).astype(np.uint8)
    full = bitarr.size // 8
    out = bytearray(np.packbits(bitarr[:full * 8]).tobytes())
    rest = bitarr[full * 8:]
    if rest.size:
        out.append(int(''.join(str(int(b)) for b in rest), 2))
    return bytes(out)


def parse(param_file: bytes, mask_file: bytes) -> Dict[str, object]:
    """restore_model's reading (:226-306) without the model construction: header, layers, coefficient tensors (flat, zeros
    re-inserted by the mask), plus the raw codebooks / labels of every quantised block."""
    pos = 0

    def take(fmt):
        nonlocal pos
        size = struct.calcsize(fmt)
        vals = struct.unpack(fmt, param_file[pos:pos + size])
        pos += size
        return vals

    n_layers, layer_width, input_dim, input_channel, output_dim, bits, grid_size, n_grids, feature_size = take('9B')
    n_clusters = int(2 ** bits)
    grid_sizes = [take('<I')[0] for _ in range(n_grids)]
    zeros = [take('<I')[0] for _ in range(n_grids)]

    def floats(n):
        return np.asarray(take('<%df' % n), dtype=np.float32)

    blocks = []

    def quantised(n):
        nonlocal pos
        centres = floats(n_clusters)
        nbytes = (n * bits) // 8 + (1 if (n * bits) % 8 else 0)
        labels = unpack_labels(param_file[pos:pos + nbytes], n, bits)
        pos += nbytes
        if bits % 8 != 0:
            labels[-1] = take('<I')[0]
        blocks.append({'centres': centres, 'labels': labels})
        return centres[labels]

    weights = [floats(input_dim * layer_width)]
    biases = [floats(layer_width)]
    for _ in range(n_layers - 1):
        weights.append(quantised(layer_width * layer_width))
        biases.append(floats(layer_width))
    weights.append(floats(output_dim * layer_width))
    biases.append(floats(output_dim))

    total = sum(grid_sizes) + sum(zeros)
    mask = np.unpackbits(np.frombuffer(mask_file, dtype=np.uint8))[:total].astype(bool)
    grids, at = [], 0
    for nz, z in zip(grid_sizes, zeros):
        vals = quantised(nz)
        full = np.zeros(nz + z, np.float32)
        full[mask[at:at + nz + z]] = vals
        at += nz + z
        grids.append(full)
    assert pos == len(param_file), 'trailing bytes in the parameter file'
    return {'header': dict(n_layers=n_layers, layer_width=layer_width, input_dim=input_dim, input_channel=input_channel,
                           output_dim=output_dim, bit_precision=bits, grid_size=grid_size, n_grids=n_grids,
                           feature_size=feature_size, grid_sizes=grid_sizes, zeros=zeros),
            'weights': weights, 'biases': biases, 'grids': grids, 'mask': mask, 'blocks': blocks}


def serialize(header: dict, weights: List[np.ndarray], biases: List[np.ndarray], blocks: List[dict], mask: np.ndarray):
    """store_model_parameters' writing (:120-223) given the codebooks (``blocks``: hidden layers first, then coefficient
    tensors).  Returns (parameter file bytes, mask file bytes)."""
    h = header
    bits = h['bit_precision']
    out = bytearray(struct.pack('9B', h['n_layers'], h['layer_width'], h['input_dim'], h['input_channel'], h['output_dim'],
                                bits, h['grid_size'], h['n_grids'], h['feature_size']))
    for v in h['grid_sizes']:
        out += struct.pack('<I', v)
    for v in h['zeros']:
        out += struct.pack('<I', v)

    def put_floats(a):
        out.extend(np.asarray(a, dtype='<f4').tobytes())

    def put_block(b):
        put_floats(b['centres'])
        out.extend(pack_labels(b['labels'], bits))
        if bits % 8 != 0:
            out.extend(struct.pack('<I', int(b['labels'][-1])))

    it = iter(blocks)
    put_floats(weights[0]); put_floats(biases[0])
    for i in range(1, h['n_layers']):
        put_block(next(it))
        put_floats(biases[i])
    put_floats(weights[-1]); put_floats(biases[-1])
    for _ in range(h['n_grids']):
        put_block(next(it))
    m = np.asarray(mask, dtype=np.uint8)
    return bytes(out), np.packbits(m).tobytes()              # packbits pads the last byte with zero bits on the right
