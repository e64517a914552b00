"""Deterministic, integer-only synthetic grids.

Multi-octave triangle-wave field plus hashed low-bit noise; exact int->float conversion and a power-of-two
scale, so the generated bits are identical on every machine and on CPU (numpy) and GPU (torch).  Used by
the tests (small shapes, numpy) and by bench.py (full BASELINE shapes, generated directly in HBM).

    lowbias32(x): x ^= x>>16; x *= 0x7feb352d; x ^= x>>15; x *= 0x846ca68b; x ^= x>>16      (uint32)
    tri(p):       p &= 0xffff; p < 0x8000 ? p : 0x10000 - p
    q   = 64 tri(37x+11y+5z) + 16 tri(151x+257y+93z+12345) + 4 tri(911x+613y+1201z+777) + tri(4099x+3001y+2503z+31337)
    nse = lowbias32(seed ^ (uint32(linear_index) * 0x9e3779b9)) & noise_mask
    value = T(q + nse - 1400000) * 2^-12

`smooth=True` drops the two high-frequency octaves (highly compressible bookend).
"""
from __future__ import annotations

import numpy as np

_OCT = (
    (64, (37, 11, 5), 0),
    (16, (151, 257, 93), 12345),
    (4, (911, 613, 1201), 777),
    (1, (4099, 3001, 2503), 31337),
)
_M32 = 0xFFFFFFFF


def _coords_axes(shape):
    """Return per-axis multipliers mapping: axis d of `shape` -> which of (x, y, z) it is."""
    dims = len(shape)
    # 1D: x = index; 2D: (y, x); 3D: (z, y, x)
    names = {1: ("x",), 2: ("y", "x"), 3: ("z", "y", "x")}[dims]
    return names


def synth_numpy(shape, dtype=np.float32, seed: int = 1, noise_mask: int = 0xFF, smooth: bool = False) -> np.ndarray:
    shape = tuple(int(s) for s in shape)
    dims = len(shape)
    names = _coords_axes(shape)
    n = int(np.prod(shape, dtype=np.int64))
    q = np.zeros(shape, dtype=np.int64)
    for weight, (ax, ay, az), k in (_OCT[:2] if smooth else _OCT):
        mult = {"x": ax, "y": ay, "z": az}
        arg = np.full((1,) * dims, k, dtype=np.uint32)
        for d in range(dims):
            c = (np.arange(shape[d], dtype=np.uint64) * np.uint64(mult[names[d]])) & np.uint64(_M32)
            view = [1] * dims
            view[d] = shape[d]
            arg = arg + c.astype(np.uint32).reshape(view)  # uint32 wraparound
        p = (arg & np.uint32(0xFFFF)).astype(np.int64)
        q += weight * np.where(p < 0x8000, p, 0x10000 - p)
    lin = np.arange(n, dtype=np.uint64).reshape(shape)
    h = (lin * np.uint64(0x9E3779B9)) & np.uint64(_M32)
    h ^= np.uint64(seed & _M32)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x7FEB352D)) & np.uint64(_M32)
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0x846CA68B)) & np.uint64(_M32)
    h ^= h >> np.uint64(16)
    nse = (h & np.uint64(noise_mask & _M32)).astype(np.int64)
    v = q + nse - 1400000
    dt = np.dtype(dtype)
    return v.astype(dt) * dt.type(2.0 ** -12)


def synth_torch_range(shape, dtype, start: int, count: int, out_flat, seed: int = 1, noise_mask: int = 0xFF,
                      smooth: bool = False, slab: int = 1 << 24):
    """Fill `out_flat[0:count]` with elements [start, start+count) (global row-major linear index) of the field of
    global shape `shape`, with torch int64 arithmetic on out_flat's device, in slabs so temporaries stay small."""
    import torch

    shape = tuple(int(s) for s in shape)
    dims = len(shape)
    names = _coords_axes(shape)
    device = out_flat.device
    strides = []
    acc = 1
    for s in reversed(shape):
        strides.append(acc)
        acc *= s
    strides = list(reversed(strides))
    M = _M32
    for s0 in range(start, start + count, slab):
        s1 = min(start + count, s0 + slab)
        lin = torch.arange(s0, s1, dtype=torch.int64, device=device)
        coords = {}
        rem = lin
        for d in range(dims):
            coords[names[d]] = rem // strides[d]
            rem = rem % strides[d]
        zero = torch.zeros_like(lin)
        x = coords.get("x", zero)
        y = coords.get("y", zero)
        z = coords.get("z", zero)
        q = torch.zeros_like(lin)
        for weight, (ax, ay, az), k in (_OCT[:2] if smooth else _OCT):
            p = (ax * x + ay * y + az * z + k) & 0xFFFF
            q += weight * torch.where(p < 0x8000, p, 0x10000 - p)
        h = (lin * 0x9E3779B9) & M
        h = h ^ (seed & M)
        h = h ^ (h >> 16)
        h = (h * 0x7FEB352D) & M
        h = h ^ (h >> 15)
        h = (h * 0x846CA68B) & M
        h = h ^ (h >> 16)
        v = q + (h & (noise_mask & M)) - 1400000
        out_flat[s0 - start: s1 - start] = v.to(dtype) * (2.0 ** -12)
    return out_flat


def synth_torch(shape, dtype, seed: int = 1, noise_mask: int = 0xFF, smooth: bool = False, device="cuda",
                slab: int = 1 << 24):
    """Same field as synth_numpy, generated on `device`."""
    import torch

    shape = tuple(int(s) for s in shape)
    n = 1
    for s in shape:
        n *= s
    out = torch.empty(n, dtype=dtype, device=device)
    synth_torch_range(shape, dtype, 0, n, out, seed, noise_mask, smooth, slab)
    return out.reshape(shape)
