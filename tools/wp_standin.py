"""A stand-in for the `warp` package (warp-lang 1.7.2 is not in this image, requirements.txt:36) that lets the reference's OWN Warp source -
cosmos_predict1/diffusion/inference/ray_triangle_intersection_warp.py, kernel body and host wrapper, unmodified - execute on the CPU:
`@wp.kernel` functions are plain Python functions run once per thread index by `wp.launch`, scalars are numpy float32 / int32 (every
arithmetic result is rounded to fp32, one operation at a time, no fused multiply-add), `wp.vec3` is three float32 components.

Used ONLY by tools/gen_golden_warp_kernel.py in the build container to pin oracle/warp_oracle.ray_triangle_depth and oracle/c/ray_tri.c to
what the reference's kernel source computes. What this cannot pin: the instruction selection of the real Warp / NVRTC build (e.g. fma
contraction inside dot / cross) - stated in the fixture's header and in DESIGN.md.
"""
from __future__ import annotations

import sys
import types

import numpy as np

f32 = np.float32
_tid = [0]


class vec3:
    __slots__ = ("x", "y", "z")

    def __init__(self, x, y, z):
        self.x, self.y, self.z = f32(x), f32(y), f32(z)

    def __sub__(self, o):
        return vec3(self.x - o.x, self.y - o.y, self.z - o.z)

    def __add__(self, o):
        return vec3(self.x + o.x, self.y + o.y, self.z + o.z)


def cross(a: vec3, b: vec3) -> vec3:  # warp/native/vec.h: (a.y b.z - a.z b.y, a.z b.x - a.x b.z, a.x b.y - a.y b.x)
    return vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x)


def dot(a: vec3, b: vec3):  # warp/native/vec.h: left-to-right sum of the component products
    return (a.x * b.x + a.y * b.y) + a.z * b.z


class _Array:
    """wp.array over a torch tensor's memory (wp.from_torch is zero-copy): reads return numpy scalars, writes go through to the tensor."""

    def __init__(self, t):
        self.t = t
        self.a = t.numpy()

    def __getitem__(self, idx):
        return self.a[idx]

    def __setitem__(self, idx, v):
        self.a[idx] = v


def install() -> types.ModuleType:
    m = types.ModuleType("warp")
    m.init = lambda: None
    m.kernel = lambda fn: fn
    m.array = lambda dtype=None, **k: None      # only used as parameter annotations
    m.array2d = lambda dtype=None, **k: None
    m.float32, m.int32 = f32, np.int32
    m.vec3, m.cross, m.dot = vec3, cross, dot
    m.abs = lambda v: abs(v)
    m.tid = lambda: _tid[0]
    m.from_torch = lambda t, dtype=None: _Array(t)
    m.synchronize = lambda: None

    def atomic_min(arr, idx, v):
        old = arr[idx]
        if v < old:
            arr[idx] = v
        return old

    m.atomic_min = atomic_min

    def launch(kernel, dim, inputs, device=None, **_k):
        args = [f32(a) if isinstance(a, float) else (np.int32(a) if isinstance(a, int) else a) for a in inputs]
        for i in range(int(dim)):
            _tid[0] = i
            kernel(*args)

    m.launch = launch
    sys.modules["warp"] = m
    return m
