"""Import shims that let the reference's OWN Python (read-only under /root/reference) run on a GPU-less host.

Used only by tools/gen_golden.py (fixture generation, in the build container) - never by the product or by tests:
/root/reference does not exist on the GPU box, the committed fixtures under tests/golden/ do.

Each shim replaces a third-party package that is absent from this image (no network): the three TransformerEngine
symbols attention.py uses, torchvision's nearest resize, megatron parallel_state, loguru, pynvml, warp, and the
lazy_config package (pulls omegaconf/hydra). Their arithmetic follows the pinned versions' published semantics
(SURVEY.md 8c); they are the "parity unpinned" part of the oracle.
"""
from __future__ import annotations

import enum
import sys
import types

import torch

REFERENCE_ROOT = "/root/reference"


def _mod(name: str, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _AnyCall:
    def __getattr__(self, _k):
        return lambda *a, **kw: None


class _FakeLogger:
    def __init__(self, *a, **k):
        self._options = (None,) * 9

    def __getattr__(self, _k):
        return lambda *a, **kw: self


class TERMSNorm(torch.nn.Module):
    def __init__(self, channels, eps=1e-6):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.ones(channels))
        self.eps = eps

    def forward(self, x):
        xf = x.float()
        return (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + self.eps) * self.weight.float()).to(x.dtype)


class TEDotProductAttention(torch.nn.Module):
    def __init__(self, heads, dim, **kw):
        super().__init__()

    def set_context_parallel_group(self, *a, **k):
        pass

    def forward(self, q, k, v, **kw):  # sbhd -> [s, b, h*d]
        q, k, v = (t.permute(1, 2, 0, 3) for t in (q, k, v))
        o = torch.nn.functional.scaled_dot_product_attention(q, k, v)
        return o.permute(2, 0, 1, 3).flatten(2)


def te_apply_rotary_pos_emb(t, freqs, tensor_format="sbhd", fused=False):
    tf = t.float()
    x1, x2 = tf.chunk(2, -1)
    return (tf * freqs.cos() + torch.cat([-x2, x1], -1) * freqs.sin()).to(t.dtype)


def install():
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    _mod("loguru", logger=_AnyCall())
    _mod("loguru._logger", Core=type("Core", (), {}), Logger=_FakeLogger)
    te = _mod("transformer_engine")
    te.pytorch = _mod("transformer_engine.pytorch", RMSNorm=TERMSNorm)
    _mod("transformer_engine.pytorch.attention", DotProductAttention=TEDotProductAttention,
         apply_rotary_pos_emb=te_apply_rotary_pos_emb)

    class _IM(enum.Enum):
        NEAREST = 0
        BICUBIC = 1

    def _resize(img, size, interpolation=None, antialias=None):
        return torch.nn.functional.interpolate(img.float(), size=size, mode="nearest").to(img.dtype)

    tvf = _mod("torchvision.transforms.functional", InterpolationMode=_IM, resize=_resize)
    _mod("torchvision", transforms=_mod("torchvision.transforms", functional=tvf, InterpolationMode=_IM))
    ps = _mod("megatron.core.parallel_state", is_initialized=lambda: False)
    _mod("megatron", core=_mod("megatron.core", parallel_state=ps))
    _mod("pynvml")
    _mod("cosmos_predict1.utils.lazy_config", instantiate=lambda cfg, *a, **k: cfg, LazyCall=None, LazyDict=dict)
    _mod("warp", init=lambda: None, kernel=lambda f: f, array=lambda **k: None, array2d=lambda **k: None, float32=float,
         int32=int, vec3=None)
    torch.Tensor.cuda = lambda self, *a, **k: self  # position_embedding.py:113,118 call .cuda() in __init__
