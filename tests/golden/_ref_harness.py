"""Harness that makes the reference (/root/reference) importable in THIS container.

Used ONLY by tests/golden/make_golden.py to capture golden vectors; never shipped,
never imported by the product or by tests that run on the GPU box.

What it does:
  * pre-seeds ``sys.modules`` with empty ``h5py`` / ``soundfile`` (I/O only) and a
    ``typeguard`` whose ``typechecked`` is the identity decorator;
  * installs a stand-in for the six ``diffusers`` symbols the reference imports
    (``Attention``, ``GEGLU``, ``GELU``, ``ApproximateGELU``, ``AdaLayerNorm``,
    ``AdaLayerNormZero``, ``LoRACompatibleLinear``, ``maybe_allow_in_graph``,
    ``get_activation``).  diffusers is NOT in the container and is unpinned by the
    reference (setup.cfg:21): the stand-in follows the published diffusers
    ``AttnProcessor2_0`` / ``GEGLU`` semantics, so rows a4.6 are "parity unpinned"
    with respect to the third-party library itself.
"""
import math
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REFERENCE_ROOT = "/root/reference"


class _Attention(nn.Module):
    """diffusers.models.attention_processor.Attention (self-attention subset).

    to_q/to_k/to_v bias-free (bias=False at the call site transformer.py:212-220),
    to_out = [Linear(inner, query_dim, bias=True), Dropout]; 2-D bool mask (B, L)
    -> (B, heads, 1, L) passed as SDPA attn_mask (True = attend); scale 1/sqrt(dim_head).
    """

    def __init__(self, query_dim, heads=8, dim_head=64, dropout=0.0, bias=False,
                 cross_attention_dim=None, upcast_attention=False, **kw):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(query_dim, inner, bias=bias)
        self.to_v = nn.Linear(query_dim, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(dropout)])

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        b, l, _ = hidden_states.shape
        q = self.to_q(hidden_states)
        k = self.to_k(hidden_states)
        v = self.to_v(hidden_states)
        h = self.heads
        d = q.shape[-1] // h
        q = q.view(b, l, h, d).transpose(1, 2)
        k = k.view(b, l, h, d).transpose(1, 2)
        v = v.view(b, l, h, d).transpose(1, 2)
        m = None
        if attention_mask is not None:
            m = attention_mask.to(torch.bool).view(b, 1, 1, l).expand(b, h, 1, l)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=m, dropout_p=0.0, is_causal=False)
        o = o.transpose(1, 2).reshape(b, l, h * d)
        o = self.to_out[0](o)
        o = self.to_out[1](o)
        return o


class _GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        h, g = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(g)


class _GELU(nn.Module):
    def __init__(self, dim_in, dim_out, approximate="none"):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out)
        self.approximate = approximate

    def forward(self, x):
        return F.gelu(self.proj(x), approximate=self.approximate)


class _Unused(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
        raise RuntimeError("stand-in: this diffusers class is never constructed on the hot path")


def _get_activation(name):
    name = name.lower()
    if name in ("silu", "swish"):
        return nn.SiLU()
    if name == "mish":
        return nn.Mish()
    if name == "gelu":
        return nn.GELU()
    if name == "relu":
        return nn.ReLU()
    raise ValueError(name)


def install():
    if "serenade" in sys.modules:
        return
    sys.dont_write_bytecode = True
    for name in ("h5py", "soundfile"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    tg = types.ModuleType("typeguard")
    tg.typechecked = lambda f=None, **k: (f if f is not None else (lambda g: g))
    sys.modules["typeguard"] = tg

    d = types.ModuleType("diffusers")
    dm = types.ModuleType("diffusers.models")
    da = types.ModuleType("diffusers.models.attention")
    dact = types.ModuleType("diffusers.models.activations")
    dap = types.ModuleType("diffusers.models.attention_processor")
    dl = types.ModuleType("diffusers.models.lora")
    du = types.ModuleType("diffusers.utils")
    dut = types.ModuleType("diffusers.utils.torch_utils")
    da.GEGLU = _GEGLU
    da.GELU = _GELU
    da.ApproximateGELU = _Unused
    da.AdaLayerNorm = _Unused
    da.AdaLayerNormZero = _Unused
    dact.get_activation = _get_activation
    dap.Attention = _Attention
    dl.LoRACompatibleLinear = nn.Linear
    dut.maybe_allow_in_graph = lambda cls: cls
    for n, m in (("diffusers", d), ("diffusers.models", dm), ("diffusers.models.attention", da),
                 ("diffusers.models.activations", dact), ("diffusers.models.attention_processor", dap),
                 ("diffusers.models.lora", dl), ("diffusers.utils", du),
                 ("diffusers.utils.torch_utils", dut)):
        sys.modules[n] = m
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
