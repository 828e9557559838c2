"""serenade_amd — MI355X-native inference hot path of imulki/serenade.

Contraction arithmetic policy (applies to plans built afterwards):

    serenade_amd.set_precision("bf16x3")   # default: every fp32 operand split into (hi, lo) bf16, 3 MFMA per
                                           # product, fp32 accumulate (~2^-17 relative per product)
    serenade_amd.set_precision("fp32")     # exact fp32 MFMA (bit-for-bit an fp32 fma chain)
    serenade_amd.set_precision("bf16x6")   # fp32-faithful emulation: operands split EXACTLY into (hi, mid, lo) bf16,
                                           # 6 MFMA per product, fp32 accumulate; dropped terms <= 2^-26 per product
                                           # (below fp32's rounding unit) at 6/16 of the fp32-MFMA cost

or the environment variable SERENADE_AMD_PRECISION=fp32|bf16x3|bf16x6.

    serenade_amd.set_attention_precision("bf16x3")   # Q K^T and P V only (None: follow set_precision)

lets the two attention contractions -- 24 576 L^2 of the estimator's 81.3e6 L + 24 576 L^2 FLOP per call: 28 % at
L = 1280, 57 % at L = 4352 -- run on the bf16 matrix cores while every conv / linear stays exact fp32 (BASELINE
configs[4]: "MFMA attention" for long-form input).  SERENADE_AMD_ATTENTION_PRECISION sets it from the environment.
"""
import os

from . import _lib, ops

_NAMES = {"fp32": _lib.PREC_FP32, "bf16x3": _lib.PREC_BF16X3, "bf16x6": _lib.PREC_BF16X6}


def set_precision(name):
    if name not in _NAMES:
        raise ValueError(f"precision must be one of {sorted(_NAMES)}, got {name!r}")
    ops.DEFAULT_PRECISION = _NAMES[name]


def set_attention_precision(name):
    if name is not None and name not in _NAMES:
        raise ValueError(f"attention precision must be None or one of {sorted(_NAMES)}, got {name!r}")
    ops.ATTENTION_PRECISION = None if name is None else _NAMES[name]


def get_attention_precision():
    return None if ops.ATTENTION_PRECISION is None else {v: k for k, v in _NAMES.items()}[ops.ATTENTION_PRECISION]


def get_precision():
    return {v: k for k, v in _NAMES.items()}[ops.DEFAULT_PRECISION]


# default: the reference's arithmetic (exact fp32); the split-bf16 modes are opt-in
set_precision(os.environ.get("SERENADE_AMD_PRECISION", "fp32"))
set_attention_precision(os.environ.get("SERENADE_AMD_ATTENTION_PRECISION") or None)
