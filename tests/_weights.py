"""Shared helpers for tests: procedural weights keyed by the reference's state_dict names.

The name/shape tables come from this repo's own mirror modules (serenade_amd.models /
serenade_amd.vocoder), whose state_dict layout equals the reference's (checked against
tests/golden/state_dict_keys.json captured from the reference)."""
import functools

import torch

from serenade_amd.utils.synth import HIFIGAN_PARAMS, SERENADE_PARAMS, fill_state_dict


@functools.lru_cache(maxsize=None)
def serenade_weights(seed=0):
    from serenade_amd.models import serenade_state_shapes
    return fill_state_dict(serenade_state_shapes(**SERENADE_PARAMS), seed=seed)


@functools.lru_cache(maxsize=None)
def hifigan_weights(seed=0, small=False):
    from serenade_amd.vocoder import hifigan_state_shapes
    params = dict(HIFIGAN_PARAMS)
    if small:
        params.update(channels=64, upsample_scales=(4, 2), upsample_kernel_sizes=(8, 4),
                      resblock_kernel_sizes=(3, 5), resblock_dilations=[(1, 2), (2, 6, 3)])
    sd = fill_state_dict(hifigan_state_shapes(**params, weight_norm=True), seed=seed)
    return fold_weight_norm(sd), params


def fold_weight_norm(sd):
    """remove_weight_norm(): weight = g * v / ||v||  (hifigan.py:206-217)."""
    out = {}
    for k, v in sd.items():
        if k.endswith("weight_g"):
            vv = sd[k[:-1] + "v"]
            norm = vv.reshape(vv.shape[0], -1).norm(dim=1).reshape(v.shape)
            out[k[:-2]] = vv * (v / norm)
        elif k.endswith("weight_v"):
            continue
        else:
            out[k] = v
    return out


def sub(w, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in w.items() if k.startswith(prefix)}


def to64(w):
    return {k: (v.double() if v.is_floating_point() else v) for k, v in w.items()}
