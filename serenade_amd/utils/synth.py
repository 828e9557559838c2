"""Procedural, name-keyed synthetic weights and inputs.

No pretrained Serenade / HiFi-GAN weights exist offline and 84 M + 13.7 M parameters
are too large to commit, so every ``state_dict`` entry is filled from a numpy
generator seeded by the parameter *name*.  The same filler is applied to the
reference modules (when capturing golden vectors), to the CPU oracle and to the
HIP-backed modules, on any machine, so results are comparable everywhere and do
not depend on the torch version.
"""
import zlib

import numpy as np
import torch


def _rng(name, seed):
    return np.random.default_rng([zlib.crc32(name.encode("utf-8")), int(seed)])


def _fill_one(name, shape, dtype, seed):
    shape = tuple(int(s) for s in shape)
    rng = _rng(name, seed)
    leaf = name.rsplit(".", 1)[-1]
    if dtype in (torch.int64, torch.int32):
        return torch.zeros(shape, dtype=dtype)
    if leaf == "running_var":
        a = rng.uniform(0.5, 1.5, size=shape)
    elif leaf == "running_mean":
        a = 0.1 * rng.standard_normal(shape)
    elif leaf == "gst_embs":
        a = 0.5 * rng.standard_normal(shape)
    elif leaf == "weight_g":
        # filled together with weight_v in fill_state_dict (needs ||v||)
        a = np.ones(shape)
    elif leaf.startswith("bias"):
        if "W_scale" in name:
            a = 1.0 + 0.1 * rng.standard_normal(shape)
        else:
            a = 0.1 * rng.standard_normal(shape)
    elif len(shape) == 1:
        # GroupNorm / LayerNorm / BatchNorm scale
        a = 1.0 + 0.1 * rng.standard_normal(shape)
    else:
        # conv / linear / GRU matrices: N(0, gain^2 / fan_in)
        fan_in = int(np.prod(shape[1:]))
        if "upsamples" in name or ".2.conv.weight" in name and "up_blocks" in name:
            # ConvTranspose1d weight is (C_in, C_out, k) with k = 2 * stride everywhere on this
            # path: each output sample sees C_in * 2 taps
            fan_in = shape[0] * 2
        gain = 1.0
        if "W_scale.weight" in name or "W_bias.weight" in name:
            gain = 0.3
        a = gain * rng.standard_normal(shape) / np.sqrt(fan_in)
    return torch.from_numpy(np.asarray(a, dtype=np.float32)).to(dtype)


def fill_state_dict(named, seed=0):
    """named: mapping name -> tensor (only shape/dtype are read).  Returns a new dict."""
    out = {}
    for name, ref in named.items():
        out[name] = _fill_one(name, ref.shape, ref.dtype, seed)
    # weight-norm pairs: g = ||v|| * u, u ~ U(0.8, 1.2)  (so the folded weight keeps fan-in scale)
    for name in list(out):
        if name.endswith("weight_g"):
            v = out[name[:-1] + "v"]
            norm = v.reshape(v.shape[0], -1).norm(dim=1)
            u = torch.from_numpy(_rng(name, seed).uniform(0.8, 1.2, size=norm.shape).astype(np.float32))
            out[name] = (norm * u).reshape(out[name].shape)
    return out


# HiFi-GAN generator_params used for every synthetic run (declared synthetic: the
# recipe's real vocoder config.yml lives in a download, conf/serenade.yaml:42-45).
# hop = 8*5*3*2 = 240 = conf/serenade.yaml:6.
HIFIGAN_PARAMS = dict(
    in_channels=80, out_channels=1, channels=512, kernel_size=7,
    upsample_scales=(8, 5, 3, 2), upsample_kernel_sizes=(16, 10, 6, 4),
    resblock_kernel_sizes=(3, 7, 11), resblock_dilations=[(1, 3, 5), (1, 3, 5), (1, 3, 5)],
    use_additional_convs=True, bias=True, nonlinear_activation="LeakyReLU",
    nonlinear_activation_params={"negative_slope": 0.1}, use_causal_conv=False,
    use_weight_norm=True,
)

SERENADE_PARAMS = dict(
    input_dim=768, output_dim=80, encoder_channels=80, decoder_channels=512,
    gst_embed_dim=256, decoder_attention_head_dim=512, mask_size=[0.1, 0.5],
)


def synth_inputs(B, T, T_ref=256, seed=1234, lengths=None, ref_lengths=None, temperature=0.667):
    """Synthetic utterance batch of SURVEY.md section 8d (numpy default_rng so it is
    torch-version independent).  Returns a dict of CPU fp32 / int64 tensors."""
    rng = np.random.default_rng(seed)
    f = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
    d = dict(
        x=f(rng.standard_normal((B, T, 768))),
        midi=f(rng.uniform(0, 1, (B, T, 1))),
        lft=f(rng.uniform(0, 1, (B, T, 1))),
        ref_x=f(rng.standard_normal((B, T_ref, 768))),
        ref_logmel=f(rng.standard_normal((B, T_ref, 80))),
        ref_midi=f(rng.uniform(0, 1, (B, T_ref, 1))),
        ref_lft=f(rng.uniform(0, 1, (B, T_ref, 1))),
        z=f(rng.standard_normal((B, 80, T_ref + T)) * temperature),
    )
    d["lengths"] = torch.tensor([T] * B if lengths is None else list(lengths), dtype=torch.int64)
    d["ref_lengths"] = torch.tensor([T_ref] * B if ref_lengths is None else list(ref_lengths),
                                    dtype=torch.int64)
    return d
