"""state_dict layout of the reference modules (names, shapes, dtypes) and a generic
container builder, so reference checkpoints load into the HIP-backed modules unchanged
(verified against tests/golden/state_dict_keys.json, which was
captured from the instantiated reference).

Reference constructors restated here:
  Serenade            serenade/models/serenade.py:36-88
  Conv1dResnet        serenade/models/serenade.py:244-308 (+ ResnetBlock :363-373)
  StyleEncoder        serenade/modules/gst/style_encoder.py:46-76,118-169,213-233,258-275
  CFM / Decoder       matcha_components/flow_matching.py:10-34, decoder.py:209-343
  HiFiGANGenerator    serenade/vocoder/models/hifigan.py:24-169
"""
from collections import OrderedDict

import torch
import torch.nn as nn

F32 = torch.float32


def _wn(d, p, co, ci, k):
    d[p + ".bias"] = ((co,), F32)
    d[p + ".weight_g"] = ((co, 1, 1), F32)
    d[p + ".weight_v"] = ((co, ci, k), F32)


def _conv(d, p, co, ci, k, weight_norm=False, transpose=False):
    shape = (ci, co, k) if transpose else (co, ci, k)
    d[p + ".bias"] = ((co,), F32)
    if weight_norm:
        d[p + ".weight_g"] = ((shape[0], 1, 1), F32)
        d[p + ".weight_v"] = (shape, F32)
    else:
        d[p + ".weight"] = (shape, F32)


def _linear(d, p, co, ci, bias=True):
    d[p + ".weight"] = ((co, ci), F32)
    if bias:
        d[p + ".bias"] = ((co,), F32)


def encoder_shapes(in_dim=768, hidden_dim=512, out_dim=80, num_layers=2):
    d = OrderedDict()
    _wn(d, "model.1", hidden_dim, in_dim, 7)
    for n in range(num_layers):
        p = f"model.{2 + n}"
        _wn(d, p + ".block.2", hidden_dim, hidden_dim, 3)
        _wn(d, p + ".block.4", hidden_dim, hidden_dim, 1)
        _wn(d, p + ".shortcut", hidden_dim, hidden_dim, 1)
    _wn(d, f"model.{2 + num_layers + 2}", out_dim, hidden_dim, 7)
    return d


def gst_shapes(idim=80, gst_tokens=50, gst_token_dim=256, gst_heads=4,
               conv_chans_list=(128, 128, 256, 256, 512, 512), gru_units=128):
    d = OrderedDict()
    f = idim
    for i, co in enumerate(conv_chans_list):
        ci = 1 if i == 0 else conv_chans_list[i - 1]
        d[f"ref_enc.convs.{3 * i}.weight"] = ((co, ci, 3, 3), F32)
        b = f"ref_enc.convs.{3 * i + 1}"
        d[b + ".weight"] = ((co,), F32)
        d[b + ".bias"] = ((co,), F32)
        d[b + ".running_mean"] = ((co,), F32)
        d[b + ".running_var"] = ((co,), F32)
        d[b + ".num_batches_tracked"] = ((), torch.int64)
        f = (f - 3 + 2) // 2 + 1
    gin = f * conv_chans_list[-1]
    d["ref_enc.gru.weight_ih_l0"] = ((3 * gru_units, gin), F32)
    d["ref_enc.gru.weight_hh_l0"] = ((3 * gru_units, gru_units), F32)
    d["ref_enc.gru.bias_ih_l0"] = ((3 * gru_units,), F32)
    d["ref_enc.gru.bias_hh_l0"] = ((3 * gru_units,), F32)
    dk = gst_token_dim // gst_heads
    d["stl.gst_embs"] = ((gst_tokens, dk), F32)
    _linear(d, "stl.mha.linear_q", gst_token_dim, gru_units)
    _linear(d, "stl.mha.linear_k", gst_token_dim, dk)
    _linear(d, "stl.mha.linear_v", gst_token_dim, dk)
    _linear(d, "stl.mha.linear_out", gst_token_dim, gst_token_dim)
    return d


def _resnet(d, p, cin, cout, temb_dim, spk_dim):
    _linear(d, p + ".mlp.1", cout, temb_dim)
    for blk, ci in (("block1", cin), ("block2", cout)):
        _conv(d, f"{p}.{blk}.block.0", cout, ci, 3)
        d[f"{p}.{blk}.block.1.weight"] = ((cout,), F32)
        d[f"{p}.{blk}.block.1.bias"] = ((cout,), F32)
    _conv(d, p + ".res_conv", cout, cin, 1)
    _linear(d, p + ".speaker_projection.W_scale", cout, spk_dim)
    _linear(d, p + ".speaker_projection.W_bias", cout, spk_dim)


def _tfm(d, p, dim, heads, head_dim):
    inner = heads * head_dim
    d[p + ".norm1.weight"] = ((dim,), F32)
    d[p + ".norm1.bias"] = ((dim,), F32)
    _linear(d, p + ".attn1.to_q", inner, dim, bias=False)
    _linear(d, p + ".attn1.to_k", inner, dim, bias=False)
    _linear(d, p + ".attn1.to_v", inner, dim, bias=False)
    _linear(d, p + ".attn1.to_out.0", dim, inner)
    d[p + ".norm3.weight"] = ((dim,), F32)
    d[p + ".norm3.bias"] = ((dim,), F32)
    _linear(d, p + ".ff.net.0.proj", dim * 4 * 2, dim)
    _linear(d, p + ".ff.net.2", dim, dim * 4)


def decoder_shapes(in_channels=242, out_channels=80, spk_embed_dim=256, channels=(512, 512),
                   attention_head_dim=512, n_blocks=1, num_mid_blocks=2, num_heads=4):
    assert n_blocks == 1
    channels = tuple(channels)
    d = OrderedDict()
    temb = channels[0] * 4
    _linear(d, "time_mlp.linear_1", temb, in_channels)
    _linear(d, "time_mlp.linear_2", temb, temb)
    co = in_channels
    for i in range(len(channels)):
        ci, co = co, channels[i]
        p = f"down_blocks.{i}"
        _resnet(d, p + ".0", ci, co, temb, spk_embed_dim)
        _tfm(d, p + ".1.0", co, num_heads, attention_head_dim)
        if i < len(channels) - 1:
            _conv(d, p + ".2.conv", co, co, 3)
        else:
            _conv(d, p + ".2", co, co, 3)
    for i in range(num_mid_blocks):
        p = f"mid_blocks.{i}"
        _resnet(d, p + ".0", channels[-1], co, temb, spk_embed_dim)
        _tfm(d, p + ".1.0", co, num_heads, attention_head_dim)
    ch = channels[::-1] + (channels[0],)
    for i in range(len(ch) - 1):
        ci, co = ch[i], ch[i + 1]
        p = f"up_blocks.{i}"
        _resnet(d, p + ".0", 2 * ci, co, temb, spk_embed_dim)
        _tfm(d, p + ".1.0", co, num_heads, attention_head_dim)
        if i < len(ch) - 2:
            _conv(d, p + ".2.conv", co, co, 4, transpose=True)
        else:
            _conv(d, p + ".2", co, co, 3)
    _conv(d, "final_block.block.0", ch[-1], ch[-1], 3)
    d["final_block.block.1.weight"] = ((ch[-1],), F32)
    d["final_block.block.1.bias"] = ((ch[-1],), F32)
    _conv(d, "final_proj", out_channels, ch[-1], 1)
    return d


def serenade_shapes(input_dim=768, output_dim=80, encoder_channels=80, decoder_channels=512,
                    gst_embed_dim=256, decoder_attention_head_dim=512, **_):
    d = OrderedDict()
    for k, v in encoder_shapes(input_dim, 512, encoder_channels, 2).items():
        d["encoder." + k] = v
    for k, v in gst_shapes(gst_tokens=50, gst_token_dim=gst_embed_dim).items():
        d["gst." + k] = v
    cond = output_dim + encoder_channels + 2
    for k, v in decoder_shapes(cond + output_dim, output_dim, gst_embed_dim,
                               (decoder_channels, decoder_channels),
                               decoder_attention_head_dim).items():
        d["cfm_decoder.estimator." + k] = v
    return d


def hifigan_shapes(in_channels=80, out_channels=1, channels=512, kernel_size=7,
                   upsample_scales=(8, 8, 2, 2), upsample_kernel_sizes=(16, 16, 4, 4),
                   resblock_kernel_sizes=(3, 7, 11),
                   resblock_dilations=((1, 3, 5), (1, 3, 5), (1, 3, 5)),
                   use_additional_convs=True, weight_norm=True, **_):
    d = OrderedDict()
    _conv(d, "input_conv", channels, in_channels, kernel_size, weight_norm)
    for i, k in enumerate(upsample_kernel_sizes):
        _conv(d, f"upsamples.{i}.1", channels // 2 ** (i + 1), channels // 2 ** i, k, weight_norm,
              transpose=True)
    nb = len(resblock_kernel_sizes)
    for i in range(len(upsample_kernel_sizes)):
        c = channels // 2 ** (i + 1)
        for j in range(nb):
            p = f"blocks.{i * nb + j}"
            for idx in range(len(resblock_dilations[j])):
                _conv(d, f"{p}.convs1.{idx}.1", c, c, resblock_kernel_sizes[j], weight_norm)
            if use_additional_convs:
                for idx in range(len(resblock_dilations[j])):
                    _conv(d, f"{p}.convs2.{idx}.1", c, c, resblock_kernel_sizes[j], weight_norm)
    _conv(d, "output_conv.1", out_channels, channels // 2 ** len(upsample_kernel_sizes), kernel_size,
          weight_norm)
    return d


def as_meta(shapes):
    """name -> meta tensor (what utils.synth.fill_state_dict wants)."""
    return OrderedDict((k, torch.empty(s, dtype=dt, device="meta")) for k, (s, dt) in shapes.items())


class ParamTree(nn.Module):
    """A bare container hierarchy whose state_dict keys are exactly the given dotted names.
    Integer-like dtypes / BatchNorm running stats become buffers, everything else a Parameter."""

    def __init__(self, shapes=None):
        super().__init__()
        for name, (shape, dtype) in (shapes or {}).items():
            self._add(name.split("."), shape, dtype)

    def _add(self, parts, shape, dtype):
        if len(parts) == 1:
            leaf = parts[0]
            t = torch.zeros(shape, dtype=dtype)
            if dtype != F32 or leaf in ("running_mean", "running_var"):
                self.register_buffer(leaf, t)
            else:
                self.register_parameter(leaf, nn.Parameter(t, requires_grad=False))
            return
        head = parts[0]
        if head not in self._modules:
            self.add_module(head, ParamTree())
        self._modules[head]._add(parts[1:], shape, dtype)
