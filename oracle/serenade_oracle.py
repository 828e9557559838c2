"""CPU ORACLE — test infrastructure, NOT product code.

A functional, pure-PyTorch (CPU, fp32 or fp64) restatement of the reference's
audio-infilling inference hot path.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import this file; the product path
(``serenade_amd``) never does and fails loudly when its HIP library is missing.

Pinning: every function below is checked against golden vectors captured from
the reference itself (``tests/golden/make_golden.py`` imports /root/reference in
the build container; fixtures in ``tests/golden/*.npz``), plus the mask tables in
``serenade/utils/masking.py:22-26,142-146``.  The reference's transformer block
delegates its arithmetic to the un-vendored, unpinned ``diffusers`` package
(setup.cfg:21); the golden vectors for that row were produced with a stand-in
that follows diffusers' published ``AttnProcessor2_0``/``GEGLU`` semantics, so
row a4.6 is "parity unpinned" with respect to diffusers itself.

All weights come in as a flat ``dict[str, Tensor]`` using the reference's
``state_dict`` key names, so a reference checkpoint can be
fed in unchanged.  Tensors follow the reference layouts: (B, C, T) for the
convolutional parts, (B, T, C) for the transformer parts.
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- masks
def make_pad_mask(lengths, maxlen=None):
    """serenade/utils/masking.py:4-121 (lengths-only form): True on padded frames."""
    if not isinstance(lengths, (list, tuple)):
        lengths = [int(v) for v in lengths.reshape(-1).tolist()]
    if maxlen is None:
        maxlen = int(max(lengths))
    rng = torch.arange(maxlen, dtype=torch.int64).unsqueeze(0)
    return rng >= torch.tensor(lengths, dtype=torch.int64).unsqueeze(1)


def make_non_pad_mask(lengths, maxlen=None):
    """serenade/utils/masking.py:124-210: True on valid frames."""
    return ~make_pad_mask(lengths, maxlen)


# --------------------------------------------------------------------------- helpers
def _sub(w, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in w.items() if k.startswith(prefix)}


def _wn_weight(w, name):
    """weight-norm folded weight: g * v / ||v|| over all dims but 0
    (torch.nn.utils.weight_norm default dim=0; serenade.py:359-360)."""
    if name + ".weight" in w:
        return w[name + ".weight"]
    g = w[name + ".weight_g"]
    v = w[name + ".weight_v"]
    norm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(g.shape)
    return v * (g / norm)


# --------------------------------------------------------------------------- a2 encoder
def conv1d_resnet(w, x):
    """Conv1dResnet.forward, serenade/models/serenade.py:282-296,310-342 with
    ResnetBlock :363-376.  x (B, T, in_dim) -> (B, T, out_dim).  num_layers = 2."""
    h = x.transpose(1, 2)
    h = F.conv1d(F.pad(h, (3, 3), mode="reflect"), _wn_weight(w, "model.1"), w["model.1.bias"])
    n_layers = 0
    while f"model.{2 + n_layers}.shortcut.bias" in w:
        n_layers += 1
    for n in range(n_layers):
        p = f"model.{2 + n}"
        d = 2 ** n
        sc = F.conv1d(h, _wn_weight(w, p + ".shortcut"), w[p + ".shortcut.bias"])
        b = F.leaky_relu(h, 0.2)
        b = F.conv1d(F.pad(b, (d, d), mode="reflect"), _wn_weight(w, p + ".block.2"),
                     w[p + ".block.2.bias"], dilation=d)
        b = F.leaky_relu(b, 0.2)
        b = F.conv1d(b, _wn_weight(w, p + ".block.4"), w[p + ".block.4.bias"])
        h = sc + b
    last = f"model.{2 + n_layers + 2}"
    h = F.leaky_relu(h, 0.2)
    h = F.conv1d(F.pad(h, (3, 3), mode="reflect"), _wn_weight(w, last), w[last + ".bias"])
    return h.transpose(1, 2)


# --------------------------------------------------------------------------- a5 GST
def reference_encoder(w, speech, bn_training=False):
    """ReferenceEncoder.forward, serenade/modules/gst/style_encoder.py:171-191.
    speech (B, T_ref, 80) -> (B, gru_units).  BatchNorm2d in eval mode (running stats) unless bn_training
    (model.train(): batch statistics; the running statistics are not updated here)."""
    h = speech.unsqueeze(1)
    i = 0
    while f"convs.{3 * i}.weight" in w:
        c, b = 3 * i, 3 * i + 1
        h = F.conv2d(h, w[f"convs.{c}.weight"], None, stride=2, padding=1)
        if bn_training:
            h = F.batch_norm(h, None, None, w[f"convs.{b}.weight"], w[f"convs.{b}.bias"], True, 0.0, 1e-5)
        else:
            h = F.batch_norm(h, w[f"convs.{b}.running_mean"], w[f"convs.{b}.running_var"],
                             w[f"convs.{b}.weight"], w[f"convs.{b}.bias"], False, 0.0, 1e-5)
        h = F.relu(h)
        i += 1
    h = h.transpose(1, 2)  # (B, T', C, F')
    bsz, tlen = h.shape[0], h.shape[1]
    h = h.contiguous().view(bsz, tlen, -1)
    return gru_last(w, h)


def gru_last(w, xs):
    """torch.nn.GRU(batch_first) last hidden state (style_encoder.py:169,188-189).
    Gate order r, z, n; n = tanh(W_in x + b_in + r * (W_hn h + b_hn))."""
    wih, whh = w["gru.weight_ih_l0"], w["gru.weight_hh_l0"]
    bih, bhh = w["gru.bias_ih_l0"], w["gru.bias_hh_l0"]
    hdim = whh.shape[1]
    h = xs.new_zeros(xs.shape[0], hdim)
    for t in range(xs.shape[1]):
        gi = xs[:, t] @ wih.t() + bih
        gh = h @ whh.t() + bhh
        r = torch.sigmoid(gi[:, :hdim] + gh[:, :hdim])
        z = torch.sigmoid(gi[:, hdim:2 * hdim] + gh[:, hdim:2 * hdim])
        n = torch.tanh(gi[:, 2 * hdim:] + r * gh[:, 2 * hdim:])
        h = (1.0 - z) * n + z * h
    return h


def style_token_layer(w, ref_embs, n_head=4):
    """StyleTokenLayer.forward, style_encoder.py:235-252 with the default path of
    MultiHeadedAttention (gst/attention.py:110-184,298-300), mask=None."""
    bsz = ref_embs.shape[0]
    toks = torch.tanh(w["gst_embs"])  # (n_tok, 64)
    q = ref_embs @ w["mha.linear_q.weight"].t() + w["mha.linear_q.bias"]  # (B, 256)
    k = toks @ w["mha.linear_k.weight"].t() + w["mha.linear_k.bias"]  # (n_tok, 256)
    v = toks @ w["mha.linear_v.weight"].t() + w["mha.linear_v.bias"]
    n_feat = q.shape[-1]
    dk = n_feat // n_head
    qh = q.view(bsz, n_head, dk)
    kh = k.view(-1, n_head, dk)
    vh = v.view(-1, n_head, dk)
    scores = torch.einsum("bhd,thd->bht", qh, kh) / math.sqrt(dk)
    attn = torch.softmax(scores, dim=-1)
    ctx = torch.einsum("bht,thd->bhd", attn, vh).reshape(bsz, n_feat)
    return ctx @ w["mha.linear_out.weight"].t() + w["mha.linear_out.bias"]


def style_encoder(w, speech, bn_training=False):
    """StyleEncoder.forward, style_encoder.py:78-91."""
    return style_token_layer(_sub(w, "stl."), reference_encoder(_sub(w, "ref_enc."), speech, bn_training))


# --------------------------------------------------------------------------- a4 UNet
def sinusoidal_pos_emb(t, dim, scale=1000.0):
    """SinusoidalPosEmb.forward, matcha_components/decoder.py:54-63."""
    if t.ndim < 1:
        t = t.unsqueeze(0)
    half = dim // 2
    e = math.log(10000) / (half - 1)
    e = torch.exp(torch.arange(half).float() * -e).to(t.dtype)
    e = scale * t.unsqueeze(1) * e.unsqueeze(0)
    return torch.cat((e.sin(), e.cos()), dim=-1)


def timestep_embedding(w, s):
    """TimestepEmbedding.forward (act silu), decoder.py:145-157."""
    s = F.linear(s, w["linear_1.weight"], w["linear_1.bias"])
    s = F.silu(s)
    return F.linear(s, w["linear_2.weight"], w["linear_2.bias"])


def block1d(w, x, mask):
    """Block1D.forward, decoder.py:66-77: (x*mask) -> Conv1d k3 -> GroupNorm(8) -> Mish -> *mask.
    GroupNorm statistics run over the full padded length."""
    h = F.conv1d(x * mask, w["block.0.weight"], w["block.0.bias"], padding=1)
    h = F.group_norm(h, 8, w["block.1.weight"], w["block.1.bias"], 1e-5)
    return F.mish(h) * mask


def speaker_adapter(w, x, spk, eps=1e-5):
    """SpeakerAdapter.forward, decoder.py:34-45 (per-frame LN over C, conditional affine)."""
    xt = x.transpose(1, -1)
    mean = xt.mean(dim=-1, keepdim=True)
    var = ((xt - mean) ** 2).mean(dim=-1, keepdim=True)
    y = (xt - mean) / (var + eps).sqrt()
    scale = F.linear(spk, w["W_scale.weight"], w["W_scale.bias"])
    bias = F.linear(spk, w["W_bias.weight"], w["W_bias.bias"])
    y = y * scale.unsqueeze(1) + bias.unsqueeze(1)
    return y.transpose(1, -1)


def resnet_block1d(w, x, mask, temb, spk):
    """ResnetBlock1D.forward, decoder.py:95-101."""
    h = block1d(_sub(w, "block1."), x, mask)
    h = h + F.linear(F.mish(temb), w["mlp.1.weight"], w["mlp.1.bias"]).unsqueeze(-1)
    h = block1d(_sub(w, "block2."), h, mask)
    out = h + F.conv1d(x * mask, w["res_conv.weight"], w["res_conv.bias"])
    return speaker_adapter(_sub(w, "speaker_projection."), out, spk)


def self_attention(w, x, key_mask, heads=4):
    """diffusers Attention (self-attention), called at transformer.py:292-301.
    x (B, L, C); key_mask bool (B, L) True = attend.  Explicit softmax(QK^T/sqrt(d)+mask)V."""
    b, l, _ = x.shape
    q = F.linear(x, w["to_q.weight"])
    k = F.linear(x, w["to_k.weight"])
    v = F.linear(x, w["to_v.weight"])
    d = q.shape[-1] // heads
    q = q.view(b, l, heads, d).transpose(1, 2)
    k = k.view(b, l, heads, d).transpose(1, 2)
    v = v.view(b, l, heads, d).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(d)
    if key_mask is not None:
        s = s.masked_fill(~key_mask.to(torch.bool).view(b, 1, 1, l), float("-inf"))
    o = torch.softmax(s, dim=-1) @ v
    o = o.transpose(1, 2).reshape(b, l, heads * d)
    return F.linear(o, w["to_out.0.weight"], w["to_out.0.bias"])


def feed_forward_geglu(w, x):
    """FeedForward with the default GEGLU branch (transformer.py:120-146; 'snake' falls
    through to GEGLU): proj -> chunk(2) -> h * gelu_erf(g) -> Linear."""
    h, g = F.linear(x, w["net.0.proj.weight"], w["net.0.proj.bias"]).chunk(2, dim=-1)
    return F.linear(h * F.gelu(g), w["net.2.weight"], w["net.2.bias"])


def basic_transformer_block(w, x, key_mask):
    """BasicTransformerBlock.forward effective path, transformer.py:286,292-304,322,347,352."""
    n = F.layer_norm(x, (x.shape[-1],), w["norm1.weight"], w["norm1.bias"], 1e-5)
    x = self_attention(_sub(w, "attn1."), n, key_mask) + x
    n = F.layer_norm(x, (x.shape[-1],), w["norm3.weight"], w["norm3.bias"], 1e-5)
    return feed_forward_geglu(_sub(w, "ff."), n) + x


def decoder_forward(w, x, mask, mu, t, spk, taps=None):
    """Decoder.forward, matcha_components/decoder.py:384-467.
    x (B, 80, L), mask (B, 1, L) {0,1} float or bool, mu (B, 162, L), t 0-dim, spk (B, 256).
    ``taps`` (optional dict) receives named intermediates for per-stage parity tests."""
    mask = mask.to(x.dtype)
    in_ch = w["time_mlp.linear_1.weight"].shape[1]
    temb = timestep_embedding(_sub(w, "time_mlp."), sinusoidal_pos_emb(t, in_ch))
    max_shape = x.shape[-1]
    h = torch.cat([x, mu], dim=1)
    hiddens, masks = [], [mask]

    def tfm(prefix, h, m):
        h = h.transpose(1, 2)
        h = basic_transformer_block(_sub(w, prefix), h, m[:, 0, :] > 0)
        return h.transpose(1, 2)

    n_down = 0
    while f"down_blocks.{n_down}.0.mlp.1.weight" in w:
        n_down += 1
    for i in range(n_down):
        p = f"down_blocks.{i}."
        m = masks[-1]
        h = resnet_block1d(_sub(w, p + "0."), h, m, temb, spk)
        if taps is not None:
            taps[p + "0"] = h
        h = tfm(p + "1.0.", h, m)
        if taps is not None:
            taps[p + "1.0"] = h
        hiddens.append(h)
        if p + "2.conv.weight" in w:
            h = F.conv1d(h * m, w[p + "2.conv.weight"], w[p + "2.conv.bias"], stride=2, padding=1)
        else:
            h = F.conv1d(h * m, w[p + "2.weight"], w[p + "2.bias"], padding=1)
        masks.append(m[:, :, ::2])
    masks = masks[:-1]
    m_mid = masks[-1]
    i = 0
    while f"mid_blocks.{i}.0.mlp.1.weight" in w:
        p = f"mid_blocks.{i}."
        h = resnet_block1d(_sub(w, p + "0."), h, m_mid, temb, spk)
        h = tfm(p + "1.0.", h, m_mid)
        i += 1
    if taps is not None:
        taps["mid"] = h
    i = 0
    while f"up_blocks.{i}.0.mlp.1.weight" in w:
        p = f"up_blocks.{i}."
        m = masks.pop()
        h = h[:, :, :max_shape]
        h = resnet_block1d(_sub(w, p + "0."), torch.cat([h, hiddens.pop()], dim=1), m, temb, spk)
        h = tfm(p + "1.0.", h, m)
        if p + "2.conv.weight" in w:
            h = F.conv_transpose1d(h * m, w[p + "2.conv.weight"], w[p + "2.conv.bias"], stride=2, padding=1)
        else:
            h = F.conv1d(h * m, w[p + "2.weight"], w[p + "2.bias"], padding=1)
        if taps is not None:
            taps[p + "2"] = h
        i += 1
    h = block1d(_sub(w, "final_block."), h, m)
    out = F.conv1d(h * m, w["final_proj.weight"], w["final_proj.bias"])
    return out * mask


# --------------------------------------------------------------------------- a3 CFM
def t_schedule(n_timesteps, dtype=torch.float32):
    """(t_k, dt_k) exactly as solve_euler accumulates them in fp32,
    matcha_components/flow_matching.py:61,79-91."""
    t_span = torch.linspace(0, 1, n_timesteps + 1, dtype=dtype)
    t, dt = t_span[0], t_span[1] - t_span[0]
    ts, dts = [], []
    for step in range(1, len(t_span)):
        ts.append(t.clone())
        dts.append(dt.clone())
        t = t + dt
        if step < len(t_span) - 1:
            dt = t_span[step + 1] - t
    return ts, dts


def solve_euler(w, z, mu, mask, spk, n_timesteps=10, trace=None):
    """CFM.solve_euler, flow_matching.py:65-93.  ``w`` = estimator weights
    (prefix 'cfm_decoder.estimator.' stripped).  z is the explicit, already
    temperature-scaled noise (flow_matching.py:57-60 draws it on the CPU)."""
    ts, dts = t_schedule(n_timesteps)
    x = z
    for t, dt in zip(ts, dts):
        d = decoder_forward(w, x, mask, mu, t.to(x.dtype), spk)
        x = x + dt.to(x.dtype) * d
        if trace is not None:
            trace.append(x.clone())
    return x


# --------------------------------------------------------------------------- a1 wiring
def serenade_inference(w, x, lengths, midi, lft, ref_x, ref_lengths, ref_logmel, ref_midi,
                       ref_lft, z, n_timesteps=10, trace=None):
    """Serenade.inference, serenade/models/serenade.py:168-221, with explicit noise ``z``
    (B, 80, T_ref+T) = randn * temperature.  Returns (T, 80) if B == 1 else (B, T, 80)."""
    enc_w = _sub(w, "encoder.")
    enc = conv1d_resnet(enc_w, x)
    spk = style_encoder(_sub(w, "gst."), ref_logmel)
    ref_enc = conv1d_resnet(enc_w, ref_x)
    ref_mu = torch.cat([ref_enc, ref_midi, ref_lft, ref_logmel], dim=-1)
    src_mu = torch.cat([enc, midi, lft, torch.zeros_like(enc[..., :ref_logmel.shape[-1]])], dim=-1)
    mu = torch.cat([ref_mu, src_mu], dim=1)
    total = lengths + ref_lengths
    mask = make_non_pad_mask(total).unsqueeze(1)
    mel = solve_euler(_sub(w, "cfm_decoder.estimator."), z, mu.permute(0, 2, 1), mask, spk,
                      n_timesteps, trace=trace).permute(0, 2, 1)
    mel = mel[:, int(ref_lengths[0]):, :]
    return mel.squeeze(0)


# --------------------------------------------------------------------------- a1' training-loss forward
def cfm_compute_loss(w, x1, mask, mu, spk, mask_l, t, z, sigma_min=1e-4):
    """CFM.compute_loss, flow_matching.py:95-133, with the random draws (t (B,1,1), z like x1) made explicit."""
    y = (1 - (1 - sigma_min) * t) * z + t * x1
    u = x1 - (1 - sigma_min) * z
    den = decoder_forward(w, y, mask, mu, t.squeeze(), spk)
    if mask_l is not None:
        den = den * mask_l
        u = u * mask_l
    loss = F.mse_loss(den, u, reduction="sum")
    denom = (torch.sum(mask_l) if mask_l is not None else torch.sum(mask)) * u.shape[1]
    return loss / denom, y


def serenade_forward(w, x, lengths, logmel, midi, lft, uniform, seg_start, t, z, mask_size=(0.1, 0.5),
                     bn_training=False):
    """Serenade.forward, serenade/models/serenade.py:90-166.  `uniform` is the value random.uniform(*mask_size)
    returned, `seg_start` the value of random.randint, (t, z) the draws of CFM.compute_loss."""
    ret = {}
    enc = conv1d_resnet(_sub(w, "encoder."), x)
    ret["gauss_mel"] = enc
    spk = style_encoder(_sub(w, "gst."), logmel, bn_training)
    mask = make_non_pad_mask(lengths).unsqueeze(1)
    msize = int(uniform * enc.size(1))
    seg_end = seg_start + msize
    mask_l = mask.clone()
    mask_l[:, :, 0:seg_start] = 0
    mask_l[:, :, seg_end:] = 0
    mask_c = mask.clone()
    mask_c[:, :, seg_start:seg_end] = 0
    prior = torch.sum(0.5 * ((logmel.permute(0, 2, 1) - enc.permute(0, 2, 1)) ** 2 + math.log(2 * math.pi)) * mask)
    ret["prior_loss"] = prior / (torch.sum(mask) * logmel.shape[-1])
    targets = logmel * mask_l.permute(0, 2, 1)
    cond = logmel * mask_c.permute(0, 2, 1)
    mu = torch.cat([enc, midi, lft, cond], dim=-1)
    ret["cfm_loss"], _ = cfm_compute_loss(_sub(w, "cfm_decoder.estimator."), targets.permute(0, 2, 1), mask,
                                          mu.permute(0, 2, 1), spk, mask_l, t, z)
    return ret


# --------------------------------------------------------------------------- a8 HiFi-GAN
def hifigan_residual_block(w, x, kernel_size, dilations, slope=0.1):
    """HiFiGANResidualBlock.forward, serenade/vocoder/layers/residual_block.py:243-258."""
    for idx, d in enumerate(dilations):
        xt = F.conv1d(F.leaky_relu(x, slope), _wn_weight(w, f"convs1.{idx}.1"), w[f"convs1.{idx}.1.bias"],
                      dilation=d, padding=(kernel_size - 1) // 2 * d)
        if f"convs2.{idx}.1.bias" in w:
            xt = F.conv1d(F.leaky_relu(xt, slope), _wn_weight(w, f"convs2.{idx}.1"), w[f"convs2.{idx}.1.bias"],
                          padding=(kernel_size - 1) // 2)
        x = xt + x
    return x


def hifigan_forward(w, c, cfg):
    """HiFiGANGenerator.forward, serenade/vocoder/models/hifigan.py:171-190.
    c (B, 80, T) -> (B, 1, T * prod(upsample_scales)).  cfg = generator_params."""
    ks = cfg.get("kernel_size", 7)
    scales = cfg["upsample_scales"]
    rks = cfg.get("resblock_kernel_sizes", (3, 7, 11))
    rds = cfg.get("resblock_dilations", [(1, 3, 5)] * 3)
    slope = cfg.get("nonlinear_activation_params", {"negative_slope": 0.1})["negative_slope"]
    c = F.conv1d(c, _wn_weight(w, "input_conv"), w["input_conv.bias"], padding=(ks - 1) // 2)
    nb = len(rks)
    for i, s in enumerate(scales):
        c = F.conv_transpose1d(F.leaky_relu(c, slope), _wn_weight(w, f"upsamples.{i}.1"),
                               w[f"upsamples.{i}.1.bias"], stride=s,
                               padding=s // 2 + s % 2, output_padding=s % 2)
        cs = 0.0
        for j in range(nb):
            cs = cs + hifigan_residual_block(_sub(w, f"blocks.{i * nb + j}."), c, rks[j], rds[j], slope)
        c = cs / nb
    c = F.conv1d(F.leaky_relu(c, 0.01), _wn_weight(w, "output_conv.1"), w["output_conv.1.bias"],
                 padding=(ks - 1) // 2)
    return torch.tanh(c)


def vocoder_decode(w, c, cfg, stats, trg_stats=None):
    """Vocoder.decode / decode_batch, serenade/vocoder/vocoder.py:51-75.
    c (T, 80) -> (T*hop,)   or   (B, T, 80) -> (B, T*hop)."""
    if trg_stats is not None:
        c = c * trg_stats["scale"] + trg_stats["mean"]
    c = (c - stats["mean"]) / stats["scale"]
    if c.ndim == 2:
        y = hifigan_forward(w, c.transpose(1, 0).unsqueeze(0), cfg)
        return y.squeeze(0).transpose(1, 0).reshape(-1)
    return hifigan_forward(w, c.transpose(2, 1), cfg).squeeze(1)
