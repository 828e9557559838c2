"""Capture golden vectors from the REFERENCE itself (run in the build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Imports /root/reference through tests/golden/_ref_harness.py (I/O stubs + diffusers
stand-in), fills every reference module with the procedural
name-keyed weights of serenade_amd/utils/synth.py (seed 0) and stores inputs and
outputs as small .npz fixtures next to this file.  The reference's files never
travel; only these vectors do.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import _ref_harness  # noqa: E402

_ref_harness.install()

from serenade.models import Serenade  # noqa: E402
from serenade.utils.masking import make_non_pad_mask, make_pad_mask  # noqa: E402
from serenade.vocoder.models.hifigan import HiFiGANGenerator  # noqa: E402

from serenade_amd.utils.synth import (HIFIGAN_PARAMS, SERENADE_PARAMS, fill_state_dict,  # noqa: E402
                                      synth_inputs)

torch.set_grad_enabled(False)
torch.manual_seed(0)


def npz(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, {k: tuple(np.shape(v)) for k, v in out.items()})


def rnd(rng, *shape):
    return torch.from_numpy(rng.standard_normal(shape).astype(np.float32))


def main():
    model = Serenade(**SERENADE_PARAMS).eval()
    model.load_state_dict(fill_state_dict(model.state_dict(), seed=0))
    est = model.cfm_decoder.estimator
    rng = np.random.default_rng(777)

    # ---- a6 masks
    npz("masks", lengths=np.array([5, 3, 2]), pad=make_pad_mask([5, 3, 2]),
        non_pad=make_non_pad_mask([5, 3, 2]))

    # ---- a2 encoder (odd and tiny lengths; reflection pad needs T > 3)
    x = rnd(rng, 2, 33, 768)
    npz("encoder", x=x, y=model.encoder(x, None))

    # ---- a5 GST
    sp = rnd(rng, 2, 70, 80)
    ref_embs = model.gst.ref_enc(sp)
    npz("gst", speech=sp, ref_embs=ref_embs, style=model.gst(sp))

    # ---- a4.x UNet pieces, full width, tiny length, padded batch + odd L
    for tag, L, lens in (("L48", 48, [48, 37]), ("L65", 65, [65, 50])):
        B = 2
        mask = make_non_pad_mask(lens).unsqueeze(1)
        xin = rnd(rng, B, 80, L)
        mu = rnd(rng, B, 162, L)
        spk = rnd(rng, B, 256)
        t = torch.tensor(0.3)
        temb = est.time_mlp(est.time_embeddings(t))
        h = torch.cat([xin, mu], dim=1)
        rb = est.down_blocks[0][0]
        b1 = rb.block1(h, mask)
        r1 = rb(h, mask, temb, spk)
        tb = est.down_blocks[0][1][0]
        t1 = tb(hidden_states=r1.transpose(1, 2), attention_mask=mask[:, 0, :], timestep=temb,
                speaker_features=spk)
        out = est(xin, mask, mu, t, spk)
        npz("decoder_" + tag, lens=np.array(lens), x=xin, mu=mu, spk=spk, t=t, temb=temb, block1=b1,
            resnet=r1, tfm=t1, out=out)

    # ---- a3 Euler loop with per-step trace (explicit noise)
    L, lens = 48, [48, 37]
    mask = make_non_pad_mask(lens).unsqueeze(1)
    mu = rnd(rng, 2, 162, L)
    spk = rnd(rng, 2, 256)
    z = rnd(rng, 2, 80, L) * 0.667
    cfm = model.cfm_decoder
    t_span = torch.linspace(0, 1, 11)
    # replicate solve_euler but keep the per-step states (flow_matching.py:79-93)
    tt, dt = t_span[0], t_span[1] - t_span[0]
    xs, cur = [], z
    for step in range(1, len(t_span)):
        cur = cur + dt * est(cur, mask, mu, tt, spk)
        tt = tt + dt
        xs.append(cur)
        if step < len(t_span) - 1:
            dt = t_span[step + 1] - tt
    final = cfm.solve_euler(z, t_span=t_span, mu=mu, mask=mask, trg_spks=spk)
    assert torch.equal(final, xs[-1])
    npz("euler_L48", lens=np.array(lens), mu=mu, spk=spk, z=z, trace=torch.stack(xs), out=final)

    # ---- a1 full inference chain, B = 1, explicit z via patched randn
    d = synth_inputs(1, 64, T_ref=16, seed=4321)
    orig = torch.randn
    torch.randn = lambda *a, **k: d["z"] / 0.667
    try:
        mel = model.inference(d["x"], d["lengths"], d["midi"], d["lft"], d["ref_x"], d["ref_lengths"],
                              d["ref_logmel"], d["ref_midi"], d["ref_lft"])
    finally:
        torch.randn = orig
    # padded batch B = 2 through the same entry point
    d2 = synth_inputs(2, 40, T_ref=16, seed=4322, lengths=[40, 29])
    torch.randn = lambda *a, **k: d2["z"] / 0.667
    try:
        mel2 = model.inference(d2["x"], d2["lengths"], d2["midi"], d2["lft"], d2["ref_x"], d2["ref_lengths"],
                               d2["ref_logmel"], d2["ref_midi"], d2["ref_lft"])
    finally:
        torch.randn = orig
    npz("inference", mel_b1=mel, mel_b2=mel2)

    # ---- a8 HiFi-GAN + a7 Vocoder arithmetic
    gen = HiFiGANGenerator(**HIFIGAN_PARAMS).eval()
    gen.load_state_dict(fill_state_dict(gen.state_dict(), seed=0))
    gen.remove_weight_norm()
    c = rnd(rng, 2, 80, 12)
    y = gen(c)
    y1 = gen.inference(mel)  # (T*240, 1) from the B = 1 chain above
    npz("hifigan", c=c, y=y, wave_b1=y1.view(-1))
    # small-channel variant (exercises other widths / scales, default ctor kernel sizes)
    small = dict(HIFIGAN_PARAMS, channels=64, upsample_scales=(4, 2), upsample_kernel_sizes=(8, 4),
                 resblock_kernel_sizes=(3, 5), resblock_dilations=[(1, 2), (2, 6, 3)])
    gs = HiFiGANGenerator(**small).eval()
    gs.load_state_dict(fill_state_dict(gs.state_dict(), seed=1))
    gs.remove_weight_norm()
    cs = rnd(rng, 1, 80, 9)
    npz("hifigan_small", c=cs, y=gs(cs))


def forward_case():
    """a1' Serenade.forward (training-loss forward, serenade.py:90-166 + flow_matching.py:95-133): capture the
    random draws the reference makes (python `random` for the infill segment, torch.rand / randn_like for t and z)
    together with its outputs, so the restatement can be checked on identical draws.

        PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py forward
    """
    import random
    model = Serenade(**SERENADE_PARAMS).eval()
    model.load_state_dict(fill_state_dict(model.state_dict(), seed=0))
    rng = np.random.default_rng(778)
    B, Tn = 2, 40
    lens = [40, 31]
    x = rnd(rng, B, Tn, 768)
    logmel = rnd(rng, B, Tn, 80)
    midi = torch.from_numpy(rng.uniform(0, 1, (B, Tn, 1)).astype(np.float32))
    lft = torch.from_numpy(rng.uniform(0, 1, (B, Tn, 1)).astype(np.float32))
    draws = {}
    o_uniform, o_randint, o_rand, o_randn_like = random.uniform, random.randint, torch.rand, torch.randn_like

    def uniform(a, b):
        draws["uniform"] = o_uniform(a, b)
        return draws["uniform"]

    def randint(a, b):
        draws["seg_start"] = o_randint(a, b)
        return draws["seg_start"]

    def rand(*a, **k):
        draws["t"] = o_rand(*a, **k)
        return draws["t"]

    def randn_like(t, **k):
        draws["z"] = o_randn_like(t, **k)
        return draws["z"]

    random.seed(5)
    torch.manual_seed(5)
    random.uniform, random.randint, torch.rand, torch.randn_like = uniform, randint, rand, randn_like
    try:
        ret = model(x, torch.tensor(lens), logmel, midi, lft)
    finally:
        random.uniform, random.randint, torch.rand, torch.randn_like = o_uniform, o_randint, o_rand, o_randn_like
    npz("forward", lens=np.array(lens), x=x, logmel=logmel, midi=midi, lft=lft, uniform=np.float64(draws["uniform"]),
        seg_start=np.int64(draws["seg_start"]), t=draws["t"], z=draws["z"], gauss_mel=ret["gauss_mel"],
        prior_loss=ret["prior_loss"], cfm_loss=ret["cfm_loss"])


def euler20_case():
    """a3 with the step count of BASELINE configs[2] (20 Euler steps), odd padded length: the reference's own
    CFM.solve_euler (flow_matching.py:65-93) on explicit noise, plus the states after steps 1, 10 and 20 from a
    replica of its loop (asserted equal to its result).

        PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py euler20
    """
    model = Serenade(**SERENADE_PARAMS).eval()
    model.load_state_dict(fill_state_dict(model.state_dict(), seed=0))
    est, cfm = model.cfm_decoder.estimator, model.cfm_decoder
    rng = np.random.default_rng(779)
    L, lens, n = 57, [57, 40], 20
    mask = make_non_pad_mask(lens).unsqueeze(1)
    mu, spk = rnd(rng, 2, 162, L), rnd(rng, 2, 256)
    z = rnd(rng, 2, 80, L) * 0.667
    t_span = torch.linspace(0, 1, n + 1)
    tt, dt = t_span[0], t_span[1] - t_span[0]
    keep, cur = {}, z
    for step in range(1, len(t_span)):
        cur = cur + dt * est(cur, mask, mu, tt, spk)
        tt = tt + dt
        if step in (1, 10, 20):
            keep[step] = cur
        if step < len(t_span) - 1:
            dt = t_span[step + 1] - tt
    final = cfm.solve_euler(z, t_span=t_span, mu=mu, mask=mask, trg_spks=spk)
    assert torch.equal(final, keep[20])
    npz("euler20_L57", lens=np.array(lens), mu=mu, spk=spk, z=z, x1=keep[1], x10=keep[10], out=final)


TRAIN_FULL = ["final_proj.weight", "final_proj.bias", "final_block.block.1.weight", "time_mlp.linear_1.bias",
              "down_blocks.0.0.block1.block.1.weight", "down_blocks.0.0.block1.block.1.bias",
              "down_blocks.0.0.speaker_projection.W_scale.bias", "mid_blocks.1.0.mlp.1.bias",
              "mid_blocks.0.1.0.norm1.weight", "up_blocks.1.1.0.norm3.bias", "up_blocks.0.1.0.attn1.to_out.0.bias"]
TRAIN_ROWS = ["down_blocks.0.0.block1.block.0.weight", "down_blocks.0.0.res_conv.weight", "down_blocks.0.2.conv.weight",
              "down_blocks.1.1.0.attn1.to_q.weight", "mid_blocks.0.1.0.attn1.to_k.weight",
              "mid_blocks.1.1.0.attn1.to_v.weight", "up_blocks.0.1.0.ff.net.0.proj.weight",
              "up_blocks.0.1.0.ff.net.2.weight", "up_blocks.0.2.conv.weight", "up_blocks.1.0.block1.block.0.weight",
              "time_mlp.linear_2.weight", "mid_blocks.0.0.speaker_projection.W_bias.weight"]


def train_grads_case():
    """f4: the reference's own CFM.compute_loss (flow_matching.py:95-133) + loss.backward() on explicit inputs, in
    eval mode (dropout off, as the HIP training step), odd padded length.  Stored: the draws it made, the loss, the
    gradient of a dozen whole parameters, the first 4 rows of a dozen big ones, d mu, d spks and the global gradient
    norm over all 192 estimator parameters (what clip_grad_norm_ would report).

        PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py train
    """
    torch.set_grad_enabled(True)
    model = Serenade(**SERENADE_PARAMS).eval()
    model.load_state_dict(fill_state_dict(model.state_dict(), seed=0))
    cfm = model.cfm_decoder
    rng = np.random.default_rng(780)
    L, lens = 45, [45, 31, 38]
    B = len(lens)
    mask = make_non_pad_mask(lens).unsqueeze(1).float()
    mask_l = mask.clone()
    mask_l[:, :, :9] = 0
    mask_l[:, :, 27:] = 0
    x1 = rnd(rng, B, 80, L) * mask_l
    mu = (rnd(rng, B, 162, L) * mask).requires_grad_(True)
    spk = rnd(rng, B, 256).requires_grad_(True)
    draws = {}
    o_rand, o_randn_like = torch.rand, torch.randn_like

    def rand(*a, **k):
        draws["t"] = o_rand(*a, **k)
        return draws["t"]

    def randn_like(t, **k):
        draws["z"] = o_randn_like(t, **k)
        return draws["z"]

    torch.manual_seed(11)
    torch.rand, torch.randn_like = rand, randn_like
    try:
        loss, _ = cfm.compute_loss(x1, mask, mu, spk, mask_l)
    finally:
        torch.rand, torch.randn_like = o_rand, o_randn_like
    loss.backward()
    g = {k: p.grad for k, p in cfm.estimator.named_parameters()}
    assert len(g) == 192 and all(v is not None for v in g.values())
    total = torch.sqrt(sum((v.double() ** 2).sum() for v in g.values()))
    out = dict(lens=np.array(lens), x1=x1, mu=mu, spk=spk, mask_l=mask_l, t=draws["t"], z=draws["z"], loss=loss,
               grad_norm=total, dmu=mu.grad, dspk=spk.grad)
    for k in TRAIN_FULL:
        out["g:" + k] = g[k]
    for k in TRAIN_ROWS:
        out["r:" + k] = g[k][:4]
    npz("train_grads_L45", **out)


FULL_WHOLE = ["encoder.model.1.weight_g", "encoder.model.3.shortcut.weight_g", "encoder.model.6.bias",
              "encoder.model.2.block.2.bias", "gst.ref_enc.convs.0.weight", "gst.ref_enc.convs.1.weight",
              "gst.ref_enc.convs.16.bias", "gst.ref_enc.gru.bias_ih_l0", "gst.ref_enc.gru.bias_hh_l0",
              "gst.stl.gst_embs", "gst.stl.mha.linear_out.bias", "gst.stl.mha.linear_k.bias",
              "cfm_decoder.estimator.final_proj.bias", "cfm_decoder.estimator.mid_blocks.0.1.0.norm1.weight",
              "cfm_decoder.estimator.down_blocks.0.0.speaker_projection.W_scale.bias"]
FULL_ROWS = ["encoder.model.1.weight_v", "encoder.model.2.block.2.weight_v", "encoder.model.3.block.4.weight_v",
             "encoder.model.6.weight_v", "gst.ref_enc.convs.15.weight", "gst.ref_enc.convs.6.weight",
             "gst.ref_enc.gru.weight_ih_l0", "gst.ref_enc.gru.weight_hh_l0", "gst.stl.mha.linear_q.weight",
             "gst.stl.mha.linear_v.weight", "cfm_decoder.estimator.down_blocks.0.0.block1.block.0.weight",
             "cfm_decoder.estimator.up_blocks.0.2.conv.weight"]


def train_full_case():
    """f4, whole model: the reference's Serenade.forward (serenade.py:90-166) in train() mode -- BatchNorm on batch
    statistics -- with every nn.Dropout set to p = 0, loss = cfm_loss + prior_loss (trainers/ssc.py:77-82), backward().
    Stored: inputs, the draws (python random for the infill segment, torch.rand / randn_like for t, z), both losses,
    whole / 4-row gradients of encoder, GST and estimator tensors, the global gradient norm over all parameters, and
    one BatchNorm layer's running statistics after the step.

        PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py train_full
    """
    import random
    torch.set_grad_enabled(True)
    model = Serenade(**SERENADE_PARAMS)
    model.load_state_dict(fill_state_dict(model.state_dict(), seed=0))
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    rng = np.random.default_rng(781)
    lens = [64, 49, 57]
    B, T = len(lens), 64
    x, logmel = rnd(rng, B, T, 768), rnd(rng, B, T, 80)
    midi, lft = rnd(rng, B, T, 1), rnd(rng, B, T, 1)
    draws = {}
    o_uniform, o_randint, o_rand, o_randn_like = random.uniform, random.randint, torch.rand, torch.randn_like

    def uniform(a, b):
        draws["uniform"] = o_uniform(a, b)
        return draws["uniform"]

    def randint(a, b):
        draws["seg_start"] = o_randint(a, b)
        return draws["seg_start"]

    def rand(*a, **k):
        draws["t"] = o_rand(*a, **k)
        return draws["t"]

    def randn_like(t, **k):
        draws["z"] = o_randn_like(t, **k)
        return draws["z"]

    random.seed(6)
    torch.manual_seed(12)
    random.uniform, random.randint, torch.rand, torch.randn_like = uniform, randint, rand, randn_like
    try:
        ret = model(x, torch.tensor(lens), logmel, midi, lft)
    finally:
        random.uniform, random.randint, torch.rand, torch.randn_like = o_uniform, o_randint, o_rand, o_randn_like
    (ret["cfm_loss"] + ret["prior_loss"]).backward()
    g = {k: p.grad for k, p in model.named_parameters()}
    assert all(v is not None for v in g.values()), [k for k, v in g.items() if v is None]
    total = torch.sqrt(sum((v.double() ** 2).sum() for v in g.values()))
    sd = model.state_dict()
    out = dict(lens=np.array(lens), x=x, logmel=logmel, midi=midi, lft=lft, uniform=np.float64(draws["uniform"]),
               seg_start=np.int64(draws["seg_start"]), t=draws["t"], z=draws["z"], cfm_loss=ret["cfm_loss"],
               prior_loss=ret["prior_loss"], gauss_mel=ret["gauss_mel"], grad_norm=total, n_params=np.int64(len(g)),
               bn_mean=sd["gst.ref_enc.convs.16.running_mean"], bn_var=sd["gst.ref_enc.convs.16.running_var"])
    for k in FULL_WHOLE:
        out["g:" + k] = g[k]
    for k in FULL_ROWS:
        out["r:" + k] = g[k][:4]
    npz("train_full_T64", **out)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "train_full":
    train_full_case()
    sys.exit(0)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "train":
    train_grads_case()
    sys.exit(0)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "euler20":
    euler20_case()
    sys.exit(0)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "forward":
    forward_case()
    sys.exit(0)


if __name__ == "__main__":
    main()
