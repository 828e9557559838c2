"""HIP-backed mirror of ``serenade.vocoder``: HiFiGANGenerator, Vocoder, load_vocoder.

    HiFiGANGenerator   serenade/vocoder/models/hifigan.py:21-284 (+ HiFiGANResidualBlock,
                       serenade/vocoder/layers/residual_block.py:143-258)
    Vocoder            serenade/vocoder/vocoder.py:11-75
    load_vocoder       serenade/vocoder/utils.py:14-63

Same constructor kwargs, method signatures and ``state_dict`` layout (with or without weight
norm), so a parallel_wavegan-style checkpoint ``["model"]["generator"]`` loads unchanged.
Everything is computed channels-last by libserenade_hip.so: the transposed convolutions run as one
implicit-GEMM launch per output phase (2 taps each), every residual-stack conv has the LeakyReLU
fused into its input gather and bias / residual / stage-mean fused into its epilogue.
"""
import logging
import os
import time

import torch
import yaml

from . import _shapes, ops
from .models import _Packed, _dev_f32, _fold_wn, _lru_get, _require_cuda
from .ops import ACT_LEAKY, POST_DIV, POST_LEAKY, RES_ADD, ConvOp

__all__ = ["HiFiGANGenerator", "Vocoder", "load_vocoder", "hifigan_state_shapes"]


def hifigan_state_shapes(**params):
    return _shapes.as_meta(_shapes.hifigan_shapes(**params))


class HiFiGANGenerator(_Packed):
    def __init__(self, in_channels=80, out_channels=1, channels=512, kernel_size=7, upsample_scales=(8, 8, 2, 2),
                 upsample_kernel_sizes=(16, 16, 4, 4), resblock_kernel_sizes=(3, 7, 11),
                 resblock_dilations=[(1, 3, 5), (1, 3, 5), (1, 3, 5)], use_additional_convs=True, bias=True,
                 nonlinear_activation="LeakyReLU", nonlinear_activation_params={"negative_slope": 0.1},
                 use_causal_conv=False, use_weight_norm=True):
        assert kernel_size % 2 == 1, "Kernel size must be odd number."
        assert len(upsample_scales) == len(upsample_kernel_sizes)
        assert len(resblock_dilations) == len(resblock_kernel_sizes)
        assert not use_causal_conv, "causal HiFi-GAN is not reachable from the decode path (SURVEY section 2)"
        assert bias and out_channels == 1 and nonlinear_activation == "LeakyReLU"
        for s, k in zip(upsample_scales, upsample_kernel_sizes):
            assert k == 2 * s
        super().__init__(_shapes.hifigan_shapes(in_channels, out_channels, channels, kernel_size, upsample_scales,
                                                upsample_kernel_sizes, resblock_kernel_sizes,
                                                [tuple(d) for d in resblock_dilations], use_additional_convs,
                                                weight_norm=use_weight_norm))
        self.in_channels, self.channels, self.kernel_size = in_channels, channels, kernel_size
        self.upsample_scales = tuple(upsample_scales)
        self.upsample_kernel_sizes = tuple(upsample_kernel_sizes)
        self.resblock_kernel_sizes = tuple(resblock_kernel_sizes)
        self.resblock_dilations = [tuple(d) for d in resblock_dilations]
        self.use_additional_convs = use_additional_convs
        self.slope = float(nonlinear_activation_params.get("negative_slope", 0.01))
        self.num_upsamples = len(upsample_kernel_sizes)
        self.num_blocks = len(resblock_kernel_sizes)
        self.hop = 1
        for s in self.upsample_scales:
            self.hop *= s

    # ------------------------------------------------------------------ reference API
    def remove_weight_norm(self):
        """hifigan.py:206-217: replace (weight_g, weight_v) by the folded weight."""

        def walk(m):
            if "weight_g" in m._parameters:
                g, v = m._parameters["weight_g"], m._parameters["weight_v"]
                norm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(g.shape)
                w = (v * (g / norm)).detach()
                del m._parameters["weight_g"], m._parameters["weight_v"]
                m.register_parameter("weight", torch.nn.Parameter(w, requires_grad=False))
            for c in m.children():
                walk(c)

        walk(self)
        self._invalidate()

    def apply_weight_norm(self):
        raise NotImplementedError("training-only (hifigan.py:219-230)")

    def register_stats(self, stats):
        """hifigan.py:232-247 (".npy" natively; ".h5" through utils.io.read_hdf5)."""
        assert stats.endswith(".h5") or stats.endswith(".npy")
        import numpy as np
        if stats.endswith(".h5"):
            from .utils.io import read_hdf5
            mean, scale = read_hdf5(stats, "mean").reshape(-1), read_hdf5(stats, "scale").reshape(-1)
        else:
            mean, scale = np.load(stats)[0].reshape(-1), np.load(stats)[1].reshape(-1)
        self.register_buffer("mean", torch.from_numpy(mean).float())
        self.register_buffer("scale", torch.from_numpy(scale).float())

    # ------------------------------------------------------------------ packing / plan
    def packed(self):
        if self._packed is None:
            dev = self._device()
            sd = {k: _dev_f32(v, dev) for k, v in self._own_state().items() if k not in ("mean", "scale")}
            P = dict(in_w=ops.pack_conv_weight(_fold_wn(sd, "input_conv")), in_b=sd["input_conv.bias"], ups=[],
                     blocks=[])
            for i, s in enumerate(self.upsample_scales):
                w = _fold_wn(sd, f"upsamples.{i}.1")
                P["ups"].append(dict(phases=ops.convtranspose_phases(w, s, s // 2 + s % 2),
                                     b=sd[f"upsamples.{i}.1.bias"], cin=w.shape[0], cout=w.shape[1], s=s))
            for bi in range(self.num_upsamples * self.num_blocks):
                j = bi % self.num_blocks
                k, dil = self.resblock_kernel_sizes[j], self.resblock_dilations[j]
                convs = []
                for idx, d in enumerate(dil):
                    p1 = f"blocks.{bi}.convs1.{idx}.1"
                    c = dict(d=d, k=k, w1=ops.pack_conv_weight(_fold_wn(sd, p1)), b1=sd[p1 + ".bias"])
                    if self.use_additional_convs:
                        p2 = f"blocks.{bi}.convs2.{idx}.1"
                        c.update(w2=ops.pack_conv_weight(_fold_wn(sd, p2)), b2=sd[p2 + ".bias"])
                    convs.append(c)
                P["blocks"].append(convs)
            wo = _fold_wn(sd, "output_conv.1")  # (1, C, k)
            P["out_w"] = wo[0].t().contiguous()  # (k, C)
            P["out_b"] = sd["output_conv.1.bias"]
            self._packed = P
        return self._packed

    def plan(self, B, T):
        key = (B, T, ops.DEFAULT_PRECISION)
        return _lru_get(self._plans, key, 4, lambda: HiFiGANPlan(self, B, T))

    @torch.no_grad()
    def forward(self, c):
        """(B, in_channels, T) -> (B, 1, T * hop)   (hifigan.py:171-190)."""
        _require_cuda(c, "HiFiGANGenerator.forward")
        return self._run_cl(c.detach().to(torch.float32).transpose(1, 2).contiguous())

    def _run_cl(self, c_cl):
        B, T, _ = c_cl.shape
        pl = self.plan(B, T)
        pl.c_in.copy_(c_cl)
        pl.run()
        return pl.wave.clone().unsqueeze(1)

    @torch.no_grad()
    def inference(self, c, normalize_before=False):
        """(T, in_channels) -> (T * hop, 1)   (hifigan.py:249-265)."""
        if not isinstance(c, torch.Tensor):
            c = torch.tensor(c, dtype=torch.float).to(self._device())
        _require_cuda(c, "HiFiGANGenerator.inference")
        if normalize_before:
            c = (c - self.mean) / self.scale
        y = self._run_cl(c.detach().to(torch.float32).unsqueeze(0).contiguous())
        return y.squeeze(0).transpose(1, 0)

    @torch.no_grad()
    def inference_batch(self, c, normalize_before=False):
        """(B, T, in_channels) -> (B, 1, T * hop)   (hifigan.py:267-284)."""
        if not isinstance(c, torch.Tensor):
            c = torch.tensor(c, dtype=torch.float).to(self._device())
        _require_cuda(c, "HiFiGANGenerator.inference_batch")
        if normalize_before:
            c = (c - self.mean) / self.scale
        return self._run_cl(c.detach().to(torch.float32).contiguous())


FUSED_UNIT_CHANNELS = (32, 64)  # stage widths whose residual units run as one fused launch (srn_hifigan_resunit)


class HiFiGANPlan:
    """Buffers + op list of one generator forward for (B, T).  Input: self.c_in (B, T, in_ch) channels-last;
    output: self.wave (B, T * hop)."""

    def __init__(self, gen, B, T):
        P = gen.packed()
        dev = gen._device()
        self.B, self.T = B, T
        f = lambda *s: torch.zeros(*s, device=dev, dtype=torch.float32)
        self.c_in = f(B, T, gen.in_channels)
        C0 = gen.channels
        h = f(B, T, C0)
        # widest (T_i * C_i) stage decides the ping-pong buffer size
        Ts, Cs = [], []
        t, c = T, C0
        for up in P["ups"]:
            t, c = t * up["s"], up["cout"]
            Ts.append(t)
            Cs.append(c)
        big = max(a * b for a, b in zip(Ts, Cs))
        u, p0, p1, xt, acc = (f(B * big) for _ in range(5))
        self.wave = f(B, Ts[-1])
        slope = gen.slope
        ol = []

        def conv(inp, cin, T_in, w, b, out, cout, T_out, taps, **kw):
            return ConvOp(in0=inp, w=w, out=out, n_batch=B, T_in=T_in, T_out=T_out, C_in=cin, N=cout,
                          in0_bs=T_in * cin, ld_in0=cin, ldw=w.shape[1], out_bs=kw.pop("out_bs", T_out * cout),
                          ld_out=cout, bias=b, taps=taps, **kw)

        ol.append(conv(self.c_in, gen.in_channels, T, P["in_w"], P["in_b"], h, C0, T, ops.conv_taps(gen.kernel_size)))
        cur, ccur, tcur = h, C0, T
        nb = gen.num_blocks
        for i, up in enumerate(P["ups"]):
            s, C, Tn = up["s"], up["cout"], Ts[i]
            for r, (taps, wp) in enumerate(up["phases"]):
                ol.append(conv(cur, ccur, tcur, wp, up["b"], u, C, tcur, taps, pro_act=ACT_LEAKY, pro_slope=slope,
                               out_bs=Tn * C, out_t_stride=s, out_t_off=r))
            for j in range(nb):
                convs = P["blocks"][i * nb + j]
                x = u
                pp = [p0, p1]
                for idx, cv in enumerate(convs):
                    last = idx == len(convs) - 1
                    dst = acc if last else pp[idx % 2]
                    fin = {}
                    if last:
                        # cs += block(c); c = cs / num_blocks  (hifigan.py:183-186), in the reference's order
                        if j > 0:
                            fin.update(res2=acc, res2_bs=Tn * C, ld_res2=C)
                        if j == nb - 1:
                            fin.update(post=POST_DIV, post_div=float(nb))
                    if "w2" in cv and C in FUSED_UNIT_CHANNELS and (cv["k"] - 1) * cv["d"] <= 50 and cv["k"] % 2 == 1:
                        # thin stage: the whole unit in one launch (resunit.hip); stage sum / mean in its epilogue
                        ol.append(ops.ResUnitOp(x=x, w1=cv["w1"], b1=cv["b1"], w2=cv["w2"], b2=cv["b2"], out=dst,
                                                n_batch=B, T=Tn, C=C, k=cv["k"], dilation=cv["d"], slope=slope,
                                                res2=fin.get("res2"), post_div=fin.get("post_div", 0.0)))
                    elif "w2" in cv:
                        # the second LeakyReLU (residual_block.py:252) has one consumer: it runs once per element in
                        # conv1's epilogue, not once per element per 32-deep step per column tile in conv2's loop (on
                        # the fp32 matrix pipe every vector instruction of the loop is taken from the MFMA stream)
                        ol.append(conv(x, C, Tn, cv["w1"], cv["b1"], xt, C, Tn, ops.conv_taps(cv["k"], cv["d"]),
                                       pro_act=ACT_LEAKY, pro_slope=slope, post=POST_LEAKY, post_div=slope))
                        ol.append(conv(xt, C, Tn, cv["w2"], cv["b2"], dst, C, Tn, ops.conv_taps(cv["k"], 1),
                                       res=x, res_mode=RES_ADD, res_bs=Tn * C, ld_res=C, **fin))
                    else:
                        ol.append(conv(x, C, Tn, cv["w1"], cv["b1"], dst, C, Tn, ops.conv_taps(cv["k"], cv["d"]),
                                       pro_act=ACT_LEAKY, pro_slope=slope, res=x, res_mode=RES_ADD, res_bs=Tn * C,
                                       ld_res=C, **fin))
                    x = dst
            cur, ccur, tcur = acc, C, Tn
            # the next stage's upsample reads `acc` and writes `u`; resblocks then overwrite acc only at their end
        ol.append(ops.out_conv_tanh_op(cur, P["out_w"], P["out_b"], self.wave, B, tcur, ccur, gen.kernel_size, 0.01))
        self.ops = ol
        self._runner = ops.GraphRunner(lambda: self.ops)
        self._keep = (h, u, p0, p1, xt, acc)

    def run(self):
        self._runner()


def load_vocoder(checkpoint, config=None, stats=None):
    """serenade/vocoder/utils.py:14-63."""
    if config is None:
        with open(os.path.join(os.path.dirname(checkpoint), "config.yml")) as f:
            config = yaml.load(f, Loader=yaml.Loader)
    generator_type = config.get("generator_type", "HiFiGANGenerator")
    generator_params = {k.replace("upsample_kernal_sizes", "upsample_kernel_sizes"): v
                        for k, v in config["generator_params"].items()}
    model = HiFiGANGenerator(**generator_params)
    model.load_state_dict(torch.load(checkpoint, map_location="cpu")["model"]["generator"])
    if stats is None:
        ext = "h5" if config["format"] == "hdf5" else "npy"
        cand = os.path.join(os.path.dirname(checkpoint), f"stats.{ext}")
        if os.path.exists(cand):
            stats = cand
    if stats is not None and generator_type != "VQVAE":
        model.register_stats(stats)
    return model


class Vocoder(object):
    """serenade/vocoder/vocoder.py:11-75."""

    def __init__(self, checkpoint, config, stats, device, trg_stats=None, take_norm_feat=True):
        with open(config) as f:
            cfg = yaml.load(f, Loader=yaml.Loader)
        model = load_vocoder(checkpoint, cfg)
        logging.info(f"Loaded model parameters from {checkpoint}.")
        from .utils.io import read_feats  # .h5 (h5py) like the reference, or .npz with the same keys
        st = {"mean": read_feats(stats, "mean"), "scale": read_feats(stats, "scale")}
        self._setup(model, cfg, st, device, trg_stats, take_norm_feat)

    @classmethod
    def from_generator(cls, model, config, stats, device, trg_stats=None, take_norm_feat=True):
        """Build from an in-memory generator and stats dict (synthetic runs: no checkpoint files exist offline)."""
        self = cls.__new__(cls)
        self._setup(model, config, stats, device, trg_stats, take_norm_feat)
        return self

    def _setup(self, model, config, stats, device, trg_stats, take_norm_feat):
        self.device = device
        if take_norm_feat:
            assert trg_stats is not None, "trg_stats must be given if take_norm_feat=True"
            self.trg_stats = {k: torch.tensor(trg_stats[k], dtype=torch.float).to(device).contiguous()
                              for k in ("mean", "scale")}
        self.take_norm_feat = take_norm_feat
        self.config = config
        self.model = model
        self.model.remove_weight_norm()
        self.model = self.model.eval().to(device)
        self.stats = {k: torch.tensor(stats[k], dtype=torch.float).to(device).contiguous()
                      for k in ("mean", "scale")}

    def _decode_cl(self, c):
        """c (B, T, 80) normalised mel -> (B, T * hop): both affine maps and the generator on the GPU."""
        _require_cuda(c, "Vocoder.decode")
        B, T, C = c.shape
        pl = self.model.plan(B, T)
        ts = self.trg_stats if self.take_norm_feat else {"scale": None, "mean": None}
        ops.renorm_op(c.detach().to(torch.float32).contiguous(), ts["scale"], ts["mean"], self.stats["mean"],
                      self.stats["scale"], pl.c_in, B * T, C)()
        pl.run()
        return pl.wave.clone()

    @torch.no_grad()
    def decode(self, c):
        """(T, 80) -> ((T * hop,), sampling_rate)   (vocoder.py:51-62)."""
        start = time.time()
        y = self._decode_cl(c.unsqueeze(0)).view(-1)
        rtf = (time.time() - start) / (len(y) / self.config["sampling_rate"])
        logging.info(f"Finished waveform generation. (RTF = {rtf:.03f}).")
        return y, self.config["sampling_rate"]

    @torch.no_grad()
    def decode_batch(self, c):
        """(B, T, 80) -> (B, T * hop)   (vocoder.py:64-75)."""
        return self._decode_cl(c)
