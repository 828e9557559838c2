"""SiFiGAN (source-filter HiFi-GAN) generator on the HIP kernels — row a9, **parity unpinned**.

The reference post-processes HiFi-GAN's output with `sifigan.models.SiFiGANGenerator(in_signal, c, dfs)[0]`
(serenade/bin/ssc_postprocessing.py:33,92-99,219-227; hyper-parameters
serenade/bin/sifigan_config/generator/sifigan.yaml:1-29), but `sifigan` is an un-vendored, unpinned dependency
whose source is not in the reference tree: this module follows the published architecture (see
oracle/sifigan_oracle.py for the restatement it is checked against) and keeps the call signature
``forward(x, c, d) -> (waveform, excitation)`` with (B, C, T) tensors.  Equality with the upstream package's
arithmetic or state_dict layout cannot be verified here.

Mapping to kernels: every Conv1d / ConvTranspose1d phase / 1x1 conv is `srn_conv_gemm`; the pitch-dependent
dilated conv gathers [x | x(t - r) | x(t + r)] rows with `srn_pd_gather` and multiplies them by [Wc | Wp | Wf] in ONE
GEMM (K = 3C); the strided down-sampling convs use in_stride; LeakyReLUs ride in GEMM prologues / epilogues.
"""
from collections import OrderedDict

import torch

from . import _shapes, ops
from .models import _Packed, _dev_f32, _fold_wn, _require_cuda, _rup
from .ops import ACT_LEAKY, POST_DIV, POST_LEAKY, RES_ADD, ConvOp

DEFAULT_PARAMS = dict(
    in_channels=43, out_channels=1, channels=512, kernel_size=7, upsample_scales=(5, 4, 3, 2),
    upsample_kernel_sizes=(10, 8, 6, 4),
    source_network_params=dict(resblock_kernel_size=3, resblock_dilations=[(1,), (1, 2), (1, 2, 4), (1, 2, 4, 8)],
                               use_additional_convs=True),
    filter_network_params=dict(resblock_kernel_sizes=(3, 5, 7), resblock_dilations=[(1, 3, 5)] * 3,
                               use_additional_convs=False),
    share_upsamples=False, share_downsamples=False, bias=True, nonlinear_activation="LeakyReLU",
    nonlinear_activation_params={"negative_slope": 0.1}, use_weight_norm=True)


def sifigan_shapes(in_channels=43, out_channels=1, channels=512, kernel_size=7, upsample_scales=(5, 4, 3, 2),
                   upsample_kernel_sizes=(10, 8, 6, 4), source_network_params=None, filter_network_params=None,
                   share_upsamples=False, share_downsamples=False, use_weight_norm=True, **_):
    sp = source_network_params or DEFAULT_PARAMS["source_network_params"]
    fp = filter_network_params or DEFAULT_PARAMS["filter_network_params"]
    wn = use_weight_norm
    d = OrderedDict()
    n = len(upsample_scales)
    _shapes._conv(d, "input_conv", channels, in_channels, kernel_size, wn)
    nets = ["sn"] if share_upsamples else ["sn", "fn"]
    for i in range(n):
        ci, co = channels // 2 ** i, channels // 2 ** (i + 1)
        for net in nets:
            _shapes._conv(d, f"{net}.upsamples.{i}.1", co, ci, upsample_kernel_sizes[i], wn, transpose=True)
        for j in range(len(sp["resblock_dilations"][i])):
            for nm in ("convsC", "convsP", "convsF"):
                _shapes._conv(d, f"sn.blocks.{i}.{nm}.{j}", co, co, 1, wn)
            if sp["use_additional_convs"]:
                _shapes._conv(d, f"sn.blocks.{i}.convsA.{j}.1", co, co, sp["resblock_kernel_size"], wn)
        nb = len(fp["resblock_kernel_sizes"])
        for j in range(nb):
            for idx in range(len(fp["resblock_dilations"][j])):
                _shapes._conv(d, f"fn.blocks.{i * nb + j}.convs1.{idx}.1", co, co, fp["resblock_kernel_sizes"][j], wn)
    cl = channels // 2 ** n
    _shapes._conv(d, "sn.output_conv", out_channels, cl, kernel_size, wn)
    _shapes._conv(d, "fn.output_conv.1", out_channels, cl, kernel_size, wn)
    _shapes._conv(d, "sn.emb", cl, 1, kernel_size, wn)
    for net in (["sn"] if share_downsamples else ["sn", "fn"]):
        for j, i in enumerate(reversed(range(1, n))):
            _shapes._conv(d, f"{net}.downsamples.{j}.0", channels // 2 ** i, channels // 2 ** (i + 1),
                          upsample_kernel_sizes[i], wn)
    return d


class SiFiGANGenerator(_Packed):
    def __init__(self, **params):
        cfg = dict(DEFAULT_PARAMS)
        cfg.update(params)
        assert cfg["out_channels"] == 1 and cfg["bias"] and cfg["nonlinear_activation"] == "LeakyReLU"
        assert cfg["source_network_params"]["resblock_kernel_size"] == 3
        for s, k in zip(cfg["upsample_scales"], cfg["upsample_kernel_sizes"]):
            assert k == 2 * s
        super().__init__(sifigan_shapes(**cfg))
        self.cfg = cfg
        self.slope = float(cfg["nonlinear_activation_params"].get("negative_slope", 0.1))
        self.hop = 1
        for s in cfg["upsample_scales"]:
            self.hop *= s

    def remove_weight_norm(self):
        def walk(m):
            if "weight_g" in m._parameters:
                g, v = m._parameters["weight_g"], m._parameters["weight_v"]
                w = (v * (g / v.reshape(v.shape[0], -1).norm(dim=1).reshape(g.shape))).detach()
                del m._parameters["weight_g"], m._parameters["weight_v"]
                m.register_parameter("weight", torch.nn.Parameter(w, requires_grad=False))
            for c in m.children():
                walk(c)

        walk(self)
        self._invalidate()

    def packed(self):
        if self._packed is not None:
            return self._packed
        cfg, dev = self.cfg, self._device()
        sd = {k: _dev_f32(v, dev) for k, v in self._own_state().items()}
        n = len(cfg["upsample_scales"])
        sp, fp = cfg["source_network_params"], cfg["filter_network_params"]
        P = {}
        cin_p = _rup(cfg["in_channels"], 4)
        P["cin_p"] = cin_p
        P["in_w"] = ops.pack_conv_weight(_fold_wn(sd, "input_conv"), cin_p)
        P["in_b"] = sd["input_conv.bias"]
        P["emb_w"] = ops.pack_conv_weight(_fold_wn(sd, "sn.emb"), 4)
        P["emb_b"] = sd["sn.emb.bias"]
        for net in ("sn", "fn"):
            src = "sn" if (net == "fn" and cfg["share_upsamples"]) else net
            P[net + "_up"] = []
            for i, s in enumerate(cfg["upsample_scales"]):
                w = _fold_wn(sd, f"{src}.upsamples.{i}.1")
                P[net + "_up"].append(dict(phases=ops.convtranspose_phases(w, s, s // 2 + s % 2),
                                           b=sd[f"{src}.upsamples.{i}.1.bias"], s=s, cout=w.shape[1]))
            src = "sn" if (net == "fn" and cfg["share_downsamples"]) else net
            P[net + "_down"] = []
            for j, i in enumerate(reversed(range(1, n))):
                s, k = cfg["upsample_scales"][i], cfg["upsample_kernel_sizes"][i]
                pad = s - (1 if k % 2 == 0 else 0)
                w = _fold_wn(sd, f"{src}.downsamples.{j}.0")
                P[net + "_down"].append(dict(w=ops.pack_conv_weight(w), b=sd[f"{src}.downsamples.{j}.0.bias"], s=s,
                                             taps=[t - pad for t in range(k)], cout=w.shape[0], cin=w.shape[1]))
        P["sn_blocks"] = []
        for i in range(n):
            blk = []
            for j, dil in enumerate(sp["resblock_dilations"][i]):
                p = f"sn.blocks.{i}."
                ws = [_fold_wn(sd, p + f"{nm}.{j}")[:, :, 0] for nm in ("convsC", "convsP", "convsF")]
                b = sd[p + f"convsC.{j}.bias"] + sd[p + f"convsP.{j}.bias"] + sd[p + f"convsF.{j}.bias"]
                e = dict(dil=dil, w3=torch.cat(ws, dim=1).contiguous(), b3=b.contiguous())
                if sp["use_additional_convs"]:
                    e.update(wa=ops.pack_conv_weight(_fold_wn(sd, p + f"convsA.{j}.1")), ba=sd[p + f"convsA.{j}.1.bias"])
                blk.append(e)
            P["sn_blocks"].append(blk)
        nb = len(fp["resblock_kernel_sizes"])
        P["fn_blocks"] = []
        for i in range(n):
            for j in range(nb):
                k = fp["resblock_kernel_sizes"][j]
                P["fn_blocks"].append([dict(k=k, d=dl, w=ops.pack_conv_weight(
                    _fold_wn(sd, f"fn.blocks.{i * nb + j}.convs1.{idx}.1")),
                    b=sd[f"fn.blocks.{i * nb + j}.convs1.{idx}.1.bias"])
                    for idx, dl in enumerate(fp["resblock_dilations"][j])])
        P["sn_out_w"] = ops.pack_conv_weight(_fold_wn(sd, "sn.output_conv"))
        P["sn_out_b"] = sd["sn.output_conv.bias"]
        P["fn_out_w"] = _fold_wn(sd, "fn.output_conv.1")[0].t().contiguous()  # (k, C)
        P["fn_out_b"] = sd["fn.output_conv.1.bias"]
        self._packed = P
        return P

    def plan(self, B, T):
        key = (B, T, ops.DEFAULT_PRECISION)
        if key not in self._plans:
            self._plans.clear()
            self._plans[key] = SiFiGANPlan(self, B, T)
        return self._plans[key]

    @torch.no_grad()
    def forward(self, x, c, d):
        """x (B, 1, T*hop) sine excitation, c (B, in_channels, T), d: list of (B, 1, T*cumprod(scales)[i]) dilation
        factors -> (waveform (B, 1, T*hop), excitation (B, 1, T*hop))."""
        _require_cuda(c, "SiFiGANGenerator.forward")
        B, _, T = c.shape
        pl = self.plan(B, T)
        pl.load(x, c, d)
        pl.run()
        return pl.wave.clone().unsqueeze(1), pl.exc.clone().view(B, 1, -1)


class SiFiGANPlan:
    def __init__(self, gen, B, T):
        P, cfg = gen.packed(), gen.cfg
        dev = gen._device()
        slope = gen.slope
        f = lambda *s: torch.zeros(*s, device=dev, dtype=torch.float32)
        scales = cfg["upsample_scales"]
        n = len(scales)
        C0 = cfg["channels"]
        Rs, Cs = [], []
        r = T
        for i, s in enumerate(scales):
            r *= s
            Rs.append(r)
            Cs.append(C0 // 2 ** (i + 1))
        Rf, Cl = Rs[-1], Cs[-1]
        self.B, self.T, self.Rs = B, T, Rs
        ks = cfg["kernel_size"]
        cin, cin_p = cfg["in_channels"], P["cin_p"]
        self._c_in = f(B, cin, T)
        self._x_in = f(B, Rf)
        self.d = [f(B, Rs[i]) for i in range(n)]
        c_cl = f(B, T, cin_p)
        x_cl = f(B, Rf, 4)
        big = max(a * b for a, b in zip(Rs, Cs))
        ol = [ops.transpose_op(self._c_in, c_cl, B, cin, T, cin * T, T, T * cin_p, cin_p),
              ops.copy_channels_op(self._x_in, Rf, 1, 0, x_cl, Rf * 4, 4, 0, B, Rf, 1)]

        def conv(inp, ci, T_in, w, b, out, co, T_out, taps, **kw):
            return ConvOp(in0=inp, w=w, out=out, n_batch=B, T_in=T_in, T_out=T_out, C_in=ci, N=co, in0_bs=T_in * ci,
                          ld_in0=ci, ldw=w.shape[1], out_bs=kw.pop("out_bs", T_out * co), ld_out=co, bias=b, taps=taps,
                          **kw)

        h = f(B, T, C0)
        ol.append(conv(c_cl, cin_p, T, P["in_w"], P["in_b"], h, C0, T, ops.conv_taps(ks)))

        def down(net, first):
            """first: (B, Rf, Cl) feature at the final rate -> list of features down to the first stage's rate"""
            feats = [first]
            cur, rc = first, Rf
            for dn in P[net + "_down"]:
                ro = rc // dn["s"]
                o = f(B, ro, dn["cout"])
                ol.append(conv(cur, dn["cin"], rc, dn["w"], dn["b"], o, dn["cout"], ro, dn["taps"], in_stride=dn["s"],
                               post=POST_LEAKY, post_div=slope))
                feats.append(o)
                cur, rc = o, ro
            return feats

        def up(net, i, src, csrc, rsrc, emb, out):
            u = P[net + "_up"][i]
            s, co, ro = u["s"], u["cout"], rsrc * u["s"]
            for ph, (taps, wp) in enumerate(u["phases"]):
                ol.append(conv(src, csrc, rsrc, wp, u["b"], out, co, rsrc, taps, pro_act=ACT_LEAKY, pro_slope=slope,
                               out_bs=ro * co, out_t_stride=s, out_t_off=ph, res=emb, res_mode=RES_ADD,
                               res_bs=ro * co, ld_res=co))

        # ---- source network
        emb0 = f(B, Rf, Cl)
        ol.append(conv(x_cl, 4, Rf, P["emb_w"], P["emb_b"], emb0, Cl, Rf, ops.conv_taps(ks)))
        embs = down("sn", emb0)
        e_a, e_b, g3, xt = f(B * big), f(B * big), f(B * 3 * big), f(B * big)
        cur, ccur, rcur = h, C0, T
        for i in range(n):
            C, R = Cs[i], Rs[i]
            x_, y_ = (e_a, e_b) if cur is not e_a else (e_b, e_a)  # never up-sample a buffer onto itself
            up("sn", i, cur, ccur, rcur, embs[-i - 1], x_)
            for blk in P["sn_blocks"][i]:
                ol.append(ops.pd_gather_op(x_, self.d[i], g3, B, R, C, blk["dil"], slope))
                if "wa" in blk:
                    ol.append(conv(g3, 3 * C, R, blk["w3"], blk["b3"], xt, C, R, [0]))
                    ol.append(conv(xt, C, R, blk["wa"], blk["ba"], y_, C, R, ops.conv_taps(3), pro_act=ACT_LEAKY,
                                   pro_slope=slope, res=x_, res_mode=RES_ADD, res_bs=R * C, ld_res=C))
                else:
                    ol.append(conv(g3, 3 * C, R, blk["w3"], blk["b3"], y_, C, R, [0], res=x_, res_mode=RES_ADD,
                                   res_bs=R * C, ld_res=C))
                x_, y_ = y_, x_
            cur, ccur, rcur = x_, C, R
        e_fin = f(B, Rf, Cl)
        ol.append(ops.copy_channels_op(cur, Rf * Cl, Cl, 0, e_fin, Rf * Cl, Cl, 0, B, Rf, Cl))
        self.exc = f(B, Rf)
        ol.append(conv(e_fin, Cl, Rf, P["sn_out_w"], P["sn_out_b"], self.exc, 1, Rf, ops.conv_taps(ks)))
        # ---- filter network
        fembs = down("fn", e_fin)
        u_, p0, p1, acc = f(B * big), f(B * big), f(B * big), f(B * big)
        fp = cfg["filter_network_params"]
        nb = len(fp["resblock_kernel_sizes"])
        cur, ccur, rcur = h, C0, T
        for i in range(n):
            C, R = Cs[i], Rs[i]
            up("fn", i, cur, ccur, rcur, fembs[-i - 1], u_)
            for j in range(nb):
                convs = P["fn_blocks"][i * nb + j]
                x_ = u_
                pp = [p0, p1]
                for idx, cv in enumerate(convs):
                    last = idx == len(convs) - 1
                    dst = acc if last else pp[idx % 2]
                    fin = {}
                    if last:
                        if j > 0:
                            fin.update(res2=acc, res2_bs=R * C, ld_res2=C)
                        if j == nb - 1:
                            fin.update(post=POST_DIV, post_div=float(nb))
                    ol.append(conv(x_, C, R, cv["w"], cv["b"], dst, C, R, ops.conv_taps(cv["k"], cv["d"]),
                                   pro_act=ACT_LEAKY, pro_slope=slope, res=x_, res_mode=RES_ADD, res_bs=R * C,
                                   ld_res=C, **fin))
                    x_ = dst
            cur, ccur, rcur = acc, C, R
        self.wave = f(B, Rf)
        ol.append(ops.out_conv_tanh_op(cur, P["fn_out_w"], P["fn_out_b"], self.wave, B, Rf, Cl, ks, slope))
        self.ops = ol

    def load(self, x, c, d):
        self._c_in.copy_(c)
        self._x_in.copy_(x.reshape(self.B, -1))
        for buf, di in zip(self.d, d):
            buf.copy_(di.reshape(self.B, -1))

    def run(self):
        for op in self.ops:
            op()
