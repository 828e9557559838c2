"""CPU ORACLE for row a9 (SiFiGAN V2 generator) — test infrastructure, NOT product code.

**PARITY UNPINNED.**  The reference calls `sifigan.models.SiFiGANGenerator(in_signal, c, dfs)[0]`
(serenade/bin/ssc_postprocessing.py:92-99,219-227) but the `sifigan` package is an un-vendored, unpinned
dependency (`pip install git+https://github.com/chomeyama/SiFiGAN@main`, README.md:22): its source is not under
/root/reference, is not installed in the container, and the reference holds no tests or golden vectors for it.
This file restates the *published* algorithm (Yoneyama et al., "Source-Filter HiFi-GAN", ICASSP 2023, and the
public SiFiGAN code as remembered) under the hyper-parameters of the reference's own config
`serenade/bin/sifigan_config/generator/sifigan.yaml:1-29`; it can only serve as the checker of this repo's HIP
implementation (self-consistency), not as proof of equality with the upstream package.

Structure restated (state_dict names follow the upstream module tree as far as it is known):
  input_conv: Conv1d(43 -> C, k7)
  source network `sn`: emb Conv1d(1 -> C/16, k7) on the sine; downsamples[i] = Conv1d(k = 2s, stride s, pad s-1) +
      LeakyReLU walking the sine embedding down to every stage's rate; per stage: LeakyReLU + ConvTranspose1d,
      + the embedding of that rate, then an AdaptiveResidualBlock whose "pitch-dependent dilated conv" reads the
      past / future sample at distance round(d[t] * dilation):  x <- convA(lrelu(convC(lrelu x) + convP(past) +
      convF(future))) + x;  output_conv Conv1d(C/16 -> 1, k7) gives the excitation.
  filter network `fn`: downsamples of the last source feature; per stage LeakyReLU + ConvTranspose1d + that
      feature, mean of three HiFi-GAN residual blocks (k 3/5/7, dilations 1/3/5, no additional convs);
      output_conv = LeakyReLU -> Conv1d(k7) -> tanh.
  forward(x, c, d) -> (waveform, excitation).
"""
import torch
import torch.nn.functional as F

DEFAULT_CFG = dict(
    in_channels=43, out_channels=1, channels=512, kernel_size=7, upsample_scales=(5, 4, 3, 2),
    upsample_kernel_sizes=(10, 8, 6, 4),
    source_network_params=dict(resblock_kernel_size=3, resblock_dilations=[(1,), (1, 2), (1, 2, 4), (1, 2, 4, 8)],
                               use_additional_convs=True),
    filter_network_params=dict(resblock_kernel_sizes=(3, 5, 7), resblock_dilations=[(1, 3, 5)] * 3,
                               use_additional_convs=False),
    share_upsamples=False, share_downsamples=False, bias=True, nonlinear_activation="LeakyReLU",
    nonlinear_activation_params={"negative_slope": 0.1}, use_weight_norm=True)


def _w(w, name):
    if name + ".weight" in w:
        return w[name + ".weight"]
    g, v = w[name + ".weight_g"], w[name + ".weight_v"]
    return v * (g / v.reshape(v.shape[0], -1).norm(dim=1).reshape(g.shape))


def pd_indexing(x, d, dilation):
    """past / future samples at the pitch-dependent distance r[t] = round_half_even(d[t] * dilation);
    zero outside the signal.  x (B, C, T), d (B, 1, T)."""
    B, C, T = x.shape
    r = torch.round(d[:, 0, :] * dilation).long()
    t = torch.arange(T).unsqueeze(0)
    ip, jf = t - r, t + r
    xp = torch.gather(x, 2, ip.clamp(0, T - 1).unsqueeze(1).expand(B, C, T)) * ((ip >= 0) & (ip < T)).unsqueeze(1)
    xf = torch.gather(x, 2, jf.clamp(0, T - 1).unsqueeze(1).expand(B, C, T)) * ((jf >= 0) & (jf < T)).unsqueeze(1)
    return xp, xf


def adaptive_residual_block(w, x, d, dilations, slope, use_additional_convs=True):
    for i, dil in enumerate(dilations):
        xt = F.leaky_relu(x, slope)
        xp, xf = pd_indexing(xt, d, dil)
        xt = (F.conv1d(xt, _w(w, f"convsC.{i}"), w[f"convsC.{i}.bias"]) +
              F.conv1d(xp, _w(w, f"convsP.{i}"), w[f"convsP.{i}.bias"]) +
              F.conv1d(xf, _w(w, f"convsF.{i}"), w[f"convsF.{i}.bias"]))
        if use_additional_convs:
            xt = F.conv1d(F.leaky_relu(xt, slope), _w(w, f"convsA.{i}.1"), w[f"convsA.{i}.1.bias"], padding=1)
        x = xt + x
    return x


def residual_block(w, x, k, dilations, slope):
    for i, dil in enumerate(dilations):
        xt = F.conv1d(F.leaky_relu(x, slope), _w(w, f"convs1.{i}.1"), w[f"convs1.{i}.1.bias"], dilation=dil,
                      padding=(k - 1) // 2 * dil)
        x = xt + x
    return x


def _sub(w, p):
    return {k[len(p):]: v for k, v in w.items() if k.startswith(p)}


def sifigan_forward(w, x, c, d, cfg=DEFAULT_CFG):
    """x (B, 1, T*hop) sine, c (B, 43, T), d list of (B, 1, T * cumprod(scales)[i]) -> (wave (B,1,T*hop), excitation)."""
    slope = cfg["nonlinear_activation_params"]["negative_slope"]
    ks, scales, uks = cfg["kernel_size"], cfg["upsample_scales"], cfg["upsample_kernel_sizes"]
    n_up = len(scales)
    sp, fp = cfg["source_network_params"], cfg["filter_network_params"]
    c = F.conv1d(c, _w(w, "input_conv"), w["input_conv.bias"], padding=(ks - 1) // 2)
    e = c

    def down(net, feats):
        out = [feats]
        for j, i in enumerate(reversed(range(1, n_up))):
            feats = F.leaky_relu(F.conv1d(feats, _w(w, f"{net}.downsamples.{j}.0"), w[f"{net}.downsamples.{j}.0.bias"],
                                          stride=scales[i], padding=scales[i] - (1 if uks[i] % 2 == 0 else 0)), slope)
            out.append(feats)
        return out

    def up(net, i, h):
        s = scales[i]
        return F.conv_transpose1d(F.leaky_relu(h, slope), _w(w, f"{net}.upsamples.{i}.1"),
                                  w[f"{net}.upsamples.{i}.1.bias"], stride=s, padding=s // 2 + s % 2,
                                  output_padding=s % 2)

    embs = down("sn", F.conv1d(x, _w(w, "sn.emb"), w["sn.emb.bias"], padding=(ks - 1) // 2))
    for i in range(n_up):
        e = up("sn", i, e) + embs[-i - 1]
        e = adaptive_residual_block(_sub(w, f"sn.blocks.{i}."), e, d[i], sp["resblock_dilations"][i], slope,
                                    sp["use_additional_convs"])
    exc = F.conv1d(e, _w(w, "sn.output_conv"), w["sn.output_conv.bias"], padding=(ks - 1) // 2)
    embs = down("sn" if cfg["share_downsamples"] else "fn", e)
    nb = len(fp["resblock_kernel_sizes"])
    for i in range(n_up):
        c = up("sn" if cfg["share_upsamples"] else "fn", i, c) + embs[-i - 1]
        cs = 0.0
        for j in range(nb):
            cs = cs + residual_block(_sub(w, f"fn.blocks.{i * nb + j}."), c, fp["resblock_kernel_sizes"][j],
                                     fp["resblock_dilations"][j], slope)
        c = cs / nb
    y = torch.tanh(F.conv1d(F.leaky_relu(c, slope), _w(w, "fn.output_conv.1"), w["fn.output_conv.1.bias"],
                            padding=(ks - 1) // 2))
    return y, exc
