"""GPU: the receptive-field (halo) conv kernel against the generic implicit-GEMM kernel and the C-ABI spec, in
split-bf16, over the tap/dilation/tile combinations of the path (k3, k7 d3, k11 d5, 2-tap transposed phases, concat
input, masks, ragged tiles, GroupNorm partials, residual epilogues)."""
import numpy as np
import pytest
import torch

import serenade_amd
from serenade_amd import _lib, ops
from tests import _emulator

pytestmark = pytest.mark.gpu


def rnd(*s, seed=0):
    return torch.from_numpy(np.random.default_rng(seed).standard_normal(s).astype(np.float32))


def nerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


CASES = [
    # (B, T, Cin, N, k, dil, tile, extra)
    (2, 300, 64, 128, 3, 1, 1, {}),
    (2, 300, 64, 128, 3, 1, 3, {}),
    (1, 517, 32, 32, 11, 5, 5, {"leaky": 0.1}),
    (2, 200, 32, 64, 7, 3, 2, {"leaky": 0.1}),
    (2, 200, 96, 64, 7, 1, 4, {}),
    (2, 130, 64, 64, 11, 1, 0, {"leaky": 0.1}),
    (3, 70, 128, 256, 3, 1, 0, {"mask": True, "gn": True}),
]


@pytest.mark.parametrize("case", CASES)
def test_halo_matches_generic_and_spec(case):
    B, Tn, C, N, k, d, tile, ex = case
    serenade_amd.set_precision("bf16x3")
    dev = torch.device("cuda:0")
    x = rnd(B, Tn, C, seed=1)
    w = ops.pack_conv_weight(rnd(N, C, k, seed=2) / np.sqrt(C * k))
    bias = rnd(N, seed=3)
    res = rnd(B, Tn, N, seed=4)
    lens = torch.tensor([Tn, max(1, Tn - 37), Tn // 2][:B], dtype=torch.int32) if ex.get("mask") else None
    base = dict(n_batch=B, T_in=Tn, T_out=Tn, C_in=C, N=N, in0_bs=Tn * C, ld_in0=C, ldw=k * C, out_bs=Tn * N,
                ld_out=N, taps=ops.conv_taps(k, d), tile=tile, res_mode=_lib.RES_ADD, res_bs=Tn * N, ld_res=N)
    if "leaky" in ex:
        base.update(pro_act=_lib.ACT_LEAKY, pro_slope=ex["leaky"])
    outs = {}
    for name, no_halo in (("halo", 2), ("generic", 1)):
        out = torch.zeros(B, Tn, N, device=dev)
        gn = torch.zeros(B, (Tn + 31) // 32, N // 32, 2, device=dev) if ex.get("gn") else None
        ops.ConvOp(in0=x.to(dev), w=w.to(dev), bias=bias.to(dev), res=res.to(dev), out=out, gn_partials=gn,
                   len_in=None if lens is None else lens.to(dev), no_halo=no_halo, **base)()
        outs[name] = (out.cpu(), None if gn is None else gn.cpu())
    assert nerr(outs["halo"][0], outs["generic"][0]) < 2e-6  # same split, same products, same k order per tap
    cpu_out = torch.zeros(B, Tn, N)
    cpu_gn = torch.zeros(B, (Tn + 31) // 32, N // 32, 2) if ex.get("gn") else None
    _emulator.emul_conv(dict(base, in0=x, w=w, bias=bias, res=res, out=cpu_out, gn_partials=cpu_gn, len_in=lens))
    assert nerr(outs["halo"][0], cpu_out) < 1e-4
    if cpu_gn is not None:
        assert nerr(outs["halo"][1], cpu_gn) < 1e-4


def test_halo_concat_input_and_transposed_phase():
    serenade_amd.set_precision("bf16x3")
    dev = torch.device("cuda:0")
    B, Tn, C0, C1, N = 2, 150, 64, 32, 64
    x, s = rnd(B, Tn, C0, seed=5), rnd(B, Tn, C1, seed=6)
    w = ops.pack_conv_weight(rnd(N, C0 + C1, 3, seed=7) / 17.0)
    kw = dict(in1_bs=Tn * C1, ld_in1=C1, C_in0=C0, n_batch=B, T_in=Tn, T_out=Tn, C_in=C0 + C1, N=N, in0_bs=Tn * C0,
              ld_in0=C0, ldw=3 * (C0 + C1), out_bs=Tn * N, ld_out=N, taps=ops.conv_taps(3))
    got = {}
    for no_halo in (2, 1):
        out = torch.zeros(B, Tn, N, device=dev)
        ops.ConvOp(in0=x.to(dev), in1=s.to(dev), w=w.to(dev), out=out, no_halo=no_halo, **kw)()
        got[no_halo] = out.cpu()
    ref = torch.zeros(B, Tn, N)
    _emulator.emul_conv(dict(kw, in0=x, in1=s, w=w, out=ref))
    assert nerr(got[2], got[1]) < 2e-6 and nerr(got[2], ref) < 1e-4
    # a 2-tap transposed-conv phase with strided output rows
    wt = rnd(C0, 32, 10, seed=8) / 10.0
    To = 5 * Tn
    outs = {}
    for no_halo in (2, 1):
        out = torch.zeros(B, To, 32, device=dev)
        for r, (taps, wp) in enumerate(ops.convtranspose_phases(wt, 5, 3)):
            ops.ConvOp(in0=x.to(dev), w=wp.to(dev), out=out, n_batch=B, T_in=Tn, T_out=Tn, C_in=C0, N=32,
                       in0_bs=Tn * C0, ld_in0=C0, ldw=wp.shape[1], out_bs=To * 32, ld_out=32, taps=taps,
                       out_t_stride=5, out_t_off=r, pro_act=_lib.ACT_LEAKY, pro_slope=0.1, no_halo=no_halo)()
        outs[no_halo] = out.cpu()
    ref = torch.nn.functional.conv_transpose1d(torch.nn.functional.leaky_relu(x, 0.1).transpose(1, 2), wt, None,
                                               stride=5, padding=3, output_padding=1).transpose(1, 2)
    assert nerr(outs[2], outs[1]) < 2e-6 and nerr(outs[2], ref) < 1e-4
