"""GPU tests of conv_f32.hip, the exact-fp32 contraction with a vector-ALU-free loop (buffer loads, scalar K walk,
out-of-range offsets for padded rows, magic-number tile coordinates; tile ids 7 / 9 / 10 in fp32).

* tiles 7, 9 and 11 (64 x 64 with loads two steps ahead: small grids, split-K) keep conv_fast.hip's LDS image and MFMA order: results must be BIT-identical to its fp32 loop
  (no_halo=5) on every addressing feature of the path (taps, dilation, stride 2, reflection, two-tensor concat, item
  lengths, heads with strides, batched B operand, LeakyReLU prologue, residual / GroupNorm-partial epilogues);
* tile 10 (32 x 64, each 32-deep step split over two wave pairs, partial sums joined through LDS) against F.conv1d in
  float64 and bit-reproducible run to run;
* the half-resolution shape of the headline (8 x 640 rows x 512 columns) takes tile 10 by itself.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import serenade_amd
from serenade_amd import _lib, ops

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "needs an MI355X"
    _lib.lib()
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def fp32_mode():
    serenade_amd.set_precision("fp32")
    yield
    serenade_amd.set_precision("fp32")  # the package default


def rnd(*s, seed=0):
    return torch.from_numpy(np.random.default_rng(seed).standard_normal(s).astype(np.float32))


def run(dev, kw, **over):
    g = {k: (v.to(dev).clone() if isinstance(v, torch.Tensor) else v) for k, v in dict(kw, **over).items()}
    ops.ConvOp(**g)()
    torch.cuda.synchronize()
    return g


def cases():
    B, Tn, C, N = 3, 333, 96, 192
    x = rnd(B, Tn, C, seed=1)
    lens = torch.tensor([333, 200, 65], dtype=torch.int32)
    base = dict(in0=x, n_batch=B, T_in=Tn, T_out=Tn, C_in=C, N=N, in0_bs=Tn * C, ld_in0=C, out_bs=Tn * N, ld_out=N)
    out = lambda *s: torch.zeros(*s)
    yield "k3 + lengths + gn partials", dict(base, w=ops.pack_conv_weight(rnd(N, C, 3, seed=2)), ldw=3 * C, bias=rnd(N, seed=3),
                                             taps=ops.conv_taps(3), len_in=lens, len_out=lens, out=out(B, Tn, N),
                                             gn_partials=torch.zeros(B, (Tn + 31) // 32, N // 32, 2))
    yield "k7 reflect leaky", dict(base, w=ops.pack_conv_weight(rnd(N, C, 7, seed=4)), ldw=7 * C, bias=rnd(N, seed=5),
                                   taps=ops.conv_taps(7), reflect=True, pro_act=_lib.ACT_LEAKY, pro_slope=0.2, out=out(B, Tn, N))
    yield "k11 dilation 5 leaky + residual", dict(base, w=ops.pack_conv_weight(rnd(N, C, 11, seed=6)), ldw=11 * C,
                                                  taps=ops.conv_taps(11, 5), pro_act=_lib.ACT_LEAKY, pro_slope=0.1,
                                                  out=out(B, Tn, N), res=rnd(B, Tn, N, seed=7), res_mode=_lib.RES_ADD,
                                                  res_bs=Tn * N, ld_res=N)
    To = (Tn + 1) // 2
    yield "k3 stride 2", dict(base, w=ops.pack_conv_weight(rnd(N, C, 3, seed=8)), ldw=3 * C, taps=ops.conv_taps(3), in_stride=2,
                              T_out=To, out_bs=To * N, out=out(B, To, N), len_in=lens)
    s = rnd(B, Tn, 64, seed=9)
    yield "concat of two tensors, k3", dict(base, in1=s, C_in0=C, C_in=C + 64, in1_bs=Tn * 64, ld_in1=64,
                                            w=ops.pack_conv_weight(rnd(N, C + 64, 3, seed=10)), ldw=3 * (C + 64),
                                            taps=ops.conv_taps(3), out=out(B, Tn, N))
    # attention-shaped: heads with column strides on A, a B operand per (batch, head)
    H, D, L = 2, 64, 200
    q, k = rnd(B, L, H * D, seed=11), rnd(B, L, H * D, seed=12)
    yield "Q K^T with heads", dict(in0=q, w=k, out=out(B, H, L, L), n_batch=B, n_head=H, T_in=L, T_out=L, C_in=D, N=L,
                                   in0_bs=L * H * D, in0_hs=D, ld_in0=H * D, w_bs=L * H * D, w_hs=D, ldw=H * D,
                                   out_bs=H * L * L, out_hs=L * L, ld_out=L, alpha=0.125)


@pytest.mark.parametrize("tile", [5, 6, 7, 9, 11])
def test_bit_identical_to_conv_fast_fp32(dev, tile):
    for name, kw in cases():
        a = run(dev, kw, tile=tile)
        b = run(dev, kw, tile=tile, no_halo=5)
        assert torch.equal(a["out"], b["out"]), f"{name}: tile {tile} differs from conv_fast.hip's fp32 loop"
        if "gn_partials" in kw:  # sums of the same 1024 stored values, accumulated in a different order
            torch.testing.assert_close(a["gn_partials"], b["gn_partials"], rtol=2e-6, atol=2e-4, msg=name)


def _ref(kw):
    """float64 F.conv1d of the same case (single-tensor, zero / reflect padded ones)"""
    x = kw["in0"].double()
    B, Tn, C = x.shape
    if "len_in" in kw:
        x = x * (torch.arange(Tn)[None] < kw["len_in"][:, None]).double().unsqueeze(-1)
    if kw.get("pro_act") == _lib.ACT_LEAKY:
        x = F.leaky_relu(x, kw["pro_slope"])
    taps = kw["taps"]
    k = len(taps)
    d = taps[1] - taps[0] if k > 1 else 1
    w = kw["w"].double().view(kw["N"], k, C).permute(0, 2, 1)
    pad = -taps[0]
    xt = x.transpose(1, 2)
    if kw.get("reflect"):
        xt, pad = F.pad(xt, (pad, pad), mode="reflect"), 0
    y = F.conv1d(xt, w, kw["bias"].double() if kw.get("bias") is not None else None, stride=kw.get("in_stride", 1), dilation=d,
                 padding=pad).transpose(1, 2)
    if "res" in kw:
        y = y + kw["res"].double()
    if "len_out" in kw:
        y = y * (torch.arange(y.shape[1])[None] < kw["len_out"][:, None]).double().unsqueeze(-1)
    return y


def test_split_step_tile_against_fp64(dev):
    for name, kw in cases():
        if "in1" in kw or "n_head" in kw:
            continue
        a = run(dev, kw, tile=10)
        b = run(dev, kw, tile=10)
        assert torch.equal(a["out"], b["out"]), f"{name}: tile 10 is not reproducible"
        ref = _ref(kw)
        e10 = ((a["out"].cpu().double() - ref).abs().max() / ref.abs().max()).item()
        e7 = ((run(dev, kw, tile=7)["out"].cpu().double() - ref).abs().max() / ref.abs().max()).item()
        assert e10 < 2e-6 and e10 < 2.0 * e7 + 1e-7, f"{name}: tile 10 {e10:.2e} vs tile 7 {e7:.2e}"


def test_half_resolution_shape_takes_the_split_step_tile(dev):
    """8 x 640 rows x 512 columns = 640 tiles of 64 x 64 (3 on some CUs, 2 on others) or 1280 of 32 x 64 (5 on each):
    pick_tile chooses the latter, and the result equals a forced tile-10 run bit for bit, a tile-7 run to fp32 noise"""
    B, Tn, C, N = 8, 640, 512, 512
    kw = dict(in0=rnd(B, Tn, C, seed=21), w=ops.pack_conv_weight(rnd(N, C, 3, seed=22) * 0.05), ldw=3 * C, bias=rnd(N, seed=23),
              taps=ops.conv_taps(3), out=torch.zeros(B, Tn, N), n_batch=B, T_in=Tn, T_out=Tn, C_in=C, N=N, in0_bs=Tn * C,
              ld_in0=C, out_bs=Tn * N, ld_out=N, gn_partials=torch.zeros(B, Tn // 32, N // 32, 2))
    auto, t10, t7 = run(dev, kw), run(dev, kw, tile=10), run(dev, kw, tile=7)
    assert torch.equal(auto["out"], t10["out"]) and torch.equal(auto["gn_partials"], t10["gn_partials"])
    assert not torch.equal(t10["out"], t7["out"])  # a different summation order, so really a different kernel
    err = ((t10["out"] - t7["out"]).abs().max() / t7["out"].abs().max()).item()
    assert err < 2e-6, err
