"""GPU parity of the thin-conv strip kernel (conv_strip.hip: C_in, N in {32, 64}, LDS-resident weights, persistent
128-row tiles) through the one C-ABI entry point, against the executable ABI spec, for every channel combination and
the epilogues HiFi-GAN / SiFiGAN use on these layers (residual_block.py:243-258, hifigan.py:171-188)."""
import pytest
import torch

import serenade_amd
from serenade_amd import _lib, ops
from tests.test_hip_parity import KTOL, dev, rnd, run_conv_both  # noqa: F401  (dev is a fixture)

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _bf16x3():
    serenade_amd.set_precision("bf16x3")
    KTOL.k = 1e-4
    yield


@pytest.mark.parametrize("cin,n,k,dil", [(32, 32, 3, 1), (32, 32, 11, 5), (64, 64, 7, 3), (64, 32, 3, 3),
                                         (32, 64, 7, 1), (64, 64, 3, 5)])
def test_strip_conv_leaky_residual(dev, cin, n, k, dil):
    B, T = 2, 1100  # 9 tiles per batch item, ragged last tile
    x = rnd(B, T, cin, seed=cin + k)
    w = ops.pack_conv_weight(rnd(n, cin, k, seed=n + dil) * 0.2)
    res = rnd(B, T, n, seed=5)
    kw = dict(in0=x, w=w, bias=rnd(n, seed=6), out=torch.zeros(B, T, n), n_batch=B, T_in=T, T_out=T, C_in=cin, N=n,
              in0_bs=T * cin, ld_in0=cin, ldw=k * cin, out_bs=T * n, ld_out=n, taps=ops.conv_taps(k, dil),
              pro_act=_lib.ACT_LEAKY, pro_slope=0.1, len_in=torch.tensor([T, 777], dtype=torch.int32),
              res=res, res_mode=_lib.RES_ADD, res_bs=T * n, ld_res=n, no_halo=4)
    run_conv_both(dev, kw)


def test_strip_stage_mean_in_place(dev):
    """third resblock of a stage: out = (acc0 + conv(x) + x_res) / 3 written over acc0 (hifigan.py:183-186)"""
    B, T, C, k = 1, 2048, 32, 11
    x = rnd(B, T, C, seed=1)
    xr = rnd(B, T, C, seed=2)
    acc0 = rnd(B, T, C, seed=3)
    kw = dict(in0=x, w=ops.pack_conv_weight(rnd(C, C, k, seed=4) * 0.2), bias=rnd(C, seed=5), out=acc0, n_batch=B,
              T_in=T, T_out=T, C_in=C, N=C, in0_bs=T * C, ld_in0=C, ldw=k * C, out_bs=T * C, ld_out=C,
              taps=ops.conv_taps(k, 1), pro_act=_lib.ACT_LEAKY, pro_slope=0.1, res=xr, res_mode=_lib.RES_ADD,
              res_bs=T * C, ld_res=C, res2=acc0, res2_bs=T * C, ld_res2=C, post=_lib.POST_DIV, post_div=3.0, no_halo=4)
    run_conv_both(dev, kw)


def test_strip_is_selected_and_matches_tiled(dev):
    """same launch with the strip / halo kernels disabled (no_halo=1 -> tiled conv_fast): identical up to rounding"""
    B, T, C, k = 2, 4096, 64, 3
    x = rnd(B, T, C, seed=7).to(dev)
    w = ops.pack_conv_weight(rnd(C, C, k, seed=8) * 0.2).to(dev)
    outs = []
    for nh in (4, 1):
        out = torch.zeros(B, T, C, device=dev)
        ops.ConvOp(in0=x, w=w, out=out, n_batch=B, T_in=T, T_out=T, C_in=C, N=C, in0_bs=T * C, ld_in0=C, ldw=k * C,
                   out_bs=T * C, ld_out=C, taps=ops.conv_taps(k, 1), no_halo=nh)()
        outs.append(out.cpu())
    assert (outs[0] - outs[1]).abs().max() <= 1e-5 * outs[1].abs().max()


@pytest.mark.parametrize("force,cin,n,k,dil", [(4, 32, 32, 11, 5), (4, 64, 64, 7, 3), (2, 128, 128, 11, 3), (2, 256, 256, 7, 1),
                                               (1, 128, 64, 3, 1)])
def test_forced_kernels_against_torch_conv1d(dev, force, cin, n, k, dil):
    """strip (4), halo (2) and tiled (1) kernels against torch.nn.functional directly -- not only the ABI emulator:
    the HiFi-GAN residual-unit conv `conv1d(leaky_relu(x), w, b, dilation)` + residual (residual_block.py:243-258)"""
    import torch.nn.functional as F
    serenade_amd.set_precision("fp32")
    B, T = 2, 1500
    x, w, b = rnd(B, T, cin, seed=1), rnd(n, cin, k, seed=2) * 0.1, rnd(n, seed=3)
    res = rnd(B, T, n, seed=4)
    ref = F.conv1d(F.leaky_relu(x, 0.1).transpose(1, 2), w, b, dilation=dil, padding=(k - 1) // 2 * dil).transpose(1, 2) + res
    out = torch.zeros(B, T, n, device=dev)
    ops.ConvOp(in0=x.to(dev), w=ops.pack_conv_weight(w).to(dev), bias=b.to(dev), out=out, n_batch=B, T_in=T, T_out=T,
               C_in=cin, N=n, in0_bs=T * cin, ld_in0=cin, ldw=k * cin, out_bs=T * n, ld_out=n, taps=ops.conv_taps(k, dil),
               pro_act=_lib.ACT_LEAKY, pro_slope=0.1, res=res.to(dev), res_mode=_lib.RES_ADD, res_bs=T * n, ld_res=n,
               no_halo=force)()
    assert ((out.cpu() - ref).abs().max() / ref.abs().max()).item() < 2e-6
