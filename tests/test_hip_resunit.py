"""GPU tests of the two round-2 launch-level kernels, through the C ABI, in both contraction modes:

* srn_hifigan_resunit (resunit.hip): one fused HiFi-GAN residual unit vs torch.nn.functional.conv1d on the same
  weights (reference semantics: serenade/vocoder/layers/residual_block.py:243-258), every (C, k, dilation) of the path,
  sequence ends inside / across tiles, with and without the stage sum / mean epilogue;
* split-K (conv_splitk.hip): small tile grids with deep contractions take the sliced path (asserted through
  srn_conv_gemm_workspace_bytes) and agree with the executable spec, with every epilogue feature it reduces.
"""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import serenade_amd
from serenade_amd import _lib, ops
from tests import _emulator

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["fp32", "bf16x3", "bf16x6"])
def precision(request):
    serenade_amd.set_precision(request.param)
    yield request.param
    serenade_amd.set_precision("fp32")  # the package default


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    _lib.lib()
    return torch.device("cuda:0")


def tol():
    return 1e-4 if serenade_amd.get_precision() == "bf16x3" else 2e-5


def nerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


def rnd(*s, seed=0, scale=1.0):
    return torch.from_numpy((scale * np.random.default_rng(seed).standard_normal(s)).astype(np.float32))


def ref_unit(x, w1, b1, w2, b2, k, d, slope, res2=None, div=0.0):
    """x (B, T, C) channels-last; w (C_out, C_in, k) torch layout; fp64 reference"""
    xd = x.double().transpose(1, 2)
    xt = F.conv1d(F.leaky_relu(xd, slope), w1.double(), b1.double(), padding=(k - 1) // 2 * d, dilation=d)
    xt = F.conv1d(F.leaky_relu(xt, slope), w2.double(), b2.double(), padding=(k - 1) // 2)
    y = (xt + xd).transpose(1, 2)
    if res2 is not None:
        y = y + res2.double()
    if div not in (0.0, 1.0):
        y = y / div
    return y


@pytest.mark.parametrize("C", [32, 64])
@pytest.mark.parametrize("k,d", [(3, 1), (3, 5), (7, 3), (11, 1), (11, 5), (5, 2)])
def test_resunit_vs_conv1d(dev, C, k, d):
    slope = 0.1
    for case, (B, T, with_sum) in enumerate([(2, 37, False), (1, 1000, True), (3, 2 * (256 - (k - 1)) + 1, True)]):
        x = rnd(B, T, C, seed=10 * case + 1)
        w1 = rnd(C, C, k, seed=10 * case + 2, scale=1.0 / np.sqrt(C * k))
        w2 = rnd(C, C, k, seed=10 * case + 3, scale=1.0 / np.sqrt(C * k))
        b1, b2 = rnd(C, seed=10 * case + 4, scale=0.1), rnd(C, seed=10 * case + 5, scale=0.1)
        res2 = rnd(B, T, C, seed=10 * case + 6) if with_sum else None
        div = 3.0 if with_sum else 0.0
        ref = ref_unit(x, w1, b1, w2, b2, k, d, slope, res2, div)
        g = lambda t: None if t is None else t.to(dev)
        out = torch.full((B, T, C), float("nan"), device=dev)
        guard = torch.zeros(64, device=dev)  # allocated right after `out`: catches writes past the end
        op = ops.ResUnitOp(x=g(x), w1=ops.pack_conv_weight(g(w1)), b1=g(b1), w2=ops.pack_conv_weight(g(w2)), b2=g(b2),
                           out=out, n_batch=B, T=T, C=C, k=k, dilation=d, slope=slope, res2=g(res2), post_div=div)
        op()
        torch.cuda.synchronize()
        assert torch.isfinite(out).all(), f"C={C} k={k} d={d} case {case}: unwritten / non-finite outputs"
        assert nerr(out, ref) < tol(), f"C={C} k={k} d={d} case {case}"
        assert guard.abs().max().item() == 0


@pytest.mark.parametrize("C", [32, 64])
def test_resunit_fp32_forms_agree_bit_for_bit(dev, C, monkeypatch):
    """exact fp32 has two implementations of the fused unit: resunit_f32.hip (buffer addressing, no vector-ALU work beside
    the fp32 MFMAs) and resunit.hip's instantiation (SERENADE_AMD_RESUNIT_SHARED_FP32=1): same LDS image, same MFMA order,
    same epilogue order -- every unit of the path, sequence ends inside / across tiles, must agree bit for bit"""
    if serenade_amd.get_precision() != "fp32":
        pytest.skip("fp32 arm only")
    slope = 0.1
    for (k, d) in [(3, 1), (3, 5), (7, 3), (11, 5)]:
        for case, (B, T, with_sum) in enumerate([(2, 37, False), (3, 2 * (256 - (k - 1)) + 1, True), (2, 1531, True)]):
            x = rnd(B, T, C, seed=case + 1).to(dev)
            w1 = ops.pack_conv_weight(rnd(C, C, k, seed=case + 2, scale=1.0 / np.sqrt(C * k)).to(dev))
            w2 = ops.pack_conv_weight(rnd(C, C, k, seed=case + 3, scale=1.0 / np.sqrt(C * k)).to(dev))
            b1, b2 = rnd(C, seed=case + 4, scale=0.1).to(dev), rnd(C, seed=case + 5, scale=0.1).to(dev)
            res2 = rnd(B, T, C, seed=case + 6).to(dev) if with_sum else None
            outs = []
            for shared in ("0", "1"):
                monkeypatch.setenv("SERENADE_AMD_RESUNIT_SHARED_FP32", shared)
                out = torch.full((B, T, C), float("nan"), device=dev)
                ops.ResUnitOp(x=x, w1=w1, b1=b1, w2=w2, b2=b2, out=out, n_batch=B, T=T, C=C, k=k, dilation=d, slope=slope,
                              res2=res2, post_div=3.0 if with_sum else 0.0)()
                torch.cuda.synchronize()
                outs.append(out)
            assert torch.isfinite(outs[0]).all()
            assert torch.equal(outs[0], outs[1]), f"C={C} k={k} d={d} case {case}"


def test_resunit_matches_the_unfused_pair_and_rejects_bad_args(dev):
    """same unit as two srn_conv_gemm launches (the round-1 path) and as one fused launch"""
    B, T, C, k, d, slope = 2, 700, 64, 7, 3, 0.1
    x, b1, b2 = rnd(B, T, C, seed=1).to(dev), rnd(C, seed=4, scale=0.1).to(dev), rnd(C, seed=5, scale=0.1).to(dev)
    w1 = ops.pack_conv_weight(rnd(C, C, k, seed=2, scale=0.05).to(dev))
    w2 = ops.pack_conv_weight(rnd(C, C, k, seed=3, scale=0.05).to(dev))
    xt, y0, y1 = (torch.zeros(B, T, C, device=dev) for _ in range(3))
    conv = lambda i, w, b, o, taps, **kw: ops.ConvOp(in0=i, w=w, out=o, n_batch=B, T_in=T, T_out=T, C_in=C, N=C,
                                                      in0_bs=T * C, ld_in0=C, ldw=w.shape[1], out_bs=T * C, ld_out=C,
                                                      bias=b, taps=taps, pro_act=ops.ACT_LEAKY, pro_slope=slope, **kw)
    conv(x, w1, b1, xt, ops.conv_taps(k, d))()
    conv(xt, w2, b2, y0, ops.conv_taps(k, 1), res=x, res_mode=ops.RES_ADD, res_bs=T * C, ld_res=C)()
    ops.ResUnitOp(x=x, w1=w1, b1=b1, w2=w2, b2=b2, out=y1, n_batch=B, T=T, C=C, k=k, dilation=d, slope=slope)()
    torch.cuda.synchronize()
    assert nerr(y1, y0) < tol()
    with pytest.raises(RuntimeError, match="32 or 64"):
        ops.ResUnitOp(x=x, w1=w1, b1=b1, w2=w2, b2=b2, out=y1, n_batch=B, T=T, C=128, k=k, dilation=d, slope=slope,
                      precision=_lib.PREC_FP32)()
    with pytest.raises(RuntimeError, match="alias"):
        ops.ResUnitOp(x=x, w1=w1, b1=b1, w2=w2, b2=b2, out=x, n_batch=B, T=T, C=C, k=k, dilation=d, slope=slope)()


SHAPE_CLASSES = [
    # (name, Z, heads, M, N, K, taps): every contraction shape class of the path, incl. the long-form attention
    ("1x1 K=256", 1, 1, 640, 512, 256, 1),
    ("linear K=512 (QKV / GEGLU width)", 1, 1, 512, 1024, 512, 1),
    ("conv k3 K=1536", 2, 1, 333, 512, 512, 3),
    ("linear K=2048 (o-proj / FF2)", 1, 1, 384, 512, 2048, 1),
    ("conv k3 K=3072 (up block on the concat)", 1, 1, 320, 512, 1024, 3),
    ("conv k11 K=2816 (HiFi-GAN, C=256)", 1, 1, 512, 256, 256, 11),
    ("Q K^T, L=1280, d=512", 1, 2, 1280, 1280, 512, 1),
    ("P V, L=1280", 1, 2, 1280, 512, 1280, 1),
    ("Q K^T, L=4352, d=512", 1, 1, 4352, 4352, 512, 1),
    ("P V, L=4352", 1, 1, 4352, 512, 4352, 1),
]


@pytest.mark.parametrize("case", SHAPE_CLASSES, ids=[c[0] for c in SHAPE_CLASSES])
def test_bf16x6_error_table_per_shape_class(dev, precision, case):
    """fp32-faithfulness of bf16x6 per contraction shape class of the path (K = 256 ... 3072 over taps, the HiFi-GAN k 11
    conv, Q K^T and P V at L = 1280 and 4352): max and rms error against fp64 must not exceed the exact-fp32 MFMA chain's
    beyond summation-order noise.  P V's A operand is a softmax row (non-negative, sums to 1) as on the path."""
    if precision != "fp32":
        pytest.skip("one arm runs both modes")
    name, Z, H, M, N, K, taps = case
    att = "Q K^T" in name or "P V" in name
    g = np.random.default_rng(abs(hash(name)) % 1000)
    if "P V" in name:
        a = torch.softmax(torch.from_numpy(g.standard_normal((Z, H, M, K)).astype(np.float32)) * 3.0, -1)
        w = torch.from_numpy(g.standard_normal((Z, H, N, K)).astype(np.float32))
    elif att:
        a = torch.from_numpy(g.standard_normal((Z, H, M, K)).astype(np.float32))
        w = torch.from_numpy(g.standard_normal((Z, H, N, K)).astype(np.float32))
    else:
        a = torch.from_numpy(g.standard_normal((Z, M, K)).astype(np.float32))
        w = torch.from_numpy((g.standard_normal((N, K, taps)) / np.sqrt(K * taps)).astype(np.float32))
    errs = {}
    for mode in ("fp32", "bf16x6"):
        serenade_amd.set_precision(mode)
        if att:
            out = torch.zeros(Z, H, M, N, device=dev)
            ops.ConvOp(in0=a.to(dev), w=w.to(dev), out=out, n_batch=Z, n_head=H, T_in=M, T_out=M, C_in=K, N=N,
                       in0_bs=H * M * K, in0_hs=M * K, ld_in0=K, w_bs=H * N * K, w_hs=N * K, ldw=K, out_bs=H * M * N,
                       out_hs=M * N, ld_out=N, alpha=1.0 / np.sqrt(K) if "Q K^T" in name else 1.0)()
            ref = torch.einsum("zhmk,zhnk->zhmn", a.double(), w.double()) * (1.0 / np.sqrt(K) if "Q K^T" in name else 1.0)
        else:
            out = torch.zeros(Z, M, N, device=dev)
            ops.ConvOp(in0=a.to(dev), w=ops.pack_conv_weight(w.to(dev)), out=out, n_batch=Z, T_in=M, T_out=M, C_in=K, N=N,
                       in0_bs=M * K, ld_in0=K, ldw=taps * K, out_bs=M * N, ld_out=N, taps=ops.conv_taps(taps))()
            ref = F.conv1d(a.double().transpose(1, 2), w.double(), padding=(taps - 1) // 2).transpose(1, 2)
        torch.cuda.synchronize()
        d = out.cpu().double() - ref
        errs[mode] = (d.abs().max().item() / ref.abs().max().item(), (d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item())
    serenade_amd.set_precision("fp32")
    print(f"{name}: relative max / rms error vs fp64  fp32 {errs['fp32'][0]:.2e} / {errs['fp32'][1]:.2e}   "
          f"bf16x6 {errs['bf16x6'][0]:.2e} / {errs['bf16x6'][1]:.2e}")
    assert errs["bf16x6"][0] <= 1.5 * errs["fp32"][0] + 1e-9 and errs["bf16x6"][1] <= 1.25 * errs["fp32"][1] + 1e-10, errs


class Mirror:
    """CPU tensors <-> device clones, so one kw dict drives the kernel and the spec"""

    def __init__(self, dev):
        self.dev, self.map = dev, {}

    def __call__(self, v):
        if isinstance(v, torch.Tensor):
            if id(v) not in self.map:
                self.map[id(v)] = (v, v.to(self.dev))
            return self.map[id(v)][1]
        if isinstance(v, tuple) and len(v) == 2 and isinstance(v[0], torch.Tensor):
            return (self(v[0]), v[1])
        return v


@pytest.mark.parametrize("case", ["plain_gn", "residual_inplace", "axpy_masked", "strided_leaky_concat"])
def test_splitk_path_matches_spec(dev, case):
    B, T, K, N = 1, 150, 256, 160  # 3 x 3 tiles of 64 x 64: the grid cannot fill the chip; 3 taps x 8 chunks
    taps = [-1, 0, 1]
    kw = dict(in0=rnd(B, T, K, seed=1), w=rnd(N, 3 * K, seed=2, scale=0.05), bias=rnd(N, seed=3), n_batch=B, T_in=T,
              T_out=T, C_in=K, N=N, in0_bs=T * K, ld_in0=K, ldw=3 * K, out_bs=T * N, ld_out=N, taps=taps,
              out=torch.zeros(B, T, N))
    if case == "plain_gn":
        kw.update(N=160, gn_partials=torch.zeros(B, (T + 31) // 32, N // 32, 2), alpha=0.5)
    elif case == "residual_inplace":
        o = rnd(B, T, N, seed=4)
        kw.update(out=o, res=o, res_mode=ops.RES_ADD, res_bs=T * N, ld_res=N, post=ops.POST_DIV, post_div=3.0,
                  res2=rnd(B, T, N, seed=5), res2_bs=T * N, ld_res2=N)
    elif case == "axpy_masked":
        o = rnd(B, T, N, seed=6)
        kw.update(out=o, res=o, res_mode=ops.RES_AXPY, beta=0.1, res_bs=T * N, ld_res=N,
                  len_out=torch.tensor([101], dtype=torch.int32), len_in=torch.tensor([120], dtype=torch.int32))
    else:
        K0 = 96
        kw.update(in0=rnd(B, T, K0, seed=7), in1=rnd(B, T, K - K0, seed=8), C_in0=K0, in0_bs=T * K0, ld_in0=K0,
                  in1_bs=T * (K - K0), ld_in1=K - K0, pro_act=ops.ACT_LEAKY, pro_slope=0.2,
                  out=torch.zeros(B, 2 * T, N), out_bs=2 * T * N, out_t_stride=2, out_t_off=1)
    m = Mirror(dev)
    gpu = {k: m(v) for k, v in kw.items()}
    op = ops.ConvOp(**gpu)
    need = _lib.lib().srn_conv_gemm_workspace_bytes(ctypes.byref(op.p))
    assert need > 0 and op.p.ws and op.p.ws_bytes >= need, "this shape must take the split-K path"
    op()
    torch.cuda.synchronize()
    _emulator.emul_conv(kw)
    for c, g_ in m.map.values():
        if c.is_floating_point():
            assert nerr(g_, c) < tol(), case


# ------------------------------------------------------------------------------------------------------------------
#  bf16x6: the fp32-faithful emulation must be AT LEAST as accurate as the exact-fp32 MFMA mode
# ------------------------------------------------------------------------------------------------------------------
def test_three_way_split_is_exact_on_the_host():
    """x = hi + mid + lo exactly, for fp32 values over 60 binades (the split ops.weight_planes and the kernels apply)"""
    rng = np.random.default_rng(0)
    x = torch.from_numpy((rng.standard_normal(200000) * np.exp2(rng.integers(-30, 30, 200000))).astype(np.float32))
    hi = x.to(torch.bfloat16)
    r1 = x - hi.float()
    mid = r1.to(torch.bfloat16)
    lo = (r1 - mid.float()).to(torch.bfloat16)
    assert torch.equal(hi.float() + mid.float() + lo.float(), x)  # (hi + mid) + lo re-rounds to x: sums are exact
    assert torch.equal(hi.double() + mid.double() + lo.double(), x.double())


def test_bf16x6_gemm_error_is_not_above_exact_fp32(dev, precision):
    """max and rms error of a K = 2048 contraction against fp64, per mode: bf16x6 must not exceed the exact-fp32 MFMA
    path's error by more than summation-order noise (it keeps 24-bit operands and drops only <= 2^-26 of a product)"""
    if precision != "fp32":
        pytest.skip("one arm runs all three modes")
    M, N, K = 384, 256, 2048
    a, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1.0 / np.sqrt(K)), rnd(N, seed=3)
    ref = a.double() @ w.double().t() + b.double()
    errs = {}
    for mode in ("fp32", "bf16x6", "bf16x3"):
        serenade_amd.set_precision(mode)
        out = torch.zeros(M, N, device=dev)
        ops.ConvOp(in0=a.to(dev), w=w.to(dev), out=out, bias=b.to(dev), n_batch=1, T_in=M, T_out=M, C_in=K, N=N,
                   ld_in0=K, ldw=K, ld_out=N)()
        torch.cuda.synchronize()
        d = (out.cpu().double() - ref)
        errs[mode] = (d.abs().max().item(), d.pow(2).mean().sqrt().item())
    serenade_amd.set_precision("fp32")
    print("GEMM error vs fp64 (max, rms):", errs)
    assert errs["bf16x6"][0] <= 1.5 * errs["fp32"][0] and errs["bf16x6"][1] <= 1.25 * errs["fp32"][1]
    assert errs["bf16x3"][1] > 4 * errs["fp32"][1]  # the two-plane split is visibly coarser: this test can tell
