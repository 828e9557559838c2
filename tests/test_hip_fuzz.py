"""GPU fuzz of the one contraction entry point (srn_conv_gemm): seeded random shapes / taps / strides / paddings /
masks / concat inputs / prologues / epilogues, each run through every kernel the dispatcher can pick (conv_fast,
conv_strip, conv_halo, the generic conv_gemm kernel; both arithmetic modes) and compared with
the executable spec of the ABI (tests/_emulator.py) on identical inputs."""
import numpy as np
import pytest
import torch

import serenade_amd
from serenade_amd import _lib, ops
from tests.test_hip_parity import KTOL, dev, rnd, run_conv_both  # noqa: F401  (dev is a fixture)

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["fp32", "bf16x3", "bf16x6"])
def precision(request):
    serenade_amd.set_precision(request.param)
    KTOL.k = 1e-4 if request.param == "bf16x3" else 2e-5
    yield request.param
    serenade_amd.set_precision("fp32")  # the package default


def make_case(seed):
    rng = np.random.default_rng(1000 + seed)
    pick = lambda xs: xs[int(rng.integers(len(xs)))]
    B = pick([1, 2, 3])
    k = pick([1, 1, 2, 3, 3, 5, 7, 11])
    dil = pick([1, 1, 2, 3, 5]) if k > 1 else 1
    stride = pick([1, 1, 1, 2]) if k == 3 and dil == 1 else 1
    C = pick([32, 64, 64, 96, 128, 256, 20, 80, 100])
    N = pick([8, 32, 32, 48, 64, 64, 80, 128, 160, 256])
    T = int(rng.integers(max(8, (k - 1) * dil + 2), 420))
    if pick([0, 0, 1]):
        T = int(rng.integers(600, 900))  # long enough for the strip kernel's tile loop
    To = (T + stride - 1) // stride
    taps = ops.conv_taps(k, dil)
    w = rnd(N, C, k, seed=seed * 7 + 1) * (1.0 / np.sqrt(C * k))
    kw = dict(in0=rnd(B, T, C, seed=seed * 7 + 2), w=ops.pack_conv_weight(w), out=torch.zeros(B, To, N), n_batch=B,
              T_in=T, T_out=To, C_in=C, N=N, in0_bs=T * C, ld_in0=C, ldw=k * C, out_bs=To * N, ld_out=N, taps=taps,
              in_stride=stride, tile=pick([0, 0, 1, 2, 3, 4, 5, 7, 9]))
    if pick([0, 1]):
        kw["bias"] = rnd(N, seed=seed * 7 + 3)
    if C % 32 == 0 and C >= 64 and pick([0, 0, 1]):  # concat input: the K range comes from two tensors
        c0 = 32 * int(rng.integers(1, C // 32))
        full = kw["in0"]
        kw.update(in0=full[:, :, :c0].contiguous(), in0_bs=T * c0, ld_in0=c0, C_in0=c0,
                  in1=full[:, :, c0:].contiguous(), in1_bs=T * (C - c0), ld_in1=C - c0)
    if stride == 1 and k > 1 and (k - 1) // 2 * dil < T - 1 and pick([0, 0, 0, 1]):
        kw["reflect"] = True
    act = pick([_lib.ACT_NONE, _lib.ACT_NONE, _lib.ACT_LEAKY, _lib.ACT_LEAKY, _lib.ACT_SILU, _lib.ACT_MISH])
    if act != _lib.ACT_NONE:
        kw.update(pro_act=act, pro_slope=0.1)
    if pick([0, 1]) and not kw.get("reflect"):
        kw["len_in"] = torch.tensor([int(rng.integers(1, T + 1)) for _ in range(B)], dtype=torch.int32)
    if pick([0, 0, 1]):
        kw["len_out"] = torch.tensor([int(rng.integers(1, To + 1)) for _ in range(B)], dtype=torch.int32)
    mode = pick(["none", "none", "add", "axpy", "mean", "post"])
    if mode == "add":
        kw.update(res=rnd(B, To, N, seed=seed * 7 + 4), res_mode=_lib.RES_ADD, res_bs=To * N, ld_res=N)
    elif mode == "axpy":  # in place, like the Euler update
        out = rnd(B, To, N, seed=seed * 7 + 4)
        kw.update(out=out, res=out, res_mode=_lib.RES_AXPY, beta=0.25, res_bs=To * N, ld_res=N)
    elif mode == "mean":  # in place over the running stage sum, like HiFi-GAN
        acc = rnd(B, To, N, seed=seed * 7 + 4)
        kw.update(out=acc, res=rnd(B, To, N, seed=seed * 7 + 5), res_mode=_lib.RES_ADD, res_bs=To * N, ld_res=N,
                  res2=acc, res2_bs=To * N, ld_res2=N, post=_lib.POST_DIV, post_div=3.0)
    elif mode == "post":
        post = pick([_lib.POST_RELU, _lib.POST_LEAKY, _lib.POST_TANH])
        kw.update(post=post, post_div=0.2)
    elif N % 32 == 0 and pick([0, 1]):
        kw["gn_partials"] = torch.zeros(B, (To + 31) // 32, N // 32, 2)
    if pick([0, 0, 0, 1]) and mode in ("none", "add", "post") and "gn_partials" not in kw:
        s, r = 3, int(rng.integers(3))  # strided output rows (a transposed-conv phase); residuals are indexed like out
        kw.update(out=torch.zeros(B, To * s, N), out_bs=To * s * N, out_t_stride=s, out_t_off=r)
        if mode == "add":
            kw.update(res=rnd(B, To * s, N, seed=seed * 7 + 6), res_bs=To * s * N)
    return kw


@pytest.mark.parametrize("seed", range(28))
def test_conv_fuzz(dev, seed):
    run_conv_both(dev, make_case(seed))
