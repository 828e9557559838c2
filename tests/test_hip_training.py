"""SURVEY 8 f4 on the GPU: the estimator's training step (serenade_amd/training.py over libserenade_hip.so).

  * every backward kernel of csrc/train.hip against torch autograd of the op it differentiates;
  * loss + gradients of all 192 estimator parameters, d mu, d spks against the REFERENCE's own
    CFM.compute_loss + backward() (tests/golden/train_grads_L45.npz) and, at a larger ragged batch, against autograd
    through the CPU oracle;
  * bit-reproducibility, and three clipped AdamW steps against torch.optim.AdamW on the oracle's gradients.
Tolerances: gradients 2e-4 of each tensor's max (fp32 sums in a different order; the forward gate is 1e-5)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import serenade_oracle as O
from serenade_amd import _lib, training
from tests._weights import serenade_weights, sub
from tests.test_training_emulated import _case, _golden_case, _oracle_grads, check_against_reference_gradients, rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "the gpu-marked tests need an MI355X"
    _lib.lib()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def weights():
    return sub(serenade_weights(), "cfm_decoder.estimator.")


def _leaf(t, dev):
    return t.to(dev).requires_grad_(True)


def test_conv_backward(dev):
    """dgrad through srn_conv_gemm (transposed weights; stride 2 by output parity), wgrad / dbias, vs F.conv1d"""
    g = torch.Generator().manual_seed(1)
    for (B, T, C, N, k, stride) in [(2, 37, 64, 96, 3, 1), (3, 50, 32, 64, 3, 2), (2, 41, 128, 32, 3, 2),
                                    (1, 9, 244, 64, 1, 1), (2, 20, 80, 52, 1, 1)]:
        x = torch.randn(B, T, C, generator=g)
        w = torch.randn(N, C, k, generator=g) / math.sqrt(C * k)
        b = torch.randn(N, generator=g)
        xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        ref = F.conv1d(xr.transpose(1, 2), wr, br, stride=stride, padding=(k - 1) // 2).transpose(1, 2)
        dy = torch.randn(ref.shape, generator=g)
        ref.backward(dy)
        xd, wd, bd = _leaf(x, dev), _leaf(w, dev), _leaf(b, dev)
        taps = [j - (k - 1) // 2 for j in range(k)]
        y = training.conv1d(xd, training.pack_conv(wd), bd, taps, stride=stride)
        assert y.shape == ref.shape and rel(y.detach().cpu(), ref.detach()) < 1e-5
        y.backward(dy.to(dev))
        for a, r in ((xd, xr), (wd, wr), (bd, br)):
            assert rel(a.grad.cpu(), r.grad) < 1e-5, (B, T, C, N, k, stride)


def test_gn_mish_backward(dev):
    g = torch.Generator().manual_seed(2)
    B, T, C = 3, 45, 256
    lens = torch.tensor([45, 31, 38])
    x = torch.randn(B, T, 64, generator=g)
    w = torch.randn(C, 64 * 3, generator=g) / 14
    gam, bet = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    maskf = (torch.arange(T)[None] < lens[:, None]).float().unsqueeze(-1)
    dy = torch.randn(B, T, C, generator=g)
    xr, wr, gr, br = (t.clone().requires_grad_(True) for t in (x, w, gam, bet))
    h = F.conv1d((xr * maskf).transpose(1, 2), wr.view(C, 3, 64).permute(0, 2, 1), None, padding=1)
    ref = (F.mish(F.group_norm(h, 8, gr, br, 1e-5)) * maskf.transpose(1, 2)).transpose(1, 2)
    ref.backward(dy)
    xd, wd, gd, bd = (_leaf(t, dev) for t in (x, w, gam, bet))
    hh, part = training.conv1d(xd * maskf.to(dev), wd, None, [-1, 0, 1], want_gn=True)
    y = training.gn_mish(hh, part, gd, bd, lens.to(dev).to(torch.int32))
    assert rel(y.detach().cpu(), ref.detach()) < 1e-5
    y.backward(dy.to(dev))
    for a, r in ((xd, xr), (wd, wr), (gd, gr), (bd, br)):
        assert rel(a.grad.cpu(), r.grad) < 2e-5


def test_row_layernorm_backward(dev):
    g = torch.Generator().manual_seed(3)
    B, T, C = 3, 70, 512
    x, dy = torch.randn(B, T, C, generator=g) * 2 + 0.3, torch.randn(B, T, C, generator=g)
    for per_b in (False, True):
        m = torch.randn((B, C) if per_b else (C,), generator=g)
        a = torch.randn((B, C) if per_b else (C,), generator=g)
        xr, mr, ar = (t.clone().requires_grad_(True) for t in (x, m, a))
        xh = F.layer_norm(xr, (C,), None, None, 1e-5)
        ref = xh * (mr.unsqueeze(1) if per_b else mr) + (ar.unsqueeze(1) if per_b else ar)
        ref.backward(dy)
        xd, md, ad = (_leaf(t, dev) for t in (x, m, a))
        y = training.row_ln(xd, md, ad)
        assert rel(y.detach().cpu(), ref.detach()) < 1e-5
        y.backward(dy.to(dev))
        for u, r in ((xd, xr), (md, mr), (ad, ar)):
            assert rel(u.grad.cpu(), r.grad) < 2e-5, per_b


def test_attention_core_backward(dev):
    g = torch.Generator().manual_seed(4)
    B, L, H, hd = 2, 45, 4, 64
    lens = torch.tensor([45, 30])
    qkv = torch.randn(B, L, 3 * H * hd, generator=g)
    do = torch.randn(B, L, H * hd, generator=g)
    r = qkv.clone().requires_grad_(True)
    q, k, v = (t.view(B, L, H, hd).transpose(1, 2) for t in r.chunk(3, dim=-1))
    s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
    s = s.masked_fill(~(torch.arange(L)[None] < lens[:, None]).view(B, 1, 1, L), float("-inf"))
    ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, L, H * hd)
    ref.backward(do)
    d = _leaf(qkv, dev)
    o = training.attention_core(d, lens.to(dev).to(torch.int32), H)
    assert rel(o.detach().cpu(), ref.detach()) < 1e-5
    o.backward(do.to(dev))
    assert rel(d.grad.cpu(), r.grad) < 2e-5


def test_geglu_backward(dev):
    g = torch.Generator().manual_seed(5)
    hg, da = torch.randn(3, 17, 256, generator=g) * 2, torch.randn(3, 17, 128, generator=g)
    r = hg.clone().requires_grad_(True)
    h, gate = r.chunk(2, dim=-1)
    ref = h * F.gelu(gate)
    ref.backward(da)
    d = _leaf(hg, dev)
    a = training.geglu(d)
    assert rel(a.detach().cpu(), ref.detach()) < 1e-6
    a.backward(da.to(dev))
    assert rel(d.grad.cpu(), r.grad) < 1e-5


def _run(est, case, dev):
    x1, mask, mu, spk, mask_l, t, z = (c.to(dev) for c in case)
    mu_r, spk_r = mu.clone().requires_grad_(True), spk.clone().requires_grad_(True)
    est.zero_grad()
    loss, _ = training.cfm_loss(est, x1, mask, mu_r, spk_r, mask_l, draws={"t": t, "z": z})
    loss.backward()
    torch.cuda.synchronize()
    return loss.detach().cpu(), {k: v.grad.detach().cpu().clone() for k, v in est.params.items()}, mu_r.grad.cpu(), spk_r.grad.cpu()


def test_gradients_match_the_reference(dev, weights, golden):
    """the reference's own compute_loss + backward() (3 utterances, L = 45 odd, padded, infill segment 9..27)"""
    g = golden("train_grads_L45")
    est = training.Estimator(weights, dev)
    loss, grads, dmu, dspk = _run(est, _golden_case(g), dev)
    check_against_reference_gradients(g, loss, grads, dmu, dspk)
    again = _run(est, _golden_case(g), dev)
    assert torch.equal(again[0], loss) and all(torch.equal(again[1][k], grads[k]) for k in grads)  # bit-reproducible


def test_gradients_match_oracle_autograd_larger_batch(dev, weights):
    case = _case(B=4, L=203, lens=(203, 150, 97, 180), seed=9)
    ref_loss, ref_g, ref_dmu, ref_dspk = _oracle_grads(weights, case)
    est = training.Estimator(weights, dev)
    loss, grads, dmu, dspk = _run(est, case, dev)
    assert abs(loss.item() - ref_loss.item()) < 1e-5 * abs(ref_loss.item())
    worst = max((rel(grads[k], ref_g[k]), k) for k in ref_g)
    assert worst[0] < 2e-4, worst
    assert rel(dmu, ref_dmu) < 2e-4 and rel(dspk, ref_dspk) < 2e-4


def test_three_training_steps_match_torch_adamw(dev, weights):
    """loss.backward -> clip_grad_norm_(1.0) -> AdamW(lr 8e-4), three times, against torch on the oracle's autograd"""
    case = _case(B=2, L=40, lens=(40, 29), seed=21)
    ref = {k: torch.nn.Parameter(v.clone()) for k, v in weights.items()}
    opt_ref = torch.optim.AdamW(ref.values(), lr=8e-4)
    est = training.Estimator(weights, dev)
    opt = training.AdamW(est, lr=8e-4, max_grad_norm=1.0)
    x1, mask, mu, spk, mask_l, t, z = case
    losses = []
    for step in range(3):
        opt_ref.zero_grad()
        l_ref, _ = O.cfm_compute_loss(ref, x1, mask, mu, spk, mask_l, t, z)
        l_ref.backward()
        n_ref = torch.nn.utils.clip_grad_norm_(ref.values(), 1.0)
        opt_ref.step()
        loss, _, _, _ = _run(est, case, dev)
        n = opt.step()
        assert abs(loss.item() - l_ref.item()) < 2e-4 * abs(l_ref.item()), step
        assert abs(n - float(n_ref)) < 1e-3 * float(n_ref), step
        losses.append(loss.item())
    assert losses[2] < losses[0]
    # Adam's first steps move every weight by ~lr regardless of the gradient's size, so compare the UPDATES
    worst = max((rel(est.params[k].detach().cpu() - weights[k], ref[k].detach() - weights[k]), k) for k in weights)
    assert worst[0] < 5e-2, worst


def test_whole_model_gradients_match_the_reference(dev, golden):
    """Serenade.forward in train() mode (BatchNorm on batch statistics, dropout off) + backward(), all 262 trainable
    tensors: encoder (weight-norm, reflection padding), GST (library path), estimator, both losses"""
    from tests.test_training_emulated import _full_case, check_whole_model
    g = golden("train_full_T64")
    c = _full_case(g)
    model = training.TrainSerenade(serenade_weights(), dev, dropout=0.0)
    d = dict(c["draws"], t=c["draws"]["t"].to(dev), z=c["draws"]["z"].to(dev))
    ret = model(c["x"].to(dev), c["lengths"].to(dev), c["logmel"].to(dev), c["midi"].to(dev), c["lft"].to(dev), draws=d)
    (ret["cfm_loss"] + ret["prior_loss"]).backward()
    torch.cuda.synchronize()
    check_whole_model(g, {k: v.detach().cpu() for k, v in ret.items()},
                      {k: v.grad.detach().cpu() for k, v in model.params.items()}, gst_conv_tol=2e-3)
    assert rel(model.buffers["gst.ref_enc.convs.16.running_var"].cpu(), torch.from_numpy(g["bn_var"])) < 1e-4


def test_gst_eval_mode_runs_on_the_running_statistics(dev):
    """with `training = False` (the reference's _eval_epoch calls model.eval(), trainers/base.py:171-190) the
    training model's GST normalises with BatchNorm's RUNNING statistics -- the same numbers the inference StyleEncoder
    produces from the same state_dict -- and leaves those statistics alone"""
    from serenade_amd import models
    from serenade_amd.utils.synth import SERENADE_PARAMS
    w = serenade_weights()
    tm = training.TrainSerenade(w, dev, dropout=0.0)
    inf = models.Serenade(**SERENADE_PARAMS)
    inf.load_state_dict(w)
    inf = inf.eval().to(dev)
    speech = torch.randn(2, 64, 80, generator=torch.Generator().manual_seed(3)).to(dev)
    before = tm.buffers["gst.ref_enc.convs.1.running_mean"].clone()
    tm.training = False
    with torch.no_grad():
        got = tm.gst(speech)
    tm.training = True
    ref = inf.gst(speech)
    assert torch.equal(before, tm.buffers["gst.ref_enc.convs.1.running_mean"])
    assert rel(got.cpu(), ref.cpu()) < 2e-5


def test_training_loop_with_dropout_runs_and_learns(dev):
    """the real configuration (dropout 0.05, random draws): five whole-model steps on one batch -- finite, the loss
    falls, parameters of all three modules move, state_dict() round-trips into the inference model"""
    from serenade_amd import models
    from serenade_amd.utils.synth import SERENADE_PARAMS
    torch.manual_seed(0)
    import random
    random.seed(0)
    w = serenade_weights()
    model = training.TrainSerenade(w, dev, dropout=0.05)
    sync, opt = training.GradSync(model), training.AdamW(model, lr=2e-4, max_grad_norm=1.0)
    B, T = 4, 96
    g = torch.Generator().manual_seed(1)
    x, logmel = torch.randn(B, T, 768, generator=g).to(dev), torch.randn(B, T, 80, generator=g).to(dev)
    midi, lft = torch.randn(B, T, 1, generator=g).to(dev), torch.randn(B, T, 1, generator=g).to(dev)
    lens = torch.tensor([96, 80, 70, 96]).to(dev)
    # the same (t, z, infill segment) every step, so that the loss sequence measures the optimisation, not the draws;
    # dropout stays random
    fixed = {"seg": torch.tensor([20, 40]).to(dev), "t": torch.rand(B, 1, 1, generator=g).to(dev),
             "z": torch.randn(B, 80, T, generator=g).to(dev)}
    losses = []
    for it in range(5):
        ret = model(x, lens, logmel, midi, lft, draws=fixed)
        loss = ret["cfm_loss"] + ret["prior_loss"]
        if it % 2:  # both gradient paths: accumulate into zeroed views / overwrite by one multi-tensor copy
            model.zero_grad()
            loss.backward()
            sync.finish()
        else:
            model.backward(loss, sync)
        n = opt.step()
        assert math.isfinite(n) and math.isfinite(loss.item())
        losses.append(loss.item())
    assert losses[-1] < losses[0]
    sd = model.state_dict()
    for k in ("encoder.model.1.weight_v", "gst.stl.gst_embs", "cfm_decoder.estimator.final_proj.weight"):
        assert not torch.equal(sd[k].cpu(), w[k])
    inf = models.Serenade(**SERENADE_PARAMS)
    inf.load_state_dict({k: v.cpu() for k, v in sd.items()})


_RCCL_ONE = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)   # before any other GPU call
from serenade_amd import training
g = torch.Generator().manual_seed(3)
sd = {f"p{i}": torch.randn(s, generator=g) for i, s in enumerate([(700, 50), (3300,), (640, 90), (30,), (1280, 40)])}
est = training.ParamStore(sd, dev)
sync = training.GradSync(est, bucket_bytes=100000, always=True)
coef = {k: torch.randn(v.shape, generator=g).to(dev) for k, v in sd.items()}
for step in range(2):
    est.zero_grad()
    sum((est.params[k] * coef[k]).sum() for k in sd).backward()
    sync.finish()
    torch.cuda.synchronize()
    assert all(torch.equal(est.params[k].grad, coef[k]) for k in sd)
assert len(sync.buckets) >= 3 and sync.launched == 2 * len(sync.buckets), (sync.buckets, sync.launched)
dist.barrier(); dist.destroy_process_group()
print("RCCL_OK", len(sync.buckets), dist.get_backend() if dist.is_initialized() else "nccl")
"""


def test_gradient_allreduce_over_rccl_group_of_one(dev):
    """the DDP replacement's collective path on the GPU: a one-rank RCCL group, bucketed asynchronous all-reduce of
    the flat gradient buffer launched from the backward hooks (N > 1 logic: gloo world-size-2 test on the CPU)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _RCCL_ONE], env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stderr[-2000:]


@pytest.mark.parametrize("B,L,lrs", [(4, 512, (0.0, 8e-4)), (8, 1024, (0.0,))])
def test_captured_step_equals_eager_step(dev, B, L, lrs):
    """GraphedStep: the whole-model step captured as a hipGraph (device-side infill segment, clip factor and Adam bias
    corrections).  (a) With the weights frozen (lr = 0) every replay reproduces the eager step's losses and ALL
    gradients on three different batches -- torch's own multi-block reductions returned stale values from the second
    replay on, which is why the step routes them through rocBLAS / srn_sumsq; (b) with the optimizer on, losses and
    gradient norms track the eager run."""
    w = serenade_weights()  # sizes large enough for multi-block reductions
    g = torch.Generator().manual_seed(8)
    lens0 = torch.tensor([L - (L // 5) * (i % 3) for i in range(B)])
    batches = [[torch.randn(B, L, 768, generator=g).to(dev), lens0.to(dev),
                torch.randn(B, L, 80, generator=g).to(dev), torch.randn(B, L, 1, generator=g).to(dev),
                torch.randn(B, L, 1, generator=g).to(dev)] for _ in range(3)]
    t, z = torch.rand(B, 1, 1, generator=g).to(dev), torch.randn(B, 80, L, generator=g).to(dev)
    segs = [(L // 5, 2 * L // 5), (0, L // 2 - 1), (3 * L // 5, L // 8)]
    for lr in lrs:
        eager = training.TrainSerenade(w, dev, dropout=0.0)
        opt_e = training.AdamW(eager, lr=lr, weight_decay=0.0 if lr == 0.0 else 0.01)
        cap = training.TrainSerenade(w, dev, dropout=0.0)
        opt_c = training.AdamW(cap, lr=lr, weight_decay=0.0 if lr == 0.0 else 0.01)
        start = cap.flat.clone()
        step = training.GraphedStep(cap, opt_c, B, L, tz=(t, z), warmup=1)
        # building the captured step (warm-up steps on zero inputs) leaves weights, optimizer state and BatchNorm
        # statistics alone
        assert torch.equal(cap.flat, start) and opt_c.steps == 0 and not opt_c.m.any()
        assert all(torch.equal(cap.buffers[k].cpu(), w[k]) for k in cap.buffers)
        for i, ((x, lens, mel, midi, lft), seg) in enumerate(zip(batches, segs)):
            eager.zero_grad()
            ret = eager(x, lens, mel, midi, lft, draws={"seg": torch.tensor(seg).to(dev), "t": t, "z": z})
            (ret["cfm_loss"] + ret["prior_loss"]).backward()
            n_ref = opt_e.step()
            cfm, prior, norm = step(x, lens, mel, midi, lft, segment=seg)
            torch.cuda.synchronize()
            tol = 1e-5 if (lr == 0.0 or i == 0) else 2e-4
            assert abs(cfm.item() - ret["cfm_loss"].item()) < tol * abs(cfm.item()), (lr, i)
            assert abs(prior.item() - ret["prior_loss"].item()) < tol * abs(prior.item()), (lr, i)
            assert abs(float(norm) - n_ref) < 10 * tol * n_ref, (lr, i)
            if lr == 0.0:
                for k in cap.params:
                    ref = eager.params[k].grad
                    if ref.abs().max() > 1e-7:  # not the mathematically-zero key bias of the token attention
                        # MIOpen's conv2d weight gradients (the GST's library path) are not run-to-run reproducible
                        tol_k = 2e-3 if k.startswith("gst.ref_enc.convs.") else 1e-4
                        assert rel(cap.params[k].grad.cpu(), ref.cpu()) < tol_k, (i, k)
        if lr == 0.0:
            assert torch.equal(cap.flat, start)


# ---- the cause of round 2's "stale reductions under hipGraph" (ADVICE r2): captured memset nodes
def _replay3(build):
    g = torch.cuda.CUDAGraph(keep_graph=True)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        build()
    return g


@pytest.mark.xfail(strict=False, reason="ROCm 7.2 / torch 2.10+rocm7.0: a hipMemsetAsync node captured into a hipGraph "
                                        "fills with garbage from the SECOND replay on (profiles/r3_graph_probe.json); "
                                        "an XPASS means the stack was fixed and count_memset_nodes' guard can go")
def test_captured_memset_node_replays_correctly(dev):
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    bufs = [torch.full((n,), 7, device=dev, dtype=torch.uint8) for n in (64, 256, 4096)]

    def build():
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for b in bufs:
            assert hip.hipMemsetAsync(ctypes.c_void_p(b.data_ptr()), 0, ctypes.c_size_t(b.numel()), st) == 0
            b.add_(1)

    g = _replay3(build)
    assert training.count_memset_nodes(g)[0] == 3
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        assert all(bool((b == 1).all()) for b in bufs)


def test_memset_guard_sees_torch_reductions_and_clears_the_product_step(dev):
    """a plain multi-block tensor.sum() inside a capture is a memset node (torch zeroes the reduction's semaphores with
    hipMemsetAsync) -- the guard counts it; single-block reductions and this library's kernels are not; GraphedStep
    checks its own captures with the same function (it would raise) and holds none"""
    x = torch.randn(1 << 24, device=dev)
    y = torch.randn(256, device=dev)
    out = {}
    x.sum(), y.sum()
    g_big = _replay3(lambda: out.__setitem__("big", x.sum()))
    g_small = _replay3(lambda: out.__setitem__("small", y.sum()))
    assert training.count_memset_nodes(g_big)[0] >= 1
    assert training.count_memset_nodes(g_small) == (0, 1)
    ss = torch.zeros(1024, device=dev, dtype=torch.float64)

    def own():  # what AdamW._grad_norm does: srn_sumsq partial sums + a single-block sum of 1024 doubles
        training._call("srn_sumsq", x, x.numel(), ss)
        out["own"] = ss.sum()

    own()
    g_own = _replay3(own)
    assert training.count_memset_nodes(g_own)[0] == 0
    for _ in range(3):  # and the own reduction replays correctly on changing data
        x.normal_()
        g_own.replay()
        torch.cuda.synchronize()
        ref = float((x.double() ** 2).sum())
        assert abs(float(out["own"]) - ref) < 1e-6 * ref
    w = serenade_weights()
    model = training.TrainSerenade(w, dev, dropout=0.05)
    step = training.GraphedStep(model, training.AdamW(model), 2, 256, warmup=1)
    assert step.nodes > 500 and training.count_memset_nodes(step.g1)[0] == 0
    with pytest.raises(ValueError):
        training.GraphedStep(model, training.AdamW(model), 2, 256, warmup=0)


# ---- srn_tn_gemm: contractions over time (weight gradients, dK, dV) against fp64 torch
@pytest.mark.parametrize("case", ["wgrad_k3", "wgrad_stride2", "wgrad_valid", "wgrad_tiny_k", "attn", "ragged_m"])
def test_tn_gemm_against_fp64(dev, case):
    from serenade_amd.ops import TnGemmOp
    g = torch.Generator().manual_seed(hash(case) % 1000)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    if case.startswith("wgrad"):
        B, T, C, N, taps, stride, T_out = {"wgrad_k3": (4, 515, 256, 512, (-1, 0, 1), 1, 515),
                                           "wgrad_stride2": (3, 301, 128, 96, (-1, 0, 1), 2, 151),
                                           "wgrad_valid": (2, 70, 64, 80, (0, 2, 4, 6), 1, 64),
                                           "wgrad_tiny_k": (3, 1, 2048, 512, (0,), 1, 1)}[case]
        x, dy = r(B, T, C), r(B, T_out, N)
        # second pass: the same contraction with per-item lengths on x (rows past them count as zero) and the bias
        # gradient (column sums of dY) riding along
        lens_cpu = torch.tensor([max(1, T - 7 * i - 3) for i in range(B)], dtype=torch.int32)
        for lens in (None, lens_cpu.to(dev)):
            dw = torch.full((N, len(taps) * C), float("nan"), device=dev)
            db = None if lens is None else torch.full((N,), float("nan"), device=dev)
            kw = dict(a=dy, b=x, n_items=B, T_a=T_out, T_b=T, M=N, N=C, lda=N, ldb=C, ldc=len(taps) * C, shifts=taps,
                      stride=stride, a_is=T_out * N, b_is=T * C, len_b=lens, colsum=db)
            TnGemmOp(out=dw, **kw)()
            ref = torch.zeros(N, len(taps), C, dtype=torch.float64)
            xd, dyd = x.double().cpu(), dy.double().cpu()
            if lens is not None:
                xd = xd * (torch.arange(T)[None, :, None] < lens_cpu[:, None, None])
            for j, o in enumerate(taps):
                for t in range(T_out):
                    tb = t * stride + o
                    if 0 <= tb < T:
                        ref[:, j] += torch.einsum("bn,bc->nc", dyd[:, t], xd[:, tb])
            got = dw.cpu().double().view(N, len(taps), C)
            assert torch.isfinite(got).all()
            assert (got - ref).abs().max() < 2e-5 * ref.abs().max(), case
            if db is not None:
                cs = dyd.sum((0, 1))
                assert (db.cpu().double() - cs).abs().max() < 2e-5 * max(1.0, cs.abs().max()), case
            # bit-reproducible (slices are added in order)
            dw2 = torch.empty_like(dw)
            db2 = None if db is None else torch.empty_like(db)
            kw["colsum"] = db2
            TnGemmOp(out=dw2, **kw)()
            assert torch.equal(dw, dw2) and (db is None or torch.equal(db, db2))
    else:
        B, H, L, hd = (2, 4, 256, 64) if case == "attn" else (2, 2, 203, 32)
        Lp = (L + 31) // 32 * 32
        P = r(B, H, L, Lp)
        do = r(B, L, H * hd)
        out = torch.zeros(B, L, 3 * H * hd, device=dev)
        TnGemmOp(a=P, b=do, out=(out, 2 * H * hd), n_items=1, T_a=L, T_b=L, M=L, N=hd, lda=Lp, ldb=H * hd, ldc=3 * H * hd,
                 n_batch=B, n_head=H, a_bs=H * L * Lp, a_hs=L * Lp, b_bs=L * H * hd, b_hs=hd, out_bs=L * 3 * H * hd,
                 out_hs=hd, alpha=0.5)()
        ref = 0.5 * torch.einsum("bhij,bihd->bjhd", P[..., :L].double().cpu(), do.double().cpu().view(B, L, H, hd))
        got = out.cpu().double()[:, :, 2 * H * hd:].view(B, L, H, hd)
        assert (got - ref).abs().max() < 2e-5 * ref.abs().max()
        assert not out[:, :, :2 * H * hd].any()  # nothing outside the head slices was touched


@pytest.mark.parametrize("case", ["wgrad_k3", "wgrad_stride2", "attn"])
def test_tn_gemm_scalar_walk_form_is_bit_identical(dev, case, monkeypatch):
    """slabs that lie inside one item (T_a % 16 == 0) take tn_lean_kernel (operand walk on the scalar unit, buffer loads,
    out-of-range rows as zeros); SRN_TN_GENERAL=1 keeps the general kernel: same LDS image and MFMA order, so the same
    bits -- with item lengths, the bias gradient riding along, stride 2 and head strides"""
    from serenade_amd.ops import TnGemmOp
    g = torch.Generator().manual_seed(len(case))
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    outs = []
    if case.startswith("wgrad"):
        B, T, C, N, taps, stride, T_out = {"wgrad_k3": (4, 512, 256, 512, (-1, 0, 1), 1, 512),
                                           "wgrad_stride2": (3, 320, 128, 96, (-1, 0, 1), 2, 160)}[case]
        x, dy = r(B, T, C), r(B, T_out, N)
        lens = torch.tensor([max(1, T - 37 * i - 3) for i in range(B)], dtype=torch.int32).to(dev)
        for general in ("0", "1"):
            monkeypatch.setenv("SRN_TN_GENERAL", general)
            dw = torch.full((N, len(taps) * C), float("nan"), device=dev)
            db = torch.full((N,), float("nan"), device=dev)
            TnGemmOp(out=dw, a=dy, b=x, n_items=B, T_a=T_out, T_b=T, M=N, N=C, lda=N, ldb=C, ldc=len(taps) * C, shifts=taps,
                     stride=stride, a_is=T_out * N, b_is=T * C, len_b=lens, colsum=db)()
            torch.cuda.synchronize()
            outs.append((dw, db))
        assert torch.isfinite(outs[0][0]).all() and torch.isfinite(outs[0][1]).all()
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    else:
        B, H, L, hd = 2, 4, 256, 64
        P, do = r(B, H, L, L), r(B, L, H * hd)
        for general in ("0", "1"):
            monkeypatch.setenv("SRN_TN_GENERAL", general)
            out = torch.zeros(B, L, 3 * H * hd, device=dev)
            TnGemmOp(a=P, b=do, out=(out, 2 * H * hd), n_items=1, T_a=L, T_b=L, M=L, N=hd, lda=L, ldb=H * hd, ldc=3 * H * hd,
                     n_batch=B, n_head=H, a_bs=H * L * L, a_hs=L * L, b_bs=L * H * hd, b_hs=hd, out_bs=L * 3 * H * hd,
                     out_hs=hd, alpha=0.5)()
            torch.cuda.synchronize()
            outs.append(out)
        assert outs[0].abs().max() > 0 and torch.equal(outs[0], outs[1])


# ---- fused weight norm + re-layout (the content encoder's convs) against torch autograd of the same expression
@pytest.mark.parametrize("shape", [(96, 64, 3), (80, 768, 7), (512, 512, 1)])
def test_weight_norm_pack_against_torch(dev, shape):
    n, c, k = shape
    g0 = torch.Generator().manual_seed(n + k)
    v = torch.randn(n, c, k, generator=g0).to(dev).requires_grad_(True)
    g = (torch.rand(n, 1, 1, generator=g0) + 0.5).to(dev).requires_grad_(True)
    co = torch.randn(n, k * c, generator=g0).to(dev)
    w, wd = training._WeightNormPack.apply(v, g)
    (w * co).sum().backward()
    v2, g2 = v.detach().clone().double().requires_grad_(True), g.detach().clone().double().requires_grad_(True)
    ref = (v2 * (g2 / v2.reshape(n, -1).norm(dim=1).reshape(n, 1, 1))).permute(0, 2, 1).reshape(n, k * c)
    (ref * co.double()).sum().backward()
    assert (w.double() - ref).abs().max() < 1e-6 * ref.abs().max()
    assert torch.equal(wd.view(c, k, n), w.view(n, k, c).permute(2, 1, 0))  # W^T of the same numbers
    assert (v.grad.double() - v2.grad).abs().max() < 2e-5 * v2.grad.abs().max()
    assert (g.grad.double() - g2.grad).abs().max() < 2e-5 * g2.grad.abs().max()
