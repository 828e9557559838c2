"""SURVEY 8 f4 on the CPU: the estimator's training step (serenade_amd/training.py) through the kernel emulator,
against autograd through the CPU oracle's `cfm_compute_loss` (flow_matching.py:95-133 over decoder.py) on the same
weights, inputs and random draws: the loss value and all 192 parameter gradients, d mu and d spks; then one clipped
AdamW step against torch.optim.AdamW + clip_grad_norm_ (trainers/ssc.py:86-96)."""
import math

import numpy as np
import pytest
import torch

from oracle import serenade_oracle as O
from serenade_amd import training
from tests import _emulator
from tests._weights import serenade_weights, sub


def _case(B=2, L=24, lens=(24, 17), seed=5):
    g = torch.Generator().manual_seed(seed)
    x1 = torch.randn(B, 80, L, generator=g)
    mu = torch.randn(B, 162, L, generator=g)
    spk = torch.randn(B, 256, generator=g)
    mask = O.make_non_pad_mask(list(lens)).unsqueeze(1).float()
    mask_l = mask.clone()
    mask_l[:, :, :5] = 0
    mask_l[:, :, 14:] = 0
    t = torch.rand(B, 1, 1, generator=g)
    z = torch.randn(B, 80, L, generator=g)
    return x1 * mask, mask, mu * mask, spk, mask_l, t, z


def _oracle_grads(w, case):
    x1, mask, mu, spk, mask_l, t, z = case
    wr = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    mu_r, spk_r = mu.clone().requires_grad_(True), spk.clone().requires_grad_(True)
    loss, _ = O.cfm_compute_loss(wr, x1, mask, mu_r, spk_r, mask_l, t, z)
    loss.backward()
    return loss.detach(), {k: v.grad for k, v in wr.items()}, mu_r.grad, spk_r.grad


def rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


def test_estimator_gradients_match_oracle_autograd():
    w = sub(serenade_weights(), "cfm_decoder.estimator.")
    case = _case()
    ref_loss, ref_g, ref_dmu, ref_dspk = _oracle_grads(w, case)
    x1, mask, mu, spk, mask_l, t, z = case
    with _emulator.installed():
        est = training.Estimator(w, torch.device("cpu"))
        mu_r, spk_r = mu.clone().requires_grad_(True), spk.clone().requires_grad_(True)
        loss, _ = training.cfm_loss(est, x1, mask, mu_r, spk_r, mask_l, draws={"t": t, "z": z})
        loss.backward()
    assert abs(loss.item() - ref_loss.item()) < 1e-5 * abs(ref_loss.item())
    assert set(est.params) == set(ref_g)
    worst = max((rel(est.params[k].grad, ref_g[k]), k) for k in ref_g)
    assert worst[0] < 2e-4, worst
    assert rel(mu_r.grad, ref_dmu) < 2e-4 and rel(spk_r.grad, ref_dspk) < 2e-4
    # the gradients live in the flat buffer the optimizer and the all-reduce work on
    off, n = est.spans["final_proj.weight"]
    assert torch.equal(est.flat_grad[off:off + n], est.params["final_proj.weight"].grad.reshape(-1))


def test_clipped_adamw_step_matches_torch():
    w = sub(serenade_weights(), "cfm_decoder.estimator.")
    w = {k: w[k] for k in list(w)[:12]}  # a dozen tensors are enough for the update rule
    g = torch.Generator().manual_seed(0)
    ref = {k: torch.nn.Parameter(v.clone()) for k, v in w.items()}
    opt_ref = torch.optim.AdamW(ref.values(), lr=8e-4)
    with _emulator.installed():
        est = training.Estimator(w, torch.device("cpu"))
        opt = training.AdamW(est, lr=8e-4, max_grad_norm=1.0)
        for _ in range(3):
            for k in w:
                gk = torch.randn(w[k].shape, generator=g) * 3.0
                est.params[k].grad.copy_(gk)
                ref[k].grad = gk.clone()
            n_ref = torch.nn.utils.clip_grad_norm_(ref.values(), 1.0)
            opt_ref.step()
            if _ == 1:  # the capturable form: scalars on the device, no host synchronisation
                opt.prepare()
                opt.step_captured()
                n = float(opt.norm)
            else:
                n = opt.step()
            assert abs(n - float(n_ref)) < 1e-4 * float(n_ref)
    for k in w:
        assert rel(est.params[k], ref[k]) < 1e-6, k


def _golden_case(g):
    t = lambda k: torch.from_numpy(np.asarray(g[k]))
    mask = O.make_non_pad_mask(g["lens"].tolist()).unsqueeze(1).float()
    return t("x1"), mask, t("mu"), t("spk"), t("mask_l"), t("t"), t("z")


def check_against_reference_gradients(g, loss, grads, dmu, dspk, tol=2e-4):
    """`g` = tests/golden/train_grads_L45.npz: the reference's own CFM.compute_loss + backward()"""
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    for key in g:
        if key.startswith("g:"):
            assert rel(grads[key[2:]], torch.from_numpy(g[key])) < tol, key
        elif key.startswith("r:"):
            assert rel(grads[key[2:]][:4], torch.from_numpy(g[key])) < tol, key
    total = math.sqrt(sum(float((v.double() ** 2).sum()) for v in grads.values()))
    assert abs(total - float(g["grad_norm"])) < 1e-4 * float(g["grad_norm"])
    assert rel(dmu, torch.from_numpy(g["dmu"])) < tol and rel(dspk, torch.from_numpy(g["dspk"])) < tol


def test_reference_gradient_fixture(golden):
    """the reference's gradients pin (a) autograd through the oracle and (b) the HIP training step's host logic"""
    g = golden("train_grads_L45")
    w = sub(serenade_weights(), "cfm_decoder.estimator.")
    case = _golden_case(g)
    loss, grads, dmu, dspk = _oracle_grads(w, case)
    check_against_reference_gradients(g, loss, grads, dmu, dspk)
    x1, mask, mu, spk, mask_l, t, z = case
    with _emulator.installed():
        est = training.Estimator(w, torch.device("cpu"))
        mu_r, spk_r = mu.clone().requires_grad_(True), spk.clone().requires_grad_(True)
        loss, _ = training.cfm_loss(est, x1, mask, mu_r, spk_r, mask_l, draws={"t": t, "z": z})
        loss.backward()
    check_against_reference_gradients(g, loss.detach(), {k: v.grad for k, v in est.params.items()}, mu_r.grad,
                                      spk_r.grad)


# ---- the whole model: Serenade.forward in train mode (BatchNorm on batch statistics, dropout off) ------------------
def _full_case(g):
    t = lambda k: torch.from_numpy(np.asarray(g[k]))
    return dict(x=t("x"), lengths=torch.from_numpy(g["lens"]), logmel=t("logmel"), midi=t("midi"), lft=t("lft"),
                draws={"uniform": float(g["uniform"]), "seg_start": int(g["seg_start"]), "t": t("t"), "z": t("z")})


def check_whole_model(g, ret, grads, tol=3e-4, gst_conv_tol=None):
    for k in ("cfm_loss", "prior_loss"):
        assert abs(float(ret[k].detach()) - float(g[k])) < 2e-5 * abs(float(g[k])), k
    assert rel(ret["gauss_mel"].detach().cpu(), torch.from_numpy(g["gauss_mel"])) < 1e-4
    assert len(grads) == int(g["n_params"])
    for key in g:
        if key[:2] not in ("g:", "r:"):
            continue
        ref = torch.from_numpy(g[key])
        got = grads[key[2:]] if key[0] == "g" else grads[key[2:]][:4]
        if ref.abs().max() < 1e-8:  # mathematically zero (the key bias of the token attention shifts every score alike)
            assert got.abs().max() < 1e-8, key
        else:
            # gst_conv_tol (GPU): MIOpen's Conv2d weight gradients are not run-to-run reproducible at the 1e-4 level
            t = gst_conv_tol if (gst_conv_tol and key[2:].startswith("gst.ref_enc.convs.")) else tol
            assert rel(got, ref) < t, key
    total = math.sqrt(sum(float((v.double() ** 2).sum()) for v in grads.values()))
    assert abs(total - float(g["grad_norm"])) < 2e-4 * float(g["grad_norm"])


def test_whole_model_gradients_match_the_reference(golden):
    """tests/golden/train_full_T64.npz = the reference's Serenade.forward in train() mode + backward(): pins the
    oracle's autograd (bn_training) and the training model's host logic (encoder weight-norm / reflection padding, GST,
    infill masks, both losses) through the emulator"""
    g = golden("train_full_T64")
    w = serenade_weights()
    c = _full_case(g)
    d = c["draws"]
    wr = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and not k.endswith(("running_mean", "running_var"))
              else v) for k, v in w.items()}
    ret = O.serenade_forward(wr, c["x"], c["lengths"].tolist(), c["logmel"], c["midi"], c["lft"], d["uniform"],
                             d["seg_start"], d["t"], d["z"], bn_training=True)
    (ret["cfm_loss"] + ret["prior_loss"]).backward()
    check_whole_model(g, ret, {k: v.grad for k, v in wr.items() if v.requires_grad})
    with _emulator.installed():
        model = training.TrainSerenade(w, torch.device("cpu"), dropout=0.0)
        ret = model(c["x"], c["lengths"], c["logmel"], c["midi"], c["lft"], draws=d)
        (ret["cfm_loss"] + ret["prior_loss"]).backward()
    check_whole_model(g, ret, {k: v.grad for k, v in model.params.items()})
    # BatchNorm running statistics moved like nn.BatchNorm2d's (momentum 0.1)
    assert rel(model.buffers["gst.ref_enc.convs.16.running_mean"], torch.from_numpy(g["bn_mean"])) < 1e-4
    assert rel(model.buffers["gst.ref_enc.convs.16.running_var"], torch.from_numpy(g["bn_var"])) < 1e-4


def test_checkpoint_resume_and_lr_schedule(tmp_path):
    """trainers/base.py:91-130 layout: train 2 steps, save, train 2 more; a fresh model + optimizer resumed from the
    file reproduces those 2 steps bit for bit; MultiStepLR halves the rate at its milestone"""
    w = serenade_weights()
    keep = [k for k in w if not k.startswith("cfm_decoder.estimator.")]  # encoder + GST are enough for the file format
    small = {k: w[k] for k in keep}

    def fake_step(model, opt, sched, i):
        model.zero_grad()
        gg = torch.Generator().manual_seed(100 + i)
        for p in model.params.values():
            p.grad.copy_(torch.randn(p.shape, generator=gg))
        opt.step()
        sched.step()

    with _emulator.installed():
        a = training.ParamStore(small, torch.device("cpu"), skip=[k for k in small if "running_" in k or "num_batches" in k])
        a.buffers = {}
        oa = training.AdamW(a, lr=1e-3)
        sa = training.MultiStepLR(oa, [3], gamma=0.5)
        for i in range(2):
            fake_step(a, oa, sa, i)
        training.save_checkpoint(str(tmp_path / "ck" / "checkpoint-2steps.pkl"), a, oa, sa, steps=2, epochs=1)
        for i in range(2, 4):
            fake_step(a, oa, sa, i)
        assert oa.lr == 5e-4  # milestone 3 passed
        b = training.ParamStore(small, torch.device("cpu"), skip=[k for k in small if "running_" in k or "num_batches" in k])
        b.buffers = {}
        ob = training.AdamW(b, lr=123.0)
        sb = training.MultiStepLR(ob, [999])
        steps, epochs = training.load_checkpoint(str(tmp_path / "ck" / "checkpoint-2steps.pkl"), b, ob, sb)
        assert (steps, epochs) == (2, 1) and ob.steps == 2 and ob.lr == 1e-3 and sb.milestones == [3]
        for i in range(2, 4):
            fake_step(b, ob, sb, i)
    assert torch.equal(a.flat, b.flat) and torch.equal(oa.m, ob.m) and torch.equal(oa.v, ob.v) and ob.lr == 5e-4
    ck = torch.load(str(tmp_path / "ck" / "checkpoint-2steps.pkl"), weights_only=False)
    assert set(ck) == {"model", "optimizer", "scheduler", "steps", "epochs"}
    # the optimizer / scheduler entries are torch's own state_dicts: torch.optim.AdamW + MultiStepLR over parameters of
    # the same shapes load them (what the reference's trainer does on --resume, trainers/base.py:113-130) and a
    # checkpoint written by torch's objects loads here
    ps = [torch.nn.Parameter(v.detach().clone()) for v in a.state_dict().values()]
    topt = torch.optim.AdamW(ps, lr=1e-3, betas=oa.betas, eps=oa.eps, weight_decay=oa.wd)
    tsch = torch.optim.lr_scheduler.MultiStepLR(topt, milestones=[3], gamma=0.5)
    topt.load_state_dict(ck["optimizer"])
    tsch.load_state_dict(ck["scheduler"])
    k0 = next(iter(a.spans))
    off, n = a.spans[k0]
    assert topt.state[ps[0]]["exp_avg"].shape == ps[0].shape and float(topt.state[ps[0]]["step"]) == 2.0
    assert tsch.last_epoch == 2 and topt.param_groups[0]["lr"] == 1e-3
    torch.save({"model": ck["model"], "optimizer": topt.state_dict(), "scheduler": tsch.state_dict(), "steps": 2,
                "epochs": 1}, str(tmp_path / "ck" / "from_torch.pkl"))
    with _emulator.installed():
        c = training.ParamStore(small, torch.device("cpu"), skip=[k for k in small if "running_" in k or "num_batches" in k])
        c.buffers = {}
        oc = training.AdamW(c, lr=9.0)
        sc = training.MultiStepLR(oc, [7])
        assert training.load_checkpoint(str(tmp_path / "ck" / "from_torch.pkl"), c, oc, sc) == (2, 1)
        for i in range(2, 4):
            fake_step(c, oc, sc, i)
    assert torch.equal(a.flat, c.flat) and torch.equal(oa.m, oc.m) and oc.lr == 5e-4
    with pytest.raises(ValueError):
        torch.save({**ck, "optimizer": {"something": 1}}, str(tmp_path / "ck" / "bad.pkl"))
        training.load_checkpoint(str(tmp_path / "ck" / "bad.pkl"), c, oc)


def test_overwrite_gradient_path_equals_accumulation():
    """ParamStore.backward_into_flat (autograd.grad + srn_multi_copy) == zero_grad + loss.backward(); a parameter the
    loss does not touch ends up zero, not stale"""
    g = torch.Generator().manual_seed(1)
    sd = {f"p{i}": torch.randn(s, generator=g) for i, s in enumerate([(7, 5), (33,), (64, 9), (3,)])}
    with _emulator.installed():
        a, b = training.ParamStore(sd, torch.device("cpu")), training.ParamStore(sd, torch.device("cpu"))
        loss = lambda st: sum((st.params[k] ** 2).sum() * (i + 1) for i, k in enumerate(sd) if k != "p1")
        for p in a.params.values():
            p.grad.fill_(7.0)  # stale values in every gradient
        a.backward_into_flat(loss(a))
        b.zero_grad()
        loss(b).backward()
    assert torch.equal(a.flat_grad, b.flat_grad) and not a.params["p1"].grad.any()


def test_backward_after_a_second_forward_raises():
    """the estimator re-lays its conv weights into persistent buffers once per forward (Estimator._relay): a backward of
    an EARLIER forward would multiply its saved activations with the newer buffers -- autograd's version check catches
    it because _PackAll bumps the buffers' version"""
    w = sub(serenade_weights(), "cfm_decoder.estimator.")
    x1, mask, mu, spk, mask_l, t, z = _case()
    with _emulator.installed():
        est = training.Estimator(w, torch.device("cpu"))
        first, _ = training.cfm_loss(est, x1, mask, mu, spk, mask_l, draws={"t": t, "z": z})
        second, _ = training.cfm_loss(est, x1, mask, mu, spk, mask_l, draws={"t": t, "z": z})
        with pytest.raises(RuntimeError, match="modified by an inplace operation"):
            first.backward()
        est.zero_grad()
        second.backward()  # the latest forward is fine
    assert est.params["final_proj.weight"].grad.abs().max() > 0
