"""world_size-2 gloo test (CPU) of the only multi-GPU step of the path: utterance sharding + waveform gather."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from serenade_amd.parallel import gather_waveforms, shard_range


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 64, 65):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total = 5  # 5 utterances over 2 ranks -> 3 + 2 (ragged), different lengths per rank
    a, b = shard_range(total, rank, world)
    n = 480 + 240 * rank
    wave = torch.stack([torch.full((n,), float(i)) + torch.arange(n) * 1e-3 for i in range(a, b)])
    lens = torch.tensor([n - 10 * i for i in range(a, b)], dtype=torch.int64)
    out = gather_waveforms(wave, lens, dst=0)
    if rank == 0:
        waves, ns = out
        q.put(([w.clone() for w in waves], [x.clone() for x in ns]))
    else:
        assert out is None
    # fixed-shape fast path (the benchmark / serving loop): one collective, no shape exchange
    same = torch.full((2, 96), float(rank)) + torch.arange(96) * 1e-2
    out2 = gather_waveforms(same, dst=0, uniform=True)
    if rank == 0:
        w2, n2 = out2
        assert [tuple(w.shape) for w in w2] == [(2, 96)] * world and all(n.tolist() == [96, 96] for n in n2)
        assert all(abs(w2[r][1, 95].item() - (r + 0.95)) < 1e-6 for r in range(world))
    else:
        assert out2 is None
    dist.barrier()
    dist.destroy_process_group()


def test_gather_waveforms_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    waves, ns = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [tuple(w.shape) for w in waves] == [(3, 480), (2, 720)]
    assert ns[0].tolist() == [480, 470, 460] and ns[1].tolist() == [690, 680]
    assert waves[1][0, 0].item() == 3.0 and abs(waves[1][1, 719].item() - (4.0 + 0.719)) < 1e-5


def test_single_process_is_identity():
    w = torch.randn(2, 10)
    waves, ns = gather_waveforms(w)
    assert waves[0] is w and ns[0].tolist() == [10, 10]


# ---- f4: the DDP replacement of the training step (serenade_amd/training.py GradSync) ------------------------------
def _grad_worker(rank, world, port, q):
    from serenade_amd import training
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(3)
    sd = {f"p{i}": torch.randn(s, generator=g) for i, s in enumerate([(7, 5), (33,), (64, 9), (3,), (128, 4), (10,)])}
    est = training.Estimator(sd, torch.device("cpu"))
    sync = training.GradSync(est, bucket_bytes=2048)  # several buckets, the last ones finish first in backward
    assert len(sync.buckets) >= 3 and sync.buckets[0][0] == 0 and sync.buckets[-1][1] == est.flat.numel()
    for step in range(2):  # twice: the bucket counters re-arm
        est.zero_grad()
        coef = {k: torch.randn(v.shape, generator=torch.Generator().manual_seed(100 * rank + i + 7 * step))
                for i, (k, v) in enumerate(sd.items())}
        used = [k for k in sd if not (step == 1 and k == "p3")]  # step 1: one parameter gets no gradient at all
        loss = sum((est.params[k] * coef[k]).sum() for k in used)
        loss.backward()
        sync.finish()
        expect = {}
        for i, k in enumerate(sd):
            both = [torch.randn(sd[k].shape, generator=torch.Generator().manual_seed(100 * r + i + 7 * step))
                    for r in range(world)]
            expect[k] = sum(both) / world if k in used else torch.zeros_like(sd[k])
        for k in sd:
            assert torch.allclose(est.params[k].grad, expect[k], atol=1e-6), (step, k)
    if rank == 0:
        q.put(len(sync.buckets))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_buckets_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    n_buckets = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert n_buckets >= 3


def _est_worker(rank, world, port, q):
    """the real backward graph: every rank differentiates its own batch through the (emulated) estimator with the
    bucketed overlap hooks on; the result must be the mean of the ranks' stand-alone gradients"""
    from serenade_amd import training
    from tests import _emulator
    from tests._weights import serenade_weights, sub
    from tests.test_training_emulated import _case
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = sub(serenade_weights(), "cfm_decoder.estimator.")
    cases = [_case(B=1, L=16, lens=(16,), seed=40 + r) for r in range(world)]
    with _emulator.installed():
        def grads_of(case, sync_factory):
            est = training.Estimator(w, torch.device("cpu"))
            sync = sync_factory(est)
            x1, mask, mu, spk, mask_l, t, z = case
            mask_l = mask.clone()
            mask_l[:, :, :3] = 0
            mask_l[:, :, 11:] = 0
            loss, _ = training.cfm_loss(est, x1, mask, mu, spk, mask_l, draws={"t": t, "z": z})
            est.zero_grad()
            loss.backward()
            if sync is not None:
                sync.finish()
            return est, sync
        est, sync = grads_of(cases[rank], lambda e: training.GradSync(e, bucket_bytes=32 << 20))
        assert len(sync.buckets) >= 3 and sync.launched == len(sync.buckets)
        if rank == 0:
            alone = [grads_of(c, lambda e: None)[0].flat_grad for c in cases]
            want = sum(alone) / world
            err = ((est.flat_grad - want).abs().max() / want.abs().max()).item()
            q.put(err)
    dist.barrier()
    dist.destroy_process_group()


def test_estimator_gradient_allreduce_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_est_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    err = q.get()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert err < 1e-6
