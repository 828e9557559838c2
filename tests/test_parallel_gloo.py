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


def _worker8(rank, world, port, q):
    """BASELINE configs[3]: 64 utterances sharded 8-way, waveforms gathered on rank 0.  Sample (u, t) of utterance u is
    u + t / 2^20 (exact in fp32), so reassembly in utterance order can be checked sample for sample."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total, n = 64, 2400  # 64 utterances x 10 frames of 240 samples (the headline's T = 1024 would be 245 760 samples each)
    a, b = shard_range(total, rank, world)
    ramp = torch.arange(4096, dtype=torch.float32) / 1048576.0
    wave = torch.stack([ramp[:n] + float(u) for u in range(a, b)])
    out = gather_waveforms(wave, dst=0, uniform=True)  # the benchmark's form: one collective, no shape exchange
    # ragged: 61 utterances (ranks 0-4 get 8, ranks 5-7 get 7), every utterance its own length, padded per rank
    a2, b2 = shard_range(61, rank, world)
    lens = torch.tensor([1200 + 37 * u for u in range(a2, b2)], dtype=torch.int64)
    nmax = int(lens.max())
    w2 = torch.zeros(b2 - a2, nmax)
    for i, u in enumerate(range(a2, b2)):
        w2[i, : lens[i]] = ramp[: lens[i]] + float(u)
    out2 = gather_waveforms(w2, lens, dst=0)
    if rank == 0:
        # numpy arrays travel by value (a tensor would travel as a shared-memory handle that dies with this process)
        q.put((torch.cat(out[0]).numpy(), [x.tolist() for x in out[1]], [w.numpy().copy() for w in out2[0]],
               [x.tolist() for x in out2[1]]))
    else:
        assert out is None and out2 is None
    dist.barrier()
    dist.destroy_process_group()


def test_c4_partitioning_gloo_world8():
    """world size 8 on the CPU: the contiguous 8-way split of 64 utterances and the gather's rank order / sample-exact
    reassembly (uniform and ragged), which world size 2 does not cover"""
    assert [shard_range(64, r, 8) for r in range(8)] == [(8 * r, 8 * r + 8) for r in range(8)]
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    import time
    t_end = time.time() + 240
    while q.empty() and time.time() < t_end and all(p.exitcode in (None, 0) for p in procs):
        time.sleep(0.2)
    if q.empty():  # a rank died or hung: do not leave its peers waiting in a collective
        for p in procs:
            p.kill()
        raise AssertionError(f"no result from rank 0 (exit codes {[p.exitcode for p in procs]})")
    full, ns, ragged, rns = q.get()
    full, ragged = torch.from_numpy(full), [torch.from_numpy(w) for w in ragged]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    ramp = torch.arange(4096, dtype=torch.float32) / 1048576.0
    assert tuple(full.shape) == (64, 2400) and ns == [[2400] * 8] * 8
    assert torch.equal(full, ramp[None, :2400] + torch.arange(64, dtype=torch.float32)[:, None])
    assert [len(x) for x in rns] == [8, 8, 8, 8, 8, 7, 7, 7]
    u = 0
    for w, lens in zip(ragged, rns):
        for i, n in enumerate(lens):
            assert n == 1200 + 37 * u
            assert torch.equal(w[i, :n], ramp[:n] + float(u)) and (w[i, n:] == 0).all()
            u += 1
    assert u == 61
