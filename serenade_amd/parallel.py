"""Multi-GPU decode: utterances shard embarrassingly across ranks (one process per GPU); the only exchange
step is the gather of the converted waveforms.  Mel frames are never exchanged.

The reference's analogue is file-level sharding with Kaldi `run.pl JOB=1:n` and no collective
(egs/gtsinger/ssc1/run.sh:141-162).  Here the collective is torch.distributed's gather over RCCL
(backend "nccl" on ROCm) — on xGMI every peer has its own link to the root, so a direct gather is used,
not a ring.  The same code runs over gloo on CPU tensors (tests)."""
import torch
import torch.distributed as dist


def rank_world():
    """(rank, world size) of this process: the initialised process group's, else torchrun's environment, else
    (0, 1)."""
    import os
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(n_items, rank, world_size):
    """Contiguous split of n_items over world_size ranks: ranks < n_items % world_size get one extra."""
    q, r = divmod(n_items, world_size)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def gather_waveforms(wave, n_samples=None, dst=0, group=None, uniform=False, always_collective=False):
    """Gather per-rank waveform batches on rank `dst`.

    wave: (B_local, N_local) float tensor (padded); n_samples: (B_local,) int64 valid lengths or None.
    Returns on dst: (list of (B_r, N_r) tensors, list of (B_r,) length tensors) in rank order; elsewhere None.
    Variable batch / length per rank is handled by first all-gathering the shapes, then padding to the max.
    uniform=True: the caller guarantees the same (B, N) and full lengths on every rank (fixed-shape serving, the
    benchmark): ONE collective, no shape exchange and no host synchronisation, so the host keeps running ahead of
    the GPU into the next batch."""
    single = not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1
    if single and not (always_collective and dist.is_available() and dist.is_initialized()):
        ns = n_samples if n_samples is not None else torch.full((wave.shape[0],), wave.shape[1], dtype=torch.int64)
        return [wave], [ns.cpu()]
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = wave.device
    if uniform:
        wave = wave.contiguous()
        bufs = [torch.empty_like(wave) for _ in range(world)] if rank == dst else None
        dist.gather(wave, bufs, dst=dst, group=group)
        if rank != dst:
            return None
        full = torch.full((wave.shape[0],), wave.shape[1], dtype=torch.int64)
        return bufs, [full.clone() for _ in range(world)]
    if n_samples is None:
        n_samples = torch.full((wave.shape[0],), wave.shape[1], dtype=torch.int64)
    shape = torch.tensor([wave.shape[0], wave.shape[1]], dtype=torch.int64, device=dev)
    shapes = [torch.zeros_like(shape) for _ in range(world)]
    dist.all_gather(shapes, shape, group=group)
    shapes = [s.cpu().tolist() for s in shapes]
    bmax = max(s[0] for s in shapes)
    nmax = max(s[1] for s in shapes)
    pad = torch.zeros(bmax, nmax, dtype=wave.dtype, device=dev)
    pad[: wave.shape[0], : wave.shape[1]] = wave
    lens = torch.zeros(bmax, dtype=torch.int64, device=dev)
    lens[: wave.shape[0]] = n_samples.to(dev)
    if rank == dst:
        bufs = [torch.empty_like(pad) for _ in range(world)]
        lbufs = [torch.empty_like(lens) for _ in range(world)]
    else:
        bufs = lbufs = None
    dist.gather(pad, bufs, dst=dst, group=group)
    dist.gather(lens, lbufs, dst=dst, group=group)
    if rank != dst:
        return None
    waves = [bufs[r][: shapes[r][0], : shapes[r][1]] for r in range(world)]
    nsamp = [lbufs[r][: shapes[r][0]].cpu() for r in range(world)]
    return waves, nsamp
