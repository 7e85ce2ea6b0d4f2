"""Multi-GPU evaluation: one process per GPU, keyframe batches sharded across ranks, ONE tiny
all-gather of per-rank metric sums at the end (RCCL over xGMI on MI355X; gloo on CPU for tests).

This replaces the reference's single-process `torch.nn.DataParallel` wrapping
(base/base_trainer.py:26-29, evaluater/evaluater.py:27-30), which scatters every batch across GPUs,
re-broadcasts the 70 MB of weights on every forward and gathers outputs to GPU 0.  Keyframes are
independent in eval mode, so there is no data-path collective at all: every rank owns a
replica of the weights and a contiguous/round-robin shard of the *batch list* (batch granularity keeps
the reference's per-batch metric semantics, evaluater.py:94-103,116), and the only exchange is
`world x (num_metrics + 1)` float64 values.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, single_rank_group=False):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun).  A one-process job creates no
    group unless `single_rank_group` asks for one (tests: a 1-rank "nccl" group runs the RCCL branch on a one-GPU box)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if (world == 1 and not single_rank_group) or dist.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"   # "nccl" is RCCL on ROCm
    kwargs = {}
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        kwargs["device_id"] = torch.device("cuda", local)
    dist.init_process_group(backend, rank=int(os.environ.get("RANK", "0")), world_size=world, **kwargs)


def world_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def group_active():
    """True when a process group exists - also a one-rank group: the collectives below then really run (RCCL with the "nccl"
    backend), so that the branch a multi-GPU job takes is the branch a one-GPU box can test."""
    return dist.is_available() and dist.is_initialized()


def shard_batches(num_batches, rank=None, world=None, contiguous=False):
    """Indices of the batches this rank evaluates. Round-robin by default (balances a sequence whose
    cost drifts); contiguous keeps neighbouring keyframes (source-frame reuse) on one rank."""
    if rank is None or world is None:
        rank, world = world_info()
    if contiguous:
        per = (num_batches + world - 1) // world
        return list(range(rank * per, min(num_batches, (rank + 1) * per)))
    return list(range(rank, num_batches, world))


def gather_sums(local_sums, device=None):
    """all_gather of a small float64 vector; returns a (world, n) tensor on every rank."""
    rank, world = world_info()
    t = torch.as_tensor(local_sums, dtype=torch.float64)
    if not group_active():
        return t.reshape(1, -1)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = t.to(device)
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return torch.stack(out).cpu()


def gather_batch_records(per_batch_metrics, batch_sizes, batch_indices, n_metrics):
    """The one collective of a multi-rank evaluation: every rank contributes (global batch index, batch size, metric vector)
    for the batches it evaluated and receives the index-sorted union.  Two all-gathers of float64 - the record counts, then the
    records padded to the largest count - a few hundred bytes per rank; NaN metric values travel unchanged (the validity rule
    is applied afterwards by `evaluate.evaluation_log`, exactly as in one process)."""
    rank, world = world_info()
    local = torch.full((len(per_batch_metrics), n_metrics + 2), float("nan"), dtype=torch.float64)
    for r, (m, bs, gi) in enumerate(zip(per_batch_metrics, batch_sizes, batch_indices)):
        local[r, 0], local[r, 1] = float(gi), float(bs)
        local[r, 2:] = torch.as_tensor([float(v) for v in m], dtype=torch.float64)
    if not group_active():
        rows = local
    else:
        counts = gather_sums([float(local.shape[0])]).reshape(-1).to(torch.int64)
        width = int(counts.max().item())
        padded = torch.full((width, n_metrics + 2), float("nan"), dtype=torch.float64)
        padded[: local.shape[0]] = local
        allr = gather_sums(padded.reshape(-1)).reshape(world, width, n_metrics + 2)
        rows = torch.cat([allr[r, : int(counts[r])] for r in range(world)]) if width else local
    order = torch.argsort(rows[:, 0]) if rows.shape[0] else torch.zeros(0, dtype=torch.int64)
    rows = rows[order]
    idx = [int(v) for v in rows[:, 0].tolist()]
    if len(set(idx)) != len(idx):
        raise RuntimeError(f"gather_batch_records: batch indices evaluated twice across ranks: {idx}")
    return [r[2:].tolist() for r in rows], [int(v) for v in rows[:, 1].tolist()], idx


def reduce_batch_metrics(per_batch_metrics):
    """Global mean over VALID batches of per-batch metric vectors evaluated on this rank's shard: what `Evaluater.eval`
    reports as 'metrics' (evaluater.py:45-49,116) - a batch with any NaN metric contributes neither values nor a count.
    Returns (means, number of valid batches).  (`Evaluater.eval(distributed=True)` uses `gather_batch_records` instead,
    which also reproduces 'metrics_correct'.)"""
    n_metrics = len(per_batch_metrics[0]) if per_batch_metrics else 0
    _, world = world_info()
    if group_active():   # ranks with an empty shard still need the vector length
        lens = gather_sums([float(n_metrics)])
        n_metrics = int(lens.max().item())
    local = torch.zeros(n_metrics + 1, dtype=torch.float64)
    for m in per_batch_metrics:
        v = torch.as_tensor(m, dtype=torch.float64)
        if bool(torch.isnan(v).any()):
            continue
        local[:n_metrics] += v
        local[n_metrics] += 1
    total = gather_sums(local).sum(0)
    valid = int(total[n_metrics].item())
    if valid == 0:
        return [float("nan")] * n_metrics, 0
    return (total[:n_metrics] / total[n_metrics]).tolist(), valid
