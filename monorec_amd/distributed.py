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


def init_from_env(backend=None, single_rank_group=False, pin_cpus=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun).  A one-process job creates no
    group unless `single_rank_group` asks for one (tests: a 1-rank "nccl" group runs the RCCL branch on a one-GPU box).
    `pin_cpus` (default: on, off with MR_PIN_CPUS=0): pin this rank to its share of the node's CPUs - BEFORE the device context and the process
    group exist, so that the helper threads HIP and RCCL spawn inherit the mask (pinning afterwards left them unpinned)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if (world == 1 and not single_rank_group) or dist.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"   # "nccl" is RCCL on ROCm
    if pin_cpus is None:
        pin_cpus = os.environ.get("MR_PIN_CPUS", "1") != "0"
    if world > 1 and pin_cpus:
        place_rank()            # this rank's share of the node's CPUs (NUMA node of its GPU), ahead of every thread the runtime will start
    kwargs = {}
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        kwargs["device_id"] = torch.device("cuda", local)
    dist.init_process_group(backend, rank=int(os.environ.get("RANK", "0")), world_size=world, **kwargs)


def _parse_cpulist(text):
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus += list(range(int(lo), int(hi or lo) + 1))
    return cpus


def rank_cpu_set(local_rank, local_world, allowed=None, gpu_numa=None):
    """CPUs one rank of a node should run on.  `gpu_numa[r]` = CPU list of the NUMA node the GPU of local rank r hangs off (None where
    unknown): the ranks whose GPUs share a node split that node's CPUs evenly, in rank order; without NUMA information every rank gets an
    even contiguous share of the allowed CPUs.  Either way a rank's busy-polling host thread, HIP's helper threads and its PNG-decode pool
    stay on one socket and off the other ranks' cores (8 ranks x (2 polling + ~5 decode) cores, nothing pinned).
    Pure function of its arguments."""
    allowed = sorted(allowed if allowed is not None else os.sched_getaffinity(0))
    aset = set(allowed)
    local_world = max(1, int(local_world))
    local_rank = min(max(0, int(local_rank)), local_world - 1)

    def split(cpus, k, n):
        # share k of n of EVERY run of consecutive CPU ids: Linux numbers the second hardware thread of core i as i + (cores of the box), so a
        # NUMA node's list is two runs (0-63,128-191) and a rank gets the same slice of both - whole cores, not another rank's sibling threads
        runs, out = [], []
        for c in cpus:
            if runs and c == runs[-1][-1] + 1:
                runs[-1].append(c)
            else:
                runs.append([c])
        for run in runs:
            per = len(run) // n
            if per == 0:
                continue
            out += run[k * per:(k + 1) * per] if k < n - 1 else run[k * per:]
        if out:
            return out
        per = max(1, len(cpus) // n)
        return cpus[k * per:(k + 1) * per] if k < n - 1 else cpus[k * per:]
    mine = gpu_numa[local_rank] if gpu_numa and local_rank < len(gpu_numa) else None
    if mine:
        node = [c for c in mine if c in aset]
        sharers = [r for r in range(min(local_world, len(gpu_numa))) if gpu_numa[r] == mine]
        share = split(node, sharers.index(local_rank), len(sharers)) if node else []
        if share:
            return share
    return split(allowed, local_rank, local_world) or allowed


_numa_cache = {}


def _gpu_numa_cpus(device_index):
    """CPU list of the NUMA node GPU `device_index` is attached to, or None (no sysfs entry, node -1, no torch device)."""
    if device_index in _numa_cache:
        return _numa_cache[device_index]
    cpus = None
    try:
        pr = torch.cuda.get_device_properties(device_index)
        addr = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        node = int(open(f"/sys/bus/pci/devices/{addr}/numa_node").read())
        if node >= 0:
            cpus = _parse_cpulist(open(f"/sys/devices/system/node/node{node}/cpulist").read())
    except Exception:
        cpus = None
    _numa_cache[device_index] = cpus
    return cpus


_placed = {}          # the first placement of this process: {"allowed": affinity before pinning, "info": summary}


def place_rank(local_rank=None, local_world=None, devices=None):
    """Pin this process - EVERY thread it has now (walks /proc/self/task) and, by inheritance, every thread started later - to its share of
    the node's CPUs (see `rank_cpu_set`) and size torch's intra-op pool to it.  Called by bench.py and `init_from_env` in multi-rank jobs,
    ahead of device / process-group initialisation; a one-rank job is left alone.  Idempotent: a second call returns the first call's summary
    instead of splitting the already narrowed mask again (place_rank() followed by init_from_env() ended up on 1 / n^2 of the CPUs).
    `devices[r]` = device index of local rank r (default r).  LOCAL_WORLD_SIZE missing: min(WORLD_SIZE, visible devices) when a device runtime is
    up, else WORLD_SIZE.  The NUMA split is used only when every local rank's GPU is visible to this process (per-rank HIP_VISIBLE_DEVICES
    isolation hides the others: then every rank takes an even share of the allowed CPUs, which never overlap).  Returns a summary dict."""
    if "info" in _placed:
        return dict(_placed["info"], repeated=True)
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if local_rank is None else int(local_rank)
    if local_world is None:
        if "LOCAL_WORLD_SIZE" in os.environ:
            local_world = int(os.environ["LOCAL_WORLD_SIZE"])
        else:
            local_world = int(os.environ.get("WORLD_SIZE", "1"))
            try:
                ndev = torch.cuda.device_count()
                if ndev > 0:
                    local_world = min(local_world, ndev) if ndev > 1 else local_world
            except Exception:
                pass
    local_world = int(local_world)
    info = {"local_rank": local_rank, "local_world_size": local_world, "pinned": False, "cpus": len(os.sched_getaffinity(0)), "numa": False}
    if local_world <= 1:
        return info
    try:
        allowed = sorted(os.sched_getaffinity(0))
        # one rank per GPU: the GPU of local rank r is device r, unless `devices` says otherwise (tests: every rank on device 0)
        try:
            ndev = torch.cuda.device_count()
        except Exception:
            ndev = 0
        want = [r if devices is None else devices[r] for r in range(local_world)]
        gpu_numa = [_gpu_numa_cpus(dv) for dv in want] if ndev > max(want) else [None] * local_world      # a rank that cannot see its peers' GPUs does not guess
        share = rank_cpu_set(local_rank, local_world, allowed=allowed, gpu_numa=gpu_numa)
        tids = [0]
        try:
            tids = [int(t) for t in os.listdir("/proc/self/task")] or [0]
        except OSError:
            pass
        for tid in tids:                                # sched_setaffinity(0, ..) alone pins the calling thread only
            try:
                os.sched_setaffinity(tid, share)
            except OSError:
                pass
        torch.set_num_threads(max(1, min(torch.get_num_threads(), len(share))))
        info.update({"pinned": True, "cpus": len(share), "cpu_range": [share[0], share[-1]], "numa": gpu_numa[local_rank] is not None, "threads_pinned": len(tids)})
        _placed["allowed"], _placed["info"] = allowed, dict(info)
    except Exception as e:          # affinity is a placement hint, never a reason to fail the job
        info["error"] = repr(e)
    return info


def host_thread_budget(cpus_for_rank):
    """Threads a rank may spend next to its enqueueing thread: (PNG-decode workers, host spin seconds).  The busy-polling host thread
    and HIP's helper threads take ~2 cores (bench line `host_cpu_ms_per_keyframe`); decode gets what is left, at most 8 (the
    reference's evaluation config runs 8 loader workers, configs/evaluate/eval_monorec.json:33); with fewer than 4 cores the host
    wait stops spinning almost at once and sleeps in hipEventSynchronize instead."""
    decode = max(1, min(8, int(cpus_for_rank) - 2))
    spin = 0.004 if cpus_for_rank >= 4 else 0.0002
    return decode, spin


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
