"""CPU, world_size 2 over gloo: batch sharding and the metric all-gather of monorec_amd.distributed
(the N>1 path of bench.py / evaluation; RCCL on the GPU box uses the same code with backend nccl)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from monorec_amd import distributed as mrd


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_metrics(batch_idx):
    g = torch.Generator().manual_seed(100 + batch_idx)
    return torch.rand(7, generator=g, dtype=torch.float64).tolist()


def _worker(rank, world, port, num_batches, contiguous, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    mrd.init_from_env("gloo")
    idx = mrd.shard_batches(num_batches, contiguous=contiguous)
    means, n = mrd.reduce_batch_metrics([_fake_metrics(i) for i in idx])
    gathered = mrd.gather_sums([float(rank), float(len(idx))])
    q.put((rank, idx, means, n, gathered.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def _run(num_batches, contiguous):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, num_batches, contiguous, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_round_robin_shards_and_metric_allgather_match_single_process():
    n = 7
    res = _run(n, contiguous=False)
    all_idx = sorted(i for _, idx, _, _, _ in res for i in idx)
    assert all_idx == list(range(n))                                  # every batch exactly once
    want = (torch.tensor([_fake_metrics(i) for i in range(n)], dtype=torch.float64).sum(0) / n).tolist()
    for rank, idx, means, count, gathered in res:
        assert count == n
        assert max(abs(a - b) for a, b in zip(means, want)) < 1e-12   # equal to 1e-12
        assert gathered == [[0.0, float(len(res[0][1]))], [1.0, float(len(res[1][1]))]]


def test_contiguous_shards_with_an_empty_rank():
    res = _run(1, contiguous=True)            # rank 1 gets nothing and must still take part in the collective
    assert res[0][1] == [0] and res[1][1] == []
    want = _fake_metrics(0)
    for _, _, means, count, _ in res:
        assert count == 1 and max(abs(a - b) for a, b in zip(means, want)) < 1e-12


def test_rank_cpu_shares_follow_the_numa_node_of_the_gpu():
    """distributed.rank_cpu_set: the ranks whose GPUs hang off one NUMA node split that node's CPUs evenly - whole cores (the same slice of
    both hardware-thread runs) -, the shares of a node's ranks are disjoint and cover it; without NUMA information: contiguous even shares."""
    n0, n1 = list(range(0, 64)) + list(range(128, 192)), list(range(64, 128)) + list(range(192, 256))
    numa = [n0] * 4 + [n1] * 4
    shares = [mrd.rank_cpu_set(r, 8, allowed=range(256), gpu_numa=numa) for r in range(8)]
    assert all(len(s) == 32 for s in shares) and sorted(c for s in shares for c in s) == list(range(256))
    assert shares[0] == list(range(0, 16)) + list(range(128, 144)) and set(shares[5]) <= set(n1)
    assert mrd.rank_cpu_set(1, 2, allowed=range(8)) == [4, 5, 6, 7] and mrd.rank_cpu_set(0, 1, allowed=range(8)) == list(range(8))
    assert mrd.rank_cpu_set(3, 4, allowed=[0, 1]) == [0, 1]                      # fewer CPUs than ranks: everything, never nothing
    assert mrd.rank_cpu_set(1, 2, allowed=range(4), gpu_numa=[[0, 1, 2, 3], None]) == [2, 3]   # unknown node for this rank: the plain split
    assert mrd.host_thread_budget(32) == (8, 0.004) and mrd.host_thread_budget(3) == (1, 0.0002)
    info = mrd.place_rank(0, 1)
    assert info["pinned"] is False                                              # a one-rank job is left alone


def _placement_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    before = sorted(os.sched_getaffinity(0))
    first = mrd.place_rank()                              # a caller that places itself first ...
    mrd.init_from_env("gloo")                             # ... must not be split a second time (1 / n^2 of the CPUs)
    again = mrd.place_rank()
    assert again.get("repeated") and again["cpus"] == first["cpus"]
    import threading
    seen = []
    t = threading.Thread(target=lambda: seen.append(sorted(os.sched_getaffinity(0))))       # threads started later inherit the mask
    t.start(); t.join()
    assert seen[0] == sorted(os.sched_getaffinity(0))
    q.put((rank, before, sorted(os.sched_getaffinity(0)), torch.get_num_threads()))
    dist.barrier()
    dist.destroy_process_group()


def test_init_from_env_pins_every_rank_of_a_multi_rank_job_to_its_share():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_placement_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, before, a0, t0), (_, _, a1, t1) = res
    if len(before) >= 2:
        assert a0 and a1 and not (set(a0) & set(a1)) and set(a0) | set(a1) <= set(before)
        assert t0 <= len(a0) and t1 <= len(a1)


def test_single_process_fallback():
    assert mrd.world_info() == (0, 1)
    assert mrd.shard_batches(5) == [0, 1, 2, 3, 4]
    means, n = mrd.reduce_batch_metrics([[1.0, 2.0], [3.0, 4.0]])
    assert n == 2 and means == [2.0, 3.0]


# ---- Evaluater.eval(distributed=True): the multi-rank evaluation reproduces the single-process log exactly ----------------------
class _StubHandle:
    def __init__(self, out):
        self._out = out

    def result(self):
        return self._out

    synchronize = result


class _StubModel:
    """Stands in for MonoRecModel on the CPU: `submit()` derives a deterministic 'prediction' from the keyframe."""
    hip_in_flight = 2

    def eval(self):
        return self

    def parameters(self):
        yield torch.zeros(1)

    def submit(self, data):
        return _StubHandle({"result": data["keyframe"].abs().mean(dim=1, keepdim=True) * 0.3 + 0.01})


def _cpu_sums(data, roi, max_distance):
    """(B, 8) per-sample sums in the column order of mr_sparse_metric_sums_f32 (count, abs_rel, sq_rel, se, sle, a1, a2, a3);
    a plain torch restatement for the host-logic test (the product reduces on the device)."""
    pred, gt = data["result"].double(), data["target"].double()
    out = torch.zeros(pred.shape[0], 8, dtype=torch.float64)
    for b in range(pred.shape[0]):
        m = gt[b] > 1 / 80
        p, g = 1 / pred[b][m].clamp_min(1 / 80), 1 / gt[b][m]
        r = torch.maximum(p / g, g / p)
        out[b] = torch.tensor([m.sum(), ((p - g).abs() / g).sum(), ((p - g) ** 2 / g).sum(), ((p - g) ** 2).sum(),
                               ((p.log() - g.log()) ** 2).sum(), (r < 1.25).sum(), (r < 1.25 ** 2).sum(), (r < 1.25 ** 3).sum()])
    return out


def _eval_batches():
    g = torch.Generator().manual_seed(7)
    batches = []
    for i, bs in enumerate([2, 2, 1, 2, 2, 2, 1]):
        kf = torch.rand(bs, 3, 8, 12, generator=g) - 0.5
        target = torch.rand(bs, 1, 8, 12, generator=g) * 0.3 + 0.02
        target[torch.rand(bs, 1, 8, 12, generator=g) < 0.5] = 0
        if i == 3:
            target[:] = 0                      # a batch without ground truth: NaN metrics -> invalid batch (evaluater.py:45-49)
        batches.append(({"keyframe": kf}, target))
    return batches


class _ShardedLoader:
    """Yields only this rank's batches, like kitti.DeviceLoader(rank=, world_size=)."""

    def __init__(self, batches, rank, world_size):
        self.batches, self.rank, self.world_size = batches, rank, world_size

    def __iter__(self):
        return iter(self.batches[self.rank::self.world_size])


def _eval_worker(rank, world, port, presharded, q):
    from monorec_amd import evaluate
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    mrd.init_from_env("gloo")
    batches = _eval_batches()
    loader = _ShardedLoader(batches, rank, world) if presharded else batches
    log = evaluate.Evaluater(_StubModel(), max_distance=80, sums_fn=_cpu_sums).eval(loader, distributed=True)
    q.put((rank, log))
    dist.barrier()
    dist.destroy_process_group()


def _run_eval(presharded):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_eval_worker, args=(r, world, port, presharded, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def _same(a, b):
    return a == b or (a != a and b != b)


def test_distributed_evaluater_equals_single_process_log():
    from monorec_amd import evaluate
    single = evaluate.Evaluater(_StubModel(), max_distance=80, sums_fn=_cpu_sums).eval(_eval_batches())
    assert single["valid_batches"] == 6                                # the batch without ground truth is invalid
    for presharded in (False, True):
        for rank, log in _run_eval(presharded):
            assert log["valid_batches"] == single["valid_batches"]
            for key in ("metrics", "metrics_correct"):                 # bit for bit, on every rank
                assert all(_same(a, b) for a, b in zip(log[key], single[key])), (presharded, rank, key, log[key], single[key])


def test_reduce_batch_metrics_skips_nan_batches():
    means, n = mrd.reduce_batch_metrics([[1.0, 2.0], [float("nan"), 5.0], [3.0, 4.0]])
    assert n == 2 and means == [2.0, 3.0]
    means, n = mrd.reduce_batch_metrics([[float("nan")]])
    assert n == 0 and means[0] != means[0]


def test_evaluater_in_flight_is_clamped_to_the_models_slots():
    from monorec_amd import evaluate
    assert evaluate.Evaluater(_StubModel(), in_flight=8, sums_fn=_cpu_sums).in_flight == 2


def test_decode_stress_reports_every_rank_of_a_shared_host():
    """tools/decode_stress.py: N processes placed like the ranks of one node sweep the device loader's frame cache at the same time - one
    decoded image per keyframe each (the cache), a decode time and a host-side keyframes/s per rank: the ceiling of `with_data_loading` at N ranks."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "tools", "decode_stress.py"), "--ranks", "2", "--keyframes", "12"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-400:]
    out = json.loads(res.stdout.strip().splitlines()[-1])
    assert out["ranks"] == 2 and [r["rank"] for r in out["per_rank"]] == [0, 1]
    for r in out["per_rank"]:
        assert r["decoded_images_per_keyframe"] == 1.0 and r["decode_ms_per_keyframe"] > 0 and r["host_keyframes_per_s"] > 0
        assert r["pinned"] == (len(os.sched_getaffinity(0)) >= 2) or not r["pinned"]
    assert abs(out["aggregate_host_keyframes_per_s"] - sum(r["host_keyframes_per_s"] for r in out["per_rank"])) < 1e-6
