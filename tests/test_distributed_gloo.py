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


def test_single_process_fallback():
    assert mrd.world_info() == (0, 1)
    assert mrd.shard_batches(5) == [0, 1, 2, 3, 4]
    means, n = mrd.reduce_batch_metrics([[1.0, 2.0], [3.0, 4.0]])
    assert n == 2 and means == [2.0, 3.0]
