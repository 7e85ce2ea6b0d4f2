"""Evaluater loop mirror (evaluater/evaluater.py:38-50,78-118): host bookkeeping on CPU, the device loop on the GPU."""
import math

import numpy as np
import pytest
import torch

from monorec_amd import evaluate, synth
from oracle import monorec_oracle as orc


def _reference_loop(per_batch, sizes):
    """evaluater.py:72-118 written out independently of monorec_amd.evaluate (numpy, like the reference)."""
    n = len(per_batch[0])
    total_metrics, total_valid, running, num = np.zeros(n), np.zeros(n), np.zeros(n), 0
    for m, bs in zip(per_batch, sizes):
        acc = np.zeros(n)
        for i, v in enumerate(m):
            acc[i] += v
        if np.any(np.isnan(acc)):
            acc, valid = np.zeros(n), np.zeros(n)
        else:
            valid = np.ones(n)
        total_metrics += acc
        total_valid += valid
        if num == 0:
            running += acc
        else:
            running = running * (num / (num + bs)) + acc * (bs / (num + bs))
        num += bs
    return (total_metrics / total_valid).tolist(), running.tolist(), total_valid[0]


def test_evaluation_log_rules():
    per_batch = [[0.5, 2.0], [float("nan"), 1.0], [0.25, 4.0], [1.0, 1.0]]
    sizes = [2, 2, 1, 3]
    log = evaluate.evaluation_log(per_batch, sizes)
    m, r, v = _reference_loop(per_batch, sizes)
    assert log["metrics"] == m and log["metrics_correct"] == r and log["valid_batches"] == v == 3
    assert log["metrics"] == [(0.5 + 0.25 + 1.0) / 3, (2.0 + 4.0 + 1.0) / 3]      # the NaN batch is dropped entirely


@pytest.mark.gpu
def test_evaluater_matches_oracle_chain(hip_lib):
    from monorec_amd import MonoRecModel
    dev = "cuda:0"
    model = MonoRecModel(cv_depth_steps=8)
    sd = synth.seeded_state_dict(model.state_dict(), seed=0)
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    batches = []
    for i in range(5):
        data = synth.make_batch(2, 64, 96, 2, seed=40 + i)
        _, target = synth.make_depth_pair(2, 64, 96, seed=60 + i)
        if i == 2:
            target[1] = 0                      # one sample without ground truth -> NaN rmse -> whole batch invalid
        batches.append((data, target))
    log = evaluate.Evaluater(model, roi=None, max_distance=80).eval(batches)
    per_batch = []
    for data, target in batches:
        with torch.no_grad():
            res = model(synth.clone_batch(data, dev))["result"].cpu().clone()
        vals = orc.sparse_metrics(res, target, None, 80)
        per_batch.append([float(vals[k]) for k in evaluate._metrics.SPARSE_METRICS])
    m, r, v = _reference_loop(per_batch, [2] * 5)
    assert log["valid_batches"] == v == 4
    for got, want in zip(log["metrics"] + log["metrics_correct"], m + r):
        assert math.isclose(got, want, rel_tol=2e-5, abs_tol=1e-7), (got, want)
    # a metric subset keeps the order given, roi is honoured
    sub = evaluate.Evaluater(model, roi=[8, 56, 8, 88], max_distance=80,
                             metric_names=("a1_sparse_metric", "abs_rel_sparse_metric")).eval(batches[:2])
    want = []
    for data, target in batches[:2]:
        with torch.no_grad():
            res = model(synth.clone_batch(data, dev))["result"].cpu().clone()
        vals = orc.sparse_metrics(res, target, [8, 56, 8, 88], 80)
        want.append([float(vals["a1_sparse_metric"]), float(vals["abs_rel_sparse_metric"])])
    m2, _, _ = _reference_loop(want, [2, 2])
    assert all(math.isclose(a, b, rel_tol=2e-5) for a, b in zip(sub["metrics"], m2))


def _dist_eval_worker(rank, world, port, q):
    import os
    import torch.distributed as dist
    from monorec_amd import MonoRecModel, distributed as mrd
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    mrd.init_from_env("gloo")                          # both ranks share the box's one GPU; the collective runs over gloo
    log = _eval_log(distributed=True)
    q.put((rank, log))
    dist.barrier()
    dist.destroy_process_group()


def _eval_log(distributed):
    from monorec_amd import MonoRecModel
    dev = "cuda:0"
    model = MonoRecModel(cv_depth_steps=8)
    model.load_state_dict(synth.seeded_state_dict(model.state_dict(), seed=0))
    model = model.to(dev).eval()
    batches = []
    for i in range(5):
        data = synth.make_batch(2 if i != 3 else 1, 64, 96, 2, seed=40 + i)
        _, target = synth.make_depth_pair(2 if i != 3 else 1, 64, 96, seed=60 + i)
        if i == 2:
            target[1] = 0
        batches.append((data, target))
    return evaluate.Evaluater(model, roi=None, max_distance=80).eval(batches, distributed=distributed)


@pytest.mark.gpu
def test_two_rank_evaluation_equals_one_rank(hip_lib):
    """model -> fused metrics -> all-gather on 2 ranks (one device, gloo) == the single-process log to 1e-12."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    single = _eval_log(distributed=False)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dist_eval_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert single["valid_batches"] == 4
    for rank, log in res:
        assert log["valid_batches"] == single["valid_batches"]
        for key in ("metrics", "metrics_correct"):
            for a, b in zip(log[key], single[key]):
                assert abs(a - b) <= 1e-12 * max(1.0, abs(b)), (rank, key, a, b)


def _rccl_single_rank_worker(port, q):
    import os
    import torch.distributed as dist
    from monorec_amd import distributed as mrd
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    mrd.init_from_env("nccl", single_rank_group=True)          # "nccl" is RCCL on ROCm
    assert dist.get_backend() == "nccl" and mrd.group_active() and mrd.world_info() == (0, 1)
    # the collective itself, on device tensors through RCCL: records incl. a NaN batch come back unchanged and index-sorted
    per_batch = [[0.25, float("nan"), 3.0], [1.0, 2.0, 3.0], [0.5, 0.125, 8.0]]
    rows, sizes, idx = mrd.gather_batch_records(per_batch, [2, 1, 2], [4, 0, 2], 3)
    assert idx == [0, 2, 4] and sizes == [1, 2, 2]
    assert rows[0] == [1.0, 2.0, 3.0] and rows[1] == [0.5, 0.125, 8.0] and rows[2][0] == 0.25 and rows[2][1] != rows[2][1]
    g = mrd.gather_sums([1.5, -2.0])
    assert tuple(g.shape) == (1, 2) and g.tolist() == [[1.5, -2.0]]
    means, valid = mrd.reduce_batch_metrics(per_batch)
    assert valid == 2 and means == [0.75, 1.0625, 5.5]
    log = _eval_log(distributed=True)                          # model -> fused metrics -> RCCL all-gather -> bookkeeping
    q.put(log)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_one_rank_rccl_group_runs_the_collective_branch(hip_lib):
    """The "nccl" (= RCCL) branch of monorec_amd.distributed / Evaluater.eval(distributed=True) on the one GPU a test box has:
    a 1-rank process group on cuda:0.  The all-gathers run on device tensors through RCCL (base/base_trainer.py:26-29 is what the
    multi-GPU path replaces); the log must equal the plain single-process log."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    single = _eval_log(distributed=False)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_single_rank_worker, args=(port, q))
    p.start()
    log = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert log["valid_batches"] == single["valid_batches"] == 4
    for key in ("metrics", "metrics_correct"):
        for a, b in zip(log[key], single[key]):
            assert abs(a - b) <= 1e-12 * max(1.0, abs(b)), (key, a, b)


@pytest.mark.gpu
def test_bench_multi_rank_branch_with_two_ranks_on_one_device(hip_lib, tmp_path):
    """The driver launches `bench.py --gpus N` as `python -m torch.distributed.run --nproc-per-node N ...`; the builder's boxes have one GPU,
    so the N > 1 control flow (rank placement, per-rank keyframe streams, the all-gather of the per-rank summaries inside the timed region,
    barrier + MAX over ranks, rank 0 prints ONE JSON line) runs here with two ranks sharing cuda:0 over gloo (MR_BENCH_ONE_DEVICE=1): both
    ranks must be seen, their rates within 10 % of each other (they share one GPU evenly), `value` = their sum, the contract keys present."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MR_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "5",
           "--spinup-seconds", "1.0", "--height", "128", "--width", "256", "--depths", "16", "--no-cpu-baseline"]
    res = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]                     # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["steps"] == 40 and d["warmup"] == 5 and d["scaling"] == "weak"
    r0, r1 = d["per_rank_keyframes_per_s"]
    assert abs(r0 - r1) <= 0.10 * max(r0, r1), (r0, r1)
    assert d["value"] <= (r0 + r1) * 1.001 and d["value"] >= 0.85 * (r0 + r1), (d["value"], r0, r1)     # MAX-over-ranks clock: <= the sum of the rates
    assert d["host_placement"]["local_world_size"] == 2 and "roofline" in d and d["config"]["parallelism"] == "dp2 (independent keyframes per rank)"
