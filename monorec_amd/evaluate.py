"""The evaluation loop of the reference (`Evaluater.eval`, evaluater/evaluater.py:38-50,78-118) around the MI355X path
(second half of SURVEY section 8 row f-1).

The reference evaluates seven metric functions per batch, each re-deriving its masks from the full tensors and each
ending in a host synchronisation (`acc_metrics[i] += metric(...)`, :43).  Here a batch costs one forward (kept in
flight through `MonoRecModel.submit`) and one fused reduction launch; the per-sample sums stay on the device and are
copied once, after the last batch.  The bookkeeping - a batch with a NaN metric counts as invalid and contributes
zeros (:45-49), `metrics` = sum over valid batches / number of valid batches (:116), `metrics_correct` = running
average weighted by batch size (:100-104) - is reproduced on the host in float64 like the numpy code it mirrors.
"""
import collections

import numpy as np
import torch

from . import metrics as _metrics


def evaluation_log(per_batch_metrics, batch_sizes):
    """per_batch_metrics: list of equal-length sequences (float, NaN allowed), batch_sizes: list of ints ->
    {'metrics', 'metrics_correct', 'valid_batches'} with the reference's rules (evaluater.py:45-49,94-118)."""
    n = len(per_batch_metrics[0]) if per_batch_metrics else 0
    total, valid_total, running = np.zeros(n), np.zeros(n), np.zeros(n)
    seen = 0
    for vals, bsz in zip(per_batch_metrics, batch_sizes):
        acc = np.zeros(n)
        for i, v in enumerate(vals):
            acc[i] += float(v)
        if np.any(np.isnan(acc)):
            acc, valid = np.zeros(n), np.zeros(n)
        else:
            valid = np.ones(n)
        total += acc
        valid_total += valid
        running = acc.copy() if seen == 0 else running * (seen / (seen + bsz)) + acc * (bsz / (seen + bsz))
        seen += bsz
    with np.errstate(invalid="ignore", divide="ignore"):
        mean = total / valid_total
    return {"metrics": mean.tolist(), "metrics_correct": running.tolist(),
            "valid_batches": float(valid_total[0]) if n else 0.0}


class Evaluater:
    """`Evaluater(model, roi=, max_distance=).eval(data_loader)` -> the reference's log dict (without the loss entries,
    which are constant zero there, evaluater.py:85-86).  `data_loader` yields `(data_dict, target)` like the reference's
    loaders; tensors may live on the host (they are moved) or already on the model's device.

    `eval(data_loader, distributed=True)` is the multi-GPU form (one process per GPU, `monorec_amd.distributed`): whole
    batches are sharded round-robin over the ranks - by the loader itself when it was built with this rank / world size
    (`kitti.DeviceLoader(rank=, world_size=)`), otherwise by skipping the batches of the other ranks - every rank evaluates
    its batches exactly as the single process would, and ONE all-gather (RCCL over xGMI; 10 float64 per batch) hands every
    rank the per-batch metric vectors, batch sizes and batch indices of all ranks.  The reference's bookkeeping
    (`evaluation_log`: NaN batch => invalid, mean over valid batches, batch-size-weighted running mean, evaluater.py:45-49,
    94-118) then runs on the index-sorted union, so the log equals the single-process log bit for bit on every rank."""

    def __init__(self, model, roi=None, max_distance=None, metric_names=_metrics.SPARSE_METRICS, in_flight=None, sums_fn=None):
        unknown = [m for m in metric_names if m not in _metrics.SPARSE_METRICS]
        if unknown:
            raise NotImplementedError(f"metrics outside the fused sparse set: {unknown}")
        self.model, self.roi, self.max_distance = model, roi, max_distance
        self.metric_names = tuple(metric_names)
        self._cols = [_metrics.SPARSE_METRICS.index(m) for m in self.metric_names]
        # forwards kept in flight: never more than the model has slots - a deeper queue would let submit() reuse a slot whose
        # resident `result` has not been reduced yet (the metric launch would then read the wrong keyframe's prediction)
        slots = int(getattr(model, "hip_in_flight", in_flight or 2))
        self.in_flight = max(1, min(int(in_flight) if in_flight is not None else slots, slots))      # default: every slot of the model
        self._sums_fn = sums_fn or _metrics.sparse_metric_sums_device   # (B, 8) per-sample sums; injectable for host-logic tests

    # the 4x4 matrices feed the model's HOST-side pose algebra (model.host_geometry): moved to the device they would have to come back,
    # and submit() would wait for that copy - a loader's CPU matrices are left where they are (the reference moves everything,
    # evaluater.py:82; the values are the same either way)
    _HOST_KEYS = ("keyframe_pose", "keyframe_intrinsics", "poses", "intrinsics", "stereoframe_pose", "stereoframe_intrinsics")

    @staticmethod
    def _to(obj, device):
        if torch.is_tensor(obj):
            return obj.to(device, non_blocking=True)
        if isinstance(obj, (list, tuple)):
            return [Evaluater._to(o, device) for o in obj]
        if isinstance(obj, dict):
            return {k: (v if k in Evaluater._HOST_KEYS else Evaluater._to(v, device)) for k, v in obj.items()}
        return obj

    def eval(self, data_loader, distributed=False):
        from . import distributed as _dist
        rank, world = _dist.world_info() if distributed else (0, 1)
        # a loader that already yields this rank's shard (DeviceLoader(rank=, world_size=)) is taken as is
        presharded = world > 1 and getattr(data_loader, "world_size", 1) == world and getattr(data_loader, "rank", None) == rank
        device = next(self.model.parameters()).device
        sums, sizes, indices = [], [], []
        pending = collections.deque()

        def collect():
            data, handle, gidx = pending.popleft()
            out = handle.synchronize()                              # the host waits: no blocked wait packet on the caller's stream
            sums.append(self._sums_fn({"result": out["result"], "target": data["target"]}, self.roi, self.max_distance))
            sizes.append(int(data["target"].shape[0]))
            indices.append(gidx)

        self.model.eval()
        with torch.no_grad():
            for i, (data, target) in enumerate(data_loader):
                if world > 1 and not presharded and i % world != rank:
                    continue
                data = self._to(data, device)
                data["target"] = self._to(target, device)
                # pose algebra while the device is busy (not with dynamic batching: a coalesced launch forms the matrices of its group)
                batching = getattr(self.model, "_batch_keyframes", 1) > 1
                token = self.model.prepare(data) if (hasattr(self.model, "prepare") and not batching) else None
                if len(pending) >= self.in_flight:                  # the slot the next submit reuses: reduce its result first
                    collect()
                handle = self.model.submit(data, token) if token is not None else self.model.submit(data)
                pending.append((data, handle, rank + i * world if presharded else i))
            while pending:
                collect()
        per_batch = []
        if sums:
            host = [s.cpu() for s in sums] if len({tuple(s.shape) for s in sums}) > 1 else list(torch.stack(sums).cpu())
            for s in host:
                vals = _metrics.metrics_from_sums(s)
                per_batch.append([float(vals[c]) for c in self._cols])
        if distributed and _dist.group_active():      # also a one-rank group: the collective is the code path under test there
            per_batch, sizes, indices = _dist.gather_batch_records(per_batch, sizes, indices, len(self._cols))
        if not per_batch:
            return evaluation_log([], [])
        return evaluation_log(per_batch, sizes)
