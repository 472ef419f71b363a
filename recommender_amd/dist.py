"""User-sharded scoring across the GPUs of one node.

One process per GPU (torchrun / torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in the CPU tests).
Tables and weights are replicated; every rank scores a contiguous user range against the whole catalog;
the only data-path collective is ONE all-gather of the per-user top-k lists (k x 8 bytes per user: ids and scores travel
in one packed int32 buffer).  With user sharding the lists are disjoint, so the "merge" is a concatenation in rank order.

`TopkGather` keeps that collective off the critical path: the gather of step i is enqueued on RCCL's stream behind step i's
top-k kernel, and the scoring kernels of step i + 1 are enqueued on the compute stream without waiting for it (two packed
buffers in flight).  Where the gather kernel actually RUNS depends on the CUs: the persistent tile kernels hold 152 of 160 KB
of LDS on every CU, so without a reservation RCCL's kernel starts in the gap between two tile launches (beside the small
projection / select kernels) and a tile launch that finds a CU still busy simply starts that workgroup later -- tiles come from
a work queue.  `reserve_cus` (or RECO_RESERVE_CUS) keeps that many CUs free of tile workgroups so the gather can run under a
tile launch; it costs reserve/256 of the scoring throughput and is off by default.
With ONE rank the lists are already complete: no collective is issued at all -- unless `force_collective` (or RECO_FORCE_GATHER=1)
asks for it: a one-rank RCCL group then runs the very same enqueue / wait / unpack code as N ranks, which is how the RCCL leg is
exercised on a one-GPU box (tests/test_gpu_dist.py; RCCL refuses two ranks on one device).

Ordering of the collective behind the kernels that produce its input ("the round-1 hang"):
RCCL ("nccl") launches the collective as a kernel on its own stream behind an event recorded on the caller's stream, so
it is stream-ordered by construction.  Any other backend (gloo: the CPU tests, and rehearsals where several ranks share
one GPU) only ever sees HOST tensors here: the packed lists are copied to the host behind the kernels that produce them (a
synchronous copy on the compute stream), gathered there, and the result is copied back.  gloo's own handling of device tensors
(staging through pinned memory from a helper thread on private streams) hung on a shared GPU -- in round 1 whenever the
producing kernels were still enqueued, and in round 3 after some tens of back-to-back steps even with the producer finished
first (8 warm-up steps passed, 48 did not).  work.wait() carries a timeout, so a stuck collective raises GatherTimeout
instead of hanging the job.
"""
import datetime
import os

import torch
import torch.distributed as dist


class GatherTimeout(RuntimeError):
    """The top-k all-gather did not complete in time.  The process group is not usable afterwards: callers restart the
    job (bench.py relaunches its ranks with the blocking gather), they do not retry in-process."""


def gather_timeout_s():
    return float(os.environ.get("RECO_GATHER_TIMEOUT_S", "60"))


def reserved_cus_setting(value=None):
    """CUs the tile kernels leave to the all-gather kernel: the argument, else RECO_RESERVE_CUS, else 0; within [0, 128]."""
    if value is None:
        value = os.environ.get("RECO_RESERVE_CUS", "0")
    r = int(value)
    if not 0 <= r <= 128:
        raise ValueError("reserve_cus must be within [0, 128], got %d" % r)
    return r


def force_collective_setting(value=None):
    """Issue the all-gather even in a one-rank group: the argument, else RECO_FORCE_GATHER (rehearsal / test switch)."""
    if value is None:
        value = os.environ.get("RECO_FORCE_GATHER", "0") not in ("", "0")
    return bool(value)


def _stream_ordered(group=None):
    """True when the backend enqueues collectives on the device behind the caller's stream (RCCL)."""
    return dist.is_initialized() and str(dist.get_backend(group)).lower() == "nccl"


def _finish_producer(t):
    """Host-waits for everything enqueued so far on the current stream of t's device (no-op for host tensors)."""
    if t.is_cuda:
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(t.device))
        ev.synchronize()


def _wait(work, stream_ordered):
    """Makes the collective's result visible to the caller's stream / thread; raises GatherTimeout instead of hanging."""
    if stream_ordered:
        work.wait()          # the compute stream waits for RCCL's stream, the host does not; a stuck collective is ended
        return               # by the process group's own watchdog (timeout set at init_process_group)
    try:
        done = work.wait(datetime.timedelta(seconds=gather_timeout_s()))
    except RuntimeError as exc:
        raise GatherTimeout("top-k all-gather did not complete within %.0f s: %s" % (gather_timeout_s(), exc)) from exc
    if done is False:
        raise GatherTimeout("top-k all-gather did not complete within %.0f s" % gather_timeout_s())


def shard_range(total, rank, world):
    """Contiguous split of range(total): the first total % world ranks get one extra element."""
    q, r = divmod(int(total), int(world))
    start = rank * q + min(rank, r)
    return start, q + (1 if rank < r else 0)


def packed_topk_buffers(rows, k, device):
    """(buf int32 [2, rows, k], ids view int32 [rows, k], scores view float32 [rows, k]): scoring writes straight into the
    two halves, the gather sends `buf` as one message."""
    buf = torch.empty((2, rows, k), dtype=torch.int32, device=device)
    return buf, buf[0], buf[1].view(torch.float32)


def _is_packed(ids, vals):
    return (ids.dtype == torch.int32 and vals.dtype == torch.float32 and ids.is_contiguous() and vals.is_contiguous()
            and ids.shape == vals.shape and ids.untyped_storage().data_ptr() == vals.untyped_storage().data_ptr()
            and vals.data_ptr() == ids.data_ptr() + ids.numel() * 4)


def _pack(ids, vals, rows):
    """[2, rows, k] int32 holding ids and the bit pattern of vals, zero-padded to `rows`."""
    n, k = ids.shape
    if n == rows and _is_packed(ids, vals):
        return torch.as_strided(ids, (2, rows, k), (rows * k, k, 1))
    buf = torch.zeros((2, rows, k), dtype=torch.int32, device=ids.device)
    buf[0, :n] = ids.to(torch.int32)
    buf[1, :n] = vals.to(torch.float32).contiguous().view(torch.int32)
    return buf


def _unpack(gathered, counts, k):
    """gathered int32 [world, 2, mx, k] -> (ids [U, k], vals [U, k]) in user order, per-rank padding dropped."""
    if all(c == gathered.shape[2] for c in counts):
        ids = gathered[:, 0].reshape(-1, k)
        vals = gathered[:, 1].reshape(-1, k)
    else:
        ids = torch.cat([gathered[r, 0, :c] for r, c in enumerate(counts)])
        vals = torch.cat([gathered[r, 1, :c] for r, c in enumerate(counts)])
    return ids, vals.contiguous().view(torch.float32)


def all_gather_topk(ids, vals, counts=None, group=None, force_collective=None):
    """Concatenate per-rank (ids [u_r,k], vals [u_r,k]) into ([U,k], [U,k]) on every rank with one collective.

    counts: list of per-rank row counts (needed when they differ: ranks pad to the maximum and the padding is dropped
    after the gather).  ids / vals that are the two halves of packed_topk_buffers() are sent in place.
    """
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force_collective_setting(force_collective)):
        return ids, vals                           # one rank: the concatenation of one shard is that shard
    world = dist.get_world_size(group)
    k = ids.shape[1]
    if counts is None:
        counts = [ids.shape[0]] * world
    mx = max(counts)
    buf = _pack(ids, vals, mx)
    ordered = _stream_ordered(group)
    if not ordered and buf.is_cuda:                 # host staging of our own (module docstring)
        host = buf.cpu()                            # stream-ordered behind the producer, returns when the copy is done
        out_host = torch.empty((world * 2, mx, k), dtype=torch.int32)
        _wait(dist.all_gather_into_tensor(out_host, host, group=group, async_op=True), ordered)
        out = out_host.to(ids.device)
    else:
        out = torch.empty((world * 2, mx, k), dtype=torch.int32, device=ids.device)    # concatenation along dim 0
        if not ordered:
            _finish_producer(buf)
        _wait(dist.all_gather_into_tensor(out, buf, group=group, async_op=True), ordered)
    return _unpack(out.view(world, 2, mx, k), counts, k)


def sharded_score_topk(model, k, total_users=None, candidates=None, group=None, ws=None, out=None):
    """Every rank scores its user shard of `model` (full catalog) and all ranks receive the merged top-k.

    Returns (ids [U,k], scores [U,k]) for all U users in user order, identical on every rank.
    """
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    device = model._compute_device()
    tower = model._tower(device)
    U = tower.n_users if total_users is None else int(total_users)
    start, count = shard_range(U, rank, world)
    ids, vals = tower.score_topk((start, count), k, candidates, None, check_ids=False, ws=ws, out=out)
    if world == 1:
        return ids, vals
    counts = [shard_range(U, r, world)[1] for r in range(world)]
    return all_gather_topk(ids, vals, counts, group)


class TopkGather:
    """Pipelined form of sharded_score_topk for back-to-back steps.

        pipe = TopkGather(model, k, total_users)
        for _ in range(steps):
            done = pipe.step()          # result of the PREVIOUS step (None on the first call), already gathered
        last = pipe.flush()             # result of the last step

    step() enqueues this rank's scoring + top-k into one of `depth` packed buffers and starts the all-gather of that buffer
    asynchronously; the gather overlaps the next step's kernels.  Results are (ids [U,k], scores [U,k]) in user order.
    A result aliases a pipeline buffer and stays valid only until the NEXT step() call, which scores (and gathers) into
    that same buffer: copy what must live longer.
    depth (or RECO_GATHER_DEPTH, default 2) = buffers in flight: step i returns the result of step i - (depth - 1).  With 2 the
    compute stream waits for gather i - 1 right behind step i's kernels -- a gather that only found free CUs when tile launch i
    ended then holds up tile launch i + 1; with 3 the awaited gather is a whole step older and the wait never bites.
    Without an initialised process group it degenerates to the local call.
    A gather that does not complete raises GatherTimeout (non-RCCL backends) or is ended by the process group's
    watchdog (RCCL): see the module docstring.
    """

    def __init__(self, model, k, total_users=None, candidates=None, group=None, reserve_cus=None, force_collective=None, depth=None):
        self.k, self.candidates, self.group = k, candidates, group
        self.depth = int(os.environ.get("RECO_GATHER_DEPTH", "2") if depth is None else depth)
        if not 2 <= self.depth <= 8:
            raise ValueError("depth must be within [2, 8], got %d" % self.depth)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # one rank: nothing to gather, no collective is issued (force_collective: module docstring)
        self.on = self.world > 1 or (dist.is_initialized() and force_collective_setting(force_collective))
        self.ordered = _stream_ordered(group)
        self.rank = dist.get_rank(group) if self.on else 0
        self.device = model._compute_device()
        self.tower = model._tower(self.device)
        # CUs kept free of tile workgroups while a gather may be in flight (module docstring); only meaningful with peers
        self.reserve_cus = reserved_cus_setting(reserve_cus) if self.on else 0
        if hasattr(self.tower, "set_reserved_cus") and (self.reserve_cus or getattr(self.tower, "reserved_cus", 0)):
            self.tower.set_reserved_cus(self.reserve_cus)
        U = self.tower.n_users if total_users is None else int(total_users)
        self.start, self.count = shard_range(U, self.rank, self.world)
        self.counts = [shard_range(U, r, self.world)[1] for r in range(self.world)]
        mx = max(self.counts)
        n_items = self.tower.n_items if candidates is None else len(candidates)
        self.ws = self.tower.topk_workspace(self.count, n_items, k)
        self.slots = []
        for _ in range(self.depth):
            buf, ids, vals = packed_topk_buffers(mx, k, self.device)
            if self.count < mx:
                buf.zero_()
            gathered = torch.empty((self.world * 2, mx, k), dtype=torch.int32, device=self.device) if self.on else None
            # a backend that is not stream-ordered gathers on the host (module docstring): staging buffers per slot
            staged = self.on and not self.ordered and self.device.type == "cuda"
            host_in = torch.empty((2, mx, k), dtype=torch.int32).pin_memory() if staged else None
            host_out = torch.empty((self.world * 2, mx, k), dtype=torch.int32).pin_memory() if staged else None
            self.slots.append({"buf": buf, "ids": ids[:self.count], "vals": vals[:self.count], "gathered": gathered, "work": None,
                               "host_in": host_in, "host_out": host_out})
        self.i = 0
        self.pending = []                              # slots whose step has been enqueued and not handed out yet, oldest first

    def _collect(self, slot):
        if not self.on:
            return slot["ids"], slot["vals"]
        work, slot["work"] = slot["work"], None
        _wait(work, self.ordered)
        if slot["host_out"] is not None:
            slot["gathered"].copy_(slot["host_out"])
        return _unpack(slot["gathered"].view(self.world, 2, -1, self.k), self.counts, self.k)

    def step(self):
        slot = self.slots[self.i % self.depth]
        self.i += 1
        self.tower.score_topk((self.start, self.count), self.k, self.candidates, None, check_ids=False, ws=self.ws,
                              out=(slot["ids"], slot["vals"]))
        if self.on:
            if slot["host_in"] is not None:
                slot["host_in"].copy_(slot["buf"])          # behind the top-k kernel on the compute stream; done when it returns
                slot["work"] = dist.all_gather_into_tensor(slot["host_out"], slot["host_in"], group=self.group, async_op=True)
            else:
                if not self.ordered:
                    _finish_producer(slot["buf"])
                slot["work"] = dist.all_gather_into_tensor(slot["gathered"], slot["buf"], group=self.group, async_op=True)
        self.pending.append(slot)
        return self._collect(self.pending.pop(0)) if len(self.pending) >= self.depth else None

    def flush(self):
        """Result of the LAST step (older ones still in flight are completed and dropped; step() hands them out one at a time)."""
        done = None
        while self.pending:
            done = self._collect(self.pending.pop(0))
        return done
