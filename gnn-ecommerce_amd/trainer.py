"""One rank's training step on a partition as a fixed launch sequence -- and, optionally, as replayed HIP graphs.

The reference's loop body (src/train_lightgcn.py:130-147) is ``zero_grad -> forward -> bpr * size + reg -> backward ->
optimizer.step()`` through autograd.  ``partition.partitioned_bpr_loss`` + ``optim.Adam(row_ranges=...)`` keep that shape
on a partition.  At 8 ranks a step's kernels take about 1.3 ms while issuing its ~115 launches from Python takes 2 ms:
the step is bound by the host.  ``PartitionedTrainer`` runs the SAME launches (``partition.step_forward`` /
``step_backward``, ``lgc_adam_step_hp`` over the rows the rank owns) without autograd in between and with every shape
fixed, which makes the step capturable: with ``graphs=True`` the launch sequence is recorded once as HIP graphs --
one per stretch of work BETWEEN two collectives, never a collective inside a capture -- and every later step replays
them around the eager all-reduce calls (``partition.Comm`` is the seam).  What changes from step to step lives in device
buffers the host refreshes before the replay: the batch's ids (3 x B int64) and Adam's six scalars (24 bytes).
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import _native
from .partition import Comm, PartitionedPropagator, check_batch, step_backward, step_forward


class _RecordingComm(Comm):
    """``Comm`` that cuts the capture at every collective: end the graph being captured, note the collective, run it
    eagerly, begin the next graph.  Handles are indices into ``self.handles`` so that a replay can wait for the work the
    REPLAYED start returned."""

    def __init__(self, inner: Comm, owner: "PartitionedTrainer"):
        super().__init__(inner.world, inner.group)
        self.inner, self.owner = inner, owner

    def start(self, block: Tensor):
        slot = self.owner._cut(("start", block))
        return ("slot", slot)

    def wait(self, handle) -> None:
        if handle is not None:
            self.owner._cut(("wait", handle[1]))

    def reduce_now(self, t: Tensor) -> None:
        self.owner._cut(("reduce", t))


class PartitionedTrainer:
    """``step(users, pos, neg)`` = one training step of this rank for one GLOBAL batch (every rank is given the same
    ``users, pos, neg``: int64 ``[batch]`` device tensors): forward on the partition, BPR + regulariser of the triples this
    rank owns, seeded backward with the per-hop item-block exchange, Adam over ``pp.owned_row_ranges()``.  Returns a
    device tensor ``[bpr, reg, loss]`` (global values, summed over the ranks; no host sync).  ``weight`` is updated in
    place; its rows other ranks own are never read or written.

    ``graphs=True``: the first call records the step (after ``warmup`` eager steps that build every lazy plan and
    scratch buffer), later calls replay it.  Same launches, same order, same bits as the eager step.  ``graphs="full"``:
    the whole step as ONE graph with its collectives inside (RCCL's collectives are capturable; an eager collective costs a
    rank two cross-stream hand-offs, ~12-23 us each, and a step has a dozen) -- only over ``nccl``; if the capture fails on
    any rank, every rank falls back to the segmented recording together (one MIN all-reduce)."""

    def __init__(self, pp: PartitionedPropagator, weight: Tensor, alphas: Sequence[float], lr: float = 0.005,
                 decay: float = 1e-4, batch: int = 1024, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                 graphs: bool = False, warmup: int = 2):
        if weight.dtype != torch.float32 or weight.dim() != 2 or not weight.is_contiguous():
            raise TypeError("weight must be a contiguous 2-D fp32 table")
        self.pp, self.w = pp, weight.detach()
        self.alphas = tuple(float(a) for a in alphas)
        self.lr, self.decay, self.batch, self.betas, self.eps = float(lr), float(decay), int(batch), betas, float(eps)
        dev = self.w.device
        self.m, self.v = torch.zeros_like(self.w), torch.zeros_like(self.w)
        self.t = 0
        self.ids = torch.zeros((3, self.batch), dtype=torch.int64, device=dev)       # users | pos | neg of the current batch
        self.hyper = torch.zeros(6, dtype=torch.float32, device=dev)                 # lgc_adam_step_hp's scalars
        # staged through a RING of pinned rows: the asynchronous copy of step t reads its row when the stream gets there, so
        # the host may not overwrite it before -- a row is reused only after the copy that read it has completed
        self._ring = 32
        self._hyper_host = torch.zeros((self._ring, 6), dtype=torch.float32)
        self._hyper_done = [None] * self._ring
        if dev.type == "cuda":
            self._hyper_host = self._hyper_host.pin_memory()
        if pp.u1 <= pp.u0:
            raise ValueError("this rank owns no user rows: a partition with more ranks than users cannot train")
        self.stats = torch.zeros(3, dtype=torch.float32, device=dev)
        self.use_graphs = bool(graphs) and dev.type == "cuda"
        self.full_graph = graphs == "full" and self.use_graphs
        self._full = None                          # the whole step as one graph (graphs="full"), once recorded
        self.full_error: Optional[str] = None
        self.warmup = int(warmup)
        self._actions: Optional[list] = None       # recorded: ("graph", g) | ("start", tensor) | ("wait", slot) | ("reduce", t)
        self._recording = None
        self._steps_seen = 0

    # -- the step as straight-line code; collectives go through self.pp.comm -----------------------------------
    def _body(self) -> None:
        pp = self.pp
        users, pos, neg = self.ids[0], self.ids[1], self.ids[2]
        local, bpr_local, reg_users, reg_items, saved = step_forward(pp, self.w, self.alphas, users, pos, neg, self.decay)
        grad = step_backward(pp, saved, self.alphas, None, self.decay / self.batch, zero_foreign=False)
        self._adam(grad)
        part = torch.stack([bpr_local, reg_users])
        pp.comm.reduce_now(part)
        self.stats[0] = part[0]
        self.stats[1] = part[1] + reg_items
        self.stats[2] = self.stats[0] + self.stats[1]

    def _adam(self, grad: Tensor) -> None:
        ops = self.pp.ops
        width = self.w.size(1)
        for lo, hi in self.pp.owned_row_ranges():
            if hi > lo:
                ops.adam_rows(self.w, grad, self.m, self.v, lo * width, hi * width, self.hyper)

    def _refresh_hyper(self) -> None:
        self.t += 1
        b1, b2 = self.betas
        slot = self.t % self._ring
        if self._hyper_done[slot] is not None:
            self._hyper_done[slot].synchronize()       # the copy that last read this row (32 steps ago) has completed
        h = self._hyper_host[slot]
        h[0], h[1], h[2], h[3] = 1.0 - b1, b2, 1.0 - b2, self.eps
        h[4] = self.lr / (1.0 - b1 ** self.t)
        h[5] = math.sqrt(1.0 - b2 ** self.t)
        self.hyper.copy_(h, non_blocking=True)
        if self.hyper.is_cuda:
            ev = self._hyper_done[slot] or torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.hyper.device))
            self._hyper_done[slot] = ev

    # -- recording ------------------------------------------------------------------------------------------------
    def _cut(self, action):
        """Called by _RecordingComm at a collective: close the graph being captured, note and RUN the collective, open the
        next graph.  Returns the slot of a started exchange."""
        rec = self._recording
        self._end_capture(rec)
        slot = None
        if action[0] == "start":
            slot = len(rec["handles"])
            rec["handles"].append(rec["inner"].start(action[1]))
            rec["actions"].append(("start", action[1], slot))
        elif action[0] == "wait":
            rec["inner"].wait(rec["handles"][action[1]])
            rec["actions"].append(("wait", action[1]))
        else:
            rec["inner"].reduce_now(action[1])
            rec["actions"].append(("reduce", action[1]))
        rec["graph"] = torch.cuda.CUDAGraph()
        rec["graph"].capture_begin(pool=rec["pool"], capture_error_mode="thread_local")
        return slot

    @staticmethod
    def _end_capture(rec) -> None:
        """Close the graph being captured and keep it.  Two collectives in a row leave an empty graph between them (torch
        warns about those; replaying one is a no-op)."""
        import warnings
        with warnings.catch_warnings():
            warnings.filterwarnings("ignore", message="The CUDA Graph is empty")
            rec["graph"].capture_end()
        rec["actions"].append(("graph", rec["graph"]))

    def _record(self) -> None:
        # capture_error_mode "thread_local": the exchange started before a segment may still be running in the
        # communication backend's own threads (gloo copies through the host; RCCL's watchdog polls events) while the next
        # segment is being captured -- legal, but under the default "global" mode any runtime call of ANOTHER thread
        # invalidates the capture.
        from . import propagate
        dev = self.w.device
        inner = self.pp.comm
        rec = {"inner": inner, "actions": [], "handles": [], "pool": torch.cuda.graph_pool_handle(),
               "graph": torch.cuda.CUDAGraph()}
        self._recording = rec
        stream = torch.cuda.Stream(dev)
        stream.wait_stream(torch.cuda.current_stream(dev))
        log, propagate.HOP_EVENT_LOG = propagate.HOP_EVENT_LOG, None
        self.pp.comm = _RecordingComm(inner, self)
        try:
            with torch.cuda.stream(stream):
                rec["graph"].capture_begin(pool=rec["pool"], capture_error_mode="thread_local")
                try:
                    self._body()
                finally:
                    self._end_capture(rec)
        finally:
            self.pp.comm = inner
            propagate.HOP_EVENT_LOG = log
            self._recording = None
        torch.cuda.current_stream(dev).wait_stream(stream)
        self._actions, self._n_slots = rec["actions"], len(rec["handles"])

    def _record_full(self) -> bool:
        """The whole step, collectives included, as ONE graph.  Returns whether EVERY rank recorded it (else nothing is kept
        and the caller records the segmented form).  The capture runs on a side stream and is always ended."""
        import warnings
        import torch.distributed as dist
        from . import propagate
        dev = self.w.device
        comm = self.pp.comm
        active = getattr(comm, "active", self.pp.world > 1)
        ok = 1
        # (no process group at all: a measurement harness with the collectives stubbed out -- nothing there to capture)
        backend = dist.get_backend(self.pp.group) if dist.is_initialized() else None
        if active and backend not in (None, "nccl"):
            self.full_error, ok = f"backend {backend} cannot be captured", 0
        graph = torch.cuda.CUDAGraph()
        if ok:
            torch.cuda.synchronize(dev)
            current, side = torch.cuda.current_stream(dev), torch.cuda.Stream(dev)
            side.wait_stream(current)
            log, propagate.HOP_EVENT_LOG = propagate.HOP_EVENT_LOG, None
            try:
                with torch.cuda.stream(side):
                    graph.capture_begin(capture_error_mode="thread_local")
                    try:
                        self._body()
                    except Exception as exc:                  # noqa: BLE001
                        self.full_error, ok = f"{type(exc).__name__}: {exc}", 0
                    finally:
                        try:
                            with warnings.catch_warnings():
                                warnings.filterwarnings("ignore", message="The CUDA Graph is empty")
                                graph.capture_end()
                        except Exception as exc:              # noqa: BLE001
                            self.full_error, ok = self.full_error or f"{type(exc).__name__}: {exc}", 0
            except Exception as exc:                          # noqa: BLE001
                self.full_error, ok = self.full_error or f"{type(exc).__name__}: {exc}", 0
            finally:
                propagate.HOP_EVENT_LOG = log
            current.wait_stream(side)
        if active and backend is not None:
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.pp.group)
            ok = int(flag.item())
        if ok:
            self._full = graph
        return bool(ok)

    def _replay(self) -> None:
        if self._full is not None:
            self._full.replay()
            return
        comm = self.pp.comm
        handles = [None] * self._n_slots
        for act in self._actions:
            kind = act[0]
            if kind == "graph":
                act[1].replay()
            elif kind == "start":
                handles[act[2]] = comm.start(act[1])
            elif kind == "wait":
                comm.wait(handles[act[1]])
            else:
                comm.reduce_now(act[1])

    # -- public ---------------------------------------------------------------------------------------------------
    def step(self, users: Tensor, pos: Tensor, neg: Tensor) -> Tensor:
        check_batch(self.w, users, pos, neg)
        if users.numel() != self.batch:
            raise ValueError(f"this trainer was built for batches of {self.batch} triples")
        self.ids[0].copy_(users, non_blocking=True)
        self.ids[1].copy_(pos, non_blocking=True)
        self.ids[2].copy_(neg, non_blocking=True)
        self._refresh_hyper()
        with torch.no_grad():
            if not self.use_graphs:
                self._body()
            elif self._actions is not None or self._full is not None:
                self._replay()
            elif self._steps_seen < self.warmup:
                self._body()                       # eager: builds every lazy work plan and scratch buffer first
            else:
                if not (self.full_graph and self._record_full()):
                    self._record()                 # the recording run IS this step (captured work does not execute ...
                self._replay()                     # ... so it is replayed once right away)
        self._steps_seen += 1
        return self.stats

    @property
    def graph_launches(self) -> int:
        if self._full is not None:
            return 1
        return 0 if self._actions is None else sum(1 for a in self._actions if a[0] == "graph")
