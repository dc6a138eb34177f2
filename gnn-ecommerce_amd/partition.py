"""Propagation over a bipartite graph partitioned across the GPUs of one node.

One process per GPU.  Users are cut into ``world`` contiguous ranges balanced by nnz (not by
count); the item block of every table is replicated.  One hop on rank p, for its range [u0, u1):

    I-step   partial[i]  = a * sum over edges (u -> i), u in [u0, u1), of val * x[u]   (all items;
                           rank 0 adds b * r[i], so the sum over ranks is the finished item block)
    exchange all-reduce(sum) of the contiguous [n_items, D] item block   -- 13.97 MB at D=64
    U-step   y[u]        = a * sum over edges (i -> u) of val * x[i] + b * r[u]      (own users)

The I-step is launched first and its all-reduce runs on RCCL's stream while the U-step (which needs
only the replicated item rows of x) AND the next hop's I-step (which needs only this rank's own user
rows) compute, so the exchange is hidden behind local work.
User rows never leave their owner during propagation; ``gather_users`` assembles the full table
only when a caller really needs it (the reference's ``get_embedding`` contract).

The literal reading of the task ("all-reduce the layer output") would move the whole [N, D]
table, 433.6 MB per hop -- more than one GPU's whole hop; see DESIGN.md.

The arithmetic lives behind a small ``ops`` protocol so that the partition/exchange logic can be
exercised on CPU ranks over gloo with a test double (tests/test_partition_gloo.py); the only
implementation shipped here is ``HipOps`` -- there is no CPU compute path in the product.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
from torch import Tensor

from . import _native
from .graph import CHUNK_LEN, SHORT_MAX, Operator, PropGraph, build_row_plan
from .propagate import DeviceOps


import warnings
warnings.filterwarnings("ignore", message=r"index_reduce\(\) is in beta")    # used once per step (propagate_sum, listed item rows)

COMM_FORCE = os.environ.get("LGCN_COMM_FORCE", "0") == "1"
ITEM_SHORT_MAX = int(os.environ.get("LGCN_ITEM_SHORT_MAX", "32"))   # 8 / 16 / 32: 114 / 113 / 112 us per hop at world 8, 202 -> 190 at world 4


def balanced_user_ranges(user_degree: Tensor, world: int) -> List[Tuple[int, int]]:
    """Contiguous user ranges with (nearly) equal edge counts.  Pure index arithmetic.

    Cut k is placed at the first user whose cumulative degree reaches k/world of the total, so
    every range is non-decreasing and the ranges tile [0, n_users) exactly.
    """
    if world < 1:
        raise ValueError("world must be >= 1")
    n_users = int(user_degree.numel())
    csum = torch.cumsum(user_degree.to(torch.int64), 0)
    total = int(csum[-1].item()) if n_users else 0
    targets = torch.tensor([(total * k + world - 1) // world for k in range(1, world)], dtype=torch.int64,
                           device=csum.device)
    cuts = torch.searchsorted(csum, targets, right=False) + 1 if n_users else targets
    cuts = [0] + [min(int(c), n_users) for c in cuts.tolist()] + [n_users]
    for k in range(1, len(cuts)):
        cuts[k] = max(cuts[k], cuts[k - 1])
    return [(cuts[k], cuts[k + 1]) for k in range(world)]


def check_bipartite(edge_index: Tensor, n_users: int, n_items: int) -> None:
    """Every edge must join a user (< n_users) and an item (>= n_users): src/utils_v2.py:146-165 layout."""
    src, dst = edge_index[0], edge_index[1]
    n = n_users + n_items
    ok = (((src < n_users) & (dst >= n_users)) | ((src >= n_users) & (dst < n_users))) & (src >= 0) & (dst >= 0) \
        & (src < n) & (dst < n)
    if not bool(ok.all().item()):
        raise ValueError("partitioned propagation needs a bipartite user|item edge list (ids < n_users are users)")


class Comm:
    """The three places a rank talks to the others, as one object the callers go through: ``start`` an all-reduce of an
    item block (asynchronous; returns a handle), ``wait`` for it (the current stream waits, not the host), and
    ``reduce_now`` for small tensors that are read right away.  ``PartitionedTrainer`` substitutes a recording version
    that cuts its HIP-graph capture at exactly these points (no collective is ever captured)."""

    def __init__(self, world: int, group: Optional[dist.ProcessGroup]):
        self.world, self.group = world, group
        # LGCN_COMM_FORCE=1: a partition of ONE rank goes through the backend's collectives as well (they are identities
        # there) -- how tests/test_partition_rccl.py drives the real RCCL process group, its stream hand-offs and its
        # watchdog thread beside the HIP-graph capture, on a box with one GPU
        self.active = world > 1 or COMM_FORCE

    def start(self, block: Tensor):
        if not self.active:
            return None
        return dist.all_reduce(block, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def wait(self, handle) -> None:
        if handle is not None:
            handle.wait()

    def reduce_now(self, t: Tensor) -> None:
        if self.active:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)


class HipOps(DeviceOps):
    """The shipped implementation of the arithmetic: HIP kernels through the C ABI (propagate.DeviceOps) plus the graph
    builders of a partition."""

    def build(self, edge_index: Tensor, edge_weight: Optional[Tensor], num_nodes: int, normalize: bool,
              keep_edge_values: bool = False) -> PropGraph:
        return PropGraph(edge_index, edge_weight, num_nodes, normalize, keep_edge_values=keep_edge_values)

    def restrict(self, op: Operator, row_begin: int, row_end: int, short_max: int = SHORT_MAX,
                 sweep_cols: Optional[Tuple[int, int]] = None) -> Operator:
        """The same CSR, work plan limited to rows [row_begin, row_end).  ``sweep_cols``: the column range all entries
        of those rows lie in (a rank's item rows only reference its own users) -- lets a large enough slice run as a
        band sweep inside that range."""
        return Operator.build(op.n_rows, op.rowptr, op.entries, row_begin, row_end, short_max, CHUNK_LEN,
                              sweep_cols=sweep_cols)

    def apply(self, op: Operator, x: Tensor, out: Tensor, a: float = 1.0, r: Optional[Tensor] = None, b: float = 0.0) -> None:
        op.apply(x, out, a=a, r=r, b=b)


class PartitionedPropagator:
    """K-hop propagate + layer sum for rank ``rank`` of ``world`` (see module docstring)."""

    def __init__(self, edge_index: Tensor, edge_weight: Optional[Tensor], n_users: int, n_items: int,
                 rank: int, world: int, group: Optional[dist.ProcessGroup] = None, normalize: bool = True,
                 ops=None):
        self.ops = ops if ops is not None else HipOps()
        self.n_users, self.n_items, self.rank, self.world, self.group = n_users, n_items, rank, world, group
        self.num_nodes = n = n_users + n_items
        check_bipartite(edge_index, n_users, n_items)
        # global build on every rank (every rank holds the COO): the per-edge values depend on the
        # GLOBAL degrees accumulated in GLOBAL edge order, so they cannot be derived
        # from a local slice
        full = self.ops.build(edge_index, edge_weight, n, normalize, keep_edge_values=True)
        rowptr = full.forward_op.rowptr
        user_degree = (rowptr[1:n_users + 1] - rowptr[:n_users])
        self.ranges = balanced_user_ranges(user_degree, world)
        self.u0, self.u1 = self.ranges[rank]
        # U-step: rows [u0, u1) of the global CSR (targets = my users, columns = items)
        self.user_op = self.ops.restrict(full.forward_op, self.u0, self.u1)
        # I-step: item rows restricted to MY users as sources, same per-edge values
        src, dst = edge_index[0], edge_index[1]
        mine = (dst >= n_users) & (src >= self.u0) & (src < self.u1)
        local = self.ops.build(edge_index[:, mine].contiguous(), full.edge_values[mine].contiguous(), n,
                               normalize=False)
        # A rank's slice of an item row is short (mean degree / world entries): slices of up to ITEM_SHORT_MAX entries
        # go through the tiled kernels, longer ones through chunks; a dense enough slice (world 2) may sweep.
        self.item_op = self._restrict_items(local.forward_op)
        self.local_nnz = int(mine.sum().item()) * 2
        self._keep = (full, local)
        self._coo = (edge_index, full.edge_values)
        self._transposed = None
        self._table_cache = {}
        self.comm = Comm(world, group)

    def _restrict_items(self, op):
        """Item rows of a local operator: every column is one of this rank's own users."""
        try:
            return self.ops.restrict(op, self.n_users, self.num_nodes, ITEM_SHORT_MAX, sweep_cols=(self.u0, self.u1))
        except TypeError:                                  # arithmetic test doubles without the sweep argument
            return self.ops.restrict(op, self.n_users, self.num_nodes, ITEM_SHORT_MAX)

    # -- hops ----------------------------------------------------------------------------------
    # A hop has two local pieces and one exchange:
    #   item step   a * (partial sums for all item rows from this rank's user rows of x); rank 0 also adds
    #               the epilogue term b * r[items], so that the SUM over ranks is the finished item block
    #               (the exchange is linear: no separate epilogue pass over the item block afterwards)
    #   exchange    all-reduce of the item block (in place in ``out``)
    #   user step   this rank's user rows of ``out`` from the (replicated) item rows of x
    # Only the user step of the NEXT hop needs the reduced item block; the next hop's item step needs
    # just this rank's own user rows.  propagate_sum therefore keeps the all-reduce of hop l in flight
    # across hop l's user step and hop l+1's item step and waits for it right before hop l+1's user step.
    def _item_step(self, item_op, x: Tensor, out: Tensor, a: float, r: Optional[Tensor], b: float):
        self.ops.apply(item_op, x, out, a, r if self.rank == 0 else None, b)
        return self.comm.start(out[self.n_users:])

    def _finish_items(self, work) -> None:
        self.comm.wait(work)

    def hop(self, x: Tensor, out: Tensor, a: float, r: Optional[Tensor], b: float) -> Tensor:
        """One complete hop (used on its own by tests and by callers that need a single LGConv)."""
        work = self._item_step(self.item_op, x, out, a, r, b)
        self.ops.apply(self.user_op, x, out, a, r, b)                        # overlaps the exchange
        self._finish_items(work)
        return out

    def _transposed_ops(self):
        """(user rows, item rows) of A^T for this rank, with the SAME per-edge values (exact adjoint):
        (A^T g)[u] = sum over edges (u -> i) of val * g[i]   for own users u,
        (A^T g)[i] = sum over edges (i -> u), u own, of val * g[u]   (partial: all-reduced like the forward)."""
        if self._transposed is None:
            ei, vals = self._coo
            src, dst = ei[0], ei[1]
            n, nu = self.num_nodes, self.n_users
            own_src = (src >= self.u0) & (src < self.u1)                      # edges u -> i, u own
            own_dst = (dst >= self.u0) & (dst < self.u1)                      # edges i -> u, u own
            g_user = self.ops.build(ei[:, own_src].contiguous(), vals[own_src].contiguous(), n, normalize=False)
            g_item = self.ops.build(ei[:, own_dst].contiguous(), vals[own_dst].contiguous(), n, normalize=False)
            self._transposed = (self.ops.restrict(g_user.transpose_op, self.u0, self.u1),
                                self._restrict_items(g_item.transpose_op), g_user, g_item)
        return self._transposed[0], self._transposed[1]

    def _lincomb(self, y: Tensor, terms) -> None:
        lc = getattr(self.ops, "lincomb", None)
        if lc is not None:
            lc(y, terms)
        else:                                                   # test doubles without a lincomb: the same sum in torch
            acc = terms[0][1] * terms[0][0]
            for c, t in terms[1:]:
                acc = acc + t * c
            y.copy_(acc)

    def _tables(self, like: Tensor, count: int, tag) -> List[Tensor]:
        """``count`` internal [N, D] tables, allocated once per (width, count, direction) and reused by every call:
        no allocation inside the hop loop."""
        key = (like.size(1), like.dtype, like.device, count, tag)
        got = self._table_cache.get(key)
        if got is None:
            got = [torch.empty_like(like) for _ in range(count)]
            self._table_cache[key] = got
        return got

    def owned_row_ranges(self) -> List[Tuple[int, int]]:
        """The rows of an [N, D] table this rank owns: its user range and the (replicated) item block -- what
        ``optim.Adam(row_ranges=...)`` updates on this rank."""
        return [(self.u0, self.u1), (self.n_users, self.num_nodes)]

    def propagate_sum(self, x0: Tensor, alphas: Sequence[float], transpose: bool = False,
                      zero_foreign_rows: bool = False, final_rows: Optional[Tensor] = None,
                      final_item_rows: Optional[Tensor] = None) -> Tensor:
        """sum_l alpha_l A^l x0 on this rank's rows (own users + all items), in the bipartite evaluation of
        propagate.bipartite_sum: with x_l = A^l x0,
            x_l[items]  = all-reduce( item step over OWN users of x_{l-1} )          l = 1 .. K
            x_l[users]  = user step of x_{l-1}[items]   (own users)                  l = 1 .. K-1
            out[users]  = alpha_0 x0[users] + user step of ( sum_{l>=1} alpha_l x_{l-1}[items] )
            out[items]  = sum_l alpha_l x_l[items]
        so no hop reads an epilogue row except the last user step, and the K-th user table is never written.
        The all-reduce of layer l is started right after its item step and waited for only where x_l[items] is first
        read -- the user step of layer l+1 (or the final sums) -- i.e. it is in flight across one user step and the next
        item step.  ``transpose``: the same with A^T (the backward pass).  ``zero_foreign_rows``: other ranks' user rows
        of the result are zero instead of undefined.  ``final_rows`` (int64 node ids): the caller will only read these
        user rows of the result (plus the item block) -- the last user step is computed for them only (lgc_spmm_rows),
        as in propagate.bipartite_sum: a training step scores a few thousand pairs.  ``final_item_rows`` (with
        ``final_rows``; int64 item node ids): ... and of the item block only these rows -- the LAST item step is then
        computed for the listed rows only (lgc_spmm_rows_split into a [len, D] table in list order) and the exchange of
        that layer is an all-reduce of that table (0.5 MB for 2 x 1024 rows) instead of the 14 MB item block; every other
        item row of the result is left undefined."""
        from . import propagate
        k = len(alphas) - 1
        if k == 0:
            return x0 * alphas[0]
        x0 = x0.contiguous()
        nu = self.n_users
        user_op, item_op = self._transposed_ops() if transpose else (self.user_op, self.item_op)
        log = propagate.HOP_EVENT_LOG if x0.is_cuda else None
        tables = [x0] + self._tables(x0, k + 1, transpose)          # x_1 .. x_K, and the mix table
        mix = tables.pop()
        out = torch.zeros_like(x0) if zero_foreign_rows else torch.empty_like(x0)
        pending = [None] * (k + 1)                                   # all-reduce of x_l[items]
        listed_items = final_item_rows is not None and final_rows is not None and propagate.SCORED_ITEM_ROWS_ONLY
        if listed_items and hasattr(item_op, "listed_rows_pay"):            # (arithmetic test doubles have no such method)
            listed_items = item_op.listed_rows_pay(2 * final_item_rows.numel())
        uniform = all(a == alphas[0] for a in alphas)
        marks = []
        for layer in range(1, k + 1):
            if log is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                marks.append(ev)
            prev, cur = tables[layer - 1], tables[layer]
            if layer == k and listed_items:
                # partial sums of the listed item rows over OWN users, in list order; summed over the ranks below
                # (ids clamped into the item block like the gathers of the scoring launch: every row it reads is computed)
                listed = final_item_rows.clamp(nu, self.num_nodes - 1)
                part = torch.zeros((listed.numel(), x0.size(1)), dtype=x0.dtype, device=x0.device)
                self.ops.apply_rows(item_op, listed, prev, part, 1.0, None, 0.0, split=True, compact=True)
                pending[layer] = self.comm.start(part)
            else:
                pending[layer] = self._item_step(item_op, prev, cur, 1.0, None, 0.0)     # needs only OWN user rows of x_{l-1}
            self._finish_items(pending[layer - 1])                                   # x_{l-1}[items] is read from here on
            if layer < k:
                self.ops.apply(user_op, prev, cur, 1.0, None, 0.0)
            else:
                # (the two weighted sums over the 14 MB item blocks, 17 us each, on a second stream beside the last item /
                # user step: 119.7 vs 107.1 us per hop at world 8 -- the cross-stream events cost more than they hide)
                self._lincomb(mix[nu:], [(alphas[l], tables[l - 1][nu:]) for l in range(1, k + 1)])
                if final_rows is None:
                    self.ops.apply(user_op, mix, out, 1.0, x0, alphas[0])
                else:
                    self.ops.apply_rows(user_op, final_rows, mix, out, 1.0, x0, alphas[0])
                self._finish_items(pending[k])
                if listed_items:
                    # out[i] = sum_{l<K} alpha_l x_l[i] + alpha_K x_K[i] for the listed rows: the first sum is ``mix`` itself
                    # when the alphas are equal (the same chain of adds as below)
                    if uniform:
                        rest = mix
                    else:
                        rest = cur
                        self._lincomb(rest[nu:], [(alphas[l], tables[l][nu:]) for l in range(0, k)])
                    # a repeated id holds the same sum at each of its list positions up to the association the all-reduce
                    # used for that position (a ring reduces different chunks of the buffer in different rank orders):
                    # taking the elementwise maximum over the repeats makes the row independent of the order of the writes
                    out.index_reduce_(0, listed, torch.add(rest[listed], part, alpha=alphas[k]), "amax", include_self=False)
                elif uniform:
                    # equal alphas (the reference's 1 / (K + 1)): sum_{l<=K} a x_l = mix + a x_K, the same chain of adds
                    # (((a x_0 + a x_1) + a x_2) + a x_3), two 14 MB reads instead of K + 1
                    self._lincomb(out[nu:], [(1.0, mix[nu:]), (alphas[k], tables[k][nu:])])
                else:
                    self._lincomb(out[nu:], [(alphas[l], tables[l][nu:]) for l in range(0, k + 1)])
        if log is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks.append(ev)
            log.extend(zip(marks[:-1], marks[1:]))
        return out

    def seeded_transpose_sum(self, rows: Tensor, vals: Tensor, alphas: Sequence[float], extra=None,
                             zero_rows=None) -> Tensor:
        """sum_l alpha_l (A^T)^l g on this rank's rows for a gradient g given by its non-zero rows (own users + items; the
        item rows complete, i.e. already summed over the ranks): propagate.seeded_sum on the rank's local halves of A^T,
        every hop's item block all-reduced like the forward pass."""
        from . import propagate
        user_t, item_t = self._transposed_ops()

        def exchange(block: Tensor):
            return _ExchangeHandle(self.comm, self.comm.start(block))

        return propagate.seeded_sum(user_t, item_t, self.user_op, self.n_users, rows, vals, alphas, self.num_nodes, extra,
                                    ops=self.ops, exchange=exchange, zero_rows=zero_rows)

    def gather_users(self, table: Tensor) -> Tensor:
        """Fill every rank's user rows of ``table`` from their owners (all ranks end with the full table): ONE
        all-gather of the ranges padded to the longest one."""
        if not getattr(self.comm, "active", self.world > 1):
            return table
        longest = max(hi - lo for lo, hi in self.ranges)
        dim = table.size(1)
        mine = torch.zeros((longest, dim), dtype=table.dtype, device=table.device)
        mine[: self.u1 - self.u0] = table[self.u0:self.u1]
        everyone = torch.empty((self.world * longest, dim), dtype=table.dtype, device=table.device)
        dist.all_gather_into_tensor(everyone, mine, group=self.group)
        for owner, (lo, hi) in enumerate(self.ranges):
            if owner != self.rank and hi > lo:
                table[lo:hi] = everyone[owner * longest: owner * longest + (hi - lo)]
        return table


class RecordedForward:
    """``pp.propagate_sum(x0, alphas)`` for one fixed input table, recorded ONCE as a HIP graph -- collectives included --
    and replayed: ``rec = RecordedForward(pp, x0, alphas); out = rec()``.

    Why: a rank's hop at 8 ranks is ~95 us of kernels, and every collective torch issues eagerly costs two cross-stream
    hand-offs (launch stream -> RCCL's stream -> launch stream: event record, barrier packet, event wait).  Measured on one
    MI355X with the rank's collectives going to a real one-rank ``nccl`` group (identities: no byte moves): 93.6 us per hop
    with the collectives stubbed out, 116.6 eager, 95.5-97.7 with the whole forward -- all-reduces and their waits as graph
    edges -- replayed (profiles/r04k).  RCCL's collectives are capturable like NCCL's; the capture itself executes nothing.

    The result tensor is owned by the graph: every call overwrites and returns the SAME tensor (valid rows: own users + all
    items).  ``x0`` is read in place at every replay (update it in place between calls).  Only attempted over ``nccl`` (gloo
    moves CUDA tensors through the host from its own threads: not capturable).  If the capture fails on ANY rank, every rank
    falls back to the eager forward -- the ranks agree on that through one MIN all-reduce, so they can never disagree about
    who replays and who launches eagerly."""

    def __init__(self, pp: "PartitionedPropagator", x0: Tensor, alphas: Sequence[float], warmup: int = 2):
        self.pp, self.x0, self.alphas = pp, x0, tuple(float(a) for a in alphas)
        self.graph, self.out, self.error = None, None, None
        from . import propagate
        for _ in range(max(warmup, 1)):                      # lazy plans, scratch tables, the communicator's first use
            pp.propagate_sum(x0, self.alphas)
        ok = 1
        active = getattr(pp.comm, "active", pp.world > 1)
        # only a backend whose collectives are stream operations can be captured: gloo moves CUDA tensors through the host
        # from its own threads, and an attempt leaves the launch stream in a broken capture
        backend = dist.get_backend(pp.group) if dist.is_initialized() else None      # None: collectives stubbed out (a harness)
        capturable = not active or backend in (None, "nccl")
        if x0.is_cuda and RECORD_FORWARD and capturable:
            torch.cuda.synchronize(x0.device)
            log, propagate.HOP_EVENT_LOG = propagate.HOP_EVENT_LOG, None     # timing events cannot be recorded into a graph
            # The capture runs on a side stream and is ended in a ``finally``: whatever goes wrong inside, this thread is
            # back on its own stream, which never was in capture mode, before anything else is launched.
            graph, out = torch.cuda.CUDAGraph(), None
            current = torch.cuda.current_stream(x0.device)
            side = torch.cuda.Stream(x0.device)
            side.wait_stream(current)
            try:
                with torch.cuda.stream(side):
                    # thread_local: the backend's watchdog thread keeps polling its events while this thread captures
                    graph.capture_begin(capture_error_mode="thread_local")
                    try:
                        out = pp.propagate_sum(x0, self.alphas)
                    except Exception as exc:                  # noqa: BLE001 -- whatever the backend raises: eager it is
                        self.error, ok = f"{type(exc).__name__}: {exc}", 0
                    finally:
                        try:
                            with warnings.catch_warnings():   # (a capture that failed at its first launch is empty)
                                warnings.filterwarnings("ignore", message="The CUDA Graph is empty")
                                graph.capture_end()
                        except Exception as exc:              # noqa: BLE001 -- an invalidated capture ends with an error
                            self.error, ok = self.error or f"{type(exc).__name__}: {exc}", 0
            except Exception as exc:                          # noqa: BLE001 -- capture_begin itself
                self.error, ok = self.error or f"{type(exc).__name__}: {exc}", 0
            finally:
                propagate.HOP_EVENT_LOG = log
            current.wait_stream(side)
            if ok:
                self.graph, self.out = graph, out
        else:
            ok = 0
            if not capturable:
                self.error = f"backend {backend} cannot be captured"
        if active and backend is not None:                    # every rank replays, or none does
            flag = torch.tensor([ok], dtype=torch.int32, device=x0.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=pp.group)
            ok = int(flag.item())
        if not ok:
            self.graph, self.out = None, None

    @property
    def recorded(self) -> bool:
        return self.graph is not None

    def __call__(self) -> Tensor:
        if self.graph is None:
            return self.pp.propagate_sum(self.x0, self.alphas)
        self.graph.replay()
        return self.out


# "0": RecordedForward never records (the eager forward, as before)
RECORD_FORWARD = os.environ.get("LGCN_RECORD_FORWARD", "1") == "1"


class _ExchangeHandle:
    """What PartitionedPropagator hands propagate.seeded_sum for one item block: the all-reduce in flight."""

    def __init__(self, comm, work):
        self.comm, self.work = comm, work

    def wait(self) -> None:
        self.comm.wait(self.work)


class _PartitionedSum(torch.autograd.Function):
    """Differentiable ``PartitionedPropagator.propagate_sum``.  The item rows of the output are replicated
    on every rank and each rank only sees the loss terms of ITS pairs, so the backward pass first sums the
    item block of the incoming gradient over the ranks (one all-reduce, 13.97 MB at D=64) and then runs the
    same pipelined hops on A^T.  Returned gradient: own user rows + all item rows (identical on every
    rank); other ranks' user rows are zero."""

    @staticmethod
    def forward(ctx, x0: Tensor, pp: "PartitionedPropagator", alphas: tuple) -> Tensor:
        ctx.pp, ctx.alphas = pp, alphas
        return pp.propagate_sum(x0.detach(), alphas)

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        pp = ctx.pp
        g = grad_out.contiguous().clone()
        pp.comm.reduce_now(g[pp.n_users:])
        return pp.propagate_sum(g, ctx.alphas, transpose=True, zero_foreign_rows=True), None, None


def partitioned_embedding(pp: PartitionedPropagator, weight: Tensor, alphas: Sequence[float]) -> Tensor:
    """``LightGCN.get_embedding`` on a partition: valid rows = own users + items; differentiable."""
    return _PartitionedSum.apply(weight, pp, tuple(float(a) for a in alphas))


def own_pairs(pp: PartitionedPropagator, edge_label_index: Tensor) -> Tensor:
    """Mask of the label pairs this rank scores: a pair (u, i) belongs to the owner of its user
    (``edge_label_index[0]``, src/utils_v2.py:184-190 puts the users there); items are local everywhere."""
    u = edge_label_index[0]
    return (u >= pp.u0) & (u < pp.u1)


# The seeded scoring node (below) is used when the batch is far smaller than the table, like propagate.scores_from_table
SEEDED_STEP = os.environ.get("LGCN_PARTITION_SEEDED", "1") == "1"


def check_batch(w: Tensor, users: Tensor, pos: Tensor, neg: Tensor) -> None:
    """The kernels take the ids as raw int64 device pointers: anything else is refused here (upstream's own gathers --
    ``init_embed[batch_usr]``, src/utils_v2.py:205-207 -- are what raises for ids outside the table; here such a pair scores
    NaN and ``check_index_status()`` reports it)."""
    for name, t in (("users", users), ("pos", pos), ("neg", neg)):
        if not torch.is_tensor(t) or t.dtype != torch.int64 or t.dim() != 1 or t.device != w.device:
            raise TypeError(f"{name} must be a 1-D int64 tensor on {w.device}")
    if not (users.numel() == pos.numel() == neg.numel()) or users.numel() == 0:
        raise ValueError("users, pos and neg must hold the same, non-zero number of ids")


def step_forward(pp: "PartitionedPropagator", w: Tensor, alphas: tuple, users: Tensor, pos: Tensor, neg: Tensor, decay: float):
    """Forward half of one rank's training step on a global batch (see _PartitionedStep): returns
    (local loss, local bpr, regulariser of own users, regulariser of the batch's items, what the backward half needs)."""
    ops = pp.ops
    size = users.numel()
    mine = (users >= pp.u0) & (users < pp.u1)
    uc = users.clamp(pp.u0, pp.u1 - 1)                       # foreign users: any own row stands in, masked out below
    idx0, idx1 = torch.cat([uc, uc]), torch.cat([pos, neg])
    emb = pp.propagate_sum(w, alphas, final_rows=uc, final_item_rows=idx1)
    scores, e0, e1, _ = ops.pair_scores_rows(emb, idx0, idx1)
    mine_b = mine.to(torch.uint8)
    bpr_local, gs = ops.bpr_loss(scores, mine_b, size)
    park = torch.full_like(uc, -1)                           # rows other ranks own: "no row" for the seed
    own_u = torch.where(mine, uc, park)
    wr = w[torch.cat([uc, pos, neg])]                        # the 3B layer-0 rows of src/utils_v2.py:193-211
    sq = wr.pow(2).sum(1)
    reg_users = (sq[:size] * mine.to(sq.dtype)).sum() * (0.5 * decay / size)
    reg_items = sq[size:].sum() * (0.5 * decay / size)
    saved = (e0, e1, gs, torch.cat([mine_b, mine_b]), torch.cat([own_u, own_u, idx1]), torch.cat([own_u, idx1]), wr)
    return bpr_local + reg_users + reg_items, bpr_local, reg_users, reg_items, saved


def step_backward(pp: "PartitionedPropagator", saved, alphas: tuple, grad_local: Optional[Tensor], reg_scale: float,
                  zero_foreign: bool) -> Tensor:
    """Backward half: the gradient of the local loss with respect to the rows this rank owns (see _PartitionedStep).
    ``grad_local``: the upstream gradient of the local loss as a 0-dim device tensor, or None for 1."""
    e0, e1, gs, mine2, rows, reg_rows, wr = saved
    vals = pp.ops.pair_seed_vals(gs, mine2, grad_local, e0, e1)         # [4B, D]: user rows | item rows
    m2 = e0.size(0)
    pp.comm.reduce_now(vals[m2:])                                        # d / d out[item of pair m]: non-zero on the user's owner
    extra = []
    if reg_scale != 0.0:
        extra = [(reg_rows, wr if grad_local is None else wr * grad_local, reg_scale)]
    zero = [(0, pp.u0), (pp.u1, pp.n_users)] if zero_foreign else None
    return pp.seeded_transpose_sum(rows, vals, alphas, extra, zero)


class _PartitionedStep(torch.autograd.Function):
    """The local part of the loss of ONE global batch (src/train_lightgcn.py:137-144: bpr * size + reg) on a rank of a
    partition, as one autograd node -- the partitioned form of propagate._ScoresFromTable with the loss folded in:

    forward   propagate on the partition, the last user step for the batch's users only (lgc_spmm_rows); lgc_pair_dot_rows
              for all 2B pairs with FIXED shapes -- a triple whose user another rank owns is scored against an own user row
              standing in for it (the id clamped into [u0, u1)) and masked out of the loss, so that no step needs the host
              to know how many triples a rank owns (no sync, no dynamic shapes); lgc_bpr_loss (the BPR term over the own
              triples and its gradient with respect to the scores, one launch); the regulariser of the rows this rank
              accounts for: its own users of the batch, every item of the batch (item rows are replicated).
    backward  the seed of the transposed propagation (lgc_pair_seed_vals): user rows from this rank's own pairs; item rows
              from EVERY rank's pairs -- d score / d out[item] = g * out[user] is known to the user's owner only, so the
              [2B, D] table of those products (zero rows for foreign pairs) is summed over the ranks: 0.5 MB instead of the
              14 MB item block of a dense gradient; then propagate.seeded_sum on the rank's local halves of A^T with the
              per-hop item-block exchange, the regulariser's rows added by the same node.  Result: the gradient of the rows
              this rank owns; other ranks' user rows are left unwritten unless ``zero_foreign``.
    Outputs: (local loss, local bpr, regulariser of own users, regulariser of the batch's items) -- the last three for
    logging, not differentiable."""

    @staticmethod
    def forward(ctx, w: Tensor, pp: "PartitionedPropagator", alphas: tuple, users: Tensor, pos: Tensor, neg: Tensor,
                decay: float, zero_foreign: bool):
        local, bpr_local, reg_users, reg_items, saved = step_forward(pp, w.detach(), alphas, users, pos, neg, decay)
        ctx.save_for_backward(*saved)
        ctx.pp, ctx.alphas, ctx.reg_scale, ctx.zero_foreign = pp, alphas, float(decay) / float(users.numel()), zero_foreign
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(bpr_local, reg_users, reg_items)
        return local, bpr_local, reg_users, reg_items

    @staticmethod
    def backward(ctx, grad_local: Optional[Tensor], *unused):
        if grad_local is None:
            grad_local = torch.zeros((), dtype=torch.float32, device=ctx.saved_tensors[0].device)
        grad_local = grad_local.to(torch.float32).contiguous()
        grad = step_backward(ctx.pp, ctx.saved_tensors, ctx.alphas, grad_local, ctx.reg_scale, ctx.zero_foreign)
        return grad, None, None, None, None, None, None, None


def partitioned_bpr_loss(pp: PartitionedPropagator, weight: Tensor, alphas: Sequence[float], users: Tensor,
                         pos: Tensor, neg: Tensor, decay: float, pair_scores=None, zero_foreign_rows: bool = True):
    """The loss of src/train_lightgcn.py:137-144 (bpr * size + reg) for one GLOBAL batch given to every rank.

    Each rank scores the triples of its own users and returns the LOCAL part of the loss whose sum over
    ranks is the reference's loss; calling ``.backward()`` on it leaves in ``weight.grad`` the full gradient
    for the rows this rank owns (own users, all items -- item rows identical on every rank).  The
    regulariser's item terms are evaluated on every rank (replicated parameters), its user terms on the
    owner only.  Returns (local_loss, global_bpr, global_reg) -- the two globals detached, for logging.

    With a batch far smaller than the table (the training case) the step is one seeded node (_PartitionedStep):
    scored-rows-only last user step, seeded backward on the rank's halves of A^T, the regulariser's gradient inside the
    same node, no dense [N, D] gradient anywhere and no host sync.  ``zero_foreign_rows=False`` additionally leaves the
    user rows other ranks own unwritten in ``weight.grad`` (a caller that updates only ``pp.owned_row_ranges()``, like
    ``optim.Adam(row_ranges=...)``, never reads them).  Otherwise the dense path below.
    """
    check_batch(weight, users, pos, neg)
    size = users.numel()
    from . import propagate
    seeded = (SEEDED_STEP and propagate.SPARSE_BACKWARD and pair_scores is None and hasattr(pp.ops, "seed_pull")
              and pp.u1 > pp.u0 and torch.is_grad_enabled() and weight.requires_grad
              and len(alphas) - 1 <= _native.MAX_TERMS - 1
              and 2 * size * propagate.SEED_ROWS_FACTOR <= weight.size(0))
    if seeded:
        alphas = tuple(float(a) for a in alphas)
        local, bpr_local, reg_users, reg_items = _PartitionedStep.apply(weight, pp, alphas, users, pos, neg, float(decay),
                                                                        zero_foreign_rows)
    else:
        if pair_scores is None:
            from .propagate import pair_dot as pair_scores
        out = partitioned_embedding(pp, weight, alphas)
        mine = (users >= pp.u0) & (users < pp.u1)
        u, p, n = users[mine], pos[mine], neg[mine]
        # a rank without own triples must still take part in the collectives of the backward pass
        bpr_local = out[:1].sum() * 0.0
        if u.numel():
            scores = pair_scores(out, torch.stack((torch.cat([u, u]), torch.cat([p, n]))))
            m = u.numel()
            bpr_local = -torch.nn.functional.logsigmoid(scores[:m] - scores[m:]).sum() / size
        reg_users = 0.5 * weight[u].norm().pow(2) / size * decay
        reg_items = 0.5 * (weight[pos].norm().pow(2) + weight[neg].norm().pow(2)) / size * decay
        # item terms are computed identically on every rank: scale so that backward() leaves the reference's
        # gradient in the item rows exactly once, while the reported global value is still their plain sum
        local = bpr_local + reg_users + reg_items
    with torch.no_grad():
        part = torch.stack([bpr_local.detach(), reg_users.detach()])
        pp.comm.reduce_now(part)
        global_bpr, global_reg = part[0], part[1] + reg_items.detach()
    return local, global_bpr, global_reg
