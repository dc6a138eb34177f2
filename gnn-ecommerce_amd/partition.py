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


ITEM_SHORT_MAX = int(os.environ.get("LGCN_ITEM_SHORT_MAX", "8"))


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


class HipOps:
    """The shipped implementation of the arithmetic: HIP kernels through the C ABI."""

    def build(self, edge_index: Tensor, edge_weight: Optional[Tensor], num_nodes: int, normalize: bool,
              keep_edge_values: bool = False) -> PropGraph:
        return PropGraph(edge_index, edge_weight, num_nodes, normalize, keep_edge_values=keep_edge_values)

    def restrict(self, op: Operator, row_begin: int, row_end: int, short_max: int = SHORT_MAX) -> Operator:
        """The same CSR, work plan limited to rows [row_begin, row_end)."""
        return Operator(op.n_rows, op.rowptr, op.entries,
                        build_row_plan(op.rowptr, row_begin, row_end, short_max, CHUNK_LEN), op.slab, op.slab_width)

    def apply(self, op: Operator, x: Tensor, out: Tensor, a: float, r: Optional[Tensor], b: float) -> None:
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
        # A rank's slice of an item row is short (mean degree / world entries) but there are only n_items
        # of them: one wavefront per row (the chunk kernel, 16 gathers in flight) beats one lane group per
        # row there, so only rows that fit the slab stay on the short-row kernel.
        self.item_op = self.ops.restrict(local.forward_op, n_users, n, ITEM_SHORT_MAX)
        self.local_nnz = int(mine.sum().item()) * 2
        self._keep = (full, local)

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
    def _item_step(self, x: Tensor, out: Tensor, a: float, r: Optional[Tensor], b: float):
        self.ops.apply(self.item_op, x, out, a, r if self.rank == 0 else None, b)
        if self.world > 1:
            return dist.all_reduce(out[self.n_users:], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return None

    @staticmethod
    def _finish_items(work) -> None:
        if work is not None:
            work.wait()

    def hop(self, x: Tensor, out: Tensor, a: float, r: Optional[Tensor], b: float) -> Tensor:
        """One complete hop (used on its own by tests and by callers that need a single LGConv)."""
        work = self._item_step(x, out, a, r, b)
        self.ops.apply(self.user_op, x, out, a, r, b)                        # overlaps the exchange
        self._finish_items(work)
        return out

    def propagate_sum(self, x0: Tensor, alphas: Sequence[float]) -> Tensor:
        """Horner form of sum_l alpha_l A^l x0 (see propagate.py); valid rows of the result: own users + items."""
        from . import propagate
        k = len(alphas) - 1
        if k == 0:
            return x0 * alphas[0]
        x0 = x0.contiguous()
        log = propagate.HOP_EVENT_LOG if x0.is_cuda else None
        # hop j (j = 0 .. k-1) maps h_in -> h_out with epilogue (a_j, b_j, r = x0)
        coef = [(alphas[k], alphas[k - 1])] + [(1.0, alphas[layer]) for layer in range(k - 2, -1, -1)]
        h_in, h_out = x0, torch.empty_like(x0)
        pending = None                                   # exchange of the previous hop's item block
        marks = []
        for j, (a, b) in enumerate(coef):
            if log is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                marks.append(ev)
            work = self._item_step(h_in, h_out, a, x0, b)   # needs only OWN user rows of h_in
            self._finish_items(pending)                  # now the previous hop's item block is needed
            self.ops.apply(self.user_op, h_in, h_out, a, x0, b)
            pending = work
            if j + 1 < k:
                h_in, h_out = h_out, torch.empty_like(x0)
        self._finish_items(pending)
        if log is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks.append(ev)
            log.extend(zip(marks[:-1], marks[1:]))
        return h_out

    def gather_users(self, table: Tensor) -> Tensor:
        """Fill every rank's user rows of ``table`` from their owners (all ranks end with the full table)."""
        if self.world == 1:
            return table
        for owner, (lo, hi) in enumerate(self.ranges):
            if hi > lo:
                dist.broadcast(table[lo:hi], src=dist.get_global_rank(self.group, owner) if self.group else owner,
                               group=self.group)
        return table
