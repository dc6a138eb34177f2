"""Test double for ``partition.HipOps``: the same protocol computed with the CPU oracle's arithmetic.

Lives under tests/ on purpose -- it lets the partition / exchange logic run on CPU ranks over gloo;
the product ships only the HIP implementation."""
from dataclasses import dataclass
from typing import Optional

import torch

from oracle import lightgcn_oracle as oracle


@dataclass
class CpuOp:
    rowptr: torch.Tensor
    cols: torch.Tensor
    vals: torch.Tensor
    row_begin: int
    row_end: int


@dataclass
class CpuGraph:
    forward_op: CpuOp
    edge_values: Optional[torch.Tensor]
    transpose_op: Optional[CpuOp] = None


class CpuOps:
    def build(self, edge_index, edge_weight, num_nodes, normalize, keep_edge_values=False):
        val = oracle.gcn_norm(edge_index, edge_weight, num_nodes) if normalize else (
            edge_weight if edge_weight is not None else torch.ones(edge_index.size(1)))
        order = torch.sort(edge_index[1], stable=True).indices
        counts = torch.bincount(edge_index[1], minlength=num_nodes)
        rowptr = torch.cat([torch.zeros(1, dtype=torch.long), counts.cumsum(0)])
        op = CpuOp(rowptr, edge_index[0][order], val[order], 0, num_nodes)
        order_t = torch.sort(edge_index[0], stable=True).indices           # rows = sources, same per-edge values
        counts_t = torch.bincount(edge_index[0], minlength=num_nodes)
        rowptr_t = torch.cat([torch.zeros(1, dtype=torch.long), counts_t.cumsum(0)])
        op_t = CpuOp(rowptr_t, edge_index[1][order_t], val[order_t], 0, num_nodes)
        return CpuGraph(op, val if keep_edge_values else None, op_t)

    def restrict(self, op, row_begin, row_end, short_max=None):
        return CpuOp(op.rowptr, op.cols, op.vals, row_begin, row_end)

    def apply(self, op, x, out, a, r, b):
        lo, hi = op.row_begin, op.row_end
        s, e = int(op.rowptr[lo]), int(op.rowptr[hi])
        rows = torch.repeat_interleave(torch.arange(lo, hi), op.rowptr[lo + 1:hi + 1] - op.rowptr[lo:hi]) - lo
        acc = torch.zeros(hi - lo, x.size(1)).index_add_(0, rows, op.vals[s:e].view(-1, 1) * x[op.cols[s:e]])
        res = a * acc
        if r is not None:
            res = res + b * r[lo:hi]
        out[lo:hi] = res
