"""The cached propagation graph: CSR of the normalised adjacency plus the launch plan.

The reference hands the model a dense ``[2, E]`` int64 COO on EVERY call and PyG re-derives
the normalisation in every layer.  Here the COO is converted once
per distinct ``(edge_index, edge_weight)`` tensor pair -- by ``lgc_build_csr`` on the device --
and the result is kept in a small cache keyed on tensor identity and version counter (H8).

Layout in HBM (per operator; A for the forward pass, A^T built lazily for the backward pass):
    rowptr   int32 [N+1]
    entries  {int32 col, fp32 val} [E]   8 B per edge, rows in edge order
    chunks   {row, begin, end, slot} int32 [C]   work list for rows longer than ``short_max``
    multi    {row, slot_begin, slot_end, 0} int32 [M]   rows cut into several chunks
    partials fp32 [n_slots, D]   scratch for those rows (allocated per width on first use)
"""
from __future__ import annotations

import os
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from . import _native

# Rows with at most SHORT_MAX entries are summed by one lane group in entry order; longer rows
# are cut into chunks of about CHUNK_LEN entries, one wavefront each.
SHORT_MAX = int(os.environ.get("LGCN_SHORT_MAX", "32"))
CHUNK_LEN = int(os.environ.get("LGCN_CHUNK_LEN", "256"))
# Width of the fixed slab of row heads the short-row kernel reads: 8 entries (0 = off).
SLAB_WIDTH = int(os.environ.get("LGCN_SLAB_WIDTH", "8"))


@dataclass
class RowPlan:
    """Work decomposition of rows [row_begin, row_end) of one CSR."""
    row_begin: int
    row_end: int
    short_max: int
    chunks: Tensor          # int32 [C, 4]
    multi: Tensor           # int32 [M, 4]
    n_slots: int

    @property
    def n_chunks(self) -> int:
        return self.chunks.size(0)

    @property
    def n_multi(self) -> int:
        return self.multi.size(0)


def build_row_plan(rowptr: Tensor, row_begin: int, row_end: int,
                   short_max: int = SHORT_MAX, chunk_len: int = CHUNK_LEN) -> RowPlan:
    """Pure index arithmetic (runs on whatever device ``rowptr`` lives on; unit-tested on CPU).

    Every row of [row_begin, row_end) with more than ``short_max`` entries is split into
    ``ceil(deg / chunk_len)`` near-equal chunks.  A row with one chunk is finished by that
    chunk (slot = -1); a row with several gets consecutive partial-sum slots, summed in slot
    order afterwards so the result does not depend on scheduling.
    """
    if short_max < 0 or chunk_len < 1:
        raise ValueError("short_max must be >= 0 and chunk_len >= 1")
    dev = rowptr.device
    i32 = dict(dtype=torch.int32, device=dev)
    rp = rowptr[row_begin:row_end + 1].to(torch.int64)
    deg = rp[1:] - rp[:-1]
    long_local = torch.nonzero(deg > short_max).flatten()
    if long_local.numel() == 0:
        return RowPlan(row_begin, row_end, short_max, torch.zeros((0, 4), **i32), torch.zeros((0, 4), **i32), 0)
    ldeg = deg[long_local]
    start = rp[long_local]
    nch = (ldeg + chunk_len - 1) // chunk_len
    per = (ldeg + nch - 1) // nch                       # near-equal split
    first = torch.cumsum(nch, 0) - nch
    owner = torch.repeat_interleave(torch.arange(long_local.numel(), device=dev), nch)
    j = torch.arange(owner.numel(), device=dev) - first[owner]
    begin = start[owner] + j * per[owner]
    end = torch.minimum(begin + per[owner], start[owner] + ldeg[owner])
    is_multi = nch > 1
    in_multi = is_multi[owner]
    slot = torch.where(in_multi, torch.cumsum(in_multi.to(torch.int64), 0) - 1, torch.full_like(owner, -1))
    chunks = torch.stack([long_local[owner] + row_begin, begin, end, slot], dim=1).to(torch.int32).contiguous()
    mnch = nch[is_multi]
    slot_begin = torch.cumsum(mnch, 0) - mnch
    multi = torch.stack([long_local[is_multi] + row_begin, slot_begin, slot_begin + mnch,
                         torch.zeros_like(mnch)], dim=1).to(torch.int32).contiguous()
    return RowPlan(row_begin, row_end, short_max, chunks, multi, int(mnch.sum().item()))


@dataclass
class Operator:
    """One sparse operator ready to be applied: CSR + plan (+ per-width scratch)."""
    n_rows: int
    rowptr: Tensor
    entries: Tensor                      # int32 [E, 2]: column, fp32 bits of the value
    plan: RowPlan
    slab: Optional[Tensor] = None        # int32 [n_rows * W, 2]: heads of every row, padded (lgc_build_slab)
    slab_width: int = 0
    _partials: Dict[int, Tensor] = field(default_factory=dict)

    @property
    def nnz(self) -> int:
        return self.entries.size(0)

    def partials(self, dim: int) -> Optional[Tensor]:
        n_slots = self.plan.n_slots
        if n_slots == 0:
            return None
        buf = self._partials.get(dim)
        if buf is None:
            buf = torch.empty((n_slots, dim), dtype=torch.float32, device=self.rowptr.device)
            self._partials[dim] = buf
        return buf

    def values(self) -> Tensor:
        return self.entries[:, 1].view(torch.float32)

    def columns(self) -> Tensor:
        return self.entries[:, 0]

    def apply(self, x: Tensor, out: Tensor, a: float = 1.0, r: Optional[Tensor] = None, b: float = 0.0) -> Tensor:
        """out[row] = a * (A x)[row] + b * r[row] for the rows of the plan.  Launches on the current stream."""
        _check_table(x, "x")
        _check_table(out, "out")
        dim = x.size(1)
        if out.size(1) != dim or (r is not None and r.size(1) != dim):
            raise ValueError("x, out and r must have the same width")
        if out.data_ptr() == x.data_ptr():
            raise ValueError("out must not alias x")
        if r is not None:
            _check_table(r, "r")
        lib = _native.load()
        if not lib.lgc_dim_ok(dim):
            raise _native.NativeLibraryError(f"embedding width {dim} is not supported by the HIP kernels")
        p = self.plan
        partials = self.partials(dim)
        table_rows = min(x.size(0), out.size(0))
        with torch.cuda.device(x.device):
            code = lib.lgc_spmm(
                _native.ptr(self.rowptr), _native.ptr(self.entries), p.row_begin, p.row_end, p.short_max,
                _native.ptr(p.chunks) if p.n_chunks else None, p.n_chunks,
                _native.ptr(p.multi) if p.n_multi else None, p.n_multi, _native.ptr(partials),
                _native.ptr(self.slab), self.slab_width, table_rows,
                _native.ptr(x), x.stride(0), _native.ptr(out), out.stride(0),
                _native.ptr(r), 0 if r is None else r.stride(0), float(a), float(b), dim,
                _native.stream_of(x.device))
        _native.check(code, "lgc_spmm")
        return out


def _check_table(t: Tensor, name: str) -> None:
    _native.require_device(t, name)
    if t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1:
        raise TypeError(f"{name} must be a 2-D fp32 tensor with unit inner stride, got {t.dtype} {tuple(t.shape)}")


class PropGraph:
    """Normalised adjacency of one ``(edge_index, edge_weight)`` pair, resident on the device."""

    def __init__(self, edge_index: Tensor, edge_weight: Optional[Tensor], num_nodes: int,
                 normalize: bool = True, short_max: int = SHORT_MAX, chunk_len: int = CHUNK_LEN,
                 keep_edge_values: bool = False):
        _native.require_device(edge_index, "edge_index")
        if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.size(0) != 2:
            raise TypeError("edge_index must be an int64 tensor of shape [2, E]")
        if edge_weight is not None:
            _native.require_device(edge_weight, "edge_weight")
            if edge_weight.dtype != torch.float32 or edge_weight.shape != (edge_index.size(1),):
                raise TypeError("edge_weight must be fp32 of shape [E]")
            if edge_weight.device != edge_index.device:
                raise RuntimeError("edge_index and edge_weight must be on the same device")
        self.device = edge_index.device
        self.num_nodes = int(num_nodes)
        self.num_edges = int(edge_index.size(1))
        self.normalize = bool(normalize)
        self.short_max, self.chunk_len = short_max, chunk_len
        # the COO is kept (by reference) for the lazy A^T build; the originals pin the cache key
        self._key_refs = (edge_index, edge_weight)
        self._edge_index = edge_index.contiguous()
        self._edge_weight = None if edge_weight is None else edge_weight.contiguous()
        self.status = torch.zeros(4, dtype=torch.int32, device=self.device)
        self.deg = torch.empty(self.num_nodes, dtype=torch.float32, device=self.device)
        self.dis = torch.empty(self.num_nodes, dtype=torch.float32, device=self.device)
        self.edge_values = (torch.empty(self.num_edges, dtype=torch.float32, device=self.device)
                            if keep_edge_values else None)
        self.forward_op = self._build(by_source=False)
        self._transpose_op: Optional[Operator] = None
        self.split = self._find_bipartite_split()
        self._halves = {}

    # -- construction ------------------------------------------------------------------
    def _build(self, by_source: bool, row_range: Optional[Tuple[int, int]] = None) -> Operator:
        lib = _native.load()
        n, e = self.num_nodes, self.num_edges
        ws_bytes = lib.lgc_build_workspace_bytes(n, e)
        if ws_bytes == 0:
            raise _native.NativeLibraryError("graph too large for int32 indexing")
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=self.device)
        rowptr = torch.empty(n + 1, dtype=torch.int32, device=self.device)
        entries = torch.empty((e, 2), dtype=torch.int32, device=self.device)
        first = not by_source
        with torch.cuda.device(self.device):
            code = lib.lgc_build_csr(
                _native.ptr(self._edge_index), _native.ptr(self._edge_weight), n, e,
                int(by_source), int(self.normalize),
                None if (first or not self.normalize) else _native.ptr(self.dis),
                _native.ptr(rowptr), _native.ptr(entries),
                _native.ptr(self.edge_values) if first else None,
                _native.ptr(self.deg) if first else None, _native.ptr(self.dis) if first else None,
                _native.ptr(ws), ws_bytes, _native.ptr(self.status), _native.stream_of(self.device))
        _native.check(code, "lgc_build_csr")
        # one host sync per build: index errors surface here, as IndexError like the reference's gather
        if int(self.status[0].item()) & _native.ST_INDEX_OOB:
            raise IndexError(f"edge_index contains node ids outside [0, {n})")
        lo, hi = row_range if row_range is not None else (0, n)
        plan = build_row_plan(rowptr, lo, hi, self.short_max, self.chunk_len)
        slab, width = None, 0
        if SLAB_WIDTH and e > 0:
            width = SLAB_WIDTH
            slab = torch.empty((n * width, 2), dtype=torch.int32, device=self.device)
            with torch.cuda.device(self.device):
                code = lib.lgc_build_slab(_native.ptr(rowptr), _native.ptr(entries), n, width, _native.ptr(slab),
                                          _native.stream_of(self.device))
            _native.check(code, "lgc_build_slab")
        return Operator(n, rowptr, entries, plan, slab, width)

    def _find_bipartite_split(self) -> Optional[int]:
        """s such that every edge joins a node < s ("users") with a node >= s ("items") -- the layout of
        src/utils_v2.py:146-165 -- or None.  Two reductions over the COO, one host sync per graph."""
        if self.num_edges == 0:
            return None
        lo = torch.minimum(self._edge_index[0], self._edge_index[1]).max()
        hi = torch.maximum(self._edge_index[0], self._edge_index[1]).min()
        lo, hi = int(lo.item()), int(hi.item())
        return lo + 1 if lo < hi else None

    def halves(self, transpose: bool = False) -> Tuple[Operator, Operator]:
        """(user-row operator, item-row operator): the same CSR with work plans limited to rows [0, split)
        and [split, N).  Only for bipartite graphs."""
        if self.split is None:
            raise ValueError("graph is not bipartite")
        got = self._halves.get(transpose)
        if got is None:
            op = self.transpose_op if transpose else self.forward_op
            got = tuple(Operator(op.n_rows, op.rowptr, op.entries,
                                 build_row_plan(op.rowptr, lo, hi, self.short_max, self.chunk_len),
                                 op.slab, op.slab_width)
                        for lo, hi in ((0, self.split), (self.split, self.num_nodes)))
            self._halves[transpose] = got
        return got

    @property
    def transpose_op(self) -> Operator:
        """A^T with the SAME per-edge values (exact adjoint, no symmetry assumption)."""
        if self._transpose_op is None:
            if self._edge_index is None:
                raise _native.NativeLibraryError("this graph was loaded from a file without its COO: the transposed "
                                                 "operator (backward pass) needs a graph built from edge_index")
            self._transpose_op = self._build(by_source=True)
        return self._transpose_op

    # -- persistence: a built graph as a flat safetensors file ------------------
    def save(self, path: str) -> None:
        """Write the forward operator (CSR, slab), degrees and metadata; nothing executable in the file.
        A serving worker can ``PropGraph.load`` it instead of re-reading the CSV, rebuilding the COO
        (torchserve/lightgcn_handler.py:32-38) and sorting it again."""
        from safetensors.torch import save_file
        op = self.forward_op
        tensors = {"rowptr": op.rowptr, "entries": op.entries, "deg": self.deg, "dis": self.dis}
        if op.slab is not None:
            tensors["slab"] = op.slab
        meta = {"format": "lgcn-graph-1", "num_nodes": str(self.num_nodes), "num_edges": str(self.num_edges),
                "normalize": str(int(self.normalize)), "slab_width": str(op.slab_width),
                "split": "" if self.split is None else str(self.split)}
        save_file({k: v.detach().cpu().contiguous() for k, v in tensors.items()}, path, metadata=meta)

    @classmethod
    def load(cls, path: str, device, short_max: int = SHORT_MAX, chunk_len: int = CHUNK_LEN) -> "PropGraph":
        """A graph usable for forward propagation (``get_embedding`` / ``recommendK``).  The COO is not stored,
        so the transposed operator (training) cannot be derived from a loaded graph."""
        from safetensors import safe_open
        device = torch.device(device)
        if device.type != "cuda":
            raise _native.NativeLibraryError("graphs are loaded onto a ROCm device only (no CPU fallback)")
        with safe_open(path, framework="pt", device="cpu") as f:
            meta = f.metadata() or {}
            if meta.get("format") != "lgcn-graph-1":
                raise ValueError(f"{path} is not a saved propagation graph")
            t = {k: f.get_tensor(k) for k in f.keys()}
        g = cls.__new__(cls)
        g.device, g.num_nodes, g.num_edges = device, int(meta["num_nodes"]), int(meta["num_edges"])
        g.normalize = bool(int(meta["normalize"]))
        g.short_max, g.chunk_len = short_max, chunk_len
        g._key_refs = g._edge_index = g._edge_weight = None
        g.status = torch.zeros(4, dtype=torch.int32, device=device)
        g.deg, g.dis, g.edge_values = t["deg"].to(device), t["dis"].to(device), None
        rowptr, entries = t["rowptr"].to(device), t["entries"].to(device)
        if rowptr.dtype != torch.int32 or rowptr.numel() != g.num_nodes + 1 or entries.shape != (g.num_edges, 2):
            raise ValueError(f"{path}: inconsistent tensor shapes")
        slab = t["slab"].to(device) if "slab" in t else None
        g.forward_op = Operator(g.num_nodes, rowptr, entries,
                                build_row_plan(rowptr, 0, g.num_nodes, short_max, chunk_len), slab,
                                int(meta.get("slab_width", "0")) if slab is not None else 0)
        g._transpose_op = None
        g.split = int(meta["split"]) if meta.get("split") else None
        g._halves = {}
        return g

    def nbytes(self) -> int:
        ops = [self.forward_op] + ([self._transpose_op] if self._transpose_op is not None else [])
        return sum(o.rowptr.numel() * 4 + o.entries.numel() * 4 + o.plan.chunks.numel() * 4 for o in ops)


# ----------------------------------------------------------------------------------------
# cache keyed on tensor identity + version (the callers pass the same tensors every step)
# ----------------------------------------------------------------------------------------
_CACHE: "OrderedDict[tuple, PropGraph]" = OrderedDict()
_CACHE_SIZE = int(os.environ.get("LGCN_GRAPH_CACHE", "4"))


def _key(edge_index: Tensor, edge_weight: Optional[Tensor], num_nodes: int, normalize: bool) -> tuple:
    w = None if edge_weight is None else (edge_weight.data_ptr(), edge_weight._version, tuple(edge_weight.shape))
    return (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape), tuple(edge_index.stride()),
            str(edge_index.device), w, int(num_nodes), bool(normalize))


def get_graph(edge_index: Tensor, edge_weight: Optional[Tensor], num_nodes: int, normalize: bool = True) -> PropGraph:
    key = _key(edge_index, edge_weight, num_nodes, normalize)
    g = _CACHE.get(key)
    if g is not None:
        _CACHE.move_to_end(key)
        return g
    g = PropGraph(edge_index, edge_weight, num_nodes, normalize)
    # the graph holds references to the COO tensors, so their addresses cannot be recycled
    # for different data while the entry is alive
    _CACHE[key] = g
    while len(_CACHE) > _CACHE_SIZE:
        _CACHE.popitem(last=False)
    return g


def clear_cache() -> None:
    _CACHE.clear()
