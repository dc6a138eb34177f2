"""Differentiable propagation on the HIP kernels.

``propagate_sum`` computes what src/lightgcn.py:91-99 computes,

    out = sum_{l=0..K} alpha_l * A^l x0 ,

but in Horner form  h_K = alpha_K x0,  h_l = alpha_l x0 + A h_{l+1},  out = h_0 :
K sparse hops, each with the layer-sum folded into its epilogue (``y = a*(A x) + b*x0``).
Compared with "propagate, then axpy into a running sum" this removes the K read-modify-write
sweeps of the [N, D] accumulator -- per hop the only traffic beyond the SpMM
itself is one read of x0's row.  The value is the same polynomial in A; only the association of
the fp32 additions differs from the reference (covered by the norm-wise 1e-5 parity gate).

The operator is linear in x0, so the backward pass is the same routine on A^T and needs no
saved activations:  d out / d x0 = sum_l alpha_l (A^T)^l.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import torch
from torch import Tensor
from torch.autograd.function import once_differentiable

from . import _native
from .graph import Operator, PropGraph, apply_rows


# bench.py sets this to a list to receive one (start, end) pair of events per hop, recorded on the
# stream the kernels are launched on; None (the default) records nothing.
HOP_EVENT_LOG: Optional[list] = None


def _timed_apply(op: Operator, x: Tensor, out: Tensor, **kw) -> None:
    if HOP_EVENT_LOG is None:
        op.apply(x, out, **kw)
        return
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    op.apply(x, out, **kw)
    end.record()
    HOP_EVENT_LOG.append((start, end))


def scratch_table(like: Tensor) -> Tensor:
    """An internal [N, D] table.  Rows are padded to a multiple of 32 floats (128 B, one cache line) when D
    is not one already (D = 90 -> stride 96), so that every gathered row covers whole lines: a 360-byte
    row at stride 360 straddles 3.8 lines on average, at stride 384 exactly 3.  Only the API-facing
    tables (embedding.weight in, the result out) keep the caller's dense layout."""
    n, d = like.shape
    if PAD_INTERNAL and d % 32 != 0:
        stride = (d + 31) // 32 * 32
        return torch.empty((n, stride), dtype=like.dtype, device=like.device)[:, :d]
    return torch.empty_like(like)


PAD_INTERNAL = os.environ.get("LGCN_PAD_INTERNAL", "1") == "1"


def horner_hops(op: Operator, x0: Tensor, alphas: Sequence[float]) -> Tensor:
    """sum_l alphas[l] * op^l x0 with len(alphas)-1 launches of ``op.apply``."""
    k = len(alphas) - 1
    if k == 0:
        return x0 * alphas[0]
    x0 = x0.contiguous()
    h = torch.empty_like(x0) if k == 1 else scratch_table(x0)
    # first hop reads x0 directly:  h_{K-1} = alpha_K * (A x0) + alpha_{K-1} * x0
    _timed_apply(op, x0, h, a=alphas[k], r=x0, b=alphas[k - 1])
    for layer in range(k - 2, -1, -1):
        nxt = torch.empty_like(x0) if layer == 0 else scratch_table(x0)
        _timed_apply(op, h, nxt, a=1.0, r=x0, b=alphas[layer])
        h = nxt
    return h


USE_BIPARTITE = os.environ.get("LGCN_BIPARTITE", "1") == "1"

class _HopSpan:
    """Events around one layer's launches (both halves), appended to HOP_EVENT_LOG when bench.py asks."""

    def __enter__(self):
        self.log = HOP_EVENT_LOG
        if self.log is not None:
            self.start, self.end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.start.record()
        return self

    def __exit__(self, *exc):
        if self.log is not None:
            self.end.record()
            self.log.append((self.start, self.end))


def bipartite_sum(user_op: Operator, item_op: Operator, split: int, x0: Tensor, alphas: Sequence[float],
                  final_rows: Optional[Tensor] = None) -> Tensor:
    """sum_l alpha_l A^l x0 for A = [[0, R], [R^T, 0]] (users first), K = len(alphas) - 1 layers.

    With x_l = A^l x0:  x_l[items] = R^T x_{l-1}[users]  (item step, gathers user rows) and
    x_l[users] = R x_{l-1}[items]  (user step, gathers item rows).  Because the user side is linear in the
    item tables,
        out[users] = alpha_0 x0[users] + R ( sum_{l=1..K} alpha_l x_{l-1}[items] ),
    so the K-th user table is never materialised, only ONE user step (the last) reads an epilogue row,
    and the weighted sums over layers are taken on the small item tables only (``lgc_lincomb``, in the
    reference's own order: out = out + x * alpha).  Same K item steps + K user steps as K plain hops.

    ``final_rows`` (int64 node ids on the device, users and items alike): the caller will only ever read these rows of the
    result -- a training step scores 2B label pairs (src/lightgcn.py:123-125).  The LAST user step, whose 1.6 M output rows
    nothing else consumes, is then computed for the listed user rows only (``lgc_spmm_rows``: a few thousand gathers, the
    same bits for rows of up to 32 entries), and the LAST item step -- otherwise a full sweep over the user table -- for the
    listed item rows only (``lgc_spmm_rows_split``; SCORED_ITEM_ROWS_ONLY); every other row of the result is left
    uninitialised.
    """
    k = len(alphas) - 1
    if k == 0:
        return x0 * alphas[0]
    x0 = x0.contiguous()
    n = x0.size(0)
    tables = [x0]                                     # x_0 .. x_K (x_K: item rows only)
    # The reference's alpha is 1 / (K + 1) for every layer (src/lightgcn.py:75-79).  Then sum_{l<K} alpha_l x_l[items] IS
    # the table the last user step gathers, so the last item step adds it as its epilogue row and writes the item block
    # of the result directly: one lgc_lincomb and the K-th item table less, the same sums in the same order
    # (((a x_0 + a x_1) + a x_2) + a x_3).
    # (only when a result row is whole cache lines: the combine's stores into 360-byte rows cost more than the lincomb saves,
    # 4.67 vs 4.65 ms per K=5, D=90 step; 1.607 vs 1.623 ms at K=3, D=64)
    uniform = UNIFORM_ALPHA_SHORTCUT and all(a == alphas[0] for a in alphas) and (UNIFORM_ALPHA_SHORTCUT == "force" or x0.size(1) % 32 == 0)
    for layer in range(1, k + 1):
        with _HopSpan():
            nxt = scratch_table(x0)
            if layer < k:
                item_op.apply(tables[-1], nxt)                                  # x_l[items]
                user_op.apply(tables[-1], nxt)                                  # x_l[users]
                tables.append(nxt)
            else:
                out = torch.empty_like(x0)
                rows_only = final_rows is not None
                if rows_only and SCORED_ITEM_ROWS_ONLY and item_op.listed_rows_pay(final_rows.numel()):
                    # the scores read the item block at the batch's item rows only: the last item step -- a full sweep over
                    # the 420 MB user table otherwise -- is computed for the listed rows (lgc_spmm_rows_split: the rows cut
                    # into chunks on the device), on top of rest = sum_{l<K} alpha_l x_l[items]
                    mix = nxt
                    _native.lincomb(mix[split:], [(alphas[l], tables[l - 1][split:]) for l in range(1, k + 1)])
                    if all(a == alphas[0] for a in alphas):
                        rest = mix                                              # the same table when the alphas are equal
                    else:
                        rest = scratch_table(x0)
                        _native.lincomb(rest[split:], [(alphas[l], tables[l][split:]) for l in range(0, k)])
                    apply_rows(item_op, final_rows, tables[-1], out, a=alphas[k], r=rest, b=1.0, split=True)
                elif uniform:
                    mix = nxt                                                   # item rows: sum_l alpha_l x_{l-1}
                    _native.lincomb(mix[split:], [(alphas[l], tables[l - 1][split:]) for l in range(1, k + 1)])
                    item_op.apply(tables[-1], out, a=alphas[k], r=mix, b=1.0)   # out[items] = mix + alpha_K x_K[items]
                else:
                    item_op.apply(tables[-1], nxt)                              # x_K[items]
                    tables.append(nxt)
                    mix = scratch_table(x0)
                    _native.lincomb(mix[split:], [(alphas[l], tables[l - 1][split:]) for l in range(1, k + 1)])
                    _native.lincomb(out[split:], [(alphas[l], tables[l][split:]) for l in range(0, k + 1)])
                if not rows_only:
                    user_op.apply(mix, out, a=1.0, r=x0, b=alphas[0])            # out[users]
                else:
                    apply_rows(user_op, final_rows, mix, out, a=1.0, r=x0, b=alphas[0])
    return out


def _layer_sum(graph: PropGraph, x: Tensor, alphas: tuple, transpose: bool, final_rows: Optional[Tensor] = None) -> Tensor:
    if USE_BIPARTITE and graph.split is not None and len(alphas) - 1 <= _native.MAX_TERMS - 1:
        user_op, item_op = graph.halves(transpose)
        return bipartite_sum(user_op, item_op, graph.split, x, alphas, final_rows if SCORED_ROWS_ONLY else None)
    return horner_hops(graph.transpose_op if transpose else graph.forward_op, x, alphas)


# equal alphas: the last item step writes the result's item block itself (bipartite_sum); "0" keeps the two lincombs
UNIFORM_ALPHA_SHORTCUT = {"0": False, "force": "force"}.get(os.environ.get("LGCN_UNIFORM_ALPHA_SHORTCUT", "1"), True)
# the forward of a scoring step computes the last user step only for the rows its label pairs name (bipartite_sum)
SCORED_ROWS_ONLY = os.environ.get("LGCN_SCORED_ROWS_ONLY", "1") == "1"
# ... and the last ITEM step as well (lgc_spmm_rows_split); "0": the full item step, as get_embedding runs it
SCORED_ITEM_ROWS_ONLY = os.environ.get("LGCN_SCORED_ITEM_ROWS_ONLY", "1") == "1"


class _PropagateSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x0: Tensor, graph: PropGraph, alphas: tuple) -> Tensor:
        ctx.graph, ctx.alphas = graph, alphas
        return _layer_sum(graph, x0.detach(), alphas, transpose=False)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out: Tensor):
        return _layer_sum(ctx.graph, grad_out.contiguous(), ctx.alphas, transpose=True), None, None


class _Hop(torch.autograd.Function):
    """One LGConv hop y = A x (the operator surface of src/lightgcn.py:96)."""

    @staticmethod
    def forward(ctx, x: Tensor, graph: PropGraph) -> Tensor:
        ctx.graph = graph
        x = x.detach().contiguous()
        return graph.forward_op.apply(x, torch.empty_like(x))

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out: Tensor):
        g = grad_out.contiguous()
        return ctx.graph.transpose_op.apply(g, torch.empty_like(g)), None


def propagate_sum(x0: Tensor, graph: PropGraph, alphas: Sequence[float]) -> Tensor:
    _native.require_device(x0, "embedding table")
    return _PropagateSum.apply(x0, graph, tuple(float(a) for a in alphas))


def hop(x: Tensor, graph: PropGraph) -> Tensor:
    _native.require_device(x, "x")
    return _Hop.apply(x, graph)


# ----------------------------------------------------------------------------------------
# pair scoring
# ----------------------------------------------------------------------------------------
_status_words = {}


def _status(device: torch.device) -> Tensor:
    t = _status_words.get(device)
    if t is None:
        t = torch.zeros(4, dtype=torch.int32, device=device)
        _status_words[device] = t
    return t


_status_snapshots = {}


def _snapshot_status(device: torch.device) -> None:
    """Enqueue an asynchronous copy of the status word to pinned host memory (after a scoring launch); the next
    scoring call looks at it without waiting.  Not while a HIP graph is being captured (trainer.PartitionedTrainer): the
    status word itself is still written by the replayed launches and read by ``check_index_status()``."""
    if torch.cuda.is_current_stream_capturing():
        return
    snap = _status_snapshots.get(device)
    if snap is None:
        snap = (torch.zeros(4, dtype=torch.int32).pin_memory(), torch.cuda.Event())
        _status_snapshots[device] = snap
    snap[0].copy_(_status(device), non_blocking=True)
    snap[1].record(torch.cuda.current_stream(device))


def raise_pending_index_error(device: torch.device) -> None:
    """Non-blocking: if an EARLIER scoring launch on this device has finished and flagged an out-of-range label
    index, raise IndexError now (the reference's gather raises at the faulty call; here the kernels never fault and
    the error surfaces at the next call, or at ``check_index_status()``)."""
    snap = _status_snapshots.get(device)
    if snap is not None and snap[1].query() and int(snap[0][0]) & _native.ST_INDEX_OOB:
        snap[0].zero_()
        _status(device).zero_()
        raise IndexError("edge_label_index of an earlier call contained node ids outside [0, num_nodes)")


def check_index_status(device: Optional[torch.device] = None) -> None:
    """Synchronising check: raise IndexError if any pair-scoring launch since the last check saw an
    out-of-range label index (the kernels skip such pairs and score them NaN instead of faulting)."""
    for dev, t in list(_status_words.items()):
        if device is not None and dev != device:
            continue
        if int(t[0].item()) & _native.ST_INDEX_OOB:
            t.zero_()
            snap = _status_snapshots.get(dev)
            if snap is not None:
                snap[1].synchronize()
                snap[0].zero_()
            raise IndexError("edge_label_index contains node ids outside [0, num_nodes)")


class _PairDot(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb: Tensor, idx0: Tensor, idx1: Tensor) -> Tensor:
        lib = _native.load()
        emb_c = emb.detach().contiguous()
        idx0, idx1 = idx0.contiguous(), idx1.contiguous()
        scores = torch.empty(idx0.numel(), dtype=torch.float32, device=emb.device)
        with torch.cuda.device(emb.device):
            code = lib.lgc_pair_dot(_native.ptr(emb_c), emb_c.stride(0), emb_c.size(1), emb_c.size(0),
                                    _native.ptr(idx0), _native.ptr(idx1), idx0.numel(), _native.ptr(scores),
                                    _native.ptr(_status(emb.device)), _native.stream_of(emb.device))
        _native.check(code, "lgc_pair_dot")
        _snapshot_status(emb.device)
        ctx.save_for_backward(emb_c, idx0, idx1)
        return scores

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_scores: Tensor):
        emb, idx0, idx1 = ctx.saved_tensors
        n = emb.size(0)
        ok = (idx0 >= 0) & (idx0 < n) & (idx1 >= 0) & (idx1 < n)       # invalid pairs scored NaN and carry no gradient
        i0, i1 = idx0.clamp(0, n - 1), idx1.clamp(0, n - 1)
        gs = torch.where(ok, grad_scores, torch.zeros_like(grad_scores)).unsqueeze(1)
        rows = torch.cat([i0, i1])
        vals = torch.cat([gs * emb[i1], gs * emb[i0]])                   # d score / d emb[i0] = emb[i1] and the reverse
        grad = torch.zeros_like(emb)
        rows_s, perm = torch.sort(rows, stable=True)                     # repeated nodes add up in position order
        segment_sum(rows_s, rows_s, vals[perm].contiguous(), grad)
        return grad, None, None


def segment_sum(key_sorted: Tensor, dest: Tensor, vals: Tensor, out: Tensor, scale: float = 1.0, accumulate: bool = False,
                vals_index: Optional[Tensor] = None) -> None:
    """out[dest[t]] (+)= scale * (sum of vals over the run of equal keys that starts at t), for every run head t with
    dest[t] >= 0: lgc_segment_sum -- one lane group per destination row, fixed order, no atomics.  ``vals_index`` (int32):
    the value row of sorted position t is vals[vals_index[t]] (the permutation of the sort; vals stays unsorted)."""
    lib = _native.load()
    _native.require_device(out, "segment_sum: out")
    if out.dtype != torch.float32 or out.dim() != 2 or out.stride(1) != 1:
        raise TypeError("segment_sum: out must be a 2-D fp32 tensor with unit inner stride")
    m = key_sorted.numel()
    for name, t in (("key_sorted", key_sorted), ("dest", dest)):
        # raw pointers go to the kernel: a host tensor or int32 data read as int64 would be a GPU fault or garbage keys
        if t.dtype != torch.int64 or t.dim() != 1 or t.numel() != m or not t.is_contiguous() or t.device != out.device:
            raise TypeError(f"segment_sum: {name} must be a contiguous int64 vector of {m} entries on {out.device}, got "
                            f"{t.dtype} {tuple(t.shape)} on {t.device}")
    if (vals.dtype != torch.float32 or vals.dim() != 2 or vals.shape != (m, out.size(1)) or not vals.is_contiguous()
            or vals.device != out.device):
        raise TypeError(f"segment_sum: vals must be a contiguous fp32 [{m}, {out.size(1)}] tensor on {out.device}, got "
                        f"{vals.dtype} {tuple(vals.shape)} on {vals.device}")
    if vals_index is not None and (vals_index.dtype != torch.int32 or vals_index.shape != (m,) or not vals_index.is_contiguous()
                                   or vals_index.device != out.device):
        raise TypeError(f"segment_sum: vals_index must be a contiguous int32 vector of {m} entries on {out.device}")
    with torch.cuda.device(out.device):
        code = lib.lgc_segment_sum(_native.ptr(key_sorted), _native.ptr(dest), _native.ptr(vals), _native.ptr(vals_index),
                                   key_sorted.numel(), float(scale),
                                   _native.ptr(out), out.stride(0), out.size(0), out.size(1), int(accumulate),
                                   _native.stream_of(out.device))
    _native.check(code, "lgc_segment_sum")


# ----------------------------------------------------------------------------------------
# scores straight from the weight table: forward = propagate + pair scoring, backward seeded
# ----------------------------------------------------------------------------------------
SPARSE_BACKWARD = os.environ.get("LGCN_SPARSE_BACKWARD", "1") == "1"
# the seeded backward is used when the table has at least this many rows per possible seed row (2 per label pair)
SEED_ROWS_FACTOR = int(os.environ.get("LGCN_SEED_ROWS_FACTOR", "32"))


def _seed_pull(op: Operator, flag: Tensor, slot: Tensor, seed_vals: Tensor, out: Tensor, mark: Optional[Tensor] = None) -> None:
    """out[rows of op] = sum over the entries whose column carries a flag of val * seed_vals[slot[col]] (lgc_seed_pull):
    the rows of ``op`` through its row / chunk plan in lgc_spmm's fixed order; rows whose ``mark`` byte is 0 are written
    as zeros without being read."""
    lib = _native.load()
    p = op.plan
    with torch.cuda.device(out.device):
        code = lib.lgc_seed_pull(_native.ptr(op.rowptr), _native.ptr(op.entries), p.row_begin, p.row_end, p.short_max,
                                 _native.ptr(p.chunks) if p.n_chunks else None, p.n_chunks,
                                 _native.ptr(p.multi) if p.n_multi else None, p.n_multi, _native.ptr(op.partials(out.size(1))),
                                 _native.ptr(flag), _native.ptr(slot), _native.ptr(mark), _native.ptr(seed_vals),
                                 seed_vals.stride(0), out.size(0), _native.ptr(out), out.stride(0), out.size(1),
                                 _native.stream_of(out.device))
    _native.check(code, "lgc_seed_pull")


def _seed_mark(op: Operator, rows_sorted: Tensor, mark: Tensor, value: int) -> None:
    """mark[col] = value for every column of the rows of ``op`` listed in ``rows_sorted`` (lgc_seed_mark)."""
    lib = _native.load()
    p = op.plan
    with torch.cuda.device(mark.device):
        code = lib.lgc_seed_mark(_native.ptr(op.rowptr), _native.ptr(op.entries), p.row_begin, p.row_end, _native.ptr(rows_sorted),
                                 rows_sorted.numel(), _native.ptr(mark), mark.numel(), int(value), _native.stream_of(mark.device))
    _native.check(code, "lgc_seed_mark")


_seed_maps = {}


def _scratch_key(device: torch.device):
    """Scratch that must be all zero between uses is owned by one (device, stream, host thread) triple: the set / pull /
    clear sequence of a backward pass is ordered on ITS stream and issued by ITS thread only, so two backward passes on
    different streams or threads (two models, a DataParallel-style host loop, the thread-ranks of
    tests/test_partition_gpu.py) each get their own buffers instead of reading or clearing each other's flags."""
    import threading
    return (device, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0, threading.get_ident())


def _seed_map_buffers(device: torch.device, n_cols: int):
    """Per (device, stream): a byte flag and an int32 slot per column, flags all zero between uses (the pull reads a
    slot only where the flag is set; the flags of a step's seeds are cleared again right after the pull)."""
    key = _scratch_key(device)
    got = _seed_maps.get(key)
    if got is None or got[0].numel() < n_cols + 1:
        got = (torch.zeros(n_cols + 1, dtype=torch.uint8, device=device), torch.empty(n_cols + 1, dtype=torch.int32, device=device))
        _seed_maps[key] = got
    return got


_seed_marks = {}
# item rows without a seed user among their columns are written as zeros unread (lgc_seed_mark); "0" reads every row
SEED_MARKS = os.environ.get("LGCN_SEED_MARKS", "1") == "1"


def _seed_mark_buffer(device: torch.device, n_rows: int) -> Tensor:
    """Per (device, stream): one byte per table row, all zero between uses."""
    key = _scratch_key(device)
    got = _seed_marks.get(key)
    if got is None or got.numel() < n_rows:
        got = torch.zeros(n_rows, dtype=torch.uint8, device=device)
        _seed_marks[key] = got
    return got


class DeviceOps:
    """The launches the seeded backward and the partitioned path are written against: one method = one (or a few)
    launches through the C ABI on the current stream.  ``partition.HipOps`` is this plus the graph builders; the CPU ranks
    of tests/test_partition_gloo.py substitute a test double with the same methods (tests/cpu_ops.py) -- the product
    ships this implementation only."""

    def apply(self, op: Operator, x: Tensor, out: Tensor, a: float = 1.0, r: Optional[Tensor] = None, b: float = 0.0) -> None:
        op.apply(x, out, a=a, r=r, b=b)

    def apply_rows(self, op: Operator, rows: Tensor, x: Tensor, out: Tensor, a: float = 1.0, r: Optional[Tensor] = None,
                   b: float = 0.0, split: bool = False, compact: bool = False) -> None:
        apply_rows(op, rows, x, out, a=a, r=r, b=b, split=split, compact=compact)

    def lincomb(self, y: Tensor, terms) -> None:
        _native.lincomb(y, terms)

    def segment_sum(self, key_sorted: Tensor, dest: Tensor, vals: Tensor, out: Tensor, scale: float = 1.0,
                    accumulate: bool = False, vals_index: Optional[Tensor] = None) -> None:
        segment_sum(key_sorted, dest, vals, out, scale=scale, accumulate=accumulate, vals_index=vals_index)

    def seed_pull(self, op: Operator, flag: Tensor, slot: Tensor, seed_vals: Tensor, out: Tensor, mark: Optional[Tensor]) -> None:
        _seed_pull(op, flag, slot, seed_vals, out, mark)

    def seed_prepare(self, rows: Tensor, split: int, n: int, flag: Optional[Tensor] = None, slot: Optional[Tensor] = None):
        """Sort the seed's row ids and derive the destination lists of its segment sums (lgc_seed_prepare, one launch):
        (rows_sorted, perm int32, dest_item, dest_slot, dest_user) -- see include/lgconv_hip.h.  With ``flag`` / ``slot``:
        the column map of the seeded pull is set for the user heads as well.  Seeds above LGC_SEED_MAX ids take the same
        steps as torch ops."""
        m = rows.numel()
        dev = rows.device
        if m > _native.SEED_MAX:
            return seed_prepare_reference(rows, split, n, flag, slot)
        rows = rows.contiguous()
        i64 = dict(dtype=torch.int64, device=dev)
        buf = torch.empty((5, max(m, 1)), **i64)               # four outputs + the sorted keys in between the two launches
        perm = torch.empty(max(m, 1), dtype=torch.int32, device=dev)
        lib = _native.load()
        with torch.cuda.device(dev):
            code = lib.lgc_seed_prepare(_native.ptr(rows), m, int(split), int(n), _native.ptr(buf[0]), _native.ptr(perm),
                                        _native.ptr(buf[1]), _native.ptr(buf[2]), _native.ptr(buf[3]), _native.ptr(flag),
                                        _native.ptr(slot), _native.ptr(buf[4]), _native.stream_of(dev))
        _native.check(code, "lgc_seed_prepare")
        return buf[0, :m], perm[:m], buf[1, :m], buf[2, :m], buf[3, :m]

    def seed_flags(self, rows_sorted: Tensor, split: int, flag: Tensor, value: int) -> None:
        lib = _native.load()
        with torch.cuda.device(flag.device):
            code = lib.lgc_seed_flags(_native.ptr(rows_sorted), rows_sorted.numel(), int(split), _native.ptr(flag), int(value),
                                      _native.stream_of(flag.device))
        _native.check(code, "lgc_seed_flags")

    def pair_scores_rows(self, emb: Tensor, idx0: Tensor, idx1: Tensor):
        """(scores [M], rows0 [M, D], rows1 [M, D], ok uint8 [M]): lgc_pair_dot_rows -- the scores of src/lightgcn.py:123-125
        and, in the same launch, the gathered rows and validity bytes its backward needs."""
        lib = _native.load()
        m, dim, dev = idx0.numel(), emb.size(1), emb.device
        scores = torch.empty(m, dtype=torch.float32, device=dev)
        rows = torch.empty((2, m, dim), dtype=torch.float32, device=dev)
        ok = torch.empty(m, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            code = lib.lgc_pair_dot_rows(_native.ptr(emb), emb.stride(0), dim, emb.size(0), _native.ptr(idx0), _native.ptr(idx1), m,
                                         _native.ptr(scores), _native.ptr(rows[0]), _native.ptr(rows[1]), _native.ptr(ok),
                                         _native.ptr(_status(dev)), _native.stream_of(dev))
        _native.check(code, "lgc_pair_dot_rows")
        _snapshot_status(dev)
        return scores, rows[0], rows[1], ok

    def pair_seed_vals(self, grad_scores: Tensor, mask: Optional[Tensor], grad_scale: Optional[Tensor], rows0: Tensor,
                       rows1: Tensor) -> Tensor:
        """vals [2M, D]: vals[m] = g_m rows1[m], vals[M + m] = g_m rows0[m], g_m = mask[m] ? grad_scores[m] * grad_scale : 0
        (lgc_pair_seed_vals; ``grad_scale`` a 0-dim device tensor or None)."""
        lib = _native.load()
        m, dim, dev = rows0.size(0), rows0.size(1), rows0.device
        grad_scores = grad_scores.contiguous()
        vals = torch.empty((2 * m, dim), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            code = lib.lgc_pair_seed_vals(_native.ptr(grad_scores), _native.ptr(mask), _native.ptr(grad_scale), _native.ptr(rows0),
                                          _native.ptr(rows1), m, dim, _native.ptr(vals), _native.stream_of(dev))
        _native.check(code, "lgc_pair_seed_vals")
        return vals

    def bpr_loss(self, scores: Tensor, mask: Optional[Tensor], size: int):
        """(loss 0-dim, grad [2B]): lgc_bpr_loss on scores = [pos | neg]."""
        lib = _native.load()
        dev = scores.device
        b = scores.numel() // 2
        out = torch.empty(1 + 2 * b, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            code = lib.lgc_bpr_loss(_native.ptr(scores), _native.ptr(mask), b, int(size), _native.ptr(out), _native.ptr(out[1:]),
                                    _native.stream_of(dev))
        _native.check(code, "lgc_bpr_loss")
        return out[0], out[1:]

    def reg_rows(self, w: Tensor, lists: Sequence[Tensor], scale: float):
        """(value 0-dim, rows int64 [sum of lengths]): lgc_reg_rows -- scale * sum over the three id lists of |w[ids]|_F^2 and
        the ids as row numbers (negative ids wrapped, out-of-range ones -1 and reported through the index status)."""
        lib = _native.load()
        dev = w.device
        a, b, c = lists
        value = torch.empty(1, dtype=torch.float32, device=dev)
        rows = torch.empty(a.numel() + b.numel() + c.numel(), dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            code = lib.lgc_reg_rows(_native.ptr(w), w.stride(0), w.size(1), w.size(0), _native.ptr(a), a.numel(), _native.ptr(b),
                                    b.numel(), _native.ptr(c), c.numel(), float(scale), _native.ptr(value), _native.ptr(rows),
                                    _native.ptr(_status(dev)), _native.stream_of(dev))
        _native.check(code, "lgc_reg_rows")
        return value[0], rows

    def seed_mark(self, op: Operator, rows_sorted: Tensor, mark: Tensor, value: int) -> None:
        _seed_mark(op, rows_sorted, mark, value)

    def pair_scores(self, emb: Tensor, idx0: Tensor, idx1: Tensor) -> Tensor:
        """scores[m] = <emb[idx0[m]], emb[idx1[m]]> (lgc_pair_dot; out-of-range pairs score NaN and raise later)."""
        lib = _native.load()
        scores = torch.empty(idx0.numel(), dtype=torch.float32, device=emb.device)
        with torch.cuda.device(emb.device):
            code = lib.lgc_pair_dot(_native.ptr(emb), emb.stride(0), emb.size(1), emb.size(0), _native.ptr(idx0),
                                    _native.ptr(idx1), idx0.numel(), _native.ptr(scores), _native.ptr(_status(emb.device)),
                                    _native.stream_of(emb.device))
        _native.check(code, "lgc_pair_dot")
        _snapshot_status(emb.device)
        return scores

    def scratch_table(self, like: Tensor) -> Tensor:
        return scratch_table(like)

    def adam_rows(self, w: Tensor, g: Tensor, m: Tensor, v: Tensor, lo: int, hi: int, hyper: Tensor) -> None:
        """One Adam step over the flat element range [lo, hi) of four same-shaped contiguous fp32 tables, the six scalars
        read from the device tensor ``hyper`` (lgc_adam_step_hp: capturable, see trainer.PartitionedTrainer)."""
        for t in (g, m, v):
            if t.shape != w.shape or t.dtype != torch.float32 or not t.is_contiguous() or t.device != w.device:
                raise TypeError("adam_rows: parameter, gradient and moments must be contiguous fp32 tensors of one shape")
        if not 0 <= lo <= hi <= w.numel() or hyper.numel() != 6 or hyper.dtype != torch.float32 or hyper.device != w.device:
            raise ValueError("adam_rows: bad range or hyper-parameter tensor")
        lib = _native.load()
        with torch.cuda.device(w.device):
            code = lib.lgc_adam_step_hp(_native.ptr(w) + 4 * lo, _native.ptr(g) + 4 * lo, _native.ptr(m) + 4 * lo,
                                        _native.ptr(v) + 4 * lo, hi - lo, _native.ptr(hyper), _native.stream_of(w.device))
        _native.check(code, "lgc_adam_step_hp")


def seed_prepare_reference(rows: Tensor, split: int, n: int, flag: Optional[Tensor] = None, slot: Optional[Tensor] = None):
    """What lgc_seed_prepare computes, as torch ops on any device (seeds above LGC_SEED_MAX ids; the CPU test double)."""
    dev = rows.device
    m = rows.numel()
    rows = torch.where((rows >= 0) & (rows < n), rows, torch.full_like(rows, -1))
    rows_s, perm = torch.sort(rows, stable=True)
    pos = torch.arange(m, device=dev)
    head = torch.ones(m, dtype=torch.bool, device=dev)
    head[1:] = rows_s[1:] != rows_s[:-1]
    user = (rows_s >= 0) & (rows_s < split)
    none = torch.full_like(rows_s, -1)
    dest_item = torch.where(head & (rows_s >= split), rows_s, none)
    dest_slot = torch.where(head & user, pos, none)
    dest_user = torch.where(head & user, rows_s, none)
    if flag is not None:
        heads = rows_s[head & user]
        flag[heads] = 1
        slot[heads] = pos[head & user].to(torch.int32)
    return rows_s, perm.to(torch.int32), dest_item, dest_slot, dest_user


DEVICE_OPS = DeviceOps()


def seeded_transpose_sum(graph: PropGraph, rows: Tensor, vals: Tensor, alphas: Sequence[float], n: int,
                         extra: Optional[Sequence] = None) -> Tensor:
    """sum_l alpha_l (A^T)^l g for a gradient g given by its non-zero rows, on one device: ``seeded_sum`` with the
    graph's own operator halves and no exchange."""
    user_t, item_t = graph.halves(True)
    return seeded_sum(user_t, item_t, graph.halves(False)[0], graph.split, rows, vals, alphas, n, extra)


def seeded_sum(user_t, item_t, user_fwd, split: int, rows: Tensor, vals: Tensor, alphas: Sequence[float], n: int,
               extra: Optional[Sequence] = None, ops: Optional[DeviceOps] = None, exchange=None,
               zero_rows: Optional[Sequence] = None) -> Tensor:
    """sum_l alpha_l (A^T)^l g for a gradient g given by its non-zero rows (``rows`` int64, repeats add up; ``vals`` [len, D])
    on a user|item graph -- what ``_PropagateSum.backward`` computes from a dense g, without ever forming it:
      * the seed is sorted once; repeated rows are summed in position order by lgc_segment_sum (no float atomics anywhere:
        the gradient has the same bits on every run, like the reference's CPU path);
      * hop 1, item side: only edges between an item and a seed USER matter -> lgc_seed_pull over the item rows of A^T
        (an entry counts if its user carries a flag; it then reads that user's row of the compact seed table) instead of
        a dense item step gathering 10 M rows of zeros, and only over the item rows some seed user's own row names
        (lgc_seed_mark: ~6 per seed user of 54 k) -- the others are zeros without a look at their entries;
      * hop 1, user side: the dense user step, reading only the 14 MB item block of g (the rest of that table is never
        initialised, let alone zero-filled);
      * the alpha_0 g term of the user rows is added to the seed rows afterwards instead of being read as a dense
        epilogue table by the last user step.
    ``user_t`` / ``item_t``: the user-row and item-row halves of A^T; ``user_fwd``: the user-row half of A (its columns
    name the item rows of A^T that hold a seed user).  Negative row ids stand for "no row" (a rank of a partition parks
    the pairs other ranks own there: a run of equal ids is summed by ONE lane group, so a thousand pairs clamped onto one
    stand-in row cost 330 us).  ``extra``: (rows, vals, scale) triples added to the result's rows
    the same way -- the regulariser's gradient (LightGCN.regularization_loss), which upstream's autograd materialises as
    three dense [N, D] tables.
    On a rank of a partition (partition.py) the halves are the rank's LOCAL operators -- the user rows it owns, and the
    item rows restricted to its own users as columns -- and ``exchange(block)`` starts the sum of an item block over the
    ranks, returning a handle whose ``wait()`` completes it (None: nothing to wait for).  The item block of layer l is
    waited for right before the user step of layer l + 1 reads it, exactly like the forward pass.  ``zero_rows``: row
    ranges of the result to zero-fill (the user rows other ranks own); default none."""
    ops = ops or DEVICE_OPS
    k = len(alphas) - 1
    dim = vals.size(1)
    dev = vals.device
    m = rows.numel()
    vals = vals.contiguous()
    # one launch (lgc_seed_prepare): the sorted ids, the permutation, the three destination lists and -- for k > 0 -- the
    # column map of the seeded pull (flag = 1, slot = position of the run's head, for every seed user)
    flag, slot = _seed_map_buffers(dev, split) if k > 0 else (None, None)
    rows_s, perm, dest_item, dest_slot, dest_user = ops.seed_prepare(rows, split, n, flag, slot)
    g_tab = torch.empty((n, dim), dtype=torch.float32, device=dev)         # only the item block is ever read
    g_tab[split:].zero_()
    ops.segment_sum(rows_s, dest_item, vals, g_tab, vals_index=perm)                        # g[items]

    def wait(work) -> None:
        if work is not None:
            work.wait()

    def finish(out: Tensor) -> Tensor:
        for e_rows, e_vals, e_scale in (extra or ()):
            e_s, e_perm, e_item, _, e_user = ops.seed_prepare(e_rows, split, n)
            ops.segment_sum(e_s, torch.maximum(e_item, e_user), e_vals.contiguous(), out, scale=e_scale, accumulate=True,
                            vals_index=e_perm)
        return out

    def new_result() -> Tensor:
        out = torch.empty((n, dim), dtype=torch.float32, device=dev)
        for lo, hi in (zero_rows or ()):
            if hi > lo:
                out[lo:hi].zero_()
        return out

    if k == 0:
        out = torch.zeros((n, dim), dtype=torch.float32, device=dev)
        ops.segment_sum(rows_s, torch.maximum(dest_item, dest_user), vals, out, scale=alphas[0], vals_index=perm)
        return finish(out)
    # compact table of the seed users: slot = position of the run's head
    gu = torch.empty((m, dim), dtype=torch.float32, device=dev)
    ops.segment_sum(rows_s, dest_slot, vals, gu, vals_index=perm)
    # Which item rows of A^T hold a seed user among their columns: row i of A^T has the columns {u : A[u, i] != 0}, so
    # the rows to read are the columns of the seed users' rows of the FORWARD operator A (not of A^T, whose user row u
    # lists {i : A[i, u] != 0} -- the same set only on a structurally symmetric edge list, which is never assumed).
    tables = [g_tab]
    pending = [None] * (k + 1)                                             # exchange of x_l[items]
    out = None
    for layer in range(1, k + 1):
        with _HopSpan():
            nxt = ops.scratch_table(g_tab)
            if layer == 1:
                mark = _seed_mark_buffer(dev, n) if SEED_MARKS else None
                try:
                    if mark is not None:                                        # the item rows next to a seed user: the only
                        ops.seed_mark(user_fwd, rows_s, mark, 1)                # rows the pull has to read
                    ops.seed_pull(item_t, flag, slot, gu, nxt, mark)            # x_1[items] from the seed users
                finally:
                    ops.seed_flags(rows_s, split, flag, 0)                      # flags and marks are all zero between steps
                    if mark is not None:
                        ops.seed_mark(user_fwd, rows_s, mark, 0)
            else:
                ops.apply(item_t, tables[-1], nxt)
            if exchange is not None:
                pending[layer] = exchange(nxt[split:])
            wait(pending[layer - 1])                                            # x_{l-1}[items] is read from here on
            if layer < k:
                ops.apply(user_t, tables[-1], nxt)
                tables.append(nxt)
            else:
                tables.append(nxt)
                out = new_result()
                mix = ops.scratch_table(g_tab)
                ops.lincomb(mix[split:], [(alphas[l], tables[l - 1][split:]) for l in range(1, k + 1)])
                ops.apply(user_t, mix, out, a=1.0)
                wait(pending[k])
                if all(a == alphas[0] for a in alphas):     # equal alphas: mix + a x_K is the same chain of adds
                    ops.lincomb(out[split:], [(1.0, mix[split:]), (alphas[k], tables[k][split:])])
                else:
                    ops.lincomb(out[split:], [(alphas[l], tables[l][split:]) for l in range(0, k + 1)])
                # + alpha_0 g on the seed users (each such row is owned by one lane group: read, add, write)
                ops.segment_sum(rows_s, dest_user, vals, out, scale=alphas[0], accumulate=True, vals_index=perm)
    return finish(out)


class RegHook:
    """Where a regulariser on the layer-0 rows of the SAME weight table hands its gradient to the scoring node of one
    forward (``LightGCN.regularization_loss``): ``terms`` collects (rows, scale) pairs; the scoring node's backward adds
    ``grad_of_the_hook * scale * w[rows]`` to the <= 3B rows of the dense gradient it already writes."""

    def __init__(self, weight: Tensor):
        self.weight, self.token, self.terms, self.spent = weight, None, [], False


class _ScoresFromTable(torch.autograd.Function):
    """scores[m] = <out[i0_m], out[i1_m]> with out = sum_l alpha_l A^l w (src/lightgcn.py:121-125) as ONE autograd node:
    keeps the 2M gathered rows instead of the [N, D] table, and runs the backward pass from the sparse seed.  Its second
    output is a zero-valued scalar ``token``: a loss term that depends on ``w`` only through a few of its rows can route
    its gradient through it (RegHook) instead of through a dense [N, D] tensor of its own."""

    @staticmethod
    def forward(ctx, w: Tensor, graph: PropGraph, alphas: tuple, idx0: Tensor, idx1: Tensor, hook: Optional[RegHook]):
        idx0, idx1 = idx0.contiguous(), idx1.contiguous()
        n_nodes = w.size(0)
        # only the rows the pairs name are read below (clamped like the gathers that follow, so that every row that is
        # gathered has been computed): the last user step is restricted to them
        # (ids outside the table are skipped by the listed-rows launches and scored NaN below: no clamp; the two rows of one
        # contiguous [2, M] label tensor are one vector already: no cat)
        m = idx0.numel()
        if (idx1.data_ptr() == idx0.data_ptr() + 8 * m and idx0.untyped_storage().data_ptr() == idx1.untyped_storage().data_ptr()):
            rows = idx0.as_strided((2 * m,), (1,))
        else:
            rows = torch.cat([idx0, idx1])
        emb = _layer_sum(graph, w.detach(), alphas, transpose=False, final_rows=rows)
        # one launch: the scores, the two gathered rows of every pair and a validity byte (an invalid pair scores NaN,
        # keeps zero rows and carries no gradient)
        scores, e0, e1, ok = DEVICE_OPS.pair_scores_rows(emb, idx0, idx1)
        ctx.save_for_backward(e0, e1, rows, ok)
        ctx.graph, ctx.alphas, ctx.n, ctx.hook, ctx.width = graph, alphas, emb.size(0), hook, emb.size(1)
        ctx.set_materialize_grads(False)                                # an unused output arrives as None, not as zeros
        return scores, torch.zeros((), dtype=torch.float32, device=emb.device)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_scores: Tensor, grad_token: Optional[Tensor]):
        e0, e1, rows, ok = ctx.saved_tensors
        hook = ctx.hook
        extra = []
        if hook is not None:
            hook.spent = True                                           # later regularisers take the plain torch path
            if grad_token is not None:
                w = hook.weight.detach()
                extra = [(r, w[r] * grad_token, scale) for r, scale in hook.terms]
        if grad_scores is None:                                          # only the regulariser was differentiated
            out = torch.zeros((ctx.n, ctx.width), dtype=torch.float32, device=e0.device)
            for r, v, scale in extra:
                r_s, perm = torch.sort(r, stable=True)
                segment_sum(r_s, r_s, v.contiguous(), out, scale=scale, accumulate=True, vals_index=perm.to(torch.int32))
            return out, None, None, None, None, None
        # pairs that share a node simply add up; d score / d out[i0] = e1, d / d out[i1] = e0; an invalid pair's id is
        # outside the table: "no row" for the seed
        vals = DEVICE_OPS.pair_seed_vals(grad_scores, ok, None, e0, e1)
        return seeded_transpose_sum(ctx.graph, rows, vals, ctx.alphas, ctx.n, extra), None, None, None, None, None


class _RegThroughHook(torch.autograd.Function):
    """value = decay/2 * (|w[u]|^2 + |w[p]|^2 + |w[n]|^2) / size (src/utils_v2.py:193-211), differentiable through the
    scoring node's token: d value / d w[r] = decay / size * w[r] per occurrence of r, added by that node's backward."""

    @staticmethod
    def forward(ctx, token: Tensor, value: Tensor) -> Tensor:
        return value + token

    @staticmethod
    def backward(ctx, grad: Tensor):
        return grad, None


def regularization_through(hook: RegHook, size: int, users: Tensor, pos: Tensor, neg: Tensor, decay: float) -> Tensor:
    """The regulariser of src/utils_v2.py:193-211 on ``hook.weight``, its gradient routed into the scoring node.  Value and
    row list in ONE launch (lgc_reg_rows) when the table qualifies (fp32, unit inner stride): upstream's expression is 13
    small launches plus nine that normalise the ids -- 0.13 ms of a 3.4 ms step."""
    w = hook.weight.detach()
    # upstream's gathers take CPU or int32 index tensors and negative (wrapping) ids on a CUDA table; the kernels take raw
    # int64 device pointers, so the lists are normalised here (no launch when they already are int64 on the device)
    lists = [t.reshape(-1).to(device=w.device, dtype=torch.int64).contiguous() for t in (users, pos, neg)]
    if FUSED_GLUE and w.dtype == torch.float32 and w.dim() == 2 and w.stride(1) == 1:
        value, rows = DEVICE_OPS.reg_rows(w, lists, 0.5 * float(decay) / float(size))
    else:
        value = (1 / 2) * (w[users].norm().pow(2) + w[pos].norm().pow(2) + w[neg].norm().pow(2)) / size * decay
        rows = torch.cat([torch.where(r < 0, r + w.size(0), r) for r in lists])
    hook.terms.append((rows, float(decay) / float(size)))
    return _RegThroughHook.apply(hook.token, value)


# the step's glue around the scoring node as single launches (lgc_reg_rows, lgc_bpr_loss); "0": upstream's torch expressions
FUSED_GLUE = os.environ.get("LGCN_FUSED_GLUE", "1") == "1"


class _BprLoss(torch.autograd.Function):
    """-mean(log sigmoid(pos - neg)) / n_pairs (src/lightgcn.py:262-286 with lambda_reg = 0) and its gradient in one launch
    (lgc_bpr_loss) instead of five forward and six backward launches of elementwise torch kernels."""

    @staticmethod
    def forward(ctx, positives: Tensor, negatives: Tensor) -> Tensor:
        n = positives.numel()
        pos, neg = positives.reshape(-1), negatives.reshape(-1)
        if (pos.is_contiguous() and neg.is_contiguous() and neg.data_ptr() == pos.data_ptr() + 4 * n
                and pos.untyped_storage().data_ptr() == neg.untyped_storage().data_ptr()):
            scores = pos.as_strided((2 * n,), (1,))            # out[:B], out[B:] of one score vector: no copy
        else:
            scores = torch.cat([pos, neg])
        loss, grad = DEVICE_OPS.bpr_loss(scores, None, n * n)   # -sum log sigmoid / n^2
        ctx.save_for_backward(grad)
        ctx.shapes = (positives.shape, negatives.shape)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out: Tensor):
        (grad,) = ctx.saved_tensors
        g = grad * grad_out
        n = g.numel() // 2
        return g[:n].reshape(ctx.shapes[0]), g[n:].reshape(ctx.shapes[1])


def bpr_loss_fused(positives: Tensor, negatives: Tensor) -> Optional[Tensor]:
    """``BPRLoss(0)(positives, negatives)`` through lgc_bpr_loss, or None when the inputs do not qualify (then the caller
    evaluates upstream's expression on torch ops)."""
    if (FUSED_GLUE and positives.is_cuda and negatives.is_cuda and positives.dtype == torch.float32
            and negatives.dtype == torch.float32 and positives.shape == negatives.shape and positives.numel() > 0
            and positives.device == negatives.device and positives.numel() <= 1 << 20):
        return _BprLoss.apply(positives, negatives)
    return None


def routable_index(t) -> bool:
    """An index argument the routed regulariser can take: an integer (not bool) tensor -- anything else upstream's
    ``init_embed[idx]`` accepts (masks, lists, slices) goes through upstream's expression on plain torch ops."""
    return torch.is_tensor(t) and t.dtype in (torch.int64, torch.int32, torch.int16, torch.int8, torch.uint8)


def regularizer_rows(t: Tensor, w: Tensor) -> Tensor:
    """Row ids of ``w[t]`` as a flat int64 vector on ``w``'s device, negative ids wrapped like torch's indexing."""
    r = t.reshape(-1).to(device=w.device, dtype=torch.int64)
    return torch.where(r < 0, r + w.size(0), r)


def scores_from_table(w: Tensor, graph: PropGraph, alphas: Sequence[float], edge_label_index: Tensor,
                      hook: Optional[RegHook] = None) -> Tensor:
    """``pair_dot(propagate_sum(w, graph, alphas), edge_label_index)``; with gradients on, a user|item graph and few
    label pairs it runs as one node whose backward pass starts from the sparse seed.  ``hook``: filled
    in (``hook.token``) when that node is used, so that a regulariser can hand it its gradient."""
    _native.require_device(w, "embedding table")
    _native.require_device(edge_label_index, "edge_label_index")
    if edge_label_index.dtype != torch.int64 or edge_label_index.dim() != 2 or edge_label_index.size(0) != 2:
        raise TypeError("edge_label_index must be an int64 tensor of shape [2, M]")
    raise_pending_index_error(w.device)
    alphas = tuple(float(a) for a in alphas)
    sparse_ok = (SPARSE_BACKWARD and torch.is_grad_enabled() and w.requires_grad and graph.split is not None
                 and USE_BIPARTITE and len(alphas) - 1 <= _native.MAX_TERMS - 1
                 and 2 * edge_label_index.size(1) * SEED_ROWS_FACTOR <= w.size(0))   # a seed far smaller than the table
    if not sparse_ok:
        few = (SCORED_ROWS_ONLY and graph.split is not None and USE_BIPARTITE and len(alphas) - 1 <= _native.MAX_TERMS - 1
               and len(alphas) > 1 and 2 * edge_label_index.size(1) * SEED_ROWS_FACTOR <= w.size(0))
        if few and not (torch.is_grad_enabled() and w.requires_grad):
            # scoring a few pairs without gradients (an evaluation batch under no_grad): the last user step and the last
            # item step for the rows the pairs name only, like the training forward
            rows = edge_label_index.reshape(-1).clamp(0, w.size(0) - 1)
            return pair_dot(_layer_sum(graph, w.detach().contiguous(), alphas, transpose=False, final_rows=rows), edge_label_index)
        return pair_dot(propagate_sum(w, graph, alphas), edge_label_index)
    scores, token = _ScoresFromTable.apply(w, graph, alphas, edge_label_index[0], edge_label_index[1], hook)
    if hook is not None:
        hook.token = token
    return scores


def pair_dot(emb: Tensor, edge_label_index: Tensor) -> Tensor:
    """scores[m] = <emb[idx[0, m]], emb[idx[1, m]]>  (src/lightgcn.py:123-125)."""
    _native.require_device(emb, "embeddings")
    _native.require_device(edge_label_index, "edge_label_index")
    if edge_label_index.dtype != torch.int64 or edge_label_index.dim() != 2 or edge_label_index.size(0) != 2:
        raise TypeError("edge_label_index must be an int64 tensor of shape [2, M]")
    if emb.dtype != torch.float32 or emb.dim() != 2:
        raise TypeError("embeddings must be a 2-D fp32 tensor")
    return _PairDot.apply(emb, edge_label_index[0], edge_label_index[1])


# ----------------------------------------------------------------------------------------
# serving tail: seen-mask + top-k on the device
# ----------------------------------------------------------------------------------------
TOPK_MAX = 256


class SeenLists:
    """The purchase matrix as a device CSR (``ptr`` int64 [n_users + 1], ``items`` int64) plus the users of a
    request: the list form of ``recommendK``'s ``interactions_t`` (seen = 1 for listed items)."""

    def __init__(self, ptr: Tensor, items: Tensor, users: Optional[Tensor] = None):
        self.ptr, self.items, self.users = ptr, items, users

    def validate(self, n_users: int, where: str = "purchase lists") -> "SeenLists":
        """lgc_mask_topk reads ptr[u], ptr[u + 1] and items[ptr[u] .. ptr[u + 1]) on trust: a truncated or stale list
        (ptr shorter than n_users + 1, not monotone, pointing past ``items``) must end here in a ValueError, not in an
        out-of-range read on the device.  One host sync; called once per loaded file, not per request."""
        def bad(msg):
            raise ValueError(f"{where}: {msg}")
        if self.ptr.dtype != torch.int64 or self.ptr.dim() != 1 or self.ptr.numel() != n_users + 1:
            bad(f"ptr must be int64 [{n_users + 1}], got {self.ptr.dtype} {tuple(self.ptr.shape)}")
        if self.items.dtype != torch.int64 or self.items.dim() != 1:
            bad(f"items must be a 1-D int64 tensor, got {self.items.dtype} {tuple(self.items.shape)}")
        if not self.ptr.is_contiguous() or not self.items.is_contiguous():
            bad("ptr and items must be contiguous")
        checks = torch.stack([(self.ptr[0] == 0).reshape(()), (self.ptr[-1] == self.items.numel()).reshape(()),
                              (self.ptr[1:] >= self.ptr[:-1]).all().reshape(())])
        if not bool(checks.all().item()):
            bad("ptr is not a non-decreasing 0 .. len(items) sequence")
        return self

    def for_users(self, users: Tensor) -> "SeenLists":
        return SeenLists(self.ptr, self.items, users)


def mask_topk(scores: Tensor, seen, k: int) -> Tensor:
    """Indices [rows, k] (int64, on the device) of the k largest ``scores * (1 - seen)`` per row, ties by lower index
    (src/lightgcn.py:175-177).  ``seen``: a dense fp32 tensor, a ``SeenLists``, or None.  k <= 256."""
    lists = seen if isinstance(seen, SeenLists) else None
    if lists is not None:
        seen = None
    _native.require_device(scores, "scores")
    if scores.dtype != torch.float32 or scores.dim() != 2 or scores.stride(1) != 1:
        raise TypeError("scores must be a 2-D fp32 tensor with unit inner stride")
    if seen is not None:
        _native.require_device(seen, "seen mask")
        if seen.dtype != torch.float32 or seen.shape != scores.shape or seen.stride(1) != 1:
            raise TypeError("the seen mask must be fp32 of the scores' shape")
    rows, cols = scores.shape
    if not 1 <= k <= cols:
        raise RuntimeError(f"selected index k out of range: k={k}, {cols} columns")      # torch.topk's error class
    out = torch.empty((rows, k), dtype=torch.int64, device=scores.device)
    lib = _native.load()
    with torch.cuda.device(scores.device):
        code = lib.lgc_mask_topk(_native.ptr(scores), scores.stride(0), _native.ptr(seen), 0 if seen is None else seen.stride(0),
                                 _native.ptr(lists.ptr) if lists else None, _native.ptr(lists.items) if lists else None,
                                 _native.ptr(lists.users) if lists else None, rows, cols, k, _native.ptr(out), None,
                                 _native.stream_of(scores.device))
    _native.check(code, "lgc_mask_topk")
    return out
