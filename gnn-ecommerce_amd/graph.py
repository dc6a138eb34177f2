"""The cached propagation graph: CSR of the normalised adjacency plus the launch plan.

The reference hands the model a dense ``[2, E]`` int64 COO on EVERY call and PyG re-derives
the normalisation in every layer.  Here the COO is converted once
per distinct ``(edge_index, edge_weight)`` tensor pair -- by ``lgc_build_csr`` on the device --
and the result is kept in a small cache keyed on tensor identity and version counter (H8).

Layout in HBM (per operator; A for the forward pass, A^T built lazily for the backward pass):
    rowptr   int32 [N+1]
    entries  {int32 col, fp32 val} [E]   8 B per edge, rows in edge order
    tiles    per width class W in (8, 16, 32): rows with at most W entries in PROCESSING ORDER --
             order int32 [slots] (row ids, -1 = padding), meta int32 [tiles] (batch lengths),
             slab {col, val} [slots * W] in the 1 KiB-piece layout of lgc_build_tiles
    chunks   {row, begin, end, slot} int32 [C]   work list for rows longer than ``short_max``
    multi    {row, slot_begin, slot_end, 0} int32 [M]   rows cut into several chunks
    partials fp32 [n_slots, D]   scratch for those rows (allocated per width on first use)
"""
from __future__ import annotations

import os
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
from torch import Tensor

from . import _native

# Rows with at most SHORT_MAX entries are summed by one lane group in entry order; longer rows
# are cut into chunks of about CHUNK_LEN entries, one wavefront each.
SHORT_MAX = int(os.environ.get("LGCN_SHORT_MAX", "32"))
CHUNK_LEN = int(os.environ.get("LGCN_CHUNK_LEN", "256"))
USER_CHUNK_LEN = int(os.environ.get("LGCN_USER_CHUNK_LEN", "4096"))   # the user half of a user|item graph (PropGraph.halves)
# Rows of up to 32 entries go through the tiled kernels (lgc_spmm_tiles) in width classes of 8 / 16 / 32 entries.
TILE_WIDTHS = (8, 16, 32)
USE_TILES = os.environ.get("LGCN_TILES", "1") == "1"
TILES_PER_WAVE = int(os.environ.get("LGCN_TILES_PER_WAVE", "1"))      # 1 ~ 2 > 4 on the full-size user step
# Processing order of the tiled rows.  "cold": rows sorted by their least-gathered column, so the rows that
# share a rarely used table row run close together in time and that row crosses the fabric about once
# (an LRU model of one 4 MiB L2 gives 47 % -> 63 % hits on the user step); "natural": row order.
TILE_ORDER = os.environ.get("LGCN_TILE_ORDER", "cold")
# Band sweep for the long rows of a bipartite half whose gathered table is far larger than the caches (the item
# step): "auto" = when the half has at least SWEEP_MIN_ENTRIES entries, "1" = whenever the table qualifies, "0" = never.
USE_SWEEP = os.environ.get("LGCN_SWEEP", "auto")
SWEEP_MIN_ENTRIES = int(os.environ.get("LGCN_SWEEP_MIN_ENTRIES", "1000000"))
SWEEP_ADAPTIVE_BANDS = os.environ.get("LGCN_SWEEP_ADAPTIVE_BANDS", "1") == "1"
SWEEP_PIECE_ENTRIES = int(os.environ.get("LGCN_SWEEP_PIECE_ENTRIES", "20"))
# 8 wavefronts per CU with 78 accumulators each beat 16 x 39 (44 % vs 20 % L2 hits on the item step: fewer, longer
# lists keep the wavefronts of a band closer together); 160 KiB of LDS per CU either way.
SWEEP_CFG = dict(n_bands=int(os.environ.get("LGCN_SWEEP_BANDS", "8")), waves_per_band_round=int(os.environ.get("LGCN_SWEEP_WAVES", "256")),
                 row_cap=int(os.environ.get("LGCN_SWEEP_ROW_CAP", "78")),
                 piece_cap=int(os.environ.get("LGCN_SWEEP_PIECE_CAP", "64")),
                 lookahead=int(os.environ.get("LGCN_SWEEP_LOOKAHEAD", "64")),
                 groups=4,
                 # rounds by piece weight: a gathered row is fetched once per round that uses it, so the long pieces
                 # share round 0 and the later rounds touch few columns (576 -> 551 us per hop); 2 = odd rounds also walk
                 # their band downwards, starting on the rows the previous round left in the Infinity Cache (-2 us)
                 round_order=int(os.environ.get("LGCN_SWEEP_ROUND_ORDER", "2")))
SWEEP_WIDE = os.environ.get("LGCN_SWEEP_WIDE", "1") == "1"
PLAN_TIMING = os.environ.get("LGCN_PLAN_TIMING", "0") == "1"     # print where PropGraph.prepare spends its time
# tables of 68..96 columns: a row takes two DPP rows, LDS rows are 96 floats -> 51 accumulators per wavefront
SWEEP_CFG_WIDE = dict(SWEEP_CFG, row_cap=int(os.environ.get("LGCN_SWEEP_ROW_CAP_WIDE", "51")), groups=2)


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



def build_row_plan_device(rowptr: Tensor, row_begin: int, row_end: int, short_max: int = SHORT_MAX,
                          chunk_len: int = CHUNK_LEN) -> RowPlan:
    """``build_row_plan`` as three launches + two scans of the library (lgc_row_plan_count / _fill): the same lists without
    torch's nonzero / cumsum / repeat_interleave kernels (lazy code-object loads on their first use in a process)."""
    if short_max < 0 or chunk_len < 1:
        raise ValueError("short_max must be >= 0 and chunk_len >= 1")
    lib = _native.load()
    dev = rowptr.device
    n = row_end - row_begin
    ws_bytes = lib.lgc_row_plan_workspace_bytes(n)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
    totals = torch.empty(3, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        code = lib.lgc_row_plan_count(_native.ptr(rowptr), row_begin, row_end, short_max, chunk_len, _native.ptr(ws), ws_bytes,
                                      _native.ptr(totals), _native.stream_of(dev))
    _native.check(code, "lgc_row_plan_count")
    n_chunks, n_multi, n_slots = totals.tolist()              # the one sync
    chunks = torch.empty((n_chunks, 4), dtype=torch.int32, device=dev)
    multi = torch.empty((n_multi, 4), dtype=torch.int32, device=dev)
    if n_chunks:
        with torch.cuda.device(dev):
            code = lib.lgc_row_plan_fill(_native.ptr(rowptr), row_begin, row_end, short_max, chunk_len, _native.ptr(ws),
                                         _native.ptr(chunks), _native.ptr(multi) if n_multi else _native.ptr(chunks),
                                         _native.stream_of(dev))
        _native.check(code, "lgc_row_plan_fill")
    return RowPlan(row_begin, row_end, short_max, chunks, multi, n_slots)


ROW_PLAN_NATIVE = os.environ.get("LGCN_ROW_PLAN_NATIVE", "1") == "1"


def tile_geometry(width: int) -> Tuple[int, int]:
    """(rows per tile R, batches of four rows B) of a width class: a tile is 1 KiB of row heads (2 KiB for 32)."""
    if width not in TILE_WIDTHS:
        raise ValueError(f"tile width must be one of {TILE_WIDTHS}")
    rows = 128 * (2 if width == 32 else 1) // width
    return rows, rows // 4


def plan_tile_classes(rowptr: Tensor, columns: Tensor, row_begin: int, row_end: int, max_len: int,
                      mode: str = "cold") -> List[Tuple[int, Tensor, Tensor]]:
    """Pure index arithmetic (any device; unit-tested on CPU): [(width, order, meta)] for the rows of
    [row_begin, row_end) with at most min(max_len, 32) entries.

    A row goes to the narrowest class that holds it (0 entries -> width 8: its output is still written).
    ``order``: int32, padded with -1 to whole tiles; inside a tile the longest rows come first and rank rho sits in
    slot (rho % 4) * B + rho // 4, i.e. lane group rho % 4, batch rho // 4 -- the four rows of a batch have similar
    lengths.  ``meta``: int32 per tile, byte bt = longest row of batch bt (how many entries the batch gathers).
    """
    dev = rowptr.device
    rp = rowptr[row_begin:row_end + 1].to(torch.int64)
    deg = rp[1:] - rp[:-1]
    n = row_end - row_begin
    cap = min(int(max_len), TILE_WIDTHS[-1])
    key = None
    if mode == "cold" and n > 0 and int(rp[-1] - rp[0]) > 0:
        cols = columns[int(rp[0]):int(rp[-1])].to(torch.int64)
        pop = torch.bincount(cols)
        rows = torch.repeat_interleave(torch.arange(n, device=dev), deg)
        k = pop[cols] * (int(cols.max()) + 1) + cols                   # popularity major, column id minor
        key = torch.full((n,), torch.iinfo(torch.int64).max, dtype=torch.int64, device=dev)
        key.scatter_reduce_(0, rows, k, "amin")
    elif mode not in ("cold", "natural"):
        raise ValueError("tile order must be 'cold' or 'natural'")
    out = []
    lo = -1
    for width in TILE_WIDTHS:
        hi = min(width, cap)
        sel = torch.nonzero((deg > lo) & (deg <= hi)).flatten()
        lo = max(lo, hi)
        if sel.numel() == 0:
            continue
        if key is not None:
            sel = sel[torch.argsort(key[sel], stable=True)]
        r_tile, b_tile = tile_geometry(width)
        pad = (-sel.numel()) % r_tile
        rows_p = torch.cat([sel, torch.full((pad,), -1, dtype=torch.int64, device=dev)]).view(-1, r_tile)
        d = torch.where(rows_p >= 0, deg[rows_p.clamp(min=0)], torch.full_like(rows_p, -1))   # padding sorts last
        rank = torch.argsort(d, dim=1, descending=True, stable=True)
        rows_s, d_s = torch.gather(rows_p, 1, rank), torch.gather(d, 1, rank).clamp(min=0)
        rho = torch.arange(r_tile, device=dev)
        slot = (rho % 4) * b_tile + rho // 4
        placed = torch.empty_like(rows_s)
        placed[:, slot] = rows_s
        bmax = d_s.view(-1, b_tile, 4).amax(dim=2)
        meta = (bmax << (8 * torch.arange(b_tile, device=dev)).view(1, -1)).sum(dim=1)
        order = torch.where(placed >= 0, placed + row_begin, placed)
        out.append((width, order.reshape(-1).to(torch.int32).contiguous(), meta.to(torch.int32).contiguous()))
    return out


@dataclass
class TileClass:
    """Rows of one width class in the device layout of lgc_build_tiles."""
    width: int
    order: Tensor       # int32 [n_tiles * R]
    meta: Tensor        # int32 [n_tiles]
    slab: Tensor        # int32 [n_tiles * R * width, 2]

    @property
    def n_tiles(self) -> int:
        return self.meta.numel()


def plan_tile_classes_device(rowptr: Tensor, entries: Tensor, row_begin: int, row_end: int, max_len: int,
                             mode: str = "cold") -> List[Tuple[int, Tensor, Tensor]]:
    """``plan_tile_classes`` as four launches of the library (lgc_tile_classes: popularity histogram, row keys, one stable
    radix sort; lgc_tile_pack per class) -- the same lists, without the two dozen torch index kernels whose first use in a
    process costs 0.3 s of lazy code-object loading.  One host sync (the three class sizes)."""
    if mode not in ("cold", "natural"):
        raise ValueError("tile order must be 'cold' or 'natural'")
    lib = _native.load()
    dev = rowptr.device
    n = row_end - row_begin
    table_rows = rowptr.numel() - 1
    ws_bytes = lib.lgc_tile_classes_workspace_bytes(n, table_rows)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
    sorted_rows = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    counts = torch.empty(4, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        code = lib.lgc_tile_classes(_native.ptr(rowptr), _native.ptr(entries), row_begin, row_end, int(max_len), int(mode == "cold"),
                                    table_rows, _native.ptr(ws), ws_bytes, _native.ptr(sorted_rows), _native.ptr(counts),
                                    _native.stream_of(dev))
    _native.check(code, "lgc_tile_classes")
    sizes = counts.tolist()                                   # the one sync
    out, start = [], 0
    for width, size in zip(TILE_WIDTHS, sizes[:3]):
        if size > 0:
            r_tile, _ = tile_geometry(width)
            n_tiles = (size + r_tile - 1) // r_tile
            order = torch.empty(n_tiles * r_tile, dtype=torch.int32, device=dev)
            meta = torch.empty(n_tiles, dtype=torch.int32, device=dev)
            with torch.cuda.device(dev):
                code = lib.lgc_tile_pack(_native.ptr(rowptr), sorted_rows.data_ptr() + 4 * start, size, width, _native.ptr(order),
                                         _native.ptr(meta), _native.stream_of(dev))
            _native.check(code, "lgc_tile_pack")
            out.append((width, order, meta))
        start += size
    return out


TILE_PLAN_NATIVE = os.environ.get("LGCN_TILE_PLAN_NATIVE", "1") == "1"


def build_tile_classes(rowptr: Tensor, entries: Tensor, row_begin: int, row_end: int, max_len: int,
                       mode: Optional[str] = None) -> List[TileClass]:
    lib = _native.load()
    classes = []
    plan = plan_tile_classes_device(rowptr, entries, row_begin, row_end, max_len, mode or TILE_ORDER) if TILE_PLAN_NATIVE \
        else plan_tile_classes(rowptr, entries[:, 0], row_begin, row_end, max_len, mode or TILE_ORDER)
    for width, order, meta in plan:
        slab = torch.empty((order.numel() * width, 2), dtype=torch.int32, device=rowptr.device)
        with torch.cuda.device(rowptr.device):
            code = lib.lgc_build_tiles(_native.ptr(rowptr), _native.ptr(entries), _native.ptr(order), order.numel(), width,
                                       _native.ptr(slab), _native.stream_of(rowptr.device))
        _native.check(code, "lgc_build_tiles")
        classes.append(TileClass(width, order, meta, slab))
    return classes


def _sweep_plan_create(rowptr: Tensor, entries: Tensor, row_begin: int, row_end: int, col_lo: int, col_hi: int,
                       cfg: Optional[dict]):
    """Host copies of rows [row_begin, row_end) -> lgc_sweep_plan_create.  Returns (handle, dims)."""
    import ctypes
    import time
    lib = _native.load()
    cfg = dict(SWEEP_CFG, **(cfg or {}))
    t0 = time.perf_counter()
    rp = rowptr[row_begin:row_end + 1]
    e0, e1 = int(rp[0]), int(rp[-1])
    rp_host = (rp.cpu() - e0).contiguous()          # relative to the first entry of the slice (kept alive across the call)
    ent_host = entries[e0:e1].cpu().contiguous()
    t1 = time.perf_counter()
    c_cfg = _native.SweepCfg(**cfg)
    code = ctypes.c_int(0)
    handle = lib.lgc_sweep_plan_create(rp_host.data_ptr(), ent_host.data_ptr() if e1 > e0 else None, 0,
                                       row_end - row_begin, col_lo, col_hi, ctypes.byref(c_cfg), ctypes.byref(code))
    if PLAN_TIMING:
        print(f"[plan] sweep plan: device -> host copy {t1 - t0:.3f} s, host planner {time.perf_counter() - t1:.3f} s", flush=True)
    if not handle:
        _native.check(code.value or -1, "lgc_sweep_plan_create")
    d = _native.SweepDims()
    try:
        _native.check(lib.lgc_sweep_plan_dims(handle, ctypes.byref(d)), "lgc_sweep_plan_dims")
    except Exception:
        lib.lgc_sweep_plan_free(handle)
        raise
    return handle, d


def sweep_plan_host(rowptr: Tensor, entries: Tensor, row_begin: int, row_end: int, col_lo: int, col_hi: int,
                    cfg: Optional[dict] = None) -> Tuple[dict, Dict[str, Tensor]]:
    """Run the host planner (lgc_sweep_plan_*) on rows [row_begin, row_end) of a CSR given as tensors on any device;
    returns (dims, CPU arrays).  Host-only code: callable without a GPU (the CPU tests decode the plan)."""
    lib = _native.load()
    handle, d = _sweep_plan_create(rowptr, entries, row_begin, row_end, col_lo, col_hi, cfg)
    try:
        dims = {k: getattr(d, k) for k, _ in _native.SweepDims._fields_}
        arrays = {"slabs": torch.empty(max(d.n_slabs, 1) * 64 * d.groups, dtype=torch.int32),
                  "wave_slab_ptr": torch.empty(d.n_waves + 1, dtype=torch.int32),
                  "wave_npieces": torch.empty(max(d.n_waves, 1), dtype=torch.int32),
                  "piece_slot": torch.empty(max(d.n_waves * d.row_cap, 1), dtype=torch.int32),
                  "multi": torch.empty((max(d.n_rows, 1), 4), dtype=torch.int32)}
        _native.check(lib.lgc_sweep_plan_export(handle, *(arrays[k].data_ptr() for k in
                                                          ("slabs", "wave_slab_ptr", "wave_npieces", "piece_slot", "multi"))),
                      "lgc_sweep_plan_export")
    finally:
        lib.lgc_sweep_plan_free(handle)
    arrays["multi"] = arrays["multi"][:d.n_rows].contiguous()
    arrays["multi"][:, 0] += row_begin                 # the planner saw rows 0 .. n of the slice
    return dims, arrays


def sweep_plan_device(rowptr: Tensor, entries: Tensor, row_begin: int, row_end: int, col_lo: int, col_hi: int,
                      cfg: Optional[dict] = None) -> Tuple[dict, Dict[str, Tensor]]:
    """The same plan with its four big arrays copied straight from the planner's memory into device tensors
    (lgc_sweep_plan_upload: no intermediate host tensors, ~90 MB less to copy); ``multi`` comes back on the host."""
    lib = _native.load()
    dev = rowptr.device
    handle, d = _sweep_plan_create(rowptr, entries, row_begin, row_end, col_lo, col_hi, cfg)
    try:
        dims = {k: getattr(d, k) for k, _ in _native.SweepDims._fields_}
        i32 = dict(dtype=torch.int32, device=dev)
        arrays = {"slabs": torch.empty(max(d.n_slabs, 1) * 64 * d.groups, **i32),
                  "wave_slab_ptr": torch.empty(d.n_waves + 1, **i32),
                  "wave_npieces": torch.empty(max(d.n_waves, 1), **i32),
                  "piece_slot": torch.empty(max(d.n_waves * d.row_cap, 1), **i32),
                  "multi": torch.empty((max(d.n_rows, 1), 4), dtype=torch.int32)}
        with torch.cuda.device(dev):
            _native.check(lib.lgc_sweep_plan_upload(handle, *(arrays[k].data_ptr() for k in
                                                              ("slabs", "wave_slab_ptr", "wave_npieces", "piece_slot")),
                                                    _native.stream_of(dev)), "lgc_sweep_plan_upload")
        _native.check(lib.lgc_sweep_plan_export_multi(handle, arrays["multi"].data_ptr()), "lgc_sweep_plan_export_multi")
    finally:
        lib.lgc_sweep_plan_free(handle)
    arrays["multi"] = arrays["multi"][:d.n_rows].contiguous()
    arrays["multi"][:, 0] += row_begin
    return dims, arrays


def sweep_plan_on_device(rowptr: Tensor, entries: Tensor, row_begin: int, row_end: int, col_lo: int, col_hi: int,
                         cfg: Optional[dict] = None) -> Optional[Tuple[dict, Dict[str, Tensor]]]:
    """The plan built by the DEVICE planner (lgc_sweep_dplan_*: the entries never leave the device, the slabs are written
    there; on the host only the piece list and the deal) -- the same arrays as ``sweep_plan_host``, bit for bit.  Returns
    None when the configuration is outside what the device planner takes (piece_cap > 64, lookahead > 64)."""
    import ctypes
    import time
    lib = _native.load()
    dev = rowptr.device
    cfg = dict(SWEEP_CFG, **(cfg or {}))
    t0 = time.perf_counter()
    e0, e1 = int(rowptr[row_begin]), int(rowptr[row_end])
    c_cfg = _native.SweepCfg(**cfg)
    ws_bytes = lib.lgc_sweep_dplan_workspace_bytes(e1 - e0, row_end - row_begin, col_hi - col_lo, ctypes.byref(c_cfg))
    if ws_bytes == 0:
        return None
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    code = ctypes.c_int(0)
    with torch.cuda.device(dev):
        handle = lib.lgc_sweep_dplan_create(_native.ptr(rowptr), _native.ptr(entries), row_begin, row_end, e0, e1 - e0, col_lo,
                                            col_hi, ctypes.byref(c_cfg), _native.ptr(ws), ws_bytes, _native.stream_of(dev),
                                            ctypes.byref(code))
    if not handle:
        if code.value == -4:                       # LGC_E_RANGE: a configuration for the host planner
            return None
        _native.check(code.value or -1, "lgc_sweep_dplan_create")
    try:
        d = _native.SweepDims()
        _native.check(lib.lgc_sweep_dplan_dims(handle, ctypes.byref(d)), "lgc_sweep_dplan_dims")
        dims = {k: getattr(d, k) for k, _ in _native.SweepDims._fields_}
        i32 = dict(dtype=torch.int32, device=dev)
        arrays = {"slabs": torch.empty(max(d.n_slabs, 1) * 64 * d.groups, **i32),
                  "wave_slab_ptr": torch.empty(d.n_waves + 1, **i32),
                  "wave_npieces": torch.empty(max(d.n_waves, 1), **i32),
                  "piece_slot": torch.empty(max(d.n_waves * d.row_cap, 1), **i32),
                  "multi": torch.empty((max(d.n_rows, 1), 4), dtype=torch.int32)}
        with torch.cuda.device(dev):
            _native.check(lib.lgc_sweep_dplan_fill(handle, *(arrays[k].data_ptr() for k in
                                                             ("slabs", "wave_slab_ptr", "wave_npieces", "piece_slot")),
                                                   _native.stream_of(dev)), "lgc_sweep_dplan_fill")
        _native.check(lib.lgc_sweep_dplan_export_multi(handle, arrays["multi"].data_ptr()), "lgc_sweep_dplan_export_multi")
    finally:
        lib.lgc_sweep_dplan_free(handle)
    del ws
    arrays["multi"] = arrays["multi"][:d.n_rows].contiguous()
    if PLAN_TIMING:
        print(f"[plan] sweep plan on the device: {time.perf_counter() - t0:.3f} s", flush=True)
    return dims, arrays


SWEEP_PLAN_ON_DEVICE = os.environ.get("LGCN_SWEEP_PLAN_ON_DEVICE", "1") == "1"


class SweepPlan:
    """Device arrays of a band-sweep plan (include/lgconv_hip.h, lgc_sweep_plan_*): built once per operator half on
    the host from a host copy of its rows, uploaded, then applied with lgc_spmm_sweep."""

    def __init__(self, rowptr: Tensor, entries: Tensor, row_begin: int, row_end: int, col_lo: int, col_hi: int,
                 cfg: Optional[dict] = None):
        dev = rowptr.device
        got = sweep_plan_on_device(rowptr, entries, row_begin, row_end, col_lo, col_hi, cfg) \
            if (rowptr.is_cuda and SWEEP_PLAN_ON_DEVICE) else None
        if got is None:                                # CPU tensors, or a configuration the device planner does not take
            plan = sweep_plan_device if rowptr.is_cuda else sweep_plan_host
            got = plan(rowptr, entries, row_begin, row_end, col_lo, col_hi, cfg)
        self.dims, arrays = got
        self.row_begin, self.row_end = row_begin, row_end
        self.slabs, self.wave_slab_ptr = arrays["slabs"].to(dev), arrays["wave_slab_ptr"].to(dev)
        self.wave_npieces, self.piece_slot = arrays["wave_npieces"].to(dev), arrays["piece_slot"].to(dev)
        # rows with up to WIDE_SLOTS partial slots are summed by one lane group each, the few with more by a wavefront
        m = arrays["multi"]
        wide = (m[:, 2] - m[:, 1]) > self.WIDE_SLOTS
        self.multi, self.multi_wide = m[~wide].contiguous().to(dev), m[wide].contiguous().to(dev)
        self._partials: Dict[int, Tensor] = {}

    WIDE_SLOTS = 32

    def partials(self, dim: int) -> Tensor:
        buf = self._partials.get(dim)
        if buf is None:
            buf = torch.empty((max(self.dims["n_slots"], 1), dim), dtype=torch.float32, device=self.slabs.device)
            self._partials[dim] = buf
        return buf

    def nbytes(self) -> int:
        return sum(t.numel() * 4 for t in (self.slabs, self.wave_slab_ptr, self.wave_npieces, self.piece_slot, self.multi,
                                           self.multi_wide))


@dataclass
class Operator:
    """One sparse operator ready to be applied: CSR + plan (+ per-width scratch).

    ``plan`` covers rows [row_begin, row_end): rows longer than ``plan.short_max`` as chunks; the shorter ones run
    through the tiled kernels when ``tiles`` is given (dim >= 4), else through lgc_spmm's row part."""
    n_rows: int
    rowptr: Tensor
    entries: Tensor                      # int32 [E, 2]: column, fp32 bits of the value
    plan: RowPlan
    tiled: bool = False                  # rows up to plan.short_max go through the tiled kernels
    sweep_cols: Optional[Tuple[int, int]] = None    # column range of a bipartite half that qualifies for the band sweep
    sweep_bands: int = 0                            # bands of that sweep (0: the configured number)
    _sweep: Dict[int, SweepPlan] = field(default_factory=dict)     # by entries per step: 4 (61..64 columns), 2 (68..96)
    _tiles: Optional[List[TileClass]] = None
    _partials: Dict[int, Tensor] = field(default_factory=dict)
    _c_structs: Dict[tuple, list] = field(default_factory=dict)
    _row_stats: Optional[tuple] = None

    @classmethod
    def build(cls, n_rows: int, rowptr: Tensor, entries: Tensor, row_begin: int, row_end: int,
              short_max: int = SHORT_MAX, chunk_len: int = CHUNK_LEN, tiles: Optional[bool] = None,
              sweep_cols: Optional[Tuple[int, int]] = None) -> "Operator":
        """Work plan for rows [row_begin, row_end) of a CSR.  With tiles, every row of at most
        min(short_max, 32) entries is a tiled row and every longer one is chunked.  ``sweep_cols`` = (lo, hi): every
        column of these rows lies in [lo, hi) (a bipartite half) -- the rows may then run as a band sweep."""
        use_tiles = USE_TILES if tiles is None else tiles
        if use_tiles and rowptr.is_cuda:
            short_max = min(short_max, TILE_WIDTHS[-1])
        plan = build_row_plan_device if (ROW_PLAN_NATIVE and rowptr.is_cuda) else build_row_plan
        op = cls(n_rows, rowptr, entries, plan(rowptr, row_begin, row_end, short_max, chunk_len),
                 bool(use_tiles and rowptr.is_cuda))
        if sweep_cols is not None and rowptr.is_cuda and USE_SWEEP != "0" and sweep_cols[1] <= 0xFFFFFF:
            # long rows over a table far beyond the caches: the item half of a user|item graph, not its user half
            n_ent = int(rowptr[row_end]) - int(rowptr[row_begin])
            n_cols = int(sweep_cols[1]) - int(sweep_cols[0])
            n_rows = max(row_end - row_begin, 1)
            long_rows = n_ent >= 8 * n_rows
            # a piece (row x band) should hold ~10 entries or its LDS zero-fill / write-out / combine outweigh the
            # reuse it buys: measured on slices of the item half, 5.1 M entries (93 per row): 325 vs 346 us with the
            # sweep, 2.5 M (46 per row): 208 vs 193, 1.3 M: 134 vs 112
            # Fewer bands for a thinner slice (a rank's item rows at world 2 / 4 / 8 hold 1/2, 1/4, 1/8 of each row's entries):
            # the most bands -- up to one per XCD, at least two -- that still leave SWEEP_PIECE_ENTRIES (20) entries per piece;
            # a band then spans 8 / bands XCDs (block % bands) and the waves per band and round grow so that a round still
            # fills the chip.  Measured (profiles/r04c): one GPU (186 entries per row) 8 bands 531 us per hop, 4 bands 548;
            # world 2 (93): 4 bands 296, 8 bands 307; world 4 (46): 2 bands 167, 4 bands 173; world 8 (23): 2 bands 105.5,
            # 1 band 108.5, 4 bands 112.6 -- i.e. ~23 entries per piece wins everywhere, but never fewer than two bands.
            bands = SWEEP_CFG["n_bands"]
            while SWEEP_ADAPTIVE_BANDS and bands > 2 and n_ent < SWEEP_PIECE_ENTRIES * bands * n_rows:
                bands //= 2
            dense_enough = n_ent >= max(SWEEP_MIN_ENTRIES, 10 * bands * n_rows)
            if long_rows and (USE_SWEEP == "1" or (dense_enough and n_cols >= 1 << 17)):
                op.sweep_cols = (int(sweep_cols[0]), int(sweep_cols[1]))
                op.sweep_bands = bands
        return op

    def sweep_plan(self, groups: int = 4) -> Optional[SweepPlan]:
        """The band-sweep plan for tables whose rows take 64 / groups lanes... i.e. 4 rows (61..64 columns) or 2 rows
        (68..96 columns) per gather instruction; built on first use."""
        if self.sweep_cols is None:
            return None
        plan = self._sweep.get(groups)
        if plan is None:
            p = self.plan
            cfg = dict(SWEEP_CFG_WIDE if groups == 2 else SWEEP_CFG)
            if self.sweep_bands and self.sweep_bands != cfg["n_bands"]:
                cfg.update(n_bands=self.sweep_bands, waves_per_band_round=cfg["waves_per_band_round"] * cfg["n_bands"] // self.sweep_bands)
            plan = SweepPlan(self.rowptr, self.entries, p.row_begin, p.row_end, *self.sweep_cols, cfg=cfg)
            self._sweep[groups] = plan
        return plan

    @property
    def sweep(self) -> Optional[SweepPlan]:
        return self.sweep_plan(4)

    @property
    def tiles(self) -> List[TileClass]:
        """Built on first use: an operator that is only ever restricted to sub-ranges never pays for it."""
        if self._tiles is None:
            p = self.plan
            self._tiles = build_tile_classes(self.rowptr, self.entries, p.row_begin, p.row_end, p.short_max)
        return self._tiles

    @property
    def nnz(self) -> int:
        return self.entries.size(0)

    def listed_rows_pay(self, n_ids: int) -> bool:
        """Whether running this half for a LIST of rows (``apply_rows(split=True)``: the last item step of a scoring
        forward) is expected to beat the full step.  The list's work is the sum of its rows' lengths, repeats included,
        gathered at the all-miss price (about twice what the full step pays per entry), and a training batch's positives
        are drawn in proportion to popularity (src/utils_v2.py:168-181: a random purchase of a random user): with half of
        ~n_ids / 2 listed rows of this half drawn that way and half uniformly, the expected work is
        n_ids / 4 * (sum d^2 / sum d + sum d / rows).  Listed rows are used while that stays below LISTED_ROWS_MAX_SHARE of
        the half's entries -- on the cosmetics-shaped graph 3.3 M of 10.2 M entries for 2 x 1024 pairs (uniform triples:
        0.4 M); a graph whose hubs are heavier keeps the full step.  One host sync, once per operator."""
        if self._row_stats is None:
            p = self.plan
            deg = (self.rowptr[p.row_begin + 1:p.row_end + 1] - self.rowptr[p.row_begin:p.row_end]).double()
            s1, s2 = torch.stack((deg.sum(), (deg * deg).sum())).tolist()
            self._row_stats = (s1, s2, max(p.row_end - p.row_begin, 1))
        s1, s2, rows = self._row_stats
        if s1 <= 0:
            return True
        return n_ids / 4.0 * (s2 / s1 + s1 / rows) <= LISTED_ROWS_MAX_SHARE * s1

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

    def c_struct(self, dim: int, sweep: int) -> "_native.OperatorC":
        """The operator as the C ABI's ``lgc_operator`` (device pointers of tensors this object keeps alive), cached per
        embedding width because the scratch buffers are per width."""
        import ctypes
        key = (dim, sweep)
        got = self._c_structs.get(key)
        if got is not None:
            return got[0]
        p = self.plan
        c = _native.OperatorC()
        c.rowptr, c.entries = _native.ptr(self.rowptr), _native.ptr(self.entries)
        c.chunks = _native.ptr(p.chunks) if p.n_chunks else None
        c.multi = _native.ptr(p.multi) if p.n_multi else None
        c.partials = _native.ptr(self.partials(dim))
        c.row_begin, c.row_end, c.short_max, c.n_chunks, c.n_multi = p.row_begin, p.row_end, p.short_max, p.n_chunks, p.n_multi
        c.tiles_per_wave = TILES_PER_WAVE
        keep = [c]
        if self.tiled and dim >= 4:
            for i, tc in enumerate(self.tiles):
                c.tiles[i] = _native.TileClassC(_native.ptr(tc.order), _native.ptr(tc.meta), _native.ptr(tc.slab), tc.n_tiles,
                                                tc.width)
            c.n_tile_classes = len(self.tiles)
        if sweep:                                   # 4 or 2 = lgc_sweep_ok's answer for this table
            sw = self.sweep_plan(int(sweep) if int(sweep) in (2, 4) else 4)
            sc = _native.SweepArraysC(_native.ptr(sw.slabs), _native.ptr(sw.wave_slab_ptr), _native.ptr(sw.wave_npieces),
                                      _native.ptr(sw.piece_slot), _native.ptr(sw.multi) if sw.multi.size(0) else None,
                                      _native.ptr(sw.multi_wide) if sw.multi_wide.size(0) else None,
                                      _native.ptr(sw.partials(dim)), sw.dims["n_waves"], sw.dims["row_cap"], sw.multi.size(0),
                                      sw.multi_wide.size(0), sw.dims["groups"])
            c.sweep = ctypes.pointer(sc)
            keep.append(sc)
        self._c_structs[key] = keep
        return c

    def apply(self, x: Tensor, out: Tensor, a: float = 1.0, r: Optional[Tensor] = None, b: float = 0.0) -> Tensor:
        """out[row] = a * (A x)[row] + b * r[row] for the rows of the plan: one ``lgc_apply`` on the current stream."""
        import ctypes
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
        table_rows = min(x.size(0), out.size(0))
        sweep = sweep_choice(lib, dim, table_rows, x.stride(0)) if self.sweep_cols is not None else 0
        c = self.c_struct(dim, sweep)
        with torch.cuda.device(x.device):
            code = lib.lgc_apply(ctypes.byref(c), table_rows, _native.ptr(x), x.stride(0), _native.ptr(out), out.stride(0),
                                 _native.ptr(r), 0 if r is None else r.stride(0), float(a), float(b), dim,
                                 _native.stream_of(x.device))
        _native.check(code, "lgc_apply")
        return out


_rows_scratch: Dict[tuple, Tuple[Tensor, Tensor]] = {}
LISTED_ROWS_MAX_SHARE = float(os.environ.get("LGCN_LISTED_ROWS_MAX_SHARE", "0.4"))   # Operator.listed_rows_pay
ROWS_SPLIT_ROOM = 16384      # partial rows beyond one per list position: 256-entry chunks up to 4 M listed entries


def _rows_split_scratch(device: torch.device, n_ids: int, dim: int) -> Tuple[Tensor, Tensor]:
    """Scratch of lgc_spmm_rows_split (chunk numbers, partial rows), owned by one (device, stream, host thread) triple like
    every other scratch buffer that lives across launches: two steps on different streams never share it."""
    import threading
    key = (device, torch.cuda.current_stream(device).cuda_stream, threading.get_ident(), dim)
    got = _rows_scratch.get(key)
    if got is None or got[0].numel() < n_ids + 2:
        cap = max(n_ids, 4096)
        got = (torch.empty(cap + 2, dtype=torch.int32, device=device),
               torch.empty((cap + ROWS_SPLIT_ROOM, dim), dtype=torch.float32, device=device))
        _rows_scratch[key] = got
    return got


def apply_rows(op: "Operator", rows: Tensor, x: Tensor, out: Tensor, a: float = 1.0, r: Optional[Tensor] = None,
               b: float = 0.0, split: bool = False, compact: bool = False) -> Tensor:
    """out[row] = a * (A x)[row] + b * r[row] for the rows listed in ``rows`` (int64, on the device; ids outside the
    operator's plan range are skipped, repeats are harmless); every other row of ``out`` is left untouched.
    ``lgc_spmm_rows``: one wavefront per listed row -- right for user rows (6 entries on average).  ``split=True``
    (``lgc_spmm_rows_split``): the listed rows are cut into chunks on the device first -- for item rows (186 entries on
    average, hubs beyond 10^5).  ``compact=True`` (with ``split``): ``out`` is a [len(rows), D] table in list order."""
    _check_table(x, "x")
    _check_table(out, "out")
    if r is not None:
        _check_table(r, "r")
    _native.require_device(rows, "rows")
    if rows.dtype != torch.int64 or rows.dim() != 1:
        raise TypeError("rows must be a 1-D int64 tensor")
    dim = x.size(1)
    if out.size(1) != dim or (r is not None and r.size(1) != dim):
        raise ValueError("x, out and r must have the same width")
    if compact and not split:
        raise ValueError("compact output needs split=True")
    rows = rows.contiguous()
    lib = _native.load()
    p = op.plan
    n_ids = rows.numel()
    with torch.cuda.device(x.device):
        if split:
            if n_ids == 0:
                return out
            if out.size(0) < (n_ids if compact else p.row_end):
                raise ValueError("out has too few rows")
            work, partials = _rows_split_scratch(x.device, n_ids, dim)
            code = lib.lgc_spmm_rows_split(_native.ptr(op.rowptr), _native.ptr(op.entries), p.row_begin, p.row_end, _native.ptr(rows),
                                           n_ids, x.size(0), _native.ptr(x), x.stride(0), _native.ptr(out), out.stride(0),
                                           out.size(0), _native.ptr(r), 0 if r is None else r.stride(0), float(a), float(b), dim,
                                           int(compact), _native.ptr(work), _native.ptr(partials), partials.size(0),
                                           _native.stream_of(x.device))
            _native.check(code, "lgc_spmm_rows_split")
            return out
        code = lib.lgc_spmm_rows(_native.ptr(op.rowptr), _native.ptr(op.entries), p.row_begin, p.row_end, _native.ptr(rows),
                                 n_ids, min(x.size(0), out.size(0)), _native.ptr(x), x.stride(0), _native.ptr(out),
                                 out.stride(0), _native.ptr(r), 0 if r is None else r.stride(0), float(a), float(b), dim,
                                 _native.stream_of(x.device))
    _native.check(code, "lgc_spmm_rows")
    return out


def sweep_choice(lib, dim: int, table_rows: int, stride: int) -> int:
    """Which band-sweep plan a gathered table of this width gets: 4 / 2 (entries per step), 0 = chunk path."""
    sweep = int(lib.lgc_sweep_ok(dim, table_rows, stride))
    return 0 if (sweep == 2 and not SWEEP_WIDE) else sweep


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
        return Operator.build(n, rowptr, entries, lo, hi, self.short_max, self.chunk_len)

    def _find_bipartite_split(self) -> Optional[int]:
        """s such that every edge joins a node < s ("users") with a node >= s ("items") -- the layout of
        src/utils_v2.py:146-165 -- or None.  Two reductions over the COO, one host sync per graph."""
        if self.num_edges == 0:
            return None
        out = torch.empty(2, dtype=torch.int64, device=self.device)
        lib = _native.load()
        with torch.cuda.device(self.device):
            code = lib.lgc_bipartite_split(_native.ptr(self._edge_index), self.num_edges, _native.ptr(out),
                                           _native.stream_of(self.device))
        _native.check(code, "lgc_bipartite_split")
        lo, hi = out.tolist()
        return lo + 1 if lo < hi else None

    def halves(self, transpose: bool = False) -> Tuple[Operator, Operator]:
        """(user-row operator, item-row operator): the same CSR with work plans limited to rows [0, split)
        and [split, N).  Only for bipartite graphs."""
        if self.split is None:
            raise ValueError("graph is not bipartite")
        got = self._halves.get(transpose)
        if got is None:
            op = self.transpose_op if transpose else self.forward_op
            # every column of a user row is an item and vice versa: each half may sweep the other side's range
            # user rows above short_max are few (12 k of 1.6 M on the cosmetics graph, the longest 2.6 k entries): one
            # chunk each, started first, finishes inside the launch and spares the half its combine launch (3 x 4.6 us of
            # a 1.6 ms step); the item rows (up to 87 k entries) keep the short chunks
            got = tuple(Operator.build(op.n_rows, op.rowptr, op.entries, lo, hi, self.short_max, chunk, sweep_cols=cols)
                        for (lo, hi), cols, chunk in (((0, self.split), (self.split, self.num_nodes),
                                                       max(self.chunk_len, USER_CHUNK_LEN)),
                                                      ((self.split, self.num_nodes), (0, self.split), self.chunk_len)))
            self._halves[transpose] = got
        return got

    def prepare(self, dim: int, table_rows: Optional[int] = None, transpose: bool = False) -> "PropGraph":
        """Build, now, every work plan a propagation of width ``dim`` will use (they are otherwise built lazily inside
        the first hop): tile classes of the short rows (device index arithmetic + lgc_build_tiles), and for a
        user|item graph the band-sweep plan of the item half (device -> host copy of its rows, host planner, upload).
        A serving worker's ``initialize`` and bench.py's ``plan_build_s`` call this; training pays it twice (A and A^T)."""
        lib = _native.load()
        rows = self.num_nodes if table_rows is None else int(table_rows)
        stride = dim if dim % 32 == 0 else (dim + 31) // 32 * 32            # propagate.scratch_table's row stride
        ops = self.halves(transpose) if self.split is not None else ((self.transpose_op if transpose else self.forward_op),)
        import time
        for op in ops:
            t0 = time.perf_counter()
            if op.tiled and dim >= 4:
                op.tiles
            if PLAN_TIMING:
                torch.cuda.synchronize(self.device)
                print(f"[plan] rows [{op.plan.row_begin}, {op.plan.row_end}): tile classes {time.perf_counter() - t0:.3f} s", flush=True)
            t0 = time.perf_counter()
            if op.sweep_cols is not None:
                groups = sweep_choice(lib, dim, rows, stride)
                if groups:
                    op.sweep_plan(groups)
            if PLAN_TIMING:
                torch.cuda.synchronize(self.device)
                print(f"[plan] rows [{op.plan.row_begin}, {op.plan.row_end}): sweep plan total {time.perf_counter() - t0:.3f} s", flush=True)
        torch.cuda.synchronize(self.device)
        return self

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
    def save(self, path: str, extra: Optional[Dict[str, Tensor]] = None, meta: Optional[Dict[str, str]] = None) -> None:
        """Write the forward operator's CSR, degrees and metadata; nothing executable in the file.  ``extra`` tensors
        (stored under ``extra.<name>``) and ``meta`` strings ride along -- id maps, purchased-items lists.
        A serving worker can ``PropGraph.load`` it instead of re-reading the CSV, rebuilding the COO
        (torchserve/lightgcn_handler.py:32-38) and sorting it again.  The tile layout is derived data and is
        rebuilt at load time (milliseconds on the device)."""
        from safetensors.torch import save_file
        op = self.forward_op
        tensors = {"rowptr": op.rowptr, "entries": op.entries, "deg": self.deg, "dis": self.dis}
        for k, v in (extra or {}).items():
            tensors["extra." + k] = v
        meta = {**{"user." + k: str(v) for k, v in (meta or {}).items()},
                "format": "lgcn-graph-2", "num_nodes": str(self.num_nodes), "num_edges": str(self.num_edges),
                "normalize": str(int(self.normalize)), "split": "" if self.split is None else str(self.split)}
        save_file({k: v.detach().cpu().contiguous() for k, v in tensors.items()}, path, metadata=meta)

    @staticmethod
    def validate_csr(rowptr: Tensor, entries: Tensor, deg: Tensor, dis: Tensor, num_nodes: int, num_edges: int,
                     split: Optional[int], where: str) -> None:
        """Everything the kernels take on trust, checked once: a truncated, stale or corrupt file must end in a
        ValueError here, not in an out-of-range gather on the device."""
        def bad(msg):
            raise ValueError(f"{where}: {msg}")
        if num_nodes < 0 or num_edges < 0 or num_nodes >= 2 ** 31 - 1 or num_edges >= 2 ** 31 - 1:
            bad("node or edge count out of range")
        if rowptr.dtype != torch.int32 or rowptr.shape != (num_nodes + 1,):
            bad(f"rowptr must be int32 [{num_nodes + 1}], got {rowptr.dtype} {tuple(rowptr.shape)}")
        if entries.dtype != torch.int32 or entries.shape != (num_edges, 2):
            bad(f"entries must be int32 [{num_edges}, 2], got {entries.dtype} {tuple(entries.shape)}")
        for name, t in (("deg", deg), ("dis", dis)):
            if t.dtype != torch.float32 or t.shape != (num_nodes,):
                bad(f"{name} must be fp32 [{num_nodes}], got {t.dtype} {tuple(t.shape)}")
        if split is not None and not 0 < split < num_nodes:
            bad(f"bipartite split {split} outside (0, {num_nodes})")
        rp = rowptr.to(torch.int64)
        checks = [rp[0] == 0, rp[-1] == num_edges, (rp[1:] >= rp[:-1]).all()]
        if num_edges:
            col = entries[:, 0]
            checks += [col.min() >= 0, col.max() < num_nodes]
        if split is not None and num_edges:
            # a stored split that the entries do not honour would make bipartite_sum return wrong tables silently:
            # rows < split may only hold columns >= split and the reverse
            rp_ok = bool(((rp[1:] >= rp[:-1]).all() & (rp[0] == 0) & (rp[-1] == num_edges)).item())
            if rp_ok:
                e_split = rp[split]                                            # first entry of row `split`
                pos = torch.arange(num_edges, device=entries.device)
                user_side = pos < e_split
                checks.append(torch.where(user_side, entries[:, 0] >= split, entries[:, 0] < split).all())
        if not bool(torch.stack([c.reshape(()) for c in checks]).all().item()):     # one host sync
            bad("row pointer is not a non-decreasing 0 .. num_edges sequence, a column id is outside the graph, or "
                "the stored bipartite split does not separate the rows' columns")

    @classmethod
    def load(cls, path: str, device, short_max: int = SHORT_MAX, chunk_len: int = CHUNK_LEN, with_extra: bool = False):
        """A graph usable for forward propagation (``get_embedding`` / ``recommendK``).  The COO is not stored,
        so the transposed operator (training) cannot be derived from a loaded graph.  ``with_extra``: return
        (graph, extra tensors on the host, user metadata) as written by ``save(extra=..., meta=...)``."""
        from safetensors import safe_open
        device = torch.device(device)
        if device.type != "cuda":
            raise _native.NativeLibraryError("graphs are loaded onto a ROCm device only (no CPU fallback)")
        with safe_open(path, framework="pt", device="cpu") as f:
            meta = f.metadata() or {}
            if meta.get("format") not in ("lgcn-graph-1", "lgcn-graph-2"):
                raise ValueError(f"{path} is not a saved propagation graph")
            t = {k: f.get_tensor(k) for k in f.keys()}
        for k in ("rowptr", "entries", "deg", "dis"):
            if k not in t:
                raise ValueError(f"{path}: tensor {k!r} is missing")
        try:
            num_nodes, num_edges = int(meta["num_nodes"]), int(meta["num_edges"])
            normalize = bool(int(meta["normalize"]))
            split = int(meta["split"]) if meta.get("split") else None
        except (KeyError, ValueError) as exc:
            raise ValueError(f"{path}: bad metadata ({exc})") from exc
        g = cls.__new__(cls)
        g.device, g.num_nodes, g.num_edges, g.normalize = device, num_nodes, num_edges, normalize
        g.short_max, g.chunk_len = short_max, chunk_len
        g._key_refs = g._edge_index = g._edge_weight = None
        g.status = torch.zeros(4, dtype=torch.int32, device=device)
        rowptr, entries = t["rowptr"].to(device), t["entries"].to(device)
        g.deg, g.dis, g.edge_values = t["deg"].to(device), t["dis"].to(device), None
        cls.validate_csr(rowptr, entries, g.deg, g.dis, num_nodes, num_edges, split, path)
        g.forward_op = Operator.build(num_nodes, rowptr, entries, 0, num_nodes, short_max, chunk_len)
        g._transpose_op = None
        g.split = split
        g._halves = {}
        if with_extra:
            return (g, {k[len("extra."):]: v for k, v in t.items() if k.startswith("extra.")},
                    {k[len("user."):]: v for k, v in meta.items() if k.startswith("user.")})
        return g

    def nbytes(self) -> int:
        ops = [self.forward_op] + ([self._transpose_op] if self._transpose_op is not None else [])
        return sum(o.rowptr.numel() * 4 + o.entries.numel() * 4 + o.plan.chunks.numel() * 4
                   + sum(tc.slab.numel() * 4 + tc.order.numel() * 4 + tc.meta.numel() * 4 for tc in (o._tiles or []))
                   for o in ops)


# ----------------------------------------------------------------------------------------
# cache keyed on tensor identity + version (the callers pass the same tensors every step)
# ----------------------------------------------------------------------------------------
_CACHE: "OrderedDict[tuple, PropGraph]" = OrderedDict()
_CACHE_SIZE = int(os.environ.get("LGCN_GRAPH_CACHE", "4"))


def _key(edge_index: Tensor, edge_weight: Optional[Tensor], num_nodes: int, normalize: bool) -> tuple:
    w = None if edge_weight is None else (edge_weight.data_ptr(), edge_weight._version, tuple(edge_weight.shape),
                                          tuple(edge_weight.stride()))
    return (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape), tuple(edge_index.stride()),
            str(edge_index.device), w, int(num_nodes), bool(normalize))


def get_graph(edge_index, edge_weight: Optional[Tensor], num_nodes: int, normalize: bool = True) -> PropGraph:
    """The cached graph of an ``(edge_index, edge_weight)`` pair; a ``PropGraph`` passed in place of ``edge_index``
    (a loaded, persisted graph) is used as it is."""
    if isinstance(edge_index, PropGraph):
        if edge_index.num_nodes != int(num_nodes):
            raise ValueError(f"the graph has {edge_index.num_nodes} nodes, the model {num_nodes}")
        return edge_index
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
