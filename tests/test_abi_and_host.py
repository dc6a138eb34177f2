"""CPU-only checks: the C-ABI library loads and exports every symbol include/lgconv_hip.h declares
(no compute calls -- there is no GPU here), argument validation that happens before any launch, and the
host-side logic (work plan, user-range partition, synthetic-graph generator)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT
import gnn_ecommerce_amd as lg
from gnn_ecommerce_amd import _native, synth
from gnn_ecommerce_amd.graph import build_row_plan
from gnn_ecommerce_amd.partition import balanced_user_ranges, check_bipartite

HEADER = os.path.join(ROOT, "include", "lgconv_hip.h")


def declared_symbols():
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(lgc_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    lib = _native.load()
    names = declared_symbols()
    assert names == sorted(_native.SIGNATURES), "ctypes table and header must list the same entry points"
    for name in names:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert lib.lgc_abi_version() == _native.ABI_VERSION
    header_version = int(re.search(r"#define LGC_ABI_VERSION (\d+)", open(HEADER).read()).group(1))
    assert header_version == _native.ABI_VERSION


def test_no_torch_types_in_the_abi():
    code = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)            # declarations only, comments stripped
    assert "torch" not in code.lower() and "at::" not in code and "hip/" not in code and "Tensor" not in code
    assert set(re.findall(r"#include <(.*?)>", code)) == {"stddef.h", "stdint.h"}


def test_library_links_one_hip_runtime_and_no_torch():
    out = os.popen(f"readelf -d {_native.LIB_PATH}").read()
    needed = re.findall(r"NEEDED.*\[(.*?)\]", out)
    assert any("amdhip64" in n for n in needed)
    assert not any("torch" in n or "c10" in n for n in needed)
    assert len([p for p in _native.runtime_libraries() if "libamdhip64" in p]) == 1


def test_error_strings_and_dim_support():
    lib = _native.load()
    assert lib.lgc_error_string(0) == b"ok"
    for code in (-1, -2, -3, -4, -5):
        assert lib.lgc_error_string(code) not in (b"ok", b"unknown error")
    for dim in (1, 2, 3, 4, 5, 7, 16, 63, 64, 65, 80, 90, 96, 128, 129, 130, 255, 256):
        assert lib.lgc_dim_ok(dim) == 1
    for dim in (0, -4, 257, 300):                   # a row must fit one 64-lane wavefront of float4 slices
        assert lib.lgc_dim_ok(dim) == 0
    assert lib.lgc_build_workspace_bytes(1000, 5000) >= 4 * 5000 * 4
    assert lib.lgc_build_workspace_bytes(10, 2 ** 31) == 0          # does not fit int32


def test_argument_errors_are_reported_before_any_launch():
    lib = _native.load()
    one = ctypes.c_void_p(16)            # never dereferenced: every call below must fail validation first
    assert lib.lgc_spmm(one, one, 0, 4, 32, None, 0, None, 0, None, 8, one, 64, ctypes.c_void_p(32), 64, None, 0,
                        1.0, 0.0, 300, None) == -2                                      # LGC_E_DIM
    assert lib.lgc_spmm(one, one, 0, 4, 32, None, 0, None, 0, None, 8, one, 64, one, 64, None, 0,
                        1.0, 0.0, 64, None) == -1                                       # y aliases x
    assert lib.lgc_spmm(one, one, 0, 4, 32, None, 0, None, 0, None, 8, ctypes.c_void_p(18), 64, ctypes.c_void_p(32), 64,
                        None, 0, 1.0, 0.0, 64, None) == -5                              # x not dword-aligned
    assert lib.lgc_spmm(one, one, 0, 4, 32, None, 3, None, 0, None, 8, one, 64, ctypes.c_void_p(32), 64, None, 0,
                        1.0, 0.0, 64, None) == -1                                       # chunks missing
    two = ctypes.c_void_p(32)
    tiles = lambda **kw: lib.lgc_spmm_tiles(one, one, one, kw.get("n", 4), kw.get("w", 8), kw.get("tpw", 1),
                                            100, one, kw.get("xs", 64), kw.get("y", two), 64, None, 0, 1.0, 0.0,
                                            kw.get("dim", 64), None)
    assert tiles(dim=300) == -2 and tiles(dim=3) == -2                                    # LGC_E_DIM (tiles need >= 4)
    assert tiles(w=12) == -1 and tiles(tpw=0) == -1 and tiles(y=one) == -1
    assert tiles(xs=32) == -1 and tiles(n=-1) == -1
    assert tiles(n=0) == 0                                                                # nothing to do: no launch
    assert lib.lgc_build_tiles(one, one, one, 15, 8, one, None) == -1                     # not whole tiles
    assert lib.lgc_build_tiles(one, one, one, 16, 9, one, None) == -1
    assert lib.lgc_build_tiles(one, one, one, 0, 8, one, None) == 0
    assert lib.lgc_build_csr(one, None, -1, 5, 0, 1, None, one, one, None, one, one, one, 1 << 20, one, None) == -1
    assert lib.lgc_build_csr(one, None, 10, 5, 1, 1, None, one, one, None, one, one, one, 1 << 20, one, None) == -1
    assert lib.lgc_build_csr(one, None, 10, 5, 0, 1, None, one, one, None, one, one, ctypes.c_void_p(256), 8, one,
                             None) == -3                                                # workspace too small
    assert lib.lgc_pair_dot(one, 8, 64, 10, one, one, 4, one, one, None) == -1          # stride < dim


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", str(tmp_path / "liblgconv_hip.so"))
    with pytest.raises(_native.NativeLibraryError, match="no CPU or PyTorch fallback"):
        _native.load()


def test_cpu_tensors_are_refused_everywhere():
    model = lg.LightGCN(12, 64, 2)
    ei = torch.tensor([[0, 5], [5, 0]])
    with pytest.raises(_native.NativeLibraryError):
        model.get_embedding(ei, None)
    with pytest.raises(_native.NativeLibraryError):
        model(ei)
    with pytest.raises(_native.NativeLibraryError):
        lg.LGConv()(torch.zeros(12, 64), ei)
    with pytest.raises(_native.NativeLibraryError):
        lg.pair_dot(torch.zeros(12, 64), ei)


def test_model_surface_matches_the_reference_contract():
    """constructor, attributes, parameters(), state_dict keys, alpha handling, repr."""
    m = lg.LightGCN(100, 90, 5)
    assert [k for k, _ in m.state_dict().items()] == ["alpha", "embedding.weight"]
    assert [tuple(p.shape) for p in m.parameters()] == [(100, 90)]
    assert torch.allclose(m.alpha, torch.full((6,), 1 / 6)) and len(m.convs) == 5
    assert all(len(list(c.parameters())) == 0 for c in m.convs)
    assert repr(m) == "LightGCN(100, 90, num_layers=5)"
    assert torch.equal(lg.LightGCN(10, 8, 2, alpha=0.3).alpha, torch.tensor([0.3, 0.3, 0.3]))
    a = torch.tensor([0.5, 0.25, 0.25])
    assert torch.equal(lg.LightGCN(10, 8, 2, alpha=a).alpha, a)
    with pytest.raises(AssertionError):
        lg.LightGCN(10, 8, 2, alpha=torch.ones(2))
    assert lg.LightGCN(10, 8, 1, normalize=False).convs[0].normalize is False
    with pytest.raises(TypeError):
        m.recommend(torch.zeros(2, 2, dtype=torch.long))         # upstream's broken method stays broken
    sd = {"alpha": torch.full((6,), 0.1), "embedding.weight": torch.zeros(100, 90)}
    m.load_state_dict(sd)
    bound = (6 / (100 + 90)) ** 0.5
    m.reset_parameters()
    assert m.embedding.weight.abs().max() <= bound and m.embedding.weight.std() > 0


def test_losses_match_oracle_on_cpu():
    """BPRLoss / link_pred_loss are plain torch and device-agnostic."""
    from oracle import lightgcn_oracle as oracle
    g = torch.Generator().manual_seed(0)
    pos, neg, par = torch.randn(64, generator=g), torch.randn(64, generator=g), torch.randn(30, 8, generator=g)
    assert torch.equal(lg.BPRLoss()(pos, neg), oracle.bpr_loss(pos, neg))
    assert torch.equal(lg.BPRLoss(1e-4)(pos, neg, par), oracle.bpr_loss(pos, neg, par, 1e-4))
    m = lg.LightGCN(30, 8, 1)
    assert torch.equal(m.recommendation_loss(pos, neg, 0), oracle.bpr_loss(pos, neg))
    y = (torch.rand(64, generator=g) > 0.5)
    assert torch.allclose(m.link_pred_loss(pos, y), torch.nn.functional.binary_cross_entropy_with_logits(pos, y.float()))


# ---------------------------------------------------------------------------------------------
# work plan
# ---------------------------------------------------------------------------------------------
def check_plan(rowptr, lo, hi, short_max, chunk_len):
    plan = build_row_plan(rowptr, lo, hi, short_max, chunk_len)
    deg = (rowptr[1:] - rowptr[:-1]).long()
    covered = torch.zeros(int(rowptr[-1]), dtype=torch.int64)
    rows_seen = {}
    for row, b, e, slot in plan.chunks.tolist():
        assert lo <= row < hi and deg[row] > short_max
        assert rowptr[row] <= b < e <= rowptr[row + 1] and e - b <= chunk_len
        covered[b:e] += 1
        rows_seen.setdefault(row, []).append((b, e, slot))
    for row in range(lo, hi):
        if deg[row] > short_max:
            parts = rows_seen[row]
            assert parts[0][0] == rowptr[row] and parts[-1][1] == rowptr[row + 1]
            assert all(parts[i][1] == parts[i + 1][0] for i in range(len(parts) - 1))      # contiguous, in order
            if len(parts) == 1:
                assert parts[0][2] == -1
            else:
                slots = [p[2] for p in parts]
                assert slots == list(range(slots[0], slots[0] + len(parts)))
            s, e = int(rowptr[row]), int(rowptr[row + 1])
            assert (covered[s:e] == 1).all()
        else:
            assert row not in rows_seen
    multi = {r: (sb, se) for r, sb, se, _ in plan.multi.tolist()}
    assert set(multi) == {r for r, p in rows_seen.items() if len(p) > 1}
    for r, (sb, se) in multi.items():
        assert [p[2] for p in rows_seen[r]] == list(range(sb, se))
    assert plan.n_slots == sum(se - sb for sb, se in multi.values())
    all_slots = sorted(s for p in rows_seen.values() for _, _, s in p if s >= 0)
    assert all_slots == list(range(plan.n_slots))
    return plan


@pytest.mark.parametrize("short_max,chunk_len", [(0, 1), (0, 5), (3, 4), (32, 256), (10 ** 6, 256)])
def test_row_plan_covers_every_long_row_exactly_once(short_max, chunk_len):
    rng = np.random.default_rng(0)
    deg = np.concatenate([rng.integers(0, 8, 200), [0, 0, 1, 1000, 257, 256, 255, 33, 32], rng.integers(0, 90, 50)])
    rowptr = torch.from_numpy(np.concatenate([[0], np.cumsum(deg)])).int()
    n = len(deg)
    check_plan(rowptr, 0, n, short_max, chunk_len)
    check_plan(rowptr, 37, n - 11, short_max, chunk_len)
    empty = check_plan(rowptr, 5, 5, short_max, chunk_len)
    assert empty.n_chunks == 0 and empty.n_multi == 0


def test_row_plan_rejects_bad_parameters():
    rowptr = torch.tensor([0, 3, 9], dtype=torch.int32)
    with pytest.raises(ValueError):
        build_row_plan(rowptr, 0, 2, -1, 16)
    with pytest.raises(ValueError):
        build_row_plan(rowptr, 0, 2, 4, 0)


# ---------------------------------------------------------------------------------------------
# partition
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_balanced_ranges_tile_and_balance(world):
    rng = np.random.default_rng(1)
    deg = torch.from_numpy(np.minimum(rng.pareto(2.0, 5000) * 3 + 1, 400).astype(np.int64))
    ranges = balanced_user_ranges(deg, world)
    assert ranges[0][0] == 0 and ranges[-1][1] == 5000
    assert all(ranges[k][1] == ranges[k + 1][0] for k in range(world - 1))
    loads = [int(deg[a:b].sum()) for a, b in ranges]
    assert sum(loads) == int(deg.sum())
    assert max(loads) - min(loads) <= 2 * int(deg.max())          # within one (heaviest) user of even


def test_balanced_ranges_degenerate_inputs():
    assert balanced_user_ranges(torch.tensor([5]), 4) == [(0, 1), (1, 1), (1, 1), (1, 1)]
    assert balanced_user_ranges(torch.zeros(0, dtype=torch.int64), 2) == [(0, 0), (0, 0)]
    r = balanced_user_ranges(torch.tensor([0, 0, 10, 0]), 2)
    assert r[0][0] == 0 and r[-1][1] == 4 and r[0][1] == r[1][0]
    with pytest.raises(ValueError):
        balanced_user_ranges(torch.tensor([1, 2]), 0)


def test_check_bipartite():
    good = torch.tensor([[0, 1, 3, 4], [3, 4, 0, 1]])
    check_bipartite(good, 3, 2)
    with pytest.raises(ValueError):
        check_bipartite(torch.tensor([[0, 1], [1, 3]]), 3, 2)       # user-user edge
    with pytest.raises(ValueError):
        check_bipartite(torch.tensor([[0], [5]]), 3, 2)             # item id out of range


# ---------------------------------------------------------------------------------------------
# synthetic generator (SURVEY.md 8d rules)
# ---------------------------------------------------------------------------------------------
def test_generator_rules_config1():
    g = synth.make_bipartite(**synth.CONFIG_SMALL, seed=0)
    assert len(g.user) == 120_000 and g.nnz == 240_000
    keys = g.user * g.n_items + g.item
    assert len(np.unique(keys)) == len(keys)                                    # unique pairs
    assert np.bincount(g.user, minlength=g.n_users).min() >= 1                  # every node covered
    assert np.bincount(g.item, minlength=g.n_items).min() >= 1
    assert set(np.unique(g.weight)).issubset(set(synth.WEIGHT_VALUES)) and g.weight.min() > 0
    assert 0.10 < (g.weight == 1.0).mean() < 0.16
    assert not np.all(np.diff(keys) > 0)                                        # shuffled
    ei, ew = g.coo()
    half = ei.size(1) // 2
    assert ei.dtype == torch.int64 and ew.dtype == torch.float32 and ei.shape == (2, 240_000)
    assert torch.equal(ei[0, :half], ei[1, half:]) and torch.equal(ei[1, :half], ei[0, half:])
    assert ei[0, :half].max() < g.n_users <= ei[1, :half].min() and ei.max() == g.num_nodes - 1
    g2 = synth.make_bipartite(**synth.CONFIG_SMALL, seed=0)
    assert np.array_equal(g.user, g2.user) and np.array_equal(g.weight, g2.weight)          # seeded
    assert not np.array_equal(g.user, synth.make_bipartite(**synth.CONFIG_SMALL, seed=1).user)
    with pytest.raises(ValueError):
        synth.make_bipartite(100, 10, 50)


def test_algorithmic_bytes_match_baseline_md():
    n, nnz = 1_693_929, 20_314_816
    assert synth.algorithmic_bytes_per_layer(n, nnz, 64) == 1_036_585_896
    assert synth.algorithmic_bytes_per_layer(n, nnz, 90) == 1_388_923_128
    c = synth.CONFIG_COSMETICS
    assert c["n_users"] + c["n_items"] == n and 2 * c["n_pairs"] == nnz


def test_xavier_table_bound_and_seed():
    w = synth.xavier_table(1000, 64, 3)
    assert w.abs().max() <= (6 / 1064) ** 0.5 and torch.equal(w, synth.xavier_table(1000, 64, 3))


# ----------------------------------------------------------------------------------------------
# tile planner (pure index arithmetic) and saved-graph validation
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["cold", "natural"])
@pytest.mark.parametrize("max_len", [0, 4, 8, 20, 32, 1000])
def test_tile_plan_lists_every_short_row_once_in_the_narrowest_class(mode, max_len):
    from gnn_ecommerce_amd.graph import TILE_WIDTHS, plan_tile_classes, tile_geometry
    gen = torch.Generator().manual_seed(max_len + 17)
    n, lo, hi = 700, 37, 655
    deg = torch.randint(0, 12, (n,), generator=gen)
    deg[torch.randint(0, n, (40,), generator=gen)] = torch.randint(12, 60, (40,), generator=gen)
    deg[5], deg[100], deg[101] = 0, 0, 33
    rowptr = torch.zeros(n + 1, dtype=torch.int32)
    rowptr[1:] = torch.cumsum(deg, 0)
    cols = torch.randint(0, 300, (int(rowptr[-1]),), generator=gen, dtype=torch.int32)
    classes = plan_tile_classes(rowptr, cols, lo, hi, max_len, mode)
    cap = min(max_len, 32)
    seen = []
    prev_width = 0
    for width, order, meta in classes:
        assert width in TILE_WIDTHS and width > prev_width and order.dtype == torch.int32 and meta.dtype == torch.int32
        r_tile, b_tile = tile_geometry(width)
        assert order.numel() % r_tile == 0 and meta.numel() == order.numel() // r_tile
        rows = order[order >= 0].long()
        d = deg[rows]
        assert ((rows >= lo) & (rows < hi)).all()
        assert (d <= min(width, cap)).all() and (d > min(prev_width, cap) if prev_width else d >= 0).all()
        # meta byte bt = the longest of the four rows {slot g * B + bt}
        tiles = order.view(-1, r_tile).long()
        dt = torch.where(tiles >= 0, deg[tiles.clamp(min=0)], torch.zeros_like(tiles))
        for bt in range(b_tile):
            want = dt[:, [g * b_tile + bt for g in range(4)]].amax(dim=1)
            assert torch.equal((meta.long() >> (8 * bt)) & 0xFF, want)
        # batches are ordered longest first inside a tile
        bmax = torch.stack([(meta.long() >> (8 * bt)) & 0xFF for bt in range(b_tile)], dim=1)
        assert (bmax[:, 1:] <= bmax[:, :-1]).all()
        seen.append(rows)
        prev_width = width
    seen = torch.cat(seen) if seen else torch.zeros(0, dtype=torch.long)
    want = torch.nonzero(deg[lo:hi] <= cap).flatten() + lo
    assert torch.equal(torch.sort(seen).values, want)             # each short row exactly once, nothing else


def test_cold_order_puts_rows_sharing_a_rare_column_next_to_each_other():
    from gnn_ecommerce_amd.graph import plan_tile_classes
    # 64 rows of 2 entries: column 0 is in every row, the second column is row // 4 + 1 (each used by 4 rows), shuffled
    perm = torch.randperm(64, generator=torch.Generator().manual_seed(0))
    second = (perm // 4 + 1).int()
    cols = torch.stack([torch.zeros(64, dtype=torch.int32), second], dim=1).reshape(-1)
    rowptr = (2 * torch.arange(65)).int()
    (width, order, _), = plan_tile_classes(rowptr, cols, 0, 64, 32, "cold")
    tiles = order.view(-1, 16).long()
    for t in tiles:                      # a tile of 16 rows = 4 whole groups of rows with the same rare column
        assert len(set(second[t].tolist())) == 4


def test_argument_errors_of_the_round3_entry_points():
    """lgc_segment_sum, lgc_seed_pull, lgc_seed_mark, lgc_adam_step, lgc_hop_exchange: bad arguments are refused before any launch."""
    lib = _native.load()
    one, two = ctypes.c_void_p(256), ctypes.c_void_p(512)
    seg = lambda **kw: lib.lgc_segment_sum(kw.get("key", one), one, one, None, kw.get("n", 8), 1.0, kw.get("y", two),
                                           kw.get("ys", 64), 100, kw.get("dim", 64), 0, None)
    assert seg(y=None) == -1 and seg(n=-1) == -1 and seg(dim=0) == -1 and seg(dim=300) == -1 and seg(ys=32) == -1
    assert seg(key=None) == -1 and seg(n=0) == 0                                  # nothing to do is not an error
    pull = lambda **kw: lib.lgc_seed_pull(one, one, 0, 4, 32, None, kw.get("n_chunks", 0), None, 0, None, kw.get("flag", one),
                                          one, kw.get("mark", None), one, kw.get("ss", 64), 8, kw.get("y", two), 64,
                                          kw.get("dim", 64), None)
    assert pull(dim=300) == -2 and pull(flag=None) == -1 and pull(ss=32) == -1 and pull(n_chunks=3) == -1
    assert pull(y=ctypes.c_void_p(514)) == -5 and pull(mark=one, y=ctypes.c_void_p(514)) == -5     # the marks are optional
    mark = lambda **kw: lib.lgc_seed_mark(kw.get("rp", one), one, 0, kw.get("re", 4), kw.get("rows", one), kw.get("n", 3),
                                          kw.get("mark", two), kw.get("len", 8), kw.get("value", 1), None)
    assert mark(rp=None) == -1 and mark(rows=None) == -1 and mark(mark=None) == -1 and mark(n=-1) == -1 and mark(re=-1) == -1
    assert mark(len=-1) == -1 and mark(value=256) == -1 and mark(value=-1) == -1 and mark(n=0) == 0
    adam = lambda **kw: lib.lgc_adam_step(kw.get("w", one), one, one, one, kw.get("n", 64), 0.1, 0.999, 0.001, 1e-8, 0.005,
                                          kw.get("bc2", 0.03), None)
    assert adam(w=None) == -1 and adam(n=-1) == -1 and adam(bc2=0.0) == -1 and adam(w=ctypes.c_void_p(260)) == -5
    assert adam(n=0) == 0
    rows = lambda **kw: lib.lgc_spmm_rows(one, one, 0, kw.get("re", 4), one, kw.get("n", 3), kw.get("tr", 8), one, 64,
                                          kw.get("y", two), kw.get("ys", 64), None, 0, 1.0, 0.0, kw.get("dim", 64), None)
    assert rows(dim=300) == -2 and rows(y=one) == -1 and rows(ys=32) == -1 and rows(n=-1) == -1 and rows(re=9) == -1
    assert rows(n=0) == 0
    split = lambda **kw: lib.lgc_spmm_rows_split(one, one, 0, kw.get("re", 4), one, kw.get("n", 3), kw.get("tr", 8), one, 64,
                                                 kw.get("y", two), kw.get("ys", 64), kw.get("yr", 8), None, 0, 1.0, 0.0,
                                                 kw.get("dim", 64), kw.get("compact", 0), kw.get("work", one),
                                                 kw.get("part", two), kw.get("pr", 100), None)
    assert split(dim=300) == -2 and split(y=one) == -1 and split(ys=32) == -1 and split(n=-1) == -1 and split(re=9) == -1
    assert split(yr=3) == -1 and split(compact=1, yr=2) == -1                      # y must hold every row that may be written
    assert split(work=None) == -1 and split(part=None) == -1 and split(pr=3) == -4 and split(pr=2) == -4
    assert split(n=0) == 0
    reg = lambda **kw: lib.lgc_reg_rows(kw.get("w", one), kw.get("stride", 64), kw.get("dim", 64), 100, kw.get("a", one), kw.get("m0", 4),
                                        one, 4, one, kw.get("m2", 4), 0.5, kw.get("value", two), None, kw.get("status", two), None)
    assert reg(w=None) == -1 and reg(value=None) == -1 and reg(status=None) == -1 and reg(stride=32) == -1 and reg(dim=0) == -1
    assert reg(m0=-1) == -1 and reg(a=None) == -1 and reg(m2=-1) == -1
    cb = _native.EXCHANGE_FN(lambda *a: 0)
    op = _native.OperatorC()
    assert lib.lgc_hop_exchange(ctypes.byref(op), ctypes.byref(op), 100, one, 64, two, 64, None, 0, 1.0, 0.0, 64, 90, 20, 1, cb,
                                None, None) == -1                                  # exchanged block beyond the table
    assert lib.lgc_hop_exchange(None, ctypes.byref(op), 100, one, 64, two, 64, None, 0, 1.0, 0.0, 64, 50, 20, 1, cb, None, None) == -1


def test_argument_errors_of_the_round4_entry_points():
    """lgc_seed_prepare, lgc_seed_flags, lgc_pair_dot_rows, lgc_pair_seed_vals, lgc_bpr_loss: refused before any launch."""
    lib = _native.load()
    one, two = ctypes.c_void_p(256), ctypes.c_void_p(512)
    prep = lambda **kw: lib.lgc_seed_prepare(kw.get("rows", one), kw.get("m", 8), kw.get("split", 4), kw.get("n", 10), one, one,
                                             kw.get("di", one), one, one, kw.get("flag", None), kw.get("slot", None),
                                             kw.get("scratch", two), None)
    assert prep(m=-1) == -4 and prep(m=8193) == -4 and prep(rows=None) == -1 and prep(di=None) == -1 and prep(split=11) == -1
    assert prep(flag=one) == -1 and prep(slot=one) == -1 and prep(m=0) == 0       # flag and slot: both or neither
    assert prep(scratch=None) == -1
    flags = lambda **kw: lib.lgc_seed_flags(kw.get("rows", one), kw.get("m", 8), 4, kw.get("flag", two), kw.get("value", 0), None)
    assert flags(rows=None) == -1 and flags(flag=None) == -1 and flags(m=-1) == -1 and flags(value=256) == -1 and flags(m=0) == 0
    pdr = lambda **kw: lib.lgc_pair_dot_rows(kw.get("emb", one), kw.get("stride", 64), kw.get("dim", 64), 100, kw.get("i0", one),
                                             one, kw.get("m", 8), kw.get("scores", two), None, None, None, kw.get("status", two), None)
    assert pdr(emb=None) == -1 and pdr(stride=32) == -1 and pdr(dim=0) == -1 and pdr(status=None) == -1 and pdr(i0=None) == -1
    assert pdr(scores=None) == -1 and pdr(m=-1) == -1 and pdr(m=0) == 0
    psv = lambda **kw: lib.lgc_pair_seed_vals(kw.get("g", one), None, None, kw.get("r0", one), one, kw.get("m", 8), kw.get("dim", 64),
                                              kw.get("vals", two), None)
    assert psv(g=None) == -1 and psv(r0=None) == -1 and psv(vals=None) == -1 and psv(m=-1) == -1 and psv(dim=0) == -1 and psv(m=0) == 0
    tc = lambda **kw: lib.lgc_tile_classes(kw.get("rp", one), one, 0, kw.get("re", 4), kw.get("ml", 32), 1, kw.get("tr", 8),
                                           kw.get("ws", two), kw.get("wsb", 1 << 20), kw.get("sr", two), kw.get("cc", two), None)
    assert tc(rp=None) == -1 and tc(sr=None) == -1 and tc(cc=None) == -1 and tc(re=-1) == -1 and tc(tr=2) == -1 and tc(ml=-1) == -1
    assert tc(ws=None) == -1 and tc(wsb=16) == -3 and tc(ws=ctypes.c_void_p(520)) == -5
    assert lib.lgc_tile_classes_workspace_bytes(1000, 5000) > 1000 * 20 and lib.lgc_tile_classes_workspace_bytes(-1, 5) == 0
    tp = lambda **kw: lib.lgc_tile_pack(kw.get("rp", one), kw.get("sr", one), kw.get("n", 8), kw.get("w", 8), kw.get("o", two), two, None)
    assert tp(rp=None) == -1 and tp(sr=None) == -1 and tp(o=None) == -1 and tp(n=-1) == -1 and tp(w=12) == -1 and tp(n=0) == 0
    up = lambda **kw: lib.lgc_sweep_plan_upload(kw.get("plan", None), one, one, one, one, None)
    assert up() == -1 and lib.lgc_sweep_plan_export_multi(None, one) == -1
    assert lib.lgc_bipartite_split(None, 5, two, None) == -1 and lib.lgc_bipartite_split(one, 5, None, None) == -1
    assert lib.lgc_bipartite_split(one, -1, two, None) == -1
    rpc = lambda **kw: lib.lgc_row_plan_count(kw.get("rp", one), 0, kw.get("re", 4), kw.get("sm", 32), kw.get("cl", 256),
                                              kw.get("ws", two), kw.get("wsb", 1 << 16), kw.get("tot", two), None)
    assert rpc(rp=None) == -1 and rpc(tot=None) == -1 and rpc(re=-1) == -1 and rpc(sm=-1) == -1 and rpc(cl=0) == -1
    assert rpc(ws=None) == -1 and rpc(wsb=8) == -3 and rpc(ws=ctypes.c_void_p(520)) == -5
    assert lib.lgc_row_plan_workspace_bytes(1000) >= 3 * 4004 and lib.lgc_row_plan_workspace_bytes(-1) == 0
    rpf = lambda **kw: lib.lgc_row_plan_fill(kw.get("rp", one), 0, kw.get("re", 4), 32, kw.get("cl", 256), kw.get("ws", two),
                                             kw.get("ch", two), two, None)
    assert rpf(rp=None) == -1 and rpf(ws=None) == -1 and rpf(ch=None) == -1 and rpf(cl=0) == -1 and rpf(re=0) == 0
    cfg = _native.SweepCfg(n_bands=8, waves_per_band_round=256, row_cap=78, piece_cap=64, lookahead=64, groups=4, round_order=2)
    big = _native.SweepCfg(n_bands=8, waves_per_band_round=256, row_cap=78, piece_cap=112, lookahead=64, groups=4, round_order=2)
    assert lib.lgc_sweep_dplan_workspace_bytes(1000, 10, 500, ctypes.byref(cfg)) > 1000 * 24
    assert lib.lgc_sweep_dplan_workspace_bytes(1000, 10, 500, ctypes.byref(big)) == 0 and lib.lgc_sweep_dplan_workspace_bytes(-1, 1, 1, ctypes.byref(cfg)) == 0
    code = ctypes.c_int(0)
    dp = lambda **kw: lib.lgc_sweep_dplan_create(kw.get("rp", one), one, 0, kw.get("re", 4), 0, kw.get("ne", 8), 0, kw.get("hi", 100),
                                                 ctypes.byref(kw.get("cfg", cfg)), kw.get("ws", two), kw.get("wsb", 1 << 24), None,
                                                 ctypes.byref(code))
    for kw, want in ((dict(rp=None), -1), (dict(re=-1), -1), (dict(ne=-1), -1), (dict(hi=0), -1), (dict(hi=1 << 25), -1), (dict(ws=None), -1),
                     (dict(wsb=64), -3), (dict(ws=ctypes.c_void_p(520)), -5), (dict(cfg=big), -4)):
        assert not dp(**kw) and code.value == want, (kw, code.value)
    assert lib.lgc_sweep_dplan_dims(None, ctypes.byref(_native.SweepDims())) == -1 and lib.lgc_sweep_dplan_fill(None, one, one, one, one, None) == -1
    assert lib.lgc_sweep_dplan_export_multi(None, one) == -1
    lib.lgc_sweep_dplan_free(None)
    bpr = lambda **kw: lib.lgc_bpr_loss(kw.get("s", one), None, kw.get("b", 8), kw.get("size", 8), kw.get("loss", two), kw.get("grad", two),
                                        None)
    assert bpr(loss=None) == -1 and bpr(s=None) == -1 and bpr(grad=None) == -1 and bpr(b=-1) == -1 and bpr(size=0) == -1


def test_saved_graph_validation_rejects_tampered_tensors():
    from gnn_ecommerce_amd import PropGraph
    rowptr = torch.tensor([0, 2, 3, 5], dtype=torch.int32)
    entries = torch.tensor([[1, 0], [2, 0], [0, 0], [0, 0], [1, 0]], dtype=torch.int32)
    deg = dis = torch.ones(3)
    ok = dict(rowptr=rowptr, entries=entries, deg=deg, dis=dis, num_nodes=3, num_edges=5, split=None, where="t")
    PropGraph.validate_csr(**ok)
    bad_col = entries.clone(); bad_col[3, 0] = 3
    neg_col = entries.clone(); neg_col[0, 0] = -1
    for change in (dict(rowptr=torch.tensor([0, 3, 2, 5], dtype=torch.int32)),         # not monotone
                   dict(rowptr=torch.tensor([0, 2, 3, 4], dtype=torch.int32)),         # does not end at num_edges
                   dict(rowptr=torch.tensor([1, 2, 3, 5], dtype=torch.int32)),         # does not start at 0
                   dict(rowptr=rowptr[:-1]), dict(rowptr=rowptr.long()),
                   dict(entries=bad_col), dict(entries=neg_col), dict(entries=entries[:4]),
                   dict(deg=torch.ones(2)), dict(dis=torch.ones(3, dtype=torch.float64)), dict(split=3), dict(split=0)):
        with pytest.raises(ValueError):
            PropGraph.validate_csr(**{**ok, **change})


def test_saved_graph_validation_checks_the_stored_split_against_the_entries():
    """a stale `split` would make bipartite_sum return wrong tables silently."""
    from gnn_ecommerce_amd import PropGraph
    # users 0, 1 | items 2, 3: user rows hold item columns and the reverse
    rowptr = torch.tensor([0, 2, 3, 5, 6], dtype=torch.int32)
    entries = torch.tensor([[2, 0], [3, 0], [2, 0], [0, 0], [1, 0], [0, 0]], dtype=torch.int32)
    deg = dis = torch.ones(4)
    ok = dict(rowptr=rowptr, entries=entries, deg=deg, dis=dis, num_nodes=4, num_edges=6, split=2, where="t")
    PropGraph.validate_csr(**ok)
    PropGraph.validate_csr(**{**ok, "split": None})
    for split in (1, 3):                                   # the stored split does not separate rows from columns
        with pytest.raises(ValueError, match="split"):
            PropGraph.validate_csr(**{**ok, "split": split})
    user_to_user = entries.clone(); user_to_user[0, 0] = 1
    with pytest.raises(ValueError, match="split"):
        PropGraph.validate_csr(**{**ok, "entries": user_to_user})


def test_purchase_lists_are_validated_before_the_kernel_indexes_them():
    """lgc_mask_topk reads list_ptr[u], list_ptr[u + 1], list_items[e] without bounds."""
    from gnn_ecommerce_amd.propagate import SeenLists
    ptr, items = torch.tensor([0, 2, 2, 5]), torch.tensor([4, 1, 0, 3, 2])
    assert SeenLists(ptr, items).validate(3).ptr is ptr
    for bad_ptr, bad_items in ((ptr[:-1], items),                        # shorter than n_users + 1 (truncated file)
                               (torch.tensor([0, 3, 2, 5]), items),       # not monotone
                               (torch.tensor([1, 2, 2, 5]), items),       # does not start at 0
                               (ptr, items[:4]),                          # points past the item list
                               (ptr.int(), items), (ptr, items.int()), (ptr, items.view(5, 1))):
        with pytest.raises(ValueError):
            SeenLists(bad_ptr, bad_items).validate(3)


# ----------------------------------------------------------------------------------------------
# band-sweep planner (host code of the library: runs without a GPU)
# ----------------------------------------------------------------------------------------------
def _decode_sweep(dims, arr, row_cap):
    """slot -> [(col, value bits)] in list order, checking the step invariants on the way."""
    groups = dims["groups"]
    slabs = arr["slabs"].numpy().view(np.uint32).reshape(-1, 16 * groups, 4)
    wsp, wnp = arr["wave_slab_ptr"].numpy(), arr["wave_npieces"].numpy()
    ps = arr["piece_slot"].numpy().reshape(-1, row_cap)
    out = {}
    for w in range(dims["n_waves"]):
        assert 0 <= wnp[w] <= row_cap
        for sb in range(wsp[w], wsp[w + 1]):
            for s in range(32):
                used = set()
                for grp in range(groups):
                    x, v = int(slabs[sb, 16 * grp + (s >> 1), 2 * (s & 1)]), int(slabs[sb, 16 * grp + (s >> 1), 2 * (s & 1) + 1])
                    col, pc = x & 0xFFFFFF, x >> 24
                    if col == 0xFFFFFF:                                  # padding: dummy accumulator, value 0
                        assert pc == row_cap and v == 0
                        continue
                    assert pc < wnp[w] and pc not in used                 # no two entries of a step share a piece
                    used.add(pc)
                    out.setdefault(int(ps[w, pc]), []).append((col, v))
    return out


@pytest.mark.parametrize("cfg", [dict(n_bands=8, waves_per_band_round=16, row_cap=39, piece_cap=64, lookahead=32),
                                 dict(n_bands=3, waves_per_band_round=4, row_cap=7, piece_cap=5, lookahead=4),
                                 dict(n_bands=1, waves_per_band_round=8, row_cap=200, piece_cap=1000, lookahead=64),
                                 dict(n_bands=4, waves_per_band_round=64, row_cap=78, piece_cap=64, lookahead=64),
                                 dict(n_bands=8, waves_per_band_round=16, row_cap=51, piece_cap=64, lookahead=64, groups=2),
                                 dict(n_bands=8, waves_per_band_round=4, row_cap=3, piece_cap=64, lookahead=32, round_order=1),
                                 dict(n_bands=3, waves_per_band_round=4, row_cap=7, piece_cap=5, lookahead=4, round_order=1),
                                 dict(n_bands=8, waves_per_band_round=4, row_cap=3, piece_cap=64, lookahead=32, round_order=2)])
def test_sweep_plan_holds_every_entry_once_in_conflict_free_steps(cfg):
    from gnn_ecommerce_amd.graph import SWEEP_CFG, sweep_plan_host
    cfg = dict(cfg, round_order=cfg.get("round_order", SWEEP_CFG["round_order"]))     # the planner's default applies
    g = synth.make_bipartite(3000, 120, 26000, seed=4)
    ei, ew = g.coo()
    n, nu = g.num_nodes, g.n_users
    dst = ei[1]
    order = torch.argsort(dst, stable=True)
    rowptr = torch.zeros(n + 1, dtype=torch.int32)
    rowptr[1:] = torch.cumsum(torch.bincount(dst, minlength=n), 0)
    entries = torch.stack([ei[0][order].int(), ew[order].view(torch.int32)], dim=1).contiguous()
    dims, arr = sweep_plan_host(rowptr, entries, nu, n, 0, nu, cfg)
    assert dims["n_entries"] == 26000 and dims["n_rows"] == 120 and dims["n_bands"] == cfg["n_bands"]
    assert dims["n_waves"] == dims["rounds"] * cfg["waves_per_band_round"] * cfg["n_bands"]
    assert cfg["piece_cap"] <= dims["piece_cap"] < 4 * cfg["piece_cap"] + 16
    got = _decode_sweep(dims, arr, cfg["row_cap"])
    multi = arr["multi"].numpy()
    assert multi.shape == (120, 4) and multi[0, 1] == 0 and multi[-1, 2] == dims["n_slots"]
    assert (multi[1:, 1] == multi[:-1, 2]).all() and (multi[:, 0] == np.arange(nu, n)).all()      # slots contiguous per row
    total = 0
    for row, sb, se, _ in multi:
        mine = [e for sl in range(sb, se) for e in got.get(sl, [])]
        want = [(int(c), int(v) & 0xFFFFFFFF) for c, v in entries[rowptr[row]:rowptr[row + 1]].tolist()]
        assert sorted(mine) == sorted(want)
        prev_hi = -1
        for sl in range(sb, se):                     # inside a slot: ascending columns; slots of a row: ascending ranges
            cols = [c for c, _ in got.get(sl, [])]
            assert cols and len(cols) <= dims["piece_cap"]
            if cfg.get("round_order") == 2:        # odd rounds walk downwards: a slot's list is sorted one way or the other
                assert cols == sorted(cols) or cols == sorted(cols, reverse=True)
                cols = sorted(cols)
            assert cols == sorted(cols) and cols[0] >= prev_hi
            prev_hi = cols[-1]
        total += len(mine)
    assert total == 26000
    assert dims["groups"] == (cfg.get("groups") or 4)
    assert dims["n_padding"] == dims["groups"] * dims["n_steps"] - 26000
    if cfg.get("round_order") in (1, 2) and dims["rounds"] > 1:
        # rounds by weight: inside a band no piece of a later round is longer than a piece of an earlier round
        wnp, ps = arr["wave_npieces"].numpy(), arr["piece_slot"].numpy().reshape(dims["n_waves"], cfg["row_cap"])
        nb, wpbr = cfg["n_bands"], cfg["waves_per_band_round"]
        for b in range(nb):
            lo_prev = None
            for r in range(dims["rounds"]):
                lens = [len(got[int(ps[w, k])]) for w in range(dims["n_waves"])
                        if (w // 4) % nb == b and (w // 4) // nb // (wpbr // 4) == r for k in range(wnp[w])]
                if not lens:
                    continue
                assert lo_prev is None or max(lens) <= lo_prev
                lo_prev = min(lens)


def test_sweep_plan_on_random_small_operators():
    """Property test (hypothesis): any CSR slice -- empty rows, rows living in one band, duplicate columns, a single
    column, more bands than columns -- and any legal planner configuration: every entry lands in exactly one slot of
    its own row, steps are conflict-free, slots of a row are contiguous."""
    from hypothesis import given, settings, strategies as st, HealthCheck
    from gnn_ecommerce_amd.graph import sweep_plan_host

    @st.composite
    def cases(draw):
        n_rows = draw(st.integers(1, 12))
        n_cols = draw(st.integers(1, 40))
        lens = draw(st.lists(st.integers(0, 30), min_size=n_rows, max_size=n_rows))
        cols = [draw(st.lists(st.integers(0, n_cols - 1), min_size=l, max_size=l)) for l in lens]
        cfg = dict(n_bands=draw(st.sampled_from([1, 2, 3, 8])), waves_per_band_round=draw(st.sampled_from([4, 8])),
                   row_cap=draw(st.sampled_from([1, 2, 5, 78])), piece_cap=draw(st.sampled_from([1, 3, 64])),
                   lookahead=draw(st.sampled_from([4, 64])),
                   groups=draw(st.sampled_from([2, 4])), round_order=draw(st.sampled_from([0, 1, 2])))
        return n_cols, cols, cfg

    @settings(max_examples=60, deadline=None, suppress_health_check=list(HealthCheck))
    @given(cases())
    def check(case):
        n_cols, cols, cfg = case
        n_rows = len(cols)
        rowptr = torch.zeros(n_rows + 1, dtype=torch.int32)
        rowptr[1:] = torch.cumsum(torch.tensor([len(c) for c in cols]), 0)
        flat = [c for row in cols for c in row]
        ne = len(flat)
        entries = torch.zeros((max(ne, 1), 2), dtype=torch.int32)
        if ne:
            entries[:ne, 0] = torch.tensor(flat, dtype=torch.int32)
            entries[:ne, 1] = torch.arange(1, ne + 1, dtype=torch.int32)          # value bits identify the entry
        dims, arr = sweep_plan_host(rowptr, entries[:ne] if ne else entries[:0], 0, n_rows, 0, n_cols, cfg)
        assert dims["n_entries"] == ne and dims["n_rows"] == n_rows
        got = _decode_sweep(dims, arr, cfg["row_cap"])
        multi = arr["multi"].numpy()[:n_rows]
        seen = 0
        for row, sb, se, _ in multi:
            mine = sorted(e for sl in range(sb, se) for e in got.get(sl, []))
            want = sorted((int(c), int(v)) for c, v in entries[rowptr[row]:rowptr[row + 1]].tolist())
            assert mine == want
            seen += len(mine)
        assert seen == ne and sum(len(v) for v in got.values()) == ne
        assert dims["n_padding"] == dims["groups"] * dims["n_steps"] - ne
    check()


def test_sweep_plan_argument_errors():
    import ctypes as ct
    lib = _native.load()
    rowptr = torch.tensor([0, 2, 3], dtype=torch.int32)
    entries = torch.tensor([[1, 0], [5, 0], [2, 0]], dtype=torch.int32)
    code = ct.c_int(0)

    def create(lo=0, hi=8, **kw):
        cfg = _native.SweepCfg(**{**dict(n_bands=8, waves_per_band_round=16, row_cap=39, piece_cap=64, lookahead=32,
                                         groups=4), **kw})
        h = lib.lgc_sweep_plan_create(rowptr.data_ptr(), entries.data_ptr(), 0, 2, lo, hi, ct.byref(cfg), ct.byref(code))
        if h:
            lib.lgc_sweep_plan_free(h)
        return bool(h), code.value
    assert create() == (True, 0)
    assert create(hi=4) == (False, -1)                     # column 5 outside [0, 4)
    assert create(waves_per_band_round=6)[0] is False and create(row_cap=0)[0] is False and create(row_cap=255)[0] is False
    assert create(n_bands=0)[0] is False and create(lookahead=2)[0] is False and create(piece_cap=0)[0] is False
    assert create(groups=3)[0] is False and create(groups=2)[0] is True
    assert create(round_order=3)[0] is False and create(round_order=1)[0] is True and create(round_order=2)[0] is True
    assert lib.lgc_sweep_ok(64, 1_693_929, 64) == 4 and lib.lgc_sweep_ok(61, 1000, 64) == 4      # entries per step
    assert lib.lgc_sweep_ok(90, 1000, 96) == 2 and lib.lgc_sweep_ok(80, 1000, 80) == 2 and lib.lgc_sweep_ok(96, 10, 96) == 2
    assert lib.lgc_sweep_ok(100, 1000, 100) == 4 and lib.lgc_sweep_ok(128, 1000, 128) == 4    # the 4-entry plan, two passes
    assert lib.lgc_sweep_ok(129, 1000, 129) == 0 and lib.lgc_sweep_ok(66, 1000, 66) == 0
    assert lib.lgc_sweep_ok(64, 1 << 24, 64) == 0 and lib.lgc_sweep_ok(64, 1000, 32) == 0
    one = ct.c_void_p(256)
    assert lib.lgc_spmm_sweep(one, one, one, one, 6, 78, 4, one, 4, None, 0, one, 1000, one, 64, ct.c_void_p(512), 64, None, 0,
                              1.0, 0.0, 64, None) == -1                              # waves not a multiple of 4
    assert lib.lgc_spmm_sweep(one, one, one, one, 8, 78, 4, one, 4, None, 0, one, 1000, one, 64, ct.c_void_p(512), 64, None, 0,
                              1.0, 0.0, 90, None) == -2                              # a 4-entry plan on a 90-wide table


@pytest.mark.gpu
def test_plain_c_host_drives_the_abi(tmp_path):
    """tests/c_host/hop_host.c: no Python, no torch -- hipMalloc'd buffers, lgc_build_csr, lgc_spmm, lgc_build_tiles +
    lgc_spmm_tiles from C, checked in C against the oracle's C restatement."""
    import subprocess
    if not torch.cuda.is_available():
        pytest.skip("no ROCm device")
    exe = str(tmp_path / "hop_host")
    lib_dir = os.path.join(ROOT, "gnn-ecommerce_amd", "csrc")
    ora_dir = os.path.join(ROOT, "oracle")
    if not os.path.isfile(os.path.join(ora_dir, "liblgconv_ref.so")):
        subprocess.run(["make", "-C", ora_dir], check=True)
    cmd = ["gcc", "-O1", os.path.join(ROOT, "tests", "c_host", "hop_host.c"), "-I" + os.path.join(ROOT, "include"),
           "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", os.path.join(lib_dir, "liblgconv_hip.so"),
           os.path.join(ora_dir, "liblgconv_ref.so"), "-L/opt/rocm/lib", "-lamdhip64", "-lm",
           "-Wl,-rpath," + lib_dir, "-Wl,-rpath," + ora_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    build = subprocess.run(cmd, capture_output=True, text=True)
    assert build.returncode == 0, build.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and run.stdout.startswith("ok "), (run.returncode, run.stdout, run.stderr)


def test_listed_rows_guard_follows_the_degree_moments():
    """Operator.listed_rows_pay (host logic, no launch): listed rows for the last item step of a scoring forward only while
    a popularity-drawn batch's expected work stays below LISTED_ROWS_MAX_SHARE of the half's entries."""
    from gnn_ecommerce_amd.graph import Operator
    flat = torch.arange(0, 1001, dtype=torch.int32) * 100                     # 1000 rows of 100 entries
    op = Operator.build(1000, flat, torch.zeros((100_000, 2), dtype=torch.int32), 0, 1000)
    assert op.listed_rows_pay(800) and not op.listed_rows_pay(808)            # n / 4 * (100 + 100) <= 0.4 * 100,000
    hub = torch.cat([torch.arange(0, 999, dtype=torch.int32) * 50, torch.tensor([49_950, 100_000], dtype=torch.int32)])
    heavy = Operator.build(1000, hub, torch.zeros((100_000, 2), dtype=torch.int32), 0, 1000)   # one row holds half of all entries
    assert heavy.listed_rows_pay(4) and not heavy.listed_rows_pay(8)         # n / 4 * (25,075 + 100) <= 40,000
    empty = Operator.build(4, torch.zeros(5, dtype=torch.int32), torch.zeros((0, 2), dtype=torch.int32), 0, 4)
    assert empty.listed_rows_pay(10 ** 6)
