"""Input generators shared by tests/golden/make_golden.py and the tests (no reference code here)."""
import numpy as np
import torch

WEIGHT_SET = np.array([0.01, 0.02, 0.03, 0.1, 0.11, 0.5, 1.0], dtype=np.float32)


def formula_weight(n, dim):
    """Exactly representable fp32 table, no RNG."""
    r = np.arange(n, dtype=np.int64)[:, None]
    c = np.arange(dim, dtype=np.int64)[None, :]
    return (((r * 131 + c * 17 + (r * c) % 29) % 257) - 128).astype(np.float32) / 1024.0


def hub_pairs(seed, n_users, n_items):
    """(user, item, weight): item 0 is linked to ~90 % of the users (degree > 1e4 at n_users=12000)."""
    rng = np.random.default_rng(seed)
    u_hub = np.arange(n_users)[rng.random(n_users) < 0.9]
    rest_u = rng.integers(n_users, size=20000)
    rest_i = rng.integers(1, n_items, size=20000)
    keys = np.unique(np.concatenate([u_hub * n_items, rest_u * n_items + rest_i]))
    keys = np.concatenate([keys, np.setdiff1d(np.arange(n_users) * n_items + 1, keys)])  # cover all users
    rng.shuffle(keys)
    u, i = keys // n_items, keys % n_items
    w = WEIGHT_SET[rng.integers(len(WEIGHT_SET), size=len(keys))]
    return u, i, w


def hub_inputs(seed, n_users, n_items, dim):
    """edge_index / edge_weight in the reference's COO layout (src/utils_v2.py:146-165) + formula table."""
    u, i, w = hub_pairs(seed, n_users, n_items)
    u_t, i_t, w_t = torch.from_numpy(u), torch.from_numpy(i + n_users), torch.from_numpy(w)
    ei = torch.stack((torch.cat([u_t, i_t]), torch.cat([i_t, u_t])))
    ew = torch.cat([w_t, w_t])
    return ei, ew, torch.from_numpy(formula_weight(n_users + n_items, dim))


def sampler_lists(n_users, n_items, seed):
    """(user order, positives per user, ignore list per user) in the shape of the reference's train_pos_list_df:
    users without purchases are absent; item ids are offset by n_users; ignore lists contain the positives."""
    rng = np.random.default_rng(seed)
    pos, ign = {}, {}
    for u in range(n_users):
        if u % 7 == 3:
            continue
        k = int(rng.integers(1, 5))
        p = (rng.integers(0, n_items, k) + n_users).tolist()
        extra = (rng.integers(0, n_items, int(rng.integers(0, 12))) + n_users).tolist()
        pos[u], ign[u] = p, sorted(set(p) | set(extra))
    ign[5] = [i + n_users for i in range(n_items) if i != 9]        # a single admissible negative
    pos[5] = [n_users + 1]
    order = sorted(pos)
    np.random.default_rng(seed + 100).shuffle(order)
    return order, pos, ign


def assert_topk_exact_up_to_ties(got, want, ref_masked_scores, rel_gap=1e-6):
    """Index work is exact: ``got[r, p] == want[r, p]`` everywhere, except where the REFERENCE's own masked
    scores of the two candidates differ by less than ``rel_gap`` relative to the row's largest |score| (a provable
    near-tie: an fp32 reordering of the sum can swap them).  ``ref_masked_scores`` is [rows, n_items]."""
    got, want = np.asarray(got), np.asarray(want)
    s = np.asarray(ref_masked_scores, dtype=np.float64)
    assert got.shape == want.shape and s.shape[0] == got.shape[0]
    bad = np.argwhere(got != want)
    for r, p in bad:
        scale = np.abs(s[r]).max()
        gap = abs(s[r, got[r, p]] - s[r, want[r, p]])
        assert gap <= rel_gap * scale, (
            f"row {r} position {p}: got item {got[r, p]} (ref score {s[r, got[r, p]]:.9g}), want {want[r, p]} "
            f"({s[r, want[r, p]]:.9g}); gap {gap:.3g} is not a tie at scale {scale:.3g}")
    # whatever was swapped, each row must still hold k distinct items
    assert all(len(set(row)) == len(row) for row in got.tolist())
    return len(bad)
