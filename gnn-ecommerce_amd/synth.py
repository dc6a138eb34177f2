"""Synthetic user-item interaction graphs shaped like the cosmetics-shop dataset.

The real CSVs are DVC pointers to an unreachable remote, so every input is generated:
heavy-tailed activity on both sides (Pareto 2.0 users / 1.5 items), every node with at least
one edge (the reference relabels only ids present in the training frame, src/utils_v2.py:48-60),
an exact number of unique (user, item) pairs, weights from the event-weight rules of
notebooks/1.data_preprocessing.ipynb / config.yaml:10-11 (all in (0, 1], ~13 % equal to 1.0),
pairs shuffled, and the COO laid out as src/utils_v2.py:146-165 lays it out:
``[[u | i + n_users], [i + n_users | u]]`` with weights ``[w | w]``.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Tuple

import numpy as np
import torch

WEIGHT_VALUES = np.array([0.01, 0.02, 0.03, 0.1, 0.11, 0.5, 1.0], dtype=np.float32)
WEIGHT_PROBS = np.array([0.40, 0.12, 0.05, 0.17, 0.08, 0.05, 0.13])

# BASELINE.json configs
CONFIG_SMALL = dict(n_users=10_000, n_items=2_000, n_pairs=120_000)            # configs[0]
CONFIG_COSMETICS = dict(n_users=1_639_358, n_items=54_571, n_pairs=10_157_408)  # configs[1..4]


@dataclass
class BipartiteGraph:
    n_users: int
    n_items: int
    user: np.ndarray      # int64 [P]  user index
    item: np.ndarray      # int64 [P]  item index (NOT yet offset)
    weight: np.ndarray    # fp32  [P]

    @property
    def num_nodes(self) -> int:
        return self.n_users + self.n_items

    @property
    def nnz(self) -> int:
        return 2 * len(self.user)

    def coo(self, device="cpu") -> Tuple[torch.Tensor, torch.Tensor]:
        """edge_index int64 [2, 2P], edge_weight fp32 [2P] in the reference's layout."""
        u = torch.from_numpy(self.user)
        i = torch.from_numpy(self.item) + self.n_users
        w = torch.from_numpy(self.weight)
        ei = torch.stack((torch.cat([u, i]), torch.cat([i, u])))
        ew = torch.cat([w, w])
        return ei.to(device), ew.to(device)


def _sampler(rng: np.random.Generator, propensity: np.ndarray):
    cdf = np.cumsum(propensity / propensity.sum())
    cdf[-1] = 1.0

    def draw(n: int) -> np.ndarray:
        return np.searchsorted(cdf, rng.random(n), side="right").astype(np.int64)

    return draw


def make_bipartite(n_users: int, n_items: int, n_pairs: int, seed: int = 0) -> BipartiteGraph:
    if n_pairs < max(n_users, n_items) or n_pairs > n_users * n_items:
        raise ValueError("n_pairs must cover every node and fit the bipartite grid")
    rng = np.random.default_rng(seed)
    # capped Pareto propensities: a few hub items with ~10 % of all users, users up to a few thousand items
    pu = np.minimum(rng.pareto(2.0, n_users) + 1.0, 1200.0)
    pi = np.minimum(rng.pareto(1.5, n_items) + 1.0, 4000.0)
    draw_u, draw_i = _sampler(rng, pu), _sampler(rng, pi)
    # coverage first: one edge per user and per item
    cover = np.concatenate([np.arange(n_users, dtype=np.int64) * n_items + draw_i(n_users),
                            draw_u(n_items) * n_items + np.arange(n_items, dtype=np.int64)])
    keys = np.unique(cover)
    n_cover = len(keys)
    while len(keys) < n_pairs:
        need = n_pairs - len(keys)
        batch = int(need * 1.15) + 1024
        keys = np.unique(np.concatenate([keys, draw_u(batch) * n_items + draw_i(batch)]))
    if len(keys) > n_pairs:
        # drop surplus non-coverage pairs (coverage pairs must stay so that no node is left isolated)
        is_cover = np.isin(keys, cover, assume_unique=False)
        extra = np.flatnonzero(~is_cover)
        drop = rng.choice(extra, size=len(keys) - n_pairs, replace=False)
        keep = np.ones(len(keys), dtype=bool)
        keep[drop] = False
        keys = keys[keep]
    assert len(keys) == n_pairs and n_cover <= n_pairs
    rng.shuffle(keys)
    user, item = keys // n_items, keys % n_items
    weight = WEIGHT_VALUES[rng.choice(len(WEIGHT_VALUES), size=n_pairs, p=WEIGHT_PROBS)]
    return BipartiteGraph(n_users, n_items, user, item, weight)


def xavier_table(num_nodes: int, dim: int, seed: int, device="cpu") -> torch.Tensor:
    """Initial embedding table exactly as src/lightgcn.py:81,87 creates it (Embedding + xavier_uniform_)."""
    g = torch.Generator().manual_seed(seed)
    w = torch.empty(num_nodes, dim)
    bound = (6.0 / (num_nodes + dim)) ** 0.5
    w.uniform_(-bound, bound, generator=g)
    return w.to(device)


def algorithmic_bytes_per_layer(num_nodes: int, nnz: int, dim: int) -> int:
    """B_min of SURVEY.md 8d: every matrix byte once (int32 col + fp32 val per edge, int32 row
    pointer), x read once, y written once."""
    return nnz * 8 + (num_nodes + 1) * 4 + 2 * num_nodes * dim * 4
