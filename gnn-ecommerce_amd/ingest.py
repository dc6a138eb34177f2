"""Interaction table -> relabelled ids -> propagation graph, in one pass, plus the persisted graph a serving worker
loads.

Upstream does this with pandas + scikit-learn on every start: ``relabelling`` (LabelEncoder on the raw user and item
ids, src/utils_v2.py:40-61), the item offset (``item_id_idx + n_users``, :128), ``df_to_graph`` (:146-165) and
``interact_matrix`` (:100-112); the TorchServe handler re-reads the processed CSV and rebuilds all of it per worker
(torchserve/lightgcn_handler.py:32-38).  Here the id maps are two ``numpy.unique`` calls (the same sorted-classes
encoding as LabelEncoder), the COO goes straight to the device in the reference's layout, and
``save_serving_graph`` writes ONE flat safetensors file -- CSR, degrees, id maps and the purchased-items lists --
that ``PropGraph.load`` + ``serving.RecommendHandler`` start from without touching a CSV.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch
from torch import Tensor


@dataclass
class Interactions:
    """A relabelled interaction table (host side)."""
    n_users: int
    n_items: int
    user_ids: np.ndarray        # raw id of user index k (sorted, as LabelEncoder.classes_)
    item_ids: np.ndarray        # raw id of item index k
    user_idx: np.ndarray        # int64 [P], row order of the input
    item_idx: np.ndarray        # int64 [P], NOT offset by n_users
    weight: np.ndarray          # fp32 [P]

    def coo(self, device="cpu"):
        """edge_index int64 [2, 2P], edge_weight fp32 [2P] exactly as src/utils_v2.py:146-165 lays them out
        ([[u | i + n_users], [i + n_users | u]], weights [w | w])."""
        u = torch.from_numpy(self.user_idx)
        i = torch.from_numpy(self.item_idx + self.n_users)
        w = torch.from_numpy(self.weight)
        return torch.stack((torch.cat([u, i]), torch.cat([i, u]))).to(device), torch.cat([w, w]).to(device)

    def seen_csr(self):
        """Purchases (weight == 1.0, src/utils_v2.py:104) as a CSR over users: (ptr int64 [n_users + 1], items int64),
        items of a user ascending and DE-DUPLICATED.

        Intended difference from upstream, parity unpinned: ``interact_matrix`` (src/utils_v2.py:100-112) keeps a
        (user, item) pair bought twice as two entries of the sparse ``interactions_t``; ``to_dense()`` sums them to 2
        and ``pred * (1 - 2)`` then scores that item ``-pred`` instead of 0 (torchserve/lightgcn_handler.py:88,
        src/lightgcn.py:175).  A list of seen items has no multiplicity: a purchased item is masked to 0 however often
        it was bought.  On a frame with one row per (user, item) the two agree; the fixtures captured from the
        reference have no duplicates (tests/golden/make_golden_serve.py drops them before the capture) and
        tests/test_ingest_serving.py::test_duplicate_purchases_are_masked_once pins what happens here."""
        buy = self.weight == np.float32(1.0)
        key = np.unique(self.user_idx[buy] * self.n_items + self.item_idx[buy])
        u, i = key // self.n_items, key % self.n_items
        ptr = np.zeros(self.n_users + 1, dtype=np.int64)
        np.cumsum(np.bincount(u, minlength=self.n_users), out=ptr[1:])
        return ptr, i.astype(np.int64)


def relabel(user_id, item_id, weight) -> Interactions:
    """Raw ids -> contiguous indices in sorted order of the raw ids (src/utils_v2.py:48-52: LabelEncoder)."""
    user_id, item_id = np.asarray(user_id), np.asarray(item_id)
    weight = np.asarray(weight, dtype=np.float32)
    if not (len(user_id) == len(item_id) == len(weight)):
        raise ValueError("user_id, item_id and weight must have the same length")
    user_ids, user_idx = np.unique(user_id, return_inverse=True)
    item_ids, item_idx = np.unique(item_id, return_inverse=True)
    return Interactions(len(user_ids), len(item_ids), user_ids, item_ids, user_idx.astype(np.int64).reshape(-1),
                        item_idx.astype(np.int64).reshape(-1), weight)


def read_interactions_csv(path: str, user_col: str = "user_id", item_col: str = "item_id", weight_col: str = "weight",
                          chunksize: int = 4_000_000) -> Interactions:
    """The preprocessed interaction CSV (train_lightgcn.py:16) read column-wise in chunks: only the three columns
    are parsed, nothing else of the frame is materialised."""
    import pandas as pd
    cols = {user_col: [], item_col: [], weight_col: []}
    for chunk in pd.read_csv(path, usecols=list(cols), chunksize=chunksize):
        for c in cols:
            cols[c].append(chunk[c].to_numpy())
    return relabel(*(np.concatenate(cols[c]) if cols[c] else np.zeros(0) for c in (user_col, item_col, weight_col)))


def build_graph(inter: Interactions, device, normalize: bool = True):
    """The device-resident propagation graph of an interaction table (COO -> CSR by lgc_build_csr)."""
    from .graph import PropGraph
    ei, ew = inter.coo(device)
    return PropGraph(ei, ew, inter.n_users + inter.n_items, normalize)


def save_serving_graph(path: str, inter: Interactions, graph=None, device=None) -> None:
    """One file for a serving worker: the built graph + id maps + purchased-items CSR."""
    if graph is None:
        graph = build_graph(inter, device)
    ptr, items = inter.seen_csr()
    for name, ids in (("user", inter.user_ids), ("item", inter.item_ids)):
        if not np.issubdtype(np.asarray(ids).dtype, np.integer):
            # LabelEncoder accepts strings; a tensor file does not: keep such maps beside the file
            raise TypeError(f"{name} ids of dtype {np.asarray(ids).dtype} cannot be stored in the graph file: the id maps "
                            "are int64 tensors -- relabel string ids to integers first and keep that map outside")
    graph.save(path, extra={"user_ids": torch.from_numpy(np.asarray(inter.user_ids, dtype=np.int64)),
                            "item_ids": torch.from_numpy(np.asarray(inter.item_ids, dtype=np.int64)),
                            "seen_ptr": torch.from_numpy(ptr), "seen_items": torch.from_numpy(items)},
               meta={"n_users": str(inter.n_users), "n_items": str(inter.n_items)})
