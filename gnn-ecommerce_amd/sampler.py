"""Mini-batch (user, positive, negative) sampler on the device -- the contract of ``batch_loader``
(src/utils_v2.py:168-181, called once per step at src/train_lightgcn.py:132).

Upstream merges two DataFrames and runs a Python rejection loop per row for every batch; after the
propagation itself takes ~2 ms that loop would be the training step.  Here the per-user lists become two
CSRs once (``TripleSampler.from_frame`` takes the same ``train_pos_list_df`` the reference builds:
columns ``user_id_idx``, ``item_id_idx_list``, ``ignor_neg_list``), the users of a batch are a random
subset without replacement of the users that have positives (``random.sample``, :174), and one HIP launch
draws a positive (uniform over the list entries, :178) and a negative (uniform over the items, rejecting
the ignore set, :169-173,179) for each of them.  Same return value: three int64 tensors
``(users, pos_items, neg_items)``, item ids already offset by ``n_users`` -- on the device, so the
``.to(device)`` copies of train_lightgcn.py:133-135 become no-ops.

Random streams cannot match Python's ``random``; what is tested is the distribution and the constraints.
"""
from __future__ import annotations

from typing import Dict, Iterable, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor

from . import _native


def lists_to_csr(n_users: int, lists: Dict[int, Sequence[int]], sort_unique: bool) -> Tuple[np.ndarray, np.ndarray]:
    """{user: items} -> (ptr int32 [n_users+1], items int64).  Host-side, once per dataset."""
    counts = np.zeros(n_users, dtype=np.int64)
    prepared = {}
    for u, items in lists.items():
        if not 0 <= int(u) < n_users:
            raise IndexError(f"user id {u} outside [0, {n_users})")
        a = np.asarray(list(items), dtype=np.int64)
        if sort_unique:
            a = np.unique(a)
        prepared[int(u)] = a
        counts[int(u)] = a.size
    ptr = np.zeros(n_users + 1, dtype=np.int64)
    np.cumsum(counts, out=ptr[1:])
    if ptr[-1] >= 2 ** 31:
        raise OverflowError("per-user lists do not fit int32 offsets")
    flat = np.empty(int(ptr[-1]), dtype=np.int64)
    for u, a in prepared.items():
        flat[ptr[u]:ptr[u + 1]] = a
    return ptr.astype(np.int32), flat


def pairs_to_csr(n_users: int, user: np.ndarray, item: np.ndarray, sort_unique: bool) -> Tuple[np.ndarray, np.ndarray]:
    """(user, item) pair arrays -> (ptr int32 [n_users+1], items int64), vectorised."""
    user, item = np.asarray(user, dtype=np.int64), np.asarray(item, dtype=np.int64)
    if user.size and (user.min() < 0 or user.max() >= n_users):
        raise IndexError(f"user ids outside [0, {n_users})")
    if sort_unique:
        keys = np.unique(user * (item.max() + 1 if item.size else 1) + item)
        base = item.max() + 1 if item.size else 1
        user, item = keys // base, keys % base
    else:
        order = np.argsort(user, kind="stable")
        user, item = user[order], item[order]
    ptr = np.zeros(n_users + 1, dtype=np.int64)
    np.cumsum(np.bincount(user, minlength=n_users), out=ptr[1:])
    if ptr[-1] >= 2 ** 31:
        raise OverflowError("per-user lists do not fit int32 offsets")
    return ptr.astype(np.int32), item


class TripleSampler:
    def __init__(self, n_users: int, n_items: int, pos_lists: Dict[int, Sequence[int]],
                 ignore_lists: Dict[int, Iterable[int]], device, seed: int = 0, _csr=None):
        self.n_users, self.n_items, self.device = int(n_users), int(n_items), torch.device(device)
        if self.device.type != "cuda":
            raise _native.NativeLibraryError("the sampler runs on a ROCm device only (no CPU fallback)")
        if _csr is not None:
            pos_ptr, pos_items, ign_ptr, ign_items = _csr
        else:
            pos_ptr, pos_items = lists_to_csr(n_users, pos_lists, sort_unique=False)
            ign_ptr, ign_items = lists_to_csr(n_users, ignore_lists, sort_unique=True)
        if pos_items.size and (pos_items.min() < n_users or pos_items.max() >= n_users + n_items):
            raise IndexError("positive item ids must already be offset by n_users (src/utils_v2.py:128)")
        self.candidates = torch.from_numpy(np.flatnonzero(np.diff(pos_ptr) > 0).astype(np.int64)).to(self.device)
        self.pos_ptr, self.pos_items = (torch.from_numpy(a).to(self.device) for a in (pos_ptr, pos_items))
        self.ign_ptr, self.ign_items = (torch.from_numpy(a).to(self.device) for a in (ign_ptr, ign_items))
        self.status = torch.zeros(4, dtype=torch.int32, device=self.device)
        self.seed, self.step = int(seed), 0
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(int(seed))

    @classmethod
    def from_frame(cls, train_pos_list_df, n_users: int, n_items: int, device, seed: int = 0) -> "TripleSampler":
        """From the frame ``ignor_neg_item_list`` returns (src/utils_v2.py:76-89)."""
        users = train_pos_list_df["user_id_idx"].tolist()
        pos = dict(zip(users, train_pos_list_df["item_id_idx_list"].tolist()))
        ign = dict(zip(users, train_pos_list_df["ignor_neg_list"].tolist()))
        return cls(n_users, n_items, pos, ign, device, seed)

    @classmethod
    def from_pairs(cls, n_users: int, n_items: int, pos_user, pos_item, ign_user, ign_item, device,
                   seed: int = 0) -> "TripleSampler":
        """From flat (user, item) arrays: positives (one row per list entry) and ignore pairs; item ids
        already offset by n_users.  Vectorised -- for datasets where per-user Python lists are too slow."""
        csr = (*pairs_to_csr(n_users, pos_user, pos_item, sort_unique=False),
               *pairs_to_csr(n_users, ign_user, ign_item, sort_unique=True))
        return cls(n_users, n_items, {}, {}, device, seed, _csr=csr)

    def sample(self, batch_size: int) -> Tuple[Tensor, Tensor, Tensor]:
        """(users, pos_items, neg_items), each int64 [batch_size] on the device."""
        n_cand = self.candidates.numel()
        if batch_size > n_cand:
            raise ValueError("Sample larger than population or is negative")       # random.sample's message
        pick = torch.randperm(n_cand, device=self.device, generator=self._gen)[:batch_size]
        users = self.candidates[pick].contiguous()
        pos = torch.empty(batch_size, dtype=torch.int64, device=self.device)
        neg = torch.empty(batch_size, dtype=torch.int64, device=self.device)
        lib = _native.load()
        with torch.cuda.device(self.device):
            code = lib.lgc_sample_triples(
                _native.ptr(users), batch_size, _native.ptr(self.pos_ptr), _native.ptr(self.pos_items),
                _native.ptr(self.ign_ptr), _native.ptr(self.ign_items), self.n_users, self.n_items,
                self.seed, self.step, _native.ptr(pos), _native.ptr(neg), _native.ptr(self.status),
                _native.stream_of(self.device))
        _native.check(code, "lgc_sample_triples")
        self.step += 1
        return users, pos, neg

    def check(self) -> None:
        """Synchronising check of the kernel's status word."""
        st = int(self.status[0].item())
        self.status.zero_()
        if st & _native.ST_INDEX_OOB:
            raise IndexError("sampler was given a user without positives or outside [0, n_users)")
        if st & _native.ST_SAMPLER_EXHAUSTED:
            raise RuntimeError("no admissible negative item found for some user (ignore set covers the catalogue)")
