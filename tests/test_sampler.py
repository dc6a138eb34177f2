"""Device mini-batch sampler vs the contract of batch_loader (src/utils_v2.py:168-181).
RNG streams cannot match Python's ``random``; the tests check constraints exactly and distributions
statistically against the oracle restatement."""
import random

import numpy as np
import pytest
import torch

from oracle import lightgcn_oracle as oracle
import gnn_ecommerce_amd as lg
from gnn_ecommerce_amd.sampler import TripleSampler, lists_to_csr

from conftest import load_golden
from tests_support import sampler_lists

N_USERS, N_ITEMS = 60, 25


def toy_lists(seed=0):
    order, pos, ign = sampler_lists(N_USERS, N_ITEMS, seed)
    return pos, ign


def test_oracle_batch_loader_reproduces_the_reference_under_the_same_seed():
    """tests/golden/sampler_ref.npz was produced by the reference's own batch_loader under random.seed."""
    z = load_golden("sampler_ref")
    order, pos, ign = sampler_lists(int(z["n_users"]), int(z["n_items"]), int(z["lists_seed"]))
    for seed in (0, 1, 2):
        rng = random.Random(seed)
        got = torch.stack([torch.stack(oracle.batch_loader(order, pos, ign, int(z["batch"]), int(z["n_users"]),
                                                           int(z["n_items"]), rng)) for _ in range(3)])
        assert torch.equal(got, torch.from_numpy(z[f"seed{seed}"]))            # index work: exact


def test_lists_to_csr_layout():
    ptr, flat = lists_to_csr(5, {0: [9, 7, 7], 3: [8]}, sort_unique=False)
    assert ptr.tolist() == [0, 3, 3, 3, 4, 4] and flat.tolist() == [9, 7, 7, 8] and ptr.dtype == np.int32
    ptr, flat = lists_to_csr(5, {0: [9, 7, 7], 3: [8]}, sort_unique=True)
    assert ptr.tolist() == [0, 2, 2, 2, 3, 3] and flat.tolist() == [7, 9, 8]
    with pytest.raises(IndexError):
        lists_to_csr(2, {5: [1]}, sort_unique=False)


def test_pairs_to_csr_matches_lists_to_csr():
    from gnn_ecommerce_amd.sampler import pairs_to_csr
    pos, ign = toy_lists(2)
    pu = np.concatenate([[u] * len(v) for u, v in pos.items()]); pi = np.concatenate([v for v in pos.values()])
    iu = np.concatenate([[u] * len(v) for u, v in ign.items()]); ii = np.concatenate([list(v) for v in ign.values()])
    perm = np.random.default_rng(0).permutation(len(iu))
    a_ptr, a_items = lists_to_csr(N_USERS, ign, sort_unique=True)
    b_ptr, b_items = pairs_to_csr(N_USERS, iu[perm], ii[perm], sort_unique=True)
    assert np.array_equal(a_ptr, b_ptr) and np.array_equal(a_items, b_items)
    a_ptr, a_items = lists_to_csr(N_USERS, pos, sort_unique=False)
    b_ptr, b_items = pairs_to_csr(N_USERS, pu, pi, sort_unique=False)
    assert np.array_equal(a_ptr, b_ptr) and np.array_equal(a_items, b_items)


def test_sampler_refuses_cpu():
    pos, ign = toy_lists()
    with pytest.raises(lg._native.NativeLibraryError):
        TripleSampler(N_USERS, N_ITEMS, pos, ign, "cpu")


def test_oracle_batch_loader_contract():
    pos, ign = toy_lists()
    u, p, n = oracle.batch_loader(sorted(pos), pos, ign, 20, N_USERS, N_ITEMS, random.Random(0))
    assert u.dtype == p.dtype == n.dtype == torch.int64 and len(set(u.tolist())) == 20
    for uu, pp, nn in zip(u.tolist(), p.tolist(), n.tolist()):
        assert pp in pos[uu] and nn not in ign[uu] and N_USERS <= nn < N_USERS + N_ITEMS


@pytest.mark.gpu
def test_constraints_and_determinism(device):
    pos, ign = toy_lists()
    s = TripleSampler(N_USERS, N_ITEMS, pos, ign, device, seed=3)
    batches = [s.sample(32) for _ in range(50)]
    s.check()
    for u, p, n in batches:
        assert u.dtype == p.dtype == n.dtype == torch.int64 and u.is_cuda and u.shape == p.shape == n.shape == (32,)
        ul, pl, nl = u.tolist(), p.tolist(), n.tolist()
        assert len(set(ul)) == 32 and all(uu in pos for uu in ul)                    # without replacement
        for uu, pp, nn in zip(ul, pl, nl):
            assert pp in pos[uu] and nn not in ign[uu] and N_USERS <= nn < N_USERS + N_ITEMS
            if uu == 5:
                assert nn == N_USERS + 9
    s2 = TripleSampler(N_USERS, N_ITEMS, pos, ign, device, seed=3)
    again = [s2.sample(32) for _ in range(50)]
    assert all(torch.equal(a[i], b[i]) for a, b in zip(batches, again) for i in range(3))    # seeded
    assert not torch.equal(batches[0][2], batches[1][2])                                      # steps differ
    with pytest.raises(ValueError):
        s.sample(N_USERS)                                                                     # > users with positives


@pytest.mark.gpu
def test_distribution_matches_oracle(device):
    """Per-user frequencies of the sampled positive and negative items against the reference semantics."""
    pos, ign = toy_lists(1)
    s = TripleSampler(N_USERS, N_ITEMS, pos, ign, device, seed=0)
    rng = random.Random(1)
    rounds, batch = 1500, 16
    dev_neg = np.zeros((N_USERS, N_ITEMS)); ref_neg = np.zeros((N_USERS, N_ITEMS))
    dev_pos = np.zeros((N_USERS, N_ITEMS)); ref_pos = np.zeros((N_USERS, N_ITEMS))
    dev_u = np.zeros(N_USERS); ref_u = np.zeros(N_USERS)
    for _ in range(rounds):
        u, p, n = (t.cpu().numpy() for t in s.sample(batch))
        np.add.at(dev_neg, (u, n - N_USERS), 1); np.add.at(dev_pos, (u, p - N_USERS), 1); np.add.at(dev_u, u, 1)
        u, p, n = (t.numpy() for t in oracle.batch_loader(sorted(pos), pos, ign, batch, N_USERS, N_ITEMS, rng))
        np.add.at(ref_neg, (u, n - N_USERS), 1); np.add.at(ref_pos, (u, p - N_USERS), 1); np.add.at(ref_u, u, 1)
    s.check()
    # users: uniform over the users with positives
    cand = np.array(sorted(pos))
    expect = rounds * batch / len(cand)
    assert np.all(dev_u[np.setdiff1d(np.arange(N_USERS), cand)] == 0)
    assert np.abs(dev_u[cand] - expect).max() < 6 * np.sqrt(expect)
    # negatives: uniform over the admissible items of each user; positives: proportional to list multiplicity
    for uu in cand[:20]:
        adm = np.array([i for i in range(N_ITEMS) if i + N_USERS not in ign[uu]])
        tot = dev_neg[uu].sum()
        assert dev_neg[uu][np.setdiff1d(np.arange(N_ITEMS), adm)].sum() == 0
        assert np.abs(dev_neg[uu][adm] - tot / len(adm)).max() < 6 * np.sqrt(tot / len(adm)) + 3
        mult = np.bincount(np.array(pos[uu]) - N_USERS, minlength=N_ITEMS) / len(pos[uu])
        assert np.abs(dev_pos[uu] / max(dev_pos[uu].sum(), 1) - mult).max() < 0.12
    # and the two samplers agree with each other within sampling noise
    assert np.abs(dev_neg.sum(0) / dev_neg.sum() - ref_neg.sum(0) / ref_neg.sum()).max() < 0.01


@pytest.mark.gpu
def test_status_flags(device):
    pos, ign = {0: [N_USERS]}, {0: [i + N_USERS for i in range(N_ITEMS)]}      # every item ignored
    s = TripleSampler(N_USERS, N_ITEMS, pos, ign, device)
    s.sample(1)
    with pytest.raises(RuntimeError):
        s.check()
