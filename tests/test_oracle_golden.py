"""The CPU oracle against the fixtures produced by the reference's own code (tests/golden/make_golden.py).

These pin everything the reference itself implements on the path -- layer sum, pair scoring, BPR,
regulariser, label layout, COO layout, Adam step order, recommendK.  The one-layer LGConv arithmetic
inside the fixtures is our restatement (PyG is absent): parity at THAT boundary is unpinned and is
cross-checked here only against an independent fp64 CSR evaluation and the scalar C restatement.
Tolerances are 1e-6 (not bit equality) because torch's vectorised CPU kernels differ by ISA."""
import ctypes
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, golden_names, load_golden, rel_fro, t
from oracle import lightgcn_oracle as oracle
from tests_support import hub_inputs

RTOL = 1e-6


@pytest.mark.parametrize("name", golden_names("train_"))
def test_train_fixture(name):
    z = load_golden(name)
    ei, ew, w0, alpha = t(z["edge_index"]), t(z["edge_weight"]), t(z["weight0"]), t(z["alpha"])
    layers = len(alpha) - 1
    n_users, n_items = int(z["n_users"]), int(z["n_items"])
    # graph layout: [[u | i], [i | u]], weights [w | w]  (src/utils_v2.py:146-165)
    half = ei.size(1) // 2
    ei2, ew2 = oracle.pairs_to_graph(ei[0, :half], ei[1, :half], ew[:half])
    assert torch.equal(ei2, ei) and torch.equal(ew2, ew)
    assert ei[0, :half].max() < n_users <= ei[1, :half].min()
    assert torch.equal(alpha, oracle.default_alpha(layers))
    emb = oracle.get_embedding(w0, alpha, ei, ew, layers)
    assert rel_fro(emb, t(z["embedding"])) <= RTOL
    users, pos, neg = t(z["users"]), t(z["pos"]), t(z["neg"])
    assert torch.equal(oracle.batch_pos_neg_edges(users, pos, neg), t(z["labels"]))
    w = w0.clone().requires_grad_(True)
    scores, bpr, reg, loss = oracle.train_step_loss(w, alpha, ei, ew, users, pos, neg, layers, z["decay"].item())
    loss.backward()
    assert torch.allclose(scores.detach(), t(z["scores"]), rtol=1e-5, atol=1e-9)
    assert abs(bpr.item() - z["bpr"].item()) <= RTOL * abs(z["bpr"].item())
    assert abs(reg.item() - z["reg"].item()) <= RTOL * abs(z["reg"].item())
    assert rel_fro(w.grad, t(z["grad"])) <= 1e-5
    # three Adam steps, src/train_lightgcn.py:130-147 order
    p = torch.nn.Parameter(w0.clone())
    opt = torch.optim.Adam([p], z["lr"].item())
    for step in range(3):
        opt.zero_grad()
        oracle.train_step_loss(p, alpha, ei, ew, users, pos, neg, layers, z["decay"].item())[3].backward()
        opt.step()
        if step == 0:
            assert rel_fro(p.detach(), t(z["weight_after_1"])) <= RTOL
    assert rel_fro(p.detach(), t(z["weight_after_3"])) <= 1e-5
    top = oracle.recommend_topk(oracle.get_embedding(w0, alpha, ei, ew, layers), n_users, n_items,
                                t(z["rec_seen"]), z["rec_users"].tolist(), 5)
    assert np.array_equal(top.numpy(), z["rec_topk"])          # index work: exact


@pytest.mark.parametrize("name", golden_names("edge_"))
def test_edge_case_fixture(name):
    z = load_golden(name)
    ei, w0, alpha = t(z["edge_index"]), t(z["weight0"]), t(z["alpha"])
    ew = t(z["edge_weight"]) if "edge_weight" in z else None
    labels = t(z["labels"]) if "labels" in z else None
    norm = bool(z["normalize"])
    emb = oracle.get_embedding(w0, alpha, ei, ew, len(alpha) - 1, norm)
    want = t(z["embedding"])
    assert torch.equal(emb.isnan(), want.isnan())
    ok = ~want.isnan().any(dim=1)
    assert rel_fro(emb[ok], want[ok]) <= RTOL
    scores = oracle.forward(w0, alpha, ei, labels, ew, len(alpha) - 1, norm)
    ws = t(z["scores"])
    fin = ~ws.isnan()
    assert torch.equal(scores.isnan(), ws.isnan())
    assert torch.allclose(scores[fin], ws[fin], rtol=1e-5, atol=1e-9)
    if name == "edge_isolated":
        assert torch.equal(emb[-2:], w0[-2:] * alpha[0])        # untouched nodes keep only the layer-0 term


def test_hub_fixture():
    z = load_golden("hub_s3")
    import hashlib
    ei, ew, w0 = hub_inputs(int(z["seed"]), int(z["n_users"]), int(z["n_items"]), int(z["dim"]))
    if hashlib.sha256(ei.numpy().tobytes() + ew.numpy().tobytes()).hexdigest().encode() != z["input_sha256"].tobytes():
        pytest.skip("numpy Generator stream differs from the one that produced the fixture")
    layers = int(z["layers"])
    assert int(z["hub_degree"]) >= 10_000
    emb = oracle.get_embedding(w0, oracle.default_alpha(layers), ei, ew, layers)
    rows = t(z["rows"])
    assert rel_fro(emb[rows], t(z["embedding_rows"])) <= RTOL
    assert torch.allclose(emb.norm(dim=1), t(z["row_l2"]), rtol=1e-5)


def test_lgconv_against_independent_fp64_csr():
    """The unpinned boundary: fp32 gather/scale/index_add vs an fp64 SpMM on the same fp32 values."""
    from gnn_ecommerce_amd import synth
    g = synth.make_bipartite(3000, 500, 30000, seed=1)
    ei, ew = g.coo()
    x = synth.xavier_table(g.num_nodes, 64, 0)
    val = oracle.gcn_norm(ei, ew, g.num_nodes)
    y = oracle.lgconv(x, ei, ew)
    y64 = oracle.lgconv_fp64(x, ei, val)
    assert rel_fro(y, y64) <= 5e-7
    # normalisation facts the restatement relies on
    deg = torch.zeros(g.num_nodes).scatter_add_(0, ei[1], ew)
    seq = np.zeros(g.num_nodes, dtype=np.float32)
    for d, w in zip(ei[1, :2000].tolist(), ew[:2000].tolist()):   # a prefix, sequentially in numpy fp32
        seq[d] = np.float32(seq[d] + np.float32(w))
    deg_prefix = torch.zeros(g.num_nodes).scatter_add_(0, ei[1, :2000], ew[:2000])
    assert np.array_equal(deg_prefix.numpy(), seq), "CPU scatter_add_ must be the sequential fp32 sum"
    assert torch.allclose(val, deg.pow(-0.5)[ei[0]] * ew * deg.pow(-0.5)[ei[1]], rtol=0, atol=0)


def test_c_restatement_matches_torch_oracle():
    """oracle/lgconv_ref.c (scalar C, sequential) vs the torch oracle: bit-exact hop given identical values."""
    so = os.path.join(ROOT, "oracle", "liblgconv_ref.so")
    if not os.path.isfile(so):
        pytest.skip("oracle/liblgconv_ref.so not built (run __graft_entry__.build())")
    lib = ctypes.CDLL(so)
    from gnn_ecommerce_amd import synth
    g = synth.make_bipartite(800, 150, 6000, seed=2)
    ei, ew = g.coo()
    n, dim = g.num_nodes, 90
    x = synth.xavier_table(n, dim, 3)
    ei_c = ei.contiguous()
    deg = torch.empty(n)
    val = torch.empty(ei.size(1))
    y = torch.empty(n, dim)
    vp = ctypes.c_void_p
    lib.lgconv_ref_norm.argtypes = [vp, vp, ctypes.c_int64, ctypes.c_int64, vp, vp]
    lib.lgconv_ref_hop.argtypes = [vp, vp, ctypes.c_int64, ctypes.c_int64, vp, ctypes.c_int64, vp]
    lib.lgconv_ref_norm(ei_c.data_ptr(), ew.data_ptr(), n, ei.size(1), deg.data_ptr(), val.data_ptr())
    assert torch.equal(deg, torch.zeros(n).scatter_add_(0, ei[1], ew))
    tv = oracle.gcn_norm(ei, ew, n)
    assert (val.view(torch.int32).long() - tv.view(torch.int32).long()).abs().max() <= 4   # rsqrt ulp, see DESIGN.md
    lib.lgconv_ref_hop(ei_c.data_ptr(), tv.data_ptr(), n, ei.size(1), x.data_ptr(), dim, y.data_ptr())
    assert torch.equal(y, oracle.lgconv(x, ei, tv, normalize=False))
