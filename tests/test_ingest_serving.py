"""The callers either side of the hot path (SURVEY.md 8a a-S, 8f N4) against fixtures captured from the reference's own
code (tests/golden/make_golden_serve.py): the training-side ingest (relabelling -> df_to_graph -> interact_matrix) and the
TorchServe handler's request -> response path."""
import os
import types

import numpy as np
import pytest
import torch

from conftest import load_golden, t
from oracle import lightgcn_oracle as oracle
from tests_support import assert_topk_exact_up_to_ties

import gnn_ecommerce_amd as lg
from gnn_ecommerce_amd import ingest


def test_relabel_and_layout_equal_the_references():
    z = load_golden("ingest_ref")
    it = ingest.relabel(z["user_id"], z["item_id"], z["weight"])
    assert it.n_users == int(z["n_users"]) and it.n_items == int(z["n_items"])
    assert np.array_equal(it.user_idx, z["user_id_idx"]) and np.array_equal(it.item_idx + it.n_users, z["item_id_idx"])
    assert np.array_equal(it.user_ids[it.user_idx], z["user_id"]) and np.array_equal(it.item_ids[it.item_idx], z["item_id"])
    ei, ew = it.coo()
    assert ei.dtype == torch.int64 and ew.dtype == torch.float32
    assert np.array_equal(ei.numpy(), z["edge_index"]) and np.array_equal(ew.numpy(), z["edge_weight"])
    ptr, items = it.seen_csr()
    rows = np.repeat(np.arange(it.n_users), np.diff(ptr))
    assert np.array_equal(rows, z["seen_indices"][0]) and np.array_equal(items, z["seen_indices"][1])


def test_duplicate_purchases_are_masked_once():
    """the list form of the seen-mask has no multiplicity (documented difference from upstream's
    interact_matrix, which would score a twice-bought item -pred)."""
    inter = ingest.relabel([10, 10, 10, 20], [7, 7, 8, 7], [1.0, 1.0, 0.5, 1.0])
    ptr, items = inter.seen_csr()
    assert ptr.tolist() == [0, 1, 2] and items.tolist() == [0, 0]      # user 0 bought item 0 twice: listed once


def test_string_ids_are_refused_with_a_clear_error(tmp_path):
    inter = ingest.relabel(["a", "b"], ["x", "y"], [1.0, 1.0])
    with pytest.raises(TypeError, match="id maps"):
        ingest.save_serving_graph(str(tmp_path / "g.safetensors"), inter, graph=object())


def test_csv_reader_parses_only_the_three_columns(tmp_path):
    import pandas as pd
    z = load_golden("ingest_ref")
    path = str(tmp_path / "interactions.csv")
    pd.DataFrame({"user_id": z["user_id"], "junk": "x", "item_id": z["item_id"], "weight": z["weight"]}).to_csv(path)
    it = ingest.read_interactions_csv(path, chunksize=97)
    assert np.array_equal(it.user_idx, z["user_id_idx"]) and np.array_equal(it.weight, z["weight"])
    with pytest.raises(ValueError):
        ingest.relabel([1, 2], [3], [1.0, 1.0])


def _reference_masked_scores(z, request):
    """masked scores of the reference handler's own call, from the oracle on the handler's own edge list"""
    ei, ew = t(z["edge_index"]), t(z["edge_weight"])
    emb = oracle.get_embedding(t(z["weight0"]), t(z["alpha"]), ei, ew, int(z["layers"]))
    nu, ni = int(z["n_users"]), int(z["n_items"])
    seen = torch.zeros(nu, ni)
    seen[z["seen_indices"][0], z["seen_indices"][1]] = 1.0
    users, items = torch.split(emb, [nu, ni])
    return (users[request] @ items.t()) * (1 - seen[request])


@pytest.mark.gpu
def test_drop_in_model_answers_the_reference_handlers_requests(device):
    """The reference handler's own arguments to recommendK (its edge list -- item ids aliased onto user ids and all,
    lightgcn_handler.py:36-38 -- its sparse-row mask, its user list, k = 20) given to the drop-in model."""
    z = load_golden("serve_ref")
    nu, ni = int(z["n_users"]), int(z["n_items"])
    model = lg.LightGCN(nu + ni, int(z["dim"]), int(z["layers"]))
    model.load_state_dict({"alpha": t(z["alpha"]), "embedding.weight": t(z["weight0"])})
    model.to(device).eval()
    ei, ew = t(z["edge_index"]).to(device), t(z["edge_weight"]).to(device)
    seen = torch.sparse_coo_tensor(t(z["seen_indices"]), torch.ones(z["seen_indices"].shape[1]), (nu, ni))
    for r in range(int(z["n_requests"])):
        req = z[f"request{r}"].tolist()
        inter = torch.index_select(seen, 0, torch.as_tensor(req)).to_dense()           # lightgcn_handler.py:88
        with torch.no_grad():
            frame = model.recommendK(ei, ew, nu, ni, inter, req, int(z["k"]))
        got = {"items": list(frame["top_rlvnt_itm"])}                                  # :94
        assert_topk_exact_up_to_ties(np.array(got["items"]), z[f"response{r}"], _reference_masked_scores(z, req).numpy())


@pytest.mark.gpu
def test_handler_counterpart_serves_from_the_persisted_graph(device, tmp_path):
    """serving.RecommendHandler with a stub TorchServe context: initialize from model_dir (checkpoint in the layout of
    src/utils_v2.py:214-232 + graph.safetensors), then preprocess -> inference -> postprocess per request; same
    responses as the reference handler gave (fixture), response type and nesting included."""
    from gnn_ecommerce_amd import serving
    from gnn_ecommerce_amd.graph import PropGraph
    z = load_golden("serve_ref")
    nu, ni = int(z["n_users"]), int(z["n_items"])
    d = str(tmp_path)
    graph = PropGraph(t(z["edge_index"]).to(device), t(z["edge_weight"]).to(device), nu + ni)   # what upstream serves
    si = z["seen_indices"]
    ptr = np.zeros(nu + 1, dtype=np.int64)
    np.cumsum(np.bincount(si[0], minlength=nu), out=ptr[1:])
    graph.save(os.path.join(d, serving.GRAPH_FILE), extra={"seen_ptr": torch.from_numpy(ptr), "seen_items": t(si[1])},
               meta={"n_users": nu, "n_items": ni})
    torch.save({"timestamp": "x", "epoch": 1, "model_state_dict": {"alpha": t(z["alpha"]), "embedding.weight": t(z["weight0"])},
                "optimizer_state_dict": {}, "precision": 0.0, "recall": 0.0,
                "hyperparams": {"latent_dim": int(z["dim"]), "n_layers": int(z["layers"])}}, os.path.join(d, "model.pt"))
    ctx = types.SimpleNamespace(manifest={"model": {"serializedFile": "model.pt"}},
                                system_properties={"model_dir": d, "gpu_id": 0})
    h = serving.RecommendHandler()
    h.initialize(ctx)
    assert h.initialized and h.k == 20
    for r in range(int(z["n_requests"])):
        req = z[f"request{r}"].tolist()
        out = h.handle([{"body": req}])
        assert isinstance(out, list) and len(out) == 1 and list(out[0]) == ["items"]
        items = out[0]["items"]
        assert isinstance(items, list) and len(items) == len(req) and all(isinstance(x, list) and len(x) == 20 for x in items)
        assert_topk_exact_up_to_ties(np.array(items), z[f"response{r}"], _reference_masked_scores(z, req).numpy())
    assert h.preprocess([{"data": [1, 2]}]) == [1, 2]                                   # "data" wins over "body"
    with pytest.raises(RuntimeError, match="Missing the model.pt file"):
        bad = types.SimpleNamespace(manifest={"model": {"serializedFile": "nope.pt"}}, system_properties=ctx.system_properties)
        serving.RecommendHandler().initialize(bad)


@pytest.mark.gpu
def test_ingest_to_serving_end_to_end(device, tmp_path):
    """CSV of raw ids -> ingest -> persisted serving graph -> handler: the graph equals the reference's training-side
    graph (fixture), seen items come back masked to 0 exactly as upstream masks them."""
    from gnn_ecommerce_amd import serving
    from gnn_ecommerce_amd.graph import PropGraph
    z = load_golden("ingest_ref")
    it = ingest.relabel(z["user_id"], z["item_id"], z["weight"])
    d = str(tmp_path)
    ingest.save_serving_graph(os.path.join(d, serving.GRAPH_FILE), it, device=device)
    g, extra, meta = PropGraph.load(os.path.join(d, serving.GRAPH_FILE), device, with_extra=True)
    ref = PropGraph(t(z["edge_index"]).to(device), t(z["edge_weight"]).to(device), it.n_users + it.n_items)
    assert torch.equal(g.forward_op.rowptr, ref.forward_op.rowptr) and torch.equal(g.forward_op.entries, ref.forward_op.entries)
    assert g.split == it.n_users and int(meta["n_users"]) == it.n_users
    assert np.array_equal(extra["user_ids"].numpy(), it.user_ids) and np.array_equal(extra["item_ids"].numpy(), it.item_ids)
    model = lg.LightGCN(it.n_users + it.n_items, 64, 2)
    torch.save({"model_state_dict": model.state_dict(), "hyperparams": {"latent_dim": 64, "n_layers": 2}},
               os.path.join(d, "m.pt"))
    h = serving.RecommendHandler()
    h.initialize(types.SimpleNamespace(manifest={"model": {"serializedFile": "m.pt"}},
                                       system_properties={"model_dir": d, "gpu_id": None}))
    users = [0, 5, it.n_users - 1]
    out = h.handle([{"body": users}])[0]["items"]
    rows = h.seen_rows(torch.as_tensor(users, device=device)).cpu()
    ptr, items = it.seen_csr()
    for k, u in enumerate(users):
        assert set(torch.nonzero(rows[k]).flatten().tolist()) == set(items[ptr[u]:ptr[u + 1]].tolist())
    emb = oracle.get_embedding(model.embedding.weight.detach(), model.alpha, t(z["edge_index"]), t(z["edge_weight"]), 2)
    want = oracle.recommend_topk(emb, it.n_users, it.n_items, rows, users, 20)
    ue, ie = torch.split(emb, [it.n_users, it.n_items])
    assert_topk_exact_up_to_ties(np.array(out), want.numpy(), ((ue[users] @ ie.t()) * (1 - rows)).numpy())
