#!/usr/bin/env python3
"""Fixtures for the two callers either side of the hot path, captured from the reference's OWN code
(run once in the build container: PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_serve.py):

  serve_ref.npz   torchserve/lightgcn_handler.py driven through initialize -> preprocess -> inference -> postprocess
                  with a stub context, on a model_dir this script writes (processed_train.csv + a checkpoint in the
                  layout of src/utils_v2.py:214-232).  `ts` (TorchServe) is not installed: its BaseHandler is supplied
                  as an empty base class; the model file is the handler's own torchserve/lightgcn.py, whose PyG import
                  is met by the oracle's LGConv exactly as in make_golden.py.  Stored: the CSV columns, the weights, the
                  edge list the handler built (it subtracts n_users from the item ids BEFORE df_to_graph,
                  lightgcn_handler.py:36-38, so item ids alias user ids -- kept, the fixture is what upstream serves),
                  the requests and the responses.
  ingest_ref.npz  src/utils_v2.py relabelling -> item offset -> df_to_graph / interact_matrix on raw ids (the
                  training-side ingest, train_lightgcn.py:16-37).
Only arrays are stored; no reference source travels."""
import os
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

import numpy as np
import pandas as pd
import torch

from make_golden import WEIGHT_SET, save, small_pairs          # noqa: E402  (also installs nothing by itself)
from oracle import lightgcn_oracle as oracle                   # noqa: E402


def shim_pyg():
    ts = types.ModuleType("torch_sparse")

    class SparseTensor:
        pass

    ts.SparseTensor = SparseTensor
    tg, tg_nn = types.ModuleType("torch_geometric"), types.ModuleType("torch_geometric.nn")
    tg_conv, tg_typing = types.ModuleType("torch_geometric.nn.conv"), types.ModuleType("torch_geometric.typing")
    tg_conv.LGConv = oracle.OracleLGConv
    tg_typing.Adj, tg_typing.OptTensor = torch.Tensor, "Optional[Tensor]"
    tg.nn, tg_nn.conv, tg.typing = tg_nn, tg_conv, tg_typing
    sys.modules.update({"torch_sparse": ts, "torch_geometric": tg, "torch_geometric.nn": tg_nn,
                        "torch_geometric.nn.conv": tg_conv, "torch_geometric.typing": tg_typing})


def serve_fixture():
    shim_pyg()
    ts_pkg, ts_th, ts_bh = types.ModuleType("ts"), types.ModuleType("ts.torch_handler"), types.ModuleType("ts.torch_handler.base_handler")

    class BaseHandler:                                  # TorchServe's base class is not installed; nothing of it is used
        pass

    ts_bh.BaseHandler = BaseHandler
    sys.modules.update({"ts": ts_pkg, "ts.torch_handler": ts_th, "ts.torch_handler.base_handler": ts_bh})
    sys.path.insert(0, "/root/reference/torchserve")
    import lightgcn_handler as ref_handler                 # imports torchserve/lightgcn.py as `lightgcn`
    import lightgcn as ref_model

    rng = np.random.default_rng(20260)
    n_users, n_items, dim, layers = 150, 60, 64, 3
    u, i, w = small_pairs(rng, n_users, n_items, 900)
    df = pd.DataFrame({"user_id": u + 100000, "item_id": i + 5000, "weight": w, "user_id_idx": u, "item_id_idx": i + n_users})
    torch.manual_seed(7)
    model = ref_model.LightGCN(n_users + n_items, dim, layers)
    weight0 = model.embedding.weight.detach().clone()
    with tempfile.TemporaryDirectory() as d:
        df.to_csv(os.path.join(d, "processed_train.csv"))
        torch.save({"timestamp": "2026-01-01 00:00:00", "epoch": 1, "model_state_dict": model.state_dict(),
                    "optimizer_state_dict": {}, "precision": 0.0, "recall": 0.0,
                    "hyperparams": {"latent_dim": dim, "n_layers": layers}}, os.path.join(d, "model.pt"))
        ctx = types.SimpleNamespace(manifest={"model": {"serializedFile": "model.pt"}},
                                    system_properties={"model_dir": d, "gpu_id": 0})
        h = ref_handler.LightGCNHandler()
        h.initialize(ctx)
        requests = [[3], [17, 42, 3, 149], list(range(0, 150, 11))]
        responses = []
        for req in requests:
            out = h.postprocess(h.inference(h.preprocess([{"body": req}])))
            assert isinstance(out, list) and len(out) == 1 and list(out[0]) == ["items"]
            responses.append(np.array(out[0]["items"], dtype=np.int64))
        seen = h.i_m_matrix.coalesce()
    arrays = dict(user_id=df["user_id"].values, item_id=df["item_id"].values, weight=df["weight"].values,
                  user_id_idx=df["user_id_idx"].values, item_id_idx=df["item_id_idx"].values,
                  n_users=n_users, n_items=n_items, dim=dim, layers=layers, weight0=weight0, alpha=model.alpha,
                  edge_index=h.edge_index, edge_weight=h.edge_weight, seen_indices=seen.indices(), k=20)
    for r, (req, resp) in enumerate(zip(requests, responses)):
        arrays[f"request{r}"] = np.array(req, dtype=np.int64)
        arrays[f"response{r}"] = resp
    arrays["n_requests"] = len(requests)
    save("serve_ref", **arrays)


def ingest_fixture():
    sys.path.insert(0, "/root/reference/src")
    import utils_v2 as utils
    rng = np.random.default_rng(77)
    n = 1200
    raw = pd.DataFrame({"user_id": rng.choice(np.arange(500000, 500400) * 3, size=n),
                        "item_id": rng.choice(np.arange(9000, 9100) * 7, size=n),
                        "weight": WEIGHT_SET[rng.integers(len(WEIGHT_SET), size=n)]})
    raw = raw.drop_duplicates(["user_id", "item_id"]).reset_index(drop=True)
    df = raw.copy()
    n_users, n_items, df, _, _ = utils.relabelling(df)
    seen = utils.interact_matrix(df, n_users, n_items).coalesce()
    df["item_id_idx"] = df["item_id_idx"] + n_users                       # utils_v2.py:128
    edge_index, edge_weight = utils.df_to_graph(df, True)
    save("ingest_ref", user_id=raw["user_id"].values, item_id=raw["item_id"].values, weight=raw["weight"].values,
         n_users=n_users, n_items=n_items, user_id_idx=df["user_id_idx"].values, item_id_idx=df["item_id_idx"].values,
         edge_index=edge_index, edge_weight=edge_weight, seen_indices=seen.indices())


if __name__ == "__main__":
    ingest_fixture()
    serve_fixture()
