#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE'S OWN CODE.

Run in the build container only (the reference tree does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What executes:  ``/root/reference/src/lightgcn.py`` (LightGCN, BPRLoss) and
``/root/reference/src/utils_v2.py`` (df_to_graph, batch_pos_neg_edges,
regularization_loss) are imported unmodified.  ``lightgcn.py`` imports three names
from packages that are not installed here and that the reference does not vendor
(``torch_sparse.SparseTensor``, ``torch_geometric.typing.{Adj,OptTensor}``,
``torch_geometric.nn.conv.LGConv``); they are supplied through ``sys.modules``.  The
first two are type annotations only.  ``LGConv`` is ``oracle.OracleLGConv`` -- our
restatement of PyG's documented layer -- so the arithmetic of ONE layer is NOT pinned by
these fixtures ("parity unpinned" at that boundary, see oracle/lightgcn_oracle.py);
everything around it (layer sum, scoring, BPR, regulariser, label/graph layout, Adam
steps, recommendK) is the reference's code producing the numbers.

Only data is written: inputs and expected outputs as ``.npz`` (no pickles, no source).
"""
import hashlib
import json
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import pandas as pd
import torch

from oracle import lightgcn_oracle as oracle

REF_SRC = "/root/reference/src"


def import_reference():
    ts = types.ModuleType("torch_sparse")

    class SparseTensor:  # annotation / isinstance target only (src/lightgcn.py:116)
        pass

    ts.SparseTensor = SparseTensor
    tg = types.ModuleType("torch_geometric")
    tg_nn = types.ModuleType("torch_geometric.nn")
    tg_conv = types.ModuleType("torch_geometric.nn.conv")
    tg_conv.LGConv = oracle.OracleLGConv
    tg_typing = types.ModuleType("torch_geometric.typing")
    tg_typing.Adj = torch.Tensor
    tg_typing.OptTensor = "Optional[Tensor]"
    tg.nn, tg_nn.conv, tg.typing = tg_nn, tg_conv, tg_typing
    sys.modules.update({"torch_sparse": ts, "torch_geometric": tg, "torch_geometric.nn": tg_nn,
                        "torch_geometric.nn.conv": tg_conv, "torch_geometric.typing": tg_typing})
    sys.path.insert(0, REF_SRC)
    import lightgcn as ref_lightgcn
    import utils_v2 as ref_utils
    return ref_lightgcn, ref_utils


WEIGHT_SET = np.array([0.01, 0.02, 0.03, 0.1, 0.11, 0.5, 1.0], dtype=np.float32)


def small_pairs(rng, n_users, n_items, n_pairs):
    """Unique (user, item) pairs, every node covered, shuffled; weights from the event-weight set."""
    keys = set()
    for u in range(n_users):
        keys.add((u, int(rng.integers(n_items))))
    for i in range(n_items):
        keys.add((int(rng.integers(n_users)), i))
    while len(keys) < n_pairs:
        u = int(min(n_users - 1, rng.pareto(2.0) * n_users / 8))
        i = int(min(n_items - 1, rng.pareto(1.5) * n_items / 8))
        keys.add((u, i))
    pairs = np.array(sorted(keys), dtype=np.int64)
    rng.shuffle(pairs)
    w = WEIGHT_SET[rng.integers(len(WEIGHT_SET), size=len(pairs))]
    return pairs[:, 0], pairs[:, 1], w


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"  {name}.npz  {os.path.getsize(path) / 1024:.0f} KiB")


def train_fixture(ref, utils, seed, dim, layers, n_users=90, n_items=40, n_pairs=420, batch=32):
    rng = np.random.default_rng(1000 * seed + dim + layers)
    u, i, w = small_pairs(rng, n_users, n_items, n_pairs)
    df = pd.DataFrame({"user_id_idx": u, "item_id_idx": i + n_users, "weight": w})
    edge_index, edge_weight = utils.df_to_graph(df, True)            # reference code
    torch.manual_seed(seed)
    model = ref.LightGCN(n_users + n_items, dim, layers)               # reference code
    weight0 = model.embedding.weight.detach().clone()
    users = torch.from_numpy(rng.choice(n_users, batch, replace=False))
    pos = torch.from_numpy(rng.integers(n_items, size=batch) + n_users)
    neg = torch.from_numpy(rng.integers(n_items, size=batch) + n_users)
    labels = utils.batch_pos_neg_edges(users, pos, neg)               # reference code
    with torch.no_grad():
        emb = model.get_embedding(edge_index, edge_weight)
    decay, lr = 1e-4, 0.005                                            # train_lightgcn.py:50-51
    opt = torch.optim.Adam(model.parameters(), lr)
    snaps = {}
    for step in range(3):                                              # train_lightgcn.py:130-147
        opt.zero_grad()
        out = model(edge_index, labels, edge_weight)
        size = len(users)
        bpr = model.recommendation_loss(out[:size], out[size:], 0) * size
        reg = utils.regularization_loss(model.embedding.weight, size, users, pos, neg, decay)
        loss = bpr + reg
        loss.backward()
        if step == 0:
            snaps.update(scores=out.detach().clone(), bpr=bpr.detach().clone(),
                         reg=reg.detach().clone(), grad=model.embedding.weight.grad.clone())
        opt.step()
        if step in (0, 2):
            snaps[f"weight_after_{step + 1}"] = model.embedding.weight.detach().clone()
    sel = list(range(0, n_users, 7))
    inter = torch.zeros(len(sel), n_items)
    for r, uu in enumerate(sel):
        inter[r, i[(u == uu) & (w == 1.0)]] = 1.0
    model2 = ref.LightGCN(n_users + n_items, dim, layers)
    model2.load_state_dict({"alpha": model.alpha, "embedding.weight": weight0})
    with torch.no_grad():
        topk_df = model2.recommendK(edge_index, edge_weight, n_users, n_items, inter, sel, 5)
    save(f"train_s{seed}_d{dim}_k{layers}", edge_index=edge_index, edge_weight=edge_weight,
         weight0=weight0, alpha=model.alpha, labels=labels, users=users, pos=pos, neg=neg,
         n_users=n_users, n_items=n_items, decay=decay, lr=lr, embedding=emb,
         rec_users=np.array(sel), rec_seen=inter, rec_topk=np.array(topk_df["top_rlvnt_itm"].tolist()),
         **snaps)


def edge_case_fixtures(ref, utils):
    rng = np.random.default_rng(77)
    n_users, n_items, dim, layers = 40, 16, 64, 3
    n = n_users + n_items + 2                 # two extra nodes that no edge touches
    u, i, w = small_pairs(rng, n_users, n_items, 150)
    df = pd.DataFrame({"user_id_idx": u, "item_id_idx": i + n_users, "weight": w})
    ei, ew = utils.df_to_graph(df, True)

    def run(tag, ei, ew, alpha=None, labels=None, **conv_kw):
        torch.manual_seed(5)
        model = ref.LightGCN(n, dim, layers, alpha=alpha, **conv_kw)
        with torch.no_grad():
            emb = model.get_embedding(ei, ew)
            scores = model(ei, labels, ew)
        extra = {} if ew is None else {"edge_weight": ew}
        if labels is not None:
            extra["labels"] = labels
        save("edge_" + tag, edge_index=ei, weight0=model.embedding.weight, alpha=model.alpha,
             embedding=emb, scores=scores, normalize=int(conv_kw.get("normalize", True)), **extra)

    lab = torch.stack((torch.arange(0, 20), torch.arange(20, 40)))
    run("isolated", ei, ew, labels=lab)                                   # nodes n-2, n-1 have degree 0
    dup_ei = torch.cat([ei, ei[:, :10]], dim=1)                           # duplicated edges add
    run("duplicate", dup_ei, torch.cat([ew, ew[:10]]), labels=lab)
    run("noweight", ei, None, labels=lab)                                 # edge_weight=None -> ones
    run("nolabels", ei, ew, labels=None)                                  # scores over all edges
    run("alpha_tensor", ei, ew, alpha=torch.tensor([0.5, 0.25, 0.0, 0.125]), labels=lab)
    run("alpha_float", ei, ew, alpha=0.3, labels=lab)
    run("no_normalize", ei, ew, labels=lab, normalize=False)
    zw = ew.clone()
    zw[3] = 0.0
    zw[3 + len(u)] = 0.0
    run("zero_weight", ei, zw, labels=lab)
    # one node whose only incident weights are negative: deg < 0 -> NaN rows propagate, as upstream
    nw = ew.clone()
    tgt = int(ei[1, 0])
    nw[ei[1] == tgt] = -0.09
    nw[ei[0] == tgt] = -0.09
    run("negative_weight", ei, nw, labels=lab)


def hub_fixture(ref, utils):
    """One item of degree >= 1e4.  Inputs are regenerated from the seed (tests/tests_support.py);
    only row samples, row norms and column sums are stored."""
    from tests_support import formula_weight, hub_pairs
    seed, n_users, n_items, dim, layers = 3, 12000, 30, 64, 3
    u, i, w = hub_pairs(seed, n_users, n_items)
    df = pd.DataFrame({"user_id_idx": u, "item_id_idx": i + n_users, "weight": w})
    ei, ew = utils.df_to_graph(df, True)
    n = n_users + n_items
    model = ref.LightGCN(n, dim, layers)
    model.load_state_dict({"alpha": model.alpha, "embedding.weight": torch.from_numpy(formula_weight(n, dim))})
    with torch.no_grad():
        emb = model.get_embedding(ei, ew)
    rows = np.concatenate([np.arange(n_users, n), np.arange(0, n_users, 997)])
    digest = hashlib.sha256(ei.numpy().tobytes() + ew.numpy().tobytes()).hexdigest()
    save("hub_s3", seed=seed, n_users=n_users, n_items=n_items, dim=dim, layers=layers,
         input_sha256=np.frombuffer(digest.encode(), dtype=np.uint8), rows=rows, embedding_rows=emb[rows],
         row_l2=emb.norm(dim=1), hub_degree=int((i == 0).sum()),
         embedding_sum64=emb.double().sum(0))


def sampler_fixture(utils):
    """batch_loader of the reference itself (src/utils_v2.py:168-181) under random.seed: pins the oracle's
    restatement of the sampler's semantics AND its RNG call order."""
    import random
    from tests_support import sampler_lists
    n_users, n_items = 60, 25
    order, pos, ign = sampler_lists(n_users, n_items, seed=1)
    frame = pd.DataFrame({"user_id_idx": order, "item_id_idx_list": [pos[u] for u in order],
                          "ignor_neg_list": [ign[u] for u in order]})
    out = {}
    for seed in (0, 1, 2):
        random.seed(seed)
        triples = [utils.batch_loader(frame, 16, n_users, n_items) for _ in range(3)]        # reference code
        out[f"seed{seed}"] = torch.stack([torch.stack(t) for t in triples])                 # [3 batches, 3, 16]
    save("sampler_ref", n_users=n_users, n_items=n_items, lists_seed=1, batch=16, **out)


def main():
    ref, utils = import_reference()
    print("reference imported from", ref.__file__)
    combos = [(s, d, k) for s in (0, 1, 2) for d in (64, 80, 90) for k in (3, 5)]
    for s, d, k in combos:
        train_fixture(ref, utils, s, d, k)
    edge_case_fixtures(ref, utils)
    hub_fixture(ref, utils)
    sampler_fixture(utils)
    meta = {"torch": torch.__version__, "numpy": np.__version__, "pandas": pd.__version__,
            "reference": "happykygo/GNN-eCommerce src/lightgcn.py, src/utils_v2.py (imported, unmodified)",
            "lgconv": "oracle.OracleLGConv (PyG absent: restated, parity unpinned at that boundary)"}
    with open(os.path.join(HERE, "MANIFEST.json"), "w") as f:
        json.dump(meta, f, indent=1)


if __name__ == "__main__":
    main()
