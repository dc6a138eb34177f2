"""CPU oracle for the LightGCN propagation hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-torch, CPU, fp32 restatement of what the reference computes on
the path named by BASELINE.json:north_star.  It exists so that the HIP path can be
checked against it.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; nothing under
``gnn-ecommerce_amd/`` does, and the product path raises when its HIP library is
missing rather than falling back to anything in here.

Pinning status
--------------
* Layer sum, pair scoring, BPR loss, regulariser, label layout, graph layout, Adam
  step order and ``recommendK`` are pinned against the reference's OWN code, executed
  in the build container by ``tests/golden/make_golden.py`` (fixtures committed under
  ``tests/golden/``) -- see ``tests/test_oracle_golden.py``.
* The arithmetic of one ``LGConv`` layer (``gcn_norm`` + gather/scale/scatter-add)
  lives in ``torch_geometric``, which the reference does not vendor or pin
  (``requirements.txt:10``) and which is not installed here.  ``lgconv`` below restates
  PyG's documented behaviour (points (i)-(vi) of section 8c).
  **At that one boundary parity is unpinned**: there is no reference test, golden
  vector or runnable PyG to hold it to.  It is cross-checked only against an
  independent fp64 CSR formulation (``lgconv_fp64``) and the scalar C restatement in
  ``oracle/lgconv_ref.c``.

Reference lines each function follows are cited in its docstring
(paths relative to /root/reference).
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch
from torch import Tensor

__all__ = [
    "gcn_norm", "lgconv", "lgconv_fp64", "OracleLGConv", "get_embedding", "pair_scores",
    "forward", "bpr_loss", "regularization_loss", "batch_pos_neg_edges", "pairs_to_graph",
    "recommend_topk", "train_step_loss", "default_alpha", "batch_loader",
]


# --------------------------------------------------------------------------------------
# One LGConv layer (third-party PyG semantics; parity unpinned -- see header)
# --------------------------------------------------------------------------------------
def gcn_norm(edge_index: Tensor, edge_weight: Optional[Tensor], num_nodes: int) -> Tensor:
    """Symmetric normalisation of the edge values, no self loops.

    Follows the call made at src/lightgcn.py:96 into PyG ``LGConv`` -> ``gcn_norm(...,
    add_self_loops=False)``:  the WEIGHTED in-degree is scattered by
    target index, sequentially in edge order, in fp32; ``deg ** -0.5`` with +inf -> 0;
    ``val = dis[src] * w * dis[dst]`` evaluated left to right.
    """
    src, dst = edge_index[0], edge_index[1]
    if edge_weight is None:
        edge_weight = torch.ones(edge_index.size(1), dtype=torch.float32)
    deg = torch.zeros(num_nodes, dtype=edge_weight.dtype).scatter_add_(0, dst, edge_weight)
    dis = deg.pow(-0.5)
    dis.masked_fill_(dis == float("inf"), 0.0)
    return dis[src] * edge_weight * dis[dst]


def lgconv(x: Tensor, edge_index: Tensor, edge_weight: Optional[Tensor] = None,
           normalize: bool = True) -> Tensor:
    """y[i] = sum over edges e with dst[e] == i of val[e] * x[src[e]].

    The operator called at src/lightgcn.py:96 (``self.convs[i](x, edge_index,
    edge_weight)``).  Gather -> broadcast multiply -> ``index_add_`` over the UNSORTED
    COO list, i.e. every output row is accumulated in edge order in fp32, the product
    being rounded before the add (no FMA) -- the dense-COO route PyG takes for a
    ``[2, E]`` tensor.  The normalisation is recomputed on every call, as upstream does.
    """
    n = x.size(0)
    src, dst = edge_index[0], edge_index[1]
    val = gcn_norm(edge_index, edge_weight, n) if normalize else edge_weight
    msg = x.index_select(0, src)
    if val is not None:
        msg = val.view(-1, 1) * msg
    return torch.zeros_like(x).index_add_(0, dst, msg)


def lgconv_fp64(x: Tensor, edge_index: Tensor, val: Tensor) -> Tensor:
    """Independent formulation used only to measure error: fp64 CSR SpMM on given fp32 values."""
    n = x.size(0)
    a = torch.sparse_coo_tensor(torch.stack([edge_index[1], edge_index[0]]), val.double(), (n, n))
    return torch.sparse.mm(a.coalesce(), x.double())


class OracleLGConv(torch.nn.Module):
    """Module form of ``lgconv`` with the constructor/forward signature of PyG's LGConv
    (``LGConv(normalize=True)``, ``forward(x, edge_index, edge_weight=None)``,
    parameter-free, ``reset_parameters`` a no-op) -- what src/lightgcn.py:82 instantiates."""

    def __init__(self, normalize: bool = True, **kwargs):
        super().__init__()
        self.normalize = normalize

    def reset_parameters(self):
        pass

    def forward(self, x: Tensor, edge_index: Tensor, edge_weight: Optional[Tensor] = None) -> Tensor:
        return lgconv(x, edge_index, edge_weight, self.normalize)


# --------------------------------------------------------------------------------------
# LightGCN around it (pinned by the reference's own code through tests/golden)
# --------------------------------------------------------------------------------------
def default_alpha(num_layers: int) -> Tensor:
    """src/lightgcn.py:72-78: uniform 1/(K+1), built from a Python float list."""
    return torch.tensor([1.0 / (num_layers + 1)] * (num_layers + 1))


def get_embedding(weight: Tensor, alpha: Tensor, edge_index: Tensor,
                  edge_weight: Optional[Tensor], num_layers: int,
                  normalize: bool = True) -> Tensor:
    """src/lightgcn.py:91-99: out = a0*x; repeat K times: x = conv(x); out = out + a_l*x."""
    x = weight
    out = x * alpha[0]
    for layer in range(num_layers):
        x = lgconv(x, edge_index, edge_weight, normalize)
        out = out + x * alpha[layer + 1]
    return out


def pair_scores(out: Tensor, edge_label_index: Tensor) -> Tensor:
    """src/lightgcn.py:123-125: row gather of both endpoints, product, sum over the feature axis."""
    return (out[edge_label_index[0]] * out[edge_label_index[1]]).sum(dim=-1)


def forward(weight: Tensor, alpha: Tensor, edge_index: Tensor,
            edge_label_index: Optional[Tensor], edge_weight: Optional[Tensor],
            num_layers: int, normalize: bool = True) -> Tensor:
    """src/lightgcn.py:101-125; labels default to the graph's own edges (:115-119)."""
    if edge_label_index is None:
        edge_label_index = edge_index
    return pair_scores(get_embedding(weight, alpha, edge_index, edge_weight, num_layers, normalize),
                       edge_label_index)


def bpr_loss(positives: Tensor, negatives: Tensor, parameters: Optional[Tensor] = None,
             lambda_reg: float = 0.0) -> Tensor:
    """src/lightgcn.py:262-286: (-mean(logsigmoid(p - n)) + lambda*||theta||^2) / n_pairs."""
    n_pairs = positives.size(0)
    log_prob = torch.nn.functional.logsigmoid(positives - negatives).mean()
    reg = 0
    if lambda_reg != 0:
        reg = lambda_reg * parameters.norm(p=2).pow(2)
    return (-log_prob + reg) / n_pairs


def regularization_loss(init_embed: Tensor, batch_size: int, users: Tensor, pos: Tensor,
                        neg: Tensor, decay: float) -> Tensor:
    """src/utils_v2.py:193-211 on the layer-0 table."""
    sq = (init_embed[users].norm().pow(2) + init_embed[pos].norm().pow(2)
          + init_embed[neg].norm().pow(2))
    return (1 / 2) * sq / batch_size * decay


def batch_pos_neg_edges(users: Tensor, pos: Tensor, neg: Tensor) -> Tensor:
    """src/utils_v2.py:184-190: label index [[u|u],[pos|neg]], shape [2, 2B]."""
    return torch.stack((torch.cat([users, users]), torch.cat([pos, neg])))


def pairs_to_graph(user_idx: Tensor, item_idx: Tensor, weight: Optional[Tensor] = None
                   ) -> Tuple[Tensor, Optional[Tensor]]:
    """src/utils_v2.py:146-165 without the DataFrame: symmetric COO [[u|i],[i|u]], weights [w|w].
    ``item_idx`` is already offset by n_users (src/utils_v2.py:128)."""
    ei = torch.stack((torch.cat([user_idx, item_idx]), torch.cat([item_idx, user_idx])))
    if weight is None:
        return ei, None
    return ei, torch.cat([weight, weight])


def recommend_topk(embeds: Tensor, n_users: int, n_items: int, interactions: Tensor,
                   user_ids: Sequence[int], k: int) -> Tensor:
    """src/lightgcn.py:172-177: split, user rows @ items^T, multiplicative seen-mask, topk indices."""
    users, items = torch.split(embeds, [n_users, n_items])
    pred = users[user_ids] @ items.t()
    masked = torch.mul(pred.cpu(), (1 - interactions))
    return masked.topk(k, dim=-1).indices


def train_step_loss(weight: Tensor, alpha: Tensor, edge_index: Tensor, edge_weight: Tensor,
                    users: Tensor, pos: Tensor, neg: Tensor, num_layers: int, decay: float
                    ) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """src/train_lightgcn.py:137-144: scores -> bpr*size + reg.  Returns (scores, bpr, reg, loss)."""
    labels = batch_pos_neg_edges(users, pos, neg)
    scores = forward(weight, alpha, edge_index, labels, edge_weight, num_layers)
    size = len(users)
    bpr = bpr_loss(scores[:size], scores[size:], weight, 0) * size
    reg = regularization_loss(weight, size, users, pos, neg, decay)
    return scores, bpr, reg, bpr + reg


def batch_loader(user_order, pos_lists: dict, ignore_lists: dict, batch_size: int, n_users: int, n_items: int, rng):
    """src/utils_v2.py:168-181 without the DataFrames; ``rng`` is a ``random.Random`` (upstream uses the module
    functions, the same Mersenne Twister).  Same call order as upstream, so the same seed gives the same triples
    (pinned by tests/golden/sampler_ref.npz): users = sample(frame's user column, B); then a positive for every
    user (``random.choice`` over the list, :178); then a negative for every user (rejection-sampled
    ``randint(0, n_items - 1) + n_users`` until it is not in the user's ignore list, :169-173,179)."""
    users = rng.sample(list(user_order), batch_size)
    pos = [rng.choice(list(pos_lists[u])) for u in users]
    neg = []
    for u in users:
        ign = ignore_lists[u]
        while True:
            cand = rng.randint(0, n_items - 1) + n_users
            if cand not in ign:
                neg.append(cand)
                break
    return torch.LongTensor(users), torch.LongTensor(pos), torch.LongTensor(neg)
