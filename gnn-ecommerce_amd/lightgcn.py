"""Drop-in ``LightGCN`` / ``BPRLoss`` for src/lightgcn.py (and its copy torchserve/lightgcn.py).

Public surface kept identical to the reference so that src/train_lightgcn.py,
src/inference_lightgcn.py and torchserve/lightgcn_handler.py run unchanged
(SURVEY.md section 8b): constructor arguments, the attributes callers read
(``embedding.weight``, ``alpha``, ``convs``, ``num_nodes`` ...), ``state_dict`` keys
(``alpha``, ``embedding.weight``), and the methods ``get_embedding``, ``forward``,
``predict_link``, ``recommend``, ``recommendK``, ``MARK_MAPK``, ``link_pred_loss``,
``recommendation_loss``.  What changes is underneath: propagation is the HIP CSR-SpMM with
the layer sum fused (``propagate.propagate_sum``) and pair scoring is one gather-dot kernel.
"""
from __future__ import annotations

from typing import Optional, Union

import torch
import torch.nn.functional as F
from torch import Tensor
from torch.nn.modules.loss import _Loss

from . import _native
from .graph import get_graph
from .lgconv import LGConv
from .propagate import (TOPK_MAX, RegHook, SeenLists, bpr_loss_fused, mask_topk, pair_dot, propagate_sum,
                        regularization_through, routable_index, scores_from_table)

__all__ = ["LightGCN", "BPRLoss", "LGConv", "regularization_loss"]


def _is_sparse_tensor(obj) -> bool:
    # torch_sparse is optional; only used to mirror the isinstance branch of src/lightgcn.py:116
    return type(obj).__name__ == "SparseTensor" and hasattr(obj, "coo")


class LightGCN(torch.nn.Module):
    """x_i = sum_l alpha_l x_i^(l),  x^(l+1) = D^-1/2 A D^-1/2 x^(l)   (He et al., 2020).

    Args mirror src/lightgcn.py:58-65: ``num_nodes``, ``embedding_dim``, ``num_layers``,
    ``alpha`` (None -> uniform 1/(K+1); float -> repeated; Tensor of K+1 entries), and
    ``**kwargs`` forwarded to every ``LGConv``.
    """

    def __init__(self, num_nodes: int, embedding_dim: int, num_layers: int,
                 alpha: Optional[Union[float, Tensor]] = None, **kwargs):
        super().__init__()
        self.num_nodes, self.embedding_dim, self.num_layers = num_nodes, embedding_dim, num_layers
        if alpha is None:
            alpha = 1. / (num_layers + 1)
        if isinstance(alpha, Tensor):
            assert alpha.size(0) == num_layers + 1
        else:
            alpha = torch.tensor([alpha] * (num_layers + 1))
        self.register_buffer('alpha', alpha)
        self.embedding = torch.nn.Embedding(num_nodes, embedding_dim)
        self.convs = torch.nn.ModuleList(LGConv(**kwargs) for _ in range(num_layers))
        self._alpha_host = None
        # serving: recommendK reuses the propagated table while neither the graph, the
        # weights nor alpha changed; set to False to recompute on every request like the reference
        self.cache_recommend_embeddings = True
        self._served = None
        self.reset_parameters()

    def reset_parameters(self):
        torch.nn.init.xavier_uniform_(self.embedding.weight)
        for conv in self.convs:
            conv.reset_parameters()

    # -- hot path ------------------------------------------------------------------------
    def _alphas(self) -> tuple:
        """Host copy of the alpha buffer (kernel arguments), refreshed when the buffer changes."""
        a = self.alpha
        tag = (a.data_ptr(), a._version, a.device)
        if self._alpha_host is None or self._alpha_host[0] != tag:
            self._alpha_host = (tag, tuple(float(v) for v in a.detach().cpu().tolist()))
        return self._alpha_host[1]

    def get_embedding(self, edge_index, edge_weight) -> Tensor:
        """All K hops and the weighted layer sum in K kernel sequences (src/lightgcn.py:91-99)."""
        x0 = self.embedding.weight
        _native.require_device(x0, "LightGCN.embedding.weight")
        normalize = self.convs[0].normalize if self.num_layers > 0 else True
        graph = get_graph(edge_index, edge_weight, self.num_nodes, normalize)
        return propagate_sum(x0, graph, self._alphas())

    def _serving_embedding(self, edge_index, edge_weight) -> Tensor:
        """The propagated table for read-only use by ``recommendK``.

        Upstream recomputes all K layers for every request (torchserve/lightgcn_handler.py:91 ->
        src/lightgcn.py:171) although, with gradients off, the result depends only on the graph, the weight
        table and alpha.  It is kept here until one of them changes (tensor identity + version counter; an
        optimizer step or load_state_dict bumps the weight's version)."""
        w = self.embedding.weight
        if torch.is_grad_enabled() or not self.cache_recommend_embeddings:
            return self.get_embedding(edge_index, edge_weight)
        normalize = self.convs[0].normalize if self.num_layers > 0 else True
        graph = get_graph(edge_index, edge_weight, self.num_nodes, normalize)
        key = (id(graph), w.data_ptr(), w._version, w.device, self._alphas())
        if self._served is None or self._served[0] != key:
            self._served = (key, graph, self.get_embedding(edge_index, edge_weight))
        return self._served[2]

    def invalidate(self) -> None:
        """Forget every derived table: the cached graphs of this process and the propagated table ``recommendK``
        reuses.  The caches key on tensor identity + version counter, which in-place torch ops bump; call this after
        writes that do not (``weight.data.copy_``, a DLPack/numpy alias, a foreign kernel).  The reference re-derives
        everything on every call."""
        from .graph import clear_cache
        clear_cache()
        self._served = None
        self._alpha_host = None

    def forward(self, edge_index, edge_label_index: Optional[Tensor] = None,
                edge_weight: Optional[Tensor] = None) -> Tensor:
        """Scores of the node pairs in ``edge_label_index`` (default: the graph's own edges)."""
        if edge_label_index is None:
            if _is_sparse_tensor(edge_index):
                edge_label_index = torch.stack(edge_index.coo()[:2], dim=0)
            else:
                edge_label_index = edge_index
        x0 = self.embedding.weight
        _native.require_device(x0, "LightGCN.embedding.weight")
        normalize = self.convs[0].normalize if self.num_layers > 0 else True
        graph = get_graph(edge_index, edge_weight, self.num_nodes, normalize)
        hook = RegHook(x0) if (torch.is_grad_enabled() and x0.requires_grad) else None
        scores = scores_from_table(x0, graph, self._alphas(), edge_label_index, hook)
        # a regulariser on the layer-0 rows of this step (regularization_loss below) routes its gradient through the
        # scoring node of THIS forward; None when the scores took the dense two-node path
        x0._lgcn_reg_hook = hook if (hook is not None and hook.token is not None) else None
        return scores

    # -- heads built on it -----------------------------------------------------------------
    def predict_link(self, edge_index, edge_label_index: Optional[Tensor] = None, prob: bool = False) -> Tensor:
        pred = self(edge_index, edge_label_index).sigmoid()
        return pred if prob else pred.round()

    def recommend(self, edge_index, src_index: Optional[Tensor] = None, dst_index: Optional[Tensor] = None,
                  k: int = 1) -> Tensor:
        # Upstream calls get_embedding(edge_index) here without the edge_weight its own signature
        # requires (src/lightgcn.py:153 vs :91) and so raises TypeError; kept bug-for-bug.
        out_src = out_dst = self.get_embedding(edge_index)
        if src_index is not None:
            out_src = out_src[src_index]
        if dst_index is not None:
            out_dst = out_dst[dst_index]
        top_index = (out_src @ out_dst.t()).topk(k, dim=-1).indices
        if dst_index is not None:
            top_index = dst_index[top_index.view(-1)].view(*top_index.size())
        return top_index

    def recommendK(self, edge_index, edge_weight, n_users, n_items, interactions_t, user_id_list, k: int = 5):
        """Top-k unseen items per user as the DataFrame src/lightgcn.py:169-182 returns
        (columns ``user_ID``, ``top_rlvnt_itm``); seen items are zeroed, not removed, as upstream."""
        import pandas as pd
        embeds = self._serving_embedding(edge_index, edge_weight)
        users, items = torch.split(embeds, [n_users, n_items])
        # Scores, the multiplicative seen-mask and the top-k all stay on the device: only the [n_sel, k] indices
        # come back (upstream ships the whole [n_sel, n_items] score matrix to the host first, :174).  Same fp32
        # products as ``torch.mul(pred.cpu(), 1 - interactions_t)``; the mask may already live on the device.
        sel = torch.as_tensor(user_id_list, device=embeds.device) if not torch.is_tensor(user_id_list) \
            else user_id_list.to(embeds.device)
        sel = sel.reshape(-1).long()
        pred = users.index_select(0, sel) @ items.t()
        if isinstance(interactions_t, SeenLists):                    # purchase lists on the device: no dense mask at all
            if k > TOPK_MAX:
                raise ValueError(f"the list form of the mask supports k <= {TOPK_MAX}")
            top_index = mask_topk(pred, interactions_t.for_users(sel), k).cpu()
            return pd.DataFrame({'user_ID': list(user_id_list), 'top_rlvnt_itm': top_index.numpy().tolist()})
        seen = interactions_t.to(device=embeds.device, dtype=pred.dtype, non_blocking=True)
        if k <= TOPK_MAX:
            top_index = mask_topk(pred, seen.expand_as(pred).contiguous(), k).cpu()       # one launch: lgc_mask_topk
        else:
            top_index = torch.mul(pred, (1 - seen)).topk(k, dim=-1).indices.cpu()
        if isinstance(user_id_list, (list, tuple)):
            # the frame upstream builds in three steps (:178-182), built directly: same columns, dtypes, index, values
            return pd.DataFrame({'user_ID': list(user_id_list), 'top_rlvnt_itm': top_index.numpy().tolist()})
        frame = pd.DataFrame(top_index.numpy())           # anything index-carrying (a Series): upstream's own steps
        frame['top_rlvnt_itm'] = frame.values.tolist()
        frame['user_ID'] = user_id_list
        return frame[['user_ID', 'top_rlvnt_itm']]

    def MARK_MAPK(self, test_pos_list_df, top_index_df, k):
        import pandas as pd
        m = pd.merge(test_pos_list_df, top_index_df, how='left', left_on='user_id_idx', right_on='user_ID')
        m['overlap_item'] = [list(set(a).intersection(b)) for a, b in zip(m.item_id_idx_list, m.top_rlvnt_itm)]
        m['recall'] = m.apply(lambda x: len(x['overlap_item']) / len(x['item_id_idx_list']), axis=1)
        m['precision'] = m.apply(lambda x: len(x['overlap_item']) / k, axis=1)
        return m['precision'].mean(), m['recall'].mean(), m

    def link_pred_loss(self, pred: Tensor, edge_label: Tensor, **kwargs) -> Tensor:
        return torch.nn.BCEWithLogitsLoss(**kwargs)(pred, edge_label.to(pred.dtype))

    def recommendation_loss(self, pos_edge_rank: Tensor, neg_edge_rank: Tensor,
                            lambda_reg: float = 1e-4, **kwargs) -> Tensor:
        return BPRLoss(lambda_reg, **kwargs)(pos_edge_rank, neg_edge_rank, self.embedding.weight)

    def regularization_loss(self, users: Tensor, pos_items: Tensor, neg_items: Tensor, decay: float,
                            batch_size: Optional[int] = None) -> Tensor:
        """``regularization_loss(model.embedding.weight, size, users, pos, neg, decay)`` of src/utils_v2.py:193-211 (the
        call at src/train_lightgcn.py:142): same value, same gradient -- see the module function below."""
        return regularization_loss(self.embedding.weight, len(users) if batch_size is None else batch_size, users,
                                   pos_items, neg_items, decay)

    def __repr__(self) -> str:
        return f'{self.__class__.__name__}({self.num_nodes}, {self.embedding_dim}, num_layers={self.num_layers})'


def regularization_loss(init_embed: Tensor, batch_size: int, batch_usr: Tensor, batch_pos: Tensor, batch_neg: Tensor,
                        decay: float) -> Tensor:
    """src/utils_v2.py:193-211 with the reference's signature:  decay / 2 * (|E0[u]|^2 + |E0[p]|^2 + |E0[n]|^2) / size.

    Upstream's autograd turns the three row gathers into three dense, zero-filled [N, D] gradients and adds them to the
    dense gradient of the scores (0.9 ms of a 5.2 ms step at 1.7 M x 64).  When ``init_embed`` is the weight a
    ``LightGCN.forward`` of this step has just scored from, the value is computed the same way but its gradient --
    ``decay / size * E0[r]`` on the <= 3B rows -- is handed to that forward's scoring node, whose backward adds it to
    the rows of the dense gradient it writes anyway.  In every other situation (no forward yet, gradients off, the
    dense two-node path, the graph already consumed) this is upstream's expression on plain torch ops."""
    hook = getattr(init_embed, "_lgcn_reg_hook", None)
    if (hook is not None and not hook.spent and hook.token is not None and hook.weight is init_embed
            and torch.is_grad_enabled() and init_embed.is_cuda and batch_size > 0       # (size 0: upstream's inf / nan)
            and all(routable_index(t) for t in (batch_usr, batch_pos, batch_neg))):
        return regularization_through(hook, batch_size, batch_usr, batch_pos, batch_neg, decay)
    reg_loss = (1 / 2) * (init_embed[batch_usr].norm().pow(2) + init_embed[batch_pos].norm().pow(2)
                          + init_embed[batch_neg].norm().pow(2)) / batch_size
    return reg_loss * decay


class BPRLoss(_Loss):
    """(-mean(log sigmoid(pos - neg)) + lambda_reg * ||parameters||^2) / n_pairs  (src/lightgcn.py:262-286)."""
    __constants__ = ['lambda_reg']
    lambda_reg: float

    def __init__(self, lambda_reg: float = 0, **kwargs) -> None:
        super().__init__(None, None, "sum", **kwargs)
        self.lambda_reg = lambda_reg

    def forward(self, positives: Tensor, negatives: Tensor, parameters: Tensor = None) -> Tensor:
        n_pairs = positives.size(0)
        if self.lambda_reg == 0 and positives.dim() == 1:
            fused = bpr_loss_fused(positives, negatives)      # the same value and gradient in one launch each way
            if fused is not None:
                return fused
        loss = -F.logsigmoid(positives - negatives).mean()
        if self.lambda_reg != 0:
            loss = loss + self.lambda_reg * parameters.norm(p=2).pow(2)
        return loss / n_pairs
