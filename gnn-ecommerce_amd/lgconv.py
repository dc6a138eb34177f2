"""``LGConv`` -- the operator surface the reference imports from PyG (src/lightgcn.py:9, :82, :96).

Same constructor and call signature as upstream's layer (``LGConv(normalize=True)``;
``forward(x, edge_index, edge_weight=None) -> Tensor[N, D]``; parameter-free;
``reset_parameters()`` a no-op; differentiable w.r.t. ``x`` only), but the work is one launch of
the HIP CSR-SpMM on the cached graph instead of gather -> scale -> scatter-add over ``[E, D]``
transients.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from .graph import get_graph
from .propagate import hop


class LGConv(torch.nn.Module):
    def __init__(self, normalize: bool = True, **kwargs):
        super().__init__()
        unknown = set(kwargs) - {"aggr", "flow", "node_dim"}
        if unknown:
            raise TypeError(f"LGConv got unexpected arguments {sorted(unknown)}")
        if kwargs.get("aggr", "add") != "add" or kwargs.get("flow", "source_to_target") != "source_to_target":
            raise NotImplementedError("only aggr='add', flow='source_to_target' (the upstream defaults) are built")
        self.normalize = bool(normalize)

    def reset_parameters(self) -> None:
        """No parameters (kept because src/lightgcn.py:88-89 calls it on every conv)."""

    def forward(self, x: Tensor, edge_index: Tensor, edge_weight: Optional[Tensor] = None) -> Tensor:
        graph = get_graph(edge_index, edge_weight, x.size(0), self.normalize)
        return hop(x, graph)

    def extra_repr(self) -> str:
        return f"normalize={self.normalize}"
