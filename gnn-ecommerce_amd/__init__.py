"""MI355X-native LightGCN propagation (the hot path of happykygo/GNN-eCommerce).

``LightGCN`` / ``BPRLoss`` / ``LGConv`` are drop-ins for the reference's ``src/lightgcn.py``
surface; the arithmetic runs in ``csrc/liblgconv_hip.so`` (hand-written HIP for gfx950) behind the
C ABI of ``include/lgconv_hip.h``.  There is no CPU fallback.

The directory is called ``gnn-ecommerce_amd`` (not an importable name); ``import gnn_ecommerce_amd``
resolves to it through the shim package of that name at the repository root.
"""
from . import _native
from .graph import PropGraph, build_row_plan, clear_cache, get_graph
from .lgconv import LGConv
from .lightgcn import BPRLoss, LightGCN, regularization_loss
from .propagate import check_index_status, hop, pair_dot, propagate_sum
from .sampler import TripleSampler
from . import ingest, serving
from .trainer import PartitionedTrainer

__all__ = ["LightGCN", "BPRLoss", "LGConv", "PropGraph", "get_graph", "clear_cache", "build_row_plan",
           "propagate_sum", "hop", "pair_dot", "check_index_status", "TripleSampler", "regularization_loss", "PartitionedTrainer", "_native"]
