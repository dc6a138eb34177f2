"""Drop-in for this file of happykygo/GNN-eCommerce: the LightGCN class surface on the MI355X propagation path."""
from gnn_ecommerce_amd.lightgcn import LightGCN, BPRLoss, LGConv  # noqa: F401
