"""Handler-shaped serving on the drop-in model: the build-side counterpart of
torchserve/lightgcn_handler.py (initialize / preprocess / inference / postprocess, SURVEY.md 8a a-S).

TorchServe itself is out of scope (and not installed): the class takes the same ``context`` object TorchServe hands a
handler -- ``context.manifest['model']['serializedFile']``, ``context.system_properties['model_dir' | 'gpu_id']`` -- and
returns what upstream returns, ``[{'items': [[k item indices], ...]}]``.  What differs is where the time goes:
  * ``initialize`` loads the persisted graph (``ingest.save_serving_graph``) instead of re-reading the CSV and
    rebuilding the COO (lightgcn_handler.py:32-38), and keeps the purchased-items lists on the device as a CSR;
  * ``inference`` calls ``LightGCN.recommendK`` with the graph object and the purchase lists (``SeenLists``): the K-layer
    propagate is reused across requests, scores / mask / top-k stay on the device (lgc_mask_topk builds the request's
    mask rows as bitmasks in LDS; the dense ``[n_sel, n_items]`` mask upstream materialises never exists).
"""
from __future__ import annotations

import os
from typing import List

import torch

from . import _native
from .graph import PropGraph
from .lightgcn import LightGCN
from .propagate import SeenLists

GRAPH_FILE = "graph.safetensors"


def load_checkpoint(path: str) -> dict:
    """A checkpoint in the layout of src/utils_v2.py:214-232 (``model_state_dict`` + ``hyperparams``), read with the
    loader that executes nothing from the file."""
    return torch.load(path, map_location="cpu", weights_only=True)


class RecommendHandler:
    def __init__(self):
        self.initialized = False

    def initialize(self, context) -> None:
        self.manifest = context.manifest
        properties = context.system_properties
        model_dir = properties.get("model_dir")
        if not torch.cuda.is_available():
            raise _native.NativeLibraryError("serving needs a ROCm device (no CPU fallback)")
        self.device = torch.device("cuda:" + str(properties.get("gpu_id") or 0))
        model_pt_path = os.path.join(model_dir, self.manifest["model"]["serializedFile"])
        if not os.path.isfile(model_pt_path):
            raise RuntimeError("Missing the model.pt file")                     # lightgcn_handler.py:29-30
        graph_path = os.path.join(model_dir, GRAPH_FILE)
        if not os.path.isfile(graph_path):
            raise RuntimeError(f"Missing {GRAPH_FILE} (write it once with gnn_ecommerce_amd.ingest.save_serving_graph)")
        self.graph, extra, meta = PropGraph.load(graph_path, self.device, with_extra=True)
        try:
            self.n_users, self.n_items = int(meta["n_users"]), int(meta["n_items"])
            self.seen_ptr, self.seen_items = extra["seen_ptr"].to(self.device), extra["seen_items"].to(self.device)
        except (KeyError, ValueError) as exc:
            raise ValueError(f"{graph_path}: not a serving graph ({exc!r} missing or malformed)") from exc
        # everything the request path takes on trust is checked here, once: node counts against the graph, the stored
        # user|item split against n_users, and the purchase lists lgc_mask_topk indexes without bounds
        if self.n_users < 1 or self.n_items < 1 or self.n_users + self.n_items != self.graph.num_nodes:
            raise ValueError(f"{graph_path}: n_users {self.n_users} + n_items {self.n_items} != {self.graph.num_nodes} nodes")
        if self.graph.split is not None and self.graph.split != self.n_users:
            raise ValueError(f"{graph_path}: bipartite split {self.graph.split} != n_users {self.n_users}")
        self.seen = SeenLists(self.seen_ptr, self.seen_items).validate(self.n_users, graph_path + ": purchase lists")
        state = load_checkpoint(model_pt_path)
        hp = state["hyperparams"]
        self.model = LightGCN(self.n_users + self.n_items, hp["latent_dim"], hp["n_layers"])
        self.model.load_state_dict(state["model_state_dict"])
        self.model.to(self.device).eval()
        self.k = 20                                                              # lightgcn_handler.py:90
        self.initialized = True

    def preprocess(self, data):
        body = data[0].get("data")
        if body is None:
            body = data[0].get("body")
        return body

    def seen_rows(self, users: torch.Tensor) -> torch.Tensor:
        """Dense fp32 [len(users), n_items] rows of the purchased-items matrix, built on the device."""
        lo, hi = self.seen_ptr[users], self.seen_ptr[users + 1]
        counts = hi - lo
        rows = torch.repeat_interleave(torch.arange(users.numel(), device=self.device), counts)
        first = torch.cumsum(counts, 0) - counts
        pos = torch.arange(int(counts.sum().item()), device=self.device) - first[rows] + lo[rows]
        out = torch.zeros((users.numel(), self.n_items), dtype=torch.float32, device=self.device)
        out[rows, self.seen_items[pos]] = 1.0
        return out

    def inference(self, data, *args, **kwargs):
        users = [int(u) for u in data]
        if any(u < 0 or u >= self.n_users for u in users):      # upstream's embedding gather raises IndexError as well
            raise IndexError(f"user index outside [0, {self.n_users})")
        with torch.no_grad():
            # the purchase lists stay a CSR on the device; lgc_mask_topk turns the request's rows into LDS bitmasks
            frame = self.model.recommendK(self.graph, None, self.n_users, self.n_items, self.seen, list(data), self.k)
            return {"items": list(frame["top_rlvnt_itm"])}

    def postprocess(self, data) -> List[dict]:
        return [data]

    def handle(self, data, context=None):
        return self.postprocess(self.inference(self.preprocess(data)))
