"""``torch.optim.Adam`` for the embedding table in one pass over it (the caller's side of the step).

The reference trains with ``torch.optim.Adam(model.parameters(), lr=...)`` (src/train_lightgcn.py:58) and calls
``optimizer.step()`` after every backward (:147): a dense update of the whole ``[N, D]`` table -- moments decay everywhere,
so exact Adam must touch every row.  ``Adam`` here is that optimizer for dense fp32 CUDA parameters with the same
hyper-parameters, the same per-parameter state (``step``, ``exp_avg``, ``exp_avg_sq``: a ``state_dict`` loads into
``torch.optim.Adam`` and back) and the same arithmetic, executed by ``lgc_adam_step``: w, g, m, v read once and w, m, v
written once (7 x 434 MB at 1.7 M x 64).  Anything else (amsgrad, weight decay, maximize, sparse or non-fp32 gradients,
CPU tensors) is refused -- use ``torch.optim.Adam`` for those.
"""
from __future__ import annotations

import math

import torch

from . import _native


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        if lr < 0.0 or eps < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError("invalid Adam hyper-parameters")
        # the param-group keys of torch.optim.Adam, so that a state_dict moves between the two in either direction
        # (src/utils_v2.py:214-232 stores optimizer.state_dict() in the checkpoint); everything but lr / betas / eps is
        # fixed at torch's default and refused otherwise
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False, maximize=False,
                                      foreach=None, capturable=False, differentiable=False, fused=None,
                                      decoupled_weight_decay=False))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _native.load()
        for group in self.param_groups:
            if group.get("weight_decay") or group.get("amsgrad") or group.get("maximize"):
                raise RuntimeError("gnn_ecommerce_amd.optim.Adam: weight_decay / amsgrad / maximize are not supported")
            beta1, beta2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                g = p.grad
                if g.is_sparse or g.dtype != torch.float32 or p.dtype != torch.float32:
                    raise RuntimeError("gnn_ecommerce_amd.optim.Adam takes dense fp32 gradients only")
                _native.require_device(p, "parameter")
                if not p.is_contiguous():
                    raise RuntimeError("gnn_ecommerce_amd.optim.Adam takes contiguous parameters only")
                g = g.contiguous()
                state = self.state[p]
                if not state:
                    state["step"] = torch.tensor(0.0)                    # torch.optim.Adam's state layout
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["step"] += 1
                t = int(state["step"].item())                            # a host tensor: no device sync
                step_size = group["lr"] / (1.0 - beta1 ** t)
                bc2_sqrt = math.sqrt(1.0 - beta2 ** t)
                with torch.cuda.device(p.device):
                    code = lib.lgc_adam_step(_native.ptr(p), _native.ptr(g), _native.ptr(state["exp_avg"]),
                                             _native.ptr(state["exp_avg_sq"]), p.numel(), 1.0 - beta1, beta2, 1.0 - beta2,
                                             group["eps"], step_size, bc2_sqrt, _native.stream_of(p.device))
                _native.check(code, "lgc_adam_step")
                # the kernel's write is invisible to autograd's version counter, which recommendK's cache keys on
                torch.autograd.graph.increment_version(p)
        return loss
