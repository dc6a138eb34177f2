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
    """``row_ranges`` (optional, a list of ``(lo, hi)`` row ranges): only these rows of every 2-D parameter are updated --
    the rows a rank OWNS in a partitioned run (its users + the replicated item block, ``partition.owned_row_ranges``).
    A row whose gradient has been zero in every step so far has ``m = v = 0`` and Adam leaves it where it is
    (``w - step_size * 0 / (0 + eps)``), so skipping the rows other ranks own changes no bit of the rows this rank
    owns and spares 7 x 4 bytes of traffic per skipped element."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, row_ranges=None):
        if lr < 0.0 or eps < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError("invalid Adam hyper-parameters")
        # the param-group keys of torch.optim.Adam, so that a state_dict moves between the two in either direction
        # (src/utils_v2.py:214-232 stores optimizer.state_dict() in the checkpoint); everything but lr / betas / eps is
        # fixed at torch's default and refused otherwise
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False, maximize=False,
                                      foreach=None, capturable=False, differentiable=False, fused=None,
                                      decoupled_weight_decay=False))
        self.row_ranges = None if row_ranges is None else [(int(lo), int(hi)) for lo, hi in row_ranges if int(hi) > int(lo)]

    @staticmethod
    def _check_buffer(name: str, t: torch.Tensor, p: torch.Tensor) -> torch.Tensor:
        """Raw pointers go to the kernel: after load_state_dict from a foreign or CPU-mapped state a moment on another
        device, of another shape or a strided view would be an out-of-bounds or host-pointer write on the GPU."""
        if not torch.is_tensor(t) or t.device != p.device or t.dtype != torch.float32 or t.shape != p.shape or not t.is_contiguous():
            raise RuntimeError(f"gnn_ecommerce_amd.optim.Adam: {name} must be a contiguous fp32 tensor of shape "
                               f"{tuple(p.shape)} on {p.device}; got "
                               f"{(t.dtype, tuple(t.shape), str(t.device)) if torch.is_tensor(t) else type(t).__name__}")
        return t

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
                if g.device != p.device or g.shape != p.shape:
                    raise RuntimeError(f"gnn_ecommerce_amd.optim.Adam: gradient {tuple(g.shape)} on {g.device} does not "
                                       f"match its parameter {tuple(p.shape)} on {p.device}")
                g = g.contiguous()
                state = self.state[p]
                if not state:
                    state["step"] = torch.tensor(0.0)                    # torch.optim.Adam's state layout
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if torch.is_tensor(state["step"]) and state["step"].device.type != "cpu":
                    # a state loaded from a capturable / fused torch Adam keeps `step` on the device: one copy now
                    # instead of a device sync in every step
                    state["step"] = state["step"].detach().to("cpu", torch.float32)
                m = self._check_buffer("exp_avg", state["exp_avg"], p)
                v = self._check_buffer("exp_avg_sq", state["exp_avg_sq"], p)
                state["step"] += 1
                t = int(state["step"].item())                            # a host tensor: no device sync
                step_size = group["lr"] / (1.0 - beta1 ** t)
                bc2_sqrt = math.sqrt(1.0 - beta2 ** t)
                if self.row_ranges is None or p.dim() != 2:
                    spans = [(0, p.numel())]
                else:
                    width = p.size(1)
                    spans = [(max(lo, 0) * width, min(hi, p.size(0)) * width) for lo, hi in self.row_ranges]
                if any(x.data_ptr() % 4 for x in (p, g, m, v)) or len({x.data_ptr() % 16 for x in (p, g, m, v)}) != 1:
                    raise RuntimeError("gnn_ecommerce_amd.optim.Adam: parameter, gradient and moments must share their "
                                       "alignment within 16 bytes (a view with a storage offset?); use torch.optim.Adam")
                with torch.cuda.device(p.device):
                    for lo, hi in spans:
                        if hi <= lo:
                            continue
                        code = lib.lgc_adam_step(_native.ptr(p) + 4 * lo, _native.ptr(g) + 4 * lo, _native.ptr(m) + 4 * lo,
                                                 _native.ptr(v) + 4 * lo, hi - lo, 1.0 - beta1, beta2, 1.0 - beta2,
                                                 group["eps"], step_size, bc2_sqrt, _native.stream_of(p.device))
                        _native.check(code, "lgc_adam_step")
                # the kernel's write is invisible to autograd's version counter, which recommendK's cache keys on
                torch.autograd.graph.increment_version(p)
        return loss
