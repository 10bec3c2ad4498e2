"""Gradient-norm clip + Adam for a whole parameter group in three launches.

The reference steps each group with `clip_grad_norm_(params, 5)` followed by `torch.optim.Adam.step()`
(seq2seq.py:35-36, :536-537, :637-638): ~340 tensors -> several hundred tiny launches per step.  `FusedAdam` is a
`torch.optim.Adam` subclass (same constructor, same `state_dict()` layout: step / exp_avg / exp_avg_sq per parameter, so
reference checkpoints load) whose `clip_and_step` runs `nast_multi_sqnorm_f32` + `nast_multi_adam_f32` over device
arrays of tensor pointers.  Arithmetic follows torch's single-tensor Adam operation by operation (optim.hip).

Supported: amsgrad=False, maximize=False, one learning rate per group (what the reference uses); anything else raises.
CPU parameters (host-only unit tests) take torch's own path.
"""
import math

import torch
from torch import optim
from torch.nn.utils import clip_grad_norm_

from . import ops


class FusedAdam(optim.Adam):
    CHUNK = 16384            # elements per workgroup of the multi-tensor kernels

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, **kw):
        if kw.get("amsgrad") or kw.get("maximize") or kw.get("capturable") or kw.get("differentiable"):
            raise NotImplementedError("FusedAdam: amsgrad / maximize / capturable / differentiable are not supported")
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self._tables = {}
        self._dev_step = None       # float64 [1] on the device: the common step count, kept for HIP-graph replays
        self._track = False
        self.grad_arena = None      # nast_amd.ddp.GradReducer whose flat buffer holds this optimizer's .grad slices

    # -- participation ---------------------------------------------------------------------------
    def track_participation(self):
        """Step only the parameters that received a gradient contribution since the last zero_grad().  torch.optim.Adam
        skips `grad is None` (what zero_grad() leaves behind in the reference, seq2seq.py:435); here the gradient buffers
        persist zero-filled (the HIP kernels accumulate in place, data parallelism keeps them in one flat buffer), so the
        contributions themselves are tracked: nast_amd.ops marks every parameter it hands a gradient sink, a
        post-accumulate hook marks the ones autograd writes.  Without this a parameter that sits out a step (the length
        predictor while `use_inference_z=False`, sos_token under data parallelism) would advance its step count, keep
        moving on its momentum and take weight decay."""
        if self._track:
            return
        self._track = True

        def mark(p):
            p._nast_touched = True
        for g in self.param_groups:
            for p in g["params"]:
                p._nast_touched = False
                p.register_post_accumulate_grad_hook(mark)

    def _takes_part(self, p):
        return p.grad is not None and (not self._track or getattr(p, "_nast_touched", False))

    # -- step count on the device (nast_amd/graph.py) ------------------------------------------
    def device_step_counter(self):
        """The optimizer's step count as a device tensor, created on first use.  A captured step reads its bias corrections
        from it (1 - beta ** step is computed on the device, in float64 like the host computes it) and increments it, so a
        replay advances Adam without the host; eager steps keep it in line.  Requires one common count for all parameters."""
        if self._dev_step is None:
            counts = {float(st["step"]) for st in self.state.values() if "step" in st}
            if len(counts) > 1:
                raise NotImplementedError("FusedAdam: parameters with different step counts cannot share a captured step")
            p0 = self.param_groups[0]["params"][0]
            self._dev_step = torch.full((1,), counts.pop() if counts else 0.0, dtype=torch.float64, device=p0.device)
        return self._dev_step

    def note_replayed_step(self):
        """Host-side bookkeeping after a graph replay ran one optimizer step on the device: the per-parameter `step`
        entries of state_dict() (CPU tensors, torch.optim.Adam's layout)."""
        steps = [st["step"] for st in self.state.values() if "step" in st]
        if steps:
            torch._foreach_add_(steps, 1.0)

    def zero_grad(self, set_to_none=True):
        """set_to_none=False (what the in-place gradient kernels need) zeroes every buffer in a few multi-tensor launches
        instead of one fill per parameter (263 launches per step for the generator group)."""
        if self._track:
            for g in self.param_groups:
                for p in g["params"]:
                    p._nast_touched = False
        if set_to_none:
            return super().zero_grad(set_to_none=True)
        if self.grad_arena is not None and self.grad_arena.arena is not None:
            return self.grad_arena.zero_grads()              # the slices of one flat buffer: one fill
        grads = [p.grad for g in self.param_groups for p in g["params"] if p.grad is not None]
        for gr in grads:
            if gr.grad_fn is not None or gr.requires_grad:
                return super().zero_grad(set_to_none=False)
        if grads:
            torch._foreach_zero_(grads)

    # -- pointer tables (rebuilt only when a tensor moved) -----------------------------------
    def _table(self, gi, group, plist):
        key = tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr())
                    for p in plist)
        ent = self._tables.get(gi)
        if ent is None or ent[0] != key:
            dev = plist[0].device
            cols = list(zip(*key))
            ptrs = torch.tensor(cols, dtype=torch.int64).to(dev)                  # [4, n] device pointers
            numels = [p.numel() for p in plist]
            sizes = torch.tensor(numels, dtype=torch.int64).to(dev)
            sq = torch.zeros(1, device=dev, dtype=torch.float32)
            # work table: chunks of CHUNK consecutive elements of one tensor, one workgroup each
            ct = [t for t, n in enumerate(numels) for _ in range(0, n, self.CHUNK)]
            co = [o for n in numels for o in range(0, n, self.CHUNK)]
            chunks = (torch.tensor(ct, dtype=torch.int32).to(dev), torch.tensor(co, dtype=torch.int64).to(dev), len(ct))
            ent = (key, ptrs, sizes, sq, chunks)
            self._tables[gi] = ent
        return ent

    @torch.no_grad()
    def clip_and_step(self, max_norm=None, capturing=False):
        """clip_grad_norm_(all parameters of the optimizer, max_norm) (skipped when None), then Adam.  Returns the
        total gradient norm as a 0-d device tensor (None without clipping).
        capturing=True (under HIP-graph capture): the bias corrections come from the device step counter, which the
        captured work increments; the host-side counts are left to note_replayed_step()."""
        plists = [[p for p in g["params"] if self._takes_part(p)] for g in self.param_groups]
        flat = [p for pl in plists for p in pl]
        if not flat:
            return None
        if not flat[0].is_cuda:
            norm = clip_grad_norm_(flat, max_norm) if max_norm is not None else None
            idle = [(p, p.grad) for g in self.param_groups for p in g["params"] if p.grad is not None and not self._takes_part(p)]
            for p, _ in idle:                               # torch's own step: hide the gradients of those that sat out
                p.grad = None
            try:
                super().step()
            finally:
                for p, gr in idle:
                    p.grad = gr
            return norm
        for p in flat:
            if p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous() or p.grad.is_sparse:
                raise NotImplementedError("FusedAdam: contiguous dense fp32 parameters and gradients only")
            st = self.state[p]
            if len(st) and st["exp_avg"].device != p.device:
                raise RuntimeError(f"FusedAdam: optimizer state on {st['exp_avg'].device} but the parameter on {p.device} "
                                   "(load_state_dict before the model was moved?): move the state with the parameters")
            if len(st) == 0:
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        stream = ops._stream()
        tabs = [self._table(gi, g, pl) if pl else None for gi, (g, pl) in enumerate(zip(self.param_groups, plists))]
        sq = None
        if max_norm is not None:
            sq = next(t for t in tabs if t is not None)[3]
            sq.zero_()
            for t in tabs:
                if t is not None:
                    _, ptrs, sizes, _, (ct, co, nc) = t
                    ops.call("nast_multi_sqnorm_f32", ptrs[1].data_ptr(), sizes.data_ptr(), sizes.numel(), sq.data_ptr(),
                             ct.data_ptr(), co.data_ptr(), nc, self.CHUNK, stream)
        for gi, (g, pl, t) in enumerate(zip(self.param_groups, plists, tabs)):
            if t is None:
                continue
            # torch.optim.Adam keeps one step count per parameter (a parameter that gets its first gradient late -- e.g. the
            # length predictor while `pre_z_batch` steps run without it, seq2seq.py:241-247 -- has a smaller one, and so have
            # restored states): one launch per distinct count, each with its own bias corrections.  Normally there is one.
            by_step = {}
            for p in pl:
                by_step.setdefault(float(self.state[p]["step"]), []).append(p)
            b1, b2 = g["betas"]
            if capturing:
                if len(by_step) != 1 or len([x for x in tabs if x is not None]) != 1:
                    raise NotImplementedError("FusedAdam: a captured step needs one parameter group with one step count")
                ds = self.device_step_counter()
                ds.add_(1.0)
                bias = torch.cat((1.0 - torch.pow(b1, ds), 1.0 - torch.pow(b2, ds))).float()
                _, ptrs, sizes, _, (ct, co, nc) = t
                ops.call("nast_multi_adam_f32", ptrs[0].data_ptr(), ptrs[1].data_ptr(), ptrs[2].data_ptr(), ptrs[3].data_ptr(),
                         sizes.data_ptr(), sizes.numel(), sq.data_ptr() if sq is not None else None,
                         float(max_norm) if max_norm is not None else 0.0, float(g["lr"]), float(b1), float(b2), float(g["eps"]),
                         float(g["weight_decay"]), 1.0, 1.0, bias.data_ptr(), ct.data_ptr(), co.data_ptr(), nc, self.CHUNK, stream)
                self._capture_keepalive = bias
                continue
            if self._dev_step is not None and gi == 0:
                self._dev_step.add_(1.0)
            for si, (step0, sub) in enumerate(sorted(by_step.items())):
                step = step0 + 1
                for p in sub:
                    self.state[p]["step"] += 1
                _, ptrs, sizes, _, (ct, co, nc) = t if len(by_step) == 1 else self._table((gi, "step", si), g, sub)
                ops.call("nast_multi_adam_f32", ptrs[0].data_ptr(), ptrs[1].data_ptr(), ptrs[2].data_ptr(), ptrs[3].data_ptr(),
                         sizes.data_ptr(), sizes.numel(), sq.data_ptr() if sq is not None else None,
                         float(max_norm) if max_norm is not None else 0.0, float(g["lr"]), float(b1), float(b2), float(g["eps"]),
                         float(g["weight_decay"]), 1.0 - b1 ** step, 1.0 - b2 ** step, None, ct.data_ptr(), co.data_ptr(), nc,
                         self.CHUNK, stream)
        return sq.sqrt().reshape(()) if sq is not None else None
