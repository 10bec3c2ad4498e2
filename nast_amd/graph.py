"""HIP-graph replay of the generator and discriminator training steps for small batches.

At the reference's own batch size (64 sentences, run.py:70) one f_step is ~2000 kernel launches of a few microseconds each:
the GPU finishes them faster than Python can issue them, and the step time is the host's.  `StepGraphs` captures the whole
step -- NoiseLayer, the three generator passes with the greedy decode, the discriminator, both backward passes, the
gradient clip and Adam -- into one HIP graph per (batch size, temperature, cyc_rec_enable, use_inference_z) and replays
it: one launch per step.

What makes the step capturable (all of it switched by ops.fixed_width, nothing else changes):
  * shapes: every tensor keeps its maximum width (ops.FIXED_WIDTH); the batch is copied into static buffers of width
    max_length, padded with <pad>; lengths mask the padding exactly as they do in the eager step;
  * no read-backs: the decode runs without its early exit, widths are not trimmed, losses are read after the replay;
  * randomness: NoiseLayer and the Gumbel noise draw through torch's generator, whose state a captured graph advances on
    every replay (ops.GRAPH_NOISE);
  * Adam: the step count lives on the device (FusedAdam.device_step_counter), the kernel reads its bias corrections there.
The first `WARM` steps of a key run eagerly in the same fixed-width mode (they are real training steps; they also create
the gradient buffers, optimizer state and pointer tables the capture relies on), the next call captures and replays.

Single process only (a captured collective would need RCCL's own capture support); the packed large-batch step is not
captured: there the GPU is the bound, and the packed layout needs the row totals on the host.
"""
import torch

from . import ops
from .utils import Storage


class _Entry:
    __slots__ = ("seen", "graph", "out", "sent", "length", "domain", "incoming")


def _hyper(opt):
    """The optimizer's hyper-parameters travel to nast_multi_adam_f32 as host scalars, i.e. a captured step holds them as
    constants: they are part of the key, so a changed lr / betas / eps / weight decay (a schedule, a restore()) captures a
    new graph instead of silently replaying the old values."""
    return tuple((float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]))
                 for g in opt.param_groups)


class StepGraphs:
    WARM = 2
    MAX_KEYS = 16        # a temperature schedule that changes every step never repeats a key: such keys stay eager and are dropped

    def __init__(self, trainer):
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            raise NotImplementedError("graph_steps: single process only")
        self.tr = trainer
        self.entries = {}
        self.stats = {"eager": 0, "captured": 0, "replayed": 0}

    def _entry(self, key, B, dev):
        e = self.entries.get(key)
        if e is None:
            L = self.tr.args.max_length
            e = _Entry()
            e.seen, e.graph, e.out = 0, None, None
            e.sent = torch.zeros(B, L, dtype=torch.int64, device=dev)
            e.length = torch.zeros(B, dtype=torch.int64, device=dev)
            e.domain = torch.zeros(B, dtype=torch.int64, device=dev)
            e.incoming = Storage()
            e.incoming.data = Storage()
            e.incoming.data.sent, e.incoming.data.sent_length, e.incoming.data.domain = e.sent, e.length, e.domain
            e.incoming.data.batch_size, e.incoming.data.seqlen = B, L
            if len(self.entries) >= self.MAX_KEYS:                  # forget the oldest key that never reached a capture
                for k, old in self.entries.items():
                    if old.graph is None:
                        del self.entries[k]
                        break
                else:
                    raise RuntimeError(f"graph_steps: more than {self.MAX_KEYS} captured step variants alive")
            self.entries[key] = e
        return e

    def _load(self, e, incoming):
        d = incoming.data
        w = d.sent.shape[1]
        pad = self.tr.param.volatile.dm.pad_id
        if w > e.sent.shape[1]:
            raise ValueError(f"graph_steps: batch width {w} exceeds max_length {e.sent.shape[1]}")
        e.sent.fill_(pad)
        e.sent[:, :w].copy_(d.sent, non_blocking=True)
        e.length.copy_(d.sent_length, non_blocking=True)
        e.domain.copy_(d.domain, non_blocking=True)

    @staticmethod
    def _pack(out):
        z = None
        vals = []
        for x in out:
            if x is None:
                if z is None:
                    z = next(v for v in out if v is not None).new_zeros(())
                x = z
            vals.append(x.detach().reshape(()).float())
        return torch.stack(vals)

    def _run(self, key, incoming, optimizer, step):
        """step(incoming, capturing) -> tuple of 0-d tensors / None.  Eager (fixed width) for the first WARM calls of a key,
        then captured once and replayed."""
        sent = incoming.data.sent
        e = self._entry(key, sent.shape[0], sent.device)
        self._load(e, incoming)
        if e.seen < self.WARM:
            e.seen += 1
            self.stats["eager"] += 1
            with ops.packed_rows(False), ops.fixed_width():
                out = self._pack(step(e.incoming, False))
            return tuple(out.tolist())
        if e.graph is None:
            optimizer.device_step_counter()               # created (and validated) outside the capture
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                with ops.packed_rows(False), ops.fixed_width():
                    e.out = self._pack(step(e.incoming, True))
            e.graph = g
            self.stats["captured"] += 1
        e.graph.replay()
        optimizer.note_replayed_step()
        self.stats["replayed"] += 1
        return tuple(e.out.tolist())

    def f_step(self, incoming, temperature, cyc_rec_enable, use_inference_z):
        tr = self.tr
        key = ("f", incoming.data.sent.shape[0], float(temperature), bool(cyc_rec_enable), bool(use_inference_z)) + _hyper(tr.optimizer_F)
        return self._run(key, incoming, tr.optimizer_F,
                         lambda inc, cap: tr._f_step(inc, temperature, cyc_rec_enable, use_inference_z=use_inference_z, capturing=cap))

    def d_step(self, incoming, temperature, use_inference_z):
        tr = self.tr
        key = ("d", incoming.data.sent.shape[0], float(temperature), bool(use_inference_z)) + _hyper(tr.optimizer_D)
        return self._run(key, incoming, tr.optimizer_D,
                         lambda inc, cap: (tr._d_step(inc, temperature, use_inference_z=use_inference_z, capturing=cap),))[0]
