"""Data-parallel gradient averaging for the NAST training step (new: the reference is single-device).

One process per GPU (torchrun), `torch.distributed` backend "nccl" = RCCL over xGMI.  The batch shards
naturally (every op is per-sentence); the only exchange is the gradient all-reduce of
the parameter group being stepped:

  * f_step runs TWO backward passes that accumulate into the same .grad (seq2seq.py:464, :530); the reducer is
    armed only for the last one, so each gradient is reduced once per step, after its final accumulation;
  * gradients are packed into flat buckets (default 16 MiB: F's 62 MB -> 4 buckets, large enough to run at
    link bandwidth on the 7x xGMI links, small enough that the first bucket leaves while backward still runs)
    in the order they become final, and each bucket's all-reduce is launched asynchronously as soon as its last
    gradient is final -> communication overlaps the rest of backward;
  * a gradient is "final" after its last contribution of the armed backward.  Contributions are either
    autograd's AccumulateGrad (post-accumulate hook) or the in-place accumulation of the HIP backward
    kernels (nast_amd.ops.grad_listener); a parameter can receive several (the token embedding is used by the
    gather, the soft-token GEMM and the eos/unk rows), so the number per parameter is learnt on the first
    armed step of each step variant (`tag`) -- that step reduces after backward, un-overlapped;
  * if a later step produces a contribution after its bucket has left (the graph changed), the step falls
    back to a full synchronous reduce and the plan is re-learnt: never a stale gradient.  That decision (and the one to
    learn a plan) is taken by ALL ranks together: `finish()` all-reduces a 2-word state (needs-redo flag, layout checksum)
    after the buckets, so no rank can issue a different sequence of collectives than its peers; a checksum mismatch (ranks
    whose backward graphs differ) raises instead of mixing gradients;
  * the bucket LAYOUT is a function of the parameter list alone (reverse registration order, as torch DDP does), not of the
    order in which a rank happened to see its gradients complete; the learnt contribution counts only decide when a bucket may
    leave;
  * every wait has a deadline (`NAST_DDP_TIMEOUT_S`, default 120 s): a stuck collective prints which rank waits for which
    bucket of which step variant and raises, instead of hanging until the job's time limit;
  * parameters that never receive a gradient (sos_token, transformer.py:41; all of F during d_step) take no part;
  * DEFAULT since round 2: the buckets are launched from `finish()`, after backward, in layout order, from the thread and
    stream that called it (`GradReducer.OVERLAP = False`).  Launching them from the gradient hooks while backward runs
    (`NAST_DDP_OVERLAP=1`) saves the 62 MB all-reduce's ~1-2 ms per 45 ms step at best (one-rank rehearsal: 46.3 vs 46.6
    ms) and was the path in use when round 1's 8-GPU run hung; with no multi-GPU box to rehearse on, the simple order wins.

Local losses are divided by the LOCAL batch size (as the reference divides by its batch size); averaging
the gradients over ranks therefore equals the single-GPU gradient on the concatenated batch when all ranks
hold equally many sentences.
"""
import datetime
import os
import sys
import zlib

import torch
import torch.distributed as dist

TIMEOUT_S = float(os.environ.get("NAST_DDP_TIMEOUT_S", "120"))

_reducers = []
FORCE_SINGLE = False   # rehearsal aid: reduce even in a process group of one rank (exercises the RCCL path on a one-GPU box)
ENABLED = os.environ.get("NAST_DDP_DISABLE") != "1"      # set False to build a model that takes no part in the process group (tests: single-process reference)


def _listener(p):
    for r in _reducers:
        r._notify_param(p)


class _Plan:
    def __init__(self, counts, buckets):
        self.counts = counts                    # param index -> contributions per armed backward
        self.buckets = buckets                  # list of lists of param indices
        self.bucket_of = {i: b for b, idxs in enumerate(buckets) for i in idxs}
        self.flat = None


class GradReducer:
    # bucket size: 16 MiB when the buckets are to overlap backward; ONE bucket (a single all-reduce per step) in the default
    # after-backward mode, where cutting the gradients buys nothing.  (Rehearsal on one GPU: several bucket all-reduces in
    # flight through a gloo process group stall in gloo's device-buffer path from the second step on, with one rank as with
    # two; one bucket runs.  RCCL with one rank runs either way.)
    BUCKET_BYTES = int(float(os.environ["NAST_DDP_BUCKET_MB"]) * (1 << 20)) if "NAST_DDP_BUCKET_MB" in os.environ else None
    # drain the device before the collectives of the after-backward mode are issued (the host reads the step's losses a
    # moment later anyway): a collective then never queues behind ~40 ms of backward kernels on a shared hardware queue
    SYNC_BEFORE = os.environ.get("NAST_DDP_SYNC", "1") == "1"
    OVERLAP = os.environ.get("NAST_DDP_OVERLAP", "0") == "1"      # launch buckets from the gradient hooks during backward
    # After-backward mode: the .grad buffers of the group ARE slices of one flat buffer (created here, kept by
    # zero_grad(set_to_none=False) and by the in-place gradient kernels), so the step's exchange is ONE all-reduce of that
    # buffer and one scaling kernel -- no packing copies, no per-step plan.  Parameters that never receive a gradient hold
    # zeros there (Adam then leaves them where they are, as it does when their .grad is None).
    ARENA = os.environ.get("NAST_DDP_ARENA", "1") == "1"

    def __init__(self, params, bucket_bytes=None, group=None):
        if bucket_bytes is None:
            bucket_bytes = self.BUCKET_BYTES if self.BUCKET_BYTES is not None else ((16 << 20) if self.OVERLAP else (1 << 62))
        self.params = [p for p in params if p.requires_grad]
        self.index = {id(p): i for i, p in enumerate(self.params)}
        self.group = group
        self.bucket_bytes = bucket_bytes
        self.enabled = ENABLED and dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or FORCE_SINGLE)
        self.world = dist.get_world_size(group) if self.enabled else 1
        self.armed = False
        self.plans = {}
        self.plan = None
        self.tag = None
        self._fired = []
        self._streams = ()
        self.stats = {"learnt": 0, "overlapped": 0, "redone": 0}     # steps by how their reduce went
        self.arena = None
        if self.enabled and not self.OVERLAP and self.ARENA and self.params and all(
                p.is_cuda and p.dtype == torch.float32 and p.device == self.params[0].device for p in self.params):
            self._make_arena()
            self.stats = {"arena": 0}
            return
        if self.enabled:
            from . import ops
            for i, p in enumerate(self.params):
                p.register_post_accumulate_grad_hook(self._make_hook(i))
            _reducers.append(self)
            ops.grad_listener = _listener

    def _make_arena(self):
        offs, off = [], 0
        for p in self.params:
            offs.append(off)
            off += (p.numel() + 63) // 64 * 64                 # every slice starts on a 256-byte boundary
        self.arena = torch.zeros(off, dtype=torch.float32, device=self.params[0].device)
        self._slices = []
        for p, o in zip(self.params, offs):
            g = self.arena[o:o + p.numel()].view_as(p)
            p.grad = g
            self._slices.append(g)

    def _finish_arena(self):
        for p, g in zip(self.params, self._slices):            # (a replaced .grad would silently drop out of the exchange)
            if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                raise RuntimeError("[nast_amd.ddp] a .grad buffer was replaced (zero_grad(set_to_none=True)?): the flat "
                                   "gradient buffer no longer holds this group's gradients")
        cur = torch.cuda.current_stream()
        for st in self._streams:                                # the exchange is issued on the current stream
            if st != cur:
                cur.wait_stream(st)
        if self.SYNC_BEFORE:
            torch.cuda.synchronize()
        self.tag = self.tag or "step"
        self._launched = [True]
        self._wait(dist.all_reduce(self.arena, op=dist.ReduceOp.SUM, group=self.group, async_op=True), "gradient all-reduce")
        if self.world > 1:
            self.arena.mul_(1.0 / self.world)
        self.stats["arena"] += 1

    def _make_hook(self, i):
        def hook(_param):
            self._notify(i)
        return hook

    def _notify_param(self, p):
        i = self.index.get(id(p))
        if i is not None:
            self._notify(i)

    def _notify(self, i):
        if not self.armed:
            return
        self._fired.append(i)
        plan = self.plan
        if plan is None:
            return
        if i not in plan.bucket_of:
            self._dirty = True
            return
        if not self.OVERLAP:
            return                              # everything leaves from finish()
        self._seen[i] = self._seen.get(i, 0) + 1
        b = plan.bucket_of[i]
        if self._launched[b]:
            self._dirty = True                  # a contribution arrived after the bucket left
        elif self._seen[i] == plan.counts[i]:
            self._ready[b] += 1
            # buckets leave strictly in layout order (collectives are matched across ranks by issue order)
            while self._next < len(plan.buckets) and self._ready[self._next] == len(plan.buckets[self._next]):
                self._launch(self._next)
                self._next += 1

    def arm(self, tag="default", streams=()):
        """Call right before the backward pass whose gradients are final for this step.  `streams`: every HIP
        stream that backward's kernels run on, when more than one (a bucket waits for all of them before it leaves)."""
        if not self.enabled:
            return
        if self.arena is not None:
            self.tag, self._streams = tag, tuple(streams)
            return
        self.armed = True
        self.tag = tag
        self._streams = tuple(streams)
        self.plan = self.plans.get(tag)
        self._fired = []
        self._dirty = False
        self._seen = {}
        if self.plan is not None:
            n = len(self.plan.buckets)
            self._ready = [0] * n
            self._launched = [False] * n
            self._work = [None] * n
            self._next = 0

    def _build_plan(self):
        counts, last = {}, {}
        for pos, i in enumerate(self._fired):
            counts[i] = counts.get(i, 0) + 1
            last[i] = pos
        # layout: reverse registration order of the parameters that take part (a function of the model, identical on every
        # rank); gradients become final roughly in that order, so the first bucket still leaves early in backward
        order = sorted(last, reverse=True)
        buckets, cur, size = [], [], 0
        for i in order:
            n = self.params[i].numel() * self.params[i].element_size()
            if cur and size + n > self.bucket_bytes:
                buckets.append(cur)
                cur, size = [], 0
            cur.append(i)
            size += n
        if cur:
            buckets.append(cur)
        plan = _Plan(counts, buckets)
        plan.checksum = zlib.crc32(repr((sorted(counts.items()), buckets)).encode()) & 0x7FFFFF      # exact in fp32
        plan.flat = [torch.empty(sum(self.params[i].numel() for i in idxs), dtype=self.params[idxs[0]].dtype,
                                 device=self.params[idxs[0]].device) for idxs in buckets]
        return plan

    def _views(self, plan, b):
        cache = plan.__dict__.setdefault("views", {})
        if b not in cache:
            cache[b] = self._make_views(plan, b)
        return cache[b]

    def _make_views(self, plan, b):
        out, off = [], 0
        for i in plan.buckets[b]:
            n = self.params[i].numel()
            out.append(plan.flat[b][off:off + n].view_as(self.params[i]))
            off += n
        return out

    def _launch(self, b):
        plan = self.plan
        # the last contribution may come from a node on one stream while earlier ones are still queued on another
        cur = torch.cuda.current_stream() if self._streams else None
        for st in self._streams:
            if st != cur:
                cur.wait_stream(st)
        grads = [self.params[i].grad for i in plan.buckets[b]]
        views = self._views(plan, b)
        for v, g in zip(views, grads):
            if g is None:
                v.zero_()
        pairs = [(v, g) for v, g in zip(views, grads) if g is not None]
        if pairs:
            torch._foreach_copy_([v for v, _ in pairs], [g for _, g in pairs])
        self._work[b] = dist.all_reduce(plan.flat[b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._launched[b] = True

    def _write_back(self, b):
        plan = self.plan
        views = self._views(plan, b)
        pairs = [(self.params[i].grad, v) for i, v in zip(plan.buckets[b], views) if self.params[i].grad is not None]
        if pairs:
            torch._foreach_mul_([v for _, v in pairs], 1.0 / self.world)
            torch._foreach_copy_([g for g, _ in pairs], [v for _, v in pairs])

    def _wait(self, work, what):
        """work.wait() with a deadline and a message that names the rank, the step variant and the bucket."""
        try:
            ok = work.wait(datetime.timedelta(seconds=TIMEOUT_S))
        except Exception as e:                       # gloo raises on timeout; RCCL's watchdog aborts with its own message
            ok, err = False, e
        else:
            err = None
        if ok is False or err is not None:
            rank = dist.get_rank(self.group)
            launched = getattr(self, "_launched", [])
            msg = (f"[nast_amd.ddp] rank {rank}: {what} of step variant '{self.tag}' did not complete within {TIMEOUT_S:.0f} s "
                   f"(buckets launched {sum(bool(x) for x in launched)}/{len(launched)}, stats {self.stats}, "
                   f"bucket sizes {[len(b) for b in self.plan.buckets] if self.plan else None}): {err}")
            print(msg, file=sys.stderr, flush=True)
            raise RuntimeError(msg)

    def _agree(self, need_sync, checksum):
        """All ranks learn whether ANY rank has to (re)do the reduce synchronously, and check that their bucket layouts
        are the same.  Two words, one all-reduce (MAX) + one (MIN); the result is read on the host, where the step reads
        its losses a moment later anyway."""
        t = torch.tensor([float(need_sync), float(checksum), -float(checksum)], device=self.params[0].device)
        self._wait(dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group, async_op=True), "state exchange")
        any_sync, cmax, ncmin = t.tolist()
        if checksum >= 0 and not need_sync and not any_sync and (cmax != checksum or -ncmin != checksum):
            raise RuntimeError(f"[nast_amd.ddp] rank {dist.get_rank(self.group)}: bucket layouts differ between ranks for step "
                               f"variant '{self.tag}' ({checksum} vs {cmax}/{-ncmin}): the ranks' backward graphs are not the same")
        return bool(any_sync)

    def finish(self):
        """Call after that backward: waits for the buckets, writes the averaged gradients back."""
        if not self.enabled:
            return
        if self.arena is not None:
            return self._finish_arena()
        self.armed = False
        local_sync = self.plan is None or self._dirty
        # A rank with a plan sends ALL its buckets, in layout order, even when it already knows that it will have to redo the
        # step (its peers, which may be clean, send theirs: the sequences of collectives must be the same everywhere)
        if self.SYNC_BEFORE and not self.OVERLAP and self.params[0].is_cuda:
            torch.cuda.synchronize()
        if self.plan is not None:
            for b in range(len(self.plan.buckets)):
                if not self._launched[b]:
                    self._launch(b)                     # (buckets whose last gradient did not come during backward)
            for b, w in enumerate(self._work):
                self._wait(w, f"bucket {b}")
        any_sync = self._agree(local_sync, -1 if self.plan is None else self.plan.checksum)
        if not any_sync:
            self.stats["overlapped"] += 1
            for b in range(len(self.plan.buckets)):
                self._write_back(b)
            return
        # some rank has no plan yet or saw a contribution after its bucket had left (the graph differs from the learnt one):
        # EVERY rank re-learns from what it saw in this backward and reduces synchronously -- never a stale gradient
        if self.plan is not None:
            self.stats["redone"] += 1
        self.stats["learnt"] += 1
        if not self._fired:
            # (a rank without any gradient in this step variant: it still has to match its peers' collectives)
            raise RuntimeError("[nast_amd.ddp] a rank produced no gradient in a step variant its peers reduce")
        self.plan = self.plans[self.tag] = self._build_plan()
        n = len(self.plan.buckets)
        self._launched = [False] * n
        self._work = [None] * n
        for b in range(n):
            self._launch(b)
        for b in range(n):
            self._wait(self._work[b], f"bucket {b} (synchronous)")
            self._write_back(b)
        if self._agree(False, self.plan.checksum):
            raise RuntimeError("[nast_amd.ddp] inconsistent state exchange")
