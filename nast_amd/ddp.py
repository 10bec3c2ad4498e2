"""Data-parallel gradient averaging for the NAST training step (new: the reference is single-device).

One process per GPU (torchrun or `bench.py --gpus N`), `torch.distributed` backend "nccl" = RCCL over xGMI.  The batch shards
naturally (every op is per-sentence); the only exchange is the gradient all-reduce of the parameter
group being stepped.

Layout.  The `.grad` buffers of a group ARE slices of one flat buffer, the *arena*, created with the reducer and kept by
`zero_grad(set_to_none=False)` and by the in-place gradient kernels (nast_amd.ops).  The arena is laid out in REVERSE
registration order of the parameters -- the order in which backward finishes them, as torch DDP assumes -- and cut at
parameter boundaries into buckets of <= 16 MiB: a bucket is a contiguous slice, so its all-reduce needs no packing copy,
no write-back and no learnt layout; every rank derives the same cut from the parameter list alone.

Exchange (default under RCCL, `NAST_DDP_OVERLAP=1`).  f_step runs TWO backward passes that accumulate into the same
gradients (seq2seq.py:464, :530); the reducer is armed for the last one only.  A bucket's all-reduce (average) is issued on
the stream, asynchronously, as soon as the last contribution to its last parameter has been enqueued during that
backward, so the exchange of the late layers' gradients runs beside the backward of the early ones.  Buckets leave strictly
in layout order (collectives are matched across ranks by issue order alone).  `finish()` issues whatever is left and makes
the stream wait for the collectives: stream-ordered, no host synchronisation on the RCCL path.

  * A parameter's gradient is "final" after its last contribution of the armed backward.  Contributions are autograd's
    AccumulateGrad (post-accumulate hook) or the in-place accumulation of the HIP backward kernels (ops.grads_written); a
    parameter can receive several (the token embedding: gather, soft-token GEMM, eos / unk rows), so the number per
    parameter is learnt on the first armed step of each step variant (`tag`) -- that step exchanges after backward.
  * Parameters that received no contribution since the last exchange (sos_token, transformer.py:41; the length predictor
    during `use_inference_z=False` steps; all of F during d_step) take no part: a bucket none of whose parameters took part
    is not sent, and `FusedAdam` skips them exactly as it skips `grad is None` (nast_amd.optim, `track_participation`).
  * A contribution that arrives after its bucket has left (the graph differs from the learnt one) never races with the
    collective and never goes stale: the in-place kernels are handed a side buffer instead (ops.sink_hook), which is
    all-reduced by `finish()` and added to the slice; the plan is then re-learnt.  (A late AccumulateGrad contribution has
    already been added to a buffer in flight: that raises.)
  * Every rank checks, one step late and without a host wait, that its peers ran the same plan (a 3-word all-reduce whose
    result is copied to pinned memory); a mismatch raises instead of mixing gradients.
  * `NAST_DDP_OVERLAP=0` (default for any other backend): all buckets leave from `finish()`, after backward.  With gloo and
    device gradients (rehearsals: several ranks sharing one GPU) each bucket is staged through pinned host memory by this
    module with blocking copies; gloo never sees a device buffer (DESIGN.md section 6 for why).
  * Host-blocking waits (gloo) have a deadline (`NAST_DDP_TIMEOUT_S`, default 120 s) and fail with rank, bucket and step
    variant in the message; on RCCL the process group's watchdog (same timeout, bench.py passes it) plays that role.

Local losses are divided by the LOCAL batch size (as the reference divides by its batch size); averaging the gradients
over ranks therefore equals the single-GPU gradient on the concatenated batch when all ranks hold equally many sentences.
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


def _sink_hook(p):
    for r in _reducers:
        buf = r._sink_param(p)
        if buf is not None:
            return buf
    return None


PROF = os.environ.get("NAST_DDP_PROF") == "1"      # host seconds and calls per entry point, printed by bench.py's debug line
prof = {}


def _timed(fn):
    if not PROF:
        return fn
    import functools
    import time

    @functools.wraps(fn)
    def wrap(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            e = prof.setdefault(fn.__name__, [0, 0.0])
            e[0] += 1
            e[1] += time.perf_counter() - t
    return wrap


class _Plan:
    """What one step variant does to the arena: contributions per parameter in the armed backward, and who takes part."""

    def __init__(self, counts, part, bucket_of, n_buckets):
        self.counts = counts                    # param index -> contributions during the armed backward
        self.part = part                        # indices of the parameters that take part in the step (any backward)
        self.need = [0] * n_buckets             # bucket -> parameters that must become final in the armed backward
        self.send = [False] * n_buckets         # bucket -> has a participating parameter
        for i in part:
            self.send[bucket_of[i]] = True
            if counts.get(i, 0) > 0:
                self.need[bucket_of[i]] += 1
        self.checksum = zlib.crc32(repr((sorted(counts.items()), sorted(part))).encode()) & 0x7FFFFF      # exact in fp32


class GradReducer:
    BUCKET_BYTES = int(float(os.environ["NAST_DDP_BUCKET_MB"]) * (1 << 20)) if "NAST_DDP_BUCKET_MB" in os.environ else (16 << 20)
    # launch buckets during backward: None = by backend (RCCL yes, anything else no)
    OVERLAP = {"1": True, "0": False}.get(os.environ.get("NAST_DDP_OVERLAP"))
    CHECK = os.environ.get("NAST_DDP_CHECK", "1") == "1"

    def __init__(self, params, bucket_bytes=None, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.index = {id(p): i for i, p in enumerate(self.params)}
        self.group = group
        self.bucket_bytes = bucket_bytes if bucket_bytes is not None else self.BUCKET_BYTES
        self.enabled = bool(ENABLED and self.params and dist.is_available() and dist.is_initialized()
                            and (dist.get_world_size(group) > 1 or FORCE_SINGLE))
        self.world = dist.get_world_size(group) if self.enabled else 1
        self.armed = False
        self.plans = {}
        self.plan = None
        self.tag = None
        self.arena = None
        self.touched = set()                    # indices that received a contribution since the last finish()
        self.stats = {"learnt": 0, "overlapped": 0, "after_backward": 0, "late": 0}     # steps by how their exchange went
        self._streams = ()
        self._fired = []
        self._late = {}
        self._check = None
        self._pinned = None
        if not self.enabled:
            return
        p0 = self.params[0]
        if not all(p.dtype == p0.dtype and p.device == p0.device for p in self.params):
            raise NotImplementedError("[nast_amd.ddp] one dtype and one device per parameter group")
        self.backend = dist.get_backend(group)
        self.nccl = self.backend == "nccl"
        self.overlap = self.OVERLAP if self.OVERLAP is not None else self.nccl
        # RCCL averages inside the collective; elsewhere sum and scale
        self._op = dist.ReduceOp.AVG if self.nccl else dist.ReduceOp.SUM
        self._scale = 1.0 if self.nccl else 1.0 / self.world
        self._make_arena()
        self._staged = self.arena.is_cuda and not self.nccl
        self._host = None
        from . import ops
        for i, p in enumerate(self.params):
            p.register_post_accumulate_grad_hook(self._make_hook(i))
        _reducers.append(self)
        ops.grad_listener = _listener
        ops.sink_hook = _sink_hook

    # ---- layout -------------------------------------------------------------------------------
    def _make_arena(self):
        order = list(range(len(self.params)))[::-1]             # reverse registration order: backward finishes the last layers first
        offs, off = {}, 0
        self.bucket_of, self.ranges = {}, []
        start, b = 0, 0
        for i in order:
            n = (self.params[i].numel() + 63) // 64 * 64        # every slice starts on a 256-byte boundary
            if off > start and (off + n - start) * self.params[i].element_size() > self.bucket_bytes:
                self.ranges.append((start, off))
                start, b = off, b + 1
            offs[i] = off
            self.bucket_of[i] = b
            off += n
        self.ranges.append((start, off))
        p0 = self.params[0]
        self.arena = torch.zeros(off, dtype=p0.dtype, device=p0.device)
        self._slices = []
        for i, p in enumerate(self.params):
            g = self.arena[offs[i]:offs[i] + p.numel()].view_as(p)
            p.grad = g
            self._slices.append(g)
        self._views = [self.arena[lo:hi] for lo, hi in self.ranges]

    @property
    def n_buckets(self):
        return len(self.ranges)

    def zero_grads(self):
        """One fill for the whole group (what zero_grad(set_to_none=False) does slice by slice)."""
        self.arena.zero_()

    # ---- notifications ------------------------------------------------------------------------
    def _make_hook(self, i):
        def hook(_param):
            self.touched.add(i)
            if self.armed and self.plan is not None and (self._launched[self.bucket_of[i]] or not self.plan.send[self.bucket_of[i]]):
                raise RuntimeError(
                    f"[nast_amd.ddp] rank {dist.get_rank(self.group)}: autograd accumulated into parameter #{i} after its gradient "
                    f"bucket had left (step variant '{self.tag}' differs from the learnt plan); set NAST_DDP_OVERLAP=0 for this model")
            self._notify(i)
        return hook

    def _notify_param(self, p):
        i = self.index.get(id(p))
        if i is not None:
            self._notify(i)

    @_timed
    def _sink_param(self, p):
        """Called by ops before an in-place gradient kernel is enqueued: the buffer it must add into when that is not the
        parameter's arena slice (its bucket is already on the wire), else None."""
        i = self.index.get(id(p))
        if i is None:
            return None
        self.touched.add(i)
        if not (self.armed and self.plan is not None):
            return None
        b = self.bucket_of[i]
        if self._launched[b] or not self.plan.send[b]:
            if i not in self._late:
                self._late[i] = torch.zeros_like(self._slices[i])
            return self._late[i]
        return None

    @_timed
    def _notify(self, i):
        if not self.armed:
            return
        self._fired.append(i)
        plan = self.plan
        if plan is None or not self.overlap or i in self._late:
            return
        self._seen[i] = self._seen.get(i, 0) + 1
        if self._seen[i] == plan.counts.get(i, 0):
            self._ready[self.bucket_of[i]] += 1
            self._advance()

    def _advance(self):
        """Launch, in layout order, every bucket whose participating parameters are all final."""
        plan = self.plan
        while self._next < self.n_buckets and self._ready[self._next] >= plan.need[self._next]:
            if plan.send[self._next]:
                self._launch(self._next)
            self._next += 1

    # ---- the step's protocol ------------------------------------------------------------------
    @_timed
    def arm(self, tag="default", streams=()):
        """Call right before the backward pass whose gradients are final for this step.  `streams`: every HIP stream that
        backward's kernels run on, when more than one (a bucket waits for all of them before it leaves)."""
        if not self.enabled:
            return
        for p, g in zip(self.params, self._slices):            # (a replaced .grad would silently drop out of the exchange)
            if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                raise RuntimeError("[nast_amd.ddp] a .grad buffer was replaced (zero_grad(set_to_none=True)?): the flat "
                                   "gradient buffer no longer holds this group's gradients")
        self.armed = True
        self.tag = tag
        self._streams = tuple(streams)
        self.plan = self.plans.get(tag)
        self._fired = []
        self._late = {}
        self._seen = {}
        n = self.n_buckets
        self._ready = [0] * n
        self._launched = [False] * n
        self._work = [None] * n
        self._next = 0
        if self.plan is not None and self.overlap:
            self._advance()                                     # buckets that backward #1 alone filled leave now

    @_timed
    def _launch(self, b):
        if self.arena.is_cuda:
            # the last contribution may come from a node on one stream while earlier ones are still queued on another
            cur = torch.cuda.current_stream()
            for st in self._streams:
                if st != cur:
                    cur.wait_stream(st)
        buf = self._views[b]
        if self._staged:
            # rehearsal backends (gloo) with device gradients: staged through pinned host memory HERE, with plain blocking
            # copies, instead of handing gloo the device buffer (ProcessGroupGloo's own device path stops
            # completing on this image from its second round on, with or without this library in the process)
            if self._host is None:
                self._host = [torch.empty(hi - lo, dtype=self.arena.dtype, pin_memory=True) for lo, hi in self.ranges]
            buf = self._host[b]
            buf.copy_(self._views[b])
        self._work[b] = dist.all_reduce(buf, op=self._op, group=self.group, async_op=True)
        self._launched[b] = True

    def _wait(self, work, what):
        """RCCL: the current stream waits for the collective (no host wait; the group's watchdog has the deadline).
        Otherwise work.wait() with a deadline and a message that names the rank, the step variant and the bucket."""
        if self.nccl:
            work.wait()
            return
        try:
            ok = work.wait(datetime.timedelta(seconds=TIMEOUT_S))
            err = None
        except Exception as e:                       # gloo raises on timeout
            ok, err = False, e
        if ok is False or err is not None:
            rank = dist.get_rank(self.group)
            msg = (f"[nast_amd.ddp] rank {rank}: {what} of step variant '{self.tag}' did not complete within {TIMEOUT_S:.0f} s "
                   f"(buckets launched {sum(self._launched)}/{self.n_buckets}, stats {self.stats}): {err}")
            print(msg, file=sys.stderr, flush=True)
            raise RuntimeError(msg)

    def _verify_previous(self, block=False):
        """The state exchange of an earlier step, once its copy has landed in pinned memory: every rank ran the same plan."""
        if self._check is None:
            return
        ev, host, mine, tag = self._check
        if ev is not None and not block and not ev.query():
            return
        if ev is not None and block:
            ev.synchronize()
        self._check = None
        cmax, ncmin = host.tolist()
        if cmax != mine or -ncmin != mine:
            raise RuntimeError(f"[nast_amd.ddp] rank {dist.get_rank(self.group)}: the ranks ran different gradient exchanges in step "
                               f"variant '{tag}' (state {mine} here, {-ncmin}..{cmax} across ranks): their backward graphs differ")

    @_timed
    def _exchange_state(self, state):
        if not self.CHECK:
            return
        self._verify_previous(block=True)                       # at most one in flight
        if self.nccl:
            t = torch.empty(2, device=self.arena.device)        # (filled by kernels: a host tensor would cost a blocking copy)
            t[0].fill_(float(state))
            t[1].fill_(-float(state))
        else:
            t = torch.tensor([float(state), -float(state)])
        w = dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group, async_op=True)
        if self.nccl:
            w.wait()
            if self._pinned is None:
                self._pinned = torch.empty(2, pin_memory=True)
            host = self._pinned
            host.copy_(t, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._check = (ev, host, float(state), self.tag)
        else:
            self._wait(w, "state exchange")
            self._check = (None, t, float(state), self.tag)
            self._verify_previous()

    @_timed
    def finish(self):
        """Call after that backward: issues what is left, makes the stream (or, without RCCL, the host) wait for the
        collectives; afterwards every participating .grad holds the average over ranks."""
        if not self.enabled:
            return
        self.armed = False
        self._verify_previous()
        plan = self.plan
        learn = plan is None
        if learn:
            counts = {}
            for i in self._fired:
                counts[i] = counts.get(i, 0) + 1
            plan = _Plan(counts, set(self.touched), self.bucket_of, self.n_buckets)
        for b in range(self._next, self.n_buckets):             # (every bucket, in the learning step and without overlap)
            if plan.send[b] and not self._launched[b]:
                self._launch(b)
        self._next = self.n_buckets
        # parameters outside the plan that an earlier (unarmed) backward of this step wrote into their slices
        stray = self.touched - plan.part
        for b in sorted({self.bucket_of[i] for i in stray}):
            if not self._launched[b]:
                self._launch(b)
        for b, w in enumerate(self._work):
            if w is not None:
                self._wait(w, f"bucket {b}")
                if self._staged:
                    self._views[b].copy_(self._host[b])
        if self._scale != 1.0:
            lo = None
            for b in range(self.n_buckets + 1):                 # one scaling kernel per run of consecutive sent buckets
                sent = b < self.n_buckets and self._launched[b]
                if sent and lo is None:
                    lo = self.ranges[b][0]
                if not sent and lo is not None:
                    self.arena[lo:self.ranges[b - 1][1]].mul_(self._scale)
                    lo = None
        # contributions that came after their bucket had left: reduced on their own, added to the slice
        for i in sorted(self._late):
            buf = self._late[i].cpu() if self._staged else self._late[i]
            self._wait(dist.all_reduce(buf, op=self._op, group=self.group, async_op=True), f"late gradient #{i}")
            self._slices[i].add_(buf.to(self.arena.device), alpha=self._scale)
        unplanned = bool(self._late) or bool(stray)
        if learn:
            self.plans[self.tag] = plan
            self.stats["learnt"] += 1
        elif unplanned:
            # re-learn from what this step did (every rank sees the same graph, hence takes the same decision)
            counts = dict(plan.counts)
            seen = {}
            for i in self._fired:
                seen[i] = seen.get(i, 0) + 1
            for i, c in seen.items():
                counts[i] = max(counts.get(i, 0), c)
            self.plans[self.tag] = _Plan(counts, plan.part | self.touched, self.bucket_of, self.n_buckets)
            self.stats["late"] += 1
        else:
            self.stats["overlapped" if self.overlap else "after_backward"] += 1
        self._exchange_state(plan.checksum * 2 + (1 if unplanned else 0))
        self._late = {}
        self.touched = set()
