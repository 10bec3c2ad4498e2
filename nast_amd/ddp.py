"""Data-parallel gradient averaging for the NAST training step (new: the reference is single-device).

One process per GPU (torchrun), `torch.distributed` backend "nccl" = RCCL over xGMI.  The batch shards
naturally (every op is per-sentence); the only exchange is the gradient all-reduce of
the parameter group being stepped:

  * f_step runs TWO backward passes that accumulate into the same .grad (seq2seq.py:464, :530); the reducer is
    armed only for the last one, so each gradient is reduced once per step, after its final accumulation;
  * gradients are packed into flat buckets (default 16 MiB: F's 62 MB -> 4 buckets, large enough to run at
    link bandwidth on the 7x xGMI links, small enough that the first bucket leaves while backward still runs)
    in the order their autograd hooks fire, and each bucket's all-reduce is launched asynchronously from the
    hook of its last gradient -> communication overlaps the rest of backward;
  * parameters that never receive a gradient (sos_token, transformer.py:41; all of F during d_step) are
    simply never seen by a hook and take no part;
  * the bucket plan is learnt on the first armed step (which reduces after backward, un-overlapped) and must
    be identical on all ranks -- it is, because every rank runs the same graph.

Local losses are divided by the LOCAL batch size (as the reference divides by its batch size); averaging
the gradients over ranks therefore equals the single-GPU gradient on the concatenated batch when all ranks
hold equally many sentences.
"""
import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, params, bucket_bytes=16 << 20, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.bucket_bytes = bucket_bytes
        self.enabled = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.world = dist.get_world_size(group) if self.enabled else 1
        self.armed = False
        self.plan = None          # list of lists of param indices
        self.bucket_of = {}
        self.flat = []            # one flat buffer per bucket
        self._fired = []
        self._count = []
        self._work = []
        self._launched = []
        if self.enabled:
            for i, p in enumerate(self.params):
                p.register_post_accumulate_grad_hook(self._make_hook(i))

    def _make_hook(self, i):
        def hook(_param):
            if not self.armed:
                return
            self._fired.append(i)
            if self.plan is not None and i in self.bucket_of:
                b = self.bucket_of[i]
                self._count[b] += 1
                if self._count[b] == len(self.plan[b]):
                    self._launch(b)
        return hook

    def arm(self):
        """Call right before the backward pass whose gradients are final for this step."""
        if not self.enabled:
            return
        self.armed = True
        self._fired = []
        if self.plan is not None:
            self._count = [0] * len(self.plan)
            self._launched = [False] * len(self.plan)
            self._work = [None] * len(self.plan)

    def _build_plan(self, order):
        plan, cur, size = [], [], 0
        for i in order:
            n = self.params[i].numel() * self.params[i].element_size()
            if cur and size + n > self.bucket_bytes:
                plan.append(cur)
                cur, size = [], 0
            cur.append(i)
            size += n
        if cur:
            plan.append(cur)
        self.plan = plan
        self.bucket_of = {i: b for b, idxs in enumerate(plan) for i in idxs}
        self.flat = [torch.empty(sum(self.params[i].numel() for i in idxs), dtype=self.params[idxs[0]].dtype,
                                 device=self.params[idxs[0]].device) for idxs in plan]
        self._count = [0] * len(plan)
        self._launched = [False] * len(plan)
        self._work = [None] * len(plan)

    def _views(self, b):
        out, off = [], 0
        for i in self.plan[b]:
            n = self.params[i].numel()
            out.append(self.flat[b][off:off + n].view_as(self.params[i]))
            off += n
        return out

    def _launch(self, b):
        grads = [self.params[i].grad for i in self.plan[b]]
        views = self._views(b)
        for v, g in zip(views, grads):
            if g is None:
                v.zero_()
        pairs = [(v, g) for v, g in zip(views, grads) if g is not None]
        if pairs:
            torch._foreach_copy_([v for v, _ in pairs], [g for _, g in pairs])
        self._work[b] = dist.all_reduce(self.flat[b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._launched[b] = True

    def finish(self):
        """Call after that backward: waits for the buckets, writes the averaged gradients back."""
        if not self.enabled:
            return
        self.armed = False
        if self.plan is None:
            seen, order = set(), []
            for i in self._fired:
                if i not in seen:
                    seen.add(i)
                    order.append(i)
            if not order:
                return
            self._build_plan(order)
        for b in range(len(self.plan)):
            if not self._launched[b]:
                self._launch(b)
        inv = 1.0 / self.world
        for b in range(len(self.plan)):
            self._work[b].wait()
            views = self._views(b)
            pairs = [(self.params[i].grad, v) for i, v in zip(self.plan[b], views) if self.params[i].grad is not None]
            if pairs:
                torch._foreach_mul_([v for _, v in pairs], inv)
                torch._foreach_copy_([g for g, _ in pairs], [v for _, v in pairs])
