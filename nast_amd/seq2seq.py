"""Trainer of NAST-StyTrans on the HIP modules: the reference's `Seq2seq` surface for the generator /
discriminator training steps (reference: styletransformer/seq2seq.py, cited as seq2seq.py:line).

In scope: __init__ (optimiser split), f_step, d_step, approx, get_lengths, pretrain, train, calc_temperature,
_preprocess_batch, get_next_batch, get_next_mix_batch.  Out of scope (needs cotk metrics + the RoBERTa
classifier): evaluate / train_process / test_process -> NotImplementedError.

Data parallelism (new; the reference is single-device): when torch.distributed is initialised, gradients of the
parameter group being stepped are averaged across ranks by nast_amd.ddp.GradReducer (bucketed all-reduce on
RCCL, fired once per step during the LAST backward so that it overlaps with it) before clip + Adam.
"""
import contextlib

import numpy as np
import torch
from torch import nn, optim

from . import ops
from .ddp import GradReducer
from .optim import FusedAdam
from .transformer import Network
from .utils import Storage, BaseModel, LongTensor, cuda, generateMask
from .utils.gumbel import gumbel_soft_and_index


_RESERVED = {}      # (device type, index) -> second stream created by reserve_streams()


def reserve_streams(device=None):
    """Create the step's second HIP stream NOW and put a first launch on it.  A process has few hardware queues
    (GPU_MAX_HW_QUEUES, 4 by default) and a stream is bound to one at its first use; streams that come later share.
    RCCL's initialisation takes several, after which a new stream lands on the main stream's queue and the two passes
    of f_step serialise (measured: 46.3 -> 50.8 ms/step).  Call this before torch.distributed.init_process_group."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    key = (device.type, device.index)
    if key not in _RESERVED:
        torch.zeros(1, device=device)                      # main stream first
        s = torch.cuda.Stream(device=device)
        with torch.cuda.stream(s):
            torch.zeros(1, device=device)
        s.synchronize()
        _RESERVED[key] = s
    return _RESERVED[key]


def item(x, default_val=0):
    return x.item() if x is not None else default_val


class Seq2seq(BaseModel):
    def __init__(self, param):
        args = param.args
        net = Network(param)
        if args.radam:
            raise NotImplementedError("--radam: utils/optimizer.py is out of scope (SURVEY.md section 2.1 row 8)")
        # torch.optim.Adam subclasses (same state_dict): clip + step of a whole group in three launches
        self.optimizer_F = FusedAdam(net.get_parameters_by_name("F"), lr=args.lr_F, weight_decay=args.L2)
        self.optimizer_D = FusedAdam(net.get_parameters_by_name("D"), lr=args.lr_D, weight_decay=args.L2)
        for opt, mod in ((self.optimizer_F, net.model_F), (self.optimizer_D, net.model_D)):
            # the reference clips model.parameters() and steps the optimizer's group (seq2seq.py:536-537): the same tensors
            assert {id(p) for g in opt.param_groups for p in g["params"]} == {id(p) for p in mod.parameters()}
        super().__init__(param, net, {"optimizer_F": self.optimizer_F, "optimizer_D": self.optimizer_D})
        self.reducer_F = GradReducer(list(self.net.model_F.parameters()))
        self.reducer_D = GradReducer(list(self.net.model_D.parameters()))
        for opt, red in ((self.optimizer_F, self.reducer_F), (self.optimizer_D, self.reducer_D)):
            opt.track_participation()          # a parameter that sits out a step is skipped, as Adam skips `grad is None`
            opt.grad_arena = red if red.enabled else None

    overlap_passes = True      # f_step: run the transfer pass on a second stream beside pass 1 (set False to serialise)
    _streams = None

    # tokens per step from which the second stream is used: always with the length predictor (its decode hook pays even at
    # 64 sentences: 24.8 -> 22.9 ms), from 4096 tokens without it (host-launch bound below).  An int sets both.
    overlap_min_tokens = {"learnable": 0, "simple": 4096}
    merge_generate_max = 1024   # d_step: generate both styles as one batch while 2B stays on the decode's small-M path

    def _overlap_threshold(self):
        t = self.overlap_min_tokens
        return t["learnable" if self.args.use_learnable else "simple"] if isinstance(t, dict) else t

    def _side_stream(self, device):
        if self._streams is None:
            self._streams = {}
        key = (device.type, device.index)
        if key not in self._streams:
            self._streams[key] = _RESERVED.get(key) or torch.cuda.Stream(device=device)
            # parameters that still go through AccumulateGrad (main stream) receive gradients from side-stream nodes on
            # purpose; autograd orders the two streams itself and only warns about the extra synchronisation
            quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
            if quiet is not None:
                quiet(False)
        # lazily created .grad buffers are zero-filled on one stream and added to from both (ops._sink)
        ops.peer_streams = (torch.cuda.current_stream(device), self._streams[key])
        return self._streams[key]

    # ---- batches (seq2seq.py:108-164) ---------------------------------------------------------
    def _preprocess_batch(self, data0, data1):
        incoming = Storage()
        incoming.data = data = Storage()
        empty = {"sent": np.zeros((0, 1), dtype=int), "sent_length": []}
        data0 = Storage(data0 if data0 is not None else empty)
        data1 = Storage(data1 if data1 is not None else empty)
        n0, n1 = data0.sent.shape[0], data1.sent.shape[0]
        data.batch_size = n0 + n1
        data.seqlen = max(data0.sent.shape[1], data1.sent.shape[1]) - 1
        sent = np.zeros((data.batch_size, data.seqlen), dtype=np.int64)
        if n0:
            sent[:n0, :data0.sent.shape[1] - 1] = np.asarray(data0.sent)[:, 1:]      # remove <go>
        if n1:
            sent[n0:, :data1.sent.shape[1] - 1] = np.asarray(data1.sent)[:, 1:]
        data.sent = LongTensor(sent)
        lens = np.concatenate([data0.sent_length, data1.sent_length], axis=0).astype(int)
        data.sent_length = LongTensor(lens) - 1
        data.domain = LongTensor([0] * n0 + [1] * n1)
        return incoming

    def get_next_batch(self, dm, key, domain, restart=True, raw=False):
        data = dm.get_next_batch(key + "_" + str(domain))
        if data is None:
            if not restart:
                return None
            dm.restart(key + "_" + str(domain))
            return self.get_next_batch(dm, key, domain, False, raw)
        if raw:
            return data
        return self._preprocess_batch(data, None) if domain == 0 else self._preprocess_batch(None, data)

    def get_next_mix_batch(self, dm, key):
        return self._preprocess_batch(self.get_next_batch(dm, key, 0, raw=True), self.get_next_batch(dm, key, 1, raw=True))

    # ---- schedules (seq2seq.py:166-280) ----------------------------------------------------------
    def calc_temperature(self, temperature_config, step):
        num = len(temperature_config)
        for i in range(num):
            t_a, s_a = temperature_config[i]
            if i == num - 1:
                return t_a
            t_b, s_b = temperature_config[i + 1]
            if s_a <= step < s_b:
                k = (step - s_a) / (s_b - s_a)
                return (1 - k) * t_a + k * t_b

    def pretrain(self, batch_num):
        self.net.model_F.train(); self.net.model_D.train()
        args, dm = self.param.args, self.param.volatile.dm
        dm.restart('train_0', args.batch_size); dm.restart('train_1', args.batch_size)
        hist = []
        for i in range(batch_num):
            self.now_batch += 1
            incoming = self.get_next_mix_batch(dm, 'train')
            hist.append(self.f_step(incoming, temperature=1.0, cyc_rec_enable=False, use_inference_z=True))
            if (i + 1) % 10 == 0:
                m = np.mean(np.array(hist, dtype=float), 0)
                hist = []
                print('[iter: {}] slf_loss:{:.4f}, rec_loss:{:.4f}, slf_length_loss:{:.4f}'.format(i + 1, m[0], m[1], m[3]))

    def train(self, batch_num):
        self.net.model_F.train(); self.net.model_D.train()
        args, dm = self.param.args, self.param.volatile.dm
        dm.restart('train_0', args.batch_size); dm.restart('train_1', args.batch_size)
        d_hist, f_hist = [], []
        for _ in range(batch_num):
            self.now_batch += 1
            temperature = self.calc_temperature(args.temperature_config, self.now_batch)
            use_z = (self.now_batch - args.F_pretrain_iter) > args.pre_z_batch
            for _ in range(args.iter_D):
                d_hist.append(self.d_step(self.get_next_mix_batch(dm, 'train'), temperature=temperature, use_inference_z=use_z))
            for _ in range(args.iter_F):
                f_hist.append(self.f_step(self.get_next_mix_batch(dm, 'train'), temperature=temperature, use_inference_z=use_z))
            if self.now_batch % args.log_steps == 0:
                f = np.mean(np.array(f_hist, dtype=float), 0)
                print(('[iter {}] d_adv_loss: {:.4f}  f_slf_loss: {:.4f}  f_cyc_loss: {:.4f}  f_adv_loss: {:.4f}  '
                       'f_slf_length_loss: {:.4f}  f_cyc_length_loss: {:.4f}  temp: {:.4f}  f_slf_gen_error: {:.4f} '
                       'f_cyc_gen_error: {:.4f}').format(self.now_batch, np.mean(d_hist), f[0], f[1], f[2], f[3], f[4],
                                                          temperature, f[5], f[6]))

    def transfer(self, incoming, domain=None, temperature=1.0, *, use_inference_z=True):
        """The generator's share of `evaluate` (seq2seq.py:303-339): greedy style transfer of one batch, without the cotk
        metrics and the RoBERTa classifier around it.
        incoming.data.{sent [B,L], sent_length [B]}; domain = source style of the batch (int) or None to read
        incoming.data.domain.  Returns a Storage with
          gen [B,T] int64     argmax tokens of the transferred sentences (`rev_log_probs.argmax(-1)`, :324)
          gen_lengths [B]     generated lengths (learnable alignment) or None
          allpos              list of B int64 arrays `pos[1:glen+1]` (:337-338) or None
          length_delta        mean(gen_lengths - inp_lengths) (:343) or None
        Runs the padded execution, so every position holds the reference's value."""
        inp_tokens, inp_lengths = incoming.data.sent, incoming.data.sent_length
        raw_styles = torch.full_like(inp_tokens[:, 0], domain) if domain is not None else incoming.data.domain
        was_training = self.net.model_F.training
        self.net.model_F.eval()
        try:
            with torch.no_grad(), ops.packed_rows(False):
                log_probs, gen_lengths, _, _, allpos = self.net.model_F(
                    inp_tokens, inp_lengths, None, None, 1 - raw_styles, generate=True, differentiable_decode=False,
                    temperature=temperature, use_inference_z=use_inference_z)
                out = Storage()
                out.gen = log_probs.argmax(-1)
        finally:
            self.net.model_F.train(was_training)
        out.gen_lengths, out.allpos, out.length_delta = None, None, None
        if self.args.use_learnable:
            glen = gen_lengths.cpu().numpy()
            pos = allpos.cpu().numpy()
            out.gen_lengths = gen_lengths
            out.allpos = [p[1:g + 1] for g, p in zip(glen, pos)]
            out.length_delta = float(np.mean(glen - inp_lengths.cpu().numpy()))
        return out

    def evaluate(self, *a, **k):
        raise NotImplementedError("evaluate needs cotk metrics + the RoBERTa classifier (out of scope, SURVEY 2.1); "
                                  "Seq2seq.transfer() is the generator's share of it")

    train_process = test_process = evaluate

    # ---- the step --------------------------------------------------------------------------------
    def approx(self, logits, mode):
        """seq2seq.py:407-414"""
        if mode == "soft":
            return logits.exp()
        if mode == "gumbel":
            return gumbel_soft_and_index(logits)[0]
        raise NotImplementedError()

    def _approx_with_index(self, logits, mode):
        if mode == "gumbel":
            return gumbel_soft_and_index(logits)      # the kernel's argmax == soft_tokens.argmax(-1)
        soft = self.approx(logits, mode)
        return soft, soft.argmax(-1)

    def _approx_packed(self, lp, mode, halves=None):
        """Seq2seq.approx on packed log-probs (ops.PackedSoft) -> packed soft tokens [rows, V].  halves=B: the batch is
        two B-sentence calls of the reference run as one (d_step)."""
        if mode == "gumbel":
            from .utils import gumbel as G
            if G._injected:                      # parity runs inject the reference's [B, T, V] draws: pack them the same way
                ps = lp.ps
                if halves is None:
                    G._injected[0] = ops.pack_rows(G._injected[0], ps)
                else:                            # two draws [B, t_raw, V], [B, t_rev, V], each at its own width
                    e0, e1 = G._injected.pop(0), G._injected.pop(0)
                    first = ps.b_of < halves
                    i0 = (ps.b_of * e0.shape[1] + ps.pos).clamp(max=e0.shape[0] * e0.shape[1] - 1)
                    i1 = ((ps.b_of - halves) * e1.shape[1] + ps.pos).clamp(min=0)
                    rows = torch.where(first.unsqueeze(-1), e0.reshape(-1, e0.shape[-1]).index_select(0, i0),
                                       e1.reshape(-1, e1.shape[-1]).index_select(0, i1))
                    pad = rows.new_ones(ps.rows - ps.total, rows.shape[-1])
                    G._injected.insert(0, torch.cat((rows, pad), 0))
            return gumbel_soft_and_index(lp.rows)[0]     # shift-invariant: logits or log-probs alike (lp.unnormalised)
        return self.approx(ops.log_softmax(lp.rows) if lp.unnormalised else lp.rows, mode)

    def get_lengths(self, tokens, eos_idx):
        """seq2seq.py:545-549: first <eos> index + 1, capped at the width."""
        return ops.first_eos_length(tokens, eos_idx)

    def _length_loss(self, losses, inp_lengths, batch_size):
        if len(losses.shape) <= 1:
            return losses
        # generateMask on the device (the reference builds it on the host: seq2seq.py:461, a D2H sync)
        mask = (torch.arange(losses.shape[1], device=losses.device).unsqueeze(0) < (inp_lengths + 1).unsqueeze(1)).float()
        return (losses * mask).sum() / batch_size * self.args.length_factor

    pack_rows = True            # Encoder / Decoder stacks of the training steps run on the real (unpadded) rows only
    pack_min_tokens = 4096      # ... when the step is GPU-bound (below, the index bookkeeping costs more host time than it saves)

    # HIP-graph replay of f_step (nast_amd/graph.py) for batches below `pack_min_tokens`, where the step is bound by the
    # host's launch rate (~2000 launches): after `StepGraphs.WARM` eager steps of a given (batch size, temperature,
    # cyc_rec_enable, use_inference_z) the step is captured once and replayed.  Off by default: the replayed step runs at
    # fixed widths and draws its Gumbel noise through torch (ops.FIXED_WIDTH / ops.GRAPH_NOISE), so the random stream
    # differs from the eager step's; single process only.
    graph_steps = False
    graph_overlap = True        # record the two-stream form of the step: the graph gets two parallel branches
    _graphs = None

    def f_step(self, incoming, temperature, cyc_rec_enable=True, *, use_inference_z):
        """seq2seq.py:416-543 -> (slf, cyc, adv, slf_len, cyc_len, slf_gen_err, cyc_gen_err) as Python floats."""
        if self.graph_steps and incoming.data.sent.is_cuda and incoming.data.sent.shape[0] * self.args.max_length < self.pack_min_tokens:
            if self._graphs is None:
                from .graph import StepGraphs
                self._graphs = StepGraphs(self)
            return self._graphs.f_step(incoming, temperature, cyc_rec_enable, use_inference_z)
        with ops.packed_rows(self.pack_rows and incoming.data.sent.numel() >= self.pack_min_tokens):
            out = self._f_step(incoming, temperature, cyc_rec_enable, use_inference_z=use_inference_z)
            ops.check_packed()       # (the step has just synchronised for its loss values)
            ops.check_decode()
            return tuple(item(x) for x in out)

    def d_step(self, incoming, temperature, *, use_inference_z):
        """seq2seq.py:551-642 -> adv_loss (float)"""
        if self.graph_steps and incoming.data.sent.is_cuda and incoming.data.sent.shape[0] * self.args.max_length < self.pack_min_tokens:
            if self._graphs is None:
                from .graph import StepGraphs
                self._graphs = StepGraphs(self)
            return self._graphs.d_step(incoming, temperature, use_inference_z)
        with ops.packed_rows(self.pack_rows and incoming.data.sent.numel() >= self.pack_min_tokens):
            out = self._d_step(incoming, temperature, use_inference_z=use_inference_z)
            ops.check_packed()
            ops.check_decode()
            return out.item()

    def _f_step(self, incoming, temperature, cyc_rec_enable=True, *, use_inference_z, capturing=False):
        """-> the seven results as 0-d device tensors (None where the reference reports 0); f_step reads them back.
        capturing=True: issued under HIP-graph capture (single stream, Adam's step count on the device)."""
        self.net.model_D.eval()
        dm = self.param.volatile.dm
        pad_idx, eos_idx = dm.pad_id, dm.eos_id
        args = self.args
        loss_fn = nn.NLLLoss(reduction='none')

        inp_tokens = incoming.data.sent
        inp_lengths = incoming.data.sent_length
        raw_styles = incoming.data.domain
        rev_styles = 1 - raw_styles
        batch_size = inp_tokens.size(0)
        token_mask = (inp_tokens != pad_idx).float()
        if ops.PACK_ROWS and inp_lengths.is_cuda:
            ops.set_total(inp_lengths, inp_lengths.sum().item())       # read while the GPU is idle (start of the step)

        self.optimizer_F.zero_grad(set_to_none=False)   # gradient buffers persist: the HIP kernels accumulate in place

        noise_inp_tokens, noise_inp_lengths = self.net.noiseLayer(inp_tokens.transpose(0, 1), inp_lengths, rev_domain=rev_styles)
        noise_inp_tokens = noise_inp_tokens.transpose(0, 1).contiguous()

        model_F = self.net.model_F
        # The transfer pass (seq2seq.py:476-490), the cycle pass forward and the discriminator forward depend on nothing the
        # self-reconstruction pass produces (backward #1 only adds to .grad): they go to a second HIP stream; the streams are
        # ordered both ways before backward #2 and joined after it.  (Small batches are bound by the host launch rate, where a
        # second stream only adds synchronisation: measured 33 -> 37 ms at 64 sentences.)
        overlap = self.overlap_passes and cyc_rec_enable and inp_tokens.is_cuda and inp_tokens.numel() >= self._overlap_threshold() \
            and (not capturing or self.graph_overlap)

        def self_pass():
            """self-reconstruction pass: forward + backward #1 (seq2seq.py:440-464)"""
            # nll_targets: Generator + NLLLoss(reduction='none') (seq2seq.py:453) run as one node on the device
            slf_nll, _, slf_length_losses, slf_gen_error, _ = model_F(
                noise_inp_tokens, noise_inp_lengths, inp_tokens, inp_lengths, raw_styles,
                generate=False, differentiable_decode=False, temperature=temperature, use_inference_z=use_inference_z,
                nll_targets=inp_tokens)
            slf_rec_loss = slf_nll * token_mask
            slf_rec_loss = slf_rec_loss.sum() / batch_size
            slf_rec_loss = slf_rec_loss * args.slf_factor
            if not cyc_rec_enable:
                self.reducer_F.arm("f_pretrain")
            if args.use_learnable:
                slf_rec_length_loss = self._length_loss(slf_length_losses, inp_lengths, batch_size)
                (slf_rec_loss + slf_rec_length_loss).backward()
            else:
                slf_rec_length_loss = None
                slf_rec_loss.backward()
            return slf_rec_loss, slf_rec_length_loss, slf_gen_error

        if not cyc_rec_enable:                                                 # seq2seq.py:471-474 (no clipping here)
            slf_rec_loss, slf_rec_length_loss, slf_gen_error = self_pass()
            self.reducer_F.finish()
            self.optimizer_F.clip_and_step(None, capturing=capturing)
            self.net.model_D.train()
            return slf_rec_loss, None, None, slf_rec_length_loss, None, slf_gen_error, None

        def transfer_pass():
            glp, glen, _, _, _ = model_F(
                inp_tokens, inp_lengths, None, None, rev_styles,
                generate=True, differentiable_decode=True, temperature=temperature, use_inference_z=use_inference_z,
                packed_log_probs=ops.PACK_ROWS and args.use_learnable)
            if isinstance(glp, ops.PackedSoft):
                # the generated sentences stay packed through the Gumbel-softmax: the cycle pass and the discriminator read
                # them through x @ table on the real rows only
                return ops.PackedSoft(self._approx_packed(glp, args.gen_approx_mode), glp.ps), glen
            soft, idx = self._approx_with_index(glp, args.gen_approx_mode)
            if not args.use_learnable:
                glen = self.get_lengths(idx, eos_idx)
            return soft, glen

        def cycle_forward():
            """transfer pass, cycle pass forward and discriminator forward (seq2seq.py:476-528)"""
            gen_soft_tokens, gen_lengths = transfer_pass()
            cyc_nll, _, cyc_length_losses, cyc_gen_error, _ = model_F(
                gen_soft_tokens, gen_lengths, inp_tokens, inp_lengths, raw_styles,
                generate=False, differentiable_decode=False, temperature=temperature, use_inference_z=use_inference_z,
                nll_targets=inp_tokens)
            cyc_rec_loss = cyc_nll * token_mask                                    # seq2seq.py:504
            cyc_rec_loss = cyc_rec_loss.sum() / batch_size
            cyc_rec_loss = cyc_rec_loss * args.cyc_factor
            cyc_rec_length_loss = self._length_loss(cyc_length_losses, inp_lengths, batch_size) if args.use_learnable else None

            # the discriminator's own parameter gradients are never used by f_step (the reference computes and
            # discards them: optimizer_D.zero_grad() precedes the next use, seq2seq.py:635), so its weights are
            # frozen for this forward and the weight-gradient GEMMs are not issued
            d_params = [p for p in self.net.model_D.parameters() if p.requires_grad]
            for p in d_params:
                p.requires_grad_(False)
            try:
                adv_log_probs = self.net.model_D(gen_soft_tokens, gen_lengths, rev_styles)
            finally:
                for p in d_params:
                    p.requires_grad_(True)
            if args.discriminator_method == 'Multi':
                adv_labels = rev_styles + 1
            else:
                adv_labels = torch.ones_like(rev_styles)
            adv_loss = loss_fn(adv_log_probs, adv_labels)
            adv_factor = (raw_styles == 0) * args.fw_adv_factor + (raw_styles == 1) * args.bw_adv_factor
            adv_loss = (adv_loss * adv_factor).sum() / batch_size

            total = cyc_rec_loss + adv_loss
            if args.use_learnable:
                total = total + cyc_rec_length_loss
            return total, cyc_rec_loss, cyc_rec_length_loss, cyc_gen_error, adv_loss

        if overlap:
            # Issue order on the host: transfer-pass encoder and the greedy decode (asynchronous: no early-exit read-backs)
            # on the side stream, then -- from the hook LengthPredictor calls right after enqueuing the decode -- the whole
            # self-reconstruction pass on the main stream, then the rest of the side stream's work.  The decode's ~9 ms of
            # tiny dependent launches thus run beside pass 1 instead of holding up the host and an idle GPU.
            main = torch.cuda.current_stream()
            side = self._side_stream(inp_tokens.device)
            side.wait_stream(main)          # inputs are ready
            first = {}

            def issue_self_pass():
                with torch.cuda.stream(main):
                    first["ret"] = self_pass()

            with ops.async_decode(issue_self_pass), torch.cuda.stream(side):
                total, cyc_rec_loss, cyc_rec_length_loss, cyc_gen_error, adv_loss = cycle_forward()
            slf_rec_loss, slf_rec_length_loss, slf_gen_error = first["ret"]
            # backward #2 runs on the side stream (the stream of its nodes' forward) and adds to the .grad buffers
            # backward #1 created and filled on the main stream: order the two streams both ways before it starts
            main.wait_stream(side)
            side.wait_stream(main)
        else:
            slf_rec_loss, slf_rec_length_loss, slf_gen_error = self_pass()
            # fixed-width execution: the decode runs all its steps without the early-exit read-backs
            with (ops.async_decode(None) if ops.FIXED_WIDTH else contextlib.nullcontext()):
                total, cyc_rec_loss, cyc_rec_length_loss, cyc_gen_error, adv_loss = cycle_forward()

        self.reducer_F.arm("f_full", streams=(main, side) if overlap else ())
        total.backward()
        if overlap:
            # autograd ran the transfer pass's backward nodes on the side stream (the stream of their forward) and
            # joins only the streams of AccumulateGrad nodes at the end of backward; the kernels here accumulate
            # into .grad in place, so no such node exists on the side stream: join it explicitly before the
            # gradients are read (clip, Adam, all-reduce write-back) and before any buffer it reads is recycled.
            main.wait_stream(side)
        self.reducer_F.finish()

        self.last_grad_norm = self.optimizer_F.clip_and_step(5, capturing=capturing)                # seq2seq.py:536-537
        self.net.model_D.train()

        return slf_rec_loss, cyc_rec_loss, adv_loss, slf_rec_length_loss, cyc_rec_length_loss, slf_gen_error, cyc_gen_error

    def _d_step(self, incoming, temperature, *, use_inference_z, capturing=False):
        """-> adv_loss as a 0-d device tensor (d_step reads it back).  capturing=True: issued under HIP-graph capture."""
        dm = self.param.volatile.dm
        self.net.model_F.eval()
        eos_idx = dm.eos_id
        args = self.args
        loss_fn = nn.NLLLoss(reduction='none')
        inp_tokens = incoming.data.sent
        inp_lengths = incoming.data.sent_length
        raw_styles = incoming.data.domain
        rev_styles = 1 - raw_styles
        batch_size = inp_tokens.size(0)
        model_F, model_D = self.net.model_F, self.net.model_D
        if ops.PACK_ROWS and inp_lengths.is_cuda:
            ops.set_total(inp_lengths, inp_lengths.sum().item())

        overlap = self.overlap_passes and inp_tokens.is_cuda and inp_tokens.numel() >= self._overlap_threshold() \
            and (not capturing or self.graph_overlap)

        def gold_forward():
            if args.discriminator_method == 'Multi':
                return model_D(inp_tokens, inp_lengths), raw_styles + 1
            return (torch.cat((model_D(inp_tokens, inp_lengths, raw_styles), model_D(inp_tokens, inp_lengths, rev_styles)), 0),
                    torch.cat((torch.ones_like(raw_styles), torch.zeros_like(rev_styles)), 0))

        gold = {}
        # fixed-width execution: the decode runs all its steps without the early-exit read-backs
        hook = ops.async_decode(None) if ops.FIXED_WIDTH else contextlib.nullcontext()
        if overlap:
            # The gold half of the loss needs nothing from the generator: its forward AND backward are issued on the second
            # stream from the decode hook (ops.async_decode), right behind the enqueue of the greedy decode, so they fill
            # the CUs that those ~800 small dependent launches leave idle.  adv_loss = gold part + generated part
            # (seq2seq.py:632 sums over the concatenation); both parts add into the same .grad buffers.
            main = torch.cuda.current_stream()
            side = self._side_stream(inp_tokens.device)
            self.optimizer_D.zero_grad(set_to_none=False)
            side.wait_stream(main)

            def gold_pass():
                with torch.cuda.stream(side):
                    glp, glab = gold_forward()
                    part = loss_fn(glp, glab).sum() / batch_size
                    part.backward()
                gold["loss"] = part.detach()

            hook = ops.async_decode(gold_pass)

        both_packed = None
        with hook, torch.no_grad():
            if inp_tokens.is_cuda and 2 * batch_size <= self.merge_generate_max:
                # The two generation passes (seq2seq.py:565-580) differ only in the style token: they run as ONE batch of
                # 2B sentences -- every op is per sentence, and the greedy skip decoding (latency-bound: ~800 small
                # dependent launches) is paid once instead of twice.  Each half is then cut back to its own width
                # (max generated length of that half), which is what the separate calls return.
                both_inp_lengths = torch.cat((inp_lengths, inp_lengths), 0)
                if hasattr(inp_lengths, "nast_total"):
                    ops.set_total(both_inp_lengths, 2 * inp_lengths.nast_total)
                both_log_probs, both_lengths, _, _, _ = model_F(
                    torch.cat((inp_tokens, inp_tokens), 0), both_inp_lengths, None, None,
                    torch.cat((raw_styles, rev_styles), 0), generate=True, differentiable_decode=True,
                    temperature=temperature, use_inference_z=use_inference_z,
                    packed_log_probs=ops.PACK_ROWS and args.use_learnable)
                raw_gen_lengths = rev_gen_lengths = None
                if isinstance(both_log_probs, ops.PackedSoft):
                    both_packed = both_log_probs             # real positions only; handled as one 2B batch below
                else:
                    raw_gen_log_probs, rev_gen_log_probs = both_log_probs[:batch_size], both_log_probs[batch_size:]
                if both_lengths is not None and both_packed is None:
                    raw_gen_lengths, rev_gen_lengths = both_lengths[:batch_size], both_lengths[batch_size:]
                if both_lengths is not None and both_packed is None and not ops.FIXED_WIDTH:
                    t_raw, t_rev, s_raw, s_rev = torch.stack((raw_gen_lengths.max(), rev_gen_lengths.max(),
                                                              raw_gen_lengths.sum(), rev_gen_lengths.sum())).tolist()
                    ops.set_total(raw_gen_lengths, s_raw)
                    ops.set_total(rev_gen_lengths, s_rev)
                    raw_gen_log_probs, rev_gen_log_probs = raw_gen_log_probs[:, :t_raw], rev_gen_log_probs[:, :t_rev]
            else:
                raw_gen_log_probs, raw_gen_lengths, _, _, _ = model_F(
                    inp_tokens, inp_lengths, None, None, raw_styles, generate=True, differentiable_decode=True,
                    temperature=temperature, use_inference_z=use_inference_z)
                rev_gen_log_probs, rev_gen_lengths, _, _, _ = model_F(
                    inp_tokens, inp_lengths, None, None, rev_styles, generate=True, differentiable_decode=True,
                    temperature=temperature, use_inference_z=use_inference_z)

        if both_packed is None:
            raw_gen_soft_tokens, raw_idx = self._approx_with_index(raw_gen_log_probs, args.dis_approx_mode)
            if not args.use_learnable:
                raw_gen_lengths = self.get_lengths(raw_idx, eos_idx)
            rev_gen_soft_tokens, rev_idx = self._approx_with_index(rev_gen_log_probs, args.dis_approx_mode)
            if not args.use_learnable:
                rev_gen_lengths = self.get_lengths(rev_idx, eos_idx)

        if not overlap:
            gold_log_probs, gold_labels = gold_forward()
        if both_packed is not None:
            # packed rows: Gumbel-softmax and the discriminator see the 2B generated sentences as one batch, raw styles first --
            # the row order of torch.cat((D(raw ...), D(rev ...)), 0) in seq2seq.py:601-616
            both_soft = ops.PackedSoft(self._approx_packed(both_packed, args.dis_approx_mode, halves=batch_size), both_packed.ps)
            if args.discriminator_method == 'Multi':
                gen_log_probs = model_D(both_soft, both_lengths)
                gen_labels = torch.cat((raw_styles + 1, torch.zeros_like(rev_styles)), 0)
            else:
                gen_log_probs = model_D(both_soft, both_lengths, torch.cat((raw_styles, rev_styles), 0))
                gen_labels = torch.cat((torch.ones_like(raw_styles), torch.zeros_like(rev_styles)), 0)
        elif args.discriminator_method == 'Multi':
            gen_log_probs = torch.cat((model_D(raw_gen_soft_tokens, raw_gen_lengths),
                                       model_D(rev_gen_soft_tokens, rev_gen_lengths)), 0)
            gen_labels = torch.cat((raw_styles + 1, torch.zeros_like(rev_styles)), 0)
        else:
            gen_log_probs = torch.cat((model_D(raw_gen_soft_tokens, raw_gen_lengths, raw_styles),
                                       model_D(rev_gen_soft_tokens, rev_gen_lengths, rev_styles)), 0)
            gen_labels = torch.cat((torch.ones_like(raw_styles), torch.zeros_like(rev_styles)), 0)

        if overlap:
            assert len(gen_log_probs.size()) == 2
            gen_part = loss_fn(gen_log_probs, gen_labels).sum() / batch_size
            main.wait_stream(side)              # the gold pass ran during the decode; the two backwards never overlap
            if not capturing:
                gold["loss"].record_stream(main)
            self.reducer_D.arm("d")
            gen_part.backward()
            adv_loss = gold["loss"] + gen_part.detach()
        else:
            adv_log_probs = torch.cat((gold_log_probs, gen_log_probs), 0)
            adv_labels = torch.cat((gold_labels, gen_labels), 0)
            adv_loss = loss_fn(adv_log_probs, adv_labels)
            assert len(adv_loss.size()) == 1
            adv_loss = adv_loss.sum() / batch_size

            self.optimizer_D.zero_grad(set_to_none=False)
            self.reducer_D.arm("d")
            adv_loss.backward()
        self.reducer_D.finish()
        self.last_grad_norm = self.optimizer_D.clip_and_step(5, capturing=capturing)                # seq2seq.py:637-638
        model_F.train()
        return adv_loss
