#!/usr/bin/env python3
"""Headline benchmark: train sentences/sec of the NAST-StyTrans (learnable alignment) generator training step.

A "step" is one `Seq2seq.f_step` as the reference defines it (seq2seq.py:416-543): NoiseLayer (device kernel), three
generator forward passes (self-reconstruction, transfer with greedy skip decoding, cycle reconstruction from
Gumbel-softmax soft tokens), one discriminator forward, two backward passes, global-norm clip and Adam.

Workload (BASELINE.json configs[2]/[3]): learnable alignment, GYAFC-shaped synthetic batch (len <= 32, vocab 9600,
d=256, h=4, 4 layers), 512 sentences per GPU (the per-GPU batch configs[3] names for the 2/4/8-GPU runs, used at
N=1 as well so that the scaling is weak: fixed work per GPU), fp32 (the reference's dtype), use_inference_z=True.
Every rank holds its own 512 sentences; gradients are averaged with bucketed RCCL all-reduce.  At N=1 the line
also carries "also.bs64" = the same step at 64 sentences (the reference's batch size).

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (see README / DESIGN.md section "Measurement").
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# algorithmic GFLOP per sentence of one f_step (SURVEY.md section 8d / BASELINE.md section 3)
GFLOP_PER_SENTENCE = {("learnable", 32): 10.1, ("learnable", 16): 5.12, ("simple", 16): 3.80}
PEAK_HBM_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable by a float4 copy)
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak (= fp32 vector peak)
PEAK_BF16_MFMA_TFLOPS = 2516.6     # MI355X_MICROARCH.md: dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16)


def synth_incoming(B, L, V, seed, device):
    """tokens ~ U{4..V-1}, lengths (incl. eos) ~ U{4..L}, sentence 0 has length L,
    first half style 0, second half style 1."""
    from nast_amd.utils import Storage
    rng = np.random.default_rng(seed)
    lens = rng.integers(4, L + 1, size=B)
    lens[0] = L
    sent = np.zeros((B, L), dtype=np.int64)
    for b in range(B):
        n = int(lens[b])
        sent[b, :n - 1] = rng.integers(4, V, size=n - 1)
        sent[b, n - 1] = 2
    inc = Storage(); inc.data = Storage()
    inc.data.sent = torch.from_numpy(sent).to(device)
    inc.data.sent_length = torch.from_numpy(lens.astype(np.int64)).to(device)
    inc.data.domain = torch.tensor([0] * (B // 2) + [1] * (B - B // 2), device=device)
    inc.data.batch_size = B
    return inc


def build_model(cfg, device):
    from helpers import make_param
    import nast_amd.utils as U
    from nast_amd.seq2seq import Seq2seq
    torch.manual_seed(0)
    np.random.seed(0)
    U.cuda_init(torch.cuda.current_device() if device == "cuda" else 0, device)
    model = Seq2seq(make_param(device=device, **cfg))
    gen = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for n, p in model.net.named_parameters():
            # reference init leaves every embedding at exactly 0 (degenerate alignment): re-draw xavier-uniform
            if n.endswith("embed.weight"):
                bound = (6.0 / (p.shape[0] + p.shape[1])) ** 0.5
                p.copy_(((torch.rand(p.shape, generator=gen) * 2 - 1) * bound).to(p.device))
        cnt = model.net.noiseLayer.cnt
        cnt.zero_()
        cnt[:, 4:] = 1.0 / (cnt.shape[1] - 4)
    model.net.model_F.train(); model.net.model_D.train()
    model.net.model_F.check_lengths = False        # the reference's host-side assert (a D2H sync per forward)
    return model


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(model, cfg, B, steps=5, warm=2):
    B = min(B, 64)       # bounded sample (the reference's own batch size): CPU throughput of this path is flat in the batch size
    """The CPU oracle (oracle/nast_oracle.py, pinned against the reference by the golden fixtures) timed on this
    box's host cores on a bounded sample of the same workload."""
    from oracle import nast_oracle as O
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    weights = {k: v.detach().cpu().clone() for k, v in model.net.state_dict().items() if k.startswith("model_")}
    tr = O.Trainer(weights, O.Cfg(**cfg))
    inc = synth_incoming(B, cfg["L"], cfg["V"], 1234, "cpu")
    sent, slen, dom = inc.data.sent, inc.data.sent_length, inc.data.domain
    times = []
    for i in range(steps + warm):
        np.random.seed(i)
        ntok, nlen = model.net.noiseLayer(sent.t(), slen, 1 - dom)
        ntok = ntok.t().contiguous().cpu(); nlen = nlen.cpu()
        t0 = time.perf_counter()
        if cfg["learnable"]:
            with torch.no_grad():
                glp = O.style_transformer(tr.w, tr.cfg, sent, slen, None, None, 1 - dom, True, 1.0, True)[0]
            E = torch.empty_like(glp).exponential_()
        else:
            E = torch.empty(B, cfg["L"], cfg["V"]).exponential_()
        t1 = time.perf_counter()
        tr.f_step(sent, slen, dom, ntok, nlen, [E])
        t2 = time.perf_counter()
        if i >= warm:
            times.append(t2 - t1)      # the shape probe (t1 - t0) is harness overhead, not part of the step
    sps = B / (sum(times) / len(times))
    return {"value": round(sps, 2), "unit": "sentences/s", "cores": threads, "kind": "port", "cpu": _cpu_model(),
            "host_cpus": os.cpu_count(),
            "sample": f"{steps} f_step of the same workload on a {B}-sentence sample after {warm} warm-up steps, torch CPU fp32, "
                      f"{threads} threads"}


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher around it: start N fresh child processes (one rank per GPU, the
    environment torch.distributed.run would give them), relay rank 0's one JSON line, and fail loudly -- exit code 1 and
    the failing rank's stderr tail -- if a rank dies or the job exceeds its deadline.  Nothing here touches the GPU: the
    children are started before this process has made any HIP call (never an exec from a process that has)."""
    import socket
    import subprocess
    import tempfile
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    deadline = time.time() + float(os.environ.get("NAST_BENCH_TIMEOUT_S", "1500"))
    logs, procs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        env.setdefault("OMP_NUM_THREADS", str(max(1, min(16, (os.cpu_count() or n) // n))))
        log = tempfile.TemporaryFile(mode="w+b")
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdin=subprocess.DEVNULL,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=log))

    def tail(r, nbytes=4000):
        logs[r].seek(0, os.SEEK_END)
        size = logs[r].tell()
        logs[r].seek(max(0, size - nbytes))
        return logs[r].read().decode(errors="replace")

    failed = None
    while failed is None:
        codes = [p.poll() for p in procs]
        bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = (bad[0], f"rank {bad[0]} exited with code {codes[bad[0]]}")
        elif all(c == 0 for c in codes):
            break
        elif time.time() > deadline:
            slow = [r for r, c in enumerate(codes) if c is None]
            failed = (slow[0], f"rank(s) {slow} still running at the deadline (NAST_BENCH_TIMEOUT_S)")
        else:
            time.sleep(0.2)
    if failed is not None:
        for p in procs:                       # the exact children started above, nothing else
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
        r, why = failed
        sys.stderr.write(f"[bench] {why}; stderr tail of rank {r}:\n{tail(r)}\n")
        for q in range(n):
            if q != r:
                sys.stderr.write(f"[bench] --- rank {q} (last lines) ---\n{tail(q, 800)}\n")
        sys.exit(1)
    out = procs[0].stdout.read().decode()
    for r in range(n):
        sys.stderr.write(tail(r, 2000))
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    if len(lines) != 1:
        sys.stderr.write(f"[bench] expected one JSON line from rank 0, got {len(lines)}\n")
        sys.exit(1)
    sys.stdout.write(lines[0] + "\n")
    sys.stdout.flush()


def main():
    # stdout carries ONE JSON line: everything else that writes to file descriptor 1 (RCCL prints its version banner there under
    # NCCL_DEBUG=VERSION, which the GPU boxes set) is sent to stderr; the line itself goes to the saved descriptor
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-per-gpu", type=int, default=512)
    ap.add_argument("--alignment", default="learnable", choices=["learnable", "simple"])
    ap.add_argument("--max-length", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-also", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="serialise f_step's transfer pass instead of a second stream")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16"],
                    help="GEMM arithmetic: fp32 MFMA (reference dtype, default) or the bf16-MFMA variant")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        os.dup2(real_stdout, 1)
        return launch_ranks(a.gpus, sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: the launcher's world size and --gpus must agree")
    # NAST_DIST_BACKEND=gloo is a rehearsal aid: several ranks may then share one GPU (RCCL needs one GPU per rank)
    backend = os.environ.get("NAST_DIST_BACKEND", "nccl")
    dev_index = local_rank % max(torch.cuda.device_count(), 1) if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    # NAST_DIST_SINGLE=1 is a rehearsal aid for a one-GPU box: a process group of ONE rank, so the bucketed all-reduce,
    # the barrier and the max-over-ranks reduction go through RCCL's real launch path (nast_amd.ddp.FORCE_SINGLE)
    dist_on = world > 1 or os.environ.get("NAST_DIST_SINGLE") == "1"
    if dist_on:
        import torch.distributed as dist
        if os.environ.get("NAST_NO_RESERVE") != "1":
            from nast_amd.seq2seq import reserve_streams
            reserve_streams()        # before RCCL takes its hardware queues (see its docstring)
        if world == 1:
            for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29531"), ("RANK", "0"), ("WORLD_SIZE", "1")):
                os.environ.setdefault(k, v)
            import nast_amd.ddp as _ddp
            _ddp.FORCE_SINGLE = True
        # a collective that never completes ends the job with RCCL's watchdog message after this long, not at the job's limit
        import datetime
        tmo = datetime.timedelta(seconds=float(os.environ.get("NAST_DDP_TIMEOUT_S", "120")))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index), timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)
    learnable = a.alignment == "learnable"
    L = a.max_length or (32 if learnable else 16)
    cfg = dict(V=9600, d=256, h=4, N=4, L=L, learnable=learnable)
    B = a.batch_per_gpu
    model = build_model(cfg, "cuda")
    if os.environ.get("NAST_PACK_MIN"):               # A/B aid: token count from which the packed-row execution is used
        model.pack_min_tokens = int(os.environ["NAST_PACK_MIN"])
    if os.environ.get("NAST_NOISE_HOST") == "1":      # A/B aid: the host (numpy) NoiseLayer instead of the device kernel
        model.net.noiseLayer.device_side = False
    from nast_amd import ops as _ops
    _ops.set_gemm_precision(a.precision)
    model.overlap_passes = not a.no_overlap
    # every rank holds different sentences (its shard of the global batch) and identical weights
    batches = [synth_incoming(B, L, cfg["V"], 1000 * rank + i, "cuda") for i in range(4)]
    np.random.seed(100 + rank)
    torch.cuda.manual_seed(100 + rank)

    debug = os.environ.get("NAST_BENCH_DEBUG")      # rehearsal aid: progress lines + a stack dump if a step stalls
    debug = debug if debug and int(debug) > 0 else None
    if debug:
        import faulthandler
        faulthandler.dump_traceback_later(int(debug), repeat=True, file=sys.stderr)

    def run(n, offset=0):
        for i in range(n):
            model.f_step(batches[(i + offset) % len(batches)], 1.0, use_inference_z=True)
            if debug:
                print(f"[rank {rank}] step {i + 1}/{n} done at {time.perf_counter():.1f}", file=sys.stderr, flush=True)

    def fence():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize()

    run(a.warmup)
    fence()
    t0 = time.perf_counter()
    run(a.steps)
    fence()
    elapsed = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    from nast_amd import _lib
    lib = _lib.load()
    init_state = {k: v.detach().clone() for k, v in model.net.state_dict().items()}

    def reset(m=None, state=None, seed=None):
        """Every leg below starts where the headline run started: initial weights, empty Adam state, the rank's seeds, no captured
        graphs -- so that no figure depends on which legs ran before it (the generated lengths, hence the widths of the cycle pass
        and of the discriminator's inputs, follow the weights)."""
        m, state = m or model, state or init_state
        with torch.no_grad():
            for k, v in m.net.state_dict().items():
                v.copy_(state[k])
        for opt in (m.optimizer_F, m.optimizer_D):
            opt.state.clear()
            opt._tables.clear()
            opt._dev_step = None
        m._graphs = None
        m.graph_steps = False
        np.random.seed(100 + rank if seed is None else seed)
        torch.manual_seed(100 + rank if seed is None else seed)
        torch.cuda.manual_seed(100 + rank if seed is None else seed)

    def timed(fn, n, warm=4):
        for i in range(warm):
            fn(i)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(n):
            fn(warm + i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n

    def gemm_roofline(m, fn, n, warm, peak, what):
        """HIP-event timing around every launch of the tiled GEMM kernel over n steps of fn (two-stream overlap off, so an event
        pair times one kernel alone) -> achieved TFLOP/s against `peak`, plus the other armed families."""
        saved, m.overlap_passes = m.overlap_passes, False
        try:
            for i in range(warm):
                fn(i)
            torch.cuda.synchronize()
            lib.nast_prof_begin(0)      # every family: 1 GEMM, 2 vocabulary row kernels, 3 alignment DP, 4 small-M GEMM, 5 LSE alignment, 6 cluster decode
            for i in range(n):
                fn(warm + i)
            torch.cuda.synchronize()
        finally:
            m.overlap_passes = saved
        ms, fl, cnt = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        lib.nast_prof_end(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(cnt))
        if ms.value <= 0:
            return None
        ach = fl.value / (ms.value * 1e-3) / 1e12
        return {"bound": "mfma", "kernel": what, "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                "traffic": None, "launches": int(cnt.value), "steps": n, "avg_launch_us": round(ms.value * 1e3 / max(cnt.value, 1), 2),
                "avg_launch_gflop": round(fl.value / max(cnt.value, 1) / 1e9, 4), "gemm_ms_per_step": round(ms.value / n, 3)}

    def family(kind):
        ms, wk, cnt = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        lib.nast_prof_get(kind, ctypes.byref(ms), ctypes.byref(wk), ctypes.byref(cnt))
        return ms.value, wk.value, int(cnt.value)

    f_fn = lambda i: model.f_step(batches[i % len(batches)], 1.0, use_inference_z=True)      # noqa: E731
    d_fn = lambda i: model.d_step(batches[i % len(batches)], 1.0, use_inference_z=True)      # noqa: E731

    roofline = None
    roofline_error = None
    gemm_flops_per_step = None
    if not a.no_roofline:
        try:
            # A second pass over the SAME steps as the headline run (initial weights, the same warm-up, then the timed steps'
            # first five) with event timing armed; kept out of the pass that produces `value`.  EVERY rank runs the steps (they
            # contain the gradient all-reduce); rank 0 reports its own kernels.
            n_roof = min(a.steps, 5)
            reset()
            roofline = gemm_roofline(model, f_fn, n_roof, a.warmup, PEAK_FP32_MFMA_TFLOPS, "nast_gemm_f32_kernel (tiled fp32 MFMA GEMM, 2/3 of kernel time)")
            if rank == 0 and roofline:
                ms1, fl1, n1 = family(1)
                ms4, fl4, n4 = family(4)      # the small-M GEMM kernel of the greedy decode (M = sentences, latency-bound)
                gemm_flops_per_step = (fl1 + fl4) / n_roof      # GEMM FLOP actually issued per step
                if ms4 > 0:
                    roofline["decode_small_m_gemm"] = {"achieved": round(fl4 / (ms4 * 1e-3) / 1e12, 2), "unit": "TFLOP/s",
                                                       "launches": n4, "avg_launch_us": round(ms4 * 1e3 / max(n4, 1), 2)}
                    allf = (fl1 + fl4) / ((ms1 + ms4) * 1e-3) / 1e12
                    roofline["gemm_family_all"] = {"achieved": round(allf, 2), "frac": round(allf / PEAK_FP32_MFMA_TFLOPS, 4), "launches": n1 + n4}
                ms6, fl6, n6 = family(6)      # the skip decode as one persistent launch (csrc/decode_cluster.hip): latency-bound by design
                if ms6 > 0:
                    roofline["decode_cluster_kernel"] = {"ms_per_decode": round(ms6 / max(n6, 1), 3), "achieved": round(fl6 / (ms6 * 1e-3) / 1e12, 2),
                                                         "unit": "TFLOP/s", "launches": n6,
                                                         "note": "32 steps x 26 dependent operations separated by cluster barriers; bound by the hand-off latency, not by MFMA"}
                # the HBM-bound families next to it: algorithmic bytes / HIP-event time
                hbm = {}
                for kind, name in ((2, "vocab_row_kernels"), (3, "alignment_dp")):
                    ms, wk, n = family(kind)
                    if ms > 0:
                        gbs = wk / (ms * 1e-3) / 1e9
                        hbm[name] = {"achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                                     "launches": n, "avg_launch_us": round(ms * 1e3 / max(n, 1), 2),
                                     "algorithmic_bytes_per_launch": round(wk / max(n, 1))}
                roofline["hbm_bound_kernels"] = hbm
                # HBM-side bytes per launch of the same kernel families from the committed PMC passes (tools/pmc_traffic.py:
                # FETCH_SIZE x2 + WRITE_SIZE, gfx950 corrections of the microarchitecture guide), newest round first
                here = os.path.dirname(os.path.abspath(__file__))
                tpath = next((q for q in (os.path.join(here, "profiles", f"r{r}_pmc_traffic.json") for r in (3, 2)) if os.path.exists(q)), None)
                if B == 512 and learnable and tpath:
                    with open(tpath) as f:
                        t = json.load(f)
                    if "nast_gemm_f32_kernel" in t:
                        roofline["traffic"] = round(t["nast_gemm_f32_kernel"]["bytes_per_launch"])
                        roofline["traffic_unit"] = f"bytes per launch, PMC passes recorded in profiles/{os.path.basename(tpath)[:-5]}.txt"
                    # the same launches as `achieved` counts: every V-wide row kernel of the step (the PMC passes ran 5 f_steps:
                    # tools/collect_profiles.sh, 2 warm-up + 3), per step and then per launch of the roofline pass
                    keys = ("nast_lse_nll_kernel", "nast_nll_logits_bwd_kernel", "nast_softmax_bwd_rr_kernel", "nast_gumbel_softmax_fwd_rr_kernel",
                            "nast_log_softmax_fwd_rr_kernel", "nast_nll_log_softmax_bwd_kernel", "nast_nll_gather_kernel")
                    fams = [t[k] for k in keys if k in t]
                    if fams and "vocab_row_kernels" in hbm:
                        per_step = sum(x["bytes_per_launch"] * x["launches"] for x in fams) / 5.0
                        hbm["vocab_row_kernels"]["traffic"] = round(per_step / (hbm["vocab_row_kernels"]["launches"] / n_roof))
                        hbm["vocab_row_kernels"]["traffic_per_step"] = round(per_step)
        except Exception as e:      # noqa: BLE001  (the headline line is printed regardless)
            import traceback
            traceback.print_exc(file=sys.stderr)
            roofline_error = f"{type(e).__name__}: {e}"[:300]
            model.overlap_passes = not a.no_overlap

    also = None
    if world == 1 and B > 64 and not a.no_also:
        # (optional legs: a failure in one of them must not cost the headline line; what was measured so far is kept)
        also = {}
        try:
            n_leg = min(a.steps, 8)
            # the step without its optimizer (clip + Adam skipped; forward + both backwards only): same start, same batches
            reset()
            step_fn = model.optimizer_F.clip_and_step
            model.optimizer_F.clip_and_step = lambda max_norm=None, capturing=False: None
            try:
                dt_fb = timed(f_fn, n_leg, a.warmup)
            finally:
                model.optimizer_F.clip_and_step = step_fn
            also["fwd_bwd_only"] = {"value": round(B / dt_fb, 2), "unit": "sentences/s", "ms_per_step": round(dt_fb * 1e3, 3), "batch": B,
                                    "note": "weights stay at their initial values here, which generate longer sentences than the weights the "
                                            "headline run reaches after a few Adam steps: the cycle pass and the discriminator are wider"}
            # the discriminator step (seq2seq.py:551-642) on the same batches, from the same initial weights
            reset()
            dt_d = timed(d_fn, n_leg, a.warmup)
            also["d_step"] = {"value": round(B / dt_d, 2), "unit": "sentences/s", "ms_per_step": round(dt_d * 1e3, 3), "batch": B}
            reset()
            rd = gemm_roofline(model, d_fn, min(a.steps, 5), a.warmup, PEAK_FP32_MFMA_TFLOPS, "nast_gemm_f32_kernel in d_step")
            if rd:
                also["d_step"]["roofline"] = rd
            f_ms = elapsed / a.steps
            # the reference schedule runs 10 d_step per 5 f_step (run.py:133-134)
            also["schedule_blend_10d_5f"] = {"value": round(15 * B / (10 * dt_d + 5 * f_ms), 2), "unit": "sentences/s"}

            # the same steps at the reference's batch size (64 sentences), eager and replayed from a HIP graph
            small = [synth_incoming(64, L, cfg["V"], 5000 + i, "cuda") for i in range(4)]
            sf = lambda i: model.f_step(small[i % 4], 1.0, use_inference_z=True)      # noqa: E731
            sd = lambda i: model.d_step(small[i % 4], 1.0, use_inference_z=True)      # noqa: E731
            n_small = min(a.steps, 10)

            def graphed(m, fn, n):
                """the same step replayed from one HIP graph (nast_amd/graph.py): eager fixed-width steps, capture, then timing"""
                m.graph_steps = True
                try:
                    return timed(fn, n, 4)
                finally:
                    m.graph_steps = False

            reset(); dt = timed(sf, n_small, 3)
            also["bs64"] = {"value": round(64 / dt, 2), "unit": "sentences/s", "ms_per_step": round(dt * 1e3, 3), "steps": n_small}
            reset(); dtg = graphed(model, sf, max(n_small, 10))
            also["bs64_graph"] = {"value": round(64 / dtg, 2), "unit": "sentences/s", "ms_per_step": round(dtg * 1e3, 3),
                                  "note": "whole f_step replayed from one HIP graph (Seq2seq.graph_steps = True)"}
            reset(); dtd = timed(sd, n_small, 3)
            reset(); dtdg = graphed(model, sd, max(n_small, 10))
            also["bs64_d_step"] = {"ms_per_step": round(dtd * 1e3, 3), "ms_per_step_graph": round(dtdg * 1e3, 3)}
            also["bs64_schedule_blend_10d_5f"] = {"value": round(15 * 64 / (10 * dtd + 5 * dt), 2), "value_graph": round(15 * 64 / (10 * dtdg + 5 * dtg), 2),
                                                  "unit": "sentences/s"}
            if a.precision == "fp32":
                # north_star's bf16 variant: GEMMs on v_mfma_f32_32x32x16_bf16 (fp32 accumulate), same step from the same start
                reset()
                _ops.set_gemm_precision("bf16")
                try:
                    dt = timed(f_fn, n_leg, a.warmup)
                    reset()
                    rb = gemm_roofline(model, f_fn, min(a.steps, 5), a.warmup, PEAK_BF16_MFMA_TFLOPS, "tiled GEMM kernel, bf16 MFMA operands")
                finally:
                    _ops.set_gemm_precision("fp32")
                also["bf16_gemm_variant"] = {"value": round(B / dt, 2), "unit": "sentences/s", "ms_per_step": round(dt * 1e3, 3), "batch": B,
                                             "dtype": "bf16 MFMA operands, fp32 accumulate", "roofline": rb}
            reset()

            # BASELINE.json configs[1]: simple alignment, YELP-shaped batch (64 sentences, len <= 16)
            cfg1 = dict(V=9600, d=256, h=4, N=4, L=16, learnable=False)
            m1 = build_model(cfg1, "cuda")
            st1 = {k: v.detach().clone() for k, v in m1.net.state_dict().items()}
            b1 = [synth_incoming(64, 16, 9600, 7000 + i, "cuda") for i in range(4)]
            f1 = lambda i: m1.f_step(b1[i % 4], 1.0, use_inference_z=True)      # noqa: E731
            dt = timed(f1, min(a.steps, 10))
            also["configs1_simple_L16_bs64"] = {"value": round(64 / dt, 2), "unit": "sentences/s", "ms_per_step": round(dt * 1e3, 3)}
            reset(m1, st1)
            dtg = graphed(m1, f1, 10)
            also["configs1_simple_L16_bs64_graph"] = {"value": round(64 / dtg, 2), "unit": "sentences/s", "ms_per_step": round(dtg * 1e3, 3)}
            del m1, st1
            # BASELINE configs[4] (stress shape: V=32000, d=1024, h=16, 8 layers, len<=128, learnable) at 64 sentences, fp32 and with
            # bf16 MFMA operands, each with the tiled GEMM kernel's roofline fraction at d = 1024
            cfg4 = dict(V=32000, d=1024, h=16, N=8, L=128, learnable=True)
            m4 = build_model(cfg4, "cuda")
            st4 = {k: v.detach().clone() for k, v in m4.net.state_dict().items()}
            b4 = [synth_incoming(64, 128, 32000, 9000 + i, "cuda") for i in range(2)]
            f4 = lambda i: m4.f_step(b4[i % 2], 1.0, use_inference_z=True)      # noqa: E731
            leg = {}
            for prec, peak in (("fp32", PEAK_FP32_MFMA_TFLOPS), ("bf16", PEAK_BF16_MFMA_TFLOPS)):
                _ops.set_gemm_precision(prec)
                try:
                    reset(m4, st4)
                    dt = timed(f4, 3, 2)
                    reset(m4, st4)
                    r4 = gemm_roofline(m4, f4, 2, 2, peak, f"tiled GEMM kernel at d=1024, {prec} MFMA operands")
                finally:
                    _ops.set_gemm_precision(a.precision)
                leg[prec] = {"value": round(64 / dt, 2), "unit": "sentences/s", "ms_per_step": round(dt * 1e3, 3), "roofline": r4}
            also["configs4_stress_L128_bs64"] = leg["fp32"]
            also["configs4_stress_L128_bs64_bf16"] = leg["bf16"]
            del m4, b4, st4
            torch.cuda.empty_cache()

            # north_star (b) names a log-sum-exp alignment DP; the reference has none: the extension kernels
            # (parity unpinned, not part of the step) timed on cost tables of the bench shape and of configs[4]'s
            lse = {}
            for name, (Bq, Tq) in (("B512_T32_S32", (512, 32)), ("B64_T128_S128", (64, 128))):
                g = torch.Generator(device="cuda").manual_seed(3)
                cost = torch.rand(Bq, Tq, Tq, device="cuda", generator=g).requires_grad_(True)
                cost0 = torch.rand(Bq, Tq, device="cuda", generator=g).requires_grad_(True)
                ln = torch.full((Bq,), Tq, device="cuda", dtype=torch.int64)
                for it in range(4):
                    if it == 1:
                        lib.nast_prof_begin(5)
                    _ops.align_lse(cost0, cost, ln, ln, 0.5).sum().backward()
                torch.cuda.synchronize()
                lib.nast_prof_end(None, None, None)
                ms, wk, n = family(5)
                lse[name] = {"fwd_plus_bwd_us": round(ms * 1e3 / 3, 2), "achieved": round(wk / (ms * 1e-3) / 1e9, 1), "unit": "GB/s",
                             "frac_of_hbm_peak": round(wk / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), "launches": n}
            also["alignment_lse_extension"] = dict(lse, note="nast_align_lse_fwd/bwd_f32: latency-bound (one wavefront per sentence, T dependent "
                                                             "steps); extension with no counterpart in the reference, outside the step")
        except Exception as e:      # noqa: BLE001
            import traceback
            traceback.print_exc(file=sys.stderr)
            also["error"] = f"{type(e).__name__}: {e}"[:300]
            model.graph_steps = False
            _ops.set_gemm_precision(a.precision)
            torch.cuda.synchronize()

    if dist_on:
        dist.barrier()
        import nast_amd.ddp as _ddp
        if _ddp.PROF:
            print(f"[rank {rank}] reducer host profile (calls, seconds) {_ddp.prof} stats {model.reducer_F.stats}", file=sys.stderr, flush=True)
        if debug:
            print(f"[rank {rank}] reducer F {model.reducer_F.stats} arena {model.reducer_F.arena is not None} buckets "
                  f"{ {k: [len(b) for b in p.buckets] for k, p in model.reducer_F.plans.items()} }", file=sys.stderr, flush=True)

    if rank == 0:
        sps = B * world * a.steps / elapsed
        key = ("learnable" if learnable else "simple", L)
        out = {
            "metric": "train sentences/sec (fwd+bwd, NAST-StyTrans learnable)" if learnable else "train sentences/sec (fwd+bwd, NAST-StyTrans simple)",
            "value": round(sps, 2), "unit": "sentences/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if a.precision == "fp32" else "bf16", "data": "synthetic",
            "config": {"workload": f"NAST-StyTrans f_step, {a.alignment} alignment, len<={L}, vocab 9600, d=256, h=4, 4 layers, "
                                   f"{B} sentences/GPU (BASELINE configs[{'2] at the per-GPU batch of configs[3' if learnable else '1'}])",
                       "global_batch": B * world, "max_length": L, "parallelism": f"dp{world}"},
        }
        if key in GFLOP_PER_SENTENCE:
            # SURVEY.md section 8d counts every sentence at the padded width.  The packed-row execution does not run the
            # padded positions, so this is the rate the PADDED algorithm would need for the same sentences/s (it can exceed
            # what the matrix cores executed); the executed rate is step_tflops_executed below.
            tf = sps * GFLOP_PER_SENTENCE[key] / 1e3
            out["step_tflops_padded_equivalent"] = round(tf, 2)
        if gemm_flops_per_step:
            tf = gemm_flops_per_step * world / (elapsed / a.steps) / 1e12
            out["step_tflops_executed"] = round(tf, 2)
            out["step_frac_of_fp32_mfma_peak"] = round(tf / (PEAK_FP32_MFMA_TFLOPS * world), 4)
        if roofline:
            out["roofline"] = roofline
        elif roofline_error:
            out["roofline_error"] = roofline_error
        if also:
            out["also"] = also
        if not a.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(model, cfg, B)
            except Exception as e:      # noqa: BLE001
                import traceback
                traceback.print_exc(file=sys.stderr)
                out["cpu_baseline_error"] = f"{type(e).__name__}: {e}"[:300]
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
