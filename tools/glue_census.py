"""Which torch (ATen) kernels one f_step still launches, and from where: every ATen op that launches a device kernel inside one
step, grouped by (op, input shapes, enclosing autograd node or Python frame).  Run on the GPU box: python tools/glue_census.py"""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from torch.profiler import profile, ProfilerActivity

cfg = dict(V=9600, d=256, h=4, N=4, L=32, learnable=True)
torch.cuda.set_device(0)
model = bench.build_model(cfg, "cuda")
batches = [bench.synth_incoming(512, 32, cfg["V"], 1000 + i, "cuda") for i in range(2)]
for i in range(3):
    model.f_step(batches[i % 2], 1.0, use_inference_z=True)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    model.f_step(batches[0], 1.0, use_inference_z=True)
    torch.cuda.synchronize()
agg = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or not ev.kernels:
        continue
    if any(c.name.startswith("aten::") and c.kernels for c in ev.cpu_children):      # count the innermost launching op
        continue
    p, where = ev.cpu_parent, "?"
    while p is not None:
        if "evaluate_function" in p.name or p.name.endswith("Backward") or "Fn" in p.name:
            where = p.name
            break
        p = p.cpu_parent
    if where == "?" and ev.stack:
        fr = [s for s in ev.stack if "nast_amd" in s]
        where = fr[0].split("nast_amd/")[-1] if fr else ev.stack[0][-60:]
    kn = ev.kernels[0].name.split("<")[0][-40:]
    agg[(ev.name, str(ev.input_shapes)[:70], where[:70], kn)] += 1
for (name, shapes, where, kn), n in sorted(agg.items(), key=lambda kv: -kv[1]):
    print(f"{n:4d} {name:22s} {shapes:70s} {where:70s} {kn}")
