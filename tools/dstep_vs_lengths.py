"""d_step time against what the length predictor generates: timed on fresh weights and after 10 / 20 / 30 f_steps (run on the GPU box).
The generated lengths set the width of the discriminator's generated inputs; they change as the generator trains."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
cfg = dict(V=9600, d=256, h=4, N=4, L=32, learnable=True)
torch.cuda.set_device(0)
model = bench.build_model(cfg, "cuda")
batches = [bench.synth_incoming(512, 32, cfg["V"], 1000 + i, "cuda") for i in range(4)]
def td(n=12):
    for i in range(3): model.d_step(batches[i % 4], 1.0, use_inference_z=True)
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(n): model.d_step(batches[i % 4], 1.0, use_inference_z=True)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
def genlen():
    F = model.net.model_F
    with torch.no_grad():
        b = batches[0].data
        out = F(b.sent, b.sent_length, None, None, 1 - b.domain, generate=True, differentiable_decode=True, temperature=1.0, use_inference_z=True)
    return float(out[1].float().mean()), int(out[1].max())
print("fresh weights: d_step %.2f ms, generated lengths mean %.1f max %d" % ((td(),) + genlen()))
for k in range(3):
    for i in range(10): model.f_step(batches[i % 4], 1.0, use_inference_z=True)
    print("after %d f_steps: d_step %.2f ms, generated lengths mean %.1f max %d" % ((10 * (k + 1), td()) + genlen()))
