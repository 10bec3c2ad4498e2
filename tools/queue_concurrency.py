"""Do kernels of two HIP streams of one process run concurrently on this GPU?  Two streams each run a chain of single-
workgroup spin kernels (torch.cuda._sleep); concurrent queues finish in the time of one chain, serialised ones in two."""
import time
import torch

torch.zeros(1, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
cyc = 2_000_000
def chain(stream, n):
    with torch.cuda.stream(stream):
        for _ in range(n):
            torch.cuda._sleep(cyc)
for label, streams in (("one stream, 20 kernels", (s1,)), ("two streams, 20 kernels each", (s1, s2))):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for s in streams:
        chain(s, 20)
    torch.cuda.synchronize()
    print(label, round((time.perf_counter() - t) * 1e3, 2), "ms", flush=True)
# the same with the default stream as one of the two (the training step's main stream)
torch.cuda.synchronize()
t = time.perf_counter()
chain(torch.cuda.current_stream(), 20); chain(s2, 20)
torch.cuda.synchronize()
print("default + side stream, 20 kernels each", round((time.perf_counter() - t) * 1e3, 2), "ms")
