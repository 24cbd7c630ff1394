"""Probe (pure PyTorch): is kernel A -> memset node -> kernel B ordered inside a replayed hipGraph when B depends on A only
through the stream order?  A = a long chain of matmuls producing X, memset = torch.zeros of a large buffer, B = X.clone()."""
import torch
torch.manual_seed(0)
a = torch.randn(2048, 2048, device="cuda:0") / 45.0
x0 = torch.randn(2048, 2048, device="cuda:0")
def body():
    x = x0
    for _ in range(20):
        x = x @ a            # A: ~20 GEMMs
    y = torch.zeros(64 * 1024 * 1024, device="cuda:0")   # 256 MB: memset node
    z = x.clone()            # B: reads A's result
    w = y[:16].clone()
    return z, w
ref, _ = body()
ref = ref.clone()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    body()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    z, w = body()
for rep in range(4):
    x0.mul_(1.0)  # no-op touch
    g.replay(); torch.cuda.synchronize()
    print(rep, "max |z - ref| =", float((z - ref).abs().max()), " zeros ok:", bool((w == 0).all()))
