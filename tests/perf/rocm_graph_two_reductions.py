"""Reproducer (pure PyTorch, no libsdeh): on torch 2.10 + ROCm 7.0/7.2 two consecutive multi-block reductions captured into one
hipGraph give a corrupted SECOND result from the second replay on -- the first reduction's semaphore / staging block is reused as
the second one's output (the corrupted words look like its counters / partial sums; kernel -> memset -> kernel ordering as such
is fine, rocm_graph_memset_order.py).  This is why the partial sums of
sdeh_weight_grad are reduced by the library's own kernel (csrc/sdeh_wgrad.hip) and why tests/test_hip_graphs.py compares replayed
gradients with eager ones."""
import torch
torch.manual_seed(0)
x1 = torch.randn(4, 1600, 64, 64, device="cuda:0"); x2 = torch.randn(4, 1600, 64, device="cuda:0")
r1, r2 = x1.sum(1), x2.sum(1)
for variant in ("static_inputs", "fresh_inputs"):
    def body():
        if variant == "fresh_inputs":
            a = torch.empty_like(x1); a.copy_(x1); b = torch.empty_like(x2); b.copy_(x2)
        else:
            a, b = x1, x2
        return a.sum(1), b.sum(1)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        w, b = body()
    for rep in range(3):
        g.replay(); torch.cuda.synchronize()
        print(variant, rep, "w err", float((w - r1).abs().max()), "b err", [float(v) for v in (b - r2).abs().amax(dim=1)])
