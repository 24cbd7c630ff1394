import torch, time
C, N = 64, 204800
a = torch.randn(C, N, device="cuda"); b = torch.randn(C, N, device="cuda")
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): r = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, r
ms, ref = t(lambda: a @ b.t())
print("plain matmul [64,N]@[N,64]: %.3f ms" % ms)
for S in (16, 64, 256, 1024):
    n = N // S
    def f():
        return torch.bmm(a.view(C, S, n).transpose(0, 1), b.view(C, S, n).permute(1, 2, 0)).sum(0)
    ms, r = t(f)
    print("split-K S=%d: %.3f ms  maxdiff %.2e" % (S, ms, (r - ref).abs().max().item() / ref.abs().max().item()))
x = torch.randn(N, 1, device="cuda")
ms, _ = t(lambda: a @ x); print("[64,N]@[N,1]: %.3f ms" % ms)
ms, _ = t(lambda: (a.view(C, 64, N // 64) * x.view(1, 64, N // 64)).sum(-1).sum(-1)); print("  as elementwise+sum: %.3f ms" % ms)
ms, _ = t(lambda: a.sum(dim=1)); print("row sum [64,N]: %.3f ms" % ms)
