import sys, torch, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eps_amd
from eps_amd import ops
dev = torch.device("cuda:0")
N, H = 576289, 256
g = torch.Generator(device=dev).manual_seed(0)
for K in (256, 316):
    a = torch.randn(N, K, generator=g, device=dev); w = torch.randn(H, K, generator=g, device=dev); b = torch.randn(H, generator=g, device=dev)
    def t(fn, it=10):
        for _ in range(3): fn()
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it): fn()
        e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it
    fl = 2.0 * N * H * K
    m1 = t(lambda: ops.gemm(a, w, bias=b, relu=True)); m2 = t(lambda: torch.relu(torch.addmm(b, a, w.t()))); m3 = t(lambda: a @ w.t())
    print(f"K={K}: eps gemm+bias+relu {m1:.3f} ms {fl/m1/1e9:.1f} TF | torch addmm+relu {m2:.3f} ms {fl/m2/1e9:.1f} TF | torch mm only {m3:.3f} ms {fl/m3/1e9:.1f} TF")
