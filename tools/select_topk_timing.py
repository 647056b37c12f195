#!/usr/bin/env python3
"""eps_select_topk_cut / _rows against the same selection in tensor ops: one small tie-heavy case, then timings at the bench's size."""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eps_amd
from eps_amd import ops, scan
dev=torch.device('cuda:0')
# failing case
for n,k in ((1000,2000),):
    g = torch.Generator().manual_seed(n + k)
    u = torch.randint(0, 1 << 19, (n,), generator=g)
    v = u + 1 + torch.randint(0, 1 << 19, (n,), generator=g)
    keys = torch.unique((v << 32) | u)
    vals = (torch.randint(0, 50, (keys.numel(),), generator=g).float() / 7).contiguous()
    want_k, want_v = scan.select_topk_torch(keys, vals, k)
    for bits in (32, 21):
        got_k, got_v = ops.select_topk(keys.to(dev), vals.to(dev), k, bits)
        print(bits, keys.numel(), got_k.numel(), want_k.numel(), torch.equal(got_k.cpu(), want_k), torch.equal(got_v.cpu(), want_v))
        if not torch.equal(got_k.cpu(), want_k):
            d = (got_k.cpu() != want_k).nonzero().flatten()
            print(" first diffs", d[:5], got_k.cpu()[d[:3]], want_k[d[:3]], got_v.cpu()[d[:3]], want_v[d[:3]])
# timing at bench size
n = 4_850_000; k = 4_000_000
g = torch.Generator(device=dev).manual_seed(1)
u = torch.randint(0, 576289, (n,), generator=g, device=dev); v = torch.randint(0, 576289, (n,), generator=g, device=dev)
keys = (torch.maximum(u,v) << 32) | torch.minimum(u,v); vals = torch.rand(n, generator=g, device=dev) * 5
def T(fn, r=10):
    fn(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(r): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/r*1e3
print("torch", T(lambda: scan.select_topk_torch(keys, vals, k)), "abi bits32", T(lambda: ops.select_topk(keys, vals, k, 32)), "abi bits20", T(lambda: ops.select_topk(keys, vals, k, 20)))
