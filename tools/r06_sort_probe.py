#!/usr/bin/env python3
"""Where the cooperative radix sort's time goes (csrc/tail_sort.hip): the by-u sort of 2.2 M survivor keys and the row sort of
2 M selected pairs against the r05 library path (rocPRIM onesweep), and timing-only ablations of the sort kernel (results wrong by
construction): no per-pass histogram, no ranking + scatter (what is left is the grid hand-overs and the bookkeeping).
`build` builds tools/bin/libeps_ts_<name>.so; `run` times whatever EPS_LIB_PATH points at; `all` runs every variant in
child processes (GPU box)."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "edge-proposal-sets_amd", "csrc")
VARIANTS = {"nohist": ["-DTS_ABL_NOHIST"], "noscatter": ["-DTS_ABL_NOSCATTER"], "syncsonly": ["-DTS_ABL_NOHIST", "-DTS_ABL_NOSCATTER"]}


def build():
    subprocess.check_call(["make", "-C", CSRC, "-s", "-j8"])
    objs = [os.path.join(CSRC, "build", f) for f in sorted(os.listdir(os.path.join(CSRC, "build"))) if f.endswith(".o") and f != "tail_sort.o"]
    os.makedirs(os.path.join(ROOT, "tools", "bin"), exist_ok=True)
    for name, flags in VARIANTS.items():
        obj = f"/tmp/_ts_{name}.o"
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-Wno-unused-result",
                               *flags, "-c", os.path.join(CSRC, "tail_sort.hip"), "-o", obj])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                               os.path.join(ROOT, "tools", "bin", f"libeps_ts_{name}.so"), obj, *objs])
        print("built", name)


def run():
    sys.path.insert(0, ROOT)
    import torch
    import eps_amd
    ops = eps_amd.ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    n, nodes = 2_200_000, 576_289
    hubs = torch.randint(0, 16384, (1529,), generator=g)
    u = hubs[torch.randint(0, 1529, (n,), generator=g)]
    v = torch.randint(16384, nodes, (n,), generator=g)
    keys = torch.unique(((v << 32) | u).to(dev))
    keys = keys[torch.randperm(keys.numel(), device=dev)]
    n = keys.numel()
    n_dev = torch.tensor([n], dtype=torch.int64, device=dev)
    m = 2_000_000
    sk = keys[:m].contiguous()
    sv = (2.87 + torch.empty(m).exponential_(2.0, generator=g)).to(dev)
    m_dev = torch.tensor([m], dtype=torch.int64, device=dev)
    perm = torch.randperm(nodes, generator=g).to(dev)

    def timed(fn, reps=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in ev:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) for a, b in ev)
        return round(t[len(t) // 2], 4)

    out = {"lib": os.environ.get("EPS_LIB_PATH", "default"), "n": n, "m": m}
    out["by_u_radix_ms"] = timed(lambda: ops.radix_sort_by_u(keys, n_dev, 20, 12))
    out["rows_radix_ms"] = timed(lambda: ops.radix_sort_rows(sk, sv, m_dev, 2 * m, 20, perm))
    out["by_u_noblock_radix_ms"] = timed(lambda: ops.radix_sort_by_u(keys, n_dev, 20, 0))
    if out["lib"] == "default":
        out["by_u_library_ms"] = timed(lambda: ops.sort_pairs_by_u(keys, 20, 12))
        out["rows_library_ms"] = timed(lambda: ops.select_rows(sk, sv, 2 * m, 20, perm))
        base = torch.tensor([2.87], device=dev)
        vals = sv

        def sel_new():
            ops.score_hist(sk, vals, m_dev, base)
            ops.score_pick_compact(sk, vals, m_dev, base, m // 2)
        out["select_hist_ms"] = timed(sel_new)
        out["select_fsel_ms"] = timed(lambda: ops.select_compact(sk, vals, m // 2))
        a = ops.radix_sort_by_u(keys, n_dev, 20, 12)
        b = ops.sort_pairs_by_u(keys, 20, 12)
        out["by_u_equal"] = bool(torch.equal(a, b))
    print(json.dumps(out))


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    elif sys.argv[1] == "run":
        run()
    else:
        env = dict(os.environ)
        subprocess.call([sys.executable, __file__, "run"], env=env)
        for name in VARIANTS:
            env["EPS_LIB_PATH"] = os.path.join(ROOT, "tools", "bin", f"libeps_ts_{name}.so")
            subprocess.call([sys.executable, __file__, "run"], env=env)
