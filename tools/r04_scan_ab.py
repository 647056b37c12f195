#!/usr/bin/env python3
"""Same-box A/B of several builds of the library (tools/build_lib_variant.sh -> tools/bin/libeps_<name>.so): for every name
one child process (EPS_LIB_PATH) runs the production main launch (eps_scan_screen, plan table, packed pieces, hubs-first labels)
over the whole ppa-like graph at a fixed bar -- kernel ms (HIP events: min / median of REPS), the kernel's candidate count, and
a digest of the survivor list after exact re-scoring, which must be the same for every build.
usage: r04_scan_ab.py name [name ...]   ('hip' = the in-tree libeps_hip.so); env: BAR, REPS, NODES, EDGES, ROUNDS (interleaved passes)"""
import hashlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker():
    sys.path.insert(0, ROOT)
    import torch, eps_amd
    from eps_amd import ops, scan, synth
    from eps_amd.heuristics import node_weight_table
    dev = torch.device("cuda:0")
    g0 = synth.ppa_like(seed=3, device=dev, n_nodes=int(os.environ.get("NODES", 576289)), n_undirected=int(os.environ.get("EDGES", 21231931)))
    w = node_weight_table(g0, ops.W_AA)
    g, perm = g0.degree_ordered()[:2]
    order = scan.column_order(g)
    bar = float(os.environ.get("BAR", "2.378"))
    reps = int(os.environ.get("REPS", "6"))
    sc = scan.screen_weights(g0, g, perm, w)
    bounds, cuts = scan.screen_tables(g)
    variant = scan.screen_variant(g)
    ts = []
    for _ in range(reps):
        res = ops.Survivors(48 << 20, bar, dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.scan_screen(g.rowptr, g.col, scan.reverse_positions(g), sc.fx32, cuts, bounds, g.n_rows, order, sc.shift, res, status, variant,
                        wpaths=scan.window_paths(g), ssum=sc.ssum, smax=sc.smax, plan=sc.plan)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    slots, ncand = res.counts()
    keys, vals = res.valid(slots)
    k2, v2 = scan.rescore_exact(g, sc, keys, torch.tensor([bar], device=dev))
    m = k2 >= 0
    o = torch.argsort(k2[m])
    digest = hashlib.sha256(k2[m][o].cpu().numpy().tobytes() + v2[m][o].cpu().numpy().tobytes()).hexdigest()[:16]
    ts.sort()
    print(json.dumps({"lib": os.environ.get("EPS_LIB_PATH", "in-tree"), "min_ms": round(ts[0], 3), "median_ms": round(ts[len(ts) // 2], 3),
                      "candidates": ncand, "screened": int(keys.numel()), "exact": int(m.sum()), "status": int(status), "digest": digest,
                      "pieces": int(sc.plan[1].shape[0]) if sc.plan else None}))


if __name__ == "__main__":
    if os.environ.get("R04_AB_WORKER"):
        worker()
        sys.exit(0)
    names = sys.argv[1:] or ["hip"]
    rounds = int(os.environ.get("ROUNDS", "1"))
    out = []
    for rd in range(rounds):
        for n in names:
            env = dict(os.environ, R04_AB_WORKER="1")
            if n != "hip":
                env["EPS_LIB_PATH"] = os.path.join(ROOT, "tools", "bin", f"libeps_{n}.so")
            r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(n, "FAILED", r.stdout[-2000:], r.stderr[-3000:])
                continue
            d = json.loads(line[-1]); d["name"] = n
            out.append(d)
            print(f"{n:14s} min {d['min_ms']:7.3f} ms  median {d['median_ms']:7.3f}  candidates {d['candidates']}  screened {d['screened']}  "
                  f"exact {d['exact']}  status {d['status']}  pieces {d['pieces']}  digest {d['digest']}", flush=True)
    if out:
        same = len({(d["digest"], d["candidates"]) for d in out}) == 1
        print("all lists identical:", same)
