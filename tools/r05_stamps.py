#!/usr/bin/env python3
"""Where wave 0 of the scan kernel spends its cycles (skipped heads, beta 0.5): a diagnostic build of csrc/scan_pieces.hip with
per-phase s_memtime sums patched into a temporary copy -> tools/bin/libeps_stamps.so; `build` builds, `run` runs ONE launch over
the bench graph (GPU box) and prints the shares.  env BAR."""
import ctypes, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "edge-proposal-sets_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "bin", "libeps_stamps.so")
NAMES = ["column set-up (ticket, header, rows, plan records) until the piece loop", "piece preamble + known edges in", "describe until its barrier has passed",
         "walk: start bits + word ranks", "walk: first look-up + row load until it ARRIVED", "walk: table updates, further units, end barrier",
         "table sweep + barrier", "ticket hand-over / column end", "(count) columns", "(count) pieces", "(count) columns with pieces",
         "(count) dead or empty columns", "  of the walk: wave 0 waiting at the walk's end barrier", "  walk (all of it) in PACKED pieces", "  walk (all of it) in direct pieces",
         "(count) packed pieces"]


def build():
    s = open(os.path.join(CSRC, "scan_pieces.hip")).read()

    def rep(old, new):
        nonlocal s
        assert s.count(old) == 1, (s.count(old), old)
        s = s.replace(old, new)
    rep('struct sp_params {', '''__device__ unsigned long long g_sp_stamp[16];
#define XS(var) unsigned long long var; asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory")
#define XA(i, a, b) xst[i] += (b) - (a)
extern "C" int eps_debug_piece_stamps(unsigned long long *out16, int reset)
{
    (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_sp_stamp), sizeof(unsigned long long) * 16);
    if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sp_stamp), z, sizeof(z)); }
    return 0;
}
struct sp_params {''')
    rep('    while (t < ncol) {\n', '    unsigned long long xst[16] = {0};\n    while (t < ncol) {\n        XS(t0);\n        xst[8] += 1;\n')
    rep('            const int np = s_np;\n', '            const int np = s_np;\n            XS(t3); XA(0, t0, t3);\n            xst[9] += np; xst[10] += 1;\n')
    rep('                const int k0 = s_pk0[pi], k1 = s_pk1[pi];\n', '                XS(e0);\n                const int k0 = s_pk0[pi], k1 = s_pk1[pi];\n')
    rep('                    for (int r = 0; r < rounds; ++r) {\n                        // ---- describe', '                    XS(e01); XA(1, e0, e01);\n                    for (int r = 0; r < rounds; ++r) {\n                        XS(d0);\n                        // ---- describe')
    rep('                        const int total = (int)(s_alloc >> 32);\n', '                        XS(d1); XA(2, d0, d1);\n                        const int total = (int)(s_alloc >> 32);\n')
    rep('                            sp_unit fa[SP_G], fb[SP_G];\n                            fetch_group(0, fa);\n', '                            sp_unit fa[SP_G], fb[SP_G];\n                            XS(w1); if (ulo == 0u) XA(3, d1, w1);\n                            fetch_group(0, fa);\n                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n                            XS(w2); XA(4, w1, w2);\n')
    rep('                        sp_barrier();        // the next round / the scan follows: descriptors and table updates are complete\n',
        '                        XS(w3);\n                        sp_barrier();        // the next round / the scan follows: descriptors and table updates are complete\n                        XS(d2); XA(5, w2x, d2); XA(12, w3, d2); XA(packed ? 13 : 14, d1, d2); xst[15] += packed ? 1 : 0;\n')
    rep('                            const int n_iter = (int)(uhi - ulo + T - 1) / T;      // uniform over the workgroup\n', '                            const int n_iter = (int)(uhi - ulo + T - 1) / T;      // uniform over the workgroup\n')
    # w2 lives in an inner scope: carry it out
    rep('                        const int total = (int)(s_alloc >> 32);\n', '                        const int total = (int)(s_alloc >> 32);\n                        unsigned long long w2x = d1;\n')
    rep('                            XS(w2); XA(4, w1, w2);\n', '                            XS(w2); XA(4, w1, w2); w2x = w2;\n')
    rep('                    // ---- scan the table: count the candidates', '                    XS(e1);\n                    // ---- scan the table: count the candidates')
    rep('                    new_keys = 0u;\n                    sp_barrier();\n', '                    new_keys = 0u;\n                    sp_barrier();\n                    XS(e2); XA(6, e1, e2);\n')
    rep('        if (!last_of_ticket) {\n', '        XS(t8);\n        if (!(dv > 0 && v > 0 && !bad_head)) { xst[11] += 1; XA(0, t0, t8); }\n        if (!last_of_ticket) {\n')
    rep('            sp_barrier();                    // (a column without pieces has no barrier of its own: s_np and the piece arrays change hands here)\n            continue;\n',
        '            sp_barrier();                    // (a column without pieces has no barrier of its own: s_np and the piece arrays change hands here)\n            { XS(t9); XA(7, t8, t9); }\n            continue;\n')
    rep('        t_end = t + (tk < batch_from ? 1u : SP_BATCH);\n    }\n', '        t_end = t + (tk < batch_from ? 1u : SP_BATCH);\n        { XS(t9); XA(7, t8, t9); }\n    }\n')
    rep('    // candidates scored by this workgroup: one atomic per wave\n', '    if (tid == 0)\n        for (int i = 0; i < 16; ++i) atomicAdd(&g_sp_stamp[i], xst[i]);\n    // candidates scored by this workgroup: one atomic per wave\n')
    tmp = os.path.join(CSRC, "_sp_stamp_tmp.hip")
    open(tmp, "w").write(s)
    subprocess.check_call(["make", "-C", CSRC, "-s", "-j8"])
    objs = [os.path.join(CSRC, "build", f) for f in sorted(os.listdir(os.path.join(CSRC, "build"))) if f.endswith(".o") and f != "scan_pieces.o"]
    try:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-Wno-unused-result", "-c", tmp, "-o", "/tmp/sp_stamp.o"])
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, "/tmp/sp_stamp.o"] + objs)
    finally:
        os.remove(tmp)
    print("built", OUT)


def run():
    os.environ["EPS_LIB_PATH"] = OUT
    sys.path.insert(0, ROOT)
    import torch, eps_amd
    from eps_amd import ops, scan, synth, _lib
    from eps_amd.heuristics import node_weight_table
    dev = torch.device("cuda:0")
    g0 = synth.ppa_like(seed=3, device=dev)
    w = node_weight_table(g0, ops.W_AA)
    g, perm = g0.degree_ordered()[:2]
    g._cache["scan_calls"] = 2
    order = scan.column_order(g)
    sc = scan.screen_weights(g0, g, perm, w)
    bounds, cuts = scan.screen_tables(g)
    bar = float(os.environ.get("BAR", 2.8758351802825928))
    ht = scan.head_tables(g, sc, scan.head_budget(bar * 2.0 ** sc.shift))
    lib = _lib.load()
    lib.eps_debug_piece_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    buf = (ctypes.c_ulonglong * 16)()
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    for rep_i in range(2):
        lib.eps_debug_piece_stamps(buf, 1)
        wk = ops.Survivors(256 << 20, bar, dev, prefill=False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.scan_screen(g.rowptr, g.col, scan.reverse_positions(g), sc.fx32, cuts, bounds, g.n_rows, order, sc.shift, wk, status, scan.screen_variant(g),
                        wpaths=ht.wpaths, ssum=sc.ssum, smax=sc.smax, plan=ht.plan, heads=ht.heads, batch_from=scan.batch_from(g, order), rowrec=sc.rowrec,
                        colrec=scan.column_records(g, sc, order, ht.plan, ht.heads, ht.live, 'stamps'))
        e1.record(); torch.cuda.synchronize()
        lib.eps_debug_piece_stamps(buf, 0)
    vals = list(buf)
    tot = sum(vals[:8])
    print(json.dumps({"kernel_ms_with_stamps": e0.elapsed_time(e1), "bar": bar, "pieces_plan": int(ht.plan[1].shape[0])}))
    for i, n in enumerate(NAMES):
        print(f"{n:75s} {vals[i]:16d}" + (f"  {100.0 * vals[i] / tot:5.1f} %" if i < 8 else ""))


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
