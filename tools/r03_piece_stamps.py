#!/usr/bin/env python3
"""Where the one-pass scan kernel's time goes: a diagnostic build of csrc/scan_pieces.hip with per-phase s_memtime sums (the
product source carries no stamps: they are patched into a temporary copy at its section comments) -> tools/libeps_spstamp.so,
then ONE launch over the ppa-like graph and the shares of wave 0's cycles.  `build` only builds (runs on the CPU box);
without arguments it runs (GPU box).  env VARIANT=0|1|2, BAR."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "edge-proposal-sets_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "libeps_spstamp.so")
NAMES = ["column setup (id, bounds)", "plan: paths per window (cut rows + wave sums)", "plan: pieces (wave 0) + barrier",
         "describe (cut gathers, unit scans, descriptors)", "walk (start bits, look-ups, loads, table updates) + barriers", "known edges in (before the walk, no barrier of its own)",
         "table scan + barrier", "ticket hand-over", "(count) pieces", "(count) direct pieces", "(count) rounds walked",
         "  walk: start bits + barrier + word ranks (first range)", "  walk: first look-up + row load until it ARRIVED (vmcnt 0)", "(count) columns", "  walk: table updates (+ further units)", "(count) paths of direct pieces (x parts)",
         "(count) 32-bit direct pieces", "(count) their paths", "(count) packed pieces", "(count) their paths", "(count) 16-bit direct pieces (+ two-word hash pieces x parts)", "(count) their paths",
         "  describe + walk of 32-bit direct pieces", "  describe + walk of packed pieces", "  describe + walk of 16-bit direct (+ two-word hash) pieces",
         "  table scan of 32-bit direct pieces", "  table scan of packed pieces", "  table scan of 16-bit direct (+ two-word hash) pieces",
         "(count) packed: unit groups walked by wave 0", "(count) packed: straggler trips of those (after the all-entries round)", "(count) hash: unit groups walked by wave 0", "(count) hash: straggler trips of those"]

def build():
    s = open(os.path.join(CSRC, "scan_pieces.hip")).read()
    def rep(old, new, count=1):
        nonlocal s
        assert s.count(old) == count, (s.count(old), old)
        s = s.replace(old, new)
    rep('struct sp_params {', '''__device__ unsigned long long g_sp_stamp[32];
#define XS(var) unsigned long long var; asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory")
#define XA(i, a, b) xst[i] += (b) - (a)
extern "C" int eps_debug_piece_stamps(unsigned long long *out32, int reset)
{
    (void)hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_sp_stamp), sizeof(unsigned long long) * 32);
    if (reset) { unsigned long long z[32] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sp_stamp), z, sizeof(z)); }
    return 0;
}
struct sp_params {''')
    rep('    while (t < ncol) {\n', '    unsigned long long xst[32] = {0};\n    while (t < ncol) {\n        XS(t0);\n        xst[13] += 1;\n')
    rep('        if (dv > 0 && v > 0) {\n', '        XS(t1); XA(0, t0, t1);\n        if (dv > 0 && v > 0) {\n')
    rep('            // ---- plan: merge windows into pieces.', '            XS(t2); XA(1, t1, t2);\n            // ---- plan: merge windows into pieces.')
    rep('            const int np = s_np;\n', '            const int np = s_np;\n            XS(t3); XA(2, t2, t3);\n            xst[8] += np;\n')
    rep('                        // ---- describe the round\'s row segments', '                        XS(d0);\n                        xst[10] += 1;\n                        // ---- describe the round\'s row segments')
    rep('                        for (uint32_t ulo = 0; ulo < (uint32_t)total; ulo += SP_UBITS) {', '                        XS(d1); XA(3, d0, d1);\n                        for (uint32_t ulo = 0; ulo < (uint32_t)total; ulo += SP_UBITS) {')
    rep('                            sp_unit fa[SP_G], fb[SP_G];\n                            fetch_group(0, fa);\n', '                            sp_unit fa[SP_G], fb[SP_G];\n                            XS(w1); if (ulo == 0u) XA(11, d1, w1);\n                            fetch_group(0, fa);\n                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n                            XS(w2); XA(12, w1, w2);\n')
    rep('                            }\n                        }\n                            if (uhi < (uint32_t)total) sp_barrier();       // (the next range rewrites the start bits)\n', '                            }\n                            XS(w3); XA(14, w2, w3);\n                        }\n                            if (uhi < (uint32_t)total) sp_barrier();       // (the next range rewrites the start bits)\n')
    rep('                        sp_barrier();        // the next round / the scan follows: descriptors and table updates are complete\n',
        '                        sp_barrier();        // the next round / the scan follows: descriptors and table updates are complete\n                        XS(d2); XA(4, d1, d2); XA(22 + xmode, d0, d2);\n')
    rep('                    // ---- known edges in: the neighbours of v inside the piece take their slots BEFORE the walk', '                    XS(e0);\n                    xst[9] += direct ? 1 : 0;\n                    xst[15] += direct ? ppaths : 0;\n                    const int xmode = (direct && !d16) ? 0 : packed ? 1 : 2;\n                    xst[16 + 2 * xmode] += 1; xst[17 + 2 * xmode] += ppaths;\n                    // ---- known edges in: the neighbours of v inside the piece take their slots BEFORE the walk')
    rep('                    for (int r = 0; r < rounds; ++r) {\n', '                    XS(e01); XA(5, e0, e01);\n                    for (int r = 0; r < rounds; ++r) {\n')
    rep('                    // ---- scan the table: count the candidates', '                    XS(e1);\n                    // ---- scan the table: count the candidates')
    rep('                    new_keys = 0u;\n                    sp_barrier();\n', '                    new_keys = 0u;\n                    sp_barrier();\n                    XS(e2); XA(6, e1, e2); XA(25 + xmode, e1, e2);\n')
    rep('        if (tid == 0) s_ticket = t_next;\n        sp_barrier();\n        t = s_ticket;\n        sp_barrier();\n',
        '        XS(t8);\n        if (tid == 0) s_ticket = t_next;\n        sp_barrier();\n        t = s_ticket;\n        sp_barrier();\n        XS(t9); XA(7, t8, t9);\n')
    rep('    // candidates scored by this workgroup: one atomic per wave\n', '    if (tid == 0)\n        for (int i = 0; i < 32; ++i) atomicAdd(&g_sp_stamp[i], xst[i]);\n    // candidates scored by this workgroup: one atomic per wave\n')
    pat = "                                bool have = false;\n"
    i1 = s.index(pat)
    s = s[:i1] + pat[:-1] + " xst[28] += 1;\n" + s[i1 + len(pat):]
    i2 = s.index(pat, i1 + 40)
    s = s[:i2] + pat[:-1] + " xst[30] += 1;\n" + s[i2 + len(pat):]
    j1 = s.index("if (++tries > 4u * (mask + 2u)) {")
    s = s[:j1] + "xst[29] += 1; " + s[j1:]
    j2 = s.index("if (++tries > 4u * (mask + 2u)) {", j1 + 60)
    s = s[:j2] + "xst[31] += 1; " + s[j2:]
    tmp = os.path.join(CSRC, "_sp_stamp_tmp.hip")
    open(tmp, "w").write(s)
    try:
        subprocess.check_call(["make", "-C", CSRC, "-s", "-j8"])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
                               "-c", tmp, "-o", "/tmp/sp_stamp.o"])
        objs = [os.path.join(CSRC, "build", f) for f in sorted(os.listdir(os.path.join(CSRC, "build")))
                if f.endswith(".o") and f != "scan_pieces.o"]
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, "/tmp/sp_stamp.o"] + objs)
    finally:
        os.remove(tmp)
    print("built", OUT)

def run():
    sys.path.insert(0, ROOT)
    import torch, eps_amd
    from eps_amd import ops, scan, synth, _lib
    from eps_amd.heuristics import node_weight_table
    dev = torch.device("cuda:0")
    g0 = synth.ppa_like(seed=3, device=dev)
    w = node_weight_table(g0, ops.W_AA)
    g, perm = g0.degree_ordered()[:2]
    sc = scan.screen_weights(g0, g, perm, w); fx32, shift, usable = sc.fx32, sc.shift, sc.usable
    bounds, cuts = scan.screen_tables(g)
    order = scan.column_order(g)
    lib = ctypes.CDLL(OUT)
    lib.eps_scan_screen.restype = ctypes.c_int
    lib.eps_scan_screen.argtypes = _lib.SIGNATURES["eps_scan_screen"][1]
    lib.eps_debug_piece_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    variant = int(os.environ.get("VARIANT", "2"))
    packed = os.environ.get("PACKED", "1") == "1"
    bar = float(os.environ.get("BAR", "2.378"))
    buf = (ctypes.c_ulonglong * 32)()
    lib.eps_debug_piece_stamps(buf, 1)
    for rep in range(2):
        res = ops.Survivors(48 << 20, bar, dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if rep == 1:
            lib.eps_debug_piece_stamps(buf, 1)
        e0.record()
        rc = lib.eps_scan_screen(g.rowptr.data_ptr(), g.col.data_ptr(), scan.reverse_positions(g).data_ptr(), fx32.data_ptr(), cuts.data_ptr(),
                                 scan.window_paths(g).data_ptr(), sc.ssum.data_ptr() if packed else None, sc.smax.data_ptr() if packed else None,
                                 sc.plan[0].data_ptr() if packed else None, sc.plan[1].data_ptr() if packed else None, bounds.data_ptr(), g.n_rows, g.nnz(), order.data_ptr(), order.numel(), shift, variant, res.rec.data_ptr(),
                                 status.data_ptr(), torch.cuda.current_stream().cuda_stream)
        e1.record(); torch.cuda.synchronize()
        assert rc == 0
    lib.eps_debug_piece_stamps(buf, 0)
    x = list(buf)
    tot = sum(x[:8])
    print(f"eps_scan_screen variant {variant}{' with packed pieces' if packed else ''} (stamped build) on the ppa-like graph, hubs-first labels, bar {bar}: {e0.elapsed_time(e1):.2f} ms; "
          f"slots, candidates {res.counts()}; s_memtime sums of thread 0 over all workgroups: {tot}")
    for i, nm in enumerate(NAMES):
        print(f"{nm:52s} {x[i]:16d} {100.0 * x[i] / tot:7.2f}%" if (i < 8 or nm.startswith("  ")) else f"{nm:52s} {x[i]:16d}")

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        run()
