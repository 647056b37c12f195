#!/usr/bin/env python3
"""Timing-only ablations of the one-pass scan kernel (results are WRONG by construction; each build removes one ingredient so
that its cost shows as the time difference): textual patches on a temporary copy of csrc/scan_pieces.hip -> tools/libeps_abl_<name>.so.
`build` builds all; without arguments it times them against the product library on the ppa-like graph (GPU box)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "edge-proposal-sets_amd", "csrc")
ABL = {
    "sb1": [("#define SP_SB 4 ", "#define SP_SB 1 ")],
    "sb2": [("#define SP_SB 4 ", "#define SP_SB 2 ")],
}

def build():
    subprocess.check_call(["make", "-C", CSRC, "-s", "-j8"])
    objs = [os.path.join(CSRC, "build", f) for f in sorted(os.listdir(os.path.join(CSRC, "build"))) if f.endswith(".o") and f != "scan_pieces.o"]
    for name, patches in ABL.items():
        s = open(os.path.join(CSRC, "scan_pieces.hip")).read()
        for old, new in patches:
            assert s.count(old) == 1, (name, s.count(old), old)
            s = s.replace(old, new)
        tmp = os.path.join(CSRC, "_sp_abl_tmp.hip")
        open(tmp, "w").write(s)
        try:
            subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-c", tmp, "-o", "/tmp/sp_abl.o"])
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(ROOT, "tools", f"libeps_abl_{name}.so"), "/tmp/sp_abl.o"] + objs)
        finally:
            os.remove(tmp)
        print("built", name)

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
    libs = [("product", _lib.load())]
    for name in ABL:
        lib = ctypes.CDLL(os.path.join(ROOT, "tools", f"libeps_abl_{name}.so"))
        lib.eps_scan_screen.restype = ctypes.c_int
        lib.eps_scan_screen.argtypes = _lib.SIGNATURES["eps_scan_screen"][1]
        libs.append((name, lib))
    for variant in [int(x) for x in os.environ.get("VARIANTS", "2,0").split(",")]:
        for name, lib in libs:
            ts = []
            for rep in range(3):
                res = ops.Survivors(48 << 20, 2.378, dev)
                status = torch.zeros(1, dtype=torch.int32, device=dev)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = lib.eps_scan_screen(g.rowptr.data_ptr(), g.col.data_ptr(), scan.reverse_positions(g).data_ptr(), fx32.data_ptr(), cuts.data_ptr(),
                                         scan.window_paths(g).data_ptr(), bounds.data_ptr(), g.n_rows, g.nnz(), order.data_ptr(), order.numel(), shift, variant, res.rec.data_ptr(),
                                         status.data_ptr(), torch.cuda.current_stream().cuda_stream)
                e1.record(); torch.cuda.synchronize()
                assert rc == 0
                ts.append(e0.elapsed_time(e1))
            print(f"variant {variant} {name:18s} {min(ts):7.2f} ms")

if __name__ == "__main__":
    build() if len(sys.argv) > 1 and sys.argv[1] == "build" else run()
