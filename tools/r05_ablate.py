#!/usr/bin/env python3
"""Timing-only ablations of the scan kernel WITH skipped heads (results are wrong by construction; each build removes one
ingredient, its cost shows as the time difference): textual patches on a copy of csrc/scan_pieces.hip -> tools/bin/libeps_abl_<name>.so.
`build` builds all; `run` times them (tools/r05_heads_ab.py 0.5 through EPS_LIB_PATH) on the GPU box."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "edge-proposal-sets_amd", "csrc")
CLEAR_ONLY = '''                    {
                        const uint32_t n4c = (((d16 || direct) ? scan_slots : (packed ? scan_slots : 2u * scan_slots)) + 3u) & ~3u;
                        for (uint32_t i = 4u * tid; i < n4c; i += 4u * T) *(uint4 *)(lds + i) = make_uint4(0u, 0u, 0u, 0u);
                    }
                    if (false) {
'''
ABL = {
    # the sweeps only CLEAR the table (write-only): what reading + testing costs
    "clearonly": [("                    if (d16) {\n                        // (two fields per word: the low one shifted up", CLEAR_ONLY + "                    } else if (d16) {\n                        // (two fields per word: the low one shifted up")],
    # the walk looks up and loads, but updates no table
    "noupdate": [("                        auto consume_group = [&](const sp_unit (&f)[SP_G]) {\n", "                        auto consume_group = [&](const sp_unit (&f)[SP_G]) {\n                            { int z = 0;\n#pragma unroll\n                              for (int q = 0; q < SP_G; ++q) z += f[q].u4[0] + f[q].u4[1] + f[q].u4[2] + f[q].u4[3] + (int)f[q].fx + f[q].nvalid;\n                              if (z == 0x7ffffff1) lds[0] = 1u; return; }\n")],
    # no walk at all: describe, barriers, sweeps
    "nowalk": [("                            const int n_iter = (int)(uhi - ulo + T - 1) / T;      // uniform over the workgroup", "                            const int n_iter = (int)(uhi - ulo) < 0 ? 1 : 0;      // uniform over the workgroup"),
               ("                            sp_unit fa[SP_G], fb[SP_G];\n                            fetch_group(0, fa);\n                            if (n_iter <= SP_G) {", "                            sp_unit fa[SP_G], fb[SP_G];\n                            if (n_iter > 0) fetch_group(0, fa);\n                            if (n_iter == 0) {} else if (n_iter <= SP_G) {")],
    # the packed walk without its straggler loop (entries that missed two probes are dropped): what deferring them could save at most
    "nostraggle": [("                                uint32_t ch = 0u, cstep = 0u, clid = 0u, cfx = 0u, tries = 0u;\n                                bool have = false;\n                                while (__ballot(have || pend != 0u)) {", "                                uint32_t ch = 0u, cstep = 0u, clid = 0u, cfx = 0u, tries = 0u;\n                                bool have = false;\n                                pend = 0u;\n                                while (__ballot(have || pend != 0u)) {")],
    # (correct results) three / two workgroups per CU instead of four: what the launch loses per workgroup slot given to something else
    "percu3": [("sp_per_cu[3] = {2, 1, 4};", "sp_per_cu[3] = {2, 1, 3};")],
    "percu2": [("sp_per_cu[3] = {2, 1, 4};", "sp_per_cu[3] = {2, 1, 2};")],
    # kind-specialised bodies (timing + registers; results incomplete by construction): only the direct pieces / only the packed ones
    # (the skip sits behind the per-row segment bookkeeping, so the pieces that ARE processed see the segments they would)
    "directonly": [("                const bool packed = !HV && (info >> 30) == 1u;\n", "                const bool packed = false;\n"),
                   ("                for (uint32_t part = 0; part < parts; ++part) {\n", "                if (!direct) continue;\n                __builtin_assume(direct);\n                for (uint32_t part = 0; part < parts; ++part) {\n")],
    "packedonly": [("                for (uint32_t part = 0; part < parts; ++part) {\n", "                if (!packed) continue;\n                __builtin_assume(packed && !direct && !d16);\n                for (uint32_t part = 0; part < parts; ++part) {\n")],
    # no known-edge marking
    "noknown": [("                    for (int j = single ? tid : na + tid; j < nb; j += T) {", "                    for (int j = single ? tid : na + tid; j < nb && j < 0; j += T) {")],
}


def build():
    subprocess.check_call(["make", "-C", CSRC, "-s", "-j8"])
    objs = [os.path.join(CSRC, "build", f) for f in sorted(os.listdir(os.path.join(CSRC, "build"))) if f.endswith(".o") and f != "scan_pieces.o"]
    os.makedirs(os.path.join(ROOT, "tools", "bin"), exist_ok=True)
    for name, patches in ABL.items():
        if sys.argv[2:] and name not in sys.argv[2:]:
            continue
        s = open(os.path.join(CSRC, "scan_pieces.hip")).read()
        for old, new in patches:
            assert s.count(old) == 1, (name, s.count(old), old[:60])
            s = s.replace(old, new)
        tmp = os.path.join(CSRC, "_sp_abl_tmp.hip")
        open(tmp, "w").write(s)
        try:
            subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-Wno-unused-result", "-c", tmp, "-o", "/tmp/sp_abl.o"])
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(ROOT, "tools", "bin", f"libeps_abl_{name}.so"), "/tmp/sp_abl.o"] + objs)
        finally:
            os.remove(tmp)
        print("built", name, flush=True)


def run():
    for name in ["hip"] + [n for n in ABL if not sys.argv[2:] or n in sys.argv[2:]]:
        env = dict(os.environ)
        if name != "hip":
            env["EPS_LIB_PATH"] = os.path.join(ROOT, "tools", "bin", f"libeps_abl_{name}.so")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "r05_heads_ab.py"), "0.5"], env=env, capture_output=True, text=True)
        rows = [json.loads(l) for l in r.stdout.splitlines() if l.startswith('{"beta"')]
        if not rows:
            print(name, "FAILED", r.stdout[-500:], r.stderr[-1500:])
            continue
        d = rows[-1]
        print(f"{name:10s} kernel min {d['kernel_min_ms']:7.3f} ms  median {d['kernel_median_ms']:7.3f}  walked slots {d['walked_slots']}  pieces {d['pieces']}", flush=True)


if __name__ == "__main__":
    build() if sys.argv[1:2] == ["build"] else run()
