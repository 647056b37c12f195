#!/usr/bin/env python3
"""Build tools/libeps_var_<name>.so from csrc/filter_scan.hip with textual patches (old -> new pairs given as a Python
literal file or via the VARIANTS dict below) -- for A/B runs with tools/scan_ab.py.  Usage: build_scan_variant.py name [patchfile]"""
import ast, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "edge-proposal-sets_amd", "csrc")
name = sys.argv[1]
src = open(os.path.join(CSRC, "filter_scan.hip")).read()
defs = [a for a in sys.argv[2:] if a.startswith("-D")]
files = [a for a in sys.argv[2:] if not a.startswith("-D")]
if files:
    for old, new in ast.literal_eval(open(files[0]).read()):
        assert src.count(old) >= 1, old
        src = src.replace(old, new)
tmp = os.path.join(CSRC, "_fs_var_tmp.hip")
open(tmp, "w").write(src)
try:
    subprocess.check_call(["make", "-C", CSRC, "-s", "-j8"])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off"] + defs +
                          ["-c", tmp, "-o", "/tmp/fs_var.o"])
    objs = [os.path.join(CSRC, "build", f) for f in sorted(os.listdir(os.path.join(CSRC, "build")))
            if f.endswith(".o") and f != "filter_scan.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                           os.path.join(ROOT, "tools", f"libeps_var_{name}.so"), "/tmp/fs_var.o"] + objs)
finally:
    os.remove(tmp)
print("built", name)
