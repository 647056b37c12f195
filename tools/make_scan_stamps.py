#!/usr/bin/env python3
"""Diagnostic build of csrc/filter_scan.hip with per-phase s_memtime sums -> tools/libeps_fsstamp.so (git-ignored).
The product source carries no stamps: this script inserts them into a temporary copy, at the phase boundaries named by
the source's own section comments, and links it with the product's other objects."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "edge-proposal-sets_amd", "csrc")
s = open(os.path.join(CSRC, "filter_scan.hip")).read()
def rep(old, new):
    global s
    assert s.count(old) == 1, old
    s = s.replace(old, new)
rep('struct fs_params {', '''__device__ unsigned long long g_fs_stamp[16];
#define XS(var) unsigned long long var; asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory")
#define XA(i, a, b) xst[i] += (b) - (a)
extern "C" int eps_debug_scan_stamps(unsigned long long *out16, int reset)
{
    (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_fs_stamp), sizeof(unsigned long long) * 16);
    if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fs_stamp), z, sizeof(z)); }
    return 0;
}
struct fs_params {''')
rep('''    while (v_cur >= 0) {
        const int32_t v = v_cur;''', '''    unsigned long long xst[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    while (v_cur >= 0) {
        XS(t0);
        const int32_t v = v_cur;''')
rep('''        int j = 0, c = 0;
        bool single = false;''', '''        XS(t1); XA(0, t0, t1);
        int j = 0, c = 0;
        bool single = false;''')
rep('''        fs_barrier();
        for (int k = tid; k < dv; k += FS_THREADS) {   // known edges out''', '''        fs_barrier();
        XS(t2); XA(1, t1, t2);
        for (int k = tid; k < dv; k += FS_THREADS) {   // known edges out''')
rep('''        // ---- plan: id ranges -> tiles''', '''        XS(t3); XA(2, t2, t3);
        // ---- plan: id ranges -> tiles''')
rep('''        const int n_tiles = s_ntiles;
''', '''        const int n_tiles = s_ntiles;
        XS(t4); XA(3, t3, t4);
        xst[14] += 1; xst[15] += single ? 1 : 0;
''')
rep('''            if (direct) {
                if (tid == 0) reserve_out(t_lo);''', '''            XS(t5);
            if (direct) {
                if (tid == 0) reserve_out(t_lo);''')
rep('''            t_lo = t_hi;
        }''', '''            XS(t6); XA(5, t5, t6); XA(direct ? 6 : 7, t4, t6);
            t_lo = t_hi;
        }
        XS(t7); XA(4, t4, t7);''')
rep('''        v_nx = v_nx2;
    }
}
''', '''        v_nx = v_nx2;
    }
    if (tid == 0)
        for (int i = 0; i < 16; ++i) atomicAdd(&g_fs_stamp[i], xst[i]);
}
''')
# inside record-mode D2, per tile: accumulate / wait at the barrier / scan / wait at the barrier; and the D1 walk
rep('''                for (int t = t_lo; t < t_hi; ++t) {
                    const uint32_t cb0 = b0, cn = n;''', '''                XS(d0); XA(8, t5, d0);
                for (int t = t_lo; t < t_hi; ++t) {
                    XS(da);
                    xst[13] += 1;
                    const uint32_t cb0 = b0, cn = n;''')
rep('''                    if (tid == 0) reserve_out(t);
                    fs_barrier();               // sums complete
                    scan_tile(t);
                    fs_barrier();               // accumulators zero again''', '''                    if (tid == 0) reserve_out(t);
                    XS(db); XA(9, da, db);
                    fs_barrier();               // sums complete
                    XS(dc); XA(10, db, dc);
                    scan_tile(t);
                    XS(dd); XA(11, dc, dd);
                    fs_barrier();               // accumulators zero again
                    XS(de); XA(12, dd, de);''')
tmp = os.path.join(CSRC, "_fs_stamp_tmp.hip")
open(tmp, "w").write(s)
try:
    subprocess.check_call(["make", "-C", CSRC, "-s", "-j8"])
    obj = os.path.join(tempfile.gettempdir(), "fs_stamp.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
                           "-c", tmp, "-o", obj])
    objs = [os.path.join(CSRC, "build", f) for f in sorted(os.listdir(os.path.join(CSRC, "build")))
            if f.endswith(".o") and f != "filter_scan.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                           os.path.join(ROOT, "tools", "libeps_fsstamp.so"), obj] + objs)
finally:
    os.remove(tmp)
print("built tools/libeps_fsstamp.so")
