// What the scan's translation units share (scan_pieces.hip: the piece kernel and its per-graph tables; scan_heads.hip: skipped
// heads -- the head table, the hub adjacency bitmaps, the completion of the walked sums).
#pragma once
#include "eps_common.h"

#define SP_M 32                 // id windows per graph (one 64-byte row of cuts per node); 48 windows: 1.99 M instead of 2.03 M pieces, -0.5 % (r04)
#define SP_FLAG 0x80000000u      // value word of a KNOWN EDGE's endpoint (put in before the walk): sums stay below 2^31, so the bit survives them

#if defined(__HIPCC__)
// The bar in the table's domain.  filter_scan.hip keeps a candidate when its 2^-40 fixed-point sum a satisfies
// float(a * 2^-40) > threshold, i.e. a >= thr_fix (monotone: found by bisection); a screening sum s >= a / 2^(40 - shift),
// so s >= floor(thr_fix / 2^(40 - shift)) holds for every such candidate.  Any bar <= 0 (or -inf): every candidate (1).
// +inf / NaN: nothing passes (SP_FLAG: sums stay below 2^31).
__device__ __forceinline__ uint32_t sp_bar_units(float thr, int shift)
{
    auto above = [&](long long a) { return (float)((double)a * (1.0 / (double)(1ll << 40))) > thr; };
    if (!above(0x7fffffffffffffffll)) return SP_FLAG;
    if (above(0ll)) return 1u;
    long long lo = 1ll, hi = 0x7fffffffffffffffll;      // smallest positive a with above(a)
    while (lo < hi) {
        const long long mid = lo + ((hi - lo) >> 1);
        if (above(mid)) hi = mid; else lo = mid + 1;
    }
    const unsigned long long q = (unsigned long long)lo >> (40 - shift);
    return q >= (unsigned long long)SP_FLAG ? SP_FLAG : (q ? (uint32_t)q : 1u);
}
#endif
