// Helpers shared by the pair-scoring kernels (generic and column-run), gfx950.
#pragma once
#include "eps_common.h"

__device__ __forceinline__ int64_t bcast64(int64_t x, int j)
{
    int lo = __builtin_amdgcn_readlane((int)(x & 0xffffffffll), j);
    int hi = __builtin_amdgcn_readlane((int)(x >> 32), j);
    return ((int64_t)hi << 32) | (uint32_t)lo;
}

// Number of elements of the ascending array a[0..n) that are < t.  Same trip count in every
// lane (n is wave-uniform), no divergent branches.
template <typename P>
__device__ __forceinline__ int lower_bound_uniform(P a, int n, int t)
{
    int pos = 0;
    for (int step = 1 << (31 - __builtin_clz(n)); step > 0; step >>= 1) {
        int np = pos + step;
        int idx = (np < n ? np : n) - 1;
        int x = a[idx];
        if (np <= n && x < t) pos = np;
    }
    return pos;
}

template <bool HAS_VAL, bool HAS_W, typename WT>
__device__ __forceinline__ void score_pair_inplace(const int32_t *__restrict__ col, const float *__restrict__ val,
                                                   const WT *__restrict__ node_w, int64_t bu, int32_t du, int64_t bv,
                                                   int32_t dv, int lane, int &cnt, float &acc_cn, WT &acc_ws)
{
    const bool swapped = du > dv;
    const int32_t slen = swapped ? dv : du, llen = swapped ? du : dv;
    const int64_t sbase = swapped ? bv : bu, lbase = swapped ? bu : bv;
    const int32_t *__restrict__ lcol = col + lbase;
    for (int s0 = 0; s0 < slen; s0 += 64) {
        const int si = s0 + lane;
        const bool act = si < slen;
        const int t = act ? col[sbase + si] : 0;
        const int pos = lower_bound_uniform(lcol, llen, t);
        const int pc = pos < llen ? pos : llen - 1;
        const bool found = act && pos < llen && lcol[pc] == t;
        cnt += __popcll(__ballot(found));
        if ((HAS_VAL || HAS_W) && found) {
            float vs = 1.0f, vl = 1.0f;
            if (HAS_VAL) { vs = val[sbase + si]; vl = val[lbase + pc]; }
            const float va = swapped ? vl : vs, vbv = swapped ? vs : vl;
            if (HAS_VAL) acc_cn += va * vbv;
            if (HAS_W) acc_ws += (WT)va * ((WT)vbv * node_w[t]);
        }
    }
}

typedef int v4i __attribute__((ext_vector_type(4)));

// Buffer resource over one adjacency row: raw (stride 0) descriptor with num_records = row bytes, so a
// dwordx4 load needs no exec masking and no bounds branch -- out-of-range dwords come back as 0 (probed on
// gfx950: per-dword range check, 4-byte-aligned bases are fine; tools/probe_bufload.hip).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(const int32_t *row, int32_t len)
{
    return __builtin_amdgcn_make_buffer_rsrc((void *)row, 0, len * 4, 0x00020000);
}

// Sum over the (few) lanes whose flag is set, in ascending lane order: deterministic, and cheaper than a
// 6-step butterfly when 1-3 lanes hold a contribution (mean CN of a candidate pair is ~1.3).
__device__ __forceinline__ int lane_get(int x, int l) { return __builtin_amdgcn_readlane(x, l); }
__device__ __forceinline__ float lane_get(float x, int l)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l));
}
__device__ __forceinline__ double lane_get(double x, int l)
{
    const long long b = __builtin_bit_cast(long long, x);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), l);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
template <typename T>
__device__ __forceinline__ T sparse_lane_sum(uint64_t mask, T x)
{
    T tot = 0;
    while (mask) {
        const int l = __builtin_ctzll(mask);
        mask &= mask - 1;
        tot += lane_get(x, l);
    }
    return tot;
}

