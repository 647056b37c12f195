// Pair scores (Common Neighbours / Adamic-Adar / Resource Allocation) by CSR neighbour-list
// intersection, gfx950.
//
// Replaces, per pair (u,v):  np.sum(A[src].multiply(A_[dst]), 1)   adamic_utils.py:22,
// train_and_eval.py:212 and  adj[e0] (.) adj[e1] -> sparse row-sum   models.py:536-542.
//
// Work decomposition (wave = 64 lanes):
//   * a wave owns chunks of 64 consecutive pairs.  Lane i fetches pair i's (u,v) and the four
//     rowptr words in parallel -- one coalesced metadata fetch per 64 pairs instead of a
//     dependent scalar chain per pair -- and lane i finally stores pair i's results, so the
//     output stores are coalesced too.
//   * the 64 pairs are then scored one after the other by the whole wave: the LONGER adjacency
//     row is staged in LDS with coalesced loads (PI_CAP entries per pass), the SHORTER row is
//     spread one element per lane and every lane runs a branch-free lower_bound over the staged
//     row.  Matches are counted with ballot+popcount (no reduction) and the weighted sums go
//     through a DPP butterfly.
//   * very lopsided pairs (long row >> short row) skip the staging and search the long row in
//     place (L2-resident binary search), so a degree-100k hub costs log2(d) probes per element
//     of the short row instead of a full read.
// HBM traffic is the algorithmic minimum: both rows once, coalesced; rowptr/pair/outputs once.
#include "eps_common.h"

#include <atomic>

#define PI_WAVES 4          // waves per workgroup
#define PI_CAP 1024         // long-row entries staged per wave and pass (4 KiB of LDS per wave)
#define PI_INPLACE_RATIO 32 // long row searched in place when long > PI_CAP && long >= ratio*short

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
__global__ __launch_bounds__(PI_WAVES * 64) void pair_scores_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, const float *__restrict__ val,
    const WT *__restrict__ node_w, const int32_t *__restrict__ pu, const int32_t *__restrict__ pv,
    int64_t n_pairs, int32_t *__restrict__ out_count, float *__restrict__ out_cn, WT *__restrict__ out_ws)
{
    __shared__ int32_t s_rows[PI_WAVES][PI_CAP];
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    int32_t *L = s_rows[wib];

    const int64_t n_chunks = (n_pairs + 63) >> 6;
    const int64_t wave0 = (int64_t)blockIdx.x * PI_WAVES + wib;
    const int64_t n_waves = (int64_t)gridDim.x * PI_WAVES;

    for (int64_t chunk = wave0; chunk < n_chunks; chunk += n_waves) {
        const int64_t p = chunk * 64 + lane;
        const bool valid = p < n_pairs;
        const int32_t nu = valid ? pu[p] : 0, nv = valid ? pv[p] : 0;
        int64_t ub = rowptr[nu], vb = rowptr[nv];
        int32_t du = valid ? (int32_t)(rowptr[nu + 1] - ub) : 0;
        int32_t dv = valid ? (int32_t)(rowptr[nv + 1] - vb) : 0;

        int32_t my_count = 0;
        float my_cn = 0.0f;
        WT my_ws = 0;

        const int64_t rem = n_pairs - chunk * 64;
        const int here = rem < 64 ? (int)rem : 64;
        for (int j = 0; j < here; ++j) {
            const int32_t dju = __builtin_amdgcn_readlane(du, j);
            const int32_t djv = __builtin_amdgcn_readlane(dv, j);
            if (dju == 0 || djv == 0) continue;  // wave-uniform
            const int64_t bju = bcast64(ub, j), bjv = bcast64(vb, j);
            const bool swapped = dju > djv;  // short row = v
            const int32_t slen = swapped ? djv : dju, llen = swapped ? dju : djv;
            const int64_t sbase = swapped ? bjv : bju, lbase = swapped ? bju : bjv;
            const int32_t *__restrict__ lcol = col + lbase;
            const int32_t *__restrict__ scol = col + sbase;

            int cnt = 0;  // wave-uniform (ballot popcounts)
            float acc_cn = 0.0f;
            WT acc_ws = 0;

            const bool inplace = llen > PI_CAP && (int64_t)llen >= (int64_t)slen * PI_INPLACE_RATIO;
            if (inplace) {
                for (int s0 = 0; s0 < slen; s0 += 64) {
                    const int si = s0 + lane;
                    const bool act = si < slen;
                    const int t = act ? scol[si] : 0;
                    const int pos = lower_bound_uniform(lcol, llen, t);
                    const int pc = pos < llen ? pos : llen - 1;
                    const bool found = act && pos < llen && lcol[pc] == t;
                    cnt += __popcll(__ballot(found));
                    if ((HAS_VAL || HAS_W) && found) {
                        float vs = 1.0f, vl = 1.0f;
                        if (HAS_VAL) { vs = val[sbase + si]; vl = val[lbase + pc]; }
                        const float va = swapped ? vl : vs, vbv = swapped ? vs : vl;  // va = A[u,w], vbv = A[v,w]
                        if (HAS_VAL) acc_cn += va * vbv;
                        if (HAS_W) {
                            const WT scaled = (WT)vbv * node_w[t];  // the A_ entry (adamic_utils.py:17)
                            acc_ws += (WT)va * scaled;
                        }
                    }
                }
            } else {
                for (int l0 = 0; l0 < llen; l0 += PI_CAP) {
                    const int n = (llen - l0) < PI_CAP ? (llen - l0) : PI_CAP;
                    __builtin_amdgcn_wave_barrier();
                    for (int i = lane; i < n; i += 64) L[i] = lcol[l0 + i];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    const bool multipass = llen > PI_CAP;
                    int first = 0, last = 0;
                    if (multipass) { first = L[0]; last = L[n - 1]; }
                    for (int s0 = 0; s0 < slen; s0 += 64) {
                        const int si = s0 + lane;
                        bool act = si < slen;
                        const int t = act ? scol[si] : 0;
                        if (multipass) act = act && t >= first && t <= last;
                        const int pos = lower_bound_uniform(L, n, t);
                        const int pc = pos < n ? pos : n - 1;
                        const bool found = act && pos < n && L[pc] == t;
                        cnt += __popcll(__ballot(found));
                        if ((HAS_VAL || HAS_W) && found) {
                            float vs = 1.0f, vl = 1.0f;
                            if (HAS_VAL) { vs = val[sbase + si]; vl = val[lbase + l0 + pc]; }
                            const float va = swapped ? vl : vs, vbv = swapped ? vs : vl;
                            if (HAS_VAL) acc_cn += va * vbv;
                            if (HAS_W) {
                                const WT scaled = (WT)vbv * node_w[t];
                                acc_ws += (WT)va * scaled;
                            }
                        }
                    }
                }
            }

            if (cnt != 0) {  // wave-uniform
                float r_cn = (float)cnt;
                if (HAS_VAL) r_cn = eps_wave_sum(acc_cn);
                WT r_ws = 0;
                if (HAS_W) r_ws = eps_wave_sum(acc_ws);
                if (lane == j) { my_count = cnt; my_cn = r_cn; my_ws = r_ws; }
            }
        }

        if (valid) {
            if (out_count) out_count[p] = my_count;
            if (out_cn) out_cn[p] = my_cn;
            if (HAS_W && out_ws) out_ws[p] = my_ws;
        }
    }
}

template <typename WT>
static int launch_pair_scores(const int64_t *rowptr, const int32_t *col, const float *val, const WT *node_w,
                              const int32_t *u, const int32_t *v, int64_t n_pairs, int32_t *count, float *cn,
                              WT *wsum, hipStream_t stream)
{
    if (n_pairs == 0) return EPS_OK;
    const int64_t n_chunks = (n_pairs + 63) / 64;
    int64_t blocks = (n_chunks + PI_WAVES - 1) / PI_WAVES;
    const int64_t max_blocks = (int64_t)eps_num_cus() * 8;  // 32 waves per CU
    if (blocks > max_blocks) blocks = max_blocks;
    dim3 grid((unsigned)blocks), block(PI_WAVES * 64);
    const bool hv = val != nullptr, hw = node_w != nullptr && wsum != nullptr;
#define PI_LAUNCH(HV, HW) \
    hipLaunchKernelGGL((pair_scores_kernel<HV, HW, WT>), grid, block, 0, stream, rowptr, col, val, node_w, u, v, \
                       n_pairs, count, cn, wsum)
    if (hv && hw) PI_LAUNCH(true, true);
    else if (hv) PI_LAUNCH(true, false);
    else if (hw) PI_LAUNCH(false, true);
    else PI_LAUNCH(false, false);
#undef PI_LAUNCH
    EPS_CHECK_LAUNCH("eps_pair_scores");
    return EPS_OK;
}

extern "C" int eps_pair_scores(const int64_t *rowptr, const int32_t *col, const float *val, const float *node_w,
                               int64_t n_nodes, const int32_t *u, const int32_t *v, int64_t n_pairs,
                               int32_t *count, float *cn, float *wsum, void *stream)
{
    EPS_REQUIRE(n_pairs >= 0 && n_nodes >= 0, "eps_pair_scores: negative size");
    EPS_REQUIRE(n_pairs == 0 || (rowptr && col && u && v), "eps_pair_scores: null graph or pair pointer");
    EPS_REQUIRE(!(wsum && !node_w), "eps_pair_scores: wsum requested without node_w");
    EPS_REQUIRE(count || cn || wsum || n_pairs == 0, "eps_pair_scores: no output requested");
    return launch_pair_scores<float>(rowptr, col, val, node_w, u, v, n_pairs, count, cn, wsum, (hipStream_t)stream);
}

extern "C" int eps_pair_scores_f64(const int64_t *rowptr, const int32_t *col, const float *val,
                                   const double *node_w, int64_t n_nodes, const int32_t *u, const int32_t *v,
                                   int64_t n_pairs, int32_t *count, double *wsum, void *stream)
{
    EPS_REQUIRE(n_pairs >= 0 && n_nodes >= 0, "eps_pair_scores_f64: negative size");
    EPS_REQUIRE(n_pairs == 0 || (rowptr && col && u && v), "eps_pair_scores_f64: null graph or pair pointer");
    EPS_REQUIRE(!(wsum && !node_w), "eps_pair_scores_f64: wsum requested without node_w");
    EPS_REQUIRE(count || wsum || n_pairs == 0, "eps_pair_scores_f64: no output requested");
    return launch_pair_scores<double>(rowptr, col, val, node_w, u, v, n_pairs, count, nullptr, wsum,
                                      (hipStream_t)stream);
}

// ---------------------------------------------------------------- K2: node weight table
__global__ void col_sums_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                const float *__restrict__ val, int64_t n_rows, float *__restrict__ colsum)
{
    // one wave per row, lanes stride the row: coalesced col/val reads, float atomics on colsum
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < n_rows; r += n_waves) {
        const int64_t b = rowptr[r], e = rowptr[r + 1];
        for (int64_t k = b + lane; k < e; k += 64) atomicAdd(&colsum[col[k]], val ? val[k] : 1.0f);
    }
}

template <typename WT>
__global__ void node_weights_kernel(const float *__restrict__ colsum, int64_t n, int mode, WT *__restrict__ w)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const WT s = (WT)colsum[i];
    WT m;
    if (sizeof(WT) == 4) m = (mode == EPS_W_AA) ? (WT)(1.0f / logf((float)s)) : (WT)(1.0f / (float)s);
    else m = (mode == EPS_W_AA) ? (WT)(1.0 / log((double)s)) : (WT)(1.0 / (double)s);
    if (isinf(m)) m = 0;
    w[i] = m;
}

extern "C" int eps_col_sums(const int64_t *rowptr, const int32_t *col, const float *val, int64_t n_rows,
                            int64_t n_cols, float *colsum, void *stream)
{
    EPS_REQUIRE(n_rows >= 0 && n_cols >= 0, "eps_col_sums: negative size");
    if (n_cols == 0) return EPS_OK;
    EPS_REQUIRE(colsum && (n_rows == 0 || (rowptr && col)), "eps_col_sums: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(colsum, 0, (size_t)n_cols * sizeof(float), s) != hipSuccess) {
        eps_set_error("eps_col_sums: memset failed");
        return EPS_ELAUNCH;
    }
    if (n_rows == 0) return EPS_OK;
    int64_t blocks = (n_rows + 3) / 4;
    const int64_t cap = (int64_t)eps_num_cus() * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(col_sums_kernel, dim3((unsigned)blocks), dim3(256), 0, s, rowptr, col, val, n_rows, colsum);
    EPS_CHECK_LAUNCH("eps_col_sums");
    return EPS_OK;
}

extern "C" int eps_node_weights(const float *colsum, int64_t n, int mode, float *w, void *stream)
{
    EPS_REQUIRE(n >= 0 && (mode == EPS_W_AA || mode == EPS_W_RA), "eps_node_weights: bad size or mode");
    if (n == 0) return EPS_OK;
    EPS_REQUIRE(colsum && w, "eps_node_weights: null pointer");
    hipLaunchKernelGGL(node_weights_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, colsum, n, mode, w);
    EPS_CHECK_LAUNCH("eps_node_weights");
    return EPS_OK;
}

extern "C" int eps_node_weights_f64(const float *colsum, int64_t n, int mode, double *w, void *stream)
{
    EPS_REQUIRE(n >= 0 && (mode == EPS_W_AA || mode == EPS_W_RA), "eps_node_weights_f64: bad size or mode");
    if (n == 0) return EPS_OK;
    EPS_REQUIRE(colsum && w, "eps_node_weights_f64: null pointer");
    hipLaunchKernelGGL(node_weights_kernel<double>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, colsum, n, mode, w);
    EPS_CHECK_LAUNCH("eps_node_weights_f64");
    return EPS_OK;
}

// =====================================================================================
// Column-run variant: candidate lists in the reference's order (filter.py:96-109: column-major,
// v ascending) hold long runs of pairs that share v (~24 k pairs per column on the ppa-like
// graph).  A workgroup takes a 4096-pair chunk, turns N(v) of the chunk's column into an LDS
// BITMAP over node ids once, and every pair (u, v) of the chunk then costs one coalesced read
// of row u plus one ds_read + bit test per element -- no staging of row v per pair, no log
// factor.  Node-id spaces that do not fit the LDS bitmap (N > 2^20) hash into it (w & mask) and
// every positive is verified by a binary search in row v (exactness kept; the same search
// yields the position of w in row v, which weighted graphs need for A[v,w]).  Pairs of the chunk
// whose v differs from the chunk's first v (run boundaries; rare in sorted lists) take an
// in-place global-memory search, so ANY pair list is scored correctly -- the host picks this
// kernel only when the list actually has long runs.
// =====================================================================================
#define PG_THREADS 1024
#define PG_WAVES (PG_THREADS / 64)
#ifndef PG_CHUNK
#define PG_CHUNK 16384
#endif
#define PG_QCAP 512             // per-wave hit queue entries (2 KiB)
#define PG_MAX_WORDS (1 << 15)  // 2^20 bits = 128 KiB
#ifndef PG_RING
#define PG_RING 4               // pairs whose row loads are kept in flight per wave
#endif

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

// Bitmap test of one row u against the chunk's column v with the weights gathered INLINE (a dependent
// global load per hit): used for weighted graphs, float64 weights, hashed bitmaps and oversized pairs.
template <bool HAS_VAL, bool HAS_W, typename WT, bool EXACT>
__device__ __forceinline__ void test_row_inline(const uint8_t *bm8, uint32_t bm_mask, const int32_t *__restrict__ vcol,
                                                int32_t dv, int64_t vb, bool v_has0, __amdgpu_buffer_rsrc_t rj,
                                                int64_t bju, int32_t dju, v4i cur0, v4i cur1,
                                                const float *__restrict__ val, const WT *__restrict__ node_w, int lane,
                                                int &cnt, float &r_cn, WT &r_ws)
{
    int h = 0;  // per-lane hit count over the whole row
    float acc_cn = 0.0f;
    WT acc_ws = 0;
    for (int k0 = 0; k0 < dju; k0 += 256) {
        v4i wv;
        if (k0 == 0) wv = cur0;
        else if (k0 == 256) wv = cur1;
        else wv = __builtin_amdgcn_raw_buffer_load_b128(rj, lane * 16 + k0 * 4, 0, 0);
        uint32_t b[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t w = (uint32_t)wv[e];
            const uint32_t idx = EXACT ? w : (w & bm_mask);
            b[e] = ((uint32_t)bm8[idx >> 3] >> (idx & 7)) & 1u;
            if (!EXACT) b[e] = (k0 + 4 * lane + e < dju) ? b[e] : 0u;  // 0-filled tail lanes
        }
        const uint32_t any = b[0] | b[1] | b[2] | b[3];
        if (__ballot(any != 0) != 0ull) {  // wave-uniform; taken for a minority of the units
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                bool hit = b[e] != 0;
                const uint32_t w = (uint32_t)wv[e];
                int pos = 0;
                if (!EXACT || HAS_VAL) {  // verify hashed positives / locate w inside row v
                    if (__ballot(hit) != 0ull) {
                        pos = lower_bound_uniform(vcol, dv, hit ? (int)w : 0);
                        const int pc = pos < dv ? pos : dv - 1;
                        hit = hit && pos < dv && vcol[pc] == (int)w;
                        pos = pc;
                    }
                }
                if (hit) {
                    ++h;
                    if (HAS_VAL || HAS_W) {
                        float va = 1.0f, vbv = 1.0f;
                        if (HAS_VAL) { va = val[bju + k0 + 4 * lane + e]; vbv = val[vb + pos]; }
                        if (HAS_VAL) acc_cn += va * vbv;
                        if (HAS_W) acc_ws += (WT)va * ((WT)vbv * node_w[w]);
                    }
                }
            }
        }
    }
    if (v_has0 && __builtin_amdgcn_readlane(cur0[0], 0) == 0 && lane == 0) {  // node 0: see the kernel comment
        ++h;
        if (HAS_VAL || HAS_W) {
            float va = 1.0f, vbv = 1.0f;
            if (HAS_VAL) { va = val[bju]; vbv = val[vb]; }
            if (HAS_VAL) acc_cn += va * vbv;
            if (HAS_W) acc_ws += (WT)va * ((WT)vbv * node_w[0]);
        }
    }
    const uint64_t hm = __ballot(h != 0);
    if (hm) {
        cnt = sparse_lane_sum<int>(hm, h);
        r_cn = HAS_VAL ? sparse_lane_sum<float>(hm, acc_cn) : (float)cnt;
        if (HAS_W) r_ws = sparse_lane_sum<WT>(hm, acc_ws);
    }
}

#ifdef PG_STAMP  // diagnostic build only: per-segment s_memtime sums (never in the shipped library)
__device__ unsigned long long g_stamp[16];
#define STAMP(var) unsigned long long var; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory")
#define STAMP_ADD(i, a, b) st[i] += (b) - (a)
extern "C" int eps_debug_stamps(unsigned long long *out16, int reset)
{
    hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * 16);
    if (reset) { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), z, sizeof(z)); }
    return 0;
}
#else
#define STAMP(var)
#define STAMP_ADD(i, a, b)
#endif

template <bool HAS_VAL, bool HAS_W, typename WT, bool EXACT>
__global__ __launch_bounds__(PG_THREADS) void pair_scores_grouped_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, const float *__restrict__ val,
    const WT *__restrict__ node_w, const int32_t *__restrict__ pu, const int32_t *__restrict__ pv, int64_t n_pairs,
    int32_t bm_words, uint32_t bm_mask, unsigned int *__restrict__ next_chunk, int32_t *__restrict__ out_count,
    float *__restrict__ out_cn, WT *__restrict__ out_ws)
{
    // Unit-weight float32 AA/RA (every dataset but collab): the per-hit weight gather node_w[w] is the only
    // dependent global load left on a pair's critical path, so it is DEFERRED: hits are appended to a per-wave
    // LDS queue (deterministic order: pair, then position in row u) and resolved for the whole 64-pair group at
    // once -- one parallel gather, then lane i adds up pair i's segment.  A pair costs no memory round trip.
    constexpr bool DEFER = EXACT && HAS_W && !HAS_VAL && sizeof(WT) == 4;
    extern __shared__ __attribute__((aligned(16))) uint32_t bm[];
    const uint8_t *bm8 = reinterpret_cast<const uint8_t *>(bm);
    const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
    uint32_t *q = bm + bm_words + wib * PG_QCAP;  // this wave's hit queue
    const int64_t n_chunks = (n_pairs + PG_CHUNK - 1) / PG_CHUNK;
#ifdef PG_STAMP
    unsigned long long st[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    STAMP(t_begin);
#endif

    // Chunks are handed out dynamically (one device-scope atomic per chunk): a column's cost follows the degrees
    // of its candidates, so a static split leaves the slowest workgroup running ~1.5x longer than the average one.
    // Inside a chunk the 64-pair groups are handed out dynamically too (LDS counter): group costs are heavy-tailed
    // (a few rows hold >10k entries), and a static split left the waves idle at the chunk barrier half of the time.
    __shared__ unsigned int s_chunk, s_group, s_first;
    for (;;) {
        __syncthreads();  // every wave is done with the previous chunk
        if (tid == 0) s_chunk = atomicAdd(next_chunk, 1u);
        __syncthreads();
        const int64_t chunk = s_chunk;
        if (chunk >= n_chunks) break;
        const int64_t c0 = chunk * PG_CHUNK;
        const int64_t cend = (c0 + PG_CHUNK) < n_pairs ? (c0 + PG_CHUNK) : n_pairs;
      // A chunk is cut into SEGMENTS at the column boundaries it contains (runs of equal v): each segment gets its own
      // bitmap, so a pair only leaves the bitmap path when the list is not grouped by v at all.
      for (int64_t seg = c0; seg < cend;) {
        const int32_t v0 = pv[seg];
        __syncthreads();  // every wave is done with the previous segment (bitmap, s_group, s_first)
        if (tid == 0) {
            s_group = 0u;
            s_first = (unsigned int)(cend - seg);
        }
        __syncthreads();
        {   // first index of the segment whose v differs from v0: every thread scans a short stretch
            const int len = (int)(cend - seg);
            const int stride = (len + PG_THREADS - 1) / PG_THREADS;
            for (int k = 0; k < stride; ++k) {
                const int off = tid * stride + k;
                if (off < len && pv[seg + off] != v0) {
                    atomicMin(&s_first, (unsigned int)off);
                    break;
                }
            }
        }
        __syncthreads();
        const int64_t c0s = seg;                 // this segment: pairs [c0s, c1)
        const int64_t c1 = seg + (int64_t)s_first;
        seg = c1;
        const int64_t vb = rowptr[v0];
        const int32_t dv = (int32_t)(rowptr[v0 + 1] - vb);
        const int32_t *__restrict__ vcol = col + vb;
        // EXACT mode: out-of-range lanes of a row load read as node id 0, so bit 0 is never set and membership of
        // node 0 (which, rows being sorted, can only be the FIRST element of a row) is settled on the side.
        const bool v_has0 = EXACT && dv > 0 && vcol[0] == 0;

        STAMP(tb1);
        for (int i = tid * 4; i < bm_words; i += PG_THREADS * 4) *reinterpret_cast<uint4 *>(&bm[i]) = make_uint4(0, 0, 0, 0);
        __syncthreads();
        for (int k = tid; k < dv; k += PG_THREADS) {
            const uint32_t w = (uint32_t)vcol[k];
            const uint32_t idx = EXACT ? w : (w & bm_mask);
            if (!EXACT || w != 0) atomicOr(&bm[idx >> 5], 1u << (idx & 31));
        }
        __syncthreads();
        STAMP(tb2);
        STAMP_ADD(1, tb1, tb2);  // 1: bitmap rebuild

        for (;;) {
            unsigned int gi = 0;
            if (lane == 0) gi = atomicAdd(&s_group, 1u);
            const int64_t g0 = c0s + (int64_t)__builtin_amdgcn_readfirstlane(gi) * 64;
            if (g0 >= c1) break;
            STAMP(tg0);
            const int64_t p = g0 + lane;
            const bool valid = p < c1;
            const int32_t nu = valid ? pu[p] : 0;
            const int32_t nv = valid ? pv[p] : v0;
            const int64_t ub = rowptr[nu];
            const int32_t du = valid ? (int32_t)(rowptr[nu + 1] - ub) : 0;
            const uint64_t same_mask = __ballot(nv == v0);
            int64_t vb2 = vb;
            int32_t dv2 = dv;
            if (same_mask != ~0ull) {  // a run boundary inside this group (rare)
                vb2 = rowptr[nv];
                dv2 = (int32_t)(rowptr[nv + 1] - vb2);
            }

            int32_t my_count = 0;
            float my_cn = 0.0f;
            WT my_ws = 0;
            int my_qstart = 0, my_qcnt = 0;  // this lane's pair: its segment of the hit queue
            int qlen = 0;                    // wave-uniform
            const int here = (c1 - g0) < 64 ? (int)(c1 - g0) : 64;

            // Resolve the queue: one parallel gather of node_w over all queued hits, then lane i adds up pair i's
            // segment in queue order.  ``open_j`` >= 0: pair open_j is still being tested (queue nearly full in the
            // middle of a row); its partial segment [open_q0, qlen) is folded in and the pair carries on from 0.
            auto flush = [&](int open_j, int open_q0) {
                if (!DEFER) return;
                if (open_j >= 0 && lane == open_j) { my_qstart = open_q0; my_qcnt = qlen - open_q0; }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                for (int t = lane; t < qlen; t += 64) q[t] = __builtin_bit_cast(uint32_t, (float)node_w[q[t]]);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                float sacc = 0.0f;
                for (int t = 0; t < my_qcnt; ++t) sacc += __builtin_bit_cast(float, q[my_qstart + t]);
                my_ws += (WT)sacc;
                my_qcnt = 0;
                qlen = 0;
                __builtin_amdgcn_wave_barrier();
            };

            // Every ring register is written by exactly one UNCONDITIONAL load per trip (pairs past the end of the
            // group get a zero-length descriptor: no traffic, zeros back): a conditional refill turns the ring into
            // phi copies, and hipcc then waits for the in-flight loads right after issuing them.
            v4i ring0[PG_RING], ring1[PG_RING];
#pragma unroll
            for (int r = 0; r < PG_RING; ++r) {
                const __amdgpu_buffer_rsrc_t r0 = row_rsrc(col + bcast64(ub, r), __builtin_amdgcn_readlane(du, r));
                ring0[r] = __builtin_amdgcn_raw_buffer_load_b128(r0, lane * 16, 0, 0);
                ring1[r] = __builtin_amdgcn_raw_buffer_load_b128(r0, lane * 16 + 1024, 0, 0);
            }
#ifdef PG_STAMP
            asm volatile("" ::"v"(du), "v"(ub));
            STAMP(tg1);
            STAMP_ADD(2, tg0, tg1);  // 2: group metadata
#endif
            for (int jb = 0; jb < 64; jb += PG_RING) {
#pragma unroll
                for (int r = 0; r < PG_RING; ++r) {
                    const int j = jb + r;
                    STAMP(tp0);
                    const int32_t dju = __builtin_amdgcn_readlane(du, j);   // 0 for lanes past the end of the chunk
                    const int64_t bju = bcast64(ub, j);
                    const __amdgpu_buffer_rsrc_t rj = row_rsrc(col + bju, dju);
                    const v4i cur0 = ring0[r], cur1 = ring1[r];
                    {
                        const int jn = (j + PG_RING) & 63;
                        const int32_t dn = (j + PG_RING) < 64 ? __builtin_amdgcn_readlane(du, jn) : 0;
                        const __amdgpu_buffer_rsrc_t rn = row_rsrc(col + bcast64(ub, jn), dn);
                        ring0[r] = __builtin_amdgcn_raw_buffer_load_b128(rn, lane * 16, 0, 0);
                        ring1[r] = __builtin_amdgcn_raw_buffer_load_b128(rn, lane * 16 + 1024, 0, 0);
                    }
#if defined(PG_ABLATE) && PG_ABLATE == 3   // timing-only: metadata + ring loads only
                    asm volatile("" ::"v"(cur0), "v"(cur1));
                    continue;
#endif
#ifdef PG_STAMP
                    STAMP(tp05);
                    STAMP_ADD(9, tp0, tp05);  // 9: readlanes + descriptors + refill issue
                    asm volatile("" ::"v"(cur0), "v"(cur1));
                    STAMP(tp1);
                    STAMP_ADD(3, tp05, tp1);  // 3: wait for this pair's ring data
#endif
                    if (dju == 0) continue;
                    int cnt = 0;
                    float r_cn = 0.0f;
                    WT r_ws = 0;
                    if ((same_mask >> j) & 1) {
                        if (dv == 0) continue;
                        if (DEFER) {
                            int q0 = qlen;
                            // units 0 and 1 come from the prefetch ring, later ones (rows > 512) are fetched on demand in
                            // a loop of their own: merging the two sources in one loop makes hipcc wait vmcnt(0) -- i.e.
                            // for the whole ring -- before every unit.
                            auto test_unit = [&](v4i wv) {
#if defined(PG_ABLATE) && PG_ABLATE == 1   // timing-only: no dependence on the row loads (they become dead)
                                wv = (v4i){lane * 4 + j, lane * 4 + 1 + 7 * j, lane * 4 + 2 + 13 * j, lane * 4 + 3 + j};
#endif
#if defined(PG_ABLATE) && PG_ABLATE == 2   // timing-only: row loads kept alive, no bitmap test
                                asm volatile("" ::"v"(wv));
                                return;
#endif
                                uint32_t b[4];
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const uint32_t w = (uint32_t)wv[e];
                                    b[e] = ((uint32_t)bm8[w >> 3] >> (w & 7)) & 1u;
                                }
                                const uint32_t any = b[0] | b[1] | b[2] | b[3];
                                if (__ballot(any != 0) != 0ull) {
                                    if (qlen > PG_QCAP - 256) {  // a unit adds at most 256 hits
                                        flush(j, q0);
                                        q0 = 0;
                                    }
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        const uint64_t m = __ballot(b[e] != 0);
                                        const int below = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
                                        if (b[e]) q[qlen + below] = (uint32_t)wv[e];
                                        const int c = __popcll(m);
                                        qlen += c;
                                        cnt += c;
                                    }
                                }
                            };
                            STAMP(tu0);
                            test_unit(cur0);
                            if (dju > 256) test_unit(cur1);
                            STAMP(tu1);
                            STAMP_ADD(4, tu0, tu1);  // 4: first two units
                            // rows beyond 512 entries: four loads in flight per trip (a load-use-load chain would
                            // pay one full memory latency per 256 entries; out-of-range units read as zeros = no hit)
#if defined(PG_ABLATE) && PG_ABLATE == 4   // timing-only: rows truncated to their first 512 entries
                            if (false)
#endif
                            for (int k0 = 512; k0 < dju; k0 += 1024) {
                                const v4i x0 = __builtin_amdgcn_raw_buffer_load_b128(rj, lane * 16 + k0 * 4, 0, 0);
                                const v4i x1 = __builtin_amdgcn_raw_buffer_load_b128(rj, lane * 16 + k0 * 4 + 1024, 0, 0);
                                const v4i x2 = __builtin_amdgcn_raw_buffer_load_b128(rj, lane * 16 + k0 * 4 + 2048, 0, 0);
                                const v4i x3 = __builtin_amdgcn_raw_buffer_load_b128(rj, lane * 16 + k0 * 4 + 3072, 0, 0);
                                test_unit(x0);
                                if (k0 + 256 < dju) test_unit(x1);
                                if (k0 + 512 < dju) test_unit(x2);
                                if (k0 + 768 < dju) test_unit(x3);
                            }
                            STAMP(tu2);
                            STAMP_ADD(5, tu1, tu2);  // 5: on-demand units of long rows
                            if (v_has0 && __builtin_amdgcn_readlane(cur0[0], 0) == 0) {  // node 0 is a common neighbour
                                if (qlen >= PG_QCAP) {
                                    flush(j, q0);
                                    q0 = 0;
                                }
                                if (lane == 0) q[qlen] = 0u;
                                qlen += 1;
                                cnt += 1;
                            }
                            if (cnt != 0 && lane == j) { my_count = cnt; my_cn = (float)cnt; my_qstart = q0; my_qcnt = qlen - q0; }
#ifdef PG_STAMP
                            asm volatile("" ::"v"(my_count), "v"(my_qstart), "v"(my_qcnt), "s"(qlen));
                            STAMP(tu3);
                            STAMP_ADD(8, tu2, tu3);  // 8: node-0 check + result select
#endif
                            continue;
                        }
                        test_row_inline<HAS_VAL, HAS_W, WT, EXACT>(bm8, bm_mask, vcol, dv, vb, v_has0, rj, bju, dju, cur0, cur1,
                                                                   val, node_w, lane, cnt, r_cn, r_ws);
                    } else {
                        const int32_t djv = __builtin_amdgcn_readlane(dv2, j);
                        if (djv != 0) {
                            float acc_cn = 0.0f;
                            WT acc_ws = 0;
                            score_pair_inplace<HAS_VAL, HAS_W, WT>(col, val, node_w, bju, dju, bcast64(vb2, j), djv, lane,
                                                                   cnt, acc_cn, acc_ws);
                            if (cnt != 0) {
                                r_cn = HAS_VAL ? eps_wave_sum(acc_cn) : (float)cnt;
                                if (HAS_W) r_ws = eps_wave_sum(acc_ws);
                            }
                        }
                    }
                    if (cnt != 0 && lane == j) { my_count = cnt; my_cn = r_cn; my_ws = r_ws; }
                }
            }
            STAMP(tf0);
            flush(-1, 0);
            STAMP(tf1);
            STAMP_ADD(6, tf0, tf1);  // 6: final flush of the group
            if (valid) {
                if (out_count) out_count[p] = my_count;
                if (out_cn) out_cn[p] = my_cn;
                if (HAS_W && out_ws) out_ws[p] = my_ws;
            }
        }
      }
    }
#ifdef PG_STAMP
    STAMP(t_end);
    st[7] = t_end - t_begin;  // 7: whole wave
    if (lane == 0)
        for (int i = 0; i < 10; ++i) atomicAdd(&g_stamp[i], st[i]);
#endif
}

// Work counters for the dynamic chunk hand-out: a small pool of device words inside the code object (nothing is
// allocated at run time); each launch takes the next slot and zeroes it on its own stream before the kernel.
#define PG_COUNTER_SLOTS 64
__device__ unsigned int g_chunk_counter[PG_COUNTER_SLOTS];
static std::atomic<unsigned int> g_counter_turn{0};

template <typename WT>
static int launch_pair_scores_grouped(const int64_t *rowptr, const int32_t *col, const float *val, const WT *node_w,
                                      int64_t n_nodes, const int32_t *u, const int32_t *v, int64_t n_pairs,
                                      int32_t *count, float *cn, WT *wsum, hipStream_t stream)
{
    if (n_pairs == 0) return EPS_OK;
    // bitmap: exact (one bit per node id) when it fits 2^20 bits, hashed (w & mask) + verified otherwise
    int64_t words = (n_nodes + 31) / 32;
    words = (words + PG_THREADS * 4 - 1) / (PG_THREADS * 4) * (PG_THREADS * 4);  // whole uint4 sweeps
    const bool exact = words <= PG_MAX_WORDS;
    if (!exact) words = PG_MAX_WORDS;
    const uint32_t mask = (uint32_t)(words * 32 - 1);
    // the per-wave hit queues exist only for the deferred-gather (exact bitmap) variants
    const size_t lds = (size_t)words * 4 + (exact ? (size_t)PG_WAVES * PG_QCAP * 4 : 0);
    const int64_t n_chunks = (n_pairs + PG_CHUNK - 1) / PG_CHUNK;
    int per_cu = (int)(163840 / (lds + 256));
    if (per_cu > 4) per_cu = 4;
    if (per_cu < 1) per_cu = 1;
    int64_t blocks = (int64_t)eps_num_cus() * per_cu;
    if (blocks > n_chunks) blocks = n_chunks;
    unsigned int *counter = nullptr;
    if (hipGetSymbolAddress((void **)&counter, HIP_SYMBOL(g_chunk_counter)) != hipSuccess) {
        eps_set_error("eps_pair_scores_grouped: cannot resolve the work counter");
        return EPS_ELAUNCH;
    }
    counter += g_counter_turn.fetch_add(1) % PG_COUNTER_SLOTS;
    if (hipMemsetAsync(counter, 0, sizeof(unsigned int), stream) != hipSuccess) {
        eps_set_error("eps_pair_scores_grouped: cannot reset the work counter");
        return EPS_ELAUNCH;
    }
    const bool hv = val != nullptr, hw = node_w != nullptr && wsum != nullptr;
#define PG_LAUNCH(HV, HW, EX)                                                                                        \
    do {                                                                                                             \
        auto kern = pair_scores_grouped_kernel<HV, HW, WT, EX>;                                                      \
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=         \
            hipSuccess) {                                                                                            \
            eps_set_error("eps_pair_scores_grouped: cannot reserve %zu bytes of LDS", lds);                          \
            return EPS_ELAUNCH;                                                                                      \
        }                                                                                                            \
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(PG_THREADS), lds, stream, rowptr, col, val, node_w, u, \
                           v, n_pairs, (int32_t)words, mask, counter, count, cn, wsum);                                       \
    } while (0)
    if (exact) {
        if (hv && hw) PG_LAUNCH(true, true, true);
        else if (hv) PG_LAUNCH(true, false, true);
        else if (hw) PG_LAUNCH(false, true, true);
        else PG_LAUNCH(false, false, true);
    } else {
        if (hv && hw) PG_LAUNCH(true, true, false);
        else if (hv) PG_LAUNCH(true, false, false);
        else if (hw) PG_LAUNCH(false, true, false);
        else PG_LAUNCH(false, false, false);
    }
#undef PG_LAUNCH
    EPS_CHECK_LAUNCH("eps_pair_scores_grouped");
    return EPS_OK;
}

extern "C" int eps_pair_scores_grouped(const int64_t *rowptr, const int32_t *col, const float *val,
                                       const float *node_w, int64_t n_nodes, const int32_t *u, const int32_t *v,
                                       int64_t n_pairs, int32_t *count, float *cn, float *wsum, void *stream)
{
    EPS_REQUIRE(n_pairs >= 0 && n_nodes >= 0, "eps_pair_scores_grouped: negative size");
    EPS_REQUIRE(n_pairs == 0 || (rowptr && col && u && v), "eps_pair_scores_grouped: null graph or pair pointer");
    EPS_REQUIRE(!(wsum && !node_w), "eps_pair_scores_grouped: wsum requested without node_w");
    EPS_REQUIRE(count || cn || wsum || n_pairs == 0, "eps_pair_scores_grouped: no output requested");
    return launch_pair_scores_grouped<float>(rowptr, col, val, node_w, n_nodes, u, v, n_pairs, count, cn, wsum,
                                             (hipStream_t)stream);
}

extern "C" int eps_pair_scores_grouped_f64(const int64_t *rowptr, const int32_t *col, const float *val,
                                           const double *node_w, int64_t n_nodes, const int32_t *u, const int32_t *v,
                                           int64_t n_pairs, int32_t *count, double *wsum, void *stream)
{
    EPS_REQUIRE(n_pairs >= 0 && n_nodes >= 0, "eps_pair_scores_grouped_f64: negative size");
    EPS_REQUIRE(n_pairs == 0 || (rowptr && col && u && v), "eps_pair_scores_grouped_f64: null graph or pair pointer");
    EPS_REQUIRE(!(wsum && !node_w), "eps_pair_scores_grouped_f64: wsum requested without node_w");
    EPS_REQUIRE(count || wsum || n_pairs == 0, "eps_pair_scores_grouped_f64: no output requested");
    return launch_pair_scores_grouped<double>(rowptr, col, val, node_w, n_nodes, u, v, n_pairs, count, nullptr, wsum,
                                              (hipStream_t)stream);
}
