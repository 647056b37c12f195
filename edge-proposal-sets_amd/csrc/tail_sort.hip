// The tail of the filter step -- what follows the scan of filter.py:96-142 under `--keep_top K`: the selections and the two
// orderings of filter.py:160-161 (`all_scores[:,2].sort(descending=True)` + row gather, for the K rows rank.py:294 reads) -- as
// hand-written gfx950 kernels with DEVICE-side sizes (r06).
//
// Until r05 this tail was ~110 library / runtime launches per step: three four-round radix selects (topk_keys.hip), 26 rocPRIM
// onesweep passes with their histogram / memset launches (topk_select.hip), ~20 torch element-wise kernels in between, and two
// host reads (the pre-filter's count sized the sorts: rocPRIM takes its n from the host).  Here:
//   * eps_score_hist + eps_score_pick_compact: a selection is ONE histogram over order-preserving score buckets (a minifloat of
//     the distance to a base score: 2^-8 of that distance per bucket) + one pick-and-compact launch.  The threshold is the lower
//     edge of the bucket that holds the k-th best value -- at most 0.1 % below the exact k-th, which is all a pre-filter or a
//     cut that is verified afterwards needs (the K rows themselves come out of the exact sort below);
//   * eps_radix_sort_by_u / eps_radix_sort_rows: a stable LSD radix sort (8-bit digits) of up to a few million records in ONE
//     cooperative launch -- every pass = per-workgroup digit histogram -> grid hand-over -> column scan -> hand-over -> stable
//     ranking (wave match via ballots, per-wave digit counters in LDS) + scatter -> hand-over; a pass whose digit is the same
//     for all records is skipped on the device.  The record count is read from DEVICE memory, the input transform (swap the
//     halves of a survivor key; mirror a selected pair into its two rows and map the ids back through the relabelling) and the
//     output transform (the [2, K] proposal tensor rank.py reads + scores) are folded into the first and last phase;
//   * the state block is left zeroed by the kernels that consume it (no memset launches).
#include "eps_common.h"
#include <string.h>

#define TS_T 1024               // threads per workgroup (16 waves)
#define TS_W (TS_T / 64)
#define TS_KPT 4                // records per thread and tile
#define TS_TILE (TS_T * TS_KPT)
#define TS_MAXG 256             // workgroups of a sort at most (one per CU)
#define TS_MAXPASS 16

// ---- order-preserving score buckets --------------------------------------------------------------------------------------
#define TS_MB 8                                   // mantissa bits of the bucket minifloat
#define TS_BINS ((32 - TS_MB + 1) << TS_MB)       // 6400

__device__ __forceinline__ uint32_t ts_ordered(float f)
{
    f = f + 0.0f;
    const uint32_t b = __builtin_bit_cast(uint32_t, f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ts_unordered(uint32_t o)
{
    const uint32_t b = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __builtin_bit_cast(float, b);
}
// bucket of a distance d >= 0 (in steps of the ordered bit pattern): d itself below 2^MB, else exponent | top MB mantissa bits
__device__ __forceinline__ uint32_t ts_bucket(uint32_t d)
{
    if (d < (1u << TS_MB)) return d;
    const int e = 31 - __clz((int)d);                       // >= MB
    return ((uint32_t)(e - TS_MB + 1) << TS_MB) | ((d >> (e - TS_MB)) & ((1u << TS_MB) - 1u));
}
// smallest distance that falls into bucket b
__device__ __forceinline__ uint32_t ts_bucket_floor(uint32_t b)
{
    if (b < (1u << TS_MB)) return b;
    const uint32_t e1 = b >> TS_MB, m = b & ((1u << TS_MB) - 1u);
    return ((1u << TS_MB) | m) << (e1 - 1u);
}

struct ts_sel_state {           // per (device, stream); zero at allocation, left zeroed by eps_score_pick_compact
    uint32_t hist[TS_BINS];
    uint32_t done;
    uint32_t pad;
    unsigned long long n_out;
};

struct ts_sort_state {          // per (device, stream); zero at allocation, left zeroed by the sort kernels
    uint32_t arrive;            // grid hand-over counter (monotone inside a launch)
    uint32_t leave;
    unsigned long long runs;    // runs of equal (v block, u) of the by-u order
    uint32_t totals[TS_MAXPASS][256];   // digit totals of every pass (they do not depend on the order: counted once, up front)
    uint32_t hist[TS_MAXG][256];        // a pass's digit counts per workgroup, each word tagged with the pass: (pass + 1) << 24 | count
};

extern "C" int64_t eps_tail_state_bytes(void)
{
    const size_t a = (sizeof(ts_sel_state) + 255) & ~(size_t)255, b = (sizeof(ts_sort_state) + 255) & ~(size_t)255;
    return (int64_t)(a + b);
}
static ts_sel_state *ts_sel_of(void *state) { return (ts_sel_state *)state; }
static ts_sort_state *ts_sort_of(void *state) { return (ts_sort_state *)((char *)state + ((sizeof(ts_sel_state) + 255) & ~(size_t)255)); }

// ---- eps_score_hist ------------------------------------------------------------------------------------------------------
// hist[bucket(ordered(val) - ordered(*base))] += 1 for every live entry: key >= 0 (when keys are given), val > -inf, and -- with
// `above` -- val > *above.  Values below *base count in bucket 0.
__global__ __launch_bounds__(TS_T) void ts_hist_kernel(const int64_t *__restrict__ keys, const float *__restrict__ vals, int64_t n_max,
                                                       const unsigned long long *__restrict__ n_dev, const float *__restrict__ base,
                                                       const float *__restrict__ above, uint32_t *__restrict__ hist)
{
    __shared__ uint32_t h[TS_BINS];
    const int tid = threadIdx.x, lane = tid & 63;
    int64_t n = n_max;
    if (n_dev) {
        const unsigned long long c = *n_dev;
        n = c < (unsigned long long)n_max ? (int64_t)c : n_max;
    }
    for (int i = tid; i < TS_BINS; i += TS_T) h[i] = 0u;
    __syncthreads();
    const uint32_t ob = ts_ordered(*base);
    const float floor_v = above ? *above : -__builtin_inff();
    const int64_t stride = (int64_t)gridDim.x * TS_T;
    // (whole waves run the loop: the ballots need every lane)
    for (int64_t i0 = (int64_t)blockIdx.x * TS_T + (tid & ~63); i0 < n; i0 += stride) {
        const int64_t i = i0 + lane;
        const float x = i < n ? vals[i] : -__builtin_inff();
        bool live = i < n && x > floor_v && x > -__builtin_inff();
        if (live && keys) live = keys[i] >= 0;
        const uint32_t o = ts_ordered(x);
        const uint32_t bin = live ? ts_bucket(o > ob ? o - ob : 0u) : 0u;
        const unsigned long long m = __ballot(live);
        if (m) {                                               // (a wave whose live lanes agree adds their count once)
            const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)bin, __builtin_ctzll(m));
            if (__ballot(live && bin == b0) == m) {
                if (lane == __builtin_ctzll(m)) atomicAdd(&h[b0], (uint32_t)__popcll(m));
            } else if (live) {
                atomicAdd(&h[bin], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < TS_BINS; i += TS_T)
        if (h[i]) atomicAdd(&hist[i], h[i]);
}

static unsigned ts_blocks(int64_t n_max, int per_block)
{
    int64_t b = (n_max + per_block - 1) / per_block;
    const int64_t cap = (int64_t)eps_num_cus();
    if (b > cap) b = cap;
    return (unsigned)(b < 1 ? 1 : b);
}

extern "C" int eps_score_hist(const int64_t *keys_or_null, const float *vals, int64_t n_max, const unsigned long long *n_dev_or_null,
                              const float *base, const float *above_or_null, void *state, void *stream)
{
    EPS_REQUIRE(n_max >= 0 && n_max < (1ll << 32), "eps_score_hist: bad size");
    EPS_REQUIRE(state && base && (n_max == 0 || vals), "eps_score_hist: null pointer");
    if (n_max == 0) return EPS_OK;
    hipLaunchKernelGGL(ts_hist_kernel, dim3(ts_blocks(n_max, TS_T * 16)), dim3(TS_T), 0, (hipStream_t)stream, keys_or_null, vals, n_max,
                       n_dev_or_null, base, above_or_null, ts_sel_of(state)->hist);
    EPS_CHECK_LAUNCH("eps_score_hist");
    return EPS_OK;
}

extern "C" int32_t eps_score_bins(void) { return TS_BINS; }

// The same histogram added to an array of the caller's (eps_score_bins() words, zeroed by the caller): what a rank of a sharded
// step sends to the others -- 25 KB instead of its re-scored scores (see eps_score_deal_plan).
extern "C" int eps_score_hist_into(const int64_t *keys_or_null, const float *vals, int64_t n_max, const unsigned long long *n_dev_or_null,
                                   const float *base, const float *above_or_null, uint32_t *hist, void *stream)
{
    EPS_REQUIRE(n_max >= 0 && n_max < (1ll << 32), "eps_score_hist_into: bad size");
    EPS_REQUIRE(hist && base && (n_max == 0 || vals), "eps_score_hist_into: null pointer");
    if (n_max == 0) return EPS_OK;
    hipLaunchKernelGGL(ts_hist_kernel, dim3(ts_blocks(n_max, TS_T * 16)), dim3(TS_T), 0, (hipStream_t)stream, keys_or_null, vals, n_max,
                       n_dev_or_null, base, above_or_null, hist);
    EPS_CHECK_LAUNCH("eps_score_hist_into");
    return EPS_OK;
}

// ---- eps_score_deal_plan: the job-wide cut and the final ordering's deal of a sharded step, from the ranks' histograms ------------
// hists[r * row_stride + b] = rank r's histogram (eps_score_hist_into over its re-scored scores, the same *base on every rank).
// Every rank runs this on the same gathered table and gets the same answers without another exchange:
//   *cut          = lower edge of the highest bucket with at least k values at or above it job-wide (-inf: fewer than k);
//   splitters[q]  = lower edge of the bucket where range q ends (q = 0 .. world - 2, descending): range q holds the selected scores in
//                   [splitters[q], splitters[q - 1]) -- bucket edges, so equal scores never straddle a boundary -- cut so that the
//                   ranges hold about equal numbers of selected pairs;
//   counts[r * world + q] = selected pairs of rank r in range q;  nsel[r] = selected pairs of rank r.
// One workgroup.  filter.py:160-161 (sort all rows on one host) dealt over the ranks: rank q orders range q.
#define TS_MAXWORLD 64
__global__ __launch_bounds__(TS_T) void ts_deal_plan_kernel(const uint32_t *__restrict__ hists, int64_t row_stride, int world, uint64_t k,
                                                            const float *__restrict__ base, float *__restrict__ cut,
                                                            float *__restrict__ splitters, int64_t *__restrict__ counts,
                                                            int64_t *__restrict__ nsel)
{
    __shared__ uint32_t s_suf[TS_BINS + 1];        // s_suf[b] = values in buckets >= b (job-wide); s_suf[TS_BINS] = 0
    __shared__ uint32_t s_part[TS_T];
    __shared__ uint32_t s_edge[TS_MAXWORLD + 1];   // s_edge[q] = first bucket of ranges < q, i.e. range q = [s_edge[q + 1], s_edge[q]); s_edge[0] = TS_BINS
    const int tid = threadIdx.x;
    constexpr int PER = (TS_BINS + TS_T - 1) / TS_T;
    // thread t owns the buckets TS_BINS - 1 - (t PER + j), j = 0 .. PER - 1 (from the top)
    uint32_t c[PER];
    uint32_t mine = 0u;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int b = TS_BINS - 1 - (tid * PER + j);
        uint32_t x = 0u;
        if (b >= 0)
            for (int r = 0; r < world; ++r) x += hists[(size_t)r * row_stride + b];
        c[j] = x;
        mine += x;
    }
    s_part[tid] = mine;
    __syncthreads();
    for (int d = 1; d < TS_T; d <<= 1) {
        const uint32_t add = tid >= d ? s_part[tid - d] : 0u;
        __syncthreads();
        s_part[tid] += add;
        __syncthreads();
    }
    {
        uint32_t run = s_part[tid] - mine;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int b = TS_BINS - 1 - (tid * PER + j);
            run += c[j];
            if (b >= 0) s_suf[b] = run;
        }
        if (tid == 0) s_suf[TS_BINS] = 0u;
    }
    __syncthreads();
    // b* = the highest bucket with at least k values at or above it; none: everything live is selected (bucket 0 up)
    if (tid == 0) s_edge[world] = 0u;
    __syncthreads();
    for (int b = tid; b < TS_BINS; b += TS_T)
        if (k != 0 && (uint64_t)s_suf[b] >= k && (uint64_t)s_suf[b + 1] < k) s_edge[world] = (uint32_t)b;
    __syncthreads();
    const uint32_t bstar = s_edge[world];
    const bool none = k == 0 || (uint64_t)s_suf[0] < k;
    const uint32_t lowest = none ? 0u : bstar;
    const uint64_t total = s_suf[lowest];
    if (tid == 0) s_edge[0] = TS_BINS;
    if (tid >= 1 && tid < world) s_edge[tid] = lowest;       // (ranges that no boundary is found for stay empty)
    __syncthreads();
    // boundary q (1 .. world - 1): the highest bucket e > lowest with at least q total / world values at or above it
    for (int b = tid; b < TS_BINS; b += TS_T) {
        if ((uint32_t)b <= lowest) continue;
        for (int q = 1; q < world; ++q) {
            const uint64_t target = (uint64_t)q * total / (uint64_t)world;
            if (target != 0 && (uint64_t)s_suf[b] >= target && (uint64_t)s_suf[b + 1] < target) s_edge[q] = (uint32_t)b;
        }
    }
    __syncthreads();
    if (tid == 0) {
        // (boundaries must not ascend: a target of 0 or a gap leaves the default `lowest`; make the sequence monotone from the top)
        for (int q = 1; q < world; ++q)
            if (s_edge[q] > s_edge[q - 1]) s_edge[q] = s_edge[q - 1];
        s_edge[world] = lowest;
        const uint32_t ob = ts_ordered(*base);
        auto edge_value = [&](uint32_t b) -> float {
            if (b == 0u) return -__builtin_inff();
            if (b >= (uint32_t)TS_BINS) return __builtin_inff();
            const uint64_t o = (uint64_t)ob + ts_bucket_floor(b);
            return ts_unordered(o > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)o);
        };
        *cut = none ? -__builtin_inff() : edge_value(bstar);
        for (int q = 1; q < world; ++q) splitters[q - 1] = edge_value(s_edge[q]);
    }
    __syncthreads();
    // counts[r][q] = rank r's values in the buckets [s_edge[q + 1], s_edge[q])
    for (int i = tid; i < world * world; i += TS_T) {
        const int r = i / world, q = i % world;
        unsigned long long x = 0ull;
        for (uint32_t b = s_edge[q + 1]; b < s_edge[q]; ++b) x += hists[(size_t)r * row_stride + b];
        counts[i] = (int64_t)x;
    }
    __syncthreads();
    if (tid < world) {
        unsigned long long x = 0ull;
        for (uint32_t b = lowest; b < (uint32_t)TS_BINS; ++b) x += hists[(size_t)tid * row_stride + b];
        nsel[tid] = (int64_t)x;
    }
}

extern "C" int eps_score_deal_plan(const uint32_t *hists, int64_t row_stride, int32_t world, int64_t k, const float *base, float *cut,
                                   float *splitters, int64_t *counts, int64_t *nsel, void *stream)
{
    EPS_REQUIRE(world >= 1 && world <= TS_MAXWORLD && k >= 0 && row_stride >= TS_BINS, "eps_score_deal_plan: bad argument");
    EPS_REQUIRE(hists && base && cut && counts && nsel && (world == 1 || splitters), "eps_score_deal_plan: null pointer");
    hipLaunchKernelGGL(ts_deal_plan_kernel, dim3(1), dim3(TS_T), 0, (hipStream_t)stream, hists, row_stride, (int)world, (uint64_t)k, base, cut,
                       splitters, counts, nsel);
    EPS_CHECK_LAUNCH("eps_score_deal_plan");
    return EPS_OK;
}

// ---- eps_score_pick_compact ----------------------------------------------------------------------------------------------
// Every workgroup reads the histogram, finds the highest bucket b* with at least k values at or above it (the same arithmetic
// on the same counters: no broadcast) and derives the threshold from that bucket's lower edge; then the entries at or above
// the threshold are compacted.  The last workgroup to finish publishes the count and leaves the state zeroed.
__global__ __launch_bounds__(TS_T) void ts_pick_compact_kernel(const int64_t *__restrict__ keys, const float *__restrict__ vals,
                                                               int64_t n_max, const unsigned long long *__restrict__ n_dev,
                                                               const float *__restrict__ base, const float *__restrict__ above,
                                                               uint64_t k, int mode, float pa, float pb, float pc, int swap_halves,
                                                               ts_sel_state *__restrict__ st, float *__restrict__ kth_out,
                                                               float *__restrict__ thr_out, int64_t *__restrict__ out_keys,
                                                               float *__restrict__ out_vals, int64_t out_cap, int64_t *__restrict__ n_out)
{
    __shared__ uint32_t s_part[TS_T];
    __shared__ uint32_t s_bstar;
    __shared__ uint32_t s_last;
    const int tid = threadIdx.x, lane = tid & 63;
    int64_t n = n_max;
    if (n_dev) {
        const unsigned long long c = *n_dev;
        n = c < (unsigned long long)n_max ? (int64_t)c : n_max;
    }
    // thread t owns the buckets [TS_BINS - 7 (t + 1), TS_BINS - 7 t) from the top (7 x 1024 >= 6400)
    constexpr int PER = (TS_BINS + TS_T - 1) / TS_T;
    uint32_t c[PER];
    uint32_t mine = 0u;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int b = TS_BINS - 1 - (tid * PER + j);
        c[j] = b >= 0 ? __atomic_load_n(&st->hist[b], __ATOMIC_RELAXED) : 0u;
        mine += c[j];
    }
    s_part[tid] = mine;
    if (tid == 0) s_bstar = 0xFFFFFFFFu;
    __syncthreads();
    // inclusive scan of the per-thread sums (from the top bucket down): Hillis-Steele over 1024 values in LDS
    for (int d = 1; d < TS_T; d <<= 1) {
        const uint32_t add = tid >= d ? s_part[tid - d] : 0u;
        __syncthreads();
        s_part[tid] += add;
        __syncthreads();
    }
    {
        const uint64_t incl = s_part[tid], before = incl - mine;
        if (k != 0 && before < k && incl >= k) {            // the k-th best value lies in one of this thread's buckets
            uint64_t run = before;
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                run += c[j];
                if (run >= k) {
                    s_bstar = (uint32_t)(TS_BINS - 1 - (tid * PER + j));
                    break;
                }
            }
        }
    }
    __syncthreads();
    const uint32_t bstar = s_bstar;
    const bool none = bstar == 0xFFFFFFFFu;                  // fewer than k values (or k == 0): -inf, everything live is kept
    const uint32_t ob = ts_ordered(*base);
    float kth = -__builtin_inff();
    if (!none) {
        const uint64_t o = (uint64_t)ob + ts_bucket_floor(bstar);
        // (bucket 0 also holds what lies below the base: its lower edge is "anything")
        kth = bstar == 0u ? -__builtin_inff() : ts_unordered(o > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)o);
    }
    float thr = kth;
    if (kth > -__builtin_inff()) {
        if (mode == 1) {
            const uint32_t o = ts_ordered(kth);
            thr = ts_unordered(o == 0x80000000u ? o - 2u : o - 1u);
        } else if (mode == 2) {
            const float low = kth - pa, rel = kth * pb;
            thr = (low > rel ? low : rel) - __builtin_fabsf(kth) * pc;
        }
    }
    if (blockIdx.x == 0 && tid == 0) {
        if (kth_out) *kth_out = kth;
        if (thr_out) *thr_out = thr;
    }
    const float floor_v = above ? *above : -__builtin_inff();
    if (out_keys) {
        const int64_t wave = ((int64_t)blockIdx.x * TS_T + tid) >> 6, n_waves = ((int64_t)gridDim.x * TS_T) >> 6;
        constexpr int U = 8;
        for (int64_t c0 = wave * 64 * U; c0 < n; c0 += n_waves * 64 * U) {
            int64_t kq[U];
            float sq[U];
            unsigned int bits = 0;
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int64_t i = c0 + j * 64 + lane;
                sq[j] = i < n ? vals[i] : -__builtin_inff();
            }
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int64_t i = c0 + j * 64 + lane;
                kq[j] = -1;
                if (i < n && sq[j] >= thr && sq[j] > floor_v && sq[j] > -__builtin_inff()) kq[j] = keys[i];
                if (kq[j] >= 0) bits |= 1u << j;
            }
            const int cnt = __popc(bits);
            int incl = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int t = __shfl_up(incl, d);
                if (lane >= d) incl += t;
            }
            const int total = __shfl(incl, 63);
            if (total == 0) continue;
            unsigned long long basep = 0;
            if (lane == 0) basep = atomicAdd(&st->n_out, (unsigned long long)total);
            const unsigned int blo = __shfl((unsigned int)basep, 0), bhi = __shfl((unsigned int)(basep >> 32), 0);
            unsigned long long pos = (((unsigned long long)bhi << 32) | blo) + (unsigned long long)(incl - cnt);
#pragma unroll
            for (int j = 0; j < U; ++j)
                if (bits & (1u << j)) {
                    if (pos < (unsigned long long)out_cap) {
                        const uint64_t kk = (uint64_t)kq[j];
                        out_keys[pos] = swap_halves ? (int64_t)((kk << 32) | (kk >> 32)) : (int64_t)kk;
                        if (out_vals) out_vals[pos] = sq[j];
                    }
                    ++pos;
                }
        }
    }
    // the last workgroup to finish publishes the count and leaves the state zeroed for the next selection
    __syncthreads();
    if (tid == 0) {
        __threadfence();
        s_last = atomicAdd(&st->done, 1u) + 1u == gridDim.x ? 1u : 0u;
        __threadfence();
    }
    __syncthreads();
    if (s_last) {
        if (tid == 0) {
            if (n_out) *n_out = (int64_t)__atomic_load_n(&st->n_out, __ATOMIC_RELAXED);
            st->n_out = 0ull;
            st->done = 0u;
        }
        for (int i = tid; i < TS_BINS; i += TS_T) st->hist[i] = 0u;
    }
}

extern "C" int eps_score_pick_compact(const int64_t *keys_or_null, const float *vals, int64_t n_max,
                                      const unsigned long long *n_dev_or_null, const float *base, const float *above_or_null, int64_t k,
                                      int32_t mode, float pa, float pb, float pc, int32_t swap_halves, float *kth_or_null,
                                      float *thr_or_null, int64_t *out_keys_or_null, float *out_vals_or_null, int64_t out_cap,
                                      int64_t *n_out_or_null, void *state, void *stream)
{
    EPS_REQUIRE(n_max >= 0 && n_max < (1ll << 32) && k >= 0 && mode >= 0 && mode <= 2 && out_cap >= 0, "eps_score_pick_compact: bad argument");
    EPS_REQUIRE(state && base && (n_max == 0 || vals), "eps_score_pick_compact: null pointer");
    EPS_REQUIRE(!out_keys_or_null || (keys_or_null && n_out_or_null), "eps_score_pick_compact: the compaction needs keys and n_out");
    EPS_REQUIRE(out_keys_or_null || !out_vals_or_null, "eps_score_pick_compact: out_vals without out_keys");
    // (runs for n_max == 0 as well: it is this launch that leaves the histogram clean and writes the outputs' "nothing")
    hipLaunchKernelGGL(ts_pick_compact_kernel, dim3(ts_blocks(n_max, TS_T * 16)), dim3(TS_T), 0, (hipStream_t)stream, keys_or_null, vals,
                       n_max, n_dev_or_null, base, above_or_null, (uint64_t)k, (int)mode, pa, pb, pc, (int)swap_halves, ts_sel_of(state),
                       kth_or_null, thr_or_null, out_keys_or_null, out_vals_or_null, out_cap, n_out_or_null);
    EPS_CHECK_LAUNCH("eps_score_pick_compact");
    return EPS_OK;
}

// ---- the radix sort ------------------------------------------------------------------------------------------------------
struct ts_pass {
    uint8_t src;                // 0: key, 1: val
    uint8_t shift;
    uint8_t bits;               // 1..8
    uint8_t blocked;            // 1: runs only when the by-u order is blocked (decided on the device)
};

struct ts_sort_params {
    int mode;                   // 0: survivor keys -> by u; 1: selected pairs -> rows
    const int64_t *in_keys;
    const float *in_vals;       // rows: scores of the pairs
    const int64_t *n_dev;       // device count of the input (by u: keys; rows: pairs) or NULL
    int64_t n_max;
    int64_t k;                  // rows: proposals wanted
    const int64_t *perm;        // rows: id i of the scanned graph is the caller's perm[i] (or NULL)
    int id_bits;
    int v_block_shift;          // by u: 0 or the block shift of the conditional passes
    uint64_t *buf_k[2];
    uint32_t *buf_v[2];         // NULL for key-only sorts
    int n_pass;
    ts_pass pass[TS_MAXPASS];
    int64_t *out_keys;          // by u: n keys (u << 32) | v
    int64_t *out_pairs;         // rows: [2][out_ld] (u row, v row)
    float *out_scores;          // rows
    int64_t out_ld;
    int64_t *n_rows_out;        // rows: min(k, 2 m) (device)
    ts_sort_state *st;
};

__device__ __forceinline__ void ts_grid_sync(uint32_t *arrive, uint32_t target)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        atomicAdd(arrive, 1u);
        while (__atomic_load_n(arrive, __ATOMIC_RELAXED) < target) __builtin_amdgcn_s_sleep(1);
        __threadfence();
    }
    __syncthreads();
}

// lanes of the wave whose (live) digit equals this lane's
__device__ __forceinline__ unsigned long long ts_match8(uint32_t digit, bool live)
{
    unsigned long long peers = __ballot(live);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const bool bit = (digit >> b) & 1u;
        const unsigned long long m = __ballot(live && bit);
        peers &= bit ? m : ~m;
    }
    return peers;
}

__device__ __forceinline__ uint32_t ts_digit(const ts_pass &ps, uint64_t key, uint32_t val)
{
    const uint32_t mask = (1u << ps.bits) - 1u;
    return ps.src ? (val >> ps.shift) & mask : (uint32_t)(key >> ps.shift) & mask;
}

__global__ __launch_bounds__(TS_T) void ts_sort_kernel(ts_sort_params p)
{
    __shared__ uint32_t s_cnt[TS_W][256];       // per-wave digit counters of a tile; then the records' wave bases
    __shared__ uint32_t s_off[256];             // running output position per digit of this workgroup
    __shared__ uint32_t s_hist[256];
    __shared__ uint32_t s_scan[TS_T];
    __shared__ uint32_t s_tot[TS_MAXPASS][256];
    __shared__ uint32_t s_last;
    const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
    const unsigned G = gridDim.x, b = blockIdx.x;
    ts_sort_state *st = p.st;
    uint32_t epoch = 0;                          // grid hand-overs so far
    const bool rows = p.mode == 1;
    const bool has_val = p.buf_v[0] != nullptr;

    int64_t n_in = p.n_max;
    if (p.n_dev) {
        const int64_t c = *p.n_dev;
        n_in = c < 0 ? 0 : (c < p.n_max ? c : p.n_max);
    }
    const int64_t n = rows ? 2 * n_in : n_in;   // records
    int64_t per = (n + G - 1) / G;
    per = (per + TS_TILE - 1) / TS_TILE * TS_TILE;
    const int64_t lo = (int64_t)b * per < n ? (int64_t)b * per : n;
    const int64_t hi = lo + per < n ? lo + per : n;

    // ---- phase 0: the input transform fills buffer 0, and every pass's digit totals are counted (they do not depend on the
    // order of the records: once, here, instead of a grid-wide exchange per pass) ---------------------------------------------
    for (int i = tid; i < p.n_pass * 256; i += TS_T) (&s_tot[0][0])[i] = 0u;
    __syncthreads();
    for (int64_t i0 = lo + (tid & ~63); i0 < hi; i0 += TS_T) {
        const int64_t i = i0 + lane;
        const bool live = i < hi;
        uint64_t key = 0ull;
        uint32_t val = 0u;
        if (live) {
            if (!rows) {
                const uint64_t k = (uint64_t)p.in_keys[i];                // v << 32 | u  ->  u << 32 | v
                key = (k << 32) | (k >> 32);
                p.buf_k[0][i] = key;
            } else {
                const int64_t j = i < n_in ? i : i - n_in;
                const uint64_t k = (uint64_t)p.in_keys[j];
                uint64_t a = k & 0xFFFFFFFFull, c = k >> 32;
                if (p.perm) {      // ids of a relabelled graph back to the caller's; the larger one is "v" again
                    a = (uint64_t)p.perm[a];
                    c = (uint64_t)p.perm[c];
                }
                const uint64_t small = a < c ? a : c, large = a < c ? c : a;
                // row of proposal (u, v) sorts by (v, u): the direct row has v = large, the mirrored one v = small
                const uint64_t vv = i < n_in ? large : small, uu = i < n_in ? small : large;
                key = (vv << p.id_bits) | uu;
                val = ~ts_ordered(p.in_vals[j]);                          // ascending in this = descending in the score
                p.buf_k[0][i] = key;
                p.buf_v[0][i] = val;
            }
        }
        for (int ip = 0; ip < p.n_pass; ++ip) {
            const uint32_t d = ts_digit(p.pass[ip], key, val);
            const unsigned long long peers = ts_match8(d, live);
            if (live && (peers & ((1ull << lane) - 1ull)) == 0ull) atomicAdd(&s_tot[ip][d], (uint32_t)__popcll(peers));
        }
    }
    __syncthreads();
    for (int i = tid; i < p.n_pass * 256; i += TS_T) {
        const uint32_t c = (&s_tot[0][0])[i];
        if (c) atomicAdd(&(&st->totals[0][0])[i], c);
    }
    int cur = 0;
    bool blocked = false;
    bool decided = false;
    epoch += 1;
    ts_grid_sync(&st->arrive, epoch * G);
    for (int i = tid; i < p.n_pass * 256; i += TS_T) (&s_tot[0][0])[i] = __atomic_load_n(&(&st->totals[0][0])[i], __ATOMIC_RELAXED);
    __syncthreads();

    for (int ip = 0; ip < p.n_pass; ++ip) {
        const ts_pass ps = p.pass[ip];
        if (ps.blocked && !decided) {
            // ---- by u: are the runs of equal (v block, u) long enough for the blocked order? (counted on the (u, v) order) ----
            unsigned int c = 0u;
            const uint64_t *src = p.buf_k[cur];
            for (int64_t i = lo + tid; i < hi; i += TS_T) {
                const uint64_t k = src[i], kp = i ? src[i - 1] : ~k;
                c += (k >> 32) != (kp >> 32) || ((uint32_t)k >> p.v_block_shift) != ((uint32_t)kp >> p.v_block_shift) ? 1u : 0u;
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
            if (lane == 0 && c) atomicAdd(&st->runs, (unsigned long long)c);
            epoch += 1;
            ts_grid_sync(&st->arrive, epoch * G);
            blocked = __atomic_load_n(&st->runs, __ATOMIC_RELAXED) * 64ull <= (unsigned long long)n;
            decided = true;
        }
        if (ps.blocked && !blocked) continue;                  // (uniform over the grid)
        // a pass whose digit is the same for every record is the identity: skipped (the same totals for everyone: uniform)
        {
            if (tid == 0) s_last = 0u;
            __syncthreads();
            if (tid < 256 && n > 0 && (int64_t)s_tot[ip][tid] == n) s_last = 1u;
            __syncthreads();
            const bool trivial = s_last != 0u;
            __syncthreads();
            if (trivial) continue;
        }
        const uint64_t *src_k = p.buf_k[cur];
        const uint32_t *src_v = has_val ? p.buf_v[cur] : nullptr;
        // ---- A: digit histogram of this workgroup's chunk, published with the pass's tag ---------------------------------------
        if (tid < 256) s_hist[tid] = 0u;
        __syncthreads();
#ifndef TS_ABL_NOHIST
        for (int64_t i0 = lo + (tid & ~63); i0 < hi; i0 += TS_T) {
            const int64_t i = i0 + lane;
            const bool live = i < hi;
            const uint64_t key = live ? src_k[i] : 0ull;
            const uint32_t val = live && ps.src ? src_v[i] : 0u;
            const uint32_t d = ts_digit(ps, key, val);
            const unsigned long long peers = ts_match8(d, live);
            if (live && (peers & ((1ull << lane) - 1ull)) == 0ull) atomicAdd(&s_hist[d], (uint32_t)__popcll(peers));
        }
#endif
        __syncthreads();
        const uint32_t tag = (uint32_t)(ip + 1) << 24;
        if (tid < 256) __atomic_store_n(&st->hist[b][tid], tag | s_hist[tid], __ATOMIC_RELAXED);
        // ---- B: this workgroup's first output position per digit = digits below (totals) + the same digit in the workgroups
        // before this one (their published counts: four threads per digit, each a quarter of the predecessors, spinning on the tag)
        {
            const int d = tid & 255, q = tid >> 8;
            uint32_t sum = 0u;
            for (unsigned bb = (unsigned)q; bb < b; bb += 4u) {
                uint32_t w;
                do {
                    w = __atomic_load_n(&st->hist[bb][d], __ATOMIC_RELAXED);
                } while ((w & 0xFF000000u) != tag);
                sum += w & 0x00FFFFFFu;
            }
            s_scan[tid] = sum;
            __syncthreads();
            const uint32_t x = tid < 256 ? s_tot[ip][tid] : 0u;
            const uint32_t before = tid < 256 ? s_scan[tid] + s_scan[tid + 256] + s_scan[tid + 512] + s_scan[tid + 768] : 0u;
            __syncthreads();
            s_scan[tid] = x;
            __syncthreads();
            for (int s = 1; s < 256; s <<= 1) {
                const uint32_t add = tid >= s && tid < 256 ? s_scan[tid - s] : 0u;
                __syncthreads();
                if (tid < 256) s_scan[tid] += add;
                __syncthreads();
            }
            if (tid < 256) s_off[tid] = s_scan[tid] - x + before;
            __syncthreads();
        }
        // ---- C: stable ranking + scatter, tile by tile -----------------------------------------------------------------------
        uint64_t *dst_k = p.buf_k[cur ^ 1];
        uint32_t *dst_v = has_val ? p.buf_v[cur ^ 1] : nullptr;
#ifndef TS_ABL_NOSCATTER
        for (int64_t t0 = lo; t0 < hi; t0 += TS_TILE) {
            for (int i = lane; i < 256; i += 64) s_cnt[wib][i] = 0u;
            uint64_t key[TS_KPT];
            uint32_t val[TS_KPT], dig[TS_KPT], rank[TS_KPT];
            bool live[TS_KPT];
#pragma unroll
            for (int j = 0; j < TS_KPT; ++j) {
                const int64_t i = t0 + (int64_t)wib * (64 * TS_KPT) + j * 64 + lane;
                live[j] = i < hi;
                key[j] = live[j] ? src_k[i] : 0ull;
                val[j] = live[j] && has_val ? src_v[i] : 0u;
                dig[j] = ts_digit(ps, key[j], val[j]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < TS_KPT; ++j) {
                const unsigned long long peers = ts_match8(dig[j], live[j]);
                const uint32_t lower = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
                const uint32_t old = live[j] ? s_cnt[wib][dig[j]] : 0u;
                rank[j] = old + lower;
                __builtin_amdgcn_wave_barrier();
                if (live[j] && lower == 0u) s_cnt[wib][dig[j]] = old + (uint32_t)__popcll(peers);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            __syncthreads();
            if (tid < 256) {                                       // bases of the waves inside the tile, on top of the running position
                uint32_t run = s_off[tid];
#pragma unroll
                for (int w = 0; w < TS_W; ++w) {
                    const uint32_t c = s_cnt[w][tid];
                    s_cnt[w][tid] = run;
                    run += c;
                }
                s_off[tid] = run;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < TS_KPT; ++j)
                if (live[j]) {
                    const uint32_t pos = s_cnt[wib][dig[j]] + rank[j];
                    dst_k[pos] = key[j];
                    if (has_val) dst_v[pos] = val[j];
                }
            __syncthreads();
        }
#endif
        cur ^= 1;
        epoch += 1;
        ts_grid_sync(&st->arrive, epoch * G);
    }

    // ---- the output transform -------------------------------------------------------------------------------------------------
    {
        const uint64_t *src_k = p.buf_k[cur];
        if (!rows) {
            for (int64_t i = lo + tid; i < hi; i += TS_T) p.out_keys[i] = (int64_t)src_k[i];
        } else {
            const int64_t take = p.k < n ? p.k : n;
            const uint32_t *src_v = p.buf_v[cur];
            const uint64_t idm = (1ull << p.id_bits) - 1ull;
            for (int64_t i = lo + tid; i < hi && i < take; i += TS_T) {
                const uint64_t k = src_k[i];
                p.out_pairs[i] = (int64_t)(k & idm);                            // u
                p.out_pairs[p.out_ld + i] = (int64_t)(k >> p.id_bits);          // v
                p.out_scores[i] = ts_unordered(~src_v[i]);
            }
            if (b == 0 && tid == 0 && p.n_rows_out) *p.n_rows_out = take;
        }
    }
    // the last workgroup to leave clears the state (nobody waits on `arrive` or reads a count any more once everyone is past the
    // last hand-over)
    __syncthreads();
    if (tid == 0) {
        __threadfence();
        s_last = atomicAdd(&st->leave, 1u) + 1u == G ? 1u : 0u;
    }
    __syncthreads();
    if (s_last) {
        for (int i = tid; i < TS_MAXPASS * 256; i += TS_T) (&st->totals[0][0])[i] = 0u;
        for (int i = tid; i < (int)G * 256; i += TS_T) (&st->hist[0][0])[i] = 0u;
        __syncthreads();
        if (tid == 0) {
            st->arrive = 0u;
            st->runs = 0ull;
            __threadfence();
            st->leave = 0u;
        }
    }
}

static int ts_launch_sort(ts_sort_params &p, int64_t n_records_max, hipStream_t s, const char *who)
{
    // (16 K records per workgroup at least: a hand-over costs the more the more workgroups take part)
    int64_t g = (n_records_max + 4 * TS_TILE - 1) / (4 * TS_TILE);
    const int64_t cap = eps_num_cus() < TS_MAXG ? eps_num_cus() : TS_MAXG;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    void *args[] = {(void *)&p};
    hipError_t err = hipLaunchCooperativeKernel((const void *)ts_sort_kernel, dim3((unsigned)g), dim3(TS_T), args, 0, s);
    if (err != hipSuccess && g > 1) {
        (void)hipGetLastError();
        err = hipLaunchCooperativeKernel((const void *)ts_sort_kernel, dim3(1), dim3(TS_T), args, 0, s);
    }
    if (err != hipSuccess) {
        (void)hipGetLastError();
        eps_set_error("%s: cooperative launch failed: %s", who, hipGetErrorString(err));
        return EPS_ELAUNCH;
    }
    return EPS_OK;
}

static int ts_add_passes(ts_sort_params &p, int src, int from_bit, int to_bit, int blocked)
{
    for (int s = from_bit; s < to_bit; s += 8) {
        if (p.n_pass >= TS_MAXPASS) return -1;
        ts_pass q;
        q.src = (uint8_t)src;
        q.shift = (uint8_t)s;
        q.bits = (uint8_t)(to_bit - s < 8 ? to_bit - s : 8);
        q.blocked = (uint8_t)blocked;
        p.pass[p.n_pass++] = q;
    }
    return 0;
}

static size_t ts_align(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" int64_t eps_radix_sort_workspace_bytes(int64_t n_records)
{
    if (n_records <= 0) return 256;
    return (int64_t)(2 * ts_align((size_t)n_records * 8) + 2 * ts_align((size_t)n_records * 4));
}

// Survivor keys (v << 32 | u, u < v, any order; *n_dev_or_null of them, at most n_max) -> out_by_u = (u << 32) | v sorted by
// (u, v), or -- v_block_shift > 0 and runs of equal (v >> shift, u) that average 64 pairs -- by (v >> shift, u, v): what
// eps_rescore_runs wants (see eps_sort_pairs_by_u, whose order this reproduces).
extern "C" int eps_radix_sort_by_u(const int64_t *keys, int64_t n_max, const int64_t *n_dev_or_null, int32_t id_bits,
                                   int32_t v_block_shift, int64_t *out_by_u, void *workspace, int64_t workspace_bytes, void *state,
                                   void *stream)
{
    EPS_REQUIRE(n_max >= 0 && n_max < (1ll << 31) && id_bits >= 1 && id_bits <= 32 && v_block_shift >= 0 && v_block_shift <= 32,
                "eps_radix_sort_by_u: bad argument");
    if (n_max == 0) return EPS_OK;
    EPS_REQUIRE(keys && out_by_u && state, "eps_radix_sort_by_u: null pointer");
    EPS_REQUIRE(workspace && ((uintptr_t)workspace & 255) == 0 && workspace_bytes >= eps_radix_sort_workspace_bytes(n_max),
                "eps_radix_sort_by_u: needs a 256-byte aligned workspace of eps_radix_sort_workspace_bytes(n_max) bytes");
    ts_sort_params p;
    memset(&p, 0, sizeof(p));
    p.mode = 0;
    p.in_keys = keys;
    p.n_dev = n_dev_or_null;
    p.n_max = n_max;
    p.id_bits = id_bits;
    p.v_block_shift = v_block_shift > 0 && v_block_shift < id_bits ? v_block_shift : 0;
    char *w = (char *)workspace;
    p.buf_k[0] = (uint64_t *)w;
    p.buf_k[1] = (uint64_t *)(w + ts_align((size_t)n_max * 8));
    p.out_keys = out_by_u;
    p.st = ts_sort_of(state);
    int rc = ts_add_passes(p, 0, 0, id_bits, 0);                                    // v ...
    if (!rc) rc = ts_add_passes(p, 0, 32, 32 + id_bits, 0);                         // ... then u (LSD)
    if (!rc && p.v_block_shift) rc = ts_add_passes(p, 0, p.v_block_shift, id_bits, 1);      // blocks of v first, if the runs stay long
    EPS_REQUIRE(!rc, "eps_radix_sort_by_u: too many passes");
    return ts_launch_sort(p, n_max, (hipStream_t)stream, "eps_radix_sort_by_u");
}

// The selected unordered pairs (sel_keys v << 32 | u in the scanned graph's labels, sel_vals their exact scores; *m_dev_or_null of
// them, at most m_max) -> the first min(k, 2 m) DIRECTED rows of the declared order (score descending, then (v, u) ascending in
// the caller's labels: perm maps the ids back, NULL = as they are): out_pairs[0][i] = u, out_pairs[1][i] = v (row stride out_ld),
// out_scores[i]; *n_rows_out (device) = the number of rows.  filter.py:160-165 for the rows rank.py:294 reads.
extern "C" int eps_radix_sort_rows(const int64_t *sel_keys, const float *sel_vals, int64_t m_max, const int64_t *m_dev_or_null, int64_t k,
                                   int32_t id_bits, const int64_t *perm_or_null, int64_t *out_pairs, int64_t out_ld, float *out_scores,
                                   int64_t *n_rows_out_or_null, void *workspace, int64_t workspace_bytes, void *state, void *stream)
{
    EPS_REQUIRE(m_max >= 0 && m_max < (1ll << 30) && k >= 0 && id_bits >= 1 && id_bits <= 32, "eps_radix_sort_rows: bad argument");
    if (m_max == 0 || k == 0) {
        if (n_rows_out_or_null && hipMemsetAsync(n_rows_out_or_null, 0, sizeof(int64_t), (hipStream_t)stream) != hipSuccess) {
            eps_set_error("eps_radix_sort_rows: cannot clear the row count");
            return EPS_ELAUNCH;
        }
        return EPS_OK;
    }
    EPS_REQUIRE(sel_keys && sel_vals && out_pairs && out_scores && state, "eps_radix_sort_rows: null pointer");
    EPS_REQUIRE(out_ld >= (k < 2 * m_max ? k : 2 * m_max), "eps_radix_sort_rows: out_ld is shorter than the rows");
    EPS_REQUIRE(workspace && ((uintptr_t)workspace & 255) == 0 && workspace_bytes >= eps_radix_sort_workspace_bytes(2 * m_max),
                "eps_radix_sort_rows: needs a 256-byte aligned workspace of eps_radix_sort_workspace_bytes(2 m_max) bytes");
    ts_sort_params p;
    memset(&p, 0, sizeof(p));
    p.mode = 1;
    p.in_keys = sel_keys;
    p.in_vals = sel_vals;
    p.n_dev = m_dev_or_null;
    p.n_max = m_max;
    p.k = k;
    p.perm = perm_or_null;
    p.id_bits = id_bits;
    const size_t rows = 2 * (size_t)m_max;
    char *w = (char *)workspace;
    p.buf_k[0] = (uint64_t *)w;                    w += ts_align(rows * 8);
    p.buf_k[1] = (uint64_t *)w;                    w += ts_align(rows * 8);
    p.buf_v[0] = (uint32_t *)w;                    w += ts_align(rows * 4);
    p.buf_v[1] = (uint32_t *)w;
    p.out_pairs = out_pairs;
    p.out_ld = out_ld;
    p.out_scores = out_scores;
    p.n_rows_out = n_rows_out_or_null;
    p.st = ts_sort_of(state);
    int rc = ts_add_passes(p, 0, 0, 2 * id_bits, 0);            // (v, u) packed: least significant criterion first ...
    if (!rc) rc = ts_add_passes(p, 1, 0, 32, 0);                // ... then the score, descending
    EPS_REQUIRE(!rc, "eps_radix_sort_rows: too many passes");
    return ts_launch_sort(p, 2 * m_max, (hipStream_t)stream, "eps_radix_sort_rows");
}

// (one empty kernel per translation unit: launching it makes the HIP runtime load this unit's code object -- eps_warm_up)
__global__ void tail_sort_warm_kernel() {}
extern "C" void eps_warm_tail_sort(void *stream) { hipLaunchKernelGGL(tail_sort_warm_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream); }
