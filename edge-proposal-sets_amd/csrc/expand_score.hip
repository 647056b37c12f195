// Fused candidate generation + scoring of the filter stage, gfx950.
//
// Replaces filter.py:96-109 -- `A2 = adj_t @ adj_t` (host SpGEMM), remove diagonal, zero the
// known edges, take the nonzeros in column-major order -- AND the scoring pass that follows it
// for the heuristic filters (adamic_utils.py:13-25, train_and_eval.py:195-216, models.py:536-542).
// The reference throws away the value of A @ A, which already is CN(u,v), and then re-derives it
// pair by pair; here one expansion of the 2-hop paths v - w - u of a column v yields, for every
// candidate u at once, the common-neighbour count and sum_w A[u,w] * (A[v,w] * node_w[w]):
// work proportional to the number of PATHS (~1.3 per candidate on the ppa-like graph) instead of
// the sum of row lengths (~500 per candidate) an intersection per pair costs.
//
// One 1024-thread workgroup per column v (columns handed out dynamically):
//   A. mark: for every w in N(v) (one wave each) and every u in N(w): set bit u of an LDS bitmap
//      over the node ids; then clear the bits of N(v) and of v itself (known edges, diagonal).
//   B. rank: per-word popcounts -> block-wide exclusive scan -> prefix[] in LDS; the total is the
//      column's candidate count (kernel 1 stops here: counts -> host cumsum -> colptr).
//   C. emit: set bits in ascending order -> cand_u[colptr[v] + rank] (ascending u: the
//      reference's column-major order for free).
//      Columns may be handed out in a caller-given order (heaviest first keeps the tail of a
//      launch short: a hub column is one workgroup's work for milliseconds).
//   D. score: walk the same paths again; a path whose u is a candidate contributes its term to
//      the candidate's slot.  Scattered global atomics are line read-modify-writes behind the
//      L2 (~5x the cost of a plain store, which the L2 absorbs) and bounded this pass, so they
//      are kept for the few paths that need them: two more LDS bitmaps over the column's
//      candidate RANKS tell the first and the second path to reach a candidate (the value
//      ds_or_rtn returns); the first stores its term into score[], the second into a scratch
//      array, and only a third or later one (10 % of the paths on the ppa-like graph; 84 % of
//      the candidates have one path, 11 % two) adds to a 64-bit accumulator.  The three kinds of
//      slot live in separate arrays: a plain-stored dirty line and a memory-side atomic on the
//      same line evict each other.  eps_expand_finish sums the three in 2^-40 FIXED POINT:
//      integer addition is associative, so the result is bit-reproducible whatever the arrival
//      order (float atomics are not), and exact up to the final rounding to float32.
// Requires a SYMMETRIC adjacency (filter.py's always is: rank.py:33 to_symmetric) and
//      The later paths of the hottest candidates (a pair of hubs has thousands of common
//      neighbours; atomics on one address serialise) are summed in a small LDS table first.
// N <= 786,432 node ids (bitmap + rank tables: 5.5 bytes per 32 ids of the 160 KiB LDS; what
// is left holds the hot table and the two arrival bitmaps, at reduced resolution for columns
// with more candidates than bits); the host falls back to the tensor-op expansion above that.
#include "eps_common.h"

#include <stdlib.h>

#define EX_THREADS 1024
#define EX_WAVES (EX_THREADS / 64)
#define EX_FIXED_SHIFT 40
#define EX_FILL_LDS 157696  // dynamic LDS of the fill kernel: 154 KiB (the rest of the 160 KiB is static LDS)
#define EX_LONGQ 1024       // rows longer than one unit queued per column and pass (static LDS)
#define EX_MAX_WPT 24       // bitmap words per thread: 24 * 1024 words * 5.5 B = 132 KiB; hot table 16 KiB; the rest: arrival bitmaps
#define EX_HOT 1024         // entries of the per-column LDS table that absorbs the later paths of the hottest candidates
#define EX_HOT_EMPTY 0xFFFFFFFFu


__device__ __forceinline__ int wave_incl_scan(int x, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(x, o);
        if (lane >= o) x += y;
    }
    return x;
}

typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ long long ex_to_fixed(float x)
{
    return __double2ll_rn((double)x * (double)(1ll << EX_FIXED_SHIFT));
}

__device__ __forceinline__ int64_t ex_bcast64(int64_t x, int j)
{
    const int lo = __builtin_amdgcn_readlane((int)(x & 0xffffffffll), j);
    const int hi = __builtin_amdgcn_readlane((int)(x >> 32), j);
    return ((int64_t)hi << 32) | (uint32_t)lo;
}

// Walk the 2-hop paths v - w - u of one column, all 16 waves of the workgroup together.
// A "unit" is 256 consecutive entries of one row (one 16-byte load per lane) and costs about the same whatever it
// holds, so the work is balanced in units: wave i takes the FIRST unit of rows i, i+16, ... -- the row descriptors
// (w, rowptr[w], rowptr[w+1]) of up to 64 of its rows fetched lane-parallel (one latency for the batch instead of a
// dependent chain per row), the next row's unit in flight while the current one is consumed (raw buffer loads:
// out-of-range lanes read 0, no bounds branch) -- and the rows longer than one unit are queued in LDS; after a barrier
// their remaining units are dealt round-robin over the waves.  (Row lengths are heavy-tailed: with whole rows per wave
// the waves of a workgroup waited at the closing barrier for a third of the pass.)
// body(k, wb, base, u4, nvalid): entries [base, base+nvalid) of row w = vcol[k] (nvalid in 0..4 per lane).
// Ends with a workgroup barrier; the next call must be separated from this one by another barrier.
template <typename Body>
__device__ __forceinline__ void for_each_path(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                              const int32_t *__restrict__ vcol, int32_t dv, int wib, int lane,
                                              int *s_long, int *s_nlong, Body body)
{
    for (int b0 = wib; b0 < dv; b0 += EX_WAVES * 64) {
        const int k_mine = b0 + EX_WAVES * lane;
        const bool ok = k_mine < dv;
        const int32_t w_mine = ok ? vcol[k_mine] : 0;
        const int64_t wb_mine = rowptr[w_mine];
        const int32_t dw_mine = ok ? (int32_t)(rowptr[w_mine + 1] - wb_mine) : 0;
        const int left = (dv - b0 + EX_WAVES - 1) / EX_WAVES;
        const int nrows = left < 64 ? left : 64;
        // ring of two rows: every ring register is written by one unconditional load per trip
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(col + ex_bcast64(wb_mine, 0)), 0,
                                                                       __builtin_amdgcn_readlane(dw_mine, 0) * 4, 0x00020000);
        v4i nxt = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, 0, 0);
        for (int j = 0; j < nrows; ++j) {
            const int32_t dw = __builtin_amdgcn_readlane(dw_mine, j);
            const int64_t wb = ex_bcast64(wb_mine, j);
            const __amdgpu_buffer_rsrc_t rj = rs;
            const v4i cur = nxt;
            {
                const int jn = (j + 1) & 63;
                const int32_t dn = j + 1 < nrows ? __builtin_amdgcn_readlane(dw_mine, jn) : 0;
                rs = __builtin_amdgcn_make_buffer_rsrc((void *)(col + ex_bcast64(wb_mine, jn)), 0, dn * 4, 0x00020000);
                nxt = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, 0, 0);
            }
            const int k = b0 + EX_WAVES * j;
            int nv = dw - 4 * lane;
            body(k, wb, 4 * lane, cur, nv < 0 ? 0 : (nv > 4 ? 4 : nv));
            if (dw > 256) {
                int q = 0;
                if (lane == 0) q = atomicAdd(s_nlong, 1);
                q = __builtin_amdgcn_readfirstlane(q);
                if (q < EX_LONGQ) {
                    if (lane == 0) s_long[q] = k;
                } else {  // queue full (a hub column with thousands of long rows): finish this row here
                    for (int e0 = 256; e0 < dw; e0 += 512) {
                        const v4i x0 = __builtin_amdgcn_raw_buffer_load_b128(rj, lane * 16 + e0 * 4, 0, 0);
                        const v4i x1 = __builtin_amdgcn_raw_buffer_load_b128(rj, lane * 16 + e0 * 4 + 1024, 0, 0);
                        int n0 = dw - e0 - 4 * lane, n1 = n0 - 256;
                        body(k, wb, e0 + 4 * lane, x0, n0 < 0 ? 0 : (n0 > 4 ? 4 : n0));
                        if (e0 + 256 < dw) body(k, wb, e0 + 256 + 4 * lane, x1, n1 < 0 ? 0 : (n1 > 4 ? 4 : n1));
                    }
                }
            }
        }
    }
    __syncthreads();
    // the units past the first one of the queued rows: unit c of queue entry i belongs to wave (c + i) & 15
    const int nl = *s_nlong < EX_LONGQ ? *s_nlong : EX_LONGQ;
    for (int q0 = 0; q0 < nl; q0 += 64) {
        const bool ok = q0 + lane < nl;
        const int k_mine = ok ? s_long[q0 + lane] : 0;
        const int32_t w_mine = vcol[k_mine];
        const int64_t wb_mine = rowptr[w_mine];
        const int32_t dw_mine = ok ? (int32_t)(rowptr[w_mine + 1] - wb_mine) : 0;
        const int n = nl - q0 < 64 ? nl - q0 : 64;
        for (int j = 0; j < n; ++j) {
            const int32_t dw = __builtin_amdgcn_readlane(dw_mine, j);
            const int64_t wb = ex_bcast64(wb_mine, j);
            const int k = __builtin_amdgcn_readlane(k_mine, j);
            int c0 = (wib - (q0 + j)) & (EX_WAVES - 1);
            if (c0 == 0) c0 = EX_WAVES;
            if (c0 * 256 >= dw) continue;
            const __amdgpu_buffer_rsrc_t rj = __builtin_amdgcn_make_buffer_rsrc((void *)(col + wb), 0, dw * 4, 0x00020000);
            for (int e0 = c0 * 256; e0 < dw; e0 += 2 * EX_WAVES * 256) {  // two of this wave's units in flight per trip
                const int e1 = e0 + EX_WAVES * 256;
                const v4i x0 = __builtin_amdgcn_raw_buffer_load_b128(rj, lane * 16 + e0 * 4, 0, 0);
                const v4i x1 = __builtin_amdgcn_raw_buffer_load_b128(rj, lane * 16 + e1 * 4, 0, 0);
                int n0 = dw - e0 - 4 * lane, n1 = dw - e1 - 4 * lane;
                body(k, wb, e0 + 4 * lane, x0, n0 < 0 ? 0 : (n0 > 4 ? 4 : n0));
                if (e1 < dw) body(k, wb, e1 + 4 * lane, x1, n1 < 0 ? 0 : (n1 > 4 ? 4 : n1));
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) *s_nlong = 0;
}

#ifdef EX_STAMP  // diagnostic build only: per-phase s_memtime sums of wave 0 of every workgroup (never in the shipped library)
__device__ unsigned long long g_ex_stamp[16];
#define XSTAMP(var) unsigned long long var; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory")
#define XSTAMP_ADD(i, a, b) xst[i] += (b) - (a)
extern "C" int eps_debug_expand_stamps(unsigned long long *out16, int reset)
{
    hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_ex_stamp), sizeof(unsigned long long) * 16);
    if (reset) { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_ex_stamp), z, sizeof(z)); }
    return 0;
}
#else
#define XSTAMP(var)
#define XSTAMP_ADD(i, a, b)
#endif

template <bool FILL, bool HAS_VAL, bool HAS_W>
__global__ __launch_bounds__(EX_THREADS) void expand_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, const float *__restrict__ val,
    const float *__restrict__ node_w, int32_t v_lo, int32_t v_hi, const int32_t *__restrict__ col_order, int32_t wpt,
    unsigned int *__restrict__ next_col,
    int64_t *__restrict__ cand_count, const int64_t *__restrict__ colptr, int32_t *__restrict__ cand_u,
    int32_t *__restrict__ cand_v, int32_t seen_words, int32_t *__restrict__ out_cn, int32_t *__restrict__ cn_second,
    int32_t *__restrict__ cn_later, float *__restrict__ out_score, float *__restrict__ score_second,
    unsigned long long *__restrict__ score_later)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int words = wpt * EX_THREADS;
    uint32_t *bm = lds;                            // bit u: u is a 2-hop endpoint of the column
    uint32_t *base32 = lds + words;                // FILL only from here.  rank of the first bit of every 8-word group
    uint8_t *pre8 = (uint8_t *)(base32 + words / 8);  // rank of a word's first bit within its group (<= 224)
    unsigned long long *hot_acc = (unsigned long long *)(base32 + words / 8 + words / 4);  // 8-byte aligned: words % 1024 == 0
    uint32_t *hot_key = (uint32_t *)(hot_acc + EX_HOT);  // candidate rank owning the entry
    uint32_t *hot_cn = hot_key + EX_HOT;
    uint32_t *seen = hot_cn + EX_HOT;                 // bit (rank >> shift): a path has reached this candidate
    uint32_t *seen2 = seen + seen_words;              // ... and a second one
    __shared__ int s_wave_tot[EX_WAVES];
    __shared__ unsigned int s_col;
    __shared__ int s_long[EX_LONGQ];
    __shared__ int s_nlong;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wib = __builtin_amdgcn_readfirstlane(tid >> 6);

    // the bitmap is all-zero between columns: every column clears exactly the words it scanned
    for (int i = tid; i < words; i += EX_THREADS) bm[i] = 0u;
    if (tid == 0) s_nlong = 0;
    if (FILL) {
        for (int i = tid; i < 2 * seen_words; i += EX_THREADS) seen[i] = 0u;
        for (int i = tid; i < EX_HOT; i += EX_THREADS) {
            hot_acc[i] = 0ull;
            hot_key[i] = EX_HOT_EMPTY;
            hot_cn[i] = 0u;
        }
    }

#ifdef EX_STAMP
    unsigned long long xst[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (;;) {
        XSTAMP(t0);
        __syncthreads();
        if (tid == 0) s_col = atomicAdd(next_col, 1u);
        __syncthreads();
        if ((int64_t)v_lo + s_col >= v_hi) break;
        XSTAMP(t1);
        XSTAMP_ADD(0, t0, t1);
        const int64_t v = (int64_t)v_lo + (col_order ? (uint32_t)col_order[s_col] : s_col);
        const int64_t vb = rowptr[v];
        const int32_t dv = (int32_t)(rowptr[v + 1] - vb);
        const int32_t *__restrict__ vcol = col + vb;
        if (dv == 0) {  // no neighbours -> no candidates
            if (!FILL && tid == 0) cand_count[v - v_lo] = 0;
            continue;
        }

        // ---- A. mark every 2-hop endpoint --------------------------------------------------
        for_each_path(rowptr, col, vcol, dv, wib, lane, s_long, &s_nlong, [&](int, int64_t, int, v4i u4, int nvalid) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (e >= nvalid) continue;
                const uint32_t u = (uint32_t)u4[e];
                atomicOr(&bm[u >> 5], 1u << (u & 31));
            }
        });
        __syncthreads();
        for (int k = tid; k < dv; k += EX_THREADS) {  // known edges out
            const uint32_t u = (uint32_t)vcol[k];
            atomicAnd(&bm[u >> 5], ~(1u << (u & 31)));
        }
        if (tid == 0) atomicAnd(&bm[(uint32_t)v >> 5], ~(1u << ((uint32_t)v & 31)));  // diagonal out
        __syncthreads();
        XSTAMP(t2);
        XSTAMP_ADD(1, t1, t2);

        // ---- B. rank: exclusive prefix of the per-word popcounts ---------------------------------
        const int w0 = tid * wpt;
        int local = 0;
        for (int i = 0; i < wpt; ++i) local += __popc(bm[w0 + i]);
        const int incl = wave_incl_scan(local, lane);
        if (lane == 63) s_wave_tot[wib] = incl;
        __syncthreads();
        int wave_base = 0, total = 0;
#pragma unroll
        for (int i = 0; i < EX_WAVES; ++i) {
            const int t = s_wave_tot[i];
            if (i < wib) wave_base += t;
            total += t;
        }
        int run = wave_base + incl - local;  // exclusive prefix of this thread's first word
        XSTAMP(t3);
        XSTAMP_ADD(2, t2, t3);

        if (!FILL) {
            if (tid == 0) cand_count[v - v_lo] = total;
            for (int i = 0; i < wpt; ++i) bm[w0 + i] = 0u;
            continue;
        }

        // ---- C. emit the candidates of this column in ascending u ---------------------------------
        const int64_t base = colptr[v - v_lo];
        const int64_t base_off = base;
        const int run0 = run;
        for (int i = 0; i < wpt; ++i) {
            uint32_t bits = bm[w0 + i];
            if (((w0 + i) & 7) == 0) base32[(w0 + i) >> 3] = (uint32_t)run;
            while (bits) {
                const int b = __builtin_ctz(bits);
                bits &= bits - 1;
                cand_u[base + run] = (w0 + i) * 32 + b;
                ++run;
            }
        }
        if (cand_v)  // one value for the whole column: whole lines, not one scattered store per candidate
            for (int i = tid; i < total; i += EX_THREADS) cand_v[base + i] = (int32_t)v;
        __syncthreads();
        run = run0;
        for (int i = 0; i < wpt; ++i) {  // group bases are complete: ranks relative to them fit a byte
            pre8[w0 + i] = (uint8_t)((uint32_t)run - base32[(w0 + i) >> 3]);
            run += __popc(bm[w0 + i]);
        }
        __syncthreads();

        XSTAMP(t4);
        XSTAMP_ADD(3, t3, t4);
        // ---- D. score: walk the paths again, give each term to its candidate's slot -------------------
        if (out_score || out_cn) {
            // arrival bitmaps over the candidate ranks of this column; columns with more candidates than bits share a bit
            // among 2^shift neighbouring ranks (an array still gets at most one plain store per bit, hence per slot)
            const int ncand = total;
            int shift = 0;
            while ((ncand >> shift) >= seen_words * 32) ++shift;
            if (shift > 0) {
                // with shared bits a candidate's first path may be taken for a second one, and cn[] / score[] (which
                // the caller does not zero: normally every slot is stored once) would keep a slot unwritten
                for (int i = tid; i < ncand; i += EX_THREADS) {
                    if (out_cn) out_cn[base_off + i] = 0;
                    if (out_score) out_score[base_off + i] = 0.0f;
                }
                __syncthreads();
            }
            XSTAMP(t5);
            XSTAMP_ADD(4, t4, t5);
            // The four entries of a lane go through the LDS look-ups STAGE BY STAGE (all four bitmap words, then all four
            // rank reads, then all four arrival tests ...): every stage is one LDS round trip for the four together,
            // where entry-by-entry code made up to five dependent round trips per entry.  An entry that has dropped out
            // still issues its (harmless) operation -- OR of 0, CAS that cannot match -- so no stage hides in a branch.
            for_each_path(rowptr, col, vcol, dv, wib, lane, s_long, &s_nlong, [&](int k, int64_t wb, int base, v4i u4, int nvalid) {
                float vw = 1.0f;                          // A[v,w] * node_w[w]: the A_ entry (adamic_utils.py:17)
                if (HAS_VAL) vw = val[vb + k];
                if (HAS_W) vw = vw * node_w[vcol[k]];
                uint32_t u[4], word[4], rank[4], sbit[4], sw[4], old1[4], old2[4], h[4], owner[4];
                bool cand[4], later[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) u[e] = e < nvalid ? (uint32_t)u4[e] : 0u;
#pragma unroll
                for (int e = 0; e < 4; ++e) word[e] = bm[u[e] >> 5];
#pragma unroll
                for (int e = 0; e < 4; ++e) rank[e] = base32[u[e] >> 8] + pre8[u[e] >> 5];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    cand[e] = e < nvalid && ((word[e] >> (u[e] & 31)) & 1u);
                    rank[e] += __popc(word[e] & ((1u << (u[e] & 31)) - 1u));
                    const uint32_t sb = rank[e] >> shift;
                    sbit[e] = cand[e] ? 1u << (sb & 31) : 0u;
                    sw[e] = cand[e] ? sb >> 5 : 0u;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) old1[e] = atomicOr(&seen[sw[e]], sbit[e]);
#pragma unroll
                for (int e = 0; e < 4; ++e) old2[e] = atomicOr(&seen2[sw[e]], old1[e] & sbit[e]);  // only if seen before
                // Third and later paths: the candidates with the longest chains (a hub pair has thousands of common
                // neighbours, and atomics on ONE address serialise at the memory side) come back in almost every row, so
                // they are the first to ask for an entry of the LDS table and keep it for the column; a candidate that
                // finds its entry taken uses the global accumulator throughout.
                bool any_later = false;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    later[e] = (old1[e] & old2[e] & sbit[e]) != 0u;
                    any_later |= later[e];
                    h[e] = later[e] ? (rank[e] * 2654435761u) >> 22 : (uint32_t)(lane + 64 * e);
                    owner[e] = rank[e] + 1u;              // "not mine" unless the CAS below says otherwise
                }
                if (__any(any_later)) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)           // a lane without a later path compares with a value no key can hold
                        owner[e] = atomicCAS(&hot_key[h[e]], later[e] ? EX_HOT_EMPTY : 0xFFFFFFFEu, rank[e]);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (!cand[e]) continue;
                    const int64_t slot = base_off + rank[e];
                    const bool first = !(old1[e] & sbit[e]);
                    const bool second = !first && !(old2[e] & sbit[e]);
                    const bool hot = later[e] && (owner[e] == EX_HOT_EMPTY || owner[e] == rank[e]);
                    if (out_cn) {
                        if (first) out_cn[slot] = 1;
                        else if (second) cn_second[slot] = 1;
                        else if (hot) atomicAdd(&hot_cn[h[e]], 1u);
                        else atomicAdd(&cn_later[slot], 1);
                    }
                    if (out_score) {
                        float term = vw;                  // A[u,w] * (A[v,w] * node_w[w]), float32 like the reference
                        if (HAS_VAL) term = val[wb + base + e] * vw;
                        if (first) out_score[slot] = term;
                        else if (second) score_second[slot] = term;
                        else if (hot) atomicAdd(&hot_acc[h[e]], (unsigned long long)ex_to_fixed(term));
                        else atomicAdd(&score_later[slot], (unsigned long long)ex_to_fixed(term));
                    }
                }
            });
            XSTAMP(t6);
            XSTAMP_ADD(5, t5, t6);
            __syncthreads();
            XSTAMP(t7);
            XSTAMP_ADD(6, t6, t7);
            for (int i = tid; i <= (ncand >> shift) / 32; i += EX_THREADS) seen[i] = seen2[i] = 0u;
            for (int i = tid; i < EX_HOT; i += EX_THREADS) {  // the owner of an entry is the only writer of its slot
                const uint32_t r = hot_key[i];
                if (r != EX_HOT_EMPTY) {
                    if (out_score) score_later[base_off + r] = hot_acc[i];
                    if (out_cn) cn_later[base_off + r] = (int32_t)hot_cn[i];
                    hot_acc[i] = 0ull;
                    hot_key[i] = EX_HOT_EMPTY;
                    hot_cn[i] = 0u;
                }
            }
        }
        for (int i = 0; i < wpt; ++i) bm[w0 + i] = 0u;
        XSTAMP(t8);
        XSTAMP_ADD(7, t4, t8);
    }
#ifdef EX_STAMP
    if (FILL && tid == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&g_ex_stamp[i], xst[i]);
#endif
}

// Completes score[] / cn[] of a fill launch: adds what the second and the later paths of a candidate left in the scratch
// arrays.  Every term goes through the same fixed-point conversion, so the sum does not depend on which path came first.
__global__ void expand_finish_kernel(int64_t n, int32_t *__restrict__ cn, const int32_t *__restrict__ cn_second,
                                     const int32_t *__restrict__ cn_later, float *__restrict__ score,
                                     const float *__restrict__ score_second, const long long *__restrict__ score_later)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (score) {
            const long long t = ex_to_fixed(score[i]) + ex_to_fixed(score_second[i]) + score_later[i];
            score[i] = (float)((double)t * (1.0 / (double)(1ll << EX_FIXED_SHIFT)));
        }
        if (cn) cn[i] = cn[i] + cn_second[i] + cn_later[i];
    }
}

static int expand_words_per_thread(int64_t n_nodes)
{
    const int64_t words = (n_nodes + 31) / 32;
    return (int)((words + EX_THREADS - 1) / EX_THREADS);
}

extern "C" int eps_expand_max_nodes(void) { return EX_MAX_WPT * EX_THREADS * 32; }

// The fill kernel always takes EX_FILL_LDS bytes: bitmap 4 B + group bases 0.5 B + byte ranks 1 B per word, the hot
// table (16 B per entry), and the two arrival bitmaps share what is left.
static int expand_seen_words(int wpt) { return (int)((EX_FILL_LDS - (size_t)wpt * EX_THREADS * 11 / 2 - EX_HOT * 16) / 8); }

// Scratch layout behind eps_expand_workspace_bytes: [score_later i64 x n][score_second f32 x n] then
// [cn_second i32 x n][cn_later i32 x n], each part only when requested.
struct ExpandScratch {
    long long *score_later = nullptr;
    float *score_second = nullptr;
    int32_t *cn_second = nullptr, *cn_later = nullptr;
};
static ExpandScratch expand_scratch(void *workspace, int64_t n, bool want_cn, bool want_score)
{
    ExpandScratch w;
    char *p = (char *)workspace;
    if (want_score) {
        w.score_later = (long long *)p;
        p += n * 8;
        w.score_second = (float *)p;
        p += n * 4;
    }
    if (want_cn) {
        w.cn_second = (int32_t *)p;
        p += n * 4;
        w.cn_later = (int32_t *)p;
    }
    return w;
}

extern "C" int64_t eps_expand_workspace_bytes(int64_t n_cand, int want_cn, int want_score)
{
    if (n_cand < 0) return 0;
    return n_cand * ((want_score ? 12 : 0) + (want_cn ? 8 : 0));
}

extern "C" int eps_expand_count(const int64_t *rowptr, const int32_t *col, int64_t n_nodes, int64_t v_lo, int64_t v_hi,
                                const int32_t *col_order, int64_t *cand_count, void *stream)
{
    EPS_REQUIRE(n_nodes >= 0 && v_lo >= 0 && v_hi >= v_lo && v_hi <= n_nodes, "eps_expand_count: bad column range");
    if (v_hi == v_lo) return EPS_OK;
    EPS_REQUIRE(rowptr && col && cand_count, "eps_expand_count: null pointer");
    const int wpt = expand_words_per_thread(n_nodes);
    EPS_REQUIRE(wpt <= EX_MAX_WPT, "eps_expand_count: %lld nodes exceed the LDS bitmap (max %d)",
                (long long)n_nodes, eps_expand_max_nodes());
    hipStream_t s = (hipStream_t)stream;
    unsigned int *counter = nullptr;
    int rc = eps_take_counter(&counter, s, "eps_expand_count");
    if (rc) return rc;
    const size_t lds = (size_t)wpt * EX_THREADS * 4;
    auto kern = expand_kernel<false, false, false>;
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        eps_set_error("eps_expand_count: cannot reserve %zu bytes of LDS", lds);
        return EPS_ELAUNCH;
    }
    int64_t blocks = (int64_t)eps_num_cus() * (lds * 2 + 1024 <= 163840 ? 2 : 1);
    if (blocks > v_hi - v_lo) blocks = v_hi - v_lo;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(EX_THREADS), lds, s, rowptr, col, (const float *)nullptr,
                       (const float *)nullptr, (int32_t)v_lo, (int32_t)v_hi, col_order, wpt, counter, cand_count,
                       (const int64_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr, 0, (int32_t *)nullptr,
                       (int32_t *)nullptr, (int32_t *)nullptr, (float *)nullptr, (float *)nullptr,
                       (unsigned long long *)nullptr);
    EPS_CHECK_LAUNCH("eps_expand_count");
    return EPS_OK;
}

extern "C" int eps_expand_fill(const int64_t *rowptr, const int32_t *col, const float *val, const float *node_w,
                               int64_t n_nodes, int64_t v_lo, int64_t v_hi, const int32_t *col_order,
                               const int64_t *colptr, int64_t n_cand, int32_t *cand_u, int32_t *cand_v, int32_t *cn,
                               float *score, void *workspace, void *stream)
{
    EPS_REQUIRE(n_nodes >= 0 && v_lo >= 0 && v_hi >= v_lo && v_hi <= n_nodes, "eps_expand_fill: bad column range");
    if (v_hi == v_lo) return EPS_OK;
    EPS_REQUIRE(rowptr && col && colptr && cand_u && n_cand >= 0, "eps_expand_fill: null pointer");
    EPS_REQUIRE(workspace || !(cn || score) || n_cand == 0, "eps_expand_fill: cn / score need the zeroed workspace");
    EPS_REQUIRE(((uintptr_t)workspace & 7) == 0, "eps_expand_fill: workspace must be 8-byte aligned");
    const int wpt = expand_words_per_thread(n_nodes);
    EPS_REQUIRE(wpt <= EX_MAX_WPT, "eps_expand_fill: %lld nodes exceed the LDS bitmap (max %d)",
                (long long)n_nodes, eps_expand_max_nodes());
    hipStream_t s = (hipStream_t)stream;
    unsigned int *counter = nullptr;
    int rc = eps_take_counter(&counter, s, "eps_expand_fill");
    if (rc) return rc;
    const size_t lds = EX_FILL_LDS;
    int seen_words = expand_seen_words(wpt);
    if (const char *dbg = getenv("EPS_DEBUG_SEEN_WORDS")) {  // tests: force the shared-bit path on small graphs
        const int v = atoi(dbg);
        if (v >= 1 && v < seen_words) seen_words = v;
    }
    const ExpandScratch ws = expand_scratch(workspace, n_cand, cn != nullptr, score != nullptr);
    int64_t blocks = eps_num_cus();
    if (blocks > v_hi - v_lo) blocks = v_hi - v_lo;
    const bool hv = val != nullptr, hw = node_w != nullptr;
#define EX_LAUNCH(HV, HW)                                                                                              \
    do {                                                                                                               \
        auto kern = expand_kernel<true, HV, HW>;                                                                       \
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=           \
            hipSuccess) {                                                                                              \
            eps_set_error("eps_expand_fill: cannot reserve %zu bytes of LDS", lds);                                    \
            return EPS_ELAUNCH;                                                                                        \
        }                                                                                                              \
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(EX_THREADS), lds, s, rowptr, col, val, node_w,           \
                           (int32_t)v_lo, (int32_t)v_hi, col_order, wpt, counter, (int64_t *)nullptr, colptr, cand_u,  \
                           cand_v, seen_words, cn, ws.cn_second, ws.cn_later, score, ws.score_second,                  \
                           (unsigned long long *)ws.score_later);                                                      \
    } while (0)
    if (hv && hw) EX_LAUNCH(true, true);
    else if (hv) EX_LAUNCH(true, false);
    else if (hw) EX_LAUNCH(false, true);
    else EX_LAUNCH(false, false);
#undef EX_LAUNCH
    EPS_CHECK_LAUNCH("eps_expand_fill");
    return EPS_OK;
}

extern "C" int eps_expand_finish(int64_t n_cand, int32_t *cn, float *score, const void *workspace, void *stream)
{
    EPS_REQUIRE(n_cand >= 0, "eps_expand_finish: negative size");
    if (n_cand == 0 || (!cn && !score)) return EPS_OK;
    EPS_REQUIRE(workspace, "eps_expand_finish: null workspace");
    const ExpandScratch ws = expand_scratch((void *)workspace, n_cand, cn != nullptr, score != nullptr);
    int64_t b = (n_cand + 255) / 256;
    const int64_t cap = (int64_t)eps_num_cus() * 8;
    if (b > cap) b = cap;
    hipLaunchKernelGGL(expand_finish_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, n_cand, cn,
                       (const int32_t *)ws.cn_second, (const int32_t *)ws.cn_later, score, (const float *)ws.score_second,
                       (const long long *)ws.score_later);
    EPS_CHECK_LAUNCH("eps_expand_finish");
    return EPS_OK;
}
