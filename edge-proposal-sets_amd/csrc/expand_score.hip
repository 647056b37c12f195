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
//   A. mark: for every w in N(v) and every u in N(w): set bit u of an LDS bitmap
//      over the node ids; then clear the bits of N(v) and of v itself (known edges, diagonal).
//   B. rank: per-word popcounts -> block-wide exclusive scan -> prefix[] in LDS; the total is the
//      column's candidate count (kernel 1 stops here: counts -> host cumsum -> colptr).
//   C. emit: set bits in ascending order -> cand_u[colptr[v] + rank] (ascending u: the
//      reference's column-major order for free).
//      Columns may be handed out in a caller-given order (heaviest first keeps the tail of a
//      launch short: a hub column is one workgroup's work for milliseconds).
//   D. score: walk the same paths again.  Writing each term to its candidate's slot -- plain
//      stores or atomics -- scatters 4-byte writes over the column's whole output segment for
//      the length of the pass: the L2 evicts the lines half-written (590 M write-backs for
//      4.8 GB of output per launch) and every atomic is a line read-modify-write behind the
//      L2.  So the pass is BLOCKED instead ("propagation blocking"): pass A also counts the
//      paths per id range (512 ranges), the ranges are grouped into tiles of <= 8192 candidate
//      ranks, and D1 appends a record (rank in tile, term) to the tile's bucket in a per-
//      workgroup scratch buffer -- sequential streams, whole lines.  D2 then takes the tiles
//      one by one: bucket -> 64-bit FIXED-POINT (2^-40) accumulators and counters in LDS (they
//      reuse the bitmap's space) -> one coalesced store of the finished float32 scores / counts.
//      Integer addition is associative: bit-reproducible whatever the arrival order (float
//      atomics are not), exact up to the final rounding.  No global atomics, no zero-filled
//      accumulator arrays, no second kernel.
// Requires a SYMMETRIC adjacency (filter.py's always is: rank.py:33 to_symmetric) and
// N <= 851,968 node ids (bitmap + rank tables: 5.5 bytes per 32 ids of the 160 KiB LDS); the
// host falls back to the tensor-op expansion above that.
#include "eps_common.h"


#define EX_THREADS 1024
#define EX_WAVES (EX_THREADS / 64)
#define EX_FIXED_SHIFT 40
#ifndef EX_RING
#define EX_RING 2           // first units of a wave's next rows in flight (1: 60 % slower list pass; 4, 8: no faster)
#endif
#define EX_LONGQ 1024       // rows longer than one unit queued per column and pass (static LDS)
#define EX_MAX_WPT 26       // bitmap words per thread: 26 * 1024 words * 5.5 B = 143 KiB + 10 KiB of tables
#define EX_RANGES 512       // id ranges per column: path histogram and tile plan
#define EX_TILE 8192        // candidate ranks per tile: 64 KiB of accumulators + 32 KiB of counters, aliasing the bitmap
#define EX_TABLE_WORDS (5 * EX_RANGES + 1)


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
// dependent chain per row), the units of the next EX_RING rows in flight while the current one is consumed (raw buffer loads:
// out-of-range lanes read 0, no bounds branch) -- and the rows longer than one unit are queued in LDS; after a barrier
// their remaining units are dealt round-robin over the waves.  (Row lengths are heavy-tailed: with whole rows per wave
// the waves of a workgroup waited at the closing barrier for a third of the pass.)
// body(k, wb, base, u4, nvalid): entries [base, base+nvalid) of row w = vcol[k] (nvalid in 0..4 per lane).
// Ends with a workgroup barrier; the next call must be separated from this one by another barrier.
// Entries of row w = [wb, wb + dw) whose id lies in [win_lo, win_hi): rows are ascending, so two lower-bound searches
// (per lane, each lane its own row) narrow the descriptor.  Only graphs wider than the LDS bitmap take this path.
__device__ __forceinline__ void ex_row_window(const int32_t *__restrict__ col, int64_t &wb, int32_t &dw, int32_t win_lo,
                                              int32_t win_hi)
{
    int32_t lo = 0, hi = dw;
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if (col[wb + mid] < win_lo) lo = mid + 1; else hi = mid;
    }
    const int32_t first = lo;
    hi = dw;
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if (col[wb + mid] < win_hi) lo = mid + 1; else hi = mid;
    }
    wb += first;
    dw = lo - first;
}

// WINDOWED: walk only the entries with ids in [win_lo, win_hi) (body's `base` is then relative to the window's first entry
// of the row, `wb` points at it).
template <bool WINDOWED, typename Body>
__device__ __forceinline__ void for_each_path(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                              const int32_t *__restrict__ vcol, int32_t dv, int wib, int lane,
                                              int *s_long, int *s_nlong, int32_t win_lo, int32_t win_hi, Body body)
{
    for (int b0 = wib; b0 < dv; b0 += EX_WAVES * 64) {
        const int k_mine = b0 + EX_WAVES * lane;
        const bool ok = k_mine < dv;
        const int32_t w_mine = ok ? vcol[k_mine] : 0;
        int64_t wb_mine = rowptr[w_mine];
        int32_t dw_mine = ok ? (int32_t)(rowptr[w_mine + 1] - wb_mine) : 0;
        if (WINDOWED) ex_row_window(col, wb_mine, dw_mine, win_lo, win_hi);
        const int left = (dv - b0 + EX_WAVES - 1) / EX_WAVES;
        const int nrows = left < 64 ? left : 64;
        // ring of EX_RING rows in flight: every ring register is written by one unconditional load per trip (rows past
        // the batch get a zero-length descriptor: the load returns zeros and costs nothing)
        auto first_unit = [&](int jj) {
            const int jl = jj & 63;
            const int32_t dn = jj < nrows ? __builtin_amdgcn_readlane(dw_mine, jl) : 0;
            const __amdgpu_buffer_rsrc_t rs =
                __builtin_amdgcn_make_buffer_rsrc((void *)(col + ex_bcast64(wb_mine, jl)), 0, dn * 4, 0x00020000);
            return __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, 0, 0);
        };
        v4i ring[EX_RING];
#pragma unroll
        for (int r = 0; r < EX_RING; ++r) ring[r] = first_unit(r);
        for (int j0 = 0; j0 < nrows; j0 += EX_RING) {
#pragma unroll
            for (int r = 0; r < EX_RING; ++r) {
                const int j = j0 + r;
                if (j >= nrows) break;
                const int32_t dw = __builtin_amdgcn_readlane(dw_mine, j);
                const int64_t wb = ex_bcast64(wb_mine, j);
                const v4i cur = ring[r];
                ring[r] = first_unit(j + EX_RING);
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
                        const __amdgpu_buffer_rsrc_t rj =
                            __builtin_amdgcn_make_buffer_rsrc((void *)(col + wb), 0, dw * 4, 0x00020000);
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
    }
    __syncthreads();
    // the units past the first one of the queued rows: unit c of queue entry i belongs to wave (c + i) & 15
    const int nl = *s_nlong < EX_LONGQ ? *s_nlong : EX_LONGQ;
    for (int q0 = 0; q0 < nl; q0 += 64) {
        const bool ok = q0 + lane < nl;
        const int k_mine = ok ? s_long[q0 + lane] : 0;
        const int32_t w_mine = vcol[k_mine];
        int64_t wb_mine = rowptr[w_mine];
        int32_t dw_mine = ok ? (int32_t)(rowptr[w_mine + 1] - wb_mine) : 0;
        if (WINDOWED) ex_row_window(col, wb_mine, dw_mine, win_lo, win_hi);
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


template <bool FILL, bool HAS_VAL, bool HAS_W, bool WINDOWED>
__global__ __launch_bounds__(EX_THREADS) void expand_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, const float *__restrict__ val,
    const float *__restrict__ node_w, int32_t n_nodes, int32_t v_lo, int32_t v_hi, const int32_t *__restrict__ col_order,
    int32_t wpt, unsigned int *__restrict__ next_col, int64_t *__restrict__ cand_count, const int64_t *__restrict__ colptr,
    int32_t *__restrict__ cand_u, int32_t *__restrict__ cand_v, int32_t range_shift, int32_t tile_half,
    int32_t region_words, int32_t win_ids,
    uint2 *__restrict__ scratch, int64_t scratch_per_block, int32_t *__restrict__ out_cn, float *__restrict__ out_score,
    eps_score_cut *__restrict__ cut, unsigned int *__restrict__ overflow)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int words = wpt * EX_THREADS;
    uint32_t *bm = lds;                            // bit u: u is a 2-hop endpoint of the column
    uint32_t *base32 = lds + words;                // FILL only from here.  rank of the first bit of every 8-word group
    uint8_t *pre8 = (uint8_t *)(base32 + words / 8);  // rank of a word's first bit within its group (<= 224)
    // D2 reuses the space of the three arrays above (they are dead once D1 has binned the paths):
    unsigned long long *acc = (unsigned long long *)lds;         // EX_TILE fixed-point sums
    uint32_t *cnt = lds + 2 * EX_TILE;                           // EX_TILE path counts
    uint32_t *hist = lds + region_words;           // paths per id range (pass A), zero between columns
    uint32_t *rinfo = hist + EX_RANGES;            // (tile of the range << 20) | first rank of that tile
    uint32_t *tile_r0 = rinfo + EX_RANGES;         // first candidate rank of a tile; [n_tiles] = column total
    uint32_t *tile_base = tile_r0 + EX_RANGES + 1; // first record of the tile's bucket in this workgroup's scratch
    uint32_t *tile_cur = tile_base + EX_RANGES;    // next free record (absolute), D1's append cursor
    __shared__ int s_wave_tot[EX_WAVES];
    __shared__ unsigned int s_col;
    __shared__ int s_long[EX_LONGQ];
    __shared__ int s_nlong;
    __shared__ int s_ntiles;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wib = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool want_d = FILL && (out_score || out_cn || cut);
    const bool want_sum = out_score || cut;
    const float cut_thr = cut ? cut->threshold : 0.f;
    const uint32_t cut_cap = cut ? cut->capacity : 0u;
    int64_t *__restrict__ cut_pos = cut ? cut->pos : nullptr;
    float *__restrict__ cut_val = cut ? cut->val : nullptr;
    uint2 *__restrict__ my_scratch = scratch + (int64_t)blockIdx.x * scratch_per_block;

    // the bitmap is all-zero between columns: every column clears exactly the words it scanned
    for (int i = tid; i < words; i += EX_THREADS) bm[i] = 0u;
    if (tid == 0) s_nlong = 0;
    if (FILL)
        for (int i = tid; i < EX_RANGES; i += EX_THREADS) hist[i] = 0u;

    for (;;) {
        __syncthreads();
        if (tid == 0) s_col = atomicAdd(next_col, 1u);
        __syncthreads();
        if ((int64_t)v_lo + s_col >= v_hi) break;
        const int64_t v = (int64_t)v_lo + (col_order ? (uint32_t)col_order[s_col] : s_col);
        const int64_t vb = rowptr[v];
        const int32_t dv = (int32_t)(rowptr[v + 1] - vb);
        const int32_t *__restrict__ vcol = col + vb;
        if (dv == 0) {  // no neighbours -> no candidates (and no paths: an empty segment in either layout)
            if (cand_count && tid == 0) cand_count[v - v_lo] = 0;
            continue;
        }

        // Graphs wider than the LDS bitmap are expanded in id WINDOWS of win_ids ids: window after window, each walking
        // only the row segments inside it (rows are ascending) -- candidates still come out in ascending u.
        const int w0 = wib * 64 * wpt + lane;      // this thread's words: w0 + 64 * t
        const int64_t base = FILL ? colptr[v - v_lo] : 0;
        const int64_t seg_len = FILL ? colptr[v - v_lo + 1] - base : 0;
        int64_t col_off = 0;                       // candidates of the windows before this one
        bool seg_overflow = false;
        for (int32_t win_lo = 0; win_lo < n_nodes; win_lo += win_ids) {
            const int32_t win_hi = (n_nodes - win_lo > win_ids) ? win_lo + win_ids : n_nodes;
            const int n_ranges = ((win_hi - win_lo - 1) >> range_shift) + 1;

            // ---- A. mark every 2-hop endpoint (and count the paths per id range for the bucket sizes of D1) -------------
            for_each_path<WINDOWED>(rowptr, col, vcol, dv, wib, lane, s_long, &s_nlong, win_lo, win_hi,
                                    [&](int, int64_t, int, v4i u4, int nvalid) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (e >= nvalid) continue;
                    const uint32_t u = (uint32_t)(u4[e] - win_lo);
                    atomicOr(&bm[u >> 5], 1u << (u & 31));
                    if (want_d) atomicAdd(&hist[u >> range_shift], 1u);
                }
            });
            for (int k = tid; k < dv; k += EX_THREADS) {  // known edges out
                const int32_t x = vcol[k];
                if (WINDOWED && (x < win_lo || x >= win_hi)) continue;
                const uint32_t u = (uint32_t)(x - win_lo);
                atomicAnd(&bm[u >> 5], ~(1u << (u & 31)));
            }
            if (tid == 0 && v >= win_lo && v < win_hi) {      // diagonal out
                const uint32_t u = (uint32_t)(v - win_lo);
                atomicAnd(&bm[u >> 5], ~(1u << (u & 31)));
            }
            __syncthreads();

            // ---- B. rank: exclusive prefix of the per-word popcounts ---------------------------------
            // Word layout: wave i owns the words [i * 64 * wpt, (i + 1) * 64 * wpt); in trip t its 64 lanes read 64 CONSECUTIVE
            // words, so the candidates a wave emits in one trip are consecutive in rank and their stores land in a few lines.
            int local = 0;
            for (int i = 0; i < wpt; ++i) local += __popc(bm[w0 + 64 * i]);
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

            if (!FILL) {
                col_off += total;
                for (int i = 0; i < wpt; ++i) bm[w0 + 64 * i] = 0u;
                __syncthreads();                   // s_wave_tot is rewritten by the next window
                continue;
            }

            // ---- C. emit the candidates of this window in ascending u; rank tables for pass D ---------------------------
            // colptr may be an UPPER-BOUND layout (segments at least as long as the column's candidate count, e.g. a prefix
            // of the two-hop path counts, which needs no counting pass): then cand_count receives the real count and the
            // rest of the segment is padded (score -inf, cn 0, cand_u -1) so that the arrays stay in candidate order.
            if (col_off + total > seg_len) {           // the caller's bound does not hold: flag it, leave the rest out
                if (tid == 0 && overflow) atomicOr(overflow, 2u);
                seg_overflow = true;
                __syncthreads();
                for (int i = 0; i < wpt; ++i) bm[w0 + 64 * i] = 0u;
                if (tid < EX_RANGES) hist[tid] = 0u;
                break;
            }
            const int64_t wbase = base + col_off;      // first output slot of this window's candidates
            {
                int wrun = wave_base;                  // rank of the first bit of the wave's current 64 words
                for (int i = 0; i < wpt; ++i) {
                    const int wi = w0 + 64 * i;
                    uint32_t bits = bm[wi];
                    const int c = __popc(bits);
                    const int inc = wave_incl_scan(c, lane);
                    int run = wrun + inc - c;          // rank of this word's first bit
                    const int gbase = __shfl(run, lane & ~7);   // ... of its 8-word group's first bit (the 8 lanes are neighbours)
                    if ((lane & 7) == 0) base32[wi >> 3] = (uint32_t)run;
                    pre8[wi] = (uint8_t)(run - gbase);
                    while (bits) {
                        const int b = __builtin_ctz(bits);
                        bits &= bits - 1;
                        cand_u[wbase + run] = win_lo + wi * 32 + b;
                        ++run;
                    }
                    wrun += __builtin_amdgcn_readlane(inc, 63);
                }
            }
            if (cand_v)  // one value for the whole column: whole lines, not one scattered store per candidate
                for (int i = tid; i < total; i += EX_THREADS) cand_v[wbase + i] = (int32_t)v;
            col_off += total;
            __syncthreads();   // rank tables complete; also orders the s_wave_tot reads above against the plan's scan below
            if (!want_d) {
                for (int i = 0; i < wpt; ++i) bm[w0 + 64 * i] = 0u;
                __syncthreads();                   // the next window marks into the words other threads just cleared
                continue;
            }

            // ---- plan: group the id ranges into tiles of <= EX_TILE candidate ranks; bucket offsets from the histogram ------
            // A range holds at most 2^range_shift <= tile_half (= EX_TILE/2) candidates, so tile = (rank at range start) /
            // tile_half never skips a tile id and a tile never spans more than 2 * tile_half <= EX_TILE ranks.
            {
                const bool in = tid < n_ranges;
                const uint32_t rs = in ? base32[tid << (range_shift - 8)] : 0u;       // rank at the start of the range
                const uint32_t paths = in ? hist[tid] : 0u;
                if (in) hist[tid] = 0u;
                const int pin = wave_incl_scan((int)paths, lane);
                if (lane == 63) s_wave_tot[wib] = pin;
                __syncthreads();
                uint32_t pbase = 0, ptotal = 0;
#pragma unroll
                for (int i = 0; i < EX_WAVES; ++i) {
                    const uint32_t t = (uint32_t)s_wave_tot[i];
                    if (i < wib) pbase += t;
                    ptotal += t;
                }
                pbase += (uint32_t)pin - paths;                                        // exclusive prefix: bucket start if first
                const uint32_t tile = rs / (uint32_t)tile_half;
                const uint32_t rs_prev = (in && tid > 0) ? base32[(tid - 1) << (range_shift - 8)] : 0u;
                if (in && (tid == 0 || rs_prev / (uint32_t)tile_half != tile)) {
                    tile_r0[tile] = rs;
                    tile_base[tile] = pbase;
                    tile_cur[tile] = pbase;
                }
                if (tid == n_ranges - 1) {
                    s_ntiles = (int)tile + 1;
                    tile_r0[tile + 1] = (uint32_t)total;
                    if ((int64_t)ptotal > scratch_per_block) {   // the host sized the scratch from the path counts: cannot happen
                        s_ntiles = 0;
                        atomicOr(overflow, 1u);
                    }
                }
                __syncthreads();
                if (in) rinfo[tid] = (tile << 20) | tile_r0[tile];
                __syncthreads();
            }
            const int n_tiles = s_ntiles;

            // ---- D1. bin: walk the paths again, append (rank in tile, term) to the tile's bucket ---------------------------
            // The four entries of a lane go through the LDS look-ups stage by stage (all four bitmap words, then all four
            // rank reads, ...): one LDS round trip per stage for the four together.  An entry that is not a candidate still
            // issues its look-ups (on id 0) but takes no record.
            if (n_tiles > 0)
                for_each_path<WINDOWED>(rowptr, col, vcol, dv, wib, lane, s_long, &s_nlong, win_lo, win_hi,
                                        [&](int k, int64_t wb, int base_e, v4i u4, int nvalid) {
                    // A path's term is (A[u,w] * A[v,w]) * node_w[w] in float32.  adamic_utils.py:17-23 associates it as
                    // A[u,w] * (A[v,w] * mult[w]) (one ulp apart at most, far inside the 1e-5 gate); this association is
                    // symmetric in (u, v), so a pair's two orientations carry the same bits and the symmetric-half scan
                    // (scan_pieces.hip) reproduces them.  Unit-valued graphs: the term is node_w[w] either way.
                    float vw = 1.0f, nw = 1.0f;
                    if (HAS_VAL) vw = val[vb + k];
                    if (HAS_W) nw = node_w[vcol[k]];
                    uint32_t u[4], word[4], rank[4], ri[4], pos[4];
                    bool cand[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) u[e] = e < nvalid ? (uint32_t)(u4[e] - win_lo) : 0u;
#pragma unroll
                    for (int e = 0; e < 4; ++e) word[e] = bm[u[e] >> 5];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        rank[e] = base32[u[e] >> 8] + pre8[u[e] >> 5];
                        ri[e] = rinfo[u[e] >> range_shift];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        cand[e] = e < nvalid && ((word[e] >> (u[e] & 31)) & 1u);
                        rank[e] += __popc(word[e] & ((1u << (u[e] & 31)) - 1u));
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) pos[e] = cand[e] ? atomicAdd(&tile_cur[ri[e] >> 20], 1u) : 0u;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (!cand[e]) continue;
                        float term = nw;
                        if (HAS_VAL) term = (val[wb + base_e + e] * vw) * nw;
                        my_scratch[pos[e]] = make_uint2(rank[e] - (ri[e] & 0xFFFFFu), __builtin_bit_cast(uint32_t, term));
                    }
                });

            // ---- D2. per tile: bucket -> fixed-point sums and counts in LDS -> coalesced float32 scores / int32 counts ----------
            for (int t = 0; t < n_tiles; ++t) {
                const uint32_t r0 = tile_r0[t], nslots = tile_r0[t + 1] - r0;
                const uint32_t b0 = tile_base[t], n = tile_cur[t] - b0;
                for (uint32_t i = tid; i < nslots; i += EX_THREADS) {
                    acc[i] = 0ull;
                    cnt[i] = 0u;
                }
                __syncthreads();
                for (uint32_t i0 = tid; i0 < n; i0 += 4 * EX_THREADS) {   // four records in flight per thread
                    uint2 rec[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint32_t i = i0 + q * EX_THREADS;
                        rec[q] = my_scratch[b0 + (i < n ? i : i0)];
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (i0 + q * EX_THREADS >= n) break;
                        if (want_sum)
                            atomicAdd(&acc[rec[q].x], (unsigned long long)ex_to_fixed(__builtin_bit_cast(float, rec[q].y)));
                        if (out_cn) atomicAdd(&cnt[rec[q].x], 1u);
                    }
                }
                __syncthreads();
                for (uint32_t i = tid; i < nslots; i += EX_THREADS) {
                    if (want_sum) {
                        // the heuristics' terms are non-negative: a negative sum is one that wrapped past 2^23 (backstop; the
                        // host checks a bound of the graph's scores before it takes this path -- candidates.fused_scores_fit)
                        if ((long long)acc[i] < 0 && overflow) atomicOr(overflow, 4u);
                        const float sc = (float)((double)(long long)acc[i] * (1.0 / (double)(1ll << EX_FIXED_SHIFT)));
                        if (out_score) out_score[wbase + r0 + i] = sc;
                        if (cut && sc > cut_thr) {      // top-K cut in the kernel: report the few candidates above the bar
                            const uint32_t q = atomicAdd(&cut->count, 1u);
                            if (q < cut_cap) {
                                cut_pos[q] = wbase + r0 + i;
                                cut_val[q] = sc;
                            }
                        }
                    }
                    if (out_cn) out_cn[wbase + r0 + i] = (int32_t)cnt[i];
                }
                __syncthreads();
            }
            for (int i = 0; i < wpt; ++i) bm[w0 + 64 * i] = 0u;   // D2 left its accumulators in the bitmap's space
            __syncthreads();                                       // ... which the next window marks into
        }
        if (cand_count && tid == 0 && !seg_overflow) cand_count[v - v_lo] = col_off;
        if (FILL && !seg_overflow)
            for (int64_t i = col_off + tid; i < seg_len; i += EX_THREADS) {   // padding of an upper-bound segment
                cand_u[base + i] = -1;
                if (cand_v) cand_v[base + i] = (int32_t)v;
                if (out_score) out_score[base + i] = -__builtin_inff();
                if (out_cn) out_cn[base + i] = 0;
            }
    }
}

// bitmap words per thread: the whole id space when it fits the LDS, else the widest window (ids are then expanded in
// windows of wpt * 1024 * 32 ids)
static int expand_words_per_thread(int64_t n_nodes)
{
    const int64_t words = (n_nodes + 31) / 32;
    const int64_t wpt = (words + EX_THREADS - 1) / EX_THREADS;
    return (int)(wpt > EX_MAX_WPT ? EX_MAX_WPT : wpt);
}

static int64_t expand_window_ids(int wpt) { return (int64_t)wpt * EX_THREADS * 32; }

extern "C" int eps_expand_max_nodes(void) { return EX_MAX_WPT * EX_THREADS * 32; }

// id ranges of the path histogram: at most EX_RANGES of them, each a whole number of 256-id groups and no wider than
// EX_TILE / 2 ids (so a range never holds more candidates than half a tile)
static int expand_range_shift(int64_t n_nodes)      // n_nodes: ids per window
{
    int s = 8;
    while (((n_nodes - 1) >> s) + 1 > EX_RANGES) ++s;
    return s;
}

// words of the LDS region shared by {bitmap, group bases, byte ranks} and {tile accumulators, tile counters}
static int expand_region_words(int wpt)
{
    const int a = wpt * EX_THREADS * 11 / 8, b = 3 * EX_TILE;
    return a > b ? a : b;
}

extern "C" int eps_expand_count(const int64_t *rowptr, const int32_t *col, int64_t n_nodes, int64_t v_lo, int64_t v_hi,
                                const int32_t *col_order, int64_t *cand_count, void *stream)
{
    EPS_REQUIRE(n_nodes >= 0 && v_lo >= 0 && v_hi >= v_lo && v_hi <= n_nodes, "eps_expand_count: bad column range");
    if (v_hi == v_lo) return EPS_OK;
    EPS_REQUIRE(rowptr && col && cand_count, "eps_expand_count: null pointer");
    EPS_REQUIRE(n_nodes < (1ll << 31), "eps_expand_count: node ids are int32");
    const int wpt = expand_words_per_thread(n_nodes);
    const int64_t win_ids = expand_window_ids(wpt);
    const bool windowed = n_nodes > win_ids;
    hipStream_t s = (hipStream_t)stream;
    unsigned int *counter = nullptr;
    int rc = eps_take_counter(&counter, s, "eps_expand_count");
    if (rc) return rc;
    const size_t lds = (size_t)wpt * EX_THREADS * 4;
    int64_t blocks = (int64_t)eps_num_cus() * (lds * 2 + 2 * 8192 <= 163840 ? 2 : 1);
    if (blocks > v_hi - v_lo) blocks = v_hi - v_lo;
#define EX_COUNT(W)                                                                                                    \
    do {                                                                                                               \
        auto kern = expand_kernel<false, false, false, W>;                                                             \
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=           \
            hipSuccess) {                                                                                              \
            eps_set_error("eps_expand_count: cannot reserve %zu bytes of LDS", lds);                                   \
            return EPS_ELAUNCH;                                                                                        \
        }                                                                                                              \
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(EX_THREADS), lds, s, rowptr, col, (const float *)nullptr, \
                           (const float *)nullptr, (int32_t)n_nodes, (int32_t)v_lo, (int32_t)v_hi, col_order, wpt,     \
                           counter, cand_count, (const int64_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr, 8,   \
                           EX_TILE / 2, 0, (int32_t)win_ids, (uint2 *)nullptr, (int64_t)0, (int32_t *)nullptr,         \
                           (float *)nullptr, (eps_score_cut *)nullptr, (unsigned int *)nullptr);                       \
    } while (0)
    if (windowed) EX_COUNT(true);
    else EX_COUNT(false);
#undef EX_COUNT
    EPS_CHECK_LAUNCH("eps_expand_count");
    return EPS_OK;
}

// Workspace of a fill launch: 8 bytes of status (device word 0, non-zero after the launch = outputs invalid: bit 0 a
// column had more two-hop paths than the buckets were sized for, bit 1 a column had more candidates than its colptr
// segment, bit 2 a fixed-point sum left the accumulators' range) + one bucket area per workgroup, 8 bytes per path of the heaviest column (none needed without cn / score).
extern "C" int64_t eps_expand_workspace_bytes(int64_t max_col_paths)
{
    if (max_col_paths < 0) return 0;
    return 8 + (int64_t)eps_num_cus() * max_col_paths * 8;
}

extern "C" int eps_expand_fill(const int64_t *rowptr, const int32_t *col, const float *val, const float *node_w,
                               int64_t n_nodes, int64_t v_lo, int64_t v_hi, const int32_t *col_order,
                               const int64_t *colptr, int64_t *cand_count, int32_t *cand_u, int32_t *cand_v, int32_t *cn,
                               float *score, eps_score_cut *cut, void *workspace, int64_t workspace_bytes, void *stream)
{
    return eps_expand_fill_tiled(rowptr, col, val, node_w, n_nodes, v_lo, v_hi, col_order, colptr, cand_count, cand_u,
                                 cand_v, cn, score, cut, workspace, workspace_bytes, 0, stream);
}

extern "C" int eps_expand_fill_tiled(const int64_t *rowptr, const int32_t *col, const float *val, const float *node_w,
                                     int64_t n_nodes, int64_t v_lo, int64_t v_hi, const int32_t *col_order,
                                     const int64_t *colptr, int64_t *cand_count, int32_t *cand_u, int32_t *cand_v,
                                     int32_t *cn, float *score, eps_score_cut *cut, void *workspace,
                                     int64_t workspace_bytes, int32_t tile_ranks, void *stream)
{
    EPS_REQUIRE(n_nodes >= 0 && v_lo >= 0 && v_hi >= v_lo && v_hi <= n_nodes, "eps_expand_fill: bad column range");
    if (v_hi == v_lo) return EPS_OK;
    EPS_REQUIRE(rowptr && col && colptr && cand_u, "eps_expand_fill: null pointer");
    EPS_REQUIRE(n_nodes < (1ll << 31), "eps_expand_fill: node ids are int32");
    const bool scored = cn || score || cut;
    EPS_REQUIRE(workspace && workspace_bytes >= 8 && ((uintptr_t)workspace & 7) == 0,
                "eps_expand_fill: needs an 8-byte aligned workspace (eps_expand_workspace_bytes)");
    const int wpt = expand_words_per_thread(n_nodes);
    const int64_t win_ids = expand_window_ids(wpt);       // < 2^20: candidate ranks of a window are packed in 20 bits
    const bool windowed = n_nodes > win_ids;
    hipStream_t s = (hipStream_t)stream;
    unsigned int *counter = nullptr;
    int rc = eps_take_counter(&counter, s, "eps_expand_fill");
    if (rc) return rc;
    const int region_words = expand_region_words(wpt);
    const size_t lds = ((size_t)region_words + EX_TABLE_WORDS) * 4;
    int64_t blocks = eps_num_cus();
    const int64_t per_block = scored ? (workspace_bytes - 8) / 8 / blocks : 0;   // records per workgroup (sized for all CUs)
    if (blocks > v_hi - v_lo) blocks = v_hi - v_lo;
    if (hipMemsetAsync(workspace, 0, 8, s) != hipSuccess) {
        eps_set_error("eps_expand_fill: cannot reset the status word");
        return EPS_ELAUNCH;
    }
    const int range_shift = expand_range_shift(windowed ? win_ids : n_nodes);
    int tile_half = EX_TILE / 2;
    if (tile_ranks) {      // an explicit, smaller tile (a range of ids must still fit half a tile)
        EPS_REQUIRE(tile_ranks % 2 == 0 && tile_ranks / 2 >= (1 << range_shift) && tile_ranks <= EX_TILE,
                    "eps_expand_fill_tiled: tile_ranks must be even, in [%d, %d]", 2 << range_shift, EX_TILE);
        tile_half = tile_ranks / 2;
    }
    const bool hv = val != nullptr, hw = node_w != nullptr;
#define EX_LAUNCH(HV, HW, W)                                                                                           \
    do {                                                                                                               \
        auto kern = expand_kernel<true, HV, HW, W>;                                                                    \
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=           \
            hipSuccess) {                                                                                              \
            eps_set_error("eps_expand_fill: cannot reserve %zu bytes of LDS", lds);                                    \
            return EPS_ELAUNCH;                                                                                        \
        }                                                                                                              \
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(EX_THREADS), lds, s, rowptr, col, val, node_w,           \
                           (int32_t)n_nodes, (int32_t)v_lo, (int32_t)v_hi, col_order, wpt, counter, cand_count,        \
                           colptr, cand_u, cand_v, range_shift, tile_half, region_words, (int32_t)win_ids,             \
                           scored ? (uint2 *)((char *)workspace + 8) : (uint2 *)nullptr, per_block, cn, score, cut,    \
                           (unsigned int *)workspace);                                                                 \
    } while (0)
    if (windowed) {
        if (hv && hw) EX_LAUNCH(true, true, true);
        else if (hv) EX_LAUNCH(true, false, true);
        else if (hw) EX_LAUNCH(false, true, true);
        else EX_LAUNCH(false, false, true);
    } else {
        if (hv && hw) EX_LAUNCH(true, true, false);
        else if (hv) EX_LAUNCH(true, false, false);
        else if (hw) EX_LAUNCH(false, true, false);
        else EX_LAUNCH(false, false, false);
    }
#undef EX_LAUNCH
    EPS_CHECK_LAUNCH("eps_expand_fill");
    return EPS_OK;
}

// (one empty kernel per translation unit: launching it makes the HIP runtime load this unit's code object -- eps_warm_up)
__global__ void expand_score_warm_kernel() {}
extern "C" void eps_warm_expand_score(void *stream) { hipLaunchKernelGGL(expand_score_warm_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream); }
