// Threshold scan of the filter stage's candidate set, gfx950: every 2-hop non-edge whose heuristic score exceeds a
// bar, found WITHOUT materialising the candidate list.
//
// What it replaces.  filter.py:96-109 builds all 2-hop non-edges on the host, :113-142 score them batch by batch
// (adamic_utils.py:13-25, train_and_eval.py:195-216, models.py:536-542), :160-161 sort all E rows -- and rank.py:294
// then reads the first `num_sorted_edge` of them.  With `--keep_top K` only candidates above the K-th best score
// matter, so this kernel computes the score of EVERY candidate but reports only those above `threshold` (the host
// derives the bar from a column sample and verifies that at least K survive: edge-proposal-sets_amd/scan.py).
//
// Three things make it cheaper than eps_expand_fill, which it otherwise follows (propagation blocking of the
// 2-hop paths of a column over an LDS bitmap, order-independent 2^-40 fixed-point sums):
//   * SYMMETRY.  For a symmetric unit-valued adjacency the score of (u,v) equals that of (v,u), term by term.
//     Column v only expands the endpoints u < v: of row w = N(v)[k] only the first revpos[e] entries (the position
//     of v in row w, a per-graph table) are walked -- half the two-hop paths -- and the host mirrors the survivors.
//   * NOTHING PER CANDIDATE LEAVES THE CHIP.  No candidate list, no score array: a survivor's u is recovered from
//     the rank tables that are still in LDS.  The only bulk traffic left is the bucket records, now 4 bytes
//     (rank in tile | index k of w in N(v)); the fixed-point weight of (v,w) comes from an LDS table built once per
//     column instead of one float->fixed conversion per path.  Columns whose candidates fit one tile accumulate
//     straight into LDS and write no records at all.
//   * PACKED UNITS.  Row heads shorter than a wave-wide unit would leave most lanes idle (mean row head: ~200
//     entries), so the walk is over 64-entry units of a virtual concatenation of the rows, four units (possibly of
//     four different rows) per wave instruction, dealt round-robin over the 16 waves: a unit -> row map is built
//     per round of <= 2048 units by one block-wide max-scan.
// Scores are bit-identical to eps_expand_fill's (same fixed-point terms, same final rounding).
//
// The same template also writes candidate LISTS (MODE = FS_COUNT / FS_EMIT, entry points eps_expand_unit_count / _fill at the
// end of this file): every endpoint of a column (whole rows) or, HALF, the endpoints below it, with the score of every
// candidate instead of the survivors above a bar.
#include "eps_common.h"
#include <string.h>
#include <stdlib.h>

// Geometry (compile-time; the shipped values are the measured best on the ppa-sized graphs).  FS_UR must be 2 * FS_THREADS.
#ifndef FS_THREADS
#define FS_THREADS 1024
#endif
#ifndef FS_WG_PER_CU
#define FS_WG_PER_CU 1       // workgroups resident per CU: the LDS is split between them
#endif
#define FS_WAVES (FS_THREADS / 64)
#define FS_FIXED_SHIFT 40
#ifndef FS_RC
#define FS_RC 512            // rows (neighbours w of v) described per round
#endif
#define FS_UR (2 * FS_THREADS)   // 64-entry units per round (two list entries per thread)
#define FS_UPAD 384          // list entries past the last unit a prefetching wave may touch: they name the empty row
#ifndef FS_RANGES
#define FS_RANGES 512        // id ranges per column: path histogram and tile plan
#endif
#define FS_CHUNK 8192        // survivor slots reserved per global atomic
#define FS_SVCAP 384         // survivors of a column parked in LDS until its tiles are done (more: resolved on the spot)
#ifndef FS_MAX_TILE_BITS
#define FS_MAX_TILE_BITS 12  // candidate ranks per tile <= 4096 (8-byte accumulators in LDS)
#endif

typedef int v4i __attribute__((ext_vector_type(4)));

// LDS carve-up, in 32-bit words from the start of the dynamic region (every section starts 8-byte aligned)
struct fs_layout {
    int words;      // bitmap words, a multiple of 1024
    int tile;       // candidate ranks per tile
    int o_base32, o_pre8, o_acc, o_hist, o_tile_r0, o_tile_base, o_tile_cur, o_ustart, o_rbase, o_rlen,
        o_ulist, o_vwfix, o_sv, total_words;
};

__host__ __device__ static inline fs_layout fs_make_layout(int words, int tile_bits)
{
    fs_layout L;
    L.words = words;
    L.tile = 1 << tile_bits;
    int o = words;
    L.o_base32 = o;  o += words / 8;
    L.o_pre8 = o;    o += words / 4;
    L.o_acc = o;     o += 2 * L.tile;
    L.o_hist = o;    o += FS_RANGES;
    L.o_tile_r0 = o; o += FS_RANGES + 2;
    L.o_tile_base = o; o += FS_RANGES + 2;
    L.o_tile_cur = o;  o += FS_RANGES + 64;       // + one trash cursor per lane
    L.o_ustart = o;  o += FS_RC + 2;
    L.o_rbase = o;   o += FS_RC + 2;
    L.o_rlen = o;    o += FS_RC + 2;
    L.o_ulist = o;   o += (FS_UR + FS_UPAD) / 2;
    L.o_vwfix = o;   o += 2 * FS_RC;
    L.o_sv = o;      o += 2 * FS_SVCAP;
    L.total_words = o;
    return L;
}

struct fs_params {
    const int64_t *rowptr;
    const int32_t *col;
    const int32_t *revpos;
    const int64_t *fixw;
    const int32_t *columns;
    int32_t n_columns;
    int32_t n_nodes;
    uint32_t col_bytes;
    int32_t words;
    int32_t tile_bits;
    int32_t range_shift;
    uint32_t cap_records;     // bucket records per workgroup
    unsigned int *next_col;
    eps_survivors *out;
    uint32_t *scratch;        // cap_records (+ 64 trash) words per workgroup
    long long *gfix;          // max_degree weights per workgroup: the weight table of columns that take several rounds
    int32_t max_degree;
    const int32_t *splits;    // [n_win - 1][n_nodes]: entries of row w below id (k + 1) * win_ids (NULL when n_win == 1)
    int32_t win_ids;          // ids per window (a multiple of 32768; the LDS bitmap's span)
    int32_t n_win;
    // FS_EMIT / FS_COUNT (the candidate LIST of a block of columns, eps_expand_unit_*): columns are col_base + columns[t]
    // (columns == NULL: col_base + t), every endpoint counts (not only u < v), outputs go to colptr-addressed segments
    int32_t col_base;
    const int64_t *colptr;    // [n_columns + 1], by column - col_base: first output slot (an upper bound layout is allowed)
    int64_t *cand_count;      // [n_columns] or NULL: candidates found per column
    int32_t *cand_u, *cand_v; // ascending u inside a column; cand_v may be NULL
    float *out_score;         // NULL: the list only
    unsigned int *status;     // bit 1: a column outgrew its segment, bit 2: a sum left the fixed-point range
    int32_t no_pad;           // FS_EMIT on an upper-bound layout: leave the unused tail of a segment unwritten (the counts say where it starts)
};

#define FS_SCAN 0             // report the candidates above a bar (symmetric half scheme)
#define FS_EMIT 1             // write every candidate (and its score)
#define FS_COUNT 2            // count the candidates per column

// Wave-wide inclusive scans on the DPP path (row shifts inside the 16-lane rows, then the two row broadcasts): six
// VALU instructions, no LDS traffic.  Lanes a shift does not reach add 0 (values are non-negative for the max form).
__device__ __forceinline__ int fs_wave_incl_scan(int x, int)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);   // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);   // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1, 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2, 3
    return x;
}

__device__ __forceinline__ int fs_max(int a, int b) { return a > b ? a : b; }

__device__ __forceinline__ int fs_wave_incl_max(int x, int)
{
    x = fs_max(x, __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false));
    x = fs_max(x, __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false));
    x = fs_max(x, __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false));
    x = fs_max(x, __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false));
    x = fs_max(x, __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false));
    x = fs_max(x, __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false));
    return x;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory counter, which would
// end every global load a wave keeps in flight across the barrier (record and row prefetches); all hand-offs between the
// waves of this kernel go through LDS except one (bucket records: D1 -> D2), which keeps __syncthreads().
__device__ __forceinline__ void fs_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 16-byte store to a dword-aligned address (global memory takes unaligned vector stores)
typedef uint32_t fs_u4a4 __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ void fs_store4(uint32_t *dst, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    fs_u4a4 v = {a, b, c, d};
    *(fs_u4a4 *)dst = v;
}

// one quad of units in flight: the 16 bytes a lane loaded plus where they came from
struct fs_unit {
    v4i u4;
    int row;      // row of the round (index into the descriptor arrays; FS_RC = the empty dummy row)
    int nvalid;   // 0..4 entries of u4 that belong to the row head
};

// HALF (FS_EMIT / FS_COUNT): the list holds the endpoints BELOW the column only, like the scan -- for symmetric patterns
// whose consumer is symmetric in (u, v) too (the GNN filters: the decoder scores h_u * h_v), so that each unordered pair is
// listed -- and decoded -- once.
template <bool WINDOWED, int MODE, bool HALF = false>
__global__ __launch_bounds__(FS_THREADS) void filter_scan_kernel(fs_params p)
{
    constexpr bool FULL = MODE != FS_SCAN && !HALF;      // all endpoints of a column, not only those below it
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const fs_layout L = fs_make_layout(p.words, p.tile_bits);
    uint32_t *bm = lds;                                   // bit u: u (< v) is a two-hop endpoint of the column
    uint32_t *base32 = lds + L.o_base32;                  // rank of the first bit of every 8-word group
    uint8_t *pre8 = (uint8_t *)(lds + L.o_pre8);          // rank of a word's first bit within its group
    unsigned long long *acc = (unsigned long long *)(lds + L.o_acc);   // fixed-point sums of one tile; zero between uses
    uint32_t *hist = lds + L.o_hist;                      // paths per id range (pass A); zero between columns
    uint32_t *tile_r0 = lds + L.o_tile_r0;                // first candidate rank of a tile; [n_tiles] = column total
    uint32_t *tile_base = lds + L.o_tile_base;            // paths of the column before the tile; [n_tiles] = all
    uint32_t *tile_cur = lds + L.o_tile_cur;              // next free record of the tile's bucket (window-relative)
    int32_t *ustart = (int32_t *)(lds + L.o_ustart);      // first unit of a row in the round's unit numbering
    uint32_t *rbase = lds + L.o_rbase;                    // first entry of the row in col[]
    uint32_t *rlen = lds + L.o_rlen;                      // entries of the row below v
    uint16_t *ulist = (uint16_t *)(lds + L.o_ulist);      // unit -> row of the round, + 1
    long long *vwfix = (long long *)(lds + L.o_vwfix);    // fixed-point weight of (v, row)
    uint32_t *sv_rank = lds + L.o_sv;                     // parked survivors: candidate rank in the window ...
    float *sv_val = (float *)(lds + L.o_sv + FS_SVCAP);   // ... and score
    __shared__ int s_wtot[4 * FS_WAVES];
    __shared__ unsigned int s_tq;
    __shared__ int s_done, s_next_c, s_ntiles, s_thi;
    __shared__ unsigned int s_nsv;
    __shared__ unsigned int s_out_cur, s_out_end;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wib = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = lane >> 4, gl = lane & 15;
    const int words = p.words, TILE = L.tile, range_shift = p.range_shift;
    const uint32_t tile_mask = (uint32_t)TILE - 1u;
    const float thr = MODE == FS_SCAN ? p.out->threshold : 0.f;
    // the bar in the accumulators' domain: the smallest sum whose float32 score exceeds thr (the conversion is monotone),
    // so the per-candidate test is one 64-bit compare and only survivors are converted
    long long thr_fix = 0;
    if (MODE == FS_SCAN) {
        auto above = [&](long long a) { return (float)((double)a * (1.0 / (double)(1ll << FS_FIXED_SHIFT))) > thr; };
        if (!above(0x7fffffffffffffffll)) {
            thr_fix = 0x7fffffffffffffffll;               // +inf / NaN bar: nothing passes (sums never reach 2^63 - 1)
        } else {
            unsigned long long lo = 0ull, hi = 0xffffffffffffffffull;    // biased by 2^63: order-preserving
            while (lo < hi) {
                const unsigned long long mid = lo + ((hi - lo) >> 1);
                if (above((long long)(mid ^ 0x8000000000000000ull))) hi = mid; else lo = mid + 1;
            }
            thr_fix = (long long)(lo ^ 0x8000000000000000ull);
        }
    }
    const uint32_t out_cap = MODE == FS_SCAN ? p.out->capacity : 0u;
    int64_t *__restrict__ out_key = MODE == FS_SCAN ? p.out->key : nullptr;
    float *__restrict__ out_val = MODE == FS_SCAN ? p.out->val : nullptr;
    uint32_t *__restrict__ my_scratch = p.scratch + (size_t)blockIdx.x * ((size_t)p.cap_records + 64);   // + a trash line
    long long *__restrict__ my_gfix = p.gfix + (size_t)blockIdx.x * (size_t)p.max_degree;
    const __amdgpu_buffer_rsrc_t col_rs = __builtin_amdgcn_make_buffer_rsrc((void *)p.col, 0, p.col_bytes, 0x00020000);

    for (int i = tid; i < words; i += FS_THREADS) bm[i] = 0u;
    for (int i = tid; i < TILE; i += FS_THREADS) acc[i] = 0ull;
    if (tid < FS_RANGES) hist[tid] = 0u;
    if (tid == 0) {
        s_out_cur = 0u;
        s_out_end = 0u;
        s_nsv = 0u;
        ustart[FS_RC] = 0;       // the dummy row: units past the end of a round read nothing
        rbase[FS_RC] = p.col_bytes >> 2;      // past the end of col[]: the buffer load returns zeros, no traffic
        rlen[FS_RC] = 0u;
    }

    // Columns are handed out by a device counter.  A column starts with a chain of dependent round trips -- ticket -> column
    // id -> row bounds -> neighbour list -> row descriptors -> first entries -- during which the CU's only workgroup has
    // nothing else to do, so the first links are fetched AHEAD, in stages that each ride on a wait the current column has
    // anyway: while column n marks its endpoints the ticket of column n + 3, the id of column n + 2 and the row bounds of
    // column n + 1 are in flight (stage 1); they are picked up after the marking walk (stage 2).
    const uint32_t *__restrict__ rowptr_lo = (const uint32_t *)p.rowptr;     // nnz < 2^30: the low words suffice
    const unsigned int ncol = (unsigned int)p.n_columns;
    // (the first three tickets of a workgroup are static -- one column of each of the three heaviest grid-wide rounds --, the
    // counter hands out the rest)
    const unsigned int t_static = 3u * gridDim.x;
    auto col_of = [&](unsigned int t) -> int32_t {
        if (MODE == FS_SCAN) return p.columns[t];
        return p.col_base + (p.columns ? p.columns[t] : (int32_t)t);
    };
    int32_t v_cur = -1, v_nx = -1;
    uint32_t vb_cur = 0, t_nx2 = blockIdx.x + 2u * gridDim.x;
    int32_t dv_cur = 0;
    {
        const unsigned int t0 = blockIdx.x;
        if (t0 < ncol) {
            v_cur = col_of(t0);
            vb_cur = rowptr_lo[2 * (size_t)v_cur];
            dv_cur = (int32_t)(rowptr_lo[2 * (size_t)v_cur + 2] - vb_cur);
        }
        if (t0 + gridDim.x < ncol) v_nx = col_of(t0 + gridDim.x);
    }
    while (v_cur >= 0) {
        const int32_t v = v_cur;
        const uint32_t vb = vb_cur;
        const int32_t dv = dv_cur;
        const int32_t *__restrict__ vcol = p.col + vb;
        const int32_t *__restrict__ vrev = FULL ? nullptr : p.revpos + vb;
        // FULL: this column's output segment; candidates of the id windows before the current one
        const int64_t seg_base = MODE == FS_EMIT ? p.colptr[v - p.col_base] : 0;
        const int64_t seg_len = MODE == FS_EMIT ? p.colptr[v - p.col_base + 1] - seg_base : 0;
        int64_t col_off = 0;
        bool seg_overflow = false;
        int staged = 0;
        unsigned int pf_t = 0;
        int32_t pf_v = -1;
        uint32_t pf_b = 0, pf_e = 0;
        auto stage1 = [&]() {
            staged = 1;
            if (tid == 0) pf_t = t_static + atomicAdd(p.next_col, 1u);
            if (t_nx2 < ncol) pf_v = col_of(t_nx2);
            if (v_nx >= 0) {
                pf_b = rowptr_lo[2 * (size_t)v_nx];
                pf_e = rowptr_lo[2 * (size_t)v_nx + 2];
            }
        };
        int32_t v_nx2 = -1, dv_nx = 0;
        uint32_t vb_nx = 0;
        auto stage2 = [&]() {
            staged = 2;
            if (tid == 0) s_tq = pf_t;
            v_nx2 = __builtin_amdgcn_readfirstlane(pf_v);
            vb_nx = __builtin_amdgcn_readfirstlane(pf_b);
            dv_nx = (int32_t)(__builtin_amdgcn_readfirstlane(pf_e) - vb_nx);
        };
        if (dv != 0 && (FULL || v != 0)) {
        // The endpoints below v are taken in id WINDOWS of win_ids ids (one window when the whole id space fits the LDS):
        // window k sees of every row head only the entries inside it -- rows are ascending and splits[] holds, per node,
        // how many of its entries lie below each window boundary -- and runs the whole pipeline on ids relative to its start.
        // (WINDOWED == false: one window from id 0, everything below folds to constants)
        const int32_t id_end = FULL ? p.n_nodes : v;           // endpoints taken: ids [0, id_end)
        for (int32_t wk = 0, win_base_id = 0; win_base_id < id_end; ++wk, win_base_id += p.win_ids) {
        const int32_t win_lo = WINDOWED ? win_base_id : 0;
        const int32_t vlim = (WINDOWED && id_end - win_lo > p.win_ids) ? p.win_ids : id_end - win_lo;     // ids of this window, relative
        const uint32_t id_max = (uint32_t)vlim - 1u;
        const int32_t *__restrict__ split_lo = (WINDOWED && wk > 0) ? p.splits + (size_t)(wk - 1) * (size_t)p.n_nodes : nullptr;
        const int32_t *__restrict__ split_hi = (WINDOWED && wk + 1 < p.n_win) ? p.splits + (size_t)wk * (size_t)p.n_nodes : nullptr;
        const int words_v = (vlim + 31) >> 5;

        // ---- describe a round: rows j0.. of N(v) from chunk c0 of row j0, until FS_RC rows or FS_UR units ------------
        // Leaves the descriptors + the unit -> row list in LDS; -> units of the round; s_done / s_next_c = the cursor after it.
        auto build_round = [&](int j0, int c0) -> int {
            int nun = 0;
            uint32_t rb = 0, rl = 0;
            long long fx = 0;
            const bool row_ok = tid < FS_RC && j0 + tid < dv;
            if (row_ok) {
                const int32_t w = vcol[j0 + tid];
                uint32_t r0;
                int32_t hi, lo = 0;                            // the row head below v (FULL: the whole row) ...
                if (FULL) {
                    r0 = rowptr_lo[2 * (size_t)w];
                    hi = (int32_t)(rowptr_lo[2 * (size_t)w + 2] - r0);
                    if (p.fixw) fx = p.fixw[w];                // (NULL: the list / the counts only)
                } else if (MODE != FS_SCAN) {                  // HALF list
                    hi = vrev[j0 + tid];
                    r0 = rowptr_lo[2 * (size_t)w];
                    if (p.fixw) fx = p.fixw[w];
                } else {
                    hi = vrev[j0 + tid];
                    r0 = (uint32_t)p.rowptr[w];
                    fx = p.fixw[w];
                }
                if (split_lo) lo = split_lo[w];                // ... cut to the window
                if (split_hi) { const int32_t h = split_hi[w]; hi = hi < h ? hi : h; }
                rl = hi > lo ? (uint32_t)(hi - lo) : 0u;
                rb = r0 + (uint32_t)lo;
                nun = (int)((rl + 63u) >> 6);
                if (tid == 0) nun -= c0;
            }
            int incl = fs_wave_incl_scan(nun, lane);
            if (lane == 63) s_wtot[wib] = incl;
            ((uint32_t *)ulist)[tid] = 0u;
            if (tid < FS_UPAD / 2) ((uint32_t *)ulist)[FS_UR / 2 + tid] = (uint32_t)(FS_RC + 1) * 0x10001u;
            if (tid == 0) {
                s_done = 0;
                s_next_c = 0;
            }
            fs_barrier();
            int woff = 0, total = 0;
#pragma unroll
            for (int i = 0; i < FS_RC / 64; ++i) {
                const int t = s_wtot[i];
                if (i < wib) woff += t;
                total += t;
            }
            incl += woff;
            const int excl = incl - nun;
            // a column that takes more than one round keeps its weights in a global table (the LDS one holds one round)
            if (row_ok && (j0 > 0 || c0 > 0 || total > FS_UR || dv > FS_RC)) my_gfix[j0 + tid] = fx;
            if (tid < FS_RC) {
                ustart[tid] = excl - (tid == 0 ? c0 : 0);
                rbase[tid] = rb;
                rlen[tid] = rl;
                vwfix[tid] = fx;
                if (nun > 0 && excl < FS_UR) ulist[excl] = (uint16_t)(tid + 1);
            }
            // rows whose units all fit are done; the first row that does not fit is cut (continues in the next round)
            const unsigned long long dm = __ballot(row_ok && incl <= FS_UR);
            if (lane == 0 && dm) atomicAdd(&s_done, __popcll(dm));
            if (row_ok && incl > FS_UR && excl <= FS_UR) s_next_c = (FS_UR - excl) + (tid == 0 ? c0 : 0);
            fs_barrier();
            // unit -> row: the last row that starts at or before the unit (block-wide inclusive max-scan, 2 units per thread)
            {
                const uint32_t pr = ((uint32_t *)ulist)[tid];
                int a = (int)(pr & 0xffffu), b = (int)(pr >> 16);
                b = b > a ? b : a;
                const int inc = fs_wave_incl_max(b, lane);
                if (lane == 63) s_wtot[wib] = inc;
                int prev = __shfl_up(inc, 1);
                if (lane == 0) prev = 0;
                fs_barrier();
                int carry = 0;
#pragma unroll
                for (int i = 0; i < FS_WAVES; ++i) {
                    const int t = s_wtot[i];
                    if (i < wib) carry = carry > t ? carry : t;
                }
                prev = prev > carry ? prev : carry;
                a = a > prev ? a : prev;
                b = b > a ? b : a;
                const int n_units = total < FS_UR ? total : FS_UR;
                if (2 * tid >= n_units) a = FS_RC + 1;            // past the round's last unit: the empty row
                if (2 * tid + 1 >= n_units) b = FS_RC + 1;
                ((uint32_t *)ulist)[tid] = (uint32_t)a | ((uint32_t)b << 16);
            }
            fs_barrier();
            return total < FS_UR ? total : FS_UR;
        };

        // ---- walk the units of a round: four units per wave instruction, quads dealt round-robin over the waves ---------
        auto fetch = [&](int q) -> fs_unit {
            fs_unit f;
            const int s = q * 4 + grp;                 // < FS_UR + FS_UPAD: entries past the round name the empty row
            f.row = (int)ulist[s] - 1;
            const int off = (s - ustart[f.row]) * 64 + gl * 4;
            const int left = (int)rlen[f.row] - off;
            f.nvalid = left < 0 ? 0 : (left > 4 ? 4 : left);
            f.u4 = __builtin_amdgcn_raw_buffer_load_b128(col_rs, (int)((rbase[f.row] + (uint32_t)off) * 4u), 0, 0);
            return f;
        };
        // a ring of three quads in flight per wave; every ring slot is refilled by one unconditional load per trip
        auto walk = [&](int n_units, auto body) {
            const int nq = (n_units + 3) >> 2;
            fs_unit ring[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                ring[r] = fetch(wib + r * FS_WAVES);
                __builtin_amdgcn_sched_barrier(0);     // issue order == ring order, or the loop head waits for vmcnt(0)
            }
            int q = wib;
            for (; q + 2 * FS_WAVES < nq; q += 3 * FS_WAVES) {       // full trips only: the back edge always has 3 in flight
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    body(ring[r]);                                  // consume the slot, THEN refill it: no register copy,
                    ring[r] = fetch(q + (r + 3) * FS_WAVES);        // so the wait before the next body is vmcnt(2), not 0
                }
            }
            if (q < nq) body(ring[0]);
            if (q + FS_WAVES < nq) body(ring[1]);
        };

        // ---- A. mark every two-hop endpoint below v; count the paths per id range ---------------------------------------
        int j = 0, c = 0;
        bool single = false;
        int n_units = 0;
        for (;;) {
            n_units = build_round(j, c);
            const int nj = j + s_done, nc = s_next_c;
            single = j == 0 && c == 0 && nj >= dv;
            if (staged < 1) stage1();
            walk(n_units, [&](const fs_unit &f) {
                // no branches: an entry past the row head ORs / adds 0 (into whatever word its stale id names: ids < N)
                uint32_t rg[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    uint32_t u = (uint32_t)(f.u4[e] - win_lo);
                    if (WINDOWED) u = u < id_max ? u : id_max;         // (a stale id may lie outside the window: clamp)
                    const uint32_t on = e < f.nvalid ? 1u : 0u;
                    atomicOr(&bm[u >> 5], on << (u & 31));
                    rg[e] = u >> range_shift;
                }
                // a lane's entries ascend: when the first and the last valid one share an id range all of them do -- one add
                const uint32_t rlast = rg[f.nvalid > 0 ? f.nvalid - 1 : 0];
                if (rg[0] == rlast) {
                    atomicAdd(&hist[rg[0]], (uint32_t)f.nvalid);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (e < f.nvalid) atomicAdd(&hist[rg[e]], 1u);
                }
            });
            j = nj;
            c = nc;
            if (j >= dv) break;
            fs_barrier();       // the next round overwrites the descriptors this walk reads
        }
        if (staged < 2) stage2();
        fs_barrier();
        for (int k = tid; k < dv; k += FS_THREADS) {   // known edges out (the diagonal is not below v)
            const uint32_t u = (uint32_t)(vcol[k] - win_lo);
            if (u < (uint32_t)vlim) atomicAnd(&bm[u >> 5], ~(1u << (u & 31)));
        }
        if (FULL && tid == 0) {                          // the diagonal out (SCAN: it is not below v)
            const uint32_t u = (uint32_t)(v - win_lo);
            if (u < (uint32_t)vlim) atomicAnd(&bm[u >> 5], ~(1u << (u & 31)));
        }
        fs_barrier();

        // ---- B. rank tables: exclusive prefix of the per-word popcounts over the words below v -------------------------
        // Thread t owns the 8-word groups t, t + 1024, ... (two 16-byte reads per group, consecutive threads on consecutive
        // groups); one wave scan per trip orders the groups, the 16 x trips wave totals are scanned once more.
        const int n_groups = (words_v + 7) >> 3;
        const int trips = (n_groups + FS_THREADS - 1) / FS_THREADS;   // <= 4 for ids < 2^20 at 1024 threads
        int gcnt[4];
        uint32_t gpre[4][2];                                          // the group's 8 byte ranks, packed
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            gcnt[i] = 0;
            gpre[i][0] = gpre[i][1] = 0u;
            const int gi = tid + i * FS_THREADS;
            if (i < trips && gi < n_groups) {
                const uint4 a = *(const uint4 *)(bm + gi * 8), b = *(const uint4 *)(bm + gi * 8 + 4);
                const int c0 = __popc(a.x), c1 = c0 + __popc(a.y), c2 = c1 + __popc(a.z), c3 = c2 + __popc(a.w);
                const int c4 = c3 + __popc(b.x), c5 = c4 + __popc(b.y), c6 = c5 + __popc(b.z);
                gcnt[i] = c6 + __popc(b.w);
                gpre[i][0] = (uint32_t)c0 << 8 | (uint32_t)c1 << 16 | (uint32_t)c2 << 24;
                gpre[i][1] = (uint32_t)c3 | (uint32_t)c4 << 8 | (uint32_t)c5 << 16 | (uint32_t)c6 << 24;
            }
        }
        int ginc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ginc[i] = 0;
            if (i < trips) {
                ginc[i] = fs_wave_incl_scan(gcnt[i], lane);
                if (lane == 63) s_wtot[i * FS_WAVES + wib] = ginc[i];
            }
        }
        fs_barrier();
        int total;
        {
            // exclusive prefix over the (trip, wave) totals, in every wave: lane l holds total l
            const int tv = lane < trips * FS_WAVES ? s_wtot[lane] : 0;
            const int tinc = fs_wave_incl_scan(tv, lane);
            total = __builtin_amdgcn_readlane(tinc, 63);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i >= trips) break;
                const int idx = i * FS_WAVES + wib;
                const int before = __builtin_amdgcn_readlane(tinc, idx) - __builtin_amdgcn_readlane(tv, idx);
                const int gi = tid + i * FS_THREADS;
                if (gi < n_groups) {
                    base32[gi] = (uint32_t)(before + ginc[i] - gcnt[i]);
                    *(uint2 *)(pre8 + gi * 8) = make_uint2(gpre[i][0], gpre[i][1]);
                }
            }
        }
        const int n_ranges = ((vlim - 1) >> range_shift) + 1;
        if (MODE == FS_SCAN && tid == 0 && total) atomicAdd(&p.out->n_candidates, (unsigned long long)total);
        bool listed = MODE == FS_COUNT;                   // FULL: nothing (more) to do for this window after the list
        const int64_t obase = seg_base + col_off;         // EMIT: first output slot of this window's candidates
        if (MODE == FS_EMIT && total) {
            if (col_off + total > seg_len) {             // the caller's segment does not hold the column: flag it, stop
                if (tid == 0) atomicOr(p.status, 2u);
                seg_overflow = true;
                listed = true;
            } else {
                // the list: thread per bitmap word (neighbouring lanes -> neighbouring ranks), ascending u
                fs_barrier();                             // rank tables complete
#ifdef FS_ABL_NOULIST      // (timing-only ablations, tools/r06_full_list_ablate.sh: the outputs are wrong)
                if (p.no_pad == 7)
#endif
                for (int wi = tid; wi < words_v; wi += FS_THREADS) {
                    uint32_t bits = bm[wi];
                    if (bits) {
                        uint32_t run = base32[wi >> 3] + pre8[wi];
                        do {
#if defined(FS_ABL_ULIST_NOSTORE)
                            asm volatile("" ::"v"(win_lo + wi * 32 + __builtin_ctz(bits)), "v"(run++));
#elif defined(FS_ABL_ULIST_SMALL)
                            p.cand_u[(obase + run++) & 0x3FFFF] = win_lo + wi * 32 + __builtin_ctz(bits);
#else
                            p.cand_u[obase + run++] = win_lo + wi * 32 + __builtin_ctz(bits);
#endif
                            bits &= bits - 1;
                        } while (bits);
                    }
                }
                if (p.cand_v)
                    for (int i = tid; i < total; i += FS_THREADS) p.cand_v[obase + i] = v;
                listed = p.out_score == nullptr;
            }
        }
        col_off += total;
        if (total == 0 || listed) {          // nothing to score (SCAN: every endpoint is a neighbour of v)
            if (MODE == FS_EMIT && total) fs_barrier();      // (the list pass read the bitmap)
            if (tid < n_ranges) hist[tid] = 0u;
            for (int i = tid; i < words_v; i += FS_THREADS) bm[i] = 0u;
            if (!WINDOWED || seg_overflow) break;
            fs_barrier();              // (the next window marks into the words other threads just cleared)
            continue;
        }
        fs_barrier();

        // ---- plan: tiles of TILE consecutive candidate ranks (the last one partial); bucket room from the path histogram ----
        // A path's tile is its rank >> tile_bits.  The paths per id range are known (pass A), the paths per TILE are not -- a range
        // whose ranks straddle a tile boundary (at most one boundary: a range holds at most half a tile of ids) splits unknowably --
        // so a straddling range reserves its paths in BOTH buckets: range i owns [pbase, pbase + paths) for its first tile and, when
        // it straddles, [pbase + paths, pbase + 2 paths) for the next.  The bucket of tile t starts where the range that holds rank
        // t * TILE puts it: at its own start when that rank is its first, at its second copy otherwise.
        {
            const bool in = tid < n_ranges;
            const uint32_t rs = in ? base32[tid << (range_shift - 8)] : 0u;   // rank at the start of the range ...
            const uint32_t re = !in ? 0u : tid + 1 < n_ranges ? base32[(tid + 1) << (range_shift - 8)] : (uint32_t)total;   // ... and past its end
            const uint32_t paths = in ? (hist[tid] + 3u) & ~3u : 0u;   // buckets start on 16-byte lines (D2 reads 4 records per load)
            if (in) hist[tid] = 0u;
            const uint32_t tile = rs >> p.tile_bits;
            const bool straddle = re > rs && ((re - 1u) >> p.tile_bits) != tile;
            const uint32_t room = straddle ? 2u * paths : paths;
            const int pin = fs_wave_incl_scan((int)room, lane);
            if (lane == 63) s_wtot[wib] = pin;
            fs_barrier();
            uint32_t pbase = 0, ptotal = 0;
#pragma unroll
            for (int i = 0; i < FS_WAVES; ++i) {
                const uint32_t t = (uint32_t)s_wtot[i];
                if (i < wib) pbase += t;
                ptotal += t;
            }
            pbase += (uint32_t)pin - room;
            if (re > rs && (rs & tile_mask) == 0u) tile_base[tile] = pbase;
            if (straddle) tile_base[tile + 1] = pbase + paths;
            const int nt = (total + TILE - 1) >> p.tile_bits;
            if (tid <= nt) tile_r0[tid] = (uint32_t)(tid << p.tile_bits) < (uint32_t)total ? (uint32_t)(tid << p.tile_bits) : (uint32_t)total;
            if (tid == 0) {
                s_ntiles = nt;
                tile_base[nt] = ptotal;
            }
            fs_barrier();
        }
        const int n_tiles = s_ntiles;
        uint32_t *ginfo = base32;

        // ---- D. score: tiles are taken in windows; a window of one tile accumulates in LDS directly, a wider one bins ------
        //      4-byte records (rank in tile | k << tile_bits) per tile in this workgroup's scratch and sums tile by tile.
        // A survivor's u comes from its rank: last 8-word group whose first rank is <= r, then the word, then the bit -- a
        // chain of ~25 dependent LDS reads.  Done on the spot it sits on the critical path of its tile (one lane works, fifteen
        // waves wait at the barrier), so survivors are parked in LDS and resolved together when the column's tiles are done.
        auto resolve = [&](uint32_t r, float sc) {
            auto first_rank = [&](int gi) { return ginfo[gi]; };
            int glo = 0, ghi = n_groups;                     // groups [glo, ghi): invariant first_rank(glo) <= r
            while (ghi - glo > 1) {
                const int mid = (glo + ghi) >> 1;
                if (first_rank(mid) <= r) glo = mid; else ghi = mid;
            }
            const uint32_t rg = r - first_rank(glo);
            int wi = glo * 8, wj = 7;
            while (wj > 0 && (uint32_t)pre8[wi + wj] > rg) --wj;
            wi += wj;
            uint32_t bits = bm[wi];
            for (uint32_t sidx = rg - pre8[wi]; sidx > 0; --sidx) bits &= bits - 1;
            const uint32_t u = (uint32_t)wi * 32u + (uint32_t)__builtin_ctz(bits);
            const uint32_t pos = atomicAdd(&s_out_cur, 1u);
            if (pos < out_cap) {
                out_key[pos] = ((int64_t)v << 32) | (int64_t)(u + (uint32_t)win_lo);
                out_val[pos] = sc;
            }
        };
        auto scan_tile = [&](int t) {     // acc -> survivors; leaves acc zero.  Caller: barrier before (sums complete).
            const uint32_t r0 = tile_r0[t], nslots = tile_r0[t + 1] - r0;
            auto survivor = [&](uint32_t i, long long a) {
                const float sc = (float)((double)a * (1.0 / (double)(1ll << FS_FIXED_SHIFT)));
                const uint32_t q = atomicAdd(&s_nsv, 1u);
                if (q < FS_SVCAP) {
                    sv_rank[q] = r0 + i;
                    sv_val[q] = sc;
                } else {
                    resolve(r0 + i, sc);                          // the parking lot is full (a low bar): on the spot
                }
            };
            // two sums per 16-byte LDS access (a slot past the tile's last is zero and is not a candidate)
            for (uint32_t i = 2 * tid; i < nslots; i += 2 * FS_THREADS) {
                const uint4 raw = *(const uint4 *)(acc + i);
                *(uint4 *)(acc + i) = make_uint4(0u, 0u, 0u, 0u);
                const long long a0 = (long long)(((unsigned long long)raw.y << 32) | raw.x);
                const long long a1 = (long long)(((unsigned long long)raw.w << 32) | raw.z);
                if (MODE == FS_EMIT) {               // every candidate's score, in candidate order (the ranks ARE the order)
                    // the heuristics' terms are non-negative: a negative sum is one that wrapped past 2^23 (backstop; the host
                    // bounds the graph's scores before it takes this path -- candidates.fused_scores_fit)
                    if ((a0 | a1) < 0) atomicOr(p.status, 4u);
                    float *o = p.out_score + obase + r0 + i;
#ifdef FS_ABL_NOSCORESTORE
                    if (a0 == 0x7fffffffffffffffll)
#endif
                    {
                    o[0] = (float)((double)a0 * (1.0 / (double)(1ll << FS_FIXED_SHIFT)));
                    if (i + 1 < nslots) o[1] = (float)((double)a1 * (1.0 / (double)(1ll << FS_FIXED_SHIFT)));
                    }
                } else {
                    if (a0 >= thr_fix) survivor(i, a0);
                    if (a1 >= thr_fix && i + 1 < nslots) survivor(i + 1, a1);
                }
            }
        };
        auto reserve_out = [&](int t) {   // thread 0, before the barrier that precedes scan_tile(t)
            if (MODE != FS_SCAN) return;
            const uint32_t nslots = tile_r0[t + 1] - tile_r0[t] + FS_SVCAP;      // this tile's + everything parked
            if (s_out_end - s_out_cur < nslots) {
                // (64-bit counter: a list that overflows can never wrap back under the capacity; chunks past it are dropped)
                const unsigned long long b64 = atomicAdd(&p.out->count, (unsigned long long)FS_CHUNK);
                const uint32_t b = b64 < (unsigned long long)out_cap ? (uint32_t)b64 : out_cap;
                s_out_cur = b;
                s_out_end = b + FS_CHUNK;
            }
        };

        int t_lo = 0;
        while (t_lo < n_tiles) {
            if (tid == 0) {
                int t_hi = t_lo + 1;
                while (t_hi < n_tiles && tile_base[t_hi + 1] - tile_base[t_lo] <= p.cap_records) ++t_hi;
                s_thi = t_hi;
            }
            fs_barrier();
            const int t_hi = s_thi;
            const bool direct = t_hi - t_lo == 1;
            const uint32_t win_base = tile_base[t_lo];
            const uint32_t span = (uint32_t)(t_hi - t_lo);
            if (!direct && tid >= t_lo && tid < t_hi) tile_cur[tid] = tile_base[tid] - win_base;
            // walk the paths again
            j = 0;
            c = 0;
            for (;;) {
                if (!single) n_units = build_round(j, c);
                else fs_barrier();
                const int nj = single ? dv : j + s_done, nc = single ? 0 : s_next_c;
                const int j_round = j;
                // per entry: is it a candidate of the window, and its rank inside its tile.  No branch per entry: an entry
                // past the row head goes through the look-ups on its stale id (< N) and is masked out of `cand`.
                auto classify = [&](const fs_unit &f, uint32_t (&rank)[4], uint32_t (&tl)[4], uint32_t (&cand)[4]) -> uint32_t {
                    uint32_t word[4], gi[4], f_u[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        uint32_t u = (uint32_t)(f.u4[e] - win_lo);
                        if (WINDOWED) u = u < id_max ? u : id_max;
                        f_u[e] = u;
                        word[e] = bm[u >> 5];
                        gi[e] = ginfo[u >> 8];
                        rank[e] = pre8[u >> 5];
                    }
                    uint32_t ncand = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const uint32_t b = f_u[e] & 31u;
                        rank[e] += gi[e] + __popc(word[e] & ((1u << b) - 1u));      // rank in the window's list ...
                        tl[e] = (rank[e] >> p.tile_bits) - (uint32_t)t_lo;           // ... its tile, relative to the tile window ...
                        rank[e] &= tile_mask;                                         // ... and the rank inside the tile
                        cand[e] = (e < f.nvalid && tl[e] < span) ? (word[e] >> b) & 1u : 0u;
                        ncand += cand[e];
                    }
                    return ncand;
                };
                if (direct) {
                    walk(n_units, [&](const fs_unit &f) {
                        uint32_t rank[4], tl[4], cand[4];
                        classify(f, rank, tl, cand);
                        const unsigned long long fx = (unsigned long long)vwfix[f.row];
#pragma unroll
                        for (int e = 0; e < 4; ++e) atomicAdd(&acc[rank[e] & tile_mask], cand[e] ? fx : 0ull);
                    });
                } else {
                    walk(n_units, [&](const fs_unit &f) {
                        uint32_t rank[4], tl[4], cand[4];
                        const uint32_t ncand = classify(f, rank, tl, cand);
                        const uint32_t krec = (uint32_t)(j_round + f.row) << p.tile_bits;
                        // entries of a lane are ascending: when its first and last candidate share a tile, all of them do --
                        // one cursor bump for the lane (a lane without candidates bumps its own trash cursor by 0)
                        const uint32_t tfirst = cand[0] ? tl[0] : cand[1] ? tl[1] : cand[2] ? tl[2] : tl[3];
                        const uint32_t tlast = cand[3] ? tl[3] : cand[2] ? tl[2] : cand[1] ? tl[1] : tl[0];
                        uint32_t pos[4];
                        if (ncand == 4u && tfirst == tlast) {
                            // the common lane: four candidates of one tile -> four consecutive records, ONE 16-byte store
                            // (dword-aligned; four scattered 4-byte stores cost four write requests per 64-byte chunk)
                            const uint32_t p0 = atomicAdd(&tile_cur[(uint32_t)t_lo + tfirst], 4u);
#ifdef FS_ABL_NORECSTORE
                            if (p0 == 0xdeadbeefu)
#endif
                            fs_store4(my_scratch + p0, rank[0] | krec, rank[1] | krec, rank[2] | krec, rank[3] | krec);
                        } else {
                        if (tfirst == tlast || ncand == 0) {
                            pos[0] = atomicAdd(&tile_cur[ncand ? (uint32_t)t_lo + tfirst : (uint32_t)(FS_RANGES + lane)], ncand);
                            pos[1] = pos[0] + cand[0];
                            pos[2] = pos[1] + cand[1];
                            pos[3] = pos[2] + cand[2];
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                pos[e] = cand[e] ? atomicAdd(&tile_cur[t_lo + tl[e]], 1u) : 0u;
                        }
                        // one store per entry, no branch: what is not a candidate of the window lands in the trash line
#pragma unroll
                        for (int e = 0; e < 4; ++e)
#ifdef FS_ABL_NORECSTORE
                            if (pos[e] == 0xdeadbeefu)
#endif
                            my_scratch[cand[e] ? pos[e] : p.cap_records + lane] = rank[e] | krec;
                        }
                    });
                }
                j = nj;
                c = nc;
                if (j >= dv) break;
                fs_barrier();
            }
            if (direct) {
                if (tid == 0) reserve_out(t_lo);
                fs_barrier();
                scan_tile(t_lo);
            } else {
                // per tile: records -> sums in LDS -> scan.  The first four records per thread of the NEXT tile are
                // requested before the scan of the current one, so their latency hides behind it.
                uint4 rec;
                uint32_t b0 = 0, n = 0;
                auto request = [&](int t) {    // thread t: records 4t .. 4t+3 of the tile's bucket (16-byte aligned)
                    b0 = tile_base[t] - win_base;
                    n = tile_cur[t] - b0;
                    rec = *(const uint4 *)(my_scratch + b0 + 4 * tid);
                };
                auto add = [&](uint32_t r, bool live) {
                    if (!live) return;                       // (mostly whole waves: the tail of a bucket)
#ifdef FS_ABL_NOACC
                    if (r != 0xdeadbeefu) return;
#endif
                    const uint32_t k = r >> p.tile_bits;
                    const long long fx = single ? vwfix[k] : my_gfix[k];
                    atomicAdd(&acc[r & tile_mask], (unsigned long long)fx);
                };
                auto add4 = [&](const uint4 &r, uint32_t first, uint32_t cn) {
                    add(r.x, first < cn);
                    add(r.y, first + 1 < cn);
                    add(r.z, first + 2 < cn);
                    add(r.w, first + 3 < cn);
                };
                __syncthreads();                   // records visible (global stores drained), cursors final
                request(t_lo);
                for (int t = t_lo; t < t_hi; ++t) {
                    const uint32_t cb0 = b0, cn = n;
                    add4(rec, 4 * tid, cn);
                    for (uint32_t i0 = 4 * (tid + FS_THREADS); i0 < cn; i0 += 4 * FS_THREADS)
                        add4(*(const uint4 *)(my_scratch + cb0 + i0), i0, cn);
                    if (t + 1 < t_hi) request(t + 1);
                    if (tid == 0) reserve_out(t);
                    fs_barrier();               // sums complete
                    scan_tile(t);
                    fs_barrier();               // accumulators zero again
                }
            }
            fs_barrier();           // scans done: the parked survivors are all in
            {
                const uint32_t nsv = s_nsv < FS_SVCAP ? s_nsv : FS_SVCAP;
                if (nsv) {             // (uniform) resolve them, one per thread; room was reserved with the tiles
                    if ((uint32_t)tid < nsv) resolve(sv_rank[tid], sv_val[tid]);
                    fs_barrier();
                    if (tid == 0) s_nsv = 0u;
                    fs_barrier();
                }
            }
            t_lo = t_hi;
        }
        for (int i = tid; i < words_v; i += FS_THREADS) bm[i] = 0u;
        if (!WINDOWED) break;
        fs_barrier();                  // the next window marks into the words other threads just cleared
        }
        }
        if (MODE != FS_SCAN) {
            if (p.cand_count && tid == 0 && !seg_overflow) p.cand_count[v - p.col_base] = col_off;
            if (MODE == FS_EMIT && !seg_overflow && !p.no_pad)
                for (int64_t i = col_off + tid; i < seg_len; i += FS_THREADS) {   // padding of an upper-bound segment
                    p.cand_u[seg_base + i] = -1;
                    if (p.cand_v) p.cand_v[seg_base + i] = v;
                    if (p.out_score) p.out_score[seg_base + i] = -__builtin_inff();
                }
        }
        if (staged == 0) {             // (a column without work: the stages still have to run)
            stage1();
            fs_barrier();              // every thread has read the previous ticket
            stage2();
        }
        fs_barrier();                  // thread 0's ticket is in LDS; the column's LDS state is clean
        t_nx2 = __builtin_amdgcn_readfirstlane(s_tq);
        v_cur = v_nx;
        vb_cur = vb_nx;
        dv_cur = dv_nx;
        v_nx = v_nx2;
    }
}

// ---- per-graph tables -----------------------------------------------------------------------------------------------
// revpos[e], e an entry of row v with w = col[e]: the number of entries of row w that are below v.  For a symmetric
// adjacency that is the position of v in row w.  One wave per row v; a lower-bound search per entry.
__global__ void reverse_positions_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                         int64_t n_nodes, int32_t *__restrict__ revpos, int64_t *__restrict__ half_paths,
                                         unsigned int *__restrict__ asymmetric)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t v = wave; v < n_nodes; v += n_waves) {
        const int64_t b = rowptr[v], e = rowptr[v + 1];
        long long below = 0;                 // the column's two-hop half paths: sum of its entries' row heads
        bool odd = false;                    // an entry (v, w) without its mirror (w, v)
        for (int64_t i = b + lane; i < e; i += 64) {
            const int32_t w = col[i];
            int64_t lo = rowptr[w], hi = rowptr[w + 1];
            const int64_t wb = lo, we = hi;
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (col[mid] < (int32_t)v) lo = mid + 1; else hi = mid;
            }
            revpos[i] = (int32_t)(lo - wb);
            below += lo - wb;
            odd |= lo >= we || col[lo] != (int32_t)v;
        }
        if (half_paths) {
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) below += __shfl_xor(below, d);
            if (lane == 0) half_paths[v] = below;
        }
        if (asymmetric && __ballot(odd) && lane == 0) atomicOr(asymmetric, 1u);
    }
}

// splits[k][w] = number of entries of row w below id (k + 1) * win_ids, k = 0 .. n_win - 2: where the id windows cut the row
__global__ void row_window_splits_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                         int64_t n_nodes, int64_t win_ids, int32_t n_cuts, int32_t *__restrict__ splits)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_nodes * n_cuts; i += stride) {
        const int64_t w = i % n_nodes, k = i / n_nodes;
        const int64_t bound = (k + 1) * win_ids;
        int64_t lo = rowptr[w], hi = rowptr[w + 1];
        const int64_t wb = lo;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (col[mid] < bound) lo = mid + 1; else hi = mid;
        }
        splits[i] = (int32_t)(lo - wb);
    }
}

__global__ void fixed_weights_kernel(const float *__restrict__ w, int64_t n, int64_t *__restrict__ fixw)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        fixw[i] = __double2ll_rn((double)w[i] * (double)(1ll << FS_FIXED_SHIFT));
}

extern "C" int eps_reverse_positions(const int64_t *rowptr, const int32_t *col, int64_t n_nodes, int32_t *revpos,
                                     int64_t *half_paths_or_null, uint32_t *asymmetric_or_null, void *stream)
{
    EPS_REQUIRE(n_nodes >= 0, "eps_reverse_positions: negative size");
    hipStream_t s = (hipStream_t)stream;
    if (asymmetric_or_null && hipMemsetAsync(asymmetric_or_null, 0, sizeof(uint32_t), s) != hipSuccess) {
        eps_set_error("eps_reverse_positions: cannot clear the flag");
        return EPS_ELAUNCH;
    }
    if (n_nodes == 0) return EPS_OK;
    EPS_REQUIRE(rowptr && col && revpos, "eps_reverse_positions: null pointer");
    int64_t blocks = (n_nodes + 3) / 4;
    const int64_t cap = (int64_t)eps_num_cus() * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(reverse_positions_kernel, dim3((unsigned)blocks), dim3(256), 0, s, rowptr, col, n_nodes, revpos,
                       half_paths_or_null, asymmetric_or_null);
    EPS_CHECK_LAUNCH("eps_reverse_positions");
    return EPS_OK;
}

extern "C" int eps_fixed_weights(const float *node_w, int64_t n, int64_t *fixw, void *stream)
{
    EPS_REQUIRE(n >= 0, "eps_fixed_weights: negative size");
    if (n == 0) return EPS_OK;
    EPS_REQUIRE(node_w && fixw, "eps_fixed_weights: null pointer");
    int64_t blocks = (n + 255) / 256;
    const int64_t cap = (int64_t)eps_num_cus() * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(fixed_weights_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, node_w, n, fixw);
    EPS_CHECK_LAUNCH("eps_fixed_weights");
    return EPS_OK;
}

// ---- launch ---------------------------------------------------------------------------------------------------------
#define FS_LDS_LIMIT (160 * 1024 / FS_WG_PER_CU - 512)     // dynamic + the few static words

// Launch geometry for an id space of n_nodes: the fewest id windows (of a multiple of 32768 ids) whose bitmap, rank tables
// and a tile of 2^FS_MAX_TILE_BITS accumulators fit the workgroup's share of the LDS; if even windows of 32768 ids do not
// leave room for such a tile, the tile shrinks.
struct fs_geometry {
    int n_win, words, tile_bits, range_shift;
    int64_t win_ids;
};

static int fs_range_shift(int64_t ids, int tile_bits)
{
    int s = 8;
    while (((ids - 1) >> s) + 1 > FS_RANGES) ++s;
    return s <= tile_bits - 1 ? s : -1;       // a range may hold at most half a tile
}

static bool fs_pick_geometry(int64_t n_nodes, fs_geometry *g)
{
    // (EPS_FS_MIN_WIN: geometry experiments -- more, narrower id windows than the LDS asks for)
    static const int64_t min_win = [] { const char *e = getenv("EPS_FS_MIN_WIN"); return e && atoi(e) > 0 ? (int64_t)atoi(e) : (int64_t)1; }();
    for (int tile_bits = FS_MAX_TILE_BITS; tile_bits >= 9; --tile_bits)
        for (int64_t n_win = min_win; n_win <= 4096; ++n_win) {
            const int64_t words = (((n_nodes + n_win - 1) / n_win + 31) / 32 + 1023) / 1024 * 1024;
            const int64_t win_ids = words * 32;
            if ((n_nodes + win_ids - 1) / win_ids != n_win) continue;        // rounding made a window superfluous
            const int rs = fs_range_shift(win_ids < n_nodes ? win_ids : n_nodes, tile_bits);
            // (four group trips per thread at most; the ranks of a window are packed in 20 bits; a window's tiles -- one per
            //  2^tile_bits ranks -- fit the tile tables)
            if ((size_t)fs_make_layout((int)words, tile_bits).total_words * 4 <= FS_LDS_LIMIT && rs >= 8 &&
                words / 8 <= 4 * FS_THREADS && win_ids <= (1 << 20) && (win_ids >> tile_bits) <= FS_RANGES) {
                g->n_win = (int)n_win;
                g->words = (int)words;
                g->tile_bits = tile_bits;
                g->range_shift = rs;
                g->win_ids = win_ids;
                return true;
            }
            if (words == 1024) break;
        }
    return false;
}

// ids per window / number of windows eps_filter_scan uses for an id space of n_nodes (the caller builds the split table for
// exactly these); returns EPS_EINVAL if no geometry fits.
extern "C" int eps_filter_scan_windows(int64_t n_nodes, int64_t *win_ids, int64_t *n_win)
{
    fs_geometry g;
    EPS_REQUIRE(n_nodes > 0 && n_nodes < (1ll << 31) && fs_pick_geometry(n_nodes, &g),
                "eps_filter_scan_windows: no launch geometry for %lld nodes", (long long)n_nodes);
    if (win_ids) *win_ids = g.win_ids;
    if (n_win) *n_win = g.n_win;
    return EPS_OK;
}

extern "C" int eps_row_window_splits(const int64_t *rowptr, const int32_t *col, int64_t n_nodes, int64_t win_ids,
                                     int64_t n_win, int32_t *splits, void *stream)
{
    EPS_REQUIRE(n_nodes >= 0 && win_ids > 0 && n_win >= 1, "eps_row_window_splits: bad size");
    if (n_nodes == 0 || n_win == 1) return EPS_OK;
    EPS_REQUIRE(rowptr && col && splits, "eps_row_window_splits: null pointer");
    int64_t blocks = (n_nodes * (n_win - 1) + 255) / 256;
    const int64_t cap = (int64_t)eps_num_cus() * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(row_window_splits_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rowptr, col,
                       n_nodes, win_ids, (int32_t)(n_win - 1), splits);
    EPS_CHECK_LAUNCH("eps_row_window_splits");
    return EPS_OK;
}

extern "C" int64_t eps_filter_scan_max_nodes(void) { return (1ll << 31) - 1; }

#define FS_DEFAULT_RECORDS (1u << 20)         // bucket records per workgroup: 4 MiB each, 1 GiB on 256 CUs

// bucket records (4 MiB + a trash line per workgroup) followed by the per-workgroup weight tables of multi-round columns
extern "C" int64_t eps_filter_scan_workspace_bytes(int64_t max_degree)
{
    if (max_degree < 0) return 0;
    return (int64_t)eps_num_cus() * FS_WG_PER_CU * (((int64_t)FS_DEFAULT_RECORDS + 64) * 4 + max_degree * 8);
}

// one launch of filter_scan_kernel<WINDOWED, MODE>: geometry, counter, LDS size (the caller filled the mode's own fields of p)
static int fs_launch(const char *who, fs_params p, int mode, const int32_t *splits, int64_t n_nodes, int64_t nnz,
                     int64_t max_degree, int64_t n_columns, void *workspace, int64_t workspace_bytes, void *stream,
                     bool half = false)
{
    EPS_REQUIRE(nnz < (1ll << 30), "%s: col[] is addressed with 32-bit byte offsets (nnz < 2^30)", who);
    EPS_REQUIRE(n_columns < (1ll << 31), "%s: too many columns", who);
    fs_geometry geo;
    EPS_REQUIRE(n_nodes < (1ll << 31) && fs_pick_geometry(n_nodes, &geo), "%s: no launch geometry for %lld nodes", who,
                (long long)n_nodes);
    EPS_REQUIRE(geo.n_win == 1 || splits, "%s: %d id windows need the row split table (eps_row_window_splits)", who, geo.n_win);
    int64_t blocks = (int64_t)eps_num_cus() * FS_WG_PER_CU;
    if (blocks > n_columns) blocks = n_columns;
    EPS_REQUIRE(max_degree >= 0 && max_degree <= n_nodes, "%s: bad max_degree", who);
    const int64_t n_wg = (int64_t)eps_num_cus() * FS_WG_PER_CU;
    EPS_REQUIRE(workspace && ((uintptr_t)workspace & 15) == 0 && workspace_bytes >= eps_filter_scan_workspace_bytes(max_degree),
                "%s: needs a 16-byte aligned workspace of eps_filter_scan_workspace_bytes(max_degree) bytes", who);
    hipStream_t s = (hipStream_t)stream;
    unsigned int *counter = nullptr;
    int rc = eps_take_counter(&counter, s, who);
    if (rc) return rc;
    p.n_columns = (int32_t)n_columns;
    p.n_nodes = (int32_t)n_nodes;
    p.col_bytes = (uint32_t)(nnz * 4);
    p.words = geo.words;
    p.splits = geo.n_win > 1 ? splits : nullptr;
    p.win_ids = (int32_t)(geo.win_ids < (1ll << 30) ? geo.win_ids : (1ll << 30));
    p.n_win = geo.n_win;
    p.tile_bits = geo.tile_bits;
    p.range_shift = geo.range_shift;
    p.cap_records = (uint32_t)FS_DEFAULT_RECORDS;
    p.next_col = counter;
    p.scratch = (uint32_t *)workspace;
    p.gfix = (long long *)((char *)workspace + n_wg * ((int64_t)FS_DEFAULT_RECORDS + 64) * 4);
    p.max_degree = (int32_t)max_degree;
    const size_t lds = (size_t)fs_make_layout(p.words, geo.tile_bits).total_words * 4;
    const bool win = geo.n_win > 1;
    void (*kern)(fs_params) =
        mode == FS_SCAN ? (win ? filter_scan_kernel<true, FS_SCAN> : filter_scan_kernel<false, FS_SCAN>)
        : mode == FS_EMIT
            ? (half ? (win ? filter_scan_kernel<true, FS_EMIT, true> : filter_scan_kernel<false, FS_EMIT, true>)
                    : (win ? filter_scan_kernel<true, FS_EMIT> : filter_scan_kernel<false, FS_EMIT>))
            : (half ? (win ? filter_scan_kernel<true, FS_COUNT, true> : filter_scan_kernel<false, FS_COUNT, true>)
                    : (win ? filter_scan_kernel<true, FS_COUNT> : filter_scan_kernel<false, FS_COUNT>));
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        eps_set_error("%s: cannot reserve %zu bytes of LDS", who, lds);
        return EPS_ELAUNCH;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(FS_THREADS), lds, s, p);
    EPS_CHECK_LAUNCH(who);
    return EPS_OK;
}

static fs_params fs_blank_params()
{
    fs_params p;
    memset(&p, 0, sizeof p);
    return p;
}

extern "C" int eps_filter_scan(const int64_t *rowptr, const int32_t *col, const int32_t *revpos, const int64_t *fixw,
                               const int32_t *splits, int64_t n_nodes, int64_t nnz, int64_t max_degree,
                               const int32_t *columns, int64_t n_columns, eps_survivors *out, void *workspace,
                               int64_t workspace_bytes, void *stream)
{
    EPS_REQUIRE(n_nodes >= 0 && n_columns >= 0 && nnz >= 0, "eps_filter_scan: negative size");
    if (n_columns == 0 || n_nodes == 0) return EPS_OK;
    EPS_REQUIRE(rowptr && col && revpos && fixw && columns && out, "eps_filter_scan: null pointer");
    fs_params p = fs_blank_params();
    p.rowptr = rowptr;
    p.col = col;
    p.revpos = revpos;
    p.fixw = fixw;
    p.columns = columns;
    p.out = out;
    return fs_launch("eps_filter_scan", p, FS_SCAN, splits, n_nodes, nnz, max_degree, n_columns, workspace, workspace_bytes, stream);
}

// ---- the candidate LIST of a block of columns of a unit-valued graph, on the scan kernel's structure ---------------------
// eps_expand_unit_count / eps_expand_unit_fill do what eps_expand_count / eps_expand_fill (expand_score.hip) do -- every 2-hop
// non-edge of columns [v_lo, v_hi) in the reference's order (filter.py:96-109), optionally with its score
// sum_w A[u,w] A[v,w] node_w[w] -- for adjacencies WITHOUT stored values, with the threshold scan's machinery: packed 64-entry
// units, 4-byte bucket records, one fixed-point weight per (v, w) from a table, 16-byte record stores.  Both orientations of
// a pair are produced (column-major order cannot be mirrored cheaply), so whole rows are walked, not row heads -- unless the
// caller passes revpos (eps_reverse_positions; symmetric pattern): then column v lists only its candidates u < v, every
// unordered pair once, for consumers that are symmetric in (u, v) themselves (the GNN filters' decoder).
extern "C" int eps_expand_unit_count(const int64_t *rowptr, const int32_t *col, const int32_t *revpos_or_null,
                                     const int32_t *splits, int64_t n_nodes, int64_t nnz, int64_t max_degree, int64_t v_lo,
                                     int64_t v_hi, const int32_t *col_order, int64_t *cand_count, void *workspace,
                                     int64_t workspace_bytes, void *stream)
{
    EPS_REQUIRE(n_nodes >= 0 && nnz >= 0 && v_lo >= 0 && v_lo <= v_hi && v_hi <= n_nodes, "eps_expand_unit_count: bad range");
    if (v_hi == v_lo) return EPS_OK;
    EPS_REQUIRE(rowptr && col && cand_count, "eps_expand_unit_count: null pointer");
    fs_params p = fs_blank_params();
    p.rowptr = rowptr;
    p.col = col;
    p.columns = col_order;
    p.col_base = (int32_t)v_lo;
    p.cand_count = cand_count;
    p.revpos = revpos_or_null;
    return fs_launch("eps_expand_unit_count", p, FS_COUNT, splits, n_nodes, nnz, max_degree, v_hi - v_lo, workspace,
                     workspace_bytes, stream, revpos_or_null != nullptr);
}

extern "C" int eps_expand_unit_fill(const int64_t *rowptr, const int32_t *col, const int32_t *revpos_or_null,
                                    const int64_t *fixw, const int32_t *splits,
                                    int64_t n_nodes, int64_t nnz, int64_t max_degree, int64_t v_lo, int64_t v_hi,
                                    const int32_t *col_order, const int64_t *colptr, int64_t *cand_count, int32_t *cand_u,
                                    int32_t *cand_v, float *score, uint32_t *status, void *workspace,
                                    int64_t workspace_bytes, void *stream)
{
    EPS_REQUIRE(n_nodes >= 0 && nnz >= 0 && v_lo >= 0 && v_lo <= v_hi && v_hi <= n_nodes, "eps_expand_unit_fill: bad range");
    if (v_hi == v_lo) return EPS_OK;
    EPS_REQUIRE(rowptr && col && colptr && cand_u && status && (fixw || !score), "eps_expand_unit_fill: null pointer");
    if (hipMemsetAsync(status, 0, sizeof(uint32_t), (hipStream_t)stream) != hipSuccess) {
        eps_set_error("eps_expand_unit_fill: cannot clear the status word");
        return EPS_ELAUNCH;
    }
    fs_params p = fs_blank_params();
    p.rowptr = rowptr;
    p.col = col;
    p.fixw = fixw;
    p.columns = col_order;
    p.col_base = (int32_t)v_lo;
    p.colptr = colptr;
    p.cand_count = cand_count;
    p.cand_u = cand_u;
    p.cand_v = cand_v;
    p.out_score = score;
    p.status = status;
    p.revpos = revpos_or_null;
    return fs_launch("eps_expand_unit_fill", p, FS_EMIT, splits, n_nodes, nnz, max_degree, v_hi - v_lo, workspace,
                     workspace_bytes, stream, revpos_or_null != nullptr);
}

// The list in ONE pass over the two-hop paths (no counting launch, no host read before the launch): the caller sizes the
// segments by an upper bound of every column's candidate count (min(two-hop paths, N): colptr_ub), the kernel fills the front of
// each segment, reports the real counts and leaves the rest of the segment unwritten.
extern "C" int eps_expand_unit_list(const int64_t *rowptr, const int32_t *col, const int32_t *revpos_or_null,
                                    const int64_t *fixw, const int32_t *splits, int64_t n_nodes, int64_t nnz, int64_t max_degree,
                                    int64_t v_lo, int64_t v_hi, const int32_t *col_order, const int64_t *colptr_ub,
                                    int64_t *cand_count, int32_t *cand_u, float *score, uint32_t *status, void *workspace,
                                    int64_t workspace_bytes, void *stream)
{
    EPS_REQUIRE(n_nodes >= 0 && nnz >= 0 && v_lo >= 0 && v_lo <= v_hi && v_hi <= n_nodes, "eps_expand_unit_list: bad range");
    if (v_hi == v_lo) return EPS_OK;
    EPS_REQUIRE(rowptr && col && colptr_ub && cand_count && cand_u && status && (fixw || !score), "eps_expand_unit_list: null pointer");
    if (hipMemsetAsync(status, 0, sizeof(uint32_t), (hipStream_t)stream) != hipSuccess) {
        eps_set_error("eps_expand_unit_list: cannot clear the status word");
        return EPS_ELAUNCH;
    }
    fs_params p = fs_blank_params();
    p.rowptr = rowptr;
    p.col = col;
    p.fixw = fixw;
    p.columns = col_order;
    p.col_base = (int32_t)v_lo;
    p.colptr = colptr_ub;
    p.cand_count = cand_count;
    p.cand_u = cand_u;
    p.out_score = score;
    p.status = status;
    p.revpos = revpos_or_null;
    p.no_pad = 1;
    return fs_launch("eps_expand_unit_list", p, FS_EMIT, splits, n_nodes, nnz, max_degree, v_hi - v_lo, workspace,
                     workspace_bytes, stream, revpos_or_null != nullptr);
}

// (one empty kernel per translation unit: launching it makes the HIP runtime load this unit's code object -- eps_warm_up)
__global__ void filter_scan_warm_kernel() {}
extern "C" void eps_warm_filter_scan(void *stream) { hipLaunchKernelGGL(filter_scan_warm_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream); }
