// Threshold scan of the filter stage's candidate set in ONE pass over the two-hop paths, gfx950 (r03).
//
// What it replaces: the same as filter_scan.hip -- filter.py:96-109 (every 2-hop non-edge), :113-142 (its heuristic score:
// adamic_utils.py:13-25, train_and_eval.py:195-216, models.py:536-542) and :160-161 under `--keep_top K` -- and it reports the
// same thing: the unordered candidates {u < v} whose score can exceed a bar.  What differs is how a column is scored.
//
// filter_scan.hip walks a column's paths twice (mark the endpoints in an id bitmap, rank them; bin the paths by rank tile)
// and sums 4-byte bucket records per tile: ~95 VALU + 18 LDS wave-instructions per path, 4x the compulsory memory traffic.
// Here a column is cut into PIECES -- runs of id windows of the endpoint space -- small enough that every endpoint of a
// piece owns a slot of an LDS table, and each path is read once and costs one table update:
//   * the id space is cut per graph into SP_M windows of equal stored-entry mass (`bounds`); `cuts[w][k]` = entries of row w
//     below bounds[k+1] (uint16) turns (row, window run) into a segment of col[] without searching;
//   * a column's paths per window are exact sums over the cut rows of its neighbours (a per-graph table, `wpaths`; summed here
//     when the caller has none), and wave 0 merges windows greedily into pieces: DIRECT when the run spans at most 2 x slots
//     ids (hub ids: the accumulator of u is slot u - lo, no key, one LDS add per path -- under hubs-first labels a third to a
//     half of all paths end there), HASH when it holds at most slots / 2 paths and known edges (open addressing, double
//     hashing, a CAS on the key word + an add on the value word), hash-PARTITIONED passes for a single window that is both
//     wide and heavy.  Unit-valued graphs with the caller's sum bounds (ssum / smax) get a third kind, PACKED: a hash piece
//     whose slot is ONE word, key (id - lo) above flag + sum -- twice the candidates per table, i.e. half as many pieces where
//     pieces are hashed, and a first-time candidate costs one LDS operation.  The field widths are set per piece: the key
//     takes the bits its id span needs, the sum field the rest, and the weights drop low bits (rounded up) until
//     min(S(v), max S(u)) + one unit per path fits -- S = a node's sum of weights over its row bounds every sum it can take
//     part in.  Between a direct run and a longer hashed one the planner compares cost per path (a piece's fixed cost is
//     worth ~2300 hashed paths, a direct path costs a quarter of a hashed one).  The known edges of v inside a piece take
//     their slots before the walk, flagged in the value word;
//   * segments are packed: a lane takes 4 consecutive entries of one row segment (16-byte load), lanes are dealt over the
//     virtual concatenation of the piece's segments (rows get dense indices; a bitmap of row starts over the unit numbering
//     plus the rank of every 32-unit word maps a unit to its row with two independent LDS reads and a popcount); a hashed
//     unit probes all four entries at once, then finishes the stragglers one entry per lane and trip (a wave needs as many
//     trips as its unluckiest entry -- 4.8 per unit -- so trips must be cheap);
//   * the table holds 32-bit SCREENING sums: weights rounded UP to 2^-shift fixed point, so a sum is an upper bound of
//     the exact 2^-40 fixed-point score of filter_scan.hip / expand_score.hip and `sum >= floor(bar)` loses no survivor.  The
//     few candidates that pass (K of 10^10) are re-scored exactly (eps_rescore_runs / eps_rescore_weighted below: int64 sums
//     of the same 2^-40 fixed-point terms, order-independent) -- the final list is bit-identical.
// Four workgroups of 256 threads per CU (variant 2) is the measured best: independent workgroups overlap each other's
// barrier-separated phases; __launch_bounds__(T, 4) keeps the fourth wave per SIMD (130 VGPRs instead of 128 cost 25-100 %).
// Symmetry, the survivor record and the dynamic column hand-out are as in filter_scan.hip.
#include "eps_common.h"
#include "scan_common.h"
#include <string.h>
#include <stdlib.h>

#define SP_EMPTY 0u             // an empty key word; a key is stored as id + 1, so a clean table is all zeros in BOTH modes
#define SP_MAXP (SP_M + 1)
#define SP_UBITS 8192           // units per range of the unit -> row bitmap (64 lanes x 128 bits: one uint4 per lane)
#ifndef SP_SB
#define SP_SB 4                 // uint4 reads a thread issues together in the table sweeps
#endif
#ifndef SP_PACKED_X8
#define SP_PACKED_X8 8u         // a packed piece holds at most this many eighths of 2^table_bits paths
#endif
#ifndef SP_MODE_RATIO
#define SP_MODE_RATIO_WIDE 8000u   // ... in a plan for sketch launches (eps_scan_plan, variant bit 24)
#define SP_MODE_RATIO 2270u     // fixed cost of a piece in hashed-path units (the planner's direct-vs-packed choice); r05: 1000 / 4000 re-measured
#endif
#ifndef SP_BATCH
#define SP_BATCH 8              // columns per ticket in the light tail of the column order
#endif
#ifndef SP_G
#define SP_EM 128               // slots of the per-workgroup set of ids a SKETCH piece has reported (a power of two)
#define SP_G 1                  // units (of 4 entries) a lane looks up, loads and inserts together
#endif

typedef int sp_v4i __attribute__((ext_vector_type(4)));

struct sp_params {
    const int64_t *rowptr;
    const int32_t *col;
    const int32_t *revpos;
    const uint32_t *fx32;       // screening weight per node (>= 1); weighted graphs: unused
    const float *val;           // stored values A[.,.] (weighted graphs: HV) or NULL
    const float *node_w;        // float node weights (weighted graphs): a path's term is (A[u,w] * A[v,w]) * node_w[w]
    float up;                   // weighted graphs: 2^shift * (1 + 2^-20), the scale of the per-row factor
    const uint16_t *cuts;       // [n_nodes][SP_M]
    const uint32_t *wpaths;     // [n_nodes][SP_M] two-hop half paths of column v per id window (per-graph table) or NULL
    const int32_t *bounds;      // [SP_M + 1]
    const int32_t *columns;
    int32_t n_columns;
    int32_t n_nodes;
    uint32_t col_bytes;
    int32_t table_bits;         // slots = 1 << table_bits (keys) + as many values
    uint32_t piece_paths;       // a hash piece holds at most this many paths (<= slots / 2)
    const uint32_t *ssum;       // [n_nodes] sum of fx32 over the node's neighbours (an upper bound of any of its pairs' sums) or NULL
    const uint32_t *smax;       // [SP_M + 1] largest ssum among the ids >= bounds[k] (0 for k = SP_M); NULL with ssum
    uint32_t packed_paths;      // a PACKED hash piece holds at most this many paths (<= slots: key and sum share a word)
    int32_t packed_dmax;        // ... and may drop at most this many low bits of the screening weights
    const uint32_t *pptr;       // [n_nodes + 1] first plan record of column v (per-graph plan table) or NULL: the kernel plans itself
    const uint4 *plan;          // one record per piece: x = paths | kind bits, y = k0 | k1 << 8 | pq << 16, z = na | nb << 16, w = lo
    uint32_t mode_ratio;        // fixed cost of a piece in units of (a hashed path's cost - a direct path's cost)
    int32_t shift;              // screening fixed point: 2^-shift
    float scale;                // 2^-shift: screening sum -> approximate score
    unsigned int *next_col;
    eps_survivors *out;
    unsigned int *status;       // bit 1: a hash table filled up (cannot happen within the piece limits; backstop); bit 2: a column's
                                //        skipped head weighs as much as the bar (the head table was built for a higher bar: nothing valid);
                                //        bit 3: a sketch piece's set of reported ids filled up (the launch is void); bit 4: sketch pieces ran
    const uint4 *colrec;        // [n_columns][2] or NULL: what a column's set-up reads of five tables, in hand-out order (one 32-byte load
                                //        at the ticket's index instead of a chain id -> row start / head / row sum / plan pointer):
                                //        {v, rowptr[v], degree, head rows | head weight, row sum, first plan record, pieces}
    const uint32_t *rowrec;     // [n_nodes][32] or NULL: per node ONE 128-byte line with everything the walk wants of a row -- words
                                //        0..15 its 32 cuts, word 16 its first entry (rowptr), word 17 its screening weight (eps_scan_row_records)
    uint32_t wide_paths;        // r06, plans for sketch launches: a packed piece of a column of <= wide_rows rows may hold this many paths (it will
    int32_t wide_rows;          //      run as a sketch piece: no keys, no sum field -- only the unit bitmap's range bounds it); 0 = off
    uint32_t sketch;            // r06: PACKED pieces of single-round columns under a bar keep no keys (see SP_EM): 0 = off
    uint32_t batch_from;        // tickets below stand for one column, tickets from here on for SP_BATCH consecutive ones (>= n_columns: none)
    const uint4 *pack;          // [nnz][2] or NULL (r06; with plan + row records): per stored entry (v, j), in CSR order, everything a
                                //        single-round column's set-up gathers for it: {w, rowptr[w], fx32[w], revpos | cut of v's FIRST piece in
                                //        row w << 16} {cuts of v's pieces 1..8 in row w, 16 bits each} -- one contiguous stream per column
                                //        instead of neighbour ids -> one 128-byte row-record line per neighbour (eps_scan_column_pack)
    const uint2 *heads;         // [n_nodes] or NULL: column v does NOT walk its first heads[v].x rows (its heaviest hub neighbours under
                                //        hubs-first labels); heads[v].y = the sum of their screening weights.  See eps_scan_heads.
};

__device__ __forceinline__ void sp_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ int sp_wave_incl_scan(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);   // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);   // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1, 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2, 3
    return x;
}

__device__ __forceinline__ uint32_t sp_wave_sum(uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_readlane(sp_wave_incl_scan((int)x), 63);
}

typedef float sp_v4f __attribute__((ext_vector_type(4)));
typedef short sp_v2s __attribute__((ext_vector_type(2)));

// Hash of an id: Fibonacci hashing.  The table index is taken from the TOP of the 32-bit product, the probe step from the
// middle, the pass of a partitioned window from the bottom.  (A full-rate 24 x 24-bit multiply was measured instead of the
// quarter-rate 32-bit one: its weaker mixing lengthens the probe sequences -- 38.0 vs 30.7 ms per launch.)
__device__ __forceinline__ uint32_t sp_mix(uint32_t x) { return x * 0x9E3779B1u; }
__device__ __forceinline__ uint32_t sp_mix2(uint32_t x) { return x * 0x85EBCA6Bu; }      // (the sketch pieces' second table)

struct sp_unit {
    sp_v4i u4;
    sp_v4f a4;                  // weighted graphs: the entries' stored values
    uint32_t fx;
    int nvalid;
};

// ---- the planner: one wave cuts a column into pieces (lane k = id window k; the extents are ballots over monotone predicates).
// Used by the scan kernel itself (no plan table) and by sp_plan_kernel (the per-graph plan table): the same code, the same pieces.
// emit(i, k0, k1, lo, hi, paths | kind bits, pq, na, nb) is called with wave-uniform arguments for piece i; returns the count.
template <class Emit>
__device__ __forceinline__ int sp_plan_column(const sp_params &p, int32_t v, int32_t dv, int lane, int32_t my_bound, uint32_t pwk,
                                              int32_t nbk, uint32_t direct_ids, Emit emit)
{
    const uint32_t ek = (uint32_t)sp_wave_incl_scan((int)pwk) - pwk;      // paths in the windows before k (lane 32: all)
    const bool pk_on = p.ssum != nullptr;      // (weighted graphs come without)
    // (what the column's walked rows can add up to: its row sum without the skipped head)
    const uint32_t sv = pk_on ? p.ssum[v] - (p.heads ? (p.heads[v].y < p.ssum[v] ? p.heads[v].y : p.ssum[v]) : 0u) : 0u;
    const uint32_t smk = pk_on ? p.smax[lane <= SP_M ? lane : SP_M] : 0u;
    // windows 0 .. kv hold ids below v
    const int kv = __popcll(__ballot(lane >= 1 && lane < SP_M && my_bound <= v - 1));
    const int32_t hi_k = my_bound < v ? my_bound : v;                    // end of the run [.., k) in id space
    int np = 0, k0 = 0;
    while (k0 <= kv) {
        const int32_t lo = __builtin_amdgcn_readlane(my_bound, k0);
        const uint32_t e0 = (uint32_t)__builtin_amdgcn_readlane((int)ek, k0);
        const bool in = lane > k0 && lane <= kv + 1;
        const int kd = k0 + __popcll(__ballot(in && (uint32_t)(hi_k - lo) <= direct_ids));
        const int32_t nb0 = __builtin_amdgcn_readlane(nbk, k0);
        // (the known edges of v inside a hash piece own slots too: they count toward its limit)
        const int kh = k0 + __popcll(__ballot(in && (ek - e0) + (uint32_t)(nbk - nb0) <= p.piece_paths));
        // PACKED hash piece: key (id - lo) and sum share a 32-bit word -- twice the keys per table.  The sum field
        // must hold any pair's sum (at most min(S(v), S(u)) / 2^d + one rounding unit per path, flag bit on top)
        // next to the key bits the run's id span needs; the weights may lose up to packed_dmax bits for it.
        // DIRECT16: a direct piece of twice the ids, two 16-bit fields (flag + 15-bit sum) per table word, under the
        // same sum bound as a packed piece
        int kp = k0, kd16 = k0;
        uint32_t ms = 0u;
        if (pk_on) {
            const uint32_t sm0 = (uint32_t)__builtin_amdgcn_readlane((int)smk, k0);
            ms = sv < sm0 ? sv : sm0;
            const uint32_t need_s = (ms >> p.packed_dmax) + (uint32_t)dv + 2u;
            if (need_s < 0x7FFFu) kd16 = k0 + __popcll(__ballot(in && (uint32_t)(hi_k - lo) <= 2u * direct_ids));
            const uint32_t span = (uint32_t)(hi_k - lo);
            const int kb = 32 - __clz((int)((span > 2u ? span : 2u) - 1u));
            const uint32_t cap = kb >= 30 ? 0u : (1u << (31 - kb)) - 1u;
            kp = k0 + __popcll(__ballot(in && (ek - e0) + (uint32_t)(nbk - nb0) <= p.packed_paths && need_s < cap));
            // (a plan for a sketch launch: such a piece keeps no keys and full 32-bit sums -- neither the id span nor the known
            //  edges nor the sum bound limit it, only its paths: one range of the unit bitmap)
            if (p.wide_paths && dv <= p.wide_rows) {
                const int kw = k0 + __popcll(__ballot(in && (ek - e0) <= p.wide_paths));
                kp = kw > kp ? kw : kp;
            }
        }
        int k1;
        uint32_t flag = 0u, pq = 0u;
        // DIRECT or PACKED when the packed run reaches further: a piece costs a fixed overhead worth `mode_ratio` extra
        // hashed paths (plan, describe, barriers, sweep), and a direct path a fraction of a hashed one -- the shorter
        // direct run wins iff (Pp - Pd) * mode_ratio < Pd * Pp  (cost per path: F / P + c_mode)
        const int kdd = kd16 > kd ? kd16 : kd;       // (the exact 32-bit sums when both kinds reach equally far)
        bool take_direct = kdd >= kh && kdd >= kp && kdd > k0;
        if (!take_direct && kdd > k0 && kp > kdd && kp >= kh) {
            const uint32_t pd = (uint32_t)__builtin_amdgcn_readlane((int)ek, kdd) - e0;
            const uint32_t pp = (uint32_t)__builtin_amdgcn_readlane((int)ek, kp) - e0;
            take_direct = (unsigned long long)(pp - pd) * p.mode_ratio < (unsigned long long)pd * pp;
        }
        if (take_direct && kd16 > kd) {
            k1 = kd16;
            flag = 0xC0000000u;
            uint32_t d = 0u;
            while ((ms >> d) + (uint32_t)dv + 2u >= 0x7FFFu) ++d;
            pq = d | (16u << 8);
        } else if (take_direct) {
            k1 = kd;
            flag = 0x80000000u;
        } else if (kp >= kh && kp > k0) {
            k1 = kp;
            flag = 0x40000000u;
            const uint32_t span = (uint32_t)(__builtin_amdgcn_readlane(hi_k, k1) - lo);
            const int kb = 32 - __clz((int)((span > 2u ? span : 2u) - 1u));
            if (p.wide_paths && dv <= p.wide_rows) {
                pq = (uint32_t)(kb < 31 ? kb : 31) << 8;              // (a sketch piece: nothing dropped, no sum field)
            } else {
                const uint32_t cap = (1u << (31 - kb)) - 1u;
                uint32_t d = 0u;
                while ((ms >> d) + (uint32_t)dv + 2u >= cap) ++d;     // (<= packed_dmax: the run passed the test with it)
                pq = d | ((uint32_t)kb << 8);
            }
        } else if (kh > k0) {
            k1 = kh;
        } else {
            k1 = k0 + 1;                                     // one window, wide and heavy: hash-partitioned passes
        }
        const uint32_t sum = (uint32_t)__builtin_amdgcn_readlane((int)ek, k1) - e0;
        if (sum) {
            emit(np, k0, k1, lo, __builtin_amdgcn_readlane(hi_k, k1), sum | flag, pq, __builtin_amdgcn_readlane(nbk, k0),
                 __builtin_amdgcn_readlane(nbk, k1));
            ++np;
        }
        k0 = k1;
    }
    return np;
}

// HV: the adjacency has stored values (collab: rank.py:32-35 keeps the summed multi-edge weights).  A path's term is then
// (A[u,w] * A[v,w]) * node_w[w] -- symmetric in (u, v) like the unit-valued one, so the half scheme holds -- and its
// screening weight is formed per path: ceil(A[u,w] * rowf) + 1 with rowf = A[v,w] * node_w[w] * 2^shift * (1 + 2^-20) per
// row (float32 products are within 2^-22 of the exact one: still an upper bound of the term's exact fixed-point value).
// (r05: the per-graph window paths are mandatory.  Until r04 a launch without them summed the window paths of every column
//  itself -- 32 counters per thread, the one code path of this kernel that spilled vector registers, reached by no caller of the
//  Python host.)
// FULL (r06): the main launch of the filter step always comes with the plan table, row and column records, sum bounds and a head
// table; compiled for exactly that, the body carries none of the fall-back pointers (cuts, fx32, pptr, columns, heads[v], ssum[v])
// and none of the branches on them -- scalar registers, which the generic body spills by the hundred (111 in r05).
// SK (r06): the launch may run packed pieces as SKETCH pieces (p.sketch); without it that code is compiled out -- it costs the
// other pieces registers (VGPR spills 0 -> 4, SGPR 40 -> 75: a hashed-only launch 8.96 -> 9.1 ms).
template <int T, bool HV, bool FULL, bool PACK, bool SK = false>
__global__ __launch_bounds__(T, 4) void scan_piece_kernel(sp_params p)      // (4 waves per SIMD: <= 128 VGPRs, the LDS share decides the rest)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int slots = 1 << p.table_bits;
    uint32_t *tkey = lds;                    // [slots]  hash mode: key (node id) or SP_EMPTY
    uint32_t *tval = lds + slots;            // [slots]  hash mode: screening sum.  Direct mode: lds[0 .. 2 slots) are the sums
    // row descriptors of the round, one uint4 per (dense) row: x = first entry of the row's segment (index into col[]),
    // y = entries of the segment, z = screening weight of the row's node, w = first 4-entry unit of the row
    uint4 *r_desc = (uint4 *)(lds + 2 * slots);          // [T + 1]
    // (offsets in words off `lds`, never through an integer cast: a pointer that loses the LDS address space turns its reads
    //  into FLAT loads, which count on vmcnt and make every look-up drain the row loads in flight)
    uint32_t *ubits = lds + ((2 * slots + 4 * (T + 1) + 3) & ~3);      // [SP_UBITS / 32] bit s: a row starts at unit s (of the range)
    uint16_t *wrank = (uint16_t *)(ubits + SP_UBITS / 32);               // [SP_UBITS / 32] rows that start before the word
    __shared__ unsigned long long s_alloc;   // units << 32 | rows handed out to the waves of a round (one 64-bit LDS add per wave)
    __shared__ int s_rbase;
    __shared__ int32_t s_pk0[SP_MAXP], s_pk1[SP_MAXP];   // pieces: window run [k0, k1)
    __shared__ int32_t s_plo[SP_MAXP], s_phi[SP_MAXP];   // ... = ids [lo, hi) (hi cut at v)
    __shared__ uint32_t s_pinfo[SP_MAXP];    // paths of the piece | direct flag << 31 | packed flag << 30
    __shared__ uint32_t s_pq[SP_MAXP];       // packed pieces: bits dropped from the weights | key bits << 8
    __shared__ int32_t s_pna[SP_MAXP], s_pnb[SP_MAXP];   // the neighbours of v with ids inside the piece's windows: vcol[na, nb)
    __shared__ int s_np;
    __shared__ unsigned int s_ticket;
    __shared__ unsigned int s_out_cur, s_out_end;
    __shared__ unsigned int s_fill_lo, s_fill_hi;       // what is left of an abandoned reservation: filled with "no survivor"
    __shared__ unsigned int s_hot;           // sketch pieces: some slot of the first half table reaches the bar (else nothing is looked at again)
    __shared__ uint32_t s_em[SP_EM];         // sketch pieces: id + 1 of every candidate reported so far (a path per report: the set dedupes)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wib = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t out_cap = p.out->capacity;
    int64_t *__restrict__ out_key = p.out->key;
    float *__restrict__ out_val = p.out->val;
    const uint32_t *__restrict__ rowptr_lo = (const uint32_t *)p.rowptr;     // nnz < 2^30: the low words suffice
    const __amdgpu_buffer_rsrc_t col_rs = __builtin_amdgcn_make_buffer_rsrc((void *)p.col, 0, p.col_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t val_rs = __builtin_amdgcn_make_buffer_rsrc((void *)(HV ? (const void *)p.val : (const void *)p.col), 0,
                                                                            p.col_bytes, 0x00020000);
    const uint32_t thr32 = sp_bar_units(p.out->threshold, p.shift);
    const uint32_t direct_ids = 2u * (uint32_t)slots;
    // Survivor slots are reserved in chunks (one global atomic each).  Without a bar every candidate of a piece survives: the
    // chunk holds a whole piece's yield.  With one, survivors are rare: a piece asks for room for 1024, and a survivor that
    // does not fit its workgroup's reservation takes a slot of its own (one more global atomic: rare).
    // (FULL: a launch with heads has a bar -- a column whose head cannot be honoured is left out below -- so the chunk size is a
    //  constant there; a caller that passes heads without a bar still gets every survivor, through single slots)
    const bool no_bar = !FULL && thr32 <= 1u;
    const bool raw_sums = FULL || p.heads != nullptr;
    const uint32_t chunk = no_bar && direct_ids > 8192u ? direct_ids : 8192u;
    const int32_t my_bound = p.bounds[lane <= SP_M ? lane : SP_M];      // lane k holds window boundary k (the plan runs in wave 0)

    for (int i = tid; i < 2 * slots; i += T) lds[i] = 0u;
    for (int i = tid; i < SP_UBITS / 32; i += T) ubits[i] = 0u;
    for (int i = tid; i < SP_EM; i += T) s_em[i] = 0u;
    if (tid == 0) {
        s_out_cur = 0u;
        s_out_end = 0u;
        s_fill_lo = 0u;
        s_fill_hi = 0u;
        s_alloc = 0ull;
    }
    bool sketch_seen = false;
    unsigned long long n_cand = 0;           // candidates seen by this thread
    uint32_t new_keys = 0u;                  // ... of the current hash piece: candidate keys this thread inserted
    const unsigned int ncol = (unsigned int)p.n_columns;
    // Columns are handed out by tickets on ONE device word; the L2 serves same-address atomics at ~11 ns each, and the light
    // columns at the end of the heaviest-first order take a workgroup ~10 us -- with 1024 workgroups drawing, the tickets, not
    // the columns, set the pace there.  So a ticket beyond `batch_from` stands for SP_BATCH consecutive columns.
    const unsigned int batch_from = p.batch_from;
    auto first_of = [&](unsigned int tk) { return tk < batch_from ? tk : batch_from + (tk - batch_from) * SP_BATCH; };
    unsigned int t = blockIdx.x < batch_from ? blockIdx.x : first_of(blockIdx.x);
    unsigned int t_end = t + (blockIdx.x < batch_from ? 1u : SP_BATCH);
    sp_barrier();

    while (t < ncol) {
        const bool last_of_ticket = t + 1u >= t_end || t + 1u >= ncol;
        unsigned int t_next = 0;
        if (tid == 0 && last_of_ticket) t_next = gridDim.x + atomicAdd(p.next_col, 1u);     // in flight while this column is scored
        int32_t v, dv;
        uint32_t vb;
        uint4 cra = make_uint4(0u, 0u, 0u, 0u), crb = make_uint4(0u, 0u, 0u, 0u);
        if (FULL || p.colrec) {
            cra = p.colrec[2 * (size_t)t];
            crb = p.colrec[2 * (size_t)t + 1];
            v = (int32_t)cra.x;
            vb = cra.y;
            dv = (int32_t)cra.z;
        } else {
            v = p.columns[t];
            vb = rowptr_lo[2 * (size_t)v];
            dv = (int32_t)(rowptr_lo[2 * (size_t)v + 2] - vb);
        }
        const int32_t *__restrict__ vcol = p.col + vb;
        const int32_t *__restrict__ vrev = p.revpos + vb;
        const int rounds = (dv + T - 1) / T;
        const bool single = rounds == 1;
        // Skipped head (eps_scan_heads): the first xv rows of the column are not walked; whatever they could add to a pair of
        // this column is at most tv, so a slot passes at thr32 - tv and eps_scan_refine adds the exact head term afterwards.
        // (the plan table that comes with a head table counts the walked rows only: the two are used together or not at all, so a
        //  head that cannot be honoured -- no bar, or a head as heavy as the bar: the table was built for a higher one -- leaves
        //  the column out and says so in the status word)
        uint32_t xv = 0u, thr_v = thr32;
        bool bad_head = false;
        if (FULL || p.heads) {
            const uint2 hd = (FULL || p.colrec) ? make_uint2(cra.w, crb.x) : p.heads[v];
            xv = hd.x;
            if (thr32 < SP_FLAG) {
                bad_head = hd.y >= thr32;
                thr_v = thr32 - hd.y;
            }
            if (bad_head && tid == 0) atomicOr(p.status, 4u);
            // a DEAD column: no pair of v sums to more than the row sum of v's screening weights -- below the bar, nothing of
            // this column can pass (a quarter of the ppa-like graph's columns at K = 4 M: a tenth of the pieces, 4 % of the paths.
            // Launches with heads only: a plain launch counts every candidate of its columns)
            if ((FULL || p.ssum) && thr32 < SP_FLAG && ((FULL || p.colrec) ? crb.y : p.ssum[v]) < thr32) bad_head = true;
        }
        if (dv > 0 && v > 0 && !bad_head) {
            uint32_t my_w = 0, my_rev = 0, my_base = 0, my_fx = 0;      // this thread's row (of the last round)
            uint32_t cut_first = 0u;
            if (PACK && single && tid < dv) {
                // (r06: ONE 16-byte load from the column's own stretch of the pack -- the chain neighbour id -> row record is gone)
                const uint4 pa = p.pack[2 * ((size_t)vb + (size_t)tid)];
                my_w = pa.x;
                my_base = pa.y;
                my_fx = pa.z;
                my_rev = pa.w & 0xFFFFu;
                cut_first = pa.w >> 16;
            }
            if (!PACK && single && tid < dv) {
                my_w = (uint32_t)vcol[tid];
                my_rev = (uint32_t)vrev[tid];
            }
            if (!PACK && single && tid < dv) {
                // (with row records the row's first entry, its weight and -- below -- its cuts come out of ONE 128-byte line: three
                //  gathers into three tables were 384 bytes of fabric traffic per walked row for 24 bytes wanted)
                my_base = (FULL || p.rowrec) ? p.rowrec[(size_t)my_w * 32 + 16] : rowptr_lo[2 * (size_t)my_w];
                my_fx = HV ? __builtin_bit_cast(uint32_t, (p.val[vb + tid] * p.node_w[my_w]) * p.up)
                           : ((FULL || p.rowrec) ? p.rowrec[(size_t)my_w * 32 + 17] : p.fx32[my_w]);
            }
            // ---- plan: merge windows into pieces.  Wave 0, lane k = window k: the extents are ballots over monotone predicates -
            if (!FULL && wib == 0 && !p.plan) {
                const uint32_t pwk = lane < SP_M ? p.wpaths[(size_t)v * SP_M + lane] : 0u;
                // lane k: neighbours of v below window boundary k (row v's own cuts: cuts[v][k - 1]; 0 for k = 0)
                const int32_t nbk = lane >= 1 && lane <= SP_M ? (int32_t)p.cuts[(size_t)v * SP_M + lane - 1] : 0;
                const int np = sp_plan_column(p, v, dv, lane, my_bound, pwk, nbk, direct_ids,
                                              [&](int i, int k0, int k1, int32_t lo, int32_t hi, uint32_t info, uint32_t pq, int32_t na, int32_t nb) {
                                                  if (lane == 0) {
                                                      s_pk0[i] = k0;
                                                      s_pk1[i] = k1;
                                                      s_plo[i] = lo;
                                                      s_phi[i] = hi;
                                                      s_pinfo[i] = info;
                                                      s_pq[i] = pq;
                                                      s_pna[i] = na;
                                                      s_pnb[i] = nb;
                                                  }
                                              });
                if (lane == 0) s_np = np;
            } else if (FULL || p.plan) {
                // the per-graph plan table: this column's records (one uint4 per piece) go straight into the piece arrays
                const uint32_t pb = (FULL || p.colrec) ? crb.z : p.pptr[v];
                const int np = (FULL || p.colrec) ? (int)crb.w : (int)(p.pptr[v + 1] - pb);
                if (tid < np) {
                    const uint4 rec = p.plan[pb + (uint32_t)tid];
                    const int k1 = (int)((rec.y >> 8) & 0xFFu);
                    const int32_t bk = p.bounds[k1];
                    s_pinfo[tid] = rec.x;
                    s_pk0[tid] = (int)(rec.y & 0xFFu);
                    s_pk1[tid] = k1;
                    s_pq[tid] = rec.y >> 16;
                    s_pna[tid] = (int32_t)(rec.z & 0xFFFFu);
                    s_pnb[tid] = (int32_t)(rec.z >> 16);
                    s_plo[tid] = (int32_t)rec.w;
                    s_phi[tid] = bk < v ? bk : v;
                }
                if (tid == 0) s_np = np;
            }
            sp_barrier();
            const int np = s_np;
            // A column of one round keeps its rows in registers, and its pieces are consecutive runs of windows: a row's segment
            // starts where its segment in the previous piece ended (runs that were skipped hold no entry of any row), and the
            // one cut a piece needs per row -- its end -- is loaded a piece ahead.
            uint32_t cut_ahead = 0u, seg_from = 0u;
            const uint16_t *__restrict__ cut_tab = (FULL || p.rowrec) ? (const uint16_t *)p.rowrec : p.cuts;      // (a row's cuts: 64 B of its record,
            const size_t cut_ld = (FULL || p.rowrec) ? 64u : (size_t)SP_M;                                        //  or its row of the cut table)
            const uint16_t *__restrict__ pack16 = (const uint16_t *)p.pack;
            if (PACK) cut_ahead = cut_first;
            else if (single && tid < dv && np > 0) cut_ahead = cut_tab[(size_t)my_w * cut_ld + s_pk1[0] - 1];

            for (int pi = 0; pi < np; ++pi) {
                const int k0 = s_pk0[pi], k1 = s_pk1[pi];
                const int na = s_pna[pi], nb = s_pnb[pi];
                const uint32_t info = s_pinfo[pi];
                const bool direct = (info >> 31) != 0u;                                   // direct, either kind
                const bool d16 = !HV && (info >> 30) == 3u;
                // r06 -- SKETCH: a packed piece of a single-round column under a bar keeps NO keys.  What a packed piece pays is not the
                // walk but the returning CAS per path and what hangs on its result (1.9 of the launch's 9.0 ms: a non-returning add
                // in its place, wrong results, 7.1 ms -- profiles/r06/scan_structures.txt).  A screening sum only has to be an UPPER
                // bound: every path adds its weight to one slot of each of two half tables under two independent hashes, a
                // candidate's sum is at most the smaller of its two slots (a count-min sketch; slots are shared, so the estimate only
                // ever errs upwards: no candidate is lost, and next to a bar of twenty path weights a tail candidate of one or two
                // paths does not get there on collisions), and a second look at the piece's paths -- two reads and a compare each --
                // reports the ids whose estimate reaches the bar, once each (s_em), unless they are neighbours of v.  Exact
                // re-scoring follows as for every survivor.  Such a piece does not count its candidates.
                const bool packed_kind = !HV && (info >> 30) == 1u;
                const bool sketch = SK && packed_kind && single && p.sketch != 0u && thr_v < SP_FLAG;      // (uniform)
                const bool packed = packed_kind && !sketch;
                const bool quant = packed || d16;                                        // weights drop pk_d low bits
                const uint32_t ppaths = info & 0x3FFFFFFFu;
                const uint32_t pq = quant ? s_pq[pi] : 0u;
                const uint32_t pk_d = pq & 0xFFu, pk_sb = quant ? 32u - (pq >> 8) : 16u;      // weight bits dropped; bits of the flag + sum field
                const uint32_t pk_flag = quant ? 1u << (pk_sb - 1u) : 0u;
                const uint32_t pk_thr = thr_v >= SP_FLAG ? 0xFFFFFFFFu : ((thr_v >> pk_d) ? (thr_v >> pk_d) : 1u);
                const int32_t lo_id = s_plo[pi], hi_id = s_phi[pi];
                // hash geometry: the smallest power-of-two table with load <= 1/2; a heavier single window goes in `parts` passes
                const uint32_t pkeys = ppaths + (uint32_t)(nb - na);           // slots the piece can need: paths + known edges
                // (closed forms, no loops: hipcc unrolls and vectorises even a three-trip scalar loop into a hundred instructions)
                int plog = 0;                                            // parts = 2^plog
                if (!direct && !packed && !sketch && pkeys > p.piece_paths) {
                    const uint32_t q = (pkeys + p.piece_paths - 1u) / p.piece_paths;      // >= 2 (piece_paths is a power of two: a shift)
                    plog = 32 - __clz((int)(q - 1u)) + 1;                // next power of two, doubled (random split: aim at a quarter load)
                }
                const uint32_t parts = 1u << plog;
                int bits;
                {
                    const uint32_t per = (pkeys + parts - 1u) >> plog;
                    const int max_bits = packed || sketch ? p.table_bits + 1 : p.table_bits;
                    const int want = per > 1u ? 33 - __clz((int)(per - 1u)) : 1;          // smallest b with 2^b >= 2 * per
                    bits = want < 10 ? 10 : (want > max_bits ? max_bits : want);
                }
                const uint32_t mask = (1u << bits) - 1u;
                const uint32_t span = (uint32_t)(hi_id - lo_id);
                const uint32_t scan_slots = d16 ? (span + 1u) >> 1 : direct ? span : (1u << bits);      // table words to sweep
                const uint32_t cand_max = direct ? span : (1u << bits);
                if (tid == 0) {
                    s_hot = 0u;
                    uint32_t need = ppaths < cand_max ? ppaths : cand_max;        // survivors <= distinct endpoints <= paths, slots
                    if (!no_bar && need > 1024u) need = 1024u;
                    if (need > chunk) need = chunk;                               // (a 16-bit direct piece without a bar: the rest takes single slots)
                    const uint32_t left = s_out_cur < s_out_end ? s_out_end - s_out_cur : 0u;
                    if (left < need) {
                        // (the slots of the old reservation that nobody took: the workgroup marks them "no survivor" after the
                        //  next barrier, so the list needs no fill before the launch and its readers stop at the slot counter)
                        s_fill_lo = s_out_cur < out_cap ? s_out_cur : out_cap;
                        s_fill_hi = left ? (s_out_end < out_cap ? s_out_end : out_cap) : s_fill_lo;
                        const unsigned long long b64 = atomicAdd(&p.out->count, (unsigned long long)chunk);
                        const uint32_t b = b64 < (unsigned long long)out_cap ? (uint32_t)b64 : out_cap;
                        s_out_cur = b;
                        s_out_end = b + chunk;
                    }
                }
                uint32_t seg_a = 0u, seg_len = 0u;
                if (single && tid < dv) {
                    const uint32_t b = cut_ahead < my_rev ? cut_ahead : my_rev;
                    seg_a = seg_from;
                    seg_len = (uint32_t)tid < xv ? 0u : b - seg_from;
                    seg_from = b;
                    if (pi + 1 < np)
                        cut_ahead = PACK && pi + 1 <= 8 ? pack16[((size_t)vb + (size_t)tid) * 16 + 7 + (pi + 1)]
                                                        : cut_tab[(size_t)my_w * cut_ld + s_pk1[pi + 1] - 1];
                }
                // (a survivor's slot: out of the workgroup's reservation, or one of its own)
                auto emit = [&](uint32_t u, uint32_t sum) {
                    uint32_t pos = atomicAdd(&s_out_cur, 1u);
                    if (pos >= s_out_end) {                          // past the reservation: a slot of its own
                        const unsigned long long q = atomicAdd(&p.out->count, 1ull);
                        pos = q < (unsigned long long)out_cap ? (uint32_t)q : out_cap;
                    }
                    if (pos < out_cap) {
                        out_key[pos] = ((int64_t)v << 32) | (int64_t)u;
                        // (a launch with skipped heads reports the WALKED sum as it is: eps_scan_refine completes it)
                        out_val[pos] = raw_sums ? __builtin_bit_cast(float, sum) : (float)sum * p.scale;
                    }
                };
                for (uint32_t part = 0; part < parts; ++part) {
                    // ---- known edges in: the neighbours of v inside the piece take their slots BEFORE the walk, with the flag bit
                    // in the value word -- whatever the paths add on top, the sweep sees that this id is no candidate (rows ascend:
                    // the neighbours inside the piece's windows are vcol[na, nb)).  The first barrier of the describe orders this
                    // against the walk.
                    // (a column of one round holds its rows in registers: lane j's row IS neighbour j)
                    for (int j = sketch ? nb : single ? tid : na + tid; j < nb; j += T) {      // (a sketch piece looks them up when it reports)
                        const uint32_t u = single ? my_w : (uint32_t)vcol[j];
                        if (j >= na && (int32_t)u >= lo_id && (int32_t)u < hi_id) {
                            if (d16) {
                                const uint32_t o = u - (uint32_t)lo_id;
                                atomicOr(&lds[o >> 1], 0x8000u << ((o & 1u) << 4));
                            } else if (direct) {
                                lds[u - (uint32_t)lo_id] = SP_FLAG;
                            } else if (packed) {
                                const uint32_t mix = sp_mix(u);
                                uint32_t h = mix >> (32 - bits);
                                const uint32_t st = ((mix >> 7) | 1u) & mask;
                                const uint32_t word = ((u - (uint32_t)lo_id) << pk_sb) | pk_flag;
                                for (uint32_t tries = 0; tries <= mask; ++tries) {
                                    if (atomicCAS(&lds[h], 0u, word) == 0u) break;
                                    h = (h + st) & mask;
                                }
                            } else {
                                const uint32_t mix = sp_mix(u);
                                if ((mix & (parts - 1u)) == part) {
                                    uint32_t h = mix >> (32 - bits);
                                    const uint32_t st = ((mix >> 7) | 1u) & mask;
                                    for (uint32_t tries = 0; tries <= mask; ++tries) {
                                        if (atomicCAS(&tkey[h], SP_EMPTY, u + 1u) == SP_EMPTY) {
                                            tval[h] = SP_FLAG;
                                            break;
                                        }
                                        h = (h + st) & mask;
                                    }
                                }
                            }
                        }
                    }
                    for (int r = 0; r < rounds; ++r) {
                        // ---- describe the round's row segments inside the piece ------------------------------------------
                        const int j = r * T + tid;
                        uint32_t w = my_w, rev = my_rev, base = my_base, fx = my_fx;
                        uint32_t len = 0u, a = 0u;
                        if (single) {
                            a = seg_a;
                            len = seg_len;
                        } else if (j < dv) {
                            w = (uint32_t)vcol[j];
                            rev = (uint32_t)vrev[j];
                            base = (FULL || p.rowrec) ? p.rowrec[(size_t)w * 32 + 16] : rowptr_lo[2 * (size_t)w];
                            fx = HV ? __builtin_bit_cast(uint32_t, (p.val[vb + j] * p.node_w[w]) * p.up)
                                    : ((FULL || p.rowrec) ? p.rowrec[(size_t)w * 32 + 17] : p.fx32[w]);
                            const uint16_t *crow = cut_tab + (size_t)w * cut_ld;
                            uint32_t b = crow[k1 - 1];
                            if (k0 > 0) a = crow[k0 - 1];
                            a = a < rev ? a : rev;
                            b = b < rev ? b : rev;
                            len = (uint32_t)j < xv ? 0u : b - a;
                        }
                        // Rows with entries in the piece get DENSE indices (block scans of the unit counts and of the non-empty
                        // flags), and the unit -> row map is a bitmap of row starts over the unit numbering plus the rank of every
                        // 32-unit word: a lane finds the row of its unit with two independent LDS reads and a popcount (a binary
                        // search over the unit prefix -- ten dependent reads -- was 12 of the launch's 32 ms).
                        const int units = (int)((len + 3u) >> 2);
                        const int flag = units > 0 ? 1 : 0;
                        // (one wave scan for both: a row has < 2^14 units, a wave < 2^20; the non-empty flags count in bits 24 up)
                        const int incl_c = sp_wave_incl_scan(units | (flag << 24));
                        const int incl = incl_c & 0xFFFFFF, incl_f = (int)((uint32_t)incl_c >> 24);
                        // Each wave takes its range of unit numbers AND of dense row indices with one 64-bit LDS add: whatever
                        // order the waves arrive in, both ranges are handed out in that same order -- a row that starts at a
                        // later unit has a larger index, which is all the start-bit ranks need (no block scan, no barrier for it).
                        unsigned long long got = 0ull;
                        if (lane == 63) got = atomicAdd(&s_alloc, ((unsigned long long)(uint32_t)incl << 32) | (unsigned long long)(uint32_t)incl_f);
                        const int woff = __builtin_amdgcn_readlane((int)(got >> 32), 63);
                        const int woff_f = __builtin_amdgcn_readlane((int)(uint32_t)got, 63);
                        const uint32_t a_u = (uint32_t)(incl - units + woff);          // first unit of this thread's row
                        if (flag) {
                            const int d = incl_f - 1 + woff_f;
                            r_desc[d] = make_uint4(base + a, len, quant ? (fx + ((1u << pk_d) - 1u)) >> pk_d : fx, a_u);
                            if (a_u < (uint32_t)SP_UBITS) atomicOr(&ubits[a_u >> 5], 1u << (a_u & 31u));      // (the first range's start bits)
                        }
                        sp_barrier();
                        if (s_fill_hi > s_fill_lo) {                     // (uniform; rare: one reservation in ~a thousand pieces)
                            const uint32_t f0 = s_fill_lo, f1 = s_fill_hi;
                            for (uint32_t i = f0 + (uint32_t)tid; i < f1; i += T) {
                                out_key[i] = -1;
                                out_val[i] = -__builtin_inff();
                            }
                        }                                                // (tid 0 closes the range after the walk's barrier)
                        const int total = (int)(s_alloc >> 32);
                        for (uint32_t ulo = 0; ulo < (uint32_t)total; ulo += SP_UBITS) {       // (one range unless > 32 k paths)
                            const uint32_t uhi = ulo + SP_UBITS < (uint32_t)total ? ulo + SP_UBITS : (uint32_t)total;
                            if (ulo > 0u) {
                                for (int i = tid; i < SP_UBITS / 32; i += T) ubits[i] = 0u;
                                if (tid == 0) s_rbase = 0;
                                sp_barrier();
                                const unsigned long long before = __ballot(flag && a_u + (uint32_t)units <= ulo);
                                if (lane == 0 && before) atomicAdd(&s_rbase, __popcll(before));
                                if (flag && a_u + (uint32_t)units > ulo && a_u < uhi) {
                                    const uint32_t pos = (a_u > ulo ? a_u : ulo) - ulo;
                                    atomicOr(&ubits[pos >> 5], 1u << (pos & 31));
                                }
                                sp_barrier();
                            }
                            const int rbase = ulo > 0u ? s_rbase : 0;
                            {   // rank of every word's first bit: every wave computes all of them (identical values: no barrier)
                                const uint4 b4 = *(const uint4 *)(ubits + 4 * lane);
                                const int c0 = __popc(b4.x), c1 = c0 + __popc(b4.y), c2 = c1 + __popc(b4.z), c3 = c2 + __popc(b4.w);
                                const int ex = sp_wave_incl_scan(c3) - c3;
                                uint2 pk;
                                pk.x = (uint32_t)ex | ((uint32_t)(ex + c0) << 16);
                                pk.y = (uint32_t)(ex + c1) | ((uint32_t)(ex + c2) << 16);
                                *(uint2 *)(wrank + 4 * lane) = pk;
                                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            }
                        // ---- walk: lane = one 4-entry unit.  A GROUP of SP_G units per lane is looked up, loaded and inserted
                        // together: its look-ups run side by side, its loads are in flight together, and its 4 x SP_G table updates
                        // share the probing loop (a trip costs one LDS round trip however many of them are still pending).
                        auto fetch_group = [&](int it0, sp_unit (&f)[SP_G]) {
                            int lo[SP_G];
#pragma unroll
                            for (int q = 0; q < SP_G; ++q) {
                                const uint32_t sr = (uint32_t)((it0 + q) * T + tid);       // unit, relative to the range
                                const uint32_t wd = sr < SP_UBITS ? sr >> 5 : 0u;
                                const uint32_t bits = ubits[wd];
                                const int rk = (int)wrank[wd];
                                lo[q] = rbase + rk + __popc(bits & ((2u << (sr & 31u)) - 1u)) - 1;
                                lo[q] = lo[q] < 0 ? 0 : lo[q];
                            }
#pragma unroll
                            for (int q = 0; q < SP_G; ++q) {
                                // (every LDS read unconditional, the range test a select: a read inside a branch makes hipcc
                                //  drain the loads in flight at the branch)
                                const int s = (int)ulo + (it0 + q) * T + tid;
                                const uint4 rd = r_desc[lo[q]];
                                const uint32_t us = rd.w, rl = rd.y, rs = rd.x;
                                f[q].fx = rd.z;
                                const bool in = s < (int)uhi;
                                const int off = (s - (int)us) * 4;
                                int left = (int)rl - off;
                                left = in ? left : 0;
                                f[q].nvalid = left < 0 ? 0 : (left > 4 ? 4 : left);
                                const uint32_t at = in ? rs + (uint32_t)off : (p.col_bytes >> 2);
                                f[q].u4 = __builtin_amdgcn_raw_buffer_load_b128(col_rs, (int)(at * 4u), 0, 0);
                                if (HV) f[q].a4 = __builtin_bit_cast(sp_v4f, __builtin_amdgcn_raw_buffer_load_b128(val_rs, (int)(at * 4u), 0, 0));
                            }
                        };
                        auto path_fx = [&](const sp_unit &f, int e) -> uint32_t {
                            if (!HV) return f.fx;
                            return (uint32_t)__builtin_ceilf(f.a4[e] * __builtin_bit_cast(float, f.fx)) + 1u;
                        };
                        auto consume_group = [&](const sp_unit (&f)[SP_G]) {
                            if (sketch) {
                                const uint32_t half = 1u << (bits - 1);
#pragma unroll
                                for (int q = 0; q < SP_G; ++q)
#pragma unroll
                                    for (int e = 0; e < 4; ++e)
                                        if (e < f[q].nvalid) {
                                            const uint32_t id = (uint32_t)f[q].u4[e];
                                            atomicAdd(&lds[sp_mix(id) >> (33 - bits)], f[q].fx);
                                            atomicAdd(&lds[half + (sp_mix2(id) >> (33 - bits))], f[q].fx);
                                        }
                            } else if (d16) {
#pragma unroll
                                for (int q = 0; q < SP_G; ++q)
#pragma unroll
                                    for (int e = 0; e < 4; ++e)
                                        if (e < f[q].nvalid) {
                                            const uint32_t o = (uint32_t)(f[q].u4[e] - lo_id);
                                            atomicAdd(&lds[o >> 1], f[q].fx << ((o & 1u) << 4));
                                        }
                            } else if (direct) {
#pragma unroll
                                for (int q = 0; q < SP_G; ++q)
#pragma unroll
                                    for (int e = 0; e < 4; ++e)
                                        if (e < f[q].nvalid) atomicAdd(&lds[(uint32_t)(f[q].u4[e] - lo_id)], path_fx(f[q], e));
                            } else if (packed) {
                                // Round one: all of the lane's entries at once (most find their slot at the first probe).  What is
                                // left -- an entry in six at load 1/2 -- is finished ONE entry per lane and trip: the trips a wave
                                // needs are set by its unluckiest entry (the longest probe sequence among 256), and a trip that
                                // looks at one entry costs a quarter of one that steps through four mostly idle ones.
                                constexpr int E = 4 * SP_G;
                                uint32_t lid[E], h[E], mixv[E], old[E];
                                uint32_t pend = 0u;
#pragma unroll
                                for (int q = 0; q < SP_G; ++q) pend |= ((1u << f[q].nvalid) - 1u) << (4 * q);
#pragma unroll
                                for (int i = 0; i < E; ++i) {
                                    const uint32_t id = (uint32_t)f[i >> 2].u4[i & 3];
                                    mixv[i] = sp_mix(id);
                                    lid[i] = id - (uint32_t)lo_id;
                                    h[i] = mixv[i] >> (32 - bits);
                                }
#ifdef SP_ABL_PACKED_NOHASH
                                // (timing-only ablation, tools/r06_packed_nohash.sh: what a PERFECT table for the sparse tail -- one
                                //  non-returning add per path, no key, no probing -- would leave of the launch; the results are wrong.
                                //  This is the measurement the sketch pieces came from: 8.96 -> 7.10 ms, profiles/r06/scan_structures.txt)
#pragma unroll
                                for (int i = 0; i < E; ++i)
                                    if (pend & (1u << i)) atomicAdd(&lds[h[i]], f[i >> 2].fx);
                                pend = 0u;
#endif
#pragma unroll
                                for (int wide = 0; wide < 2; ++wide) {           // (two wide rounds: a fifth of the entries miss the first)
#pragma unroll
                                    for (int i = 0; i < E; ++i)
                                        if (pend & (1u << i)) old[i] = atomicCAS(&lds[h[i]], 0u, (lid[i] << pk_sb) | f[i >> 2].fx);
#pragma unroll
                                    for (int i = 0; i < E; ++i)
                                        if (pend & (1u << i)) {
                                            if (old[i] == 0u) {                  // a candidate seen for the first time: key and weight went in at once
                                                pend &= ~(1u << i);
                                                ++new_keys;
                                            } else if ((old[i] >> pk_sb) == lid[i]) {
                                                atomicAdd(&lds[h[i]], f[i >> 2].fx);
                                                pend &= ~(1u << i);
                                            } else if (wide == 0) {
                                                h[i] = (h[i] + (((mixv[i] >> 7) | 1u) & mask)) & mask;
                                            }
                                        }
                                    if (wide == 0 && !__ballot(pend != 0u)) break;
                                }
                                uint32_t ch = 0u, cstep = 0u, clid = 0u, cfx = 0u, tries = 0u;
                                bool have = false;
                                while (__ballot(have || pend != 0u)) {
                                    if (!have && pend != 0u) {                   // take up the next unfinished entry
                                        const int e = __ffs((int)pend) - 1;
                                        pend &= pend - 1u;
                                        uint32_t m = mixv[0];
                                        ch = h[0];
                                        clid = lid[0];
                                        cfx = f[0].fx;
#pragma unroll
                                        for (int i = 1; i < E; ++i)
                                            if (e == i) {
                                                m = mixv[i];
                                                ch = h[i];
                                                clid = lid[i];
                                                cfx = f[i >> 2].fx;
                                            }
                                        cstep = ((m >> 7) | 1u) & mask;
                                        have = true;
                                    }
                                    if (have) {
                                        ch = (ch + cstep) & mask;
                                        const uint32_t o = atomicCAS(&lds[ch], 0u, (clid << pk_sb) | cfx);
                                        if (o == 0u) {
                                            ++new_keys;
                                            have = false;
                                        } else if ((o >> pk_sb) == clid) {
                                            atomicAdd(&lds[ch], cfx);
                                            have = false;
                                        }
                                    }
                                    if (++tries > 4u * (mask + 2u)) {            // the table is full (backstop; never within the piece limits)
                                        if (have || pend) atomicOr(p.status, 2u);
                                        have = false;
                                        pend = 0u;
                                    }
                                }
                            } else {
                                constexpr int E = 4 * SP_G;
                                uint32_t key[E], h[E], mixv[E];
                                uint32_t pend = 0u;
#pragma unroll
                                for (int q = 0; q < SP_G; ++q) pend |= ((1u << f[q].nvalid) - 1u) << (4 * q);
#pragma unroll
                                for (int i = 0; i < E; ++i) {
                                    const uint32_t id = (uint32_t)f[i >> 2].u4[i & 3];
                                    mixv[i] = sp_mix(id);
                                    key[i] = id + 1u;
                                    h[i] = mixv[i] >> (32 - bits);           // (the probe step is only formed on a collision)
                                }
                                if (parts > 1u) {                            // (uniform: a partitioned window keeps its own pass's ids)
#pragma unroll
                                    for (int i = 0; i < E; ++i)
                                        if ((mixv[i] & (parts - 1u)) != part) pend &= ~(1u << i);
                                }
                                // (round one for all entries at once, then one unfinished entry per lane and trip: see the packed walk)
                                uint32_t old[E], pfx[E];
#pragma unroll
                                for (int i = 0; i < E; ++i) pfx[i] = path_fx(f[i >> 2], i & 3);
#pragma unroll
                                for (int i = 0; i < E; ++i)
                                    if (pend & (1u << i)) old[i] = atomicCAS(&tkey[h[i]], SP_EMPTY, key[i]);
#pragma unroll
                                for (int i = 0; i < E; ++i)
                                    if ((pend & (1u << i)) && (old[i] == SP_EMPTY || old[i] == key[i])) {
                                        atomicAdd(&tval[h[i]], pfx[i]);
                                        pend &= ~(1u << i);
                                        new_keys += old[i] == SP_EMPTY ? 1u : 0u;      // a candidate seen for the first time
                                    }
                                uint32_t ch = 0u, cstep = 0u, ckey = 0u, cfx = 0u, tries = 0u;
                                bool have = false;
                                while (__ballot(have || pend != 0u)) {
                                    if (!have && pend != 0u) {
                                        const int e = __ffs((int)pend) - 1;
                                        pend &= pend - 1u;
                                        uint32_t m = mixv[0];
                                        ch = h[0];
                                        ckey = key[0];
                                        cfx = pfx[0];
#pragma unroll
                                        for (int i = 1; i < E; ++i)
                                            if (e == i) {
                                                m = mixv[i];
                                                ch = h[i];
                                                ckey = key[i];
                                                cfx = pfx[i];
                                            }
                                        cstep = ((m >> 7) | 1u) & mask;
                                        have = true;
                                    }
                                    if (have) {
                                        ch = (ch + cstep) & mask;
                                        const uint32_t o = atomicCAS(&tkey[ch], SP_EMPTY, ckey);
                                        if (o == SP_EMPTY || o == ckey) {
                                            atomicAdd(&tval[ch], cfx);
                                            new_keys += o == SP_EMPTY ? 1u : 0u;
                                            have = false;
                                        }
                                    }
                                    if (++tries > 4u * (mask + 2u)) {            // the table is full (backstop; never within the piece limits)
                                        if (have || pend) atomicOr(p.status, 2u);
                                        have = false;
                                        pend = 0u;
                                    }
                                }
                            }
                        };
                        {
                            const int n_iter = (int)(uhi - ulo + T - 1) / T;      // uniform over the workgroup
                            sp_unit fa[SP_G], fb[SP_G];
                            fetch_group(0, fa);
                            if (n_iter <= SP_G) {          // one group (every hash piece: <= piece_paths / 4 units): nothing to overlap
                                consume_group(fa);
                            } else {
                                // (refills are unconditional -- a group past the end is empty units at an out-of-range address:
                                //  no traffic -- because a conditional refill becomes a phi copy and hipcc then drains vmcnt)
                                for (int it0 = 0; it0 < n_iter; it0 += 2 * SP_G) {
                                    fetch_group(it0 + SP_G, fb);
                                    consume_group(fa);
                                    fetch_group(it0 + 2 * SP_G, fa);
                                    consume_group(fb);
                                }
                            }
                        }
                        if (sketch) {
                            // the second look (one range: a piece of <= packed_paths paths has fewer than SP_UBITS units)
                            if (tid == 0 && !sketch_seen) atomicOr(p.status, 16u);      // (informational: sketch pieces ran in this launch)
                            sketch_seen = true;
                            sp_barrier();
                            const uint32_t half = 1u << (bits - 1);
                            // An id's estimate is at most its slot in the FIRST half table: when no slot there reaches the bar -- the
                            // usual case in the tail: survivors are a handful per thousand pieces -- nothing needs a second look.
                            {
                                uint32_t mx = 0u;
                                for (uint32_t i = 4u * (uint32_t)tid; i < half; i += 4u * T) {
                                    const uint4 a4 = *(const uint4 *)(lds + i);
                                    const uint32_t m01 = a4.x > a4.y ? a4.x : a4.y, m23 = a4.z > a4.w ? a4.z : a4.w;
                                    const uint32_t m = m01 > m23 ? m01 : m23;
                                    mx = m > mx ? m : mx;
                                }
                                if (__ballot(mx >= thr_v) && lane == 0) s_hot = 1u;
                            }
                            sp_barrier();
                            if (s_hot) {
                            const int n_iter = (int)(uhi - ulo + T - 1) / T;
                            auto report = [&](uint32_t u, uint32_t est) {
                                int a = na, b = nb;                       // a neighbour of v is no candidate (rows ascend: vcol[na, nb))
                                while (a < b) {
                                    const int mid = (a + b) >> 1;
                                    if (vcol[mid] < (int32_t)u) a = mid + 1; else b = mid;
                                }
                                if (a < nb && vcol[a] == (int32_t)u) return;
                                const uint32_t em_mask = (p.sketch >> 1 ? p.sketch >> 1 : (uint32_t)SP_EM) - 1u;      // (a smaller set: tests)
                                uint32_t h = (sp_mix(u) >> 9) & em_mask;
                                for (uint32_t tries = 0; tries <= em_mask; ++tries) {
                                    const uint32_t old = atomicCAS(&s_em[h], 0u, u + 1u);
                                    if (old == 0u) {                      // the first path of u that gets here reports it
                                        emit(u, est);
                                        return;
                                    }
                                    if (old == u + 1u) return;
                                    h = (h + 1u) & em_mask;
                                }
                                atomicOr(p.status, 8u);                   // the set is full: the launch is void (the host repeats it without sketch pieces)
                            };
                            auto look = [&](const sp_unit (&f)[SP_G]) {
                                uint32_t est[4 * SP_G];
#pragma unroll
                                for (int i = 0; i < 4 * SP_G; ++i) {
                                    const uint32_t id = (uint32_t)f[i >> 2].u4[i & 3];
                                    const uint32_t ea = lds[sp_mix(id) >> (33 - bits)], eb = lds[half + (sp_mix2(id) >> (33 - bits))];
                                    est[i] = ea < eb ? ea : eb;
                                }
#pragma unroll
                                for (int i = 0; i < 4 * SP_G; ++i)
                                    if ((i & 3) < f[i >> 2].nvalid && est[i] >= thr_v) report((uint32_t)f[i >> 2].u4[i & 3], est[i]);
                            };
                            sp_unit fa[SP_G], fb[SP_G];
                            fetch_group(0, fa);
                            if (n_iter <= SP_G) {
                                look(fa);
                            } else {
                                for (int it0 = 0; it0 < n_iter; it0 += 2 * SP_G) {
                                    fetch_group(it0 + SP_G, fb);
                                    look(fa);
                                    fetch_group(it0 + 2 * SP_G, fa);
                                    look(fb);
                                }
                            }
                            }
                        }
                            if (uhi < (uint32_t)total) sp_barrier();       // (the next range rewrites the start bits)
                        }
                        sp_barrier();        // the next round / the scan follows: descriptors and table updates are complete
                        // (leave the start bits and the hand-out counter clean for the next describe; a barrier separates them from
                        //  it: the sweep's own, or -- columns of several rounds -- one more)
                        for (int i = tid; i < SP_UBITS / 32; i += T) ubits[i] = 0u;
                        if (tid == 0) {
                            s_alloc = 0ull;
                            s_fill_hi = 0u;
                        }
                        if (r + 1 < rounds) sp_barrier();
                    }
                    // ---- scan the table: count the candidates, report the survivors, leave it clean --------------------------
                    // (the sweeps read in batches of SP_SB uint4 per thread before they look at any of them: one LDS round trip per
                    //  batch instead of one per 16 bytes; the trip counts are uniform over the workgroup)
                    // A word is tested as a SIGNED number: a known edge's flag is its sign bit, so `(int)word >= bar` is false for it
                    // (and for an empty word), and the largest of a uint4's four words decides with one branch whether any of them
                    // needs a closer look -- survivors are a handful per piece.
                    uint32_t cnt_here = 0u;
                    const int thr_s = thr_v >= SP_FLAG ? 0x7FFFFFFF : (int)thr_v;      // (sums stay below 2^31 - 2)
                    if (d16) {
                        // (two fields per word: the low one shifted up, the high one masked, both signed as above.  The packed
                        //  16-bit instructions -- v_pk_min / max / add_i16 on the word as it is, 5 instead of 9 per word -- were
                        //  measured: 17.75 vs 17.57 ms on the same box)
                        const int thr_h = pk_thr >= 0x8000u ? 0x7FFFFFFF : (int)(pk_thr << 16);
                        const uint32_t n4 = (scan_slots + 3u) & ~3u;
                        for (uint32_t i0 = 0; i0 < n4; i0 += 4u * SP_SB * T) {
                            uint4 s4[SP_SB];
#pragma unroll
                            for (int b = 0; b < SP_SB; ++b) {
                                const uint32_t i = i0 + 4u * T * b + 4u * tid;
                                s4[b] = i < n4 ? *(const uint4 *)(lds + i) : make_uint4(0u, 0u, 0u, 0u);
                            }
#pragma unroll
                            for (int b = 0; b < SP_SB; ++b) {
                                const uint32_t i = i0 + 4u * T * b + 4u * tid;
                                if (i < n4) *(uint4 *)(lds + i) = make_uint4(0u, 0u, 0u, 0u);
                                const uint32_t wv[4] = {s4[b].x, s4[b].y, s4[b].z, s4[b].w};
                                int x[8];
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    x[2 * e] = (int)(wv[e] << 16);
                                    x[2 * e + 1] = (int)(wv[e] & 0xFFFF0000u);
                                }
#pragma unroll
                                for (int e = 0; e < 8; ++e) cnt_here += x[e] > 0 ? 1u : 0u;
                                int m = x[0];
#pragma unroll
                                for (int e = 1; e < 8; ++e) m = x[e] > m ? x[e] : m;
                                if (m >= thr_h) {
#pragma unroll
                                    for (int e = 0; e < 8; ++e)
                                        if (x[e] >= thr_h) emit((uint32_t)lo_id + 2u * i + e, ((uint32_t)x[e] >> 16) << pk_d);
                                }
                            }
                        }
                    } else if (direct) {
                        const uint32_t n4 = (scan_slots + 3u) & ~3u;
                        for (uint32_t i0 = 0; i0 < n4; i0 += 4u * SP_SB * T) {
                            uint4 s4[SP_SB];
#pragma unroll
                            for (int b = 0; b < SP_SB; ++b) {
                                const uint32_t i = i0 + 4u * T * b + 4u * tid;
                                s4[b] = i < n4 ? *(const uint4 *)(lds + i) : make_uint4(0u, 0u, 0u, 0u);
                            }
#pragma unroll
                            for (int b = 0; b < SP_SB; ++b) {
                                const uint32_t i = i0 + 4u * T * b + 4u * tid;
                                if (i < n4) *(uint4 *)(lds + i) = make_uint4(0u, 0u, 0u, 0u);
                                const int sv[4] = {(int)s4[b].x, (int)s4[b].y, (int)s4[b].z, (int)s4[b].w};
#pragma unroll
                                for (int e = 0; e < 4; ++e) cnt_here += sv[e] > 0 ? 1u : 0u;      // reached and not a known edge
                                const int m01 = sv[0] > sv[1] ? sv[0] : sv[1], m23 = sv[2] > sv[3] ? sv[2] : sv[3];
                                if ((m01 > m23 ? m01 : m23) >= thr_s) {
#pragma unroll
                                    for (int e = 0; e < 4; ++e)
                                        if (sv[e] >= thr_s) emit((uint32_t)lo_id + i + e, (uint32_t)sv[e]);
                                }
                            }
                        }
                    } else if (sketch) {
                        // (nothing to read: the second look reported what there was; both half tables and the set go back to zero)
                        for (uint32_t i = 4u * (uint32_t)tid; i < scan_slots; i += 4u * T) *(uint4 *)(lds + i) = make_uint4(0u, 0u, 0u, 0u);
                        for (int i = tid; i < SP_EM; i += T) s_em[i] = 0u;
                    } else if (packed) {
                        // (shifted left by the key bits a word is flag | sum at the top: signed again)
                        const uint32_t kb = 32u - pk_sb;
                        const int thr_p = pk_thr >= pk_flag ? 0x7FFFFFFF : (int)(pk_thr << kb);
                        for (uint32_t i0 = 0; i0 < scan_slots; i0 += 4u * SP_SB * T) {
                            uint4 s4[SP_SB];
#pragma unroll
                            for (int b = 0; b < SP_SB; ++b) {
                                const uint32_t i = i0 + 4u * T * b + 4u * tid;
                                s4[b] = i < scan_slots ? *(const uint4 *)(lds + i) : make_uint4(0u, 0u, 0u, 0u);
                            }
#pragma unroll
                            for (int b = 0; b < SP_SB; ++b) {
                                const uint32_t i = i0 + 4u * T * b + 4u * tid;
                                if (i < scan_slots) *(uint4 *)(lds + i) = make_uint4(0u, 0u, 0u, 0u);
                                const uint32_t sv[4] = {s4[b].x, s4[b].y, s4[b].z, s4[b].w};
                                const int x[4] = {(int)(sv[0] << kb), (int)(sv[1] << kb), (int)(sv[2] << kb), (int)(sv[3] << kb)};
                                const int m01 = x[0] > x[1] ? x[0] : x[1], m23 = x[2] > x[3] ? x[2] : x[3];
                                if ((m01 > m23 ? m01 : m23) >= thr_p) {
#pragma unroll
                                    for (int e = 0; e < 4; ++e)
                                        if (x[e] >= thr_p) emit((uint32_t)lo_id + (sv[e] >> pk_sb), ((uint32_t)x[e] >> kb) << pk_d);
                                }
                            }
                        }
                    } else {
                        // (the candidates were counted when their keys went in; the key words are only read for a survivor)
                        for (uint32_t i0 = 0; i0 < scan_slots; i0 += 4u * SP_SB * T) {
                            uint4 s4[SP_SB];
#pragma unroll
                            for (int b = 0; b < SP_SB; ++b) {
                                const uint32_t i = i0 + 4u * T * b + 4u * tid;
                                s4[b] = i < scan_slots ? *(const uint4 *)(tval + i) : make_uint4(0u, 0u, 0u, 0u);
                            }
#pragma unroll
                            for (int b = 0; b < SP_SB; ++b) {
                                const uint32_t i = i0 + 4u * T * b + 4u * tid;
                                const int sv[4] = {(int)s4[b].x, (int)s4[b].y, (int)s4[b].z, (int)s4[b].w};
                                const int m01 = sv[0] > sv[1] ? sv[0] : sv[1], m23 = sv[2] > sv[3] ? sv[2] : sv[3];
                                if ((m01 > m23 ? m01 : m23) >= thr_s) {
#pragma unroll
                                    for (int e = 0; e < 4; ++e)
                                        if (sv[e] >= thr_s) emit(tkey[i + e] - 1u, (uint32_t)sv[e]);
                                }
                                if (i < scan_slots) {
                                    *(uint4 *)(tkey + i) = make_uint4(0u, 0u, 0u, 0u);
                                    *(uint4 *)(tval + i) = make_uint4(0u, 0u, 0u, 0u);
                                }
                            }
                        }
                    }
                    n_cand += (unsigned long long)cnt_here + (unsigned long long)new_keys;
                    new_keys = 0u;
                    sp_barrier();
                }
            }
        }
        if (!last_of_ticket) {
            ++t;
            sp_barrier();                    // (a column without pieces has no barrier of its own: s_np and the piece arrays change hands here)
            continue;
        }
        if (tid == 0) s_ticket = t_next;
        sp_barrier();
        const unsigned int tk = s_ticket;
        sp_barrier();
        t = first_of(tk);
        t_end = t + (tk < batch_from ? 1u : SP_BATCH);
    }
    // what is left of the workgroup's last reservation
    {
        const uint32_t f0 = s_out_cur < out_cap ? s_out_cur : out_cap, f1 = s_out_end < out_cap ? s_out_end : out_cap;
        for (uint32_t i = f0 + (uint32_t)tid; i < f1; i += T) {
            out_key[i] = -1;
            out_val[i] = -__builtin_inff();
        }
    }
    // candidates scored by this workgroup: one atomic per wave
    {
        unsigned long long x = n_cand;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d);
        if (lane == 0 && x) atomicAdd(&p.out->n_candidates, x);
    }
}

// ---- per-graph tables --------------------------------------------------------------------------------------------------
// cuts[w][k] = number of entries of row w with id < bounds[k + 1], k = 0 .. SP_M - 1 (uint16: needs max degree < 65536)
__global__ void sp_cuts_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, int64_t n_nodes,
                               const int32_t *__restrict__ bounds, uint16_t *__restrict__ cuts)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_nodes * SP_M; i += stride) {
        const int64_t w = i / SP_M;
        const int k = (int)(i % SP_M);
        const int32_t bound = bounds[k + 1];
        int64_t lo = rowptr[w], hi = rowptr[w + 1];
        const int64_t wb = lo;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (col[mid] < bound) lo = mid + 1; else hi = mid;
        }
        cuts[i] = (uint16_t)(lo - wb);
    }
}

// wpaths[v][k] = two-hop half paths of column v that end in id window k: the sum over v's rows of the row head's entries
// inside the window (exact, from the cut table).  One wave per column; the scan's planner then reads 128 bytes per column
// instead of a cut row per (column, neighbour).
__global__ void sp_window_paths_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                       const int32_t *__restrict__ revpos, const uint16_t *__restrict__ cuts, int64_t n_nodes,
                                       const uint2 *__restrict__ heads, uint32_t *__restrict__ wpaths,
                                       const int32_t *__restrict__ columns, int64_t n_columns)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    // (columns given: only those rows of the table are computed -- the bar sample of a one-shot run, r06)
    const int64_t count = columns ? n_columns : n_nodes;
    for (int64_t idx = wave; idx < count; idx += n_waves) {
        const int64_t v = columns ? (int64_t)columns[idx] : idx;
        const int64_t b = rowptr[v] + (heads ? (int64_t)heads[v].x : 0ll), e = rowptr[v + 1];      // (a skipped head is not walked)
        uint32_t cnt[SP_M];
#pragma unroll
        for (int k = 0; k < SP_M; ++k) cnt[k] = 0u;
        for (int64_t i = b + lane; i < e; i += 64) {
            const uint32_t rev = (uint32_t)revpos[i];
            const uint4 *row = (const uint4 *)(cuts + (size_t)col[i] * SP_M);
            uint32_t prev = 0u;
#pragma unroll
            for (int q = 0; q < SP_M / 8; ++q) {
                const uint4 c = row[q];
                const uint32_t wds[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    uint32_t a = wds[h] & 0xFFFFu, bb = wds[h] >> 16;
                    a = a < rev ? a : rev;
                    bb = bb < rev ? bb : rev;
                    cnt[q * 8 + h * 2] += a - prev;
                    cnt[q * 8 + h * 2 + 1] += bb - a;
                    prev = bb;
                }
            }
        }
        uint32_t mine = 0u;
#pragma unroll
        for (int k = 0; k < SP_M; ++k) {
            const uint32_t s = sp_wave_sum(cnt[k]);
            if (lane == k) mine = s;
        }
        if (lane < SP_M) wpaths[v * SP_M + lane] = mine;
    }
}

// fx32[i] = max(1, ceil(fixw[i] / 2^(40 - shift))): the node weights of the scan in the screening fixed point, rounded UP
__global__ void sp_screen_weights_kernel(const int64_t *__restrict__ fixw, int64_t n, int shift, uint32_t *__restrict__ fx32,
                                         unsigned int *__restrict__ bad)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int down = 40 - shift;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const long long f = fixw[i];
        if (f < 0) {
            atomicOr(bad, 1u);                        // negative weights: the sums are no upper bounds any more
            fx32[i] = 1u;
            continue;
        }
        const unsigned long long q = ((unsigned long long)f + ((1ull << down) - 1ull)) >> down;
        if (q > 0xFFFFFFFFull) atomicOr(bad, 2u);
        fx32[i] = q ? (uint32_t)q : 1u;
    }
}

// The window boundaries: M windows of equal stored-entry mass -- bounds[k] = 1 + the first node whose row ENDS at or beyond
// k x nnz / M (rowptr is the prefix of the degrees), made non-decreasing; bounds[0] = 0, bounds[M] = N.  One small block.
__global__ __launch_bounds__(64) void sp_bounds_kernel(const int64_t *__restrict__ rowptr, int64_t n_nodes, int32_t *__restrict__ bounds)
{
    __shared__ int32_t b[SP_M + 1];
    const int k = threadIdx.x;
    if (k <= SP_M) {
        int64_t r = k == 0 ? 0 : n_nodes;
        if (k > 0 && k < SP_M) {
            const double target = (double)k * ((double)rowptr[n_nodes] / (double)SP_M);
            int64_t lo = 0, hi = n_nodes;                     // smallest i with rowptr[i + 1] >= target (n_nodes if none)
            while (lo < hi) {
                const int64_t mid = lo + ((hi - lo) >> 1);
                if ((double)rowptr[mid + 1] >= target) hi = mid; else lo = mid + 1;
            }
            r = lo + 1 < n_nodes ? lo + 1 : n_nodes;
        }
        b[k] = (int32_t)r;
    }
    __syncthreads();
    if (k == 0) {
        int32_t m = 0;
        for (int i = 0; i <= SP_M; ++i) {
            m = b[i] > m ? b[i] : m;
            bounds[i] = m;
        }
    }
}

extern "C" int eps_scan_bounds(const int64_t *rowptr, int64_t n_nodes, int32_t *bounds, void *stream)
{
    EPS_REQUIRE(n_nodes >= 0 && n_nodes < (1ll << 31) && rowptr && bounds, "eps_scan_bounds: bad argument");
    hipLaunchKernelGGL(sp_bounds_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, rowptr, n_nodes, bounds);
    EPS_CHECK_LAUNCH("eps_scan_bounds");
    return EPS_OK;
}

// ssum[v] = sum of the screening weights over row v, clamped to 2^31 - 1: no pair with endpoint v sums to more (the bound the
// packed and 16-bit direct pieces are sized by).  One wave per row; on the way: the largest ssum per id window (-> smax, the
// suffix maxima, by sp_suffix_max_kernel) and the smallest screening weight of a node with at least two neighbours (only such
// a node is ever a common neighbour: the floor under a path's term that bounds the number of paths behind a sum).
__global__ __launch_bounds__(256) void sp_row_sums_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                          const uint32_t *__restrict__ fx32, const int32_t *__restrict__ bounds,
                                                          int64_t n_nodes, uint32_t *__restrict__ ssum, uint32_t *__restrict__ wmax,
                                                          uint32_t *__restrict__ min_fx)
{
    const int lane = threadIdx.x & 63;
    const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= n_nodes) return;
    const int64_t b = rowptr[v], e = rowptr[v + 1];
    unsigned long long acc = 0ull;
    for (int64_t i = b + lane; i < e; i += 64) acc += fx32[col[i]];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    if (lane == 0) {
        const uint32_t sv = acc < 0x7FFFFFFFull ? (uint32_t)acc : 0x7FFFFFFFu;
        ssum[v] = sv;
        int k = 0;                                            // the window of v: the last k with bounds[k] <= v
        for (int step = 32; step >= 1; step >>= 1)
            if (k + step < SP_M && bounds[k + step] <= (int32_t)v) k += step;
        // (look before the atomic: the cells only move one way, so a value that cannot move them needs no atomic -- the last
        //  window holds half the nodes, and 300 k atomics on one address would take 30 ms)
        if (sv > __atomic_load_n(&wmax[k], __ATOMIC_RELAXED)) atomicMax(&wmax[k], sv);
        if (e - b >= 2) {
            const uint32_t fv = fx32[v];
            if (fv < __atomic_load_n(min_fx, __ATOMIC_RELAXED)) atomicMin(min_fx, fv);
        }
    }
}

__global__ void sp_suffix_max_kernel(const uint32_t *__restrict__ wmax, uint32_t *__restrict__ smax)
{
    if (threadIdx.x == 0) {
        uint32_t m = 0u;
        smax[SP_M] = 0u;
        for (int k = SP_M - 1; k >= 0; --k) {
            m = wmax[k] > m ? wmax[k] : m;
            smax[k] = m;
        }
    }
}

extern "C" int eps_scan_row_sums(const int64_t *rowptr, const int32_t *col, const uint32_t *fx32, const int32_t *bounds,
                                 int64_t n_nodes, uint32_t *ssum, uint32_t *smax, uint32_t *min_fx, void *workspace, void *stream)
{
    EPS_REQUIRE(n_nodes >= 0 && n_nodes < (1ll << 31), "eps_scan_row_sums: bad size");
    EPS_REQUIRE(smax && min_fx && workspace, "eps_scan_row_sums: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(workspace, 0, SP_M * sizeof(uint32_t), s) != hipSuccess || hipMemsetAsync(min_fx, 0xFF, sizeof(uint32_t), s) != hipSuccess) {
        eps_set_error("eps_scan_row_sums: cannot clear the workspace");
        return EPS_ELAUNCH;
    }
    if (n_nodes > 0) {
        EPS_REQUIRE(rowptr && col && fx32 && bounds && ssum, "eps_scan_row_sums: null pointer");
        hipLaunchKernelGGL(sp_row_sums_kernel, dim3((unsigned)((n_nodes + 3) / 4)), dim3(256), 0, s, rowptr, col, fx32, bounds, n_nodes,
                           ssum, (uint32_t *)workspace, min_fx);
    }
    hipLaunchKernelGGL(sp_suffix_max_kernel, dim3(1), dim3(64), 0, s, (const uint32_t *)workspace, smax);
    EPS_CHECK_LAUNCH("eps_scan_row_sums");
    return EPS_OK;
}

// ---- exact re-scoring of the screened survivors -----------------------------------------------------------------------------
// The survivors of a scan are pairs of hubs: a few thousand nodes u recur in hundreds of pairs each.  The list comes sorted by
// (u, v) as keys (u << 32) | v; a workgroup takes 256 consecutive pairs, and for every run of equal u inside them turns N(u)
// into an LDS bitmap over the id space (windows of RS_BITS ids when the space is wider), streams the short rows N(v) of the
// run against it -- one wave per pair, coalesced -- and sums the exact weights of the hits in float64.  The weights are
// multiples of 2^-40 below 2^12, so the float64 sum is exact whatever the order: (float)sum is eps_filter_scan's score, bit
// for bit (adamic_utils.py:13-25 / train_and_eval.py:195-216 / models.py:536-542 with the engine's fixed-point definition).
#ifndef RS_THREADS
#define RS_THREADS 1024
#endif
#define RS_CHUNK 256
#define RS_BITS (1 << 20)       // ids per bitmap window: 128 KiB of LDS
#define RS_SHORT 512            // rows up to this long go through rescore_short_kernel
#ifndef RS_GROUP
#define RS_GROUP 128            // consecutive 256-pair chunks that go to the same XCD (32 k pairs: most of a block of 2^9 v)
#endif
#ifndef RS_GB
#define RS_GB 4                 // weight gathers of a trip issued together (r06: 16 x 64-bit partial sums in flight were 32 of the kernel's 94 VGPRs)
#endif
#ifndef RS_MINW
#define RS_MINW 8               // waves per SIMD the kernel is compiled for: 8 = two 1024-thread workgroups per CU (<= 64 VGPRs)
#endif
#ifndef RS_NB
#define RS_NB 8                 // entries of N(v) a lane has in flight per trip: a trip is three dependent latencies (row, bitmap,
#endif                          // weights) and the survivors' rows are long (~1100 entries on the ppa-like graph: 2.2 G entries to stream
                                // for 2 M pairs -- 4 / 8 / 12 / 16 in flight: 6.4 / 5.5 / 5.4 / 5.1 ms for all 4.85 M pairs)

__global__ __launch_bounds__(RS_THREADS, RS_MINW) void rescore_runs_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                                 const int64_t *__restrict__ fixw, int32_t n_nodes,
                                                                 const int64_t *__restrict__ keys, int64_t n,
                                                                 float *__restrict__ out, unsigned int *__restrict__ next_chunk,
                                                                 const int64_t *__restrict__ n_dev)
{
    if (n_dev) {                 // (r06: the list's length lives on the device -- the sorts in front read it there too)
        const int64_t c = *n_dev;
        n = c < 0 ? 0 : (c < n ? c : n);
    }
    extern __shared__ __attribute__((aligned(16))) uint32_t bm[];          // RS_BITS / 32 words
    __shared__ unsigned long long s_starts[RS_CHUNK / 64];
    __shared__ long long s_sum[RS_CHUNK];
    __shared__ unsigned int s_c;
    const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
    constexpr int W = RS_THREADS / 64;
    const int words = (n_nodes < RS_BITS ? (n_nodes + 31) >> 5 : RS_BITS >> 5);
    // (one descriptor over all of col[]: 16-byte loads at 4-byte-aligned offsets, out-of-range lanes read zeros at a far offset)
    const __amdgpu_buffer_rsrc_t col_rs = __builtin_amdgcn_make_buffer_rsrc((void *)col, 0, (int)(uint32_t)(rowptr[n_nodes] * 4), 0x00020000);
    for (int i = tid; i < words; i += RS_THREADS) bm[i] = 0u;
    const int64_t n_chunks = (n + RS_CHUNK - 1) / RS_CHUNK;
    // XCD-aware hand-out (r05).  The pairs come sorted by (block of 2^9 consecutive v, u, v): neighbouring chunks stream the rows of
    // the same few hundred v -- 2 MB, which an XCD's 4 MB of L2 holds, if the workgroups of that XCD work on the same chunks.  So
    // groups of RS_GROUP consecutive chunks are dealt round-robin over the eight XCDs, each XCD draws from ITS counter (the id from
    // HW_REG_XCC_ID: blockIdx says which blocks share an XCD, not which), and an XCD that runs out helps the next one.  Placement is
    // speed only: any workgroup may score any chunk.
    unsigned int xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    for (;;) {
        if (tid == 0) {
            unsigned int got = 0xFFFFFFFFu;
            for (unsigned int j = 0; j < 8u; ++j) {
                const unsigned int y = (xcc + j) & 7u;
                const unsigned int t = atomicAdd(&next_chunk[y], 1u);
                const unsigned long long c = ((unsigned long long)(t / RS_GROUP) * 8ull + y) * RS_GROUP + t % RS_GROUP;
                if (c < (unsigned long long)n_chunks) {
                    got = (unsigned int)c;
                    break;
                }
            }
            s_c = got;
        }
        __syncthreads();
        const int64_t c = s_c == 0xFFFFFFFFu ? n_chunks : (int64_t)s_c;
        if (c >= n_chunks) break;
        const int64_t c0 = c * RS_CHUNK;
        const int cn = (int)(n - c0 < RS_CHUNK ? n - c0 : RS_CHUNK);
        // run starts inside the chunk
        if (tid < RS_CHUNK) {
            s_sum[tid] = 0ll;
            bool start = false;
            if (tid < cn) start = tid == 0 || (keys[c0 + tid] >> 32) != (keys[c0 + tid - 1] >> 32);
            const unsigned long long m = __ballot(start);
            if (lane == 0) s_starts[wib] = m;
        }
        __syncthreads();
        int s = 0;
        while (s < cn) {
            // end of the run that starts at s: the next start bit after s
            int e = cn;
            for (int q = s >> 6; q < RS_CHUNK / 64; ++q) {
                unsigned long long m = s_starts[q];
                if (q == (s >> 6)) m &= (s & 63) == 63 ? 0ull : ~0ull << ((s & 63) + 1);
                if (m) {
                    e = q * 64 + __builtin_ctzll(m);
                    break;
                }
            }
            if (e > cn) e = cn;
            const int32_t u = (int32_t)(keys[c0 + s] >> 32);
            const int64_t ub = rowptr[u], ue = rowptr[u + 1];
            if (ue - ub <= RS_SHORT) {        // a short row: its pairs are rescore_short_kernel's (no bitmap, no barriers)
                s = e;
                continue;
            }
            for (int32_t wlo = 0; wlo < n_nodes; wlo += RS_BITS) {
                // N(u) inside the id window -> bits (rows ascend; a plain scan of the row is cheap next to the pairs)
                for (int64_t i = ub + tid; i < ue; i += RS_THREADS) {
                    const uint32_t x = (uint32_t)(col[i] - wlo);
                    if (x < (uint32_t)RS_BITS) atomicOr(&bm[x >> 5], 1u << (x & 31));
                }
                __syncthreads();
                // two pairs per wave, one per half: the rows N(v) are short (a few hundred entries), so half a wave with four
                // loads in flight per lane covers a row in two or three trips, and twice as many pairs are in flight per CU
                for (int p0 = s + 2 * wib; p0 < e; p0 += 2 * W) {
                    const int pi = p0 + (lane >> 5);
                    const bool live = pi < e;
                    const int32_t v = live ? (int32_t)(keys[c0 + pi] & 0xFFFFFFFFll) : 0;
                    // (32-bit entry indices: nnz < 2^30 -- r06: the 64-bit index arithmetic and sixteen 64-bit partial sums per lane
                    //  had the kernel at 94 VGPRs = ONE 1024-thread workgroup per CU; at <= 64 two are resident and the waves that
                    //  hide this kernel's three dependent latencies per trip double)
                    const uint32_t vb = live ? (uint32_t)rowptr[v] : 0u, ve = live ? (uint32_t)rowptr[v + 1] : 0u;
                    const int hl = lane & 31;
                    long long acc = 0ll;
                    uint32_t longest = ve - vb;
                    {
                        const uint32_t o = (uint32_t)__shfl_xor((int)longest, 32);
                        longest = o > longest ? o : longest;
                    }
                    // (16-byte loads, four entries a lane: a quarter of the vector-memory instructions of one-entry loads for the
                    //  same bytes -- the rows are what this kernel streams, 8.8 GB per step on the bench graph)
                    for (uint32_t off = 0; off < longest; off += 32 * RS_NB) {     // (uniform trip count over the wave)
                        sp_v4i wv[RS_NB / 4];
#pragma unroll
                        for (int b = 0; b < RS_NB / 4; ++b) {
                            const uint32_t i = vb + off + (uint32_t)(b * 128 + 4 * hl);
                            wv[b] = __builtin_amdgcn_raw_buffer_load_b128(col_rs, (int)(i < ve ? i * 4u : 0xFFFFFFF0u), 0, 0);
                        }
                        // (the weight gathers of a trip in batches of RS_GB: all of a batch's loads issued before any is added)
#pragma unroll
                        for (int h = 0; h < RS_NB / RS_GB; ++h) {
                            long long add[RS_GB];
#pragma unroll
                            for (int bb = 0; bb < RS_GB; ++bb) {
                                const int b = h * RS_GB + bb;
                                const uint32_t i = vb + off + (uint32_t)((b >> 2) * 128 + 4 * hl + (b & 3));
                                const int32_t w = wv[b >> 2][b & 3];
                                const uint32_t x = (uint32_t)(w - wlo);
                                const bool hit = i < ve && x < (uint32_t)RS_BITS && ((bm[x >> 5] >> (x & 31)) & 1u);
#ifdef RS_ABL_NOWEIGHT           /* (timing-only ablation, wrong results: what the dependent weight gathers of a trip cost) */
                                add[bb] = hit ? 1ll : 0ll;
#else
                                add[bb] = hit ? (long long)fixw[w] : 0ll;
#endif
                            }
#pragma unroll
                            for (int bb = 0; bb < RS_GB; ++bb) acc += add[bb];
                        }
                    }
#pragma unroll
                    for (int d = 16; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
                    if (hl == 0 && live) s_sum[pi] += acc;
                }
                __syncthreads();
                for (int64_t i = ub + tid; i < ue; i += RS_THREADS) {
                    const uint32_t x = (uint32_t)(col[i] - wlo);
                    if (x < (uint32_t)RS_BITS) bm[x >> 5] = 0u;
                }
                __syncthreads();
            }
            s = e;
        }
        if (tid < cn) {
            const int32_t u = (int32_t)(keys[c0 + tid] >> 32);
            if (rowptr[u + 1] - rowptr[u] > RS_SHORT) out[c0 + tid] = (float)((double)s_sum[tid] * (1.0 / (double)(1ll << 40)));
        }
        __syncthreads();
    }
}

// The pairs whose u has at most RS_SHORT entries (most distinct u have few survivors each: a bitmap per run would cost three
// workgroup barriers for a handful of pairs).  Under hubs-first labels v is the lighter endpoint, so both rows are short: a
// wave stages the shorter row in its own 2 KiB of LDS (no barrier: wave-private), spreads the other row over its lanes and
// looks every entry up by a binary search in LDS.  Same exact float64 sums.
#define RSS_THREADS 256
__global__ __launch_bounds__(RSS_THREADS) void rescore_short_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                                  const int64_t *__restrict__ fixw, const int64_t *__restrict__ keys,
                                                                  int64_t n, float *__restrict__ out, const int64_t *__restrict__ n_dev)
{
    if (n_dev) {
        const int64_t c = *n_dev;
        n = c < 0 ? 0 : (c < n ? c : n);
    }
    __shared__ int32_t s_stage[RSS_THREADS / 64][RS_SHORT];
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    int32_t *stage = s_stage[wib];
    const int64_t wave = (int64_t)blockIdx.x * (RSS_THREADS / 64) + wib;
    const int64_t n_waves = (int64_t)gridDim.x * (RSS_THREADS / 64);
    for (int64_t pi = wave; pi < n; pi += n_waves) {
        const int64_t key = keys[pi];
        const int32_t u = (int32_t)(key >> 32), v = (int32_t)(key & 0xFFFFFFFFll);
        const int64_t ub = rowptr[u], ue = rowptr[u + 1];
        if (ue - ub > RS_SHORT) continue;                      // rescore_runs_kernel's
        const int64_t vb = rowptr[v], ve = rowptr[v + 1];
        const bool u_short = ue - ub <= ve - vb;
        const int64_t sb = u_short ? ub : vb, se = u_short ? ue : ve;       // staged (the shorter: <= RS_SHORT entries)
        const int64_t lb = u_short ? vb : ub, le = u_short ? ve : ue;       // spread over the lanes
        const int ns = (int)(se - sb);
        int pow2 = 1;
        while (pow2 < ns) pow2 <<= 1;
        for (int i = lane; i < pow2; i += 64) stage[i] = i < ns ? col[sb + i] : 0x7fffffff;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        long long acc = 0ll;
        for (int64_t i0 = lb; i0 < le; i0 += 64) {
            const int64_t i = i0 + lane;
            const int32_t w = i < le ? col[i] : -1;
            int lo = 0;                                           // last position with stage[pos] <= w
            for (int step = pow2 >> 1; step >= 1; step >>= 1)
                if (stage[lo + step] <= w) lo += step;
            if (w >= 0 && ns > 0 && stage[lo] == w) acc += (long long)fixw[w];
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
        if (lane == 0) out[pi] = (float)((double)acc * (1.0 / (double)(1ll << 40)));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the next pair overwrites the staged row)
    }
}

// A weighted pair: term = (A[u,w] * A[v,w]) * node_w[w] in float32 (the association eps_expand_fill uses: symmetric in u, v),
// converted to 2^-40 fixed point and summed in int64.  One wave per pair: the shorter row spread over the lanes, each entry
// looked up in the longer row by a binary search in global memory (the lists of weighted graphs -- collab -- are short).
__global__ __launch_bounds__(256) void rescore_weighted_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                              const float *__restrict__ val, const float *__restrict__ node_w,
                                                              const int64_t *__restrict__ keys, int64_t n, float *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t pi = wave; pi < n; pi += n_waves) {
        const int64_t key = keys[pi];
        const int32_t u = (int32_t)(key >> 32), v = (int32_t)(key & 0xFFFFFFFFll);
        const int64_t ub = rowptr[u], ue = rowptr[u + 1], vb = rowptr[v], ve = rowptr[v + 1];
        const bool u_short = ue - ub <= ve - vb;
        const int64_t sb = u_short ? ub : vb, se = u_short ? ue : ve, lb = u_short ? vb : ub, le = u_short ? ve : ue;
        long long acc = 0ll;
        for (int64_t i = sb + lane; i < se; i += 64) {
            const int32_t w = col[i];
            int64_t lo = lb, hi = le;
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (col[mid] < w) lo = mid + 1; else hi = mid;
            }
            if (lo < le && col[lo] == w) {
                const float term = (val[i] * val[lo]) * node_w[w];
                acc += __double2ll_rn((double)term * (double)(1ll << 40));
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
        if (lane == 0) out[pi] = (float)((double)acc * (1.0 / (double)(1ll << 40)));
    }
}

// keys: (u << 32) | v sorted ascending (runs of equal u); fixw[i] = the 2^-40 fixed-point weight of node i (eps_fixed_weights);
// out[i] = score of pair i as float32 of the exact sum.  Unit-valued adjacency.
static int rescore_runs_launch(const int64_t *rowptr, const int32_t *col, const int64_t *fixw, int64_t n_nodes, const int64_t *keys,
                               int64_t n, const int64_t *n_dev, float *out, void *stream);

extern "C" int eps_rescore_runs(const int64_t *rowptr, const int32_t *col, const int64_t *fixw, int64_t n_nodes,
                                const int64_t *keys, int64_t n, float *out, void *stream)
{
    return rescore_runs_launch(rowptr, col, fixw, n_nodes, keys, n, nullptr, out, stream);
}

// The same with the list's length read on the DEVICE: min(*n_dev, n_max) pairs (the grid is sized for n_max).
extern "C" int eps_rescore_runs_dev(const int64_t *rowptr, const int32_t *col, const int64_t *fixw, int64_t n_nodes,
                                    const int64_t *keys, int64_t n_max, const int64_t *n_dev, float *out, void *stream)
{
    EPS_REQUIRE(n_dev, "eps_rescore_runs_dev: null count");
    return rescore_runs_launch(rowptr, col, fixw, n_nodes, keys, n_max, n_dev, out, stream);
}

static int rescore_runs_launch(const int64_t *rowptr, const int32_t *col, const int64_t *fixw, int64_t n_nodes, const int64_t *keys,
                               int64_t n, const int64_t *n_dev, float *out, void *stream)
{
    EPS_REQUIRE(n >= 0 && n_nodes >= 0 && n_nodes < (1ll << 31), "eps_rescore_runs: bad size");      // (col[] is addressed with 32-bit byte offsets: nnz < 2^30, like eps_scan_screen)
    if (n == 0) return EPS_OK;
    EPS_REQUIRE(rowptr && col && fixw && keys && out, "eps_rescore_runs: null pointer");
    hipStream_t s = (hipStream_t)stream;
    unsigned int *counter = nullptr;
    const int rc = eps_take_counters8(&counter, s, "eps_rescore_runs");
    if (rc) return rc;
    const size_t lds = (size_t)(n_nodes < RS_BITS ? ((n_nodes + 31) >> 5) : (RS_BITS >> 5)) * 4 + 16;
    if (hipFuncSetAttribute((const void *)rescore_runs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        eps_set_error("eps_rescore_runs: cannot reserve %zu bytes of LDS", lds);
        return EPS_ELAUNCH;
    }
    int64_t blocks = (n + RS_CHUNK - 1) / RS_CHUNK;
    const int64_t per_cu = (160 * 1024 - 2048) / (int64_t)(lds + 2560);            // workgroups the LDS lets a CU hold
    const int64_t cap = (int64_t)eps_num_cus() * (per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu));
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(rescore_runs_kernel, dim3((unsigned)blocks), dim3(RS_THREADS), lds, s, rowptr, col, fixw, (int32_t)n_nodes, keys,
                       n, out, counter, n_dev);
    {
        int64_t sb = (n + RSS_THREADS / 64 - 1) / (RSS_THREADS / 64);
        const int64_t scap = (int64_t)eps_num_cus() * 8;
        if (sb > scap) sb = scap;
        hipLaunchKernelGGL(rescore_short_kernel, dim3((unsigned)sb), dim3(RSS_THREADS), 0, s, rowptr, col, fixw, keys, n, out, n_dev);
    }
    EPS_CHECK_LAUNCH("eps_rescore_runs");
    return EPS_OK;
}

// The same for an adjacency with stored values (any order of the keys).
extern "C" int eps_rescore_weighted(const int64_t *rowptr, const int32_t *col, const float *val, const float *node_w,
                                    int64_t n_nodes, const int64_t *keys, int64_t n, float *out, void *stream)
{
    EPS_REQUIRE(n >= 0 && n_nodes >= 0, "eps_rescore_weighted: bad size");
    if (n == 0) return EPS_OK;
    EPS_REQUIRE(rowptr && col && val && node_w && keys && out, "eps_rescore_weighted: null pointer");
    int64_t blocks = (n + 3) / 4;
    const int64_t cap = (int64_t)eps_num_cus() * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(rescore_weighted_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rowptr, col, val, node_w, keys,
                       n, out);
    EPS_CHECK_LAUNCH("eps_rescore_weighted");
    return EPS_OK;
}

extern "C" int32_t eps_scan_windows(void) { return SP_M; }

// rowrec[w * 32 + 0 .. 15] = the 32 cuts of row w, [16] = its first entry (rowptr, low word), [17] = its screening weight, rest 0:
// one 128-byte line per node holds what the scan's walk gathers per row.
__global__ __launch_bounds__(256) void sp_rowrec_kernel(const uint16_t *__restrict__ cuts, const int64_t *__restrict__ rowptr,
                                                        const uint32_t *__restrict__ fx32, int64_t n_nodes, uint32_t *__restrict__ rowrec)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_nodes * 32; i += stride) {
        const int64_t w = i >> 5;
        const int k = (int)(i & 31);
        uint32_t x = 0u;
        if (k < SP_M / 2) x = ((const uint32_t *)cuts)[w * (SP_M / 2) + k];
        else if (k == 16) x = (uint32_t)rowptr[w];
        else if (k == 17) x = fx32[w];
        rowrec[i] = x;
    }
}

extern "C" int eps_scan_row_records(const uint16_t *cuts, const int64_t *rowptr, const uint32_t *fx32, int64_t n_nodes, uint32_t *rowrec,
                                    void *stream)
{
    EPS_REQUIRE(n_nodes >= 0 && n_nodes < (1ll << 31), "eps_scan_row_records: bad size");
    if (n_nodes == 0) return EPS_OK;
    EPS_REQUIRE(cuts && rowptr && fx32 && rowrec && ((uintptr_t)rowrec & 127) == 0 && ((uintptr_t)cuts & 3) == 0,
                "eps_scan_row_records: null or misaligned pointer (row records are 128-byte lines)");
    static_assert(SP_M == 32, "a row record holds 32 cuts in its first 64 bytes");
    int64_t blocks = (n_nodes * 32 + 255) / 256;
    const int64_t cap = (int64_t)eps_num_cus() * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(sp_rowrec_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, cuts, rowptr, fx32, n_nodes, rowrec);
    EPS_CHECK_LAUNCH("eps_scan_row_records");
    return EPS_OK;
}

extern "C" int eps_scan_cuts(const int64_t *rowptr, const int32_t *col, int64_t n_nodes, const int32_t *bounds, uint16_t *cuts,
                             void *stream)
{
    EPS_REQUIRE(n_nodes >= 0, "eps_scan_cuts: negative size");
    if (n_nodes == 0) return EPS_OK;
    EPS_REQUIRE(rowptr && col && bounds && cuts, "eps_scan_cuts: null pointer");
    EPS_REQUIRE(((uintptr_t)cuts & 15) == 0, "eps_scan_cuts: cuts must be 16-byte aligned");
    int64_t blocks = (n_nodes * SP_M + 255) / 256;
    const int64_t cap = (int64_t)eps_num_cus() * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(sp_cuts_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rowptr, col, n_nodes, bounds, cuts);
    EPS_CHECK_LAUNCH("eps_scan_cuts");
    return EPS_OK;
}

extern "C" int eps_scan_window_paths(const int64_t *rowptr, const int32_t *col, const int32_t *revpos, const uint16_t *cuts,
                                     int64_t n_nodes, const uint32_t *heads_or_null, uint32_t *wpaths, void *stream)
{
    EPS_REQUIRE(n_nodes >= 0, "eps_scan_window_paths: negative size");
    if (n_nodes == 0) return EPS_OK;
    EPS_REQUIRE(rowptr && col && revpos && cuts && wpaths && ((uintptr_t)cuts & 15) == 0, "eps_scan_window_paths: null or misaligned pointer");
    int64_t blocks = (n_nodes + 3) / 4;
    const int64_t cap = (int64_t)eps_num_cus() * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(sp_window_paths_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rowptr, col, revpos, cuts,
                       n_nodes, (const uint2 *)heads_or_null, wpaths, (const int32_t *)nullptr, (int64_t)0);
    EPS_CHECK_LAUNCH("eps_scan_window_paths");
    return EPS_OK;
}

// The same table for the listed columns only (rows of other columns are left as they are): the bar sample of a one-shot run scans
// ~1000 columns with the launch planning them itself -- the whole-graph table (0.7 ms) and the plan built from it (1.0 ms) are then
// only built when a launch without skipped heads wants them.
extern "C" int eps_scan_window_paths_columns(const int64_t *rowptr, const int32_t *col, const int32_t *revpos, const uint16_t *cuts,
                                             int64_t n_nodes, const int32_t *columns, int64_t n_columns, uint32_t *wpaths, void *stream)
{
    EPS_REQUIRE(n_nodes >= 0 && n_columns >= 0, "eps_scan_window_paths_columns: negative size");
    if (n_nodes == 0 || n_columns == 0) return EPS_OK;
    EPS_REQUIRE(rowptr && col && revpos && cuts && columns && wpaths, "eps_scan_window_paths_columns: null pointer");
    EPS_REQUIRE(((uintptr_t)cuts & 15) == 0, "eps_scan_window_paths_columns: cuts must be 16-byte aligned");
    int64_t blocks = (n_columns + 3) / 4;
    const int64_t cap = (int64_t)eps_num_cus() * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(sp_window_paths_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rowptr, col, revpos, cuts,
                       n_nodes, (const uint2 *)nullptr, wpaths, columns, n_columns);
    EPS_CHECK_LAUNCH("eps_scan_window_paths_columns");
    return EPS_OK;
}

extern "C" int eps_scan_screen_weights(const int64_t *fixw, int64_t n, int32_t shift, uint32_t *fx32, uint32_t *bad,
                                       void *stream)
{
    EPS_REQUIRE(n >= 0 && shift >= 0 && shift <= 40, "eps_scan_screen_weights: bad argument");
    EPS_REQUIRE(bad, "eps_scan_screen_weights: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(bad, 0, sizeof(uint32_t), s) != hipSuccess) {
        eps_set_error("eps_scan_screen_weights: cannot clear the flag");
        return EPS_ELAUNCH;
    }
    if (n == 0) return EPS_OK;
    EPS_REQUIRE(fixw && fx32, "eps_scan_screen_weights: null pointer");
    int64_t blocks = (n + 255) / 256;
    const int64_t cap = (int64_t)eps_num_cus() * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(sp_screen_weights_kernel, dim3((unsigned)blocks), dim3(256), 0, s, fixw, n, (int)shift, fx32, bad);
    EPS_CHECK_LAUNCH("eps_scan_screen_weights");
    return EPS_OK;
}

// ---- the per-column pack (r06) ---------------------------------------------------------------------------------------------
// pack[e] for stored entry e = (v, j) of the scanned graph, CSR order (two uint4): what a single-round column's set-up wants of its
// j-th neighbour w -- id, first entry, screening weight (row record words 16, 17), the reverse position, and the cuts of row w at
// the ends of column v's first nine pieces (row record words 0..15, indexed by the plan's k1 - 1).  One wave per column.
__global__ __launch_bounds__(256) void sp_pack_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                      const int32_t *__restrict__ revpos, const uint32_t *__restrict__ rowrec,
                                                      const uint32_t *__restrict__ pptr, const uint4 *__restrict__ plan, int64_t n_nodes,
                                                      uint4 *__restrict__ pack)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t v = wave; v < n_nodes; v += n_waves) {
        const int64_t vb = rowptr[v], ve = rowptr[v + 1];
        const uint32_t pb = pptr[v];
        const int np = (int)(pptr[v + 1] - pb);
        int k1 = 1;
        if (lane < np && lane < 9) k1 = (int)((plan[pb + (uint32_t)lane].y >> 8) & 0xFFu);
        int kk[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) kk[i] = __shfl(k1, i);
        for (int64_t e = vb + lane; e < ve; e += 64) {
            const uint32_t w = (uint32_t)col[e];
            const uint32_t *__restrict__ rr = rowrec + (size_t)w * 32;
            const uint16_t *__restrict__ cc = (const uint16_t *)rr;
            uint32_t c[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) c[i] = i < np ? (uint32_t)cc[kk[i] - 1] : 0u;
            pack[2 * e] = make_uint4(w, rr[16], rr[17], ((uint32_t)revpos[e] & 0xFFFFu) | (c[0] << 16));
            pack[2 * e + 1] = make_uint4(c[1] | (c[2] << 16), c[3] | (c[4] << 16), c[5] | (c[6] << 16), c[7] | (c[8] << 16));
        }
    }
}

extern "C" int eps_scan_column_pack(const int64_t *rowptr, const int32_t *col, const int32_t *revpos, const uint32_t *rowrec,
                                    const uint32_t *pptr, const uint32_t *plan, int64_t n_nodes, uint32_t *pack, void *stream)
{
    EPS_REQUIRE(n_nodes >= 0, "eps_scan_column_pack: negative size");
    if (n_nodes == 0) return EPS_OK;
    EPS_REQUIRE(rowptr && col && revpos && rowrec && pptr && plan && pack, "eps_scan_column_pack: null pointer");
    EPS_REQUIRE(((uintptr_t)plan & 15) == 0 && ((uintptr_t)pack & 15) == 0 && ((uintptr_t)rowrec & 127) == 0,
                "eps_scan_column_pack: misaligned table");
    int64_t blocks = (n_nodes + 3) / 4;
    const int64_t cap = (int64_t)eps_num_cus() * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(sp_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rowptr, col, revpos, rowrec, pptr,
                       (const uint4 *)plan, n_nodes, (uint4 *)pack);
    EPS_CHECK_LAUNCH("eps_scan_column_pack");
    return EPS_OK;
}

// ---- launch -------------------------------------------------------------------------------------------------------------
// variant: 0 = 512 threads, 8192-slot table (two workgroups per CU); 1 = 1024 threads, 16384 slots (one per CU);
//          2 = 256 threads, 4096 slots (four per CU).  Also measured (27.8 ms for variant 2 at the time): 256 threads / 8192 slots
//          (two per CU) 38.5 ms, 128 / 4096 (four) 38.2, 128 / 2048 (seven) 50.6, 64 / 2048 (seven) 71.6, 64 / 4096 (four) 61.6 --
//          waves per CU and paths per piece both count, and LDS trades one for the other.  320 threads (five waves per SIMD at 96
//          registers, 16 spilled) with the 4096-slot table: 21.8 ms against 17.9.
static int sp_launch(const int64_t *rowptr, const int32_t *col, const float *val, const int32_t *revpos, const uint32_t *fx32,
                     const float *node_w, const uint16_t *cuts, const uint32_t *wpaths, const uint32_t *ssum, const uint32_t *smax,
                     const uint32_t *pptr, const uint32_t *plan, const uint32_t *heads, const uint32_t *rowrec, const uint32_t *pack,
                     const int32_t *bounds, int64_t n_nodes, int64_t nnz, const int32_t *columns, const uint32_t *colrec, int64_t n_columns, int64_t batch_from,
                     int32_t shift, int32_t variant, eps_survivors *out, uint32_t *status, void *stream);

extern "C" int eps_scan_screen(const int64_t *rowptr, const int32_t *col, const int32_t *revpos, const uint32_t *fx32,
                               const uint16_t *cuts, const uint32_t *wpaths, const uint32_t *ssum_or_null,
                               const uint32_t *smax_or_null, const uint32_t *pptr_or_null, const uint32_t *plan_or_null,
                               const uint32_t *heads_or_null, const uint32_t *rowrec_or_null, const uint32_t *pack_or_null,
                               const int32_t *bounds, int64_t n_nodes, int64_t nnz,
                               const int32_t *columns, const uint32_t *colrec_or_null, int64_t n_columns, int64_t batch_from, int32_t shift,
                               int32_t variant,
                               eps_survivors *out, uint32_t *status, void *stream)
{
    EPS_REQUIRE(!pack_or_null || (plan_or_null && rowrec_or_null && ((uintptr_t)pack_or_null & 15) == 0),
                "eps_scan_screen: the column pack comes with the plan table and the row records it was built from, 16-byte aligned");
    EPS_REQUIRE(n_columns == 0 || n_nodes == 0 || fx32, "eps_scan_screen: null pointer");
    EPS_REQUIRE((ssum_or_null == nullptr) == (smax_or_null == nullptr), "eps_scan_screen: ssum and smax come together");
    EPS_REQUIRE(!heads_or_null || plan_or_null, "eps_scan_screen: a head table comes with the plan table built for it");
    return sp_launch(rowptr, col, nullptr, revpos, fx32, nullptr, cuts, wpaths, ssum_or_null, smax_or_null, pptr_or_null,
                     plan_or_null, heads_or_null, rowrec_or_null, pack_or_null, bounds, n_nodes, nnz, columns, colrec_or_null, n_columns,
                     batch_from, shift, variant, out, status, stream);
}

// The same scan on a SYMMETRIC adjacency with stored values (val[e] == val[mirror of e]); node_w = the float node weights.
extern "C" int eps_scan_screen_weighted(const int64_t *rowptr, const int32_t *col, const float *val, const int32_t *revpos,
                                        const float *node_w, const uint16_t *cuts, const uint32_t *wpaths,
                                        const int32_t *bounds, int64_t n_nodes, int64_t nnz, const int32_t *columns,
                                        int64_t n_columns, int32_t shift, int32_t variant, eps_survivors *out, uint32_t *status,
                                        void *stream)
{
    EPS_REQUIRE(n_columns == 0 || n_nodes == 0 || (val && node_w), "eps_scan_screen_weighted: null pointer");
    return sp_launch(rowptr, col, val, revpos, nullptr, node_w, cuts, wpaths, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                     bounds, n_nodes, nnz, columns, nullptr, n_columns, n_columns, shift, variant, out, status, stream);
}

static const int sp_threads_of[3] = {512, 1024, 256}, sp_bits_of[3] = {13, 14, 12}, sp_per_cu[3] = {2, 1, 4};
// `variant` arguments carry the geometry in their low byte and, in byte 1, an optional limit on the low weight bits a packed /
// 16-bit direct piece may drop: (dmax + 1) << 8, 0 = the default (shift - 8).  The caller lowers it when the smallest weight
// of the graph is small next to 2^(dmax - shift): the bound on (screening score - exact score) per path is (2^d + 1) units, and
// the pre-filter in front of the exact re-scoring is only as sharp as that unit is small next to a path's weight (resource
// allocation on a graph with hubs: weights of 1 / 13 230 -- r05: 13.5 ms of re-scoring at d = 13, a third of it at d = 11).
#define SP_VARIANT_GEOM(v) ((v) & 0xFF)
#define SP_VARIANT_DMAX(v) ((((v) >> 8) & 0xFF) - 1)
// ... and in bit 16 (eps_scan_screen only; r06): packed pieces of single-round columns run as SKETCH pieces (see the kernel)
#define SP_VARIANT_WIDE(v) (((v) >> 24) & 1)          // (plans and launches) bit 24: packed pieces of single-round columns hold up to 2^(bits + 1) paths
#define SP_VARIANT_SKETCH(v) (((v) >> 16) & 0xFF)      // bit 0: on; bits 1..7: slots of the reported-id set (a power of two <= SP_EM; 0 = SP_EM)

// what the planner reads of the geometry: the scan launch and the plan-table launch must agree on it
static void sp_plan_geometry(sp_params &p, const uint16_t *cuts, const uint32_t *wpaths, const uint32_t *ssum, const uint32_t *smax,
                             const int32_t *bounds, int64_t n_nodes, int32_t shift, int32_t variant, int32_t dmax_arg)
{
    const int bits = sp_bits_of[variant];
    p.cuts = cuts;
    p.wpaths = wpaths;
    p.bounds = bounds;
    p.n_nodes = (int32_t)n_nodes;
    p.table_bits = bits;
    p.piece_paths = (1u << bits) / 2u;
    p.ssum = ssum;
    p.smax = smax;
    p.packed_paths = (SP_PACKED_X8 << bits) / 8u;      // (8: load factor 1/2 of the 2^(bits + 1)-word table)
    p.packed_dmax = shift > 8 ? (shift - 8 < 24 ? shift - 8 : 24) : 0;      // (weights keep at least 2^-8 resolution ...
    if (dmax_arg >= 0 && dmax_arg < p.packed_dmax) p.packed_dmax = dmax_arg;   //  ... or what the caller asks for: see SP_VARIANT_DMAX)
    // measured on the ppa-like graph (tools/r03_screen_ab.py): 1500 -> 23.8 ms, 2270 -> 23.5, 4000 -> 23.7; packed_paths 3584 / 4096 /
    // 5120 -> 24.0 / 23.5 / 24.1 ms
    p.mode_ratio = SP_MODE_RATIO;
    p.shift = shift;
}

// The per-graph plan table: every column's pieces, planned once (they depend on the graph, the window tables and -- through the
// sum bounds -- the weight table, not on the bar or the columns of a launch).  One wave per column runs the scan kernel's own
// planner; out == NULL counts (pcount[v] = pieces of column v), else the records go to out[pptr[v] ..].
__global__ __launch_bounds__(256) void sp_plan_kernel(sp_params p, const int64_t *__restrict__ rowptr, uint32_t direct_ids,
                                                      uint32_t *__restrict__ pcount, const uint32_t *__restrict__ pptr, uint4 *__restrict__ out,
                                                      uint32_t *__restrict__ d_used)
{
    const int lane = threadIdx.x & 63;
    const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= p.n_nodes) return;
    const int32_t dv = (int32_t)(rowptr[v + 1] - rowptr[v]);
    int np = 0;
    if (dv > 0 && v > 0) {
        const int32_t my_bound = p.bounds[lane <= SP_M ? lane : SP_M];
        const uint32_t pwk = lane < SP_M ? p.wpaths[(size_t)v * SP_M + lane] : 0u;
        const int32_t nbk = lane >= 1 && lane <= SP_M ? (int32_t)p.cuts[(size_t)v * SP_M + lane - 1] : 0;
        const uint32_t pb = out ? pptr[v] : 0u;
        np = sp_plan_column(p, (int32_t)v, dv, lane, my_bound, pwk, nbk, direct_ids,
                            [&](int i, int k0, int k1, int32_t lo, int32_t hi, uint32_t info, uint32_t pq, int32_t na, int32_t nb) {
                                if (out && lane == 0) {
                                    out[pb + (uint32_t)i] = make_uint4(info, (uint32_t)k0 | ((uint32_t)k1 << 8) | (pq << 16),
                                                                       (uint32_t)na | ((uint32_t)nb << 16), (uint32_t)lo);
                                    // (look first: two million atomics on one word would take 20 ms)
                                    if (d_used && (info & 0x40000000u) && (pq & 0xFFu) > __atomic_load_n(d_used, __ATOMIC_RELAXED))
                                        atomicMax(d_used, pq & 0xFFu);
                                }
                            });
    }
    if (!out && lane == 0) pcount[v] = (uint32_t)np;
}

extern "C" int eps_scan_plan(const int64_t *rowptr, const uint16_t *cuts, const uint32_t *wpaths, const uint32_t *ssum_or_null,
                             const uint32_t *smax_or_null, const uint32_t *heads_or_null, const int32_t *bounds, int64_t n_nodes,
                             int32_t shift, int32_t variant, uint32_t *pcount, const uint32_t *pptr_or_null, uint32_t *plan_or_null,
                             uint32_t *d_used_or_null, void *stream)
{
    EPS_REQUIRE(n_nodes >= 0 && n_nodes < (1ll << 31), "eps_scan_plan: bad size");
    if (d_used_or_null && hipMemsetAsync(d_used_or_null, 0, sizeof(uint32_t), (hipStream_t)stream) != hipSuccess) {
        eps_set_error("eps_scan_plan: cannot clear d_used");
        return EPS_ELAUNCH;
    }
    if (n_nodes == 0) return EPS_OK;
    EPS_REQUIRE(rowptr && cuts && wpaths && bounds, "eps_scan_plan: null pointer");
    EPS_REQUIRE((ssum_or_null == nullptr) == (smax_or_null == nullptr), "eps_scan_plan: ssum and smax come together");
    EPS_REQUIRE((pptr_or_null == nullptr) == (plan_or_null == nullptr), "eps_scan_plan: pptr and plan come together");
    EPS_REQUIRE(pptr_or_null || pcount, "eps_scan_plan: nowhere to put the counts");
    const int32_t dmax_arg = SP_VARIANT_DMAX(variant);
    const bool wide_arg = SP_VARIANT_WIDE(variant) != 0;
    variant = SP_VARIANT_GEOM(variant);
    EPS_REQUIRE(shift >= 0 && shift <= 40 && variant >= 0 && variant <= 2, "eps_scan_plan: bad shift / variant");
    EPS_REQUIRE(((uintptr_t)plan_or_null & 15) == 0, "eps_scan_plan: plan must be 16-byte aligned");
    sp_params p;
    memset(&p, 0, sizeof p);
    sp_plan_geometry(p, cuts, wpaths, ssum_or_null, smax_or_null, bounds, n_nodes, shift, variant, dmax_arg);
    if (wide_arg && ssum_or_null) {                      // (a plan for a sketch launch: eps_scan_screen with the same bit AND the sketch bit)
        p.wide_paths = 2u << sp_bits_of[variant];        // (= SP_UBITS at the 256-thread geometry: one range of the unit bitmap)
        if (p.wide_paths > SP_UBITS) p.wide_paths = SP_UBITS;
        p.wide_rows = sp_threads_of[variant];
        // (a sketch path costs about what a direct path costs, so a piece's fixed cost weighs more against the difference: the
        //  longer packed run wins more often -- 5000 / 10000 / 20000: 6.47 / 6.47 / 6.52 ms against 6.75 at 2270 on the ppa-like graph)
        p.mode_ratio = SP_MODE_RATIO_WIDE;
    }
    p.heads = (const uint2 *)heads_or_null;
    hipLaunchKernelGGL(sp_plan_kernel, dim3((unsigned)((n_nodes + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p, rowptr,
                       2u << sp_bits_of[variant], pcount, pptr_or_null, (uint4 *)plan_or_null, d_used_or_null);
    EPS_CHECK_LAUNCH("eps_scan_plan");
    return EPS_OK;
}

// What a plan costs in RE-WALKED paths: a two-word hash piece (kind bits 00) whose paths + known edges exceed a piece's capacity is
// walked in `parts` hash-partitioned passes (the scan kernel's rule: next power of two of the quotient, doubled).
// out[0] += paths x (parts - 1) over such pieces, out[1] += paths over all pieces.
__global__ __launch_bounds__(256) void sp_rewalk_kernel(const uint4 *__restrict__ plan, int64_t n_rec, uint32_t cap,
                                                        unsigned long long *__restrict__ out)
{
    unsigned long long re = 0ull, all = 0ull;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rec; i += (int64_t)gridDim.x * blockDim.x) {
        const uint4 r = plan[i];
        const uint32_t paths = r.x & 0x3FFFFFFFu;
        all += paths;
        if ((r.x >> 30) == 0u) {
            const uint32_t keys = paths + ((r.z >> 16) - (r.z & 0xFFFFu));
            if (keys > cap) {
                const uint32_t q = (keys + cap - 1u) / cap;
                const uint32_t parts = 2u << (32 - __clz((int)(q - 1u)));
                re += (unsigned long long)paths * (parts - 1u);
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        re += __shfl_xor(re, d);
        all += __shfl_xor(all, d);
    }
    if ((threadIdx.x & 63) == 0) {
        if (re) atomicAdd(&out[0], re);
        if (all) atomicAdd(&out[1], all);
    }
}

extern "C" int eps_scan_plan_rewalk(const uint32_t *plan, int64_t n_rec, int32_t variant, unsigned long long *out2, void *stream)
{
    EPS_REQUIRE(n_rec >= 0 && out2 && variant >= 0 && variant <= 2, "eps_scan_plan_rewalk: bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(out2, 0, 2 * sizeof(unsigned long long), s) != hipSuccess) {
        eps_set_error("eps_scan_plan_rewalk: cannot clear the result");
        return EPS_ELAUNCH;
    }
    if (n_rec == 0) return EPS_OK;
    EPS_REQUIRE(plan && ((uintptr_t)plan & 15) == 0, "eps_scan_plan_rewalk: plan must be 16-byte aligned");
    int64_t blocks = (n_rec + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(sp_rewalk_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const uint4 *)plan, n_rec,
                       (1u << sp_bits_of[variant]) / 2u, out2);
    EPS_CHECK_LAUNCH("eps_scan_plan_rewalk");
    return EPS_OK;
}

static int sp_launch(const int64_t *rowptr, const int32_t *col, const float *val, const int32_t *revpos, const uint32_t *fx32,
                     const float *node_w, const uint16_t *cuts, const uint32_t *wpaths, const uint32_t *ssum, const uint32_t *smax,
                     const uint32_t *pptr, const uint32_t *plan, const uint32_t *heads, const uint32_t *rowrec, const uint32_t *pack,
                     const int32_t *bounds, int64_t n_nodes, int64_t nnz, const int32_t *columns, const uint32_t *colrec, int64_t n_columns, int64_t batch_from,
                     int32_t shift, int32_t variant, eps_survivors *out, uint32_t *status, void *stream)
{
    EPS_REQUIRE(n_nodes >= 0 && n_columns >= 0 && nnz >= 0, "eps_scan_screen: negative size");
    EPS_REQUIRE(status, "eps_scan_screen: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(status, 0, sizeof(uint32_t), s) != hipSuccess) {
        eps_set_error("eps_scan_screen: cannot clear the status word");
        return EPS_ELAUNCH;
    }
    if (n_columns == 0 || n_nodes == 0) return EPS_OK;
    EPS_REQUIRE(rowptr && col && revpos && cuts && wpaths && bounds && columns && out, "eps_scan_screen: null pointer");
    EPS_REQUIRE(nnz < (1ll << 30), "eps_scan_screen: col[] is addressed with 32-bit byte offsets (nnz < 2^30)");
    EPS_REQUIRE(n_nodes < (1ll << 31) && n_columns < (1ll << 31), "eps_scan_screen: too many nodes / columns");
    const uint32_t sketch_arg = (uint32_t)SP_VARIANT_SKETCH(variant);
    EPS_REQUIRE(!SP_VARIANT_WIDE(variant) || ((sketch_arg & 1u) && plan && !val && SP_VARIANT_GEOM(variant) == 2),
                "eps_scan_screen: a plan with wide packed pieces (variant bit 24) is for sketch launches (bit 16) of the 256-thread geometry");
    const int32_t dmax_arg = SP_VARIANT_DMAX(variant);
    variant = SP_VARIANT_GEOM(variant);
    EPS_REQUIRE(shift >= 0 && shift <= 40 && variant >= 0 && variant <= 2, "eps_scan_screen: bad shift / variant");
    EPS_REQUIRE(((uintptr_t)cuts & 15) == 0, "eps_scan_screen: cuts must be 16-byte aligned");
    EPS_REQUIRE(((uintptr_t)colrec & 15) == 0 && ((uintptr_t)rowrec & 127) == 0, "eps_scan_screen: misaligned column / row records");
    EPS_REQUIRE(!colrec || plan, "eps_scan_screen: column records come with the plan table they were built from");
    EPS_REQUIRE((pptr == nullptr) == (plan == nullptr) && ((uintptr_t)plan & 15) == 0, "eps_scan_screen: pptr and plan come together, 16-byte aligned");
    const int T = sp_threads_of[variant], bits = sp_bits_of[variant];
    unsigned int *counter = nullptr;
    const int rc = eps_take_counter(&counter, s, "eps_scan_screen");
    if (rc) return rc;
    sp_params p;
    memset(&p, 0, sizeof p);
    p.rowptr = rowptr;
    p.col = col;
    p.revpos = revpos;
    p.fx32 = fx32;
    p.val = val;
    p.node_w = node_w;
    p.up = ldexpf(1.0f, shift) * (1.0f + ldexpf(1.0f, -20));
    sp_plan_geometry(p, cuts, wpaths, ssum, smax, bounds, n_nodes, shift, variant, dmax_arg);
    p.pptr = pptr;
    p.plan = (const uint4 *)plan;
    p.heads = (const uint2 *)heads;
    p.rowrec = rowrec;
    p.pack = (const uint4 *)pack;
    p.colrec = (const uint4 *)colrec;
    p.columns = columns;
    p.n_columns = (int32_t)n_columns;
    p.batch_from = batch_from < 0 || batch_from > n_columns ? (uint32_t)n_columns : (uint32_t)batch_from;
    p.sketch = val || !(sketch_arg & 1u) ? 0u : sketch_arg;
    p.col_bytes = (uint32_t)(nnz * 4);
    p.scale = ldexpf(1.0f, -shift);
    p.next_col = counter;
    p.out = out;
    p.status = status;
    int64_t blocks = (int64_t)eps_num_cus() * sp_per_cu[variant];
    if (blocks > n_columns) blocks = n_columns;
    const size_t lds = ((size_t)(2 << bits) + 4 * (size_t)(T + 1) + 8) * 4 + (SP_UBITS / 32) * 6 + 32;
    void (*kern)(sp_params) =
        val ? (variant == 0 ? scan_piece_kernel<512, true, false, false> : variant == 1 ? scan_piece_kernel<1024, true, false, false>
                                                                                       : scan_piece_kernel<256, true, false, false>)
            : (variant == 0 ? scan_piece_kernel<512, false, false, false> : variant == 1 ? scan_piece_kernel<1024, false, false, false>
                                                                                         : scan_piece_kernel<256, false, false, false>);
    // (the step's main launch: every table present -> the body compiled for exactly that; with the column pack, its set-up as well)
    // (EPS_SCAN_GENERIC=1 in the environment keeps the generic body for same-box A/Bs: tools/r05_heads_ab.py)
    static const bool generic_only = [] { const char *e = getenv("EPS_SCAN_GENERIC"); return e && e[0] == '1'; }();
    const bool full = !generic_only && !val && variant == 2 && plan && colrec && rowrec && heads && ssum;
    EPS_REQUIRE(!pack || full || generic_only, "eps_scan_screen: the column pack serves the main launch only (variant 2, plan, records, heads, sum bounds)");
    if (full) kern = pack ? scan_piece_kernel<256, false, true, true> : scan_piece_kernel<256, false, true, false>;
    if (p.sketch && variant == 2)      // (the 256-thread geometry only: the one the step's main launch runs in)
        kern = full ? (pack ? scan_piece_kernel<256, false, true, true, true> : scan_piece_kernel<256, false, true, false, true>)
                    : scan_piece_kernel<256, false, false, false, true>;
    else
        p.sketch = 0u;
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        eps_set_error("eps_scan_screen: cannot reserve %zu bytes of LDS", lds);
        return EPS_ELAUNCH;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(T), lds, s, p);
    EPS_CHECK_LAUNCH("eps_scan_screen");
    return EPS_OK;
}

// (one empty kernel per translation unit: launching it makes the HIP runtime load this unit's code object -- eps_warm_up)
__global__ void scan_pieces_warm_kernel() {}
extern "C" void eps_warm_scan_pieces(void *stream) { hipLaunchKernelGGL(scan_pieces_warm_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream); }
