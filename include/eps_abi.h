/*
 * eps_abi.h -- C ABI of libeps_hip.so, the MI355X (gfx950) edge-scoring engine.
 *
 * This is the drop-in boundary for the Filter-and-Rank hot path of CUAI/Edge-Proposal-Sets.
 * The reference has no FFI of its own for this path: the work is done by third-party
 * kernels (SciPy sparsetools, torch_sparse, cuBLAS) behind plain Python call sites.  Each
 * entry point below names the reference call site(s) (file:line under /root/reference)
 * whose arithmetic it replaces.  The Python host (edge-proposal-sets_amd/) binds these
 * with ctypes and keeps the reference's own signatures on top.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer borrowed from the caller (the caller keeps the
 *     buffer alive until the stream has been synchronised); nothing is allocated inside;
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream); all
 *     calls are asynchronous on it and re-entrant; no global mutable state;
 *   - return value: EPS_OK (0) or a negative EPS_E* code; the message for the calling
 *     thread's last failure is returned by eps_last_error();
 *   - CSR graphs: rowptr int64[N+1], col int32[nnz] strictly ascending inside a row
 *     (coalesced), val float32[nnz] or NULL (NULL == every stored value is 1.0f).
 */
#ifndef EPS_ABI_H
#define EPS_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EPS_ABI_VERSION 7   /* 7: the tail of the filter step with device-side sizes -- eps_score_hist / _pick_compact, eps_radix_sort_by_u / _rows, eps_rescore_runs_dev; eps_scan_screen takes a column pack, eps_scan_column_pack, eps_expand_unit_list, eps_reverse_positions_sorted (r06); 6: skipped heads -- eps_scan_heads / _hub_rows / _refine; eps_scan_window_paths / _plan / _screen take a head table (r05); 2: eps_col_sums / eps_node_weights_f64 signatures (r02); 3: 64-bit survivor count, eps_scan_* (r03); 4: eps_scan_screen takes ssum / smax; 5: eps_select_compact, eps_scan_screen marks unused slots itself (r04) */

#define EPS_OK 0
#define EPS_EINVAL (-1)   /* bad argument (null pointer, negative size, unsupported shape) */
#define EPS_ELAUNCH (-2)  /* HIP launch / runtime error */
#define EPS_ENODEV (-3)   /* no usable gfx950 device */

/* node-weight modes for eps_node_weights */
#define EPS_W_AA 0 /* 1/log(colsum), inf -> 0   (adamic_utils.py:15-16)     */
#define EPS_W_RA 1 /* 1/colsum,      inf -> 0   (train_and_eval.py:203-204) */

int eps_version(void);
const char *eps_last_error(void);

/* eps_warm_up: load every code object of the library now (one empty kernel per translation unit on a private stream, waited
 * for) instead of at the first use of each -- a fresh process pays 10-30 ms per larger unit, and filter.py is one process per
 * graph (submit_job.py:20-21).  Thread-safe; meant for a background thread at start-up.  Blocking. */
int eps_warm_up(void);

/* Number of compute units / name of the current device (host query; for launch sizing
 * reports in bench.py).  name may be NULL. */
int eps_device_info(int *n_cu, char *name, int name_len);

/* ---- K2: per-node weight table ------------------------------------------------------
 * Replaces `A.sum(0)`, `1/np.log(.)`, inf->0 (adamic_utils.py:15-16) and `1/A.sum(axis=0)`,
 * inf->0 (train_and_eval.py:203-204).
 * eps_col_sums: colsum[c] = sum_r A[r,c], accumulated in FLOAT64 (exact, hence order-independent, for integer-valued
 *               A up to 2^53): colsum_f64 (n_cols doubles, cleared here) receives the sums, colsum_f32 (optional) the
 *               same sums rounded once to float32 -- what SciPy's float32 `A.sum(0)` holds for such A.
 * eps_node_weights: w[i] = mode==AA ? 1/logf(colsum[i]) : 1/colsum[i]; +-inf -> 0   (float32 sums, float32 math).
 * eps_node_weights_f64: the same in float64 from the float64 sums: filter.py:130-141, where A is int64 and the
 *               reference's math is float64. */
int eps_col_sums(const int64_t *rowptr, const int32_t *col, const float *val, int64_t n_rows,
                 int64_t n_cols, double *colsum_f64, float *colsum_f32, void *stream);
int eps_node_weights(const float *colsum, int64_t n, int mode, float *w, void *stream);
int eps_node_weights_f64(const double *colsum, int64_t n, int mode, double *w, void *stream);

/* ---- K1/K3: pair scores by CSR neighbour-list intersection ------------------------------
 * Replaces `np.sum(A[src].multiply(A_[dst]), 1)` (adamic_utils.py:22, train_and_eval.py:212)
 * and `adj[e0] (.) adj[e1]` + sparse row-sum (models.py:536-542).  For each pair p:
 *   count[p] = |N(u) ^ N(v)|                                   (int32, exact)
 *   cn[p]    = sum_w A[u,w] * A[v,w]                           (float32; == count if val NULL)
 *   wsum[p]  = sum_w A[u,w] * (A[v,w] * node_w[w])             (float32; AA / RA)
 * Any of the three outputs may be NULL (skipped); wsum needs node_w != NULL.
 * u, v: int32[n_pairs] node ids in [0, n_nodes).  One wavefront scores one pair at a time. */
int eps_pair_scores(const int64_t *rowptr, const int32_t *col, const float *val,
                    const float *node_w, int64_t n_nodes, const int32_t *u, const int32_t *v,
                    int64_t n_pairs, int32_t *count, float *cn, float *wsum, void *stream);

/* float64 accumulate with float64 node weights; wsum is float64 (filter.py:141 RA path). */
int eps_pair_scores_f64(const int64_t *rowptr, const int32_t *col, const float *val,
                        const double *node_w, int64_t n_nodes, const int32_t *u,
                        const int32_t *v, int64_t n_pairs, int32_t *count, double *wsum,
                        void *stream);

/* Column-run variant of the two calls above, same arguments and results, for pair lists in the
 * reference's candidate order (filter.py:96-109: column-major, long runs of equal v): the
 * neighbourhood of the run's v is kept as an LDS bitmap and every pair costs one read of row u.
 * Correct for ANY list (pairs off the chunk's run take an in-place search) but only fast when v
 * has runs of hundreds of pairs or more; the Python host picks it from the run statistics. */
int eps_pair_scores_grouped(const int64_t *rowptr, const int32_t *col, const float *val,
                            const float *node_w, int64_t n_nodes, const int32_t *u,
                            const int32_t *v, int64_t n_pairs, int32_t *count, float *cn,
                            float *wsum, void *stream);
int eps_pair_scores_grouped_f64(const int64_t *rowptr, const int32_t *col, const float *val,
                                const double *node_w, int64_t n_nodes, const int32_t *u,
                                const int32_t *v, int64_t n_pairs, int32_t *count, double *wsum,
                                void *stream);

/* ---- K7 (+K1): fused candidate generation and scoring of the filter stage ------------------
 * Replaces filter.py:96-109 (`A2 = adj_t @ adj_t`, remove diagonal, zero known edges, nonzeros in
 * column-major order) together with the heuristic scoring that follows it (adamic_utils.py:13-25,
 * train_and_eval.py:195-216, models.py:536-542): one expansion of the 2-hop paths of a column
 * yields every candidate of the column with its common-neighbour count and
 * sum_w A[u,w]*(A[v,w]*node_w[w]).  The adjacency must be symmetric (rank.py:33).  Id spaces of
 * up to eps_expand_max_nodes() ids fit the LDS bitmap at once; wider ones are expanded in id windows
 * of that size (each window walks only the row segments inside it), same outputs, same order.
 *   col_order (both calls; may be NULL): a permutation of [0, v_hi - v_lo) -- the order in which the
 *                     columns v_lo + col_order[i] are handed to the workgroups.  Results do not
 *                     depend on it; heaviest-first shortens the tail of a launch.
 *   eps_expand_count: cand_count[v - v_lo] = number of 2-hop non-edges (u, v), v in [v_lo, v_hi).
 *   eps_expand_fill : colptr = exclusive prefix of cand_count (int64[n_cols+1], device); writes
 *                     cand_u (ascending inside a column == the reference's order), cand_v (column
 *                     id per candidate; optional) and, each optional, the common-neighbour count
 *                     cn (int32) and the weighted sum score (float32); neither needs zeroing.
 *                     cn / score need `workspace` (device, 8-byte aligned, contents arbitrary):
 *                     eps_expand_workspace_bytes(P) bytes, P >= the largest number of two-hop
 *                     paths sum_{w in N(v)} deg(w) of any column in the range.  The paths of a
 *                     column are binned there by candidate-rank tile, then summed tile by tile in
 *                     LDS in 2^-40 fixed point: integer addition makes the sums independent of the
 *                     arrival order (P = 0 without cn / score).  colptr may also be an UPPER-BOUND
 *                     layout -- any non-decreasing int64[n_cols+1] whose segments are at least as
 *                     long as the columns' candidate counts, e.g. the prefix of the two-hop path
 *                     counts, which needs no eps_expand_count pass: cand_count (optional) then
 *                     receives the real count per column and the rest of each segment is padded
 *                     with cand_u = -1, score = -inf, cn = 0, so the arrays stay in candidate
 *                     order.  After the launch the first 4 bytes of the workspace are 0, or
 *                     non-zero if P (bit 0) or a colptr segment (bit 1) was too small, or a sum
 *                     left the accumulators' range |sum| < 2^23 (bit 2; terms are assumed non-negative): outputs
 *                     invalid. */
/* Optional top-K cut of eps_expand_fill, a DEVICE-resident record: every candidate whose score exceeds `threshold`
 * is reported as (pos, val) -- pos = its index in the launch's candidate arrays -- in arrival order (sort by pos to
 * restore candidate order).  count (zeroed by the caller) may end above capacity: the list is then incomplete.
 * With a cut the score array itself is optional. */
typedef struct eps_score_cut {
    float threshold;
    uint32_t capacity;
    uint32_t count;
    uint32_t reserved;
    int64_t *pos;
    float *val;
} eps_score_cut;

int eps_expand_max_nodes(void);
int eps_expand_count(const int64_t *rowptr, const int32_t *col, int64_t n_nodes, int64_t v_lo,
                     int64_t v_hi, const int32_t *col_order, int64_t *cand_count, void *stream);
int eps_expand_fill(const int64_t *rowptr, const int32_t *col, const float *val,
                    const float *node_w, int64_t n_nodes, int64_t v_lo, int64_t v_hi,
                    const int32_t *col_order, const int64_t *colptr, int64_t *cand_count,
                    int32_t *cand_u, int32_t *cand_v, int32_t *cn, float *score,
                    eps_score_cut *cut, void *workspace, int64_t workspace_bytes, void *stream);
/* eps_expand_fill with an explicit tile size (candidate ranks summed per LDS pass; 0 = the default 8192): results do not
 * depend on it -- small graphs are single-tile at the default, so the parity tests force several tiles per column. */
int eps_expand_fill_tiled(const int64_t *rowptr, const int32_t *col, const float *val,
                          const float *node_w, int64_t n_nodes, int64_t v_lo, int64_t v_hi,
                          const int32_t *col_order, const int64_t *colptr, int64_t *cand_count,
                          int32_t *cand_u, int32_t *cand_v, int32_t *cn, float *score,
                          eps_score_cut *cut, void *workspace, int64_t workspace_bytes,
                          int32_t tile_ranks, void *stream);
int64_t eps_expand_workspace_bytes(int64_t max_col_paths);

/* ---- K7+K1+K8: threshold scan of the whole candidate set (filter.py:96-142 + :160-161 under --keep_top) ------------
 * Computes the score of EVERY 2-hop non-edge of the given columns like eps_expand_fill does (same 2^-40 fixed-point
 * sums, bit-identical float32 scores) but writes neither the candidate list nor the score array: only the candidates
 * whose score exceeds out->threshold are reported.  Requires a SYMMETRIC adjacency with UNIT values (val == NULL;
 * filter.py's is: rank.py:33-35 for every dataset but collab) and exploits the symmetry: column v expands only the
 * endpoints u < v, so each unordered candidate pair {u, v} is scored once and reported once, as
 * key = (v << 32) | u with u < v; the caller mirrors it (the scores of (u,v) and (v,u) are equal term by term).
 *   eps_reverse_positions: revpos[e] for entry e of row v, w = col[e]: number of entries of row w below v (the position
 *                     of v in row w).  Per-graph table, int32[nnz].  The same pass can give half_paths[v] = sum of revpos
 *                     over row v (int64[N]: the two-hop half paths of column v = its work in the scan and the bound of
 *                     its survivors) and *asymmetric (device word, cleared by the call) = 1 if some entry (v, w) has no
 *                     mirror (w, v) -- the half scheme needs a symmetric pattern.
 *   eps_fixed_weights: fixw[i] = round(node_w[i] * 2^40): the per-node weights (eps_node_weights; all ones for the
 *                     common-neighbour count of models.py:536-542) in the accumulators' fixed point, int64[N].
 *   eps_filter_scan : columns = int32[n_columns] column ids in hand-out order (any subset, any order: a heaviest-first
 *                     order shortens the tail; a sample of columns estimates the bar; a rank's shard under
 *                     torch.distributed).  out is DEVICE-resident: threshold and capacity set by the caller, count
 *                     zeroed, key[] pre-filled with -1.  Slots are handed out in chunks, so after the launch the first
 *                     min(count, capacity) slots hold the survivors interleaved with untouched (-1) slots;
 *                     count > capacity means survivors were dropped.  max_degree: the longest row of the graph.
 *                     workspace: eps_filter_scan_workspace_bytes(max_degree) bytes of scratch (bucket records +
 *                     per-workgroup weight tables), 16-byte aligned. */
typedef struct eps_survivors {
    float threshold;
    uint32_t capacity;
    unsigned long long count;        /* slots handed out (64-bit: a list that overflows cannot wrap back under capacity) */
    int64_t *key;
    float *val;
    unsigned long long n_candidates; /* out (zeroed by the caller): unordered candidate pairs scored by the launch */
} eps_survivors;

int64_t eps_filter_scan_max_nodes(void);
int64_t eps_filter_scan_workspace_bytes(int64_t max_degree);
/* Id spaces wider than the LDS bitmap are scanned in id windows: eps_filter_scan_windows reports the window size / count
 * the launch will use for n_nodes ids, eps_row_window_splits fills the per-graph table it then needs:
 * splits[k * n_nodes + w] = number of entries of row w below id (k + 1) * win_ids, k = 0 .. n_win - 2 (int32). */
int eps_filter_scan_windows(int64_t n_nodes, int64_t *win_ids, int64_t *n_win);
int eps_row_window_splits(const int64_t *rowptr, const int32_t *col, int64_t n_nodes, int64_t win_ids,
                          int64_t n_win, int32_t *splits, void *stream);
int eps_reverse_positions(const int64_t *rowptr, const int32_t *col, int64_t n_nodes, int32_t *revpos,
                          int64_t *half_paths_or_null, uint32_t *asymmetric_or_null, void *stream);
/* ---- per-graph tables built by the library (r04; csrc/graph_prep.hip): filter.py runs once per graph (submit_job.py:20-21), so
 * these are on the critical path of every run --------------------------------------------------------------------------------
 * eps_relabel_graph: the copy of a coalesced CSR graph under a node permutation -- row i of the copy = row perm[i] of the graph
 *   with every id x replaced by inv[x] (inv[perm[i]] = i, int32), rows sorted ascending again (a segmented radix sort over the
 *   id_bits bits of the ids; values travel with their entries).  new_rowptr = the caller's prefix sum of the degrees in perm
 *   order (int64[n_nodes + 1]).  workspace: eps_relabel_graph_workspace_bytes(n_nodes, nnz, with_values) bytes, 256-byte aligned.
 * eps_reverse_positions_symmetric: what eps_reverse_positions reports (revpos, half_paths = its row sums, *asymmetric) for a
 *   SYMMETRIC pattern with half the searches: entry (v, w), w > v, looks v up in row w and writes both its own position and --
 *   at the place it found v -- its mirror's.  On any other pattern *asymmetric comes back 1 and revpos is not usable.
 *   stats (optional, 3 DEVICE uint64): largest degree, largest half_paths[v], sum of half_paths -- the scalars the scan reads.
 * eps_score_bound: *bound (DEVICE double) = max over rows v of sum_w |A[v,w]| |node_w[w]| max_u |A[u,w]| in float64 (without
 *   stored values: sum of |node_w| over the row; node_w NULL: the degree) -- an upper bound of every fused score of the graph.
 *   workspace: n_cols uint32 with stored values, else may be NULL. */
int64_t eps_relabel_graph_workspace_bytes(int64_t n_nodes, int64_t nnz, int32_t with_values);
int eps_relabel_graph(const int64_t *rowptr, const int32_t *col, const float *val_or_null, const int64_t *perm, const int32_t *inv,
                      const int64_t *new_rowptr, int64_t n_nodes, int64_t nnz, int32_t id_bits, int32_t *out_col,
                      float *out_val_or_null, void *workspace, int64_t workspace_bytes, void *stream);
int eps_reverse_positions_symmetric(const int64_t *rowptr, const int32_t *col, int64_t n_nodes, int64_t nnz, int32_t *revpos,
                                    int64_t *half_paths, uint32_t *asymmetric, unsigned long long *stats_or_null, void *stream);
/* eps_reverse_positions_sorted (r06): the same outputs without a search -- a stable radix sort of the entry indices by column id
 * puts the mirror of the CSR's j-th entry at place j (symmetric patterns), one pass scatters the positions and checks the
 * pattern on the way; id_bits = bits of the largest id; workspace: eps_reverse_positions_sorted_workspace_bytes() bytes,
 * 256-byte aligned.  (filter.py:96-109 leaves the adjacency as it loaded it; this table is the scan's: the position of v in
 * row w for every stored (v, w) -- what lets column v walk only the endpoints u < v of its rows.) */
int64_t eps_reverse_positions_sorted_workspace_bytes(int64_t n_nodes, int64_t nnz);
int eps_reverse_positions_sorted(const int64_t *rowptr, const int32_t *col, int64_t n_nodes, int64_t nnz, int32_t id_bits,
                                 int32_t *revpos, int64_t *half_paths, uint32_t *asymmetric, unsigned long long *stats_or_null,
                                 void *workspace, int64_t workspace_bytes, void *stream);
/* eps_node_order: order[i] (int32) = the node with the i-th largest key, ties by ascending id (a stable descending radix sort)
 * -- keys = the degrees (rowptr given: the hubs-first labels) or keys[] (int64, non-negative: the scan's heaviest-first column
 * order from the half paths).  Optional by-products for eps_relabel_graph: perm64[i] = order[i], inv32[order[i]] = i,
 * new_rowptr[0 .. n] = exclusive prefix of the keys in that order.  workspace: eps_node_order_workspace_bytes(n) bytes, 256-byte
 * aligned. */
int64_t eps_node_order_workspace_bytes(int64_t n);
int eps_node_order(const int64_t *rowptr_or_null, const int64_t *keys_or_null, int64_t n, int32_t *order_or_null,
                   int64_t *perm64_or_null, int32_t *inv32_or_null, int64_t *new_rowptr_or_null, void *workspace,
                   int64_t workspace_bytes, void *stream);
int eps_score_bound(const int64_t *rowptr, const int32_t *col, const float *val_or_null, const float *node_w_or_null,
                    int64_t n_rows, int64_t n_cols, int64_t nnz, double *bound, void *workspace, void *stream);
int eps_fixed_weights(const float *node_w, int64_t n, int64_t *fixw, void *stream);
int eps_filter_scan(const int64_t *rowptr, const int32_t *col, const int32_t *revpos, const int64_t *fixw,
                    const int32_t *splits_or_null, int64_t n_nodes, int64_t nnz, int64_t max_degree,
                    const int32_t *columns, int64_t n_columns, eps_survivors *out, void *workspace,
                    int64_t workspace_bytes, void *stream);

/* ---- the same threshold scan in ONE pass over the two-hop paths (r03; csrc/scan_pieces.hip) --------------------------------
 * Reports what eps_filter_scan reports -- key = (v << 32) | u, u < v, of every 2-hop non-edge of the given columns whose
 * score can exceed out->threshold (filter.py:96-142 + :160-161 under --keep_top) -- but a column is scored in PIECES (runs of
 * id windows of its endpoints) whose candidates each own a slot of an LDS table, so every path is read once and costs one
 * table update.  The table holds 32-bit SCREENING sums of weights rounded UP to 2^-shift fixed point: an upper bound of the
 * exact 2^-40 fixed-point score, so no candidate above the bar is lost and a few just below it pass as well; out->val holds
 * the screening score (sum * 2^-shift).  The caller re-scores the survivors exactly (eps_rescore_runs; for a graph with
 * stored values eps_rescore_weighted: int64 sums of the 2^-40 fixed-point terms, order-independent, bit-identical to
 * eps_filter_scan's sums) and drops those <= the bar.
 * Needs what eps_filter_scan needs (symmetric adjacency, revpos) plus max degree < 65536 and weights >= 0; unit-valued
 * adjacencies go through eps_scan_screen, adjacencies with stored values through eps_scan_screen_weighted.
 *   eps_scan_windows      : M, the number of id windows per graph (32).
 *   eps_scan_cuts         : cuts[w * M + k] = entries of row w with id < bounds[k + 1] (uint16; 16-byte aligned), for the
 *                           caller's window boundaries bounds[0 .. M] (bounds[0] = 0, bounds[M] = n_nodes, non-decreasing;
 *                           windows of equal stored-entry mass balance the pieces).  Per-graph table.
 *   eps_scan_window_paths : wpaths[v * M + k] = two-hop half paths of column v that end in id window k (uint32; exact, from
 *                           the cut table).  Per-graph table; mandatory since v6 (a launch used to sum them itself without).
 *   eps_scan_screen_weights: fx32[i] = max(1, ceil(fixw[i] / 2^(40 - shift))); *bad (device word, cleared by the call):
 *                           bit 1 a negative weight, bit 2 a weight that does not fit 32 bits.  shift must keep every
 *                           screening sum of the graph below 2^31 (bit 31 of a table word flags a known edge) (the caller's score bound: eps_amd.scan.screen_shift).
 *   eps_scan_screen       : out as for eps_filter_scan (slots are handed out in chunks: 8192 per request under a bar, 2 x table
 *                           slots -- 8192 / 16384 / 32768 for variant 2 / 0 / 1 -- without one, where a piece's whole yield
 *                           survives; a survivor beyond its workgroup's reservation takes a slot of its own);
 *                           ssum / smax (optional, both or neither; per graph and weight table): ssum[v] = sum of fx32 over
 *                           row v (any pair's screening sum is at most the smaller of its two endpoints' ssum), smax[k] =
 *                           the largest ssum among the ids >= bounds[k] (k = 0 .. M; smax[M] = 0).  With them, pieces whose
 *                           sums provably fit next to their key bits keep key and sum in ONE table word (twice the
 *                           candidates per piece); their weights drop up to shift - 8 low bits, rounded up (still an upper
 *                           bound; out->val stays in units of 2^-shift);
 *                           batch_from: columns[0 .. batch_from) are handed to the workgroups one at a time, the rest eight per
 *                           draw (the hand-out is an atomic on one word, ~11 ns each: light columns -- give the list
 *                           heaviest first -- would wait for it); n_columns or more, or negative: one at a time throughout;
 *                           variant (low byte) 0: 512 threads / 8192-slot table (2 workgroups per CU), 1: 1024 / 16384 (1), 2: 256 /
 *                           4096 (4); byte 1 of `variant` (eps_scan_plan and eps_scan_screen alike): 0, or dmax + 1 = a limit on
 *                           the low weight bits a packed / 16-bit direct piece may drop (default shift - 8).  Lower it when the
 *                           graph's smallest weight is small: the screening score exceeds the exact one by up to 2^d + 1 units
 *                           per path, and the pre-filter before the exact re-scoring is as sharp as that is small next to a weight;
 *                           bit 16 of `variant` (r06; eps_scan_screen, the 256-thread geometry, unit-valued graphs, a launch under a
 *                           bar): packed pieces of single-round columns run as SKETCH pieces -- no keys: every path adds its
 *                           weight to one slot of each of two half tables (two hashes, non-returning adds), a candidate's sum
 *                           is at most the smaller of its two slots (an UPPER bound: what a screen needs and no more, so the
 *                           reported sums are not exact even where hashed pieces' would be -- re-score), the ids that reach the
 *                           bar are reported once each from a second look at the piece's paths; the caller keeps 8192 x the heaviest
 *                           fx32 below 2^32 (a coarser `shift` if need be), so that no slot can wrap; bits 17..23: slots of the
 *                           per-workgroup set of reported ids (a power of two <= 64; 0 = 128); bit 24 (eps_scan_plan AND the
 *                           eps_scan_screen launch that uses the plan, together with bit 16): a packed piece of a column of
 *                           <= 256 rows holds up to 8192 paths -- without keys neither key bits, sum field nor table slots bound it;
 *                           *status (device word, cleared by the call): value 2 (bit 1) = a table filled up (results
 *                           invalid; cannot happen within the planner's piece limits: a backstop); value 8 (bit 3) = a sketch
 *                           piece had more ids to report than its set holds (results invalid: repeat the launch without
 *                           bit 16, on a plan without bit 24); value 16 (bit 4) = sketch pieces ran (informational). */
int32_t eps_scan_windows(void);
/* eps_rescore_runs: exact scores of screened survivors.  keys = (u << 32) | v, sorted ascending (runs of equal u: the hubs
 * recur); fixw = eps_fixed_weights(node_w); out[i] = float32 of the exact int64 sum of the 2^-40 fixed-point weights over the
 * common neighbours of pair i -- bit-identical to eps_filter_scan's / eps_expand_fill's score.  Unit-valued adjacency.
 * eps_rescore_weighted: the same for an adjacency with stored values: term = (A[u,w] * A[v,w]) * node_w[w] in float32, each
 * converted to 2^-40 fixed point (eps_expand_fill's weighted score); keys in any order. */
int eps_rescore_runs(const int64_t *rowptr, const int32_t *col, const int64_t *fixw, int64_t n_nodes, const int64_t *keys,
                     int64_t n, float *out, void *stream);
int eps_rescore_weighted(const int64_t *rowptr, const int32_t *col, const float *val, const float *node_w, int64_t n_nodes,
                         const int64_t *keys, int64_t n, float *out, void *stream);
int eps_scan_cuts(const int64_t *rowptr, const int32_t *col, int64_t n_nodes, const int32_t *bounds, uint16_t *cuts,
                  void *stream);
int eps_scan_screen_weights(const int64_t *fixw, int64_t n, int32_t shift, uint32_t *fx32, uint32_t *bad, void *stream);
int eps_scan_window_paths(const int64_t *rowptr, const int32_t *col, const int32_t *revpos, const uint16_t *cuts,
                          int64_t n_nodes, const uint32_t *heads_or_null, uint32_t *wpaths, void *stream);
/* eps_scan_window_paths_columns (r06): the rows of that table for the listed columns only (the other rows are left untouched; no
 * head table): what a launch without a plan table over a few columns -- the bar sample, filter.py:96-142 on every 512-th column --
 * needs, so that a one-shot run builds the whole-graph table and its plan only if a launch without skipped heads asks for them. */
int eps_scan_window_paths_columns(const int64_t *rowptr, const int32_t *col, const int32_t *revpos, const uint16_t *cuts, int64_t n_nodes,
                                  const int32_t *columns, int64_t n_columns, uint32_t *wpaths, void *stream);
int eps_scan_screen(const int64_t *rowptr, const int32_t *col, const int32_t *revpos, const uint32_t *fx32,
                    const uint16_t *cuts, const uint32_t *wpaths, const uint32_t *ssum_or_null,
                    const uint32_t *smax_or_null, const uint32_t *pptr_or_null, const uint32_t *plan_or_null,
                    const uint32_t *heads_or_null, const uint32_t *rowrec_or_null, const uint32_t *pack_or_null, const int32_t *bounds,
                    int64_t n_nodes, int64_t nnz,
                    const int32_t *columns, const uint32_t *colrec_or_null, int64_t n_columns, int64_t batch_from, int32_t shift,
                    int32_t variant, eps_survivors *out, uint32_t *status, void *stream);
/* pack (optional, r06; the main launch only: variant 2 with plan table, row / column records, heads and sum bounds; 16-byte
 * aligned): eps_scan_column_pack's table -- 32 bytes per stored entry (v, j) in CSR order: {col, rowptr[col] (low word), fx32[col],
 * revpos | cut of column v's FIRST piece in row col << 16} {the cuts of v's pieces 1..8 in that row, 16 bits each}.  A column of
 * at most 256 rows then sets up from ONE contiguous stream (a 16-byte load per row) instead of neighbour ids -> one 128-byte
 * row-record line per neighbour.  Built per (graph, weights, plan table): filter.py:96-142 reads no such table -- it is the scan's. */
int eps_scan_column_pack(const int64_t *rowptr, const int32_t *col, const int32_t *revpos, const uint32_t *rowrec, const uint32_t *pptr,
                         const uint32_t *plan, int64_t n_nodes, uint32_t *pack, void *stream);
/* colrec (optional; needs the plan table; 16-byte aligned): 8 words per entry of `columns`, in the same order -- {v, rowptr[v] (low
 * word), degree, heads[2 v], heads[2 v + 1], ssum[v], pptr[v], pptr[v + 1] - pptr[v]} (zeros where a table is absent) -- what a
 * column's set-up otherwise reads through a chain id -> five tables, as ONE 32-byte load at the hand-out index.  The caller builds
 * it with plain gathers (eps_amd.scan.column_records). */
/* eps_scan_row_records: rowrec[w * 32 + 0 .. 15] = the 32 cuts of row w, [16] = rowptr[w] (low word), [17] = fx32[w], rest 0 -- ONE
 * 128-byte line per node with everything eps_scan_screen gathers per walked row (r05: three gathers into three tables were 384
 * bytes of fabric traffic per row for 24 bytes wanted).  Per (graph, weight table); 128-byte aligned; optional for eps_scan_screen. */
int eps_scan_row_records(const uint16_t *cuts, const int64_t *rowptr, const uint32_t *fx32, int64_t n_nodes, uint32_t *rowrec,
                         void *stream);
/* ---- skipped heads (r05; csrc/scan_heads.hip): half of all two-hop paths run through a few thousand hub rows whose weights are
 * the smallest there are, so under a bar a column need not walk them (still filter.py:96-142 + :160-161 under --keep_top) ------
 *   eps_scan_heads     : heads[2 v] = x_v, heads[2 v + 1] = T_v (uint32 pairs, 8-byte aligned): the longest prefix of row v with
 *                        ids < n_hub whose screening weights fx32 sum to T_v <= budget, at most max_rows (<= 65535) of them.  Built per
 *                        (graph, weight table, budget);
 *                        the budget is a fraction of the bar IN TABLE UNITS (bar x 2^shift).
 *   eps_scan_window_paths / eps_scan_plan / eps_scan_screen with heads: the column walks its rows from x_v on (window paths and
 *                        plan count those rows only; the three tables go together, and a head table needs the plan table built
 *                        for it), a table slot passes at bar - T_v, and out->val holds the WALKED sum as raw uint32 bits.
 *                        *status bit 2 (value 4): some T_v >= the bar in table units (the tables were built for a higher bar):
 *                        those columns were left out, the list is not valid.  A launch without a bar must not bring heads.
 *   eps_scan_hub_rows  : hubrows[w * words + (x >> 5)] bit (x & 31) = "x is a neighbour of w" for w < n_hub <= min(65536, n_nodes),
 *                        words = eps_scan_hub_row_words(n_nodes) (16-byte aligned rows): the adjacency rows of the first
 *                        n_hub ids as bitmaps over the id space.  Per-graph table (cleared and filled by the call).
 *   eps_scan_refine    : completes a walked list: every valid slot (key >= 0) of `walked` gets the exact head term -- the sum
 *                        of fx32[w] over the skipped rows w = col[rowptr[v] + j], j < x_v, that hold u (bit u of hub row w) -- added;
 *                        sums at or above out->threshold (in table units, as eps_scan_screen rounds it) are appended to `out`
 *                        (compact: out->count = their number; val = sum x 2^-shift) -- the list a launch without heads reports,
 *                        in another order.  out->count must be zeroed by the caller. */
int eps_scan_heads(const int64_t *rowptr, const int32_t *col, const uint32_t *fx32, int64_t n_nodes, int32_t n_hub,
                   uint32_t budget, int32_t max_rows, uint32_t *heads, void *stream);
int64_t eps_scan_hub_row_words(int64_t n_nodes);
int eps_scan_hub_rows(const int64_t *rowptr, const int32_t *col, int64_t n_nodes, int32_t n_hub, uint32_t *hubrows, void *stream);
int eps_scan_refine(const eps_survivors *walked, const uint32_t *heads, const uint32_t *hubrows, int32_t n_hub,
                    const uint32_t *fx32, const int64_t *rowptr, const int32_t *col, int64_t n_nodes, int32_t shift,
                    eps_survivors *out, void *stream);
/* eps_scan_plan: the per-graph PLAN TABLE of eps_scan_screen -- every column's pieces, planned once by the scan kernel's own
 * planner (they depend on the graph, the window tables, the variant and, through ssum / smax, the weight table; not on the bar
 * or on the columns of a launch).  Two calls: with pptr / plan NULL it counts (pcount[v] = pieces of column v, uint32[n_nodes]);
 * the caller forms pptr = exclusive prefix sum (uint32[n_nodes + 1]) and calls again with plan = uint32[4 * pptr[n_nodes]]
 * (16-byte aligned; one 16-byte record per piece).  Pass the SAME wpaths / ssum / smax / shift / variant to eps_scan_screen
 * together with pptr / plan: a launch then reads a column's records instead of planning it (5 % of the launch).
 * d_used_or_null (fill call): the largest number of low weight bits any packed / 16-bit direct piece of the plan drops -- what
 * the caller's bound on (screening score - exact score) needs. */
/* eps_scan_bounds: the window boundaries eps_scan_cuts / eps_scan_screen expect -- bounds[0 .. M], M = eps_scan_windows(), windows
 * of equal stored-entry mass (bounds[k] = 1 + the first node whose row ends at or beyond k x nnz / M; non-decreasing).
 * eps_scan_row_sums: the per-node sum bounds of eps_scan_screen / eps_scan_plan: ssum[v] = min(2^31 - 1, sum of fx32 over row v),
 * smax[k] = the largest ssum among ids >= bounds[k] (k = 0 .. M; smax[M] = 0), *min_fx = the smallest fx32 of a node with at least
 * two neighbours (0xFFFFFFFF when there is none): (min_fx - 1) x 2^-shift is a floor under every path's exact term -- what bounds
 * the number of paths behind a screening sum.  workspace: M uint32. */
int eps_scan_bounds(const int64_t *rowptr, int64_t n_nodes, int32_t *bounds, void *stream);
int eps_scan_row_sums(const int64_t *rowptr, const int32_t *col, const uint32_t *fx32, const int32_t *bounds, int64_t n_nodes,
                      uint32_t *ssum, uint32_t *smax, uint32_t *min_fx, void *workspace, void *stream);
/* eps_scan_plan_rewalk: out2[0] = two-hop half paths that eps_scan_screen would walk AGAIN under this plan (hash-partitioned
 * passes of windows that are both wide and heavy), out2[1] = all half paths of the plan: what decides whether the one-pass
 * kernel suits a graph (eps_amd.scan.screen_variant). */
int eps_scan_plan_rewalk(const uint32_t *plan, int64_t n_rec, int32_t variant, unsigned long long *out2, void *stream);
int eps_scan_plan(const int64_t *rowptr, const uint16_t *cuts, const uint32_t *wpaths, const uint32_t *ssum_or_null,
                  const uint32_t *smax_or_null, const uint32_t *heads_or_null, const int32_t *bounds, int64_t n_nodes, int32_t shift,
                  int32_t variant, uint32_t *pcount, const uint32_t *pptr_or_null, uint32_t *plan_or_null, uint32_t *d_used_or_null,
                  void *stream);
/* eps_scan_screen_weighted: the scan on a SYMMETRIC adjacency WITH stored values (collab: rank.py:32-35 keeps the summed
 * multi-edge weights; val[e] must equal the value of e's mirror entry and be positive).  A path's term is
 * (A[u,w] * A[v,w]) * node_w[w] -- symmetric in (u, v), so the half scheme holds -- and its screening weight is formed per
 * path from the float values, rounded up.  node_w = the float32 node weights (no fx32 table); re-score with
 * eps_rescore_weighted. */
int eps_scan_screen_weighted(const int64_t *rowptr, const int32_t *col, const float *val, const int32_t *revpos,
                             const float *node_w, const uint16_t *cuts, const uint32_t *wpaths, const int32_t *bounds,
                             int64_t n_nodes, int64_t nnz, const int32_t *columns, int64_t n_columns, int32_t shift,
                             int32_t variant, eps_survivors *out, uint32_t *status, void *stream);

/* ---- the candidate list of a block of columns of a graph WITHOUT stored values, on the scan kernel's structure ----------
 * Same results as eps_expand_count / eps_expand_fill above (filter.py:96-109: every 2-hop non-edge of columns
 * [v_lo, v_hi), column-major, u ascending; score = sum_w A[u,w] A[v,w] node_w[w] in 2^-40 fixed point, rounded once --
 * bit-identical), for unit-valued adjacencies: packed 64-entry units, 4-byte bucket records, one fixed-point weight per
 * (v, w) from a table (fixw = eps_fixed_weights(node_w); NULL with score == NULL: the list only).  col_order (optional)
 * = hand-out order, a permutation of range(v_hi - v_lo).  colptr[v_hi - v_lo + 1] = exclusive prefix of the counts
 * (or any upper-bound layout: the rest of a segment is padded with cand_u -1 / score -inf and cand_count, if given,
 * receives the real counts).  status: one device word, cleared by the call; after the stream has drained bit 1 = a
 * column outgrew its segment, bit 2 = a sum left the fixed-point range -- the outputs are then invalid.  splits /
 * max_degree / workspace as for eps_filter_scan (eps_filter_scan_workspace_bytes(max_degree) bytes).
 * revpos_or_null (eps_reverse_positions; symmetric pattern only): non-NULL selects the HALF list -- column v lists its
 * candidates u < v only, every unordered pair once -- for consumers that are symmetric in (u, v) themselves, like the
 * LinkPredictor decode of h_u * h_v (models.py:478-485): half the list, half the decode, the same scores. */
int eps_expand_unit_count(const int64_t *rowptr, const int32_t *col, const int32_t *revpos_or_null,
                          const int32_t *splits_or_null, int64_t n_nodes, int64_t nnz, int64_t max_degree, int64_t v_lo,
                          int64_t v_hi, const int32_t *col_order, int64_t *cand_count, void *workspace,
                          int64_t workspace_bytes, void *stream);
int eps_expand_unit_fill(const int64_t *rowptr, const int32_t *col, const int32_t *revpos_or_null, const int64_t *fixw,
                         const int32_t *splits_or_null,
                         int64_t n_nodes, int64_t nnz, int64_t max_degree, int64_t v_lo, int64_t v_hi,
                         const int32_t *col_order, const int64_t *colptr, int64_t *cand_count, int32_t *cand_u,
                         int32_t *cand_v, float *score, uint32_t *status, void *workspace, int64_t workspace_bytes,
                         void *stream);
/* eps_expand_unit_list (r06): the same list in ONE pass over the two-hop paths -- what filter.py:96-142 does once per candidate
 * -- without the counting launch and without a host read before the launch: colptr_ub[v_hi - v_lo + 1] is an exclusive
 * prefix of UPPER BOUNDS of the columns' candidate counts (min(two-hop paths of the column, n_nodes) always holds), column v
 * fills the front of [colptr_ub[v], colptr_ub[v + 1]) with its candidates (u ascending) and their scores -- the same values
 * as eps_expand_unit_fill, bit for bit --, cand_count[v] receives how many, and the rest of the segment is left unwritten
 * (no cand_v: the column of slot i is the segment that holds it).  status / workspace / splits / revpos_or_null as above. */
int eps_expand_unit_list(const int64_t *rowptr, const int32_t *col, const int32_t *revpos_or_null, const int64_t *fixw,
                         const int32_t *splits_or_null, int64_t n_nodes, int64_t nnz, int64_t max_degree, int64_t v_lo,
                         int64_t v_hi, const int32_t *col_order, const int64_t *colptr_ub, int64_t *cand_count,
                         int32_t *cand_u, float *score, uint32_t *status, void *workspace, int64_t workspace_bytes,
                         void *stream);

/* ---- K4/K5: CSR x dense SpMM with fused epilogue -----------------------------------------
 * Replaces torch_sparse spmm_sum / spmm_mean inside GCNConv / SAGEConv (models.py:183-186,
 * :436-439) plus the bias add and the ReLU of the layer loop.
 *   mean == 0:  Y[i,:] = sum_k val[k] * X[col[k],:]
 *   mean != 0:  Y[i,:] = (sum_k X[col[k],:]) / max(rowlen(i), 1)      (values ignored)
 *   then  + bias[:] (if non-NULL), then ReLU (if relu != 0).
 * X is row-major [n_cols, ldx] (first f columns used), Y row-major [n_rows, ldy]. */
int eps_spmm_csr(const int64_t *rowptr, const int32_t *col, const float *val, int64_t n_rows,
                 const float *x, int64_t ldx, int32_t f, const float *bias, int relu, int mean,
                 float *y, int64_t ldy, void *stream);

/* gcn_norm (torch_geometric 1.7.0 GCNConv, third-party): given A^ (diagonal already set to 1)
 * compute dis = rowsum^-1/2 (inf -> 0) and val_out[k] = (val[k]*dis[row])*dis[col[k]].
 * dis is an n_rows scratch/output buffer. */
int eps_gcn_norm(const int64_t *rowptr, const int32_t *col, const float *val, int64_t n_rows,
                 float *dis, float *val_out, void *stream);

/* ---- dense f32 GEMM on the f32-input MFMA (exact f32) -------------------------------------
 * C[M,N] = act(A[M,K] * B[N,K]^T + bias[N] (+ C if accumulate)); B is row-major [N,K]
 * (torch.nn.Linear layout; GCNConv.weight [in,out] is passed transposed by the host).
 * Serves `x @ W` of GCNConv and lin_l / lin_r of SAGEConv (models.py:183, :436). */
int eps_gemm_f32(const float *a, int64_t lda, const float *b, int64_t ldb, const float *bias,
                 int relu, int accumulate, float *c, int64_t ldc, int64_t m, int32_t n, int32_t k,
                 void *stream);
/* `relu`: bit 0 = ReLU; bit 1 (value 2) = the product is symmetric and only the 128 x 128 tiles on and below the diagonal are
 * computed (the rest of C is left untouched).
 *
 * ---- common neighbours of a DENSE graph through the matrix cores (r05; csrc/dense_cn.hip) --------------------------------
 * filter.py:96-121 with CommonNeighborsPredictor('simple') (models.py:536-542) on a graph like ogbl-ddi (N = 4,267, 11.7 % of
 * all pairs are edges): CN = A A^T as one dense float32 product (exact below 2^24), the candidate list a masked read of it.
 *   eps_dense_adjacency : a[v * ld + w] = 1.0f for every stored entry, 0 elsewhere; the matrix has `rows` >= n_nodes rows of
 *                         ld >= n_nodes floats (padding rows / columns zero: the product's tiles want multiples of 128).
 *   eps_dense_candidates: the candidates of column v -- c[v * ld + u] > 0, a[v * ld + u] == 0, u < v (below_only: each unordered
 *                         pair once; c needs its lower tiles) or u != v (the reference's directed list; c needs all tiles) -- in
 *                         ascending u.  First call (colptr NULL): counts[v] = their number; the caller forms colptr = exclusive
 *                         prefix sum; second call: keys[colptr[v] + i] = (v << 32) | u, vals[..] = c[v * ld + u], and / or
 *                         rows[3 (colptr[v] + i) ..] = (float u, float v, count): the proposal file's row (filter.py:119).
 *   eps_dense_mirror_lower: c[u * ld + v] = c[v * ld + u] for u < v -- the upper triangle of a product computed with the triangle
 *                         flag, for readers that want whole rows. */
int eps_dense_adjacency(const int64_t *rowptr, const int32_t *col, int64_t n_nodes, int64_t ld, int64_t rows, float *a,
                        void *stream);
int eps_dense_candidates(const float *a, const float *c, int64_t n_nodes, int64_t ld, int32_t below_only, int64_t *counts,
                         const int64_t *colptr_or_null, int64_t *keys_or_null, float *vals_or_null, float *rows_or_null,
                         void *stream);
int eps_dense_mirror_lower(float *c, int64_t n, int64_t ld, void *stream);

/* ---- K6: fused LinkPredictor decode ------------------------------------------------------
 * Replaces h[edges[0]], h[edges[1]] gathers (models.py:506) + LinkPredictor.forward
 * (models.py:478-485): out[p] = sigmoid(W_L ... relu(W_1 (h[u_p] (.) h[v_p]) + b_1) ... + b_L).
 * h: [n_nodes, hdim] row-major (ld = hdim).  w[l]: device pointers to row-major [out_l, in_l]
 * matrices (torch.nn.Linear layout), b[l]: [out_l]; hidden width == hdim for every hidden
 * layer, last layer out == 1.  w / b are HOST arrays of n_layers device pointers.
 * apply_sigmoid == 0 returns the pre-sigmoid logit.  Supported: hdim % 4 == 0, hdim <= 256,
 * 1 <= n_layers <= 8. */
int eps_mlp_decode(const float *h, int64_t n_nodes, int32_t hdim, const int32_t *u,
                   const int32_t *v, int64_t n_pairs, const float *const *w,
                   const float *const *b, int32_t n_layers, int apply_sigmoid, float *out,
                   void *stream);

/* ---- top-K selection with the declared tie rule ------------------------------------------
 * Replaces `all_scores[:,2].sort(descending=True)` (filter.py:160-161) for the K rows rank.py
 * ever reads (rank.py:294).  Declared order: score descending, then `id` ascending (== a stable
 * descending sort over the reference's candidate order; the reference's own torch.sort is
 * unstable on ties).  (score, id) is packed into an int64 whose SIGNED order realises the rule:
 *   key = (ordered_bits(score) ^ 0x80000000) << 32 | (0xFFFFFFFF - id),   id < 2^32,
 * so a plain descending sort / top-K / k-way merge of keys -- on one GPU or across shards --
 * gives the same result.  id = ids[i] if ids != NULL else id_base + i. */
/* k-th largest value of a float32 array (1 <= k <= n < 2^32) by radix select over the same order-preserving bit pattern:
 * the bar of the threshold scan and the cut of the final top-K need this one value, not a sorted array.  *kth (device)
 * receives it; torch.sort semantics for the order (-0 == +0; NaNs sort above +inf).  workspace:
 * eps_kth_largest_workspace_bytes() bytes, 8-byte aligned, contents arbitrary. */
int64_t eps_kth_largest_workspace_bytes(void);
int eps_kth_largest_f32(const float *x, int64_t n, int64_t k, float *kth, void *workspace, void *stream);
/* The same radix select in steps, for a vector that is spread over the ranks of a torch.distributed job (the bar of the
 * sharded scan, the final cut of filter.py:160-161 over all ranks' survivors).  state = eps_kth_largest_workspace_bytes()
 * bytes on the device, 8-byte aligned: { uint32 prefix, mask; uint64 k; uint32 hist[256] } (hist at byte 16).  Per round
 * (shift = 24, 16, 8, 0): eps_kth_hist_f32 adds the histogram of this rank's values (n may be 0), the caller sums hist over
 * the ranks (one all-reduce of 1 KiB), eps_kth_pick narrows the prefix and clears hist.  After the round with shift 0,
 * *out = the k-th largest of the union, -inf when the union holds fewer than k values. */
/* eps_select_compact: radix select + threshold + compaction of a short list in ONE launch (the selections of a filter step --
 * filter.py:160-161 keeps the K best rows -- were launch-latency bound: ten launches per select, one more per compaction).
 *   n = min(*n_dev_or_null, n_max): the slot counter of an eps_survivors list (eps_scan_screen marks the unused slots of its
 *   reservations "no survivor" itself, so a reader stops at the counter and the list needs no fill before the launch);
 *   entries with value -inf (and, when keys is given, key < 0) are no values;
 *   *kth = the k-th largest value, -inf when there are fewer than k (k == 0: -inf, nothing selected);
 *   *thr = mode 0: kth; mode 1: the largest float below kth (an inclusive bar for a scan that keeps scores above its
 *          threshold); mode 2: max(kth - pa, kth * pb) - |kth| * pc (a lower bound of the exact score behind a screening
 *          score: eps_amd.scan.Screen.lower_bound);
 *   out_keys / out_vals (out_cap entries each; both NULL: selection only) receive the entries with key >= 0 and value >= *thr in
 *   arbitrary order, *n_out (DEVICE int64) their number -- entries beyond out_cap are counted, not stored.  state: eps_select_compact_workspace_bytes() bytes, 8-byte aligned.
 * Single device; the sharded job-wide select stays eps_kth_begin / _hist_f32 / _pick with one all-reduce per round. */
int64_t eps_select_compact_workspace_bytes(void);
int eps_select_compact(const int64_t *keys_or_null, const float *vals, int64_t n_max, const unsigned long long *n_dev_or_null,
                       int64_t k, int32_t mode, float pa, float pb, float pc, float *kth_or_null, float *thr_or_null,
                       int64_t *out_keys_or_null, float *out_vals_or_null, int64_t out_cap, int64_t *n_out_or_null, void *state,
                       void *stream);
int eps_kth_begin(void *state, int64_t k, void *stream);
int eps_kth_hist_f32(const float *x, int64_t n, void *state, int32_t shift, void *stream);
int eps_kth_pick(void *state, int32_t shift, float *out_or_null, void *stream);
/* The k best DIRECTED proposals of a list of n unordered survivors of eps_filter_scan (keys v << 32 | u with u < v, one
 * entry per pair, no -1 slots; both orientations carry the pair's score), sorted by the declared rule -- score descending,
 * then key ascending (the reference's column-major candidate order): filter.py:160-161 for the rows rank.py:294 reads.
 *   eps_select_topk_cut : the pairs whose score reaches the ceil(k/2)-th best one, compacted into sel_keys / sel_vals
 *                         (n entries each, arbitrary order); *n_sel (DEVICE word) = their number m, which the caller reads.
 *   eps_select_topk_rows: those m pairs -> 2 m rows (v << 32 | u of proposal (u, v), score), sorted; the first
 *                         min(k, 2 m) go to out_keys / out_vals.  id_bits: every node id is below 2^id_bits.
 * workspaces: eps_select_topk_cut_workspace_bytes() / eps_select_topk_rows_workspace_bytes(m) bytes, 256-byte aligned. */
/* eps_compact_survivors: the survivors among the first n slots of an eps_survivors list (untouched slots keep key -1),
 * compacted in arbitrary order into out_keys / out_vals (n entries each); *n_out (DEVICE word) = how many.  workspace as for
 * eps_select_topk_cut. */
/* eps_select_topk_rows_relabelled: eps_select_topk_rows with the pairs' ids mapped through perm first (int64[n_nodes]: id i of a
 * relabelled, scanned graph is the caller's id perm[i]) -- rows and their order are in the caller's labels.
 * eps_sort_pairs_by_u: survivor keys v << 32 | u (u < v) -> u << 32 | v sorted by (u, v) (two stable radix sorts over the
 * id_bits bits of v, then of u): runs of equal u with ascending v, the input eps_rescore_runs wants.  v_block_shift > 0: by
 * (v >> v_block_shift, u, v) instead -- blocks of consecutive v first, so that concurrent workgroups of eps_rescore_runs stream
 * the same rows (0: off) -- WHEN the runs of equal (block, u) still average 64 pairs; otherwise the (u, v) order is returned
 * (eps_rescore_runs pays one bitmap of N(u) per run; counted and decided on the device).  workspace:
 * eps_sort_pairs_by_u_workspace_bytes(n) bytes, 256-byte aligned. */
/* eps_compact_between: the entries (key >= 0) whose score lies in [*lo, *hi) (DEVICE floats; either may be NULL: open end),
 * compacted in arbitrary order; *n_out (DEVICE int64) = how many.  A sharded filter step deals its final ordering over the
 * ranks by score range. */
int eps_compact_between(const int64_t *keys, const float *vals, int64_t n, const float *lo_or_null, const float *hi_or_null,
                        int64_t *out_keys, float *out_vals, int64_t *n_out, void *stream);
int eps_select_topk_rows_relabelled(const int64_t *sel_keys, const float *sel_vals, int64_t m, int64_t k, int32_t id_bits,
                                    const int64_t *perm, int64_t *out_keys, float *out_vals, void *workspace,
                                    int64_t workspace_bytes, void *stream);
/* eps_select_topk_rows_pairs (r06): the same rows written as the [2, k] proposal tensor rank.py:294 reads -- out_pairs[i] = u,
 * out_pairs[pairs_ld + i] = v (pairs_ld >= min(k, 2 m)) -- instead of packed keys; perm_or_null as in _relabelled. */
int eps_select_topk_rows_pairs(const int64_t *sel_keys, const float *sel_vals, int64_t m, int64_t k, int32_t id_bits,
                               const int64_t *perm_or_null, int64_t *out_pairs, int64_t pairs_ld, float *out_vals, void *workspace,
                               int64_t workspace_bytes, void *stream);
int64_t eps_sort_pairs_by_u_workspace_bytes(int64_t n);
int eps_sort_pairs_by_u(const int64_t *keys, int64_t n, int32_t id_bits, int32_t v_block_shift, int64_t *out_by_u, void *workspace,
                        int64_t workspace_bytes, void *stream);
int64_t eps_select_topk_cut_workspace_bytes(void);
int eps_compact_survivors(const int64_t *keys, const float *vals, int64_t n, int64_t *out_keys, float *out_vals,
                          int64_t *n_out, void *workspace, int64_t workspace_bytes, void *stream);
/* eps_compact_at_least: the survivors (key >= 0) among n slots whose score is at least *cut_or_null (a DEVICE float: the
 * job-wide cut from eps_kth_pick; NULL keeps every survivor), compacted in arbitrary order; *n_out (DEVICE int64, zeroed by
 * the call) = how many.  No workspace, no host round trip. */
int eps_compact_at_least(const int64_t *keys, const float *vals, int64_t n, const float *cut_or_null, int64_t *out_keys,
                         float *out_vals, int64_t *n_out, void *stream);
int eps_select_topk_cut(const int64_t *keys, const float *vals, int64_t n, int64_t k, int64_t *sel_keys, float *sel_vals,
                        int64_t *n_sel, void *workspace, int64_t workspace_bytes, void *stream);
int64_t eps_select_topk_rows_workspace_bytes(int64_t m);
int eps_select_topk_rows(const int64_t *sel_keys, const float *sel_vals, int64_t m, int64_t k, int32_t id_bits,
                         int64_t *out_keys, float *out_vals, void *workspace, int64_t workspace_bytes, void *stream);
/* ---- the tail of the filter step with DEVICE-side sizes (r06; csrc/tail_sort.hip) ------------------------------------------
 * What follows the scan under `--keep_top K` -- the pre-filter in front of the exact re-scoring, the cut, and the two orderings
 * of filter.py:160-161 for the K rows rank.py:294 reads -- without a host read in between: every count stays in device memory.
 *   state: eps_tail_state_bytes() bytes, 256-byte aligned, ZEROED once by the caller and owned by one stream; the kernels that
 *          consume it leave it zeroed (no memset launches per step).
 *   eps_score_hist: adds the histogram of a list's live scores (key >= 0 when keys are given, value > -inf and > *above_or_null)
 *          over order-preserving buckets of their distance to *base (DEVICE floats) to the state; n = min(*n_dev_or_null, n_max).
 *   eps_score_pick_compact: *kth = the lower edge of the highest bucket with at least k values at or above it (at most 2^-8 of the
 *          distance to *base below the exact k-th best value; -inf when fewer than k values or k == 0), *thr derived as
 *          eps_select_compact derives it (mode 0 / 1 / 2); the live entries with value >= *thr go to out_keys / out_vals
 *          (arbitrary order; swap_halves: key halves exchanged, v << 32 | u -> u << 32 | v and back), *n_out (DEVICE) = how many
 *          -- beyond out_cap counted, not stored.  Leaves the state's histogram zeroed.
 *   eps_radix_sort_by_u: the order eps_sort_pairs_by_u produces, in ONE cooperative launch, n read from the device.
 *   eps_radix_sort_rows: eps_select_topk_rows[_relabelled] in ONE cooperative launch, m read from the device; writes the
 *          proposal tensor itself: out_pairs[i] = u, out_pairs[out_ld + i] = v, out_scores[i] for the first min(k, 2 m) rows of
 *          the declared order; *n_rows_out_or_null (DEVICE) = that number.  workspace: eps_radix_sort_workspace_bytes(records)
 *          bytes (records = n_max resp. 2 m_max), 256-byte aligned.
 *   eps_rescore_runs_dev: eps_rescore_runs over min(*n_dev, n_max) pairs. */
int64_t eps_tail_state_bytes(void);
/* eps_score_bins / eps_score_hist_into / eps_score_deal_plan: the same selection for a SHARDED step.  Every rank adds the histogram
 * of its re-scored scores to an array of its own (eps_score_bins() words, zeroed by the caller, the same *base on every rank) and
 * the arrays are all-gathered -- 25 KB per rank instead of its scores; eps_score_deal_plan then gives every rank, from the same
 * table hists[r * row_stride + bucket], the job-wide *cut (lower edge of the highest bucket with at least k values at or above it;
 * -inf when fewer), the world - 1 descending splitters[] of the final ordering (bucket edges: equal scores never straddle one;
 * range q = the selected scores in [splitters[q], splitters[q - 1]) holds about 1 / world of the selected pairs), counts[r * world
 * + q] = selected pairs of rank r in range q, and nsel[r] -- without another exchange.  filter.py:160-161 dealt over the ranks. */
int32_t eps_score_bins(void);
int eps_score_hist_into(const int64_t *keys_or_null, const float *vals, int64_t n_max, const unsigned long long *n_dev_or_null,
                        const float *base, const float *above_or_null, uint32_t *hist, void *stream);
int eps_score_deal_plan(const uint32_t *hists, int64_t row_stride, int32_t world, int64_t k, const float *base, float *cut,
                        float *splitters, int64_t *counts, int64_t *nsel, void *stream);
int eps_score_hist(const int64_t *keys_or_null, const float *vals, int64_t n_max, const unsigned long long *n_dev_or_null,
                   const float *base, const float *above_or_null, void *state, void *stream);
int eps_score_pick_compact(const int64_t *keys_or_null, const float *vals, int64_t n_max, const unsigned long long *n_dev_or_null,
                           const float *base, const float *above_or_null, int64_t k, int32_t mode, float pa, float pb, float pc,
                           int32_t swap_halves, float *kth_or_null, float *thr_or_null, int64_t *out_keys_or_null,
                           float *out_vals_or_null, int64_t out_cap, int64_t *n_out_or_null, void *state, void *stream);
int64_t eps_radix_sort_workspace_bytes(int64_t n_records);
int eps_radix_sort_by_u(const int64_t *keys, int64_t n_max, const int64_t *n_dev_or_null, int32_t id_bits, int32_t v_block_shift,
                        int64_t *out_by_u, void *workspace, int64_t workspace_bytes, void *state, void *stream);
int eps_radix_sort_rows(const int64_t *sel_keys, const float *sel_vals, int64_t m_max, const int64_t *m_dev_or_null, int64_t k,
                        int32_t id_bits, const int64_t *perm_or_null, int64_t *out_pairs, int64_t out_ld, float *out_scores,
                        int64_t *n_rows_out_or_null, void *workspace, int64_t workspace_bytes, void *state, void *stream);
int eps_rescore_runs_dev(const int64_t *rowptr, const int32_t *col, const int64_t *fixw, int64_t n_nodes, const int64_t *keys,
                         int64_t n_max, const int64_t *n_dev, float *out, void *stream);
int eps_pack_keys(const float *score, const int64_t *ids_or_null, int64_t id_base, int64_t n,
                  int64_t *keys, void *stream);
int eps_unpack_keys(const int64_t *keys, int64_t n, float *score_or_null, int64_t *id_or_null,
                    void *stream);

#ifdef __cplusplus
}
#endif
#endif /* EPS_ABI_H */
