// Common-neighbour counts of a DENSE graph through the matrix cores, gfx950 (r05).
//
// What it replaces: filter.py:96-109 (every 2-hop non-edge) + :113-121 with CommonNeighborsPredictor('simple'), models.py:536-542
// (score = sum_w adj[u,w] adj[v,w]) on a graph like ogbl-ddi -- N = 4,267, 11.7 % of all pairs are edges.  On such a graph the
// sparse machinery (column blocks, LDS tables, hundreds of launches for 16 M candidates) is the wrong tool: the adjacency fits a
// 4352 x 4352 float matrix, CN = A A^T is ONE dense product (counts are exact in float32 below 2^24; A is symmetric, so only the
// tiles on and below the diagonal are computed: eps_gemm_f32 with the triangle flag), and the candidate list is a masked read of
// the product: C[v][u] > 0, A[v][u] == 0 -- column-major like the reference's list; u < v (each unordered pair once) or every
// u != v (both orientations: the rows of the proposal file, which then only need ONE stable sort by the integer count).
#include "eps_common.h"

// A[v * ld + w] = 1 for every stored entry (v, w); the matrix was cleared by the caller (hipMemsetAsync).
__global__ __launch_bounds__(256) void dn_scatter_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                         int64_t n_nodes, int64_t ld, float *__restrict__ A)
{
    const int lane = threadIdx.x & 63;
    const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= n_nodes) return;
    const int64_t b = rowptr[v], e = rowptr[v + 1];
    for (int64_t i = b + lane; i < e; i += 64) A[v * ld + col[i]] = 1.0f;
}

extern "C" int eps_dense_adjacency(const int64_t *rowptr, const int32_t *col, int64_t n_nodes, int64_t ld, int64_t rows, float *a,
                                   void *stream)
{
    EPS_REQUIRE(n_nodes >= 0 && ld >= n_nodes && rows >= n_nodes && rows * ld < (1ll << 31), "eps_dense_adjacency: bad shape");
    if (rows == 0) return EPS_OK;
    EPS_REQUIRE(rowptr && col && a, "eps_dense_adjacency: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(a, 0, (size_t)rows * (size_t)ld * 4, s) != hipSuccess) {
        eps_set_error("eps_dense_adjacency: cannot clear the matrix");
        return EPS_ELAUNCH;
    }
    if (n_nodes)
        hipLaunchKernelGGL(dn_scatter_kernel, dim3((unsigned)((n_nodes + 3) / 4)), dim3(256), 0, s, rowptr, col, n_nodes, ld, a);
    EPS_CHECK_LAUNCH("eps_dense_adjacency");
    return EPS_OK;
}

// One wave per column v: its candidates u in ascending order -- u < v (`below` = the unordered list: each pair once), or every
// u != v (the reference's directed list; C must then hold all tiles).  FILL = false: counts[v]; FILL = true: keys / vals from
// colptr[v].
template <bool FILL>
__global__ __launch_bounds__(256) void dn_candidates_kernel(const float *__restrict__ A, const float *__restrict__ C, int64_t n_nodes,
                                                            int64_t ld, int below, int64_t *__restrict__ counts,
                                                            const int64_t *__restrict__ colptr, int64_t *__restrict__ keys,
                                                            float *__restrict__ vals, float *__restrict__ rows)
{
    const int lane = threadIdx.x & 63;
    const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= n_nodes) return;
    const float *__restrict__ ar = A + v * ld, *__restrict__ cr = C + v * ld;
    int64_t at = FILL ? colptr[v] : 0;
    const int64_t u_end = below ? v : n_nodes;
    for (int64_t u0 = 0; u0 < u_end; u0 += 64) {
        const int64_t u = u0 + lane;
        const float c = u < u_end ? cr[u] : 0.0f;
        const bool cand = u < u_end && u != v && c > 0.0f && ar[u] == 0.0f;
        const unsigned long long m = __ballot(cand);
        if (FILL && cand) {
            const int64_t pos = at + __popcll(m & ((1ull << lane) - 1ull));
            if (keys) keys[pos] = (v << 32) | u;
            vals[pos] = c;
            if (rows) {                          // (the proposal file's row: ids as floats, filter.py:119)
                rows[3 * pos] = (float)u;
                rows[3 * pos + 1] = (float)v;
                rows[3 * pos + 2] = c;
            }
        }
        at += __popcll(m);
    }
    if (!FILL && lane == 0) counts[v] = at;
}

extern "C" int eps_dense_candidates(const float *a, const float *c, int64_t n_nodes, int64_t ld, int32_t below_only, int64_t *counts,
                                    const int64_t *colptr_or_null, int64_t *keys_or_null, float *vals_or_null, float *rows_or_null,
                                    void *stream)
{
    EPS_REQUIRE(n_nodes >= 0 && ld >= n_nodes, "eps_dense_candidates: bad shape");
    if (n_nodes == 0) return EPS_OK;
    EPS_REQUIRE(a && c, "eps_dense_candidates: null pointer");
    const dim3 grid((unsigned)((n_nodes + 3) / 4));
    if (colptr_or_null) {
        EPS_REQUIRE(vals_or_null && (keys_or_null || rows_or_null), "eps_dense_candidates: the fill needs vals and keys or rows");
        hipLaunchKernelGGL(dn_candidates_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a, c, n_nodes, ld, (int)below_only, nullptr,
                           colptr_or_null, keys_or_null, vals_or_null, rows_or_null);
    } else {
        EPS_REQUIRE(counts, "eps_dense_candidates: nowhere to put the counts");
        hipLaunchKernelGGL(dn_candidates_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a, c, n_nodes, ld, (int)below_only, counts,
                           nullptr, nullptr, nullptr, nullptr);
    }
    EPS_CHECK_LAUNCH("eps_dense_candidates");
    return EPS_OK;
}

// C[u][v] = C[v][u] for u < v: the upper triangle of a symmetric product whose lower tiles were computed.  32 x 32 tiles through
// LDS (both sides coalesced); tiles strictly below the diagonal write their transpose, diagonal tiles mirror themselves.
__global__ __launch_bounds__(256) void dn_mirror_kernel(float *__restrict__ C, int64_t n, int64_t ld)
{
    __shared__ float t[32][33];
    const int64_t nt = (n + 31) / 32;
    // tile (i, j), j <= i, from the linear block id over the lower triangle
    int64_t b = blockIdx.x, i = (int64_t)((__builtin_sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
    while (i * (i + 1) / 2 > b) --i;
    while ((i + 1) * (i + 2) / 2 <= b) ++i;
    const int64_t j = b - i * (i + 1) / 2;
    if (i >= nt) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int64_t row = i * 32 + r, cc = j * 32 + tx;
        t[r][tx] = row < n && cc < n ? C[row * ld + cc] : 0.0f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int64_t row = j * 32 + r, cc = i * 32 + tx;           // the transposed tile: element (row, cc) = t[tx][r]
        if (row < n && cc < n && cc > row) C[row * ld + cc] = t[tx][r];
    }
}

extern "C" int eps_dense_mirror_lower(float *c, int64_t n, int64_t ld, void *stream)
{
    EPS_REQUIRE(n >= 0 && ld >= n, "eps_dense_mirror_lower: bad shape");
    if (n == 0) return EPS_OK;
    EPS_REQUIRE(c, "eps_dense_mirror_lower: null pointer");
    const int64_t nt = (n + 31) / 32;
    hipLaunchKernelGGL(dn_mirror_kernel, dim3((unsigned)(nt * (nt + 1) / 2)), dim3(256), 0, (hipStream_t)stream, c, n, ld);
    EPS_CHECK_LAUNCH("eps_dense_mirror_lower");
    return EPS_OK;
}

__global__ void dense_cn_warm_kernel() {}
extern "C" void eps_warm_dense_cn(void *stream) { hipLaunchKernelGGL(dense_cn_warm_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream); }
