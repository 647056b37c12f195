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
