// Skipped heads of the one-pass threshold scan, gfx950 (r05).
//
// What it replaces: still filter.py:96-142 + :160-161 under `--keep_top K` (every 2-hop non-edge, its heuristic score --
// adamic_utils.py:13-25, train_and_eval.py:195-216, models.py:536-542 --, the K best rows); this file is about NOT walking the
// paths that cannot decide whether a pair reaches the bar.
//
// A column v of eps_scan_screen walks the rows w of its neighbours and adds weight[w] to the table slot of every endpoint u.
// Half of all two-hop paths run through a few thousand hub rows (row w carries d_w^2 / 2 paths), while a hub's weight is the
// SMALLEST there is (1 / log d_w, 1 / d_w; CN: the same as everybody's).  So a column leaves its first x_v rows -- ids ascend
// inside a row: under hubs-first labels these are its heaviest hub neighbours -- unwalked, as long as their screening weights
// sum to at most a budget B = beta x bar (eps_scan_heads: x_v, T_v per column).  A pair's score is then at most
// walked(u, v) + T_v: the table sweep lets a slot pass at bar - T_v (per column; everything the walk never touched is below
// T_v < bar and needs no look), and the few slots that pass -- 16 M of 6.4 G candidates at beta = 1/2 on the ppa-like graph --
// get the head's EXACT term added here (eps_scan_refine) before they are compared with the bar itself:
//     head(u, v) = sum of weight[w] over w in N(u) & N(v), w < b_v   (b_v = 1 + the last skipped row's id),
// one bit of the per-graph HUB ROW BITMAPS per skipped row (eps_scan_hub_rows: the adjacency rows of the ids below n_hub as
// bitmaps over the id space; heads never reach beyond n_hub) and one weight look-up per common hub.  What comes out is the list
// eps_scan_screen reports without heads -- screening sums of all paths, same units -- for half the table updates.
#include "scan_common.h"

// heads[v] = {x_v, T_v}: the longest prefix of row v with ids < n_hub whose screening weights sum to <= budget, at most max_rows
// entries (the first rows of a column carry most of its paths; eps_scan_refine pays for every skipped row of a slot that passes:
// resource allocation, where a hub weighs next to nothing, would skip hundreds).
__global__ __launch_bounds__(256) void sp_heads_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                       const uint32_t *__restrict__ fx32, int64_t n_nodes, int32_t n_hub,
                                                       uint32_t budget, uint32_t max_rows, uint2 *__restrict__ heads)
{
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_nodes) return;
    const int64_t b = rowptr[v], e = rowptr[v + 1];
    uint32_t x = 0u, t = 0u;
    for (int64_t i = b; i < e && x < max_rows; ++i) {
        const int32_t w = col[i];
        if (w >= n_hub) break;
        const uint32_t f = fx32[w];
        if (f > budget - t) break;
        t += f;
        ++x;
    }
    heads[v] = make_uint2(x, t);
}

extern "C" int eps_scan_heads(const int64_t *rowptr, const int32_t *col, const uint32_t *fx32, int64_t n_nodes, int32_t n_hub,
                              uint32_t budget, int32_t max_rows, uint32_t *heads, void *stream)
{
    EPS_REQUIRE(n_nodes >= 0 && n_nodes < (1ll << 31) && n_hub >= 0 && max_rows >= 0 && max_rows <= 65535, "eps_scan_heads: bad size");
    if (n_nodes == 0) return EPS_OK;
    EPS_REQUIRE(rowptr && col && fx32 && heads && ((uintptr_t)heads & 7) == 0, "eps_scan_heads: null or misaligned pointer");
    hipLaunchKernelGGL(sp_heads_kernel, dim3((unsigned)((n_nodes + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rowptr, col, fx32,
                       n_nodes, n_hub, budget, (uint32_t)max_rows, (uint2 *)heads);
    EPS_CHECK_LAUNCH("eps_scan_heads");
    return EPS_OK;
}

// hubrows[w * words + (x >> 5)] bit (x & 31) = 1 iff x is a neighbour of hub w < n_hub (words = ceil(n_nodes / 32) rounded up to a
// multiple of 4): the adjacency rows of the first n_hub ids as bitmaps over the id space.  HUB-major on purpose: the slots a
// piece reports share their column (the same few hub rows w) and lie in one id window (nearby bits of those rows), so the
// look-ups of eps_scan_refine hit lines their neighbours just touched.  The table is cleared first, then one wave per hub row sets
// its bits.
#define SH_MAX_HUB 65536       // most hub rows a table may hold
#define SH_LDS_HUB 4096        // eps_scan_refine keeps the weights of this many hubs in LDS
__global__ __launch_bounds__(256) void sp_hub_rows_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                          int32_t n_hub, int64_t words, uint32_t *__restrict__ hubrows)
{
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_hub) return;
    uint32_t *__restrict__ row = hubrows + (size_t)w * words;
    const int64_t b = rowptr[w], e = rowptr[w + 1];
    for (int64_t i = b + lane; i < e; i += 64) {
        const uint32_t x = (uint32_t)col[i];
        atomicOr(&row[x >> 5], 1u << (x & 31u));
    }
}

extern "C" int64_t eps_scan_hub_row_words(int64_t n_nodes) { return ((n_nodes + 31) / 32 + 3) / 4 * 4; }

extern "C" int eps_scan_hub_rows(const int64_t *rowptr, const int32_t *col, int64_t n_nodes, int32_t n_hub, uint32_t *hubrows,
                                 void *stream)
{
    EPS_REQUIRE(n_nodes >= 0 && n_nodes < (1ll << 31), "eps_scan_hub_rows: bad size");
    EPS_REQUIRE(n_hub >= 0 && n_hub <= SH_MAX_HUB && n_hub <= n_nodes, "eps_scan_hub_rows: n_hub must be at most 65536 and at most n_nodes");
    if (n_nodes == 0 || n_hub == 0) return EPS_OK;
    EPS_REQUIRE(rowptr && col && hubrows && ((uintptr_t)hubrows & 15) == 0, "eps_scan_hub_rows: null or misaligned pointer");
    const int64_t words = eps_scan_hub_row_words(n_nodes);
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(hubrows, 0, (size_t)n_hub * (size_t)words * 4, s) != hipSuccess) {
        eps_set_error("eps_scan_hub_rows: cannot clear the table");
        return EPS_ELAUNCH;
    }
    hipLaunchKernelGGL(sp_hub_rows_kernel, dim3((unsigned)((n_hub + 3) / 4)), dim3(256), 0, s, rowptr, col, n_hub, words, hubrows);
    EPS_CHECK_LAUNCH("eps_scan_hub_rows");
    return EPS_OK;
}

// One thread per slot of the walked list: key = (v << 32) | u, val = the WALKED screening sum as raw bits (eps_scan_screen with a
// head table).  The pair's complete screening sum = walked + head(u, v); at or above the bar it goes to `out` (score = sum x
// 2^-shift, like a launch without heads).  A wave collects what passes in its own LDS buffer and reserves room in `out` once
// per few hundred pairs: one atomic per wave and trip on the list's ONE counter would cost 11 ns each, 2.8 ms for the 250 k
// trips of the ppa-like graph's walked list.
#define SH_WBUF 384
__global__ __launch_bounds__(256) void sp_refine_kernel(const eps_survivors *__restrict__ in, const uint2 *__restrict__ heads,
                                                        const uint32_t *__restrict__ hubrows, int32_t n_hub, int64_t words,
                                                        const uint32_t *__restrict__ fx32, const int64_t *__restrict__ rowptr,
                                                        const int32_t *__restrict__ col, int32_t n_nodes, int32_t shift, float scale,
                                                        eps_survivors *__restrict__ out)
{
    __shared__ uint32_t s_fx[SH_LDS_HUB];
    __shared__ int64_t s_key[4][SH_WBUF];
    __shared__ float s_val[4][SH_WBUF];
    const bool fx_lds = n_hub <= SH_LDS_HUB;                     // (a wider hub table: the weights come from the L2)
    for (int i = threadIdx.x; i < n_hub && fx_lds; i += 256) s_fx[i] = fx32[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    int64_t *wkey = s_key[wib];
    float *wval = s_val[wib];
    const uint32_t thr32 = sp_bar_units(out->threshold, shift);
    const unsigned long long handed = in->count;
    const int64_t n = handed < (unsigned long long)in->capacity ? (int64_t)handed : (int64_t)in->capacity;
    const int64_t *__restrict__ in_key = in->key;
    const uint32_t *__restrict__ in_val = (const uint32_t *)in->val;
    int64_t *__restrict__ out_key = out->key;
    float *__restrict__ out_val = out->val;
    const uint32_t out_cap = out->capacity;
    const int64_t stride = (int64_t)gridDim.x * 256;
    int held = 0;                                                // pairs in the wave's buffer (uniform)
    auto flush = [&]() {
        unsigned long long base = 0ull;
        if (lane == 0) base = atomicAdd(&out->count, (unsigned long long)held);
        base = __shfl(base, 0);
        for (int j = lane; j < held; j += 64) {
            const unsigned long long pos = base + (unsigned long long)j;
            if (pos < (unsigned long long)out_cap) {
                out_key[pos] = wkey[j];
                out_val[pos] = wval[j];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (the buffer is rewritten from the front)
        held = 0;
    };
    for (int64_t i0 = (int64_t)blockIdx.x * 256 + (threadIdx.x & ~63); i0 < n; i0 += stride) {
        const int64_t i = i0 + lane;
        const int64_t key = i < n ? in_key[i] : -1;
        uint32_t total = 0u;
        bool pass = false;
        if (key >= 0) {
            const int32_t v = (int32_t)(key >> 32), u = (int32_t)(key & 0xFFFFFFFFll);
            const uint2 hd = heads[v];
            // the head term: the skipped rows of column v are its first hd.x neighbours (hubs, ids < n_hub); row w counts iff u
            // is a neighbour of hub w -- bit u of hub w's bitmap row, four rows in flight
            // From the END of the head -- ids ascend, degrees fall, weights grow: its heaviest rows first -- and only as far as the
            // pair can still reach the bar: `rem` is the weight of the rows not looked at yet (hd.y = their sum at the start), and a
            // slot whose walked sum + found + rem falls below the bar is out.  Most slots pass the walk's lowered threshold by
            // little and share none of the first few rows: they leave after one trip (resource allocation skips dozens of rows
            // per column: every slot used to pay for each).
            uint32_t c = 0u, rem = hd.y;
            const uint32_t s_walk = in_val[i];
            const int32_t *__restrict__ vcol = col + rowptr[v];
            const uint32_t *__restrict__ ubit = hubrows + ((uint32_t)u >> 5);
            bool out_of_reach = thr32 >= SP_FLAG;
            for (int j1 = (int)hd.x; j1 > 0 && !out_of_reach; j1 -= 4) {
                uint32_t wq[4], mq[4], fq[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) wq[q] = j1 - 1 - q >= 0 ? (uint32_t)vcol[j1 - 1 - q] : 0u;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    mq[q] = ubit[(size_t)wq[q] * words];
                    fq[q] = j1 - 1 - q >= 0 ? (fx_lds ? s_fx[wq[q]] : fx32[wq[q]]) : 0u;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    rem -= fq[q];
                    if ((mq[q] >> ((uint32_t)u & 31u)) & 1u) c += fq[q];
                }
                out_of_reach = (unsigned long long)s_walk + c + rem < (unsigned long long)thr32;
            }
            total = s_walk + c;
            pass = !out_of_reach && total >= thr32;
        }
        const unsigned long long m = __ballot(pass);
        if (m) {
            if (pass) {
                const int at = held + __popcll(m & ((1ull << lane) - 1ull));
                wkey[at] = key;
                wval[at] = (float)total * scale;
            }
            held += __popcll(m);
            if (held > SH_WBUF - 64) flush();
        }
    }
    if (held) flush();
}

extern "C" int eps_scan_refine(const eps_survivors *walked, const uint32_t *heads, const uint32_t *hubrows, int32_t n_hub,
                               const uint32_t *fx32, const int64_t *rowptr, const int32_t *col, int64_t n_nodes, int32_t shift,
                               eps_survivors *out, void *stream)
{
    EPS_REQUIRE(n_nodes >= 0 && n_nodes < (1ll << 31) && shift >= 0 && shift <= 40, "eps_scan_refine: bad argument");
    EPS_REQUIRE(n_hub >= 0 && n_hub <= SH_MAX_HUB && n_hub <= n_nodes, "eps_scan_refine: n_hub must be at most 65536 and at most n_nodes");
    EPS_REQUIRE(walked && out && heads && (hubrows || n_hub == 0) && fx32 && rowptr && col, "eps_scan_refine: null pointer");
    EPS_REQUIRE(((uintptr_t)hubrows & 15) == 0 && ((uintptr_t)heads & 7) == 0, "eps_scan_refine: misaligned table");
    hipLaunchKernelGGL(sp_refine_kernel, dim3((unsigned)(eps_num_cus() * 8)), dim3(256), 0, (hipStream_t)stream, walked,
                       (const uint2 *)heads, hubrows, n_hub, eps_scan_hub_row_words(n_nodes), fx32, rowptr, col, (int32_t)n_nodes, shift,
                       ldexpf(1.0f, -shift), out);
    EPS_CHECK_LAUNCH("eps_scan_refine");
    return EPS_OK;
}

__global__ void scan_heads_warm_kernel() {}
extern "C" void eps_warm_scan_heads(void *stream) { hipLaunchKernelGGL(scan_heads_warm_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream); }
