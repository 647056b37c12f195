// (score, id) <-> uint64 sort keys realising the declared tie rule, gfx950.
//
// Replaces `all_scores[:,2].sort(descending=True)` + row gather (filter.py:160-161) for the
// rows rank.py ever consumes (rank.py:294).  The reference's torch.sort is unstable on ties
// (SURVEY 8 trap 9), so the order is DECLARED: score descending, then candidate id ascending
// (== torch.sort(descending=True, stable=True) over the reference's candidate order).
// key (int64, SIGNED order) = (ordered_bits(score) ^ 0x80000000) << 32 | (0xFFFFFFFF - id); a plain descending sort of the keys
// (any algorithm, any sharding) then yields exactly that order, which is what makes the
// multi-GPU top-K merge shard-count invariant.
#include "eps_common.h"
#include <stdlib.h>

__device__ __forceinline__ uint32_t ordered_bits(float f)
{
    f = f + 0.0f;  // -0.0 -> +0.0 so that the two zeros tie, as they do in torch.sort
    const uint32_t b = __builtin_bit_cast(uint32_t, f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float unordered_bits(uint32_t o)
{
    const uint32_t b = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __builtin_bit_cast(float, b);
}

__global__ void pack_keys_kernel(const float *__restrict__ score, const int64_t *__restrict__ id, int64_t id_base,
                                 int64_t n, int64_t *__restrict__ keys)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t gid = (uint64_t)(id ? id[i] : id_base + i);
        keys[i] = (int64_t)(((uint64_t)(ordered_bits(score[i]) ^ 0x80000000u) << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)gid));
    }
}

__global__ void unpack_keys_kernel(const int64_t *__restrict__ keys, int64_t n, float *__restrict__ score,
                                   int64_t *__restrict__ id)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t k = (uint64_t)keys[i];
        if (score) score[i] = unordered_bits((uint32_t)(k >> 32) ^ 0x80000000u);
        if (id) id[i] = (int64_t)(0xFFFFFFFFu - (uint32_t)(k & 0xFFFFFFFFu));
    }
}

static unsigned ew_blocks(int64_t n)
{
    int64_t b = (n + 255) / 256;
    const int64_t cap = (int64_t)eps_num_cus() * 8;
    return (unsigned)(b > cap ? cap : b);
}

extern "C" int eps_pack_keys(const float *score, const int64_t *id_or_null, int64_t id_base, int64_t n,
                             int64_t *keys, void *stream)
{
    EPS_REQUIRE(n >= 0 && id_base >= 0, "eps_pack_keys: negative size");
    if (n == 0) return EPS_OK;
    EPS_REQUIRE(score && keys, "eps_pack_keys: null pointer");
    EPS_REQUIRE(id_or_null || id_base + n <= (1ll << 32), "eps_pack_keys: ids must stay below 2^32");
    hipLaunchKernelGGL(pack_keys_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, score, id_or_null,
                       id_base, n, keys);
    EPS_CHECK_LAUNCH("eps_pack_keys");
    return EPS_OK;
}

extern "C" int eps_unpack_keys(const int64_t *keys, int64_t n, float *score, int64_t *id, void *stream)
{
    EPS_REQUIRE(n >= 0, "eps_unpack_keys: negative size");
    if (n == 0) return EPS_OK;
    EPS_REQUIRE(keys && (score || id), "eps_unpack_keys: null pointer");
    hipLaunchKernelGGL(unpack_keys_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, keys, n, score, id);
    EPS_CHECK_LAUNCH("eps_unpack_keys");
    return EPS_OK;
}

// ---- k-th largest of a float32 array by radix select --------------------------------------------------------------------
// The bar of the threshold scan and the final top-K cut (filter.py:160-161 keeps the K best rows) need one VALUE, not a
// sorted array: four rounds over the order-preserving bit pattern of the floats, 8 bits per round -- a 256-bin histogram of
// the values that still match the prefix found so far, then the bin that holds the k-th largest.  4 reads of the array
// instead of a sort; everything stays on the device (state: prefix, remaining k).
struct kth_state {
    uint32_t prefix, mask;     // bits decided so far (in ordered_bits space) and which bits those are
    uint64_t k;                // rank (1-based, from the top) still to find inside the matching values
    uint32_t hist[256];
};

__global__ void kth_init_kernel(kth_state *__restrict__ st, uint64_t k) { st->k = k; }

__global__ void kth_hist_kernel(const float *__restrict__ x, int64_t n, kth_state *__restrict__ st, int shift)
{
    __shared__ uint32_t h[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) h[i] = 0u;
    __syncthreads();
    const uint32_t prefix = st->prefix, mask = st->mask;
    const int lane = threadIdx.x & 63;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t n_up = (n + 63) & ~63ll;                    // (whole waves run the loop: the ballots below need every lane)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_up; i += stride) {
        const uint32_t o = i < n ? ordered_bits(x[i]) : 0u;
        // (-inf is "no value": an untouched survivor slot.  Left out of the histogram, a list of 2^32 slots times any number of
        //  ranks cannot wrap the 32-bit bins, and "fewer than k values" still answers -inf -- which is what counting them gave)
        const bool live = i < n && o != 0x007FFFFFu && (o & mask) == prefix;
        const uint32_t bin = (o >> shift) & 255u;
        // The leading digits of a score list are nearly constant -- in the first rounds every lane of a wave wants the same
        // bin, and 64 LDS atomics on one address run one after the other.  A wave whose live lanes agree adds their count once.
        const unsigned long long m = __ballot(live);
        if (m) {
            const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)bin, __builtin_ctzll(m));
            if (__ballot(live && bin == b0) == m) {
                if (lane == __builtin_ctzll(m)) atomicAdd(&h[b0], (uint32_t)__popcll(m));
            } else if (live) {
                atomicAdd(&h[bin], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += blockDim.x)
        if (h[i]) atomicAdd(&st->hist[i], h[i]);
}

__global__ void kth_pick_kernel(kth_state *__restrict__ st, int shift, float *__restrict__ out)
{
    // one wave: lane l owns the bins 255 - 4 l .. 252 - 4 l (from the top); a wave scan of the lane totals finds the lane, and
    // that lane the bin, in which the k-th largest falls.  Fewer than k values in all (seen in the first round, whose
    // histogram counts everything): the answer is -inf, and the later rounds leave it alone.
    const int lane = threadIdx.x & 63;
    uint32_t c[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) c[j] = st->hist[255 - 4 * lane - j];
    const uint64_t k = st->k;
    const uint32_t mask = st->mask, prefix = st->prefix;
    uint64_t incl = (uint64_t)c[0] + c[1] + c[2] + c[3];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    const uint64_t total = __shfl(incl, 63);
    __builtin_amdgcn_s_barrier();                             // (one wave: every lane has read its bins before they are cleared)
#pragma unroll
    for (int j = 0; j < 4; ++j) st->hist[255 - 4 * lane - j] = 0u;
    if (mask != 0xFFFFFFFFu) {
        if (shift == 24 && (k == 0 || k > total)) {
            if (lane == 0) {
                st->prefix = ordered_bits(-__builtin_inff());
                st->mask = 0xFFFFFFFFu;
            }
        } else {
            // the first lane (from the top) whose running total reaches k; k > total after the first round cannot happen
            // (the matching values were counted by the round before) -- bin 0 takes the rest, as before
            const unsigned long long reach = __ballot(incl >= k);
            const int owner = reach ? __builtin_ctzll(reach) : 63;
            if (lane == owner) {
                uint64_t kk = k - (incl - ((uint64_t)c[0] + c[1] + c[2] + c[3]));      // rank inside this lane's four bins
                int b = 255 - 4 * lane;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (b == 0 || kk <= c[j]) break;
                    kk -= c[j];
                    --b;
                }
                st->k = kk;
                st->prefix = prefix | ((uint32_t)b << shift);
                st->mask = mask | (255u << shift);
                if (shift == 0 && out) *out = unordered_bits(prefix | (uint32_t)b);
            }
        }
    }
    if (mask == 0xFFFFFFFFu && shift == 0 && out && lane == 0) *out = unordered_bits(prefix);
}

extern "C" int64_t eps_kth_largest_workspace_bytes(void) { return (int64_t)sizeof(kth_state); }

static unsigned kth_blocks(int64_t n)
{
    int64_t blocks = (n + 256 * 16 - 1) / (256 * 16);
    const int64_t cap = (int64_t)eps_num_cus() * 8;
    if (blocks > cap) blocks = cap;
    return (unsigned)(blocks < 1 ? 1 : blocks);
}

extern "C" int eps_kth_largest_f32(const float *x, int64_t n, int64_t k, float *kth, void *workspace, void *stream)
{
    EPS_REQUIRE(n > 0 && k >= 1 && k <= n, "eps_kth_largest_f32: need 1 <= k <= n (k=%lld, n=%lld)", (long long)k, (long long)n);
    EPS_REQUIRE(x && kth && workspace && ((uintptr_t)workspace & 7) == 0, "eps_kth_largest_f32: null or misaligned pointer");
    EPS_REQUIRE(n < (1ll << 32), "eps_kth_largest_f32: histogram counters are 32-bit");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(workspace, 0, sizeof(kth_state), s) != hipSuccess) {
        eps_set_error("eps_kth_largest_f32: cannot initialise the state");
        return EPS_ELAUNCH;
    }
    hipLaunchKernelGGL(kth_init_kernel, dim3(1), dim3(1), 0, s, (kth_state *)workspace, (uint64_t)k);
    for (int shift = 24; shift >= 0; shift -= 8) {
        hipLaunchKernelGGL(kth_hist_kernel, dim3(kth_blocks(n)), dim3(256), 0, s, x, n, (kth_state *)workspace, shift);
        hipLaunchKernelGGL(kth_pick_kernel, dim3(1), dim3(64), 0, s, (kth_state *)workspace, shift, kth);
    }
    EPS_CHECK_LAUNCH("eps_kth_largest_f32");
    return EPS_OK;
}

// ---- the same radix select in steps, for a vector that is spread over the ranks of a job ----------------------------------
// state (device, eps_kth_largest_workspace_bytes() bytes, 8-byte aligned) = { uint32 prefix, mask; uint64 k; uint32 hist[256] }.
// Per round (shift = 24, 16, 8, 0): every rank adds the histogram of ITS values to state.hist (eps_kth_hist_f32), the caller
// sums the 256 counters over the ranks (one all-reduce of 1 KiB: hist starts at byte 16), every rank picks the same bin
// (eps_kth_pick; clears hist).  After the round with shift 0, *out is the k-th largest of the union, or -inf when the union
// holds fewer than k values.
extern "C" int eps_kth_begin(void *state, int64_t k, void *stream)
{
    EPS_REQUIRE(state && ((uintptr_t)state & 7) == 0 && k >= 0, "eps_kth_begin: bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(state, 0, sizeof(kth_state), s) != hipSuccess) {
        eps_set_error("eps_kth_begin: cannot initialise the state");
        return EPS_ELAUNCH;
    }
    hipLaunchKernelGGL(kth_init_kernel, dim3(1), dim3(1), 0, s, (kth_state *)state, (uint64_t)k);
    EPS_CHECK_LAUNCH("eps_kth_begin");
    return EPS_OK;
}

extern "C" int eps_kth_hist_f32(const float *x, int64_t n, void *state, int32_t shift, void *stream)
{
    EPS_REQUIRE(n >= 0 && state && (shift == 24 || shift == 16 || shift == 8 || shift == 0), "eps_kth_hist_f32: bad argument");
    EPS_REQUIRE(n < (1ll << 32), "eps_kth_hist_f32: a rank's vector holds at most 2^32 - 1 values (32-bit bins; n=%lld)", (long long)n);
    if (n == 0) return EPS_OK;
    EPS_REQUIRE(x, "eps_kth_hist_f32: null pointer");
    hipLaunchKernelGGL(kth_hist_kernel, dim3(kth_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, n, (kth_state *)state, (int)shift);
    EPS_CHECK_LAUNCH("eps_kth_hist_f32");
    return EPS_OK;
}

extern "C" int eps_kth_pick(void *state, int32_t shift, float *out_or_null, void *stream)
{
    EPS_REQUIRE(state && (shift == 24 || shift == 16 || shift == 8 || shift == 0), "eps_kth_pick: bad argument");
    hipLaunchKernelGGL(kth_pick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (kth_state *)state, (int)shift, out_or_null);
    EPS_CHECK_LAUNCH("eps_kth_pick");
    return EPS_OK;
}

// ---- radix select + compaction in ONE launch --------------------------------------------------------------------------
// The selections of a filter step (filter.py:160-161 keeps the K best rows; rank.py:294 reads them) are short lists -- a few
// million survivor slots -- and each was ten launches (init, 4 x histogram + pick) plus a compaction: launch latency, not data.
// Here one grid does all of it: four rounds of 256-bin histograms with a grid-wide hand-over between them (every workgroup
// adds its LDS histogram to the round's global one, waits until all have, then picks the bin itself -- the same arithmetic on
// the same counters, so every workgroup agrees without a broadcast), then -- optionally -- the threshold derived from the
// k-th value and the compaction of the entries at or above it.  The grid is at most one workgroup per CU and goes out as a
// COOPERATIVE launch (hipLaunchCooperativeKernel): resident as a whole, so the hand-over cannot deadlock.
//   n = min(*n_dev, n_max) when n_dev is given (the slot counter of an eps_survivors list: its readers stop there);
//   -inf values and entries with key < 0 are "no value" (untouched slots);
//   *kth = the k-th largest value (-inf when fewer than k values);
//   *thr = mode 0: kth; mode 1: the largest float below kth (an inclusive bar for a kernel that keeps scores ABOVE its
//          threshold); mode 2: max(kth - a, kth * b) - |kth| * c (a lower bound of what a screening score kth can be worth
//          exactly: eps_amd.scan.Screen.lower_bound);
//   out_keys / out_vals (optional, out_cap entries each) receive the entries with value >= *thr (and key >= 0), *n_out their
//   number -- which may exceed out_cap: the entries beyond it are counted, not stored (the caller comes back with more room).
struct fsel_state {
    uint32_t hist[4][256];
    uint32_t arrived;
    uint32_t pad;
    unsigned long long n_out;
};

__device__ __forceinline__ void fsel_grid_sync(uint32_t *arrived, uint32_t target)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        atomicAdd(arrived, 1u);
        while (__atomic_load_n(arrived, __ATOMIC_RELAXED) < target) __builtin_amdgcn_s_sleep(2);
        __threadfence();
    }
    __syncthreads();
}

#define FSEL_T 1024
#define FSEL_U 8        // independent loads a thread has in flight per trip

__global__ __launch_bounds__(FSEL_T) void fsel_kernel(const int64_t *__restrict__ keys, const float *__restrict__ vals, int64_t n_max,
                                                     const unsigned long long *__restrict__ n_dev, uint64_t k, int mode, float pa,
                                                     float pb, float pc, fsel_state *__restrict__ st, float *__restrict__ kth_out,
                                                     float *__restrict__ thr_out, int64_t *__restrict__ out_keys,
                                                     float *__restrict__ out_vals, int64_t out_cap, int64_t *__restrict__ n_out)
{
    __shared__ uint32_t h[256];
    __shared__ uint32_t s_prefix, s_mask, s_done;
    __shared__ uint64_t s_k;
    const int tid = threadIdx.x, lane = tid & 63;
    int64_t n = n_max;
    if (n_dev) {
        const unsigned long long c = *n_dev;
        n = c < (unsigned long long)n_max ? (int64_t)c : n_max;
    }
    const int64_t stride = (int64_t)gridDim.x * FSEL_T * FSEL_U;
    uint32_t prefix = 0u, mask = 0u;
    uint64_t kk = k;
    bool none = k == 0;                                        // fewer than k values: the answer is -inf
#pragma unroll 1
    for (int r = 0; r < 4 && k != 0; ++r) {
        const int shift = 24 - 8 * r;
        if (tid < 256) h[tid] = 0u;
        __syncthreads();
        // (a value of -inf is "no value": untouched and dropped slots carry it next to their key -1, so the rounds read the
        //  scores alone; whole waves run the loop: the ballots need every lane)
        for (int64_t base = (int64_t)blockIdx.x * FSEL_T * FSEL_U; base < n; base += stride) {
            float x[FSEL_U];
#pragma unroll
            for (int j = 0; j < FSEL_U; ++j) {
                const int64_t i = base + (int64_t)j * FSEL_T + tid;
                x[j] = i < n ? vals[i] : -__builtin_inff();
            }
#pragma unroll
            for (int j = 0; j < FSEL_U; ++j) {
                const uint32_t o = ordered_bits(x[j]);
                const bool live = o != 0x007FFFFFu && (o & mask) == prefix;
                const uint32_t bin = (o >> shift) & 255u;
                const unsigned long long m = __ballot(live);
                if (m) {                                       // (a wave whose live lanes agree adds their count once)
                    const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)bin, __builtin_ctzll(m));
                    if (__ballot(live && bin == b0) == m) {
                        if (lane == __builtin_ctzll(m)) atomicAdd(&h[b0], (uint32_t)__popcll(m));
                    } else if (live) {
                        atomicAdd(&h[bin], 1u);
                    }
                }
            }
        }
        __syncthreads();
        if (tid < 256 && h[tid]) atomicAdd(&st->hist[r][tid], h[tid]);
        fsel_grid_sync(&st->arrived, (uint32_t)(r + 1) * gridDim.x);
        if (tid < 64) {            // wave 0: lane l owns bins 255 - 4 l .. 252 - 4 l (from the top)
            uint32_t c[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = __atomic_load_n(&st->hist[r][255 - 4 * lane - j], __ATOMIC_RELAXED);
            uint64_t incl = (uint64_t)c[0] + c[1] + c[2] + c[3];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint64_t up = __shfl_up(incl, d);
                if (lane >= d) incl += up;
            }
            const uint64_t total = __shfl(incl, 63);
            if (r == 0 && kk > total) {
                if (lane == 0) s_done = 1u;
            } else {
                if (r == 0 && lane == 0) s_done = 0u;
                const unsigned long long reach = __ballot(incl >= kk);
                const int owner = reach ? __builtin_ctzll(reach) : 63;
                if (lane == owner) {
                    uint64_t k2 = kk - (incl - ((uint64_t)c[0] + c[1] + c[2] + c[3]));
                    int b = 255 - 4 * lane;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (b == 0 || k2 <= c[j]) break;
                        k2 -= c[j];
                        --b;
                    }
                    s_k = k2;
                    s_prefix = prefix | ((uint32_t)b << shift);
                    s_mask = mask | (255u << shift);
                }
            }
        }
        __syncthreads();
        if (r == 0 && s_done) none = true;
        if (none) break;                                       // (uniform over the grid: every workgroup read the same counters)
        prefix = s_prefix;
        mask = s_mask;
        kk = s_k;
        __syncthreads();
    }
    const float kth = none ? -__builtin_inff() : unordered_bits(prefix);
    float thr = kth;
    if (!none && mode == 1) {
        // the largest float below kth (kth is finite here): one step down in the ordered bit pattern
        const uint32_t ob = ordered_bits(kth);
        thr = unordered_bits(ob == 0x80000000u ? ob - 2u : ob - 1u);       // (below +0.0 comes -0.0 == 0: one more step)
    } else if (!none && mode == 2) {
        const float low = kth - pa, rel = kth * pb;
        thr = (low > rel ? low : rel) - __builtin_fabsf(kth) * pc;
    }
    if (blockIdx.x == 0 && tid == 0) {
        if (kth_out) *kth_out = kth;
        if (thr_out) *thr_out = thr;
    }
    if (!out_keys) return;
    // compaction: a wave takes 512 consecutive entries at a time and reserves room for all its hits with one atomic
    const int64_t wave = ((int64_t)blockIdx.x * FSEL_T + tid) >> 6, n_waves = ((int64_t)gridDim.x * FSEL_T) >> 6;
    for (int64_t c0 = wave * 64 * FSEL_U; c0 < n; c0 += n_waves * 64 * FSEL_U) {
        int64_t kq[FSEL_U];
        float sq[FSEL_U];
        unsigned int bits = 0;
#pragma unroll
        for (int j = 0; j < FSEL_U; ++j) {
            const int64_t i = c0 + j * 64 + lane;
            sq[j] = i < n ? vals[i] : -__builtin_inff();
        }
#pragma unroll
        for (int j = 0; j < FSEL_U; ++j) {
            const int64_t i = c0 + j * 64 + lane;
            kq[j] = -1;
            if (i < n && sq[j] >= thr) kq[j] = keys[i];          // (the key is read for the few entries that pass)
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
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(&st->n_out, (unsigned long long)total);
        const unsigned int blo = __shfl((unsigned int)base, 0), bhi = __shfl((unsigned int)(base >> 32), 0);
        unsigned long long pos = (((unsigned long long)bhi << 32) | blo) + (unsigned long long)(incl - cnt);
#pragma unroll
        for (int j = 0; j < FSEL_U; ++j)
            if (bits & (1u << j)) {
                if (pos < (unsigned long long)out_cap) {       // (entries past the arrays' room are counted, not stored)
                    out_keys[pos] = kq[j];
                    out_vals[pos] = sq[j];
                }
                ++pos;
            }
    }
    // the last workgroup to finish publishes the count (st->n_out is complete then)
    __syncthreads();
    if (tid == 0) {
        __threadfence();
        const uint32_t done = atomicAdd(&st->arrived, 1u) + 1u;
        // (rounds that ran: 4 unless the list held fewer than k values)
        const uint32_t rounds_run = none && k != 0 ? 1u : (k == 0 ? 0u : 4u);
        if (done == (rounds_run + 1u) * gridDim.x && n_out) *n_out = (int64_t)__atomic_load_n(&st->n_out, __ATOMIC_RELAXED);
    }
}

extern "C" int64_t eps_select_compact_workspace_bytes(void) { return (int64_t)sizeof(fsel_state); }

// k >= 0 (0: no selection, *kth = *thr = -inf: every entry with key >= 0 is kept); keys may be NULL when there is no compaction
// (then every slot counts, -inf apart); out_keys / out_vals NULL: selection only.  state: eps_select_compact_workspace_bytes()
// bytes, 8-byte aligned, for this call alone until the stream has passed it.
extern "C" int eps_select_compact(const int64_t *keys, const float *vals, int64_t n_max, const unsigned long long *n_dev_or_null,
                                  int64_t k, int32_t mode, float pa, float pb, float pc, float *kth_or_null, float *thr_or_null,
                                  int64_t *out_keys_or_null, float *out_vals_or_null, int64_t out_cap, int64_t *n_out_or_null,
                                  void *state, void *stream)
{
    EPS_REQUIRE(out_cap >= 0, "eps_select_compact: negative room");
    EPS_REQUIRE(n_max >= 0 && n_max < (1ll << 32) && k >= 0 && mode >= 0 && mode <= 2, "eps_select_compact: bad argument");
    EPS_REQUIRE(state && ((uintptr_t)state & 7) == 0, "eps_select_compact: null or misaligned state");
    EPS_REQUIRE((out_keys_or_null == nullptr) == (out_vals_or_null == nullptr), "eps_select_compact: out_keys and out_vals come together");
    EPS_REQUIRE(!out_keys_or_null || (keys && n_out_or_null), "eps_select_compact: the compaction needs keys and n_out");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(state, 0, sizeof(fsel_state), s) != hipSuccess ||
        (n_out_or_null && hipMemsetAsync(n_out_or_null, 0, sizeof(int64_t), s) != hipSuccess)) {
        eps_set_error("eps_select_compact: cannot initialise the state");
        return EPS_ELAUNCH;
    }
    EPS_REQUIRE(n_max == 0 || vals, "eps_select_compact: null pointer");
    // k == 0: the rounds are skipped inside (none = true from the start) -- the kernel's `rounds_run` bookkeeping counts on it
    // (at least 64 K entries per workgroup: a grid-wide hand-over costs the more the more workgroups take part, and a list of a
    //  couple of million entries is four of them long -- fsel_kernel averaged 0.21 ms per call in the r04 step with one
    //  workgroup per 8 K entries)
    int64_t blocks = (n_max + FSEL_T * 64 - 1) / (FSEL_T * 64);
    const int64_t cap = (int64_t)eps_num_cus();                // one workgroup per CU at most
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    // A COOPERATIVE launch: the runtime validates that the whole grid fits the device at once, so the hand-over between the rounds
    // cannot wait for a workgroup of THIS launch that was never placed (r04 launched it plainly and relied on "one workgroup per
    // CU is always resident").  That is a guarantee about this process's launch only: CUs held by a persistent
    // kernel of ANOTHER process on the same device (ranks sharing a GPU: the --one-device tests) are not reserved against, and a
    // hand-over can then stall until that kernel lets go.  For that set-up EPS_SELECT_ONE_WORKGROUP=1 in the environment selects
    // the single-workgroup grid (no hand-over between workgroups at all: slow, never stuck) -- also what a refused grid falls back to.
    {
        static const int one = [] { const char *e = getenv("EPS_SELECT_ONE_WORKGROUP"); return e && e[0] == '1' ? 1 : 0; }();
        if (one) blocks = 1;
    }
    uint64_t k64 = (uint64_t)k;
    int mode_i = (int)mode;
    fsel_state *st = (fsel_state *)state;
    void *args[] = {(void *)&keys, (void *)&vals, (void *)&n_max, (void *)&n_dev_or_null, (void *)&k64, (void *)&mode_i, (void *)&pa,
                    (void *)&pb, (void *)&pc, (void *)&st, (void *)&kth_or_null, (void *)&thr_or_null, (void *)&out_keys_or_null,
                    (void *)&out_vals_or_null, (void *)&out_cap, (void *)&n_out_or_null};
    hipError_t err = hipLaunchCooperativeKernel((const void *)fsel_kernel, dim3((unsigned)blocks), dim3(FSEL_T), args, 0, s);
    if (err != hipSuccess && blocks > 1) {
        (void)hipGetLastError();
        err = hipLaunchCooperativeKernel((const void *)fsel_kernel, dim3(1), dim3(FSEL_T), args, 0, s);
    }
    if (err != hipSuccess) {
        (void)hipGetLastError();
        eps_set_error("eps_select_compact: cooperative launch failed: %s", hipGetErrorString(err));
        return EPS_ELAUNCH;
    }
    return EPS_OK;
}

// (one empty kernel per translation unit: launching it makes the HIP runtime load this unit's code object -- eps_warm_up)
__global__ void topk_keys_warm_kernel() {}
extern "C" void eps_warm_topk_keys(void *stream)
{
    // (a COOPERATIVE launch: what eps_select_compact issues -- the runtime's one-off set-up for that launch type rides with the
    //  warm-up instead of the first select of a one-shot run)
    void *no_args[] = {nullptr};
    if (hipLaunchCooperativeKernel((const void *)topk_keys_warm_kernel, dim3(1), dim3(64), no_args, 0, (hipStream_t)stream) != hipSuccess) {
        (void)hipGetLastError();
        hipLaunchKernelGGL(topk_keys_warm_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream);
    }
}
