// CSR x dense SpMM with fused epilogue (GCNConv / SAGEConv aggregate), gfx950, float32.
//
// Replaces torch_sparse spmm_sum / spmm_mean inside GCNConv / SAGEConv (models.py:183-186,
// :436-439; SAGE semantics witnessed by models.py:380-384) plus the bias add and ReLU of the
// layer loops (models.py:184, :437).
//
// HBM-bound gather: one wave per output row; the 64 lanes cover 256 feature columns as
// float4, so every neighbour costs exactly one coalesced 1-KiB row read (global_load_dwordx4).
// Column indices / values are fetched 64 at a time (coalesced) and broadcast with
// v_readlane; 32 neighbour rows (32 KiB) are kept in flight per wave.  Accumulation is sequential in
// ascending neighbour order, like the reference's CPU kernel.  No MFMA: this is a gather.
#include "eps_common.h"

#include <type_traits>

#ifndef SP_FLIGHT
#define SP_FLIGHT 32   // neighbour rows in flight per wave (2: 7.3 ms, 4: 6.6, 8: 6.5, 16: 6.3, 32: 6.0 per ppa-like layer)
#endif

template <int VEC>
struct VecT;
template <>
struct VecT<4> {
    using type = float4;
};
template <>
struct VecT<1> {
    using type = float;
};

__device__ __forceinline__ void fma_vec(float4 &a, float s, const float4 &x)
{
    a.x = fmaf(s, x.x, a.x);
    a.y = fmaf(s, x.y, a.y);
    a.z = fmaf(s, x.z, a.z);
    a.w = fmaf(s, x.w, a.w);
}
__device__ __forceinline__ void fma_vec(float &a, float s, const float &x) { a = fmaf(s, x, a); }
__device__ __forceinline__ float4 vzero(float4) { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float vzero(float) { return 0.f; }

template <int VEC, bool HAS_VAL>
__global__ __launch_bounds__(256) void spmm_csr_kernel(const int64_t *__restrict__ rowptr,
                                                       const int32_t *__restrict__ col,
                                                       const float *__restrict__ val, int64_t n_rows,
                                                       const float *__restrict__ x, int64_t ldx, int32_t f,
                                                       const float *__restrict__ bias, int relu, int mean,
                                                       float *__restrict__ y, int64_t ldy)
{
    using V = typename VecT<VEC>::type;
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;

    for (int64_t r = wave; r < n_rows; r += n_waves) {
        const int64_t b = rowptr[r], e = rowptr[r + 1];
        const float cnt = (float)((e - b) > 0 ? (e - b) : 1);  // mean = sum / count (a true division, like the reference)
        for (int32_t c0 = 0; c0 < f; c0 += 64 * VEC) {
            const int32_t c = c0 + lane * VEC;
            const bool act = c < f;
            V acc = vzero(V());
            for (int64_t k0 = b; k0 < e; k0 += 64) {
                const int nk = (e - k0) < 64 ? (int)(e - k0) : 64;
                const int my_col = lane < nk ? col[k0 + lane] : 0;
                float my_val = 1.0f;
                if (HAS_VAL) my_val = lane < nk ? val[k0 + lane] : 0.0f;
                // SP_FLIGHT neighbour rows in flight per wave.  Full groups run unguarded; the last, partial group of a row
                // takes the guarded form (wave-uniform guards = scalar branches), so a row's tail costs one round trip too.
                auto group = [&](int j, auto guarded) {
                    constexpr bool G = decltype(guarded)::value;
                    int cj[SP_FLIGHT];
                    V xv[SP_FLIGHT];
                    float vj[SP_FLIGHT];
#pragma unroll
                    for (int q = 0; q < SP_FLIGHT; ++q) cj[q] = __builtin_amdgcn_readlane(my_col, (j + q) & 63);
#pragma unroll
                    for (int q = 0; q < SP_FLIGHT; ++q) {
                        xv[q] = vzero(V());
                        if ((!G || j + q < nk) && act) xv[q] = *reinterpret_cast<const V *>(x + (int64_t)cj[q] * ldx + c);
                    }
#pragma unroll
                    for (int q = 0; q < SP_FLIGHT; ++q) {
                        vj[q] = 1.f;
                        if (HAS_VAL)
                            vj[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_val), (j + q) & 63));
                    }
#pragma unroll
                    for (int q = 0; q < SP_FLIGHT; ++q)
                        if (!G || j + q < nk) fma_vec(acc, vj[q], xv[q]);
                };
                int j = 0;
                for (; j + SP_FLIGHT <= nk; j += SP_FLIGHT) group(j, std::false_type{});
                if (j < nk) group(j, std::true_type{});
            }
            if (act) {
                float a[VEC];
                if constexpr (VEC == 4) { a[0] = acc.x; a[1] = acc.y; a[2] = acc.z; a[3] = acc.w; }
                else a[0] = acc;
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    float t = a[i];
                    if (mean) t = t / cnt;
                    if (bias) t += bias[c + i];
                    if (relu) t = t > 0.f ? t : 0.f;
                    a[i] = t;
                }
                if constexpr (VEC == 4) *reinterpret_cast<float4 *>(y + r * ldy + c) = make_float4(a[0], a[1], a[2], a[3]);
                else y[r * ldy + c] = a[0];
            }
        }
    }
}

extern "C" int eps_spmm_csr(const int64_t *rowptr, const int32_t *col, const float *val, int64_t n_rows,
                            const float *x, int64_t ldx, int32_t f, const float *bias, int relu, int mean,
                            float *y, int64_t ldy, void *stream)
{
    EPS_REQUIRE(n_rows >= 0 && f >= 0 && ldx >= f && ldy >= f, "eps_spmm_csr: bad shape (n_rows=%lld f=%d ldx=%lld ldy=%lld)",
                (long long)n_rows, f, (long long)ldx, (long long)ldy);
    if (n_rows == 0 || f == 0) return EPS_OK;
    EPS_REQUIRE(rowptr && col && x && y, "eps_spmm_csr: null pointer");
    const bool vec4 = (f % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && (((uintptr_t)x | (uintptr_t)y) % 16 == 0);
    int64_t blocks = (n_rows + 3) / 4;
    const int64_t cap = (int64_t)eps_num_cus() * 8 * 4;  // grid-stride beyond 8 resident blocks/CU x4 for balance
    if (blocks > cap) blocks = cap;
    dim3 grid((unsigned)blocks), block(256);
    hipStream_t s = (hipStream_t)stream;
    const bool hv = (val != nullptr) && !mean;
#define SP_LAUNCH(VEC, HV) \
    hipLaunchKernelGGL((spmm_csr_kernel<VEC, HV>), grid, block, 0, s, rowptr, col, val, n_rows, x, ldx, f, bias, relu, mean, y, ldy)
    if (vec4 && hv) SP_LAUNCH(4, true);
    else if (vec4) SP_LAUNCH(4, false);
    else if (hv) SP_LAUNCH(1, true);
    else SP_LAUNCH(1, false);
#undef SP_LAUNCH
    EPS_CHECK_LAUNCH("eps_spmm_csr");
    return EPS_OK;
}

// ------------------------------------------------------------------------------ gcn_norm
// torch_geometric 1.7.0 gcn_norm on a SparseTensor (third-party; restated): deg = rowsum(A^),
// dis = deg^-1/2 with inf -> 0, val' = (val * dis[row]) * dis[col].
__global__ void gcn_dis_kernel(const int64_t *__restrict__ rowptr, const float *__restrict__ val, int64_t n_rows,
                               float *__restrict__ dis)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < n_rows; r += n_waves) {
        const int64_t b = rowptr[r], e = rowptr[r + 1];
        float s = 0.f;
        if (val) {
            for (int64_t k = b + lane; k < e; k += 64) s += val[k];
            s = eps_wave_sum(s);
        } else {
            s = (float)(e - b);
        }
        float d = powf(s, -0.5f);
        if (isinf(d)) d = 0.f;
        if (lane == 0) dis[r] = d;
    }
}

__global__ void gcn_scale_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                 const float *__restrict__ val, int64_t n_rows, const float *__restrict__ dis,
                                 float *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < n_rows; r += n_waves) {
        const int64_t b = rowptr[r], e = rowptr[r + 1];
        const float dr = dis[r];
        for (int64_t k = b + lane; k < e; k += 64) out[k] = ((val ? val[k] : 1.0f) * dr) * dis[col[k]];
    }
}

extern "C" int eps_gcn_norm(const int64_t *rowptr, const int32_t *col, const float *val, int64_t n_rows, float *dis,
                            float *val_out, void *stream)
{
    EPS_REQUIRE(n_rows >= 0, "eps_gcn_norm: negative size");
    if (n_rows == 0) return EPS_OK;
    EPS_REQUIRE(rowptr && col && dis && val_out, "eps_gcn_norm: null pointer");
    int64_t blocks = (n_rows + 3) / 4;
    const int64_t cap = (int64_t)eps_num_cus() * 16;
    if (blocks > cap) blocks = cap;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gcn_dis_kernel, dim3((unsigned)blocks), dim3(256), 0, s, rowptr, val, n_rows, dis);
    hipLaunchKernelGGL(gcn_scale_kernel, dim3((unsigned)blocks), dim3(256), 0, s, rowptr, col, val, n_rows, dis, val_out);
    EPS_CHECK_LAUNCH("eps_gcn_norm");
    return EPS_OK;
}

// (one empty kernel per translation unit: launching it makes the HIP runtime load this unit's code object -- eps_warm_up)
__global__ void spmm_csr_warm_kernel() {}
extern "C" void eps_warm_spmm_csr(void *stream) { hipLaunchKernelGGL(spmm_csr_warm_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream); }
