// Dense float32 GEMM on the f32-input MFMA (v_mfma_f32_32x32x2_f32: exact f32, bitwise a
// k-ordered fmaf chain), gfx950.
//
//   C[M,N] = act(A[M,K] * B[N,K]^T + bias[N])      (B in torch.nn.Linear layout)
//
// Serves the dense halves of the GNN layers: `x @ W` of GCNConv (models.py:183/186, weight
// passed transposed by the host) and lin_l / lin_r of SAGEConv (models.py:436/439).
//
// 128x128 block tile, BK = 32, four waves of 64x64 (2x2 MFMA 32x32 tiles, 64 accumulator
// registers).  Both operands are k-contiguous, so a lane's ds_read_b128 at [row][8j+4h..+3]
// feeds four consecutive MFMA k-steps of A and of B alike (the two lane halves own k-subsets
// {8j..8j+3} and {8j+4..8j+7}; A and B use the same assignment, so the sum over k is complete).
// LDS rows are padded to 36 floats: the 16-lane ds_read_b128 groups then hit 16 distinct 16-B
// bank slots.  Global->register->LDS double buffering, one barrier per K-chunk.
#include "eps_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define G_BM 128
#define G_BN 128
#define G_BK 32
#define G_LD 36

__device__ __forceinline__ float4 load4_guard(const float *__restrict__ base, int64_t row, int64_t nrows, int64_t ld,
                                              int k, int kmax, bool fast)
{
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < nrows) {
        const float *p = base + row * ld + k;
        if (fast && k + 3 < kmax) {
            r = *reinterpret_cast<const float4 *>(p);
        } else {
            if (k + 0 < kmax) r.x = p[0];
            if (k + 1 < kmax) r.y = p[1];
            if (k + 2 < kmax) r.z = p[2];
            if (k + 3 < kmax) r.w = p[3];
        }
    }
    return r;
}

__global__ __launch_bounds__(256) void gemm_f32_kernel(const float *__restrict__ A, int64_t lda,
                                                       const float *__restrict__ B, int64_t ldb,
                                                       const float *__restrict__ bias, int relu, int accumulate,
                                                       float *__restrict__ C, int64_t ldc, int64_t M, int32_t N,
                                                       int32_t K, int fastA, int fastB)
{
    __shared__ __attribute__((aligned(16))) float As[2][G_BM][G_LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][G_BN][G_LD];

    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = w >> 1, wn = w & 1;
    const int n_nblk = (N + G_BN - 1) / G_BN;
    const int64_t m0 = (int64_t)(blockIdx.x / n_nblk) * G_BM;
    const int n0 = (blockIdx.x % n_nblk) * G_BN;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = (K + G_BK - 1) / G_BK;
    float4 ra[4], rb[4];

    auto gload = [&](int kc) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + 256 * i;
            const int row = q >> 3, c4 = q & 7;
            ra[i] = load4_guard(A, m0 + row, M, lda, kc * G_BK + c4 * 4, K, fastA);
            rb[i] = load4_guard(B, n0 + row, N, ldb, kc * G_BK + c4 * 4, K, fastB);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + 256 * i;
            const int row = q >> 3, c4 = q & 7;
            *reinterpret_cast<float4 *>(&As[buf][row][c4 * 4]) = ra[i];
            *reinterpret_cast<float4 *>(&Bs[buf][row][c4 * 4]) = rb[i];
        }
    };

    gload(0);
    lstore(0);
    __syncthreads();

    for (int kc = 0; kc < nk; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < nk) gload(kc + 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ko = 8 * j + 4 * h;
            const float4 a0 = *reinterpret_cast<const float4 *>(&As[buf][wm * 64 + r][ko]);
            const float4 a1 = *reinterpret_cast<const float4 *>(&As[buf][wm * 64 + 32 + r][ko]);
            const float4 b0 = *reinterpret_cast<const float4 *>(&Bs[buf][wn * 64 + r][ko]);
            const float4 b1 = *reinterpret_cast<const float4 *>(&Bs[buf][wn * 64 + 32 + r][ko]);
            const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
            const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[s], bv0[s], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[s], bv1[s], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[s], bv0[s], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[s], bv1[s], acc[1][1], 0, 0, 0);
            }
        }
        if (kc + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }

    // epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int cc = n0 + wn * 64 + ni * 32 + r;
            if (cc >= N) continue;
            const float bv = bias ? bias[cc] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t rr = m0 + wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (rr < M) {
                    float t = acc[mi][ni][e] + bv;
                    float *cp = C + rr * ldc + cc;
                    if (accumulate) t += *cp;
                    if (relu) t = t > 0.f ? t : 0.f;
                    *cp = t;
                }
            }
        }
}

extern "C" int eps_gemm_f32(const float *a, int64_t lda, const float *b, int64_t ldb, const float *bias, int relu,
                            int accumulate, float *c, int64_t ldc, int64_t m, int32_t n, int32_t k, void *stream)
{
    EPS_REQUIRE(m >= 0 && n >= 0 && k >= 0 && lda >= k && ldb >= k && ldc >= n,
                "eps_gemm_f32: bad shape (m=%lld n=%d k=%d lda=%lld ldb=%lld ldc=%lld)", (long long)m, n, k,
                (long long)lda, (long long)ldb, (long long)ldc);
    if (m == 0 || n == 0) return EPS_OK;
    EPS_REQUIRE(a && b && c, "eps_gemm_f32: null pointer");
    const int fastA = (lda % 4 == 0) && ((uintptr_t)a % 16 == 0);
    const int fastB = (ldb % 4 == 0) && ((uintptr_t)b % 16 == 0);
    const int64_t mblk = (m + G_BM - 1) / G_BM;
    const int64_t nblk = (n + G_BN - 1) / G_BN;
    EPS_REQUIRE(mblk * nblk < (1ll << 31), "eps_gemm_f32: grid too large");
    hipLaunchKernelGGL(gemm_f32_kernel, dim3((unsigned)(mblk * nblk)), dim3(256), 0, (hipStream_t)stream, a, lda, b,
                       ldb, bias, relu, accumulate, c, ldc, m, n, k, fastA, fastB);
    EPS_CHECK_LAUNCH("eps_gemm_f32");
    return EPS_OK;
}
