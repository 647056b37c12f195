// Dense float32 GEMM on the f32-input MFMA (v_mfma_f32_32x32x2_f32: exact f32, bitwise a
// k-ordered fmaf chain), gfx950.
//
//   C[M,N] = act(A[M,K] * B[N,K]^T + bias[N])      (B in torch.nn.Linear layout)
//
// Serves the dense halves of the GNN layers: `x @ W` of GCNConv (models.py:183/186, weight
// passed transposed by the host) and lin_l / lin_r of SAGEConv (models.py:436/439).
//
// 128x128 block tile, BK = 16, four waves of 64x64 (2x2 MFMA 32x32 tiles, 64 accumulator
// registers).  Both operands are k-contiguous, so a lane's ds_read_b128 at [row][8j+4h..+3]
// feeds four consecutive MFMA k-steps of A and of B alike (the two lane halves own k-subsets
// {8j..8j+3} and {8j+4..8j+7}; A and B use the same assignment, so the sum over k is complete).
// LDS rows are padded by 4 floats (BK + 4): the 16-lane ds_read_b128 groups then hit 16 distinct 16-B
// bank slots.  Global->register->LDS double buffering, one barrier per K-chunk.
#include "eps_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define G_BM 128
#define G_BN 128
#ifndef G_BK
#define G_BK 16  // 40 KiB of LDS per workgroup -> four resident workgroups per CU on the aligned path (128 VGPRs); 32 and 64 measured slower
#endif
#define G_LD (G_BK + 4)
#define G_F4 (G_BK / 4)          // float4 per tile row
#define G_NLD (G_BM * G_F4 / 256)  // float4 per thread and operand
#define G_NJ (G_BK / 8)

typedef float v4f __attribute__((ext_vector_type(4)));

// One float4 of an operand tile through a raw buffer descriptor that covers rows [row0, nrows) of the matrix:
// rows past the end fall outside the descriptor and read as zeros (no exec-mask branch), the K tail is pushed out
// of range by a select on the offset.  `fast` == rows 16-byte aligned (ld % 4 == 0); otherwise four dword loads.
template <bool ALIGNED>
__device__ __forceinline__ v4f tile_load4(__amdgpu_buffer_rsrc_t rs, int row, int ld, int k, int kmax, bool fast)
{
    const int off = (row * ld + k) * 4;
    if (ALIGNED) return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs, k + 3 < kmax ? off : 0x7ffffff0, 0, 0));
    if (fast) {
        v4f t = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs, k + 3 < kmax ? off : 0x7ffffff0, 0, 0));
        if (k < kmax && k + 3 >= kmax) {  // straddling float4 of the K tail (K % 4 != 0): element-wise
            t.x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
            t.y = k + 1 < kmax ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off + 4, 0, 0)) : 0.f;
            t.z = k + 2 < kmax ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off + 8, 0, 0)) : 0.f;
            t.w = 0.f;
        }
        return t;
    }
    v4f t;
    t.x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, k + 0 < kmax ? off : 0x7ffffff0, 0, 0));
    t.y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, k + 1 < kmax ? off + 4 : 0x7ffffff0, 0, 0));
    t.z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, k + 2 < kmax ? off + 8 : 0x7ffffff0, 0, 0));
    t.w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, k + 3 < kmax ? off + 12 : 0x7ffffff0, 0, 0));
    return t;
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const float *base, int64_t row0, int64_t nrows, int64_t ld)
{
    int64_t bytes = (nrows - row0) * ld * 4;          // rows [row0, nrows): only the block's 128 rows are ever addressed
    const int64_t cap = (int64_t)G_BM * ld * 4 + 64;  // ... so the descriptor never needs to exceed one tile (+ slack)
    if (bytes > cap) bytes = cap;
    if (bytes < 0) bytes = 0;
    return __builtin_amdgcn_make_buffer_rsrc((void *)(base + row0 * ld), 0, (int)bytes, 0x00020000);
}

// ALIGNED: every operand row and the C rows are 16-byte aligned and K % 4 == 0 (every layer of the recipes): no
// element-wise tail paths, which keeps the kernel at 128 VGPRs = four workgroups (16 waves) per CU instead of three
// (r02 lab, tools/gemm_lab.hip: 104 -> 110 TF), and the MFMA groups run at raised wave priority so that a wave with
// matrix work is issued ahead of the waves that are staging operands (-> 118 TF).
template <bool ALIGNED>
__global__ __launch_bounds__(256, ALIGNED ? 4 : 2) void gemm_f32_kernel(const float *__restrict__ A, int64_t lda,
                                                       const float *__restrict__ B, int64_t ldb,
                                                       const float *__restrict__ bias, int relu, int accumulate,
                                                       float *__restrict__ C, int64_t ldc, int64_t M, int32_t N,
                                                       int32_t K, int fastA_, int fastB_, int fastC_)
{
    const bool fastA = ALIGNED || fastA_, fastB = ALIGNED || fastB_, fastC = ALIGNED || fastC_;
    // one LDS block: the two operand double buffers, and -- after the last K-chunk -- the staging tile of the epilogue
    __shared__ __attribute__((aligned(16))) float smem[2 * (G_BM + G_BN) * G_LD];
    float(*As)[G_BM][G_LD] = reinterpret_cast<float(*)[G_BM][G_LD]>(smem);
    float(*Bs)[G_BN][G_LD] = reinterpret_cast<float(*)[G_BN][G_LD]>(smem + 2 * G_BM * G_LD);
    static_assert(64 * (G_BN + 4) <= 2 * (G_BM + G_BN) * G_LD, "the epilogue's staging tile must fit the operand buffers");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wm = w >> 1, wn = w & 1;
    const int n_nblk = (N + G_BN - 1) / G_BN;
    const int64_t m0 = (int64_t)(blockIdx.x / n_nblk) * G_BM;
    const int n0 = (blockIdx.x % n_nblk) * G_BN;
    // (flag bit 1 of `relu`: the caller wants the tiles on and below the diagonal only -- a symmetric product, e.g. A A^T of a
    //  dense adjacency: csrc/dense_cn.hip -- so a workgroup whose tile lies strictly above it has nothing to do)
    if ((relu & 2) && (int64_t)n0 > m0) return;
    relu &= 1;
    const __amdgpu_buffer_rsrc_t ra_rs = tile_rsrc(A, m0, M, lda);
    const __amdgpu_buffer_rsrc_t rb_rs = tile_rsrc(B, n0, N, ldb);
    const int ilda = (int)lda, ildb = (int)ldb;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = (K + G_BK - 1) / G_BK;
    v4f ra[G_NLD], rb[G_NLD];

#define G_GLOAD(kc)                                                                             \
    _Pragma("unroll") for (int i = 0; i < G_NLD; ++i)                                           \
    {                                                                                           \
        const int q = tid + 256 * i;                                                            \
        const int row = q / G_F4, c4 = q % G_F4;                                                \
        ra[i] = tile_load4<ALIGNED>(ra_rs, row, ilda, (kc)*G_BK + c4 * 4, K, fastA);                     \
        rb[i] = tile_load4<ALIGNED>(rb_rs, row, ildb, (kc)*G_BK + c4 * 4, K, fastB);                     \
    }
#define G_LSTORE(buf)                                                                           \
    _Pragma("unroll") for (int i = 0; i < G_NLD; ++i)                                           \
    {                                                                                           \
        const int q = tid + 256 * i;                                                            \
        const int row = q / G_F4, c4 = q % G_F4;                                                \
        *reinterpret_cast<v4f *>(&As[buf][row][c4 * 4]) = ra[i];                                \
        *reinterpret_cast<v4f *>(&Bs[buf][row][c4 * 4]) = rb[i];                                \
    }

    G_GLOAD(0);
    G_LSTORE(0);
    __syncthreads();

    for (int kc = 0; kc < nk; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < nk) { G_GLOAD(kc + 1); }
        // all fragment reads of the chunk up front; MFMAs drain them behind counted lgkmcnt waits
        float4 a0[G_NJ], a1[G_NJ], b0[G_NJ], b1[G_NJ];
#pragma unroll
        for (int j = 0; j < G_NJ; ++j) {
            const int ko = 8 * j + 4 * h;
            a0[j] = *reinterpret_cast<const float4 *>(&As[buf][wm * 64 + r][ko]);
            a1[j] = *reinterpret_cast<const float4 *>(&As[buf][wm * 64 + 32 + r][ko]);
            b0[j] = *reinterpret_cast<const float4 *>(&Bs[buf][wn * 64 + r][ko]);
            b1[j] = *reinterpret_cast<const float4 *>(&Bs[buf][wn * 64 + 32 + r][ko]);
        }
#pragma unroll
        for (int j = 0; j < G_NJ; ++j) {
            const float av0[4] = {a0[j].x, a0[j].y, a0[j].z, a0[j].w}, av1[4] = {a1[j].x, a1[j].y, a1[j].z, a1[j].w};
            const float bv0[4] = {b0[j].x, b0[j].y, b0[j].z, b0[j].w}, bv1[4] = {b1[j].x, b1[j].y, b1[j].z, b1[j].w};
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[s], bv0[s], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[s], bv1[s], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[s], bv0[s], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[s], bv1[s], acc[1][1], 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
            if (j == G_NJ / 2) {
                // the other LDS buffer has been free since the last barrier: park the next chunk there while the
                // matrix pipe still has a quarter of this chunk queued, so the barrier below finds the writes done
                __builtin_amdgcn_sched_barrier(0);
                if (kc + 1 < nk) { G_LSTORE(buf ^ 1); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }
#undef G_GLOAD
#undef G_LSTORE

    // epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
    // Written straight from the accumulators that is 64 dword stores per lane: the store ISSUE, not the bandwidth, sets the
    // tail of every tile.  With 16-byte aligned C rows the tile goes through LDS instead (the operand buffers are free
    // after the last barrier): two passes of 64 rows, accumulators -> Cs[64][128+4] (conflict-free dword writes), then every
    // thread stores 8 float4 -- lanes 0-31 one 512-byte row, lanes 32-63 the next -- bias / accumulate / ReLU applied there.
    if (fastC) {
        float(*Cs)[G_BN + 4] = reinterpret_cast<float(*)[G_BN + 4]>(smem);   // 64 x 132 floats = 33 KiB of the 40
        const int c4 = (tid & 31) * 4;
        const int cc = n0 + c4;
        v4f bv = {0.f, 0.f, 0.f, 0.f};
        if (bias && cc < N) bv = *reinterpret_cast<const v4f *>(bias + cc);   // N % 4 == 0 on this path
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if (wm == pass) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                        for (int e = 0; e < 16; ++e)
                            Cs[mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * h][wn * 64 + ni * 32 + r] = acc[mi][ni][e];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int rl = (tid >> 5) + 8 * i;
                const int64_t rr = m0 + pass * 64 + rl;
                if (rr < M && cc < N) {
                    v4f t = *reinterpret_cast<const v4f *>(&Cs[rl][c4]) + bv;
                    v4f *cp = reinterpret_cast<v4f *>(C + rr * ldc + cc);
                    if (accumulate) t += *cp;
                    if (relu) {
                        t.x = t.x > 0.f ? t.x : 0.f;
                        t.y = t.y > 0.f ? t.y : 0.f;
                        t.z = t.z > 0.f ? t.z : 0.f;
                        t.w = t.w > 0.f ? t.w : 0.f;
                    }
                    *cp = t;
                }
            }
            if (pass == 0) __syncthreads();
        }
        return;
    }
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
    const int fastC = (ldc % 4 == 0) && ((uintptr_t)c % 16 == 0) && (n % 4 == 0) && (!bias || (uintptr_t)bias % 16 == 0);
    const int64_t mblk = (m + G_BM - 1) / G_BM;
    const int64_t nblk = (n + G_BN - 1) / G_BN;
    EPS_REQUIRE(mblk * nblk < (1ll << 31), "eps_gemm_f32: grid too large");
    EPS_REQUIRE(lda < (1 << 22) && ldb < (1 << 22), "eps_gemm_f32: leading dimension too large for 32-bit tile offsets");
    if (fastA && fastB && fastC && k % 4 == 0)
        hipLaunchKernelGGL(gemm_f32_kernel<true>, dim3((unsigned)(mblk * nblk)), dim3(256), 0, (hipStream_t)stream, a, lda, b,
                           ldb, bias, relu, accumulate, c, ldc, m, n, k, 1, 1, 1);
    else
        hipLaunchKernelGGL(gemm_f32_kernel<false>, dim3((unsigned)(mblk * nblk)), dim3(256), 0, (hipStream_t)stream, a, lda, b,
                           ldb, bias, relu, accumulate, c, ldc, m, n, k, fastA, fastB, fastC);
    EPS_CHECK_LAUNCH("eps_gemm_f32");
    return EPS_OK;
}

// (one empty kernel per translation unit: launching it makes the HIP runtime load this unit's code object -- eps_warm_up)
__global__ void gemm_f32_warm_kernel() {}
extern "C" void eps_warm_gemm_f32(void *stream) { hipLaunchKernelGGL(gemm_f32_warm_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream); }
