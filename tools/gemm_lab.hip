// GEMM laboratory (not part of the library): variants of csrc/gemm_f32.hip's tiled kernel timed side by side on the layer
// shape of the ppa recipe, [576,289 x K] x [256 x K]^T + bias + ReLU.   hipcc -O3 --offload-arch=gfx950 tools/gemm_lab.hip -o /tmp/gemm_lab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define G_BN 128
#define G_BK 16
#define G_LD (G_BK + 4)
#define G_F4 (G_BK / 4)
#define G_NJ (G_BK / 8)

__device__ __forceinline__ v4f tile_load4(__amdgpu_buffer_rsrc_t rs, int row, int ld, int k, int kmax)
{
    const int off = (row * ld + k) * 4;
    v4f t = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs, k + 3 < kmax ? off : 0x7ffffff0, 0, 0));
    return t;
}

template <int BM>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const float *base, int64_t row0, int64_t nrows, int64_t ld)
{
    int64_t bytes = (nrows - row0) * ld * 4;
    const int64_t cap = (int64_t)BM * ld * 4 + 64;
    if (bytes > cap) bytes = cap;
    if (bytes < 0) bytes = 0;
    return __builtin_amdgcn_make_buffer_rsrc((void *)(base + row0 * ld), 0, (int)bytes, 0x00020000);
}

// BM = 128 (4 waves) or 256 (8 waves); PRIO: raise the wave priority around the MFMA groups; NOMEM: only the first chunk
// is loaded (structure ceiling without operand traffic); NOC: no C stores
template <int BM, int PRIO, int NOMEM, int NOC, int WGS>
__global__ __launch_bounds__(2 * BM, WGS) void gemm_lab_kernel(const float *__restrict__ A, int64_t lda, const float *__restrict__ B,
                                                            int64_t ldb, const float *__restrict__ bias, int relu,
                                                            float *__restrict__ C, int64_t ldc, int64_t M, int32_t N, int32_t K)
{
    constexpr int NT = 2 * BM;
    constexpr int NLA = BM * G_F4 / NT;      // float4 of A per thread (2)
    constexpr int NLB = G_BN * G_F4 / NT;    // float4 of B per thread (2 or 1)
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + G_BN) * G_LD];
    float(*As)[BM][G_LD] = reinterpret_cast<float(*)[BM][G_LD]>(smem);
    float(*Bs)[G_BN][G_LD] = reinterpret_cast<float(*)[G_BN][G_LD]>(smem + 2 * BM * G_LD);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wm = w >> 1, wn = w & 1;
    const int n_nblk = (N + G_BN - 1) / G_BN;
    const int64_t m0 = (int64_t)(blockIdx.x / n_nblk) * BM;
    const int n0 = (blockIdx.x % n_nblk) * G_BN;
    const __amdgpu_buffer_rsrc_t ra_rs = tile_rsrc<BM>(A, m0, M, lda);
    const __amdgpu_buffer_rsrc_t rb_rs = tile_rsrc<G_BN>(B, n0, N, ldb);
    const int ilda = (int)lda, ildb = (int)ldb;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = (K + G_BK - 1) / G_BK;
    v4f ra[NLA], rb[NLB];

#define G_GLOAD(kc)                                                                             \
    _Pragma("unroll") for (int i = 0; i < NLA; ++i)                                             \
    {                                                                                           \
        const int q = tid + NT * i;                                                             \
        ra[i] = tile_load4(ra_rs, q / G_F4, ilda, (kc)*G_BK + (q % G_F4) * 4, K);               \
    }                                                                                           \
    _Pragma("unroll") for (int i = 0; i < NLB; ++i)                                             \
    {                                                                                           \
        const int q = tid + NT * i;                                                             \
        rb[i] = tile_load4(rb_rs, q / G_F4, ildb, (kc)*G_BK + (q % G_F4) * 4, K);               \
    }
#define G_LSTORE(buf)                                                                           \
    _Pragma("unroll") for (int i = 0; i < NLA; ++i)                                             \
    {                                                                                           \
        const int q = tid + NT * i;                                                             \
        *reinterpret_cast<v4f *>(&As[buf][q / G_F4][(q % G_F4) * 4]) = ra[i];                   \
    }                                                                                           \
    _Pragma("unroll") for (int i = 0; i < NLB; ++i)                                             \
    {                                                                                           \
        const int q = tid + NT * i;                                                             \
        *reinterpret_cast<v4f *>(&Bs[buf][q / G_F4][(q % G_F4) * 4]) = rb[i];                   \
    }

    G_GLOAD(0);
    G_LSTORE(0);
    if (NOMEM) { G_LSTORE(1); }
    __syncthreads();

    for (int kc = 0; kc < nk; ++kc) {
        const int buf = kc & 1;
        if (!NOMEM && kc + 1 < nk) { G_GLOAD(kc + 1); }
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
            if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[s], bv0[s], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[s], bv1[s], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[s], bv0[s], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[s], bv1[s], acc[1][1], 0, 0, 0);
            }
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            if (j == G_NJ / 2) {
                __builtin_amdgcn_sched_barrier(0);
                if (!NOMEM && kc + 1 < nk) { G_LSTORE(buf ^ 1); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }
#undef G_GLOAD
#undef G_LSTORE

    float(*Cs)[G_BN + 4] = reinterpret_cast<float(*)[G_BN + 4]>(smem);
    const int c4 = (tid & 31) * 4;
    const int cc = n0 + c4;
    v4f bv = {0.f, 0.f, 0.f, 0.f};
    if (bias && cc < N) bv = *reinterpret_cast<const v4f *>(bias + cc);
    constexpr int RPP = NT / 32;          // rows stored per pass-iteration
#pragma unroll
    for (int pass = 0; pass < BM / 64; ++pass) {
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
        for (int i = 0; i < 64 / RPP; ++i) {
            const int rl = (tid >> 5) + RPP * i;
            const int64_t rr = m0 + pass * 64 + rl;
            if (rr < M && cc < N) {
                v4f t = *reinterpret_cast<const v4f *>(&Cs[rl][c4]) + bv;
                if (relu) {
                    t.x = t.x > 0.f ? t.x : 0.f;
                    t.y = t.y > 0.f ? t.y : 0.f;
                    t.z = t.z > 0.f ? t.z : 0.f;
                    t.w = t.w > 0.f ? t.w : 0.f;
                }
                if (NOC == 2) __builtin_nontemporal_store(t, reinterpret_cast<v4f *>(C + rr * ldc + cc));
                else if (!NOC || t.x == 12345.678f) *reinterpret_cast<v4f *>(C + rr * ldc + cc) = t;
            }
        }
        if (pass + 1 < BM / 64) __syncthreads();
    }
}

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            printf("%s: %s\n", #x, hipGetErrorString(e));                          \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

template <int BM, int PRIO, int NOMEM, int NOC, int WGS>
static float run(const char *name, const float *A, const float *B, const float *bias, float *C, int64_t M, int N, int K,
                 const float *Cref, std::vector<float> *keep)
{
    const int64_t mblk = (M + BM - 1) / BM, nblk = (N + G_BN - 1) / G_BN;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto launch = [&] {
        hipLaunchKernelGGL((gemm_lab_kernel<BM, PRIO, NOMEM, NOC, WGS>), dim3((unsigned)(mblk * nblk)), dim3(2 * BM), 0, 0, A, (int64_t)K, B,
                           (int64_t)K, bias, 1, C, (int64_t)N, M, N, K);
    };
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    float best = 1e9f, sum = 0.f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 10;
        sum += ms;
        if (ms < best) best = ms;
    }
    const double fl = 2.0 * M * N * K;
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, gemm_lab_kernel<BM, PRIO, NOMEM, NOC, WGS>, 2 * BM, 0));
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, (const void *)gemm_lab_kernel<BM, PRIO, NOMEM, NOC, WGS>));
    // spot check against the first variant's output
    double maxd = 0;
    if (!NOMEM && NOC != 1) {
        std::vector<float> hc(1 << 20);
        CK(hipMemcpy(hc.data(), C + (M / 2) * N, hc.size() * 4, hipMemcpyDeviceToHost));
        if (keep->empty()) *keep = hc;
        else
            for (size_t i = 0; i < hc.size(); ++i) {
                double d = fabs((double)hc[i] - (double)(*keep)[i]);
                if (d > maxd) maxd = d;
            }
    }
    printf("%-34s K=%d  mean %.3f ms  best %.3f ms  %.1f TF (best %.1f)  wg/CU %d  vgpr %d  lds %zu  maxdiff %.1e\n", name, K, sum / 5, best,
           fl / (sum / 5) / 1e9, fl / best / 1e9, occ, fa.numRegs, fa.sharedSizeBytes, maxd);
    return sum / 5;
}

int main(int argc, char **argv)
{
    const int64_t M = 576289;
    const int N = 256;
    for (int K : {316, 256}) {
        float *A, *B, *bias, *C;
        CK(hipMalloc(&A, M * K * 4));
        CK(hipMalloc(&B, (size_t)N * K * 4));
        CK(hipMalloc(&bias, N * 4));
        CK(hipMalloc(&C, M * N * 4));
        std::vector<float> h((size_t)M * K);
        uint32_t s = 12345;
        const bool lowent = argc > 1 && !strcmp(argv[1], "low");   // 16 random bits per value: reads high (less toggling, higher clocks)
        for (auto &x : h) {
            if (lowent) {
                s = s * 1664525u + 1013904223u;
                x = ((s >> 8) & 0xffff) / 32768.f - 1.f;
            } else {                                                // ~N(0,1) with full mantissas, like torch.randn
                float t = 0.f;
                for (int i = 0; i < 12; ++i) {
                    s = s * 1664525u + 1013904223u;
                    t += (s >> 8) * (1.f / 16777216.f);
                }
                x = t - 6.f;
            }
        }
        CK(hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(B, h.data() + 777, (size_t)N * K * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(bias, h.data() + 99, N * 4, hipMemcpyHostToDevice));
        std::vector<float> keep;
        run<128, 0, 0, 0, 3>("BM128 (shipped structure)", A, B, bias, C, M, N, K, nullptr, &keep);
        run<128, 1, 0, 0, 3>("BM128 + setprio", A, B, bias, C, M, N, K, nullptr, &keep);
        run<128, 1, 0, 2, 3>("BM128 + setprio + nt stores", A, B, bias, C, M, N, K, nullptr, &keep);
        run<128, 1, 0, 1, 3>("BM128 + setprio no C stores", A, B, bias, C, M, N, K, nullptr, &keep);
        run<128, 1, 1, 0, 3>("BM128 + setprio no loads", A, B, bias, C, M, N, K, nullptr, &keep);
        run<128, 1, 1, 1, 3>("BM128 + setprio no loads no stores", A, B, bias, C, M, N, K, nullptr, &keep);
        run<256, 1, 0, 2, 2>("BM256 + setprio + nt stores", A, B, bias, C, M, N, K, nullptr, &keep);
        run<128, 0, 1, 0, 3>("BM128 no operand loads", A, B, bias, C, M, N, K, nullptr, &keep);
        run<128, 0, 0, 1, 3>("BM128 no C stores", A, B, bias, C, M, N, K, nullptr, &keep);
        run<128, 0, 1, 1, 3>("BM128 no loads, no stores", A, B, bias, C, M, N, K, nullptr, &keep);
        run<256, 0, 0, 0, 2>("BM256 8 waves, 2 wg/CU", A, B, bias, C, M, N, K, nullptr, &keep);
        run<256, 1, 0, 0, 2>("BM256 8 waves + setprio", A, B, bias, C, M, N, K, nullptr, &keep);
        run<256, 0, 0, 0, 1>("BM256 8 waves, regs free", A, B, bias, C, M, N, K, nullptr, &keep);
        run<256, 0, 1, 1, 2>("BM256 no loads, no stores", A, B, bias, C, M, N, K, nullptr, &keep);
        CK(hipFree(A));
        CK(hipFree(B));
        CK(hipFree(bias));
        CK(hipFree(C));
    }
    return 0;
}
