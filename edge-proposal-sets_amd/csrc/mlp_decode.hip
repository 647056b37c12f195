// Fused LinkPredictor decode on the f32-input MFMA, gfx950.
//
// Replaces h[edges[0]] / h[edges[1]] gathers (models.py:506) + LinkPredictor.forward
// (models.py:478-485): Hadamard -> (L-1) x [Linear, ReLU] -> Linear(H,1) -> sigmoid, one float
// per candidate edge out.  Dropout is the identity in eval mode (models.py:483 with
// training=False), which is the only mode the scoring path runs in.
//
// One 512-thread workgroup (8 waves, two per SIMD) per CU walks 64-edge tiles (persistent):
//   1. gather: each wave builds 8 rows of X = h[u] (.) h[v] straight into LDS (one coalesced
//      1-KiB row read per endpoint, float4 per lane, all 16 reads of a wave in flight at once);
//   2. hidden layers: X[64,H] stays in LDS; W_l streams through LDS in K-chunks of 32
//      (register-staged double buffer, one barrier per chunk); each wave owns 32 rows x 2 MFMA
//      32x32 column tiles (32 accumulator registers); bias + ReLU are applied in the
//      accumulators and written back over X -- activations never leave the CU;
//   3. last layer (H -> 1) is an 8-lanes-per-row dot product over the LDS tile + sigmoid.
// f32 in, f32 accumulate (v_mfma_f32_32x32x2_f32 == fmaf chain): the 1e-5 parity gate rules out
// bf16/"xf32" shortcuts (and gfx950 has no xf32).
#include "eps_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));  // native vector: stays in VGPRs (HIP's float4 struct copies lower to memcpy -> scratch)

#define D_BM 64        // edges per tile
#define D_HMAX 256     // widest hidden size held in LDS
#define D_XLD (D_HMAX + 4)
#define D_BK 32
#define D_WLD 36
#define D_MAXL 8

struct DecodeParams {
    const float *w[D_MAXL];
    const float *b[D_MAXL];
};

// Static-index select: a runtime index into the by-value kernel argument would force the
// struct into scratch memory.
__device__ __forceinline__ const float *pick(const float *const (&a)[D_MAXL], int l)
{
    const float *p = a[0];
#pragma unroll
    for (int i = 1; i < D_MAXL; ++i) p = (i == l) ? a[i] : p;
    return p;
}

#define D_THREADS 512  // 8 waves = two per SIMD: one wave's LDS/barrier stalls are covered by its SIMD partner's MFMAs
#define D_WREGS (D_HMAX * (D_BK / 4) / D_THREADS)  // float4 of a W chunk staged per thread (4)

// W chunk kc = rows [0,Hp) x cols [kc*32, kc*32+32) of the row-major [H,H] weight, staged global -> registers ->
// LDS.  Raw buffer loads over the H*H matrix: rows >= H fall outside the descriptor and read as zeros with no branch;
// columns >= H (only when H is not a multiple of 32) are pushed out of range by a select on the offset.  Keeping
// the staging free of exec-mask branches is what lets hipcc issue the loads back to back instead of load-wait pairs.
__device__ __forceinline__ void w_gload(v4f (&rw)[D_WREGS], __amdgpu_buffer_rsrc_t wr, int H, int tid, int kc)
{
#pragma unroll
    for (int i = 0; i < D_WREGS; ++i) {
        const int q = tid + D_THREADS * i;
        const int row = q >> 3, c4 = q & 7;
        const int kcol = kc * D_BK + c4 * 4;     // H % 4 == 0: a float4 is entirely inside or outside [0,H)
        const int off = (row * H + kcol) * 4;
        rw[i] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(wr, kcol < H ? off : 0x7ffffff0, 0, 0));
    }
}

__device__ __forceinline__ void w_lstore(const v4f (&rw)[D_WREGS], float (*Wb)[D_WLD], int tid)
{
#pragma unroll
    for (int i = 0; i < D_WREGS; ++i) {
        const int q = tid + D_THREADS * i;
        const int row = q >> 3, c4 = q & 7;
        *reinterpret_cast<v4f *>(&Wb[row][c4 * 4]) = rw[i];  // all D_HMAX rows: rows >= H carry zeros
    }
}

__global__ __launch_bounds__(D_THREADS) void mlp_decode_kernel(const float *__restrict__ hmat, int32_t H,
                                                               const int32_t *__restrict__ pu,
                                                               const int32_t *__restrict__ pv, int64_t n_pairs,
                                                               DecodeParams prm, int32_t n_layers, int apply_sigmoid,
                                                               float *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) float Xs[D_BM][D_XLD];
    __shared__ __attribute__((aligned(16))) float Ws[2][D_HMAX][D_WLD];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform -> scalar branches, no exec masking
    const int r = lane & 31, hh = lane >> 5;
    const int wm = w >> 2, wn = w & 3;      // wave -> rows [32*wm, +32), column tiles wn and wn+4
    const int Hp = (H + 31) & ~31;          // H padded to the MFMA tile width; pad columns are kept at zero
    const int n_ntiles = Hp >> 5;           // 32-column output tiles (<= 8)
    const int t0 = wn, t1 = wn + 4;
    const bool has0 = t0 < n_ntiles, has1 = t1 < n_ntiles;
    const int nk = Hp / D_BK;
    const int64_t n_tiles = (n_pairs + D_BM - 1) / D_BM;
    const int h4 = H >> 2;                  // float4 per row
    const int hp4 = Hp >> 2;

    // The rows of h a tile gathers (2 x 64 KiB-rows) are PREFETCHED into registers during the previous tile's MFMA
    // phase: 16 float4 per lane (8 rows x 2 endpoints; a row is one float4 per lane at H <= 256).  Every prefetch
    // register is written by one unconditional load per tile (tiles past the end re-read row 0: no phi copies, so
    // hipcc keeps counted vmcnt waits instead of draining the queue).
    const int cl = lane < h4 ? lane : 0;  // this lane's float4 column of a row (clamped for H < 256)
    auto tile_ids = [&](int64_t t, int32_t &mu, int32_t &mv) {
        const int64_t p = t * D_BM + lane;
        const bool ok = t < n_tiles && p < n_pairs;
        const int64_t pc = ok ? p : 0;
        mu = pu[pc];
        mv = pv[pc];
        if (!ok) { mu = 0; mv = 0; }
    };
    v4f ga[8], gb[8];
    int32_t mu, mv;
    tile_ids(blockIdx.x, mu, mv);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int64_t un = __builtin_amdgcn_readlane(mu, w * 8 + i), vn = __builtin_amdgcn_readlane(mv, w * 8 + i);
        ga[i] = *reinterpret_cast<const v4f *>(hmat + un * H + 4 * cl);
        gb[i] = *reinterpret_cast<const v4f *>(hmat + vn * H + 4 * cl);
    }

    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t e0 = tile * D_BM;
        int32_t nmu, nmv;                       // ids of the NEXT tile of this workgroup: in flight during the X build
        tile_ids(tile + gridDim.x, nmu, nmv);
        // ---- 1. Hadamard of the prefetched rows into LDS: wave w owns rows 8w..8w+7 ---------------------------
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v4f pr = ga[i] * gb[i];
            if (lane >= h4) pr = (v4f){0.f, 0.f, 0.f, 0.f};  // pad columns (and idle lanes when H < 256)
            if (lane < hp4) *reinterpret_cast<v4f *>(&Xs[w * 8 + i][4 * lane]) = pr;
        }
        __syncthreads();
        // first W chunk of the first hidden layer, THEN the next tile's row prefetch: the chunk's wait (counted vmcnt)
        // leaves the younger prefetch loads in flight under the MFMA phase
        v4f rw[D_WREGS];
        {
            const float *W0 = pick(prm.w, 0);
            const __amdgpu_buffer_rsrc_t wr0 = __builtin_amdgcn_make_buffer_rsrc((void *)W0, 0, n_layers > 1 ? H * H * 4 : 0, 0x00020000);
            w_gload(rw, wr0, H, tid, 0);
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the W loads OLDER than the prefetch (vmcnt retires in issue order)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t un = __builtin_amdgcn_readlane(nmu, w * 8 + i), vn = __builtin_amdgcn_readlane(nmv, w * 8 + i);
            ga[i] = *reinterpret_cast<const v4f *>(hmat + un * H + 4 * cl);
            gb[i] = *reinterpret_cast<const v4f *>(hmat + vn * H + 4 * cl);
        }
        // stage chunk 0 here, in straight-line code: inside the layer loop the wait would be merged with the back edge
        // (where the W loads are the youngest) and degrade to vmcnt(0), draining the prefetch
        w_lstore(rw, Ws[0], tid);
        __syncthreads();

        // ---- 2. hidden layers -------------------------------------------------------------
        for (int l = 0; l + 1 < n_layers; ++l) {
            const float *__restrict__ W = pick(prm.w, l);
            const float *__restrict__ Bv = pick(prm.b, l);
            f32x16 acc[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

            const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void *)W, 0, H * H * 4, 0x00020000);
            // (chunk 0 of this layer is already in Ws[0]: staged before the loop / at the end of the previous layer)
            for (int kc = 0; kc < nk; ++kc) {
                const int buf = kc & 1;
                if (kc + 1 < nk) w_gload(rw, wr, H, tid, kc + 1);
                // all fragment reads of the chunk are issued up front (A: 4, B: 4 per column tile); the MFMAs then
                // drain them in order behind counted lgkmcnt waits
                float4 af[4], bf0[4], bf1[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) af[j] = *reinterpret_cast<const float4 *>(&Xs[wm * 32 + r][kc * D_BK + 8 * j + 4 * hh]);
                if (has0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) bf0[j] = *reinterpret_cast<const float4 *>(&Ws[buf][t0 * 32 + r][8 * j + 4 * hh]);
                }
                if (has1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) bf1[j] = *reinterpret_cast<const float4 *>(&Ws[buf][t1 * 32 + r][8 * j + 4 * hh]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float av[4] = {af[j].x, af[j].y, af[j].z, af[j].w};
                    if (has0) {
                        const float bv[4] = {bf0[j].x, bf0[j].y, bf0[j].z, bf0[j].w};
#pragma unroll
                        for (int ss = 0; ss < 4; ++ss) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ss], bv[ss], acc[0], 0, 0, 0);
                    }
                    if (has1) {
                        const float bv[4] = {bf1[j].x, bf1[j].y, bf1[j].z, bf1[j].w};
#pragma unroll
                        for (int ss = 0; ss < 4; ++ss) acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ss], bv[ss], acc[1], 0, 0, 0);
                    }
                }
                if (kc + 1 < nk) w_lstore(rw, Ws[buf ^ 1], tid);
                __syncthreads();
            }
            // every wave has finished reading X (barrier above): overwrite it with relu(acc + b)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const bool has = ni == 0 ? has0 : has1;
                if (!has) continue;
                const int cc = (ni == 0 ? t0 : t1) * 32 + r;
                const float bv = cc < H ? Bv[cc] : 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int rr = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                    const float t = acc[ni][e] + bv;
                    Xs[rr][cc] = t > 0.f ? t : 0.f;
                }
            }
            {   // request chunk 0 of the next hidden layer (zero-length descriptor after the last one: no traffic)
                const bool more = l + 2 < n_layers;
                const float *Wn = pick(prm.w, more ? l + 1 : 0);
                const __amdgpu_buffer_rsrc_t wrn = __builtin_amdgcn_make_buffer_rsrc((void *)Wn, 0, more ? H * H * 4 : 0, 0x00020000);
                w_gload(rw, wrn, H, tid, 0);
            }
            __syncthreads();
            w_lstore(rw, Ws[0], tid);   // zeros after the last hidden layer (zero-length descriptor)
            __syncthreads();
        }

        // ---- 3. last layer: H -> 1, sigmoid (8 lanes per row) ---------------------------------------
        {
            const float *__restrict__ wl = pick(prm.w, n_layers - 1);
            const int row = tid >> 3, part = tid & 7;
            float s = 0.f;
            for (int c = part; c < h4; c += 8) {
                const float4 x = *reinterpret_cast<const float4 *>(&Xs[row][4 * c]);
                const float4 q = *reinterpret_cast<const float4 *>(wl + 4 * c);
                s = fmaf(x.x, q.x, s);
                s = fmaf(x.y, q.y, s);
                s = fmaf(x.z, q.z, s);
                s = fmaf(x.w, q.w, s);
            }
            s += eps_dpp_f<0xB1>(s);   // quad_perm [1,0,3,2]
            s += eps_dpp_f<0x4E>(s);   // quad_perm [2,3,0,1]
            s += eps_dpp_f<0x141>(s);  // row_half_mirror: the other quad of the 8-lane group
            const int64_t p = e0 + row;
            if (part == 0 && p < n_pairs) {
                float z = s + pick(prm.b, n_layers - 1)[0];
                if (apply_sigmoid) z = 1.0f / (1.0f + expf(-z));
                out[p] = z;
            }
        }
        __syncthreads();  // X is rebuilt by the next tile's gather
    }
}

extern "C" int eps_mlp_decode(const float *h, int64_t n_nodes, int32_t hdim, const int32_t *u, const int32_t *v,
                              int64_t n_pairs, const float *const *w, const float *const *b, int32_t n_layers,
                              int apply_sigmoid, float *out, void *stream)
{
    EPS_REQUIRE(n_pairs >= 0 && n_nodes >= 0, "eps_mlp_decode: negative size");
    EPS_REQUIRE(hdim > 0 && hdim % 4 == 0 && hdim <= D_HMAX, "eps_mlp_decode: hdim=%d unsupported (need %%4==0, <=%d)",
                hdim, D_HMAX);
    EPS_REQUIRE(n_layers >= 1 && n_layers <= D_MAXL, "eps_mlp_decode: n_layers=%d unsupported (1..%d)", n_layers, D_MAXL);
    if (n_pairs == 0) return EPS_OK;
    EPS_REQUIRE(h && u && v && w && b && out, "eps_mlp_decode: null pointer");
    EPS_REQUIRE((uintptr_t)h % 16 == 0, "eps_mlp_decode: h must be 16-byte aligned");
    DecodeParams prm;
    for (int l = 0; l < D_MAXL; ++l) {
        prm.w[l] = l < n_layers ? w[l] : nullptr;
        prm.b[l] = l < n_layers ? b[l] : nullptr;
        if (l < n_layers) {
            EPS_REQUIRE(w[l] && b[l], "eps_mlp_decode: null weight/bias pointer at layer %d", l);
            EPS_REQUIRE((uintptr_t)w[l] % 16 == 0, "eps_mlp_decode: weight %d must be 16-byte aligned", l);
        }
    }
    const int64_t n_tiles = (n_pairs + D_BM - 1) / D_BM;
    int64_t blocks = eps_num_cus();
    if (blocks > n_tiles) blocks = n_tiles;
    hipLaunchKernelGGL(mlp_decode_kernel, dim3((unsigned)blocks), dim3(D_THREADS), 0, (hipStream_t)stream, h, hdim, u, v,
                       n_pairs, prm, n_layers, apply_sigmoid, out);
    EPS_CHECK_LAUNCH("eps_mlp_decode");
    return EPS_OK;
}
