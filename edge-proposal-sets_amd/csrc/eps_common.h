// Shared host/device helpers for libeps_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/eps_abi.h"

#define EPS_WAVE 64

void eps_set_error(const char *fmt, ...);

#define EPS_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            eps_set_error(__VA_ARGS__);   \
            return EPS_EINVAL;            \
        }                                 \
    } while (0)

#define EPS_CHECK_LAUNCH(name)                                                   \
    do {                                                                         \
        hipError_t e__ = hipGetLastError();                                      \
        if (e__ != hipSuccess) {                                                 \
            eps_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return EPS_ELAUNCH;                                                  \
        }                                                                        \
    } while (0)

// Number of CUs of the current device (cached per process; 256 on MI355X).
int eps_num_cus();

// A zeroed device word for a kernel's dynamic work hand-out (see eps_common.hip).
int eps_take_counter(unsigned int **counter, hipStream_t stream, const char *who);
// ... eight consecutive zeroed words (a hand-out with one counter per XCD)
int eps_take_counters8(unsigned int **counters, hipStream_t stream, const char *who);
// ... eight zeroed words EPS_SPREAD_STRIDE words apart (each on a 256-byte line of its own: counter y at [y * EPS_SPREAD_STRIDE])
#define EPS_SPREAD_STRIDE 64
int eps_take_counters8_spread(unsigned int **counters, hipStream_t stream, const char *who);

#if defined(__HIPCC__)
// ---- wave-level reductions (all 64 lanes receive the total) --------------------------
// quad_perm / row_mirror DPP for the first four butterfly steps (VALU rate, no LDS traffic),
// ds_bpermute-backed shuffles for the two cross-row steps.
template <int CTRL>
__device__ __forceinline__ float eps_dpp_f(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int eps_dpp_i(int x)
{
    return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, true);
}

__device__ __forceinline__ float eps_wave_sum(float x)
{
    x += eps_dpp_f<0xB1>(x);   // quad_perm [1,0,3,2]
    x += eps_dpp_f<0x4E>(x);   // quad_perm [2,3,0,1]
    x += eps_dpp_f<0x141>(x);  // row_half_mirror
    x += eps_dpp_f<0x140>(x);  // row_mirror
    x += __shfl_xor(x, 16);
    x += __shfl_xor(x, 32);
    return x;
}

__device__ __forceinline__ double eps_wave_sum(double x)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) x += __shfl_xor(x, o);
    return x;
}

__device__ __forceinline__ int eps_lane() { return threadIdx.x & 63; }
#endif
