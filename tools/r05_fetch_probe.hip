// Calibration of FETCH_SIZE / TCC_EA0_RDREQ for the ACCESS SHAPE of the scan kernel's walk ("settle the FETCH_SIZE x2
// question for this kernel"): a wave's 64 lanes take consecutive 16-byte units of the virtual concatenation of short row
// segments (~236 B each, 4-byte aligned, at random places of a 2 GiB buffer: far beyond the Infinity Cache), one
// raw_buffer_load_b128 per lane.  The host knows exactly which bytes, 64-B lines and 128-B lines were touched; run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./r05_fetch_probe      and      --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
// and compare.  A second kernel streams the same number of bytes fully coalesced (the guide's calibrated case: x2).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef int v4i __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void walk_like(const int *buf, uint32_t buf_bytes, const uint32_t *unit_at, int64_t n_units, int *sink)
{
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)buf, 0, buf_bytes, 0x00020000);
    int acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_units; i += (int64_t)gridDim.x * blockDim.x) {
        const v4i v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)unit_at[i], 0, 0);
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678) sink[0] = acc;
}

__global__ __launch_bounds__(256) void stream_like(const int *buf, int64_t n_units, int *sink)
{
    int acc = 0;
    const v4i *p = (const v4i *)buf;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_units; i += (int64_t)gridDim.x * blockDim.x) {
        const v4i v = p[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678) sink[0] = acc;
}

int main()
{
    const uint64_t buf_bytes = 2040ull << 20;                  // (< 2^31: 32-bit buffer offsets)
    const int64_t n_units = 1ll << 25;                         // 512 MiB of 16-byte loads
    int *buf, *sink;
    uint32_t *unit_at;
    hipMalloc(&buf, buf_bytes);
    hipMalloc(&sink, 4);
    hipMalloc(&unit_at, n_units * 4);
    hipMemset(buf, 1, buf_bytes);
    std::vector<uint32_t> at(n_units);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    std::vector<uint64_t> l64, l128;
    l64.reserve(n_units * 2); l128.reserve(n_units * 2);
    int64_t u = 0, n_seg = 0, bytes = 0;
    while (u < n_units) {
        const uint32_t len = 40 + (uint32_t)(rnd() % 40);      // entries of the segment: 40 .. 79 (160 .. 316 B; mean 59 entries)
        const uint32_t start = (uint32_t)(rnd() % ((buf_bytes - 4096) / 4)) * 4u;
        const uint32_t units = (len + 3) / 4;
        for (uint32_t k = 0; k < units && u < n_units; ++k, ++u) {
            at[u] = start + 16u * k;
            // (the last unit of a segment reads 16 bytes too: the kernel's loads are whole units, the tail lanes masked later)
            for (uint32_t b = at[u]; b < at[u] + 16; b += 4) { l64.push_back(b >> 6); l128.push_back(b >> 7); }
            bytes += 16;
        }
        ++n_seg;
    }
    std::sort(l64.begin(), l64.end()); std::sort(l128.begin(), l128.end());
    const int64_t n64 = std::unique(l64.begin(), l64.end()) - l64.begin(), n128 = std::unique(l128.begin(), l128.end()) - l128.begin();
    hipMemcpy(unit_at, at.data(), n_units * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(walk_like, dim3(4096), dim3(256), 0, 0, buf, (uint32_t)buf_bytes, unit_at, n_units, sink);
    hipLaunchKernelGGL(stream_like, dim3(4096), dim3(256), 0, 0, buf, n_units, sink);
    hipDeviceSynchronize();
    printf("{\"units\": %lld, \"segments\": %lld, \"bytes_loaded\": %lld, \"distinct_64B_lines\": %lld, \"bytes_of_64B_lines\": %lld, "
           "\"distinct_128B_lines\": %lld, \"bytes_of_128B_lines\": %lld, \"stream_bytes\": %lld, \"unit_table_bytes\": %lld}\n",
           (long long)n_units, (long long)n_seg, (long long)bytes, (long long)n64, (long long)n64 * 64, (long long)n128, (long long)n128 * 128,
           (long long)n_units * 16, (long long)n_units * 4);
    return 0;
}
