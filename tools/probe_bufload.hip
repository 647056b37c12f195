// Hardware probe: raw buffer loads on gfx950 -- 4-byte-aligned b128 loads and per-dword range checking.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k(const int* p, int off, int n, int* out) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(p + off), 0, n * 4, 0x00020000);
    v4i v = __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x * 16, 0, 0);
    out[threadIdx.x * 4 + 0] = v.x; out[threadIdx.x * 4 + 1] = v.y; out[threadIdx.x * 4 + 2] = v.z; out[threadIdx.x * 4 + 3] = v.w;
}
int main() {
    int *d, *o; int h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 1000 + i;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, 64 * 4 * 4); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    int ho[256];
    for (int off = 0; off < 4; ++off) for (int n : {0, 1, 5, 6, 7, 8, 130}) {
        k<<<1, 64>>>(d, off, n, o); hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 256; ++i) { int want = i < n ? 1000 + off + i : 0; if (ho[i] != want) { if (bad < 3) printf("  off=%d n=%d i=%d got %d want %d\n", off, n, i, ho[i], want); ++bad; } }
        printf("off=%d n=%d mismatches=%d\n", off, n, bad);
    }
    return 0;
}
