// Probe: are unaligned ds_read_b32 / ds_read_b64 correct on gfx950, and what do they cost next to
// ds_read2_b32 + v_alignbyte_b32 (random addresses, 1024 threads per block, one block per CU)?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s\n", hipGetErrorString(e)); return 1; } } while (0)
constexpr int N = 48 * 1024;
__device__ __forceinline__ uint32_t ld4_aligned(const uint8_t* lds, uint32_t off) {
    const uint32_t* w = reinterpret_cast<const uint32_t*>(lds + (off & ~3u));
    return __builtin_amdgcn_alignbyte(w[1], w[0], off & 3u);
}
__device__ __forceinline__ uint32_t ld4_unaligned(const uint8_t* lds, uint32_t off) {
    uint32_t v; __builtin_memcpy(&v, lds + off, 4); return v;
}
__device__ __forceinline__ uint64_t ld8_unaligned(const uint8_t* lds, uint32_t off) {
    uint64_t v; __builtin_memcpy(&v, lds + off, 8); return v;
}
template <int MODE> __global__ __launch_bounds__(1024) void k(uint32_t* out, uint32_t* bad, int iters) {
    __shared__ __attribute__((aligned(16))) uint8_t s[N + 16];
    for (int i = threadIdx.x; i < N + 16; i += 1024) s[i] = (uint8_t)((i * 131u + (i >> 8) * 7u) & 0xFF);
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u, acc = 0, nbad = 0;
    for (int it = 0; it < iters; it++) {
        x = x * 1664525u + 1013904223u;
        const uint32_t off = (x >> 8) % (N - 8);
        uint32_t v;
        if (MODE == 0) v = ld4_aligned(s, off);
        else if (MODE == 1) v = ld4_unaligned(s, off);
        else { uint64_t w = ld8_unaligned(s, off); v = (uint32_t)w ^ (uint32_t)(w >> 32); }
        if (MODE <= 1 && it < 64) {
            uint32_t e = 0;
            for (int b = 0; b < 4; b++) e |= (uint32_t)s[off + b] << (8 * b);
            if (e != v) nbad++;
        }
        if (MODE == 2 && it < 64) {
            uint32_t e = 0;
            for (int b = 0; b < 4; b++) e |= ((uint32_t)s[off + b] ^ (uint32_t)s[off + 4 + b]) << (8 * b);
            if (e != v) nbad++;
        }
        acc ^= v;
        x ^= v & 1;  // dependent chain like a pointer chase
    }
    out[blockIdx.x * 1024 + threadIdx.x] = acc;
    if (nbad) atomicAdd(bad, nbad);
}
template <int MODE> int run(const char* name) {
    uint32_t *out, *bad; CHECK(hipMalloc(&out, 256 * 1024 * 4)); CHECK(hipMalloc(&bad, 4)); CHECK(hipMemset(bad, 0, 4));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const int iters = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, out, bad, 100);
    CHECK(hipDeviceSynchronize()); CHECK(hipMemset(bad, 0, 4));
    CHECK(hipEventRecord(a)); hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, out, bad, iters); CHECK(hipEventRecord(b));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    uint32_t hb; CHECK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    printf("%-28s wrong=%u  %.3f ms  %.2f Gread/s\n", name, hb, ms, 256.0 * 1024 * iters / ms / 1e6);
    return 0;
}
int main() { return run<0>("read2_b32+alignbyte") | run<1>("unaligned ds_read_b32") | run<2>("unaligned ds_read_b64"); }
