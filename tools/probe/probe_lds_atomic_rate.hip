// Probe: cycles per 64-lane LDS instruction on a 8192-entry table, by kind of instruction and address pattern, with 1, 2
// or 4 waves per CU issuing them (16 in flight per wave between waits).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
enum { K_MAX_RTN, K_MAX_NORTN, K_XCHG, K_READ, K_MAX_RTN_HALF, K_ADD_RTN };
template <int KIND, int PATTERN>  // PATTERN 0: random, 1: lane-linear (no bank conflict), 2: random within 32 distinct banks per half
__global__ __launch_bounds__(64) void probe(uint32_t *out, unsigned long long *cyc, int iters, uint32_t seed) {
    __shared__ uint32_t tab[8192];
    __shared__ uint32_t pad[2048];  // 40 KiB per workgroup: at most four per CU; the grid decides how many run
    const uint32_t lane = threadIdx.x;
    for (int i = lane; i < 8192; i += 64) tab[i] = 0;
    if (seed == 0xFFFFFFFFu) pad[lane] = 1;
    __syncthreads();
    uint32_t s = seed * 2654435761u + lane * 40503u + blockIdx.x * 977u + 1u;
    uint32_t h0[16];
    for (int k = 0; k < 16; k++) {  // sixteen addresses per lane, shifted by a multiple of 64 entries every round
        s = s * 1664525u + 1013904223u;
        if (PATTERN == 0) h0[k] = s >> 19;
        else if (PATTERN == 1) h0[k] = ((s >> 19) & ~63u) | lane;
        else h0[k] = ((s >> 19) & ~31u) | (lane & 31u);
    }
    uint32_t acc = 0;
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        uint32_t r[16];
        const uint32_t shift = (uint32_t)it * 192u, v = 0x40000000u + (uint32_t)it;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint32_t h = (h0[k] + shift) & 8191u;
            if (KIND == K_MAX_RTN) r[k] = __hip_atomic_fetch_max(&tab[h], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (KIND == K_MAX_NORTN) { (void)__hip_atomic_fetch_max(&tab[h], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); r[k] = 0; }
            else if (KIND == K_XCHG) r[k] = __hip_atomic_exchange(&tab[h], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (KIND == K_READ) r[k] = tab[h];
            else if (KIND == K_ADD_RTN) r[k] = __hip_atomic_fetch_add(&tab[h], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else { r[k] = 0; if (lane & 1) r[k] = __hip_atomic_fetch_max(&tab[h], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        }
#pragma unroll
        for (int k = 0; k < 16; k++) acc += r[k];
        if (KIND == K_READ) tab[(acc & 63u) * 128u + lane] = acc;  // keeps the reads of a round from being hoisted
    }
    const long long t1 = clock64();
    if (lane == 0) atomicAdd(cyc, (unsigned long long)(t1 - t0));
    out[blockIdx.x * 64 + lane] = acc;
}
template <int KIND, int PATTERN>
static void run(const char *name, uint32_t *out, unsigned long long *cyc, int cus) {
    const int iters = 2000;
    for (int per_cu = 1; per_cu <= 4; per_cu *= 2) {  // 40 KiB of LDS per workgroup: up to 4 per CU
        (void)hipMemset(cyc, 0, 8);
        hipLaunchKernelGGL((probe<KIND, PATTERN>), dim3(cus * per_cu), dim3(64), 0, 0, out, cyc, iters, 12345u);
        unsigned long long c = 0;
        (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double per_wave = (double)c / (cus * per_cu) / (iters * 16.0);
        printf("%-28s %d wave(s)/CU: %6.1f cycles per instruction and wave, %6.1f per instruction and CU\n", name, per_cu,
               per_wave, per_wave / per_cu);
    }
}
int main() {
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    uint32_t *out;
    unsigned long long *cyc;
    (void)hipMalloc(&out, (size_t)cus * 4 * 64 * 4);
    (void)hipMalloc(&cyc, 8);
    run<K_READ, 0>("ds_read_b32 random", out, cyc, cus);
    run<K_READ, 1>("ds_read_b32 linear", out, cyc, cus);
    run<K_MAX_RTN, 0>("ds_max_rtn random", out, cyc, cus);
    run<K_MAX_RTN, 1>("ds_max_rtn linear", out, cyc, cus);
    run<K_MAX_RTN, 2>("ds_max_rtn 32 banks apart", out, cyc, cus);
    run<K_MAX_NORTN, 0>("ds_max (no return) random", out, cyc, cus);
    run<K_XCHG, 0>("ds_wrxchg_rtn random", out, cyc, cus);
    run<K_ADD_RTN, 0>("ds_add_rtn random", out, cyc, cus);
    run<K_MAX_RTN_HALF, 0>("ds_max_rtn random, odd lanes", out, cyc, cus);
    return 0;
}
