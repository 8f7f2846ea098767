// Probe: does an LDS atomic exchange by 64 lanes on colliding addresses return values in ascending lane order?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(const uint32_t* keys, uint32_t* viol, int rounds) {
    __shared__ uint32_t tab[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) tab[i] = 0;
    __syncthreads();
    __shared__ uint32_t hs[64];
    uint32_t bad = 0, bad2 = 0;
    for (int r = 0; r < rounds; r++) {
        uint32_t h = keys[r * 64 + threadIdx.x];
        uint32_t q = r * 64 + threadIdx.x + 1;
        uint32_t old = atomicExch(&tab[h], q);
        if (old >= q) bad++;                 // a later lane's value seen by an earlier lane
        // also check strict chain: old must be the nearest lower lane with same h in this round, or older round
        hs[threadIdx.x] = h;
        __syncthreads();
        uint32_t expect_inbatch = 0;
        for (int j = 0; j < (int)threadIdx.x; j++) { if (hs[j] == h) expect_inbatch = r * 64 + j + 1; }
        __syncthreads();
        if (expect_inbatch && old != expect_inbatch) bad2++;
        if (!expect_inbatch && old > (uint32_t)(r * 64)) bad2++;
    }
    atomicAdd(viol, bad);
    atomicAdd(viol + 1, bad2);
}
int main() {
    const int rounds = 20000;
    uint32_t* hk = (uint32_t*)malloc(rounds * 64 * 4);
    uint64_t s = 88172645463325252ULL;
    for (int i = 0; i < rounds * 64; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; int mod = (i / 64) % 4 == 0 ? 7 : ((i/64)%4==1 ? 64 : ((i/64)%4==2 ? 1024: 4096)); hk[i] = (uint32_t)(s >> 33) % mod; }
    uint32_t *dk, *dv; hipMalloc(&dk, rounds * 64 * 4); hipMalloc(&dv, 8); hipMemset(dv, 0, 8);
    hipMemcpy(dk, hk, rounds * 64 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(512), dim3(64), 0, 0, dk, dv, rounds);
    uint32_t v[2] = {1, 1}; hipMemcpy(v, dv, 8, hipMemcpyDeviceToHost);
    printf("order violations: %u, chain violations: %u (over %d rounds x 512 blocks)\n", v[0], v[1], rounds);
    return 0;
}
