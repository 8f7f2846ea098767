// Probe: one LDS atomic max with return (ds_max_rtn_u32) by 64 lanes, even lanes storing a value larger than anything
// in the table (an insert), odd lanes "storing" 0 (a read that changes nothing), on colliding addresses: does every
// lane get the value its address held after exactly the lower lanes of the same instruction, i.e. are the lanes of
// one instruction served in ascending lane order, reads among writes included?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void probe(const uint32_t *keys, uint32_t *viol, int rounds) {
    __shared__ uint32_t tab[4096], shadow[4096];
    __shared__ uint32_t hs[64];
    for (int i = threadIdx.x; i < 4096; i += 64) tab[i] = shadow[i] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x;
    const bool ins = (lane & 1u) == 0;
    uint32_t bad = 0;
    for (int r = 0; r < rounds; r++) {
        const uint32_t h = keys[r * 64 + lane];
        const uint32_t v = 0x80000000u + (uint32_t)r * 64u + lane;  // grows with the round and with the lane
        const uint32_t got = __hip_atomic_fetch_max(&tab[h], ins ? v : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        hs[lane] = h;
        __syncthreads();
        uint32_t expect = shadow[h];
        for (int j = (int)lane - 1; j >= 0; j--)
            if ((j & 1) == 0 && hs[j] == h) {
                expect = 0x80000000u + (uint32_t)r * 64u + (uint32_t)j;
                break;
            }
        if (got != expect) bad++;
        __syncthreads();
        bool last = ins;
        for (int j = (int)lane + 1; j < 64 && last; j++)
            if ((j & 1) == 0 && hs[j] == h) last = false;
        if (last) shadow[h] = v;
        __syncthreads();
    }
    atomicAdd(viol, bad);
}
int main() {
    const int rounds = 20000;
    uint32_t *hk = (uint32_t *)malloc(rounds * 64 * 4);
    uint64_t s = 88172645463325252ULL;
    for (int i = 0; i < rounds * 64; i++) {
        s ^= s << 13;
        s ^= s >> 7;
        s ^= s << 17;
        const int kind = (i / 64) % 4;
        const int mod = kind == 0 ? 7 : (kind == 1 ? 64 : (kind == 2 ? 1024 : 4096));
        hk[i] = (uint32_t)(s >> 33) % mod;
    }
    uint32_t *dk, *dv;
    (void)hipMalloc(&dk, rounds * 64 * 4);
    (void)hipMalloc(&dv, 4);
    (void)hipMemset(dv, 0, 4);
    (void)hipMemcpy(dk, hk, rounds * 64 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1024), dim3(64), 0, 0, dk, dv, rounds);
    uint32_t v = 1;
    (void)hipMemcpy(&v, dv, 4, hipMemcpyDeviceToHost);
    printf("lane-order violations of ds_max_rtn (reads among inserts): %u over %d rounds x 1024 waves x 64 lanes\n", v, rounds);
    return v != 0;
}
