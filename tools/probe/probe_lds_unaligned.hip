// Checks that an unaligned four-byte LDS read (one ds_read_b32 at any byte offset) returns the right bytes on
// this GPU, for every offset modulo 4 and across bank and 128-byte boundaries.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_lds tools/probe/probe_lds_unaligned.hip && /tmp/probe_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned char *in, unsigned *out, unsigned base) {
    __shared__ __attribute__((aligned(16))) unsigned char s[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) s[i] = in[i];
    __syncthreads();
    unsigned w;
    __builtin_memcpy(&w, s + base + threadIdx.x * 7, 4);
    out[blockIdx.x * blockDim.x + threadIdx.x] = w;
}
int main() {
    std::vector<unsigned char> h(8192);
    for (int i = 0; i < 8192; i++) h[i] = (unsigned char)(i * 131 + (i >> 8) * 17 + 5);
    unsigned char *d_in;
    unsigned *d_out;
    hipMalloc(&d_in, 8192);
    hipMalloc(&d_out, 256 * 4);
    hipMemcpy(d_in, h.data(), 8192, hipMemcpyHostToDevice);
    long bad = 0, n = 0;
    for (unsigned base = 0; base < 64; base++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d_in, d_out, base);
        std::vector<unsigned> o(256);
        hipMemcpy(o.data(), d_out, 256 * 4, hipMemcpyDeviceToHost);
        for (int t = 0; t < 256; t++) {
            const unsigned a = base + t * 7;
            const unsigned want = h[a] | (h[a + 1] << 8) | (h[a + 2] << 16) | ((unsigned)h[a + 3] << 24);
            bad += o[t] != want;
            n++;
        }
    }
    printf("unaligned LDS dword reads: %ld checked, %ld wrong\n", n, bad);
    return bad != 0;
}
