// Adler-32 as a segmented reduction, stored (Raw) blocks, and the stream
// header / trailer.
#include "zg_internal.h"

namespace zg {

constexpr uint32_t kAdlerMod = 65521;  // ADLER_MODULO, reference src/adler.rs:1

// ---------------------------------------------------------------------------
// k_adler: one workgroup per segment.  For a segment x[0..N) taken alone
// (a starts at 1, b at 0; src/adler.rs:6-10):
//   a = 1 + sum x[i]                     (mod 65521)
//   b = N + sum (N - i) * x[i]           (mod 65521)
// computed from the two plain sums s1 = sum x[i], t = sum i*x[i] in 64 bits.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_adler(const uint8_t *__restrict__ data, uint64_t n, uint64_t seg_bytes,
                                                uint32_t *__restrict__ partial) {
    __shared__ uint64_t s_s1[4], s_t[4];
    const uint64_t s0 = (uint64_t)blockIdx.x * seg_bytes;
    const uint64_t len = (s0 + seg_bytes < n) ? seg_bytes : n - s0;
    const uint8_t *p = data + s0;
    const uint32_t tid = threadIdx.x;
    uint64_t s1 = 0, t = 0;
    if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
        const uint4 *v = reinterpret_cast<const uint4 *>(p);
        const uint64_t nv = len >> 4;
        for (uint64_t i = tid; i < nv; i += 256) {
            const uint4 q = v[i];
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
            uint32_t ls = 0, lt = 0;  // local sums over 16 bytes: lt = sum j*x[j], j = 0..15
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t b0 = w[k] & 0xFF, b1 = (w[k] >> 8) & 0xFF, b2 = (w[k] >> 16) & 0xFF, b3 = w[k] >> 24;
                ls += b0 + b1 + b2 + b3;
                lt += (4 * k) * b0 + (4 * k + 1) * b1 + (4 * k + 2) * b2 + (4 * k + 3) * b3;
            }
            s1 += ls;
            t += (i << 4) * (uint64_t)ls + lt;
        }
        for (uint64_t i = (nv << 4) + tid; i < len; i += 256) {
            s1 += p[i];
            t += i * (uint64_t)p[i];
        }
    } else {
        for (uint64_t i = tid; i < len; i += 256) {
            s1 += p[i];
            t += i * (uint64_t)p[i];
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_down(s1, off);
        t += __shfl_down(t, off);
    }
    if ((tid & 63) == 0) {
        s_s1[tid >> 6] = s1;
        s_t[tid >> 6] = t;
    }
    __syncthreads();
    if (tid == 0) {
        s1 = s_s1[0] + s_s1[1] + s_s1[2] + s_s1[3];
        t = s_t[0] + s_t[1] + s_t[2] + s_t[3];
        const uint64_t a = (1 + s1) % kAdlerMod;
        const uint64_t b = (len % kAdlerMod + ((len % kAdlerMod) * (s1 % kAdlerMod)) % kAdlerMod + kAdlerMod -
                            t % kAdlerMod) % kAdlerMod;
        partial[2 * blockIdx.x] = (uint32_t)a;
        partial[2 * blockIdx.x + 1] = (uint32_t)b;
    }
}

// Folds the segment values into the running (a, b) in order:
// (a1,b1) then a segment (a2,b2) of length L gives a = a1 + a2 - 1,
// b = b1 + b2 + L * (a1 - 1), all mod 65521.
__global__ void k_adler_fold(const uint32_t *__restrict__ partial, uint64_t n, uint64_t seg_bytes, uint32_t n_seg,
                             uint32_t *state) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint64_t a = state[0], b = state[1];
    for (uint32_t i = 0; i < n_seg; i++) {
        const uint64_t s0 = (uint64_t)i * seg_bytes;
        const uint64_t len = (s0 + seg_bytes < n) ? seg_bytes : n - s0;
        const uint64_t a2 = partial[2 * i], b2 = partial[2 * i + 1];
        b = (b + b2 + (len % kAdlerMod) * ((a + kAdlerMod - 1) % kAdlerMod)) % kAdlerMod;
        a = (a + a2 + kAdlerMod - 1) % kAdlerMod;
    }
    state[0] = (uint32_t)a;
    state[1] = (uint32_t)b;
}

hipError_t launch_adler(const uint8_t *data, uint64_t n, uint64_t seg_bytes, uint32_t n_seg, uint32_t *partial,
                        hipStream_t s) {
    if (!n_seg) return hipSuccess;
    hipLaunchKernelGGL(k_adler, dim3(n_seg), dim3(256), 0, s, data, n, seg_bytes, partial);
    return hipGetLastError();
}
hipError_t launch_adler_fold(const uint32_t *partial, uint64_t n, uint64_t seg_bytes, uint32_t n_seg,
                             uint32_t *state, hipStream_t s) {
    if (!n_seg) return hipSuccess;
    hipLaunchKernelGGL(k_adler_fold, dim3(1), dim3(64), 0, s, partial, n, seg_bytes, n_seg, state);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// k_stored: Strategy::Raw (reference src/compress.rs:172-195 with LEN = the
// block's own length): block b copies input [b*16384, ...) behind a five-byte
// header at output offset 2 + b*16389.  One workgroup per stored block; the
// destination is written in aligned dwords assembled from two aligned source
// dwords.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_stored(const uint8_t *__restrict__ data, uint64_t n, uint32_t n_blocks,
                                                 uint8_t *__restrict__ out) {
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    const uint64_t in0 = (uint64_t)b * kStoredBlock;
    const uint32_t len = (uint32_t)((in0 + kStoredBlock < n) ? kStoredBlock : n - in0);
    uint8_t *o = out + 2 + (uint64_t)b * (kStoredBlock + 5);
    if (tid == 0) {
        o[0] = (b + 1 == n_blocks) ? 1 : 0;  // BFINAL, BTYPE = 00, padding
        o[1] = (uint8_t)(len & 0xFF);
        o[2] = (uint8_t)(len >> 8);
        o[3] = (uint8_t)(~len & 0xFF);
        o[4] = (uint8_t)((~len >> 8) & 0xFF);
    }
    uint8_t *dst = o + 5;
    const uint8_t *src = data + in0;
    // leading bytes up to the first aligned destination dword
    const uint32_t head = (uint32_t)((4 - (reinterpret_cast<uintptr_t>(dst) & 3)) & 3);
    const uint32_t h = head < len ? head : len;
    if (tid < h) dst[tid] = src[tid];
    const uint32_t nw = (len - h) >> 2;  // whole destination dwords
    const uintptr_t sa = reinterpret_cast<uintptr_t>(src + h);
    const uint32_t sh = (uint32_t)(sa & 3);
    const uint32_t *sw = reinterpret_cast<const uint32_t *>(sa & ~(uintptr_t)3);
    uint32_t *dw = reinterpret_cast<uint32_t *>(dst + h);
    if (sh == 0) {
        for (uint32_t i = tid; i < nw; i += 256) dw[i] = sw[i];
    } else {
        // the last source dword touched is sw[nw], which holds bytes of this block (len - h - 4*nw < 4 tail
        // bytes or the start of the next block / padding) and is inside the allocation when n is padded
        for (uint32_t i = tid; i < nw; i += 256) dw[i] = __builtin_amdgcn_alignbyte(sw[i + 1], sw[i], sh);
    }
    const uint32_t done = h + (nw << 2);
    if (tid < len - done) dst[done + tid] = src[done + tid];
}

hipError_t launch_stored(const uint8_t *data, uint64_t n, uint8_t *out, hipStream_t s) {
    const uint64_t nb = n == 0 ? 1 : (n + kStoredBlock - 1) / kStoredBlock;
    hipLaunchKernelGGL(k_stored, dim3((uint32_t)nb), dim3(256), 0, s, data, n, (uint32_t)nb, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// k_trailer: zlib header bytes 78 01 (src/compress.rs:53-72) and the
// big-endian Adler-32 after the last block (src/compress.rs:82-83).
// ---------------------------------------------------------------------------
__global__ void k_trailer(uint8_t *out, const uint64_t *total_bits, const uint32_t *adler_state,
                          uint32_t adler_override, int use_override, int write_header, int write_adler,
                          uint64_t *out_bytes) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (write_header) {
        out[0] = 0x78;
        out[1] = 0x01;
    }
    uint64_t nb = (total_bits[0] + 7) >> 3;
    if (write_adler) {
        const uint32_t ad = use_override ? adler_override : ((adler_state[1] << 16) | adler_state[0]);
        out[nb + 0] = (uint8_t)(ad >> 24);
        out[nb + 1] = (uint8_t)(ad >> 16);
        out[nb + 2] = (uint8_t)(ad >> 8);
        out[nb + 3] = (uint8_t)ad;
        nb += 4;
    }
    out_bytes[0] = nb;
}

hipError_t launch_trailer(uint8_t *out, const uint64_t *total_bits, const uint32_t *adler_state,
                          uint32_t adler_override, int use_override, int write_header, int write_adler,
                          uint64_t *out_bytes, hipStream_t s) {
    hipLaunchKernelGGL(k_trailer, dim3(1), dim3(64), 0, s, out, total_bits, adler_state, adler_override,
                       use_override, write_header, write_adler, out_bytes);
    return hipGetLastError();
}

}  // namespace zg
