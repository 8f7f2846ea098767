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
                                                const uint32_t *__restrict__ seg_len,
                                                uint32_t *__restrict__ partial) {
    __shared__ uint64_t s_s1[4], s_t[4];
    const uint64_t s0 = (uint64_t)blockIdx.x * seg_bytes;
    const uint64_t len = seg_len ? (uint64_t)seg_len[blockIdx.x] : ((s0 + seg_bytes < n) ? seg_bytes : n - s0);
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

// Folds the segment values into the running (a, b) in order: (a1,b1) followed by a segment (a2,b2) of length L
// gives a = a1 + a2 - 1, b = b1 + b2 + L * (a1 - 1), all mod 65521.  The rule is associative, so the fold is a
// two-stage ordered tree.
struct AdlerPart {
    uint64_t a, b, len;  // a, b as for a stream that starts at (1, 0); len mod 65521
};
__device__ __forceinline__ AdlerPart adler_join(AdlerPart x, AdlerPart y) {
    AdlerPart r;
    r.b = (x.b + y.b + (y.len % kAdlerMod) * ((x.a + kAdlerMod - 1) % kAdlerMod)) % kAdlerMod;
    r.a = (x.a + y.a + kAdlerMod - 1) % kAdlerMod;
    r.len = (x.len + y.len) % kAdlerMod;
    return r;
}
// stage 1: every block joins 1024 consecutive segments (4 per thread, then an ordered tree over the block)
__global__ __launch_bounds__(256) void k_adler_fold1(const uint32_t *__restrict__ partial, uint64_t n,
                                                      uint64_t seg_bytes, uint32_t n_seg, uint64_t *block_part) {
    __shared__ AdlerPart s_part[256];
    const uint32_t tid = threadIdx.x;
    const uint32_t lo = (blockIdx.x * 256 + tid) * 4;
    AdlerPart acc{1, 0, 0};
    for (uint32_t i = lo; i < lo + 4 && i < n_seg; i++) {
        const uint64_t s0 = (uint64_t)i * seg_bytes;
        const uint64_t len = (s0 + seg_bytes < n) ? seg_bytes : n - s0;
        acc = adler_join(acc, AdlerPart{partial[2 * i], partial[2 * i + 1], len});
    }
    s_part[tid] = acc;
    __syncthreads();
    for (uint32_t stride = 1; stride < 256; stride <<= 1) {
        if ((tid & (2 * stride - 1)) == 0) s_part[tid] = adler_join(s_part[tid], s_part[tid + stride]);
        __syncthreads();
    }
    if (tid == 0) {
        block_part[3 * blockIdx.x] = s_part[0].a;
        block_part[3 * blockIdx.x + 1] = s_part[0].b;
        block_part[3 * blockIdx.x + 2] = s_part[0].len;
    }
}
// stage 2: one block joins the block results behind the running state
// With `result` set the fold starts from the empty stream (1, 0) instead of the running state and leaves the
// finished checksum b << 16 | a there.
__global__ __launch_bounds__(256) void k_adler_fold2(const uint64_t *__restrict__ block_part, uint32_t n_blocks,
                                                      uint32_t *state, uint32_t *result) {
    __shared__ AdlerPart s_part[256];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (n_blocks + 255) / 256;
    const uint32_t lo = tid * per, hi = (lo + per < n_blocks) ? lo + per : n_blocks;
    AdlerPart acc{1, 0, 0};
    for (uint32_t i = lo; i < hi; i++)
        acc = adler_join(acc, AdlerPart{block_part[3 * i], block_part[3 * i + 1], block_part[3 * i + 2]});
    s_part[tid] = acc;
    __syncthreads();
    for (uint32_t stride = 1; stride < 256; stride <<= 1) {
        if ((tid & (2 * stride - 1)) == 0) s_part[tid] = adler_join(s_part[tid], s_part[tid + stride]);
        __syncthreads();
    }
    if (tid == 0) {
        if (result) {
            result[0] = ((uint32_t)s_part[0].b << 16) | (uint32_t)s_part[0].a;
        } else {
            const AdlerPart run = adler_join(AdlerPart{state[0], state[1], 0}, s_part[0]);
            state[0] = (uint32_t)run.a;
            state[1] = (uint32_t)run.b;
        }
    }
}

hipError_t launch_adler(const uint8_t *data, uint64_t n, uint64_t seg_bytes, uint32_t n_seg, const uint32_t *seg_len,
                        uint32_t *partial, hipStream_t s) {
    if (!n_seg) return hipSuccess;
    hipLaunchKernelGGL(k_adler, dim3(n_seg), dim3(256), 0, s, data, n, seg_bytes, seg_len, partial);
    return hipGetLastError();
}
hipError_t launch_adler_fold(const uint32_t *partial, uint64_t n, uint64_t seg_bytes, uint32_t n_seg,
                             uint64_t *block_part, uint32_t *state, hipStream_t s) {
    if (!n_seg) return hipSuccess;
    const uint32_t nb = (n_seg + 1023) / 1024;
    hipLaunchKernelGGL(k_adler_fold1, dim3(nb), dim3(256), 0, s, partial, n, seg_bytes, n_seg, block_part);
    hipLaunchKernelGGL(k_adler_fold2, dim3(1), dim3(256), 0, s, block_part, nb, state, (uint32_t *)nullptr);
    return hipGetLastError();
}
hipError_t launch_adler_result(const uint32_t *partial, uint64_t n, uint64_t seg_bytes, uint32_t n_seg,
                               uint64_t *block_part, uint32_t *result, hipStream_t s) {
    const uint32_t nb = (n_seg + 1023) / 1024;  // n_seg == 0: the checksum of the empty stream, 1
    if (nb) hipLaunchKernelGGL(k_adler_fold1, dim3(nb), dim3(256), 0, s, partial, n, seg_bytes, n_seg, block_part);
    hipLaunchKernelGGL(k_adler_fold2, dim3(1), dim3(256), 0, s, block_part, nb, (uint32_t *)nullptr, result);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// k_stored: Strategy::Raw (reference src/compress.rs:172-195 with LEN = the
// block's own length): block b copies input [b*16384, ...) behind a five-byte
// header at output offset 2 + b*16389, and its Adler-32 part is taken from the
// same registers (the input is read once).  One workgroup per stored block:
// aligned 16-byte loads, 16-byte stores at the destination's own alignment.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_stored(const uint8_t *__restrict__ data, uint64_t n, uint32_t n_blocks,
                                                 uint8_t *__restrict__ out, uint32_t *__restrict__ partial) {
    __shared__ uint64_t s_s1[4], s_t[4];
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
    uint64_t s1 = 0, t = 0;
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const uint4 *v = reinterpret_cast<const uint4 *>(src);
        const uint32_t nv = len >> 4;
        for (uint32_t i = tid; i < nv; i += 256) {
            const uint4 q = v[i];
            __builtin_memcpy(dst + ((uint64_t)i << 4), &q, 16);  // unaligned 16-byte store
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
            uint32_t ls = 0, lt = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t b0 = w[k] & 0xFF, b1 = (w[k] >> 8) & 0xFF, b2 = (w[k] >> 16) & 0xFF, b3 = w[k] >> 24;
                ls += b0 + b1 + b2 + b3;
                lt += (4 * k) * b0 + (4 * k + 1) * b1 + (4 * k + 2) * b2 + (4 * k + 3) * b3;
            }
            s1 += ls;
            t += (uint64_t)(i << 4) * ls + lt;
        }
        for (uint32_t i = (nv << 4) + tid; i < len; i += 256) {
            const uint8_t x = src[i];
            dst[i] = x;
            s1 += x;
            t += (uint64_t)i * x;
        }
    } else {
        for (uint32_t i = tid; i < len; i += 256) {
            const uint8_t x = src[i];
            dst[i] = x;
            s1 += x;
            t += (uint64_t)i * x;
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
        const uint64_t bb = (len % kAdlerMod + ((len % kAdlerMod) * (s1 % kAdlerMod)) % kAdlerMod + kAdlerMod -
                             t % kAdlerMod) % kAdlerMod;
        partial[2 * b] = (uint32_t)a;
        partial[2 * b + 1] = (uint32_t)bb;
    }
}

hipError_t launch_stored(const uint8_t *data, uint64_t n, uint8_t *out, uint32_t *partial, hipStream_t s) {
    const uint64_t nb = n == 0 ? 1 : (n + kStoredBlock - 1) / kStoredBlock;
    hipLaunchKernelGGL(k_stored, dim3((uint32_t)nb), dim3(256), 0, s, data, n, (uint32_t)nb, out, partial);
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
