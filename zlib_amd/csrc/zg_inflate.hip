// Inflate on the device: zlib::decompress (reference src/decompress.rs:34-189, tree decoding
// src/huffman.rs:469-592, bit order src/bitreader.rs:59-139).  Decoding one DEFLATE stream is serial, so the
// parallelism is across streams: k_inflate gives every stream of a record batch a lane; k_inflate_wave gives
// each of a few large streams a wave, whose first lane decodes into an LDS ring (the 32 KiB window never
// leaves LDS) while all lanes write the output and sum the Adler-32.  Codes are decoded canonically, one bit at a time, against per-length
// counts (the reference walks a pointer tree one bit at a time; both accept exactly the prefix codes that the
// code lengths describe).
//
// Where the reference panics or misbehaves on a corrupt stream (missing distance symbol, distance before the
// start of the output, reads past the end of the input) the kernel reports a status instead.
#include "zg_internal.h"

namespace zg {

namespace {

constexpr uint32_t kInfLanes = 64;

// decode tables of one stream, in LDS
struct InflateTabs {
    uint16_t ll_cnt[16];
    uint16_t ll_sym[288];
    uint16_t d_cnt[16];
    uint16_t d_sym[32];
    uint8_t lens[320];
    uint16_t offs[16];  // make_code scratch
};

struct BitSrc {
    const uint8_t *in;
    uint64_t pos, end;  // next byte to fetch, one past the last
    uint64_t buf;
    uint32_t cnt;       // bits in buf, including `pad` zero bits appended once the input ran out
    uint32_t pad;
    bool over;          // bits were consumed that the input does not have

    __device__ __forceinline__ void fill(uint32_t need) {
        if (cnt <= 32 && pos + 4 <= end) {  // the common case: four bytes with one (unaligned) load
            uint32_t w;
            __builtin_memcpy(&w, in + pos, 4);
            buf |= (uint64_t)w << cnt;
            cnt += 32;
            pos += 4;
            return;
        }
        while (cnt < need) {
            if (pos < end) {
                buf |= (uint64_t)in[pos] << cnt;
                pos++;
            } else {
                pad += 8;
            }
            cnt += 8;
        }
    }
    __device__ __forceinline__ void drop(uint32_t n) {
        buf >>= n;
        cnt -= n;
        if (cnt < pad) over = true;
    }
    __device__ __forceinline__ uint32_t bits(uint32_t n) {  // n <= 16, LSB first (BitReader::read_bits)
        if (n == 0) return 0;
        fill(n);
        const uint32_t v = (uint32_t)buf & ((1u << n) - 1u);
        drop(n);
        return v;
    }
    // BitReader::read_byte drops what is left of the current byte: give whole buffered bytes back
    __device__ __forceinline__ void align() {
        const uint32_t real = cnt > pad ? cnt - pad : 0u;
        pos -= real >> 3;
        buf = 0;
        cnt = 0;
        pad = 0;
    }
};

// code counts per length 1..15, two to a register, so that decoding a symbol touches no memory until the
// symbol itself is looked up
struct Counts {
    uint32_t w[8];
    __device__ __forceinline__ void load(const uint16_t *cnt) {
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = (uint32_t)cnt[2 * i] | ((uint32_t)cnt[2 * i + 1] << 16);
    }
};

// counts per code length and the symbols in canonical order; false when the lengths over-subscribe the code space
__device__ bool make_code(uint16_t *cnt, uint16_t *sym, uint16_t *offs, const uint8_t *len, uint32_t n) {
    for (uint32_t i = 0; i < 16; i++) cnt[i] = 0;
    for (uint32_t i = 0; i < n; i++) cnt[len[i]]++;
    int left = 1;
    for (uint32_t l = 1; l < 16; l++) {
        left <<= 1;
        left -= (int)cnt[l];
        if (left < 0) return false;
    }
    offs[1] = 0;
    for (uint32_t l = 1; l < 15; l++) offs[l + 1] = (uint16_t)(offs[l] + cnt[l]);
    for (uint32_t i = 0; i < n; i++)
        if (len[i]) sym[offs[len[i]]++] = (uint16_t)i;
    return true;
}

// one symbol; -1 when the bits match no code.  Canonical decoding: at every length, the codes of that length
// are the next `count` values after the (doubled) first code of the previous length.
__device__ __forceinline__ int decode_sym(BitSrc &b, const Counts &c, const uint16_t *sym) {
    b.fill(15);
    const uint32_t v = (uint32_t)b.buf;
    int code = 0, first = 0, index = 0;
#pragma unroll
    for (int l = 1; l < 16; l++) {
        code |= (int)((v >> (l - 1)) & 1u);
        const int n = (int)((c.w[l >> 1] >> ((l & 1) * 16)) & 0xFFFFu);
        if (code - n < first) {
            b.drop((uint32_t)l);
            return (int)sym[index + (code - first)];
        }
        index += n;
        first += n;
        first <<= 1;
        code <<= 1;
    }
    b.drop(15);
    return -1;
}

__device__ __forceinline__ uint32_t len_base(uint32_t i) {
    return i < 8 ? 3 + i : (i == 28 ? 258u : ((4u + (i & 3u)) << ((i >> 2) - 1u)) + 3u);
}
__device__ __forceinline__ uint32_t len_extra(uint32_t i) { return (i < 8 || i == 28) ? 0u : (i >> 2) - 1u; }
__device__ __forceinline__ uint32_t dst_base(uint32_t c) { return c < 4 ? c + 1 : ((2u + (c & 1u)) << ((c >> 1) - 1u)) + 1u; }
__device__ __forceinline__ uint32_t dst_extra(uint32_t c) { return c < 4 ? 0u : (c >> 1) - 1u; }

// Decoder state that survives an interruption (the wave kernel pauses decoding to flush its output ring).
struct InfState {
    BitSrc b;
    Counts cl, cd;         // literal/length and distance code counts of the current block
    uint32_t phase;        // 0: at a block header, 1: inside a coded block, 2: inside a stored block, 3: finished
    uint32_t stored_left;  // bytes of the current stored block not yet copied
    uint32_t st;           // InflateStatus
    bool last;             // the current block is the final one
};

__device__ __forceinline__ void inf_begin(InfState &s, const uint8_t *in, uint64_t pos, uint64_t end) {
    s.b.in = in;
    s.b.pos = pos;
    s.b.end = end;
    s.b.buf = 0;
    s.b.cnt = 0;
    s.b.pad = 0;
    s.b.over = false;
    s.phase = 0;
    s.stored_left = 0;
    s.st = INF_OK;
    s.last = false;
    // header, src/decompress.rs:37-69
    const uint32_t cmf = s.b.bits(8), flg = s.b.bits(8);
    if ((cmf & 15u) != 8u)
        s.st = INF_BAD_METHOD;
    else if ((cmf >> 4) > 7u)
        s.st = INF_BAD_WINDOW;
    else if ((cmf * 256u + flg) % 31u != 0u)
        s.st = INF_BAD_FCHECK;
    else if ((flg >> 5) & 1u)
        s.st = INF_PRESET_DICT;
    if (s.st != INF_OK) s.phase = 3;
}

// Decodes until the stream ends, an error is found, or the sink asks for a pause.  A sink offers
// put(byte) / copy(dist, len) (both may set an error through fail()), size() = bytes produced so far, and
// pause() = "stop after this symbol".
template <class Sink>
__device__ __forceinline__ void inf_run(InfState &s, InflateTabs &t, Sink &sink) {
    BitSrc &b = s.b;
    while (s.phase != 3 && s.st == INF_OK && !sink.pause()) {
        if (b.over) {
            s.st = INF_TRUNCATED;
            break;
        }
        if (s.phase == 0) {
            if (s.last) {
                s.phase = 3;
                break;
            }
            s.last = b.bits(1) != 0;
            const uint32_t type = b.bits(2);
            if (type == 0) {  // stored, src/decompress.rs:124-132 (NLEN is read and ignored there too)
                b.align();
                s.stored_left = b.bits(16);
                (void)b.bits(16);
                b.align();
                if (b.over || b.pos + s.stored_left > b.end) {
                    s.st = INF_TRUNCATED;
                    break;
                }
                s.phase = 2;
                continue;
            }
            if (type == 3) {
                s.st = INF_BAD_BLOCK;
                break;
            }
            if (type == 1) {  // get_zlib_fixed, src/huffman.rs:353-367
                for (uint32_t i = 0; i < 288; i++) t.lens[i] = (uint8_t)(i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8);
                make_code(t.ll_cnt, t.ll_sym, t.offs, t.lens, 288);
                for (uint32_t i = 0; i < 30; i++) t.lens[i] = 5;
                make_code(t.d_cnt, t.d_sym, t.offs, t.lens, 30);
            } else {  // decode_trees, src/huffman.rs:469-592
                const uint32_t nlen = b.bits(5) + 257, ndist = b.bits(5) + 1, ncode = b.bits(4) + 4;
                if (nlen > 286 || ndist > 30) {
                    s.st = INF_BAD_CODES;
                    break;
                }
                const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                for (uint32_t i = 0; i < 19; i++) t.lens[i] = 0;
                for (uint32_t i = 0; i < ncode; i++) t.lens[order[i]] = (uint8_t)b.bits(3);
                if (!make_code(t.d_cnt, t.d_sym, t.offs, t.lens, 19)) {  // the code-length code borrows the distance tables
                    s.st = INF_BAD_CODES;
                    break;
                }
                Counts cc;
                cc.load(t.d_cnt);
                uint32_t i = 0;
                while (i < nlen + ndist && s.st == INF_OK) {
                    const int sy = decode_sym(b, cc, t.d_sym);
                    if (sy < 0) {
                        s.st = INF_BAD_CODES;
                    } else if (sy < 16) {
                        t.lens[i++] = (uint8_t)sy;
                    } else {
                        uint32_t v = 0, rep;
                        if (sy == 16) {
                            if (i == 0) {
                                s.st = INF_BAD_CODES;
                                break;
                            }
                            v = t.lens[i - 1];
                            rep = 3 + b.bits(2);
                        } else if (sy == 17) {
                            rep = 3 + b.bits(3);
                        } else {
                            rep = 11 + b.bits(7);
                        }
                        if (i + rep > nlen + ndist) {
                            s.st = INF_BAD_CODES;
                            break;
                        }
                        while (rep--) t.lens[i++] = (uint8_t)v;
                    }
                }
                if (s.st != INF_OK) break;
                if (!make_code(t.ll_cnt, t.ll_sym, t.offs, t.lens, nlen) ||
                    !make_code(t.d_cnt, t.d_sym, t.offs, t.lens + nlen, ndist)) {
                    s.st = INF_BAD_CODES;
                    break;
                }
            }
            s.cl.load(t.ll_cnt);
            s.cd.load(t.d_cnt);
            s.phase = 1;
        } else if (s.phase == 2) {
            while (s.stored_left && s.st == INF_OK && !sink.pause()) {
                sink.put(b.in[b.pos], s.st);
                b.pos++;
                s.stored_left--;
            }
            if (!s.stored_left) s.phase = 0;
        } else {  // inflate_block_data, src/decompress.rs:151-189
            const int sy = decode_sym(b, s.cl, t.ll_sym);
            if (sy < 0) {
                s.st = INF_BAD_SYMBOL;
            } else if (sy < 256) {
                sink.put((uint32_t)sy, s.st);
            } else if (sy == 256) {
                s.phase = 0;
            } else {
                const uint32_t li = (uint32_t)sy - 257u;
                if (li > 28) {
                    s.st = INF_BAD_SYMBOL;
                    break;
                }
                const uint32_t len = len_base(li) + b.bits(len_extra(li));
                const int ds = decode_sym(b, s.cd, t.d_sym);
                if (ds < 0 || ds > 29) {
                    s.st = INF_BAD_SYMBOL;
                    break;
                }
                const uint32_t dist = dst_base((uint32_t)ds) + b.bits(dst_extra((uint32_t)ds));
                if (dist > sink.size())
                    s.st = INF_BAD_DISTANCE;
                else
                    sink.copy(dist, len, s.st);
            }
        }
    }
    if (s.st == INF_OK && b.over) s.st = INF_TRUNCATED;
}

// the four bytes after the last block, big endian (src/decompress.rs:74-87)
__device__ __forceinline__ uint32_t inf_trailer(InfState &s) {
    s.b.align();
    uint32_t want = 0;
    for (int i = 0; i < 4; i++) want = (want << 8) | s.b.bits(8);
    if (s.b.over) s.st = INF_TRUNCATED;
    return want;
}

// --- sinks ---------------------------------------------------------------------------------------------
struct CountSink {  // lengths only: a match adds its length, nothing is read back
    uint64_t n = 0;
    __device__ __forceinline__ bool pause() const { return false; }
    __device__ __forceinline__ uint64_t size() const { return n; }
    __device__ __forceinline__ void put(uint32_t, uint32_t &) { n++; }
    __device__ __forceinline__ void copy(uint32_t, uint32_t len, uint32_t &) { n += len; }
};

struct GlobalSink {  // one lane writes its stream's bytes straight to global memory and sums Adler-32 as it goes
    uint8_t *o;
    uint64_t cap, n = 0;
    uint32_t a = 1, s2 = 0, pending = 0;  // src/adler.rs:5-13 with the modulo deferred
    __device__ __forceinline__ bool pause() const { return false; }
    __device__ __forceinline__ uint64_t size() const { return n; }
    __device__ __forceinline__ void settle() {
        a %= 65521u;
        s2 %= 65521u;
        pending = 0;
    }
    __device__ __forceinline__ void put(uint32_t byte, uint32_t &st) {
        if (n >= cap) {
            st = INF_OVERFLOW;
            return;
        }
        o[n++] = (uint8_t)byte;
        a += byte;
        s2 += a;
        if (++pending >= 5000) settle();  // 5552 bytes is the most the 32-bit sums can take; a match adds up to 258
    }
    __device__ __forceinline__ void copy(uint32_t dist, uint32_t len, uint32_t &st) {
        uint32_t k = 0;
        if (dist >= 4 && n + len <= cap) {  // source and destination do not overlap within four bytes
            for (; k + 4 <= len; k += 4) {
                uint32_t w;
                __builtin_memcpy(&w, o + n - dist, 4);
                __builtin_memcpy(o + n, &w, 4);
                for (uint32_t j = 0; j < 4; j++) {
                    a += (w >> (8 * j)) & 0xFFu;
                    s2 += a;
                }
                n += 4;
                pending += 4;
            }
            if (pending >= 5000) settle();
        }
        for (; k < len && st == INF_OK; k++) put(o[n - dist], st);
    }
};

// The wave kernel's sink: lane 0 decodes into a 64 KiB ring in LDS (which always holds the 32 KiB window);
// every kRingFlush bytes the whole wave writes the new part to global memory and folds it into the Adler-32.
constexpr uint32_t kRing = 65536, kRingFlush = 32768 - 512;
struct RingSink {
    uint8_t *ring;
    uint64_t cap, n = 0, flushed = 0;
    __device__ __forceinline__ bool pause() const { return n - flushed >= kRingFlush; }
    __device__ __forceinline__ uint64_t size() const { return n; }
    __device__ __forceinline__ void put(uint32_t byte, uint32_t &st) {
        if (n >= cap) {
            st = INF_OVERFLOW;
            return;
        }
        ring[n & (kRing - 1)] = (uint8_t)byte;
        n++;
    }
    __device__ __forceinline__ void copy(uint32_t dist, uint32_t len, uint32_t &st) {
        if (n + len > cap) {
            st = INF_OVERFLOW;
            return;
        }
        for (uint32_t k = 0; k < len; k++) {
            ring[n & (kRing - 1)] = ring[(n - dist) & (kRing - 1)];
            n++;
        }
    }
};

}  // namespace

// One lane per stream (record batches).  out == nullptr: lengths only.
__global__ __launch_bounds__(kInfLanes) void k_inflate(const uint8_t *__restrict__ in,
                                                        const uint64_t *__restrict__ in_off, uint32_t n_streams,
                                                        uint8_t *__restrict__ out,
                                                        const uint64_t *__restrict__ out_off,
                                                        uint64_t *__restrict__ out_len, uint32_t *__restrict__ status,
                                                        int size_only) {
    __shared__ InflateTabs s_tabs[kInfLanes];
    const uint32_t sid = blockIdx.x * kInfLanes + threadIdx.x;
    if (sid >= n_streams) return;
    InflateTabs &t = s_tabs[threadIdx.x];
    InfState s;
    inf_begin(s, in, in_off[sid], in_off[sid + 1]);
    if (size_only) {
        CountSink sink;
        inf_run(s, t, sink);
        out_len[sid] = sink.n;
    } else {
        GlobalSink sink;
        sink.o = out + out_off[sid];
        sink.cap = out_off[sid + 1] - out_off[sid];
        inf_run(s, t, sink);
        if (s.st == INF_OK) {
            const uint32_t want = inf_trailer(s);
            sink.settle();
            if (s.st == INF_OK && want != ((sink.s2 << 16) | sink.a)) s.st = INF_BAD_CHECKSUM;
        }
        out_len[sid] = sink.n;
    }
    status[sid] = s.st;
}

// One wave per stream (few, large streams): lane 0 decodes into the LDS ring, all lanes flush.
__global__ __launch_bounds__(kInfLanes) void k_inflate_wave(const uint8_t *__restrict__ in,
                                                             const uint64_t *__restrict__ in_off,
                                                             uint8_t *__restrict__ out,
                                                             const uint64_t *__restrict__ out_off,
                                                             uint64_t *__restrict__ out_len,
                                                             uint32_t *__restrict__ status) {
    __shared__ __attribute__((aligned(16))) uint8_t s_ring[kRing];
    __shared__ InflateTabs s_tab;
    __shared__ uint64_t s_n;
    __shared__ uint32_t s_done;
    const uint32_t sid = blockIdx.x, lane = threadIdx.x;
    uint8_t *o = out + out_off[sid];
    InfState s;
    RingSink sink;
    sink.ring = s_ring;
    sink.cap = out_off[sid + 1] - out_off[sid];
    if (lane == 0) inf_begin(s, in, in_off[sid], in_off[sid + 1]);
    uint64_t flushed = 0;
    uint64_t a = 1, b2 = 0;  // Adler-32 state, kept by every lane identically
    for (;;) {
        if (lane == 0) {
            inf_run(s, s_tab, sink);
            s_n = sink.n;
            s_done = (s.phase == 3 || s.st != INF_OK) ? 1u : 0u;
        }
        __syncthreads();
        const uint64_t n = s_n;
        const uint32_t done = s_done;
        // flush [flushed, n): at most kRingFlush + 258 bytes, still in the ring.  For bytes x_0..x_{L-1}:
        // a' = a + sum x, b' = b + L*a + sum (L - j) x_j
        const uint32_t L = (uint32_t)(n - flushed);
        uint64_t s1 = 0, tw = 0;
        for (uint32_t j = lane; j < L; j += kInfLanes) {
            const uint32_t x = s_ring[(flushed + j) & (kRing - 1)];
            o[flushed + j] = (uint8_t)x;
            s1 += x;
            tw += (uint64_t)(L - j) * x;
        }
        for (int off = 32; off > 0; off >>= 1) {
            s1 += __shfl_xor(s1, off);
            tw += __shfl_xor(tw, off);
        }
        b2 = (b2 + (uint64_t)L % 65521u * a + tw) % 65521u;
        a = (a + s1) % 65521u;
        flushed = n;
        __syncthreads();
        if (lane == 0) sink.flushed = flushed;
        if (done) break;
    }
    if (lane == 0) {
        if (s.st == INF_OK) {
            const uint32_t want = inf_trailer(s);
            if (s.st == INF_OK && want != (uint32_t)((b2 << 16) | a)) s.st = INF_BAD_CHECKSUM;
        }
        out_len[sid] = flushed;
        status[sid] = s.st;
    }
}

hipError_t launch_inflate(const uint8_t *in, const uint64_t *in_off, uint32_t n_streams, uint8_t *out,
                          const uint64_t *out_off, uint64_t *out_len, uint32_t *status, bool size_only, bool large,
                          hipStream_t s) {
    if (!n_streams) return hipSuccess;
    if (large && !size_only)
        hipLaunchKernelGGL(k_inflate_wave, dim3(n_streams), dim3(kInfLanes), 0, s, in, in_off, out, out_off, out_len,
                           status);
    else
        hipLaunchKernelGGL(k_inflate, dim3((n_streams + kInfLanes - 1) / kInfLanes), dim3(kInfLanes), 0, s, in, in_off,
                           n_streams, out, out_off, out_len, status, size_only ? 1 : 0);
    return hipGetLastError();
}

}  // namespace zg
