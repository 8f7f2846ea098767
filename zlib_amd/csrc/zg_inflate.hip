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

// ---------------------------------------------------------------------------------------------------------
// Inflate of ONE stream whose block boundaries are known (the bit offset of every block: what zg_last_block_offsets
// reports for a stream of this library), all blocks at once.  Block i holds the bytes [i * chunk, (i + 1) * chunk) of
// the output, so every block can be decoded by a wave of its own -- except that its matches may reach up to 32 KiB
// back into the output of the block before, which is being decoded at the same time.  Three steps:
//   k_inflate_blocks   a wave per block decodes into 16-bit units: a byte, or, for a byte copied from before the
//                      block's start, a marker 0x8000 | offset into the 32 KiB window in front of the block (copies
//                      of markers copy markers);
//   k_resolve_windows  one workgroup goes through the blocks in order and turns the last 32 KiB of every block --
//                      the window of the next one -- into bytes: 32 KiB of work per block, the only serial part;
//   k_resolve_all      every unit becomes a byte, markers through the block's window.
// The decoder itself reads its input from an LDS buffer that the whole wave refills (a lane that fetches its bits
// from global memory four bytes at a time waits a memory latency every few symbols) and decodes literal/length and
// distance symbols by one table look-up each (codes of up to 11 / 9 bits; longer ones go on bit by bit).
// ---------------------------------------------------------------------------------------------------------
namespace {
constexpr uint32_t kBlkRing = 65536;            // units in the output ring
constexpr uint32_t kBlkFlush = 32768 - 512;     // units produced between two flushes
constexpr uint32_t kBlkIn = 16384;              // bytes of input held in LDS
constexpr uint32_t kBlkInHalf = kBlkIn / 2;
constexpr uint32_t kLlBits = 11, kDBits = 9;    // bits the look-up tables resolve at once
constexpr uint32_t kMark = 0x8000u;

struct BlkTabs {
    uint16_t ll_lut[1u << kLlBits];  // symbol | length << 9; 0 = longer than the table resolves
    uint16_t d_lut[1u << kDBits];    // symbol | length << 5
    uint16_t ll_cnt[16], d_cnt[16];  // canonical counts and symbol order for the codes the tables do not resolve
    uint16_t ll_sym[288], d_sym[32];
    uint16_t offs[16];
    uint8_t lens[320];
};

// bits out of the LDS input ring, which holds the stream from byte `base` on, kBlkIn bytes at a time
struct LdsBits {
    const uint32_t *ring;  // kBlkIn bytes as words
    uint64_t base;         // absolute byte offset of ring word 0 (the ring is filled from there in order)
    uint64_t pos;          // bytes taken so far, counted from base (a multiple of four)
    uint64_t buf;
    uint32_t cnt;
    __device__ __forceinline__ void fill() {  // at least 32 bits in buf afterwards
        if (cnt <= 32) {
            buf |= (uint64_t)ring[(pos >> 2) & (kBlkIn / 4 - 1)] << cnt;
            cnt += 32;
            pos += 4;
        }
    }
    __device__ __forceinline__ uint32_t bits(uint32_t n) {  // n <= 16
        fill();
        const uint32_t v = (uint32_t)buf & ((1u << n) - 1u);
        buf >>= n;
        cnt -= n;
        return v;
    }
    __device__ __forceinline__ void drop(uint32_t n) {
        buf >>= n;
        cnt -= n;
    }
    __device__ __forceinline__ uint64_t taken() const { return base + pos; }                // absolute, bytes
    __device__ __forceinline__ uint64_t bit_pos() const { return (base + pos) * 8 - cnt; }  // of the next bit
};

// bit-by-bit continuation for codes longer than the table resolves (canonical decoding, as decode_sym)
__device__ __forceinline__ int slow_sym(LdsBits &b, const uint16_t *cnt, const uint16_t *sym) {
    b.fill();
    const uint32_t v = (uint32_t)b.buf;
    int code = 0, first = 0, index = 0;
    for (int l = 1; l < 16; l++) {
        code |= (int)((v >> (l - 1)) & 1u);
        const int n = (int)cnt[l];
        if (code - n < first) {
            b.drop((uint32_t)l);
            return (int)sym[index + (code - first)];
        }
        index += n;
        first += n;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}
__device__ __forceinline__ uint32_t rev_bits(uint32_t v, uint32_t n) { return __brev(v) >> (32u - n); }
}  // namespace

__global__ __launch_bounds__(kInfLanes) void k_inflate_blocks(const uint8_t *__restrict__ in, uint64_t in_bytes,
                                                               const uint64_t *__restrict__ bit_off, uint32_t n_blocks,
                                                               uint64_t chunk, uint64_t out_total,
                                                               uint16_t *__restrict__ units,
                                                               uint32_t *__restrict__ status) {
    __shared__ uint16_t s_ring[kBlkRing];
    __shared__ __attribute__((aligned(16))) uint32_t s_in[kBlkIn / 4];
    __shared__ BlkTabs s_t;
    __shared__ uint64_t s_n;       // units produced so far
    __shared__ uint64_t s_need;    // the decoder has taken every input byte below this (absolute)
    __shared__ uint32_t s_state;   // 0 running, 1 end of block, 2+ failed with InflateStatus (state - 2)
    const uint32_t blk = blockIdx.x, lane = threadIdx.x;
    const uint64_t out0 = (uint64_t)blk * chunk;
    const uint64_t want = out0 + chunk <= out_total ? chunk : out_total - out0;  // bytes this block must give
    const uint64_t bit0 = bit_off[blk];
    const uint64_t in0 = (bit0 >> 3) & ~(uint64_t)3;  // first input byte taken (four-byte steps from here on)
    uint16_t *const o = units + out0;

    // the window in front of the block: markers
    for (uint32_t j = lane; j < 32768; j += kInfLanes) s_ring[kBlkRing - 32768 + j] = (uint16_t)(kMark | j);
    // input: bytes [in0 + k * kBlkInHalf, ...) live in half (k & 1) of the ring; `loaded` = one past the last byte held
    uint64_t loaded = in0;
    auto refill = [&](uint64_t upto) {  // all lanes: load half-rings until `upto` bytes are there
        while (loaded < upto) {
            const uint32_t half = (uint32_t)((loaded - in0) / kBlkInHalf) & 1u;
            for (uint32_t w = lane; w < kBlkInHalf / 4; w += kInfLanes) {
                const uint64_t at = loaded + (uint64_t)w * 4;
                uint32_t v = 0;
                if (at + 4 <= in_bytes) {
                    __builtin_memcpy(&v, in + at, 4);
                } else {
                    for (uint32_t k = 0; k < 4; k++)
                        if (at + k < in_bytes) v |= (uint32_t)in[at + k] << (8 * k);
                }
                s_in[half * (kBlkInHalf / 4) + w] = v;
            }
            loaded += kBlkInHalf;
        }
    };
    refill(in0 + kBlkIn);
    if (lane == 0) {
        s_n = 0;
        s_state = 0;
        s_need = in0;
    }
    __syncthreads();

    LdsBits b;
    b.ring = s_in;
    b.base = in0;
    b.pos = 0;
    b.buf = 0;
    b.cnt = 0;
    uint32_t st = INF_OK;
    bool fixed = false;
    // ---- block header (lane 0) ----
    if (lane == 0) {
        b.fill();
        b.drop((uint32_t)(bit0 - in0 * 8));  // at most 31 bits
        const uint32_t bfinal = b.bits(1);
        const uint32_t type = b.bits(2);
        if ((bfinal != 0) != (blk + 1 == n_blocks)) st = INF_BAD_BLOCK;  // the last block, and only it, is final
        if (type == 0 || type == 3) {
            st = INF_BAD_BLOCK;  // stored blocks are not cut at chunks: such a stream goes through zg_decompress
        } else if (type == 1) {
            fixed = true;
            for (uint32_t i = 0; i < 288; i++) s_t.lens[i] = (uint8_t)(i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8);
            for (uint32_t i = 0; i < 30; i++) s_t.lens[288 + i] = 5;
            s_t.offs[0] = 288;
            s_t.offs[1] = 30;
        } else {
            const uint32_t nlen = b.bits(5) + 257, ndist = b.bits(5) + 1, ncode = b.bits(4) + 4;
            if (nlen > 286 || ndist > 30) st = INF_BAD_CODES;
            const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t cl[19];
            for (uint32_t i = 0; i < 19; i++) cl[i] = 0;
            for (uint32_t i = 0; i < ncode && st == INF_OK; i++) cl[order[i]] = (uint8_t)b.bits(3);
            if (st == INF_OK && !make_code(s_t.d_cnt, s_t.d_sym, s_t.offs, cl, 19)) st = INF_BAD_CODES;
            uint32_t i = 0;
            while (st == INF_OK && i < nlen + ndist) {
                const int sy = slow_sym(b, s_t.d_cnt, s_t.d_sym);
                if (sy < 0) {
                    st = INF_BAD_CODES;
                } else if (sy < 16) {
                    s_t.lens[i++] = (uint8_t)sy;
                } else {
                    uint32_t v = 0, rep;
                    if (sy == 16) {
                        if (i == 0) {
                            st = INF_BAD_CODES;
                            break;
                        }
                        v = s_t.lens[i - 1];
                        rep = 3 + b.bits(2);
                    } else if (sy == 17) {
                        rep = 3 + b.bits(3);
                    } else {
                        rep = 11 + b.bits(7);
                    }
                    if (i + rep > nlen + ndist) {
                        st = INF_BAD_CODES;
                        break;
                    }
                    while (rep--) s_t.lens[i++] = (uint8_t)v;
                }
            }
            // the distance lengths move to a place of their own behind the literal/length lengths
            if (st == INF_OK) {
                uint8_t dl[30];
                for (uint32_t k = 0; k < ndist; k++) dl[k] = s_t.lens[nlen + k];
                for (uint32_t k = nlen; k < 288; k++) s_t.lens[k] = 0;
                for (uint32_t k = 0; k < 30; k++) s_t.lens[288 + k] = k < ndist ? dl[k] : 0;
                s_t.offs[0] = (uint16_t)nlen;
                s_t.offs[1] = (uint16_t)ndist;
            }
        }
        if (st == INF_OK) {
            const uint32_t nlen = s_t.offs[0], ndist = s_t.offs[1];
            if (!make_code(s_t.ll_cnt, s_t.ll_sym, s_t.offs, s_t.lens, nlen) ||
                !make_code(s_t.d_cnt, s_t.d_sym, s_t.offs, s_t.lens + 288, ndist))
                st = INF_BAD_CODES;
        }
        s_state = st == INF_OK ? 0u : 2u + st;
    }
    (void)fixed;
    __syncthreads();
    if (s_state != 0) {
        if (lane == 0) status[blk] = s_state - 2u;
        return;
    }
    // ---- look-up tables (all lanes): canonical codes in symbol order, stored bit-reversed (the stream carries a
    // code's bits first bit first, the reader takes them as the low bits) ----
    for (uint32_t i = lane; i < (1u << kLlBits); i += kInfLanes) s_t.ll_lut[i] = 0;
    for (uint32_t i = lane; i < (1u << kDBits); i += kInfLanes) s_t.d_lut[i] = 0;
    __syncthreads();
    {
        // first code of every length
        uint32_t first_ll[16], first_d[16];
        uint32_t code = 0;
        first_ll[0] = 0;
        for (uint32_t l = 1; l < 16; l++) {
            code = (code + (l > 1 ? s_t.ll_cnt[l - 1] : 0u)) << 1;
            first_ll[l] = code;
        }
        code = 0;
        first_d[0] = 0;
        for (uint32_t l = 1; l < 16; l++) {
            code = (code + (l > 1 ? s_t.d_cnt[l - 1] : 0u)) << 1;
            first_d[l] = code;
        }
        // the rank of a symbol among those of its length = its place in the canonical order (ll_sym / d_sym):
        // lane j takes entries j, j + 64, ... of the order
        uint32_t base = 0;
        for (uint32_t l = 1; l <= kLlBits; l++) {
            const uint32_t n = s_t.ll_cnt[l];
            for (uint32_t r = lane; r < n; r += kInfLanes) {
                const uint32_t sym = s_t.ll_sym[base + r];
                const uint32_t rc = rev_bits(first_ll[l] + r, l);
                for (uint32_t k = rc; k < (1u << kLlBits); k += 1u << l) s_t.ll_lut[k] = (uint16_t)(sym | (l << 9));
            }
            base += n;
        }
        base = 0;
        for (uint32_t l = 1; l <= kDBits; l++) {
            const uint32_t n = s_t.d_cnt[l];
            for (uint32_t r = lane; r < n; r += kInfLanes) {
                const uint32_t sym = s_t.d_sym[base + r];
                const uint32_t rc = rev_bits(first_d[l] + r, l);
                for (uint32_t k = rc; k < (1u << kDBits); k += 1u << l) s_t.d_lut[k] = (uint16_t)(sym | (l << 5));
            }
            base += n;
        }
    }
    __syncthreads();

    // ---- symbols (lane 0), flushes and refills (all lanes) ----
    uint64_t n = 0, flushed = 0;
    for (;;) {
        if (lane == 0) {
            while (st == INF_OK) {
                if (n - flushed >= kBlkFlush || b.taken() + 64 > loaded) break;  // room for the longest match / bits at hand
                b.fill();
                uint32_t e = s_t.ll_lut[(uint32_t)b.buf & ((1u << kLlBits) - 1u)];
                int sy;
                if (e) {
                    b.drop(e >> 9);
                    sy = (int)(e & 0x1FFu);
                } else {
                    sy = slow_sym(b, s_t.ll_cnt, s_t.ll_sym);
                }
                if (sy < 0) {
                    st = INF_BAD_SYMBOL;
                } else if (sy < 256) {
                    s_ring[n & (kBlkRing - 1)] = (uint16_t)sy;
                    n++;
                } else if (sy == 256) {
                    st = 0x100u;  // end of block
                } else {
                    const uint32_t li = (uint32_t)sy - 257u;
                    if (li > 28) {
                        st = INF_BAD_SYMBOL;
                        break;
                    }
                    const uint32_t len = len_base(li) + b.bits(len_extra(li));
                    b.fill();
                    e = s_t.d_lut[(uint32_t)b.buf & ((1u << kDBits) - 1u)];
                    int ds;
                    if (e) {
                        b.drop(e >> 5);
                        ds = (int)(e & 0x1Fu);
                    } else {
                        ds = slow_sym(b, s_t.d_cnt, s_t.d_sym);
                    }
                    if (ds < 0 || ds > 29) {
                        st = INF_BAD_SYMBOL;
                        break;
                    }
                    const uint32_t dist = dst_base((uint32_t)ds) + b.bits(dst_extra((uint32_t)ds));
                    // before the start of the stream, or more than a window back
                    if (dist > n + (blk ? 32768u : 0u) || (blk && dist > n && out0 < dist - n)) {
                        st = INF_BAD_DISTANCE;
                        break;
                    }
                    for (uint32_t k = 0; k < len; k++) {
                        s_ring[n & (kBlkRing - 1)] = s_ring[(n - dist) & (kBlkRing - 1)];
                        n++;
                    }
                }
                if (n > want) st = INF_OVERFLOW;  // more than the block's share of the output
            }
            s_n = n;
            s_need = b.taken();
            s_state = st == INF_OK ? 0u : (st == 0x100u ? 1u : 2u + st);
        }
        __syncthreads();
        n = s_n;
        const uint32_t state = s_state;
        const uint64_t upto = n < want ? n : want;
        for (uint64_t j = flushed + lane; j < upto; j += kInfLanes) o[j] = s_ring[j & (kBlkRing - 1)];
        flushed = n;
        if (state != 0) {
            if (lane == 0) {
                uint32_t fin = state == 1 ? INF_OK : state - 2u;
                if (fin == INF_OK && n != want) fin = INF_TRUNCATED;  // the block ended short of its share
                // where the block ended: the next block must begin there (the last one: the trailer, after padding)
                if (fin == INF_OK && blk + 1 < n_blocks && b.bit_pos() != bit_off[blk + 1]) fin = INF_BAD_BLOCK;
                if (blk + 1 == n_blocks) {  // behind the statuses: the bit at which the last block ended
                    const uint64_t e = b.bit_pos();
                    status[n_blocks] = (uint32_t)e;
                    status[n_blocks + 1] = (uint32_t)(e >> 32);
                }
                status[blk] = fin;
            }
            return;
        }
        refill(s_need + kBlkIn - kBlkInHalf);  // every half below the reader's position is free
        __syncthreads();
    }
}

// the window of block i + 1 = the last 32 KiB of block i, as bytes.  One workgroup, blocks in order; the window of
// the block in hand stays in LDS.  win holds 32 KiB per block (block 0 has none).
__global__ __launch_bounds__(1024) void k_resolve_windows(const uint16_t *__restrict__ units, uint32_t n_blocks,
                                                           uint64_t chunk, uint8_t *__restrict__ win) {
    __shared__ uint8_t s_win[2][32768];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = 0; i + 1 < n_blocks; i++) {
        const uint16_t *tail = units + (uint64_t)(i + 1) * chunk - 32768;  // chunk >= 32768
        const uint8_t *prev = s_win[i & 1];
        uint8_t *cur = s_win[(i + 1) & 1];
        uint8_t *w = win + (uint64_t)(i + 1) * 32768;
        for (uint32_t j = tid; j < 32768; j += 1024) {
            const uint32_t u = tail[j];
            const uint8_t v = (u & kMark) ? prev[u & 0x7FFFu] : (uint8_t)u;  // block 0 holds no markers
            cur[j] = v;
            w[j] = v;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_resolve_all(const uint16_t *__restrict__ units, uint64_t n, uint64_t chunk,
                                                      const uint8_t *__restrict__ win, uint8_t *__restrict__ out) {
    const uint64_t p0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (p0 >= n) return;
    if (p0 + 8 <= n && (p0 % chunk) + 8 <= chunk) {
        const uint4 q = *reinterpret_cast<const uint4 *>(units + p0);  // p0 is a multiple of 8: 16-byte aligned
        const uint32_t u[8] = {q.x & 0xFFFFu, q.x >> 16, q.y & 0xFFFFu, q.y >> 16, q.z & 0xFFFFu, q.z >> 16, q.w & 0xFFFFu, q.w >> 16};
        const uint8_t *w = win + (p0 / chunk) * 32768;
        uint64_t r = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t v = (u[k] & kMark) ? w[u[k] & 0x7FFFu] : (u[k] & 0xFFu);
            r |= (uint64_t)v << (8 * k);
        }
        *reinterpret_cast<uint64_t *>(out + p0) = r;
    } else {
        for (uint64_t p = p0; p < p0 + 8 && p < n; p++) {
            const uint32_t u = units[p];
            out[p] = (u & kMark) ? win[(p / chunk) * 32768 + (u & 0x7FFFu)] : (uint8_t)u;
        }
    }
}

hipError_t launch_inflate_blocks(const uint8_t *in, uint64_t in_bytes, const uint64_t *bit_off, uint32_t n_blocks,
                                 uint64_t chunk, uint64_t out_total, uint16_t *units, uint8_t *win, uint8_t *out,
                                 uint32_t *status, hipStream_t s) {
    if (!n_blocks) return hipSuccess;
    hipLaunchKernelGGL(k_inflate_blocks, dim3(n_blocks), dim3(kInfLanes), 0, s, in, in_bytes, bit_off, n_blocks, chunk,
                       out_total, units, status);
    if (n_blocks > 1)
        hipLaunchKernelGGL(k_resolve_windows, dim3(1), dim3(1024), 0, s, units, n_blocks, chunk, win);
    if (out_total)
        hipLaunchKernelGGL(k_resolve_all, dim3((uint32_t)((out_total + 2047) / 2048)), dim3(256), 0, s, units, out_total,
                           chunk, win, out);
    return hipGetLastError();
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
