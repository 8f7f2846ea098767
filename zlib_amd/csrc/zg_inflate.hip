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
// back into the output of the block before, which is being decoded at the same time.  Four steps:
//   k_decode_tokens    a wave per block turns the block's bits into tokens (a literal, or length and distance).
//                      Decoding is serial, so the wave runs it in step: all state is the same in every lane, the
//                      compiler keeps it in scalar registers and the loop's branches are scalar branches -- a lane
//                      decoding alone under an execution mask needs several times the instructions.  Input words
//                      come through scalar loads, one word ahead of their use; literal/length and distance symbols
//                      are resolved by one look-up each in tables in LDS (codes of up to 10 / 9 bits, with the base
//                      value and the number of extra bits in the entry; longer codes go on bit by bit).  Token k of a
//                      group of 64 is written into lane k of a register and the group stored with one instruction.
//                      No output ring: 7 KiB of LDS per block, a dozen blocks per CU.
//   k_expand_tokens    a wave per block turns tokens into 16-bit units: a byte, or, for a byte copied from before
//                      the block's start, a marker 0x8000 | offset into the 32 KiB window in front of the block
//                      (copies of markers copy markers).  64 tokens at a time: a scan of their lengths gives the
//                      positions, runs of literals are written by their lanes, every match is copied by the whole
//                      wave, in stream order, in an LDS ring that holds the last 32 Ki units.
//   k_resolve_windows  one workgroup goes through the blocks in order and turns the last 32 KiB of every block --
//                      the window of the next one -- into bytes: 32 KiB of work per block, the only serial part;
//   k_resolve_all      every unit becomes a byte, markers through the block's window.
// Tokens and units are kept for a pass of blocks at a time (the windows for all).
// ---------------------------------------------------------------------------------------------------------
namespace {
constexpr uint32_t kBlkRing = 32768;            // units in the output ring: exactly the window
constexpr uint32_t kBlkFlush = 8192;            // units produced between two flushes (at least)
constexpr uint32_t kLlBits = 10, kDBits = 9;    // bits the look-up tables resolve at once
constexpr uint32_t kMark = 0x8000u;
constexpr uint32_t kTokMatch = 0x80000000u;     // a match token: this | (length - 3) << 15 | (distance - 1); else the byte
// literal/length entry: code length | extra bits << 4 | value << 8 (the byte, or the base length) | kind << 30
// (0 literal, 1 end of block, 2 length); distance entry: code length | extra bits << 4 | base distance << 8.
// 0 = the code is longer than the table resolves.  Symbols that must not occur have the base value 0.
constexpr uint32_t kEntEob = 0x40000000u, kEntLen = 0x80000000u;

struct BlkTabs {
    uint32_t ll_lut[1u << kLlBits];
    uint32_t d_lut[1u << kDBits];
    uint16_t ll_cnt[16], d_cnt[16];  // canonical counts and symbol order for the codes the tables do not resolve
    uint16_t ll_sym[288], d_sym[32];
    uint16_t offs[16];
    uint8_t lens[320];
};

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

__device__ __forceinline__ uint32_t ll_entry(uint32_t sym, uint32_t code_len) {
    if (sym < 256) return code_len | (sym << 8);
    if (sym == 256) return code_len | kEntEob;
    const uint32_t li = sym - 257u;
    if (li > 28) return code_len | kEntLen;  // 286, 287: base 0
    return code_len | (len_extra(li) << 4) | (len_base(li) << 8) | kEntLen;
}
__device__ __forceinline__ uint32_t d_entry(uint32_t sym, uint32_t code_len) {
    if (sym > 29) return code_len;  // 30, 31: base 0
    return code_len | (dst_extra(sym) << 4) | (dst_base(sym) << 8);
}

// Bits of the stream for a decoder that the whole wave runs in step.  `in` is four-byte aligned and the block's
// first word; offsets are 32-bit (a block's bits are far fewer than 2^32 bytes; 64-bit comparisons have no scalar
// form).  Bytes from `lim` on read as zero.
struct WaveBits {
    const uint8_t *in;
    uint32_t lim;    // bytes of the stream from `in` on (saturated)
    uint32_t pos;    // bytes taken into buf so far (a multiple of four); the word at pos is already in nextw
    uint64_t buf;
    uint32_t cnt, nextw;
    __device__ __forceinline__ uint32_t word(uint32_t p) const {
        if (__builtin_expect(p + 4u <= lim, 1)) return *reinterpret_cast<const uint32_t *>(in + p);  // a scalar load: not waited for here
        uint32_t v = 0;
        for (uint32_t k = 0; k < 4; k++)
            if (p + k < lim) v |= (uint32_t)in[p + k] << (8 * k);
        return uni(v);
    }
    __device__ __forceinline__ void start(uint32_t skip_bits) {  // skip_bits < 32
        pos = 0;
        buf = 0;
        cnt = 0;
        nextw = word(0);
        fill();
        drop(skip_bits);
    }
    __device__ __forceinline__ void fill() {  // at least 33 bits in buf afterwards
        if (cnt <= 32) {
            buf |= (uint64_t)nextw << cnt;
            cnt += 32;
            pos += 4;
            nextw = word(pos);
        }
    }
    __device__ __forceinline__ void drop(uint32_t n) {
        buf >>= n;
        cnt -= n;
    }
    __device__ __forceinline__ uint64_t bit_pos() const { return (uint64_t)pos * 8 - cnt; }  // of the next bit, from `in`
};

// code counts per length 1..15 in registers, the same in every lane
struct WaveCounts {
    uint32_t w[8];
    __device__ __forceinline__ void load(const uint16_t *cnt) {
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = uni((uint32_t)cnt[2 * i] | ((uint32_t)cnt[2 * i + 1] << 16));
    }
};
// bit-by-bit continuation for codes longer than the table resolves (canonical decoding, as decode_sym); the code's
// bits are dropped.  -1: the bits match no code.
__device__ __forceinline__ int wave_slow_sym(WaveBits &b, const WaveCounts &c, const uint16_t *sym) {
    const uint32_t v = (uint32_t)b.buf;
    int code = 0, first = 0, index = 0;
    for (int l = 1; l < 16; l++) {
        code |= (int)((v >> (l - 1)) & 1u);
        const int n = (int)((c.w[l >> 1] >> ((l & 1) * 16)) & 0xFFFFu);
        if (code - n < first) {
            b.drop((uint32_t)l);
            return (int)uni(sym[index + (code - first)]);
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

// tok: tok_stride entries per block (at least chunk + 64, a multiple of 64); ntok / status: one entry per block of the
// pass (status has two more entries behind the stream's last block: the bit at which that block ended).  `in` is
// four-byte aligned and the bit offsets count from it.
__global__ __launch_bounds__(kInfLanes) void k_decode_tokens(const uint8_t *__restrict__ in, uint64_t in_bytes,
                                                              const uint64_t *__restrict__ bit_off, uint64_t bit_shift,
                                                              uint32_t b0, uint32_t n_blocks, uint64_t chunk,
                                                              uint64_t out_total, uint32_t *__restrict__ tok,
                                                              uint64_t tok_stride, uint32_t *__restrict__ ntok,
                                                              uint32_t *__restrict__ status) {
    __shared__ BlkTabs s_t;
    __shared__ uint64_t s_bit;     // where the header ended
    __shared__ uint32_t s_state;   // 0 running, 2+ failed with InflateStatus (state - 2)
    const uint32_t blk = b0 + blockIdx.x, lane = threadIdx.x;
    const uint64_t out0 = (uint64_t)blk * chunk;
    const uint64_t want = out0 + chunk <= out_total ? chunk : out_total - out0;  // bytes this block must give
    const uint64_t bit0 = bit_off[blk] + bit_shift;
    uint32_t *const tk = tok + (uint64_t)blockIdx.x * tok_stride;

    // ---- block header (lane 0; a few hundred bits) ----
    if (lane == 0) {
        uint32_t st = INF_OK;
        BitSrc b;
        b.in = in;
        b.pos = bit0 >> 3;
        b.end = in_bytes;
        b.buf = 0;
        b.cnt = 0;
        b.pad = 0;
        b.over = false;
        (void)b.bits((uint32_t)(bit0 & 7));
        const uint32_t bfinal = b.bits(1);
        const uint32_t type = b.bits(2);
        if ((bfinal != 0) != (blk + 1 == n_blocks)) st = INF_BAD_BLOCK;  // the last block, and only it, is final
        if (type == 0 || type == 3) {
            st = INF_BAD_BLOCK;  // stored blocks are not cut at chunks: such a stream goes through zg_decompress
        } else if (type == 1) {
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
            Counts cc;
            cc.load(s_t.d_cnt);
            uint32_t i = 0;
            while (st == INF_OK && i < nlen + ndist) {
                const int sy = decode_sym(b, cc, s_t.d_sym);
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
        if (st == INF_OK && (b.over || b.pad)) st = INF_TRUNCATED;
        s_bit = b.pos * 8 - b.cnt;
        s_state = st == INF_OK ? 0u : 2u + st;
    }
    __syncthreads();
    if (s_state != 0) {
        if (lane == 0) {
            status[blk] = s_state - 2u;
            ntok[blockIdx.x] = 0;
        }
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
                const uint32_t e = ll_entry(s_t.ll_sym[base + r], l);
                const uint32_t rc = rev_bits(first_ll[l] + r, l);
                for (uint32_t k = rc; k < (1u << kLlBits); k += 1u << l) s_t.ll_lut[k] = e;
            }
            base += n;
        }
        base = 0;
        for (uint32_t l = 1; l <= kDBits; l++) {
            const uint32_t n = s_t.d_cnt[l];
            for (uint32_t r = lane; r < n; r += kInfLanes) {
                const uint32_t e = d_entry(s_t.d_sym[base + r], l);
                const uint32_t rc = rev_bits(first_d[l] + r, l);
                for (uint32_t k = rc; k < (1u << kDBits); k += 1u << l) s_t.d_lut[k] = e;
            }
            base += n;
        }
    }
    __syncthreads();

    // ---- symbols: the whole wave in step ----
    WaveCounts llc, dc;
    llc.load(s_t.ll_cnt);
    dc.load(s_t.d_cnt);
    const uint64_t hdr_bit = (uint64_t)uni((uint32_t)s_bit) | ((uint64_t)uni((uint32_t)(s_bit >> 32)) << 32);
    const uint64_t w0 = (hdr_bit >> 3) & ~(uint64_t)3;  // the word the first symbol begins in
    WaveBits b;
    b.in = in + w0;
    b.lim = in_bytes - w0 < 0xFFFFFFF0ull ? (uint32_t)(in_bytes - w0) : 0xFFFFFFF0u;
    b.start((uint32_t)(hdr_bit - w0 * 8));
    // units so far and the most a distance may reach back, in 32 bits (a block gives at most 2^28 bytes)
    const uint32_t want32 = (uint32_t)want;
    const uint32_t back = blk == 0 ? 0u : (out0 < 32768u ? (uint32_t)out0 : 32768u);
    uint32_t st = INF_OK, nt = 0, tokv = 0, n = 0;
    for (;;) {
        b.fill();
        uint32_t e = uni(s_t.ll_lut[(uint32_t)b.buf & ((1u << kLlBits) - 1u)]);
        if (__builtin_expect(e != 0, 1)) {
            b.drop(e & 15u);
        } else {
            const int sy = wave_slow_sym(b, llc, s_t.ll_sym);
            if (sy < 0) {
                st = INF_BAD_SYMBOL;
                break;
            }
            e = ll_entry((uint32_t)sy, 0);
        }
        uint32_t t;
        if ((e >> 30) == 0) {
            t = (e >> 8) & 0xFFu;
            n += 1;
        } else if (__builtin_expect((e & kEntEob) != 0, 0)) {
            break;
        } else {
            const uint32_t xb = (e >> 4) & 15u;
            const uint32_t len = ((e >> 8) & 0x1FFu) + ((uint32_t)b.buf & ((1u << xb) - 1u));
            if (__builtin_expect(((e >> 8) & 0x1FFu) == 0, 0)) {
                st = INF_BAD_SYMBOL;
                break;
            }
            b.drop(xb);
            b.fill();
            uint32_t f = uni(s_t.d_lut[(uint32_t)b.buf & ((1u << kDBits) - 1u)]);
            if (__builtin_expect(f != 0, 1)) {
                b.drop(f & 15u);
            } else {
                const int ds = wave_slow_sym(b, dc, s_t.d_sym);
                if (ds < 0) {
                    st = INF_BAD_SYMBOL;
                    break;
                }
                f = d_entry((uint32_t)ds, 0);
            }
            const uint32_t yb = (f >> 4) & 15u;
            const uint32_t dist = (f >> 8) + ((uint32_t)b.buf & ((1u << yb) - 1u));
            if (__builtin_expect((f >> 8) == 0, 0)) {
                st = INF_BAD_SYMBOL;
                break;
            }
            b.drop(yb);
            if (__builtin_expect(dist > n + back, 0)) {  // before the start of the stream, or more than a window back
                st = INF_BAD_DISTANCE;
                break;
            }
            t = kTokMatch | ((len - 3u) << 15) | (dist - 1u);
            n += len;
        }
        tokv = lane == (nt & 63u) ? t : tokv;
        nt++;
        if ((nt & 63u) == 0) {
            // More than the block's share of the output?  Looked at once per group: a token gives a unit at least,
            // so the tokens stay within the block's room for them.
            if (__builtin_expect(n > want32, 0)) {
                st = INF_OVERFLOW;
                break;
            }
            tk[nt - 64u + lane] = tokv;
        }
    }
    if (st == INF_OK && n > want32) st = INF_OVERFLOW;
    if (st == INF_OK && (nt & 63u)) tk[(nt & ~63u) + lane] = tokv;  // the lanes behind the last token hold old ones: never read
    if (st == INF_OK && n != want32) st = INF_TRUNCATED;  // the block ended short of its share
    // where the block ended: the next block must begin there (the last one: the trailer, after padding)
    const uint64_t end_bit = w0 * 8 + b.bit_pos();
    if (st == INF_OK && blk + 1 < n_blocks && end_bit != bit_off[blk + 1] + bit_shift) st = INF_BAD_BLOCK;
    if (lane == 0) {
        if (blk + 1 == n_blocks) {  // behind the statuses: the bit at which the last block ended
            const uint64_t e = end_bit - bit_shift;
            status[n_blocks] = (uint32_t)e;
            status[n_blocks + 1] = (uint32_t)(e >> 32);
        }
        status[blk] = st;
        ntok[blockIdx.x] = st == INF_OK ? nt : 0u;
    }
}

// units of block b0 + blockIdx.x from its tokens; `units` holds the blocks of the pass, chunk entries each
__global__ __launch_bounds__(kInfLanes) void k_expand_tokens(const uint32_t *__restrict__ tok, uint64_t tok_stride,
                                                              const uint32_t *__restrict__ ntok, uint32_t b0,
                                                              uint64_t chunk, uint16_t *__restrict__ units) {
    __shared__ __attribute__((aligned(16))) uint16_t s_ring[kBlkRing];
    const uint32_t lane = threadIdx.x;
    const uint32_t nt = ntok[blockIdx.x];
    if (nt == 0) return;  // an empty block, or one that did not decode
    const uint32_t *const tk = tok + (uint64_t)blockIdx.x * tok_stride;
    uint16_t *const o = units + (uint64_t)blockIdx.x * chunk;
    // the window in front of the block: markers (unit -32768 + j lives in slot j).  The first block of the stream has
    // no window; its decoder has refused every distance that reaches in front of it.
    for (uint32_t j = lane; j < 32768; j += kInfLanes) s_ring[j] = (uint16_t)(kMark | j);
    __syncthreads();
    uint32_t pos = 0, flushed = 0;  // units so far; flushed is a multiple of 8
    uint32_t ahead = lane < nt ? tk[lane] : 0u;
    for (uint32_t t0 = 0; t0 < nt; t0 += kInfLanes) {
        const uint32_t t = ahead;
        if (t0 + kInfLanes < nt) ahead = t0 + kInfLanes + lane < nt ? tk[t0 + kInfLanes + lane] : 0u;  // the next group
        const bool valid = t0 + lane < nt;
        const bool is_m = valid && (t & kTokMatch);
        const uint32_t len = !valid ? 0u : is_m ? ((t >> 15) & 0xFFu) + 3u : 1u;
        uint32_t incl = len;
        for (uint32_t off = 1; off < kInfLanes; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        const uint32_t at = pos + incl - len;  // where this lane's token begins
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        uint64_t mm = __ballot(is_m);
        int prev = -1;
        while (mm) {
            const int m = __builtin_ctzll(mm);
            mm &= mm - 1;
            if (m > prev + 1 && (int)lane > prev && (int)lane < m) s_ring[at & (kBlkRing - 1)] = (uint16_t)(t & 0xFFu);
            const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)at, m);
            const uint32_t tm = (uint32_t)__builtin_amdgcn_readlane((int)t, m);
            const uint32_t dist = (tm & 0x7FFFu) + 1u, ml = ((tm >> 15) & 0xFFu) + 3u;
            // the match's units in parallel (a source that reaches into the match itself repeats with period `dist`)
            for (uint32_t k = lane; k < ml; k += kInfLanes) {
                uint32_t kk = k;
                if (kk >= dist) kk %= dist;
                const uint16_t v = s_ring[(a - dist + kk) & (kBlkRing - 1)];
                s_ring[(a + k) & (kBlkRing - 1)] = v;
            }
            prev = m;
        }
        if (valid && !is_m && (int)lane > prev) s_ring[at & (kBlkRing - 1)] = (uint16_t)(t & 0xFFu);
        pos += total;
        const bool last = t0 + kInfLanes >= nt;
        if (pos - flushed >= kBlkFlush || last) {
            __syncthreads();
            const uint32_t whole = pos & ~7u;  // eight units to a lane and store; what is left goes with the next flush
            for (uint32_t j = flushed + lane * 8; j < whole; j += kInfLanes * 8)
                *reinterpret_cast<uint4 *>(o + j) = *reinterpret_cast<const uint4 *>(&s_ring[j & (kBlkRing - 1)]);
            flushed = whole;
            if (last)  // ... or, at the end, unit by unit
                for (uint32_t j = whole + lane; j < pos; j += kInfLanes) o[j] = s_ring[j & (kBlkRing - 1)];
            __syncthreads();
        }
    }
}

// the window of block i + 1 = the last 32 KiB of block i, as bytes, for the blocks [b0, b1) of a pass.  One
// workgroup, blocks in order; the window of the block in hand stays in LDS (the first one's comes from the pass
// before).  win holds 32 KiB per block of the stream (block 0 has none); units holds the blocks of the pass.
__global__ __launch_bounds__(1024) void k_resolve_windows(const uint16_t *__restrict__ units, uint32_t b0, uint32_t b1,
                                                           uint32_t n_blocks, uint64_t chunk, uint8_t *__restrict__ win) {
    __shared__ uint8_t s_win[2][32768];
    const uint32_t tid = threadIdx.x;
    if (b0 > 0)
        for (uint32_t j = tid; j < 32768; j += 1024) s_win[b0 & 1][j] = win[(uint64_t)b0 * 32768 + j];
    __syncthreads();
    for (uint32_t i = b0; i < b1 && i + 1 < n_blocks; i++) {
        const uint16_t *tail = units + (uint64_t)(i - b0 + 1) * chunk - 32768;  // chunk >= 32768
        const uint8_t *prev = s_win[i & 1];
        uint8_t *cur = s_win[(i + 1) & 1];
        uint8_t *w = win + (uint64_t)(i + 1) * 32768;
        uint32_t u[32];
#pragma unroll
        for (uint32_t k = 0; k < 32; k++) u[k] = tail[tid + k * 1024];  // all loads in flight together
#pragma unroll
        for (uint32_t k = 0; k < 32; k++) {
            const uint32_t j = tid + k * 1024;
            const uint8_t v = (u[k] & kMark) ? prev[u[k] & 0x7FFFu] : (uint8_t)u[k];  // block 0 holds no markers
            cur[j] = v;
            w[j] = v;
        }
        __syncthreads();
    }
}

// bytes [first, first + n) of the output from the pass's units (first = the pass's first block * chunk)
__global__ __launch_bounds__(256) void k_resolve_all(const uint16_t *__restrict__ units, uint64_t first, uint64_t n,
                                                      uint64_t chunk, const uint8_t *__restrict__ win,
                                                      uint8_t *__restrict__ out) {
    const uint64_t q0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 8;  // inside the pass
    if (q0 >= n) return;
    const uint64_t p0 = first + q0;
    if (q0 + 8 <= n && (p0 % chunk) + 8 <= chunk) {
        const uint4 q = *reinterpret_cast<const uint4 *>(units + q0);  // q0 is a multiple of 8: 16-byte aligned
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
        for (uint64_t q = q0; q < q0 + 8 && q < n; q++) {
            const uint32_t u = units[q];
            out[first + q] = (u & kMark) ? win[((first + q) / chunk) * 32768 + (u & 0x7FFFu)] : (uint8_t)u;
        }
    }
}

// Blocks are decoded `pass_blocks` at a time: a pass takes pass_blocks * inflate_tok_stride(chunk) entries of tok,
// pass_blocks * chunk of units, pass_blocks of ntok; win 32 KiB and status one entry per block of the stream (+ 2).
uint64_t inflate_tok_stride(uint64_t chunk) { return (chunk + 64 + 63) & ~(uint64_t)63; }

// Two passes are in flight: while the tokens of pass p are expanded and resolved on stream s2, pass p + 1 is decoded
// on stream s (the decoder keeps the scalar units busy, the expander the vector units and the LDS).  tok, ntok and
// units hold two passes each (the second `pass` entries behind the first); ev: four events of the caller's.
hipError_t launch_inflate_blocks(const uint8_t *in, uint64_t in_bytes, const uint64_t *bit_off, uint32_t n_blocks,
                                 uint64_t chunk, uint64_t out_total, uint32_t pass_blocks, uint32_t *tok, uint32_t *ntok,
                                 uint16_t *units, uint8_t *win, uint8_t *out, uint32_t *status, hipStream_t s,
                                 hipStream_t s2, hipEvent_t *ev) {
    if (!n_blocks || !pass_blocks) return hipSuccess;
    // the decoder loads words: it starts from the aligned address below `in` and counts its bits from there
    const uint64_t mis = reinterpret_cast<uintptr_t>(in) & 3u;
    const uint64_t stride = inflate_tok_stride(chunk);
    hipEvent_t *const decoded = ev, *const resolved = ev + 2;
    hipError_t e = hipEventRecord(decoded[0], s);  // what the caller has queued on s comes first on s2 as well
    if (e == hipSuccess) e = hipStreamWaitEvent(s2, decoded[0], 0);
    uint32_t pass = 0;
    for (uint32_t b0 = 0; b0 < n_blocks && e == hipSuccess; b0 += pass_blocks, pass++) {
        const uint32_t b1 = n_blocks - b0 < pass_blocks ? n_blocks : b0 + pass_blocks;
        const uint64_t first = (uint64_t)b0 * chunk;
        const uint64_t n = ((uint64_t)b1 * chunk < out_total ? (uint64_t)b1 * chunk : out_total) - first;
        const uint32_t h = pass & 1u;  // which half of the buffers
        uint32_t *const tk = tok + (uint64_t)h * pass_blocks * stride, *const nt = ntok + (uint64_t)h * pass_blocks;
        uint16_t *const un = units + (uint64_t)h * pass_blocks * chunk;
        if (pass >= 2) e = hipStreamWaitEvent(s, resolved[h], 0);  // the pass before last is through with this half
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(k_decode_tokens, dim3(b1 - b0), dim3(kInfLanes), 0, s, in - mis, in_bytes + mis, bit_off, mis * 8,
                           b0, n_blocks, chunk, out_total, tk, stride, nt, status);
        e = hipEventRecord(decoded[h], s);
        if (e == hipSuccess) e = hipStreamWaitEvent(s2, decoded[h], 0);
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(k_expand_tokens, dim3(b1 - b0), dim3(kInfLanes), 0, s2, tk, stride, nt, b0, chunk, un);
        if (n_blocks > 1)  // (in block order: the passes follow one another on s2)
            hipLaunchKernelGGL(k_resolve_windows, dim3(1), dim3(1024), 0, s2, un, b0, b1, n_blocks, chunk, win);
        if (n)
            hipLaunchKernelGGL(k_resolve_all, dim3((uint32_t)((n + 2047) / 2048)), dim3(256), 0, s2, un, first, n, chunk,
                               win, out);
        e = hipEventRecord(resolved[h], s2);
    }
    // s goes on when everything on s2 is done
    if (e == hipSuccess && pass >= 1) e = hipStreamWaitEvent(s, resolved[(pass - 1) & 1u], 0);
    if (e == hipSuccess && pass >= 2) e = hipStreamWaitEvent(s, resolved[pass & 1u], 0);
    return e == hipSuccess ? hipGetLastError() : e;
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
