// Everything between the per-position match table and the packed bit stream:
// greedy parse resolution (k_parse_*), unit emission + literal/length/distance
// histograms (k_emit*), per-block code construction (k_huffman), and the
// prefix-sum / scatter-pack bit writer (k_tile_bits, k_offsets, k_encode).
#include "zg_huff_core.h"
#include "zg_internal.h"

namespace zg {

// ---------------------------------------------------------------------------
// Greedy parse resolution.  The parse of LZ77Compressor::compress (reference
// src/lz77.rs:156-204) is the chain pos -> pos + max(1, L(pos)) from the chunk
// start.  It is resolved without walking the chunk serially:
//   k_parse_spec   one lane per sub-tile of kParseSub positions walks the chain
//                  from the sub-tile's first position (a guess for every
//                  sub-tile but the first), flagging the positions it visits
//                  (one bit per position: sub-tile g of the batch owns the
//                  words [g * kParseSub / 32, (g + 1) * kParseSub / 32));
//   k_parse_fix    one lane per sub-tile takes the previous sub-tile's exit as
//                  its true entry, walks from there until it meets a flagged
//                  position (greedy parses re-synchronise within a few units),
//                  then swaps the flags of the guessed prefix for the true one;
//   k_parse_check  one wave per chunk: if some sub-tile never met its guess
//                  (long runs), everything after it is re-walked serially.
// The exits recorded by k_parse_fix are exact as long as every earlier
// sub-tile merged, which is what k_parse_check verifies.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t parse_step(const uint32_t *__restrict__ match, uint64_t idx) {
    const uint32_t len = match[idx] >> 16;
    return len ? len : 1u;
}

// k_parse_spec: a workgroup takes kSpecSubs consecutive sub-tiles.  Their match lengths are staged in LDS as one
// byte each (coalesced reads; 255 stands for "255 or more: look it up"), one lane per sub-tile then walks its
// chain through LDS, collecting the visited positions in a bit mask, and the workgroup writes the flags out as
// whole words.  A sub-tile sits kSpecStride bytes from the next so that the walkers spread over the LDS banks.
#ifndef ZG_SPEC_REGION
#define ZG_SPEC_REGION 16384
#endif
constexpr uint32_t kSpecRegion = ZG_SPEC_REGION;            // positions per workgroup
constexpr uint32_t kSpecSubs = kSpecRegion / kParseSub;
constexpr uint32_t kSpecStride = kParseSub + 4;
constexpr uint32_t kFlagWords = kParseSub / 32;  // flag words of a sub-tile
constexpr uint32_t kSpecWords = kFlagWords + 1;
constexpr uint32_t kSpecThreads = 256;

__global__ __launch_bounds__(kSpecThreads) void k_parse_spec(uint64_t first, uint64_t end, uint64_t chunk_bytes,
                                                              const uint32_t *__restrict__ clen, uint32_t c0,
                                                              uint32_t n_chunks, uint32_t spc,
                                                              const uint32_t *__restrict__ match, uint64_t base,
                                                              uint32_t *__restrict__ flagw,
                                                              uint32_t *__restrict__ sub_exit) {
    __shared__ __attribute__((aligned(16))) uint8_t s_len[kSpecSubs * kSpecStride];
    __shared__ uint32_t s_bits[kSpecSubs * kSpecWords];
    __shared__ uint64_t s_s0[kSpecSubs];  // first position of the sub-tile
    __shared__ uint32_t s_n[kSpecSubs];   // its positions (0: beyond the chunk's end)
    const uint32_t tid = threadIdx.x;
    const uint64_t gid0 = (uint64_t)blockIdx.x * kSpecSubs;
    if (tid < kSpecSubs) {
        const uint64_t gid = gid0 + tid;
        const uint32_t ci = (uint32_t)(gid / spc), j = (uint32_t)(gid % spc);
        uint64_t s0 = 0;
        uint32_t n = 0;
        if (ci < n_chunks) {
            const uint64_t cs = first + (uint64_t)(c0 + ci) * chunk_bytes;
            const uint64_t ce = chunk_end(cs, chunk_bytes, end, clen, c0 + ci);
            s0 = cs + (uint64_t)j * kParseSub;
            if (s0 < ce) n = ce - s0 < kParseSub ? (uint32_t)(ce - s0) : kParseSub;
        }
        s_s0[tid] = s0;
        s_n[tid] = n;
    }
    for (uint32_t i = tid; i < kSpecSubs * kSpecWords; i += kSpecThreads) s_bits[i] = 0;
    __syncthreads();
    // four rounds of loads are in flight before the first is used: one round at a time, the loop ran at the
    // pace of the memory latency
    constexpr uint32_t kStageUnroll = 4;
    for (uint32_t idx0 = tid * 4; idx0 < kSpecRegion; idx0 += kSpecThreads * 4 * kStageUnroll) {
        uint4 q[kStageUnroll];
        uint32_t fast = 0;
#pragma unroll
        for (uint32_t u = 0; u < kStageUnroll; u++) {
            const uint32_t idx = idx0 + u * kSpecThreads * 4;
            const uint32_t sub = idx / kParseSub, off = idx % kParseSub, n = s_n[sub];
            const uint32_t *m = match + (s_s0[sub] + off - base);
            q[u] = make_uint4(0, 0, 0, 0);
            if (off + 4 <= n && (reinterpret_cast<uintptr_t>(m) & 15) == 0) {
                q[u] = *reinterpret_cast<const uint4 *>(m);
                fast |= 1u << u;
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < kStageUnroll; u++) {
            const uint32_t idx = idx0 + u * kSpecThreads * 4;
            const uint32_t sub = idx / kParseSub, off = idx % kParseSub, n = s_n[sub];
            if (off >= n) continue;
            uint32_t pack = 0;
            if (fast & (1u << u)) {
                const uint32_t l[4] = {q[u].x >> 16, q[u].y >> 16, q[u].z >> 16, q[u].w >> 16};
#pragma unroll
                for (uint32_t i = 0; i < 4; i++) pack |= (l[i] < 255u ? l[i] : 255u) << (8 * i);
            } else {
                const uint32_t *m = match + (s_s0[sub] + off - base);
                for (uint32_t i = 0; i < 4; i++) {
                    if (off + i < n) {
                        const uint32_t len = m[i] >> 16;
                        pack |= (len < 255u ? len : 255u) << (8 * i);
                    }
                }
            }
            *reinterpret_cast<uint32_t *>(&s_len[sub * kSpecStride + off]) = pack;
        }
    }
    __syncthreads();
    if (tid < kSpecSubs) {
        const uint64_t gid = gid0 + tid;
        const uint32_t ci = (uint32_t)(gid / spc), j = (uint32_t)(gid % spc);
        const uint32_t n = s_n[tid];
        const uint64_t s0 = s_s0[tid];
        const uint8_t *len_t = s_len + tid * kSpecStride;
        uint32_t p = 0, wi = 0, word = 0;
        while (p < n) {
            const uint32_t w = p >> 5;
            if (w != wi) {
                s_bits[tid * kSpecWords + wi] = word;
                word = 0;
                wi = w;
            }
            word |= 1u << (p & 31u);
            uint32_t len = len_t[p];
            if (len == 255u) len = match[s0 + p - base] >> 16;
            p += len ? len : 1u;
        }
        if (n) s_bits[tid * kSpecWords + wi] = word;
        if (ci < n_chunks && (n != 0 || j == 0)) {
            const uint64_t cs = first + (uint64_t)(c0 + ci) * chunk_bytes;
            sub_exit[gid] = (uint32_t)((n ? s0 : cs) + p - cs);
        }
    }
    __syncthreads();
    // the flags leave as they are, a bit per position: the workgroup's sub-tiles own one run of words
    const uint64_t n_sub = (uint64_t)n_chunks * spc;
    for (uint32_t i = tid; i < kSpecSubs * kFlagWords; i += kSpecThreads) {
        const uint32_t sub = i / kFlagWords, w = i % kFlagWords;
        if (gid0 + sub < n_sub) flagw[(gid0 + sub) * kFlagWords + w] = s_bits[sub * kSpecWords + w];
    }
}

__global__ __launch_bounds__(64) void k_parse_fix(uint64_t first, uint64_t end, uint64_t chunk_bytes,
        const uint32_t *__restrict__ clen, uint32_t c0,
                                                   uint32_t n_chunks, uint32_t spc,
                                                   const uint32_t *__restrict__ match, uint64_t base,
                                                   uint32_t *__restrict__ flagw, const uint32_t *__restrict__ sub_exit,
                                                   uint32_t *__restrict__ sub_exit2, uint8_t *__restrict__ merged) {
    // the flag words of the workgroup's 64 sub-tiles, worked on in LDS (a lane touches its own sub-tile's only)
    __shared__ uint32_t s_f[64 * kSpecWords];
    const uint64_t gid = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    const uint64_t n_sub = (uint64_t)n_chunks * spc;
    for (uint32_t i = threadIdx.x; i < 64 * kFlagWords; i += 64) {
        const uint64_t g = (uint64_t)blockIdx.x * 64 + i / kFlagWords;
        s_f[(i / kFlagWords) * kSpecWords + i % kFlagWords] = g < n_sub ? flagw[g * kFlagWords + i % kFlagWords] : 0u;
    }
    __syncthreads();
    uint32_t *const mine = s_f + threadIdx.x * kSpecWords;
    const uint32_t ci = (uint32_t)(gid / spc), j = (uint32_t)(gid % spc);
    [&] {
        if (ci >= n_chunks) return;
        const uint64_t cs = first + (uint64_t)(c0 + ci) * chunk_bytes;
        const uint64_t ce = chunk_end(cs, chunk_bytes, end, clen, c0 + ci);
        const uint64_t s0 = cs + (uint64_t)j * kParseSub;
        const uint64_t k = gid;
        if (j == 0) {
            sub_exit2[k] = sub_exit[k];
            merged[k] = 1;
            return;
        }
        if (s0 >= ce) {
            merged[k] = 1;
            return;
        }
        const uint64_t s1 = (s0 + kParseSub < ce) ? s0 + kParseSub : ce;
        const uint64_t e = cs + sub_exit[k - 1];  // true entry if every earlier sub-tile merged
        if (e == s0) {
            sub_exit2[k] = sub_exit[k];
            merged[k] = 1;
            return;
        }
        // where does the true chain meet the guessed one?
        uint64_t t = e;
        bool met = false;
        while (t < s1) {
            const uint32_t off = (uint32_t)(t - s0);
            if ((mine[off >> 5] >> (off & 31u)) & 1u) {
                met = true;
                break;
            }
            t += parse_step(match, t - base);
        }
        const uint64_t stop = met ? t : s1;  // flags from here on are already right (or there are none)
        for (uint64_t q = s0; q < stop;) {   // drop the guessed prefix
            const uint32_t off = (uint32_t)(q - s0);
            mine[off >> 5] &= ~(1u << (off & 31u));
            q += parse_step(match, q - base);
        }
        for (uint64_t u = e; u < stop;) {    // put the true prefix
            const uint32_t off = (uint32_t)(u - s0);
            mine[off >> 5] |= 1u << (off & 31u);
            u += parse_step(match, u - base);
        }
        sub_exit2[k] = met ? sub_exit[k] : (uint32_t)(t - cs);
        merged[k] = met ? 1 : 0;
    }();
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 64 * kFlagWords; i += 64) {
        const uint64_t g = (uint64_t)blockIdx.x * 64 + i / kFlagWords;
        if (g < n_sub) flagw[g * kFlagWords + i % kFlagWords] = s_f[(i / kFlagWords) * kSpecWords + i % kFlagWords];
    }
}

__global__ __launch_bounds__(64) void k_parse_check(uint64_t first, uint64_t end, uint64_t chunk_bytes,
        const uint32_t *__restrict__ clen, uint32_t c0,
                                                     uint32_t spc, const uint32_t *__restrict__ match, uint64_t base,
                                                     uint32_t *__restrict__ flagw, const uint32_t *__restrict__ sub_exit2,
                                                     const uint8_t *__restrict__ merged) {
    const uint32_t ci = blockIdx.x, lane = threadIdx.x;
    const uint64_t cs = first + (uint64_t)(c0 + ci) * chunk_bytes;
    const uint64_t ce = chunk_end(cs, chunk_bytes, end, clen, c0 + ci);
    const uint32_t nsub = (uint32_t)((ce - cs + kParseSub - 1) / kParseSub);
    uint32_t bad = 0xFFFFFFFFu;  // first sub-tile that never met its guess
    for (uint32_t j0 = 0; j0 < nsub && bad == 0xFFFFFFFFu; j0 += 64) {
        const uint32_t j = j0 + lane;
        const bool nm = j < nsub && !merged[(uint64_t)ci * spc + j];
        const uint64_t b = __ballot(nm);
        if (b) bad = j0 + (uint32_t)(__ffsll((unsigned long long)b) - 1);
    }
    if (bad == 0xFFFFFFFFu || bad + 1 >= nsub) return;
    // everything after sub-tile `bad` was resolved from a wrong entry: redo it serially
    uint32_t *const fw = flagw + (uint64_t)ci * spc * kFlagWords;  // the chunk's flags: bit r = position cs + r
    for (uint32_t w = (bad + 1) * kFlagWords + lane; w < nsub * kFlagWords; w += 64) fw[w] = 0;
    __syncthreads();
    if (lane == 0) {
        uint64_t p = cs + sub_exit2[(uint64_t)ci * spc + bad];
        // (a sub-tile's exit lies past the sub-tile: only cleared words are written)
        uint32_t wi = 0xFFFFFFFFu, word = 0;
        while (p < ce) {
            const uint32_t r = (uint32_t)(p - cs);
            if ((r >> 5) != wi) {
                if (wi != 0xFFFFFFFFu) fw[wi] = word;
                wi = r >> 5;
                word = 0;
            }
            word |= 1u << (r & 31u);
            p += parse_step(match, p - base);
        }
        if (wi != 0xFFFFFFFFu) fw[wi] = word;
    }
}

// ---------------------------------------------------------------------------
// Unit emission: flagged positions become packed LZ77 units, in order, and
// are counted as HuffmanTree::from_lz77 does (src/huffman.rs:243-264).
// k_emit_count sums the flags per tile, k_emit_scan turns the sums into
// offsets per chunk, k_emit writes the units and the histograms.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kEmitThreads) void k_emit_count(uint32_t spc, uint32_t tpc,
                                                              const uint32_t *__restrict__ flagw,
                                                              uint32_t *__restrict__ tile_cnt) {
    __shared__ uint32_t s_red[kEmitThreads / 64];
    const uint32_t ci = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    // the tile's flags: kEmitTile / 32 words, zero past the chunk's end (and absent past the chunk's last sub-tile)
    const uint32_t w0 = tile * (kEmitTile / 32), wn = spc * kFlagWords;
    const uint32_t *fw = flagw + (uint64_t)ci * wn;
    uint32_t cnt = 0;
    for (uint32_t i = tid; i < kEmitTile / 32; i += kEmitThreads)
        if (w0 + i < wn) cnt += (uint32_t)__builtin_popcount(fw[w0 + i]);
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
    if ((tid & 63) == 0) s_red[tid >> 6] = cnt;
    __syncthreads();
    if (tid == 0) {
        uint32_t t = 0;
        for (uint32_t i = 0; i < kEmitThreads / 64; i++) t += s_red[i];
        tile_cnt[(uint64_t)ci * tpc + tile] = t;
    }
}

constexpr uint32_t kScanFew = 16;  // chunks of up to this many tiles are scanned by a thread each
__global__ __launch_bounds__(256) void k_emit_scan(uint32_t tpc, uint32_t *__restrict__ tile_cnt,
                                                    uint32_t *__restrict__ ntok) {
    __shared__ uint32_t s_part[256];
    const uint32_t ci = blockIdx.x, tid = threadIdx.x;
    uint32_t *tc = tile_cnt + (uint64_t)ci * tpc;
    const uint32_t per = (tpc + 255) / 256;
    const uint32_t lo = tid * per, hi = (lo + per < tpc) ? lo + per : tpc;
    uint32_t s = 0;
    for (uint32_t i = lo; i < hi; i++) s += tc[i];
    s_part[tid] = s;
    __syncthreads();
    for (uint32_t off = 1; off < 256; off <<= 1) {
        const uint32_t v = tid >= off ? s_part[tid - off] : 0u;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    uint32_t run = s_part[tid] - s;
    for (uint32_t i = lo; i < hi; i++) {
        const uint32_t v = tc[i];
        tc[i] = run;
        run += v;
    }
    if (tid == 255) ntok[ci] = s_part[255];
}

// the same for chunks of a few tiles (records): a thread per chunk instead of a workgroup
__global__ __launch_bounds__(256) void k_emit_scan_few(uint32_t n_chunks, uint32_t tpc, uint32_t *__restrict__ tile_cnt,
                                                        uint32_t *__restrict__ ntok) {
    const uint32_t ci = blockIdx.x * 256 + threadIdx.x;
    if (ci >= n_chunks) return;
    uint32_t *tc = tile_cnt + (uint64_t)ci * tpc;
    uint32_t run = 0;
    for (uint32_t i = 0; i < tpc; i++) {
        const uint32_t v = tc[i];
        tc[i] = run;
        run += v;
    }
    ntok[ci] = run;
}

__global__ __launch_bounds__(kEmitThreads) void k_emit(const uint8_t *__restrict__ data, uint64_t first, uint64_t end,
                                                        uint64_t chunk_bytes, const uint32_t *__restrict__ clen,
                                                        uint32_t c0, uint32_t spc, uint32_t tpc,
                                                        const uint32_t *__restrict__ flagw,
                                                        const uint32_t *__restrict__ match, uint64_t base,
                                                        const uint32_t *__restrict__ tile_off,
                                                        uint32_t *__restrict__ ntok_one,
                                                        uint32_t *__restrict__ tokens, uint64_t tok_base,
                                                        uint32_t *__restrict__ freq, uint32_t *__restrict__ status) {
    // ntok_one != nullptr: chunks of one tile (tpc == 1).  The tile's units begin the chunk's and their number is
    // the chunk's, written here: no counting pass ran and tile_off is not read.
    __shared__ uint32_t s_hist[kTabStride];
    __shared__ uint32_t s_wave[kEmitThreads / 64];
    __shared__ uint32_t s_tok[kEmitTile];
    const uint32_t ci = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    const uint64_t cs = first + (uint64_t)(c0 + ci) * chunk_bytes;
    const uint64_t ce = chunk_end(cs, chunk_bytes, end, clen, c0 + ci);
    const uint64_t t0 = cs + (uint64_t)tile * kEmitTile;
    if (t0 >= ce) {
        if (ntok_one && tid == 0) ntok_one[ci] = 0;
        return;
    }
    const uint32_t tn = (uint32_t)((ce - t0 < kEmitTile) ? ce - t0 : kEmitTile);
    for (uint32_t i = tid; i < kTabStride; i += kEmitThreads) s_hist[i] = 0;
    constexpr uint32_t kPer = kEmitTile / kEmitThreads;  // consecutive positions per thread
    const uint32_t b0 = tid * kPer;
    static_assert(kPer == 16, "a thread takes half a flag word");
    // flags of my positions as a bit mask (bits past the chunk's end are zero)
    const uint32_t wi = (tile * kEmitTile + b0) >> 5;
    const uint32_t fl = wi < spc * kFlagWords ? (flagw[(uint64_t)ci * spc * kFlagWords + wi] >> (b0 & 16u)) & 0xFFFFu : 0u;
    const uint32_t cnt = (uint32_t)__builtin_popcount(fl);
    // The results and bytes of all my positions, flagged or not, by whole-line loads issued together: nearly every
    // line holds a flagged position, and a load per flagged position, each waited for before the next (and one more
    // for a literal's byte), kept a thread waiting for memory several times over.
    const uint32_t *mt = match + (t0 - base) + b0;
    const uint8_t *dt = data + t0 + b0;
    uint32_t mv[kPer], dw[kPer / 4];
    if (b0 + kPer <= tn && (reinterpret_cast<uintptr_t>(mt) & 15) == 0) {
#pragma unroll
        for (uint32_t k = 0; k < kPer / 4; k++) {
            const uint4 q = reinterpret_cast<const uint4 *>(mt)[k];
            mv[4 * k] = q.x;
            mv[4 * k + 1] = q.y;
            mv[4 * k + 2] = q.z;
            mv[4 * k + 3] = q.w;
        }
        __builtin_memcpy(dw, dt, kPer);
    } else {
#pragma unroll
        for (uint32_t i = 0; i < kPer; i++) mv[i] = b0 + i < tn ? mt[i] : 0u;
#pragma unroll
        for (uint32_t k = 0; k < kPer / 4; k++) {
            dw[k] = 0;
#pragma unroll
            for (uint32_t i = 0; i < 4; i++)
                if (b0 + 4 * k + i < tn) dw[k] |= (uint32_t)dt[4 * k + i] << (8 * i);
        }
    }
    // where my units go: a scan inside every wave, then over the wave totals
    uint32_t incl = cnt;
    for (uint32_t off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off);
        if ((tid & 63u) >= off) incl += up;
    }
    if ((tid & 63u) == 63u) s_wave[tid >> 6] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < kEmitThreads / 64; w++) {
        const uint32_t t = s_wave[w];
        before += w < (tid >> 6) ? t : 0u;
        total += t;
    }
    // the units of the tile are collected in LDS and leave in one coalesced copy
    uint32_t o = before + incl - cnt;
#pragma unroll
    for (uint32_t i = 0; i < kPer; i++) {
        if (!((fl >> i) & 1u)) continue;
        uint32_t m = mv[i];
        if (m == kUnvisited) {  // the parse stands where no walker searched: the caller redoes the call
            atomicOr(status, 4u);
            m = 0;
        }
        uint32_t tok;
        if (m) {
            const uint32_t len = m >> 16, dist = m & 0xFFFFu;
            tok = ZG_TOK_MARKER | (len << 16) | dist;
            atomicAdd(&s_hist[257u + length_code_index(len)], 1u);
            atomicAdd(&s_hist[kDOff + dist_code(dist)], 1u);
        } else {
            tok = (dw[i >> 2] >> ((i & 3u) * 8u)) & 0xFFu;
            atomicAdd(&s_hist[tok], 1u);
        }
        s_tok[o++] = tok;
    }
    __syncthreads();
    uint32_t *tok_out = tokens + (cs - tok_base) + (ntok_one ? 0u : tile_off[(uint64_t)ci * tpc + tile]);
    if (ntok_one && tid == 0) ntok_one[ci] = total;
    for (uint32_t i = tid; i < total; i += kEmitThreads) tok_out[i] = s_tok[i];
    for (uint32_t i = tid; i < kTabStride; i += kEmitThreads)
        if (s_hist[i]) atomicAdd(&freq[(uint64_t)ci * kTabStride + i], s_hist[i]);
}

hipError_t launch_parse(const ChunkGrid &g, uint32_t c0, uint32_t c1, const uint32_t *match, uint64_t base,
                        uint32_t *flag, uint32_t *sub_exit, uint32_t *sub_exit2, uint8_t *merged,
                        uint32_t *tile_cnt, uint32_t *tokens, uint64_t tok_base, uint32_t *ntok, uint32_t *freq,
                        uint32_t *status, hipStream_t s) {
    if (c1 <= c0) return hipSuccess;
    const uint32_t nc = c1 - c0;
    const uint32_t spc = parse_subs_per_chunk(g.chunk_bytes);
    const uint32_t tpc = emit_tiles_per_chunk(g.chunk_bytes);
    // flag: parse_flag_bytes() bytes; k_parse_spec writes the word of every sub-tile of every chunk, set or not
    hipError_t e = hipMemsetAsync(freq, 0, (size_t)nc * kTabStride * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    const uint64_t lanes = (uint64_t)nc * spc;
    const uint32_t blocks = (uint32_t)((lanes + 63) / 64);
    hipLaunchKernelGGL(k_parse_spec, dim3((uint32_t)((lanes + kSpecSubs - 1) / kSpecSubs)), dim3(kSpecThreads), 0, s,
                       g.first, g.end, g.chunk_bytes, g.chunk_len, c0, nc, spc, match, base, flag, sub_exit);
    hipLaunchKernelGGL(k_parse_fix, dim3(blocks), dim3(64), 0, s, g.first, g.end, g.chunk_bytes, g.chunk_len, c0, nc, spc,
                       match, base, flag, sub_exit, sub_exit2, merged);
    hipLaunchKernelGGL(k_parse_check, dim3(nc), dim3(64), 0, s, g.first, g.end, g.chunk_bytes, g.chunk_len, c0, spc, match,
                       base, flag, sub_exit2, merged);
    if (tpc > 1) {
        hipLaunchKernelGGL(k_emit_count, dim3(tpc, nc), dim3(kEmitThreads), 0, s, spc, tpc, flag, tile_cnt);
        if (tpc <= kScanFew)
            hipLaunchKernelGGL(k_emit_scan_few, dim3((nc + 255) / 256), dim3(256), 0, s, nc, tpc, tile_cnt, ntok);
        else
            hipLaunchKernelGGL(k_emit_scan, dim3(nc), dim3(256), 0, s, tpc, tile_cnt, ntok);
    }
    hipLaunchKernelGGL(k_emit, dim3(tpc, nc), dim3(kEmitThreads), 0, s, g.data, g.first, g.end, g.chunk_bytes, g.chunk_len,
                       c0, spc, tpc, flag, match, base, tile_cnt, tpc > 1 ? nullptr : ntok, tokens, tok_base, freq, status);
    return hipGetLastError();
}

// Histogram of an existing token stream (used by zg_huffman_from_tokens).
__global__ __launch_bounds__(256) void k_histogram(const uint32_t *__restrict__ tokens, uint64_t n,
                                                    uint32_t *__restrict__ freq) {
    __shared__ uint32_t s_hist[kTabStride];
    for (uint32_t i = threadIdx.x; i < kTabStride; i += 256) s_hist[i] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint32_t t = tokens[i];
        if (ZG_TOK_IS_MARKER(t)) {
            atomicAdd(&s_hist[257u + length_code_index(ZG_TOK_LEN(t))], 1u);
            atomicAdd(&s_hist[kDOff + dist_code(ZG_TOK_DIST(t))], 1u);
        } else {
            atomicAdd(&s_hist[t & 0xFFu], 1u);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < kTabStride; i += 256)
        if (s_hist[i]) atomicAdd(&freq[i], s_hist[i]);
}
__global__ void k_histogram_finish(uint32_t *freq) {
    if (threadIdx.x == 0 && freq[256] == 0) freq[256] = 1;
}
hipError_t launch_histogram(const uint32_t *tokens, uint64_t n, uint32_t *freq, hipStream_t s) {
    hipError_t e = hipMemsetAsync(freq, 0, kTabStride * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    if (n) {
        uint32_t blocks = (uint32_t)((n + 255) / 256);
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(k_histogram, dim3(blocks), dim3(256), 0, s, tokens, n, freq);
    }
    hipLaunchKernelGGL(k_histogram_finish, dim3(1), dim3(64), 0, s, freq);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// k_huffman: one wave per chunk builds the block's code tables and header.
// build_block_wave is build_block (zg_huff_core.h) with the wide steps spread
// over the 64 lanes -- sorting the symbols by count (rank sort), leaf depths
// (every leaf climbs to the root), canonical code assignment (rank among the
// symbols of equal length), header bits (prefix sum + OR) -- and the inherently
// serial ones (two-queue merge, run-length coding of the lengths, the 19-symbol
// code-length code) left to lane 0.  Same results bit for bit.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ uint32_t wave_max(uint32_t v) {
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = __shfl_xor(v, off);
        v = o > v ? o : v;
    }
    return v;
}

// lengths of one alphabet; all lanes call it, `len` and `w` live in LDS
__device__ void wave_huff_lengths(const uint32_t *freq, int n, int max_bits, uint8_t *len, HuffWork &w, uint32_t lane) {
    // used symbols in ascending (count, symbol) order.  They are first gathered in symbol order (count in
    // w.weight, symbol in w.parent: both are rewritten by the merge afterwards); the rank of one is then the number
    // of gathered entries below it, a loop over the used symbols only.
    uint32_t mu = 0;  // uniform: entries gathered so far
    for (int b = 0; b < n; b += 64) {
        const int i = b + (int)lane;
        const uint32_t f = i < n ? freq[i] : 0u;
        if (i < n) len[i] = 0;
        const uint64_t bal = __ballot(f != 0u);
        if (f) {
            const uint32_t at = mu + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            w.weight[at] = f;
            w.parent[at] = (uint16_t)i;
        }
        mu += (uint32_t)__popcll(bal);
    }
    const int m = (int)mu;
    __syncthreads();
    uint32_t rk[(kHuffMaxSyms + 63) / 64], sy[(kHuffMaxSyms + 63) / 64];
#pragma unroll
    for (int r = 0; r < (int)((kHuffMaxSyms + 63) / 64); r++) {
        const int k = r * 64 + (int)lane;
        rk[r] = 0;
        sy[r] = 0;
        if (k < m) {
            const uint32_t f = w.weight[k];
            uint32_t rank = 0;
            for (int j = 0; j < m; j++) {
                const uint32_t g = w.weight[j];
                rank += (g < f) | ((g == f) & (j < k));
            }
            rk[r] = rank;
            sy[r] = w.parent[k];
        }
    }
    __syncthreads();  // the gathered entries have been read by everybody
#pragma unroll
    for (int r = 0; r < (int)((kHuffMaxSyms + 63) / 64); r++)
        if (r * 64 + (int)lane < m) w.sorted[rk[r]] = (uint16_t)sy[r];
    __syncthreads();
    if (m == 0) return;
    if (m == 1) {  // lone symbol: one bit (RFC 1951 3.2.7); the reference panics here
        if (lane == 0) len[w.sorted[0]] = 1;
        __syncthreads();
        return;
    }
    // the weights in sorted order, by all lanes (one lane loading them one by one waited a memory latency each)
    for (int i = (int)lane; i < m; i += 64) w.weight[i] = freq[w.sorted[i]];
    __syncthreads();
    // (the merge on one lane: run by the whole wave in step, its state in scalar registers, it was slower -- a CU has
    // one scalar unit for its eighteen waves of this kernel and four vector units)
    if (lane == 0) huff_merge_loaded(m, w);
    __syncthreads();
    const int root = 2 * m - 2;
    uint32_t mx = 0;
    for (int i = (int)lane; i < m; i += 64) {
        uint32_t d = 0;
        for (int node = i; node != root; node = w.parent[node]) d++;
        w.weight[i] = d;  // huff_limit reads the leaf depths from here
        len[w.sorted[i]] = (uint8_t)(d > 255u ? 255u : d);
        mx = d > mx ? d : mx;
    }
    mx = wave_max(mx);
    __syncthreads();
    if ((int)mx > max_bits) {
        if (lane == 0) huff_limit(m, max_bits, len, w);
        __syncthreads();
    }
}

// canonical codes of one alphabet (huff_codes); cnt[17] and nxt[17] are LDS scratch
__device__ void wave_huff_codes(const uint8_t *len, int n, uint32_t *tab, uint32_t *cnt, uint32_t *nxt, uint32_t lane) {
    if (lane < 17) cnt[lane] = 0;
    __syncthreads();
    for (int i = (int)lane; i < n; i += 64)
        if (len[i]) atomicAdd(&cnt[len[i]], 1u);
    __syncthreads();
    if (lane == 0) {
        uint32_t code = 0;
        nxt[0] = 0;
        for (int bits = 1; bits <= 16; bits++) {
            code = (code + (bits > 1 ? cnt[bits - 1] : 0u)) << 1;
            nxt[bits] = code;
        }
    }
    __syncthreads();
    uint32_t seen[17];  // uniform: symbols of each length in the groups of 64 gone by
#pragma unroll
    for (int L = 0; L < 17; L++) seen[L] = 0;
    for (int b = 0; b < n; b += 64) {
        const int i = b + (int)lane;
        const uint32_t l = i < n ? len[i] : 0u;
        uint32_t rank = 0;  // symbols of the same length before this one
#pragma unroll
        for (uint32_t L = 1; L <= 16; L++) {
            const uint64_t bal = __ballot(l == L);
            if (l == L) rank = seen[L] + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            seen[L] += (uint32_t)__popcll(bal);
        }
        if (i < n) tab[i] = l ? (bit_reverse(nxt[l] + rank, l) | (l << 16)) : 0u;
    }
    __syncthreads();
}

// run-length items of one vector of code lengths (rle_lengths) with all lanes: every lane that stands on the
// first element of a run codes that run (rle_run, first to count its items, then at the place a prefix sum over
// the runs gives it).  n <= 320; returns the index behind the last item (uniform).
__device__ __forceinline__ int wave_rle_lengths(const uint8_t *cl, int n, uint8_t *rsym, uint8_t *rextra, int out, uint32_t lane) {
    constexpr int kGroups = 5;
    uint64_t starts[kGroups];
#pragma unroll
    for (int g = 0; g < kGroups; g++) {
        const int i = g * 64 + (int)lane;
        starts[g] = __ballot(i < n && (i == 0 || cl[i] != cl[i - 1]));
    }
#pragma unroll
    for (int g = 0; g < kGroups; g++) {
        const int i = g * 64 + (int)lane;
        const bool st = (starts[g] >> lane) & 1ull;
        int items = 0, count = 0;
        uint8_t v = 0;
        if (st) {
            // the run ends where the next one starts: above this lane in the same group, else in a later group
            int nxt_start = n;
            const uint64_t above = lane < 63 ? starts[g] & ~((2ull << lane) - 1ull) : 0ull;
            if (above) {
                nxt_start = g * 64 + __ffsll((unsigned long long)above) - 1;
            } else {
#pragma unroll
                for (int h = kGroups - 1; h > g; h--)
                    if (starts[h]) nxt_start = h * 64 + __ffsll((unsigned long long)starts[h]) - 1;
            }
            count = nxt_start - i;
            v = cl[i];
            items = rle_run<false>(v, count, rsym, rextra, 0);
        }
        uint32_t incl = (uint32_t)items;  // inclusive scan over the wave
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off);
            if ((int)lane >= off) incl += o;
        }
        if (st) rle_run<true>(v, count, rsym, rextra, out + (int)incl - items);
        out += (int)__shfl(incl, 63);
    }
    return out;
}

__device__ uint32_t build_block_wave(const uint32_t *freq_ll, const uint32_t *freq_d, int strategy, bool bfinal,
                                     BlockTables &tab, uint32_t *hdr, uint64_t *payload_bits, HeaderWork &hw,
                                     uint32_t *cnt, uint32_t *nxt, uint32_t *shared2, uint32_t lane) {
    uint32_t hbits = 0;
    if (strategy == ZG_FIXED) {  // get_zlib_fixed, src/huffman.rs:353-367
        for (int i = (int)lane; i < 288; i += 64) hw.ll_len[i] = (uint8_t)(i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8);
        if (lane < 30) hw.d_len[lane] = 5;
        __syncthreads();
        wave_huff_codes(hw.ll_len, 288, tab.ll, cnt, nxt, lane);
        wave_huff_codes(hw.d_len, 30, tab.d, cnt, nxt, lane);
        if (lane == 0) {
            BitSink sink{hdr, 0};
            sink.put(bfinal ? 1u : 0u, 1);
            sink.put(1, 2);
        }
        hbits = 3;
    } else {
        wave_huff_lengths(freq_ll, 286, 15, hw.ll_len, hw.hw, lane);
        wave_huff_lengths(freq_d, 30, 15, hw.d_len, hw.hw, lane);
        wave_huff_codes(hw.ll_len, 286, tab.ll, cnt, nxt, lane);
        wave_huff_codes(hw.d_len, 30, tab.d, cnt, nxt, lane);
        // get_codelengths, src/compress.rs:376-407
        uint32_t lm = 257, dm = 0, dany = 0;
        for (int i = 258 + (int)lane; i < 286; i += 64)
            if (hw.ll_len[i]) lm = (uint32_t)i;
        if (lane < 30 && hw.d_len[lane]) {
            dm = lane;
            dany = 1;
        }
        const int lmax = (int)wave_max(lm);
        const int dmax = wave_max(dany) ? (int)wave_max(dm) : 1;
        const int nl = lmax + 1, nd = dmax + 1;
        if (lane < 19) hw.cl_freq[lane] = 0;
        int nr = wave_rle_lengths(hw.ll_len, nl, hw.rsym, hw.rextra, 0, lane);  // the two vectors are coded separately
        nr = wave_rle_lengths(hw.d_len, nd, hw.rsym, hw.rextra, nr, lane);
        __syncthreads();
        for (int i = (int)lane; i < nr; i += 64) atomicAdd(&hw.cl_freq[hw.rsym[i]], 1u);
        __syncthreads();
        wave_huff_lengths(hw.cl_freq, 19, 7, hw.cl_len, hw.hw, lane);  // get_code_tree
        wave_huff_codes(hw.cl_len, 19, hw.cl_tab, cnt, nxt, lane);
        if (lane == 0) {
            const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            int nh = 19;  // get_hcodes: trailing zeros trimmed
            while (nh > 4 && hw.cl_len[order[nh - 1]] == 0) nh--;
            BitSink sink{hdr, 0};
            sink.put(bfinal ? 1u : 0u, 1);
            sink.put(2, 2);
            sink.put((uint32_t)(nl - 257), 5);
            sink.put((uint32_t)(nd - 1), 5);
            sink.put((uint32_t)(nh - 4), 4);
            for (int i = 0; i < nh; i++) sink.put(hw.cl_len[order[i]], 3);
            shared2[1] = sink.nbits;
        }
        __syncthreads();
        // the run-length items: every lane codes one, a running prefix sum places them
        uint32_t at = shared2[1];
        for (int i0 = 0; i0 < nr; i0 += 64) {
            const int i = i0 + (int)lane;
            uint32_t val = 0, nb = 0;
            if (i < nr) {
                const uint32_t sy = hw.rsym[i], t = hw.cl_tab[sy];
                val = t & 0xFFFFu;
                nb = t >> 16;
                if (sy == 16) {
                    val |= ((uint32_t)hw.rextra[i] - 3u) << nb;
                    nb += 2;
                } else if (sy == 17) {
                    val |= ((uint32_t)hw.rextra[i] - 3u) << nb;
                    nb += 3;
                } else if (sy == 18) {
                    val |= ((uint32_t)hw.rextra[i] - 11u) << nb;
                    nb += 7;
                }
            }
            uint32_t incl = nb;  // inclusive scan over the wave
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t o = __shfl_up(incl, off);
                if ((int)lane >= off) incl += o;
            }
            if (nb) {
                const uint32_t pos = at + incl - nb, wd = pos >> 5, sh = pos & 31u;
                atomicOr(&hdr[wd], val << sh);
                if (sh + nb > 32) atomicOr(&hdr[wd + 1], val >> (32u - sh));
            }
            at += __shfl(incl, 63);
        }
        hbits = at;
        __syncthreads();
    }
    // bits of all units plus the end-of-block symbol
    uint64_t bits = 0;
    for (int i = (int)lane; i < 286; i += 64) {
        const uint32_t f = freq_ll[i];
        if (!f) continue;
        const uint32_t extra = i > 256 ? length_extra_bits((uint32_t)i - 257u) : 0u;
        bits += (uint64_t)f * ((tab.ll[i] >> 16) + extra);
    }
    if (lane < 30 && freq_d[lane]) bits += (uint64_t)freq_d[lane] * ((tab.d[lane] >> 16) + dist_extra_bits(lane));
    for (int off = 32; off > 0; off >>= 1) bits += __shfl_xor(bits, off);
    *payload_bits = bits;
    return hbits;
}

__global__ __launch_bounds__(64) void k_huffman(uint32_t n_chunks, int strategy, const uint32_t *__restrict__ freq,
                                                 int64_t final_chunk, uint32_t *__restrict__ tab,
                                                 uint32_t *__restrict__ hdr, uint32_t *__restrict__ hdr_bits,
                                                 uint64_t *__restrict__ block_bits) {
    __shared__ uint32_t s_freq[kTabStride];
    __shared__ BlockTables s_tab;
    __shared__ uint32_t s_hdr[kHdrWords];
    __shared__ HeaderWork s_work;
    __shared__ uint32_t s_cnt[17], s_nxt[17], s_misc[2];

    const uint32_t c = blockIdx.x;
    if (c >= n_chunks) return;
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < kTabStride; i += 64) s_freq[i] = freq[(uint64_t)c * kTabStride + i];
    for (uint32_t i = lane; i < kHdrWords; i += 64) s_hdr[i] = 0;
    __syncthreads();
    if (lane == 0 && s_freq[256] == 0) s_freq[256] = 1;  // "ensure terminating character is encoded", src/huffman.rs:264
    __syncthreads();
    uint64_t payload = 0;
    const uint32_t hbits =
        build_block_wave(s_freq, s_freq + kDOff, strategy, (int64_t)c == final_chunk || final_chunk == kAllFinal, s_tab,
                         s_hdr, &payload, s_work, s_cnt, s_nxt, s_misc, lane);
    __syncthreads();
    for (uint32_t i = lane; i < kHuffMaxSyms; i += 64) tab[(uint64_t)c * kTabStride + i] = s_tab.ll[i];
    if (lane < 32) tab[(uint64_t)c * kTabStride + kDOff + lane] = s_tab.d[lane];
    for (uint32_t i = lane; i < kHdrWords; i += 64) hdr[(uint64_t)c * kHdrWords + i] = s_hdr[i];
    if (lane == 0) {
        hdr_bits[c] = hbits;
        block_bits[c] = (uint64_t)hbits + payload;
    }
}

hipError_t launch_huffman(uint32_t n_chunks, int strategy, const uint32_t *freq, int64_t final_chunk, uint32_t *tab,
                          uint32_t *hdr, uint32_t *hdr_bits, uint64_t *block_bits, hipStream_t s) {
    if (!n_chunks) return hipSuccess;
    hipLaunchKernelGGL(k_huffman, dim3(n_chunks), dim3(64), 0, s, n_chunks, strategy, freq, final_chunk, tab, hdr,
                       hdr_bits, block_bits);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Bit writer.  A chunk's units (plus one virtual end-of-block unit) are cut
// into tiles of kEncTile; k_tile_bits sums the coded length of each tile,
// k_offsets turns the sums into exclusive offsets per chunk and the block
// lengths into absolute bit offsets, k_encode packs every unit at its offset.
// What is written equals write_compressed_data (src/compress.rs:276-374)
// through BitWriter::write_bits (src/bitwriter.rs:90-98).
// ---------------------------------------------------------------------------
struct Coded {
    uint64_t bits;  // LSB-first payload
    uint32_t n;     // number of bits (<= 48)
};
__device__ __forceinline__ Coded code_unit(uint32_t tok, const uint32_t *ll, const uint32_t *dt) {
    Coded c;
    if (!ZG_TOK_IS_MARKER(tok)) {  // literal_writer
        const uint32_t t = ll[tok & 0x1FFu];
        c.bits = t & 0xFFFFu;
        c.n = t >> 16;
        return c;
    }
    const uint32_t len = ZG_TOK_LEN(tok), dist = ZG_TOK_DIST(tok);
    const uint32_t li = length_code_index(len);  // length_writer
    const uint32_t lt = ll[257u + li];
    uint64_t bits = lt & 0xFFFFu;
    uint32_t n = lt >> 16;
    const uint32_t leb = length_extra_bits(li);
    bits |= (uint64_t)(len - length_base(li)) << n;
    n += leb;
    const uint32_t dc = dist_code(dist);  // distance_writer
    const uint32_t dtab = dt[dc];
    bits |= (uint64_t)(dtab & 0xFFFFu) << n;
    n += dtab >> 16;
    const uint32_t deb = dist_extra_bits(dc);
    bits |= (uint64_t)(dist - dist_base(dc)) << n;
    n += deb;
    c.bits = bits;
    c.n = n;
    return c;
}
constexpr uint32_t kTokEob = 0x100u;  // virtual unit: end-of-block symbol 256 (literal path of code_unit)

__global__ __launch_bounds__(kEncThreads) void k_tile_bits(uint32_t max_tiles, const uint32_t *__restrict__ tokens,
                                                            uint64_t first, uint64_t chunk_bytes, uint32_t c0,
                                                            uint64_t tok_base, const uint32_t *__restrict__ ntok,
                                                            const uint32_t *__restrict__ tab,
                                                            uint32_t *__restrict__ tile_bits) {
    __shared__ uint32_t s_ll[kHuffMaxSyms];
    __shared__ uint32_t s_d[32];
    __shared__ uint32_t s_red[kEncThreads / 64];
    const uint32_t ci = blockIdx.y, tid = threadIdx.x;
    const uint32_t nunits = ntok[ci] + 1u;  // + end of block
    const uint32_t nt = (uint32_t)(((uint64_t)nunits + kEncTile - 1) / kEncTile);
    // The sums are only ever added up to where a tile's bits begin: nobody asks for the last tile's (the block's
    // length comes with its code tables).  For a block of one tile -- nearly every 4 KiB record -- there is nothing
    // to do.
    if (blockIdx.x == 0 && tid == 0) tile_bits[(uint64_t)ci * max_tiles + nt - 1] = 0;
    if (blockIdx.x + 1 >= nt) return;
    for (uint32_t i = tid; i < kHuffMaxSyms; i += kEncThreads) s_ll[i] = tab[(uint64_t)ci * kTabStride + i];
    if (tid < 32) s_d[tid] = tab[(uint64_t)ci * kTabStride + kDOff + tid];
    __syncthreads();
    // a workgroup takes the tiles blockIdx.x, blockIdx.x + gridDim.x, ...: one each when the grid has a column per
    // tile, all of a chunk's when it has one column (chunks of few tiles: a column per possible tile would be
    // workgroups that find nothing to do)
    for (uint32_t tile = blockIdx.x; tile + 1 < nt; tile += gridDim.x) {
    const uint32_t *tok = tokens + (first + (uint64_t)(c0 + ci) * chunk_bytes - tok_base);
    // all of a thread's units are loaded before the first is looked at (a load inside the loop was waited for on
    // every turn); indices past the block's end are clamped and their units dropped
    constexpr uint32_t kPer = kEncTile / kEncThreads;
    uint32_t t[kPer];
#pragma unroll
    for (uint32_t k = 0; k < kPer; k++) {
        const uint32_t u = tile * kEncTile + k * kEncThreads + tid;
        t[k] = tok[u + 1 < nunits ? u : (nunits >= 2u ? nunits - 2u : 0u)];  // (the last unit is the end of block: not in memory)
    }
    uint32_t sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < kPer; k++) {
        const uint32_t u = tile * kEncTile + k * kEncThreads + tid;
        const uint32_t n = code_unit(u + 1 < nunits ? t[k] : kTokEob, s_ll, s_d).n;  // (past the end: any valid unit)
        sum += u < nunits ? n : 0u;
    }
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
    if ((tid & 63) == 0) s_red[tid >> 6] = sum;
    __syncthreads();
    if (tid == 0) {
        uint32_t t = 0;
        for (uint32_t i = 0; i < kEncThreads / 64; i++) t += s_red[i];
        tile_bits[(uint64_t)ci * max_tiles + tile] = t;
    }
    __syncthreads();  // s_red is written again on the next turn
    }
}

// exclusive scan of each chunk's tile sums (in place), one workgroup per chunk
__global__ __launch_bounds__(256) void k_tile_scan(uint32_t max_tiles, const uint32_t *__restrict__ ntok,
                                                    uint32_t *__restrict__ tile_bits) {
    __shared__ uint32_t s_part[256];
    const uint32_t ci = blockIdx.x, tid = threadIdx.x;
    const uint32_t nt = (ntok[ci] + 1u + kEncTile - 1) / kEncTile;
    uint32_t *tb = tile_bits + (uint64_t)ci * max_tiles;
    const uint32_t per = (nt + 255) / 256;
    const uint32_t lo = tid * per, hi = (lo + per < nt) ? lo + per : nt;
    uint32_t s = 0;
    for (uint32_t i = lo; i < hi; i++) s += tb[i];
    s_part[tid] = s;
    __syncthreads();
    for (uint32_t off = 1; off < 256; off <<= 1) {
        const uint32_t v = tid >= off ? s_part[tid - off] : 0u;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    uint32_t run = s_part[tid] - s;
    for (uint32_t i = lo; i < hi; i++) {
        const uint32_t v = tb[i];
        tb[i] = run;
        run += v;
    }
}

__global__ __launch_bounds__(256) void k_tile_scan_few(uint32_t n_chunks, uint32_t max_tiles,
                                                        const uint32_t *__restrict__ ntok,
                                                        uint32_t *__restrict__ tile_bits) {
    const uint32_t ci = blockIdx.x * 256 + threadIdx.x;
    if (ci >= n_chunks) return;
    const uint32_t nt = (ntok[ci] + 1u + kEncTile - 1) / kEncTile;
    uint32_t *tb = tile_bits + (uint64_t)ci * max_tiles;
    uint32_t run = 0;
    for (uint32_t i = 0; i < nt; i++) {
        const uint32_t v = tb[i];
        tb[i] = run;
        run += v;
    }
}

// absolute bit offset of every block: carry[0] is the offset of the first
// block of this batch on entry and of the next batch on exit
__global__ __launch_bounds__(256) void k_block_offsets(uint32_t n_chunks, const uint64_t *__restrict__ block_bits,
                                                        uint64_t *__restrict__ bit_off, uint64_t *carry) {
    __shared__ uint64_t s_part[256];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (n_chunks + 255) / 256;
    const uint32_t lo = tid * per, hi = (lo + per < n_chunks) ? lo + per : n_chunks;
    uint64_t s = 0;
    for (uint32_t i = lo; i < hi && lo < n_chunks; i++) s += block_bits[i];
    s_part[tid] = s;
    __syncthreads();
    for (uint32_t off = 1; off < 256; off <<= 1) {
        const uint64_t v = tid >= off ? s_part[tid - off] : 0ull;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    const uint64_t base = carry[0];
    uint64_t run = base + s_part[tid] - s;
    for (uint32_t i = lo; i < hi && lo < n_chunks; i++) {
        bit_off[i] = run;
        run += block_bits[i];
    }
    __syncthreads();
    if (tid == 255) carry[0] = base + s_part[255];
}

// Record batches: every chunk is a zlib stream of its own, 2 header bytes + its block padded to a byte + 4
// bytes of Adler-32 (reference src/compress.rs:51-86).  carry[0] is the byte offset of the batch's first
// stream on entry and of the next batch's on exit.
__global__ __launch_bounds__(1024) void k_batch_offsets(uint32_t n_chunks, const uint64_t *__restrict__ block_bits,
                                                         uint64_t *__restrict__ bit_off,
                                                         const uint64_t *__restrict__ carry_in,
                                                         uint64_t *__restrict__ carry_out,
                                                         uint64_t *__restrict__ out_offsets) {
    // A workgroup per 8192 streams, eight neighbours per thread.  It first adds up the lengths of all streams
    // before its own (eight workgroups at most: the last reads 56 values per thread, all of them in the L2), then
    // scans its own: inside every wave, then over the 16 wave totals.  The total goes to another word than the one
    // the start is read from: a workgroup may begin after the last one has finished.
    constexpr uint32_t kPer = 8;
    __shared__ uint64_t s_wave[16];
    __shared__ uint64_t s_before[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint32_t i0 = blockIdx.x * 1024u * kPer;
    uint64_t pre = 0;
    for (uint32_t j0 = 0; j0 < i0; j0 += 1024 * kPer) {
        uint64_t v[kPer];
#pragma unroll
        for (uint32_t k = 0; k < kPer; k++) v[k] = block_bits[j0 + k * 1024u + tid];  // (all below i0 <= n_chunks)
#pragma unroll
        for (uint32_t k = 0; k < kPer; k++) pre += 6 + ((v[k] + 7) >> 3);
    }
    for (int off = 32; off > 0; off >>= 1) pre += __shfl_down(pre, off);
    if (lane == 0) s_before[wv] = pre;
    const uint32_t i = i0 + tid * kPer;
    uint64_t len[kPer], mine = 0;
#pragma unroll
    for (uint32_t k = 0; k < kPer; k++) {
        len[k] = i + k < n_chunks ? 6 + ((block_bits[i + k] + 7) >> 3) : 0;
        mine += len[k];
    }
    uint64_t incl = mine;
    for (uint32_t off = 1; off < 64; off <<= 1) {
        const uint64_t up = __shfl_up(incl, off);
        if (lane >= off) incl += up;
    }
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    uint64_t before = carry_in[0], total = 0;
    for (uint32_t w = 0; w < 16; w++) {
        const uint64_t t = s_wave[w];
        before += s_before[w] + (w < wv ? t : 0);
        total += t + s_before[w];
    }
    uint64_t at = before + incl - mine;
#pragma unroll
    for (uint32_t k = 0; k < kPer; k++) {
        if (i + k < n_chunks) {
            out_offsets[i + k] = at;
            bit_off[i + k] = (at + 2) * 8;
        }
        at += len[k];
    }
    if (tid == 0 && blockIdx.x + 1 == gridDim.x) {
        const uint64_t end = carry_in[0] + total;
        carry_out[0] = end;
        out_offsets[n_chunks] = end;  // end of the last stream (the next batch rewrites it)
    }
}
__global__ __launch_bounds__(256) void k_batch_frame(uint32_t n_chunks, const uint64_t *__restrict__ block_bits,
                                                      const uint64_t *__restrict__ out_offsets,
                                                      const uint32_t *__restrict__ adler_part,
                                                      uint8_t *__restrict__ out) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n_chunks) return;
    uint8_t *o = out + out_offsets[c];
    o[0] = 0x78;  // CMF, src/compress.rs:53-62
    o[1] = 0x01;  // FLG, src/compress.rs:64-72
    uint8_t *t = o + 2 + ((block_bits[c] + 7) >> 3);
    const uint32_t a = adler_part[2 * c], b = adler_part[2 * c + 1];
    t[0] = (uint8_t)(b >> 8);  // big-endian b << 16 | a, src/compress.rs:82-83
    t[1] = (uint8_t)b;
    t[2] = (uint8_t)(a >> 8);
    t[3] = (uint8_t)a;
}
hipError_t launch_batch_offsets(uint32_t n_chunks, const uint64_t *block_bits, uint64_t *bit_off,
                                const uint64_t *carry_in, uint64_t *carry_out, uint64_t *out_offsets, hipStream_t s) {
    if (!n_chunks) return hipSuccess;
    if (carry_in == carry_out) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_batch_offsets, dim3((n_chunks + 8191) / 8192), dim3(1024), 0, s, n_chunks, block_bits, bit_off,
                       carry_in, carry_out, out_offsets);
    return hipGetLastError();
}
hipError_t launch_batch_frame(uint32_t n_chunks, const uint64_t *block_bits, const uint64_t *out_offsets,
                              const uint32_t *adler_part, uint8_t *out, hipStream_t s) {
    if (!n_chunks) return hipSuccess;
    hipLaunchKernelGGL(k_batch_frame, dim3((n_chunks + 255) / 256), dim3(256), 0, s, n_chunks, block_bits, out_offsets,
                       adler_part, out);
    return hipGetLastError();
}

hipError_t launch_tile_bits(uint32_t n_chunks, uint32_t max_tiles, const uint32_t *tokens, const ChunkGrid &g,
                            uint32_t c0, uint64_t tok_base, const uint32_t *ntok, const uint32_t *tab,
                            uint32_t *tile_bits, hipStream_t s) {
    if (!n_chunks) return hipSuccess;
    // (the last possible tile is a last tile wherever it exists)
    const uint32_t cols = max_tiles <= kScanFew ? 1u : max_tiles - 1u;
    hipLaunchKernelGGL(k_tile_bits, dim3(cols, n_chunks), dim3(kEncThreads), 0, s, max_tiles, tokens, g.first,
                       g.chunk_bytes, c0, tok_base, ntok, tab, tile_bits);
    return hipGetLastError();
}
hipError_t launch_tile_scan(uint32_t n_chunks, uint32_t max_tiles, const uint32_t *ntok, uint32_t *tile_bits,
                            hipStream_t s) {
    if (!n_chunks) return hipSuccess;
    if (max_tiles <= kScanFew)
        hipLaunchKernelGGL(k_tile_scan_few, dim3((n_chunks + 255) / 256), dim3(256), 0, s, n_chunks, max_tiles, ntok,
                           tile_bits);
    else
        hipLaunchKernelGGL(k_tile_scan, dim3(n_chunks), dim3(256), 0, s, max_tiles, ntok, tile_bits);
    return hipGetLastError();
}
hipError_t launch_offsets(uint32_t n_chunks, uint32_t max_tiles, const uint32_t *ntok, uint32_t *tile_bits,
                          const uint64_t *block_bits, uint64_t *bit_off, uint64_t *carry, hipStream_t s) {
    if (!n_chunks) return hipSuccess;
    const hipError_t e = launch_tile_scan(n_chunks, max_tiles, ntok, tile_bits, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_block_offsets, dim3(1), dim3(256), 0, s, n_chunks, block_bits, bit_off, carry);
    return hipGetLastError();
}

// k_encode: one workgroup per tile of units.  Units are packed into an LDS
// image of the tile's bit range with LDS atomic OR, then the image is written
// out: interior words with plain stores, the two boundary words (shared with
// the neighbouring tile, block header or block) with global atomic OR.  The
// output buffer must be zero before the first launch.
constexpr uint32_t kEncWords = (kEncTile * 48u + 31u) / 32u + 3u;

__global__ __launch_bounds__(kEncThreads) void k_encode(uint32_t max_tiles, const uint32_t *__restrict__ tokens,
                                                         uint64_t first, uint64_t chunk_bytes, uint32_t c0,
                                                         uint64_t tok_base, const uint32_t *__restrict__ ntok,
                                                         const uint32_t *__restrict__ tab,
                                                         const uint32_t *__restrict__ hdr,
                                                         const uint32_t *__restrict__ hdr_bits,
                                                         const uint32_t *__restrict__ tile_off,
                                                         const uint64_t *__restrict__ bit_off, uint32_t *out,
                                                         uint64_t out_cap_bits, uint32_t *status) {
    __shared__ uint32_t s_ll[kHuffMaxSyms];
    __shared__ uint32_t s_d[32];
    __shared__ uint32_t s_img[kEncWords];
    __shared__ uint32_t s_scan[kEncThreads / 64];

    const uint32_t ci = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    const uint32_t nunits = ntok[ci] + 1u;
    if ((uint64_t)tile * kEncTile >= nunits) return;
    const uint64_t blk_bit = bit_off[ci];
    const uint32_t hb = hdr_bits[ci];
    const uint64_t tile_bit = blk_bit + hb + tile_off[(uint64_t)ci * max_tiles + tile];

    for (uint32_t i = tid; i < kHuffMaxSyms; i += kEncThreads) s_ll[i] = tab[(uint64_t)ci * kTabStride + i];
    if (tid < 32) s_d[tid] = tab[(uint64_t)ci * kTabStride + kDOff + tid];
    for (uint32_t i = tid; i < kEncWords; i += kEncThreads) s_img[i] = 0;
    __syncthreads();

    // the block header travels with tile 0
    if (tile == 0) {
        const uint32_t nw = (hb + 31u) >> 5;
        const uint32_t sh = (uint32_t)(blk_bit & 31u);
        const uint64_t w0 = blk_bit >> 5;
        if (blk_bit + hb > out_cap_bits) {
            if (tid == 0) atomicOr(status, 1u);
        } else {
            for (uint32_t i = tid; i < nw; i += kEncThreads) {
                const uint32_t v = hdr[(uint64_t)ci * kHdrWords + i];
                if (v) {
                    atomicOr(&out[w0 + i], v << sh);
                    if (sh) atomicOr(&out[w0 + i + 1], v >> (32u - sh));
                }
            }
        }
    }

    constexpr uint32_t kPer = kEncTile / kEncThreads;
    const uint32_t *tok = tokens + (first + (uint64_t)(c0 + ci) * chunk_bytes - tok_base);
    Coded cu[kPer];
    uint32_t sum = 0;
    const uint32_t u0 = tile * kEncTile + tid * kPer;
    // all of a thread's units are loaded before the first is looked at (see k_tile_bits)
    uint32_t tk[kPer];
#pragma unroll
    for (uint32_t k = 0; k < kPer; k++) {
        const uint32_t u = u0 + k;
        tk[k] = tok[u + 1 < nunits ? u : (nunits >= 2u ? nunits - 2u : 0u)];
    }
#pragma unroll
    for (uint32_t k = 0; k < kPer; k++) {
        const uint32_t u = u0 + k;
        cu[k] = code_unit(u + 1 < nunits ? tk[k] : kTokEob, s_ll, s_d);  // (past the end: any valid unit, dropped below)
        if (u >= nunits) {
            cu[k].bits = 0;
            cu[k].n = 0;
        }
        sum += cu[k].n;
    }
    // where a thread's bits go: a scan inside every wave, then over the wave totals
    uint32_t incl = sum;
    for (uint32_t off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off);
        if ((tid & 63u) >= off) incl += up;
    }
    if ((tid & 63u) == 63u) s_scan[tid >> 6] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < kEncThreads / 64; w++) {
        const uint32_t v = s_scan[w];
        before += w < (tid >> 6) ? v : 0u;
        total += v;
    }
    const uint32_t img_sh = (uint32_t)(tile_bit & 31u);  // bit position of the tile start inside image word 0
    uint32_t pos = img_sh + before + incl - sum;
#pragma unroll
    for (uint32_t k = 0; k < kPer; k++) {
        const uint32_t n = cu[k].n;
        if (n) {
            const uint32_t w = pos >> 5, sh = pos & 31u;
            const uint64_t lo = cu[k].bits << sh;
            atomicOr(&s_img[w], (uint32_t)lo);
            if (sh + n > 32) atomicOr(&s_img[w + 1], (uint32_t)(lo >> 32));
            if (sh + n > 64) atomicOr(&s_img[w + 2], (uint32_t)(cu[k].bits >> (64u - sh)));
            pos += n;
        }
    }
    __syncthreads();
    if (total == 0) return;
    if (tile_bit + total > out_cap_bits) {
        if (tid == 0) atomicOr(status, 1u);
        return;
    }
    const uint32_t nw = (img_sh + total + 31u) >> 5;
    const uint64_t w0 = tile_bit >> 5;
    for (uint32_t i = tid; i < nw; i += kEncThreads) {
        const uint32_t v = s_img[i];
        if (i == 0 || i + 1 == nw) {
            if (v) atomicOr(&out[w0 + i], v);
        } else {
            out[w0 + i] = v;
        }
    }
}

// k_zero_seams: k_encode stores the interior words of a tile's bit range and ORs into its first and last word
// and into the words of the block header, which it shares with its neighbours.  Only those words have to be zero
// beforehand, not the whole output: for every block the words from the first one that begins inside it up to the
// one its first tile starts in (the header's), and for every tile the word its range ends in (and the one before,
// should the range end on a word boundary).  The word a block
// begins in the middle of is the end word of the tile before it -- of this batch, of the batch before (zeroed then,
// and already ORed into: it must not be touched again), or the first word of the output, which the caller zeroes.
__global__ __launch_bounds__(256) void k_zero_seams(uint32_t max_tiles, const uint32_t *__restrict__ ntok,
                                                     const uint32_t *__restrict__ hdr_bits,
                                                     const uint32_t *__restrict__ tile_off,
                                                     const uint64_t *__restrict__ bit_off,
                                                     const uint64_t *__restrict__ block_bits, uint32_t *out,
                                                     uint64_t out_cap_bits, uint32_t separate) {
    const uint32_t ci = blockIdx.y;
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    const uint32_t nt = (ntok[ci] + 1u + kEncTile - 1) / kEncTile;
    const uint64_t blk_bit = bit_off[ci], last_word = out_cap_bits >> 5;
    // the word this batch begins in the middle of holds bits of the batch before: never cleared here, even when a
    // short block ends in it
    const uint64_t keep = (bit_off[0] & 31u) ? bit_off[0] >> 5 : ~0ull;
    const uint32_t hb = hdr_bits[ci];
    // separate != 0: every block is a stream of its own between two header bytes and four trailer bytes (written
    // byte by byte afterwards).  The word a block begins in the middle of is then no tile's end word: the bytes from
    // the block's first to the end of that word are cleared one by one -- the bytes below them may hold the stream
    // before, of this batch or of the last one.
    if (separate && blockIdx.x == 0 && threadIdx.x == 0 && (blk_bit & 31u)) {
        uint8_t *ob = reinterpret_cast<uint8_t *>(out);
        for (uint64_t b = blk_bit >> 3; b < ((blk_bit >> 5) + 1) * 4 && b * 8 < out_cap_bits; b++) ob[b] = 0;
    }
    if (blockIdx.x == 0) {
        const uint64_t w_hi = (blk_bit + hb) >> 5;
        for (uint64_t w = ((blk_bit + 31) >> 5) + threadIdx.x; w <= w_hi && w <= last_word; w += 256)
            if (w != keep) out[w] = 0;
    }
    if (t < nt) {
        const uint64_t start_bit = blk_bit + hb + tile_off[(uint64_t)ci * max_tiles + t];
        const uint64_t end_bit = blk_bit + (t + 1 < nt ? (uint64_t)hb + tile_off[(uint64_t)ci * max_tiles + t + 1]
                                                      : block_bits[ci]);
        const uint64_t w = end_bit >> 5;
        if (w <= last_word && w != keep) out[w] = 0;
        // a range that ends on a word boundary: its last word is ORed into all the same (unless it is also its first
        // word, which the tile before or the header's span has cleared)
        if ((end_bit & 31u) == 0 && w > (start_bit >> 5) + 1 && w - 1 <= last_word && w - 1 != keep) out[w - 1] = 0;
    }
}

// the same for chunks of up to 64 tiles (records): a wave per chunk instead of a workgroup of mostly idle threads
__global__ __launch_bounds__(256) void k_zero_seams_few(uint32_t n_chunks, uint32_t max_tiles,
                                                         const uint32_t *__restrict__ ntok,
                                                         const uint32_t *__restrict__ hdr_bits,
                                                         const uint32_t *__restrict__ tile_off,
                                                         const uint64_t *__restrict__ bit_off,
                                                         const uint64_t *__restrict__ block_bits, uint32_t *out,
                                                         uint64_t out_cap_bits, uint32_t separate) {
    const uint32_t ci = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (ci >= n_chunks) return;
    const uint32_t nt = (ntok[ci] + 1u + kEncTile - 1) / kEncTile;
    const uint64_t blk_bit = bit_off[ci], last_word = out_cap_bits >> 5;
    const uint64_t keep = (bit_off[0] & 31u) ? bit_off[0] >> 5 : ~0ull;
    const uint32_t hb = hdr_bits[ci];
    if (separate && lane == 0 && (blk_bit & 31u)) {
        uint8_t *ob = reinterpret_cast<uint8_t *>(out);
        for (uint64_t b = blk_bit >> 3; b < ((blk_bit >> 5) + 1) * 4 && b * 8 < out_cap_bits; b++) ob[b] = 0;
    }
    const uint64_t w_hi = (blk_bit + hb) >> 5;
    for (uint64_t w = ((blk_bit + 31) >> 5) + lane; w <= w_hi && w <= last_word; w += 64)
        if (w != keep) out[w] = 0;
    if (lane < nt) {
        const uint64_t start_bit = blk_bit + hb + tile_off[(uint64_t)ci * max_tiles + lane];
        const uint64_t end_bit = blk_bit + (lane + 1 < nt ? (uint64_t)hb + tile_off[(uint64_t)ci * max_tiles + lane + 1]
                                                         : block_bits[ci]);
        const uint64_t w = end_bit >> 5;
        if (w <= last_word && w != keep) out[w] = 0;
        if ((end_bit & 31u) == 0 && w > (start_bit >> 5) + 1 && w - 1 <= last_word && w - 1 != keep) out[w - 1] = 0;
    }
}

hipError_t launch_zero_seams(uint32_t n_chunks, uint32_t max_tiles, const uint32_t *ntok, const uint32_t *hdr_bits,
                             const uint32_t *tile_off, const uint64_t *bit_off, const uint64_t *block_bits,
                             uint32_t *out, uint64_t out_cap_bits, bool separate, hipStream_t s) {
    if (!n_chunks) return hipSuccess;
    if (max_tiles <= 64)
        hipLaunchKernelGGL(k_zero_seams_few, dim3((n_chunks + 3) / 4), dim3(256), 0, s, n_chunks, max_tiles, ntok, hdr_bits,
                           tile_off, bit_off, block_bits, out, out_cap_bits, separate ? 1u : 0u);
    else
        hipLaunchKernelGGL(k_zero_seams, dim3((max_tiles + 255) / 256, n_chunks), dim3(256), 0, s, max_tiles, ntok,
                           hdr_bits, tile_off, bit_off, block_bits, out, out_cap_bits, separate ? 1u : 0u);
    return hipGetLastError();
}

hipError_t launch_encode(uint32_t n_chunks, uint32_t max_tiles, const uint32_t *tokens, const ChunkGrid &g,
                         uint32_t c0, uint64_t tok_base, const uint32_t *ntok, const uint32_t *tab,
                         const uint32_t *hdr, const uint32_t *hdr_bits, const uint32_t *tile_off,
                         const uint64_t *bit_off, uint32_t *out, uint64_t out_cap_bits, uint32_t *status,
                         hipStream_t s) {
    if (!n_chunks) return hipSuccess;
    hipLaunchKernelGGL(k_encode, dim3(max_tiles, n_chunks), dim3(kEncThreads), 0, s, max_tiles, tokens, g.first,
                       g.chunk_bytes, c0, tok_base, ntok, tab, hdr, hdr_bits, tile_off, bit_off, out, out_cap_bits,
                       status);
    return hipGetLastError();
}

}  // namespace zg
