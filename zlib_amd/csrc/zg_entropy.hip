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
//                  sub-tile but the first), flagging the positions it visits;
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

__global__ __launch_bounds__(64) void k_parse_spec(uint64_t first, uint64_t end, uint64_t chunk_bytes,
        const uint32_t *__restrict__ clen, uint32_t c0,
                                                    uint32_t n_chunks, uint32_t spc,
                                                    const uint32_t *__restrict__ match, uint64_t base,
                                                    uint8_t *__restrict__ flag, uint32_t *__restrict__ sub_exit) {
    const uint64_t gid = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    const uint32_t ci = (uint32_t)(gid / spc), j = (uint32_t)(gid % spc);
    if (ci >= n_chunks) return;
    const uint64_t cs = first + (uint64_t)(c0 + ci) * chunk_bytes;
    const uint64_t ce = chunk_end(cs, chunk_bytes, end, clen, c0 + ci);
    const uint64_t s0 = cs + (uint64_t)j * kParseSub;
    if (s0 >= ce && !(j == 0)) return;
    const uint64_t s1 = (s0 + kParseSub < ce) ? s0 + kParseSub : ce;
    uint64_t p = s0;
    while (p < s1) {
        flag[p - base] = 1;
        p += parse_step(match, p - base);
    }
    sub_exit[(uint64_t)ci * spc + j] = (uint32_t)(p - cs);
}

__global__ __launch_bounds__(64) void k_parse_fix(uint64_t first, uint64_t end, uint64_t chunk_bytes,
        const uint32_t *__restrict__ clen, uint32_t c0,
                                                   uint32_t n_chunks, uint32_t spc,
                                                   const uint32_t *__restrict__ match, uint64_t base,
                                                   uint8_t *__restrict__ flag, const uint32_t *__restrict__ sub_exit,
                                                   uint32_t *__restrict__ sub_exit2, uint8_t *__restrict__ merged) {
    const uint64_t gid = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    const uint32_t ci = (uint32_t)(gid / spc), j = (uint32_t)(gid % spc);
    if (ci >= n_chunks) return;
    const uint64_t cs = first + (uint64_t)(c0 + ci) * chunk_bytes;
    const uint64_t ce = chunk_end(cs, chunk_bytes, end, clen, c0 + ci);
    const uint64_t s0 = cs + (uint64_t)j * kParseSub;
    const uint64_t k = (uint64_t)ci * spc + j;
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
        if (flag[t - base]) {
            met = true;
            break;
        }
        t += parse_step(match, t - base);
    }
    const uint64_t stop = met ? t : s1;  // flags from here on are already right (or there are none)
    for (uint64_t s = s0; s < stop;) {   // drop the guessed prefix
        flag[s - base] = 0;
        s += parse_step(match, s - base);
    }
    for (uint64_t u = e; u < stop;) {    // put the true prefix
        flag[u - base] = 1;
        u += parse_step(match, u - base);
    }
    sub_exit2[k] = met ? sub_exit[k] : (uint32_t)(t - cs);
    merged[k] = met ? 1 : 0;
}

__global__ __launch_bounds__(64) void k_parse_check(uint64_t first, uint64_t end, uint64_t chunk_bytes,
        const uint32_t *__restrict__ clen, uint32_t c0,
                                                     uint32_t spc, const uint32_t *__restrict__ match, uint64_t base,
                                                     uint8_t *__restrict__ flag, const uint32_t *__restrict__ sub_exit2,
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
    const uint64_t r0 = cs + (uint64_t)(bad + 1) * kParseSub;
    for (uint64_t p = r0 + lane; p < ce; p += 64) flag[p - base] = 0;
    __syncthreads();
    if (lane == 0) {
        uint64_t p = cs + sub_exit2[(uint64_t)ci * spc + bad];
        while (p < ce) {
            flag[p - base] = 1;
            p += parse_step(match, p - base);
        }
    }
}

// ---------------------------------------------------------------------------
// Unit emission: flagged positions become packed LZ77 units, in order, and
// are counted as HuffmanTree::from_lz77 does (src/huffman.rs:243-264).
// k_emit_count sums the flags per tile, k_emit_scan turns the sums into
// offsets per chunk, k_emit writes the units and the histograms.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kEmitThreads) void k_emit_count(uint64_t first, uint64_t end, uint64_t chunk_bytes,
                                                              const uint32_t *__restrict__ clen, uint32_t c0,
                                                              uint32_t tpc,
                                                              const uint8_t *__restrict__ flag, uint64_t base,
                                                              uint32_t *__restrict__ tile_cnt) {
    __shared__ uint32_t s_red[kEmitThreads / 64];
    const uint32_t ci = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    const uint64_t cs = first + (uint64_t)(c0 + ci) * chunk_bytes;
    const uint64_t ce = chunk_end(cs, chunk_bytes, end, clen, c0 + ci);
    const uint64_t t0 = cs + (uint64_t)tile * kEmitTile;
    uint32_t cnt = 0;
    if (t0 < ce) {
        const uint32_t tn = (uint32_t)((ce - t0 < kEmitTile) ? ce - t0 : kEmitTile);
        const uint8_t *f = flag + (t0 - base);
        for (uint32_t i = tid; i < tn; i += kEmitThreads) cnt += f[i];
    }
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
    if ((tid & 63) == 0) s_red[tid >> 6] = cnt;
    __syncthreads();
    if (tid == 0) {
        uint32_t t = 0;
        for (uint32_t i = 0; i < kEmitThreads / 64; i++) t += s_red[i];
        tile_cnt[(uint64_t)ci * tpc + tile] = t;
    }
}

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

__global__ __launch_bounds__(kEmitThreads) void k_emit(const uint8_t *__restrict__ data, uint64_t first, uint64_t end,
                                                        uint64_t chunk_bytes, const uint32_t *__restrict__ clen,
                                                        uint32_t c0, uint32_t tpc,
                                                        const uint8_t *__restrict__ flag,
                                                        const uint32_t *__restrict__ match, uint64_t base,
                                                        const uint32_t *__restrict__ tile_off,
                                                        uint32_t *__restrict__ tokens, uint64_t tok_base,
                                                        uint32_t *__restrict__ freq) {
    __shared__ uint32_t s_hist[kTabStride];
    __shared__ uint32_t s_scan[kEmitThreads];
    const uint32_t ci = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    const uint64_t cs = first + (uint64_t)(c0 + ci) * chunk_bytes;
    const uint64_t ce = chunk_end(cs, chunk_bytes, end, clen, c0 + ci);
    const uint64_t t0 = cs + (uint64_t)tile * kEmitTile;
    if (t0 >= ce) return;
    const uint32_t tn = (uint32_t)((ce - t0 < kEmitTile) ? ce - t0 : kEmitTile);
    for (uint32_t i = tid; i < kTabStride; i += kEmitThreads) s_hist[i] = 0;
    constexpr uint32_t kPer = kEmitTile / kEmitThreads;  // consecutive positions per thread
    const uint32_t b0 = tid * kPer;
    const uint8_t *f = flag + (t0 - base);
    uint32_t fl = 0, cnt = 0;  // flags of my positions as a bit mask
#pragma unroll
    for (uint32_t i = 0; i < kPer; i++)
        if (b0 + i < tn && f[b0 + i]) {
            fl |= 1u << i;
            cnt++;
        }
    s_scan[tid] = cnt;
    __syncthreads();
    for (uint32_t off = 1; off < kEmitThreads; off <<= 1) {
        const uint32_t v = tid >= off ? s_scan[tid - off] : 0u;
        __syncthreads();
        s_scan[tid] += v;
        __syncthreads();
    }
    uint32_t *tok_out = tokens + (cs - tok_base) + tile_off[(uint64_t)ci * tpc + tile] + (s_scan[tid] - cnt);
    const uint32_t *mt = match + (t0 - base);
    while (fl) {
        const uint32_t i = (uint32_t)__builtin_ctz(fl);
        fl &= fl - 1;
        const uint32_t r = b0 + i;
        const uint32_t mv = mt[r];
        uint32_t tok;
        if (mv) {
            const uint32_t len = mv >> 16, dist = mv & 0xFFFFu;
            tok = ZG_TOK_MARKER | (len << 16) | dist;
            atomicAdd(&s_hist[257u + length_code_index(len)], 1u);
            atomicAdd(&s_hist[kDOff + dist_code(dist)], 1u);
        } else {
            tok = data[t0 + r];
            atomicAdd(&s_hist[tok], 1u);
        }
        *tok_out++ = tok;
    }
    __syncthreads();
    for (uint32_t i = tid; i < kTabStride; i += kEmitThreads)
        if (s_hist[i]) atomicAdd(&freq[(uint64_t)ci * kTabStride + i], s_hist[i]);
}

hipError_t launch_parse(const ChunkGrid &g, uint32_t c0, uint32_t c1, const uint32_t *match, uint64_t base,
                        uint8_t *flag, uint64_t flag_bytes, uint32_t *sub_exit, uint32_t *sub_exit2, uint8_t *merged,
                        uint32_t *tile_cnt, uint32_t *tokens, uint64_t tok_base, uint32_t *ntok, uint32_t *freq,
                        hipStream_t s) {
    if (c1 <= c0) return hipSuccess;
    const uint32_t nc = c1 - c0;
    const uint32_t spc = parse_subs_per_chunk(g.chunk_bytes);
    const uint32_t tpc = emit_tiles_per_chunk(g.chunk_bytes);
    hipError_t e = hipMemsetAsync(flag, 0, flag_bytes, s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(freq, 0, (size_t)nc * kTabStride * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    const uint64_t lanes = (uint64_t)nc * spc;
    const uint32_t blocks = (uint32_t)((lanes + 63) / 64);
    hipLaunchKernelGGL(k_parse_spec, dim3(blocks), dim3(64), 0, s, g.first, g.end, g.chunk_bytes, g.chunk_len, c0, nc, spc,
                       match, base, flag, sub_exit);
    hipLaunchKernelGGL(k_parse_fix, dim3(blocks), dim3(64), 0, s, g.first, g.end, g.chunk_bytes, g.chunk_len, c0, nc, spc,
                       match, base, flag, sub_exit, sub_exit2, merged);
    hipLaunchKernelGGL(k_parse_check, dim3(nc), dim3(64), 0, s, g.first, g.end, g.chunk_bytes, g.chunk_len, c0, spc, match,
                       base, flag, sub_exit2, merged);
    hipLaunchKernelGGL(k_emit_count, dim3(tpc, nc), dim3(kEmitThreads), 0, s, g.first, g.end, g.chunk_bytes, g.chunk_len, c0,
                       tpc, flag, base, tile_cnt);
    hipLaunchKernelGGL(k_emit_scan, dim3(nc), dim3(256), 0, s, tpc, tile_cnt, ntok);
    hipLaunchKernelGGL(k_emit, dim3(tpc, nc), dim3(kEmitThreads), 0, s, g.data, g.first, g.end, g.chunk_bytes, g.chunk_len,
                       c0, tpc, flag, match, base, tile_cnt, tokens, tok_base, freq);
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
// k_huffman: one wave per chunk; lane 0 runs the block builder over LDS.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_huffman(uint32_t n_chunks, int strategy, const uint32_t *__restrict__ freq,
                                                 int64_t final_chunk, uint32_t *__restrict__ tab,
                                                 uint32_t *__restrict__ hdr, uint32_t *__restrict__ hdr_bits,
                                                 uint64_t *__restrict__ block_bits) {
    __shared__ uint32_t s_freq[kTabStride];
    __shared__ BlockTables s_tab;
    __shared__ uint32_t s_hdr[kHdrWords];
    __shared__ HeaderWork s_work;
    __shared__ uint32_t s_hbits;
    __shared__ uint64_t s_payload;

    const uint32_t c = blockIdx.x;
    if (c >= n_chunks) return;
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < kTabStride; i += 64) s_freq[i] = freq[(uint64_t)c * kTabStride + i];
    for (uint32_t i = lane; i < kHdrWords; i += 64) s_hdr[i] = 0;
    __syncthreads();
    if (lane == 0) {
        if (s_freq[256] == 0) s_freq[256] = 1;  // "ensure terminating character is encoded", src/huffman.rs:264
        uint64_t payload = 0;
        s_hbits = build_block(s_freq, s_freq + kDOff, strategy, (int64_t)c == final_chunk || final_chunk == kAllFinal, s_tab, s_hdr,
                              &payload,
                              s_work);
        s_payload = payload;
    }
    __syncthreads();
    for (uint32_t i = lane; i < kHuffMaxSyms; i += 64) tab[(uint64_t)c * kTabStride + i] = s_tab.ll[i];
    if (lane < 32) tab[(uint64_t)c * kTabStride + kDOff + lane] = s_tab.d[lane];
    for (uint32_t i = lane; i < kHdrWords; i += 64) hdr[(uint64_t)c * kHdrWords + i] = s_hdr[i];
    if (lane == 0) {
        hdr_bits[c] = s_hbits;
        block_bits[c] = (uint64_t)s_hbits + s_payload;
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
    const uint32_t ci = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    const uint32_t nunits = ntok[ci] + 1u;  // + end of block
    if ((uint64_t)tile * kEncTile >= nunits) return;
    for (uint32_t i = tid; i < kHuffMaxSyms; i += kEncThreads) s_ll[i] = tab[(uint64_t)ci * kTabStride + i];
    if (tid < 32) s_d[tid] = tab[(uint64_t)ci * kTabStride + kDOff + tid];
    __syncthreads();
    const uint32_t *tok = tokens + (first + (uint64_t)(c0 + ci) * chunk_bytes - tok_base);
    uint32_t sum = 0;
    for (uint32_t k = 0; k < kEncTile / kEncThreads; k++) {
        const uint32_t u = tile * kEncTile + k * kEncThreads + tid;
        if (u < nunits) sum += code_unit(u + 1 == nunits ? kTokEob : tok[u], s_ll, s_d).n;
    }
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
    if ((tid & 63) == 0) s_red[tid >> 6] = sum;
    __syncthreads();
    if (tid == 0) {
        uint32_t t = 0;
        for (uint32_t i = 0; i < kEncThreads / 64; i++) t += s_red[i];
        tile_bits[(uint64_t)ci * max_tiles + tile] = t;
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
__global__ __launch_bounds__(256) void k_batch_offsets(uint32_t n_chunks, const uint64_t *__restrict__ block_bits,
                                                        uint64_t *__restrict__ bit_off, uint64_t *carry,
                                                        uint64_t *__restrict__ out_offsets) {
    __shared__ uint64_t s_part[256];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (n_chunks + 255) / 256;
    const uint32_t lo = tid * per, hi = (lo + per < n_chunks) ? lo + per : n_chunks;
    uint64_t s = 0;
    for (uint32_t i = lo; i < hi && lo < n_chunks; i++) s += 6 + ((block_bits[i] + 7) >> 3);
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
        out_offsets[i] = run;
        bit_off[i] = (run + 2) * 8;
        run += 6 + ((block_bits[i] + 7) >> 3);
    }
    __syncthreads();
    if (tid == 255) {
        carry[0] = base + s_part[255];
        out_offsets[n_chunks] = base + s_part[255];  // end of the last stream (the next batch rewrites it)
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
hipError_t launch_batch_offsets(uint32_t n_chunks, const uint64_t *block_bits, uint64_t *bit_off, uint64_t *carry,
                                uint64_t *out_offsets, hipStream_t s) {
    if (!n_chunks) return hipSuccess;
    hipLaunchKernelGGL(k_batch_offsets, dim3(1), dim3(256), 0, s, n_chunks, block_bits, bit_off, carry, out_offsets);
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
    hipLaunchKernelGGL(k_tile_bits, dim3(max_tiles, n_chunks), dim3(kEncThreads), 0, s, max_tiles, tokens, g.first,
                       g.chunk_bytes, c0, tok_base, ntok, tab, tile_bits);
    return hipGetLastError();
}
hipError_t launch_offsets(uint32_t n_chunks, uint32_t max_tiles, const uint32_t *ntok, uint32_t *tile_bits,
                          const uint64_t *block_bits, uint64_t *bit_off, uint64_t *carry, hipStream_t s) {
    if (!n_chunks) return hipSuccess;
    hipLaunchKernelGGL(k_tile_scan, dim3(n_chunks), dim3(256), 0, s, max_tiles, ntok, tile_bits);
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
    __shared__ uint32_t s_scan[kEncThreads];

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
#pragma unroll
    for (uint32_t k = 0; k < kPer; k++) {
        const uint32_t u = u0 + k;
        if (u < nunits) {
            cu[k] = code_unit(u + 1 == nunits ? kTokEob : tok[u], s_ll, s_d);
        } else {
            cu[k].bits = 0;
            cu[k].n = 0;
        }
        sum += cu[k].n;
    }
    s_scan[tid] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < kEncThreads; off <<= 1) {
        const uint32_t v = tid >= off ? s_scan[tid - off] : 0u;
        __syncthreads();
        s_scan[tid] += v;
        __syncthreads();
    }
    const uint32_t total = s_scan[kEncThreads - 1];
    const uint32_t img_sh = (uint32_t)(tile_bit & 31u);  // bit position of the tile start inside image word 0
    uint32_t pos = img_sh + s_scan[tid] - sum;
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
