// Device helpers shared by the match kernels: tile geometry, staging of a tile's bytes and links into LDS,
// unaligned LDS reads, the XCD-aware tile order, the sixteen-candidate compare of the direct window scan.
#pragma once
#include "zg_internal.h"

namespace zg {

// Workgroups are dealt round-robin to the eight XCDs, each with an L2 of its own.  Neighbouring tiles (of the match
// kernels, of the far fix-up) read mostly the same memory, so the block index is turned into a tile index that
// gives every XCD a contiguous run of tiles: block b runs on XCD b % 8 and takes the (b / 8)-th tile of that XCD's run.
__device__ __forceinline__ uint32_t xcd_tile_index(uint32_t b, uint32_t n) {
#ifdef ZG_NO_XCD_REMAP
    return b;
#else
    const uint32_t per = n >> 3, rem = n & 7u, xcd = b & 7u, slot = b >> 3;
    return xcd * per + (xcd < rem ? xcd : rem) + slot;
#endif
}

// Geometry of a match workgroup: positions per tile, positions of window staged before them, threads, and how
// many parked level-2 walks it can hold.  GeomStream is the general one; GeomRecord serves isolated chunks (records) of at most 4 KiB, whose window is
// the record itself -- a twelfth of the LDS, so that many records are resident per CU.
#ifndef ZG_MATCH_STREAM_THREADS
#define ZG_MATCH_STREAM_THREADS 1024
#endif
#ifndef ZG_LO_STREAM_THREADS
#define ZG_LO_STREAM_THREADS 1024
#endif
template <uint32_t kTileT, uint32_t kWinT, uint32_t kThreadsT, uint32_t kParkedT>
struct MatchGeom {
    static constexpr uint32_t tile = kTileT, win = kWinT, threads = kThreadsT, parked = kParkedT;
    static constexpr uint32_t lo_threads = kThreadsT == 1024 ? ZG_LO_STREAM_THREADS : kThreadsT;  // level 2
    static constexpr uint32_t match_threads = kThreadsT == 1024 ? ZG_MATCH_STREAM_THREADS : kThreadsT;  // level 1
    static constexpr uint32_t data_lds = kWinT + kTileT + 16 + 272;  // 16: alignment slack, 272: look-ahead + pad
    static constexpr uint32_t next_lds = kWinT + kTileT + 16;
};
using GeomStream = MatchGeom<kMatchTile, kMaxWindow, kMatchThreads, 1024>;
#ifndef ZG_RECORD_THREADS
#define ZG_RECORD_THREADS 256
#endif
using GeomRecord = MatchGeom<kMatchRecordTile, 0, ZG_RECORD_THREADS, 64>;

// four bytes at an arbitrary LDS byte offset, from two aligned dwords.  gfx950 also reads an LDS dword at any
// alignment with one ds_read_b32 (tools/probe/probe_lds_unaligned.hip: correct for every offset), but far more
// slowly: with it k_match took 9.0 ms instead of 4.1 on 64 MiB of text.
__device__ __forceinline__ uint32_t lds_ld4(const uint8_t *lds, uint32_t off) {
    const uint32_t *w = reinterpret_cast<const uint32_t *>(lds + (off & ~3u));
    const uint32_t lo = w[0], hi = w[1];
    return __builtin_amdgcn_alignbyte(hi, lo, off & 3u);
}

// common prefix of the strings at LDS offsets so and po, at most cap bytes
__device__ __forceinline__ uint32_t lds_lcp(const uint8_t *lds, uint32_t so, uint32_t po, uint32_t cap) {
    uint32_t m = 0;
    while (m < cap) {
        const uint32_t x = lds_ld4(lds, so + m) ^ lds_ld4(lds, po + m);
        if (x) {
            m += (uint32_t)__builtin_ctz(x) >> 3;
            break;
        }
        m += 4;
    }
    return m < cap ? m : cap;
}

// copies links [w0, w0 + nlinks) of the view into LDS
__device__ __forceinline__ void stage_links(const uint16_t *__restrict__ next, uint64_t base, uint64_t w0,
                                            uint32_t nlinks, uint16_t *s_next, uint32_t tid, uint32_t nthreads) {
    const uint16_t *src = next + (w0 - base);
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
        uint4 *d4 = reinterpret_cast<uint4 *>(s_next);
        const uint32_t nv = nlinks >> 3;
        for (uint32_t i = tid; i < nv; i += nthreads) d4[i] = s4[i];
        for (uint32_t i = (nv << 3) + tid; i < nlinks; i += nthreads) s_next[i] = src[i];
    } else {
        for (uint32_t i = tid; i < nlinks; i += nthreads) s_next[i] = src[i];
    }
}

// copies bytes [w0, w0 + nbytes) and links [w0, w0 + nlinks) of the view into LDS
__device__ __forceinline__ void stage_tile(const uint8_t *__restrict__ data, const uint16_t *__restrict__ next,
                                           uint64_t base, uint64_t w0, uint32_t nbytes, uint32_t nlinks,
                                           uint8_t *s_data, uint16_t *s_next, uint32_t tid, uint32_t nthreads,
                                           uint32_t data_cap) {
    // stage bytes: 16-byte vectors when the source is aligned, bytes otherwise
    if (((reinterpret_cast<uintptr_t>(data) + w0) & 15) == 0) {
        const uint4 *src = reinterpret_cast<const uint4 *>(data + w0);
        uint4 *dst = reinterpret_cast<uint4 *>(s_data);
        const uint32_t nv = nbytes >> 4;
        for (uint32_t i = tid; i < nv; i += nthreads) dst[i] = src[i];
        for (uint32_t i = (nv << 4) + tid; i < nbytes; i += nthreads) s_data[i] = data[w0 + i];
    } else {
        for (uint32_t i = tid; i < nbytes; i += nthreads) s_data[i] = data[w0 + i];
    }
    for (uint32_t i = nbytes + tid; i < nbytes + 16 && i < data_cap; i += nthreads) s_data[i] = 0;
    stage_links(next, base, w0, nlinks, s_next, tid, nthreads);
}

// The sixteen candidates a .. a + 15 (a a multiple of 16) against the four bytes p0: the lowest j whose candidate
// agrees in four (kFull) or three bytes, 16 if none.  kEdge also keeps candidates inside [lo, hi).
template <bool kEdge, bool kFull>
__device__ __forceinline__ uint32_t scan16(const uint8_t *s_data, uint32_t a, uint32_t p0, uint32_t lo, uint32_t hi) {
    uint32_t h = 16u;
    if (kEdge && a >= hi) return h;
    const uint4 q = *reinterpret_cast<const uint4 *>(s_data + a);
    const uint32_t w[5] = {q.x, q.y, q.z, q.w, *reinterpret_cast<const uint32_t *>(s_data + a + 16)};
    const uint32_t want = kFull ? p0 : (p0 & 0x00FFFFFFu);
#pragma unroll
    for (int j = 15; j >= 0; j--) {
        uint32_t v = (j & 3) == 0 ? w[j >> 2] : __builtin_amdgcn_alignbyte(w[(j >> 2) + 1], w[j >> 2], (uint32_t)(j & 3));
        if (!kFull) v &= 0x00FFFFFFu;
        bool eq = v == want;
        if (kEdge) eq = eq && a + (uint32_t)j >= lo && a + (uint32_t)j < hi;
        h = eq ? (uint32_t)j : h;
    }
    return h;
}

constexpr uint32_t kNoCandidate = 0xFFFFFFFFu;

// the sixteen candidates a .. a + 15 (a a multiple of 16): bit j is set when candidate a + j lies in [lo, hi) and
// its first four bytes equal p0
__device__ __forceinline__ uint32_t match16(const uint8_t *s_data, uint32_t a, uint32_t p0, uint32_t lo, uint32_t hi) {
    if (a >= hi || a + 16u <= lo) return 0u;
    const uint4 q = *reinterpret_cast<const uint4 *>(s_data + a);
    const uint32_t w[5] = {q.x, q.y, q.z, q.w, *reinterpret_cast<const uint32_t *>(s_data + a + 16)};
    uint32_t mask = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const uint32_t v = (j & 3) == 0 ? w[j >> 2] : __builtin_amdgcn_alignbyte(w[(j >> 2) + 1], w[j >> 2], (uint32_t)(j & 3));
        mask |= (v == p0 ? 1u : 0u) << j;
    }
    const uint32_t below = lo > a ? (1u << (lo - a)) - 1u : 0u;           // candidates under lo
    const uint32_t upto = hi - a < 16u ? (1u << (hi - a)) - 1u : 0xFFFFu;  // candidates under hi
    return mask & upto & ~below;
}

// The rest of a level-1 search by the whole wave: every candidate of [c_lo, c_hi) whose first four bytes equal the
// position's is measured, in the reference's ascending order and under its rules (see k_match): a candidate of
// rec_min bytes or more ends the search when its growing match reaches the position (distance <= length + 1,
// within max_len), is recorded when it is strictly longer than the best so far, and ends the search when it
// cannot be exceeded (length == cap).  All arguments are uniform over the wave; best / bestd / termd are the
// walk's state so far and receive the result.
__device__ __forceinline__ void wave_search(const uint8_t *s_data, uint32_t c_lo, uint32_t c_hi, uint32_t ppo,
                                            uint32_t pp0, uint32_t cap, uint32_t rec_min, uint32_t max_len,
                                            uint32_t lane, uint32_t &best, uint32_t &bestd, uint32_t &termd) {
    for (uint32_t b0 = c_lo & ~15u; b0 < c_hi; b0 += 1024) {
        const uint32_t a = b0 + lane * 16;
        uint32_t hits = match16(s_data, a, pp0, c_lo, c_hi);
        uint32_t lm = 0, ls = 0, lstop = kNoCandidate;  // my longest candidate and where my range stops the search
        bool lterm = false;
        while (hits) {
            const uint32_t s = a + (uint32_t)__builtin_ctz(hits);
            hits &= hits - 1;
            uint32_t m = 4;
            while (m < cap) {
                const uint32_t x = lds_ld4(s_data, s + m) ^ lds_ld4(s_data, ppo + m);
                if (x) {
                    m += (uint32_t)__builtin_ctz(x) >> 3;
                    break;
                }
                m += 4;
            }
            if (m > cap) m = cap;
            if (m < rec_min) continue;
            const uint32_t d = ppo - s;
            if (d <= max_len && m + 1 >= d) {  // the growing match reaches the position
                lstop = s;
                lterm = true;
                break;
            }
            if (m > lm) {
                lm = m;
                ls = s;
            }
            if (m == cap) {  // cannot be exceeded: recorded (it beats whatever came before) and the search ends
                lstop = s;
                break;
            }
        }
        // lanes hold ascending ranges: everything behind the first stopping lane is never looked at
        const uint64_t stops = __ballot(lstop != kNoCandidate);
        const uint32_t tl = stops ? (uint32_t)__ffsll((unsigned long long)stops) - 1u : 64u;
        const uint32_t em = lane <= tl ? lm : 0u;
        uint32_t wm = em;
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)wm, off);
            wm = o > wm ? o : wm;
        }
        if (wm > best) {  // strictly longer than everything farther away; ties inside the round go to the lowest lane
            const uint64_t who = __ballot(em == wm);
            const int src = __ffsll((unsigned long long)who) - 1;
            best = wm;
            bestd = ppo - (uint32_t)__shfl((int)ls, src);
        }
        if (stops) {
            if (__shfl((int)lterm, (int)tl)) termd = ppo - (uint32_t)__shfl((int)lstop, (int)tl);
            break;
        }
    }
}

}  // namespace zg
