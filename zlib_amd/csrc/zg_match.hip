// LZ77 match finding for gfx950: a forward-chain 3-gram index (k_index) and an
// exhaustive, position-parallel longest-match search over LDS-staged windows
// (k_match).  Together they reproduce, for EVERY position of a chunk, the
// result of the candidate loop of LZ77Compressor::compress
// (reference src/lz77.rs:156-190):
//
//   candidates s run upward from max(0, pos - window) while s + min_len < pos;
//   M(s) = common prefix of data[s..] and data[pos..], capped at
//          min(chunk_end - pos, max_len - 1);
//   a candidate with M >= min_len and s + M + 1 >= pos ends the whole search
//          (the reference's loop guard fails while the match is still growing);
//   otherwise it is recorded iff M >= min_len and M > best (strict: ties keep
//          the farthest candidate).
//
// Only positions sharing the first three bytes can reach M >= min_len (>= 3),
// so walking the per-hash forward chain from the farthest in-window member
// visits every relevant candidate in the reference's order.
#include "zg_internal.h"

#ifndef ZG_MATCH_STEPS
#define ZG_MATCH_STEPS 8
#endif

namespace zg {

// ---------------------------------------------------------------------------
// k_index: one wave per segment walks positions in batches of 64, keeping per
// hash bucket the newest member (head) and the farthest in-window member
// (tail), plus a 64 Ki-entry ring of forward links.  Outputs, per position p:
//   far[p]  = p - (smallest bucket member >= p - window), 0 if none
//   next[s] = distance from s to the next bucket member, 0 if none within window
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t hash3(uint32_t tri) { return (tri * 0x9E3779B1u) >> (32 - kIdxHashBits); }

constexpr uint32_t kNone = 0xFFFFFFFFu;

__global__ __launch_bounds__(64) void k_index(const uint8_t *__restrict__ data, uint64_t n_view, uint64_t r0,
                                               uint64_t r1, uint64_t far_lo, uint32_t window,
                                               uint16_t *__restrict__ far, uint16_t *__restrict__ next,
                                               uint64_t base) {
    __shared__ uint32_t s_head[kIdxBuckets];    // newest member + 1 (segment-relative), 0 = empty
    __shared__ uint16_t s_tail[kIdxBuckets];    // low 16 bits of the farthest in-window member
    __shared__ uint16_t s_ring[65536];          // forward link of position (q & 0xFFFF)

    const uint32_t lane = threadIdx.x;
    const uint64_t a = r0 + (uint64_t)blockIdx.x * kIdxSegment;  // first owned position
    const uint64_t b = (a + kIdxSegment < r1) ? a + kIdxSegment : r1;
    const uint64_t a0 = a > window ? a - window : 0;  // warm-up start
    if (a >= r1) return;

    for (uint32_t i = lane; i < kIdxBuckets; i += 64) s_head[i] = 0;
    for (uint32_t i = lane; i < 65536 / 2; i += 64) reinterpret_cast<uint32_t *>(s_ring)[i] = 0;
    __syncthreads();

    const uint32_t q_end = (uint32_t)(b - a0);
    const uint32_t q_main = (uint32_t)(a - a0);
    for (uint32_t qb = 0; qb < q_end; qb += 64) {
        const uint32_t q = qb + lane;
        const uint64_t p = a0 + q;
        const bool inseg = q < q_end;
        const bool valid = inseg && (p + 3 <= n_view);
        uint32_t h = 0;
        if (valid) {
            uint32_t tri = (uint32_t)data[p] | ((uint32_t)data[p + 1] << 8) | ((uint32_t)data[p + 2] << 16);
            h = hash3(tri);
        }
        // lanes of this batch that share my bucket
        uint64_t mygrp = 0;
        uint64_t todo = __ballot(valid);
        while (todo) {
            const int leader = __ffsll((unsigned long long)todo) - 1;
            const uint32_t hl = __shfl(h, leader);
            const bool same = valid && (h == hl);
            const uint64_t grp = __ballot(same);
            if (same) mygrp = grp;
            todo &= ~grp;
        }
        uint32_t pp = kNone;  // previous bucket member (segment-relative)
        uint32_t f = kNone;   // farthest in-window bucket member
        bool is_last = false;
        if (valid) {
            const uint64_t below = mygrp & ((1ull << lane) - 1ull);
            const int prevlane = below ? 63 - __clzll((long long)below) : -1;
            const int firstlane = __ffsll((unsigned long long)mygrp) - 1;
            is_last = (mygrp >> lane) == 1ull;
            const uint32_t hq1 = s_head[h];
            const uint32_t hq = hq1 - 1u;
            const bool has_old = hq1 != 0u && (q - hq) <= window;
            if (prevlane >= 0)
                pp = qb + (uint32_t)prevlane;
            else if (has_old)
                pp = hq;
            if (has_old) {
                uint32_t t = hq - ((hq - (uint32_t)s_tail[h]) & 0xFFFFu);
                const uint32_t lo = q > window ? q - window : 0u;
                // every link on the way from tail to head was written in an earlier batch
                for (int guard = 0; t < lo && guard < 65536; ++guard) {
                    const uint32_t dl = s_ring[t & 0xFFFFu];
                    if (dl == 0) break;
                    t += dl;
                }
                f = t;
            } else if (prevlane >= 0) {
                f = qb + (uint32_t)firstlane;
            }
        }
        // all reads of this batch are done (one wave, program order); now publish
        if (valid) s_ring[q & 0xFFFFu] = 0;
        if (valid && pp != kNone) s_ring[pp & 0xFFFFu] = (uint16_t)(q - pp);
        if (valid && is_last) {
            s_head[h] = q + 1u;
            s_tail[h] = (uint16_t)((f != kNone ? f : q) & 0xFFFFu);
        }
        if (inseg && q >= q_main) {
            if (valid && pp != kNone && a0 + pp >= base) next[(a0 + pp) - base] = (uint16_t)(q - pp);
            if (p >= far_lo) far[p - base] = (valid && f != kNone) ? (uint16_t)(q - f) : (uint16_t)0;
        }
    }
}

// k_index_x: the same index, with the per-batch grouping done by the LDS itself and sixteen waves per segment.
// One ds_wrxchg on the bucket head hands every lane the bucket member inserted just before it, including
// lanes of the same batch, PROVIDED the LDS serialises same-address lanes of one instruction in ascending lane
// order.  gfx950 does (probed over 6.5*10^8 colliding exchanges, tools/probe/probe_exch.hip); the kernel still
// checks every returned value (a member handed out of order is >= the lane's own position) and raises bit 1 of
// *status, upon which the host redoes the call with k_index.
// The waves of a workgroup share the tables but own disjoint hash buckets (bucket & 15 == wave), hence also
// disjoint ring slots, so they never synchronise: each wave scans all positions of the segment, gathers the
// next (up to) 64 positions of its own buckets into a small staging area, and inserts them as one batch.
#ifndef ZG_IDX_WAVES
#define ZG_IDX_WAVES 8
#endif
constexpr uint32_t kIdxWaves = ZG_IDX_WAVES;
constexpr uint32_t kIdxSyncEvery = 8192;  // positions between two tail sweeps / wave barriers

__global__ __launch_bounds__(64 * kIdxWaves) void k_index_x(const uint8_t *__restrict__ data, uint64_t n_view,
                                                             uint64_t r0, uint64_t r1, uint64_t far_lo,
                                                             uint32_t window, uint16_t *__restrict__ far,
                                                             uint16_t *__restrict__ next, uint64_t base,
                                                             uint32_t *status) {
    __shared__ uint32_t s_head[kIdxBuckets];
    __shared__ uint16_t s_tail[kIdxBuckets];
    __shared__ uint16_t s_ring[65536];
    __shared__ uint32_t s_stq[kIdxWaves][64];
    __shared__ uint16_t s_sth[kIdxWaves][64];

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t a = r0 + (uint64_t)blockIdx.x * kIdxSegment;
    const uint64_t b = (a + kIdxSegment < r1) ? a + kIdxSegment : r1;
    const uint64_t a0 = a > window ? a - window : 0;
    if (a >= r1) return;

    for (uint32_t i = tid; i < kIdxBuckets; i += 64 * kIdxWaves) s_head[i] = 0;
    for (uint32_t i = tid; i < 65536 / 2; i += 64 * kIdxWaves) reinterpret_cast<uint32_t *>(s_ring)[i] = 0;
    __syncthreads();

    const uint32_t q_end = (uint32_t)(b - a0);
    const uint32_t q_main = (uint32_t)(a - a0);
    bool violated = false;

    // inserts the first `cnt` staged positions of this wave (ascending, all in buckets this wave owns)
    auto insert_batch = [&](uint32_t cnt) {
        const bool valid = lane < cnt;
        const uint32_t q = valid ? s_stq[wave][lane] : 0u;
        const uint32_t h = valid ? (uint32_t)s_sth[wave][lane] : 0u;
        uint32_t t16 = 0, head1 = 0, old1 = 0;
        if (valid) {
            t16 = s_tail[h];    // tail and head as the previous batches left them
            head1 = s_head[h];
            old1 = atomicExch(&s_head[h], q + 1u);  // member inserted just before me (+1), batch mates included
            violated |= (old1 != 0u && old1 - 1u >= q);
        }
        const uint32_t hq = head1 - 1u;
        const bool first_of_group = valid && old1 == head1;
        const bool has_old = valid && head1 != 0u && (q - hq) <= window;
        uint32_t pp = kNone;
        if (valid && !first_of_group)
            pp = old1 - 1u;
        else if (has_old)
            pp = hq;
        // the first lane of every bucket group publishes itself; the group's last lane rewrites the tail below
        if (first_of_group) s_tail[h] = (uint16_t)(q & 0xFFFFu);
        if (valid) s_ring[q & 0xFFFFu] = 0;
        if (valid && pp != kNone) s_ring[pp & 0xFFFFu] = (uint16_t)(q - pp);
        uint32_t f = kNone;
        bool is_last = false;
        if (valid) {
            if (has_old) {
                uint32_t t = hq - ((hq - t16) & 0xFFFFu);
                const uint32_t lo = q > window ? q - window : 0u;
                for (int guard = 0; t < lo && guard < 65536; ++guard) {  // may run on into this batch's links
                    const uint32_t dl = s_ring[t & 0xFFFFu];
                    if (dl == 0) break;
                    t += dl;
                }
                f = t;
            } else if (!first_of_group) {
                const uint32_t gf = s_tail[h];  // the group's first lane, a few hundred positions back at most
                f = q - ((q - gf) & 0xFFFFu);
            }
            is_last = s_head[h] == q + 1u;
        }
        if (valid && is_last) s_tail[h] = (uint16_t)((f != kNone ? f : q) & 0xFFFFu);
        if (valid && q >= q_main) {
            const uint64_t p = a0 + q;
            if (pp != kNone && a0 + pp >= base) next[(a0 + pp) - base] = (uint16_t)(q - pp);
            if (p >= far_lo) far[p - base] = (f != kNone) ? (uint16_t)(q - f) : (uint16_t)0;
        }
    };

    auto load_tri = [&](uint32_t q) -> uint32_t {  // three bytes at segment position q, ~0 if none
        const uint64_t p = a0 + q;
        if (q < q_end && p + 4 <= n_view) {
            uint32_t v;
            __builtin_memcpy(&v, data + p, 4);  // one unaligned dword load
            return v & 0x00FFFFFFu;
        }
        if (q < q_end && p + 3 <= n_view)
            return (uint32_t)data[p] | ((uint32_t)data[p + 1] << 8) | ((uint32_t)data[p + 2] << 16);
        return 0xFFFFFFFFu;
    };
    uint32_t count = 0;    // staged positions (wave uniform)
    uint32_t first_q = 0;  // segment position of the first staged one
    const uint32_t span_limit = window - 64;  // batch mates must lie within each other's window (window >= 2048 here)
    constexpr int kAhead = 8;  // raw batches whose bytes are in flight
    uint32_t tri_cur[kAhead], tri_nxt[kAhead];
#pragma unroll
    for (int k = 0; k < kAhead; k++) tri_cur[k] = load_tri((uint32_t)k * 64 + lane);
    for (uint32_t qg = 0; qg < q_end; qg += 64 * kAhead) {
        if (qg != 0 && (qg & (kIdxSyncEvery - 1)) == 0) {
            // Ring slots are shared by positions 65536 apart, which different waves may own.  Keeping every
            // bucket's tail within window + kIdxSyncEvery of the scan point, and the waves within kIdxSyncEvery
            // of each other, keeps a slot's old owner done with it before the new owner writes it
            // (window + 2 * kIdxSyncEvery + span of a staged batch < 65536).
            if (count) {
                insert_batch(count);
                count = 0;
            }
            __syncthreads();
            for (uint32_t k = 0; k < kIdxBuckets / (64 * kIdxWaves); k++) {  // this wave's buckets
                const uint32_t hb = ((k * 64 + lane) * kIdxWaves) | wave;
                const uint32_t head1 = s_head[hb];
                if (head1 != 0u) {
                    const uint32_t hq = head1 - 1u;
                    const uint32_t lo = qg > window ? qg - window : 0u;
                    if (hq < lo) {
                        s_head[hb] = 0;  // every member has left the window
                    } else {
                        uint32_t t = hq - ((hq - (uint32_t)s_tail[hb]) & 0xFFFFu);
                        for (int guard = 0; t < lo && guard < 65536; ++guard) {
                            const uint32_t dl = s_ring[t & 0xFFFFu];
                            if (dl == 0) break;
                            t += dl;
                        }
                        s_tail[hb] = (uint16_t)(t & 0xFFFFu);
                    }
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int k = 0; k < kAhead; k++) tri_nxt[k] = load_tri(qg + (uint32_t)(kAhead + k) * 64 + lane);
#pragma unroll
        for (int k = 0; k < kAhead; k++) {
            const uint32_t qb = qg + (uint32_t)k * 64;
            const uint32_t q = qb + lane;
            const uint32_t tri = tri_cur[k];
            const bool valid = tri != 0xFFFFFFFFu;
            const uint32_t h = valid ? hash3(tri) : 0u;
            const bool mine = valid && (h & (kIdxWaves - 1)) == wave;
            const uint64_t mask = __ballot(mine);
            const uint32_t n = (uint32_t)__popcll(mask);
            if (count != 0 && (count + n > 64 || qb + 63 - first_q > span_limit)) {
                insert_batch(count);
                count = 0;
            }
            if (count == 0 && n != 0) first_q = qb + (uint32_t)(__ffsll((unsigned long long)mask) - 1);
            if (mine) {
                const uint32_t slot = count + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
                s_stq[wave][slot] = q;
                s_sth[wave][slot] = (uint16_t)h;
            }
            count += n;
        }
#pragma unroll
        for (int k = 0; k < kAhead; k++) tri_cur[k] = tri_nxt[k];
    }
    if (count) insert_batch(count);
    // positions of the owned range that were never inserted (no three bytes left) have no far candidate
    for (uint64_t p = (a > far_lo ? a : far_lo) + tid; p < b; p += 64 * kIdxWaves)
        if (p + 3 > n_view) far[p - base] = 0;
    if (__any(violated) && lane == 0) atomicOr(status, 2u);
}

// k_index_brute: the same two arrays for small windows (LZ77Compressor's default is 144), one thread per
// position, by direct scan of the window on either side.
__global__ __launch_bounds__(256) void k_index_brute(const uint8_t *__restrict__ data, uint64_t n_view, uint64_t r0,
                                                      uint64_t r1, uint64_t far_lo, uint32_t window,
                                                      uint16_t *__restrict__ far, uint16_t *__restrict__ next,
                                                      uint64_t base) {
    const uint64_t p = r0 + (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= r1) return;
    auto hash_at = [&](uint64_t x) -> uint32_t {
        return hash3((uint32_t)data[x] | ((uint32_t)data[x + 1] << 8) | ((uint32_t)data[x + 2] << 16));
    };
    uint32_t fd = 0, nd = 0;
    if (p + 3 <= n_view) {
        const uint32_t h = hash_at(p);
        const uint64_t lo = p > window ? p - window : 0;
        for (uint64_t s = lo; s < p; s++)
            if (hash_at(s) == h) {
                fd = (uint32_t)(p - s);
                break;
            }
        for (uint64_t q = p + 1; q <= p + window && q + 3 <= n_view; q++)
            if (hash_at(q) == h) {
                nd = (uint32_t)(q - p);
                break;
            }
    }
    next[p - base] = (uint16_t)nd;
    if (p >= far_lo) far[p - base] = (uint16_t)fd;
}

hipError_t launch_index(const uint8_t *data, uint64_t n_view, uint64_t r0, uint64_t r1, uint64_t far_lo,
                        uint32_t window, uint16_t *far, uint16_t *next, uint64_t base, uint32_t *status, bool safe,
                        hipStream_t s) {
    if (r1 <= r0) return hipSuccess;
    const uint64_t nseg = (r1 - r0 + kIdxSegment - 1) / kIdxSegment;
    if (window < 2048)
        hipLaunchKernelGGL(k_index_brute, dim3((uint32_t)((r1 - r0 + 255) / 256)), dim3(256), 0, s, data, n_view, r0,
                           r1, far_lo, window, far, next, base);
    else if (safe)
        hipLaunchKernelGGL(k_index, dim3((uint32_t)nseg), dim3(64), 0, s, data, n_view, r0, r1, far_lo, window, far,
                           next, base);
    else
        hipLaunchKernelGGL(k_index_x, dim3((uint32_t)nseg), dim3(64 * kIdxWaves), 0, s, data, n_view, r0, r1, far_lo,
                           window, far, next, base, status);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// k_match: one workgroup per tile of kMatchTile positions.  The tile's bytes
// plus the preceding window and the forward links of that range are staged in
// LDS; each thread then resolves positions tid, tid + 1024, ... of the tile.
// ---------------------------------------------------------------------------
constexpr uint32_t kDataLds = kMaxWindow + kMatchTile + 16 + 272;  // 16: alignment slack, 272: look-ahead + pad
constexpr uint32_t kNextLds = kMaxWindow + kMatchTile + 16;

// four bytes at an arbitrary LDS byte offset, from two aligned dwords
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

__global__ __launch_bounds__(kMatchThreads) void k_match(const uint8_t *__restrict__ data, uint64_t first,
                                                          uint64_t end, uint64_t chunk_bytes, uint32_t c0,
                                                          uint32_t tiles_per_chunk, MatchParams mp,
                                                          const uint16_t *__restrict__ far,
                                                          const uint16_t *__restrict__ next, uint64_t base,
                                                          uint32_t *__restrict__ match) {
    // one LDS image: bytes first (so that byte reads need no base add), links behind them
    __shared__ __attribute__((aligned(16))) uint8_t s_mem[kDataLds + 2 * kNextLds];
    __shared__ uint32_t s_counter;  // next unclaimed position of the tile
    uint8_t *const s_data = s_mem;
    uint16_t *const s_next = reinterpret_cast<uint16_t *>(s_mem + kDataLds);

    const uint32_t chunk = c0 + blockIdx.x / tiles_per_chunk;
    const uint32_t tile = blockIdx.x % tiles_per_chunk;
    const uint64_t cs = first + (uint64_t)chunk * chunk_bytes;
    const uint64_t ce = (cs + chunk_bytes < end) ? cs + chunk_bytes : end;
    const uint64_t t0 = cs + (uint64_t)tile * kMatchTile;
    if (t0 >= ce) return;
    const uint64_t t1 = (t0 + kMatchTile < ce) ? t0 + kMatchTile : ce;
    const uint64_t w0 = (t0 > mp.window ? t0 - mp.window : 0) & ~(uint64_t)15;  // LDS origin, 16-byte aligned
    const uint64_t w1 = (t1 + mp.max_len < ce) ? t1 + mp.max_len : ce;          // bytes needed: [w0, w1)
    const uint32_t nbytes = (uint32_t)(w1 - w0);
    const uint32_t nlinks = (uint32_t)(t1 - w0);
    const uint32_t tid = threadIdx.x;
    if (tid == 0) s_counter = 0;

    // stage bytes: 16-byte vectors when the source is aligned, bytes otherwise
    if (((reinterpret_cast<uintptr_t>(data) + w0) & 15) == 0) {
        const uint4 *src = reinterpret_cast<const uint4 *>(data + w0);
        uint4 *dst = reinterpret_cast<uint4 *>(s_data);
        const uint32_t nv = nbytes >> 4;
        for (uint32_t i = tid; i < nv; i += kMatchThreads) dst[i] = src[i];
        for (uint32_t i = (nv << 4) + tid; i < nbytes; i += kMatchThreads) s_data[i] = data[w0 + i];
    } else {
        for (uint32_t i = tid; i < nbytes; i += kMatchThreads) s_data[i] = data[w0 + i];
    }
    for (uint32_t i = nbytes + tid; i < nbytes + 16 && i < kDataLds; i += kMatchThreads) s_data[i] = 0;
    {
        const uint16_t *src = next + (w0 - base);
        if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
            const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
            uint4 *d4 = reinterpret_cast<uint4 *>(s_next);
            const uint32_t nv = nlinks >> 3;
            for (uint32_t i = tid; i < nv; i += kMatchThreads) d4[i] = s4[i];
            for (uint32_t i = (nv << 3) + tid; i < nlinks; i += kMatchThreads) s_next[i] = src[i];
        } else {
            for (uint32_t i = tid; i < nlinks; i += kMatchThreads) s_next[i] = src[i];
        }
    }
    __syncthreads();

    // Search.  Chain lengths are very uneven (Zipf-like), so positions are not bound to threads: a lane claims
    // the next unresolved position of the tile from an LDS counter whenever it finishes one.  Every iteration
    // all lanes run kHeadSteps branch-free "head" steps: one four-byte probe of the candidate and its forward
    // link.  The probe looks at the only place that can still matter: for a candidate too far away to end the
    // search, the four bytes ending at offset `best` (they must all agree for it to beat the best length so
    // far); for a candidate within max_len of the position, or while nothing has been found, the first three
    // bytes.  The two rarer kinds of work -- measuring a candidate that survived the probe, and retiring a
    // position / claiming the next one -- are deferred until enough lanes of the wave wait for them, so that
    // each of the three code paths runs with most of its lanes active.
    const uint32_t min_len = mp.min_len, max_len = mp.max_len;
    const uint32_t tn = (uint32_t)(t1 - t0);
    const uint32_t pbase = (uint32_t)(t0 - w0);  // LDS offset of tile position 0
    const uint16_t *far_t = far + (t0 - base);
    uint32_t *match_t = match + (t0 - base);
    const uint64_t tail_room = ce - t0;  // bytes from tile position 0 to the end of the chunk

    enum : uint32_t { HEAD = 0, EVAL = 1, FETCH = 2, DONE = 3 };
    constexpr int kHeadSteps = ZG_MATCH_STEPS;  // head steps between checks of the deferred work
    constexpr uint32_t kEvalBatch = 16, kFetchBatch = 8;
    uint32_t mode = FETCH, idx = 0, po = pbase, so = pbase, cap = 0, best = 0, bestd = 0, p0 = 0, p1 = 0;
    uint32_t toff = 0, pt = 0, tmask = 0x00FFFFFFu;  // far-candidate probe: offset, expected bytes, byte mask
    bool has_pos = false;  // a position is loaded and its result is not yet written
    uint32_t nxt_idx = atomicAdd(&s_counter, 1u);
    uint32_t nxt_far = nxt_idx < tn ? far_t[nxt_idx] : 0u;

    for (;;) {
#pragma unroll
        for (int step = 0; step < kHeadSteps; step++) {
            const uint32_t d = po - so;
            const bool nearby = d <= max_len;  // may end the search: its first bytes decide
            const uint32_t off = nearby ? 0u : toff;
            const uint32_t want = nearby ? p0 : pt;
            const uint32_t msk = nearby ? 0x00FFFFFFu : tmask;
            const uint32_t w = lds_ld4(s_data, so + off);
            const uint32_t nx = s_next[so];
            const bool is_head = mode == HEAD;
            const bool guard_end = d <= min_len;  // loop guard: s + min_len < pos
            const bool survive = ((w ^ want) & msk) == 0;
            const bool to_adv = is_head & !guard_end & !survive;
            const bool to_fetch = is_head & (guard_end | (!survive & (nx == 0)));
            const bool to_eval = is_head & !guard_end & survive;
            so = to_adv ? so + nx : so;
            mode = to_fetch ? (uint32_t)FETCH : (to_eval ? (uint32_t)EVAL : mode);
        }
        uint32_t n_head = (uint32_t)__popcll(__ballot(mode == HEAD));
        uint32_t n_eval = (uint32_t)__popcll(__ballot(mode == EVAL));
        // ---- measure surviving candidates ----
        if (n_eval >= kEvalBatch || (n_eval != 0 && n_head == 0)) {
            if (mode == EVAL) {
                uint32_t m;
                const uint32_t x0 = lds_ld4(s_data, so) ^ p0;
                if (x0) {
                    m = (uint32_t)__builtin_ctz(x0) >> 3;
                } else {
                    const uint32_t x1 = lds_ld4(s_data, so + 4) ^ p1;
                    if (x1) {
                        m = 4 + ((uint32_t)__builtin_ctz(x1) >> 3);
                    } else {
                        m = 8;
                        while (m < cap) {
                            const uint32_t x = lds_ld4(s_data, so + m) ^ lds_ld4(s_data, po + m);
                            if (x) {
                                m += (uint32_t)__builtin_ctz(x) >> 3;
                                break;
                            }
                            m += 4;
                        }
                    }
                }
                if (m > cap) m = cap;
                const uint32_t d = po - so;
                bool finish = false;
                if (m >= min_len) {
                    if (d <= max_len && m + 1 >= d) {  // the growing match reached pos: search over
                        finish = true;
                    } else if (m > best) {
                        best = m;
                        bestd = d;
                        toff = m - 3;
                        pt = lds_ld4(s_data, po + toff);
                        tmask = 0xFFFFFFFFu;
                        finish = m == cap;  // cannot be exceeded
                    }
                }
                const uint32_t nx = s_next[so];
                so += nx;
                mode = (finish || nx == 0) ? FETCH : HEAD;
            }
            n_head = (uint32_t)__popcll(__ballot(mode == HEAD));
            n_eval = 0;
        }
        // ---- retire finished positions and claim new ones ----
        const uint32_t n_fetch = (uint32_t)__popcll(__ballot(mode == FETCH));
        if (n_fetch >= kFetchBatch || (n_fetch != 0 && n_head == 0 && n_eval == 0)) {
            if (mode == FETCH) {
                if (has_pos) match_t[idx] = best ? ((best << 16) | bestd) : 0u;
                has_pos = false;
                if (nxt_idx >= tn) {
                    mode = DONE;
                } else {
                    idx = nxt_idx;
                    const uint32_t fd = nxt_far;
                    nxt_idx = atomicAdd(&s_counter, 1u);
                    nxt_far = nxt_idx < tn ? far_t[nxt_idx] : 0u;
                    const uint64_t remain = tail_room - idx;
                    cap = remain < (uint64_t)(max_len - 1) ? (uint32_t)remain : max_len - 1;
                    best = 0;
                    bestd = 0;
                    has_pos = true;
                    // the reference searches only while pos < len - min_len, and records nothing below min_len
                    if (fd != 0 && remain > min_len && cap >= min_len) {
                        po = pbase + idx;
                        so = po - fd;
                        p0 = lds_ld4(s_data, po);
                        p1 = lds_ld4(s_data, po + 4);
                        toff = 0;
                        pt = p0;
                        tmask = 0x00FFFFFFu;
                        mode = HEAD;
                    }
                }
            }
        } else if (n_fetch == 0 && n_head == 0 && n_eval == 0) {
            break;  // every lane is DONE
        }
    }
}

hipError_t launch_match(const ChunkGrid &g, uint32_t c0, uint32_t c1, MatchParams mp, const uint16_t *far,
                        const uint16_t *next, uint64_t base, uint32_t *match, hipStream_t s) {
    if (c1 <= c0) return hipSuccess;
    const uint32_t tpc = (uint32_t)((g.chunk_bytes + kMatchTile - 1) / kMatchTile);
    const uint64_t nblocks = (uint64_t)(c1 - c0) * tpc;
    hipLaunchKernelGGL(k_match, dim3((uint32_t)nblocks), dim3(kMatchThreads), 0, s, g.data, g.first, g.end,
                       g.chunk_bytes, c0, tpc, mp, far, next, base, match);
    return hipGetLastError();
}

}  // namespace zg
