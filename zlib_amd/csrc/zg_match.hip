// LZ77 match finding for gfx950: forward-chain K-gram indexes (K = 5, 4 and 3)
// and an exhaustive, position-parallel longest-match search, in two levels.  Together they reproduce, for EVERY position of a
// chunk, the result of the candidate loop of LZ77Compressor::compress
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
// Level 1 (k_match) walks the chain of positions sharing the hash of the first
// FIVE bytes: it sees every candidate with M >= 5 in the reference's order, so
// it settles every position whose answer is five bytes or longer, and finds the
// search-ending candidate T if that one is five bytes or longer.  A search-ending
// candidate shorter than that sits at distance 4 or 5 and cannot precede a
// recorded candidate of five bytes or more (those lie at distance >= 7), so
// level 1 never needs it.
// Level 2 (k_match_lo, only when min_len <= 4) takes the positions level 1 left
// open.  All their candidates below T match fewer than five bytes, so the
// answer is the farthest candidate with four common bytes, else the farthest
// with three: the first true member of the FOUR-byte chain, else of the
// THREE-byte chain.
#include "zg_internal.h"
#include "zg_match_util.h"

#include <cstdlib>
#include <type_traits>

#ifndef ZG_MATCH_STEPS
#define ZG_MATCH_STEPS 9
#endif
#ifndef ZG_MATCH_STEPS_REC
#define ZG_MATCH_STEPS_REC 8
#endif
#ifndef ZG_MATCH_COOP_LANES
#define ZG_MATCH_COOP_LANES 4  // a draining wave with at most this many walking lanes finishes long walks together; 0: never
#endif
#ifndef ZG_MATCH_COOP_GAIN
#define ZG_MATCH_COOP_GAIN 4   // ... by this factor (the comparison stalls the wave's other walks and measures every true candidate)
#endif
#ifndef ZG_MATCH_COOP_BASE
#define ZG_MATCH_COOP_BASE 3   // ... those expected to take more than this many trips plus one per 2 KiB left
#endif
#ifndef ZG_MATCH_LATE
#define ZG_MATCH_LATE 0  // positions at the end of a 16 Ki tile that are claimed without holding the claim ahead
#endif
#ifndef ZG_MATCH_NEAR
#define ZG_MATCH_NEAR 48  // candidates this close to the position are measured without a probe
#endif

// -DZG_MATCH_DEBUG=1: phase times of k_match summed over its workgroups; =2: event counts as well (their atomics
// slow the kernel down many times over).  tools/debug_match.py reads them.
#ifdef ZG_MATCH_DEBUG
__device__ unsigned long long g_mdbg[32];
extern "C" int zg_match_debug_counters(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mdbg), sizeof(g_mdbg));
}
#define MDBG(...) __VA_ARGS__
#if ZG_MATCH_DEBUG >= 2
#define MDBG2(...) __VA_ARGS__
#else
#define MDBG2(...)
#endif
#else
#define MDBG(...)
#define MDBG2(...)
#endif

namespace zg {

// ---------------------------------------------------------------------------
// Index.  Outputs, per position p (members of a bucket = positions whose first
// K bytes hash alike):
//   far[p]  = p - (smallest bucket member >= p - window), 0 if none
//   next[s] = distance from s to the next bucket member, 0 if none within window
// ---------------------------------------------------------------------------
// 32-bit mix of the first K bytes; a table of 2^b buckets takes its top b bits
template <int K>
__device__ __forceinline__ uint32_t key_hash(uint32_t lo, uint32_t hi);
template <>
__device__ __forceinline__ uint32_t key_hash<3>(uint32_t lo, uint32_t) {
    return (lo & 0x00FFFFFFu) * 0x9E3779B1u;
}
template <>
__device__ __forceinline__ uint32_t key_hash<4>(uint32_t lo, uint32_t) {
    return lo * 0x9E3779B1u;
}
template <>
__device__ __forceinline__ uint32_t key_hash<5>(uint32_t lo, uint32_t hi) {
    return lo * 0x9E3779B1u + __umul24(hi & 0xFFu, 0x85EBCBu);  // (a 24-bit product issues at the full rate)
}

// the first K bytes at view position p (p + K <= n_view): low four in lo, the fifth in hi
template <int K>
__device__ __forceinline__ void load_key(const uint8_t *__restrict__ data, uint64_t p, uint64_t n_view, uint32_t &lo,
                                         uint32_t &hi) {
    if (p + 4 <= n_view) {
        __builtin_memcpy(&lo, data + p, 4);  // one unaligned dword load
    } else {
        lo = (uint32_t)data[p] | ((uint32_t)data[p + 1] << 8) | ((uint32_t)data[p + 2] << 16);
    }
    hi = (K > 4) ? (uint32_t)data[p + 4] : 0u;
}

// eight (K = 5) or four (K = 3) bytes loaded at a position, or at the last offset of the view that allows such a
// load (the bytes are shifted into place afterwards): loads without branches stay in flight across iterations
template <int K>
struct RawKey {
    uint32_t a, b;
};
template <int K>
__device__ __forceinline__ uint32_t hash_at(const uint8_t *__restrict__ data, uint64_t p, uint64_t n_view) {
    uint32_t lo, hi;
    load_key<K>(data, p, n_view, lo, hi);
    return key_hash<K>(lo, hi);
}

constexpr uint32_t kNone = 0xFFFFFFFFu;



// k_index: the conservative variant.  One wave per segment walks positions in batches of 64, keeping per
// bucket the newest member (head) and the farthest in-window member (tail), plus a 64 Ki-entry ring of forward
// links; lanes of a batch that share a bucket are grouped with ballots.
template <int K>
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
        const bool valid = inseg && (p + K <= n_view);
        const uint32_t h = valid ? hash_at<K>(data, p, n_view) >> (32 - kIdxHashBits) : 0u;
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

// k_index_r: the fast variant.  One wave per segment walks its positions DOWNWARD, with a single LDS table and no
// walking along chains:
//   head[h]  the lowest member inserted so far, i.e. the nearest member above the position being inserted.
//            One returning LDS atomic on it hands every lane its successor, lanes of the same instruction
//            included, PROVIDED the LDS serves same-address lanes of one instruction in ascending lane order.
//            gfx950 does (probed over 10^9 colliding operations: tools/probe/probe_exch.hip for the exchange,
//            tools/probe/probe_max.hip for the maximum with reads among the writes); the kernel still checks every
//            returned value (a successor handed out of order is not above the lane's own position) and raises
//            bit 1 of *status, upon which the host redoes the call with k_index.
// The successor gives next[] of the lane's own position: a contiguous store.  far[p] is read off the same
// table when the walk has come down to p - window: the look-up sits right behind the insert of position
// p - window in the same instruction, so the head it sees is the lowest member at or above p - window.
// Positions less than a window above the first position of their slot
// (the start of the data; the first byte of an isolated record) are answered in one sweep when the walk
// reaches that first position.  Slots start on the grid of batch groups, so a group never straddles two.
// A segment walks `window` positions above and below the seg_len ones it owns; the host picks seg_len so that
// the waves of a launch fill the machine a whole number of times.
#ifndef ZG_IDX_GROUP
#define ZG_IDX_GROUP 16
#endif
constexpr uint32_t kIdxGroup = ZG_IDX_GROUP;         // batches of 64 positions handled together
constexpr uint32_t kIdxSpan = 64 * kIdxGroup;
#ifndef ZG_IDX_CAP_KB
#define ZG_IDX_CAP_KB 1024
#endif
constexpr uint32_t kIdxSegMin = 16u << 10, kIdxSegCap = ZG_IDX_CAP_KB << 10;  // positions owned by one k_index_r wave
#ifndef ZG_IDX_BITS
#define ZG_IDX_BITS 13
#endif
constexpr uint32_t kIdxRBits = ZG_IDX_BITS;
constexpr uint32_t kIdxRBuckets = 1u << kIdxRBits;

// -DZG_INDEX_DEBUG: begin and end (100 MHz ticks) of every wave of the last k_index_r launches, for
// tools/debug_index.py (how the segments of a launch fill the machine)
#ifdef ZG_INDEX_DEBUG
__device__ unsigned long long g_idbg[1 + 3 * 65536];
__device__ unsigned long long g_iph[8];  // cycles per phase of a group, summed over all waves; [7]: groups
extern "C" int zg_index_debug(unsigned long long *out, size_t n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_idbg), n * sizeof(unsigned long long));
}
extern "C" int zg_index_debug_reset() {
    const unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_iph), zero, sizeof zero);
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_idbg), zero, sizeof zero[0]);
}
extern "C" int zg_index_debug_phases(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_iph), 8 * sizeof(unsigned long long));
}
#define IDBG(...) __VA_ARGS__
#else
#define IDBG(...)
#endif

// LDS of one k_index_r wave (32-bit words): the table; and, for the kernel without slots, the bytes of the current
// group's two regions (its own positions, and the positions a window above whose far[] it answers; each kIdxSpan
// bytes, the key bytes behind the last one and alignment slack) and the group's results (next[] and far[], 16 bits
// each), which leave in 16-byte stores.
constexpr uint32_t kIdxStageBytes = kIdxSpan + 256;
constexpr uint32_t kIdxLdsWordsStream = kIdxRBuckets + 2 * kIdxStageBytes / 4 + 2 * kIdxSpan / 2;
constexpr uint32_t kIdxLdsWordsIso = kIdxRBuckets + kIdxStageBytes / 4;  // with slots: the table and one region of bytes

template <int K, bool kIso>
__device__ __forceinline__ void index_r_segment(uint32_t *s_mem, uint32_t seg, uint32_t seg_len,
                                                const uint8_t *__restrict__ data, uint64_t n_view, uint64_t r0,
                                                uint64_t r1, uint64_t far_lo, uint32_t window,
                                                uint16_t *__restrict__ far, uint16_t *__restrict__ next,
                                                uint64_t base, uint32_t *status, uint64_t iso_first,
                                                uint32_t iso_bytes_arg) {
    const uint32_t iso_bytes = kIso ? iso_bytes_arg : 0u;
    uint32_t *const s_head = s_mem;
    const uint32_t lane = threadIdx.x;
    const uint64_t a = r0 + (uint64_t)seg * seg_len;
    if (a >= r1) return;
    const uint64_t b = (a + seg_len < r1) ? a + seg_len : r1;
    const uint32_t reach = (iso_bytes && iso_bytes < window) ? iso_bytes : window;  // longest link that can matter
    const uint32_t rup = (reach + kIdxSpan - 1u) & ~(kIdxSpan - 1u);
    const uint64_t a0 = (a - r0 > rup) ? a - rup : r0;        // lowest position walked; on the group grid from r0
    const uint64_t top = (b + reach < r1) ? b + reach : r1;   // one past the highest position walked

    for (uint32_t i = lane; i < kIdxRBuckets; i += 64) s_head[i] = 0;
    __syncthreads();

    // everything below is relative to a0 and fits 32 bits
    const uint32_t q_top = (uint32_t)(top - a0);
    const uint32_t q_a = (uint32_t)(a - a0), q_b = (uint32_t)(b - a0);
    const uint32_t q_far = far_lo > a ? (far_lo < b ? (uint32_t)(far_lo - a0) : q_b) : q_a;  // first far[] stored
    const uint64_t room = n_view - a0;
    const uint32_t n_rel = room < 0x7FFFFFFFull ? (uint32_t)room : 0x7FFFFFFFu;  // bytes from a0 to the view's end
    // last offset a key load may start at; below zero when the view ends within a few bytes of a0 (n_view >= 64)
    const int32_t last = (int32_t)n_rel - (K > 4 ? 8 : 4);
    const uint8_t *dseg = data + a0;
    uint16_t *far_seg = far + (a0 - base);
    uint16_t *next_seg = next + (a0 - base);
    bool violated = false;
    IDBG(unsigned long long iph[8] = {0, 0, 0, 0, 0, 0, 0, 0};)

    // (offsets are taken from a base at or below every clamped offset, so that they are unsigned 32-bit values:
    // a scalar base plus a vector offset, without the 64-bit address arithmetic of a signed index)
    auto st16 = [](uint16_t *base_p, uint32_t i, uint32_t v) {  // base_p[i] = v with a 32-bit byte offset (i < 2^31)
        *reinterpret_cast<uint16_t *>(reinterpret_cast<uint8_t *>(base_p) + (i << 1)) = (uint16_t)v;
    };
    const int32_t low = last < 0 ? last : 0;
    const uint8_t *dlow = dseg + low;
    auto raw_at = [&](uint32_t q) __attribute__((always_inline)) -> RawKey<K> {  // branch-free: the offset is clamped into the view
        const int32_t qc = (int32_t)q < last ? (int32_t)q : last;
        const uint32_t off = (uint32_t)(qc - low);
        RawKey<K> r;
        __builtin_memcpy(&r.a, dlow + off, 4);
        r.b = 0;
        if (K > 4) __builtin_memcpy(&r.b, dlow + off + 4u, 4);
        return r;
    };
    auto hash_of = [&](RawKey<K> r, uint32_t q, bool exact) __attribute__((always_inline)) -> uint32_t {
        if (exact) return key_hash<K>(r.a, r.b) >> (32 - kIdxRBits);  // the load was not clamped
        const uint32_t sh = (int32_t)q > last ? ((uint32_t)((int32_t)q - last) & 7u) * 8u : 0u;
        if (K > 4) {
            const uint64_t v = (((uint64_t)r.b << 32) | r.a) >> sh;
            return key_hash<K>((uint32_t)v, (uint32_t)(v >> 32)) >> (32 - kIdxRBits);
        }
        return key_hash<K>(r.a >> (sh & 31u), 0u) >> (32 - kIdxRBits);
    };

    // The table holds kEnc - position of the lowest member so far (0 = empty; positions are relative to a0 and
    // small): the walk runs downward, so a newer member has the larger entry, and an LDS atomic MAX with return is
    // at once
    //   an insert   (operand = the position's entry, larger than what is there: the entry is replaced and the old
    //                one, the member's successor, comes back), and
    //   a look-up   (operand 0: nothing changes, the entry comes back).
    // One group = kIdxSpan positions, highest first, handled in one of two ways:
    //   insert mode (no far[] is answered by this group): 16 instructions, lane l of instruction k inserts position
    //       qg + kIdxSpan - 1 - (64 k + l);
    //   interleaved mode: 32 instructions; in instruction i the even lane 2j inserts position p = qg + kIdxSpan - 1 -
    //       (32 i + j) and the odd lane 2j + 1 looks up the bucket of position p + window.  The LDS serves the lanes
    //       of one instruction in ascending lane order (probed: tools/probe/probe_max.hip, reads among inserts
    //       included), so that look-up sees exactly the members at or above p -- the lowest of them is the farthest
    //       member p + window can reach, far[p + window], with no walk along the links of the batch (a look-up behind
    //       a whole batch of 64 inserts had to follow up to 63 of them: a fifth of the kernel's time on random
    //       bytes, three fifths on packed records).
    // None of these LDS operations waits for another, so the wave waits for the LDS once per group.
    constexpr uint32_t kEnc = 0x7FFFFFFFu;
    struct GroupKeys {
        RawKey<K> k[2 * kIdxGroup];  // insert mode uses the first half
    };
    const uint32_t pair = lane >> 1;
    const bool odd = (lane & 1u) != 0;
    const uint32_t odd01 = odd ? 1u : 0u;
    const uint32_t look_off = odd ? window : 0u;  // the key of an odd lane lies a window above the lane's position
    int32_t worst = 1;  // least (entry of the lane's position - entry handed back): 0 or below = handed out of order
    auto ins_pos = [&](uint32_t qg, int k) { return qg + (kIdxSpan - 1u) - ((uint32_t)k * 64 + lane); };
    auto il_pos = [&](uint32_t qg, int i) { return qg + (kIdxSpan - 1u) - ((uint32_t)i * 32 + pair); };
    // does any / every position looked up by the group at qg get its far[] from it?  (uniform)
    auto look_none = [&](uint32_t qg, uint32_t slot_end) {
        return qg + window >= q_b || qg + window >= slot_end || qg + window + kIdxSpan <= q_far;
    };
    auto look_all = [&](uint32_t qg, uint32_t slot_end) {
        return qg + window >= q_far && qg + window + kIdxSpan <= q_b && qg + window + kIdxSpan <= slot_end;
    };
    auto slot_end_of = [&](uint32_t qg, uint32_t relg) { return iso_bytes ? qg + (iso_bytes - relg) : 0xFFFFFFFFu; };
    // keys by one load per lane and instruction; qg may be "below zero" (wrapped): its loads are clamped into the
    // view and never used
    auto load_keys = [&](GroupKeys &buf, uint32_t qg, bool with_look) __attribute__((always_inline)) {
        if (with_look) {
#pragma unroll
            for (int i = 0; i < 2 * (int)kIdxGroup; i++) {
                const uint32_t q = il_pos(qg, i) + look_off;
                buf.k[i] = raw_at((int32_t)q > 0 ? q : 0u);
            }
        } else {
#pragma unroll
            for (int k = 0; k < (int)kIdxGroup; k++) {
                const uint32_t qi = ins_pos(qg, k);
                buf.k[k] = raw_at((int32_t)qi > 0 ? qi : 0u);
                // (every entry is written on both paths: the compiler would otherwise merge the two paths' last
                // stores into one through a pointer chosen at run time, and the keys would live in scratch memory)
                buf.k[kIdxGroup + k] = buf.k[k];
            }
        }
    };
    auto table_max = [&](uint32_t h, uint32_t operand) __attribute__((always_inline)) {
        return __hip_atomic_fetch_max(&s_head[h], operand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    // far[] of the positions less than a window above the first position of a slot (or of the data), answered when
    // the walk has reached that first position qg: every member of the slot is in the table and nothing below it
    auto sweep_slot = [&](uint32_t qg, uint32_t slot_end) __attribute__((always_inline)) {
        uint32_t pe = qg + reach < q_b ? qg + reach : q_b;
        pe = pe < slot_end ? pe : slot_end;
        const uint32_t lo = qg > q_far ? qg : q_far;
        for (uint32_t pb = qg + ((lo - qg) & ~63u); pb < pe; pb += kIdxSpan) {
            RawKey<K> r[kIdxGroup];
#pragma unroll
            for (int j = 0; j < (int)kIdxGroup; j++) r[j] = raw_at(pb + (uint32_t)j * 64 + lane);
#pragma unroll
            for (int j = 0; j < (int)kIdxGroup; j++) {
                const uint32_t p = pb + (uint32_t)j * 64 + lane;
                const uint32_t m1 = s_head[hash_of(r[j], p, false)];
                if (p >= q_far && p < pe) st16(far_seg, p, (p + K <= n_rel && m1) ? p - (kEnc - m1) : 0u);
            }
        }
    };

    // ---- the general forms: any group, lane by lane ----
    // cur holds the keys of the group at qg.  kNext / kLook: 0 = the group stores no next[] / answers no far[], 1 =
    // every lane does, 2 = decided lane by lane.  kFull: every key of the group lies well inside the view.
    auto group_ins = [&](const GroupKeys &cur, uint32_t qg, uint32_t slot_end, bool sweep, auto full_c,
                         auto next_c) __attribute__((always_inline)) {
        constexpr bool full = decltype(full_c)::value;
        constexpr int kNext = decltype(next_c)::value;
        uint32_t got[kIdxGroup];
#pragma unroll
        for (int k = 0; k < (int)kIdxGroup; k++) {
            const uint32_t qi = ins_pos(qg, k);
            const bool valid = full || (qi < q_top && qi + K <= n_rel);
            got[k] = table_max(hash_of(cur.k[k], qi, full), valid ? kEnc - qi : 0u);  // the member inserted just before
        }
#pragma unroll
        for (int k = 0; k < (int)kIdxGroup; k++) {
            const uint32_t qi = ins_pos(qg, k);
            const bool valid = full || (qi < q_top && qi + K <= n_rel);
            const uint32_t d = kEnc - qi - got[k];  // distance to the successor; huge if there is none
            const int32_t chk = valid ? (int32_t)d : 1;
            worst = chk < worst ? chk : worst;
            if (kNext) {
                // a link counts within the window and within the slot
                const uint32_t nd = (valid && d <= window && kEnc - got[k] < slot_end) ? d : 0u;
                if (kNext == 1 || (qi >= q_a && qi < q_b)) st16(next_seg, qi, nd);
            }
        }
        if (sweep) sweep_slot(qg, slot_end);
    };
    auto group_il = [&](const GroupKeys &cur, uint32_t qg, uint32_t slot_end, bool sweep, auto full_c, auto next_c,
                        auto look_c) __attribute__((always_inline)) {
        constexpr bool full = decltype(full_c)::value;
        constexpr int kNext = decltype(next_c)::value, kLook = decltype(look_c)::value;
        uint32_t got[2 * kIdxGroup];
        uint32_t vmask = 0;  // bit i: this lane's operation in instruction i is a valid insert / a wanted look-up
#pragma unroll
        for (int i = 0; i < 2 * (int)kIdxGroup; i++) {
            const uint32_t pos = il_pos(qg, i), kp = pos + look_off;
            const bool v_ins = full || (pos < q_top && pos + K <= n_rel);
            const bool v_look = kLook == 1 || (kp >= q_far && kp < q_b && kp < slot_end && (full || kp + K <= n_rel));
            const bool v = odd ? v_look : v_ins;
            vmask |= (v ? 1u : 0u) << i;
            got[i] = table_max(hash_of(cur.k[i], kp, full), (v && !odd) ? kEnc - pos : 0u);
        }
        // even lanes store next[p], odd lanes far[p + window], through a pointer of the lane's own
        uint16_t *const sp = odd ? far_seg + (il_pos(qg, 0) + window) : next_seg + il_pos(qg, 0);
#pragma unroll
        for (int i = 0; i < 2 * (int)kIdxGroup; i++) {
            const uint32_t pos = il_pos(qg, i);
            const bool v = (vmask >> i) & 1u;
            // Odd lanes take one more than the entry of their position, so that for both kinds of lane d is 1 or
            // more when all is in order: an insert gets a member above its position, a look-up one at or above it.
            // Even lanes: d = distance to the successor; odd: d = window + 1 - far.
            const uint32_t d = kEnc - pos + odd01 - got[i];
            const int32_t chk = v ? (int32_t)d : 1;
            worst = chk < worst ? chk : worst;
            const uint32_t nd = (v && d <= window && kEnc - got[i] < slot_end) ? d : 0u;
            const uint32_t fd = (v && d <= window + 1u) ? window + 1u - d : 0u;
            const bool st = odd ? v : (kNext == 1 || (kNext == 2 && pos >= q_a && pos < q_b));
            if (st) sp[-32 * i] = (uint16_t)(odd ? fd : nd);
        }
        if (sweep) sweep_slot(qg, slot_end);
    };
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    using C2 = std::integral_constant<int, 2>;
    auto general = [&](const GroupKeys &cur, uint32_t qg, uint32_t slot_end, bool sweep, bool full, bool n0, bool n1,
                       bool l0, bool l1) __attribute__((always_inline)) {
        if (kIso && full && n1 && l0)
            group_ins(cur, qg, slot_end, sweep, std::true_type{}, C1{});
        else if (kIso && full && n0 && l0)
            group_ins(cur, qg, slot_end, sweep, std::true_type{}, C0{});
        else if (l0)
            group_ins(cur, qg, slot_end, sweep, std::false_type{}, C2{});
        else
            group_il(cur, qg, slot_end, sweep, std::false_type{}, C2{}, C2{});
        (void)l1;
    };
    const uint32_t n_groups = (q_top + kIdxSpan - 1u) / kIdxSpan;
    // every key of the group at q (and of the positions it looks up) lies well inside the view
    auto is_full = [&](uint32_t q) { return q + kIdxSpan <= q_top && q + kIdxSpan + window + 72u <= n_rel; };

    if constexpr (kIso) {
        // Slots (records).  No group answers a far[] a window above it when the slots are shorter than the window;
        // a slot's far[] are read from the table when the walk has reached the slot's first position.  The bytes of
        // a group come in by one 16-byte and one 4-byte load per lane, a group ahead of their use, and the keys are
        // cut out of them in LDS (a key load per lane and instruction, unaligned, kept the address unit busy longer
        // than everything else took); groups at the edges of the view load their keys lane by lane.
        // relg: offset of the group's first position inside its slot
        auto rel_down = [&](uint32_t rel) { return rel >= kIdxSpan ? rel - kIdxSpan : rel + iso_bytes - kIdxSpan; };
        uint32_t relg = (uint32_t)((a0 - iso_first + (uint64_t)(n_groups - 1u) * kIdxSpan) % iso_bytes);
        uint8_t *const s_stage = reinterpret_cast<uint8_t *>(s_mem + kIdxRBuckets);
        const uint32_t mis_a = (uint32_t)(reinterpret_cast<uintptr_t>(dseg) & 15u);  // groups start on multiples of kIdxSpan from a0
        // the staged loads start up to 15 bytes below the group's first byte and end up to 272 behind its last
        auto stageable = [&](uint32_t q) {
            return q + kIdxSpan <= q_top && q + kIdxSpan + 320u <= n_rel && (q != 0u || a0 >= 16u);
        };
        struct RawA {
            uint4 a;
            uint32_t at;
        };
        auto prefetch_a = [&](RawA &r, uint32_t q) __attribute__((always_inline)) {
            const uint8_t *const pa = dseg + q - mis_a;
            r.a = *reinterpret_cast<const uint4 *>(pa + lane * 16u);
            r.at = *reinterpret_cast<const uint32_t *>(pa + kIdxSpan + lane * 4u);
        };
        auto stage_a = [&](const RawA &r) __attribute__((always_inline)) {
            asm volatile("" ::: "memory");  // the typed reads before are of the bytes that were here
            *reinterpret_cast<uint4 *>(s_stage + lane * 16u) = r.a;
            *reinterpret_cast<uint32_t *>(s_stage + kIdxSpan + lane * 4u) = r.at;
            asm volatile("" ::: "memory");  // the typed reads below are of these bytes
        };
        // insert-mode keys of the staged group: instruction k, lane l holds position qg + kIdxSpan - 1 - (64 k + l)
        auto keys_from_stage = [&](GroupKeys &buf) __attribute__((always_inline)) {
            const uint32_t kb0 = mis_a + (kIdxSpan - 1u) - lane;
            const uint32_t *const wp = reinterpret_cast<const uint32_t *>(s_stage + (kb0 & ~3u) - 64u * (kIdxGroup - 1u));
            const uint32_t sh = kb0 & 3u;
#pragma unroll
            for (int k = 0; k < (int)kIdxGroup; k++) {
                const uint32_t d0 = wp[((int)kIdxGroup - 1 - k) * 16], d1 = wp[((int)kIdxGroup - 1 - k) * 16 + 1];
                buf.k[k].a = __builtin_amdgcn_alignbyte(d1, d0, sh);
                buf.k[k].b = d1 >> (sh * 8u);  // (its lowest byte is the key's fifth)
                buf.k[kIdxGroup + k] = buf.k[k];
            }
        };
        // far[] of the staged group at pb, positions ascending with the lane, from the table as it stands
        auto sweep_staged = [&](uint32_t pb, uint32_t pe) __attribute__((always_inline)) {
            const uint32_t kb0 = mis_a + lane;
            const uint32_t *const wp = reinterpret_cast<const uint32_t *>(s_stage + (kb0 & ~3u));
            const uint32_t sh = kb0 & 3u;
            uint32_t h[kIdxGroup];
#pragma unroll
            for (int j = 0; j < (int)kIdxGroup; j++) {
                const uint32_t d0 = wp[j * 16], d1 = wp[j * 16 + 1];
                h[j] = key_hash<K>(__builtin_amdgcn_alignbyte(d1, d0, sh), K > 4 ? d1 >> (sh * 8u) : 0u) >> (32 - kIdxRBits);
            }
#pragma unroll
            for (int j = 0; j < (int)kIdxGroup; j++) h[j] = s_head[h[j]];
#pragma unroll
            for (int j = 0; j < (int)kIdxGroup; j++) {
                const uint32_t p = pb + (uint32_t)j * 64 + lane;
                if (p >= q_far && p < pe) st16(far_seg, p, h[j] ? p - (kEnc - h[j]) : 0u);
            }
        };
        constexpr uint32_t kSweepAhead = 3;  // groups of a slot whose bytes are fetched ahead for the sweep
        RawA raw = RawA{make_uint4(0, 0, 0, 0), 0u};
        if (stageable((n_groups - 1u) * kIdxSpan)) prefetch_a(raw, (n_groups - 1u) * kIdxSpan);
        for (uint32_t g = n_groups; g > 0;) {
            --g;
            const uint32_t qg = g * kIdxSpan, qn = qg - kIdxSpan;
            const uint32_t slot_end = slot_end_of(qg, relg);
            const bool sweep = relg == 0u;
            const bool n0 = qg >= q_b || qg + kIdxSpan <= q_a, n1 = qg >= q_a && qg + kIdxSpan <= q_b;
            const bool l0 = look_none(qg, slot_end);
            const bool st_g = l0 && stageable(qg), st_n = g >= 1u && stageable(qn);
            // the sweep of a slot: its groups above this one come in again (the staged sweep needs every one of them
            // well inside the view and the slot made of whole groups)
            uint32_t pe = qg + reach < q_b ? qg + reach : q_b;
            pe = pe < slot_end ? pe : slot_end;
            const uint32_t n_sw = sweep && pe > qg ? (pe - qg + kIdxSpan - 1u) / kIdxSpan : 0u;
            const bool sw_staged = sweep && st_g && n_sw >= 1u && n_sw <= kSweepAhead + 1u && qg >= q_far &&
                                   stageable(qg + (n_sw - 1u) * kIdxSpan);
            RawA sw[kSweepAhead];
#pragma unroll
            for (uint32_t j = 0; j < kSweepAhead; j++) {
                sw[j] = RawA{make_uint4(0, 0, 0, 0), 0u};
                if (sw_staged && j + 1u < n_sw) prefetch_a(sw[j], qg + (j + 1u) * kIdxSpan);
            }
            GroupKeys cur;
            if (st_g) {
                stage_a(raw);
                if (st_n) prefetch_a(raw, qn);
                keys_from_stage(cur);
            } else {
                if (st_n) prefetch_a(raw, qn);
                load_keys(cur, qg, !l0);
            }
            general(cur, qg, slot_end, sweep && !sw_staged, is_full(qg), n0, n1, l0, look_all(qg, slot_end));
            if (sw_staged) {
                sweep_staged(qg, pe);  // this group's bytes are still staged
#pragma unroll
                for (uint32_t j = 0; j < kSweepAhead; j++) {
                    if (j + 1u < n_sw) {
                        stage_a(sw[j]);
                        sweep_staged(qg + (j + 1u) * kIdxSpan, pe);
                    }
                }
            }
            relg = rel_down(relg);
        }
    } else {
        // No slots: the common case -- every key inside the view, every look-up wanted, the whole group owned or not
        // owned -- is spelled out without branches (the kernel is bound by the number of instructions it issues and
        // by the address unit's work: about fifteen instructions per LDS operation), and its bytes and results pass
        // through LDS: a group's two regions of bytes come in by two 16-byte loads and one 4-byte load per lane (a
        // key load per lane and instruction, unaligned and with neighbouring lanes in two different regions, kept
        // the address unit busy longer than everything else took), the results leave in 16-byte stores.
        uint8_t *const s_stage = reinterpret_cast<uint8_t *>(s_mem + kIdxRBuckets);
        uint16_t *const s_out = reinterpret_cast<uint16_t *>(s_stage + 2 * kIdxStageBytes);
        const uint32_t mis_a = (uint32_t)(reinterpret_cast<uintptr_t>(dseg) & 15u);           // groups start on multiples of
        const uint32_t mis_b = (uint32_t)(reinterpret_cast<uintptr_t>(dseg + window) & 15u);  // kIdxSpan from a0
        // results leave in aligned 16-byte stores
        const bool out_aligned = (reinterpret_cast<uintptr_t>(next_seg) & 15u) == 0 &&
                                 (reinterpret_cast<uintptr_t>(far_seg + window) & 15u) == 0;
        // the staged loads start up to 15 bytes below the group's first byte and end up to 272 behind its last
        auto is_fast = [&](uint32_t q) {
            return out_aligned && q + kIdxSpan <= q_top && q + kIdxSpan + window + 320u <= n_rel && (q != 0u || a0 >= 16u) &&
                   (q >= q_b || q + kIdxSpan <= q_a || (q >= q_a && q + kIdxSpan <= q_b)) &&
                   (look_none(q, 0xFFFFFFFFu) || look_all(q, 0xFFFFFFFFu)) && !(q == 0u && a0 == r0);
        };
        struct RawBlocks {
            uint4 a, b;
            uint32_t at, bt;
        };
        auto prefetch = [&](RawBlocks &r, uint32_t q, bool with_look) __attribute__((always_inline)) {
            const uint8_t *const pa = dseg + q - mis_a;
            r.a = *reinterpret_cast<const uint4 *>(pa + lane * 16u);
            r.at = *reinterpret_cast<const uint32_t *>(pa + kIdxSpan + lane * 4u);
            if (with_look) {
                const uint8_t *const pb = dseg + q + window - mis_b;
                r.b = *reinterpret_cast<const uint4 *>(pb + lane * 16u);
                r.bt = *reinterpret_cast<const uint32_t *>(pb + kIdxSpan + lane * 4u);
            } else {
                r.b = r.a;  // (both paths write every field: see load_keys)
                r.bt = r.at;
            }
        };
        auto stage = [&](const RawBlocks &r, bool with_look) __attribute__((always_inline)) {
            *reinterpret_cast<uint4 *>(s_stage + lane * 16u) = r.a;
            *reinterpret_cast<uint32_t *>(s_stage + kIdxSpan + lane * 4u) = r.at;
            if (with_look) {
                *reinterpret_cast<uint4 *>(s_stage + kIdxStageBytes + lane * 16u) = r.b;
                *reinterpret_cast<uint32_t *>(s_stage + kIdxStageBytes + kIdxSpan + lane * 4u) = r.bt;
            }
            asm volatile("" ::: "memory");  // the typed reads below are of these bytes
        };
        // bucket of the key that starts sh bytes into the two staged words d0, d1
        auto staged_hash = [&](uint32_t d0, uint32_t d1, uint32_t sh) __attribute__((always_inline)) {
            const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, sh);
            return key_hash<K>(lo, K > 4 ? d1 >> (sh * 8u) : 0u) >> (32 - kIdxRBits);
        };
        // fast_ins: 16 x 64 inserts; kNext: the positions are owned (their next[] is stored)
        auto fast_ins = [&](uint32_t qg, auto next_c) __attribute__((always_inline)) {
            constexpr bool kNext = decltype(next_c)::value;
            IDBG(const long long dt0 = clock64();)
            const uint32_t kb0 = mis_a + (kIdxSpan - 1u) - lane;  // byte of instruction 0's key; instruction k: 64 k below
            const uint32_t *const wp = reinterpret_cast<const uint32_t *>(s_stage + (kb0 & ~3u) - 64u * (kIdxGroup - 1u));
            const uint32_t sh = kb0 & 3u;
            const uint32_t e0 = kEnc - ins_pos(qg, 0);  // entry of instruction k's position: e0 + 64 k
            // (all the reads of the staged bytes first: the compiler keeps a read behind every earlier atomic, which
            // it takes to touch the same memory, and a read issued behind an atomic waits for that atomic's return)
            uint32_t kw[kIdxGroup][2], got[kIdxGroup];
#pragma unroll
            for (int k = 0; k < (int)kIdxGroup; k++) {
                kw[k][0] = wp[((int)kIdxGroup - 1 - k) * 16];
                kw[k][1] = wp[((int)kIdxGroup - 1 - k) * 16 + 1];
            }
#pragma unroll
            for (int k = 0; k < (int)kIdxGroup; k++)
                got[k] = table_max(staged_hash(kw[k][0], kw[k][1], sh), e0 + 64u * (uint32_t)k);
            IDBG(const long long dt1 = clock64(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const long long dt2 = clock64();)
            uint16_t *const np = next_seg + ins_pos(qg, 0);
#pragma unroll
            for (int k = 0; k < (int)kIdxGroup; k++) {
                const uint32_t d = e0 + 64u * (uint32_t)k - got[k];  // distance to the successor; huge if there is none
                worst = (int32_t)d < worst ? (int32_t)d : worst;
                if (kNext) np[-64 * k] = (uint16_t)(d <= window ? d : 0u);
            }
            IDBG(const long long dt3 = clock64(); iph[0] += dt1 - dt0; iph[1] += dt2 - dt1; iph[2] += dt3 - dt2; iph[7] += 1;)
        };
        // fast_il: 32 x (32 inserts, 32 look-ups)
        auto fast_il = [&](uint32_t qg, auto next_c) __attribute__((always_inline)) {
            constexpr bool kNext = decltype(next_c)::value;
            IDBG(const long long dt0 = clock64();)
            const uint32_t kb0 = (odd ? kIdxStageBytes + mis_b : mis_a) + (kIdxSpan - 1u) - pair;  // instruction i: 32 i below
            const uint32_t *const wp = reinterpret_cast<const uint32_t *>(s_stage + (kb0 & ~3u) - 32u * (2u * kIdxGroup - 1u));
            const uint32_t sh = kb0 & 3u;
            // Entry of instruction i's position: e0 + 32 i.  Odd lanes take one more, so that for both kinds of lane
            // d = (that) - (entry handed back) is 1 or more when all is in order: an insert gets a member above its
            // position, a look-up one at or above it.  Even lanes: d = distance to the successor; odd: window + 1 - far.
            const uint32_t e0 = kEnc - il_pos(qg, 0) + odd01;
            const uint32_t ins_mask = odd ? 0u : 0xFFFFFFFFu;
            uint32_t kw[2 * kIdxGroup][2], got[2 * kIdxGroup];  // (all reads first: see fast_ins)
#pragma unroll
            for (int i = 0; i < 2 * (int)kIdxGroup; i++) {
                kw[i][0] = wp[(2 * (int)kIdxGroup - 1 - i) * 8];
                kw[i][1] = wp[(2 * (int)kIdxGroup - 1 - i) * 8 + 1];
            }
#pragma unroll
            for (int i = 0; i < 2 * (int)kIdxGroup; i++)
                got[i] = table_max(staged_hash(kw[i][0], kw[i][1], sh), (e0 + 32u * (uint32_t)i) & ins_mask);
            IDBG(const long long dt1 = clock64(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const long long dt2 = clock64();)
            const int32_t sgn = odd ? -1 : 1, off = odd ? (int32_t)window + 1 : 0;
            const uint32_t lim = window + odd01;
            // even lanes: next[p] of the group's positions, odd lanes: far[p + window]; collected in LDS in position
            // order (instruction i: 32 i entries below instruction 0's)
            uint16_t *const op = s_out + (odd ? kIdxSpan : 0u) + (kIdxSpan - 1u) - pair - 32u * (2u * kIdxGroup - 1u);
#pragma unroll
            for (int i = 0; i < 2 * (int)kIdxGroup; i++) {
                const uint32_t d = e0 + 32u * (uint32_t)i - got[i];
                worst = (int32_t)d < worst ? (int32_t)d : worst;
                const uint32_t val = (uint32_t)(__mul24(sgn, (int32_t)d) + off);  // d, or window + 1 - d
                got[i] = d <= lim ? val : 0u;  // no successor within the window (an empty bucket gives a huge d)
            }
            if (kNext || odd) {
#pragma unroll
                for (int i = 0; i < 2 * (int)kIdxGroup; i++) op[(2 * (int)kIdxGroup - 1 - i) * 32] = (uint16_t)got[i];
            }
            asm volatile("" ::: "memory");  // the typed reads below are of these entries
            // 2 x kIdxSpan bytes each: two 16-byte blocks per lane
            const uint4 *const o4 = reinterpret_cast<const uint4 *>(s_out);
            uint4 *const f4 = reinterpret_cast<uint4 *>(far_seg + qg + window);
            f4[lane] = o4[kIdxSpan / 8 + lane];
            f4[64 + lane] = o4[kIdxSpan / 8 + 64 + lane];
            if (kNext) {
                uint4 *const n4 = reinterpret_cast<uint4 *>(next_seg + qg);
                n4[lane] = o4[lane];
                n4[64 + lane] = o4[64 + lane];
            }
            asm volatile("" ::: "memory");
            IDBG(const long long dt3 = clock64(); iph[3] += dt1 - dt0; iph[4] += dt2 - dt1; iph[5] += dt3 - dt2; iph[6] += 1;)
        };
        static_assert(kIdxSpan == 1024, "two 16-byte blocks per lane and array of results");
        RawBlocks raw;
        {
            const uint32_t q = (n_groups - 1u) * kIdxSpan;
            if (is_fast(q)) prefetch(raw, q, !look_none(q, 0xFFFFFFFFu));
            else raw = RawBlocks{make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0), 0u, 0u};
        }
        for (uint32_t g = n_groups; g > 0;) {
            --g;
            const uint32_t qg = g * kIdxSpan, qn = qg - kIdxSpan;
            const bool sweep = g == 0u && a0 == r0;
            const bool n0 = qg >= q_b || qg + kIdxSpan <= q_a, n1 = qg >= q_a && qg + kIdxSpan <= q_b;
            const bool l0 = look_none(qg, 0xFFFFFFFFu), l1 = look_all(qg, 0xFFFFFFFFu);
            const bool fast_n = g >= 1u && is_fast(qn);
            const bool look_n = fast_n && !look_none(qn, 0xFFFFFFFFu);
            if (is_fast(qg)) {
                stage(raw, !l0);
                if (fast_n) prefetch(raw, qn, look_n);
                if (l1 && n1)
                    fast_il(qg, std::true_type{});
                else if (l1)
                    fast_il(qg, std::false_type{});
                else if (n1)
                    fast_ins(qg, std::true_type{});
                else
                    fast_ins(qg, std::false_type{});
            } else {
                if (fast_n) prefetch(raw, qn, look_n);
                GroupKeys cur;
                load_keys(cur, qg, !l0);
                general(cur, qg, 0xFFFFFFFFu, sweep, is_full(qg), n0, n1, l0, l1);
            }
        }
    }
    violated |= worst <= 0;
    if (__any(violated) && lane == 0) atomicOr(status, 2u);
    IDBG(if (lane == 0) for (int i = 0; i < 8; i++) atomicAdd(&g_iph[i], iph[i]);)
}

// One workgroup (one wave) per segment and key length: blockIdx.y selects the key length, so that all the
// indexes of a call are built by one launch and share the machine.
template <bool kIso>
__global__ __launch_bounds__(64) void k_index_r(const uint8_t *__restrict__ data, uint64_t n_view, uint64_t r0,
                                                 uint64_t r1, uint64_t far_lo, uint32_t window, IndexSet set,
                                                 uint64_t base, uint32_t *status, uint64_t iso_first,
                                                 uint32_t iso_bytes, uint32_t seg_len) {
    // the table (a constant minus the lowest member so far, 0 = empty) and, without slots, the group's bytes and results
    __shared__ __attribute__((aligned(16))) uint32_t s_mem[kIso ? kIdxLdsWordsIso : kIdxLdsWordsStream];
#ifdef ZG_IDX_LDS_PAD  // measurement: fewer waves per CU
    __shared__ uint32_t s_pad[ZG_IDX_LDS_PAD];
    if (n_view == 0x123456789ull) s_pad[threadIdx.x] = 1;
#endif
    const int key = set.key[blockIdx.y];
    uint16_t *far = set.far[blockIdx.y];
    uint16_t *next = set.next[blockIdx.y];
#ifdef ZG_INDEX_DEBUG
    const unsigned long long dbg_t0 = wall_clock64();
    struct DbgEnd {
        unsigned long long t0;
        int key;
        __device__ ~DbgEnd() {
            if (threadIdx.x == 0) {
                const unsigned long long slot = atomicAdd(&g_idbg[0], 1ull);
                if (slot < 65536) {
                    g_idbg[1 + 3 * slot] = t0;
                    g_idbg[2 + 3 * slot] = wall_clock64();
                    g_idbg[3 + 3 * slot] = (unsigned long long)blockIdx.x | ((unsigned long long)key << 32);
                }
            }
        }
    } dbg_end{dbg_t0, key};
#endif
    if (key == 5)
        index_r_segment<5, kIso>(s_mem, blockIdx.x, seg_len, data, n_view, r0, r1, far_lo, window, far, next, base, status,
                                 iso_first, iso_bytes);
    else if (key == 4)
        index_r_segment<4, kIso>(s_mem, blockIdx.x, seg_len, data, n_view, r0, r1, far_lo, window, far, next, base, status,
                                 iso_first, iso_bytes);
    else
        index_r_segment<3, kIso>(s_mem, blockIdx.x, seg_len, data, n_view, r0, r1, far_lo, window, far, next, base, status,
                                 iso_first, iso_bytes);
}

#ifdef ZG_INDEX_DEBUG
extern "C" int zg_index_waves_per_cu() {
    int nb = 0;
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_index_r<false>, 64, 0) == hipSuccess ? nb : -1;
}
#endif

// k_index_brute: the same two arrays for small windows (LZ77Compressor's default is 144), one thread per
// position, by direct scan of the window on either side.
template <int K>
__global__ __launch_bounds__(256) void k_index_brute(const uint8_t *__restrict__ data, uint64_t n_view, uint64_t r0,
                                                      uint64_t r1, uint64_t far_lo, uint32_t window,
                                                      uint16_t *__restrict__ far, uint16_t *__restrict__ next,
                                                      uint64_t base, uint64_t iso_first, uint32_t iso_bytes) {
    const uint64_t p = r0 + (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= r1) return;
    uint32_t fd = 0, nd = 0;
    if (p + K <= n_view) {
        const uint32_t h = hash_at<K>(data, p, n_view);
        uint64_t lo = p > window ? p - window : 0, hi_lim = ~0ull;
        if (iso_bytes) {  // stay inside the position's slot
            const uint64_t slot = p - (p - iso_first) % iso_bytes;
            lo = lo > slot ? lo : slot;
            hi_lim = slot + iso_bytes;
        }
        for (uint64_t s = lo; s < p; s++)
            if (hash_at<K>(data, s, n_view) == h) {
                fd = (uint32_t)(p - s);
                break;
            }
        for (uint64_t q = p + 1; q <= p + window && q + K <= n_view && q < hi_lim; q++)
            if (hash_at<K>(data, q, n_view) == h) {
                nd = (uint32_t)(q - p);
                break;
            }
    }
    next[p - base] = (uint16_t)nd;
    if (p >= far_lo) far[p - base] = (uint16_t)fd;
}

template <int K>
static hipError_t launch_index_slow(const uint8_t *data, uint64_t n_view, uint64_t r0, uint64_t r1, uint64_t far_lo,
                                    uint32_t window, uint16_t *far, uint16_t *next, uint64_t base, bool brute,
                                    uint64_t iso_first, uint32_t iso_bytes, hipStream_t s) {
    if (brute) {
        hipLaunchKernelGGL(k_index_brute<K>, dim3((uint32_t)((r1 - r0 + 255) / 256)), dim3(256), 0, s, data, n_view,
                           r0, r1, far_lo, window, far, next, base, iso_first, iso_bytes);
    } else {
        const uint64_t nseg = (r1 - r0 + kIdxSegment - 1) / kIdxSegment;
        const hipError_t e = hipMemsetAsync(next + (r0 - base), 0, (r1 - r0) * sizeof(uint16_t), s);  // links are scattered
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_index<K>, dim3((uint32_t)nseg), dim3(64), 0, s, data, n_view, r0, r1, far_lo, window, far,
                           next, base);
    }
    return hipGetLastError();
}

hipError_t launch_index(const uint8_t *data, uint64_t n_view, uint64_t r0, uint64_t r1, uint64_t far_lo,
                        uint32_t window, const IndexSet &set, uint64_t base, uint32_t *status, bool safe,
                        uint64_t iso_first, uint32_t iso_bytes, hipStream_t s) {
    if (r1 <= r0 || set.n == 0) return hipSuccess;
    // the conservative k_index has no notion of isolated slots: those fall back to the direct scan, which has
    const bool brute = window < 2048 || n_view < 64 || (safe && iso_bytes);
    if (brute || safe) {
        for (uint32_t which = 0; which < set.n; which++) {
            const int key = set.key[which];
            uint16_t *far = set.far[which], *next = set.next[which];
            const hipError_t e =
                key == 5 ? launch_index_slow<5>(data, n_view, r0, r1, far_lo, window, far, next, base, brute, iso_first,
                                                iso_bytes, s)
                : key == 4 ? launch_index_slow<4>(data, n_view, r0, r1, far_lo, window, far, next, base, brute,
                                                  iso_first, iso_bytes, s)
                           : launch_index_slow<3>(data, n_view, r0, r1, far_lo, window, far, next, base, brute,
                                                  iso_first, iso_bytes, s);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    const uint32_t ny = set.n;
    // Segment length.  A wave walks its segment plus `reach` positions on either side.  The launch is cut into whole
    // fillings of the machine -- a few waves over would cost a filling of their own: 256 MiB of 4 KiB records took
    // 5.2 ms with 1026 waves on 1024 places, 3.1 ms with 1023.
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        cus = 256;
    // waves of this kernel that a CU holds at once (its LDS table allows five, its registers two per SIMD)
    // waves of the kernel that a CU holds at once.  The occupancy query counts five tables of 32 KiB in the CU's 160 KiB
    // of LDS; the hardware was seen to admit four, and a plan made for five then needs a second, nearly empty
    // filling (256 MiB of 4 KiB records: 4.9 ms planned for five, 3.2 ms planned for four).
    static const int per_cu_stream = [] {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_index_r<false>, 64, 0) != hipSuccess || nb <= 0) nb = 4;
        return nb < 4 ? nb : 4;
    }();
    static const int per_cu_iso = [] {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_index_r<true>, 64, 0) != hipSuccess || nb <= 0) nb = 4;
        return nb < 4 ? nb : 4;
    }();
    static const int cap_env = [] {  // measurement hook: longest segment in KiB
        const char *e = std::getenv("ZLIB_AMD_INDEX_CAP_KB");
        return e ? atoi(e) : 0;
    }();
    const int per_cu = iso_bytes ? per_cu_iso : per_cu_stream;
    const uint64_t slots = (uint64_t)per_cu * (uint64_t)cus, n_pos = r1 - r0;
    const uint64_t reach = iso_bytes && iso_bytes < window ? iso_bytes : window;
    // Every position costs the same whatever the data (no walk depends on it), so the waves of a filling end together
    // and long segments are best: a segment of 768 Ki positions walks 1.08 of them, one of 256 Ki 1.25 (2.26 ms
    // against 2.63 per 256 MiB and three key lengths).
    const uint64_t seg_cap = cap_env > 0 ? (uint64_t)cap_env << 10 : kIdxSegCap;
    const uint64_t cap = reach <= 4096 ? (2048ull << 10) : seg_cap;
    // whole fillings: as many as segments of `cap` need, and the segments then as long as those fillings allow
    // (a twentieth over a whole number of fillings stretches the segments instead of adding a filling)
    const uint64_t fillings = (n_pos * ny + slots * cap * 19 / 20) / (slots * cap);
    const uint64_t segs_per_key = slots * (fillings ? fillings : 1) / ny;
    uint64_t seg_len = (n_pos + segs_per_key - 1) / (segs_per_key ? segs_per_key : 1);
    seg_len = (seg_len + kIdxSpan - 1) & ~(uint64_t)(kIdxSpan - 1);
    seg_len = seg_len < kIdxSegMin ? kIdxSegMin : seg_len;
    const uint64_t nseg = (n_pos + seg_len - 1) / seg_len;
    if (iso_bytes)
        hipLaunchKernelGGL(k_index_r<true>, dim3((uint32_t)nseg, ny), dim3(64), 0, s, data, n_view, r0, r1, far_lo, window,
                           set, base, status, iso_first, iso_bytes, (uint32_t)seg_len);
    else
        hipLaunchKernelGGL(k_index_r<false>, dim3((uint32_t)nseg, ny), dim3(64), 0, s, data, n_view, r0, r1, far_lo, window,
                           set, base, status, iso_first, iso_bytes, (uint32_t)seg_len);
    return hipGetLastError();
}

uint64_t index_lookback(uint32_t window) { return window; }

// ---------------------------------------------------------------------------
// k_match / k_match_lo: one workgroup per tile of kMatchTile positions.  The
// tile's bytes plus the preceding window and the forward links of that range
// are staged in LDS; the threads then claim positions of the tile one by one.
// ---------------------------------------------------------------------------

template <class G>
__global__ __launch_bounds__(G::match_threads) void k_match(const uint8_t *__restrict__ data, uint64_t first,
                                                          uint64_t end, uint64_t chunk_bytes,
                                                          const uint32_t *__restrict__ clen, uint32_t isolated,
                                                          uint32_t c0, uint32_t tiles_per_chunk, MatchParams mp,
                                                          uint32_t rec_min, uint32_t emit_term,
                                                          const uint16_t *__restrict__ far,
                                                          const uint16_t *__restrict__ next, uint64_t base,
                                                          uint32_t *__restrict__ match,
                                                          uint32_t *__restrict__ open_list,
                                                          uint32_t *__restrict__ open_cnt,
                                                          const uint8_t *__restrict__ walk_chunk) {
    // one LDS image: bytes first (so that byte reads need no base add), links behind them
    __shared__ __attribute__((aligned(16))) uint8_t s_mem[G::data_lds + 2 * G::next_lds];
    __shared__ uint32_t s_counter;  // next unclaimed position of the tile
    __shared__ uint32_t s_nopen;    // positions handed on to level 2
    // far[] of the tile's last kLate positions.  A lane holds its next claim ahead of need, so that the position's
    // far[] entry is on its way from memory while the current walk runs -- but when the tile runs out, every lane
    // then has two walks left while idle lanes have none.  In the last part of the tile positions are therefore
    // claimed only when needed, their far[] entries taken from this copy.
    // (Records only: 5 % there; the large tiles lose a little by it.)
    constexpr uint32_t kLate = G::tile >= 16384u ? (uint32_t)ZG_MATCH_LATE : G::tile / 4u;
    __shared__ uint16_t s_far_late[kLate ? kLate : 1u];
    uint8_t *const s_data = s_mem;
    uint16_t *const s_next = reinterpret_cast<uint16_t *>(s_mem + G::data_lds);

    const uint32_t bid = xcd_tile_index(blockIdx.x, gridDim.x);
    const uint32_t chunk = c0 + bid / tiles_per_chunk;
    const uint32_t tile = bid % tiles_per_chunk;
    const uint64_t cs = first + (uint64_t)chunk * chunk_bytes;
    const uint64_t ce = chunk_end(cs, chunk_bytes, end, clen, chunk);
    const uint64_t t0 = cs + (uint64_t)tile * G::tile;
    // nothing to do past the chunk's end, nor in a chunk that is left to the search along the parse (k_walk)
    if (t0 >= ce || (walk_chunk && walk_chunk[chunk - c0])) {
        if (emit_term && threadIdx.x == 0) open_cnt[bid] = 0;
        return;
    }
    const uint64_t t1 = (t0 + G::tile < ce) ? t0 + G::tile : ce;
    uint64_t wlo = t0 > mp.window ? t0 - mp.window : 0;
    if (isolated && wlo < cs) wlo = cs;               // nothing before the chunk's own first byte exists for it
    const uint64_t w0 = wlo & ~(uint64_t)15;          // LDS origin, 16-byte aligned
    const uint64_t w1 = (t1 + mp.max_len < ce) ? t1 + mp.max_len : ce;          // bytes needed: [w0, w1)
    const uint32_t nbytes = (uint32_t)(w1 - w0);
    const uint32_t nlinks = (uint32_t)(t1 - w0);
    const uint32_t tid = threadIdx.x;
    if (tid == 0) {
        s_counter = 0;
        s_nopen = 0;
    }
    MDBG(__shared__ unsigned long long s_t_exh; const unsigned long long t_begin = wall_clock64();
         if (tid == 0) s_t_exh = 0;)

    const uint32_t late0 = (uint32_t)(t1 - t0) > kLate ? (uint32_t)(t1 - t0) - kLate : 0u;
    for (uint32_t i = tid; late0 + i < (uint32_t)(t1 - t0); i += G::match_threads) s_far_late[i] = far[(t0 - base) + late0 + i];
    stage_tile(data, next, base, w0, nbytes, nlinks, s_data, s_next, tid, G::match_threads, G::data_lds);
    __syncthreads();
    MDBG(const unsigned long long t_staged = wall_clock64();)

    // Search.  Chain lengths are very uneven (Zipf-like), so positions are not bound to threads: a lane claims
    // the next unresolved position of the tile from an LDS counter whenever it finishes one.  Every iteration
    // all lanes run kHeadSteps branch-free "head" steps: one four-byte probe of the candidate and its forward
    // link.  The probe looks at the only place that can still matter: for a candidate too far away to end the
    // search, the four bytes ending at offset `best` (they must all agree for it to beat the best length so
    // far); for a candidate within max_len of the position, or while nothing has been found, the first bytes
    // (four when only lengths of four or more count here, else three).  The two rarer kinds of work -- measuring a candidate that survived the probe, and retiring a
    // position / claiming the next one -- are deferred until enough lanes of the wave wait for them, so that
    // each of the three code paths runs with most of its lanes active.
    const uint32_t min_len = mp.min_len, max_len = mp.max_len;
    const uint32_t tn = (uint32_t)(t1 - t0);
    const uint32_t pbase = (uint32_t)(t0 - w0);  // LDS offset of tile position 0
    const uint32_t lo_off = (uint32_t)(wlo - w0); // LDS offset of the lowest candidate there is
    const uint16_t *far_t = far + (t0 - base);
    uint32_t *open_t = open_list + (uint64_t)bid * G::tile;
    uint32_t *match_t = match + (t0 - base);
    const uint64_t tail_room = ce - t0;  // bytes from tile position 0 to the end of the chunk
    const uint32_t tail_room32 = tail_room < 0x7FFFFFFFull ? (uint32_t)tail_room : 0x7FFFFFFFu;

    // head steps between checks of the deferred work; walks inside a record are short and meet a candidate to
    // measure every few steps, so their trips are shorter
    constexpr int kHeadSteps = G::tile == kMatchRecordTile ? ZG_MATCH_STEPS_REC : ZG_MATCH_STEPS;
#ifndef ZG_EVAL_BATCH
#define ZG_EVAL_BATCH 1
#endif
#ifndef ZG_FETCH_BATCH
#define ZG_FETCH_BATCH 2
#endif
    constexpr uint32_t kEvalBatch = ZG_EVAL_BATCH, kFetchBatch = ZG_FETCH_BATCH;
#ifndef ZG_SWITCH_HOPS
#define ZG_SWITCH_HOPS 1
#endif
#ifndef ZG_SWITCH_NOFF
#define ZG_SWITCH_NOFF 7   // offsets looked at when a walk changes chains
#endif
#ifndef ZG_SWITCH_GAIN4
#define ZG_SWITCH_GAIN4 8  // quarters: the new chain's links must be this many times as long
#endif
#ifndef ZG_SWITCH_TRIPS
#define ZG_SWITCH_TRIPS 1
#endif
#ifndef ZG_SWITCH_TRIPS_REC
#define ZG_SWITCH_TRIPS_REC 2  // inside a record
#endif
#ifndef ZG_SWITCH_EVERY
#define ZG_SWITCH_EVERY 4
#endif
#ifndef ZG_SWITCH_REMAIN
#define ZG_SWITCH_REMAIN 48
#endif
#ifndef ZG_SWITCH_BATCH
#define ZG_SWITCH_BATCH 4
#endif
    // a walk may change chains after kSwitchTrips trips (0: never), once kSwitchBatch lanes want to
    constexpr uint32_t kSwitchTrips = G::tile == kMatchRecordTile ? ZG_SWITCH_TRIPS_REC : ZG_SWITCH_TRIPS;
    constexpr uint32_t kSwitchBatch = ZG_SWITCH_BATCH, kSwitchRemain = ZG_SWITCH_REMAIN;
    // what a lane is doing: walking (hd), waiting to have a candidate measured (ev), waiting to retire its
    // position and claim another (ft), or none of them (no positions left).  Kept as predicates, so that the
    // bookkeeping of a head step is scalar mask arithmetic.
    bool hd = false, ev = false, ft = true;
    uint32_t idx = 0, po = pbase, so = pbase, cap = 0, best = 0, bestd = 0, p0 = 0, p1 = 0;
    int32_t so_near = 0, so_end = 0;  // candidates at or above: within max_len of the position / past the loop guard
    const uint32_t mask0 = rec_min >= 4 ? 0xFFFFFFFFu : 0x00FFFFFFu;  // first-bytes probe
    uint32_t toff = 0, pt = 0, tmask = mask0;  // far-candidate probe: offset, expected bytes, byte mask
    // Chain switching.  Once a candidate has matched `best` >= 6 bytes, every candidate that can beat it agrees
    // with the position in bytes best-5 .. best-1 as well, so it is found on the chain of position pos + o,
    // o = best - 5, o positions above the candidate itself -- often a far shorter chain (a frequent five bytes
    // followed by a rare sixth).  The best candidate, moved up by o, is a member of that chain, so the walk can
    // continue from there.  The cursor `so` is therefore a CHAIN position: the candidate is so - o, and
    // so_near, so_end, toff and noff (= -o) are kept shifted by o so that a head step costs the same.
    // Candidates the new chain skips match fewer than `best` bytes; they matter only if one of them would end
    // the search (distance d <= its length + 1, hence d <= best): those few distances are tested directly
    // when the walk changes chains.  Only walks that have run long change chains.
    uint32_t noff = 0;    // minus the chain offset o
    uint32_t st = 0;      // o | largest distance tested for a search-ending candidate << 8 | trips since the claim << 17
    bool need_sw = false; // best has grown since the chain was chosen
    uint32_t round = 0;   // trips of the workgroup's loop (uniform)
    uint32_t termd = 0;                        // distance of the candidate that ended the search, if one did
    uint32_t dref = 0;                         // distance of the candidate the trips in `st` are counted from
    bool drain = false;                        // uniform: a lane of this wave has found no position left to claim
    bool has_pos = false;  // a position is loaded and its result is not yet written
    constexpr uint32_t kNoClaim = 0xFFFFFFFFu;  // nxt_idx: no claim is held
    uint32_t nxt_idx = late0 != 0u ? atomicAdd(&s_counter, 1u) : kNoClaim;
    uint32_t nxt_far = nxt_idx < tn ? far_t[nxt_idx] : 0u;

    // Which candidates are measured.  One that beats the best length so far agrees with the position in the four
    // bytes ending at offset `best`.  One that ends the search at distance d matches d - 1 bytes or more, so beyond
    // kNear positions back it agrees in the first kNear bytes.  Either way it agrees in the four bytes ending at
    // offset min(best, kNear - 1): that is the probe.  Only the candidates within kNear of the position are
    // measured whatever they look like.
    constexpr uint32_t kNear = ZG_MATCH_NEAR;
    // (not inside records: their chains are short)
    constexpr uint32_t kCoopLanes = G::tile == kMatchRecordTile ? 0u : (uint32_t)ZG_MATCH_COOP_LANES, kCoopBase = ZG_MATCH_COOP_BASE, kCoopGain = ZG_MATCH_COOP_GAIN;
    const uint32_t near = max_len < kNear ? max_len : kNear;
    for (;;) {
#pragma unroll
        for (int step = 0; step < kHeadSteps; step++) {
            // a candidate close to the position may end the search whatever its length: it is always
            // measured; one farther away only if it passes the probe
            const bool nearby = (int32_t)so >= so_near;
            const uint32_t w = lds_ld4(s_data, so + toff);
            const uint32_t nx = s_next[so];
            const bool guard_end = (int32_t)so >= so_end;  // loop guard: s + min_len < pos
            const bool survive = (((w ^ pt) & tmask) == 0) | nearby;
            const bool to_adv = hd & !guard_end & !survive;
            const bool to_fetch = hd & (guard_end | (!survive & (nx == 0)));
            const bool to_eval = hd & !guard_end & survive;
            so += to_adv ? nx : 0u;
            MDBG2(if (hd) atomicAdd(&g_mdbg[0], 1ull);)
            hd = to_adv & !to_fetch;
            ev |= to_eval;
            ft |= to_fetch;
        }
        MDBG2(if ((tid & 63) == 0) atomicAdd(&g_mdbg[10], 1ull);)
        uint32_t n_head = (uint32_t)__popcll(__ballot(hd));
        MDBG2(if ((tid & 63) == 0) {
            if (*(volatile uint32_t *)&s_counter >= tn) {
                atomicAdd(&g_mdbg[14], 1ull);
                atomicAdd(&g_mdbg[16], (unsigned long long)n_head);
            }
            if (n_head <= 8) atomicAdd(&g_mdbg[15], 1ull);
            if (n_head <= 2) atomicAdd(&g_mdbg[17], 1ull);
        })
        uint32_t n_eval = (uint32_t)__popcll(__ballot(ev));
        // ---- measure surviving candidates ----
        if (n_eval >= kEvalBatch || (n_eval != 0 && n_head == 0)) {
            MDBG2(if ((tid & 63) == 0) atomicAdd(&g_mdbg[11], 1ull);)
            if (ev) {
                MDBG2(atomicAdd(&g_mdbg[2], 1ull);)
                uint32_t m;
                const uint32_t sc = so;  // chain position of the candidate
                so += noff;              // the candidate itself
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
                if (m >= rec_min) {
                    if (d <= max_len && m + 1 >= d) {  // the growing match reached pos: search over
                        finish = true;
                        termd = d;
                    } else if (m > best) {
                        best = m;
                        bestd = d;
                        const uint32_t mp_ = m < kNear - 1u ? m : kNear - 1u;  // offset of the probe's last byte
                        pt = lds_ld4(s_data, po + mp_ - 3);
                        toff = mp_ - 3 + noff;
                        tmask = 0xFFFFFFFFu;
                        finish = m == cap;  // cannot be exceeded
                        need_sw = kSwitchTrips != 0 && m >= 6;
                        st &= 0x1FFFFu;  // trips are counted from the best candidate on
                        dref = d;
                    }
                }
                const uint32_t nx = s_next[sc];
                so = sc + nx;
                ev = false;
                ft = finish || nx == 0;
                hd = !ft;
            }
            n_head = (uint32_t)__popcll(__ballot(hd));
            n_eval = 0;
        }
        // ---- long walks: move to the chain that the best length so far selects ----
        st += 1u << 17;
        ++round;
        if (kSwitchTrips != 0 && (round & (ZG_SWITCH_EVERY - 1)) == 0) {  // looked at every few trips only
            // worth it when the chain walked now is dense: at the rate of members per byte seen since the best
            // candidate, more than kSwitchRemain steps are left before the position
            const uint32_t trips = st >> 17, dcur = po - so - noff, cov = (kCoopLanes != 0 ? dref : bestd) - dcur;
            const bool sw = hd & need_sw & (trips >= kSwitchTrips) &
                            (trips * (uint32_t)kHeadSteps * (dcur >> 4) >= kSwitchRemain * (cov >> 4));
            const uint32_t n_sw = (uint32_t)__popcll(__ballot(sw));
            if (n_sw >= kSwitchBatch || (n_sw != 0 && n_sw * 4 >= n_head)) {
                if (sw) {
                    MDBG2(atomicAdd(&g_mdbg[9], 1ull);)
                    const uint32_t o_old = st & 0xFFu, tchk = (st >> 8) & 0x1FFu;
                    const uint32_t room = pbase + tn - po;  // links are staged up to the end of the tile
                    uint32_t o_hi = best - 5u;
                    o_hi = o_hi < room ? o_hi : room;
                    o_hi = o_hi < 255u ? o_hi : 255u;
                    // Any offset up to best - 5 will do; the chain to take is the sparsest one, judged by how far
                    // two links lead from the best candidate on each of them (the present chain included: a
                    // change has to promise links at least twice as long).
                    uint32_t o_new = o_old;
#if ZG_SWITCH_HOPS
                    {
                        const uint32_t cb = po - bestd;  // the best candidate
                        const uint32_t h1 = s_next[cb + o_old];
                        uint32_t h_best = ((h1 + s_next[cb + o_old + h1]) * (uint32_t)ZG_SWITCH_GAIN4) >> 2;
#pragma unroll
                        for (uint32_t k = 0; k < (uint32_t)ZG_SWITCH_NOFF; k++) {
                            const uint32_t o = o_hi - k;
                            const bool ok = k <= o_hi && o > o_old;
                            const uint32_t q = cb + (ok ? o : o_old);
                            const uint32_t n1 = s_next[q];
                            const uint32_t h = n1 + s_next[q + n1];
                            if (ok && h > h_best) {
                                h_best = h;
                                o_new = o;
                            }
                        }
                    }
#else
                    o_new = o_hi;
#endif
                    if (o_new > o_old) {
                        int32_t end_u = so_end - (int32_t)o_old;
                        // a skipped candidate at distance d ends the search iff it matches max(rec_min, d - 1) bytes
                        const uint32_t d_lo = tchk > min_len ? tchk : min_len;
                        uint32_t d_hi = best < po - lo_off ? best : po - lo_off;
                        for (uint32_t d = d_hi; d > d_lo; --d) {
                            if (lds_ld4(s_data, po - d) != p0) continue;  // rec_min >= 5: four equal bytes at least
                            const uint32_t need = d - 1u > rec_min ? d - 1u : rec_min;
                            if (need <= cap && (int32_t)(po - d) < end_u && lds_lcp(s_data, po - d, po, need) >= need) {
                                end_u = (int32_t)(po - d);
                                break;
                            }
                        }
                        so = po - bestd + o_new;
                        so_near = (int32_t)po - (int32_t)near + (int32_t)o_new;
                        so_end = end_u + (int32_t)o_new;
                        noff = 0u - o_new;
                        toff = (best < kNear - 1u ? best : kNear - 1u) - 3u - o_new;
                        st = (best << 8) | o_new;  // trips start again
                        dref = bestd;
                    }
                    need_sw = false;
                }
                n_head = (uint32_t)__popcll(__ballot(hd));
            }
        }
        // ---- retire finished positions and claim new ones ----
        const uint32_t n_fetch = (uint32_t)__popcll(__ballot(ft));
        if (n_fetch >= kFetchBatch || (n_fetch != 0 && n_head == 0 && n_eval == 0)) {
            bool gone = false;
            if (ft) {
                // Order matters for the memory counter, which retires in issue order: first take the values
                // prefetched by the previous visit (nothing younger is in flight, so this costs no wait), then
                // issue this visit's stores and, last, the loads for the next position.
                uint32_t fd = nxt_far;
                asm volatile("" : "+v"(fd));
                MDBG2(if (__ffsll(__ballot(1)) - 1 == (int)(tid & 63)) atomicAdd(&g_mdbg[12], 1ull);)
                MDBG2(atomicAdd(&g_mdbg[13], 1ull);)
                if (has_pos) {
                    match_t[idx] = best ? ((best << 16) | bestd) : 0u;
                    if (best == 0u && emit_term) {  // still open: handed on to level 2
                        const uint32_t slot = atomicAdd(&s_nopen, 1u);
                        open_t[slot] = idx | (termd << 16);
                    }
                }
                has_pos = false;
                uint32_t take = nxt_idx;
                if (take == kNoClaim) {  // the late part of the tile: claimed now, far[] from the copy in LDS
                    take = atomicAdd(&s_counter, 1u);
                    fd = take < tn ? s_far_late[take - late0] : 0u;
                }
                if (take >= tn) {
                    MDBG(if (s_t_exh == 0) s_t_exh = wall_clock64();)
                    gone = true;
                    nxt_idx = tn;
                    ft = false;  // no positions left
                } else {
                    idx = take;
                    const uint32_t remain32 = tail_room32 - idx;  // bytes to the end of the chunk, saturated
                    cap = remain32 < max_len - 1 ? remain32 : max_len - 1;
                    best = 0;
                    bestd = 0;
                    termd = 0;
                    has_pos = true;
                    // the reference searches only while pos < len - min_len, and records nothing below min_len
                    if (fd != 0 && remain32 > min_len && cap >= rec_min) {
                        po = pbase + idx;
                        so = po - fd;
                        dref = fd;
                        so_near = (int32_t)po - (int32_t)near;
                        so_end = (int32_t)po - (int32_t)min_len;
                        p0 = lds_ld4(s_data, po);
                        p1 = lds_ld4(s_data, po + 4);
                        toff = 0;
                        pt = p0;
                        tmask = mask0;
                        noff = 0;
                        st = 0;
                        hd = true;
                        ft = false;
                        MDBG2(atomicAdd(&g_mdbg[4], 1ull);)
                    }
                    need_sw = false;
                    // the next claim is held ahead only while the counter is below the late part
                    nxt_idx = take < late0 ? atomicAdd(&s_counter, 1u) : kNoClaim;
                    const uint32_t ni = nxt_idx < tn ? nxt_idx : 0u;  // unconditional loads: tn >= 1 here
                    nxt_far = far_t[ni];
                }
            }
            drain = drain || __ballot(gone) != 0;
        } else if (n_fetch == 0 && n_head == 0 && n_eval == 0) {
            break;  // no lane has anything left
        }
        // ---- the end of the tile: long walks are finished by the whole wave ----
        // When the tile has no position left to claim, the lanes of a wave fall idle one by one while a few walks
        // go on, and the workgroup lasts as long as its longest one -- hundreds of trips for a rare key in the
        // bucket of a frequent one.  A wave with at most kCoopLanes lanes still walking therefore takes over
        // every walk whose remaining steps (members per byte so far x bytes left) would take longer than
        // comparing the rest of the window directly, sixteen candidates per lane and round (wave_search: the
        // same candidates in the same order under the same rules, so the same answer).
        if (kCoopLanes != 0 && drain && (round & 3u) == 0 && n_head != 0 && n_head <= kCoopLanes) {
            {
                const uint32_t trips = st >> 17, dcur = po - so - noff, cov = dref - dcur;
                const bool coop = hd & (trips >= 2u) & (trips * dcur >= kCoopGain * (kCoopBase + (dcur >> 11)) * cov);
                uint64_t todo = __ballot(coop);
                while (todo) {
                    const int l = __ffsll((unsigned long long)todo) - 1;
                    todo &= todo - 1;
                    const uint32_t c_lo = (uint32_t)__builtin_amdgcn_readlane((int)(so + noff), l);
                    const uint32_t c_hi = (uint32_t)__builtin_amdgcn_readlane((int)((uint32_t)so_end + noff), l);
                    const uint32_t ppo = (uint32_t)__builtin_amdgcn_readlane((int)po, l);
                    const uint32_t pp0 = (uint32_t)__builtin_amdgcn_readlane((int)p0, l);
                    const uint32_t pcap = (uint32_t)__builtin_amdgcn_readlane((int)cap, l);
                    uint32_t wb = (uint32_t)__builtin_amdgcn_readlane((int)best, l);
                    uint32_t wbd = (uint32_t)__builtin_amdgcn_readlane((int)bestd, l);
                    uint32_t wt = (uint32_t)__builtin_amdgcn_readlane((int)termd, l);
                    wave_search(s_data, c_lo, c_hi, ppo, pp0, pcap, rec_min, max_len, tid & 63u, wb, wbd, wt);
                    if ((tid & 63u) == (uint32_t)l) {
                        best = wb;
                        bestd = wbd;
                        termd = wt;
                        hd = false;
                        ft = true;  // retired by the next visit of the claim block
                    }
                    MDBG2(if ((tid & 63u) == (uint32_t)l) atomicAdd(&g_mdbg[18], 1ull);)
                }
            }
        }
    }
    MDBG(const unsigned long long t_mine = wall_clock64(); __syncthreads(); if (tid == 0) {
        const unsigned long long t_end = wall_clock64();
        atomicAdd(&g_mdbg[20], t_end - t_begin);
        atomicAdd(&g_mdbg[21], t_staged - t_begin);
        atomicAdd(&g_mdbg[22], t_end - (s_t_exh ? s_t_exh : t_end));
        atomicAdd(&g_mdbg[23], t_end - t_mine);
        atomicAdd(&g_mdbg[24], 1ull);
    })
    if (emit_term) {
        __syncthreads();
        if (tid == 0) open_cnt[bid] = s_nopen;
    }
}

// Level 2: the positions level 1 handed on, as a per-tile list of (tile position | distance of the level-1
// candidate that ended the search << 16).  Candidates at or below that distance are not looked at.  Every
// candidate in reach matches at most four bytes, so the answer is the farthest candidate with four common bytes
// (more than five positions back: at distance four or five such a candidate ends the search instead), else
// the farthest with three.  Both are the first TRUE member -- hash collisions aside -- of a chain: the
// four-byte chain from far4[pos], then the three-byte chain from far3[pos], so the walks are a few steps long.
// The tile's bytes and window are staged in LDS as in level 1; the links of the four-byte index share their LDS
// space with those of the three-byte index, which replace them for a second pass over the positions the first
// pass could not settle (marked in the list itself).
constexpr uint32_t kOpenAgain = 0x80000000u;  // list entry: four-byte chain exhausted, three-byte chain to be walked
#ifndef ZG_LO_STEPS
#define ZG_LO_STEPS 4
#endif
#ifndef ZG_LO_FETCH_BATCH
#define ZG_LO_FETCH_BATCH 8
#endif
#ifndef ZG_LO_BUDGET
#define ZG_LO_BUDGET 64
#endif
#ifndef ZG_LO_THREE_FIRST_TENTHS
#define ZG_LO_THREE_FIRST_TENTHS 9  // ... this many positions in ten open
#endif
#ifndef ZG_LO_THREE_FIRST
#define ZG_LO_THREE_FIRST 1  // tiles with nine positions in ten open: three-byte pass before the four-byte pass; 0: never
#endif

// One pass of level 2 over a tile's list.  Walk lengths are very uneven (a rare key that shares its bucket with
// a frequent one walks through all of the latter's members), so, as in level 1, lanes claim list entries from
// an LDS counter whenever they finish one, and all lanes run branch-free steps in between.
//   kPass 1: four-byte chain; the first candidate more than max(5, excl) back that agrees in four bytes is the
//            answer; entries left without one are marked kOpenAgain.
//   kPass 2: three-byte chain, marked entries only; the first candidate good for anything decides.
#if defined(ZG_DEBUG_COUNTERS) || defined(ZG_DEBUG_PHASES)
__device__ unsigned long long g_dbg[64];
extern "C" int zg_debug_counters(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), sizeof(g_dbg));
}
#endif

// A walk that has used up its step budget (a rare key sharing its bucket with a frequent one: thousands of
// members, one lane busy while the rest of the tile has finished) is parked -- (list index, cursor) in s_park --
// and finished afterwards by a whole wave: the 64 lanes compare the window from the cursor upward directly
// with the position's bytes, which finds the same first agreement in a bounded number of passes.
#ifdef ZG_DEBUG_PHASES
__device__ long long g_t0[4096];
#endif
// kMode 0: the usual order (four-byte pass over every entry, three-byte pass over the entries it marks).
// kMode 2, 1: the other order, for tiles in which level 1 left almost every position open (incompressible data: a
//          position there has a three-byte candidate once in five hundred, a four-byte one next to never, yet each
//          pass walks its four colliding bucket members): the three-byte pass runs FIRST, over every entry, records
//          what it finds and marks the entry; the four-byte pass then looks at the marked entries only and, where it
//          finds a candidate, overrides the record.  The same answers: a candidate agreeing in four bytes agrees in
//          three, so an entry without a three-byte candidate has no four-byte one; four bytes beat three at any
//          distance; and a search the three-byte pass sees ended (distance <= length + 1) has no four-byte candidate
//          beyond distance five, for that one would have been the farther three-byte candidate.
//          The three-byte pass of that order (kMode 2) goes on along its chain after its first candidate, which
//          holds the four-byte candidates as well, and takes the first of those beyond distance five; only an
//          entry whose walk runs out of budget is left to a four-byte pass (kMode 1: marked entries only), and a
//          tile without such an entry is done after one pass.
template <int kPass, int kMode, uint32_t kParked, uint32_t kThreads>
__device__ __forceinline__ void lo_pass(const uint8_t *s_data, const uint16_t *s_next, uint32_t *s_counter,
                                        uint32_t *s_nparked, uint32_t *s_park, uint32_t *s_second, uint32_t *open_t,
                                        const uint32_t *far_t, uint32_t n_open, uint32_t *match_t, uint32_t pbase,
                                        uint32_t tail_room32, uint32_t min_len, uint32_t max_len) {
    constexpr int kSteps = ZG_LO_STEPS;
    constexpr uint32_t kFetchBatch = ZG_LO_FETCH_BATCH;
    constexpr uint32_t kBudgetTrips = (ZG_LO_BUDGET + kSteps - 1) / kSteps;
    uint32_t trips = 0;
    bool hd = false, ft = true, has_pos = false;
    uint32_t cur_i = 0, cur_e = 0, idx = 0, po = pbase, so = pbase, lim = 0, p0 = 0, excl = 0, res = 0;
    uint32_t nxt_i = atomicAdd(s_counter, 1u);
    uint32_t nxt_e = nxt_i < n_open ? open_t[nxt_i] : 0u;
    uint32_t nxt_f = nxt_i < n_open ? far_t[nxt_i] : 0u;
#ifdef ZG_DEBUG_PHASES
    unsigned long long dbg_trips = 0, dbg_heads = 0, dbg_thin = 0, dbg_steps = 0, dbg_claim = 0;
#endif
    for (;;) {
#ifdef ZG_DEBUG_PHASES
        const long long dbg_ta = clock64();
#endif
#pragma unroll
        for (int step = 0; step < kSteps; step++) {
            const uint32_t d = po - so;
            const uint32_t w = lds_ld4(s_data, so);
            const uint32_t nx = s_next[so];
            const bool stop_pre = d <= excl;
            bool fin;
            if (kPass == 1) {
                const bool hit = w == p0;
                res = (hd & !stop_pre & hit) ? ((4u << 16) | d) : res;
                fin = hd & (stop_pre | hit | (nx == 0u));
            } else if (kMode == 2) {
                const uint32_t x = w ^ p0;
                uint32_t m = ((uint32_t)__ffs((int)x) - 1u) >> 3;  // equal leading bytes; huge when all four agree
                m = m < lim ? m : lim;
                const bool live = hd & !stop_pre;
                const bool first = live & (m >= min_len) & (res == 0u);  // the farthest candidate of three bytes or four
                const bool ends = d <= m + 1u;                            // ... reached by its growing match: search over
                const bool four = live & (x == 0u) & (lim >= 4u) & (d > 5u);  // four bytes beyond distance five: the answer
                res = (first & !ends) ? ((m << 16) | d) : res;
                res = four ? ((4u << 16) | d) : res;
                fin = hd & (stop_pre | (first & ends) | four | (nx == 0u));
            } else {
                const uint32_t x = w ^ p0;
                uint32_t m = ((uint32_t)__ffs((int)x) - 1u) >> 3;  // equal leading bytes; huge when all four agree
                m = m < lim ? m : lim;
                const bool good = m >= min_len;
                const bool ends = d <= m + 1u;  // the growing match reaches pos: nothing recorded
                res = (hd & !stop_pre & good & !ends) ? ((m << 16) | d) : res;
                fin = hd & (stop_pre | good | (nx == 0u));
            }
#ifdef ZG_DEBUG_COUNTERS
            if (hd) atomicAdd(&g_dbg[(kPass - 1) * 16 + (idx & 15)], 1ull);
#endif
            so += (hd & !fin) ? nx : 0u;
            ft |= fin;
            hd &= !fin;
        }
        trips += 1;
        if (kMode == 2 && hd && trips >= kBudgetTrips && res != 0u) {
            // a three-byte candidate is recorded, the four-byte one is left to a pass over the four-byte chains
            match_t[idx] = res;
            open_t[cur_i] = cur_e | kOpenAgain;
            *s_second = 1u;
            hd = false;
            ft = true;
            has_pos = false;
        }
        if (hd && trips >= kBudgetTrips) {
            const uint32_t slot = atomicAdd(s_nparked, 1u);
            if (slot < kParked) {
                s_park[slot * 2 + 0] = cur_i;
                s_park[slot * 2 + 1] = so;
                hd = false;
                ft = true;
                has_pos = false;
            } else {
                trips = 0;  // no room: walk on
            }
        }
        const uint32_t n_head = (uint32_t)__popcll(__ballot(hd));
        const uint32_t n_fetch = (uint32_t)__popcll(__ballot(ft));
#ifdef ZG_DEBUG_PHASES
        dbg_trips += 1;
        dbg_heads += n_head;
        if (n_head != 0 && n_head <= 4) dbg_thin += 1;
        const long long dbg_tb = clock64();
        dbg_steps += (unsigned long long)(dbg_tb - dbg_ta);
#endif
        if (n_fetch >= kFetchBatch || (n_fetch != 0 && n_head == 0)) {
            if (ft) {
                // take the prefetched entry before anything younger is put in flight (see k_match)
                uint32_t ex = nxt_e, ef = nxt_f;
                asm volatile("" : "+v"(ex), "+v"(ef));
                if (has_pos) {
                    if (res) {
                        match_t[idx] = res;
                        if (kMode == 1 && kPass == 2) open_t[cur_i] = cur_e | kOpenAgain;  // the four-byte pass has a look
                    } else if (kMode == 0 && kPass == 1) {
                        open_t[cur_i] = cur_e | kOpenAgain;
                    }
                }
                has_pos = false;
                if (nxt_i >= n_open) {
                    ft = false;  // nothing left to claim
                } else {
                    cur_i = nxt_i;
                    cur_e = ex;
                    idx = ex & 0xFFFFu;
                    const uint32_t termd = (ex >> 16) & 0x1FFu;
                    const uint32_t remain32 = tail_room32 - idx;
                    const uint32_t cap = remain32 < max_len - 1 ? remain32 : max_len - 1;
                    lim = cap < 4u ? cap : 4u;
                    res = 0;
                    trips = 0;
                    // the reference searches only while pos < len - min_len, and records nothing below min_len
                    const bool valid = remain32 > min_len && lim >= min_len && ((kMode == 0 ? kPass == 1 : kPass == 2) || (ex & kOpenAgain));
                    const uint32_t fd = kPass == 1 ? (lim >= 4u ? (ef & 0xFFFFu) : 0u) : (ef >> 16);
                    has_pos = valid;
#ifdef ZG_DEBUG_COUNTERS
                    if (valid) atomicAdd(&g_dbg[32 + (kPass - 1) * 16 + (idx & 15)], 1ull);
#endif
                    if (valid && fd != 0) {
                        po = pbase + idx;
                        so = po - fd;
                        p0 = lds_ld4(s_data, po);
                        excl = min_len > termd ? min_len : termd;  // candidates lie more than this below pos
                        if (kPass == 1) excl = excl > 5u ? excl : 5u;  // four bytes at distance 4 or 5 end the search
                        hd = true;
                        ft = false;
                    }
                    nxt_i = atomicAdd(s_counter, 1u);
                    const uint32_t ni = nxt_i < n_open ? nxt_i : 0u;  // unconditional loads: n_open >= 1 here
                    nxt_e = open_t[ni];
                    nxt_f = far_t[ni];
                }
            }
        } else if (n_fetch == 0 && n_head == 0) {
            break;  // no lane has anything left
        }
#ifdef ZG_DEBUG_PHASES
        dbg_claim += (unsigned long long)(clock64() - dbg_tb);
#endif
    }
#ifdef ZG_DEBUG_PHASES
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&g_dbg[kPass == 1 ? 16 : 24], dbg_trips);
        atomicAdd(&g_dbg[kPass == 1 ? 17 : 25], dbg_heads);
        atomicAdd(&g_dbg[kPass == 1 ? 18 : 26], dbg_steps);
        atomicAdd(&g_dbg[kPass == 1 ? 19 : 27], dbg_claim);
        atomicAdd(&g_dbg[kPass == 1 ? 20 : 28], dbg_thin);
    }
#endif
    __syncthreads();
#ifdef ZG_DEBUG_PHASES
    if (threadIdx.x == 0) atomicAdd(&g_dbg[kPass == 1 ? 2 : 5], (unsigned long long)(clock64() - *reinterpret_cast<volatile long long *>(&g_t0[blockIdx.x & 4095])));
    if (threadIdx.x == 0) g_t0[blockIdx.x & 4095] = clock64();
    __syncthreads();
#endif

    // parked walks, dealt round-robin to the waves
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t n_raw = (uint32_t)__builtin_amdgcn_readfirstlane((int)*s_nparked);
    const uint32_t n_parked = n_raw < kParked ? n_raw : kParked;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const bool full = kPass == 1 || min_len >= 4u;  // four bytes have to agree, else three
    for (uint32_t k = wave; k < n_parked; k += kThreads / 64) {
        const uint32_t pi = s_park[k * 2 + 0], s_lo = s_park[k * 2 + 1];
        const uint32_t pe = open_t[pi];
        const uint32_t pidx = pe & 0xFFFFu, ptermd = (pe >> 16) & 0x1FFu;
        const uint32_t ppo = pbase + pidx;
        uint32_t pexcl = min_len > ptermd ? min_len : ptermd;  // candidates lie more than this below pos
        if (kPass == 1) pexcl = pexcl > 5u ? pexcl : 5u;
        const uint32_t s_hi = ppo - pexcl;                      // one past the last candidate
        const uint32_t pp0 = lds_ld4(s_data, ppo);
        uint32_t hit = 0xFFFFFFFFu;                             // lowest candidate agreeing in the wanted bytes
        // sixteen candidates per lane and round; rounds that touch an end of the range check each candidate
        const uint32_t a_lo = s_lo & ~15u;
        for (uint32_t b0 = a_lo; b0 < s_hi; b0 += 1024) {
            const uint32_t a = b0 + lane * 16;
            uint32_t j;
            if (b0 == a_lo || b0 + 1024 > s_hi)
                j = full ? scan16<true, true>(s_data, a, pp0, s_lo, s_hi) : scan16<true, false>(s_data, a, pp0, s_lo, s_hi);
            else
                j = full ? scan16<false, true>(s_data, a, pp0, s_lo, s_hi) : scan16<false, false>(s_data, a, pp0, s_lo, s_hi);
            // lowest hit over the wave: lanes hold ascending, disjoint ranges
            const uint64_t mh = __ballot(j < 16u);
            if (mh) {
                const int src = __ffsll((unsigned long long)mh) - 1;
                hit = (uint32_t)__shfl((int)(a + j), src);
                break;
            }
        }
        if (lane == 0) {
            if (kPass == 1) {
                if (hit != 0xFFFFFFFFu)
                    match_t[pidx] = (4u << 16) | (ppo - hit);
                else if (kMode == 0)
                    open_t[pi] = pe | kOpenAgain;
            } else if (hit != 0xFFFFFFFFu) {
                const uint32_t remain32 = tail_room32 - pidx;
                const uint32_t cap = remain32 < max_len - 1 ? remain32 : max_len - 1;
                const uint32_t plim = cap < 4u ? cap : 4u;
                const uint32_t hit_x = lds_ld4(s_data, hit) ^ pp0;
                uint32_t m = hit_x ? ((uint32_t)__builtin_ctz(hit_x) >> 3) : 4u;
                m = m < plim ? m : plim;
                const uint32_t d = ppo - hit;
                if (d > m + 1u) {  // else the growing match reaches pos
                    match_t[pidx] = (m << 16) | d;
                    if (kMode != 0) open_t[pi] = pe | kOpenAgain;
                }
            }
        }
    }
}

template <class G>
__global__ __launch_bounds__(G::lo_threads) void k_match_lo(const uint8_t *__restrict__ data, uint64_t first,
                                                             uint64_t end, uint64_t chunk_bytes,
                                                             const uint32_t *__restrict__ clen, uint32_t isolated,
                                                             uint32_t c0, uint32_t tiles_per_chunk, MatchParams mp,
                                                             const uint16_t *__restrict__ far4,
                                                             const uint16_t *__restrict__ next4,
                                                             const uint16_t *__restrict__ far3,
                                                             const uint16_t *__restrict__ next3, uint64_t base,
                                                             uint32_t *__restrict__ match,
                                                             uint32_t *__restrict__ open_list,
                                                             uint32_t *__restrict__ open_far,
                                                             const uint32_t *__restrict__ open_cnt) {
    __shared__ __attribute__((aligned(16))) uint8_t s_mem[G::data_lds + 2 * G::next_lds];
    __shared__ uint32_t s_park[G::parked * 2];  // list index, cursor
    __shared__ uint32_t s_counter;              // next unclaimed list entry
    __shared__ uint32_t s_nparked;
    __shared__ uint32_t s_second;               // three-byte pass first: an entry needs the four-byte pass after all
    uint8_t *const s_data = s_mem;
    uint16_t *const s_next = reinterpret_cast<uint16_t *>(s_mem + G::data_lds);

    const uint32_t bid = xcd_tile_index(blockIdx.x, gridDim.x);
    const uint32_t n_open = open_cnt[bid];
    if (n_open == 0) return;
    const uint32_t chunk = c0 + bid / tiles_per_chunk;
    const uint32_t tile = bid % tiles_per_chunk;
    const uint64_t cs = first + (uint64_t)chunk * chunk_bytes;
    const uint64_t ce = chunk_end(cs, chunk_bytes, end, clen, chunk);
    const uint64_t t0 = cs + (uint64_t)tile * G::tile;
    const uint64_t t1 = (t0 + G::tile < ce) ? t0 + G::tile : ce;
    uint64_t wlo = t0 > mp.window ? t0 - mp.window : 0;
    if (isolated && wlo < cs) wlo = cs;
    const uint64_t w0 = wlo & ~(uint64_t)15;
    const uint64_t w1 = (t1 + 8 < ce) ? t1 + 8 : ce;  // four bytes of look-ahead are all this level reads
    const uint32_t nbytes = (uint32_t)(w1 - w0);
    const uint32_t nlinks = (uint32_t)(t1 - w0);
    const uint32_t tid = threadIdx.x;
    uint32_t *open_t = open_list + (uint64_t)bid * G::tile;
    uint32_t *far_t = open_far + (uint64_t)bid * G::tile;
    uint32_t *match_t = match + (t0 - base);
    if (tid == 0) {
        s_counter = 0;
        s_nparked = 0;
        s_second = 0;
    }
    // the chain entries of the listed positions, beside the list: far4 | far3 << 16
    {
        const uint16_t *far4_t = far4 + (t0 - base), *far3_t = far3 + (t0 - base);
        for (uint32_t i = tid; i < n_open; i += G::lo_threads) {
            const uint32_t idx = open_t[i] & 0xFFFFu;
            far_t[i] = (uint32_t)far4_t[idx] | ((uint32_t)far3_t[idx] << 16);
        }
    }
#ifdef ZG_DEBUG_PHASES
    if (tid == 0) g_t0[blockIdx.x & 4095] = clock64();
#endif
    const uint32_t pbase = (uint32_t)(t0 - w0);
    const uint64_t tail_room = ce - t0;
    const uint32_t tail_room32 = tail_room < 0x7FFFFFFFull ? (uint32_t)tail_room : 0x7FFFFFFFu;
    // almost every position of the tile open: the three-byte pass first (see lo_pass)
    const bool three_first = ZG_LO_THREE_FIRST != 0 && mp.min_len == 3u && (uint64_t)n_open * 10u >= (t1 - t0) * (uint64_t)ZG_LO_THREE_FIRST_TENTHS;
    stage_tile(data, three_first ? next3 : next4, base, w0, nbytes, nlinks, s_data, s_next, tid, G::lo_threads, G::data_lds);
    __syncthreads();
#ifdef ZG_DEBUG_PHASES
    if (tid == 0) {
        atomicAdd(&g_dbg[1], (unsigned long long)(clock64() - g_t0[blockIdx.x & 4095]));
        g_t0[blockIdx.x & 4095] = clock64();
        atomicAdd(&g_dbg[0], 1ull);
        atomicAdd(&g_dbg[8], (unsigned long long)n_open);
    }
    __syncthreads();
#endif
    if (three_first)
        lo_pass<2, 2, G::parked, G::lo_threads>(s_data, s_next, &s_counter, &s_nparked, s_park, &s_second, open_t, far_t,
                                                n_open, match_t, pbase, tail_room32, mp.min_len, mp.max_len);
    else
        lo_pass<1, 0, G::parked, G::lo_threads>(s_data, s_next, &s_counter, &s_nparked, s_park, &s_second, open_t, far_t,
                                                n_open, match_t, pbase, tail_room32, mp.min_len, mp.max_len);
    __syncthreads();  // also orders the list entries written in the first pass before their readers in the second
    // the three-byte pass first, and every walk stayed within its budget: all answers are there
    if (three_first && s_second == 0u && s_nparked == 0u) return;
    __syncthreads();  // (the counters are read before they are cleared)
#ifdef ZG_DEBUG_PHASES
    if (tid == 0) {
        atomicAdd(&g_dbg[3], (unsigned long long)(clock64() - g_t0[blockIdx.x & 4095]));
        atomicAdd(&g_dbg[9], (unsigned long long)s_nparked);
        g_t0[blockIdx.x & 4095] = clock64();
    }
    __syncthreads();
#endif
    if (tid == 0) {
        s_counter = 0;
        s_nparked = 0;
    }
    stage_links(three_first ? next4 : next3, base, w0, nlinks, s_next, tid, G::lo_threads);
    __syncthreads();
#ifdef ZG_DEBUG_PHASES
    if (tid == 0) {
        atomicAdd(&g_dbg[4], (unsigned long long)(clock64() - g_t0[blockIdx.x & 4095]));
        g_t0[blockIdx.x & 4095] = clock64();
    }
    __syncthreads();
#endif
    if (three_first)
        lo_pass<1, 1, G::parked, G::lo_threads>(s_data, s_next, &s_counter, &s_nparked, s_park, &s_second, open_t, far_t,
                                                n_open, match_t, pbase, tail_room32, mp.min_len, mp.max_len);
    else
        lo_pass<2, 0, G::parked, G::lo_threads>(s_data, s_next, &s_counter, &s_nparked, s_park, &s_second, open_t, far_t,
                                                n_open, match_t, pbase, tail_room32, mp.min_len, mp.max_len);
#ifdef ZG_DEBUG_PHASES
    __syncthreads();
    if (tid == 0) {
        atomicAdd(&g_dbg[6], (unsigned long long)(clock64() - g_t0[blockIdx.x & 4095]));
        atomicAdd(&g_dbg[10], (unsigned long long)s_nparked);
    }
#endif
}

hipError_t launch_match(const ChunkGrid &g, uint32_t c0, uint32_t c1, MatchParams mp, uint32_t rec_min,
                        const uint16_t *far, const uint16_t *next, uint64_t base, uint32_t *match,
                        uint32_t *open_list, uint32_t *open_cnt, const uint8_t *walk_chunk, hipStream_t s) {
    if (c1 <= c0) return hipSuccess;
    const uint32_t tpc = match_tiles_per_chunk(g);
    const uint64_t nblocks = (uint64_t)(c1 - c0) * tpc;
    if (match_tile_positions(g) == kMatchRecordTile)
        hipLaunchKernelGGL(k_match<GeomRecord>, dim3((uint32_t)nblocks), dim3(GeomRecord::match_threads), 0, s, g.data,
                           g.first, g.end, g.chunk_bytes, g.chunk_len, g.isolated, c0, tpc, mp, rec_min,
                           open_list ? 1u : 0u, far, next, base, match, open_list, open_cnt, walk_chunk);
    else
        hipLaunchKernelGGL(k_match<GeomStream>, dim3((uint32_t)nblocks), dim3(GeomStream::match_threads), 0, s, g.data,
                           g.first, g.end, g.chunk_bytes, g.chunk_len, g.isolated, c0, tpc, mp, rec_min,
                           open_list ? 1u : 0u, far, next, base, match, open_list, open_cnt, walk_chunk);
    return hipGetLastError();
}

hipError_t launch_match_lo(const ChunkGrid &g, uint32_t c0, uint32_t c1, MatchParams mp, const uint16_t *far4,
                           const uint16_t *next4, const uint16_t *far3, const uint16_t *next3, uint64_t base,
                           uint32_t *match, uint32_t *open_list, uint32_t *open_far, const uint32_t *open_cnt,
                           hipStream_t s) {
    if (c1 <= c0) return hipSuccess;
    const uint32_t tpc = match_tiles_per_chunk(g);
    const uint64_t nblocks = (uint64_t)(c1 - c0) * tpc;
    if (match_tile_positions(g) == kMatchRecordTile)
        hipLaunchKernelGGL(k_match_lo<GeomRecord>, dim3((uint32_t)nblocks), dim3(GeomRecord::lo_threads), 0, s, g.data,
                           g.first, g.end, g.chunk_bytes, g.chunk_len, g.isolated, c0, tpc, mp, far4, next4, far3, next3,
                           base, match, open_list, open_far, open_cnt);
    else
        hipLaunchKernelGGL(k_match_lo<GeomStream>, dim3((uint32_t)nblocks), dim3(GeomStream::lo_threads), 0, s, g.data,
                           g.first, g.end, g.chunk_bytes, g.chunk_len, g.isolated, c0, tpc, mp, far4, next4, far3, next3,
                           base, match, open_list, open_far, open_cnt);
    return hipGetLastError();
}

}  // namespace zg
