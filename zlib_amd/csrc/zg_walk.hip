// k_walk: the longest-match search of LZ77Compressor::compress (reference src/lz77.rs:156-198) carried out only
// where the greedy parse can go.
//
// The parse of a chunk is the chain pos -> pos + max(1, L(pos)) from the chunk's first byte; positions it jumps
// over never need a match.  A workgroup takes a run of consecutive tiles of one chunk.  In every tile, walkers
// start at every kGap-th position (guesses) and at the positions where walkers of the previous tile left off
// (for the first tile of a chunk: its first byte, the one true start).  A walker searches the position it stands
// on, steps by the result and stops as soon as it steps on a position some walker has already visited: from there
// on the two chains are one.  The set V of visited positions is therefore closed under the step, and it contains
// the chunk's first byte, so it contains the whole greedy parse; guessed walkers join a chain that passes them
// within a unit or two, which keeps V close to the parse itself (text: a quarter of all positions, data with long
// matches: little more than the units).  Positions outside V get the value kUnvisited in match[].
//
// Two launches.  Mode 0 does the above; the positions reached beyond a run's last tile are left in exit_bits.
// Mode 1 starts walkers at those positions in the next run's first tile (where nobody knew them) and follows them
// -- into later tiles if need be -- until each has met a visited position.
//
// The search itself is k_match's (zg_match.hip), here along the FOUR-byte chains (links of tile and window in
// LDS): it settles every length of four or more.  A length of three (min_len == 3) is found beside it, for the
// same position and at the same time: when no candidate of four or more is recorded, the answer is the farthest
// candidate with three common bytes, unless that one ends the search (k_match_lo's rule), i.e. the first true
// member of the three-byte chain from far3[pos].  That walk takes a few steps (hash collisions only) and its links
// do not fit in LDS next to the four-byte links, so it reads them from global memory, one step per trip of the
// loop, each load issued a trip before it is used.  A walk that exceeds its budget is finished by the whole wave
// comparing the window directly, as in k_match_lo.
#include <cstdlib>

#include "zg_internal.h"
#include "zg_match_util.h"

#ifndef ZG_WALK_STEPS
#define ZG_WALK_STEPS 8
#endif
#ifndef ZG_WALK_LO_BUDGET
#define ZG_WALK_LO_BUDGET 32
#endif
#ifndef ZG_WALK_L1_BUDGET
#define ZG_WALK_L1_BUDGET 32
#endif
#ifndef ZG_WALK_SW_TRIPS
#define ZG_WALK_SW_TRIPS 4
#endif
#ifndef ZG_WALK_SW_EVERY
#define ZG_WALK_SW_EVERY 4
#endif
#ifndef ZG_WALK_SW_BATCH
#define ZG_WALK_SW_BATCH 4
#endif
#ifndef ZG_WALK_SW_REMAIN
#define ZG_WALK_SW_REMAIN 256
#endif
#ifndef ZG_WALK_WARM
#define ZG_WALK_WARM 1
#endif

namespace zg {

#ifdef ZG_WALK_DEBUG
__device__ unsigned long long g_wdbg[32];
extern "C" int zg_walk_debug_counters(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wdbg), sizeof(g_wdbg));
}
#define WDBG(...) __VA_ARGS__
#else
#define WDBG(...)
#endif

namespace {

constexpr uint32_t kNoPos = 0xFFFFFFFFu;

// lowest candidate in [s_lo, s_hi) whose first four (full) or three bytes equal pp0's, kNoPos if none; all
// arguments are uniform over the wave, which compares 1024 candidates per round
__device__ __forceinline__ uint32_t wave_scan(const uint8_t *s_data, uint32_t s_lo, uint32_t s_hi, uint32_t pp0,
                                              bool full, uint32_t lane) {
    uint32_t hit = kNoPos;
    const uint32_t a_lo = s_lo & ~15u;
    for (uint32_t b0 = a_lo; b0 < s_hi; b0 += 1024) {
        const uint32_t a = b0 + lane * 16;
        uint32_t j;
        if (b0 == a_lo || b0 + 1024 > s_hi)
            j = full ? scan16<true, true>(s_data, a, pp0, s_lo, s_hi) : scan16<true, false>(s_data, a, pp0, s_lo, s_hi);
        else
            j = full ? scan16<false, true>(s_data, a, pp0, s_lo, s_hi) : scan16<false, false>(s_data, a, pp0, s_lo, s_hi);
        const uint64_t mh = __ballot(j < 16u);
        if (mh) {  // lanes hold ascending, disjoint ranges
            const int src = __ffsll((unsigned long long)mh) - 1;
            hit = (uint32_t)__shfl((int)(a + j), src);
            break;
        }
    }
    return hit;
}

// reads n 16-bit elements from p on (one word per 128-byte line), to no purpose but to have them in the L2 when
// the walks ask for them; the values are only summed, so that no load is waited for before the caller's own
// loads have been issued
__device__ __forceinline__ uint32_t warm_lines(const uint16_t *p, uint32_t n, uint32_t tid, uint32_t nthreads) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p) & ~(uintptr_t)127;
    const uintptr_t e = reinterpret_cast<uintptr_t>(p + n);
    uint32_t acc = 0;
    for (uintptr_t q = a + (uintptr_t)tid * 128; q < e; q += (uintptr_t)nthreads * 128)
        acc += __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(q));
    return acc;
}

}  // namespace

// The chain level 1 walks: positions hashed by their first kWalkKey bytes.  Every candidate with that many common
// bytes is a member, so the walk settles all lengths >= kWalkKey; once the best length is B > kWalkKey, a
// candidate that beats it also agrees in bytes B - kWalkKey .. B - 1 and is found on the chain of position
// pos + B - kWalkKey (see k_match on changing chains).
constexpr uint32_t kWalkKey = 4;

template <class G>
__global__ __launch_bounds__(G::threads) void k_walk(
    const uint8_t *__restrict__ data, uint64_t first, uint64_t end, uint64_t chunk_bytes,
    const uint32_t *__restrict__ clen, uint32_t isolated, uint32_t c0, uint32_t tiles_per_run, uint32_t runs_per_chunk,
    MatchParams mp, uint32_t rec_min, uint32_t lo3, const uint16_t *__restrict__ far4,
    const uint16_t *__restrict__ next4, const uint16_t *__restrict__ far3, const uint16_t *__restrict__ next3,
    uint64_t base, uint32_t *match, uint32_t *__restrict__ exit_bits, uint32_t fix_mode, uint32_t gap_shift,
    const uint8_t *__restrict__ walk_chunk) {
    __shared__ __attribute__((aligned(16))) uint8_t s_mem[G::data_lds + 2 * G::next_lds];
    __shared__ uint32_t s_vis[G::tile / 32];       // positions of the tile some walker has stood on
    __shared__ uint32_t s_exit[kWalkExitWords];    // bit i: a walker left the tile for its position tn + i
    __shared__ uint16_t s_items[kWalkExitWords * 32];  // tile positions at which walkers of the previous tile arrive
    __shared__ uint32_t s_nitems;
    __shared__ uint32_t s_counter;  // next unclaimed start: arrivals first, then the guesses
    uint8_t *const s_data = s_mem;
    uint16_t *const s_next = reinterpret_cast<uint16_t *>(s_mem + G::data_lds);

    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t bid = xcd_tile_index(blockIdx.x, gridDim.x);
    const uint32_t chunk = c0 + bid / runs_per_chunk;
    const uint32_t run = bid % runs_per_chunk;
    if (fix_mode && run == 0) return;  // a chunk's first run starts at the chunk's first byte: nothing unknown
    if (walk_chunk && !walk_chunk[chunk - c0]) return;  // this chunk is searched in full by k_match
    const uint64_t cs = first + (uint64_t)chunk * chunk_bytes;
    const uint64_t ce = chunk_end(cs, chunk_bytes, end, clen, chunk);
    const uint32_t tiles_in_chunk = (uint32_t)((ce - cs + G::tile - 1) / G::tile);
    uint32_t tile = run * tiles_per_run;
    if (tile >= tiles_in_chunk) return;
    uint32_t tile_end = tile + tiles_per_run < tiles_in_chunk ? tile + tiles_per_run : tiles_in_chunk;
    if (fix_mode) tile_end = tiles_in_chunk;  // as far as a walker needs to go
    if (tid < kWalkExitWords) s_exit[tid] = fix_mode ? exit_bits[(uint64_t)(bid - 1) * kWalkExitWords + tid] : 0u;

    const uint32_t min_len = mp.min_len, max_len = mp.max_len;
    const bool scan_full3 = min_len >= 4u;

    for (; tile < tile_end; ++tile) {
        const uint64_t t0 = cs + (uint64_t)tile * G::tile;
        const uint64_t t1 = (t0 + G::tile < ce) ? t0 + G::tile : ce;
        const uint32_t tn = (uint32_t)(t1 - t0);
        uint32_t *match_t = match + (t0 - base);
        if (tid == 0) {
            s_nitems = 0;
            s_counter = G::threads;  // the first starts are dealt by thread number
        }
        __syncthreads();  // s_exit is complete: the walkers of the previous tile (or the loads above) are done
        if (tid < kWalkExitWords * 32) {
            const bool arrives = ((s_exit[tid >> 5] >> (tid & 31u)) & 1u) != 0u && tid < tn;
            if (arrives && (!fix_mode || match_t[tid] == kUnvisited)) s_items[atomicAdd(&s_nitems, 1u)] = (uint16_t)tid;
        }
        __syncthreads();
        const uint32_t n_items = s_nitems;
        if (fix_mode && n_items == 0) break;  // every arrival met a visited position: the chains are joined
        for (uint32_t i = tid; i < G::tile / 32; i += G::threads) s_vis[i] = 0;
        __syncthreads();  // everybody has read s_exit and s_nitems
        if (tid < kWalkExitWords) s_exit[tid] = 0;

        WDBG(const long long dbg_t_stage = clock64();)
        uint64_t wlo = t0 > mp.window ? t0 - mp.window : 0;
        if (isolated && wlo < cs) wlo = cs;       // nothing before the chunk's own first byte exists for it
        const uint64_t w0 = wlo & ~(uint64_t)15;  // LDS origin, 16-byte aligned
        const uint64_t w1 = (t1 + max_len < ce) ? t1 + max_len : ce;  // bytes needed: [w0, w1)
        const uint32_t nbytes = (uint32_t)(w1 - w0);
        const uint32_t nlinks = (uint32_t)(t1 - w0);
        const uint16_t *far4_t = far4 + (t0 - base), *far3_t = far3 + (t0 - base);
        const uint16_t *next3_w = next3 + (w0 - base);
#if ZG_WALK_WARM
        uint32_t warm = warm_lines(far4_t, tn, tid, G::threads);
        if (lo3) {
            warm += warm_lines(far3_t, tn, tid, G::threads);
            warm += warm_lines(next3_w, nlinks, tid, G::threads);
        }
#endif
        stage_tile(data, next4, base, w0, nbytes, nlinks, s_data, s_next, tid, G::threads, G::data_lds);
#if ZG_WALK_WARM
        asm volatile("" ::"v"(warm));
#endif
        if (fix_mode) {  // what earlier launches have visited
            for (uint32_t i0 = tid & ~63u; i0 < tn; i0 += G::threads) {
                const uint32_t i = i0 + lane;
                const uint64_t b = __ballot(i < tn && match_t[i] != kUnvisited);
                if (lane == 0) {
                    s_vis[i0 >> 5] = (uint32_t)b;
                    s_vis[(i0 >> 5) + 1] = (uint32_t)(b >> 32);
                }
            }
        }
        __syncthreads();

        WDBG(const long long dbg_t_loop = clock64();)
        WDBG(unsigned long long dbg_trips = 0, dbg_head = 0, dbg_l2 = 0, dbg_fetch = 0, dbg_fetchx = 0, dbg_vis = 0,
                                dbg_evalx = 0, dbg_c_head = 0, dbg_c_eval = 0, dbg_c_l2 = 0, dbg_c_fetch = 0, dbg_scan = 0,
                                dbg_l2steps = 0, dbg_thin = 0;)
        const uint32_t pbase = (uint32_t)(t0 - w0);   // LDS offset of tile position 0
        const uint32_t lo_off = (uint32_t)(wlo - w0);  // LDS offset of the lowest candidate there is
        const uint64_t tail_room = ce - t0;            // bytes from tile position 0 to the end of the chunk
        const uint32_t tail_room32 = tail_room < 0x7FFFFFFFull ? (uint32_t)tail_room : 0x7FFFFFFFu;

        constexpr int kHeadSteps = ZG_WALK_STEPS;
        constexpr uint32_t kSwitchTrips = ZG_WALK_SW_TRIPS, kSwitchEvery = ZG_WALK_SW_EVERY, kSwitchBatch = ZG_WALK_SW_BATCH,
                           kSwitchRemain = ZG_WALK_SW_REMAIN;
        constexpr uint32_t kLoBudget = ZG_WALK_LO_BUDGET, kL1Budget = ZG_WALK_L1_BUDGET;
        // What a lane is doing.  Level 1 (four-byte chain, LDS): hd walking, ev a candidate to be measured, l1f
        // finished.  Level 2 (three-byte chain, global links): a3 walking.  ft: the lane wants the retire-and-claim
        // block (a settled position, far values to take up, or no position yet).
        bool hd = false, ev = false, l1f = false, a3 = false, ft = true, has_pos = false, pend = false;
        uint32_t idx = 0, po = pbase, so = pbase, cap = 0, best = 0, bestd = 0, p0 = 0, p1 = 0, termd = 0;
        int32_t so_near = 0, so_end = 0;
        uint32_t toff = 0, pt = 0, tmask = 0xFFFFFFFFu, noff = 0, st = 0, round = 0;
        bool need_sw = false;
        uint32_t so3 = pbase, nx3 = 0, r3 = 0, lim = 0, n2 = 0, n1 = 0;  // n1, n2: trips of the two walks
        uint32_t nidx = 0, ld4 = 0, ld3 = 0;  // the position about to be taken up and its two far values
        uint64_t pf4 = 0, pf3 = 0;             // far values of positions idx + 1 .. idx + 4, 16 bits each
        // starts are claimed a visit ahead: `pre` is the lane's next start (the first ones are dealt by thread
        // number), `ci` the counter value of a claim under way
        uint32_t pre = kNoPos, ci = tid;
        bool claimed = true;

        for (;;) {
            WDBG(const long long dbg_ta = clock64();)
            uint32_t n_head = (uint32_t)__popcll(__ballot(hd));
            // ---- level 1: head steps (see k_match) ----
            if (n_head != 0) {
#pragma unroll
                for (int step = 0; step < kHeadSteps; step++) {
                    if (step != 0 && (step & 1) == 0 && !__ballot(hd)) break;  // nobody is walking any more
                    const bool nearby = (int32_t)so >= so_near;
                    const uint32_t w = lds_ld4(s_data, so + toff);
                    const uint32_t nx = s_next[so];
                    const bool guard_end = (int32_t)so >= so_end;
                    const bool survive = (((w ^ pt) & tmask) == 0) | nearby;
                    const bool to_adv = hd & !guard_end & !survive;
                    const bool to_done = hd & (guard_end | (!survive & (nx == 0)));
                    const bool to_eval = hd & !guard_end & survive;
                    so += to_adv ? nx : 0u;
                    hd = to_adv & !to_done;
                    ev |= to_eval;
                    l1f |= to_done;
                }
                n_head = (uint32_t)__popcll(__ballot(hd));
            }
            uint32_t n_eval = (uint32_t)__popcll(__ballot(ev));
            WDBG(const long long dbg_tb = clock64(); dbg_c_head += dbg_tb - dbg_ta; dbg_trips++; dbg_head += n_head;
                 dbg_evalx += n_eval != 0;)
            // ---- level 1: measure surviving candidates ----
            if (n_eval != 0) {
                if (ev) {
                    uint32_t m;
                    const uint32_t sc = so;
                    so += noff;
                    // the candidate's first eight bytes and its link in one round trip
                    const uint32_t x0 = lds_ld4(s_data, so) ^ p0;
                    const uint32_t x1 = lds_ld4(s_data, so + 4) ^ p1;
                    const uint32_t nxe = s_next[sc];
                    if (x0) {
                        m = (uint32_t)__builtin_ctz(x0) >> 3;
                    } else if (x1) {
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
                            // bytes m - 3 .. m of the position: out of its first eight when they reach that far
                            pt = m <= 7u ? (uint32_t)((((uint64_t)p1 << 32) | p0) >> ((m - 3u) * 8u))
                                         : lds_ld4(s_data, po + m - 3);
                            toff = m - 3 + noff;
                            tmask = 0xFFFFFFFFu;
                            finish = m == cap;
                            need_sw = m >= kWalkKey + 1;
                            st &= 0x1FFFFu;
                        }
                    }
                    so = sc + nxe;
                    ev = false;
                    l1f = finish || nxe == 0;
                    hd = !l1f;
                }
                n_head = (uint32_t)__popcll(__ballot(hd));
            }
            // ---- level 1: long walks move to the chain the best length so far selects (see k_match) ----
            st += 1u << 17;
            ++round;
            if ((round & (kSwitchEvery - 1)) == 0 && n_head != 0) {
                const uint32_t trips = st >> 17, dcur = po - so - noff, cov = bestd - dcur;
                const bool sw = hd & need_sw & (trips >= kSwitchTrips) &
                                (trips * (uint32_t)kHeadSteps * (dcur >> 4) >= kSwitchRemain * (cov >> 4));
                const uint32_t n_sw = (uint32_t)__popcll(__ballot(sw));
                if (n_sw >= kSwitchBatch || (n_sw != 0 && n_sw * 4 >= n_head)) {
                    if (sw) {
                        const uint32_t o_old = st & 0xFFu, tchk = (st >> 8) & 0x1FFu;
                        const uint32_t room = pbase + tn - po;  // links are staged up to the end of the tile
                        uint32_t o_new = best - kWalkKey;
                        o_new = o_new < room ? o_new : room;
                        o_new = o_new < 255u ? o_new : 255u;
                        if (o_new > o_old) {
                            int32_t end_u = so_end - (int32_t)o_old;
                            // a skipped candidate at distance d ends the search iff it matches max(rec_min, d - 1)
                            // bytes; it then agrees with the position in its first four bytes at least
                            const uint32_t d_lo = tchk > min_len ? tchk : min_len;
                            const uint32_t d_hi = best < po - lo_off ? best : po - lo_off;
                            for (uint32_t d = d_hi; d > d_lo; --d) {
                                if (lds_ld4(s_data, po - d) != p0) continue;
                                const uint32_t need = d - 1u > rec_min ? d - 1u : rec_min;
                                if (need <= cap && (int32_t)(po - d) < end_u &&
                                    lds_lcp(s_data, po - d, po, need) >= need) {
                                    end_u = (int32_t)(po - d);
                                    break;
                                }
                            }
                            so = po - bestd + o_new;
                            so_near = (int32_t)po - (int32_t)max_len + (int32_t)o_new;
                            so_end = end_u + (int32_t)o_new;
                            noff = 0u - o_new;
                            toff = best - 3u - o_new;
                            st = (best << 8) | o_new;
                        }
                        need_sw = false;
                    }
                    n_head = (uint32_t)__popcll(__ballot(hd));
                }
            }
            // ---- level 1: a walk that has run long (a frequent key) is finished by the whole wave, which compares
            // the rest of the window directly: bounded time where the chain may have thousands of members ----
            {
                n1 += hd ? 1u : 0u;
                uint64_t longw = __ballot(hd & (n1 >= kL1Budget));
                while (longw) {
                    WDBG(dbg_scan++;)
                    const int ldr = __ffsll((unsigned long long)longw) - 1;
                    longw &= longw - 1;
                    const uint32_t ppo = (uint32_t)__shfl((int)po, ldr), pp0 = (uint32_t)__shfl((int)p0, ldr);
                    const uint32_t c_lo = (uint32_t)__shfl((int)(so + noff), ldr);          // the candidate itself
                    const uint32_t c_hi = (uint32_t)__shfl((int)(so_end + (int32_t)noff), ldr);  // loop guard
                    const uint32_t pcap = (uint32_t)__shfl((int)cap, ldr);
                    uint32_t wb = (uint32_t)__shfl((int)best, ldr), wbd = (uint32_t)__shfl((int)bestd, ldr), wt = 0;
                    wave_search(s_data, c_lo, c_hi, ppo, pp0, pcap, rec_min, max_len, lane, wb, wbd, wt);
                    if ((int)lane == ldr) {
                        best = wb;
                        bestd = wbd;
                        termd = wt ? wt : termd;
                        hd = false;
                        l1f = true;
                    }
                }
                n_head = (uint32_t)__popcll(__ballot(hd));
            }
            WDBG(const long long dbg_tc = clock64(); dbg_c_eval += dbg_tc - dbg_tb;)
            // ==== everything that touches global memory happens from here to the end of the trip: what was asked
            // for a trip ago is used first, then this trip's stores and loads are issued, so that a wave waits for
            // memory once per trip at most ====
            // ---- level 2: one step on the three-byte chain with the link asked for a trip ago ----
            a3 &= !(l1f & (best != 0u));  // a length of rec_min or more makes it moot
            // ... unless no lane has anything to exchange: a wave left with one long level-1 walk goes round its
            // head steps without paying for the memory round trip
            if (!__ballot(a3 | ft | (has_pos & l1f & !hd & !ev))) {
                WDBG(dbg_c_l2 += clock64() - dbg_tc;)
                if (n_head == 0 && n_eval == 0) break;  // no lane has anything left
                continue;
            }
            bool want_link = false, want_far = false, want_store = false;  // this trip's memory operations
            uint32_t store_idx = 0, store_val = 0;
            uint32_t n_l2 = (uint32_t)__popcll(__ballot(a3));
            WDBG(dbg_l2 += n_l2; dbg_l2steps += n_l2; dbg_thin += (n_head + n_l2 + n_eval) <= 8;)
            if (n_l2 != 0) {
                if (a3) {
                    const uint32_t d = po - so3;
                    const uint32_t x = lds_ld4(s_data, so3) ^ p0;
                    const uint32_t nx = nx3;
                    uint32_t m = ((uint32_t)__ffs((int)x) - 1u) >> 3;  // equal leading bytes; huge when all four agree
                    m = m < lim ? m : lim;
                    const bool stop = d <= min_len;
                    const bool good = !stop & (m >= min_len);
                    r3 = good ? ((m << 16) | d) : r3;
                    const bool fin = stop | good | (nx == 0u);
                    so3 += fin ? 0u : nx;
                    want_link = !fin;
                    a3 = !fin;
                }
                n2 += 1;
                // walks out of budget (a rare key in the bucket of a frequent one): the wave compares the window
                uint64_t over = __ballot(a3 & (n2 >= kLoBudget));
                while (over) {
                    WDBG(dbg_scan++;)
                    const int ldr = __ffsll((unsigned long long)over) - 1;
                    over &= over - 1;
                    const uint32_t ppo = (uint32_t)__shfl((int)po, ldr), pp0 = (uint32_t)__shfl((int)p0, ldr);
                    const uint32_t s_lo = (uint32_t)__shfl((int)so3, ldr);
                    const uint32_t hit = wave_scan(s_data, s_lo, ppo - min_len, pp0, scan_full3, lane);
                    if ((int)lane == ldr) {
                        if (hit != kNoPos) {
                            const uint32_t hx = lds_ld4(s_data, hit) ^ p0;
                            uint32_t m = hx ? ((uint32_t)__builtin_ctz(hx) >> 3) : 4u;
                            m = m < lim ? m : lim;
                            r3 = (m << 16) | (po - hit);
                        }
                        a3 = false;
                        want_link = false;
                    }
                }
                n_l2 = (uint32_t)__popcll(__ballot(a3));
            }
            // ---- a position is settled once level 1 has finished and, unless it found a length, level 2 too ----
            {
                const bool settled = has_pos & !ft & l1f & !hd & !ev & ((best != 0u) | !a3);
                if (settled) {
                    if (best == 0u) {
                        const uint32_t e3 = min_len > termd ? min_len : termd;
                        const uint32_t d3 = r3 & 0xFFFFu, m3 = r3 >> 16;
                        if (d3 > e3 && d3 > m3 + 1u) {  // else the growing match reaches pos: nothing recorded
                            best = m3;
                            bestd = d3;
                        }
                    }
                    a3 = false;
                    ft = true;
                }
            }
            // ---- retire settled positions, step, claim starts, take up the next position ----
            const uint32_t n_fetch = (uint32_t)__popcll(__ballot(ft));
            WDBG(const long long dbg_td = clock64(); dbg_c_l2 += dbg_td - dbg_tc;)
            bool want_pf = false, want_claim = false;
            if (n_fetch != 0) {
                WDBG(dbg_fetch += n_fetch; dbg_fetchx++; dbg_vis += __popcll(__ballot(ft & pend));)
                // Which position does the lane take up, are its far values at hand, must it be checked for an
                // earlier visit?  Then the LDS operations of all these lanes go out together (visit mark, the
                // position's first eight bytes, the claim of another start) and are used after one wait.
                uint32_t cand = kNoPos, f4 = 0, f3 = 0;
                bool have_far = false, need_vis = false;
                if (ft) {
                    if (pend) {  // the far values asked for on the previous visit
                        cand = nidx;
                        f4 = ld4;
                        f3 = ld3;
                        have_far = true;
                        pend = false;
                    } else {
                        if (claimed) {  // the start claimed on an earlier visit
                            claimed = false;
                            pre = kNoPos;
                            if (ci < n_items) {
                                pre = s_items[ci];
                            } else if (!fix_mode) {
                                const uint32_t k = ci - n_items;
                                if (k < (G::tile >> gap_shift) && (k << gap_shift) < tn) pre = k << gap_shift;
                            }
                        }
                        uint32_t q = kNoPos;  // where this lane's walker stands next
                        if (has_pos) {
                            want_store = true;
                            store_idx = idx;
                            store_val = best ? ((best << 16) | bestd) : 0u;
                            has_pos = false;
                            const uint32_t step = best ? best : 1u;
                            q = idx + step;
                            if (q >= tn) {  // out of the tile: the next tile's business, if the chunk goes on
                                const uint32_t e = q - tn;
                                if (t0 + q < ce) atomicOr(&s_exit[e >> 5], 1u << (e & 31u));
                                q = kNoPos;
                            } else if (step <= 4u) {  // the far values of the next four positions came with this one's
                                const uint32_t sh = (step - 1u) * 16u;
                                f4 = (uint32_t)(pf4 >> sh) & 0xFFFFu;
                                f3 = (uint32_t)(pf3 >> sh) & 0xFFFFu;
                                have_far = true;
                            }
                        }
                        if (q == kNoPos) {  // a new walker: arrivals from the previous tile first, then the guesses
                            have_far = false;
                            q = pre;
                            pre = kNoPos;
                            if (q != kNoPos)
                                want_claim = true;
                            else
                                ft = false;  // no starts left: this lane is done
                        }
                        cand = q;
                        need_vis = q != kNoPos;
                    }
                }
                uint32_t old = 0, np0 = 0, np1 = 0;
                const uint32_t vbit = 1u << (cand & 31u);
                if (cand != kNoPos) {
                    if (need_vis) old = atomicOr(&s_vis[cand >> 5], vbit);
                    np0 = lds_ld4(s_data, pbase + cand);
                    np1 = lds_ld4(s_data, pbase + cand + 4);
                }
                if (want_claim) {
                    ci = atomicAdd(&s_counter, 1u);
                    claimed = true;
                }
                if (cand != kNoPos && !(need_vis && (old & vbit))) {  // else: somebody has been here, the walker ends
                    if (have_far) {
                        idx = cand;
                        const uint32_t remain32 = tail_room32 - idx;  // bytes to the end of the chunk, saturated
                        cap = remain32 < max_len - 1 ? remain32 : max_len - 1;
                        best = 0;
                        bestd = 0;
                        termd = 0;
                        r3 = 0;
                        n2 = 0;
                        n1 = 0;
                        need_sw = false;
                        has_pos = true;
                        ft = false;
                        po = pbase + idx;
                        p0 = np0;
                        p1 = np1;
                        want_pf = true;
                        // the reference searches only while pos < len - min_len, and records nothing below min_len
                        const bool searchable = remain32 > min_len;
                        if (f4 != 0 && searchable && cap >= rec_min) {
                            so = po - f4;
                            so_near = (int32_t)po - (int32_t)max_len;
                            so_end = (int32_t)po - (int32_t)min_len;
                            toff = 0;
                            pt = p0;
                            tmask = 0xFFFFFFFFu;
                            noff = 0;
                            st = 0;
                            hd = true;
                            l1f = false;
                        } else {
                            l1f = true;
                        }
                        lim = cap < 4u ? cap : 4u;
                        if (lo3 && searchable && lim >= min_len && f3 != 0) {
                            so3 = po - f3;
                            want_link = true;
                            a3 = true;
                        }
                    } else {
                        nidx = cand;
                        want_far = true;
                        pend = true;
                    }
                }
            } else if (n_head == 0 && n_eval == 0 && n_l2 == 0) {
                break;  // no lane has anything left
            }
            // This trip's store and loads, after everything that used last trip's.  The loads are issued by every
            // lane on every trip, wanted or not (the addresses are always valid: cursors and positions of the tile),
            // as straight-line code: a load under a branch makes the compiler wait for it at the end of the branch.
            // Their results are looked at on the next trip, by the lanes that wanted them.
            // Lanes that want nothing all read element 0, which costs the memory pipeline one access for the lot.
            (void)want_pf;
            if (want_store) match_t[store_idx] = store_val;
            nx3 = next3_w[want_link ? so3 : 0u];
            const uint32_t far_at = want_far ? nidx : 0u, pf_at = has_pos ? idx : 0u;
            ld4 = far4_t[far_at];
            ld3 = far3_t[far_at];
            __builtin_memcpy(&pf4, far4_t + pf_at + 1, 8);  // the four positions behind this one (the arrays have slack)
            __builtin_memcpy(&pf3, far3_t + pf_at + 1, 8);
            WDBG(dbg_c_fetch += clock64() - dbg_td;)
        }
        WDBG(if (lane == 0) {
            atomicAdd(&g_wdbg[4], dbg_trips); atomicAdd(&g_wdbg[5], dbg_head); atomicAdd(&g_wdbg[6], dbg_l2);
            atomicAdd(&g_wdbg[7], dbg_fetch); atomicAdd(&g_wdbg[8], dbg_fetchx); atomicAdd(&g_wdbg[9], dbg_vis);
            atomicAdd(&g_wdbg[10], dbg_evalx); atomicAdd(&g_wdbg[11], dbg_c_head); atomicAdd(&g_wdbg[12], dbg_c_eval);
            atomicAdd(&g_wdbg[13], dbg_c_l2); atomicAdd(&g_wdbg[14], dbg_c_fetch); atomicAdd(&g_wdbg[15], dbg_scan);
            atomicAdd(&g_wdbg[16], dbg_thin); atomicMax(&g_wdbg[17], dbg_trips); atomicAdd(&g_wdbg[18], dbg_l2steps);
            atomicAdd(&g_wdbg[19 + (fix_mode ? 1 : 0)], 1ull);
        })
        WDBG(const long long dbg_t_end = clock64();)
        __syncthreads();
        WDBG(if (tid == 0) {
            atomicAdd(&g_wdbg[0], 1ull); atomicAdd(&g_wdbg[1], (unsigned long long)(dbg_t_loop - dbg_t_stage));
            atomicAdd(&g_wdbg[2], (unsigned long long)(clock64() - dbg_t_loop));
            atomicAdd(&g_wdbg[3], (unsigned long long)(clock64() - dbg_t_end));
        })
        if (!fix_mode) {  // positions nobody stood on
            for (uint32_t i = tid; i < tn; i += G::threads)
                if (!((s_vis[i >> 5] >> (i & 31u)) & 1u)) match_t[i] = kUnvisited;
        }
    }
    if (!fix_mode) {
        __syncthreads();
        if (tid < kWalkExitWords) exit_bits[(uint64_t)bid * kWalkExitWords + tid] = s_exit[tid];
    }
}

uint32_t walk_tiles_per_run(const ChunkGrid &g, uint32_t n_chunks, uint32_t cus) {
    // runs are what the machine is filled with: at least eight per CU when the input is long enough for that,
    // and as long as that allows (the first tile of every run but a chunk's first is walked from guesses only
    // and mended by the second launch)
    const uint32_t tpc = match_tiles_per_chunk(g);
    const uint64_t tiles = (uint64_t)n_chunks * tpc;
    uint64_t r = tiles / (8ull * cus);
    if (r < 1) r = 1;
    if (r > 16) r = 16;
    if (r > tpc) r = tpc;
    return (uint32_t)r;
}

// Which search for a chunk?  The search along the parse pays where chains are long AND matches are long, so that
// the parse stands on few positions, each with a long search (log lines, records of a fixed layout with text in
// them: 4 x on generated log lines); with long chains and short matches (packed binary records) the search of every
// position is as fast now that its long walks change to sparser chains, and where the first candidate already
// matches everything there is to match (runs, short periods) every search ends at once.  So a chunk goes along the parse
// when, over 256 sampled positions, the mean estimated chain length exceeds ZG_PROBE_THRESHOLD and the mean of the longest
// common prefix with one of the chain's first eight members (counted up to kProbeLcp bytes) lies in
// [ZG_PROBE_LCP_LO, ZG_PROBE_LCP_HI).
#ifndef ZG_PROBE_THRESHOLD
#define ZG_PROBE_THRESHOLD 96
#endif
#ifndef ZG_PROBE_LCP_LO
#define ZG_PROBE_LCP_LO 16
#endif
#ifndef ZG_PROBE_LCP_HI
#define ZG_PROBE_LCP_HI 56
#endif
constexpr uint32_t kProbeSteps = 8;  // members of a sampled chain that are walked; the rest is extrapolated
constexpr uint32_t kProbeLcp = 64;

__global__ __launch_bounds__(256) void k_chunk_probe(const uint8_t *__restrict__ data, uint64_t first, uint64_t end,
                                                      uint64_t chunk_bytes, const uint32_t *__restrict__ clen,
                                                      uint32_t c0, const uint16_t *__restrict__ far5,
                                                      const uint16_t *__restrict__ next5, uint64_t base,
                                                      uint8_t *__restrict__ walk_chunk, uint32_t threshold,
                                                      uint32_t lcp_lo, uint32_t lcp_hi) {
    __shared__ uint32_t s_sum[4], s_lcp[4];
    const uint32_t ci = blockIdx.x, tid = threadIdx.x;
    const uint64_t cs = first + (uint64_t)(c0 + ci) * chunk_bytes;
    const uint64_t ce = chunk_end(cs, chunk_bytes, end, clen, c0 + ci);
    const uint64_t len = ce - cs;
    uint32_t steps = 0;  // (estimated) members of the sampled position's chain inside its window
    uint32_t longest = 0;  // most bytes (up to kProbeLcp) one of the chain's first members has in common with the position
    if (len >= 256) {
        // 256 positions spread over the chunk, each moved by a pseudo-random amount inside its stretch: evenly
        // spaced samples would all fall on the same field of fixed-size records
        const uint64_t stretch = len / 256;
        const uint64_t p = cs + stretch * tid + ((tid * 2654435761u) >> 8) % stretch;
        const uint32_t fd = far5[p - base];
        const bool measure = p + kProbeLcp + 8 <= ce;
        uint64_t s = p - fd;
        while (fd != 0 && s < p && steps < kProbeSteps) {
            steps++;
            if (measure) {
                uint32_t m = 0;
                while (m < kProbeLcp) {
                    uint64_t x, y;
                    __builtin_memcpy(&x, data + s + m, 8);
                    __builtin_memcpy(&y, data + p + m, 8);
                    if (x != y) {
                        m += (uint32_t)__builtin_ctzll(x ^ y) >> 3;
                        break;
                    }
                    m += 8;
                }
                m = m < kProbeLcp ? m : kProbeLcp;
                longest = m > longest ? m : longest;
            }
            const uint32_t nx = next5[s - base];
            if (nx == 0) break;
            s += nx;
        }
        // a chain that goes on is taken to keep the spacing of its first members up to the position
        if (steps == kProbeSteps && s < p) steps = (uint32_t)((uint64_t)fd * kProbeSteps / (s - (p - fd)));
    }
    for (int off = 32; off > 0; off >>= 1) {
        steps += __shfl_down(steps, off);
        longest += __shfl_down(longest, off);
    }
    if ((tid & 63) == 0) {
        s_sum[tid >> 6] = steps;
        s_lcp[tid >> 6] = longest;
    }
    __syncthreads();
    if (tid == 0) {
        const uint32_t chain = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
        const uint32_t lcp = s_lcp[0] + s_lcp[1] + s_lcp[2] + s_lcp[3];
        walk_chunk[ci] = (chain > threshold * 256u && lcp >= lcp_lo * 256u && lcp < lcp_hi * 256u) ? 1 : 0;
    }
}
hipError_t launch_chunk_probe(const ChunkGrid &g, uint32_t c0, uint32_t c1, const uint16_t *far5,
                              const uint16_t *next5, uint64_t base, uint8_t *walk_chunk, hipStream_t s) {
    if (c1 <= c0) return hipSuccess;
    static const uint32_t threshold = [] {
        const char *e = std::getenv("ZLIB_AMD_PROBE_THRESHOLD");  // tuning hook
        const int v = e ? atoi(e) : 0;
        return (uint32_t)(v > 0 ? v : ZG_PROBE_THRESHOLD);
    }();
    hipLaunchKernelGGL(k_chunk_probe, dim3(c1 - c0), dim3(256), 0, s, g.data, g.first, g.end, g.chunk_bytes,
                       g.chunk_len, c0, far5, next5, base, walk_chunk, threshold, (uint32_t)ZG_PROBE_LCP_LO,
                       (uint32_t)ZG_PROBE_LCP_HI);
    return hipGetLastError();
}

hipError_t launch_walk(const ChunkGrid &g, uint32_t c0, uint32_t c1, MatchParams mp, const uint16_t *far4,
                       const uint16_t *next4, const uint16_t *far3, const uint16_t *next3, uint64_t base,
                       uint32_t *match, uint32_t *exit_bits, uint32_t tiles_per_run, const uint8_t *walk_chunk,
                       hipStream_t s) {
    if (c1 <= c0) return hipSuccess;
    const uint32_t tpc = match_tiles_per_chunk(g);
    const uint32_t rpc = (tpc + tiles_per_run - 1) / tiles_per_run;
    const uint64_t nblocks = (uint64_t)(c1 - c0) * rpc;
    const uint32_t rec_min = mp.min_len > kWalkKey ? mp.min_len : kWalkKey;
    const uint32_t lo3 = mp.min_len < kWalkKey ? 1u : 0u;  // lengths of three: the three-byte chain beside level 1
    static const uint32_t gap_shift = [] {
        const char *e = std::getenv("ZLIB_AMD_WALK_GAP_SHIFT");  // tuning hook: walkers start every 2^shift positions
        const int v = e ? atoi(e) : 0;
        return (uint32_t)((v >= 1 && v <= 14) ? v : 4);
    }();
    for (uint32_t mode = 0; mode < 2; mode++) {
        if (mode == 1 && rpc == 1) break;  // every run starts a chunk
        hipLaunchKernelGGL(k_walk<GeomStream>, dim3((uint32_t)nblocks), dim3(GeomStream::threads), 0, s, g.data, g.first,
                           g.end, g.chunk_bytes, g.chunk_len, g.isolated, c0, tiles_per_run, rpc, mp, rec_min, lo3, far4,
                           next4, far3, next3, base, match, exit_bits, mode, gap_shift, walk_chunk);
    }
    return hipGetLastError();
}

}  // namespace zg
