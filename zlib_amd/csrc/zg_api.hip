// Host orchestration and the C ABI (include/zlib_amd.h).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "zg_internal.h"

using namespace zg;


namespace {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + (bytes >> 3) + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            want = bytes + 256;
            e = hipMalloc(&p, want);
        }
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

enum KernelId { K_INDEX, K_WALK, K_MATCH, K_MATCH_LO, K_PARSE, K_HUFFMAN, K_TILEBITS, K_OFFSETS, K_ENCODE, K_ADLER, K_STORED, K_INFLATE, K_MISC, K_COUNT };
const char *const kKernelNames[K_COUNT] = {"k_index", "k_walk", "k_match", "k_match_lo", "k_parse", "k_huffman", "k_tile_bits",
                                           "k_offsets", "k_encode", "k_adler", "k_stored", "k_inflate", "misc"};

// device scalars
struct Scalars {
    uint64_t carry;       // absolute bit offset of the next block
    uint64_t out_bytes;   // final size written by k_trailer
    uint32_t adler[2];    // running a, b
    uint32_t status;      // bit 0: output capacity exceeded, bit 1: index exchange out of order, bit 2: the parse
                          // stood on a position the walk kernel had not searched
    uint32_t pad;
    uint64_t carry2;      // record batches: the passes' running byte offset alternates between carry and carry2
};

// Scratch of zg_adler32_device_noalloc: a fixed array in the code object, of which every GPU the library runs
// on holds its own copy until the process ends.  Words [0, 2 * kNoAllocSegs): the (a, b) of every 256 KiB
// segment; behind them 3 * 64-bit words per 1024 segments for the first stage of the fold.  Host code never
// uses the array by name: zg_init asks the runtime for its address on the context's device and keeps that.
constexpr uint64_t kNoAllocSeg = 256 * 1024;
constexpr uint64_t kNoAllocMax = ZG_ADLER32_NOALLOC_MAX;
constexpr uint64_t kNoAllocSegs = kNoAllocMax / kNoAllocSeg;
constexpr uint64_t kNoAllocWords64 = kNoAllocSegs + 3 * (kNoAllocSegs / 1024 + 1);  // 64-bit words in all
static __device__ __align__(256) uint64_t g_noalloc_scratch[kNoAllocWords64];
// one lock per GPU (a fixed array: taking it allocates nothing): calls on two contexts of one device take turns at
// that device's copy of the array
constexpr int kNoAllocLocks = 64;
static std::mutex g_noalloc_mu[kNoAllocLocks];

}  // namespace

struct zg_ctx {
    int device = 0;
    uint64_t *noalloc_scratch = nullptr;  // device address of g_noalloc_scratch on `device`
    hipStream_t stream = nullptr;
    hipStream_t copy = nullptr;  // uploads of host input, overlapped with the kernels on `stream`
    std::mutex mu;
    std::string err;
    DevBuf inf_tok, inf_ntok, inf_units, inf_win, block_off_all, walk_exit, walk_chunk, shard_buf, far, next, far4, next4, far3, next3, open_list, open_far, open_cnt, rec_len, rec_off, inf_off, inf_len, inf_status, match, tokens, ntok, freq, tab, hdr, hdr_bits, block_bits, bit_off, tile_bits, adler_part,
        scalars, stage_in, stage_out, misc, flag, sub_exit, sub_exit2, merged, tile_cnt, adler_blk;
    // shard state kept between zg_shard_begin and zg_shard_finish
    bool shard_active = false;
    uint64_t shard_total_bits = 0;
    hipEvent_t inf_ev[4] = {nullptr, nullptr, nullptr, nullptr};  // zg_decompress_blocks_device: two passes in flight;
                                                                   // record batches: the checksums' stream
    // state of a piecewise shard run (zg_shard_open / zg_shard_next)
    struct ShardRun {
        bool open = false;
        ChunkGrid g{};
        int strategy = 0, last_rank = 0;
        uint32_t max_tiles = 0, bc = 0;
        uint32_t queued = 0, reported = 0, n_batches = 0;  // batches whose kernels are queued / whose piece was handed out
        uint32_t *packed = nullptr;
        uint64_t cap_bits = 0, bytes_reported = 0;
        size_t halo = 0, n = 0;
        std::vector<hipEvent_t> ev;  // one per batch: its bit total has landed in h_carry
        uint64_t *h_carry = nullptr;  // pinned: bits of all blocks up to and including batch b
        size_t h_carry_cap = 0;
    } run;
    // set once the exchange-based index kernel has reported an out-of-order exchange: use k_index from then on
    bool index_safe = false;
    // How a chunk's matches are found: kWalkAuto (default) samples the chunk's five-byte chains and sends chunks
    // with long chains (structured data) to the search along the parse (k_walk), the others to the search of
    // every position (k_match, k_match_lo), which is faster on text-like and incompressible input;
    // ZLIB_AMD_WALK=1 / 0 forces one way for every chunk.  walk_off is also set should the parse ever stand on an
    // unsearched position (never seen; the check is cheap).
    int walk_mode = 2;  // 0 never, 1 always, 2 by probe
    bool walk_off = false;
    int cus = 256;
    // bit offset of every block written by the last zg_compress(_device) / zg_shard_begin call (zg_last_block_offsets)
    std::vector<uint64_t> block_offs;
    // profiling
    bool prof = false;
    int prof_only = -1;  // >= 0: only launches of this kernel id are timed
    struct Ev {
        int id;
        hipEvent_t a, b;
    };
    std::vector<Ev> events;
    double prof_ms[K_COUNT] = {0};
    uint64_t prof_n[K_COUNT] = {0};
};

namespace {

zg_ctx *g_default = nullptr;
std::mutex g_default_mu;

int fail_hip(zg_ctx *c, hipError_t e, const char *what) {
    c->err = std::string(what) + ": " + hipGetErrorString(e);
    return ZG_ERR_HIP;
}
#define HIPCHK(call)                                          \
    do {                                                      \
        hipError_t e_ = (call);                               \
        if (e_ != hipSuccess) return fail_hip(c, e_, #call);  \
    } while (0)

struct ProfScope {
    zg_ctx *c;
    int id;
    hipStream_t s;  // the stream the bracketed launches go to
    hipEvent_t a = nullptr, b = nullptr;
    ProfScope(zg_ctx *c_, int id_, hipStream_t s_ = nullptr) : c(c_), id(id_), s(s_ ? s_ : c_->stream) {
        if (c->prof && (c->prof_only < 0 || c->prof_only == id_) && hipEventCreate(&a) == hipSuccess &&
            hipEventCreate(&b) == hipSuccess)
            (void)hipEventRecord(a, s);
    }
    ~ProfScope() {
        if (a && b) {
            (void)hipEventRecord(b, s);
            c->events.push_back({id, a, b});
        }
    }
};
// ZLIB_AMD_TRACE=1 names every step on stderr and waits for it: tells which kernel a fault or hang belongs to
static const bool g_trace = [] {
    const char *e = std::getenv("ZLIB_AMD_TRACE");
    return e && e[0] == '1';
}();
#define LAUNCH(id, call)                                              \
    do {                                                              \
        if (g_trace) fprintf(stderr, "[zlib_amd] %s: %s\n", kKernelNames[id], #call); \
        {                                                             \
            ProfScope ps_(c, id);                                     \
            HIPCHK(call);                                             \
        }                                                             \
        if (g_trace) {                                                \
            HIPCHK(hipStreamSynchronize(c->stream));                  \
            fprintf(stderr, "[zlib_amd]   done\n");                   \
        }                                                             \
    } while (0)

void prof_collect(zg_ctx *c) {
    for (auto &e : c->events) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
            c->prof_ms[e.id] += ms;
            c->prof_n[e.id] += 1;
        }
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    c->events.clear();
}

// Largest chunk / record slot: bit offsets of the encode tiles inside a block are 32-bit (at most 15 bits per
// input byte) and k_adler sums i * x[i] over one slot in 64 bits.
constexpr size_t kMaxChunkBytes = (size_t)256 << 20;

struct Plan {
    size_t chunk_bytes;
    size_t batch_chunks;
};
int make_plan(const zg_opts *opts, Plan *p) {
    size_t cb = (opts && opts->chunk_bytes) ? opts->chunk_bytes : ((size_t)1 << 20);
    size_t bb = (opts && opts->batch_bytes) ? opts->batch_bytes : ((size_t)256 << 20);
    if (cb < 1024 || (cb % 1024) != 0) return ZG_ERR_ARG;
    if (cb > kMaxChunkBytes) return ZG_ERR_UNSUPPORTED;
    size_t bc = bb / cb;
    if (bc < 1) bc = 1;
    if (bc > 32768) bc = 32768;
    p->chunk_bytes = cb;
    p->batch_chunks = bc;
    return ZG_OK;
}

uint32_t max_tiles_for(size_t chunk_bytes) { return (uint32_t)((chunk_bytes + 1 + kEncTile - 1) / kEncTile); }

// Strategy::Auto (reference src/compress.rs:89-169): decided on the host from the first KiB.  The entropy
// sum runs in ascending symbol order (the reference follows HashMap order) and heuristic 2 sums the five
// smallest counts, as the reference code does.
int auto_choice(const uint8_t *head, size_t n) {
    if (n < 256) return ZG_RAW;
    if (n < 1024) return ZG_FIXED;
    size_t freq[256] = {0};
    for (size_t i = 0; i < 1024; i++) freq[head[i]]++;
    double ent = 0.0, preference = 0.0;
    size_t distinct = 0;
    std::vector<uint32_t> counts;
    for (int s = 0; s < 256; s++)
        if (freq[s]) {
            const double p = (double)freq[s] / 1024.0;
            ent += p * std::log2(p);
            distinct++;
            counts.push_back((uint32_t)freq[s]);
        }
    const double proximity = 1.0 - std::fabs((-ent / std::log2((double)distinct)) - 1.0);
    preference += proximity > 0.75 ? proximity : 0.5 * proximity;
    std::sort(counts.begin(), counts.end());
    uint32_t sum = 0;
    for (size_t i = 0; i < std::min<size_t>(5, counts.size()); i++) sum += counts[i];
    const double proportion = (double)sum / 1024.0;
    preference -= proportion > 0.5 ? proportion : 0.5 * proportion;
    if (preference > 0.1) return ZG_FIXED;
    if (preference < -0.1) return ZG_DYNAMIC;
    const double ratio = (double)distinct * std::log2((double)n) / (double)n;
    return ratio < 0.5 ? ZG_DYNAMIC : ZG_FIXED;
}

int ensure_chunk_meta(zg_ctx *c, size_t nc, uint32_t max_tiles) {
    HIPCHK(c->ntok.ensure(nc * sizeof(uint32_t)));
    HIPCHK(c->freq.ensure(nc * kTabStride * sizeof(uint32_t)));
    HIPCHK(c->tab.ensure(nc * kTabStride * sizeof(uint32_t)));
    HIPCHK(c->hdr.ensure(nc * kHdrWords * sizeof(uint32_t)));
    HIPCHK(c->hdr_bits.ensure(nc * sizeof(uint32_t)));
    HIPCHK(c->block_bits.ensure(nc * sizeof(uint64_t)));
    HIPCHK(c->bit_off.ensure(nc * sizeof(uint64_t)));
    HIPCHK(c->tile_bits.ensure(nc * (size_t)max_tiles * sizeof(uint32_t)));
    return ZG_OK;
}

// index + match + parse + code construction for chunks [c0, c1) of the grid
int run_front(zg_ctx *c, const ChunkGrid &g, uint32_t c0, uint32_t c1, MatchParams mp, int strategy,
              int64_t final_chunk, bool want_codes) {
    const uint64_t bs = g.first + (uint64_t)c0 * g.chunk_bytes;
    const uint64_t be = std::min<uint64_t>(g.end, g.first + (uint64_t)c1 * g.chunk_bytes);
    // isolated chunks see nothing before their own first byte; others need the index a window behind them
    const uint64_t back = g.isolated ? 0 : index_lookback(mp.window);
    const uint64_t ib0 = (bs > back ? bs - back : 0) & ~(uint64_t)15;  // scratch arrays start here
    const uint64_t ix0 = g.isolated ? bs : ib0;                        // first indexed position
    const uint64_t span = be - ib0;
    const uint64_t iso_first = g.isolated ? g.first : 0;
    const uint32_t iso_bytes = g.isolated ? (uint32_t)g.chunk_bytes : 0u;
    // Two ways to the match of every position the parse stands on.  The walk (k_walk): the search runs along the
    // parse only, on the four-byte chains, with the three-byte chains beside it for lengths of three.  The full
    // search (k_match, k_match_lo): every position, five-byte chains first, then four- and three-byte chains for
    // the positions left open; it also serves small records and stands in should the walk's check ever fail.
    const bool two_level = mp.min_len < 5;  // full search: lengths 3 and 4 need the shorter chains
    const bool can_walk = !c->walk_off && match_tile_positions(g) == kMatchTile;
    const bool walk_all = can_walk && c->walk_mode == 1;
    const bool walk_some = can_walk && c->walk_mode == 2 && two_level;  // chunk by chunk, after a look at the chains
    const bool walk = walk_all || walk_some, full = !walk_all;
    const bool need5 = full, need4 = walk || two_level, need3 = (walk && mp.min_len < 4) || (full && two_level);
    const uint32_t rec_min = mp.min_len < 5 ? 5u : mp.min_len;
    if (need5) {
        HIPCHK(c->far.ensure(span * 2 + 64));
        HIPCHK(c->next.ensure(span * 2 + 64));
    }
    if (need4) {
        HIPCHK(c->far4.ensure(span * 2 + 64));
        HIPCHK(c->next4.ensure(span * 2 + 64));
    }
    if (need3) {
        HIPCHK(c->far3.ensure(span * 2 + 64));
        HIPCHK(c->next3.ensure(span * 2 + 64));
    }
    HIPCHK(c->match.ensure(span * 4 + 64));
    HIPCHK(c->tokens.ensure((be - bs) * 4 + 64));
    if (span) {
        uint32_t *status = &c->scalars.as<Scalars>()->status;
        IndexSet set;
        if (need5) set.add(5, c->far.as<uint16_t>(), c->next.as<uint16_t>());
        if (need4) set.add(4, c->far4.as<uint16_t>(), c->next4.as<uint16_t>());
        if (need3) set.add(3, c->far3.as<uint16_t>(), c->next3.as<uint16_t>());
        if (two_level && full) {
            const uint64_t ntiles = (uint64_t)(c1 - c0) * match_tiles_per_chunk(g);
            HIPCHK(c->open_list.ensure(ntiles * match_tile_positions(g) * sizeof(uint32_t)));
            HIPCHK(c->open_far.ensure(ntiles * match_tile_positions(g) * sizeof(uint32_t)));
            HIPCHK(c->open_cnt.ensure(ntiles * sizeof(uint32_t)));
        }
        LAUNCH(K_INDEX, launch_index(g.data, g.end, ix0, be, bs, mp.window, set, ib0, status, c->index_safe, iso_first,
                                     iso_bytes, c->stream));
        const uint8_t *walk_chunk = nullptr;  // null: every chunk goes the one way
        if (walk_some) {
            HIPCHK(c->walk_chunk.ensure(c1 - c0));
            LAUNCH(K_MISC, launch_chunk_probe(g, c0, c1, c->far.as<uint16_t>(), c->next.as<uint16_t>(), ib0,
                                              c->walk_chunk.as<uint8_t>(), c->stream));
            walk_chunk = c->walk_chunk.as<uint8_t>();
        }
        if (full) {
            if (two_level) {
                LAUNCH(K_MATCH, launch_match(g, c0, c1, mp, rec_min, c->far.as<uint16_t>(), c->next.as<uint16_t>(), ib0,
                                             c->match.as<uint32_t>(), c->open_list.as<uint32_t>(),
                                             c->open_cnt.as<uint32_t>(), walk_chunk, c->stream));
                LAUNCH(K_MATCH_LO, launch_match_lo(g, c0, c1, mp, c->far4.as<uint16_t>(), c->next4.as<uint16_t>(),
                                                   c->far3.as<uint16_t>(), c->next3.as<uint16_t>(), ib0,
                                                   c->match.as<uint32_t>(), c->open_list.as<uint32_t>(),
                                                   c->open_far.as<uint32_t>(), c->open_cnt.as<uint32_t>(), c->stream));
            } else {
                LAUNCH(K_MATCH, launch_match(g, c0, c1, mp, rec_min, c->far.as<uint16_t>(), c->next.as<uint16_t>(), ib0,
                                             c->match.as<uint32_t>(), nullptr, nullptr, walk_chunk, c->stream));
            }
        }
        if (walk) {
            const uint32_t tpr = walk_tiles_per_run(g, c1 - c0, (uint32_t)c->cus);
            const uint64_t runs = (uint64_t)(c1 - c0) * ((match_tiles_per_chunk(g) + tpr - 1) / tpr);
            HIPCHK(c->walk_exit.ensure(runs * kWalkExitWords * sizeof(uint32_t)));
            const uint16_t *f3 = need3 ? c->far3.as<uint16_t>() : c->far4.as<uint16_t>();  // unread without need3
            const uint16_t *n3 = need3 ? c->next3.as<uint16_t>() : c->next4.as<uint16_t>();
            LAUNCH(K_WALK, launch_walk(g, c0, c1, mp, c->far4.as<uint16_t>(), c->next4.as<uint16_t>(), f3, n3, ib0,
                                       c->match.as<uint32_t>(), c->walk_exit.as<uint32_t>(), tpr, walk_chunk,
                                       c->stream));
        }
    }
    {
        const uint64_t nc = c1 - c0;
        const uint64_t spc = parse_subs_per_chunk(g.chunk_bytes), tpc = emit_tiles_per_chunk(g.chunk_bytes);
        HIPCHK(c->flag.ensure(parse_flag_bytes(nc, g.chunk_bytes) + 64));
        HIPCHK(c->sub_exit.ensure(nc * spc * 4));
        HIPCHK(c->sub_exit2.ensure(nc * spc * 4));
        HIPCHK(c->merged.ensure(nc * spc));
        HIPCHK(c->tile_cnt.ensure(nc * tpc * 4));
        LAUNCH(K_PARSE, launch_parse(g, c0, c1, c->match.as<uint32_t>(), ib0, c->flag.as<uint32_t>(),
                                     c->sub_exit.as<uint32_t>(), c->sub_exit2.as<uint32_t>(), c->merged.as<uint8_t>(),
                                     c->tile_cnt.as<uint32_t>(), c->tokens.as<uint32_t>(), bs, c->ntok.as<uint32_t>(),
                                     c->freq.as<uint32_t>(), &c->scalars.as<Scalars>()->status, c->stream));
    }
    if (want_codes)
        LAUNCH(K_HUFFMAN, launch_huffman(c1 - c0, strategy, c->freq.as<uint32_t>(), final_chunk, c->tab.as<uint32_t>(),
                                         c->hdr.as<uint32_t>(), c->hdr_bits.as<uint32_t>(),
                                         c->block_bits.as<uint64_t>(), c->stream));
    return ZG_OK;
}

// bit offsets + packing for chunks [c0, c1); the carry scalar holds the first block's bit offset
int run_back(zg_ctx *c, const ChunkGrid &g, uint32_t c0, uint32_t c1, uint32_t max_tiles, uint32_t *out,
             uint64_t out_cap_bits) {
    Scalars *sc = c->scalars.as<Scalars>();
    const uint64_t bs = g.first + (uint64_t)c0 * g.chunk_bytes;
    LAUNCH(K_TILEBITS, launch_tile_bits(c1 - c0, max_tiles, c->tokens.as<uint32_t>(), g, c0, bs,
                                        c->ntok.as<uint32_t>(), c->tab.as<uint32_t>(), c->tile_bits.as<uint32_t>(),
                                        c->stream));
    LAUNCH(K_OFFSETS, launch_offsets(c1 - c0, max_tiles, c->ntok.as<uint32_t>(), c->tile_bits.as<uint32_t>(),
                                     c->block_bits.as<uint64_t>(), c->bit_off.as<uint64_t>(), &sc->carry, c->stream));
    // the words k_encode ORs into; the rest of the output needs no clearing
    LAUNCH(K_OFFSETS, launch_zero_seams(c1 - c0, max_tiles, c->ntok.as<uint32_t>(), c->hdr_bits.as<uint32_t>(),
                                        c->tile_bits.as<uint32_t>(), c->bit_off.as<uint64_t>(),
                                        c->block_bits.as<uint64_t>(), out, out_cap_bits, false, c->stream));
    LAUNCH(K_ENCODE, launch_encode(c1 - c0, max_tiles, c->tokens.as<uint32_t>(), g, c0, bs, c->ntok.as<uint32_t>(),
                                   c->tab.as<uint32_t>(), c->hdr.as<uint32_t>(), c->hdr_bits.as<uint32_t>(),
                                   c->tile_bits.as<uint32_t>(), c->bit_off.as<uint64_t>(), out, out_cap_bits,
                                   &sc->status, c->stream));
    return ZG_OK;
}

// keeps the bit offsets of blocks [c0, c1) of the current call: batch by batch into a device array of one entry
// per block (a copy between device buffers, which does not hold the host up), fetched in one piece at the end of the
// call by fetch_block_offsets -- a copy to pageable host memory after every batch made the host wait for that
// batch's kernels before it could queue the next batch's upload
int keep_block_offsets(zg_ctx *c, uint32_t c0, uint32_t c1) {
    if (c->block_offs.size() < c1) return ZG_OK;
    HIPCHK(hipMemcpyAsync(c->block_off_all.as<uint64_t>() + c0, c->bit_off.p, (size_t)(c1 - c0) * sizeof(uint64_t),
                          hipMemcpyDeviceToDevice, c->stream));
    return ZG_OK;
}
// queued before the call's last synchronisation
int fetch_block_offsets(zg_ctx *c) {
    if (c->block_offs.empty()) return ZG_OK;
    HIPCHK(hipMemcpyAsync(c->block_offs.data(), c->block_off_all.p, c->block_offs.size() * sizeof(uint64_t),
                          hipMemcpyDeviceToHost, c->stream));
    return ZG_OK;
}

// Adler-32 of data[0..n) folded into the running state
int run_adler(zg_ctx *c, const uint8_t *data, uint64_t n) {
    if (!n) return ZG_OK;
    const uint64_t seg = 256 * 1024;
    Scalars *sc = c->scalars.as<Scalars>();
    const uint64_t nseg = (n + seg - 1) / seg;
    HIPCHK(c->adler_part.ensure(nseg * 2 * sizeof(uint32_t)));
    HIPCHK(c->adler_blk.ensure((nseg / 1024 + 1) * 3 * sizeof(uint64_t)));
    LAUNCH(K_ADLER, launch_adler(data, n, seg, (uint32_t)nseg, nullptr, c->adler_part.as<uint32_t>(), c->stream));
    LAUNCH(K_ADLER, launch_adler_fold(c->adler_part.as<uint32_t>(), n, seg, (uint32_t)nseg, c->adler_blk.as<uint64_t>(),
                                      sc->adler, c->stream));
    return ZG_OK;
}

int init_scalars(zg_ctx *c, uint64_t carry) {
    HIPCHK(c->scalars.ensure(sizeof(Scalars)));
    Scalars h{};
    h.carry = carry;
    h.adler[0] = 1;
    h.adler[1] = 0;
    HIPCHK(hipMemcpyAsync(c->scalars.p, &h, sizeof h, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));  // h is a stack object
    return ZG_OK;
}

size_t stored_size(size_t n) {
    const size_t nb = n == 0 ? 1 : (n + kStoredBlock - 1) / kStoredBlock;
    return 2 + n + 5 * nb + 4;
}

constexpr size_t kTouchMin = (size_t)64 << 20;  // zg_compress: inputs from this size on get their result pages touched ahead
constexpr int kRetrySafe = 1;    // internal: the index kernel asked for the conservative variant
constexpr int kRetryNoWalk = 2;  // internal: the parse left the positions the walk kernel had searched

// h_in != nullptr: the input is still in host memory; it is copied to d_in batch by batch on
// the context's copy stream, each batch's kernels waiting only for their own bytes, so that the upload of
// batch k + 1 runs beside the kernels of batch k.
int compress_device_once(zg_ctx *c, const uint8_t *d_in, size_t n, int strategy, const zg_opts *opts,
                         uint8_t *d_out, size_t out_cap, size_t *out_n, const uint8_t *h_in = nullptr) {
    Plan plan;
    int rc = make_plan(opts, &plan);
    if (rc) return rc;
    if (strategy < ZG_AUTO || strategy > ZG_RAW) return ZG_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(d_out) & 3) != 0) return ZG_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    if (strategy == ZG_AUTO) {
        uint8_t head[1024];
        const size_t hn = std::min<size_t>(n, 1024);
        if (hn && h_in) memcpy(head, h_in, hn);
        else if (hn) HIPCHK(hipMemcpy(head, d_in, hn, hipMemcpyDeviceToHost));
        strategy = auto_choice(head, n);
    }
    if (out_cap < zg_compress_bound(n, strategy, opts)) return ZG_ERR_CAPACITY;
    // uploads: [0, up_done) is on its way; `upload(upto)` queues the rest up to `upto` and makes the compute
    // stream wait for it
    size_t up_done = 0;
    auto upload = [&](size_t upto) -> int {
        if (!h_in) return ZG_OK;
        upto = std::min(upto, n);
        if (upto > up_done) {
            HIPCHK(hipMemcpyAsync(const_cast<uint8_t *>(d_in) + up_done, h_in + up_done, upto - up_done,
                                  hipMemcpyHostToDevice, c->copy));
            up_done = upto;
        }
        hipEvent_t ev;
        HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        hipError_t e = hipEventRecord(ev, c->copy);
        if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, ev, 0);
        (void)hipEventDestroy(ev);  // released once it has completed
        HIPCHK(e);
        return ZG_OK;
    };
    rc = init_scalars(c, 16);
    if (rc) return rc;
    Scalars *sc = c->scalars.as<Scalars>();

    c->block_offs.clear();
    if (strategy == ZG_RAW) {
        rc = upload(n);
        if (rc) return rc;
        const uint64_t nblk = n == 0 ? 1 : (n + kStoredBlock - 1) / kStoredBlock;
        HIPCHK(c->adler_part.ensure(nblk * 2 * sizeof(uint32_t)));
        HIPCHK(c->adler_blk.ensure((nblk / 1024 + 1) * 3 * sizeof(uint64_t)));
        LAUNCH(K_STORED, launch_stored(d_in, n, d_out, c->adler_part.as<uint32_t>(), c->stream));
        if (n)  // the stored-block kernel left one Adler part per 16 KiB block
            LAUNCH(K_ADLER, launch_adler_fold(c->adler_part.as<uint32_t>(), n, kStoredBlock, (uint32_t)nblk,
                                              c->adler_blk.as<uint64_t>(), sc->adler, c->stream));
        const uint64_t bits = (uint64_t)(stored_size(n) - 4) * 8;
        HIPCHK(hipMemcpyAsync(&sc->carry, &bits, sizeof bits, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        LAUNCH(K_MISC, launch_trailer(d_out, &sc->carry, sc->adler, 0, 0, 1, 1, &sc->out_bytes, c->stream));
    } else {
        ChunkGrid g;
        g.data = d_in;
        g.first = 0;
        g.end = n;
        g.chunk_bytes = plan.chunk_bytes;
        g.n_chunks = n == 0 ? 1u : (uint32_t)((n + plan.chunk_bytes - 1) / plan.chunk_bytes);
        const uint32_t max_tiles = max_tiles_for(plan.chunk_bytes);
        const size_t bc = std::min<size_t>(plan.batch_chunks, g.n_chunks);
        rc = ensure_chunk_meta(c, bc, max_tiles);
        if (rc) return rc;
        const size_t bound = zg_compress_bound(n, strategy, opts);
        try {
            c->block_offs.assign(g.n_chunks, 0);
        } catch (...) {
            return ZG_ERR_NOMEM;
        }
        HIPCHK(c->block_off_all.ensure((size_t)g.n_chunks * sizeof(uint64_t)));
        // the word the first block begins in; k_zero_seams clears the other words that are ORed into
        LAUNCH(K_MISC, hipMemsetAsync(d_out, 0, std::min<size_t>(bound, 8), c->stream));
        const MatchParams mp{kMaxWindow, 3, 258};  // get_zlib_compressor, src/compress.rs:451-456
        rc = upload((size_t)std::min<uint64_t>(g.n_chunks, bc) * plan.chunk_bytes);
        if (rc) return rc;
        for (uint32_t c0 = 0; c0 < g.n_chunks; c0 += (uint32_t)bc) {
            const uint32_t c1 = (uint32_t)std::min<uint64_t>(g.n_chunks, (uint64_t)c0 + bc);
            const int64_t final_chunk = (c1 == g.n_chunks) ? (int64_t)(c1 - c0 - 1) : -1;
            rc = run_front(c, g, c0, c1, mp, strategy, final_chunk, true);
            if (rc) return rc;
            rc = run_back(c, g, c0, c1, max_tiles, reinterpret_cast<uint32_t *>(d_out), (uint64_t)(bound - 4) * 8);
            if (rc) return rc;
            rc = keep_block_offsets(c, c0, c1);
            if (rc) return rc;
            // the next batch's bytes travel while this batch's kernels run
            rc = upload(((size_t)c1 + bc) * plan.chunk_bytes);
            if (rc) return rc;
        }
        rc = run_adler(c, d_in, n);
        if (rc) return rc;
        LAUNCH(K_MISC, launch_trailer(d_out, &sc->carry, sc->adler, 0, 0, 1, 1, &sc->out_bytes, c->stream));
        rc = fetch_block_offsets(c);
        if (rc) return rc;
    }
    Scalars h;
    HIPCHK(hipMemcpyAsync(&h, c->scalars.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->prof) prof_collect(c);
    if (h.status & 2u) return kRetrySafe;
    if (h.status & 4u) return kRetryNoWalk;
    if (h.status & 1u) return ZG_ERR_CAPACITY;
    *out_n = (size_t)h.out_bytes;
    return ZG_OK;
}

int compress_device_locked(zg_ctx *c, const uint8_t *d_in, size_t n, int strategy, const zg_opts *opts,
                           uint8_t *d_out, size_t out_cap, size_t *out_n, const uint8_t *h_in = nullptr) {
    int rc = compress_device_once(c, d_in, n, strategy, opts, d_out, out_cap, out_n, h_in);
    for (int attempt = 0; attempt < 2 && (rc == kRetrySafe || rc == kRetryNoWalk); attempt++) {
        if (rc == kRetrySafe && c->index_safe) break;
        if (rc == kRetryNoWalk && c->walk_off) break;
        (rc == kRetrySafe ? c->index_safe : c->walk_off) = true;
        rc = compress_device_once(c, d_in, n, strategy, opts, d_out, out_cap, out_n);  // the input is on the device by now
    }
    return (rc == kRetrySafe || rc == kRetryNoWalk) ? ZG_ERR_INTERNAL : rc;
}

zg_ctx *resolve(zg_ctx *ctx, int *rc) {
    if (ctx) return ctx;
    std::lock_guard<std::mutex> lk(g_default_mu);
    if (!g_default) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        *rc = zg_init(dev, &g_default);
        if (*rc) g_default = nullptr;
    }
    return g_default;
}

}  // namespace

extern "C" {

int zg_init(int device, zg_ctx **ctx) {
    if (!ctx) return ZG_ERR_ARG;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count)
        return ZG_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return ZG_ERR_NO_DEVICE;
    zg_ctx *c = new (std::nothrow) zg_ctx();
    if (!c) return ZG_ERR_NOMEM;
    c->device = device;
    if (const char *e = std::getenv("ZLIB_AMD_SAFE_INDEX")) c->index_safe = e[0] == '1';  // test hook
    if (const char *e = std::getenv("ZLIB_AMD_WALK")) c->walk_mode = e[0] == '1' ? 1 : (e[0] == '0' ? 0 : 2);
    if (hipDeviceGetAttribute(&c->cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || c->cus <= 0)
        c->cus = 256;
    // this device's copy of the fixed scratch array (the current device is `device`)
    if (hipGetSymbolAddress(reinterpret_cast<void **>(&c->noalloc_scratch), HIP_SYMBOL(g_noalloc_scratch)) !=
            hipSuccess ||
        !c->noalloc_scratch) {
        delete c;
        return ZG_ERR_NO_DEVICE;
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking) != hipSuccess) {
        if (c->stream) (void)hipStreamDestroy(c->stream);
        delete c;
        return ZG_ERR_NO_DEVICE;
    }
    *ctx = c;
    return ZG_OK;
}

void zg_destroy(zg_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    prof_collect(c);
    DevBuf *bufs[] = {&c->inf_tok, &c->inf_ntok, &c->inf_units, &c->inf_win, &c->block_off_all, &c->walk_exit, &c->walk_chunk, &c->shard_buf, &c->far, &c->next, &c->far4, &c->next4, &c->far3, &c->next3, &c->open_list, &c->open_far, &c->open_cnt, &c->rec_len, &c->rec_off, &c->inf_off, &c->inf_len, &c->inf_status, &c->match, &c->tokens, &c->ntok, &c->freq, &c->tab, &c->hdr, &c->hdr_bits,
                      &c->block_bits, &c->bit_off, &c->tile_bits, &c->adler_part, &c->scalars, &c->stage_in,
                      &c->stage_out, &c->misc, &c->flag, &c->sub_exit, &c->sub_exit2, &c->merged, &c->tile_cnt,
                      &c->adler_blk};
    for (DevBuf *b : bufs) b->release();
    for (hipEvent_t e : c->run.ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->inf_ev)
        if (e) (void)hipEventDestroy(e);
    if (c->run.h_carry) (void)hipHostFree(c->run.h_carry);
    (void)hipStreamDestroy(c->stream);
    (void)hipStreamDestroy(c->copy);
    delete c;
}

const char *zg_strerror(int code) {
    switch (code) {
    case ZG_OK: return "ok";
    case ZG_ERR_ARG: return "invalid argument";
    case ZG_ERR_NO_DEVICE: return "no usable HIP device (this library has no CPU path)";
    case ZG_ERR_HIP: return "HIP runtime call failed";
    case ZG_ERR_NOMEM: return "out of memory";
    case ZG_ERR_CAPACITY: return "output buffer too small";
    case ZG_ERR_UNSUPPORTED: return "parameters not supported by the device path";
    case ZG_ERR_INTERNAL: return "internal consistency check failed";
    case ZG_ERR_DATA: return "invalid or corrupt zlib stream";
    case ZG_ERR_IO: return "a read or write of the stream front end failed";
    default: return "unknown error";
    }
}

const char *zg_last_error(zg_ctx *ctx) {
    int rc = 0;
    zg_ctx *c = resolve(ctx, &rc);
    return c ? c->err.c_str() : "no context";
}

void zg_free(void *p) { free(p); }

size_t zg_compress_bound(size_t n, int strategy, const zg_opts *opts) {
    Plan plan;
    if (make_plan(opts, &plan)) return 0;
    if (strategy == ZG_RAW) return (stored_size(n) + 3) & ~(size_t)3;
    // worst case per unit: a 15-bit literal code; per block: header of at most kHdrWords words + EOB.
    // Auto may resolve to Raw, so take the larger of the two.
    const size_t nchunks = n == 0 ? 1 : (n + plan.chunk_bytes - 1) / plan.chunk_bytes;
    size_t bits_bytes = 2 + (n * 15 + 7) / 8 + nchunks * (kHdrWords * 4 + 2) + 4 + 8;
    if (strategy == ZG_AUTO) bits_bytes = std::max(bits_bytes, stored_size(n));
    return (bits_bytes + 3) & ~(size_t)3;
}

int zg_compress_device(zg_ctx *ctx, const void *d_in, size_t n, int strategy, const zg_opts *opts, void *d_out,
                       size_t out_cap, size_t *out_n) {
    int rc = 0;
    zg_ctx *c = resolve(ctx, &rc);
    if (!c) return rc ? rc : ZG_ERR_NO_DEVICE;
    if (!d_out || !out_n || (!d_in && n)) return ZG_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    return compress_device_locked(c, static_cast<const uint8_t *>(d_in), n, strategy, opts,
                                  static_cast<uint8_t *>(d_out), out_cap, out_n);
}

int zg_compress(zg_ctx *ctx, const uint8_t *in, size_t n, int strategy, const zg_opts *opts, uint8_t **out,
                size_t *out_n) {
    int rc = 0;
    zg_ctx *c = resolve(ctx, &rc);
    if (!c) return rc ? rc : ZG_ERR_NO_DEVICE;
    if (!out || !out_n || (!in && n)) return ZG_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    if (strategy == ZG_AUTO) strategy = auto_choice(in, n);
    const size_t bound = zg_compress_bound(n, strategy, opts);
    if (!bound) return ZG_ERR_ARG;
    HIPCHK(c->stage_in.ensure(n + 64));
    HIPCHK(c->stage_out.ensure(bound));
    // The input goes up batch by batch beside the kernels (compress_device_once); the stream comes back in one
    // copy.
    // The result buffer is reserved at its bound before the kernels start (untouched pages cost nothing) and a
    // helper thread touches its first pages while the GPU works, so that the download does not pay for the
    // page faults of fresh memory; the buffer is cut to the stream's size at the end.
    uint8_t *h = static_cast<uint8_t *>(malloc(bound));  // nullptr: the bound is more than the system grants; allocate afterwards
    std::atomic<bool> stop{false};
    std::thread toucher;
    if (h && n >= kTouchMin) {
        try {
            toucher = std::thread([h, n, bound, &stop] {
                const size_t upto = std::min(n, bound);
                for (size_t off = 0; off < upto && !stop.load(std::memory_order_relaxed); off += 4096)
                    *reinterpret_cast<volatile uint8_t *>(h + off) = 0;
            });
        } catch (...) {  // no thread to be had: the download touches the pages itself
        }
    }
    size_t got = 0;
    rc = compress_device_locked(c, c->stage_in.as<uint8_t>(), n, strategy, opts, c->stage_out.as<uint8_t>(), bound,
                                &got, n ? in : nullptr);
    stop.store(true, std::memory_order_relaxed);
    if (toucher.joinable()) toucher.join();
    if (rc) (void)hipStreamSynchronize(c->copy);  // an upload may still be reading the caller's buffer
    if (!rc && !h) {
        h = static_cast<uint8_t *>(malloc(got ? got : 1));
        if (!h) return ZG_ERR_NOMEM;
    }
    hipError_t ce = rc ? hipSuccess : hipMemcpy(h, c->stage_out.p, got, hipMemcpyDeviceToHost);
    if (rc || ce != hipSuccess) {
        free(h);
        if (rc) return rc;
        HIPCHK(ce);
    }
    if (uint8_t *cut = static_cast<uint8_t *>(realloc(h, got ? got : 1))) h = cut;
    *out = h;
    *out_n = got;
    return ZG_OK;
}

// ---------------------------------------------------------------------------
// Record batches: every record is compressed as a zlib stream of its own, i.e. compress(record, strategy)
// of the reference for each of them (src/compress.rs:51-86), all records in flight together.  A record is
// one isolated chunk; records sit in slots of `stride` bytes.
// ---------------------------------------------------------------------------
size_t zg_compress_batch_bound(size_t stride, size_t n_records, int strategy) {
    (void)strategy;
    // per record: header + trailer, block header, a 15-bit code per byte, end of block, padding
    const size_t per = 6 + kHdrWords * 4 + (stride * 15 + 7) / 8 + 2 + 1;
    return ((per * n_records + 8) + 3) & ~(size_t)3;
}

namespace {
int compress_batch_once(zg_ctx *c, const uint8_t *d_in, size_t stride, const uint32_t *d_len, size_t n_records,
                        int strategy, uint8_t *d_out, size_t out_cap, uint64_t *d_off, size_t *out_bytes) {
    if (strategy != ZG_DYNAMIC && strategy != ZG_FIXED) return ZG_ERR_UNSUPPORTED;
    if (stride < 1024 || (stride % 1024) != 0) return ZG_ERR_ARG;
    if (stride > kMaxChunkBytes) return ZG_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(d_out) & 3) != 0) return ZG_ERR_ARG;
    if (n_records > 0xFFFFFFFFull) return ZG_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    if (out_cap < zg_compress_batch_bound(stride, n_records, strategy)) return ZG_ERR_CAPACITY;
    c->block_offs.clear();  // zg_last_block_offsets speaks of whole-stream calls only
    int rc = init_scalars(c, 0);  // carry = byte offset of the next stream
    if (rc) return rc;
    Scalars *sc = c->scalars.as<Scalars>();
    if (n_records == 0) {
        *out_bytes = 0;
        return ZG_OK;
    }
    ChunkGrid g;
    g.data = d_in;
    g.first = 0;
    g.end = (uint64_t)stride * n_records;
    g.chunk_bytes = stride;
    g.n_chunks = (uint32_t)n_records;
    g.chunk_len = d_len;
    g.isolated = 1;
    const uint32_t max_tiles = max_tiles_for(stride);
    size_t bc = ((size_t)256 << 20) / stride;  // records per pass over the scratch buffers
    if (const char *e = getenv("ZLIB_AMD_BATCH_PASS_MIB"))  // measurement hook
        if (atoi(e) > 0) bc = ((size_t)atoi(e) << 20) / stride;
    if (bc < 1) bc = 1;
    // a record is a row of the kernels' grids: 65 535 at most -- which is also 32 whole fillings of the machine by
    // the match kernel's workgroups (passes of equal size, 61 681 records for a million, were slower than sixteen of
    // 65 535 and one of sixteen)
    if (bc > 65535) bc = 65535;
    if (bc > n_records) bc = n_records;
    rc = ensure_chunk_meta(c, bc, max_tiles);
    if (rc) return rc;
    HIPCHK(c->adler_part.ensure(bc * 2 * sizeof(uint32_t)));
    const MatchParams mp{kMaxWindow, 3, 258};  // get_zlib_compressor, src/compress.rs:451-456
    // The checksums need nothing but the records: they are taken on the second stream while the first computes the
    // streams' offsets and packs the bits (kernels of few workgroups among them).  Under the launch trace everything
    // stays on one stream.
    const bool side = !g_trace;
    if (side)
        for (int i = 0; i < 2; i++)
            if (!c->inf_ev[i]) HIPCHK(hipEventCreateWithFlags(&c->inf_ev[i], hipEventDisableTiming));
    hipStream_t s_adler = side ? c->copy : c->stream;
    uint32_t pass = 0;
    for (uint32_t c0 = 0; c0 < g.n_chunks; c0 += (uint32_t)bc, pass++) {
        const uint32_t c1 = (uint32_t)std::min<uint64_t>(g.n_chunks, (uint64_t)c0 + bc);
        rc = run_front(c, g, c0, c1, mp, strategy, kAllFinal, true);
        if (rc) return rc;
        const uint64_t bs = (uint64_t)c0 * stride;
        if (side) {
            // (in the first stream's order this is behind the frame kernel of the pass before, the last reader of
            // the checksums)
            HIPCHK(hipEventRecord(c->inf_ev[0], c->stream));
            HIPCHK(hipStreamWaitEvent(s_adler, c->inf_ev[0], 0));
            {
                ProfScope ps_(c, K_ADLER, s_adler);
                HIPCHK(launch_adler(d_in + bs, (uint64_t)(c1 - c0) * stride, stride, c1 - c0,
                                    d_len ? d_len + c0 : nullptr, c->adler_part.as<uint32_t>(), s_adler));
            }
            HIPCHK(hipEventRecord(c->inf_ev[1], s_adler));
        }
        LAUNCH(K_TILEBITS, launch_tile_bits(c1 - c0, max_tiles, c->tokens.as<uint32_t>(), g, c0, bs,
                                            c->ntok.as<uint32_t>(), c->tab.as<uint32_t>(), c->tile_bits.as<uint32_t>(),
                                            c->stream));
        LAUNCH(K_OFFSETS, launch_tile_scan(c1 - c0, max_tiles, c->ntok.as<uint32_t>(), c->tile_bits.as<uint32_t>(),
                                           c->stream));
        LAUNCH(K_OFFSETS, launch_batch_offsets(c1 - c0, c->block_bits.as<uint64_t>(), c->bit_off.as<uint64_t>(),
                                               (pass & 1u) ? &sc->carry2 : &sc->carry,
                                               (pass & 1u) ? &sc->carry : &sc->carry2, d_off + c0, c->stream));
        // the words k_encode ORs into; the rest of the output needs no clearing
        LAUNCH(K_OFFSETS, launch_zero_seams(c1 - c0, max_tiles, c->ntok.as<uint32_t>(), c->hdr_bits.as<uint32_t>(),
                                            c->tile_bits.as<uint32_t>(), c->bit_off.as<uint64_t>(),
                                            c->block_bits.as<uint64_t>(), reinterpret_cast<uint32_t *>(d_out),
                                            (uint64_t)out_cap * 8, true, c->stream));
        LAUNCH(K_ENCODE, launch_encode(c1 - c0, max_tiles, c->tokens.as<uint32_t>(), g, c0, bs, c->ntok.as<uint32_t>(),
                                       c->tab.as<uint32_t>(), c->hdr.as<uint32_t>(), c->hdr_bits.as<uint32_t>(),
                                       c->tile_bits.as<uint32_t>(), c->bit_off.as<uint64_t>(),
                                       reinterpret_cast<uint32_t *>(d_out), (uint64_t)out_cap * 8, &sc->status,
                                       c->stream));
        if (side)
            HIPCHK(hipStreamWaitEvent(c->stream, c->inf_ev[1], 0));
        else
            LAUNCH(K_ADLER, launch_adler(d_in + bs, (uint64_t)(c1 - c0) * stride, stride, c1 - c0,
                                         d_len ? d_len + c0 : nullptr, c->adler_part.as<uint32_t>(), c->stream));
        LAUNCH(K_MISC, launch_batch_frame(c1 - c0, c->block_bits.as<uint64_t>(), d_off + c0,
                                          c->adler_part.as<uint32_t>(), d_out, c->stream));
    }
    Scalars h;
    HIPCHK(hipMemcpyAsync(&h, c->scalars.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->prof) prof_collect(c);
    if (h.status & 2u) return kRetrySafe;  // the exchange-order check of the index kernel tripped
    if (h.status & 4u) return kRetryNoWalk;
    if (h.status & 1u) return ZG_ERR_CAPACITY;
    *out_bytes = (size_t)((pass & 1u) ? h.carry2 : h.carry);
    return ZG_OK;
}

int compress_batch_locked(zg_ctx *c, const uint8_t *d_in, size_t stride, const uint32_t *d_len, size_t n_records,
                          int strategy, uint8_t *d_out, size_t out_cap, uint64_t *d_off, size_t *out_bytes) {
    int rc = compress_batch_once(c, d_in, stride, d_len, n_records, strategy, d_out, out_cap, d_off, out_bytes);
    for (int attempt = 0; attempt < 2 && (rc == kRetrySafe || rc == kRetryNoWalk); attempt++) {
        // once more with the index kernel that relies on no LDS ordering / with every position searched
        if (rc == kRetrySafe && c->index_safe) break;
        if (rc == kRetryNoWalk && c->walk_off) break;
        (rc == kRetrySafe ? c->index_safe : c->walk_off) = true;
        rc = compress_batch_once(c, d_in, stride, d_len, n_records, strategy, d_out, out_cap, d_off, out_bytes);
    }
    if (rc) (void)hipStreamSynchronize(c->copy);  // a pass's checksums may still be under way
    return (rc == kRetrySafe || rc == kRetryNoWalk) ? ZG_ERR_INTERNAL : rc;
}
}  // namespace

int zg_compress_batch_device(zg_ctx *ctx, const void *d_in, size_t stride, const uint32_t *d_len, size_t n_records,
                             int strategy, void *d_out, size_t out_cap, uint64_t *d_out_offsets, size_t *out_bytes) {
    int rc = 0;
    zg_ctx *c = resolve(ctx, &rc);
    if (!c) return rc ? rc : ZG_ERR_NO_DEVICE;
    if (!d_out || !out_bytes || !d_out_offsets || (!d_in && n_records)) return ZG_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    return compress_batch_locked(c, static_cast<const uint8_t *>(d_in), stride, d_len, n_records, strategy,
                                 static_cast<uint8_t *>(d_out), out_cap, d_out_offsets, out_bytes);
}

int zg_compress_batch(zg_ctx *ctx, const uint8_t *in, const size_t *offsets, size_t n_records, int strategy,
                      uint8_t **out, size_t **out_offsets) {
    int rc = 0;
    zg_ctx *c = resolve(ctx, &rc);
    if (!c) return rc ? rc : ZG_ERR_NO_DEVICE;
    if (!out || !out_offsets || !offsets || (!in && n_records && offsets[n_records] != offsets[0])) return ZG_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    size_t max_len = 0;
    for (size_t i = 0; i < n_records; i++) {
        if (offsets[i + 1] < offsets[i]) return ZG_ERR_ARG;
        max_len = std::max(max_len, offsets[i + 1] - offsets[i]);
    }
    if (max_len > 0xFFFFFFFFull) return ZG_ERR_ARG;
    const size_t stride = std::max<size_t>(1024, (max_len + 1023) / 1024 * 1024);
    // slots of `stride` bytes: a batch of equally long records whose length is a multiple of 1024 is copied as is
    const bool packed = n_records && max_len == stride && offsets[n_records] - offsets[0] == stride * n_records;
    std::vector<uint32_t> lens(n_records);
    for (size_t i = 0; i < n_records; i++) lens[i] = (uint32_t)(offsets[i + 1] - offsets[i]);
    HIPCHK(c->stage_in.ensure(stride * n_records + 64));
    HIPCHK(c->rec_len.ensure(n_records * sizeof(uint32_t) + 4));
    HIPCHK(c->rec_off.ensure((n_records + 1) * sizeof(uint64_t)));
    if (packed) {
        HIPCHK(hipMemcpyAsync(c->stage_in.p, in + offsets[0], stride * n_records, hipMemcpyHostToDevice, c->stream));
    } else if (n_records) {
        std::vector<uint8_t> slots(stride * n_records, 0);
        for (size_t i = 0; i < n_records; i++)
            if (lens[i]) memcpy(slots.data() + i * stride, in + offsets[i], lens[i]);
        HIPCHK(hipMemcpy(c->stage_in.p, slots.data(), slots.size(), hipMemcpyHostToDevice));
    }
    if (n_records)
        HIPCHK(hipMemcpy(c->rec_len.p, lens.data(), n_records * sizeof(uint32_t), hipMemcpyHostToDevice));
    const size_t bound = zg_compress_batch_bound(stride, n_records, strategy);
    HIPCHK(c->stage_out.ensure(bound));
    size_t got = 0;
    rc = compress_batch_locked(c, c->stage_in.as<uint8_t>(), stride, c->rec_len.as<uint32_t>(), n_records, strategy,
                               c->stage_out.as<uint8_t>(), bound, c->rec_off.as<uint64_t>(), &got);
    if (rc) return rc;
    uint8_t *h = static_cast<uint8_t *>(malloc(got ? got : 1));
    size_t *ho = static_cast<size_t *>(malloc((n_records + 1) * sizeof(size_t)));
    std::vector<uint64_t> off64(n_records + 1, 0);
    if (!h || !ho) {
        free(h);
        free(ho);
        return ZG_ERR_NOMEM;
    }
    hipError_t ce = got ? hipMemcpy(h, c->stage_out.p, got, hipMemcpyDeviceToHost) : hipSuccess;
    if (ce == hipSuccess && n_records)
        ce = hipMemcpy(off64.data(), c->rec_off.p, (n_records + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost);
    if (ce != hipSuccess) {
        free(h);
        free(ho);
        HIPCHK(ce);
    }
    for (size_t i = 0; i <= n_records; i++) ho[i] = (size_t)off64[i];
    *out = h;
    *out_offsets = ho;
    return ZG_OK;
}

// ---------------------------------------------------------------------------
// zlib::decompress(&[u8]) -> Result<Vec<u8>, String> (reference src/decompress.rs:34-88) on the device, for
// one stream or many.  Err(String) becomes ZG_ERR_DATA with the reference's message in zg_last_error; so do
// the corrupt-body cases where the reference panics.
// ---------------------------------------------------------------------------
namespace {
const char *inflate_message(uint32_t st) {
    switch (st) {
    case INF_BAD_METHOD: return "CM is not a supported compression method";     // src/decompress.rs:44-46
    case INF_BAD_WINDOW: return "Invalid compression info, must be < 7";        // :51-55
    case INF_BAD_FCHECK: return "CMF + FLAGS checksum failed!";                 // :61-63
    case INF_PRESET_DICT: return "Preset dictionaries are not supported";       // :67-69
    case INF_BAD_BLOCK: return "Invalid block type";                            // :113
    case INF_BAD_CHECKSUM: return "Checksum is invalid";                        // :86
    case INF_BAD_CODES: return "invalid code lengths in a dynamic block";
    case INF_BAD_SYMBOL: return "bits that match no code (the reference panics: Invalid decoded value)";
    case INF_BAD_DISTANCE: return "distance reaches before the start of the output";
    case INF_TRUNCATED: return "input ends inside the stream";
    case INF_OVERFLOW: return "output slot too small";
    default: return "ok";
    }
}

// first failing stream -> ZG_ERR_DATA (or ZG_ERR_CAPACITY) with a message; st = host copy of the statuses
int inflate_verdict(zg_ctx *c, const std::vector<uint32_t> &st) {
    for (size_t i = 0; i < st.size(); i++)
        if (st[i] != INF_OK) {
            c->err = std::string(inflate_message(st[i])) + " (stream " + std::to_string(i) + ")";
            return st[i] == INF_OVERFLOW ? ZG_ERR_CAPACITY : ZG_ERR_DATA;
        }
    return ZG_OK;
}
}  // namespace

int zg_decompress_batch_device(zg_ctx *ctx, const void *d_in, const uint64_t *d_in_offsets, size_t n_streams,
                               void *d_out, const uint64_t *d_out_offsets, uint64_t *d_out_len, uint32_t *d_status) {
    int rc = 0;
    zg_ctx *c = resolve(ctx, &rc);
    if (!c) return rc ? rc : ZG_ERR_NO_DEVICE;
    if (!d_in_offsets || !d_out_len || !d_status || n_streams > 0xFFFFFFFFull) return ZG_ERR_ARG;
    if ((d_out && !d_out_offsets) || (!d_in && n_streams)) return ZG_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    // d_out == NULL: sizes only (no output is written, the Adler-32 is not checked)
    LAUNCH(K_INFLATE, launch_inflate(static_cast<const uint8_t *>(d_in), d_in_offsets, (uint32_t)n_streams,
                                     static_cast<uint8_t *>(d_out), d_out_offsets, d_out_len, d_status,
                                     d_out == nullptr, false, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->prof) prof_collect(c);
    return ZG_OK;
}

int zg_decompress_batch(zg_ctx *ctx, const uint8_t *in, const size_t *offsets, size_t n_streams, uint8_t **out,
                        size_t **out_offsets) {
    int rc = 0;
    zg_ctx *c = resolve(ctx, &rc);
    if (!c) return rc ? rc : ZG_ERR_NO_DEVICE;
    if (!out || !out_offsets || !offsets || n_streams > 0xFFFFFFFFull) return ZG_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    for (size_t i = 0; i < n_streams; i++)
        if (offsets[i + 1] < offsets[i]) return ZG_ERR_ARG;
    const size_t total_in = n_streams ? offsets[n_streams] - offsets[0] : 0;
    if (total_in && !in) return ZG_ERR_ARG;
    std::vector<uint64_t> off(n_streams + 1, 0), olen(n_streams, 0), ooff(n_streams + 1, 0);
    std::vector<uint32_t> st(n_streams, 0);
    for (size_t i = 0; i <= n_streams; i++) off[i] = offsets[i] - offsets[0];
    HIPCHK(c->stage_in.ensure(total_in + 64));
    HIPCHK(c->inf_off.ensure(2 * (n_streams + 1) * sizeof(uint64_t)));
    HIPCHK(c->inf_len.ensure((n_streams + 1) * sizeof(uint64_t)));
    HIPCHK(c->inf_status.ensure((n_streams + 1) * sizeof(uint32_t)));
    uint64_t *d_off = c->inf_off.as<uint64_t>(), *d_ooff = d_off + (n_streams + 1);
    if (total_in) HIPCHK(hipMemcpy(c->stage_in.p, in + offsets[0], total_in, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_off, off.data(), (n_streams + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
    size_t total_out = 0;
    if (n_streams) {
        // pass 1: sizes (the output length of a stream follows from its symbols alone)
        LAUNCH(K_INFLATE, launch_inflate(c->stage_in.as<uint8_t>(), d_off, (uint32_t)n_streams, nullptr, nullptr,
                                         c->inf_len.as<uint64_t>(), c->inf_status.as<uint32_t>(), true, false,
                                         c->stream));
        HIPCHK(hipMemcpyAsync(olen.data(), c->inf_len.p, n_streams * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(st.data(), c->inf_status.p, n_streams * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        rc = inflate_verdict(c, st);
        if (rc) return rc;
        for (size_t i = 0; i < n_streams; i++) ooff[i + 1] = ooff[i] + olen[i];
        total_out = (size_t)ooff[n_streams];
        HIPCHK(c->stage_out.ensure(total_out + 64));
        HIPCHK(hipMemcpy(d_ooff, ooff.data(), (n_streams + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
        // pass 2: bytes and checksums; a wave per stream when the streams are few and long
        const bool large = n_streams <= 4096 && total_out / n_streams >= (64u << 10);
        LAUNCH(K_INFLATE, launch_inflate(c->stage_in.as<uint8_t>(), d_off, (uint32_t)n_streams,
                                         c->stage_out.as<uint8_t>(), d_ooff, c->inf_len.as<uint64_t>(),
                                         c->inf_status.as<uint32_t>(), false, large, c->stream));
        HIPCHK(hipMemcpyAsync(st.data(), c->inf_status.p, n_streams * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (c->prof) prof_collect(c);
        rc = inflate_verdict(c, st);
        if (rc) return rc;
    }
    uint8_t *h = static_cast<uint8_t *>(malloc(total_out ? total_out : 1));
    size_t *ho = static_cast<size_t *>(malloc((n_streams + 1) * sizeof(size_t)));
    if (!h || !ho) {
        free(h);
        free(ho);
        return ZG_ERR_NOMEM;
    }
    if (total_out) {
        const hipError_t ce = hipMemcpy(h, c->stage_out.p, total_out, hipMemcpyDeviceToHost);
        if (ce != hipSuccess) {
            free(h);
            free(ho);
            HIPCHK(ce);
        }
    }
    for (size_t i = 0; i <= n_streams; i++) ho[i] = (size_t)ooff[i];
    *out = h;
    *out_offsets = ho;
    return ZG_OK;
}

// One stream of this library, every block by a wave of its own (zg_inflate.hip, k_inflate_blocks).
int zg_decompress_blocks_device(zg_ctx *ctx, const void *d_stream, size_t stream_bytes, const uint64_t *bit_offsets,
                                size_t n_blocks, size_t chunk_bytes, void *d_out, size_t n) {
    int rc = 0;
    zg_ctx *c = resolve(ctx, &rc);
    if (!c) return rc ? rc : ZG_ERR_NO_DEVICE;
    if (!d_stream || !bit_offsets || n_blocks < 1 || n_blocks > 0xFFFFFFF0ull || (!d_out && n) || stream_bytes < 6)
        return ZG_ERR_ARG;
    if (chunk_bytes < kMaxWindow) return ZG_ERR_UNSUPPORTED;  // a block's window must lie in the one block before it
    const size_t want_blocks = n == 0 ? 1 : (n + chunk_bytes - 1) / chunk_bytes;
    if (want_blocks != n_blocks || bit_offsets[0] != 16) return ZG_ERR_ARG;
    for (size_t i = 1; i < n_blocks; i++)
        if (bit_offsets[i] <= bit_offsets[i - 1] || bit_offsets[i] >= (uint64_t)(stream_bytes - 4) * 8) return ZG_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    const uint8_t *in = static_cast<const uint8_t *>(d_stream);
    uint8_t edge[6];  // header and trailer (src/decompress.rs:37-69, 74-87)
    HIPCHK(hipMemcpy(edge, in, 2, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(edge + 2, in + stream_bytes - 4, 4, hipMemcpyDeviceToHost));
    uint32_t hst = INF_OK;
    if ((edge[0] & 15u) != 8u) hst = INF_BAD_METHOD;
    else if ((edge[0] >> 4) > 7u) hst = INF_BAD_WINDOW;
    else if ((edge[0] * 256u + edge[1]) % 31u != 0u) hst = INF_BAD_FCHECK;
    else if ((edge[1] >> 5) & 1u) hst = INF_PRESET_DICT;
    if (hst != INF_OK) {
        c->err = inflate_message(hst);
        return ZG_ERR_DATA;
    }
    // tokens and units are kept for two passes of blocks at a time (4 GiB of output each: sixteen blocks of 1 MiB per
    // CU decoding while the pass before is expanded; 8 GiB of the mixed corpus took 1.49 s with passes of 2 GiB, 1.30
    // with 4, 1.45 with 4 and one pass at a time), the windows for all blocks
    uint64_t pass_bytes = (uint64_t)4 << 30;
    if (const char *e = getenv("ZLIB_AMD_INFLATE_PASS_MIB"))  // tests: several passes on a small stream
        if (atoi(e) > 0) pass_bytes = (uint64_t)atoi(e) << 20;
    const uint32_t pass_blocks = (uint32_t)std::min<uint64_t>(n_blocks, std::max<uint64_t>(1, pass_bytes / chunk_bytes));
    const size_t halves = n_blocks > pass_blocks ? 2 : 1;
    HIPCHK(c->inf_tok.ensure(halves * pass_blocks * inflate_tok_stride(chunk_bytes) * sizeof(uint32_t)));
    HIPCHK(c->inf_ntok.ensure(halves * pass_blocks * sizeof(uint32_t)));
    HIPCHK(c->inf_units.ensure((halves * pass_blocks * chunk_bytes + 64) * sizeof(uint16_t)));
    for (hipEvent_t &e : c->inf_ev)
        if (!e) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIPCHK(c->inf_win.ensure(n_blocks * (size_t)kMaxWindow));
    HIPCHK(c->inf_off.ensure(n_blocks * sizeof(uint64_t)));
    HIPCHK(c->inf_status.ensure((n_blocks + 2) * sizeof(uint32_t)));
    rc = init_scalars(c, 0);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(c->inf_off.p, bit_offsets, n_blocks * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    LAUNCH(K_INFLATE, launch_inflate_blocks(in, stream_bytes, c->inf_off.as<uint64_t>(), (uint32_t)n_blocks, chunk_bytes, n,
                                            pass_blocks, c->inf_tok.as<uint32_t>(), c->inf_ntok.as<uint32_t>(),
                                            c->inf_units.as<uint16_t>(), c->inf_win.as<uint8_t>(),
                                            static_cast<uint8_t *>(d_out), c->inf_status.as<uint32_t>(), c->stream,
                                            c->copy, c->inf_ev));
    rc = run_adler(c, static_cast<const uint8_t *>(d_out), n);
    if (rc) return rc;
    std::vector<uint32_t> st(n_blocks + 2, 0);
    Scalars h;
    HIPCHK(hipMemcpyAsync(st.data(), c->inf_status.p, (n_blocks + 2) * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(&h, c->scalars.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->prof) prof_collect(c);
    const uint64_t end_bit = (uint64_t)st[n_blocks] | ((uint64_t)st[n_blocks + 1] << 32);
    st.resize(n_blocks);
    rc = inflate_verdict(c, st);
    if (rc) return rc;
    if (((end_bit + 7) >> 3) + 4 != stream_bytes) {
        c->err = "the last block does not end where the Adler-32 begins";
        return ZG_ERR_DATA;
    }
    const uint32_t want = ((uint32_t)edge[2] << 24) | ((uint32_t)edge[3] << 16) | ((uint32_t)edge[4] << 8) | edge[5];
    if (want != ((h.adler[1] << 16) | h.adler[0])) {
        c->err = inflate_message(INF_BAD_CHECKSUM);
        return ZG_ERR_DATA;
    }
    return ZG_OK;
}

int zg_decompress(zg_ctx *ctx, const uint8_t *in, size_t n, uint8_t **out, size_t *out_n) {
    if (!out || !out_n) return ZG_ERR_ARG;
    const size_t offsets[2] = {0, n};
    size_t *oo = nullptr;
    const int rc = zg_decompress_batch(ctx, in, offsets, 1, out, &oo);
    if (rc) return rc;
    *out_n = oo[1];
    free(oo);
    return ZG_OK;
}

uint32_t zg_adler32_combine(uint32_t adler1, uint32_t adler2, uint64_t len2) {
    const uint64_t M = 65521;
    const uint64_t a1 = adler1 & 0xFFFF, b1 = adler1 >> 16, a2 = adler2 & 0xFFFF, b2 = adler2 >> 16;
    const uint64_t a = (a1 + a2 + M - 1) % M;
    const uint64_t b = (b1 + b2 + (len2 % M) * ((a1 + M - 1) % M)) % M;
    return (uint32_t)(b << 16 | a);
}

namespace {
// the caller holds c->mu
int adler32_device_locked(zg_ctx *c, const void *d_in, size_t n, uint32_t *out) {
    int rc = init_scalars(c, 0);
    if (rc) return rc;
    rc = run_adler(c, static_cast<const uint8_t *>(d_in), n);
    if (rc) return rc;
    Scalars h;
    HIPCHK(hipMemcpyAsync(&h, c->scalars.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->prof) prof_collect(c);
    *out = (h.adler[1] << 16) | h.adler[0];
    return ZG_OK;
}
}  // namespace

int zg_adler32_device(zg_ctx *ctx, const void *d_in, size_t n, uint32_t *out) {
    int rc = 0;
    zg_ctx *c = resolve(ctx, &rc);
    if (!c) return rc ? rc : ZG_ERR_NO_DEVICE;
    if (!out || (!d_in && n)) return ZG_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    return adler32_device_locked(c, d_in, n, out);
}

// Allocates nothing: the segment values go to the fixed scratch array of the context's device, the result to
// caller memory, and no event, buffer or context is created (a NULL ctx would create the default one).
int zg_adler32_device_noalloc(zg_ctx *c, const void *d_in, size_t n, uint32_t *d_out) {
    if (!c || !d_out || (!d_in && n)) return ZG_ERR_ARG;
    if (n > kNoAllocMax) return ZG_ERR_CAPACITY;  // more segments than the fixed scratch holds
    std::lock_guard<std::mutex> lk(c->mu);
    std::lock_guard<std::mutex> dev_lk(g_noalloc_mu[(unsigned)c->device % kNoAllocLocks]);  // held until the stream is idle
    HIPCHK(hipSetDevice(c->device));
    const uint32_t nseg = (uint32_t)((n + kNoAllocSeg - 1) / kNoAllocSeg);
    uint32_t *partial = reinterpret_cast<uint32_t *>(c->noalloc_scratch);
    uint64_t *block_part = c->noalloc_scratch + kNoAllocSegs;
    // every word the fold reads is written by this call first: nothing of the previous call's contents is used
    HIPCHK(launch_adler(static_cast<const uint8_t *>(d_in), n, kNoAllocSeg, nseg, nullptr, partial, c->stream));
    HIPCHK(launch_adler_result(partial, n, kNoAllocSeg, nseg, block_part, d_out, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return ZG_OK;
}

int zg_adler32(zg_ctx *ctx, const uint8_t *in, size_t n, uint32_t *out) {
    int rc = 0;
    zg_ctx *c = resolve(ctx, &rc);
    if (!c) return rc ? rc : ZG_ERR_NO_DEVICE;
    if (!out || (!in && n)) return ZG_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);  // one lock over upload and kernels: stage_in stays this call's
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(c->stage_in.ensure(n + 64));
    if (n) HIPCHK(hipMemcpyAsync(c->stage_in.p, in, n, hipMemcpyHostToDevice, c->stream));
    return adler32_device_locked(c, c->stage_in.p, n, out);
}

int zg_lz77(zg_ctx *ctx, const uint8_t *in, size_t n, size_t start, uint32_t window, uint32_t min_len,
            uint32_t max_len, uint32_t **tokens, size_t *n_tokens) {
    int rc = 0;
    zg_ctx *c = resolve(ctx, &rc);
    if (!c) return rc ? rc : ZG_ERR_NO_DEVICE;
    if (!tokens || !n_tokens || (!in && n) || start > n) return ZG_ERR_ARG;
    if (min_len < 3 || max_len > kMaxMatch) return ZG_ERR_UNSUPPORTED;
    if (n - start > kMaxChunkBytes) return ZG_ERR_UNSUPPORTED;  // the call is one chunk
    if (window > kMaxWindow) window = kMaxWindow;  // with_window_size / set_window_size clamp, src/lz77.rs:90-120
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(c->stage_in.ensure(n + 64));
    if (n) HIPCHK(hipMemcpyAsync(c->stage_in.p, in, n, hipMemcpyHostToDevice, c->stream));
    ChunkGrid g;
    g.data = c->stage_in.as<uint8_t>();
    g.first = start;
    g.end = n;
    g.chunk_bytes = std::max<uint64_t>(n - start, 1);
    g.n_chunks = 1;
    rc = ensure_chunk_meta(c, 1, 1);
    if (rc) return rc;
    uint32_t nt = 0;
    for (int attempt = 0;; attempt++) {
        rc = init_scalars(c, 0);
        if (rc) return rc;
        rc = run_front(c, g, 0, 1, MatchParams{window, min_len, max_len}, ZG_DYNAMIC, -1, false);
        if (rc) return rc;
        Scalars hs;
        HIPCHK(hipMemcpyAsync(&nt, c->ntok.p, sizeof nt, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(&hs, c->scalars.p, sizeof hs, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (!(hs.status & 6u)) break;
        if (attempt >= 2 || ((hs.status & 2u) && c->index_safe) || ((hs.status & 4u) && c->walk_off))
            return ZG_ERR_INTERNAL;
        if (hs.status & 2u) c->index_safe = true;
        if (hs.status & 4u) c->walk_off = true;
    }
    uint32_t *h = static_cast<uint32_t *>(malloc((nt ? nt : 1) * sizeof(uint32_t)));
    if (!h) return ZG_ERR_NOMEM;
    if (nt) {
        const hipError_t ce = hipMemcpy(h, c->tokens.p, (size_t)nt * sizeof(uint32_t), hipMemcpyDeviceToHost);
        if (ce != hipSuccess) {
            free(h);
            HIPCHK(ce);
        }
    }
    if (c->prof) prof_collect(c);
    *tokens = h;
    *n_tokens = nt;
    return ZG_OK;
}

int zg_huffman_from_tokens(zg_ctx *ctx, const uint32_t *tokens, size_t n, uint8_t ll_len[286], uint16_t ll_code[286],
                           uint8_t d_len[30], uint16_t d_code[30]) {
    int rc = 0;
    zg_ctx *c = resolve(ctx, &rc);
    if (!c) return rc ? rc : ZG_ERR_NO_DEVICE;
    if ((!tokens && n) || !ll_len || !ll_code || !d_len || !d_code) return ZG_ERR_ARG;
    for (size_t i = 0; i < n; i++) {
        const uint32_t t = tokens[i];
        if (ZG_TOK_IS_MARKER(t)) {
            if (ZG_TOK_LEN(t) < 3 || ZG_TOK_LEN(t) > 258 || ZG_TOK_DIST(t) < 1 || ZG_TOK_DIST(t) > 32768)
                return ZG_ERR_ARG;  // asserts of length_writer / distance_writer, src/compress.rs:325-326,355-356
        } else if (t > 255) {
            return ZG_ERR_ARG;
        }
    }
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(c->misc.ensure(n * sizeof(uint32_t) + 64));
    rc = ensure_chunk_meta(c, 1, 1);
    if (rc) return rc;
    if (n) HIPCHK(hipMemcpyAsync(c->misc.p, tokens, n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    LAUNCH(K_PARSE, launch_histogram(c->misc.as<uint32_t>(), n, c->freq.as<uint32_t>(), c->stream));
    LAUNCH(K_HUFFMAN, launch_huffman(1, ZG_DYNAMIC, c->freq.as<uint32_t>(), 0, c->tab.as<uint32_t>(),
                                     c->hdr.as<uint32_t>(), c->hdr_bits.as<uint32_t>(), c->block_bits.as<uint64_t>(),
                                     c->stream));
    uint32_t tab[kTabStride];
    HIPCHK(hipMemcpyAsync(tab, c->tab.p, sizeof tab, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->prof) prof_collect(c);
    for (int i = 0; i < 286; i++) {
        ll_len[i] = (uint8_t)(tab[i] >> 16);
        ll_code[i] = (uint16_t)(tab[i] & 0xFFFF);
    }
    for (int i = 0; i < 30; i++) {
        d_len[i] = (uint8_t)(tab[kDOff + i] >> 16);
        d_code[i] = (uint16_t)(tab[kDOff + i] & 0xFFFF);
    }
    return ZG_OK;
}

}  // extern "C"

namespace {
// zg_shard_begin with the context's lock held by the caller
int shard_begin_locked(zg_ctx *c, const void *d_in, size_t halo, size_t n, int strategy, const zg_opts *opts,
                       int last_rank, zg_shard_info *info) {
    if (!info || (!d_in && (n || halo)) || halo > kMaxWindow) return ZG_ERR_ARG;
    if (strategy != ZG_DYNAMIC && strategy != ZG_FIXED) return ZG_ERR_UNSUPPORTED;
    Plan plan;
    int rc = make_plan(opts, &plan);
    if (rc) return rc;
    HIPCHK(hipSetDevice(c->device));
    c->shard_active = false;
    ChunkGrid g;
    g.data = static_cast<const uint8_t *>(d_in);
    g.first = halo;
    g.end = halo + n;
    g.chunk_bytes = plan.chunk_bytes;
    // a run without bytes adds nothing to the stream -- unless it is the last: the stream of an empty input is one
    // empty final block
    g.n_chunks = n == 0 ? (last_rank ? 1u : 0u) : (uint32_t)((n + plan.chunk_bytes - 1) / plan.chunk_bytes);
    const uint32_t max_tiles = max_tiles_for(plan.chunk_bytes);
    const size_t bc = std::max<size_t>(1, std::min<size_t>(plan.batch_chunks, g.n_chunks));
    const size_t bound = (zg_compress_bound(n, strategy, opts) + 3) & ~(size_t)3;
    const MatchParams mp{kMaxWindow, 3, 258};
    // The rank's blocks are packed from bit 0 of a buffer of its own, batch by batch as in a whole-stream call;
    // zg_shard_finish moves them to the bit phase the stream gives them.
    for (int attempt = 0;; attempt++) {
        rc = ensure_chunk_meta(c, bc, max_tiles);
        if (rc) return rc;
        HIPCHK(c->shard_buf.ensure(bound + 8));
        rc = init_scalars(c, 0);
        if (rc) return rc;
        try {
            c->block_offs.assign(g.n_chunks, 0);
        } catch (...) {
            return ZG_ERR_NOMEM;
        }
        HIPCHK(c->block_off_all.ensure((size_t)g.n_chunks * sizeof(uint64_t)));
        LAUNCH(K_MISC, hipMemsetAsync(c->shard_buf.p, 0, 8, c->stream));  // see compress_device_once
        for (uint32_t c0 = 0; c0 < g.n_chunks; c0 += (uint32_t)bc) {
            const uint32_t c1 = (uint32_t)std::min<uint64_t>(g.n_chunks, (uint64_t)c0 + bc);
            const int64_t final_chunk = (last_rank && c1 == g.n_chunks) ? (int64_t)(c1 - c0 - 1) : -1;
            rc = run_front(c, g, c0, c1, mp, strategy, final_chunk, true);
            if (rc) return rc;
            rc = run_back(c, g, c0, c1, max_tiles, c->shard_buf.as<uint32_t>(), (uint64_t)bound * 8);
            if (rc) return rc;
            rc = keep_block_offsets(c, c0, c1);
            if (rc) return rc;
        }
        rc = run_adler(c, g.data + halo, n);
        if (rc) return rc;
        rc = fetch_block_offsets(c);
        if (rc) return rc;
        Scalars h;
        HIPCHK(hipMemcpyAsync(&h, c->scalars.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if ((h.status & 2u) && !c->index_safe && attempt < 2) {  // redo with the conservative index kernel
            c->index_safe = true;
            continue;
        }
        if ((h.status & 4u) && !c->walk_off && attempt < 2) {  // redo with every position searched
            c->walk_off = true;
            continue;
        }
        if (h.status & 6u) return ZG_ERR_INTERNAL;
        if (h.status & 1u) return ZG_ERR_CAPACITY;
        info->total_bits = h.carry;
        info->adler = (h.adler[1] << 16) | h.adler[0];
        info->n_chunks = g.n_chunks;
        c->shard_total_bits = h.carry;
        c->shard_active = true;
        return ZG_OK;
    }
}

// out = in shifted up by s bits (0 <= s <= 16), as 32-bit little-endian words
__global__ __launch_bounds__(256) void k_shift_bits(const uint32_t *__restrict__ in, uint64_t n_in,
                                                     uint32_t *__restrict__ out, uint64_t n_out, uint32_t s) {
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n_out) return;
    const uint32_t hi = j < n_in ? in[j] : 0u;
    const uint32_t lo = (j >= 1 && j - 1 < n_in) ? in[j - 1] : 0u;
    out[j] = s ? (hi << s) | (lo >> (32 - s)) : hi;
}

// zg_shard_finish with the context's lock held by the caller
int shard_finish_locked(zg_ctx *c, uint64_t base_bit, int first_rank, int last_rank, uint32_t adler, void *d_out,
                        size_t out_cap, size_t *out_bytes) {
    if (!d_out || !out_bytes || (reinterpret_cast<uintptr_t>(d_out) & 3)) return ZG_ERR_ARG;
    if (!c->shard_active) return ZG_ERR_ARG;
    if (first_rank && base_bit != 16) return ZG_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    Scalars *sc = c->scalars.as<Scalars>();
    const uint64_t local0 = first_rank ? 16 : (base_bit & 7);  // bit offset of the first block inside d_out
    const uint64_t end_bit = local0 + c->shard_total_bits;
    const uint64_t need = ((end_bit + 7) >> 3) + (last_rank ? 4 : 0);
    const uint64_t padded = (need + 3) & ~(uint64_t)3;
    if (out_cap < padded) return ZG_ERR_CAPACITY;
    {
        ProfScope ps(c, K_ENCODE);
        const uint64_t n_in = (c->shard_total_bits + 31) / 32, n_out = padded / 4;
        hipLaunchKernelGGL(k_shift_bits, dim3((uint32_t)((n_out + 255) / 256)), dim3(256), 0, c->stream,
                           c->shard_buf.as<uint32_t>(), n_in, static_cast<uint32_t *>(d_out), n_out, (uint32_t)local0);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipMemcpyAsync(&sc->carry, &end_bit, sizeof end_bit, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    LAUNCH(K_MISC, launch_trailer(static_cast<uint8_t *>(d_out), &sc->carry, sc->adler, adler, 1, first_rank ? 1 : 0,
                                  last_rank ? 1 : 0, &sc->out_bytes, c->stream));
    Scalars h;
    HIPCHK(hipMemcpyAsync(&h, c->scalars.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->prof) prof_collect(c);
    c->shard_active = false;
    if (h.status & 1u) return ZG_ERR_CAPACITY;
    *out_bytes = (size_t)h.out_bytes;
    return ZG_OK;
}
}  // namespace

namespace {
// ---------------------------------------------------------------------------
// The piecewise form of zg_shard_begin: the rank's blocks are packed from bit 0 of a buffer of the CALLER's, batch
// by batch, and after every batch the caller learns how many bytes of it are final -- so that it can ship them to
// the rank that assembles the stream while the next batch's kernels run.  zg_shard_next queues the next batch
// before it waits for the oldest one.
// ---------------------------------------------------------------------------
int shard_queue_batch(zg_ctx *c) {
    zg_ctx::ShardRun &r = c->run;
    const uint32_t b = r.queued;
    const uint32_t c0 = b * r.bc;
    const uint32_t c1 = (uint32_t)std::min<uint64_t>(r.g.n_chunks, (uint64_t)c0 + r.bc);
    const int64_t final_chunk = (r.last_rank && c1 == r.g.n_chunks) ? (int64_t)(c1 - c0 - 1) : -1;
    const MatchParams mp{kMaxWindow, 3, 258};
    int rc = ZG_OK;
    if (c1 > c0) {
        rc = run_front(c, r.g, c0, c1, mp, r.strategy, final_chunk, true);
        if (rc) return rc;
        rc = run_back(c, r.g, c0, c1, r.max_tiles, r.packed, r.cap_bits);
        if (rc) return rc;
        rc = keep_block_offsets(c, c0, c1);
        if (rc) return rc;
    }
    if (c1 == r.g.n_chunks) {  // the last batch: the rank's Adler-32 rides behind it
        rc = run_adler(c, r.g.data + r.halo, r.n);
        if (rc) return rc;
        rc = fetch_block_offsets(c);
        if (rc) return rc;
    }
    Scalars *sc = c->scalars.as<Scalars>();
    HIPCHK(hipMemcpyAsync(r.h_carry + b, &sc->carry, sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipEventRecord(r.ev[b], c->stream));
    r.queued = b + 1;
    return ZG_OK;
}

// out word j (from the word base_bit lies in) = the run's bits shifted up by base_bit % 32; the first word is ORed
// into what the stream holds already (the end of the run before, or the header), the last one stored with zeros
// above the run's end
__global__ __launch_bounds__(256) void k_merge_bits(const uint32_t *__restrict__ src, uint64_t n_bits,
                                                     uint32_t *__restrict__ dst, uint64_t base_bit) {
    const uint32_t s = (uint32_t)(base_bit & 31u);
    const uint64_t n_out = (s + n_bits + 31) >> 5, n_src = (n_bits + 31) >> 5;
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n_out) return;
    auto word = [&](uint64_t i) -> uint32_t {  // source word i, without the bits behind the run's end
        if (i >= n_src) return 0u;
        uint32_t v = src[i];
        if (i + 1 == n_src && (n_bits & 31u)) v &= (1u << (n_bits & 31u)) - 1u;
        return v;
    };
    const uint32_t hi = word(j);
    const uint32_t lo = j >= 1 ? word(j - 1) : 0u;
    const uint32_t v = s ? (hi << s) | (lo >> (32u - s)) : hi;
    uint32_t *d = dst + (base_bit >> 5) + j;
    if (j == 0)
        *d |= v;  // alone on this word: the run before has been merged by an earlier launch on the same stream
    else
        *d = v;
}
}  // namespace

extern "C" {

int zg_shard_open(zg_ctx *ctx, const void *d_in, size_t halo, size_t n, int strategy, const zg_opts *opts,
                  int last_rank, void *d_packed, size_t packed_cap) {
    int rc = 0;
    zg_ctx *c = resolve(ctx, &rc);
    if (!c) return rc ? rc : ZG_ERR_NO_DEVICE;
    if ((!d_in && (n || halo)) || halo > kMaxWindow || !d_packed || (reinterpret_cast<uintptr_t>(d_packed) & 3))
        return ZG_ERR_ARG;
    if (strategy != ZG_DYNAMIC && strategy != ZG_FIXED) return ZG_ERR_UNSUPPORTED;
    Plan plan;
    rc = make_plan(opts, &plan);
    if (rc) return rc;
    if (packed_cap < zg_compress_bound(n, strategy, opts) + 8) return ZG_ERR_CAPACITY;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    zg_ctx::ShardRun &r = c->run;
    r.open = false;
    c->shard_active = false;
    r.g = ChunkGrid{};
    r.g.data = static_cast<const uint8_t *>(d_in);
    r.g.first = halo;
    r.g.end = halo + n;
    r.g.chunk_bytes = plan.chunk_bytes;
    // (a run without bytes adds nothing to the stream unless it is the last: see zg_shard_begin)
    r.g.n_chunks = n == 0 ? (last_rank ? 1u : 0u) : (uint32_t)((n + plan.chunk_bytes - 1) / plan.chunk_bytes);
    r.strategy = strategy;
    r.last_rank = last_rank;
    r.max_tiles = max_tiles_for(plan.chunk_bytes);
    r.bc = (uint32_t)std::max<size_t>(1, std::min<size_t>(plan.batch_chunks, r.g.n_chunks));
    r.n_batches = std::max<uint32_t>(1, (r.g.n_chunks + r.bc - 1) / r.bc);  // a run of no chunks: one pass that gives nothing
    r.queued = r.reported = 0;
    r.packed = static_cast<uint32_t *>(d_packed);
    r.cap_bits = (uint64_t)(packed_cap & ~(size_t)3) * 8;
    r.bytes_reported = 0;
    r.halo = halo;
    r.n = n;
    try {
        while (r.ev.size() < r.n_batches) {
            hipEvent_t e;
            HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            r.ev.push_back(e);
        }
        c->block_offs.assign(r.g.n_chunks, 0);
    } catch (...) {
        return ZG_ERR_NOMEM;
    }
    if (r.h_carry_cap < r.n_batches) {
        if (r.h_carry) (void)hipHostFree(r.h_carry);
        r.h_carry = nullptr;
        r.h_carry_cap = 0;
        HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&r.h_carry), (size_t)r.n_batches * sizeof(uint64_t), hipHostMallocDefault));
        r.h_carry_cap = r.n_batches;
    }
    rc = ensure_chunk_meta(c, r.bc, r.max_tiles);
    if (rc) return rc;
    HIPCHK(c->block_off_all.ensure((size_t)r.g.n_chunks * sizeof(uint64_t)));
    rc = init_scalars(c, 0);
    if (rc) return rc;
    LAUNCH(K_MISC, hipMemsetAsync(d_packed, 0, 8, c->stream));  // the word the first block begins in (compress_device_once)
    rc = shard_queue_batch(c);
    if (rc) return rc;
    r.open = true;
    return ZG_OK;
}

int zg_shard_next(zg_ctx *ctx, zg_shard_piece *piece, zg_shard_info *info) {
    int rc = 0;
    zg_ctx *c = resolve(ctx, &rc);
    if (!c) return rc ? rc : ZG_ERR_NO_DEVICE;
    if (!piece || !info) return ZG_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    zg_ctx::ShardRun &r = c->run;
    if (!r.open || r.reported >= r.n_batches) return ZG_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    if (r.queued < r.n_batches) {  // the next batch's kernels go out before this thread waits for the oldest batch
        rc = shard_queue_batch(c);
        if (rc) {
            r.open = false;
            return rc;
        }
    }
    const uint32_t b = r.reported;
    HIPCHK(hipEventSynchronize(r.ev[b]));
    const uint64_t bits = r.h_carry[b];
    const bool last = b + 1 == r.n_batches;
    const uint64_t done = last ? (bits + 7) >> 3 : bits >> 3;  // bytes that will not change any more
    piece->byte_offset = r.bytes_reported;
    piece->n_bytes = done - r.bytes_reported;
    piece->bits_so_far = bits;
    piece->last = last ? 1 : 0;
    r.bytes_reported = done;
    r.reported = b + 1;
    if (last) {
        Scalars h;
        HIPCHK(hipMemcpyAsync(&h, c->scalars.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (c->prof) prof_collect(c);
        r.open = false;
        // pieces have left already: a run that has to be redone (never seen) is the caller's to repeat
        if (h.status & 2u) c->index_safe = true;
        if (h.status & 4u) c->walk_off = true;
        if (h.status & 6u) return ZG_ERR_INTERNAL;
        if (h.status & 1u) return ZG_ERR_CAPACITY;
        info->total_bits = h.carry;
        info->adler = (h.adler[1] << 16) | h.adler[0];
        info->n_chunks = r.g.n_chunks;
    }
    return ZG_OK;
}

int zg_shard_merge(zg_ctx *ctx, const void *d_src, uint64_t n_bits, void *d_dst, size_t dst_cap, uint64_t base_bit,
                   int first_run) {
    int rc = 0;
    zg_ctx *c = resolve(ctx, &rc);
    if (!c) return rc ? rc : ZG_ERR_NO_DEVICE;
    if (!d_dst || (!d_src && n_bits) || (reinterpret_cast<uintptr_t>(d_dst) & 3) || (reinterpret_cast<uintptr_t>(d_src) & 3))
        return ZG_ERR_ARG;
    if (first_run && base_bit != 16) return ZG_ERR_ARG;
    if ((((base_bit + n_bits + 31) >> 5) + 2) * 4 > dst_cap) return ZG_ERR_CAPACITY;  // room for the Adler-32 as well
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    if (first_run) {  // 78 01 (src/compress.rs:53-72), zeros above it
        const uint32_t head = 0x00000178u;
        HIPCHK(hipMemcpyAsync(d_dst, &head, sizeof head, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));  // `head` is a stack object
    }
    if (n_bits) {
        ProfScope ps(c, K_ENCODE);
        const uint64_t n_out = ((base_bit & 31u) + n_bits + 31) >> 5;
        hipLaunchKernelGGL(k_merge_bits, dim3((uint32_t)((n_out + 255) / 256)), dim3(256), 0, c->stream,
                           static_cast<const uint32_t *>(d_src), n_bits, static_cast<uint32_t *>(d_dst), base_bit);
        HIPCHK(hipGetLastError());
    }
    return ZG_OK;
}

int zg_shard_seal(zg_ctx *ctx, void *d_dst, size_t dst_cap, uint64_t end_bit, uint32_t adler, size_t *out_bytes) {
    int rc = 0;
    zg_ctx *c = resolve(ctx, &rc);
    if (!c) return rc ? rc : ZG_ERR_NO_DEVICE;
    if (!d_dst || !out_bytes) return ZG_ERR_ARG;
    if (((end_bit + 7) >> 3) + 4 > dst_cap) return ZG_ERR_CAPACITY;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(c->scalars.ensure(sizeof(Scalars)));
    Scalars *sc = c->scalars.as<Scalars>();
    HIPCHK(hipMemcpyAsync(&sc->carry, &end_bit, sizeof end_bit, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    LAUNCH(K_MISC, launch_trailer(static_cast<uint8_t *>(d_dst), &sc->carry, sc->adler, adler, 1, 0, 1, &sc->out_bytes,
                                  c->stream));
    uint64_t nb = 0;
    HIPCHK(hipMemcpyAsync(&nb, &sc->out_bytes, sizeof nb, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->prof) prof_collect(c);
    *out_bytes = (size_t)nb;
    return ZG_OK;
}

}  // extern "C"

extern "C" {

int zg_shard_begin(zg_ctx *ctx, const void *d_in, size_t halo, size_t n, int strategy, const zg_opts *opts,
                   int last_rank, zg_shard_info *info) {
    int rc = 0;
    zg_ctx *c = resolve(ctx, &rc);
    if (!c) return rc ? rc : ZG_ERR_NO_DEVICE;
    std::lock_guard<std::mutex> lk(c->mu);
    return shard_begin_locked(c, d_in, halo, n, strategy, opts, last_rank, info);
}

int zg_shard_finish(zg_ctx *ctx, uint64_t base_bit, int first_rank, int last_rank, uint32_t adler, void *d_out,
                    size_t out_cap, size_t *out_bytes) {
    int rc = 0;
    zg_ctx *c = resolve(ctx, &rc);
    if (!c) return rc ? rc : ZG_ERR_NO_DEVICE;
    std::lock_guard<std::mutex> lk(c->mu);
    return shard_finish_locked(c, base_bit, first_rank, last_rank, adler, d_out, out_cap, out_bytes);
}

// ---------------------------------------------------------------------------
// zg_compress_multi: one stream from several GPUs, driven from one process.  The input is cut into contiguous runs
// of whole chunks, one run per device (zg_shard_begin on a context of its own, one host thread per device); the
// three integers per run that fix the layout of the stream are exchanged on the host; every device then moves
// its blocks to their bit phase (zg_shard_finish) and copies them straight into its own byte range of the result.
// Two neighbouring runs share at most one byte, which is merged on the host afterwards.
// ---------------------------------------------------------------------------
namespace {
std::mutex g_multi_mu;
std::vector<zg_ctx *> g_multi_ctx;  // context of run r, kept between calls

struct MultiRun {
    zg_ctx *c = nullptr;
    size_t start = 0, n = 0, halo = 0;
    zg_shard_info info{};
    uint64_t base_bit = 0;
    size_t nb = 0;         // bytes of the run at its bit phase
    uint8_t first_byte = 0;
    int rc = ZG_OK;
};

int multi_begin(MultiRun &r, const uint8_t *in, int strategy, const zg_opts *opts, bool last) {
    zg_ctx *c = r.c;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(c->stage_in.ensure(r.halo + r.n + 64));
        if (r.halo + r.n)
            HIPCHK(hipMemcpyAsync(c->stage_in.p, in + r.start - r.halo, r.halo + r.n, hipMemcpyHostToDevice, c->stream));
    }
    return zg_shard_begin(c, c->stage_in.p, r.halo, r.n, strategy, opts, last ? 1 : 0, &r.info);
}

int multi_finish(MultiRun &r, bool first, bool last, uint32_t adler, uint8_t *out, size_t out_total) {
    zg_ctx *c = r.c;
    const uint64_t local0 = first ? 16 : (r.base_bit & 7);
    const size_t need = (size_t)(((local0 + r.info.total_bits + 7) >> 3) + (last ? 4 : 0));
    const size_t cap = (need + 3 + 8) & ~(size_t)3;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(c->stage_out.ensure(cap));
    }
    int rc = zg_shard_finish(c, r.base_bit, first ? 1 : 0, last ? 1 : 0, adler, c->stage_out.p, cap, &r.nb);
    if (rc) return rc;
    const size_t off = first ? 0 : (size_t)(r.base_bit >> 3);
    if (off + r.nb > out_total || r.nb == 0) return ZG_ERR_INTERNAL;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(hipSetDevice(c->device));
    // the first byte may be shared with the run before: it is merged by the caller once every run has landed
    HIPCHK(hipMemcpy(&r.first_byte, c->stage_out.p, 1, hipMemcpyDeviceToHost));
    if (r.nb > 1)
        HIPCHK(hipMemcpy(out + off + 1, c->stage_out.as<uint8_t>() + 1, r.nb - 1, hipMemcpyDeviceToHost));
    return ZG_OK;
}
}  // namespace

int zg_compress_multi(const int *devices, int n_devices, const uint8_t *in, size_t n, int strategy,
                      const zg_opts *opts, uint8_t **out, size_t *out_n) {
    if (!devices || n_devices < 1 || n_devices > 64 || !out || !out_n || (!in && n)) return ZG_ERR_ARG;
    Plan plan;
    int rc = make_plan(opts, &plan);
    if (rc) return rc;
    if (strategy == ZG_AUTO) strategy = auto_choice(in, n);
    if (strategy < ZG_AUTO || strategy > ZG_RAW) return ZG_ERR_ARG;
    std::lock_guard<std::mutex> guard(g_multi_mu);
    auto ctx_of = [&](size_t r, int device, zg_ctx **c) -> int {
        if (g_multi_ctx.size() <= r) g_multi_ctx.resize(r + 1, nullptr);
        if (g_multi_ctx[r] && g_multi_ctx[r]->device != device) {
            zg_destroy(g_multi_ctx[r]);
            g_multi_ctx[r] = nullptr;
        }
        if (!g_multi_ctx[r]) {
            const int e = zg_init(device, &g_multi_ctx[r]);
            if (e) return e;
        }
        *c = g_multi_ctx[r];
        return ZG_OK;
    };
    const size_t n_chunks = n == 0 ? 1 : (n + plan.chunk_bytes - 1) / plan.chunk_bytes;
    const size_t n_runs = std::min<size_t>((size_t)n_devices, n_chunks);
    if (strategy == ZG_RAW || n_runs == 1) {  // stored blocks are a plain copy; a single run needs no assembly
        zg_ctx *c = nullptr;
        rc = ctx_of(0, devices[0], &c);
        return rc ? rc : zg_compress(c, in, n, strategy, opts, out, out_n);
    }
    std::vector<MultiRun> runs(n_runs);
    for (size_t r = 0; r < n_runs; r++) {
        rc = ctx_of(r, devices[r], &runs[r].c);
        if (rc) return rc;
        const size_t c0 = r * n_chunks / n_runs, c1 = (r + 1) * n_chunks / n_runs;
        runs[r].start = std::min(n, c0 * plan.chunk_bytes);
        runs[r].n = std::min(n, c1 * plan.chunk_bytes) - runs[r].start;
        runs[r].halo = std::min<size_t>(kMaxWindow, runs[r].start);
    }
    auto in_parallel = [&](auto &&fn) -> int {
        std::vector<std::thread> pool;
        try {
            for (size_t r = 1; r < n_runs; r++) pool.emplace_back([&, r] { runs[r].rc = fn(runs[r], r); });
        } catch (...) {  // no more threads to be had: the remaining runs take their turn on this one
            for (size_t r = pool.size() + 1; r < n_runs; r++) runs[r].rc = fn(runs[r], r);
        }
        runs[0].rc = fn(runs[0], 0);
        for (auto &t : pool) t.join();
        for (auto &r : runs)
            if (r.rc) return r.rc;
        return ZG_OK;
    };
    // A call that fails leaves nothing behind on any device: no run waiting for its zg_shard_finish, no staging memory.
    auto give_up = [&](int code) -> int {
        for (auto &r : runs) {
            if (!r.c) continue;
            std::lock_guard<std::mutex> lk(r.c->mu);
            if (hipSetDevice(r.c->device) != hipSuccess) continue;
            (void)hipStreamSynchronize(r.c->stream);
            r.c->shard_active = false;
            r.c->stage_in.release();
            r.c->stage_out.release();
            r.c->shard_buf.release();
        }
        return code;
    };
    // The caller's input is pageable; page-locking it for the call lets every device read its run with one DMA
    // (measured against plain pageable copies; see ZLIB_AMD_MULTI_PIN).
    const char *pin_env = getenv("ZLIB_AMD_MULTI_PIN");
    const bool pin = pin_env && atoi(pin_env) > 0 && n >= ((size_t)64 << 20);
    const bool pinned_in = pin && hipHostRegister(const_cast<uint8_t *>(in), n, hipHostRegisterDefault) == hipSuccess;
    if (pin && !pinned_in) (void)hipGetLastError();
    auto unpin_in = [&] {
        if (pinned_in) (void)hipHostUnregister(const_cast<uint8_t *>(in));
    };
    rc = in_parallel([&](MultiRun &r, size_t i) { return multi_begin(r, in, strategy, opts, i + 1 == n_runs); });
    unpin_in();
    if (rc) return give_up(rc);
    // the layout of the stream: 16 header bits, the runs' blocks back to back, padding to a byte, the Adler-32
    uint64_t bit = 16;
    uint32_t adler = 1;
    for (auto &r : runs) {
        r.base_bit = bit;
        bit += r.info.total_bits;
        adler = zg_adler32_combine(adler, r.info.adler, r.n);
    }
    const size_t total = (size_t)((bit + 7) >> 3) + 4;
    uint8_t *h = static_cast<uint8_t *>(malloc(total));
    if (!h) return give_up(ZG_ERR_NOMEM);
    const bool pinned_out = pin && hipHostRegister(h, total, hipHostRegisterDefault) == hipSuccess;
    if (pin && !pinned_out) (void)hipGetLastError();
    rc = in_parallel([&](MultiRun &r, size_t i) { return multi_finish(r, i == 0, i + 1 == n_runs, adler, h, total); });
    if (pinned_out) (void)hipHostUnregister(h);
    if (rc) {
        free(h);
        return give_up(rc);
    }
    for (size_t i = 0; i < n_runs; i++) {  // in order: a run's first byte joins the previous run's last
        const size_t off = i == 0 ? 0 : (size_t)(runs[i].base_bit >> 3);
        if (i != 0 && (runs[i].base_bit & 7))
            h[off] |= runs[i].first_byte;
        else
            h[off] = runs[i].first_byte;
    }
    *out = h;
    *out_n = total;
    return ZG_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// zg_compress_stream / zg_compress_file: the input arrives through a callback and the stream leaves through
// another, a segment (a whole number of chunks, one pass over the scratch buffers by default) at a time.  Three
// threads work on neighbouring segments: one reads segment k + 1 into pinned memory and uploads it, this one
// compresses segment k (zg_shard_begin / zg_shard_finish on the segment and the 32 KiB before it), one downloads
// and hands on segment k - 1.  Two neighbouring segments share at most one byte of the stream; it is held back
// until the second one has arrived.
// ---------------------------------------------------------------------------
namespace {

struct StreamIn {  // one of two input slots
    uint8_t *h = nullptr;  // pinned: halo + segment
    void *d = nullptr;
    size_t halo = 0, n = 0;
    bool last = false;
};
struct StreamOut {  // one of two output slots and what the writer is to do with it
    DevBuf d;
    uint8_t *h = nullptr;  // pinned, grown on demand by the writer
    size_t h_cap = 0;
    size_t off = 0, len = 0;       // bytes [off, off + len) of d belong to the stream
    bool or_first = false;         // the first of them shares its byte with the byte held back
    bool hold_last = false;        // the last of them is not complete yet
    int64_t clear_at = -1;         // stored blocks: header byte whose BFINAL bit does not belong to the stream
    uint8_t tail[4] = {0, 0, 0, 0};
    size_t tail_n = 0;             // bytes appended behind them (Adler-32 of a stream of stored blocks)
    bool last = false;
};

struct StreamPipe {
    std::mutex mu;
    std::condition_variable cv;
    size_t filled = 0, consumed = 0, packed = 0, written = 0;
    int rc = ZG_OK;  // first failure of any of the three threads
    std::string what;
    void fail(int code, const char *msg) {
        std::lock_guard<std::mutex> lk(mu);
        if (!rc) {
            rc = code;
            what = msg;
        }
        cv.notify_all();
    }
    template <class F> bool wait(F &&ready) {  // false: another thread has failed
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return rc != ZG_OK || ready(); });
        return rc == ZG_OK;
    }
    template <class F> void post(F &&change) {
        std::lock_guard<std::mutex> lk(mu);
        change();
        cv.notify_all();
    }
};

size_t gcd_sz(size_t a, size_t b) {
    while (b) {
        const size_t t = a % b;
        a = b;
        b = t;
    }
    return a;
}

int compress_stream_locked(zg_ctx *c, zg_read_fn read_fn, void *read_user, zg_write_fn write_fn, void *write_user,
                           int strategy, const zg_opts *opts, uint64_t size_hint, uint64_t *in_total,
                           uint64_t *out_total) {
    Plan plan;
    int rc = make_plan(opts, &plan);
    if (rc) return rc;
    if (strategy < ZG_AUTO || strategy > ZG_RAW) return ZG_ERR_ARG;
    // A segment is a whole number of chunks and of 16 KiB stored blocks, so that the blocks of the stream are the
    // blocks of a call on the whole input whatever the strategy turns out to be.
    const size_t unit = plan.chunk_bytes / gcd_sz(plan.chunk_bytes, kStoredBlock) * kStoredBlock;
    if (unit > ((size_t)1 << 30)) return ZG_ERR_UNSUPPORTED;
    size_t seg = std::max<size_t>(1, plan.batch_chunks * plan.chunk_bytes / unit) * unit;
    if (size_hint && size_hint < seg) seg = (size_t)((size_hint + unit) / unit) * unit;  // room for the hint + 1 byte
    // never shorter than the window: the halo of a segment is taken from the one segment before it, and every chunk
    // must see the 32 KiB of input before it, as in a call on the whole input
    const size_t seg_min = (kMaxWindow + unit - 1) / unit * unit;
    if (seg < seg_min) seg = seg_min;
    const size_t halo_max = kMaxWindow;
    HIPCHK(hipSetDevice(c->device));

    StreamIn in[2];
    StreamOut out[2];
    StreamPipe pipe;
    hipStream_t rs = nullptr, ws = nullptr;
    auto release = [&] {
        for (auto &s : in) {
            if (s.h) (void)hipHostFree(s.h);
            if (s.d) (void)hipFree(s.d);
        }
        for (auto &s : out) {
            if (s.h) (void)hipHostFree(s.h);
            s.d.release();
        }
        if (rs) (void)hipStreamDestroy(rs);
        if (ws) (void)hipStreamDestroy(ws);
    };
    hipError_t e = hipStreamCreateWithFlags(&rs, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ws, hipStreamNonBlocking);
    for (auto &s : in) {
        if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&s.h), halo_max + seg, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc(&s.d, halo_max + seg + 64);
    }
    if (e != hipSuccess) {
        release();
        return fail_hip(c, e, "buffers of zg_compress_stream");
    }

    uint64_t n_in = 0, n_out = 0;
    const int device = c->device;

    auto reader = [&] {
        if (hipSetDevice(device) != hipSuccess) return pipe.fail(ZG_ERR_HIP, "hipSetDevice (reader)");
        uint8_t carry = 0;
        for (size_t k = 0;; k++) {
            if (!pipe.wait([&] { return k < pipe.consumed + 2; })) return;
            StreamIn &s = in[k & 1];
            const StreamIn &prev = in[(k & 1) ^ 1];
            s.halo = k ? halo_max : 0;  // a segment that has a successor is full
            if (k) memcpy(s.h, prev.h + prev.halo + prev.n - s.halo, s.halo);
            uint8_t *p = s.h + s.halo;
            size_t n = 0;
            if (k) p[n++] = carry;
            bool eof = false;
            while (n < seg && !eof) {
                const size_t r = read_fn(read_user, p + n, seg - n);
                if (r == (size_t)-1 || r > seg - n) return pipe.fail(ZG_ERR_IO, "the read callback failed");
                eof = r == 0;
                n += r;
            }
            if (!eof) {  // is there a byte behind this segment?  it opens the next one
                const size_t r = read_fn(read_user, &carry, 1);
                if (r == (size_t)-1 || r > 1) return pipe.fail(ZG_ERR_IO, "the read callback failed");
                eof = r == 0;
            }
            s.n = n;
            s.last = eof;
            if (s.halo + n) {
                hipError_t he = hipMemcpyAsync(s.d, s.h, s.halo + n, hipMemcpyHostToDevice, rs);
                if (he == hipSuccess) he = hipStreamSynchronize(rs);
                if (he != hipSuccess) return pipe.fail(ZG_ERR_HIP, hipGetErrorString(he));
            }
            n_in += n;
            pipe.post([&] { pipe.filled = k + 1; });
            if (eof) return;
        }
    };

    auto writer = [&] {
        if (hipSetDevice(device) != hipSuccess) return pipe.fail(ZG_ERR_HIP, "hipSetDevice (writer)");
        uint8_t held = 0;
        bool have_held = false;
        for (size_t k = 0;; k++) {
            if (!pipe.wait([&] { return pipe.packed > k; })) return;
            StreamOut &s = out[k & 1];
            if (s.h_cap < s.len + 8) {
                if (s.h) (void)hipHostFree(s.h);
                s.h = nullptr;
                s.h_cap = s.len + (s.len >> 3) + 4096;
                if (hipHostMalloc(reinterpret_cast<void **>(&s.h), s.h_cap, hipHostMallocDefault) != hipSuccess) {
                    s.h_cap = 0;
                    return pipe.fail(ZG_ERR_NOMEM, "pinned memory for the stream's bytes");
                }
            }
            if (s.len) {
                hipError_t he = hipMemcpyAsync(s.h, s.d.as<uint8_t>() + s.off, s.len, hipMemcpyDeviceToHost, ws);
                if (he == hipSuccess) he = hipStreamSynchronize(ws);
                if (he != hipSuccess) return pipe.fail(ZG_ERR_HIP, hipGetErrorString(he));
            }
            const bool last = s.last;
            uint8_t *p = s.h;
            size_t len = s.len;
            if (s.clear_at >= 0 && (size_t)s.clear_at >= s.off && (size_t)s.clear_at - s.off < len)
                p[(size_t)s.clear_at - s.off] &= 0xFE;
            if (have_held) {
                if (s.or_first && len) {
                    p[0] |= held;
                } else if (write_fn(write_user, &held, 1) != 0) {
                    return pipe.fail(ZG_ERR_IO, "the write callback failed");
                } else {
                    n_out += 1;
                }
                have_held = false;
            }
            if (s.hold_last && len) {
                held = p[--len];
                have_held = true;
            }
            if (len && write_fn(write_user, p, len) != 0) return pipe.fail(ZG_ERR_IO, "the write callback failed");
            n_out += len;
            if (s.tail_n) {
                if (write_fn(write_user, s.tail, s.tail_n) != 0) return pipe.fail(ZG_ERR_IO, "the write callback failed");
                n_out += s.tail_n;
            }
            pipe.post([&] { pipe.written = k + 1; });
            if (last) return;
        }
    };

    std::thread rt, wt;
    try {
        rt = std::thread(reader);
        wt = std::thread(writer);
    } catch (...) {
        pipe.fail(ZG_ERR_NOMEM, "no thread for the stream's reader or writer");
    }

    // this thread: the kernels
    uint64_t bit = 16;
    uint32_t adler = 1;
    for (size_t k = 0;; k++) {
        if (!pipe.wait([&] { return pipe.filled > k; })) break;
        StreamIn &si = in[k & 1];
        if (k == 0 && strategy == ZG_AUTO) {
            // the length enters the choice below 8 KiB only; a first segment with a successor is longer than that
            strategy = auto_choice(si.h, si.last ? si.n : std::max<size_t>(si.n + 1, 8192));
        }
        if (!pipe.wait([&] { return k < pipe.written + 2; })) break;
        StreamOut &so = out[k & 1];
        so = StreamOut{std::move(so.d), so.h, so.h_cap};
        so.last = si.last;
        int r = ZG_OK;
        if (strategy == ZG_RAW) {
            const size_t cap = (stored_size(si.n) + 3) & ~(size_t)3;
            hipError_t he = so.d.ensure(cap);
            if (he != hipSuccess) {
                pipe.fail(fail_hip(c, he, "so.d.ensure"), "device memory for the stream's bytes");
                break;
            }
            size_t got = 0;
            r = compress_device_locked(c, static_cast<const uint8_t *>(si.d) + si.halo, si.n, ZG_RAW, opts,
                                       so.d.as<uint8_t>(), cap, &got);
            uint8_t t[4] = {0, 0, 0, 0};
            if (!r && got >= 6 &&
                hipMemcpy(t, so.d.as<uint8_t>() + got - 4, 4, hipMemcpyDeviceToHost) != hipSuccess)
                r = ZG_ERR_HIP;
            if (!r && got < 6) r = ZG_ERR_INTERNAL;
            if (!r) {
                const uint32_t a = ((uint32_t)t[0] << 24) | ((uint32_t)t[1] << 16) | ((uint32_t)t[2] << 8) | t[3];
                adler = k ? zg_adler32_combine(adler, a, si.n) : a;
                so.off = k ? 2 : 0;
                so.len = got - 4 - so.off;
                const size_t nblk = si.n == 0 ? 1 : (si.n + kStoredBlock - 1) / kStoredBlock;
                if (!si.last) so.clear_at = (int64_t)(2 + (nblk - 1) * (kStoredBlock + 5));
                if (si.last) {
                    so.tail_n = 4;
                    for (int i = 0; i < 4; i++) so.tail[i] = (uint8_t)(adler >> (24 - 8 * i));
                }
            }
        } else {
            zg_shard_info info{};
            r = shard_begin_locked(c, si.d, si.halo, si.n, strategy, opts, si.last ? 1 : 0, &info);
            if (!r) {
                adler = k ? zg_adler32_combine(adler, info.adler, si.n) : info.adler;
                const uint64_t local0 = k ? (bit & 7) : 16;
                const size_t need = (size_t)(((local0 + info.total_bits + 7) >> 3) + (si.last ? 4 : 0));
                const size_t cap = (need + 3 + 8) & ~(size_t)3;
                hipError_t he = so.d.ensure(cap);
                if (he != hipSuccess) {
                    pipe.fail(fail_hip(c, he, "so.d.ensure"), "device memory for the stream's bytes");
                    break;
                }
                size_t nb = 0;
                r = shard_finish_locked(c, bit, k == 0, si.last ? 1 : 0, adler, so.d.p, cap, &nb);
                so.len = nb;
                so.or_first = k != 0 && (bit & 7) != 0;
                bit += info.total_bits;
                so.hold_last = !si.last && (bit & 7) != 0;
            }
        }
        if (r) {
            pipe.fail(r, "a segment could not be compressed");
            break;
        }
        const bool last = si.last;
        pipe.post([&] {
            pipe.consumed = k + 1;
            pipe.packed = k + 1;
        });
        if (last) break;
    }
    if (rt.joinable()) rt.join();
    if (wt.joinable()) wt.join();
    release();
    if (pipe.rc) {
        if (pipe.rc != ZG_ERR_HIP || c->err.empty()) c->err = pipe.what;
        return pipe.rc;
    }
    if (in_total) *in_total = n_in;
    if (out_total) *out_total = n_out;
    return ZG_OK;
}

size_t file_read(void *user, uint8_t *buf, size_t cap) {
    FILE *f = static_cast<FILE *>(user);
    const size_t r = fread(buf, 1, cap, f);
    return (r == 0 && ferror(f)) ? (size_t)-1 : r;
}
int file_write(void *user, const uint8_t *buf, size_t n) {
    return fwrite(buf, 1, n, static_cast<FILE *>(user)) == n ? 0 : -1;
}
}  // namespace

extern "C" {

int zg_compress_stream(zg_ctx *ctx, zg_read_fn read_fn, void *read_user, zg_write_fn write_fn, void *write_user,
                       int strategy, const zg_opts *opts, uint64_t size_hint, uint64_t *in_total,
                       uint64_t *out_total) {
    int rc = 0;
    zg_ctx *c = resolve(ctx, &rc);
    if (!c) return rc ? rc : ZG_ERR_NO_DEVICE;
    if (!read_fn || !write_fn) return ZG_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    return compress_stream_locked(c, read_fn, read_user, write_fn, write_user, strategy, opts, size_hint, in_total,
                                  out_total);
}

int zg_compress_file(zg_ctx *ctx, const char *in_path, const char *out_path, int strategy, const zg_opts *opts,
                     uint64_t *in_bytes, uint64_t *out_bytes) {
    int rc = 0;
    zg_ctx *c = resolve(ctx, &rc);
    if (!c) return rc ? rc : ZG_ERR_NO_DEVICE;
    if (!in_path || !out_path) return ZG_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    FILE *fi = fopen(in_path, "rb");
    if (!fi) {
        c->err = std::string("cannot open ") + in_path;
        return ZG_ERR_IO;
    }
    uint64_t hint = 0;
    struct stat st, st_out;
    const bool have_st = fstat(fileno(fi), &st) == 0;
    if (have_st && S_ISREG(st.st_mode) && st.st_size > 0) hint = (uint64_t)st.st_size;
    // the output is created (and truncated) before the first read: it must not be the input
    if (have_st && stat(out_path, &st_out) == 0 && st.st_dev == st_out.st_dev && st.st_ino == st_out.st_ino) {
        fclose(fi);
        c->err = std::string("input and output are the same file: ") + in_path;
        return ZG_ERR_ARG;
    }
    FILE *fo = fopen(out_path, "wb");
    if (!fo) {
        fclose(fi);
        c->err = std::string("cannot create ") + out_path;
        return ZG_ERR_IO;
    }
    rc = compress_stream_locked(c, file_read, fi, file_write, fo, strategy, opts, hint, in_bytes, out_bytes);
    fclose(fi);
    if (fclose(fo) != 0 && !rc) {
        c->err = std::string("cannot finish ") + out_path;
        rc = ZG_ERR_IO;
    }
    if (rc) (void)remove(out_path);  // no partial stream is left behind
    return rc;
}

#ifdef ZG_INDEX_DEBUG
// Measurement hook (debug builds only): the three indexes of n bytes of device memory, `reps` times, nothing else.
// Returns the mean milliseconds of a launch, or a negative status.
double zg_debug_index_only(zg_ctx *c, const void *d_in, size_t n, int reps) {
    if (!c || !d_in || n < 65536) return -1.0;
    std::lock_guard<std::mutex> lk(c->mu);
    if (hipSetDevice(c->device) != hipSuccess) return -2.0;
    if (init_scalars(c, 0)) return -3.0;
    const uint64_t span = n;
    DevBuf *bufs[6] = {&c->far, &c->next, &c->far4, &c->next4, &c->far3, &c->next3};
    for (DevBuf *b : bufs)
        if (b->ensure(span * 2 + 64) != hipSuccess) return -4.0;
    IndexSet set;
    set.add(5, c->far.as<uint16_t>(), c->next.as<uint16_t>());
    set.add(4, c->far4.as<uint16_t>(), c->next4.as<uint16_t>());
    set.add(3, c->far3.as<uint16_t>(), c->next3.as<uint16_t>());
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -5.0;
    uint32_t *status = &c->scalars.as<Scalars>()->status;
    const uint8_t *data = static_cast<const uint8_t *>(d_in);
    (void)launch_index(data, n, 0, n, 0, kMaxWindow, set, 0, status, false, 0, 0, c->stream);  // warm-up
    (void)hipEventRecord(a, c->stream);
    for (int i = 0; i < reps; i++)
        if (launch_index(data, n, 0, n, 0, kMaxWindow, set, 0, status, false, 0, 0, c->stream) != hipSuccess) return -6.0;
    (void)hipEventRecord(b, c->stream);
    if (hipStreamSynchronize(c->stream) != hipSuccess) return -7.0;
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    return (double)ms / (reps > 0 ? reps : 1);
}
#endif

size_t zg_last_block_offsets(zg_ctx *ctx, uint64_t *bit_offsets, size_t cap) {
    int rc = 0;
    zg_ctx *c = resolve(ctx, &rc);
    if (!c) return 0;
    std::lock_guard<std::mutex> lk(c->mu);
    const size_t n = c->block_offs.size();
    if (bit_offsets)
        for (size_t i = 0; i < n && i < cap; i++) bit_offsets[i] = c->block_offs[i];
    return n;
}

int zg_profile_enable(zg_ctx *ctx, int on) {
    int rc = 0;
    zg_ctx *c = resolve(ctx, &rc);
    if (!c) return rc ? rc : ZG_ERR_NO_DEVICE;
    std::lock_guard<std::mutex> lk(c->mu);
    if (on < 0 || on >= 2 + K_COUNT) return ZG_ERR_ARG;
    c->prof = on != 0;
    c->prof_only = on >= 2 ? on - 2 : -1;
    return ZG_OK;
}
int zg_profile_reset(zg_ctx *ctx) {
    int rc = 0;
    zg_ctx *c = resolve(ctx, &rc);
    if (!c) return rc ? rc : ZG_ERR_NO_DEVICE;
    std::lock_guard<std::mutex> lk(c->mu);
    (void)hipStreamSynchronize(c->stream);
    prof_collect(c);
    for (int i = 0; i < K_COUNT; i++) {
        c->prof_ms[i] = 0;
        c->prof_n[i] = 0;
    }
    return ZG_OK;
}
int zg_profile_count(zg_ctx *) { return K_COUNT; }
int zg_profile_get(zg_ctx *ctx, int index, const char **name, double *ms, uint64_t *launches) {
    int rc = 0;
    zg_ctx *c = resolve(ctx, &rc);
    if (!c) return rc ? rc : ZG_ERR_NO_DEVICE;
    if (index < 0 || index >= K_COUNT) return ZG_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    if (name) *name = kKernelNames[index];
    if (ms) *ms = c->prof_ms[index];
    if (launches) *launches = c->prof_n[index];
    return ZG_OK;
}

}  // extern "C"
