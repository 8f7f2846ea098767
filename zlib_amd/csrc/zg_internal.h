// Internal declarations shared by the HIP translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/zlib_amd.h"

namespace zg {

constexpr uint32_t kMaxWindow = 32768;  // MAX_WINDOW_SIZE, src/lz77.rs:20
constexpr uint32_t kMaxMatch = 258;     // largest max_match_length the device path accepts
constexpr uint32_t kNumLL = 286;        // literal/length alphabet, src/huffman.rs:785-789
constexpr uint32_t kNumD = 30;          // distance alphabet, src/huffman.rs:808-810
constexpr uint32_t kDOff = 288;         // offset of the distance part inside per-chunk tables
constexpr uint32_t kTabStride = 320;    // u32 words per chunk in freq / code tables
constexpr uint32_t kHdrWords = 160;     // u32 words reserved per chunk for the block header bits
constexpr uint32_t kStoredBlock = 16384;  // SIXTEEN_KB, src/compress.rs:14

// index kernel geometry
constexpr uint32_t kIdxHashBits = 12;
constexpr uint32_t kIdxBuckets = 1u << kIdxHashBits;
constexpr uint32_t kIdxSegment = 256 * 1024;  // positions owned by one index workgroup

// match kernel geometry
constexpr uint32_t kMatchTile = 16384;  // positions resolved by one match workgroup
constexpr uint32_t kMatchThreads = 1024;
constexpr uint32_t kMatchRecordTile = 4096;  // tile of the small geometry used for isolated chunks up to this size

// walk kernel: value of match[] at positions no walker stood on (length 0, an impossible distance), and the
// words of the bit set in which a run of tiles leaves the positions reached beyond its last tile
constexpr uint32_t kUnvisited = 0x0000FFFFu;
constexpr uint32_t kWalkExitWords = 10;

// parse / emit kernel geometry
constexpr uint32_t kParseSub = 512;   // positions per speculative parse sub-tile (> kMaxMatch)
constexpr uint32_t kEmitTile = 4096;   // positions per unit-emission workgroup
constexpr uint32_t kEmitThreads = 256;
inline uint32_t parse_subs_per_chunk(uint64_t chunk_bytes) {
    return (uint32_t)((chunk_bytes + kParseSub - 1) / kParseSub);
}
// bytes of the parse's flags (a bit per position, whole sub-tiles) for n_chunks chunks
inline uint64_t parse_flag_bytes(uint64_t n_chunks, uint64_t chunk_bytes) {
    return n_chunks * parse_subs_per_chunk(chunk_bytes) * (kParseSub / 8);
}
inline uint32_t emit_tiles_per_chunk(uint64_t chunk_bytes) {
    return (uint32_t)((chunk_bytes + kEmitTile - 1) / kEmitTile);
}

// encode kernel geometry
constexpr uint32_t kEncTile = 2048;  // tokens per encode workgroup
constexpr uint32_t kEncThreads = 256;

struct MatchParams {
    uint32_t window;   // LZ77Compressor::window_size
    uint32_t min_len;  // LZ77Compressor::min_match_length
    uint32_t max_len;  // LZ77Compressor::max_match_length
};

// A view of device-resident input: `data` points at view position 0.
// Positions are view-relative.  Chunk c covers
// [first + c*chunk_bytes, min(end, first + (c+1)*chunk_bytes)).
struct ChunkGrid {
    const uint8_t *data;
    uint64_t first;        // view position of the first owned byte (= halo length)
    uint64_t end;          // view position one past the last owned byte
    uint64_t chunk_bytes;
    uint32_t n_chunks;     // >= 1 (an empty input has one empty chunk)
    // record batches: chunk c holds chunk_len[c] (<= chunk_bytes) bytes at its slot when chunk_len is set, and
    // with `isolated` no chunk sees the input before its own first byte (every chunk is a stream of its own)
    const uint32_t *chunk_len = nullptr;
    uint32_t isolated = 0;
};

// positions one match workgroup resolves, and workgroups per chunk
inline uint32_t match_tile_positions(const ChunkGrid &g) {
    return (g.isolated && g.chunk_bytes <= kMatchRecordTile) ? kMatchRecordTile : kMatchTile;
}
inline uint32_t match_tiles_per_chunk(const ChunkGrid &g) {
    const uint32_t t = match_tile_positions(g);
    return (uint32_t)((g.chunk_bytes + t - 1) / t);
}

#if defined(__HIPCC__)
__device__ __forceinline__ uint64_t chunk_end(uint64_t cs, uint64_t chunk_bytes, uint64_t end, const uint32_t *clen,
                                              uint32_t chunk) {
    if (clen) return cs + clen[chunk];
    return (cs + chunk_bytes < end) ? cs + chunk_bytes : end;
}
#endif

// launchers (each enqueues on `s` and returns the HIP status of the launch)
// index of positions [r0, r1) by the hash of their first key_bytes (3 or 5) bytes; far[] for [far_lo, r1) only.
// Members below r0 are not seen: r0 must lie index_lookback(window) or more behind far_lo, or be the first
// position that can matter (the start of the data; the first byte of an isolated slot).  The arrays start at
// view position `base` <= r0.  Every next[] of [r0, r1) and far[] of [far_lo, r1) is written.
// iso_bytes != 0: positions belong to slots of iso_bytes (a multiple of 1024) starting at iso_first, chains
// never leave a slot, and r0 is the first byte of a slot.
// One launch builds the index for every key length of the set (3, 4 or 5 bytes).
struct IndexSet {
    uint32_t n = 0;
    int key[3] = {0, 0, 0};
    uint16_t *far[3] = {nullptr, nullptr, nullptr};
    uint16_t *next[3] = {nullptr, nullptr, nullptr};
    void add(int k, uint16_t *f, uint16_t *nx) {
        key[n] = k;
        far[n] = f;
        next[n] = nx;
        n++;
    }
};
hipError_t launch_index(const uint8_t *data, uint64_t n_view, uint64_t r0, uint64_t r1, uint64_t far_lo,
                        uint32_t window, const IndexSet &set, uint64_t base, uint32_t *status, bool safe,
                        uint64_t iso_first, uint32_t iso_bytes, hipStream_t s);
uint64_t index_lookback(uint32_t window);
// level 1: lengths >= rec_min along the five-byte chains.  With open_list it also lists, per tile, the positions
// it leaves without a length (tile position | distance of the candidate that ended the search << 16; a tile's
// list starts at tile index * match_tile_positions(g), its length goes to open_cnt[tile index]).
// walk_chunk (may be null): chunks c0 + i with walk_chunk[i] != 0 are left alone (k_walk searches them)
hipError_t launch_match(const ChunkGrid &g, uint32_t c0, uint32_t c1, MatchParams mp, uint32_t rec_min,
                        const uint16_t *far, const uint16_t *next, uint64_t base, uint32_t *match,
                        uint32_t *open_list, uint32_t *open_cnt, const uint8_t *walk_chunk, hipStream_t s);
// level 2: lengths 3 and 4 for the listed positions, from the four-byte and the three-byte index
hipError_t launch_match_lo(const ChunkGrid &g, uint32_t c0, uint32_t c1, MatchParams mp, const uint16_t *far4,
                           const uint16_t *next4, const uint16_t *far3, const uint16_t *next3, uint64_t base,
                           uint32_t *match, uint32_t *open_list, uint32_t *open_far, const uint32_t *open_cnt,
                           hipStream_t s);
// the search along the greedy parse only (zg_walk.hip): match[] of every position of chunks [c0, c1) of a grid
// with the general tile geometry -- a result where a walker stood, kUnvisited elsewhere.  exit_bits: kWalkExitWords
// words per run of tiles_per_run tiles.
uint32_t walk_tiles_per_run(const ChunkGrid &g, uint32_t n_chunks, uint32_t cus);
// far4 / next4: the four-byte index; far3 / next3: the three-byte index, read only when mp.min_len == 3.
// walk_chunk (may be null: every chunk): only chunks c0 + i with walk_chunk[i] != 0 are searched.
hipError_t launch_walk(const ChunkGrid &g, uint32_t c0, uint32_t c1, MatchParams mp, const uint16_t *far4,
                       const uint16_t *next4, const uint16_t *far3, const uint16_t *next3, uint64_t base,
                       uint32_t *match, uint32_t *exit_bits, uint32_t tiles_per_run, const uint8_t *walk_chunk,
                       hipStream_t s);
// Which way is cheaper for a chunk?  The full search pays for every member of every five-byte chain, the walk
// for the positions of the parse only: walk_chunk[i] = 1 for chunks whose five-byte chains are long (the mean
// length over 256 sampled positions, extrapolated from each chain's first eight members, exceeds a threshold)
// -- structured data, where the positions inside long matches sit on dense chains and the parse never visits
// them.
hipError_t launch_chunk_probe(const ChunkGrid &g, uint32_t c0, uint32_t c1, const uint16_t *far5,
                              const uint16_t *next5, uint64_t base, uint8_t *walk_chunk, hipStream_t s);
// status: bit 2 is raised when the parse stands on a position whose match[] is kUnvisited
hipError_t launch_parse(const ChunkGrid &g, uint32_t c0, uint32_t c1, const uint32_t *match, uint64_t base,
                        uint32_t *flag, uint32_t *sub_exit, uint32_t *sub_exit2, uint8_t *merged,
                        uint32_t *tile_cnt, uint32_t *tokens, uint64_t tok_base, uint32_t *ntok, uint32_t *freq,
                        uint32_t *status, hipStream_t s);
hipError_t launch_histogram(const uint32_t *tokens, uint64_t n, uint32_t *freq, hipStream_t s);
constexpr int64_t kAllFinal = -2;  // launch_huffman: every chunk's block is the last of its stream
hipError_t launch_huffman(uint32_t n_chunks, int strategy, const uint32_t *freq, int64_t final_chunk, uint32_t *tab,
                          uint32_t *hdr, uint32_t *hdr_bits, uint64_t *block_bits, hipStream_t s);
hipError_t launch_tile_bits(uint32_t n_chunks, uint32_t max_tiles, const uint32_t *tokens, const ChunkGrid &g,
                            uint32_t c0, uint64_t tok_base, const uint32_t *ntok, const uint32_t *tab,
                            uint32_t *tile_bits, hipStream_t s);
hipError_t launch_tile_scan(uint32_t n_chunks, uint32_t max_tiles, const uint32_t *ntok, uint32_t *tile_bits,
                            hipStream_t s);
hipError_t launch_offsets(uint32_t n_chunks, uint32_t max_tiles, const uint32_t *ntok, uint32_t *tile_bits,
                          const uint64_t *block_bits, uint64_t *bit_off, uint64_t *carry, hipStream_t s);
hipError_t launch_zero_seams(uint32_t n_chunks, uint32_t max_tiles, const uint32_t *ntok, const uint32_t *hdr_bits,
                             const uint32_t *tile_off, const uint64_t *bit_off, const uint64_t *block_bits,
                             uint32_t *out, uint64_t out_cap_bits, bool separate, hipStream_t s);
hipError_t launch_encode(uint32_t n_chunks, uint32_t max_tiles, const uint32_t *tokens, const ChunkGrid &g,
                         uint32_t c0, uint64_t tok_base, const uint32_t *ntok, const uint32_t *tab,
                         const uint32_t *hdr, const uint32_t *hdr_bits, const uint32_t *tile_off,
                         const uint64_t *bit_off, uint32_t *out, uint64_t out_cap_bits, uint32_t *status,
                         hipStream_t s);
// per segment (a, b) of a stream starting at (1, 0); seg_len, when set, holds each segment's byte count
hipError_t launch_adler(const uint8_t *data, uint64_t n, uint64_t seg_bytes, uint32_t n_seg, const uint32_t *seg_len,
                        uint32_t *partial, hipStream_t s);
// record batches: byte offset of every record's stream (the batch's first in carry_in[0], the next batch's first
// to carry_out[0], which must be another word) and its block's bit offset; then the two header bytes and the
// Adler-32 around every block
hipError_t launch_batch_offsets(uint32_t n_chunks, const uint64_t *block_bits, uint64_t *bit_off,
                                const uint64_t *carry_in, uint64_t *carry_out, uint64_t *out_offsets, hipStream_t s);
hipError_t launch_batch_frame(uint32_t n_chunks, const uint64_t *block_bits, const uint64_t *out_offsets,
                              const uint32_t *adler_part, uint8_t *out, hipStream_t s);
hipError_t launch_adler_fold(const uint32_t *partial, uint64_t n, uint64_t seg_bytes, uint32_t n_seg,
                             uint64_t *block_part /* 3 words per 1024 segments */, uint32_t *state /* [a, b] running */,
                             hipStream_t s);
// the same fold from the empty stream, the finished checksum (b << 16 | a) stored at `result`
hipError_t launch_adler_result(const uint32_t *partial, uint64_t n, uint64_t seg_bytes, uint32_t n_seg,
                               uint64_t *block_part, uint32_t *result, hipStream_t s);
hipError_t launch_stored(const uint8_t *data, uint64_t n, uint8_t *out, uint32_t *partial, hipStream_t s);
// inflate: one lane per stream; stream i is in[in_off[i] .. in_off[i+1]), its output slot out[out_off[i] ..
// out_off[i+1]); size_only counts the output bytes without writing them (and without checking the Adler-32);
// large selects the wave-per-stream kernel (few, long streams)
enum InflateStatus : uint32_t {
    INF_OK = 0,
    INF_BAD_METHOD,    // CM != 8
    INF_BAD_WINDOW,    // CINFO > 7
    INF_BAD_FCHECK,    // CMF/FLG checksum
    INF_PRESET_DICT,   // FDICT set
    INF_BAD_BLOCK,     // block type 3
    INF_BAD_CODES,     // code lengths that describe no prefix code, or a malformed code description
    INF_BAD_SYMBOL,    // bits that match no code, or a symbol outside the alphabet
    INF_BAD_DISTANCE,  // a match reaching before the start of the output
    INF_TRUNCATED,     // the input ends inside the stream
    INF_BAD_CHECKSUM,  // Adler-32 mismatch
    INF_OVERFLOW,      // the output slot is too small
};
hipError_t launch_inflate(const uint8_t *in, const uint64_t *in_off, uint32_t n_streams, uint8_t *out,
                          const uint64_t *out_off, uint64_t *out_len, uint32_t *status, bool size_only, bool large,
                          hipStream_t s);
// One stream with known block boundaries, all blocks at once: bit_off[i] = bit offset of block i in `in`, block i
// inflates to output bytes [i * chunk, (i + 1) * chunk) (the last to what is left of out_total); chunk >= 32768.
// units: 2 bytes per output byte of scratch, win: 32 KiB per block; status[i] = InflateStatus of block i.
uint64_t inflate_tok_stride(uint64_t chunk);
hipError_t launch_inflate_blocks(const uint8_t *in, uint64_t in_bytes, const uint64_t *bit_off, uint32_t n_blocks,
                                 uint64_t chunk, uint64_t out_total, uint32_t pass_blocks, uint32_t *tok, uint32_t *ntok,
                                 uint16_t *units, uint8_t *win, uint8_t *out, uint32_t *status, hipStream_t s,
                                 hipStream_t s2, hipEvent_t *ev);
hipError_t launch_trailer(uint8_t *out, const uint64_t *total_bits, const uint32_t *adler_state,
                          uint32_t adler_override, int use_override, int write_header, int write_adler,
                          uint64_t *out_bytes, hipStream_t s);

}  // namespace zg
