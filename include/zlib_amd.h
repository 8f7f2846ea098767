/*
 * zlib_amd.h -- C ABI of the MI355X-native DEFLATE/zlib compressor.
 *
 * This is the boundary a binding of the mkpro118/zlib crate would call
 * (Rust `extern "C"` block, Python ctypes, ...).  Plain pointers and sizes
 * only.  Every compute entry point runs hand-written HIP kernels for gfx950;
 * there is no CPU fallback: without a usable GPU the calls return
 * ZG_ERR_NO_DEVICE.
 *
 * Conventions: 0 = success, negative = error (zg_strerror); the library
 * allocates host outputs with malloc and the caller releases them with
 * zg_free; nothing unwinds or aborts across the boundary.  A zg_ctx owns its
 * HIP streams (the one its kernels run on and a second for uploads, the
 * second half of overlapped passes and the checksums of record batches) and
 * its scratch buffers on one device; device inputs must be complete when a
 * call is made (these streams are not ordered after the caller's).
 * Calls on the same ctx are
 * serialised by an internal mutex, different ctx objects are independent.
 * Passing ctx == NULL uses a lazily created process-wide context on the
 * current HIP device.
 *
 * Reference items each entry point stands in for are cited as file:line of
 * the reference crate.
 */
#ifndef ZLIB_AMD_H
#define ZLIB_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ---- */
enum {
    ZG_OK = 0,
    ZG_ERR_ARG = -1,         /* bad argument                                         */
    ZG_ERR_NO_DEVICE = -2,   /* no HIP device / HIP runtime failure at init           */
    ZG_ERR_HIP = -3,         /* a HIP call failed (zg_last_error has the text)        */
    ZG_ERR_NOMEM = -4,       /* host or device allocation failed                      */
    ZG_ERR_CAPACITY = -5,    /* caller-provided output buffer too small               */
    ZG_ERR_UNSUPPORTED = -6, /* parameters outside what the device path implements    */
    ZG_ERR_INTERNAL = -7,    /* a device-side consistency check failed                */
    ZG_ERR_DATA = -8,        /* decompress: invalid or corrupt zlib stream            */
    ZG_ERR_IO = -9           /* stream / file front end: a read or write failed       */
};

/* Strategy: src/compress.rs:28-34 (same order). */
enum { ZG_AUTO = 0, ZG_DYNAMIC = 1, ZG_FIXED = 2, ZG_RAW = 3 };

/* Packed LZ77 unit (src/lz77.rs:23-27): Literal(b) = b (0..255);
 * Marker(len, dist) = 0x80000000 | len << 16 | dist. */
#define ZG_TOK_MARKER 0x80000000u
#define ZG_TOK_IS_MARKER(t) (((t) & ZG_TOK_MARKER) != 0)
#define ZG_TOK_LEN(t) (((t) >> 16) & 0x1FFu)
#define ZG_TOK_DIST(t) ((t) & 0xFFFFu)

typedef struct zg_ctx zg_ctx;

/* Options; zero-initialise for defaults. */
typedef struct zg_opts {
    size_t chunk_bytes; /* input bytes per DEFLATE block; 0 = 1 MiB; multiple of 1024, <= 256 MiB */
    size_t batch_bytes; /* input bytes processed per pass over the scratch buffers; 0 = 256 MiB */
} zg_opts;

/* ---- context ---- */
int zg_init(int device, zg_ctx **ctx);
void zg_destroy(zg_ctx *ctx);
const char *zg_strerror(int code);
const char *zg_last_error(zg_ctx *ctx); /* text of the last HIP failure on this ctx */
void zg_free(void *p);

/* ---- zlib::compress(&[u8], &Strategy) -> Vec<u8>   (src/compress.rs:51-86) ----
 * Host buffers in and out.  The stream is 78 01, one DEFLATE block per chunk
 * (each matched against the preceding 32 KiB of input), BFINAL on the last,
 * Adler-32 big-endian.  With n <= chunk_bytes it is the reference's stream
 * byte for byte wherever the reference does not panic, with one exception:
 * ZG_RAW on more than 16 KiB.  The reference writes LEN = data.len() as u16
 * into every stored block (src/compress.rs:183), which no inflater accepts;
 * this library writes each block's own length.  Empty input with ZG_RAW
 * (the reference underflows, src/compress.rs:173) gives one empty final
 * stored block.  chunk_bytes above 256 MiB: ZG_ERR_UNSUPPORTED. */
int zg_compress(zg_ctx *ctx, const uint8_t *in, size_t n, int strategy, const zg_opts *opts,
                uint8_t **out, size_t *out_n);

/* Same with device-resident input and output (d_in, d_out are device pointers
 * on the ctx's device; d_out needs zg_compress_bound bytes, 4-byte aligned).
 * Runs on the ctx stream and synchronises it before returning *out_n. */
size_t zg_compress_bound(size_t n, int strategy, const zg_opts *opts);
int zg_compress_device(zg_ctx *ctx, const void *d_in, size_t n, int strategy, const zg_opts *opts,
                       void *d_out, size_t out_cap, size_t *out_n);

/* The same call spread over several GPUs of one process: zlib::compress
 * (src/compress.rs:51-86) for a caller that owns more than one device.  The
 * input is cut into contiguous runs of whole chunks, one per entry of `devices`
 * (HIP device indices; an index may appear more than once), every device
 * compresses its run on a context the library keeps, and the blocks are
 * assembled into ONE stream, byte for byte the stream zg_compress returns for
 * the same input and options.  Host buffers in and out; *out is malloc'ed
 * (zg_free).  Calls are serialised by a lock of their own. */
int zg_compress_multi(const int *devices, int n_devices, const uint8_t *in, size_t n, int strategy,
                      const zg_opts *opts, uint8_t **out, size_t *out_n);

/* The same call for input that is not in memory as a whole (a file, a pipe, a
 * socket): zlib::compress (src/compress.rs:51-86) takes a slice, so a caller of
 * the reference has to load everything first; here the input is pulled through
 * read_fn segment by segment (a whole number of chunks, batch_bytes by default)
 * and the stream is pushed through write_fn as it is finished, while the next
 * segment is being read and uploaded and the previous one downloaded.  The
 * bytes handed to write_fn, in order, are byte for byte what zg_compress
 * returns for the whole input with the same strategy and options.
 *   read_fn(user, buf, cap): put up to cap bytes into buf and return their
 *     number; 0 = end of input; (size_t)-1 = failure (ZG_ERR_IO).
 *   write_fn(user, buf, n): take n bytes; non-zero = failure (ZG_ERR_IO).
 * Both are called from threads of the library, never at the same time as
 * themselves.  The context (the default one for ctx == NULL) is locked for
 * the whole call: a callback that calls any zg_* function on the same
 * context waits for ever -- give such a callback a context of its own.  size_hint: the input's length if known (only sizes the
 * buffers; 0 = unknown).  *in_total / *out_total (may be NULL) receive the byte
 * counts.  With chunk_bytes not a multiple of 16 KiB a segment is
 * chunk_bytes * 16 KiB / gcd of the two; above 1 GiB: ZG_ERR_UNSUPPORTED.
 * zg_compress_file does the same between two paths; the two must not name the
 * same file (ZG_ERR_ARG), and a call that fails removes what it has written. */
typedef size_t (*zg_read_fn)(void *user, uint8_t *buf, size_t cap);
typedef int (*zg_write_fn)(void *user, const uint8_t *buf, size_t n);
int zg_compress_stream(zg_ctx *ctx, zg_read_fn read_fn, void *read_user, zg_write_fn write_fn,
                       void *write_user, int strategy, const zg_opts *opts, uint64_t size_hint,
                       uint64_t *in_total, uint64_t *out_total);
int zg_compress_file(zg_ctx *ctx, const char *in_path, const char *out_path, int strategy,
                     const zg_opts *opts, uint64_t *in_bytes, uint64_t *out_bytes);

/* Where the blocks of the last zg_compress, zg_compress_device (Dynamic / Fixed)
 * or zg_shard_begin call on this ctx start: bit_offsets[i] receives the bit
 * offset of chunk i's block in that call's output (zg_shard_begin: counted from
 * the rank's first block).  Returns the number of blocks; at most cap entries
 * are written.  No counterpart in the reference, whose streams have one block;
 * a reader that wants to inflate the blocks of a stream in parallel needs
 * these (they are not part of the stream), and the tests use them to compare
 * single blocks of very large streams with the oracle's. */
size_t zg_last_block_offsets(zg_ctx *ctx, uint64_t *bit_offsets, size_t cap);

/* ---- many records, each a zlib stream of its own ----
 * compress(record, strategy) of the reference (src/compress.rs:51-86) for every
 * record of a batch at once: record i is in[offsets[i] .. offsets[i+1]), its
 * stream is out[out_offsets[i] .. out_offsets[i+1]) and is what zg_compress
 * returns for that record alone whenever the record fits one chunk.  Strategy
 * ZG_DYNAMIC or ZG_FIXED (others: ZG_ERR_UNSUPPORTED).  *out and *out_offsets
 * (n_records + 1 entries) are malloc'ed; release with zg_free. */
int zg_compress_batch(zg_ctx *ctx, const uint8_t *in, const size_t *offsets, size_t n_records, int strategy,
                      uint8_t **out, size_t **out_offsets);

/* Same on device memory: record i occupies d_len[i] bytes (d_len == NULL: all
 * `stride` bytes) at d_in + i * stride; stride is a multiple of 1024, at most
 * 256 MiB (ZG_ERR_UNSUPPORTED beyond).  d_out
 * needs zg_compress_batch_bound bytes, 4-byte aligned; d_out_offsets receives
 * n_records + 1 byte offsets; *out_bytes the total. */
size_t zg_compress_batch_bound(size_t stride, size_t n_records, int strategy);
int zg_compress_batch_device(zg_ctx *ctx, const void *d_in, size_t stride, const uint32_t *d_len,
                             size_t n_records, int strategy, void *d_out, size_t out_cap,
                             uint64_t *d_out_offsets, size_t *out_bytes);

/* ---- zlib::decompress(&[u8]) -> Result<Vec<u8>, String>   (src/decompress.rs:34-88) ----
 * Inflate on the device.  Err(String) of the reference (bad CM / CINFO / FCHECK /
 * FDICT, block type 3, Adler-32 mismatch) is ZG_ERR_DATA with the reference's
 * message in zg_last_error; corrupt bodies, where the reference panics, likewise.
 * One stream is decoded by one lane (DEFLATE decoding is serial), so a single
 * large stream is slow; the batch forms decode all streams in parallel.
 * *out (and *out_offsets, n_streams + 1 entries) are malloc'ed: zg_free. */
int zg_decompress(zg_ctx *ctx, const uint8_t *in, size_t n, uint8_t **out, size_t *out_n);
int zg_decompress_batch(zg_ctx *ctx, const uint8_t *in, const size_t *offsets, size_t n_streams,
                        uint8_t **out, size_t **out_offsets);
/* Device memory: stream i is d_in[d_in_offsets[i] .. d_in_offsets[i+1]); its output goes
 * to the slot d_out[d_out_offsets[i] .. d_out_offsets[i+1]); d_out_len[i] receives the
 * number of bytes, d_status[i] 0 or a non-zero reason (11 = slot too small).  With
 * d_out == NULL only the lengths are computed (no Adler-32 check).  The call itself
 * returns ZG_OK unless the launch failed: look at d_status. */
int zg_decompress_batch_device(zg_ctx *ctx, const void *d_in, const uint64_t *d_in_offsets, size_t n_streams,
                               void *d_out, const uint64_t *d_out_offsets, uint64_t *d_out_len,
                               uint32_t *d_status);

/* zlib::decompress (src/decompress.rs:34-88; its loop over blocks :97-118) for ONE
 * large stream of this library, with all blocks decoded at once.  A DEFLATE
 * stream does not say where its blocks begin, so a decoder normally goes
 * through them one after the other; the caller passes what zg_last_block_offsets
 * reported after compressing: bit_offsets[i] (host memory) = bit offset of block
 * i, n_blocks = ceil(n / chunk_bytes) (1 for n = 0), n = the length of the
 * uncompressed data (block i inflates to its bytes [i * chunk_bytes, ...)).
 * chunk_bytes must be 32768 or more (ZG_ERR_UNSUPPORTED otherwise) and the blocks
 * Dynamic or Fixed.  d_stream / d_out are device pointers; d_out needs n bytes.
 * A stream that does not fit the description, or is corrupt, or fails its
 * Adler-32: ZG_ERR_DATA (zg_last_error says what and in which block).  Scratch:
 * 6 bytes per byte of output for at most 8 GiB of output at a time (tokens and
 * 16-bit units of the blocks in flight: two passes of 4 GiB), and 32 KiB per
 * block of the stream. */
int zg_decompress_blocks_device(zg_ctx *ctx, const void *d_stream, size_t stream_bytes, const uint64_t *bit_offsets,
                                size_t n_blocks, size_t chunk_bytes, void *d_out, size_t n);

/* ---- sharded compress: one rank of a multi-GPU job ----
 * A rank owns `n` input bytes (a whole number of chunks unless it is the last
 * rank) and sees `halo` (<= 32768) bytes of the input that precede them:
 * d_in points at the first halo byte.  zg_shard_begin compresses the rank's
 * chunks into blocks packed from bit 0 of a buffer the context keeps, and
 * reports the rank's totals; after the ranks have exchanged them,
 * zg_shard_finish moves the blocks to bit phase (base_bit % 8) in d_out.  Byte 0 of d_out is byte base_bit/8 of the final
 * stream; when the phase is non-zero it holds only this rank's bits and must
 * be OR-ed with the previous rank's last byte.  first_rank writes the 78 01
 * header in front of its blocks (base_bit must then be 16, the first bit after
 * the header; byte 0 of d_out is then byte 0 of the stream), last_rank sets BFINAL on its last block
 * and appends the Adler-32 passed in `adler` (combined by the caller with
 * zg_adler32_combine).  A rank with n == 0 adds no block (total_bits 0) -- unless
 * it is the last rank: the stream of an empty input is one empty final block. */
typedef struct zg_shard_info {
    uint64_t total_bits; /* bits of all blocks of this rank (no zlib header/trailer) */
    uint32_t adler;      /* Adler-32 of this rank's n bytes taken alone             */
    uint32_t n_chunks;
} zg_shard_info;
int zg_shard_begin(zg_ctx *ctx, const void *d_in, size_t halo, size_t n, int strategy,
                   const zg_opts *opts, int last_rank, zg_shard_info *info);
int zg_shard_finish(zg_ctx *ctx, uint64_t base_bit, int first_rank, int last_rank, uint32_t adler,
                    void *d_out, size_t out_cap, size_t *out_bytes);
uint32_t zg_adler32_combine(uint32_t adler1, uint32_t adler2, uint64_t len2);
/* The same in pieces, for an exchange that runs beside the kernels.  zg_shard_open
 * starts compressing the rank's chunks, one pass over the scratch buffers
 * (batch_bytes of input) at a time, into d_packed (a device buffer of the
 * CALLER's, 4-byte aligned, zg_compress_bound(n) + 8 bytes), from its bit 0
 * on.  Every zg_shard_next call queues the kernels of the next pass, waits for
 * the oldest unreported one and says which bytes of d_packed have become final
 * with it -- bytes [byte_offset, byte_offset + n_bytes); the byte a pass ends in
 * the middle of belongs to the next piece -- so that the caller can ship them
 * while the GPU works on.  The piece with last != 0 ends the run and fills *info.
 * The rank that assembles the stream merges every rank's bytes, in stream
 * order, at the bit offset the totals give it (zg_shard_merge; first_run writes
 * 78 01 in front, base_bit 16) and closes with zg_shard_seal (padding to a
 * byte, Adler-32).  d_src / d_dst are 4-byte aligned device buffers; d_src must
 * be readable up to the end of the word its last bit lies in, d_dst needs 8
 * bytes beyond the last bit.  Merges on one ctx run in the order of the calls.
 * No counterpart in the reference (src/compress.rs:51-86 is one call on one
 * slice). */
typedef struct zg_shard_piece {
    uint64_t byte_offset; /* first byte of the piece, counted from d_packed             */
    uint64_t n_bytes;     /* bytes of d_packed that are final from there on             */
    uint64_t bits_so_far; /* bits of all blocks finished so far                        */
    int last;             /* non-zero: the rank's last piece                            */
} zg_shard_piece;
int zg_shard_open(zg_ctx *ctx, const void *d_in, size_t halo, size_t n, int strategy, const zg_opts *opts,
                  int last_rank, void *d_packed, size_t packed_cap);
int zg_shard_next(zg_ctx *ctx, zg_shard_piece *piece, zg_shard_info *info);
int zg_shard_merge(zg_ctx *ctx, const void *d_src, uint64_t n_bits, void *d_dst, size_t dst_cap, uint64_t base_bit,
                   int first_run);
int zg_shard_seal(zg_ctx *ctx, void *d_dst, size_t dst_cap, uint64_t end_bit, uint32_t adler, size_t *out_bytes);


/* ---- zlib::adler::adler32(&[u8]) -> u32   (src/adler.rs:5-13) ---- */
int zg_adler32(zg_ctx *ctx, const uint8_t *in, size_t n, uint32_t *out);
int zg_adler32_device(zg_ctx *ctx, const void *d_in, size_t n, uint32_t *out);

/* Adler-32 of device memory without any allocation.  No counterpart in the
 * reference: it is for callers that cannot allocate device memory while they
 * run (tight serving loops, harnesses that forbid hipMalloc).  The
 * per-segment sums live in a fixed array inside the library's code object
 * (128 KiB, held by every GPU the library runs on until the process ends)
 * instead of a buffer of the context, so n is limited to
 * ZG_ADLER32_NOALLOC_MAX bytes (ZG_ERR_CAPACITY beyond).  ctx must be a context
 * from zg_init (NULL: ZG_ERR_ARG -- the default context would have to be
 * created); d_out is a DEVICE pointer to one uint32_t on the ctx's device and
 * holds the checksum when the call returns.  The kernels run on the ctx's own
 * stream, which the call synchronises before it returns: it is NOT ordered
 * after work the caller has queued on other streams and cannot be recorded
 * into a HIP graph.  All calls on one GPU share that GPU's array; the library
 * serialises them with a lock per device, so calls on two contexts of one
 * device from two threads take turns (processes have a copy each). */
#define ZG_ADLER32_NOALLOC_MAX ((uint64_t)4 << 30)
int zg_adler32_device_noalloc(zg_ctx *ctx, const void *d_in, size_t n, uint32_t *d_out);

/* ---- LZ77Compressor::compress(&self, &[u8]) -> Vec<LZ77Unit>   (src/lz77.rs:143-207) ----
 * window/min_len/max_len are the public fields of LZ77Compressor
 * (src/lz77.rs:31-38; defaults 144/5/100, zlib 32768/3/258).  Positions
 * before `start` are window only (start = 0 is the reference call).  The
 * device path needs 3 <= min_len and max_len <= 258 (ZG_ERR_UNSUPPORTED
 * otherwise). */
int zg_lz77(zg_ctx *ctx, const uint8_t *in, size_t n, size_t start, uint32_t window, uint32_t min_len,
            uint32_t max_len, uint32_t **tokens, size_t *n_tokens);

/* ---- HuffmanTree::from_lz77 + assign + to_canonical   (src/huffman.rs:238-288,
 * 619-645, 714-734; src/compress.rs:296-314) ----
 * Code lengths and LSB-first (bit-reversed) codes of the two canonical trees
 * for a token stream; unused symbols have length 0. */
int zg_huffman_from_tokens(zg_ctx *ctx, const uint32_t *tokens, size_t n, uint8_t ll_len[286],
                           uint16_t ll_code[286], uint8_t d_len[30], uint16_t d_code[30]);

/* ---- measurement ----
 * With profiling on, every kernel launch is bracketed by HIP events on the
 * stream it goes to; zg_profile_get returns accumulated milliseconds and launch
 * counts per kernel since the last zg_profile_reset.  on = 0: off; 1: every
 * kernel; 2 + i: only the kernel that zg_profile_get reports at index i (the
 * events of the others cost a long call about one per cent). */
int zg_profile_enable(zg_ctx *ctx, int on);
int zg_profile_reset(zg_ctx *ctx);
int zg_profile_count(zg_ctx *ctx);
int zg_profile_get(zg_ctx *ctx, int index, const char **name, double *ms, uint64_t *launches);

/* ---- synthetic corpora for tests and benchmarks (host side) ----
 * kind: 0 text, 1 random, 2 structured binary, 3 mixed (text/random/text/binary
 * per MiB), 4 records.  Deterministic in (kind, seed, offset): buf receives
 * bytes [offset, offset+n) of the corpus. */
int zg_corpus_fill(int kind, uint64_t seed, uint64_t offset, uint8_t *buf, size_t n);

#ifdef __cplusplus
}
#endif
#endif
