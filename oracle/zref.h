/*
 * zref.h -- C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is a CPU restatement of the
 * algorithms of the mkpro118/zlib crate (reference paths are cited per
 * function in zref.cpp).  It exists to check the HIP product path; nothing in
 * the product (zlib_amd/) links, loads or calls it.  Only tests/, the smoke
 * check and the cpu_baseline leg of bench.py may use it.
 *
 * Parity pinning: the reference is Rust and no Rust toolchain exists in the
 * build container, so the reference itself was never run.  The restatement is
 * pinned by every known-answer test the reference's own unit tests hold for
 * this path (ported in tests/test_oracle_kat.py) and cross-checked with
 * system zlib as an independent inflater.  LZ77 token streams have no golden
 * vectors in the reference ("token parity unpinned" beyond the verbatim loop).
 */
#ifndef ZREF_H
#define ZREF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Status codes.  Negative values below -100 name the reference panic that the
 * strict (reference-faithful) entry points would have hit. */
enum {
    ZO_OK = 0,
    ZO_ERR_ARG = -1,
    ZO_ERR_DECODE = -2,          /* decompress: Err(String) of the reference   */
    ZO_ERR_MALFORMED = -3,       /* decompress: reference would panic on body  */
    ZO_PANIC_LL_TOO_LONG = -101, /* src/compress.rs:303 */
    ZO_PANIC_D_TOO_LONG = -102,  /* src/compress.rs:307 */
    ZO_PANIC_CL_TOO_LONG = -103, /* src/compress.rs:423 */
    ZO_PANIC_NO_ENCODING = -104, /* src/compress.rs:318,331,363 */
    ZO_PANIC_HCODES = -105,      /* src/compress.rs:447 */
    ZO_PANIC_RAW_DEFECT = -106,  /* src/compress.rs:173,183 (empty / >16 KiB) */
    ZO_PANIC_ZERO_LEN_CODE = -107 /* src/huffman.rs:633 shift overflow (debug) */
};

/* Strategy, same order as src/compress.rs:28-34. */
enum { ZO_AUTO = 0, ZO_DYNAMIC = 1, ZO_FIXED = 2, ZO_RAW = 3 };

/* Packed LZ77 token: literal = byte value (0..255);
 * marker = 0x80000000 | length << 16 | distance (distance 1..32768). */
#define ZO_TOK_MARKER 0x80000000u
#define ZO_TOK_IS_MARKER(t) (((t) & ZO_TOK_MARKER) != 0)
#define ZO_TOK_LEN(t) (((t) >> 16) & 0x1FFu)
#define ZO_TOK_DIST(t) ((t) & 0xFFFFu)

void zo_free(void *p);

/* ---- bitwriter.rs ---- */
typedef struct zo_bw zo_bw;
zo_bw *zo_bw_new(void);
void zo_bw_free(zo_bw *w);
void zo_bw_write_bit(zo_bw *w, unsigned bit);
void zo_bw_write_bits(zo_bw *w, size_t value, size_t nbits);
void zo_bw_write_byte(zo_bw *w, unsigned byte);
void zo_bw_write_bytes(zo_bw *w, const uint8_t *p, size_t n);
void zo_bw_flush_byte(zo_bw *w);
/* finish: flushes and copies the buffer out (malloc'd, caller zo_free). */
int zo_bw_finish(zo_bw *w, uint8_t **out, size_t *out_n);

/* ---- bitreader.rs ---- */
typedef struct zo_br zo_br;
zo_br *zo_br_new(const uint8_t *mem, size_t n); /* borrows mem */
void zo_br_free(zo_br *r);
int zo_br_read_byte(zo_br *r);  /* -1: out of bounds (reference panics) */
int zo_br_read_bit(zo_br *r);   /* -1: out of bounds */
long long zo_br_read_bits(zo_br *r, size_t n);
long long zo_br_read_bytes(zo_br *r, size_t n);
int zo_code_to_bytes(size_t code, size_t length, uint8_t **out, size_t *out_n);

/* ---- adler.rs ---- */
uint32_t zo_adler32(const uint8_t *data, size_t n);

/* ---- huffman.rs tables and code lookups ---- */
const size_t *zo_code_length_codes_order(void); /* 19 */
const size_t *zo_length_extra_bits(void);       /* 29 */
const size_t *zo_length_base(void);             /* 29 */
const size_t *zo_distance_extra_bits(void);     /* 30 */
const size_t *zo_distance_base(void);           /* 30 */
size_t zo_get_length_code(size_t length);
size_t zo_get_distance_code(size_t distance);

/* ---- huffman.rs HuffmanTree ---- */
typedef struct zo_tree zo_tree;
/* Tie-break between two equal-weight INTERNAL nodes (unspecified in the
 * reference, src/huffman.rs:103-116): FIFO = older node pops first (the
 * build's documented choice); HEAP = leave it to the binary heap exactly as a
 * Rust std BinaryHeap would, with the leaves fed in an order shuffled by
 * `seed` (models the randomly ordered HashMap, src/huffman.rs:187-191). */
enum { ZO_TIE_FIFO = 0, ZO_TIE_HEAP = 1 };
zo_tree *zo_tree_new(void);
void zo_tree_free(zo_tree *t);
void zo_tree_insert(zo_tree *t, size_t code, size_t length, uint32_t symbol);
void zo_tree_assign(zo_tree *t);
int zo_tree_encode(const zo_tree *t, uint32_t symbol, size_t *code, size_t *len); /* 1 found */
size_t zo_tree_n_codes(const zo_tree *t);
size_t zo_tree_max_code_len(const zo_tree *t);
zo_tree *zo_tree_to_canonical(const zo_tree *t); /* NULL if not assigned (reference panics) */
zo_tree *zo_tree_from_bitlen_list(const size_t *bitlen, size_t n_bitlen, const uint32_t *alphabet,
                                  size_t n_alphabet);
zo_tree *zo_tree_from_data(const uint8_t *data, size_t n, int tie, uint64_t seed);
zo_tree *zo_tree_from_freq(const uint32_t *syms, const size_t *counts, size_t n, int tie, uint64_t seed);
void zo_tree_get_zlib_fixed(zo_tree **ltree, zo_tree **dtree);
int zo_tree_from_lz77(const uint32_t *tokens, size_t n, int tie, uint64_t seed, zo_tree **ltree,
                      zo_tree **dtree);
long zo_tree_decode(const zo_tree *t, zo_br *r); /* symbol, -1 = None, -2 = reader out of bounds */
int zo_tree_decode_trees(zo_br *r, zo_tree **ltree, zo_tree **dtree);
/* Lists the (symbol, code, len) triples of the encodings map in ascending
 * symbol order; returns the count (arrays may be NULL to query). */
size_t zo_tree_encodings(const zo_tree *t, uint32_t *syms, size_t *codes, size_t *lens, size_t cap);

/* ---- lz77.rs ---- */
/* Verbatim restatement of LZ77Compressor::compress (src/lz77.rs:143-207),
 * extended by `start`: positions < start are window only and produce no
 * tokens (start = 0 is the reference function itself). */
int zo_lz77_compress(const uint8_t *data, size_t n, size_t start, size_t window, size_t min_len,
                     size_t max_len, uint32_t **tokens, size_t *n_tokens);
/* The verbatim loop with the reference's full prefix re-compare on every step
 * (identical result, the reference's cost model; used for CPU timing). */
int zo_lz77_compress_literal(const uint8_t *data, size_t n, size_t start, size_t window, size_t min_len,
                             size_t max_len, uint32_t **tokens, size_t *n_tokens);
/* Same result through a 3-gram index (only for min_len == 3); property-tested
 * against the verbatim loop. */
int zo_lz77_fast(const uint8_t *data, size_t n, size_t start, size_t window, size_t max_len,
                 uint32_t **tokens, size_t *n_tokens);

/* ---- compress.rs pieces ---- */
int zo_literal_writer(const zo_tree *ltree, zo_bw *w, uint32_t sym);
int zo_length_writer(const zo_tree *ltree, zo_bw *w, size_t length);
int zo_distance_writer(const zo_tree *dtree, zo_bw *w, size_t distance);
int zo_write_compressed_data(zo_bw *w, const uint32_t *tokens, size_t n, const zo_tree *ltree,
                             const zo_tree *dtree);
int zo_write_dynamic_header(zo_bw *w, const zo_tree *ltree, const zo_tree *dtree, int strict);
/* code-length vectors of get_codelengths (src/compress.rs:376-407) */
int zo_get_codelengths(const zo_tree *ltree, const zo_tree *dtree, size_t **cl1, size_t *n1,
                       size_t **cl2, size_t *n2);
/* RLE pairs: sym[i], extra[i] (extra = -1 for None). */
int zo_zlib_rle(const size_t *cl1, size_t n1, const size_t *cl2, size_t n2, size_t **sym,
                long **extra, size_t *n);
/* run_length_encode: value[i], count[i] (count = 0 encodes Once(value)). */
int zo_run_length_encode(const size_t *data, size_t n, size_t **value, size_t **count, size_t *m);
int zo_rle_to_zlib_codes(const size_t *value, const size_t *count, size_t m, size_t **sym,
                         long **extra, size_t *n);
int zo_zlib_rle_encode_zero(size_t repetitions, size_t **sym, long **extra, size_t *n);
int zo_zlib_rle_encode_nonzero(size_t num, size_t repetitions, size_t **sym, long **extra, size_t *n);
/* get_code_tree (strict: panics become status codes; otherwise the build's
 * length limiting applies). */
int zo_get_code_tree(const size_t *sym, size_t n, int strict, zo_tree **out);
int zo_get_hcodes(const zo_tree *code_tree, size_t **hcodes, size_t *n);

/* ---- whole streams ---- */
/* Reference-faithful compress (src/compress.rs:51-86): one block for the
 * whole input; returns the ZO_PANIC_* code where the reference panics or is
 * defective. */
int zo_compress_reference(const uint8_t *data, size_t n, int strategy, uint8_t **out, size_t *out_n);
/* The build's stream: one DEFLATE block per `chunk_bytes` of input, each
 * matched against the preceding 32 KiB of input, BFINAL on the last block;
 * length-limited codes / lone-distance-symbol fix where the reference would
 * panic; Raw = correct 16 KiB stored blocks.  `fast` selects zo_lz77_fast. */
int zo_compress_chunked(const uint8_t *data, size_t n, int strategy, size_t chunk_bytes, int fast,
                        uint8_t **out, size_t *out_n);
/* Per-chunk introspection used by the parity tests. */
int zo_chunk_tokens(const uint8_t *data, size_t n, size_t chunk_bytes, size_t chunk_index, int fast,
                    uint32_t **tokens, size_t *n_tokens);
/* Build-mode tables for a token stream: lengths and LSB-first (bit-reversed)
 * codes for 286 lit/len and 30 distance symbols. */
int zo_chunk_tables(const uint32_t *tokens, size_t n, uint8_t ll_len[286], uint16_t ll_code[286],
                    uint8_t d_len[30], uint16_t d_code[30]);
/* Bit length of the block the build emits for these tokens (3 header bits +
 * dynamic header + payload + EOB); strategy ZO_DYNAMIC or ZO_FIXED. */
int zo_block_bits(const uint32_t *tokens, size_t n, int strategy, uint64_t *bits);
/* The bytes of that block alone (zero padded to a byte) and its bit length. */
int zo_write_block(const uint32_t *tokens, size_t n, int strategy, int bfinal, uint8_t **out, size_t *out_n,
                   uint64_t *bits);
/* Strategy::Auto decision (src/compress.rs:89-169): returns ZO_RAW/FIXED/DYNAMIC. */
int zo_auto_choice(const uint8_t *data, size_t n);

/* ---- decompress.rs ---- */
int zo_decompress(const uint8_t *in, size_t n, uint8_t **out, size_t *out_n, char *err, size_t errcap);

/* ---- cpu baseline helper: reference-style Dynamic compress of `n_slices`
 * slices (slice i = data[off[i] .. off[i]+len[i]) with `halo[i]` bytes of
 * window before it), one std::thread per slice up to `threads`; returns the
 * total compressed bits per slice. ---- */
int zo_bench_slices(const uint8_t *data, const size_t *off, const size_t *len, const size_t *halo,
                    size_t n_slices, int threads, uint64_t *bits_out);

#ifdef __cplusplus
}
#endif
#endif
