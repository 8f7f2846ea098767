// Per-block Huffman construction and dynamic-header serialisation, written as
// straight-line single-lane code so that the same source runs on one lane of a
// wave (zg_entropy.hip) and, compiled for the host, under the unit tests.
//
// What it reproduces (reference file:line):
//   code lengths   HuffmanTree::from_freq + FreqNode ordering  src/huffman.rs:95-122,185-216
//                  (min count first; equal count: leaf before internal, lower
//                  symbol first; two equal internal nodes: older first)
//   canonical code from_bitlen_list / assign / to_canonical      src/huffman.rs:311-337,619-645,714-734
//                  (codes stored bit-reversed for the LSB-first writer)
//   header         write_dynamic_header and helpers              src/compress.rs:231-274,376-569
// plus the two policies the reference lacks (it panics there): a lone symbol
// gets a one-bit code, and codes longer than the limit are shortened by count
// rebalancing (drop one code from the limit, split the deepest shorter code),
// lengths then dealt longest-first in ascending (count, symbol) order.
#pragma once
#include <stdint.h>

#include "../../include/zlib_amd.h"

#ifndef ZG_HD
#ifdef __HIPCC__
#define ZG_HD __host__ __device__
#else
#define ZG_HD
#endif
#endif

namespace zg {

constexpr int kHuffMaxSyms = 288;

// DEFLATE length -> (code - 257), extra bit count and extra value
// (LENGTH_BASE / LENGTH_EXTRA_BITS, src/huffman.rs:30-37, get_length_code :848-853)
ZG_HD inline uint32_t length_code_index(uint32_t len) {
    if (len >= 258) return 28;
    const uint32_t t = len - 3;
    if (t < 8) return t;
    const uint32_t e = 31u - (uint32_t)__builtin_clz(t);
    return 4u * (e - 2u) + 4u + ((t >> (e - 2u)) & 3u);
}
ZG_HD inline uint32_t length_extra_bits(uint32_t idx) { return (idx < 8 || idx == 28) ? 0u : (idx >> 2) - 1u; }
ZG_HD inline uint32_t length_base(uint32_t idx) {
    if (idx == 28) return 258;
    if (idx < 8) return idx + 3;
    return 3u + ((4u + (idx & 3u)) << ((idx >> 2) - 1u));
}
// DEFLATE distance -> code, extra bit count, base
// (DISTANCE_BASE / DISTANCE_EXTRA_BITS, src/huffman.rs:38-45, get_distance_code :866-871)
ZG_HD inline uint32_t dist_code(uint32_t d) {
    if (d <= 4) return d - 1;
    const uint32_t t = d - 1;
    const uint32_t e = 31u - (uint32_t)__builtin_clz(t);
    return 2u * e + ((t >> (e - 1u)) & 1u);
}
ZG_HD inline uint32_t dist_extra_bits(uint32_t code) { return code < 4 ? 0u : (code >> 1) - 1u; }
ZG_HD inline uint32_t dist_base(uint32_t code) {
    if (code < 4) return code + 1;
    return 1u + ((2u + (code & 1u)) << ((code >> 1) - 1u));
}

ZG_HD inline uint32_t bit_reverse(uint32_t v, uint32_t nbits) {
    uint32_t r = 0;
    for (uint32_t i = 0; i < nbits; i++) r |= ((v >> i) & 1u) << (nbits - 1 - i);
    return r;
}

struct HuffWork {  // scratch for one alphabet (fits LDS: ~4.7 KiB)
    uint16_t sorted[kHuffMaxSyms];       // used symbols, ascending (count, symbol)
    uint32_t weight[2 * kHuffMaxSyms];   // node weights: leaves then internal nodes
    uint16_t parent[2 * kHuffMaxSyms];
    uint16_t num[40];                    // codes per length
};

// Code lengths (0 = unused) for `n` symbols with counts `freq`, limited to max_bits, in four steps that the
// device code also uses one by one (it runs the first and third with all lanes of a wave).

// step 1: used symbols in ascending (count, symbol) order -> w.sorted[0..m); clears len[]
ZG_HD inline int huff_sort(const uint32_t *freq, int n, uint8_t *len, HuffWork &w) {
    int m = 0;
    for (int i = 0; i < n; i++) {
        len[i] = 0;
        if (freq[i]) w.sorted[m++] = (uint16_t)i;
    }
    if (m < 2) return m;
    // heap sort by (count, symbol); symbols are already ascending so equal counts stay ordered by key
    auto less = [&](uint16_t a, uint16_t b) { return freq[a] != freq[b] ? freq[a] < freq[b] : a < b; };
    for (int start = m / 2 - 1; start >= 0; start--) {
        int root = start;
        for (;;) {
            int child = 2 * root + 1;
            if (child >= m) break;
            if (child + 1 < m && less(w.sorted[child], w.sorted[child + 1])) child++;
            if (!less(w.sorted[root], w.sorted[child])) break;
            uint16_t t = w.sorted[root];
            w.sorted[root] = w.sorted[child];
            w.sorted[child] = t;
            root = child;
        }
    }
    for (int endi = m - 1; endi > 0; endi--) {
        uint16_t t = w.sorted[0];
        w.sorted[0] = w.sorted[endi];
        w.sorted[endi] = t;
        int root = 0;
        for (;;) {
            int child = 2 * root + 1;
            if (child >= endi) break;
            if (child + 1 < endi && less(w.sorted[child], w.sorted[child + 1])) child++;
            if (!less(w.sorted[root], w.sorted[child])) break;
            uint16_t t2 = w.sorted[root];
            w.sorted[root] = w.sorted[child];
            w.sorted[child] = t2;
            root = child;
        }
    }
    return m;
}

// step 2 (m >= 2): two-queue Huffman over the sorted leaves 0..m-1; internal nodes m..2m-2 in creation order.
// The heads of both queues are kept in locals so that an iteration does not wait on memory it just wrote.
// (the weights of the sorted leaves are in w.weight[0..m))
ZG_HD inline void huff_merge_loaded(int m, HuffWork &w) {
    int li = 0, ii = m, ni = m;
    uint32_t lw = w.weight[0];  // weight of the next leaf (valid while li < m)
    uint32_t iw = 0;            // weight of the next internal node (valid while ii < ni)
    for (int k = 0; k < m - 1; k++) {
        int pick[2];
        uint32_t sum = 0;
        for (int j = 0; j < 2; j++) {
            const bool leaf = li < m && (ii >= ni || lw <= iw);  // leaf wins ties
            if (leaf) {
                pick[j] = li++;
                sum += lw;
                if (li < m) lw = w.weight[li];
            } else {
                pick[j] = ii++;
                sum += iw;
                if (ii < ni) iw = w.weight[ii];
            }
        }
        w.weight[ni] = sum;
        if (ii == ni) iw = sum;  // the queue of internal nodes was empty: the new node is its head
        w.parent[pick[0]] = (uint16_t)ni;
        w.parent[pick[1]] = (uint16_t)ni;
        ni++;
    }
}

ZG_HD inline void huff_merge(const uint32_t *freq, int m, HuffWork &w) {
    for (int i = 0; i < m; i++) w.weight[i] = freq[w.sorted[i]];
    huff_merge_loaded(m, w);
}

// step 3: depth of every node top-down (parents have larger indices), reusing weight[] for the depth
ZG_HD inline void huff_depths(int m, HuffWork &w) {
    const int root = 2 * m - 2;
    w.weight[root] = 0;
    for (int i = root - 1; i >= 0; i--) w.weight[i] = w.weight[w.parent[i]] + 1;
}

// step 4: lengths from the leaf depths in w.weight[0..m); count rebalancing when the deepest exceeds max_bits
ZG_HD inline void huff_limit(int m, int max_bits, uint8_t *len, HuffWork &w) {
    for (int i = 0; i <= max_bits; i++) w.num[i] = 0;
    for (int i = 0; i < m; i++) {
        const int d = (int)w.weight[i];
        w.num[d > max_bits ? max_bits : d]++;
    }
    uint64_t total = 0;
    for (int i = max_bits; i >= 1; i--) total += (uint64_t)w.num[i] << (max_bits - i);
    while (total != ((uint64_t)1 << max_bits)) {
        w.num[max_bits]--;
        for (int i = max_bits - 1; i > 0; i--)
            if (w.num[i]) {
                w.num[i]--;
                w.num[i + 1] += 2;
                break;
            }
        total--;
    }
    int j = 0;
    for (int bits = max_bits; bits >= 1; bits--)
        for (int c = 0; c < (int)w.num[bits]; c++) len[w.sorted[j++]] = (uint8_t)bits;
}

ZG_HD inline void huff_lengths(const uint32_t *freq, int n, int max_bits, uint8_t *len, HuffWork &w) {
    const int m = huff_sort(freq, n, len, w);
    if (m == 0) return;
    if (m == 1) {  // lone symbol: one bit (RFC 1951 3.2.7); the reference panics here
        len[w.sorted[0]] = 1;
        return;
    }
    huff_merge(freq, m, w);
    huff_depths(m, w);
    int mx = 0;
    for (int i = 0; i < m; i++) {
        const int d = (int)w.weight[i];
        len[w.sorted[i]] = (uint8_t)(d > 255 ? 255 : d);
        if (d > mx) mx = d;
    }
    if (mx > max_bits) huff_limit(m, max_bits, len, w);
}

// Canonical codes, bit-reversed (LSB-first), packed as code | len << 16.
ZG_HD inline void huff_codes(const uint8_t *len, int n, uint32_t *tab) {
    uint32_t bl_count[17] = {0}, next_code[17] = {0};
    for (int i = 0; i < n; i++) bl_count[len[i]]++;
    bl_count[0] = 0;
    uint32_t code = 0;
    for (int bits = 1; bits <= 16; bits++) {
        code = (code + bl_count[bits - 1]) << 1;
        next_code[bits] = code;
    }
    for (int i = 0; i < n; i++) {
        const uint32_t l = len[i];
        tab[i] = l ? (bit_reverse(next_code[l]++, l) | (l << 16)) : 0u;
    }
}

struct BitSink {  // LSB-first bit packer into u32 words (BitWriter, src/bitwriter.rs:66-98)
    uint32_t *words;
    uint32_t nbits;
    ZG_HD void put(uint32_t value, uint32_t n) {
        if (n == 0) return;
        const uint32_t w = nbits >> 5, sh = nbits & 31u;
        value &= (n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u));
        if (sh == 0) words[w] = 0;
        words[w] |= value << sh;
        if (sh + n > 32) words[w + 1] = value >> (32 - sh);
        else if (sh + n == 32) { /* next word starts clean on its first put */ }
        nbits += n;
    }
};

struct BlockTables {
    uint32_t ll[kHuffMaxSyms];  // code | len << 16 for literal/length symbols 0..285
    uint32_t d[32];             // same for distance symbols 0..29
};

struct HeaderWork {
    uint8_t ll_len[kHuffMaxSyms];
    uint8_t d_len[32];
    uint8_t rsym[kHuffMaxSyms + 32 + 8];    // RLE symbols 0..18
    uint8_t rextra[kHuffMaxSyms + 32 + 8];  // repeat count, 0 = no extra field
    uint32_t cl_freq[19];
    uint8_t cl_len[19];
    uint32_t cl_tab[19];
    HuffWork hw;
};

// code-length run-length coding (run_length_encode + rle_to_zlib_codes, src/compress.rs:472-569).
// One run of `count` equal values v: its items go to rsym / rextra from index `out` on (kEmit) or are only
// counted; returns the index behind them.
template <bool kEmit>
ZG_HD inline int rle_run(uint8_t v, int count, uint8_t *rsym, uint8_t *rextra, int out) {
    auto item = [&](uint8_t sym, uint8_t extra) {
        if (kEmit) {
            rsym[out] = sym;
            rextra[out] = extra;
        }
        out++;
    };
    if (count == 1) {  // Once(v)
        item(v, 0);
    } else if (v == 0) {  // zlib_rle_encode_zero
        int rep = count;
        while (rep > 0) {
            if (rep >= 3 && rep <= 10) {
                item(17, (uint8_t)rep);
                break;
            } else if (rep >= 11) {
                const int cur = rep > 138 ? 138 : rep;
                rep -= cur;
                item(18, (uint8_t)cur);
            } else {
                for (int i = 0; i < rep; i++) item(0, 0);
                break;
            }
        }
    } else {  // zlib_rle_encode_nonzero: the value itself, then a repeat of up to 6, and again
        int rep = count;
        while (rep > 0) {
            if (rep <= 3) {
                for (int i = 0; i < rep; i++) item(v, 0);
                break;
            }
            item(v, 0);
            rep -= 1;
            const int cur = rep > 6 ? 6 : rep;
            rep -= cur;
            item(16, (uint8_t)cur);
        }
    }
    return out;
}

// one vector, run by run
ZG_HD inline int rle_lengths(const uint8_t *cl, int n, uint8_t *rsym, uint8_t *rextra, int out) {
    int pos = 0;
    while (pos < n) {
        const uint8_t v = cl[pos];
        int count = 1;
        while (pos + count < n && cl[pos + count] == v) count++;
        pos += count;
        out = rle_run<true>(v, count, rsym, rextra, out);
    }
    return out;
}

// Builds both code tables from the block's histograms and serialises the block
// header (BFINAL, BTYPE and, for Dynamic, the code description) into hdr.
// freq_ll must already count the end-of-block symbol once.  Returns the number
// of header bits; *payload_bits receives the bits of all tokens plus EOB.
ZG_HD inline uint32_t build_block(const uint32_t *freq_ll, const uint32_t *freq_d, int strategy, bool bfinal,
                                  BlockTables &tab, uint32_t *hdr, uint64_t *payload_bits, HeaderWork &hw) {
    BitSink sink{hdr, 0};
    sink.put(bfinal ? 1u : 0u, 1);
    if (strategy == ZG_FIXED) {  // get_zlib_fixed, src/huffman.rs:353-367
        sink.put(1, 2);
        for (int i = 0; i < 288; i++) hw.ll_len[i] = (uint8_t)(i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8);
        for (int i = 0; i < 30; i++) hw.d_len[i] = 5;
        huff_codes(hw.ll_len, 288, tab.ll);
        huff_codes(hw.d_len, 30, tab.d);
    } else {
        sink.put(2, 2);
        huff_lengths(freq_ll, 286, 15, hw.ll_len, hw.hw);
        huff_lengths(freq_d, 30, 15, hw.d_len, hw.hw);
        huff_codes(hw.ll_len, 286, tab.ll);
        huff_codes(hw.d_len, 30, tab.d);
        // get_codelengths, src/compress.rs:376-407
        int lmax = 257;
        for (int i = 285; i > 257; i--)
            if (hw.ll_len[i]) {
                lmax = i;
                break;
            }
        int dmax = 1;
        bool any_d = false;
        for (int i = 29; i >= 0; i--)
            if (hw.d_len[i]) {
                dmax = i;
                any_d = true;
                break;
            }
        if (!any_d) dmax = 1;
        const int nl = lmax + 1, nd = dmax + 1;
        int nr = rle_lengths(hw.ll_len, nl, hw.rsym, hw.rextra, 0);  // the two vectors are coded separately
        nr = rle_lengths(hw.d_len, nd, hw.rsym, hw.rextra, nr);
        for (int i = 0; i < 19; i++) hw.cl_freq[i] = 0;
        for (int i = 0; i < nr; i++) hw.cl_freq[hw.rsym[i]]++;
        huff_lengths(hw.cl_freq, 19, 7, hw.cl_len, hw.hw);  // get_code_tree
        huff_codes(hw.cl_len, 19, hw.cl_tab);
        const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        int nh = 19;  // get_hcodes: trailing zeros trimmed
        while (nh > 4 && hw.cl_len[order[nh - 1]] == 0) nh--;
        sink.put((uint32_t)(nl - 257), 5);
        sink.put((uint32_t)(nd - 1), 5);
        sink.put((uint32_t)(nh - 4), 4);
        for (int i = 0; i < nh; i++) sink.put(hw.cl_len[order[i]], 3);
        for (int i = 0; i < nr; i++) {
            const uint32_t s = hw.rsym[i], t = hw.cl_tab[s];
            sink.put(t & 0xFFFFu, t >> 16);
            if (s == 16) sink.put((uint32_t)hw.rextra[i] - 3u, 2);
            else if (s == 17) sink.put((uint32_t)hw.rextra[i] - 3u, 3);
            else if (s == 18) sink.put((uint32_t)hw.rextra[i] - 11u, 7);
        }
    }
    uint64_t bits = 0;
    for (int i = 0; i < 286; i++) {
        if (!freq_ll[i]) continue;
        const uint32_t extra = i > 256 ? length_extra_bits((uint32_t)i - 257u) : 0u;
        bits += (uint64_t)freq_ll[i] * ((tab.ll[i] >> 16) + extra);
    }
    for (int i = 0; i < 30; i++) {
        if (!freq_d[i]) continue;
        bits += (uint64_t)freq_d[i] * ((tab.d[i] >> 16) + dist_extra_bits((uint32_t)i));
    }
    *payload_bits = bits;
    return sink.nbits;
}

}  // namespace zg
