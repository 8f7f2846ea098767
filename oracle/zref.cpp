/*
 * zref.cpp -- CPU oracle: a restatement of the mkpro118/zlib crate's
 * algorithms in plain C++17, one section per reference file.
 *
 * TEST INFRASTRUCTURE ONLY (see zref.h).  Citations are file:line under the
 * reference tree.  Where the reference panics or is undefined, the strict
 * entry points return a ZO_PANIC_* code; the "build mode" entry points apply
 * the documented extensions (length-limited codes, one-bit code for a lone
 * symbol, correct stored-block LEN, chunked blocks with a 32 KiB window).
 *
 * Parity pinning: the reference could not be compiled here (Rust, no
 * toolchain); this file is pinned by the reference's own known-answer unit
 * tests (tests/test_oracle_kat.py) and by system zlib as an independent
 * inflater.  LZ77 token-level parity is pinned only by following the
 * reference loop statement by statement.
 */
#include "zref.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <thread>
#include <tuple>
#include <utility>
#include <vector>

/* ------------------------------------------------------------------------ */
/* helpers                                                                  */
/* ------------------------------------------------------------------------ */
template <class T> static T *dup_out(const std::vector<T> &v) {
    T *p = (T *)malloc((v.size() ? v.size() : 1) * sizeof(T));
    if (p && !v.empty()) memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
}
extern "C" void zo_free(void *p) { free(p); }

/* ------------------------------------------------------------------------ */
/* src/bitwriter.rs:19-190                                                  */
/* ------------------------------------------------------------------------ */
struct zo_bw {
    std::vector<uint8_t> buffer; /* :20 */
    uint8_t byte = 0;            /* :21 */
    size_t numbits = 0;          /* :22 */
    /* bits written so far including padding introduced by flushes */
    uint64_t bitpos() const { return (uint64_t)buffer.size() * 8 + numbits; }
};
static void bw_flush_byte(zo_bw *w) { /* :162-169 */
    if (w->numbits > 0) {
        w->buffer.push_back(w->byte);
        w->byte = 0;
        w->numbits = 0;
    }
}
static void bw_write_bit(zo_bw *w, unsigned bit) { /* :66-73 */
    if (w->numbits == 8) bw_flush_byte(w);
    w->byte |= (uint8_t)((bit & 1u) << w->numbits);
    w->numbits += 1;
}
static void bw_write_bits(zo_bw *w, size_t value, size_t nbits) { /* :90-98, one bit per turn */
    while (nbits > 0) {
        bw_write_bit(w, (unsigned)(value & 1));
        value >>= 1;
        nbits -= 1;
    }
}
static void bw_write_byte(zo_bw *w, unsigned byte) { /* :114-119 */
    bw_flush_byte(w);
    w->buffer.push_back((uint8_t)byte);
}
static void bw_write_bytes(zo_bw *w, const uint8_t *p, size_t n) { /* :134-138 */
    for (size_t i = 0; i < n; i++) bw_write_byte(w, p[i]);
}
extern "C" {
zo_bw *zo_bw_new(void) { return new zo_bw(); }
void zo_bw_free(zo_bw *w) { delete w; }
void zo_bw_write_bit(zo_bw *w, unsigned bit) { bw_write_bit(w, bit); }
void zo_bw_write_bits(zo_bw *w, size_t value, size_t nbits) { bw_write_bits(w, value, nbits); }
void zo_bw_write_byte(zo_bw *w, unsigned byte) { bw_write_byte(w, byte); }
void zo_bw_write_bytes(zo_bw *w, const uint8_t *p, size_t n) { bw_write_bytes(w, p, n); }
void zo_bw_flush_byte(zo_bw *w) { bw_flush_byte(w); }
int zo_bw_finish(zo_bw *w, uint8_t **out, size_t *out_n) { /* :186-189 */
    bw_flush_byte(w);
    *out = dup_out(w->buffer);
    *out_n = w->buffer.size();
    return *out ? ZO_OK : ZO_ERR_ARG;
}
}

/* ------------------------------------------------------------------------ */
/* src/bitreader.rs:18-178                                                  */
/* ------------------------------------------------------------------------ */
struct zo_br {
    const uint8_t *mem;
    size_t n;
    size_t pos = 0;
    uint8_t byte = 0;
    long numbits = 0;
    bool oob = false; /* the reference panics on an out-of-bounds read (:61) */
};
static int br_read_byte(zo_br *r) { /* :59-64 */
    r->numbits = 0;
    if (r->pos >= r->n) {
        r->oob = true;
        return -1;
    }
    return r->mem[r->pos++];
}
static int br_read_bit(zo_br *r) { /* :79-92 */
    if (r->numbits <= 0) {
        int b = br_read_byte(r);
        if (b < 0) return -1;
        r->byte = (uint8_t)b;
        r->numbits = 8;
    }
    r->numbits -= 1;
    int bit = r->byte & 1;
    r->byte >>= 1;
    return bit;
}
static long long br_read_bits(zo_br *r, size_t n) { /* :107-115 */
    unsigned long long out = 0;
    for (size_t i = 0; i < n; i++) {
        int b = br_read_bit(r);
        if (b < 0) return -1;
        out |= (unsigned long long)b << i;
    }
    return (long long)out;
}
static long long br_read_bytes(zo_br *r, size_t n) { /* :130-139, little endian */
    unsigned long long out = 0;
    for (size_t i = 0; i < n; i++) {
        int b = br_read_byte(r);
        if (b < 0) return -1;
        out |= (unsigned long long)b << (8 * i);
    }
    return (long long)out;
}
extern "C" {
zo_br *zo_br_new(const uint8_t *mem, size_t n) {
    zo_br *r = new zo_br();
    r->mem = mem;
    r->n = n;
    return r;
}
void zo_br_free(zo_br *r) { delete r; }
int zo_br_read_byte(zo_br *r) { return br_read_byte(r); }
int zo_br_read_bit(zo_br *r) { return br_read_bit(r); }
long long zo_br_read_bits(zo_br *r, size_t n) { return br_read_bits(r, n); }
long long zo_br_read_bytes(zo_br *r, size_t n) { return br_read_bytes(r, n); }
int zo_code_to_bytes(size_t code, size_t length, uint8_t **out, size_t *out_n) { /* :158-178 */
    std::vector<uint8_t> bytes(1, 0);
    size_t numbits = 0;
    for (size_t k = length; k-- > 0;) {
        if (numbits >= 8) {
            bytes.push_back(0);
            numbits = 0;
        }
        bytes.back() |= (uint8_t)(((code & ((size_t)1 << k)) != 0) << numbits);
        numbits += 1;
    }
    *out = dup_out(bytes);
    *out_n = bytes.size();
    return ZO_OK;
}
}

/* ------------------------------------------------------------------------ */
/* src/adler.rs:1-13                                                        */
/* ------------------------------------------------------------------------ */
extern "C" uint32_t zo_adler32(const uint8_t *data, size_t n) {
    const uint32_t M = 65521; /* :1 */
    uint32_t a = 1, b = 0;
    for (size_t i = 0; i < n; i++) {
        a = (a + data[i]) % M; /* :8 */
        b = (b + a) % M;       /* :9 */
    }
    return b << 16 | a; /* :13 */
}

/* ------------------------------------------------------------------------ */
/* src/huffman.rs:26-49 tables, :848-871 lookups                            */
/* ------------------------------------------------------------------------ */
static const size_t CODE_LENGTH_CODES_ORDER[19] = {16, 17, 18, 0, 8,  7, 9,  6, 10, 5,
                                                   11, 4,  12, 3, 13, 2, 14, 1, 15};
static const size_t LENGTH_EXTRA_BITS[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2,
                                             2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const size_t LENGTH_BASE[29] = {3,  4,  5,  6,  7,  8,  9,  10, 11,  13,  15,  17,  19,  23, 27,
                                       31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const size_t DISTANCE_EXTRA_BITS[30] = {0, 0, 0, 0, 1, 1, 2, 2,  3,  3,  4,  4,  5,  5,  6,
                                               6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
static const size_t DISTANCE_BASE[30] = {1,   2,   3,   4,   5,   7,    9,    13,   17,   25,
                                         33,  49,  65,  97,  129, 193,  257,  385,  513,  769,
                                         1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const size_t ZLIB_WINDOW_SIZE = (size_t)1 << 15; /* :47 */
static const size_t ZLIB_MIN_STRING_LENGTH = 3;         /* :48 */
static const size_t ZLIB_MAX_STRING_LENGTH = 258;       /* :49 */

/* Rust's slice::binary_search contract: Ok(i) when found, Err(i) = insertion
 * point otherwise.  The tables are strictly increasing so any correct binary
 * search returns the same value. */
static bool rust_binary_search(const size_t *tab, size_t n, size_t x, size_t *idx) {
    size_t lo = 0, hi = n;
    while (lo < hi) {
        size_t mid = lo + (hi - lo) / 2;
        if (tab[mid] < x)
            lo = mid + 1;
        else
            hi = mid;
    }
    *idx = lo;
    return lo < n && tab[lo] == x;
}
static size_t get_length_code(size_t length) { /* :848-853 */
    size_t i;
    return rust_binary_search(LENGTH_BASE, 29, length, &i) ? i + 257 : i + 256;
}
static size_t get_distance_code(size_t distance) { /* :866-871 */
    size_t i;
    return rust_binary_search(DISTANCE_BASE, 30, distance, &i) ? i : i - 1;
}
extern "C" {
const size_t *zo_code_length_codes_order(void) { return CODE_LENGTH_CODES_ORDER; }
const size_t *zo_length_extra_bits(void) { return LENGTH_EXTRA_BITS; }
const size_t *zo_length_base(void) { return LENGTH_BASE; }
const size_t *zo_distance_extra_bits(void) { return DISTANCE_EXTRA_BITS; }
const size_t *zo_distance_base(void) { return DISTANCE_BASE; }
size_t zo_get_length_code(size_t length) { return get_length_code(length); }
size_t zo_get_distance_code(size_t distance) { return get_distance_code(distance); }
}

/* ------------------------------------------------------------------------ */
/* src/huffman.rs:51-762 HuffmanTree                                        */
/* ------------------------------------------------------------------------ */
struct Node { /* :53-60 */
    long sym = -1;
    int left = -1, right = -1;
};
struct zo_tree { /* :68-74 */
    std::vector<Node> nodes;
    int root = 0;
    bool has_map = false;
    std::map<uint32_t, std::pair<size_t, size_t>> map; /* symbol -> (code, len) */
    bool zero_len_code = false; /* assign() met a leaf at depth 0 (:633 overflows in debug) */
    zo_tree() { nodes.push_back(Node()); }
};

static void tree_insert(zo_tree *t, size_t code, size_t length, uint32_t symbol) { /* :407-440 */
    int node = t->root;
    for (size_t k = length; k-- > 0;) {
        if ((code & ((size_t)1 << k)) == 0) {
            if (t->nodes[node].left < 0) {
                t->nodes.push_back(Node());
                t->nodes[node].left = (int)t->nodes.size() - 1;
            }
            node = t->nodes[node].left;
        } else {
            if (t->nodes[node].right < 0) {
                t->nodes.push_back(Node());
                t->nodes[node].right = (int)t->nodes.size() - 1;
            }
            node = t->nodes[node].right;
        }
    }
    t->nodes[node].sym = (long)symbol;
}

static uint32_t reverse_bits32(uint32_t v) {
    uint32_t r = 0;
    for (int i = 0; i < 32; i++) r |= ((v >> i) & 1u) << (31 - i);
    return r;
}

static void tree_assign(zo_tree *t) { /* :619-645, breadth first */
    t->has_map = true;
    std::vector<std::tuple<int, size_t, size_t>> queue;
    queue.emplace_back(t->root, (size_t)0, (size_t)0);
    for (size_t qi = 0; qi < queue.size(); qi++) {
        auto [node, code, length] = queue[qi];
        const Node &nd = t->nodes[node];
        if (nd.sym >= 0) {
            /* :632-633 ((code as u32).reverse_bits() >> (32 - length)); length 0
             * overflows the shift: debug builds panic, release builds wrap. */
            size_t c;
            if (length == 0) {
                t->zero_len_code = true;
                c = reverse_bits32((uint32_t)code);
            } else {
                c = reverse_bits32((uint32_t)code) >> (32 - length);
            }
            t->map[(uint32_t)nd.sym] = {c, length};
        } else {
            if (nd.left >= 0) queue.emplace_back(nd.left, code << 1, length + 1);
            if (nd.right >= 0) queue.emplace_back(nd.right, (code << 1) | 1, length + 1);
        }
    }
}

static size_t tree_depth(const zo_tree *t, int node) { /* :756-761 */
    const Node &nd = t->nodes[node];
    size_t l = nd.left >= 0 ? tree_depth(t, nd.left) : 0;
    size_t r = nd.right >= 0 ? tree_depth(t, nd.right) : 0;
    return 1 + std::max(l, r);
}
static size_t tree_max_code_len(const zo_tree *t) { /* :748-753 */
    if (t->has_map) {
        size_t m = 0;
        for (auto &kv : t->map) m = std::max(m, kv.second.second);
        return m;
    }
    return tree_depth(t, t->root);
}

static zo_tree *tree_from_bitlen_list(const size_t *bitlen, size_t nbl, const uint32_t *alphabet, size_t nalpha) {
    /* :311-337; the counts run over the whole bitlen slice (:312-321) while the
     * inserts zip it with the alphabet and stop at the shorter one (:328-336) */
    size_t n = std::min(nbl, nalpha);
    size_t max_bits = 0;
    for (size_t i = 0; i < nbl; i++) max_bits = std::max(max_bits, bitlen[i]);
    std::vector<size_t> bitlen_count(max_bits + 1, 0);
    for (size_t y = 1; y <= max_bits; y++)
        for (size_t i = 0; i < nbl; i++) bitlen_count[y] += (bitlen[i] == y);
    std::vector<size_t> next_code = {0, 0};
    for (size_t bits = 1; bits < max_bits; bits++)
        next_code.push_back((next_code[bits] + bitlen_count[bits]) << 1);
    zo_tree *t = new zo_tree();
    for (size_t i = 0; i < n; i++) {
        if (bitlen[i] == 0) continue;
        tree_insert(t, next_code[bitlen[i]], bitlen[i], alphabet[i]);
        next_code[bitlen[i]] += 1;
    }
    return t;
}

static zo_tree *tree_to_canonical(const zo_tree *t) { /* :714-734 */
    if (!t->has_map) return nullptr;
    std::vector<uint32_t> syms;
    std::vector<size_t> bl;
    for (auto &kv : t->map) { /* std::map iterates in ascending symbol order = sort_unstable */
        syms.push_back(kv.first);
        bl.push_back(kv.second.second);
    }
    zo_tree *c = tree_from_bitlen_list(bl.data(), bl.size(), syms.data(), syms.size());
    tree_assign(c);
    return c;
}

/* ---- from_freq: src/huffman.rs:185-216 with FreqNode ordering :95-122 ---- */
struct FreqNode {
    size_t count;
    int node;      /* index into the pool */
    long sym;      /* symbol of a leaf, -1 for an internal node */
    uint64_t seq;  /* creation order, used only by ZO_TIE_FIFO */
};
/* Ordering::{Less,Equal,Greater} as -1/0/1; this is `self.cmp(other)`. */
static int freqnode_cmp(const FreqNode &self, const FreqNode &other, int tie) {
    int cmp = (other.count < self.count) ? -1 : (other.count > self.count) ? 1 : 0; /* :105 reversed */
    if (cmp != 0) return cmp;
    if (self.sym >= 0 && other.sym >= 0) /* :108 sym_other.cmp(&sym_self) */
        return (other.sym < self.sym) ? -1 : (other.sym > self.sym) ? 1 : 0;
    if (self.sym >= 0) return 1;   /* :109 leaf beats internal */
    if (other.sym >= 0) return -1; /* :110 */
    if (tie == ZO_TIE_FIFO)        /* :111 is Equal in the reference; the build picks the older node */
        return (self.seq < other.seq) ? 1 : (self.seq > other.seq) ? -1 : 0;
    return 0;
}
/* A max-heap with the element moves of Rust's std::collections::BinaryHeap
 * (rebuild / push = sift_up / pop = swap_remove + sift_down_to_bottom + sift_up),
 * restated from the published std source; with ZO_TIE_HEAP the result for
 * equal-weight internal nodes depends on exactly these moves. */
struct RustHeap {
    std::vector<FreqNode> d;
    int tie;
    bool le(const FreqNode &a, const FreqNode &b) const { return freqnode_cmp(a, b, tie) <= 0; }
    bool lt(const FreqNode &a, const FreqNode &b) const { return freqnode_cmp(a, b, tie) < 0; }
    bool ge(const FreqNode &a, const FreqNode &b) const { return freqnode_cmp(a, b, tie) >= 0; }
    size_t sift_up(size_t start, size_t pos) {
        FreqNode elem = d[pos];
        while (pos > start) {
            size_t parent = (pos - 1) / 2;
            if (le(elem, d[parent])) break;
            d[pos] = d[parent];
            pos = parent;
        }
        d[pos] = elem;
        return pos;
    }
    void sift_down_range(size_t pos, size_t end) {
        FreqNode elem = d[pos];
        size_t child = 2 * pos + 1;
        while (child <= (end >= 2 ? end - 2 : 0) && end >= 2) {
            child += le(d[child], d[child + 1]) ? 1 : 0;
            if (ge(elem, d[child])) {
                d[pos] = elem;
                return;
            }
            d[pos] = d[child];
            pos = child;
            child = 2 * pos + 1;
        }
        if (child == end - 1 && lt(elem, d[child])) {
            d[pos] = d[child];
            pos = child;
        }
        d[pos] = elem;
    }
    void sift_down_to_bottom(size_t pos) {
        size_t end = d.size(), start = pos;
        FreqNode elem = d[pos];
        size_t child = 2 * pos + 1;
        while (end >= 2 && child <= end - 2) {
            child += le(d[child], d[child + 1]) ? 1 : 0;
            d[pos] = d[child];
            pos = child;
            child = 2 * pos + 1;
        }
        if (child == end - 1) {
            d[pos] = d[child];
            pos = child;
        }
        d[pos] = elem;
        sift_up(start, pos);
    }
    void rebuild() {
        size_t n = d.size() / 2;
        while (n > 0) {
            n -= 1;
            sift_down_range(n, d.size());
        }
    }
    void push(const FreqNode &x) {
        d.push_back(x);
        sift_up(0, d.size() - 1);
    }
    FreqNode pop() {
        FreqNode item = d.back();
        d.pop_back();
        if (!d.empty()) {
            std::swap(item, d[0]);
            sift_down_to_bottom(0);
        }
        return item;
    }
};

static uint64_t xorshift64s(uint64_t &s) {
    s ^= s >> 12;
    s ^= s << 25;
    s ^= s >> 27;
    return s * 0x2545F4914F6CDD1DULL;
}

static zo_tree *tree_from_freq(std::vector<std::pair<uint32_t, size_t>> freq, int tie, uint64_t seed) {
    /* :187-191 the HashMap iteration order is unspecified: FIFO mode is order
     * independent (total order); HEAP mode shuffles by `seed`. */
    if (tie == ZO_TIE_HEAP) {
        uint64_t s = seed * 0x9E3779B97F4A7C15ULL + 0x1234567ULL;
        if (!s) s = 1;
        for (size_t i = freq.size(); i > 1; i--) std::swap(freq[i - 1], freq[xorshift64s(s) % i]);
    }
    zo_tree *t = new zo_tree();
    t->nodes.clear();
    RustHeap heap;
    heap.tie = tie;
    uint64_t seq = 0;
    for (auto &kv : freq) {
        if (kv.second == 0) continue; /* :189 */
        Node nd;
        nd.sym = (long)kv.first;
        t->nodes.push_back(nd);
        heap.d.push_back(FreqNode{kv.second, (int)t->nodes.size() - 1, (long)kv.first, seq++});
    }
    heap.rebuild(); /* collect::<BinaryHeap<_>>() = From<Vec> = rebuild */
    if (heap.d.empty()) { /* :194-196 */
        delete t;
        return new zo_tree();
    }
    while (heap.d.size() > 1) { /* :199-209 */
        FreqNode left = heap.pop();
        FreqNode right = heap.pop();
        Node parent;
        parent.left = left.node;
        parent.right = right.node;
        t->nodes.push_back(parent);
        heap.push(FreqNode{left.count + right.count, (int)t->nodes.size() - 1, -1, seq++});
    }
    t->root = heap.pop().node; /* :211 */
    return t;
}

static zo_tree *tree_from_data(const uint8_t *data, size_t n, int tie, uint64_t seed) { /* :159-167 */
    std::map<uint32_t, size_t> m;
    for (size_t i = 0; i < n; i++) m[data[i]] += 1;
    std::vector<std::pair<uint32_t, size_t>> f(m.begin(), m.end());
    return tree_from_freq(f, tie, seed);
}

static bool tok_is_marker(uint32_t t) { return ZO_TOK_IS_MARKER(t); }

/* histograms of from_lz77 (:243-264) */
static void lz77_histograms(const uint32_t *tokens, size_t n, size_t sym_freq[286], size_t dist_freq[30]) {
    for (int i = 0; i < 286; i++) sym_freq[i] = 0;
    for (int i = 0; i < 30; i++) dist_freq[i] = 0;
    for (size_t i = 0; i < n; i++) {
        uint32_t t = tokens[i];
        if (!tok_is_marker(t)) {
            sym_freq[t & 0xFF] += 1; /* :251 */
        } else {
            sym_freq[get_length_code(ZO_TOK_LEN(t))] += 1;   /* :254,257 */
            dist_freq[get_distance_code(ZO_TOK_DIST(t))] += 1; /* :255,258 */
        }
    }
    sym_freq[256] = std::max<size_t>(1, sym_freq[256]); /* :264 */
}

static void tree_from_lz77(const uint32_t *tokens, size_t n, int tie, uint64_t seed, zo_tree **lt,
                           zo_tree **dt) { /* :238-288 (check_lz77 is the caller's business) */
    size_t sf[286], df[30];
    lz77_histograms(tokens, n, sf, df);
    std::vector<std::pair<uint32_t, size_t>> a, b;
    for (uint32_t i = 0; i < 286; i++) a.push_back({i, sf[i]});
    for (uint32_t i = 0; i < 30; i++) b.push_back({i, df[i]});
    *lt = tree_from_freq(a, tie, seed);
    *dt = tree_from_freq(b, tie, seed ^ 0xD15Cu);
}

static void tree_get_zlib_fixed(zo_tree **lt, zo_tree **dt) { /* :353-367 */
    std::vector<size_t> bitlen(144, 8);
    bitlen.insert(bitlen.end(), 256 - 144, 9);
    bitlen.insert(bitlen.end(), 280 - 256, 7);
    bitlen.insert(bitlen.end(), 288 - 280, 8);
    std::vector<uint32_t> alpha;
    for (uint32_t i = 0; i < 286; i++) alpha.push_back(i); /* :785-789, 286 symbols: zip drops 286,287 */
    *lt = tree_from_bitlen_list(bitlen.data(), bitlen.size(), alpha.data(), 286);
    std::vector<size_t> dl(30, 5);
    std::vector<uint32_t> dalpha;
    for (uint32_t i = 0; i < 30; i++) dalpha.push_back(i);
    *dt = tree_from_bitlen_list(dl.data(), 30, dalpha.data(), 30);
}

static long tree_decode(const zo_tree *t, zo_br *r) { /* :469-493 */
    int node = t->root;
    while (t->nodes[node].left >= 0 || t->nodes[node].right >= 0) {
        int bit = br_read_bit(r);
        if (bit < 0) return -2;
        int next = bit == 0 ? t->nodes[node].left : t->nodes[node].right;
        if (next < 0) return -1;
        node = next;
    }
    return t->nodes[node].sym;
}

static int tree_decode_trees(zo_br *r, zo_tree **lt, zo_tree **dt) { /* :520-592 */
    long long v;
    if ((v = br_read_bits(r, 5)) < 0) return ZO_ERR_MALFORMED;
    size_t hlit = (size_t)v + 257;
    if ((v = br_read_bits(r, 5)) < 0) return ZO_ERR_MALFORMED;
    size_t hdist = (size_t)v + 1;
    if ((v = br_read_bits(r, 4)) < 0) return ZO_ERR_MALFORMED;
    size_t hclen = (size_t)v + 4;
    size_t cl_bl[19] = {0};
    for (size_t i = 0; i < hclen; i++) {
        if ((v = br_read_bits(r, 3)) < 0) return ZO_ERR_MALFORMED;
        cl_bl[CODE_LENGTH_CODES_ORDER[i]] = (size_t)v;
    }
    uint32_t cl_alpha[19];
    for (uint32_t i = 0; i < 19; i++) cl_alpha[i] = i;
    zo_tree *clt = tree_from_bitlen_list(cl_bl, 19, cl_alpha, 19);
    std::vector<size_t> bitlen;
    size_t maxlen = hlit + hdist;
    while (bitlen.size() < maxlen) {
        long sym = tree_decode(clt, r);
        if (sym < 0) { /* :549 panic */
            delete clt;
            return ZO_ERR_MALFORMED;
        }
        if (sym <= 15) {
            bitlen.push_back((size_t)sym);
        } else if (sym == 16) {
            if (bitlen.empty() || (v = br_read_bits(r, 2)) < 0) { /* :560-562 expect */
                delete clt;
                return ZO_ERR_MALFORMED;
            }
            size_t prev = bitlen.back();
            bitlen.insert(bitlen.end(), (size_t)v + 3, prev);
        } else if (sym == 17) {
            if ((v = br_read_bits(r, 3)) < 0) {
                delete clt;
                return ZO_ERR_MALFORMED;
            }
            bitlen.insert(bitlen.end(), (size_t)v + 3, 0);
        } else if (sym == 18) {
            if ((v = br_read_bits(r, 7)) < 0) {
                delete clt;
                return ZO_ERR_MALFORMED;
            }
            bitlen.insert(bitlen.end(), (size_t)v + 11, 0);
        } else { /* :578 */
            delete clt;
            return ZO_ERR_MALFORMED;
        }
    }
    delete clt;
    std::vector<uint32_t> la, da;
    for (uint32_t i = 0; i < 286; i++) la.push_back(i);
    for (uint32_t i = 0; i < 30; i++) da.push_back(i);
    /* :583-589 */
    *lt = tree_from_bitlen_list(bitlen.data(), hlit, la.data(), 286);
    *dt = tree_from_bitlen_list(bitlen.data() + hlit, bitlen.size() - hlit, da.data(), 30);
    return ZO_OK;
}

extern "C" {
zo_tree *zo_tree_new(void) { return new zo_tree(); }
void zo_tree_free(zo_tree *t) { delete t; }
void zo_tree_insert(zo_tree *t, size_t code, size_t length, uint32_t symbol) {
    tree_insert(t, code, length, symbol);
}
void zo_tree_assign(zo_tree *t) { tree_assign(t); }
int zo_tree_encode(const zo_tree *t, uint32_t symbol, size_t *code, size_t *len) { /* :668-670 */
    if (!t->has_map) return 0;
    auto it = t->map.find(symbol);
    if (it == t->map.end()) return 0;
    if (code) *code = it->second.first;
    if (len) *len = it->second.second;
    return 1;
}
size_t zo_tree_n_codes(const zo_tree *t) { return t->has_map ? t->map.size() : 0; } /* :382-384 */
size_t zo_tree_max_code_len(const zo_tree *t) { return tree_max_code_len(t); }
zo_tree *zo_tree_to_canonical(const zo_tree *t) { return tree_to_canonical(t); }
zo_tree *zo_tree_from_bitlen_list(const size_t *bitlen, size_t n_bitlen, const uint32_t *alphabet,
                                  size_t n_alphabet) {
    return tree_from_bitlen_list(bitlen, n_bitlen, alphabet, n_alphabet);
}
zo_tree *zo_tree_from_data(const uint8_t *data, size_t n, int tie, uint64_t seed) {
    return tree_from_data(data, n, tie, seed);
}
zo_tree *zo_tree_from_freq(const uint32_t *syms, const size_t *counts, size_t n, int tie, uint64_t seed) {
    std::vector<std::pair<uint32_t, size_t>> f;
    for (size_t i = 0; i < n; i++) f.push_back({syms[i], counts[i]});
    return tree_from_freq(f, tie, seed);
}
void zo_tree_get_zlib_fixed(zo_tree **ltree, zo_tree **dtree) { tree_get_zlib_fixed(ltree, dtree); }
int zo_tree_from_lz77(const uint32_t *tokens, size_t n, int tie, uint64_t seed, zo_tree **ltree,
                      zo_tree **dtree) {
    for (size_t i = 0; i < n; i++)
        if (tok_is_marker(tokens[i]) && ZO_TOK_LEN(tokens[i]) < 3) return ZO_ERR_ARG; /* :253 */
    tree_from_lz77(tokens, n, tie, seed, ltree, dtree);
    return ZO_OK;
}
long zo_tree_decode(const zo_tree *t, zo_br *r) { return tree_decode(t, r); }
int zo_tree_decode_trees(zo_br *r, zo_tree **ltree, zo_tree **dtree) {
    return tree_decode_trees(r, ltree, dtree);
}
size_t zo_tree_encodings(const zo_tree *t, uint32_t *syms, size_t *codes, size_t *lens, size_t cap) {
    if (!t->has_map) return 0;
    size_t i = 0;
    for (auto &kv : t->map) {
        if (i < cap) {
            if (syms) syms[i] = kv.first;
            if (codes) codes[i] = kv.second.first;
            if (lens) lens[i] = kv.second.second;
        }
        i++;
    }
    return i;
}
}

/* ------------------------------------------------------------------------ */
/* src/lz77.rs:143-207                                                      */
/* ------------------------------------------------------------------------ */
static uint32_t tok_literal(uint8_t b) { return b; }
static uint32_t tok_marker(size_t len, size_t dist) {
    return ZO_TOK_MARKER | ((uint32_t)len << 16) | (uint32_t)dist;
}
/* With `literal` set, the slice comparison at :171-175 re-compares the whole
 * prefix on every step exactly as the reference does (its cost model, used
 * for the CPU baseline timing); otherwise the same predicate is evaluated
 * incrementally (equal prefixes are known from the previous step).  The
 * results are identical. */
static std::vector<uint32_t> lz77_compress(const uint8_t *data, size_t n, size_t start, size_t window_size,
                                           size_t min_len, size_t max_len, bool literal = false) {
    std::vector<uint32_t> compressed;
    size_t pos = start;
    size_t last_pos;
    if (n > min_len) { /* :150-154 */
        last_pos = n - min_len;
    } else {
        for (size_t i = start; i < n; i++) compressed.push_back(tok_literal(data[i]));
        return compressed;
    }
    while (pos < last_pos) { /* :156 */
        size_t search_start = pos > window_size ? pos - window_size : 0; /* :157-161 */
        size_t match_length = min_len;                                   /* :163 */
        bool found_match = false;
        size_t best_match_distance = 0, best_match_length = 0;
        while (search_start + match_length < pos) { /* :169 */
            size_t end1 = std::min(n, search_start + match_length); /* :170-171 */
            size_t end2 = std::min(n, pos + match_length);          /* :172-173 */
            size_t l1 = end1 - search_start, l2 = end2 - pos;
            bool eq;
            if (l1 != l2) {
                eq = false;
            } else if (literal || !found_match) {
                eq = memcmp(data + search_start, data + pos, l1) == 0;
            } else { /* prefix of length match_length-1 already known equal */
                eq = data[search_start + l1 - 1] == data[pos + l1 - 1];
            }
            if (eq && match_length < max_len) { /* :175 */
                match_length += 1;
                found_match = true;
            } else {
                size_t true_match_length = match_length - 1; /* :179 */
                if (found_match && true_match_length > best_match_length) {
                    best_match_distance = pos - search_start;
                    best_match_length = true_match_length;
                }
                match_length = min_len;
                search_start += 1;
                found_match = false;
            }
        }
        if (best_match_length > 0) { /* :192-198 */
            compressed.push_back(tok_marker(best_match_length, best_match_distance));
            pos += best_match_length;
        } else {
            compressed.push_back(tok_literal(data[pos]));
            pos += 1;
        }
    }
    for (size_t i = pos; i < n; i++) compressed.push_back(tok_literal(data[i])); /* :201-204 */
    return compressed;
}

/* The same token stream through an exact 3-gram index.  For position `pos`,
 * only candidates sharing the first three bytes can pass the first compare of
 * the reference loop; they are visited in the reference's ascending order and
 * its three rules are applied unchanged: a candidate whose growing match
 * reaches `pos` ends the whole search without being recorded; a candidate is
 * recorded at length min(M, max_len-1); strict '>' keeps the farthest. */
static std::vector<uint32_t> lz77_fast(const uint8_t *data, size_t n, size_t start, size_t window_size,
                                       size_t max_len) {
    const size_t min_len = 3;
    std::vector<uint32_t> out;
    if (n <= min_len) {
        for (size_t i = start; i < n; i++) out.push_back(tok_literal(data[i]));
        return out;
    }
    size_t last_pos = n - min_len;
    size_t npos = n - 2; /* positions with a full 3-gram */
    std::vector<uint64_t> idx(npos);
    for (size_t p = 0; p < npos; p++) {
        uint64_t key = ((uint64_t)data[p] << 16) | ((uint64_t)data[p + 1] << 8) | data[p + 2];
        idx[p] = key << 32 | (uint64_t)p;
    }
    std::sort(idx.begin(), idx.end());
    std::vector<uint32_t> rank(npos);
    for (size_t i = 0; i < npos; i++) rank[(uint32_t)idx[i]] = (uint32_t)i;
    size_t pos = start;
    while (pos < last_pos) {
        size_t lo_pos = pos > window_size ? pos - window_size : 0;
        uint64_t key = idx[rank[pos]] >> 32;
        auto first = std::lower_bound(idx.begin(), idx.begin() + rank[pos], key << 32 | (uint64_t)lo_pos);
        size_t cap = n - pos;
        size_t best_len = 0, best_dist = 0;
        for (auto it = first; it != idx.begin() + rank[pos]; ++it) {
            size_t s = (uint32_t)*it;
            if (s + min_len >= pos) break; /* loop guard at match_length == min_len */
            size_t m = 3;
            while (m < cap && data[s + m] == data[pos + m]) m++;
            size_t k = std::min(m + 1, max_len); /* value of match_length when the compare fails */
            if (s + k >= pos) break;             /* the loop guard fails first: search over */
            size_t len = k - 1;
            if (len > best_len) {
                best_len = len;
                best_dist = pos - s;
                if (len == max_len - 1 || len == cap) break; /* cannot be exceeded */
            }
        }
        if (best_len > 0) {
            out.push_back(tok_marker(best_len, best_dist));
            pos += best_len;
        } else {
            out.push_back(tok_literal(data[pos]));
            pos += 1;
        }
    }
    for (size_t i = pos; i < n; i++) out.push_back(tok_literal(data[i]));
    return out;
}

extern "C" {
int zo_lz77_compress(const uint8_t *data, size_t n, size_t start, size_t window, size_t min_len,
                     size_t max_len, uint32_t **tokens, size_t *n_tokens) {
    if (start > n || min_len == 0) return ZO_ERR_ARG;
    auto v = lz77_compress(data, n, start, std::min(window, ZLIB_WINDOW_SIZE), min_len, max_len);
    *tokens = dup_out(v);
    *n_tokens = v.size();
    return ZO_OK;
}
int zo_lz77_fast(const uint8_t *data, size_t n, size_t start, size_t window, size_t max_len,
                 uint32_t **tokens, size_t *n_tokens) {
    if (start > n || max_len < 4) return ZO_ERR_ARG;
    auto v = lz77_fast(data, n, start, std::min(window, ZLIB_WINDOW_SIZE), max_len);
    *tokens = dup_out(v);
    *n_tokens = v.size();
    return ZO_OK;
}
/* the verbatim loop with the reference's full re-compare on every step */
int zo_lz77_compress_literal(const uint8_t *data, size_t n, size_t start, size_t window, size_t min_len,
                             size_t max_len, uint32_t **tokens, size_t *n_tokens) {
    if (start > n || min_len == 0) return ZO_ERR_ARG;
    auto v = lz77_compress(data, n, start, std::min(window, ZLIB_WINDOW_SIZE), min_len, max_len, true);
    *tokens = dup_out(v);
    *n_tokens = v.size();
    return ZO_OK;
}
}

/* ------------------------------------------------------------------------ */
/* src/compress.rs:231-569                                                  */
/* ------------------------------------------------------------------------ */
typedef std::pair<size_t, long> RlePair; /* (symbol, extra or -1 for None) */

static void zlib_rle_encode_zero(std::vector<RlePair> &acc, size_t repetitions) { /* :514-541 */
    while (repetitions > 0) {
        if (repetitions >= 3 && repetitions <= 10) { /* SHORT_ZERO_MIN..=SHORT_ZERO_MAX */
            acc.push_back({17, (long)repetitions});
            break;
        } else if (repetitions >= 11) { /* LONG_ZERO_MIN.. */
            size_t current;
            if (repetitions > 138) {
                repetitions -= 138;
                current = 138;
            } else {
                current = repetitions;
                repetitions = 0;
            }
            acc.push_back({18, (long)current});
        } else {
            for (size_t i = 0; i < repetitions; i++) acc.push_back({0, -1});
            break;
        }
    }
}
static void zlib_rle_encode_nonzero(std::vector<RlePair> &acc, size_t num, size_t repetitions) {
    /* :543-569 */
    while (repetitions > 0) {
        if (repetitions <= 3) { /* ..=PREVIOUS_MIN */
            for (size_t i = 0; i < repetitions; i++) acc.push_back({num, -1});
            break;
        } else { /* PREVIOUS_MIN_RANGE.. */
            acc.push_back({num, -1});
            repetitions -= 1;
            size_t current;
            if (repetitions > 6) {
                repetitions -= 6;
                current = 6;
            } else {
                current = repetitions;
                repetitions = 0;
            }
            acc.push_back({16, (long)current});
        }
    }
}
/* run_length_encode :472-496; count 0 stands for Once(value) */
static std::vector<std::pair<size_t, size_t>> run_length_encode(const size_t *data, size_t len) {
    std::vector<std::pair<size_t, size_t>> enc;
    size_t pos = 0;
    while (pos < len) {
        size_t count = 1;
        while (pos + 1 < len && data[pos] == data[pos + 1]) {
            pos += 1;
            count += 1;
        }
        if (count > 1)
            enc.push_back({data[pos], count});
        else
            enc.push_back({data[pos], 0});
        pos += 1;
    }
    return enc;
}
static std::vector<RlePair> rle_to_zlib_codes(const std::vector<std::pair<size_t, size_t>> &rle) {
    /* :498-512 */
    std::vector<RlePair> acc;
    for (auto &e : rle) {
        if (e.second == 0) {
            acc.push_back({e.first, -1});
        } else if (e.first == 0) {
            zlib_rle_encode_zero(acc, e.second);
        } else {
            zlib_rle_encode_nonzero(acc, e.first, e.second);
        }
    }
    return acc;
}
static std::vector<RlePair> zlib_rle(const size_t *cl1, size_t n1, const size_t *cl2, size_t n2) {
    /* :458-470: the two vectors are encoded separately, then concatenated */
    auto a = rle_to_zlib_codes(run_length_encode(cl1, n1));
    auto b = rle_to_zlib_codes(run_length_encode(cl2, n2));
    a.insert(a.end(), b.begin(), b.end());
    return a;
}

static void get_codelengths(const zo_tree *ltree, const zo_tree *dtree, std::vector<size_t> &l,
                            std::vector<size_t> &d) { /* :376-407 */
    uint32_t lmax = 0x101; /* :390-394: max key if > 0x100, else 0x101 */
    if (!ltree->map.empty() && ltree->map.rbegin()->first > 0x100) lmax = ltree->map.rbegin()->first;
    uint32_t dmax = 1; /* :395 */
    if (!dtree->map.empty()) dmax = dtree->map.rbegin()->first;
    l.clear();
    d.clear();
    for (uint32_t c = 0; c <= lmax; c++) {
        auto it = ltree->map.find(c);
        l.push_back(it == ltree->map.end() ? 0 : it->second.second);
    }
    for (uint32_t c = 0; c <= dmax; c++) {
        auto it = dtree->map.find(c);
        d.push_back(it == dtree->map.end() ? 0 : it->second.second);
    }
}

/* The build's code-length policy where the reference has none.
 * (1) A lone used symbol gets a one-bit code (RFC 1951 section 3.2.7).
 * (2) Lengths above `max_bits` are limited with the count-rebalancing rule
 *     published in miniz (tdefl_huffman_enforce_max_code_size): fold the
 *     counts of too-long codes into max_bits, then until the Kraft sum is
 *     exactly one drop one code from max_bits and split the deepest shorter
 *     code into two; lengths are then dealt out longest-first to the symbols
 *     in ascending (count, symbol) order.
 * `lens` is indexed by symbol; returns true if anything changed. */
static bool build_fix_lengths(std::vector<size_t> &lens, const std::vector<size_t> &counts, size_t max_bits) {
    std::vector<uint32_t> used;
    for (uint32_t s = 0; s < counts.size(); s++)
        if (counts[s] > 0) used.push_back(s);
    if (used.size() == 1) {
        bool ch = lens[used[0]] != 1;
        lens[used[0]] = 1;
        return ch;
    }
    size_t mx = 0;
    for (uint32_t s : used) mx = std::max(mx, lens[s]);
    if (mx <= max_bits) return false;
    std::vector<size_t> num(std::max(mx, max_bits) + 2, 0);
    for (uint32_t s : used) num[lens[s]]++;
    for (size_t i = max_bits + 1; i < num.size(); i++) {
        num[max_bits] += num[i];
        num[i] = 0;
    }
    uint64_t total = 0;
    for (size_t i = max_bits; i >= 1; i--) total += (uint64_t)num[i] << (max_bits - i);
    while (total != ((uint64_t)1 << max_bits)) {
        num[max_bits]--;
        for (size_t i = max_bits - 1; i > 0; i--)
            if (num[i]) {
                num[i]--;
                num[i + 1] += 2;
                break;
            }
        total--;
    }
    std::sort(used.begin(), used.end(), [&](uint32_t a, uint32_t b) {
        return counts[a] != counts[b] ? counts[a] < counts[b] : a < b;
    });
    size_t j = 0;
    for (size_t bits = max_bits; bits >= 1; bits--)
        for (size_t c = 0; c < num[bits]; c++) lens[used[j++]] = bits;
    return true;
}

/* Build-mode code lengths for a histogram: Huffman by the reference's
 * algorithm with the FIFO tie-break, then build_fix_lengths. */
static std::vector<size_t> build_lengths(const std::vector<size_t> &counts, size_t max_bits) {
    std::vector<std::pair<uint32_t, size_t>> f;
    for (uint32_t s = 0; s < counts.size(); s++) f.push_back({s, counts[s]});
    zo_tree *t = tree_from_freq(f, ZO_TIE_FIFO, 0);
    tree_assign(t);
    std::vector<size_t> lens(counts.size(), 0);
    for (auto &kv : t->map) lens[kv.first] = kv.second.second;
    delete t;
    build_fix_lengths(lens, counts, max_bits);
    return lens;
}
/* canonical tree (assigned) for a length vector indexed by symbol */
static zo_tree *canonical_from_lengths(const std::vector<size_t> &lens) {
    std::vector<uint32_t> syms;
    std::vector<size_t> bl;
    for (uint32_t s = 0; s < lens.size(); s++)
        if (lens[s]) {
            syms.push_back(s);
            bl.push_back(lens[s]);
        }
    zo_tree *t = tree_from_bitlen_list(bl.data(), bl.size(), syms.data(), syms.size());
    tree_assign(t);
    return t;
}

static int get_code_tree(const std::vector<RlePair> &combined, bool strict, zo_tree **out) { /* :410-429 */
    std::vector<uint8_t> codes_only;
    for (auto &p : combined) codes_only.push_back((uint8_t)p.first);
    if (strict) {
        zo_tree *t = tree_from_data(codes_only.data(), codes_only.size(), ZO_TIE_FIFO, 0);
        tree_assign(t);
        if (t->zero_len_code) {
            delete t;
            return ZO_PANIC_ZERO_LEN_CODE;
        }
        zo_tree *c = tree_to_canonical(t);
        delete t;
        if (tree_max_code_len(c) > 7) { /* :422-426 */
            delete c;
            return ZO_PANIC_CL_TOO_LONG;
        }
        *out = c;
        return ZO_OK;
    }
    std::vector<size_t> counts(19, 0);
    for (uint8_t c : codes_only) counts[c]++;
    *out = canonical_from_lengths(build_lengths(counts, 7));
    return ZO_OK;
}

static int get_hcodes(const zo_tree *code_tree, std::vector<size_t> &hcodes) { /* :432-449 */
    hcodes.clear();
    for (int i = 0; i < 19; i++) {
        auto it = code_tree->map.find((uint32_t)CODE_LENGTH_CODES_ORDER[i]);
        hcodes.push_back(it == code_tree->map.end() ? 0 : it->second.second);
    }
    size_t trailing = 0; /* :444 rev().position(x > 0).unwrap_or(0) */
    bool any = false;
    for (size_t i = 0; i < 19; i++)
        if (hcodes[18 - i] > 0) {
            trailing = i;
            any = true;
            break;
        }
    if (!any) trailing = 0;
    hcodes.resize(19 - trailing);
    if (hcodes.size() < 4) return ZO_PANIC_HCODES; /* :447 */
    return ZO_OK;
}

static int write_dynamic_header(zo_bw *w, const zo_tree *ltree, const zo_tree *dtree, bool strict) {
    /* :231-274 */
    std::vector<size_t> lcl, dcl;
    get_codelengths(ltree, dtree, lcl, dcl);
    auto combined = zlib_rle(lcl.data(), lcl.size(), dcl.data(), dcl.size());
    zo_tree *code_tree = nullptr;
    int rc = get_code_tree(combined, strict, &code_tree);
    if (rc) return rc;
    std::vector<size_t> hcodes;
    rc = get_hcodes(code_tree, hcodes);
    if (rc) {
        delete code_tree;
        return rc;
    }
    size_t hlit = lcl.size() - 257, hdist = dcl.size() - 1, hclen = hcodes.size() - 4;
    bw_write_bits(w, hlit, 5);
    bw_write_bits(w, hdist, 5);
    bw_write_bits(w, hclen, 4);
    for (size_t h : hcodes) bw_write_bits(w, h, 3);
    for (auto &p : combined) {
        auto it = code_tree->map.find((uint32_t)p.first);
        if (it == code_tree->map.end()) {
            delete code_tree;
            return ZO_PANIC_NO_ENCODING;
        }
        bw_write_bits(w, it->second.first, it->second.second);
        if (p.second >= 0) {
            switch (p.first) { /* :265-270 */
            case 16: bw_write_bits(w, (size_t)p.second - 3, 2); break;
            case 17: bw_write_bits(w, (size_t)p.second - 3, 3); break;
            case 18: bw_write_bits(w, (size_t)p.second - 11, 7); break;
            default: break;
            }
        }
    }
    delete code_tree;
    return ZO_OK;
}

static int literal_writer(const zo_tree *ltree, zo_bw *w, uint32_t sym) { /* :316-321 */
    auto it = ltree->map.find(sym);
    if (!ltree->has_map || it == ltree->map.end()) return ZO_PANIC_NO_ENCODING;
    bw_write_bits(w, it->second.first, it->second.second);
    return ZO_OK;
}
static int length_writer(const zo_tree *ltree, zo_bw *w, size_t length) { /* :324-347 */
    if (length < 3 || length > 258) return ZO_ERR_ARG; /* :325-326 */
    size_t code = get_length_code(length);
    auto it = ltree->map.find((uint32_t)code);
    if (!ltree->has_map || it == ltree->map.end()) return ZO_PANIC_NO_ENCODING;
    bw_write_bits(w, it->second.first, it->second.second);
    size_t extra_bits = LENGTH_EXTRA_BITS[code - 257];
    if (extra_bits > 0) bw_write_bits(w, length - LENGTH_BASE[code - 257], extra_bits);
    return ZO_OK;
}
static int distance_writer(const zo_tree *dtree, zo_bw *w, size_t distance) { /* :350-374 */
    if (distance < 1 || distance > 32768) return ZO_ERR_ARG; /* :355-356 */
    size_t code = get_distance_code(distance);
    auto it = dtree->map.find((uint32_t)code);
    if (!dtree->has_map || it == dtree->map.end()) return ZO_PANIC_NO_ENCODING;
    bw_write_bits(w, it->second.first, it->second.second);
    size_t extra_bits = DISTANCE_EXTRA_BITS[code];
    if (extra_bits > 0) bw_write_bits(w, distance - DISTANCE_BASE[code], extra_bits);
    return ZO_OK;
}
static int write_compressed_data(zo_bw *w, const uint32_t *tokens, size_t n, const zo_tree *ltree,
                                 const zo_tree *dtree) { /* :276-294 */
    for (size_t i = 0; i < n; i++) {
        uint32_t t = tokens[i];
        int rc;
        if (!tok_is_marker(t)) {
            rc = literal_writer(ltree, w, t & 0xFF);
        } else {
            rc = length_writer(ltree, w, ZO_TOK_LEN(t));
            if (!rc) rc = distance_writer(dtree, w, ZO_TOK_DIST(t));
        }
        if (rc) return rc;
    }
    return literal_writer(ltree, w, 0x100); /* :293 */
}

/* create_dynamic_trees :296-314, strict */
static int create_dynamic_trees_strict(const uint32_t *tokens, size_t n, zo_tree **lt, zo_tree **dt) {
    zo_tree *l, *d;
    tree_from_lz77(tokens, n, ZO_TIE_FIFO, 0, &l, &d);
    size_t lmax = tree_max_code_len(l), dmax = tree_max_code_len(d); /* no map yet: node depth */
    int rc = ZO_OK;
    if (lmax > 15) rc = ZO_PANIC_LL_TOO_LONG;
    else if (dmax > 15) rc = ZO_PANIC_D_TOO_LONG;
    if (!rc) {
        tree_assign(l);
        tree_assign(d);
        if (l->zero_len_code || d->zero_len_code) rc = ZO_PANIC_ZERO_LEN_CODE;
    }
    if (rc) {
        delete l;
        delete d;
        return rc;
    }
    *lt = tree_to_canonical(l);
    *dt = tree_to_canonical(d);
    delete l;
    delete d;
    return ZO_OK;
}
/* build-mode trees: same construction, with build_fix_lengths where the
 * reference would panic; RFC 1951 limit of 15 bits. */
static void create_dynamic_trees_build(const uint32_t *tokens, size_t n, zo_tree **lt, zo_tree **dt) {
    size_t sf[286], df[30];
    lz77_histograms(tokens, n, sf, df);
    std::vector<size_t> lc(sf, sf + 286), dc(df, df + 30);
    *lt = canonical_from_lengths(build_lengths(lc, 15));
    *dt = canonical_from_lengths(build_lengths(dc, 15));
}

static std::vector<uint32_t> zlib_tokens(const uint8_t *data, size_t n, size_t start, bool fast) {
    /* get_zlib_compressor :451-456 */
    return fast ? lz77_fast(data, n, start, ZLIB_WINDOW_SIZE, ZLIB_MAX_STRING_LENGTH)
                : lz77_compress(data, n, start, ZLIB_WINDOW_SIZE, ZLIB_MIN_STRING_LENGTH,
                                ZLIB_MAX_STRING_LENGTH);
}

static void write_zlib_header(zo_bw *w) { /* :53-72 */
    const uint8_t cmf = 0x70 | 0x08;
    bw_write_byte(w, cmf);
    size_t fcheck = 31 - (((size_t)cmf * 256) % 31);
    uint8_t flg = (uint8_t)fcheck & (uint8_t) ~(0x20 | 0xC0);
    bw_write_byte(w, flg);
}
static void write_trailer(zo_bw *w, const uint8_t *data, size_t n) { /* :82-83 */
    uint32_t c = zo_adler32(data, n);
    uint8_t be[4] = {(uint8_t)(c >> 24), (uint8_t)(c >> 16), (uint8_t)(c >> 8), (uint8_t)c};
    bw_write_bytes(w, be, 4);
}

static int auto_choice(const uint8_t *data, size_t n) { /* :89-169 */
    const size_t ONE_KB = 1024;
    if (n < 256) return ZO_RAW;
    if (n < ONE_KB) return ZO_FIXED;
    double preference = 0.0;
    size_t freq[256] = {0};
    for (size_t i = 0; i < ONE_KB; i++) freq[data[i]]++;
    /* :126-130; the reference sums in HashMap order, here ascending symbol order */
    double ent = 0.0;
    size_t distinct = 0;
    for (int s = 0; s < 256; s++)
        if (freq[s]) {
            double p = (double)freq[s] / (double)ONE_KB;
            ent += p * std::log2(p);
            distinct++;
        }
    double entropy = -ent;
    double log2_alphabet = std::log2((double)distinct);
    double proximity = 1.0 - std::fabs((entropy / log2_alphabet) - 1.0);
    preference += proximity > 0.75 ? proximity : 0.5 * proximity;
    /* :141-145 the five SMALLEST counts (as written) */
    std::vector<uint32_t> sorted;
    for (int s = 0; s < 256; s++)
        if (freq[s]) sorted.push_back((uint32_t)freq[s]);
    std::sort(sorted.begin(), sorted.end());
    size_t max_idx = std::min<size_t>(5, sorted.size());
    uint32_t sum = 0;
    for (size_t i = 0; i < max_idx; i++) sum += sorted[i];
    double proportion = (double)sum / (double)ONE_KB;
    preference -= proportion > 0.5 ? proportion : 0.5 * proportion;
    if (preference > 0.1) return ZO_FIXED;
    if (preference < -0.1) return ZO_DYNAMIC;
    double unique_chars = (double)distinct, data_len = (double)n;
    double estimated_ratio = unique_chars * std::log2(data_len) / data_len;
    return estimated_ratio < 0.5 ? ZO_DYNAMIC : ZO_FIXED;
}

/* one Fixed or Dynamic block for `tokens` */
static int write_block(zo_bw *w, const uint32_t *tokens, size_t n, int strategy, bool bfinal, bool strict) {
    bw_write_bit(w, bfinal ? 1 : 0); /* :199 / :215 (always 1 in the reference) */
    zo_tree *lt = nullptr, *dt = nullptr;
    int rc = ZO_OK;
    if (strategy == ZO_FIXED) {
        bw_write_bits(w, 1, 2); /* :201 */
        tree_get_zlib_fixed(&lt, &dt);
        tree_assign(lt);
        tree_assign(dt);
    } else {
        bw_write_bits(w, 2, 2); /* :217 */
        if (strict) {
            rc = create_dynamic_trees_strict(tokens, n, &lt, &dt);
            if (rc) return rc;
        } else {
            create_dynamic_trees_build(tokens, n, &lt, &dt);
        }
        rc = write_dynamic_header(w, lt, dt, strict);
    }
    if (!rc) rc = write_compressed_data(w, tokens, n, lt, dt);
    delete lt;
    delete dt;
    return rc;
}

static void write_stored_blocks(zo_bw *w, const uint8_t *data, size_t n) {
    /* :172-195 with LEN = the block's own length and an empty final block for
     * empty input (the reference's LEN is data.len() for every block). */
    const size_t B = 16 * 1024;
    size_t nblocks = n == 0 ? 1 : (n + B - 1) / B;
    for (size_t b = 0; b < nblocks; b++) {
        size_t off = b * B, len = std::min(B, n - off);
        bw_write_bit(w, b + 1 == nblocks);
        bw_write_bits(w, 0, 2);
        uint16_t l16 = (uint16_t)len, nl = (uint16_t)~l16;
        uint8_t hdr[4] = {(uint8_t)(l16 & 0xff), (uint8_t)(l16 >> 8), (uint8_t)(nl & 0xff), (uint8_t)(nl >> 8)};
        bw_write_bytes(w, hdr, 4);
        bw_write_bytes(w, data + off, len);
    }
}

extern "C" {
int zo_literal_writer(const zo_tree *ltree, zo_bw *w, uint32_t sym) { return literal_writer(ltree, w, sym); }
int zo_length_writer(const zo_tree *ltree, zo_bw *w, size_t length) { return length_writer(ltree, w, length); }
int zo_distance_writer(const zo_tree *dtree, zo_bw *w, size_t distance) {
    return distance_writer(dtree, w, distance);
}
int zo_write_compressed_data(zo_bw *w, const uint32_t *tokens, size_t n, const zo_tree *ltree,
                             const zo_tree *dtree) {
    return write_compressed_data(w, tokens, n, ltree, dtree);
}
int zo_write_dynamic_header(zo_bw *w, const zo_tree *ltree, const zo_tree *dtree, int strict) {
    return write_dynamic_header(w, ltree, dtree, strict != 0);
}
int zo_get_codelengths(const zo_tree *ltree, const zo_tree *dtree, size_t **cl1, size_t *n1, size_t **cl2,
                       size_t *n2) {
    if (!ltree->has_map || !dtree->has_map) return ZO_ERR_ARG; /* :381-386 expect */
    std::vector<size_t> l, d;
    get_codelengths(ltree, dtree, l, d);
    *cl1 = dup_out(l);
    *n1 = l.size();
    *cl2 = dup_out(d);
    *n2 = d.size();
    return ZO_OK;
}
static int out_pairs(const std::vector<RlePair> &v, size_t **sym, long **extra, size_t *n) {
    std::vector<size_t> s;
    std::vector<long> e;
    for (auto &p : v) {
        s.push_back(p.first);
        e.push_back(p.second);
    }
    *sym = dup_out(s);
    *extra = dup_out(e);
    *n = v.size();
    return ZO_OK;
}
int zo_zlib_rle(const size_t *cl1, size_t n1, const size_t *cl2, size_t n2, size_t **sym, long **extra,
                size_t *n) {
    return out_pairs(zlib_rle(cl1, n1, cl2, n2), sym, extra, n);
}
int zo_run_length_encode(const size_t *data, size_t n, size_t **value, size_t **count, size_t *m) {
    auto r = run_length_encode(data, n);
    std::vector<size_t> v, c;
    for (auto &p : r) {
        v.push_back(p.first);
        c.push_back(p.second);
    }
    *value = dup_out(v);
    *count = dup_out(c);
    *m = r.size();
    return ZO_OK;
}
int zo_rle_to_zlib_codes(const size_t *value, const size_t *count, size_t m, size_t **sym, long **extra,
                         size_t *n) {
    std::vector<std::pair<size_t, size_t>> r;
    for (size_t i = 0; i < m; i++) r.push_back({value[i], count[i]});
    return out_pairs(rle_to_zlib_codes(r), sym, extra, n);
}
int zo_zlib_rle_encode_zero(size_t repetitions, size_t **sym, long **extra, size_t *n) {
    std::vector<RlePair> acc;
    zlib_rle_encode_zero(acc, repetitions);
    return out_pairs(acc, sym, extra, n);
}
int zo_zlib_rle_encode_nonzero(size_t num, size_t repetitions, size_t **sym, long **extra, size_t *n) {
    std::vector<RlePair> acc;
    zlib_rle_encode_nonzero(acc, num, repetitions);
    return out_pairs(acc, sym, extra, n);
}
int zo_get_code_tree(const size_t *sym, size_t n, int strict, zo_tree **out) {
    std::vector<RlePair> c;
    for (size_t i = 0; i < n; i++) c.push_back({sym[i], -1});
    return get_code_tree(c, strict != 0, out);
}
int zo_get_hcodes(const zo_tree *code_tree, size_t **hcodes, size_t *n) {
    std::vector<size_t> h;
    int rc = get_hcodes(code_tree, h);
    *hcodes = dup_out(h);
    *n = h.size();
    return rc;
}

int zo_compress_reference(const uint8_t *data, size_t n, int strategy, uint8_t **out, size_t *out_n) {
    /* src/compress.rs:51-86 */
    zo_bw w;
    write_zlib_header(&w);
    if (strategy == ZO_AUTO) strategy = auto_choice(data, n);
    int rc = ZO_OK;
    if (strategy == ZO_RAW) {
        if (n == 0 || n > 16 * 1024) return ZO_PANIC_RAW_DEFECT; /* :173 underflow, :183 wrong LEN */
        write_stored_blocks(&w, data, n); /* identical to :172-195 when there is exactly one block */
    } else if (strategy == ZO_FIXED || strategy == ZO_DYNAMIC) {
        auto tokens = zlib_tokens(data, n, 0, false);
        rc = write_block(&w, tokens.data(), tokens.size(), strategy, true, true);
    } else {
        return ZO_ERR_ARG;
    }
    if (rc) return rc;
    write_trailer(&w, data, n);
    bw_flush_byte(&w);
    *out = dup_out(w.buffer);
    *out_n = w.buffer.size();
    return ZO_OK;
}

static std::vector<uint32_t> chunk_tokens(const uint8_t *data, size_t n, size_t chunk_bytes, size_t c,
                                          bool fast) {
    size_t cs = c * chunk_bytes, ce = std::min(n, cs + chunk_bytes);
    size_t lo = cs > ZLIB_WINDOW_SIZE ? cs - ZLIB_WINDOW_SIZE : 0;
    return zlib_tokens(data + lo, ce - lo, cs - lo, fast);
}

int zo_compress_chunked(const uint8_t *data, size_t n, int strategy, size_t chunk_bytes, int fast,
                        uint8_t **out, size_t *out_n) {
    if (chunk_bytes == 0) return ZO_ERR_ARG;
    zo_bw w;
    write_zlib_header(&w);
    if (strategy == ZO_AUTO) strategy = auto_choice(data, n);
    if (strategy == ZO_RAW) {
        write_stored_blocks(&w, data, n);
    } else if (strategy == ZO_FIXED || strategy == ZO_DYNAMIC) {
        size_t nchunks = n == 0 ? 1 : (n + chunk_bytes - 1) / chunk_bytes;
        for (size_t c = 0; c < nchunks; c++) {
            auto tokens = chunk_tokens(data, n, chunk_bytes, c, fast != 0);
            int rc = write_block(&w, tokens.data(), tokens.size(), strategy, c + 1 == nchunks, false);
            if (rc) return rc;
        }
    } else {
        return ZO_ERR_ARG;
    }
    write_trailer(&w, data, n);
    bw_flush_byte(&w);
    *out = dup_out(w.buffer);
    *out_n = w.buffer.size();
    return ZO_OK;
}

int zo_chunk_tokens(const uint8_t *data, size_t n, size_t chunk_bytes, size_t chunk_index, int fast,
                    uint32_t **tokens, size_t *n_tokens) {
    if (chunk_bytes == 0 || (n > 0 && chunk_index * chunk_bytes >= n) || (n == 0 && chunk_index > 0))
        return ZO_ERR_ARG;
    auto v = chunk_tokens(data, n, chunk_bytes, chunk_index, fast != 0);
    *tokens = dup_out(v);
    *n_tokens = v.size();
    return ZO_OK;
}

int zo_chunk_tables(const uint32_t *tokens, size_t n, uint8_t ll_len[286], uint16_t ll_code[286],
                    uint8_t d_len[30], uint16_t d_code[30]) {
    zo_tree *lt, *dt;
    create_dynamic_trees_build(tokens, n, &lt, &dt);
    memset(ll_len, 0, 286);
    memset(ll_code, 0, 286 * 2);
    memset(d_len, 0, 30);
    memset(d_code, 0, 30 * 2);
    for (auto &kv : lt->map) {
        ll_len[kv.first] = (uint8_t)kv.second.second;
        ll_code[kv.first] = (uint16_t)kv.second.first;
    }
    for (auto &kv : dt->map) {
        d_len[kv.first] = (uint8_t)kv.second.second;
        d_code[kv.first] = (uint16_t)kv.second.first;
    }
    delete lt;
    delete dt;
    return ZO_OK;
}

int zo_block_bits(const uint32_t *tokens, size_t n, int strategy, uint64_t *bits) {
    zo_bw w;
    int rc = write_block(&w, tokens, n, strategy, true, false);
    *bits = w.bitpos();
    return rc;
}

int zo_write_block(const uint32_t *tokens, size_t n, int strategy, int bfinal, uint8_t **out, size_t *out_n,
                   uint64_t *bits) {
    zo_bw w;
    int rc = write_block(&w, tokens, n, strategy, bfinal != 0, false);
    *bits = w.bitpos();
    bw_flush_byte(&w);
    *out = dup_out(w.buffer);
    *out_n = w.buffer.size();
    return rc;
}

int zo_auto_choice(const uint8_t *data, size_t n) { return auto_choice(data, n); }
}

/* ------------------------------------------------------------------------ */
/* src/decompress.rs:34-189                                                 */
/* ------------------------------------------------------------------------ */
static int inflate_block_data(zo_br *r, const zo_tree *lt, const zo_tree *dt, std::vector<uint8_t> &buf) {
    /* :151-189 */
    for (;;) {
        long sym = tree_decode(lt, r);
        if (sym == -2) return ZO_ERR_MALFORMED; /* read past the end: panic at bitreader.rs:61 */
        if (sym < 0) return ZO_OK;              /* :158-160 a failed decode ends the block silently */
        if (sym <= 255) {
            buf.push_back((uint8_t)sym);
        } else if (sym == 256) {
            return ZO_OK;
        } else if (sym <= 285) {
            size_t idx = (size_t)sym - 257;
            long long e = br_read_bits(r, LENGTH_EXTRA_BITS[idx]);
            if (e < 0) return ZO_ERR_MALFORMED;
            size_t length = (size_t)e + LENGTH_BASE[idx];
            long dsym = tree_decode(dt, r);
            if (dsym < 0 || dsym > 29) return ZO_ERR_MALFORMED; /* :174 panic */
            e = br_read_bits(r, DISTANCE_EXTRA_BITS[dsym]);
            if (e < 0) return ZO_ERR_MALFORMED;
            size_t dist = (size_t)e + DISTANCE_BASE[dsym];
            if (dist > buf.size()) return ZO_ERR_MALFORMED; /* :183 index underflow */
            for (size_t i = 0; i < length; i++) buf.push_back(buf[buf.size() - dist]);
        } else {
            return ZO_ERR_MALFORMED; /* :186 */
        }
    }
}

extern "C" int zo_decompress(const uint8_t *in, size_t n, uint8_t **out, size_t *out_n, char *err,
                             size_t errcap) {
    auto fail = [&](int code, const char *msg) {
        if (err && errcap) snprintf(err, errcap, "%s", msg);
        return code;
    };
    zo_br r;
    r.mem = in;
    r.n = n;
    int cmf = br_read_byte(&r);
    if (cmf < 0) return fail(ZO_ERR_MALFORMED, "input exhausted");
    int cm = cmf & 0xF;
    if (cm != 8) { /* :44-46 */
        char m[96];
        snprintf(m, sizeof m, "CM = %d is not a supported compression method", cm);
        return fail(ZO_ERR_DECODE, m);
    }
    int cinfo = (cmf >> 4) & 0xF;
    if (cinfo > 7) { /* :51-55 */
        char m[96];
        snprintf(m, sizeof m, "Invalid compression info, must be < 7, found %d", cinfo);
        return fail(ZO_ERR_DECODE, m);
    }
    int flags = br_read_byte(&r);
    if (flags < 0) return fail(ZO_ERR_MALFORMED, "input exhausted");
    if (((size_t)cmf * 256 + (size_t)flags) % 31 != 0) /* :59-63 */
        return fail(ZO_ERR_DECODE, "CMF + FLAGS checksum failed!");
    if ((flags >> 5) & 1) return fail(ZO_ERR_DECODE, "Preset dictionaries are not supported"); /* :65-69 */
    std::vector<uint8_t> buf;
    bool final_block = false;
    while (!final_block) { /* inflate :97-118 */
        int b = br_read_bit(&r);
        if (b < 0) return fail(ZO_ERR_MALFORMED, "input exhausted");
        final_block = b == 1;
        long long bt = br_read_bits(&r, 2);
        if (bt < 0) return fail(ZO_ERR_MALFORMED, "input exhausted");
        int rc = ZO_OK;
        if (bt == 0) { /* :124-132, NLEN read and ignored */
            long long len = br_read_bytes(&r, 2), nlen = br_read_bytes(&r, 2);
            if (len < 0 || nlen < 0) return fail(ZO_ERR_MALFORMED, "input exhausted");
            for (long long i = 0; i < len; i++) {
                int v = br_read_byte(&r);
                if (v < 0) return fail(ZO_ERR_MALFORMED, "input exhausted");
                buf.push_back((uint8_t)v);
            }
        } else if (bt == 1) { /* :137-140 */
            zo_tree *lt, *dt;
            tree_get_zlib_fixed(&lt, &dt);
            rc = inflate_block_data(&r, lt, dt, buf);
            delete lt;
            delete dt;
        } else if (bt == 2) { /* :145-149 */
            zo_tree *lt = nullptr, *dt = nullptr;
            rc = tree_decode_trees(&r, &lt, &dt);
            if (!rc) rc = inflate_block_data(&r, lt, dt, buf);
            delete lt;
            delete dt;
        } else {
            return fail(ZO_ERR_DECODE, "Invalid block type"); /* :113 */
        }
        if (rc) return fail(rc, "malformed block body");
    }
    uint32_t adler = zo_adler32(buf.data(), buf.size()); /* :75 */
    uint32_t checksum = 0;
    for (int i = 0; i < 4; i++) { /* :78-82 big endian */
        int v = br_read_byte(&r);
        if (v < 0) return fail(ZO_ERR_MALFORMED, "input exhausted");
        checksum = checksum << 8 | (uint32_t)v;
    }
    if (adler != checksum) return fail(ZO_ERR_DECODE, "Checksum is invalid"); /* :83-87 */
    *out = dup_out(buf);
    *out_n = buf.size();
    return ZO_OK;
}

/* ------------------------------------------------------------------------ */
/* cpu baseline helper                                                      */
/* ------------------------------------------------------------------------ */
extern "C" int zo_bench_slices(const uint8_t *data, const size_t *off, const size_t *len, const size_t *halo,
                               size_t n_slices, int threads, uint64_t *bits_out) {
    if (threads < 1) threads = 1;
    std::atomic<size_t> next{0};
    std::atomic<int> status{0};
    auto work = [&]() {
        for (;;) {
            size_t i = next.fetch_add(1);
            if (i >= n_slices) return;
            const uint8_t *base = data + off[i] - halo[i];
            /* the reference's matcher with its own compare cost model */
            auto tokens = lz77_compress(base, halo[i] + len[i], halo[i], ZLIB_WINDOW_SIZE,
                                        ZLIB_MIN_STRING_LENGTH, ZLIB_MAX_STRING_LENGTH, true);
            zo_bw w;
            int rc = write_block(&w, tokens.data(), tokens.size(), ZO_DYNAMIC, true, false);
            if (rc) status.store(rc);
            volatile uint32_t sink = zo_adler32(data + off[i], len[i]); /* the trailer work of compress() */
            (void)sink;
            bits_out[i] = w.bitpos();
        }
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++) pool.emplace_back(work);
    for (auto &t : pool) t.join();
    return status.load();
}
