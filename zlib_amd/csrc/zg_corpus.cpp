// Deterministic synthetic corpora for tests and benchmarks (host side only).
// Every MiB segment is generated from its own xorshift64* stream, so any byte
// range can be produced independently and in parallel.
#include <stdint.h>
#include <string.h>

#include <string>

#include <algorithm>
#include <thread>
#include <vector>

#include "../../include/zlib_amd.h"

namespace {

struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) {
        // splitmix64 to spread the seed
        uint64_t z = seed + 0x9E3779B97F4A7C15ULL;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        s = z ^ (z >> 31);
        if (!s) s = 0x1234567ULL;
    }
    uint64_t next() {
        s ^= s >> 12;
        s ^= s << 25;
        s ^= s >> 27;
        return s * 0x2545F4914F6CDD1DULL;
    }
    uint32_t below(uint32_t n) { return (uint32_t)((next() >> 32) * (uint64_t)n >> 32); }
};

constexpr size_t kSeg = 1u << 20;
constexpr int kVocab = 4096;

struct Vocab {
    std::vector<std::string> words;
    std::vector<uint32_t> cdf;  // cumulative Zipf weights scaled to 2^32
    Vocab() {
        static const char letters[] = "eeeeeeeeeeeetttttttttaaaaaaaaooooooooiiiiiiinnnnnnnsssssshhhhhhrrrrrrddddllllcccuuummwwffggyyppbbvkjxqz";
        Rng r(0xC0FFEEULL);
        for (int i = 0; i < kVocab; i++) {
            const int len = 2 + (int)r.below(8);
            std::string w;
            for (int k = 0; k < len; k++) w.push_back(letters[r.below(sizeof(letters) - 1)]);
            words.push_back(w);
        }
        double total = 0;
        for (int i = 0; i < kVocab; i++) total += 1.0 / (i + 1);
        double acc = 0;
        for (int i = 0; i < kVocab; i++) {
            acc += 1.0 / (i + 1);
            double v = acc / total * 4294967296.0;
            cdf.push_back(v >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)v);
        }
    }
};
const Vocab &vocab() {
    static Vocab v;
    return v;
}

void gen_text(Rng &r, uint8_t *out, size_t n) {
    const Vocab &v = vocab();
    size_t pos = 0, col = 0;
    while (pos < n) {
        const uint32_t u = (uint32_t)(r.next() >> 32);
        const int idx = (int)(std::lower_bound(v.cdf.begin(), v.cdf.end(), u) - v.cdf.begin());
        const std::string &w = v.words[idx < kVocab ? idx : kVocab - 1];
        for (char ch : w) {
            if (pos >= n) return;
            out[pos++] = (uint8_t)ch;
        }
        col += w.size() + 1;
        if (pos >= n) return;
        if (col > 72 + r.below(16)) {
            out[pos++] = '\n';
            col = 0;
        } else {
            out[pos++] = ' ';
        }
    }
}

void gen_random(Rng &r, uint8_t *out, size_t n) {
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        const uint64_t v = r.next();
        memcpy(out + i, &v, 8);
    }
    if (i < n) {
        const uint64_t v = r.next();
        memcpy(out + i, &v, n - i);
    }
}

// 16-byte records: u32 counter, u32 small-range value, three mantissa bytes + a fixed exponent byte, four zero bytes
void gen_binary(Rng &r, uint64_t seg_index, uint8_t *out, size_t n) {
    uint32_t counter = (uint32_t)(seg_index * (kSeg / 16));
    for (size_t i = 0; i < n; i += 16) {
        uint8_t rec[16];
        const uint32_t small = r.below(1000);
        const uint32_t mant = (uint32_t)(r.next() >> 40);
        memcpy(rec, &counter, 4);
        memcpy(rec + 4, &small, 4);
        rec[8] = (uint8_t)mant;
        rec[9] = (uint8_t)(mant >> 8);
        rec[10] = (uint8_t)((mant >> 16) & 0x7F);
        rec[11] = 0x3F;
        memset(rec + 12, 0, 4);
        counter++;
        memcpy(out + i, rec, std::min<size_t>(16, n - i));
    }
}

// 4 KiB records of JSON-like lines built from the text vocabulary; a few degenerate ones
void gen_record(uint64_t seed, uint64_t rec_index, uint8_t *out) {
    Rng r(seed * 0x100000001B3ULL + rec_index * 0x9E3779B97F4A7C15ULL + 5);
    const uint32_t kind = r.below(64);
    if (kind == 0) {
        memset(out, 0, 4096);
        return;
    }
    if (kind == 1) {
        gen_random(r, out, 4096);
        return;
    }
    if (kind == 2) {
        static const char phrase[] = "status=ok ";
        for (size_t i = 0; i < 4096; i++) out[i] = (uint8_t)phrase[i % (sizeof(phrase) - 1)];
        return;
    }
    const Vocab &v = vocab();
    std::string s;
    uint32_t id = (uint32_t)rec_index * 7u;
    while (s.size() < 4096) {
        s += "{\"id\":" + std::to_string(id++) + ",\"name\":\"" + v.words[r.below(256)] + "\",\"tags\":[\"" +
             v.words[r.below(64)] + "\",\"" + v.words[r.below(1024)] + "\"],\"score\":" + std::to_string(r.below(10000)) +
             ",\"ok\":" + (r.below(2) ? "true" : "false") + "}\n";
    }
    memcpy(out, s.data(), 4096);
}

// kind of MiB segment `seg` of the mixed corpus: text, random, text, binary
int mixed_kind(uint64_t seg) {
    static const int pat[4] = {0, 1, 0, 2};
    return pat[seg & 3];
}

void fill_segment(int kind, uint64_t seed, uint64_t seg, uint8_t *tmp) {
    Rng r(seed ^ (seg * 0xD1342543DE82EF95ULL + 0x2545F4914F6CDD1DULL * (uint64_t)(kind + 1)));
    const int k = kind == 3 ? mixed_kind(seg) : kind;
    if (k == 0) gen_text(r, tmp, kSeg);
    else if (k == 1) gen_random(r, tmp, kSeg);
    else gen_binary(r, seg, tmp, kSeg);
}

}  // namespace

extern "C" int zg_corpus_fill(int kind, uint64_t seed, uint64_t offset, uint8_t *buf, size_t n) {
    if (kind < 0 || kind > 4 || (!buf && n)) return ZG_ERR_ARG;
    if (!n) return ZG_OK;
    (void)vocab();
    if (kind == 4) {
        std::vector<uint8_t> rec(4096);
        uint64_t pos = offset;
        const uint64_t end = offset + n;
        while (pos < end) {
            const uint64_t ri = pos / 4096, ro = pos % 4096;
            gen_record(seed, ri, rec.data());
            const size_t take = (size_t)std::min<uint64_t>(4096 - ro, end - pos);
            memcpy(buf + (pos - offset), rec.data() + ro, take);
            pos += take;
        }
        return ZG_OK;
    }
    const uint64_t seg0 = offset / kSeg, seg1 = (offset + n - 1) / kSeg;
    const uint64_t nseg = seg1 - seg0 + 1;
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 1;
    if (nt > 16) nt = 16;
    if (nt > nseg) nt = (unsigned)nseg;
    auto work = [&](unsigned t) {
        std::vector<uint8_t> tmp(kSeg);
        for (uint64_t s = seg0 + t; s <= seg1; s += nt) {
            fill_segment(kind, seed, s, tmp.data());
            const uint64_t lo = std::max<uint64_t>(offset, s * kSeg), hi = std::min<uint64_t>(offset + n, (s + 1) * kSeg);
            memcpy(buf + (lo - offset), tmp.data() + (lo - s * kSeg), (size_t)(hi - lo));
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nt; t++) pool.emplace_back(work, t);
    for (auto &th : pool) th.join();
    return ZG_OK;
}
