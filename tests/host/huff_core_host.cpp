// Host build of the single-lane block builder (zlib_amd/csrc/zg_huff_core.h)
// for the CPU unit tests.  Test scaffolding only: the product runs this code
// on the GPU (k_huffman); nothing here is a CPU path of the library.
#include <string.h>

#include "../../zlib_amd/csrc/zg_huff_core.h"

extern "C" int hc_build_block(const uint32_t *freq_ll, const uint32_t *freq_d, int strategy, int bfinal,
                              uint32_t *tab_ll /*288*/, uint32_t *tab_d /*32*/, uint32_t *hdr /*160*/,
                              uint32_t *hdr_bits, uint64_t *payload_bits) {
    static thread_local zg::HeaderWork hw;
    zg::BlockTables tab;
    memset(&tab, 0, sizeof tab);
    memset(hdr, 0, 160 * 4);
    *hdr_bits = zg::build_block(freq_ll, freq_d, strategy, bfinal != 0, tab, hdr, payload_bits, hw);
    memcpy(tab_ll, tab.ll, sizeof tab.ll);
    memcpy(tab_d, tab.d, sizeof tab.d);
    return 0;
}
extern "C" void hc_codes(uint32_t len, uint32_t dist, uint32_t *out /*6*/) {
    const uint32_t li = zg::length_code_index(len), dc = zg::dist_code(dist);
    out[0] = 257 + li;
    out[1] = zg::length_extra_bits(li);
    out[2] = zg::length_base(li);
    out[3] = dc;
    out[4] = zg::dist_extra_bits(dc);
    out[5] = zg::dist_base(dc);
}
