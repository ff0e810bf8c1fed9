// GPU test program for the GNU Radio source adaptor (cariboulite_amd/csrc/gr_source) compiled against the compile-check
// stub of the GNU Radio API slice (tests/cpp/gr_api_stub -- NOT GNU Radio): work() vs the oracle's restatement of what the
// reference's work() does with the same bytes -- unpack (caribou_smi.c:295-393), ((float)v)/4096.0f
// (CaribouLiteRadioCpp.cpp:91), and the tag loop (caribouLiteSource_impl.cc:113-119).
// Built and run by tests/test_gr_source.py; links liboracle (checker only).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "caribouLiteSourceHip.h"
#include "cl_oracle.h"

#define CHECK(c) do { if (!(c)) { printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

// n RX words; sync bit: dense = a random bit per word, otherwise only at the listed positions
static std::vector<uint8_t> make_stream(size_t n, int hif, uint32_t seed, bool dense, const std::vector<size_t> &pps)
{
    std::vector<uint8_t> b(4 * n);
    uint32_t s = seed;
    for (size_t k = 0; k < n; k++) {
        s = s * 1664525u + 1013904223u; uint32_t i13 = (s >> 8) & 0x1FFF;
        s = s * 1664525u + 1013904223u; uint32_t q13 = (s >> 8) & 0x1FFF;
        uint32_t a = hif ? q13 : i13, bb = hif ? i13 : q13;
        uint32_t w = 0x80004000u | (a << 17) | (bb << 1) | (dense ? ((s >> 30) & 1u) : 0u);
        memcpy(&b[4 * k], &w, 4);
    }
    for (size_t p : pps) b[4 * p] |= 1;
    return b;
}

// one work() call of `ask` items on the next bytes of `b` (starting at sample `at`), checked against the oracle
static int one_call(gr::caribouLite::caribouLiteSourceHip &blk, const std::vector<uint8_t> &b, size_t at, size_t ask, int hif,
                    bool with_meta, size_t *n_tags_out)
{
    std::vector<gr_complex> out(ask);
    std::vector<uint8_t> meta(ask, 0xEE);
    gr_vector_const_void_star in;
    gr_vector_void_star outs{out.data(), meta.data()};
    blk.stub_tags.clear();
    const int got = blk.work((int)ask, in, outs);
    CHECK(got == (int)ask);
    std::vector<int16_t> want(2 * (ask + 2));
    std::vector<uint8_t> wmeta(ask + 2);
    CHECK(orc_rx_data_analyze(hif, b.data() + 4 * at, 4 * ask, want.data(), wmeta.data()) == 0);
    std::vector<float> wf(2 * ask);
    orc_cs16_to_cf32(want.data(), wf.data(), ask);
    CHECK(memcmp(out.data(), wf.data(), 8 * ask) == 0);
    if (!with_meta) {
        CHECK(blk.stub_tags.empty());
        for (size_t k = 0; k < ask; k++) CHECK(meta[k] == 0xEE);               // the second output is not ours to touch
        return 0;
    }
    CHECK(memcmp(meta.data(), wmeta.data(), ask) == 0);
    std::vector<uint32_t> widx(ask + 1);
    const size_t k = orc_sync_tags(wmeta.data(), ask, widx.data(), ask);
    CHECK(blk.stub_tags.size() == k);
    for (size_t j = 0; j < k; j++) {
        CHECK(blk.stub_tags[j].offset == widx[j]);
        CHECK(blk.stub_tags[j].key->symbol == "pps" && blk.stub_tags[j].value->boolean);
    }
    if (n_tags_out) *n_tags_out = k;
    return 0;
}

int main()
{
    cl_smi *smi = cl_smi_init(0);
    CHECK(smi);
    const size_t MTU = 131072;
    {   // S1G with the meta output: a dense stream (more tags than the eager copy brings), then the real thing (a few markers)
        auto blk = gr::caribouLite::caribouLiteSourceHip::make(smi, 0, false, 40, 2.5e6f, 4e6f, 9e8f, true, 0);
        CHECK(blk->name() == "caribouLiteSource" && blk->output_signature()->sizeof_stream_items.size() == 2);
        gr_vector_const_void_star in; std::vector<gr_complex> o(16); std::vector<uint8_t> m(16);
        gr_vector_void_star outs{o.data(), m.data()};
        CHECK(blk->work(16, in, outs) == 0 && blk->stub_tags.empty());            // nothing fed yet: no items, no tags
        auto dense = make_stream(MTU + 50000, 0, 11, true, {});
        cl_smi_feed_bytes(smi, dense.data(), dense.size());
        size_t k = 0;
        if (one_call(*blk, dense, 0, MTU, 0, true, &k)) return 1;
        CHECK(k > 60000);
        if (one_call(*blk, dense, MTU, 50000, 0, true, &k)) return 1;              // a short call: tags stop at its end
        CHECK(k > 20000);
        auto sparse = make_stream(2 * MTU, 0, 12, false, {0, 1, 4095, 65536, MTU - 1, MTU, MTU + 77, 2 * MTU - 1});
        cl_smi_feed_bytes(smi, sparse.data(), sparse.size());
        if (one_call(*blk, sparse, 0, MTU, 0, true, &k)) return 1;
        CHECK(k == 5);
        if (one_call(*blk, sparse, MTU, MTU, 0, true, &k)) return 1;
        CHECK(k == 3);
        auto none = make_stream(1000, 0, 13, false, {});
        cl_smi_feed_bytes(smi, none.data(), none.size());
        if (one_call(*blk, none, 0, 1000, 0, true, &k)) return 1;
        CHECK(k == 0);
    }
    {   // HiF without the meta output: no tag work, the second output untouched
        auto blk = gr::caribouLite::caribouLiteSourceHip::make(smi, 1, false, 40, 2.5e6f, 4e6f, 2.4e9f, false, 0);
        auto b = make_stream(MTU, 1, 14, true, {});
        cl_smi_feed_bytes(smi, b.data(), b.size());
        if (one_call(*blk, b, 0, MTU, 1, false, NULL)) return 1;
    }
    cl_smi_close(smi);
    printf("gr source ok\n");
    return 0;
}
