// GPU test program for the C++ API seam: CaribouLiteRadio (sync + async) vs the oracle.
// Built and run by tests/test_gpu_cpp_api.py; links libcariboulite_cpp + liboracle (checker only).
#include <cassert>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "CaribouLiteHip.hpp"
#include "cl_oracle.h"

#define CHECK(c) do { if (!(c)) { printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

static std::vector<uint8_t> make_stream(size_t n, int hif, uint32_t seed)
{
    std::vector<uint8_t> b(4 * n);
    uint32_t s = seed;
    for (size_t k = 0; k < n; k++) {
        s = s * 1664525u + 1013904223u; uint32_t i13 = (s >> 8) & 0x1FFF;
        s = s * 1664525u + 1013904223u; uint32_t q13 = (s >> 8) & 0x1FFF;
        uint32_t a = hif ? q13 : i13, bb = hif ? i13 : q13;
        uint32_t w = 0x80004000u | (a << 17) | (bb << 1) | ((s >> 30) & 1u);
        memcpy(&b[4 * k], &w, 4);
    }
    return b;
}

int main()
{
    cl_smi *smi = cl_smi_init(0);
    CHECK(smi);
    const size_t MTU = 131072;
    // ---------------- sync API
    {
        CaribouLiteRadio s1g(smi, CaribouLiteRadio::S1G, CaribouLiteRadio::Sync);
        CHECK(s1g.GetNativeMtuSample() == MTU);
        std::vector<std::complex<short>> out(MTU);
        CHECK(s1g.ReadSamples(out.data(), 100) == 0);                 // not receiving yet
        s1g.StartReceiving();
        auto b = make_stream(2 * MTU, 0, 7);
        cl_smi_feed_bytes(smi, b.data(), b.size());
        std::vector<uint8_t> meta(MTU);
        int ret = s1g.ReadSamples(out.data(), MTU, meta.data());
        CHECK(ret == (int)MTU);
        std::vector<int16_t> want(2 * (MTU + 2)); std::vector<uint8_t> wmeta(MTU + 2);
        CHECK(orc_rx_data_analyze(0, b.data(), 4 * MTU, want.data(), wmeta.data()) == 0);
        CHECK(memcmp(out.data(), want.data(), 4 * MTU) == 0 && memcmp(meta.data(), wmeta.data(), MTU) == 0);
        std::vector<std::complex<float>> fo(MTU);
        ret = s1g.ReadSamples(fo.data(), MTU);
        CHECK(ret == (int)MTU);
        CHECK(orc_rx_data_analyze(0, b.data() + 4 * MTU, 4 * MTU, want.data(), NULL) == 0);
        std::vector<float> wf(2 * MTU);
        orc_cs16_to_cf32(want.data(), wf.data(), MTU);
        CHECK(memcmp(fo.data(), wf.data(), 8 * MTU) == 0);            // ((float)v)/4096.0f, bit-exact
        CHECK(s1g.ReadSamples(out.data(), 100) == 0);                 // drained
        // TX: float and short, MTU chunking, (uint16_t)(f*4096) quirk for negatives
        s1g.StartTransmitting();
        std::vector<std::complex<float>> tx(MTU + 1000);
        uint32_t s = 99;
        for (auto &v : tx) { s = s * 1664525u + 1013904223u; float a = ((int)(s >> 8) % 8000 - 4000) / 4096.0f * 0.97f;
                             s = s * 1664525u + 1013904223u; float c = ((int)(s >> 8) % 8000 - 4000) / 4096.0f * 0.97f; v = {a, c}; }
        CHECK(s1g.WriteSamples(tx.data(), tx.size()) == (int)tx.size());
        std::vector<uint8_t> got(4 * tx.size() + 16);
        CHECK(cl_smi_drain_bytes(smi, got.data(), got.size()) == 4 * tx.size());
        std::vector<int16_t> q(2 * tx.size()); std::vector<uint8_t> wb(4 * tx.size());
        orc_cf32_to_cs16((const float *)tx.data(), q.data(), tx.size());
        orc_generate_data(ORC_TX_DOCUMENTED, q.data(), tx.size(), wb.data());
        CHECK(memcmp(got.data(), wb.data(), wb.size()) == 0);
    }
    // ---------------- async API: callbacks from the reader thread, one radio at a time
    {
        CaribouLiteRadio s1g(smi, CaribouLiteRadio::S1G), hif(smi, CaribouLiteRadio::HiF);
        s1g.SetSibling(&hif); hif.SetSibling(&s1g);
        std::mutex mx; std::vector<std::complex<float>> got; std::vector<uint8_t> gmeta; size_t calls = 0;
        auto b = make_stream(3 * 4096, 1, 11);
        // with the pps tags on: inside the callback GetSyncTags() describes the chunk being handed over -- the positions the
        // GNU Radio source's loop over `m` would find (caribouLiteSource_impl.cc:113-119)
        hif.EnableSyncTags(true);
        bool tags_ok = true; size_t tags_seen = 0;
        hif.StartReceiving([&](CaribouLiteRadio *r, const std::complex<float> *d, CaribouLiteMeta *m, size_t n) {
            std::lock_guard<std::mutex> g(mx);
            got.insert(got.end(), d, d + n);
            for (size_t k = 0; k < n; k++) gmeta.push_back(m[k].sync);
            const uint32_t *at = NULL;
            const size_t nt = r->GetSyncTags(&at);
            size_t j = 0;
            for (size_t k = 0; k < n; k++)
                if (m[k].sync == 1) { if (j >= nt || at[j] != k) tags_ok = false; j++; }
            if (j != nt) tags_ok = false;
            tags_seen += nt;
            calls++;
        }, 4096);
        cl_smi_feed_bytes(smi, b.data(), b.size());
        for (int w = 0; w < 500; w++) {
            { std::lock_guard<std::mutex> g(mx); if (got.size() >= 3 * 4096) break; }
            std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
        hif.StopReceiving();
        std::lock_guard<std::mutex> g(mx);
        CHECK(got.size() == 3 * 4096 && calls == 3);
        CHECK(tags_ok && tags_seen > 3 * 4096 / 4);                     // (the test stream's sync bit is random: about half the samples)
        std::vector<int16_t> want(2 * (3 * 4096 + 2)); std::vector<uint8_t> wmeta(3 * 4096 + 2);
        CHECK(orc_rx_data_analyze(1, b.data(), b.size(), want.data(), wmeta.data()) == 0);
        std::vector<float> wf(2 * 3 * 4096);
        orc_cs16_to_cf32(want.data(), wf.data(), 3 * 4096);
        CHECK(memcmp(got.data(), wf.data(), wf.size() * 4) == 0 && memcmp(gmeta.data(), wmeta.data(), 3 * 4096) == 0);
        // starting the sibling stops this one (CaribouLiteRadioCpp.cpp:535-537)
        size_t n_int = 0;
        s1g.StartReceiving([&](CaribouLiteRadio *, const std::complex<short> *, size_t n) { std::lock_guard<std::mutex> g2(mx); n_int += n; }, 0);
    }
    cl_smi_close(smi);
    printf("cpp api ok\n");
    return 0;
}
