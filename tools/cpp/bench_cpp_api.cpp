// PCIe-inclusive cost of one CaribouLiteRadio::ReadSamples call (one MTU): host SMI bytes in (cl_smi_feed_bytes), host samples
// (+ meta, + pps tags) out -- the GNU Radio source's work() minus GNU Radio.  Built and run by tools/bench_cpp_api.py.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#include "CaribouLiteHip.hpp"

int main()
{
    cl_smi *smi = cl_smi_init(0);
    if (!smi) return 1;
    const size_t MTU = 131072, K = 64;
    std::vector<uint8_t> b(4 * MTU * K);
    uint32_t s = 5;
    for (size_t k = 0; k < MTU * K; k++) {
        s = s * 1664525u + 1013904223u;
        const uint32_t w = 0x80004000u | (((s >> 8) & 0x1FFF) << 17) | (((s >> 3) & 0x1FFF) << 1) | (k % 4000000 == 0);
        memcpy(&b[4 * k], &w, 4);
    }
    CaribouLiteRadio r(smi, CaribouLiteRadio::S1G, CaribouLiteRadio::Sync);
    r.StartReceiving();
    std::vector<std::complex<float>> f(MTU);
    std::vector<std::complex<short>> i16(MTU);
    std::vector<uint8_t> meta(MTU);
    printf("{");
    for (int mode = 0; mode < 4; mode++) {        // 0 short, 1 short + meta, 2 float + meta, 3 float + meta + tags
        r.EnableSyncTags(mode == 3);
        double total = 0; size_t calls = 0, tags = 0;
        for (int rep = 0; rep < 3; rep++) {
            cl_smi_feed_bytes(smi, b.data(), b.size());
            const auto t0 = std::chrono::steady_clock::now();
            for (size_t k = 0; k < K; k++) {
                const int got = mode < 2 ? r.ReadSamples(i16.data(), MTU, mode ? meta.data() : NULL) : r.ReadSamples(f.data(), MTU, meta.data());
                if (got != (int)MTU) { printf("short read %d\n", got); return 1; }
                const uint32_t *at; tags += r.GetSyncTags(&at);
            }
            const auto t1 = std::chrono::steady_clock::now();
            if (rep) { total += std::chrono::duration<double>(t1 - t0).count(); calls += K; }
        }
        static const char *names[4] = {"cs16", "cs16_meta", "cf32_meta", "cf32_meta_tags"};
        printf("%s\"%s\": {\"us_per_mtu_call\": %.1f, \"tags\": %zu}", mode ? ", " : "", names[mode], total / calls * 1e6, tags);
    }
    printf("}\n");
    cl_smi_close(smi);
    return 0;
}
