// Drives cariboulite_amd/csrc/soapy_module/SoapyCaribouliteHip.cpp through the SoapySDR::Device virtuals
// (compile-check stub of the API under tests/cpp/soapy_api_stub): find, make, setupStream, feed through the
// settings hook, readStream; the decoded samples are the SURVEY.md section 8c known answers of the compiled reference.
#include <SoapySDR/Registry.hpp>
#include <SoapySDR/Formats.hpp>

#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../cariboulite_amd/csrc/soapy_module/SoapyCaribouliteHip.cpp"

int main()
{
    auto &tab = SoapySDR::registryTable();
    if (tab.size() != 2 || tab[0].name != "Cariboulite" || tab[1].name != "CaribouliteGroup") { printf("FAIL registry\n"); return 1; }
    SoapySDR::Kwargs q;
    if (tab[0].find(q).size() != 2) { printf("FAIL find\n"); return 1; }
    q["channel"] = "S1G";
    if (tab[0].find(q).size() != 1) { printf("FAIL find filter\n"); return 1; }
    SoapySDR::Device *dev = tab[0].make(q);
    double fs = 0;
    if (dev->getNativeStreamFormat(SOAPY_SDR_RX, 0, fs) != "CS16" || fs != 4095.0) { printf("FAIL native format\n"); return 1; }
    if (dev->getStreamFormats(SOAPY_SDR_RX, 0).size() != 4 || dev->getNumChannels(SOAPY_SDR_RX) != 1) { printf("FAIL formats\n"); return 1; }
    bool threw = false;
    try { dev->setupStream(SOAPY_SDR_RX, "CU4"); } catch (const std::runtime_error &) { threw = true; }
    if (!threw) { printf("FAIL unknown format must throw\n"); return 1; }
    SoapySDR::Stream *rx = dev->setupStream(SOAPY_SDR_RX, SOAPY_SDR_CS16);
    if (dev->getStreamMTU(rx) != 131072) { printf("FAIL mtu\n"); return 1; }
    dev->activateStream(rx);
    // 8 known-answer words, then zeros up to one native batch
    static const uint32_t kat[8] = {0x80004000u, 0x80027FFFu, 0xBFFE4002u, 0x9FFE6001u, 0xA0005FFEu, 0x89A45E3Fu, 0xB65C61C2u, 0x80C87F39u};
    static const int16_t want[8][2] = {{0, 0}, {1, -1}, {-1, 1}, {4095, -4096}, {-4096, 4095}, {1234, 3871}, {-1234, -3871}, {100, -100}};
    static uint32_t words[131072];
    for (int i = 0; i < 131072; i++) words[i] = 0x80004000u;
    memcpy(words, kat, sizeof kat);
    char val[64];
    snprintf(val, sizeof val, "%llx:%llu", (unsigned long long)(uintptr_t)words, (unsigned long long)sizeof words);
    dev->writeSetting("SMI_FEED_PTR", val);
    static int16_t buf[131072][2];
    void *buffs[1] = {buf};
    int flags = 0; long long t = 0;
    const int n = dev->readStream(rx, buffs, 131072, flags, t, 100000);
    if (n != 131072) { printf("FAIL readStream returned %d\n", n); return 1; }
    for (int i = 0; i < 8; i++)
        if (buf[i][0] != want[i][0] || buf[i][1] != want[i][1]) { printf("FAIL sample %d: %d %d\n", i, buf[i][0], buf[i][1]); return 1; }
    const void *cb[1] = {buf};
    if (dev->writeStream(rx, cb, 16, flags) != -5) { printf("FAIL direction check\n"); return 1; }
    dev->deactivateStream(rx);
    dev->closeStream(rx);
    delete dev;
    // ---- three boards' worth of devices as ONE multi-channel device: SoapySDR's own readStream(stream, buffs[N]) shape
    SoapySDR::Kwargs gq;
    if (!tab[1].find(gq).empty()) { printf("FAIL group find without channels\n"); return 1; }
    gq["channels"] = "S1G,HiF,S1G";
    if (tab[1].find(gq).size() != 1) { printf("FAIL group find\n"); return 1; }
    SoapySDR::Device *grp = tab[1].make(gq);
    if (grp->getNumChannels(SOAPY_SDR_RX) != 3 || grp->getDriverKey() != "CaribouliteGroup") { printf("FAIL group channels\n"); return 1; }
    threw = false;
    try { grp->setupStream(SOAPY_SDR_RX, SOAPY_SDR_CS16, std::vector<size_t>{0, 2}); } catch (const std::runtime_error &) { threw = true; }
    if (!threw) { printf("FAIL a group stream spans all channels\n"); return 1; }
    SoapySDR::Stream *gs = grp->setupStream(SOAPY_SDR_RX, SOAPY_SDR_CS16, std::vector<size_t>{0, 1, 2});
    grp->activateStream(gs);
    for (int c = 0; c < 2; c++) {                                   // channels 0 and 1 get a batch, channel 2 nothing
        snprintf(val, sizeof val, "%llx:%llu", (unsigned long long)(uintptr_t)words, (unsigned long long)sizeof words);
        char key[32]; snprintf(key, sizeof key, "SMI_FEED_PTR:%d", c);
        grp->writeSetting(key, val);
    }
    static int16_t gbuf[3][131072][2];
    void *gb[3] = {gbuf[0], gbuf[1], gbuf[2]};
    const int gn = grp->readStream(gs, gb, 131072, flags, t, 100000);
    if (gn != 131072 || grp->readSetting("GROUP_RETS") != "131072,131072,0") { printf("FAIL group readStream %d [%s]\n", gn, grp->readSetting("GROUP_RETS").c_str()); return 1; }
    for (int i = 0; i < 8; i++) {
        if (gbuf[0][i][0] != want[i][0] || gbuf[0][i][1] != want[i][1]) { printf("FAIL group S1G sample %d\n", i); return 1; }
        if (gbuf[1][i][0] != want[i][1] || gbuf[1][i][1] != want[i][0]) { printf("FAIL group HiF sample %d (I and Q change fields, caribou_smi.c:361-378)\n", i); return 1; }
    }
    grp->deactivateStream(gs);
    delete grp;
    // ---- the same over the node layer's several-groups-at-once shape: gpus="0" + SHARDS=2 (two groups on GPU 0, their calls on two threads)
    gq["gpus"] = "0"; gq["SHARDS"] = "2";
    grp = tab[1].make(gq);
    gs = grp->setupStream(SOAPY_SDR_RX, SOAPY_SDR_CS16, std::vector<size_t>{0, 1, 2});
    grp->activateStream(gs);
    if (grp->readSetting("GROUP_SHARDS") != "2") { printf("FAIL group shards [%s]\n", grp->readSetting("GROUP_SHARDS").c_str()); return 1; }
    for (int c = 1; c < 3; c++) {                                   // channels 1 and 2 (one in each shard) get a batch
        snprintf(val, sizeof val, "%llx:%llu", (unsigned long long)(uintptr_t)words, (unsigned long long)sizeof words);
        char key[32]; snprintf(key, sizeof key, "SMI_FEED_PTR:%d", c);
        grp->writeSetting(key, val);
    }
    memset(gbuf, 0, sizeof gbuf);
    const int gn2 = grp->readStream(gs, gb, 131072, flags, t, 100000);
    if (gn2 != 131072 || grp->readSetting("GROUP_RETS") != "0,131072,131072") { printf("FAIL sharded group readStream %d [%s]\n", gn2, grp->readSetting("GROUP_RETS").c_str()); return 1; }
    for (int i = 0; i < 8; i++) {
        if (gbuf[2][i][0] != want[i][0] || gbuf[2][i][1] != want[i][1]) { printf("FAIL sharded group S1G sample %d\n", i); return 1; }
        if (gbuf[1][i][0] != want[i][1] || gbuf[1][i][1] != want[i][0]) { printf("FAIL sharded group HiF sample %d\n", i); return 1; }
    }
    grp->deactivateStream(gs);
    delete grp;
    printf("OK soapy module: find/make/setup/feed/read through the Device virtuals\n");
    return 0;
}
