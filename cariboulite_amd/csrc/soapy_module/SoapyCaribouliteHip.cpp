// SoapyCaribouliteHip.cpp -- the real SoapySDR device module over the C-ABI.
//
// Compiled ONLY where SoapySDR's headers exist (they are absent from the build
// image and from /root/reference: SoapySDR is a system package there too,
// software/libcariboulite/README.md:15).  It registers under the same key the
// reference registers (soapy_api/SoapyCariboulite.cpp:119) and forwards every
// stream virtual of soapy_api/Cariboulite.hpp:65-93 to include/cariboulite_hip.h:
//
//   g++ -std=c++11 -fPIC -shared SoapyCaribouliteHip.cpp -I../../../include -L../.. -lcariboulite_host
//       -lcariboulite_hip -lSoapySDR -o libSoapyCaribouliteHip.so
//
// tests/test_soapy_module.py compiles it against a compile-check stub of the API slice used here
// (tests/cpp/soapy_api_stub) and drives it through the Device virtuals on the GPU box.
#include <SoapySDR/Device.hpp>
#include <SoapySDR/Formats.hpp>
#include <SoapySDR/Registry.hpp>

#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "cariboulite_hip.h"

namespace {

struct KwArrays {
    std::vector<const char *> k, v;
    explicit KwArrays(const SoapySDR::Kwargs &a)
    {
        for (const auto &it : a) { k.push_back(it.first.c_str()); v.push_back(it.second.c_str()); }
    }
};

class CaribouliteHip : public SoapySDR::Device {
    cl_device *dev_;

public:
    explicit CaribouliteHip(const SoapySDR::Kwargs &args)
    {
        KwArrays a(args);
        dev_ = cl_device_make(a.k.data(), a.v.data(), a.k.size());
        if (!dev_) throw std::runtime_error("Channel type is not specified correctly");   // Cariboulite.cpp:27
    }
    ~CaribouliteHip() override { cl_device_unmake(dev_); }

    std::string getDriverKey() const override { return "Cariboulite"; }
    size_t getNumChannels(const int) const override { return 1; }                          // Cariboulite.hpp:59
    bool getFullDuplex(const int, const size_t) const override { return false; }           // :60

    std::vector<std::string> getStreamFormats(const int dir, const size_t ch) const override
    {
        const char *f[8];
        const size_t n = cl_getStreamFormats(dev_, dir, ch, f, 8);
        return std::vector<std::string>(f, f + n);
    }
    std::string getNativeStreamFormat(const int dir, const size_t ch, double &fullScale) const override
    {
        return cl_getNativeStreamFormat(dev_, dir, ch, &fullScale);
    }
    SoapySDR::ArgInfoList getStreamArgsInfo(const int, const size_t) const override { return SoapySDR::ArgInfoList(); }

    SoapySDR::Stream *setupStream(const int dir, const std::string &format, const std::vector<size_t> &channels,
                                  const SoapySDR::Kwargs &args) override
    {
        KwArrays a(args);
        cl_stream *s = cl_setupStream(dev_, dir, format.c_str(), channels.data(), channels.size(), a.k.data(),
                                      a.v.data(), a.k.size());
        if (!s) throw std::runtime_error(cl_device_last_error(dev_));                      // StreamFunctions.cpp:114
        return reinterpret_cast<SoapySDR::Stream *>(s);
    }
    void closeStream(SoapySDR::Stream *s) override { cl_closeStream(dev_, reinterpret_cast<cl_stream *>(s)); }
    size_t getStreamMTU(SoapySDR::Stream *s) const override { return cl_getStreamMTU(dev_, reinterpret_cast<cl_stream *>(s)); }
    int activateStream(SoapySDR::Stream *s, const int flags, const long long timeNs, const size_t numElems) override
    {
        return cl_activateStream(dev_, reinterpret_cast<cl_stream *>(s), flags, timeNs, numElems);
    }
    int deactivateStream(SoapySDR::Stream *s, const int flags, const long long timeNs) override
    {
        return cl_deactivateStream(dev_, reinterpret_cast<cl_stream *>(s), flags, timeNs);
    }
    int readStream(SoapySDR::Stream *s, void *const *buffs, const size_t numElems, int &flags, long long &timeNs,
                   const long timeoutUs) override
    {
        return cl_readStream(dev_, reinterpret_cast<cl_stream *>(s), buffs, numElems, &flags, &timeNs, timeoutUs);
    }
    int writeStream(SoapySDR::Stream *s, const void *const *buffs, const size_t numElems, int &flags,
                    const long long timeNs, const long timeoutUs) override
    {
        return cl_writeStream(dev_, reinterpret_cast<cl_stream *>(s), buffs, numElems, &flags, timeNs, timeoutUs);
    }
    void setBandwidth(const int dir, const size_t ch, const double bw) override { cl_setBandwidth(dev_, dir, ch, bw); }

    // the /dev/smi replacement, reachable through Soapy's generic settings hook
    void writeSetting(const std::string &key, const std::string &value) override
    {
        if (key == "SMI_FEED_PTR") {   // "<host pointer>:<bytes>" handed over by a co-located feeder
            unsigned long long p = 0, n = 0;
            if (sscanf(value.c_str(), "%llx:%llu", &p, &n) == 2)
                cl_smi_feed_bytes(cl_device_smi(dev_), reinterpret_cast<const uint8_t *>(p), (size_t)n);
        }
    }
};

SoapySDR::KwargsList findCaribouliteHip(const SoapySDR::Kwargs &args)
{
    // one Soapy device per channel type, like the reference (SoapyCariboulite.cpp:46-69)
    SoapySDR::KwargsList out;
    for (const char *ch : {"S1G", "HiF"}) {
        if (args.count("channel") && args.at("channel") != ch) continue;
        SoapySDR::Kwargs d;
        d["driver"] = "Cariboulite"; d["channel"] = ch; d["device_id"] = "0";
        d["label"] = std::string("CaribouLite-HIP ") + ch;
        out.push_back(d);
    }
    return out;
}

SoapySDR::Device *makeCaribouliteHip(const SoapySDR::Kwargs &args) { return new CaribouliteHip(args); }

SoapySDR::Registry registerCaribouliteHip("Cariboulite", &findCaribouliteHip, &makeCaribouliteHip, SOAPY_SDR_ABI_VERSION);

}   // namespace
