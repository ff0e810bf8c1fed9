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

// ---------------------------------------------------------------------------------------------------------------------
// Many boards as ONE multi-channel Soapy device: driver=CaribouliteGroup, channels="S1G,HiF,S1G,..." (one entry per
// underlying Cariboulite device, i.e. per board and channel type); gpus="0,1,...,7" spreads them over the GPUs of the node, channel i on
// gpus[i mod N] (independent streams, no data-path collective: cl_node runs one cl_group per GPU, all at once); without it they all
// live on the GPU the `gpu` kwarg names (default 0).  SoapySDR's own multi-channel stream shape --
// setupStream(dir, fmt, {0 .. N-1}) and readStream(stream, buffs[N], numElems, ...) -- which the reference declines (one channel per
// device, Cariboulite.hpp:59) maps one-to-one onto cl_node_readStream / cl_node_writeStream (= cl_group_* per GPU): every channel is exactly the
// reference's device for that board, all of them read (or written) in one call.  readStream returns the LARGEST count any channel
// delivered (SoapySDR has one return value for all channels); readSetting("GROUP_RETS") gives the last call's count per channel
// ("131072,131072,0,..."): a channel that re-synchronised or timed out says so there, exactly as its own readStream would have.
class CaribouliteGroupHip : public SoapySDR::Device {
    std::vector<cl_device *> devs_;
    std::vector<cl_stream *> streams_;
    cl_node *grp_ = nullptr;
    SoapySDR::Kwargs group_args_;
    mutable std::vector<int> rets_;

    void drop_group() { if (grp_) { cl_node_unmake(grp_); grp_ = nullptr; } }

public:
    explicit CaribouliteGroupHip(const SoapySDR::Kwargs &args)
    {
        const std::string list = args.count("channels") ? args.at("channels") : "";
        std::vector<std::string> gpus;
        if (args.count("gpus")) {
            const std::string g = args.at("gpus");
            for (size_t p = 0; p <= g.size();) {
                const size_t c = g.find(',', p);
                gpus.push_back(g.substr(p, c == std::string::npos ? std::string::npos : c - p));
                if (c == std::string::npos) break;
                p = c + 1;
            }
        }
        size_t pos = 0;
        while (pos <= list.size() && !list.empty()) {
            const size_t comma = list.find(',', pos);
            const std::string ch = list.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
            SoapySDR::Kwargs one = args;
            one.erase("channels"); one.erase("gpus"); one["driver"] = "Cariboulite"; one["channel"] = ch;
            if (!gpus.empty()) one["gpu"] = gpus[devs_.size() % gpus.size()];
            KwArrays a(one);
            cl_device *d = cl_device_make(a.k.data(), a.v.data(), a.k.size());
            if (!d) { for (cl_device *x : devs_) cl_device_unmake(x); throw std::runtime_error("CaribouliteGroup: channels=\"S1G,HiF,...\""); }
            devs_.push_back(d);
            if (comma == std::string::npos) break;
            pos = comma + 1;
        }
        if (devs_.empty()) throw std::runtime_error("CaribouliteGroup: channels=\"S1G,HiF,...\"");
        for (const auto &it : args)                          // group kwargs (SUBBATCH, COPY_THREADS, READAHEAD, ...) pass through
            if (it.first != "driver" && it.first != "channels" && it.first != "gpu" && it.first != "gpus") group_args_[it.first] = it.second;
        rets_.assign(devs_.size(), 0);
    }
    ~CaribouliteGroupHip() override
    {
        drop_group();                                        // (before its members: the group holds their seams)
        for (cl_device *d : devs_) cl_device_unmake(d);
    }

    std::string getDriverKey() const override { return "CaribouliteGroup"; }
    size_t getNumChannels(const int) const override { return devs_.size(); }
    bool getFullDuplex(const int, const size_t) const override { return false; }
    std::vector<std::string> getStreamFormats(const int dir, const size_t ch) const override
    {
        const char *f[8];
        const size_t n = cl_getStreamFormats(devs_.at(ch), dir, 0, f, 8);
        return std::vector<std::string>(f, f + n);
    }
    std::string getNativeStreamFormat(const int dir, const size_t ch, double &fullScale) const override
    {
        return cl_getNativeStreamFormat(devs_.at(ch), dir, 0, &fullScale);
    }

    // all channels, in order (an empty list means all): the group is made here, behind the members' setupStream
    SoapySDR::Stream *setupStream(const int dir, const std::string &format, const std::vector<size_t> &channels,
                                  const SoapySDR::Kwargs &args) override
    {
        if (!channels.empty()) {
            if (channels.size() != devs_.size()) throw std::runtime_error("CaribouliteGroup: a stream spans all channels");
            for (size_t i = 0; i < channels.size(); i++)
                if (channels[i] != i) throw std::runtime_error("CaribouliteGroup: a stream spans all channels, in order");
        }
        drop_group();
        KwArrays a(args);
        streams_.clear();
        const size_t zero = 0;
        for (cl_device *d : devs_) {
            cl_stream *s = cl_setupStream(d, dir, format.c_str(), &zero, 1, a.k.data(), a.v.data(), a.k.size());
            if (!s) throw std::runtime_error(cl_device_last_error(d));
            streams_.push_back(s);
        }
        KwArrays ga(group_args_);
        grp_ = cl_node_make(devs_.data(), devs_.size(), ga.k.data(), ga.v.data(), ga.k.size());
        if (!grp_) throw std::runtime_error(cl_node_last_error(nullptr));
        return reinterpret_cast<SoapySDR::Stream *>(grp_);
    }
    void closeStream(SoapySDR::Stream *) override
    {
        for (size_t i = 0; i < streams_.size(); i++) cl_closeStream(devs_[i], streams_[i]);
    }
    size_t getStreamMTU(SoapySDR::Stream *) const override { return streams_.empty() ? 0 : cl_getStreamMTU(devs_[0], streams_[0]); }
    int activateStream(SoapySDR::Stream *, const int flags, const long long timeNs, const size_t numElems) override
    {
        int rc = 0;
        for (size_t i = 0; i < streams_.size(); i++) rc |= cl_activateStream(devs_[i], streams_[i], flags, timeNs, numElems);
        return rc;
    }
    int deactivateStream(SoapySDR::Stream *, const int flags, const long long timeNs) override
    {
        int rc = 0;
        for (size_t i = 0; i < streams_.size(); i++) rc |= cl_deactivateStream(devs_[i], streams_[i], flags, timeNs);
        return rc;
    }
    int readStream(SoapySDR::Stream *, void *const *buffs, const size_t numElems, int &, long long &, const long timeoutUs) override
    {
        if (!grp_) return -5;
        if (cl_node_readStream(grp_, buffs, numElems, rets_.data(), timeoutUs) < 0) return -1;      // SOAPY_SDR_TIMEOUT where the runtime failed
        int most = 0, wrong = 0;
        for (int r : rets_) { if (r > most) most = r; if (r < 0) wrong = r; }
        return wrong ? wrong : most;                         // (NOT_SUPPORTED: a group set up for TX)
    }
    int writeStream(SoapySDR::Stream *, const void *const *buffs, const size_t numElems, int &, const long long, const long timeoutUs) override
    {
        if (!grp_) return -5;
        if (cl_node_writeStream(grp_, buffs, numElems, rets_.data(), timeoutUs) < 0) return -1;
        int most = 0, wrong = 0;
        for (int r : rets_) { if (r > most) most = r; if (r < 0) wrong = r; }
        return wrong ? wrong : most;
    }
    void setBandwidth(const int dir, const size_t ch, const double bw) override { cl_setBandwidth(devs_.at(ch), dir, 0, bw); }

    std::string readSetting(const std::string &key) const override
    {
        std::string out;
        if (key == "GROUP_RETS")
            for (size_t i = 0; i < rets_.size(); i++) out += (i ? "," : "") + std::to_string(rets_[i]);
        if (key == "GROUP_SHARDS") out = std::to_string(grp_ ? cl_node_shards(grp_) : 0);      // groups the calls run at once (GPUs x SHARDS)
        return out;
    }
    void writeSetting(const std::string &key, const std::string &value) override
    {
        // "SMI_FEED_PTR:<channel>" = "<host pointer>:<bytes>" (a co-located feeder per board)
        if (key.compare(0, 13, "SMI_FEED_PTR:") == 0) {
            const size_t ch = (size_t)std::stoul(key.substr(13));
            unsigned long long p = 0, n = 0;
            if (ch < devs_.size() && sscanf(value.c_str(), "%llx:%llu", &p, &n) == 2)
                cl_smi_feed_bytes(cl_device_smi(devs_[ch]), reinterpret_cast<const uint8_t *>(p), (size_t)n);
        }
    }
};

SoapySDR::KwargsList findCaribouliteGroupHip(const SoapySDR::Kwargs &args)
{
    SoapySDR::KwargsList out;
    if (args.count("channels")) {                            // (nothing to enumerate: the client says which boards make the group)
        SoapySDR::Kwargs d = args;
        d["driver"] = "CaribouliteGroup"; d["label"] = "CaribouLite-HIP group of " + args.at("channels");
        out.push_back(d);
    }
    return out;
}

SoapySDR::Device *makeCaribouliteGroupHip(const SoapySDR::Kwargs &args) { return new CaribouliteGroupHip(args); }

SoapySDR::Registry registerCaribouliteHip("Cariboulite", &findCaribouliteHip, &makeCaribouliteHip, SOAPY_SDR_ABI_VERSION);
SoapySDR::Registry registerCaribouliteGroupHip("CaribouliteGroup", &findCaribouliteGroupHip, &makeCaribouliteGroupHip, SOAPY_SDR_ABI_VERSION);

}   // namespace
