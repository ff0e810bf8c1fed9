// COMPILE-CHECK STUB -- not SoapySDR.  The build image has no SoapySDR headers, so the real adaptor
// (cariboulite_amd/csrc/soapy_module/SoapyCaribouliteHip.cpp) would otherwise never see a compiler.  This file
// declares only the slice of the public SoapySDR 0.8 C++ API the adaptor overrides (names and signatures as
// listed in SURVEY.md section 8b from soapy_api/Cariboulite.hpp:65-93); tests/test_soapy_module.py builds the
// adaptor against it and drives it through these virtuals.  Where SoapySDR is installed, use its own headers.
#pragma once
#include <cstddef>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

namespace SoapySDR {
typedef std::map<std::string, std::string> Kwargs;
typedef std::vector<Kwargs> KwargsList;
struct ArgInfo { std::string key, value, name, description; };
typedef std::vector<ArgInfo> ArgInfoList;
class Stream;

class Device {
public:
    virtual ~Device() {}
    virtual std::string getDriverKey() const { return ""; }
    virtual size_t getNumChannels(const int) const { return 0; }
    virtual bool getFullDuplex(const int, const size_t) const { return false; }
    virtual std::vector<std::string> getStreamFormats(const int, const size_t) const { return std::vector<std::string>(); }
    virtual std::string getNativeStreamFormat(const int, const size_t, double &fullScale) const { fullScale = 0; return ""; }
    virtual ArgInfoList getStreamArgsInfo(const int, const size_t) const { return ArgInfoList(); }
    virtual Stream *setupStream(const int, const std::string &, const std::vector<size_t> & = std::vector<size_t>(),
                                const Kwargs & = Kwargs()) { return nullptr; }
    virtual void closeStream(Stream *) {}
    virtual size_t getStreamMTU(Stream *) const { return 0; }
    virtual int activateStream(Stream *, const int = 0, const long long = 0, const size_t = 0) { return -5; }
    virtual int deactivateStream(Stream *, const int = 0, const long long = 0) { return -5; }
    virtual int readStream(Stream *, void *const *, const size_t, int &, long long &, const long = 100000) { return -5; }
    virtual int writeStream(Stream *, const void *const *, const size_t, int &, const long long = 0, const long = 100000) { return -5; }
    virtual void setBandwidth(const int, const size_t, const double) {}
    virtual void writeSetting(const std::string &, const std::string &) {}
    virtual std::string readSetting(const std::string &) const { return ""; }
};
}   // namespace SoapySDR
