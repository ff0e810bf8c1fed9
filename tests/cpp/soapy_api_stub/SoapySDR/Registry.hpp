// COMPILE-CHECK STUB -- see Device.hpp in this directory.  Registry keeps the registered entry points in a
// process-wide table so that the test can find and call them the way SoapySDR::Device::make() would.
#pragma once
#include "Device.hpp"
#define SOAPY_SDR_ABI_VERSION "0.8-stub"

namespace SoapySDR {
typedef KwargsList (*FindFunction)(const Kwargs &);
typedef Device *(*MakeFunction)(const Kwargs &);
struct RegistryEntry { std::string name, abi; FindFunction find; MakeFunction make; };
inline std::vector<RegistryEntry> &registryTable()
{
    static std::vector<RegistryEntry> t;
    return t;
}
class Registry {
public:
    Registry(const std::string &name, const FindFunction &find, const MakeFunction &make, const std::string &abi)
    {
        registryTable().push_back(RegistryEntry{name, abi, find, make});
    }
};
}   // namespace SoapySDR
