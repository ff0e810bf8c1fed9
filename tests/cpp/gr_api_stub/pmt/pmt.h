// COMPILE-CHECK STUB -- NOT GNU Radio's pmt.  The slice of the API that cariboulite_amd/csrc/gr_source uses, so that the
// adaptor can be compiled and driven in an image without GNU Radio (tests/test_gr_source.py).  Tags are kept as plain values.
#pragma once
#include <memory>
#include <string>

namespace pmt {
struct pmt_base {
    std::string symbol;
    bool boolean = false;
};
typedef std::shared_ptr<pmt_base> pmt_t;
inline pmt_t string_to_symbol(const std::string &s) { auto p = std::make_shared<pmt_base>(); p->symbol = s; return p; }
inline pmt_t from_bool(bool b) { auto p = std::make_shared<pmt_base>(); p->boolean = b; return p; }
}  // namespace pmt
