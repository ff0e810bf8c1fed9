// COMPILE-CHECK STUB -- NOT GNU Radio (see pmt/pmt.h).  gr::sync_block as far as a source block's work() needs it;
// add_item_tag records what it is given so that a test can read the tags back.
#pragma once
#include <complex>
#include <cstdint>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include <gnuradio/io_signature.h>
#include <pmt/pmt.h>

typedef std::complex<float> gr_complex;
typedef std::vector<const void *> gr_vector_const_void_star;
typedef std::vector<void *> gr_vector_void_star;

namespace gr {
struct tag_t {
    uint64_t offset;
    pmt::pmt_t key, value;
};

class sync_block {
public:
    virtual ~sync_block() {}
    virtual int work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items) = 0;
    const std::string &name() const { return name_; }
    io_signature::sptr output_signature() const { return out_; }
    std::vector<tag_t> stub_tags;                   // what add_item_tag was given, in call order

protected:
    sync_block(const std::string &name, io_signature::sptr in, io_signature::sptr out) : name_(name), in_(in), out_(out) {}
    void add_item_tag(unsigned int which_output, uint64_t abs_offset, const pmt::pmt_t &key, const pmt::pmt_t &value)
    {
        (void)which_output;
        stub_tags.push_back(tag_t{abs_offset, key, value});
    }

private:
    std::string name_;
    io_signature::sptr in_, out_;
};
}  // namespace gr

namespace gnuradio {
template <class T, class... Args> std::shared_ptr<T> make_block_sptr(Args &&...args) { return std::make_shared<T>(std::forward<Args>(args)...); }
}  // namespace gnuradio
