// caribouLiteSourceHip.h -- the GNU Radio source block over the GPU path (the caller SURVEY.md section 8(f) rank 1 names:
// software/gr-caribouLite/lib/caribouLiteSource_impl.{h,cc}).
//
// Compiled ONLY where GNU Radio's headers exist (absent from the build image and from /root/reference, where gnuradio is a
// system package too): tests/test_gr_source.py compiles it against a compile-check stub of the API slice used here
// (tests/cpp/gr_api_stub) and drives work() on the GPU box.
//
//   g++ -std=c++17 -fPIC -shared caribouLiteSourceHip.cc -I../cpp_api -I../../../include -L../.. -lcariboulite_cpp
//       -lcariboulite_host -lcariboulite_hip $(pkg-config --cflags --libs gnuradio-runtime) -o libgnuradio-caribouLiteHip.so
//
// Same block name, output signature (gr_complex samples + optional uint8 meta stream), constructor arguments and work()
// contract as caribouLiteSource_impl; the hardware session of the reference's CaribouLite singleton is the `cl_smi *` the
// bytes are fed through (cl_smi_feed_bytes / cl_smi_feed_fd), and modem control (gain, bandwidth, frequency) is out of scope
// (SURVEY.md section 2), so those arguments are kept and not acted on.
#pragma once
#include <gnuradio/sync_block.h>

#include <memory>

#include "CaribouLiteHip.hpp"

namespace gr {
namespace caribouLite {

class caribouLiteSourceHip : public gr::sync_block
{
private:
    CaribouLiteRadio::RadioType _channel;        // RadioType, not a frequency (caribouLiteSource_impl.h:23)
    bool _enable_agc;
    float _rx_gain, _rx_bw, _sample_rate, _frequency;
    size_t _mtu_size;
    bool _provide_meta;
    std::unique_ptr<CaribouLiteRadio> _radio;

public:
    typedef std::shared_ptr<caribouLiteSourceHip> sptr;
    static sptr make(cl_smi *smi, int channel = 0, bool enable_agc = false, float rx_gain = 40, float rx_bw = 2500000,
                     float sample_rate = 4000000, float freq = 900000000, bool provide_meta = false, uint8_t pmod_state = 0);

    caribouLiteSourceHip(cl_smi *smi, int channel, bool enable_agc, float rx_gain, float rx_bw, float sample_rate, float freq,
                         bool provide_meta, uint8_t pmod_state);
    ~caribouLiteSourceHip() override;

    int work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items) override;
};

}  // namespace caribouLite
}  // namespace gr
