// caribouLiteSourceHip.cc -- see caribouLiteSourceHip.h.  Follows software/gr-caribouLite/lib/caribouLiteSource_impl.cc
// (lines cited); what differs is where the work happens: the int13 unpack, the /4096 conversion and the search of the meta
// plane for pps markers all run on the GPU, and work() touches only the few tag positions that come back.
#include "caribouLiteSourceHip.h"

#include <gnuradio/io_signature.h>

namespace gr {
namespace caribouLite {

caribouLiteSourceHip::sptr caribouLiteSourceHip::make(cl_smi *smi, int channel, bool enable_agc, float rx_gain, float rx_bw,
                                                      float sample_rate, float freq, bool provide_meta, uint8_t pmod_state)
{
    return gnuradio::make_block_sptr<caribouLiteSourceHip>(smi, channel, enable_agc, rx_gain, rx_bw, sample_rate, freq, provide_meta,
                                                           pmod_state);
}

// :56-95 -- one or two outputs: the samples, and one meta byte per sample when asked for
caribouLiteSourceHip::caribouLiteSourceHip(cl_smi *smi, int channel, bool enable_agc, float rx_gain, float rx_bw, float sample_rate,
                                           float freq, bool provide_meta, uint8_t /* pmod_state: board control */)
    : gr::sync_block("caribouLiteSource", gr::io_signature::make(0, 0, 0),
                     gr::io_signature::make2(1, 2, sizeof(gr_complex), sizeof(uint8_t))),
      _channel((CaribouLiteRadio::RadioType)channel), _enable_agc(enable_agc), _rx_gain(rx_gain), _rx_bw(rx_bw),
      _sample_rate(sample_rate), _frequency(freq), _provide_meta(provide_meta),
      _radio(new CaribouLiteRadio(smi, _channel, CaribouLiteRadio::Sync))           // work() reads synchronously (:109)
{
    _mtu_size = _radio->GetNativeMtuSample();
    _radio->EnableSyncTags(_provide_meta);      // the tag loop of :113-119 runs where the meta plane is written
    _radio->StartReceiving();                   // :94 "do the thing"
}

caribouLiteSourceHip::~caribouLiteSourceHip() { _radio->StopReceiving(); }           // :100-103

// :106-123
int caribouLiteSourceHip::work(int noutput_items, gr_vector_const_void_star & /* a source: no inputs */, gr_vector_void_star &output_items)
{
    auto out_samples = static_cast<gr_complex *>(output_items[0]);
    auto out_meta = _provide_meta ? static_cast<uint8_t *>(output_items[1]) : (uint8_t *)NULL;
    const int read_samples = _radio->ReadSamples(out_samples, static_cast<size_t>(noutput_items), out_meta);
    if (read_samples <= 0) return 0;

    if (_provide_meta) {
        // the reference walks out_meta[0 .. read_samples) here; the same positions, in the same order, arrive ready-made.
        // The offset is passed as the reference passes it (:116): the index inside this call's output.
        static const pmt::pmt_t key = pmt::string_to_symbol("pps");
        const uint32_t *at = NULL;
        const size_t n_tags = _radio->GetSyncTags(&at);
        for (size_t k = 0; k < n_tags; k++) add_item_tag(0, at[k], key, pmt::from_bool(true));
    }
    return read_samples;
}

}  // namespace caribouLite
}  // namespace gr
