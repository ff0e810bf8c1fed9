// CaribouLiteHip.hpp -- the sample-path half of the reference's C++ API (CaribouLite.hpp:87-262,
// CaribouLiteRadioCpp.cpp:5-169, 526-646) over the MI355X host layer (include/cariboulite_hip.h).
// Same class / method names, argument meaning and return values for:
//   ReadSamples (complex<short> | complex<float>), WriteSamples (both), StartReceiving (5 overloads,
//   callback thread), StopReceiving, StartTransmitting, StopTransmitting, GetNativeMtuSample, FlushBuffers.
// Hardware control (gain, frequency, bandwidth, RSSI ...) is out of scope.
// the second production caller of the radio C seam, used by the
// GNU Radio block (software/gr-caribouLite/lib/caribouLiteSource_impl.cc:104-121) and the cpp examples.
#pragma once
#include <atomic>
#include <complex>
#include <cstdint>
#include <functional>
#include <string>
#include <thread>

#include "cariboulite_hip.h"

#pragma pack(push, 1)
struct CaribouLiteMeta { uint8_t sync; };          // CaribouLite.hpp: cariboulite_sample_meta alias
#pragma pack(pop)

class CaribouLiteRadio {
public:
    enum RadioType { S1G = 0, HiF = 1 };
    enum RxCbType { None = 0, FloatSync = 1, Float = 2, IntSync = 3, Int = 4 };
    enum ApiType { Async = 0, Sync = 1 };

    // `smi` stands where the CaribouLite singleton's hardware session stands; bytes reach it through
    // cl_smi_feed_bytes (the /dev/smi replacement).  `other` = the sibling channel's radio (only one
    // radio receives at a time, CaribouLiteRadioCpp.cpp:535-537); may be set later.
    CaribouLiteRadio(cl_smi *smi, RadioType type, ApiType api_type = Async);
    virtual ~CaribouLiteRadio();
    void SetSibling(CaribouLiteRadio *other) { _other = other; }

    // Activation
    void StartReceiving(std::function<void(CaribouLiteRadio *, const std::complex<float> *, CaribouLiteMeta *, size_t)> on_data_ready, size_t samples_per_chunk = 0);
    void StartReceiving(std::function<void(CaribouLiteRadio *, const std::complex<float> *, size_t)> on_data_ready, size_t samples_per_chunk = 0);
    void StartReceiving(std::function<void(CaribouLiteRadio *, const std::complex<short> *, CaribouLiteMeta *, size_t)> on_data_ready, size_t samples_per_chunk = 0);
    void StartReceiving(std::function<void(CaribouLiteRadio *, const std::complex<short> *, size_t)> on_data_ready, size_t samples_per_chunk = 0);
    void StartReceiving();
    void StopReceiving();
    void StartTransmitting();
    void StopTransmitting();

    // Synchronous reading and writing
    int ReadSamples(std::complex<float> *samples, size_t num_to_read, uint8_t *meta = NULL);
    int ReadSamples(std::complex<short> *samples, size_t num_to_read, uint8_t *meta = NULL);
    int WriteSamples(std::complex<float> *samples, size_t num_to_write);
    int WriteSamples(std::complex<short> *samples, size_t num_to_write);

    size_t GetNativeMtuSample();
    std::string GetRadioName();
    void FlushBuffers();

private:
    static void CaribouLiteRxThread(CaribouLiteRadio *radio);
    void StartReceivingInternal(size_t samples_per_chunk);

    cl_smi *_smi;
    cl_radio *_radio;
    CaribouLiteRadio *_other = NULL;
    RadioType _type;
    RxCbType _rxCallbackType;
    ApiType _api_type;
    std::atomic<bool> _rx_thread_running{false}, _rx_is_active{false}, _tx_is_active{false};
    std::thread *_rx_thread = NULL;
    size_t _rx_samples_per_chunk = 0;
    cl_sample_complex_int16 *_read_samples = NULL;
    cl_sample_meta *_read_metadata = NULL;
    std::complex<short> *_write_samples = NULL;
    std::function<void(CaribouLiteRadio *, const std::complex<float> *, CaribouLiteMeta *, size_t)> _on_data_ready_fm;
    std::function<void(CaribouLiteRadio *, const std::complex<float> *, size_t)> _on_data_ready_f;
    std::function<void(CaribouLiteRadio *, const std::complex<short> *, CaribouLiteMeta *, size_t)> _on_data_ready_im;
    std::function<void(CaribouLiteRadio *, const std::complex<short> *, size_t)> _on_data_ready_i;
};
