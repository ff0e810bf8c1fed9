// CaribouLiteHip.hpp -- the sample-path half of the reference's C++ API on the MI355X.
//
// Public surface = CaribouLite.hpp:173-196 (same class / method names, argument meaning, return values):
//   ReadSamples (complex<short> | complex<float>), WriteSamples (both), StartReceiving (5 overloads),
//   StopReceiving, StartTransmitting, StopTransmitting, GetNativeMtuSample, GetRadioName, FlushBuffers.
// Everything behind it is designed around the device path: a radio owns one
// SampleEngine -- device buffers, pinned host mirrors, the SMI seam's HIP stream -- and every conversion the
// reference does in a host loop (CaribouLiteRadioCpp.cpp:41-45, :91, :148-149) is a kernel launch between the
// seam and the PCIe copy.  Callbacks and ReadSamples see the pinned mirrors; nothing is converted on the CPU.
// Hardware control (gain, frequency, bandwidth, RSSI ...) is out of scope.
// Callers: the GNU Radio block (software/gr-caribouLite/lib/caribouLiteSource_impl.cc:104-121), examples/cpp_api.
#pragma once
#include <complex>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <memory>
#include <string>

#include "cariboulite_hip.h"

#pragma pack(push, 1)
struct CaribouLiteMeta { uint8_t sync; };          // CaribouLite.hpp: cariboulite_sample_meta alias
#pragma pack(pop)

class CaribouLiteRadio {
public:
    enum RadioType { S1G = 0, HiF = 1 };
    enum RadioDir { Rx = 0, Tx = 1 };
    enum RxCbType { None = 0, FloatSync = 1, Float = 2, IntSync = 3, Int = 4 };
    enum ApiType { Async = 0, Sync = 1 };

    // `smi` stands where the CaribouLite singleton's hardware session stands: bytes reach it through
    // cl_smi_feed_bytes (the /dev/smi replacement).
    CaribouLiteRadio(cl_smi *smi, RadioType type, ApiType api_type = Async);
    virtual ~CaribouLiteRadio();
    CaribouLiteRadio(const CaribouLiteRadio &) = delete;
    CaribouLiteRadio &operator=(const CaribouLiteRadio &) = delete;
    // the other channel of the board: starting one radio's reception stops the other's (CaribouLiteRadioCpp.cpp:535-537)
    void SetSibling(CaribouLiteRadio *other);

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

    // pps tags (not in the reference's class: the device-side form of the loop its callers run over `meta`, e.g. the
    // GNU Radio source's work(), gr-caribouLite/lib/caribouLiteSource_impl.cc:113-119).  Once enabled, every fetch also
    // compacts the positions with meta[i].sync == 1 on the GPU (clhip_sync_tags) and brings them over with the samples;
    // GetSyncTags() returns how many the LAST ReadSamples / the chunk a callback is being handed holds and points
    // `*positions` at them (ascending; valid until the next fetch).
    void EnableSyncTags(bool on);
    size_t GetSyncTags(const uint32_t **positions) const;

    // General
    size_t GetNativeMtuSample();
    std::string GetRadioName();
    void FlushBuffers();

private:
    struct SampleEngine;                   // device buffers + pinned mirrors + the launches (CaribouLiteHip.cpp)
    struct Reception;                      // the callback thread of the Async flavour
    std::unique_ptr<SampleEngine> engine_;
    std::unique_ptr<Reception> reception_;
    CaribouLiteRadio *sibling_ = nullptr;
    const RadioType kind_;
    const ApiType flavour_;
    void arm(size_t samples_per_chunk, std::function<void(int)> deliver, bool wants_float);
};
