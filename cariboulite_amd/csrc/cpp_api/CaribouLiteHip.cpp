// CaribouLiteHip.cpp -- see CaribouLiteHip.hpp.  Each method cites the reference lines it mirrors
// (software/libcariboulite/src/CaribouLiteRadioCpp.cpp).
#include "CaribouLiteHip.hpp"

#include <chrono>
#include <cstdio>
#include <cstring>
#include <iostream>

// :5-69  reader thread: poll every 2 ms while inactive, read one chunk, convert, notify
void CaribouLiteRadio::CaribouLiteRxThread(CaribouLiteRadio *radio)
{
    const size_t mtu_size = radio->GetNativeMtuSample();
    std::complex<short> *rx_buffer = new std::complex<short>[mtu_size];
    CaribouLiteMeta *rx_meta_buffer = new CaribouLiteMeta[mtu_size];
    std::complex<float> *rx_complex_data = new std::complex<float>[mtu_size];
    while (radio->_rx_thread_running) {
        if (!radio->_rx_is_active) {
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
            continue;
        }
        int ret = cl_radio_read_samples(radio->_radio, (cl_sample_complex_int16 *)rx_buffer,
                                        (cl_sample_meta *)rx_meta_buffer, radio->_rx_samples_per_chunk);
        if (ret < 0) continue;                                    // :27-35
        if (ret == 0) {                                           // :36 (the reference spins; we yield)
            std::this_thread::sleep_for(std::chrono::microseconds(200));
            continue;
        }
        if (radio->_rxCallbackType == FloatSync || radio->_rxCallbackType == Float) {
            for (int i = 0; i < ret; i++) {                       // :41-45  short / 4096.0 (double), narrowed
                rx_complex_data[i].real(rx_buffer[i].real() / 4096.0);
                rx_complex_data[i].imag(rx_buffer[i].imag() / 4096.0);
            }
        }
        try {
            switch (radio->_rxCallbackType) {                     // :51-59
            case FloatSync: if (radio->_on_data_ready_fm) radio->_on_data_ready_fm(radio, rx_complex_data, rx_meta_buffer, ret); break;
            case Float: if (radio->_on_data_ready_f) radio->_on_data_ready_f(radio, rx_complex_data, ret); break;
            case IntSync: if (radio->_on_data_ready_im) radio->_on_data_ready_im(radio, rx_buffer, rx_meta_buffer, ret); break;
            case Int: if (radio->_on_data_ready_i) radio->_on_data_ready_i(radio, rx_buffer, ret); break;
            default: break;
            }
        } catch (std::exception &e) {
            std::cout << "OnDataReady Exception: " << e.what() << std::endl;
        }
    }
    delete[] rx_buffer; delete[] rx_meta_buffer; delete[] rx_complex_data;
}

// :72-94
int CaribouLiteRadio::ReadSamples(std::complex<float> *samples, size_t num_to_read, uint8_t *meta)
{
    if (samples == NULL) { printf("samples_is_null=%d", _read_samples == NULL); return 0; }
    int ret = ReadSamples((std::complex<short> *)NULL, num_to_read, meta);
    if (ret <= 0) return ret;
    for (size_t i = 0; i < (size_t)ret; i++)
        samples[i] = {((float)_read_samples[i].i) / 4096.0f, ((float)_read_samples[i].q) / 4096.0f};   // :91
    return ret;
}

// :97-130
int CaribouLiteRadio::ReadSamples(std::complex<short> *samples, size_t num_to_read, uint8_t *meta)
{
    if (!_rx_is_active || _read_samples == NULL || _read_metadata == NULL || num_to_read == 0) {
        printf("reading from closed stream: rx_active = %d, _read_samples_is_null=%d, _read_metadata_is_null=%d, num_to_read=%ld\n",
               (int)_rx_is_active, _read_samples == NULL, _read_metadata == NULL, (long)num_to_read);
        return 0;
    }
    // the reference reads into MTU-sized internal buffers without a bound (:105-108); a larger request
    // would overrun them there, so it is clamped here
    if (num_to_read > GetNativeMtuSample()) num_to_read = GetNativeMtuSample();
    int ret = cl_radio_read_samples(_radio, _read_samples, _read_metadata, num_to_read);
    if (ret <= 0) return ret;
    if (samples)
        for (size_t i = 0; i < (size_t)ret; i++) samples[i] = {_read_samples[i].i, _read_samples[i].q};
    if (meta) memcpy(meta, _read_metadata, (size_t)ret);
    return ret;
}

// :133-161  (uint16_t)(f * 4096) stored into a short: on the reference's build hosts this is the
// truncating conversion with 16-bit wrap, i.e. the same bits as (int16_t)(int32_t)(f * 4096)
int CaribouLiteRadio::WriteSamples(std::complex<float> *samples, size_t num_to_write)
{
    size_t written_so_far = 0, left_to_write = num_to_write;
    const size_t mtu_size = GetNativeMtuSample();
    while (written_so_far < num_to_write) {
        size_t current_write = left_to_write, k = written_so_far;
        if (current_write > mtu_size) current_write = mtu_size;
        for (size_t i = 0; i < current_write; i++, k++) {
            _write_samples[i].real((short)(uint16_t)(int32_t)(samples[k].real() * 4096));
            _write_samples[i].imag((short)(uint16_t)(int32_t)(samples[k].imag() * 4096));
        }
        int ret = WriteSamples(_write_samples, current_write);
        if (ret <= 0) break;
        written_so_far += ret;
        left_to_write -= ret;
    }
    return (int)written_so_far;
}

// :164-169
int CaribouLiteRadio::WriteSamples(std::complex<short> *samples, size_t num_to_write)
{
    return cl_radio_write_samples(_radio, (cl_sample_complex_int16 *)samples, num_to_write);
}

// :172-197
CaribouLiteRadio::CaribouLiteRadio(cl_smi *smi, RadioType type, ApiType api_type)
    : _smi(smi), _type(type), _rxCallbackType(None), _api_type(api_type)
{
    _radio = cl_radio_create(smi, type == HiF ? CL_CHANNEL_HIF : CL_CHANNEL_S1G);
    const size_t mtu_size = GetNativeMtuSample();
    if (_api_type == Async) {
        _rx_thread_running = true;
        _rx_thread = new std::thread(CaribouLiteRadio::CaribouLiteRxThread, this);
    } else {
        _read_samples = new cl_sample_complex_int16[mtu_size];
        _read_metadata = new cl_sample_meta[mtu_size];
    }
    _write_samples = new std::complex<short>[mtu_size];
}

// :200-224
CaribouLiteRadio::~CaribouLiteRadio()
{
    StopReceiving();
    StopTransmitting();
    if (_api_type == Async) {
        _rx_thread_running = false;
        _rx_thread->join();
        delete _rx_thread;
    } else {
        delete[] _read_samples; _read_samples = NULL;
        delete[] _read_metadata; _read_metadata = NULL;
    }
    delete[] _write_samples; _write_samples = NULL;
    cl_radio_destroy(_radio);
}

// :526-541
void CaribouLiteRadio::StartReceivingInternal(size_t samples_per_chunk)
{
    _rx_samples_per_chunk = (samples_per_chunk == 0) ? GetNativeMtuSample() : samples_per_chunk;
    if (_rx_samples_per_chunk > GetNativeMtuSample()) _rx_samples_per_chunk = GetNativeMtuSample();
    if (_other) _other->StopReceiving();           // only one radio receives at once
    _rx_is_active = true;
}

#define CL_START_RX(MEMBER, TYPE)                                                   \
    if (_api_type == Sync) { StartReceiving(); return; }      /* :546-550 */        \
    MEMBER = on_data_ready; _rxCallbackType = TYPE; StartReceivingInternal(samples_per_chunk);

void CaribouLiteRadio::StartReceiving(std::function<void(CaribouLiteRadio *, const std::complex<float> *, CaribouLiteMeta *, size_t)> on_data_ready, size_t samples_per_chunk) { CL_START_RX(_on_data_ready_fm, FloatSync) }
void CaribouLiteRadio::StartReceiving(std::function<void(CaribouLiteRadio *, const std::complex<float> *, size_t)> on_data_ready, size_t samples_per_chunk) { CL_START_RX(_on_data_ready_f, Float) }
void CaribouLiteRadio::StartReceiving(std::function<void(CaribouLiteRadio *, const std::complex<short> *, CaribouLiteMeta *, size_t)> on_data_ready, size_t samples_per_chunk) { CL_START_RX(_on_data_ready_im, IntSync) }
void CaribouLiteRadio::StartReceiving(std::function<void(CaribouLiteRadio *, const std::complex<short> *, size_t)> on_data_ready, size_t samples_per_chunk) { CL_START_RX(_on_data_ready_i, Int) }

void CaribouLiteRadio::StartReceiving()             // :596-601
{
    _on_data_ready_im = nullptr;
    _rxCallbackType = None;
    StartReceivingInternal(0);
}
void CaribouLiteRadio::StopReceiving() { _rx_is_active = false; }                        // :604-608
void CaribouLiteRadio::StartTransmitting() { _rx_is_active = false; _tx_is_active = true; }   // :611-618
void CaribouLiteRadio::StopTransmitting() { _tx_is_active = false; }                     // :641-646
size_t CaribouLiteRadio::GetNativeMtuSample() { return cl_radio_get_native_mtu_size_samples(_radio); }   // :667-670
std::string CaribouLiteRadio::GetRadioName() { return _type == HiF ? "CaribouLite HiF" : "CaribouLite S1G"; }
void CaribouLiteRadio::FlushBuffers()               // :681-: drop what the fifo holds
{
    uint8_t tmp[4096];
    (void)tmp;
    while (cl_smi_pending_bytes(_smi)) {
        cl_sample_complex_int16 dump[1024];
        if (cl_smi_read(_smi, CL_CHANNEL_S1G, dump, NULL, 1024) == 0) break;
    }
}
