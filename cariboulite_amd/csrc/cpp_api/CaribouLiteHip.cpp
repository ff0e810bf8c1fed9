// CaribouLiteHip.cpp -- see CaribouLiteHip.hpp.  Behaviour follows the reference's
// software/libcariboulite/src/CaribouLiteRadioCpp.cpp (lines cited where a rule comes from it); the structure is
// this project's own: one SampleEngine per radio keeps every sample on the GPU between the SMI seam and the PCIe
// copy, and the Async flavour's thread delivers through one closure bound when reception is armed.
#include "CaribouLiteHip.hpp"

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <exception>
#include <mutex>
#include <stdexcept>
#include <thread>

namespace {

// RAII over the C-ABI allocators: a device array and a pinned host array of T
template <class T> class DeviceArray {
public:
    explicit DeviceArray(size_t n) : p_(static_cast<T *>(clhip_malloc(n * sizeof(T)))) {}
    ~DeviceArray() { clhip_free(p_); }
    DeviceArray(const DeviceArray &) = delete;
    DeviceArray &operator=(const DeviceArray &) = delete;
    T *get() const { return p_; }
private:
    T *p_;
};
template <class T> class PinnedArray {
public:
    explicit PinnedArray(size_t n) : p_(static_cast<T *>(clhip_host_alloc(n * sizeof(T)))) {}
    ~PinnedArray() { clhip_host_free(p_); }
    PinnedArray(const PinnedArray &) = delete;
    PinnedArray &operator=(const PinnedArray &) = delete;
    T *get() const { return p_; }
private:
    T *p_;
};

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// SampleEngine: what a radio owns on the GPU.  One native batch (MTU) of every representation, device side and
// pinned host mirror, all on the SMI seam's stream so a fetch is: seam read (unpack kernel, re-sync semantics of
// caribou_smi.c:295-393 included) -> optional /4096 kernel -> one copy per plane the caller asked for.
// ---------------------------------------------------------------------------------------------------------------
struct CaribouLiteRadio::SampleEngine {
    cl_smi *const smi;
    cl_radio *const seam;                  // the radio C seam of this channel (cariboulite_radio.h:592-619)
    const size_t mtu;
    void *const stream;
    // native int16 pairs: persists across calls like the reference's _read_samples / rx_buffer, so the slots a
    // re-sync leaves untouched (caribou_smi.c:382-389) hold what they held there
    DeviceArray<int16_t> d_native;
    DeviceArray<uint8_t> d_sync;
    DeviceArray<float> d_float;            // also the staging of float TX samples
    PinnedArray<std::complex<short>> h_native;
    PinnedArray<CaribouLiteMeta> h_sync;
    PinnedArray<std::complex<float>> h_float;
    // pps tags of the fetched chunk: word 0 = how many, words 1.. = positions (one copy brings the count and the first
    // TAGS_EAGER positions; a chunk with more of them -- a test stream, never a real second marker -- costs a second copy)
    enum { TAGS_EAGER = 1023 };
    DeviceArray<uint32_t> d_tags;
    DeviceArray<uint8_t> d_tag_ws;         // per-tile counts of clhip_sync_tags (only a native batch above 256 KiB needs any)
    PinnedArray<uint32_t> h_tags;
    std::atomic<bool> tags_wanted{false};

    SampleEngine(cl_smi *s, int channel)
        : smi(s), seam(cl_radio_create(s, channel)), mtu(cl_radio_get_native_mtu_size_samples(seam)), stream(cl_smi_stream(s)),
          d_native(2 * (mtu + 8)), d_sync(mtu + 8), d_float(2 * (mtu + 8)), h_native(mtu + 8), h_sync(mtu + 8), h_float(mtu + 8),
          d_tags(1 + mtu + 8), d_tag_ws(clhip_sync_tags_ws_bytes(mtu + 8) + 4), h_tags(1 + mtu + 8)
    {
        if (h_tags.get()) h_tags.get()[0] = 0;
    }
    ~SampleEngine() { cl_radio_destroy(seam); }
    bool usable() const
    {
        return seam && d_native.get() && d_sync.get() && d_float.get() && h_native.get() && h_sync.get() && h_float.get() &&
               d_tags.get() && d_tag_ws.get() && h_tags.get();
    }

    // One seam read of up to `count` samples (count <= mtu).  Afterwards the pinned mirrors hold: the sync plane,
    // and the float plane (as_float) or the native plane.  Returns the seam's own code (samples, 0, -1, -2, -3).
    int fetch(size_t count, bool as_float)
    {
        clhip_set_device(cl_smi_device(smi));
        h_tags.get()[0] = 0;
        const int got = cl_radio_read_samples_device(seam, d_native.get(), d_sync.get(), count);
        if (got <= 0) return got;
        const size_t n = static_cast<size_t>(got);
        int bad = clhip_memcpy_d2h(h_sync.get(), d_sync.get(), n, stream);
        const bool tags = tags_wanted.load();
        if (tags)           // the callers' `if (meta[i] == 1)` loop, on the plane where it lies (caribouLiteSource_impl.cc:113-119)
            bad = bad || clhip_sync_tags(d_sync.get(), n, d_tags.get() + 1, mtu + 8, d_tags.get(), d_tag_ws.get(), stream) ||
                  clhip_memcpy_d2h(h_tags.get(), d_tags.get(), sizeof(uint32_t) * (1 + (n < TAGS_EAGER ? n : (size_t)TAGS_EAGER)), stream);
        if (as_float)       // ((float)v) / 4096.0f on every slot, stale ones included (CaribouLiteRadioCpp.cpp:41-45,:91)
            bad = bad || clhip_convert_from_cs16(d_native.get(), n, CL_FORMAT_CF32, d_float.get(), stream) ||
                  clhip_memcpy_d2h(h_float.get(), d_float.get(), n * sizeof(std::complex<float>), stream);
        else
            bad = bad || clhip_memcpy_d2h(h_native.get(), d_native.get(), n * sizeof(std::complex<short>), stream);
        if (bad || clhip_stream_sync(stream)) return -1;
        if (tags && h_tags.get()[0] > TAGS_EAGER &&
            (clhip_memcpy_d2h(h_tags.get() + 1 + TAGS_EAGER, d_tags.get() + 1 + TAGS_EAGER,
                              sizeof(uint32_t) * (h_tags.get()[0] - TAGS_EAGER), stream) || clhip_stream_sync(stream)))
            return -1;
        return got;
    }

    // float samples -> the 16-bit store of CaribouLiteRadioCpp.cpp:148-149 -> seam write, at most one MTU.
    // `(uint16_t)(f * 4096)` kept in a short is, on the reference's x86-64 build, the truncating conversion's low
    // 16 bits -- what clhip_convert_to_cs16 computes for CL_FORMAT_CF32.
    int send_float(const std::complex<float> *src, size_t count)
    {
        clhip_set_device(cl_smi_device(smi));
        std::memcpy(static_cast<void *>(h_float.get()), src, count * sizeof(std::complex<float>));
        if (clhip_memcpy_h2d(d_float.get(), h_float.get(), count * sizeof(std::complex<float>), stream) ||
            clhip_convert_to_cs16(d_float.get(), CL_FORMAT_CF32, count, d_native.get(), stream))
            return -1;
        return cl_radio_write_samples_device(seam, d_native.get(), count);
    }
};

// ---------------------------------------------------------------------------------------------------------------
// Reception: the Async flavour's thread.  Idle -> naps 2 ms (CaribouLiteRadioCpp.cpp:16-20).  Armed -> fetches one
// chunk in the representation the armed callback wants and hands the pinned mirrors to `deliver`.
// ---------------------------------------------------------------------------------------------------------------
struct CaribouLiteRadio::Reception {
    std::atomic<bool> alive{false}, listening{false};
    std::mutex plan_lock;                          // guards the three fields below against re-arming mid-flight
    std::function<void(int)> deliver;              // bound to the user's callback and the engine's mirrors
    bool float_plane = false;
    size_t chunk = 0;
    std::thread worker;

    void run(SampleEngine *engine)
    {
        while (alive.load()) {
            if (!listening.load()) { std::this_thread::sleep_for(std::chrono::milliseconds(2)); continue; }
            std::function<void(int)> out;
            bool as_float; size_t want;
            { std::lock_guard<std::mutex> g(plan_lock); out = deliver; as_float = float_plane; want = chunk; }
            const int got = engine->fetch(want, as_float);
            if (got < 0) continue;                                         // every seam error is dropped (:27-35)
            if (got == 0) { std::this_thread::sleep_for(std::chrono::microseconds(200)); continue; }   // the reference spins here
            if (!out) continue;                                            // StartReceiving() without a callback
            try { out(got); }
            catch (const std::exception &e) { std::fprintf(stderr, "CaribouLiteRadio: data callback threw: %s\n", e.what()); }
        }
    }
};

CaribouLiteRadio::CaribouLiteRadio(cl_smi *smi, RadioType type, ApiType api_type)
    : engine_(new SampleEngine(smi, type == HiF ? CL_CHANNEL_HIF : CL_CHANNEL_S1G)), reception_(new Reception), kind_(type),
      flavour_(api_type)
{
    if (!engine_->usable()) throw std::runtime_error(std::string("CaribouLiteRadio: GPU buffers unavailable: ") + clhip_last_error());
    if (flavour_ == Async) {
        reception_->alive = true;
        reception_->worker = std::thread([this] { reception_->run(engine_.get()); });
    }
}

CaribouLiteRadio::~CaribouLiteRadio()
{
    StopReceiving();
    StopTransmitting();
    reception_->alive = false;
    if (reception_->worker.joinable()) reception_->worker.join();
}

void CaribouLiteRadio::SetSibling(CaribouLiteRadio *other) { sibling_ = other; }

// StartReceivingInternal (:526-541): chunk = request or MTU, never above it; the sibling channel stops first
void CaribouLiteRadio::arm(size_t samples_per_chunk, std::function<void(int)> deliver, bool wants_float)
{
    const size_t mtu = engine_->mtu;
    {
        std::lock_guard<std::mutex> g(reception_->plan_lock);
        reception_->chunk = (samples_per_chunk == 0 || samples_per_chunk > mtu) ? mtu : samples_per_chunk;
        reception_->deliver = std::move(deliver);
        reception_->float_plane = wants_float;
    }
    if (sibling_) sibling_->StopReceiving();
    reception_->listening = true;
}

// The Sync flavour ignores callbacks (:546-550 and the three siblings).
void CaribouLiteRadio::StartReceiving(std::function<void(CaribouLiteRadio *, const std::complex<float> *, CaribouLiteMeta *, size_t)> on_data_ready, size_t samples_per_chunk)
{
    if (flavour_ == Sync) return StartReceiving();
    SampleEngine *e = engine_.get();
    arm(samples_per_chunk, [this, e, on_data_ready](int n) { if (on_data_ready) on_data_ready(this, e->h_float.get(), e->h_sync.get(), (size_t)n); }, true);
}
void CaribouLiteRadio::StartReceiving(std::function<void(CaribouLiteRadio *, const std::complex<float> *, size_t)> on_data_ready, size_t samples_per_chunk)
{
    if (flavour_ == Sync) return StartReceiving();
    SampleEngine *e = engine_.get();
    arm(samples_per_chunk, [this, e, on_data_ready](int n) { if (on_data_ready) on_data_ready(this, e->h_float.get(), (size_t)n); }, true);
}
void CaribouLiteRadio::StartReceiving(std::function<void(CaribouLiteRadio *, const std::complex<short> *, CaribouLiteMeta *, size_t)> on_data_ready, size_t samples_per_chunk)
{
    if (flavour_ == Sync) return StartReceiving();
    SampleEngine *e = engine_.get();
    arm(samples_per_chunk, [this, e, on_data_ready](int n) { if (on_data_ready) on_data_ready(this, e->h_native.get(), e->h_sync.get(), (size_t)n); }, false);
}
void CaribouLiteRadio::StartReceiving(std::function<void(CaribouLiteRadio *, const std::complex<short> *, size_t)> on_data_ready, size_t samples_per_chunk)
{
    if (flavour_ == Sync) return StartReceiving();
    SampleEngine *e = engine_.get();
    arm(samples_per_chunk, [this, e, on_data_ready](int n) { if (on_data_ready) on_data_ready(this, e->h_native.get(), (size_t)n); }, false);
}
void CaribouLiteRadio::StartReceiving() { arm(0, nullptr, false); }                      // :596-601: no callback, full MTU

void CaribouLiteRadio::StopReceiving() { reception_->listening = false; }                 // :604-608
void CaribouLiteRadio::StartTransmitting() { reception_->listening = false; }             // :611-618: transmit = not receiving
void CaribouLiteRadio::StopTransmitting() {}                                              // :641-646: modem control only

// :97-130.  Only the Sync flavour owns read buffers there (_read_samples is NULL otherwise), so only it can read.
int CaribouLiteRadio::ReadSamples(std::complex<short> *samples, size_t num_to_read, uint8_t *meta)
{
    if (!reception_->listening || flavour_ != Sync || num_to_read == 0) {
        std::fprintf(stderr, "CaribouLiteRadio::ReadSamples: stream closed (receiving=%d, sync api=%d, requested=%zu)\n",
                     (int)reception_->listening.load(), (int)(flavour_ == Sync), num_to_read);
        return 0;
    }
    // the reference reads into its MTU-sized buffers without a bound (:105-108); a larger request is clamped here
    const int got = engine_->fetch(num_to_read > engine_->mtu ? engine_->mtu : num_to_read, false);
    if (got <= 0) return got;
    if (samples) std::memcpy(static_cast<void *>(samples), engine_->h_native.get(), (size_t)got * sizeof(std::complex<short>));
    if (meta) std::memcpy(meta, engine_->h_sync.get(), (size_t)got);
    return got;
}

// :72-94
int CaribouLiteRadio::ReadSamples(std::complex<float> *samples, size_t num_to_read, uint8_t *meta)
{
    if (samples == NULL) { std::fprintf(stderr, "CaribouLiteRadio::ReadSamples: no destination buffer\n"); return 0; }
    if (!reception_->listening || flavour_ != Sync || num_to_read == 0) return ReadSamples((std::complex<short> *)NULL, num_to_read, meta);
    const int got = engine_->fetch(num_to_read > engine_->mtu ? engine_->mtu : num_to_read, true);
    if (got <= 0) return got;
    std::memcpy(static_cast<void *>(samples), engine_->h_float.get(), (size_t)got * sizeof(std::complex<float>));
    if (meta) std::memcpy(meta, engine_->h_sync.get(), (size_t)got);
    return got;
}

// :133-161: MTU-sized pieces until one is refused; the sum of what the seam accepted
int CaribouLiteRadio::WriteSamples(std::complex<float> *samples, size_t num_to_write)
{
    size_t sent = 0;
    while (sent < num_to_write) {
        const size_t piece = num_to_write - sent > engine_->mtu ? engine_->mtu : num_to_write - sent;
        const int took = engine_->send_float(samples + sent, piece);
        if (took <= 0) break;
        sent += (size_t)took;
    }
    return (int)sent;
}

// :164-169
int CaribouLiteRadio::WriteSamples(std::complex<short> *samples, size_t num_to_write)
{
    return cl_radio_write_samples(engine_->seam, reinterpret_cast<cl_sample_complex_int16 *>(samples), num_to_write);
}

void CaribouLiteRadio::EnableSyncTags(bool on) { engine_->tags_wanted = on; }
size_t CaribouLiteRadio::GetSyncTags(const uint32_t **positions) const
{
    if (positions) *positions = engine_->h_tags.get() + 1;
    return engine_->h_tags.get()[0];
}

size_t CaribouLiteRadio::GetNativeMtuSample() { return engine_->mtu; }                    // :667-670
std::string CaribouLiteRadio::GetRadioName()                                              // cariboulite.c:223-245 (full board)
{
    return kind_ == HiF ? "CaribouLite 6GHz" : "CaribouLite S1G";
}
void CaribouLiteRadio::FlushBuffers() { cl_smi_flush_fifo(engine_->smi); }                // :681-685 -> caribou_smi.c:772-783
