/* cl_soapy.c -- the SoapySDR device/stream calls of the reference plugin
 * (soapy_api/Cariboulite.hpp:65-93, CaribouliteStreamFunctions.cpp,
 * CaribouliteStream.cpp) as a C API over opaque handles: same names, argument
 * meaning, clamping, error squashing and return codes.  Where the reference
 * throws std::runtime_error this API returns NULL and records the message.
 * The data path (unpack, IIR, conversions, FIR / resample / FM, pack) runs
 * on the GPU through the clhip_* shim. */
#include <math.h>
#include <pthread.h>
#include <unistd.h>

#include "cl_internal.h"

#define DIG_FILT_ORDER 6     /* CaribouliteStream.hpp:24 */
#define NUM_NATIVE_MTUS_PER_QUEUE 10   /* CaribouliteStream.cpp:8 */

/* ------------------------------------------------------------- filter design */
/* scipy.signal.firwin(n, cutoff, window="hamming", fs=fs) * gain, rounded to fp32 */
int cl_design_lowpass(int n_taps, double cutoff_hz, double fs_hz, double gain, float *taps_out)
{
    if (n_taps < 1 || n_taps > 4096 || !(cutoff_hz > 0) || !(cutoff_hz < fs_hz / 2) || !taps_out) return -1;
    const double pi = 3.14159265358979323846, c = cutoff_hz / (fs_hz / 2), alpha = 0.5 * (n_taps - 1);
    double *h = (double *)malloc(sizeof(double) * n_taps), sum = 0;
    if (!h) return -1;
    for (int i = 0; i < n_taps; i++) {
        const double m = i - alpha, x = c * m;
        const double sinc = x == 0.0 ? 1.0 : sin(pi * x) / (pi * x);
        const double w = n_taps == 1 ? 1.0 : 0.54 - 0.46 * cos(2 * pi * i / (n_taps - 1));
        h[i] = c * sinc * w;
        sum += h[i];
    }
    for (int i = 0; i < n_taps; i++) taps_out[i] = (float)(gain * h[i] / sum);
    free(h);
    return 0;
}

/* iir1 Butterworth::LowPass<order>::setup(fs, fc) as published: analog prototype poles
 * exp(j(pi/2 + (2i+1)pi/(2N))), bilinear low-pass map, one biquad per conjugate pair with
 * a double zero at z=-1, unit DC gain folded into the first section. sos_out: order/2 x 5. */
int cl_design_butter_lowpass(int order, double fs_hz, double fc_hz, double *sos_out)
{
    if (order < 2 || (order & 1) || order > 8 || !(fc_hz > 0) || !(fc_hz < fs_hz / 2) || !sos_out) return -1;
    const double pi = 3.14159265358979323846, k = tan(pi * fc_hz / fs_hz);
    const int pairs = order / 2;
    double gain = 1.0;
    for (int i = 0; i < pairs; i++) {
        const double th = pi / 2 + (2 * i + 1) * pi / (2.0 * order);
        const double pr = cos(th), pim = sin(th);
        const double nr = 1 + k * pr, ni = k * pim, dr = 1 - k * pr, di = -k * pim, den = dr * dr + di * di;
        const double zr = (nr * dr + ni * di) / den, zi = (ni * dr - nr * di) / den;
        double *s = sos_out + 5 * i;
        s[0] = 1; s[1] = 2; s[2] = 1; s[3] = -2 * zr; s[4] = zr * zr + zi * zi;
        gain *= (s[0] + s[1] + s[2]) / (1 + s[3] + s[4]);
    }
    const double scale = 1.0 / fabs(gain);
    sos_out[0] *= scale; sos_out[1] *= scale; sos_out[2] *= scale;
    return 0;
}

/* -------------------------------------------------------------------- kwargs */
static const char *kw(const char *const *keys, const char *const *vals, size_t n, const char *key)
{
    for (size_t i = 0; i < n; i++)
        if (keys[i] && vals[i] && !strcmp(keys[i], key)) return vals[i];
    return NULL;
}

/* device-to-device between a linear buffer and a span of the ring (elements = CS16 samples, 4 bytes) */
static int ring_span_copy(cl_ring *q, const cl_ring_span *sp, int16_t *d_linear, int to_ring, void *hs)
{
    uint8_t *base = (uint8_t *)cl_ring_storage(q), *lin = (uint8_t *)d_linear;
    for (int k = 0; k < 2; k++) {
        if (!sp->len[k]) continue;
        uint8_t *slot = base + 4 * sp->pos[k];
        if (to_ring ? clhip_memcpy_d2d(slot, lin, 4 * sp->len[k], hs) : clhip_memcpy_d2d(lin, slot, 4 * sp->len[k], hs)) return -1;
        lin += 4 * sp->len[k];
    }
    return 0;
}

/* ReaderThread  CaribouliteStream.cpp:16-49 */
static void *reader_thread_fn(void *arg)
{
    cl_stream *st = (cl_stream *)arg;
    cl_smi *smi = st->dev->smi;
    while (__atomic_load_n(&st->reader_thread_running, __ATOMIC_ACQUIRE)) {
        if (!__atomic_load_n(&st->stream_active, __ATOMIC_ACQUIRE)) { cl_smi_readahead_cancel(smi); usleep(10000); continue; }   /* :24-28 */
        /* cariboulite_radio_read_samples(radio, interm_native_buffer1, ..., mtu_size)  :30-33, then
         * rx_queue->put(interm_native_buffer1, ret)  :44 -- the put's device-to-device copy is queued behind the
         * analysis on the seam's stream, so the two cost one synchronisation; a read that then turns out to have
         * failed (-3) cancels the put, which nobody has seen */
        const long expect = cl_smi_ra_launch(smi, st->dev->channel, st->mtu_size, st->d_native1);
        cl_ring_span sp;
        size_t room = 0;
        int put_open = 0;
        /* (only where the put displaces nothing: a read that then turns out to have failed must leave the ring as it was -- the
         * reference puts nothing, CaribouliteStream.cpp:34-44 -- and an opened put has taken the oldest elements' slots already; a
         * full ring's put waits for the verdict, below) */
        if (expect > 0 && smi->ra_pending && cl_ring_size(st->rx_queue) + (size_t)expect < cl_ring_capacity(st->rx_queue)) {
            room = cl_ring_put_begin(st->rx_queue, (size_t)expect, &sp);
            put_open = 1;
            if (room && ring_span_copy(st->rx_queue, &sp, st->d_native1, 1, smi->stream)) room = 0;
        }
        int ret = expect < 0 ? (int)expect : (smi->ra_pending ? cl_smi_ra_finish(smi) : (int)expect);
        if (ret == CL_SMI_ERR_IO) fprintf(stderr, "SMI reading operation failed\n");
        else if (ret == CL_SMI_ERR_SYNC) fprintf(stderr, "SMI data synchronization failed\n");
        if (ret < 0) ret = 0;                                            /* :34-42 */
        if (put_open) {
            if (ret == expect && room) cl_ring_put_end(st->rx_queue, room);
            else if (room) cl_ring_put_abandon(st->rx_queue);            /* the copy into the span was queued already */
            else cl_ring_put_cancel(st->rx_queue);
        } else if (ret > 0) {                                            /* (a call whose last read() was already waited for) */
            size_t n = cl_ring_put_begin(st->rx_queue, (size_t)ret, &sp);
            if (n && (ring_span_copy(st->rx_queue, &sp, st->d_native1, 1, smi->stream) || clhip_stream_sync(smi->stream))) n = 0;
            cl_ring_put_end(st->rx_queue, n);
        }
        if (!ret) cl_smi_wait_bytes(smi, 2000);                          /* nothing pending: poll(POLLIN) with a timeout, do not spin */
    }
    cl_smi_readahead_cancel(smi);
    return NULL;
}

static void zc_drop_all(cl_stream *st);

static void stream_stop_async(cl_stream *st)
{
    if (__atomic_load_n(&st->reader_thread_running, __ATOMIC_ACQUIRE)) {
        __atomic_store_n(&st->reader_thread_running, 0, __ATOMIC_RELEASE);
        pthread_join(st->reader_thread, NULL);
    }
    if (st->rx_queue) { cl_ring_destroy(st->rx_queue); st->rx_queue = NULL; }
    clhip_free(st->d_native1); st->d_native1 = NULL;
    if (st->astream) { clhip_stream_sync(st->astream); clhip_stream_destroy(st->astream); st->astream = NULL; }
    clhip_free(st->d_aiq); st->d_aiq = NULL; st->aiq_cap = 0;
    st->use_async = 0;
}

/* ------------------------------------------------------------ device / stream */
static void stream_free(cl_stream *st)
{
    if (!st) return;
    stream_stop_async(st);
    for (int i = 0; i < 3; i++) clhip_iir_destroy(st->iir[i]);
    if (st->dev && st->dev->smi) clhip_stream_sync(st->dev->smi->stream);
    zc_drop_all(st);
    clhip_free(st->d_filt); clhip_free(st->d_conv); clhip_host_free(st->h_conv);
    if (st->rx_pipe) clhip_rx_pipe_destroy(st->rx_pipe);
    if (st->tx_pipe) clhip_tx_pipe_destroy(st->tx_pipe);
    free(st);
}

/* SoapySDR::Stream::Stream  CaribouliteStream.cpp:52-98 */
static cl_stream *stream_new(cl_device *dev)
{
    cl_stream *st = (cl_stream *)calloc(1, sizeof *st);
    if (!st) return NULL;
    st->dev = dev;
    st->mtu_size = cl_radio_get_native_mtu_size_samples(dev->radio);
    st->format = CL_FORMAT_CS16;                       /* :77 */
    st->native_dir = CL_SOAPY_SDR_RX;
    st->filter_type = CL_DIGFILT_NONE;                 /* :84 */
    const double bw[3] = {20e3, 50e3, 100e3};          /* :85-91 setup(4e6, bw/2) */
    for (int i = 0; i < 3; i++) {
        cl_design_butter_lowpass(DIG_FILT_ORDER, 4e6, bw[i] / 2, st->sos[i]);
        st->iir[i] = clhip_iir_create(st->sos[i], DIG_FILT_ORDER / 2, 1);
        if (!st->iir[i]) { stream_free(st); return NULL; }
    }
    return st;
}

cl_device *cl_device_make(const char *const *keys, const char *const *vals, size_t n)
{
    const char *ch = kw(keys, vals, n, "channel");
    int channel;
    if (ch && !strcmp(ch, "HiF")) channel = CL_CHANNEL_HIF;           /* Cariboulite.cpp:17-24 */
    else if (ch && !strcmp(ch, "S1G")) channel = CL_CHANNEL_S1G;
    else return NULL;                                                  /* "Channel type is not specified correctly" :27 */
    const char *g = kw(keys, vals, n, "gpu");
    cl_device *dev = (cl_device *)calloc(1, sizeof *dev);
    if (!dev) return NULL;
    dev->channel = channel;
    dev->smi = cl_smi_init(g ? atoi(g) : 0);
    if (!dev->smi) { free(dev); return NULL; }
    dev->radio = cl_radio_create(dev->smi, channel);
    dev->stream = dev->radio ? stream_new(dev) : NULL;
    if (!dev->stream) { cl_device_unmake(dev); return NULL; }          /* "Stream allocation failed" :33 */
    return dev;
}

void cl_device_unmake(cl_device *dev)
{
    if (!dev) return;
    if (dev->smi) clhip_set_device(dev->smi->device);
    stream_free(dev->stream);                                          /* Cariboulite.cpp:38-41 */
    cl_radio_destroy(dev->radio);
    cl_smi_close(dev->smi);
    free(dev);
}

cl_smi *cl_device_smi(cl_device *dev) { return dev ? dev->smi : NULL; }
const char *cl_device_last_error(cl_device *dev) { return dev ? dev->err : "no device"; }

/* CaribouliteStreamFunctions.cpp:11-19 */
size_t cl_getStreamFormats(const cl_device *dev, int direction, size_t channel, const char **formats, size_t max)
{
    (void)dev; (void)direction; (void)channel;
    static const char *f[4] = {"CS16", "CS8", "CF32", "CF64"};
    for (size_t i = 0; i < 4 && i < max; i++) formats[i] = f[i];
    return 4;
}

/* CaribouliteStreamFunctions.cpp:31-35 */
const char *cl_getNativeStreamFormat(const cl_device *dev, int direction, size_t channel, double *fullScale)
{
    (void)dev; (void)direction; (void)channel;
    if (fullScale) *fullScale = (double)((1 << 12) - 1);
    return "CS16";
}

/* Stream::setFormat  CaribouliteStream.cpp:158-173 */
static int set_format(cl_stream *st, const char *fmt)
{
    if (!fmt) return -1;
    if (!strcmp(fmt, "CS16")) st->format = CL_FORMAT_CS16;
    else if (!strcmp(fmt, "CS8")) st->format = CL_FORMAT_CS8;
    else if (!strcmp(fmt, "CF32")) st->format = CL_FORMAT_CF32;
    else if (!strcmp(fmt, "CF64")) st->format = CL_FORMAT_CF64;
    else return -1;
    return 0;
}

static int parse_dsp(cl_device *dev, cl_dsp_cfg *d, const char *const *keys, const char *const *vals, size_t n)
{
    memset(d, 0, sizeof *d);
    d->up = d->down = 1;
    const char *fir = kw(keys, vals, n, "FIR"), *rs = kw(keys, vals, n, "RESAMP"),
               *dm = kw(keys, vals, n, "DEMOD"), *md = kw(keys, vals, n, "MOD");
    if (fir) {                                   /* FIR=<ntaps>:<cutoff_hz> */
        int nt = 0; double fc = 0;
        if (sscanf(fir, "%d:%lf", &nt, &fc) != 2 || nt < 1 || nt > 128 || cl_design_lowpass(nt, fc, 4e6, 1.0, d->fir)) {
            cl_seterr(dev->err, sizeof dev->err, "setupStream invalid FIR spec %s", fir);
            return -1;
        }
        d->n_fir = nt; d->enabled = 1;
    }
    if (rs) {                                    /* RESAMP=<L>/<M>: 8 taps per phase, gain L */
        int L = 0, M = 0;
        if (sscanf(rs, "%d/%d", &L, &M) != 2 || L < 1 || M < 1 || 8 * L > 40) {
            cl_seterr(dev->err, sizeof dev->err, "setupStream invalid RESAMP spec %s", rs);
            return -1;
        }
        d->up = L; d->down = M; d->n_rs = 8 * L;
        /* scipy firwin(8L, 1/max(L,M)) with the cut-off normalised to Nyquist = 1 */
        if (cl_design_lowpass(8 * L, 1.0 / (L > M ? L : M), 2.0, (double)L, d->rs)) return -1;
        d->enabled = 1;
    }
    if (dm) {
        if (strcmp(dm, "FM")) { cl_seterr(dev->err, sizeof dev->err, "setupStream invalid DEMOD %s", dm); return -1; }
        d->demod_fm = 1; d->enabled = 1;
    }
    if (md) {
        if (sscanf(md, "FM:%lf", &d->mod_kf) != 1) { cl_seterr(dev->err, sizeof dev->err, "setupStream invalid MOD %s", md); return -1; }
        d->mod_fm = 1; d->enabled = 1;
    }
    if (d->enabled && !d->n_fir && (d->demod_fm || (d->up != 1 || d->down != 1)) ) {
        /* RX stages hang off the FIR kernel: a 1-tap identity FIR stands in when none is requested */
        d->n_fir = 1; d->fir[0] = 1.0f;
    }
    return 0;
}

/* CaribouliteStreamFunctions.cpp:100-139 */
cl_stream *cl_setupStream(cl_device *dev, int direction, const char *format, const size_t *channels, size_t n_channels,
                          const char *const *keys, const char *const *vals, size_t n_kwargs)
{
    (void)channels; (void)n_channels;
    if (!dev) return NULL;
    cl_stream *st = dev->stream;                       /* the preallocated stream :105 */
    if (set_format(st, format) != 0) {                 /* :109-115 throws */
        cl_seterr(dev->err, sizeof dev->err, "setupStream invalid format %s", format ? format : "(null)");
        return NULL;
    }
    st->native_dir = direction == CL_SOAPY_SDR_TX ? CL_SOAPY_SDR_TX : CL_SOAPY_SDR_RX;   /* :117 */
    /* "CW" kwarg drives a modem hardware override (:123-135): no host data-path effect */
    cl_dsp_cfg d;
    if (parse_dsp(dev, &d, keys, vals, n_kwargs)) return NULL;
    if (d.enabled && st->format != CL_FORMAT_CF32) {
        cl_seterr(dev->err, sizeof dev->err, "setupStream: FIR/RESAMP/DEMOD/MOD stages need format CF32");
        return NULL;
    }
    clhip_set_device(dev->smi->device);
    if (st->rx_pipe) { clhip_rx_pipe_destroy(st->rx_pipe); st->rx_pipe = NULL; }
    if (st->tx_home) st->tx_home(st->tx_home_ctx, st->tx_home_member);     /* (a stream group lets go of the pipe's state first) */
    if (st->tx_pipe) { clhip_tx_pipe_destroy(st->tx_pipe); st->tx_pipe = NULL; }
    st->dsp = d;
    if (d.enabled && st->native_dir == CL_SOAPY_SDR_RX) {
        st->rx_pipe = clhip_rx_pipe_create(1, dev->channel, d.fir, d.n_fir, d.n_rs ? d.rs : NULL, d.n_rs, d.up, d.down,
                                           d.demod_fm ? CL_PIPE_OUT_FM_DEMOD : CL_PIPE_OUT_IQ);
        if (!st->rx_pipe) { cl_seterr(dev->err, sizeof dev->err, "setupStream: %s", clhip_last_error()); return NULL; }
    } else if (d.enabled) {
        st->tx_pipe = clhip_tx_pipe_create(1, d.mod_fm ? d.mod_kf : 0.0, 4e6, d.n_rs ? d.rs : NULL, d.n_rs, d.up, d.down,
                                           dev->smi->tx_mode);
        if (!st->tx_pipe) { cl_seterr(dev->err, sizeof dev->err, "setupStream: %s", clhip_last_error()); return NULL; }
    }
    __atomic_store_n(&st->stream_active, 0, __ATOMIC_RELEASE);   /* :137 activate_channel(..., false); (the reader thread reads the flag) */
    stream_stop_async(st);
    clhip_stream_sync(dev->smi->stream);
    zc_drop_all(st);
    const char *zc = kw(keys, vals, n_kwargs, "ZEROCOPY");
    st->zero_copy = zc && !strcmp(zc, "1") && st->native_dir == CL_SOAPY_SDR_RX;
    const char *as = kw(keys, vals, n_kwargs, "ASYNC");
    if (as && !strcmp(as, "1") && st->native_dir == CL_SOAPY_SDR_RX) {
        /* rx_queue(mtu * NUM_NATIVE_MTUS_PER_QUEUE, override writes, blocking reads)  :70-75 */
        st->rx_queue = cl_ring_create_device(dev->smi->device, st->mtu_size * NUM_NATIVE_MTUS_PER_QUEUE, sizeof(cl_sample_complex_int16), 1, 1);
        st->d_native1 = (int16_t *)clhip_malloc(sizeof(cl_sample_complex_int16) * (st->mtu_size + 8));
        st->astream = clhip_stream_create();
        if (!st->rx_queue || !st->d_native1 || !st->astream) { cl_seterr(dev->err, sizeof dev->err, "setupStream: ASYNC allocation failed"); return NULL; }
        /* (interm_native_buffer1: the slots a re-synchronised read() leaves untouched keep what the read before left -- zeros at first) */
        if (clhip_memset(st->d_native1, 0, sizeof(cl_sample_complex_int16) * (st->mtu_size + 8), st->astream) || clhip_stream_sync(st->astream)) return NULL;
        st->use_async = 1;
        __atomic_store_n(&st->reader_thread_running, 1, __ATOMIC_RELEASE);
        if (pthread_create(&st->reader_thread, NULL, reader_thread_fn, st)) { __atomic_store_n(&st->reader_thread_running, 0, __ATOMIC_RELEASE); return NULL; }
    }
    return st;
}

void   cl_closeStream(cl_device *dev, cl_stream *stream) { (void)dev; if (stream) __atomic_store_n(&stream->stream_active, 0, __ATOMIC_RELEASE); }   /* :147-150 */
size_t cl_getStreamMTU(const cl_device *dev, cl_stream *stream) { (void)stream; return cl_radio_get_native_mtu_size_samples(dev->radio); }
int    cl_activateStream(cl_device *dev, cl_stream *stream, int flags, long long timeNs, size_t numElems)
{
    (void)dev; (void)flags; (void)timeNs; (void)numElems;
    __atomic_store_n(&stream->stream_active, 1, __ATOMIC_RELEASE);   /* :191; the 20 ms settle sleep is modem hardware */
    return 0;
}
int    cl_deactivateStream(cl_device *dev, cl_stream *stream, int flags, long long timeNs)
{
    (void)dev; (void)flags; (void)timeNs;
    __atomic_store_n(&stream->stream_active, 0, __ATOMIC_RELEASE);
    return 0;
}

/* Cariboulite.cpp:395-417 */
void cl_setBandwidth(cl_device *dev, int direction, size_t channel, double bw)
{
    (void)channel;
    if (direction != CL_SOAPY_SDR_RX) return;
    cl_stream *st = dev->stream;
    if (bw < 160000.0) {
        if (bw <= 20000.0) st->filter_type = CL_DIGFILT_20KHZ;
        else if (bw <= 50000.0) st->filter_type = CL_DIGFILT_50KHZ;
        else if (bw <= 100000.0) st->filter_type = CL_DIGFILT_100KHZ;
        else st->filter_type = CL_DIGFILT_NONE;
    } else st->filter_type = CL_DIGFILT_NONE;
}
int cl_getDigitalFilter(const cl_device *dev) { return dev->stream->filter_type; }
/* ASYNC=1: samples the reader thread has queued and no readStream has taken yet (rx_queue->size(), circular_buffer.h) */
size_t cl_stream_queue_size(const cl_device *dev, const cl_stream *st) { (void)dev; return st && st->rx_queue ? cl_ring_size(st->rx_queue) : 0; }

/* ------------------------------------------------------------------- RX path */
static size_t fmt_bytes(int fmt) { return fmt == CL_FORMAT_CF32 ? 8 : fmt == CL_FORMAT_CF64 ? 16 : fmt == CL_FORMAT_CS8 ? 2 : 4; }

/* A read is   SOURCE -> STAGES -> SINK   with one epilogue (synchronise, verdicts, deliver or redo once):
 *
 *   SOURCE  Stream::Read (CaribouliteStream.cpp:260-279): the call's native samples, on the device.  Either the RAW WORDS of a call
 *           that is one read() the host has seen to be in sync (caribou_smi_find_buffer_offset returns 0 exactly when the first
 *           four words carry the pattern, caribou_smi.c:235-292, and the staged bytes are in pinned host memory) -- nothing has
 *           been launched, the read's verdict arrives with the epilogue's synchronisation -- or INT16 SAMPLES, complete: the
 *           chunk loop with its re-syncs, extrapolated samples, untouched slots and "-3" (cl_smi_read_device*), or what the
 *           ASYNC reader thread queued in the device ring.
 *   STAGES  Stream::ReadSamples* (:282-367) + the extension stages of setupStream's kwargs: [IIR] ->
 *           [/4096 conversion | FIR -> L/M or FM demod | nothing].  From raw words the first stage's own input conversion is the
 *           13-bit field extraction of caribou_smi_rx_data_analyze (:338-378): unpack + format conversion in one launch, the IIR
 *           fed from words, the fused pipe.
 *   SINK    where the last stage stores and how that reaches the client's buffer (below).
 *
 * The reference's dispatch is one switch over the format (CaribouliteStream.cpp:370-382); so is this one. */
/* (cl_source: cl_internal.h) */

static int source_acquire(cl_device *dev, cl_stream *st, size_t n, long timeout_us, void *ring_host_dst, cl_source *src)
{
    cl_smi *smi = dev->smi;
    memset(src, 0, sizeof *src);
    if (st->use_async) {
        /* Stream::Read with USE_ASYNC: rx_queue->get(buffer, num_samples, timeout_us)  :262-263.  The popped samples move
         * device-to-device into the consumer's linear buffer (the claimed span stays this call's until get_end -- the reader
         * thread's puts go on meanwhile, only one that would have to displace these very elements waits) */
        src->hs = st->astream;
        if (cl_ensure((void **)&st->d_aiq, &st->aiq_cap, n + 8, 4, 0)) return 0;
        cl_ring_span sp;
        const size_t claimed = cl_ring_get_begin(st->rx_queue, n, (int)timeout_us, &sp);
        if (!claimed) return 0;
        int bad = 0;
        if (ring_host_dst) {
            /* no device stage follows (plain CS16): the ring's slots go straight to the sink's host side, no device-to-device hop */
            uint8_t *base = (uint8_t *)cl_ring_storage(st->rx_queue), *dst = (uint8_t *)ring_host_dst;
            for (int k = 0; k < 2 && !bad; k++) {
                if (!sp.len[k]) continue;
                bad = clhip_memcpy_d2h(dst, base + 4 * sp.pos[k], 4 * sp.len[k], st->astream);
                dst += 4 * sp.len[k];
            }
            src->host_filled = 1;
        } else {
            bad = ring_span_copy(st->rx_queue, &sp, st->d_aiq, 0, st->astream);
            src->d_cs16 = st->d_aiq;
        }
        if (bad) {
            clhip_stream_sync(st->astream);
            cl_ring_get_end(st->rx_queue, claimed);
            return 0;
        }
        src->n = (int)claimed; src->ring_claimed = claimed;
        return src->n;
    }
    src->hs = smi->stream;
    int ret;
    if (n <= st->mtu_size) {
        /* up to one native batch (what every client of the reference asks for): the chunk-at-a-time reader, which has the NEXT
         * batch's bytes on their way to the device while this one is worked on */
        smi->want_words = 1;
        const long expect = cl_smi_ra_launch(smi, dev->channel, n, NULL);
        if (smi->fast_used && expect > 0) {
            src->n = (int)expect; src->d_words = smi->fast_words; src->pending = 1;
            return src->n;
        }
        ret = (expect < 0 || !smi->ra_pending) ? (int)expect : cl_smi_ra_finish(smi);
    } else
        ret = cl_smi_read_device(smi, dev->channel, n, 0, NULL);
    if (ret < 0) {
        if (ret == CL_SMI_ERR_IO) printf("reader thread failed to read SMI!\n");    /* :270 */
        src->err = ret;
        ret = 0;                                                                    /* :266-276 */
    }
    src->n = ret; src->d_cs16 = smi->d_iq;
    return ret;
}

/* hs has been synchronised (or is, here): the ring elements an ASYNC call claimed are the reader thread's again */
static void source_release(cl_stream *st, cl_source *src, int synced)
{
    if (!src->ring_claimed) return;
    if (!synced) clhip_stream_sync(src->hs);
    cl_ring_get_end(st->rx_queue, src->ring_claimed);
    src->ring_claimed = 0;
}

/* ---- SINK: where the LAST device stage of a read stores its results, and how they reach the client's (pageable) buffer ----
 * Host wall time of one native batch, launch to samples in the client's buffer (tools/microbench/ingest_shape.hip,
 * profiles/r03/ingest_shape.txt; 4 / 8 / 12 output bytes per sample):
 *   CL_SINK_CLIENT  the client's buffer itself, REGISTERED by the client (cl_stream_register_buffer on a ZEROCOPY=1 stream): the
 *                   kernel's stores cross PCIe themselves, nothing is left to do after the synchronisation: 29 / 39 / 48 us;
 *   CL_SINK_MIRROR  the stream's mapped pinned mirror, memcpy into the client's buffer after the synchronisation: 48 / 80 / 107 us
 *                   (up to 4 MiB: above every MTU-sized output; FIR64 + 3/2 = 1.5 MiB: 116-117 us against 127-130 through BOUNCE);
 *   CL_SINK_BOUNCE  a device buffer, copied by the copy engine into the stream's pinned mirror before the synchronisation and from
 *                   there by memcpy: outputs above the mirror route's limit (CS16 calls of many native batches).
 * The library never hands memory it does not own to the runtime's copy engine or registers it behind the client's back (DESIGN.md
 * section 7: the runtime's in-place pinning of pageable copies >= 1 MiB ended test sessions with GPU page faults on host heap
 * addresses). */
enum { CL_SINK_CLIENT, CL_SINK_MIRROR, CL_SINK_BOUNCE };
#define CL_MIRROR_MAX_BYTES ((size_t)4 << 20)
/* (cl_sink: cl_internal.h) */

/* ZEROCOPY=1 + cl_stream_register_buffer: the address kernels use for a client pointer inside a registered buffer, or NULL (not
 * asked for, not registered, misaligned for the 16-byte stores): the caller then takes the mirror */
static void *client_device_addr(cl_stream *st, void *out, size_t bytes)
{
    if (!st->zero_copy || !bytes || ((uintptr_t)out & 15)) return NULL;
    uint8_t *p = (uint8_t *)out;
    for (int i = 0; i < st->zc_n; i++)
        if (p >= st->zc[i].base && p + bytes <= st->zc[i].base + st->zc[i].len) return st->zc[i].dev + (p - st->zc[i].base);
    return NULL;
}

static void zc_drop_all(cl_stream *st)
{
    /* (the seam's persistent buffer may be standing on filtered samples in a registered client buffer: taken over before it goes) */
    if (st->zc_n && st->dev && st->dev->smi && st->dev->smi->prev_is_cs16) cl_smi_restore_prev_words(st->dev->smi, st->dev->channel);
    for (int i = 0; i < st->zc_n; i++) clhip_host_unregister(st->zc[i].base);
    st->zc_n = 0;
}

/* Explicit registration of a client buffer with the GPU for a ZEROCOPY=1 stream: readStream calls whose buffs[0] lies inside a
 * registered buffer have their last kernel store into it directly.  The registration covers the pages around [p, p + bytes) and
 * stands until cl_stream_unregister_buffers, the next setupStream or the device's end -- the client keeps the buffer allocated
 * that long.  At most CL_ZC_SLOTS buffers; a full table refuses (no eviction: registrations never churn in steady state).
 * 0, or -1 (cl_device_last_error). */
int cl_stream_register_buffer(cl_device *dev, cl_stream *st, void *p, size_t bytes)
{
    if (!dev || !st || !p || !bytes) return -1;
    if (!st->zero_copy) { cl_seterr(dev->err, sizeof dev->err, "cl_stream_register_buffer: the stream was not set up with ZEROCOPY=1"); return -1; }
    clhip_set_device(dev->smi->device);
    const uintptr_t pg = 4096;
    uintptr_t lo = (uintptr_t)p & ~(pg - 1), hi = ((uintptr_t)p + bytes + pg - 1) & ~(pg - 1);
    /* registrations are whole pages, and buffers that come from one heap share pages with their neighbours: a new buffer whose
     * pages touch a registered range is registered together with it, as one range (this is set-up time, not the data path:
     * the stream's device work is waited for before the older registration is replaced) */
    for (int i = 0; i < st->zc_n; i++) {
        const uintptr_t b0 = (uintptr_t)st->zc[i].base, b1 = b0 + st->zc[i].len;
        if (lo >= b1 || hi <= b0) continue;
        if (lo >= b0 && hi <= b1) return 0;                        /* inside a registered range already */
        clhip_stream_sync(dev->smi->stream);
        if (st->astream) clhip_stream_sync(st->astream);
        clhip_host_unregister(st->zc[i].base);
        if (b0 < lo) lo = b0;
        if (b1 > hi) hi = b1;
        st->zc[i] = st->zc[--st->zc_n];
        i = -1;                                                    /* the grown range may touch another one */
    }
    if (st->zc_n == CL_ZC_SLOTS) { cl_seterr(dev->err, sizeof dev->err, "cl_stream_register_buffer: %d buffers are registered already", CL_ZC_SLOTS); return -1; }
    uint8_t *d = (uint8_t *)clhip_host_register((void *)lo, hi - lo);
    if (!d) { cl_seterr(dev->err, sizeof dev->err, "cl_stream_register_buffer: %s", clhip_last_error()); return -1; }
    st->zc[st->zc_n].base = (uint8_t *)lo; st->zc[st->zc_n].len = hi - lo; st->zc[st->zc_n].dev = d;
    st->zc_n++;
    st->stats.zero_copy_registrations++;
    return 0;
}

void cl_stream_unregister_buffers(cl_device *dev, cl_stream *st)
{
    if (!dev || !st) return;
    clhip_set_device(dev->smi->device);
    clhip_stream_sync(dev->smi->stream);               /* nothing may still be storing into them */
    if (st->astream) clhip_stream_sync(st->astream);
    zc_drop_all(st);
}

static int sink_open(cl_stream *st, void *out, size_t bytes, cl_sink *sk)
{
    sk->d_dst = client_device_addr(st, out, bytes);
    if (sk->d_dst) { sk->kind = CL_SINK_CLIENT; return 0; }
    /* (the seam's persistent buffer may be standing on filtered samples in one of the buffers that is about to grow) */
    if ((bytes + 64 > st->h_conv_cap || bytes + 64 > st->conv_cap) && st->dev->smi->prev_is_cs16 && cl_smi_restore_prev_words(st->dev->smi, st->dev->channel)) return -1;
    if (cl_ensure((void **)&st->h_conv, &st->h_conv_cap, bytes + 64, 1, 1)) return -1;
    if (bytes <= CL_MIRROR_MAX_BYTES && (sk->d_dst = clhip_host_device_ptr(st->h_conv)) != NULL) { sk->kind = CL_SINK_MIRROR; return 0; }
    if (cl_ensure(&st->d_conv, &st->conv_cap, bytes + 64, 1, 0)) return -1;
    sk->d_dst = st->d_conv; sk->kind = CL_SINK_BOUNCE;
    return 0;
}

/* before the synchronisation (hs = the stream the last stage was queued on) ... */
static int sink_queue(cl_stream *st, const cl_sink *sk, size_t bytes, void *hs)
{
    return bytes && sk->kind == CL_SINK_BOUNCE ? clhip_memcpy_d2h(st->h_conv, sk->d_dst, bytes, hs) : 0;
}

/* ... and after it, once the call is known to deliver */
static void sink_deliver(cl_stream *st, const cl_sink *sk, void *out, size_t bytes)
{
    if (sk->kind == CL_SINK_CLIENT) st->stats.zero_copy_reads++;
    else memcpy(out, st->h_conv, bytes);
}

/* Stream::ReadSamples(int16*)  CaribouliteStream.cpp:291-298: the selected low-pass over all `n` slots of the native read
 * exactly as the reference loop runs (slots it leaves untouched after a re-sync hold stale samples there too).  Out of place --
 * the source keeps the unfiltered samples -- so that a call the single-pass kernel gave up on can be made again.  Asynchronous
 * on hs; the verdict is clhip_iir_status() after the synchronise.  d_dst NULL: into st->d_filt. */
static const int16_t *filter_source(cl_device *dev, cl_stream *st, const cl_source *src, int16_t *d_dst)
{
    cl_smi *smi = dev->smi;
    const size_t n = (size_t)src->n;
    if (st->iir_home) st->iir_home(st->iir_home_ctx, st->iir_home_member);     /* (a stream group hands the filters' state back first) */
    clhip_iir *flt = st->iir[st->filter_type - 1];
    if (!d_dst) {
        if (cl_ensure((void **)&st->d_filt, &st->filt_cap, n + 8, 4, 0)) return NULL;
        d_dst = st->d_filt;
    }
    const int16_t *d_raw = src->d_cs16;
    if (src->d_words) {
        /* the raw words of the read() go straight into the filter launch: the 13-bit field extraction is the filter's own input
         * conversion -- no unpack launch, no int16 intermediate.  Only when the filter is on its scan path (it has overrun before,
         * or its memory is too long for the single-pass kernel: -2) are the words unpacked first. */
        const int rc = clhip_iir_run_smi(flt, dev->channel, src->d_words, d_dst, n, n, src->hs);
        if (rc == 0) return cl_smi_set_prev_cs16(smi, dev->channel, d_dst, n) ? NULL : d_dst;
        if (rc != -2 || clhip_smi_unpack_aligned(dev->channel, src->d_words, n * 4, CL_FORMAT_CS16, smi->d_iq, NULL, src->hs)) return NULL;
        smi->prev_words = NULL;                            /* (the persistent int16 buffer is current now) */
        d_raw = smi->d_iq;
    }
    if (clhip_iir_run(flt, d_raw, d_dst, n, n, src->hs)) return NULL;
    /* what the call delivers IS the persistent buffer from now on: the reference filters in place in the buffer it read into */
    return cl_smi_set_prev_cs16(smi, dev->channel, d_dst, n) ? NULL : d_dst;
}

/* queue the call's stages on src->hs, the last one storing into the sink; *got = elements the call yields.  0 or -1 */
static int stages_queue(cl_device *dev, cl_stream *st, const cl_source *src, const cl_sink *sk, long *got)
{
    cl_smi *smi = dev->smi;
    const size_t n = (size_t)src->n;
    const int filt = st->filter_type != CL_DIGFILT_NONE;
    *got = src->n;
    if (st->rx_pipe) {
        /* extension stages where a client of readStream(CF32) would apply them */
        const int16_t *d_f = filt ? filter_source(dev, st, src, NULL) : src->d_cs16;
        if (filt && !d_f) return -1;
        *got = d_f ? clhip_rx_pipe_run(st->rx_pipe, CL_PIPE_IN_CS16, d_f, 0, n, sk->d_dst, 0, src->hs)
                   : clhip_rx_pipe_run(st->rx_pipe, CL_PIPE_IN_SMI_WORDS, src->d_words, 0, n, sk->d_dst, 0, src->hs);   /* one fused launch from the raw words */
        return *got < 0 ? -1 : 0;
    }
    if (st->format == CL_FORMAT_CS16) {                    /* :282-301 */
        if (filt) return filter_source(dev, st, src, (int16_t *)sk->d_dst) ? 0 : -1;
        if (src->d_words) {                                /* unpack straight into the sink and the persistent int16 buffer */
            smi->prev_words = NULL;
            return clhip_smi_unpack_aligned(dev->channel, src->d_words, n * 4, CL_FORMAT_CS16, sk->d_dst, smi->d_iq, src->hs) ? -1 : 0;
        }
        return 0;                                          /* nothing to launch: the epilogue copies the slots the reference writes */
    }
    /* :304-367: every one of the n slots is converted, stale ones included */
    if (!filt && src->d_words) {                           /* one launch: unpack in the client's format into the sink, int16 pairs into the persistent buffer */
        smi->prev_words = NULL;
        return clhip_smi_unpack_aligned(dev->channel, src->d_words, n * 4, st->format, sk->d_dst, smi->d_iq, src->hs) ? -1 : 0;
    }
    const int16_t *d_f = filt ? filter_source(dev, st, src, NULL) : src->d_cs16;
    if (!d_f) return -1;
    return clhip_convert_from_cs16(d_f, n, st->format, sk->d_dst, src->hs) ? -1 : 0;
}

/* after the synchronise: 0 = the filtered samples are good (or no filter ran); 1 = the call overran -- the filter's state
 * is back where it was, the object has switched to the scan path, the caller repeats its stages once */
static int filter_overran(cl_device *dev, cl_stream *st)
{
    if (st->filter_type == CL_DIGFILT_NONE || clhip_iir_status(st->iir[st->filter_type - 1]) == 0) return 0;
    st->stats.iir_overruns++;
    cl_seterr(dev->err, sizeof dev->err, "readStream: %s", clhip_last_error());
    return 1;
}

/* Stream::ReadSamplesGen  CaribouliteStream.cpp:370-382, in two halves: _begin acquires the source (the read() of the call: a FIFO pop
 * or the reader thread's ring) and QUEUES everything behind it -- analysis, low-pass, stages, the way out -- on the source's stream;
 * _end synchronises, takes the verdicts (the read()'s, the filter's: a call the single-pass kernel gave up on is queued again, once)
 * and delivers.  cl_readStream is the two back to back; a stream group runs the _begin of all its one-by-one members before the
 * first _end, so that their chains cross PCIe and run together instead of one after the other. */
int cl_stream_read_begin(cl_device *dev, cl_stream *st, void *out, size_t numElems, long timeoutUs, cl_read_ctx *c)
{
    memset(c, 0, sizeof *c);
    c->out = out;
    if (st->native_dir != CL_SOAPY_SDR_RX) { c->ret = CL_SOAPY_SDR_NOT_SUPPORTED; return 0; }       /* :248-251 */
    cl_smi *smi = dev->smi;
    clhip_set_device(smi->device);
    if (st->format != CL_FORMAT_CS16 && numElems > st->mtu_size) numElems = st->mtu_size;   /* :306,328,351; CS16 is not clamped (:282-301) */
    if (!numElems) return 0;
    const int plain_cs16 = st->format == CL_FORMAT_CS16 && !st->rx_pipe && st->filter_type == CL_DIGFILT_NONE;
    int sink_ready = 0;
    void *ring_dst = NULL;
    if (plain_cs16 && st->use_async) {                     /* the ring's samples ARE the output: the sink comes first */
        if (sink_open(st, out, numElems * 4, &c->sk)) return 0;
        sink_ready = 1;
        ring_dst = c->sk.kind == CL_SINK_CLIENT ? out : (void *)st->h_conv;
    }
    if (source_acquire(dev, st, numElems, timeoutUs, ring_dst, &c->src) <= 0) {
        /* CS16 is read straight into the client's buffer (:282-301): what the chunk loop wrote before the read() that found no sync
         * ("-3", caribou_smi.c:665-668) is in it, although the call reports 0 elements */
        if (plain_cs16 && !st->use_async && c->src.err == CL_SMI_ERR_SYNC) cl_smi_copy_out(smi, (cl_sample_complex_int16 *)out, NULL, -1);
        return 0;
    }
    const size_t n = (size_t)c->src.n;
    if (plain_cs16 && !c->src.d_words && !st->use_async) {
        /* no device stage behind the read: exactly the slots the reference writes (caribou_smi.c:344-389) go to the client */
        c->ret = cl_smi_copy_out(smi, (cl_sample_complex_int16 *)out, NULL, -1) ? 0 : c->src.n;
        return 0;
    }
    c->ob = st->rx_pipe ? (st->dsp.demod_fm ? 4 : 8) : fmt_bytes(st->format);
    const size_t max_out = (st->rx_pipe ? clhip_rx_pipe_out_count(st->rx_pipe, n) : n) * c->ob;
    if (!sink_ready && sink_open(st, out, max_out, &c->sk)) { if (c->src.pending) cl_smi_ra_finish(smi); source_release(st, &c->src, 0); return 0; }
    if (c->src.host_filled) c->got = c->src.n;             /* (ASYNC, plain CS16: already on its way to the sink's host side) */
    else {
        c->bad = stages_queue(dev, st, &c->src, &c->sk, &c->got);
        if (!c->bad) c->bad = sink_queue(st, &c->sk, (size_t)(c->got > 0 ? c->got : 0) * c->ob, c->src.hs);
    }
    c->open = 1;
    return 1;
}

int cl_stream_read_end(cl_device *dev, cl_stream *st, cl_read_ctx *c)
{
    if (!c->open) return c->ret;
    cl_smi *smi = dev->smi;
    clhip_set_device(smi->device);
    c->open = 0;
    for (int attempt = 0;; attempt++) {
        if (attempt) {                                     /* (the stages again, behind a filter that is on its scan path now) */
            c->got = 0;
            c->bad = stages_queue(dev, st, &c->src, &c->sk, &c->got);
            if (!c->bad) c->bad = sink_queue(st, &c->sk, (size_t)(c->got > 0 ? c->got : 0) * c->ob, c->src.hs);
        }
        /* the one synchronisation; with it the verdict of the read() whose words the stages took (host-certain: in sync) */
        int fr = c->src.n;
        if (c->src.pending) { fr = cl_smi_ra_finish(smi); c->src.pending = 0; }
        else if (clhip_stream_sync(c->src.hs)) fr = CL_SMI_ERR_IO;
        if (c->bad) { clhip_stream_sync(c->src.hs); fr = CL_SMI_ERR_IO; }
        source_release(st, &c->src, 1);
        if (fr == CL_SMI_ERR_IO) printf("reader thread failed to read SMI!\n");    /* :270 */
        if (fr <= 0) return 0;                                                      /* :266-276 */
        if (!filter_overran(dev, st)) {
            if (c->got <= 0) return 0;
            sink_deliver(st, &c->sk, c->out, (size_t)c->got * c->ob);
            return (int)c->got;
        }
        /* the single-pass filter kernel gave up: its state is back where it was and the object is on the scan path; whatever ran
         * behind it ran on invalid samples: undone, and the stages are queued again, once */
        if (st->rx_pipe) clhip_rx_pipe_rollback(st->rx_pipe);
        if (attempt) return 0;
    }
}

static int read_stream(cl_device *dev, cl_stream *st, void *const *buffs, size_t numElems, long timeoutUs)
{
    cl_read_ctx c;
    cl_stream_read_begin(dev, st, buffs[0], numElems, timeoutUs, &c);
    return cl_stream_read_end(dev, st, &c);
}

int cl_stream_read(cl_device *dev, cl_stream *st, void *const *buffs, size_t numElems, long timeoutUs) { return read_stream(dev, st, buffs, numElems, timeoutUs); }

/* Stream::Read + Stream::ReadSamples(int16*) with the result left on the DEVICE, complete (cl_group.c: a group's pipe slot runs
 * from them) */
int cl_stream_read_native(cl_device *dev, cl_stream *st, size_t n, long timeout_us, const int16_t **d_iq)
{
    cl_smi *smi = dev->smi;
    cl_source src;
    if (st->use_async) { if (source_acquire(dev, st, n, timeout_us, NULL, &src) <= 0) return 0; source_release(st, &src, 0); }
    else {
        /* int16 samples wanted: the chunk loop proper (it also brings the persistent buffer up to date first) */
        int ret = n <= st->mtu_size ? cl_smi_read_device_ra(smi, dev->channel, n, NULL) : cl_smi_read_device(smi, dev->channel, n, 0, NULL);
        if (ret < 0) { if (ret == CL_SMI_ERR_IO) printf("reader thread failed to read SMI!\n"); ret = 0; }
        if (ret <= 0) return 0;
        memset(&src, 0, sizeof src);
        src.n = ret; src.d_cs16 = smi->d_iq; src.hs = smi->stream;
    }
    if (st->filter_type == CL_DIGFILT_NONE) { *d_iq = src.d_cs16; return src.n; }
    for (int attempt = 0;; attempt++) {                 /* a call the single-pass kernel gave up on is made again, once */
        const int16_t *d_f = filter_source(dev, st, &src, NULL);
        if (!d_f || clhip_stream_sync(src.hs)) return 0;
        if (!filter_overran(dev, st)) { *d_iq = d_f; return src.n; }
        if (attempt) return 0;
    }
}

int cl_readStream(cl_device *dev, cl_stream *st, void *const *buffs, size_t numElems, int *flags, long long *timeNs, long timeoutUs)
{
    (void)flags; (void)timeNs;                         /* never written; timeoutUs only matters in ASYNC mode */
    const int ret = read_stream(dev, st, buffs, numElems, timeoutUs);
    st->stats.read_calls++;
    if (ret > 0) st->stats.elements_read += (uint64_t)ret; else if (ret == 0) st->stats.reads_empty++;
    return ret;
}

void cl_getStreamStats(const cl_device *dev, const cl_stream *st, cl_stream_stats *out)
{
    (void)dev;
    if (!out) return;
    if (st) *out = st->stats; else memset(out, 0, sizeof *out);
}
unsigned long cl_stream_iir_overruns(const cl_stream *st) { return st ? (unsigned long)st->stats.iir_overruns : 0; }
void cl_stream_set_iir_poll_bound(cl_stream *st, int polls)
{
    if (st) for (int i = 0; i < 3; i++) clhip_iir_set_poll_bound(st->iir[i], polls);
}

/* ------------------------------------------------------------------- TX path */
/* Stream::WriteSamplesGen  CaribouliteStream.cpp:247-258 */
static int write_stream(cl_device *dev, cl_stream *st, const void *const *buffs, size_t numElems);

int cl_writeStream(cl_device *dev, cl_stream *st, const void *const *buffs, size_t numElems, int *flags, long long timeNs, long timeoutUs)
{
    (void)flags; (void)timeNs; (void)timeoutUs;
    const int ret = write_stream(dev, st, buffs, numElems);
    st->stats.write_calls++;
    if (ret > 0) st->stats.elements_written += (uint64_t)ret; else if (ret == 0) st->stats.writes_empty++;
    return ret;
}

static int write_stream(cl_device *dev, cl_stream *st, const void *const *buffs, size_t numElems)
{
    if (st->native_dir != CL_SOAPY_SDR_TX) return CL_SOAPY_SDR_NOT_SUPPORTED;       /* :285-288 */
    cl_smi *smi = dev->smi;
    clhip_set_device(smi->device);
    const void *in = buffs[0];
    if (st->format == CL_FORMAT_CS16) {                /* :182-196 */
        int ret = cl_radio_write_samples(dev->radio, (cl_sample_complex_int16 *)in, numElems);
        if (ret < 0) { if (ret == -1) printf("Failed to write\n"); ret = 0; }
        return ret;
    }
    if (numElems > st->mtu_size) numElems = st->mtu_size;                          /* :201,217,234 */
    if (numElems == 0) return 0;
    if (st->tx_pipe && st->tx_home) st->tx_home(st->tx_home_ctx, st->tx_home_member);   /* (a stream group hands the modulator's state back first) */
    const size_t n = numElems, ib = fmt_bytes(st->format);
    if (cl_ensure(&st->d_conv, &st->conv_cap, n * 16 + 64, 1, 0) ||
        cl_smi_ensure_iq(smi, n + 8) ||
        cl_ensure((void **)&smi->d_bytes, &smi->bytes_cap, 4 * n * (size_t)(st->tx_pipe ? st->dsp.up : 1) + 256, 1, 0))
        return 0;
    /* the client's samples reach the device through a pinned buffer of ours (see smi_write_core) */
    if (cl_ensure((void **)&smi->h_txin, &smi->h_txin_cap, n * ib + 64, 1, 1)) return 0;
    memcpy(smi->h_txin, in, n * ib);
    /* the packed words go straight into the pinned TX FIFO (caribou_smi_write's chunk loop, caribou_smi.c:738-759, appends
     * native-batch pieces of one contiguous array): room for the most a call can produce, committed once it is known to be good */
    uint8_t *room = cl_smi_tx_reserve(smi, 4 * n * (size_t)(st->tx_pipe ? st->dsp.up : 1) + 64);
    if (!room) return 0;
    /* MTU-sized calls: the first kernel reads the pinned samples and the last one stores into the FIFO's room across PCIe
     * themselves -- no copy-engine call on either side (cl_write_mapped_max: A/B).  Not for a pipe fed CF32: its kernel reads
     * every input several times. */
    const void *d_in = n * ib <= cl_write_mapped_max() && !(st->tx_pipe && !st->dsp.mod_fm) ? clhip_host_device_ptr(smi->h_txin) : NULL;
    uint8_t *d_room = d_in ? (uint8_t *)cl_fifo_device_ptr(&smi->tx, room) : NULL;
    /* (every exit behind a queued copy or launch synchronises first -- `fail`: the kernels read smi->h_txin and store into the FIFO's
     * reserved room, both of which the next call may move or overwrite) */
    if (!d_room) {
        d_in = st->d_conv;
        if (clhip_memcpy_h2d(st->d_conv, smi->h_txin, n * ib, smi->stream)) goto fail;
    }
    uint8_t *d_words = d_room ? d_room : smi->d_bytes;
    size_t n_packed = n;
    /* a modulator call whose look-back gave up (dispatch-order mode) has put the pipe back where it was and switched it
     * to ticket order: the call is simply made again, once */
    for (int attempt = 0;; attempt++) {
        n_packed = n;
        if (st->tx_pipe) {
            /* MOD=FM: the I component carries the real message (SURVEY.md a13 "if given I/Q, use I") */
            long got;
            if (st->dsp.mod_fm) {
                /* the I rail as a dense message, taken on the device behind the samples (d_conv holds 16 bytes per element) */
                float *d_msg = (float *)st->d_conv + 2 * n;
                if (clhip_take_i_rail((const float *)d_in, n, d_msg, smi->stream)) goto fail;
                got = clhip_tx_pipe_run(st->tx_pipe, CL_TXPIPE_IN_FM_MESSAGE, d_msg, 0, n, d_words, 0, NULL, 0, smi->stream);
            } else
                got = clhip_tx_pipe_run(st->tx_pipe, CL_TXPIPE_IN_CF32, d_in, 0, n, d_words, 0, NULL, 0, smi->stream);
            if (got < 0) goto fail;
            n_packed = (size_t)got;
        } else {
            /* :199-244 and caribou_smi.c:684-717 on the same sample, one launch */
            if (clhip_convert_pack(d_in, st->format, n, smi->tx_mode, d_words, smi->stream)) goto fail;
        }
        if (n_packed && ((!d_room && clhip_memcpy_d2h(room, smi->d_bytes, 4 * n_packed, smi->stream)) || clhip_stream_sync(smi->stream))) goto fail;
        /* the modulator's verdict on this very call: invalid words never reach the fd (squashed to 0 like every
         * write error, CaribouliteStream.cpp:185-194) */
        if (!st->tx_pipe || clhip_tx_pipe_status(st->tx_pipe) == 0) break;
        cl_seterr(dev->err, sizeof dev->err, "writeStream: %s", clhip_last_error());
        st->stats.tx_overruns++;
        if (attempt) return 0;
    }
    cl_smi_tx_commit(smi, 4 * n_packed);
    return (int)n;      /* elements consumed from the caller's buffer */
fail:
    clhip_stream_sync(smi->stream);
    return 0;
}
