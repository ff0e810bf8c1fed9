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
    while (st->reader_thread_running) {
        if (!st->stream_active) { cl_smi_readahead_cancel(smi); usleep(10000); continue; }   /* :24-28 */
        /* cariboulite_radio_read_samples(radio, interm_native_buffer1, ..., mtu_size)  :30-33, then
         * rx_queue->put(interm_native_buffer1, ret)  :44 -- the put's device-to-device copy is queued behind the
         * analysis on the seam's stream, so the two cost one synchronisation; a read that then turns out to have
         * failed (-3) cancels the put, which nobody has seen */
        const long expect = cl_smi_ra_launch(smi, st->dev->channel, st->mtu_size, st->d_native1);
        cl_ring_span sp;
        size_t room = 0;
        int put_open = 0;
        if (expect > 0 && smi->ra_pending) {
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
    if (st->reader_thread_running) {
        st->reader_thread_running = 0;
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
    st->stream_active = 0;                             /* :137 activate_channel(..., false) */
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
        st->use_async = 1;
        st->reader_thread_running = 1;
        if (pthread_create(&st->reader_thread, NULL, reader_thread_fn, st)) { st->reader_thread_running = 0; return NULL; }
    }
    return st;
}

void   cl_closeStream(cl_device *dev, cl_stream *stream) { (void)dev; if (stream) stream->stream_active = 0; }   /* :147-150 */
size_t cl_getStreamMTU(const cl_device *dev, cl_stream *stream) { (void)stream; return cl_radio_get_native_mtu_size_samples(dev->radio); }
int    cl_activateStream(cl_device *dev, cl_stream *stream, int flags, long long timeNs, size_t numElems)
{
    (void)dev; (void)flags; (void)timeNs; (void)numElems;
    stream->stream_active = 1;                         /* :191; the 20 ms settle sleep is modem hardware */
    return 0;
}
int    cl_deactivateStream(cl_device *dev, cl_stream *stream, int flags, long long timeNs)
{
    (void)dev; (void)flags; (void)timeNs;
    stream->stream_active = 0;
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

/* ------------------------------------------------------------------- RX path */
static size_t fmt_bytes(int fmt) { return fmt == CL_FORMAT_CF32 ? 8 : fmt == CL_FORMAT_CF64 ? 16 : fmt == CL_FORMAT_CS8 ? 2 : 4; }

/* Stream::Read + Stream::ReadSamples(int16*)  CaribouliteStream.cpp:260-301:
 * native read (errors squashed to 0) then the optional IIR, result left on the device */
static int read_native_device(cl_stream *st, size_t n, int *aligned, long timeout_us)
{
    cl_smi *smi = st->dev->smi;
    if (st->use_async) {
        /* Stream::Read with USE_ASYNC: rx_queue->get(buffer, num_samples, timeout_us)  :262-263.  The popped samples
         * move device-to-device into the consumer's linear buffer (the claimed span stays this call's until get_end --
         * the reader thread's puts go on meanwhile, only one that would have to displace these very elements waits),
         * and every later stage reads them there. */
        if (cl_ensure((void **)&st->d_aiq, &st->aiq_cap, n + 8, 4, 0)) return 0;
        cl_ring_span sp;
        const size_t claimed = cl_ring_get_begin(st->rx_queue, n, (int)timeout_us, &sp);
        if (aligned) *aligned = 0;
        if (!claimed) return 0;
        int bad;
        if (st->format == CL_FORMAT_CS16 && st->filter_type == CL_DIGFILT_NONE) {
            /* no device stage follows: the ring's slots go straight to the pinned mirror (*aligned = 2 tells the caller) */
            bad = cl_ensure((void **)&st->h_conv, &st->h_conv_cap, claimed * 4 + 64, 1, 1);
            uint8_t *base = (uint8_t *)cl_ring_storage(st->rx_queue), *dst = (uint8_t *)st->h_conv;
            for (int k = 0; k < 2 && !bad; k++) {
                if (!sp.len[k]) continue;
                bad = clhip_memcpy_d2h(dst, base + 4 * sp.pos[k], 4 * sp.len[k], st->astream);
                dst += 4 * sp.len[k];
            }
            if (aligned) *aligned = 2;
        } else
            bad = ring_span_copy(st->rx_queue, &sp, st->d_aiq, 0, st->astream);
        bad = bad || clhip_stream_sync(st->astream);
        cl_ring_get_end(st->rx_queue, claimed);
        if (bad) return 0;
        return (int)claimed;
    }
    /* up to one native batch per call (what every client of the reference asks for): the chunk-at-a-time reader,
     * which has the NEXT batch's bytes on their way to the device while this one is analysed and copied out */
    int ret;
    if (n <= st->mtu_size) { ret = cl_smi_read_device_ra(smi, st->dev->channel, n, NULL); if (aligned) *aligned = 0; }
    else ret = cl_smi_read_device(smi, st->dev->channel, n, 0, aligned);
    if (ret < 0) {
        if (ret == CL_SMI_ERR_IO) printf("reader thread failed to read SMI!\n");    /* :270 */
        ret = 0;                                                                    /* :266-276 */
    }
    return ret;
}

/* Stream::ReadSamples(int16*)  CaribouliteStream.cpp:291-298: the selected low-pass over all `n` slots of the native
 * read exactly as the reference loop runs (slots it leaves untouched after a re-sync hold stale samples there too).
 * Out of place -- d_raw keeps the unfiltered samples, st->d_filt takes the result -- so that a call the single-pass
 * kernel gave up on can be made again.  Asynchronous on hs; the verdict is clhip_iir_status() after the synchronise. */
static const int16_t *filter_native(cl_stream *st, const int16_t *d_raw, size_t n, void *hs, int16_t *d_dst)
{
    if (st->filter_type == CL_DIGFILT_NONE) return d_raw;
    if (!d_dst) {
        if (cl_ensure((void **)&st->d_filt, &st->filt_cap, n + 8, 4, 0)) return NULL;
        d_dst = st->d_filt;
    }
    if (clhip_iir_run(st->iir[st->filter_type - 1], d_raw, d_dst, n, n, hs)) return NULL;
    return d_dst;
}

/* ---- where the LAST device stage of a read stores its results, and how they reach the client's (pageable) buffer ----
 * Host wall time of one native batch, launch to samples in the client's buffer (tools/microbench/ingest_shape.hip,
 * profiles/r03/ingest_shape.txt; 4 / 8 / 12 output bytes per sample):
 *   CL_SINK_CLIENT  the client's buffer itself, registered with the GPU on first use (stream kwarg ZEROCOPY=1: the client
 *                   promises that a buffer it has passed stays mapped while the stream exists) -- the kernel's stores cross
 *                   PCIe themselves, nothing is left to do after the synchronisation: 29 / 39 / 48 us;
 *   CL_SINK_MIRROR  the stream's mapped pinned mirror, memcpy into the client's buffer after the synchronisation: 48 / 80 /
 *                   107 us (the memcpy reads lines the device has just written: 12 us per 512 KiB, three times its warm rate);
 *   CL_SINK_STAGED  a device buffer, copied by the copy engine into the pageable buffer before the synchronisation:
 *                   58 / 91 / 65 us -- above 1 MiB the runtime pins the target in place instead of bouncing it, which
 *                   is why this WAS the route for outputs larger than that (now only with CL_READ_STAGED=1);
 *   CL_SINK_BOUNCE  a device buffer, copied by the copy engine into the stream's pinned mirror before the synchronisation
 *                   and from there by memcpy: the route for outputs above the mirror route's limit -- 1 MiB until the end of
 *                   round 3, when FIR64 + 3/2 (1.5 MiB) was measured on both: 127-130 us here, 116-117 through the mapped
 *                   mirror; the limit is 4 MiB now (CL_MIRROR_MAX_KB), above every MTU-sized output.  (Round 3's first form handed the client's
 *                   pageable buffer to the runtime -- CL_SINK_STAGED -- which pins such a target in place from 1 MiB up:
 *                   77-81 us for FIR64 + 3/2 against 120 here, and the mechanism behind the GPU page faults on host heap
 *                   addresses of DESIGN.md section 7.  Whoever wants the copy engine in his own buffers says so: ZEROCOPY=1.) */
enum { CL_SINK_CLIENT, CL_SINK_MIRROR, CL_SINK_STAGED, CL_SINK_BOUNCE };
static size_t mirror_max_bytes(void)
{
    static size_t v;                                               /* A/B: CL_MIRROR_MAX_KB (default below) */
    if (!v) v = getenv("CL_MIRROR_MAX_KB") ? (size_t)atol(getenv("CL_MIRROR_MAX_KB")) << 10 : (size_t)4 << 20;
    return v;
}
#define CL_MIRROR_MAX_BYTES mirror_max_bytes()
typedef struct { int kind; void *d_dst; void *bounce; } cl_sink;

static void *mirror_for(cl_stream *st, size_t bytes)
{
    if (cl_ensure((void **)&st->h_conv, &st->h_conv_cap, bytes + 64, 1, 1)) return NULL;
    return clhip_host_device_ptr(st->h_conv);
}

/* ZEROCOPY=1: the address kernels use for the client's buffer (registering the pages it covers on first sight; a small
 * table, oldest entry dropped first), or NULL: not asked for, misaligned for the 16-byte stores, or the runtime refused
 * (e.g. the range overlaps an older registration only in part) -- the caller then takes one of the other routes. */
static void zc_drop_all(cl_stream *st)
{
    for (int i = 0; i < st->zc_n; i++) clhip_host_unregister(st->zc[i].base);
    st->zc_n = st->zc_next = 0;
}

static void *client_device_addr(cl_stream *st, void *out, size_t bytes)
{
    if (!st->zero_copy || !bytes || ((uintptr_t)out & 15)) return NULL;
    uint8_t *p = (uint8_t *)out;
    for (int i = 0; i < st->zc_n; i++)
        if (p >= st->zc[i].base && p + bytes <= st->zc[i].base + st->zc[i].len) return st->zc[i].dev + (p - st->zc[i].base);
    const uintptr_t pg = 4096, lo = (uintptr_t)p & ~(pg - 1), hi = ((uintptr_t)p + bytes + pg - 1) & ~(pg - 1);
    int slot = st->zc_n;
    if (slot == CL_ZC_SLOTS) { slot = st->zc_next; st->zc_next = (st->zc_next + 1) % CL_ZC_SLOTS; clhip_host_unregister(st->zc[slot].base); st->zc[slot].len = 0; }
    uint8_t *dev = (uint8_t *)clhip_host_register((void *)lo, hi - lo);
    if (!dev) {
        if (slot < st->zc_n) { st->zc[slot] = st->zc[st->zc_n - 1]; st->zc_n--; st->zc_next = 0; }   /* the evicted entry is gone */
        return NULL;
    }
    st->zc[slot].base = (uint8_t *)lo; st->zc[slot].len = hi - lo; st->zc[slot].dev = dev;
    if (slot == st->zc_n) st->zc_n++;
    st->stats.zero_copy_registrations++;
    return dev + ((uintptr_t)p - lo);
}

static int sink_open(cl_stream *st, void *out, size_t bytes, cl_sink *sk)
{
    sk->d_dst = client_device_addr(st, out, bytes);
    if (sk->d_dst) { sk->kind = CL_SINK_CLIENT; return 0; }
    if (bytes <= CL_MIRROR_MAX_BYTES) {
        sk->d_dst = mirror_for(st, bytes);
        if (!st->h_conv) return -1;
        if (sk->d_dst) { sk->kind = CL_SINK_MIRROR; return 0; }
    }
    if (cl_ensure(&st->d_conv, &st->conv_cap, bytes + 64, 1, 0)) return -1;
    sk->d_dst = st->d_conv;
    static int staged = -1;                                        /* A/B: CL_READ_STAGED=1 = the runtime copies into the client's buffer */
    if (staged < 0) staged = getenv("CL_READ_STAGED") ? atoi(getenv("CL_READ_STAGED")) != 0 : 0;
    if (staged) { sk->kind = CL_SINK_STAGED; return 0; }
    if (cl_ensure((void **)&st->h_conv, &st->h_conv_cap, bytes + 64, 1, 1)) return -1;
    sk->kind = CL_SINK_BOUNCE; sk->bounce = st->h_conv;
    return 0;
}

/* before the synchronisation (hs = the stream the last stage was queued on) ... */
static int sink_queue(const cl_sink *sk, void *out, size_t bytes, void *hs)
{
    if (!bytes) return 0;
    if (sk->kind == CL_SINK_STAGED) return clhip_memcpy_d2h(out, sk->d_dst, bytes, hs);
    if (sk->kind == CL_SINK_BOUNCE) return clhip_memcpy_d2h(sk->bounce, sk->d_dst, bytes, hs);
    return 0;
}

/* ... and after it, once the call is known to deliver */
static void sink_deliver(cl_stream *st, const cl_sink *sk, void *out, size_t bytes)
{
    if (sk->kind == CL_SINK_MIRROR || sk->kind == CL_SINK_BOUNCE) memcpy(out, st->h_conv, bytes);
    if (sk->kind == CL_SINK_CLIENT) st->stats.zero_copy_reads++;
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

static int read_stream(cl_device *dev, cl_stream *st, void *const *buffs, size_t numElems, long timeoutUs);
int cl_stream_read(cl_device *dev, cl_stream *st, void *const *buffs, size_t numElems, long timeoutUs) { return read_stream(dev, st, buffs, numElems, timeoutUs); }

/* A/B switch CL_READ_FAST=0: no one-launch unpack of a read() the host has seen to be in sync (search + unpack + offset
 * read-back for every call, as for the calls that do not qualify) */
static int read_fast_enabled(void)
{
    static int on = -1;
    if (on < 0) on = getenv("CL_READ_FAST") ? atoi(getenv("CL_READ_FAST")) != 0 : 1;
    return on;
}

int cl_stream_read_native(cl_device *dev, cl_stream *st, size_t n, long timeout_us, const int16_t **d_iq)
{
    cl_smi *smi = dev->smi;
    int aligned = 0;
    const int res = read_native_device(st, n, &aligned, timeout_us);
    if (res <= 0) return res;
    if (st->use_async && aligned == 2) return 0;        /* (plain CS16 went to the pinned mirror: not a device-side caller's route) */
    const int16_t *d_raw = st->use_async ? st->d_aiq : smi->d_iq;
    void *hs = st->use_async ? st->astream : smi->stream;
    for (int attempt = 0;; attempt++) {                 /* a call the single-pass kernel gave up on is made again, once */
        const int16_t *d_f = filter_native(st, d_raw, (size_t)res, hs, NULL);
        if (!d_f || clhip_stream_sync(hs)) return 0;
        if (!filter_overran(dev, st)) { *d_iq = d_f; return res; }
        if (attempt) return 0;
    }
}

/* Stream::ReadSamplesGen  CaribouliteStream.cpp:370-382 */
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

static int read_stream(cl_device *dev, cl_stream *st, void *const *buffs, size_t numElems, long timeoutUs)
{
    if (st->native_dir != CL_SOAPY_SDR_RX) return CL_SOAPY_SDR_NOT_SUPPORTED;       /* :248-251 */
    cl_smi *smi = dev->smi;
    clhip_set_device(smi->device);
    void *out = buffs[0];
    static int single_sync = -1;
    /* A/B switch: 0 = analysis and copy-out synchronised separately; 1 = one synchronisation, every format through the
     * pinned mirror; 2 (default) = one synchronisation, converted formats copied straight into the client's buffer
     * (all of their slots are written anyway; measured 134 -> 108 us per CF32 batch), CS16 through the mirror (only the
     * slots the reference writes may be touched, and which those are is known after the synchronisation) */
    if (single_sync < 0) single_sync = getenv("CL_READ_SINGLE_SYNC") ? atoi(getenv("CL_READ_SINGLE_SYNC")) : 2;
    if (single_sync && !st->use_async && !st->rx_pipe && st->filter_type == CL_DIGFILT_NONE && numElems && numElems <= st->mtu_size) {
        /* One native batch, no state-carrying stage behind the read: everything the call needs is queued on the seam's
         * stream behind the chunk analysis -- the conversion kernel (every slot, stale ones included, :304-367) and
         * the device-to-host copy into the pinned mirror -- so the call pays ONE synchronisation; the verdict of the
         * last read() arrives with it and a failed read simply discards what was queued.  Host side: CS16 copies
         * exactly the slots the reference writes (caribou_smi.c:344-389), the other formats all of them. */
        const size_t eb = fmt_bytes(st->format);
        cl_sink sk;
        if (sink_open(st, out, numElems * eb, &sk) ||
            cl_ensure((void **)&st->h_conv, &st->h_conv_cap, numElems * eb + 64, 1, 1) ||
            (st->format != CL_FORMAT_CS16 && cl_ensure(&st->d_conv, &st->conv_cap, numElems * 16 + 64, 1, 0)))
            return 0;
        if (sk.kind == CL_SINK_STAGED || sk.kind == CL_SINK_BOUNCE) { sk.d_dst = st->d_conv; sk.bounce = st->h_conv; }   /* (the line above may have moved them) */
        /* the short cut: a call that is one read() the host can see to be in sync is unpacked by ONE launch, in the
         * client's format, straight into the sink (cl_smi_ra_launch): no search launch, no offset read-back, no conversion
         * launch; what is left here is the one synchronisation and the sink's own last step */
        smi->fast_out = read_fast_enabled() ? sk.d_dst : NULL;
        smi->fast_format = st->format;
        const long expect = cl_smi_ra_launch(smi, dev->channel, numElems, NULL);
        smi->fast_out = NULL;
        if (smi->fast_used) {
            const int bad = expect > 0 ? sink_queue(&sk, out, (size_t)expect * eb, smi->stream) : 0;
            int fr = cl_smi_ra_finish(smi);
            if (bad) fr = CL_SMI_ERR_IO;
            if (fr == CL_SMI_ERR_IO) printf("reader thread failed to read SMI!\n");      /* :270 */
            if (fr <= 0) return 0;
            sink_deliver(st, &sk, out, (size_t)fr * eb);
            return fr;
        }
        /* the staged bytes are in pinned host memory: when every chunk of the call starts with the sync pattern the
         * call WILL deliver `expect` samples into every slot (offset 0 <=> those four words), so the device-to-host
         * copy may target the client's buffer itself.  Otherwise it goes to the pinned mirror and the client's
         * buffer is written after the verdict: nothing on a failed read, for CS16 only the slots the reference writes. */
        const int direct = single_sync == 2 && smi->ra_certain;
        void *dst = direct ? out : st->h_conv;
        int ret;
        if (expect > 0 && smi->ra_pending) {
            int bad;
            if (st->format == CL_FORMAT_CS16) bad = clhip_memcpy_d2h(dst, smi->d_iq, (size_t)expect * 4, smi->stream);
            else bad = clhip_convert_from_cs16(smi->d_iq, (size_t)expect, st->format, st->d_conv, smi->stream) ||
                       clhip_memcpy_d2h(dst, st->d_conv, (size_t)expect * eb, smi->stream);
            ret = cl_smi_ra_finish(smi);
            if (bad) ret = CL_SMI_ERR_IO;
        } else {                                           /* nothing pending, or the loop ended on an earlier read() */
            ret = expect < 0 ? (int)expect : (smi->ra_pending ? cl_smi_ra_finish(smi) : (int)expect);
            dst = st->h_conv;
            if (ret > 0) {
                int bad;
                if (st->format == CL_FORMAT_CS16) bad = clhip_memcpy_d2h(dst, smi->d_iq, (size_t)ret * 4, smi->stream);
                else bad = clhip_convert_from_cs16(smi->d_iq, (size_t)ret, st->format, st->d_conv, smi->stream) ||
                           clhip_memcpy_d2h(dst, st->d_conv, (size_t)ret * eb, smi->stream);
                if (bad || clhip_stream_sync(smi->stream)) ret = CL_SMI_ERR_IO;
            }
        }
        if (ret == CL_SMI_ERR_IO) printf("reader thread failed to read SMI!\n");    /* :270 */
        if (ret <= 0) return 0;                                                     /* :266-276 */
        if (dst == out) return ret;
        if (st->format != CL_FORMAT_CS16) { memcpy(out, st->h_conv, (size_t)ret * eb); return ret; }
        for (size_t i = 0; i < smi->n_chunks; i++) {
            const cl_chunk *c = &smi->chunks[i];
            const size_t shortening = c->offs > 0 ? (size_t)(c->offs / 4 + 1) : 0;
            const size_t nn = (c->len - 4 * shortening) / 4, n_iq = nn + (shortening > 0 && nn >= 2 ? 1 : 0);
            memcpy((uint8_t *)out + 4 * c->slot0, (const uint8_t *)st->h_conv + 4 * c->slot0, 4 * n_iq);
        }
        return ret;
    }
    if (st->format == CL_FORMAT_CS16) {                /* :282-301, no MTU clamp */
        int aligned = 0;
        int res;
        if (read_fast_enabled() && !st->use_async && st->filter_type != CL_DIGFILT_NONE && numElems && numElems <= st->mtu_size) {
            /* One native batch through the low-pass: when the call is one read() the host can see to be in sync, the
             * unpack launch and the filter launch are queued back to back and the call pays ONE synchronisation (the
             * read's verdict arrives with it); the filter stores straight into the sink. */
            /* ... and the unpack is the filter's own input conversion (clhip_iir_run_smi): the raw words of the read() go
             * straight into the filter launch -- no unpack launch, no int16 intermediate.  Only when the filter is on its
             * scan path (it has overrun before, or its memory is too long for the single-pass kernel) are the words
             * unpacked first. */
            smi->fast_out = CL_FAST_WORDS_ONLY;
            const long expect = cl_smi_ra_launch(smi, dev->channel, numElems, NULL);
            smi->fast_out = NULL;
            if (smi->fast_used && expect > 0) {
                cl_sink sk;
                clhip_iir *flt = st->iir[st->filter_type - 1];
                int bad = sink_open(st, out, (size_t)expect * 4, &sk);
                int unpacked = 0;
                if (!bad) {
                    const int rc = clhip_iir_run_smi(flt, dev->channel, smi->fast_words, (int16_t *)sk.d_dst, (size_t)expect, (size_t)expect, smi->stream);
                    if (rc == -2) {
                        unpacked = 1;
                        bad = clhip_smi_unpack_aligned(dev->channel, smi->fast_words, (size_t)expect * 4, CL_FORMAT_CS16, smi->d_iq, NULL, smi->stream) ||
                              !filter_native(st, smi->d_iq, (size_t)expect, smi->stream, (int16_t *)sk.d_dst);
                    } else bad = rc != 0;
                }
                bad = bad || sink_queue(&sk, out, (size_t)expect * 4, smi->stream);
                int fr = cl_smi_ra_finish(smi);                              /* the synchronisation */
                if (bad) fr = CL_SMI_ERR_IO;
                if (fr == CL_SMI_ERR_IO) printf("reader thread failed to read SMI!\n");    /* :270 */
                if (fr <= 0) return 0;
                if (filter_overran(dev, st)) {                               /* made again, once, on the scan path (from int16 samples) */
                    if ((!unpacked && clhip_smi_unpack_aligned(dev->channel, smi->fast_words, (size_t)fr * 4, CL_FORMAT_CS16, smi->d_iq, NULL, smi->stream)) ||
                        !filter_native(st, smi->d_iq, (size_t)fr, smi->stream, (int16_t *)sk.d_dst) ||
                        sink_queue(&sk, out, (size_t)fr * 4, smi->stream) || clhip_stream_sync(smi->stream) || filter_overran(dev, st))
                        return 0;
                }
                sink_deliver(st, &sk, out, (size_t)fr * 4);
                return fr;
            }
            res = (expect < 0 || !smi->ra_pending) ? (int)expect : cl_smi_ra_finish(smi);
            if (res < 0) {
                if (res == CL_SMI_ERR_IO) printf("reader thread failed to read SMI!\n");   /* :270 */
                res = 0;                                                                    /* :266-276 */
            }
        } else
            res = read_native_device(st, numElems, &aligned, timeoutUs);
        if (res <= 0) return res;
        if (st->use_async && aligned == 2) {        /* the ring's slots went straight to the pinned mirror */
            memcpy(out, st->h_conv, (size_t)res * 4);
            return res;
        }
        if (!st->use_async && st->filter_type == CL_DIGFILT_NONE) { if (cl_smi_copy_out(smi, (cl_sample_complex_int16 *)out, NULL, -1)) return 0; return res; }
        /* the filter (or, ASYNC without one, nothing) is the last device stage: its results go straight into the sink */
        void *hs = st->use_async ? st->astream : smi->stream;
        const int16_t *d_raw = st->use_async ? st->d_aiq : smi->d_iq;
        cl_sink sk;
        if (sink_open(st, out, (size_t)res * 4, &sk)) return 0;
        for (int attempt = 0;; attempt++) {          /* a call the single-pass kernel gave up on is made again, once */
            const int filt = st->filter_type != CL_DIGFILT_NONE;
            const int16_t *d_f = filter_native(st, d_raw, (size_t)res, hs, filt ? (int16_t *)sk.d_dst : NULL);
            if (!d_f) return 0;
            if (!filt ? clhip_memcpy_d2h(sk.kind == CL_SINK_MIRROR || sk.kind == CL_SINK_BOUNCE ? st->h_conv : out, d_f, (size_t)res * 4, hs)
                      : sink_queue(&sk, out, (size_t)res * 4, hs)) return 0;
            if (clhip_stream_sync(hs)) return 0;
            if (!filter_overran(dev, st)) break;
            if (attempt) return 0;
        }
        sink_deliver(st, &sk, out, (size_t)res * 4);
        return res;
    }
    if (numElems > st->mtu_size) numElems = st->mtu_size;                          /* :306,328,351 */
    if (st->rx_pipe && !st->use_async && st->filter_type == CL_DIGFILT_NONE) {
        /* extension stages straight from the staged raw words: no int16 intermediate, one fused launch, the sync
         * verdict checked on the device; re-sync / "-3" as caribou_smi_read has them (clhip_rx_pipe_run_smi) */
        const size_t ob = st->dsp.demod_fm ? 4 : 8, max_out = clhip_rx_pipe_out_count(st->rx_pipe, numElems) * ob;
        cl_sink sk;
        if (sink_open(st, out, max_out, &sk) || cl_ensure(&st->d_conv, &st->conv_cap, max_out + 64, 1, 0)) return 0;
        long got = 0;
        /* a call the host can see to be in sync stores its outputs straight into the client's registered buffer or the mapped
         * mirror; any other call (and every output larger than the mirror route pays for) leaves them on the device and
         * copies them into the client's buffer under the pipe call's own synchronisation, once it is known to deliver */
        const int held = sk.kind == CL_SINK_STAGED || sk.kind == CL_SINK_BOUNCE;   /* outputs stay on the device until the call is known to deliver */
        const int ret = cl_smi_read_pipe_device(smi, dev->channel, numElems, st->rx_pipe, st->d_conv, &got,
                                                sk.kind == CL_SINK_BOUNCE ? st->h_conv : out, held ? NULL : sk.d_dst);
        if (ret == CL_SMI_ERR_IO) printf("reader thread failed to read SMI!\n");    /* :270 */
        if (ret <= 0 || got <= 0) return 0;                                         /* :266-276 */
        if (sk.kind == CL_SINK_BOUNCE) memcpy(out, st->h_conv, (size_t)got * ob);   /* (the copy engine wrote the pinned mirror) */
        else if (!held && smi->pipe_out_used == sk.d_dst) sink_deliver(st, &sk, out, (size_t)got * ob);
        return (int)got;
    }
    int aligned = 0;
    int res = read_native_device(st, numElems, &aligned, timeoutUs);
    if (res <= 0) return res;
    const size_t n = (size_t)res;
    const int16_t *d_raw = st->use_async ? st->d_aiq : smi->d_iq;      /* the native samples of this call */
    void *hs = st->use_async ? st->astream : smi->stream;
    if (st->filter_type != CL_DIGFILT_NONE) aligned = 0;              /* filtered samples: the fused raw-word path no longer applies */
    /* everything behind the native read is queued on one stream and synchronised once; if the filter's verdict then says
     * the single-pass kernel gave up, its state is already back where it was: the stages are queued again, once */
    const size_t ob = st->rx_pipe ? (st->dsp.demod_fm ? 4 : 8) : fmt_bytes(st->format);
    const size_t max_out = (st->rx_pipe ? clhip_rx_pipe_out_count(st->rx_pipe, n) : n) * ob;
    cl_sink sk;                                                       /* the last stage stores straight into the sink */
    if (sink_open(st, out, max_out, &sk)) return 0;
    for (int attempt = 0;; attempt++) {
        const int16_t *d_iq = filter_native(st, d_raw, n, hs, NULL);
        if (!d_iq) return 0;
        long got;
        if (st->rx_pipe) {
            /* extension stages where a client of readStream(CF32) would apply them */
            if (aligned)   /* every chunk in sync: one fused launch straight from the raw SMI words */
                got = clhip_rx_pipe_run(st->rx_pipe, CL_PIPE_IN_SMI_WORDS, smi->d_bytes, 0, n, sk.d_dst, 0, hs);
            else           /* re-synchronised, IIR-filtered or popped from the ring: from the native int16 samples */
                got = clhip_rx_pipe_run(st->rx_pipe, CL_PIPE_IN_CS16, d_iq, 0, n, sk.d_dst, 0, hs);
            if (got < 0) return 0;
        } else {
            /* :304-367: every one of the `res` slots is converted, stale ones included */
            if (clhip_convert_from_cs16(d_iq, n, st->format, sk.d_dst, hs)) return 0;
            got = res;
        }
        if (sink_queue(&sk, out, (size_t)got * ob, hs)) return 0;
        if (clhip_stream_sync(hs)) return 0;
        if (!filter_overran(dev, st)) {
            sink_deliver(st, &sk, out, (size_t)got * ob);
            return (int)got;
        }
        if (st->rx_pipe) clhip_rx_pipe_rollback(st->rx_pipe);         /* the pipe ran on invalid samples: undo it too */
        if (attempt) return 0;
    }
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
    const size_t n = numElems, ib = fmt_bytes(st->format);
    if (cl_ensure(&st->d_conv, &st->conv_cap, n * 16 + 64, 1, 0) ||
        cl_ensure((void **)&smi->d_iq, &smi->iq_cap, n + 8, 4, 0) ||
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
    if (!d_room) {
        d_in = st->d_conv;
        if (clhip_memcpy_h2d(st->d_conv, smi->h_txin, n * ib, smi->stream)) return 0;
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
                if (clhip_take_i_rail((const float *)d_in, n, d_msg, smi->stream)) return 0;
                got = clhip_tx_pipe_run(st->tx_pipe, CL_TXPIPE_IN_FM_MESSAGE, d_msg, 0, n, d_words, 0, NULL, 0, smi->stream);
            } else
                got = clhip_tx_pipe_run(st->tx_pipe, CL_TXPIPE_IN_CF32, d_in, 0, n, d_words, 0, NULL, 0, smi->stream);
            if (got < 0) return 0;
            n_packed = (size_t)got;
        } else {
            /* :199-244 and caribou_smi.c:684-717 on the same sample, one launch */
            if (clhip_convert_pack(d_in, st->format, n, smi->tx_mode, d_words, smi->stream)) return 0;
        }
        if (n_packed && ((!d_room && clhip_memcpy_d2h(room, smi->d_bytes, 4 * n_packed, smi->stream)) || clhip_stream_sync(smi->stream))) return 0;
        /* the modulator's verdict on this very call: invalid words never reach the fd (squashed to 0 like every
         * write error, CaribouliteStream.cpp:185-194) */
        if (!st->tx_pipe || clhip_tx_pipe_status(st->tx_pipe) == 0) break;
        cl_seterr(dev->err, sizeof dev->err, "writeStream: %s", clhip_last_error());
        st->stats.tx_overruns++;
        if (attempt) return 0;
    }
    cl_smi_tx_commit(smi, 4 * n_packed);
    return (int)n;      /* elements consumed from the caller's buffer */
}
