/* cariboulite_radio_hip.c -- Binding B of INTEGRATION.md section 3: the three pass-through functions of the
 * reference's radio C seam (software/libcariboulite/src/cariboulite_radio.c:1258-1315, declared at
 * cariboulite_radio.h:592-619) re-implemented over the MI355X host layer.  A maintainer drops this file into
 * libcariboulite in place of those three definitions and links -lcariboulite_host -lcariboulite_hip.
 *
 * What stays on the host is what talks to the kernel driver, done the way caribou_smi.c does it:
 *   read:  per native batch, read() -> on nothing, poll(POLLIN, timeout from the sample rate) -> read() again
 *          (caribou_smi_timeout_read :466-492, caribou_smi_calc_read_timeout :624-629, the chunk loop :643-679);
 *          every read() lands in the GPU path's pinned FIFO and is analysed as ONE chunk, like the reference analyses
 *          what one read() returned (a short read stays a short chunk);
 *   write: ioctl(SET_STREAM_STATUS, tx) once per call (:731), then per native batch poll(POLLOUT, timeout) + write()
 *          (caribou_smi_timeout_write :444-463, the loop :737-759).
 * The sample arithmetic (sync search, unpack, re-sync extrapolation, pack) runs on the GPU behind cl_smi_read / _write.
 *
 * Compiled for real against the reference's own headers where /root/reference exists (oracle/Makefile `ref` ->
 * oracle/_ref/libbinding_b.so, together with tests/binding_b/bb_harness.c) and driven from a pipe by
 * tests/test_binding_b.py. */
#include <errno.h>
#include <poll.h>
#include <pthread.h>
#include <sys/ioctl.h>
#include <unistd.h>

#include "cariboulite_radio.h"
#include "cariboulite_setup.h"
#include "cariboulite_hip.h"

/* layout identity of the sample types (cariboulite_radio.h:119-128 vs cariboulite_hip.h) */
_Static_assert(sizeof(cariboulite_sample_complex_int16) == sizeof(cl_sample_complex_int16), "CS16 sample layout");
_Static_assert(sizeof(cariboulite_sample_meta) == sizeof(cl_sample_meta), "sample meta layout");
_Static_assert(CARIBOU_SMI_BYTES_PER_SAMPLE == CL_BYTES_PER_SAMPLE, "bytes per SMI sample");

/* one seam per process (the board has one SMI device), one radio handle per channel: the board's two channels are two
 * Soapy devices, possibly on two threads, so the lazy construction is once-only */
static cl_smi *g_smi;
static cl_radio *g_radio[2];
static pthread_once_t g_once = PTHREAD_ONCE_INIT;
static pthread_mutex_t g_io = PTHREAD_MUTEX_INITIALIZER;       /* one fd, half-duplex: one call at a time owns it */

static void hip_init(void)
{
    g_smi = cl_smi_init(0);
    if (g_smi) {
        g_radio[CL_CHANNEL_S1G] = cl_radio_create(g_smi, CL_CHANNEL_S1G);
        g_radio[CL_CHANNEL_HIF] = cl_radio_create(g_smi, CL_CHANNEL_HIF);
    }
}

static cl_radio *hip_radio(cariboulite_radio_state_st *radio)
{
    pthread_once(&g_once, hip_init);
    if (!g_smi) return NULL;
    return g_radio[radio->smi_channel_id == caribou_smi_channel_2400 ? CL_CHANNEL_HIF : CL_CHANNEL_S1G];
}

/* caribou_smi_poll :396-440 */
static int smi_poll(int fd, int events, uint32_t timeout_ms)
{
    struct pollfd fds = {.fd = fd, .events = (short)events, .revents = 0};
    for (;;) {
        const int ret = poll(&fds, 1, (int)timeout_ms);
        if (ret == -1) { if (errno == EINTR || errno == EAGAIN) continue; return -1; }
        if (ret == 0) return 0;
        return (fds.revents & POLLIN) || (fds.revents & POLLOUT);
    }
}

int cariboulite_radio_read_samples(cariboulite_radio_state_st *radio, cariboulite_sample_complex_int16 *buffer,
                                   cariboulite_sample_meta *metadata, size_t length)
{
    cl_radio *r = hip_radio(radio);
    if (!r) return -1;
    caribou_smi_st *smi = &radio->sys->smi;
    const int ch = radio->smi_channel_id == caribou_smi_channel_2400 ? CL_CHANNEL_HIF : CL_CHANNEL_S1G;
    size_t left = length * CARIBOU_SMI_BYTES_PER_SAMPLE, read_so_far = 0;
    int rc = 0;
    pthread_mutex_lock(&g_io);
    while (left) {
        const size_t cur = left > smi->native_batch_len ? smi->native_batch_len : left;
        uint32_t to_ms = (uint32_t)((2 * cur * 1000) / smi->sample_rate);                  /* :624-629 */
        if (to_ms < 1) to_ms = 1;
        to_ms *= 2;
        uint8_t *slot = cl_smi_feed_reserve(g_smi, cur);       /* pinned memory the GPU's copy engine reads from: no staging copy */
        if (!slot) { rc = -1; break; }
        ssize_t ret = read(smi->filedesc, slot, cur);          /* :473 try reading the file */
        if (ret <= 0) {
            const int p = smi_poll(smi->filedesc, POLLIN, to_ms);
            if (p < 0) { rc = -1; break; }
            if (p == 0) break;                                 /* "Reading timed-out" :657-661 */
            ret = read(smi->filedesc, slot, cur);
            if (ret < 0) { rc = -1; break; }
            if (ret == 0) break;
        }
        cl_smi_feed_commit(g_smi, (size_t)ret);
        /* the bytes of THIS read() are one chunk of the analysis (:663): the seam's next read() returns exactly them */
        cl_smi_set_max_read(g_smi, (size_t)ret);
        const int got = cl_smi_read(g_smi, ch, (cl_sample_complex_int16 *)(buffer ? buffer + read_so_far : NULL),
                                    (cl_sample_meta *)(metadata ? metadata + read_so_far : NULL),
                                    ((size_t)ret + CARIBOU_SMI_BYTES_PER_SAMPLE - 1) / CARIBOU_SMI_BYTES_PER_SAMPLE);
        cl_smi_set_max_read(g_smi, 0);
        if (got < 0) { rc = got; break; }                      /* -1, -2 (debug mode), -3 (sync lost): as caribou_smi_read returns them */
        read_so_far += (size_t)ret / CARIBOU_SMI_BYTES_PER_SAMPLE;        /* :677 */
        left -= (size_t)ret;                                   /* :678 */
    }
    cl_smi_feed_commit(g_smi, 0);                              /* (a reservation the loop left open -- time-out, error -- ends here) */
    pthread_mutex_unlock(&g_io);
    if (rc == -1) fprintf(stderr, "SMI reading operation failed\n");           /* cariboulite_radio.c:1276-1283 */
    else if (rc == -3) fprintf(stderr, "SMI data synchronization failed\n");
    return rc < 0 ? rc : (int)read_so_far;
}

int cariboulite_radio_write_samples(cariboulite_radio_state_st *radio, cariboulite_sample_complex_int16 *buffer,
                                    size_t length)
{
    cl_radio *r = hip_radio(radio);
    if (!r) return -1;
    caribou_smi_st *smi = &radio->sys->smi;
    uint32_t to_ms = (uint32_t)((2 * length * 1000) / CARIBOU_SMI_SAMPLE_RATE);            /* :724-725 */
    if (to_ms < 2) to_ms = 2;
    pthread_mutex_lock(&g_io);
    if (ioctl(smi->filedesc, SMI_STREAM_IOC_SET_STREAM_STATUS, smi_stream_tx_channel) != 0) {   /* :727-735 */
        pthread_mutex_unlock(&g_io);
        printf("caribou_smi_set_driver_streaming_state -> Failed\n");
        fprintf(stderr, "SMI writing operation failed\n");
        return -1;
    }
    smi->state = smi_stream_tx_channel;
    /* pack on the GPU (one launch for the whole call), then hand the bytes over in native batches */
    int n = cl_radio_write_samples(r, (cl_sample_complex_int16 *)buffer, length);
    size_t got, written = 0;
    while (n >= 0 && (got = cl_smi_drain_bytes(g_smi, smi->write_temp_buffer, smi->native_batch_len)) > 0) {
        const int p = smi_poll(smi->filedesc, POLLOUT, to_ms);                             /* :444-463 */
        if (p < 0) { n = -1; break; }
        if (p == 0) break;                                     /* timeout: what was written so far (:755) */
        const ssize_t w = write(smi->filedesc, smi->write_temp_buffer, got);
        if (w < 0) { n = -1; break; }
        written += got / CARIBOU_SMI_BYTES_PER_SAMPLE;         /* :757 counts the batch, whatever write() returned */
    }
    while (cl_smi_drain_bytes(g_smi, smi->write_temp_buffer, smi->native_batch_len) > 0) { }   /* a timed-out call leaves nothing queued */
    pthread_mutex_unlock(&g_io);
    if (n < 0) { fprintf(stderr, "SMI writing operation failed\n"); return -1; }
    return (int)written;
}

size_t cariboulite_radio_get_native_mtu_size_samples(cariboulite_radio_state_st *radio)
{
    return radio->sys->smi.native_batch_len / CARIBOU_SMI_BYTES_PER_SAMPLE;                /* cariboulite_radio.c:1310-1315 */
}
