/* cariboulite_radio_hip.c -- Binding B of INTEGRATION.md section 3: the three pass-through functions of the
 * reference's radio C seam (software/libcariboulite/src/cariboulite_radio.c:1258-1315, declared at
 * cariboulite_radio.h:592-619) re-implemented over the MI355X host layer.  A maintainer drops this file into
 * libcariboulite in place of those three definitions and links -lcariboulite_host -lcariboulite_hip.
 *
 * tests/test_binding_b.py compiles it (-fsyntax-only) against the reference's own cariboulite_radio.h where
 * /root/reference exists, so that the three signatures, the struct members touched and the sample types are
 * machine-checked against the header they must match. */
#include <unistd.h>

#include "cariboulite_radio.h"
#include "cariboulite_setup.h"
#include "cariboulite_hip.h"

static cl_smi *g_smi;
static cl_radio *g_radio[2];

/* layout identity of the sample types (cariboulite_radio.h:119-128 vs cariboulite_hip.h) */
_Static_assert(sizeof(cariboulite_sample_complex_int16) == sizeof(cl_sample_complex_int16), "CS16 sample layout");
_Static_assert(sizeof(cariboulite_sample_meta) == sizeof(cl_sample_meta), "sample meta layout");
_Static_assert(CARIBOU_SMI_BYTES_PER_SAMPLE == CL_BYTES_PER_SAMPLE, "bytes per SMI sample");

static cl_radio *hip_radio(cariboulite_radio_state_st *radio)
{
    if (!g_smi) g_smi = cl_smi_init(0);
    const int ch = radio->smi_channel_id == caribou_smi_channel_2400 ? CL_CHANNEL_HIF : CL_CHANNEL_S1G;
    if (!g_radio[ch]) g_radio[ch] = cl_radio_create(g_smi, ch);
    return g_radio[ch];
}

int cariboulite_radio_read_samples(cariboulite_radio_state_st *radio, cariboulite_sample_complex_int16 *buffer,
                                   cariboulite_sample_meta *metadata, size_t length)
{
    /* pump: move what the kernel driver has into the GPU path's FIFO, one read() per native batch
     * (the read()/poll() pair of caribou_smi_timeout_read, caribou_smi.c:466-492) */
    cl_radio *r = hip_radio(radio);
    caribou_smi_st *smi = &radio->sys->smi;
    size_t want = length * CARIBOU_SMI_BYTES_PER_SAMPLE;
    while (want) {
        const size_t cur = want > smi->native_batch_len ? smi->native_batch_len : want;
        uint8_t *slot = cl_smi_feed_reserve(g_smi, cur);       /* pinned memory the GPU's copy engine reads from: no staging copy */
        if (!slot) return -1;
        const ssize_t ret = read(smi->filedesc, slot, cur);
        if (ret <= 0) break;
        cl_smi_feed_commit(g_smi, (size_t)ret);
        want -= (size_t)ret;
    }
    return cl_radio_read_samples(r, (cl_sample_complex_int16 *)buffer, (cl_sample_meta *)metadata, length);
}

int cariboulite_radio_write_samples(cariboulite_radio_state_st *radio, cariboulite_sample_complex_int16 *buffer,
                                    size_t length)
{
    const int n = cl_radio_write_samples(hip_radio(radio), (cl_sample_complex_int16 *)buffer, length);
    caribou_smi_st *smi = &radio->sys->smi;
    size_t got;
    while ((got = cl_smi_drain_bytes(g_smi, smi->write_temp_buffer, smi->native_batch_len)) > 0)
        if (write(smi->filedesc, smi->write_temp_buffer, got) < 0) return -1;      /* caribou_smi.c:444-463 */
    return n;
}

size_t cariboulite_radio_get_native_mtu_size_samples(cariboulite_radio_state_st *radio)
{
    return cl_radio_get_native_mtu_size_samples(hip_radio(radio));
}
