/* bb_harness.c -- TEST INFRASTRUCTURE: builds the reference's own cariboulite_radio_state_st / sys_st around an
 * ordinary file descriptor (a pipe) and calls Binding B's three functions through the reference's declarations.
 * Compiled with tests/binding_b/cariboulite_radio_hip.c against the reference headers into oracle/_ref/libbinding_b.so
 * (oracle/Makefile `ref`); tests/test_binding_b.py drives it.  No reference code is copied: only its headers are included. */
#include <stdlib.h>
#include <string.h>

#include "cariboulite_radio.h"
#include "cariboulite_setup.h"

typedef struct { sys_st sys; cariboulite_radio_state_st radio; } bb_ctx;

void *bb_open(int fd, int channel_2400, size_t native_batch_len)
{
    bb_ctx *c = (bb_ctx *)calloc(1, sizeof *c);
    if (!c) return NULL;
    c->sys.smi.initialized = 1;
    c->sys.smi.filedesc = fd;
    c->sys.smi.native_batch_len = native_batch_len;
    c->sys.smi.sample_rate = CARIBOU_SMI_SAMPLE_RATE;
    c->sys.smi.write_temp_buffer = (uint8_t *)malloc(native_batch_len + 1024);
    c->radio.sys = &c->sys;
    c->radio.smi_channel_id = channel_2400 ? caribou_smi_channel_2400 : caribou_smi_channel_900;
    return c;
}

void bb_close(void *h)
{
    bb_ctx *c = (bb_ctx *)h;
    if (!c) return;
    free(c->sys.smi.write_temp_buffer);
    free(c);
}

int bb_read(void *h, int16_t *iq, uint8_t *meta, size_t length)
{
    bb_ctx *c = (bb_ctx *)h;
    return cariboulite_radio_read_samples(&c->radio, (cariboulite_sample_complex_int16 *)iq, (cariboulite_sample_meta *)meta, length);
}

int bb_write(void *h, int16_t *iq, size_t length)
{
    bb_ctx *c = (bb_ctx *)h;
    return cariboulite_radio_write_samples(&c->radio, (cariboulite_sample_complex_int16 *)iq, length);
}

size_t bb_mtu(void *h) { return cariboulite_radio_get_native_mtu_size_samples(&((bb_ctx *)h)->radio); }
