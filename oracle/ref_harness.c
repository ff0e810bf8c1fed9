/*
 * ref_harness.c -- thin exported wrappers around the REFERENCE's own
 * caribou_smi.c, compiled from where it lies under /root/reference by
 * oracle/Makefile into oracle/_ref/libref_smi.so (git-ignored).
 *
 * TEST INFRASTRUCTURE ONLY.  This file contains no reference code: it
 * #includes the reference translation unit at build time so that its
 * `static` functions (rx_data_analyze, find_buffer_offset, generate_data)
 * are reachable, and drives caribou_smi_read() from an ordinary fd.
 * Used to pin oracle/cl_oracle.c and to generate tests/golden/ fixtures.
 */
#include "caribou_smi/caribou_smi.c" /* -I$(REF)/software/libcariboulite/src */

#include <fcntl.h>

static void ref_dev_init(caribou_smi_st *dev, int fd, size_t native_batch_len)
{
    memset(dev, 0, sizeof(*dev));
    dev->initialized = 1;
    dev->filedesc = fd;
    dev->native_batch_len = native_batch_len;
    dev->sample_rate = CARIBOU_SMI_SAMPLE_RATE;
    dev->debug_mode = caribou_smi_none;
    dev->read_temp_buffer = malloc(native_batch_len + 1024);
    dev->write_temp_buffer = malloc(native_batch_len + 1024);
}

static void ref_dev_free(caribou_smi_st *dev)
{
    free(dev->read_temp_buffer);
    free(dev->write_temp_buffer);
}

int ref_find_buffer_offset(uint8_t *buffer, size_t len)
{
    caribou_smi_st dev;
    memset(&dev, 0, sizeof dev);
    dev.debug_mode = caribou_smi_none;
    return caribou_smi_find_buffer_offset(&dev, buffer, len);
}

/* channel: 0 = caribou_smi_channel_900 (S1G), 1 = caribou_smi_channel_2400 (HiF) */
int ref_rx_data_analyze(int channel, uint8_t *data, size_t data_length,
                        int16_t *iq_out, uint8_t *meta_out)
{
    caribou_smi_st dev;
    memset(&dev, 0, sizeof dev);
    dev.debug_mode = caribou_smi_none;
    return caribou_smi_rx_data_analyze(&dev, (caribou_smi_channel_en)channel, data, data_length,
                                       (caribou_smi_sample_complex_int16 *)iq_out,
                                       (caribou_smi_sample_meta *)meta_out);
}

/* caribou_smi_read() driven from a regular file holding the byte stream. */
int ref_smi_read_file(const char *path, int channel, int16_t *iq, uint8_t *meta,
                      size_t length_samples, size_t native_batch_len)
{
    int fd = open(path, O_RDONLY);
    if (fd < 0) return -100;
    caribou_smi_st dev;
    ref_dev_init(&dev, fd, native_batch_len);
    int ret = caribou_smi_read(&dev, (caribou_smi_channel_en)channel,
                               (caribou_smi_sample_complex_int16 *)iq,
                               (caribou_smi_sample_meta *)meta, length_samples);
    ref_dev_free(&dev);
    close(fd);
    return ret;
}

/* caribou_smi_read() on a descriptor the caller owns (a pipe, non-blocking or not): the read()/poll()/read() pattern of
 * caribou_smi_timeout_read runs against it as it does against /dev/smi */
int ref_smi_read_fd(int fd, int channel, int16_t *iq, uint8_t *meta, size_t length_samples, size_t native_batch_len)
{
    caribou_smi_st dev;
    ref_dev_init(&dev, fd, native_batch_len);
    int ret = caribou_smi_read(&dev, (caribou_smi_channel_en)channel,
                               (caribou_smi_sample_complex_int16 *)iq,
                               (caribou_smi_sample_meta *)meta, length_samples);
    ref_dev_free(&dev);
    return ret;
}

void ref_generate_data(const int16_t *iq, size_t n_samples, uint8_t *out)
{
    caribou_smi_st dev;
    memset(&dev, 0, sizeof dev);
    caribou_smi_generate_data(&dev, out, n_samples * CARIBOU_SMI_BYTES_PER_SAMPLE,
                              (caribou_smi_sample_complex_int16 *)iq);
}

/* caribou_smi_read() in a debug mode on a regular file: returns the reference's return code and
 * hands back the debug counters it maintains (state carried in/out so calls can be chained). */
int ref_debug_read_file(const char *path, int mode, size_t length_samples, size_t native_batch_len,
                        uint32_t *accum, uint32_t *cur, uint8_t *last_byte, double *error_rate)
{
    int fd = open(path, O_RDONLY);
    if (fd < 0) return -100;
    caribou_smi_st dev;
    ref_dev_init(&dev, fd, native_batch_len);
    dev.debug_mode = (caribou_smi_debug_mode_en)mode;
    dev.debug_data.error_accum_counter = *accum;
    dev.debug_data.cur_err_cnt = *cur;
    dev.debug_data.last_correct_byte = *last_byte;
    dev.debug_data.error_rate = *error_rate;
    gettimeofday(&dev.debug_data.last_time, NULL);
    int16_t *iq = malloc(4 * (length_samples + 8));
    int ret = caribou_smi_read(&dev, caribou_smi_channel_900, (caribou_smi_sample_complex_int16 *)iq, NULL, length_samples);
    *accum = dev.debug_data.error_accum_counter;
    *cur = dev.debug_data.cur_err_cnt;
    *last_byte = dev.debug_data.last_correct_byte;
    *error_rate = dev.debug_data.error_rate;
    free(iq);
    ref_dev_free(&dev);
    close(fd);
    return ret;
}

/* smi_calculate_performance() itself (smi_utils.c:233-244).  It reads the wall clock and leaves the reading in *old_time:
 * the caller gets result AND the (sec, usec) the reference used, and can so check a restatement bit for bit. */
double ref_bitrate_ema(size_t bytes, long old_sec, long old_usec, double old_mbps, long *cur_sec, long *cur_usec)
{
    struct timeval t;
    t.tv_sec = old_sec; t.tv_usec = old_usec;
    const double r = smi_calculate_performance(bytes, &t, old_mbps);
    *cur_sec = (long)t.tv_sec; *cur_usec = (long)t.tv_usec;
    return r;
}
