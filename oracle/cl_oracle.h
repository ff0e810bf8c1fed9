/*
 * cl_oracle.h -- CPU restatement of the CaribouLite host sample-stream path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call it, and only as the checker / the timed CPU baseline.  The product
 * (cariboulite_amd/, include/) never links, imports or executes this code.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference/software/libcariboulite/src unless noted).  Integer stages
 * are pinned bit-exact against the compiled reference (oracle/_ref, built by
 * oracle/Makefile from the sources where they lie) and against the committed
 * fixtures in tests/golden/.  The float stages named by BASELINE.json
 * (FIR / rational resampler / FM / CW) have NO reference implementation, and
 * the IIR arithmetic lives in the un-vendored third-party iir1 library:
 * for those stages this oracle is "parity unpinned" by the reference and is
 * pinned instead by float64 scipy fixtures (oracle/gen_golden.py).
 */
#ifndef CL_ORACLE_H
#define CL_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_BYTES_PER_SAMPLE 4
#define ORC_CH_S1G 0 /* caribou_smi_channel_900  (caribou_smi.h:47) */
#define ORC_CH_HIF 1 /* caribou_smi_channel_2400 (caribou_smi.h:48) */

/* ---- RX integer stages (caribou_smi.c) ---------------------------------- */
int orc_find_buffer_offset(const uint8_t *buffer, size_t len);
int orc_rx_data_analyze(int channel, const uint8_t *data, size_t data_length,
                        int16_t *iq_out, uint8_t *meta_out);

/* A byte source standing in for the /dev/smi fd: read() hands out at most
 * max_read bytes per call (0 = unlimited) until the buffer is exhausted. */
typedef struct {
    const uint8_t *data;
    size_t len;
    size_t pos;
    size_t max_read;
} orc_byte_source;

int orc_smi_read(orc_byte_source *src, int channel, int16_t *iq, uint8_t *meta,
                 size_t length_samples, size_t native_batch_len);

/* ---- TX integer stages -------------------------------------------------- */
#define ORC_TX_DOCUMENTED 0 /* layout of caribou_smi.c:693-696, inputs used      */
#define ORC_TX_AS_WRITTEN 1 /* caribou_smi.c:700-701 as shipped: ii=0xFFFF, qq=0 */
void orc_generate_data(int mode, const int16_t *iq, size_t n_samples, uint8_t *out);
/* FPGA TX byte parser (firmware/smi_ctrl.v:178-254, debug counter override
 * at :245-246 ignored): bytes -> 32-bit modem words in the RX word layout. */
size_t orc_fpga_tx_parse(const uint8_t *bytes, size_t n_bytes, uint32_t *words_out);

/* ---- Soapy stream conversions (soapy_api/CaribouliteStream.cpp) --------- */
void orc_cs16_to_cf32(const int16_t *in, float *out, size_t n_samples);
void orc_cs16_to_cf64(const int16_t *in, double *out, size_t n_samples);
void orc_cs16_to_cs8(const int16_t *in, int8_t *out, size_t n_samples);
void orc_cf32_to_cs16(const float *in, int16_t *out, size_t n_samples);
void orc_cf64_to_cs16(const double *in, int16_t *out, size_t n_samples);
void orc_cs8_to_cs16(const int8_t *in, int16_t *out, size_t n_samples);

/* ---- IIR: Butterworth LowPass<6> as iir1 designs/evaluates it ----------- */
#define ORC_IIR_MAX_STAGES 8
typedef struct {
    int n_stages;
    double b0[ORC_IIR_MAX_STAGES], b1[ORC_IIR_MAX_STAGES], b2[ORC_IIR_MAX_STAGES];
    double a1[ORC_IIR_MAX_STAGES], a2[ORC_IIR_MAX_STAGES];
    double v1[ORC_IIR_MAX_STAGES], v2[ORC_IIR_MAX_STAGES]; /* DF-II state */
} orc_iir;
void orc_iir_butter_lowpass(orc_iir *f, int order, double fs, double fc);
void orc_iir_reset(orc_iir *f);
double orc_iir_step(orc_iir *f, double in);
/* CaribouliteStream.cpp:291-298: in-place on interleaved int16 I/Q. */
void orc_iir_apply_cs16(orc_iir *fi, orc_iir *fq, int16_t *iq, size_t n_samples);

/* ---- Build-defined float stages ----------- */
typedef struct {
    int n_taps;
    const float *taps;
    double *hist; /* 2*(n_taps-1) doubles, interleaved I/Q, oldest first */
} orc_fir;
void orc_fir_f64(orc_fir *f, const float *x, size_t n, double *y);
/* fp32 variant used ONLY as the timed CPU baseline; hist_f32 = 2*(T-1) floats */
void orc_fir_f32(const float *taps, int n_taps, float *hist_f32,
                 const float *x, size_t n, float *y);

typedef struct {
    int L, M, n_taps; /* n_taps = taps of the prototype h_rs */
    const float *taps;
    double *hist;     /* 2*hist_len doubles, interleaved, oldest first */
    int hist_len;     /* ceil(n_taps/L) - 1 ... callers allocate n_taps */
    uint64_t n_in;    /* inputs consumed so far (streaming phase) */
} orc_resamp;
size_t orc_resamp_f64(orc_resamp *r, const double *x, size_t n, double *y);
size_t orc_resamp_f32(const float *taps, int n_taps, int L, int M, float *hist_f32,
                      uint64_t *n_in, const float *x, size_t n, float *y);

void orc_fm_demod_f64(double prev[2], const double *x, size_t n, double *y);
void orc_fm_demod_f32(float prev[2], const float *x, size_t n, float *y);
void orc_fm_mod_f64(double *phase, double kf, double fs, const float *m, size_t n, float *out);
void orc_cw_tone(double *phase, double f, double fs, size_t n, float *out);

/* ---- Whole RX pipe on the CPU: unpack -> /4096 -> FIR -> L/M (fp32) ----- */
/* The "port" cpu_baseline of bench.py: same stage order as the HIP pipe,
 * fp32 arithmetic, separate passes like the reference's own read path. */
size_t orc_rx_pipe_f32(int channel, const uint8_t *bytes, size_t n_bytes,
                       const float *fir_taps, int fir_n, float *fir_hist,
                       const float *rs_taps, int rs_n, int L, int M, float *rs_hist,
                       uint64_t *rs_n_in, int16_t *tmp_iq, float *tmp_cf32,
                       float *tmp_fir, float *out);

/* ---- link-integrity (debug) modes: caribou_smi.h:22-28, caribou_smi.c:172-215,266-283 ---- */
#define ORC_DEBUG_NONE 0
#define ORC_DEBUG_LFSR 1
#define ORC_DEBUG_PUSH 2
#define ORC_DEBUG_PULL 3
typedef struct {
    uint32_t error_accum_counter;
    uint32_t cur_err_cnt;
    uint8_t last_correct_byte;
    double error_rate;
} orc_debug_data;
uint8_t orc_lfsr(uint8_t n);
int orc_debug_find_offset(int mode, const uint8_t *buffer, size_t len);
int orc_debug_analyze(orc_debug_data *d, int mode, const uint8_t *data, size_t len);
double orc_bitrate_ema(size_t bytes, long old_sec, long old_usec, long cur_sec, long cur_usec, double old_mbps);   /* smi_utils.c:233-244 */

/* ---- circular_buffer<T> semantics (datatypes/circular_buffer.h) on 32-bit elements ---- */
typedef struct {
    uint32_t *buf;
    size_t max_size, head, tail;
    int override_write;
} orc_ring;
void orc_ring_init(orc_ring *r, size_t size, int override_write, uint32_t *storage);
size_t orc_ring_capacity_for(size_t size);
size_t orc_ring_size(const orc_ring *r);
size_t orc_ring_put(orc_ring *r, const uint32_t *data, size_t length);
size_t orc_ring_get(orc_ring *r, uint32_t *data, size_t length, int block_read);

/* pps tags of a meta plane: caribouLiteSource_impl.cc:113-119 */
size_t orc_sync_tags(const uint8_t *meta, size_t n, uint32_t *idx, size_t cap);

size_t orc_rx_pipe_f32_mt(int channel, const uint8_t *bytes, size_t n_bytes, size_t native_batch_len,
                          const float *fir_taps, int fir_n, const float *rs_taps, int rs_n, int L, int M,
                          int16_t *iq_buf, float *x_buf, float *y_buf, float *out, int n_threads);

#ifdef __cplusplus
}
#endif
#endif
