/*
 * cl_oracle.c -- CPU restatement of the CaribouLite host sample-stream path.
 *
 * TEST INFRASTRUCTURE ONLY (see cl_oracle.h): the checker for tests/ and
 * smoke(), and the timed "port" CPU baseline of bench.py.  Never shipped,
 * never on the product path.
 *
 * Pinning status:
 *   - orc_find_buffer_offset / orc_rx_data_analyze / orc_smi_read /
 *     orc_generate_data(ORC_TX_AS_WRITTEN): bit-exact against the compiled
 *     reference (oracle/_ref) and tests/golden/smi_*.npz.
 *   - orc_generate_data(ORC_TX_DOCUMENTED): the reference ignores its inputs
 *     (caribou_smi.c:700-701); pinned by the documented layout
 *     (caribou_smi.c:693-696) and cross-checked through the FPGA parser
 *     restatement orc_fpga_tx_parse (firmware/smi_ctrl.v:194-243).
 *   - conversions: 4-line loops restated from CaribouliteStream.cpp; exact.
 *   - IIR: arithmetic lives in berndporr/iir1 (un-vendored git submodule,
 *     .gitmodules:1-3, no pinned commit) -> PARITY UNPINNED by the reference;
 *     restates iir1's published algorithm, cross-checked vs scipy.signal.
 *   - FIR / resampler / FM / CW: no reference code -> PARITY UNPINNED by the
 *     reference; pinned by float64 scipy fixtures.
 */
#include "cl_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================== */
/* RX integer stages                                                        */
/* ======================================================================== */

static inline uint32_t ld_u32_le(const uint8_t *p)
{
    /* the reference does an unaligned uint32 load on a little-endian host
     * (caribou_smi.c:251-254) */
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

/* caribou_smi.c:235-292 (debug_mode == caribou_smi_none branch) */
int orc_find_buffer_offset(const uint8_t *buffer, size_t len)
{
    if (len <= ORC_BYTES_PER_SAMPLE * 4) return 0;
    for (size_t offs = 0; offs < len - ORC_BYTES_PER_SAMPLE * 4; offs++) {
        int ok = 1;
        for (int w = 0; w < 4 && ok; w++)
            ok = (ld_u32_le(buffer + offs + 4 * w) & 0xC001C000u) == 0x80004000u;
        if (ok) return (int)offs;
    }
    return -1;
}

static inline int16_t sx13(uint32_t v)
{
    /* caribou_smi.c:356-357: if (v >= 0x1000) v -= 0x2000 */
    int16_t r = (int16_t)(v & 0x1FFF);
    if (r >= (int16_t)0x1000) r -= (int16_t)0x2000;
    return r;
}

/* caribou_smi.c:295-393 */
int orc_rx_data_analyze(int channel, const uint8_t *data, size_t data_length,
                        int16_t *iq_out, uint8_t *meta_out)
{
    int offs = orc_find_buffer_offset(data, data_length);
    if (offs < 0) return -1;

    int shortening = (offs > 0) ? (offs / ORC_BYTES_PER_SAMPLE + 1) : 0; /* :319 */
    size_t actual_length = data_length - (size_t)shortening * ORC_BYTES_PER_SAMPLE;
    const uint8_t *words = data + offs;
    size_t n = actual_length / ORC_BYTES_PER_SAMPLE;
    size_t i;

    for (i = 0; i < n; i++) {
        uint32_t s = ld_u32_le(words + 4 * i);
        if (meta_out) meta_out[i] = (uint8_t)(s & 1u);
        if (iq_out) {
            int16_t lo = sx13(s >> 1);  /* bits 13:1  */
            int16_t hi = sx13(s >> 17); /* bits 29:17 */
            if (channel != ORC_CH_HIF) { /* S1G :342-359: q = low field, i = high field */
                iq_out[2 * i + 0] = hi;
                iq_out[2 * i + 1] = lo;
            } else {                     /* HiF :361-378: i = low field, q = high field */
                iq_out[2 * i + 0] = lo;
                iq_out[2 * i + 1] = hi;
            }
        }
    }
    /* :382-389 one extrapolated sample when the buffer was re-synchronised.
     * The reference dereferences cmplx_vec unconditionally here and reads
     * slots i-1, i-2; it is only defined for iq_out != NULL and n >= 2. */
    if (shortening > 0 && iq_out && n >= 2) {
        for (int c = 0; c < 2; c++) {
            int a = iq_out[2 * (i - 1) + c], b = iq_out[2 * (i - 2) + c];
            iq_out[2 * i + c] = (int16_t)(110 * a / 100 - b / 10);
        }
    }
    return offs;
}

static size_t src_read(orc_byte_source *s, uint8_t *dst, size_t n)
{
    size_t avail = s->len - s->pos;
    if (n > avail) n = avail;
    if (s->max_read && n > s->max_read) n = s->max_read;
    memcpy(dst, s->data + s->pos, n);
    s->pos += n;
    return n;
}

/* caribou_smi.c:632-682.  Return: samples "read" (read_so_far), or -3 on
 * sync failure.  A drained source behaves like the poll timeout (:657-661). */
int orc_smi_read(orc_byte_source *src, int channel, int16_t *iq, uint8_t *meta,
                 size_t length_samples, size_t native_batch_len)
{
    size_t left = length_samples * ORC_BYTES_PER_SAMPLE, read_so_far = 0;
    uint8_t *tmp = (uint8_t *)malloc(native_batch_len + 1024);
    while (left) {
        int16_t *so = iq ? iq + 2 * read_so_far : NULL;
        uint8_t *mo = meta ? meta + read_so_far : NULL;
        size_t cur = left > native_batch_len ? native_batch_len : left;
        size_t ret = src_read(src, tmp, cur);
        if (ret == 0) break;
        if (orc_rx_data_analyze(channel, tmp, ret, so, mo) < 0) { free(tmp); return -3; }
        read_so_far += ret / ORC_BYTES_PER_SAMPLE;
        left -= ret;
    }
    free(tmp);
    return (int)read_so_far;
}

/* ======================================================================== */
/* TX integer stages                                                        */
/* ======================================================================== */

/* caribou_smi.c:684-717 */
void orc_generate_data(int mode, const int16_t *iq, size_t n_samples, uint8_t *out)
{
    for (size_t i = 0; i < n_samples; i++) {
        int32_t ii, qq;
        if (mode == ORC_TX_AS_WRITTEN) { ii = 0xFFFF; qq = 0; } /* :700-701 */
        else { ii = iq[2 * i]; qq = iq[2 * i + 1]; }
        ii &= 0x1FFF; qq &= 0x1FFF;
        uint32_t s = 0x7; s <<= 5;             /* SOF | MODEM_TX_CTRL | COND_TX_CTRL :684-686,705 */
        s |= (uint32_t)(ii >> 8) & 0x1F; s <<= 8;
        s |= (uint32_t)(ii >> 1) & 0x7F; s <<= 2;
        s |= (uint32_t)(ii & 0x1);       s <<= 6;
        s |= (uint32_t)(qq >> 7) & 0x3F; s <<= 8;
        s |= (uint32_t)(qq & 0x7F);
        /* samples[i] = __builtin_bswap32(s) on a little-endian host :713 */
        out[4 * i + 0] = (uint8_t)(s >> 24);
        out[4 * i + 1] = (uint8_t)(s >> 16);
        out[4 * i + 2] = (uint8_t)(s >> 8);
        out[4 * i + 3] = (uint8_t)(s);
    }
}

/* firmware/smi_ctrl.v:178-254 byte parser state machine (the debug counter
 * override of :245-246 is not modelled: this is the documented behaviour). */
size_t orc_fpga_tx_parse(const uint8_t *bytes, size_t n_bytes, uint32_t *words_out)
{
    int state = 0; uint32_t r = 0, txc = 0; size_t nw = 0;
    for (size_t k = 0; k < n_bytes; k++) {
        uint8_t b = bytes[k];
        switch (state) {
        case 0:
            if (b & 0x80) {
                r = 0x80000000u; txc = (b >> 6) & 1u;
                r |= (uint32_t)(b & 0x1F) << 25; state = 1;
            } else words_out[nw++] = 0; /* unsynchronised: push a zero word */
            break;
        case 1:
            if (!(b & 0x80)) { r |= (uint32_t)(b & 0x7F) << 18; state = 2; } else state = 0;
            break;
        case 2:
            if (!(b & 0x80)) {
                r |= (uint32_t)((b >> 6) & 1u) << 17; r |= txc << 16; r |= 0x1u << 14;
                r |= (uint32_t)(b & 0x3F) << 8; state = 3;
            } else state = 0;
            break;
        default:
            if (!(b & 0x80)) words_out[nw++] = (r & 0xFFFFFF00u) | ((uint32_t)(b & 0x7F) << 1);
            state = 0;
            break;
        }
    }
    return nw;
}

/* ======================================================================== */
/* Soapy stream conversions                                                 */
/* ======================================================================== */

/* CaribouliteStream.cpp:315-321 */
void orc_cs16_to_cf32(const int16_t *in, float *out, size_t n)
{
    const float max_val = 4096.0f;
    for (size_t i = 0; i < 2 * n; i++) out[i] = (float)in[i] / max_val;
}
/* CaribouliteStream.cpp:337-343 */
void orc_cs16_to_cf64(const int16_t *in, double *out, size_t n)
{
    for (size_t i = 0; i < 2 * n; i++) out[i] = (double)in[i] / 4096.0;
}
/* CaribouliteStream.cpp:360-364 */
void orc_cs16_to_cs8(const int16_t *in, int8_t *out, size_t n)
{
    for (size_t i = 0; i < 2 * n; i++) out[i] = (int8_t)((in[i] >> 5) & 0x00FF);
}
/* float -> int16 as x86-64 gcc compiles "(int16_t)(f * 4096.0f)": cvttss2si
 * to int32 (0x80000000 when out of range / NaN), then the low 16 bits.  In
 * range this is plain truncation toward zero; out of range it is UB in C and
 * this is the behaviour the reference binary has on the build host. */
static inline int16_t f2i16(double v)
{
    int32_t t;
    if (!(v > -2147483649.0 && v < 2147483648.0)) t = (int32_t)0x80000000u;
    else t = (int32_t)v;
    return (int16_t)(uint16_t)((uint32_t)t & 0xFFFFu);
}
/* CaribouliteStream.cpp:199-212 */
void orc_cf32_to_cs16(const float *in, int16_t *out, size_t n)
{
    const float max_val = 4096.0f;
    for (size_t i = 0; i < 2 * n; i++) out[i] = f2i16((double)(in[i] * max_val));
}
/* CaribouliteStream.cpp:215-228 */
void orc_cf64_to_cs16(const double *in, int16_t *out, size_t n)
{
    for (size_t i = 0; i < 2 * n; i++) out[i] = f2i16(in[i] * 4096.0);
}
/* CaribouliteStream.cpp:232-244 */
void orc_cs8_to_cs16(const int8_t *in, int16_t *out, size_t n)
{
    for (size_t i = 0; i < 2 * n; i++) out[i] = (int16_t)(((int16_t)in[i]) << 5);
}

/* ======================================================================== */
/* IIR -- iir1 Butterworth LowPass<N> (call sites CaribouliteStream.cpp:    */
/* 85-91, 295-296; DIG_FILT_ORDER = 6 CaribouliteStream.hpp:24).            */
/* Algorithm restated from iir1's published design (v1.9.x):                */
/*   analog prototype poles  p_i = exp(j(pi/2 + (2i+1)pi/(2N))), zeros inf  */
/*   low-pass bilinear map   z = (1 + f p)/(1 - f p), f = tan(pi fc/fs)     */
/*   one biquad per conjugate pair: a = [1,-2Re(p),|p|^2], b = [1,2,1]      */
/*   overall gain normalised to 1 at DC, folded into the FIRST stage's b    */
/*   evaluation: cascade of Direct-Form-II sections in double.              */
/* ======================================================================== */
void orc_iir_butter_lowpass(orc_iir *f, int order, double fs, double fc)
{
    const double pi = 3.14159265358979323846;
    int pairs = order / 2;
    double k = tan(pi * fc / fs);
    memset(f, 0, sizeof(*f));
    f->n_stages = pairs;
    double gain_re = 1.0, gain_im = 0.0; /* response at DC (real) */
    for (int i = 0; i < pairs; i++) {
        double th = pi / 2 + (2 * i + 1) * pi / (2.0 * order);
        double pr = cos(th), pim = sin(th);
        /* z = (1 + k p) / (1 - k p) */
        double nr = 1 + k * pr, ni = k * pim, dr = 1 - k * pr, di = -k * pim;
        double den = dr * dr + di * di;
        double zr = (nr * dr + ni * di) / den, zi = (ni * dr - nr * di) / den;
        f->a1[i] = -2 * zr; f->a2[i] = zr * zr + zi * zi;
        f->b0[i] = 1; f->b1[i] = 2; f->b2[i] = 1; /* double zero at z = -1 */
        double h = (f->b0[i] + f->b1[i] + f->b2[i]) / (1 + f->a1[i] + f->a2[i]);
        gain_re *= h;
    }
    (void)gain_im;
    double scale = 1.0 / fabs(gain_re);
    f->b0[0] *= scale; f->b1[0] *= scale; f->b2[0] *= scale;
}
void orc_iir_reset(orc_iir *f)
{
    memset(f->v1, 0, sizeof f->v1); memset(f->v2, 0, sizeof f->v2);
}
double orc_iir_step(orc_iir *f, double in)
{
    double out = in;
    for (int s = 0; s < f->n_stages; s++) {
        double w = out - f->a1[s] * f->v1[s] - f->a2[s] * f->v2[s];
        out = f->b0[s] * w + f->b1[s] * f->v1[s] + f->b2[s] * f->v2[s];
        f->v2[s] = f->v1[s]; f->v1[s] = w;
    }
    return out;
}
/* CaribouliteStream.cpp:291-298: buffer[i].i = (int16_t)filter_i->filter((float)buffer[i].i)
 * iir1's filter<Sample> returns static_cast<Sample>(double): the result is
 * rounded to float before the int16 truncation. */
void orc_iir_apply_cs16(orc_iir *fi, orc_iir *fq, int16_t *iq, size_t n)
{
    for (size_t k = 0; k < n; k++) {
        iq[2 * k + 0] = f2i16((double)(float)orc_iir_step(fi, (double)(float)iq[2 * k + 0]));
        iq[2 * k + 1] = f2i16((double)(float)orc_iir_step(fq, (double)(float)iq[2 * k + 1]));
    }
}

/* ======================================================================== */
/* Build-defined float stages (parity unpinned)    */
/* ======================================================================== */

/* y[n] = sum_{k<T} h[k] x[n-k], complex x (fp32), real h (fp32), fp64 sum,
 * zero initial history, T-1 samples carried across calls. */
void orc_fir_f64(orc_fir *f, const float *x, size_t n, double *y)
{
    int T = f->n_taps, H = T - 1;
    for (size_t i = 0; i < n; i++) {
        double ar = 0, ai = 0;
        for (int k = 0; k < T; k++) {
            long j = (long)i - k;
            double xr, xi;
            if (j >= 0) { xr = x[2 * j]; xi = x[2 * j + 1]; }
            else { xr = f->hist[2 * (H + j)]; xi = f->hist[2 * (H + j) + 1]; }
            ar += (double)f->taps[k] * xr; ai += (double)f->taps[k] * xi;
        }
        y[2 * i] = ar; y[2 * i + 1] = ai;
    }
    /* carry the last H inputs */
    if (H > 0) {
        if (n >= (size_t)H) {
            for (int j = 0; j < 2 * H; j++) f->hist[j] = x[2 * (n - H) + j];
        } else {
            memmove(f->hist, f->hist + 2 * n, sizeof(double) * 2 * (H - n));
            for (size_t j = 0; j < 2 * n; j++) f->hist[2 * (H - n) + j] = x[j];
        }
    }
}

/* y[i] for i in [0, n) reading x[i-k] directly (x must have T-1 valid samples
 * before index 0).  Blocks of 8 complex outputs = 16 floats so that gcc turns
 * the inner loop into two 8-wide FMAs per tap. */
typedef float orc_v8f __attribute__((vector_size(32), aligned(4)));   /* unaligned 8-float vector */

static void fir_f32_contig(const float *taps, int T, const float *x, size_t n, float *y)
{
    size_t i = 0;
    for (; i + 16 <= n; i += 16) {                   /* 16 complex outputs = 4 x 8 floats */
        orc_v8f a0 = {0, 0, 0, 0, 0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        const float *px = x + 2 * (long)i;
        for (int k = T - 1; k >= 0; k--) {           /* descending k, like the HIP kernels */
            const float h = taps[k];
            a0 += h * *(const orc_v8f *)(px - 2 * k);
            a1 += h * *(const orc_v8f *)(px - 2 * k + 8);
            a2 += h * *(const orc_v8f *)(px - 2 * k + 16);
            a3 += h * *(const orc_v8f *)(px - 2 * k + 24);
        }
        *(orc_v8f *)(y + 2 * i) = a0;
        *(orc_v8f *)(y + 2 * i + 8) = a1;
        *(orc_v8f *)(y + 2 * i + 16) = a2;
        *(orc_v8f *)(y + 2 * i + 24) = a3;
    }
    for (; i < n; i++) {
        float ar = 0, ai = 0;
        for (int k = T - 1; k >= 0; k--) {
            ar += taps[k] * x[2 * ((long)i - k)]; ai += taps[k] * x[2 * ((long)i - k) + 1];
        }
        y[2 * i] = ar; y[2 * i + 1] = ai;
    }
}

void orc_fir_f32(const float *taps, int T, float *hist, const float *x, size_t n, float *y)
{
    int H = T - 1;
    /* head: outputs that reach into the carried history */
    size_t head = n < (size_t)H ? n : (size_t)H;
    for (size_t i = 0; i < head; i++) {
        float ar = 0, ai = 0;
        for (int k = T - 1; k >= 0; k--) {
            long j = (long)i - k;
            float xr, xi;
            if (j >= 0) { xr = x[2 * j]; xi = x[2 * j + 1]; }
            else { xr = hist[2 * (H + j)]; xi = hist[2 * (H + j) + 1]; }
            ar += taps[k] * xr; ai += taps[k] * xi;
        }
        y[2 * i] = ar; y[2 * i + 1] = ai;
    }
    if (n > head) fir_f32_contig(taps, T, x + 2 * head, n - head, y + 2 * head);
    if (H > 0) {
        if (n >= (size_t)H) memcpy(hist, x + 2 * (n - H), sizeof(float) * 2 * H);
        else {
            memmove(hist, hist + 2 * n, sizeof(float) * 2 * (H - n));
            memcpy(hist + 2 * (H - n), x, sizeof(float) * 2 * n);
        }
    }
}

/* scipy.signal.upfirdn(h, x, up=L, down=M) semantics, streaming:
 *   out[m] = sum_i h[p + i L] x[b - i],  t = m M, b = t / L, p = t % L
 * for every m with b < n_total (the first ceil(n L / M) outputs of upfirdn). */
size_t orc_resamp_f64(orc_resamp *r, const double *x, size_t n, double *y)
{
    int L = r->L, M = r->M, K = (r->n_taps + L - 1) / L, H = r->hist_len;
    uint64_t n0 = r->n_in, n1 = n0 + n;
    uint64_t m0 = (n0 * L + M - 1) / M, m1 = (n1 * L + M - 1) / M;
    size_t no = 0;
    for (uint64_t m = m0; m < m1; m++) {
        uint64_t t = m * M, b = t / L; int p = (int)(t % L);
        double ar = 0, ai = 0;
        for (int i = 0; i < K; i++) {
            int k = p + i * L;
            if (k >= r->n_taps) break;
            long j = (long)(b - n0) - i; /* index into this chunk */
            double xr, xi;
            if (j >= 0) { xr = x[2 * j]; xi = x[2 * j + 1]; }
            else if (H + j >= 0) { xr = r->hist[2 * (H + j)]; xi = r->hist[2 * (H + j) + 1]; }
            else { xr = xi = 0; }
            ar += (double)r->taps[k] * xr; ai += (double)r->taps[k] * xi;
        }
        y[2 * no] = ar; y[2 * no + 1] = ai; no++;
    }
    if (H > 0) {
        if (n >= (size_t)H) memcpy(r->hist, x + 2 * (n - H), sizeof(double) * 2 * H);
        else {
            memmove(r->hist, r->hist + 2 * n, sizeof(double) * 2 * (H - n));
            memcpy(r->hist + 2 * (H - n), x, sizeof(double) * 2 * n);
        }
    }
    r->n_in = n1;
    return no;
}

size_t orc_resamp_f32(const float *taps, int n_taps, int L, int M, float *hist,
                      uint64_t *n_in, const float *x, size_t n, float *y)
{
    int K = (n_taps + L - 1) / L, H = K - 1;
    uint64_t n0 = *n_in, n1 = n0 + n;
    uint64_t m0 = (n0 * L + M - 1) / M, m1 = (n1 * L + M - 1) / M;
    size_t no = 0;
    for (uint64_t m = m0; m < m1; m++) {
        uint64_t t = m * M, b = t / L; int p = (int)(t % L);
        float ar = 0, ai = 0;
        long jb = (long)(b - n0);
        if (jb >= H) {
            const float *px = x + 2 * jb;
            for (int i = 0; i < K; i++) {
                int k = p + i * L;
                if (k >= n_taps) break;
                ar += taps[k] * px[-2 * i]; ai += taps[k] * px[-2 * i + 1];
            }
        } else {
            for (int i = 0; i < K; i++) {
                int k = p + i * L;
                if (k >= n_taps) break;
                long j = jb - i; float xr, xi;
                if (j >= 0) { xr = x[2 * j]; xi = x[2 * j + 1]; }
                else if (H + j >= 0) { xr = hist[2 * (H + j)]; xi = hist[2 * (H + j) + 1]; }
                else { xr = xi = 0; }
                ar += taps[k] * xr; ai += taps[k] * xi;
            }
        }
        y[2 * no] = ar; y[2 * no + 1] = ai; no++;
    }
    if (H > 0) {
        if (n >= (size_t)H) memcpy(hist, x + 2 * (n - H), sizeof(float) * 2 * H);
        else {
            memmove(hist, hist + 2 * n, sizeof(float) * 2 * (H - n));
            memcpy(hist + 2 * (H - n), x, sizeof(float) * 2 * n);
        }
    }
    *n_in = n1;
    return no;
}

/* y[n] = atan2(Im z, Re z), z = x[n] conj(x[n-1]); x[-1] = 0 -> y[0] = 0 */
void orc_fm_demod_f64(double prev[2], const double *x, size_t n, double *y)
{
    double pr = prev[0], pi_ = prev[1];
    for (size_t i = 0; i < n; i++) {
        double a = x[2 * i], b = x[2 * i + 1];
        double re = a * pr + b * pi_, im = b * pr - a * pi_;
        /* a13: z = 0 (e.g. x[-1] = 0) demodulates to 0 whatever the signs of its zeros (np.angle(0) = 0) */
        y[i] = (re == 0.0 && im == 0.0) ? 0.0 : atan2(im, re);
        pr = a; pi_ = b;
    }
    prev[0] = pr; prev[1] = pi_;
}
void orc_fm_demod_f32(float prev[2], const float *x, size_t n, float *y)
{
    float pr = prev[0], pi_ = prev[1];
    for (size_t i = 0; i < n; i++) {
        float a = x[2 * i], b = x[2 * i + 1];
        float re = a * pr + b * pi_, im = b * pr - a * pi_;
        y[i] = (re == 0.0f && im == 0.0f) ? 0.0f : atan2f(im, re);
        pr = a; pi_ = b;
    }
    prev[0] = pr; prev[1] = pi_;
}

/* phi[n] = wrap(phi[n-1] + 2 pi kf m[n] / fs) in fp64, wrap to (-pi, pi];
 * out = (cos phi, sin phi) rounded to fp32. */
void orc_fm_mod_f64(double *phase, double kf, double fs, const float *m, size_t n, float *out)
{
    const double pi = 3.14159265358979323846, w = 2 * pi * kf / fs;
    double ph = *phase;
    for (size_t i = 0; i < n; i++) {
        ph += w * (double)m[i];
        if (ph > pi || ph <= -pi) { ph -= 2 * pi * floor((ph + pi) / (2 * pi)); if (ph <= -pi) ph += 2 * pi; }
        out[2 * i] = (float)cos(ph); out[2 * i + 1] = (float)sin(ph);
    }
    *phase = ph;
}
/* examples/cpp_api/sync_tx_api/main.cpp:42-55 style tone, phase kept in fp64:
 * I = cos(2 pi f t), Q = sin(2 pi f t), t = n / fs. */
void orc_cw_tone(double *phase, double f, double fs, size_t n, float *out)
{
    const double pi = 3.14159265358979323846, w = 2 * pi * f / fs;
    double ph = *phase;
    for (size_t i = 0; i < n; i++) {
        out[2 * i] = (float)cos(ph); out[2 * i + 1] = (float)sin(ph);
        ph += w;
        if (ph > pi) ph -= 2 * pi; else if (ph <= -pi) ph += 2 * pi;
    }
    *phase = ph;
}

/* ======================================================================== */
/* The fp32 CPU pipe (bench.py "port" cpu_baseline)                         */
/* ======================================================================== */
size_t orc_rx_pipe_f32(int channel, const uint8_t *bytes, size_t n_bytes,
                       const float *fir_taps, int fir_n, float *fir_hist,
                       const float *rs_taps, int rs_n, int L, int M, float *rs_hist,
                       uint64_t *rs_n_in, int16_t *tmp_iq, float *tmp_cf32,
                       float *tmp_fir, float *out)
{
    /* same pass structure as the reference read path: analyse -> convert ->
     * (new stages) filter -> resample, one pass per stage */
    if (orc_rx_data_analyze(channel, bytes, n_bytes, tmp_iq, NULL) < 0) return 0;
    size_t n = n_bytes / ORC_BYTES_PER_SAMPLE;
    orc_cs16_to_cf32(tmp_iq, tmp_cf32, n);
    orc_fir_f32(fir_taps, fir_n, fir_hist, tmp_cf32, n, tmp_fir);
    return orc_resamp_f32(rs_taps, rs_n, L, M, rs_hist, rs_n_in, tmp_fir, n, out);
}

/* All-cores variant of the same pipe over ONE long stream from zero state:
 * native chunks are analysed/converted in parallel (each chunk re-synchronised
 * on its own, caribou_smi.c:643-679), then FIR and resampler run in parallel
 * over blocks of the contiguous CF32 sequence.  x_buf: 2*(n+T) floats,
 * y_buf: 2*(n+8) floats, out: 2*ceil(n*L/M) floats.  Returns outputs. */
size_t orc_rx_pipe_f32_mt(int channel, const uint8_t *bytes, size_t n_bytes, size_t native_batch_len,
                          const float *fir_taps, int fir_n, const float *rs_taps, int rs_n, int L, int M,
                          int16_t *iq_buf, float *x_buf, float *y_buf, float *out, int n_threads)
{
    const size_t n = n_bytes / ORC_BYTES_PER_SAMPLE;
    const long n_chunks = (long)((n_bytes + native_batch_len - 1) / native_batch_len);
    const int H = fir_n - 1, K = (rs_n + L - 1) / L, HR = K - 1;
    float *x = x_buf + 2 * H;                /* x[-H..-1] = zero history */
    float *y = y_buf + 2 * HR;
    memset(x_buf, 0, sizeof(float) * 2 * H);
    memset(y_buf, 0, sizeof(float) * 2 * HR);
    int bad = 0;
#pragma omp parallel for num_threads(n_threads) schedule(dynamic, 1) reduction(|:bad)
    for (long c = 0; c < n_chunks; c++) {
        size_t off = (size_t)c * native_batch_len;
        size_t len = n_bytes - off < native_batch_len ? n_bytes - off : native_batch_len;
        int16_t *iq = iq_buf + 2 * (off / 4);
        if (orc_rx_data_analyze(channel, bytes + off, len, iq, NULL) < 0) { bad = 1; continue; }
        orc_cs16_to_cf32(iq, x + 2 * (off / 4), len / 4);
    }
    if (bad) return 0;
    const size_t blk = 16384;
    const long n_blk = (long)((n + blk - 1) / blk);
#pragma omp parallel for num_threads(n_threads) schedule(dynamic, 4)
    for (long b = 0; b < n_blk; b++) {
        size_t i0 = (size_t)b * blk, cnt = n - i0 < blk ? n - i0 : blk;
        fir_f32_contig(fir_taps, fir_n, x + 2 * i0, cnt, y + 2 * i0);
    }
    const size_t n_out = (n * (size_t)L + M - 1) / M;
    const size_t oblk = 16384;
    const long n_oblk = (long)((n_out + oblk - 1) / oblk);
#pragma omp parallel for num_threads(n_threads) schedule(dynamic, 4)
    for (long b = 0; b < n_oblk; b++) {
        size_t m0 = (size_t)b * oblk, m1 = m0 + oblk < n_out ? m0 + oblk : n_out;
        for (size_t m = m0; m < m1; m++) {
            uint64_t t = (uint64_t)m * M; long bi = (long)(t / L); int p = (int)(t % L);
            float ar = 0, ai = 0;
            for (int i = 0; i < K; i++) {
                int k = p + i * L;
                if (k >= rs_n) break;
                ar += rs_taps[k] * y[2 * (bi - i)]; ai += rs_taps[k] * y[2 * (bi - i) + 1];
            }
            out[2 * m] = ar; out[2 * m + 1] = ai;
        }
    }
    return n_out;
}

/* ======================================================================== */
/* Link-integrity (debug) modes -- SURVEY.md section 8(f) rank 3            */
/* ======================================================================== */
static unsigned popcount32(uint32_t x) { unsigned c = 0; while (x) { c += x & 1u; x >>= 1; } return c; }

/* smi_utils.c:220-224 */
uint8_t orc_lfsr(uint8_t n)
{
    uint8_t bit = ((n >> 2) ^ (n >> 3)) & 1;
    return (uint8_t)((n >> 1) | (bit << 7));
}

/* caribou_smi.c:235-292, debug branches (:266-283): push/pull search the first byte offset whose
 * unaligned word is within 3 bit flips of 0xABCDEF01; lfsr mode does not search. */
int orc_debug_find_offset(int mode, const uint8_t *buffer, size_t len)
{
    if (len <= ORC_BYTES_PER_SAMPLE * 4) return 0;
    if (mode == ORC_DEBUG_LFSR) return 0;
    for (size_t offs = 0; offs < len - ORC_BYTES_PER_SAMPLE; offs++)
        if (popcount32(ld_u32_le(buffer + offs) ^ 0xABCDEF01u) < 4) return (int)offs;
    return -1;
}

/* caribou_smi_rx_data_analyze (:295-330) + caribou_smi_anayze_smi_debug (:172-215) for one read()
 * chunk; the wall-clock bitrate EMA is orc_bitrate_ema below (the caller supplies the clock).  Returns offs (or -1). */
int orc_debug_analyze(orc_debug_data *d, int mode, const uint8_t *data, size_t len)
{
    int offs = orc_debug_find_offset(mode, data, len);
    if (offs < 0) return -1;
    size_t shortening = offs > 0 ? (size_t)(offs / ORC_BYTES_PER_SAMPLE + 1) : 0;
    size_t alen = len - shortening * ORC_BYTES_PER_SAMPLE;
    const uint8_t *p = data + offs;
    uint32_t cur = 0;
    if (mode == ORC_DEBUG_LFSR) {
        for (size_t i = 0; i < alen; i++) {
            if (p[i] != orc_lfsr(d->last_correct_byte) || p[i] == 0) { d->error_accum_counter++; cur++; }
            d->last_correct_byte = p[i];
        }
    } else {
        for (size_t i = 0; i < alen / 4; i++)
            if (ld_u32_le(p + 4 * i) != 0xABCDEF01u) { d->error_accum_counter += 4; cur += 4; }
    }
    d->cur_err_cnt = cur;
    d->error_rate = d->error_rate * 0.9 + (double)cur / (double)alen * 0.1;
    if (d->error_rate < 1e-8) d->error_rate = 0.0;
    return offs;
}

/* smi_calculate_performance (caribou_smi/smi_utils.c:233-244) with the two clock readings as arguments: the reference
 * names the difference elapsed_us but forms it in SECONDS (tv_sec difference + tv_usec difference / 1e6), so the figure is
 * bytes * 8 / seconds / 1e6 = Mbit/s, blended 0.98 : 0.02 into the running value.  (caribou_smi.c:210 calls it once per
 * analysed chunk with the chunk's length.) */
__attribute__((optimize("fp-contract=off")))     /* the reference's plain build rounds the product and the sum separately */
double orc_bitrate_ema(size_t bytes, long old_sec, long old_usec, long cur_sec, long cur_usec, double old_mbps)
{
    double elapsed_us = (cur_sec - old_sec) + ((double)(cur_usec - old_usec)) / 1000000.0;
    double speed_mbps = (double)(bytes * 8) / elapsed_us / 1e6;
    return old_mbps * 0.98 + speed_mbps * 0.02;
}

/* ======================================================================== */
/* circular_buffer<T> (datatypes/circular_buffer.h:16-164), non-blocking     */
/* semantics restated on 32-bit elements -- SURVEY.md section 8(f) rank 2    */
/* ======================================================================== */
void orc_ring_init(orc_ring *r, size_t size, int override_write, uint32_t *storage)
{
    size_t cap = 1;                       /* :21-25 next power of two */
    while (cap < size) cap <<= 1;
    r->buf = storage; r->max_size = cap; r->head = r->tail = 0; r->override_write = override_write;
}
size_t orc_ring_capacity_for(size_t size) { size_t cap = 1; while (cap < size) cap <<= 1; return cap; }
size_t orc_ring_size(const orc_ring *r) { return r->head - r->tail; }

/* :37-62 */
size_t orc_ring_put(orc_ring *r, const uint32_t *data, size_t length)
{
    size_t sz = r->head - r->tail;
    if ((r->max_size - sz) < length && r->override_write) r->tail += length - (r->max_size - sz);   /* drop the oldest */
    size_t len = length < r->max_size - r->head + r->tail ? length : r->max_size - r->head + r->tail;
    size_t hi = r->head & (r->max_size - 1);
    size_t l = len < r->max_size - hi ? len : r->max_size - hi;
    memcpy(r->buf + hi, data, l * sizeof(uint32_t));
    memcpy(r->buf, data + l, (len - l) * sizeof(uint32_t));
    r->head += len;
    return len;
}

/* :64-93 with the wait already over: block_read returns 0 unless `length` items are present */
size_t orc_ring_get(orc_ring *r, uint32_t *data, size_t length, int block_read)
{
    size_t sz = r->head - r->tail;
    if (block_read && sz < length) return 0;
    size_t len = length < sz ? length : sz;
    size_t ti = r->tail & (r->max_size - 1);
    size_t l = len < r->max_size - ti ? len : r->max_size - ti;
    if (data) {
        memcpy(data, r->buf + ti, l * sizeof(uint32_t));
        memcpy(data + l, r->buf, (len - l) * sizeof(uint32_t));
    }
    r->tail += len;
    return len;
}

/* ---- pps tags: the GNU Radio source block's work()  (software/gr-caribouLite/lib/caribouLiteSource_impl.cc:113-119) ----
 * `for (i = 0; i < read_samples; i++) if (out_meta[i] == 1) add_item_tag(0, i, "pps", true)`: the offsets it tags, in the
 * order it tags them.  Returns how many there are; the first `cap` are stored. */
size_t orc_sync_tags(const uint8_t *meta, size_t n, uint32_t *idx, size_t cap)
{
    size_t k = 0;
    for (size_t i = 0; i < n; i++)
        if (meta[i] == 1) {
            if (k < cap) idx[k] = (uint32_t)i;
            k++;
        }
    return k;
}
