// clhip_smi.hip -- SMI byte-stream integer stages for gfx950:
//   sync search   (caribou_smi.c:235-292)
//   int13 unpack  (caribou_smi.c:295-393) fused with the Soapy RX conversions
//                 (soapy_api/CaribouliteStream.cpp:304-367)
//   TX pack       (caribou_smi.c:684-717)
//   CS16 <-> CF32 / CF64 / CS8 conversions (CaribouliteStream.cpp:199-244)
// All HBM-bound byte/integer work: one 16-byte load per lane, bit-field
// extracts, 16-byte stores; results are bit-exact with the reference.
#include "clhip_common.h"

#define SYNC_MASK 0xC001C000u
#define SYNC_BITS 0x80004000u

// ---------------------------------------------------------------------------
// sync search
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ld_u32_bytes(const uint8_t *p)
{
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

__device__ __forceinline__ bool sync4_bytes(const uint8_t *p)
{
    bool ok = true;
#pragma unroll
    for (int w = 0; w < 4; w++) ok = ok && ((ld_u32_bytes(p + 4 * w) & SYNC_MASK) == SYNC_BITS);
    return ok;
}

// One workgroup per chunk.  Common case (aligned stream): the first test
// passes and the kernel ends after 16 bytes.  Otherwise the chunk is scanned
// in segments with first-match (smallest offset) semantics.
__global__ __launch_bounds__(256) void smi_find_offsets_kernel(
    const uint8_t *__restrict__ bytes, size_t total, size_t stride, size_t chunk_len,
    int n_chunks, int32_t *__restrict__ offs_out)
{
    const int c = blockIdx.x;
    const int tid = threadIdx.x;
    if (c >= n_chunks) return;
    const size_t start = (size_t)c * stride;
    const size_t len = start >= total ? 0 : (chunk_len < total - start ? chunk_len : total - start);
    const uint8_t *p = bytes + start;
    if (len <= 16) {                    // caribou_smi.c:240-243
        if (tid == 0) offs_out[c] = 0;
        return;
    }
    const size_t limit = len - 16;      // candidates: offs in [0, limit)

    __shared__ int s_found;
    if (tid == 0) s_found = sync4_bytes(p) ? 0 : 0x7fffffff;
    __syncthreads();
    if (s_found == 0) {
        if (tid == 0) offs_out[c] = 0;
        return;
    }

    const bool aligned = (((uintptr_t)p) & 3) == 0;
    if (aligned) {
        // word j covers byte offsets 4j..4j+3; the test at offset 4j+s needs
        // aligned words j..j+4 (funnel-shifted by s bytes).
        const uint32_t *w = (const uint32_t *)p;
        const size_t n_words_cand = (limit + 3) / 4;       // words holding candidate offsets
        for (size_t base = 0; base < n_words_cand; base += 256) {
            size_t j = base + tid;
            int best = 0x7fffffff;
            if (j < n_words_cand) {
                // words j..j+3 are always inside the chunk (4j < len-16); the
                // fifth one may straddle its end
                uint32_t v[5];
#pragma unroll
                for (int k = 0; k < 4; k++) v[k] = w[j + k];
                const size_t b4 = 4 * (j + 4);
                if (b4 + 4 <= len) v[4] = w[j + 4];
                else {               // ragged chunk tail: assemble the bytes that exist
                    v[4] = 0;
                    for (int t = 0; t < 4; t++)
                        if (b4 + t < len) v[4] |= (uint32_t)p[b4 + t] << (8 * t);
                }
#pragma unroll
                for (int s = 3; s >= 0; s--) {
                    size_t o = 4 * j + s;
                    bool ok = o < limit;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        uint32_t x = s ? __builtin_amdgcn_alignbyte(v[k + 1], v[k], s) : v[k];
                        ok = ok && ((x & SYNC_MASK) == SYNC_BITS);
                    }
                    if (ok) best = (int)o;
                }
            }
            if (best != 0x7fffffff) atomicMin(&s_found, best);
            __syncthreads();
            const int f = s_found;      // every lane reads before anyone moves on
            __syncthreads();
            if (f != 0x7fffffff) break;
        }
    } else {
        for (size_t base = 0; base < limit; base += 256) {
            size_t o = base + tid;
            if (o < limit && sync4_bytes(p + o)) atomicMin(&s_found, (int)o);
            __syncthreads();
            const int f = s_found;
            __syncthreads();
            if (f != 0x7fffffff) break;
        }
    }
    __syncthreads();
    if (tid == 0) offs_out[c] = (s_found == 0x7fffffff) ? -1 : s_found;
}

extern "C" int clhip_smi_find_offsets(const uint8_t *d_bytes, size_t total_bytes,
                                      size_t chunk_stride_bytes, size_t chunk_len_bytes,
                                      int n_chunks, int32_t *d_offs, void *stream)
{
    if (n_chunks <= 0) return 0;
    if (!d_bytes || !d_offs || chunk_stride_bytes == 0) {
        clhip_set_error("clhip_smi_find_offsets: bad arguments");
        return -1;
    }
    hipLaunchKernelGGL(smi_find_offsets_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream,
                       d_bytes, total_bytes, chunk_stride_bytes, chunk_len_bytes, n_chunks, d_offs);
    CLHIP_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------
// unpack (+ conversion)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void unpack_word(uint32_t w, int channel, int &i, int &q)
{
    const int a = clhip_field_a(w), b = clhip_field_b(w);
    i = channel == CL_CHANNEL_HIF ? b : a;   // caribou_smi.c:342-378
    q = channel == CL_CHANNEL_HIF ? a : b;
}

template <int FMT> struct OutElem;
template <> struct OutElem<CL_FORMAT_CS16> { typedef uint32_t type; };   // int16 pair
template <> struct OutElem<CL_FORMAT_CF32> { typedef f32x2 type; };
template <> struct OutElem<CL_FORMAT_CS8> { typedef uint16_t type; };    // int8 pair
template <> struct OutElem<CL_FORMAT_CF64> { typedef double2 type; };

template <int FMT>
__device__ __forceinline__ typename OutElem<FMT>::type make_elem(int i, int q)
{
    if constexpr (FMT == CL_FORMAT_CS16) {
        return ((uint32_t)(uint16_t)(int16_t)i) | ((uint32_t)(uint16_t)(int16_t)q << 16);
    } else if constexpr (FMT == CL_FORMAT_CF32) {
        // (float)v / 4096.0f  CaribouliteStream.cpp:319-320 (exact: power of two)
        f32x2 r = {(float)i / 4096.0f, (float)q / 4096.0f};
        return r;
    } else if constexpr (FMT == CL_FORMAT_CS8) {
        // (int8_t)((v >> 5) & 0xFF)  CaribouliteStream.cpp:362-363
        return (uint16_t)(((uint32_t)(i >> 5) & 0xFFu) | (((uint32_t)(q >> 5) & 0xFFu) << 8));
    } else {
        return make_double2((double)i / 4096.0, (double)q / 4096.0);     // :341-342
    }
}

// grid = (blocks per chunk, n_chunks); each lane handles 4 consecutive samples
template <int FMT>
__global__ __launch_bounds__(256) void smi_unpack_kernel(
    int channel, const uint8_t *__restrict__ bytes, size_t total, size_t stride, size_t chunk_len,
    const int32_t *__restrict__ offs_in, typename OutElem<FMT>::type *__restrict__ out,
    uint8_t *__restrict__ meta)
{
    typedef typename OutElem<FMT>::type elem_t;
    const int c = blockIdx.y;
    const size_t start = (size_t)c * stride;
    if (start >= total) return;
    const size_t len = chunk_len < total - start ? chunk_len : total - start;
    const int offs = offs_in ? offs_in[c] : 0;
    if (offs < 0) return;                                    // sync failure: nothing written
    const size_t shortening = offs > 0 ? (size_t)(offs / 4 + 1) : 0;   // caribou_smi.c:319
    const size_t n = (len - 4 * shortening) / 4;             // :320,344
    const uint8_t *p = bytes + start + offs;
    const size_t slot0 = start / 4;
    elem_t *o = out ? out + slot0 : nullptr;
    uint8_t *m = meta ? meta + slot0 : nullptr;

    const size_t g0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const size_t gstep = (size_t)gridDim.x * blockDim.x * 4;
    const bool aligned = (((uintptr_t)p) & 15) == 0 &&
                         (!o || (((uintptr_t)o) & (4 * sizeof(elem_t) - 1)) == 0) &&
                         (!m || (((uintptr_t)m) & 3) == 0);
    for (size_t k = g0; k < n; k += gstep) {
        if (aligned && k + 4 <= n) {
            const u32x4 w = __builtin_nontemporal_load((const u32x4 *)(p + 4 * k));
            elem_t e[4];
            uint32_t mm = 0;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                int i, q;
                unpack_word(w[t], channel, i, q);
                e[t] = make_elem<FMT>(i, q);
                mm |= (w[t] & 1u) << (8 * t);                // meta.sync = s & 1  :348
            }
            if (o) {
                if constexpr (sizeof(elem_t) == 4) {
                    u32x4 v = {(uint32_t)e[0], (uint32_t)e[1], (uint32_t)e[2], (uint32_t)e[3]};
                    *(u32x4 *)(o + k) = v;
                } else if constexpr (sizeof(elem_t) == 2) {
                    u32x2 v = {(uint32_t)e[0] | ((uint32_t)e[1] << 16), (uint32_t)e[2] | ((uint32_t)e[3] << 16)};
                    *(u32x2 *)(o + k) = v;
                } else if constexpr (sizeof(elem_t) == 8) {
                    f32x4 v0 = {e[0].x, e[0].y, e[1].x, e[1].y}, v1 = {e[2].x, e[2].y, e[3].x, e[3].y};
                    *(f32x4 *)(o + k) = v0;
                    *(f32x4 *)(o + k + 2) = v1;
                } else {
#pragma unroll
                    for (int t = 0; t < 4; t++) o[k + t] = e[t];
                }
            }
            if (m) *(uint32_t *)(m + k) = mm;
        } else {
            for (size_t t = k; t < n && t < k + 4; t++) {
                const uint32_t w = ld_u32_bytes(p + 4 * t);
                int i, q;
                unpack_word(w, channel, i, q);
                if (o) o[t] = make_elem<FMT>(i, q);
                if (m) m[t] = (uint8_t)(w & 1u);
            }
        }
    }
    // one extrapolated sample after a re-synchronised chunk (:382-389); the
    // reference needs n >= 2 (it reads slots n-1 and n-2)
    if (shortening > 0 && o && n >= 2 && blockIdx.x == 0 && threadIdx.x == 0) {
        int i1, q1, i2, q2;
        unpack_word(ld_u32_bytes(p + 4 * (n - 1)), channel, i1, q1);
        unpack_word(ld_u32_bytes(p + 4 * (n - 2)), channel, i2, q2);
        const int ie = (int16_t)(110 * i1 / 100 - i2 / 10);
        const int qe = (int16_t)(110 * q1 / 100 - q2 / 10);
        o[n] = make_elem<FMT>(ie, qe);
    }
}

extern "C" int clhip_smi_unpack(int channel, const uint8_t *d_bytes, size_t total_bytes,
                                size_t chunk_stride_bytes, size_t chunk_len_bytes, int n_chunks,
                                const int32_t *d_offs, int format, void *d_out, uint8_t *d_meta,
                                void *stream)
{
    if (n_chunks <= 0 || total_bytes == 0) return 0;
    if (!d_bytes || chunk_stride_bytes == 0 || (chunk_stride_bytes & 3)) {
        clhip_set_error("clhip_smi_unpack: bad arguments (stride must be a multiple of 4)");
        return -1;
    }
    const size_t per_chunk = chunk_len_bytes / 4;
    unsigned bx = (unsigned)clhip_div_up(per_chunk ? per_chunk : 1, 256 * 4);
    if (bx > 2048) bx = 2048;
    if ((size_t)bx * n_chunks > (1u << 20)) bx = (unsigned)((1u << 20) / n_chunks) + 1;
    dim3 grid(bx, n_chunks), block(256);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH(F)                                                                              \
    hipLaunchKernelGGL(smi_unpack_kernel<F>, grid, block, 0, s, channel, d_bytes, total_bytes, \
                       chunk_stride_bytes, chunk_len_bytes, d_offs,                            \
                       (OutElem<F>::type *)d_out, d_meta)
    switch (format) {
    case CL_FORMAT_CS16: LAUNCH(CL_FORMAT_CS16); break;
    case CL_FORMAT_CF32: LAUNCH(CL_FORMAT_CF32); break;
    case CL_FORMAT_CS8: LAUNCH(CL_FORMAT_CS8); break;
    case CL_FORMAT_CF64: LAUNCH(CL_FORMAT_CF64); break;
    default: clhip_set_error("clhip_smi_unpack: unknown format %d", format); return -1;
    }
#undef LAUNCH
    CLHIP_CHECK_LAUNCH();
    return 0;
}

// One read() chunk the caller KNOWS to be in sync (caribou_smi_find_buffer_offset returns 0 exactly when the chunk's words at
// byte offsets 0, 4, 8, 12 carry the sync pattern, and the host has looked at them in its pinned staging memory): no search,
// every slot is written, and the samples leave twice in one pass -- as int16 pairs into the persistent native buffer
// (what the Stream's interm buffer holds for the calls that follow) and, in the client's format, into `out`, which may be
// mapped pinned HOST memory: the stores then cross PCIe themselves and the call needs no device-to-host copy.
template <int FMT>
__global__ __launch_bounds__(256) void smi_unpack_aligned_kernel(int channel, const u32x4 *__restrict__ words, size_t n_groups,
                                                                typename OutElem<FMT>::type *__restrict__ out, u32x4 *__restrict__ cs16)
{
    typedef typename OutElem<FMT>::type elem_t;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < n_groups; g += (size_t)gridDim.x * blockDim.x) {
        const u32x4 w = __builtin_nontemporal_load(words + g);
        elem_t e[4];
        u32x4 c;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            int i, q;
            unpack_word(w[t], channel, i, q);
            e[t] = make_elem<FMT>(i, q);
            c[t] = ((uint32_t)i & 0xFFFFu) | ((uint32_t)q << 16);
        }
        if (cs16) cs16[g] = c;
        if constexpr (FMT == CL_FORMAT_CS16) {
            *((u32x4 *)out + g) = c;
        } else if constexpr (sizeof(elem_t) == 2) {
            u32x2 v = {(uint32_t)e[0] | ((uint32_t)e[1] << 16), (uint32_t)e[2] | ((uint32_t)e[3] << 16)};
            *((u32x2 *)out + g) = v;
        } else if constexpr (sizeof(elem_t) == 8) {
            f32x4 v0 = {e[0].x, e[0].y, e[1].x, e[1].y}, v1 = {e[2].x, e[2].y, e[3].x, e[3].y};
            *((f32x4 *)out + 2 * g) = v0;
            *((f32x4 *)out + 2 * g + 1) = v1;
        } else {
#pragma unroll
            for (int t = 0; t < 4; t++) out[4 * g + t] = e[t];
        }
    }
}

extern "C" int clhip_smi_unpack_aligned(int channel, const uint8_t *d_bytes, size_t n_bytes, int format, void *out, int16_t *d_cs16,
                                        void *stream)
{
    if (n_bytes == 0) return 0;
    if (!d_bytes || !out || (n_bytes & 15) || (((uintptr_t)d_bytes | (uintptr_t)out | (uintptr_t)d_cs16) & 15)) {
        clhip_set_error("clhip_smi_unpack_aligned: 16-byte aligned buffers and a multiple of 4 samples");
        return -1;
    }
    const size_t n_groups = n_bytes / 16;
    unsigned grid = (unsigned)clhip_div_up(n_groups, 256);
    if (grid > 2048) grid = 2048;
    hipStream_t s = (hipStream_t)stream;
    const u32x4 *w = (const u32x4 *)d_bytes;
    u32x4 *c = (u32x4 *)d_cs16;
    switch (format) {
    case CL_FORMAT_CS16: hipLaunchKernelGGL(smi_unpack_aligned_kernel<CL_FORMAT_CS16>, dim3(grid), dim3(256), 0, s, channel, w, n_groups, (OutElem<CL_FORMAT_CS16>::type *)out, c); break;
    case CL_FORMAT_CF32: hipLaunchKernelGGL(smi_unpack_aligned_kernel<CL_FORMAT_CF32>, dim3(grid), dim3(256), 0, s, channel, w, n_groups, (OutElem<CL_FORMAT_CF32>::type *)out, c); break;
    case CL_FORMAT_CS8: hipLaunchKernelGGL(smi_unpack_aligned_kernel<CL_FORMAT_CS8>, dim3(grid), dim3(256), 0, s, channel, w, n_groups, (OutElem<CL_FORMAT_CS8>::type *)out, c); break;
    case CL_FORMAT_CF64: hipLaunchKernelGGL(smi_unpack_aligned_kernel<CL_FORMAT_CF64>, dim3(grid), dim3(256), 0, s, channel, w, n_groups, (OutElem<CL_FORMAT_CF64>::type *)out, c); break;
    default: clhip_set_error("clhip_smi_unpack_aligned: unknown format %d", format); return -1;
    }
    CLHIP_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------
// CS16 <-> other formats
// ---------------------------------------------------------------------------
template <int FMT>
__global__ __launch_bounds__(256) void from_cs16_kernel(const uint32_t *__restrict__ in, size_t n,
                                                        typename OutElem<FMT>::type *__restrict__ out)
{
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; k < n; k += step) {
        const uint32_t w = in[k];
        out[k] = make_elem<FMT>((int)(int16_t)(w & 0xFFFF), (int)(int16_t)(w >> 16));
    }
}

extern "C" int clhip_convert_from_cs16(const int16_t *d_iq, size_t n, int format, void *d_out, void *stream)
{
    if (n == 0) return 0;
    unsigned grid = (unsigned)clhip_div_up(n, 256);
    if (grid > 8192) grid = 8192;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t *in = (const uint32_t *)d_iq;
    switch (format) {
    case CL_FORMAT_CS16: CLHIP_CHECK(hipMemcpyAsync(d_out, d_iq, 4 * n, hipMemcpyDeviceToDevice, s)); return 0;
    case CL_FORMAT_CF32: hipLaunchKernelGGL(from_cs16_kernel<CL_FORMAT_CF32>, dim3(grid), dim3(256), 0, s, in, n, (f32x2 *)d_out); break;
    case CL_FORMAT_CS8: hipLaunchKernelGGL(from_cs16_kernel<CL_FORMAT_CS8>, dim3(grid), dim3(256), 0, s, in, n, (uint16_t *)d_out); break;
    case CL_FORMAT_CF64: hipLaunchKernelGGL(from_cs16_kernel<CL_FORMAT_CF64>, dim3(grid), dim3(256), 0, s, in, n, (double2 *)d_out); break;
    default: clhip_set_error("clhip_convert_from_cs16: unknown format %d", format); return -1;
    }
    CLHIP_CHECK_LAUNCH();
    return 0;
}

// (int16_t)(f * 4096.0f) as the reference binary computes it on its x86-64
// build host: truncate toward zero to int32 (out of range / NaN -> 0x80000000),
// keep the low 16 bits (CaribouliteStream.cpp:207-208,223-224).
__device__ __forceinline__ uint32_t f2i16(float v)
{
    int t = (v >= -2147483648.0f && v < 2147483648.0f) ? (int)v : (int)0x80000000;
    return (uint32_t)t & 0xFFFFu;
}
__device__ __forceinline__ uint32_t d2i16(double v)
{
    int t = (v > -2147483649.0 && v < 2147483648.0) ? (int)v : (int)0x80000000;
    return (uint32_t)t & 0xFFFFu;
}

template <int FMT>
__global__ __launch_bounds__(256) void to_cs16_kernel(const typename OutElem<FMT>::type *__restrict__ in,
                                                      size_t n, uint32_t *__restrict__ out)
{
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; k < n; k += step) {
        if constexpr (FMT == CL_FORMAT_CF32) {
            const f32x2 v = in[k];
            out[k] = f2i16(v.x * 4096.0f) | (f2i16(v.y * 4096.0f) << 16);
        } else if constexpr (FMT == CL_FORMAT_CF64) {
            const double2 v = in[k];
            out[k] = d2i16(v.x * 4096.0) | (d2i16(v.y * 4096.0) << 16);
        } else {   // CS8: ((int16_t)v) << 5   CaribouliteStream.cpp:238-239
            const uint16_t v = in[k];
            const int i = (int8_t)(v & 0xFF), q = (int8_t)(v >> 8);
            out[k] = ((uint32_t)(i << 5) & 0xFFFFu) | (((uint32_t)(q << 5) & 0xFFFFu) << 16);
        }
    }
}

extern "C" int clhip_convert_to_cs16(const void *d_in, int format, size_t n, int16_t *d_iq, void *stream)
{
    if (n == 0) return 0;
    unsigned grid = (unsigned)clhip_div_up(n, 256);
    if (grid > 8192) grid = 8192;
    hipStream_t s = (hipStream_t)stream;
    uint32_t *out = (uint32_t *)d_iq;
    switch (format) {
    case CL_FORMAT_CS16: CLHIP_CHECK(hipMemcpyAsync(d_iq, d_in, 4 * n, hipMemcpyDeviceToDevice, s)); return 0;
    case CL_FORMAT_CF32: hipLaunchKernelGGL(to_cs16_kernel<CL_FORMAT_CF32>, dim3(grid), dim3(256), 0, s, (const f32x2 *)d_in, n, out); break;
    case CL_FORMAT_CS8: hipLaunchKernelGGL(to_cs16_kernel<CL_FORMAT_CS8>, dim3(grid), dim3(256), 0, s, (const uint16_t *)d_in, n, out); break;
    case CL_FORMAT_CF64: hipLaunchKernelGGL(to_cs16_kernel<CL_FORMAT_CF64>, dim3(grid), dim3(256), 0, s, (const double2 *)d_in, n, out); break;
    default: clhip_set_error("clhip_convert_to_cs16: unknown format %d", format); return -1;
    }
    CLHIP_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------
// TX pack (caribou_smi.c:684-717)
//   byte0 [SOF TXC CTX I12..I8] byte1 [0 I7..I1] byte2 [0 I0 Q12..Q7] byte3 [0 Q6..Q0]
//   built MSB-first in a u32, byte-swapped so byte0 is first in memory.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pack_tx_word(int mode, uint32_t iq)
{
    uint32_t ii = iq & 0xFFFFu, qq = iq >> 16;
    if (mode == CL_TX_AS_WRITTEN) { ii = 0xFFFFu; qq = 0; }   // :700-701 as shipped
    ii &= 0x1FFFu; qq &= 0x1FFFu;
    const uint32_t s = (0x7u << 29) | ((ii >> 8) << 24) | (((ii >> 1) & 0x7Fu) << 16) |
                       ((ii & 1u) << 14) | ((qq >> 7) << 8) | (qq & 0x7Fu);
    return __builtin_bswap32(s);
}

__global__ __launch_bounds__(256) void smi_pack_kernel(int mode, const uint32_t *__restrict__ iq, size_t n,
                                                       uint32_t *__restrict__ out)
{
    const size_t n4 = n / 4;
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    const bool aligned = ((((uintptr_t)iq) | ((uintptr_t)out)) & 15) == 0;
    if (aligned) {
        for (size_t j = k; j < n4; j += step) {
            const u32x4 v = ((const u32x4 *)iq)[j];
            u32x4 r;
#pragma unroll
            for (int t = 0; t < 4; t++) r[t] = pack_tx_word(mode, v[t]);
            ((u32x4 *)out)[j] = r;
        }
        for (size_t j = 4 * n4 + k; j < n; j += step) out[j] = pack_tx_word(mode, iq[j]);
    } else {
        for (size_t j = k; j < n; j += step) out[j] = pack_tx_word(mode, iq[j]);
    }
}

extern "C" int clhip_smi_pack(int mode, const int16_t *d_iq, size_t n, uint8_t *d_bytes, void *stream)
{
    if (n == 0) return 0;
    if ((((uintptr_t)d_iq) | ((uintptr_t)d_bytes)) & 3) {
        clhip_set_error("clhip_smi_pack: buffers must be 4-byte aligned");
        return -1;
    }
    unsigned grid = (unsigned)clhip_div_up(clhip_div_up(n, 4), 256);
    if (grid > 8192) grid = 8192;
    if (grid == 0) grid = 1;
    hipLaunchKernelGGL(smi_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, mode,
                       (const uint32_t *)d_iq, n, (uint32_t *)d_bytes);
    CLHIP_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------
// writeStream of a non-native format in ONE launch: the conversion loop (CaribouliteStream.cpp:199-244) and the TX pack
// (caribou_smi.c:684-717) on the same sample -- the int16 pair the reference parks in its intermediate buffer lives in
// a register.  Same device functions as the two separate kernels: bit-identical bytes.
// ---------------------------------------------------------------------------
template <int FMT> __device__ __forceinline__ uint32_t sample_to_cs16(const typename OutElem<FMT>::type v)
{
    if constexpr (FMT == CL_FORMAT_CF32) return f2i16(v.x * 4096.0f) | (f2i16(v.y * 4096.0f) << 16);
    else if constexpr (FMT == CL_FORMAT_CF64) return d2i16(v.x * 4096.0) | (d2i16(v.y * 4096.0) << 16);
    else if constexpr (FMT == CL_FORMAT_CS8) {
        const int i = (int8_t)(v & 0xFF), q = (int8_t)(v >> 8);
        return ((uint32_t)(i << 5) & 0xFFFFu) | (((uint32_t)(q << 5) & 0xFFFFu) << 16);
    } else return v;
}

template <int FMT>
__global__ __launch_bounds__(256) void convert_pack_kernel(int mode, const typename OutElem<FMT>::type *__restrict__ in, size_t n,
                                                           uint32_t *__restrict__ out)
{
    const size_t step = (size_t)gridDim.x * blockDim.x;
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if ((((uintptr_t)out) & 15) == 0) {                        // four words per lane: 16-byte stores
        const size_t n4 = n / 4;
        for (size_t j = k; j < n4; j += step) {
            u32x4 r;
#pragma unroll
            for (int t = 0; t < 4; t++) r[t] = pack_tx_word(mode, sample_to_cs16<FMT>(in[4 * j + t]));
            ((u32x4 *)out)[j] = r;
        }
        for (size_t j = 4 * n4 + k; j < n; j += step) out[j] = pack_tx_word(mode, sample_to_cs16<FMT>(in[j]));
    } else {
        for (size_t j = k; j < n; j += step) out[j] = pack_tx_word(mode, sample_to_cs16<FMT>(in[j]));
    }
}

extern "C" int clhip_convert_pack(const void *d_in, int format, size_t n, int mode, uint8_t *d_bytes, void *stream)
{
    if (n == 0) return 0;
    if (!d_in || !d_bytes || (((uintptr_t)d_bytes) & 3)) { clhip_set_error("clhip_convert_pack: bad arguments"); return -1; }
    unsigned grid = (unsigned)clhip_div_up(clhip_div_up(n, 4), 256);
    if (grid > 8192) grid = 8192;
    hipStream_t s = (hipStream_t)stream;
    uint32_t *out = (uint32_t *)d_bytes;
    switch (format) {
    case CL_FORMAT_CS16: hipLaunchKernelGGL(convert_pack_kernel<CL_FORMAT_CS16>, dim3(grid), dim3(256), 0, s, mode, (const uint32_t *)d_in, n, out); break;
    case CL_FORMAT_CF32: hipLaunchKernelGGL(convert_pack_kernel<CL_FORMAT_CF32>, dim3(grid), dim3(256), 0, s, mode, (const f32x2 *)d_in, n, out); break;
    case CL_FORMAT_CS8: hipLaunchKernelGGL(convert_pack_kernel<CL_FORMAT_CS8>, dim3(grid), dim3(256), 0, s, mode, (const uint16_t *)d_in, n, out); break;
    case CL_FORMAT_CF64: hipLaunchKernelGGL(convert_pack_kernel<CL_FORMAT_CF64>, dim3(grid), dim3(256), 0, s, mode, (const double2 *)d_in, n, out); break;
    default: clhip_set_error("clhip_convert_pack: unknown format %d", format); return -1;
    }
    CLHIP_CHECK_LAUNCH();
    return 0;
}

// The same for up to CLHIP_PACK_ROWS streams in one launch (a stream group's writeStream): row r converts in[r] into out[r], the rows
// being wherever they are -- the clients' samples in one pinned or device buffer, the packed words in the members' TX FIFOs.
struct PackRows { const void *in[CLHIP_PACK_ROWS]; uint32_t *out[CLHIP_PACK_ROWS]; };

template <int FMT>
__global__ __launch_bounds__(256) void convert_pack_rows_kernel(int mode, PackRows rows, size_t n)
{
    const typename OutElem<FMT>::type *__restrict__ in = (const typename OutElem<FMT>::type *)rows.in[blockIdx.y];
    uint32_t *__restrict__ out = rows.out[blockIdx.y];
    const size_t step = (size_t)gridDim.x * blockDim.x;
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if ((((uintptr_t)out) & 15) == 0) {
        const size_t n4 = n / 4;
        for (size_t j = k; j < n4; j += step) {
            u32x4 r;
#pragma unroll
            for (int t = 0; t < 4; t++) r[t] = pack_tx_word(mode, sample_to_cs16<FMT>(in[4 * j + t]));
            ((u32x4 *)out)[j] = r;
        }
        for (size_t j = 4 * n4 + k; j < n; j += step) out[j] = pack_tx_word(mode, sample_to_cs16<FMT>(in[j]));
    } else {
        for (size_t j = k; j < n; j += step) out[j] = pack_tx_word(mode, sample_to_cs16<FMT>(in[j]));
    }
}

extern "C" int clhip_convert_pack_rows(const void *const *d_in_rows, int format, size_t n, int n_rows, int mode, uint8_t *const *d_bytes_rows, void *stream)
{
    if (n == 0 || n_rows == 0) return 0;
    if (!d_in_rows || !d_bytes_rows || n_rows < 0 || n_rows > CLHIP_PACK_ROWS) { clhip_set_error("clhip_convert_pack_rows: 1 .. %d rows", CLHIP_PACK_ROWS); return -1; }
    PackRows rows;
    for (int r = 0; r < CLHIP_PACK_ROWS; r++) {
        rows.in[r] = r < n_rows ? d_in_rows[r] : nullptr; rows.out[r] = r < n_rows ? (uint32_t *)d_bytes_rows[r] : nullptr;
        if (r < n_rows && (!rows.in[r] || !rows.out[r] || (((uintptr_t)rows.out[r]) & 3))) { clhip_set_error("clhip_convert_pack_rows: bad row %d", r); return -1; }
    }
    unsigned gx = (unsigned)clhip_div_up(clhip_div_up(n, 4), 256);
    if (gx > 2048) gx = 2048;
    const dim3 grid(gx, (unsigned)n_rows);
    hipStream_t s = (hipStream_t)stream;
    switch (format) {
    case CL_FORMAT_CS16: hipLaunchKernelGGL(convert_pack_rows_kernel<CL_FORMAT_CS16>, grid, dim3(256), 0, s, mode, rows, n); break;
    case CL_FORMAT_CF32: hipLaunchKernelGGL(convert_pack_rows_kernel<CL_FORMAT_CF32>, grid, dim3(256), 0, s, mode, rows, n); break;
    case CL_FORMAT_CS8: hipLaunchKernelGGL(convert_pack_rows_kernel<CL_FORMAT_CS8>, grid, dim3(256), 0, s, mode, rows, n); break;
    case CL_FORMAT_CF64: hipLaunchKernelGGL(convert_pack_rows_kernel<CL_FORMAT_CF64>, grid, dim3(256), 0, s, mode, rows, n); break;
    default: clhip_set_error("clhip_convert_pack_rows: unknown format %d", format); return -1;
    }
    CLHIP_CHECK_LAUNCH();
    return 0;
}

// Rows of packed words (one stride apart on the device) to n_rows destinations of their own -- the rooms reserved in the members' pinned
// TX FIFOs, stored across PCIe by the launch itself: one launch per sub-batch instead of a copy-engine call per member
struct WordRows { uint32_t *out[CLHIP_PACK_ROWS]; };

__global__ __launch_bounds__(256) void words_to_rows_kernel(const uint32_t *__restrict__ in, size_t in_stride_words, size_t n, WordRows rows)
{
    const uint32_t *__restrict__ src = in + (size_t)blockIdx.y * in_stride_words;
    uint32_t *__restrict__ dst = rows.out[blockIdx.y];
    const size_t step = (size_t)gridDim.x * blockDim.x;
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (((((uintptr_t)dst) | ((uintptr_t)src)) & 15) == 0) {
        const size_t n4 = n / 4;
        for (size_t j = k; j < n4; j += step) ((u32x4 *)dst)[j] = ((const u32x4 *)src)[j];
        for (size_t j = 4 * n4 + k; j < n; j += step) dst[j] = src[j];
    } else {
        for (size_t j = k; j < n; j += step) dst[j] = src[j];
    }
}

extern "C" int clhip_words_to_rows(const uint8_t *d_words, size_t in_stride_bytes, size_t n_words, int n_rows, uint8_t *const *d_dst_rows, void *stream)
{
    if (n_words == 0 || n_rows == 0) return 0;
    if (!d_words || !d_dst_rows || n_rows < 0 || n_rows > CLHIP_PACK_ROWS || (in_stride_bytes & 3) || (((uintptr_t)d_words) & 3)) {
        clhip_set_error("clhip_words_to_rows: bad arguments (1 .. %d rows)", CLHIP_PACK_ROWS);
        return -1;
    }
    WordRows rows;
    for (int r = 0; r < CLHIP_PACK_ROWS; r++) {
        rows.out[r] = r < n_rows ? (uint32_t *)d_dst_rows[r] : nullptr;
        if (r < n_rows && (!rows.out[r] || (((uintptr_t)rows.out[r]) & 3))) { clhip_set_error("clhip_words_to_rows: bad row %d", r); return -1; }
    }
    unsigned gx = (unsigned)clhip_div_up(clhip_div_up(n_words, 4), 256);
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(words_to_rows_kernel, dim3(gx, (unsigned)n_rows), dim3(256), 0, (hipStream_t)stream, (const uint32_t *)d_words, in_stride_bytes / 4,
                       n_words, rows);
    CLHIP_CHECK_LAUNCH();
    return 0;
}

// Rows of results on the device to destinations of their own, lengths of their own -- the clients' REGISTERED buffers of a stream group's
// sub-batch (cl_group_register_buffers), stored across PCIe by one launch instead of a copy-engine call per member
struct ByteRows { const uint8_t *src[CLHIP_PACK_ROWS]; uint8_t *dst[CLHIP_PACK_ROWS]; size_t bytes[CLHIP_PACK_ROWS]; };

__global__ __launch_bounds__(256) void rows_to_rows_kernel(ByteRows rows)
{
    const uint8_t *__restrict__ src = rows.src[blockIdx.y];
    uint8_t *__restrict__ dst = rows.dst[blockIdx.y];
    const size_t n = rows.bytes[blockIdx.y];
    const size_t step = (size_t)gridDim.x * blockDim.x;
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (((((uintptr_t)dst) | ((uintptr_t)src)) & 15) == 0) {
        const size_t n16 = n / 16;
        for (size_t j = k; j < n16; j += step) ((u32x4 *)dst)[j] = ((const u32x4 *)src)[j];
        for (size_t j = 16 * n16 + k; j < n; j += step) dst[j] = src[j];
    } else if (((((uintptr_t)dst) | ((uintptr_t)src)) & 3) == 0) {
        const size_t n4 = n / 4;
        for (size_t j = k; j < n4; j += step) ((uint32_t *)dst)[j] = ((const uint32_t *)src)[j];
        for (size_t j = 4 * n4 + k; j < n; j += step) dst[j] = src[j];
    } else {
        for (size_t j = k; j < n; j += step) dst[j] = src[j];
    }
}

extern "C" int clhip_rows_to_rows(const void *const *d_src_rows, void *const *d_dst_rows, const size_t *row_bytes, int n_rows, void *stream)
{
    if (n_rows == 0) return 0;
    if (!d_src_rows || !d_dst_rows || !row_bytes || n_rows < 0 || n_rows > CLHIP_PACK_ROWS) { clhip_set_error("clhip_rows_to_rows: bad arguments (1 .. %d rows)", CLHIP_PACK_ROWS); return -1; }
    ByteRows rows;
    size_t most = 0;
    for (int r = 0; r < CLHIP_PACK_ROWS; r++) {
        rows.src[r] = r < n_rows ? (const uint8_t *)d_src_rows[r] : nullptr; rows.dst[r] = r < n_rows ? (uint8_t *)d_dst_rows[r] : nullptr;
        rows.bytes[r] = r < n_rows ? row_bytes[r] : 0;
        if (r < n_rows && rows.bytes[r] && (!rows.src[r] || !rows.dst[r])) { clhip_set_error("clhip_rows_to_rows: bad row %d", r); return -1; }
        if (rows.bytes[r] > most) most = rows.bytes[r];
    }
    if (!most) return 0;
    unsigned gx = (unsigned)clhip_div_up(clhip_div_up(most, 16), 256);
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(rows_to_rows_kernel, dim3(gx, (unsigned)n_rows), dim3(256), 0, (hipStream_t)stream, rows);
    CLHIP_CHECK_LAUNCH();
    return 0;
}

// the I rail of interleaved CF32 as a dense fp32 message (the FM modulator's input: SURVEY.md a13 "if given I/Q, use I")
__global__ __launch_bounds__(256) void take_i_rail_kernel(const f32x2 *__restrict__ in, size_t n, float *__restrict__ out)
{
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += step) out[k] = in[k].x;
}

extern "C" int clhip_take_i_rail(const float *d_cf32, size_t n, float *d_msg, void *stream)
{
    if (n == 0) return 0;
    if (!d_cf32 || !d_msg) { clhip_set_error("clhip_take_i_rail: bad arguments"); return -1; }
    unsigned grid = (unsigned)clhip_div_up(n, 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(take_i_rail_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const f32x2 *)d_cf32, n, d_msg);
    CLHIP_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------
// link-integrity (debug) modes: caribou_smi.c:172-215 (analyse), :266-283 (search)
//   LFSR  : every byte must be lfsr(previous byte) and non-zero (smi_utils.c:220-224)
//   push / pull : every word must be 0xABCDEF01; the search accepts < 4 flipped bits
// res[0] = offs (-1: none), res[1] = erroneous bytes, res[2] = first error (byte index, -1: none),
// res[3] = last byte seen (carried into the next call's LFSR check)
// ---------------------------------------------------------------------------
#define SMI_DEBUG_WORD 0xABCDEF01u

__global__ __launch_bounds__(256) void smi_debug_find_kernel(int mode, const uint8_t *__restrict__ p, size_t len,
                                                             uint32_t last_in, int32_t *__restrict__ res)
{
    __shared__ int s_found;
    const int tid = threadIdx.x;
    if (tid == 0) { res[1] = 0; res[2] = 0x7fffffff; res[3] = (int32_t)last_in; s_found = 0x7fffffff; }
    __syncthreads();
    if (len <= 16 || mode == 1 /* lfsr: no search */) {
        if (tid == 0) res[0] = 0;
        return;
    }
    const size_t limit = len - 4;                    // :268  offs < len - BYTES_PER_SAMPLE
    for (size_t base = 0; base < limit; base += 256) {
        const size_t o = base + tid;
        if (o < limit && __popc(ld_u32_bytes(p + o) ^ SMI_DEBUG_WORD) < 4) atomicMin(&s_found, (int)o);
        __syncthreads();
        const int f = s_found;
        __syncthreads();
        if (f != 0x7fffffff) break;
    }
    if (tid == 0) res[0] = s_found == 0x7fffffff ? -1 : s_found;
}

__global__ __launch_bounds__(256) void smi_debug_count_kernel(int mode, const uint8_t *__restrict__ data, size_t len,
                                                              uint32_t last_in, int32_t *__restrict__ res)
{
    const int offs = res[0];
    if (offs < 0) return;
    const size_t shortening = offs > 0 ? (size_t)(offs / 4 + 1) : 0;     // caribou_smi.c:319
    const size_t alen = len - 4 * shortening;
    const uint8_t *p = data + offs;
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    unsigned cnt = 0;
    int first = 0x7fffffff;
    if (mode == 1) {
        for (size_t i = k; i < alen; i += step) {
            const uint32_t prev = i ? p[i - 1] : last_in;
            const uint32_t want = ((prev >> 1) | ((((prev >> 2) ^ (prev >> 3)) & 1u) << 7)) & 0xFFu;
            if (p[i] != want || p[i] == 0) { cnt++; if ((int)i < first) first = (int)i; }
        }
        if (k == 0 && alen) res[3] = p[alen - 1];
    } else {
        for (size_t i = k; i < alen / 4; i += step)
            if (ld_u32_bytes(p + 4 * i) != SMI_DEBUG_WORD) { cnt += 4; if ((int)(4 * i) < first) first = (int)(4 * i); }
    }
    // wave-level reduction, one atomic per wave
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        cnt += __shfl_down(cnt, o, 64);
        const int f2 = __shfl_down(first, o, 64);
        first = f2 < first ? f2 : first;
    }
    if ((threadIdx.x & 63) == 0) {
        if (cnt) atomicAdd((unsigned *)&res[1], cnt);
        if (first != 0x7fffffff) atomicMin(&res[2], first);
    }
}

extern "C" int clhip_smi_debug_analyze(int mode, const uint8_t *d_bytes, size_t len, uint32_t last_correct_byte,
                                       int32_t *d_res, void *stream)
{
    if (mode < 1 || mode > 3 || !d_bytes || !d_res) { clhip_set_error("clhip_smi_debug_analyze: bad arguments"); return -1; }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(smi_debug_find_kernel, dim3(1), dim3(256), 0, s, mode, d_bytes, len, last_correct_byte, d_res);
    unsigned grid = (unsigned)clhip_div_up(len ? len : 1, 256 * 16);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(smi_debug_count_kernel, dim3(grid), dim3(256), 0, s, mode, d_bytes, len, last_correct_byte, d_res);
    CLHIP_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------
// pps tags: the positions i with meta[i].sync == 1, ascending -- what the GNU Radio source's work() finds with a
// host loop over every sample of the meta plane (gr-caribouLite/lib/caribouLiteSource_impl.cc:113-119) as one
// ordered compaction on the device, so that only the (few) positions cross PCIe and the host touches no sample.
// HBM-bound byte work: 16 bytes per lane and round, `== 1` on four bytes at a time with an exact zero-byte test
// (18 VALU operations per 16 bytes: at 1 byte per sample the test, not the memory, would bound a naive loop).
// The plane may start at any byte address: lanes work on the 16-byte grid of the ADDRESS (virtual index
// v = i + mis, mis = address & 15); the at most two groups that straddle an end of the plane are read byte by byte
// by the one lane that owns them.  Two passes over a workgroup's 16 rounds: a FAST one that only asks "any marker in
// my group?" with all 16 loads in flight, and -- for the rounds that hold one, a handful per second of stream -- a
// placing one that re-reads the group (cache hit), scans the workgroup and stores the positions.
// Small planes (one MTU): ONE workgroup of 1024 lanes.  Large ones: a counting launch (one 64 KiB tile per
// workgroup) and an emitting launch in which a workgroup whose tile holds a marker adds up the counts before it (a
// few KB from L2) and places its tile; tiles without markers are not read twice.
// ---------------------------------------------------------------------------
#define TAG_TILE 65536u                         // bytes of the plane per workgroup tile: 256 lanes x 16 B x 16 rounds
#define TAG_SMALL (4u * TAG_TILE)               // up to here one workgroup of 1024 lanes does it all in one launch

// per dword: all-ones unless one of its bytes equals 1 (then the 0x80 of that byte is clear) -- exact, no borrow between bytes
__device__ __forceinline__ uint32_t tag_t(uint32_t x)
{
    const uint32_t y = x ^ 0x01010101u;
    return ((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu;
}
__device__ __forceinline__ uint32_t tag_ones4(uint32_t x)      // bit k <- byte k of x == 1
{
    const uint32_t z = ~tag_t(x);
    return ((z >> 7) & 1u) | ((z >> 14) & 2u) | ((z >> 21) & 4u) | ((z >> 28) & 8u);
}
__device__ __forceinline__ bool tag_whole(size_t v0, size_t v_lo, size_t v_hi) { return v0 >= v_lo && v0 + 16 <= v_hi; }
// the group at v0 if it lies wholly inside the plane, else words without a marker (nothing uses the result at once: a
// lane's loads of all its rounds are in flight together)
__device__ __forceinline__ u32x4 tag_load16(const uint8_t *__restrict__ m_al, size_t v0, size_t v_lo, size_t v_hi)
{
    u32x4 w = {0u, 0u, 0u, 0u};
    if (tag_whole(v0, v_lo, v_hi)) w = __builtin_nontemporal_load((const u32x4 *)(m_al + v0));
    return w;
}
__device__ __forceinline__ uint32_t tag_count16(u32x4 w)       // how many of the 16 bytes equal 1
{
    return 128u - (__popc(tag_t(w.x)) + __popc(tag_t(w.y)) + __popc(tag_t(w.z)) + __popc(tag_t(w.w)));
}
// bit b <- the byte at virtual index v0 + b lies inside the plane and equals 1 (any group: the placing pass, the edge groups)
__device__ __forceinline__ uint32_t tag_mask16(const uint8_t *__restrict__ m_al, size_t v0, size_t v_lo, size_t v_hi)
{
    if (v0 >= v_hi || v0 + 16 <= v_lo) return 0;
    if (tag_whole(v0, v_lo, v_hi)) {
        const u32x4 w = *(const u32x4 *)(m_al + v0);
        return tag_ones4(w.x) | (tag_ones4(w.y) << 4) | (tag_ones4(w.z) << 8) | (tag_ones4(w.w) << 12);
    }
    uint32_t mask = 0;
    for (int b = 0; b < 16; b++)
        if (v0 + b >= v_lo && v0 + b < v_hi && m_al[v0 + b] == 1) mask |= 1u << b;
    return mask;
}

// The fast pass over a workgroup's 16 rounds of NT lanes x 16 bytes from virtual index v_wg: bit r of the result <- this
// lane's group of round r holds a marker; *n_mine <- how many markers this lane's groups hold.
template <int NT>
__device__ __forceinline__ uint32_t tag_fast_pass(const uint8_t *__restrict__ m_al, size_t v_wg, size_t v_lo, size_t v_hi, uint32_t *n_mine)
{
    const size_t v_lane = v_wg + (size_t)threadIdx.x * 16;
    u32x4 w[16];
#pragma unroll
    for (int r = 0; r < 16; r++) w[r] = tag_load16(m_al, v_lane + (size_t)r * (NT * 16), v_lo, v_hi);
    uint32_t mine = 0, n = 0;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const uint32_t c = tag_count16(w[r]);
        n += c;
        mine |= (c ? 1u : 0u) << r;
    }
    // the groups that straddle the plane's ends (none when the plane starts and ends on the 16-byte grid): the fast pass
    // saw them as empty; the lane whose round grid they lie on reads them byte by byte
    const size_t g_lo = v_lo & ~(size_t)15, g_hi = v_hi & ~(size_t)15;
    const bool lo_partial = (v_lo & 15) != 0;
    const bool hi_partial = (v_hi & 15) != 0 && !(lo_partial && g_hi == g_lo);
#pragma unroll
    for (int e = 0; e < 2; e++) {
        const size_t g = e == 0 ? g_lo : g_hi;
        if ((e == 0 ? lo_partial : hi_partial) && g >= v_lane && (g - v_lane) % (NT * 16) == 0 && (g - v_lane) / (NT * 16) < 16) {
            const uint32_t c = __popc(tag_mask16(m_al, g, v_lo, v_hi));
            n += c;
            mine |= (c ? 1u : 0u) << (int)((g - v_lane) / (NT * 16));
        }
    }
    *n_mine = n;
    return mine;
}

// One round of the placing pass (this lane: the group at v0): positions are appended at idx[base ...] in address order;
// returns the round's number of markers (uniform).  s_w: NT / 64 words of LDS.
template <int NT>
__device__ __forceinline__ uint32_t tag_place(const uint8_t *__restrict__ m_al, size_t v0, size_t v_lo, size_t v_hi, uint32_t base,
                                              uint32_t *__restrict__ idx, size_t cap, uint32_t *s_w)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t mask = tag_mask16(m_al, v0, v_lo, v_hi);
    const uint32_t c = __popc(mask);
    uint32_t inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
    }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; w++) {
        const uint32_t t = s_w[w];
        if (w < wave) before += t;
        total += t;
    }
    __syncthreads();                                           // s_w is free for the next round
    size_t at = (size_t)base + before + (inc - c);
    while (mask) {
        const int b = __builtin_ctz(mask);
        mask &= mask - 1;
        if (at < cap) idx[at] = (uint32_t)(v0 + b - v_lo);
        at++;
    }
    return total;
}

// the placing pass over the rounds whose bit is set in the workgroup's bitmap (s_bits: the lanes' bitmaps OR-ed together)
template <int NT>
__device__ __forceinline__ uint32_t tag_place_rounds(const uint8_t *__restrict__ m_al, size_t v_wg, size_t v_lo, size_t v_hi, uint32_t mine,
                                                     uint32_t base, uint32_t *__restrict__ idx, size_t cap, uint32_t *s_w, uint32_t *s_bits)
{
    if (mine) atomicOr(s_bits, mine);
    __syncthreads();
    const uint32_t rounds = *s_bits;
    for (int r = 0; r < 16; r++)
        if (rounds >> r & 1u)
            base += tag_place<NT>(m_al, v_wg + (size_t)r * (NT * 16) + (size_t)threadIdx.x * 16, v_lo, v_hi, base, idx, cap, s_w);
    return base;
}

__global__ __launch_bounds__(1024) void sync_tags_small_kernel(const uint8_t *__restrict__ m_al, size_t v_lo, size_t v_hi,
                                                               uint32_t *__restrict__ idx, size_t cap, uint32_t *__restrict__ count)
{
    __shared__ uint32_t s_w[16], s_bits;
    if (threadIdx.x == 0) s_bits = 0;
    __syncthreads();
    uint32_t n_mine;
    const uint32_t mine = tag_fast_pass<1024>(m_al, 0, v_lo, v_hi, &n_mine);
    const uint32_t total = tag_place_rounds<1024>(m_al, 0, v_lo, v_hi, mine, 0, idx, cap, s_w, &s_bits);
    if (threadIdx.x == 0) *count = total;
}

__global__ __launch_bounds__(256) void sync_tags_count_kernel(const uint8_t *__restrict__ m_al, size_t v_lo, size_t v_hi,
                                                              uint32_t *__restrict__ tile_counts)
{
    __shared__ uint32_t s_w[4];
    const int tid = threadIdx.x;
    uint32_t c;
    (void)tag_fast_pass<256>(m_al, (size_t)blockIdx.x * TAG_TILE, v_lo, v_hi, &c);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((tid & 63) == 0) s_w[tid >> 6] = c;
    __syncthreads();
    if (tid == 0) tile_counts[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

__global__ __launch_bounds__(256) void sync_tags_emit_kernel(const uint8_t *__restrict__ m_al, size_t v_lo, size_t v_hi,
                                                             const uint32_t *__restrict__ tile_counts, uint32_t *__restrict__ idx,
                                                             size_t cap, uint32_t *__restrict__ count)
{
    __shared__ uint32_t s_w[4], s_bits;
    const int tid = threadIdx.x;
    const unsigned t = blockIdx.x, last = gridDim.x - 1;
    const uint32_t in_tile = tile_counts[t];
    if (in_tile == 0 && t != last) return;                     // nothing to place, and the total is the last tile's to write
    if (tid == 0) s_bits = 0;
    uint32_t part = 0;
    for (unsigned u = tid; u < t; u += 256) part += tile_counts[u];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o, 64);
    if ((tid & 63) == 0) s_w[tid >> 6] = part;
    __syncthreads();
    const uint32_t base = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    __syncthreads();
    if (t == last && tid == 0) *count = base + in_tile;
    if (in_tile == 0 || (size_t)base >= cap) return;
    uint32_t n_mine;
    const uint32_t mine = tag_fast_pass<256>(m_al, (size_t)t * TAG_TILE, v_lo, v_hi, &n_mine);
    (void)tag_place_rounds<256>(m_al, (size_t)t * TAG_TILE, v_lo, v_hi, mine, base, idx, cap, s_w, &s_bits);
}

extern "C" size_t clhip_sync_tags_ws_bytes(size_t n)
{
    return n + 15 <= TAG_SMALL ? 0 : sizeof(uint32_t) * clhip_div_up(n + 15, TAG_TILE);
}

extern "C" int clhip_sync_tags(const uint8_t *d_meta, size_t n, uint32_t *d_idx, size_t cap, uint32_t *d_count,
                               void *d_ws, void *stream)
{
    if (!d_count || (n && !d_meta) || (cap && !d_idx) || n > 0xFFFFFFFFull) { clhip_set_error("clhip_sync_tags: bad arguments"); return -1; }
    hipStream_t s = (hipStream_t)stream;
    const size_t mis = (size_t)((uintptr_t)d_meta & 15);
    const uint8_t *m_al = d_meta - mis;
    const size_t v_lo = mis, v_hi = mis + n;
    if (v_hi <= TAG_SMALL) {
        hipLaunchKernelGGL(sync_tags_small_kernel, dim3(1), dim3(1024), 0, s, m_al, v_lo, v_hi, d_idx, cap, d_count);
    } else {
        if (!d_ws) { clhip_set_error("clhip_sync_tags: %zu bytes need a workspace of clhip_sync_tags_ws_bytes()", n); return -1; }
        const unsigned tiles = (unsigned)clhip_div_up(v_hi, TAG_TILE);
        hipLaunchKernelGGL(sync_tags_count_kernel, dim3(tiles), dim3(256), 0, s, m_al, v_lo, v_hi, (uint32_t *)d_ws);
        hipLaunchKernelGGL(sync_tags_emit_kernel, dim3(tiles), dim3(256), 0, s, m_al, v_lo, v_hi, (const uint32_t *)d_ws, d_idx, cap, d_count);
    }
    CLHIP_CHECK_LAUNCH();
    return 0;
}
