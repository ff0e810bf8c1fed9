// clhip_tx.hip -- FM / CW stages and the TX pipe (BASELINE.json config 5):
//   fp32 message -> FM modulate (phase accumulated in fp64) -> L/M polyphase resample
//   -> (int16_t)(f * 4096.0f)  (soapy_api/CaribouliteStream.cpp:199-212)
//   -> int13 pack to SMI TX bytes (caribou_smi.c:684-717)
// FM / CW / resampling have no reference implementation (SURVEY.md section 8
// row a13 is the spec); the quantise + pack tail is bit-exact integer work.
//
// The phase recursion phi[n] = phi[n-1] + w m[n] is a prefix sum: per-block
// fp64 partial sums, a scan of the block sums, then a per-block scan that adds
// the block offset, wraps to (-pi, pi] and evaluates sincos in fp32.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <new>

#include "clhip_common.h"

#define TWO_PI 6.283185307179586476925286766559
#define FM_BLOCK 256
#define FM_PER_THREAD 4
#define FM_ELEMS (FM_BLOCK * FM_PER_THREAD)

__device__ __forceinline__ double wrap_pi(double ph) { return ph - TWO_PI * rint(ph * (1.0 / TWO_PI)); }

__device__ __forceinline__ f32x2 phasor(double ph)
{
    // reduce in fp64, evaluate in fp32: |err| ~ 2e-7, far inside the 1e-5 bar
    const float r = (float)(wrap_pi(ph) * (1.0 / 3.14159265358979323846));   // in [-1, 1]
    float sn, cs;
    sincospif(r, &sn, &cs);
    f32x2 o = {cs, sn};
    return o;
}

// ---------------------------------------------------------------------------
// FM demod (standalone stage): y[n] = atan2(Im z, Re z), z = x[n] conj(x[n-1])
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fm_demod_kernel(const f32x2 *__restrict__ x, size_t n,
                                                       const float *__restrict__ prev, float *__restrict__ out)
{
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; j < n; j += step) {
        const f32x2 c = x[j];
        f32x2 p;
        if (j > 0) p = x[j - 1];
        else { p.x = prev[0]; p.y = prev[1]; }
        out[j] = clhip_atan2f(c.y * p.x - c.x * p.y, c.x * p.x + c.y * p.y);
    }
}
__global__ void fm_demod_carry_kernel(const f32x2 *__restrict__ x, size_t n, float *__restrict__ prev)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) { prev[0] = x[n - 1].x; prev[1] = x[n - 1].y; }
}

extern "C" int clhip_fm_demod(const float *d_iq, size_t n, float *d_prev, float *d_out, void *stream)
{
    if (n == 0) return 0;
    if (!d_iq || !d_prev || !d_out) { clhip_set_error("clhip_fm_demod: null buffer"); return -1; }
    unsigned grid = (unsigned)clhip_div_up(n, 256);
    if (grid > 8192) grid = 8192;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(fm_demod_kernel, dim3(grid), dim3(256), 0, s, (const f32x2 *)d_iq, n, d_prev, d_out);
    hipLaunchKernelGGL(fm_demod_carry_kernel, dim3(1), dim3(64), 0, s, (const f32x2 *)d_iq, n, d_prev);
    CLHIP_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------
// CW tone: I = cos(2 pi f n / fs + phase0), Q = sin(...)   (examples/cpp_api/sync_tx_api/main.cpp:42-55)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cw_tone_kernel(double w, double phase0, size_t n, f32x2 *__restrict__ out)
{
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    // keep the product exact-ish for huge n: reduce w*j modulo 2pi in "turns"
    const double turns = w * (1.0 / TWO_PI);
    for (; j < n; j += step) {
        double t = turns * (double)j;
        t -= rint(t);
        out[j] = phasor(phase0 + t * TWO_PI);
    }
}

extern "C" int clhip_cw_tone(double f_hz, double fs_hz, double phase0, size_t n, float *d_iq_out, void *stream)
{
    if (n == 0) return 0;
    if (!d_iq_out || fs_hz <= 0) { clhip_set_error("clhip_cw_tone: bad arguments"); return -1; }
    unsigned grid = (unsigned)clhip_div_up(n, 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(cw_tone_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, TWO_PI * f_hz / fs_hz, phase0, n,
                       (f32x2 *)d_iq_out);
    CLHIP_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------
// FM modulate
// ---------------------------------------------------------------------------
// pass 1: per-block sum of w*m  (grid.y = stream)
__global__ __launch_bounds__(FM_BLOCK) void fm_block_sum_kernel(const float *__restrict__ m, long m_stride, size_t n,
                                                              double w, double *__restrict__ bsum, long n_blocks)
{
    __shared__ double sh[FM_BLOCK / 64];
    const float *mm = m + (long)blockIdx.y * m_stride;
    const size_t base = (size_t)blockIdx.x * FM_ELEMS + (size_t)threadIdx.x * FM_PER_THREAD;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < FM_PER_THREAD; k++)
        if (base + k < n) s += w * (double)mm[base + k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < FM_BLOCK / 64; k++) t += sh[k];
        bsum[(long)blockIdx.y * n_blocks + blockIdx.x] = t;
    }
}

// pass 2: exclusive scan of the block sums of one stream (one workgroup per stream);
// boff[b] = phase0 + sum_{b'<b} bsum[b'];  new phase = wrap(phase0 + total)
__global__ __launch_bounds__(256) void fm_block_scan_kernel(double *__restrict__ bsum, long n_blocks,
                                                            const double *__restrict__ phase_in,
                                                            double *__restrict__ phase_new)
{
    __shared__ double sh[256];
    double *b = bsum + (long)blockIdx.x * n_blocks;
    const int t = threadIdx.x;
    const long per = (n_blocks + 255) / 256;
    const long lo = (long)t * per, hi = lo + per < n_blocks ? lo + per : n_blocks;
    double s = 0.0;
    for (long k = lo; k < hi; k++) s += b[k];
    sh[t] = s;
    __syncthreads();
    if (t == 0) {                       // 256 partials: serial exclusive scan is plenty
        double run = phase_in[blockIdx.x];
        for (int k = 0; k < 256; k++) { const double v = sh[k]; sh[k] = run; run += v; }
        phase_new[blockIdx.x] = wrap_pi(run);
    }
    __syncthreads();
    double run = sh[t];
    for (long k = lo; k < hi; k++) { const double v = b[k]; b[k] = run; run += v; }
}

// pass 3: per-block inclusive scan + offset -> phasor
__global__ __launch_bounds__(FM_BLOCK) void fm_apply_kernel(const float *__restrict__ m, long m_stride, size_t n,
                                                          double w, const double *__restrict__ boff, long n_blocks,
                                                          f32x2 *__restrict__ out, long out_stride)
{
    __shared__ double sh[FM_BLOCK / 64];
    const float *mm = m + (long)blockIdx.y * m_stride;
    f32x2 *oo = out + (long)blockIdx.y * out_stride;
    const size_t base = (size_t)blockIdx.x * FM_ELEMS + (size_t)threadIdx.x * FM_PER_THREAD;
    double v[FM_PER_THREAD], s = 0.0;
#pragma unroll
    for (int k = 0; k < FM_PER_THREAD; k++) {
        s += (base + k < n) ? w * (double)mm[base + k] : 0.0;
        v[k] = s;                                      // inclusive within the lane
    }
    // inclusive scan of lane totals across the wave, then across waves
    double incl = s;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
    }
    if (lane == 63) sh[threadIdx.x >> 6] = incl;
    __syncthreads();
    double wave_off = 0.0;
    for (int k = 0; k < (int)(threadIdx.x >> 6); k++) wave_off += sh[k];
    const double excl = boff[(long)blockIdx.y * n_blocks + blockIdx.x] + wave_off + (incl - s);
#pragma unroll
    for (int k = 0; k < FM_PER_THREAD; k++)
        if (base + k < n) oo[base + k] = phasor(excl + v[k]);
}

__device__ __forceinline__ uint32_t tx_f2i16(float v)
{
    const int t = (v >= -2147483648.0f && v < 2147483648.0f) ? (int)v : (int)0x80000000;
    return (uint32_t)t & 0xFFFFu;
}

__device__ __forceinline__ uint32_t tx_pack_word(int mode, uint32_t ii, uint32_t qq)
{
    if (mode == CL_TX_AS_WRITTEN) { ii = 0xFFFFu; qq = 0; }       // caribou_smi.c:700-701
    // caribou_smi.c:693-711 builds s = 111 | I12..I8 | 0 I7..I1 | 0 I0 Q12..Q7 | 0 Q6..Q0 MSB-first and byte-swaps it;
    // the same word assembled directly in memory (little-endian) order, reading only the 13 low bits of ii / qq:
    //   byte0 = 0xE0 | I12..I8   byte1 = I7..I1   byte2 = I0<<6 | Q12..Q7   byte3 = Q6..Q0
    return 0xE0u | ((ii >> 8) & 0x1Fu) | ((ii << 7) & 0x7F00u) | ((ii & 1u) << 22) | ((qq << 9) & 0x3F0000u) |
           ((qq << 24) & 0x7F000000u);
}

// pass 3 fused with the TX tail (config 5): per 1024-message block
//   phase scan + offset -> phasors (kept in LDS, with the KP-1 samples before the block)
//   -> L/M polyphase leg per output -> (int16_t)(f*4096.0f) -> 13-bit pack -> one 4-byte store.
// The modulated CF32 signal never goes to HBM.  Samples before the block come from the carried
// history (first block of a call) or are rebuilt backwards from the block's phase offset.
#define TXF_HMAX 8
__global__ __launch_bounds__(FM_BLOCK) void tx_fm_fused_kernel(
    const float *__restrict__ m, long m_stride, size_t n, double w, const double *__restrict__ boff, long n_blocks,
    const f32x2 *__restrict__ hist_in, f32x2 *__restrict__ hist_out, int H,
    const float *__restrict__ rs, int n_rs, int L, int M, unsigned long long n0, long n_out, int pack_mode,
    uint32_t *__restrict__ words, long w_stride, f32x2 *__restrict__ tap, long tap_stride)
{
    __shared__ double sh[FM_BLOCK / 64];
    __shared__ f32x2 xs[TXF_HMAX + FM_ELEMS];          // xs[TXF_HMAX + i] = modulated sample base+i
    const int s = blockIdx.y, t = threadIdx.x;
    const float *mm = m + (long)s * m_stride;
    const size_t base = (size_t)blockIdx.x * FM_ELEMS, tb = base + (size_t)t * FM_PER_THREAD;
    const double off = boff[(long)s * n_blocks + blockIdx.x];      // phase after sample base-1
    double v[FM_PER_THREAD], sum = 0.0;
#pragma unroll
    for (int k = 0; k < FM_PER_THREAD; k++) {
        sum += (tb + k < n) ? w * (double)mm[tb + k] : 0.0;
        v[k] = sum;
    }
    double incl = sum;
    const int lane = t & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
    }
    if (lane == 63) sh[t >> 6] = incl;
    // the H samples before the block
    if (t < H) {
        const int k = t + 1;                             // sample base - k
        f32x2 hv;
        if (base >= (size_t)k) {                         // inside this call: phase = off - sum of the k-1 increments after it
            double ph = off;
            for (int i = 1; i < k; i++) ph -= w * (double)mm[base - i];
            hv = phasor(ph);
        } else {                                         // before this call: carried history (oldest first, H entries)
            const long idx = (long)H - (long)(k - (long)base);
            hv = idx >= 0 ? hist_in[(long)s * H + idx] : f32x2{0.f, 0.f};
        }
        xs[TXF_HMAX - k] = hv;
    }
    __syncthreads();
    double wave_off = 0.0;
    for (int k = 0; k < (t >> 6); k++) wave_off += sh[k];
    const double excl = off + wave_off + (incl - sum);
#pragma unroll
    for (int k = 0; k < FM_PER_THREAD; k++)
        xs[TXF_HMAX + t * FM_PER_THREAD + k] = (tb + k < n) ? phasor(excl + v[k]) : f32x2{0.f, 0.f};
    __syncthreads();
    // history for the next call: the last H modulated samples of the stream
    if (H > 0 && base + FM_ELEMS >= n && t < H) {
        const long g = (long)n - H + t;                  // sample index inside this call
        const long rel = g - (long)base;                 // relative to this block
        f32x2 hv;
        if (rel >= -TXF_HMAX) hv = xs[TXF_HMAX + rel];
        else hv = f32x2{0.f, 0.f};                       // call shorter than H spanning >1 block cannot happen (H <= 7 < FM_ELEMS)
        if (g < 0) {                                     // call shorter than H: shift the old history
            const long idx = (long)H + g;
            hv = idx >= 0 ? hist_in[(long)s * H + idx] : f32x2{0.f, 0.f};
        }
        hist_out[(long)s * H + t] = hv;
    }
    // outputs whose newest input sample lies in this block: b in [base, base + FM_ELEMS) and b < n
    const unsigned long long g0 = n0 + base, g1 = n0 + ((base + FM_ELEMS < n) ? base + FM_ELEMS : n);
    const unsigned long long m_first = (g0 * L + M - 1) / M, m_end = (g1 * L + M - 1) / M;
    const unsigned long long m_call0 = (n0 * L + M - 1) / M;
    const int KP = (n_rs + L - 1) / L;
    // 64-bit index arithmetic once per block (uniform); 32-bit per output:
    //   mo*M - g0*L = r0 + dm*M with r0 = m_first*M - g0*L in [0, M)
    const unsigned r0 = (unsigned)(m_first * M - g0 * L);
    const unsigned n_mo = (unsigned)(m_end - m_first);
    for (unsigned dm = t; dm < n_mo; dm += FM_BLOCK) {
        const unsigned long long mo = m_first + dm;
        const unsigned tp = r0 + dm * (unsigned)M;
        const int b = (int)(tp / (unsigned)L);           // newest input, relative to the block
        const int p = (int)(tp % (unsigned)L);
        f32x2 acc = {0.f, 0.f};
        for (int i = 0; i < KP; i++) {
            const int k = p + i * L;
            if (k >= n_rs) break;
            acc += xs[TXF_HMAX + b - i] * rs[k];
        }
        const long j = (long)(mo - m_call0);
        if (j < n_out) {
            if (tap) tap[(long)s * tap_stride + j] = acc;
            words[(long)s * w_stride + j] = tx_pack_word(pack_mode, tx_f2i16(acc.x * 4096.0f), tx_f2i16(acc.y * 4096.0f));
        }
    }
}

// ---------------------------------------------------------------------------
// Config 5 fast path: FM message -> phasors -> L/M polyphase -> quantise -> pack, instantiated per
// (L, M, KP) like the RX pipe's fused kernels; every other shape takes tx_fm_fused_kernel above.
//
// Phase is kept in TURNS (fp64): phi_t[n] = phi_t[n-1] + (kf/fs) m[n].  v_fract_f64 wraps it, and
// v_sin_f32 / v_cos_f32 take turns directly, so a phasor costs fract + cvt + two transcendental ops.
// One workgroup per superblock of TXQ_NSUB sub-blocks (tx_fm_chain_kernel; small calls: one sub-block per workgroup,
// tx_fm_chain1_kernel): the superblock's fp64 sum, a decoupled look-back over the sums before it, then its sub-blocks of
// TXQ_SUB messages with a running fp64 offset; each lane owns PER consecutive messages (local fp64 prefix, wave scan by
// DPP), writes its phasors to a padded LDS row, then computes PER*L/M outputs from an 8-row-sample-overlapping register
// window with the polyphase taps in SGPRs (compile-time phases), quantises, packs and stores 16-byte pieces.
// (Round 3 built and measured around this: a three-launch form (sums, scan, fused), messages kept in registers / in LDS
// between the two passes, a register prefetch of the next sub-block, stores transposed through LDS, four look-back words
// per lane, a barrier-free sub-block loop, a two-role launch of summers and workers -- DESIGN.md section 8 has the numbers;
// none was faster and none is here any more.)
// LDS rows: PER samples (8 B each) + 16 B pad -> lane pitch 112 B at PER = 12: ds_write_b128 /
// ds_read_b128 are conflict-free.  Row -1 holds the 8 samples before the sub-block.
// ---------------------------------------------------------------------------
#ifndef TXQ_NT
#define TXQ_NT 256                         // lanes per workgroup.  Config 5 through the chain kernel, one box (round 3): 64 lanes x 12 / 24 sub-blocks
#endif                                     // 0.43-0.50 ms, 128 x 6 / 12: 0.29 / 0.39, 256 x 6: 0.269-0.277, 512 x 3 / 4 / 6: 0.32 / 0.31 / 0.31, 1024 x 3: 0.37
                                           // -- barriers are not what it waits for: smaller workgroups mean more look-back words per message
// sub-blocks per superblock (round 2, look-back in front of the arithmetic): 4 -> 0.290 ms, 5 -> 0.276, 6 -> 0.268, 7 -> 0.276,
// 8 -> 0.274 (config 5, 2^27 messages); round 3, at five waves per SIMD (96 VGPRs: 7 and 8 sub-blocks spill): 4 -> 0.281, 5 -> 0.277,
// 6 -> 0.270, 7 -> 0.288-0.298, 8 -> 0.326 (tools/bench_tx.py, one box)
#ifndef TXQ_NSUB
#define TXQ_NSUB 8                         // (round 3, join-free interior path at four waves per SIMD: 6 -> 0.2447-0.2471 ms, 8 -> 0.2396-0.2398, 12 -> 0.298)
#endif
#ifndef TXQ_CHAIN_WAVES
#define TXQ_CHAIN_WAVES 4                  // waves per SIMD the chain kernel's register budget is cut for.  5 (96 VGPRs) was round 3's first choice; with the
#endif                                     // interior superblocks on a path of their own (tx_chain_work<C, true>) the sum pass wants its 18-24 loads in flight
                                           // at once, which 96 registers cannot hold without spilling: 4 waves (109 VGPRs, no spills) 0.2447 ms against 0.2487 at 5
typedef __attribute__((address_space(4))) float tx_cfloat_t;
typedef f32x4 __attribute__((aligned(4))) f32x4_a4;      // 16-byte global access at dword alignment (unaligned mode)
typedef u32x4 __attribute__((aligned(4))) u32x4_a4;

// Inclusive prefix sum of one double per lane over the wave's 64 lanes, by DPP: row_shr 1, 2, 4, 8 inside the rows of 16,
// then row_bcast:15 (rows 1 and 3 take the totals of rows 0 and 2) and row_bcast:31 (rows 2 and 3 take the total of the
// lower half).  Six steps of two v_mov_dpp and one v_add_f64 -- no LDS round trip; the shuffle form (__shfl_up = two
// ds_bpermute_b32 per step and a select) was 0.7 us of a sub-block's 3.3 in the config-5 kernel's dependent chain.
// Lanes a step does not reach add +0.0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_take_f64(double v)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, ROW_MASK, 0xF, false);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double wave_inclusive_scan_f64(double v)
{
    v += dpp_take_f64<0x111, 0xF>(v);          // row_shr:1
    v += dpp_take_f64<0x112, 0xF>(v);          // row_shr:2
    v += dpp_take_f64<0x114, 0xF>(v);          // row_shr:4
    v += dpp_take_f64<0x118, 0xF>(v);          // row_shr:8
    v += dpp_take_f64<0x142, 0xA>(v);          // row_bcast:15 -> rows 1, 3
    v += dpp_take_f64<0x143, 0xC>(v);          // row_bcast:31 -> rows 2, 3
    return v;
}

template <int L_, int M_, int KP_> struct TxCfg {
    static constexpr int L = L_, M = M_, KP = KP_;
    static constexpr int PER = 4 * M_;                   // messages per lane per sub-block (float4 loads, whole output groups)
    static constexpr int NOUT = PER * L_ / M_;           // outputs per lane per sub-block
    static constexpr int SUB = TXQ_NT * PER, SB = SUB * TXQ_NSUB;
    static constexpr int ROW = PER * 8 + 16;             // LDS bytes per lane row
    static constexpr int HSLOT = 8;                      // samples kept before the sub-block (>= KP - 1)
    static_assert(KP_ - 1 <= HSLOT && HSLOT <= PER, "history must fit the tail of one row");
    static_assert(NOUT % 4 == 0, "16-byte stores of packed words");
};

__device__ __forceinline__ f32x2 phasor_turns(double t)
{
    const float r = (float)__builtin_amdgcn_fract(t);    // [0, 1): abs error 2^-25 turns = 1.9e-7 rad
    f32x2 o = {__builtin_amdgcn_cosf(r), __builtin_amdgcn_sinf(r)};
    return o;
}

// Index space of the config-5 kernels: VIRTUAL message index i' = i + phi, phi = n0 mod M, so that i' = 0 sits on
// polyphase phase 0 whatever the call's start; the phi virtual messages before the call carry no phase
// increment (their phasors come from the history) and the outputs they own were emitted by the previous call.

// (int16_t)(f * 4096.0f) as the x86-64 reference build does it (cvttss2si, low 16 bits): v_cvt_i32_f32
// saturates where cvttss2si returns 0x80000000, which only changes the low half for f >= 2^31; NaN gives 0 on both
__device__ __forceinline__ uint32_t tx_f2i16_fast(float v)
{
    const int t = v < 2147483648.0f ? (int)v : 0;
    return (uint32_t)t & 0xFFFFu;
}
// The config-5 kernels quantise resampled UNIT phasors: the host has shown |v| < 2^31 (clhip_tx_pipe::q_bounded), so the
// overflow branch cannot be taken and v_cvt_i32_f32 alone has cvttss2si's low half (NaN -> 0 on both): one instruction
// where the general form needs a compare and a select as well.
__device__ __forceinline__ uint32_t tx_f2i16_bounded(float v) { return (uint32_t)(int)v & 0xFFFFu; }

#ifndef TXQ_STAMPS
#define TXQ_STAMPS 0                       // diagnostic build (tools/tx_phase_stamps.py): thread 0 of every workgroup stamps s_memrealtime at the phase boundaries
#endif
#define TX_STAMP_WGS 8192
#define TX_STAMP_N 32                      // 0-5 the superblock's phases, 5 + sb the end of sub-block sb (sb < 16), 24-27 inside sub-block 3
#if TXQ_STAMPS
__device__ unsigned long long g_tx_stamps[TX_STAMP_WGS * TX_STAMP_N];
#define TXS(i) do { if (threadIdx.x == 0) g_tx_stamps[(size_t)(stamp_id & (TX_STAMP_WGS - 1)) * TX_STAMP_N + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TXS(i) do { } while (0)
#endif
template <class C>
__device__ __forceinline__ void tx_store_words(const uint32_t (&wd)[C::NOUT], uint32_t *wp)
{
#pragma unroll
    for (int q = 0; q < C::NOUT / 4; q++) { const u32x4 v = {wd[4 * q], wd[4 * q + 1], wd[4 * q + 2], wd[4 * q + 3]}; *(u32x4_a4 *)(wp + 4 * q) = v; }
}

// One sub-block in the absolute frame (its start phase known).  CHECKED = false: every message and every output of the sub-block is inside
// the call (workgroup-uniform), so there is not a single bounds test or divergent branch in it.
template <class C, bool CHECKED>
__device__ __forceinline__ void tx_load_msgs(const float *mm, size_t n, int phi, size_t tb, float (&mv)[C::PER])
{
    constexpr int PER = C::PER;
    if (!CHECKED || (tb >= (size_t)phi && tb + PER <= n)) {
#pragma unroll
        for (int q = 0; q < PER / 4; q++) {
            const f32x4 v = *(const f32x4_a4 *)(mm + tb + 4 * q);
            mv[4 * q] = v.x; mv[4 * q + 1] = v.y; mv[4 * q + 2] = v.z; mv[4 * q + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < PER; k++) mv[k] = (tb + k >= (size_t)phi && tb + k < n) ? mm[tb + k] : 0.f;
    }
}

// unchecked, the address as (workgroup-uniform base) + (32-bit lane offset): one offset register serves every sub-block,
// the bases live in SGPRs (the 64-bit lane addresses of the general form cost a register pair per sub-block in flight)
template <class C>
__device__ __forceinline__ void tx_load_msgs_u(const float *ub, unsigned lane_off, float (&mv)[C::PER])
{
#pragma unroll
    for (int q = 0; q < C::PER / 4; q++) {
        const f32x4 v = *(const f32x4_a4 *)(ub + (lane_off + 4u * q));
        mv[4 * q] = v.x; mv[4 * q + 1] = v.y; mv[4 * q + 2] = v.z; mv[4 * q + 3] = v.w;
    }
}

template <class C, bool CHECKED>
__device__ __forceinline__ double tx_fast_subblock(const float (&mv)[C::PER], size_t n, int phi, int skip, double wt, double off,
                                                   size_t base, unsigned char *rows, double *sh, const f32x2 *hist_in_s,
                                                   f32x2 *hist_out_s, const tx_cfloat_t *__restrict__ rs, long n_out,
                                                   int pack_mode, uint32_t *words_s, f32x2 *tap_s, unsigned stamp_id = 0, bool stamped = false)
{
    constexpr int L = C::L, M = C::M, KP = C::KP, PER = C::PER, NOUT = C::NOUT, ROW = C::ROW, HS = C::HSLOT, H = KP - 1;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    unsigned char *myrow = rows + (t + 1) * ROW;
    const size_t tb = base + (size_t)t * PER;
    if (TXQ_STAMPS && stamped) {                               // (diagnostic build) the messages have landed
#pragma unroll
        for (int k = 0; k < PER; k++) asm volatile("" :: "v"(mv[k]));
        TXS(24);
    }
    // local fp64 prefix (turns)
    double c[PER], run = 0.0;
#pragma unroll
    for (int k = 0; k < PER; k++) { run = __builtin_fma((double)mv[k], wt, run); c[k] = run; }
    const double incl = wave_inclusive_scan_f64(run);
    if (lane == 63) sh[wave] = incl;
    __syncthreads();                                           // also: the previous sub-block's window reads are done
    if (TXQ_STAMPS && stamped) TXS(25);
    double woff = off, total = 0.0;
#pragma unroll
    for (int k = 0; k < TXQ_NT / 64; k++) { const double v = sh[k]; total += v; if (k < wave) woff += v; }
    const double excl = woff + (incl - run);
#pragma unroll
    for (int k = 0; k < PER; k += 2) {
        f32x2 a = phasor_turns(excl + c[k]), b = phasor_turns(excl + c[k + 1]);
        if (CHECKED) {
            if (tb + k >= n) a = f32x2{0.f, 0.f};
            if (tb + k + 1 >= n) b = f32x2{0.f, 0.f};
            if (tb == 0) {                                     // virtual messages before the call: carried history
                if (k < phi) a = hist_in_s[H - phi + k];
                if (k + 1 < phi) b = hist_in_s[H - phi + k + 1];
            }
        }
        const f32x4 q = {a.x, a.y, b.x, b.y};
        *(f32x4 *)(myrow + 8 * k) = q;
    }
    off += total; off -= floor(off);
    __syncthreads();
    if (TXQ_STAMPS && stamped) TXS(26);
    // history for the next call: the last H phasors of the stream live in this sub-block's rows (or row -1)
    if (CHECKED && H > 0 && base + C::SUB >= n && t < H) {
        const long rel = (long)n - H + t - (long)base;         // >= -H
        const long r = rel >= 0 ? rel / PER : -1, cidx = rel >= 0 ? rel % PER : PER + rel;
        hist_out_s[t] = *(const f32x2 *)(rows + (r + 1) * ROW + 8 * cidx);
    }
    // window: the HS samples before the lane's first message + its PER messages
    f32x2 x[HS + PER];
#pragma unroll
    for (int k = 0; k < HS; k += 2) {
        const f32x4 q = *(const f32x4 *)(myrow - ROW + 8 * (PER - HS + k));
        x[k] = q.xy; x[k + 1] = q.zw;
    }
#pragma unroll
    for (int k = 0; k < PER; k += 2) {
        const f32x4 q = *(const f32x4 *)(myrow + 8 * k);
        x[HS + k] = q.xy; x[HS + k + 1] = q.zw;
    }
    float tp[KP * L];
#pragma unroll
    for (int i = 0; i < KP * L; i++) tp[i] = rs[i];
    uint32_t wd[NOUT];
    f32x2 o[NOUT];
#pragma unroll
    for (int u = 0; u < NOUT; u++) {
        const int bb = (u * M) / L, p = (u * M) % L;          // newest message (lane-relative), polyphase leg
        f32x2 acc = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < KP; i++) acc += x[HS + bb - i] * tp[p + i * L];
        o[u] = acc;
        wd[u] = tx_pack_word(CL_TX_DOCUMENTED, tx_f2i16_bounded(acc.x), tx_f2i16_bounded(acc.y));      // (the taps carry the 4096)
    }
    if (pack_mode == CL_TX_AS_WRITTEN) {                       // uniform: the shipped packer ignores its input
#pragma unroll
        for (int u = 0; u < NOUT; u++) wd[u] = tx_pack_word(CL_TX_AS_WRITTEN, 0, 0);
    }
    if (TXQ_STAMPS && stamped) {
#pragma unroll
        for (int u = 0; u < NOUT; u++) asm volatile("" :: "v"(wd[u]));
        TXS(27);
    }
    const long j0 = (long)(tb / M) * L - skip;                 // tb is a multiple of M; the first `skip` outputs are not ours
    uint32_t *wp = words_s + j0;
    if (!CHECKED) {
        tx_store_words<C>(wd, wp);
    } else if (j0 >= 0 && j0 + NOUT <= n_out) {
#pragma unroll
        for (int q = 0; q < NOUT / 4; q++) { const u32x4 v = {wd[4 * q], wd[4 * q + 1], wd[4 * q + 2], wd[4 * q + 3]}; *(u32x4_a4 *)(wp + 4 * q) = v; }
    } else {
#pragma unroll
        for (int u = 0; u < NOUT; u++) if (j0 + u >= 0 && j0 + u < n_out) wp[u] = wd[u];
    }
    if (tap_s) {
#pragma unroll
        for (int u = 0; u < NOUT; u++) if (!CHECKED || (j0 + u >= 0 && j0 + u < n_out)) tap_s[j0 + u] = o[u] * (1.0f / 4096.0f);   // exact
    }
    // the last HS samples of this sub-block become row -1 of the next one
    f32x2 keep = {0.f, 0.f};
    if (t < HS) keep = *(const f32x2 *)(rows + TXQ_NT * ROW + 8 * (PER - HS + t));
    __syncthreads();
    if (t < HS) *(f32x2 *)(rows + 8 * (PER - HS + t)) = keep;
    return off;
}

// ---------------------------------------------------------------------------
// Single-pass variant: the message is read ONCE.  Each workgroup takes a ticket (so that every superblock
// before it is owned by a workgroup that is already running), sums its 18432 messages, publishes
// its fp64 aggregate, and finds its phase offset by decoupled look-back over its predecessors' aggregates /
// inclusive prefixes (flags carry the launch epoch: no reset between launches).  The poll loop is bounded: on
// overrun it raises *err and carries on with a wrong phase instead of hanging the GPU.
// ---------------------------------------------------------------------------
// One 64-bit word per superblock, written and polled with relaxed agent-scope atomics (no fences, hence no
// L2 write-back / invalidate traffic): [63:62] state {1: aggregate, 2: inclusive prefix} | [61:48] launch epoch |
// [47:0] phase in turns mod 1 as 48-bit fixed point.  Integer sums mod 2^48 are exact and associative, so the
// offset a workgroup finds does not depend on how far back it had to look.
struct TxLookBack {
    unsigned long long *st;          // [n_streams][n_super]
    unsigned int *ticket;            // monotonically increasing across launches
    unsigned int ticket_base, epoch; // epoch in 1 .. 16383
    int *err;
    int poll_bound;                  // polls of one predecessor word before giving up
};
#define TXLB_MASK 0xFFFFFFFFFFFFull
#define TXLB_W 1
__device__ __forceinline__ unsigned long long txlb_fix(double turns)
{
    const double f = turns - floor(turns);
    return (unsigned long long)(f * 281474976710656.0) & TXLB_MASK;       // 2^48
}

// ---------------------------------------------------------------------------
// A sub-block worked RELATIVE to its own start (phase 0 there), for when the phase it starts at is not known yet:
//   tx_unit_compute   fp64 prefix inside the sub-block -> phasors -> LDS rows (row -1: the HS samples before the
//                     sub-block, rebuilt from the H messages before it -- relative phasors need no offset) -> polyphase
//                     outputs, unrotated.  Returns the sub-block's phase sum (turns).
//   tx_unit_emit      rotate the outputs by the phase the sub-block starts at -- the resampler is linear, so rotating its
//                     outputs equals rotating its inputs -- quantise, pack, store; the stream's history for the next call.
// The stream's first sub-block (base == 0) knows its phase, the carried one, takes the carried phasors as they are, and
// is emitted with the identity rotation.
// ---------------------------------------------------------------------------
template <class C>
__device__ __forceinline__ double tx_unit_compute(const float *mm, size_t n, int phi, double wt, double start, size_t base, bool interior,
                                                  unsigned char *rows, double *sh, const f32x2 *hist_in_s, const tx_cfloat_t *__restrict__ rs,
                                                  f32x2 (&o)[C::NOUT], const float *pre_regs = nullptr)
{
    constexpr int L = C::L, M = C::M, KP = C::KP, PER = C::PER, NOUT = C::NOUT, ROW = C::ROW, HS = C::HSLOT, H = KP - 1;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const size_t tb = base + (size_t)t * PER;
    unsigned char *myrow = rows + (t + 1) * ROW;
    float mv[PER];
    if (pre_regs) {                                                // (inlined: the caller's registers)
#pragma unroll
        for (int k = 0; k < PER; k++) mv[k] = pre_regs[k];
    } else if (interior) tx_load_msgs<C, false>(mm, n, phi, tb, mv); else tx_load_msgs<C, true>(mm, n, phi, tb, mv);
    double c[PER], run = 0.0;
#pragma unroll
    for (int k = 0; k < PER; k++) { run = __builtin_fma((double)mv[k], wt, run); c[k] = run; }
    const double incl = wave_inclusive_scan_f64(run);
    if (lane == 63) sh[wave] = incl;
    __syncthreads();                                           // also: an earlier sub-block's window reads are done
    double woff = start, total = 0.0;
#pragma unroll
    for (int k = 0; k < TXQ_NT / 64; k++) { const double v = sh[k]; total += v; if (k < wave) woff += v; }
    const double excl = woff + (incl - run);
#pragma unroll
    for (int k = 0; k < PER; k += 2) {
        f32x2 a = phasor_turns(excl + c[k]), bb = phasor_turns(excl + c[k + 1]);
        if (!interior) {
            if (tb + k >= n) a = f32x2{0.f, 0.f};
            if (tb + k + 1 >= n) bb = f32x2{0.f, 0.f};
            if (tb == 0) {                                         // virtual messages before the call: carried history (absolute frame)
                if (k < phi) a = hist_in_s[H - phi + k];
                if (k + 1 < phi) bb = hist_in_s[H - phi + k + 1];
            }
        }
        const f32x4 q = {a.x, a.y, bb.x, bb.y};
        *(f32x4 *)(myrow + 8 * k) = q;
    }
    if (t < HS) {                                                  // the HS samples before the sub-block -> tail of row -1
        const int k = t + 1;                                       // message base - k
        f32x2 hv = {0.f, 0.f};
        if (k <= H) {
            if (base >= (size_t)k) {
                double ph = start;
                for (int i = 1; i < k; i++) ph -= wt * (double)mm[base - i];
                hv = phasor_turns(ph);
            } else {                                               // base == 0: real message -k - phi
                const long idx = (long)H - k - phi;
                if (idx >= 0) hv = hist_in_s[idx];
            }
        }
        *(f32x2 *)(rows + 8 * (PER - k)) = hv;
    }
    __syncthreads();
    f32x2 x[HS + PER];                                             // window: the HS samples before the lane's first message + its PER messages
#pragma unroll
    for (int k = 0; k < HS; k += 2) {
        const f32x4 q = *(const f32x4 *)(myrow - ROW + 8 * (PER - HS + k));
        x[k] = q.xy; x[k + 1] = q.zw;
    }
#pragma unroll
    for (int k = 0; k < PER; k += 2) {
        const f32x4 q = *(const f32x4 *)(myrow + 8 * k);
        x[HS + k] = q.xy; x[HS + k + 1] = q.zw;
    }
    float tp[KP * L];
#pragma unroll
    for (int i = 0; i < KP * L; i++) tp[i] = rs[i];
#pragma unroll
    for (int u = 0; u < NOUT; u++) {
        const int bq = (u * M) / L, p = (u * M) % L;          // newest message (lane-relative), polyphase leg
        f32x2 acc = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < KP; i++) acc += x[HS + bq - i] * tp[p + i * L];
        o[u] = acc;
    }
    return total;
}

template <class C>
__device__ __forceinline__ void tx_unit_emit(f32x2 (&o)[C::NOUT], const f32x2 rot, size_t n, int skip, size_t base, bool interior,
                                             unsigned char *rows, f32x2 *hist_out_s, long n_out, int pack_mode, uint32_t *words_s, f32x2 *tap_s)
{
    constexpr int L = C::L, M = C::M, KP = C::KP, PER = C::PER, NOUT = C::NOUT, ROW = C::ROW, H = KP - 1;
    const int t = threadIdx.x;
    const size_t tb = base + (size_t)t * PER;
    // history for the next call: the last H phasors of the stream live in this sub-block's rows (or row -1), unrotated
    if (!interior && H > 0 && base + C::SUB >= n && t < H) {
        const long rel = (long)n - H + t - (long)base;             // >= -H
        const long r = rel >= 0 ? rel / PER : -1, cidx = rel >= 0 ? rel % PER : PER + rel;
        const f32x2 hv = *(const f32x2 *)(rows + (r + 1) * ROW + 8 * cidx);
        hist_out_s[t] = f32x2{hv.x * rot.x - hv.y * rot.y, hv.x * rot.y + hv.y * rot.x};
    }
    uint32_t wd[NOUT];
#pragma unroll
    for (int u = 0; u < NOUT; u++) {
        const f32x2 v = {__builtin_fmaf(o[u].x, rot.x, -o[u].y * rot.y), __builtin_fmaf(o[u].x, rot.y, o[u].y * rot.x)};
        o[u] = v;
        wd[u] = tx_pack_word(CL_TX_DOCUMENTED, tx_f2i16_bounded(v.x), tx_f2i16_bounded(v.y));          // (the taps carry the 4096)
    }
    if (pack_mode == CL_TX_AS_WRITTEN) {                       // uniform: the shipped packer ignores its input
#pragma unroll
        for (int u = 0; u < NOUT; u++) wd[u] = tx_pack_word(CL_TX_AS_WRITTEN, 0, 0);
    }
    const long j0 = (long)(tb / M) * L - skip;                 // tb is a multiple of M; the first `skip` outputs are not ours
    uint32_t *wp = words_s + j0;
    if (interior) {                                            // (workgroup-uniform)
        tx_store_words<C>(wd, wp);
    } else if (j0 >= 0 && j0 + NOUT <= n_out) {
#pragma unroll
        for (int q = 0; q < NOUT / 4; q++) { const u32x4 v = {wd[4 * q], wd[4 * q + 1], wd[4 * q + 2], wd[4 * q + 3]}; *(u32x4_a4 *)(wp + 4 * q) = v; }
    } else {
#pragma unroll
        for (int u = 0; u < NOUT; u++) if (j0 + u >= 0 && j0 + u < n_out) wp[u] = wd[u];
    }
    if (tap_s) {
#pragma unroll
        for (int u = 0; u < NOUT; u++) if (interior || (j0 + u >= 0 && j0 + u < n_out)) tap_s[j0 + u] = o[u] * (1.0f / 4096.0f);   // exact
    }
}

// The decoupled look-back of one wave over the words before st[b] (see TxLookBack): the phase, as 48-bit fixed point,
// after the last message before unit b.  Bounded polls; an overrun raises *lb.err and carries on with a made-up word.
__device__ __forceinline__ unsigned long long tx_look_back(const TxLookBack &lb, const unsigned long long *st, long b, unsigned long long pin, int lane)
{
    // one word per lane and round: the window back to the nearest finished prefix is ~150 words in the steady state of config 5
    // (three dependent round trips past the XCD's L2, tools/tx_phase_stamps.py: 5.1 us; four words per lane and round were slower)
    const unsigned long long e = (unsigned long long)lb.epoch << 48;
    unsigned long long acc = 0;
    long j0 = b - 1;
    bool done = false;
    int guard = 0;
    while (!done) {
        unsigned long long w[TXLB_W];
        bool ok;
        do {
            ok = true;
#pragma unroll
            for (int i = 0; i < TXLB_W; i++) {
                const long j = j0 - (long)TXLB_W * lane - i;
                w[i] = j >= 0 ? __hip_atomic_load(st + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                              : ((3ull << 62) | e | pin);          // state 3 = before the stream's first unit
            }
#pragma unroll
            for (int i = 0; i < TXLB_W; i++) ok = ok && ((w[i] >> 48) & 0x3FFF) == lb.epoch && (w[i] >> 62) != 0;
            if (++guard > lb.poll_bound) {                         // bounded: give up, say so, carry on with made-up words
                *lb.err = 1;
#pragma unroll
                for (int i = 0; i < TXLB_W; i++) w[i] = (2ull << 62) | e;
                ok = true;
            }
        } while (!ok);
        // the lane's own words, nearest first: sum up to and including its first finished prefix
        unsigned long long mine = 0;
        bool have = false;
#pragma unroll
        for (int i = 0; i < TXLB_W; i++) {
            if (!have) mine += w[i] & TXLB_MASK;
            have = have || (w[i] >> 62) >= 2;
        }
        const unsigned long long hv = __ballot(have);
        const int first = hv ? __builtin_ctzll(hv) : 64;
        unsigned long long v = lane <= first ? mine : 0ull;
#pragma unroll
        for (int oo = 32; oo > 0; oo >>= 1) v += __shfl_xor(v, oo, 64);
        acc += v;
        if (first < 64) done = true; else j0 -= 64 * TXLB_W;
    }
    return acc & TXLB_MASK;
}

// The work of one superblock.  INT = true: the superblock lies strictly inside the call (all but the stream's first and
// last one or two), every bounds test is gone at compile time and so is every join between a checked and an unchecked
// variant of a sub-block: with those joins in the instruction stream the compiler's wait-count insertion (which has to
// cover the pending loads of EITHER arm) put `s_waitcnt vmcnt(1)` in front of the third and every later sub-block's loads
// of the sum pass, so the pass made four round trips to memory with three loads in flight instead of one with eighteen.
template <class C, bool INT>
__device__ __forceinline__ void tx_chain_work(
    const float *__restrict__ mm, size_t n, int phi, int skip, double wt, const TxLookBack &lb, long n_super, int s, long b,
    size_t sbase, const double *__restrict__ phase_in, double *__restrict__ phase_new,
    const f32x2 *__restrict__ hist_in, f32x2 *__restrict__ hist_out, const tx_cfloat_t *__restrict__ rs,
    long n_out, int pack_mode, uint32_t *__restrict__ words, long w_stride, f32x2 *__restrict__ tap, long tap_stride,
    unsigned char *rows, double *sh, double &sh_off)
{
    constexpr int KP = C::KP, PER = C::PER, ROW = C::ROW, HS = C::HSLOT, H = KP - 1;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const bool interior = INT || (sbase > 0 && sbase + C::SB < n);   // workgroup-uniform: all sub-blocks unchecked
    const unsigned stamp_id = (unsigned)b;
    TXS(0);

    // aggregate of the superblock; the messages are read again from L2 / Infinity Cache by the sub-block loop
    // (keeping 48 of them per lane in registers would cost two waves of occupancy, and occupancy hides the look-back)
    double part = 0.0;
    constexpr bool CARRY = INT;                                    // interior superblocks: sub-block 0's messages stay in registers from the sum pass to
    float mv0[PER], mv1[PER];                                      // its arithmetic, and sub-block 1's are requested before the look-back instead of after it
#pragma unroll
    for (int sbr = 0; sbr < TXQ_NSUB; sbr++) {
        const int sb = TXQ_NSUB - 1 - sbr;                          // last sub-block first: the second pass then starts on the most recently read lines (-4 % HBM reads)
        const size_t tb = sbase + (size_t)sb * C::SUB + (size_t)t * PER;
        float mv[PER];
        if (INT) tx_load_msgs_u<C>(mm + (sbase + (size_t)sb * C::SUB), (unsigned)t * PER, mv);
        else if (interior) tx_load_msgs<C, false>(mm, n, phi, tb, mv);
        else tx_load_msgs<C, true>(mm, n, phi, tb, mv);
        double r = 0.0;
#pragma unroll
        for (int k = 0; k < PER; k++) { r = __builtin_fma((double)mv[k], wt, r); if (CARRY && sb == 0) mv0[k] = mv[k]; }
        part += r;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o, 64);
    if (lane == 0) sh[wave] = part;
    __syncthreads();
    TXS(1);
    unsigned long long *st = lb.st + (long)s * n_super;
    const unsigned long long e = (unsigned long long)lb.epoch << 48;
    const double pin_turns = phase_in[s] * (1.0 / TWO_PI);
    unsigned long long mine = 0;
    if (wave == 0) {
        double total = 0.0;
#pragma unroll
        for (int k = 0; k < TXQ_NT / 64; k++) total += sh[k];
        mine = txlb_fix(total);
        if (lane == 0) __hip_atomic_store(st + b, (1ull << 62) | e | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();                                               // sh[] is free again
    TXS(2);
    uint32_t *words_s = words + (long)s * w_stride;
    f32x2 *tap_s = tap ? tap + (long)s * tap_stride : nullptr;
    // The superblock's FIRST sub-block is worked before its phase is known -- relative to its own start, its outputs held
    // back (tx_unit_compute) -- so that the look-back stands behind a sixth of the superblock's arithmetic instead of
    // in front of all of it: by then the predecessors have long published.  (Ablation, round 2: no look-back wait 0.253
    // against 0.281 ms.)  The stream's first superblock knows its phase and works in the absolute frame throughout.
    f32x2 o0[C::NOUT];
    const bool int0 = INT || (sbase > 0 && sbase + C::SUB < n);
    const double tot0 = tx_unit_compute<C>(mm, n, phi, wt, b == 0 ? pin_turns : 0.0, sbase, int0, rows, sh, hist_in + (long)s * H, rs, o0,
                                           CARRY ? mv0 : nullptr);
    if (CARRY) tx_load_msgs_u<C>(mm + (sbase + (size_t)C::SUB), (unsigned)t * PER, mv1);   // in flight across the look-back
    TXS(3);
    if (wave == 0) {
        const unsigned long long pin = txlb_fix(pin_turns);
        const unsigned long long acc = b > 0 ? tx_look_back(lb, st, b, pin, lane) : pin;
        if (lane == 0) {
            const unsigned long long inc = (acc + mine) & TXLB_MASK;
            __hip_atomic_store(st + b, (2ull << 62) | e | inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sh_off = (double)acc * (1.0 / 281474976710656.0);     // phase (turns) after message sbase-1
            if (b == n_super - 1) {
                const double it = (double)inc * (1.0 / 281474976710656.0);
                phase_new[s] = wrap_pi(TWO_PI * (it - rint(it)));
            }
        }
    }
    __syncthreads();
    TXS(4);
    const double off0 = b == 0 ? pin_turns : sh_off;
    const f32x2 rot = phasor_turns(b == 0 ? 0.0 : off0);
    tx_unit_emit<C>(o0, rot, n, skip, sbase, int0, rows, hist_out + (long)s * H, n_out, pack_mode, words_s, tap_s);
    // the last HS samples of sub-block 0, turned into the absolute frame, become row -1 of sub-block 1
    f32x2 keep = {0.f, 0.f};
    if (t < HS) {
        const f32x2 kv = *(const f32x2 *)(rows + TXQ_NT * ROW + 8 * (PER - HS + t));
        keep = f32x2{kv.x * rot.x - kv.y * rot.y, kv.x * rot.y + kv.y * rot.x};
    }
    __syncthreads();
    if (t < HS) *(f32x2 *)(rows + 8 * (PER - HS + t)) = keep;
    double off = off0 + tot0;
    off -= floor(off);
    TXS(5);
#pragma unroll
    for (int sb = 1; sb < TXQ_NSUB; sb++) {
        const size_t base = sbase + (size_t)sb * C::SUB;          // workgroup-uniform
        if (!INT && base >= n) break;
        float mv[PER];
        if (CARRY && sb == 1) {
#pragma unroll
            for (int k = 0; k < PER; k++) mv[k] = mv1[k];
        }
        if (INT || (base > 0 && base + C::SUB < n)) {
            if (!(CARRY && sb == 1)) {
                if (INT) tx_load_msgs_u<C>(mm + base, (unsigned)t * PER, mv);
                else tx_load_msgs<C, false>(mm, n, phi, base + (size_t)t * PER, mv);
            }
            off = tx_fast_subblock<C, false>(mv, n, phi, skip, wt, off, base, rows, sh, hist_in + (long)s * H,
                                             hist_out + (long)s * H, rs, n_out, pack_mode, words_s, tap_s, stamp_id, sb == 3);
        } else {
            tx_load_msgs<C, true>(mm, n, phi, base + (size_t)t * PER, mv);
            off = tx_fast_subblock<C, true>(mv, n, phi, skip, wt, off, base, rows, sh, hist_in + (long)s * H,
                                            hist_out + (long)s * H, rs, n_out, pack_mode, words_s, tap_s);
        }
        TXS(5 + sb);
    }
}

template <class C>
__global__ __launch_bounds__(TXQ_NT, TXQ_CHAIN_WAVES) void tx_fm_chain_kernel(
    const float *__restrict__ m, long m_stride, size_t n, int phi, int skip, double wt, TxLookBack lb, long n_super,
    int n_streams, const double *__restrict__ phase_in, double *__restrict__ phase_new,
    const f32x2 *__restrict__ hist_in, f32x2 *__restrict__ hist_out, const float *__restrict__ rs_dev,
    long n_out, int pack_mode, uint32_t *__restrict__ words, long w_stride, f32x2 *__restrict__ tap, long tap_stride)
{
    constexpr int ROW = C::ROW;
    __shared__ __attribute__((aligned(16))) unsigned char rows[(TXQ_NT + 1) * ROW];   // row r at (r + 1) * ROW
    __shared__ double sh[TXQ_NT / 64 + 1];
    __shared__ double sh_off;
    __shared__ unsigned int sh_ticket;
    const int t = threadIdx.x;
    // Order: workgroups are dispatched in index order, so every superblock before this one belongs to a
    // workgroup that has at least been dispatched and does not wait for us.  lb.ticket != NULL: take a ticket
    // instead (same-address atomics retire at ~12 ns each, which throttles the start of 10^4 workgroups).
    if (lb.ticket) {
        if (t == 0) sh_ticket = atomicAdd(lb.ticket, 1u) - lb.ticket_base;
        __syncthreads();
    }
    // (read through readfirstlane: a value that has been through LDS counts as divergent, and with it every address, pointer
    // and loop bound derived from the superblock's number would live in vector registers)
    const unsigned int T = __builtin_amdgcn_readfirstlane(lb.ticket ? sh_ticket : blockIdx.x);
    if (T >= (unsigned)(n_super * n_streams)) {                    // (workgroup-uniform) never index past the launch's superblocks --
        if (t == 0) *lb.err = 1;                                   // and never silently: a skipped superblock fails the call
        return;
    }
    const int s = (int)(T % (unsigned)n_streams);
    const long b = (long)(T / (unsigned)n_streams);
    const float *mm = m + (long)s * m_stride - phi;
    const size_t sbase = (size_t)b * C::SB;
    const tx_cfloat_t *__restrict__ rs = (const tx_cfloat_t *)rs_dev;
    if (sbase > 0 && sbase + C::SB < n)                            // workgroup-uniform
        tx_chain_work<C, true>(mm, n, phi, skip, wt, lb, n_super, s, b, sbase, phase_in, phase_new, hist_in, hist_out, rs, n_out, pack_mode,
                               words, w_stride, tap, tap_stride, rows, sh, sh_off);
    else
        tx_chain_work<C, false>(mm, n, phi, skip, wt, lb, n_super, s, b, sbase, phase_in, phase_new, hist_in, hist_out, rs, n_out, pack_mode,
                                words, w_stride, tap, tap_stride, rows, sh, sh_off);
}

// ---------------------------------------------------------------------------
// Single read, one sub-block per workgroup: the kernel of calls up to 2^22 messages (clhip_tx_pipe_run picks by size).  The look-back unit is ONE
// sub-block of 256 x PER messages, the whole of which a workgroup holds in registers and LDS: every message is read
// from memory once, the unit's sum is published at once and its look-back stands BEHIND its own arithmetic.  Correct
// and traffic-minimal, but a look-back per 3072 messages costs more than it saves: 0.394 ms on config 5 against 0.269
// for the superblock kernel (a unit's arithmetic is ~3 us, its look-back ~8 us with a thousand equally young units
// resident; the ticket order, one atomic per unit, 0.54 ms).
// ---------------------------------------------------------------------------
template <class C>
__global__ __launch_bounds__(TXQ_NT, 4) void tx_fm_chain1_kernel(
    const float *__restrict__ m, long m_stride, size_t n, int phi, int skip, double wt, TxLookBack lb, long n_units,
    int n_streams, const double *__restrict__ phase_in, double *__restrict__ phase_new,
    const f32x2 *__restrict__ hist_in, f32x2 *__restrict__ hist_out, const float *__restrict__ rs_dev,
    long n_out, int pack_mode, uint32_t *__restrict__ words, long w_stride, f32x2 *__restrict__ tap, long tap_stride)
{
    constexpr int KP = C::KP, ROW = C::ROW, H = KP - 1;
    __shared__ __attribute__((aligned(16))) unsigned char rows[(TXQ_NT + 1) * ROW];   // row r at (r + 1) * ROW
    __shared__ double sh[TXQ_NT / 64 + 1];
    __shared__ double sh_off;
    __shared__ unsigned int sh_ticket;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (lb.ticket) {
        if (t == 0) sh_ticket = atomicAdd(lb.ticket, 1u) - lb.ticket_base;
        __syncthreads();
    }
    const unsigned int T = lb.ticket ? sh_ticket : blockIdx.x;
    if (T >= (unsigned)(n_units * n_streams)) {                    // (workgroup-uniform) a ticket beyond the launch's units: the host's
        if (t == 0) *lb.err = 1;                                   // base is out of step -- say so, a skipped unit must not go unnoticed
        return;
    }
    const int s = (int)(T % (unsigned)n_streams);
    const long b = (long)(T / (unsigned)n_streams);
    const float *mm = m + (long)s * m_stride - phi;                // indexed by the virtual message index; n = virtual count
    const size_t base = (size_t)b * C::SUB;
    const bool interior = base > 0 && base + C::SUB < n;          // workgroup-uniform: no bounds tests, no history
    unsigned long long *st = lb.st + (long)s * n_units;
    const unsigned long long e = (unsigned long long)lb.epoch << 48;
    const double pin_turns = phase_in[s] * (1.0 / TWO_PI);

    f32x2 o[C::NOUT];
    const double total = tx_unit_compute<C>(mm, n, phi, wt, b == 0 ? pin_turns : 0.0, base, interior, rows, sh, hist_in + (long)s * H,
                                            (const tx_cfloat_t *)rs_dev, o);
    const unsigned long long mine = txlb_fix(total);
    // (the aggregate leaves behind the arithmetic here: one barrier earlier it would need a second pass over the sums)
    if (t == 0) __hip_atomic_store(st + b, (1ull << 62) | e | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wave == 0) {
        const unsigned long long pin = txlb_fix(pin_turns);
        const unsigned long long acc = b > 0 ? tx_look_back(lb, st, b, pin, lane) : pin;
        if (lane == 0) {
            const unsigned long long inc = (acc + mine) & TXLB_MASK;
            __hip_atomic_store(st + b, (2ull << 62) | e | inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sh_off = b > 0 ? (double)acc * (1.0 / 281474976710656.0) : 0.0;      // (the first unit's phasors are absolute already)
            if (b == n_units - 1) {
                const double it = (double)inc * (1.0 / 281474976710656.0);
                phase_new[s] = wrap_pi(TWO_PI * (it - rint(it)));
            }
        }
    }
    __syncthreads();
    tx_unit_emit<C>(o, phasor_turns(sh_off), n, skip, base, interior, rows, hist_out + (long)s * H, n_out, pack_mode,
                    words + (long)s * w_stride, tap ? tap + (long)s * tap_stride : nullptr);
}

typedef TxCfg<2, 3, 8> TxCfgC5;      // config 5: 2/3 resampler, 16 prototype taps

extern "C" size_t clhip_fm_mod_workspace_bytes(size_t n) { return (clhip_div_up(n, FM_ELEMS) + 2) * sizeof(double); }

static int fm_mod_launch(const float *d_msg, long m_stride, size_t n, int n_streams, double w, double *d_phase,
                         f32x2 *d_out, long out_stride, double *ws, hipStream_t s)
{
    const long n_blocks = (long)clhip_div_up(n, FM_ELEMS);
    dim3 grid((unsigned)n_blocks, n_streams);
    hipLaunchKernelGGL(fm_block_sum_kernel, grid, dim3(FM_BLOCK), 0, s, d_msg, m_stride, n, w, ws, n_blocks);
    // phase_new is written to a scratch slot first: pass 3 of other streams may still need nothing from
    // d_phase, but keep in/out distinct within the launch that reads it
    double *phase_new = ws + (size_t)n_blocks * n_streams;
    hipLaunchKernelGGL(fm_block_scan_kernel, dim3(n_streams), dim3(256), 0, s, ws, n_blocks, d_phase, phase_new);
    hipLaunchKernelGGL(fm_apply_kernel, grid, dim3(FM_BLOCK), 0, s, d_msg, m_stride, n, w, ws, n_blocks, d_out,
                       out_stride);
    CLHIP_CHECK(hipMemcpyAsync(d_phase, phase_new, sizeof(double) * n_streams, hipMemcpyDeviceToDevice, s));
    CLHIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int clhip_fm_mod(const float *d_msg, size_t n, double kf_hz, double fs_hz, double *d_phase,
                            float *d_iq_out, void *d_ws, size_t ws_bytes, void *stream)
{
    if (n == 0) return 0;
    if (!d_msg || !d_phase || !d_iq_out || !d_ws || fs_hz <= 0 || ws_bytes < clhip_fm_mod_workspace_bytes(n)) {
        clhip_set_error("clhip_fm_mod: bad arguments or workspace too small");
        return -1;
    }
    return fm_mod_launch(d_msg, 0, n, 1, TWO_PI * kf_hz / fs_hz, d_phase, (f32x2 *)d_iq_out, 0, (double *)d_ws,
                         (hipStream_t)stream);
}

// ---------------------------------------------------------------------------
// TX pipe
// ---------------------------------------------------------------------------
#define TX_MAX_RS 40

struct clhip_tx_pipe {
    int n_streams, n_rs, L, M, kp, pack_mode;
    double w;                        // 2 pi kf / fs
    float rs[TX_MAX_RS];
    float *d_rs;
    float *d_rs_q;                   // the taps x 4096 (exact: a power of two) for the config-5 kernels, whose quantiser then needs no multiply
    f32x2 *hist[2]; int cur;         // [n_streams][kp-1] modulated samples preceding the call
    double *d_phase2; int pcur;      // [2][n_streams] ping-pong: the single-launch path writes the other half directly
    double *d_phase;                 // = d_phase2 + pcur * n_streams: the current phase of every stream
    unsigned long long undo_n_total; int undo_cur, undo_pcur; bool can_undo;   // pre-call state of the last run
    int poll_bound;                  // look-back poll bound (diagnostic knob, default 2^22)
    hipStream_t last_stream; bool last_stream_valid;
    bool force_ticket;               // a look-back poll overran once in dispatch order: this pipe orders by ticket from now on
    bool q_bounded;                  // FM mode: |resampled phasor| x 4096 < 2^31 whatever the message (finite taps, sum |taps| bounded)
    unsigned long long n_total;
    f32x2 *Y; size_t y_cap;          // modulated signal workspace (per stream)
    double *ws; size_t ws_cap;
    // single-pass (look-back) state of the config-5 path
    unsigned long long *lb_st; size_t lb_cap;                            // per stream x superblock
    unsigned int *lb_ticket; int *lb_err, *lb_err_dev; unsigned int ticket_total, epoch;   // lb_err: pinned host word the kernel can raise
};

// one output per lane: upfirdn polyphase leg -> quantise -> pack
__global__ __launch_bounds__(256) void tx_resample_pack_kernel(const f32x2 *__restrict__ x, long x_stride,
                                                               const f32x2 *__restrict__ hist, int H,
                                                               const float *__restrict__ rs, int n_rs, int L, int M,
                                                               unsigned long long n0, long n_out, int pack_mode,
                                                               uint32_t *__restrict__ words, long w_stride,
                                                               f32x2 *__restrict__ tap, long tap_stride)
{
    const int s = blockIdx.y;
    const f32x2 *xs = x + (long)s * x_stride;
    const f32x2 *hs = hist + (long)s * H;
    const unsigned long long m0 = (n0 * L + M - 1) / M;
    const int KP = (n_rs + L - 1) / L;
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < n_out; j += (long)gridDim.x * blockDim.x) {
        const unsigned long long tp = (m0 + j) * M;
        const long b = (long)(tp / L - n0);
        const int p = (int)(tp % L);
        f32x2 acc = {0.f, 0.f};
        for (int i = 0; i < KP; i++) {
            const int k = p + i * L;
            if (k >= n_rs) break;
            const long idx = b - i;
            f32x2 v;
            if (idx >= 0) v = xs[idx];
            else if (H + idx >= 0) v = hs[H + idx];
            else { v.x = 0.f; v.y = 0.f; }
            acc += v * rs[k];
        }
        if (tap) tap[(long)s * tap_stride + j] = acc;
        // (int16_t)(f * 4096.0f)  CaribouliteStream.cpp:207-208
        words[(long)s * w_stride + j] = tx_pack_word(pack_mode, tx_f2i16(acc.x * 4096.0f), tx_f2i16(acc.y * 4096.0f));
    }
}

__global__ void tx_update_hist_kernel(const f32x2 *__restrict__ x, long x_stride, long n, const f32x2 *__restrict__ hin,
                                      f32x2 *__restrict__ hout, int H)
{
    const int s = blockIdx.x;
    for (int j = threadIdx.x; j < H; j += blockDim.x) {
        const long g = n - H + j;
        f32x2 v;
        if (g >= 0) v = x[(long)s * x_stride + g];
        else v = hin[(long)s * H + H + g];
        hout[(long)s * H + j] = v;
    }
}

extern "C" clhip_tx_pipe *clhip_tx_pipe_create(int n_streams, double fm_kf_hz, double fs_hz, const float *h_rs,
                                               int n_rs, int up, int down, int pack_mode)
{
    if (n_streams <= 0 || up <= 0 || down <= 0 || fs_hz <= 0) { clhip_set_error("clhip_tx_pipe_create: bad arguments"); return nullptr; }
    const bool resamp = !(up == 1 && down == 1);
    if (resamp && (!h_rs || n_rs <= 0 || n_rs > TX_MAX_RS)) {
        clhip_set_error("clhip_tx_pipe_create: resampler needs 1..%d prototype taps", TX_MAX_RS);
        return nullptr;
    }
    clhip_tx_pipe *p = new (std::nothrow) clhip_tx_pipe();
    if (!p) return nullptr;
    memset(p, 0, sizeof *p);
    p->n_streams = n_streams; p->L = up; p->M = down; p->pack_mode = pack_mode;
    p->w = TWO_PI * fm_kf_hz / fs_hz;
    if (resamp) { p->n_rs = n_rs; memcpy(p->rs, h_rs, sizeof(float) * n_rs); }
    else { p->n_rs = 1; p->rs[0] = 1.0f; }
    p->kp = (p->n_rs + up - 1) / up;
    {   // the FM modulator's resampler input is a unit phasor (or NaN, from a NaN / infinite message): its outputs are bounded by
        // the l1 norm of the taps, and the quantiser of the config-5 kernels may then skip cvttss2si's overflow branch
        double l1 = 0;
        bool finite = true;
        for (int i = 0; i < p->n_rs; i++) { finite = finite && isfinite(p->rs[i]); l1 += fabs((double)p->rs[i]); }
        p->q_bounded = finite && l1 * 4096.0 * 1.5 < 2147483648.0;      // (x 1.5: both components through the fp32 rotation)
    }
    const int H = p->kp - 1 > 0 ? p->kp - 1 : 1;
    p->d_rs = (float *)clhip_malloc(sizeof p->rs);
    p->d_phase2 = (double *)clhip_malloc(sizeof(double) * 2 * n_streams);
    p->d_phase = p->d_phase2; p->pcur = 0;
    p->poll_bound = getenv("CLHIP_TX_POLL_BOUND") ? atoi(getenv("CLHIP_TX_POLL_BOUND")) : -1;   // diagnostic knob; -1 = by ordering mode
    for (int i = 0; i < 2; i++) p->hist[i] = (f32x2 *)clhip_malloc(sizeof(f32x2) * H * n_streams);
    if (!p->d_rs || !p->d_phase || !p->hist[0] || !p->hist[1]) { clhip_tx_pipe_destroy(p); return nullptr; }
    (void)hipMemcpy(p->d_rs, p->rs, sizeof p->rs, hipMemcpyHostToDevice);
    {
        float q[TX_MAX_RS];
        for (int i = 0; i < TX_MAX_RS; i++) q[i] = p->rs[i] * 4096.0f;
        p->d_rs_q = (float *)clhip_malloc(sizeof q);
        if (!p->d_rs_q) { clhip_tx_pipe_destroy(p); return nullptr; }
        (void)hipMemcpy(p->d_rs_q, q, sizeof q, hipMemcpyHostToDevice);
    }
    clhip_tx_pipe_reset(p);
    return p;
}

extern "C" void clhip_tx_pipe_destroy(clhip_tx_pipe *p)
{
    if (!p) return;
    clhip_free(p->d_rs); clhip_free(p->d_rs_q); clhip_free(p->d_phase2); clhip_free(p->hist[0]); clhip_free(p->hist[1]);
    clhip_free(p->Y); clhip_free(p->ws);
    clhip_free(p->lb_st); clhip_free(p->lb_ticket);
    if (p->lb_err) (void)hipHostFree(p->lb_err);
    delete p;
}

extern "C" void clhip_tx_pipe_reset(clhip_tx_pipe *p)
{
    const int H = p->kp - 1 > 0 ? p->kp - 1 : 1;
    if (p->last_stream_valid) (void)hipStreamSynchronize(p->last_stream);      // a run may still be using the state
    (void)hipMemset(p->d_phase2, 0, sizeof(double) * 2 * p->n_streams);
    p->pcur = 0; p->d_phase = p->d_phase2; p->can_undo = false;
    if (p->lb_err) *p->lb_err = 0;
    (void)hipMemset(p->hist[0], 0, sizeof(f32x2) * H * p->n_streams);
    (void)hipMemset(p->hist[1], 0, sizeof(f32x2) * H * p->n_streams);
    p->cur = 0; p->n_total = 0;
    (void)hipStreamSynchronize(nullptr);       // the fills ran on the null stream; non-blocking streams do not wait for it
}

// Place a pipe inside a long stream (time slicing over GPUs): as after clhip_tx_pipe_reset, but the next
// run is message n_total of the stream (polyphase phase n_total mod M) and the modulator's phase before it is h_phase_rad[s]
// -- the 8-byte hand-off a slice owner receives: the phase is a prefix sum of the messages, the one state of the TX pipe
// that no halo can rebuild.  The resampler history is zero: the owner runs the kp - 1 (or more) messages before its slice
// first and discards what they produce (cariboulite_amd/shard.py run_fm_time_slice).
extern "C" int clhip_tx_pipe_seek(clhip_tx_pipe *p, unsigned long long n_total, const double *h_phase_rad)
{
    if (!p || !h_phase_rad) { clhip_set_error("clhip_tx_pipe_seek: null argument"); return -1; }
    clhip_tx_pipe_reset(p);
    CLHIP_CHECK(hipMemcpy(p->d_phase, h_phase_rad, sizeof(double) * p->n_streams, hipMemcpyHostToDevice));
    p->n_total = n_total;
    return 0;
}

// After the caller has synchronised the stream of the last clhip_tx_pipe_run: 0 = its output is valid.  -1 = the
// look-back guard fired (a workgroup gave up waiting for a predecessor's phase and carried on with a made-up one):
// the bytes of that call must not be used; the pipe is put back to its pre-call state (phase, resampler history,
// polyphase phase are ping-pong / host state the failed launch did not overwrite), so the call can be repeated.
extern "C" int clhip_tx_pipe_status(clhip_tx_pipe *p)
{
    if (!p) return -1;
    if (!p->lb_err || !*(volatile int *)p->lb_err) return 0;
    *p->lb_err = 0;
    p->force_ticket = true;
    if (p->can_undo) {
        p->cur = p->undo_cur; p->pcur = p->undo_pcur; p->n_total = p->undo_n_total;
        p->d_phase = p->d_phase2 + (size_t)p->pcur * p->n_streams;
        p->can_undo = false;
    }
    clhip_set_error("clhip_tx_pipe_status: look-back poll overran; the output of the last call is invalid, pipe state restored");
    return -1;
}

extern "C" void clhip_tx_pipe_set_poll_bound(clhip_tx_pipe *p, int polls) { if (p) p->poll_bound = polls < 0 ? -1 : polls; }

extern "C" size_t clhip_tx_pipe_out_count(const clhip_tx_pipe *p, size_t n_in)
{
    const unsigned long long n0 = p->n_total, n1 = n0 + n_in;
    return (size_t)((n1 * p->L + p->M - 1) / p->M - (n0 * p->L + p->M - 1) / p->M);
}

// The carried state of ONE stream -- the modulator's phase and the resampler's history -- moves from one pipe to another of the same
// configuration (a stream group's multi-stream pipe <-> a member's own: cl_group_writeStream).  The polyphase position is the PIPE's
// (its streams advance together): _position / _set_position.  Both pipes idle: their last runs synchronised and asked for their verdict.
extern "C" unsigned long long clhip_tx_pipe_position(const clhip_tx_pipe *p) { return p ? p->n_total : 0; }
extern "C" int clhip_tx_pipe_pack_mode(const clhip_tx_pipe *p) { return p ? p->pack_mode : -1; }

extern "C" int clhip_tx_pipe_set_position(clhip_tx_pipe *p, unsigned long long n_total)
{
    if (!p) { clhip_set_error("clhip_tx_pipe_set_position: null pipe"); return -1; }
    p->n_total = n_total; p->can_undo = false;
    return 0;
}

extern "C" int clhip_tx_pipe_move_stream(clhip_tx_pipe *dst, int ds, clhip_tx_pipe *src, int ss, void *stream)
{
    if (!dst || !src || ds < 0 || ss < 0 || ds >= dst->n_streams || ss >= src->n_streams) { clhip_set_error("clhip_tx_pipe_move_stream: bad arguments"); return -1; }
    if (dst->L != src->L || dst->M != src->M || dst->n_rs != src->n_rs || dst->w != src->w || memcmp(dst->rs, src->rs, sizeof(float) * (size_t)src->n_rs)) {
        clhip_set_error("clhip_tx_pipe_move_stream: the pipes are configured differently");
        return -1;
    }
    hipStream_t s = (hipStream_t)stream;
    const int H = src->kp - 1;
    if (H > 0)
        CLHIP_CHECK(hipMemcpyAsync(dst->hist[dst->cur] + (size_t)ds * H, src->hist[src->cur] + (size_t)ss * H, sizeof(f32x2) * (size_t)H, hipMemcpyDeviceToDevice, s));
    CLHIP_CHECK(hipMemcpyAsync(dst->d_phase + ds, src->d_phase + ss, sizeof(double), hipMemcpyDeviceToDevice, s));
    CLHIP_CHECK(hipStreamSynchronize(s));
    dst->can_undo = false; src->can_undo = false;      // (a roll-back would bring the other halves of the ping-pong state back)
    return 0;
}

extern "C" long clhip_tx_pipe_run(clhip_tx_pipe *p, int in_kind, const void *d_in, size_t in_stride, size_t n_in,
                                  uint8_t *d_bytes, size_t out_stride_bytes, float *d_iq_tap, size_t iq_tap_stride,
                                  void *stream)
{
    if (!p || (in_kind != CL_TXPIPE_IN_FM_MESSAGE && in_kind != CL_TXPIPE_IN_CF32)) {
        clhip_set_error("clhip_tx_pipe_run: bad arguments");
        return -1;
    }
    if (n_in == 0) return 0;
    if (!d_in || !d_bytes || (out_stride_bytes & 3) || (((uintptr_t)d_bytes) & 3)) {
        clhip_set_error("clhip_tx_pipe_run: null / misaligned buffer");
        return -1;
    }
    hipStream_t s = (hipStream_t)stream;
    const size_t n_out = clhip_tx_pipe_out_count(p, n_in);
    const int H = p->kp - 1;
    p->undo_n_total = p->n_total; p->undo_cur = p->cur; p->undo_pcur = p->pcur; p->can_undo = true;
    p->last_stream = s; p->last_stream_valid = true;
    const f32x2 *x = (const f32x2 *)d_in;
    long x_stride = (long)in_stride;
    if (p->q_bounded && in_kind == CL_TXPIPE_IN_FM_MESSAGE && p->L == TxCfgC5::L && p->M == TxCfgC5::M &&
        p->n_rs == TxCfgC5::KP * TxCfgC5::L && (((uintptr_t)d_in) & 3) == 0) {
        // config 5 instantiation, in the virtual index space i' = i + phi
        typedef TxCfgC5 C;
        const int phi = (int)(p->n_total % (unsigned long long)p->M), skip = (phi * p->L + p->M - 1) / p->M;
        const size_t nv = n_in + (size_t)phi;
        const long n_super = (long)clhip_div_up(nv, (size_t)C::SB);
        const double wt = p->w * (1.0 / TWO_PI);
        // By size: calls of up to 2^22 messages take the one-sub-block-per-workgroup kernel -- an MTU is 43 workgroups there and 6
        // superblocks of 8 sequential sub-blocks in the chain kernel: 7.4 us against 17 (tools/bench_tx.py: 2^20 8.4 / 17.6, 2^21 10.9 /
        // 18.2, 2^22 17.7 / 20.9, 2^23 30.5 / 25.9, 2^24 55 / 38) -- larger ones the chain kernel (superblocks of TXQ_NSUB sub-blocks, the
        // look-back behind the first sub-block's arithmetic).  Both are single launches with a single look-back.
        const bool one_sub = (size_t)nv * (size_t)p->n_streams <= ((size_t)1 << 22);
        const long n_units = (long)clhip_div_up(nv, (size_t)C::SUB);
        {
            const long n_lb = one_sub ? n_units : n_super;        // look-back words per stream
            const size_t need = (size_t)n_lb * p->n_streams;
            if (need > p->lb_cap || !p->lb_ticket) {
                clhip_free(p->lb_st);
                p->lb_st = (unsigned long long *)clhip_malloc(sizeof(unsigned long long) * need);
                if (!p->lb_ticket) {
                    p->lb_ticket = (unsigned int *)clhip_malloc(sizeof(unsigned int));
                    if (p->lb_ticket) CLHIP_CHECK(hipMemsetAsync(p->lb_ticket, 0, sizeof(unsigned int), s));
                    if (hipHostMalloc((void **)&p->lb_err, sizeof(int), hipHostMallocMapped) != hipSuccess) p->lb_err = nullptr;
                    if (p->lb_err) {
                        *p->lb_err = 0;
                        if (hipHostGetDevicePointer((void **)&p->lb_err_dev, p->lb_err, 0) != hipSuccess) p->lb_err_dev = nullptr;
                    }
                }
                p->lb_cap = p->lb_st ? need : 0;
                if (!p->lb_cap || !p->lb_ticket || !p->lb_err || !p->lb_err_dev) {
                    clhip_set_error("clhip_tx_pipe_run: look-back state allocation failed");
                    return -1;
                }
                CLHIP_CHECK(hipMemsetAsync(p->lb_st, 0, sizeof(unsigned long long) * need, s));   // epoch 0 is never used
            }
            if (++p->epoch > 0x3FFF) {                              // 14-bit epoch wrapped: forget every old entry
                p->epoch = 1;
                CLHIP_CHECK(hipMemsetAsync(p->lb_st, 0, sizeof(unsigned long long) * p->lb_cap, s));
            }
            // Forward progress.  Default: superblocks are ordered by blockIdx -- workgroups are dispatched in index order on
            // this hardware (not by contract), so a predecessor is always at least dispatched; 7 % faster than a ticket
            // per workgroup (profiles/r02/c5_lookback_*_bench.json: the ticket's atomic round trip sits in front of
            // every workgroup's first load).  Correctness does not rest on that order: the poll is bounded, an
            // overrun is reported by clhip_tx_pipe_status() for the SAME call with the pipe rolled back, and from
            // then on this pipe takes tickets (every predecessor then belongs to a running workgroup that waits only
            // for ITS predecessors: the look-back ends whatever the dispatch order).
            const int use_ticket = p->force_ticket ? 1 : 0;
            if (*(volatile int *)p->lb_err) {                       // raised by an earlier launch nobody asked about
                clhip_set_error("clhip_tx_pipe_run: a look-back poll overran in an earlier call and clhip_tx_pipe_status() "
                                "was not consulted; output of that call is invalid");
                *p->lb_err = 0;
                p->can_undo = false;
                return -1;
            }
            // dispatch order: give up after ~0.2 s of polling and let the ticket order take over; tickets: the wait is always finite
            const int bound = p->poll_bound >= 0 ? p->poll_bound : (use_ticket ? 1 << 22 : 1 << 18);
            TxLookBack lb = {p->lb_st, use_ticket ? p->lb_ticket : nullptr, p->ticket_total, p->epoch, p->lb_err_dev, bound};
            const unsigned n_wg = (unsigned)(n_lb * p->n_streams);
            double *phase_new = p->d_phase2 + (size_t)(p->pcur ^ 1) * p->n_streams;   // the other half: late workgroups still read d_phase
            if (one_sub)
                hipLaunchKernelGGL(tx_fm_chain1_kernel<C>, dim3(n_wg), dim3(TXQ_NT), 0, s, (const float *)d_in, (long)in_stride, nv, phi,
                                   skip, wt, lb, n_units, p->n_streams, (const double *)p->d_phase, phase_new, p->hist[p->cur],
                                   p->hist[p->cur ^ 1], p->d_rs_q, (long)n_out, p->pack_mode, (uint32_t *)d_bytes,
                                   (long)(out_stride_bytes / 4), (f32x2 *)d_iq_tap, (long)iq_tap_stride);
            else
                hipLaunchKernelGGL(tx_fm_chain_kernel<C>, dim3(n_wg), dim3(TXQ_NT), 0, s, (const float *)d_in, (long)in_stride, nv, phi,
                                   skip, wt, lb, n_super, p->n_streams, (const double *)p->d_phase, phase_new, p->hist[p->cur],
                                   p->hist[p->cur ^ 1], p->d_rs_q, (long)n_out, p->pack_mode, (uint32_t *)d_bytes,
                                   (long)(out_stride_bytes / 4), (f32x2 *)d_iq_tap, (long)iq_tap_stride);
            if (use_ticket) p->ticket_total += n_wg;               // the device counter moves only when tickets are taken
            CLHIP_CHECK_LAUNCH();
            p->pcur ^= 1; p->d_phase = phase_new;
            p->cur ^= 1;
            p->n_total += n_in;
            return (long)n_out;
        }
    }
    if (in_kind == CL_TXPIPE_IN_FM_MESSAGE && H <= TXF_HMAX - 1) {
        // fused path: block sums -> block scan -> phasor + resample + quantise + pack in one kernel
        const long n_blocks = (long)clhip_div_up(n_in, FM_ELEMS);
        const size_t wsn = (size_t)(n_blocks + 2) * p->n_streams + 8;
        if (wsn > p->ws_cap) {
            clhip_free(p->ws);
            p->ws = (double *)clhip_malloc(sizeof(double) * wsn);
            p->ws_cap = p->ws ? wsn : 0;
            if (!p->ws) return -1;
        }
        dim3 grid((unsigned)n_blocks, p->n_streams);
        double *phase_new = p->ws + (size_t)n_blocks * p->n_streams;
        hipLaunchKernelGGL(fm_block_sum_kernel, grid, dim3(FM_BLOCK), 0, s, (const float *)d_in, (long)in_stride, n_in,
                           p->w, p->ws, n_blocks);
        hipLaunchKernelGGL(fm_block_scan_kernel, dim3(p->n_streams), dim3(256), 0, s, p->ws, n_blocks, p->d_phase, phase_new);
        hipLaunchKernelGGL(tx_fm_fused_kernel, grid, dim3(FM_BLOCK), 0, s, (const float *)d_in, (long)in_stride, n_in, p->w,
                           p->ws, n_blocks, p->hist[p->cur], p->hist[p->cur ^ 1], H, p->d_rs, p->n_rs, p->L, p->M,
                           p->n_total, (long)n_out, p->pack_mode, (uint32_t *)d_bytes, (long)(out_stride_bytes / 4),
                           (f32x2 *)d_iq_tap, (long)iq_tap_stride);
        CLHIP_CHECK(hipMemcpyAsync(p->d_phase, phase_new, sizeof(double) * p->n_streams, hipMemcpyDeviceToDevice, s));
        CLHIP_CHECK_LAUNCH();
        if (H > 0) p->cur ^= 1;
        p->n_total += n_in;
        return (long)n_out;
    }
    if (in_kind == CL_TXPIPE_IN_FM_MESSAGE) {           // long resampler history: unfused fallback
        if (n_in > p->y_cap) {
            clhip_free(p->Y);
            p->Y = (f32x2 *)clhip_malloc(sizeof(f32x2) * n_in * p->n_streams);
            p->y_cap = p->Y ? n_in : 0;
            if (!p->Y) return -1;
        }
        const size_t wsn = (clhip_div_up(n_in, FM_ELEMS) + 2) * p->n_streams + 8;
        if (wsn > p->ws_cap) {
            clhip_free(p->ws);
            p->ws = (double *)clhip_malloc(sizeof(double) * wsn);
            p->ws_cap = p->ws ? wsn : 0;
            if (!p->ws) return -1;
        }
        if (fm_mod_launch((const float *)d_in, (long)in_stride, n_in, p->n_streams, p->w, p->d_phase, p->Y,
                          (long)p->y_cap, p->ws, s))
            return -1;
        x = p->Y; x_stride = (long)p->y_cap;
    }
    if (n_out) {
        unsigned gx = (unsigned)clhip_div_up(n_out, 256);
        if (gx > 4096) gx = 4096;
        hipLaunchKernelGGL(tx_resample_pack_kernel, dim3(gx, p->n_streams), dim3(256), 0, s, x, x_stride,
                           p->hist[p->cur], H, p->d_rs, p->n_rs, p->L, p->M, p->n_total, (long)n_out, p->pack_mode,
                           (uint32_t *)d_bytes, (long)(out_stride_bytes / 4), (f32x2 *)d_iq_tap, (long)iq_tap_stride);
    }
    if (H > 0) {
        hipLaunchKernelGGL(tx_update_hist_kernel, dim3(p->n_streams), dim3(64), 0, s, x, x_stride, (long)n_in,
                           p->hist[p->cur], p->hist[p->cur ^ 1], H);
        p->cur ^= 1;
    }
    CLHIP_CHECK_LAUNCH();
    p->n_total += n_in;
    return (long)n_out;
}

extern "C" int clhip_tx_debug_nsub(void) { return TXQ_NSUB; }
extern "C" int clhip_tx_debug_stamps(void *h_out)
{
#if TXQ_STAMPS
    if (h_out) CLHIP_CHECK(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_tx_stamps), sizeof(unsigned long long) * TX_STAMP_WGS * TX_STAMP_N));
    return TX_STAMP_WGS * TX_STAMP_N;
#else
    (void)h_out;
    return 0;
#endif
}
