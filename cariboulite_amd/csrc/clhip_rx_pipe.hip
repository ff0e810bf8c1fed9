// clhip_rx_pipe.hip -- the RX pipe on gfx950:
//   raw SMI words -> int13 I/Q (caribou_smi.c:338-378) -> x/4096
//   (CaribouliteStream.cpp:315-321) -> FIR(T) -> [L/M polyphase | FM demod]
//
// Fused kernel (one launch, intermediates never leave the CU):
//   * a workgroup owns NT*R consecutive FIR outputs of one stream; the raw
//     words of that tile plus T + HFA samples of halo are loaded with 16-byte
//     coalesced loads, unpacked ONCE, and staged in LDS as (I,Q) float pairs
//     (bank-conflict-free padded layout, see lds_off());
//   * every lane then computes R consecutive FIR outputs from a sliding
//     window: each staged sample is read from LDS once per lane and feeds up
//     to R packed-f32 FMAs (taps live in SGPRs, pre-scaled by 1/4096 so the
//     int->float scale costs nothing and stays bit-identical);
//   * the last K-1 FIR outputs of each lane are handed to the next lane
//     through LDS, the polyphase legs (or the phase-difference demod) run
//     from registers, and the results leave with 16-byte stores.
// No MFMA: these are 1-D tap dot products (BASELINE.json north_star).
//
// Generic path (gen_* kernels): a second, plain implementation of the same
// spec -- any T / L / M / call length / phase -- used when no fused
// instantiation matches and as an independent cross-check in tests.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <new>

#include "clhip_common.h"

typedef __attribute__((address_space(4))) float cfloat_t;   // constant address space (scalar-loadable)

#ifndef CLHIP_FFA_DPP
#define CLHIP_FFA_DPP 0
#endif
// Timing / energy ablations of the fused kernel (tools/c2_energy_budget.sh builds one library per mask into abl/ and
// measures time, socket power and clock of each; RESULTS ARE INVALID for every non-zero mask).  Default 0: nothing of
// this is compiled into the shipped kernel.
//   1 no unpack (raw words bit-cast)      2 FIR window from registers, no ds_read_b128      4 one tap per block instead of a
//   tap window (no scalar tap traffic)    8 no staging ds_write_b128     16 output stores without the LDS transposes
//   32 no output stores, no transposes    64 no resampler FMAs           128 no input loads
//   256 no workgroup barriers             512 no FIR FMAs
#ifndef CLHIP_RX_ABL
#define CLHIP_RX_ABL 0
#endif
// Wave priority by phase: a wave that is moving data (staging, issuing the next tile's loads, second stage, transposes,
// output stores) issues ahead of the waves of its SIMD that are inside their FIR, so that memory traffic starts as early
// as it can and flows while others compute.  2 (default): all of those phases at priority 3, the FIR at 0; 1: staging +
// load issue only; 0: off.  Measured on one box: 0.932-0.937 / 0.935 / 0.917-0.918 ms for 0 / 1 / 2.
#ifndef CLHIP_RX_PRIO
#define CLHIP_RX_PRIO 2
#endif
// dst[lane] = src[lane + 1] across the whole wave (DPP wave_shl:1, ctrl 0x130); lane 63 keeps its own value
__device__ __forceinline__ float clhip_wave_shl1(float v)
{
    const int i = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, 0x130, 0xF, 0xF, false));
}

#define MODE_IQ 0
#define MODE_FM 1

constexpr int cgcd(int a, int b) { return b == 0 ? a : cgcd(b, a % b); }
constexpr int clcm(int a, int b) { return a / cgcd(a, b) * b; }
constexpr int round_up(int a, int b) { return (a + b - 1) / b * b; }

#define PIPE_MAX_FIR 128
#define PIPE_MAX_RS 40

struct PipeArgs {
    const void *in;          // stream s at in + s*in_stride elements of in_kind
    long in_stride;
    const f32x2 *hist_in;    // [n_streams][halo] pre-FIR samples (CF32 scale) preceding `in`
    void *out;               // f32x2 (MODE_IQ) or float (MODE_FM)
    long out_stride;
    long n_in;               // new input samples per stream
    long n_out;              // outputs per stream for this call
    int in_kind;             // CL_PIPE_IN_*
    int n_streams;
    int n_int;               // tiles 1 .. n_int-1 of every stream are interior (no bounds checks)
    int n_edge;              // edge tiles per stream: tile 0 and tiles n_int .. (checked path)
    int grid_int;            // workgroups [0, grid_int) are persistent interior workers, the rest edge workers
    int halo;                // history samples per stream
    f32x2 *hist_out;         // [n_streams][halo] history for the NEXT call (ping-pong with hist_in)
    unsigned long long *diag; // diagnostic build only: per-wave cycle sums per phase [grid*4][8]
    int channel;             // CL_CHANNEL_*
    float in_scale;          // 4096 for integer inputs (taps carry 1/4096), 1 for CF32
    const float *fir;        // T taps, pre-multiplied by 1/in_scale (device, read-only)
    const float *rs;         // resampler prototype taps (device, read-only)
    // optional device-side sync validation of raw-word input: the fused path is
    // only valid for chunks whose sync offset is 0 (caribou_smi.c:235-292)
    const int32_t *chunk_offs;   // [n_streams][chunks_per_stream] from clhip_smi_find_offsets, or NULL
    int chunk_shift;             // log2(samples per chunk): chunks are a power of two (native = 2^17)
    long chunks_per_stream;
    int32_t *bad_flag;           // set to 1 when a needed chunk has offs != 0 (tile writes nothing)
    // dynamic tile queue of the persistent interior workers: queue[0] = tickets drawn.  Two counters take turns between
    // the launches of a pipe; a launch's first worker zeroes the one the next launch will use (an exit count that the last
    // worker out reset had every worker of the grid end on a same-address atomic: ~12 ns each, one after the other)
    unsigned int *queue;         // base of the workers' idle words (RX_QUEUE_IDLE ...)
    unsigned int *tickets;       // this launch's ticket counter
    unsigned int *queue_next;    // the counter the pipe's NEXT launch will draw from: zeroed by this launch's first worker
    int queue_k;                 // items per grab; 0 = static striding
#if CLHIP_RX_BOUNDS
    const void *b_in_lo, *b_in_hi, *b_out_lo, *b_out_hi;       // extent of the call's input and output buffers (diagnostic build)
#endif
};

// Diagnostic build (-DCLHIP_RX_BOUNDS=1, tools/oob_bounds_check.py): every global access of the fused kernel is compared with
// the extent of the call's buffers; an access outside is counted, its site and address recorded, and NOT performed.  Nothing of
// this is compiled into the shipped kernel.
#ifndef CLHIP_RX_BOUNDS
#define CLHIP_RX_BOUNDS 0
#endif
#if CLHIP_RX_BOUNDS
__device__ unsigned long long g_rxb[16];          // [0] violations, [1] first site, [2] first address, [3] its lower bound, [4] upper bound, [8 + site] per-site counts
__device__ __forceinline__ bool rxb_ok(const PipeArgs &a, int site, const void *p, size_t bytes, const void *lo, const void *hi)
{
    const unsigned char *q = (const unsigned char *)p;
    if (q >= (const unsigned char *)lo && q + bytes <= (const unsigned char *)hi) return true;
    if (atomicAdd(&g_rxb[0], 1ull) == 0) { g_rxb[1] = (unsigned long long)site; g_rxb[2] = (unsigned long long)q; g_rxb[3] = (unsigned long long)lo; g_rxb[4] = (unsigned long long)hi; }
    atomicAdd(&g_rxb[8 + (site & 7)], 1ull);
    return false;
}
#define RXB_IN(a, site, p, bytes) rxb_ok(a, site, p, bytes, (a).b_in_lo, (a).b_in_hi)
#define RXB_OUT(a, site, p, bytes) rxb_ok(a, site, p, bytes, (a).b_out_lo, (a).b_out_hi)
extern "C" int clhip_rx_debug_bounds(unsigned long long *h16)
{
    return hipMemcpyFromSymbol(h16, HIP_SYMBOL(g_rxb), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
extern "C" int clhip_rx_debug_bounds_reset(void)
{
    unsigned long long z[16] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_rxb), z, sizeof z) == hipSuccess ? 0 : -1;
}
#else
#define RXB_IN(a, site, p, bytes) true
#define RXB_OUT(a, site, p, bytes) true
#endif

template <int T_, int L_, int M_, int KP_, int MODE_, int R_, int NT_, bool FFA_ = false, bool PK_ = true>
struct PipeCfg {
    static constexpr bool PK = PK_;
    static constexpr bool FFA = FFA_;      // 2-parallel fast FIR: 3 half-length sub-filters instead of 4
    static constexpr int T = T_, L = L_, M = M_, KP = KP_, MODE = MODE_, R = R_, NT = NT_;
    static constexpr bool RESAMP = !(L == 1 && M == 1);
    static constexpr int HF = MODE == MODE_FM ? 1 : (RESAMP ? KP - 1 : 0);   // FIR outputs of history
    static constexpr int NFIR = NT * R;             // FIR outputs per tile
    // FIR outputs recomputed per tile as the second stage's history.  At least HF, a multiple of lcm(4, M) -- and, above
    // all, such that a tile's NEW outputs fill whole 128-byte lines (and its new inputs too): tile k's outputs start at
    // k x (NFIR - HFA) x L / M elements, and when that is not a multiple of a line, every 1 KiB store instruction of three
    // tiles in four straddles lines -- 7 whole ones and 2 partial ones -- which the memory system takes 30 % longer to
    // write (tools/microbench/write_shape.hip: 0.80 -> 1.06 ms for config 2's traffic with the output base 32 bytes off
    // a line; the kernel with every arithmetic instruction removed took the same 0.92 ms as the whole kernel).  Config 2:
    // 32 instead of 8 (two lanes of the tile's 256 recompute history: 0.6 % more arithmetic).
    static constexpr int pick_hfa()
    {
        if (HF == 0) return 0;
        constexpr int step = clcm(4, M), ob = MODE == MODE_FM ? 4 : 8;
        const int least = round_up(HF, step);
#ifndef CLHIP_RX_UNALIGNED_TILES
        for (int h = least; h <= NFIR / 8; h += step) {
            const long in = NFIR - h, out = MODE == MODE_FM ? in : in * L / M;
            if ((in * L) % M == 0 && (out * ob) % 128 == 0 && (in * 4) % 128 == 0) return h;
        }
#endif
        return least;
    }
    static constexpr int HFA = pick_hfa();
    static constexpr int TILE_IN = NFIR - HFA;      // new inputs per tile
    // FFA: y[0] of a lane needs B[-1] = the previous lane's B[R/2-1].  When the tile's very first FIR output
    // is never consumed (HFA > HF) it can come by shuffle instead of being recomputed by every lane.
    static constexpr bool BSHUF = FFA_ && (HFA > HF);
    static constexpr int HALO = T + HFA;            // local 0 <-> global S - HALO
    static constexpr int NLOAD = NFIR + T;          // staged samples
    static constexpr int NOUT = MODE == MODE_FM ? R : R * L / M;             // outputs per lane
    static constexpr int SKIP0 = MODE == MODE_FM ? HFA : HFA * L / M;        // lane 0's history-only outputs
    static constexpr int TSTRIDE = (R * 8 + 16);    // bytes between lanes' windows in LDS
    static constexpr int IN_BYTES = (NLOAD * 8 + (NLOAD / R + 1) * 16 + 63) / 64 * 64;
    static constexpr int TAIL_BYTES = (NT / 64) * 8 * 8;   // one 8-sample tail slot per wave
    static constexpr int LDS_BYTES = IN_BYTES + TAIL_BYTES + 16;   // + the workgroup's queue slot
    static_assert(T % 4 == 0 && R % 4 == 0, "T and R must be multiples of 4");
    static_assert((R * L) % M == 0 && (HFA * L) % M == 0, "lane outputs must be integral");
    static_assert(HF <= 8 && HF <= R, "history too long for the tail exchange");
    static_assert(MODE == MODE_FM || (NOUT % 2 == 0 && SKIP0 % 2 == 0), "16-byte stores of float2 pairs");
    static_assert(MODE != MODE_FM || (NOUT % 4 == 0 && SKIP0 % 4 == 0), "16-byte stores of 4 floats");
};

// LDS byte offset of staged sample j: 8 B per (I,Q) pair plus a 16-byte pad
// after every R samples, so that lane windows start 8R+16 bytes apart and the
// 16 lanes a ds_read_b128 services together hit 64 distinct banks.
template <int R> __device__ __forceinline__ int lds_off(int j) { return j * 8 + (j / R) * 16; }

__device__ __forceinline__ f32x2 load_sample(const PipeArgs &a, const void *base, long g)
{
    // one pre-FIR sample in the LDS domain (unscaled integers for integer inputs)
    if (a.in_kind != CL_PIPE_IN_CF32 ? !RXB_IN(a, 1, (const uint32_t *)base + g, 4) : !RXB_IN(a, 1, (const f32x2 *)base + g, 8)) { f32x2 z = {0.f, 0.f}; return z; }
    if (a.in_kind == CL_PIPE_IN_SMI_WORDS) {
        const uint32_t w = ((const uint32_t *)base)[g];
        const int fa = clhip_field_a(w), fb = clhip_field_b(w);
        f32x2 r = {(float)(a.channel == CL_CHANNEL_HIF ? fb : fa), (float)(a.channel == CL_CHANNEL_HIF ? fa : fb)};
        return r;
    } else if (a.in_kind == CL_PIPE_IN_CS16) {
        const uint32_t w = ((const uint32_t *)base)[g];
        f32x2 r = {(float)(int16_t)(w & 0xFFFF), (float)(int16_t)(w >> 16)};
        return r;
    } else {
        return ((const f32x2 *)base)[g];
    }
}

#if (CLHIP_RX_ABL & 256)
#define RX_BARRIER() __builtin_amdgcn_wave_barrier()
#else
#define RX_BARRIER() __syncthreads()
#endif

// acc += x * tap on an (I,Q) pair: one v_pk_fma_f32 (PK) or two v_fmac_f32.
template <bool PK>
__device__ __forceinline__ void fma2(f32x2 &acc, const f32x2 x, const float tap)
{
    if constexpr (PK) acc += x * tap;
    else {
        acc.x = __builtin_fmaf(x.x, tap, acc.x);
        acc.y = __builtin_fmaf(x.y, tap, acc.y);
    }
}

// One 16-byte group = 4 samples.  KIND and the channel type are compile-time here so that the
// per-sample cost is exactly v_bfe_i32 x2 + v_cvt_f32_i32 x2 (no per-sample selects).
template <int KIND, bool HIF>
__device__ __forceinline__ void convert4(const u32x4 w, f32x2 (&v)[4])
{
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if constexpr ((CLHIP_RX_ABL & 1) != 0) {
            v[k].x = __builtin_bit_cast(float, w[k] & 0x3fffffffu); v[k].y = v[k].x;
        } else if constexpr (KIND == CL_PIPE_IN_SMI_WORDS) {
            const int fa = clhip_field_a(w[k]), fb = clhip_field_b(w[k]);
            v[k].x = (float)(HIF ? fb : fa);        // caribou_smi.c:342-378
            v[k].y = (float)(HIF ? fa : fb);
        } else {
            v[k].x = (float)(int16_t)(w[k] & 0xFFFF);
            v[k].y = (float)(int16_t)(w[k] >> 16);
        }
    }
}

template <int R>
__device__ __forceinline__ void lds_put4(unsigned char *lds, int j, const f32x2 (&v)[4])
{
    unsigned char *d = lds + lds_off<R>(j);
    f32x4 q0 = {v[0].x, v[0].y, v[1].x, v[1].y}, q1 = {v[2].x, v[2].y, v[3].x, v[3].y};
    *(f32x4 *)d = q0;
    *(f32x4 *)(d + 16) = q1;
}

// Register image of one interior tile's raw input: IT 16-byte groups per lane
// (integer kinds) or 2*IT (CF32).  Loads are issued here and consumed one tile
// later, so HBM latency hides under the previous tile's FIR.
template <class C, int KIND>
struct TileRegs {
    static constexpr int NG = C::NLOAD / 4;
    static constexpr int IT = (NG + C::NT - 1) / C::NT;
    static constexpr int NV = KIND == CL_PIPE_IN_CF32 ? 2 * IT : IT;
    u32x4 w[NV];
};

template <class C, int KIND>
__device__ __forceinline__ void tile_issue_loads(const PipeArgs &a, TileRegs<C, KIND> &r, const void *in, long g0, int t)
{
    constexpr int NG = TileRegs<C, KIND>::NG, IT = TileRegs<C, KIND>::IT;
#pragma unroll
    for (int it = 0; it < IT; it++) {
        const int grp = t + it * C::NT;
        if (NG % C::NT == 0 || it < IT - 1 || grp < NG) {
            if constexpr (KIND == CL_PIPE_IN_CF32) {
                if (RXB_IN(a, 2, (const f32x2 *)in + g0 + 4 * grp, 32)) {
                r.w[2 * it] = *(const u32x4 *)((const f32x2 *)in + g0 + 4 * grp);
                r.w[2 * it + 1] = *(const u32x4 *)((const f32x2 *)in + g0 + 4 * grp + 2);
                }
            } else if constexpr ((CLHIP_RX_ABL & 128) != 0) {
                r.w[it] = u32x4{(uint32_t)grp, (uint32_t)g0, 3u, 4u};
            } else {
                if (RXB_IN(a, 2, (const uint32_t *)in + g0 + 4 * grp, 16))
                r.w[it] = __builtin_nontemporal_load((const u32x4 *)((const uint32_t *)in + g0 + 4 * grp));
            }
        }
    }
}

template <class C, int KIND, bool HIF>
__device__ __forceinline__ void tile_regs_to_lds(const TileRegs<C, KIND> &r, unsigned char *lds, int t)
{
    constexpr int NG = TileRegs<C, KIND>::NG, IT = TileRegs<C, KIND>::IT, R = C::R, NT = C::NT;
    // group grp = t + it*NT holds samples 4*grp..: lds_off = 32*grp + 16*(4*grp/R).  With NT*4 a
    // multiple of R the per-iteration stride is a compile-time constant: one lane base, immediates after.
    static_assert((NT * 4) % R == 0, "staging stride must be a whole number of padded rows");
    constexpr int STRIDE = NT * 32 + (NT * 4 / R) * 16;
    unsigned char *base = lds + 32 * t + 16 * ((4 * t) / R);
#pragma unroll
    for (int it = 0; it < IT; it++) {
        const int grp = t + it * NT;
        if (NG % NT == 0 || it < IT - 1 || grp < NG) {
            unsigned char *d = base + it * STRIDE;
            if constexpr (KIND == CL_PIPE_IN_CF32) {
                *(u32x4 *)d = r.w[2 * it];
                *(u32x4 *)(d + 16) = r.w[2 * it + 1];
            } else {
                f32x2 v[4];
                convert4<KIND, HIF>(r.w[it], v);
                f32x4 q0 = {v[0].x, v[0].y, v[1].x, v[1].y}, q1 = {v[2].x, v[2].y, v[3].x, v[3].y};
                if constexpr ((CLHIP_RX_ABL & 8) != 0) {
                    asm volatile("" :: "v"(q0), "v"(q1));                    // converted, not written
                } else {
                    *(f32x4 *)d = q0;
                    *(f32x4 *)(d + 16) = q1;
                }
            }
        }
    }
}

// Edge tiles (stream start: history; stream end: zero fill): checked, un-prefetched path.
template <class C>
__device__ __forceinline__ void stage_tile_slow(const PipeArgs &a, const void *in, const f32x2 *hist, long S,
                                             unsigned char *lds, int t)
{
    const long g0 = S - C::HALO;
    for (int j = t * 4; j < C::NLOAD; j += C::NT * 4) {
        const long g = g0 + j;
        f32x2 v[4];
        if (g < 0) {                       // HALO % 4 == 0: the whole group is history
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = hist[C::HALO + g + k] * a.in_scale;
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                f32x2 z = {0.f, 0.f};
                v[k] = (g + k < a.n_in) ? load_sample(a, in, g + k) : z;
            }
        }
        lds_put4<C::R>(lds, j, v);
    }
}

// ---------------------------------------------------------------------------
// compute phases shared by the interior (persistent) and edge kernels
// ---------------------------------------------------------------------------

// FIR: R outputs per lane from a sliding window.  Lane window w = 0..T+R-1 is staged sample
// R*t + w; output r (FIR index R*t + r of the tile) uses tap k = T + r - w of window sample w.
// The window is walked in blocks of R samples: block b pairs sample j of the block with tap
// (T - R*b) + (r - j), so each block needs 2R-1 consecutive taps (scalar loads -> SGPRs) and R
// samples (one ds_read_b128 per pair).  Every accumulator sees its taps in descending k order.
template <class C>
__device__ __forceinline__ void fir_tile(const unsigned char *lds, int t, const float *fir, f32x2 (&acc)[C::R])
{
    constexpr int T = C::T, R = C::R;
    static_assert(T % R == 0, "window is walked in blocks of R samples");
#pragma unroll
    for (int r = 0; r < R; r++) { acc[r].x = 0.f; acc[r].y = 0.f; }
    const unsigned char *win = lds + t * C::TSTRIDE;
    // taps are read-only for the whole launch: the constant address space makes every uniform tap
    // load a scalar (SMEM) load into SGPRs, never a vector load into VGPRs
    const cfloat_t *__restrict__ h = (const cfloat_t *)fir;
    f32x2 x[R];
#define LOAD_BLOCK(B)                                                              \
    _Pragma("unroll") for (int j = 0; j < R; j += 2) {                             \
        const f32x4 xx = *(const f32x4 *)(win + (B) * C::TSTRIDE + j * 8);         \
        x[j] = xx.xy; x[j + 1] = xx.zw;                                            \
    }
    {   // first block: k = T + r - j < T  <=>  r < j
        LOAD_BLOCK(0)
        float tp[R];
#pragma unroll
        for (int i = 1; i < R; i++) tp[i] = h[T - i];            // tp[i] = h[T - i]
#pragma unroll
        for (int j = 1; j < R; j++)
#pragma unroll
            for (int r = 0; r < j; r++) fma2<C::PK>(acc[r], x[j], tp[j - r]);
    }
#pragma unroll 1
    for (int b = 1; b < T / R; b++) {
        LOAD_BLOCK(b)
        const cfloat_t *__restrict__ hb = h + (T - R * b);          // k = hb index (r - j) in [-(R-1), R-1]
        float tp[2 * R - 1];
#pragma unroll
        for (int i = 0; i < 2 * R - 1; i++) tp[i] = hb[i - (R - 1)];
#pragma unroll
        for (int j = 0; j < R; j++)
#pragma unroll
            for (int r = 0; r < R; r++) fma2<C::PK>(acc[r], x[j], tp[(R - 1) + r - j]);
    }
    {   // last block: k = r - j >= 0
        LOAD_BLOCK(T / R)
        float tp[R];
#pragma unroll
        for (int i = 0; i < R; i++) tp[i] = h[i];
#pragma unroll
        for (int j = 0; j < R; j++)
#pragma unroll
            for (int r = j; r < R; r++) fma2<C::PK>(acc[r], x[j], tp[r - j]);
    }
#undef LOAD_BLOCK
}

// 2-parallel fast FIR (FFA).  Split window, taps and outputs by parity:
//   X0[j] = x[2j], X1[j] = x[2j+1];  H0[v] = h[2v], H1[v] = h[2v+1], HS = H0 + H1 (host, fp32)
//   A[u] = sum_v H0[v] X0[T/2+u-v]   B[u] = sum_v H1[v] X1[T/2+u-v]   C[u] = sum_v HS[v] (X0+X1)[T/2+u-v]
//   y[2u] = A[u] + B[u-1]            y[2u+1] = C[u] - A[u] - B[u]
// Three (T/2)-tap sub-filters for R/2 outputs each instead of four: 800 packed FMAs + 64 adds per
// lane instead of 1024 at T=64, R=16.  Same results to rounding (not bit-identical to the direct
// form: the summation order differs), well inside the 1e-5 bar.  `ffa` = [H0 | H1 | HS], T/2 each.
template <class C>
__device__ __forceinline__ void fir_tile_ffa(const unsigned char *lds, int t, const float *ffa, f32x2 (&acc)[C::R],
                                             f32x2 &b_last)
{
    constexpr int U0 = C::BSHUF ? 0 : -1;           // first B output computed by this lane
    constexpr int T = C::T, R = C::R, TH = T / 2, RH = R / 2;      // decimated: TH taps, RH outputs, blocks of RH
    static_assert(TH % RH == 0 && R % 4 == 0, "FFA walks the decimated window in blocks of R/2");
    const unsigned char *win = lds + t * C::TSTRIDE;
    const cfloat_t *__restrict__ h0 = (const cfloat_t *)ffa, *__restrict__ h1 = h0 + TH, *__restrict__ hs = h0 + 2 * TH;
    f32x2 A[RH], Bm[RH + 1], Cc[RH];              // Bm[u+1] = B[u], u = -1..RH-1
#pragma unroll
    for (int u = 0; u < RH; u++) { A[u].x = A[u].y = 0.f; Cc[u].x = Cc[u].y = 0.f; }
#pragma unroll
    for (int u = 0; u <= RH; u++) { Bm[u].x = Bm[u].y = 0.f; }
    f32x2 x0[RH], x1[RH];
    // decimated block b holds j = RH*b + jj (window samples 2j, 2j+1 = one ds_read_b128);
    // sample jj meets output u through tap d = (TH - RH*b) + (u - jj)
#if (CLHIP_RX_ABL & 2)
#define LOAD_DBLOCK_LDS(B)                                                                  \
    _Pragma("unroll") for (int jj = 0; jj < RH; jj++) {                                     \
        f32x4 xx = {(float)t, (float)(B), (float)jj, 1.0f};                                 \
        asm volatile("" : "+v"(xx));                                                        \
        x0[jj] = xx.xy; x1[jj] = xx.zw;                                                     \
    }
#else
#define LOAD_DBLOCK_LDS(B)                                                                  \
    _Pragma("unroll") for (int jj = 0; jj < RH; jj++) {                                     \
        const int w = 2 * (RH * (B) + jj);                                                  \
        const f32x4 xx = *(const f32x4 *)(win + w * 8 + (w / R) * 16);                      \
        x0[jj] = xx.xy; x1[jj] = xx.zw;                                                     \
    }
#endif
#if CLHIP_FFA_DPP
    // Window block B of lane t is block 0 of lane t + B (lanes own R consecutive outputs and a block is R samples): after
    // block 0 has been read from LDS, every later block arrives from the next lane by a one-lane wave shift
    // (v_mov_b32_dpp wave_shl:1) of the registers that hold the previous block; only the lanes whose source lies
    // in the next wave (the last B lanes) read LDS.  32 moves replace 8 ds_read_b128 per block.
    const int lane_w = t & 63;
#define LOAD_DBLOCK(B)                                                                      \
    if ((B) == 0) { LOAD_DBLOCK_LDS(0) }                                                    \
    else {                                                                                  \
        _Pragma("unroll") for (int jj = 0; jj < RH; jj++) {                                 \
            x0[jj].x = clhip_wave_shl1(x0[jj].x); x0[jj].y = clhip_wave_shl1(x0[jj].y);     \
            x1[jj].x = clhip_wave_shl1(x1[jj].x); x1[jj].y = clhip_wave_shl1(x1[jj].y);     \
        }                                                                                   \
        if (lane_w >= 64 - (B)) { LOAD_DBLOCK_LDS(B) }                                      \
    }
#else
#define LOAD_DBLOCK(B) LOAD_DBLOCK_LDS(B)
#endif
#define FFA_BLOCK(TAPBASE, LO, HI)                                                          \
    _Pragma("unroll") for (int jj = 0; jj < RH; jj++) {                                     \
        const f32x2 xs = x0[jj] + x1[jj];                                                   \
        _Pragma("unroll") for (int u = U0; u < RH; u++) {                                   \
            const int d = (TAPBASE) + u - jj;         /* relative to the block's tap window */ \
            if (d >= (LO) && d <= (HI)) {                                                   \
                fma2<C::PK>(Bm[u + 1], x1[jj], t1[d - (LO)]);                               \
                if (u >= 0) { fma2<C::PK>(A[u], x0[jj], t0[d - (LO)]); fma2<C::PK>(Cc[u], xs, ts[d - (LO)]); } \
            }                                                                               \
        }                                                                                   \
    }
    {   // first block (b = 0): taps d = TH + u - jj, valid d <= TH-1; window of taps [TH-RH-1 .. TH-1]
        LOAD_DBLOCK(0)
        float t0[RH + 1], t1[RH + 1], ts[RH + 1];
#pragma unroll
        for (int i = 0; i <= RH; i++) { t0[i] = h0[TH - RH - 1 + i]; t1[i] = h1[TH - RH - 1 + i]; ts[i] = hs[TH - RH - 1 + i]; }
        FFA_BLOCK(TH, TH - RH - 1, TH - 1)
    }
#pragma unroll 1
    for (int b = 1; b < TH / RH; b++) {
        LOAD_DBLOCK(b)
        const int base = TH - RH * b;               // d = base + u - jj in [base-RH, base+RH-1]
        float t0[2 * RH], t1[2 * RH], ts[2 * RH];
#pragma unroll
        for (int i = 0; i < 2 * RH; i++) {
            const int ii = (CLHIP_RX_ABL & 4) ? 0 : i;
            t0[i] = h0[base - RH + ii]; t1[i] = h1[base - RH + ii]; ts[i] = hs[base - RH + ii];
        }
#pragma unroll
        for (int jj = 0; jj < RH; jj++) {
            const f32x2 xs = x0[jj] + x1[jj];
#pragma unroll
            for (int u = U0; u < RH; u++) {
                const int i = RH + u - jj;          // d - (base - RH), always in [0, 2RH-1]
                if constexpr ((CLHIP_RX_ABL & 512) != 0) {
                    if (jj == 0) { Bm[u + 1] += x1[0]; if (u >= 0) { A[u] += x0[0]; Cc[u] += xs; } }     // the window is consumed, the products are not formed
                } else {
                    fma2<C::PK>(Bm[u + 1], x1[jj], t1[i]);
                    if (u >= 0) { fma2<C::PK>(A[u], x0[jj], t0[i]); fma2<C::PK>(Cc[u], xs, ts[i]); }
                }
            }
        }
        if constexpr ((CLHIP_RX_ABL & 512) != 0) {
#pragma unroll
            for (int jj = 1; jj < RH; jj++) asm volatile("" :: "v"(x0[jj]), "v"(x1[jj]));
        }
    }
    {   // last block (b = TH/RH): d = u - jj >= 0; taps [0 .. RH-1]
        LOAD_DBLOCK(TH / RH)
        float t0[RH], t1[RH], ts[RH];
#pragma unroll
        for (int i = 0; i < RH; i++) { t0[i] = h0[i]; t1[i] = h1[i]; ts[i] = hs[i]; }
        FFA_BLOCK(0, 0, RH - 1)
    }
#undef LOAD_DBLOCK
#undef LOAD_DBLOCK_LDS
#undef FFA_BLOCK
#pragma unroll
    for (int u = 0; u < RH; u++) {
        acc[2 * u] = A[u] + Bm[u];                  // A[u] + B[u-1]   (u = 0 with BSHUF: B[-1] added later)
        acc[2 * u + 1] = Cc[u] - A[u] - Bm[u + 1];  // C[u] - A[u] - B[u]
    }
    b_last = Bm[RH];
}

// Second stage: polyphase resampler / FM demod / pass-through from registers.  The HF FIR
// outputs before a lane's own come from the previous lane: a one-lane shuffle inside the wave,
// an LDS slot across waves.  Contains the workgroup barrier that retires the staged input tile.
// Result: the lane's NOUT output elements as PL 16-byte pieces.
template <class C>
__device__ __forceinline__ void second_stage(unsigned char *lds, int t, const float *rs, f32x2 (&acc)[C::R],
                                             f32x4 (&pc)[C::NOUT * (C::MODE == MODE_FM ? 4 : 8) / 16], const f32x2 b_last)
{
    constexpr int R = C::R, L = C::L, M = C::M, KP = C::KP, HF = C::HF, NOUT = C::NOUT;
    constexpr int PL = NOUT * (C::MODE == MODE_FM ? 4 : 8) / 16;
    const int lane = t & 63, wave = t >> 6;
    f32x2 yh[HF > 0 ? HF : 1];
    if constexpr (HF > 0) {
        f32x2 *tslot = (f32x2 *)(lds + C::IN_BYTES);        // [waves][8], past the input tile
        if (lane == 63) {
#pragma unroll
            for (int i = 0; i < HF; i++) tslot[wave * 8 + i] = acc[R - HF + i];
            if constexpr (C::BSHUF) tslot[wave * 8 + 7] = b_last;      // HF <= 7: slot 7 is free
        }
    }
    RX_BARRIER();          // FIR reads of the staged tile are done (its LDS is reused below); tail slots visible
    if constexpr (HF > 0) {
        const f32x2 *tslot = (const f32x2 *)(lds + C::IN_BYTES);
#pragma unroll
        for (int i = 0; i < HF; i++) {
            const f32x2 v = acc[R - HF + i];
            yh[i].x = __shfl_up(v.x, 1, 64);
            yh[i].y = __shfl_up(v.y, 1, 64);
            if (lane == 0) yh[i] = tslot[(wave > 0 ? wave - 1 : 0) * 8 + i];   // wave 0 / lane 0: history-only outputs
        }
    }
    if constexpr (C::BSHUF) {
        // y[0] += B[-1]: the previous lane's last B output (the tile's first lane never uses its y[0])
        static_assert(HF > 0 && HF <= 7, "B hand-off rides in the spare tail slot");
        f32x2 bp;
        bp.x = __shfl_up(b_last.x, 1, 64);
        bp.y = __shfl_up(b_last.y, 1, 64);
        if (lane == 0) bp = ((const f32x2 *)(lds + C::IN_BYTES))[(wave > 0 ? wave - 1 : 0) * 8 + 7];
        acc[0] += bp;
    }
#define YY(i) ((i) < 0 ? yh[HF + (i)] : acc[(i)])
    if constexpr (C::MODE == MODE_IQ) {
        f32x2 o[NOUT];
        if constexpr (C::RESAMP) {
            float rsv[KP * L];
#pragma unroll
            for (int i = 0; i < KP * L; i++) rsv[i] = ((const cfloat_t *)rs)[i];
#pragma unroll
            for (int m = 0; m < NOUT; m++) {
                const int tp = m * M, b = tp / L, p = tp % L;
                f32x2 sacc = {0.f, 0.f};
#pragma unroll
                for (int i = 0; i < ((CLHIP_RX_ABL & 64) ? 1 : KP); i++) fma2<C::PK>(sacc, YY(b - i), rsv[p + i * L]);
                o[m] = sacc;
            }
        } else {
#pragma unroll
            for (int m = 0; m < NOUT; m++) o[m] = acc[m];
        }
#pragma unroll
        for (int k = 0; k < PL; k++) { pc[k].x = o[2 * k].x; pc[k].y = o[2 * k].y; pc[k].z = o[2 * k + 1].x; pc[k].w = o[2 * k + 1].y; }
    } else {
        // FM phase-difference demod: atan2(Im z, Re z), z = y[n] conj(y[n-1])
        float o[R];
#pragma unroll
        for (int i = 0; i < R; i++) {
            const f32x2 c = YY(i), p = YY(i - 1);
            o[i] = clhip_atan2f(c.y * p.x - c.x * p.y, c.x * p.x + c.y * p.y);
        }
#pragma unroll
        for (int k = 0; k < PL; k++) { pc[k].x = o[4 * k]; pc[k].y = o[4 * k + 1]; pc[k].z = o[4 * k + 2]; pc[k].w = o[4 * k + 3]; }
    }
#undef YY
}

// Coalesced store.  A lane's LB output bytes are contiguous but lane-strided stores would hand
// the memory system 64 separate 16-byte pieces per instruction.  Each wave therefore transposes
// through its own slice of the (now dead) input tile, half a wave at a time: 32 lanes write
// their pieces (LB+16 B pitch: conflict-free), then all 64 lanes read consecutive pieces and
// store 1 KiB-contiguous runs.  Wave-private LDS, in-order DS pipe: no workgroup barrier.
// CHECKED = per-element bounds [lo, hi) (edge tiles); interior tiles only mask the history-only
// outputs of the tile's first lane (e < lo).
template <class C, bool CHECKED>
__device__ __forceinline__ void store_tile(const PipeArgs &a, unsigned char *lds, int t, unsigned char *outb, long tile_e0, long lo, long hi,
                                           const f32x4 (&pc)[C::NOUT * (C::MODE == MODE_FM ? 4 : 8) / 16])
{
    constexpr int NOUT = C::NOUT, OB = C::MODE == MODE_FM ? 4 : 8, LB = NOUT * OB, PL = LB / 16;
    constexpr int EPP = 16 / OB;                         // elements per piece
    constexpr int PITCH = LB + 16;
    constexpr int HALF_PIECES = 32 * PL;                 // pieces per half wave
    constexpr int NJ = (HALF_PIECES + 63) / 64;
    static_assert(32 * PITCH * (C::NT / 64) <= C::IN_BYTES, "per-wave transpose slices must fit the input tile");
    const int lane = t & 63, wave = t >> 6;
    if constexpr ((CLHIP_RX_ABL & 32) != 0) {
#pragma unroll
        for (int k = 0; k < PL; k++) asm volatile("" :: "v"(pc[k]));
        return;
    }
    if constexpr ((CLHIP_RX_ABL & 16) != 0 && !CHECKED) {        // (interior tiles only: edge tiles keep their bounds checks)
        // the same 1 KiB-contiguous store instructions, fed from the lane's own pieces (wrong data in the right places)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            unsigned char *hb = outb + (tile_e0 + (long)NOUT * (wave * 64 + h * 32)) * OB + lane * 16;
#pragma unroll
            for (int j = 0; j < NJ; j++) __builtin_nontemporal_store(pc[(h * NJ + j) % PL], (f32x4 *)(hb + j * 1024));
        }
        return;
    }
    unsigned char *scr = lds + wave * (32 * PITCH);
    // piece p = lane + 64 j sits at row p / PL, column p % PL of the scratch: divide once, then step
    const int row0 = lane / PL, col0 = lane % PL;
    const unsigned char *rd0 = scr + row0 * PITCH + col0 * 16;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        if ((lane >> 5) == h) {
            unsigned char *w = scr + (lane & 31) * PITCH;
#pragma unroll
            for (int k = 0; k < PL; k++) *(f32x4 *)(w + 16 * k) = pc[k];
        }
        __builtin_amdgcn_wave_barrier();
        const long half_e0 = tile_e0 + (long)NOUT * (wave * 64 + h * 32);      // first element of this half wave
        unsigned char *hb = outb + half_e0 * OB + lane * 16;
        f32x4 v[NJ];
        int col = col0;
        const unsigned char *rd = rd0;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            if (HALF_PIECES % 64 == 0 || lane + 64 * j < HALF_PIECES) v[j] = *(const f32x4 *)rd;
            // advance 64 pieces: 64 = (64 / PL) rows + (64 % PL) columns, with carry
            col += 64 % PL;
            rd += (64 / PL) * PITCH + (64 % PL) * 16;
            if (col >= PL) { col -= PL; rd += PITCH - PL * 16; }
        }
        const bool plain = !CHECKED && (wave | h) != 0;      // wave-uniform: nothing to mask in this half
        if (plain) {
#pragma unroll
            for (int j = 0; j < NJ; j++)
                if (HALF_PIECES % 64 == 0 || lane + 64 * j < HALF_PIECES)
                    if (RXB_OUT(a, 4, hb + j * 1024, 16))
                    __builtin_nontemporal_store(v[j], (f32x4 *)(hb + j * 1024));     // streamed out, never read back here
        } else {
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                const int p = lane + 64 * j;
                if (HALF_PIECES % 64 == 0 || p < HALF_PIECES) {
                    const long e = half_e0 + (long)p * EPP;
                    if (e >= lo && (!CHECKED || e + EPP <= hi)) {
                        if (RXB_OUT(a, 5, hb + j * 1024, 16))
                        *(f32x4 *)(hb + j * 1024) = v[j];
                    } else if (CHECKED) {
#pragma unroll
                        for (int k = 0; k < EPP; k++) {
                            if (e + k >= lo && e + k < hi) {
                                if (!RXB_OUT(a, 6, hb + j * 1024 + k * OB, OB)) continue;
                                if constexpr (OB == 8) { f32x2 q = {v[j][2 * k], v[j][2 * k + 1]}; *(f32x2 *)(hb + j * 1024 + k * OB) = q; }
                                else *(float *)(hb + j * 1024 + k * OB) = v[j][k];
                            }
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// wave-uniform: do the chunks this tile reads all have sync offset 0?  (device-side validation)
// Two sources: the per-chunk results of clhip_smi_find_offsets (a.chunk_offs), or -- a.chunk_offs == NULL with the
// check armed -- the chunk's own first four words: caribou_smi_find_buffer_offset (caribou_smi.c:235-292) returns 0
// exactly when the words at byte offsets 0, 4, 8, 12 all carry the sync pattern (0 is the smallest candidate), or when
// the chunk is at most 16 bytes long.  Four scalar loads per tile replace a kernel launch per call.
template <class C>
__device__ __forceinline__ bool tile_sync_bad(const PipeArgs &a, int s, long S)
{
    bool bad = false;
    if (a.bad_flag) {
        const long first = S - C::HALO > 0 ? S - C::HALO : 0;
        const long last = (S + C::TILE_IN < a.n_in ? S + C::TILE_IN : a.n_in) - 1;
        if (a.chunk_offs) {
            // written by clhip_smi_find_offsets before this launch, read-only here: scalar loads (SMEM).  A vector
            // load would put an s_waitcnt vmcnt(0) at the top of every tile and drain the previous tile's stores.
            typedef __attribute__((address_space(4))) int32_t cint_t;
            const cint_t *o = (const cint_t *)(a.chunk_offs + (long)s * a.chunks_per_stream);
            for (int c = (int)(first >> a.chunk_shift); c <= (int)(last >> a.chunk_shift); c++) bad |= o[c] != 0;
        } else {
            typedef __attribute__((address_space(4))) uint32_t cu32_t;
            const cu32_t *w = (const cu32_t *)((const uint32_t *)a.in + (long)s * a.in_stride);
            for (long c = first >> a.chunk_shift; c <= (last >> a.chunk_shift); c++) {
                const long c0 = c << a.chunk_shift;
                if (a.n_in - c0 > 4 && RXB_IN(a, 3, (const uint32_t *)a.in + (long)s * a.in_stride + c0, 16)) {     // len <= 16 bytes: offset 0 by definition (:249-252)
                    const uint32_t m = (w[c0] & w[c0 + 1] & w[c0 + 2] & w[c0 + 3]) & 0xC001C000u;
                    const uint32_t z = (w[c0] | w[c0 + 1] | w[c0 + 2] | w[c0 + 3]) & 0xC001C000u;
                    bad |= m != 0x80004000u || z != 0x80004000u;    // every word: (w & 0xC001C000) == 0x80004000
                }
            }
        }
        // (the flag may live in mapped host memory -- clhip_rx_pipe_run_smi reads it without a copy -- hence a store that
        // needs no PCIe atomics; every writer stores the same 1)
        if (bad && threadIdx.x == 0) __hip_atomic_store(a.bad_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // the tile writes nothing
    }
    return bad;
}

// ---------------------------------------------------------------------------
// edge worker: tile 0 of a stream (history) or a tile at the stream end (zero fill, partial
// outputs): checked staging and checked stores.  The worker of tile 0 also writes the stream's
// history for the next call.  Runs in the same launch as the interior workers (extra workgroups
// at the end of the grid), so a call is ONE kernel.
// ---------------------------------------------------------------------------
template <class C>
__device__ __forceinline__ void rx_pipe_edge_worker(const PipeArgs &a, unsigned char *lds, int e)
{
    constexpr int OB = C::MODE == MODE_FM ? 4 : 8, PL = C::NOUT * OB / 16;
    const int t = threadIdx.x;
    const int s = e / a.n_edge, ei = e % a.n_edge;
    const int tile = ei == 0 ? 0 : a.n_int + ei - 1;
    const long S = (long)tile * C::TILE_IN;
    const void *in = a.in_kind == CL_PIPE_IN_CF32 ? (const void *)((const f32x2 *)a.in + (long)s * a.in_stride)
                                                  : (const void *)((const uint32_t *)a.in + (long)s * a.in_stride);
    const bool bad = tile_sync_bad<C>(a, s, S);
    stage_tile_slow<C>(a, in, a.hist_in + (long)s * C::HALO, S, lds, t);
    __syncthreads();
    f32x2 acc[C::R];
    f32x4 pc[PL];
    f32x2 b_last = {0.f, 0.f};
    if constexpr (C::FFA) fir_tile_ffa<C>(lds, t, a.fir, acc, b_last); else fir_tile<C>(lds, t, a.fir, acc);
    second_stage<C>(lds, t, a.rs, acc, pc, b_last);
    const long tile_e0 = C::MODE == MODE_FM ? (S - C::HFA) : (S - C::HFA) / C::M * C::L;
    const long lo = C::MODE == MODE_FM ? S : S / C::M * C::L;
    store_tile<C, true>(a, lds, t, (unsigned char *)a.out + (long)s * a.out_stride * OB, tile_e0, lo, bad ? 0 : a.n_out, pc);
    if (ei == 0) pipe_update_hist(a, s, t, C::NT);
}

// ---------------------------------------------------------------------------
// interior kernel: persistent workgroups over the tiles that lie fully inside a stream
// (tile index 1 .. n_int-1).  No bounds checks; the raw words of the NEXT item are loaded
// into registers while the current one computes, so HBM latency hides under the FIR.
// One kernel per (config, input kind, channel type): each gets its own register allocation.
// ---------------------------------------------------------------------------
// queue[0] = items handed out, queue[1] = workers that have left; from word RX_QUEUE_IDLE on, one idle word per worker
// (16 words apart) for the tiles on which a worker takes nothing from the queue (see the kernel)
#define RX_QUEUE_IDLE 64
#define RX_QUEUE_MAX_WORKERS 4096
#define RX_QUEUE_WORDS (RX_QUEUE_IDLE + 16 * RX_QUEUE_MAX_WORKERS)
// One grab of the tile queue, whose result is not needed before the next tile.  The counter's address goes through a
// register the compiler cannot see through: with a uniform address its atomic optimizer rewrites the add as a wave
// reduction whose result it broadcasts (v_readfirstlane) -- and waits for, with s_waitcnt vmcnt(0), right where the
// atomic is issued: behind the previous tile's twelve output stores, whose full drain time every second tile then
// stood in front of the next tile's loads.
__device__ __forceinline__ unsigned int rx_queue_grab(unsigned int *queue, unsigned int by = 1u)
{
    uintptr_t qa = (uintptr_t)queue;
    asm volatile("" : "+v"(qa));
    return __hip_atomic_fetch_add((__attribute__((address_space(1))) unsigned int *)qa, by, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// s_memtime stamp (diagnostic build only; the shipped kernels execute none)
#define DIAG_STAMP(T) do { if constexpr (DIAG) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(T) :: "memory"); __builtin_amdgcn_sched_barrier(0); } } while (0)

template <class C, int KIND, bool HIF, bool DIAG = false>
__global__ __launch_bounds__(C::NT, KIND == CL_PIPE_IN_CF32 ? 3 : 4)      // integer inputs: 4 waves/SIMD (<= 128 VGPRs)
void rx_pipe_fused_kernel(const PipeArgs a)
{
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, ts5 = 0, ts6 = 0;
    unsigned long long d_stage = 0, d_bar0 = 0, d_fir = 0, d_second = 0, d_store = 0, d_bar2 = 0, d_tiles = 0;
    unsigned long long k_t0 = 0, k_r0 = 0;
    if constexpr (DIAG) { k_t0 = __builtin_amdgcn_s_memtime(); k_r0 = __builtin_amdgcn_s_memrealtime(); }
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int OB = C::MODE == MODE_FM ? 4 : 8, PL = C::NOUT * OB / 16;
    const int n_edge_wg = a.n_edge * a.n_streams;
    if ((int)blockIdx.x < n_edge_wg) {                       // workgroup-uniform: the first workgroups are edge workers
        rx_pipe_edge_worker<C>(a, lds, (int)blockIdx.x);     // (dispatched first, they overlap the interior work)
        return;
    }
    const int per_stream = a.n_int - 1;                      // interior tiles per stream: 1 .. n_int-1
    const int items = per_stream * a.n_streams;
    // Work distribution: worker w starts on item w; after that it pulls chunks of queue_k consecutive
    // items from an atomic counter, so every worker stays busy until the queue is dry: the kernel's tail
    // is a few tiles, not one worker lifetime.  The grab for the chunk AFTER the next one is issued when a
    // chunk is entered and travels through a VGPR of thread 0 and one LDS slot, a whole tile ahead of its
    // use.  (Chunks, not single tiles: same-address device-scope atomics retire at ~12 ns each.)
    // queue_k == 0: static striding by grid_int (A/B switch).
    int *qslot = (int *)(lds + C::IN_BYTES + C::TAIL_BYTES);
    const int K = a.queue_k;
    const int dyn0 = a.grid_int + K * a.grid_int;              // first item handed out by ticket (see below)
    int pos = 0, end = 0, nb_base = 0;           // current chunk [pos, end), base of the chunk after it

    TileRegs<C, KIND> regs;
    int item = (int)blockIdx.x - n_edge_wg;
    {
        // The worker's first chunk is static -- the K items behind the grid's first tiles that its own number selects -- so
        // that nothing stands between the launch and the first tile's loads: a ticket drawn here cost every workgroup of the
        // grid one same-address atomic before its first load (~12 ns each, one after the other: the last of 1 024 workers
        // started 12 us late, which is what a one-round launch -- one second of one stream -- mostly consisted of).
        // Tickets number the chunks behind those: [dyn0 + K * ticket, + K).
        if (K > 0) nb_base = a.grid_int + K * item;
        if (K > 0 && item == 0 && threadIdx.x == 0) __atomic_store_n(a.queue_next, 0u, __ATOMIC_RELAXED);   // (the next launch starts behind this one's end)
        const int s0 = item / per_stream, tile0 = 1 + item % per_stream;
        const void *in0 = KIND == CL_PIPE_IN_CF32 ? (const void *)((const f32x2 *)a.in + (long)s0 * a.in_stride)
                                                 : (const void *)((const uint32_t *)a.in + (long)s0 * a.in_stride);
        tile_issue_loads<C, KIND>(a, regs, in0, (long)tile0 * C::TILE_IN - C::HALO, threadIdx.x);
        unsigned int g0 = 0;
        if (K > 0 && threadIdx.x == 0) g0 = rx_queue_grab(a.tickets);
        // the first tile's words are waited for here (once per worker), so that no path into the loop carries
        // pending loads: the staging at the loop top then needs no vmcnt wait at all (see the note after the FIR)
#pragma unroll
        for (int k = 0; k < TileRegs<C, KIND>::NV; k++) asm volatile("" : "+v"(regs.w[k]));
        asm volatile("" : "+v"(g0));
        if (K > 0 && threadIdx.x == 0) *qslot = (int)g0;     // read behind the loop's first barrier
    }
    while (item < items) {
        // Keep per-iteration values per-iteration: without these the compiler hoists every tap load
        // (88 SGPRs -> spilled to VGPR lanes) and every lane address computation (50+ VGPRs) out of
        // the persistent loop, which costs two waves of occupancy.
        int t = threadIdx.x;
        const float *fir = a.fir, *rs = a.rs;
        asm volatile("" : "+v"(t));
        asm volatile("" : "+s"(fir), "+s"(rs));
        const int s = item / per_stream, tile = 1 + item % per_stream;
        const long S = (long)tile * C::TILE_IN;              // first new input of this tile
        const bool bad = tile_sync_bad<C>(a, s, S);

        DIAG_STAMP(ts0);
#if CLHIP_RX_PRIO
        __builtin_amdgcn_s_setprio(3);                       // a wave that is moving data issues ahead of the waves that are in their FIR
#endif
        tile_regs_to_lds<C, KIND, HIF>(regs, lds, t);
        DIAG_STAMP(ts1);
        RX_BARRIER();
        DIAG_STAMP(ts2);
        int next;
        bool want_grab = false;
        if (K == 0) next = item + a.grid_int;
        else if (pos < end) next = pos++;
        else {                                               // enter the next chunk; ask for the one after it (below)
            pos = nb_base; end = pos + K;
            nb_base = dyn0 + K * __builtin_amdgcn_readfirstlane(*qslot);
            next = pos++;
            want_grab = next < items;
        }
        // The grab for the chunk after next goes out in front of the prefetch loads and is collected with them, behind
        // the FIR.  EVERY tile issues the atomic -- on the tiles that take nothing from the queue it adds 0 to an idle word
        // of the worker's own -- so that the register it returns in is written by the atomic and by nothing else.  With a
        // second definition (a zero, a copy) on the other path the compiler protects that write against the atomic it
        // believes may still be in flight from an earlier tile: s_waitcnt vmcnt(0) at the top of EVERY tile, right
        // behind the previous tile's twelve output stores -- each tile's loads then left only after the previous tile's
        // stores had drained (measured: the kernel took 0.92 ms whatever arithmetic was removed from it).
        unsigned int grabbed = 0;
        if (K > 0 && threadIdx.x == 0) {
            const int wk = (int)blockIdx.x - n_edge_wg;
            unsigned int *slot = want_grab ? a.tickets : a.queue + RX_QUEUE_IDLE + 16 * (wk & (RX_QUEUE_MAX_WORKERS - 1));
            grabbed = rx_queue_grab(slot, want_grab ? 1u : 0u);
        }
        if (next < items) {                                  // prefetch the next item's raw words
            const int sn = next / per_stream, tn = 1 + next % per_stream;
            const void *inn = KIND == CL_PIPE_IN_CF32 ? (const void *)((const f32x2 *)a.in + (long)sn * a.in_stride)
                                                     : (const void *)((const uint32_t *)a.in + (long)sn * a.in_stride);
            tile_issue_loads<C, KIND>(a, regs, inn, (long)tn * C::TILE_IN - C::HALO, t);
        }
        f32x2 acc[C::R];
        f32x4 pc[PL];
        f32x2 b_last = {0.f, 0.f};
#if CLHIP_RX_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        if constexpr (C::FFA) fir_tile_ffa<C>(lds, t, fir, acc, b_last); else fir_tile<C>(lds, t, fir, acc);
#if CLHIP_RX_PRIO
        __builtin_amdgcn_s_setprio(CLHIP_RX_PRIO > 1 ? 3 : 0);
#endif
        // Retire the prefetch here, where only loads are outstanding and they have long landed.  vmcnt counts
        // loads and stores together: waiting for these registers at the top of the next tile would also wait
        // for this tile's 12 output stores to reach memory.
#pragma unroll
        for (int k = 0; k < TileRegs<C, KIND>::NV; k++) asm volatile("" : "+v"(regs.w[k]));
        // (the grab, older than those loads, has returned with them: it moves to its LDS slot, read behind barriers)
        asm volatile("" : "+v"(grabbed));
        if (want_grab && threadIdx.x == 0) *qslot = (int)grabbed;
        if constexpr (DIAG) {                                // pin the phase's results before its stamp
#pragma unroll
            for (int k = 0; k < C::R; k++) asm volatile("" : "+v"(acc[k]));
        }
        DIAG_STAMP(ts3);
        second_stage<C>(lds, t, rs, acc, pc, b_last);
        if constexpr (DIAG) {
#pragma unroll
            for (int k = 0; k < PL; k++) asm volatile("" : "+v"(pc[k]));
        }
        DIAG_STAMP(ts4);
        const long tile_e0 = C::MODE == MODE_FM ? (S - C::HFA) : (S - C::HFA) / C::M * C::L;
        const long lo = C::MODE == MODE_FM ? S : S / C::M * C::L;
        if (!bad) store_tile<C, false>(a, lds, t, (unsigned char *)a.out + (long)s * a.out_stride * OB, tile_e0, lo, 0, pc);
        DIAG_STAMP(ts5);
        RX_BARRIER();                                        // the next item's staging overwrites this LDS
        DIAG_STAMP(ts6);
        if constexpr (DIAG) {
            d_stage += ts1 - ts0; d_bar0 += ts2 - ts1; d_fir += ts3 - ts2; d_second += ts4 - ts3;
            d_store += ts5 - ts4; d_bar2 += ts6 - ts5; d_tiles += 1;
        }
        item = next;
    }
    if constexpr (DIAG) {
        if (a.diag && (threadIdx.x & 63) == 0) {
            unsigned long long *o = a.diag + ((size_t)blockIdx.x * (C::NT / 64) + (threadIdx.x >> 6)) * 8;
            o[0] = d_stage; o[1] = d_bar0; o[2] = d_fir; o[3] = d_second; o[4] = d_store; o[5] = d_bar2; o[6] = d_tiles;
            // shader clock of this wave's lifetime: d(s_memtime) / d(s_memrealtime) x 100 MHz, packed as two 32-bit deltas
            const unsigned long long dt = __builtin_amdgcn_s_memtime() - k_t0, dr = __builtin_amdgcn_s_memrealtime() - k_r0;
            o[7] = (dt << 32) | (dr & 0xffffffffull);
        }
    }
}

// ---------------------------------------------------------------------------
// history update: hist_out = last `halo` samples of [hist_in | in[0..n_in)]
// ---------------------------------------------------------------------------
__device__ __forceinline__ void pipe_update_hist(const PipeArgs &a, int s, int t, int nt)
{
    const int halo = a.halo;
    const void *in = a.in_kind == CL_PIPE_IN_CF32
                         ? (const void *)((const f32x2 *)a.in + (long)s * a.in_stride)
                         : (const void *)((const uint32_t *)a.in + (long)s * a.in_stride);
    const float inv = 1.0f / a.in_scale;
    for (int j = t; j < halo; j += nt) {
        const long g = a.n_in - halo + j;         // position in the new-input index space
        f32x2 v;
        if (g >= 0) v = load_sample(a, in, g) * inv;
        else if (halo + g >= 0) v = a.hist_in[(long)s * halo + halo + g];
        else { v.x = 0.f; v.y = 0.f; }
        a.hist_out[(long)s * halo + j] = v;
    }
}

__global__ void pipe_update_hist_kernel(PipeArgs a)     // generic path only
{
    pipe_update_hist(a, blockIdx.x, threadIdx.x, blockDim.x);
}

// ---------------------------------------------------------------------------
// generic path: three plain kernels through device workspaces
//   X[s] = [hist (halo) | converted input (n)]           (CF32 scale)
//   Y[s][i] = FIR output at input index i - HFg,  i in [0, HFg + n)
//   out: resampler / FM demod / copy of Y
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gen_stage_kernel(PipeArgs a, int halo, f32x2 *__restrict__ X, long x_stride)
{
    const int s = blockIdx.y;
    const void *in = a.in_kind == CL_PIPE_IN_CF32
                         ? (const void *)((const f32x2 *)a.in + (long)s * a.in_stride)
                         : (const void *)((const uint32_t *)a.in + (long)s * a.in_stride);
    const float inv = 1.0f / a.in_scale;
    f32x2 *x = X + (long)s * x_stride;
    const long total = halo + a.n_in;
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += (long)gridDim.x * blockDim.x)
        x[j] = j < halo ? a.hist_in[(long)s * halo + j] : load_sample(a, in, j - halo) * inv;
}

// Generic FIR (any T <= PIPE_MAX_FIR): a workgroup stages 2048 + T - 1 samples in LDS, a lane owns 8 consecutive
// outputs and walks its window once; window position d meets output r through tap (T-1) + r - d, read with scalar
// loads from a zero-padded copy of the taps (8 zeros on both sides: no bounds tests).  Every accumulator sees its
// taps in descending k order, like the fused direct form (bit-identical results).
#define GEN_PER 8
#define GEN_TILE (256 * GEN_PER)
#define GEN_PAD 8
__global__ __launch_bounds__(256) void gen_fir_kernel(const f32x2 *__restrict__ X, long x_stride, int halo,
                                                      const float *__restrict__ taps_pad, int T, int hfg,
                                                      long n_in, f32x2 *__restrict__ Y, long y_stride)
{
    __shared__ f32x2 xs[(GEN_TILE + PIPE_MAX_FIR) / 8 * 9 + 9];      // one pad per 8 samples: lane stride 18 dwords, conflict-free
    const int s = blockIdx.y, t = threadIdx.x;
    const f32x2 *x = X + (long)s * x_stride + halo;      // x[0] = first new input; negative = history
    f32x2 *y = Y + (long)s * y_stride;
    const long total = hfg + n_in;
    const cfloat_t *__restrict__ tp = (const cfloat_t *)taps_pad + GEN_PAD;      // tp[k], k in [-8, T+8)
    for (long j0 = (long)blockIdx.x * GEN_TILE; j0 < total; j0 += (long)gridDim.x * GEN_TILE) {
        const long n0 = j0 - hfg;                        // input index of the tile's first output
        __syncthreads();                                 // the previous tile's windows are done
        for (int i = t; i < GEN_TILE + T - 1; i += 256) {
            const long g = n0 - (T - 1) + i;
            f32x2 z = {0.f, 0.f};
            xs[i + (i >> 3)] = g < n_in ? x[g] : z;
        }
        __syncthreads();
        f32x2 acc[GEN_PER];
#pragma unroll
        for (int r = 0; r < GEN_PER; r++) { acc[r].x = 0.f; acc[r].y = 0.f; }
        const f32x2 *w = xs + t * (GEN_PER + 1);         // w[d + d/8] = x[n0 + 8t - (T-1) + d]
        for (int d = 0; d < T - 1 + GEN_PER; d++) {
            const f32x2 xv = w[d + (d >> 3)];
            const cfloat_t *q = tp + (T - 1 - d);       // tap for output r: q[r]
#pragma unroll
            for (int r = 0; r < GEN_PER; r++) acc[r] += xv * q[r];
        }
#pragma unroll
        for (int r = 0; r < GEN_PER; r++)
            if (j0 + t * GEN_PER + r < total) y[j0 + t * GEN_PER + r] = acc[r];
    }
}

__global__ __launch_bounds__(256) void gen_resample_kernel(const f32x2 *__restrict__ Y, long y_stride, int hfg,
                                                           const float *__restrict__ rs, int n_rs, int L, int M,
                                                           unsigned long long n0, long n_out,
                                                           f32x2 *__restrict__ out, long out_stride)
{
    const int s = blockIdx.y;
    const f32x2 *y = Y + (long)s * y_stride + hfg;       // y[0] = FIR output at the first new input
    f32x2 *o = out + (long)s * out_stride;
    const unsigned long long m0 = (n0 * L + M - 1) / M;
    const int KP = (n_rs + L - 1) / L;
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < n_out; j += (long)gridDim.x * blockDim.x) {
        const unsigned long long tp = (m0 + j) * M;
        const long b = (long)(tp / L - n0);
        const int p = (int)(tp % L);
        f32x2 accv = {0.f, 0.f};
        for (int i = 0; i < KP; i++) {
            const int k = p + i * L;
            if (k < n_rs) accv += y[b - i] * rs[k];
        }
        o[j] = accv;
    }
}

__global__ __launch_bounds__(256) void gen_fm_kernel(const f32x2 *__restrict__ Y, long y_stride, int hfg,
                                                     long n, float *__restrict__ out, long out_stride)
{
    const int s = blockIdx.y;
    const f32x2 *y = Y + (long)s * y_stride + hfg;
    float *o = out + (long)s * out_stride;
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (long)gridDim.x * blockDim.x) {
        const f32x2 c = y[j], p = y[j - 1];
        o[j] = clhip_atan2f(c.y * p.x - c.x * p.y, c.x * p.x + c.y * p.y);
    }
}

__global__ __launch_bounds__(256) void gen_copy_kernel(const f32x2 *__restrict__ Y, long y_stride, int hfg, long n,
                                                       f32x2 *__restrict__ out, long out_stride)
{
    const int s = blockIdx.y;
    const f32x2 *y = Y + (long)s * y_stride + hfg;
    f32x2 *o = out + (long)s * out_stride;
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (long)gridDim.x * blockDim.x) o[j] = y[j];
}

// ---------------------------------------------------------------------------
// host side of the pipe object
// ---------------------------------------------------------------------------
typedef PipeCfg<64, 3, 2, 8, MODE_IQ, 16, 256> CfgC2;     // config 2: FIR64 + 3/2
typedef PipeCfg<64, 3, 2, 8, MODE_IQ, 16, 256, true> CfgC2f;   // the same with the 2-parallel fast FIR
typedef PipeCfg<64, 1, 1, 1, MODE_FM, 16, 256> CfgC3;     // config 3: FIR64 + FM demod
typedef PipeCfg<128, 5, 4, 8, MODE_IQ, 16, 256> CfgC4;    // config 4: FIR128 + 5/4
typedef PipeCfg<64, 1, 1, 1, MODE_FM, 16, 256, true> CfgC3f;
typedef PipeCfg<128, 5, 4, 8, MODE_IQ, 16, 256, true> CfgC4f;
typedef PipeCfg<64, 1, 1, 1, MODE_IQ, 16, 256, true> CfgF64f;
typedef PipeCfg<128, 1, 1, 1, MODE_IQ, 16, 256, true> CfgF128f;
typedef PipeCfg<64, 1, 2, 8, MODE_IQ, 16, 256> CfgD2;      // decimators after FIR64: 1/2, 1/4, 3/4 (8 taps per phase)
typedef PipeCfg<64, 1, 2, 8, MODE_IQ, 16, 256, true> CfgD2f;
typedef PipeCfg<64, 1, 4, 8, MODE_IQ, 16, 256> CfgD4;
typedef PipeCfg<64, 1, 4, 8, MODE_IQ, 16, 256, true> CfgD4f;
typedef PipeCfg<64, 3, 4, 8, MODE_IQ, 16, 256> CfgD34;
typedef PipeCfg<64, 3, 4, 8, MODE_IQ, 16, 256, true> CfgD34f;
typedef PipeCfg<128, 3, 2, 8, MODE_IQ, 16, 256> CfgL32;     // longer FIRs with the other second stages: any FIR=<ntaps>
typedef PipeCfg<128, 3, 2, 8, MODE_IQ, 16, 256, true> CfgL32f;   // kwarg of 65..128 taps (zero-padded) stays fused
typedef PipeCfg<128, 1, 1, 1, MODE_FM, 16, 256> CfgLFM;
typedef PipeCfg<128, 1, 1, 1, MODE_FM, 16, 256, true> CfgLFMf;
typedef PipeCfg<64, 5, 4, 8, MODE_IQ, 16, 256> CfgS54;      // FIR <= 64 taps + 5/4
typedef PipeCfg<64, 5, 4, 8, MODE_IQ, 16, 256, true> CfgS54f;
typedef PipeCfg<64, 1, 1, 1, MODE_IQ, 16, 256> CfgF64;    // FIR64 only
typedef PipeCfg<128, 1, 1, 1, MODE_IQ, 16, 256> CfgF128;  // FIR128 only

struct clhip_rx_pipe {
    int n_streams, channel, T, n_rs, L, M, mode;
    float fir[PIPE_MAX_FIR], rs[PIPE_MAX_RS];
    int halo;                      // history length kept per stream (pre-FIR samples)
    int hfg;                       // FIR-output history the second stage needs
    f32x2 *hist[2];                // ping-pong [n_streams][halo]
    int cur;
    unsigned long long n_total;    // inputs consumed so far (per stream)
    unsigned long long undo_n_total; bool can_undo;   // pre-call state of the last run (clhip_rx_pipe_rollback)
    // streams advancing INDEPENDENTLY (stream groups at the host boundary: clhip_rx_pipe_epoch_begin / _run_range / _epoch_end):
    unsigned long long *nt_s;      // [n_streams] inputs consumed per stream; the classic calls keep every entry == n_total
    uint8_t *ran;                  // [n_streams] streams a range run of the open epoch has advanced
    bool epoch_open;
    int32_t *d_flag, *h_flag;      // clhip_rx_pipe_run_smi: the device-side sync verdict, a word of pinned host memory the kernel
    bool flag_mapped;              //  stores to directly (d_flag = its device address); if the device cannot reach it, a device word + copy
    bool offs_writeback;           // run_smi zeroes d_offs after an in-kernel verdict (off: the caller only reads h_offs)
    void *h_sink;                  // one-shot: the next run_smi also copies its outputs here (before its synchronisation)
    hipStream_t last_stream; bool last_stream_valid;   // where the last run was queued (reset waits for it)
    bool force_generic;
    int fused_id;                  // -1 = none
    // generic workspaces
    const int32_t *chk_offs; size_t chk_chunk_samples; int32_t *chk_flag;   // optional sync validation
    float *d_fir, *d_fir_int, *d_rs;   // taps; d_fir_int = taps/4096 for integer inputs
    float *d_fir_pad;                  // [8 zeros | taps | zeros] for the generic FIR kernel
    float *d_ffa, *d_ffa_int;          // [H0 | H1 | H0+H1] for the 2-parallel fast FIR, same two scalings
    bool ffa;                          // the selected fused instantiation uses them
    unsigned long long *diag;          // optional stamp buffer (diagnostic kernel build)
    unsigned int *queue;               // tile queue of the fused kernel: two ticket counters that take turns between launches + the workers' idle words
    int queue_parity;                  // which counter the next launch that draws tickets uses (the launch before it zeroed that one)
    f32x2 *X, *Y;
    size_t x_cap, y_cap;           // elements per stream
};

// Fused instantiations exist for T = 64 and T = 128.  A FIR of any other length up to 128 taps runs through the next
// one up with its taps zero-padded at the END (k >= n_fir): y[n] = sum_k h[k] x[n-k] is unchanged, the extra products
// are exact zeros added to the accumulators first (taps are walked in descending k), so the direct form stays
// bit-identical to the generic kernels' result for the unpadded filter; the pipe simply keeps a longer history.
static int fused_lookup(int n_fir, int L, int M, int n_rs, int mode, int *halo, int *T_fused)
{
    const bool rs = !(L == 1 && M == 1);
    const int T = n_fir <= 64 ? 64 : 128;
    *T_fused = T;
    if (mode == CL_PIPE_OUT_IQ && T == 64 && L == 3 && M == 2 && n_rs == 24) { *halo = CfgC2::HALO; return 0; }
    if (mode == CL_PIPE_OUT_FM_DEMOD && T == 64 && !rs) { *halo = CfgC3::HALO; return 1; }
    if (mode == CL_PIPE_OUT_IQ && T == 128 && L == 5 && M == 4 && n_rs == 40) { *halo = CfgC4::HALO; return 2; }
    if (mode == CL_PIPE_OUT_IQ && T == 64 && !rs) { *halo = CfgF64::HALO; return 3; }
    if (mode == CL_PIPE_OUT_IQ && T == 128 && !rs) { *halo = CfgF128::HALO; return 4; }
    if (mode == CL_PIPE_OUT_IQ && T == 64 && L == 1 && M == 2 && n_rs == 8) { *halo = CfgD2::HALO; return 5; }
    if (mode == CL_PIPE_OUT_IQ && T == 64 && L == 1 && M == 4 && n_rs == 8) { *halo = CfgD4::HALO; return 6; }
    if (mode == CL_PIPE_OUT_IQ && T == 64 && L == 3 && M == 4 && n_rs == 24) { *halo = CfgD34::HALO; return 7; }
    if (mode == CL_PIPE_OUT_IQ && T == 128 && L == 3 && M == 2 && n_rs == 24) { *halo = CfgL32::HALO; return 8; }
    if (mode == CL_PIPE_OUT_FM_DEMOD && T == 128 && !rs) { *halo = CfgLFM::HALO; return 9; }
    if (mode == CL_PIPE_OUT_IQ && T == 64 && L == 5 && M == 4 && n_rs == 40) { *halo = CfgS54::HALO; return 10; }
    return -1;
}

extern "C" clhip_rx_pipe *clhip_rx_pipe_create(int n_streams, int channel, const float *h_fir, int n_fir,
                                               const float *h_rs, int n_rs, int up, int down, int out_mode)
{
    if (n_streams <= 0 || n_fir <= 0 || n_fir > PIPE_MAX_FIR || !h_fir || up <= 0 || down <= 0) {
        clhip_set_error("clhip_rx_pipe_create: bad arguments (1..%d FIR taps)", PIPE_MAX_FIR);
        return nullptr;
    }
    const bool resamp = !(up == 1 && down == 1);
    if (resamp && (!h_rs || n_rs <= 0 || n_rs > PIPE_MAX_RS)) {
        clhip_set_error("clhip_rx_pipe_create: resampler needs 1..%d prototype taps", PIPE_MAX_RS);
        return nullptr;
    }
    if (out_mode == CL_PIPE_OUT_FM_DEMOD && resamp) {
        clhip_set_error("clhip_rx_pipe_create: FM demod output does not take a resampler");
        return nullptr;
    }
    clhip_rx_pipe *p = new (std::nothrow) clhip_rx_pipe();
    if (!p) return nullptr;
    memset(p, 0, sizeof *p);
    p->offs_writeback = true;
    p->n_streams = n_streams; p->channel = channel; p->T = n_fir; p->L = up; p->M = down;
    p->n_rs = resamp ? n_rs : 0; p->mode = out_mode;
    memcpy(p->fir, h_fir, sizeof(float) * n_fir);
    if (resamp) memcpy(p->rs, h_rs, sizeof(float) * n_rs);
    const int kp = resamp ? (n_rs + up - 1) / up : 1;
    p->hfg = out_mode == CL_PIPE_OUT_FM_DEMOD ? 1 : (resamp ? kp - 1 : 0);
    int halo = 0, t_fused = n_fir;
    p->fused_id = fused_lookup(n_fir, up, down, p->n_rs, out_mode, &halo, &t_fused);
    p->halo = p->fused_id >= 0 ? halo : round_up(n_fir - 1 + p->hfg, 4) + 4;
    const size_t hb = sizeof(f32x2) * (size_t)n_streams * p->halo;
    for (int i = 0; i < 2; i++) {
        p->hist[i] = (f32x2 *)clhip_malloc(hb);
        if (!p->hist[i]) { clhip_rx_pipe_destroy(p); return nullptr; }
        (void)hipMemset(p->hist[i], 0, hb);
    }
    p->nt_s = (unsigned long long *)calloc((size_t)n_streams, sizeof(unsigned long long));
    p->ran = (uint8_t *)calloc((size_t)n_streams, 1);
    if (!p->nt_s || !p->ran) { clhip_rx_pipe_destroy(p); return nullptr; }
    p->d_fir = (float *)clhip_malloc(sizeof(float) * PIPE_MAX_FIR);
    p->d_fir_int = (float *)clhip_malloc(sizeof(float) * PIPE_MAX_FIR);
    p->d_rs = (float *)clhip_malloc(sizeof(float) * PIPE_MAX_RS);
    p->queue = (unsigned int *)clhip_malloc((RX_QUEUE_WORDS + 64) * sizeof(unsigned int));   // + the second ticket counter, 256 bytes behind
    if (!p->d_fir || !p->d_fir_int || !p->d_rs || !p->queue) { clhip_rx_pipe_destroy(p); return nullptr; }
    (void)hipMemset(p->queue, 0, (RX_QUEUE_WORDS + 64) * sizeof(unsigned int));
    float scaled[PIPE_MAX_FIR];
    for (int k = 0; k < PIPE_MAX_FIR; k++) scaled[k] = p->fir[k] / 4096.0f;   // exact: power of two
    (void)hipMemcpy(p->d_fir, p->fir, sizeof(float) * PIPE_MAX_FIR, hipMemcpyHostToDevice);
    {
        float padded[PIPE_MAX_FIR + 2 * GEN_PAD + GEN_PER];
        memset(padded, 0, sizeof padded);
        memcpy(padded + GEN_PAD, p->fir, sizeof(float) * n_fir);
        p->d_fir_pad = (float *)clhip_malloc(sizeof padded);
        if (!p->d_fir_pad) { clhip_rx_pipe_destroy(p); return nullptr; }
        (void)hipMemcpy(p->d_fir_pad, padded, sizeof padded, hipMemcpyHostToDevice);
    }
    (void)hipMemcpy(p->d_fir_int, scaled, sizeof(float) * PIPE_MAX_FIR, hipMemcpyHostToDevice);
    if (p->fused_id >= 0 || (n_fir & 1) == 0) {
        // [H0 | H1 | H0+H1] of the filter as the fused kernel sees it (zero-padded to its T: p->fir is zero past n_fir)
        const int th = (p->fused_id >= 0 ? t_fused : n_fir) / 2;
        float f[2][3 * PIPE_MAX_FIR / 2];
        for (int v = 0; v < th; v++) {
            for (int k = 0; k < 2; k++) {
                const float *src = k ? scaled : p->fir;
                f[k][v] = src[2 * v]; f[k][th + v] = src[2 * v + 1]; f[k][2 * th + v] = src[2 * v] + src[2 * v + 1];
            }
        }
        p->d_ffa = (float *)clhip_malloc(sizeof f[0]);
        p->d_ffa_int = (float *)clhip_malloc(sizeof f[0]);
        if (!p->d_ffa || !p->d_ffa_int) { clhip_rx_pipe_destroy(p); return nullptr; }
        (void)hipMemcpy(p->d_ffa, f[0], sizeof f[0], hipMemcpyHostToDevice);
        (void)hipMemcpy(p->d_ffa_int, f[1], sizeof f[1], hipMemcpyHostToDevice);
    }
    (void)hipMemcpy(p->d_rs, p->rs, sizeof(float) * PIPE_MAX_RS, hipMemcpyHostToDevice);
    // the fills above ran on the null stream, which the shim's non-blocking streams do not wait for
    (void)hipStreamSynchronize(nullptr);
    return p;
}

extern "C" void clhip_rx_pipe_destroy(clhip_rx_pipe *p)
{
    if (!p) return;
    clhip_free(p->hist[0]); clhip_free(p->hist[1]);
    clhip_free(p->d_fir); clhip_free(p->d_fir_int); clhip_free(p->d_rs); clhip_free(p->d_fir_pad); clhip_free(p->d_ffa); clhip_free(p->d_ffa_int);
    clhip_free(p->X); clhip_free(p->Y); clhip_free(p->queue);
    if (!p->flag_mapped) clhip_free(p->d_flag);
    clhip_host_free(p->h_flag);
    free(p->nt_s); free(p->ran);
    delete p;
}

extern "C" void clhip_rx_pipe_reset(clhip_rx_pipe *p)
{
    const size_t hb = sizeof(f32x2) * (size_t)p->n_streams * p->halo;
    // a run of this pipe may still be reading / writing the history on its (non-blocking) stream
    if (p->last_stream_valid) (void)hipStreamSynchronize(p->last_stream);
    (void)hipMemset(p->hist[0], 0, hb);
    (void)hipMemset(p->hist[1], 0, hb);
    (void)hipStreamSynchronize(nullptr);
    p->cur = 0; p->n_total = 0; p->can_undo = false; p->epoch_open = false;
    for (int i = 0; i < p->n_streams; i++) p->nt_s[i] = 0;
}

extern "C" void clhip_rx_pipe_seek(clhip_rx_pipe *p, unsigned long long n_total)
{
    p->n_total = n_total; p->can_undo = false;
    for (int i = 0; i < p->n_streams; i++) p->nt_s[i] = n_total;
}

// A run never touches the history it read (ping-pong buffers) and the polyphase phase is a host counter, so the
// pre-call state of the LAST run is still complete: undoing it is a pointer flip.  The caller must have synchronised
// the stream of that run (its kernels may still be writing the other history buffer).
extern "C" size_t clhip_rx_pipe_out_elem_bytes(const clhip_rx_pipe *p) { return p->mode == CL_PIPE_OUT_FM_DEMOD ? sizeof(float) : sizeof(f32x2); }
extern "C" void clhip_rx_pipe_set_host_sink(clhip_rx_pipe *p, void *h_out) { if (p) p->h_sink = h_out; }
extern "C" void clhip_rx_pipe_set_offs_writeback(clhip_rx_pipe *p, int on) { if (p) p->offs_writeback = on != 0; }

extern "C" int clhip_rx_pipe_rollback(clhip_rx_pipe *p)
{
    if (!p || !p->can_undo) { clhip_set_error("clhip_rx_pipe_rollback: no run to undo"); return -1; }
    p->cur ^= 1;
    p->n_total = p->undo_n_total;
    for (int i = 0; i < p->n_streams; i++) p->nt_s[i] = p->n_total;
    p->can_undo = false;
    return 0;
}
extern "C" size_t clhip_rx_pipe_halo(const clhip_rx_pipe *p) { return (size_t)p->halo; }

extern "C" void clhip_rx_pipe_force_generic(clhip_rx_pipe *p, int on) { p->force_generic = on != 0; }

// Diagnostic only (tools/phase_stamps.py): run config 2 through the s_memtime-stamped build of the fused
// kernel; d_buf receives, per wave, the cycle sums of {stage, barrier, FIR, second stage, store, barrier, tiles}.
extern "C" void clhip_rx_pipe_set_diag(clhip_rx_pipe *p, unsigned long long *d_buf) { p->diag = d_buf; }

extern "C" void clhip_rx_pipe_set_sync_check(clhip_rx_pipe *p, const int32_t *d_offs, size_t chunk_samples,
                                             int32_t *d_bad_flag)
{
    if (chunk_samples & (chunk_samples - 1)) { d_offs = nullptr; d_bad_flag = nullptr; clhip_set_error("sync check needs a power-of-two chunk size"); }
    p->chk_offs = d_offs; p->chk_chunk_samples = chunk_samples; p->chk_flag = d_bad_flag;
}

static size_t out_count_at(const clhip_rx_pipe *p, unsigned long long n0, size_t n_in)
{
    if (p->mode == CL_PIPE_OUT_FM_DEMOD || (p->L == 1 && p->M == 1)) return n_in;
    const unsigned long long n1 = n0 + n_in;
    return (size_t)((n1 * p->L + p->M - 1) / p->M - (n0 * p->L + p->M - 1) / p->M);
}
extern "C" size_t clhip_rx_pipe_out_count(const clhip_rx_pipe *p, size_t n_in) { return out_count_at(p, p->n_total, n_in); }
extern "C" size_t clhip_rx_pipe_out_count_stream(const clhip_rx_pipe *p, int s, size_t n_in)
{
    return p && s >= 0 && s < p->n_streams ? out_count_at(p, p->nt_s[s], n_in) : 0;
}
extern "C" unsigned long long clhip_rx_pipe_stream_total(const clhip_rx_pipe *p, int s) { return p && s >= 0 && s < p->n_streams ? p->nt_s[s] : 0; }

static int uses_fused_at(const clhip_rx_pipe *p, unsigned long long n0, int in_kind)
{
    if (p->force_generic || p->fused_id < 0) return 0;
    if (in_kind < 0 || in_kind > 2) return 0;
    // the tile-local polyphase pattern needs the call to start on a phase-0 input
    if (((n0 % (unsigned long long)p->M) * p->L) % p->M != 0) return 0;
    return 1;
}
extern "C" int clhip_rx_pipe_uses_fused(const clhip_rx_pipe *p, size_t n_in, int in_kind) { (void)n_in; return uses_fused_at(p, p->n_total, in_kind); }

template <class C, int KIND, bool HIF, bool DIAG = false>
static int launch_pipe(PipeArgs &a, hipStream_t s)
{
    const long items = (long)(a.n_int - 1) * a.n_streams;
    // persistent interior grid = what stays resident; worker w starts on item w and then pulls items
    // from the pipe's tile queue; edge workers sit in front of them in the same launch
    // per (instantiation, device): one process may drive several GPUs, from several threads
    static std::mutex mu;
    static int resident_on[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    int resident;
    {
        std::lock_guard<std::mutex> lock(mu);
        const int slot = dev >= 0 && dev < 64 ? dev : 0;
        if (!resident_on[slot]) {
            (void)hipFuncSetAttribute((const void *)rx_pipe_fused_kernel<C, KIND, HIF, DIAG>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
            int cus = 256, per_cu = 0;
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)rx_pipe_fused_kernel<C, KIND, HIF, DIAG>,
                                                             C::NT, C::LDS_BYTES) != hipSuccess || per_cu < 1)
                per_cu = 2;
            // The occupancy API under-reports this kernel (4 x 37.5 KB of LDS and 4 waves per SIMD at 128 VGPRs both
            // fit a CU; profiles/r01/c_pmc.json shows 3.8 resident waves per SIMD with the grid below).  A persistent
            // grid larger than what is resident is still correct -- workers never wait for each other, the surplus
            // workgroups simply start when a slot frees up and find the queue (nearly) dry -- so the measured
            // optimum is preferred over the API's answer.
            const int by_lds = (160 * 1024) / C::LDS_BYTES, by_waves = 16 / (C::NT / 64);
            const int want = by_lds < by_waves ? by_lds : by_waves;
            if (per_cu < want && want >= 4) per_cu = want;
            // queue mode: exactly what is resident (the tile queue keeps every worker busy to the end);
            // static striding: 4x oversubscribed, queued workgroups back-fill as residents finish
            if (a.queue_k == 0) per_cu *= 4;
            resident_on[slot] = cus * per_cu;
        }
        resident = resident_on[slot];
    }
    a.grid_int = (int)(items < resident ? items : resident);
    if (items <= resident) a.queue_k = 0;                      // one tile per worker: nothing to hand out, no atomics at all
    const unsigned grid = (unsigned)a.grid_int + (unsigned)a.n_edge * a.n_streams;
    hipLaunchKernelGGL((rx_pipe_fused_kernel<C, KIND, HIF, DIAG>), dim3(grid), dim3(C::NT), C::LDS_BYTES, s, a);
    CLHIP_CHECK_LAUNCH();
    return 0;
}

template <class C>
static int launch_fused(PipeArgs &a, hipStream_t s)
{
    const long tiles = (a.n_in + C::TILE_IN - 1) / C::TILE_IN;
    if (tiles <= 0) return 0;
    // interior tile: 1 <= tile and tile*TILE_IN - HALO + NLOAD <= n_in (then S + TILE_IN <= n_in too)
    long n_int = a.n_in >= C::NLOAD - C::HALO ? (a.n_in - C::NLOAD + C::HALO) / C::TILE_IN + 1 : 1;
    if (n_int < 1) n_int = 1;
    if (n_int > tiles) n_int = tiles;
    a.n_int = (int)n_int;
    a.n_edge = (int)(1 + (tiles - n_int));                  // tile 0 + the tail tiles
    switch (a.in_kind) {
    case CL_PIPE_IN_SMI_WORDS:
        if constexpr (C::FFA && C::T == 64 && C::L == 3) {      // the one diagnostic (stamped) instantiation
            if (a.diag && a.channel != CL_CHANNEL_HIF) return launch_pipe<C, CL_PIPE_IN_SMI_WORDS, false, true>(a, s);
        }
        return a.channel == CL_CHANNEL_HIF ? launch_pipe<C, CL_PIPE_IN_SMI_WORDS, true>(a, s)
                                           : launch_pipe<C, CL_PIPE_IN_SMI_WORDS, false>(a, s);
    case CL_PIPE_IN_CS16: return launch_pipe<C, CL_PIPE_IN_CS16, false>(a, s);
    default: return launch_pipe<C, CL_PIPE_IN_CF32, false>(a, s);
    }
}

static int ensure_ws(clhip_rx_pipe *p, size_t n_in)
{
    const size_t xe = p->halo + n_in, ye = p->hfg + n_in;
    if (xe > p->x_cap) {
        clhip_free(p->X);
        p->X = (f32x2 *)clhip_malloc(sizeof(f32x2) * xe * p->n_streams);
        p->x_cap = p->X ? xe : 0;
        if (!p->X) return -1;
    }
    if (ye > p->y_cap) {
        clhip_free(p->Y);
        p->Y = (f32x2 *)clhip_malloc(sizeof(f32x2) * ye * p->n_streams);
        p->y_cap = p->Y ? ye : 0;
        if (!p->Y) return -1;
    }
    return 0;
}

// One launch (fused) or launch chain (generic) over the streams [first, first + count) of the pipe, all of them at input position
// `nt` (equal up to multiples of 2 M: the same polyphase phase and fast-FIR parity): reads hist[cur], writes hist[cur ^ 1]; the
// callers move the pipe's state.  d_in / d_out point at stream `first`'s row.
static long run_impl(clhip_rx_pipe *p, int first, int count, unsigned long long nt, int in_kind, const void *d_in, size_t in_stride,
                     size_t n_in, void *d_out, size_t out_stride, hipStream_t s)
{
    p->last_stream = s; p->last_stream_valid = true;
    const size_t n_out = out_count_at(p, nt, n_in);

    PipeArgs a;
    memset(&a, 0, sizeof a);
    a.in = d_in; a.in_stride = (long)in_stride;
    a.hist_in = p->hist[p->cur] + (size_t)first * p->halo;
    a.out = d_out; a.out_stride = (long)out_stride;
    a.n_in = (long)n_in; a.n_out = (long)n_out;
    a.in_kind = in_kind; a.channel = p->channel; a.n_streams = count;
    a.halo = p->halo; a.hist_out = p->hist[p->cur ^ 1] + (size_t)first * p->halo; a.diag = p->diag; a.queue = p->queue;
    a.tickets = p->queue + (p->queue_parity ? RX_QUEUE_WORDS : 0); a.queue_next = p->queue + (p->queue_parity ? 0 : RX_QUEUE_WORDS);
    // items per ticket: by the shape's tile rate.  Same-address device-scope atomics retire at ~12 ns each, and a ticket per
    // tile is one atomic per 13.5 ns for config 2 (0.9 ms / 66 052 tiles) -- within the counter's rate, asked for a whole tile
    // ahead: 1 against 2 on four boxes 0.8926 / 0.8966, 0.919 / 0.925, 0.8953 / 0.8988, 0.8950 / 0.8976 ms (every pair in that
    // order), config 4 2.448 / 2.455 -- but one per 11.9 ns for config 3 (FM demod: 4 bytes out per sample, short tiles), past
    // it: 0.8164 against 0.7790 ms.  So: one tile per ticket for the two resampling shapes that were measured, two otherwise.
    // (static striding over a 4x oversubscribed grid -- queue_k = 0 -- measured 0.898-0.934 ms against 0.893: profiles/r03/c2_grid_sweep.txt)
    a.queue_k = (p->fused_id == 0 || p->fused_id == 2) ? 1 : 2;
    a.in_scale = in_kind == CL_PIPE_IN_CF32 ? 1.0f : 4096.0f;
    a.fir = in_kind == CL_PIPE_IN_CF32 ? p->d_fir : p->d_fir_int;
    const float *ffa_taps = in_kind == CL_PIPE_IN_CF32 ? p->d_ffa : p->d_ffa_int;
    a.rs = p->d_rs;
    if (in_kind == CL_PIPE_IN_SMI_WORDS && p->chk_flag && p->chk_chunk_samples) {
        int sh = 0;
        while (((size_t)1 << sh) < p->chk_chunk_samples) sh++;
        a.chunk_offs = p->chk_offs; a.chunk_shift = sh;
        a.chunks_per_stream = (long)clhip_div_up(n_in, p->chk_chunk_samples);
        a.bad_flag = p->chk_flag;
    }

#if CLHIP_RX_BOUNDS
    {   // the extent of the call's buffers (all streams)
        const size_t esz = in_kind == CL_PIPE_IN_CF32 ? 8 : 4, ob = p->mode == CL_PIPE_OUT_FM_DEMOD ? 4 : 8;
        a.b_in_lo = d_in; a.b_in_hi = (const unsigned char *)d_in + ((size_t)(count - 1) * in_stride + n_in) * esz;
        a.b_out_lo = d_out; a.b_out_hi = (unsigned char *)d_out + ((size_t)(count - 1) * out_stride + n_out) * ob;
    }
#endif
    bool fused_done = false;
    if (uses_fused_at(p, nt, in_kind)) {
        int rc = -1;
        fused_done = true;
        // 2-parallel fast FIR by default (CLHIP_FFA=0: direct form).  Its lane parity is tied to the
        // absolute sample index, so a call that starts on an odd index uses the direct form.
        static const int ffa_env = getenv("CLHIP_FFA") ? atoi(getenv("CLHIP_FFA")) : 1;
        const bool ffa = ffa_env && ffa_taps && (nt & 1) == 0;
        if (ffa) a.fir = ffa_taps;
        switch (p->fused_id) {
        case 0: rc = ffa ? launch_fused<CfgC2f>(a, s) : launch_fused<CfgC2>(a, s); break;
        case 1: rc = ffa ? launch_fused<CfgC3f>(a, s) : launch_fused<CfgC3>(a, s); break;
        case 2: rc = ffa ? launch_fused<CfgC4f>(a, s) : launch_fused<CfgC4>(a, s); break;
        case 3: rc = ffa ? launch_fused<CfgF64f>(a, s) : launch_fused<CfgF64>(a, s); break;
        case 4: rc = ffa ? launch_fused<CfgF128f>(a, s) : launch_fused<CfgF128>(a, s); break;
        case 5: rc = ffa ? launch_fused<CfgD2f>(a, s) : launch_fused<CfgD2>(a, s); break;
        case 6: rc = ffa ? launch_fused<CfgD4f>(a, s) : launch_fused<CfgD4>(a, s); break;
        case 7: rc = ffa ? launch_fused<CfgD34f>(a, s) : launch_fused<CfgD34>(a, s); break;
        case 8: rc = ffa ? launch_fused<CfgL32f>(a, s) : launch_fused<CfgL32>(a, s); break;
        case 9: rc = ffa ? launch_fused<CfgLFMf>(a, s) : launch_fused<CfgLFM>(a, s); break;
        case 10: rc = ffa ? launch_fused<CfgS54f>(a, s) : launch_fused<CfgS54>(a, s); break;
        }
        if (rc) return -1;
        if (a.queue_k > 0) p->queue_parity ^= 1;              // (a launch with one tile per worker draws no tickets: launch_pipe)
    } else {
        if (ensure_ws(p, n_in)) return -1;
        const unsigned gx = (unsigned)(clhip_div_up(p->halo + n_in, 256) > 4096 ? 4096 : clhip_div_up(p->halo + n_in, 256));
        dim3 grid(gx, count), block(256);               // (the workspaces are scratch: rows [0, count) whatever `first` is)
        hipLaunchKernelGGL(gen_stage_kernel, grid, block, 0, s, a, p->halo, p->X, (long)p->x_cap);
        const unsigned gf = (unsigned)(clhip_div_up(p->hfg + n_in, GEN_TILE) > 8192 ? 8192 : clhip_div_up(p->hfg + n_in, GEN_TILE));
        hipLaunchKernelGGL(gen_fir_kernel, dim3(gf, count), block, 0, s, p->X, (long)p->x_cap, p->halo, p->d_fir_pad, p->T,
                           p->hfg, (long)n_in, p->Y, (long)p->y_cap);
        if (p->mode == CL_PIPE_OUT_FM_DEMOD)
            hipLaunchKernelGGL(gen_fm_kernel, grid, block, 0, s, p->Y, (long)p->y_cap, p->hfg, (long)n_in,
                               (float *)d_out, (long)out_stride);
        else if (p->L == 1 && p->M == 1)
            hipLaunchKernelGGL(gen_copy_kernel, grid, block, 0, s, p->Y, (long)p->y_cap, p->hfg, (long)n_in,
                               (f32x2 *)d_out, (long)out_stride);
        else
            hipLaunchKernelGGL(gen_resample_kernel, grid, block, 0, s, p->Y, (long)p->y_cap, p->hfg, p->d_rs, p->n_rs,
                               p->L, p->M, nt, (long)n_out, (f32x2 *)d_out, (long)out_stride);
        CLHIP_CHECK_LAUNCH();
    }
    if (!fused_done) {
        hipLaunchKernelGGL(pipe_update_hist_kernel, dim3(count), dim3(128), 0, s, a);
        CLHIP_CHECK_LAUNCH();
    }
    return (long)n_out;
}

extern "C" long clhip_rx_pipe_run(clhip_rx_pipe *p, int in_kind, const void *d_in, size_t in_stride,
                                  size_t n_in, void *d_out, size_t out_stride, void *stream)
{
    if (!p || in_kind < 0 || in_kind > 2) { clhip_set_error("clhip_rx_pipe_run: bad arguments"); return -1; }
    if (p->epoch_open) { clhip_set_error("clhip_rx_pipe_run: an epoch of range runs is open"); return -1; }
    if (n_in == 0) return 0;
    if (!d_in || !d_out) { clhip_set_error("clhip_rx_pipe_run: null buffer"); return -1; }
    for (int i = 1; i < p->n_streams; i++)
        if (p->nt_s[i] != p->nt_s[0]) { clhip_set_error("clhip_rx_pipe_run: the pipe's streams have advanced independently (range runs): they no longer move as one"); return -1; }
    const long n_out = run_impl(p, 0, p->n_streams, p->n_total, in_kind, d_in, in_stride, n_in, d_out, out_stride, (hipStream_t)stream);
    if (n_out < 0) return n_out;
    p->undo_n_total = p->n_total; p->can_undo = true;
    p->cur ^= 1;
    p->n_total += n_in;
    for (int i = 0; i < p->n_streams; i++) p->nt_s[i] = p->n_total;
    return n_out;
}

// ---------------------------------------------------------------------------
// Streams of ONE pipe advancing independently: what a stream group at the host boundary needs (cl_group_readStream: N Soapy
// devices of one GPU read in one call -- the reference's unit is one device per channel, soapy_api/SoapyCariboulite.cpp:46-69,
// each with its own caribou_smi_read chunk loop, caribou_smi.c:632-682, so one stream may deliver while its neighbour re-syncs
// or returns -3).  An EPOCH is one such call: every stream reads hist[cur] and writes hist[cur ^ 1] at most once, through
// range runs over disjoint runs of neighbouring streams (base + stride addressing: no kernel change, one launch per run); a
// stream no run touched keeps its state (its history is copied across when the epoch ends); then the buffers flip for all.
// ---------------------------------------------------------------------------
extern "C" int clhip_rx_pipe_epoch_begin(clhip_rx_pipe *p)
{
    if (!p || p->epoch_open) { clhip_set_error("clhip_rx_pipe_epoch_begin: bad pipe or epoch already open"); return -1; }
    memset(p->ran, 0, (size_t)p->n_streams);
    p->epoch_open = true;
    p->can_undo = false;
    return 0;
}

extern "C" long clhip_rx_pipe_run_range(clhip_rx_pipe *p, int first, int count, int in_kind, const void *d_in, size_t in_stride,
                                        size_t n_in, void *d_out, size_t out_stride, void *stream)
{
    if (!p || !p->epoch_open || in_kind < 0 || in_kind > 2 || first < 0 || count < 1 || first + count > p->n_streams) {
        clhip_set_error("clhip_rx_pipe_run_range: bad arguments (is an epoch open?)");
        return -1;
    }
    if (n_in == 0) return 0;
    if (!d_in || !d_out) { clhip_set_error("clhip_rx_pipe_run_range: null buffer"); return -1; }
    const unsigned long long cls = 2ull * (unsigned long long)p->M;
    for (int i = first; i < first + count; i++) {
        if (p->ran[i]) { clhip_set_error("clhip_rx_pipe_run_range: stream %d already ran in this epoch", i); return -1; }
        if (p->nt_s[i] % cls != p->nt_s[first] % cls) {
            clhip_set_error("clhip_rx_pipe_run_range: streams %d and %d are on different polyphase phases: split the range", first, i);
            return -1;
        }
    }
    const long n_out = run_impl(p, first, count, p->nt_s[first], in_kind, d_in, in_stride, n_in, d_out, out_stride, (hipStream_t)stream);
    if (n_out < 0) return n_out;
    for (int i = first; i < first + count; i++) { p->ran[i] = 1; p->nt_s[i] += n_in; }
    return n_out;
}

// A range run of the OPEN epoch is taken back for stream s (the caller read ahead of its client and the client went another way):
// the stream counts as not run -- its history of the epoch before is untouched (ping-pong), what the run wrote for the next epoch is
// overwritten by a later run of this epoch or by epoch_end's copy across -- and its input counter is rewound.  Work queued behind it
// on the same HIP stream sees the right order.
extern "C" int clhip_rx_pipe_unrun_stream(clhip_rx_pipe *p, int s, size_t n_in)
{
    if (!p || !p->epoch_open || s < 0 || s >= p->n_streams || !p->ran[s] || p->nt_s[s] < n_in) {
        clhip_set_error("clhip_rx_pipe_unrun_stream: stream %d has no run of the open epoch to take back", s);
        return -1;
    }
    p->ran[s] = 0;
    p->nt_s[s] -= n_in;
    return 0;
}

extern "C" int clhip_rx_pipe_epoch_end(clhip_rx_pipe *p, void *stream)
{
    if (!p || !p->epoch_open) { clhip_set_error("clhip_rx_pipe_epoch_end: no epoch open"); return -1; }
    hipStream_t s = (hipStream_t)stream;
    const size_t row = sizeof(f32x2) * (size_t)p->halo;
    int i = 0;
    while (i < p->n_streams) {                       // runs of untouched streams: their history moves across unchanged
        if (p->ran[i]) { i++; continue; }
        int j = i;
        while (j < p->n_streams && !p->ran[j]) j++;
        CLHIP_CHECK(hipMemcpyAsync(p->hist[p->cur ^ 1] + (size_t)i * p->halo, p->hist[p->cur] + (size_t)i * p->halo, row * (size_t)(j - i),
                                   hipMemcpyDeviceToDevice, s));
        i = j;
    }
    p->last_stream = s; p->last_stream_valid = true;
    p->cur ^= 1;
    p->n_total = p->nt_s[0];
    p->epoch_open = false;
    return 0;
}

// ---------------------------------------------------------------------------
// caribou_smi_read (caribou_smi.c:632-682) feeding the pipe, bytes resident on the device.
//   fast path: per-chunk sync search, then ONE fused launch straight from the raw words with the search results
//              checked on the device (no host round trip between the two);
//   a chunk out of sync (offs > 0): the call is undone and redone the reference's way -- skip offs bytes, unpack
//              n = (len - 4*(offs/4+1))/4 samples, extrapolate one, leave the rest of the chunk's slots as they
//              are (:319-325,382-389) -- into d_cs16, and the pipe runs from those int16 samples;
//   a chunk without sync (offs < 0): CL_SMI_ERR_SYNC (:665-668), pipe state as before the call.
// ---------------------------------------------------------------------------
extern "C" long clhip_rx_pipe_run_smi(clhip_rx_pipe *p, const uint8_t *d_bytes, size_t stream_stride_bytes,
                                      size_t n_bytes, size_t chunk_len_bytes, int32_t *d_offs, int32_t *h_offs,
                                      int16_t *d_cs16, void *d_out, size_t out_stride, void *stream)
{
    if (!p || !d_bytes || !d_offs || !d_out || chunk_len_bytes == 0 || (chunk_len_bytes & 3)) {
        clhip_set_error("clhip_rx_pipe_run_smi: bad arguments");
        return -1;
    }
    const size_t n_in = n_bytes / 4;                        // read_so_far += ret / 4 (:677)
    void *sink = p->n_streams == 1 ? p->h_sink : nullptr;   // (one stream: the outputs are one contiguous run)
    p->h_sink = nullptr;
    const size_t ob = p->mode == CL_PIPE_OUT_FM_DEMOD ? sizeof(float) : sizeof(f32x2);
    if (n_in == 0) return 0;
    if (p->n_streams > 1 && (stream_stride_bytes & 3)) { clhip_set_error("clhip_rx_pipe_run_smi: stream stride must be whole words"); return -1; }
    hipStream_t s = (hipStream_t)stream;
    const int n_chunks = (int)clhip_div_up(n_bytes, chunk_len_bytes);
    if (!p->d_flag) {
        p->h_flag = (int32_t *)clhip_host_alloc(sizeof(int32_t));
        if (!p->h_flag) return -1;
        p->d_flag = (int32_t *)clhip_host_device_ptr(p->h_flag);
        p->flag_mapped = p->d_flag != nullptr;
        if (!p->flag_mapped) p->d_flag = (int32_t *)clhip_malloc(sizeof(int32_t));
        if (!p->d_flag) return -1;
    }
    const size_t chunk_samples = chunk_len_bytes / 4;
    const bool dev_check = (chunk_samples & (chunk_samples - 1)) == 0 && (n_bytes & 3) == 0 && ((uintptr_t)d_bytes & 15) == 0;
    // the fused kernel verifies each chunk's sync words itself; the byte-granular search only runs when that fails
    const bool in_kernel = dev_check && clhip_rx_pipe_uses_fused(p, n_in, CL_PIPE_IN_SMI_WORDS) != 0;
    auto search = [&]() -> int {
        for (int st = 0; st < p->n_streams; st++)
            if (clhip_smi_find_offsets(d_bytes + (size_t)st * stream_stride_bytes, n_bytes, chunk_len_bytes, chunk_len_bytes,
                                       n_chunks, d_offs + (size_t)st * n_chunks, s))
                return -1;
        return 0;
    };
    if (!in_kernel && search()) return -1;
    // saved so that a caller-armed check (bench.py) survives this call
    const int32_t *keep_offs = p->chk_offs; const size_t keep_cs = p->chk_chunk_samples; int32_t *keep_flag = p->chk_flag;
    long got = -1;
    bool redo = !dev_check;
    if (dev_check) {
        // the generic kernels do not look at the chunk table: a call that takes them is judged from the table itself
        const bool flag_valid = clhip_rx_pipe_uses_fused(p, n_in, CL_PIPE_IN_SMI_WORDS) != 0;
        // (every earlier launch that could raise the flag has been synchronised with: this function always does)
        if (p->flag_mapped) *(volatile int32_t *)p->h_flag = 0;
        else CLHIP_CHECK(hipMemsetAsync(p->d_flag, 0, sizeof(int32_t), s));
        clhip_rx_pipe_set_sync_check(p, in_kernel ? nullptr : d_offs, chunk_samples, p->d_flag);
        got = clhip_rx_pipe_run(p, CL_PIPE_IN_SMI_WORDS, d_bytes, stream_stride_bytes / 4, n_in, d_out, out_stride, stream);
        p->chk_offs = keep_offs; p->chk_chunk_samples = keep_cs; p->chk_flag = keep_flag;
        if (got < 0) return -1;
        if (!p->flag_mapped) CLHIP_CHECK(hipMemcpyAsync(p->h_flag, p->d_flag, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        // the outputs ride out under the same synchronisation (speculatively: a redo below overwrites all of them)
        if (sink && got > 0 && clhip_memcpy_d2h(sink, d_out, (size_t)got * ob, s)) return -1;   // (in pieces when `sink` is pageable client memory)
        CLHIP_CHECK(hipStreamSynchronize(s));
        redo = !flag_valid || *(volatile int32_t *)p->h_flag != 0;
    }
    if (!redo) {
        if (h_offs) memset(h_offs, 0, sizeof(int32_t) * (size_t)n_chunks * p->n_streams);
        if (in_kernel && p->offs_writeback) CLHIP_CHECK(hipMemsetAsync(d_offs, 0, sizeof(int32_t) * (size_t)n_chunks * p->n_streams, s));   // what the search would have written
        return got;
    }
    if (in_kernel && search()) return -1;                   // now the byte-granular search
    // slow path: what did the search find?
    const size_t n_offs = (size_t)n_chunks * p->n_streams;
    int32_t stack_offs[64];
    int32_t *ho = h_offs ? h_offs : (n_offs <= 64 ? stack_offs : (int32_t *)malloc(sizeof(int32_t) * n_offs));
    if (!ho) return -1;
    hipError_t e = hipMemcpyAsync(ho, d_offs, sizeof(int32_t) * n_offs, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    bool any_bad = false, any_lost = false;
    if (e == hipSuccess)
        for (size_t i = 0; i < n_offs; i++) { any_bad |= ho[i] != 0; any_lost |= ho[i] < 0; }
    if (ho != h_offs && ho != stack_offs) free(ho);
    if (e != hipSuccess) { clhip_set_error("clhip_rx_pipe_run_smi: %s", hipGetErrorString(e)); return -1; }
    if (dev_check && !any_bad && (n_bytes & 3) == 0) return got;      // generic-path pipe, everything in sync: done (sink filled above)
    if (dev_check && clhip_rx_pipe_rollback(p)) return -1;            // undo the raw-word run
    if (any_lost) { clhip_set_error("SMI data synchronization failed"); return CL_SMI_ERR_SYNC; }
    if (!d_cs16) { clhip_set_error("clhip_rx_pipe_run_smi: chunks out of sync; redo from re-synchronised int16 samples"); return CL_PIPE_ERR_RESYNC; }
    const size_t cs_stride = n_in + 2;                       // int16 pairs per stream (one spare slot for the extrapolated sample)
    for (int st = 0; st < p->n_streams; st++)
        if (clhip_smi_unpack(p->channel, d_bytes + (size_t)st * stream_stride_bytes, n_bytes, chunk_len_bytes, chunk_len_bytes,
                             n_chunks, d_offs + (size_t)st * n_chunks, CL_FORMAT_CS16, d_cs16 + 2 * cs_stride * st, nullptr, s))
            return -1;
    got = clhip_rx_pipe_run(p, CL_PIPE_IN_CS16, d_cs16, cs_stride, n_in, d_out, out_stride, stream);
    if (got < 0) return -1;
    if (sink && got > 0 && clhip_memcpy_d2h(sink, d_out, (size_t)got * ob, s)) return -1;   // (in pieces when `sink` is pageable client memory)
    CLHIP_CHECK(hipStreamSynchronize(s));
    return got;
}
