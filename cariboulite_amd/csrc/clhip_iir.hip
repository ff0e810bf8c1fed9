// clhip_iir.hip -- the optional RX low-pass of Stream::ReadSamples
// (soapy_api/CaribouliteStream.cpp:291-298): per rail
//     y = (int16_t)(float) LP6( (float)x ),   LP6 = iir1 Butterworth::LowPass<6>
// i.e. a cascade of Direct-Form-II biquads evaluated in fp64 with state carried
// for the life of the stream.  fp32 state fails the 1e-5 bar at the reference's
// narrow cut-offs, so everything here is fp64.
//
// The recursion is strictly sequential per rail; it is made parallel as a blocked
// linear-recurrence scan over the cascade's D = 2*n_stages-dim state z:
//     z[n] = F z[n-1] + g x[n]
// A lane owns a SEG = 64-sample segment, a workgroup is ONE wave and owns a tile of 64 segments: no workgroup
// barrier anywhere in K1 / K3, every scan over a tile is a shuffle scan.
//   K1  zero-state end vector of every segment as a 64-tap "matrix FIR"
//           zs = sum_k (F^(63-k) g) x[k]          (independent FMAs, taps by scalar loads)
//       written out, then reduced over the tile by a Kogge-Stone shuffle scan with P^(2^d), P = F^64
//       -> the tile's zero-carry end vector
//   K2a groups of 256 tiles (one per lane) are scanned in parallel with Q^(2^d), Q = P^64 (Kogge-Stone
//       through LDS): X[tile] = state entering the tile if its group started from rest; group end vectors
//   K2b one wave per stream scans the group ends with (Q^256)^(2^d): gc[g] = state entering group g;
//       K3 rebuilds its tile carry as X[tile] + Q^i gc[g]
//   K3  the same shuffle scan over u = zs (+ P * tile carry on the first lane) gives the state after every
//       segment; shifted by one lane it is every lane's true start state.  The lane then runs the
//       recursion over its segment and writes the truncated int16 outputs.
//       The tile's global loads are in flight while the scan runs.
// That four-kernel scan has no communication between workgroups; it is the path of filters whose memory is too long
// for the single-pass kernel below (the default), and the one a call is repeated on when a single-pass launch gave up.
// Tiles travel through LDS (coalesced 16-byte global accesses on one side, one row of 64+4 dwords per
// lane on the other: lane t reading 16 bytes of row t touches banks 4t..4t+3 -- conflict-free).
// F, g, P^(2^d), P^i, Q^(2^d) and Q^i are built on the host in fp64 by simulating the cascade, once per filter
// (iir_plan_for keeps them on the device in a buffer the shim owns).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <new>
#include <vector>

#include "clhip_common.h"

#define IIR_MAX_STAGES 4
#define IIR_MAX_DIM (2 * IIR_MAX_STAGES)
#define IIR_SEG 64                         // the four-kernel scan: samples per lane segment (both rails per lane)
#define IIR_TILE 64                        // one wave per workgroup: no workgroup barrier anywhere in K1 / K3
#define IIR_K2_LANES 256
#define IIR_GROUP 256                      // tiles per K2a workgroup
#define IIR_MSZ (IIR_MAX_DIM * IIR_MAX_DIM) // matrices are stored 8x8, row-major, zero outside DxD

typedef __attribute__((address_space(4))) double cdouble_t;   // read-only tables: scalar (SMEM) loads when uniform
typedef __attribute__((address_space(1))) double gdouble_t;   // per-lane table reads: global (not flat) loads

struct IirCoef {
    int n_stages, dim;
    double b0[IIR_MAX_STAGES], b1[IIR_MAX_STAGES], b2[IIR_MAX_STAGES], a1[IIR_MAX_STAGES], a2[IIR_MAX_STAGES];
};

// one cascade step on a DF-II state (v1,v2 per stage); returns the output
template <int NS>
__host__ __device__ __forceinline__ double iir_step(const IirCoef &c, double *z, double in)
{
    double out = in;
#pragma unroll
    for (int s = 0; s < NS; s++) {
        // the feed-forward part of the old state does not wait for w: three dependent FMAs per stage, not five
        const double ff = __builtin_fma(c.b1[s], z[2 * s], c.b2[s] * z[2 * s + 1]);
        const double w = __builtin_fma(-c.a2[s], z[2 * s + 1], __builtin_fma(-c.a1[s], z[2 * s], out));
        out = __builtin_fma(c.b0[s], w, ff);
        z[2 * s + 1] = z[2 * s];
        z[2 * s] = w;
    }
    return out;
}

// (int16_t)(float)y with the x86 conversion semantics of the reference build (cvttss2si, low half):
// v_cvt_i32_f32 saturates where cvttss2si returns 0x80000000; the low 16 bits differ only for f >= 2^31
template <bool BOUNDED = false>
__device__ __forceinline__ uint32_t iir_to_i16(double y)
{
    // BOUNDED: the host has shown |y| < 2^31 for int16 inputs (32768 x the l1 gain of the cascade), so the overflow
    // branch of cvttss2si cannot be taken and v_cvt_i32_f32 alone gives the same low 16 bits
    if constexpr (BOUNDED) return (uint32_t)(int)(float)y & 0xFFFFu;
    const float f = (float)y;
    const int t = f < 2147483648.0f ? (int)f : 0;
    return (uint32_t)t & 0xFFFFu;
}

// out[r] (+)= sum_c m[r][c] v[c] over the DxD corner of an 8x8 table.  Every table is a power of the cascade's
// one-sample transition F, and F is block lower triangular in the state order [stage 0 | stage 1 | ...]: a stage sees
// its own two state words and, through its input, those of the stages before it, never a later one.  Entries with
// c > (r | 1) are exact zeros in every power (the host builds them by products of such matrices), so they are skipped:
// 24 FMAs instead of 36 for three biquads.
__host__ __device__ constexpr bool iir_mat_nonzero(int r, int c) { return c <= (r | 1); }

template <int D, bool ACC, class MP>
__device__ __forceinline__ void matvec(MP m, const double *v, double *out)
{
#pragma unroll
    for (int r = 0; r < D; r++) {
        double s = ACC ? out[r] : 0.0;
#pragma unroll
        for (int c = 0; c < D; c++)
            if (iir_mat_nonzero(r, c)) s = __builtin_fma(m[r * IIR_MAX_DIM + c], v[c], s);
        out[r] = s;
    }
}

// ---------------------------------------------------------------------------
// tile staging
// ---------------------------------------------------------------------------
#define IIR_PITCH (IIR_SEG + 4)             // dwords per lane row: 16-byte aligned rows, bank = 4 * lane
#define IIR_LDS_WORDS (IIR_TILE * IIR_PITCH)

#define IIR_NLD (IIR_SEG / 4)                // 16-byte pieces per lane per tile

// A whole, 16-byte aligned tile travels through registers: every global load is issued first (IIR_NLD x 16 bytes in
// flight per lane) and committed to LDS later.  A ragged or unaligned tile (the last one of a stream, or a stream
// that starts on an odd sample) is read word by word at commit time by a rolled loop that needs no registers to speak
// of; words beyond the stream read as zero.  n_left = samples of this stream from the tile start (>= 1).
__device__ __forceinline__ bool iir_tile_whole(const uint32_t *x, long n_left)      // workgroup-uniform
{
    return (((uintptr_t)x & 15) == 0) && n_left >= (long)IIR_TILE * IIR_SEG;
}

__device__ __forceinline__ void iir_tile_issue(const uint32_t *__restrict__ x, long n_left, u32x4 (&r)[IIR_NLD], int t)
{
    if (!iir_tile_whole(x, n_left)) return;
#pragma unroll
    for (int q = 0; q < IIR_NLD; q++) r[q] = *(const u32x4 *)(x + (q * IIR_TILE + t) * 4);
}

__device__ __forceinline__ void iir_tile_commit(const uint32_t *__restrict__ x, long n_left, const u32x4 (&r)[IIR_NLD], uint32_t *sm, int t)
{
    if (iir_tile_whole(x, n_left)) {
#pragma unroll
        for (int q = 0; q < IIR_NLD; q++) {
            const int i = (q * IIR_TILE + t) * 4;
            const int row = i / IIR_SEG, col = i % IIR_SEG;      // 4 consecutive words stay in one row
            *(u32x4 *)(sm + row * IIR_PITCH + col) = r[q];
        }
        return;
    }
#pragma unroll 1
    for (int q = 0; q < IIR_NLD; q++) {
        const int i = (q * IIR_TILE + t) * 4;
        u32x4 v = {0, 0, 0, 0};
        if (i < n_left) v.x = x[i];
        if (i + 1 < n_left) v.y = x[i + 1];
        if (i + 2 < n_left) v.z = x[i + 2];
        if (i + 3 < n_left) v.w = x[i + 3];
        *(u32x4 *)(sm + (i / IIR_SEG) * IIR_PITCH + i % IIR_SEG) = v;
    }
}

__device__ __forceinline__ void iir_tile_store(uint32_t *__restrict__ x, long n_left, const uint32_t *sm, int t)
{
    if ((((uintptr_t)x & 15) == 0) && n_left >= (long)IIR_TILE * IIR_SEG) {
#pragma unroll
        for (int q = 0; q < IIR_NLD; q++) {
            const int i = (q * IIR_TILE + t) * 4;
            *(u32x4 *)(x + i) = *(const u32x4 *)(sm + (i / IIR_SEG) * IIR_PITCH + i % IIR_SEG);
        }
        return;
    }
#pragma unroll 1
    for (int q = 0; q < IIR_NLD; q++) {
        const int i = (q * IIR_TILE + t) * 4;
        const u32x4 v = *(const u32x4 *)(sm + (i / IIR_SEG) * IIR_PITCH + i % IIR_SEG);
        if (i < n_left) x[i] = v.x;
        if (i + 1 < n_left) x[i + 1] = v.y;
        if (i + 2 < n_left) x[i + 2] = v.z;
        if (i + 3 < n_left) x[i + 3] = v.w;
    }
}

// Kogge-Stone inclusive scan for the recurrence  v_i <- v_i + M^(2^d) v_(i - 2^d),  pow2[d] = M^(2^d),
// leaving v_i = sum_{j<=i} M^(i-j) v_j (both rails: v = [I rail D | Q rail D]).
// wave_scan: the 64 lanes of a wave by shuffles (no LDS storage, no barrier).
template <int D>
__device__ __forceinline__ void wave_scan(double (&v)[2 * D], const cdouble_t *__restrict__ pow2, int lane)
{
#pragma unroll 1
    for (int d = 0; d < 6; d++) {
        double pv[2 * D];
#pragma unroll
        for (int k = 0; k < 2 * D; k++) pv[k] = __shfl_up(v[k], 1 << d, 64);
        if (lane >= (1 << d)) {
            const cdouble_t *m = pow2 + d * IIR_MSZ;
            matvec<D, true>(m, pv, v);
            matvec<D, true>(m, pv + D, v + D);
        }
    }
}

// tile_scan: a tile is one wave (IIR_TILE = 64): the shuffle scan is the whole scan.  `sh` = IIR_TILE rows of 2D+1
// doubles; unless LAST_ONLY (only lane 63's value is wanted) sh[t] receives lane t's result for its neighbour.
template <int D, bool LAST_ONLY>
__device__ __forceinline__ void tile_scan(double (&v)[2 * D], const cdouble_t *__restrict__ pow2, double *sh, int t)
{
    static_assert(IIR_TILE == 64, "one wave per tile");
    constexpr int RS = 2 * D + 1;
    wave_scan<D>(v, pow2, t);
    if (!LAST_ONLY) {
#pragma unroll
        for (int k = 0; k < 2 * D; k++) sh[t * RS + k] = v[k];
        __syncthreads();                     // single wave: orders the LDS writes before the neighbour's reads
    }
}

// the 256 lanes of K2 (one workgroup): plain LDS exchange every round
template <int D>
__device__ __forceinline__ void ks_scan256(double (&v)[2 * D], const cdouble_t *__restrict__ pow2, double *sh, int t)
{
    constexpr int RS = 2 * D + 1;
#pragma unroll 1
    for (int d = 0; d < 8; d++) {
#pragma unroll
        for (int k = 0; k < 2 * D; k++) sh[t * RS + k] = v[k];
        __syncthreads();
        const int src = t - (1 << d);
        if (src >= 0) {
            double pv[2 * D];
#pragma unroll
            for (int k = 0; k < 2 * D; k++) pv[k] = sh[src * RS + k];
            const cdouble_t *m = pow2 + d * IIR_MSZ;
            matvec<D, true>(m, pv, v);
            matvec<D, true>(m, pv + D, v + D);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// The segment's matrix FIR: zs += sum_k (F^(63-k) g) x[k] over one block of the lane's LDS row, taps by scalar
// (SMEM) loads.  SMEM returns out of order, so the only wait it has is "everything outstanding": taps loaded in
// the same stretch as their use expose the full scalar-cache latency once per sample (what hipcc emits for the
// plain loop: 2 waves per SIMD then run this phase at half the fp64 rate).  Here the taps of the NEXT pair of
// samples are requested right after the current pair's have been waited for -- the empty asm that names them is
// that wait -- and arrive while the current pair's 24 FMAs issue.
// ---------------------------------------------------------------------------
template <int D>
struct IirTaps2 { double g[2][D]; };        // taps of two consecutive samples (the later sample first in the table)

template <int D>
__device__ __forceinline__ void iir_taps_load(IirTaps2<D> &tp, const cdouble_t *g)   // g -> taps of the pair's FIRST sample
{
#pragma unroll
    for (int r = 0; r < D; r++) { tp.g[0][r] = g[r]; tp.g[1][r] = g[r - IIR_MAX_DIM]; }
}
template <int D>
__device__ __forceinline__ void iir_taps_wait(IirTaps2<D> &tp)
{
#pragma unroll
    for (int r = 0; r < D; r++) asm volatile("" : "+s"(tp.g[0][r]), "+s"(tp.g[1][r]));
    __builtin_amdgcn_sched_barrier(0);
}
template <int D>
__device__ __forceinline__ void iir_fir_pair(double *v, const IirTaps2<D> &tp, uint32_t w0, uint32_t w1)
{
    const double xi0 = (double)(int16_t)(w0 & 0xFFFF), xq0 = (double)(int16_t)(w0 >> 16);
    const double xi1 = (double)(int16_t)(w1 & 0xFFFF), xq1 = (double)(int16_t)(w1 >> 16);
#pragma unroll
    for (int r = 0; r < D; r++) {
        v[r] = __builtin_fma(tp.g[0][r], xi0, v[r]);
        v[D + r] = __builtin_fma(tp.g[0][r], xq0, v[D + r]);
    }
#pragma unroll
    for (int r = 0; r < D; r++) {
        v[r] = __builtin_fma(tp.g[1][r], xi1, v[r]);
        v[D + r] = __builtin_fma(tp.g[1][r], xq1, v[D + r]);
    }
}
// x = the lane's LDS row (IIR_SEG words); G = plan->G
template <int D>
__device__ __forceinline__ void iir_segment_fir(double *v, const uint32_t *x, const cdouble_t *G)
{
    constexpr int BLK = 16;
    IirTaps2<D> ta, tb;
    iir_taps_load<D>(ta, G + (IIR_SEG - 1) * IIR_MAX_DIM);
#pragma unroll 1
    for (int kb = 0; kb < IIR_SEG; kb += BLK) {
        u32x4 xr[BLK / 4];
#pragma unroll
        for (int k = 0; k < BLK / 4; k++) xr[k] = *(const u32x4 *)(x + kb + 4 * k);
#pragma unroll
        for (int k = 0; k < BLK / 4; k++) asm volatile("" : "+v"(xr[k]));          // the block's row reads are waited for here, once
        const cdouble_t *gb = G + (IIR_SEG - 1 - kb) * IIR_MAX_DIM;                // taps of sample kb
#pragma unroll
        for (int k = 0; k < BLK; k += 4) {
            iir_taps_wait<D>(ta);
            iir_taps_load<D>(tb, gb - (k + 2) * IIR_MAX_DIM);
            __builtin_amdgcn_sched_barrier(0);
            iir_fir_pair<D>(v, ta, xr[k / 4][0], xr[k / 4][1]);
            __builtin_amdgcn_sched_barrier(0);
            iir_taps_wait<D>(tb);
            // the pair after the block's last one: the next block's first pair, or (past the segment) a harmless re-read
            const cdouble_t *gn = (kb + k + 4 < IIR_SEG) ? gb - (k + 4) * IIR_MAX_DIM : G + IIR_MAX_DIM;
            iir_taps_load<D>(ta, gn);
            __builtin_amdgcn_sched_barrier(0);
            iir_fir_pair<D>(v, tb, xr[k / 4][2], xr[k / 4][3]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    iir_taps_wait<D>(ta);
}

#define RL_SEGS 32                         // the single-pass kernel: segments (lane pairs) per tile
struct IirRailTab {                        // per (filter, segment length)
    double pow2[10][IIR_MSZ];              // P^(2^d), P = F^SEG; [5] = Q = P^32; [5 + d] = Q^(2^d)
    double ptab[IIR_MSZ][RL_SEGS];         // ptab[8 r + c][m] = (P^m)[r][c]: entry-major, one load = 32 x 8 contiguous bytes
    int seg, horizon;                      // horizon in tiles of 32 x seg samples; 0 = none within RL_HMAX (or the output gain is unbounded)
};

struct IirPlan {
    IirCoef coef;
    double G[IIR_SEG][IIR_MAX_DIM];         // G[j] = F^j g
    double pow2[8][IIR_MSZ];                // P^(2^d), P = F^SEG                         } the four-kernel scan's
    double Q[IIR_MSZ];                      // P^TILE                                     } tables (64 x 64-sample
    double qpow2[14][IIR_MSZ];              // Q^(2^d); [8 + d] = (Q^256)^(2^d)           } tiles, both rails per
    double qpow[IIR_GROUP][IIR_MSZ];        // Q^i                                        } lane)
    IirRailTab rail[3];                     // the single-pass kernel's, per segment length (kRailSegs)
};


// K1: zero-state end vector per segment (written to ZS) and the tile's zero-carry end vector.
//   ZS[stream][seg][2D], tend[stream][tile][2D]
template <int NS>
__global__ __launch_bounds__(IIR_TILE) void iir_k1_kernel(const IirPlan *__restrict__ plan, const uint32_t *__restrict__ iq,
                                                         long stride, long n, long n_seg, long n_tiles,
                                                         double *__restrict__ ZS, double *__restrict__ tend)
{
    constexpr int D = 2 * NS, RS = 2 * D + 1;
    extern __shared__ __attribute__((aligned(16))) uint32_t iir_sm[];
    const int t = threadIdx.x;
    const long tile0 = (long)blockIdx.x * IIR_TILE * IIR_SEG;
    {
        u32x4 r[IIR_NLD];
        iir_tile_issue(iq + (long)blockIdx.y * stride + tile0, n - tile0, r, t);
        iir_tile_commit(iq + (long)blockIdx.y * stride + tile0, n - tile0, r, iir_sm, t);
    }
    __syncthreads();
    const cdouble_t *__restrict__ G = (const cdouble_t *)&plan->G[0][0];
    const uint32_t *x = iir_sm + t * IIR_PITCH;
    double v[2 * D];
#pragma unroll
    for (int k = 0; k < 2 * D; k++) v[k] = 0.0;
#pragma unroll 2
    for (int k = 0; k < IIR_SEG; k += 4) {                  // (many short-lived waves per SIMD hide the tap loads here)
        const u32x4 w = *(const u32x4 *)(x + k);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const double xi = (double)(int16_t)(w[j] & 0xFFFF), xq = (double)(int16_t)(w[j] >> 16);
            const cdouble_t *g = G + (IIR_SEG - 1 - (k + j)) * IIR_MAX_DIM;
#pragma unroll
            for (int r = 0; r < D; r++) {
                const double gr = g[r];
                v[r] = __builtin_fma(gr, xi, v[r]);
                v[D + r] = __builtin_fma(gr, xq, v[D + r]);
            }
        }
    }
    const long seg = (long)blockIdx.x * IIR_TILE + t;
    if (seg < n_seg) {
        double *o = ZS + ((long)blockIdx.y * n_seg + seg) * 2 * D;
#pragma unroll
        for (int k = 0; k < 2 * D; k++) o[k] = v[k];
    }
    __syncthreads();                         // the staged tile is dead: its LDS carries the scan exchange
    tile_scan<D, true>(v, (const cdouble_t *)&plan->pow2[0][0], (double *)iir_sm, t);
    if (t == IIR_TILE - 1) {
        double *o = tend + ((long)blockIdx.y * n_tiles + blockIdx.x) * 2 * D;
#pragma unroll
        for (int k = 0; k < 2 * D; k++) o[k] = v[k];
    }
    (void)RS;
}

// K2a: scan inside each group of 256 tiles.  X[stream][tile][2D] (exclusive, zero carry-in),
// gend[stream][group][2D] (inclusive result of the group's last tile)
template <int NS>
__global__ __launch_bounds__(IIR_GROUP) void iir_k2a_kernel(const IirPlan *__restrict__ plan, long n_tiles, long n_groups,
                                                           const double *__restrict__ tend, double *__restrict__ X,
                                                           double *__restrict__ gend)
{
    constexpr int D = 2 * NS, RS = 2 * D + 1;
    __shared__ double sh[IIR_GROUP * RS];
    const int t = threadIdx.x;
    const long tile = (long)blockIdx.x * IIR_GROUP + t;
    const double *te = tend + ((long)blockIdx.y * n_tiles + tile) * 2 * D;
    double v[2 * D];
#pragma unroll
    for (int k = 0; k < 2 * D; k++) v[k] = tile < n_tiles ? te[k] : 0.0;
    ks_scan256<D>(v, (const cdouble_t *)&plan->qpow2[0][0], sh, t);
#pragma unroll
    for (int k = 0; k < 2 * D; k++) sh[t * RS + k] = v[k];
    __syncthreads();
    if (tile < n_tiles) {
        double *o = X + ((long)blockIdx.y * n_tiles + tile) * 2 * D;
#pragma unroll
        for (int k = 0; k < 2 * D; k++) o[k] = t == 0 ? 0.0 : sh[(t - 1) * RS + k];
    }
    if (t == IIR_GROUP - 1) {
        double *o = gend + ((long)blockIdx.y * n_groups + blockIdx.x) * 2 * D;
#pragma unroll
        for (int k = 0; k < 2 * D; k++) o[k] = v[k];
    }
}

// K2b: gc[stream][group][2D] = state entering the group.  One wave per stream: lane g holds the end vector of group
// base + g (lane 0 also takes QG * carry-in), one shuffle scan with QG^(2^d), QG = Q^256, and the exclusive
// shift is the answer; 64 groups (2^26 samples) per round.
template <int NS>
__global__ __launch_bounds__(64) void iir_k2b_kernel(const IirPlan *__restrict__ plan, long n_groups,
                                                    const double *__restrict__ gend, double *__restrict__ gc,
                                                    const double *__restrict__ state)
{
    constexpr int D = 2 * NS;
    const int s = blockIdx.x, lane = threadIdx.x;
    const cdouble_t *__restrict__ qg = (const cdouble_t *)&plan->qpow2[8][0];          // QG^(2^d) = qpow2[8 + d]
    const double *ge = gend + (long)s * n_groups * 2 * D;
    double *o = gc + (long)s * n_groups * 2 * D;
    double carry[2 * D];
#pragma unroll
    for (int k = 0; k < 2 * D; k++) carry[k] = state[(long)s * 2 * IIR_MAX_DIM + (k / D) * IIR_MAX_DIM + (k % D)];
    for (long base = 0; base < n_groups; base += 64) {
        const long g = base + lane;
        double v[2 * D];
#pragma unroll
        for (int k = 0; k < 2 * D; k++) v[k] = g < n_groups ? ge[g * 2 * D + k] : 0.0;
        if (lane == 0) {
            matvec<D, true>(qg, carry, v);
            matvec<D, true>(qg, carry + D, v + D);
        }
        wave_scan<D>(v, qg, lane);
#pragma unroll
        for (int k = 0; k < 2 * D; k++) {
            const double prev = __shfl_up(v[k], 1, 64);
            if (g < n_groups) o[g * 2 * D + k] = lane == 0 ? carry[k] : prev;
            carry[k] = __shfl(v[k], 63, 64);
        }
    }
}

// two consecutive samples through the cascade, stage by stage: the state is written once per pair
// ((v1, v2) <- (w_B, w_A)), so nothing is shifted between samples
// B121: the host has seen b = (1, 2, 1) exactly in every stage after the first (what a Butterworth / Chebyshev low-pass
// design puts there; the gain sits in stage 0).  fma(2, z0, 1 * z1) and fma(1, w, ff) round exactly like 2 z0 + z1 and
// w + ff, so those stages take four operations instead of five with bit-identical results.
template <int NS, bool B121 = false>
__device__ __forceinline__ void iir_step2(const IirCoef &c, double *z, double in0, double in1, double &out0, double &out1)
{
    double o0 = in0, o1 = in1;
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const double z0 = z[2 * s], z1 = z[2 * s + 1];
        const bool unit = B121 && s > 0;
        const double ffa = unit ? __builtin_fma(2.0, z0, z1) : __builtin_fma(c.b1[s], z0, c.b2[s] * z1);
        const double wa = __builtin_fma(-c.a2[s], z1, __builtin_fma(-c.a1[s], z0, o0));
        o0 = unit ? wa + ffa : __builtin_fma(c.b0[s], wa, ffa);
        const double ffb = unit ? __builtin_fma(2.0, wa, z0) : __builtin_fma(c.b1[s], wa, c.b2[s] * z0);
        const double wb = __builtin_fma(-c.a2[s], z0, __builtin_fma(-c.a1[s], wa, o1));
        o1 = unit ? wb + ffb : __builtin_fma(c.b0[s], wb, ffb);
        z[2 * s] = wb;
        z[2 * s + 1] = wa;
    }
    out0 = o0; out1 = o1;
}

template <int NS, bool FULL, bool BOUNDED = false, bool B121 = false>
__device__ __forceinline__ void iir_k3_segment(const IirCoef &c, uint32_t *x, long cnt, double *zi, double *zq)
{
#pragma unroll 2
    for (int k = 0; k < IIR_SEG; k += 4) {
        if (!FULL && k >= cnt) break;
        u32x4 w = *(const u32x4 *)(x + k);
        if (FULL || k + 4 <= cnt) {
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
                // filter((float)x): int16 -> float -> double is exact
                double yi0, yi1, yq0, yq1;
                iir_step2<NS, B121>(c, zi, (double)(int16_t)(w[j] & 0xFFFF), (double)(int16_t)(w[j + 1] & 0xFFFF), yi0, yi1);
                iir_step2<NS, B121>(c, zq, (double)(int16_t)(w[j] >> 16), (double)(int16_t)(w[j + 1] >> 16), yq0, yq1);
                w[j] = iir_to_i16<BOUNDED>(yi0) | (iir_to_i16<BOUNDED>(yq0) << 16);
                w[j + 1] = iir_to_i16<BOUNDED>(yi1) | (iir_to_i16<BOUNDED>(yq1) << 16);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (k + j < cnt) {
                    const double yi = iir_step<NS>(c, zi, (double)(int16_t)(w[j] & 0xFFFF));
                    const double yq = iir_step<NS>(c, zq, (double)(int16_t)(w[j] >> 16));
                    w[j] = iir_to_i16<BOUNDED>(yi) | (iir_to_i16<BOUNDED>(yq) << 16);
                }
            }
        }
        *(u32x4 *)(x + k) = w;
    }
}

template <int NS>
__global__ __launch_bounds__(IIR_TILE) void iir_k3_kernel(const IirPlan *__restrict__ plan, IirCoef c, const uint32_t *in, uint32_t *out,
                                                         long stride, long n, long n_seg, long n_tiles, long n_groups,
                                                         const double *__restrict__ ZS, const double *__restrict__ X,
                                                         const double *__restrict__ gc, double *__restrict__ state)
{
    constexpr int D = 2 * NS, RS = 2 * D + 1;
    extern __shared__ __attribute__((aligned(16))) uint32_t iir_sm[];
    const int t = threadIdx.x;
    const long tile0 = (long)blockIdx.x * IIR_TILE * IIR_SEG;
    const uint32_t *xt = in + (long)blockIdx.y * stride + tile0;
    uint32_t *xo = out + (long)blockIdx.y * stride + tile0;
    u32x4 raw[IIR_NLD];
    iir_tile_issue(xt, n - tile0, raw, t);                   // in flight while the start states are computed
    const long seg = (long)blockIdx.x * IIR_TILE + t;
    const cdouble_t *pow2 = (const cdouble_t *)&plan->pow2[0][0];
    double u[2 * D], cv[2 * D];
    {   // state entering the tile = X[tile] + Q^i gc[group], i = tile index inside its group (all uniform)
        const int g = (int)(blockIdx.x / IIR_GROUP), gi = (int)(blockIdx.x % IIR_GROUP);
        const cdouble_t *xx = (const cdouble_t *)(X + ((long)blockIdx.y * n_tiles + blockIdx.x) * 2 * D);
        const cdouble_t *gg = (const cdouble_t *)(gc + ((long)blockIdx.y * n_groups + g) * 2 * D);
        const cdouble_t *qi = (const cdouble_t *)&plan->qpow[gi][0];
        double gv[2 * D];
#pragma unroll
        for (int k = 0; k < 2 * D; k++) { cv[k] = xx[k]; gv[k] = gg[k]; }
        matvec<D, true>(qi, gv, cv);
        matvec<D, true>(qi, gv + D, cv + D);
    }
    {
        const double *z = ZS + ((long)blockIdx.y * n_seg + (seg < n_seg ? seg : n_seg - 1)) * 2 * D;
#pragma unroll
        for (int k = 0; k < 2 * D; k++) u[k] = seg < n_seg ? z[k] : 0.0;
    }
    if (t == 0) {                                            // the tile's carry enters through the first segment
        matvec<D, true>(pow2, cv, u);
        matvec<D, true>(pow2, cv + D, u + D);
    }
    double *sh = (double *)iir_sm;                           // the tile region is still empty
    tile_scan<D, false>(u, pow2, sh, t);                     // u = state after the lane's segment; sh[t] = the same
    double zi[D], zq[D];
#pragma unroll
    for (int k = 0; k < D; k++) {
        zi[k] = t == 0 ? cv[k] : sh[(t - 1) * RS + k];
        zq[k] = t == 0 ? cv[D + k] : sh[(t - 1) * RS + D + k];
    }
    __syncthreads();                                         // exchange rows are read: the tile may land
    iir_tile_commit(xt, n - tile0, raw, iir_sm, t);
    __syncthreads();
    if (seg < n_seg) {
        uint32_t *x = iir_sm + t * IIR_PITCH;
        if (tile0 + (long)IIR_TILE * IIR_SEG <= n) iir_k3_segment<NS, true>(c, x, IIR_SEG, zi, zq);
        else {
            const long cnt = n - seg * IIR_SEG < IIR_SEG ? n - seg * IIR_SEG : IIR_SEG;
            iir_k3_segment<NS, false>(c, x, cnt, zi, zq);
        }
        if (seg == n_seg - 1) {
            double *st = state + (long)blockIdx.y * 2 * IIR_MAX_DIM;
#pragma unroll
            for (int k = 0; k < D; k++) { st[k] = zi[k]; st[IIR_MAX_DIM + k] = zq[k]; }
        }
    }
    __syncthreads();
    iir_tile_store(xo, n - tile0, iir_sm, t);
}

// ---------------------------------------------------------------------------
// Single pass (default): one launch, every output written once, every input read once (+ the chunk tails, below).
//
// Shape: the two rails of a segment sit on NEIGHBOURING LANES -- lane t = (segment t >> 1, rail t & 1) -- so a lane
// carries D doubles of state where a lane that owns both rails carries 2D, a tile is 32 segments x SEG samples
// (8.5 KB of LDS at SEG = 64) and the kernel fits four waves per SIMD at full-length segments: fp64 issue needs
// that many waves to approach its rate (tools/microbench/fir64_phase.hip: 15 T DFMA-lanes/s at one wave per SIMD, 20 at
// two, 25-31 at four to eight).  Both lanes of a pair read the same LDS words (a broadcast), each takes its half;
// each lane writes its own (int16)(float) results back into the row with 16-bit LDS stores.
//
// Per tile: coalesced load -> LDS rows -> every segment's zero-state end vector as a SEG-tap matrix FIR (taps by
// scalar loads, double-buffered in SGPRs) -> Kogge-Stone shuffle scan over the 32 segments with P^(2^d), P = F^SEG ->
// lane start state = exclusive scan value + P^m cv (cv = the state entering the tile, P^m from a per-segment table)
// -> DF-II recursion two samples per cascade pass over the LDS row -> coalesced store.
//
// Ownership: a wave takes CHUNKS of consecutive tiles of one stream (as many as share the call evenly over the
// resident waves: 8 for 2^26 samples, 1 for a native batch) and walks a chunk in order, so cv of every tile but the
// chunk's first is simply the state the wave's own recursion has just left in lane pair 31 -- exact, nothing to wait
// for, no lock step between waves (waves that free-run drift apart, so the loads of one overlap the recursion of
// another; waves that wait for each other every tile all load, compute and store at the same time).  The state
// entering a chunk's FIRST tile comes from other waves:
//     cv = sum_{k=1..H} Q^(k-1) aggregate(tile - k)     (the carried state stands in for tile -1),   Q = P^32
// where an AGGREGATE is a tile's zero-carry end vector (the scan's last value) and H the horizon: the number of tiles
// after which ANY reachable state has decayed below 1e-18 absolute (host: iir_rail_tab_build).  Lane pair h fetches
// aggregate(tile-1-h); H <= 6 folds them by Horner, longer horizons (short tiles of a narrow filter) by a log-step
// tree with Q^(2^d).  The aggregates a chunk's successor will ask for -- those of the chunk's last H tiles -- are the
// first thing a wave produces (prologue: load, segment FIR, scan, publish; nothing waited for; the chunk tail is read
// twice: H / chunk of the input), so a wave that asks finds them there or on their way.
//
// Order and progress: chunks are handed out by atomic ticket from one of 64 counters (class = block index mod 64;
// chunk = class + 64 * ticket; counters 4 KB apart -- atomics to one 256-byte block retire one at a time); the next
// ticket is requested at the top of a chunk, so its round trip is never waited for.  A chunk's predecessors have
// smaller indices and every chunk that has been taken publishes without waiting, so the smallest chunk nobody has
// taken is the only thing the launch can be waiting for, and it is taken as soon as a wave of its class finishes what
// it holds: the launch moves as long as one wave of every class is running -- the first 64 workgroups dispatched --
// whatever else shares the GPU and however many of the launched waves are resident.  Polls are bounded all the same:
// one that gives up zeroes what it did not get (no NaN reaches the state), raises the object's pinned overrun word and
// a device-side abort flag that ends every other poll of the launch; the host rolls the call back.
//
// Priority: waves of a SIMD issue by priority, then age, so four equal shares end far apart (rail_set_priority).
//
// Publication: an aggregate is D doubles per rail written with relaxed agent-scope 64-bit atomic stores into slots
// pre-set to all-ones (a NaN no aggregate can be): a reader polls until none of its D words is the sentinel, so no
// flag, no fence and no store ordering is needed.  Two aggregate buffers alternate between launches and every launch
// re-arms the one it does not use (each wave a share), so a call is ONE launch: no memset.  The counters are put
// back to zero by the last wave out.  The carried state is ping-pong (read from one half, written to the other).
// ---------------------------------------------------------------------------
#define IIR_SENTINEL 0xFFFFFFFFFFFFFFFFull
#define RL_HMAX 32                         // a lane pair per predecessor
#define RL_HORNER_MAX 6                    // longer horizons fold by the log-step tree
#define RL_CLASSES 64
#ifndef RL_CTL_STRIDE
#define RL_CTL_STRIDE 1024                 // words between control words: 4 KB apart, each in a memory channel of its own (atomics to
#endif                                     // ONE 256-byte block retire one at a time whatever their addresses: 32 768 of them, 0.19 ms)
#define RL_CTL_EXIT (64 * RL_CTL_STRIDE)   // ctl[c * stride]: class counters; then: waves that have left, abort flag
#define RL_CTL_ABORT (65 * RL_CTL_STRIDE)
#define RL_CTL_WORDS (66 * RL_CTL_STRIDE)
#ifndef RL_WAVES
#define RL_WAVES 4                         // waves per SIMD the register budget is cut for
#endif
#ifndef RL_PREFETCH
#define RL_PREFETCH 1                      // the next tile's words wait in registers during the recursion
#endif

struct IirRailArgs {
    const double *G;                       // [64][8]: G[j] = F^j g
    const IirRailTab *tab;
    IirCoef c;
    const uint32_t *in;
    uint32_t *out;
    long stride, n, n_seg, n_tiles;
    int n_streams, horizon;
    int chunk_tiles;                       // consecutive tiles of one stream a wave takes per ticket
    long chunks_per_stream;
    unsigned int *ctl;
    unsigned long long *agg;               // this launch's aggregates: [stream][tile][rail][D], all-ones on entry
    unsigned long long *agg_other;         // the other buffer and how much of it the launches before this one used
    long other_words;
    const double *state_in;
    double *state_out;
    unsigned int *overrun;                 // the object's pinned host word (device address)
    int poll_bound, n_classes, dynamic, prio;
    int in_sh0, in_sh1, in_width;          // where the two rails sit in an input word: CS16 {0, 16, 16}; raw SMI words {17, 1, 13} (S1G) / {1, 17, 13} (HiF)
    unsigned long long *stamps;            // diagnostics (CLHIP_IIR_STAMPS=1): [RL_STAMP_WAVES][RL_STAMP_TILES][RL_STAMP_PHASES] of s_memrealtime
};
#define RL_STAMP_WAVES 64
#define RL_STAMP_TILES 16
#define RL_STAMP_PHASES 12
#define RL_STAMP_ALLWAVES 8192             // behind the phase table: [wave][start, end, tiles done | HW_ID | XCC_ID, ticks spent polling for first-tile aggregates, first chunk | class << 32]
#define RL_STAMP_WORDS 5
#define RL_STAMP(p) do { if (A.stamps && blockIdx.x < RL_STAMP_WAVES && it < RL_STAMP_TILES && t0 == 0) \
        A.stamps[((size_t)blockIdx.x * RL_STAMP_TILES + it) * RL_STAMP_PHASES + (p)] = __builtin_amdgcn_s_memrealtime(); } while (0)

// zs += sum_k (F^(SEG-1-k) g) x[k] for the lane's rail over its LDS row; taps as in iir_segment_fir
template <int D>
__device__ __forceinline__ void rail_fir_pair(double *v, const IirTaps2<D> &tp, uint32_t w0, uint32_t w1, int sh, int wd)
{
    const double x0 = (double)(int)__builtin_amdgcn_sbfe((int)w0, sh, wd), x1 = (double)(int)__builtin_amdgcn_sbfe((int)w1, sh, wd);
#pragma unroll
    for (int r = 0; r < D; r++) v[r] = __builtin_fma(tp.g[0][r], x0, v[r]);
#pragma unroll
    for (int r = 0; r < D; r++) v[r] = __builtin_fma(tp.g[1][r], x1, v[r]);
}

template <int D, int SEG>
__device__ __forceinline__ void rail_segment_fir(double *v, const uint32_t *x, const cdouble_t *G, int sh, int wd)
{
    constexpr int BLK = 16;
    static_assert(SEG % BLK == 0, "segments are whole blocks");
    IirTaps2<D> ta, tb;
    iir_taps_load<D>(ta, G + (SEG - 1) * IIR_MAX_DIM);
#pragma unroll 1
    for (int kb = 0; kb < SEG; kb += BLK) {
        u32x4 xr[BLK / 4];
#pragma unroll
        for (int k = 0; k < BLK / 4; k++) xr[k] = *(const u32x4 *)(x + kb + 4 * k);
#pragma unroll
        for (int k = 0; k < BLK / 4; k++) asm volatile("" : "+v"(xr[k]));
        const cdouble_t *gb = G + (SEG - 1 - kb) * IIR_MAX_DIM;
#pragma unroll
        for (int k = 0; k < BLK; k += 4) {
            iir_taps_wait<D>(ta);
            iir_taps_load<D>(tb, gb - (k + 2) * IIR_MAX_DIM);
            __builtin_amdgcn_sched_barrier(0);
            rail_fir_pair<D>(v, ta, xr[k / 4][0], xr[k / 4][1], sh, wd);
            __builtin_amdgcn_sched_barrier(0);
            iir_taps_wait<D>(tb);
            const cdouble_t *gn = (kb + k + 4 < SEG) ? gb - (k + 4) * IIR_MAX_DIM : G + IIR_MAX_DIM;
            iir_taps_load<D>(ta, gn);
            __builtin_amdgcn_sched_barrier(0);
            rail_fir_pair<D>(v, tb, xr[k / 4][2], xr[k / 4][3], sh, wd);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    iir_taps_wait<D>(ta);
}

// The recursion of one rail over the lane's row, in place: a lane reads the pair's packed words 16 bytes at a time and
// writes its own 16-bit results back with ds_write_b16 at constant offsets from its row (+2 bytes for the Q rail) -- the
// write goes down the LDS pipe, where there is room, instead of a DPP move + v_perm_b32 per sample on the VALU, which
// is the unit this kernel is bound by.  Both lanes of a pair have read a 16-byte group before either writes into it
// (one wave, LDS operations in order).
template <int NS, int SEG, bool FULL, bool B121>
__device__ __forceinline__ void rail_recursion(const IirCoef &c, uint32_t *x, long cnt, double *z, int sh, int wd, int rail)
{
    uint16_t *xo = (uint16_t *)x + rail;
#pragma unroll 2
    for (int k = 0; k < SEG; k += 4) {
        if (!FULL && k >= cnt) break;
        const u32x4 w = *(const u32x4 *)(x + k);
        if (FULL || k + 4 <= cnt) {
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
                double y0, y1;
                iir_step2<NS, B121>(c, z, (double)(int)__builtin_amdgcn_sbfe((int)w[j], sh, wd), (double)(int)__builtin_amdgcn_sbfe((int)w[j + 1], sh, wd), y0, y1);
                xo[2 * (k + j)] = (uint16_t)(uint32_t)(int)(float)y0;            // (the low half: cvttss2si's, see iir_to_i16)
                xo[2 * (k + j + 1)] = (uint16_t)(uint32_t)(int)(float)y1;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (k + j < cnt)                                        // uniform over the pair
                    xo[2 * (k + j)] = (uint16_t)(uint32_t)(int)(float)iir_step<NS>(c, z, (double)(int)__builtin_amdgcn_sbfe((int)w[j], sh, wd));
            }
        }
    }
}

template <int SEG>
__device__ __forceinline__ bool rail_tile_whole(const uint32_t *in, const uint32_t *out, long n_left)      // wave-uniform
{
    return ((((uintptr_t)in | (uintptr_t)out) & 15) == 0) && n_left >= (long)RL_SEGS * SEG;
}

// (a tile that is not whole is read word by word at commit time; its registers are DEFINED all the same, so that the
// compiler does not see last tile's values flow round the persistent loop and keep 4 x SEG / 8 registers live through it)
template <int SEG>
__device__ __forceinline__ void rail_tile_issue(const uint32_t *__restrict__ x, bool whole, u32x4 (&r)[SEG / 8], int t)
{
    if (whole) {
#pragma unroll
        for (int q = 0; q < SEG / 8; q++) r[q] = *(const u32x4 *)(x + (q * 64 + t) * 4);
    } else {
#pragma unroll
        for (int q = 0; q < SEG / 8; q++) r[q] = u32x4{0, 0, 0, 0};
    }
}

template <int SEG>
__device__ __forceinline__ void rail_tile_commit(const uint32_t *__restrict__ x, bool whole, long n_left, const u32x4 (&r)[SEG / 8], uint32_t *sm, int t)
{
    constexpr int PITCH = SEG + 4;
    if (whole) {
#pragma unroll
        for (int q = 0; q < SEG / 8; q++) {
            const int i = (q * 64 + t) * 4;
            *(u32x4 *)(sm + (i / SEG) * PITCH + i % SEG) = r[q];
        }
        return;
    }
#pragma unroll 1
    for (int q = 0; q < SEG / 8; q++) {                          // ragged or unaligned: word by word, zeros past the end
        const int i = (q * 64 + t) * 4;
        u32x4 v = {0, 0, 0, 0};
        if (i < n_left) v.x = x[i];
        if (i + 1 < n_left) v.y = x[i + 1];
        if (i + 2 < n_left) v.z = x[i + 2];
        if (i + 3 < n_left) v.w = x[i + 3];
        *(u32x4 *)(sm + (i / SEG) * PITCH + i % SEG) = v;
    }
}

template <int SEG>
__device__ __forceinline__ void rail_tile_store(uint32_t *__restrict__ x, bool whole, long n_left, const uint32_t *sm, int t)
{
    constexpr int PITCH = SEG + 4;
    if (whole) {
#pragma unroll
        for (int q = 0; q < SEG / 8; q++) {
            const int i = (q * 64 + t) * 4;
            *(u32x4 *)(x + i) = *(const u32x4 *)(sm + (i / SEG) * PITCH + i % SEG);
        }
        return;
    }
#pragma unroll 1
    for (int q = 0; q < SEG / 8; q++) {
        const int i = (q * 64 + t) * 4;
        const u32x4 v = *(const u32x4 *)(sm + (i / SEG) * PITCH + i % SEG);
        if (i < n_left) x[i] = v.x;
        if (i + 1 < n_left) x[i + 1] = v.y;
        if (i + 2 < n_left) x[i + 2] = v.z;
        if (i + 3 < n_left) x[i + 3] = v.w;
    }
}

// Waves of a SIMD issue by priority, then age: at equal priority the oldest wave of four runs nearly unimpeded and the
// youngest gets what is left, so equal shares of work end far apart (measured: the first wave of a SIMD leaves after
// 100 us, the last after 211) and the SIMD spends its last third under-occupied.  A wave therefore lowers its priority as it
// advances through its share (3 -> 0): whoever is behind issues first, and the four finish together.
__device__ __forceinline__ void rail_set_priority(int done, int of)
{
    const int q = (4 * done) / (of > 0 ? of : 1);
    if (q <= 0) __builtin_amdgcn_s_setprio(3);
    else if (q == 1) __builtin_amdgcn_s_setprio(2);
    else if (q == 2) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
}

__device__ __forceinline__ unsigned rail_tile_of(unsigned cls, unsigned nc, unsigned ticket, unsigned total)
{
    const unsigned long long t = (unsigned long long)cls + (unsigned long long)nc * ticket;
    return t < total ? (unsigned)t : total;
}

template <int NS, int SEG, bool B121>
__global__ __launch_bounds__(64, RL_WAVES) void iir_rail_kernel(const IirRailArgs A)
{
    constexpr int D = 2 * NS, PITCH = SEG + 4, NLD = SEG / 8;
    constexpr long TILE = (long)RL_SEGS * SEG;
    extern __shared__ __attribute__((aligned(16))) uint32_t iir_sm[];
    const int t0 = threadIdx.x;
    const unsigned ns = (unsigned)A.n_streams, C = (unsigned)A.chunk_tiles;
    const unsigned total = (unsigned)A.chunks_per_stream * ns;         // chunks of the launch (< 2^31, host-checked)
    const unsigned NC = (unsigned)A.n_classes;
    const unsigned cls = blockIdx.x & (NC - 1);

    // re-arm the aggregate buffer the NEXT launch will use (this launch does not touch it otherwise)
    for (long i = (long)blockIdx.x * 64 + t0; i < A.other_words; i += (long)gridDim.x * 64) A.agg_other[i] = IIR_SENTINEL;

    if (A.stamps && t0 == 0 && blockIdx.x < RL_STAMP_ALLWAVES)
        A.stamps[(size_t)RL_STAMP_WAVES * RL_STAMP_TILES * RL_STAMP_PHASES + RL_STAMP_WORDS * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    unsigned int tk = 0;
    if (t0 == 0) tk = atomicAdd(A.ctl + cls * RL_CTL_STRIDE, 1u);
    // (a ticket beyond the launch's chunks saturates: more tickets than chunks are only ever taken by waves on their way out)
    unsigned T = rail_tile_of(cls, NC, (unsigned)__builtin_amdgcn_readfirstlane((int)tk), total);
    const unsigned NW = gridDim.x;
    // (diagnostics; written at once so that nothing of it stays live: this kernel has no scalar register to spare)
    if (A.stamps && t0 == 0 && blockIdx.x < RL_STAMP_ALLWAVES) {
        A.stamps[(size_t)RL_STAMP_WAVES * RL_STAMP_TILES * RL_STAMP_PHASES + RL_STAMP_WORDS * blockIdx.x + 3] = 0;
        A.stamps[(size_t)RL_STAMP_WAVES * RL_STAMP_TILES * RL_STAMP_PHASES + RL_STAMP_WORDS * blockIdx.x + 4] = (unsigned long long)T | ((unsigned long long)cls << 32);
    }
    constexpr bool PF = RL_PREFETCH && NS < 4;      // (four biquads: the start-state phase needs the registers)
    int it = -1;
    while (T < total) {
        // ---- one chunk: tiles b0..b1 of stream s, taken in order by this wave alone.  The state entering the chunk's
        // first tile comes from the H tiles before it (other waves' -- their aggregates, below); every later tile starts
        // from the exact state the wave itself has just left.  The aggregates the NEXT chunk's owner will look for, those of
        // this chunk's last H tiles, are made first of all (prologue: load, segment FIR, scan, publish -- nothing waited
        // for), so that nobody ever waits long for them.
        unsigned int tkn = 0;
        {
            uintptr_t ca = (uintptr_t)(A.ctl + cls * RL_CTL_STRIDE);
            asm volatile("" : "+v"(ca));
            // the next chunk's ticket: back long before it is needed.  (The counter's address goes through a register the
            // compiler cannot see through: with a uniform address its atomic optimizer rewrites the add as a wave reduction
            // whose result it broadcasts -- and waits for, with everything else in flight -- right here.)
            if (A.dynamic && t0 == 0)
                tkn = __hip_atomic_fetch_add((__attribute__((address_space(1))) unsigned int *)ca, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const long s = T % ns;
        const long b0 = (long)(T / ns) * C;
        const long b1 = (b0 + C < A.n_tiles ? b0 + C : A.n_tiles) - 1;
        const int H = A.horizon;
        // prologue tiles: those of the chunk's last H that come after its first tile (the stream's last chunk has no
        // successor: none); the first tile, when it is among the last H (chunks no longer than the horizon -- small
        // calls: one tile per wave), publishes in its own full pass, before it waits for anything
        const bool has_succ = b1 + 1 < A.n_tiles;
        const long pa = b1 + 1 - H > b0 + 1 ? b1 + 1 - H : b0 + 1;     // first prologue tile
        const int np = has_succ && pa <= b1 ? (int)(b1 - pa + 1) : 0;
        const bool first_publishes = has_succ && b0 >= b1 + 1 - H;
        const uint32_t *sin = A.in + s * A.stride;
        uint32_t *sout = A.out + s * A.stride;
        u32x4 raw[NLD];
        auto tile_whole = [&](long b) { return rail_tile_whole<SEG>(sin + b * TILE, sout + b * TILE, A.n - b * TILE); };
        if constexpr (PF) {
            const long b = np ? pa : b0;
            rail_tile_issue<SEG>(sin + b * TILE, tile_whole(b), raw, t0);
        }
        // rows <- tile b; v <- after the scan, the state the lane's segment leaves when the tile is entered at rest.
        // bn (when have_next) is the tile the NEXT step works on: a prologue step requests its words as soon as its own
        // have left the registers.
        auto rows_fir_scan = [&](long b, bool have_next, long bn, bool issue_next_now, int t, const cdouble_t *G, const cdouble_t *pow2, double (&v)[D]) {
            const int m = t >> 1, sh = (t & 1) ? A.in_sh1 : A.in_sh0;
            const uint32_t *xin = sin + b * TILE;
            const bool whole = tile_whole(b);
            if constexpr (!PF) rail_tile_issue<SEG>(xin, whole, raw, t);
            rail_tile_commit<SEG>(xin, whole, A.n - b * TILE, raw, iir_sm, t);
            __syncthreads();
            if constexpr (PF) {
                if (issue_next_now) rail_tile_issue<SEG>(sin + bn * TILE, have_next && tile_whole(bn), raw, t);
            }
            RL_STAMP(1);
#pragma unroll
            for (int k = 0; k < D; k++) v[k] = 0.0;
            rail_segment_fir<D, SEG>(v, iir_sm + m * PITCH, G, sh, A.in_width);
            RL_STAMP(2);
            {
                // Kogge-Stone over the 32 segments of the rail: v_m <- v_m + P^(2^d) v_(m - 2^d)
#pragma unroll 1
                for (int d = 0; d < 5; d++) {
                    double pv[D];
#pragma unroll
                    for (int k = 0; k < D; k++) pv[k] = __shfl_up(v[k], 2 << d, 64);
                    if (m >= (1 << d)) matvec<D, true>(pow2 + d * IIR_MSZ, pv, v);
                }
            }
            RL_STAMP(3);
        };
        auto publish = [&](long b, int t, const double (&v)[D]) {
            if ((t >> 1) == RL_SEGS - 1) {
                unsigned long long *mine = A.agg + ((s * A.n_tiles + b) * 2 + (t & 1)) * D;
#pragma unroll
                for (int k = 0; k < D; k++)
                    __hip_atomic_store(mine + k, __builtin_bit_cast(unsigned long long, v[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        };
        // ---- prologue
        const int nsteps = np + (int)(b1 - b0 + 1);
#pragma unroll 1
        for (int i = 0; i < np; i++) {
            it++;
            if (A.prio) rail_set_priority(i, nsteps);
            RL_STAMP(0);
            int t = t0;
            const IirRailTab *tab = A.tab;
            const double *Gp = A.G;
            asm volatile("" : "+v"(t));
            asm volatile("" : "+s"(tab), "+s"(Gp));
            const long b = pa + i, bn = i + 1 < np ? b + 1 : b0;
            double v[D];
            rows_fir_scan(b, true, bn, true, t, (const cdouble_t *)Gp, (const cdouble_t *)&tab->pow2[0][0], v);
            publish(b, t, v);
            RL_STAMP(4); RL_STAMP(5); RL_STAMP(6); RL_STAMP(7); RL_STAMP(8); RL_STAMP(9);
            __syncthreads();                                           // (single wave: orders this step's row reads before the next commit)
            RL_STAMP(10);
        }
        // ---- the chunk's tiles in order
        double zend[D];                                 // the state the wave's previous tile left (lane pair 31's, after its recursion)
#pragma unroll
        for (int k = 0; k < D; k++) zend[k] = 0.0;
#pragma unroll 1
        for (long b = b0; b <= b1; b++) {
            it++;
            if (A.prio) rail_set_priority(np + (int)(b - b0), nsteps);
            RL_STAMP(0);
            // per-iteration values stay per-iteration: otherwise the compiler hoists every lane address of the staging
            // code and every scalar table load out of the persistent loop and spills them
            int t = t0;
            const IirRailTab *tab = A.tab;
            const double *Gp = A.G;
            asm volatile("" : "+v"(t));
            asm volatile("" : "+s"(tab), "+s"(Gp));
            const cdouble_t *pow2 = (const cdouble_t *)&tab->pow2[0][0];
            const int m = t >> 1, rail = t & 1, sh = rail ? A.in_sh1 : A.in_sh0;
            const long tile0 = b * TILE;
            uint32_t *xout = sout + tile0;
            const bool whole = tile_whole(b);
            uint32_t *x = iir_sm + m * PITCH;
            const bool have_next = b < b1;
            const long bn = have_next ? b + 1 : b;
            double v[D];
            rows_fir_scan(b, have_next, bn, false, t, (const cdouble_t *)Gp, pow2, v);
            if (b == b0 && first_publishes) publish(b, t, v);
            RL_STAMP(4);
            // state entering the tile
            double cv[D];
            if (b == b0) {
                const long j = b - 1 - m;                              // lane pair m looks at tile b-1-m; tile -1 = the carried state
                const bool want = m < H && j >= -1;
                double a[D];
#pragma unroll
                for (int k = 0; k < D; k++) a[k] = 0.0;
                // (the carried state is read like an aggregate that is already there: one code path, one wait)
                bool pending = want;
                const unsigned long long *theirs = j >= 0 ? A.agg + ((s * A.n_tiles + j) * 2 + rail) * D
                                                          : (const unsigned long long *)(A.state_in + s * 2 * IIR_MAX_DIM + rail * IIR_MAX_DIM);
                int guard = 0;
                if (A.stamps && t0 == 0 && blockIdx.x < RL_STAMP_ALLWAVES)      // ticks spent polling: -start here, +end behind the loop
                    A.stamps[(size_t)RL_STAMP_WAVES * RL_STAMP_TILES * RL_STAMP_PHASES + RL_STAMP_WORDS * blockIdx.x + 3] -= __builtin_amdgcn_s_memrealtime();
                while (__any(pending)) {
                    if (pending) {
                        // the D stores land in any order: take all of them every time and check each (one round trip through
                        // the fabric per poll)
                        bool ok = true;
#pragma unroll
                        for (int k = 0; k < D; k++) {
                            const unsigned long long w = __hip_atomic_load(theirs + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            ok &= w != IIR_SENTINEL;
                            a[k] = __builtin_bit_cast(double, w);
                        }
                        // (the abort flag is ONE word for the whole launch: looked at on every poll by every waiting lane it
                        // becomes the hottest address of the chip -- measured: 0.29 ms of a 0.52 ms launch; a healthy wait ends
                        // within a few polls and never looks)
                        unsigned ab = 0;
                        if (!ok && (guard & 255) == 255) ab = __hip_atomic_load(A.ctl + RL_CTL_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (ok && A.poll_bound >= 0) pending = false;
                        else if (ab || ++guard > A.poll_bound) {
                            // gave up: the call is void (the host rolls it back), nothing undefined may travel on
#pragma unroll
                            for (int k = 0; k < D; k++) a[k] = 0.0;
                            __hip_atomic_store(A.ctl + RL_CTL_ABORT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_fetch_add(A.overrun, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            pending = false;
                        }
                    }
                    if (__any(pending)) __builtin_amdgcn_s_sleep(2);
                }
                if (A.stamps && t0 == 0 && blockIdx.x < RL_STAMP_ALLWAVES)
                    A.stamps[(size_t)RL_STAMP_WAVES * RL_STAMP_TILES * RL_STAMP_PHASES + RL_STAMP_WORDS * blockIdx.x + 3] += __builtin_amdgcn_s_memrealtime();
                RL_STAMP(5);
                if (H <= RL_HORNER_MAX) {
                    // cv = a_0 + Q (a_1 + Q (a_2 + ...)), the pair's vector broadcast by one shuffle per word
                    const cdouble_t *qm = pow2 + 5 * IIR_MSZ;
#pragma unroll
                    for (int k = 0; k < D; k++) cv[k] = 0.0;
                    for (int h = H - 1; h >= 0; h--) {                 // uniform
                        double nx[D];
#pragma unroll
                        for (int k = 0; k < D; k++) nx[k] = __shfl(a[k], 2 * h + rail, 64);
                        matvec<D, true>(qm, cv, nx);
#pragma unroll
                        for (int k = 0; k < D; k++) cv[k] = nx[k];
                    }
                } else {
                    // sum_h Q^h a_h by a log-step tree: a_h <- a_h + Q^(2^d) a_(h + 2^d); pair 0 ends with the sum
#pragma unroll 1
                    for (int d = 0; d < 5; d++) {
                        double pv[D];
#pragma unroll
                        for (int k = 0; k < D; k++) pv[k] = __shfl_down(a[k], 2 << d, 64);
                        if (m + (1 << d) < RL_SEGS) matvec<D, true>(pow2 + (5 + d) * IIR_MSZ, pv, a);
                    }
#pragma unroll
                    for (int k = 0; k < D; k++) cv[k] = __shfl(a[k], rail, 64);
                }
            } else {
                RL_STAMP(5);
#pragma unroll
                for (int k = 0; k < D; k++) cv[k] = zend[k];
            }
            RL_STAMP(6);
            // the lane's true start state: what the segments before it left (zero carry) + P^m cv.  P^m for the lane's segment
            // comes from a per-segment table by 8-byte loads (12 KB, stays in L2), issued here and not before the wait: at four
            // waves per SIMD the other waves cover the round trip, and the D (D + 2) registers are free during the wait.
            // Four biquads take the rows in two halves (80 registers of table entries would not fit beside the state).
            double z[D];
            {
                const gdouble_t *pt = (const gdouble_t *)&tab->ptab[0][0] + m;
                constexpr int RCH = NS < 4 ? D : D / 2;
#pragma unroll
                for (int r0 = 0; r0 < D; r0 += RCH) {
                    double pe[RCH][D];
#pragma unroll
                    for (int r = 0; r < RCH; r++)
#pragma unroll
                        for (int cc = 0; cc < D; cc++)
                            if (iir_mat_nonzero(r0 + r, cc)) pe[r][cc] = pt[((r0 + r) * IIR_MAX_DIM + cc) * RL_SEGS];
#pragma unroll
                    for (int r = 0; r < RCH; r++) {
                        const double p = __shfl_up(v[r0 + r], 2, 64);
                        double st = m == 0 ? 0.0 : p;
#pragma unroll
                        for (int cc = 0; cc < D; cc++)
                            if (iir_mat_nonzero(r0 + r, cc)) st = __builtin_fma(pe[r][cc], cv[cc], st);
                        z[r0 + r] = st;
                    }
                    if (r0 + RCH < D) {
#pragma unroll
                        for (int r = 0; r < RCH; r++) asm volatile("" : "+v"(z[r0 + r]));
                        asm volatile("" ::: "memory");
                    }
                }
            }
            // the start states are complete (and the P^m entries dead) before the prefetch registers fill
#pragma unroll
            for (int k = 0; k < D; k++) asm volatile("" : "+v"(z[k]));
            asm volatile("" ::: "memory");
            RL_STAMP(7);
            // the next tile's words go out now and land while the recursion (the longest phase) runs
            if constexpr (PF) rail_tile_issue<SEG>(sin + bn * TILE, have_next && tile_whole(bn), raw, t);
            RL_STAMP(8);
            const long seg = b * RL_SEGS + m;
            if (seg < A.n_seg) {
                if (tile0 + TILE <= A.n) rail_recursion<NS, SEG, true, B121>(A.c, x, SEG, z, sh, A.in_width, rail);
                else {
                    const long cnt = A.n - seg * SEG < SEG ? A.n - seg * SEG : SEG;
                    rail_recursion<NS, SEG, false, B121>(A.c, x, cnt, z, sh, A.in_width, rail);
                }
                if (seg == A.n_seg - 1) {
                    double *so = A.state_out + s * 2 * IIR_MAX_DIM + rail * IIR_MAX_DIM;
#pragma unroll
                    for (int k = 0; k < D; k++) so[k] = z[k];
                }
            }
            // what the tile leaves: lane pair 31's state (only a whole tile has a successor in the chunk)
#pragma unroll
            for (int k = 0; k < D; k++) zend[k] = __shfl(z[k], 2 * (RL_SEGS - 1) + rail, 64);
            RL_STAMP(9);
            __syncthreads();
            rail_tile_store<SEG>(xout, whole, A.n - tile0, iir_sm, t);
            __syncthreads();                                           // the rows are free for the next tile
            RL_STAMP(10);
        }
        T = A.dynamic ? rail_tile_of(cls, NC, (unsigned)__builtin_amdgcn_readfirstlane((int)tkn), total) : T + NW;
    }
    if (A.stamps && t0 == 0 && blockIdx.x < RL_STAMP_ALLWAVES) {
        A.stamps[(size_t)RL_STAMP_WAVES * RL_STAMP_TILES * RL_STAMP_PHASES + RL_STAMP_WORDS * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        // steps done | HW_ID (wave, SIMD, CU, SH, SE: where the wave ran) << 16 | XCC_ID << 48   (tools/iir_wave_balance.py)
        const unsigned hw = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11));
        const unsigned xcc = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (31 << 11));
        A.stamps[(size_t)RL_STAMP_WAVES * RL_STAMP_TILES * RL_STAMP_PHASES + RL_STAMP_WORDS * blockIdx.x + 2] =
            (unsigned long long)(it + 1) | ((unsigned long long)hw << 16) | ((unsigned long long)(xcc & 0xF) << 48);

    }
    // the last wave out puts the counters back for the next launch
    if (t0 == 0) {
        if (atomicAdd(A.ctl + RL_CTL_EXIT, 1u) == (unsigned)gridDim.x - 1u) {
            for (int k = 0; k < 66; k++) __hip_atomic_store(A.ctl + k * RL_CTL_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---------------------------------------------------------------------------
// host: transition matrices by simulating the cascade
// ---------------------------------------------------------------------------
static void mat_mul(int dim, const double *a, const double *b, double *o)
{
    double t[IIR_MSZ] = {0};
    for (int r = 0; r < dim; r++)
        for (int c = 0; c < dim; c++) {
            double s = 0;
            for (int k = 0; k < dim; k++) s += a[r * IIR_MAX_DIM + k] * b[k * IIR_MAX_DIM + c];
            t[r * IIR_MAX_DIM + c] = s;
        }
    memcpy(o, t, sizeof t);
}

static double host_step(const IirCoef &c, double *z, double in)
{
    switch (c.n_stages) {
    case 1: return iir_step<1>(c, z, in);
    case 2: return iir_step<2>(c, z, in);
    case 3: return iir_step<3>(c, z, in);
    default: return iir_step<4>(c, z, in);
    }
}

static const int kRailSegs[3] = {16, 32, 64};

// Horizon of the single-pass kernel for tiles of RL_SEGS x seg samples.  int16 inputs bound every reachable state
// component c by 32768 * sum_n |h_c[n]| (h_c = impulse response of that component); a state entering tile b-H reaches
// tile b as Q^H s, so max_r sum_c |Q^H[r][c]| smax[c] bounds what dropping it costs.  Below 1e-12 absolute the tile may
// ignore it: the outputs are integers, a state of this filter is of the order of 1e5..1e6 for full-scale input, so one
// ulp of it is 1e-11..1e-10 and the blocked evaluation's own rounding (a few hundred fp64 operations per state word)
// is orders of magnitude above the bound.  (1e-18 until round 3: H = 2 tiles for the reference's 50 kHz filter at
// 64-sample segments where 1e-12 gives 1 -- one prologue tile per chunk instead of two, 3 % of the run time.)  0 = no such H
// within RL_HMAX, or the cascade's l1 gain lets 32768 x it pass 2^30 (the kernel's v_cvt_i32_f32 equals cvttss2si's low half only for |y| < 2^31).
static double iir_horizon_eps() { return 1e-12; }

static void iir_rail_tab_build(int dim, const double *F, int seg, const double *smax, bool bounded, IirRailTab *tb)
{
    memset(tb, 0, sizeof *tb);
    tb->seg = seg;
    double P[IIR_MSZ];
    memcpy(P, F, sizeof P);
    for (int k = 1; k < seg; k++) mat_mul(dim, P, F, P);
    memcpy(tb->pow2[0], P, sizeof P);
    for (int d = 1; d < 10; d++) mat_mul(dim, tb->pow2[d - 1], tb->pow2[d - 1], tb->pow2[d]);
    static_assert(RL_SEGS == 32, "Q = P^32 = pow2[5]");
    double Pt[IIR_MSZ] = {0};
    for (int r = 0; r < dim; r++) Pt[r * IIR_MAX_DIM + r] = 1.0;
    for (int m = 0; m < RL_SEGS; m++) {
        for (int e = 0; e < IIR_MSZ; e++) tb->ptab[e][m] = Pt[e];
        mat_mul(dim, Pt, P, Pt);
    }
    if (!bounded) return;
    double Qk[IIR_MSZ];
    memcpy(Qk, tb->pow2[5], sizeof Qk);
    for (int k = 1; k <= RL_HMAX; k++) {
        double worst = 0;
        for (int r = 0; r < dim; r++) {
            double acc = 0;
            for (int cc = 0; cc < dim; cc++) acc += fabs(Qk[r * IIR_MAX_DIM + cc]) * 65536.0 * smax[cc];
            worst = fmax(worst, acc);
        }
        if (worst < iir_horizon_eps()) { tb->horizon = k; break; }
        mat_mul(dim, Qk, tb->pow2[5], Qk);
    }
}

static void iir_plan_build(const double *sos, int n_stages, IirPlan *pl)
{
    memset(pl, 0, sizeof *pl);
    IirCoef &c = pl->coef;
    c.n_stages = n_stages; c.dim = 2 * n_stages;
    for (int s = 0; s < n_stages; s++) {
        c.b0[s] = sos[5 * s]; c.b1[s] = sos[5 * s + 1]; c.b2[s] = sos[5 * s + 2];
        c.a1[s] = sos[5 * s + 3]; c.a2[s] = sos[5 * s + 4];
    }
    const int dim = c.dim;
    double F[IIR_MSZ] = {0};
    for (int j = 0; j < dim; j++) {          // column j = one zero-input step from e_j
        double z[IIR_MAX_DIM] = {0};
        z[j] = 1.0;
        (void)host_step(c, z, 0.0);
        for (int r = 0; r < dim; r++) F[r * IIR_MAX_DIM + j] = z[r];
    }
    {   // G[j] = F^j g: the state j steps after a unit input into a resting filter
        double z[IIR_MAX_DIM] = {0};
        (void)host_step(c, z, 1.0);
        for (int j = 0; j < IIR_SEG; j++) {
            for (int r = 0; r < dim; r++) pl->G[j][r] = z[r];
            (void)host_step(c, z, 0.0);
        }
    }
    double P[IIR_MSZ];
    memcpy(P, F, sizeof P);
    for (int k = 1; k < IIR_SEG; k++) mat_mul(dim, P, F, P);     // F^SEG
    memcpy(pl->pow2[0], P, sizeof P);
    for (int d = 1; d < 8; d++) mat_mul(dim, pl->pow2[d - 1], pl->pow2[d - 1], pl->pow2[d]);
    static_assert(IIR_TILE == 64, "Q = P^TILE = P^(2^6)");
    memcpy(pl->Q, pl->pow2[6], sizeof pl->Q);
    static_assert(IIR_GROUP == 256, "Q^GROUP = Q^(2^8)");
    memcpy(pl->qpow2[0], pl->Q, sizeof pl->Q);
    for (int d = 1; d < 14; d++) mat_mul(dim, pl->qpow2[d - 1], pl->qpow2[d - 1], pl->qpow2[d]);
    for (int r = 0; r < dim; r++) pl->qpow[0][r * IIR_MAX_DIM + r] = 1.0;
    for (int i = 1; i < IIR_GROUP; i++) mat_mul(dim, pl->qpow[i - 1], pl->Q, pl->qpow[i]);
    // what the states and the output can reach from int16 inputs (l1 gains by simulation until the response has died)
    double smax[IIR_MAX_DIM] = {0}, z[IIR_MAX_DIM] = {0};
    double ygain = fabs(host_step(c, z, 1.0));
    bool settled = false;
    for (long i = 0; i < 8000000 && !settled; i++) {
        double m = 0;
        for (int r = 0; r < dim; r++) { smax[r] += fabs(z[r]); m = fmax(m, fabs(z[r])); }
        if (!(m < 1e300)) break;                             // diverging: not a filter this path can bound
        if (i > 64 && m < 1e-40) settled = true;
        ygain += fabs(host_step(c, z, 0.0));
    }
    const bool bounded = settled && 32768.0 * ygain < 1073741824.0;
    for (int i = 0; i < 3; i++) iir_rail_tab_build(dim, F, kRailSegs[i], smax, bounded, &pl->rail[i]);
}

// Transition tables per (device, filter): built once, uploaded once into a buffer the shim owns, kept for the life
// of the process (a filter's tables are ~210 KB; an SDR session uses a handful).
struct IirPlanEntry {
    int device, n_stages;
    double sos[5 * IIR_MAX_STAGES];
    IirPlan host;                  // coef goes to the kernels by value
    IirPlan *dev;                  // device copy
};

static const IirPlanEntry *iir_plan_for(const double *sos, int n_stages)
{
    static std::mutex mu;
    static std::vector<IirPlanEntry *> cache;
    int device = 0;
    (void)hipGetDevice(&device);
    std::lock_guard<std::mutex> lock(mu);
    for (const IirPlanEntry *e : cache)
        if (e->device == device && e->n_stages == n_stages && !memcmp(e->sos, sos, sizeof(double) * 5 * n_stages)) return e;
    // entries are never dropped: filter objects keep pointers into them
    IirPlanEntry *e = new (std::nothrow) IirPlanEntry();
    if (!e) { clhip_set_error("clhip_iir: out of memory"); return nullptr; }
    e->device = device; e->n_stages = n_stages;
    memcpy(e->sos, sos, sizeof(double) * 5 * n_stages);
    iir_plan_build(sos, n_stages, &e->host);
    e->dev = (IirPlan *)clhip_malloc(sizeof(IirPlan));
    if (!e->dev || hipMemcpy(e->dev, &e->host, sizeof(IirPlan), hipMemcpyHostToDevice) != hipSuccess) {
        clhip_set_error("clhip_iir: cannot place the filter tables on the device");
        clhip_free(e->dev); delete e;
        return nullptr;
    }
    cache.push_back(e);
    return e;
}

static size_t iir_scan_ws_doubles(size_t n_samples)
{
    const size_t n_seg = clhip_div_up(n_samples, IIR_SEG), n_tiles = clhip_div_up(n_seg, IIR_TILE);
    return (n_seg + 2 * n_tiles + 2 * clhip_div_up(n_tiles, IIR_GROUP) + 4) * 2 * IIR_MAX_DIM;
}

typedef void (*iir_rail_fn)(const IirRailArgs);
template <int NS, int SEG>
static iir_rail_fn rail_pick(bool b121)
{
    if constexpr (NS > 1) { if (b121) return iir_rail_kernel<NS, SEG, true>; }
    return iir_rail_kernel<NS, SEG, false>;
}
static iir_rail_fn rail_kernel_for(int ns, int seg, bool b121)
{
    switch (ns * 100 + seg) {
    case 116: return rail_pick<1, 16>(b121);
    case 132: return rail_pick<1, 32>(b121);
    case 164: return rail_pick<1, 64>(b121);
    case 216: return rail_pick<2, 16>(b121);
    case 232: return rail_pick<2, 32>(b121);
    case 264: return rail_pick<2, 64>(b121);
    case 316: return rail_pick<3, 16>(b121);
    case 332: return rail_pick<3, 32>(b121);
    case 364: return rail_pick<3, 64>(b121);
    case 416: return rail_pick<4, 16>(b121);
    case 432: return rail_pick<4, 32>(b121);
    case 464: return rail_pick<4, 64>(b121);
    }
    return nullptr;
}

// Waves the device keeps resident for a kernel instantiation: the launch's size (more would only queue up behind the
// persistent ones and leave at once; progress does not depend on the number, see the kernel's header).
static int iir_resident_waves(iir_rail_fn fn, int seg)
{
    struct Ent { int device; iir_rail_fn fn; int waves; };
    static std::mutex mu;
    static std::vector<Ent> cache;
    int device = 0, cus = 256, per_cu = 0;
    (void)hipGetDevice(&device);
    std::lock_guard<std::mutex> lock(mu);
    for (const Ent &e : cache) if (e.device == device && e.fn == fn) return e.waves;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64, RL_SEGS * (seg + 4) * 4) != hipSuccess || per_cu < 1) per_cu = 1;
    if (per_cu > 4) per_cu -= per_cu % 4;                        // the same number of waves on each of the CU's four SIMDs
    cache.push_back({device, fn, cus * per_cu});
    return cus * per_cu;
}

// ---------------------------------------------------------------------------
// The filter object: coefficients, carried state (ping-pong), the single-pass kernel's control words and aggregate
// buffers, a pinned overrun word of its own -- one per Soapy stream and filter selection, so two streams on one
// GPU never see each other's verdicts.
// ---------------------------------------------------------------------------
struct clhip_iir {
    int device, n_stages, n_streams;
    const IirPlanEntry *pe;
    bool b121;
    double *d_state;                        // [2][n_streams][2 * IIR_MAX_DIM]
    int cur;                                // the half that holds the state entering the next call
    unsigned int *d_ctl;
    unsigned long long *d_agg[2];
    size_t agg_cap;                         // words per buffer
    size_t dirty[2];                        // words of each buffer that may not be all-ones
    int acur;
    double *d_scan_ws; size_t scan_ws_doubles;
    unsigned int *h_over, *d_over;          // pinned, device-mapped
    int poll_bound, seg_force, dynamic;
    bool force_scan;
    // the last call, for the verdict and the redo
    bool can_undo; int undo_cur;
    const int16_t *last_in; int16_t *last_out; size_t last_stride, last_n; hipStream_t last_stream; bool last_valid;
    bool last_was_rail, last_is_words;
    unsigned long long *d_stamps;           // diagnostics
};

extern "C" void clhip_iir_destroy(clhip_iir *f)
{
    if (!f) return;
    if (f->last_valid) (void)hipStreamSynchronize(f->last_stream);
    clhip_free(f->d_stamps);
    clhip_free(f->d_state); clhip_free(f->d_ctl); clhip_free(f->d_agg[0]); clhip_free(f->d_agg[1]); clhip_free(f->d_scan_ws);
    if (f->h_over) (void)hipHostFree(f->h_over);
    delete f;
}

extern "C" clhip_iir *clhip_iir_create(const double *h_sos, int n_stages, int n_streams)
{
    if (!h_sos || n_stages < 1 || n_stages > IIR_MAX_STAGES || n_streams < 1) {
        clhip_set_error("clhip_iir_create: bad arguments (1..%d biquads)", IIR_MAX_STAGES);
        return nullptr;
    }
    clhip_iir *f = new (std::nothrow) clhip_iir();
    if (!f) return nullptr;
    memset(f, 0, sizeof *f);
    (void)hipGetDevice(&f->device);
    f->n_stages = n_stages; f->n_streams = n_streams;
    f->pe = iir_plan_for(h_sos, n_stages);
    if (!f->pe) { delete f; return nullptr; }
    const IirCoef &c = f->pe->host.coef;
    f->b121 = n_stages > 1;                 // b = (1, 2, 1) exactly in every stage after the first: the four-operation stage form
    for (int k = 1; k < n_stages; k++) f->b121 = f->b121 && c.b0[k] == 1.0 && c.b1[k] == 2.0 && c.b2[k] == 1.0;
    const size_t st_bytes = sizeof(double) * 2 * 2 * IIR_MAX_DIM * n_streams;
    f->d_state = (double *)clhip_malloc(st_bytes);
    f->d_ctl = (unsigned int *)clhip_malloc(sizeof(unsigned int) * RL_CTL_WORDS);
    if (hipHostMalloc((void **)&f->h_over, 64, hipHostMallocMapped) != hipSuccess) f->h_over = nullptr;
    if (f->h_over) {
        *f->h_over = 0;
        if (hipHostGetDevicePointer((void **)&f->d_over, f->h_over, 0) != hipSuccess) f->d_over = nullptr;
    }
    if (!f->d_state || !f->d_ctl || !f->d_over || hipMemset(f->d_state, 0, st_bytes) != hipSuccess ||
        hipMemset(f->d_ctl, 0, sizeof(unsigned int) * RL_CTL_WORDS) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) {
        clhip_set_error("clhip_iir_create: device allocation failed");
        clhip_iir_destroy(f);
        return nullptr;
    }
    // tests force the give-up path with -1; the default bound is ~0.1 s of polling, after which the call is rolled back
    // and repeated on the scan path (a launch that moves at all never gets there: see the kernel's header)
    f->poll_bound = 1 << 16;
    f->seg_force = 0;                    // (clhip_iir_set_shape: tests force every segment length and both tile orders)
    f->dynamic = 1;
    f->force_scan = getenv("CLHIP_IIR_ONEPASS") && atoi(getenv("CLHIP_IIR_ONEPASS")) == 0;   // A/B: the four-kernel scan for everything
    if (getenv("CLHIP_IIR_STAMPS") && atoi(getenv("CLHIP_IIR_STAMPS"))) {
        f->d_stamps = (unsigned long long *)clhip_malloc(sizeof(unsigned long long) * (RL_STAMP_WAVES * RL_STAMP_TILES * RL_STAMP_PHASES + RL_STAMP_WORDS * RL_STAMP_ALLWAVES));
        if (f->d_stamps) (void)hipMemset(f->d_stamps, 0, sizeof(unsigned long long) * (RL_STAMP_WAVES * RL_STAMP_TILES * RL_STAMP_PHASES + RL_STAMP_WORDS * RL_STAMP_ALLWAVES));
    }
    return f;
}

// diagnostics: the phase time stamps of the last single-pass launch (CLHIP_IIR_STAMPS=1 at create): [64 waves][16 tiles][12 phases]
// of the 100 MHz real-time counter, 0 where nothing was recorded, then [8192 waves][start, end, tiles done]; returns the
// number of words, 0 when not enabled
extern "C" size_t clhip_iir_debug_stamps(clhip_iir *f, unsigned long long *h_out)
{
    if (!f || !f->d_stamps) return 0;
    const size_t nw = (size_t)RL_STAMP_WAVES * RL_STAMP_TILES * RL_STAMP_PHASES + RL_STAMP_WORDS * RL_STAMP_ALLWAVES;
    if (f->last_valid) (void)hipStreamSynchronize(f->last_stream);
    if (h_out && hipMemcpy(h_out, f->d_stamps, sizeof(unsigned long long) * nw, hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return nw;
}

extern "C" void clhip_iir_set_poll_bound(clhip_iir *f, int polls) { if (f) f->poll_bound = polls; }
extern "C" void clhip_iir_set_shape(clhip_iir *f, int seg, int dynamic) { if (f) { f->seg_force = seg == 16 || seg == 32 || seg == 64 ? seg : 0; f->dynamic = dynamic != 0; } }

// The filter's MEMORY in samples: after that many samples nothing a state could have held is left above the single-pass
// kernel's bound (1e-12 absolute for full-scale int16 input: iir_rail_tab_build).  A filter that starts from rest that far
// before a point of the stream is, from that point on, in the state of a filter that has seen the whole stream -- which is
// what lets one long stream be cut into time slices for several GPUs with a halo instead of a state hand-off
// (cariboulite_amd/shard.py run_iir_time_slice).  0: the memory is longer than the kernel's horizon.
extern "C" size_t clhip_iir_memory_samples(const clhip_iir *f)
{
    if (!f) return 0;
    size_t best = 0;
    for (int i = 0; i < 3; i++) {
        const int h = f->pe->host.rail[i].horizon;
        if (h <= 0) continue;
        const size_t m = (size_t)h * RL_SEGS * (size_t)kRailSegs[i];
        if (!best || m < best) best = m;
    }
    return best;
}
extern "C" int clhip_iir_on_scan_path(const clhip_iir *f) { return f && f->force_scan ? 1 : 0; }

extern "C" int clhip_iir_set_state(clhip_iir *f, const double *h_state)
{
    if (!f) return -1;
    if (f->last_valid) CLHIP_CHECK(hipStreamSynchronize(f->last_stream));
    const size_t half = (size_t)2 * IIR_MAX_DIM * f->n_streams;
    if (h_state) CLHIP_CHECK(hipMemcpy(f->d_state + f->cur * half, h_state, sizeof(double) * half, hipMemcpyHostToDevice));
    else CLHIP_CHECK(hipMemset(f->d_state + f->cur * half, 0, sizeof(double) * half));
    CLHIP_CHECK(hipStreamSynchronize(nullptr));
    f->can_undo = false;
    return 0;
}

extern "C" int clhip_iir_get_state(clhip_iir *f, double *h_state)
{
    if (!f || !h_state) return -1;
    if (f->last_valid) CLHIP_CHECK(hipStreamSynchronize(f->last_stream));
    const size_t half = (size_t)2 * IIR_MAX_DIM * f->n_streams;
    CLHIP_CHECK(hipMemcpy(h_state, f->d_state + f->cur * half, sizeof(double) * half, hipMemcpyDeviceToHost));
    return 0;
}

// segment length for a call: short segments spread a small call over many waves (one native batch of 131072 samples
// is 128 tiles at 32-sample segments, 32 at 64) and shorten each wave's serial work; long ones pay the per-tile scan
// least often (measured, one stream: 2^17 samples 13.6 / 13.1 / 15.3 us at 16 / 32 / 64; 2^26: 0.28 ms at 32, 0.23 at 64).
// -1 = no single-pass shape for this filter (memory too long): the scan.
static int iir_pick_shape(const clhip_iir *f, size_t n)
{
    // measured per size (tools/bench_iir.py, CLHIP_IIR_SEG forced; us per call at SEG 16 / 32 / 64): 2^16 11.6 / 12.1 / 14.9,
    // 2^17 13.1 / 12.7 / 14.9, 2^18 16.2 / 14.4 / 15.5, 2^19 22.9 / 17.5 / 17.1, 2^20 34.9 / 24.3 / 20.5, 2^21 60 / 36.6 / 27.6:
    // short segments pay only while the call is too small to give every CU a wave
    static const size_t tiles_max[3] = {128, 256, (size_t)-1};
    const IirPlan &pl = f->pe->host;
    for (int i = 0; i < 3; i++) {
        if (!pl.rail[i].horizon) continue;
        if (f->seg_force) { if (kRailSegs[i] == f->seg_force) return i; continue; }
        const size_t tiles = clhip_div_up(clhip_div_up(n, (size_t)kRailSegs[i]), RL_SEGS) * f->n_streams;
        if (tiles <= tiles_max[i]) return i;
    }
    return -1;
}

template <int NS>
static void iir_launch_scan(const IirPlan *d_plan, const IirCoef &coef, const double *st_in, double *st_out, const uint32_t *in,
                            uint32_t *out, long stride, long n, int n_streams, double *ws, hipStream_t s)
{
    constexpr int D2 = 4 * NS;               // doubles per state pair
    const long n_seg = (long)clhip_div_up((size_t)n, IIR_SEG), n_tiles = (long)clhip_div_up((size_t)n_seg, IIR_TILE);
    const long n_groups = (long)clhip_div_up((size_t)n_tiles, IIR_GROUP);
    double *ZS = ws, *tend = ZS + n_seg * n_streams * D2, *X = tend + n_tiles * n_streams * D2;
    double *gend = X + n_tiles * n_streams * D2, *gc = gend + n_groups * n_streams * D2;
    dim3 grid((unsigned)n_tiles, n_streams), block(IIR_TILE);
    static_assert(IIR_TILE * (2 * IIR_MAX_DIM + 1) * 8 <= IIR_LDS_WORDS * 4, "the scan exchange fits the tile's LDS");
    hipLaunchKernelGGL(iir_k1_kernel<NS>, grid, block, IIR_LDS_WORDS * 4, s, d_plan, in, stride, n, n_seg, n_tiles, ZS, tend);
    hipLaunchKernelGGL(iir_k2a_kernel<NS>, dim3((unsigned)n_groups, n_streams), dim3(IIR_GROUP), 0, s, d_plan, n_tiles, n_groups,
                       (const double *)tend, X, gend);
    hipLaunchKernelGGL(iir_k2b_kernel<NS>, dim3((unsigned)n_streams), dim3(64), 0, s, d_plan, n_groups, (const double *)gend, gc, st_in);
    hipLaunchKernelGGL(iir_k3_kernel<NS>, grid, block, IIR_LDS_WORDS * 4, s, d_plan, coef, in, out, stride, n, n_seg, n_tiles,
                       n_groups, (const double *)ZS, (const double *)X, (const double *)gc, st_out);
}

static int iir_run_impl(clhip_iir *f, const int16_t *d_in, int16_t *d_out, size_t stride_samples, size_t n_samples, void *stream,
                        int in_sh0, int in_sh1, int in_width);

extern "C" int clhip_iir_run(clhip_iir *f, const int16_t *d_in, int16_t *d_out, size_t stride_samples, size_t n_samples, void *stream)
{
    return iir_run_impl(f, d_in, d_out, stride_samples, n_samples, stream, 0, 16, 16);
}

// The same filter fed straight from raw SMI RX words (4 bytes per sample, every word in sync: caribou_smi.c:338-378 -- S1G:
// i = bits 29..17, q = bits 13..1; HiF: the other way round): the unpack is the filter's own field extraction, so a read
// that is one in-sync read() needs no unpack launch and no int16 intermediate (cl_readStream does this).  Out of place
// only.  Returns -2 when this call would take the scan path (a filter outside the single-pass kernel's horizon, or an
// object that has overrun before): the caller then unpacks first and calls clhip_iir_run.  After an overrun of THIS call
// clhip_iir_status() restores the state as usual; the repeat goes through clhip_smi_unpack* + clhip_iir_run.
extern "C" int clhip_iir_run_smi(clhip_iir *f, int channel, const uint8_t *d_words, int16_t *d_out, size_t stride_samples,
                                 size_t n_samples, void *stream)
{
    if (!f || !d_words || (const void *)d_words == (const void *)d_out) { clhip_set_error("clhip_iir_run_smi: null / in-place buffers"); return -1; }
    if (f->force_scan || iir_pick_shape(f, n_samples) < 0) return -2;
    const bool hif = channel == CL_CHANNEL_HIF;
    return iir_run_impl(f, (const int16_t *)d_words, d_out, stride_samples, n_samples, stream, hif ? 1 : 17, hif ? 17 : 1, 13);
}

static int iir_run_impl(clhip_iir *f, const int16_t *d_in, int16_t *d_out, size_t stride_samples, size_t n_samples, void *stream,
                        int in_sh0, int in_sh1, int in_width)
{
    if (!f || !d_in || !d_out || (((uintptr_t)d_in | (uintptr_t)d_out) & 3)) {
        clhip_set_error("clhip_iir_run: null / misaligned buffer");
        return -1;
    }
    if (n_samples == 0) return 0;
    if (f->n_streams > 1 && stride_samples < n_samples) { clhip_set_error("clhip_iir_run: streams overlap"); return -1; }
    if (*(volatile unsigned int *)f->h_over) {                 // raised by an earlier launch nobody asked about
        clhip_set_error("clhip_iir_run: an earlier call overran and clhip_iir_status() was not consulted; its output and the state are invalid");
        *f->h_over = 0;
        f->can_undo = false;
        f->force_scan = true;
        return -1;
    }
    hipStream_t s = (hipStream_t)stream;
    const IirPlan &plan = f->pe->host;
    const size_t half = (size_t)2 * IIR_MAX_DIM * f->n_streams;
    const double *st_in = f->d_state + f->cur * half;
    double *st_out = f->d_state + (f->cur ^ 1) * half;
    const long n = (long)n_samples, stride = (long)stride_samples;
    const int shape = f->force_scan ? -1 : iir_pick_shape(f, n_samples);
    if (shape >= 0) {
        const int seg = kRailSegs[shape];
        const long n_seg = (long)clhip_div_up(n_samples, (size_t)seg), n_tiles = (long)clhip_div_up((size_t)n_seg, RL_SEGS);
        const size_t words = (size_t)n_tiles * f->n_streams * 4 * f->n_stages;      // [stream][tile][rail][D]
        if (words > f->agg_cap) {
            // grow both buffers (rare: the first call, or a longer one than any before)
            if (f->last_valid) CLHIP_CHECK(hipStreamSynchronize(f->last_stream));
            for (int i = 0; i < 2; i++) {
                clhip_free(f->d_agg[i]);
                f->d_agg[i] = (unsigned long long *)clhip_malloc(sizeof(unsigned long long) * words);
                if (!f->d_agg[i]) { f->agg_cap = 0; return -1; }
                CLHIP_CHECK(hipMemsetAsync(f->d_agg[i], 0xFF, sizeof(unsigned long long) * words, s));
                f->dirty[i] = 0;
            }
            f->agg_cap = words;
        }
        iir_rail_fn fn = rail_kernel_for(f->n_stages, seg, f->b121);
        const int resident = iir_resident_waves(fn, seg);
        // a wave takes CHUNKS of consecutive tiles of one stream: as many as share the launch evenly over the resident
        // waves (one native batch: 1; 2^26 samples: 8), so that only a chunk's first tile looks at other waves' work
        // (round 3, 2^26 samples: chunks of 4 / 16 tiles instead of 8 lose -- 0.252 / 0.248 against 0.233 ms -- as do 12 or 8 waves per CU)
        long chunk = (long)clhip_div_up((size_t)(n_tiles * f->n_streams), (size_t)resident);
        if (chunk > n_tiles) chunk = n_tiles;
        const long chunks_per_stream = (long)clhip_div_up((size_t)n_tiles, (size_t)chunk);
        const long total = chunks_per_stream * f->n_streams;
        if (total >= 0x7fffffffL) { clhip_set_error("clhip_iir_run: call too large"); return -1; }
        const unsigned grid = (unsigned)(total < resident ? total : resident);
        int nc = RL_CLASSES;
        while ((unsigned)nc > grid) nc >>= 1;
        IirRailArgs a;
        a.G = &f->pe->dev->G[0][0]; a.tab = &f->pe->dev->rail[shape]; a.c = plan.coef;
        a.in = (const uint32_t *)d_in; a.out = (uint32_t *)d_out;
        a.stride = stride; a.n = n; a.n_seg = n_seg; a.n_tiles = n_tiles;
        a.n_streams = f->n_streams; a.horizon = plan.rail[shape].horizon;
        a.chunk_tiles = (int)chunk; a.chunks_per_stream = chunks_per_stream;
        a.ctl = f->d_ctl;
        a.agg = f->d_agg[f->acur]; a.agg_other = f->d_agg[f->acur ^ 1]; a.other_words = (long)f->dirty[f->acur ^ 1];
        a.state_in = st_in; a.state_out = st_out;
        a.overrun = f->d_over; a.poll_bound = f->poll_bound; a.n_classes = nc; a.dynamic = f->dynamic;
        a.stamps = f->d_stamps;
        a.in_sh0 = in_sh0; a.in_sh1 = in_sh1; a.in_width = in_width;
        a.prio = chunk > 1;                    // a wave lowers its priority as it advances through its chunk (the four waves of a SIMD finish 32 instead of 96 us apart: -4 %)
        hipLaunchKernelGGL(fn, dim3(grid), dim3(64), RL_SEGS * (seg + 4) * 4, s, a);
        CLHIP_CHECK_LAUNCH();
        f->dirty[f->acur] = words > f->dirty[f->acur] ? words : f->dirty[f->acur];
        f->dirty[f->acur ^ 1] = 0;
        f->acur ^= 1;
        f->last_was_rail = true;
    } else {
        if (in_width != 16) { clhip_set_error("clhip_iir_run: the scan path takes int16 samples"); return -1; }
        const size_t need = iir_scan_ws_doubles(n_samples) * f->n_streams;
        if (need > f->scan_ws_doubles) {
            if (f->last_valid) CLHIP_CHECK(hipStreamSynchronize(f->last_stream));
            clhip_free(f->d_scan_ws);
            f->d_scan_ws = (double *)clhip_malloc(sizeof(double) * need);
            f->scan_ws_doubles = f->d_scan_ws ? need : 0;
            if (!f->d_scan_ws) return -1;
        }
        const IirPlan *dp = f->pe->dev;
        const uint32_t *in = (const uint32_t *)d_in;
        uint32_t *out = (uint32_t *)d_out;
        switch (f->n_stages) {
        case 1: iir_launch_scan<1>(dp, plan.coef, st_in, st_out, in, out, stride, n, f->n_streams, f->d_scan_ws, s); break;
        case 2: iir_launch_scan<2>(dp, plan.coef, st_in, st_out, in, out, stride, n, f->n_streams, f->d_scan_ws, s); break;
        case 3: iir_launch_scan<3>(dp, plan.coef, st_in, st_out, in, out, stride, n, f->n_streams, f->d_scan_ws, s); break;
        default: iir_launch_scan<4>(dp, plan.coef, st_in, st_out, in, out, stride, n, f->n_streams, f->d_scan_ws, s); break;
        }
        CLHIP_CHECK_LAUNCH();
        f->last_was_rail = false;
    }
    f->undo_cur = f->cur; f->can_undo = true;
    f->cur ^= 1;
    f->last_in = d_in; f->last_out = d_out; f->last_stride = stride_samples; f->last_n = n_samples; f->last_stream = s; f->last_valid = true;
    f->last_is_words = in_width != 16;
    return 0;
}

// After the caller has synchronised the stream of the last clhip_iir_run: 0 = its output is valid.  -1 = a poll of the
// single-pass kernel gave up: the samples of that call must not be used; the carried state is back where it was
// before the call (the launch wrote the other half), and from now on this object takes the scan path (no waiting
// between workgroups), so the call can simply be made again.
extern "C" int clhip_iir_status(clhip_iir *f)
{
    if (!f) return -1;
    if (!*(volatile unsigned int *)f->h_over) return 0;
    *f->h_over = 0;
    f->force_scan = true;
    if (f->can_undo) { f->cur = f->undo_cur; f->can_undo = false; }
    clhip_set_error("clhip_iir_status: a tile of the single-pass kernel gave up waiting for its predecessors; the output of the last call is invalid, filter state restored");
    return -1;
}

// The last call is taken back (a caller that filtered ahead of its client, who then went another way): its stream must have been
// synchronised.  The carried state is again what it was before that call; a verdict that call left behind is consumed.
// 0, or -1 when there is no call to take back.
extern "C" int clhip_iir_unrun(clhip_iir *f)
{
    if (!f) return -1;
    if (clhip_iir_status(f) != 0) return 0;               // it had overrun: status has put the state back already
    if (!f->can_undo) { clhip_set_error("clhip_iir_unrun: no call to take back"); return -1; }
    f->cur = f->undo_cur; f->can_undo = false;
    return 0;
}

// Synchronise, ask, and repair: 0 = good; 1 = the call overran and has been made again on the scan path (d_out holds
// the right samples now, the state has advanced once); -1 = overran in place (the input is gone: state restored,
// the caller re-produces the input and calls again) or a runtime error.
extern "C" int clhip_iir_finish(clhip_iir *f)
{
    if (!f || !f->last_valid) return 0;
    CLHIP_CHECK(hipStreamSynchronize(f->last_stream));
    if (clhip_iir_status(f) == 0) return 0;
    if (f->last_in == f->last_out || f->last_is_words) return -1;        // (raw words: the caller unpacks and calls clhip_iir_run)
    if (clhip_iir_run(f, f->last_in, f->last_out, f->last_stride, f->last_n, f->last_stream)) return -1;
    CLHIP_CHECK(hipStreamSynchronize(f->last_stream));
    return clhip_iir_status(f) == 0 ? 1 : -1;
}
