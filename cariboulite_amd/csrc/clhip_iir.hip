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
//       recursion over its segment and writes the truncated int16 outputs in place.
//       The tile's global loads are in flight while the scan runs.
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
// Compile-time shape of the single-pass kernel (defaults = the measured optimum; the others are kept buildable because
// they were measured): samples per lane segment (a multiple of 16), waves per SIMD the register
// budget is cut for, and whether the next tile's words wait in registers during the recursion.
#ifndef IIR_SEG
#define IIR_SEG 64
#endif
#ifndef IIR_WAVES_PER_SIMD
#define IIR_WAVES_PER_SIMD 2
#endif
#ifndef IIR_PREFETCH
#define IIR_PREFETCH 1                     // the single-pass kernel keeps the next tile's words in registers during the recursion
#endif
#define IIR_TILE 64                        // one wave per workgroup: no workgroup barrier anywhere in K1 / K3
#define IIR_K2_LANES 256
#define IIR_GROUP 256                      // tiles per K2a workgroup
#define IIR_MSZ (IIR_MAX_DIM * IIR_MAX_DIM) // matrices are stored 8x8, row-major, zero outside DxD

typedef __attribute__((address_space(4))) double cdouble_t;   // read-only tables: scalar (SMEM) loads when uniform

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

struct IirPlan {
    IirCoef coef;
    double G[IIR_SEG][IIR_MAX_DIM];         // G[j] = F^j g
    double pow2[8][IIR_MSZ];                // P^(2^d), P = F^SEG
    double Q[IIR_MSZ];                      // P^TILE
    double qpow2[14][IIR_MSZ];              // Q^(2^d); [8 + d] = (Q^256)^(2^d) chains the groups
    double qpow[IIR_GROUP][IIR_MSZ];        // Q^i
    double ptab[IIR_MSZ][IIR_TILE];         // ptab[8 r + c][t] = (P^t)[r][c]: entry-major, so the 64 lanes of one load read 512 contiguous bytes
    int horizon;                            // tiles after which a carried state has decayed below 1e-18 (0: unknown / too long)
                                            // (only set when 32768 x the cascade's l1 gain also stays below 2^30: see iir_to_i16)
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
__global__ __launch_bounds__(IIR_TILE) void iir_k3_kernel(const IirPlan *__restrict__ plan, IirCoef c, uint32_t *__restrict__ iq,
                                                         long stride, long n, long n_seg, long n_tiles, long n_groups,
                                                         const double *__restrict__ ZS, const double *__restrict__ X,
                                                         const double *__restrict__ gc, double *__restrict__ state)
{
    constexpr int D = 2 * NS, RS = 2 * D + 1;
    extern __shared__ __attribute__((aligned(16))) uint32_t iir_sm[];
    const int t = threadIdx.x;
    const long tile0 = (long)blockIdx.x * IIR_TILE * IIR_SEG;
    uint32_t *xt = iq + (long)blockIdx.y * stride + tile0;
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
    iir_tile_store(xt, n - tile0, iir_sm, t);
}


// ---------------------------------------------------------------------------
// Single pass (default).  The stream is read ONCE and written once: a persistent wave takes tiles by ticket,
// stages a tile in LDS, forms every segment's zero-state end vector (the K1 matrix FIR), scans them over the tile
// (zero carry) and publishes the tile's zero-carry end vector -- its AGGREGATE.  The state entering the tile is then
//     cv = sum_{k=1..H} Q^(k-1) aggregate(tile - k)            (+ Q^tile * carried state for the first H tiles)
// where H = plan->horizon is the number of tiles after which ANY reachable state has decayed below 1e-18 in absolute
// terms (int16 inputs bound the state; the bound is evaluated on the host from |Q^H| and the filter's l1 gains, far
// below the 1e-8 rounding noise the fp64 sums carry anyway).  So a tile waits for H predecessors' aggregates -- not
// for their prefixes: there is no chain through the launch, every tile is done a fixed time after it starts, and a
// filter too narrow for H <= IIR_HMAX takes the four-kernel scan instead.  Lane t's start state is the exclusive
// scan value + P^t cv; it runs the recursion over its LDS row in place and the tile leaves with coalesced stores
// while the next tile's loads (issued before the recursion) are already in registers.
//   Publication: an aggregate is 2D doubles written with relaxed agent-scope 64-bit atomic stores into slots the
// host pre-set to all-ones (a NaN no aggregate can be): a reader polls until none of the 2D words is the sentinel,
// so no flag, no fence and no store ordering is needed.  Producing an aggregate never waits for anything, so the
// poll always ends once the producer's wave is resident (it is bounded all the same).
// ---------------------------------------------------------------------------
#define IIR_HMAX 8
#define IIR_SENTINEL 0xFFFFFFFFFFFFFFFFull

// A wave's rank: start order within its class (block index mod 64), classes interleaved -- rank = class + 64 * (how many
// waves of the class started before it).  Workgroups are dispatched in index order, so this is the start order of the
// launch to within a few waves, and a wave that is running still only ever waits for waves that started before it or
// are starting now, which is what keeps a launch that is only partly resident moving.  One counter for all waves would
// give the exact order, but same-address atomics retire at ~12 ns each: 2 048 of them delay the last wave by 25 us, a
// tenth of this kernel's run time (and a ticket per TILE, 16 384 of them, two thirds of it).  The 64 counters start at
// all-ones.
#define IIR_RANK_CLASSES 64
__device__ __forceinline__ long iir_take_rank(unsigned int *ticket, int t0)
{
    const unsigned int cls = blockIdx.x & (IIR_RANK_CLASSES - 1);
    unsigned int tk = 0;
    if (t0 == 0) tk = atomicAdd(ticket + cls, 1u) + 1u;
    return (long)cls + (long)IIR_RANK_CLASSES * (long)(unsigned)__builtin_amdgcn_readfirstlane((int)tk);
}

template <int NS, bool B121>
__global__ __launch_bounds__(IIR_TILE, IIR_WAVES_PER_SIMD) void iir_onepass_kernel(const IirPlan *__restrict__ plan, IirCoef c, uint32_t *__restrict__ iq,
                                                                  long stride, long n, long n_seg, long n_tiles, int n_streams,
                                                                  unsigned int *ticket, unsigned long long *agg,
                                                                  unsigned int *readers, double *state_io,
                                                                  int horizon, unsigned int *overruns, int poll_bound, int dbg, int stagger_ticks)
{
    constexpr int D = 2 * NS, D2 = 2 * D;
    extern __shared__ __attribute__((aligned(16))) uint32_t iir_sm[];
    const int t0 = threadIdx.x;
    const long total = n_tiles * n_streams;

    // ONE ticket per wave: its rank.  Rank r takes tiles r, r + G, r + 2G, ... (G = waves launched): a tile's H
    // predecessors belong to the H ranks before it, which are at the same point of their own lists, so nobody waits
    // long; ranks 0..H-1 wrap to the last ranks of the previous round.  Aggregates are published BEFORE a wave waits
    // for anything, so a late-starting rank delays its successors, never deadlocks them.
    long T = iir_take_rank(ticket, t0);
    const long NW = (long)gridDim.x;                                  // waves launched
    if (stagger_ticks != 0 && T < total) {
        // Spread the waves' phases over one tile period (rank r starts r/NW of a period late): identical waves
        // started together stay in lock step, so the whole chip would load, compute and store in unison and the
        // memory system would idle during the compute phases.  The two waves of a SIMD (ranks r and r + NW/2 in
        // dispatch order) end up half a period apart.  100 MHz constant clock.
        // stagger_ticks > 0: by rank over the launch; < 0: by ring (= stream of the rank's first tile), |stagger_ticks| apart per ring
        const unsigned long long until = __builtin_amdgcn_s_memrealtime() +
            (stagger_ticks > 0 ? (unsigned long long)(T * stagger_ticks / NW) : (unsigned long long)((T % n_streams) * (long)(-stagger_ticks)));
        while (__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(8);
    }
#if IIR_PREFETCH
    u32x4 raw[IIR_NLD];
    if (T < total) {
        const long b = T / n_streams, s = T % n_streams;
        iir_tile_issue(iq + s * stride + b * IIR_TILE * IIR_SEG, n - b * IIR_TILE * IIR_SEG, raw, t0);
    }
#endif
    while (T < total) {
        // per-iteration values stay per-iteration: otherwise the compiler hoists every lane address of the staging
        // code and every scalar table load out of the persistent loop and spills them
        int t = t0;
        const IirPlan *pl = plan;
        asm volatile("" : "+v"(t));
        asm volatile("" : "+s"(pl));
        const cdouble_t *pow2 = (const cdouble_t *)&pl->pow2[0][0];
        const cdouble_t *__restrict__ G = (const cdouble_t *)&pl->G[0][0];
        const long b = T / n_streams, s = T % n_streams;
        const long tile0 = b * IIR_TILE * IIR_SEG;
        uint32_t *xt = iq + s * stride + tile0;
#if !IIR_PREFETCH
        if (iir_tile_whole(xt, n - tile0)) {
            u32x4 raw[IIR_NLD];
#pragma unroll
            for (int q = 0; q < IIR_NLD; q++) raw[q] = *(const u32x4 *)(xt + (q * IIR_TILE + t) * 4);
#pragma unroll
            for (int q = 0; q < IIR_NLD; q++) {
                const int i = (q * IIR_TILE + t) * 4;
                *(u32x4 *)(iir_sm + (i / IIR_SEG) * IIR_PITCH + i % IIR_SEG) = raw[q];
            }
        } else {
            u32x4 none[IIR_NLD];
            iir_tile_commit(xt, n - tile0, none, iir_sm, t);
        }
#else
        iir_tile_commit(xt, n - tile0, raw, iir_sm, t);
#endif
        __syncthreads();
        const long Tn = T + NW;
        // zero-state end vector of the lane's segment: zs = sum_k (F^(63-k) g) x[k]
        uint32_t *x = iir_sm + t * IIR_PITCH;
        double v[D2];
#pragma unroll
        for (int k = 0; k < D2; k++) v[k] = 0.0;
        if (!(dbg & 4)) {
            iir_segment_fir<D>(v, x, G);
        }
        if (!(dbg & 8)) wave_scan<D>(v, pow2, t);                      // v = state after the lane's segment, zero carry-in
        unsigned long long *mine = agg + (s * n_tiles + b) * D2;
        if (t == IIR_TILE - 1) {
#pragma unroll
            for (int k = 0; k < D2; k++)
                __hip_atomic_store(mine + k, (unsigned long long)__builtin_bit_cast(unsigned long long, v[k]), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        // state entering the tile
        double cv[D2];
        {
            const long j = b - 1 - t;                                  // lane t looks at tile b-1-t; tile -1 = the carried state
            const bool want = t < horizon && j >= -1;
            double a[D2];
#pragma unroll
            for (int k = 0; k < D2; k++) a[k] = 0.0;
            if (want && j == -1) {
                const double *st = state_io + s * 2 * IIR_MAX_DIM;
#pragma unroll
                for (int k = 0; k < D2; k++) a[k] = __builtin_bit_cast(double, __hip_atomic_load((const unsigned long long *)st + (k / D) * IIR_MAX_DIM + (k % D), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // the old state is in registers before we say so
                __hip_atomic_fetch_add(readers + s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            bool pending = want && j >= 0 && !(dbg & 1);
            const unsigned long long *theirs = agg + (s * n_tiles + (j >= 0 ? j : 0)) * D2;
            int guard = 0;
            while (__any(pending)) {
                if (pending) {
                    // the 2D stores land in any order: take all of them every time and check each (one round trip through
                    // the fabric per poll; watching one word first would add a second trip to the poll that succeeds)
                    bool ok = true;
#pragma unroll
                    for (int k = 0; k < D2; k++) {
                        const unsigned long long w = __hip_atomic_load(theirs + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok &= w != IIR_SENTINEL;
                        a[k] = __builtin_bit_cast(double, w);
                    }
                    if (ok && poll_bound >= 0) pending = false;
                    else if (++guard > poll_bound) {                      // never reached once the producer's wave is resident
                        __hip_atomic_fetch_add(overruns, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // pinned host word: clhip_iir_overruns()
                        pending = false;
                    }
                }
                if (__any(pending)) __builtin_amdgcn_s_sleep(2);       // ~128 cycles: leave the issue slots and the fabric to the others
            }
            // cv = a_0 + Q (a_1 + Q (a_2 + ...)): Horner over the lanes that looked, the lane's vector broadcast by
            // v_readlane, the one matrix Q by scalar loads -- no per-lane tables, nothing to reduce
            const cdouble_t *qm = (const cdouble_t *)&pl->Q[0];
#pragma unroll
            for (int k = 0; k < D2; k++) cv[k] = 0.0;
            for (int h = horizon - 1; h >= 0; h--) {                   // uniform
                double nx[D2];
#pragma unroll
                for (int k = 0; k < D2; k++) {
                    const unsigned long long bits = __builtin_bit_cast(unsigned long long, a[k]);
                    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)bits, h);
                    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(bits >> 32), h);
                    nx[k] = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
                }
                matvec<D, true>(qm, cv, nx);
                matvec<D, true>(qm, cv + D, nx + D);
#pragma unroll
                for (int k = 0; k < D2; k++) cv[k] = nx[k];
            }
        }
        // the lane's true start state: what the lanes before it left (zero carry) + P^t cv.  P^t comes from the plan's
        // per-lane table (its non-zero entries: D (D + 2) / 2 coalesced 8-byte loads from a 12 KB table that stays in
        // L2) and is used once -- one product instead of the six a P^(2^d) ladder over the bits of t costs the wave.
        // (Four biquads: 40 entries in flight would spill; they keep the ladder, on the scan's scalar tables.)
        double zi[D], zq[D];
        if constexpr (NS > 3) {
#pragma unroll 1
            for (int d = 0; d < 6; d++) {
                if (t & (1 << d)) {
                    const cdouble_t *m = pow2 + d * IIR_MSZ;
                    double nx[D2];
                    matvec<D, false>(m, cv, nx);
                    matvec<D, false>(m, cv + D, nx + D);
#pragma unroll
                    for (int k = 0; k < D2; k++) cv[k] = nx[k];
                }
            }
#pragma unroll
            for (int k = 0; k < D2; k++) {
                const double p = __shfl_up(v[k], 1, 64);
                const double st = (t == 0 ? 0.0 : p) + cv[k];
                if (k < D) zi[k] = st; else zq[k - D] = st;
            }
        } else {
            const double *ptab = &pl->ptab[0][0] + t;
#pragma unroll
            for (int r = 0; r < D; r++) {
                const double pi = __shfl_up(v[r], 1, 64), pq = __shfl_up(v[D + r], 1, 64);
                double si = t == 0 ? 0.0 : pi, sq = t == 0 ? 0.0 : pq;
#pragma unroll
                for (int cc = 0; cc < D; cc++)
                    if (iir_mat_nonzero(r, cc)) {
                        const double p = ptab[(r * IIR_MAX_DIM + cc) * IIR_TILE];
                        si = __builtin_fma(p, cv[cc], si);
                        sq = __builtin_fma(p, cv[D + cc], sq);
                    }
                zi[r] = si; zq[r] = sq;
            }
        }
        // the next tile's words go out now and land while the recursion (the longest phase) runs
#if IIR_PREFETCH
        if (Tn < total) {
            const long bn = Tn / n_streams, sn = Tn % n_streams;
            iir_tile_issue(iq + sn * stride + bn * IIR_TILE * IIR_SEG, n - bn * IIR_TILE * IIR_SEG, raw, t);
        }
#endif
        const long seg = b * IIR_TILE + t;
        if (seg < n_seg && !(dbg & 2)) {
            if (tile0 + (long)IIR_TILE * IIR_SEG <= n) iir_k3_segment<NS, true, true, B121>(c, x, IIR_SEG, zi, zq);
            else {
                const long cnt = n - seg * IIR_SEG < IIR_SEG ? n - seg * IIR_SEG : IIR_SEG;
                iir_k3_segment<NS, false, true, B121>(c, x, cnt, zi, zq);
            }
            if (seg == n_seg - 1) {
                // the stream's new carried state replaces the old one in place: wait until the first tiles (the only
                // readers of the old one; all of them older than this tile) have taken it.  Counter starts at all-ones.
                const unsigned want_readers = (unsigned)((long)horizon < n_tiles ? (long)horizon : n_tiles) - 1u;
                int spin = 0;
                while (__hip_atomic_load(readers + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want_readers) {
                    if (++spin > poll_bound) { __hip_atomic_fetch_add(overruns, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                double *so = state_io + s * 2 * IIR_MAX_DIM;
#pragma unroll
                for (int k = 0; k < D; k++) { so[k] = zi[k]; so[IIR_MAX_DIM + k] = zq[k]; }
            }
        }
        __syncthreads();
        iir_tile_store(xt, n - tile0, iir_sm, t);
        __syncthreads();                                               // the rows are free for the next tile
        T = Tn;
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
    {   // P^t per lane, entry-major
        double Pt[IIR_MSZ] = {0};
        for (int r = 0; r < dim; r++) Pt[r * IIR_MAX_DIM + r] = 1.0;
        for (int t = 0; t < IIR_TILE; t++) {
            for (int e = 0; e < IIR_MSZ; e++) pl->ptab[e][t] = Pt[e];
            mat_mul(dim, Pt, P, Pt);
        }
    }
    static_assert(IIR_GROUP == 256, "Q^GROUP = Q^(2^8)");
    memcpy(pl->qpow2[0], pl->Q, sizeof pl->Q);
    for (int d = 1; d < 14; d++) mat_mul(dim, pl->qpow2[d - 1], pl->qpow2[d - 1], pl->qpow2[d]);
    for (int r = 0; r < dim; r++) pl->qpow[0][r * IIR_MAX_DIM + r] = 1.0;
    for (int i = 1; i < IIR_GROUP; i++) mat_mul(dim, pl->qpow[i - 1], pl->Q, pl->qpow[i]);
    // Horizon of the single-pass kernel.  int16 inputs bound every reachable state component c by
    // 32768 * sum_n |h_c[n]| (h_c = impulse response of that component); a state entering tile b-H reaches tile b
    // as Q^H s, so max_r sum_c |Q^H[r][c]| smax[c] bounds what dropping it costs.  Below 1e-18 (absolute; the
    // outputs are integers and the fp64 sums themselves carry ~1e-8 of rounding) the tile may ignore it.
    pl->horizon = 0;
    {
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
        if (settled && 32768.0 * ygain < 1073741824.0) {
            double Qk[IIR_MSZ];
            memcpy(Qk, pl->Q, sizeof Qk);
            for (int k = 1; k <= IIR_HMAX; k++) {
                double worst = 0;
                for (int r = 0; r < dim; r++) {
                    double acc = 0;
                    for (int cc = 0; cc < dim; cc++) acc += fabs(Qk[r * IIR_MAX_DIM + cc]) * 65536.0 * smax[cc];
                    worst = fmax(worst, acc);
                }
                if (worst < 1e-18) { pl->horizon = k; break; }
                mat_mul(dim, Qk, pl->Q, Qk);
            }
        }
    }
}

// Transition tables per (device, filter): built once, uploaded once into a buffer the shim owns, kept for the life
// of the process (a filter's tables are 175 KB; an SDR session uses a handful).  Nothing about a plan lives in the
// caller's workspace, so a workspace freed and reallocated at the same address cannot resurrect a stale table.
struct IirPlanEntry {
    int device, n_stages;
    double sos[5 * IIR_MAX_STAGES];
    IirPlan host;                  // coef goes to K3 by value
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
    if (cache.size() >= 256) {                  // pathological filter churn: drop THIS device's tables once nothing is in
        (void)hipDeviceSynchronize();           // flight on it (entries of other devices may be in use by other threads)
        std::vector<IirPlanEntry *> keep;
        for (IirPlanEntry *e : cache) {
            if (e->device == device) { clhip_free(e->dev); delete e; }
            else keep.push_back(e);
        }
        cache.swap(keep);
    }
    IirPlanEntry *e = new (std::nothrow) IirPlanEntry();
    if (!e) { clhip_set_error("clhip_iir_cs16: out of memory"); return nullptr; }
    e->device = device; e->n_stages = n_stages;
    memcpy(e->sos, sos, sizeof(double) * 5 * n_stages);
    iir_plan_build(sos, n_stages, &e->host);
    e->dev = (IirPlan *)clhip_malloc(sizeof(IirPlan));
    if (!e->dev || hipMemcpy(e->dev, &e->host, sizeof(IirPlan), hipMemcpyHostToDevice) != hipSuccess) {
        clhip_set_error("clhip_iir_cs16: cannot place the filter tables on the device");
        clhip_free(e->dev); delete e;
        return nullptr;
    }
    cache.push_back(e);
    return e;
}

static size_t iir_var_bytes(size_t n_samples)
{
    const size_t n_seg = clhip_div_up(n_samples, IIR_SEG), n_tiles = clhip_div_up(n_seg, IIR_TILE);
    return (n_seg + 2 * n_tiles + 2 * clhip_div_up(n_tiles, IIR_GROUP) + 4) * 2 * IIR_MAX_DIM * sizeof(double);
}

extern "C" size_t clhip_iir_workspace_bytes(size_t n_samples, int n_stages)
{
    (void)n_stages;
    return 256 + iir_var_bytes(n_samples);
}

// Workgroups (= waves) the single-pass kernel launches: as many as the device keeps resident at once, per kernel
// instantiation and device.  Never more: a rank's first tiles wait for the aggregates of the ranks before it, so every
// launched wave must be running (a wave that waits for a slot would be waited for by the waves that hold the slots).
template <class K>
static int iir_resident_waves(K kernel, int slot)
{
    static std::mutex mu;
    static int cached[32][64];
    int device = 0, cus = 256, per_cu = 0;
    (void)hipGetDevice(&device);
    std::lock_guard<std::mutex> lock(mu);
    if (device >= 0 && device < 64 && cached[slot][device]) return cached[slot][device];
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, IIR_TILE, IIR_LDS_WORDS * 4) != hipSuccess || per_cu < 1) per_cu = 1;
    if (per_cu > 4) per_cu -= per_cu % 4;                        // the same number of waves on each of the CU's four SIMDs: the ring
                                                                 // advances at the pace of its slowest wave, and a SIMD with one
                                                                 // wave more than the others sets that pace for everybody
    const char *e = getenv("CLHIP_IIR_WG_PER_CU");               // experiment knob: fewer than the device would hold
    if (e && atoi(e) > 0 && atoi(e) < per_cu) per_cu = atoi(e);
    if (device >= 0 && device < 64) cached[slot][device] = cus * per_cu;
    return cus * per_cu;
}

// The single-pass kernel's polls are bounded (a wave waits only for waves that are running, so the bound is never
// reached on a GPU the launch has to itself; a launch squeezed to a handful of resident waves by other work could reach
// it).  A poll that gives up counts itself in one word of pinned, device-mapped host memory per device; the results of
// that call are then wrong and clhip_iir_overruns() says so once the stream has been synchronised.
static unsigned int *iir_overrun_word(unsigned int **dev_ptr)
{
    static std::mutex mu;
    static unsigned int *host[64], *dev[64];
    int device = 0;
    (void)hipGetDevice(&device);
    if (device < 0 || device >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!host[device]) {
        unsigned int *h = nullptr, *d = nullptr;
        if (hipHostMalloc((void **)&h, 64, hipHostMallocMapped) != hipSuccess) return nullptr;
        *h = 0;
        if (hipHostGetDevicePointer((void **)&d, h, 0) != hipSuccess) { (void)hipHostFree(h); return nullptr; }
        host[device] = h; dev[device] = d;
    }
    if (dev_ptr) *dev_ptr = dev[device];
    return host[device];
}

extern "C" int clhip_iir_overruns(void)
{
    unsigned int *h = iir_overrun_word(nullptr);
    if (!h) return 0;
    const unsigned int n = __atomic_exchange_n(h, 0u, __ATOMIC_RELAXED);
    return n > 0x7fffffffu ? 0x7fffffff : (int)n;
}

template <int NS>
static int iir_launch_onepass(const IirPlan *d_plan, const IirPlan &plan, double *d_state, uint32_t *d_iq, long stride, long n,
                              int n_streams, double *ws, hipStream_t s)
{
    constexpr int D2 = 4 * NS;
    const long n_seg = (long)clhip_div_up((size_t)n, IIR_SEG), n_tiles = (long)clhip_div_up((size_t)n_seg, IIR_TILE);
    // workspace, all of it pre-set to all-ones by ONE memset: [64 rank counters]
    // [per stream: readers of the old carried state][aggregates: n_streams x n_tiles x 2D]
    unsigned int *ticket = (unsigned int *)ws;
    unsigned int *d_overruns = nullptr;
    if (!iir_overrun_word(&d_overruns)) { clhip_set_error("clhip_iir_cs16: cannot map the overrun counter"); return -1; }
    const char *pb = getenv("CLHIP_IIR_POLL_BOUND");                  // tests force the give-up path with -1
    const int poll_bound = pb ? atoi(pb) : (1 << 20);
    constexpr size_t TK = IIR_RANK_CLASSES * sizeof(unsigned int) / sizeof(double);
    unsigned int *readers = (unsigned int *)(ws + TK);
    const size_t rd_doubles = ((size_t)n_streams + 1) / 2;
    unsigned long long *agg = (unsigned long long *)(ws + TK + rd_doubles);
    CLHIP_CHECK(hipMemsetAsync(ws, 0xFF, sizeof(double) * (TK + rd_doubles + (size_t)n_tiles * n_streams * D2), s));
    const long total = n_tiles * n_streams;
    bool b121 = true;                       // b = (1, 2, 1) exactly in every stage after the first: the four-operation stage form
    for (int k = 1; k < NS; k++) b121 = b121 && plan.coef.b0[k] == 1.0 && plan.coef.b1[k] == 2.0 && plan.coef.b2[k] == 1.0;
    const int dbg = getenv("CLHIP_IIR_DBG") ? atoi(getenv("CLHIP_IIR_DBG")) : 0;                       // timing ablations only (results invalid)
    const bool unit_b = b121 && NS > 1 && !(dbg & 16);
    const int resident = unit_b ? iir_resident_waves(iir_onepass_kernel<NS, true>, 2 * NS) : iir_resident_waves(iir_onepass_kernel<NS, false>, 2 * NS + 1);
    const unsigned grid = (unsigned)(total < resident ? total : resident);
    const int stagger = total >= 2L * grid ? (getenv("CLHIP_IIR_STAGGER_US") ? atoi(getenv("CLHIP_IIR_STAGGER_US")) : 0) * 100 : 0;   // experiment knob: measured neutral
    if (unit_b)
        hipLaunchKernelGGL((iir_onepass_kernel<NS, true>), dim3(grid), dim3(IIR_TILE), IIR_LDS_WORDS * 4, s, d_plan, plan.coef, d_iq,
                           stride, n, n_seg, n_tiles, n_streams, ticket, agg, readers, d_state, plan.horizon, d_overruns, poll_bound, dbg, stagger);
    else
        hipLaunchKernelGGL((iir_onepass_kernel<NS, false>), dim3(grid), dim3(IIR_TILE), IIR_LDS_WORDS * 4, s, d_plan, plan.coef, d_iq,
                           stride, n, n_seg, n_tiles, n_streams, ticket, agg, readers, d_state, plan.horizon, d_overruns, poll_bound, dbg, stagger);
    return 0;
}

template <int NS>
static void iir_launch(const IirPlan *d_plan, const IirCoef &coef, double *d_state, uint32_t *d_iq, long stride, long n,
                       int n_streams, double *ws, hipStream_t s)
{
    constexpr int D2 = 4 * NS;               // doubles per state pair
    const long n_seg = (long)clhip_div_up((size_t)n, IIR_SEG), n_tiles = (long)clhip_div_up((size_t)n_seg, IIR_TILE);
    const long n_groups = (long)clhip_div_up((size_t)n_tiles, IIR_GROUP);
    double *ZS = ws, *tend = ZS + n_seg * n_streams * D2, *X = tend + n_tiles * n_streams * D2;
    double *gend = X + n_tiles * n_streams * D2, *gc = gend + n_groups * n_streams * D2;
    dim3 grid((unsigned)n_tiles, n_streams), block(IIR_TILE);
    static_assert(IIR_TILE * (2 * IIR_MAX_DIM + 1) * 8 <= IIR_LDS_WORDS * 4, "the scan exchange fits the tile's LDS");
    hipLaunchKernelGGL(iir_k1_kernel<NS>, grid, block, IIR_LDS_WORDS * 4, s, d_plan, (const uint32_t *)d_iq, stride, n, n_seg,
                       n_tiles, ZS, tend);
    hipLaunchKernelGGL(iir_k2a_kernel<NS>, dim3((unsigned)n_groups, n_streams), dim3(IIR_GROUP), 0, s, d_plan, n_tiles, n_groups,
                       (const double *)tend, X, gend);
    hipLaunchKernelGGL(iir_k2b_kernel<NS>, dim3((unsigned)n_streams), dim3(64), 0, s, d_plan, n_groups, (const double *)gend, gc,
                       (const double *)d_state);
    hipLaunchKernelGGL(iir_k3_kernel<NS>, grid, block, IIR_LDS_WORDS * 4, s, d_plan, coef, d_iq, stride, n, n_seg, n_tiles,
                       n_groups, (const double *)ZS, (const double *)X, (const double *)gc, d_state);
}

// d_state: 2*IIR_MAX_DIM doubles per stream, layout [rail][2*stage + {0:v1,1:v2}]
extern "C" int clhip_iir_cs16_batch(const double *h_sos, int n_stages, double *d_state, int16_t *d_iq,
                                    size_t stride_samples, size_t n_samples, int n_streams, void *d_ws,
                                    size_t ws_bytes, void *stream)
{
    if (n_samples == 0 || n_streams <= 0) return 0;
    if (!h_sos || n_stages < 1 || n_stages > IIR_MAX_STAGES || !d_state || !d_iq || !d_ws) {
        clhip_set_error("clhip_iir_cs16: bad arguments (1..%d biquads)", IIR_MAX_STAGES);
        return -1;
    }
    const size_t need = 256 + iir_var_bytes(n_samples) * n_streams;
    if (ws_bytes < need) {
        clhip_set_error("clhip_iir_cs16: workspace too small (%zu < %zu)", ws_bytes, need);
        return -1;
    }
    hipStream_t s = (hipStream_t)stream;
    const IirPlanEntry *pe = iir_plan_for(h_sos, n_stages);
    if (!pe) return -1;
    const IirPlan &plan = pe->host;
    const IirPlan *d_plan = pe->dev;
    double *wsv = (double *)d_ws;
    // single pass unless the filter's memory is too long for it (or CLHIP_IIR_ONEPASS=0: the four-kernel scan, A/B)
    static const int onepass_env = getenv("CLHIP_IIR_ONEPASS") ? atoi(getenv("CLHIP_IIR_ONEPASS")) : 1;
    if (onepass_env && plan.horizon >= 1 && plan.horizon <= IIR_HMAX) {
        int rc;
        switch (n_stages) {
        case 1: rc = iir_launch_onepass<1>(d_plan, plan, d_state, (uint32_t *)d_iq, (long)stride_samples, (long)n_samples, n_streams, wsv, s); break;
        case 2: rc = iir_launch_onepass<2>(d_plan, plan, d_state, (uint32_t *)d_iq, (long)stride_samples, (long)n_samples, n_streams, wsv, s); break;
        case 3: rc = iir_launch_onepass<3>(d_plan, plan, d_state, (uint32_t *)d_iq, (long)stride_samples, (long)n_samples, n_streams, wsv, s); break;
        default: rc = iir_launch_onepass<4>(d_plan, plan, d_state, (uint32_t *)d_iq, (long)stride_samples, (long)n_samples, n_streams, wsv, s); break;
        }
        if (rc) return -1;
        CLHIP_CHECK_LAUNCH();
        return 0;
    }
    switch (n_stages) {
    case 1: iir_launch<1>(d_plan, plan.coef, d_state, (uint32_t *)d_iq, (long)stride_samples, (long)n_samples, n_streams, wsv, s); break;
    case 2: iir_launch<2>(d_plan, plan.coef, d_state, (uint32_t *)d_iq, (long)stride_samples, (long)n_samples, n_streams, wsv, s); break;
    case 3: iir_launch<3>(d_plan, plan.coef, d_state, (uint32_t *)d_iq, (long)stride_samples, (long)n_samples, n_streams, wsv, s); break;
    default: iir_launch<4>(d_plan, plan.coef, d_state, (uint32_t *)d_iq, (long)stride_samples, (long)n_samples, n_streams, wsv, s); break;
    }
    CLHIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int clhip_iir_cs16(const double *h_sos, int n_stages, double *d_state, int16_t *d_iq, size_t n_samples,
                              void *d_ws, size_t ws_bytes, void *stream)
{
    return clhip_iir_cs16_batch(h_sos, n_stages, d_state, d_iq, n_samples, n_samples, 1, d_ws, ws_bytes, stream);
}
