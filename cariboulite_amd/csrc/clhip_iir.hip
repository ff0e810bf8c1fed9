// clhip_iir.hip -- the optional RX low-pass of Stream::ReadSamples
// (soapy_api/CaribouliteStream.cpp:291-298): per rail
//     y = (int16_t)(float) LP6( (float)x ),   LP6 = iir1 Butterworth::LowPass<6>
// i.e. a cascade of Direct-Form-II biquads evaluated in fp64 with state carried
// for the life of the stream.  fp32 state fails the 1e-5 bar at the reference's
// narrow cut-offs, so everything here is fp64.
//
// The recursion is strictly sequential per rail; it is made parallel as a
// blocked linear-recurrence scan over the cascade's 2*n_stages-dim state z:
//     z[n] = F z[n-1] + g x[n]
//   pass 1  every lane runs a SEG-sample segment from zero state  -> zero-state end vector
//   pass 2a per tile of 256 segments: Kogge-Stone scan with P^(2^d), P = F^SEG  (LDS)
//   pass 2b per stream: tiles chained sequentially with Q = P^256             (one lane)
//   pass 3  every lane rebuilds its true start state (scan result + P^i * tile carry),
//           re-runs its segment and writes the truncated int16 outputs in place
// F, P^(2^d), P^i and Q are built on the host in fp64 by simulating the cascade.
#include <math.h>
#include <string.h>

#include "clhip_common.h"

#define IIR_MAX_STAGES 4
#define IIR_MAX_DIM (2 * IIR_MAX_STAGES)
#define IIR_SEG 64
#define IIR_TILE 256

struct IirCoef {
    int n_stages, dim;
    double b0[IIR_MAX_STAGES], b1[IIR_MAX_STAGES], b2[IIR_MAX_STAGES], a1[IIR_MAX_STAGES], a2[IIR_MAX_STAGES];
};

// one cascade step on a DF-II state (v1,v2 per stage); returns the output
__host__ __device__ __forceinline__ double iir_step(const IirCoef &c, double *z, double in)
{
    double out = in;
#pragma unroll
    for (int s = 0; s < IIR_MAX_STAGES; s++) {
        if (s < c.n_stages) {
            const double w = out - c.a1[s] * z[2 * s] - c.a2[s] * z[2 * s + 1];
            out = c.b0[s] * w + c.b1[s] * z[2 * s] + c.b2[s] * z[2 * s + 1];
            z[2 * s + 1] = z[2 * s];
            z[2 * s] = w;
        }
    }
    return out;
}

// (int16_t)(float)y with the x86 conversion semantics of the reference build
__device__ __forceinline__ uint32_t iir_to_i16(double y)
{
    const float f = (float)y;
    const int t = (f >= -2147483648.0f && f < 2147483648.0f) ? (int)f : (int)0x80000000;
    return (uint32_t)t & 0xFFFFu;
}

// 8x8 mat-vec; rows/columns beyond the filter's dimension are zero in the tables
__device__ __forceinline__ void matvec(const double *__restrict__ m, const double *v, double *out)
{
#pragma unroll
    for (int r = 0; r < IIR_MAX_DIM; r++) {
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < IIR_MAX_DIM; c++) s += m[r * IIR_MAX_DIM + c] * v[c];
        out[r] = s;
    }
}

// A tile is IIR_TILE segments x IIR_SEG samples = 16384 int16 pairs (64 KiB).  Lanes walk their own
// segment sequentially, so direct global access would touch 64 cache lines per load instruction;
// instead the tile is copied through LDS: coalesced 16-byte global accesses on one side, a
// (IIR_SEG+1)-dword row pitch on the other so that lane t reading word k of row t hits bank (t+k)%32.
#define IIR_PITCH (IIR_SEG + 1)
#define IIR_LDS_WORDS (IIR_TILE * IIR_PITCH)

__device__ __forceinline__ void iir_tile_load(const uint32_t *__restrict__ x, long n_left, uint32_t *sm, int t)
{
    // n_left = samples of this stream from the tile start (>= 1); words beyond it are not read
    constexpr int TOTAL = IIR_TILE * IIR_SEG;
    const bool vec = ((uintptr_t)x & 15) == 0;
    for (int i = t * 4; i < TOTAL; i += IIR_TILE * 4) {
        uint32_t w[4] = {0, 0, 0, 0};
        if (vec && i + 4 <= n_left) {
            const u32x4 v = *(const u32x4 *)(x + i);
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        } else {
            for (int k = 0; k < 4; k++) if (i + k < n_left) w[k] = x[i + k];
        }
        const int row = i / IIR_SEG, col = i % IIR_SEG;          // 4 consecutive words stay in one row
        for (int k = 0; k < 4; k++) sm[row * IIR_PITCH + col + k] = w[k];
    }
}

__device__ __forceinline__ void iir_tile_store(uint32_t *__restrict__ x, long n_left, const uint32_t *sm, int t)
{
    constexpr int TOTAL = IIR_TILE * IIR_SEG;
    const bool vec = ((uintptr_t)x & 15) == 0;
    for (int i = t * 4; i < TOTAL; i += IIR_TILE * 4) {
        const int row = i / IIR_SEG, col = i % IIR_SEG;
        uint32_t w[4];
        for (int k = 0; k < 4; k++) w[k] = sm[row * IIR_PITCH + col + k];
        if (vec && i + 4 <= n_left) {
            u32x4 v = {w[0], w[1], w[2], w[3]};
            *(u32x4 *)(x + i) = v;
        } else {
            for (int k = 0; k < 4; k++) if (i + k < n_left) x[i + k] = w[k];
        }
    }
}

// pass 1: zero-state response of every segment.  ws_seg[stream][seg][rail][dim]
__global__ __launch_bounds__(IIR_TILE) void iir_pass1_kernel(IirCoef c, const uint32_t *__restrict__ iq, long stride,
                                                           long n, long n_seg, double *__restrict__ ws_seg)
{
    extern __shared__ uint32_t iir_sm[];
    const int t = threadIdx.x;
    const long tile0 = (long)blockIdx.x * IIR_TILE * IIR_SEG;
    iir_tile_load(iq + (long)blockIdx.y * stride + tile0, n - tile0, iir_sm, t);
    __syncthreads();
    const long seg = (long)blockIdx.x * IIR_TILE + t;
    if (seg >= n_seg) return;
    const uint32_t *x = iir_sm + t * IIR_PITCH;
    const long cnt = n - seg * IIR_SEG < IIR_SEG ? n - seg * IIR_SEG : IIR_SEG;
    double zi[IIR_MAX_DIM], zq[IIR_MAX_DIM];
#pragma unroll
    for (int k = 0; k < IIR_MAX_DIM; k++) { zi[k] = 0.0; zq[k] = 0.0; }
    for (long k = 0; k < cnt; k++) {
        const uint32_t w = x[k];
        (void)iir_step(c, zi, (double)(int16_t)(w & 0xFFFF));
        (void)iir_step(c, zq, (double)(int16_t)(w >> 16));
    }
    double *o = ws_seg + ((long)blockIdx.y * n_seg + seg) * 2 * IIR_MAX_DIM;
#pragma unroll
    for (int k = 0; k < IIR_MAX_DIM; k++) { o[k] = zi[k]; o[IIR_MAX_DIM + k] = zq[k]; }
}

// Kogge-Stone inclusive scan over the 256 lanes of a workgroup for the recurrence
//   v_i <- v_i + M^(2^d) v_(i - 2^d),   pow2[d] = M^(2^d)
// leaving v_i = sum_{j<=i} M^(i-j) v_j (both rails).  sh: [IIR_TILE][2*IIR_MAX_DIM+1] doubles.
__device__ __forceinline__ void ks_scan(double (&v)[2 * IIR_MAX_DIM], const double *__restrict__ pow2,
                                        double (*sh)[2 * IIR_MAX_DIM + 1], int t)
{
    for (int d = 0; d < 8; d++) {
#pragma unroll
        for (int k = 0; k < 2 * IIR_MAX_DIM; k++) sh[t][k] = v[k];
        __syncthreads();
        const int src = t - (1 << d);
        if (src >= 0) {
            double pv[2 * IIR_MAX_DIM], add[IIR_MAX_DIM];
#pragma unroll
            for (int k = 0; k < 2 * IIR_MAX_DIM; k++) pv[k] = sh[src][k];
            const double *m = pow2 + d * IIR_MAX_DIM * IIR_MAX_DIM;
            matvec(m, pv, add);
#pragma unroll
            for (int k = 0; k < IIR_MAX_DIM; k++) v[k] += add[k];
            matvec(m, pv + IIR_MAX_DIM, add);
#pragma unroll
            for (int k = 0; k < IIR_MAX_DIM; k++) v[IIR_MAX_DIM + k] += add[k];
        }
        __syncthreads();
    }
}

// pass 2a: inclusive scan inside each tile (zero carry), in place:
//   E[i] = sum_{j<=i} P^(i-j) zs[j]     pow2[d] = P^(2^d)
__global__ __launch_bounds__(IIR_TILE) void iir_pass2a_kernel(const double *__restrict__ pow2, long n_seg,
                                                            double *__restrict__ ws_seg)
{
    __shared__ double sh[IIR_TILE][2 * IIR_MAX_DIM + 1];
    const int t = threadIdx.x;
    const long seg = (long)blockIdx.x * IIR_TILE + t;
    double *o = ws_seg + ((long)blockIdx.y * n_seg + seg) * 2 * IIR_MAX_DIM;
    double v[2 * IIR_MAX_DIM];
#pragma unroll
    for (int k = 0; k < 2 * IIR_MAX_DIM; k++) v[k] = seg < n_seg ? o[k] : 0.0;
    ks_scan(v, pow2, sh, t);
    if (seg < n_seg) {
#pragma unroll
        for (int k = 0; k < 2 * IIR_MAX_DIM; k++) o[k] = v[k];
    }
}

// pass 2b: chain the tiles of one stream.  carry[k] = state entering tile k:
//   carry[0] = state_in,  carry[k+1] = Q carry[k] + E[last segment of tile k],  Q = P^TILE
// The same scan one level up: 256 tiles per step with Q^(2^d), then Q^(i+1) times the step's
// carry-in.  One workgroup per stream; a 2^26-sample stream is 16 steps.
__global__ __launch_bounds__(IIR_TILE) void iir_pass2b_kernel(const double *__restrict__ qpow2 /* Q^(2^d) */,
                                                            const double *__restrict__ qpow /* Q^i, i<=TILE */,
                                                            long n_seg, long n_tiles, const double *__restrict__ ws_seg,
                                                            double *__restrict__ carry, const double *__restrict__ state)
{
    __shared__ double sh[IIR_TILE][2 * IIR_MAX_DIM + 1];
    __shared__ double cc[2 * IIR_MAX_DIM];
    const int s = blockIdx.x, t = threadIdx.x;
    if (t < 2 * IIR_MAX_DIM) cc[t] = state[(long)s * 2 * IIR_MAX_DIM + t];
    __syncthreads();
    for (long base = 0; base < n_tiles; base += IIR_TILE) {
        const long tile = base + t;
        double v[2 * IIR_MAX_DIM];
        // E of the tile's last segment (a ragged last tile only feeds the unused carry after the stream)
        const long last = (tile + 1) * IIR_TILE - 1 < n_seg ? (tile + 1) * IIR_TILE - 1 : n_seg - 1;
#pragma unroll
        for (int k = 0; k < 2 * IIR_MAX_DIM; k++)
            v[k] = tile < n_tiles ? ws_seg[((long)s * n_seg + last) * 2 * IIR_MAX_DIM + k] : 0.0;
        ks_scan(v, qpow2, sh, t);
        // state entering tile+1 = v + Q^(t+1) * carry-in of this step
        double c[2 * IIR_MAX_DIM], add[IIR_MAX_DIM];
#pragma unroll
        for (int k = 0; k < 2 * IIR_MAX_DIM; k++) c[k] = cc[k];
        const double *m = qpow + (long)(t + 1) * IIR_MAX_DIM * IIR_MAX_DIM;
        matvec(m, c, add);
#pragma unroll
        for (int k = 0; k < IIR_MAX_DIM; k++) v[k] += add[k];
        matvec(m, c + IIR_MAX_DIM, add);
#pragma unroll
        for (int k = 0; k < IIR_MAX_DIM; k++) v[IIR_MAX_DIM + k] += add[k];
        double *cr = carry + ((long)s * (n_tiles + 1)) * 2 * IIR_MAX_DIM;
        if (t == 0) {
#pragma unroll
            for (int k = 0; k < 2 * IIR_MAX_DIM; k++) cr[base * 2 * IIR_MAX_DIM + k] = c[k];
        }
        if (tile + 1 <= n_tiles && tile < n_tiles) {
#pragma unroll
            for (int k = 0; k < 2 * IIR_MAX_DIM; k++) cr[(tile + 1) * 2 * IIR_MAX_DIM + k] = v[k];
        }
        __syncthreads();
        if (t == IIR_TILE - 1) {
#pragma unroll
            for (int k = 0; k < 2 * IIR_MAX_DIM; k++) cc[k] = v[k];
        }
        __syncthreads();
    }
}

// pass 3: true start state per segment, re-run, write int16 in place; the lane
// owning the last segment also writes the stream's new carried state.
__global__ __launch_bounds__(IIR_TILE) void iir_pass3_kernel(IirCoef c, uint32_t *__restrict__ iq, long stride, long n,
                                                           long n_seg, long n_tiles, const double *__restrict__ ppow,
                                                           const double *__restrict__ ws_seg,
                                                           const double *__restrict__ carry, double *__restrict__ state)
{
    extern __shared__ uint32_t iir_sm[];
    const int t = threadIdx.x;
    const long tile0 = (long)blockIdx.x * IIR_TILE * IIR_SEG;
    uint32_t *xt = iq + (long)blockIdx.y * stride + tile0;
    iir_tile_load(xt, n - tile0, iir_sm, t);
    __syncthreads();
    const long seg = (long)blockIdx.x * IIR_TILE + t;
    if (seg < n_seg) {
        const double *cr = carry + ((long)blockIdx.y * (n_tiles + 1) + blockIdx.x) * 2 * IIR_MAX_DIM;
        const double *m = ppow + (long)t * IIR_MAX_DIM * IIR_MAX_DIM;          // P^t
        double zi[IIR_MAX_DIM], zq[IIR_MAX_DIM], cv[IIR_MAX_DIM];
#pragma unroll
        for (int k = 0; k < IIR_MAX_DIM; k++) { zi[k] = 0.0; zq[k] = 0.0; }
#pragma unroll
        for (int k = 0; k < IIR_MAX_DIM; k++) cv[k] = cr[k];
        matvec(m, cv, zi);
#pragma unroll
        for (int k = 0; k < IIR_MAX_DIM; k++) cv[k] = cr[IIR_MAX_DIM + k];
        matvec(m, cv, zq);
        if (t > 0) {
            const double *e = ws_seg + ((long)blockIdx.y * n_seg + seg - 1) * 2 * IIR_MAX_DIM;
#pragma unroll
            for (int k = 0; k < IIR_MAX_DIM; k++) { zi[k] += e[k]; zq[k] += e[IIR_MAX_DIM + k]; }
        }
        uint32_t *x = iir_sm + t * IIR_PITCH;
        const long cnt = n - seg * IIR_SEG < IIR_SEG ? n - seg * IIR_SEG : IIR_SEG;
        for (long k = 0; k < cnt; k++) {
            const uint32_t w = x[k];
            // filter((float)x): int16 -> float -> double is exact
            const double yi = iir_step(c, zi, (double)(int16_t)(w & 0xFFFF));
            const double yq = iir_step(c, zq, (double)(int16_t)(w >> 16));
            x[k] = iir_to_i16(yi) | (iir_to_i16(yq) << 16);
        }
        if (seg == n_seg - 1) {
            double *st = state + (long)blockIdx.y * 2 * IIR_MAX_DIM;
#pragma unroll
            for (int k = 0; k < IIR_MAX_DIM; k++) { st[k] = zi[k]; st[IIR_MAX_DIM + k] = zq[k]; }
        }
    }
    __syncthreads();
    iir_tile_store(xt, n - tile0, iir_sm, t);
}

// ---------------------------------------------------------------------------
// host: transition matrices by simulating the cascade
// ---------------------------------------------------------------------------
static void mat_mul(int dim, const double *a, const double *b, double *o)
{
    double t[IIR_MAX_DIM * IIR_MAX_DIM] = {0};
    for (int r = 0; r < dim; r++)
        for (int c = 0; c < dim; c++) {
            double s = 0;
            for (int k = 0; k < dim; k++) s += a[r * IIR_MAX_DIM + k] * b[k * IIR_MAX_DIM + c];
            t[r * IIR_MAX_DIM + c] = s;
        }
    memcpy(o, t, sizeof t);
}

struct IirPlan {
    IirCoef coef;
    double pow2[8][IIR_MAX_DIM * IIR_MAX_DIM];              // P^(2^d)
    double ppow[IIR_TILE + 1][IIR_MAX_DIM * IIR_MAX_DIM];   // P^i
    double qpow2[8][IIR_MAX_DIM * IIR_MAX_DIM];             // Q^(2^d), Q = P^TILE
    double qpow[IIR_TILE + 1][IIR_MAX_DIM * IIR_MAX_DIM];   // Q^i
};

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
    double F[IIR_MAX_DIM * IIR_MAX_DIM] = {0};
    for (int j = 0; j < dim; j++) {          // column j = one zero-input step from e_j
        double z[IIR_MAX_DIM] = {0};
        z[j] = 1.0;
        (void)iir_step(c, z, 0.0);
        for (int r = 0; r < dim; r++) F[r * IIR_MAX_DIM + j] = z[r];
    }
    double P[IIR_MAX_DIM * IIR_MAX_DIM];
    memcpy(P, F, sizeof P);
    for (int k = 0; k < 6; k++) mat_mul(dim, P, P, P);     // F^64 (IIR_SEG = 64)
    static_assert(IIR_SEG == 64, "P = F^SEG is built by six squarings");
    memcpy(pl->pow2[0], P, sizeof P);
    for (int d = 1; d < 8; d++) mat_mul(dim, pl->pow2[d - 1], pl->pow2[d - 1], pl->pow2[d]);
    for (int r = 0; r < dim; r++) pl->ppow[0][r * IIR_MAX_DIM + r] = 1.0;
    for (int i = 1; i <= IIR_TILE; i++) mat_mul(dim, pl->ppow[i - 1], P, pl->ppow[i]);
    const double *Q = pl->ppow[IIR_TILE];
    memcpy(pl->qpow2[0], Q, sizeof pl->qpow2[0]);
    for (int d = 1; d < 8; d++) mat_mul(dim, pl->qpow2[d - 1], pl->qpow2[d - 1], pl->qpow2[d]);
    for (int r = 0; r < dim; r++) pl->qpow[0][r * IIR_MAX_DIM + r] = 1.0;
    for (int i = 1; i <= IIR_TILE; i++) mat_mul(dim, pl->qpow[i - 1], Q, pl->qpow[i]);
}

extern "C" size_t clhip_iir_workspace_bytes(size_t n_samples, int n_stages)
{
    (void)n_stages;
    const size_t n_seg = clhip_div_up(n_samples, IIR_SEG), n_tiles = clhip_div_up(n_seg, IIR_TILE);
    return sizeof(IirPlan) + 256 + (n_seg + n_tiles + 2) * 2 * IIR_MAX_DIM * sizeof(double);
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
    const size_t n_seg = clhip_div_up(n_samples, IIR_SEG), n_tiles = clhip_div_up(n_seg, IIR_TILE);
    const size_t need = sizeof(IirPlan) + 256 + (n_seg + n_tiles + 2) * 2 * IIR_MAX_DIM * sizeof(double) * n_streams;
    if (ws_bytes < need) {
        clhip_set_error("clhip_iir_cs16: workspace too small (%zu < %zu)", ws_bytes, need);
        return -1;
    }
    hipStream_t s = (hipStream_t)stream;
    static thread_local IirPlan plan;        // stays valid until the async copy below has been enqueued+run
    static thread_local double last_sos[5 * IIR_MAX_STAGES];
    static thread_local int last_n = 0;
    static thread_local void *last_ws = nullptr;
    unsigned char *ws = (unsigned char *)d_ws;
    IirPlan *d_plan = (IirPlan *)ws;
    const bool changed = last_n != n_stages || memcmp(last_sos, h_sos, sizeof(double) * 5 * n_stages);
    if (changed) {
        iir_plan_build(h_sos, n_stages, &plan);
        memcpy(last_sos, h_sos, sizeof(double) * 5 * n_stages);
        last_n = n_stages;
    }
    if (changed || last_ws != d_ws) {
        // tables live at the head of the workspace; re-sent only when the filter or the workspace changes
        // (a pageable-source copy is staged by the runtime before the call returns)
        CLHIP_CHECK(hipMemcpyAsync(d_plan, &plan, sizeof plan, hipMemcpyHostToDevice, s));
        last_ws = d_ws;
    }
    double *ws_seg = (double *)(ws + ((sizeof(IirPlan) + 255) & ~(size_t)255));
    double *carry = ws_seg + n_seg * n_streams * 2 * IIR_MAX_DIM;
    const double *d_pow2 = &d_plan->pow2[0][0];
    const double *d_ppow = &d_plan->ppow[0][0];
    const double *d_qpow2 = &d_plan->qpow2[0][0];
    const double *d_qpow = &d_plan->qpow[0][0];
    dim3 grid((unsigned)n_tiles, n_streams), block(IIR_TILE);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void *)iir_pass1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, IIR_LDS_WORDS * 4);
        (void)hipFuncSetAttribute((const void *)iir_pass3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, IIR_LDS_WORDS * 4);
        attr = true;
    }
    hipLaunchKernelGGL(iir_pass1_kernel, grid, block, IIR_LDS_WORDS * 4, s, plan.coef, (const uint32_t *)d_iq, (long)stride_samples,
                       (long)n_samples, (long)n_seg, ws_seg);
    hipLaunchKernelGGL(iir_pass2a_kernel, grid, block, 0, s, d_pow2, (long)n_seg, ws_seg);
    hipLaunchKernelGGL(iir_pass2b_kernel, dim3(n_streams), dim3(IIR_TILE), 0, s, d_qpow2, d_qpow, (long)n_seg,
                       (long)n_tiles, ws_seg, carry, (const double *)d_state);
    hipLaunchKernelGGL(iir_pass3_kernel, grid, block, IIR_LDS_WORDS * 4, s, plan.coef, (uint32_t *)d_iq, (long)stride_samples,
                       (long)n_samples, (long)n_seg, (long)n_tiles, d_ppow, ws_seg, carry, d_state);
    CLHIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int clhip_iir_cs16(const double *h_sos, int n_stages, double *d_state, int16_t *d_iq, size_t n_samples,
                              void *d_ws, size_t ws_bytes, void *stream)
{
    return clhip_iir_cs16_batch(h_sos, n_stages, d_state, d_iq, n_samples, n_samples, 1, d_ws, ws_bytes, stream);
}
