/*
 * cariboulite_hip.h -- C-ABI of the MI355X (gfx950) sample-stream hot path.
 *
 * Two layers, both plain C (pointers and sizes only; no HIP, torch or C++
 * types cross this boundary; `void *stream` is a hipStream_t, NULL = the
 * default stream):
 *
 *   clhip_*   libcariboulite_hip.so -- the thin shim over the hand-written HIP
 *             kernels.  Every data pointer is a DEVICE pointer unless the
 *             name says `h_`.  Launches are asynchronous on `stream`.
 *   cl_* /    libcariboulite_host.so -- host C code that keeps the reference's
 *   caribou_  own call surface: the SMI user-driver seam, the radio
 *   cariboul  pass-through trio, and the SoapySDR device/stream calls, with the
 *             /dev/smi fd replaced by an injected byte stream.
 *
 * Each entry point cites the reference interface it replaces (paths relative
 * to /root/reference/software/libcariboulite/src).  INTEGRATION.md shows the
 * reference-side binding.
 */
#ifndef CARIBOULITE_HIP_H
#define CARIBOULITE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------ */
/* Types shared with the reference                                          */
/* ------------------------------------------------------------------------ */

/* caribou_smi/caribou_smi.h:53-65, cariboulite_radio.h:119-128 */
#pragma pack(push, 1)
typedef struct { int16_t i; int16_t q; } cl_sample_complex_int16;
typedef struct { uint8_t sync; } cl_sample_meta;
typedef struct { int8_t i; int8_t q; } cl_sample_complex_int8;     /* soapy_api/CaribouliteStream.hpp:28-32 */
typedef struct { float i; float q; } cl_sample_complex_float;      /* :49-53 */
typedef struct { double i; double q; } cl_sample_complex_double;   /* :56-60 */
#pragma pack(pop)

#define CL_BYTES_PER_SAMPLE   4          /* caribou_smi.h:42 CARIBOU_SMI_BYTES_PER_SAMPLE */
#define CL_SAMPLE_RATE        4000000    /* caribou_smi.h:43 CARIBOU_SMI_SAMPLE_RATE      */
#define CL_NATIVE_BATCH_LEN   524288     /* caribou_smi.c:78 default native_batch_len     */
#define CL_NATIVE_MTU_SAMPLES 131072     /* cariboulite_radio.c:1310-1315                 */

/* caribou_smi.h:45-49 caribou_smi_channel_en */
#define CL_CHANNEL_S1G 0   /* caribou_smi_channel_900  */
#define CL_CHANNEL_HIF 1   /* caribou_smi_channel_2400 */

/* soapy_api/CaribouliteStream.hpp:68-74 CaribouliteFormat */
#define CL_FORMAT_CF32 0
#define CL_FORMAT_CS16 1
#define CL_FORMAT_CS8  2
#define CL_FORMAT_CF64 3

/* soapy_api/CaribouliteStream.hpp:77-84 DigitalFilterType */
#define CL_DIGFILT_NONE   0
#define CL_DIGFILT_20KHZ  1
#define CL_DIGFILT_50KHZ  2
#define CL_DIGFILT_100KHZ 3

/* caribou_smi.c:653-675 return codes of caribou_smi_read */
#define CL_SMI_ERR_IO        (-1)
#define CL_SMI_ERR_DEBUGMODE (-2)
#define CL_SMI_ERR_SYNC      (-3)

/* SoapySDR/Errors.h values the reference returns (CaribouliteStreamFunctions.cpp:248-251) */
#define CL_SOAPY_SDR_RX 1
#define CL_SOAPY_SDR_TX 0
#define CL_SOAPY_SDR_NOT_SUPPORTED (-5)
#define CL_SOAPY_SDR_TIMEOUT (-1)

/* TX packer behaviour (caribou_smi.c:684-717) */
#define CL_TX_DOCUMENTED 0  /* layout of :693-696 applied to the caller's samples          */
#define CL_TX_AS_WRITTEN 1  /* :700-701 as shipped (ii=0xFFFF, qq=0): FF 7F 40 00 per sample */

/* ======================================================================== */
/* Layer 1: clhip_* -- HIP kernel shim (libcariboulite_hip.so)              */
/* ======================================================================== */

/* runtime plumbing so that C callers need no HIP headers */
int         clhip_device_count(void);
int         clhip_set_device(int device);
const char *clhip_last_error(void);
const char *clhip_arch_name(void);                  /* "gfx950" on the target       */
void       *clhip_malloc(size_t bytes);             /* device memory                */
void        clhip_free(void *d_ptr);
void       *clhip_host_alloc(size_t bytes);         /* pinned, device-visible host  */
void        clhip_host_free(void *h_ptr);
void       *clhip_host_device_ptr(void *h_ptr);     /* the address kernels use for that memory (NULL: not reachable) */
void       *clhip_host_register(void *h_ptr, size_t bytes);   /* pin + map memory the caller owns; returns the address kernels
                                                               * use, NULL when it cannot be registered */
void        clhip_host_unregister(void *h_ptr);
/* host <-> device copies on a stream.  Host memory that is not page-locked (not from clhip_host_alloc / hipHostMalloc, not
 * registered) is copied in pieces of 512 KiB: the HIP runtime would pin the caller's pages in place for a pageable copy of
 * 1 MiB and more and let the copy engine into the process's heap, which has ended long-lived processes with a GPU page
 * fault on a host address; in pieces every byte goes through the runtime's own pinned staging buffers. */
int         clhip_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes, void *stream);
int         clhip_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes, void *stream);
int         clhip_memcpy_d2d(void *d_dst, const void *d_src, size_t bytes, void *stream);
/* `height` rows of `width` bytes from page-locked host memory (rows h_pitch apart) to device rows d_pitch apart in ONE copy-engine
 * operation (what a stream group does with its members' batches when they lie at one stride in its pinned slab); refuses pageable memory */
int         clhip_memcpy2d_h2d(void *d_dst, size_t d_pitch, const void *h_src, size_t h_pitch, size_t width, size_t height, void *stream);
/* Diagnostics: the last 256 things this library asked the runtime to do with host memory it does not own -- registrations,
 * releases, copies above one piece (what kind of memory it found) -- so that a GPU page fault on a host address can be set
 * against them.  _ops copies the ring (oldest first), _ops_dump writes it as text with write(2) only (callable from a signal
 * handler: tests/cpp/abrt_trace.c does on SIGABRT), _copy_counters: [0] pageable copies made in pieces, [1] page-locked
 * copies made whole, [2] the largest pageable range ever handed to ONE runtime copy (bytes; never above 512 KiB), [3]
 * pageable bytes copied. */
#define CLHIP_OP_REGISTER        1
#define CLHIP_OP_REGISTER_FAILED 2
#define CLHIP_OP_UNREGISTER      3
#define CLHIP_OP_H2D_PIECES      4
#define CLHIP_OP_D2H_PIECES      5
#define CLHIP_OP_H2D_LOCKED      6
#define CLHIP_OP_D2H_LOCKED      7
typedef struct { uint64_t seq; uint32_t op; uint32_t pad; uint64_t base; uint64_t len; } clhip_op_record;
size_t      clhip_debug_ops(clhip_op_record *out, size_t max);
void        clhip_debug_ops_dump(int fd);
void        clhip_debug_copy_counters(uint64_t out[4]);
int         clhip_debug_sticky_error(void);          /* the runtime's pending "last error" of this thread, consumed (0: none): nothing handled may leave one behind */
int         clhip_memset(void *d_dst, int value, size_t bytes, void *stream);
void       *clhip_stream_create(void);
void        clhip_stream_destroy(void *stream);
int         clhip_stream_sync(void *stream);
/* HIP-event timing on the stream the kernels are launched on (bench.py) */
void       *clhip_event_create(void);
void        clhip_event_destroy(void *event);
int         clhip_event_record(void *event, void *stream);
float       clhip_event_elapsed_ms(void *start, void *stop); /* synchronises on stop */
int         clhip_stream_wait_event(void *stream, void *event); /* later work on `stream` waits for `event` */
int         clhip_event_sync(void *event);                       /* the host waits for `event` */

/*
 * Sync search -- replaces caribou_smi_find_buffer_offset
 * (caribou_smi/caribou_smi.c:235-292, debug_mode none).  One result per chunk:
 * d_offs[c] = smallest byte offset in [0, len_c-16) whose four unaligned LE
 * words all satisfy (w & 0xC001C000) == 0x80004000; 0 if len_c <= 16; -1 if
 * none.  Chunk c starts at d_bytes + c*chunk_stride_bytes and is
 * len_c = min(chunk_len_bytes, total_bytes - c*chunk_stride_bytes) long.
 */
int clhip_smi_find_offsets(const uint8_t *d_bytes, size_t total_bytes,
                           size_t chunk_stride_bytes, size_t chunk_len_bytes,
                           int n_chunks, int32_t *d_offs, void *stream);

/*
 * Unpack -- replaces caribou_smi_rx_data_analyze (caribou_smi.c:295-393) fused
 * with the Soapy RX conversions (soapy_api/CaribouliteStream.cpp:304-367).
 * Per chunk c (same chunking as above, offsets from clhip_smi_find_offsets):
 * words start at chunk + offs; n = (len_c - 4*short)/4, short = offs>0 ?
 * offs/4+1 : 0; sample k of chunk c lands in slot c*chunk_stride_bytes/4 + k;
 * one extrapolated sample (110*a/100 - b/10) follows when short > 0; slots
 * the reference leaves untouched are left untouched; chunks with offs < 0
 * write nothing.  d_out is in `format` (CL_FORMAT_*); d_meta may be NULL.
 */
int clhip_smi_unpack(int channel, const uint8_t *d_bytes, size_t total_bytes,
                     size_t chunk_stride_bytes, size_t chunk_len_bytes, int n_chunks,
                     const int32_t *d_offs, int format, void *d_out, uint8_t *d_meta,
                     void *stream);

/* One read() chunk KNOWN to be in sync (sync offset 0: caribou_smi_find_buffer_offset returns 0 exactly when the words at
 * byte offsets 0, 4, 8, 12 carry the pattern, caribou_smi.c:235-292 -- the host can see that in its pinned staging
 * memory): no search, every slot written; the int16 pairs go to d_cs16 (the persistent native buffer; may be NULL) and, in
 * `format`, to `out`, which may be mapped pinned host memory (clhip_host_device_ptr): the call then needs no
 * device-to-host copy.  16-byte aligned buffers, n_bytes a multiple of 16. */
int clhip_smi_unpack_aligned(int channel, const uint8_t *d_bytes, size_t n_bytes, int format, void *out,
                             int16_t *d_cs16, void *stream);

/*
 * Link-integrity (debug) modes -- replaces caribou_smi_anayze_smi_debug and the debug branches of
 * caribou_smi_find_buffer_offset (caribou_smi.c:172-215, 266-283) for one read() chunk.
 * mode = CL_SMI_DEBUG_LFSR / _PUSH / _PULL.  d_res (4 x int32, device): [0] sync offset (-1 none),
 * [1] erroneous bytes in the chunk, [2] first erroneous byte (0x7fffffff none), [3] last byte seen.
 */
#define CL_SMI_DEBUG_NONE 0   /* caribou_smi.h:22-28 caribou_smi_debug_mode_en */
#define CL_SMI_DEBUG_LFSR 1
#define CL_SMI_DEBUG_PUSH 2
#define CL_SMI_DEBUG_PULL 3
int clhip_smi_debug_analyze(int mode, const uint8_t *d_bytes, size_t len, uint32_t last_correct_byte,
                            int32_t *d_res, void *stream);

/*
 * pps tags -- replaces the per-sample host loop of the GNU Radio source block's work()
 * (software/gr-caribouLite/lib/caribouLiteSource_impl.cc:113-119: `if (out_meta[i] == 1) add_item_tag(0, i, "pps", true)`)
 * by one ordered compaction of the meta plane on the device: d_idx[0 .. min(count, cap)) = the positions i < n with
 * d_meta[i] == 1 (exactly 1, as the loop tests), ascending; *d_count = how many there are (also beyond cap).
 * d_meta may start at any byte address (a plane the unpack wrote at an offset).  n < 2^32.  Planes above 256 KiB
 * need d_ws = clhip_sync_tags_ws_bytes(n) bytes of device memory (per-tile counts; no initialisation needed).
 */
size_t clhip_sync_tags_ws_bytes(size_t n);
int clhip_sync_tags(const uint8_t *d_meta, size_t n, uint32_t *d_idx, size_t cap, uint32_t *d_count, void *d_ws, void *stream);

/* RX/TX format conversions on native CS16 (CaribouliteStream.cpp:199-244,304-367) */
int clhip_convert_from_cs16(const int16_t *d_iq, size_t n_samples, int format, void *d_out, void *stream);
int clhip_convert_to_cs16(const void *d_in, int format, size_t n_samples, int16_t *d_iq, void *stream);
/* The two steps of a non-native writeStream -- the conversion loop (CaribouliteStream.cpp:199-244) and
 * caribou_smi_generate_data (caribou_smi.c:684-717) -- in one launch; bit-identical to clhip_convert_to_cs16 + clhip_smi_pack.
 * d_bytes: 4-byte aligned, 4 * n_samples bytes.  mode = CL_TX_DOCUMENTED / CL_TX_AS_WRITTEN. */
int clhip_convert_pack(const void *d_in, int format, size_t n_samples, int mode, uint8_t *d_bytes, void *stream);
/* ... for up to CLHIP_PACK_ROWS streams in one launch (cl_group_writeStream): row r converts n_samples from d_in_rows[r] into
 * d_bytes_rows[r] (host arrays of device-visible addresses, read at the call). */
#define CLHIP_PACK_ROWS 8
int clhip_convert_pack_rows(const void *const *d_in_rows, int format, size_t n_samples, int n_rows, int mode, uint8_t *const *d_bytes_rows, void *stream);
/* Rows of packed words on the device (row r at d_words + r * in_stride_bytes) stored by ONE launch into up to CLHIP_PACK_ROWS destinations
 * of their own -- device addresses of the rooms reserved in the members' pinned TX FIFOs (cl_group_writeStream's modulator lanes: the
 * appends of caribou_smi_write's chunk loop, caribou_smi.c:738-759, for a whole sub-batch). */
int clhip_words_to_rows(const uint8_t *d_words, size_t in_stride_bytes, size_t n_words, int n_rows, uint8_t *const *d_dst_rows, void *stream);
/* ... and rows of results of any element size (lengths of their own) into destinations of their own: the clients' registered buffers of
 * a sub-batch (cl_group_register_buffers; cl_group_readStream's direct route).  Any alignment. */
int clhip_rows_to_rows(const void *const *d_src_rows, void *const *d_dst_rows, const size_t *row_bytes, int n_rows, void *stream);
/* The I rail of interleaved CF32 samples as a dense fp32 message: what Stream::WriteSamples would hand an FM modulator
 * ("if given I/Q, use I"), taken on the device instead of in a host loop. */
int clhip_take_i_rail(const float *d_cf32, size_t n_samples, float *d_msg, void *stream);

/* TX pack -- replaces caribou_smi_generate_data (caribou_smi.c:684-717) */
int clhip_smi_pack(int mode, const int16_t *d_iq, size_t n_samples, uint8_t *d_bytes, void *stream);

/*
 * IIR -- replaces the per-sample iir1 loop of Stream::ReadSamples
 * (CaribouliteStream.cpp:291-298): y = (int16)(float)LP(float(x)) on both
 * rails, Direct-Form-II biquad cascade in fp64.  A filter OBJECT stands where
 * the Stream's Iir::Butterworth::LowPass<6> members stand (CaribouliteStream.hpp:
 * 124-131): it owns the coefficients and the carried state of n_streams
 * independent streams -- 16 doubles per stream, [rail I|Q][8] with entry
 * 2*stage+0 = v1, 2*stage+1 = v2, up to 4 biquads -- which persists from call
 * to call for the life of the object (:84-91: the reference never resets it).
 * sos = n_stages rows of {b0,b1,b2,a1,a2} (host pointer, a0 = 1).
 */
typedef struct clhip_iir clhip_iir;
clhip_iir *clhip_iir_create(const double *h_sos, int n_stages, int n_streams);   /* current device; state = rest */
void   clhip_iir_destroy(clhip_iir *f);
/* n_samples of every stream (stream s at d_in/d_out + s*stride_samples int16 pairs; 4-byte aligned; d_out may be
 * d_in), asynchronous on `stream`; the state advances by the call.  One call in flight per object. */
int    clhip_iir_run(clhip_iir *f, const int16_t *d_in, int16_t *d_out, size_t stride_samples, size_t n_samples,
                     void *stream);
/* The same filter fed straight from raw SMI RX words (4 bytes per sample, all in sync): the 13-bit field extraction of
 * caribou_smi_rx_data_analyze (caribou_smi.c:338-378) is the filter's own input conversion -- no unpack launch, no int16
 * intermediate.  Out of place.  -2 = this call would take the scan path: unpack first and call clhip_iir_run. */
int    clhip_iir_run_smi(clhip_iir *f, int channel, const uint8_t *d_words, int16_t *d_out, size_t stride_samples,
                         size_t n_samples, void *stream);
/* Verdict on the LAST clhip_iir_run, to be asked after synchronising its stream and before its samples are used.
 * 0 = good.  -1 = a tile of the single-pass kernel gave up waiting for the tiles before it (its polls are bounded;
 * a launch that moves at all never gets there): the samples of that call are invalid, the carried state is back
 * where it was before the call, and the object takes the four-kernel scan (no waiting between workgroups) from now
 * on -- the call can simply be made again.  The word behind this is the object's own (pinned host memory): two
 * streams filtering on one GPU never see each other's verdicts. */
int    clhip_iir_status(clhip_iir *f);
/* synchronise + status + repair: 0 = good; 1 = the call had overrun and has been repeated on the scan path (d_out is
 * good now, the state has advanced once); -1 = overran with d_out == d_in (the input is gone: state restored, the
 * caller re-produces the input and calls again), or a runtime error */
int    clhip_iir_finish(clhip_iir *f);
/* take the LAST call back (its stream synchronised): the carried state is what it was before it; 0 / -1 (no call to take back) */
int    clhip_iir_unrun(clhip_iir *f);
/* carried state: 16 doubles per stream (NULL = rest); both synchronise with the last call */
int    clhip_iir_set_state(clhip_iir *f, const double *h_state);
int    clhip_iir_get_state(clhip_iir *f, double *h_state);
/* Samples after which nothing of an earlier state is left above 1e-12 (full-scale int16 input): a filter started from rest
 * that far before a point of a stream is in the whole stream's state from that point on -- time slices of one long stream
 * for several GPUs need a halo of this length, not a state hand-off.  0 = longer than the single-pass kernel's horizon. */
size_t clhip_iir_memory_samples(const clhip_iir *f);
void   clhip_iir_set_poll_bound(clhip_iir *f, int polls);   /* test hook: -1 forces every poll to give up */
/* test hook: force the single-pass kernel's segment length (16 / 32 / 64 samples per lane; 0 = by the call's size) and its tile order
 * (1 = chunks by atomic ticket, the default; 0 = a rank per wave) */
void   clhip_iir_set_shape(clhip_iir *f, int seg, int dynamic);
int    clhip_iir_on_scan_path(const clhip_iir *f);           /* 1 once an overrun (or CLHIP_IIR_ONEPASS=0) has switched it */
/* diagnostics (objects created under CLHIP_IIR_STAMPS=1): per-phase time stamps of the last single-pass launch,
 * [64 waves][16 tiles][12 phases] of the 100 MHz real-time counter; returns the word count (0: not enabled) */
size_t clhip_iir_debug_stamps(clhip_iir *f, unsigned long long *h_out);

/*
 * The RX pipe: raw SMI words -> int13 I/Q -> x/4096 -> FIR(T) -> [L/M polyphase
 * resampler | FM phase-difference demod | nothing] -> fp32, for n_streams
 * independent streams in one launch.  The stages after the unpack have no
 * reference implementation (SURVEY.md section 8 row a13 is their spec); they
 * sit where a client of readStream(CF32) would apply them.
 */
#define CL_PIPE_IN_SMI_WORDS 0   /* aligned raw RX words, 4 B/sample            */
#define CL_PIPE_IN_CS16      1   /* native int16 pairs (after IIR / re-sync)    */
#define CL_PIPE_IN_CF32      2   /* already-converted complex float            */
#define CL_PIPE_OUT_IQ       0   /* FIR (+ resampler) -> complex float          */
#define CL_PIPE_OUT_FM_DEMOD 1   /* FIR -> atan2(x[n] conj x[n-1]) -> float     */

typedef struct clhip_rx_pipe clhip_rx_pipe;   /* opaque, owns taps + per-stream state */

clhip_rx_pipe *clhip_rx_pipe_create(int n_streams, int channel,
                                    const float *h_fir_taps, int n_fir_taps,
                                    const float *h_rs_taps, int n_rs_taps, int up, int down,
                                    int out_mode);
void   clhip_rx_pipe_destroy(clhip_rx_pipe *p);
void   clhip_rx_pipe_reset(clhip_rx_pipe *p);                    /* zero history, phase 0      */
/* Time slicing of ONE long stream over several GPUs: place the pipe at absolute input index
 * n_total (sets the polyphase phase; history is left as it is).  A slice owner resets, seeks to (slice start - halo),
 * runs the halo samples before its slice once (outputs discarded) and then its slice: its outputs equal the
 * single-pipe outputs bit for bit.  halo = clhip_rx_pipe_halo(). */
void   clhip_rx_pipe_seek(clhip_rx_pipe *p, unsigned long long n_total);
size_t clhip_rx_pipe_halo(const clhip_rx_pipe *p);               /* pre-FIR samples of history kept per stream */
size_t clhip_rx_pipe_out_count(const clhip_rx_pipe *p, size_t n_in); /* outputs the next run yields */
int    clhip_rx_pipe_uses_fused(const clhip_rx_pipe *p, size_t n_in, int in_kind);
/* d_in: stream s at d_in + s*in_stride_elems (elements of in_kind);
 * d_out: stream s at d_out + s*out_stride_elems (complex float or float).
 * Returns outputs per stream (>= 0) or a negative error. */
long   clhip_rx_pipe_run(clhip_rx_pipe *p, int in_kind, const void *d_in, size_t in_stride_elems,
                         size_t n_in, void *d_out, size_t out_stride_elems, void *stream);
/* The streams of ONE pipe advancing independently -- what a stream group at the host boundary needs (cl_group_readStream
 * below: the reference's unit is one Soapy device per channel, soapy_api/SoapyCariboulite.cpp:46-69, each with its own
 * caribou_smi_read chunk loop, caribou_smi/caribou_smi.c:632-682, so one stream may deliver while its neighbour re-syncs or
 * returns -3).  An EPOCH is one such call: between _epoch_begin and _epoch_end every stream runs at most once, through
 * range runs over disjoint runs of NEIGHBOURING streams [first, first + count) -- one launch per run, same kernels, d_in /
 * d_out pointing at stream `first`'s row; the streams of one range must be on the same polyphase phase (inputs consumed
 * equal modulo 2 * down).  A stream no run touched keeps its state.  While an epoch is open, and once the streams have
 * drifted apart, the whole-pipe calls (clhip_rx_pipe_run, _run_smi, _rollback) refuse.  _out_count_stream = outputs the
 * next run of stream s yields for n_in inputs. */
int    clhip_rx_pipe_epoch_begin(clhip_rx_pipe *p);
long   clhip_rx_pipe_run_range(clhip_rx_pipe *p, int first, int count, int in_kind, const void *d_in, size_t in_stride_elems,
                               size_t n_in, void *d_out, size_t out_stride_elems, void *stream);
int    clhip_rx_pipe_epoch_end(clhip_rx_pipe *p, void *stream);
int    clhip_rx_pipe_unrun_stream(clhip_rx_pipe *p, int s, size_t n_in);   /* take back stream s's range run of the OPEN epoch (n_in inputs): it may run again, or not at all */
size_t clhip_rx_pipe_out_count_stream(const clhip_rx_pipe *p, int s, size_t n_in);
unsigned long long clhip_rx_pipe_stream_total(const clhip_rx_pipe *p, int s);   /* inputs stream s has consumed */
/* diagnostic: route config 2 through the s_memtime-stamped build (tools/phase_stamps.py); NULL = off */
void   clhip_rx_pipe_set_diag(clhip_rx_pipe *p, unsigned long long *d_buf);
/* force the multi-kernel generic path (second implementation, used by tests) */
void   clhip_rx_pipe_force_generic(clhip_rx_pipe *p, int on);
/* Device-side sync validation for CL_PIPE_IN_SMI_WORDS runs: d_offs holds the
 * per-chunk results of clhip_smi_find_offsets ([n_streams][ceil(n_in/chunk_samples)]).
 * d_offs == NULL with a flag: every tile verifies the sync words at the head of the chunks it reads itself
 * (offset 0 <=> the first four words carry the pattern, caribou_smi.c:235-292) -- no search launch at all.
 * A tile that needs a chunk with offs != 0 writes nothing and sets *d_bad_flag = 1;
 * the caller then synchronises, calls clhip_rx_pipe_rollback() and re-runs that call through
 * clhip_smi_unpack + CL_PIPE_IN_CS16 (clhip_rx_pipe_run_smi does all of this).  Pass NULL to disable. */
void   clhip_rx_pipe_set_sync_check(clhip_rx_pipe *p, const int32_t *d_offs, size_t chunk_samples,
                                    int32_t *d_bad_flag);
/* Undo the most recent clhip_rx_pipe_run (one level): history and polyphase phase are again what they were
 * before that call -- a run never overwrites the history it read.  The stream of that run must have been
 * synchronised.  0, or -1 when there is no run to undo (fresh / reset / seeked pipe, or already undone). */
int    clhip_rx_pipe_rollback(clhip_rx_pipe *p);
/*
 * One caribou_smi_read-shaped call (caribou_smi/caribou_smi.c:632-682) feeding the pipe, with the bytes of the
 * call resident on the device: stream s at d_bytes + s*stream_stride_bytes, n_bytes each, analysed in chunks
 * of chunk_len_bytes exactly like the reference's native batches.
 *   every chunk in sync  -> per-chunk sync search + ONE fused launch from the raw words, verdict checked on the device;
 *   a chunk with offs > 0 -> the raw-word run is rolled back and the call is redone with the reference's re-sync
 *                           semantics (:319-325,382-389: skipped bytes, one extrapolated sample, untouched slots)
 *                           through clhip_smi_unpack into d_cs16 ([n_streams][n_bytes/4 + 2] int16 pairs; it stands
 *                           where the Stream's interm_native_buffer stands, so "untouched" slots keep what it held);
 *   a chunk without sync -> CL_SMI_ERR_SYNC (:665-668) and the pipe keeps its pre-call state.
 * d_offs: [n_streams][ceil(n_bytes/chunk_len_bytes)] device scratch, holds the search results afterwards; h_offs
 * (optional, host) receives a copy.  Synchronises `stream`.  Returns outputs per stream, or a negative error.
 * d_cs16 == NULL: a call that needs the re-sync route returns CL_PIPE_ERR_RESYNC with the pipe rolled back and the
 * offsets in d_offs / h_offs, and the caller runs clhip_smi_unpack + clhip_rx_pipe_run(CL_PIPE_IN_CS16) itself. */
#define CL_PIPE_ERR_RESYNC (-4)
/* One-shot, single-stream pipes: the NEXT clhip_rx_pipe_run_smi also copies its outputs to h_out (host memory) before its
 * synchronisation, so a host caller pays one synchronisation per call (cl_readStream does this). */
void   clhip_rx_pipe_set_host_sink(clhip_rx_pipe *p, void *h_out);
/* A call whose chunks the fused kernel itself found in sync leaves zeros in h_offs and (default, on != 0) writes the same
 * zeros to d_offs with one more memset on the stream; a caller that only ever reads h_offs turns that off. */
void   clhip_rx_pipe_set_offs_writeback(clhip_rx_pipe *p, int on);
size_t clhip_rx_pipe_out_elem_bytes(const clhip_rx_pipe *p);       /* 8 (complex float) or 4 (FM demod) */
long   clhip_rx_pipe_run_smi(clhip_rx_pipe *p, const uint8_t *d_bytes, size_t stream_stride_bytes, size_t n_bytes,
                             size_t chunk_len_bytes, int32_t *d_offs, int32_t *h_offs, int16_t *d_cs16,
                             void *d_out, size_t out_stride_elems, void *stream);

/*
 * The TX pipe: fp32 message -> FM modulate (fp64 phase) -> L/M resample ->
 * x*4096 truncate -> int13 pack -> SMI TX bytes (config 5), or complex float
 * -> resample -> pack.  Returns samples packed per stream.
 */
#define CL_TXPIPE_IN_FM_MESSAGE 0
#define CL_TXPIPE_IN_CF32       1
typedef struct clhip_tx_pipe clhip_tx_pipe;
clhip_tx_pipe *clhip_tx_pipe_create(int n_streams, double fm_kf_hz, double fs_hz,
                                    const float *h_rs_taps, int n_rs_taps, int up, int down,
                                    int pack_mode);
void   clhip_tx_pipe_destroy(clhip_tx_pipe *p);
void   clhip_tx_pipe_reset(clhip_tx_pipe *p);
/* as _reset, but the next run is message n_total of a longer stream and the modulator's phase before it is h_phase_rad[stream]
 * (radians): the carried-state hand-off of a time slice; the resampler history is rebuilt by running
 * the messages just before the slice and discarding their outputs */
int    clhip_tx_pipe_seek(clhip_tx_pipe *p, unsigned long long n_total, const double *h_phase_rad);
size_t clhip_tx_pipe_out_count(const clhip_tx_pipe *p, size_t n_in);
long   clhip_tx_pipe_run(clhip_tx_pipe *p, int in_kind, const void *d_in, size_t in_stride_elems,
                         size_t n_in, uint8_t *d_bytes, size_t out_stride_bytes,
                         float *d_iq_tap, size_t iq_tap_stride, void *stream);
/* Verdict on the LAST clhip_tx_pipe_run, to be asked after synchronising its stream and before the bytes are
 * handed on (cl_writeStream does: nothing reaches the TX FIFO otherwise, caribou_smi.c:738-759).  0 = valid.
 * -1 = the single-launch FM path's bounded look-back gave up on a predecessor: the bytes are invalid, the pipe is
 * back in its pre-call state and from now on orders its superblocks by an atomic ticket instead of by dispatch
 * order -- every predecessor then belongs to a running workgroup and the wait is always finite -- so the caller
 * simply repeats the call (cl_writeStream does, once).  Dispatch order is the default because it is 7 % faster
 * (profiles/r02/c5_lookback_*_bench.json) and has never been observed to fail; correctness does not depend on it. */
int    clhip_tx_pipe_status(clhip_tx_pipe *p);
/* diagnostic knob: polls of a predecessor before the look-back gives up (tests force the failure with 0); < 0 = default */
void   clhip_tx_pipe_set_poll_bound(clhip_tx_pipe *p, int polls);
/* The carried state of ONE stream -- what a Stream object with a modulator keeps between two WriteSamples calls: the modulator's phase
 * and the resampler's history -- moves from one pipe to another of the same configuration: a stream group's multi-stream pipe and a
 * member's own (cl_group_writeStream: one Soapy device per channel, soapy_api/SoapyCariboulite.cpp:46-69, written together).  Both pipes
 * idle (last runs synchronised, verdicts asked); synchronises `stream`.  The polyphase position (messages taken so far; only its
 * remainder mod `down` matters) is the PIPE's, its streams advance together: _position reads it, _set_position places a pipe none of
 * whose streams carries state yet. */
unsigned long long clhip_tx_pipe_position(const clhip_tx_pipe *p);
int    clhip_tx_pipe_pack_mode(const clhip_tx_pipe *p);            /* the pack mode the pipe was created with */
int    clhip_tx_pipe_set_position(clhip_tx_pipe *p, unsigned long long n_total);
int    clhip_tx_pipe_move_stream(clhip_tx_pipe *dst, int dst_stream, clhip_tx_pipe *src, int src_stream, void *stream);

/* standalone FM / CW stages on device buffers */
int clhip_fm_demod(const float *d_iq, size_t n, float *d_prev_iq /*2 floats, in/out*/, float *d_out, void *stream);
int clhip_fm_mod(const float *d_msg, size_t n, double kf_hz, double fs_hz,
                 double *d_phase /*1 double, in/out*/, float *d_iq_out,
                 void *d_workspace, size_t workspace_bytes, void *stream);
size_t clhip_fm_mod_workspace_bytes(size_t n);
int clhip_cw_tone(double f_hz, double fs_hz, double phase0, size_t n, float *d_iq_out, void *stream);

/* ======================================================================== */
/* Layer 2: host C code with the reference's call surface                   */
/* (libcariboulite_host.so; links libcariboulite_hip.so)                    */
/* ======================================================================== */

/* --- SMI user-driver seam: caribou_smi/caribou_smi.h:85-106 ------------- */
typedef struct cl_smi cl_smi;   /* stands where caribou_smi_st stands */

cl_smi *cl_smi_init(int device);                       /* caribou_smi_init  caribou_smi.c:513-581 */
int     cl_smi_close(cl_smi *dev);                     /* caribou_smi_close :584-595              */
/* injection points replacing the /dev/smi fd: bytes
 * queued here are what read() on the fd would have returned, in order */
int     cl_smi_feed_bytes(cl_smi *dev, const uint8_t *h_bytes, size_t n_bytes);
/* zero-copy form: a pointer into the seam's pinned byte FIFO with room for n_bytes -- read(fd, p, n) straight into
 * it -- and the commit of what arrived; the host-to-device copies of the read-ahead reader start from that memory */
uint8_t *cl_smi_feed_reserve(cl_smi *dev, size_t n_bytes);
int     cl_smi_feed_commit(cl_smi *dev, size_t n_bytes);
size_t  cl_smi_pending_bytes(const cl_smi *dev);
void    cl_smi_set_max_read(cl_smi *dev, size_t max_bytes_per_read); /* model short reads */
/* replay front-end: queue bytes from a file / pipe / socket fd with the
 * reference's read pattern (<= one native batch per read(), short and ragged reads included) */
long    cl_smi_feed_fd(cl_smi *dev, int fd, size_t max_bytes);
long    cl_smi_feed_file(cl_smi *dev, const char *path, size_t offset_bytes, size_t max_bytes);
long    cl_smi_drain_to_fd(cl_smi *dev, int fd, size_t max_bytes);     /* TX mirror: write() per native batch */
/* bytes write() on the fd would have received */
size_t  cl_smi_drain_bytes(cl_smi *dev, uint8_t *h_bytes, size_t max_bytes);
void    cl_smi_set_tx_mode(cl_smi *dev, int cl_tx_mode);
/* caribou_smi_read  caribou_smi.c:632-682 (same arguments and return codes) */
int     cl_smi_read(cl_smi *dev, int channel, cl_sample_complex_int16 *buffer,
                    cl_sample_meta *metadata, size_t length_samples);
/* caribou_smi_write caribou_smi.c:720-762 */
int     cl_smi_write(cl_smi *dev, int channel, cl_sample_complex_int16 *buffer, size_t length_samples);
/* caribou_smi_set_debug_mode caribou_smi.c:612-615; in a debug mode cl_smi_read analyses ONE chunk,
 * updates the counters and returns CL_SMI_ERR_DEBUGMODE (-2), like caribou_smi.c:670-675 */
typedef struct {                /* caribou_smi_debug_data_st (caribou_smi.h:30-39) */
    uint32_t error_accum_counter;
    uint32_t cur_err_cnt;
    uint8_t  last_correct_byte;
    double   error_rate;
    double   bitrate;           /* Mbit/s, smi_calculate_performance's 0.98 : 0.02 blend (smi_utils.c:233-244)      */
    long     last_time_sec, last_time_usec;   /* the clock reading of the previous analysed chunk ({0, 0} at first, */
} cl_smi_debug_data;                          /* as the reference's zero-initialised struct timeval)                */
void    cl_smi_set_debug_mode(cl_smi *dev, int cl_smi_debug_mode);
const cl_smi_debug_data *cl_smi_get_debug_data(const cl_smi *dev);
/* The wall clock behind `bitrate`: gettimeofday() unless a caller supplies its own (tests replay recorded readings, a
 * file replay can pass the capture's own time base).  now(user, &sec, &usec); NULL restores gettimeofday. */
typedef void (*cl_smi_clock_fn)(void *user, long *sec, long *usec);
void    cl_smi_set_debug_clock(cl_smi *dev, cl_smi_clock_fn now, void *user);
/* counters of the seam: what the reference only logs (cariboulite_radio.c:1276-1283 "SMI reading operation failed" /
 * "synchronization failed", caribou_smi.c:657-668 "Reading timed-out" / the -3 exit) a caller can now read */
typedef struct {
    uint64_t samples_read;       /* samples delivered by read calls (read_so_far summed, caribou_smi.c:677)           */
    uint64_t resyncs;            /* chunks that came back with a sync offset > 0 (re-sync + extrapolated sample, :319-325,382-389) */
    uint64_t sync_losses;        /* chunks without sync: the call returned -3 (:665-668)                              */
    uint64_t timeouts;           /* read calls that found nothing pending ("Reading timed-out", :657-661)             */
    uint64_t io_errors;          /* calls that returned -1                                                            */
    uint64_t samples_written;    /* samples packed and queued by write calls (:757)                                   */
} cl_smi_stats;
void cl_smi_get_stats(const cl_smi *dev, cl_smi_stats *out);
/* caribou_smi_get_native_batch_samples caribou_smi.c:765-769 */
size_t  cl_smi_get_native_batch_samples(cl_smi *dev);
int     cl_smi_flush_fifo(cl_smi *dev);                /* caribou_smi_flush_fifo :772-783: drop the pending RX bytes */
/* Device-resident forms of the seam for callers whose next stage runs on the GPU (the C++ API, the stream
 * object): same chunk loop, slots, untouched-slot behaviour and return codes as cl_smi_read / cl_smi_write,
 * with `d_iq` (length + 1 slots) / `d_meta` (may be NULL) in DEVICE memory.  A read is complete on return; a
 * write consumes samples that are complete or were produced on cl_smi_stream(). */
int     cl_smi_read_to_device(cl_smi *dev, int channel, int16_t *d_iq, uint8_t *d_meta, size_t length_samples);
int     cl_smi_write_from_device(cl_smi *dev, int channel, const int16_t *d_iq, size_t length_samples);
void   *cl_smi_stream(cl_smi *dev);        /* the hipStream_t the seam launches on */
int     cl_smi_device(const cl_smi *dev);  /* its HIP device */

/* --- the ASYNC mode's sample ring: circular_buffer<T> (datatypes/circular_buffer.h:16-164) re-imagined ------
 * Power-of-two capacity; put() discards the oldest elements when override_write (else it is cut to what fits);
 * get() waits up to timeout_us and yields nothing unless all `length` elements are present (block_read), else
 * min(length, held) at once -- the observable behaviour of the reference's template.  The STORAGE is one array in
 * device memory (cl_ring_create_device) or host memory (cl_ring_create); the bookkeeping is on the host.  Producers
 * and consumers that own a GPU stream use the span calls: _begin names at most two linear pieces of the storage
 * (element positions), the caller moves the data itself (hipMemcpyAsync D2D, a kernel), makes sure the move is complete,
 * and calls _end, which publishes / releases.  The ring is NOT locked in between (one producer, one consumer): an open
 * put owns the elements behind the newest one, an open get the oldest ones, and a put that has to displace the oldest
 * elements of a full ring waits for an open get to end. */
typedef struct cl_ring cl_ring;
typedef struct { size_t pos[2], len[2]; } cl_ring_span;      /* elements; piece 1 is the wrapped part (pos 0) */
cl_ring *cl_ring_create(size_t size_elems, size_t elem_bytes, int override_write, int block_read);
cl_ring *cl_ring_create_device(int device, size_t size_elems, size_t elem_bytes, int override_write, int block_read);
void     cl_ring_destroy(cl_ring *r);
void    *cl_ring_storage(const cl_ring *r);                  /* base of the array (device or host pointer) */
int      cl_ring_on_device(const cl_ring *r);
size_t   cl_ring_put_begin(cl_ring *r, size_t length, cl_ring_span *span);              /* returns elements accepted */
void     cl_ring_put_end(cl_ring *r, size_t accepted);
void     cl_ring_put_cancel(cl_ring *r);                       /* instead of _end, span untouched: the put never happened */
void     cl_ring_put_abandon(cl_ring *r);                      /* instead of _end, span possibly written: nothing published, what was displaced is gone */
size_t   cl_ring_get_begin(cl_ring *r, size_t length, int timeout_us, cl_ring_span *span); /* 0: nothing claimed */
void     cl_ring_get_end(cl_ring *r, size_t claimed);
size_t   cl_ring_put(cl_ring *r, const void *data, size_t length);                      /* host data, either storage */
size_t   cl_ring_get(cl_ring *r, void *data, size_t length, int timeout_us);
void     cl_ring_reset(cl_ring *r);
size_t   cl_ring_size(cl_ring *r);
size_t   cl_ring_capacity(const cl_ring *r);

/* --- radio pass-through trio: cariboulite_radio.h:592-619 ---------------- */
typedef struct cl_radio cl_radio;   /* stands where cariboulite_radio_state_st stands */
cl_radio *cl_radio_create(cl_smi *smi, int channel);
void      cl_radio_destroy(cl_radio *radio);
int    cl_radio_read_samples(cl_radio *radio, cl_sample_complex_int16 *buffer,
                             cl_sample_meta *metadata, size_t length);    /* cariboulite_radio.c:1258-1285 */
int    cl_radio_write_samples(cl_radio *radio, cl_sample_complex_int16 *buffer,
                              size_t length);                             /* :1288-1307 */
size_t cl_radio_get_native_mtu_size_samples(cl_radio *radio);             /* :1310-1315 */
/* the read / write pair with DEVICE buffers (see cl_smi_read_to_device / cl_smi_write_from_device) */
int    cl_radio_read_samples_device(cl_radio *radio, int16_t *d_iq, uint8_t *d_meta, size_t length);
int    cl_radio_write_samples_device(cl_radio *radio, const int16_t *d_iq, size_t length);
cl_smi *cl_radio_smi(cl_radio *radio);

/* --- SoapySDR device/stream calls: soapy_api/Cariboulite.hpp:65-93 ------- */
typedef struct cl_device cl_device;   /* stands where class Cariboulite stands        */
typedef struct cl_stream cl_stream;   /* stands where class SoapySDR::Stream stands   */

/* makeCariboulite / Cariboulite::Cariboulite (Cariboulite.cpp:10-35):
 * kwargs "channel=S1G|HiF" selects the radio; anything else fails (NULL), as
 * the reference throws.  "gpu=N" picks the HIP device (extension). */
cl_device *cl_device_make(const char *const *keys, const char *const *vals, size_t n_kwargs);
void       cl_device_unmake(cl_device *dev);
cl_smi    *cl_device_smi(cl_device *dev);      /* to feed / drain SMI bytes */
const char *cl_device_last_error(cl_device *dev);

/* getStreamFormats CaribouliteStreamFunctions.cpp:11-19: returns count, fills up to max */
size_t cl_getStreamFormats(const cl_device *dev, int direction, size_t channel,
                           const char **formats, size_t max_formats);
/* getNativeStreamFormat :31-35 */
const char *cl_getNativeStreamFormat(const cl_device *dev, int direction, size_t channel, double *fullScale);
/* setupStream :100-139 -- NULL + cl_device_last_error() where the reference throws.
 * Extension kwargs (SURVEY.md section 5 "Config / flags"): FIR=<ntaps>:<cutoff_hz>,
 * RESAMP=<L>/<M>, DEMOD=FM, MOD=FM:<kf_hz>; ASYNC=1 enables the reader thread + ring of the reference's
 * compiled-out USE_ASYNC path (CaribouliteStream.cpp:11,16-49,70-75); ZEROCOPY=1 (RX) lets the client REGISTER buffers
 * of its own with the GPU (cl_stream_register_buffer below): a readStream whose buffs[0] lies inside a registered buffer has its
 * last kernel store into it directly; any other pointer takes the default route (the stream's pinned mirror + memcpy).  Nothing
 * is ever registered behind the client's back (round 3 registered "on first sight" and evicted: a user-pointer mapping of heap
 * pages the client may since have freed -- the same family of effects as the runtime's in-place pinning);
 * defaults = reference behaviour. */
cl_stream *cl_setupStream(cl_device *dev, int direction, const char *format,
                          const size_t *channels, size_t n_channels,
                          const char *const *keys, const char *const *vals, size_t n_kwargs);
/* ZEROCOPY=1 streams: register [p, p + bytes) (the pages around it) with the GPU until _unregister_buffers, the next setupStream
 * or the device's end -- the client keeps the buffer allocated that long.  At most 8 buffers per stream; a full table refuses
 * (no eviction: registrations never churn in steady state).  0, or -1 + cl_device_last_error.  _unregister_buffers waits for the
 * stream's device work first. */
int    cl_stream_register_buffer(cl_device *dev, cl_stream *stream, void *p, size_t bytes);
void   cl_stream_unregister_buffers(cl_device *dev, cl_stream *stream);
void   cl_closeStream(cl_device *dev, cl_stream *stream);                         /* :147-150 */
size_t cl_getStreamMTU(const cl_device *dev, cl_stream *stream);                  /* :162-165 */
int    cl_activateStream(cl_device *dev, cl_stream *stream, int flags,
                         long long timeNs, size_t numElems);                      /* :186-195 */
int    cl_deactivateStream(cl_device *dev, cl_stream *stream, int flags, long long timeNs); /* :212-216 */
int    cl_readStream(cl_device *dev, cl_stream *stream, void *const *buffs, size_t numElems,
                     int *flags, long long *timeNs, long timeoutUs);              /* :239-254 */
int    cl_writeStream(cl_device *dev, cl_stream *stream, const void *const *buffs, size_t numElems,
                      int *flags, long long timeNs, long timeoutUs);              /* :276-291 */
/* setBandwidth Cariboulite.cpp:395-417: RX bw < 160 kHz selects the IIR */
void   cl_setBandwidth(cl_device *dev, int direction, size_t channel, double bw);
int    cl_getDigitalFilter(const cl_device *dev);
size_t cl_stream_queue_size(const cl_device *dev, const cl_stream *stream);   /* ASYNC=1: samples queued by the reader thread and not read yet (0 without ASYNC) */
/* counters of the stream calls (SURVEY.md section 5 "metrics"): the reference squashes every error to 0 elements
 * (CaribouliteStream.cpp:185-194,266-276) and prints; here the caller can ask what happened */
typedef struct {
    uint64_t read_calls, elements_read;      /* readStream calls / elements they returned                              */
    uint64_t reads_empty;                    /* readStream calls that returned 0 (timeout, -1, -3: all squashed)       */
    uint64_t iir_overruns;                   /* reads whose IIR launch gave up waiting and was repeated on the scan path */
    uint64_t write_calls, elements_written;  /* writeStream calls / elements they consumed                             */
    uint64_t writes_empty;                   /* writeStream calls that returned 0                                      */
    uint64_t tx_overruns;                    /* writes whose modulator look-back gave up and was repeated in ticket order */
    uint64_t zero_copy_registrations;        /* ZEROCOPY=1: client buffers registered with the GPU so far (cl_stream_register_buffer) */
    uint64_t zero_copy_reads;                /* ZEROCOPY=1: reads whose last kernel stored into the client's buffer itself */
} cl_stream_stats;
void   cl_getStreamStats(const cl_device *dev, const cl_stream *stream, cl_stream_stats *out);
unsigned long cl_stream_iir_overruns(const cl_stream *stream);         /* = iir_overruns above */
/* test hook: hands clhip_iir_set_poll_bound to the stream's three filters */
void   cl_stream_set_iir_poll_bound(cl_stream *stream, int polls);

/* --- stream group: N devices of one GPU read in one call ------------------------------------------------------
 * The reference's unit is one SoapySDR device per channel (soapy_api/SoapyCariboulite.cpp:46-69: every board enumerates an
 * S1G and a HiF device), each readStream its own caribou_smi_read chunk loop (caribou_smi/caribou_smi.c:632-682,
 * soapy_api/CaribouliteStreamFunctions.cpp:239-254).  cl_group_readStream(g, buffs, numElems, rets, timeoutUs) IS
 *     for i in 0 .. n-1:  rets[i] = cl_readStream(devs[i], stream_i, &buffs[i], numElems, ...)
 * -- per-stream state, re-sync, untouched slots and the "-3" / timeout squashing exactly those of the N single calls, every
 * stream's result independent of its neighbours' -- executed as ONE pipeline: the pending native batches of the streams that
 * are in sync (the host sees the sync words in its pinned FIFO memory) go to the device SUBBATCH streams at a time, one
 * launch per sub-batch (the fused unpack + FIR + resample / demod kernel over several streams, or the unpack in the
 * stream format), while the sub-batch before it is on its way back and the one behind it on its way in (three HIP
 * streams), and the last hop into the clients' pageable buffers is shared by COPY_THREADS threads.  A stream that cannot take
 * that route (a slipped or lost chunk, a short read, ASYNC=1, a debug mode) takes its own device's single-stream
 * route inside the same call.  The reference's low-pass (cl_setBandwidth below 160 kHz) is batched too where a whole sub-batch of a
 * lane without extension stages has the same filter selected: one multi-stream filter launch fed from the raw words; the filters'
 * carried state moves between the members' own objects and the group's as the members change routes (never copied, never reset).
 * Make the group AFTER cl_setupStream of every member (RX); members are grouped by channel type and stream configuration
 * (format, FIR / RESAMP / DEMOD kwargs); a group with extension stages owns their state (one n-stream pipe per
 * configuration), so its members are read through the group from then on.  kwargs: SUBBATCH=<streams per launch> (4 where a stream delivers 1.5 MiB or more per call, else 8),
 * COPY_THREADS=<n> (2; 0 = the caller copies), SINK=copy (the sub-batch's outputs leave through
 * a device buffer and the copy engine instead of being stored into the mapped pinned mirror by the kernel itself),
 * INGEST_STREAMS=<1 .. 8> (HIP streams the sub-batches' copies in take turns on; 2), SLAB_MB=<MiB> (pinned FIFO room per member
 * in the group's ONE slab, 8: the members' byte FIFOs live there from cl_group_make to cl_group_unmake, a slice each, so that the
 * batches of members that are fed and read in step lie one stride apart and travel as one 2-D copy per sub-batch; a FIFO that
 * outgrows its slice moves into a buffer of its own and its batches come in by copies of their own; 0 = no slab), READAHEAD=<0|1|2>
 * (2, the default: before a call waits for its own results, the members' NEXT batches -- where they are pending already and in sync --
 * are staged in their FIFOs, copied in and launched over into a second pinned mirror, so that the next call finds its results
 * computed and the GPU does not wait for the host between two calls; 1: staged and copied in only; 0: nothing ahead.  Bytes read
 * ahead count as pending until that call takes them, and any other reader of the member's device -- its own readStream, a flush, a
 * call with another numElems -- finds them pending, in order; a run made ahead of a client who then goes another way is taken
 * back); a TX group: TX_COPY_MB=<MiB> (8: the copies in carry neighbouring sub-batches until they are that long -- 4 MiB copies cross PCIe
 * one at a time, two 8 MiB copies share the link; 0 = one copy per sub-batch).  Returns the number of streams that delivered (> 0 elements), or -1 on a
 * runtime error (cl_group_last_error; NULL group = the last cl_group_make failure). */
typedef struct cl_group cl_group;
typedef struct {
    uint64_t calls;              /* cl_group_readStream calls                                                      */
    uint64_t batched_reads;      /* member reads that took the batched route                                       */
    uint64_t single_reads;       /* member reads that took their device's single-stream route                      */
    uint64_t direct_reads;       /* batched reads stored straight into a registered client buffer (one launch per sub-batch) */
    uint64_t launches;           /* kernel launches of the batched route                                           */
    uint64_t errors;             /* calls that ended with a runtime error                                          */
    uint64_t copies_2d;          /* copies in that carried several members' batches at once (one slab stride apart) */
    uint64_t last_queue_us;      /* the last call: everything staged and queued after ... us                       */
    uint64_t last_arrive_us;     /*                the last sub-batch had arrived and was handed to the copy threads */
    uint64_t last_total_us;      /*                returned                                                         */
    uint64_t ahead_reads;        /* batched reads whose batch the call BEFORE had already staged and copied in (READAHEAD) */
} cl_group_stats;
cl_group   *cl_group_make(cl_device *const *devs, size_t n_devs, const char *const *keys, const char *const *vals, size_t n_kwargs);
void        cl_group_unmake(cl_group *g);                     /* BEFORE cl_device_unmake of any member: the group holds the members' seams (their FIFOs live in its slab) */
size_t      cl_group_size(const cl_group *g);
int         cl_group_readStream(cl_group *g, void *const *buffs, size_t numElems, int *rets, long timeoutUs);
/* A group of devices set up for TX (boards are half duplex, Cariboulite.hpp:60: a group reads or writes): N cl_writeStream calls as
 * one -- rets[i] is what cl_writeStream(devs[i], stream_i, &buffs[i], numElems, ...) returns, and on return every member's packed
 * words are in its TX FIFO behind what it held (Stream::WriteSamplesGen, CaribouliteStream.cpp:199-258, over caribou_smi_write,
 * caribou_smi.c:720-762).  Members without a modulator share launches of up to eight streams (conversion + caribou_smi_generate_data,
 * words stored straight into the room reserved in each pinned FIFO) while the next sub-batch's samples are copied in; members with
 * MOD / RESAMP kwargs of ONE configuration share launches too (a multi-stream TX pipe of the group's per sub-batch: I rails -> FM
 * modulator -> resampler -> quantiser -> pack; a member's modulator phase and resampler history move between its own pipe and the
 * group's as it changes routes, clhip_tx_pipe_move_stream; all of a sub-batch or none -- members whose polyphase positions differ mod
 * `down`, or one of whom has no buffer in the call, go through their own devices); a CS16 call above one MTU takes the member's own
 * device's writeStream inside the call.  The call returns with the
 * launches queued (write-behind by one call); the members' seams land what is in flight before anything looks at or adds to a TX
 * FIFO (cl_smi_drain_bytes / _drain_to_fd from any thread, cl_writeStream on a member), so the words are there for whoever asks.  A
 * runtime error of launches already reported as consumed is returned by the NEXT call.  No thread may be draining a member while the
 * group is unmade.  Returns the number of members that consumed elements, or -1 (cl_group_last_error).  cl_group_getStats:
 * batched_reads / single_reads count the writes. */
int         cl_group_writeStream(cl_group *g, const void *const *buffs, size_t numElems, int *rets, long timeoutUs);
int         cl_group_flush(cl_group *g);                      /* a TX group: wait for what cl_group_writeStream has in flight and commit it (0 / -1); a no-op for an RX group */
const char *cl_group_last_error(const cl_group *g);
void        cl_group_getStats(const cl_group *g, cl_group_stats *out);
void        cl_group_set_iir_poll_bound(cl_group *g, int polls);      /* test hook: clhip_iir_set_poll_bound for the group's own filter objects */
void        cl_group_set_tx_poll_bound(cl_group *g, int polls);       /* test hook: clhip_tx_pipe_set_poll_bound for the group's own modulator pipes (made or yet to be made) */
/* Explicit zero-copy: one client buffer per member (bytes_each long), registered with the GPU here and kept registered until
 * _unregister_buffers / cl_group_unmake -- the client keeps them allocated that long.  A call whose buffs[i] lies inside
 * member i's registered buffer is stored there across PCIe by a launch (no pinned mirror, no memcpy: clhip_rows_to_rows); any other pointer takes
 * the default route.  Nothing is ever registered behind the client's back. */
int         cl_group_register_buffers(cl_group *g, void *const *buffs, size_t bytes_each);
void        cl_group_unregister_buffers(cl_group *g);

/* ---- ONE call over the stream groups of SEVERAL GPUs (csrc/host/cl_node.c) ----
 * Independent channel streams shard over the GPUs of a node with no data-path collective (one SoapySDR device per channel,
 * soapy_api/SoapyCariboulite.cpp:46-69): the devices are made with their `gpu` kwarg, cl_node_make sorts them into
 * one cl_group per GPU and a call runs every group's call at once, each on a thread of its own (shard 0 on the caller's).  buffs[i] /
 * rets[i] are member i's, in the order the devices were given, exactly as in cl_group_readStream / cl_group_writeStream; the return value
 * is the groups' sum, or -1 if a group failed (cl_node_last_error; NULL node = the last cl_node_make failure).  kwarg SHARDS=<k> cuts every
 * GPU's members into k groups (contiguous blocks): the several-groups-at-once shape on a one-GPU box (tests); every other kwarg is
 * cl_group_make's.  cl_node_unmake BEFORE cl_device_unmake of any member. */
typedef struct cl_node cl_node;
cl_node    *cl_node_make(cl_device *const *devs, size_t n_devs, const char *const *keys, const char *const *vals, size_t n_kwargs);
void        cl_node_unmake(cl_node *nd);
size_t      cl_node_size(const cl_node *nd);
size_t      cl_node_shards(const cl_node *nd);                       /* groups the node runs (GPUs x SHARDS, none empty) */
cl_group   *cl_node_group(const cl_node *nd, size_t shard);          /* (its statistics, registered buffers, test hooks) */
int         cl_node_shard_of(const cl_node *nd, size_t member);      /* the shard member i went to (-1: none) */
int         cl_node_readStream(cl_node *nd, void *const *buffs, size_t numElems, int *rets, long timeoutUs);
int         cl_node_writeStream(cl_node *nd, const void *const *buffs, size_t numElems, int *rets, long timeoutUs);
int         cl_node_flush(cl_node *nd);                              /* cl_group_flush of every group (0 / -1) */
int         cl_node_register_buffers(cl_node *nd, void *const *buffs, size_t bytes_each);   /* cl_group_register_buffers of every group: buffs[i] is member i's (0 / -1) */
void        cl_node_unregister_buffers(cl_node *nd);
const char *cl_node_last_error(const cl_node *nd);

/* host helper: scipy.signal.firwin(ntaps, cutoff, window="hamming", fs=fs)
 * (the tap design SURVEY.md section 8 a13 specifies), rounded to fp32 */
int    cl_design_lowpass(int n_taps, double cutoff_hz, double fs_hz, double gain, float *taps_out);
/* host helper: iir1-style Butterworth low-pass as n/2 biquads {b0,b1,b2,a1,a2} */
int    cl_design_butter_lowpass(int order, double fs_hz, double fc_hz, double *sos_out);

#ifdef __cplusplus
}
#endif
#endif /* CARIBOULITE_HIP_H */
