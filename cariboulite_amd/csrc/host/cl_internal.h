/* cl_internal.h -- shared internals of the host C layer (libcariboulite_host.so).
 * Host code stays C (BASELINE.json north_star); every GPU operation goes
 * through the clhip_* C-ABI of libcariboulite_hip.so. */
#ifndef CL_INTERNAL_H
#define CL_INTERNAL_H

#include <pthread.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cariboulite_hip.h"

/* Growable byte FIFO standing where the /dev/smi kernel kfifo stands.  One linear buffer, three cursors:
 *     [keep, head)        bytes a reader has STAGED to the device (host-to-device copies read them in place) but whose
 *                         read() has not been confirmed yet -- they still belong to the FIFO and can be taken back;
 *     [head, head + len)  bytes pending;
 * The RX FIFO lives in pinned host memory: the feeder (or read(fd, ...) itself, cl_smi_feed_reserve / _commit) writes
 * where the DMA engine reads, no staging copy in between.  Ownership of the buffer's MOVES: only the producer side
 * (cl_fifo_reserve: compaction, growth) ever moves it, after waiting for the copies in flight (`dma_stream`); one
 * producer at a time, so a pointer handed out by a reservation stays good until its commit.  The consumer side never
 * moves it: bytes a reader gives back that do not fit in front of `head` go to the FRONT STASH, a second small buffer
 * only the consumer touches, which pop drains first and the in-place readers step around (they take the copying route
 * while it holds anything). */
#define CL_FIFO_DMA_STREAMS 3
typedef struct {
    uint8_t *data;
    size_t cap, keep, head, len;
    int pinned;                  /* hipHostMalloc'ed */
    int reserved;                /* a producer holds a pointer from cl_smi_feed_reserve it has not committed yet: nobody else may move `data` */
    int external;                /* `data` is a slice of memory somebody else owns (a stream group's pinned slab): never freed here; a FIFO that
                                  * outgrows it moves into a buffer of its own */
    void *dma_stream[CL_FIFO_DMA_STREAMS];   /* streams whose copies read FIFO memory in place (waited for before the buffer moves): the seam's
                                  * own, a stream group's copy of the current batch, the group's read-ahead */
    uint8_t *front;              /* front stash: [front_head, front_cap) are pending bytes OLDER than everything in `data` */
    size_t front_cap, front_head;
} cl_fifo;

void   cl_fifo_free(cl_fifo *f);
int    cl_fifo_adopt(cl_fifo *f, uint8_t *slice, size_t cap);   /* move the FIFO's bytes into `slice` (memory of the FIFO's own kind: pinned for a pinned FIFO) and live there from now on (0 / -1: does not fit) */
int    cl_fifo_leave(cl_fifo *f);                               /* the reverse: into a buffer of its own (0 / -1) */
uint8_t *cl_fifo_reserve(cl_fifo *f, size_t n);              /* room for n more bytes at the tail (may move the buffer) */
void   cl_fifo_commit(cl_fifo *f, size_t n);
int    cl_fifo_push(cl_fifo *f, const uint8_t *src, size_t n);
void  *cl_fifo_device_ptr(const cl_fifo *f, const uint8_t *at);   /* device address of a byte of a pinned FIFO's buffer, or NULL */
size_t cl_write_mapped_max(void);
uint8_t *cl_smi_tx_reserve(struct cl_smi *dev, size_t n);           /* TX FIFO: producer side, under fifo_mu */
void   cl_smi_tx_commit(struct cl_smi *dev, size_t n);
size_t cl_fifo_pop(cl_fifo *f, uint8_t *dst, size_t n);      /* consume with a copy (front stash first); dst may be NULL (discard) */
size_t cl_fifo_stage(cl_fifo *f, size_t n, uint8_t **where); /* take up to n bytes IN PLACE: they stay owned until confirmed */
void   cl_fifo_confirm(cl_fifo *f, size_t n);                /* the oldest n staged bytes are consumed for good */
void   cl_fifo_unstage(cl_fifo *f, size_t n);                /* the NEWEST n staged bytes are pending again */
int    cl_fifo_unpop(cl_fifo *f, const uint8_t *src, size_t n);   /* give bytes back at the FRONT of the pending ones; never moves `data` */
static inline size_t cl_fifo_front_len(const cl_fifo *f) { return f->front_cap - f->front_head; }
static inline size_t cl_fifo_pending(const cl_fifo *f) { return f->len + cl_fifo_front_len(f); }

#define CL_MAX_CHUNKS_INLINE 64

/* one read() of the reference's chunk loop (caribou_smi.c:643-679) */
typedef struct {
    size_t stage_off;   /* byte offset of the chunk inside the device staging buffer */
    size_t len;         /* bytes read() returned                                      */
    size_t slot0;       /* read_so_far when the chunk was analysed                     */
    int32_t offs;       /* sync offset found on the GPU (-1 = none)                    */
} cl_chunk;

#define CL_RA_SLOTS 3          /* device slots the read-ahead rotates: this read(), the one staged ahead, and the previous call's words */

struct cl_smi {
    int device;
    void *stream;                 /* hipStream_t of this SMI instance */
    size_t native_batch_len;      /* caribou_smi.c:74-81 */
    uint32_t sample_rate;
    cl_fifo rx, tx;
    pthread_mutex_t fifo_mu;      /* feeder thread vs reader thread (ASYNC mode) */
    pthread_cond_t fifo_fed;      /* signalled by every feed: poll(POLLIN) of the reference's timeout read */
    size_t max_read;
    int tx_mode;
    /* device / pinned buffers, grown on demand */
    uint8_t *d_bytes; size_t bytes_cap;
    /* the raw words of the previous call when they went straight into the caller's own kernel (which never materialises
     * int16 samples in d_iq), kept so that a re-sync in a LATER call finds in d_iq what the reference's persistent
     * intermediate buffer would hold in the slots it leaves untouched (cl_smi_restore_prev_words) */
    const uint8_t *prev_words; size_t prev_words_len; int prev_is_cs16;   /* (… or the int16 samples a filtered call delivered: cl_smi_set_prev_cs16) */
    int16_t *d_iq; size_t iq_cap;         /* samples */
    uint8_t *d_meta; size_t meta_cap;
    uint8_t *h_stage; size_t h_stage_cap; /* pinned host staging */
    uint8_t *h_txin; size_t h_txin_cap;   /* pinned staging of the samples a write call was handed (the runtime never sees the caller's pointer) */
    int32_t *d_offs; size_t offs_cap; int32_t *h_offs; size_t h_offs_cap;
    cl_chunk *chunks; size_t chunks_cap, n_chunks;
    int debug_mode;               /* caribou_smi_debug_mode_en */
    cl_smi_debug_data debug_data;
    cl_smi_clock_fn debug_clock; void *debug_clock_user;   /* NULL: gettimeofday */
    int32_t *d_dbg; int32_t *h_dbg; /* 4 ints each */
    /* read-ahead reader (cl_smi_read_device_ra): the NEXT read() is popped into the other pinned slot and its
     * host-to-device copy runs on `cstream` while the current chunk is analysed on `stream`.  The bytes are taken IN
     * PLACE from the pinned RX FIFO (cl_fifo_stage): what the feeder wrote is what the DMA engine reads */
    void *cstream; void *ev_copied[CL_RA_SLOTS];
    uint8_t *d_slot[CL_RA_SLOTS]; size_t slot_cap;   /* device side of the read-ahead; the host side is the pinned RX FIFO itself */
    struct { int valid, slot, head_ok; size_t len; } ahead;
    int next_slot;
    /* bytes a stream GROUP has staged ahead for its NEXT call (cl_group.c: their copy to the device is already queued): the newest staged
     * bytes of the FIFO.  Every entry of the seam's own readers gives them back first (cl_smi_foreign_cancel) and bumps the epoch, so that
     * the group can tell that its read-ahead is void */
    size_t foreign_ahead; unsigned foreign_epoch;
    /* a TX stream group holds words of this seam in flight (write-behind): called before anything looks at or adds to the TX FIFO */
    void (*tx_settle)(void *ctx, int member); void *tx_settle_ctx; int tx_settle_member;
    size_t ahead_bytes;          /* (atomic) what the consumer holds staged ahead, its own read-ahead + a group's: cl_smi_pending_bytes counts it, from any thread */
    int ra_pending; size_t ra_samples;     /* between cl_smi_ra_launch and cl_smi_ra_finish */
    size_t inplace_len;                    /* bytes of a one-read() call staged in place on `stream`: confirmed once that stream has been synchronised */
    /* cl_smi_ra_launch's short cut for a call that is ONE read() the host has seen to be in sync: set want_words before the
     * call; fast_used says whether it was taken; fast_words = the call's raw words on the device, ready on dev->stream */
    int want_words, fast_used;
    const uint8_t *fast_words;
    /* statistics (SURVEY.md section 5 "Metrics"): */
    uint64_t stat_samples, stat_resyncs, stat_sync_failures, stat_timeouts, stat_io_errors, stat_written;
    char err[256];
};

/* device-resident read: runs the chunk loop, leaves CS16 (+meta) in dev->d_iq /
 * dev->d_meta, fills dev->chunks.  Returns read_so_far, or CL_SMI_ERR_*;
 * *all_aligned = 1 when every chunk had offs == 0 and a whole number of samples. */
int cl_smi_read_device(cl_smi *dev, int channel, size_t length_samples, int want_meta, int *all_aligned);
/* the same chunk loop with results in caller-owned DEVICE buffers (NULL = the seam's own / no metadata) */
int cl_smi_read_device_to(cl_smi *dev, int channel, size_t length_samples, int16_t *d_iq, uint8_t *d_meta, int *all_aligned);
int cl_smi_ensure_iq(cl_smi *dev, size_t samples);   /* dev->d_iq holds at least `samples` int16 pairs: zeros at first, contents kept when it grows (0 / -1) */
int cl_smi_set_prev_words(cl_smi *dev, int channel, const uint8_t *w, size_t len);   /* raw words that stand in for dev->d_iq from now on (0 / -1) */
int cl_smi_set_prev_cs16(cl_smi *dev, int channel, const int16_t *p, size_t n_samples);   /* int16 samples (what a filtered call delivered) that stand in for dev->d_iq */
int cl_smi_restore_prev_words(cl_smi *dev, int channel);   /* bring dev->d_iq up to date from the previous call's raw words (0 / -1) */
/* the same chunk loop, one chunk at a time, with the next read() staged and copied ahead (reader threads) */
int cl_smi_read_device_ra(cl_smi *dev, int channel, size_t length_samples, int16_t *d_iq);
long cl_smi_ra_launch(cl_smi *dev, int channel, size_t length_samples, int16_t *d_iq);   /* its two halves: the caller may queue */
int cl_smi_ra_finish(cl_smi *dev);                                                      /* work on the seam's stream in between */
void cl_smi_readahead_cancel(cl_smi *dev);
void cl_smi_foreign_cancel(cl_smi *dev);
uint8_t *cl_smi_tx_reserve_raw(cl_smi *dev, size_t n);   /* cl_smi_tx_reserve without the call-back (the group's own reservation) */
void cl_smi_ahead_note(cl_smi *dev);     /* after every change of ahead / foreign_ahead */                 /* a group's read-ahead on this seam is given back (pending again) */
int cl_smi_head_in_sync(const uint8_t *chunk, size_t len);   /* offs == 0 decided on the host from the staged bytes */
/* poll(POLLIN, timeout) on the injected byte stream: returns 1 when bytes are pending (at once or within timeout_us) */
int cl_smi_wait_bytes(cl_smi *dev, long timeout_us);     /* bytes staged ahead go back to the front of the FIFO */
/* copy the slots the reference writes from the device results to host buffers */
int cl_smi_copy_out(cl_smi *dev, cl_sample_complex_int16 *buffer, cl_sample_meta *metadata, int upto_chunk);
int cl_ensure(void **p, size_t *cap, size_t need, size_t elem, int pinned);
void cl_seterr(char *dst, size_t n, const char *fmt, ...);

struct cl_radio {
    cl_smi *smi;
    int channel;
};

/* ---- the Soapy stream / device objects (cl_soapy.c; cl_group.c reads their configuration and drives their seams) ---- */
#define CL_ZC_SLOTS 8                  /* client buffers a ZEROCOPY stream keeps registered */
typedef struct {
    int enabled;
    int n_fir; float fir[128];
    int up, down, n_rs; float rs[40];
    int demod_fm;
    int mod_fm; double mod_kf;
} cl_dsp_cfg;

struct cl_stream {
    cl_device *dev;
    int format;                  /* CL_FORMAT_*                                 */
    int native_dir;              /* CL_SOAPY_SDR_RX / TX (setInnerStreamType)    */
    int stream_active;                     /* (atomic: read by the reader thread) */
    size_t mtu_size;
    int filter_type;             /* CL_DIGFILT_*                                 */
    double sos[3][15];           /* filt20 / filt50 / filt100: 3 biquads x {b0,b1,b2,a1,a2} */
    /* a stream group may hold the carried state of this stream's filters in multi-stream objects of its own: before the stream's own
     * objects are used (or its filters destroyed) the group hands the state back */
    void (*iir_home)(void *ctx, int member); void *iir_home_ctx; int iir_home_member;
    clhip_iir *iir[3];           /* filt20 / filt50 / filt100 (CaribouliteStream.hpp:124-131): state per filter, I and Q rails, never
                                  * reset, not even when the selection changes (CaribouliteStream.cpp:127-141) */
    int16_t *d_filt; size_t filt_cap;    /* the filtered samples: the IIR runs out of place, so a call can be repeated */
    cl_stream_stats stats;               /* (iir_overruns: calls the single-pass kernel gave up on; each was repeated on the scan path) */
    void *d_conv; size_t conv_cap;       /* converted output / TX input staging (bytes) */
    void *h_conv; size_t h_conv_cap;     /* pinned host mirror */
    int zero_copy;                       /* kwarg ZEROCOPY=1: client buffers are registered with the GPU and written by the last kernel */
    struct { uint8_t *base, *dev; size_t len; } zc[CL_ZC_SLOTS];
    int zc_n;
    cl_dsp_cfg dsp;
    clhip_rx_pipe *rx_pipe;
    clhip_tx_pipe *tx_pipe;
    /* a TX stream group may hold this stream's modulator phase and resampler history in a multi-stream pipe of its own: before the
     * stream's own pipe is used (or destroyed) the group lands what it has in flight and hands the state back */
    void (*tx_home)(void *ctx, int member); void *tx_home_ctx; int tx_home_member;
    /* ASYNC mode: reader thread + ring (CaribouliteStream.cpp:16-49,70-75).  The ring's storage is DEVICE memory:
     * the reader thread's unpacked samples go device-to-device into it while the next native batch is already
     * being copied host-to-device on a second HIP stream (cl_smi_read_device_ra); the consumer's stages read
     * them from there, so a sample crosses PCIe once as raw bytes and once as the client's output format. */
    int use_async;
    cl_ring *rx_queue;
    pthread_t reader_thread;
    int reader_thread_running;             /* (atomic: the client's thread and the reader thread) */
    int16_t *d_native1;                  /* interm_native_buffer1 of the reference, on the device (reader thread's) */
    void *astream;                       /* consumer-side HIP stream and linear CS16 device buffer: the reader */
    int16_t *d_aiq; size_t aiq_cap;      /* thread owns the cl_smi ones */
};

struct cl_device {
    cl_smi *smi;
    cl_radio *radio;
    int channel;
    cl_stream *stream;           /* ONE preallocated stream per device (Cariboulite.cpp:30) */
    char err[256];
};


/* Stream::ReadSamplesGen (CaribouliteStream.cpp:370-382) without the call counters: what cl_readStream runs */
/* a read in flight (cl_soapy.c: SOURCE -> STAGES -> SINK) */
typedef struct {
    int n;                        /* samples of the call (read_so_far); <= 0: nothing to deliver */
    const uint8_t *d_words;       /* raw words of a one-read() in-sync call, ready on hs -- or NULL */
    const int16_t *d_cs16;        /* native int16 samples, complete -- or NULL */
    void *hs;                     /* the HIP stream the stages are queued on */
    int pending;                  /* cl_smi_ra_finish is the epilogue's synchronisation (it carries the read's verdict) */
    int err;                      /* the chunk loop's error where the call delivers nothing (CL_SMI_ERR_*); 0 otherwise */
    int host_filled;              /* ASYNC, plain CS16: the ring's elements are already on their way to the sink's host side (no device stage follows) */
    size_t ring_claimed;          /* ASYNC: elements of the ring claimed by this call; the copy out of them is queued on hs, the claim ends
                                   * (cl_ring_get_end) once hs has been synchronised -- the epilogue's one synchronisation */
} cl_source;
typedef struct { int kind; void *d_dst; } cl_sink;
typedef struct {
    cl_source src; cl_sink sk;
    void *out; size_t ob;         /* the client's buffer; bytes per output element */
    long got; int bad;            /* outputs the stages yield; something could not be queued */
    int open;                     /* _begin queued work that _end has to wait for */
    int ret;                      /* the call's result where it was known at once (nothing pending, wrong direction, a plain CS16 chunk loop) */
} cl_read_ctx;
int cl_stream_read_begin(cl_device *dev, cl_stream *st, void *out, size_t numElems, long timeoutUs, cl_read_ctx *c);   /* 1: _end has work to wait for */
int cl_stream_read_end(cl_device *dev, cl_stream *st, cl_read_ctx *c);                                               /* the call's return value */
int cl_stream_read(cl_device *dev, cl_stream *st, void *const *buffs, size_t numElems, long timeoutUs);
/* Stream::Read + Stream::ReadSamples(int16*) (CaribouliteStream.cpp:260-301) with the result left on the DEVICE, complete:
 * *d_iq = the call's native samples (through the selected low-pass, overrun redo included).  Returns what Stream::Read returns
 * (errors squashed to 0). */
int cl_stream_read_native(cl_device *dev, cl_stream *st, size_t n, long timeout_us, const int16_t **d_iq);

#endif
