/* jaero_hip.h -- C ABI of libjaero_hip.so: the MI355X (gfx950) batched Aero demodulator.
 *
 * This is the drop-in boundary for the reference's demodulator hot path.  One `jaero_ctx` is a BANK of
 * `nchannels` independent demodulators of the same kind and rate living on one GPU; every entry point below
 * replaces one member of the reference's per-object QIODevice surface (citations are relative to
 * /root/reference/).  Plain pointers and sizes only -- no Qt, torch or HIP types in the signatures
 * (streams are passed as `void*` = hipStream_t, NULL = the default stream).
 *
 *   reference (one object per channel)                               this ABI (one call, all channels)
 *   ---------------------------------------------------------------  -----------------------------------------
 *   OqpskDemodulator(parent)+setSettings(Settings)+start()           jaero_create
 *     JAERO/oqpskdemodulator.cpp:8-117,175-289,312-315
 *   MskDemodulator(parent)+setSettings(Settings)+start()             jaero_create (kind = JAERO_KIND_MSK)
 *     JAERO/mskdemodulator.cpp:9-84,135-263,296-299
 *   setSettings on a live object                                     jaero_set_settings
 *   setAFC / setSQL / setCPUReduce                                   jaero_set_flags
 *     JAERO/oqpskdemodulator.cpp:149-163, JAERO/mskdemodulator.cpp:105-118
 *   DCDstatSlot(bool)            JAERO/oqpskdemodulator.cpp:679-684   jaero_set_dcd
 *   CenterFreqChangedSlot(double) JAERO/oqpskdemodulator.cpp:291-310  jaero_center_freq_changed
 *   (two objects per stereo device: JAERO/audioburstoqpskdemodulator.cpp:8-10: channels are independent)  jaero_comm_* / jaero_fan_out_pcm / jaero_gather_softbits
 *   writeData(const char*,qint64) JAERO/oqpskdemodulator.cpp:334-627, jaero_write
 *                                 JAERO/mskdemodulator.cpp:313-488
 *   signal processDemodulatedSoftBits(QVector<short>)                jaero_read_softbits / jaero_softbits_view
 *     JAERO/oqpskdemodulator.h:66, emitted at oqpskdemodulator.cpp:583-591, mskdemodulator.cpp:472-477
 *   signals Plottables / MSESignal / EbNoMeasurmentSignal / SignalStatus   jaero_read_status (+ status log)
 *     emitted together at JAERO/oqpskdemodulator.cpp:670-675, JAERO/mskdemodulator.cpp:510-517
 *   FreqOffsetEstimateSlot + CoarseFreqEstimate::ProcessBasebandData  internal (runs inside jaero_write at the
 *     JAERO/oqpskdemodulator.cpp:414-428,629-677; coarsefreqestimate.cpp:90-137   same sample the reference does)
 *   JConvolutionalCodec::Decode_Continuous / Decode_soft              jaero_viterbi_* (batched, stateless blocks)
 *     JAERO/jconvolutionalcodec.cpp:151-201,90-119
 *   BurstOqpskDemodulator / BurstMskDemodulator: ctor+setSettings+start   jaero_create (kind = JAERO_KIND_BURST_*)
 *     JAERO/burstoqpskdemodulator.cpp:4-131,202-277  JAERO/burstmskdemodulator.cpp:9-84,150-325
 *   their writeData / writeDataSlot                                    jaero_write (same call)
 *     JAERO/burstoqpskdemodulator.cpp:300-737  JAERO/burstmskdemodulator.cpp:371-754
 *   their processDemodulatedSoftBits (with the -1 start-of-burst marker)  jaero_read_softbits (marker kept as -1)
 *   their SignalStatus / EbNoMeasurmentSignal / Plottables emissions   jaero_read_events
 *   channel_stereo / channel_select_other (two objects fed the L and R     two channels of one bank fed with
 *     samples of one interleaved stream, audioburstoqpskdemodulator.cpp:8-10)  JAERO_PCM_FRAME_MAJOR input
 *   stop() / destructor                                               jaero_destroy
 *
 * Conventions: every function returns 0 on success or a negative JAERO_E* code (the reference has no error
 * channel; this is additive).  The caller owns every buffer it passes; the library owns all device state.  One host
 * thread per ctx.  jaero_write is asynchronous on the given stream; the read_* calls synchronise that stream.
 * There is NO CPU fallback: if no gfx950 device is usable jaero_create fails with JAERO_ENODEV.
 */
#ifndef JAERO_HIP_H
#define JAERO_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JAERO_ABI_VERSION 1

/* demodulator kinds */
#define JAERO_KIND_MSK 0         /* MskDemodulator          JAERO/mskdemodulator.h:17                      */
#define JAERO_KIND_OQPSK 1       /* OqpskDemodulator        JAERO/oqpskdemodulator.h:15                    */
#define JAERO_KIND_BURST_MSK 2   /* BurstMskDemodulator     JAERO/burstmskdemodulator.h:22   (600 / 1200)  */
#define JAERO_KIND_BURST_OQPSK 3 /* BurstOqpskDemodulator   JAERO/burstoqpskdemodulator.h:19 (10500)       */

/* error codes */
#define JAERO_OK 0
#define JAERO_EINVAL (-1)   /* bad argument / unsupported settings combination        */
#define JAERO_ENODEV (-2)   /* no usable HIP device                                    */
#define JAERO_ENOMEM (-3)   /* device or host allocation failed                        */
#define JAERO_EHIP (-4)     /* a HIP runtime call failed (see jaero_last_error); also: every call but jaero_destroy on a bank whose
                             * jaero_write failed AFTER its first state-advancing launch (device state and the host's schedule mirror
                             * disagree: nothing can continue from there).  A write that failed earlier -- input copy, transpose --
                             * left the bank as it was and may be repeated. */
#define JAERO_EOVERFLOW (-5)/* soft-bit / log capacity exceeded since the last read    */
#define JAERO_ENOTSUP (-6)  /* kind / rate not implemented                             */
#define JAERO_W_RATE 1      /* (jaero_ingest_push only) warning: sample rate differs, data queued anyway */

/* jaero_create flags */
#define JAERO_FLAG_EBNO 1u            /* run the EbNo meters (OQPSKEbNoMeasure/MSKEbNoMeasure, diagnostic only)   */
#define JAERO_FLAG_STATUS_LOG 2u      /* keep one status row per FreqOffsetEstimateSlot call (tests)              */
#define JAERO_FLAG_CAPTURE_SYMBOLS 4u /* keep the soft symbol (pt_qpsk / pt_msk) + mse of every symbol (tests)     */
#define JAERO_FLAG_TRACE 8u           /* burst kinds: also log peak-detector firings and every trident check (tests)   */

/* burst event kinds (jaero_read_events): what the burst classes emit as Qt signals while demodulating */
#define JAERO_EV_SIGNAL 0  /* SignalStatus(value != 0)        burstoqpskdemodulator.cpp:511,542  burstmskdemodulator.cpp:545,593 */
#define JAERO_EV_EBNO 1    /* EbNoMeasurmentSignal(value)     burstoqpskdemodulator.cpp:581      burstmskdemodulator.cpp:637     */
#define JAERO_EV_FREQ 2    /* Plottables(freq_est = value)    burstoqpskdemodulator.cpp:275,484  burstmskdemodulator.cpp:196,342,538 */
#define JAERO_EV_PEAK 3    /* (JAERO_FLAG_TRACE) pdet.update() returned true                                                      */
#define JAERO_EV_TRIDENT 4 /* (JAERO_FLAG_TRACE) trident check ran: value = +metric accepted / -metric rejected                   */

/* PCM layouts accepted by jaero_write */
#define JAERO_PCM_CHANNEL_MAJOR 0 /* pcm[ch * nsamples + i]  : one contiguous mono stream per channel            */
#define JAERO_PCM_FRAME_MAJOR 1   /* pcm[i * nchannels + ch] : interleaved frames, like multichannel PCM audio   */

/* Mirrors OqpskDemodulator::Settings / MskDemodulator::Settings
 * (JAERO/oqpskdemodulator.h:20-39, JAERO/mskdemodulator.h:24-45). */
typedef struct jaero_settings
{
    int kind;                    /* JAERO_KIND_*                                   */
    int coarsefreqest_fft_power; /* 2^power point coarse-frequency FFT (13 or 14)  */
    double freq_center;          /* Hz                                             */
    double lockingbw;            /* Hz                                             */
    double fb;                   /* bit rate: 10500 or 8400 (OQPSK; 8400: continuous kind, fft power 14), 600 / 1200 (MSK) */
    double Fs;                   /* sample rate: 48000; MSK kind also 24000, 12000 */
    double signalthreshold;      /* mse threshold                                  */
} jaero_settings;

/* What the reference emits from FreqOffsetEstimateSlot, per channel. */
typedef struct jaero_status
{
    double mse;         /* MSESignal                               */
    double ebno;        /* EbNoMeasurmentSignal (0 unless JAERO_FLAG_EBNO) */
    double freq_est;    /* Plottables arg 1 = mixer2.GetFreqHz()   */
    double freq_center; /* Plottables arg 2 = mixer_center.GetFreqHz() */
    int signal;         /* SignalStatus                            */
    int n_estimates;    /* number of FreqOffsetEstimateSlot calls so far */
} jaero_status;

typedef struct jaero_ctx jaero_ctx;

/* Create a bank of nchannels demodulators on HIP device `device`.  All channels must agree on kind, fb, Fs and
 * coarsefreqest_fft_power (they share kernels and ring geometry); freq_center, lockingbw and signalthreshold are
 * per channel.  If `per_channel_stride` is 0, settings[0] is used for every channel.
 * max_write_samples bounds nsamples of one jaero_write (sizes staging buffers); softbit_capacity is the number of
 * soft bits each channel can hold between reads (0 = default: enough for 2*max_write_samples). */
int jaero_create(int device, int nchannels, const jaero_settings *settings, int per_channel_stride,
                 unsigned flags, int max_write_samples, int softbit_capacity, jaero_ctx **out);
void jaero_destroy(jaero_ctx *ctx);

/* setSettings on a live channel (channel = -1: every channel): the channel's state becomes what the reference's setSettings leaves behind
 * (JAERO/oqpskdemodulator.cpp:175-289, JAERO/mskdemodulator.cpp:135-263).
 *   - freq_center / lockingbw / signalthreshold of one or all channels: in place, enqueued on the stream of the bank's last jaero_write.
 *   - one channel of an 8400 bps bank (same fb / Fs / FFT power): in place as well; that channel's prefilter restarts as JFastFir::SetKernel
 *     leaves it (empty history, 2048 exact zeros in front of its first output, JAERO/oqpskdemodulator.cpp:278-283) while the transform blocks
 *     stay on the bank's grid of absolute multiples of 2048 samples -- the same filtered values up to transform round-off.
 *   - fb, Fs or the FFT power (shared by the channels of a bank), and a setSettings of a WHOLE 8400 bps bank (every prefilter restarts, the
 *     block grid with them): whole bank only (channel = -1, or a one-channel bank).  The bank is re-created behind the handle and receives what the reference keeps in the old
 *     object: oscillator phases, loop-filter / rotator / timing states, the symbol-rate windows (MSK: msema, the first entries of dt and
 *     delayedsmpl in buffer order), the EbNo meter of the OQPSK kind, the coarse ring and the smoothed spectrum, flags, unread outputs.
 *     Control plane: allocates and synchronises the device; pointers from the *_view calls are stale afterwards.  An OQPSK bank keeps Fs.
 *     The new bank exists beside the old one until the state has moved: a bank that fills more than half of the device memory cannot change
 *     rate this way (JAERO_ENOMEM, the old bank stays as it was).
 *   - burst banks (JAERO/burstoqpskdemodulator.cpp:202-277, JAERO/burstmskdemodulator.cpp:150-325), one or all channels, same fb and Fs: in
 *     place and on that stream too.  AGCs, EbNo meter, burst-timing averages and the Hilbert filter restart from empty, the peak detector is
 *     locked for twice its length and the trident buffer refills (and is checked once, whatever is in the air), mixer2 returns to
 *     freq_center; d1 / d2 / the peak detector's lines (burst MSK: delayedsmpl) keep their CONTENTS with the pointer back at zero, as
 *     DelayThing::setLength leaves them; startstop, the oscillator phases and RxDataBits survive (burst MSK: cntr = 0, mse = 10, dcd = false,
 *     new matched filters).  A Plottables row is appended to the channel's event log.
 *   - JAERO_EINVAL: another kind (another class in the reference), or fb / Fs / FFT power for one channel of several;
 *     Burst MSK with its other bit rate (600 <-> 1200 bps; whole bank): as for the continuous kinds a sibling bank takes the old one's place and
 *     receives what the reference keeps -- the DelayThings' first min(old, new) entries in storage order, startstop, oscillator phases, msema,
 *     unread outputs.
 *     JAERO_ENOTSUP: another Fs for a burst bank, another fb for burst OQPSK (create a new bank; the Qt
 *     adaptors of integration/qt do). */
int jaero_set_settings(jaero_ctx *ctx, int channel, const jaero_settings *s);
int jaero_set_flags(jaero_ctx *ctx, int channel, int afc, int sql, int cpu_reduce);
int jaero_set_dcd(jaero_ctx *ctx, int channel, int dcd);
/* CenterFreqChangedSlot: continuous kinds JAERO/oqpskdemodulator.cpp:291-310 / mskdemodulator.cpp:265-282; burst MSK
 * JAERO/burstmskdemodulator.cpp:327-342 (clamp, mixer2 follows under AFC or is pulled to within lockingbw/2, Plottables emission in the
 * event log); burst OQPSK: accepted, no effect -- the reference's slot is empty (burstoqpskdemodulator.cpp:284-289). */
int jaero_center_freq_changed(jaero_ctx *ctx, int channel, double freq_center_hz);

/* = writeData for every channel: nsamples of real int16 PCM per channel.  `pcm` is a host pointer
 * (is_device_ptr = 0; copied with hipMemcpyAsync) or a device pointer on the ctx's device (is_device_ptr = 1). */
int jaero_write(jaero_ctx *ctx, const int16_t *pcm, int nsamples, int layout, int is_device_ptr, void *stream);

/* Drain the soft bits channel `channel` produced (values 0..255 as the reference's QVector<short>), oldest first.
 * *n receives the count copied (<= cap).  Grouping into 32 (OQPSK) / 12 (MSK) emissions is the caller's. */
int jaero_read_softbits(jaero_ctx *ctx, int channel, int16_t *dst, int cap, int *n);
/* Batched drain: dst[ch * cap_per_channel + k], counts[ch].  Resets every channel's buffer. */
int jaero_read_softbits_all(jaero_ctx *ctx, int16_t *dst, int cap_per_channel, int *counts);
/* Device-side view for zero-copy consumers (RCCL gather, a downstream device decoder): int16 [nch][capacity]
 * and int32 counts[nch] on the ctx's device.  jaero_discard_softbits resets the counts on `stream`. */
int jaero_softbits_view(jaero_ctx *ctx, void **dev_softbits, void **dev_counts, int *capacity);
int jaero_discard_softbits(jaero_ctx *ctx, void *stream);

int jaero_read_status(jaero_ctx *ctx, int channel, jaero_status *st);
/* status log rows of 6 doubles [n, freq_est, freq_center, mse, ebno, signal] (JAERO_FLAG_STATUS_LOG) */
int jaero_read_status_log(jaero_ctx *ctx, int channel, double *rows, int caprows, int *nrows);
/* soft symbols rows of 3 doubles [re, im, mse] (JAERO_FLAG_CAPTURE_SYMBOLS) */
int jaero_read_symbols(jaero_ctx *ctx, int channel, double *rows, int caprows, int *nrows);

/* Burst kinds: rows of 3 doubles [absolute sample index (count of samples written before it), JAERO_EV_* kind, value],
 * oldest first, drained by the call.  The first row of every channel is the Plottables emission of setSettings. */
int jaero_read_events(jaero_ctx *ctx, int channel, double *rows, int caprows, int *nrows);

/* Batched K=7 r=1/2 {109,79} soft Viterbi (libcorrect semantics as used by JConvolutionalCodec).
 * jaero_viterbi_decode_soft: nblocks independent blocks of nsoft soft bytes each (0..255, 128 = erasure);
 *   = correct_convolutional_decode_soft per block; bits_out[b * (nsoft/2) + k] one byte per decoded bit
 *   (the first nsoft/2 - 6 are decoded data, the rest 0).
 * jaero_viterbi_continuous: = JConvolutionalCodec::Decode_Continuous for nstreams independent streams, one
 *   block of nsoft soft bytes each per call; the 62-byte overlap of each stream is kept in the ctx-less state
 *   buffer `overlap_state` (nstreams*64 bytes: bytes 0..61 = the kept soft bytes, byte 62 = how many are valid, 0 or
 *   62; zero it for a fresh stream).  nbits_out[s] receives the number of bits returned for stream s (nsoft/2, fewer
 *   on the first block of a stream, exactly as the reference).
 * Both take host pointers (is_device_ptr=0) or device pointers (1). */
int jaero_viterbi_decode_soft(int device, const uint8_t *soft, int nblocks, int nsoft, uint8_t *bits_out,
                              int is_device_ptr, void *stream);
int jaero_viterbi_continuous(int device, const uint8_t *soft, int nstreams, int nsoft, int paddinglength,
                             uint8_t *overlap_state, uint8_t *bits_out, int *nbits_out, int is_device_ptr, void *stream);

/* ---- Aero-L bit pipeline around the Viterbi (continuous P-channel path of AeroL::Decode, JAERO/aerol.cpp:1124-2039) ----
 * A jaero_aerol_ctx is a bank of nchannels AeroL objects in non-burst mode at one bit rate (600 / 1200 / 10500):
 *   AeroL(parent) + setSettings(fb,false)   JAERO/aerol.cpp:904-977,990-1072      jaero_aerol_create
 *   processDemodulatedSoftBits(QVector<short>)  JAERO/aerol.cpp:2077-2099         jaero_aerol_write (all channels at once; the soft
 *                                               bits can stay on the device: pass jaero_softbits_view's pointers, is_device_ptr = 1)
 *   the signal units Decode() checks and prints  JAERO/aerol.cpp:1583-1600          jaero_aerol_read_sus: rows of 16 int32
 *                                               [frame number, unit index k, 12 bytes (10 payload + CRC), crc_ok, frame-info word]
 *   DataCarrierDetect(bool) and the "Error short frame" notice  :1593-1596,1995-2010   jaero_aerol_read_events: rows of 3 int64
 *                                               [soft-bit index, kind (0 = DCD, 1 = short frame (value = its length), 2 = unique word), value]
 *   updateDCD() from the 1 s QTimer            JAERO/aerol.cpp:1109-1122          jaero_aerol_tick_dcd (call once per second of signal)
 * What follows a CRC-clean signal unit in the reference (message names, ISU / ACARS reassembly, plane database) is text and control
 * plane and stays with the caller. */
typedef struct jaero_aerol_ctx jaero_aerol_ctx;
int jaero_aerol_create(int device, int nchannels, int fb, int max_softbits_per_write, int su_capacity, jaero_aerol_ctx **out);
void jaero_aerol_destroy(jaero_aerol_ctx *ctx);
/* soft[ch * stride + k], k < counts[ch] <= max_count <= stride; host pointers (copied) or device pointers.  Burst-mode banks read rows that are
 * 16-byte aligned (soft 16-byte aligned, stride a multiple of 8) eight entries per load and skip through inert stretches; other rows go bit by bit:
 * the same results, more slowly. */
int jaero_aerol_write(jaero_aerol_ctx *ctx, const int16_t *soft, const int *counts, int stride, int max_count, int is_device_ptr, void *stream);
int jaero_aerol_read_sus(jaero_aerol_ctx *ctx, int channel, int32_t *rows, int caprows, int *nrows);
int jaero_aerol_read_events(jaero_aerol_ctx *ctx, int channel, long long *rows, int caprows, int *nrows);
int jaero_aerol_tick_dcd(jaero_aerol_ctx *ctx, int *dcd_out /* optional [nchannels] */);
/* ---- burst mode: R / T channel packets (AeroL with setSettings(fb, burstmode = true), JAERO/aerol.cpp:996-1003,1062-1070) ----
 * The bank behind a burst demodulator bank (JAERO_KIND_BURST_OQPSK): unique word with tolerance 4 that has to come ~80 soft bits after
 * the demodulator's start-of-burst marker (JAERO/aerol.cpp:1192-1200), RTChannelDeleaveFECScram::update (JAERO/aerol.h:785-873: trial
 * decodes of the collected block at 2, 5, 8 .. 95 interleaver columns until the CRCs of an R packet or of a T packet's header and
 * signal units pass), end of signal after one second of bits.  At 600 / 1200 bps (behind JAERO_KIND_BURST_MSK): one detector, the word
 * within 250 soft bits of the marker, RTChannelDeleaveFECScram::updateMSK (aerol.h:631-782: R test at 5 blocks, the unit count read at 11,
 * the T packet decoded at the announced length), three seconds of bits.
 *   jaero_aerol_read_packets: rows of 16 int32 [packet number, chunk, 12 bytes (zero padded), total bytes of the packet, type]
 *       type 1 = R packet (20 bytes: 17 + CRC + the flush byte), 2 = T packet (6 header bytes incl. CRC, then 12 per signal unit)
 *   jaero_aerol_read_events additionally reports kind 3 = the " Bad R/T Packet" notice (JAERO/aerol.cpp:1289-1293,1531)
 * The reference drops the rest of the demodulator's current group of soft bits at the end of a signal; the bank re-derives the groups
 * from the stream (a marker is one entry, soft bits come in pairs, a group is complete at >= 32 entries after a pair).  The input is a burst
 * demodulator's output: a marker stands between pairs, never inside one (JAERO/burstoqpskdemodulator.cpp:546-585); for other streams the grouping,
 * and with it what is dropped at the end of a signal, is not the reference's. */
/* C channel (jaero_aerol_create with fb = 8400: AeroL::DecodeC aerol.cpp:2187-2502): jaero_aerol_read_sus rows are the three sub-band signal units of
 * a frame [frame, k, 12 bytes, crc_ok, 0]; jaero_aerol_read_voice rows are 304 bytes: uint32 frame number, then the 300 voice bytes
 * the reference hands to Voicesignal(data, hex). */
int jaero_aerol_read_voice(jaero_aerol_ctx *ctx, int channel, uint8_t *rows, int caprows, int *nrows);
int jaero_aerol_create_burst(int device, int nchannels, int fb, int max_softbits_per_write, int packet_row_capacity, jaero_aerol_ctx **out);
int jaero_aerol_read_packets(jaero_aerol_ctx *ctx, int channel, int32_t *rows, int caprows, int *nrows);
/* HIP-event time per kernel class since the last reset: which 0 = k_aerol_bits, 1 = Viterbi, 2 = k_aerol_post */
int jaero_aerol_profile_enable(jaero_aerol_ctx *ctx, int on);
int jaero_aerol_profile_read(jaero_aerol_ctx *ctx, int which, double *total_ms, int *launches, int reset);

/* ---- batched ingest: the recAudio(QByteArray, quint32 sampleRate) -> dataReceived slot of every channel
 * (JAERO/zmq_audioreceiver.cpp:40-79 -> oqpskdemodulator.cpp:686-693, mskdemodulator.cpp:528-537) in front of one bank.
 * Messages arrive per channel, any size (<= 192000 bytes are taken, as the reference's receive buffer), any order;
 * jaero_ingest_pump turns what all channels have in common into jaero_write calls of chunk_samples from pinned memory.
 *   jaero_ingest_push   = dataReceived of one channel.  Returns 0; JAERO_W_RATE (> 0) when sample_rate != Fs for the OQPSK
 *                         kinds (the reference only logs "Sample rate not supported by demodulator" and demodulates anyway);
 *                         JAERO_ENOTSUP for the MSK kinds (the reference would re-apply its settings at the new rate; a bank
 *                         shares Fs); JAERO_EOVERFLOW when the channel's FIFO cannot take the message (nothing queued).
 *   jaero_ingest_queued = samples queued for `channel`, or (channel = -1) the count every channel has in common
 *   jaero_ingest_pump   = flush != 0 also writes the common remainder below one chunk; *chunks = jaero_write calls made
 *   jaero_ingest_stats  = [rate warnings, samples refused, samples per channel written]
 * The transport (sockets) is the caller's; nothing here blocks except on the staging buffer two chunks back. */
typedef struct jaero_ingest jaero_ingest;
int jaero_ingest_create(jaero_ctx *bank, int chunk_samples, int capacity_samples, jaero_ingest **out);
void jaero_ingest_destroy(jaero_ingest *ing);
int jaero_ingest_push(jaero_ingest *ing, int channel, const void *pcm_bytes, int nbytes, unsigned sample_rate);
int jaero_ingest_queued(const jaero_ingest *ing, int channel);
int jaero_ingest_pump(jaero_ingest *ing, int flush, void *stream, int *chunks);
int jaero_ingest_stats(const jaero_ingest *ing, long long *three);

/* Host-only debugging aid (no device needed): the sample indices at which jaero_write would run the coarse-frequency
 * estimate for a fresh channel fed `nwrites` writes of write_sizes[i] samples.  Returns the number of triggers
 * (>= 0; up to `cap` are stored) or a negative error. */
int jaero_debug_schedule(int fft_power, int Fs, int cpu_reduce, const int *write_sizes, int nwrites,
                         long long *trigger_samples, int cap, int *segments_out);

/* Host-only: the same for `nch` channels of one bank that hold their own flags (flags0[ch]: 1 AFC, 2 SQL, 4 cpuReduce, 8 DCD) and change them
 * between writes: events[k] = {before_write, channel (-1: all), kind (0 jaero_set_flags bits, 1 jaero_set_dcd, 2 jaero_set_settings), value}.
 * Stores (sample, channel) pairs, one per firing of a channel's estimate (JAERO/oqpskdemodulator.cpp:410-431 with per-object cpuReduce). */
int jaero_debug_schedule_lanes(int fft_power, int Fs, int nch, const int *flags0, const int *write_sizes, int nwrites,
                               const int *events, int nevents, long long *trig_sample_channel, int cap, int *segments_out);

/* Test hook: the 8400 bps prefilter kernel alone.  n complex samples (re, im interleaved, host pointers) through the kernel
 * RRC(alpha, 2049 taps, 48 kHz, fsym symbols/s) with JFastFir's latency for nfft = 4096 (out[m] = sum_k h[k] x[m - 2048 - k]):
 * JFastFir::SetKernel + update as JAERO/oqpskdemodulator.cpp:278-283,366-368 use it and JAERO/tests/jfastfir_tests.cpp:31-58 pins it. */
int jaero_debug_prefilter(int device, const double *in_reim, int n, double alpha, double fsym, double *out_reim);
/* Test hook: the first n prefiltered complex samples (re, im pairs) of the last jaero_write of an 8400 bps bank, channel ch
 * (cval_prefiltered, JAERO/oqpskdemodulator.cpp:343-381). */
int jaero_debug_read_prefiltered(jaero_ctx *ctx, int channel, double *out_reim, int n);
/* Test hook: the Viterbi decoder picks its layout by size (one block per wavefront below 16 384 blocks, one per lane from there); tests
 * force one so that both meet the oracle at small sizes.  mode: 0 = by size (default), 1 = wave, 2 = lanes.  Process-wide. */
int jaero_debug_viterbi_layout(int mode);

/* ------------------------------------------------------------------------------------------------ multi-GPU edge operations
 * The path shards by channel with no steady-state exchange (the reference runs its two stereo burst channels as two unrelated objects,
 * JAERO/audioburstoqpskdemodulator.cpp:8-10); the north star names two operations at the edges: fan out shared PCM, gather decoded bits.
 * One jaero_comm per GPU (one process or thread each): RCCL point-to-point sends over xGMI, grouped per call; contiguous channel ranges
 * [rank * N / W, (rank + 1) * N / W) (jaero_shard_range), the same as jaero_amd/dist.py.  RCCL is loaded on first use (dlopen): hosts with
 * one GPU never need it.  world = 1 with id = NULL is a communicator without RCCL (both operations are local copies).
 * ANY RCCL error invalidates the jaero_comm: a group that failed half-queued is closed, the communicator is aborted (ncclCommAbort) and every
 * later call on it returns JAERO_EHIP -- destroy it and create a new one on every rank. */
typedef struct jaero_comm jaero_comm;
#define JAERO_COMM_ID_BYTES 128
int jaero_shard_range(int nch_total, int rank, int world, int *lo, int *hi);
int jaero_comm_get_unique_id(void *id_128_bytes);            /* on one rank; hand the 128 bytes to the others (= ncclGetUniqueId) */
int jaero_comm_create(int device, int rank, int world, const void *id_128_bytes, jaero_comm **out);
void jaero_comm_destroy(jaero_comm *comm);
/* rank `src` holds frame-major PCM [nsamples][nch_total] on its device (the layout jaero_write takes with JAERO_PCM_FRAME_MAJOR); every
 * rank receives its channel slice, contiguous, [nsamples][hi - lo], in d_mine.  Enqueued on `stream`. */
int jaero_fan_out_pcm(jaero_comm *comm, int src, const int16_t *d_frames_all, int nsamples, int nch_total, int16_t *d_mine, void *stream);
/* every rank's soft-bit rows [hi - lo][cap] and counts [hi - lo] (the buffers behind jaero_softbits_view) arrive on rank `dst` as
 * [nch_total][cap] / [nch_total].  Enqueued on `stream`. */
int jaero_gather_softbits(jaero_comm *comm, int dst, const int16_t *d_soft, const int *d_counts, int nch_total, int cap,
                          int16_t *d_soft_all, int *d_counts_all, void *stream);

/* introspection */
int jaero_abi_version(void);
int jaero_num_channels(const jaero_ctx *ctx);
const char *jaero_strerror(int code);
const char *jaero_last_error(void);
/* Time (ms) the GPU spent in each kernel class over the jaero_write calls since the last reset, measured with HIP
 * events on the launch stream; enabled by jaero_profile_enable(ctx,1).  which: 0 = sample-loop kernel,
 * 1 = coarse-frequency kernel (burst kinds: trident check), 2 = PCM transpose / history push, 3 = Hilbert FIR (burst),
 * 4 = burst front end (burst).  *launches receives the launch count. */
int jaero_profile_enable(jaero_ctx *ctx, int on);
int jaero_profile_read(jaero_ctx *ctx, int which, double *total_ms, int *launches, int reset);
/* The name (up to the template arguments) of the kernel this bank launches for class `which`, as a profiler prints it: lets a harness
 * check that counter summaries it holds (profiles/pmc_summary*.json) belong to the kernel that actually ran. */
int jaero_profile_kernel(jaero_ctx *ctx, int which, char *buf, int cap);

#ifdef __cplusplus
}
#endif
#endif /* JAERO_HIP_H */
