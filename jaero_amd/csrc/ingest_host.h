// ingest_host.h -- batched ingest in front of jaero_write (the part that is not networking).
//
// The reference delivers network audio one channel at a time: ZMQAudioReceiver::process receives (topic, u32 sample rate,
// PCM of at most 192000 bytes) and emits recAudio(QByteArray, quint32) (JAERO/zmq_audioreceiver.cpp:40-79), which lands
// in the demodulator's dataReceived slot: a sample-rate check followed by writeData(audio, audio.length())
// (JAERO/oqpskdemodulator.cpp:686-693, burstoqpskdemodulator.cpp dataReceived, mskdemodulator.cpp:528-537).
// A bank wants the opposite shape: one jaero_write of the same number of samples for every channel.  jaero_ingest is
// the adaptor: per-channel FIFOs in one pinned host allocation; jaero_ingest_push is dataReceived for one channel
// (messages of any size, in any channel order); jaero_ingest_pump hands whole chunks -- the samples every channel has
// in common -- to jaero_write straight from pinned memory.
// Layout: the FIFOs are cut into slots of `chunk` samples, ring[slot][channel][chunk]: a push copies a message once,
// to its channel's row of the slot(s) it falls in; a full slot IS the channel-major buffer jaero_write takes, so the
// pump passes it on without another host copy.  All channels are consumed in lockstep, so there is one read position.  Demodulator output does not depend on how a stream is cut
// into writes, so the soft bits equal those of the reference fed the same messages.
// No sockets here: the transport stays with the caller (INTEGRATION.md shows the ZMQ loop that feeds push()).
#pragma once

struct jaero_ingest
{
    jaero_ctx *bank = nullptr;
    int nch = 0, chunk = 0, nslots = 0;
    int16_t *ring = nullptr;  // pinned [nslots][nch][chunk]
    int16_t *stage = nullptr; // pinned [nch][chunk]: short writes (flush / realignment after one) are gathered here
    std::vector<hipEvent_t> slot_ev; // recorded after the jaero_write that read the slot; + one for `stage`
    std::vector<char> slot_busy;
    std::vector<long long> wpos; // per channel: samples pushed so far
    long long rpos = 0;          // samples of every channel handed to jaero_write so far
    long long rate_warnings = 0, dropped_samples = 0;

    long long capacity() const { return (long long)nslots * chunk; }
    int16_t *at(long long abs, int ch) const
    {
        const long long k = abs / chunk;
        return ring + ((size_t)(k % nslots) * nch + ch) * chunk + (abs - k * chunk);
    }
};

extern "C" int jaero_ingest_create(jaero_ctx *bank, int chunk_samples, int capacity_samples, jaero_ingest **out)
{
    if (!bank || !out) return fail(JAERO_EINVAL, "jaero_ingest_create: null argument");
    if (chunk_samples <= 0 || chunk_samples > bank->max_write)
        return fail(JAERO_EINVAL, "jaero_ingest_create: chunk_samples %d must be in 1..max_write_samples (%d)", chunk_samples, bank->max_write);
    HIPCHK(hipSetDevice(bank->device));
    jaero_ingest *g = new (std::nothrow) jaero_ingest();
    if (!g) return fail(JAERO_ENOMEM, "jaero_ingest_create: out of memory");
    g->bank = bank;
    g->nch = jaero_num_channels(bank);
    g->chunk = chunk_samples;
    g->nslots = (capacity_samples + chunk_samples - 1) / chunk_samples;
    if (g->nslots < 3) g->nslots = 3;
    g->wpos.assign(g->nch, 0);
    g->slot_busy.assign(g->nslots + 1, 0);
    g->slot_ev.assign(g->nslots + 1, nullptr);
    hipError_t e = hipHostMalloc((void **)&g->ring, sizeof(int16_t) * (size_t)g->nslots * g->nch * g->chunk, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&g->stage, sizeof(int16_t) * (size_t)g->nch * g->chunk, hipHostMallocDefault);
    for (int k = 0; k <= g->nslots && e == hipSuccess; k++) e = hipEventCreateWithFlags(&g->slot_ev[k], hipEventDisableTiming);
    if (e != hipSuccess)
    {
        const int rc = fail(JAERO_ENOMEM, "jaero_ingest_create: pinned allocation failed: %s", hipGetErrorString(e));
        for (hipEvent_t ev : g->slot_ev) if (ev) (void)hipEventDestroy(ev);
        if (g->stage) (void)hipHostFree(g->stage);
        if (g->ring) (void)hipHostFree(g->ring);
        delete g;
        return rc;
    }
    *out = g;
    return 0;
}

extern "C" void jaero_ingest_destroy(jaero_ingest *g)
{
    if (!g) return;
    (void)hipSetDevice(g->bank->device);
    for (int k = 0; k <= g->nslots; k++)
    {
        if (g->slot_busy[k]) (void)hipEventSynchronize(g->slot_ev[k]);
        (void)hipEventDestroy(g->slot_ev[k]);
    }
    (void)hipHostFree(g->stage);
    (void)hipHostFree(g->ring);
    delete g;
}

static int ingest_wait(jaero_ingest *g, int k)
{
    if (g->slot_busy[k]) { HIPCHK(hipEventSynchronize(g->slot_ev[k])); g->slot_busy[k] = 0; }
    return 0;
}

// = dataReceived(audio, sampleRate) of channel `channel`.  nbytes of little-endian int16 mono; as ZMQAudioReceiver's
// receive buffer, at most 192000 bytes of one message are taken; an odd trailing byte is ignored (writeData works on
// len/2 samples).  Returns 0, JAERO_W_RATE (> 0: sample rate differs from the bank's Fs; the OQPSK kinds only log
// that and demodulate anyway, and so does this; MSK would re-create the channel at the new rate, which a bank that
// shares Fs cannot do: JAERO_ENOTSUP, nothing queued), or JAERO_EOVERFLOW when the channel's FIFO cannot hold the
// message (nothing queued: pump and push again).
extern "C" int jaero_ingest_push(jaero_ingest *g, int channel, const void *pcm_bytes, int nbytes, unsigned sample_rate)
{
    if (!g || channel < 0 || channel >= g->nch || nbytes < 0 || (nbytes > 0 && !pcm_bytes)) return fail(JAERO_EINVAL, "jaero_ingest_push: bad arguments");
    int rc = 0;
    if ((double)sample_rate != g->bank->settings[0].Fs)
    {
        if (g->bank->settings[0].kind == JAERO_KIND_MSK || g->bank->settings[0].kind == JAERO_KIND_BURST_MSK)
            return fail(JAERO_ENOTSUP, "jaero_ingest_push: channel %d sample rate %u differs from the bank's %g (an MSK channel would switch rate; create a bank at that rate)",
                        channel, sample_rate, g->bank->settings[0].Fs);
        g->rate_warnings++;
        rc = JAERO_W_RATE;
    }
    if (nbytes > 192000) nbytes = 192000;
    int n = nbytes / 2;
    long long w = g->wpos[channel];
    if (w + n - g->rpos > g->capacity())
    {
        g->dropped_samples += n;
        return fail(JAERO_EOVERFLOW, "jaero_ingest_push: channel %d FIFO full (%lld queued, %d offered, capacity %lld)", channel, w - g->rpos, n, g->capacity());
    }
    const int16_t *src = (const int16_t *)pcm_bytes;
    while (n > 0)
    {
        const long long k = w / g->chunk;
        const int off = (int)(w - k * g->chunk);
        const int m = n < g->chunk - off ? n : g->chunk - off;
        const int rcw = ingest_wait(g, (int)(k % g->nslots)); // a slot handed to jaero_write is reusable once its copy has run
        if (rcw) return rcw;
        memcpy(g->at(w, channel), src, sizeof(int16_t) * (size_t)m);
        src += m; w += m; n -= m;
    }
    g->wpos[channel] = w;
    return rc;
}

extern "C" int jaero_ingest_queued(const jaero_ingest *g, int channel)
{
    if (!g || channel < -1 || channel >= g->nch) return fail(JAERO_EINVAL, "jaero_ingest_queued: bad arguments");
    if (channel >= 0) return (int)(g->wpos[channel] - g->rpos);
    long long m = g->wpos[0];
    for (int c = 1; c < g->nch; c++) m = g->wpos[c] < m ? g->wpos[c] : m;
    return (int)(m - g->rpos); // what every channel has in common
}

// Write every whole chunk all channels have in common (and, with flush != 0, the common remainder too) with
// jaero_write on `stream`.  *chunks receives the number of jaero_write calls made.  A whole slot goes to jaero_write
// as it lies; a short write (flush) is gathered into `stage`, and the next write is cut so that the read position
// returns to a slot boundary.
extern "C" int jaero_ingest_pump(jaero_ingest *g, int flush, void *stream, int *chunks)
{
    if (!g) return fail(JAERO_EINVAL, "jaero_ingest_pump: null ctx");
    HIPCHK(hipSetDevice(g->bank->device));
    hipStream_t st = (hipStream_t)stream;
    int done = 0, rc = 0;
    for (;;)
    {
        const int common = jaero_ingest_queued(g, -1);
        const int off = (int)(g->rpos % g->chunk);
        int n = 0;
        if (common >= g->chunk - off) n = g->chunk - off;
        else if (flush && common > 0) n = common;
        if (!n) break;
        const int slot = (int)((g->rpos / g->chunk) % g->nslots);
        if (n == g->chunk)
        {
            rc = jaero_write(g->bank, g->ring + (size_t)slot * g->nch * g->chunk, n, JAERO_PCM_CHANNEL_MAJOR, 0, st);
            if (rc) break;
            HIPCHK(hipEventRecord(g->slot_ev[slot], st));
            g->slot_busy[slot] = 1;
        }
        else
        {
            rc = ingest_wait(g, g->nslots);
            if (rc) break;
            for (int c = 0; c < g->nch; c++) memcpy(g->stage + (size_t)c * n, g->at(g->rpos, c), sizeof(int16_t) * (size_t)n);
            rc = jaero_write(g->bank, g->stage, n, JAERO_PCM_CHANNEL_MAJOR, 0, st);
            if (rc) break;
            HIPCHK(hipEventRecord(g->slot_ev[g->nslots], st));
            g->slot_busy[g->nslots] = 1;
        }
        g->rpos += n;
        done++;
    }
    if (chunks) *chunks = done;
    return rc;
}

// counters: [0] pushes whose sample rate differed from the bank's, [1] samples refused because a FIFO was full,
// [2] samples per channel handed to jaero_write so far
extern "C" int jaero_ingest_stats(const jaero_ingest *g, long long *three)
{
    if (!g || !three) return fail(JAERO_EINVAL, "jaero_ingest_stats: null argument");
    three[0] = g->rate_warnings; three[1] = g->dropped_samples; three[2] = g->rpos;
    return 0;
}
