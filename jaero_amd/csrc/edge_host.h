// edge_host.h -- the two multi-GPU edge operations of the north star behind the C ABI: fan out shared PCM, gather soft bits.
//
// The demodulator path shards by channel with NO steady-state exchange (the reference runs its two stereo burst channels as two
// unrelated objects, JAERO/audioburstoqpskdemodulator.cpp:8-10); what a multi-GPU host needs from a communication library is only the input
// fan-out and the result gather at the edges.  One process (or thread) per GPU, RCCL point-to-point sends grouped per call (xGMI is
// point-to-point; the volumes are tiny next to one link).  Same semantics as jaero_amd/dist.py (which bench.py uses over torch.distributed):
// contiguous channel ranges [rank * N / W, (rank + 1) * N / W).
// RCCL is loaded with dlopen the first time a communicator of more than one rank (or an explicit single-rank RCCL communicator: tests) is
// created, so libjaero_hip.so has no link-time dependency on it and single-GPU hosts never load it.
#pragma once
#include <dlfcn.h>

struct JRccl
{
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, /* ncclUniqueId by value: 128 bytes */ struct JId128, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*CommAbort)(void *) = nullptr; // optional: used when a group fails half-queued
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
struct JId128 { char b[JAERO_COMM_ID_BYTES]; };
static JRccl g_rccl;

static int rccl_load()
{
    if (g_rccl.lib) return 0;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(JAERO_ENOTSUP, "RCCL is not available (%s): multi-GPU edge operations need librccl.so", dlerror());
#define SYM(field, name) do { *(void **)(&g_rccl.field) = dlsym(h, name); if (!g_rccl.field) return fail(JAERO_ENOTSUP, "librccl.so lacks %s", name); } while (0)
    SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
    SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv"); SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    *(void **)(&g_rccl.CommAbort) = dlsym(h, "ncclCommAbort");
    g_rccl.lib = h;
    return 0;
}
#define RCCLCHK(call, what) do { const int r_ = (call); if (r_ != 0) return fail(JAERO_EHIP, "RCCL %s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "error"); } while (0)
#define JRCCL_CHAR 0 /* ncclInt8 / ncclChar: payloads are moved as bytes */
// An open ncclGroupStart must be closed on EVERY way out: a Send / Recv that fails inside the group returns through RCCLCHK, and with the
// thread's group depth left above zero every later RCCL call on that thread (ncclCommDestroy included) would be queued or hang.  But closing
// a group COMMITS what was queued before the failure, and the peers' matching operations of a half-queued group never come: the stream or
// the communicator may then block for good.  So the error path closes the group, ABORTS the communicator (ncclCommAbort where the library
// has it: queued work is cancelled) and marks the jaero_comm broken -- every later call on it fails with JAERO_EHIP instead of hanging.
// Any RCCL error invalidates the jaero_comm; destroy it and create a new one on every rank.
struct jaero_comm;
static void jaero_comm_break(jaero_comm *c);
struct JRcclGroup
{
    jaero_comm *comm;
    bool open = false;
    explicit JRcclGroup(jaero_comm *c) : comm(c) {}
    int start() { const int r = g_rccl.GroupStart(); open = (r == 0); return r; }
    int end() { open = false; const int r = g_rccl.GroupEnd(); if (r != 0) jaero_comm_break(comm); return r; }
    // an early return between start() and end(): the communicator is aborted FIRST (its queued operations die with it), then the group is closed
    ~JRcclGroup() { if (open) { jaero_comm_break(comm); g_rccl.GroupEnd(); } }
};

struct jaero_comm
{
    int device = 0, rank = 0, world = 1;
    void *nccl = nullptr;          // ncclComm_t, or null: a one-rank communicator without RCCL
    int16_t *stage = nullptr;      // fan-out source: packed slices of the peers
    size_t stage_elems = 0;
    hipEvent_t stage_ev = nullptr; // recorded behind the sends that read `stage`: the next call's packing (on whatever stream) waits for it
    bool stage_busy = false;
    bool broken = false;           // an RCCL call failed inside a group: the communicator was aborted, nothing more can be sent through it
};
static void jaero_comm_break(jaero_comm *c)
{
    if (!c || c->broken) return;
    c->broken = true;
    if (c->nccl && g_rccl.CommAbort) { g_rccl.CommAbort(c->nccl); c->nccl = nullptr; } // (without ncclCommAbort the handle is kept for jaero_comm_destroy)
}
#define COMM_BROKEN_CHK(c, who) do { if ((c) && (c)->broken) return fail(JAERO_EHIP, who ": an earlier RCCL operation on this communicator failed; it was aborted -- destroy it and create a new one on every rank"); } while (0)

static inline void shard_range(int nch_total, int rank, int world, int &lo, int &hi)
{
    lo = (int)(((long long)rank * nch_total) / world);
    hi = (int)(((long long)(rank + 1) * nch_total) / world);
}

// frames [nsamples][nch_total] -> dst [nsamples][hi - lo]: the column slice of one rank, contiguous
__global__ void k_edge_pack(const int16_t *__restrict__ frames, int nch_total, int lo, int width, int nsamples, int16_t *__restrict__ dst)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, n = (size_t)nsamples * width;
    if (i >= n) return;
    const size_t s = i / width, c = i - s * width;
    dst[i] = frames[s * nch_total + lo + c];
}

extern "C" int jaero_shard_range(int nch_total, int rank, int world, int *lo, int *hi)
{
    if (nch_total < 0 || world <= 0 || rank < 0 || rank >= world || !lo || !hi) return fail(JAERO_EINVAL, "jaero_shard_range: bad arguments");
    shard_range(nch_total, rank, world, *lo, *hi);
    return 0;
}

extern "C" int jaero_comm_get_unique_id(void *id)
{
    if (!id) return fail(JAERO_EINVAL, "jaero_comm_get_unique_id: null id");
    int rc = rccl_load();
    if (rc) return rc;
    RCCLCHK(g_rccl.GetUniqueId(id), "ncclGetUniqueId");
    return 0;
}

extern "C" int jaero_comm_create(int device, int rank, int world, const void *id, jaero_comm **out)
{
    if (!out || world <= 0 || rank < 0 || rank >= world || (world > 1 && !id)) return fail(JAERO_EINVAL, "jaero_comm_create: bad arguments");
    *out = nullptr;
    HIPCHK(hipSetDevice(device));
    jaero_comm *c = new jaero_comm();
    c->device = device; c->rank = rank; c->world = world;
    if (id)
    {
        int rc = rccl_load();
        if (rc) { delete c; return rc; }
        JId128 u;
        memcpy(u.b, id, sizeof u.b);
        const int r = g_rccl.CommInitRank(&c->nccl, world, u, rank);
        if (r != 0) { delete c; return fail(JAERO_EHIP, "RCCL ncclCommInitRank: %s", g_rccl.GetErrorString(r)); }
    }
    *out = c;
    return 0;
}

extern "C" void jaero_comm_destroy(jaero_comm *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    if (c->nccl && c->broken)
    {
        // a failed group whose communicator could not be aborted (this librccl has no ncclCommAbort): ncclCommDestroy on it can wait forever for
        // the half-queued group, which is the hang the abort was there to remove -- the handle is leaked instead, and said so
        fail(JAERO_EHIP, "jaero_comm_destroy: the communicator had failed and librccl offers no ncclCommAbort; its RCCL handle is leaked rather than destroyed");
    }
    else if (c->nccl) g_rccl.CommDestroy(c->nccl);
    if (c->stage_ev) hipEventDestroy(c->stage_ev);
    if (c->stage) hipFree(c->stage);
    delete c;
}

extern "C" int jaero_fan_out_pcm(jaero_comm *c, int src, const int16_t *d_frames_all, int nsamples, int nch_total, int16_t *d_mine, void *stream)
{
    if (!c || src < 0 || src >= c->world || nsamples < 0 || nch_total < 0 || !d_mine || (c->rank == src && !d_frames_all))
        return fail(JAERO_EINVAL, "jaero_fan_out_pcm: bad arguments");
    COMM_BROKEN_CHK(c, "jaero_fan_out_pcm");
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    int lo, hi;
    shard_range(nch_total, c->rank, c->world, lo, hi);
    auto pack = [&](int l, int h, int16_t *dst) {
        const size_t n = (size_t)nsamples * (h - l);
        if (n) hipLaunchKernelGGL(k_edge_pack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_frames_all, nch_total, l, h - l, nsamples, dst);
    };
    if (c->rank != src)
    {
        if (!c->nccl) return fail(JAERO_EINVAL, "jaero_fan_out_pcm: a one-rank communicator has no peer %d", src);
        RCCLCHK(g_rccl.Recv(d_mine, (size_t)nsamples * (hi - lo) * sizeof(int16_t), JRCCL_CHAR, src, c->nccl, st), "ncclRecv");
        return 0;
    }
    // the source packs every rank's column slice contiguously (its own straight into d_mine) and sends them in one group.  The staging
    // buffer belongs to the communicator: a call on another stream than the previous one waits for that call's sends first.
    if (c->stage_busy) HIPCHK(hipStreamWaitEvent(st, c->stage_ev, 0));
    const size_t need = (size_t)nsamples * (nch_total - (hi - lo));
    if (c->nccl && need > c->stage_elems)
    {
        if (c->stage) HIPCHK(hipFree(c->stage));
        c->stage = nullptr; c->stage_elems = 0;
        if (hipMalloc(&c->stage, need * sizeof(int16_t)) != hipSuccess) return fail(JAERO_ENOMEM, "jaero_fan_out_pcm: out of device memory");
        c->stage_elems = need;
    }
    pack(lo, hi, d_mine);
    if (c->nccl)
    {
        size_t off = 0;
        // (a one-rank RCCL communicator -- tests -- sends its own slice to itself through RCCL as well)
        const bool self_loop = c->world == 1;
        int16_t *self_buf = nullptr;
        if (self_loop)
        {
            if ((size_t)nsamples * (hi - lo) > c->stage_elems)
            {
                if (c->stage) HIPCHK(hipFree(c->stage));
                c->stage_elems = (size_t)nsamples * (hi - lo);
                if (hipMalloc(&c->stage, c->stage_elems * sizeof(int16_t)) != hipSuccess) return fail(JAERO_ENOMEM, "jaero_fan_out_pcm: out of device memory");
            }
            self_buf = c->stage;
            pack(lo, hi, self_buf);
        }
        for (int r = 0; r < c->world; r++)
            if (r != src)
            {
                int l, h;
                shard_range(nch_total, r, c->world, l, h);
                pack(l, h, c->stage + off);
                off += (size_t)nsamples * (h - l);
            }
        HIPCHK(hipGetLastError());
        JRcclGroup grp(c);
        RCCLCHK(grp.start(), "ncclGroupStart");
        off = 0;
        for (int r = 0; r < c->world; r++)
            if (r != src)
            {
                int l, h;
                shard_range(nch_total, r, c->world, l, h);
                const size_t n = (size_t)nsamples * (h - l);
                RCCLCHK(g_rccl.Send(c->stage + off, n * sizeof(int16_t), JRCCL_CHAR, r, c->nccl, st), "ncclSend");
                off += n;
            }
        if (self_loop)
        {
            const size_t n = (size_t)nsamples * (hi - lo) * sizeof(int16_t);
            RCCLCHK(g_rccl.Send(self_buf, n, JRCCL_CHAR, 0, c->nccl, st), "ncclSend (self)");
            RCCLCHK(g_rccl.Recv(d_mine, n, JRCCL_CHAR, 0, c->nccl, st), "ncclRecv (self)");
        }
        RCCLCHK(grp.end(), "ncclGroupEnd");
        if (!c->stage_ev) HIPCHK(hipEventCreateWithFlags(&c->stage_ev, hipEventDisableTiming));
        HIPCHK(hipEventRecord(c->stage_ev, st));
        c->stage_busy = true;
    }
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int jaero_gather_softbits(jaero_comm *c, int dst, const int16_t *d_soft, const int *d_counts, int nch_total, int cap,
                                     int16_t *d_soft_all, int *d_counts_all, void *stream)
{
    if (!c || dst < 0 || dst >= c->world || nch_total < 0 || cap <= 0 || !d_soft || !d_counts || (c->rank == dst && (!d_soft_all || !d_counts_all)))
        return fail(JAERO_EINVAL, "jaero_gather_softbits: bad arguments");
    COMM_BROKEN_CHK(c, "jaero_gather_softbits");
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    int lo, hi;
    shard_range(nch_total, c->rank, c->world, lo, hi);
    const size_t mine_b = (size_t)(hi - lo) * cap * sizeof(int16_t), mine_c = (size_t)(hi - lo) * sizeof(int);
    if (c->rank != dst)
    {
        if (!c->nccl) return fail(JAERO_EINVAL, "jaero_gather_softbits: a one-rank communicator has no peer %d", dst);
        JRcclGroup grp(c);
        RCCLCHK(grp.start(), "ncclGroupStart");
        RCCLCHK(g_rccl.Send(d_soft, mine_b, JRCCL_CHAR, dst, c->nccl, st), "ncclSend");
        RCCLCHK(g_rccl.Send(d_counts, mine_c, JRCCL_CHAR, dst, c->nccl, st), "ncclSend");
        RCCLCHK(grp.end(), "ncclGroupEnd");
        return 0;
    }
    const bool self_loop = c->nccl && c->world == 1;
    if (!self_loop)
    {
        HIPCHK(hipMemcpyAsync(d_soft_all + (size_t)lo * cap, d_soft, mine_b, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpyAsync(d_counts_all + lo, d_counts, mine_c, hipMemcpyDeviceToDevice, st));
    }
    if (c->nccl)
    {
        JRcclGroup grp(c);
        RCCLCHK(grp.start(), "ncclGroupStart");
        for (int r = 0; r < c->world; r++)
        {
            if (r == dst && !self_loop) continue;
            int l, h;
            shard_range(nch_total, r, c->world, l, h);
            RCCLCHK(g_rccl.Recv(d_soft_all + (size_t)l * cap, (size_t)(h - l) * cap * sizeof(int16_t), JRCCL_CHAR, r, c->nccl, st), "ncclRecv");
            RCCLCHK(g_rccl.Recv(d_counts_all + l, (size_t)(h - l) * sizeof(int), JRCCL_CHAR, r, c->nccl, st), "ncclRecv");
        }
        if (self_loop)
        {
            RCCLCHK(g_rccl.Send(d_soft, mine_b, JRCCL_CHAR, 0, c->nccl, st), "ncclSend (self)");
            RCCLCHK(g_rccl.Send(d_counts, mine_c, JRCCL_CHAR, 0, c->nccl, st), "ncclSend (self)");
        }
        RCCLCHK(grp.end(), "ncclGroupEnd");
    }
    return 0;
}
