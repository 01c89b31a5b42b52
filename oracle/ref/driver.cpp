// TEST INFRASTRUCTURE ONLY.  Driver for oracle/_ref/jaero_ref: runs the UNMODIFIED reference demodulators
// (compiled from /root/reference/JAERO/*.cpp where they lie; nothing is copied into this repo) on a raw int16
// file and dumps what they emit.  ONE CHANNEL PER PROCESS: the reference keeps per-channel state in
// function-local statics (JAERO/oqpskdemodulator.cpp:393,487,496,498,540,641,652; JAERO/mskdemodulator.cpp:434,493).
//
//   jaero_ref oqpsk|msk <in.s16> <outprefix> [key=value ...]
//       writes <outprefix>.soft   int16 soft bits exactly as passed to processDemodulatedSoftBits
//              <outprefix>.status float64 rows [n_estimate, freq_est, freq_center, mse, ebno, signal]
//                                 one row per FreqOffsetEstimateSlot call
//   jaero_ref burstoqpsk|burstmsk <in.s16> <outprefix> [key=value ...]
//       same outputs; the soft stream keeps the -1 start-of-burst markers; one status row per SignalStatus emission,
//       plus <outprefix>.events float64 rows [sample index of the write that carried it, kind, value]:
//       kind 0 = SignalStatus(value), 1 = EbNoMeasurmentSignal(value), 2 = Plottables freq_est
//   jaero_ref aerol <in.s16 soft bits> <out.txt> fb=10500|1200|600|8400 [group=32] [burst=0]
//       the UNMODIFIED AeroL (JAERO/aerol.cpp): soft bits are handed to processDemodulatedSoftBits in groups, what it writes to its
//       sink device (signal-unit dumps, "Bad CRC", ...) goes to <out.txt>; DataCarrierDetect emissions are interleaved as
//       lines "#DCD <0|1> <index of the first soft bit of the group that carried it>"
//   jaero_ref viterbi_cont <in.u8> <out.u8> blocklen=N [padding=24]
//       JConvolutionalCodec::Decode_Continuous per block of N soft bytes (code 2,7,{109,79} as AeroL, aerol.cpp:936-940)
//   jaero_ref viterbi_soft <in.u8> <out.u8> blocklen=N
//       JConvolutionalCodec::Decode_soft per block
//   jaero_ref fft|ifft|fftr|ifftr <in.f64> <out.f64> n=N
//       FFTWrapper / FFTrWrapper transform (golden-vector pinning of the JFFT shim)
//   jaero_ref fastfir <in.c128> <out.c128> alpha=0.6 K=2048 nfft=4096 Fs=48000 fsym=5250
//   jaero_ref time oqpsk|msk <in.s16> [key=value ...]    -> prints seconds spent inside writeData
#include <functional>
#include <QCoreApplication>
#include <QVector>
#include <QByteArray>
#include <QFile>
#include <QString>
#include <QMap>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <vector>
#include "oqpskdemodulator.h"
#include "mskdemodulator.h"
#include "burstoqpskdemodulator.h"
#include "burstmskdemodulator.h"
#include "fftwrapper.h"
#include "fftrwrapper.h"
#include "jconvolutionalcodec.h"
#include "aerol.h"
#include <QBuffer>

static QMap<QString, QString> kv;
static double getd(const char *k, double def) { return kv.contains(k) ? kv[k].toDouble() : def; }
static int geti(const char *k, int def) { return kv.contains(k) ? kv[k].toInt() : def; }

static QByteArray readall(const char *path)
{
    QFile f(path);
    if (!f.open(QIODevice::ReadOnly)) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    return f.readAll();
}
static void writeall(const QString &path, const void *p, size_t n)
{
    QFile f(path);
    if (!f.open(QIODevice::WriteOnly)) { fprintf(stderr, "cannot write %s\n", path.toUtf8().data()); exit(2); }
    f.write((const char *)p, n);
}

struct Capture
{
    std::vector<short> soft;
    std::vector<double> status;
    double freq_est = 0, freq_center = 0, mse = 0, ebno = 0;
    double nest = 0;
};

template <class DEMOD>
static void hook(DEMOD &d, Capture &c)
{
    QObject::connect(&d, &DEMOD::processDemodulatedSoftBits, [&c](const QVector<short> &v) { for (int i = 0; i < v.size(); i++) c.soft.push_back(v[i]); });
    QObject::connect(&d, &DEMOD::Plottables, [&c](double fe, double fc, double) { c.freq_est = fe; c.freq_center = fc; });
    QObject::connect(&d, &DEMOD::MSESignal, [&c](double m) { c.mse = m; });
    QObject::connect(&d, &DEMOD::EbNoMeasurmentSignal, [&c](double e) { c.ebno = e; });
    QObject::connect(&d, &DEMOD::SignalStatus, [&c](bool s) {
        c.status.push_back(c.nest); c.status.push_back(c.freq_est); c.status.push_back(c.freq_center);
        c.status.push_back(c.mse); c.status.push_back(c.ebno); c.status.push_back(s ? 1.0 : 0.0);
        c.nest += 1.0;
    });
}

// burst demodulators: no MSESignal; SignalStatus(true/false) at burst start/timeout, EbNo once per burst
static long g_write_start = 0;
template <class DEMOD>
static void hook_burst(DEMOD &d, Capture &c)
{
    QObject::connect(&d, &DEMOD::processDemodulatedSoftBits, [&c](const QVector<short> &v) { for (int i = 0; i < v.size(); i++) c.soft.push_back(v[i]); });
    QObject::connect(&d, &DEMOD::Plottables, [&c](double fe, double fc, double) {
        c.freq_est = fe; c.freq_center = fc;
        c.status.push_back((double)g_write_start); c.status.push_back(2.0); c.status.push_back(fe); });
    QObject::connect(&d, &DEMOD::EbNoMeasurmentSignal, [&c](double e) {
        c.ebno = e;
        c.status.push_back((double)g_write_start); c.status.push_back(1.0); c.status.push_back(e); });
    QObject::connect(&d, &DEMOD::SignalStatus, [&c](bool s) {
        c.status.push_back((double)g_write_start); c.status.push_back(0.0); c.status.push_back(s ? 1.0 : 0.0); });
}

// set_at=N [set_fb= set_Fs= set_lockingbw= set_freq_center= set_power=]: setSettings on the live object before the write that starts at or
// after sample N (a user changing the rate in the settings dialog; MskDemodulator::dataReceived on audio at another rate)
static std::function<void()> g_set_again;
// flags_at=N flags_afc= flags_sql= flags_cpureduce= (and flags_at2=N flags2_*): setAFC / setSQL / setCPUReduce on the live object before the write that
// starts at or after sample N (a user ticking the boxes while it runs) -- pins the oracle's mid-stream flag changes, which the GPU flags-matrix tests build on
static std::function<void(int)> g_flags_again;

template <class DEMOD>
static double feed(DEMOD &d, const QByteArray &pcm)
{
    long set_at = (long)getd("set_at", -1), set_at2 = (long)getd("set_at2", -1); // set_at2: a second setSettings (inside the first one's transient)
    int chunk = geti("chunk", 4096);
    long dcd_at = (long)getd("dcd_at", -1);         // sample index at which DCDstatSlot(true) is called (chunk aligned)
    long dcd_off_at = (long)getd("dcd_off_at", -1);
    long cf_at = (long)getd("center_at", -1);       // sample index at which CenterFreqChangedSlot(center_hz) is called
    long flags_at = (long)getd("flags_at", -1), flags_at2 = (long)getd("flags_at2", -1);
    double cf_hz = getd("center_hz", 0);
    long nsamp = pcm.size() / 2;
    const char *p = pcm.constData();
    double secs = 0;
    for (long s = 0; s < nsamp;)
    {
        g_write_start = s; // what a slot called between two writes emits is stamped with the write that follows it
        if (dcd_at >= 0 && s >= dcd_at) { d.DCDstatSlot(true); dcd_at = -1; }
        if (dcd_off_at >= 0 && s >= dcd_off_at) { d.DCDstatSlot(false); dcd_off_at = -1; }
        if (cf_at >= 0 && s >= cf_at) { d.CenterFreqChangedSlot(cf_hz); cf_at = -1; }
        if (set_at >= 0 && s >= set_at) { if (g_set_again) g_set_again(); set_at = -1; }
        if (set_at2 >= 0 && s >= set_at2) { if (g_set_again) g_set_again(); set_at2 = -1; }
        if (flags_at >= 0 && s >= flags_at) { if (g_flags_again) g_flags_again(1); flags_at = -1; }
        if (flags_at2 >= 0 && s >= flags_at2) { if (g_flags_again) g_flags_again(2); flags_at2 = -1; }
        long n = chunk;
        if (s + n > nsamp) n = nsamp - s;
        g_write_start = s;
        auto t0 = std::chrono::steady_clock::now();
        d.writeData(p + 2 * s, 2 * n);
        secs += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        s += n;
    }
    return secs;
}

static double run_oqpsk(const QByteArray &pcm, Capture &c)
{
    OqpskDemodulator d(0);
    hook(d, c);
    OqpskDemodulator::Settings s;
    s.fb = getd("fb", 10500); s.Fs = getd("Fs", 48000); s.freq_center = getd("freq_center", 8000);
    s.lockingbw = getd("lockingbw", 10500); s.coarsefreqest_fft_power = geti("power", 14);
    s.signalthreshold = getd("threshold", 0.65);
    d.setAFC(geti("afc", 0)); d.setSQL(geti("sql", 0)); d.setCPUReduce(geti("cpureduce", 0));
    d.DCDstatSlot(false);
    d.setSettings(s);
    d.start();
    g_set_again = [&d, s]() mutable {
        s.fb = getd("set_fb", s.fb); s.Fs = getd("set_Fs", s.Fs); s.freq_center = getd("set_freq_center", s.freq_center);
        s.lockingbw = getd("set_lockingbw", s.lockingbw); s.coarsefreqest_fft_power = geti("set_power", s.coarsefreqest_fft_power);
        d.setSettings(s);
    };
    g_flags_again = [&d](int which) {
        const char *pre = which == 1 ? "flags_" : "flags2_";
        d.setAFC(geti((std::string(pre) + "afc").c_str(), 0)); d.setSQL(geti((std::string(pre) + "sql").c_str(), 0));
        d.setCPUReduce(geti((std::string(pre) + "cpureduce").c_str(), 0));
    };
    const double secs = feed(d, pcm);
    g_set_again = nullptr; g_flags_again = nullptr;
    return secs;
}

static double run_msk(const QByteArray &pcm, Capture &c)
{
    MskDemodulator d(0);
    hook(d, c);
    MskDemodulator::Settings s;
    s.fb = getd("fb", 1200); s.Fs = getd("Fs", 48000); s.freq_center = getd("freq_center", 1000);
    s.lockingbw = getd("lockingbw", 1800); s.coarsefreqest_fft_power = geti("power", 13);
    s.signalthreshold = getd("threshold", 0.5);
    d.setAFC(geti("afc", 0)); d.setSQL(geti("sql", 0)); d.setCPUReduce(geti("cpureduce", 0));
    d.DCDstatSlot(false);
    d.setSettings(s);
    d.start();
    g_set_again = [&d, s]() mutable {
        s.fb = getd("set_fb", s.fb); s.Fs = getd("set_Fs", s.Fs); s.freq_center = getd("set_freq_center", s.freq_center);
        s.lockingbw = getd("set_lockingbw", s.lockingbw); s.coarsefreqest_fft_power = geti("set_power", s.coarsefreqest_fft_power);
        d.setSettings(s);
    };
    g_flags_again = [&d](int which) {
        const char *pre = which == 1 ? "flags_" : "flags2_";
        d.setAFC(geti((std::string(pre) + "afc").c_str(), 0)); d.setSQL(geti((std::string(pre) + "sql").c_str(), 0));
        d.setCPUReduce(geti((std::string(pre) + "cpureduce").c_str(), 0));
    };
    const double secs = feed(d, pcm);
    g_set_again = nullptr; g_flags_again = nullptr;
    return secs;
}

// The burst classes leave members uninitialised (BurstOqpskDemodulator::rotator_freq is read on the first sample,
// JAERO/burstoqpskdemodulator.h:186, burstoqpskdemodulator.cpp:572): construct them in zeroed storage so that a run
// is deterministic.
struct BurstOqpskFeed : public BurstOqpskDemodulator
{
    BurstOqpskFeed() : BurstOqpskDemodulator(0) {}
    void DCDstatSlot(bool) {} // the burst OQPSK demodulator has no DCD input
    static void *operator new(size_t n) { return calloc(1, n); }
    static void operator delete(void *p) { free(p); }
};
struct BurstMskFeed : public BurstMskDemodulator
{
    BurstMskFeed() : BurstMskDemodulator(0) {}
    static void *operator new(size_t n) { return calloc(1, n); }
    static void operator delete(void *p) { free(p); }
};

static double run_burstoqpsk(const QByteArray &pcm, Capture &c)
{
    BurstOqpskFeed *dp = new BurstOqpskFeed();
    BurstOqpskFeed &d = *dp;
    hook_burst<BurstOqpskDemodulator>(d, c);
    BurstOqpskDemodulator::Settings s;
    s.fb = getd("fb", 10500); s.Fs = getd("Fs", 48000); s.freq_center = getd("freq_center", 8000);
    s.lockingbw = getd("lockingbw", 10500); s.coarsefreqest_fft_power = geti("power", 13);
    s.signalthreshold = getd("threshold", 0.6);
    d.setAFC(geti("afc", 0)); d.setSQL(geti("sql", 0)); d.setCPUReduce(geti("cpureduce", 0));
    d.setScatterPointType(BurstOqpskDemodulator::SPT_None);
    d.setSettings(s);
    d.start();
    g_set_again = [&d, s]() mutable {
        s.freq_center = getd("set_freq_center", s.freq_center); s.lockingbw = getd("set_lockingbw", s.lockingbw);
        s.signalthreshold = getd("set_threshold", s.signalthreshold);
        d.setSettings(s);
    };
    const double secs = feed(d, pcm);
    g_set_again = nullptr;
    return secs;
}

static double run_burstmsk(const QByteArray &pcm, Capture &c)
{
    BurstMskFeed *dp = new BurstMskFeed();
    BurstMskFeed &d = *dp;
    hook_burst<BurstMskDemodulator>(d, c);
    BurstMskDemodulator::Settings s;
    s.fb = getd("fb", 1200); s.Fs = getd("Fs", 48000); s.freq_center = getd("freq_center", 1000);
    s.lockingbw = getd("lockingbw", 1800); s.coarsefreqest_fft_power = geti("power", 13);
    s.signalthreshold = getd("threshold", 0.6); s.symbolspercycle = geti("symbolspercycle", 16);
    d.setAFC(geti("afc", 0)); d.setSQL(geti("sql", 0)); d.setCPUReduce(geti("cpureduce", 0));
    d.setScatterPointType(BurstMskDemodulator::SPT_None);
    d.DCDstatSlot(false);
    d.setSettings(s);
    d.start();
    g_set_again = [&d, s]() mutable {
        s.fb = getd("set_fb", s.fb); s.freq_center = getd("set_freq_center", s.freq_center); s.lockingbw = getd("set_lockingbw", s.lockingbw);
        s.signalthreshold = getd("set_threshold", s.signalthreshold);
        d.setSettings(s);
    };
    const double secs = feed(d, pcm);
    g_set_again = nullptr;
    return secs;
}

int main(int argc, char **argv)
{
    QCoreApplication app(argc, argv);
    if (argc < 4) { fprintf(stderr, "usage: see driver.cpp header\n"); return 2; }
    QString mode = argv[1];
    int first_kv = 4;
    if (mode == "time") first_kv = 4;
    for (int i = first_kv; i < argc; i++)
    {
        QString a = argv[i];
        int e = a.indexOf('=');
        if (e > 0) kv[a.left(e)] = a.mid(e + 1);
    }
    if (mode == "oqpsk" || mode == "msk")
    {
        QByteArray pcm = readall(argv[2]);
        Capture c;
        if (mode == "oqpsk") run_oqpsk(pcm, c); else run_msk(pcm, c);
        writeall(QString(argv[3]) + ".soft", c.soft.data(), c.soft.size() * sizeof(short));
        writeall(QString(argv[3]) + ".status", c.status.data(), c.status.size() * sizeof(double));
        return 0;
    }
    if (mode == "burstoqpsk" || mode == "burstmsk")
    {
        QByteArray pcm = readall(argv[2]);
        Capture c;
        if (mode == "burstoqpsk") run_burstoqpsk(pcm, c); else run_burstmsk(pcm, c);
        writeall(QString(argv[3]) + ".soft", c.soft.data(), c.soft.size() * sizeof(short));
        writeall(QString(argv[3]) + ".events", c.status.data(), c.status.size() * sizeof(double));
        return 0;
    }
    if (mode == "aerol")
    {
        QByteArray in = readall(argv[2]);
        const short *sp = (const short *)in.constData();
        long n = in.size() / 2;
        // realimag, muw and lastframeinfo are never initialised by the reference (JAERO/aerol.h:956,975,990): zeroed storage makes a
        // run deterministic
        struct AeroLZ : public AeroL
        {
            AeroLZ() : AeroL(0) {}
            static void *operator new(size_t n) { return calloc(1, n); }
            static void operator delete(void *p) { free(p); }
        };
        AeroLZ *ap = new AeroLZ();
        AeroL &a = *ap;
        QBuffer sink;
        sink.open(QIODevice::ReadWrite);
        a.ConnectSinkDevice(&sink);
        long gstart = 0;
        QByteArray dcdlog;
        QObject::connect(&a, &AeroL::DataCarrierDetect, [&](bool d) { sink.write(QString("#DCD %1 %2\n").arg(d ? 1 : 0).arg(gstart).toLatin1()); });
        // C channel (fb=8400): the voice bytes DecodeC hands to Voicesignal(data, hex), one "#V <hex>" line per frame
        QObject::connect(&a, static_cast<void (AeroL::*)(QByteArray &, QString &)>(&AeroL::Voicesignal),
                         [&](QByteArray &d, QString &) { sink.write(QByteArray("#V ") + d.toHex() + "\n"); });
        a.setSettings(getd("fb", 10500), geti("burst", 0) != 0);
        int group = geti("group", 32);
        // group=0: the groups a burst demodulator emits (burstoqpskdemodulator.cpp:546-585): a start-of-burst marker (negative) is one
        // entry, soft bits come in pairs, and a group is handed over once it holds >= 32 entries after a pair
        auto t0 = std::chrono::steady_clock::now();
        for (long s = 0; s < n;)
        {
            long m;
            if (group > 0) m = (s + group <= n) ? group : n - s;
            else
            {
                long e = s; int cnt = 0;
                while (e < n)
                {
                    if (sp[e] < 0) { e++; cnt++; continue; }
                    if (e + 1 < n) { e += 2; cnt += 2; } else { e++; cnt++; }
                    if (cnt >= 32) break;
                }
                m = e - s;
            }
            QVector<short> v(m);
            for (long i = 0; i < m; i++) v[i] = sp[s + i];
            gstart = s;
            a.processDemodulatedSoftBits(v);
            s += m;
        }
        double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        writeall(argv[3], sink.data().constData(), sink.data().size());
        printf("%.6f %ld\n", secs, n); // seconds inside processDemodulatedSoftBits (bench.py's cpu_baseline)
        return 0;
    }
    if (mode == "time")
    {
        QString which = argv[2];
        QByteArray pcm = readall(argv[3]);
        Capture c;
        double secs = (which == "oqpsk") ? run_oqpsk(pcm, c) : (which == "msk") ? run_msk(pcm, c)
                    : (which == "burstoqpsk") ? run_burstoqpsk(pcm, c) : run_burstmsk(pcm, c);
        printf("%.6f %ld %zu\n", secs, (long)(pcm.size() / 2), c.soft.size());
        return 0;
    }
    if (mode == "viterbi_cont" || mode == "viterbi_soft")
    {
        QByteArray in = readall(argv[2]);
        int blocklen = geti("blocklen", 5078);
        JConvolutionalCodec codec;
        QVector<quint16> polys; polys.push_back(109); polys.push_back(79);
        codec.SetCode(2, 7, polys, geti("padding", 24));
        std::vector<unsigned char> out;
        for (int off = 0; off + blocklen <= in.size(); off += blocklen)
        {
            QByteArray blk = in.mid(off, blocklen);
            QVector<int> bits = (mode == "viterbi_cont") ? codec.Decode_Continuous(blk) : codec.Decode_soft(blk, blocklen);
            // block header: number of bits (uint32 LE) then one byte per bit
            unsigned int n = bits.size();
            for (int k = 0; k < 4; k++) out.push_back((n >> (8 * k)) & 255);
            for (int k = 0; k < bits.size(); k++) out.push_back((unsigned char)bits[k]);
        }
        writeall(argv[3], out.data(), out.size());
        return 0;
    }
    if (mode == "fft" || mode == "ifft")
    {
        QByteArray in = readall(argv[2]);
        int n = geti("n", 16);
        QVector<cpx_type> x(n), y(n);
        memcpy(x.data(), in.constData(), sizeof(cpx_type) * n);
        FFTWrapper<double> f(n, mode == "ifft");
        f.transform(x, y);
        writeall(argv[3], y.data(), sizeof(cpx_type) * n);
        return 0;
    }
    if (mode == "fftr")
    {
        QByteArray in = readall(argv[2]);
        int n = geti("n", 16);
        QVector<double> x(n); QVector<cpx_type> y(n);
        memcpy(x.data(), in.constData(), sizeof(double) * n);
        FFTrWrapper<double> f(n);
        f.transform(x, y);
        writeall(argv[3], y.data(), sizeof(cpx_type) * n);
        return 0;
    }
    if (mode == "ifftr")
    {
        QByteArray in = readall(argv[2]);
        int n = geti("n", 16);
        QVector<cpx_type> x(n); QVector<double> y(n);
        memcpy(x.data(), in.constData(), sizeof(cpx_type) * n);
        FFTrWrapper<double> f(n);
        f.transform(x, y);
        writeall(argv[3], y.data(), sizeof(double) * n);
        return 0;
    }
    if (mode == "fastfir")
    {
        QByteArray in = readall(argv[2]);
        int n = in.size() / sizeof(cpx_type);
        QVector<cpx_type> x(n);
        memcpy(x.data(), in.constData(), sizeof(cpx_type) * n);
        JFastFir fir;
        RootRaisedCosine rrc;
        rrc.design(getd("alpha", 0.6), geti("K", 2048), getd("Fs", 48000), getd("fsym", 5250));
        fir.SetKernel(rrc.Points, geti("nfft", 4096));
        int chunk = geti("chunk", n);
        for (int off = 0; off < n; off += chunk)
        {
            int m = (off + chunk <= n) ? chunk : n - off;
            QVector<cpx_type> blk = x.mid(off, m);
            fir.update(blk);
            for (int i = 0; i < m; i++) x[off + i] = blk[i];
        }
        writeall(argv[3], x.data(), sizeof(cpx_type) * n);
        return 0;
    }
    fprintf(stderr, "unknown mode\n");
    return 2;
}
