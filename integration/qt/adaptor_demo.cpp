// adaptor_demo -- TEST DRIVER.  Wires a demodulator to the UNMODIFIED AeroL exactly as MainWindow does
// (JAERO/mainwindow.cpp:198-202,234-237) and writes what AeroL prints to its console device:
//
//   adaptor_demo ref|hip oqpsk|msk|burstoqpsk|burstmsk <in.s16> <out.txt> [fb=10500] [lockingbw=..] [freq_center=..] [chunk=4096] [prefb=..] [dump=1] [datarate=..]
//
// prefb: setSettings is first called with that bit rate, 20 000 samples of the input are written, then setSettings with the real one
// (a user changing the rate in the settings dialog): the rest must decode as if nothing had happened before.
//
// "ref" = the reference's own OqpskDemodulator / MskDemodulator; "hip" = HipOqpskDemodulator / HipMskDemodulator
// (integration/qt/hipdemodulator.h over libjaero_hip.so).  tests/test_qt_adaptor.py requires the two outputs to be equal.
// Built by `make -C oracle adaptor` into oracle/_ref/ (it links the reference's objects, which never enter this repository).
#include <QCoreApplication>
#include <QBuffer>
#include <functional>
#include <QFile>
#include <QMap>
#include <cstdio>
#include <cstdlib>
#include "oqpskdemodulator.h"
#include "mskdemodulator.h"
#include "burstoqpskdemodulator.h"
#include "burstmskdemodulator.h"
#include "aerol.h"
#include "hipdemodulator.h"

static QMap<QString, QString> kv;
static double getd(const char *k, double def) { return kv.contains(k) ? kv[k].toDouble() : def; }

struct AeroLZ : public AeroL // realimag, muw, lastframeinfo are never initialised by the reference (aerol.h:956,975,990)
{
    AeroLZ() : AeroL(0) {}
    static void *operator new(size_t n) { return calloc(1, n); }
    static void operator delete(void *p) { free(p); }
};

// The burst classes leave members uninitialised (BurstOqpskDemodulator::rotator_freq is read on the first sample,
// JAERO/burstoqpskdemodulator.h:186, burstoqpskdemodulator.cpp:572): constructed in zeroed storage so that a run is deterministic.
struct BurstOqpskZ : public BurstOqpskDemodulator
{
    BurstOqpskZ() : BurstOqpskDemodulator(0) {}
    void DCDstatSlot(bool) {} // the burst OQPSK demodulator has no DCD input
    static void *operator new(size_t n) { return calloc(1, n); }
    static void operator delete(void *p) { free(p); }
};
struct BurstMskZ : public BurstMskDemodulator
{
    BurstMskZ() : BurstMskDemodulator(0) {}
    static void *operator new(size_t n) { return calloc(1, n); }
    static void operator delete(void *p) { free(p); }
};

static QIODevice *g_sink = nullptr;
// set_at=N [set_lockingbw= set_freq_center=]: setSettings on the live demodulator in front of the first write at or behind sample N
static std::function<void()> g_set_again;
template <class DEMOD>
static void run_burst(DEMOD &d, AeroL &a, const QByteArray &pcm, int chunk)
{
    QObject::connect(&d, &DEMOD::processDemodulatedSoftBits, &a, &AeroL::processDemodulatedSoftBits);
    if (kv.contains("dump")) // the groups themselves, as AeroL receives them: "G <n>: v v v ..."
        QObject::connect(&d, &DEMOD::processDemodulatedSoftBits, [](const QVector<short> &v) {
            QString ln = QString("G %1:").arg(v.size());
            for (int i = 0; i < v.size(); i++) ln += QString(" %1").arg(v[i]);
            g_sink->write((ln + "\n").toLatin1());
        });
    d.start();
    const char *p = pcm.constData();
    const long nb = pcm.size();
    long set_at = kv.contains("set_at") ? 2L * kv["set_at"].toLong() : -1;
    for (long s = 0; s < nb; s += 2L * chunk)
    {
        if (set_at >= 0 && s >= set_at) { if (g_set_again) g_set_again(); set_at = -1; }
        d.write(p + s, (nb - s < 2L * chunk) ? nb - s : 2L * chunk);
    }
    d.stop();
}

template <class DEMOD>
static void run(DEMOD &d, AeroL &a, const QByteArray &pcm, int chunk)
{
    QObject::connect(&d, &DEMOD::processDemodulatedSoftBits, &a, &AeroL::processDemodulatedSoftBits);
    QObject::connect(&a, &AeroL::DataCarrierDetect, &d, &DEMOD::DCDstatSlot);
    d.setAFC(false); d.setSQL(false); d.setCPUReduce(false);
    d.start();
    const char *p = pcm.constData();
    const long nb = pcm.size();
    if (kv.contains("datarate")) // through dataReceived(audio, sampleRate), the ZMQ path: the MSK classes follow the incoming rate
        for (long s = 0; s < nb; s += 2L * chunk) d.dataReceived(QByteArray(p + s, int((nb - s < 2L * chunk) ? nb - s : 2L * chunk)), quint32(getd("datarate", 48000)));
    else
        for (long s = 0; s < nb; s += 2L * chunk) d.write(p + s, (nb - s < 2L * chunk) ? nb - s : 2L * chunk);
    d.stop();
}

int main(int argc, char **argv)
{
    QCoreApplication app(argc, argv);
    if (argc < 5) { fprintf(stderr, "usage: adaptor_demo ref|hip oqpsk|msk in.s16 out.txt [key=value ...]\n"); return 2; }
    for (int i = 5; i < argc; i++) { QString s = argv[i]; int e = s.indexOf('='); if (e > 0) kv[s.left(e)] = s.mid(e + 1); }
    const QString impl = argv[1], kind = argv[2];
    QFile f(argv[3]);
    if (!f.open(QIODevice::ReadOnly)) { fprintf(stderr, "cannot open %s\n", argv[3]); return 2; }
    const QByteArray pcm = f.readAll();
    const int chunk = (int)getd("chunk", 4096);
    AeroLZ *ap = new AeroLZ();
    QBuffer sink;
    sink.open(QIODevice::ReadWrite);
    g_sink = &sink;
    ap->ConnectSinkDevice(&sink);
    QObject::connect(ap, &AeroL::DataCarrierDetect, [&](bool d) { sink.write(QString("#DCD %1\n").arg(d ? 1 : 0).toLatin1()); });
    if (kind == "oqpsk")
    {
        const double fb = getd("fb", 10500);
        ap->setSettings(fb, false);
        if (impl == "ref")
        {
            OqpskDemodulator d(0);
            OqpskDemodulator::Settings s;
            if (kv.contains("prefb"))
            {
                s.fb = getd("prefb", 8400); s.lockingbw = s.fb;
                d.setSettings(s);
                d.start();
                d.write(pcm.constData(), 40000);
                d.stop();
            }
            s.fb = fb; s.lockingbw = getd("lockingbw", fb); s.freq_center = getd("freq_center", 8000);
            d.setSettings(s);
            run(d, *ap, pcm, chunk);
        }
        else
        {
            HipOqpskDemodulator d(0);
            HipOqpskDemodulator::Settings s;
            if (kv.contains("prefb"))
            {
                s.fb = getd("prefb", 8400); s.lockingbw = s.fb;
                d.setSettings(s);
                d.start();
                d.write(pcm.constData(), 40000);
                d.stop();
            }
            s.fb = fb; s.lockingbw = getd("lockingbw", fb); s.freq_center = getd("freq_center", 8000);
            d.setSettings(s);
            run(d, *ap, pcm, chunk);
        }
    }
    else if (kind == "burstoqpsk")
    {
        const double fb = getd("fb", 10500);
        ap->setSettings(fb, true);
        if (impl == "ref")
        {
            BurstOqpskZ *d = new BurstOqpskZ();
            BurstOqpskDemodulator::Settings s;
            s.fb = fb; s.lockingbw = getd("lockingbw", fb); s.freq_center = getd("freq_center", 8000); s.Fs = 48000; s.signalthreshold = 0.6;
            s.coarsefreqest_fft_power = 13; s.channel_stereo = false; s.zmqAudio = false;
            d->setAFC(false); d->setSQL(false); d->setCPUReduce(false);
            d->setScatterPointType(BurstOqpskDemodulator::SPT_None);
            d->setSettings(s);
            g_set_again = [d, s]() mutable { s.lockingbw = getd("set_lockingbw", s.lockingbw); s.freq_center = getd("set_freq_center", s.freq_center); d->setSettings(s); };
            run_burst<BurstOqpskDemodulator>(*d, *ap, pcm, chunk);
        }
        else
        {
            HipBurstOqpskDemodulator d(0);
            HipBurstOqpskDemodulator::Settings s;
            s.fb = fb; s.lockingbw = getd("lockingbw", fb); s.freq_center = getd("freq_center", 8000);
            d.setAFC(false); d.setSQL(false); d.setCPUReduce(false);
            d.setScatterPointType(HipBurstOqpskDemodulator::SPT_None);
            d.setSettings(s);
            g_set_again = [&d, s]() mutable { s.lockingbw = getd("set_lockingbw", s.lockingbw); s.freq_center = getd("set_freq_center", s.freq_center); d.setSettings(s); };
            run_burst(d, *ap, pcm, chunk);
        }
    }
    else if (kind == "burstmsk")
    {
        const double fb = getd("fb", 1200);
        ap->setSettings(fb, true);
        if (impl == "ref")
        {
            BurstMskZ *d = new BurstMskZ();
            BurstMskDemodulator::Settings s;
            s.fb = fb; s.lockingbw = getd("lockingbw", 1.5 * fb); s.freq_center = getd("freq_center", 1000); s.Fs = 48000; s.signalthreshold = 0.6;
            s.coarsefreqest_fft_power = 13; s.symbolspercycle = 16; s.zmqAudio = false;
            d->setAFC(false); d->setSQL(false); d->setCPUReduce(false);
            d->setScatterPointType(BurstMskDemodulator::SPT_None);
            d->DCDstatSlot(false);
            d->setSettings(s);
            QObject::connect(ap, &AeroL::DataCarrierDetect, d, &BurstMskDemodulator::DCDstatSlot); // only the MSK burst class has the input
            g_set_again = [d, s]() mutable { s.fb = getd("set_fb", s.fb); s.lockingbw = getd("set_lockingbw", s.lockingbw); s.freq_center = getd("set_freq_center", s.freq_center); d->setSettings(s); };
            run_burst<BurstMskDemodulator>(*d, *ap, pcm, chunk);
        }
        else
        {
            HipBurstMskDemodulator d(0);
            HipBurstMskDemodulator::Settings s;
            s.fb = fb; s.lockingbw = getd("lockingbw", 1.5 * fb); s.freq_center = getd("freq_center", 1000);
            d.setAFC(false); d.setSQL(false); d.setCPUReduce(false);
            d.setScatterPointType(HipBurstMskDemodulator::SPT_None);
            d.DCDstatSlot(false);
            d.setSettings(s);
            QObject::connect(ap, &AeroL::DataCarrierDetect, &d, &HipBurstMskDemodulator::DCDstatSlot);
            g_set_again = [&d, s]() mutable { s.fb = getd("set_fb", s.fb); s.lockingbw = getd("set_lockingbw", s.lockingbw); s.freq_center = getd("set_freq_center", s.freq_center); d.setSettings(s); };
            run_burst(d, *ap, pcm, chunk);
        }
    }
    else
    {
        const double fb = getd("fb", 1200);
        ap->setSettings(fb, false);
        if (impl == "ref")
        {
            MskDemodulator d(0);
            MskDemodulator::Settings s;
            s.fb = fb; s.lockingbw = getd("lockingbw", 1.5 * fb); s.freq_center = getd("freq_center", 1000);
            d.setSettings(s);
            run(d, *ap, pcm, chunk);
        }
        else
        {
            HipMskDemodulator d(0);
            HipMskDemodulator::Settings s;
            s.fb = fb; s.lockingbw = getd("lockingbw", 1.5 * fb); s.freq_center = getd("freq_center", 1000);
            d.setSettings(s);
            run(d, *ap, pcm, chunk);
        }
    }
    QFile o(argv[4]);
    if (!o.open(QIODevice::WriteOnly)) { fprintf(stderr, "cannot write %s\n", argv[4]); return 2; }
    o.write(sink.data());
    return 0;
}
