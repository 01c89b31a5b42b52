// adaptor_demo -- TEST DRIVER.  Wires a demodulator to the UNMODIFIED AeroL exactly as MainWindow does
// (JAERO/mainwindow.cpp:198-202,234-237) and writes what AeroL prints to its console device:
//
//   adaptor_demo ref|hip oqpsk|msk <in.s16> <out.txt> [fb=10500] [lockingbw=..] [freq_center=..] [chunk=4096] [prefb=..]
//
// prefb: setSettings is first called with that bit rate, 20 000 samples of the input are written, then setSettings with the real one
// (a user changing the rate in the settings dialog): the rest must decode as if nothing had happened before.
//
// "ref" = the reference's own OqpskDemodulator / MskDemodulator; "hip" = HipOqpskDemodulator / HipMskDemodulator
// (integration/qt/hipdemodulator.h over libjaero_hip.so).  tests/test_qt_adaptor.py requires the two outputs to be equal.
// Built by `make -C oracle adaptor` into oracle/_ref/ (it links the reference's objects, which never enter this repository).
#include <QCoreApplication>
#include <QBuffer>
#include <QFile>
#include <QMap>
#include <cstdio>
#include <cstdlib>
#include "oqpskdemodulator.h"
#include "mskdemodulator.h"
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

template <class DEMOD>
static void run(DEMOD &d, AeroL &a, const QByteArray &pcm, int chunk)
{
    QObject::connect(&d, &DEMOD::processDemodulatedSoftBits, &a, &AeroL::processDemodulatedSoftBits);
    QObject::connect(&a, &AeroL::DataCarrierDetect, &d, &DEMOD::DCDstatSlot);
    d.setAFC(false); d.setSQL(false); d.setCPUReduce(false);
    d.start();
    const char *p = pcm.constData();
    const long nb = pcm.size();
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
