// hipdemodulator.h -- QIODevice adaptors that put libjaero_hip.so under JAERO's existing wiring.
//
// HipOqpskDemodulator / HipMskDemodulator have the member functions, slots and signals MainWindow uses on
// OqpskDemodulator / MskDemodulator (JAERO/oqpskdemodulator.h:15-152, JAERO/mskdemodulator.h:17-160), so the connect() lines of
// JAERO/mainwindow.cpp:198-202,234-237 keep working: each object is a one-channel bank of include/jaero_hip.h.
// HipBurstOqpskDemodulator / HipBurstMskDemodulator do the same for BurstOqpskDemodulator / BurstMskDemodulator
// (JAERO/burstoqpskdemodulator.h:15-232, JAERO/burstmskdemodulator.h:22-220): soft bits in the reference's groups with the -1
// start-of-burst marker, SignalStatus / EbNoMeasurmentSignal / Plottables from the bank's event log.
// Add to JAERO.pro: HEADERS += hipdemodulator.h, INCLUDEPATH += <repo>/include, LIBS += -L<repo>/jaero_amd -ljaero_hip.
// Compiled (moc + g++ against Qt 5.9.7) and driven by the unmodified AeroL in integration/qt/adaptor_demo.cpp /
// tests/test_qt_adaptor.py.
#pragma once
#include <QIODevice>
#include <QVector>
#include <QByteArray>
#include <vector>
extern "C" {
#include "jaero_hip.h"
}

class HipDemodulatorBase : public QIODevice
{
    Q_OBJECT
public:
    explicit HipDemodulatorBase(QObject *parent, int kind_, int group_) : QIODevice(parent), kind(kind_), group(group_) {}
    ~HipDemodulatorBase() override { if (ctx) jaero_destroy(ctx); }
    void setAFC(bool v) { afc = v; pushFlags(); }             // oqpskdemodulator.cpp:149-152
    void setSQL(bool v) { sql = v; pushFlags(); }             // :154-157
    void setCPUReduce(bool v) { cpuReduce = v; pushFlags(); } // :159-163
    void start() { open(QIODevice::WriteOnly); }              // :312-315
    void stop() { close(); }
    double getCurrentFreq()
    {
        jaero_status st;
        return (ctx && jaero_read_status(ctx, 0, &st) == JAERO_OK) ? st.freq_center : freq_center;
    }
    qint64 readData(char *, qint64) override { return 0; }
    // = writeData of the reference (oqpskdemodulator.cpp:334-627, mskdemodulator.cpp:313-488): len bytes of little-endian int16 mono
    qint64 writeData(const char *data, qint64 len) override
    {
        if (!ctx || len < 2) return len;
        const int16_t *pcm = reinterpret_cast<const int16_t *>(data);
        qint64 n = len / 2;
        for (qint64 s = 0; s < n; s += maxWrite)
        {
            const int m = int(n - s < maxWrite ? n - s : maxWrite);
            if (jaero_write(ctx, pcm + s, m, JAERO_PCM_CHANNEL_MAJOR, /*host pointer*/ 0, nullptr) != JAERO_OK)
            {
                emit WarningTextSignal(QString("libjaero_hip: %1").arg(jaero_last_error()));
                return len;
            }
            drain();
        }
        return len;
    }
signals:
    void processDemodulatedSoftBits(const QVector<short> &soft_bits);
    void Plottables(double freq_est, double freq_center, double bandwidth);
    void MSESignal(double mse);
    void SignalStatus(bool gotasignal);
    void EbNoMeasurmentSignal(double EbNo);
    void SampleRateChanged(double Fs);
    void BitRateChanged(double fb, bool burstmode);
    void WarningTextSignal(QString str);
public slots:
    void CenterFreqChangedSlot(double f) { if (ctx) jaero_center_freq_changed(ctx, 0, f); }          // :291-310
    void DCDstatSlot(bool d) { dcd = d; if (ctx) jaero_set_dcd(ctx, -1, d); }                       // :679-684
    // OqpskDemodulator::dataReceived only warns about another rate (:686-693); MskDemodulator::dataReceived re-applies its last settings with
    // the incoming rate (mskdemodulator.cpp:528-537) -- here: jaero_set_settings with that rate, which carries the demodulator's state into
    // a bank at the new rate behind the same handle (applySettings)
    void dataReceived(const QByteArray &audio, quint32 sampleRate)
    {
        if (double(sampleRate) != Fs && kind == JAERO_KIND_MSK && ctx)
        {
            jaero_settings js = cur;
            js.Fs = double(sampleRate);
            applySettings(js);
        }
        writeData(audio.constData(), audio.length());
    }
protected:
    void applySettings(const jaero_settings &js)
    {
        if (!ctx)
        {
            if (jaero_create(0, 1, &js, 0, JAERO_FLAG_EBNO | JAERO_FLAG_STATUS_LOG, maxWrite, 0, &ctx) != JAERO_OK)
            {
                emit WarningTextSignal(QString("libjaero_hip: %1").arg(jaero_last_error()));
                ctx = nullptr;
                return;
            }
            pushFlags();
            jaero_set_dcd(ctx, -1, dcd);
        }
        else
        {
            // A change of rate / sample rate / FFT size (and any setSettings at 8400 bps) re-creates the one-channel bank behind the handle
            // with what the reference's setSettings keeps in the old object -- oscillator phases, loop states, symbol-rate windows, coarse
            // ring and spectrum (jaero_hip.h).  Soft bits that did not fill a group yet stay in `pending`, as RxDataBits survives there.
            const int rc = jaero_set_settings(ctx, 0, &js);
            if (rc == JAERO_EINVAL)
            {
                // bad settings: nothing changed, the demodulator keeps running as it was (the reference has no such case)
                emit WarningTextSignal(QString("libjaero_hip: %1").arg(jaero_last_error()));
                return;
            }
            if (rc != JAERO_OK)
            {
                // the carry-over itself failed (no memory for the sibling bank, unread outputs larger than the new buffers, a bank whose
                // write failed earlier): audio at the new rate must not run into the old-rate bank, and retrying the carry-over on every
                // call would cost a create / destroy each time -- start afresh at the new settings, as the adaptors did before round 3
                emit WarningTextSignal(QString("libjaero_hip: %1; restarting the demodulator at the new settings").arg(jaero_last_error()));
                jaero_destroy(ctx);
                ctx = nullptr;
                if (jaero_create(0, 1, &js, 0, JAERO_FLAG_EBNO | JAERO_FLAG_STATUS_LOG, maxWrite, 0, &ctx) != JAERO_OK)
                {
                    emit WarningTextSignal(QString("libjaero_hip: %1").arg(jaero_last_error()));
                    ctx = nullptr;
                    return;
                }
                pushFlags();
                jaero_set_dcd(ctx, -1, dcd);
            }
        }
        cur = js;
        if (js.Fs != Fs) { Fs = js.Fs; emit SampleRateChanged(Fs); }
        if (js.fb != fb) { fb = js.fb; emit BitRateChanged(fb, false); }
        lockingbw = js.lockingbw;
        freq_center = js.freq_center;
    }
private:
    jaero_settings cur{};
    void pushFlags() { if (ctx) jaero_set_flags(ctx, -1, afc, sql, cpuReduce); }
    void drain()
    {
        // soft bits in the reference's groups (32 for OQPSK :583-591, 12 for MSK mskdemodulator.cpp:472-477); what does not fill a
        // group yet waits for the next write, as RxDataBits does
        int n = 0;
        buf.resize(1 << 16);
        if (jaero_read_softbits(ctx, 0, buf.data(), int(buf.size()), &n) != JAERO_OK) n = 0;
        for (int i = 0; i < n; i++)
        {
            pending.push_back(buf[i]);
            if (pending.size() >= group) { emit processDemodulatedSoftBits(pending); pending.clear(); }
        }
        // one status row per FreqOffsetEstimateSlot call: [n, freq_est, freq_center, mse, ebno, signal] (:670-675)
        double rows[64 * 6];
        int nr = 0;
        if (jaero_read_status_log(ctx, 0, rows, 64, &nr) != JAERO_OK) nr = 0;
        for (int r = 0; r < nr; r++)
        {
            const double *q = rows + 6 * r;
            emit Plottables(q[1], q[2], lockingbw);
            emit EbNoMeasurmentSignal(q[4]);
            emit MSESignal(q[3]);
            emit SignalStatus(q[5] != 0);
        }
    }
    jaero_ctx *ctx = nullptr;
    const int kind, group;
    const int maxWrite = 1 << 16;
    bool afc = false, sql = false, cpuReduce = false, dcd = false;
    double Fs = 0, fb = 0, lockingbw = 0, freq_center = 0;
    QVector<short> pending;
    std::vector<int16_t> buf;
};

class HipOqpskDemodulator : public HipDemodulatorBase
{
    Q_OBJECT
public:
    struct Settings // == OqpskDemodulator::Settings (JAERO/oqpskdemodulator.h:20-39)
    {
        int coarsefreqest_fft_power = 14;
        double freq_center = 8000, lockingbw = 10500, fb = 10500, Fs = 48000, signalthreshold = 0.65;
        bool zmqAudio = false;
    };
    explicit HipOqpskDemodulator(QObject *parent = nullptr) : HipDemodulatorBase(parent, JAERO_KIND_OQPSK, 32) {}
    void setSettings(Settings s) // oqpskdemodulator.cpp:175-289
    {
        jaero_settings js{JAERO_KIND_OQPSK, s.coarsefreqest_fft_power, s.freq_center, s.lockingbw, s.fb, s.Fs, s.signalthreshold};
        applySettings(js);
    }
};

class HipMskDemodulator : public HipDemodulatorBase
{
    Q_OBJECT
public:
    struct Settings // == MskDemodulator::Settings (JAERO/mskdemodulator.h:24-45)
    {
        int coarsefreqest_fft_power = 13;
        double freq_center = 1000, lockingbw = 900, fb = 600, Fs = 48000, signalthreshold = 0.5;
        bool zmqAudio = false;
    };
    explicit HipMskDemodulator(QObject *parent = nullptr) : HipDemodulatorBase(parent, JAERO_KIND_MSK, 12) {}
    void setSettings(Settings s) // mskdemodulator.cpp:135-263
    {
        jaero_settings js{JAERO_KIND_MSK, s.coarsefreqest_fft_power, s.freq_center, s.lockingbw, s.fb, s.Fs, s.signalthreshold};
        applySettings(js);
    }
};

// ------------------------------------------------------------------------------------------------ burst kinds
class HipBurstDemodulatorBase : public QIODevice
{
    Q_OBJECT
public:
    enum ScatterPointType { SPT_constellation, SPT_phaseoffseterror, SPT_phaseoffsetest, SPT_None };
    explicit HipBurstDemodulatorBase(QObject *parent, int kind_, int group_) : QIODevice(parent), kind(kind_), group(group_) {}
    ~HipBurstDemodulatorBase() override { if (ctx) jaero_destroy(ctx); }
    void setAFC(bool v) { afc = v; pushFlags(); }
    void setSQL(bool v) { sql = v; pushFlags(); }
    void setCPUReduce(bool v) { cpuReduce = v; pushFlags(); }
    void setScatterPointType(ScatterPointType) {} // GUI only: the library produces no scatter points
    void invalidatesettings() { Fs = -1; fb = -1; }
    void start() { open(QIODevice::WriteOnly); }
    void stop() { close(); }
    double getCurrentFreq() { return freq_est; }
    qint64 readData(char *, qint64) override { return 0; }
    // = writeData / writeDataSlot of the reference (burstoqpskdemodulator.cpp:300-737, burstmskdemodulator.cpp:371-754), mono int16
    qint64 writeData(const char *data, qint64 len) override
    {
        if (!ctx || len < 2) return len;
        const int16_t *pcm = reinterpret_cast<const int16_t *>(data);
        const qint64 n = len / 2;
        for (qint64 s = 0; s < n; s += maxWrite)
        {
            const int m = int(n - s < maxWrite ? n - s : maxWrite);
            if (jaero_write(ctx, pcm + s, m, JAERO_PCM_CHANNEL_MAJOR, /*host pointer*/ 0, nullptr) != JAERO_OK)
            {
                emit WarningTextSignal(QString("libjaero_hip: %1").arg(jaero_last_error()));
                return len;
            }
            drain();
        }
        return len;
    }
signals:
    void processDemodulatedSoftBits(const QVector<short> &soft_bits);
    void Plottables(double freq_est, double freq_center, double bandwidth);
    void MSESignal(double mse);
    void SignalStatus(bool gotasignal);
    void EbNoMeasurmentSignal(double EbNo);
    void SampleRateChanged(double Fs);
    void BitRateChanged(double fb, bool burstmode);
    void WarningTextSignal(const QString &str);
public slots:
    void CenterFreqChangedSlot(double f) { if (ctx && jaero_center_freq_changed(ctx, 0, f) == JAERO_OK) freq_center = f; }
    void DCDstatSlot(bool d) { dcd = d; if (ctx) jaero_set_dcd(ctx, -1, d); }
    void dataReceived(const QByteArray &audio, quint32) { writeData(audio.constData(), audio.length()); }
protected:
    void applySettings(const jaero_settings &js)
    {
        // Same bit rate and sample rate: setSettings on the live object (jaero_set_settings on a burst bank, k_burst_settings.h) -- AGCs, EbNo
        // meter, Hilbert filter, peak detector and trident fill restart, the delay lines keep their contents with the pointers at zero,
        // RxDataBits (`pending`) survives, as in the reference.  Burst MSK also changes its bit rate this way (a sibling bank with the survivors
        // behind the same handle); whatever the library refuses is answered by replacing the one-channel bank.
        const int rc = (ctx && js.Fs == Fs) ? jaero_set_settings(ctx, 0, &js) : JAERO_ENOTSUP;
        if (rc == JAERO_EINVAL)
        {
            // settings the library rejects as such (they would fail jaero_create's validation too): warn and keep demodulating with the old
            // bank, as the continuous adaptor does -- replacing the bank would leave this object without one
            emit WarningTextSignal(QString("libjaero_hip: %1").arg(jaero_last_error()));
            return;
        }
        if (rc == JAERO_OK)
        {
            if (kind == JAERO_KIND_BURST_MSK) dcd = false; // burstmskdemodulator.cpp:322
            if (js.fb != fb) { fb = js.fb; emit BitRateChanged(fb, true); } // burst MSK 600 <-> 1200: the bank behind the handle was re-created with the survivors
            lockingbw = js.lockingbw;
            freq_center = js.freq_center;
            drainEvents(); // the Plottables emission at the end of setSettings
            return;
        }
        if (ctx) { jaero_destroy(ctx); ctx = nullptr; }
        if (jaero_create(0, 1, &js, 0, 0, maxWrite, 0, &ctx) != JAERO_OK)
        {
            emit WarningTextSignal(QString("libjaero_hip: %1").arg(jaero_last_error()));
            ctx = nullptr;
            return;
        }
        pushFlags();
        jaero_set_dcd(ctx, -1, dcd);
        pending.clear();
        if (js.Fs != Fs) { Fs = js.Fs; emit SampleRateChanged(Fs); }
        if (js.fb != fb) { fb = js.fb; emit BitRateChanged(fb, true); }
        lockingbw = js.lockingbw;
        freq_center = js.freq_center;
        drainEvents(); // the Plottables emission at the end of setSettings
    }
private:
    void pushFlags() { if (ctx) jaero_set_flags(ctx, -1, afc, sql, cpuReduce); }
    void drainEvents()
    {
        double rows[256 * 3];
        int nr = 0;
        if (jaero_read_events(ctx, 0, rows, 256, &nr) != JAERO_OK) nr = 0;
        for (int r = 0; r < nr; r++)
        {
            const int k = int(rows[3 * r + 1]);
            const double v = rows[3 * r + 2];
            if (k == JAERO_EV_SIGNAL) emit SignalStatus(v != 0);
            else if (k == JAERO_EV_EBNO) emit EbNoMeasurmentSignal(v);
            else if (k == JAERO_EV_FREQ) { freq_est = v; emit Plottables(v, freq_center, lockingbw); }
        }
    }
    void drain()
    {
        // The bank hands over the emitted groups back to back (what has not filled a group yet stays on the device, as RxDataBits
        // does); the groups are cut as the reference cuts them: the start-of-burst marker (-1) is one entry, soft bits come in
        // pairs, a group goes out once it holds >= `group` entries after a pair (burstoqpskdemodulator.cpp:546-585 with 32,
        // burstmskdemodulator.cpp with 12).  AeroL's burst mode depends on these boundaries (it drops the rest of a group at signal end).
        int n = 0;
        buf.resize(1 << 16);
        if (jaero_read_softbits(ctx, 0, buf.data(), int(buf.size()), &n) != JAERO_OK) n = 0;
        int i = 0;
        while (i < n)
        {
            if (buf[i] < 0) { pending.push_back(buf[i++]); continue; }
            pending.push_back(buf[i++]);
            if (i < n) pending.push_back(buf[i++]);
            if (pending.size() >= group) { emit processDemodulatedSoftBits(pending); pending.clear(); }
        }
        drainEvents();
    }
    jaero_ctx *ctx = nullptr;
    const int kind, group;
    const int maxWrite = 1 << 16;
    bool afc = false, sql = false, cpuReduce = false, dcd = false;
    double Fs = 0, fb = 0, lockingbw = 0, freq_center = 0, freq_est = 0;
    QVector<short> pending;
    std::vector<int16_t> buf;
};

class HipBurstOqpskDemodulator : public HipBurstDemodulatorBase
{
    Q_OBJECT
public:
    struct Settings // == BurstOqpskDemodulator::Settings (JAERO/burstoqpskdemodulator.h:24-45)
    {
        int coarsefreqest_fft_power = 13;
        double freq_center = 8000, lockingbw = 10500, fb = 10500, Fs = 48000, signalthreshold = 0.6;
        bool channel_stereo = false, zmqAudio = false;
    };
    explicit HipBurstOqpskDemodulator(QObject *parent = nullptr) : HipBurstDemodulatorBase(parent, JAERO_KIND_BURST_OQPSK, 32) {}
    void setSettings(Settings s) // burstoqpskdemodulator.cpp:202-277
    {
        jaero_settings js{JAERO_KIND_BURST_OQPSK, s.coarsefreqest_fft_power, s.freq_center, s.lockingbw, s.fb, s.Fs, s.signalthreshold};
        applySettings(js);
    }
};

class HipBurstMskDemodulator : public HipBurstDemodulatorBase
{
    Q_OBJECT
public:
    struct Settings // == BurstMskDemodulator::Settings (JAERO/burstmskdemodulator.h:29-50)
    {
        int coarsefreqest_fft_power = 13;
        double freq_center = 1000, lockingbw = 1800, fb = 1200, Fs = 48000;
        int symbolspercycle = 16;
        double signalthreshold = 0.6;
        bool zmqAudio = false;
    };
    explicit HipBurstMskDemodulator(QObject *parent = nullptr) : HipBurstDemodulatorBase(parent, JAERO_KIND_BURST_MSK, 12) {}
    void setSettings(Settings s) // burstmskdemodulator.cpp:150-325
    {
        jaero_settings js{JAERO_KIND_BURST_MSK, s.coarsefreqest_fft_power, s.freq_center, s.lockingbw, s.fb, s.Fs, s.signalthreshold};
        applySettings(js);
    }
};
