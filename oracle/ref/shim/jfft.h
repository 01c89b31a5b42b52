// TEST INFRASTRUCTURE ONLY (oracle/_ref build) -- never linked into the product library.
//
// Stand-in for jontio/JFFT, which JAERO links but does not vendor (cloned at HEAD, unpinned:
// ci-linux-build.sh:152-162, JAERO.pro:24,28).  Written from scratch against the API surface the
// reference actually calls:
//   JFFT::init / fft / ifft / fft_real / ifft_real      (JAERO/fftwrapper.cpp:19-35, JAERO/fftrwrapper.cpp:19-39)
//   JFastFir::SetKernel(x3) / update                    (JAERO/oqpskdemodulator.cpp:114,283,368, JAERO/DSP.cpp:788,
//                                                        JAERO/burstoqpskdemodulator.cpp:344)
// Conventions are pinned by the reference's own golden-vector tests (see tests/test_oracle_fft_golden.py):
//   * fft  : unnormalised forward DFT, exp(-j2pi nk/N)          (JAERO/tests/fftwrapper_tests.cpp:27-29)
//   * ifft : inverse DFT scaled by 1/N (FFTWrapper multiplies N back, JAERO/fftwrapper.cpp:27-33)
//   * fft_real / ifft_real : full-length output / Hermitian completion from the lower half
//                                                                 (JAERO/tests/fftrwrapper_tests.cpp:28-30)
//   * JFastFir : block fast convolution whose output equals the causal convolution delayed by
//     L = nfft - K + 1 samples (JAERO/tests/jfastfir_tests.cpp + data files).
//     The default nfft of the 1-argument SetKernel is UNPINNED (inferred "x4 rule of thumb",
//     JAERO/oqpskdemodulator.cpp:283): nfft = 4 * 2^ceil(log2 K).
#ifndef ORACLE_SHIM_JFFT_H
#define ORACLE_SHIM_JFFT_H

#include <QVector>
#include <complex>
#include <vector>
#include <cmath>
#include <cassert>

class JFFT
{
public:
    typedef std::complex<double> cpx_type;
    JFFT() : nfft(0) {}
    void init(int n)
    {
        nfft = n;
        assert(n > 0 && (n & (n - 1)) == 0);
        tw.resize(n / 2 > 0 ? n / 2 : 1);
        for (int i = 0; i < n / 2; i++)
        {
            double a = -2.0 * M_PI * ((double)i) / ((double)n);
            tw[i] = cpx_type(cos(a), sin(a));
        }
        rev.resize(n);
        int bits = 0;
        while ((1 << bits) < n) bits++;
        for (int i = 0; i < n; i++)
        {
            int r = 0;
            for (int b = 0; b < bits; b++) if (i & (1 << b)) r |= 1 << (bits - 1 - b);
            rev[i] = r;
        }
    }
    void fft(QVector<cpx_type> &x) { assert(x.size() == nfft); run(x.data(), false); }
    void ifft(QVector<cpx_type> &x)
    {
        assert(x.size() == nfft);
        run(x.data(), true);
        double s = 1.0 / ((double)nfft);
        for (int i = 0; i < nfft; i++) x[i] *= s;
    }
    void fft(std::vector<cpx_type> &x) { assert((int)x.size() == nfft); run(x.data(), false); }
    void ifft(std::vector<cpx_type> &x)
    {
        assert((int)x.size() == nfft);
        run(x.data(), true);
        double s = 1.0 / ((double)nfft);
        for (int i = 0; i < nfft; i++) x[i] *= s;
    }
    void fft_real(const QVector<double> &in, QVector<cpx_type> &out)
    {
        assert(in.size() == nfft);
        out.resize(nfft);
        for (int i = 0; i < nfft; i++) out[i] = cpx_type(in[i], 0.0);
        run(out.data(), false);
    }
    void ifft_real(const QVector<cpx_type> &in, QVector<double> &out)
    {
        assert(in.size() == nfft);
        std::vector<cpx_type> t(nfft);
        for (int k = 0; k <= nfft / 2; k++) t[k] = in[k];
        for (int k = 1; k < nfft / 2; k++) t[nfft - k] = std::conj(in[k]);
        t[0] = cpx_type(t[0].real(), 0.0);
        if (nfft > 1) t[nfft / 2] = cpx_type(t[nfft / 2].real(), 0.0);
        run(t.data(), true);
        out.resize(nfft);
        double s = 1.0 / ((double)nfft);
        for (int i = 0; i < nfft; i++) out[i] = t[i].real() * s;
    }
    int size() const { return nfft; }
private:
    void run(cpx_type *x, bool inverse)
    {
        const int n = nfft;
        for (int i = 0; i < n; i++) { int r = rev[i]; if (r > i) std::swap(x[i], x[r]); }
        for (int len = 2; len <= n; len <<= 1)
        {
            int half = len >> 1;
            int step = n / len;
            for (int base = 0; base < n; base += len)
            {
                for (int j = 0; j < half; j++)
                {
                    cpx_type w = tw[j * step];
                    if (inverse) w = std::conj(w);
                    cpx_type a = x[base + j];
                    cpx_type b = x[base + j + half];
                    cpx_type t(b.real() * w.real() - b.imag() * w.imag(), b.real() * w.imag() + b.imag() * w.real());
                    x[base + j] = a + t;
                    x[base + j + half] = a - t;
                }
            }
        }
    }
    int nfft;
    std::vector<cpx_type> tw;
    std::vector<int> rev;
};

class JFastFir
{
public:
    typedef std::complex<double> cpx_type;
    JFastFir() : nfft(0), K(0), L(0) {}
    void SetKernel(const QVector<cpx_type> &k)
    {
        int p = 1;
        while (p < k.size()) p <<= 1;
        SetKernel(k, 4 * p);
    }
    void SetKernel(const QVector<double> &k)
    {
        QVector<cpx_type> c(k.size());
        for (int i = 0; i < k.size(); i++) c[i] = cpx_type(k[i], 0.0);
        SetKernel(c);
    }
    void SetKernel(const QVector<double> &k, int _nfft)
    {
        QVector<cpx_type> c(k.size());
        for (int i = 0; i < k.size(); i++) c[i] = cpx_type(k[i], 0.0);
        SetKernel(c, _nfft);
    }
    void SetKernel(const QVector<cpx_type> &k, int _nfft)
    {
        K = k.size();
        nfft = _nfft;
        assert(nfft >= K);
        L = nfft - K + 1;
        fft.init(nfft);
        H.assign(nfft, cpx_type(0, 0));
        for (int i = 0; i < K; i++) H[i] = k[i];
        fft.fft(H);
        inbuf.clear();
        tail.assign(nfft, cpx_type(0, 0));
        outq.assign(L, cpx_type(0, 0)); // L samples of latency
        outq_rd = 0;
    }
    // in-place streaming filter: y[n] = (h * x)[n - L]
    void update(QVector<cpx_type> &data)
    {
        if (!nfft) return;
        for (int i = 0; i < data.size(); i++)
        {
            inbuf.push_back(data[i]);
            if ((int)inbuf.size() == L) processblock();
            data[i] = outq[outq_rd++];
        }
        if (outq_rd > 0)
        {
            outq.erase(outq.begin(), outq.begin() + outq_rd);
            outq_rd = 0;
        }
    }
private:
    void processblock()
    {
        std::vector<cpx_type> blk(nfft, cpx_type(0, 0));
        for (int i = 0; i < L; i++) blk[i] = inbuf[i];
        inbuf.clear();
        fft.fft(blk);
        for (int i = 0; i < nfft; i++) blk[i] *= H[i];
        fft.ifft(blk);
        // overlap-add
        for (int i = 0; i < nfft; i++) tail[i] += blk[i];
        for (int i = 0; i < L; i++) outq.push_back(tail[i]);
        for (int i = 0; i + L < nfft; i++) tail[i] = tail[i + L];
        for (int i = nfft - L; i < nfft; i++) tail[i] = cpx_type(0, 0);
    }
    JFFT fft;
    int nfft, K, L;
    std::vector<cpx_type> H, inbuf, tail, outq;
    size_t outq_rd;
};

#endif
