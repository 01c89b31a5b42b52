// Check of jd_atan2 / jd_hypot (jaero_amd/csrc/jd_libm.h, compiled here for the host) against the host libm (glibc 2.35) and against
// __float128:
//   g++ -O2 -ffp-contract=off -fopenmp -DHYPOT -x c++ scripts/atan2_check.c -o /tmp/atan2_check -lm -lquadmath
//   /tmp/atan2_check [millions of calls, default 320] [millions per family also against __float128, default 20]
// The hardware hooks are replaced by deliberately WORSE stand-ins (a reciprocal seed good to 13 bits only) so that what is checked
// does not depend on the accuracy of v_rcp_f64.  Prints, per operand family: calls, differences from libm, and on the __float128
// sample: results that are not the correctly rounded value (a few in 1e7: the exit status asks for fewer than 1e-5), how often LIBM's is not (glibc 2.35's atan2
// is within 0.503 ulp, not correctly rounded: ~1.0e-3 of its results are the other neighbour -- and which ones depends on whether the
// ifunc picked its fma build, so "the host's bits" is not a function of the arguments alone; jd_atan2 returns the correctly rounded
// value in all but ~2e-7 of the calls and therefore differs from libm exactly there), how often Ziv's rounding test at 2^-64 would ask for
// a second stage ("stage 2": what a correctly rounded version would have to pay for, see jd_atan2.h), and the largest error of the
// unrounded (hi + lo) in units of 2^-64 of the result.  hypot: bit-identical to libm, differences must be 0.
// Operand families are what the loops produce: unit-circle oscillator value times a resonator output pair of any scale
// (the symbol timing error, JAERO/oqpskdemodulator.cpp:480-484), plus uniform angles at random radii, near-axis and near-diagonal
// angles, exact table points, and wide exponent gaps.
#include <math.h>
#include <quadmath.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define JDA_HOST_CHECK 1
#define JDA_FN static inline
#define JDA_SQRT(x) sqrt(x)
#define JDA_FMA(a, b, c) fma(a, b, c)
#define JDA_LIB_ATAN2(y, x) atan2(y, x)
#define JDA_LIB_HYPOT(x, y) hypot(x, y)
static inline uint32_t jda_hi32(double x) { uint64_t u; memcpy(&u, &x, 8); return (uint32_t)(u >> 32); }
static inline uint32_t jda_lo32(double x) { uint64_t u; memcpy(&u, &x, 8); return (uint32_t)u; }
static inline double jda_words(uint32_t h, uint32_t l) { uint64_t u = ((uint64_t)h << 32) | l; double x; memcpy(&x, &u, 8); return x; }
// reciprocal seed: the exact reciprocal with all but 13 mantissa bits cleared
static inline double jda_bad_rcp(double x)
{
    double r = 1.0 / x;
    uint64_t u; memcpy(&u, &r, 8); u &= ~((1ull << 39) - 1); memcpy(&r, &u, 8);
    return r;
}
#define JDA_RCP(x) jda_bad_rcp(x)
struct JdAtanLane;
static inline void jda_fetch(const JdAtanLane &T, int i, double &A_hi, double &A_lo);
#include "../jaero_amd/csrc/jd_libm.h"
static inline void jda_fetch(const JdAtanLane &, int i, double &A_hi, double &A_lo) { A_hi = JD_ATAN_HI[i]; A_lo = (double)JD_ATAN_LOF[i]; }

static inline uint64_t rng(uint64_t *s) { uint64_t x = *s; x ^= x << 13; x ^= x >> 7; x ^= x << 17; *s = x; return x; }
static inline double u01(uint64_t *s) { return (double)(rng(s) >> 11) * 0x1p-53; }
static inline int same(double a, double b) { uint64_t u, v; memcpy(&u, &a, 8); memcpy(&v, &b, 8); return u == v || (a != a && b != b); }

// stage 1 again, keeping the unrounded pair, to measure its error against __float128 (must mirror jd_atan2; asserted below)
static double stage1_pair(double y, double x, double *Rp, double *lop, int *okp)
{
    JdAtanLane T;
    const double ax = fabs(x), ay = fabs(y);
    const bool sw = ay > ax;
    const double mx = sw ? ay : ax, mn = sw ? ax : ay;
    const bool xneg = x < 0;
    const double u0 = mn * JDA_RCP(mx);
    const double k = fma(u0, 64.0, 0x1.8p52);
    const int i = (int)(jda_lo32(k) & 0x7fu);
    const double c = (k - 0x1.8p52) * 0.015625;
    double A_hi, A_lo;
    jda_fetch(T, i, A_hi, A_lo);
    const double ph = c * mx, pl = fma(c, mx, -ph);
    const double s = mn - ph;
    const double n_hi = s - pl, n_lo = (s - n_hi) - pl;
    const double qh = c * mn, ql = fma(c, mn, -qh);
    const double d_hi = mx + qh, d_lo = (qh - (d_hi - mx)) + ql;
    double rd = JDA_RCP(d_hi);
    rd = fma(fma(-d_hi, rd, 1.0), rd, rd);
    rd = fma(fma(-d_hi, rd, 1.0), rd, rd);
    const double t_hi = n_hi * rd;
    const double e = fma(-t_hi, d_lo, fma(-d_hi, t_hi, n_hi) + n_lo);
    const double t_lo = e * rd;
    const double t2 = t_hi * t_hi;
    const double P = fma(t2, fma(t2, fma(t2, 0x1.c71c71c71c71cp-4, -0x1.2492492492492p-3), 0x1.999999999999ap-3), -0x1.5555555555555p-2);
    double lo = fma(t_hi * t2, P, t_lo) + A_lo;
    const double s1 = A_hi + t_hi;
    lo += t_hi - (s1 - A_hi);
    const double m = sw ? 1.0 : (xneg ? 2.0 : 0.0);
    const double sigma = (sw != xneg) ? -1.0 : 1.0;
    const double K_hi = m * JDA_PIO2_HI, hs = sigma * s1;
    const double R = K_hi + hs;
    const double lo2 = fma(sigma, lo, m * JDA_PIO2_LO) + (hs - (R - K_hi));
    const double err = 0x1p-64 * R;
    *Rp = R; *lop = lo2; *okp = (R + (lo2 - err)) == (R + (lo2 + err)); // Ziv's test at 2^-64: how often a second stage would run
    return R + lo2;
}

struct Stat { long n, diff, stage2, nq, notcr, libnotcr; double maxerr; };

static void one(double y, double x, Stat *st, int quad)
{
    JdAtanLane T;
    const double want = atan2(y, x);
    const double got = jd_atan2(y, x, T);
    st->n++;
    if (!same(want, got))
    {
        st->diff++;
    }
    if (quad)
    {
        double R, lo; int ok;
        stage1_pair(y, x, &R, &lo, &ok);
        if (!ok) st->stage2++;
        const __float128 ex_s = atan2q((__float128)y, (__float128)x);
        const __float128 exact = fabsq(ex_s);
        const double cr = (double)ex_s;
        st->nq++;
        if (!same(cr, got)) st->notcr++;
        if (!same(cr, want)) st->libnotcr++;
        const __float128 ours = (__float128)R + (__float128)lo;
        const double rel = (double)(fabsq(ours - exact) / exact) * 0x1p64;
        if (rel > st->maxerr) st->maxerr = rel;
    }
}

int main(int argc, char **argv)
{
    const long M = (argc > 1 ? atol(argv[1]) : 320) * 1000000L;
    const long Q = argc > 2 ? atol(argv[2]) * 1000000L : 20000000L; // of each family, this many also against __float128
    const char *names[] = {"osc x resonator pair", "uniform angle, radius 2^[-40,40]", "near axes / diagonals", "table points +- few ulp", "wide exponent gap",
                           "specials"};
    const int NF = 6;
    long total_diff = 0, total_notcr = 0, total_q = 0;
    for (int fam = 0; fam < NF; fam++)
    {
        const long n = fam == 0 ? M / 2 : fam == 5 ? 1 : M / 8;
        Stat tot = {0, 0, 0, 0, 0, 0, 0.0};
#pragma omp parallel
        {
            Stat st = {0, 0, 0, 0, 0, 0, 0.0};
#pragma omp for schedule(static)
            for (long b = 0; b < 1024; b++)
            {
                uint64_t s = 0x9e3779b97f4a7c15ull * (uint64_t)(b + 1 + 4096 * fam) + 12345;
                for (int w = 0; w < 8; w++) rng(&s);
                const long per = n / 1024 + 1;
                for (long j = 0; j < per; j++)
                {
                    double x, y;
                    const int quad = j < Q / 1024;
                    if (fam == 0)
                    {
                        // (so.x + i so.y) * (m_re + i m_im): so on the unit circle (table value), m = resonator outputs, log-uniform scale
                        const double ph = 2 * M_PI * u01(&s), sc = exp2(80 * u01(&s) - 60);
                        const double cx = cos(ph), cy = sin(ph);
                        const double mre = sc * (2 * u01(&s) - 1), mim = sc * (2 * u01(&s) - 1) * exp2(-6 * u01(&s));
                        x = cx * mre - cy * mim; y = cx * mim + cy * mre;
                    }
                    else if (fam == 1)
                    {
                        const double ph = 2 * M_PI * u01(&s), r = exp2(80 * u01(&s) - 40);
                        x = r * cos(ph); y = r * sin(ph);
                    }
                    else if (fam == 2)
                    {
                        const double base = (double)(rng(&s) & 7) * (M_PI / 4), d = exp2(-60 * u01(&s)) * (u01(&s) - 0.5);
                        const double r = exp2(20 * u01(&s) - 10);
                        x = r * cos(base + d); y = r * sin(base + d);
                    }
                    else if (fam == 3)
                    {
                        const int i = (int)(rng(&s) % 65), q = (int)(rng(&s) & 7);
                        double mx = 1.0 + u01(&s), mn = mx * ((double)i / 64.0 + (i & 1 ? 1 : -1) * (1.0 / 128.0) * (double)(rng(&s) & 1));
                        mn = mn * (1.0 + ((double)(int)(rng(&s) & 15) - 8) * 0x1p-52);
                        if (mn <= 0) mn = 0x1p-30 * u01(&s) + 0x1p-200;
                        x = (q & 1) ? mn : mx; y = (q & 1) ? mx : mn;
                        if (q & 2) x = -x;
                        if (q & 4) y = -y;
                    }
                    else if (fam == 4)
                    {
                        const double a = exp2(600 * u01(&s) - 300) * (1 + u01(&s)), b2 = exp2(600 * u01(&s) - 300) * (1 + u01(&s));
                        x = (rng(&s) & 1) ? a : -a; y = (rng(&s) & 1) ? b2 : -b2;
                    }
                    else
                    {
                        const double sp[] = {0.0, -0.0, 1.0, -1.0, INFINITY, -INFINITY, NAN, 0x1p-1074, -0x1p-1074, 0x1p1023, -0x1p1023, 0x1p-1022, 1e-310, 3.0};
                        for (unsigned a = 0; a < sizeof sp / 8; a++)
                            for (unsigned c = 0; c < sizeof sp / 8; c++) one(sp[a], sp[c], &st, 0);
                        break;
                    }
                    one(y, x, &st, quad);
                }
            }
#pragma omp critical
            {
                tot.n += st.n; tot.diff += st.diff; tot.stage2 += st.stage2; tot.nq += st.nq; tot.notcr += st.notcr; tot.libnotcr += st.libnotcr;
                if (st.maxerr > tot.maxerr) tot.maxerr = st.maxerr;
            }
        }
        printf("%-34s calls %10ld differ from libm %7ld (%.2e) | __float128 sample %9ld: not correctly rounded here %ld, libm %ld (%.2e); stage 2 in %ld (%.2e), max "
               "stage-1 error %.3f x 2^-64\n",
               names[fam], tot.n, tot.diff, (double)tot.diff / (double)tot.n, tot.nq, tot.notcr, tot.libnotcr, (double)tot.libnotcr / (double)(tot.nq ? tot.nq : 1), tot.stage2,
               (double)tot.stage2 / (double)(tot.nq ? tot.nq : 1), tot.maxerr);
        total_notcr += tot.notcr; total_q += tot.nq;
    }
#ifdef HYPOT
    {
        long n = 0, diff = 0;
#pragma omp parallel for reduction(+ : n, diff)
        for (long b = 0; b < 1024; b++)
        {
            uint64_t s = 0xda942042e4dd58b5ull * (uint64_t)(b + 1) + 99;
            for (int w = 0; w < 8; w++) rng(&s);
            for (long j = 0; j < M / 1024; j++)
            {
                const int mode = (int)(rng(&s) & 3);
                double x, y;
                if (mode == 0) { x = 4 * u01(&s) - 2; y = 4 * u01(&s) - 2; }
                else if (mode == 1) { const double sc = exp2(120 * u01(&s) - 80); x = sc * (2 * u01(&s) - 1); y = sc * (2 * u01(&s) - 1) * exp2(-50 * u01(&s)); }
                else if (mode == 2) { const double ph = 2 * M_PI * u01(&s), r = exp2(8 * u01(&s) - 6); x = r * cos(ph); y = r * sin(ph); }
                else { x = (double)(int)(rng(&s) % 2001 - 1000) * 0x1p-9; y = (double)(int)(rng(&s) % 2001 - 1000) * 0x1p-9; }
                const double want = hypot(x, y), got = jd_hypot(x, y);
                n++;
                if (!same(want, got)) { diff++; if (diff <= 3) fprintf(stderr, "  DIFF hypot(%a, %a): libm %a here %a\n", x, y, want, got); }
            }
        }
        const double sp[] = {0.0, -0.0, 1.0, INFINITY, -INFINITY, NAN, 0x1p-1074, 0x1p1023, 0x1p-1022, 0x1p600, 0x1p-600, 3.0};
        for (unsigned a = 0; a < sizeof sp / 8; a++)
            for (unsigned c = 0; c < sizeof sp / 8; c++) { n++; if (!same(hypot(sp[a], sp[c]), jd_hypot(sp[a], sp[c]))) diff++; }
        printf("%-36s calls %11ld  differ from libm %ld\n", "hypot", n, diff);
        total_diff += diff;
    }
#endif
    const double rate = (double)total_notcr / (double)(total_q ? total_q : 1);
    printf("atan2 results that are not the correctly rounded value: %ld of %ld (%.2e); hypot differences from libm: %ld\n", total_notcr, total_q, rate, total_diff);
    return total_diff != 0 || rate > 1e-5;
}
