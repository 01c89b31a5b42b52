// Bit-identity check of the fdlibm / glibc-2.35 tanh and expm1 restated in jaero_device.h (jd_tanh, jd_expm1) against the host libm:
// gcc -O2 -ffp-contract=off -DGROUPED scripts/tanh_check.c -lm && ./a.out   (12 s; prints 0 mismatches on 2e8 arguments; add -DFAST to check the straight-line form jd_tanh uses for 2^-55 <= |x| < 6.5; without -DGROUPED the
// polynomial is in fdlibm's original Horner form, which glibc 2.35 does not use: ~1e-4 of the results then differ in the last bit)
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static inline uint32_t hi32(double x) { uint64_t u; memcpy(&u, &x, 8); return (uint32_t)(u >> 32); }
static inline uint32_t lo32(double x) { uint64_t u; memcpy(&u, &x, 8); return (uint32_t)u; }
static inline double with_hi(double x, uint32_t h) { uint64_t u; memcpy(&u, &x, 8); u = (u & 0xffffffffull) | ((uint64_t)h << 32); memcpy(&x, &u, 8); return x; }
static inline double from_words(uint32_t h, uint32_t l) { uint64_t u = ((uint64_t)h << 32) | l; double x; memcpy(&x, &u, 8); return x; }

static double my_expm1(double x)
{
    const double one = 1.0, huge = 1.0e+300, tiny = 1.0e-300, o_threshold = 7.09782712893383973096e+02,
                 ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10, invln2 = 1.44269504088896338700e+00,
                 Q1 = -3.33333333333331316428e-02, Q2 = 1.58730158725481460165e-03, Q3 = -7.93650757867487942473e-05,
                 Q4 = 4.00821782732936239552e-06, Q5 = -2.01099218183624371326e-07;
    double y, hi, lo, c = 0, t, e, hxs, hfx, r1, h2, h4, R1, R2, R3;
    int32_t k, xsb;
    uint32_t hx;
    hx = hi32(x);
    xsb = hx & 0x80000000;
    hx &= 0x7fffffff;
    if (hx >= 0x4043687A)
    {
        if (hx >= 0x40862E42)
        {
            if (hx >= 0x7ff00000)
            {
                uint32_t low = lo32(x);
                if (((hx & 0xfffff) | low) != 0) return x + x;
                else return (xsb == 0) ? x : -1.0;
            }
            if (x > o_threshold) return huge * huge;
        }
        if (xsb != 0)
        {
            if (x + tiny < 0.0) return tiny - one;
        }
    }
    if (hx > 0x3fd62e42)
    {
        if (hx < 0x3FF0A2B2)
        {
            if (xsb == 0) { hi = x - ln2_hi; lo = ln2_lo; k = 1; }
            else { hi = x + ln2_hi; lo = -ln2_lo; k = -1; }
        }
        else
        {
            k = invln2 * x + ((xsb == 0) ? 0.5 : -0.5);
            t = k;
            hi = x - t * ln2_hi;
            lo = t * ln2_lo;
        }
        x = hi - lo;
        c = (hi - x) - lo;
    }
    else if (hx < 0x3c900000) { return x; }
    else k = 0;
    hfx = 0.5 * x;
    hxs = x * hfx;
#ifdef GROUPED
    R1 = one + hxs * Q1; h2 = hxs * hxs;
    R2 = Q2 + hxs * Q3; h4 = h2 * h2;
    R3 = Q4 + hxs * Q5;
    r1 = R1 + h2 * R2 + h4 * R3;
#else
    r1 = one + hxs * (Q1 + hxs * (Q2 + hxs * (Q3 + hxs * (Q4 + hxs * Q5))));
#endif
    t = 3.0 - r1 * hfx;
    e = hxs * ((r1 - t) / (6.0 - x * t));
    if (k == 0) return x - (x * e - hxs);
    else
    {
        e = (x * (e - c) - c);
        e -= hxs;
        if (k == -1) return 0.5 * (x - e) - 0.5;
        if (k == 1)
        {
            if (x < -0.25) return -2.0 * (e - (x + 0.5));
            else return one + 2.0 * (x - e);
        }
        if (k <= -2 || k > 56)
        {
            uint32_t high;
            y = one - (e - x);
            high = hi32(y);
            y = with_hi(y, high + (k << 20));
            return y - one;
        }
        t = one;
        if (k < 20)
        {
            uint32_t high;
            t = with_hi(t, 0x3ff00000 - (0x200000 >> k));
            y = t - (e - x);
            high = hi32(y);
            y = with_hi(y, high + (k << 20));
        }
        else
        {
            uint32_t high;
            t = with_hi(t, ((0x3ff - k) << 20));
            y = x - (e + t);
            y += one;
            high = hi32(y);
            y = with_hi(y, high + (k << 20));
        }
    }
    return y;
}
static double my_tanh(double x)
{
    const double one = 1.0, two = 2.0, tiny = 1.0e-300;
    double t, z;
    int32_t jx, ix;
    uint32_t lx;
    jx = (int32_t)hi32(x); lx = lo32(x);
    ix = jx & 0x7fffffff;
    if (ix >= 0x7ff00000) { if (jx >= 0) return one / x + one; else return one / x - one; }
    if (ix < 0x40360000)
    {
        if ((ix | lx) == 0) return x;
        if (ix < 0x3c800000) return x * (one + x);
        { int big = ix >= 0x3ff00000; double ax2 = two * fabs(x); t = my_expm1(big ? ax2 : -ax2); double q = (big ? two : t) / (t + two); z = big ? one - q : -q; }
    }
    else z = one - tiny;
    return (jx >= 0) ? z : -z;
}
/* straight-line tanh for 2^-55 <= |x| < 6.5 */
static double tanh_fast(double xin)
{
    const double one = 1.0, two = 2.0, ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10, invln2 = 1.44269504088896338700e+00,
                 Q1 = -3.33333333333331316428e-02, Q2 = 1.58730158725481460165e-03, Q3 = -7.93650757867487942473e-05,
                 Q4 = 4.00821782732936239552e-06, Q5 = -2.01099218183624371326e-07;
    const int32_t jx = (int32_t)hi32(xin), ix = jx & 0x7fffffff;
    const int big = ix >= 0x3ff00000;
    const double ax2 = two * fabs(xin);
    double x = big ? ax2 : -ax2;
    /* expm1(x), k in {-3..0} or {2..19} */
    const uint32_t hx = hi32(x) & 0x7fffffff;
    const int reduce = hx > 0x3fd62e42;
    const int k = reduce ? (int)(invln2 * x + (big ? 0.5 : -0.5)) : 0;
    const double tk = k;
    const double hi = x - tk * ln2_hi, lo = tk * ln2_lo;
    x = hi - lo;
    const double c = (hi - x) - lo;
    const double hfx = 0.5 * x, hxs = x * hfx;
    const double R1 = one + hxs * Q1, h2 = hxs * hxs, R2 = Q2 + hxs * Q3, h4 = h2 * h2, R3 = Q4 + hxs * Q5;
    const double r1 = R1 + h2 * R2 + h4 * R3;
    const double t3 = 3.0 - r1 * hfx;
    const double e = hxs * ((r1 - t3) / (6.0 - x * t3));
    const double r_k0 = x - (x * e - hxs);
    double e2 = (x * (e - c) - c);
    e2 -= hxs;
    const double r_m1 = 0.5 * (x - e2) - 0.5;
    double yn = one - (e2 - x);
    yn = with_hi(yn, hi32(yn) + ((uint32_t)k << 20));
    const double r_neg = yn - one;
    const double tm = with_hi(one, 0x3ff00000 - (0x200000 >> (k & 31)));
    double ym = tm - (e2 - x);
    ym = with_hi(ym, hi32(ym) + ((uint32_t)k << 20));
    const double t = (k == 0) ? r_k0 : (k == -1) ? r_m1 : (k <= -2) ? r_neg : ym;
    const double q = (big ? two : t) / (t + two);
    const double z = big ? one - q : -q;
    return (jx >= 0) ? z : -z;
}
static uint64_t s[2] = {0x9E3779B97F4A7C15ull, 0xD1B54A32D192ED03ull};
static inline uint64_t rnd(void) { uint64_t a = s[0], b = s[1]; s[0] = b; a ^= a << 23; s[1] = a ^ b ^ (a >> 17) ^ (b >> 26); return s[1] + b; }
int main(void)
{
    long bad_t = 0, bad_e = 0, n = 0;
    for (long it = 0; it < 200000000L; it++)
    {
        uint64_t u = rnd();
        double x;
        int mode = it % 3;
        if (mode == 0) x = ((double)(int64_t)u) * 0x1p-61;          /* uniform in (-4, 4) */
        else if (mode == 1) x = ((double)(int64_t)u) * 0x1p-58;     /* (-32, 32) */
        else { uint64_t m = u & 0xFFFFFFFFFFFFFull; int e = (int)((u >> 52) % 40) - 30; uint64_t b = ((uint64_t)(1023 + e) << 52) | m; memcpy(&x, &b, 8); if (u >> 63) x = -x; }
#ifdef FAST
        double a = (fabs(x) < 6.5 && fabs(x) >= 0x1p-55) ? tanh_fast(x) : my_tanh(x), b = tanh(x);
#else
        double a = my_tanh(x), b = tanh(x);
#endif
        if (memcmp(&a, &b, 8)) { if (bad_t < 5) printf("tanh differs x=%a mine=%a libm=%a\n", x, a, b); bad_t++; }
        double c = my_expm1(x), d = expm1(x);
        if (memcmp(&c, &d, 8)) { if (bad_e < 5) printf("expm1 differs x=%a mine=%a libm=%a\n", x, c, d); bad_e++; }
        n++;
    }
    printf("n=%ld tanh mismatches=%ld expm1 mismatches=%ld\n", n, bad_t, bad_e);
    return 0;
}
