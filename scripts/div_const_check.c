// Exhaustive-ish check of jd_div_const (k_oqpsk_fb.h): x / d through the reciprocal and fma corrections against the IEEE quotient, for the
// constants the sample loops divide by.  gcc -O2 -ffp-contract=off -march=native scripts/div_const_check.c -lm && ./a.out  (38 s; prints 0 differences)
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
static inline double divc(double x, double d, double rd)
{
    double q0 = x * rd;
    double r0 = fma(-d, q0, x);
    double q1 = fma(r0, rd, q0);
    double r1 = fma(-d, q1, x);
    return fma(r1, rd, q1);
}
static inline double divc3(double x, double d, double rd)
{
    double q0 = x * rd;
    double r0 = fma(-d, q0, x);
    return fma(r0, rd, q0);
}
static uint64_t s[2] = {0x9E3779B97F4A7C15ull, 0xD1B54A32D192ED03ull};
static inline uint64_t rnd(void) { uint64_t a = s[0], b = s[1]; s[0] = b; a ^= a << 23; s[1] = a ^ b ^ (a >> 17) ^ (b >> 26); return s[1] + b; }
int main(void)
{
    const double ds[] = {192000.0, 96000.0, 48000.0, 360.0, 19999.0, 800.0, 400.0, 600.0, 40.0, 128.0*3, 2340.0, 585.0, 1170.0};
    for (unsigned k = 0; k < sizeof ds / sizeof ds[0]; k++)
    {
        const double d = ds[k], rd = 1.0 / d;
        long bad5 = 0, bad3 = 0, n = 0;
        for (long it = 0; it < 150000000L; it++)
        {
            uint64_t u = rnd();
            double x;
            int mode = it & 3;
            if (mode == 0) { // random mantissa, exponent in [-60, 60]
                uint64_t m = u & 0xFFFFFFFFFFFFFull; int e = (int)((u >> 52) % 121) - 60;
                uint64_t bits = ((uint64_t)(1023 + e) << 52) | m; memcpy(&x, &bits, 8); if (u >> 63) x = -x;
            } else if (mode == 1) { // near a multiple: x = RN(d * q) +- few ulps, q random
                uint64_t m = u & 0xFFFFFFFFFFFFFull; int e = (int)((u >> 52) % 61) - 30;
                uint64_t bits = ((uint64_t)(1023 + e) << 52) | m; double q; memcpy(&q, &bits, 8);
                x = d * q; int j = (int)((u >> 58) & 7) - 3; for (; j > 0; j--) x = nextafter(x, INFINITY); for (; j < 0; j++) x = nextafter(x, -INFINITY);
            } else if (mode == 2) { // midpoint-ish quotient: q = (m + 0.5 ulp) -> x = d*q computed in long double
                uint64_t m = u & 0xFFFFFFFFFFFFFull; int e = (int)((u >> 52) % 41) - 20;
                uint64_t bits = ((uint64_t)(1023 + e) << 52) | m; double q; memcpy(&q, &bits, 8);
                long double qm = ((long double)q + (long double)nextafter(q, INFINITY)) / 2;
                x = (double)(qm * (long double)d); int j = (int)((u >> 58) & 7) - 3; for (; j > 0; j--) x = nextafter(x, INFINITY); for (; j < 0; j++) x = nextafter(x, -INFINITY);
            } else { // moderate uniform values
                x = (double)(int64_t)u * 0x1p-40;
            }
            double t = x / d;
            if (divc(x, d, rd) != t) bad5++;
            if (divc3(x, d, rd) != t) bad3++;
            n++;
        }
        printf("d=%g: n=%ld bad(5-op)=%ld bad(3-op)=%ld\n", d, n, bad5, bad3);
    }
    return 0;
}
