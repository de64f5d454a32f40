/* Exhaustive accuracy check of include/kajo_strictmath.h against the correctly rounded value (binary64 libm, rounded
 * to binary32) -- the claims of the header's accuracy paragraph. Minutes of CPU time on 8 threads.
 *   gcc -O2 -mfma -ffp-contract=off -Iinclude -o /tmp/sm_exh tools/strictmath_exhaustive.c -lm -lpthread && /tmp/sm_exh
 *   sin, cos   every normal binary32 in [-2, 6.5]
 *   asin, acos every normal binary32 in [-1, 1]
 *   pow        every 7th binary32 in (0, 1] (incl. subnormals) for nine exponents
 */
#include "kajo_strictmath.h"
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>

static long ulpd(float a, float b)
{
    if (a != a && b != b)
        return 0;
    int32_t ia = (int32_t)ksm_bits32(a), ib = (int32_t)ksm_bits32(b);
    if (ia < 0) ia = (int32_t)0x80000000 - ia;
    if (ib < 0) ib = (int32_t)0x80000000 - ib;
    long d = (long)ia - ib;
    return d < 0 ? -d : d;
}

typedef struct { int fn; uint32_t lo, hi, step; int neg; float y; long max[2], exact[2], n; } Job;

static void* run(void* p)
{
    Job* j = p;
    j->max[0] = j->max[1] = j->exact[0] = j->exact[1] = j->n = 0;
    for (uint64_t u = j->lo; u < j->hi; u += j->step) {
        float x = ksm_from_bits32((uint32_t)u | (j->neg ? 0x80000000u : 0u)), g0, g1, w0, w1;
        if (j->fn == 0) { kajo_sincosf(x, &g0, &g1); w0 = (float)sin((double)x); w1 = (float)cos((double)x); }
        else if (j->fn == 1) { g0 = kajo_asinf(x); g1 = kajo_acosf(x); w0 = (float)asin((double)x); w1 = (float)acos((double)x); }
        else { g0 = g1 = kajo_powf(x, j->y); w0 = w1 = (float)pow((double)x, (double)j->y); }
        long d0 = ulpd(g0, w0), d1 = ulpd(g1, w1);
        if (d0 > j->max[0]) j->max[0] = d0;
        if (d1 > j->max[1]) j->max[1] = d1;
        j->exact[0] += d0 == 0;
        j->exact[1] += d1 == 0;
        j->n++;
    }
    return 0;
}

static int sweep(const char* name, int fn, uint32_t start, uint32_t top, uint32_t step, int neg, float y)
{
    enum { T = 8 };
    pthread_t th[T];
    Job jobs[T];
    for (int i = 0; i < T; i++) {
        jobs[i] = (Job){fn, (uint32_t)(start + (uint64_t)(top - start) * i / T), (uint32_t)(start + (uint64_t)(top - start) * (i + 1) / T), step, neg, y};
        pthread_create(&th[i], 0, run, &jobs[i]);
    }
    long m[2] = {0, 0}, e[2] = {0, 0}, n = 0;
    for (int i = 0; i < T; i++) {
        pthread_join(th[i], 0);
        for (int k = 0; k < 2; k++) { if (jobs[i].max[k] > m[k]) m[k] = jobs[i].max[k]; e[k] += jobs[i].exact[k]; }
        n += jobs[i].n;
    }
    printf("%-34s n = %10ld  max ulp %ld / %ld  correctly rounded %.5f / %.5f\n", name, n, m[0], m[1], (double)e[0] / n, (double)e[1] / n);
    return m[0] > 1 || m[1] > 1;
}

int main(void)
{
    int bad = 0;
    const uint32_t first = 0x00800000u; /* smallest normal */
    bad |= sweep("sin / cos  [2^-126, 6.5]", 0, first, ksm_bits32(6.5f) + 1, 1, 0, 0);
    bad |= sweep("sin / cos  [-2, -2^-126]", 0, first, ksm_bits32(2.0f) + 1, 1, 1, 0);
    bad |= sweep("asin / acos [2^-126, 1]", 1, first, ksm_bits32(1.0f) + 1, 1, 0, 0);
    bad |= sweep("asin / acos [-1, -2^-126]", 1, first, ksm_bits32(1.0f) + 1, 1, 1, 0);
    const float ys[] = {100.f, 1000.f, 10.f, 3.f, 2.2f, .5f, 1.f / 2.2f, 1.f / 11.f, 1.f / 101.f};
    for (int i = 0; i < 9; i++) {
        char name[64];
        snprintf(name, sizeof name, "pow(x, %g), x in (0, %g]", ys[i], ys[i] == 2.2f ? 16.0 : 1.0);
        bad |= sweep(name, 2, 1, ksm_bits32(ys[i] == 2.2f ? 16.0f : 1.0f) + 1, 7, 0, ys[i]);
    }
    printf(bad ? "FAILED: an argument is more than 1 ulp off\n" : "ok: everything within 1 ulp of the correctly rounded value\n");
    return bad;
}
