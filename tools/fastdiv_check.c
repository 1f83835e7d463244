#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
static float seq(float n, float a, float y0)
{
    float e = fmaf(-a, y0, 1.0f);
    float y = fmaf(e, y0, y0);
    float q = n * y;
    float r = fmaf(-a, q, n);
    q = fmaf(r, y, q);
    r = fmaf(-a, q, n);
    q = fmaf(r, y, q);
    return q;
}
int main(void)
{
    uint64_t s = 88172645463325252ULL; long bad = 0, badcount[3] = {0,0,0}; long N = 200000000;
    for (long i = 0; i < N; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        float a = -(float)((s & 0xFFFFFF) + 1) / 16777216.0f * 1.2f;           /* (-1.2, 0) */
        if ((s >> 60) == 0) a *= 1e-3f;                                          /* grazing */
        float n = -(float)(((s >> 24) & 0xFFFFFF) + 1) / 16777216.0f * 400.0f;   /* path lengths up to 400 m */
        if ((s >> 56 & 15) == 1) n *= 1e-4f;
        float y = 1.0f / a, ref = n / a;
        float ys[3] = {y, nextafterf(y, 0.0f), nextafterf(y, -INFINITY)};
        for (int k = 0; k < 3; ++k) if (seq(n, a, ys[k]) != ref) { ++bad; ++badcount[k]; if (bad < 5) printf("n=%a a=%a k=%d got %a ref %a\n", n, a, k, seq(n,a,ys[k]), ref); }
    }
    printf("checked %ld, mismatches %ld (%ld %ld %ld)\n", N, bad, badcount[0], badcount[1], badcount[2]);
    return 0;
}
