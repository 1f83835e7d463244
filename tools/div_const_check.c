// Empirical check: for float a, b:  q0 = a*y (y = RN(1/b)), r = fma(-b,q0,a), q1 = fma(r,y,q0)  ==  a/b (IEEE) ?
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
static uint64_t s = 88172645463325252ULL;
static inline uint64_t rnd(void){ s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static inline float frand(float lo, float hi){ return lo + (hi-lo) * (float)((rnd() >> 11) * (1.0/9007199254740992.0)); }
int main(void){
    long bad = 0, n = 0;
    for (int ib = 0; ib < 4000; ++ib) {
        float b;
        if (ib < 64) { float t[] = {8.f,7.f,9.f,3.f,2.f,6.f,5.f,1.f,0.5f,10.f,12.f,400.f,1.9999999f,3.9999998f,0.99999994f,1.0000001f}; b = t[ib%16]; }
        else b = frand(0.01f, 500.f);
        if (ib % 97 == 0) { union {float f; uint32_t u;} x; x.f = b; x.u |= 0x007fffff; b = x.f; }   // all-ones significand
        volatile float y = 1.0f / b;
        for (int i = 0; i < 500000; ++i) {
            float a = (i & 1) ? frand(-300.f, 300.f) : frand(-1e-3f, 20.f);
            float q0 = a * y;
            float r = fmaf(-b, q0, a);
            float q1 = fmaf(r, y, q0);
            volatile float q = a / b;
            if (q1 != q) { if (bad < 10) printf("mismatch a=%a b=%a q1=%a q=%a\n", a, b, q1, q); ++bad; }
            ++n;
        }
    }
    printf("tested %ld pairs, mismatches %ld\n", n, bad);
    return 0;
}
