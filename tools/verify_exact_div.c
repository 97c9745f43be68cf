// verify_exact_div.c — exhaustive check of "division by a constant without the divide" (used in de_math.h / oracle_math.h).
//
//   q0 = x * rc;  r = fma(-q0, c, x);  q = fma(r, rc, q0)        with rc = RN(1/c)
//
// returns the correctly rounded x / c for EVERY binary32 x only for suitable c.  The property depends on x's
// significand alone (scaling by 2^k is exact away from under/overflow), so testing all 2^23 significands at two
// exponents per sign settles it for the whole normal range.  Build: gcc -O2 -mfma -ffp-contract=off tools/verify_exact_div.c -lm
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static long check(float c, int verbose) {
    volatile float one = 1.0f;
    float rc = one / c;
    long bad = 0;
    for (int e = 0; e < 3; ++e) {
        uint32_t expo = (e == 0) ? 127u : (e == 1 ? 140u : 100u);
        for (uint32_t m = 0; m < (1u << 23); ++m) {
            float x = u2f((expo << 23) | m);
            volatile float want = x / c;
            float q0 = x * rc;
            float r = __builtin_fmaf(-q0, c, x);
            float q = __builtin_fmaf(r, rc, q0);
            if (f2u(q) != f2u(want)) { if (verbose && bad < 3) printf("   x=%a q=%a want=%a\n", x, q, want); ++bad; }
        }
    }
    return bad;
}
int main(void) {
    const float consts[] = {255.0f, 0x1.921fb6p+1f /* pi */, 49.0f, 532307548.4168f, 1.225f, 8136.646f, 6000.0f, 12000.0f, 3.0f, 441.0f, 2.4f, 0.1f * 0.029f, 0.02f * 0.029f};
    const char* names[] = {"255", "pi", "49", "532307548.4168", "1.225", "8136.646", "6000", "12000", "3", "441", "2.4", "0.1f*0.029f", "0.02f*0.029f"};
    for (unsigned i = 0; i < sizeof(consts) / sizeof(consts[0]); ++i) {
        long bad = check(consts[i], 1);
        printf("c = %-16s rc = %a : %ld mismatches of %ld\n", names[i], 1.0f / consts[i], bad, 3L << 23);
    }
    return 0;
}
