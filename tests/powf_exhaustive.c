/* TEST INFRASTRUCTURE.  Compares t-route_amd/csrc/det_pow.h with this machine's libm powf over
 * float bit patterns, at the exponents the engine uses: 2/3 and 5/3 rounded to float
 * (MCsingleSegStime_f2py_NOLOOP.f90:252-253) and 3/2 (module_levelpool.F:304).
 *
 *   gcc -O2 -ffp-contract=off -mfma -fopenmp tests/powf_exhaustive.c -lm -o /tmp/powf_exh
 *   /tmp/powf_exh [stride]        stride 1 = every one of the 2^32 patterns (about a minute on 8 cores)
 *
 * Prints the number of mismatching patterns per exponent; exit status 1 if any. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../t-route_amd/csrc/det_pow.h"

int main(int argc, char **argv)
{
    const uint64_t stride = argc > 1 ? strtoull(argv[1], 0, 10) : 1;
    /* 2/3 and 5/3: Muskingum-Cunge kernel; 3/2: level-pool weir law (module_levelpool.F:304) */
    const float ys[3] = {2.0f / 3.0f, 5.0f / 3.0f, 3.0f / 2.0f};
    int bad_total = 0;
    for (int e = 0; e < 3; ++e) {
        const float y = ys[e];
        uint64_t bad = 0, n = 0;
#pragma omp parallel for reduction(+ : bad, n) schedule(static)
        for (uint64_t u = 0; u < (1ull << 32); u += stride) {
            float x, a, b;
            uint32_t ux = (uint32_t)u, ua, ub;
            memcpy(&x, &ux, 4);
            a = powf(x, y);
            b = trmc_det_powf(x, y, trmc_pow_tab_init);
            memcpy(&ua, &a, 4);
            memcpy(&ub, &b, 4);
            n++;
            if (ua != ub && !(a != a && b != b)) bad++;
        }
        printf("y=%.9g: %llu patterns, %llu mismatches\n", y, (unsigned long long)n, (unsigned long long)bad);
        bad_total += bad != 0;
    }
    return bad_total ? 1 : 0;
}
