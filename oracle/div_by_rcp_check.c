/* TEST INFRASTRUCTURE.  CPU check of the exact division the HIP kernels use for BalancedAllocation's quotients
 * (open-simulator_amd/csrc/simon_device.h: div_by_rcp): q = req / alloc formed from the stored reciprocal
 * RN(1 / alloc) with two Markstein corrections must equal the IEEE quotient Go computes
 * (V/framework/plugins/noderesources/balanced_allocation.go:82-119: float64(requested) / float64(allocatable)).
 * Operands: integers below 2^53 -- random, Ki/Mi-aligned, near 1, tiny, far above 1, and the all-ones significands
 * (the hard case of Markstein's theorem).  usage: div_by_rcp_check [iterations per bit width]; exit 1 on a mismatch. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

static double div_by_rcp(double num, double den, double rc) {
    double q = num * rc;
    double e = fma(-q, den, num);
    q = fma(e, rc, q);
    e = fma(-q, den, num);
    q = fma(e, rc, q);
    return q;
}

/* one correction: exact for den < 2^31, num < 2^32 (proof in simon_device.h: div_by_rcp1) -- the score-table kernel's domain */
static double div_by_rcp1(double num, double den, double rc) {
    double q = num * rc;
    double e = fma(-q, den, num);
    return fma(e, rc, q);
}

static uint64_t state = 88172645463325252ull;
static uint64_t rnd(void) { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return state; }

int main(int argc, char** argv) {
    long iters = argc > 1 ? atol(argv[1]) : 400000, bad = 0, n = 0;
    for (int bits = 1; bits <= 52; bits++) {
        for (long it = 0; it < iters; it++) {
            uint64_t den = (rnd() >> (64 - bits)) | 1ull;
            if (it & 1) den = (den >> 10 << 10) | (1ull << (bits > 10 ? bits - 1 : 0));
            uint64_t num = rnd() % (den + den / 8 + 2);
            if ((it & 7) == 3) num = den - (rnd() % 3);
            if ((it & 7) == 5) num = rnd() % 4096;
            if ((it & 7) == 7) num = (rnd() >> 11) % ((den << (bits < 30 ? 20 : 1)) + 1);      /* far above 1 */
            double d = (double)den, x = (double)num;
            n++;
            if (div_by_rcp(x, d, 1.0 / d) != x / d) {
                if (bad++ < 10) printf("MISMATCH num=%llu den=%llu\n", (unsigned long long)num, (unsigned long long)den);
            }
        }
    }
    for (int bits = 2; bits <= 53; bits++) {
        uint64_t den = (1ull << bits) - 1;
        for (long it = 0; it < iters / 16 + 1; it++) {
            uint64_t num = rnd() % (den + 3);
            double d = (double)den, x = (double)num;
            n++;
            if (div_by_rcp(x, d, 1.0 / d) != x / d) {
                if (bad++ < 20) printf("MISMATCH(all-ones) num=%llu den=%llu\n", (unsigned long long)num, (unsigned long long)den);
            }
        }
    }
    /* the one-correction variant on ITS domain: denominators of 1..31 bits (random, 2^k - 1 - {0,1,2}, 2^(k-1) + {0..4}),
     * numerators up to 2 den + 1, den - {0,1,2}, small, and up to 2^31 */
    long n1 = 0, bad1 = 0;
    for (int bits = 1; bits <= 31; bits++) {
        for (long it = 0; it < iters * 4; it++) {
            uint64_t den = (rnd() >> (64 - bits)) | 1ull;
            if (bits > 1 && (it & 3) == 1) den = (1ull << bits) - 1 - (rnd() % 3);
            if ((it & 3) == 2) den = (1ull << (bits - 1)) + (rnd() % 5);
            if (den == 0) den = 1;
            uint64_t num = rnd() % (2 * den + 2);
            if ((it & 7) == 3) num = den - (rnd() % 3);
            if ((it & 15) == 5) num = rnd() % 4096;
            if ((it & 15) == 9) num = rnd() >> 33;
            double d = (double)den, x = (double)num;
            n1++;
            if (div_by_rcp1(x, d, 1.0 / d) != x / d) {
                if (bad1++ < 10) printf("MISMATCH(one correction) num=%llu den=%llu\n", (unsigned long long)num, (unsigned long long)den);
            }
        }
    }
    printf("checked %ld operand pairs, %ld mismatches; one-correction variant on its domain: %ld pairs, %ld mismatches\n", n, bad, n1, bad1);
    return bad != 0 || bad1 != 0;
}
