/*
 * oracle/orc_rand.c -- the two generators the path touches (TEST INFRASTRUCTURE).
 *
 * 1. orc_ref_rand_*: restatement of the reference's scene PRNG,
 *    /root/reference/src/rand.c:7-13 (Rand_Next), :21 (Rand_Int), :29
 *    (Rand_Double).  Pinned by the golden values of SURVEY.md section 8c
 *    (tests/golden/rand_golden.json).
 * 2. orc_ode_rand*: [ODE-recall] ODE's global LCG (misc.cpp dRand/dRandInt)
 *    that QuickStep uses to reshuffle constraint rows.  Unpinned.
 */
#include "orc_internal.h"

/* ---- reference rand.c ---------------------------------------------------- */
static uint32_t ref_state = 0;                 /* rand.c:5 */

void orc_ref_rand_seed(uint32_t s) { ref_state = s; }

uint32_t orc_ref_rand_next(void)               /* rand.c:7-13 */
{
    ref_state += 0xE120FC15u;                  /* Weyl increment */
    uint64_t t = (uint64_t)ref_state * 0x4A39B70Du;
    uint32_t m1 = (uint32_t)((t >> 32) ^ t);
    t = (uint64_t)m1 * 0x12FAD5C9u;
    return (uint32_t)((t >> 32) ^ t);
}

int32_t orc_ref_rand_int(int32_t min, int32_t max)   /* rand.c:15-22 */
{
    if (min >= max) return 0;
    return (int32_t)(orc_ref_rand_next() % (uint32_t)(max - min)) + min;
}

double orc_ref_rand_double(double min, double max)   /* rand.c:24-30 */
{
    return min + orc_ref_rand_next() / (double)0xFFFFFFFFu * (max - min);
}

/* ---- ODE global LCG ------------------------------------------------------ */
static uint32_t ode_seed = 0;

void orc_rand_seed(uint32_t s) { ode_seed = s; }

uint32_t orc_ode_rand(void)
{
    ode_seed = 1664525u * ode_seed + 1013904223u;
    return ode_seed;
}

int orc_ode_rand_int(int n)
{
    return (int)(((uint64_t)orc_ode_rand() * (uint32_t)n) >> 32);
}
