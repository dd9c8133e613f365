/*
 * oracle/orc_boxbox.c -- dBoxBox (box-box SAT + face clipping) restated.
 * TEST INFRASTRUCTURE (see orc.h).  Placeholder until SURVEY section 8 row f-2 is
 * built: BASELINE configs 1-4 never bring two boxes into AABB overlap (grid
 * pitch 2.5 m, sides <= 1.0 m), so reaching this is a scene error, not a
 * silent zero.
 */
#include <stdio.h>
#include <stdlib.h>
#include "orc_internal.h"

int orc_collide_box_box(const real *p1, const real *R1, const real *side1,
                        const real *p2, const real *R2, const real *side2,
                        int maxc, orc_contactgeom *out)
{
    (void)p1; (void)R1; (void)side1; (void)p2; (void)R2; (void)side2; (void)maxc; (void)out;
    fprintf(stderr, "orc_collide_box_box: box-box narrowphase not in the oracle yet (row f-2)\n");
    abort();
    return 0;
}
