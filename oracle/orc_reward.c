/* oracle/orc_reward.c -- the five task reward predicates (gym_guided_vision/gym_guided_vision/env.py:
 * InsertPeg :425-472, SlotInsertion :546-589, SewNeedle :640-690, TubeTransfer :738-779,
 * HookPackage :820-863) restated over per-geom class bits (compile.py:geom_class).
 * TEST INFRASTRUCTURE ONLY (orc.h). */
#include "orc.h"

enum { CL = 1, CR = 2, CT = 4, CA = 8, CB = 16, CC = 32, CD = 64 };

static int has(int c1, int c2, int a, int b) { return ((c1 & a) && (c2 & b)) || ((c2 & a) && (c1 & b)); }

int orc_max_reward(const orc_model* m) {
    static const int mx[5] = {4, 4, 5, 3, 4}; /* env.py:423, 509, 598, 699, 788 */
    return mx[m->task_id];
}

int orc_reward_from_pairs(const orc_model* m, const int* gp, int ncon, int* latch) {
    int tl = 0, tr = 0, a_table = 0, b_table = 0, ab = 0, cd = 0, ad = 0, ac = 0;
    int t = m->task_id;
    for (int i = 0; i < ncon; i++) {
        if (gp[2 * i] < 0 || gp[2 * i + 1] < 0) continue;
        int c1 = m->geom_class[gp[2 * i]], c2 = m->geom_class[gp[2 * i + 1]];
        a_table |= has(c1, c2, CT, CA);
        b_table |= has(c1, c2, CT, CB);
        ab |= has(c1, c2, CA, CB);
        cd |= has(c1, c2, CC, CD);
        ad |= has(c1, c2, CA, CD);
        ac |= has(c1, c2, CA, CC);
        switch (t) {
            case 0: tr |= has(c1, c2, CA, CR); tl |= has(c1, c2, CB, CL); break; /* peg-right, hole-left */
            case 3: tr |= has(c1, c2, CA, CR); tl |= has(c1, c2, CB, CL); break; /* tube1-right, tube2-left */
            default: tr |= has(c1, c2, CA, CR); tl |= has(c1, c2, CA, CL); break;
        }
    }
    int r = 0;
    switch (t) {
        case 0: /* InsertPeg env.py:463-472 */
            if (tl && tr) r = 1;
            if (tl && tr && !a_table && !b_table) r = 2;
            if (ab && !a_table && !b_table) r = 3;
            if (ac) r = 4;
            break;
        case 1: /* SlotInsertion :580-589 */
            if (tl && tr) r = 1;
            if (tl && tr && !a_table) r = 2;
            if (ab && !a_table) r = 3;
            if (cd) r = 4;
            break;
        case 2: /* SewNeedle :679-690, latched _threaded_needle */
            if (cd) *latch = 1;
            if (tr) r = 1;
            if (tr && !a_table) r = 2;
            if (ab && !a_table) r = 3;
            if (*latch) r = 4;
            if (tl && !tr && !a_table && !ad && *latch) r = 5;
            break;
        case 3: /* TubeTransfer :772-779 */
            if (tl && tr) r = 1;
            if (tl && tr && !a_table && !b_table) r = 2;
            if (cd) r = 3;
            break;
        case 4: /* HookPackage :854-863; hook = class B, package-* = class A */
            if (tl && tr) r = 1;
            if (tl && tr && !a_table) r = 2;
            if (ab && !a_table) r = 3;
            if (cd) r = 4;
            break;
    }
    return r;
}
