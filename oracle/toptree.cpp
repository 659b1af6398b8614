/* toptree.cpp — CPU restatement of the reference's top-tree walks (export detection).
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 *   GravTopTreeWalk::toptree_visit<COUNT|EXPORT>      libgadget/gravshort2.hpp:362-438
 *   TopTreeWalk::toptree_visit (cull_node)            libgadget/localtreewalk2.h:210-259, 154-182
 *   TopTreeWalk::export_particle / export_count       libgadget/localtreewalk2.h:269-324
 *
 * Pinning: the reference has no unit test of its own for the export table; what pins this file is the closure
 * property tests/test_gpu_toptree.py checks with it — primary walk over the local tree + secondary walks at the
 * exported NodeLists reproduce the single-domain walk interaction for interaction — on top of the walk
 * restatements that the reference's gates pin (oracle/README.md). */
#include <math.h>
#include "oracle.h"

namespace {

struct Exporter { /* the state TopTreeWalk keeps per target: localtreewalk2.h:203, 326-333 */
    const shq_topleaf *TopLeaves;
    int64_t lastnode;
    int lasttask = 0;
    int nodelistindex = 0;

    /* export_particle, localtreewalk2.h:269-312; table == NULL: export_count, :315-324 */
    int64_t add(int no, int target, int64_t nexp, shq_data_index *table, int64_t BunchSize)
    {
        const shq_topleaf *tl = &TopLeaves[no - lastnode];
        const int task = tl->Task;
        if(!table) {
            if(nexp >= 1 && lasttask == task && nodelistindex < 4) {
                nodelistindex++;
                return nexp;
            }
            lasttask = task;
            nodelistindex = 1;
            return nexp + 1;
        }
        if(nexp >= 1 && lasttask == task) {
            if(nodelistindex < 4) {
                table[nexp - 1].NodeList[nodelistindex] = tl->treenode;
                nodelistindex++;
                return nexp;
            }
        }
        if(nexp >= BunchSize)
            return -1;
        table[nexp].Task = task;
        table[nexp].Index = target;
        table[nexp].NodeList[0] = tl->treenode;
        for(int i = 1; i < 4; i++)
            table[nexp].NodeList[i] = -1;
        nodelistindex = 1;
        lasttask = task;
        return nexp + 1;
    }
};

/* gravshort2.hpp:152-167 */
inline int discard_node(double len, double r2, const double center[3], const double inpos[3], double Box, double rcut, double rcut2)
{
    if(r2 <= rcut2)
        return 0;
    const double eff_dist = rcut + 0.5 * len;
    for(int i = 0; i < 3; i++)
        if(fabs(orc_nearest(center[i] - inpos[i], Box)) > eff_dist)
            return 1;
    return 0;
}

/* gravshort2.hpp:172-193 */
inline int open_node(double len, double mass, double r2, const double center[3], const double inpos[3], double Box, double aold,
                     int TreeUseBH, double BHOpeningAngle2)
{
    if((TreeUseBH == 0) && (mass * len * len > r2 * r2 * aold))
        return 1;
    if(len * len / r2 > BHOpeningAngle2)
        return 1;
    const double inside = 0.6 * len;
    return fabs(orc_nearest(center[0] - inpos[0], Box)) < inside && fabs(orc_nearest(center[1] - inpos[1], Box)) < inside &&
           fabs(orc_nearest(center[2] - inpos[2], Box)) < inside;
}

/* localtreewalk2.h:154-182 */
inline int cull_node(const double *Pos, double BoxSize, double Hsml, const shq_node *cur, bool symmetric)
{
    double dist = (symmetric ? fmax(cur->hmax, Hsml) : Hsml) + 0.5 * cur->len;
    double r2 = 0;
    for(int d = 0; d < 3; d++) {
        const double dx = orc_nearest(cur->center[d] - Pos[d], BoxSize);
        if(dx > dist || dx < -dist)
            return 0;
        r2 += dx * dx;
    }
    dist += 0.5 * (1.7320508075688772 - 1.0) * cur->len;
    return r2 > dist * dist ? 0 : 1;
}

template <typename Skip>
int64_t toptree(const shq_node *nodes, int64_t firstnode, int64_t lastnode, const shq_topleaf *topleaves, const int32_t *targets,
                int64_t ntargets, int32_t *counts, shq_data_index *table, int64_t capacity, Skip skip)
{
    const shq_node *N = nodes - firstnode;
    int64_t total = 0;
    for(int64_t t = 0; t < ntargets; t++) {
        const int target = targets ? targets[t] : (int) t;
        Exporter ex{topleaves, lastnode};
        int64_t nexp = 0;
        int no = (int) firstnode; /* the top-tree walk always starts from the root */
        while(no >= 0) {
            const shq_node *nop = &N[no];
            if(skip(target, nop)) { /* discarded, or accepted without opening: no export */
                no = nop->sibling;
                continue;
            }
            if(SHQ_NODE_CHILDTYPE(nop->flags) == SHQ_PSEUDO_NODE_TYPE) {
                nexp = ex.add(nop->suns[0], target, nexp, table ? table + total : nullptr, capacity - total);
                if(nexp < 0)
                    return -1;
                no = nop->sibling;
                continue;
            }
            if(SHQ_NODE_TOPLEVEL(nop->flags) && !SHQ_NODE_INTERNALTOPLEVEL(nop->flags)) { /* a local top-level leaf */
                no = nop->sibling;
                continue;
            }
            no = nop->suns[0];
        }
        if(counts)
            counts[t] = (int32_t) nexp;
        total += nexp;
    }
    return total;
}

} // namespace

extern "C" int64_t orc_grav_toptree(const shq_node *nodes, int64_t firstnode, int64_t lastnode, const shq_topleaf *topleaves,
                                    const double *pos, const double *oldacc, const int32_t *targets, int64_t ntargets,
                                    const shq_grav_params *p, int32_t *counts, shq_data_index *table, int64_t capacity)
{
    const double rcut = p->Rcut, rcut2 = rcut * rcut, Box = p->BoxSize;
    return toptree(nodes, firstnode, lastnode, topleaves, targets, ntargets, counts, table, capacity, [&](int target, const shq_node *nop) {
        const double *inpos = &pos[3 * (int64_t) target];
        const double aold = p->ErrTolForceAcc * oldacc[target];
        double dx[3];
        for(int i = 0; i < 3; i++)
            dx[i] = orc_nearest(nop->cofm[i] - inpos[i], Box);
        const double r2 = dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2];
        return discard_node(nop->len, r2, nop->center, inpos, Box, rcut, rcut2) ||
               !open_node(nop->len, nop->mass, r2, nop->center, inpos, Box, aold, p->TreeUseBH, p->BHOpeningAngle2);
    });
}

extern "C" int64_t orc_ngb_toptree(const shq_node *nodes, int64_t firstnode, int64_t lastnode, const shq_topleaf *topleaves,
                                   const double *pos, const double *hsml, int symmetric, double BoxSize, const int32_t *targets,
                                   int64_t ntargets, int32_t *counts, shq_data_index *table, int64_t capacity)
{
    return toptree(nodes, firstnode, lastnode, topleaves, targets, ntargets, counts, table, capacity, [&](int target, const shq_node *nop) {
        return 0 == cull_node(&pos[3 * (int64_t) target], BoxSize, hsml[target], nop, symmetric != 0);
    });
}
