/* oracle/grav.cpp — CPU restatement of the reference oct-tree and short-range gravity walk.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 */
#include "oracle.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include <vector>
#include <omp.h>

#define NODEFULL (1 << 16) /* libgadget/forcetree.h:14 */

namespace {

inline void set_childtype(shq_node &nd, unsigned t) { nd.flags = (nd.flags & ~(3u << 3)) | (t << 3); }
inline unsigned childtype(const shq_node &nd) { return SHQ_NODE_CHILDTYPE(nd.flags); }

struct Builder {
    const double *pos;
    const float *mass;
    const double *hsml;
    shq_node *N;      /* shifted so N[firstnode] is the first node (forcetree.cpp:1388) */
    int64_t firstnode, lastnode, nnext;
    int32_t *father;
    bool overflow = false;

    /* libgadget/forcetree.cpp:277-283 get_subnode */
    int subnode(const shq_node &nd, const double *p) const
    {
        return (p[0] > nd.center[0]) + ((p[1] > nd.center[1]) << 1) + ((p[2] > nd.center[2]) << 2);
    }
    /* libgadget/forcetree.cpp:302-328 init_internal_node */
    void init_child(shq_node &c, const shq_node &parent, int sub)
    {
        const double lenhalf = 0.25 * parent.len;
        c.len = 0.5 * parent.len;
        c.sibling = -10;
        c.father = -10;
        c.flags = 0; /* ChildType = PARTICLE_NODE_TYPE */
        for(int j = 0; j < 3; j++)
            c.center[j] = parent.center[j] + ((sub & (1 << j)) ? lenhalf : -lenhalf);
        for(int j = 0; j < SHQ_NMAXCHILD; j++)
            c.suns[j] = -1;
        c.noccupied = 0;
        c.cofm[0] = c.cofm[1] = c.cofm[2] = 0;
        c.mass = 0;
        c.hmax = 0;
    }
    /* libgadget/forcetree.cpp:947-966 add_particle_moment_to_node; all particles here are
     * inactive for hmax purposes only when hsml is supplied (tests treat them as such). */
    void add_moment(shq_node &nd, int p)
    {
        nd.mass += mass[p];
        for(int k = 0; k < 3; k++)
            nd.cofm[k] += mass[p] * pos[3 * p + k];
        if(hsml) {
            for(int j = 0; j < 3; j++) {
                double v = fabs(pos[3 * p + j] - nd.center[j]) + hsml[p] - nd.len / 2.;
                if(v > nd.hmax)
                    nd.hmax = v;
            }
        }
    }
    /* libgadget/forcetree.cpp:352-361 modify_internal_node */
    void attach(int node, int slot, int p)
    {
        if(father)
            father[p] = node;
        N[node].suns[slot] = p;
        add_moment(N[node], p);
    }
    /* libgadget/forcetree.cpp:366-477 create_new_node_layer */
    bool new_layer(int firstparent, int p_toplace)
    {
        int parent = firstparent;
        while(1) {
            shq_node &np = N[parent];
            int newsuns[SHQ_NMAXCHILD];
            int oldsuns[SHQ_NMAXCHILD];
            memcpy(oldsuns, np.suns, sizeof(oldsuns));
            if(nnext + 8 > lastnode) {
                overflow = true;
                return false;
            }
            newsuns[0] = (int) nnext;
            nnext += 8;
            for(int i = 0; i < 8; i++) {
                newsuns[i] = newsuns[0] + i;
                init_child(N[newsuns[i]], np, i);
                N[newsuns[i]].father = parent;
            }
            for(int i = 0; i < SHQ_NMAXCHILD; i++) {
                int sub = subnode(np, &pos[3 * (int64_t) oldsuns[i]]);
                shq_node &ch = N[newsuns[sub]];
                attach(newsuns[sub], ch.noccupied, oldsuns[i]);
                ch.noccupied++;
            }
            memcpy(np.suns, newsuns, sizeof(newsuns));
            for(int i = 0; i < 7; i++)
                N[np.suns[i]].sibling = np.suns[i + 1];
            N[np.suns[7]].sibling = np.sibling;
            np.cofm[0] = np.cofm[1] = np.cofm[2] = 0;
            np.mass = 0;
            np.hmax = 0;
            int sub = subnode(np, &pos[3 * (int64_t) p_toplace]);
            int child = np.suns[sub];
            if(N[child].noccupied < SHQ_NMAXCHILD) {
                attach(child, N[child].noccupied, p_toplace);
                N[child].noccupied++;
                break;
            }
            set_childtype(N[child], SHQ_NODE_NODE_TYPE);
            N[child].noccupied = NODEFULL;
            parent = child;
        }
        set_childtype(N[firstparent], SHQ_NODE_NODE_TYPE);
        N[firstparent].noccupied = NODEFULL;
        return true;
    }
    /* libgadget/forcetree.cpp:481-520 add_particle_to_tree */
    bool add(int i, int cur)
    {
        while(N[cur].noccupied >= NODEFULL) {
            int sub = subnode(N[cur], &pos[3 * (int64_t) i]);
            cur = N[cur].suns[sub];
        }
        int nocc = N[cur].noccupied;
        N[cur].noccupied++;
        if(nocc < SHQ_NMAXCHILD) {
            attach(cur, nocc, i);
            return true;
        }
        return new_layer(cur, i);
    }
    /* libgadget/forcetree.cpp:968-983 force_get_sibling */
    static int get_sibling(int sib, int j, const int *suns)
    {
        for(int jj = j + 1; jj < 8; jj++)
            if(suns[jj] >= 0)
                return suns[jj];
        return sib;
    }
    /* libgadget/forcetree.cpp:985-1005 force_update_particle_node */
    void update_particle_node(int no)
    {
        shq_node &nd = N[no];
        if(nd.mass > 0) {
            for(int j = 0; j < 3; j++)
                nd.cofm[j] /= nd.mass;
        } else {
            for(int j = 0; j < 3; j++)
                nd.cofm[j] = nd.center[j];
        }
    }
    /* libgadget/forcetree.cpp:1016-1103 force_update_node_recursive */
    void update_recursive(int no, int sib)
    {
        int *suns = N[no].suns;
        int jj = 0;
        for(int j = 0; j < 8; j++, jj++) {
            while(jj < 8 && !SHQ_NODE_TOPLEVEL(N[suns[jj]].flags) &&
                  childtype(N[suns[jj]]) == SHQ_PARTICLE_NODE_TYPE && N[suns[jj]].noccupied == 0)
                jj++;
            suns[j] = (jj < 8) ? suns[jj] : -1;
        }
        for(int j = 0; j < 8; j++) {
            int p = suns[j];
            if(p < 0)
                continue;
            int nextsib = get_sibling(sib, j, suns);
            N[p].sibling = nextsib;
            if(childtype(N[p]) == SHQ_PARTICLE_NODE_TYPE)
                update_particle_node(p);
            if(childtype(N[p]) == SHQ_NODE_NODE_TYPE)
                update_recursive(p, nextsib);
        }
        shq_node &nd = N[no];
        for(int j = 0; j < 8; j++) {
            int p = suns[j];
            if(p < 0)
                continue;
            nd.mass += N[p].mass;
            nd.cofm[0] += N[p].mass * N[p].cofm[0];
            nd.cofm[1] += N[p].mass * N[p].cofm[1];
            nd.cofm[2] += N[p].mass * N[p].cofm[2];
            if(N[p].hmax > nd.hmax)
                nd.hmax = N[p].hmax;
        }
        if(nd.mass > 0) {
            nd.cofm[0] /= nd.mass;
            nd.cofm[1] /= nd.mass;
            nd.cofm[2] /= nd.mass;
        }
    }
};

} // namespace

extern "C" int64_t orc_tree_build(const double *pos, const float *mass, const double *hsml,
                                  const int32_t *idx, int64_t n, int64_t numpart_total,
                                  double BoxSize, shq_node *nodes, int64_t maxnodes,
                                  int32_t *father)
{
    Builder b;
    b.pos = pos;
    b.mass = mass;
    b.hsml = hsml;
    b.firstnode = numpart_total;
    b.lastnode = numpart_total + maxnodes;
    b.N = nodes - b.firstnode;
    b.father = father;
    b.nnext = b.firstnode;
    /* root: libgadget/forcetree.cpp:655-680 force_tree_create_topnodes, single top leaf
     * (trivial_domain of tests/test_forcetree.cpp:294-314) */
    shq_node &root = b.N[b.nnext++];
    memset(&root, 0, sizeof(root));
    root.len = BoxSize * 1.001;
    for(int i = 0; i < 3; i++)
        root.center[i] = BoxSize / 2.;
    for(int i = 0; i < SHQ_NMAXCHILD; i++)
        root.suns[i] = -1;
    root.noccupied = 0;
    root.father = -1;
    root.sibling = -1;
    root.flags = 2u; /* TopLevel=1, InternalTopLevel=0, ChildType=PARTICLE */
    for(int64_t k = 0; k < n; k++) {
        int i = idx ? idx[k] : (int) k;
        if(!b.add(i, (int) b.firstnode))
            return -1;
    }
    /* moments: libgadget/forcetree.cpp:1118-1142 force_update_node_parallel (local top leaf) */
    root.flags |= 4u; /* DependsOnLocalMass */
    if(childtype(root) == SHQ_NODE_NODE_TYPE)
        b.update_recursive((int) b.firstnode, root.sibling);
    else
        b.update_particle_node((int) b.firstnode);
    return b.nnext - b.firstnode;
}

/* libgadget/gravity.h:48-60 GravShortTable::apply_short_range_window */
static inline int short_range_window(const shq_grav_params *p, double r, double *fac, double *pot)
{
    const double i = (r / p->cellsize / p->dx);
    size_t tabindex = (size_t) floor(i);
    if(tabindex >= SHQ_NGRAVTAB - 1)
        return 1;
    *fac *= (tabindex + 1 - i) * p->shortrange_table[tabindex] + (i - tabindex) * p->shortrange_table[tabindex + 1];
    *pot *= (tabindex + 1 - i) * p->shortrange_table_potential[tabindex] +
            (i - tabindex) * p->shortrange_table_potential[tabindex + 1];
    return 0;
}

/* libgadget/gravshort2.hpp:326-358 apply_accn */
extern "C" int orc_apply_accn(const double dx[3], double r2, double mass, const shq_grav_params *p,
                              double acc[3], double *pot)
{
    const double h = p->ForceSoftening;
    const double r = sqrt(r2);
    double fac = mass / (r2 * r);
    double facpot = -mass / r;
    if(r2 < h * h) {
        double wp;
        const double h3_inv = 1.0 / h / h / h;
        const double u = r / h;
        if(u < 0.5) {
            fac = mass * h3_inv * (10.666666666667 + u * u * (32.0 * u - 38.4));
            wp = -2.8 + u * u * (5.333333333333 + u * u * (6.4 * u - 9.6));
        } else {
            fac = mass * h3_inv *
                  (21.333333333333 - 48.0 * u + 38.4 * u * u - 10.666666666667 * u * u * u -
                   0.066666666667 / (u * u * u));
            wp = -3.2 + 0.066666666667 / u +
                 u * u * (10.666666666667 + u * (-16.0 + u * (9.6 - 2.133333333333 * u)));
        }
        facpot = mass / h * wp;
    }
    if(0 == short_range_window(p, r, &fac, &facpot)) {
        for(int i = 0; i < 3; i++)
            acc[i] += dx[i] * fac;
        *pot += facpot;
        return 1;
    }
    return 0;
}

/* libgadget/gravshort2.hpp:152-167 */
static inline int discard_node(double len, double r2, const double center[3], const double inpos[3],
                               double BoxSize, double rcut, double rcut2)
{
    if(r2 <= rcut2)
        return 0;
    const double eff_dist = rcut + 0.5 * len;
    for(int i = 0; i < 3; i++)
        if(fabs(orc_nearest(center[i] - inpos[i], BoxSize)) > eff_dist)
            return 1;
    return 0;
}

/* libgadget/gravshort2.hpp:172-193 */
static inline int open_node(double len, double mass, double r2, const double center[3],
                            const double inpos[3], double BoxSize, double aold, int TreeUseBH,
                            double BHOpeningAngle2)
{
    if((TreeUseBH == 0) && (mass * len * len > r2 * r2 * aold))
        return 1;
    double bhangle = len * len / r2;
    if(bhangle > BHOpeningAngle2)
        return 1;
    const double inside = 0.6 * len;
    if(fabs(orc_nearest(center[0] - inpos[0], BoxSize)) < inside &&
       fabs(orc_nearest(center[1] - inpos[1], BoxSize)) < inside &&
       fabs(orc_nearest(center[2] - inpos[2], BoxSize)) < inside)
        return 1;
    return 0;
}

/* libgadget/gravshort2.hpp:227-322 GravLocalTreeWalk::visit<TREEWALK_PRIMARY> */
extern "C" void orc_grav_walk(const shq_node *nodes, int64_t firstnode, const double *pos,
                              const float *mass, const double *oldacc, const int32_t *targets,
                              int64_t ntargets, const shq_grav_params *p, double *acc_out,
                              double *pot_out, int64_t *nint_out)
{
    const shq_node *N = nodes - firstnode;
    const double rcut = p->Rcut, rcut2 = rcut * rcut, Box = p->BoxSize;
#pragma omp parallel for schedule(dynamic, 64)
    for(int64_t t = 0; t < ntargets; t++) {
        const int64_t i = targets ? targets[t] : t;
        const double *inpos = &pos[3 * i];
        const double aold = p->ErrTolForceAcc * oldacc[i];
        double acc[3] = {0, 0, 0}, pot = 0;
        int64_t nint = 0;
        int no = (int) firstnode;
        while(no >= 0) {
            const shq_node *nop = &N[no];
            double dx[3];
            for(int k = 0; k < 3; k++)
                dx[k] = orc_nearest(nop->cofm[k] - inpos[k], Box);
            const double r2 = dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2];
            if(discard_node(nop->len, r2, nop->center, inpos, Box, rcut, rcut2)) {
                no = nop->sibling;
                continue;
            }
            if(!open_node(nop->len, nop->mass, r2, nop->center, inpos, Box, aold, p->TreeUseBH, p->BHOpeningAngle2)) {
                no = nop->sibling;
                orc_apply_accn(dx, r2, nop->mass, p, acc, &pot);
                nint++;
                continue;
            }
            const unsigned ct = SHQ_NODE_CHILDTYPE(nop->flags);
            if(ct == SHQ_PARTICLE_NODE_TYPE) {
                for(int c = 0; c < nop->noccupied; c++) {
                    const int pp = nop->suns[c];
                    for(int k = 0; k < 3; k++)
                        dx[k] = orc_nearest(pos[3 * (int64_t) pp + k] - inpos[k], Box);
                    const double rr2 = dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2];
                    orc_apply_accn(dx, rr2, mass[pp], p, acc, &pot);
                    nint++;
                }
                no = nop->sibling;
                continue;
            } else if(ct == SHQ_PSEUDO_NODE_TYPE) {
                no = nop->sibling;
                continue;
            }
            no = nop->suns[0];
        }
        acc_out[3 * t + 0] = acc[0];
        acc_out[3 * t + 1] = acc[1];
        acc_out[3 * t + 2] = acc[2];
        if(pot_out)
            pot_out[t] = pot;
        if(nint_out)
            nint_out[t] = nint;
    }
}

/* libgadget/gravshort2.hpp:227-322 GravLocalTreeWalk::visit<TREEWALK_GHOSTS>: an imported query (position and
 * OldAcc of a particle of another rank, treewalk2.h:742-812) walks the branches hanging off the top-level
 * nodes in its NodeList (NODELISTLENGTH = 4, -1 terminated): each branch from its start node until the walk
 * reaches another TopLevel node (gravshort2.hpp:258-261).  Raw sums, no postprocess: the exporting rank
 * reduces and post-processes them (ev_reduce_export_result). */
extern "C" void orc_grav_walk_secondary(const shq_node *nodes, int64_t firstnode, const double *pos, const float *mass,
                                        const double *qpos, const int32_t *qnodelist, const double *qoldacc, int64_t nq,
                                        const shq_grav_params *p, double *acc_out, double *pot_out, int64_t *nint_out)
{
    const shq_node *N = nodes - firstnode;
    const double rcut = p->Rcut, rcut2 = rcut * rcut, Box = p->BoxSize;
#pragma omp parallel for schedule(dynamic, 64)
    for(int64_t t = 0; t < nq; t++) {
        const double *inpos = &qpos[3 * t];
        const double aold = p->ErrTolForceAcc * qoldacc[t];
        double acc[3] = {0, 0, 0}, pot = 0;
        int64_t nint = 0;
        for(int listindex = 0; listindex < 4; listindex++) {
            int no = qnodelist[4 * t + listindex];
            const int startno = no;
            if(no < 0)
                break;
            while(no >= 0) {
                const shq_node *nop = &N[no];
                if(SHQ_NODE_TOPLEVEL(nop->flags) && no != startno)
                    break;
                double dx[3];
                for(int k = 0; k < 3; k++)
                    dx[k] = orc_nearest(nop->cofm[k] - inpos[k], Box);
                const double r2 = dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2];
                if(discard_node(nop->len, r2, nop->center, inpos, Box, rcut, rcut2)) {
                    no = nop->sibling;
                    continue;
                }
                if(!open_node(nop->len, nop->mass, r2, nop->center, inpos, Box, aold, p->TreeUseBH, p->BHOpeningAngle2)) {
                    no = nop->sibling;
                    orc_apply_accn(dx, r2, nop->mass, p, acc, &pot);
                    nint++;
                    continue;
                }
                const unsigned ct = SHQ_NODE_CHILDTYPE(nop->flags);
                if(ct == SHQ_PARTICLE_NODE_TYPE) {
                    for(int c = 0; c < nop->noccupied; c++) {
                        const int pp = nop->suns[c];
                        for(int k = 0; k < 3; k++)
                            dx[k] = orc_nearest(pos[3 * (int64_t) pp + k] - inpos[k], Box);
                        const double rr2 = dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2];
                        orc_apply_accn(dx, rr2, mass[pp], p, acc, &pot);
                        nint++;
                    }
                    no = nop->sibling;
                    continue;
                } else if(ct == SHQ_PSEUDO_NODE_TYPE) {
                    no = nop->sibling;
                    continue;
                }
                no = nop->suns[0];
            }
        }
        acc_out[3 * t + 0] = acc[0];
        acc_out[3 * t + 1] = acc[1];
        acc_out[3 * t + 2] = acc[2];
        if(pot_out)
            pot_out[t] = pot;
        if(nint_out)
            nint_out[t] = nint;
    }
}

/* libgadget/gravshort2.hpp:88-107 GravTreeOutput::postprocess */
extern "C" void orc_grav_postprocess(const float *mass, const int32_t *targets, int64_t ntargets,
                                     const shq_grav_params *p, int update_potential, double *acc,
                                     double *pot)
{
    for(int64_t t = 0; t < ntargets; t++) {
        const int64_t i = targets ? targets[t] : t;
        acc[3 * t + 0] *= p->G;
        acc[3 * t + 1] *= p->G;
        acc[3 * t + 2] *= p->G;
        if(update_potential && pot) {
            pot[t] += mass[i] / (p->ForceSoftening / 2.8);
            pot[t] -= 2.8372975 * pow(mass[i], 2.0 / 3) * p->cbrtrho0;
            pot[t] *= p->G;
        }
    }
}

/* tests/test_gravity.cpp:41-76 grav_force + :121-143 force_direct (mass generalised from the
 * test's unit mass). */
extern "C" void orc_force_direct(const double *pos, const float *mass, int64_t n, double BoxSize,
                                 double G, double h, int repeat, double *accn)
{
#pragma omp parallel for schedule(dynamic, 16)
    for(int64_t i = 0; i < n; i++) {
        double a[3] = {0, 0, 0};
        for(int xx = -repeat; xx <= repeat; xx++)
            for(int yy = -repeat; yy <= repeat; yy++)
                for(int zz = -repeat; zz <= repeat; zz++) {
                    const double off[3] = {BoxSize * xx, BoxSize * yy, BoxSize * zz};
                    for(int64_t j = 0; j < n; j++) {
                        double dist[3], r2 = 0;
                        for(int d = 0; d < 3; d++) {
                            dist[d] = off[d] + pos[3 * i + d] - pos[3 * j + d];
                            r2 += dist[d] * dist[d];
                        }
                        if(r2 == 0)
                            continue;
                        const double r = sqrt(r2);
                        double fac = 1 / (r2 * r);
                        if(r < h) {
                            double h_inv = 1.0 / h;
                            double h3_inv = h_inv * h_inv * h_inv;
                            double u = r * h_inv;
                            if(u < 0.5)
                                fac = 1. * h3_inv * (10.666666666667 + u * u * (32.0 * u - 38.4));
                            else
                                fac = 1. * h3_inv *
                                      (21.333333333333 - 48.0 * u + 38.4 * u * u -
                                       10.666666666667 * u * u * u - 0.066666666667 / (u * u * u));
                        }
                        for(int d = 0; d < 3; d++)
                            a[d] += -dist[d] * fac * G * mass[j];
                    }
                }
        accn[3 * i + 0] = a[0];
        accn[3 * i + 1] = a[1];
        accn[3 * i + 2] = a[2];
    }
}

extern "C" int orc_num_threads(void) { return omp_get_max_threads(); }
