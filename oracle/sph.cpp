/* oracle/sph.cpp — CPU restatement of the reference SPH kernels, density (with the Hsml
 * iteration) and hydro force loops.  TEST INFRASTRUCTURE ONLY (see oracle.h).
 */
#include "oracle.h"
#include "sph_kernels.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <omp.h>

#define ORC_GAMMA (5.0 / 3.0) /* physconst.h:35-36 */
#define ORC_GAMMA_MINUS1 (ORC_GAMMA - 1)
#define ORC_MAXITER 400 /* treewalk2.h:21 */

/* libgadget/densitykernel.hpp:28-178; golden values tests/test_densitykernel.cpp:13-36 */
extern "C" int orc_density_kernel(int type, double H, double u, double eta, double out[5])
{
    if(type != 1 && type != 2 && type != 4)
        return 1;
    OrcKernel k(type, H);
    out[0] = OrcKernel::desnumngb(type, eta);
    out[1] = k.volume();
    out[2] = k.wk(u);
    out[3] = k.dwk(u);
    out[4] = k.dW(u);
    return 0;
}

namespace {

inline bool is_garbage(const orc_sph_arrays *a, int64_t i) { return a->flags && (a->flags[i] & 1); }
inline bool is_swallowed(const orc_sph_arrays *a, int64_t i) { return a->flags && (a->flags[i] & 2); }

/* KickFactorData::SPH_VelPred, density2.h:89-98 */
inline void vel_pred(const orc_sph_arrays *a, const shq_kick_factors *kf, int64_t i, double *v)
{
    const int pi = a->pi[i];
    for(int j = 0; j < 3; j++)
        v[j] = a->vel[3 * i + j] + kf->gravkicks[a->bin_grav[i]] * a->treeacc[3 * i + j] + a->gravpm[3 * i + j] * kf->FgravkickB +
               kf->hydrokicks[a->bin_hydro[i]] * a->hydroaccel[3 * pi + j];
}

/* KickFactorData::SPH_EntVarPred, density2.h:115-128 */
inline double entvar_pred(const orc_sph_arrays *a, const shq_kick_factors *kf, int64_t i)
{
    const int pi = a->pi[i];
    double e = a->entropy[pi] + a->dtentropy[pi] * kf->dloga_kick[a->bin_hydro[i]];
    if(e < 0.05 * a->entropy[pi])
        e = 0.05 * a->entropy[pi];
    if(e <= 0)
        return 0;
    return exp(1. / ORC_GAMMA * log(e));
}

/* cull_node<symmetric>, localtreewalk2.h:154-182 */
inline int cull_node(const double *Pos, double BoxSize, double Hsml, const shq_node *cur, bool symmetric)
{
    double dist = (symmetric ? fmax(cur->hmax, Hsml) : Hsml) + 0.5 * cur->len;
    double r2 = 0;
    for(int d = 0; d < 3; d++) {
        double dx = orc_nearest(cur->center[d] - Pos[d], BoxSize);
        if(dx > dist)
            return 0;
        if(dx < -dist)
            return 0;
        r2 += dx * dx;
    }
    const double FACT1 = 0.5 * (1.7320508075688772 - 1.0);
    dist += FACT1 * cur->len;
    if(r2 > dist * dist)
        return 0;
    return 1;
}

struct DensityResult { /* densitytree2.hpp:291-306 */
    double EgyRho = 0, DhsmlEgyDensity = 0, Rho = 0, DhsmlDensity = 0, Ngb = 0, Div = 0;
    double Rot[3] = {0, 0, 0}, GradRho[3] = {0, 0, 0};
};

} // namespace

/* set_init_hsml, density2.cpp:154-204 */
extern "C" void orc_set_init_hsml(const shq_node *nodes, int64_t firstnode, const int32_t *father, orc_sph_arrays *a,
                                  double MeanGasSeparation, double DesNumNgb)
{
    const shq_node *N = nodes - firstnode;
    for(int64_t i = 0; i < a->n; i++) {
        if(a->type[i] != 0 && a->type[i] != 5)
            continue;
        if(is_garbage(a, i))
            continue;
        int64_t no = i;
        do {
            int64_t p = (no >= firstnode) ? N[no].father : father[no];
            if(p < firstnode)
                break;
            no = p;
        } while(10 * DesNumNgb * a->mass[i] > N[no].mass);
        a->hsml[i] = MeanGasSeparation;
        if(no >= firstnode) {
            double testhsml = N[no].len * pow(3.0 / (4 * M_PI) * DesNumNgb * a->mass[i] / N[no].mass, 1.0 / 3);
            if(testhsml < 500. * MeanGasSeparation)
                a->hsml[i] = testhsml;
        }
    }
}

/* density() + TreeWalk::do_hsml_loop + DensityLocalTreeWalk::ngbiter + DensityOutput::postprocess:
 * density2.cpp:105-151, treewalk2.h:480-557, densitytree2.hpp:117-257,362-423. */
extern "C" int orc_density(shq_node *nodes, int64_t firstnode, const int32_t *father, orc_sph_arrays *a,
                           const int32_t *active, int64_t nactive, const shq_density_params *p, double *EntVarPred,
                           double *GradRho, int *niter_out, int64_t *nint_out)
{
    shq_node *N = nodes - firstnode;
    const int64_t n = a->n;
    const double Box = p->BoxSize;
    const int ktype = p->DensityKernelType;
    std::vector<double> Left(n, 0.0), Right(n, Box), NumNgb(n, 0.0), DhsmlDensityFactor(n, 0.0);
    std::vector<double> Rot(3 * (size_t) (a->nsph > 0 ? a->nsph : 1), 0.0);
    std::vector<double> evp_store;
    /* DensityPriv ctor, densitytree2.hpp:32-51: cache EntVarPred when (nearly) all are active */
    const bool cache = EntVarPred != nullptr;
    if(cache) {
        for(int64_t i = 0; i < n; i++)
            if(a->type[i] == 0 && !is_garbage(a, i))
                EntVarPred[a->pi[i]] = entvar_pred(a, &p->kf, i);
    }
    /* build_queue with DensityQuery::haswork, treewalk2.h:388-421, densitytree2.hpp:280-288 */
    std::vector<int32_t> queue;
    const int64_t nq0 = active ? nactive : n;
    for(int64_t k = 0; k < nq0; k++) {
        const int32_t i = active ? active[k] : (int32_t) k;
        if(is_garbage(a, i) || is_swallowed(a, i))
            continue;
        if(a->type[i] == 0 || a->type[i] == 5)
            queue.push_back(i);
    }
    int niter = 0;
    int64_t nint_total = 0;
    while(true) {
        const int64_t size = (int64_t) queue.size();
        std::vector<int32_t> todo(size, -1);
        int64_t nint_iter = 0;
#pragma omp parallel for schedule(dynamic, 32) reduction(+ : nint_iter)
        for(int64_t q = 0; q < size; q++) {
            const int32_t i = queue[q];
            const double *Pos = &a->pos[3 * (int64_t) i];
            const double Hsml = a->hsml[i];
            const int Type = a->type[i];
            double Vel[3];
            if(Type == 0)
                vel_pred(a, &p->kf, i, Vel);
            else
                for(int j = 0; j < 3; j++)
                    Vel[j] = a->vel[3 * (int64_t) i + j];
            OrcKernel kernel(ktype, Hsml);
            DensityResult out;
            /* LocalNgbTreeWalk::visit<PRIMARY>, localtreewalk2.h:378-437 (asymmetric, GASMASK) */
            int64_t no = firstnode;
            while(no >= 0) {
                const shq_node *cur = &N[no];
                if(0 == cull_node(Pos, Box, Hsml, cur, false)) {
                    no = cur->sibling;
                    continue;
                }
                const unsigned ct = SHQ_NODE_CHILDTYPE(cur->flags);
                if(ct == SHQ_PARTICLE_NODE_TYPE) {
                    for(int c = 0; c < cur->noccupied; c++) {
                        const int64_t other = cur->suns[c];
                        if(is_garbage(a, other))
                            continue;
                        if(!((1 << a->type[other]) & 1)) /* GASMASK */
                            continue;
                        nint_iter++;
                        /* ngbiter, densitytree2.hpp:362-423 */
                        double dist[3], r2 = 0;
                        for(int d = 0; d < 3; d++) {
                            dist[d] = orc_nearest(Pos[d] - a->pos[3 * other + d], Box);
                            r2 += dist[d] * dist[d];
                        }
                        if(r2 >= Hsml * Hsml)
                            continue;
                        const int pj = a->pi[other];
                        if(p->WindsDecouple && Type == 5 && a->delaytime && a->delaytime[pj] > 0)
                            continue;
                        const double r = sqrt(r2);
                        const double u = r / kernel.H;
                        const double wk = kernel.wk(u);
                        out.Ngb += wk * kernel.volume();
                        const double dwk = kernel.dwk(u);
                        const double mass_j = a->mass[other];
                        out.Rho += (mass_j * wk);
                        const double density_dW = kernel.dW(u);
                        out.DhsmlDensity += mass_j * density_dW;
                        double VelPred[3];
                        vel_pred(a, &p->kf, other, VelPred);
                        const double evp = cache ? EntVarPred[pj] : entvar_pred(a, &p->kf, other);
                        out.EgyRho += mass_j * evp * wk;
                        out.DhsmlEgyDensity += mass_j * evp * density_dW;
                        if(r <= 0)
                            continue;
                        const double fac = mass_j * dwk / r;
                        double dv[3];
                        for(int d = 0; d < 3; d++)
                            dv[d] = Vel[d] - VelPred[d];
                        out.Div += -fac * (dist[0] * dv[0] + dist[1] * dv[1] + dist[2] * dv[2]);
                        double rot[3] = {dv[1] * dist[2] - dv[2] * dist[1], dv[2] * dist[0] - dv[0] * dist[2],
                                         dv[0] * dist[1] - dv[1] * dist[0]};
                        for(int d = 0; d < 3; d++) {
                            out.Rot[d] += fac * rot[d];
                            out.GradRho[d] += fac * dist[d];
                        }
                    }
                    no = cur->sibling;
                    continue;
                } else if(ct == SHQ_PSEUDO_NODE_TYPE) {
                    no = cur->sibling;
                    continue;
                }
                no = cur->suns[0];
            }
            /* DensityResult::reduce<PRIMARY>, densitytree2.hpp:308-343 */
            NumNgb[i] = out.Ngb;
            DhsmlDensityFactor[i] = out.DhsmlDensity;
            const int pi = a->pi[i];
            if(Type == 0) {
                a->density[pi] = out.Rho;
                a->divvel[pi] = out.Div;
                Rot[3 * (size_t) pi + 0] = out.Rot[0];
                Rot[3 * (size_t) pi + 1] = out.Rot[1];
                Rot[3 * (size_t) pi + 2] = out.Rot[2];
                if(GradRho) {
                    GradRho[3 * (size_t) pi + 0] = out.GradRho[0];
                    GradRho[3 * (size_t) pi + 1] = out.GradRho[1];
                    GradRho[3 * (size_t) pi + 2] = out.GradRho[2];
                }
                a->egywtdensity[pi] = out.EgyRho;
                a->dhsmlegydensityfactor[pi] = out.DhsmlEgyDensity;
            } else if(Type == 5) {
                a->bh_density[pi] = out.Rho;
                a->bh_divvel[pi] = out.Div;
            }
        }
        nint_total += nint_iter;
        /* the host loop of do_hsml_loop, treewalk2.h:503-524, with DensityOutput::postprocess */
        for(int64_t q = 0; q < size; q++) {
            const int32_t i = queue[q];
            const int Type = a->type[i];
            const int pi = a->pi[i];
            int done = 0;
            double density = -1;
            if(Type == 0)
                density = a->density[pi];
            else if(Type == 5)
                density = a->bh_density[pi];
            double &DhsmlDens = DhsmlDensityFactor[i];
            DhsmlDens *= a->hsml[i] / (3 * density);
            DhsmlDens = 1 / (1 + DhsmlDens);
            if(p->update_hsml) {
                /* density_check_neighbours, densitytree2.hpp:177-257 */
                double desnumngb = p->DesNumNgb;
                if(p->BlackHoleOn && Type == 5)
                    desnumngb = p->DesNumNgbBH;
                if(NumNgb[i] < (desnumngb - p->MaxNumNgbDeviation) || (NumNgb[i] > (desnumngb + p->MaxNumNgbDeviation))) {
                    do {
                        if((Right[i] - Left[i]) < 1.0e-5 * Right[i]) {
                            a->hsml[i] = Right[i];
                            done = 1;
                            break;
                        }
                        if(NumNgb[i] < desnumngb)
                            Left[i] = a->hsml[i];
                        else
                            Right[i] = a->hsml[i];
                        if((Right[i] < Box && Left[i] > 0) || (a->hsml[i] * 1.26 > 0.99 * Box))
                            a->hsml[i] = cbrt(0.5 * (pow(Left[i], 3) + pow(Right[i], 3)));
                        else {
                            double DensFac = DhsmlDensityFactor[i];
                            double fac = 1.26;
                            if(NumNgb[i] > 0)
                                fac = 1 - (NumNgb[i] - desnumngb) / (3 * NumNgb[i]) * DensFac;
                            if(Right[i] > 0.99 * Box && Left[i] > 0)
                                if(DensFac <= 0 || fabs(NumNgb[i] - desnumngb) >= 0.5 * desnumngb || fac > 1.26)
                                    fac = 1.26;
                            if(Right[i] < 0.99 * Box && Left[i] == 0)
                                if(DensFac <= 0 || fac < 1. / 3)
                                    fac = 1. / 3;
                            a->hsml[i] *= fac;
                        }
                        if(Right[i] < p->MinGasHsml) {
                            a->hsml[i] = p->MinGasHsml;
                            done = 1;
                            break;
                        }
                        done = 0;
                    } while(0);
                } else {
                    if(a->hsml[i] < p->MinGasHsml)
                        a->hsml[i] = p->MinGasHsml;
                    done = 1;
                }
            }
            if(Type == 0) {
                if(p->DoEgyDensity) {
                    const double EntPred = cache ? EntVarPred[pi] : entvar_pred(a, &p->kf, i);
                    a->dhsmlegydensityfactor[pi] *= a->hsml[i] / (3 * a->egywtdensity[pi]);
                    a->dhsmlegydensityfactor[pi] *= -DhsmlDens;
                    a->egywtdensity[pi] /= EntPred;
                } else
                    a->dhsmlegydensityfactor[pi] = DhsmlDens;
                const double *Roti = &Rot[3 * (size_t) pi];
                a->curlvel[pi] = sqrt(Roti[0] * Roti[0] + Roti[1] * Roti[1] + Roti[2] * Roti[2]) / a->density[pi];
                a->divvel[pi] /= a->density[pi];
                a->dthsml[i] = (1.0 / 3) * a->divvel[pi] * a->hsml[i];
            } else if(Type == 5) {
                a->bh_divvel[pi] /= a->bh_density[pi];
                a->dthsml[i] = (1.0 / 3) * a->bh_divvel[pi] * a->hsml[i];
            }
            if(!p->update_hsml)
                done = 0; /* postprocess returns done = 0 when not updating; the loop exits anyway */
            if(0 == done)
                todo[q] = i;
            else if(p->update_hsml && Type == 0 && father) {
                /* update_tree_hmax_father, forcetree.cpp:1285-1313 (tree holds gas only: GASMASK) */
                shq_node *node = &N[father[i]];
                double newhmax = 0;
                for(int j = 0; j < 3; j++)
                    newhmax = fmax(newhmax, fabs(a->pos[3 * (int64_t) i + j] - node->center[j]) + a->hsml[i] - node->len / 2.);
                if(newhmax > node->hmax)
                    node->hmax = newhmax;
            }
        }
        std::vector<int32_t> redo;
        for(int64_t q = 0; q < size; q++)
            if(todo[q] >= 0)
                redo.push_back(todo[q]);
        niter++;
        queue.swap(redo);
        if(!p->update_hsml || queue.empty())
            break;
        if(niter > ORC_MAXITER) {
            if(niter_out)
                *niter_out = niter;
            return 1;
        }
    }
    if(niter_out)
        *niter_out = niter;
    if(nint_out)
        *nint_out = nint_total;
    return 0;
}

/* Recompute hmax of every node from the current Hsml of the gas/BH particles it holds and
 * propagate the maximum to the ancestors: the end state of force_tree_calc_moments
 * (forcetree.cpp:947-966,1080-1101) after density() has updated the leaves. */
extern "C" void orc_update_hmax(shq_node *nodes, int64_t firstnode, int64_t numnodes, const orc_sph_arrays *a)
{
    shq_node *N = nodes - firstnode;
    for(int64_t k = 0; k < numnodes; k++)
        nodes[k].hmax = 0;
    for(int64_t k = 0; k < numnodes; k++) {
        shq_node *nd = &nodes[k];
        if(SHQ_NODE_CHILDTYPE(nd->flags) != SHQ_PARTICLE_NODE_TYPE || nd->father < -1)
            continue;
        for(int c = 0; c < nd->noccupied; c++) {
            const int64_t pp = nd->suns[c];
            if(a->type[pp] != 0 && a->type[pp] != 5)
                continue;
            for(int j = 0; j < 3; j++)
                nd->hmax = fmax(nd->hmax, fabs(a->pos[3 * pp + j] - nd->center[j]) + a->hsml[pp] - nd->len / 2.);
        }
        double h = nd->hmax;
        int64_t f = nd->father;
        while(f >= firstnode && N[f].hmax < h) {
            N[f].hmax = h;
            f = N[f].father;
        }
    }
}

namespace {
/* hydratree2.hpp:21-34 */
inline double density_pred(double Density, double DivVel, double dtdrift)
{
    double d = Density - DivVel * Density * dtdrift;
    return (d >= 1e-6 * Density) ? d : 1e-6 * Density;
}
/* hydratree2.hpp:47-58 */
inline double pressure_predict(double eom, double evp)
{
    if(evp * eom <= 0)
        return 0;
    return exp(ORC_GAMMA * log(evp * eom));
}
} // namespace

/* hydro_force() + HydroQuery/HydroResult/HydroLocalTreeWalk::ngbiter/HydroOutput::postprocess:
 * hydra2.cpp:76-110, hydratree2.hpp:83-119,134-148,165-191,201-228,253-378. */
extern "C" void orc_hydro(const shq_node *nodes, int64_t firstnode, orc_sph_arrays *a, const int32_t *active,
                          int64_t nactive, const shq_hydro_params *p, const double *EntVarPred, int64_t *nint_out)
{
    const shq_node *N = nodes - firstnode;
    const int64_t n = a->n;
    const double Box = p->BoxSize;
    const int ktype = p->DensityKernelType;
    const bool DISPH = p->DensityIndependentSphOn != 0;
    /* HydroPriv ctor: PressurePred cache only when EntVarPred is given (hydratree2.hpp:103-118) */
    std::vector<double> PressurePred;
    if(EntVarPred) {
        PressurePred.assign((size_t) (a->nsph > 0 ? a->nsph : 1), 0.0);
        for(int64_t i = 0; i < n; i++) {
            if(a->type[i] != 0 || is_garbage(a, i))
                continue;
            const int pi = a->pi[i];
            const double eom = density_pred(DISPH ? a->egywtdensity[pi] : a->density[pi], a->divvel[pi], p->drifts[a->bin_hydro[i]]);
            PressurePred[pi] = pressure_predict(eom, EntVarPred[pi]);
        }
    }
    std::vector<int32_t> queue;
    const int64_t nq0 = active ? nactive : n;
    for(int64_t k = 0; k < nq0; k++) {
        const int32_t i = active ? active[k] : (int32_t) k;
        if(is_garbage(a, i) || is_swallowed(a, i))
            continue;
        if(a->type[i] == 0)
            queue.push_back(i);
    }
    const int64_t size = (int64_t) queue.size();
    std::vector<double> res(5 * (size_t) (size > 0 ? size : 1));
    int64_t nint = 0;
#pragma omp parallel for schedule(dynamic, 32) reduction(+ : nint)
    for(int64_t q = 0; q < size; q++) {
        const int32_t i = queue[q];
        const int pi = a->pi[i];
        const double *Pos = &a->pos[3 * (int64_t) i];
        /* HydroQuery ctor */
        const double iHsml = a->hsml[i], iMass = a->mass[i], iDensity = a->density[pi];
        const double iEgyRho = DISPH ? a->egywtdensity[pi] : a->density[pi];
        const double iDhsml = a->dhsmlegydensityfactor[pi];
        double iVel[3];
        vel_pred(a, &p->kf, i, iVel);
        const double iEntVarPred = EntVarPred ? EntVarPred[pi] : entvar_pred(a, &p->kf, i);
        const double eomdensity_i = DISPH ? a->egywtdensity[pi] : a->density[pi];
        const double iPressure = EntVarPred ? PressurePred[pi] : pressure_predict(eomdensity_i, iEntVarPred);
        const double soundspeed_q = sqrt(ORC_GAMMA * iPressure / eomdensity_i);
        const double iF1 = fabs(a->divvel[pi]) / (fabs(a->divvel[pi]) + a->curlvel[pi] + 0.0001 * soundspeed_q / iHsml / p->fac_mu);
        const int iBin = a->bin_hydro[i];
        /* HydroResult ctor / HydroLocalTreeWalk ctor */
        double Acc[3] = {0, 0, 0}, DtEntropy = 0;
        double MaxSignalVel = sqrt(ORC_GAMMA * iPressure / iEgyRho);
        const double soundspeed_i = sqrt(ORC_GAMMA * iPressure / iEgyRho);
        const double p_over_rho2_i = iPressure / (iEgyRho * iEgyRho);
        OrcKernel kernel_i(ktype, iHsml);
        int64_t no = firstnode;
        while(no >= 0) {
            const shq_node *cur = &N[no];
            if(0 == cull_node(Pos, Box, iHsml, cur, true)) {
                no = cur->sibling;
                continue;
            }
            const unsigned ct = SHQ_NODE_CHILDTYPE(cur->flags);
            if(ct == SHQ_PARTICLE_NODE_TYPE) {
                for(int c = 0; c < cur->noccupied; c++) {
                    const int64_t other = cur->suns[c];
                    if(is_garbage(a, other))
                        continue;
                    if(!((1 << a->type[other]) & 1))
                        continue;
                    nint++;
                    double dist[3], r2 = 0;
                    for(int d = 0; d < 3; d++) {
                        dist[d] = orc_nearest(Pos[d] - a->pos[3 * other + d], Box);
                        r2 += dist[d] * dist[d];
                    }
                    const double jHsml = a->hsml[other];
                    if(r2 <= 0 || !(r2 < iHsml * iHsml || r2 < jHsml * jHsml))
                        continue;
                    const int pj = a->pi[other];
                    if(a->delaytime && a->delaytime[pj] > 0)
                        continue;
                    OrcKernel kernel_j(ktype, jHsml);
                    double VelPred[3];
                    vel_pred(a, &p->kf, other, VelPred);
                    const double EVP = EntVarPred ? EntVarPred[pj] : entvar_pred(a, &p->kf, other);
                    const int bin = a->bin_hydro[other];
                    const double density_j = density_pred(a->density[pj], a->divvel[pj], p->drifts[bin]);
                    const double eomdensity_j = density_pred(DISPH ? a->egywtdensity[pj] : a->density[pj], a->divvel[pj], p->drifts[bin]);
                    const double Pressure_j = EntVarPred ? PressurePred[pj] : pressure_predict(eomdensity_j, EVP);
                    const double p_over_rho2_j = Pressure_j / (eomdensity_j * eomdensity_j);
                    const double soundspeed_j = sqrt(ORC_GAMMA * Pressure_j / eomdensity_j);
                    double vsig = soundspeed_i + soundspeed_j;
                    if(vsig > MaxSignalVel)
                        MaxSignalVel = vsig;
                    double dv[3];
                    for(int d = 0; d < 3; d++)
                        dv[d] = iVel[d] - VelPred[d];
                    const double vdotr = dist[0] * dv[0] + dist[1] * dv[1] + dist[2] * dv[2];
                    const double vdotr2 = vdotr + p->hubble_a2 * r2;
                    const double r = sqrt(r2);
                    const double dwk_i = kernel_i.dwk(r / kernel_i.H);
                    const double dwk_j = kernel_j.dwk(r / kernel_j.H);
                    double visc = 0;
                    if(vdotr2 < 0) {
                        const double mu_ij = p->fac_mu * vdotr2 / r;
                        const double rho_ij = 0.5 * (iDensity + density_j);
                        vsig = soundspeed_i + soundspeed_j - 3 * mu_ij;
                        if(vsig > MaxSignalVel)
                            MaxSignalVel = vsig;
                        const double f2 = fabs(a->divvel[pj]) / (fabs(a->divvel[pj]) + a->curlvel[pj] + 0.0001 * soundspeed_j / p->fac_mu / jHsml);
                        visc = 0.25 * p->ArtBulkViscConst * vsig * (-mu_ij) / rho_ij * (iF1 + f2);
                        double dloga = 2 * fmax(p->kf.dloga_for_bin[iBin], p->kf.dloga_for_bin[bin]);
                        if(dloga > 0 && (dwk_i + dwk_j) < 0) {
                            if((iMass + a->mass[other]) > 0)
                                visc = fmin(visc, 0.5 * p->fac_vsic_fix * vdotr2 / (0.5 * (iMass + a->mass[other]) * (dwk_i + dwk_j) * r * dloga));
                        }
                    }
                    const double mj = a->mass[other];
                    const double hfc_visc = 0.5 * mj * visc * (dwk_i + dwk_j) / r;
                    double hfc = hfc_visc;
                    double rr1 = 1, rr2 = 1;
                    if(DISPH) {
                        rr1 = 0, rr2 = 0;
                        hfc += mj * (dwk_i * p_over_rho2_i * EVP / iEntVarPred + dwk_j * p_over_rho2_j * iEntVarPred / EVP) / r;
                        if(p->DensityContrastLimit >= 0) {
                            rr1 = iEgyRho / iDensity;
                            rr2 = eomdensity_j / density_j;
                            if(p->DensityContrastLimit > 0) {
                                rr1 = fmin(rr1, p->DensityContrastLimit);
                                rr2 = fmin(rr2, p->DensityContrastLimit);
                            }
                        }
                    }
                    hfc += mj * (p_over_rho2_i * iDhsml * dwk_i * rr1 + p_over_rho2_j * a->dhsmlegydensityfactor[pj] * dwk_j * rr2) / r;
                    for(int d = 0; d < 3; d++)
                        Acc[d] += (-hfc * dist[d]);
                    DtEntropy += (0.5 * hfc_visc * vdotr2);
                }
                no = cur->sibling;
                continue;
            } else if(ct == SHQ_PSEUDO_NODE_TYPE) {
                no = cur->sibling;
                continue;
            }
            no = cur->suns[0];
        }
        res[5 * q + 0] = Acc[0];
        res[5 * q + 1] = Acc[1];
        res[5 * q + 2] = Acc[2];
        res[5 * q + 3] = DtEntropy;
        res[5 * q + 4] = MaxSignalVel;
    }
    /* reduce<PRIMARY> then postprocess (after all walks: neighbours read Density etc. unchanged) */
    for(int64_t q = 0; q < size; q++) {
        const int32_t i = queue[q];
        const int pi = a->pi[i];
        a->hydroaccel[3 * (size_t) pi + 0] = res[5 * q + 0];
        a->hydroaccel[3 * (size_t) pi + 1] = res[5 * q + 1];
        a->hydroaccel[3 * (size_t) pi + 2] = res[5 * q + 2];
        a->dtentropy[pi] = res[5 * q + 3];
        a->maxsignalvel[pi] = res[5 * q + 4];
        /* HydroOutput::postprocess, hydratree2.hpp:134-148 */
        a->dtentropy[pi] *= ORC_GAMMA_MINUS1 / (p->hubble_a2 * pow(a->density[pi], ORC_GAMMA_MINUS1));
        if(a->delaytime && a->delaytime[pi] > 0) {
            for(int k = 0; k < 3; k++)
                a->hydroaccel[3 * (size_t) pi + k] = 0;
            a->dtentropy[pi] = 0;
            /* winds_decoupled_hydro, winds.h:60-68 */
            double windspeed = p->WindSpeed * p->atime;
            const double fac_mu = pow(p->atime, 3 * (ORC_GAMMA - 1) / 2) / p->atime;
            windspeed *= fac_mu;
            double hsml_c = cbrt(p->WindFreeTravelDensThresh / a->density[pi]) * p->atime;
            a->maxsignalvel[pi] = hsml_c * fmax(2 * windspeed, a->maxsignalvel[pi]);
        }
    }
    if(nint_out)
        *nint_out = nint;
}

/* ---- stellar density ------------------------------------------------------------------------
 * stellar_density(): stellar_density2.cpp:306-341; StellarDensityLocalTreeWalk::ngbiter :219-254; stellareffhsml :38-54;
 * StellarDensityOutput::postprocess :113-154; ngb_narrow_down treewalk.c:1349-1406; loop treewalk2.h:480-557.
 * Targets: the star particles in `queue`; neighbours: gas (a->density by slot).  Per star NHSML = 10 trial radii are evaluated
 * in one walk; the search radius shrinks during the walk once an inner radius already holds enough neighbours.
 * StarVolumeSPH is indexed by particle (the reference indexes by star slot; the caller maps).  Returns 0, or 1 if MAXITER
 * is exceeded. */
#define ORC_NHSML 10

static double stellareffhsml(int i, double left, double right, double Hsml, double BoxSize)
{
    if(right > 0.99 * BoxSize)
        right = Hsml * ((1. + ORC_NHSML) / ORC_NHSML);
    if(left == 0)
        left = 0.1 * Hsml;
    const double rvol = pow(right, 3), lvol = pow(left, 3);
    return pow((1. * i + 1) / (1. * ORC_NHSML + 1) * (rvol - lvol) + lvol, 1. / 3);
}

/* treewalk.c:1349-1406 (note the integer desnumngb) */
static double ngb_narrow_down(double *right, double *left, const double *radius, const double *numNgb, int maxcmpt, int desnumngb,
                              int *closeidx, double BoxSize)
{
    int close = 0;
    double ngbdist = fabs(numNgb[0] - desnumngb);
    for(int j = 1; j < maxcmpt; j++) {
        const double newdist = fabs(numNgb[j] - desnumngb);
        if(newdist < ngbdist) {
            ngbdist = newdist;
            close = j;
        }
    }
    if(closeidx)
        *closeidx = close;
    for(int j = 0; j < maxcmpt; j++) {
        if(numNgb[j] < desnumngb)
            *left = radius[j];
        if(numNgb[j] > desnumngb) {
            *right = radius[j];
            break;
        }
    }
    double hsml = radius[close];
    if(*right > 0.99 * BoxSize) {
        double dngbdv = 0;
        if(maxcmpt > 1 && (radius[maxcmpt - 1] > radius[maxcmpt - 2]))
            dngbdv = (numNgb[maxcmpt - 1] - numNgb[maxcmpt - 2]) / (pow(radius[maxcmpt - 1], 3) - pow(radius[maxcmpt - 2], 3));
        double newhsml = 4 * hsml;
        if(dngbdv > 0) {
            const double dngb = (desnumngb - numNgb[maxcmpt - 1]);
            const double newvolume = pow(hsml, 3) + dngb / dngbdv;
            if(pow(newvolume, 1. / 3) < newhsml)
                newhsml = pow(newvolume, 1. / 3);
        }
        hsml = newhsml;
    }
    if(hsml > *right)
        hsml = *right;
    if(*left == 0) {
        double dngbdv = 0;
        if(radius[1] > radius[0])
            dngbdv = (numNgb[1] - numNgb[0]) / (pow(radius[1], 3) - pow(radius[0], 3));
        if(maxcmpt == 1 && radius[0] > 0)
            dngbdv = numNgb[0] / pow(radius[0], 3);
        if(dngbdv > 0) {
            const double dngb = desnumngb - numNgb[0];
            const double newvolume = pow(hsml, 3) + dngb / dngbdv;
            hsml = pow(newvolume, 1. / 3);
        }
    }
    if(hsml < *left)
        hsml = *left;
    return hsml;
}

extern "C" int orc_stellar_density(const shq_node *nodes, int64_t firstnode, orc_sph_arrays *a, const int32_t *queue, int64_t nqueue,
                                   double BoxSize, double DesNumNgb, double MaxNgbDeviation, int SPHWeighting, int ktype,
                                   double *StarVolumeSPH, int *niter_out, int64_t *nint_out)
{
    const shq_node *N = nodes - firstnode;
    const int64_t n = a->n;
    std::vector<double> Left(n, 0.0), Right(n, BoxSize);
    std::vector<int32_t> cur(queue, queue + nqueue);
    int niter = 0;
    int64_t nint = 0;
    while(true) {
        const int64_t size = (int64_t) cur.size();
        std::vector<int32_t> todo(size, -1);
        int64_t nint_iter = 0;
#pragma omp parallel for schedule(dynamic, 8) reduction(+ : nint_iter)
        for(int64_t q = 0; q < size; q++) {
            const int32_t i = cur[q];
            const double *Pos = &a->pos[3 * (int64_t) i];
            double HsmlEval[ORC_NHSML], Ngb[ORC_NHSML] = {0}, Vol[ORC_NHSML] = {0};
            for(int k = 0; k < ORC_NHSML; k++)
                HsmlEval[k] = stellareffhsml(k, Left[i], Right[i], a->hsml[i], BoxSize);
            double Hsml = HsmlEval[ORC_NHSML - 1];
            int maxcmpte = ORC_NHSML;
            int64_t no = firstnode;
            while(no >= 0) {
                const shq_node *c = &N[no];
                if(0 == cull_node(Pos, BoxSize, Hsml, c, false)) {
                    no = c->sibling;
                    continue;
                }
                const unsigned ct = SHQ_NODE_CHILDTYPE(c->flags);
                if(ct == SHQ_PARTICLE_NODE_TYPE) {
                    for(int s = 0; s < c->noccupied; s++) {
                        const int64_t other = c->suns[s];
                        if(is_garbage(a, other) || !((1 << a->type[other]) & 1))
                            continue;
                        nint_iter++;
                        /* ngbiter, stellar_density2.cpp:219-254 */
                        double r2 = 0;
                        for(int d = 0; d < 3; d++) {
                            const double dd = orc_nearest(Pos[d] - a->pos[3 * other + d], BoxSize);
                            r2 += dd * dd;
                        }
                        if(!(r2 < HsmlEval[maxcmpte - 1] * HsmlEval[maxcmpte - 1]))
                            continue;
                        const double r = sqrt(r2);
                        for(int k = 0; k < maxcmpte; k++) {
                            if(r2 < HsmlEval[k] * HsmlEval[k]) {
                                OrcKernel kernel(ktype, HsmlEval[k]);
                                const double wk = kernel.wk(r / HsmlEval[k]);
                                Ngb[k] += wk * kernel.volume();
                                double thisvol = a->mass[other] / a->density[a->pi[other]];
                                if(SPHWeighting)
                                    thisvol *= wk;
                                Vol[k] += thisvol;
                            }
                        }
                        for(int k = 0; k < ORC_NHSML; k++) {
                            if(Ngb[k] > DesNumNgb) {
                                maxcmpte = k + 1;
                                Hsml = HsmlEval[k];
                                break;
                            }
                        }
                    }
                    no = c->sibling;
                    continue;
                } else if(ct == SHQ_PSEUDO_NODE_TYPE) {
                    no = c->sibling;
                    continue;
                }
                no = c->suns[0];
            }
            /* StellarDensityOutput::postprocess, stellar_density2.cpp:113-154 */
            double evalhsml[ORC_NHSML];
            for(int k = 0; k < maxcmpte; k++)
                evalhsml[k] = HsmlEval[k];
            int close = 0;
            a->hsml[i] = ngb_narrow_down(&Right[i], &Left[i], evalhsml, Ngb, maxcmpte, (int) DesNumNgb, &close, BoxSize);
            const double numngb = Ngb[close];
            StarVolumeSPH[i] = Vol[close];
            if(numngb < (DesNumNgb - MaxNgbDeviation) || numngb > (DesNumNgb + MaxNgbDeviation)) {
                if((Right[i] - Left[i]) < 1.0e-4 * Left[i])
                    continue; /* very tight bounds: done */
                todo[q] = i;
            }
        }
        nint += nint_iter;
        niter++;
        std::vector<int32_t> next;
        for(int64_t q = 0; q < size; q++)
            if(todo[q] >= 0)
                next.push_back(todo[q]);
        cur.swap(next);
        if(cur.empty())
            break;
        if(niter > ORC_MAXITER)
            return 1;
    }
    if(niter_out)
        *niter_out = niter;
    if(nint_out)
        *nint_out = nint;
    return 0;
}

/* ---- black-hole velocity dispersion: blackhole_veldisp(), veldisp2.cpp:164-199 --------------------------------------
 * BHVelDispLocalTreeWalk::ngbiter (:126-144) over the dark-matter tree, DM_VelPred (density2.h:104-111), postprocess
 * (:49-63).  out[q][5] = NumDM, V1sumDM[3], V2sumDM for the q-th black hole of `queue`; vdisp[q] as BHP().VDisp would
 * be set (left untouched where the reference leaves it).  No reference fixture ("parity unpinned"): a dozen lines, checked
 * against brute-force sums in tests/test_oracle_cpu.py. */
extern "C" void orc_bh_veldisp(const shq_node *nodes, int64_t firstnode, const orc_sph_arrays *a, const int32_t *queue, int64_t nqueue,
                               double BoxSize, const shq_kick_factors *kf, double *out, double *vdisp)
{
    const shq_node *N = nodes - firstnode;
#pragma omp parallel for schedule(dynamic, 4)
    for(int64_t q = 0; q < nqueue; q++) {
        const int64_t i = queue[q];
        const double *Pos = &a->pos[3 * i];
        const double Hsml = a->hsml[i];
        double num = 0, v1[3] = {0, 0, 0}, v2 = 0;
        int64_t no = firstnode;
        while(no >= 0) {
            const shq_node *c = &N[no];
            if(0 == cull_node(Pos, BoxSize, Hsml, c, false)) {
                no = c->sibling;
                continue;
            }
            const unsigned ct = SHQ_NODE_CHILDTYPE(c->flags);
            if(ct == SHQ_PARTICLE_NODE_TYPE) {
                for(int s = 0; s < c->noccupied; s++) {
                    const int64_t other = c->suns[s];
                    if(is_garbage(a, other) || !((1 << a->type[other]) & 2)) /* DMMASK */
                        continue;
                    double r2 = 0;
                    for(int d = 0; d < 3; d++) {
                        const double dd = orc_nearest(Pos[d] - a->pos[3 * other + d], BoxSize);
                        r2 += dd * dd;
                    }
                    if(r2 <= 0 || !(r2 < Hsml * Hsml))
                        continue;
                    num += 1;
                    for(int d = 0; d < 3; d++) {
                        const double vp = a->vel[3 * other + d] + kf->gravkicks[a->bin_grav[other]] * a->treeacc[3 * other + d] +
                                          a->gravpm[3 * other + d] * kf->FgravkickB;
                        const double vel = vp - a->vel[3 * i + d];
                        v1[d] += vel;
                        v2 += vel * vel;
                    }
                }
                no = c->sibling;
                continue;
            } else if(ct == SHQ_PSEUDO_NODE_TYPE) {
                no = c->sibling;
                continue;
            }
            no = c->suns[0];
        }
        out[5 * q] = num;
        out[5 * q + 1] = v1[0];
        out[5 * q + 2] = v1[1];
        out[5 * q + 3] = v1[2];
        out[5 * q + 4] = v2;
        if(num > 0) {
            double vd = v2 / num;
            for(int d = 0; d < 3; d++)
                vd -= pow(v1[d] / num, 2);
            if(vd > 0)
                vdisp[q] = sqrt(vd / 3);
        }
    }
}

/* ---- wind velocity dispersion: winds_find_vel_disp(), veldisp2.cpp:203-528 (see the stellar density above for the loop) --
 * vdispeffdmradius :216-229, WindVDispLocalTreeWalk::ngbiter :440-479, WindVDispOutput::postprocess :285-320.
 * vdisp[q] for the q-th gas particle of `queue` (left untouched where the reference leaves SphP.VDisp alone), dmradius[q] the
 * converged radius.  "Parity unpinned": checked against brute-force sums in tests/test_oracle_cpu.py. */
#define ORC_NWINDHSML 5
#define ORC_NUMDMNGB 40
extern "C" int orc_wind_veldisp(const shq_node *nodes, int64_t firstnode, const orc_sph_arrays *a, const int32_t *queue, int64_t nqueue,
                                double BoxSize, const shq_kick_factors *kf, double Time, double hubble, double *vdisp, double *dmradius,
                                int *niter_out)
{
    const shq_node *N = nodes - firstnode;
    std::vector<double> Left(nqueue, 0.0), Right(nqueue, BoxSize), DM(nqueue);
    for(int64_t q = 0; q < nqueue; q++)
        DM[q] = a->hsml[queue[q]];
    std::vector<int64_t> cur(nqueue);
    for(int64_t q = 0; q < nqueue; q++)
        cur[q] = q;
    int niter = 0;
    while(!cur.empty()) {
        const int64_t size = (int64_t) cur.size();
        std::vector<int64_t> todo(size, -1);
#pragma omp parallel for schedule(dynamic, 8)
        for(int64_t c = 0; c < size; c++) {
            const int64_t q = cur[c], i = queue[q];
            const double *Pos = &a->pos[3 * i];
            double rad[ORC_NWINDHSML], num[ORC_NWINDHSML] = {0}, v1[ORC_NWINDHSML][3] = {{0}}, v2[ORC_NWINDHSML] = {0};
            {
                double right = Right[q], left = Left[q];
                if(right > 0.99 * BoxSize)
                    right = DM[q];
                if(left == 0)
                    left = 0.1 * DM[q];
                const double rvol = pow(right, 3), lvol = pow(left, 3);
                for(int k = 0; k < ORC_NWINDHSML; k++)
                    rad[k] = pow((1.0 * k + 1) / (1.0 * ORC_NWINDHSML + 1) * (rvol - lvol) + lvol, 1. / 3);
            }
            double Hsml = rad[ORC_NWINDHSML - 1];
            int maxcmpte = ORC_NWINDHSML;
            int64_t no = firstnode;
            while(no >= 0) {
                const shq_node *nd = &N[no];
                if(0 == cull_node(Pos, BoxSize, Hsml, nd, false)) {
                    no = nd->sibling;
                    continue;
                }
                const unsigned ct = SHQ_NODE_CHILDTYPE(nd->flags);
                if(ct == SHQ_PARTICLE_NODE_TYPE) {
                    for(int s = 0; s < nd->noccupied; s++) {
                        const int64_t other = nd->suns[s];
                        if(is_garbage(a, other) || !((1 << a->type[other]) & 2))
                            continue;
                        double dist[3], r2 = 0;
                        for(int d = 0; d < 3; d++) {
                            dist[d] = orc_nearest(Pos[d] - a->pos[3 * other + d], BoxSize);
                            r2 += dist[d] * dist[d];
                        }
                        if(r2 <= 0 || !(r2 < rad[ORC_NWINDHSML - 1] * rad[ORC_NWINDHSML - 1]))
                            continue;
                        const double r = sqrt(r2);
                        for(int k = 0; k < maxcmpte; k++) {
                            if(r < rad[k]) {
                                num[k] += 1;
                                for(int d = 0; d < 3; d++) {
                                    const double vp = a->vel[3 * other + d] + kf->gravkicks[a->bin_grav[other]] * a->treeacc[3 * other + d] +
                                                      a->gravpm[3 * other + d] * kf->FgravkickB;
                                    const double vel = vp - a->vel[3 * i + d] + hubble * Time * Time * dist[d];
                                    v1[k][d] += vel;
                                    v2[k] += vel * vel;
                                }
                            }
                        }
                        for(int k = 0; k < ORC_NWINDHSML; k++) {
                            if(num[k] > ORC_NUMDMNGB) {
                                maxcmpte = k + 1;
                                Hsml = rad[k];
                                break;
                            }
                        }
                    }
                    no = nd->sibling;
                    continue;
                } else if(ct == SHQ_PSEUDO_NODE_TYPE) {
                    no = nd->sibling;
                    continue;
                }
                no = nd->suns[0];
            }
            int close = 0;
            DM[q] = ngb_narrow_down(&Right[q], &Left[q], rad, num, maxcmpte, ORC_NUMDMNGB, &close, BoxSize);
            const double numngb = num[close];
            if((numngb >= (ORC_NUMDMNGB - 1) && numngb <= (ORC_NUMDMNGB + 1)) || (Right[q] - Left[q] < 5e-6 * Left[q])) {
                double vd = v2[close] / numngb;
                for(int d = 0; d < 3; d++)
                    vd -= pow(v1[close][d] / numngb, 2);
                if(vd > 0)
                    vdisp[q] = sqrt(vd / 3);
            } else
                todo[c] = q;
        }
        niter++;
        std::vector<int64_t> next;
        for(int64_t c = 0; c < size; c++)
            if(todo[c] >= 0)
                next.push_back(todo[c]);
        cur.swap(next);
        if(!cur.empty() && niter > ORC_MAXITER)
            return 1;
    }
    for(int64_t q = 0; q < nqueue; q++)
        dmradius[q] = DM[q];
    if(niter_out)
        *niter_out = niter;
    return 0;
}

/* ---- black-hole repositioning / dynamical-friction sums: bhdynfric.cpp:44-295 ------------------------------------------
 * BHReposLocalTreeWalk::ngbiter :160-174, BHDynFricLocalTreeWalk::ngbiter :193-224, raw results per black hole of `queue`:
 * out[q][12] = MinPot, MinPotPos[3], MinPotVel[3], SurroundingDensity, SurroundingVel[3], SurroundingRmsVel (before
 * postprocess).  potential: by particle.  "Parity unpinned": checked against brute force in tests/test_oracle_cpu.py. */
extern "C" void orc_bh_dynfric(const shq_node *nodes, int64_t firstnode, const orc_sph_arrays *a, const double *potential,
                               const int32_t *queue, int64_t nqueue, double BoxSize, const shq_kick_factors *kf, int method, int ktype,
                               int typemask, double *out)
{
    const shq_node *N = nodes - firstnode;
#pragma omp parallel for schedule(dynamic, 4)
    for(int64_t q = 0; q < nqueue; q++) {
        const int64_t i = queue[q];
        const double *Pos = &a->pos[3 * i];
        OrcKernel kernel(ktype, a->hsml[i]);
        const double H = kernel.H;
        double minpot = 1.0e29, mp[3] = {-1, -1, -1}, mv[3] = {0, 0, 0}, dens = 0, sv[3] = {0, 0, 0}, rms = 0;
        int64_t no = firstnode;
        while(no >= 0) {
            const shq_node *c = &N[no];
            if(0 == cull_node(Pos, BoxSize, a->hsml[i], c, false)) {
                no = c->sibling;
                continue;
            }
            const unsigned ct = SHQ_NODE_CHILDTYPE(c->flags);
            if(ct == SHQ_PARTICLE_NODE_TYPE) {
                for(int s = 0; s < c->noccupied; s++) {
                    const int64_t other = c->suns[s];
                    if(is_garbage(a, other) || !((1 << a->type[other]) & typemask))
                        continue;
                    double r2 = 0;
                    for(int d = 0; d < 3; d++) {
                        const double dd = orc_nearest(Pos[d] - a->pos[3 * other + d], BoxSize);
                        r2 += dd * dd;
                    }
                    if(r2 >= H * H)
                        continue;
                    if(potential[other] < minpot) {
                        minpot = potential[other];
                        for(int d = 0; d < 3; d++) {
                            mp[d] = a->pos[3 * other + d];
                            mv[d] = a->vel[3 * other + d];
                        }
                    }
                    if(method > 0 && (a->type[other] == 4 || (a->type[other] == 1 && method > 1))) {
                        const double wk = kernel.wk(sqrt(r2) / H);
                        dens += a->mass[other] * wk;
                        for(int d = 0; d < 3; d++) {
                            const double vp = a->vel[3 * other + d] + kf->gravkicks[a->bin_grav[other]] * a->treeacc[3 * other + d] +
                                              a->gravpm[3 * other + d] * kf->FgravkickB;
                            sv[d] += a->mass[other] * wk * vp;
                            rms += a->mass[other] * wk * pow(vp, 2);
                        }
                    }
                }
                no = c->sibling;
                continue;
            } else if(ct == SHQ_PSEUDO_NODE_TYPE) {
                no = c->sibling;
                continue;
            }
            no = c->suns[0];
        }
        double *o = out + 12 * q;
        o[0] = minpot;
        for(int d = 0; d < 3; d++) {
            o[1 + d] = mp[d];
            o[4 + d] = mv[d];
            o[8 + d] = sv[d];
        }
        o[7] = dens;
        o[11] = rms;
    }
}
