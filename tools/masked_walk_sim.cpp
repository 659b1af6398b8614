/* masked_walk_sim.cpp — design probe (not product): EXACT per-target semantics walked by groups.
 * A group of up to 8 targets walks the union of its members' reference walks; every stack entry carries the mask of members
 * whose own walk reaches that node.  A cheap box test (T1) settles nodes on which all masked members provably agree; only the
 * rest get the per-member tests (T2).  Counts what a lanes-are-sources kernel would execute. */
#include <math.h>
#include <stdint.h>
#include <omp.h>
#include <vector>
#include "shenqi_hip.h"

static inline double nearest(double x, double L) { return (x > 0.5 * L) ? (x - L) : ((x < -0.5 * L) ? (x + L) : x); }

extern "C" void masked_walk_sim(const shq_node *nodes, int64_t firstnode, const double *pos, const double *oldacc, int64_t ntargets,
                                const int32_t *gsz, const shq_grav_params *p, int64_t *out /* per group start: T1, T2, list entries, pair evals, leaf-particle entries, exact interactions */)
{
    const shq_node *N = nodes - firstnode;
    const double rcut = p->Rcut, rcut2 = rcut * rcut, L = p->BoxSize;
    std::vector<int64_t> starts;
    for(int64_t t = 0; t < ntargets; t += gsz[t])
        starts.push_back(t);
#pragma omp parallel for schedule(dynamic, 16)
    for(int64_t gi = 0; gi < (int64_t) starts.size(); gi++) {
        const int64_t t0 = starts[gi];
        const int gs = gsz[t0];
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, amin = 1e300, amax = 0;
        double ao[8];
        for(int m = 0; m < gs; m++) {
            const int64_t i = t0 + m;
            for(int k = 0; k < 3; k++) {
                lo[k] = fmin(lo[k], pos[3 * i + k]);
                hi[k] = fmax(hi[k], pos[3 * i + k]);
            }
            ao[m] = p->ErrTolForceAcc * oldacc[i];
            amin = fmin(amin, ao[m]);
            amax = fmax(amax, ao[m]);
        }
        double c[3], h[3];
        for(int k = 0; k < 3; k++) {
            c[k] = 0.5 * (lo[k] + hi[k]);
            h[k] = 0.5 * (hi[k] - lo[k]);
        }
        struct Ent { int no; unsigned mask; };
        std::vector<Ent> st;
        st.push_back({(int) firstnode, (1u << gs) - 1});
        int64_t T1 = 0, T2 = 0, entries = 0, pairs = 0, pentries = 0, exact = 0;
        while(!st.empty()) {
            const Ent e = st.back();
            st.pop_back();
            const shq_node *nd = &N[e.no];
            T1++;
            double r2min = 0, r2max = 0, cminmax = 0, cmaxmax = 0;
            bool insidepossible = true, insideall = true;
            for(int k = 0; k < 3; k++) {
                const double d = fabs(nearest(nd->cofm[k] - c[k], L));
                const double dmin = fmax(d - h[k], 0.0), dmax = d + h[k];
                r2min += dmin * dmin;
                r2max += dmax * dmax;
                const double q = fabs(nearest(nd->center[k] - c[k], L));
                cminmax = fmax(cminmax, fmax(q - h[k], 0.0));
                cmaxmax = fmax(cmaxmax, q + h[k]);
                if(fmax(q - h[k], 0.0) >= 0.6 * nd->len)
                    insidepossible = false;
                if(q + h[k] >= 0.6 * nd->len)
                    insideall = false;
            }
            const double rcuthl = rcut + 0.5 * nd->len;
            const bool alldiscard = r2min > rcut2 && cminmax > rcuthl;
            if(alldiscard)
                continue;
            const bool nonediscards = r2max <= rcut2 || cmaxmax <= rcuthl;
            const double mlen2 = nd->mass * nd->len * nd->len, l2 = nd->len * nd->len;
            const bool allopen = (p->TreeUseBH == 0 && mlen2 > r2max * r2max * amax) || (l2 > r2max * p->BHOpeningAngle2) || insideall;
            const bool noneopen = !(p->TreeUseBH == 0 && mlen2 > r2min * r2min * amin) && !(l2 > r2min * p->BHOpeningAngle2) && !insidepossible;
            unsigned accm, openm;
            if(nonediscards && noneopen) {
                accm = e.mask;
                openm = 0;
            } else if(nonediscards && allopen) {
                accm = 0;
                openm = e.mask;
            } else {
                T2++;
                accm = openm = 0;
                for(int m = 0; m < gs; m++) {
                    if(!(e.mask >> m & 1))
                        continue;
                    const double *q = &pos[3 * (t0 + m)];
                    double r2 = 0, cmax = 0;
                    bool inside = true;
                    for(int k = 0; k < 3; k++) {
                        const double d = nearest(nd->cofm[k] - q[k], L);
                        r2 += d * d;
                        const double u = fabs(nearest(nd->center[k] - q[k], L));
                        cmax = fmax(cmax, u);
                        if(u >= 0.6 * nd->len)
                            inside = false;
                    }
                    if(r2 > rcut2 && cmax > rcuthl)
                        continue;
                    const bool open = (p->TreeUseBH == 0 && mlen2 > r2 * r2 * ao[m]) || (l2 > r2 * p->BHOpeningAngle2) || inside;
                    if(open)
                        openm |= 1u << m;
                    else
                        accm |= 1u << m;
                }
            }
            if(accm) {
                entries++;
                pairs += gs; /* every member slot is evaluated (masked lanes idle) unless the whole member loop is skipped */
                exact += __builtin_popcount(accm);
            }
            if(openm) {
                const unsigned ct = SHQ_NODE_CHILDTYPE(nd->flags);
                if(ct == SHQ_PARTICLE_NODE_TYPE) {
                    entries += nd->noccupied;
                    pentries += nd->noccupied;
                    pairs += (int64_t) nd->noccupied * gs;
                    exact += (int64_t) nd->noccupied * __builtin_popcount(openm);
                } else if(ct == SHQ_NODE_NODE_TYPE) {
                    int ch = nd->suns[0];
                    while(ch >= 0 && ch != nd->sibling) {
                        st.push_back({ch, openm});
                        ch = N[ch].sibling;
                    }
                }
            }
        }
        int64_t *o = out + 6 * t0;
        o[0] = T1; o[1] = T2; o[2] = entries; o[3] = pairs; o[4] = pentries; o[5] = exact;
    }
}
