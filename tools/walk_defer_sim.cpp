/* walk_defer_sim.cpp — design probe (not product): what a wave of the exact union walk (csrc/grav_walk.hip) would execute if
 * accepted sources were not evaluated at once but queued in a wave-shared ring of R source slots, every lane holding a bit
 * mask of the slots it must still evaluate, and drained in rounds in which every lane pops ITS oldest pending slot.
 * Counts evaluation rounds per wave for: the present kernel (a round per accepting node visit, a round per particle of an
 * opened leaf), a full drain whenever the ring is full, and a sliding ring whose base advances in steps of `gran` slots.
 * Build: g++ -O2 -fopenmp -shared -fPIC -Iinclude tools/walk_defer_sim.cpp -o build/libwalk_defer_sim.so */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <omp.h>
#include <algorithm>
#include <deque>
#include <vector>
#include "shenqi_hip.h"

static inline double nearest(double x, double L) { return (x > 0.5 * L) ? (x - L) : ((x < -0.5 * L) ? (x + L) : x); }

struct RingSim {
    int R, gran;       /* slots, base granularity (gran == 0: full drain when full) */
    int64_t base = 0;  /* oldest slot id still reserved */
    int64_t next = 0;  /* next slot id */
    int64_t rounds = 0, idle = 0, drains = 0;
    std::deque<int64_t> q[64];
    RingSim(int R_, int g_) : R(R_), gran(g_) {}
    bool any() const
    {
        for(int l = 0; l < 64; l++)
            if(!q[l].empty())
                return true;
        return false;
    }
    void round()
    {
        rounds++;
        for(int l = 0; l < 64; l++) {
            if(!q[l].empty())
                q[l].pop_front();
            else
                idle++;
        }
    }
    void advance()
    {
        int64_t oldest = next;
        for(int l = 0; l < 64; l++)
            if(!q[l].empty())
                oldest = std::min(oldest, q[l].front());
        if(gran > 0)
            base = std::max(base, oldest / gran * gran);
    }
    void push(uint64_t mask, int cnt)
    {
        /* make room for cnt slots */
        if(next + cnt - base > R) {
            drains++;
            if(gran == 0) {
                while(any())
                    round();
                base = next;
            } else {
                while(next + cnt - base > R) {
                    round();
                    advance();
                }
            }
        }
        for(int k = 0; k < cnt; k++) {
            for(int l = 0; l < 64; l++)
                if(mask >> l & 1)
                    q[l].push_back(next);
            next++;
        }
    }
    void finish()
    {
        while(any())
            round();
    }
};

/* out per sampled wave: [0] node visits, [1] accepting visits (rounds now), [2] leaf rounds now, [3] interactions (sum over lanes),
 * [4] max interactions of a lane, [5 + 3 c .. ] for config c: rounds, drains, idle lane-rounds */
extern "C" void walk_defer_sim(const shq_node *nodes, int64_t firstnode, const double *pos, const double *oldacc, int64_t ntargets,
                               const shq_grav_params *p, int64_t wave_stride, int nconf, const int32_t *confR, const int32_t *confG,
                               const int32_t *confT, int64_t *out, int nout)
{
    const shq_node *N = nodes - firstnode;
    const double rcut = p->Rcut, rcut2 = rcut * rcut, L = p->BoxSize;
    const int64_t nwaves = (ntargets + 63) / 64;
    const int64_t nsamp = (nwaves + wave_stride - 1) / wave_stride;
#pragma omp parallel for schedule(dynamic, 4)
    for(int64_t s = 0; s < nsamp; s++) {
        const int64_t w = s * wave_stride;
        int64_t *o = out + s * nout;
        memset(o, 0, sizeof(int64_t) * nout);
        const int64_t t0 = w * 64;
        const int nl = (int) std::min<int64_t>(64, ntargets - t0);
        int mynext[64];
        int64_t nint[64] = {};
        double ao[64];
        for(int l = 0; l < 64; l++) {
            mynext[l] = l < nl ? (int) firstnode : -2;
            ao[l] = l < nl ? p->ErrTolForceAcc * oldacc[t0 + l] : 0;
        }
        std::vector<RingSim> sims;
        std::vector<char> leafonly;
        std::vector<int64_t> immediate(nconf, 0);
        for(int c = 0; c < nconf; c++) {
            leafonly.push_back(confG[c] < 0);
            sims.emplace_back(confR[c], confG[c] < 0 ? -confG[c] - 1 : confG[c]);
        }
        int cur = (int) firstnode;
        while(cur >= 0) {
            const shq_node *nd = &N[cur];
            o[0]++;
            uint64_t accm = 0, openm = 0;
            const double rcuthl = rcut + 0.5 * nd->len, mlen2 = nd->mass * nd->len * nd->len, l2 = nd->len * nd->len;
            for(int l = 0; l < nl; l++) {
                if(mynext[l] != cur)
                    continue;
                const double *q = &pos[3 * (t0 + l)];
                double r2 = 0, cmax = 0;
                for(int k = 0; k < 3; k++) {
                    const double d = nearest(nd->cofm[k] - q[k], L);
                    r2 += d * d;
                    cmax = fmax(cmax, fabs(nearest(nd->center[k] - q[k], L)));
                }
                const bool discard = r2 > rcut2 && cmax > rcuthl;
                const bool open = (p->TreeUseBH == 0 && mlen2 > r2 * r2 * ao[l]) || (l2 > r2 * p->BHOpeningAngle2) || cmax < 0.6 * nd->len;
                if(discard)
                    continue;
                if(open)
                    openm |= 1ull << l;
                else
                    accm |= 1ull << l;
            }
            if(accm) {
                o[1]++;
                for(int l = 0; l < 64; l++)
                    if(accm >> l & 1)
                        nint[l]++;
                for(size_t c = 0; c < sims.size(); c++) {
                    if(leafonly[c] || __builtin_popcountll(accm) >= confT[c]) {
                        sims[c].rounds++; /* nodes evaluated at once */
                        immediate[c]++;
                    } else
                        sims[c].push(accm, 1);
                }
            }
            const unsigned ct = SHQ_NODE_CHILDTYPE(nd->flags);
            int next;
            if(ct == SHQ_PARTICLE_NODE_TYPE) {
                if(openm) {
                    o[2] += nd->noccupied;
                    for(int l = 0; l < 64; l++)
                        if(openm >> l & 1)
                            nint[l] += nd->noccupied;
                    for(auto &sm : sims)
                        sm.push(openm, nd->noccupied);
                }
                for(int l = 0; l < nl; l++)
                    if(mynext[l] == cur)
                        mynext[l] = nd->sibling;
                next = nd->sibling;
            } else if(ct == SHQ_PSEUDO_NODE_TYPE) {
                for(int l = 0; l < nl; l++)
                    if(mynext[l] == cur)
                        mynext[l] = nd->sibling;
                next = nd->sibling;
            } else {
                for(int l = 0; l < nl; l++)
                    if(mynext[l] == cur)
                        mynext[l] = (openm >> l & 1) ? nd->suns[0] : nd->sibling;
                next = openm ? nd->suns[0] : nd->sibling;
            }
            cur = next;
        }
        for(int l = 0; l < nl; l++) {
            o[3] += nint[l];
            o[4] = std::max(o[4], nint[l]);
        }
        for(int c = 0; c < nconf; c++) {
            sims[c].finish();
            o[5 + 3 * c] = sims[c].rounds;
            o[6 + 3 * c] = sims[c].drains;
            o[7 + 3 * c] = immediate[c];
        }
    }
}
