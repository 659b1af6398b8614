/* fof.hip — friends-of-friends groups of the resident particles, one task (libgadget/fof.cpp).
 *
 *   fof_label_primary, fofp_merge, update_root, fof_primary_ngbiter     fof.cpp:300-581
 *   fof_label_secondary, fof_secondary_ngbiter / _postprocess            fof.cpp:1142-1270
 *   treewalk_visit_ngbiter / _nolist_ngbiter neighbour test, cull_node   treewalk.c:946-961, 1183-1196, 990-1019
 *   fof_fof, fof_compile_base, fof_assign_grnr                            fof.cpp:159-256, 710-766, 1048-1096
 *   add_particle_to_group, fof_finish_group_properties                    fof.cpp:583-705
 *
 * Linking is a union-find over particle indices: one thread per primary particle walks the tree of the primary types with the
 * reference's cull, and every neighbour within the linking length is united with it — the larger root hooked under the smaller by
 * atomicCAS, as fofp_merge does, with path halving on the way.  The end state does not depend on the order of the unions: the
 * connected components, labelled with their smallest particle ID.  The walks are thread-per-particle on purpose: FOF runs at
 * snapshot times only and is bound by the ~20 neighbours per particle, not by the node fetches the wave-union walks were built for.
 * The catalogue is a radix sort by label, run detection by scans, and one thread per kept group for the property sums (in particle
 * index order, so the sums are reproducible; the reference's order is whatever its unstable sort left). */
#include "common.hpp"
#include <string.h>
#include <vector>
#include <algorithm>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

namespace {

inline unsigned nblk(long long n, int t = 256) { return (unsigned) ((n + t - 1) / t); }

__device__ __forceinline__ double nearest(double x, double Box) { return (x > 0.5 * Box) ? (x - Box) : ((x < -0.5 * Box) ? (x + Box) : x); }

struct FofTree {
    const NodeB *B;
    const NodeC *C;
    const double4 *posm_leaf;
    const int32_t *leaf_pidx;
    double Box;
    /* leaf-slot range [lo, hi) of every node's particles, through the build's numbering: lo[order[r]] for pool record r */
    const int32_t *order, *lo, *hi, *parent;
    double clique_len; /* a node this small or smaller holds particles that are all within the linking length of each other */
};

/* cull_node, asymmetric (treewalk.c:990-1019): true = the node cannot hold a neighbour */
__device__ __forceinline__ bool cull(const NodeB &b, const double4 p, double hsml, double Box)
{
#pragma clang fp contract(off)
    double dist = hsml + 0.5 * b.len;
    double r2 = 0;
    const double pos[3] = {p.x, p.y, p.z};
    for(int d = 0; d < 3; d++) {
        const double dx = nearest(b.center[d] - pos[d], Box);
        if(dx > dist || dx < -dist)
            return true;
        r2 += dx * dx;
    }
    dist += 0.366025403785 * b.len; /* FACT1, treewalk.c:19 */
    return r2 > dist * dist;
}

/* r2 as treewalk_visit_ngbiter accumulates it (the early exit does not change the outcome) */
__device__ __forceinline__ double ngb_r2(const double4 p, const double4 q, double Box)
{
#pragma clang fp contract(off)
    const double d0 = nearest(p.x - q.x, Box), d1 = nearest(p.y - q.y, Box), d2 = nearest(p.z - q.z, Box);
    double r2 = d0 * d0;
    r2 += d1 * d1;
    r2 += d2 * d2;
    return r2;
}

__device__ __forceinline__ int uf_find(int32_t *parent, int i)
{
    int r = i;
    while(true) {
        const int p = ((volatile int32_t *) parent)[r];
        if(p == r)
            return r;
        const int g = ((volatile int32_t *) parent)[p];
        if(g != p)
            atomicMin(&parent[r], g); /* path halving; parents only ever decrease (update_root's `t > r` condition) */
        r = p;
    }
}

/* fofp_merge, fof.cpp:481-542: the higher root becomes a child of the lower */
__device__ void uf_unite(int32_t *parent, int a, int b)
{
    while(true) {
        int h1 = uf_find(parent, a), h2 = uf_find(parent, b);
        if(h1 == h2)
            return;
        if(h1 > h2) {
            const int t = h1;
            h1 = h2;
            h2 = t;
        }
        if(atomicCAS(&parent[h2], h2, h1) == h2)
            return;
    }
}

/* Nodes whose cube diagonal is at most the linking length are cliques: every particle inside is a friend of every other.
 * fof_clique_kernel unites each top-most clique node's particles once; the walk then treats such a node as one unit — the first
 * of its particles found within the linking length links the target to all of them — which is what keeps the caustic of the
 * S-cluster (thousands of friends per particle) from costing a pair test per friend. */
__global__ void fof_clique_kernel(long long nt, FofTree t, const int32_t *pfather, const double4 *cen, int32_t *parent)
{
    const long long k = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= nt)
        return;
    const int i = t.leaf_pidx[k];
    int bn = t.order[pfather[i]]; /* the particle's leaf, in the build's numbering */
    if(!(cen[bn].w <= t.clique_len))
        return;
    for(int pa = t.parent[bn]; pa >= 0 && cen[pa].w <= t.clique_len; pa = t.parent[bn])
        bn = pa; /* the top-most clique node above the particle */
    const int first = t.leaf_pidx[t.lo[bn]];
    if(first != i)
        uf_unite(parent, first, i);
}

__global__ __launch_bounds__(256) void fof_link_kernel(long long nt, FofTree t, double linkl, int32_t *parent, int nn)
{
    const long long k = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= nt)
        return;
    const int i = t.leaf_pidx[k];
    const double4 p = t.posm_leaf[k];
    const double h2 = linkl * linkl;
    int no = 0;
    int myroot = i; /* some member of the target's set, as close to its root as the last look found */
    int cend = -1, cskip = -1; /* inside a clique node: pool records [.., cend) are its sub-tree, cskip is where the walk goes on after it */
    while(no >= 0) {
        if(no >= cend)
            cend = -1;
        const NodeC c = t.C[no];
        const NodeB b = t.B[no];
        if(c.type == SHQ_PSEUDO_NODE_TYPE || cull(b, p, linkl, t.Box)) {
            no = c.sibling;
            continue;
        }
        if(cend < 0 && b.len <= t.clique_len) { /* a clique: ONE friend inside is enough; its sub-tree is searched, not listed */
            /* ... and none is needed when the target already sits in the clique's set: sets only ever merge, so the sub-tree can be
             * passed over (in the caustic of the S-cluster nearly every clique a particle meets is in its group already: two or three
             * dependent loads instead of leaf records, positions and a unite) */
            myroot = uf_find(parent, myroot);
            if(uf_find(parent, t.leaf_pidx[t.lo[t.order[no]]]) == myroot) {
                no = c.sibling;
                continue;
            }
            cend = c.sibling >= 0 ? c.sibling : nn;
            cskip = c.sibling;
        }
        if(c.type == SHQ_PARTICLE_NODE_TYPE) {
            bool hit = false;
            for(int s = c.child; s < c.child + c.count; s++) {
                const int j = t.leaf_pidx[s];
                /* the neighbour test of treewalk_visit_ngbiter, treewalk.c:946-961; outside cliques a pair is united by its lower
                 * index (fof_primary_ngbiter: lv->target <= other) */
                if((cend >= 0 || i <= j) && ngb_r2(p, t.posm_leaf[s], t.Box) <= h2) {
                    uf_unite(parent, i, j);
                    if(cend >= 0) {
                        hit = true;
                        break;
                    }
                }
            }
            if(hit) {
                no = cskip;
                cend = -1;
            } else
                no = c.sibling;
        } else
            no = c.child;
    }
}

__global__ void fof_init_kernel(long long n, int32_t *parent, const unsigned long long *ids, unsigned long long *minid, unsigned long long *label)
{
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n)
        return;
    parent[i] = (int32_t) i;
    minid[i] = ids[i];
    label[i] = ids[i];
}

__global__ void fof_minid_kernel(long long nt, const int32_t *leaf_pidx, int32_t *parent, const unsigned long long *ids, unsigned long long *minid)
{
    const long long k = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= nt)
        return;
    const int i = leaf_pidx[k];
    const int r = uf_find(parent, i);
    if(r != i && ids[i] < ((volatile unsigned long long *) minid)[r]) /* a group of millions has one root: only improvements go to the atomic */
        atomicMin(&minid[r], ids[i]);
}

__global__ void fof_label_kernel(long long nt, const int32_t *leaf_pidx, int32_t *parent, const unsigned long long *minid, unsigned long long *label)
{
    const long long k = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= nt)
        return;
    const int i = leaf_pidx[k];
    label[i] = minid[uf_find(parent, i)];
}

/* fof_label_secondary: the whole radius loop of one particle in one thread (the radii of different particles are independent) */
__global__ __launch_bounds__(256) void fof_secondary_kernel(long long n, FofTree t, const double4 *posm, const uint8_t *pflags, const double *hsml,
                                                            double linkl, int secondary_mask, const unsigned long long *label_in,
                                                            unsigned long long *label_out, unsigned long long *nattached)
{
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n)
        return;
    const unsigned f = pflags[i];
    const int type = f >> 4;
    if((f & 3u) || !((1 << type) & secondary_mask))
        return;
    const double4 p = posm[i];
    float h = (float) (0.4 * linkl);
    if((type == 0 || type == 4 || type == 5) && hsml && (double) h < 0.5 * hsml[i])
        h = (float) (0.5 * hsml[i]);
    while(true) {
        double cur = (double) h;         /* iter->base.Hsml */
        double distance = 1e29;          /* O->Distance = LARGE */
        int other = -1;
        int no = 0;
        while(no >= 0) {
            const NodeC c = t.C[no];
            if(c.type == SHQ_PSEUDO_NODE_TYPE || cull(t.B[no], p, cur, t.Box)) {
                no = c.sibling;
                continue;
            }
            if(c.type == SHQ_PARTICLE_NODE_TYPE) {
                for(int s = c.child; s < c.child + c.count; s++) {
                    const double r2 = ngb_r2(p, t.posm_leaf[s], t.Box);
                    if(r2 > cur * cur)
                        continue;
                    const double r = sqrt(r2);
                    if(r < distance) {
                        distance = r;
                        other = t.leaf_pidx[s];
                    }
                    cur = r; /* fof_secondary_ngbiter: no need to look further than this neighbour */
                }
                no = c.sibling;
            } else
                no = c.child;
        }
        if(other >= 0) {
            label_out[i] = label_in[other];
            atomicAdd(nattached, 1ull);
            return;
        }
        if((double) h < 4 * linkl) /* fof_secondary_postprocess, fof.cpp:1176-1191 */
            h *= 2.0f;
        else
            return;
    }
}

__global__ void fof_iota_kernel(long long n, int32_t *v)
{
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n)
        v[i] = (int32_t) i;
}

__global__ void fof_flag_kernel(long long n, const unsigned long long *keys, int32_t *flags)
{
    const long long k = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(k < n)
        flags[k] = (k == 0 || keys[k] != keys[k - 1]) ? 1 : 0;
}

__global__ void fof_runstart_kernel(long long n, const int32_t *flags, const int32_t *runid_incl, int32_t *runstart, long long nruns)
{
    const long long k = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(k < n && flags[k])
        runstart[runid_incl[k] - 1] = (int32_t) k;
    if(k == 0)
        runstart[nruns] = (int32_t) n;
}

__global__ void fof_keep_kernel(long long nruns, const int32_t *runstart, int minlength, int32_t *keep)
{
    const long long r = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(r < nruns)
        keep[r] = (runstart[r + 1] - runstart[r] >= minlength) ? 1 : 0;
}

__global__ void fof_groups_kernel(long long nruns, const int32_t *runstart, const int32_t *keep, const int32_t *keptidx_excl, int32_t *gstart,
                                  int32_t *glen, unsigned int *lenkey, int32_t *gval)
{
    const long long r = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(r >= nruns || !keep[r])
        return;
    const int g = keptidx_excl[r];
    gstart[g] = runstart[r];
    glen[g] = runstart[r + 1] - runstart[r];
    lenkey[g] = 0xffffffffu - (unsigned) glen[g]; /* UINT64_MAX - Length, fof.cpp:1394 */
    gval[g] = g;
}

__global__ void fof_grnr_kernel(long long ng, const int32_t *order, int32_t *grnr)
{
    const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(t < ng)
        grnr[order[t]] = (int32_t) (t + 1);
}

__global__ void fof_partgrnr_kernel(long long n, const int32_t *idx, const int32_t *runid_incl, const int32_t *keep, const int32_t *keptidx_excl,
                                    const int32_t *grnr, int32_t *part_grnr)
{
    const long long k = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= n)
        return;
    const int r = runid_incl[k] - 1;
    part_grnr[idx[k]] = keep[r] ? grnr[keptidx_excl[r]] : -1;
}

struct FofSums {
    int Length;
    int LenType[6];
    double MassType[6], Mass, CM[3], Vel[3], Imom[3][3], Jmom[3], MaxDens;
    int seed_index;
};

struct FofPropArgs {
    const int32_t *gstart, *glen, *grnr, *idx;
    const unsigned long long *keys;
    const double4 *posm;
    const double *vel;
    const uint8_t *pflags;
    const double *density, *delaytime;
    int winds_decouple;
    double Box;
    shq_fof_group *out;
};

/* add_particle_to_group (fof.cpp:583-655) over the sorted positions [k0, k1) */
__device__ void fof_accumulate(FofSums &G, int k0, int k1, const double first[3], const FofPropArgs &a)
{
#pragma clang fp contract(off)
    for(int k = k0; k < k1; k++) {
        const int i = a.idx[k];
        const double4 q = a.posm[i];
        const unsigned type = a.pflags[i] >> 4;
        const double m = q.w;
        G.Length++;
        G.Mass += m;
        if(type < 6) {
            G.LenType[type]++;
            G.MassType[type] += m;
        }
        if(type == 0 && a.density && !(a.winds_decouple && a.delaytime && a.delaytime[i] > 0))
            if(a.density[i] > G.MaxDens) {
                G.MaxDens = a.density[i];
                G.seed_index = i;
            }
        const double pos[3] = {q.x, q.y, q.z};
        double rel[3], xyz[3], v[3], jm[3];
        for(int d = 0; d < 3; d++) {
            rel[d] = nearest(pos[d] - first[d], a.Box);
            xyz[d] = rel[d] + first[d];
            v[d] = a.vel ? a.vel[3 * (long long) i + d] : 0.0;
        }
        jm[0] = rel[1] * v[2] - v[1] * rel[2]; /* crossproduct, densitykernel.h:63-75 */
        jm[1] = rel[2] * v[0] - v[2] * rel[0];
        jm[2] = rel[0] * v[1] - v[0] * rel[1];
        for(int d1 = 0; d1 < 3; d1++) {
            G.CM[d1] += m * xyz[d1];
            G.Vel[d1] += m * v[d1];
            G.Jmom[d1] += m * jm[d1];
            for(int d2 = 0; d2 < 3; d2++)
                G.Imom[d1][d2] += m * rel[d1] * rel[d2];
        }
    }
}

/* Group::reduce, fof.h:85-119 */
__device__ void fof_reduce(FofSums &G, const FofSums &s)
{
#pragma clang fp contract(off)
    G.Length += s.Length;
    G.Mass += s.Mass;
    for(int j = 0; j < 6; j++) {
        G.LenType[j] += s.LenType[j];
        G.MassType[j] += s.MassType[j];
    }
    if(s.MaxDens > G.MaxDens) {
        G.MaxDens = s.MaxDens;
        G.seed_index = s.seed_index;
    }
    for(int d1 = 0; d1 < 3; d1++) {
        G.CM[d1] += s.CM[d1];
        G.Vel[d1] += s.Vel[d1];
        G.Jmom[d1] += s.Jmom[d1];
        for(int d2 = 0; d2 < 3; d2++)
            G.Imom[d1][d2] += s.Imom[d1][d2];
    }
}

/* fof_finish_group_properties (fof.cpp:657-705) and the output record */
__device__ void fof_finish(const FofSums &S, long long g, const double first[3], const float firstf[3], const FofPropArgs &a)
{
#pragma clang fp contract(off)
    shq_fof_group G;
    memset(&G, 0, sizeof(G));
    G.MinID = a.keys[a.gstart[g]];
    G.GrNr = a.grnr[g];
    G.Length = S.Length;
    G.seed_index = S.seed_index;
    G.MaxDens = S.MaxDens;
    G.Mass = S.Mass;
    for(int j = 0; j < 6; j++) {
        G.LenType[j] = S.LenType[j];
        G.MassType[j] = S.MassType[j];
    }
    for(int d = 0; d < 3; d++) {
        G.FirstPos[d] = firstf[d];
        G.CM[d] = S.CM[d];
        G.Vel[d] = S.Vel[d];
        G.Jmom[d] = S.Jmom[d];
        for(int d2 = 0; d2 < 3; d2++)
            G.Imom[d][d2] = S.Imom[d][d2];
    }
    double cm[3], rel[3], vcm[3], jcm[3];
    for(int d = 0; d < 3; d++) {
        G.Vel[d] /= G.Mass;
        vcm[d] = G.Vel[d];
        cm[d] = G.CM[d] / G.Mass;
        rel[d] = nearest(cm[d] - first[d], a.Box);
        int guard = 0;
        while(cm[d] >= a.Box && guard++ < 64) /* fof_periodic_wrap */
            cm[d] -= a.Box;
        while(cm[d] < 0 && guard++ < 64)
            cm[d] += a.Box;
        G.CM[d] = cm[d];
    }
    jcm[0] = rel[1] * vcm[2] - vcm[1] * rel[2];
    jcm[1] = rel[2] * vcm[0] - vcm[2] * rel[0];
    jcm[2] = rel[0] * vcm[1] - vcm[0] * rel[1];
    for(int d = 0; d < 3; d++)
        G.Jmom[d] -= jcm[d] * G.Mass;
    for(int d1 = 0; d1 < 3; d1++)
        for(int d2 = 0; d2 < 3; d2++) {
            const double diff = rel[d1] * rel[d2];
            G.Imom[d1][d2] -= G.Mass * diff;
        }
    a.out[g] = G;
}

constexpr int FOF_BIG = 4096; /* groups longer than this are summed by a workgroup */

__device__ __forceinline__ void fof_first(long long g, const FofPropArgs &a, double first[3], float firstf[3])
{
    const double4 q = a.posm[a.idx[a.gstart[g]]];
    firstf[0] = (float) q.x;
    firstf[1] = (float) q.y;
    firstf[2] = (float) q.z;
    for(int d = 0; d < 3; d++)
        first[d] = (double) firstf[d];
}

/* groups up to FOF_BIG members: one thread each, members in sorted (= particle index) order — the reference's loop, fof.cpp:843-850 */
__global__ __launch_bounds__(64) void fof_props_kernel(long long ng, const FofPropArgs a)
{
    const long long g = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(g >= ng || a.glen[g] > FOF_BIG)
        return;
    FofSums S;
    memset(&S, 0, sizeof(S));
    S.seed_index = -1;
    double first[3];
    float firstf[3];
    fof_first(g, a, first, firstf);
    fof_accumulate(S, a.gstart[g], a.gstart[g] + a.glen[g], first, a);
    fof_finish(S, g, first, firstf, a);
}

/* longer groups: FOF_SPLIT workgroups of 256 threads sum FOF_SPLIT x 256 consecutive slices of the member list; the partial groups
 * are added in slice order with Group::reduce — the way the reference adds the parts of a group that lies on several tasks — first
 * inside a workgroup, then (second kernel) across the workgroups.  Deterministic; differs from the one-thread order by rounding only. */
constexpr int FOF_SPLIT = 64;

__global__ __launch_bounds__(256) void fof_props_big_kernel(int nbig, const int32_t *biglist, const FofPropArgs a, FofSums *partial)
{
    __shared__ FofSums part[256];
    const int b = blockIdx.x / FOF_SPLIT, piece = blockIdx.x % FOF_SPLIT;
    const long long g = biglist[b];
    double first[3];
    float firstf[3];
    fof_first(g, a, first, firstf);
    const long long s0 = a.gstart[g], len = a.glen[g];
    const long long chunk = (len + (long long) FOF_SPLIT * 256 - 1) / ((long long) FOF_SPLIT * 256);
    const long long slice = (long long) piece * 256 + threadIdx.x;
    const long long k0 = s0 + chunk * slice, k1 = k0 + chunk < s0 + len ? k0 + chunk : s0 + len;
    FofSums S;
    memset(&S, 0, sizeof(S));
    S.seed_index = -1;
    if(k0 < k1)
        fof_accumulate(S, (int) k0, (int) k1, first, a);
    part[threadIdx.x] = S;
    __syncthreads();
    if(threadIdx.x == 0) {
        FofSums T = part[0];
        for(int t = 1; t < 256; t++)
            fof_reduce(T, part[t]);
        partial[blockIdx.x] = T;
    }
}

__global__ void fof_props_big_finish_kernel(int nbig, const int32_t *biglist, const FofPropArgs a, const FofSums *partial)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if(b >= nbig)
        return;
    const long long g = biglist[b];
    double first[3];
    float firstf[3];
    fof_first(g, a, first, firstf);
    FofSums T = partial[(size_t) b * FOF_SPLIT];
    for(int p = 1; p < FOF_SPLIT; p++)
        fof_reduce(T, partial[(size_t) b * FOF_SPLIT + p]);
    fof_finish(T, g, first, firstf, a);
}

__global__ void fof_bigflag_kernel(long long ng, const int32_t *glen, int32_t *flag)
{
    const long long g = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(g < ng)
        flag[g] = glen[g] > FOF_BIG ? 1 : 0;
}
__global__ void fof_biglist_kernel(long long ng, const int32_t *flag, const int32_t *excl, int32_t *list)
{
    const long long g = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(g < ng && flag[g])
        list[excl[g]] = (int32_t) g;
}

/* members of the kept groups, group after group */
__global__ void fof_member_offsets_kernel(long long ng, const int32_t *glen, const long long *off_excl, shq_fof_group *groups)
{
    const long long g = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(g < ng)
        groups[g].first_member = off_excl[g];
}
__global__ void fof_glen64_kernel(long long ng, const int32_t *glen, long long *out)
{
    const long long g = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(g < ng)
        out[g] = glen[g];
}
__global__ void fof_members_kernel(long long n, const int32_t *idx, const int32_t *runid_incl, const int32_t *keep, const int32_t *keptidx_excl,
                                   const int32_t *runstart, const long long *off_excl, int32_t *members)
{
    const long long k = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= n)
        return;
    const int r = runid_incl[k] - 1;
    if(keep[r])
        members[off_excl[keptidx_excl[r]] + (k - runstart[r])] = idx[k];
}

template <typename T> int scan_excl(shq_context *ctx, const T *in, T *out, size_t n)
{
    size_t tmp = 0;
    SHQ_HIP(rocprim::exclusive_scan(nullptr, tmp, in, out, T(0), n, rocprim::plus<T>(), ctx->stream));
    SHQ_TRY(ctx->act_temp.reserve(tmp + 16));
    SHQ_HIP(rocprim::exclusive_scan((void *) ctx->act_temp.ptr, tmp, in, out, T(0), n, rocprim::plus<T>(), ctx->stream));
    return SHQ_OK;
}
template <typename T> int scan_incl(shq_context *ctx, const T *in, T *out, size_t n)
{
    size_t tmp = 0;
    SHQ_HIP(rocprim::inclusive_scan(nullptr, tmp, in, out, n, rocprim::plus<T>(), ctx->stream));
    SHQ_TRY(ctx->act_temp.reserve(tmp + 16));
    SHQ_HIP(rocprim::inclusive_scan((void *) ctx->act_temp.ptr, tmp, in, out, n, rocprim::plus<T>(), ctx->stream));
    return SHQ_OK;
}

} // namespace

extern "C" int shq_fof(shq_context *ctx, const shq_fof_params *fp, const uint64_t *ids, uint64_t *minid_by_particle, int32_t *grnr_by_particle,
                       int64_t *ngroups)
{
    SHQ_CHECK(ctx && fp && ids, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->have_parts, SHQ_ERR_STATE, "fof: upload particles first");
    SHQ_CHECK(fp->BoxSize > 0 && fp->LinkingLength > 0 && fp->HaloMinLength >= 1, SHQ_ERR_INVALID, "fof: BoxSize, LinkingLength, HaloMinLength must be > 0");
    SHQ_CHECK(fp->PrimaryLinkTypes > 0 && fp->PrimaryLinkTypes < 64 && fp->SecondaryLinkTypes >= 0 && fp->SecondaryLinkTypes < 64, SHQ_ERR_INVALID,
              "fof: bad type masks");
    SHQ_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const long long n = ctx->numpart;
    ctx->fof_ngroups = -1;
    if(ngroups)
        *ngroups = 0;
    if(n == 0) {
        ctx->fof_ngroups = 0;
        ctx->fof_nmembers = 0;
        return SHQ_OK;
    }
    SHQ_CHECK(n < (1ll << 31) - 64, SHQ_ERR_INVALID, "fof: too many particles");
    /* the tree of the primary types (fof.cpp:176-178) */
    SHQ_TRY(shq_tree_build(ctx, fp->BoxSize, fp->PrimaryLinkTypes, nullptr, 0, nullptr));
    /* That tree is fof's own, as in the reference (fof_fof builds and frees it, fof.cpp:176-178 and :254): the context's resident
     * tree is gone from the caller's point of view, whatever happens below.  A later walk must rebuild or upload one — it fails
     * with SHQ_ERR_STATE instead of walking a tree that holds the primary link types only. */
    ctx->have_tree = false;
    ctx->tb_built = false;
    ctx->have_tree_targets = false;
    ctx->have_toptree = false;
    const long long nt = ctx->ntreeparts;
    FofTree t = {ctx->nodeB.ptr, ctx->nodeC.ptr, ctx->posm_leaf.ptr, ctx->leaf_pidx.ptr, fp->BoxSize, ctx->tb.order[1].ptr, ctx->tb.lo.ptr, ctx->tb.hi.ptr,
                 ctx->tb.parent.ptr, fp->LinkingLength / 1.7320508075688774 * (1 - 1e-12)};

    const size_t cap = (size_t) n;
    SHQ_TRY(ctx->fof_parent.reserve(cap));
    SHQ_TRY(ctx->fof_u64[0].reserve(cap)); /* ids, later sorted keys */
    SHQ_TRY(ctx->fof_u64[1].reserve(cap)); /* minid of roots */
    SHQ_TRY(ctx->fof_u64[2].reserve(cap)); /* labels */
    SHQ_TRY(ctx->fof_u64[3].reserve(cap)); /* labels after the secondary pass */
    SHQ_TRY(ctx->act_counts.reserve(6 * (SHQ_TIMEBINS + 1) + 64));
    unsigned long long *d_ids = ctx->fof_u64[0].ptr, *d_minid = ctx->fof_u64[1].ptr, *d_label = ctx->fof_u64[2].ptr, *d_label2 = ctx->fof_u64[3].ptr;
    SHQ_HIP(hipMemcpyAsync(d_ids, ids, sizeof(uint64_t) * cap, hipMemcpyHostToDevice, st));
    fof_init_kernel<<<dim3(nblk(n)), dim3(256), 0, st>>>(n, ctx->fof_parent.ptr, d_ids, d_minid, d_label);
    if(nt > 0) {
        fof_clique_kernel<<<dim3(nblk(nt)), dim3(256), 0, st>>>(nt, t, ctx->pfather.ptr, ctx->tb.cen.ptr, ctx->fof_parent.ptr);
        fof_link_kernel<<<dim3(nblk(nt)), dim3(256), 0, st>>>(nt, t, fp->LinkingLength, ctx->fof_parent.ptr, (int) ctx->numnodes);
        fof_minid_kernel<<<dim3(nblk(nt)), dim3(256), 0, st>>>(nt, ctx->leaf_pidx.ptr, ctx->fof_parent.ptr, d_ids, d_minid);
        fof_label_kernel<<<dim3(nblk(nt)), dim3(256), 0, st>>>(nt, ctx->leaf_pidx.ptr, ctx->fof_parent.ptr, d_minid, d_label);
    }
    SHQ_HIP(hipGetLastError());
    /* secondary types: nearest primary particle */
    SHQ_HIP(hipMemcpyAsync(d_label2, d_label, sizeof(uint64_t) * cap, hipMemcpyDeviceToDevice, st));
    unsigned long long *d_natt = ctx->act_counts.ptr;
    SHQ_HIP(hipMemsetAsync(d_natt, 0, sizeof(unsigned long long), st));
    if(nt > 0 && fp->SecondaryLinkTypes) {
        const double *d_hsml = (ctx->have_sph || ctx->have_dyn) ? ctx->hsml.ptr : nullptr;
        fof_secondary_kernel<<<dim3(nblk(n)), dim3(256), 0, st>>>(n, t, ctx->posm.ptr, ctx->pflags.ptr, d_hsml, fp->LinkingLength, fp->SecondaryLinkTypes,
                                                                  d_label, d_label2, d_natt);
        SHQ_HIP(hipGetLastError());
    }
    if(minid_by_particle)
        SHQ_HIP(hipMemcpyAsync(minid_by_particle, d_label2, sizeof(uint64_t) * cap, hipMemcpyDeviceToHost, st));

    /* catalogue: sort by label (stable: members in particle index order) */
    SHQ_TRY(ctx->fof_i32[0].reserve(cap + 1)); /* iota / flags */
    SHQ_TRY(ctx->fof_i32[1].reserve(cap + 1)); /* sorted particle indices */
    SHQ_TRY(ctx->fof_i32[2].reserve(cap + 1)); /* run id (inclusive scan of flags) */
    SHQ_TRY(ctx->fof_i32[3].reserve(cap + 1)); /* run starts */
    SHQ_TRY(ctx->fof_i32[4].reserve(cap + 1)); /* keep flag per run */
    SHQ_TRY(ctx->fof_i32[5].reserve(cap + 1)); /* kept index per run (exclusive scan) */
    int32_t *d_iota = ctx->fof_i32[0].ptr, *d_idx = ctx->fof_i32[1].ptr, *d_runid = ctx->fof_i32[2].ptr, *d_runstart = ctx->fof_i32[3].ptr,
            *d_keep = ctx->fof_i32[4].ptr, *d_keptidx = ctx->fof_i32[5].ptr;
    unsigned long long *d_keys = ctx->fof_u64[0].ptr; /* the ids are no longer needed */
    fof_iota_kernel<<<dim3(nblk(n)), dim3(256), 0, st>>>(n, d_iota);
    {
        size_t tmp = 0;
        SHQ_HIP(rocprim::radix_sort_pairs(nullptr, tmp, d_label2, d_keys, d_iota, d_idx, cap, 0, 64, st));
        SHQ_TRY(ctx->act_temp.reserve(tmp + 16));
        SHQ_HIP(rocprim::radix_sort_pairs((void *) ctx->act_temp.ptr, tmp, d_label2, d_keys, d_iota, d_idx, cap, 0, 64, st));
    }
    int32_t *d_flags = d_iota;
    fof_flag_kernel<<<dim3(nblk(n)), dim3(256), 0, st>>>(n, d_keys, d_flags);
    SHQ_TRY(scan_incl(ctx, (const int32_t *) d_flags, d_runid, cap));
    int32_t h_nruns = 0;
    SHQ_HIP(hipMemcpyAsync(&h_nruns, d_runid + (n - 1), sizeof(int32_t), hipMemcpyDeviceToHost, st));
    SHQ_HIP(hipStreamSynchronize(st));
    const long long nruns = h_nruns;
    fof_runstart_kernel<<<dim3(nblk(n)), dim3(256), 0, st>>>(n, d_flags, d_runid, d_runstart, nruns);
    fof_keep_kernel<<<dim3(nblk(nruns)), dim3(256), 0, st>>>(nruns, d_runstart, fp->HaloMinLength, d_keep);
    SHQ_HIP(hipMemsetAsync(d_keep + nruns, 0, sizeof(int32_t), st));
    SHQ_TRY(scan_excl(ctx, (const int32_t *) d_keep, d_keptidx, (size_t) nruns + 1));
    int32_t h_ng = 0;
    SHQ_HIP(hipMemcpyAsync(&h_ng, d_keptidx + nruns, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    SHQ_HIP(hipStreamSynchronize(st));
    const long long ng = h_ng;
    const size_t gcap = (size_t) std::max<long long>(ng, 1);
    SHQ_TRY(ctx->fof_g32[0].reserve(gcap)); /* gstart */
    SHQ_TRY(ctx->fof_g32[1].reserve(gcap)); /* glen */
    SHQ_TRY(ctx->fof_g32[2].reserve(gcap)); /* grnr */
    SHQ_TRY(ctx->fof_g32[3].reserve(gcap)); /* group ids / order */
    SHQ_TRY(ctx->fof_g32[4].reserve(gcap));
    SHQ_TRY(ctx->fof_gkey[0].reserve(gcap));
    SHQ_TRY(ctx->fof_gkey[1].reserve(gcap));
    SHQ_TRY(ctx->fof_groups.reserve(gcap));
    SHQ_TRY(ctx->fof_goff[0].reserve(gcap + 1));
    SHQ_TRY(ctx->fof_goff[1].reserve(gcap + 1));
    SHQ_TRY(ctx->fof_partgrnr.reserve(cap));
    int32_t *d_gstart = ctx->fof_g32[0].ptr, *d_glen = ctx->fof_g32[1].ptr, *d_grnr = ctx->fof_g32[2].ptr;
    long long nmembers = 0;
    if(ng > 0) {
        fof_groups_kernel<<<dim3(nblk(nruns)), dim3(256), 0, st>>>(nruns, d_runstart, d_keep, d_keptidx, d_gstart, d_glen, ctx->fof_gkey[0].ptr, ctx->fof_g32[3].ptr);
        {
            /* groups are in MinID order; a stable sort on UINT_MAX - Length gives (length descending, MinID ascending) */
            size_t tmp = 0;
            SHQ_HIP(rocprim::radix_sort_pairs(nullptr, tmp, ctx->fof_gkey[0].ptr, ctx->fof_gkey[1].ptr, ctx->fof_g32[3].ptr, ctx->fof_g32[4].ptr, (size_t) ng, 0, 32, st));
            SHQ_TRY(ctx->act_temp.reserve(tmp + 16));
            SHQ_HIP(rocprim::radix_sort_pairs((void *) ctx->act_temp.ptr, tmp, ctx->fof_gkey[0].ptr, ctx->fof_gkey[1].ptr, ctx->fof_g32[3].ptr, ctx->fof_g32[4].ptr,
                                              (size_t) ng, 0, 32, st));
        }
        fof_grnr_kernel<<<dim3(nblk(ng)), dim3(256), 0, st>>>(ng, ctx->fof_g32[4].ptr, d_grnr);
        const double *d_vel = (ctx->have_sph || ctx->have_dyn) ? ctx->vel.ptr : nullptr;
        const double *d_dens = ctx->have_sph ? ctx->g_density.ptr : nullptr;
        const double *d_delay = ctx->have_sph ? ctx->g_delaytime.ptr : nullptr;
        const FofPropArgs pa = {d_gstart, d_glen, d_grnr, d_idx, d_keys, ctx->posm.ptr, d_vel, ctx->pflags.ptr, d_dens, d_delay, fp->WindsDecoupleSph, fp->BoxSize,
                                ctx->fof_groups.ptr};
        fof_props_kernel<<<dim3(nblk(ng, 64)), dim3(64), 0, st>>>(ng, pa);
        {
            /* the long groups: a list of their numbers, one workgroup each */
            int32_t *d_flag = ctx->fof_g32[3].ptr, *d_excl = ctx->fof_g32[4].ptr; /* the GrNr sort is done with them */
            SHQ_TRY(ctx->fof_biglist.reserve(gcap + 1));
            fof_bigflag_kernel<<<dim3(nblk(ng)), dim3(256), 0, st>>>(ng, d_glen, d_flag);
            SHQ_TRY(scan_excl(ctx, (const int32_t *) d_flag, d_excl, (size_t) ng));
            int32_t lastf = 0, laste = 0;
            SHQ_HIP(hipMemcpyAsync(&lastf, d_flag + (ng - 1), sizeof(int32_t), hipMemcpyDeviceToHost, st));
            SHQ_HIP(hipMemcpyAsync(&laste, d_excl + (ng - 1), sizeof(int32_t), hipMemcpyDeviceToHost, st));
            SHQ_HIP(hipStreamSynchronize(st));
            const int nbig = lastf + laste;
            if(nbig > 0) {
                fof_biglist_kernel<<<dim3(nblk(ng)), dim3(256), 0, st>>>(ng, d_flag, d_excl, ctx->fof_biglist.ptr);
                SHQ_TRY(ctx->fof_partial.reserve((size_t) nbig * FOF_SPLIT * sizeof(FofSums)));
                FofSums *d_partial = reinterpret_cast<FofSums *>(ctx->fof_partial.ptr);
                fof_props_big_kernel<<<dim3((unsigned) nbig * FOF_SPLIT), dim3(256), 0, st>>>(nbig, ctx->fof_biglist.ptr, pa, d_partial);
                fof_props_big_finish_kernel<<<dim3(nblk(nbig, 64)), dim3(64), 0, st>>>(nbig, ctx->fof_biglist.ptr, pa, d_partial);
            }
        }
        fof_glen64_kernel<<<dim3(nblk(ng)), dim3(256), 0, st>>>(ng, d_glen, ctx->fof_goff[0].ptr);
        SHQ_HIP(hipMemsetAsync(ctx->fof_goff[0].ptr + ng, 0, sizeof(long long), st));
        SHQ_TRY(scan_excl(ctx, (const long long *) ctx->fof_goff[0].ptr, ctx->fof_goff[1].ptr, (size_t) ng + 1));
        fof_member_offsets_kernel<<<dim3(nblk(ng)), dim3(256), 0, st>>>(ng, d_glen, ctx->fof_goff[1].ptr, ctx->fof_groups.ptr);
        SHQ_HIP(hipMemcpyAsync(&nmembers, ctx->fof_goff[1].ptr + ng, sizeof(long long), hipMemcpyDeviceToHost, st));
    }
    fof_partgrnr_kernel<<<dim3(nblk(n)), dim3(256), 0, st>>>(n, d_idx, d_runid, d_keep, d_keptidx, d_grnr, ctx->fof_partgrnr.ptr);
    SHQ_HIP(hipGetLastError());
    if(grnr_by_particle)
        SHQ_HIP(hipMemcpyAsync(grnr_by_particle, ctx->fof_partgrnr.ptr, sizeof(int32_t) * cap, hipMemcpyDeviceToHost, st));
    SHQ_HIP(hipStreamSynchronize(st));
    ctx->fof_ngroups = ng;
    ctx->fof_nmembers = nmembers;
    ctx->fof_nruns = nruns;
    if(ngroups)
        *ngroups = ng;
    return SHQ_OK;
}

extern "C" int shq_fof_groups_download(shq_context *ctx, shq_fof_group *groups, int64_t capacity)
{
    SHQ_CHECK(ctx && (groups || capacity == 0), SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->fof_ngroups >= 0, SHQ_ERR_STATE, "fof_groups_download: run shq_fof first");
    SHQ_CHECK(capacity >= ctx->fof_ngroups, SHQ_ERR_INVALID, "fof_groups_download: capacity %ld < %ld groups", (long) capacity, (long) ctx->fof_ngroups);
    SHQ_HIP(hipSetDevice(ctx->device));
    if(ctx->fof_ngroups > 0)
        SHQ_HIP(hipMemcpy(groups, ctx->fof_groups.ptr, sizeof(shq_fof_group) * (size_t) ctx->fof_ngroups, hipMemcpyDeviceToHost));
    return SHQ_OK;
}

extern "C" int shq_fof_members(shq_context *ctx, int32_t *members, int64_t capacity, int64_t *nmembers)
{
    SHQ_CHECK(ctx, SHQ_ERR_INVALID, "null context");
    SHQ_CHECK(ctx->fof_ngroups >= 0, SHQ_ERR_STATE, "fof_members: run shq_fof first");
    if(nmembers)
        *nmembers = ctx->fof_nmembers;
    if(!members)
        return SHQ_OK;
    SHQ_CHECK(capacity >= ctx->fof_nmembers, SHQ_ERR_INVALID, "fof_members: capacity %ld < %ld", (long) capacity, (long) ctx->fof_nmembers);
    SHQ_CHECK(ctx->have_parts && ctx->fof_i32[1].ptr, SHQ_ERR_STATE, "fof_members: the particle set changed since shq_fof");
    SHQ_HIP(hipSetDevice(ctx->device));
    const long long n = ctx->numpart;
    if(ctx->fof_nmembers > 0) {
        SHQ_TRY(ctx->fof_members.reserve((size_t) ctx->fof_nmembers));
        fof_members_kernel<<<dim3(nblk(n)), dim3(256), 0, ctx->stream>>>(n, ctx->fof_i32[1].ptr, ctx->fof_i32[2].ptr, ctx->fof_i32[4].ptr, ctx->fof_i32[5].ptr,
                                                                        ctx->fof_i32[3].ptr, ctx->fof_goff[1].ptr, ctx->fof_members.ptr);
        SHQ_HIP(hipGetLastError());
        SHQ_HIP(hipMemcpyAsync(members, ctx->fof_members.ptr, sizeof(int32_t) * (size_t) ctx->fof_nmembers, hipMemcpyDeviceToHost, ctx->stream));
        SHQ_HIP(hipStreamSynchronize(ctx->stream));
    }
    return SHQ_OK;
}

namespace {
struct SeedGroup {
    const shq_fof_group *g;
    double minmass, minstar;
    /* Marked[i], fof.cpp:1294-1298 */
    __device__ bool operator()(const int32_t i) const
    {
        const shq_fof_group &G = g[i];
        return G.Mass >= minmass && G.MassType[4] >= minstar && G.LenType[5] == 0 && G.seed_index >= 0;
    }
};
__global__ void fof_seed_index_kernel(long long n, const int32_t *which, const shq_fof_group *g, int32_t *out)
{
    const long long k = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(k < n)
        out[k] = g[which[k]].seed_index;
}
} // namespace

extern "C" int shq_fof_seed_select(shq_context *ctx, double MinFoFMassForNewSeed, double MinMStarForNewSeed, int32_t *d_seed_index, int64_t capacity, int64_t *nseeds)
{
    SHQ_CHECK(ctx && nseeds, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->fof_ngroups >= 0, SHQ_ERR_STATE, "fof_seed_select: run shq_fof first");
    *nseeds = 0;
    const long long ng = ctx->fof_ngroups;
    if(ng == 0)
        return SHQ_OK;
    SHQ_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    SHQ_TRY(ctx->ex_list.reserve((size_t) ng));
    SHQ_TRY(ctx->ex_counts.reserve(64));
    size_t tmp = 0;
    rocprim::counting_iterator<int32_t> it(0);
    const SeedGroup pred{ctx->fof_groups.ptr, MinFoFMassForNewSeed, MinMStarForNewSeed};
    SHQ_HIP(rocprim::select(nullptr, tmp, it, ctx->ex_list.ptr, ctx->ex_counts.ptr, (size_t) ng, pred, st));
    SHQ_TRY(ctx->ex_bytes.reserve(tmp + 16));
    SHQ_HIP(rocprim::select(ctx->ex_bytes.ptr, tmp, it, ctx->ex_list.ptr, ctx->ex_counts.ptr, (size_t) ng, pred, st));
    unsigned long long h = 0;
    SHQ_HIP(hipMemcpyAsync(&h, ctx->ex_counts.ptr, sizeof(h), hipMemcpyDeviceToHost, st));
    SHQ_HIP(hipStreamSynchronize(st));
    *nseeds = (int64_t) h;
    if(!d_seed_index || h == 0)
        return SHQ_OK;
    SHQ_CHECK(capacity >= (int64_t) h, SHQ_ERR_INVALID, "fof_seed_select: capacity %ld < %ld seeds", (long) capacity, (long) h);
    fof_seed_index_kernel<<<dim3(nblk((long long) h)), dim3(256), 0, st>>>((long long) h, ctx->ex_list.ptr, ctx->fof_groups.ptr, d_seed_index);
    SHQ_HIP(hipGetLastError());
    return SHQ_OK;
}

namespace {
/* one workgroup per (group, column): 256 contiguous slices of the member list, each summed in member order by one thread, the slice
 * sums added in slice order by thread 0.  A group of up to 256 members is the plain serial sum of add_particle_to_group. */
__global__ __launch_bounds__(256) void fof_colsum_kernel(const shq_fof_group *g, const int32_t *members, const double *values, int ncol, double *sums)
{
#pragma clang fp contract(off)
    __shared__ double part[256];
    const long long grp = blockIdx.x / ncol;
    const int col = (int) (blockIdx.x % ncol);
    const long long first = g[grp].first_member, len = g[grp].Length;
    const long long per = (len + 255) / 256;
    const long long a = first + (long long) threadIdx.x * per, e = a + per < first + len ? a + per : first + len;
    double s = 0;
    for(long long k = a; k < e; k++)
        s += values[(size_t) members[k] * ncol + col];
    part[threadIdx.x] = s;
    __syncthreads();
    if(threadIdx.x == 0) {
        double t = 0;
        for(int j = 0; j < 256; j++)
            t += part[j];
        sums[(size_t) grp * ncol + col] = t;
    }
}
} // namespace

extern "C" int shq_fof_group_sums(shq_context *ctx, const double *values_by_particle, int ncol, double *sums_by_group)
{
    SHQ_CHECK(ctx && (ncol == 0 || (values_by_particle && sums_by_group)), SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->fof_ngroups >= 0, SHQ_ERR_STATE, "fof_group_sums: run shq_fof first");
    SHQ_CHECK(ncol >= 0 && ncol <= 1024, SHQ_ERR_INVALID, "fof_group_sums: %d columns", ncol);
    const long long ng = ctx->fof_ngroups, n = ctx->numpart;
    if(ng == 0 || ncol == 0)
        return SHQ_OK;
    SHQ_CHECK(ctx->have_parts && ctx->fof_i32[1].ptr, SHQ_ERR_STATE, "fof_group_sums: the particle set changed since shq_fof");
    SHQ_CHECK(ng * ncol < (1ll << 31), SHQ_ERR_INVALID, "fof_group_sums: %ld groups x %d columns is too many for one launch", (long) ng, ncol);
    SHQ_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    SHQ_TRY(ctx->fof_members.reserve((size_t) std::max<int64_t>(ctx->fof_nmembers, 1)));
    fof_members_kernel<<<dim3(nblk(n)), dim3(256), 0, st>>>(n, ctx->fof_i32[1].ptr, ctx->fof_i32[2].ptr, ctx->fof_i32[4].ptr, ctx->fof_i32[5].ptr, ctx->fof_i32[3].ptr,
                                                           ctx->fof_goff[1].ptr, ctx->fof_members.ptr);
    SHQ_HIP(hipGetLastError());
    SHQ_TRY(ctx->ex_u64.reserve((size_t) n * ncol + (size_t) ng * ncol));
    double *d_val = reinterpret_cast<double *>(ctx->ex_u64.ptr), *d_sum = d_val + (size_t) n * ncol;
    SHQ_HIP(hipMemcpyAsync(d_val, values_by_particle, sizeof(double) * (size_t) n * ncol, hipMemcpyHostToDevice, st));
    fof_colsum_kernel<<<dim3((unsigned) (ng * ncol)), dim3(256), 0, st>>>(ctx->fof_groups.ptr, ctx->fof_members.ptr, d_val, ncol, d_sum);
    SHQ_HIP(hipGetLastError());
    SHQ_HIP(hipMemcpyAsync(sums_by_group, d_sum, sizeof(double) * (size_t) ng * ncol, hipMemcpyDeviceToHost, st));
    SHQ_HIP(hipStreamSynchronize(st));
    return SHQ_OK;
}

static_assert(sizeof(shq_fof_group) == 272, "shq_fof_group layout (shenqi_amd/capi.py FOF_GROUP_DTYPE)");
