/* tree_build.hip — oct-tree construction + moments on the device (gfx950).
 *
 * Replaces, for a single-domain tree, force_tree_rebuild / force_tree_create_nodes (insertion build,
 * libgadget/forcetree.cpp:727-859 with create_new_node_layer :393-470), force_tree_calc_moments /
 * force_update_node_parallel (:1118-1142, :1016-1103: sibling/child threading, mass, cofm, hmax) and
 * the host-side repack of shq_tree_upload.  SURVEY.md §8(f) rank 1: with hierarchical time steps the
 * reference rebuilds a tree twice per step, and once the walks are fast the host build dominates.
 *
 * The reference inserts particles one by one; the tree it ends up with does not depend on the
 * insertion order: a node holding more than NMAXCHILD = 8 particles is split into the sub-octants that
 * are not empty (get_subnode, forcetree.cpp:277-283: `pos > center` per axis; child centre = centre +-
 * len/4, init_internal_node :302-328), recursively.  So the tree is built top-down from sorted keys:
 *   1. every particle descends 21 levels from the root cell with exactly the reference's floating-point
 *      operations and records its octant at each level: a 63-bit key (excluded particles get ~0);
 *   2. a radix sort by key (rocPRIM) puts the particles in depth-first leaf order; inside a leaf the
 *      reference keeps them in insertion (= candidate sequence) order, which a small per-leaf sort of
 *      the <= 8 sequence numbers restores once the leaves are known;
 *   3. breadth-first, one pair of kernels per level: every internal node of the level finds its eight
 *      octant boundaries by binary search on the key digit, an exclusive scan allocates the children
 *      (contiguous per parent), a second kernel writes them (centre, len, father, sibling threading)
 *      and queues those with more than 8 particles for the next level;
 *   4. moments bottom-up, level by level, summed in the reference's order without fma contraction;
 *   5. nodes are ranked in depth-first pre-order (sort by first particle, then level) and written
 *      straight into the walk kernels' pool records; the particle copy in leaf order is the sorted order.
 * Trees deeper than 21 levels (more than 8 particles within Box/2^21 of each other) are refused with an
 * error: the caller must fall back to its host build; nothing is truncated silently. */
#include <cstring>
#include <math.h>
#include <vector>
#include "common.hpp"
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>

namespace {

constexpr int TB_LEVELS = 21;

/* octant digit of `key` at level l (0 = children of the root) */
__device__ __forceinline__ int key_digit(unsigned long long key, int l) { return (int) ((key >> (3 * (TB_LEVELS - 1 - l))) & 7ull); }

__global__ void tb_key_kernel(long long ncand, const int32_t *__restrict__ cand, const double4 *__restrict__ posm,
                              const uint8_t *__restrict__ pflags, int mask, double Box, unsigned long long *keys, int32_t *idx)
{
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    bool ok = false;
    if(i < ncand) {
        const int p = cand ? cand[i] : (int) i;
        const unsigned f = pflags[p];
        ok = !(f & 3u) && (((1u << (f >> 4)) & (unsigned) mask) != 0u);
        unsigned long long key = ~0ull;
        if(ok) {
            const double4 q = posm[p];
            /* Fast path: the cell boundaries of every level are, up to the rounding of the reference's centre
             * updates (~1e-16 L), the dyadic fractions of the root cell, so a position that is not within
             * 2^-20 of a finest-level (2^-21) cell boundary falls on the same side of every centre as its
             * fixed-point coordinate says: the key is the bit interleave of the three 21-bit cell indices.
             * The few positions that close to a boundary (6e-6 of all) take the
             * level-by-level descent below, which repeats the reference's arithmetic exactly. */
            const double len0 = Box * 1.001, lo0 = Box / 2. - 0.5 * len0, sc = 2097152.0 / len0; /* 2^21 cells */
            const double tx = (q.x - lo0) * sc, ty = (q.y - lo0) * sc, tz = (q.z - lo0) * sc;
            const double fx = tx - floor(tx), fy = ty - floor(ty), fz = tz - floor(tz);
            const double eps = 1.0 / 1048576.; /* 1e-6 cells: >> the 1e-8 the roundings can add up to, and rare enough
                                                  that hardly any wave has to run both paths */
            const bool safe = fx > eps && fx < 1 - eps && fy > eps && fy < 1 - eps && fz > eps && fz < 1 - eps && tx > 0 && ty > 0 &&
                              tz > 0 && tx < 2097152.0 && ty < 2097152.0 && tz < 2097152.0;
            if(safe) {
                auto spread = [](unsigned long long v) { /* 21 bits -> every third bit */
                    v &= 0x1fffffull;
                    v = (v | (v << 32)) & 0x1f00000000ffffull;
                    v = (v | (v << 16)) & 0x1f0000ff0000ffull;
                    v = (v | (v << 8)) & 0x100f00f00f00f00full;
                    v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
                    v = (v | (v << 2)) & 0x1249249249249249ull;
                    return v;
                };
                key = spread((unsigned long long) tx) | (spread((unsigned long long) ty) << 1) | (spread((unsigned long long) tz) << 2);
            } else {
                double cx = Box / 2., cy = Box / 2., cz = Box / 2., len = Box * 1.001; /* forcetree.cpp:661 */
                key = 0;
                for(int l = 0; l < TB_LEVELS; l++) {
                    const int s = (q.x > cx) + ((q.y > cy) << 1) + ((q.z > cz) << 2);
                    key = (key << 3) | (unsigned long long) s;
                    const double lenhalf = 0.25 * len;
                    cx = cx + ((s & 1) ? lenhalf : -lenhalf);
                    cy = cy + ((s & 2) ? lenhalf : -lenhalf);
                    cz = cz + ((s & 4) ? lenhalf : -lenhalf);
                    len = 0.5 * len;
                }
            }
        }
        keys[i] = key;
        idx[i] = (int32_t) i; /* position in the candidate sequence: restores the order inside a leaf */
    }
}

/* number of particles in the tree = first sorted position holding the excluded-particle key ~0
 * (one counter bumped by every wave cost 3 ms of same-address atomics at 256^3) */
__global__ void tb_count_kernel(const unsigned long long *__restrict__ keys, long long ncand, unsigned long long *nvalid)
{
    long long a = 0, e = ncand;
    while(a < e) {
        const long long mid = a + ((e - a) >> 1);
        if(keys[mid] != ~0ull)
            a = mid + 1;
        else
            e = mid;
    }
    *nvalid = (unsigned long long) a;
}

struct TbNodes {
    int32_t *lo, *hi, *parent, *sibling, *firstchild, *nchild, *level;
    double4 *cen;  /* centre, len */
    double4 *mom;  /* cofm, mass */
    double *hmax;
    int32_t *top;              /* domain build: TopNodes index, -1 below the top tree */
    unsigned long long *path;  /* domain build: octant digits from the root, left-aligned */
};

/* The top tree of a domain decomposition as geometry (shq_topnode_geo): daughters per octant, and per TopNode
 * kind 0 internal / 1 leaf of this task / 2 leaf of another task (pseudo) with its TopLeaves index. */
struct TbGeo {
    const int4 *c0, *c1;
    const int2 *kind;
    __device__ int child(int t, int s) const
    {
        const int4 v = s < 4 ? c0[t] : c1[t];
        const int k = s & 3;
        return k == 0 ? v.x : (k == 1 ? v.y : (k == 2 ? v.z : v.w));
    }
};
enum { TOPK_INTERNAL = 0, TOPK_LOCAL = 1, TOPK_PSEUDO = 2 };

__global__ void tb_root_kernel(TbNodes nd, int n, double Box, int dom)
{
    if(dom) {
        nd.top[0] = 0;
        nd.path[0] = 0ull;
    }
    nd.lo[0] = 0;
    nd.hi[0] = n;
    nd.parent[0] = -1;
    nd.sibling[0] = -1;
    nd.firstchild[0] = -1;
    nd.nchild[0] = 0;
    nd.level[0] = 0;
    nd.cen[0] = make_double4(Box / 2., Box / 2., Box / 2., Box * 1.001);
}

/* octant boundaries of every internal node of the level: bounds[9 f + s] = first sorted position whose
 * digit is >= s; packed[f] = (#children << 32) | #children with more than NMAXCHILD particles */
/* The level loop keeps its bookkeeping on the device (round 4): how many nodes exist, the frontier of the level at hand and of the next
 * one, where every level starts in the pool.  The host queues all TB_LEVELS levels back to back - fixed grids, grid-stride loops over a
 * frontier whose length the kernels read here - and looks at the result once: a build used to make one host round trip per level,
 * which is what stretched it threefold beside the FFT passes of an early PM. */
struct TbState {
    int nn, nf, nf_next, maxdepth, overflow, pad_[3];
    int level_start[TB_LEVELS + 3];
};

__global__ void tb_state_init_kernel(TbState *st, int nf0)
{
    st->nn = 1;
    st->nf = nf0;
    st->nf_next = 0;
    st->maxdepth = 0;
    st->overflow = 0;
    st->level_start[0] = 0;
    for(int l = 1; l < TB_LEVELS + 3; l++)
        st->level_start[l] = 1;
}

/* the level is through: its children are the pool's newest nodes, its internal children the next frontier */
__global__ void tb_advance_kernel(TbState *st, int level)
{
    if(st->nf > 0)
        st->maxdepth = level + 1;
    for(int l = level + 2; l < TB_LEVELS + 3; l++)
        st->level_start[l] = st->nn;
    st->nf = st->nf_next;
    st->nf_next = 0;
}

template <bool DOM>
__global__ void tb_split_kernel(const TbState *st, const int32_t *__restrict__ frontier, TbNodes nd, const unsigned long long *__restrict__ keys,
                                int level, int32_t *bounds, unsigned long long *packed, TbGeo geo)
{
    const int nf = st->nf;
    for(int f = blockIdx.x * blockDim.x + threadIdx.x; f < nf; f += gridDim.x * blockDim.x) {
    const int no = frontier[f];
    const int lo = nd.lo[no], hi = nd.hi[no];
    int b[9];
    b[0] = lo;
    b[8] = hi;
    for(int s = 1; s < 8; s++) {
        int a = b[s - 1], e = hi; /* first k in [a, e) with digit >= s */
        while(a < e) {
            const int mid = a + ((e - a) >> 1);
            if(key_digit(keys[mid], level) < s)
                a = mid + 1;
            else
                e = mid;
        }
        b[s] = a;
    }
    unsigned nch = 0, nint = 0;
    const int t = DOM ? nd.top[no] : -1;
    if(DOM && t >= 0 && geo.kind[t].x == TOPK_INTERNAL) {
        /* an internal top-level node has all eight daughters whether or not they hold particles (forcetree.cpp:880-915,
         * never removed: :1040-1045); a daughter goes on when it is internal too, or a leaf of this task with a sub-tree */
        nch = 8;
        for(int s = 0; s < 8; s++) {
            const int k = geo.kind[geo.child(t, s)].x;
            nint += (k == TOPK_INTERNAL) || (k == TOPK_LOCAL && b[s + 1] - b[s] > SHQ_NMAXCHILD);
        }
    } else
        for(int s = 0; s < 8; s++) {
            const int c = b[s + 1] - b[s];
            nch += c > 0;
            nint += c > SHQ_NMAXCHILD;
        }
    for(int s = 0; s < 9; s++)
        bounds[9 * f + s] = b[s];
    packed[f] = ((unsigned long long) nch << 32) | nint;
    }
}

/* A node's children take the next packed[f] >> 32 pool slots, contiguous and in octant order, wherever the pool's end happens to be when
 * the node gets there (one atomic per node): the numbering inside a level depends on the order the threads arrive in, the tree does not -
 * the walk pool is laid out by the pre-order rank, which is a sort on (first particle, level) / the octant path, not on these numbers. */
template <bool DOM>
__global__ void tb_children_kernel(TbState *st, const int32_t *__restrict__ frontier, TbNodes nd, int level, const int32_t *__restrict__ bounds,
                                   const unsigned long long *__restrict__ packed, int cap, int32_t *next_frontier, int *err, TbGeo geo)
{
    const int nf = st->nf;
    for(int f = blockIdx.x * blockDim.x + threadIdx.x; f < nf; f += gridDim.x * blockDim.x) {
    const int no = frontier[f];
    const int want = (int) (packed[f] >> 32), wantq = (int) (packed[f] & 0xffffffffull);
    const int base = atomicAdd(&st->nn, want);
    int qoff = atomicAdd(&st->nf_next, wantq);
    if((long long) base + want > (long long) cap) { /* the pool is too small: the host doubles it and builds again */
        st->overflow = 1;
        continue;
    }
    const double4 pc = nd.cen[no];
    const int psib = nd.sibling[no];
    const double lenhalf = 0.25 * pc.w; /* init_internal_node, forcetree.cpp:302-328 */
    const int t = DOM ? nd.top[no] : -1;
    const bool forced = DOM && t >= 0 && geo.kind[t].x == TOPK_INTERNAL;
    int nch = 0;
    for(int s = 0; s < 8; s++)
        nch += forced || bounds[9 * f + s + 1] > bounds[9 * f + s];
    int j = 0;
    for(int s = 0; s < 8; s++) {
        const int lo = bounds[9 * f + s], hi = bounds[9 * f + s + 1];
        if(hi == lo && !forced)
            continue;
        const int c = base + j;
        int ckind = -1;
        if(DOM) {
            const int ct = forced ? geo.child(t, s) : -1;
            nd.top[c] = ct;
            nd.path[c] = nd.path[no] | ((unsigned long long) s << (3 * (TB_LEVELS - 1 - level)));
            if(ct >= 0) {
                ckind = geo.kind[ct].x;
                if(ckind == TOPK_PSEUDO && hi > lo)
                    *err = 3; /* a particle of this rank inside another task's top leaf: "Bad topleaf", forcetree.cpp:807-808 */
            }
        }
        nd.lo[c] = lo;
        nd.hi[c] = hi;
        nd.parent[c] = no;
        nd.level[c] = level + 1;
        nd.firstchild[c] = -1;
        nd.nchild[c] = 0;
        nd.sibling[c] = (j + 1 < nch) ? c + 1 : psib; /* forcetree.cpp:968-983,1055-1061 */
        nd.cen[c] = make_double4(pc.x + ((s & 1) ? lenhalf : -lenhalf), pc.y + ((s & 2) ? lenhalf : -lenhalf),
                                 pc.z + ((s & 4) ? lenhalf : -lenhalf), 0.5 * pc.w);
        const bool goes_on = DOM && ckind >= 0 ? (ckind == TOPK_INTERNAL || (ckind == TOPK_LOCAL && hi - lo > SHQ_NMAXCHILD))
                                               : hi - lo > SHQ_NMAXCHILD;
        if(goes_on) {
            if(level + 1 >= TB_LEVELS)
                *err = 2; /* more than NMAXCHILD particles in one cell of the deepest level */
            else
                next_frontier[qoff++] = c;
        }
        j++;
    }
    nd.firstchild[no] = base;
    nd.nchild[no] = nch;
    }
}

/* leaves list their particles in candidate-sequence order (the sort ordered them by the deeper key
 * digits); also turns sequence numbers into particle indices */
__global__ void tb_leafsort_kernel(int nn, TbNodes nd, const int32_t *__restrict__ seq, const int32_t *__restrict__ cand, int32_t *idx)
{
    const int no = blockIdx.x * blockDim.x + threadIdx.x;
    if(no >= nn || nd.nchild[no] != 0)
        return;
    const int lo = nd.lo[no], cnt = nd.hi[no] - lo;
    if(cnt > SHQ_NMAXCHILD) { /* only in the refused too-deep case */
        for(int k = 0; k < cnt; k++)
            idx[lo + k] = cand ? cand[seq[lo + k]] : seq[lo + k];
        return;
    }
    int v[SHQ_NMAXCHILD];
#pragma unroll
    for(int k = 0; k < SHQ_NMAXCHILD; k++)
        v[k] = k < cnt ? seq[lo + k] : 0x7fffffff;
#pragma unroll
    for(int i = 1; i < SHQ_NMAXCHILD; i++) /* insertion sort, fully unrolled: v stays in registers */
#pragma unroll
        for(int j = i; j > 0; j--)
            if(v[j] < v[j - 1]) {
                const int t = v[j];
                v[j] = v[j - 1];
                v[j - 1] = t;
            }
#pragma unroll
    for(int k = 0; k < SHQ_NMAXCHILD; k++)
        if(k < cnt)
            idx[lo + k] = cand ? cand[v[k]] : v[k];
}

/* mass, centre of mass, hmax of the nodes [first, last) of one level; children are one level deeper and done */
__global__ void tb_moments_kernel(int first, int last, TbNodes nd, const int32_t *__restrict__ idx, const double4 *__restrict__ posm,
                                  const uint8_t *__restrict__ pflags, const double *__restrict__ hsml)
{
#pragma clang fp contract(off) /* the reference's sums are separate multiplies and adds */
    const int no = first + blockIdx.x * blockDim.x + threadIdx.x;
    if(no >= last)
        return;
    const double4 c = nd.cen[no];
    double mass = 0, c0 = 0, c1 = 0, c2 = 0, hmax = 0;
    const int nch = nd.nchild[no];
    if(nch == 0) { /* leaf: forcetree.cpp:947-966 (moments) and :985-1005 (hmax), particles in leaf order */
        for(int k = nd.lo[no]; k < nd.hi[no]; k++) {
            const int p = idx[k];
            const double4 q = posm[p];
            mass = mass + q.w;
            c0 = c0 + q.w * q.x;
            c1 = c1 + q.w * q.y;
            c2 = c2 + q.w * q.z;
            const unsigned type = pflags[p] >> 4;
            if(hsml && (type == 0 || type == 5)) {
                const double h = hsml[p];
                hmax = fmax(hmax, fabs(q.x - c.x) + h - c.w / 2.);
                hmax = fmax(hmax, fabs(q.y - c.y) + h - c.w / 2.);
                hmax = fmax(hmax, fabs(q.z - c.z) + h - c.w / 2.);
            }
        }
        if(mass > 0) {
            c0 /= mass;
            c1 /= mass;
            c2 /= mass;
        } else {
            c0 = c.x;
            c1 = c.y;
            c2 = c.z;
        }
    } else { /* forcetree.cpp:1080-1101 */
        const int fc = nd.firstchild[no];
        for(int j = 0; j < nch; j++) {
            const double4 m = nd.mom[fc + j];
            mass = mass + m.w;
            c0 = c0 + m.w * m.x;
            c1 = c1 + m.w * m.y;
            c2 = c2 + m.w * m.z;
            hmax = fmax(hmax, nd.hmax[fc + j]);
        }
        if(mass > 0) {
            c0 /= mass;
            c1 /= mass;
            c2 /= mass;
        }
    }
    nd.mom[no] = make_double4(c0, c1, c2, mass);
    nd.hmax[no] = hmax;
}

__global__ void tb_orderkey_kernel(int nn, TbNodes nd, unsigned long long *okeys, int32_t *oval)
{
    const int no = blockIdx.x * blockDim.x + threadIdx.x;
    if(no >= nn)
        return;
    okeys[no] = ((unsigned long long) (unsigned) nd.lo[no] << 8) | (unsigned long long) nd.level[no];
    oval[no] = no;
}

/* domain build: empty top-level nodes share `lo` with what follows them, so the pre-order comes from the octant path: sort by
 * level first, then (stable) by the left-aligned path — an ancestor and its first-daughter chain tie on the path and keep level order */
__global__ void tb_levelkey_kernel(int nn, TbNodes nd, unsigned long long *okeys, int32_t *oval)
{
    const int no = blockIdx.x * blockDim.x + threadIdx.x;
    if(no >= nn)
        return;
    okeys[no] = (unsigned long long) nd.level[no];
    oval[no] = no;
}
__global__ void tb_pathkey_kernel(int nn, TbNodes nd, const int32_t *__restrict__ order, unsigned long long *okeys)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if(r < nn)
        okeys[r] = nd.path[order[r]];
}

__global__ void tb_rank_kernel(int nn, const int32_t *__restrict__ order, int32_t *rank)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if(r < nn)
        rank[order[r]] = r;
}

/* pool records in pre-order (common.hpp); record nn is the pad the walks may touch */
__global__ void tb_pack_kernel(int nn, const int32_t *__restrict__ order, const int32_t *__restrict__ rank, TbNodes nd,
                               const int32_t *__restrict__ idx, NodeA *A, NodeB *B, NodeC *C, NodeG *G, double *H, int32_t *pfather,
                               double Box, TbGeo geo)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if(r > nn)
        return;
    NodeA a;
    NodeB b;
    NodeC c;
    double hmax = 0;
    if(r == nn) {
        memset(&a, 0, sizeof(a));
        memset(&b, 0, sizeof(b));
        c.sibling = -1;
        c.child = -1;
        c.type = SHQ_PSEUDO_NODE_TYPE;
        c.count = 0;
    } else {
        const int no = order[r];
        const double4 m = nd.mom[no], ce = nd.cen[no];
        a.cofm[0] = m.x; a.cofm[1] = m.y; a.cofm[2] = m.z; a.mass = m.w;
        b.center[0] = ce.x; b.center[1] = ce.y; b.center[2] = ce.z; b.len = ce.w;
        const int sib = nd.sibling[no];
        c.sibling = sib >= 0 ? rank[sib] : -1;
        const int t = geo.kind ? nd.top[no] : -1;
        if(t >= 0 && geo.kind[t].x == TOPK_PSEUDO) { /* another task's top leaf: skipped by the local walks */
            c.type = SHQ_PSEUDO_NODE_TYPE;
            c.child = geo.kind[t].y;
            c.count = 0;
        } else if(nd.nchild[no] == 0) {
            c.type = SHQ_PARTICLE_NODE_TYPE;
            c.child = nd.lo[no];
            c.count = nd.hi[no] - nd.lo[no];
            if(pfather)
                for(int k = nd.lo[no]; k < nd.hi[no]; k++)
                    pfather[idx[k]] = r;
        } else {
            c.type = SHQ_NODE_NODE_TYPE;
            c.child = rank[nd.firstchild[no]];
            c.count = 0;
        }
        hmax = nd.hmax[no];
    }
    A[r] = a;
    B[r] = b;
    C[r] = c;
    H[r] = hmax;
    NodeG g;
    memset(&g, 0, sizeof(g));
    for(int k = 0; k < 3; k++) {
        g.cofm[k] = a.cofm[k];
        g.center[k] = b.center[k];
    }
    g.mass = a.mass;
    g.len = b.len;
    g.sibling = c.sibling; g.child = c.child; g.type = c.type; g.count = c.count;
    g.bhlim = 0; /* filled per walk, like rcuthl */
    g.mlen2 = g.mass * g.len * g.len; /* (mass * len) * len, as shall_we_open_node evaluates it */
    g.inside = 0.6 * g.len;
    g.rcut2 = 0;
    g.wraplim = fmax(0.5 * Box - 0.5 * g.len, 0.0); /* the root (1.001 Box): zero = always wrap; fill_rcuthl_kernel adds the interior flag */
    G[r] = g;
}

__global__ void tb_leafcopy_kernel(long long n, long long npad, const int32_t *__restrict__ idx, const double4 *__restrict__ posm,
                                   double4 *posm_leaf, int32_t *leaf_pidx)
{
    const long long k = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= npad)
        return;
    if(k < n) {
        const int p = idx[k];
        posm_leaf[k] = posm[p];
        leaf_pidx[k] = p;
    } else {
        posm_leaf[k] = make_double4(0, 0, 0, 0);
        leaf_pidx[k] = 0;
    }
}

/* the tree in the reference's NODE format (forcetree.h:38-66), numbered in pre-order from `firstnode` */
__global__ void tb_export_kernel(int nn, long long firstnode, const int32_t *__restrict__ order, const int32_t *__restrict__ rank,
                                 TbNodes nd, const int32_t *__restrict__ idx, shq_node *out, TbGeo geo, long long lastnode)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if(r >= nn)
        return;
    const int no = order[r];
    shq_node o;
    const double4 m = nd.mom[no], ce = nd.cen[no];
    const int sib = nd.sibling[no], par = nd.parent[no];
    o.sibling = sib >= 0 ? (int32_t) (firstnode + rank[sib]) : -1;
    o.father = par >= 0 ? (int32_t) (firstnode + rank[par]) : -1;
    o.len = ce.w;
    o.center[0] = ce.x; o.center[1] = ce.y; o.center[2] = ce.z;
    o.cofm[0] = m.x; o.cofm[1] = m.y; o.cofm[2] = m.z;
    o.mass = m.w;
    o.hmax = nd.hmax[no];
    for(int j = 0; j < SHQ_NMAXCHILD; j++)
        o.suns[j] = -1;
    const int nch = nd.nchild[no];
    if(nch == 0) {
        const int cnt = nd.hi[no] - nd.lo[no];
        for(int k = 0; k < cnt && k < SHQ_NMAXCHILD; k++)
            o.suns[k] = idx[nd.lo[no] + k];
        o.noccupied = cnt;
        o.flags = (unsigned) SHQ_PARTICLE_NODE_TYPE << 3;
    } else {
        for(int j = 0; j < nch; j++)
            o.suns[j] = (int32_t) (firstnode + rank[nd.firstchild[no] + j]);
        o.noccupied = 1 << 16; /* NODEFULL */
        o.flags = (unsigned) SHQ_NODE_NODE_TYPE << 3;
    }
    if(geo.kind) {
        const int t = nd.top[no];
        if(t >= 0) {
            const int k = geo.kind[t].x;
            o.flags |= 2u; /* TopLevel */
            if(k == TOPK_INTERNAL)
                o.flags |= 1u; /* InternalTopLevel */
            if(k == TOPK_PSEUDO) {
                o.flags = ((unsigned) SHQ_PSEUDO_NODE_TYPE << 3) | 2u;
                o.suns[0] = (int32_t) (lastnode + geo.kind[t].y); /* forcetree.cpp:905 */
                o.noccupied = 0;
            }
        }
        if(nd.hi[no] > nd.lo[no])
            o.flags |= 4u; /* DependsOnLocalMass */
    } else if(par < 0)
        o.flags |= 2u | 4u; /* TopLevel, DependsOnLocalMass */
    out[r] = o;
}

__global__ void tb_father_kernel(long long np, long long firstnode, const int32_t *__restrict__ pfather, int32_t *out)
{
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(i < np)
        out[i] = pfather[i] >= 0 ? (int32_t) (firstnode + pfather[i]) : -1;
}

inline unsigned nblk(long long n, int t = 256) { return (unsigned) ((n + t - 1) / t); }

int reserve_nodes(shq_context *ctx, size_t cap)
{
    TreeBuildBufs &b = ctx->tb;
    SHQ_TRY(b.lo.reserve(cap));
    SHQ_TRY(b.hi.reserve(cap));
    SHQ_TRY(b.parent.reserve(cap));
    SHQ_TRY(b.sibling.reserve(cap));
    SHQ_TRY(b.firstchild.reserve(cap));
    SHQ_TRY(b.nchild.reserve(cap));
    SHQ_TRY(b.level.reserve(cap));
    SHQ_TRY(b.cen.reserve(cap));
    SHQ_TRY(b.mom.reserve(cap));
    SHQ_TRY(b.hmax.reserve(cap));
    SHQ_TRY(b.frontier[0].reserve(cap));
    SHQ_TRY(b.frontier[1].reserve(cap));
    SHQ_TRY(b.okeys[0].reserve(cap));
    SHQ_TRY(b.okeys[1].reserve(cap));
    SHQ_TRY(b.order[0].reserve(cap));
    SHQ_TRY(b.order[1].reserve(cap));
    SHQ_TRY(b.rank.reserve(cap));
    SHQ_TRY(b.top.reserve(cap));
    SHQ_TRY(b.path.reserve(cap));
    return SHQ_OK;
}

TbNodes node_view(shq_context *ctx)
{
    TreeBuildBufs &b = ctx->tb;
    TbNodes v;
    v.lo = b.lo.ptr; v.hi = b.hi.ptr; v.parent = b.parent.ptr; v.sibling = b.sibling.ptr;
    v.firstchild = b.firstchild.ptr; v.nchild = b.nchild.ptr; v.level = b.level.ptr;
    v.cen = b.cen.ptr; v.mom = b.mom.ptr; v.hmax = b.hmax.ptr;
    v.top = b.top.ptr; v.path = b.path.ptr;
    return v;
}

} // namespace

struct BelowLimit {
    int limit;
    __device__ bool operator()(const int32_t &p) const { return p < limit; }
};

/* Walk targets in tree (leaf) order: the particles of the tree that are this rank's own, in the order
 * the leaves list them.  64 consecutive entries are 8-16 neighbouring leaves, so a wave's targets stay
 * spatially compact even when the particle index order has gone stale after many drifts (the reference
 * re-sorts its particle array along the Peano-Hilbert curve at every domain decomposition instead). */
/* Peano-Hilbert key of a position (Skilling's axes-to-transpose transform, as host/hostapi.cpp shqh_hilbert_order):
 * consecutive keys are spatial neighbours — an octant (Morton) order jumps across the box at every octant boundary —
 * so 64 consecutive targets make a more compact group and a shorter union walk. */
__global__ void hilbert_key_kernel(long long n, const int32_t *__restrict__ targets, const double4 *__restrict__ posm, double L,
                                   unsigned long long *keys)
{
    const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= n)
        return;
    const double4 p = posm[targets ? (long long) targets[t] : t];
    const int bits = 21;
    const double scale = (double) (1 << bits) / (L * 1.001);
    const double xs[3] = {p.x, p.y, p.z};
    unsigned int X[3];
    for(int j = 0; j < 3; j++) {
        double v = (xs[j] + L / 2000.) * scale;
        v = v < 0 ? 0 : v;
        v = v > (double) ((1 << bits) - 1) ? (double) ((1 << bits) - 1) : v;
        X[j] = (unsigned int) v;
    }
    const unsigned int M = 1u << (bits - 1);
    for(unsigned int Q = M; Q > 1; Q >>= 1) {
        const unsigned int P = Q - 1;
        for(int j = 0; j < 3; j++) {
            if(X[j] & Q)
                X[0] ^= P;
            else {
                const unsigned int tt = (X[0] ^ X[j]) & P;
                X[0] ^= tt;
                X[j] ^= tt;
            }
        }
    }
    X[1] ^= X[0];
    X[2] ^= X[1];
    unsigned int tt = 0;
    for(unsigned int Q = M; Q > 1; Q >>= 1)
        if(X[2] & Q)
            tt ^= Q - 1;
    for(int j = 0; j < 3; j++)
        X[j] ^= tt;
    unsigned long long key = 0;
    for(int b = bits - 1; b >= 0; b--)
        key = (key << 3) | ((unsigned long long) ((X[0] >> b) & 1) << 2) | ((unsigned long long) ((X[1] >> b) & 1) << 1) |
              (unsigned long long) ((X[2] >> b) & 1);
    keys[t] = key;
}

namespace {
__global__ void iota64_kernel(long long n, long long *out)
{
    const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(t < n)
        out[t] = t;
}
} // namespace

extern "C" int shq_hilbert_order(shq_context *ctx, const double *d_posm, int64_t n, double BoxSize, int64_t *d_order)
{
    SHQ_CHECK(ctx && (n == 0 || (d_posm && d_order)), SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(n >= 0 && BoxSize > 0, SHQ_ERR_INVALID, "hilbert_order: n = %ld, BoxSize = %g", (long) n, BoxSize);
    if(n == 0)
        return SHQ_OK;
    SHQ_HIP(hipSetDevice(ctx->device));
    TreeBuildBufs &b = ctx->tb;
    SHQ_TRY(b.keys[0].reserve((size_t) n));
    SHQ_TRY(b.keys[1].reserve((size_t) n));
    SHQ_TRY(ctx->hilb_iota.reserve((size_t) n));
    hilbert_key_kernel<<<dim3(nblk(n)), dim3(256), 0, ctx->stream>>>(n, nullptr, (const double4 *) d_posm, BoxSize, b.keys[0].ptr);
    SHQ_HIP(hipGetLastError());
    iota64_kernel<<<dim3(nblk(n)), dim3(256), 0, ctx->stream>>>(n, ctx->hilb_iota.ptr);
    SHQ_HIP(hipGetLastError());
    size_t tmp = 0;
    SHQ_HIP(rocprim::radix_sort_pairs(nullptr, tmp, b.keys[0].ptr, b.keys[1].ptr, ctx->hilb_iota.ptr, (long long *) d_order, (size_t) n, 0, 63, ctx->stream));
    SHQ_TRY(b.temp.reserve(tmp + 16));
    SHQ_HIP(rocprim::radix_sort_pairs(b.temp.ptr, tmp, b.keys[0].ptr, b.keys[1].ptr, ctx->hilb_iota.ptr, (long long *) d_order, (size_t) n, 0, 63, ctx->stream));
    return SHQ_OK;
}

int shq_build_tree_targets(shq_context *ctx)
{
    if(ctx->have_tree_targets)
        return SHQ_OK;
    SHQ_CHECK(ctx->have_tree, SHQ_ERR_STATE, "tree-order targets: no tree");
    const long long n = ctx->ntreeparts;
    const int limit = (int) (ctx->nlocal);
    TreeBuildBufs &b = ctx->tb;
    SHQ_TRY(ctx->tree_targets.reserve((size_t) (n > 0 ? n : 1)));
    SHQ_TRY(b.counters.reserve(4));
    ctx->ntree_targets = 0;
    if(n > 0) {
        size_t tmp = 0;
        unsigned long long *d_count = b.counters.ptr + 2;
        SHQ_HIP(rocprim::select(nullptr, tmp, ctx->leaf_pidx.ptr, ctx->tree_targets.ptr, d_count, (size_t) n, BelowLimit{limit}, ctx->stream));
        SHQ_TRY(b.temp.reserve(tmp + 16));
        SHQ_HIP(rocprim::select(b.temp.ptr, tmp, ctx->leaf_pidx.ptr, ctx->tree_targets.ptr, d_count, (size_t) n, BelowLimit{limit}, ctx->stream));
        unsigned long long h = 0;
        SHQ_HIP(hipMemcpyAsync(&h, d_count, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
        SHQ_HIP(hipStreamSynchronize(ctx->stream));
        ctx->ntree_targets = (int64_t) h;
        /* leaf order is an octant order: re-sort the targets along the Peano-Hilbert curve */
        const long long nt = (long long) h;
        if(nt > 64 && ctx->treeBox > 0) {
            SHQ_TRY(b.keys[0].reserve((size_t) nt));
            SHQ_TRY(b.keys[1].reserve((size_t) nt));
            SHQ_TRY(b.idx[0].reserve((size_t) nt));
            hilbert_key_kernel<<<dim3(nblk(nt)), dim3(256), 0, ctx->stream>>>(nt, ctx->tree_targets.ptr, ctx->posm.ptr, ctx->treeBox, b.keys[0].ptr);
            SHQ_HIP(hipGetLastError());
            SHQ_HIP(rocprim::radix_sort_pairs(nullptr, tmp, b.keys[0].ptr, b.keys[1].ptr, ctx->tree_targets.ptr, b.idx[0].ptr, (size_t) nt, 0, 63,
                                              ctx->stream));
            SHQ_TRY(b.temp.reserve(tmp + 16));
            SHQ_HIP(rocprim::radix_sort_pairs(b.temp.ptr, tmp, b.keys[0].ptr, b.keys[1].ptr, ctx->tree_targets.ptr, b.idx[0].ptr, (size_t) nt, 0, 63,
                                              ctx->stream));
            SHQ_HIP(hipMemcpyAsync(ctx->tree_targets.ptr, b.idx[0].ptr, sizeof(int32_t) * (size_t) nt, hipMemcpyDeviceToDevice, ctx->stream));
        }
    }
    ctx->have_tree_targets = true;
    ctx->tree_targets_ntree = ctx->ntreeparts;
    ctx->tree_targets_age = 0;
    return SHQ_OK;
}

/* dom: build under a domain decomposition (geo tables already on the device) */
static int tree_build_impl(shq_context *ctx, double BoxSize, int mask, const int32_t *active, int64_t nactive,
                           shq_tree_build_stats *stats, const bool dom)
{
    SHQ_CHECK(ctx, SHQ_ERR_INVALID, "null context");
    SHQ_CHECK(ctx->have_parts, SHQ_ERR_STATE, "tree_build: upload particles first");
    SHQ_CHECK(BoxSize > 0, SHQ_ERR_INVALID, "tree_build: BoxSize must be > 0");
    SHQ_CHECK(!active || nactive >= 0, SHQ_ERR_INVALID, "tree_build: bad active list");
    SHQ_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    TreeBuildBufs &b = ctx->tb;
    const long long np = ctx->numpart;
    const int32_t *d_cand = nullptr;
    int64_t ncand_ = 0;
    SHQ_TRY(shq_resolve_active(ctx, active, nactive, np, &d_cand, &ncand_));
    const long long ncand = ncand_;
    SHQ_CHECK(ncand < (1ll << 31) - 64, SHQ_ERR_INVALID, "tree_build: too many particles");
    ctx->have_tree = false;
    /* The tree-order target list (SHQ_WALK_TREE_ORDER: the tree's own particles along the Peano-Hilbert curve of their positions when it
     * was made) is an ORDER of targets, never a result: any order gives every particle the same bits.  The reference refreshes its
     * particle order at domain decompositions, not every step (domain.cpp:268); a rebuild over the same particle set keeps the list for
     * up to tree_targets_refresh builds (default 8; 1.8 ms per step at 256^3 otherwise), and drops it when the build is another set
     * (active list, mask) or the number of tree particles changes (checked after the build, below). */
    const bool keep_targets = ctx->have_tree_targets && !active && mask == ctx->tree_targets_mask && np == ctx->tree_targets_np &&
                              ctx->tree_targets_age + 1 < ctx->tree_targets_refresh;
    ctx->have_tree_targets = false;
    SHQ_HIP(hipEventRecord(ctx->ev_begin[16], st));

    /* 1. keys */
    const size_t pcap = (size_t) (ncand > 0 ? ncand : 1);
    SHQ_TRY(b.keys[0].reserve(pcap));
    SHQ_TRY(b.keys[1].reserve(pcap));
    SHQ_TRY(b.idx[0].reserve(pcap));
    SHQ_TRY(b.idx[1].reserve(pcap));
    SHQ_TRY(b.counters.reserve(4));
    SHQ_HIP(hipMemsetAsync(b.counters.ptr, 0, sizeof(unsigned long long) * 4, st));
    if(ncand > 0) {
        tb_key_kernel<<<dim3(nblk(ncand)), dim3(256), 0, st>>>(ncand, d_cand, ctx->posm.ptr, ctx->pflags.ptr, mask, BoxSize,
                                                               b.keys[0].ptr, b.idx[0].ptr);
        SHQ_HIP(hipGetLastError());
    }
    unsigned long long h_nvalid = 0;

    /* 2. stable sort by key: depth-first leaf order, ties in candidate order */
    if(ncand > 0) {
        size_t tmp = 0;
        SHQ_HIP(rocprim::radix_sort_pairs(nullptr, tmp, b.keys[0].ptr, b.keys[1].ptr, b.idx[0].ptr, b.idx[1].ptr, (size_t) ncand, 0, 64, st));
        SHQ_TRY(b.temp.reserve(tmp + 16));
        SHQ_HIP(rocprim::radix_sort_pairs(b.temp.ptr, tmp, b.keys[0].ptr, b.keys[1].ptr, b.idx[0].ptr, b.idx[1].ptr, (size_t) ncand, 0, 64, st));
        tb_count_kernel<<<1, 1, 0, st>>>(b.keys[1].ptr, ncand, b.counters.ptr);
        SHQ_HIP(hipMemcpyAsync(&h_nvalid, b.counters.ptr, sizeof(h_nvalid), hipMemcpyDeviceToHost, st));
    }
    SHQ_HIP(hipStreamSynchronize(st));
    const long long n = (long long) h_nvalid;
    const unsigned long long *keys = b.keys[1].ptr;
    const int32_t *seq = b.idx[1].ptr; /* candidate sequence numbers in key order */
    int32_t *idx = b.idx[0].ptr;       /* particle indices in leaf order, filled once the leaves are known */

    SHQ_CHECK(!(dom && ctx->dom_kind[0] == TOPK_PSEUDO && n > 0), SHQ_ERR_INVALID,
              "tree_build: the whole box belongs to another task but this rank holds particles (Bad topleaf, forcetree.cpp:807)");
    /* 3. nodes, breadth first */
    TbGeo geo = {nullptr, nullptr, nullptr};
    if(dom)
        geo = TbGeo{b.geo_child[0].ptr, b.geo_child[1].ptr, b.geo_kind.ptr};
    int level_start[TB_LEVELS + 3];
    int nn = 0, maxdepth = 0, h_err = 0;
    size_t cap = (size_t) (0.6 * (double) n) + 4096;
    SHQ_TRY(b.state.reserve((sizeof(TbState) + sizeof(int) - 1) / sizeof(int)));
    TbState *d_state = reinterpret_cast<TbState *>(b.state.ptr);
    for(int attempt = 0;; attempt++) {
        SHQ_CHECK(attempt < 5, SHQ_ERR_NOMEM, "tree_build: node pool overflow");
        SHQ_CHECK(cap < (1ull << 31) - 64, SHQ_ERR_NOMEM, "tree_build: node pool beyond 2^31 nodes");
        SHQ_TRY(reserve_nodes(ctx, cap));
        SHQ_TRY(b.bounds.reserve(9 * cap));
        SHQ_TRY(b.packed[0].reserve(cap + 1));
        TbNodes nd = node_view(ctx);
        tb_root_kernel<<<1, 1, 0, st>>>(nd, (int) n, BoxSize, dom ? 1 : 0);
        const int nf0 = (n > SHQ_NMAXCHILD || (dom && ctx->dom_kind[0] == TOPK_INTERNAL)) ? 1 : 0;
        tb_state_init_kernel<<<1, 1, 0, st>>>(d_state, nf0);
        if(nf0)
            SHQ_HIP(hipMemsetAsync(b.frontier[0].ptr, 0, sizeof(int32_t), st)); /* frontier = {root} */
        /* every level queued, no host round trip: a level whose frontier is empty costs three launches that find nothing to do */
        const dim3 lg(n > (1 << 20) ? 2048u : 256u), lb(256);
        int fsel = 0;
        for(int level = 0; nf0 && level < TB_LEVELS; level++) {
            if(dom) {
                tb_split_kernel<true><<<lg, lb, 0, st>>>(d_state, b.frontier[fsel].ptr, nd, keys, level, b.bounds.ptr, b.packed[0].ptr, geo);
                tb_children_kernel<true><<<lg, lb, 0, st>>>(d_state, b.frontier[fsel].ptr, nd, level, b.bounds.ptr, b.packed[0].ptr, (int) cap,
                                                            b.frontier[fsel ^ 1].ptr, reinterpret_cast<int *>(b.counters.ptr + 1), geo);
            } else {
                tb_split_kernel<false><<<lg, lb, 0, st>>>(d_state, b.frontier[fsel].ptr, nd, keys, level, b.bounds.ptr, b.packed[0].ptr, geo);
                tb_children_kernel<false><<<lg, lb, 0, st>>>(d_state, b.frontier[fsel].ptr, nd, level, b.bounds.ptr, b.packed[0].ptr, (int) cap,
                                                             b.frontier[fsel ^ 1].ptr, reinterpret_cast<int *>(b.counters.ptr + 1), geo);
            }
            tb_advance_kernel<<<1, 1, 0, st>>>(d_state, level);
            fsel ^= 1;
        }
        SHQ_HIP(hipGetLastError());
        TbState hs;
        SHQ_HIP(hipMemcpyAsync(&hs, d_state, sizeof(hs), hipMemcpyDeviceToHost, st));
        SHQ_HIP(hipMemcpyAsync(&h_err, b.counters.ptr + 1, sizeof(int), hipMemcpyDeviceToHost, st)); /* one round trip for both */
        SHQ_HIP(hipStreamSynchronize(st));
        if(hs.overflow) {
            cap *= 2;
            continue;
        }
        nn = hs.nn;
        maxdepth = hs.maxdepth;
        for(int l = 0; l < TB_LEVELS + 3; l++)
            level_start[l] = hs.level_start[l];
        break;
    }
    SHQ_CHECK(h_err != 3, SHQ_ERR_INVALID, "tree_build: a particle of this rank lies in a top leaf of another task (Bad topleaf, forcetree.cpp:807)");
    SHQ_CHECK(h_err == 0, SHQ_ERR_INVALID, "tree_build: more than %d particles closer than Box/2^%d: deeper than the device build supports",
              SHQ_NMAXCHILD, TB_LEVELS);
    TbNodes nd = node_view(ctx);
    tb_leafsort_kernel<<<dim3(nblk(nn)), dim3(256), 0, st>>>(nn, nd, seq, d_cand, idx);
    SHQ_HIP(hipGetLastError());

    /* 4. moments, deepest level first */
    const double *d_hsml = (ctx->have_sph || ctx->have_dyn) ? ctx->hsml.ptr : nullptr;
    for(int l = maxdepth; l >= 0; l--) {
        const int first = level_start[l], last = level_start[l + 1];
        if(last > first)
            tb_moments_kernel<<<dim3(nblk(last - first)), dim3(256), 0, st>>>(first, last, nd, idx, ctx->posm.ptr, ctx->pflags.ptr, d_hsml);
    }
    SHQ_HIP(hipGetLastError());

    /* 5. pre-order ranks and the walk pool */
    if(!dom) {
        tb_orderkey_kernel<<<dim3(nblk(nn)), dim3(256), 0, st>>>(nn, nd, b.okeys[0].ptr, b.order[0].ptr);
        size_t tmp = 0;
        SHQ_HIP(rocprim::radix_sort_pairs(nullptr, tmp, b.okeys[0].ptr, b.okeys[1].ptr, b.order[0].ptr, b.order[1].ptr, (size_t) nn, 0, 40, st));
        SHQ_TRY(b.temp.reserve(tmp + 16));
        SHQ_HIP(rocprim::radix_sort_pairs(b.temp.ptr, tmp, b.okeys[0].ptr, b.okeys[1].ptr, b.order[0].ptr, b.order[1].ptr, (size_t) nn, 0, 40, st));
        tb_rank_kernel<<<dim3(nblk(nn)), dim3(256), 0, st>>>(nn, b.order[1].ptr, b.rank.ptr);
    } else {
        size_t tmp = 0, tmp2 = 0;
        SHQ_HIP(rocprim::radix_sort_pairs(nullptr, tmp, b.okeys[0].ptr, b.okeys[1].ptr, b.order[0].ptr, b.order[1].ptr, (size_t) nn, 0, 8, st));
        SHQ_HIP(rocprim::radix_sort_pairs(nullptr, tmp2, b.okeys[0].ptr, b.okeys[1].ptr, b.order[0].ptr, b.order[1].ptr, (size_t) nn, 0, 64, st));
        SHQ_TRY(b.temp.reserve(std::max(tmp, tmp2) + 16));
        tb_levelkey_kernel<<<dim3(nblk(nn)), dim3(256), 0, st>>>(nn, nd, b.okeys[0].ptr, b.order[0].ptr);
        SHQ_HIP(rocprim::radix_sort_pairs(b.temp.ptr, tmp, b.okeys[0].ptr, b.okeys[1].ptr, b.order[0].ptr, b.order[1].ptr, (size_t) nn, 0, 8, st));
        tb_pathkey_kernel<<<dim3(nblk(nn)), dim3(256), 0, st>>>(nn, nd, b.order[1].ptr, b.okeys[0].ptr);
        SHQ_HIP(rocprim::radix_sort_pairs(b.temp.ptr, tmp2, b.okeys[0].ptr, b.okeys[1].ptr, b.order[1].ptr, b.order[0].ptr, (size_t) nn, 0, 64, st));
        /* the pre-order is in order[0]; the rest of the file reads order[1] */
        SHQ_HIP(hipMemcpyAsync(b.order[1].ptr, b.order[0].ptr, sizeof(int32_t) * (size_t) nn, hipMemcpyDeviceToDevice, st));
        tb_rank_kernel<<<dim3(nblk(nn)), dim3(256), 0, st>>>(nn, b.order[1].ptr, b.rank.ptr);
    }
    SHQ_TRY(ctx->nodeA.reserve((size_t) nn + 1));
    SHQ_TRY(ctx->nodeB.reserve((size_t) nn + 1));
    SHQ_TRY(ctx->nodeC.reserve((size_t) nn + 1));
    SHQ_TRY(ctx->nodeG.reserve((size_t) nn + 1));
    SHQ_TRY(ctx->node_hmax.reserve((size_t) nn + 1));
    SHQ_TRY(ctx->pfather.reserve((size_t) (np > 0 ? np : 1)));
    SHQ_HIP(hipMemsetAsync(ctx->pfather.ptr, 0xff, sizeof(int32_t) * (size_t) (np > 0 ? np : 1), st));
    tb_pack_kernel<<<dim3(nblk(nn + 1)), dim3(256), 0, st>>>(nn, b.order[1].ptr, b.rank.ptr, nd, idx, ctx->nodeA.ptr, ctx->nodeB.ptr,
                                                              ctx->nodeC.ptr, ctx->nodeG.ptr, ctx->node_hmax.ptr, ctx->pfather.ptr, BoxSize, geo);
    const long long npad = n + SHQ_NMAXCHILD;
    SHQ_TRY(ctx->posm_leaf.reserve((size_t) npad));
    SHQ_TRY(ctx->leaf_pidx.reserve((size_t) npad));
    tb_leafcopy_kernel<<<dim3(nblk(npad)), dim3(256), 0, st>>>(n, npad, idx, ctx->posm.ptr, ctx->posm_leaf.ptr, ctx->leaf_pidx.ptr);
    SHQ_HIP(hipGetLastError());
    SHQ_HIP(hipEventRecord(ctx->ev_end[16], st));
    SHQ_HIP(hipStreamSynchronize(st));

    ctx->node_order.clear();    /* empty = identity: a downloaded tree is numbered in pool order (filling 7 M entries on the host cost a
                                   resident step 2.9 ms between the build and the walk) */
    ctx->node_rank.clear();     /* identity */
    ctx->numnodes = nn;
    ctx->firstnode = np;
    ctx->root = 0;
    ctx->ntreeparts = n;
    ctx->treeBox = BoxSize;
    ctx->have_tree = true;
    ctx->node_rcut = -1;
    ctx->have_group_aux = false;
    ctx->have_father = true;
    ctx->tb_built = true;
    ctx->tb_domain = dom;
    if(keep_targets && ctx->ntreeparts == ctx->tree_targets_ntree) {
        ctx->have_tree_targets = true;
        ctx->tree_targets_age++;
    } else
        ctx->tree_targets_age = 0;
    ctx->tree_targets_mask = active ? -1 : mask;
    ctx->tree_targets_np = np;
    SHQ_TRY(shq_walk_prereserve(ctx));
    ctx->have_toptree = false;
    if(stats) {
        stats->nparticles = n;
        stats->numnodes = nn;
        stats->maxdepth = maxdepth;
        float ms = 0;
        (void) hipEventElapsedTime(&ms, ctx->ev_begin[16], ctx->ev_end[16]);
        stats->build_ms = ms;
    }
    return SHQ_OK;
}

extern "C" int shq_tree_build(shq_context *ctx, double BoxSize, int mask, const int32_t *active, int64_t nactive,
                              shq_tree_build_stats *stats)
{
    return tree_build_impl(ctx, BoxSize, mask, active, nactive, stats, false);
}

namespace {

/* per TopNode: rec[9 t ...] = cofm[3], mass, hmax, center[3], len; rank[t] = pre-order number */
__global__ void tb_topgather_kernel(int nn, TbNodes nd, const int32_t *__restrict__ rank, double *rec, int32_t *toprank)
{
    const int no = blockIdx.x * blockDim.x + threadIdx.x;
    if(no >= nn)
        return;
    const int t = nd.top[no];
    if(t < 0)
        return;
    const double4 m = nd.mom[no], c = nd.cen[no];
    double *o = rec + 9 * (size_t) t;
    o[0] = m.x; o[1] = m.y; o[2] = m.z; o[3] = m.w; o[4] = nd.hmax[no];
    o[5] = c.x; o[6] = c.y; o[7] = c.z; o[8] = c.w;
    toprank[t] = rank[no];
}

/* moments of the top-level nodes after the exchange: into the build arrays (for a later download) and the walk pool */
__global__ void tb_topscatter_kernel(int nn, TbNodes nd, const int32_t *__restrict__ rank, const double *__restrict__ rec, NodeA *A, NodeG *G, double *H)
{
    const int no = blockIdx.x * blockDim.x + threadIdx.x;
    if(no >= nn)
        return;
    const int t = nd.top[no];
    if(t < 0)
        return;
    const double *o = rec + 9 * (size_t) t;
    nd.mom[no] = make_double4(o[0], o[1], o[2], o[3]);
    nd.hmax[no] = o[4];
    const int r = rank[no];
    NodeA a = A[r];
    a.cofm[0] = o[0]; a.cofm[1] = o[1]; a.cofm[2] = o[2]; a.mass = o[3];
    A[r] = a;
    NodeG g = G[r];
    g.cofm[0] = o[0]; g.cofm[1] = o[1]; g.cofm[2] = o[2]; g.mass = o[3];
    g.mlen2 = g.mass * g.len * g.len;
    G[r] = g;
    H[r] = o[4];
}

} // namespace

extern "C" int shq_tree_build_domain(shq_context *ctx, double BoxSize, int mask, const int32_t *active, int64_t nactive,
                                     const shq_topnode_geo *topnodes, int ntopnodes, shq_topleaf *topleaves, int ntopleaves, int ThisTask,
                                     int64_t firstnode, shq_topleaf_moments *local_moments, shq_tree_build_stats *stats)
{
    SHQ_CHECK(ctx && topnodes && topleaves, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ntopnodes >= 1 && ntopleaves >= 1, SHQ_ERR_INVALID, "tree_build_domain: empty top tree");
    SHQ_CHECK(ctx->have_parts && firstnode >= ctx->numpart, SHQ_ERR_INVALID, "tree_build_domain: upload particles first; firstnode must be >= NumPart");
    SHQ_HIP(hipSetDevice(ctx->device));
    /* validate the table: every daughter in range, every leaf index in range and used once, reached once from the root */
    std::vector<int32_t> kind((size_t) ntopnodes, -1), seen((size_t) ntopnodes, 0), leafseen((size_t) ntopleaves, 0);
    std::vector<int4> c0((size_t) ntopnodes), c1((size_t) ntopnodes);
    std::vector<int2> kk((size_t) ntopnodes);
    {
        std::vector<int32_t> stack(1, 0);
        while(!stack.empty()) {
            const int32_t t = stack.back();
            stack.pop_back();
            SHQ_CHECK(t >= 0 && t < ntopnodes, SHQ_ERR_INVALID, "tree_build_domain: daughter %d outside the TopNodes table", t);
            SHQ_CHECK(!seen[t], SHQ_ERR_INVALID, "tree_build_domain: TopNode %d is reached twice", t);
            seen[t] = 1;
            const shq_topnode_geo &g = topnodes[t];
            const bool leaf = g.daughter[0] < 0;
            if(leaf) {
                for(int s = 0; s < 8; s++)
                    SHQ_CHECK(g.daughter[s] < 0, SHQ_ERR_INVALID, "tree_build_domain: TopNode %d has some daughters only", t);
                SHQ_CHECK(g.leaf >= 0 && g.leaf < ntopleaves && !leafseen[g.leaf], SHQ_ERR_INVALID, "tree_build_domain: bad leaf index %d of TopNode %d", g.leaf, t);
                leafseen[g.leaf] = 1;
                kind[t] = topleaves[g.leaf].Task == ThisTask ? TOPK_LOCAL : TOPK_PSEUDO;
            } else {
                kind[t] = TOPK_INTERNAL;
                for(int s = 7; s >= 0; s--)
                    stack.push_back(g.daughter[s]);
            }
            c0[t] = make_int4(g.daughter[0], g.daughter[1], g.daughter[2], g.daughter[3]);
            c1[t] = make_int4(g.daughter[4], g.daughter[5], g.daughter[6], g.daughter[7]);
            kk[t] = make_int2(kind[t], leaf ? g.leaf : -1);
        }
    }
    for(int t = 0; t < ntopnodes; t++)
        SHQ_CHECK(seen[t], SHQ_ERR_INVALID, "tree_build_domain: TopNode %d is not reachable from the root", t);
    for(int l = 0; l < ntopleaves; l++)
        SHQ_CHECK(leafseen[l], SHQ_ERR_INVALID, "tree_build_domain: top leaf %d belongs to no TopNode", l);
    TreeBuildBufs &b = ctx->tb;
    SHQ_TRY(b.geo_child[0].reserve((size_t) ntopnodes));
    SHQ_TRY(b.geo_child[1].reserve((size_t) ntopnodes));
    SHQ_TRY(b.geo_kind.reserve((size_t) ntopnodes));
    SHQ_HIP(hipMemcpy(b.geo_child[0].ptr, c0.data(), sizeof(int4) * (size_t) ntopnodes, hipMemcpyHostToDevice));
    SHQ_HIP(hipMemcpy(b.geo_child[1].ptr, c1.data(), sizeof(int4) * (size_t) ntopnodes, hipMemcpyHostToDevice));
    SHQ_HIP(hipMemcpy(b.geo_kind.ptr, kk.data(), sizeof(int2) * (size_t) ntopnodes, hipMemcpyHostToDevice));
    ctx->dom_geo.assign(topnodes, topnodes + ntopnodes);
    ctx->dom_kind = kind;
    ctx->dom_thistask = ThisTask;
    ctx->dom_leaf_task.resize((size_t) ntopleaves);
    for(int l = 0; l < ntopleaves; l++)
        ctx->dom_leaf_task[l] = topleaves[l].Task;
    SHQ_TRY(tree_build_impl(ctx, BoxSize, mask, active, nactive, stats, true));

    /* the top-level nodes back to the host: numbers for TopLeaves[].treenode, moments of this task's leaves */
    const int nn = (int) ctx->numnodes;
    SHQ_TRY(b.topbuf.reserve(9 * (size_t) ntopnodes + (size_t) ntopnodes));
    double *d_rec = b.topbuf.ptr;
    int32_t *d_rank = reinterpret_cast<int32_t *>(b.topbuf.ptr + 9 * (size_t) ntopnodes);
    SHQ_HIP(hipMemsetAsync(d_rank, 0xff, sizeof(int32_t) * (size_t) ntopnodes, ctx->stream));
    tb_topgather_kernel<<<dim3(nblk(nn)), dim3(256), 0, ctx->stream>>>(nn, node_view(ctx), b.rank.ptr, d_rec, d_rank);
    SHQ_HIP(hipGetLastError());
    ctx->dom_rec.resize(9 * (size_t) ntopnodes);
    ctx->dom_rank.resize((size_t) ntopnodes);
    SHQ_HIP(hipMemcpyAsync(ctx->dom_rec.data(), d_rec, sizeof(double) * 9 * (size_t) ntopnodes, hipMemcpyDeviceToHost, ctx->stream));
    SHQ_HIP(hipMemcpyAsync(ctx->dom_rank.data(), d_rank, sizeof(int32_t) * (size_t) ntopnodes, hipMemcpyDeviceToHost, ctx->stream));
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    bool remote = false;
    for(int t = 0; t < ntopnodes; t++) {
        SHQ_CHECK(ctx->dom_rank[t] >= 0, SHQ_ERR_DEVICE, "tree_build_domain: TopNode %d has no tree node", t);
        if(kind[t] == TOPK_INTERNAL)
            continue;
        const int l = topnodes[t].leaf;
        topleaves[l].treenode = (int32_t) (firstnode + ctx->dom_rank[t]);
        if(local_moments) {
            shq_topleaf_moments m = {{0, 0, 0}, 0, 0};
            if(kind[t] == TOPK_LOCAL) {
                const double *o = &ctx->dom_rec[9 * (size_t) t];
                m.s[0] = o[0]; m.s[1] = o[1]; m.s[2] = o[2]; m.mass = o[3]; m.hmax = o[4];
            }
            local_moments[l] = m;
        }
        remote = remote || kind[t] == TOPK_PSEUDO;
    }
    ctx->firstnode = firstnode;
    if(!remote) { /* a one-task domain: nothing to exchange, the top tree can be installed now */
        std::vector<shq_topleaf_moments> own((size_t) ntopleaves);
        for(int t = 0; t < ntopnodes; t++)
            if(kind[t] != TOPK_INTERNAL) {
                const double *o = &ctx->dom_rec[9 * (size_t) t];
                own[topnodes[t].leaf] = shq_topleaf_moments{{o[0], o[1], o[2]}, o[3], o[4]};
            }
        SHQ_TRY(shq_tree_set_topleaf_moments(ctx, own.data(), ntopleaves));
    }
    return SHQ_OK;
}

extern "C" int shq_tree_set_topleaf_moments(shq_context *ctx, const shq_topleaf_moments *moments, int ntopleaves)
{
    SHQ_CHECK(ctx && moments, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->have_tree && ctx->tb_built && ctx->tb_domain, SHQ_ERR_STATE, "set_topleaf_moments: no tree from shq_tree_build_domain");
    SHQ_CHECK(ntopleaves == (int) ctx->dom_leaf_task.size(), SHQ_ERR_INVALID, "set_topleaf_moments: %d leaves, the domain has %zu", ntopleaves,
              ctx->dom_leaf_task.size());
    SHQ_HIP(hipSetDevice(ctx->device));
    const int ntop = (int) ctx->dom_geo.size();
    std::vector<double> &rec = ctx->dom_rec;
    /* force_exchange_pseudodata, forcetree.cpp:1186-1198: the other tasks' leaves take the gathered values */
    for(int t = 0; t < ntop; t++)
        if(ctx->dom_kind[t] == TOPK_PSEUDO) {
            const shq_topleaf_moments &m = moments[ctx->dom_geo[t].leaf];
            double *o = &rec[9 * (size_t) t];
            o[0] = m.s[0]; o[1] = m.s[1]; o[2] = m.s[2]; o[3] = m.mass; o[4] = m.hmax;
        }
    /* force_treeupdate_pseudos, forcetree.cpp:1211-1281: post-order over the internal top-level nodes, eight daughters in order */
    {
        std::vector<int32_t> stack(1, 0), post;
        while(!stack.empty()) {
            const int32_t t = stack.back();
            stack.pop_back();
            if(ctx->dom_kind[t] != TOPK_INTERNAL)
                continue;
            post.push_back(t);
            for(int s = 0; s < 8; s++)
                stack.push_back(ctx->dom_geo[t].daughter[s]);
        }
        for(size_t k = post.size(); k-- > 0;) { /* reverse pre-order: daughters before their parent */
            const int32_t t = post[k];
            double mass = 0, c0 = 0, c1 = 0, c2 = 0, hmax = 0;
            for(int s = 0; s < 8; s++) {
                const double *p = &rec[9 * (size_t) ctx->dom_geo[t].daughter[s]];
                mass += p[3];
                c0 += p[3] * p[0];
                c1 += p[3] * p[1];
                c2 += p[3] * p[2];
                if(p[4] > hmax)
                    hmax = p[4];
            }
            double *o = &rec[9 * (size_t) t];
            if(mass) {
                c0 /= mass;
                c1 /= mass;
                c2 /= mass;
            } else {
                c0 = o[5];
                c1 = o[6];
                c2 = o[7];
            }
            o[0] = c0; o[1] = c1; o[2] = c2; o[3] = mass; o[4] = hmax;
        }
    }
    TreeBuildBufs &b = ctx->tb;
    const int nn = (int) ctx->numnodes;
    SHQ_TRY(b.topbuf.reserve(9 * (size_t) ntop + (size_t) ntop));
    SHQ_HIP(hipMemcpyAsync(b.topbuf.ptr, rec.data(), sizeof(double) * 9 * (size_t) ntop, hipMemcpyHostToDevice, ctx->stream));
    tb_topscatter_kernel<<<dim3(nblk(nn)), dim3(256), 0, ctx->stream>>>(nn, node_view(ctx), b.rank.ptr, b.topbuf.ptr, ctx->nodeA.ptr, ctx->nodeG.ptr,
                                                                        ctx->node_hmax.ptr);
    SHQ_HIP(hipGetLastError());
    ctx->node_rcut = -1; /* per-walk node fields are refilled */
    ctx->have_group_aux = false;
    /* the top tree for the export-detection walks (what shq_toptree_upload assembles from a host tree): pre-order over TopNodes */
    std::vector<TopNodeG> h;
    std::vector<int32_t> where((size_t) ntop, -1), sibling_of((size_t) ntop, -1);
    {
        struct Item { int32_t t, sib; };
        std::vector<Item> stack(1, Item{0, -1});
        std::vector<int32_t> order;
        while(!stack.empty()) {
            const Item it = stack.back();
            stack.pop_back();
            where[it.t] = (int32_t) order.size();
            sibling_of[it.t] = it.sib;
            order.push_back(it.t);
            if(ctx->dom_kind[it.t] == TOPK_INTERNAL)
                for(int s = 7; s >= 0; s--)
                    stack.push_back(Item{ctx->dom_geo[it.t].daughter[s], s < 7 ? ctx->dom_geo[it.t].daughter[s + 1] : it.sib});
        }
        h.resize(order.size());
        for(size_t j = 0; j < order.size(); j++) {
            const int32_t t = order[j];
            const double *o = &rec[9 * (size_t) t];
            TopNodeG g;
            memset(&g, 0, sizeof(g));
            for(int k = 0; k < 3; k++) {
                g.cofm[k] = o[k];
                g.center[k] = o[5 + k];
            }
            g.mass = o[3];
            g.hmax = o[4];
            g.len = o[8];
            g.sibling = sibling_of[t] >= 0 ? where[sibling_of[t]] : -1; /* a later node: filled below */
            g.child = -1;
            g.leaf = -1;
            g.kind = ctx->dom_kind[t] == TOPK_INTERNAL ? 0 : (ctx->dom_kind[t] == TOPK_LOCAL ? 1 : 2);
            if(ctx->dom_kind[t] == TOPK_PSEUDO)
                g.leaf = ctx->dom_geo[t].leaf;
            h[j] = g;
        }
        for(size_t j = 0; j < order.size(); j++) { /* links to nodes later in the pre-order */
            const int32_t t = order[j];
            h[j].sibling = sibling_of[t] >= 0 ? where[sibling_of[t]] : -1;
            if(ctx->dom_kind[t] == TOPK_INTERNAL)
                h[j].child = where[ctx->dom_geo[t].daughter[0]];
        }
    }
    std::vector<int2> hl((size_t) ntopleaves);
    for(int t = 0; t < ntop; t++)
        if(ctx->dom_kind[t] != TOPK_INTERNAL)
            hl[ctx->dom_geo[t].leaf] = make_int2(ctx->dom_leaf_task[ctx->dom_geo[t].leaf], (int32_t) (ctx->firstnode + ctx->dom_rank[t]));
    SHQ_TRY(ctx->topnodes.reserve(h.size()));
    SHQ_TRY(ctx->topleaves.reserve(hl.size()));
    SHQ_HIP(hipMemcpyAsync(ctx->topnodes.ptr, h.data(), sizeof(TopNodeG) * h.size(), hipMemcpyHostToDevice, ctx->stream));
    SHQ_HIP(hipMemcpyAsync(ctx->topleaves.ptr, hl.data(), sizeof(int2) * hl.size(), hipMemcpyHostToDevice, ctx->stream));
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    ctx->ntopnodes = (int64_t) h.size();
    ctx->have_toptree = true;
    return SHQ_OK;
}

extern "C" int shq_tree_download(shq_context *ctx, int64_t firstnode, shq_node *nodes, int64_t capacity, int32_t *father,
                                 int64_t *numnodes)
{
    SHQ_CHECK(ctx, SHQ_ERR_INVALID, "null context");
    SHQ_CHECK(ctx->have_tree && ctx->tb_built, SHQ_ERR_STATE, "tree_download: no device-built tree (call shq_tree_build first)");
    const int nn = (int) ctx->numnodes;
    if(numnodes)
        *numnodes = nn;
    SHQ_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    TreeBuildBufs &b = ctx->tb;
    if(nodes) {
        SHQ_CHECK(capacity >= nn, SHQ_ERR_INVALID, "tree_download: capacity %ld < %d nodes", (long) capacity, nn);
        SHQ_CHECK(firstnode >= ctx->numpart && firstnode + nn < (1ll << 31), SHQ_ERR_INVALID, "tree_download: firstnode %ld must be >= NumPart and fit int32", (long) firstnode);
        SHQ_TRY(b.exportbuf.reserve((size_t) nn));
        TbGeo geo = {nullptr, nullptr, nullptr};
        if(ctx->tb_domain)
            geo = TbGeo{b.geo_child[0].ptr, b.geo_child[1].ptr, b.geo_kind.ptr};
        tb_export_kernel<<<dim3(nblk(nn)), dim3(256), 0, st>>>(nn, firstnode, b.order[1].ptr, b.rank.ptr, node_view(ctx), b.idx[0].ptr, b.exportbuf.ptr,
                                                               geo, firstnode + capacity);
        SHQ_HIP(hipGetLastError());
        SHQ_HIP(hipMemcpyAsync(nodes, b.exportbuf.ptr, sizeof(shq_node) * (size_t) nn, hipMemcpyDeviceToHost, st));
    }
    if(father && ctx->numpart > 0) {
        SHQ_TRY(b.idx[1].reserve((size_t) ctx->numpart)); /* the sequence numbers are no longer needed */
        tb_father_kernel<<<dim3(nblk(ctx->numpart)), dim3(256), 0, st>>>(ctx->numpart, firstnode, ctx->pfather.ptr, b.idx[1].ptr);
        SHQ_HIP(hipGetLastError());
        SHQ_HIP(hipMemcpyAsync(father, b.idx[1].ptr, sizeof(int32_t) * (size_t) ctx->numpart, hipMemcpyDeviceToHost, st));
    }
    SHQ_HIP(hipStreamSynchronize(st));
    return SHQ_OK;
}

/* ---- domain_maintain's particle loop (libgadget/domain.cpp:296-330, 347-368) --------------------------------------------------------
 * After the drift: a particle that is still inside the cell of its top leaf keeps it (inside_topleaf: |2 (Pos - center)| <= len on
 * every axis), otherwise TopLeaf = domain_get_topleaf(PEANO(Pos)) — here the descent through the daughter-per-octant table on the
 * integer coordinates PEANO() forms (utils/peano.h:15-21), which is the same leaf; then layoutfunc: the leaf's task, or -1 (stay) for
 * an invalid leaf and for inactive dark matter when no dark-matter tree is wanted.  Garbage is left alone. */
namespace {
__global__ void dom_topleaf_kernel(long long n, const double4 *__restrict__ posm, const uint8_t *__restrict__ pflags, const uint8_t *__restrict__ bin_grav, TbGeo geo,
                                   const double4 *__restrict__ leafbox, const int32_t *__restrict__ leaftask, int ntopleaves, double Box, int dmtree, long long Ti,
                                   int32_t *topleaf, int32_t *target, unsigned long long *nchanged)
{
#pragma clang fp contract(off)
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n)
        return;
    const unsigned f = pflags[i];
    if(f & 1u) {
        if(target)
            target[i] = -1;
        return;
    }
    const int type = f >> 4;
    if(!dmtree && type == 1) {
        const int bin = bin_grav[i];
        const bool active = bin <= 0 || Ti <= 0 || (Ti % (1ll << bin)) == 0; /* is_timebin_active, timestep.cpp:132-139 */
        if(!active) {
            if(target)
                target[i] = -1;
            return;
        }
    }
    const double4 p = posm[i];
    int tl = topleaf[i];
    bool inside = false;
    if(tl >= 0 && tl < ntopleaves) {
        const double4 b = leafbox[tl];
        inside = (fabs(2 * (p.x - b.x)) <= b.w) && (fabs(2 * (p.y - b.y)) <= b.w) && (fabs(2 * (p.z - b.z)) <= b.w);
    }
    if(!inside) {
        const double DomainFac = 1.0 / (Box * 1.001) * 2097152.0;
        const int ix = (int) ((p.x + Box / 2000) * DomainFac), iy = (int) ((p.y + Box / 2000) * DomainFac), iz = (int) ((p.z + Box / 2000) * DomainFac);
        int t = 0;
        for(int l = 0; l < 21 && geo.kind[t].x == TOPK_INTERNAL; l++) {
            const int s = ((ix >> (20 - l)) & 1) + 2 * ((iy >> (20 - l)) & 1) + 4 * ((iz >> (20 - l)) & 1);
            t = geo.child(t, s);
        }
        tl = geo.kind[t].y;
        topleaf[i] = tl;
        const unsigned long long movers = __ballot(1); /* one atomic per wave: millions of adds to one word serialise in the L2 */
        if((int) (threadIdx.x & 63) == __ffsll((long long) movers) - 1)
            atomicAdd(nchanged, (unsigned long long) __popcll(movers));
    }
    if(target)
        target[i] = (tl >= 0 && tl < ntopleaves) ? leaftask[tl] : -1;
}
} // namespace

extern "C" int shq_domain_maintain_topleaf(shq_context *ctx, int dmtree, int64_t Ti_Current, int32_t *d_topleaf, int32_t *d_target, int64_t *nchanged)
{
    SHQ_CHECK(ctx && d_topleaf, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->have_parts && ctx->tb_domain && !ctx->dom_geo.empty() && ctx->dom_rec.size() == 9 * ctx->dom_geo.size(), SHQ_ERR_STATE,
              "domain_maintain_topleaf: needs the domain of a shq_tree_build_domain call");
    SHQ_CHECK(dmtree || ctx->have_dyn || ctx->have_sph, SHQ_ERR_STATE, "domain_maintain_topleaf: the gravity time bins must be resident (shq_dynamics_upload)");
    SHQ_HIP(hipSetDevice(ctx->device));
    const int ntl = (int) ctx->dom_leaf_task.size(), ntn = (int) ctx->dom_geo.size();
    std::vector<double4> box((size_t) ntl);
    for(int t = 0; t < ntn; t++) {
        const int l = ctx->dom_geo[t].leaf;
        if(ctx->dom_geo[t].daughter[0] < 0 && l >= 0 && l < ntl) {
            const double *r = &ctx->dom_rec[9 * (size_t) t];
            box[l] = make_double4(r[5], r[6], r[7], r[8]);
        }
    }
    TreeBuildBufs &b = ctx->tb;
    hipStream_t st = ctx->stream;
    SHQ_TRY(b.topbuf.reserve(4 * (size_t) ntl + (size_t) ntl + 8));
    double4 *d_box = reinterpret_cast<double4 *>(b.topbuf.ptr);
    int32_t *d_task = reinterpret_cast<int32_t *>(b.topbuf.ptr + 4 * (size_t) ntl);
    SHQ_TRY(b.counters.reserve(4));
    SHQ_HIP(hipMemcpyAsync(d_box, box.data(), sizeof(double4) * (size_t) ntl, hipMemcpyHostToDevice, st));
    SHQ_HIP(hipMemcpyAsync(d_task, ctx->dom_leaf_task.data(), sizeof(int32_t) * (size_t) ntl, hipMemcpyHostToDevice, st));
    SHQ_HIP(hipMemsetAsync(b.counters.ptr, 0, sizeof(unsigned long long), st));
    const long long n = ctx->numpart;
    if(n > 0) {
        const TbGeo geo{b.geo_child[0].ptr, b.geo_child[1].ptr, b.geo_kind.ptr};
        dom_topleaf_kernel<<<dim3(nblk(n)), dim3(256), 0, st>>>(n, ctx->posm.ptr, ctx->pflags.ptr, ctx->bin_grav.ptr, geo, d_box, d_task, ntl, ctx->treeBox, dmtree,
                                                               (long long) Ti_Current, d_topleaf, d_target, b.counters.ptr);
        SHQ_HIP(hipGetLastError());
    }
    unsigned long long h = 0;
    SHQ_HIP(hipMemcpyAsync(&h, b.counters.ptr, sizeof(h), hipMemcpyDeviceToHost, st));
    SHQ_HIP(hipStreamSynchronize(st)); /* box / task staging live on this frame */
    if(nchanged)
        *nchanged = (int64_t) h;
    return SHQ_OK;
}
