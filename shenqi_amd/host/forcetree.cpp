/* forcetree.cpp — top-down parallel construction of the reference-format oct-tree. */
#include "forcetree.hpp"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <atomic>
#include <omp.h>

namespace {

struct Build {
    const particle_data *P;
    NODE *N; /* shifted by firstnode */
    int64_t firstnode, lastnode;
    std::atomic<int64_t> nnext;
    int *father;
    std::atomic<int> err{0};
    int *perm, *tmp;

    int64_t alloc(int n)
    {
        int64_t r = nnext.fetch_add(n);
        if(r + n > lastnode) {
            err.store(1);
            return -1;
        }
        return r;
    }
    static void set_type(NODE &nd, unsigned t) { nd.flags = (nd.flags & ~(3u << 3)) | (t << 3); }

    /* node `no` covers perm[lo,hi); its center/len/father/sibling are already set. */
    void fill(int no, int64_t lo, int64_t hi, int depth)
    {
        NODE &nd = N[no];
        const int64_t cnt = hi - lo;
        nd.cofm[0] = nd.cofm[1] = nd.cofm[2] = 0;
        nd.mass = 0;
        nd.hmax = 0;
        for(int j = 0; j < NMAXCHILD; j++)
            nd.suns[j] = -1;
        if(cnt <= NMAXCHILD) {
            /* leaf: forcetree.cpp:352-361 modify_internal_node + :947-966 moments + :985-1005 */
            set_type(nd, SHQ_PARTICLE_NODE_TYPE);
            nd.noccupied = (int) cnt;
            for(int64_t k = 0; k < cnt; k++) {
                const int p = perm[lo + k];
                nd.suns[k] = p;
                if(father)
                    father[p] = no;
                const particle_data &pp = P[p];
                nd.mass += pp.Mass;
                for(int d = 0; d < 3; d++)
                    nd.cofm[d] += pp.Mass * pp.Pos[d];
                if(pp.Type == 0 || pp.Type == 5) {
                    for(int d = 0; d < 3; d++) {
                        double v = fabs(pp.Pos[d] - nd.center[d]) + pp.Hsml - nd.len / 2.;
                        if(v > nd.hmax)
                            nd.hmax = v;
                    }
                }
            }
            if(nd.mass > 0) {
                for(int d = 0; d < 3; d++)
                    nd.cofm[d] /= nd.mass;
            } else {
                for(int d = 0; d < 3; d++)
                    nd.cofm[d] = nd.center[d];
            }
            return;
        }
        if(depth > 80) { /* > NMAXCHILD particles at one place: forcetree.cpp:393-401 */
            err.store(2);
            return;
        }
        /* stable 8-way partition by sub-octant (forcetree.cpp:277-283 get_subnode) */
        int64_t c[9] = {0};
        for(int64_t k = lo; k < hi; k++) {
            const double *pos = P[perm[k]].Pos;
            const int s = (pos[0] > nd.center[0]) + ((pos[1] > nd.center[1]) << 1) + ((pos[2] > nd.center[2]) << 2);
            c[s + 1]++;
        }
        for(int s = 0; s < 8; s++)
            c[s + 1] += c[s];
        {
            int64_t w[8];
            for(int s = 0; s < 8; s++)
                w[s] = lo + c[s];
            for(int64_t k = lo; k < hi; k++) {
                const double *pos = P[perm[k]].Pos;
                const int s = (pos[0] > nd.center[0]) + ((pos[1] > nd.center[1]) << 1) + ((pos[2] > nd.center[2]) << 2);
                tmp[w[s]++] = perm[k];
            }
            memcpy(perm + lo, tmp + lo, sizeof(int) * cnt);
        }
        int nch = 0;
        for(int s = 0; s < 8; s++)
            if(c[s + 1] > c[s])
                nch++;
        const int64_t first = alloc(nch);
        if(first < 0)
            return;
        set_type(nd, SHQ_NODE_NODE_TYPE);
        nd.noccupied = (1 << 16); /* NODEFULL */
        int j = 0;
        int child_of[8];
        int64_t clo[8], chi[8];
        for(int s = 0; s < 8; s++) {
            if(c[s + 1] == c[s])
                continue;
            const int ch = (int) (first + j);
            NODE &cn = N[ch];
            /* forcetree.cpp:302-328 init_internal_node */
            const double lenhalf = 0.25 * nd.len;
            cn.len = 0.5 * nd.len;
            for(int d = 0; d < 3; d++)
                cn.center[d] = nd.center[d] + ((s & (1 << d)) ? lenhalf : -lenhalf);
            cn.father = no;
            cn.flags = 0;
            nd.suns[j] = ch;
            child_of[j] = ch;
            clo[j] = lo + c[s];
            chi[j] = lo + c[s + 1];
            j++;
        }
        /* sibling threading: forcetree.cpp:968-983,1055-1061 */
        for(int k = 0; k < nch; k++)
            N[child_of[k]].sibling = (k + 1 < nch) ? child_of[k + 1] : nd.sibling;
        for(int k = 0; k < nch; k++) {
            if(chi[k] - clo[k] > 4096 && depth < 8) {
#pragma omp task default(shared) firstprivate(k)
                fill(child_of[k], clo[k], chi[k], depth + 1);
            } else
                fill(child_of[k], clo[k], chi[k], depth + 1);
        }
#pragma omp taskwait
        /* moments: forcetree.cpp:1080-1101 */
        for(int k = 0; k < nch; k++) {
            const NODE &cn = N[child_of[k]];
            nd.mass += cn.mass;
            nd.cofm[0] += cn.mass * cn.cofm[0];
            nd.cofm[1] += cn.mass * cn.cofm[1];
            nd.cofm[2] += cn.mass * cn.cofm[2];
            if(cn.hmax > nd.hmax)
                nd.hmax = cn.hmax;
        }
        if(nd.mass > 0) {
            nd.cofm[0] /= nd.mass;
            nd.cofm[1] /= nd.mass;
            nd.cofm[2] /= nd.mass;
        }
    }
};

} // namespace

int force_tree_rebuild_mask(ForceTree *tree, const part_manager_type *PartManager, int mask,
                            const ActiveParticles *act, int alloc_father)
{
    memset(tree, 0, sizeof(*tree));
    const particle_data *P = PartManager->Base;
    const int64_t np = PartManager->NumPart;
    std::vector<int> perm;
    perm.reserve(np);
    const int64_t nloop = (act && act->ActiveParticle) ? act->NumActiveParticle : np;
    for(int64_t k = 0; k < nloop; k++) {
        const int i = (act && act->ActiveParticle) ? act->ActiveParticle[k] : (int) k;
        if(P[i].IsGarbage || P[i].Swallowed)
            continue;
        if(!((1 << P[i].Type) & mask))
            continue;
        perm.push_back(i);
    }
    const int64_t n = (int64_t) perm.size();
    /* node budget: <= 1 + 8/9-ish per particle in the worst case of a top-down tree with
     * removed empties: every internal node has >= 9 particles... but clustered inputs make
     * long single-child chains, so be generous (TreeAllocFactor analogue, forcetree.cpp:33-35) */
    int64_t maxnodes = (int64_t) (1.5 * n) + 4096;
    for(int attempt = 0; attempt < 4; attempt++) {
        NODE *base = (NODE *) malloc(sizeof(NODE) * maxnodes);
        if(!base)
            return 3;
        std::vector<int> tmp(n > 0 ? n : 1);
        std::vector<int> work(perm);
        Build b;
        b.P = P;
        b.firstnode = np;
        b.lastnode = np + maxnodes;
        b.N = base - np;
        b.nnext.store(np + 1);
        b.father = nullptr;
        int *father = nullptr;
        if(alloc_father) {
            father = (int *) malloc(sizeof(int) * (np > 0 ? np : 1));
            for(int64_t i = 0; i < np; i++)
                father[i] = -1;
            b.father = father;
        }
        b.perm = work.data();
        b.tmp = tmp.data();
        NODE &root = b.N[np];
        memset(&root, 0, sizeof(root));
        root.len = PartManager->BoxSize * 1.001; /* forcetree.cpp:661 */
        for(int d = 0; d < 3; d++)
            root.center[d] = PartManager->BoxSize / 2.;
        root.father = -1;
        root.sibling = -1;
        root.flags = 2u | 4u; /* TopLevel, DependsOnLocalMass */
#pragma omp parallel
#pragma omp single
        b.fill((int) np, 0, n, 0);
        if(b.err.load() == 1) {
            free(base);
            free(father);
            maxnodes *= 2;
            continue;
        }
        if(b.err.load() == 2) {
            free(base);
            free(father);
            return 2; /* more than NMAXCHILD coincident particles */
        }
        tree->tree_allocated_flag = 1;
        tree->hmax_computed_flag = 1;
        tree->moments_computed_flag = 1;
        tree->full_particle_tree_flag = 0;
        tree->firstnode = np;
        tree->lastnode = np + maxnodes;
        tree->numnodes = b.nnext.load() - np;
        tree->mask = mask;
        tree->NumParticles = n;
        tree->Nodes_base = base;
        tree->Nodes = base - np;
        tree->Father = father;
        tree->nfather = alloc_father ? np : 0;
        tree->BoxSize = PartManager->BoxSize;
        return 0;
    }
    return 1;
}

void force_tree_free(ForceTree *tree)
{
    if(!tree->tree_allocated_flag)
        return;
    free(tree->Nodes_base);
    free(tree->Father);
    memset(tree, 0, sizeof(*tree));
}

shq_tree_view force_tree_view(const ForceTree *tree)
{
    shq_tree_view v;
    v.nodes_base = tree->Nodes_base;
    v.firstnode = tree->firstnode;
    v.lastnode = tree->lastnode;
    v.numnodes = tree->numnodes;
    v.rootnode = (int32_t) tree->firstnode;
    v.full_particle_tree_flag = tree->full_particle_tree_flag;
    v.BoxSize = tree->BoxSize;
    v.father = tree->Father;
    return v;
}
