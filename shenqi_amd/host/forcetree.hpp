/* forcetree.hpp — host-side oct-tree in the reference's NODE format.
 *
 * The reference keeps tree ownership on the host (libgadget/forcetree.cpp); a shenqi build
 * hands its own ForceTree to the C-ABI.  This builder exists for the standalone driver, tests
 * and bench: it produces the SAME tree the reference's insertion algorithm produces (a node is
 * internal iff it holds more than NMAXCHILD particles; children ordered by sub-octant
 * x + 2y + 4z with empty ones removed, forcetree.cpp:277-283,1031-1048; root length
 * 1.001*BoxSize centred on BoxSize/2, forcetree.cpp:661-663; moments forcetree.cpp:947-1103)
 * but is built top-down by stable partitioning, in parallel.  Single top-leaf domain only
 * (the `trivial_domain` of tests/test_forcetree.cpp:294-314). */
#ifndef SHQH_FORCETREE_HPP
#define SHQH_FORCETREE_HPP
#include "partmanager.hpp"

#define NMAXCHILD SHQ_NMAXCHILD
#define GASMASK (1)
#define DMMASK (2)
#define NUMASK (1 << 2)
#define STARMASK (1 << 4)
#define BHMASK (1 << 5)
#define ALLMASK ((1 << 6) - 1)

typedef shq_node NODE; /* binary mirror of struct NODE, forcetree.h:38-66 */

struct ForceTree {
    int tree_allocated_flag;
    int hmax_computed_flag;
    int moments_computed_flag;
    int full_particle_tree_flag;
    int64_t firstnode;
    int64_t lastnode;
    int64_t numnodes;
    int mask;
    int64_t NumParticles;
    NODE *Nodes;      /* shifted: Nodes[firstnode] is the root */
    NODE *Nodes_base;
    int *Father;
    int64_t nfather;
    double BoxSize;
};

/* Build a tree over the particles whose type bit is in `mask` and that are neither garbage
 * nor swallowed (forcetree.cpp:805-806), compute moments (and hmax from Hsml for gas/BH).
 * act == NULL or act->ActiveParticle == NULL => all particles.  Returns 0 or an error code. */
int force_tree_rebuild_mask(ForceTree *tree, const part_manager_type *PartManager, int mask,
                            const ActiveParticles *act, int alloc_father);
static inline int force_tree_full(ForceTree *tree, const part_manager_type *PartManager)
{
    int rc = force_tree_rebuild_mask(tree, PartManager, ALLMASK, nullptr, 1);
    if(rc == 0)
        tree->full_particle_tree_flag = 1;
    return rc;
}
void force_tree_free(ForceTree *tree);
static inline int force_tree_allocated(const ForceTree *tt) { return tt->tree_allocated_flag; }
shq_tree_view force_tree_view(const ForceTree *tree);
#endif
