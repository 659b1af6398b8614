/* shenqi_hip.h — C-ABI of the MI355X (gfx950) TreePM + SPH force engine.
 *
 * This is the drop-in boundary for shenqi's device seam: the four free functions the
 * reference calls when `UseGPU` is set,
 *     grav_short_tree_cuda   (libgadget/gravshort2.hpp:440-443, caller gravshort-tree2.cpp:151-154)
 *     density_cuda           (libgadget/densitytree2.hpp:437-439, caller density2.cpp:116-119)
 *     hydro_force_cuda       (libgadget/hydratree2.hpp:391-393,  caller hydra2.cpp:89-91)
 *     petapm_fft_r2c/_c2r + the host PM deposit/transfer/readout loops
 *                            (libgadget/petapm.h:136,143-144; petapm.cpp:392-465)
 * All entry points are extern "C", take plain pointers / sizes / POD structs and return an int
 * status (0 = ok).  On failure shq_last_error() returns a message; the shenqi-side shim turns
 * a non-zero status into endrun() (reference error convention: libgadget/utils/endrun.h:6,
 * device-launch check treewalk2.cuh:326-328).
 *
 * Ownership: the caller owns every host array; the library owns all device memory (its own
 * pools, never the caller's allocator: reference arena is LIFO, utils/memory.c).
 * Threading: one context per rank/GPU, calls come from the rank's main thread
 * (MPI_THREAD_FUNNELED, gadget/main.cpp:35).  Calls are synchronous unless stated.
 *
 * Two levels are offered:
 *   one-shot  (shq_grav_short_tree, shq_pm_force, shq_density, shq_hydro_force):
 *             host pointers in, host pointers out — what the reference call sites need;
 *   resident  (shq_particles_upload / shq_tree_upload / *_run / *_download):
 *             data stays in HBM between calls; used by time-step loops that keep particles
 *             on the device and by bench.py (timed region starts with inputs in HBM).
 */
#ifndef SHENQI_HIP_H
#define SHENQI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SHQ_OK 0
#define SHQ_ERR_INVALID 1   /* bad argument / inconsistent view */
#define SHQ_ERR_DEVICE 2    /* HIP runtime / launch failure */
#define SHQ_ERR_NOMEM 3     /* device allocation failed */
#define SHQ_ERR_STATE 4     /* call order violated (e.g. run before upload) */
#define SHQ_ERR_NOCONV 5    /* Hsml iteration did not converge within MAXITER */

#define SHQ_NGRAVTAB 512    /* NGRAVTAB, libgadget/gravity.h:35 */
#define SHQ_NMAXCHILD 8     /* NMAXCHILD, libgadget/forcetree.h:13 */
#define SHQ_NODELISTLENGTH 4 /* NODELISTLENGTH, libgadget/localtreewalk2.h */
#define SHQ_TIMEBINS 46     /* TIMEBINS, libgadget/timebinmgr.h */

typedef struct shq_context shq_context;

/* ---- context ------------------------------------------------------------------------- */

/* Create a context on HIP device `device`.  `stream` is a hipStream_t to launch on (e.g.
 * torch's current stream) or NULL for a library-owned stream. */
int shq_init(int device, void *stream, shq_context **out);
void shq_shutdown(shq_context *ctx);
/* Last error message of the calling thread ("" if none). */
const char *shq_last_error(void);
/* Block until all work queued on the context's stream has finished. */
int shq_synchronize(shq_context *ctx);
/* The hipStream_t the context launches on (for event timing by the caller). */
void *shq_stream(shq_context *ctx);
/* HIP-event timer on the context's stream: begin/end bracket; elapsed returns ms of slot. */
int shq_timer_begin(shq_context *ctx, int slot);
int shq_timer_end(shq_context *ctx, int slot);
int shq_timer_elapsed_ms(shq_context *ctx, int slot, double *ms);
/* elapsed time between two recorded timer events (which: 0 begin, 1 end), e.g. the caller's slot 1 begin -> the library's walk (slot 19)
 * begin: where the phases of a resident step lie on one time line although they run on different streams.  Library slots: 8 PM begin,
 * 9 / 10 FFT pipeline begin / end, 13 PM end (all as begin events), 16 tree build, 19 tree walk. */
int shq_timer_between_ms(shq_context *ctx, int slot_a, int which_a, int slot_b, int which_b, double *ms);
/* Library version string. */
const char *shq_version(void);

/* ---- data views (reference layouts, no copies on the host side) ---------------------- */

/* Strided AoS view of `struct particle_data` (libgadget/partmanager.h:9-71), in the spirit of
 * PetaPMParticleStruct (libgadget/petapm.h:62-72): base pointer + element size + byte offsets.
 * An offset of SHQ_NOFIELD means "field not supplied". */
#define SHQ_NOFIELD ((size_t)-1)
typedef struct shq_part_view {
    void *base;             /* PartManager->Base */
    size_t elsize;          /* sizeof(struct particle_data) = 160 */
    int64_t numpart;        /* PartManager->NumPart */
    size_t off_pos;         /* double Pos[3] */
    size_t off_mass;        /* float  Mass */
    size_t off_type;        /* unsigned char Type */
    size_t off_flags;       /* unsigned int bitfield word: bit0 IsGarbage, bit1 Swallowed */
    size_t off_pi;          /* int PI (slot index) */
    size_t off_vel;         /* double Vel[3] */
    size_t off_treeacc;     /* double FullTreeGravAccel[3] */
    size_t off_gravpm;      /* double GravPM[3] */
    size_t off_potential;   /* double Potential */
    size_t off_hsml;        /* double Hsml */
    size_t off_dthsml;      /* double DtHsml */
    size_t off_timebin_hydro;   /* unsigned char TimeBinHydro */
    size_t off_timebin_gravity; /* unsigned char TimeBinGravity */
} shq_part_view;

/* Strided view of `struct sph_particle_data` (libgadget/slotsmanager.h:97-131), indexed by PI. */
typedef struct shq_sph_view {
    void *base;
    size_t elsize;          /* 176 */
    int64_t numslots;
    size_t off_density, off_egywtdensity, off_entropy, off_dtentropy, off_maxsignalvel;
    size_t off_hydroaccel;  /* double[3] */
    size_t off_dhsmlegydensityfactor, off_divvel, off_curlvel, off_delaytime;
} shq_sph_view;

/* Binary mirror of `struct NODE` (libgadget/forcetree.h:38-66), 120 bytes, so the host tree
 * (forcetree.cpp keeps ownership) is passed as-is. */
typedef struct shq_node {
    int32_t sibling;
    int32_t father;
    double len;
    double center[3];
    double cofm[3];     /* mom.cofm */
    double mass;        /* mom.mass */
    double hmax;        /* mom.hmax */
    int32_t suns[SHQ_NMAXCHILD];
    int32_t noccupied;
    uint32_t flags;     /* bit0 InternalTopLevel, bit1 TopLevel, bit2 DependsOnLocalMass, bits3-4 ChildType */
} shq_node;
#define SHQ_NODE_INTERNALTOPLEVEL(f) ((f) & 1u)
#define SHQ_NODE_TOPLEVEL(f) (((f) >> 1) & 1u)
#define SHQ_NODE_CHILDTYPE(f) (((f) >> 3) & 3u)
#define SHQ_PARTICLE_NODE_TYPE 0
#define SHQ_NODE_NODE_TYPE 1
#define SHQ_PSEUDO_NODE_TYPE 2

/* View of `ForceTree` (libgadget/forcetree.h:75-112). Index space as in the reference:
 * [0,firstnode) particles, [firstnode,firstnode+numnodes) nodes, >= lastnode pseudo. */
typedef struct shq_tree_view {
    const shq_node *nodes_base;  /* ForceTree.Nodes_base (Nodes = nodes_base - firstnode) */
    int64_t firstnode;
    int64_t lastnode;
    int64_t numnodes;
    int32_t rootnode;            /* node the primary walk starts from (== firstnode) */
    int32_t full_particle_tree_flag;
    double BoxSize;
    const int32_t *father;       /* ForceTree.Father (parent node of every particle) or NULL */
} shq_tree_view;

/* ---- short-range gravity -------------------------------------------------------------- */

/* POD mirror of GravTreeParams (libgadget/gravshort2.hpp:21-55) incl. GravShortTable
 * (libgadget/gravity.h:32-61) held by value exactly as the reference does. */
typedef struct shq_grav_params {
    double BoxSize;
    double cellsize;        /* BoxSize / Nmesh */
    double Rcut;            /* TreeParams.Rcut * Asmth * cellsize */
    double G;
    double cbrtrho0;
    double ForceSoftening;  /* FORCE_SOFTENING() = 2.8 * GravitySoftening */
    double ErrTolForceAcc;
    double BHOpeningAngle2; /* already squared; = MaxBHOpeningAngle^2 when TreeUseBH == 0 */
    int32_t TreeUseBH;
    int32_t pad_;
    float shortrange_table[SHQ_NGRAVTAB];
    float shortrange_table_potential[SHQ_NGRAVTAB];
    double dx;              /* table spacing in mesh cells */
} shq_grav_params;

typedef struct shq_walk_stats {
    int64_t ntargets;
    int64_t ninteractions;      /* sum over targets of the reference's visit() return value */
    int64_t min_interactions;
    int64_t max_interactions;
    int64_t nnodes_visited;     /* node tests executed by wavefronts (union walk) */
    int64_t nwave_interactions; /* interaction evaluations issued by wavefronts (x64 lanes each) */
    int64_t nwave_node_interactions; /* ... of which monopole (node) evaluations */
    int64_t nnode_interactions; /* per-target interactions that were node monopoles */
    double kernel_ms;           /* HIP-event time of the walk kernel(s) */
} shq_walk_stats;

/* Walk flavour.  Both reproduce the reference's per-target opening decisions and interaction set (interaction counts equal
 * the reference's as integers; parity bar runtests.cpp:441-443, SURVEY.md §8(c) rung L1).
 * EXACT: one target per lane, the wave walks the union of its 64 targets' walks depth-first; every target's contributions are
 *   summed in the reference's order.  Also serves the secondary (imported-query) walks.
 * GROUP: one SOURCE per lane; the wave takes its 64 targets as 8 groups of 8, every pending node carries the mask of the
 *   members whose own walk reaches it, node tests run 64 nodes at a time and each accepted source is applied to exactly the
 *   members that accept it.  Same interaction set per target, summed in a different order (forces agree to ~1e-15
 *   relative).  Measured at 256^3 (S-cluster): 73 ms against 39.5 ms for all particles, but 3.9 against 7.2 ms for every 64th and
 *   3.2 against 11.3 ms for every 512th particle of the same tree: the walk of choice for SPARSE active lists over a full tree.
 * AUTO: GROUP when an active list holds fewer than a tenth of the tree's particles (where the two lines above cross), EXACT
 *   otherwise — what the hierarchical integrator's levels should pass (timestep.cpp:437-446 walks ever shorter lists). */
#define SHQ_WALK_EXACT 0
#define SHQ_WALK_GROUP 1
#define SHQ_WALK_AUTO 2
/* flag, or-ed into walk_mode: with active == NULL, take the targets of the tree's particles sorted along a
 * Peano-Hilbert curve of their CURRENT positions (keys + radix sort on the device, ~2 ms for 1.7e7) instead of
 * particle-index order — same results per particle; keeps target groups compact when the particle
 * order has gone stale (resident stepping without the reference's periodic Peano-Hilbert re-sort) */
#define SHQ_WALK_TREE_ORDER 0x100
/* flag, or-ed into walk_mode: leave the raw sums of the primary walk on the device and skip
 * GravTreeOutput::postprocess; shq_grav_reduce_export_results then adds the results of the exported queries
 * (ev_reduce_export_result, treewalk2.h:793-812) and shq_grav_postprocess finishes (ev_postprocess, :375-382) —
 * the order the reference's distributed walk keeps. */
#define SHQ_WALK_DEFER_POSTPROCESS 0x200

/* Order of n device-resident positions (rows of 4 doubles: x, y, z, anything) along the Peano-Hilbert curve the walk groups its
 * targets by (the order SHQ_WALK_TREE_ORDER walks in): d_order[k] = row index of the k-th particle, equal keys in index order.
 * What a driver calls after a domain exchange in place of the reference's particle sort (domain.cpp:268, slots_gc_sorted), without
 * bringing the positions to the host.  Both pointers are device pointers; the work is queued on the context's stream. */
int shq_hilbert_order(shq_context *ctx, const double *d_posm, int64_t n, double BoxSize, int64_t *d_order);

/* One-shot replacement of grav_short_tree_cuda(): walks the local tree for the `nactive`
 * targets in `active` (NULL => all particles, as ActiveParticles with a NULL list), writes
 * Accel[target][0..2] (already multiplied by G: GravTreeOutput::postprocess,
 * gravshort2.hpp:88-107) and, when update_potential, also FullTreeGravAccel and Potential
 * inside the particle view. `accel` is indexed by particle index, numpart rows. */
int shq_grav_short_tree(shq_context *ctx, const shq_tree_view *tree, const shq_part_view *parts,
                        const int32_t *active, int64_t nactive, const shq_grav_params *params,
                        double (*accel)[3], int update_potential, int walk_mode,
                        shq_walk_stats *stats);

/* Resident API. */
int shq_particles_upload(shq_context *ctx, const shq_part_view *parts);
/* One-shot operators upload the views they are handed — particles, SPH state, tree, and the ID array of the sub-grid walks: at 2 x 10^6
 * particles 50-170 ms of packing and PCIe around 4-7 ms of kernels, and run.cpp:621-681 calls a handful of them per step on the same
 * particles.  With a bit of `mask` set the caller vouches that the context's copy of that input IS the view it passes next: same base
 * pointer and count, contents unchanged on the host since the copy was made or changed only by the library's own operators (which
 * write what they change to the views and to the context alike; shq_metal_return, which changes masses and densities in the caller's
 * records only, takes the particle and SPH bits back itself).  Those uploads are then skipped.  A drift, a kick, a device-side particle
 * hand-over or mask = 0 ends it. */
#define SHQ_CURRENT_PARTICLES 1
#define SHQ_CURRENT_SPH 2
#define SHQ_CURRENT_TREE 4
#define SHQ_CURRENT_IDS 8
int shq_set_inputs_current(shq_context *ctx, int mask);

int shq_tree_upload(shq_context *ctx, const shq_tree_view *tree);

/* Device tree build.  Replaces, for a single-domain tree, force_tree_rebuild / force_tree_create_nodes
 * (libgadget/forcetree.cpp:727-859), force_tree_calc_moments / force_update_node_parallel (:1016-1142)
 * and the host repack + H2D of shq_tree_upload: builds the oct-tree of the uploaded particles whose type
 * is in `mask` (bit t = particle type t; garbage and swallowed particles are skipped, as in
 * forcetree.cpp:692-705), optionally restricted to the `active` list (host pointer, ActiveParticles
 * order), computes mass / centre of mass / hmax and installs it as the context's current tree.
 * The tree is the one the reference's insertion build produces for the same particles (same cells,
 * same leaves in the same particle order, same moments to the bit).  Returns SHQ_ERR_INVALID when more
 * than 8 particles lie within Box/2^21 of each other (deeper than the device build supports): the caller
 * then keeps its host build; nothing is truncated. */
typedef struct shq_tree_build_stats {
    int64_t nparticles;   /* particles in the tree */
    int64_t numnodes;
    int32_t maxdepth;
    float build_ms;       /* device time, HIP events */
} shq_tree_build_stats;
int shq_tree_build(shq_context *ctx, double BoxSize, int mask, const int32_t *active, int64_t nactive,
                   shq_tree_build_stats *stats);
/* The device-built tree in the reference's format (struct NODE, forcetree.h:38-66): `nodes[k]` is node
 * number firstnode + k (pre-order, root first), links (sibling, father, suns of internal nodes) are node
 * numbers, leaf suns are particle indices; `father[i]` = node number of the leaf holding particle i or
 * -1 (ForceTree.Father).  Either output may be NULL; *numnodes always returns the node count.  After shq_tree_build_domain the
 * top-level nodes carry TopLevel / InternalTopLevel and pseudo nodes suns[0] = firstnode + capacity + leaf (lastnode is the end of
 * the caller's node array). */
int shq_tree_download(shq_context *ctx, int64_t firstnode, shq_node *nodes, int64_t capacity, int32_t *father,
                      int64_t *numnodes);

/* Device tree build under a domain decomposition: the top-tree and pseudo-particle part of force_tree_create_nodes
 * (force_tree_create_topnodes / force_create_node_for_topnode, libgadget/forcetree.cpp:651-690, 868-930), the moments of the
 * local sub-trees, and force_exchange_pseudodata / force_treeupdate_pseudos (:1136-1281) around the caller's all-gather.
 *
 * The top tree arrives as geometry: for every TopNode its eight daughters in octant order `count = i + 2 j + 4 k`
 * (the slot force_create_node_for_topnode puts TopNodes[Daughter + sub] in, sub = 7 & peano_hilbert_key(2x + i, 2y + j, 2z + k,
 * bits) — the reference-side shim fills the table with its own key function), -1 eight times for a leaf, and
 * the leaf's TopLeaves index.  topleaves[].Task says whose each leaf is.
 *
 * shq_tree_build_domain: as shq_tree_build, but every TopNode is a tree node whether or not it holds particles (empty top-level
 * nodes are never removed, forcetree.cpp:1040-1045), a leaf of ThisTask roots an ordinary sub-tree of this rank's particles, a
 * leaf of another task is a pseudo node (ChildType PSEUDO, suns[0] = lastnode + leaf).  SHQ_ERR_INVALID if a particle of the
 * list lies in a leaf of another task (the reference ends the run: "Bad topleaf").  Outputs: topleaves[].treenode = firstnode +
 * pre-order number of the leaf's node (what shq_tree_download numbers it), local_moments[leaf] = {s, mass, hmax} of the leaves
 * of ThisTask (zero elsewhere): the caller's contribution to the all-gather of force_exchange_pseudodata.
 * shq_tree_set_topleaf_moments: after the all-gather; the moments of the other tasks' leaves go into their pseudo nodes, the
 * internal top-level nodes are re-summed over their eight daughters (force_treeupdate_pseudos) and the top tree is installed for
 * shq_grav_toptree_exports / shq_ngb_toptree_exports (no shq_toptree_upload needed).  A one-task domain needs no second call. */
typedef struct shq_topleaf {   /* struct topleaf_data, libgadget/domain.h:20-24 */
    int32_t Task;
    int32_t topnode;
    int32_t treenode;
} shq_topleaf;
typedef struct shq_topnode_geo {
    int32_t daughter[8];    /* TopNodes index per octant, all -1 for a leaf */
    int32_t leaf;           /* TopLeaves index of a leaf, -1 otherwise */
    int32_t pad_;
} shq_topnode_geo;
typedef struct shq_topleaf_moments {   /* struct topleaf_momentsdata, forcetree.cpp:1129-1134 */
    double s[3];
    double mass;
    double hmax;
} shq_topleaf_moments;
int shq_tree_build_domain(shq_context *ctx, double BoxSize, int mask, const int32_t *active, int64_t nactive,
                          const shq_topnode_geo *topnodes, int ntopnodes, shq_topleaf *topleaves, int ntopleaves, int ThisTask,
                          int64_t firstnode, shq_topleaf_moments *local_moments, shq_tree_build_stats *stats);
int shq_tree_set_topleaf_moments(shq_context *ctx, const shq_topleaf_moments *moments, int ntopleaves);
/* The particle loop of domain_maintain() (libgadget/domain.cpp:296-330, 347-368) after shq_drift, on the domain of the last
 * shq_tree_build_domain (its top tree and the cells of its top-level nodes): a live particle that is still inside the cell of its
 * top leaf keeps it (inside_topleaf, bounds included), any other gets TopLeaf = domain_get_topleaf(PEANO(Pos)) (domain.h:68-76,
 * utils/peano.h:15-21: the descent through the daughter table on PEANO()'s integer coordinates finds the same leaf); d_target
 * (may be NULL) = layoutfunc: the leaf's task, -1 for garbage and, when dmtree == 0, for dark matter whose gravity bin is not
 * active at Ti_Current (it stays and keeps its leaf).  d_topleaf / d_target are DEVICE arrays of one int per resident particle —
 * d_target is what shq_exchange_plan takes.  *nchanged = particles that left their leaf. */
int shq_domain_maintain_topleaf(shq_context *ctx, int dmtree, int64_t Ti_Current, int32_t *d_topleaf, int32_t *d_target, int64_t *nchanged);

/* Resident drift and kick: with the particles, their velocities and the force
 * arrays in HBM a step is  shq_drift -> shq_tree_build -> shq_pm_run / shq_grav_short_run -> kicks,
 * without a PCIe crossing.  Same operations in the same order as the reference, so the state stays
 * bit-identical to a host integration.  Black-hole repositioning (drift.cpp:32-53) and the black-hole part of
 * do_hydro_kick need the BH slot fields (shq_bh_dynamics_upload below); the particle loops of the integer time line are
 * shq_find_timesteps and its relatives below, DriftKickTimes and the sync points stay with the host.
 *
 * shq_dynamics_upload: Vel, Hsml, DtHsml, TimeBinGravity of the particles uploaded before.
 * shq_drift: drift_all_particles / real_drift_particle (libgadget/drift.cpp:16-99):
 *     Pos += Vel * ddrift + random_shift, wrapped into (0, BoxSize]; gas Hsml += DtHsml * ddrift, capped at
 *     BoxSize / 2; garbage / swallowed particles only follow the shift.  Invalidates the tree and the PM result.
 *     SHQ_ERR_INVALID for a non-finite position or a gas Hsml <= 0 (the reference ends the run).
 * shq_kick_short: apply_half_kick, gravity part (timestep.cpp:838-872, 962-968):
 *     Vel += A * gravkick[TimeBinGravity] for the active list (NULL => all), A = FullTreeGravAccel, or the
 *     walk's Accel output (AccelStore, timestep.cpp:273) when from_accel_store; inactive bins carry 0.
 * shq_kick_pm: apply_PM_half_kick (timestep.cpp:937-959): Vel += GravPM * Fgravkick for every particle.
 * shq_dynamics_download: Pos, Vel, Hsml back into the caller's particle array. */
int shq_dynamics_upload(shq_context *ctx, const shq_part_view *parts);

/* Active-particle lists on the device (build_active_particles / build_active_sublist,
 * libgadget/timestep.cpp:1286-1390).  The lists stay in HBM; SHQ_ACTIVE_RESIDENT / SHQ_SUBLIST_RESIDENT passed as
 * the `active` argument of shq_tree_build, shq_grav_short_run and shq_kick_short select them (nactive is ignored),
 * so a sub-step of the hierarchical integrator never moves an index list over PCIe.
 *
 * shq_timebins_upload: TimeBinGravity / TimeBinHydro of the resident particles (host arrays of numpart bytes;
 *     NULL keeps what is there, zero if nothing is).  shq_dynamics_upload and shq_sph_upload also set them.
 * shq_build_active_particles: ActivePredicate (timestep.cpp:1265-1282) over all particles, in index order (std::copy_if
 *     is stable): not garbage / swallowed, and gravity bin active or (gas / BH and hydro bin active) at Ti_Current
 *     (is_timebin_active, timestep.cpp:132-139).  is_pm_step: the reference's PM-step branch, every particle is active
 *     and ActiveParticle stays NULL (NumActiveHydro then counts the type-0/5 records, where the reference reports its
 *     slot-array sizes).  info may be NULL.
 * shq_build_active_sublist: SubActivePredicate (timestep.cpp:1354-1371) over the resident list: gravity bin <= maxtimebin
 *     and active.
 * shq_active_download: either list to the host (count always returned; list may be NULL). */
typedef struct shq_active_info {
    int64_t NumActiveParticle, NumActiveGravity, NumActiveHydro;
    int64_t TimeBinCountType[6 * (SHQ_TIMEBINS + 1)];   /* [(TIMEBINS + 1) * type + bin] */
} shq_active_info;
#define SHQ_ACTIVE_RESIDENT ((const int32_t *) (intptr_t) -1)
#define SHQ_SUBLIST_RESIDENT ((const int32_t *) (intptr_t) -2)
#define SHQ_SPH_QUEUE_RESIDENT ((const int32_t *) (intptr_t) -3)   /* the current work queue of an open SPH walk */
int shq_timebins_upload(shq_context *ctx, const uint8_t *bin_gravity, const uint8_t *bin_hydro);
int shq_build_active_particles(shq_context *ctx, int64_t Ti_Current, int is_pm_step, shq_active_info *info);
int shq_build_active_sublist(shq_context *ctx, int maxtimebin, int64_t Ti_Current, int64_t *nsub);
int shq_active_download(shq_context *ctx, int sublist, int32_t *list, int64_t capacity, int64_t *count);
int shq_drift(shq_context *ctx, double ddrift, double BoxSize, const double random_shift[3]);
int shq_kick_short(shq_context *ctx, const double gravkick[SHQ_TIMEBINS + 1], const int32_t *active, int64_t nactive,
                   int from_accel_store);
int shq_kick_pm(shq_context *ctx, double Fgravkick);
/* do_hydro_kick for gas (timestep.cpp:970-1003) as apply_half_kick / apply_hydro_half_kick call it (:875-886, :914-934):
 * Vel += HydroAccel * hydrokick[TimeBinHydro], the MaxGasVel clamp (|Vel| / atime <= MaxGasVel; *nlimited counts the
 * clamped particles, which the reference logs), Entropy += DtEntropy * dt_entr[TimeBinHydro] — on the SPH state the last
 * shq_density / shq_hydro_force (or phase calls) left resident; from_hydro_output takes HydroAccel / DtEntropy from that
 * hydro run instead of the uploaded SphP values.  Inactive bins carry 0 in both tables.  The black-hole part (DFAccel,
 * DragAccel) needs the BH slot arrays and stays with the host.  shq_entropy_download: Entropy by particle index. */
int shq_kick_hydro(shq_context *ctx, const double hydrokick[SHQ_TIMEBINS + 1], const double dt_entr[SHQ_TIMEBINS + 1], double atime,
                   double MaxGasVel, const int32_t *active, int64_t nactive, int from_hydro_output, int64_t *nlimited);
int shq_entropy_download(shq_context *ctx, double *entropy_by_particle);
int shq_dynamics_download(shq_context *ctx, const shq_part_view *parts);

/* The integer time line on the device (libgadget/timestep.cpp:157-194, 307-446, 584-822, 1012-1110):
 * new time bins for the resident particles from the accelerations, smoothing lengths and signal velocities that are in HBM
 * already.  The sync-point table, the cosmology and DriftKickTimes stay with the host (integration/reference_side/timestep.cpp mirrors
 * find_timesteps / find_hydro_timesteps / hierarchical_gravity_and_timesteps over these calls); the per-particle loops and
 * their reductions run here.  IEEE sqrt and divide, no fma contraction: the bins equal the host loop's as integers.
 *
 * shq_timeline: what TimeBinMgr::dti_from_dloga and get_dloga_for_bin read at Ti_Current (timebinmgr.h:120-176, 228-243):
 *     loga_now = SyncPoints[lastsnap].loga + (Ti_Current & (TIMEBASE - 1)) * Dloga_interval; the segment the step starts in
 *     (seg_snap[0], lower sync point index after the reference's clamps) with its two ends seg_loga[0..1], and when a further
 *     sync point exists (nseg = 2) the one after, seg_snap[1] = seg_snap[0] + 1, ending at seg_loga[2]. */
typedef struct shq_timeline {
    int64_t Ti_Current;
    double loga_now;
    double Dloga_interval;      /* Dloga_interval_ti(Ti_Current), 0 past the last sync point */
    int32_t nseg;
    int32_t pad_;
    int64_t seg_snap[2];
    double seg_loga[3];
} shq_timeline;

/* TimestepParams (timestep.cpp:35-50) and the arguments of the loops.  fac3 = pow(atime, 3 (1 - GAMMA) / 2) (timestep.cpp:1049),
 * ForceSoftening = FORCE_SOFTENING(), dti_max = times->PM_length after the PM-step update of the caller. */
typedef struct shq_timestep_params {
    double ErrTolIntAccuracy, CourantFac, MinSizeTimestep;
    double ForceSoftening;
    double atime, hubble, fac3;
    int64_t dti_max;
    int32_t ForceEqualTimesteps;
    int32_t isFirstTimeStep;
    int32_t mintimebin, mingravtimebin;   /* times->mintimebin / mingravtimebin on entry (find_hydro_timesteps reads them) */
    shq_timeline tl;
} shq_timestep_params;

typedef struct shq_timestep_result {
    int32_t badstepsizecount;
    int32_t mTimeBin, maxTimeBin;         /* min / max new bin over the list (TIMEBINS / 0 when the list is empty) */
    int32_t mintimebin;                   /* find_hydro_timesteps: times->mintimebin after its fix-ups (timestep.cpp:677-696) */
    int64_t ntiaccel, nticourant, ntihsml, ntiaccrete, ntineighbour;
    int64_t nbh, dynratio;                /* find_hydro_timesteps: BHs on the list, sum of TimeBinDynFric - TimeBinHydro */
    int32_t maxdyndiff;
    int32_t nbadbin;                      /* particles the reference would print_bad_timebin for */
    int64_t dti_min;                      /* ForceEqualTimesteps: find_global_timestep (before the caller's MPI minimum) */
    int64_t timebincounts[SHQ_TIMEBINS + 1]; /* shq_hier_gravity_bins */
} shq_timestep_result;

/* find_timesteps particle loop (timestep.cpp:733-792): new TimeBinHydro = TimeBinGravity for the list (NULL: all; the
 * resident lists are accepted) from the gravity criterion on FullTreeGravAccel + GravPM and, for gas / BH, the hydro criteria
 * (MaxSignalVel of the last hydro run or upload, Hsml, DtHsml; the BH neighbour limiter needs shq_bh_dynamics_upload).
 * ForceEqualTimesteps: dti_min_global (the caller's reduction of shq_find_global_timestep over ranks) is every particle's step.
 * isFirstTimeStep: set_bh_first_timestep(mTimeBin) with mTimeBin_global >= 0, otherwise with this list's own minimum.
 * The caller applies the DriftKickTimes updates of timestep.cpp:707-722, 806-821. */
int shq_find_timesteps(shq_context *ctx, const shq_timestep_params *p, const int32_t *active, int64_t nactive, int64_t dti_min_global,
                       int mTimeBin_global, shq_timestep_result *res);
/* find_global_timestep (timestep.cpp:195-221): min over all live particles of the converted step; res->dti_min, res->nbadbin. */
int shq_find_global_timestep(shq_context *ctx, const shq_timestep_params *p, shq_timestep_result *res);
/* find_hydro_timesteps (timestep.cpp:583-703): TimeBinHydro of gas and BH on the list, capped by TimeBinGravity; BHs also get
 * TimeBinDynFric (get_timestep_dynfric_dloga, :1085-1110).  res->mintimebin carries the fix-ups of :677-696 for ONE rank; a
 * caller with several ranks reduces mTimeBin first and passes it back as mTimeBin_global to shq_set_bh_first_timestep. */
int shq_find_hydro_timesteps(shq_context *ctx, const shq_timestep_params *p, const int32_t *active, int64_t nactive, shq_timestep_result *res);
int shq_set_bh_first_timestep(shq_context *ctx, int mTimeBin);
/* hierarchical_gravity_and_timesteps, the three particle loops (timestep.cpp:356-380, 407-414, 449-464):
 * shq_hier_gravity_bins: TimeBinGravity = min(bin of the rounded-down gravity step, largest_active) and timebincounts;
 * shq_hier_push_down: TimeBinGravity = min(TimeBinGravity, push_down_bin) over the list;
 * shq_hier_refine: TimeBinGravity = ti - 1 where the step from the CURRENT level's acceleration is shorter than bin ti's
 *     (res->badstepsizecount counts them when ti == 1).
 * from_accel_store: the acceleration is the last walk's Accel output (AccelStore) instead of FullTreeGravAccel. */
int shq_hier_gravity_bins(shq_context *ctx, const shq_timestep_params *p, const int32_t *active, int64_t nactive, int from_accel_store,
                          int largest_active, shq_timestep_result *res);
int shq_hier_push_down(shq_context *ctx, const int32_t *active, int64_t nactive, int push_down_bin);
int shq_hier_refine(shq_context *ctx, const shq_timestep_params *p, const int32_t *active, int64_t nactive, int from_accel_store, int ti,
                    shq_timestep_result *res);
/* The sub-step levels of hierarchical_gravity_and_timesteps (timestep.cpp:417-476) in one call, resident: for ti = largest_active - 1
 * ... 1: sub-list (gravity bin <= ti) -> tree of the sub-list -> walk without potential -> shq_hier_refine -> hierarchical kick
 * Vel += Accel * gravkick_level[ti] (the caller's apply_hierarchical_grav_kick factor of level ti, :247-287, the lower level's kick
 * already subtracted).  Stops at the first empty sub-list (*mingravtimebin = ti + 1, :427-431; unchanged otherwise).
 * walk_mode: SHQ_WALK_EXACT / _GROUP / _AUTO for the level walks.  levels (may be NULL, capacity SHQ_TIMEBINS): what each level did. */
typedef struct shq_hier_level {
    int32_t timebin, walk_mode;       /* the level's bin; the walk it ran (what SHQ_WALK_AUTO resolved to) */
    int64_t nparticles, tree_nodes;   /* sub-list length = particles of the level's tree; its nodes */
    double tree_build_ms, walk_ms;    /* HIP-event times of the device tree build and of the walk kernel(s) */
} shq_hier_level;
int shq_hier_gravity_levels(shq_context *ctx, const shq_timestep_params *p, const shq_grav_params *gp, double BoxSize, int treemask,
                            int64_t Ti_Current, int largest_active, const double gravkick_level[SHQ_TIMEBINS + 1], int walk_mode,
                            int *mingravtimebin, int64_t *badstepsizecount, shq_hier_level *levels, int *nlevels);
/* get_long_range_timestep_dloga's particle loop (timestep.cpp:1153-1166): per type, sum of |Vel|^2, smallest positive mass and
 * count over the live particles.  The sum runs in a fixed order (blocks of 256 in index order, then the block sums in order),
 * so it is reproducible; the reference's OpenMP reduction has no fixed order. */
int shq_velocity_moments(shq_context *ctx, double v2sum[6], double min_mass[6], int64_t count[6]);
int shq_timebins_download(shq_context *ctx, uint8_t *bin_gravity, uint8_t *bin_hydro);
/* SphP[].MaxSignalVel by particle index (what the hydro criterion reads): upload for a state that did not come from a hydro
 * run of this context (NULL keeps it; gas particles only are read). */
int shq_maxsignalvel_upload(shq_context *ctx, const double *maxsignalvel_by_particle);
/* The gas state the SPH operators keep resident (Hsml, Vel, time bins, and per gas particle Entropy, DtEntropy, HydroAccel, DelayTime,
 * Density, EgyWtDensity, DhsmlEgyDensityFactor, DivVel, CurlVel, MaxSignalVel from its slot), without running one of them: what a
 * caller does before shq_fof (MaxDens / seed_index read Density and DelayTime) or shq_find_timesteps when no density / hydro call
 * preceded it in the step (e.g. right after a restart).  The particles must be the resident set. */
int shq_sph_state_upload(shq_context *ctx, const shq_part_view *parts, const shq_sph_view *sph);

/* Particle exchange between tasks on opaque records (ExchangePlan, libgadget/exchange.hpp:31-537).
 * The arrays are the reference's own — particle_data[MaxPart] and the per-type slot arrays — through pointers the device can
 * read and write (shenqi allocates them managed; a port keeps them in HBM); records are copied as bytes, the library reads only
 * the flag byte, Type, PI and the slots' ReverseLink.  The collectives between the phases are the caller's.
 *   shq_exchange_plan   build_exchange_list (:158-176) + the counting half of build_export_buffer (:178-204): particles whose
 *                       d_target (layoutfunc, one int per particle) names another task, in index order, garbage / swallowed ones
 *                       excluded; the first min(nexchange, maxlast) of them (maxlast <= 0: all; find_iter_space is the caller's
 *                       memory policy) are counted into toGo[task] = {base, slots[6]}.
 *   shq_exchange_pack   the pack loop of exchange_once (:369-392): per target task the base records in list order into
 *                       d_partbuf at toGoOffset[task].base, the slot records of every enabled type in the same order into
 *                       d_slotbuf[type] at toGoOffset[task].slots[type]; then slots_mark_garbage on the source (IsGarbage,
 *                       ReverseLink = MaxPart + 100).
 *   shq_exchange_unpack after the caller's alltoallv put the arrivals behind NumPart / behind each slot array's size: PI of
 *                       every arrival renumbered in arrival order per source task and type (:483-511).
 * The slot compaction the reference may interleave (slots_gc when memory is short) stays with the caller. */
typedef struct shq_exchange_layout {
    size_t part_elsize, off_flags, off_type, off_pi;
    size_t slot_elsize[6];      /* 0: slot type not enabled */
    size_t off_reverselink;     /* particle_data_ext::ReverseLink, first member of every slot struct */
} shq_exchange_layout;
typedef struct shq_exchange_entry {   /* ExchangePlanEntry, exchange.hpp:18-21 */
    int64_t base;
    int64_t slots[6];
} shq_exchange_entry;
int shq_exchange_plan(shq_context *ctx, const shq_exchange_layout *layout, const void *d_parts, int64_t numpart, const int32_t *d_target,
                      int ThisTask, int NTask, int64_t maxlast, int64_t *nexchange, int64_t *last, shq_exchange_entry *toGo);
int shq_exchange_pack(shq_context *ctx, const shq_exchange_layout *layout, void *d_parts, void *const d_slots[6], int64_t MaxPart,
                      const shq_exchange_entry *toGoOffset, int NTask, void *d_partbuf, void *const d_slotbuf[6]);
int shq_exchange_unpack(shq_context *ctx, const shq_exchange_layout *layout, void *d_parts, int64_t numpart_old, const int64_t slot_size_old[6],
                        const shq_exchange_entry *toGet, const shq_exchange_entry *toGetOffset, int NTask);
/* slots_gc (libgadget/slotsmanager.cpp:132-370): garbage particles squeezed out of the particle array (order kept: slots_gc_base),
 * then for every slot type with compact[type] != 0: ReverseLink = particle index (invalid for garbage: slots_gc_mark), unreferenced
 * slots squeezed out in order (slots_gc_sweep), PI of the surviving particles renumbered (slots_gc_collect).  *numpart and
 * slot_size[] are updated.  What the exchange runs between pack and receive when memory is short (exchange.hpp:398-406).
 * Slot records of a type that is not compacted keep their places, as in the reference. */
int shq_slots_gc(shq_context *ctx, const shq_exchange_layout *layout, void *d_parts, int64_t *numpart, int64_t MaxPart, void *const d_slots[6],
                 int64_t slot_size[6], const int compact[6]);
/* slots_gc_sorted (libgadget/slotsmanager.cpp:417-510): the particle array sorted by type, then by Peano-Hilbert key, garbage
 * (TypeKey 255) last and trimmed off; every enabled slot array sorted by its particles' new positions (ReverseLink), unreferenced
 * slots trimmed, PI renumbered.  d_keys[i] = PEANO(Part[i].Pos, BoxSize) comes from the caller (its own key function: the loop at
 * :439-447); equal (type, key) pairs keep their order here (the reference's sort is unstable). */
int shq_slots_gc_sorted(shq_context *ctx, const shq_exchange_layout *layout, void *d_parts, int64_t *numpart, int64_t MaxPart, void *const d_slots[6],
                        int64_t slot_size[6], const uint64_t *d_keys);

/* slots_split_particle and slots_convert (libgadget/slotsmanager.cpp:27-126) for lists of particles: what star formation does
 * with NewStars / NewParents (sfr_eff.cpp:344-372, placement = firststarslot + i), black-hole seeding (blackhole.cpp:1040) and the
 * wind spawns do one particle at a time.  The reference hands out new particle and slot indices with atomic counters, so their
 * order is the threads'; here entry k of the list gets NumPart + k / slot_size + k.  All pointers except the size arrays are device
 * pointers; the entries of a list must be distinct.
 *   shq_slots_split_particles   entry k: Generation of the parent ++ (4 bits of the flag byte, wrapping as the bit field does),
 *                               Base[NumPart + k] = Base[parent], child ID = (ID & 0x00ff..ff) + (Generation << 56), child Mass =
 *                               childmass[k], parent Mass -= childmass[k] (float, formed in double), child PI = -1; *numpart += n;
 *                               d_children (may be NULL) receives the new indices.  SHQ_ERR_NOMEM when NumPart + n > MaxPart
 *                               ("Tried to spawn ... no space left", :107), nothing touched.
 *   shq_slots_convert           entry k: the old slot (if the old type has slots and PI >= 0) gets ReverseLink = MaxPart + 100, a new
 *                               slot of ptype at slot_size[ptype] + k is filled with the poison byte 101 (slots_connect_new_slot)
 *                               and becomes the particle's PI, Type = ptype; slot_size[ptype] += n.  A ptype without slots only
 *                               changes Type.  SHQ_ERR_NOMEM when the new slots would pass slot_maxsize[ptype] ("Tried to use
 *                               non-allocated slot", :76): reserving is the caller's (sfr_reserve_slots, fof_seed), before the call. */
typedef struct shq_spawn_layout {
    size_t off_id, off_mass;    /* particle_data::ID (uint64), ::Mass (float) */
    int generation_shift;       /* first bit of the 4-bit Generation field inside the flag byte at shq_exchange_layout::off_flags */
    int pad_;
} shq_spawn_layout;
int shq_slots_split_particles(shq_context *ctx, const shq_exchange_layout *layout, const shq_spawn_layout *spawn, void *d_parts, int64_t *numpart,
                              int64_t MaxPart, const int32_t *d_parents, const double *d_childmass, int64_t n, int32_t *d_children);
int shq_slots_convert(shq_context *ctx, const shq_exchange_layout *layout, void *d_parts, int64_t numpart, int64_t MaxPart, void *const d_slots[6],
                      int64_t slot_size[6], const int64_t slot_maxsize[6], const int32_t *d_index, int64_t n, int ptype);

/* Scope note: star formation itself (the sfr_eff.cpp criteria, cooling, the random draws) is OUT of scope; this entry is
 * kept only as the slot-manager side of it — the one place where slots_convert / slots_split_particle (the
 * row these belong to) have to fill a freshly converted slot from another slot array while both stay in HBM.  The caller decides
 * who forms a star; nothing here evaluates star-formation physics.
 * make_particle_star (libgadget/sfr_eff.cpp:604-630) for the lists of the star-formation merge step (:344-372): entry k converts
 * children[k] (the parent itself, or the particle split off it) to a star at slot firststarslot + k and fills the slot from the
 * PARENT's gas slot as it was before the conversion: FormationTime = Time, LastEnrichmentMyr = TotalMassReturned = 0, BirthDensity,
 * VDisp, Metallicity, Metals[nmetals].  A parent that is not gas is SHQ_ERR_INVALID ("Only gas forms stars"), nothing touched.
 * Field types are the reference's (star: float FormationTime / LastEnrichmentMyr / BirthDensity / VDisp / Metals, double
 * TotalMassReturned / Metallicity; gas: double Density / VDisp / Metallicity, float Metals). */
typedef struct shq_star_spawn_layout {
    size_t star_formationtime, star_lastenrichmentmyr, star_totalmassreturned, star_birthdensity, star_vdisp, star_metallicity, star_metals;
    size_t sph_density, sph_vdisp, sph_metallicity, sph_metals;
    int nmetals, pad_;
} shq_star_spawn_layout;
int shq_make_particle_stars(shq_context *ctx, const shq_exchange_layout *layout, const shq_star_spawn_layout *sl, void *d_parts, int64_t numpart, int64_t MaxPart,
                            void *const d_slots[6], int64_t slot_size[6], const int64_t slot_maxsize[6], const int32_t *d_children, const int32_t *d_parents,
                            int64_t n, double Time);
/* blackhole_make_one (libgadget/blackhole.cpp:1029-1088) for a list of gas particles (fof_seed's ImportGroups[n].seed_index,
 * fof.cpp:1378-1381): conversion to type 5 + every field the reference initialises; seedmass[k] is BHP.Mass = Mseed
 * (SeedBlackHoleMass, or the caller's bh_powerlaw_seed_mass(ID) draw); with SeedBHDynMass > 0 the particle's mass moves to Mtrack
 * and becomes SeedBHDynMass, otherwise Mtrack = -1.  Not gas: SHQ_ERR_INVALID ("Only Gas turns into blackholes"). */
typedef struct shq_bh_seed_layout {
    size_t bh_mass, bh_mseed, bh_mdot, bh_formationtime, bh_swallowid, bh_density, bh_timebindynfric, bh_minpotpos, bh_dfaccel, bh_df_surroundingvel,
        bh_dragaccel, bh_df_surroundingrmsvel, bh_df_surroundingdensity, bh_jumptominpot, bh_countprogs, bh_mtrack, bh_kineticfdbkenergy, bh_vdisp;
    size_t part_pos, part_mass, part_timebin_hydro;
} shq_bh_seed_layout;
int shq_blackhole_make_seeds(shq_context *ctx, const shq_exchange_layout *layout, const shq_bh_seed_layout *bl, void *d_parts, int64_t numpart, int64_t MaxPart,
                             void *const d_slots[6], int64_t slot_size[6], const int64_t slot_maxsize[6], const int32_t *d_index, const double *d_seedmass, int64_t n,
                             double atime, double SeedBHDynMass);

/* Friends-of-friends groups of the resident particles (the first legacy-API user: libgadget/fof.cpp, one task).
 *   fof_label_primary (:368-581): particles of the primary types within LinkingLength of each other (r2 <= L^2, the neighbour
 *       test of treewalk_visit_ngbiter, treewalk.c:946-961) are one group; the reference's lock-free union-find (fofp_merge,
 *       update_root) ends in the connected components labelled with their smallest particle ID, and so does this one
 *       (atomicCAS hooking of the larger root under the smaller, path halving, one pass over a tree of the primary types);
 *   fof_label_secondary (:1142-1270): every particle of the secondary types takes the label of the nearest primary particle within
 *       a search radius that starts at max(0.4 L, 0.5 Hsml) (float) and doubles while it is below 4 L; ties go to the particle met
 *       first in the tree's depth-first order, as in the reference's nolist walk;
 *   fof_fof / fof_compile_base / fof_assign_grnr / add_particle_to_group / fof_finish_group_properties (:159-256, 583-766, 1048-1096):
 *       particles sorted by label, groups shorter than HaloMinLength dropped, GrNr = 1.. by decreasing length then MinID,
 *       Length / LenType / MassType / Mass / CM / Vel / Imom / Jmom / MaxDens + seed_index per group, groups ordered by MinID.
 * The reference's sort of the labels is not stable, so which member is FirstPos and the order of a group's sums are unspecified
 * there; here the member with the lowest particle index comes first and sums run in index order (one thread per group: deterministic).
 * The slot-resident sums (Sfr, metal masses, BH_Mass / BH_Mdot) stay with the caller: shq_fof_members hands it every group's member
 * list.  Several tasks (ghost labels through the export walk, fof_reduce_groups) are not covered.
 * shq_fof builds its own tree over the primary types (force_tree_rebuild_mask(&dmtree, ..., FOFPrimaryLinkTypes)) in the context's
 * tree storage and leaves the context WITHOUT a resident tree, as fof_fof frees its tree (fof.cpp:254): the next gravity / SPH /
 * export call needs shq_tree_build or shq_tree_upload first and returns SHQ_ERR_STATE otherwise.
 * ids: Part[].ID by particle index (host).  minid_by_particle / grnr_by_particle (host, may be NULL): HaloLabel[].MinID and Part[].GrNr
 * (-1 outside groups).  WindsDecoupleSph: winds_is_particle_decoupled applies (DelayTime > 0 gas never seeds). */
typedef struct shq_fof_params {
    double BoxSize;
    double LinkingLength;       /* FOFHaloComovingLinkingLength */
    int32_t PrimaryLinkTypes, SecondaryLinkTypes;   /* type masks */
    int32_t HaloMinLength;
    int32_t WindsDecoupleSph;
} shq_fof_params;
typedef struct shq_fof_group {
    uint64_t MinID;
    int32_t Length, GrNr;
    int32_t LenType[6];
    float FirstPos[3];
    int32_t seed_index;
    double MassType[6];
    double Mass;
    double CM[3];
    double Vel[3];
    double Imom[3][3];
    double Jmom[3];
    double MaxDens;
    int64_t first_member;       /* offset of the group's members in the list of shq_fof_members */
} shq_fof_group;
int shq_fof(shq_context *ctx, const shq_fof_params *params, const uint64_t *ids, uint64_t *minid_by_particle, int32_t *grnr_by_particle,
            int64_t *ngroups);
int shq_fof_groups_download(shq_context *ctx, shq_fof_group *groups, int64_t capacity);
/* The slot-resident sums of add_particle_to_group (fof.cpp:599-618: MassHeIonized, Sfr, GasMetalMass, GasMetalElemMass[],
 * StellarMetalMass, StellarMetalElemMass[], BH_Mdot, BH_Mass): per group and column the sum of values_by_particle[i][col] over the
 * group's members, in member (particle index) order.  The caller fills the columns from its slots (zero where a type does not
 * contribute); sums_by_group is [ngroups][ncol] in catalogue order.  Groups of more than 256 members are summed in 256 slices. */
int shq_fof_group_sums(shq_context *ctx, const double *values_by_particle, int ncol, double *sums_by_group);
/* The marking loop of fof_seed (fof.cpp:1290-1302) on the resident catalogue: the seed_index (densest gas particle) of every group
 * with Mass >= MinFoFMassForNewSeed, MassType[4] >= MinMStarForNewSeed, no black hole yet and a seed candidate, in catalogue order,
 * into the DEVICE array d_seed_index (NULL: count only) — the list shq_blackhole_make_seeds takes.  One task: seed_task is this task. */
int shq_fof_seed_select(shq_context *ctx, double MinFoFMassForNewSeed, double MinMStarForNewSeed, int32_t *d_seed_index, int64_t capacity, int64_t *nseeds);
/* particle indices of all kept groups, group after group (order of shq_fof_groups_download), members in index order */
int shq_fof_members(shq_context *ctx, int32_t *members, int64_t capacity, int64_t *nmembers);

/* Black-hole slot fields of the resident step (bh_particle_data, slotsmanager.h:35-73).  With them on the device
 * shq_drift repositions a BH with JumpToMinPot set (drift.cpp:32-53, when shq_set_bh_reposition is on), shq_kick_hydro adds
 * the dynamic-friction and drag kicks (timestep.cpp:973-979, factor bh_gravkick[TimeBinHydro] as apply_half_kick passes it),
 * the time-step loops read minTimeBin and write TimeBinDynFric.  download writes TimeBinDynFric and JumpToMinPot back. */
typedef struct shq_bh_dyn_view {
    void *base;
    size_t elsize;
    int64_t numslots;
    size_t off_mintimebin, off_timebindynfric, off_jumptominpot;       /* unsigned char, unsigned char, char */
    size_t off_dfaccel, off_df_surroundingvel, off_dragaccel;         /* MyFloat[3] (double) */
    size_t off_minpotpos, off_minpotvel;                              /* double[3], MyFloat[3] */
} shq_bh_dyn_view;
int shq_bh_dynamics_upload(shq_context *ctx, const shq_part_view *parts, const shq_bh_dyn_view *bh);
int shq_bh_dynamics_download(shq_context *ctx, const shq_part_view *parts, const shq_bh_dyn_view *bh);
int shq_set_bh_reposition(shq_context *ctx, int enabled);
int shq_kick_bh(shq_context *ctx, const double gravkick[SHQ_TIMEBINS + 1], const int32_t *active, int64_t nactive);
/* active: host int32 list or NULL. The walk and postprocess are queued on the stream. */
int shq_grav_short_run(shq_context *ctx, const shq_grav_params *params, const int32_t *active,
                       int64_t nactive, int update_potential, int walk_mode);
/* The same walk + postprocess for the own particles [first, first + count) without a list: a piece of the work-set
 * TreeWalk::run_on_queue (libgadget/treewalk2.h:282-330) is given.  The pieces of one evaluation must start at first = 0 and
 * together cover what shq_grav_short_run(active = NULL) covers; their interaction statistics add up (kernel_ms is the last
 * piece's).  Lets the caller queue other work on the stream between the pieces. */
int shq_grav_short_run_range(shq_context *ctx, const shq_grav_params *params, int64_t first, int64_t count,
                             int update_potential, int walk_mode);
/* Copy results back: accel[numpart][3] (may be NULL), potential[numpart] (may be NULL),
 * ninteractions[numpart] (may be NULL). Synchronises. */
/* Wave-level walk counters in shq_walk_stats (nnodes_visited, nwave_*, nnode_interactions): 0 off (default: they stay
 * zero; the counters cost the walk ~4 %, they press on its scalar register budget), 1 on, 2 plus per-lane-participation
 * histograms printed to stderr.  ninteractions, min / max and kernel_ms are always filled. */
int shq_set_walk_stats(shq_context *ctx, int level);
/* How shq_grav_short_run launches the exact primary walk (a tuning / test knob; results are the same interaction sets either way).
 * persist: 0 one 64-target task per wave; 1 (default) persistent waves taking tasks from per-XCD counters once the launch exceeds
 * what the chip holds at once; 2 persistent waves for every launch.  leaf_ring: 1 (default) leaf particles go through the
 * wave-private LDS ring of the persistent walk, 0 they are evaluated as the leaf is opened (the reference's summation order). */
int shq_set_walk_launch(shq_context *ctx, int persist, int leaf_ring);
/* The production launch (persistent waves + leaf ring, relative criterion) does not enter a subtree that at most eight of a wave's 64
 * lanes open: it notes it, and a second kernel walks the noted subtrees with one lane per (target, node) pair: the
 * same interaction sets, a different order of the sum (forces to 1e-13 of the largest).  On by default; 0 (or SHQ_WALK_SPARSE=0)
 * makes the main walk enter every subtree itself.  A test / tuning knob like shq_set_walk_launch.
 * The pair kernel fetches the first 80 bytes of a node's record and recomputes the products of {mass, len} and the walk parameters that
 * fill the rest (same expressions, same bits) whenever the device found that true of every record of the pool when it last filled
 * them; 2 makes it fetch whole records regardless.  shq_walk_pair_lean: 1 if the current pool passed that check, 0 if not (or not
 * checked yet), < 0 on error. */
int shq_set_walk_sparse(shq_context *ctx, int enable);
int shq_walk_pair_lean(shq_context *ctx);
/* Where the pair kernel runs: 1 (default; SHQ_WALK_OVERLAP) beside the main walk on the context's second stream once a launch has
 * at least 128 tasks per CU — three workgroups of the main walk and one of the pair kernel per CU, the pair kernel taking a task when
 * the main walk has raised its flag (results, noted subtrees and OldAcc cross at the device's coherence point); 0 behind it on the
 * same stream; 2 beside it whatever the size of the launch (tests).  Same interaction sets and the same sums in every mode. */
int shq_set_walk_overlap(shq_context *ctx, int mode);
/* The pair kernel's failures are STICKY and loud (the reference checks every launch and ends the run, treewalk2.cuh:351-353).
 *  - A pair stack that ran full dropped pairs: the accelerations of that launch are incomplete.  The error word stays up on the device
 *    whatever is launched afterwards; a copy follows every launch to pinned host memory, and shq_synchronize, every *_download,
 *    shq_kick_short, shq_hier_refine, shq_walk_pair_status (which wait for the stream) and the next shq_treepm_step /
 *    shq_grav_short_run (which do not: they see the launches completed by then) return SHQ_ERR_DEVICE once; the report clears it.
 *  - A wave of the pair kernel running BESIDE the main walk that gave up waiting for a task (the two kernels were not co-resident:
 *    nothing in HIP promises they are) is not an error any more: a mop-up pass behind both kernels walks every task not marked done,
 *    in the order of the pair kernel run behind the walk (same sums, same bits).  It returns at once when nobody gave up.
 * shq_walk_pair_status: launches since shq_init that needed the mop-up pass, the deepest pair stack seen (entries, of SHQ_SPARSE_STACK
 * = 4096 per wave), the tasks the last launch's mop-up walked; waits for the stream and returns the sticky error like the others.
 * shq_set_walk_debug (tests): pair_spin_max > 0 = polls of a task's flag before a live pair wave gives up (default 2^22; 1 starves
 * it: every task goes to the mop-up pass); pair_stack_cap > 0 = pairs per wave stack (1344 .. 4096; small values force the overflow). */
int shq_walk_pair_status(shq_context *ctx, int64_t *recovered_launches, int64_t *stack_high_water, int64_t *last_mopped_tasks);
int shq_set_walk_debug(shq_context *ctx, int pair_spin_max, int pair_stack_cap);
/* Checker utility: direct summation as the reference's own gravity test does it (force_direct / grav_force,
 * libgadget/tests/test_gravity.cpp:41-76,121-143): accel[ns][3] (host) = acceleration at the ns sample positions (host, [ns][3]) from
 * the first nsrc resident particles and their (2 repeat + 1)^3 periodic images, spline-softened below h.  Partial sums over a
 * rank's local particles add up across ranks (bench.py's sampled force check of the sharded TreePM step). */
int shq_direct_force_sample(shq_context *ctx, const double *sample_pos, int64_t ns, int64_t nsrc, double BoxSize, double G, double h,
                            int repeat, double *accel);
int shq_grav_short_download(shq_context *ctx, double (*accel)[3], double *potential,
                            int64_t *ninteractions, shq_walk_stats *stats);
/* Secondary walk: TreeWalk::ev_secondary (libgadget/treewalk2.h:618-700) with GravLocalTreeWalk::visit
 * <TREEWALK_GHOSTS> (gravshort2.hpp:243-322).  `queries` are the imported GravTreeQuery records of other ranks'
 * particles (binary mirror: Pos[3], NodeList[NODELISTLENGTH = 4], OldAcc; gravshort2.hpp:123-128,
 * localtreewalk2.h:76-115); each walks the branches under the top-level nodes of its NodeList (node numbers
 * of the uploaded tree, -1 terminated) of the context's current tree and particles.  `results` receive the raw
 * GravTreeResult sums (Acc not multiplied by G, Potential without the self term: the exporting rank reduces and
 * post-processes them, treewalk2.h:780-812).  Synchronous; host pointers.  With this entry point the library
 * also serves under shenqi's own export / import machinery on multi-rank runs. */
typedef struct shq_grav_query {
    double Pos[3];
    int32_t NodeList[4];
    double OldAcc;
} shq_grav_query;
typedef struct shq_grav_result {
    double Acc[3];
    double Potential;
} shq_grav_result;
int shq_grav_short_secondary(shq_context *ctx, const shq_grav_params *params, const shq_grav_query *queries, int64_t nq,
                             shq_grav_result *results, int64_t *ninteractions, int update_potential);
/* ---- top-tree walk: which remote top-leaves must a target visit ----
 * The reference's ev_count_exports + ev_toptree (treewalk2.cuh:243-334) for the host's own export / import
 * machinery (treewalk2.h:618-812): per target, walk only the TopLevel nodes of the host tree; every pseudo node
 * that the walk would open becomes an export, consecutive leaves of one task coalescing into NodeList[4]
 * (TopTreeWalk::export_particle / export_count, localtreewalk2.h:269-324).
 *
 * shq_toptree_upload: the TopLevel nodes of `tree` and the domain's TopLeaves table (pseudo node suns[0] - lastnode
 *     indexes it).  Only needed when the tree has pseudo nodes; independent of shq_tree_upload.
 * shq_grav_toptree_exports: GravTopTreeWalk::toptree_visit (gravshort2.hpp:362-438) for the resident targets
 *     (positions, OldAcc as for shq_grav_short_run).
 * shq_ngb_toptree_exports: TopTreeWalk::toptree_visit with cull_node<symmetric> (localtreewalk2.h:154-182, 210-259),
 *     search radius = the resident Hsml (density: symmetric 0; hydro: symmetric 1, needs the top nodes' hmax).
 * Outputs: exportcounts[t] = inclusive running total of exports up to target t (the reference's scanned
 * exportcounts, used by its BunchSize logic; may be NULL); table = the DataIndexTable, target t's entries contiguous at
 * [exportcounts[t-1], exportcounts[t]) in the reference's order; *nexport always returns the total.  With table == NULL
 * only counts are produced; with capacity < total nothing is written and SHQ_ERR_NOMEM is returned. */
typedef struct shq_data_index { /* struct data_index, libgadget/localtreewalk2.h:188-193 */
    int32_t Task;
    int32_t Index;
    int32_t NodeList[4];
} shq_data_index;
int shq_toptree_upload(shq_context *ctx, const shq_tree_view *tree, const shq_topleaf *topleaves, int ntopleaves);
int shq_grav_toptree_exports(shq_context *ctx, const shq_grav_params *params, const int32_t *active, int64_t nactive,
                             int32_t *exportcounts, shq_data_index *table, int64_t capacity, int64_t *nexport);
int shq_ngb_toptree_exports(shq_context *ctx, int symmetric, double BoxSize, const int32_t *active, int64_t nactive,
                            int32_t *exportcounts, shq_data_index *table, int64_t capacity, int64_t *nexport);
/* The same export detection with everything left in HBM (what ev_count_exports + ev_toptree do on managed memory,
 * treewalk2.cuh:243-334): active = NULL (all own particles), a resident handle, or — active_on_device != 0 — a DEVICE list.  Back
 * come the number of exports and task_counts[ntask], the entries per destination task (the send counts of
 * ev_send_recv_export_import, treewalk2.h:618-700); the table stays resident.  shq_grav_export_pack then writes, on the device,
 * the GravTreeQuery records of the table in task order (entries of one task in table order) into d_queries and the particle
 * index of every record into d_place (the `place` argument of shq_grav_reduce_export_results): the send buffer of the caller's
 * all-to-all, never on the host. */
int shq_grav_toptree_exports_resident(shq_context *ctx, const shq_grav_params *params, const int32_t *active, int64_t nactive,
                                      int active_on_device, int ntask, int64_t *nexport, int64_t *task_counts);
int shq_grav_export_pack(shq_context *ctx, shq_grav_query *d_queries, int32_t *d_place);
/* GravTreeResult::reduce<TREEWALK_GHOSTS> (gravshort2.hpp:136-146) for n returned results: Accel[place[k]] += results[k].Acc
 * and, with update_potential, Potential likewise, in the order k = 0..n-1 (the reference's loop over its export table, so
 * a target's partial sums are added in the same order).  Needs a preceding shq_grav_short_run with
 * SHQ_WALK_DEFER_POSTPROCESS.  shq_grav_postprocess: GravTreeOutput::postprocess (gravshort2.hpp:88-107) for the
 * targets of that run (same active argument). */
int shq_grav_reduce_export_results(shq_context *ctx, const int32_t *place, const shq_grav_result *results, int64_t n,
                                   int update_potential);
int shq_grav_postprocess(shq_context *ctx, const shq_grav_params *params, const int32_t *active, int64_t nactive,
                         int update_potential);
/* Set the per-particle OldAcc inputs on the device from the device-resident
 * FullTreeGravAccel + GravPM of the last shq_grav_short_run / shq_pm_run
 * (grav_get_abs_accel, gravshort2.hpp:111-121). */
int shq_grav_refresh_oldacc(shq_context *ctx, double G);

/* ---- SPH density and hydro force ------------------------------------------------------- */

#define SHQ_DENSITY_KERNEL_CUBIC_SPLINE 1   /* enum DensityKernelType, libgadget/density2.h */
#define SHQ_DENSITY_KERNEL_QUINTIC_SPLINE 2
#define SHQ_DENSITY_KERNEL_QUARTIC_SPLINE 4

/* POD mirror of KickFactorData (libgadget/density2.h:52-129): the host (timebinmgr) fills it. */
typedef struct shq_kick_factors {
    double FgravkickB;
    double gravkicks[SHQ_TIMEBINS + 1];
    double hydrokicks[SHQ_TIMEBINS + 1];
    double dloga_kick[SHQ_TIMEBINS + 1];
    double dloga_for_bin[SHQ_TIMEBINS + 1];
} shq_kick_factors;

/* POD mirror of DensityPriv + the density_params it reads (libgadget/densitytree2.hpp:10-52,
 * density2.h:16-33). */
typedef struct shq_density_params {
    double BoxSize;
    double DesNumNgb;           /* GetNumNgb(kernel) */
    double DesNumNgbBH;         /* DesNumNgb * BlackHoleNgbFactor */
    double MinGasHsml;
    double MaxNumNgbDeviation;
    int32_t update_hsml;
    int32_t BlackHoleOn;
    int32_t DoEgyDensity;
    int32_t WindsDecouple;      /* winds_ever_decouple() */
    int32_t DensityKernelType;
    int32_t pad_;
    shq_kick_factors kf;
} shq_density_params;

/* POD mirror of HydroPriv (libgadget/hydratree2.hpp:60-125). */
typedef struct shq_hydro_params {
    double BoxSize;
    double atime;
    double fac_mu;              /* pow(atime, 3(GAMMA-1)/2) / atime */
    double fac_vsic_fix;        /* hubble * pow(atime, 3 GAMMA_MINUS1) */
    double hubble_a2;           /* hubble * atime^2 */
    double drifts[SHQ_TIMEBINS + 1];
    double ArtBulkViscConst;
    double DensityContrastLimit;
    double WindSpeed;
    double WindFreeTravelDensThresh;
    int32_t DensityIndependentSphOn;
    int32_t DensityKernelType;
    shq_kick_factors kf;
} shq_hydro_params;

/* View of the BH slots density() writes for Type-5 targets (bh_particle_data.Density/.DivVel). */
typedef struct shq_bh_view {
    void *base;
    size_t elsize;
    int64_t numslots;
    size_t off_density, off_divvel;
} shq_bh_view;

typedef struct shq_sph_stats {
    int64_t ntargets;
    int64_t ninteractions;   /* ngbiter calls summed over targets and iterations */
    int32_t niterations;     /* Hsml iterations (do_hsml_loop, treewalk2.h:480-557) */
    int32_t pad_;
    double kernel_ms;        /* HIP-event time of the walk kernels */
    double hsml_max_tried;   /* density calls: the largest Hsml any walk of the loop searched with (0 for other operators).  A sharded
                                caller sizes its ghost halo by this, not by the Hsml the loop ends with: an intermediate guess beyond the
                                halo undercounts NumNgb and steers the iteration (treewalk2.h:480-557 runs every guess against all ranks) */
} shq_sph_stats;

/* One-shot replacement of density_cuda() *including* the host Hsml loop the reference keeps on
 * the CPU (do_hsml_loop + DensityOutput::postprocess + density_check_neighbours,
 * treewalk2.h:480-557, densitytree2.hpp:117-257).  Targets: active particles of Type 0 / 5 that
 * are neither garbage nor swallowed (DensityQuery::haswork).  Writes Part[].Hsml / DtHsml,
 * SphP[].Density / EgyWtDensity / DhsmlEgyDensityFactor / DivVel / CurlVel, BhP[].Density /
 * DivVel, and - as update_tree_hmax_father does (forcetree.cpp:1285-1313) - raises mom.hmax of
 * the leaf holding each finished target in the caller's NODE array.  EntVarPred[numslots]
 * (out, may be NULL) receives the predicted entropy variable handed to hydro
 * (density2.cpp:147); GradRho_mag[numslots] (out, may be NULL) |grad rho| (density2.cpp:135-143).
 * Returns SHQ_ERR_NOCONV if MAXITER (400) is exceeded. */
int shq_density(shq_context *ctx, const shq_tree_view *tree, shq_node *nodes_rw,
                const shq_part_view *parts, const shq_sph_view *sph, const shq_bh_view *bh,
                const int32_t *active, int64_t nactive, const shq_density_params *params,
                double *EntVarPred, double *GradRho_mag, shq_sph_stats *stats);

/* One-shot replacement of hydro_force_cuda(): symmetric neighbour walk (cull radius
 * max(hmax_node, Hsml_i)), HydroResult::reduce and HydroOutput::postprocess
 * (hydratree2.hpp:127-149,201-379).  Writes SphP[].HydroAccel / DtEntropy / MaxSignalVel for
 * the active Type-0 targets.  EntVarPred may be NULL (then computed per particle). */
int shq_hydro_force(shq_context *ctx, const shq_tree_view *tree, const shq_part_view *parts,
                    const shq_sph_view *sph, const int32_t *active, int64_t nactive,
                    const shq_hydro_params *params, const double *EntVarPred, shq_sph_stats *stats);

/* ---- SPH walks in phases: the reference's distributed walk (treewalk2.h:277-373, do_hsml_loop :480-557) ----------
 * shq_density / shq_hydro_force above are  open -> { ev_primary -> ev_postprocess } -> close  in one call.  A multi-rank
 * run under shenqi's own export / import machinery drives the same phases itself — they are the hooks a TreeWalk
 * backend overrides (treewalk2.cuh:212-394) — and inserts the exchange between ev_primary and ev_postprocess:
 *
 *   shq_density_open(...)                        upload, predicted velocities, Left/Right bounds, work queue (haswork)
 *   repeat:
 *     shq_density_ev_primary(ctx)                raw sums of the queue's targets over the local tree (pseudo nodes skipped)
 *     shq_sph_exports(ctx, ...)                  top-tree walk of the queue with its current Hsml -> data_index table
 *                                                (needs shq_toptree_upload; symmetric for hydro)
 *     shq_sph_fill_queries(ctx, table, n, q)     DensityQuery / HydroQuery records of the table's entries
 *     ... MPI: queries out, other ranks' queries in ...
 *     shq_density_ev_secondary(ctx, ...)         visit<TREEWALK_GHOSTS> of imported queries against the local tree: raw
 *                                                DensityResult sums (any rank may call this while its own walk is open)
 *     ... MPI: results back ...
 *     shq_density_ev_reduce(ctx, place, res, n)  DensityResult::reduce<TREEWALK_GHOSTS>, entries of a target in order
 *     shq_density_ev_postprocess(ctx, &nredo)    DensityOutput::postprocess + Hsml update; nredo = targets to redo
 *   until every rank reports nredo == 0 (ranks with an empty queue keep serving secondaries)
 *   shq_density_close(...)                       results back into the caller's arrays
 *
 * Hydro is the same with one pass.  The structs are binary mirrors of DensityQuery / DensityResult
 * (densitytree2.hpp:260-344) and HydroQuery / HydroResult (hydratree2.hpp:151-228). */
typedef struct shq_density_query {
    double Pos[3];
    int32_t NodeList[4];
    double Vel[3];
    double Hsml;
    int32_t Type;
    int32_t pad_;
} shq_density_query;               /* 80 bytes */
typedef struct shq_density_result {
    double EgyRho, DhsmlEgyDensity, Rho, DhsmlDensity, Ngb, Div;
    double Rot[3];
    double GradRho[3];
} shq_density_result;              /* 96 bytes */
typedef struct shq_hydro_query {
    double Pos[3];
    int32_t NodeList[4];
    double EgyRho, EntVarPred;
    double Vel[3];
    double Hsml, Mass, Density, Pressure, F1, SPH_DhsmlDensityFactor;
    int32_t TimeBinHydro;
    int32_t pad_;
} shq_hydro_query;                 /* 136 bytes */
typedef struct shq_hydro_result {
    double Acc[3];
    double DtEntropy, MaxSignalVel;
} shq_hydro_result;                /* 40 bytes */

int shq_density_open(shq_context *ctx, const shq_tree_view *tree, const shq_part_view *parts, const shq_sph_view *sph,
                     const shq_bh_view *bh, const int32_t *active, int64_t nactive, const shq_density_params *params,
                     int want_gradrho, int64_t *nqueue);
int shq_density_ev_primary(shq_context *ctx);
int shq_density_ev_secondary(shq_context *ctx, const shq_density_params *params, const shq_density_query *queries, int64_t nq,
                             shq_density_result *results, int64_t *ninteractions_total);
int shq_density_ev_reduce(shq_context *ctx, const int32_t *place, const shq_density_result *results, int64_t n);
int shq_density_ev_postprocess(shq_context *ctx, int64_t *nredo);
int shq_density_close(shq_context *ctx, shq_node *nodes_rw, const shq_part_view *parts, const shq_sph_view *sph,
                      const shq_bh_view *bh, double *EntVarPred, double *GradRho_mag, shq_sph_stats *stats);

int shq_hydro_open(shq_context *ctx, const shq_tree_view *tree, const shq_part_view *parts, const shq_sph_view *sph,
                   const int32_t *active, int64_t nactive, const shq_hydro_params *params, const double *EntVarPred,
                   int64_t *nqueue);
int shq_hydro_ev_primary(shq_context *ctx);
int shq_hydro_ev_secondary(shq_context *ctx, const shq_hydro_params *params, const shq_hydro_query *queries, int64_t nq,
                           shq_hydro_result *results, int64_t *ninteractions_total);
int shq_hydro_ev_reduce(shq_context *ctx, const int32_t *place, const shq_hydro_result *results, int64_t n);
int shq_hydro_ev_postprocess(shq_context *ctx);
int shq_hydro_close(shq_context *ctx, const shq_part_view *parts, const shq_sph_view *sph, shq_sph_stats *stats);

/* ---- the SPH operators on a gas set that is already in HBM (sharded runs: local + imported ghost gas as rows of a device buffer
 * that came out of the all-to-all; shenqi_amd/dist.py DistSPHDevice) -----------------------------------------------------------
 * A row holds what density() / hydro_force() read of a neighbour and write of a target, SHQ_GAS_NCOL doubles:
 *    0-2 Pos   3 Mass   4-6 Vel   7 Hsml   8-10 FullTreeGravAccel   11-13 GravPM   14-16 HydroAccel   17 Entropy   18 DtEntropy
 *    19 DelayTime   20 Density   21 EgyWtDensity   22 DhsmlEgyDensityFactor   23 DivVel   24 CurlVel   25 MaxSignalVel   26 DtHsml
 *    27 TimeBinGravity + 256 TimeBinHydro (as a double)
 * shq_gas_set_device : the n rows become the resident particle set (all Type 0; the first nlocal are this rank's own = the
 *                      targets); any tree is dropped: shq_tree_build(ctx, BoxSize, GASMASK, NULL, 0, ...) comes next.
 * shq_density_resident / shq_hydro_resident : density() with its Hsml loop (densitytree2.hpp:117-257, treewalk2.h:480-557) and
 *                      hydro_force() (hydratree2.hpp:127-379) for the nlocal targets over the tree of all n; EntVarPred is evaluated
 *                      per particle.  Same kernels as shq_density / shq_hydro_force.
 * shq_gas_get_device : results back into the first n rows of d_rows: which & 1: Hsml, DtHsml, Density, EgyWtDensity,
 *                      DhsmlEgyDensityFactor, DivVel, CurlVel; which & 2: HydroAccel, DtEntropy, MaxSignalVel.
 * Only statistics cross PCIe. */
#define SHQ_GAS_NCOL 28
int shq_gas_set_device(shq_context *ctx, const double *d_rows, int64_t n, int64_t nlocal);
int shq_gas_get_device(shq_context *ctx, double *d_rows, int64_t n, int which);
int shq_density_resident(shq_context *ctx, const shq_density_params *params, shq_sph_stats *stats);
int shq_hydro_resident(shq_context *ctx, const shq_hydro_params *params, shq_sph_stats *stats);

/* For the walk that is open: the export table of its current queue (TopTreeWalk::toptree_visit with cull_node, symmetric
 * for hydro; outputs as shq_ngb_toptree_exports, exportcounts indexed by queue position) and the query records of a
 * table's entries (queries: n records of shq_density_query or shq_hydro_query, whichever walk is open). */
int shq_sph_exports(shq_context *ctx, int32_t *exportcounts, shq_data_index *table, int64_t capacity, int64_t *nexport);
int shq_sph_fill_queries(shq_context *ctx, const shq_data_index *table, int64_t n, void *queries);

/* ---- stellar density ------------------------------------------------------------------------------
 * stellar_density() (libgadget/stellar_density2.cpp:306-341): the SPH volume weight sum(m_j / rho_j [* w_k]) of the gas
 * around each star of `queue` (the reference's StarQueue: build_stellar_density_queue, :285-303, stays on the host — it reads
 * the star slots), with its own Hsml iteration over ten trial radii per walk (stellareffhsml, ngbiter, postprocess,
 * ngb_narrow_down).  `tree` is the gas tree (GASMASK); gas densities come from the SPH view.  Writes Part[].Hsml of the
 * stars and StarVolumeSPH[PI of the star].  A star with Hsml == 0 is refused (the reference re-seeds it from its father
 * node).  SHQ_ERR_NOCONV beyond MAXITER iterations. */
typedef struct shq_stellar_params {
    double BoxSize;
    double DesNumNgb;           /* GetNumNgb(GetDensityKernelType()) */
    double MaxNgbDeviation;
    int32_t SPHWeighting;
    int32_t DensityKernelType;
} shq_stellar_params;
int shq_stellar_density(shq_context *ctx, const shq_tree_view *tree, const shq_part_view *parts, const shq_sph_view *sph,
                        const int32_t *queue, int64_t nqueue, const shq_stellar_params *params, double *StarVolumeSPH,
                        shq_sph_stats *stats);

/* blackhole_veldisp() (libgadget/veldisp2.cpp:164-199): for the active black holes (type 5, not garbage / swallowed) the
 * number of dark-matter particles inside Hsml and the first and second moments of their predicted velocities
 * (DM_VelPred, density2.h:104-111) relative to the hole's, and VDisp = sqrt((V2/N - |V1/N|^2) / 3) where that is positive
 * (BHVelDispOutput::postprocess, :49-63).  `tree` is the dark-matter tree (DMMASK); the particle view needs Vel,
 * FullTreeGravAccel, GravPM, Hsml, TimeBinGravity and PI.  Outputs are indexed by black-hole slot PI; NumDM / V1sumDM /
 * V2sumDM may be NULL; VDisp[PI] is written only where NumDM > 0 and the variance is positive, as the reference does. */
int shq_bh_veldisp(shq_context *ctx, const shq_tree_view *tree, const shq_part_view *parts, const int32_t *active, int64_t nactive,
                   const shq_kick_factors *kf, double *NumDM, double (*V1sumDM)[3], double *V2sumDM, double *VDisp);

/* winds_find_vel_disp(), wind part (libgadget/veldisp2.cpp:203-528): for the gas particles of `queue` (the reference's
 * ActiveVDisp from build_vdisp_queue, :376-396, stays on the host: it reads densities and the star-formation threshold) the
 * 1-D velocity dispersion of the ~40 nearest dark-matter particles, found by a density-like loop over five trial radii per
 * walk (NUMDMNGB 40 +- 1), with the Hubble flow hubble * Time^2 * dist in the relative velocity.  `tree` is the dark-matter
 * tree; the particle view needs Vel, FullTreeGravAccel, GravPM, Hsml (the starting DMRadius), TimeBinGravity and PI.
 * VDisp[PI] is written where the reference sets SphP.VDisp (positive variance).  SHQ_ERR_NOCONV beyond MAXITER. */
int shq_wind_veldisp(shq_context *ctx, const shq_tree_view *tree, const shq_part_view *parts, const int32_t *queue, int64_t nqueue,
                     const shq_kick_factors *kf, double Time, double hubble, double *VDisp, shq_sph_stats *stats);

/* blackhole_minpot() and the treewalk of blackhole_dynfric() (libgadget/bhdynfric.cpp:44-295, 313-345): for the black holes
 * of `queue` (ActiveBlackHoles / DynFricActive) over the caller's tree (ALLMASK for repositioning; STARMASK + BHMASK, plus
 * DMMASK for BH_DynFrictionMethod > 1, for friction — `typemask` says which types of the tree's particles count):
 *   - the neighbour of lowest Potential inside the hole's Hsml: MinPot / MinPotPos / MinPotVel of slot PI are replaced where it
 *     lies below the value already there (BHReposResult::reduce, :106-118; the caller initialises them as
 *     blackhole_init_potential does, :296-310) and `updated[PI]` is set to 1;
 *   - with method > 0 the friction sums after BHDynFricOutput::postprocess (:66-82): DF_SurroundingDensity (the raw
 *     kernel-weighted mass), DF_SurroundingVel and DF_SurroundingRmsVel normalised where the density is positive.
 * The particle view needs Vel, FullTreeGravAccel, GravPM, Potential, Hsml, TimeBinGravity, Type and PI. */
typedef struct shq_bh_dynfric_out {
    double *MinPot;               /* [nslots], in / out */
    double (*MinPotPos)[3];       /* in / out */
    double (*MinPotVel)[3];       /* in / out */
    int32_t *updated;             /* [nslots] or NULL */
    double *DF_SurroundingDensity; /* out (method > 0), may be NULL */
    double (*DF_SurroundingVel)[3];
    double *DF_SurroundingRmsVel;
} shq_bh_dynfric_out;
int shq_bh_dynfric(shq_context *ctx, const shq_tree_view *tree, const shq_part_view *parts, const int32_t *queue, int64_t nqueue,
                   const shq_kick_factors *kf, int BH_DynFrictionMethod, int DensityKernelType, int typemask,
                   const shq_bh_dynfric_out *out);

/* The two legacy-API tree walks of blackhole() (libgadget/blackhole.cpp:217-370), after blackhole_dynfric:
 * symmetric neighbour search over gas + black holes on the caller's tree (GASMASK + BHMASK with hmax: blackhole() calls
 * force_tree_calc_moments when it is missing), one black hole per lane of the SPH operators' wave-collective walk.
 *   shq_bh_accretion   blackhole_accretion_ngbiter / _reduce / _postprocess (:373-692) for the holes of `queue` (ActiveBlackHoles):
 *       BH_SwallowID marks for mergers (r < 2 ForceSoftening / 2.8, the reposition / MergeGravBound rule with check_grav_bound,
 *       the reference's compare-and-swap: the larger ID swallows, an inactive hole can be swallowed by a smaller one);
 *       SPH_SwallowID marks for stochastically swallowed gas (the largest ID + 1 that drew the particle: w = Table[ID % size] <
 *       (BH_Mass - Mass or Mtrack) wk / Density); BH_FeedbackWeightSum, BH_Entropy, BH_SurroundingGasVel, MgasEnc; then Mdot
 *       (Bondi-Hoyle, Eddington cap), BHP.Mass += Mdot dtime, DragAccel, KineticFdbkEnergy / KEflag.  encounter, Mdot, Mass,
 *       DragAccel, KineticFdbkEnergy are written into the slots; the BHPriv arrays into `work` (by slot index, as there).
 *   shq_bh_feedback    blackhole_feedback_ngbiter / _reduce / _postprocess (:728-965) for the holes of `queue` that are not marked
 *       themselves: marked holes and gas are swallowed (Swallowed / slots_mark_garbage, SwallowID, SwallowTime, mass, momentum,
 *       CountProgs), thermal energy into the unswallowed gas inside the kernel (compare-and-swap on the entropy, capped at
 *       MaxThermalU; BHHeated where eeqos[i] != 0) or the released kinetic energy as kicks in the direction get_random_dir draws,
 *       minTimeBin; then BHP.Mass, Part.Vel, Mtrack / Part.Mass and the KineticFdbkEnergy reset.  Particle flags, Vel, Mass, the gas
 *       Entropy and the slots are updated in the caller's arrays.
 * Derived constants are the caller's: EddingtonConst = 4 pi GRAVITY LIGHTCGS PROTONMASS / (0.1 LIGHTCGS^2 THOMPSON) (:379),
 * LightOverUnitVel = LIGHTCGS / UnitVelocity_in_cm_per_s, MaxThermalU = 5e8 / u_to_temp_fac (add_injected_BH_energy, :700-710),
 * Hubble = CP->Hubble, hubble = hubble_function(atime).  The random table is RandTable::Table.  ids[i] = Part[i].ID. */
typedef struct shq_bh_params {
    double BoxSize, ForceSoftening, SeedBHDynMass, atime, a3inv, hubble, GravInternal;
    double BlackHoleAccretionFactor, BlackHoleEddingtonFactor, BlackHoleFeedbackFactor;
    double EddingtonConst, UnitTime_in_s, HubbleParam, LightOverUnitVel, MaxThermalU, OmegaBaryon, Hubble;
    double BHKE_EddingtonThrFactor, BHKE_EddingtonMFactor, BHKE_EddingtonMPivot, BHKE_EddingtonMIndex, BHKE_EffRhoFactor, BHKE_EffCap, BHKE_InjEnergyThr,
        BHKE_SfrCritOverDensity;
    int DensityKernelType, WindsDecoupleSph, RepositionEnabled, MergeGravBound, BH_DRAG, BlackHoleKineticOn;
} shq_bh_params;
typedef struct shq_bh_slot_view {
    void *base;
    size_t elsize;
    int64_t numslots;
    size_t off_mass, off_mdot, off_density, off_mtrack, off_dfaccel, off_vdisp, off_kineticfdbkenergy, off_dragaccel, off_encounter, off_countprogs,
        off_mintimebin, off_swallowid, off_swallowtime;
} shq_bh_slot_view;
typedef struct shq_bh_work {        /* struct BHPriv, blackhole.h:9-45: the caller's arrays, by slot index */
    uint64_t *SPH_SwallowID;        /* [gas slots] */
    uint64_t *BH_SwallowID;         /* [black-hole slots] */
    double *BH_FeedbackWeightSum, *BH_Entropy;
    double (*BH_SurroundingGasVel)[3];
    double *MgasEnc;
    int32_t *KEflag;
    double *BH_accreted_Mass, *BH_accreted_BHMass;   /* feedback only */
    double (*BH_accreted_momentum)[3];
} shq_bh_work;
int shq_bh_accretion(shq_context *ctx, const shq_tree_view *tree, const shq_part_view *parts, const shq_sph_view *sph, const shq_bh_slot_view *bh,
                     const uint64_t *ids, const int32_t *queue, int64_t nqueue, const shq_kick_factors *kf, const shq_bh_params *params, int64_t Ti_Current,
                     const double *rnd_table, int64_t rnd_size, const shq_bh_work *work);
int shq_bh_feedback(shq_context *ctx, const shq_tree_view *tree, const shq_part_view *parts, const shq_sph_view *sph, const shq_bh_slot_view *bh,
                    const uint64_t *ids, const int32_t *queue, int64_t nqueue, const shq_kick_factors *kf, const shq_bh_params *params, int64_t MaxPart,
                    const double *rnd_table, int64_t rnd_size, const uint8_t *eeqos, const shq_bh_work *work, int64_t *n_sph_swallowed,
                    int64_t *n_bh_swallowed);

/* winds_and_feedback() (libgadget/winds.cpp:295-369): the two asymmetric walks over the gas tree for the new
 * stars of the step — sfr_wind_weight_ngbiter (:411-447: mass of the gas inside the star's Hsml that is not already a wind particle,
 * into TotalWeight[star slot]) and sfr_wind_feedback_ngbiter (:510-565: a gas particle becomes a kick candidate when
 * Table[(star ID + gas ID) % size] < windeff Mass / TotalWeight, with get_wind_params, :489-507) — then the reference's resolution:
 * the candidates sorted by (particle, distance, star ID), the first of every particle kicks (wind_do_kick, :449-471: Vel along
 * get_wind_dir, Entropy += therm / enttou, DelayTime when winds decouple); the results go into the caller's arrays.  WindModel carries the reference's
 * flag bits (WIND_DECOUPLE_SPH 2, WIND_USE_HALO 4, WIND_FIXED_EFFICIENCY 8); the subgrid model (bit 1) does nothing here, as there.
 * `tree` is the gas tree; the star view gives STARP.VDisp (float).  `kicks` (may be NULL) receives the sorted candidate list. */
typedef struct shq_wind_params {
    double BoxSize, Time;
    double WindFreeTravelLength, MaxWindFreeTravelTime, WindEfficiency, WindSpeed, WindSigma0, WindSpeedFactor, MinWindVelocity, WindThermalFactor;
    int WindModel, pad_;
} shq_wind_params;
typedef struct shq_wind_kick {      /* struct StarKick, winds.cpp:175-207 */
    int32_t part_index, pad_;
    double StarDistance;
    uint64_t StarID;
    double StarKickVelocity, StarTherm;
} shq_wind_kick;
typedef struct shq_star_view {
    void *base;
    size_t elsize;
    int64_t numslots;
    size_t off_vdisp;               /* float */
} shq_star_view;
int shq_winds_and_feedback(shq_context *ctx, const shq_tree_view *tree, const shq_part_view *parts, const shq_sph_view *sph, const shq_star_view *stars,
                           const uint64_t *ids, const int32_t *NewStars, int64_t NumNewStars, const shq_wind_params *params, const double *rnd_table,
                           int64_t rnd_size, double *TotalWeight, shq_wind_kick *kicks, int64_t kicks_capacity, int64_t *nkicks, int64_t *nkicked);
/* The two halves of the call above for a multi-rank driver (shenqi_amd/dist.py:DistWinds): the walks alone — TotalWeight and the sorted
 * candidate list, nothing applied; a candidate's particle may be an imported ghost — and the resolution + kicks for a candidate list
 * gathered from all ranks on the particles' owner (any order: it is sorted again). */
int shq_winds_candidates(shq_context *ctx, const shq_tree_view *tree, const shq_part_view *parts, const shq_sph_view *sph, const shq_star_view *stars,
                         const uint64_t *ids, const int32_t *NewStars, int64_t NumNewStars, const shq_wind_params *params, const double *rnd_table,
                         int64_t rnd_size, double *TotalWeight, shq_wind_kick *kicks, int64_t kicks_capacity, int64_t *nkicks);
int shq_winds_apply(shq_context *ctx, const shq_part_view *parts, const shq_sph_view *sph, const uint64_t *ids, const shq_wind_kick *kicks, int64_t nk,
                    const shq_wind_params *params, const double *rnd_table, int64_t rnd_size, int64_t *nkicked);

/* The wind model's two particle loops, on the caller's arrays (gas particles of `list`; NULL: all particles, non-gas skipped):
 *   shq_winds_evolve    winds_evolve (winds.cpp:370-387) as cooling_and_starformation calls it per star-forming gas particle: a wind
 *                       particle recouples when its physical density has dropped below WindFreeTravelDensThresh, otherwise its
 *                       DelayTime (capped at MaxWindFreeTravelTime) shrinks by the particle's hydro step dloga_for_bin / hubble.
 *   shq_winds_subgrid   winds_subgrid + winds_make_after_sf (:272-292, 567-585), the subgrid model (WindModel bit 1; a no-op without
 *                       it): gas particle list[k] with stellar mass StellarMasses[k] formed this step is kicked (wind_do_kick) when
 *                       Table[(ID + 2) % size] < 1 - exp(-windeff sm / Mass), with get_wind_params on its SphP.VDisp.  *nkicked counts them.
 * StellarMasses is indexed like the list here (the reference indexes it by slot). */
int shq_winds_evolve(shq_context *ctx, const shq_part_view *parts, const shq_sph_view *sph, const int32_t *list, int64_t nlist, double a3inv, double hubble,
                     double WindFreeTravelDensThresh, double MaxWindFreeTravelTime, const shq_kick_factors *kf);
int shq_winds_subgrid(shq_context *ctx, const shq_part_view *parts, const shq_sph_view *sph, size_t sph_off_vdisp, const uint64_t *ids, const int32_t *list,
                      int64_t nlist, const double *StellarMasses, const shq_wind_params *params, const double *rnd_table, int64_t rnd_size, int64_t *nkicked);

/* The treewalk of metal_return() (libgadget/metal_return.cpp:513-530, 573-667), after stellar_density
 * (shq_stellar_density) and with the per-star yields of metal_return_copy (:540-571: the IMF / yield-table integrals stay with the
 * caller): every gas particle inside the kernel of a star of `queue` (r2 > 0, r2 < H^2) receives
 * returnfraction = wk (Mass / Density) / StarVolumeSPH of the star's MassGenerated, MetalGenerated and MetalSpeciesGenerated, unless that
 * would lift it above MaxGasMass: Metals[] (float), Metallicity, Mass (float) and Density are updated with the reference's expressions
 * (:622-660).  The reference serialises the updates of a particle with a spin lock, in whatever order its threads arrive; here the
 * stars reach a particle in queue order (a triple list sorted by particle and queue position, applied by one thread per particle).
 * MassReturn[k] = the mass star queue[k] gave away (metal_return_reduce); the star's own bookkeeping (metal_return_postprocess:
 * Mass -= MassReturn, TotalMassReturned, LastEnrichmentMyr) is three assignments the caller keeps.  All per-star arrays are indexed by
 * queue position. */
typedef struct shq_gas_metal_view {
    void *base;
    size_t elsize;
    int64_t numslots;
    size_t off_density, off_metallicity, off_metals;   /* double, double, float[nmetals] */
    int nmetals, pad_;                                 /* NMETALS = 9 */
} shq_gas_metal_view;
int shq_metal_return(shq_context *ctx, const shq_tree_view *tree, const shq_part_view *parts, const shq_gas_metal_view *gas, const int32_t *queue, int64_t nqueue,
                     const double *StarVolumeSPH, const double *MassGenerated, const double *MetalGenerated, const double *MetalSpeciesGenerated /* [nqueue][nmetals] */,
                     double MaxGasMass, int SPHWeighting, int DensityKernelType, double *MassReturn, int64_t *npairs);

/* ---- long-range PM --------------------------------------------------------------------- */

/* Mirror of the PetaPM fields gravpm.cpp reads (libgadget/petapm.h:87-112). */
typedef struct shq_pm_params {
    int32_t Nmesh;
    int32_t pad_;
    double BoxSize;
    double Asmth;
    double G;
} shq_pm_params;

/* One-shot replacement of gravpm_force()'s compute (gravpm.cpp:60-119 minus tree/regions/P(k)):
 * zero GravPM, CIC deposit, r2c, potential_transfer, c2r, readout.  Writes
 * gravpm[numpart][3] and ADDS the PM potential into potential[numpart] (readout_potential,
 * gravpm.cpp:489-491) when potential != NULL.  Particles with the Swallowed flag are skipped
 * (RegionInd = -2, gravpm.cpp:176-178). */
int shq_pm_force(shq_context *ctx, const shq_pm_params *pm, const shq_part_view *parts,
                 double (*gravpm)[3], double *potential);
/* Resident: uses the uploaded particles. */
int shq_pm_run(shq_context *ctx, const shq_pm_params *pm);
int shq_pm_download(shq_context *ctx, double (*gravpm)[3], double *pm_potential);
/* The force part of a PM step as ONE call: gravpm_force, then grav_short_tree for every particle, in the reference's order
 * (libgadget/run.cpp:518-563 — the PM first, because the walk's opening criterion uses FullTreeGravAccel of the last step + the
 * NEW GravPM, gravshort2.hpp:111-121).  Equivalent to shq_pm_run; shq_grav_refresh_oldacc; shq_grav_short_run(all particles,
 * walk_mode = SHQ_WALK_EXACT, optionally | SHQ_WALK_TREE_ORDER) and bit-identical to that sequence.  By default the PM's readout kernel
 * forms OldAcc as it stores GravPM (one pass over the particles instead of two).  With SHQ_TREEPM_FUSE=1 or shq_treepm_set_fuse(ctx, 1),
 * when the walk at hand can carry it (relative criterion, no diagnostic counters, every resident particle a target, mesh below 2^32
 * bytes), the CIC readout and the OldAcc refresh are done by the walk's tasks for their own 64 targets before they walk; that was the
 * default in round 3 and is off since round 4: beside the pair kernel the prologue's loads cost the walk more (2.0 ms) than the two
 * kernels they replace (1.6 ms).  Results land where the separate calls leave them (shq_pm_download, shq_grav_short_download, the
 * resident arrays the kicks read) with the same bits on either route; shq_treepm_last_fused reports which one the last call took. */
int shq_treepm_step(shq_context *ctx, const shq_pm_params *pm, const shq_grav_params *params, int update_potential, int walk_mode);
int shq_treepm_set_fuse(shq_context *ctx, int enable);
int shq_treepm_last_fused(shq_context *ctx, int *fused);
/* gravpm_force started early, for a resident step: the PM of the CURRENT positions (deposit, transforms, readout) is queued on the
 * library's second stream behind everything queued so far, and the call returns; the caller goes on with what needs the positions but
 * not the PM - shq_tree_build - and then calls shq_treepm_step with the same Nmesh, which joins this PM instead of running its own
 * (same results bit for bit).  G > 0: the readout forms OldAcc = |FullTreeGravAccel + GravPM| / G as shq_treepm_step's does.  The
 * reference has no order between force_tree_full and gravpm_force either (run.cpp:476-538; the PM uses no tree).  Every other consumer
 * of PM results (shq_pm_download, shq_kick_pm, ...) joins it too; shq_drift and a particle upload discard it. */
int shq_pm_start(shq_context *ctx, const shq_pm_params *pm, double G);
/* Debug / parity taps: copy the mesh after deposit (Nmesh^3 doubles, [x][y][z]) and the
 * potential mesh after c2r. Valid after shq_pm_run with keep_meshes set. */
/* HIP-event durations (ms) of the last shq_pm_run's phases: [0] zero+deposit+convert, [1] r2c,
 * [2] potential transfer, [3] c2r, [4] readout, [5] total. Synchronises. */
int shq_pm_phase_ms(shq_context *ctx, double ms[6]);
/* How the undivided PM runs its five FFT passes (a test / tuning knob; the potential mesh is the same to the bit either way): 1 (default;
 * SHQ_FFT_TRANSPOSED) the mesh changes layout from pass to pass between the mesh and a scratch mesh of the same size, so that one side
 * of every pass moves contiguous 48 KB tiles; 0 in place, column tiles of 64-byte pieces on both sides. */
int shq_pm_set_fft_transposed(shq_context *ctx, int enable);

/* Power spectrum of the PM density, the side product of potential_transfer (measure_power_spectrum /
 * powerspectrum_add_mode, libgadget/gravpm.cpp:323-376, :430).  After shq_pm_measure_power(ctx, 1) every PM
 * run (shq_pm_run / shq_pm_force) also accumulates, per logarithmic k bin (size = Nmesh bins, bin =
 * floor((size-1) / log(sqrt(3) Nmesh / 2) * log(k2) / 2)): power[i] += w |delta_k|^2 / (sinc^2 ...)^2,
 * kk[i] += w |k| (mesh units), nmodes[i] += w, and norm = |delta_0|^2 — the sums pm->ps holds before
 * powerspectrum_sum (powerspectrum.cpp:53-88), which the caller applies (MPI reduction, normalisation, units).
 * Costs one extra pass pair per PM run (the fused X pass never materialises the spectrum), so it is off
 * by default.  Single-GPU PM only. */
int shq_pm_measure_power(shq_context *ctx, int enable);
int shq_pm_download_power(shq_context *ctx, int size, double *kk, double *power, int64_t *nmodes, double *norm);
int shq_pm_set_debug(shq_context *ctx, int keep_meshes);
/* The deposit mesh of the NEXT shq_pm_run is cleared in the shadow of the tree walk: the first production-size shq_grav_short_run
 * (exact walk, no diagnostic counters, a task's share <= 64 KB) after a PM run writes the zeros from inside the walk kernel — 3.7 GB
 * at Nmesh 768, spread over the walk's 34 ms on a memory system the walk leaves idle — and that shq_pm_run skips its 0.63 ms
 * clearing kernel (petapm.cpp:1304-1310 deposits into a zeroed mesh either way: results are bit-identical).  The mesh is the
 * library's own buffer and nothing reads it after the readout (shq_pm_set_debug(1) keeps COPIES); any other writer of it
 * (shq_fft_r2c / c2r, a different Nmesh) resets the state.  On by default (SHQ_PM_SCRUB=0 or shq_pm_set_mesh_scrub(ctx, 0) turn it
 * off); shq_pm_mesh_prezeroed reports whether the next shq_pm_run will skip the clearing. */
int shq_pm_set_mesh_scrub(shq_context *ctx, int enable);
int shq_pm_mesh_prezeroed(shq_context *ctx, int *zeroed);
int shq_pm_download_mesh(shq_context *ctx, int which /*0 density,1 potential*/, double *mesh);

/* ---- slab-sharded PM for multi-GPU runs (one rank per GPU, x-slabs of the mesh) ------------------
 * The mesh is split into slabs of x-planes exactly as petapm splits its real-space pencils over
 * np0 (petapm.cpp:243-257, here np1 = 1).  These are the LOCAL phases on device buffers the
 * caller owns (e.g. torch tensors used for the RCCL exchanges); the exchanges themselves - ghost
 * planes to the neighbours, the all-to-all transposes of the 2-D/1-D FFT stages - are done by the
 * caller with torch.distributed.  See shenqi_amd/dist.py.
 *   deposit : d_mesh_i64 is int64 [nplanes + 1][Nmesh][Nmesh + 2] (last plane = ghost of the right
 *             neighbour; [Nmesh] planes and no ghost when nplanes == Nmesh); zeroed then filled with
 *             the fixed-point CIC deposit of the uploaded particles, all of which must lie in the slab.
 *   green   : potential_transfer on a y-slab of the transposed half spectrum, complex128
 *             [nyl][Nmesh/2 + 1][Nmesh] (x fastest; the reference's Fourier layout, petapm.cpp:258-282).
 *   readout : d_phi_ext is f64 [nplanes + 5][Nmesh][Nmesh + 2], planes plane0-2 .. plane0+nplanes+2
 *             of the potential (periodic); fills the device-resident GravPM / PM potential. */
/* Particle set already in HBM: d_posm = double[n][4] rows (x, y, z, m); the first nlocal rows are this
 * rank's own particles (PM deposit/readout and the default walk targets), the rest imported ghosts
 * that only act as sources in the tree; nlocal == 0 is a rank that owns nothing (no targets, no deposit, no readout).
 * Previous-step accelerations are kept when n is unchanged.  The tree of the previous set is dropped — a walk before the next
 * shq_tree_build / shq_tree_upload is refused — unless keep_tree != 0, by which the caller states that these are the very
 * positions (same count, same order) the current tree was built from (a force evaluation repeated on frozen positions). */
int shq_particles_set_device(shq_context *ctx, const void *d_posm, int64_t n, int64_t nlocal, int keep_tree);
int shq_pm_slab_deposit(shq_context *ctx, const shq_pm_params *pm, int plane0, int nplanes, void *d_mesh_i64);
int shq_pm_slab_green(shq_context *ctx, const shq_pm_params *pm, int y0, int nyl, void *d_spec);
int shq_pm_slab_readout(shq_context *ctx, const shq_pm_params *pm, int plane0, int nplanes, const void *d_phi_ext);
/* The same phases on the bespoke FFT passes (csrc/fft3d.hip), for mesh sizes that have them
 * (shq_pm_slab_pitch != 0): replaces the 2-D r2c / 1-D FFT / c2r of the slab pipeline (heffte's role in
 * petapm.cpp:281-303) and the separate convert, transpose-to-x-fastest and Green sweeps.  One buffer
 * [nalloc][Nmesh][zp] doubles (zp = shq_pm_slab_pitch, in doubles) is in turn the int64 deposit mesh, the
 * (y, z) half spectrum (complex pitch zp / 2) and the potential; the slab's first own plane is buffer plane
 * `xoff` (2 ghost planes in front, 1 + 3 behind; xoff = 0 and nalloc = Nmesh for a single rank).
 *   slab2_deposit : zero + fixed-point CIC deposit; the plane behind the slab receives the right ghost.
 *   slab2_fft_yz  : direction 0: int64 planes -> half spectrum in (y, z) (in place, `nplanes` planes starting
 *                   at d_planes); direction 1: back to real space.  Unscaled both ways.
 *   slab2_xgreen  : on the y-slab [Nmesh][nyl][zp / 2] the all-to-all delivers (x slowest: the received
 *                   blocks are used as they are), x forward, potential_transfer, x inverse in ONE pass.
 *   slab2_readout : as readout, pitch zp, first own plane at `xoff` of `nalloc` planes. */
int shq_pm_slab_pitch(int Nmesh);
int shq_pm_slab2_deposit(shq_context *ctx, const shq_pm_params *pm, int plane0, int nplanes, int xoff, int nalloc,
                         void *d_mesh_i64);
/* slab2_deposit with `nghost` (1 or 2) planes behind the slab open to the deposit: 2 when the slab also holds the particles of part
 * of its right-hand neighbour's first plane (slabs cut below the plane so that a plane through a cluster's core can be shared:
 * the reference balances at top-leaf granularity, domain.cpp:620-700); slab2_readout then wants 2 + 4 ghost planes (nalloc =
 * nplanes + 6), which its (xoff, nalloc) arguments already express. */
int shq_pm_slab2_deposit_ghosts(shq_context *ctx, const shq_pm_params *pm, int plane0, int nplanes, int xoff, int nalloc, int nghost,
                                void *d_mesh_i64);
int shq_pm_slab2_fft_yz(shq_context *ctx, int Nmesh, void *d_planes, int nplanes, int direction);
/* the same with the pack / unpack of the mesh transposes fused into the Y pass: direction 0 stores the (y, z) spectrum of the planes
 * into d_packed = [nranks][nplanes][Nmesh / nranks][pitch / 2] complex (rows [destination rank][x plane]: the send buffer of the
 * all-to-all), direction 1 starts from d_packed in that layout (what the return all-to-all delivers: rows [source rank][x plane]) */
int shq_pm_slab2_fft_yz_packed(shq_context *ctx, int Nmesh, void *d_planes, int nplanes, int direction, void *d_packed, int nranks);
int shq_pm_slab2_xgreen(shq_context *ctx, const shq_pm_params *pm, void *d_spec, int y0, int nyl);
int shq_pm_slab2_readout(shq_context *ctx, const shq_pm_params *pm, int plane0, int nplanes, int xoff, int nalloc,
                         const void *d_phi);
/* Fixed-point deposit scale 2^e: chosen per context from the local mass sum at particle upload;
 * ranks of one job must agree on it (set it from the global mass sum) so meshes add exactly; e = -1 returns to the
 * per-context choice at the next particle upload. */
int shq_pm_get_deposit_log2scale(shq_context *ctx);
int shq_pm_set_deposit_log2scale(shq_context *ctx, int e);

/* Drop-ins for petapm_fft_r2c / petapm_fft_c2r (libgadget/petapm.cpp:49-71) on one rank: unscaled 3-D transforms of an
 * Nmesh^3 real array ([x][y][z], z fastest: real_space_region, petapm.cpp:256-260) to / from its half spectrum in the
 * reference's Fourier layout [y][z'][x], x fastest, z' <= Nmesh / 2 (fourier_space_region, petapm.cpp:262-270: what
 * pm_apply_transfer_function, :1258-1298, enumerates).  Host pointers.  The _xyz pair keeps the spectrum as [x][y][z'].
 * ONE RANK ONLY (NTask == 1, the 1 x 1 rank grid of petapm.cpp:220-226): for NTask > 1 the reference's 2-D np0 x np1 pencil
 * layouts (petapm.cpp:217-282) are neither produced nor consumed by this library; a multi-rank run replaces petapm_force as a
 * whole with the x-slab pipeline (shq_pm_slab2_* + the caller's all-to-all, INTEGRATION.md "PM on several ranks") — an
 * MI355X-first choice for 8 GPUs on point-to-point xGMI (one transpose pair per PM step instead of ten), not a drop-in for
 * petapm_fft_r2c / c2r there.  The other petapm clients (plane.cpp:326-341, uvbg.cpp:575) get these two calls and nothing more. */
int shq_fft_r2c(shq_context *ctx, int Nmesh, const double *real, double *complx);
int shq_fft_c2r(shq_context *ctx, int Nmesh, const double *complx, double *real);
/* The transfer functions of the other petapm clients: pm_apply_transfer_function (petapm.cpp:1258-1298) followed by
 * petapm_fft_c2r, for one function per call.  Every transfer function of libgenic/zeldovich.cpp:271-321 (density, disp_x/y/z,
 * vel_x/y/z), the lensing planes' neutrino correction (libgadget/plane.cpp:283-304) and the gravity PM's force_x/y/z (gravpm.cpp:464-488)
 * multiplies a mode by  T(k2) x { 1 | i kpos[axis] | i diff_kernel(kpos[axis] 2 pi / Nmesh) }  with T a function of the INTEGER
 * k2 = kx^2 + ky^2 + kz^2 (through |k| = sqrt(k2) 2 pi / BoxSize): the caller tabulates T by k2 with its own functions (DeltaSpec,
 * dlogGrowth, its spline ...), which keeps the values the reference's.  table: host, 3 (Nmesh/2)^2 + 1 entries.
 * zero_mode: what happens to k2 = 0 - 0 the mode is left as it is (the `if(k2)` of the zeldovich transfers), 1 it is set to zero
 * (plane.cpp:286), 2 it is multiplied by T[0] like every other mode (uvbg.cpp:211-215 divide_by_ncell; the reionisation filters
 * filter_pm, uvbg.cpp:218-250, are SHQ_TF_RADIAL tables in k R as well).  complx: [y][z'][x] as shq_fft_r2c returns it; real: [x][y][z], unscaled.  Host pointers; synchronous. */
#define SHQ_TF_RADIAL 0
#define SHQ_TF_GRADIENT 1
#define SHQ_TF_DIFF 2
typedef struct shq_pm_transfer {
    int32_t kind, axis, zero_mode, pad_;
    const double *table;
} shq_pm_transfer;
int shq_pm_apply(shq_context *ctx, int Nmesh, const double *complx, const shq_pm_transfer *tf, double *real);
int shq_fft_r2c_xyz(shq_context *ctx, int Nmesh, const double *real, double *complx);
int shq_fft_c2r_xyz(shq_context *ctx, int Nmesh, const double *complx, double *real);

#ifdef __cplusplus
}
#endif
#endif /* SHENQI_HIP_H */
