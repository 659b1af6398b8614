/* common.hpp — context, error plumbing and device buffers of libshenqi_hip (gfx950 only). */
#ifndef SHQ_COMMON_HPP
#define SHQ_COMMON_HPP

#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>
#include "../../include/shenqi_hip.h"

void shq_set_error(const char *fmt, ...);

#define SHQ_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if(e_ != hipSuccess) {                                                                 \
            shq_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return (e_ == hipErrorOutOfMemory) ? SHQ_ERR_NOMEM : SHQ_ERR_DEVICE;               \
        }                                                                                      \
    } while(0)

#define SHQ_CHECK(cond, code, ...)                                                             \
    do {                                                                                       \
        if(!(cond)) {                                                                          \
            shq_set_error(__VA_ARGS__);                                                        \
            return (code);                                                                     \
        }                                                                                      \
    } while(0)

#define SHQ_TRY(call)                                                                          \
    do {                                                                                       \
        int rc_ = (call);                                                                      \
        if(rc_ != SHQ_OK)                                                                      \
            return rc_;                                                                        \
    } while(0)

/* Grow-only device buffer from the library's own pool (never the caller's arena). */
template <typename T> struct DevBuf {
    T *ptr = nullptr;
    size_t cap = 0; /* elements */
    int reserve(size_t n)
    {
        if(n <= cap)
            return SHQ_OK;
        if(ptr)
            (void) hipFree(ptr);
        ptr = nullptr;
        cap = 0;
        hipError_t e = hipMalloc((void **) &ptr, n * sizeof(T));
        if(e != hipSuccess) {
            shq_set_error("hipMalloc of %zu bytes failed: %s", n * sizeof(T), hipGetErrorString(e));
            ptr = nullptr;
            return SHQ_ERR_NOMEM;
        }
        cap = n;
        return SHQ_OK;
    }
    void release()
    {
        if(ptr)
            (void) hipFree(ptr);
        ptr = nullptr;
        cap = 0;
    }
};

/* Pinned host staging buffer. */
template <typename T> struct PinBuf {
    T *ptr = nullptr;
    size_t cap = 0;
    int reserve(size_t n)
    {
        if(n <= cap)
            return SHQ_OK;
        if(ptr)
            (void) hipHostFree(ptr);
        ptr = nullptr;
        cap = 0;
        hipError_t e = hipHostMalloc((void **) &ptr, n * sizeof(T), hipHostMallocDefault);
        if(e != hipSuccess) {
            shq_set_error("hipHostMalloc of %zu bytes failed: %s", n * sizeof(T), hipGetErrorString(e));
            ptr = nullptr;
            return SHQ_ERR_NOMEM;
        }
        cap = n;
        return SHQ_OK;
    }
    void release()
    {
        if(ptr)
            (void) hipHostFree(ptr);
        ptr = nullptr;
        cap = 0;
    }
};

/* Walk-time node pool record split in three 16/32-byte streams (wave-uniform scalar loads). */
struct NodeA { double cofm[3]; double mass; };   /* 32 B */
struct NodeB { double center[3]; double len; };  /* 32 B */
struct NodeC { int32_t sibling; int32_t child; int32_t type; int32_t count; }; /* 16 B: child = first
    child node (NODE) or first leaf-order particle slot (PARTICLE); count = noccupied for leaves */
struct NodeH { double hmax; };                    /* SPH only */
/* Gravity walk record: everything one node test needs in one 128-byte aligned line, including the
 * per-node products the opening tests use (computed once at pack time instead of per visit). */
struct alignas(128) NodeG {
    double cofm[3], mass;
    double center[3], len;
    int32_t sibling, child, type, count;
    double bhlim;    /* len * len / BHOpeningAngle^2 of the current walk parameters: the Barnes-Hut test is r2 < bhlim */
    double mlen2;    /* mass * len * len */
    double inside;   /* 0.6 * len */
    double rcut2;    /* Rcut^2 of the current walk parameters (the same in every record): arrives with the node instead of
                        occupying a scalar register pair for the whole walk, which under the 96-SGPR cap the compiler re-read
                        from the kernel arguments at every visit */
    double wraplim;  /* Box / 2 - len / 2: while every |center - pos| stays below it, neither the centre nor the
                        centre of mass (inside the cell) needs the periodic wrap */
    double rcuthl;   /* Rcut + len / 2 of the discard test, filled for the Rcut of the current walk parameters */
};
static_assert(sizeof(NodeG) == 128, "NodeG must be one 128-byte line");

/* One TopLevel node of the host tree for the export-detection walk (toptree.hip) */
struct alignas(32) TopNodeG {
    double cofm[3], mass;
    double center[3], len;
    double hmax;
    int32_t sibling, child; /* indices into the top-node array (pre-order), -1 = end */
    int32_t kind;           /* 0 internal top-level node, 1 local top-level leaf, 2 pseudo node */
    int32_t leaf;           /* pseudo: index into TopLeaves */
    double pad;
};
static_assert(sizeof(TopNodeG) == 96, "TopNodeG layout");

struct GravStatsDev {
    unsigned long long ninteractions;
    unsigned long long nvisited;
    unsigned long long nwave_applies;
    unsigned long long nwave_node_applies;
    unsigned long long nnode_interactions;
    long long min_int;
    long long max_int;
    /* SHQ_WALK_STATS=2 diagnostics: rounds by number of participating lanes (8 buckets of 8 lanes) */
    unsigned long long hist_visit[8], hist_node[8], hist_leaf[8];
    unsigned long long lonely[4]; /* per wave: sum / max over lanes of interactions met in rounds of <= 8, <= 16 lanes */
};

/* Wave-wide vote as a 64-bit lane mask.  HIP's __ballot(int) compares its argument with zero, so a boolean
 * predicate is first materialised (v_cndmask) and then compared again (v_cmp_ne): two VALU instructions per
 * vote, a third of the walk's per-node instruction count.  The builtin takes the predicate as it is. */
__device__ __forceinline__ unsigned long long shq_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

/* XCD-aware block remap (bijective for any grid size): workgroups are dealt round-robin over the
 * 8 XCDs, so block b runs on XCD b % 8.  Give every XCD one contiguous eighth of the work so
 * that spatially adjacent target groups share an L2 (4 MiB per XCD, not coherent across XCDs).
 * A performance hint only: correctness never depends on placement. */
__device__ __forceinline__ unsigned xcd_block(unsigned b, unsigned nb, unsigned K = 16)
{
    if(K == 0)
        return b;
    /* Chunked: within every run of 8*K blocks, XCD x takes K consecutive ones.  (One contiguous
     * eighth per XCD has better locality but piles a clustered region onto a single XCD: measured
     * 77 ms vs 59 ms round-robin on the S-cluster 256^3 walk.) */
    const unsigned super = b / (8u * K);
    if((super + 1u) * 8u * K > nb)
        return b; /* ragged tail: identity */
    const unsigned within = b - super * 8u * K;
    return super * 8u * K + (within & 7u) * K + (within >> 3);
}

#define SHQ_NTIMERS 20 /* 0-7 caller, 8-13 PM phases, 14 SPH, 16 tree build, last: walk */

/* scratch of the device tree build (tree_build.hip) */
struct TreeBuildBufs {
    DevBuf<unsigned long long> keys[2], okeys[2], packed[2], counters;
    DevBuf<int> state;                   /* TbState (tree_build.hip): the level loop's bookkeeping */
    DevBuf<int32_t> idx[2], order[2], rank, frontier[2], bounds;
    DevBuf<int32_t> lo, hi, parent, sibling, firstchild, nchild, level;
    DevBuf<int32_t> top;                 /* domain build: TopNodes index of a top-level node, -1 below the top tree */
    DevBuf<unsigned long long> path;     /* domain build: octant digits from the root, left-aligned (pre-order sort key) */
    DevBuf<int4> geo_child[2];           /* domain build: daughters 0-3 / 4-7 per TopNode */
    DevBuf<int2> geo_kind;               /* (kind: 0 internal, 1 leaf of this task, 2 pseudo; TopLeaves index) */
    DevBuf<double> topbuf;               /* gather / scatter of the top-level nodes' records */
    DevBuf<double4> cen, mom;
    DevBuf<double> hmax;
    DevBuf<char> temp;
    DevBuf<shq_node> exportbuf;
    void release()
    {
        for(int i = 0; i < 2; i++) {
            keys[i].release(); okeys[i].release(); packed[i].release(); idx[i].release(); order[i].release(); frontier[i].release();
        }
        counters.release(); rank.release(); bounds.release(); lo.release(); hi.release(); parent.release(); sibling.release();
        firstchild.release(); nchild.release(); level.release(); cen.release(); mom.release(); hmax.release(); temp.release();
        exportbuf.release(); top.release(); path.release(); geo_child[0].release(); geo_child[1].release(); geo_kind.release(); topbuf.release(); state.release();
    }
};

struct shq_context {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    /* SHQ_PM_OVERLAP=1: shq_pm_run works on its own (high-priority) stream so that the bandwidth-bound PM
     * could overlap the VALU-bound walk; every entry point that reads PM results or changes PM inputs joins
     * it first (shq_join_pm).  Off by default: measured, nothing overlaps — the walk's 7 waves x 72 VGPRs
     * per SIMD are refilled by the next walk workgroup the moment one retires, so a PM workgroup (200+ VGPRs
     * per wave, 61 KB LDS) never finds room until the walk has drained (its first kernel waits 49 ms). */
    /* SHQ_WALK_FREE_CUS = k: the pieces of shq_grav_short_run_range run on a stream masked to leave k compute units of every
     * XCD free, for the collective the caller runs beside them.  Off by default: measured with k = 2 on a one-rank RCCL group,
     * the walk pieces got 10 % slower and the RCCL round queued beside them still ended only when the walk drained */
    hipStream_t stream_walk = nullptr;
    hipEvent_t ev_walk_in = nullptr, ev_walk_out = nullptr;
    hipStream_t stream_pm = nullptr;
    hipEvent_t ev_pm_ready = nullptr, ev_pm_done = nullptr;
    /* the pair kernel of the sparse subtrees runs beside the main walk on this stream (lowest priority: when both have workgroups
     * pending, the main walk's go first) */
    hipStream_t stream_pair = nullptr;
    hipEvent_t ev_pair_fork = nullptr, ev_pair_join = nullptr;
    /* shq_set_inputs_current: which of the context's copies the caller vouches for, and what they are copies of */
    int inputs_current = 0;
    const void *cur_parts = nullptr, *cur_sph = nullptr, *cur_tree = nullptr, *cur_ids = nullptr;
    int64_t cur_parts_n = -1, cur_sph_n = -1, cur_tree_n = -1, cur_tree_first = -1, cur_ids_n = -1;
    hipEvent_t ev_sph[4] = {nullptr, nullptr, nullptr, nullptr}; /* sph.hip, launch_two_kernel: walk done / evaluation done, two list regions */
    bool pm_pending = false;
    bool pm_prestarted = false;          /* shq_pm_start: the PM of the current positions is queued on stream_pm; shq_treepm_step takes it over */
    bool pm_prestarted_oldacc = false;   /* ... and its readout kernel formed OldAcc */
    int pm_prestarted_nmesh = 0;
    bool pm_overlap = false;
    bool treepm_fuse = false;  /* SHQ_TREEPM_FUSE / shq_treepm_set_fuse: shq_treepm_step carries the readout in the walk's task prologue.  Off since
                                  round 4: with the pair kernel beside the main walk the prologue's 104 eight-byte loads per lane cost the walk
                                  2.0 ms (28.5 -> 30.5) where the readout kernel with its 28 sixteen-byte loads takes 1.44 (same-box A/B) */
    double readout_oldacc_G = 0; /* > 0: the PM's readout kernel also forms OldAcc from FullTreeGravAccel and the new GravPM (shq_treepm_step) */
    bool last_step_fused = false;
    bool fuse_readout = false; /* shq_treepm_step: the next exact walk carries the PM readout + OldAcc refresh in its prologue */
    double fuse_cell = 0, fuse_ffac = 0, fuse_G = 0;
    DevBuf<double> mesh_alt;   /* shq_treepm_step: the walk reads one mesh and clears the other (they change places per step) */
    bool pm_scrub = true;      /* SHQ_PM_SCRUB: the first full tree walk after a PM run clears the PM mesh for the next deposit */
    bool mesh_zeroed = false;  /* ctx->mesh is all zero (set by that walk, reset by pm_prepare) */
    size_t mesh_words = 0;     /* 8-byte words of ctx->mesh in use (pm_prepare) */
    hipEvent_t ev_begin[SHQ_NTIMERS] = {};
    hipEvent_t ev_end[SHQ_NTIMERS] = {};

    /* ---- particle store, by particle index (sorted SoA, Peano order as the host keeps it) */
    int64_t numpart = 0;
    int64_t nlocal = 0;        /* the first nlocal particles are this rank's own (targets, PM deposit / readout); the rest are
                                  imported ghosts.  shq_particles_upload: all of them; shq_particles_set_device: as given, 0 included */
    DevBuf<double4> posm;      /* x,y,z,mass */
    DevBuf<double> oldacc;     /* |FullTreeGravAccel + GravPM| / G */
    DevBuf<double> treeacc;    /* [N][3] FullTreeGravAccel */
    DevBuf<double> gravpm;     /* [N][3] */
    DevBuf<double> pmpot;      /* [N] PM potential contribution */
    DevBuf<double> acc;        /* [N][3] walk output (Accel) */
    DevBuf<double> pot;        /* [N] tree potential */
    DevBuf<int32_t> nint;      /* [N] interactions */
    DevBuf<uint8_t> pflags;    /* bit0 garbage, bit1 swallowed; bits 4-7 type */
    DevBuf<int32_t> active;    /* uploaded active list */
    DevBuf<unsigned long long> top_task; /* resident export detection: entries per destination task */
    DevBuf<int32_t> top_sort;            /* (task, entry) pairs before / after the stable sort */
    DevBuf<long long> top_slot;          /* entry -> slot in the task-ordered send buffer */
    long long top_ntargets = 0, top_nexport = -1;
    DevBuf<TopNodeG> topnodes;
    DevBuf<int2> topleaves;
    DevBuf<int32_t> top_counts;
    DevBuf<shq_data_index> top_table;
    int64_t ntopnodes = 0;
    /* the domain of the last shq_tree_build_domain (host copies: the top tree has a few thousand nodes) */
    bool tb_domain = false;
    std::vector<shq_topnode_geo> dom_geo;
    std::vector<int32_t> dom_kind, dom_rank, dom_leaf_task; /* per TopNode: kind, pre-order number; per leaf: Task */
    std::vector<double> dom_rec;                           /* per TopNode: cofm[3], mass, hmax, center[3], len */
    int dom_thistask = 0;
    bool have_toptree = false;
    bool grav_raw = false;     /* acc / pot hold raw sums of a deferred-postprocess walk */
    DevBuf<shq_grav_result> gq_res;
    DevBuf<int32_t> act_list, act_sub;   /* resident ActiveParticle list and gravity sub-list (shq_build_active_*) */
    DevBuf<unsigned long long> act_counts;
    DevBuf<char> act_temp;
    DevBuf<uint8_t> act_flag;
    int64_t n_act = -1, n_sub = -1;      /* -1: not built */
    bool act_all = false;                /* PM step: the list is NULL, every particle is active */
    DevBuf<GravStatsDev> gstats;
    bool have_parts = false;
    int stats_guard = 0;       /* SHQ_WALK_STATS_GUARD (A/B knob): 1 reads before atomicMin / atomicMax (slower), 2 drops the wave tallies */
    bool allow_padding = false; /* SHQ_WALK_PADDING=1: -1 entries of a gravity target list are idle lanes (tools/walk_cell_probe.py) */

    /* ---- node pool */
    int64_t numnodes = 0;
    int64_t firstnode = 0;
    int32_t root = 0;          /* packed index of the root */
    int64_t ntreeparts = 0;    /* particles referenced by leaves */
    DevBuf<NodeA> nodeA;
    DevBuf<NodeB> nodeB;
    DevBuf<NodeC> nodeC;
    DevBuf<NodeG> nodeG;       /* merged record for the gravity walk */
    DevBuf<double4> posm_leaf; /* leaf-ordered copy of (x,y,z,m) */
    DevBuf<int32_t> leaf_pidx; /* leaf slot -> particle index */
    bool have_tree = false;
    /* source-parallel walk (grav_group.hip): per-tree children lists, rebuilt when have_group_aux is false; the pool of
     * interaction-list chunks and its bookkeeping */
    DevBuf<float4> nodeF;      /* 64-byte T1 records: f32 (cofm, mass), (centre, len) + children / leaf slots */
    bool have_group_aux = false;
    DevBuf<int> walk_counters;
    DevBuf<int32_t> walk_pool_idx, walk_chunk_cnt, walk_chunk_next, walk_group_head;
    DevBuf<uint8_t> walk_pool_msk;
    int64_t walk_pool_chunks = 0;
    double walk_chunks_per_target = 0; /* measured in the last batch: sizes the next one */
    double node_rcut = -1;     /* Rcut / BHOpeningAngle2 the pool's rcuthl and bhlim fields were filled for (< 0: stale) */
    double node_bh2 = -1;
    double treeBox = 0;
    DevBuf<double> node_hmax;  /* mom.hmax per packed node (SPH symmetric cull) */
    DevBuf<int32_t> pfather;   /* particle -> packed index of the leaf holding it, or -1 */
    std::vector<int32_t> node_order; /* packed index -> index into the caller's nodes_base */
    std::vector<int32_t> node_rank;  /* caller's node index -> packed index (-1: unreachable); empty = identity */
    bool have_father = false;
    TreeBuildBufs tb;
    DevBuf<int32_t> tree_targets; /* own particles in leaf order (SHQ_WALK_TREE_ORDER) */
    int64_t ntree_targets = 0;
    bool have_tree_targets = false;
    int tree_targets_refresh = 8;  /* SHQ_TREE_TARGETS_REFRESH: rebuilds over the same particle set that keep the list (1: a new list per build) */
    int tree_targets_age = 0, tree_targets_mask = -2;
    long long tree_targets_np = -1, tree_targets_ntree = -1;
    DevBuf<long long> hilb_iota;  /* shq_hilbert_order: 0 .. n-1, the values of its sort */
    /* black-hole accretion / feedback walks: per-call uploads (sph_capi.hip) */
    DevBuf<int32_t> bhw_bhp, bhw_queue;
    DevBuf<char> bhw_rec;
    DevBuf<unsigned long long> bhw_ids, bhw_sphsw, bhw_bhsw, bhw_swid;
    DevBuf<double> bhw_rnd, bhw_out;
    DevBuf<uint8_t> bhw_eeqos, bhw_heated, bhw_touched;
    DevBuf<int32_t> bhw_tlist;
    DevBuf<double> bhw_trows;
    DevBuf<unsigned long long> metal_keys[2];
    DevBuf<double> metal_val[2], metal_star, metal_gd;
    DevBuf<float> metal_gf;
    DevBuf<char> wind_kicks;
    DevBuf<double> wind_d;
    DevBuf<unsigned long long> wind_cnt;
    bool tb_built = false;     /* the current tree came from shq_tree_build (downloadable) */

    /* ---- SPH state, by particle index (gas fields gathered from their slots at upload) */
    bool have_sph = false;
    bool have_dyn = false;     /* Vel / Hsml / DtHsml / TimeBinGravity resident (shq_dynamics_upload) */
    DevBuf<double> hsml, dthsml, vel;
    DevBuf<uint8_t> bin_grav, bin_hydro;
    DevBuf<double> g_entropy, g_dtentropy, g_hydroaccel, g_delaytime;
    DevBuf<double> g_density, g_egywt, g_dhsmlegy, g_divvel, g_curlvel;
    DevBuf<double> g_hydroaccel_out, g_dtentropy_out, g_maxsignalvel;
    bool gas_resident = false; /* the particle set came from shq_gas_set_device (rows of SHQ_GAS_NCOL doubles) */
    DevBuf<int> gas_bad;
    /* particle exchange (exchange.hip) */
    DevBuf<int32_t> ex_list, ex_val[3];
    DevBuf<unsigned int> ex_key[4];
    DevBuf<unsigned long long> ex_counts, ex_u64;
    DevBuf<long long> ex_i64;
    DevBuf<char> ex_bytes;
    std::vector<shq_exchange_entry> ex_togo;
    int64_t ex_last = -1;
    int ex_ntask = 0;
    /* friends-of-friends (fof.hip) */
    DevBuf<int32_t> fof_parent, fof_i32[6], fof_g32[5], fof_partgrnr, fof_members, fof_biglist;
    DevBuf<unsigned long long> fof_u64[4];
    DevBuf<unsigned int> fof_gkey[2];
    DevBuf<long long> fof_goff[2];
    DevBuf<shq_fof_group> fof_groups;
    DevBuf<char> fof_partial;
    int64_t fof_ngroups = -1, fof_nmembers = 0, fof_nruns = 0;
    /* black-hole slot fields of the resident step (timestep.hip): by BH ordinal, bh_pidx ascending particle indices */
    bool have_bh_dyn = false, bh_reposition = false;
    int64_t nbh = 0;
    DevBuf<int32_t> bh_pidx;
    DevBuf<uint8_t> bh_u8;     /* minTimeBin[nbh], TimeBinDynFric[nbh], JumpToMinPot[nbh] */
    DevBuf<double> bh_vec;     /* DFAccel, DF_SurroundingVel, DragAccel, MinPotPos, MinPotVel: [5][nbh][3] */
    DevBuf<double4> velp, hydC, hydD, velp_leaf;
    DevBuf<char> hydrec_leaf;  /* HydRec[] (sph.hip): 128-byte neighbour records for the hydro evaluation */
    DevBuf<double> hsml_leaf;
    DevBuf<int32_t> ngarb_leaf;
    DevBuf<float4> posf_leaf; /* (x, y, z) rounded to f32 and the f32 pre-test bound of the particle's own Hsml: sph.hip, ngb_walk PRE32 */
    DevBuf<int32_t> flag_leaf;
    DevBuf<double> s_numngb, s_dhsmldens, s_left, s_right, s_rot, s_gradrho, s_evp_in;
    DevBuf<int32_t> s_todo, s_queue2, s_queue3, s_blockcount;
    DevBuf<int32_t> s_nlist;   /* per-lane neighbour lists of the SPH walks */
    DevBuf<int32_t> s_queue0;  /* the work queue of the open SPH walk */
    std::vector<int32_t> sph_queue_host;
    DevBuf<int32_t> s_nlist2;  /* ... of the secondary (imported-query) walks, which run while a primary walk is open */
    struct SphRun {            /* a density / hydro walk opened in phases (shq_sph_*_begin ... _end) */
        shq_density_params dp;
        shq_hydro_params hp;
        int want_gradrho = 0;
        const int32_t *cur = nullptr;
        long long size = 0, nq0 = 0;
        int wsel = 0, niter = 0;
        int phase = 0;          /* 0 none, 1 density, 2 hydro */
    } sphrun;
    DevBuf<int32_t> s_ncount, s_redo, s_redo2; /* list lengths; targets whose lists overflowed (a wave each); the heaviest of those (a workgroup each) */
    DevBuf<long long> s_counters;

    /* ---- PM */
    int pm_nmesh = 0;
    hipfftHandle plan_r2c = 0, plan_c2r = 0;
    bool have_plans = false;
    bool pm_custom_fft = false; /* bespoke 5-pass FFT pipeline (fft3d.hip) instead of rocFFT */
    bool fft_transposed = true; /* SHQ_FFT_TRANSPOSED / shq_pm_set_fft_transposed: the undivided PM's five passes change the mesh layout
                                   between the mesh and a scratch mesh (mesh_alt) instead of working in place (fft3d.hip) */
    int pm_zp = 0;             /* z pitch of the mesh in doubles */
    DevBuf<double> fft_tw;     /* twiddles exp(-2 pi i k / N) */
    DevBuf<double> fft_gax;    /* per-axis Green's function factor of the transposing pipeline's X pass (fft3d.hip) */
    int fft_gax_n = 0;
    double fft_gax_asmth2 = -1;
    const double *fft_gax_src = nullptr;
    int fft_tw_n = 0;
    DevBuf<double> mesh;       /* padded in-place real/complex mesh: N*N*(N+2) doubles */
    DevBuf<double> sinctab;    /* 1/sinc^2 per mesh index */
    int sinctab_n = 0;
    DevBuf<int> pm_oob;        /* out-of-slab flag written by the deposit/readout kernels */
    DevBuf<double> dbg_rho, dbg_pot;
    int pm_keep = 0;
    int pm_log2scale = 30;     /* fixed-point deposit scale 2^e, set at particle upload */
    int pm_log2scale_user = -1; /* >=0: forced by the caller (multi-rank consistency) */
    double mass_sum = 0;
    DevBuf<float> gravtab;     /* [2][512] window table */
    bool have_pm_result = false;
    bool pm_measure_power = false; /* shq_pm_measure_power: accumulate P(k) during the next PM runs */
    bool have_power = false;
    int ps_nbins = 0, ps_bintab_n = 0;
    DevBuf<double> ps_sums;    /* [nbins] power, [nbins] kk, [nbins] modes (u64), norm */
    DevBuf<int32_t> ps_bintab; /* bin of every k2 */

    shq_walk_stats last_stats = {};
    int walk_variant = 3;      /* SHQ_WALK_VARIANT: 0 prefetch+leaf4, 1 prefetch+leaf2, 2 leaf4, 3 leaf2 (fastest: no SGPR spills) */
    int walk_stats = 0;        /* SHQ_WALK_STATS / shq_set_walk_stats: wave-level counters (1), + histograms (2); off by default: 4 % */
    int walk_persist = 1;      /* SHQ_WALK_PERSIST: persistent waves taking 64-target tasks from per-XCD counters (0: one task per wave) */
    int walk_ring = 1;         /* SHQ_WALK_RING: leaf particles through the wave-private LDS ring (persistent walk only) */
    int num_cus = 256;         /* compute units of the device */
    DevBuf<unsigned int> walk_tasks; /* the task counters of the persistent walk */
    DevBuf<int4> sp_items, sp_stack;  /* SHQ_WALK_SPARSE: noted subtrees per task, pair stacks of the pair kernel's waves */
    DevBuf<int32_t> sp_count;
    DevBuf<int> sp_flags;             /* SpFlag (grav_walk.hip): [0, 8) per launch, [8, 16) sticky, [16] the pair kernel's task counter */
    PinBuf<int> sp_host;              /* the sticky words as the last completed launch left them */
    int sp_stack_cap = 0;             /* pairs per pair-kernel wave (0: SHQ_SPARSE_STACK); shq_set_walk_debug */
    unsigned sp_spin_max = 0;         /* polls before a live pair wave gives up (0: default) */
    int walk_sparse = 1;              /* SHQ_WALK_SPARSE (0: the main walk enters every subtree itself; 2: pair kernel on full records) */
    DevBuf<int> node_lean_bad;        /* [0] != 0: some record's second half is not reproducible from {mass, len} (fill_rcuthl_kernel) */
    bool node_lean_checked = false;
    int walk_overlap = 1;             /* SHQ_WALK_OVERLAP: the pair kernel beside the main walk (second stream) instead of behind it */
    bool sp_check_pending = false;
    int xcd_k = 32;            /* SHQ_XCD_K: blocks per XCD chunk in the remap (0 = off); 32 measured best (2 %) */
    float last_walk_ms = 0;
    int last_walk_mode = 0;    /* what SHQ_WALK_AUTO resolved to in the last launch */

    /* host staging */
    PinBuf<char> stage;
};

/* capi.hip: make the main stream wait for an outstanding asynchronous PM run */
int shq_join_pm(shq_context *ctx);
/* grav_walk.hip */
/* Device pointer and length of an active list argument of the C-ABI: NULL (all n_all), a host list
 * (uploaded), or one of the SHQ_ACTIVE_RESIDENT / SHQ_SUBLIST_RESIDENT handles. dynamics.hip */
/* ordinal of particle i in the ascending list of black-hole particle indices, -1 if it is not there */
__device__ inline long long shq_bh_ordinal(const int32_t *pidx, long long nbh, int32_t i)
{
    long long lo = 0, hi = nbh;
    while(lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if(pidx[mid] < i)
            lo = mid + 1;
        else
            hi = mid;
    }
    return (lo < nbh && pidx[lo] == i) ? lo : -1;
}
int shq_resolve_active(shq_context *ctx, const int32_t *active, int64_t nactive, int64_t n_all, const int32_t **d_active, int64_t *nt);
/* first: with d_active == NULL the targets are the particles [first, first + ntargets) (the per-particle arrays are handed to
 * the kernels shifted by `first`); a range with first > 0 adds to the interaction statistics of the ranges before it */
int shq_launch_grav_walk(shq_context *ctx, const shq_grav_params *p, const int32_t *d_active,
                         int64_t ntargets, int update_potential, int walk_mode, int64_t first = 0);
int shq_launch_grav_postprocess(shq_context *ctx, const shq_grav_params *p, const int32_t *d_active,
                                int64_t ntargets, int update_potential, int64_t first = 0);
int shq_launch_oldacc(shq_context *ctx, double G);
void shq_launch_stats_init(shq_context *ctx);
void shq_fill_node_walk_params(shq_context *ctx, const shq_grav_params *p);
int shq_walk_reserve_sparse(shq_context *ctx, long long nwaves);
int shq_walk_check_status(shq_context *ctx, bool sync);
int shq_walk_prereserve(shq_context *ctx);
const int *shq_walk_error_word(shq_context *ctx); /* device address of the pair kernel's sticky error word, or null before any sparse launch */
/* grav_group.hip */
int shq_launch_grav_walk_group(shq_context *ctx, const shq_grav_params *p, const int32_t *d_active, int64_t ntargets, int update_potential,
                               int64_t first);
int shq_launch_grav_walk_ghosts(shq_context *ctx, const shq_grav_params *p, const double4 *d_qpos, const double *d_qoldacc,
                                const int32_t *d_qstart, int64_t nq, double *d_acc, double *d_pot, int32_t *d_nint, int update_potential);
/* pm.hip */
int shq_pm_execute(shq_context *ctx, const shq_pm_params *pm, bool readout = true);
int shq_pm_run_on_pm_stream(shq_context *ctx, const shq_pm_params *pm, bool low_priority);
bool shq_walk_can_fuse_readout(shq_context *ctx, const shq_grav_params *p, const int32_t *d_active, int64_t ntargets, int64_t first);
bool shq_walk_can_fuse_readout_pre(shq_context *ctx, const shq_grav_params *p, int64_t ntargets);
void shq_pm_destroy_plans(shq_context *ctx);
int shq_fft_roundtrip_r2c(shq_context *ctx, int N, const double *real, double *complx, bool ref_layout);
int shq_fft_roundtrip_c2r(shq_context *ctx, int N, const double *complx, double *real, bool ref_layout, const shq_pm_transfer *tf = nullptr);
/* tree_build.hip */
int shq_build_tree_targets(shq_context *ctx);
/* fft3d.hip */
bool shq_fft3d_supported(int N);
int shq_fft3d_pitch(int N);
int shq_fft3d_run(shq_context *ctx, double *d_mesh, int N, int zp, int stage, bool from_i64, double inv_scale,
                  const double *d_sinctab, double asmth2, double pot_factor);
int shq_fft3d_run_transposed(shq_context *ctx, double *d_mesh, double *d_scratch, int N, int zp, bool from_i64, double inv_scale,
                             const double *d_sinctab, double asmth2, double pot_factor);
int shq_fft3d_run_slab(shq_context *ctx, double *d_mesh, int N, int zp, int stage, bool from_i64, double inv_scale,
                       const double *d_sinctab, double asmth2, double pot_factor, int nslab, int y0);
int shq_fft3d_run_slab_packed(shq_context *ctx, double *d_mesh, int N, int zp, int stage, bool from_i64, double inv_scale,
                              const double *d_sinctab, double asmth2, double pot_factor, int nslab, int y0, double *d_packed, int nranks);
/* sph.hip */
int shq_sph_prepare(shq_context *ctx, const shq_kick_factors *kf, const shq_hydro_params *hp, const double *d_evp_in);
int shq_sph_density_device(shq_context *ctx, const shq_density_params *p, const int32_t *d_queue, int64_t nq,
                           int want_gradrho, shq_sph_stats *stats);
int shq_sph_hydro_device(shq_context *ctx, const shq_hydro_params *p, const int32_t *d_queue, int64_t nq, shq_sph_stats *stats);
int shq_sph_hydro_begin(shq_context *ctx, const shq_hydro_params *p, const int32_t *d_queue, int64_t nq);
int shq_sph_hydro_primary(shq_context *ctx);
int shq_sph_hydro_post(shq_context *ctx);
int shq_sph_hydro_end(shq_context *ctx, shq_sph_stats *stats);
int shq_sph_hydro_reduce(shq_context *ctx, const int32_t *d_place, const void *d_results, int64_t n);
int shq_sph_hydro_secondary(shq_context *ctx, const shq_hydro_params *p, const double4 *d_qposm, const double *d_qhsml,
                            const double4 *d_qvelp, const double4 *d_qC, const double4 *d_qD, const int4 *d_qseg, int64_t nq, double *d_out,
                            unsigned long long *d_nint);
int shq_sph_fill_queries_device(shq_context *ctx, const shq_data_index *d_table, int64_t n, void *d_out);
int shq_sph_density_begin(shq_context *ctx, const shq_density_params *p, const int32_t *d_queue, int64_t nq, int want_gradrho);
int shq_sph_density_primary(shq_context *ctx);
int shq_sph_density_post(shq_context *ctx, int64_t *nredo);
int shq_sph_density_end(shq_context *ctx, shq_sph_stats *stats);
int shq_sph_density_reduce(shq_context *ctx, const int32_t *d_place, const void *d_results, int64_t n);
int shq_sph_density_secondary(shq_context *ctx, const shq_density_params *p, const double4 *d_qposm, const double *d_qhsml,
                              const double4 *d_qvelp, const uint8_t *d_qflags, const int4 *d_qseg, int64_t nq, double *d_out,
                              unsigned long long *d_nint);
int shq_sph_gradrho_mag(shq_context *ctx, double *d_out);
int shq_bh_dynfric_device(shq_context *ctx, const shq_kick_factors *kf, double BoxSize, int kernel_type, int typemask, int method,
                          const double *d_potential, const int32_t *d_queue, int64_t nq, double *d_out);
int shq_wind_veldisp_device(shq_context *ctx, const shq_kick_factors *kf, double BoxSize, double hubble_a2, const int32_t *d_queue, int64_t nq,
                            double *d_dmradius, double *d_vdisp, shq_sph_stats *stats);
/* black-hole accretion / feedback walks (sph.hip) */
struct BhRec {
    double Mass, Density, Mtrack, DFAccel[3], VDisp, KineticFdbkEnergy, Mdot, FeedbackWeightSum;
    int32_t CountProgs, KEflag;
};

struct BhWalkArgs {
    const int32_t *bhp;        /* particle indices of all black holes, ascending */
    long long nbh;
    BhRec *bh;
    const unsigned long long *ids;
    const double *rnd;
    unsigned long long rndsize;
    const double *vel, *treeacc, *gravpm, *delay;
    double *velw;              /* Vel, written by the kinetic kicks */
    double *entropy;           /* written by the thermal feedback */
    const double *density;
    const uint8_t *bin_grav, *bin_hydro;
    uint8_t *pflags;
    const uint8_t *eeqos;      /* sfreff_on_eeqos per particle, or NULL */
    uint8_t *heated;           /* BHHeated per particle (out) */
    uint8_t *touched;          /* gas particles whose entropy, velocity or flags the feedback walk changed (out; may be NULL) */
    const int32_t *leaf_pidx;
    unsigned long long *sph_swallow; /* by particle index */
    unsigned long long *bh_swallow;  /* by black-hole ordinal */
    unsigned long long *bh_swallowid_out; /* BHP.SwallowID of swallowed holes, by ordinal */
    double *out;
    shq_bh_params P;
    long long Ti_Current;
};

/* wind walks (sph.hip) */
struct WindWalkArgs {
    const unsigned long long *ids;
    const double *rnd;
    unsigned long long rndsize;
    const int32_t *leaf_pidx;
    double *totalweight;           /* by queue entry */
    const double *vdisp;           /* by queue entry */
    unsigned long long *nvisited, *nkicks;
    unsigned long long maxkicks;
    shq_wind_kick *kicks;
    shq_wind_params P;
};
int shq_wind_walk_device(shq_context *ctx, const WindWalkArgs *w, const int32_t *d_queue, int64_t nq, bool kick);
int shq_winds_evolve_device(shq_context *ctx, const int32_t *d_list, int64_t n, double a3inv, double hubble, double DensThresh, double MaxTravelTime,
                            const shq_kick_factors *kf);
int shq_winds_subgrid_device(shq_context *ctx, const WindWalkArgs *w, const int32_t *d_list, int64_t n, const double *d_stellarmass, const double *d_vdisp,
                             unsigned long long *d_nkicked);
int shq_wind_resolve_device(shq_context *ctx, const WindWalkArgs *w, long long nk, shq_wind_kick *d_sorted, unsigned long long *d_napplied, int *d_odd, bool apply);
/* metal return (sph.hip) */
#define SHQ_NMETALS 9
struct MetalWalkArgs {
    unsigned long long *cursor;
    unsigned long long capacity;
    unsigned long long *keys;      /* particle << 32 | queue position */
    double *wk;
    const int32_t *leaf_pidx;
    int SPHWeighting;
    double MaxGasMass;
    /* by queue position */
    const double *starvolume, *massgenerated, *metalgenerated, *speciesgenerated;
    /* gas state by particle index */
    float *gmass, *gmetals;
    double *gdensity, *gmetallicity;
    uint8_t *touched;              /* gas particles the return changed (out; may be NULL) */
};
/* rows of 3 + SHQ_NMETALS doubles for the gas particles of d_list: mass, density, metallicity, metals */
int shq_metal_rows_gather(shq_context *ctx, const MetalWalkArgs *w, const int32_t *d_list, int64_t m, double *d_rows);
int shq_metal_return_device(shq_context *ctx, MetalWalkArgs *w, int kernel_type, double BoxSize, const int32_t *d_queue, int64_t nq, double *d_massreturn,
                            int64_t *npairs_out);
int shq_bh_accretion_device(shq_context *ctx, const shq_kick_factors *kf, const BhWalkArgs *w, const int32_t *d_queue, int64_t nq, double *d_post);
int shq_bh_feedback_device(shq_context *ctx, const shq_kick_factors *kf, const BhWalkArgs *w, const int32_t *d_queue, int64_t nq);
/* the particles marked in `mark` (one byte each), ascending, into d_list (room for n entries); *m = their number (one host round trip) */
int shq_marked_list(shq_context *ctx, const uint8_t *d_mark, int64_t n, int32_t *d_list, int64_t *m);
/* rows of eight doubles for the particles of d_list: vx, vy, vz, entropy, delay time, mass word of posm, flag byte, `extra` byte (or 0) */
int shq_rows_gather(shq_context *ctx, const int32_t *d_list, int64_t m, const uint8_t *d_extra, double *d_rows);
int shq_u64_gather(shq_context *ctx, const int32_t *d_list, int64_t m, const unsigned long long *d_src, unsigned long long *d_out);
int shq_bh_veldisp_device(shq_context *ctx, const shq_kick_factors *kf, double BoxSize, const int32_t *d_queue, int64_t nq, double *d_out);
int shq_sph_stellar_density_device(shq_context *ctx, const shq_stellar_params *p, const int32_t *d_queue, int64_t nq, double *d_starvol,
                                   shq_sph_stats *stats);

#endif
