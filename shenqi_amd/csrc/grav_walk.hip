/* grav_walk.hip — short-range Barnes-Hut / relative-criterion tree walk for gfx950.
 *
 * Replaces treewalk_primary_kernel<GravTreeWalk> (libgadget/treewalk2.cuh:104-134, one thread
 * per particle, AoS managed memory) + GravLocalTreeWalk::visit (libgadget/gravshort2.hpp:227-322)
 * with a wavefront-collective walk:
 *
 *   - one 64-lane wavefront owns 64 consecutive targets (Peano-ordered => spatially compact);
 *   - the wavefront walks the UNION of its lanes' reference walks.  `cur` (the node being
 *     tested) is wave-uniform, so the node pool is read with scalar loads; every lane keeps
 *     `mynext`, the node its own reference walk would visit next, and takes part in a node test
 *     only while mynext == cur.  A lane that discards or accepts a node sets mynext = sibling
 *     and sleeps until the union walk reaches that sibling; the union opens a node when any
 *     awake lane opens it.  Per target this yields exactly the reference's opening decisions,
 *     interaction set and summation order (depth-first), i.e. the EXACT flavour;
 *   - the node pool is in depth-first pre-order (packed at upload): the first child of node i is
 *     i+1 and the `sibling` of a leaf is i+1 too, so the record after the current one is the
 *     next to be visited unless a whole subtree is skipped; it is fetched speculatively;
 *   - leaf particles come from a leaf-ordered (x,y,z,m) copy, fetched LEAFB at a time with
 *     wave-uniform (scalar) loads;
 *   - the 2x512-float TreePM window table (gravity.h:32-61) is staged in LDS as f64
 *     {f[i], f[i+1]-f[i], p[i], p[i+1]-p[i]}: an interaction needs two ds_read_b128 and two fma.
 *
 * All arithmetic is f64 as in the reference (LOW_PRECISION=double).  The kernel is VALU-issue
 * bound (f64 ops issue at 4 cycles per wave instruction), so the code below is written to keep
 * the per-visit instruction count down.
 */
#include "common.hpp"
#include "pm_readout.hpp"
#include <utility>
#include <stdlib.h>

namespace {

struct WalkArgs {
    const NodeG *nodeG;
    const double4 *posm;       /* by particle index */
    const double4 *posm_leaf;  /* leaf order */
    const double *oldacc;
    const int32_t *targets;    /* may be null */
    const int32_t *qstart;     /* GHOSTS: [ntargets][4] packed start nodes of every query, ascending, -1 = unused */
    double *acc;               /* [N][3] by particle index */
    double *pot;
    int32_t *nint;
    GravStatsDev *stats;
    long long ntargets;
    int root;
    double Box, invBox, halfBox;
    double rcut, rcut2;
    double h, h2, h_inv, h3_inv;
    double inv_celldx;         /* 1 / (cellsize * dx) */
    double errtol, bh2;
    int useBH;
    unsigned xcdK;
    int stats_guard;
    const float *tab_f;
    const float *tab_p;
    unsigned int *task_counters; /* PERSIST: 8 counters, one per XCD region, 64 B apart */
    long long nwaves;            /* PERSIST: number of 64-target tasks */
    int task_run_log2;           /* PERSIST: log2 of the run of consecutive tasks a region owns */
    uint4 *scrub;                /* PM mesh to clear in the shadow of the walk (null: none): task w zeroes the 16-byte units */
    long long scrub_n16;         /*   [w * scrub_per_task, (w + 1) * scrub_per_task) below scrub_n16 */
    int scrub_per_task;          /*   (a multiple of 64) */
    /* READOUT (shq_treepm_step): the task prologue reads GravPM and the PM potential of its targets off the potential mesh and forms
     * OldAcc from them, as gravpm_force -> grav_get_abs_accel do in the reference's order (run.cpp:518-538, gravshort2.hpp:111-121) */
    const double *pm_mesh;       /* [N][N][zp] potential */
    int pmN, pmzp;
    double pmcell, pmffac, G;
    const double *treeacc;       /* FullTreeGravAccel of the previous step, [N][3] */
    const uint8_t *pflags;
    double *gravpm, *pmpot, *oldacc_out;
    /* SPARSE: subtrees that at most SHQ_SPARSE_LANES lanes of a task enter are not walked by the task's wave; it notes them here
     * (slot k of task w: {first node, end node, lane mask}) and grav_pair_kernel walks them with one lane per (target, node) pair */
    int4 *sp_items;
    int32_t *sp_count;
    int4 *sp_stack;              /* pair kernel: sp_stack_cap pairs per resident wave */
    int *sp_flags;               /* pair kernel: the status words of SpFlag below */
    int sp_stack_cap;            /* pairs per resident wave (SHQ_SPARSE_STACK unless a test shrank it) */
    unsigned sp_spin_max;        /* live: polls of a task's flag before the wave gives up (a test knob starves it) */
    int sp_mop;                  /* this launch is the mop-up pass behind a live pair kernel (see grav_pair_kernel) */
    unsigned int *sp_task;       /* pair kernel: task counter */
    int sp_live;                 /* the pair kernel runs BESIDE the main walk (second stream) and takes a task once sp_count[task] >= 0 */
    const int *sp_lean_bad;      /* pair kernel: *sp_lean_bad == 0 says fill_rcuthl_kernel found every record's products reproducible from
                                    {mass, len} by lean_products() below (null: not checked) */
};

/* NEAREST (partmanager.h:99) as d - L*rint(d/L): one multiply, one round, one fma. For
 * |d| < L (always true for positions inside the box) this equals the reference's one-shot wrap
 * except within an ulp of |d| = L/2, where both images are equally valid. */
__device__ __forceinline__ double wrapd(double d, double L, double invL)
{
    return fma(-L, rint(d * invL), d);
}

/* 1/sqrt(x) to f64 accuracy: hardware estimate + one third-order Newton step (the form OCML
 * uses), without the inf/zero special casing: callers clamp x away from zero. */
__device__ __forceinline__ double rsqrt_fast(double x)
{
    const double y0 = __builtin_amdgcn_rsq(x);
    const double e = fma(-x * y0, y0, 1.0);
    return fma(y0 * e, fma(e, 0.375, 0.5), y0);
}

/* A double constant materialised in scalar registers AT THE POINT OF USE.  The softened branch of apply_accn needs ten constants
 * that are no inline literals; left to itself the compiler keeps them in twenty VGPRs for the whole kernel (hoisted out of the walk
 * and, in the persistent variant, out of the task loop), which is the difference between 8 and 6 waves per SIMD.  The volatile asm
 * pins the two s_mov_b32 inside the (rare) branch. */
__device__ __forceinline__ double sconst(double c)
{
    unsigned lo = (unsigned) (unsigned long long) __double_as_longlong(c), hi = (unsigned) ((unsigned long long) __double_as_longlong(c) >> 32);
    asm volatile("" : "+s"(lo), "+s"(hi));
    return __longlong_as_double((long long) (((unsigned long long) hi << 32) | lo));
}

/* apply_accn (gravshort2.hpp:326-358) + apply_short_range_window (gravity.h:48-60).
 * mass/(r2*r) is formed as mass*rinv^3 (a few ulp from the reference's sqrt + divide); a
 * coincident source (r2 == 0, the target itself) is clamped to a tiny r2 so that it falls in
 * the softened branch and contributes dx*fac = 0 exactly as in the reference. */
/* CLAMP: the source may coincide with the target (a leaf particle that is the target itself).  An accepted
 * node never does: a target inside the node's cell opens it (0.6 len test), and outside the cell it cannot sit on
 * the centre of mass, which lies inside. */
template <bool POT, bool CLAMP = true>
__device__ __forceinline__ void apply_accn(const double4 *__restrict__ tab, double dx, double dy, double dz, double r2,
                                           double mass, const WalkArgs &a, double &ax, double &ay, double &az,
                                           double &pot)
{
    const double r2c = CLAMP ? fmax(r2, 1e-280) : r2;
    const double rinv = rsqrt_fast(r2c);
    const double r = r2c * rinv;
    const double mr = mass * rinv;
    double fac = mr * rinv * rinv;
    double npot = mr; /* minus the potential factor: the Newtonian -m/r enters the sum through the negated-operand form of the fma below */
    /* a wave-uniform test first: no lane inside the softening length is the common case, and a scalar compare-and-branch is one
     * scalar instruction less than saving, masking and restoring exec around an empty branch */
    if(shq_ballot(r2 < a.h2) != 0ull) {
        asm volatile("" ::: "memory"); /* keeps the scalar branch apart from the lane mask below */
        if(r2 < a.h2) {
            const double u = r * a.h_inv;
            double wp;
            if(u < 0.5) {
                fac = mass * a.h3_inv * (sconst(10.666666666667) + u * u * (32.0 * u - sconst(38.4)));
                wp = sconst(-2.8) + u * u * (sconst(5.333333333333) + u * u * (sconst(6.4) * u - sconst(9.6)));
                npot = -(mass * a.h_inv * wp);
            } else {
                /* the reference's 0.0667 / u^3 and 0.0667 / u terms are the Newtonian force and potential themselves
                 * (mass h^-3 / u^3 = mass / r^3, mass h^-1 / u = mass / r), already formed above from 1/r: no divisions */
                fac = fma(mass * a.h3_inv, sconst(21.333333333333) - 48.0 * u + sconst(38.4) * u * u - sconst(10.666666666667) * u * u * u,
                          sconst(-0.066666666667) * fac);
                wp = sconst(-3.2) + u * u * (sconst(10.666666666667) + u * (-16.0 + u * (sconst(9.6) - sconst(2.133333333333) * u)));
                npot = -fma(mass * a.h_inv, wp, sconst(0.066666666667) * mr);
            }
        }
    }
    /* apply_short_range_window (gravity.h:48-60): beyond the table (index >= NTAB - 1) the source contributes nothing.  No branch:
     * the index is clamped to the last row, which the LDS copy of the table holds as zeros (row NTAB - 2 keeps its own differences). */
    const double fi = fmin(r * a.inv_celldx, (double) (SHQ_NGRAVTAB - 1));
    const int ti = (int) fi;                     /* fi >= 0: truncation == floor */
    const double w1 = __builtin_amdgcn_fract(fi); /* fi - floor(fi) */
    if(POT) {
        const double4 t = tab[ti];
        fac *= fma(w1, t.y, t.x);
        pot = fma(-npot, fma(w1, t.w, t.z), pot);
    } else {
        const double2 t = *reinterpret_cast<const double2 *>(&tab[ti]);
        fac *= fma(w1, t.y, t.x);
    }
    ax = fma(dx, fac, ax);
    ay = fma(dy, fac, ay);
    az = fma(dz, fac, az);
}

/* wrapm: wave-uniform, zero when the leaf's own visit found every awake lane further than len/2 from the L/2 limit:
 * the particles lie inside the leaf cell, so their displacements need no periodic wrap either */
template <bool POT>
__device__ __forceinline__ void leaf_particle(const double4 *__restrict__ tab, const double4 q, double px, double py,
                                              double pz, const WalkArgs &a, double &ax, double &ay, double &az,
                                              double &pot, const unsigned long long wrapm)
{
    double ex = q.x - px, ey = q.y - py, ez = q.z - pz;
    if(wrapm != 0ull) { /* a scalar compare: a bool would be turned into a lane mask and back */
        ex = wrapd(ex, a.Box, a.invBox);
        ey = wrapd(ey, a.Box, a.invBox);
        ez = wrapd(ez, a.Box, a.invBox);
    }
    const double rr2 = ex * ex + ey * ey + ez * ez;
    apply_accn<POT>(tab, ex, ey, ez, rr2, q.w, a, ax, ay, az, pot);
}

/* POT: accumulate the potential.  PREFETCH: speculative fetch of pool[cur+1].  LEAFB: leaf
 * particles fetched per batch (2 or 4).  STATS: wave-level counters for the bench.
 * amdgpu_num_sgpr(96): the kernel wants 106 SGPRs, which allocates 112 and caps a SIMD at 7 waves; held to 96 (ten values
 * parked in lanes of a spare VGPR, still 64 VGPRs) it runs 8: walk 44.45 -> 43.6 ms in a same-box A/B. */
/* PERSIST: the grid is as many workgroups as the chip holds at once and every wave takes 64-target tasks from a counter until
 * none is left, instead of one task per wave: a finished wave's slot is refilled at once rather than when the slowest of its
 * workgroup's four waves ends (measured before: 6.5 of 8 wave slots per SIMD occupied on average), and the window table is
 * staged once per resident workgroup, not once per 256 targets.  Tasks keep the XCD-chunked order of the one-shot grid: region x
 * (= the XCD a workgroup most likely runs on, blockIdx % 8) owns every eighth run of 4 K tasks; a wave whose region is exhausted
 * takes from the next one, so the clustered part of the box cannot leave seven XCDs idle. */
/* The fields a task needs only before and after its walk (list, particle and output pointers, counters) must not stay in
 * registers across the walk: hoisted out of the task loop they cost ~20 SGPRs, which under the 96-SGPR cap spill into VGPR lanes
 * and from there to scratch.  They are re-read from the kernel argument segment through a pointer the optimiser cannot see
 * through, at the points of use. */
/* constant address space on the device pass; the host pass only parses the kernels */
#if defined(__HIP_DEVICE_COMPILE__)
#define SHQ_CONSTANT_AS __attribute__((address_space(4)))
#else
#define SHQ_CONSTANT_AS
#endif
typedef const SHQ_CONSTANT_AS WalkArgs *WalkArgsK;
__device__ __forceinline__ WalkArgsK walk_cold_args()
{
    WalkArgsK kp = (WalkArgsK) __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    return kp;
}

/* What the main walk hands to the pair kernel while both run (results of a task, its noted subtrees, OldAcc) crosses XCDs, whose L2s
 * are not coherent with each other inside a launch: those few stores and loads go to the device's coherence point (agent-scope
 * relaxed atomics: sc1), ordered by the wave's own s_waitcnt and the task's flag; everything else stays cached as before */
template <typename T> __device__ __forceinline__ void agent_store(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T> __device__ __forceinline__ T agent_load(const T *p) { return __hip_atomic_load(const_cast<T *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

/* returns the next task of this wave (wave-uniform), or -1 when all eight regions are exhausted */
__device__ __forceinline__ int walk_next_task(WalkArgsK c, int &region, int &tried)
{
    const int lg = c->task_run_log2;
    const int nwaves = (int) c->nwaves;
    unsigned int *counters = c->task_counters;
    while(tried < 8) {
        unsigned k = 0;
        if((threadIdx.x & 63) == 0)
            k = atomicAdd(&counters[16 * region], 1u);
        k = (unsigned) __builtin_amdgcn_readfirstlane((int) k);
        const unsigned t = ((((k >> lg) << 3) + (unsigned) region) << lg) | (k & ((1u << lg) - 1u));
        if(k < 0x40000000u && t < (unsigned) nwaves)
            return (int) t;
        region = (region + 1) & 7;
        tried++;
    }
    return -1;
}

/* RING (leaf ring): the particles of an opened leaf are not evaluated at once — a round per particle in which only the lanes that
 * opened this leaf work, < 10 % of the lanes on average — but copied (LDS-DMA, no registers) into a wave-private ring of SHQ_LEAF_RING
 * sources in LDS, every lane noting in a bit mask which slots are its own.  When the ring is full, and at the end of the task, the
 * wave drains it: in every round each lane pops ITS oldest slot, so lanes that opened different leaves work side by side.  A lane's
 * particles are still evaluated in the order its walk met them; only their position relative to the node interactions changes (the
 * last bits of the sums).  tools/walk_defer_sim.py: leaf rounds 340 -> 219 per wave (64^3 S-cluster) at 32 slots. */
#define SHQ_LEAF_RING 32
#ifndef SHQ_SPARSE_LANES
#define SHQ_SPARSE_LANES 8     /* a subtree entered by at most this many lanes goes to the pair kernel */
#endif
#ifndef SHQ_SPARSE_CAP
#define SHQ_SPARSE_CAP 96      /* ... while the task has a free slot (31.7 per task on average at 256^3) */
#endif
#define SHQ_SPARSE_STACK 4096  /* pairs per resident wave of the pair kernel: 9 x the deepest stack seen (443 at 256^3 S-cluster, shq_walk_pair_status) */
#define SHQ_SPARSE_SPIN (1u << 22)
#ifndef SHQ_PAIR_DEPTH2
#define SHQ_PAIR_DEPTH2 1      /* the pair kernel takes two batches of 64 pairs per round (0: one) */
#endif
/* The pair kernel's status words (ctx->sp_flags).  [0, 8) belong to one launch and are cleared by the next; [8, 16) are STICKY: no
 * launch clears them, a copy travels to pinned host memory behind every launch, and the library's entry points return SHQ_ERR_DEVICE
 * once the error word is up (shq_walk_check_status) - the reference checks every launch and ends the run (treewalk2.cuh:351-353). */
enum SpFlag {
    SP_OVERFLOW = 0,   /* a wave's pair stack ran full: it dropped pairs (sums incomplete) */
    SP_GAVEUP = 1,     /* live: a wave gave up waiting for a task's flag; the mop-up pass behind the walk takes what it left */
    SP_MOPPED = 2,     /* tasks the mop-up pass walked */
    SP_HIGHWATER = 3,  /* deepest pair stack of the launch */
    SP_STICKY_ERROR = 8,     /* bit 0: some launch since the last report overflowed a pair stack */
    SP_STICKY_RECOVERED = 9, /* launches in which the mop-up pass had to finish the live pair kernel's work */
    SP_STICKY_HIGHWATER = 10,
    SP_TASK_COUNTER = 16
};
#define SP_TASK_DONE (-2)      /* sp_count[task] once the pair kernel has added the task's sums back: a second pass skips it */
/* the pair kernel's workgroups: waves per workgroup (one window table each), workgroups per CU, waves per SIMD */
#ifndef SHQ_PAIR_WAVES
#define SHQ_PAIR_WAVES 8
#define SHQ_PAIR_WG_PER_CU 2
#define SHQ_PAIR_EU 4
#endif

template <bool POT>
__device__ __forceinline__ void leaf_ring_drain(const double4 *__restrict__ tab, const double4 *ring, unsigned &mymask, double px, double py,
                                                double pz, const WalkArgs &a, double &ax, double &ay, double &az, double &pot,
                                                const unsigned long long wrapm)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* the LDS-DMA copies into the ring have landed */
    while(shq_ballot(mymask != 0u) != 0ull) {
        if(mymask != 0u) {
            const int b = __builtin_ctz(mymask);
            mymask &= mymask - 1u;
            const double4 q = ring[b];
            leaf_particle<POT>(tab, q, px, py, pz, a, ax, ay, az, pot, wrapm);
        }
    }
}

template <bool POT, bool PREFETCH, int LEAFB, int STATS, bool BH, bool GHOSTS = false, bool PERSIST = false, bool RING = false, bool READOUT = false,
          bool SPARSE = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_num_sgpr(96), amdgpu_num_vgpr(64))) void grav_walk_exact_kernel(const WalkArgs a)
{
    extern __shared__ double4 ring_all[]; /* RING: SHQ_LEAF_RING slots per wave */
    double4 *const ring = ring_all + (threadIdx.x >> 6) * SHQ_LEAF_RING;
    __shared__ double4 tab[SHQ_NGRAVTAB];
    for(int i = threadIdx.x; i < SHQ_NGRAVTAB; i += blockDim.x) {
        const int j = (i + 1 < SHQ_NGRAVTAB) ? i + 1 : i;
        const double f0 = a.tab_f[i], f1 = a.tab_f[j], p0 = a.tab_p[i], p1 = a.tab_p[j];
        tab[i] = i + 1 < SHQ_NGRAVTAB ? make_double4(f0, f1 - f0, p0, p1 - p0) : make_double4(0, 0, 0, 0); /* last row: see apply_accn */
    }
    __syncthreads();

    /* the node pool and the leaf-ordered particle copy are read-only for the whole launch: read through the constant address
     * space, so that the loads stay scalar (s_load) although the task loop puts a task's result stores before the next task's
     * node loads (the scalar cache is not coherent with vector stores, and the compiler must assume they may alias) */
    typedef const SHQ_CONSTANT_AS NodeG *NodeGK;
    typedef const SHQ_CONSTANT_AS double4 *Double4K;
    const NodeGK nodeG = (NodeGK) (size_t) a.nodeG;
    const Double4K posm_leaf = (Double4K) (size_t) a.posm_leaf;
    const int lane = threadIdx.x & 63;
    int region = blockIdx.x & 7, tried = 0;
    long long wave = PERSIST ? (long long) walk_next_task(walk_cold_args(), region, tried)
                             : (long long) xcd_block(blockIdx.x, gridDim.x, a.xcdK) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    while(wave >= 0) {
    const long long t = wave * 64 + lane;
    long long pi = 0;
    double px = 0, py = 0, pz = 0, aold = 0;
    bool valid;
    {
        const WalkArgs c = PERSIST ? *walk_cold_args() : a;
        valid = t < c.ntargets;
        if(valid) {
            pi = c.targets ? (long long) c.targets[t] : t;
            valid = pi >= 0; /* a negative list entry is an idle lane (padding of cell-aligned target groups) */
        }
        if(valid) {
            const double4 p = c.posm[pi];
            px = p.x;
            py = p.y;
            pz = p.z;
            if(READOUT) {
                /* readout_potential / readout_force_{x,y,z} (gravpm.cpp:489-500) for this target, then grav_get_abs_accel
                 * (gravshort2.hpp:111-121) with the NEW GravPM: 104 loads and ~330 vector instructions per lane and task (of 63 k),
                 * the loads' latency hidden behind the walks of the SIMD's other waves.  Same operations as pm_readout_kernel +
                 * oldacc_kernel, so the same bits. */
                double g0 = 0, g1 = 0, g2 = 0, gp = 0;
                pm_readout_lean(c.pm_mesh, c.pmN, c.pmzp, c.pmcell, c.pmffac, px, py, pz, !(c.pflags && (c.pflags[pi] & 2)), g0, g1, g2, gp);
                c.gravpm[3 * pi + 0] = g0;
                c.gravpm[3 * pi + 1] = g1;
                c.gravpm[3 * pi + 2] = g2;
                c.pmpot[pi] = gp;
                const double gv[3] = {g0, g1, g2};
                double s = 0;
                {
#pragma clang fp contract(off)
                    for(int j = 0; j < 3; j++) {
                        const double ax = c.treeacc[3 * pi + j] + gv[j];
                        s += ax * ax;
                    }
                }
                const double oa = sqrt(s) / c.G;
                if(SPARSE)
                    agent_store(&c.oldacc_out[pi], oa);
                else
                    c.oldacc_out[pi] = oa;
                aold = c.errtol * oa;
            } else
                aold = c.errtol * c.oldacc[pi];
        }
        /* The PM's next deposit mesh is cleared here: 3.7 GB of stores spread over the walk's 34 ms (0.1 TB/s of a memory system
         * the walk leaves idle, a dozen store instructions per task beside 64 k arithmetic ones) instead of a 0.63 ms kernel of
         * their own.  shq_grav_short_run decides (pm.hip: mesh_zeroed). */
        if(c.scrub_per_task > 0) {
            const long long u0 = wave * c.scrub_per_task + lane;
            for(int k = 0; k < c.scrub_per_task; k += 64)
                if(u0 + k < c.scrub_n16) {
                    unsigned z = 0;
                    asm volatile("" : "+v"(z)); /* the zeros are made here: hoisted out of the task loop they hold four registers for good */
                    c.scrub[u0 + k] = make_uint4(z, z, z, z);
                }
        }
    }
    double ax = 0, ay = 0, az = 0, pot = 0;
    int nint = 0;
    int sp_n = 0;                      /* SPARSE: subtrees this task has handed to the pair kernel (wave-uniform) */
    unsigned ringmask = 0;             /* RING: this lane's pending slots */
    int ringfill = 0;                  /* RING: slots in use (wave-uniform) */
    unsigned long long ringwrap = 0;   /* RING: some pending leaf needs the periodic wrap (wave-uniform) */
    unsigned int node_int_wave = 0; /* STATS: monopole interactions of the whole wave (wave-uniform: no VGPR) */
    int mynext = valid ? a.root : -2;
    int cur = a.root;
    /* GHOSTS (GravLocalTreeWalk::visit<TREEWALK_GHOSTS>, gravshort2.hpp:243-261): an imported query walks only
     * the branches under the top-level nodes of its NodeList, i.e. the pre-order index ranges
     * [start, sibling(start)).  The lane waits at the start of its next branch; the wave cursor still begins
     * at the root and descends wherever an awake lane opens OR a lane waits further down (see below). */
    int seg1 = -1, seg2 = -1, seg3 = -1, myend = -1;
    if(GHOSTS) {
        mynext = -2;
        if(valid) {
            const int4 st = *reinterpret_cast<const int4 *>(a.qstart + 4 * t);
            mynext = st.x >= 0 ? st.x : -2;
            seg1 = st.y; seg2 = st.z; seg3 = st.w;
            if(mynext >= 0)
                myend = a.nodeG[mynext].sibling; /* per-lane (vector) load */
        }
    }
    unsigned int visited = 0, wave_applies = 0, wave_node_applies = 0;
    unsigned int hv[8] = {}, hn[8] = {}, hl[8] = {};
    unsigned int lonely8 = 0, lonely16 = 0; /* STATS == 2: this lane's interactions in rounds of <= 8 / <= 16 lanes */

    do { /* cur >= 0: the root on entry, then every node the union walk reaches */
        /* the compiler fetches the record with four scalar loads (x16, x4, x8, x4).  Two s_load_dwordx16 of the two 64-byte halves
         * were measured instead: 39.6 against 33.9 ms in a same-box A/B — not worth the two issue slots */
        const NodeG nd = nodeG[cur];
        if(STATS)
            visited++;
        const bool act = (mynext == cur);
        if(STATS == 2)
            hv[(__popcll(shq_ballot(mynext == cur)) - 1) >> 3 & 7]++;

        /* gravshort2.hpp:262-265 */
        /* The periodic wrap is the identity unless some |d| exceeds L/2.  The centre of mass lies inside the cell,
         * so while every |center - pos| of the awake lanes stays below L/2 - len/2 (precomputed with the node)
         * neither vector needs it: one wave-uniform test on the maximum that the opening tests need anyway
         * skips the six wraps for the overwhelming majority of nodes. */
        double dx = nd.cofm[0] - px, dy = nd.cofm[1] - py, dz = nd.cofm[2] - pz;
        double ux = nd.center[0] - px, uy = nd.center[1] - py, uz = nd.center[2] - pz;
        double cmax = fmax(fmax(fabs(ux), fabs(uy)), fabs(uz));
        /* Wave votes are formed from the lane masks of the single comparisons (shq_ballot of one compare is that compare's own
         * result) and combined with scalar logic; the per-lane decisions are those masks read back as lane conditions
         * (inverse ballot: no instruction).  A vote on a compound boolean would cost two VALU instructions, and lane booleans
         * computed beside the masks would repeat the whole scalar algebra: every scalar instruction holds the SIMD's scalar pipe
         * for four cycles, as long as an f64 instruction holds the vector pipe, and the walk issues nearly as many of the one as
         * of the other (SQ_INSTS_SALU 61 k, SQ_INSTS_VALU 66 k per task). */
        const unsigned long long actm = shq_ballot(mynext == cur);
        unsigned long long wrapm = 0ull;
        int wl_hi = __double2hiint(nd.wraplim);
        asm volatile("" : "+s"(wl_hi)); /* a 32-bit scalar compare (left alone the compiler widens it to a 64-bit VECTOR compare) */
        if(wl_hi >= 0) { /* not an interior node (fill_rcuthl_kernel): the wrap may matter */
            asm volatile("" ::: "memory");
            wrapm = shq_ballot(cmax >= nd.wraplim) & actm; /* >=: a zero limit (the root) means always; wrapping at equality is the identity */
            if(wrapm != 0ull) {
                dx = wrapd(dx, a.Box, a.invBox);
                dy = wrapd(dy, a.Box, a.invBox);
                dz = wrapd(dz, a.Box, a.invBox);
                ux = wrapd(ux, a.Box, a.invBox);
                uy = wrapd(uy, a.Box, a.invBox);
                uz = wrapd(uz, a.Box, a.invBox);
                cmax = fmax(fmax(fabs(ux), fabs(uy)), fabs(uz));
            }
        }
        const double r2 = dx * dx + dy * dy + dz * dz;
        /* shall_we_discard_node, gravshort2.hpp:152-167 (rcuthl = Rcut + len / 2, per node).  (Making the second compare conditional
         * on some lane being beyond Rcut at all — a scalar branch for a vector compare — measured 34.1 against 33.6 ms.) */
        const unsigned long long keepm = actm & ~(shq_ballot(r2 > nd.rcut2) & shq_ballot(cmax > nd.rcuthl));
        /* shall_we_open_node, gravshort2.hpp:172-193 (len*len/r2 > theta2 written without the divide;
         * mass*len*len and 0.6*len come precomputed with the node) */
        const unsigned long long openm = (BH ? 0ull : shq_ballot(nd.mlen2 > r2 * r2 * aold)) | shq_ballot(r2 < nd.bhlim) |
                                         shq_ballot(cmax < nd.inside);
        const unsigned long long acceptm = keepm & ~openm, doopenm = keepm & openm;
        const bool accept = __builtin_amdgcn_inverse_ballot_w64(acceptm), doopen = __builtin_amdgcn_inverse_ballot_w64(doopenm);

        {
            if(STATS && acceptm != 0ull) {
                wave_applies++;
                wave_node_applies++;
                node_int_wave += (unsigned int) __popcll(acceptm);
            }
            if(STATS == 2 && acceptm != 0ull) {
                const int pc = __popcll(acceptm);
                hn[(pc - 1) >> 3 & 7]++;
                if(accept && pc <= 8)
                    lonely8++;
                if(accept && pc <= 16)
                    lonely16++;
            }
            if(accept) {
                apply_accn<POT, false>(tab, dx, dy, dz, r2, nd.mass, a, ax, ay, az, pot);
                nint++;
            }
        }
        int next;
        if(nd.type == SHQ_PARTICLE_NODE_TYPE) {
            /* gravshort2.hpp:290-304: every particle of an opened leaf is evaluated */
            asm volatile("" ::: "memory"); /* two scalar compare-and-branch pairs, not a merged boolean (five more scalar instructions) */
            if(RING && doopenm != 0ull) {
                const int cnt = nd.count;
                if(ringfill + cnt > SHQ_LEAF_RING) {
                    leaf_ring_drain<POT>(tab, ring, ringmask, px, py, pz, a, ax, ay, az, pot, ringwrap);
                    ringfill = 0;
                    ringwrap = 0;
                }
                /* cnt x 32 bytes of the leaf-ordered copy -> ring slots [ringfill, ringfill + cnt): lane l moves 16 bytes */
                if(lane < 2 * cnt)
                    __builtin_amdgcn_global_load_lds(reinterpret_cast<const double2 *>(a.posm_leaf + nd.child) + lane,
                                                     (__attribute__((address_space(3))) void *) (ring + ringfill), 16, 0, 0);
                if(doopen) {
                    ringmask |= ((1u << cnt) - 1u) << ringfill;
                    nint += cnt;
                }
                ringfill += cnt;
                ringwrap |= wrapm;
            }
            if(!RING && doopenm != 0ull) {
                const Double4K lp = posm_leaf + nd.child;
                const int cnt = nd.count;
                if(STATS)
                    wave_applies += cnt;
                if(STATS == 2) {
                    const int pc = __popcll(doopenm);
                    hl[(pc - 1) >> 3 & 7] += cnt;
                    if(doopen && pc <= 8)
                        lonely8 += cnt;
                    if(doopen && pc <= 16)
                        lonely16 += cnt;
                }
                /* leaf slots are contiguous and the array is padded by NMAXCHILD entries */
#pragma unroll
                for(int b = 0; b < SHQ_NMAXCHILD; b += LEAFB) {
                    if(b < cnt) {
                        double4 q[LEAFB];
#pragma unroll
                        for(int k = 0; k < LEAFB; k++)
                            q[k] = lp[b + k];
                        if(doopen) {
#pragma unroll
                            for(int k = 0; k < LEAFB; k++)
                                if(b + k < cnt)
                                    leaf_particle<POT>(tab, q[k], px, py, pz, a, ax, ay, az, pot, wrapm);
                        }
                    }
                }
                if(doopen)
                    nint += cnt;
            }
        }
        /* Links.  A leaf or a pseudo node (gravshort2.hpp:305-315: skipped by the local walk) is left through its sibling whatever
         * the lane decided; an internal node is entered by the lanes that open it.  Written as one select on a wave-uniform
         * "is internal" flag rather than three branches on the node type: the compiler lowered those to a dozen scalar flag
         * moves and branches per visit. */
        {
            /* the lanes that descend: those that open an internal node (a wave-uniform select on the node type, not a branch) */
            unsigned long long descendm = nd.type == SHQ_NODE_NODE_TYPE ? doopenm : 0ull;
            if(SPARSE && descendm != 0ull) {
                asm volatile("" ::: "memory");
                if(__popcll(descendm) <= SHQ_SPARSE_LANES && sp_n < SHQ_SPARSE_CAP) {
                    /* few lanes want this subtree: note it and pass on, as if they had accepted the node (its own monopole was
                     * evaluated above for the lanes that accept it; the lanes in the mask OPEN it: the pair kernel starts at its
                     * first child) */
                    if(lane == 0) {
                        unsigned long long *const it = reinterpret_cast<unsigned long long *>(walk_cold_args()->sp_items + (wave * SHQ_SPARSE_CAP + sp_n));
                        agent_store(it, ((unsigned long long) (unsigned) nd.sibling << 32) | (unsigned) nd.child);
                        agent_store(it + 1, descendm);
                    }
                    sp_n++;
                    descendm = 0ull;
                }
                descendm = ((unsigned long long) (unsigned) __builtin_amdgcn_readfirstlane((int) (descendm >> 32)) << 32) |
                           (unsigned) __builtin_amdgcn_readfirstlane((int) descendm);
            }
            const unsigned long long descendm_lanes = descendm;
            if(GHOSTS && nd.type == SHQ_NODE_NODE_TYPE) /* a lane waits at a branch below this node: go down even if nobody opens it */
                descendm |= shq_ballot(mynext > cur && (nd.sibling < 0 || mynext < nd.sibling));
            /* Two moves of a scalar operand under the two lane masks, written out: as a select the compiler moves both values into
             * vector registers first (2 v_mov + 2 v_cndmask, 12 vector-pipe cycles per visit against 4 here), and the vector pipe
             * is the busier one (SQ_ACTIVE_INST_VALU 0.90, SQ_ACTIVE_INST_SCA 0.52 of the cycles).  exec is the whole wave here
             * (top level of the visit loop) and is restored. */
            {
                const unsigned long long staym = actm & ~descendm_lanes;
                unsigned long long saved;
                asm volatile("s_mov_b64 %[sv], exec\n\t"
                             "s_mov_b64 exec, %[m1]\n\t"
                             "v_mov_b32 %[dst], %[sib]\n\t"
                             "s_mov_b64 exec, %[m2]\n\t"
                             "v_mov_b32 %[dst], %[chd]\n\t"
                             "s_mov_b64 exec, %[sv]"
                             : [dst] "+v"(mynext), [sv] "=&s"(saved)
                             : [m1] "s"(staym), [m2] "s"(descendm_lanes), [sib] "s"(nd.sibling), [chd] "s"(nd.child));
            }
            next = descendm != 0ull ? nd.child : nd.sibling;
        }
        if(GHOSTS && act && mynext == myend) { /* branch done: wait at the next one of the NodeList */
            mynext = seg1 >= 0 ? seg1 : -2;
            seg1 = seg2;
            seg2 = seg3;
            seg3 = -1;
            myend = mynext >= 0 ? a.nodeG[mynext].sibling : -1;
        }
        cur = __builtin_amdgcn_readfirstlane(next);
    } while(cur >= 0);

    if(RING)
        leaf_ring_drain<POT>(tab, ring, ringmask, px, py, pz, a, ax, ay, az, pot, ringwrap);
    {
    const WalkArgs c = PERSIST ? *walk_cold_args() : a;
    if(valid) {
        if(SPARSE) {
            agent_store(&c.acc[3 * pi + 0], ax);
            agent_store(&c.acc[3 * pi + 1], ay);
            agent_store(&c.acc[3 * pi + 2], az);
            if(POT)
                agent_store(&c.pot[pi], pot);
            agent_store(&c.nint[pi], (int32_t) nint);
        } else {
            c.acc[3 * pi + 0] = ax;
            c.acc[3 * pi + 1] = ay;
            c.acc[3 * pi + 2] = az;
            if(POT)
                c.pot[pi] = pot;
            c.nint[pi] = nint;
        }
    }
    if(SPARSE) { /* the task's flag, behind everything the pair kernel reads of it */
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if(lane == 0)
            agent_store(&c.sp_count[wave], (int32_t) sp_n);
    }
    /* statistics (treewalk2.h:446-448 interaction min/max) */
    long long mn = valid ? (long long) nint : __double_as_longlong(sconst(__longlong_as_double(0x7fffffffffffll))), mx = valid ? nint : 0, sm = valid ? nint : 0;
    for(int off = 32; off > 0; off >>= 1) {
        long long o1 = __shfl_xor(mn, off), o2 = __shfl_xor(mx, off), o3 = __shfl_xor(sm, off);
        mn = o1 < mn ? o1 : mn;
        mx = o2 > mx ? o2 : mx;
        sm += o3;
    }
    unsigned int l8s = lonely8, l8m = lonely8, l16s = lonely16, l16m = lonely16;
    if(STATS == 2)
        for(int off = 32; off > 0; off >>= 1) {
            l8s += __shfl_xor(l8s, off);
            l16s += __shfl_xor(l16s, off);
            l8m = max(l8m, (unsigned int) __shfl_xor(l8m, off));
            l16m = max(l16m, (unsigned int) __shfl_xor(l16m, off));
        }
    if(!SPARSE && lane == 0 && c.stats && c.stats_guard != 2) { /* SPARSE: the pair kernel tallies, once the counts are complete */
        atomicAdd(&c.stats->ninteractions, (unsigned long long) sm);
        /* 262 144 waves end here with three no-return atomics on one cache line.  Measured at 256^3 (same box, SHQ_WALK_STATS_GUARD):
         * as they are 39.3 ms, without any of them 39.3 ms, with a read first so that only improvements reach atomicMin / atomicMax
         * 40.9 ms — the read has to come back through the line the atomics are queued on, and the wave holds its slot meanwhile. */
        if(c.stats_guard != 1 || mn < ((volatile long long *) &c.stats->min_int)[0])
            atomicMin(&c.stats->min_int, mn);
        if(c.stats_guard != 1 || mx > ((volatile long long *) &c.stats->max_int)[0])
            atomicMax(&c.stats->max_int, mx);
        if(STATS) {
            atomicAdd(&c.stats->nvisited, (unsigned long long) visited);
            atomicAdd(&c.stats->nwave_applies, (unsigned long long) wave_applies);
            atomicAdd(&c.stats->nwave_node_applies, (unsigned long long) wave_node_applies);
            atomicAdd(&c.stats->nnode_interactions, (unsigned long long) node_int_wave);
        }
        if(STATS == 2)
            for(int b = 0; b < 8; b++) {
                atomicAdd(&c.stats->hist_visit[b], (unsigned long long) hv[b]);
                atomicAdd(&c.stats->hist_node[b], (unsigned long long) hn[b]);
                atomicAdd(&c.stats->hist_leaf[b], (unsigned long long) hl[b]);
            }
        if(STATS == 2) {
            atomicAdd(&c.stats->lonely[0], (unsigned long long) l8s);
            atomicAdd(&c.stats->lonely[1], (unsigned long long) l8m);
            atomicAdd(&c.stats->lonely[2], (unsigned long long) l16s);
            atomicAdd(&c.stats->lonely[3], (unsigned long long) l16m);
        }
    }
    }
    wave = PERSIST ? (long long) walk_next_task(walk_cold_args(), region, tried) : -1;
    } /* task loop */
}

/* ---- the sparse subtrees: one lane per (target, node) pair ---------------------------------------------------------------
 * A wave of the main walk spends a node visit's ~45 vector instructions whether 64 or 3 of its lanes are awake, and the subtrees
 * that at most eight lanes enter are 21 % of its visits and 27 % of its monopole rounds for 2.4 % of the interactions.  Here a wave
 * takes ONE task's noted subtrees (same 64 targets, target slot = lane of the main walk) and keeps a stack of pairs {node, end,
 * slot}: it pops 64, every lane loads ITS node's record, tests it against ITS target with the reference's expressions
 * (gravshort2.hpp:152-193, the same code path as the main walk: identical decisions), evaluates it on accept into the target's LDS
 * accumulator, walks a leaf's particles on open, pushes the first child on open of an internal node, and pushes the sibling while
 * the end of the subtree is not reached.  All lanes run the same code whatever their pair is.  The sums are added to what the main
 * walk stored for the task's targets (this wave alone touches them: no atomics, fixed order), and the wave tallies the
 * interaction statistics the main walk left to it.  Per-target interaction sets are the reference's; the order of the sum is not. */
/* The per-node products of a NodeG record that follow from {mass, len} and the walk parameters, in the expressions that fill them
 * (tree_build.hip pack kernel / capi.hip upload: mlen2, inside, wraplim; fill_rcuthl_kernel: rcuthl).  0.5 * len and 0.5 * Box are exact,
 * so each sum is one rounding whether or not the compiler contracts it, and the products have no additions to contract. */
struct LeanProducts { double rcuthl, mlen2, inside, wraplim; };
__device__ __forceinline__ LeanProducts lean_products(double mass, double len, double rcut, double Box)
{
    LeanProducts r;
    const double hl = 0.5 * len;
    r.rcuthl = rcut + hl;
    r.mlen2 = mass * len * len;
    r.inside = 0.6 * len;
    r.wraplim = fmax(0.5 * Box - hl, 0.0);
    return r;
}
/* level of a node whose side is rootlen 2^-level: same mantissa, so the high words differ by level << 20 */
__device__ __forceinline__ int lean_level(double rootlen, double len) { return (__double2hiint(rootlen) - __double2hiint(len)) >> 20; }

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wpass-failed"
template <bool POT, int WAVES>
__device__ __forceinline__ void grav_pair_body(const WalkArgs &a)
{
    /* the workgroup's waves share one window table: 52 KB per workgroup of eight waves, two of them per CU (behind the walk); 34 KB per
     * workgroup of four (beside it) */
    __shared__ double4 tab[SHQ_NGRAVTAB];
    __shared__ double4 tgt_all[WAVES][64];   /* px, py, pz, errtol * OldAcc of the task's targets */
    __shared__ double acc_all[WAVES][4][64]; /* ax, ay, az, pot per target */
    __shared__ int cnt_all[WAVES][64];       /* interaction count per target */
    __shared__ double bh_lvl[32];                     /* len^2 / theta^2 by tree level */
    /* The mop-up pass: launched behind every live pair kernel, on the main stream (the main walk and the live pair kernel are through:
     * every task's flag is up).  Nothing to do unless a live wave gave up waiting (SP_GAVEUP) - then the tasks not marked SP_TASK_DONE
     * are walked here, in the stride order of the pair kernel run behind the walk: same sums, same order per task. */
    if(a.sp_mop && agent_load(a.sp_flags + SP_GAVEUP) == 0)
        return;
    const double rootlen = a.Box * 1.001;             /* forcetree.cpp:661 */
    if(threadIdx.x < 32) {
        const double l = ldexp(rootlen, -(int) threadIdx.x);
        bh_lvl[threadIdx.x] = l * l / a.bh2; /* fill_rcuthl_kernel's expression on the same len */
    }
    const bool lean = __builtin_amdgcn_readfirstlane(a.sp_lean_bad ? (*a.sp_lean_bad == 0) : 0) != 0;
    for(int i = threadIdx.x; i < SHQ_NGRAVTAB; i += blockDim.x) {
        const int j = (i + 1 < SHQ_NGRAVTAB) ? i + 1 : i;
        const double f0 = a.tab_f[i], f1 = a.tab_f[j], p0 = a.tab_p[i], p1 = a.tab_p[j];
        tab[i] = i + 1 < SHQ_NGRAVTAB ? make_double4(f0, f1 - f0, p0, p1 - p0) : make_double4(0, 0, 0, 0);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double4 *const tgt = tgt_all[wv];
    double(*const acc)[64] = acc_all[wv];
    int *const cntl = cnt_all[wv];
    int4 *const stack = a.sp_stack + ((size_t) blockIdx.x * WAVES + wv) * (size_t) a.sp_stack_cap;
    int sp_high = 0;
    int mopped = 0;
    const unsigned long long below = (1ull << lane) - 1ull;
    /* tasks in a fixed stride over the resident waves, and ONE set of tallies per wave at the end: a returning atomic on one word
     * completes ~88 times per microsecond chip-wide and three no-return ones on one line take ~12 ns each (MI355X_MICROARCH.md) —
     * per task that was 10 ms for this kernel, whatever it did in between */
    long long w_sum = 0, w_min = 0x7fffffffffffll, w_max = 0;
    /* live (beside the main walk): the next task in order from one counter — the main walk finishes its tasks roughly in order, 9 per
     * microsecond, and only one of this kernel's workgroups fits a CU beside it: the others start when it ends and share what is left */
    auto next_task = [&](long long prev) -> long long {
        if(a.sp_live) {
            unsigned k = 0;
            if(lane == 0)
                k = atomicAdd(a.sp_task, 1u);
            return (long long) (unsigned) __builtin_amdgcn_readfirstlane((int) k);
        }
        return prev < 0 ? (long long) blockIdx.x * WAVES + wv : prev + (long long) gridDim.x * WAVES;
    };
    for(long long task = next_task(-1); task < a.nwaves; task = next_task(task)) {
        const long long t = task * 64 + lane;
        const bool valid = t < a.ntargets;
        const long long pi = valid ? (a.targets ? (long long) a.targets[t] : t) : -1;
        const bool have = pi >= 0;
        int cnt = agent_load(&a.sp_count[task]);
        if(a.sp_mop) {
            if(__builtin_amdgcn_readfirstlane(cnt) == SP_TASK_DONE)
                continue;
            mopped++;
        }
        if(a.sp_live) {
            /* beside the main walk: wait for the task (the main walk never waits for this kernel, and every task below nwaves gets its
             * flag; the bound only keeps a broken launch from hanging the card: ~4 s, then the error flag) */
            for(unsigned spin = 0; cnt < 0 && spin < a.sp_spin_max; spin++) {
                __builtin_amdgcn_s_sleep(32);
                cnt = agent_load(&a.sp_count[task]);
            }
            cnt = __builtin_amdgcn_readfirstlane(cnt);
            if(cnt < 0) {
                if(lane == 0)
                    agent_store(a.sp_flags + SP_GAVEUP, 1);
                break;
            }
        }
        int sp = 0;
        if(cnt > 0) {
            double4 me = make_double4(0, 0, 0, 0);
            if(have) {
                const double4 p = a.posm[pi];
                me = make_double4(p.x, p.y, p.z, a.errtol * agent_load(&a.oldacc[pi]));
            }
            tgt[lane] = me;
            for(int k = 0; k < 4; k++)
                acc[k][lane] = 0.0;
            cntl[lane] = 0;
            /* the noted subtrees as pairs (first node, end, slot), in the order the task noted them: the items are fetched 64 at a
             * time, one per lane (one trip to memory, not one per item), and handed round with readlane */
            for(int k0 = 0; k0 < cnt; k0 += 64) {
                int4 mine = make_int4(0, 0, 0, 0);
                if(k0 + lane < cnt) {
                    const unsigned long long *const it = reinterpret_cast<const unsigned long long *>(a.sp_items + (task * SHQ_SPARSE_CAP + k0 + lane));
                    const unsigned long long lo = agent_load(it), hi = agent_load(it + 1);
                    mine = make_int4((int) (unsigned) lo, (int) (unsigned) (lo >> 32), (int) (unsigned) hi, (int) (unsigned) (hi >> 32));
                }
                const int kn = cnt - k0 < 64 ? cnt - k0 : 64;
                for(int k = 0; k < kn; k++) {
                    const int ix = __builtin_amdgcn_readlane(mine.x, k), iy = __builtin_amdgcn_readlane(mine.y, k);
                    const unsigned long long m = ((unsigned long long) (unsigned) __builtin_amdgcn_readlane(mine.w, k) << 32) |
                                                 (unsigned) __builtin_amdgcn_readlane(mine.z, k);
                    if((m >> lane) & 1ull)
                        stack[sp + __popcll(m & below)] = make_int4(ix, iy, lane, 0);
                    sp += __popcll(m);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        /* TWO batches of up to 64 pairs per round, the top 128 entries of the stack: both batches' pops are one trip to memory and both
         * batches' records the next, where a batch at a time made them two trips per 64 pairs - the kernel waits for memory, not for
         * arithmetic, and beside the main walk it has one wave per SIMD and registers to spare (128 fit behind the main walk's 6 x 64).
         * Batch A's records are requested first, so its evaluation starts while B's are still on their way. */
        struct PairLd {
            int node, end, slot, kind;
            double4 src;
            double cx, cy, cz, len;
            int nsib, nchild, ntype, ncount;
        };
        struct PairOut {
            int push_sib, push_child, child_end, leaf0, leafn;
        };
        auto pair_pop = [&](int base, int np, PairLd &p) {
            p.node = -1, p.end = -1, p.slot = 0, p.kind = 0;
            if(lane < np) {
                const int4 pr = stack[base + lane];
                p.node = pr.x, p.end = pr.y, p.slot = pr.z, p.kind = pr.w;
            }
        };
        /* ONE body for both kinds of pair: a node record begins with {cofm, mass}, which is what a leaf particle's record is, so the
         * source is one load from either array and the rest of a node's record is requested right behind it */
        auto pair_fetch = [&](PairLd &p) {
            p.src = make_double4(0, 0, 0, 0);
            p.cx = p.cy = p.cz = p.len = 0;
            p.nsib = -1, p.nchild = -1, p.ntype = 0, p.ncount = 0;
            if(p.node >= 0) {
                const double4 *const sp4 = p.kind == 1 ? a.posm_leaf + p.node : reinterpret_cast<const double4 *>(a.nodeG + p.node);
                p.src = *sp4;
                if(p.kind == 0) {
                    const NodeG *const nd = a.nodeG + p.node;
                    p.len = nd->len;
                    p.cx = nd->center[0], p.cy = nd->center[1], p.cz = nd->center[2];
                    p.nsib = nd->sibling, p.nchild = nd->child, p.ntype = nd->type, p.ncount = nd->count;
                }
            }
        };
        auto pair_eval = [&](const PairLd &p, PairOut &o) {
            o.push_sib = -1, o.push_child = -1, o.child_end = -1, o.leaf0 = 0, o.leafn = 0;
            double ax = 0, ay = 0, az = 0, pot = 0;
            const bool live = p.node >= 0, isnode = live && p.kind == 0;
            double wraplim = 0, rcut2 = 0, rcuthl = 0, mlen2 = 0, bhlim = 0, inside = 0;
            if(isnode) {
                if(lean) {
                    /* this kernel is bound by the 16-byte pieces its lanes fetch from 64 different lines (a second record's worth of
                     * loads per pair: 3.5 -> 5.8 ms), not by arithmetic: the second half of the record - products of {mass, len} and
                     * the walk parameters - is recomputed with the expressions that filled it (checked record by record by
                     * fill_rcuthl_kernel), five pieces per pair instead of eight.  Without the interior flag: such a node's wrap
                     * only ever changes lanes that are discarded either way (the comment of fill_rcuthl_kernel) */
                    const LeanProducts lp = lean_products(p.src.w, p.len, a.rcut, a.Box);
                    rcuthl = lp.rcuthl, mlen2 = lp.mlen2, inside = lp.inside, wraplim = lp.wraplim, rcut2 = a.rcut2;
                    bhlim = bh_lvl[lean_level(rootlen, p.len) & 31];
                } else {
                    const NodeG *const nd = a.nodeG + p.node;
                    bhlim = nd->bhlim, mlen2 = nd->mlen2, inside = nd->inside, rcut2 = nd->rcut2, wraplim = nd->wraplim, rcuthl = nd->rcuthl;
                }
            }
            const double4 tg = tgt[p.slot];
            double dx = p.src.x - tg.x, dy = p.src.y - tg.y, dz = p.src.z - tg.z;
            double ux = p.cx - tg.x, uy = p.cy - tg.y, uz = p.cz - tg.z;
            double cmax = fmax(fmax(fabs(ux), fabs(uy)), fabs(uz));
            /* the wrap is the identity on every component it is not needed for: applying it per lane gives the main walk's bits.  A
             * leaf particle is always wrapped (gravshort2.hpp:290-304), a node unless it is interior (sign bit of wraplim) */
            const bool nwrap = isnode && __double2hiint(wraplim) >= 0 && cmax >= wraplim;
            if(nwrap || (live && !isnode)) {
                dx = wrapd(dx, a.Box, a.invBox);
                dy = wrapd(dy, a.Box, a.invBox);
                dz = wrapd(dz, a.Box, a.invBox);
            }
            if(nwrap) {
                ux = wrapd(ux, a.Box, a.invBox);
                uy = wrapd(uy, a.Box, a.invBox);
                uz = wrapd(uz, a.Box, a.invBox);
                cmax = fmax(fmax(fabs(ux), fabs(uy)), fabs(uz));
            }
            const double r2 = dx * dx + dy * dy + dz * dz;
            bool ev = live && !isnode;
            if(isnode) {
                const bool keep = !(r2 > rcut2 && cmax > rcuthl);
                const bool open = (!a.useBH && mlen2 > r2 * r2 * tg.w) || r2 < bhlim || cmax < inside;
                ev = keep && !open;
                if(keep && open) {
                    if(p.ntype == SHQ_NODE_NODE_TYPE) {
                        o.push_child = p.nchild;
                        o.child_end = p.nsib;
                    } else if(p.ntype == SHQ_PARTICLE_NODE_TYPE) { /* its particles become pairs of their own: one per lane and batch */
                        o.leaf0 = p.nchild;
                        o.leafn = p.ncount;
                    }
                }
                if(p.nsib != p.end && p.nsib >= 0)
                    o.push_sib = p.nsib;
            }
            if(ev) { /* CLAMP for the particle that is the target itself; a no-op on an accepted node's r^2 */
                apply_accn<POT, true>(tab, dx, dy, dz, r2, p.src.w, a, ax, ay, az, pot);
                atomicAdd(&acc[0][p.slot], ax);
                atomicAdd(&acc[1][p.slot], ay);
                atomicAdd(&acc[2][p.slot], az);
                if(POT)
                    atomicAdd(&acc[3][p.slot], pot);
                atomicAdd(&cntl[p.slot], 1);
            }
        };
        auto push_nodes = [&](int what, int what_end, int slot) {
            const unsigned long long m = shq_ballot(what >= 0);
            if(what >= 0)
                stack[sp + __popcll(m & below)] = make_int4(what, what_end, slot, 0);
            sp += __popcll(m);
        };
        /* a leaf's particles: the lane's first slot is the prefix sum of the counts below it, formed from the votes on the four
         * bits of the count (count <= 8) instead of one vote per particle */
        auto push_leaves = [&](int leaf0, int leafn, int slot) {
            if(shq_ballot(leafn > 0) != 0ull) {
                const unsigned long long b0 = shq_ballot((leafn & 1) != 0), b1 = shq_ballot((leafn & 2) != 0),
                                         b2 = shq_ballot((leafn & 4) != 0), b3 = shq_ballot((leafn & 8) != 0);
                const int off = __popcll(b0 & below) + 2 * __popcll(b1 & below) + 4 * __popcll(b2 & below) + 8 * __popcll(b3 & below);
                for(int k = 0; k < SHQ_NMAXCHILD; k++) {
                    if(shq_ballot(k < leafn) == 0ull)
                        break;
                    if(k < leafn)
                        stack[sp + off + k] = make_int4(leaf0 + k, -1, slot, 1);
                }
                sp += __popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2) + 8 * __popcll(b3);
            }
        };
        while(sp > 0) {
            const int npa = sp < 64 ? sp : 64;
            sp -= npa;
            const int npb = SHQ_PAIR_DEPTH2 ? (sp < 64 ? sp : 64) : 0;
            sp -= npb;
            PairLd pa, pb;
            PairOut oa, ob;
            pair_pop(sp + npb, npa, pa);
            if(npb > 0)
                pair_pop(sp, npb, pb);
            pair_fetch(pa);
            if(npb > 0)
                pair_fetch(pb);
            pair_eval(pa, oa);
            if(npb > 0)
                pair_eval(pb, ob);
            /* siblings first, then children, leaf particles on top: the stack stays depth-bounded */
            push_nodes(oa.push_sib, pa.end, pa.slot);
            if(npb > 0)
                push_nodes(ob.push_sib, pb.end, pb.slot);
            push_nodes(oa.push_child, oa.child_end, pa.slot);
            if(npb > 0)
                push_nodes(ob.push_child, ob.child_end, pb.slot);
            push_leaves(oa.leaf0, oa.leafn, pa.slot);
            if(npb > 0)
                push_leaves(ob.leaf0, ob.leafn, pb.slot);
            sp_high = sp > sp_high ? sp : sp_high;
            if(sp > a.sp_stack_cap - 1280) { /* never seen (SP_HIGHWATER is reported); loud and sticky if it happens: the pairs are dropped */
                if(lane == 0) {
                    agent_store(a.sp_flags + SP_OVERFLOW, 1);
                    atomicOr(a.sp_flags + SP_STICKY_ERROR, 1);
                }
                sp = 0;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        /* add to what the main walk stored, and tally */
        long long nint = 0;
        if(have) {
            nint = agent_load(&a.nint[pi]);
            if(cnt > 0) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                a.acc[3 * pi + 0] = agent_load(&a.acc[3 * pi + 0]) + acc[0][lane];
                a.acc[3 * pi + 1] = agent_load(&a.acc[3 * pi + 1]) + acc[1][lane];
                a.acc[3 * pi + 2] = agent_load(&a.acc[3 * pi + 2]) + acc[2][lane];
                if(POT)
                    a.pot[pi] = agent_load(&a.pot[pi]) + acc[3][lane];
                nint += (long long) cntl[lane];
                a.nint[pi] = (int32_t) nint;
            }
        }
        long long mn = have ? nint : 0x7fffffffffffll, mx = have ? nint : 0, sm = have ? nint : 0;
        for(int off = 32; off > 0; off >>= 1) {
            const long long o1 = __shfl_xor(mn, off), o2 = __shfl_xor(mx, off), o3 = __shfl_xor(sm, off);
            mn = o1 < mn ? o1 : mn;
            mx = o2 > mx ? o2 : mx;
            sm += o3;
        }
        w_sum += sm;
        w_min = mn < w_min ? mn : w_min;
        w_max = mx > w_max ? mx : w_max;
        /* the task is through: a second pass (the mop-up behind a live kernel that gave up somewhere) must not add its sums again */
        if(lane == 0)
            agent_store(&a.sp_count[task], (int32_t) SP_TASK_DONE);
        __builtin_amdgcn_wave_barrier();
    }
    if(lane == 0 && a.stats) {
        atomicAdd(&a.stats->ninteractions, (unsigned long long) w_sum);
        atomicMin(&a.stats->min_int, w_min);
        atomicMax(&a.stats->max_int, w_max);
    }
    if(lane == 0 && sp_high > 0) {
        atomicMax(a.sp_flags + SP_HIGHWATER, sp_high);
        atomicMax(a.sp_flags + SP_STICKY_HIGHWATER, sp_high);
    }
    if(lane == 0 && mopped > 0)
        atomicAdd(a.sp_flags + SP_MOPPED, mopped);
    if(a.sp_mop && blockIdx.x == 0 && threadIdx.x == 0)
        atomicAdd(a.sp_flags + SP_STICKY_RECOVERED, 1);
}

/* One register budget for both launch forms, 128 (waves_per_eu(4, 4) is there for the budget it implies): beside the main walk (workgroups
 * of four waves, one per SIMD) that is what is left of a SIMD's 512 registers behind the main walk's six waves of 64; behind it (workgroups
 * of eight waves, two per CU) the two batches per round need it as well.  The same code in both: the same sums in the same order. */
template <bool POT> __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void grav_pair_kernel_live(const WalkArgs a) { grav_pair_body<POT, 4>(a); }
template <bool POT> __global__ __launch_bounds__(64 * SHQ_PAIR_WAVES) __attribute__((amdgpu_waves_per_eu(SHQ_PAIR_EU, SHQ_PAIR_EU))) void grav_pair_kernel(const WalkArgs a)
{
    grav_pair_body<POT, SHQ_PAIR_WAVES>(a);
}

#pragma clang diagnostic pop

/* GravTreeOutput::postprocess, gravshort2.hpp:88-107 */
__global__ void grav_postprocess_kernel(const int32_t *targets, long long ntargets, const double4 *posm, double *acc,
                                        double *pot, double *treeacc, double G, double h, double cbrtrho0,
                                        int update_potential)
{
    const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= ntargets)
        return;
    const long long i = targets ? (long long) targets[t] : t;
    if(i < 0)
        return;
    const double a0 = acc[3 * i + 0] * G, a1 = acc[3 * i + 1] * G, a2 = acc[3 * i + 2] * G;
    acc[3 * i + 0] = a0;
    acc[3 * i + 1] = a1;
    acc[3 * i + 2] = a2;
    if(update_potential) {
        treeacc[3 * i + 0] = a0;
        treeacc[3 * i + 1] = a1;
        treeacc[3 * i + 2] = a2;
        const double m = posm[i].w;
        double p = pot[i];
        p += m / (h / 2.8);
        p -= 2.8372975 * pow(m, 2.0 / 3) * cbrtrho0;
        p *= G;
        pot[i] = p;
    }
}

/* grav_get_abs_accel, gravshort2.hpp:111-121 */
__global__ void oldacc_kernel(long long n, const double *treeacc, const double *gravpm, double *oldacc, double G)
{
#pragma clang fp contract(off) /* the same roundings as the readout kernel's and the walk prologue's copy of this sum */
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n)
        return;
    double s = 0;
    for(int j = 0; j < 3; j++) {
        const double ax = treeacc[3 * i + j] + gravpm[3 * i + j];
        s += ax * ax;
    }
    oldacc[i] = sqrt(s) / G;
}

/* the discard test compares with Rcut + len / 2 and the Barnes-Hut test with len^2 / theta^2: wave-uniform
 * expressions that would cost every visit VALU instructions (there is no scalar f64 ALU); they are stored with the
 * node whenever the walk parameters or the tree change */
/* INTERIOR nodes carry wraplim with the sign bit set (the walk reads the sign on the scalar pipe and skips the wrap vote, the group
 * walk takes |wraplim|).  A node is interior when its cell keeps M >= Rcut + 1.5 len from every face of the box.  Then for a target
 * inside the box and a coordinate k with |center_k - pos_k| > Box/2 - len/2 — the only case in which the periodic wrap is not the
 * identity on both displacement vectors — the lane is discarded with and without the wrap: without it |center_k - pos_k| > Box/2 - len/2
 * >= Rcut + len/2 and |cofm_k - pos_k| > Box/2 - len >= Rcut (Box >= 2 M + len); with it the wrapped displacement crosses a face, so
 * it is >= M + len/2 for the centre and >= M for the centre of mass, both beyond the same limits (shall_we_discard_node,
 * gravshort2.hpp:152-167).  Every other lane's arithmetic is untouched. */
__global__ void fill_rcuthl_kernel(NodeG *g, long long n, double rcut, double bh2, double Box, int *lean_bad)
{
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n) {
        const double len = g[i].len, hl = 0.5 * len;
        g[i].rcuthl = rcut + hl;
        g[i].rcut2 = rcut * rcut;
        g[i].bhlim = len * len / bh2; /* len^2 / r2 > theta^2 (gravshort2.hpp:179-182) as r2 < len^2 / theta^2 */
        double M = Box;
        for(int k = 0; k < 3; k++)
            M = fmin(M, fmin(g[i].center[k] - hl, Box - (g[i].center[k] + hl)));
        /* the root's side is 1.001 Box (forcetree.cpp): its limit would be negative — "always wrap" — and must not read as the
         * interior flag: zero says the same (the vote below is cmax >= wraplim) */
        const double wl = fmax(0.5 * Box - hl, 0.0);
        g[i].wraplim = (len > 0 && wl > 0 && M >= rcut + 1.5 * len) ? -wl : wl;
        /* may the pair kernel recompute this record's second half from {mass, len}?  Yes when len is rootlen 2^-level to the bit (the
         * Barnes-Hut limit then comes from a table by level, the expression above on the same number) and lean_products() returns
         * what is stored.  The record behind the pool is never visited. */
        if(lean_bad && i < n - 1) {
            const double rootlen = Box * 1.001;
            const int k = lean_level(rootlen, len);
            const LeanProducts lp = lean_products(g[i].mass, len, rcut, Box);
            const bool ok = len > 0 && k >= 0 && k < 32 && ldexp(rootlen, -k) == len && lp.rcuthl == g[i].rcuthl &&
                            lp.mlen2 == g[i].mlen2 && lp.inside == g[i].inside && lp.wraplim == wl;
            if(!ok)
                *lean_bad = 1;
        }
    }
}

__global__ void stats_init_kernel(GravStatsDev *s)
{
    s->ninteractions = 0;
    s->nvisited = 0;
    s->nwave_applies = 0;
    s->nwave_node_applies = 0;
    s->nnode_interactions = 0;
    s->min_int = 0x7fffffffffffll;
    s->max_int = 0;
    for(int b = 0; b < 8; b++)
        s->hist_visit[b] = s->hist_node[b] = s->hist_leaf[b] = 0;
    for(int b = 0; b < 4; b++)
        s->lonely[b] = 0;
}

/* BH: pure Barnes-Hut opening angle (TreeUseBH, the seeding walk before any acceleration exists) as its
 * own instantiation: the relative criterion drops out at compile time, and a profile lists the seeding
 * walk and the production walk as two kernels. */
template <bool POT, bool PREFETCH, int LEAFB, bool BH>
void launch_variant_bh(int stats, bool persist, dim3 grid, dim3 block, hipStream_t stream, const WalkArgs &a, size_t dyn_lds = 0)
{
    constexpr bool RO = !PREFETCH && LEAFB == 2 && !BH; /* the fused readout and the sparse-subtree hand-over exist for the production variant only */
    if(a.pm_mesh && a.sp_items && RO && !stats && persist && block.x == 512)
        grav_walk_exact_kernel<POT, PREFETCH, LEAFB, 0, BH, false, RO, RO, RO, RO><<<grid, block, dyn_lds, stream>>>(a);
    else if(a.sp_items && RO && !stats && persist && block.x == 512)
        grav_walk_exact_kernel<POT, PREFETCH, LEAFB, 0, BH, false, RO, RO, false, RO><<<grid, block, dyn_lds, stream>>>(a);
    else if(a.pm_mesh && RO && !stats && persist && block.x == 512)
        grav_walk_exact_kernel<POT, PREFETCH, LEAFB, 0, BH, false, RO, RO, RO><<<grid, block, dyn_lds, stream>>>(a);
    else if(a.pm_mesh && RO && !stats && !persist)
        grav_walk_exact_kernel<POT, PREFETCH, LEAFB, 0, BH, false, false, false, RO><<<grid, block, 0, stream>>>(a);
    else if(stats == 2 && POT && !PREFETCH && LEAFB == 2)
        grav_walk_exact_kernel<POT, PREFETCH, LEAFB, (POT && !PREFETCH && LEAFB == 2) ? 2 : 1, BH><<<grid, block, 0, stream>>>(a);
    else if(stats)
        grav_walk_exact_kernel<POT, PREFETCH, LEAFB, 1, BH><<<grid, block, 0, stream>>>(a);
    else if(persist && !PREFETCH && LEAFB == 2 && block.x == 512)
        grav_walk_exact_kernel<POT, PREFETCH, LEAFB, 0, BH, false, (!PREFETCH && LEAFB == 2), (!PREFETCH && LEAFB == 2)><<<grid, block, dyn_lds, stream>>>(a);
    else if(persist && !PREFETCH && LEAFB == 2)
        grav_walk_exact_kernel<POT, PREFETCH, LEAFB, 0, BH, false, (!PREFETCH && LEAFB == 2)><<<grid, block, dyn_lds, stream>>>(a);
    else
        grav_walk_exact_kernel<POT, PREFETCH, LEAFB, 0, BH><<<grid, block, 0, stream>>>(a);
}

template <bool POT, bool PREFETCH, int LEAFB>
void launch_variant(int stats, bool persist, dim3 grid, dim3 block, hipStream_t stream, const WalkArgs &a, size_t dyn_lds = 0)
{
    if(a.useBH)
        launch_variant_bh<POT, PREFETCH, LEAFB, true>(stats, persist, grid, block, stream, a, dyn_lds);
    else
        launch_variant_bh<POT, PREFETCH, LEAFB, false>(stats, persist, grid, block, stream, a, dyn_lds);
}

} // namespace

void shq_launch_stats_init(shq_context *ctx) { stats_init_kernel<<<1, 1, 0, ctx->stream>>>(ctx->gstats.ptr); }

/* the per-node fields that depend on the walk parameters (rcuthl, bhlim), refilled when Rcut, the opening angle or the tree change */
void shq_fill_node_walk_params(shq_context *ctx, const shq_grav_params *p)
{
    if((ctx->node_rcut != p->Rcut || ctx->node_bh2 != p->BHOpeningAngle2) && ctx->numnodes > 0) {
        const long long n = ctx->numnodes + 1;
        int *lean_bad = nullptr;
        if(ctx->node_lean_bad.reserve(16) == SHQ_OK && hipMemsetAsync(ctx->node_lean_bad.ptr, 0, sizeof(int) * 16, ctx->stream) == hipSuccess)
            lean_bad = ctx->node_lean_bad.ptr;
        ctx->node_lean_checked = lean_bad != nullptr;
        fill_rcuthl_kernel<<<dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, ctx->stream>>>(ctx->nodeG.ptr, n, p->Rcut, p->BHOpeningAngle2, p->BoxSize, lean_bad);
        ctx->node_rcut = p->Rcut;
        ctx->node_bh2 = p->BHOpeningAngle2;
    }
}

static void fill_walk_args(shq_context *ctx, const shq_grav_params *p, WalkArgs &a)
{
    shq_fill_node_walk_params(ctx, p);
    a.nodeG = ctx->nodeG.ptr;
    a.posm = ctx->posm.ptr;
    a.posm_leaf = ctx->posm_leaf.ptr;
    a.oldacc = ctx->oldacc.ptr;
    a.targets = nullptr;
    a.qstart = nullptr;
    a.acc = ctx->acc.ptr;
    a.pot = ctx->pot.ptr;
    a.nint = ctx->nint.ptr;
    a.stats = ctx->gstats.ptr;
    a.ntargets = 0;
    a.root = ctx->root;
    a.Box = p->BoxSize;
    a.invBox = 1.0 / p->BoxSize;
    a.halfBox = 0.5 * p->BoxSize;
    a.rcut = p->Rcut;
    a.rcut2 = p->Rcut * p->Rcut;
    a.h = p->ForceSoftening;
    a.h2 = a.h * a.h;
    a.h_inv = 1.0 / a.h;
    a.h3_inv = 1.0 / a.h / a.h / a.h;
    a.inv_celldx = 1.0 / (p->cellsize * p->dx);
    a.errtol = p->ErrTolForceAcc;
    a.bh2 = p->BHOpeningAngle2;
    a.useBH = p->TreeUseBH;
    a.xcdK = (unsigned) ctx->xcd_k;
    a.stats_guard = ctx->stats_guard;
    a.tab_f = ctx->gravtab.ptr;
    a.tab_p = ctx->gravtab.ptr + SHQ_NGRAVTAB;
    a.task_counters = nullptr;
    a.scrub = nullptr;
    a.scrub_n16 = 0;
    a.scrub_per_task = 0;
    a.pm_mesh = nullptr;
    a.pmN = a.pmzp = 0;
    a.pmcell = a.pmffac = a.G = 0;
    a.treeacc = nullptr;
    a.pflags = nullptr;
    a.gravpm = a.pmpot = a.oldacc_out = nullptr;
    a.sp_items = nullptr;
    a.sp_count = nullptr;
    a.sp_stack = nullptr;
    a.sp_flags = nullptr;
    a.sp_stack_cap = SHQ_SPARSE_STACK;
    a.sp_spin_max = SHQ_SPARSE_SPIN;
    a.sp_mop = 0;
    a.sp_lean_bad = nullptr;
    a.sp_live = 0;
    a.sp_task = nullptr;
    a.nwaves = 0;
    a.task_run_log2 = 7;
}

/* secondary (GHOSTS) walk over query arrays already on the device: a carries the query positions as posm, the
 * queries' OldAcc, qstart, and the result arrays */
int shq_launch_grav_walk_ghosts(shq_context *ctx, const shq_grav_params *p, const double4 *d_qpos, const double *d_qoldacc,
                                const int32_t *d_qstart, int64_t nq, double *d_acc, double *d_pot, int32_t *d_nint, int update_potential)
{
    SHQ_CHECK(ctx->have_tree, SHQ_ERR_STATE, "secondary walk: no tree");
    SHQ_CHECK(p->ForceSoftening > 0 && p->cellsize > 0 && p->dx > 0, SHQ_ERR_INVALID, "grav params: softening/cellsize/dx must be > 0");
    SHQ_TRY(ctx->gravtab.reserve(2 * SHQ_NGRAVTAB));
    SHQ_TRY(ctx->gstats.reserve(1));
    SHQ_HIP(hipMemcpyAsync(ctx->gravtab.ptr, p->shortrange_table, sizeof(float) * SHQ_NGRAVTAB, hipMemcpyHostToDevice, ctx->stream));
    SHQ_HIP(hipMemcpyAsync(ctx->gravtab.ptr + SHQ_NGRAVTAB, p->shortrange_table_potential, sizeof(float) * SHQ_NGRAVTAB,
                           hipMemcpyHostToDevice, ctx->stream));
    stats_init_kernel<<<1, 1, 0, ctx->stream>>>(ctx->gstats.ptr);
    if(nq == 0)
        return SHQ_OK;
    WalkArgs a;
    fill_walk_args(ctx, p, a);
    a.posm = d_qpos;
    a.oldacc = d_qoldacc;
    a.qstart = d_qstart;
    a.acc = d_acc;
    a.pot = d_pot;
    a.nint = d_nint;
    a.ntargets = nq;
    const long long nwaves = (nq + 63) / 64;
    const dim3 grid((unsigned) ((nwaves + 3) / 4)), block(256);
    if(update_potential) {
        if(a.useBH)
            grav_walk_exact_kernel<true, false, 2, 0, true, true><<<grid, block, 0, ctx->stream>>>(a);
        else
            grav_walk_exact_kernel<true, false, 2, 0, false, true><<<grid, block, 0, ctx->stream>>>(a);
    } else {
        if(a.useBH)
            grav_walk_exact_kernel<false, false, 2, 0, true, true><<<grid, block, 0, ctx->stream>>>(a);
        else
            grav_walk_exact_kernel<false, false, 2, 0, false, true><<<grid, block, 0, ctx->stream>>>(a);
    }
    SHQ_HIP(hipGetLastError());
    return SHQ_OK;
}

/* Can the exact walk over these targets carry the PM readout in its prologue?  Every particle must be a target exactly once (a PM step:
 * all time bins active), the walk must be the production instantiation with the relative criterion, and the mesh must be 32-bit
 * addressable. */
bool shq_walk_can_fuse_readout_pre(shq_context *ctx, const shq_grav_params *p, int64_t ntargets)
{ /* what can be known before the PM has allocated its mesh */
    const long long nwaves = (ntargets + 63) / 64, blocks = (nwaves + 3) / 4;
    const bool persist = ctx->walk_persist && (ctx->walk_persist == 2 || blocks > (long long) ctx->num_cus * 8);
    return ntargets > 0 && ntargets == ctx->numpart && ctx->numpart == ctx->nlocal && !ctx->walk_stats && ctx->walk_variant == 3 && !p->TreeUseBH &&
           (!persist || ctx->walk_ring) && !getenv("SHQ_WALK_BLOCKS_PER_CU") && ctx->treeacc.ptr && ctx->have_tree;
}

bool shq_walk_can_fuse_readout(shq_context *ctx, const shq_grav_params *p, const int32_t *d_active, int64_t ntargets, int64_t first)
{
    /* targets: every particle once — no list, or the library's own list of all the tree's particles in leaf order */
    return (!d_active || d_active == ctx->tree_targets.ptr) && first == 0 && shq_walk_can_fuse_readout_pre(ctx, p, ntargets) && ctx->mesh.ptr && ctx->mesh_words > 0 &&
           ctx->mesh_words < (1ull << 29); /* 32-bit byte offsets */
}

/* The sparse-subtree buffers for launches of up to nwaves 64-target tasks: the noted subtrees and flags per task, the pair stacks
 * of as many waves as the larger of the two pair-kernel grids holds, the status words and their pinned host copy.  Called when a tree
 * is installed (for the tree's own particles), so that the walk's launch path neither allocates nor - hipFree is a device-wide
 * synchronisation - waits; a launch over more targets than that grows them once. */
int shq_walk_reserve_sparse(shq_context *ctx, long long nwaves)
{
    if(nwaves < 1)
        nwaves = 1;
    const long long stack_waves = (long long) ctx->num_cus * (SHQ_PAIR_WG_PER_CU * SHQ_PAIR_WAVES > 16 ? SHQ_PAIR_WG_PER_CU * SHQ_PAIR_WAVES : 16);
    SHQ_TRY(ctx->sp_items.reserve((size_t) nwaves * SHQ_SPARSE_CAP));
    SHQ_TRY(ctx->sp_count.reserve((size_t) nwaves));
    if(ctx->sp_stack_cap <= 0)
        ctx->sp_stack_cap = SHQ_SPARSE_STACK;
    if(ctx->sp_spin_max == 0)
        ctx->sp_spin_max = SHQ_SPARSE_SPIN;
    SHQ_TRY(ctx->sp_stack.reserve((size_t) stack_waves * (size_t) ctx->sp_stack_cap));
    if(!ctx->sp_flags.ptr) {
        SHQ_TRY(ctx->sp_flags.reserve(32));
        SHQ_HIP(hipMemsetAsync(ctx->sp_flags.ptr, 0, sizeof(int) * 32, ctx->stream));
    }
    if(!ctx->sp_host.ptr) {
        SHQ_TRY(ctx->sp_host.reserve(8));
        for(int k = 0; k < 8; k++)
            ctx->sp_host.ptr[k] = 0;
    }
    return SHQ_OK;
}

const int *shq_walk_error_word(shq_context *ctx) { return ctx->sp_flags.ptr ? ctx->sp_flags.ptr + SP_STICKY_ERROR : nullptr; }

/* at tree installation: the buffers of a walk over all of the context's particles, if such a walk would take the sparse path */
int shq_walk_prereserve(shq_context *ctx)
{
    const long long nwaves = (ctx->numpart + 63) / 64;
    const bool persist = ctx->walk_persist == 2 || (ctx->walk_persist && (nwaves + 3) / 4 > (long long) ctx->num_cus * 8);
    if(persist && ctx->walk_ring && ctx->walk_sparse && ctx->walk_variant == 3)
        return shq_walk_reserve_sparse(ctx, nwaves);
    return SHQ_OK;
}

/* The pair kernel's sticky error word, as the last completed launch left it in pinned host memory.  sync: wait for everything queued
 * on the stream first (the calls that synchronise anyway); without it the check costs nothing and sees the launches that have
 * completed so far - a resident loop learns of a failed step at a later step's entry, and at its final synchronisation at the latest.
 * The report clears the word: the caller has been told (the shenqi-side shim turns the code into endrun, treewalk2.cuh:351-353). */
int shq_walk_check_status(shq_context *ctx, bool sync)
{
    if(!ctx->sp_host.ptr)
        return SHQ_OK;
    if(sync && ctx->sp_check_pending) {
        SHQ_HIP(hipStreamSynchronize(ctx->stream));
        ctx->sp_check_pending = false;
    }
    const volatile int *h = ctx->sp_host.ptr;
    if(h[0] != 0) {
        SHQ_HIP(hipStreamSynchronize(ctx->stream)); /* copies of later launches still in flight would put the word back */
        SHQ_HIP(hipMemsetAsync(ctx->sp_flags.ptr + SP_STICKY_ERROR, 0, sizeof(int), ctx->stream));
        ctx->sp_host.ptr[0] = 0;
        ctx->sp_check_pending = false;
        shq_set_error("grav walk: a pair stack of the sparse-subtree kernel overflowed in an earlier launch and its pairs were dropped: the accelerations "
                      "of that launch are incomplete (SHQ_WALK_SPARSE=0 avoids the kernel)");
        return SHQ_ERR_DEVICE;
    }
    return SHQ_OK;
}

extern "C" int shq_set_walk_debug(shq_context *ctx, int pair_spin_max, int pair_stack_cap)
{
    SHQ_CHECK(ctx, SHQ_ERR_INVALID, "null context");
    SHQ_CHECK(pair_spin_max >= 0 && (pair_stack_cap == 0 || (pair_stack_cap >= 1344 && pair_stack_cap <= SHQ_SPARSE_STACK)), SHQ_ERR_INVALID,
              "walk debug: pair_spin_max >= 0 (0: default), pair_stack_cap 0 (default) or 1344..%d", SHQ_SPARSE_STACK);
    ctx->sp_spin_max = pair_spin_max > 0 ? (unsigned) pair_spin_max : SHQ_SPARSE_SPIN;
    ctx->sp_stack_cap = pair_stack_cap > 0 ? pair_stack_cap : SHQ_SPARSE_STACK; /* never above what was reserved */
    return SHQ_OK;
}

extern "C" int shq_walk_pair_status(shq_context *ctx, int64_t *recovered_launches, int64_t *stack_high_water, int64_t *last_mopped_tasks)
{
    SHQ_CHECK(ctx, SHQ_ERR_INVALID, "null context");
    int w[16] = {};
    if(ctx->sp_flags.ptr) {
        SHQ_HIP(hipSetDevice(ctx->device));
        SHQ_HIP(hipMemcpyAsync(w, ctx->sp_flags.ptr, sizeof(w), hipMemcpyDeviceToHost, ctx->stream));
        SHQ_HIP(hipStreamSynchronize(ctx->stream));
    }
    if(recovered_launches)
        *recovered_launches = w[SP_STICKY_RECOVERED];
    if(stack_high_water)
        *stack_high_water = w[SP_STICKY_HIGHWATER];
    if(last_mopped_tasks)
        *last_mopped_tasks = w[SP_MOPPED];
    return shq_walk_check_status(ctx, true);
}

int shq_launch_grav_walk(shq_context *ctx, const shq_grav_params *p, const int32_t *d_active, int64_t ntargets,
                         int update_potential, int walk_mode, int64_t first)
{
    if(walk_mode == SHQ_WALK_AUTO)
        walk_mode = (d_active && ntargets * 10 < ctx->ntreeparts) ? SHQ_WALK_GROUP : SHQ_WALK_EXACT;
    ctx->last_walk_mode = walk_mode;
    if(walk_mode == SHQ_WALK_GROUP)
        return shq_launch_grav_walk_group(ctx, p, d_active, ntargets, update_potential, first);
    SHQ_CHECK(first >= 0 && (first == 0 || !d_active) && (first + ntargets <= ctx->numpart || (d_active && ctx->allow_padding)), SHQ_ERR_INVALID,
              "grav walk: bad target range [%ld, +%ld)", (long) first, (long) ntargets);
    SHQ_CHECK(ctx->have_parts && ctx->have_tree, SHQ_ERR_STATE, "grav walk: particles and tree must be uploaded first");
    SHQ_CHECK(walk_mode == SHQ_WALK_EXACT, SHQ_ERR_INVALID, "unknown walk_mode %d", walk_mode);
    SHQ_CHECK(p->ForceSoftening > 0 && p->cellsize > 0 && p->dx > 0, SHQ_ERR_INVALID, "grav params: softening/cellsize/dx must be > 0");
    SHQ_CHECK(ntargets >= 0 && (ntargets <= ctx->numpart || (d_active && ctx->allow_padding)), SHQ_ERR_INVALID, "grav walk: ntargets %ld out of range", (long) ntargets);
    SHQ_TRY(ctx->gravtab.reserve(2 * SHQ_NGRAVTAB));
    SHQ_TRY(ctx->gstats.reserve(1));
    SHQ_HIP(hipMemcpyAsync(ctx->gravtab.ptr, p->shortrange_table, sizeof(float) * SHQ_NGRAVTAB, hipMemcpyHostToDevice, ctx->stream));
    SHQ_HIP(hipMemcpyAsync(ctx->gravtab.ptr + SHQ_NGRAVTAB, p->shortrange_table_potential, sizeof(float) * SHQ_NGRAVTAB,
                           hipMemcpyHostToDevice, ctx->stream));
    if(first == 0)
        stats_init_kernel<<<1, 1, 0, ctx->stream>>>(ctx->gstats.ptr);
    if(ntargets == 0)
        return SHQ_OK;

    WalkArgs a;
    fill_walk_args(ctx, p, a);
    a.targets = d_active;
    a.ntargets = ntargets;
    /* a range of targets: target t of the launch is particle first + t (the leaf copies and the tree are not indexed by target) */
    a.posm += first;
    a.oldacc += first;
    a.acc += 3 * first;
    a.pot += first;
    a.nint += first;

    /* 4 waves per workgroup: measured 86.9 / 55.4 / 46.7 / 50.5 ms for 64 / 128 / 256 / 512 threads (the LDS window
     * table is per workgroup; larger groups wait for their slowest wave) */
    const int threads = 256;
    const long long nwaves = (ntargets + 63) / 64;
    const long long blocks = (nwaves + (threads / 64) - 1) / (threads / 64);
    SHQ_CHECK(blocks < (1ll << 31), SHQ_ERR_INVALID, "grav walk: too many targets for one launch");
    /* tuning knobs (diagnostic): SHQ_WALK_VARIANT = 0..3 selects prefetch/leaf-batch; the
     * default is the measured-fastest one. */
    const int variant = ctx->walk_variant;
    const int stats = ctx->walk_stats;
    /* persistent waves (the default for the production variant without counters): as many workgroups as fit the chip at
     * 8 per CU, tasks from eight per-XCD counters */
    static const int bpc_env = getenv("SHQ_WALK_BLOCKS_PER_CU") ? atoi(getenv("SHQ_WALK_BLOCKS_PER_CU")) : 8;
    const int bpc = bpc_env >= 1 && bpc_env <= 8 ? bpc_env : 8; /* resident workgroups per CU (diagnostic: fewer leave room for other streams) */
    const bool persist = ctx->walk_persist && !stats && variant == 3 && (ctx->walk_persist == 2 || blocks > (long long) ctx->num_cus * bpc);
    /* leaf ring: workgroups of 8 waves (4 per CU: one window table per 8 waves leaves the LDS room for the rings) */
    const bool ring = persist && ctx->walk_ring && bpc == 8;
    /* sparse subtrees to the pair kernel (SHQ_WALK_SPARSE): the production launch with the relative criterion only */
    const bool sparse = ring && ctx->walk_sparse && !p->TreeUseBH;
    /* SHQ_WALK_OVERLAP: the pair kernel beside the main walk instead of behind it — the main walk saturates the vector ALUs with six
     * waves per SIMD as well as with eight (29.6 against 29.9 ms), the pair kernel waits for memory: three workgroups of the main walk
     * and one of the pair kernel per CU, the pair kernel on the second stream taking a task when its flag is up */
    const bool live = sparse && ctx->walk_overlap && !ctx->pm_overlap && ctx->stream_pair && ctx->ev_pair_fork && ctx->ev_pair_join && !stats &&
                      (ctx->walk_overlap == 2 || nwaves >= (long long) ctx->num_cus * 32 * 4); /* 2: whatever the size (tests) */
    const long long pair_blocks = (long long) ctx->num_cus * (live ? 4 : SHQ_PAIR_WG_PER_CU); /* live: of four waves */
    const long long mop_blocks = (long long) ctx->num_cus * SHQ_PAIR_WG_PER_CU;                /* the pass behind a live pair kernel */
    if(sparse) {
        /* no-ops after shq_walk_reserve_sparse (tree installation) unless the launch is larger than the tree's own particles */
        SHQ_TRY(shq_walk_reserve_sparse(ctx, nwaves));
        SHQ_HIP(hipMemsetAsync(ctx->sp_flags.ptr, 0, sizeof(int) * SP_STICKY_ERROR, ctx->stream));
        SHQ_HIP(hipMemsetAsync(ctx->sp_flags.ptr + SP_TASK_COUNTER, 0, sizeof(int) * 16, ctx->stream));
        a.sp_items = ctx->sp_items.ptr;
        a.sp_count = ctx->sp_count.ptr;
        a.sp_stack = ctx->sp_stack.ptr;
        a.sp_flags = ctx->sp_flags.ptr;
        a.sp_stack_cap = ctx->sp_stack_cap;
        a.sp_spin_max = ctx->sp_spin_max;
        a.sp_lean_bad = ctx->node_lean_checked && ctx->walk_sparse != 2 ? ctx->node_lean_bad.ptr : nullptr;
        if(live) {
            SHQ_HIP(hipMemsetAsync(ctx->sp_count.ptr, 0xff, sizeof(int32_t) * (size_t) nwaves, ctx->stream));
            a.sp_live = 1;
        }
        a.sp_task = reinterpret_cast<unsigned int *>(ctx->sp_flags.ptr + SP_TASK_COUNTER);
        a.nwaves = nwaves;
    }
    long long launch_blocks = blocks;
    size_t dyn_lds = 0;
    if(persist) {
        SHQ_TRY(ctx->walk_tasks.reserve(8 * 16));
        SHQ_HIP(hipMemsetAsync(ctx->walk_tasks.ptr, 0, sizeof(unsigned int) * 8 * 16, ctx->stream));
        a.task_counters = ctx->walk_tasks.ptr;
        a.nwaves = nwaves;
        int lg = 0;
        while((1u << lg) < 4u * (a.xcdK ? a.xcdK : 1u))
            lg++;
        a.task_run_log2 = lg;
        /* (Handing the runs out longest-first, by the interaction counts of the previous walk — list scheduling on the measured task
         * costs promises 1.002 x the ideal makespan against 1.026 x in launch order, tools/walk_task_costs.py — was built and
         * measured: 34.3 against 34.0 ms run by run, 35.2 task by task.  Runs that follow each other in the box share their
         * nodes in the scalar cache, and the tail is short anyway: the last waves have their SIMD to themselves.) */
        const long long wpb = ring ? 8 : 4;                                   /* waves per workgroup */
        const long long need = (nwaves + wpb - 1) / wpb, resident = (long long) ctx->num_cus * (ring ? (live ? 3 : 4) : bpc);
        launch_blocks = need < resident ? need : resident;
        /* fewer than 8 resident workgroups per CU are enforced through the LDS allocation (160 KB per CU), not left to the dispatcher */
        if(bpc < 8)
            dyn_lds = (size_t) ((160 * 1024 / bpc - 16384) / 1024 * 1024 - 1024);
        if(ring)
            dyn_lds = sizeof(double4) * SHQ_LEAF_RING * 8;
    }
    /* clear the PM mesh for the next deposit in this walk's shadow: the last shq_pm_run is through with it (same stream), nothing
     * else is known to want it (pm_keep copies what it keeps), and the walk is large enough for a task's share to be a few stores */
    bool scrubbed = false, swap_meshes = false;
    if(ctx->pm_scrub && !stats && variant == 3 && ctx->mesh.ptr && ctx->mesh_words > 0 && !ctx->mesh_zeroed && !ctx->pm_overlap) {
        const long long n16 = (long long) (ctx->mesh_words / 2);
        const long long per = ((n16 + nwaves - 1) / nwaves + 63) / 64 * 64;
        if(ctx->mesh_words % 2 == 0 && per <= 4096) {
            a.scrub = (uint4 *) ctx->mesh.ptr;
            a.scrub_n16 = n16;
            a.scrub_per_task = (int) per;
            scrubbed = true;
        }
    }
    /* shq_treepm_step: the PM's readout and the OldAcc refresh ride in the task prologue (it checked that this launch can carry them) */
    if(ctx->fuse_readout) {
        SHQ_CHECK(shq_walk_can_fuse_readout(ctx, p, d_active, ntargets, first), SHQ_ERR_STATE, "fused readout requested for a launch that cannot carry it");
        a.pm_mesh = ctx->mesh.ptr;
        a.pmN = ctx->pm_nmesh;
        a.pmzp = ctx->pm_zp;
        a.pmcell = ctx->fuse_cell;
        a.pmffac = ctx->fuse_ffac;
        a.G = ctx->fuse_G;
        a.treeacc = ctx->treeacc.ptr;
        a.pflags = ctx->pflags.ptr;
        a.gravpm = ctx->gravpm.ptr;
        a.pmpot = ctx->pmpot.ptr;
        a.oldacc_out = ctx->oldacc.ptr;
        /* the mesh is being read: the zeros go to the second mesh, and the two change places after the launch */
        if(scrubbed && ctx->mesh_alt.reserve(ctx->mesh_words) == SHQ_OK) {
            a.scrub = (uint4 *) ctx->mesh_alt.ptr;
            swap_meshes = true;
        } else {
            a.scrub = nullptr;
            a.scrub_per_task = 0;
            scrubbed = false;
        }
    }
    const dim3 grid((unsigned) launch_blocks), block(ring ? 512 : threads);
    SHQ_HIP(hipEventRecord(ctx->ev_begin[SHQ_NTIMERS - 1], ctx->stream));
    if(live) /* what precedes the walk on this stream (the PM, the flags' reset) also precedes the pair kernel on the other */
        SHQ_HIP(hipEventRecord(ctx->ev_pair_fork, ctx->stream));
    if(update_potential) {
        switch(variant) {
        case 0: launch_variant<true, false, 4>(stats, persist, grid, block, ctx->stream, a); break;
        case 1: launch_variant<true, false, 2>(stats, persist, grid, block, ctx->stream, a); break;
        case 2: launch_variant<true, false, 4>(stats, persist, grid, block, ctx->stream, a); break;
        default: launch_variant<true, false, 2>(stats, persist, grid, block, ctx->stream, a, dyn_lds); break;
        }
    } else {
        switch(variant) {
        case 0: launch_variant<false, false, 4>(stats, persist, grid, block, ctx->stream, a); break;
        case 1: launch_variant<false, false, 2>(stats, persist, grid, block, ctx->stream, a); break;
        case 2: launch_variant<false, false, 4>(stats, persist, grid, block, ctx->stream, a); break;
        default: launch_variant<false, false, 2>(stats, persist, grid, block, ctx->stream, a, dyn_lds); break;
        }
    }
    SHQ_HIP(hipGetLastError());
    if(sparse) { /* the noted subtrees, one lane per (target, node) pair; inside the walk's timer */
        /* live: submitted AFTER the main walk, which never waits for it — if the two are run one after the other after all (a
         * profiler collecting counters serialises dispatches) the pair kernel finds every flag up */
        hipStream_t ps = live ? ctx->stream_pair : ctx->stream;
        if(live)
            SHQ_HIP(hipStreamWaitEvent(ps, ctx->ev_pair_fork, 0));
        const dim3 pg((unsigned) pair_blocks);
        if(live) { /* workgroups of four waves, one per SIMD: 6 x 64 registers of the main walk + 80 */
            if(update_potential)
                grav_pair_kernel_live<true><<<pg, dim3(256), 0, ps>>>(a);
            else
                grav_pair_kernel_live<false><<<pg, dim3(256), 0, ps>>>(a);
        } else if(update_potential)
            grav_pair_kernel<true><<<pg, dim3(64 * SHQ_PAIR_WAVES), 0, ps>>>(a);
        else
            grav_pair_kernel<false><<<pg, dim3(64 * SHQ_PAIR_WAVES), 0, ps>>>(a);
        SHQ_HIP(hipGetLastError());
        if(live) {
            SHQ_HIP(hipEventRecord(ctx->ev_pair_join, ps));
            SHQ_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_pair_join, 0));
            /* co-residency of the two kernels is what HIP does not promise (a tool that serialises queues, the pair stream dispatched
             * first): a live wave that gave up waiting leaves its tasks to this pass behind both kernels - it returns at once when no
             * wave gave up, and otherwise walks every task not marked done in the order of the pair kernel run behind the walk */
            WalkArgs m = a;
            m.sp_live = 0;
            m.sp_mop = 1;
            const dim3 mg((unsigned) mop_blocks);
            if(update_potential)
                grav_pair_kernel<true><<<mg, dim3(64 * SHQ_PAIR_WAVES), 0, ctx->stream>>>(m);
            else
                grav_pair_kernel<false><<<mg, dim3(64 * SHQ_PAIR_WAVES), 0, ctx->stream>>>(m);
            SHQ_HIP(hipGetLastError());
        }
        /* the sticky words follow every launch to pinned host memory: the entry points read them there without a round trip */
        SHQ_HIP(hipMemcpyAsync(ctx->sp_host.ptr, ctx->sp_flags.ptr + SP_STICKY_ERROR, sizeof(int) * 8, hipMemcpyDeviceToHost, ctx->stream));
        ctx->sp_check_pending = true;
    }
    SHQ_HIP(hipEventRecord(ctx->ev_end[SHQ_NTIMERS - 1], ctx->stream));
    if(swap_meshes) {
        std::swap(ctx->mesh.ptr, ctx->mesh_alt.ptr);
        std::swap(ctx->mesh.cap, ctx->mesh_alt.cap);
    }
    if(scrubbed)
        ctx->mesh_zeroed = true;
    return SHQ_OK;
}

int shq_launch_grav_postprocess(shq_context *ctx, const shq_grav_params *p, const int32_t *d_active, int64_t ntargets,
                                int update_potential, int64_t first)
{
    if(ntargets == 0)
        return SHQ_OK;
    const int threads = 256;
    const long long blocks = (ntargets + threads - 1) / threads;
    grav_postprocess_kernel<<<dim3((unsigned) blocks), dim3(threads), 0, ctx->stream>>>(
        d_active, ntargets, ctx->posm.ptr + first, ctx->acc.ptr + 3 * first, ctx->pot.ptr + first, ctx->treeacc.ptr + 3 * first, p->G,
        p->ForceSoftening, p->cbrtrho0, update_potential);
    SHQ_HIP(hipGetLastError());
    return SHQ_OK;
}

/* ---- sampled direct summation (a checker utility: force_direct / grav_force of the reference's own gravity test,
 * tests/test_gravity.cpp:41-76,121-143): the acceleration at `ns` sample positions from the first `nsrc` resident particles and
 * their (2 repeat + 1)^3 periodic images, spline-softened below h.  One workgroup per (sample, source chunk). */
namespace {
__global__ __launch_bounds__(256) void direct_sample_kernel(const double *__restrict__ tpos, int ns, const double4 *__restrict__ posm,
                                                            long long nsrc, long long chunk, double Box, double G, double h, int repeat,
                                                            double *__restrict__ out)
{
    const int t = blockIdx.x;
    const long long j0 = (long long) blockIdx.y * chunk, j1 = j0 + chunk < nsrc ? j0 + chunk : nsrc;
    const double px = tpos[3 * t], py = tpos[3 * t + 1], pz = tpos[3 * t + 2];
    const double h_inv = 1.0 / h, h3_inv = h_inv * h_inv * h_inv;
    double ax = 0, ay = 0, az = 0;
    for(long long j = j0 + threadIdx.x; j < j1; j += blockDim.x) {
        const double4 q = posm[j];
        for(int xx = -repeat; xx <= repeat; xx++)
            for(int yy = -repeat; yy <= repeat; yy++)
                for(int zz = -repeat; zz <= repeat; zz++) {
                    const double dx = Box * xx + px - q.x, dy = Box * yy + py - q.y, dz = Box * zz + pz - q.z;
                    const double r2 = dx * dx + dy * dy + dz * dz;
                    if(r2 == 0)
                        continue;
                    const double r = sqrt(r2);
                    double fac = 1 / (r2 * r);
                    if(r < h) {
                        const double u = r * h_inv;
                        if(u < 0.5)
                            fac = h3_inv * (10.666666666667 + u * u * (32.0 * u - 38.4));
                        else
                            fac = h3_inv * (21.333333333333 - 48.0 * u + 38.4 * u * u - 10.666666666667 * u * u * u - 0.066666666667 / (u * u * u));
                    }
                    const double f = -fac * G * q.w;
                    ax += dx * f;
                    ay += dy * f;
                    az += dz * f;
                }
    }
    __shared__ double red[3][4];
    for(int off = 32; off > 0; off >>= 1) {
        ax += __shfl_xor(ax, off);
        ay += __shfl_xor(ay, off);
        az += __shfl_xor(az, off);
    }
    if((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = ax;
        red[1][threadIdx.x >> 6] = ay;
        red[2][threadIdx.x >> 6] = az;
    }
    __syncthreads();
    if(threadIdx.x < 3)
        atomicAdd(&out[3 * t + threadIdx.x], red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
}
} // namespace

extern "C" int shq_direct_force_sample(shq_context *ctx, const double *sample_pos, int64_t ns, int64_t nsrc, double BoxSize, double G, double h,
                                       int repeat, double *accel)
{
    SHQ_CHECK(ctx && sample_pos && accel, SHQ_ERR_INVALID, "direct_force_sample: null argument");
    SHQ_CHECK(ctx->have_parts && nsrc >= 0 && nsrc <= ctx->numpart, SHQ_ERR_STATE, "direct_force_sample: %ld sources but %ld resident particles", (long) nsrc,
              (long) ctx->numpart);
    SHQ_CHECK(ns >= 0 && ns <= 65535 && BoxSize > 0 && h > 0 && repeat >= 0 && repeat <= 2, SHQ_ERR_INVALID, "direct_force_sample: bad arguments");
    SHQ_HIP(hipSetDevice(ctx->device));
    for(int64_t i = 0; i < 3 * ns; i++)
        accel[i] = 0;
    if(ns == 0 || nsrc == 0)
        return SHQ_OK;
    DevBuf<double> d_t, d_o;
    SHQ_TRY(d_t.reserve((size_t) (3 * ns)));
    SHQ_TRY(d_o.reserve((size_t) (3 * ns)));
    SHQ_HIP(hipMemcpyAsync(d_t.ptr, sample_pos, sizeof(double) * 3 * ns, hipMemcpyHostToDevice, ctx->stream));
    SHQ_HIP(hipMemsetAsync(d_o.ptr, 0, sizeof(double) * 3 * ns, ctx->stream));
    const long long chunk = 1 << 16;
    const unsigned nchunks = (unsigned) ((nsrc + chunk - 1) / chunk);
    direct_sample_kernel<<<dim3((unsigned) ns, nchunks), dim3(256), 0, ctx->stream>>>(d_t.ptr, (int) ns, ctx->posm.ptr, nsrc, chunk, BoxSize, G, h, repeat, d_o.ptr);
    SHQ_HIP(hipGetLastError());
    SHQ_HIP(hipMemcpyAsync(accel, d_o.ptr, sizeof(double) * 3 * ns, hipMemcpyDeviceToHost, ctx->stream));
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    d_t.release();
    d_o.release();
    return SHQ_OK;
}

int shq_launch_oldacc(shq_context *ctx, double G)
{
    SHQ_CHECK(ctx->have_parts, SHQ_ERR_STATE, "refresh_oldacc: no particles uploaded");
    if(ctx->numpart == 0)
        return SHQ_OK;
    const int threads = 256;
    const long long blocks = (ctx->numpart + threads - 1) / threads;
    oldacc_kernel<<<dim3((unsigned) blocks), dim3(threads), 0, ctx->stream>>>(ctx->numpart, ctx->treeacc.ptr,
                                                                              ctx->gravpm.ptr, ctx->oldacc.ptr, G);
    SHQ_HIP(hipGetLastError());
    return SHQ_OK;
}
