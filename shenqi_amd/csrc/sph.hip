/* sph.hip — SPH density (with the smoothing-length iteration) and hydro force for gfx950.
 *
 * Replaces treewalk_primary_kernel<DensityTreeWalk*> / <HydroTreeWalk*> (treewalk2.cuh:104-134,
 * densitycuda.cu, hydracuda.cu) AND the host-side Hsml loop the reference keeps on the CPU
 * (TreeWalk::do_hsml_loop, treewalk2.h:480-557; DensityOutput::postprocess +
 * density_check_neighbours, densitytree2.hpp:117-257; queue compaction).
 *
 * Structure (same wavefront-collective walk as grav_walk.hip):
 *   - a wavefront owns 64 consecutive targets of the work queue; the union of the lanes'
 *     neighbour walks is traversed with a wave-uniform `cur` and per-lane `mynext`; the cull
 *     test (cull_node, localtreewalk2.h:154-182) is evaluated per lane, so every target sees
 *     exactly the candidate set the reference's walk gives it, in the same order;
 *   - per-neighbour quantities that the reference recomputes for every pair (SPH_VelPred,
 *     SPH_EntVarPred, density/pressure prediction, sound speed, the Balsara factor f2 of j) are
 *     pure functions of particle j, so a prepass evaluates them once per particle and stores them
 *     in leaf order;
 *   - node records and candidate particles are staged through LDS (64-node window of the pre-order
 *     pool, 64-candidate tile) and accepted neighbours go to per-lane lists that are evaluated lane
 *     by lane (see ngb_walk below);
 *   - the Hsml iteration runs on the device: walk -> postprocess (bisection / Newton step,
 *     Left/Right brackets) -> order-preserving compaction of the redo queue; the host only reads
 *     back the queue length once per iteration.
 * All arithmetic is f64.
 */
#include "common.hpp"
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <math.h>
#include <stdlib.h>

#define SPH_GAMMA (5.0 / 3.0)      /* physconst.h:35 */
#define SPH_GAMMA_MINUS1 (SPH_GAMMA - 1)
#define SPH_MAXITER 400            /* treewalk2.h:21 */

namespace {

__device__ __forceinline__ double wrapd(double d, double L, double invL) { return fma(-L, rint(d * invL), d); }

/* Price (2012) kernels as libgadget/densitykernel.hpp:28-178 defines them (integer powers
 * written as products). KT: 1 cubic, 2 quintic, 4 quartic. */
template <int KT> struct Kern {
    static constexpr double support = (KT == 1) ? 4.0 : ((KT == 2) ? 6.0 : 5.0);
    double H, Wknorm, dnorm;
    __device__ __forceinline__ explicit Kern(double H_) : H(H_)
    {
        const double sigma = (KT == 1) ? (1 / M_PI) : ((KT == 2) ? (1 / (120 * M_PI)) : (1 / (20 * M_PI)));
        const double s = support / 2. / H;
        Wknorm = sigma * (s * s * s);
        dnorm = Wknorm * support / 2. / H;
    }
    static __device__ __forceinline__ double p3(double x) { return x * x * x; }
    static __device__ __forceinline__ double p4(double x) { const double y = x * x; return y * y; }
    static __device__ __forceinline__ double p5(double x) { const double y = x * x; return y * y * x; }
    __device__ __forceinline__ double wk_int(double q) const
    {
        if(KT == 1) {
            if(q < 1.0) return 0.25 * p3(2 - q) - p3(1 - q);
            if(q < 2.0) return 0.25 * p3(2 - q);
            return 0.0;
        } else if(KT == 4) {
            if(q < 0.5) return p4(2.5 - q) - 5 * p4(1.5 - q) + 10 * p4(0.5 - q);
            if(q < 1.5) return p4(2.5 - q) - 5 * p4(1.5 - q);
            if(q < 2.5) return p4(2.5 - q);
            return 0.0;
        } else {
            if(q < 1.0) return p5(3 - q) - 6 * p5(2 - q) + 15 * p5(1 - q);
            if(q < 2.0) return p5(3 - q) - 6 * p5(2 - q);
            if(q < 3.0) return p5(3 - q);
            return 0.0;
        }
    }
    __device__ __forceinline__ double dwk_int(double q) const
    {
        if(KT == 1) {
            if(q < 1.0) return -0.25 * 3 * (2 - q) * (2 - q) + 3 * (1 - q) * (1 - q);
            if(q < 2.0) return -0.25 * 3 * (2 - q) * (2 - q);
            return 0.0;
        } else if(KT == 4) {
            if(q < 0.5) return -4 * p3(2.5 - q) + 20 * p3(1.5 - q) - 40 * p3(0.5 - q);
            if(q < 1.5) return -4 * p3(2.5 - q) + 20 * p3(1.5 - q);
            if(q < 2.5) return -4 * p3(2.5 - q);
            return 0.0;
        } else {
            if(q < 1.0) return -5 * p4(3 - q) + 30 * p4(2 - q) - 75 * p4(1 - q);
            if(q < 2.0) return -5 * p4(3 - q) + 30 * p4(2 - q);
            if(q < 3.0) return -5 * p4(3 - q);
            return 0.0;
        }
    }
    __device__ __forceinline__ double wk(double u) const { return Wknorm * wk_int(u * support / 2.); }
    __device__ __forceinline__ double dwk(double u) const { return dnorm * dwk_int(u * support / 2.); }
    __device__ __forceinline__ double volume() const { return (4.0 / 3 * M_PI) * (H * H * H); }
};

/* KickFactorData::SPH_EntVarPred, density2.h:115-128 */
__device__ __forceinline__ double entvar_pred(double Entropy, double DtEntropy, double dloga)
{
    double e = Entropy + DtEntropy * dloga;
    if(e < 0.05 * Entropy)
        e = 0.05 * Entropy;
    if(e <= 0)
        return 0;
    return exp(1. / SPH_GAMMA * log(e));
}
/* SPH_DensityPred, hydratree2.hpp:21-34 */
__device__ __forceinline__ double density_pred(double Density, double DivVel, double dtdrift)
{
    const double d = Density - DivVel * Density * dtdrift;
    return (d >= 1e-6 * Density) ? d : 1e-6 * Density;
}
/* PressurePredict, hydratree2.hpp:47-58 */
__device__ __forceinline__ double pressure_predict(double eom, double evp)
{
    if(evp * eom <= 0)
        return 0;
    return exp(SPH_GAMMA * log(evp * eom));
}

/* Everything the hydro pair evaluation needs of neighbour j besides its position, in ONE cache line:
 * five separate leaf-ordered arrays cost five line fetches per (lane, pair), and the evaluation kernel
 * is bound by exactly that L2 -> L1 traffic. */
/* everything the hydro pair evaluation reads of a neighbour, in ONE 128-byte line (round 4: position and mass moved in, in the place of
 * the padding and of the EntVarPred copy next to the velocity: eight 16-byte gathers per pair instead of ten — the evaluation kernel is
 * bound by the texture addresser's cycles per gather instruction, not by arithmetic) */
struct alignas(128) HydRec {
    double4 posm; /* x, y, z, mass */
    double4 velh; /* predicted velocity, Hsml */
    double4 C;    /* EntVarPred, density_j, soundspeed_j, p_over_rho2_j */
    double4 D;    /* Dhsml_j, rr2_j, f2_j, dloga_for_bin_j */
};
static_assert(sizeof(HydRec) == 128, "HydRec must be one 128-byte line");

struct SphDev {
    /* node pool */
    const NodeB *nodeB;
    const NodeC *nodeC;
    double *hmax;               /* per node */
    const int32_t *pfather;     /* particle -> packed father leaf */
    int root;
    int npool;                  /* packed nodes */
    /* leaf-order neighbour data */
    const double4 *posm_leaf;   /* x,y,z,m */
    const double4 *velp_leaf;   /* predicted velocity, EntVarPred */
    const HydRec *hydrec_leaf;  /* one 128-byte line per neighbour for the hydro pair evaluation */
    const double *hsml_leaf;
    const int32_t *flag_leaf;   /* bit0 skip (garbage / not gas), bit1 wind-decoupled */
    const float4 *posf_leaf;    /* f32 pre-test copy: x, y, z rounded; w = pre32_bound(Hsml); x = NaN: skip, x = inf: never accepted */
    const int32_t *ngarb_leaf;  /* at a leaf's first slot: how many of its particles are to be skipped (x = NaN above) */
    /* per particle */
    const double4 *posm;
    const uint8_t *pflags;
    double *hsml;
    double *dthsml;
    const double4 *velp;
    const double4 *hydC;
    const double4 *hydD;
    /* density scratch / outputs, by particle index */
    double *numngb, *dhsmldens, *left, *right;
    double *rho, *egyrho, *dhsmlegy, *div, *curl;
    double *rot;      /* [N][3] */
    double *gradrho;  /* [N][3] or null */
    /* hydro outputs */
    double *hacc;     /* [N][3] */
    double *dtent, *maxsig;
    double Box, invBox;
    /* targets too heavy even for a wave of their own: handed on to the one-target-per-workgroup kernel */
    int32_t *heavy2;
    long long *nheavy2;
};

/* ---- prepass: per-particle predicted quantities -------------------------------------------- */
struct PredArgs {
    long long n;
    const uint8_t *pflags;
    const double *vel, *treeacc, *gravpm, *hydroaccel; /* [N][3] by particle index */
    const uint8_t *bin_grav, *bin_hydro;
    const double *entropy, *dtentropy;
    const double *hsml;
    const double *density, *egywt, *dhsmlegy, *divvel, *curlvel;
    double4 *velp, *hydC, *hydD;
    const double *evp_in; /* caller-provided EntVarPred by particle index, or null */
    shq_kick_factors kf;
    double drifts[SHQ_TIMEBINS + 1];
    int hydro;          /* also fill hydC/hydD */
    int DISPH;
    double fac_mu, contrast;
};

__global__ void sph_predict_kernel(const PredArgs a)
{
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= a.n)
        return;
    const int type = a.pflags[i] >> 4;
    if(type != 0) {
        /* non-gas (BH targets use their own velocity: DensityQuery ctor, densitytree2.hpp:270-277) */
        a.velp[i] = make_double4(a.vel[3 * i], a.vel[3 * i + 1], a.vel[3 * i + 2], 0.0);
        return;
    }
    const int bg = a.bin_grav[i], bh = a.bin_hydro[i];
    /* SPH_VelPred, density2.h:89-98 */
    double v[3];
    for(int j = 0; j < 3; j++)
        v[j] = a.vel[3 * i + j] + a.kf.gravkicks[bg] * a.treeacc[3 * i + j] + a.gravpm[3 * i + j] * a.kf.FgravkickB +
               a.kf.hydrokicks[bh] * a.hydroaccel[3 * i + j];
    const double evp = a.evp_in ? a.evp_in[i] : entvar_pred(a.entropy[i], a.dtentropy[i], a.kf.dloga_kick[bh]);
    a.velp[i] = make_double4(v[0], v[1], v[2], evp);
    if(a.hydro) {
        /* ngbiter's j-side quantities, hydratree2.hpp:283-300,325-326,357-364 */
        const double density_j = density_pred(a.density[i], a.divvel[i], a.drifts[bh]);
        const double eom_j = density_pred(a.DISPH ? a.egywt[i] : a.density[i], a.divvel[i], a.drifts[bh]);
        const double P = pressure_predict(eom_j, evp);
        const double cs = sqrt(SPH_GAMMA * P / eom_j);
        a.hydC[i] = make_double4(evp, density_j, cs, P / (eom_j * eom_j));
        const double f2 = fabs(a.divvel[i]) / (fabs(a.divvel[i]) + a.curlvel[i] + 0.0001 * cs / a.fac_mu / a.hsml[i]);
        double rr2 = 1;
        if(a.DISPH) {
            rr2 = 0;
            if(a.contrast >= 0) {
                rr2 = eom_j / density_j;
                if(a.contrast > 0)
                    rr2 = fmin(rr2, a.contrast);
            }
        }
        a.hydD[i] = make_double4(a.dhsmlegy[i], rr2, f2, a.kf.dloga_for_bin[bh]);
    }
}

/* ---- the f32 pre-test of the candidate scan (round 4) ----------------------------------------------------------------
 * The scan of a candidate tile tests ~2500 candidates per wave, of which a lane is interested in a tenth and accepts a
 * twentieth, in f64: 3 subtractions, 3 multiply-adds and a compare at four cycles each.  On gfx950 the f32 VOP2 forms issue at
 * twice that rate, and the decision does not have to be exact THERE: the lists may hold a superset as long as the evaluation
 * applies the reference's own test to every entry (it recomputes r2 in f64 anyway).  So the scan runs on coordinates rounded
 * to f32 (error <= 2^-24 |x| each, <= 2^-22 Box on a displacement with room to spare) against the bound
 * (h + 2^-20 Box)^2 (1 + 2^-20) rounded up: r < h in f64 implies the f32 test passes.  For h / Box = 3e-3 (1024^3) the lists
 * grow by 0.1 %.  Leaves whose displacements may need the periodic wrap keep the f64 scan (their tile does). */
__device__ __forceinline__ float pre32_bound(double h, double eps)
{
    const double b = (h + eps) * (h + eps) * (1.0 + 0x1p-20);
    float f = (float) b;
    if((double) f < b)
        f = __uint_as_float(__float_as_uint(f) + 1u); /* b > 0 and finite: the next float up */
    return f;
}

__global__ void sph_gather_leaf_kernel(long long nleaf, const int32_t *pidx, const double4 *velp, const double4 *hydC,
                                       const double4 *hydD, const double *hsml, const uint8_t *pflags, const double *delay,
                                       double4 *velp_leaf, HydRec *hydrec_leaf, double *hsml_leaf,
                                       int32_t *flag_leaf, const double4 *posm_leaf = nullptr, float4 *posf_leaf = nullptr, double pre_eps = 0)
{
    const long long s = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(s >= nleaf)
        return;
    const int p = pidx[s];
    velp_leaf[s] = velp[p];
    if(hydrec_leaf) {
        HydRec r;
        const double4 v = velp[p];
        r.posm = posm_leaf[s];
        r.velh = make_double4(v.x, v.y, v.z, hsml[p]);
        r.C = hydC[p];
        r.D = hydD[p];
        hydrec_leaf[s] = r;
    }
    hsml_leaf[s] = hsml[p];
    const uint8_t f = pflags[p];
    uint8_t o = 0;
    if((f & 1) || (f >> 4) != 0) /* IsGarbage, or type changed since the tree was built (GASMASK) */
        o |= 1;
    if(delay && delay[p] > 0)
        o |= 2;
    flag_leaf[s] = o;
    if(posf_leaf) {
        const double4 q = posm_leaf[s];
        float4 f = make_float4((float) q.x, (float) q.y, (float) q.z, pre32_bound(hsml[p], pre_eps));
        if(o & 1)
            f.x = __builtin_nanf("");             /* not a candidate at all */
        else if(hydrec_leaf && (o & 2))
            f.x = __builtin_inff();               /* hydro never accepts a wind-decoupled particle (it still counts as a candidate) */
        posf_leaf[s] = f;
    }
}

/* per leaf node: how many of its particles the scans pass over (flag bit 0), kept at the leaf's first slot: the f32 scan takes a
 * leaf's candidate count from its node and subtracts this */
__global__ void sph_leaf_ngarb_kernel(int npool, const NodeC *__restrict__ nodeC, const int32_t *__restrict__ flag_leaf, int32_t *ngarb_leaf)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= npool)
        return;
    const NodeC nc = nodeC[i];
    if(nc.type != SHQ_PARTICLE_NODE_TYPE || nc.count <= 0)
        return;
    int ng = 0;
    for(int j = 0; j < nc.count; j++)
        ng += flag_leaf[nc.child + j] & 1;
    ngarb_leaf[nc.child] = ng;
}

/* ---- density walk ------------------------------------------------------------------------------ */
/* Per-lane neighbour lists.  The union walk hands every leaf particle to all lanes whose search
 * sphere touches the leaf, but only a few of them actually have it within their kernel support; doing
 * the pair arithmetic (~100 f64 instructions for density, ~200 for hydro) under that mask keeps 10-20 %
 * of the lanes busy.  So the walk only runs the distance test (a dozen instructions) and appends the
 * accepted leaf slot to the lane's list; at the end of the walk (or when a list is full) every lane
 * works through ITS OWN list with vector loads, all lanes busy.  A lane meets its neighbours in exactly
 * the order of the depth-first walk, so sums are bit-identical to the immediate evaluation.
 * A lane collects its ~100 neighbours in a burst while the walk passes its corner of the group's
 * volume, so short LDS lists flushed whenever one lane fills up ran 431 pair rounds per wave for 112
 * pairs per target; the lists therefore live in global memory (L2-resident scratch, one region per
 * resident wave of a persistent grid, [entry][lane] so appends and reads coalesce) and are long enough
 * to be drained once. */
#define NL_CAP 256      /* list entries per lane; quintic-kernel neighbourhoods hold ~113, symmetric hydro lists up to ~200 */
#define NL_ROWS (NL_CAP + 64) /* rows of a wave's list region: a lane's fill is checked against NL_CAP once per candidate tile (<= 64 appends) */
#define NL_MAXBLOCKS 4096 /* persistent workgroups (4 waves each) that own a list region */

template <class F> __device__ __forceinline__ void nl_flush(const int32_t *myl, int &fill, F &&pair)
{
    int s_next = fill > 0 ? myl[0] : 0;
    for(int j = 0; shq_ballot(j < fill) != 0ull; j++) {
        const int s = s_next;
        if(j + 1 < fill)
            s_next = myl[(j + 1) * 64]; /* in flight while this pair is evaluated */
        if(j < fill)
            pair(s);
    }
    fill = 0;
}

/* ---- the neighbour walk shared by density and hydro ------------------------------------------------
 * Same wavefront-collective union walk as before (wave-uniform `cur`, per-lane `mynext`, cull_node per
 * lane), restructured around LDS so that no step waits on a dependent global load per node or per
 * candidate (with 4 waves per SIMD those ~1000-cycle waits kept the VALUs 44 % busy):
 *   - node window: the pool is in depth-first pre-order and a walk mostly moves forward in it, so the
 *     wave fetches 64 consecutive node records at a time (coalesced) into LDS and reads the node under
 *     the cursor from there; a jump outside the window reloads it;
 *   - candidate tile: leaves some lane wants are queued (slot range + the mask of interested lanes);
 *     when 64 candidates are queued the wave gathers them in ONE coalesced load, parks position, Hsml
 *     and flag in LDS, and every lane runs the distance test over the tile with broadcast reads;
 *   - accepted candidates go to the lane's list (see above) and are evaluated lane by lane.
 * Leaves are queued and tiles are scanned in walk order, so each lane still meets its neighbours in
 * depth-first order. */
#ifndef SPH_NODE_SCALAR
#define SPH_NODE_SCALAR 0 /* 1: the walk reads the node under its cursor with scalar loads instead of from an LDS window (A/B knob) */
#endif
#ifndef SPH_PROBE
#define SPH_PROBE 0 /* timing probes of the walk kernels (tools/sph_ab.sh with build_variant.sh): never in a shipped build */
#endif
#ifndef SPH_LEAF_ASM
#define SPH_LEAF_ASM 1 /* the walk-only kernels (KEEP) fetch the PRE32 scan's leaf records by inline-asm scalar loads straight into the
                          carried registers; 0: the compiler's loads everywhere (A/B knob).  tests/test_leaf_asm_isa_cpu.py reads the ISA
                          of those kernels and fails if the compiler ever copies or spills the registers while the loads travel */
#endif
#ifndef SPH_WALK_WPB
#define SPH_WALK_WPB 1 /* waves per block of the walk-only kernels (MODE 1) */
#endif
#define NW_WIN 32 /* nodes per window: half a window costs a few more reloads and buys 1.8 KB of LDS per wave */
/* per wave: node window (centre + len, links, hmax for the symmetric cull) and candidate tile (position, interested
 * lanes per queued leaf, leaf records, slot with the two flag bits on top, Hsml for the symmetric test): 4.5 KB (density) / 5.4 KB (hydro), so
 * that the walk kernels reach 7-8 waves per SIMD instead of 5 */
#define NW_LDS_PER_WAVE(SYM) (NW_WIN * (32 + 16 + ((SYM) ? 8 : 0)) + 64 * (32 + 8 + 4 + 4 + ((SYM) ? 8 : 0)))

/* KEEP: only build the lists (two-kernel path): nothing is evaluated, `fill` returns the list length, and a
 * lane whose list would overflow sets `ovf` (its wave is then redone by the fused kernel). */
/* GHOSTS (LocalNgbTreeWalk::visit<TREEWALK_GHOSTS>, localtreewalk2.h:378-437): an imported query walks only the branches
 * under the top-level nodes of its NodeList = the pre-order index ranges [start, sibling(start)); `seg` holds the (sorted)
 * packed start indices.  As in the gravity walk a lane waits at the start of its next branch and the wave cursor, which
 * still begins at the root, also descends wherever a lane waits further down. */
/* PRE32: tiles none of whose leaves may need the periodic wrap are scanned with the f32 pre-test (pre32_bound above; thri = the
 * target's own bound); `pair` must then apply the exact test itself. */
template <bool SYM, bool KEEP, bool GHOSTS, bool PRE32 = false, class Accept, class Pair>
__device__ __forceinline__ unsigned int ngb_walk(const SphDev &a, char *lds_wave, int32_t *myl, const bool valid, const double px,
                                                 const double py, const double pz, const double h, Accept &&accept, Pair &&pair,
                                                 unsigned int *dbg, int &fill, bool &ovf, const int4 seg = make_int4(-1, -1, -1, -1),
                                                 const float thri = 0.f)
{
    double4 *winB = reinterpret_cast<double4 *>(lds_wave);
    int4 *winC = reinterpret_cast<int4 *>(lds_wave + NW_WIN * 32);
    double *winH = reinterpret_cast<double *>(lds_wave + NW_WIN * 48); /* SYM only */
    char *tl = lds_wave + NW_WIN * (SYM ? 56 : 48);
    double4 *tq = reinterpret_cast<double4 *>(tl);
    unsigned long long *lqm = reinterpret_cast<unsigned long long *>(tl + 64 * 32); /* per queued LEAF: the lanes that want it */
    int *tsl = reinterpret_cast<int *>(tl + 64 * 40);                  /* leaf slot | flags << 30 once gathered */
    int *lqi = reinterpret_cast<int *>(tl + 64 * 44);                  /* per queued leaf: first candidate | count << 8 | may-wrap << 16 */
    double *th = reinterpret_cast<double *>(tl + 64 * 48);             /* SYM only */
    const int lane = threadIdx.x & 63;
    const double halfBox = 0.5 * a.Box;
    unsigned int nint = 0;
    int ncand = 0, nleafq = 0;
    fill = 0;
    ovf = false;
    int mynext = valid ? a.root : -2;
    /* PRE32: the leaf waiting for its f32 records (scalar loads: a leaf's records are consecutive and the same for every lane, so
     * they travel through the scalar cache into scalar registers: no LDS tile, no gather, no vector register) */
    typedef const float __attribute__((address_space(4))) *FloatK;
    typedef const int32_t __attribute__((address_space(4))) *IntK;
    typedef __attribute__((address_space(1))) char *GChar;
    typedef const double __attribute__((address_space(4))) *DoubleK;
    const DoubleK nodeBK = (DoubleK) (size_t) a.nodeB, hmaxK = (DoubleK) (size_t) a.hmax;
    const IntK nodeCK = (IntK) (size_t) a.nodeC;
    (void) nodeBK; (void) hmaxK; (void) nodeCK;
    const FloatK posfK = (FloatK) (size_t) a.posf_leaf;
    auto ldrec = [&](const int slot) { return make_float4(posfK[4 * slot], posfK[4 * slot + 1], posfK[4 * slot + 2], posfK[4 * slot + 3]); };
    const IntK ngarbK = (IntK) (size_t) a.ngarb_leaf;
    const float pfx = (float) px, pfy = (float) py, pfz = (float) pz;
    GChar region = nullptr; /* the wave's list region as a scalar base: an append is one store with a 32-bit lane offset */
    if(PRE32) {
        const unsigned long long rb = (unsigned long long) (myl - lane);
        region = (GChar) (size_t) (((unsigned long long) (unsigned) __builtin_amdgcn_readfirstlane((int) (rb >> 32)) << 32) |
                                   (unsigned) __builtin_amdgcn_readfirstlane((int) rb));
    }
    int p_cn = 0, p_sb = 0, p_ng = 0;
    unsigned long long p_km = 0ull;
    /* KEEP (the walk-only kernels of the two-kernel path): the waiting leaf's records by inline-asm scalar loads straight into the
     * loop-carried registers, waited for in process_pending.  The compiler's own loads go through temporaries and are waited for and
     * copied at the join (s_waitcnt lgkmcnt(0) + a column of s_mov right behind the loads): nothing travelled while the walk went on.
     * The asm hides the loads from the compiler, so nothing may copy or spill those registers between the loads and the wait: true of
     * the walk-only kernels (checked on their ISA by tests/test_leaf_asm_isa_cpu.py at every build of the test suite), not of the fused
     * kernels, whose evaluation code makes the allocator spill scalars: those keep the compiler's loads. */
    constexpr bool LEAF_ASM = SPH_LEAF_ASM && KEEP && !GHOSTS;
    typedef float f4s __attribute__((ext_vector_type(4)));
    f4s pd0, pd1, pd2, pd3, pd4, pd5, pd6, pd7;
    pd0 = pd1 = pd2 = pd3 = pd4 = pd5 = pd6 = pd7 = (f4s) (0.f);

    /* scan the queued candidates: one coalesced gather, then broadcast reads.  Leaf by leaf (round 4): what is the same for a leaf's
     * particles - which lanes want it, whether a displacement to it can need the periodic wrap at all - is read once per leaf into
     * scalar registers (the interested lanes become the lane condition through an inverse ballot: no vector instruction), and a lane's
     * fill is checked against NL_CAP once per tile (the list region has 64 rows of slack) instead of once per candidate:
     * 22 -> 13 vector instructions per candidate, the same candidates in the same order with the same accept decisions. */
    auto scan_tile = [&]() {
        if(lane < ncand) {
            const int s = tsl[lane];
            tq[lane] = a.posm_leaf[s];
            tsl[lane] = s | (a.flag_leaf[s] << 30);
            if(SYM)
                th[lane] = a.hsml_leaf[s];
        }
        __builtin_amdgcn_wave_barrier();
        for(int L = 0; L < nleafq; L++) {
            const int info = __builtin_amdgcn_readfirstlane(lqi[L]);
            const unsigned long long kmv = lqm[L];
            const unsigned long long kms = ((unsigned long long) (unsigned) __builtin_amdgcn_readfirstlane((int) (kmv >> 32)) << 32) |
                                           (unsigned) __builtin_amdgcn_readfirstlane((int) kmv);
            const int c0 = info & 0xff, cn = (info >> 8) & 0xff;
            const bool maywrap = (info >> 16) != 0;
            /* what holds for the whole leaf: the lane wants it and has not left the walk (a lane leaves at a tile's end only) */
            const bool keepL = __builtin_amdgcn_inverse_ballot_w64(kms) && !(KEEP && ovf);
            /* the candidate's slot and flags are the same in every lane: scalar registers, and a garbage particle (rare) is passed
             * over by a scalar branch instead of a lane condition; eight copies of the body with compile-time LDS offsets */
#pragma unroll
            for(int k = 0; k < SHQ_NMAXCHILD; k++) {
                if(k >= cn)
                    break;
                const int sf = __builtin_amdgcn_readfirstlane(tsl[c0 + k]), s = sf & 0x3fffffff, fl = (int) ((unsigned) sf >> 30);
                if(fl & 1)
                    continue;
                const double4 q = tq[c0 + k];
                const double hj = SYM ? th[c0 + k] : 0.0;
                double d0 = px - q.x, d1 = py - q.y, d2 = pz - q.z;
                if(maywrap) { /* wave-uniform; wrapping a displacement that does not need it is the identity */
                    d0 = wrapd(d0, a.Box, a.invBox);
                    d1 = wrapd(d1, a.Box, a.invBox);
                    d2 = wrapd(d2, a.Box, a.invBox);
                }
                const double r2 = d0 * d0 + d1 * d1 + d2 * d2;
                if(keepL) {
                    nint++;
                    if(accept(r2, hj, fl)) {
                        myl[fill * 64] = s;
                        fill++;
                    }
                }
            }
        }
        if(KEEP) {
            if(fill >= NL_CAP) { /* this target's list does not fit: it leaves the walk (the others' lists stay good) and is
                                    walked on its own by a whole wave afterwards (heavy_walk) */
                ovf = true;
                fill = 0;
                mynext = -2;
            }
        } else if(shq_ballot(fill >= NL_CAP) != 0ull) {
            if(dbg)
                dbg[2] += NL_CAP;
            nl_flush(myl, fill, pair);
        }
        if(dbg)
            dbg[1] += ncand;
        __builtin_amdgcn_wave_barrier();
        ncand = 0;
        nleafq = 0;
    };

    /* PRE32: the f32 pre-test of the waiting leaf's candidates, straight-line code per leaf size; the candidates are scalar operands */
    auto process_pending = [&]() {
        if(p_cn == 0 || (SPH_PROBE == 1 && SYM)) /* probe 1: the hydro walk without its candidates */
            return;
        if(LEAF_ASM)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(pd0), "+s"(pd1), "+s"(pd2), "+s"(pd3), "+s"(pd4), "+s"(pd5), "+s"(pd6), "+s"(pd7), "+s"(p_ng));
        if(__builtin_amdgcn_inverse_ballot_w64(p_km) && !(KEEP && ovf)) { /* the lane mask of the whole leaf */
            nint += p_cn - p_ng;
            unsigned int off = ((unsigned int) fill * 64u + (unsigned int) lane) * 4u;
#if SPH_PROBE == 2 /* timing probe: the hydro walk's tests without the list (the lists stay empty) */
#define PRE32_STORE(k)                                                                                                                  \
    if(SYM)                                                                                                                             \
        nint++;                                                                                                                         \
    else {                                                                                                                              \
        *reinterpret_cast<__attribute__((address_space(1))) int32_t *>(region + off) = p_sb + (k);                                      \
        off += 256u;                                                                                                                    \
    }
#else
#define PRE32_STORE(k)                                                                                                                  \
    *reinterpret_cast<__attribute__((address_space(1))) int32_t *>(region + off) = p_sb + (k);                                          \
    off += 256u;
#endif
#define PRE32_BODY(k, Q)                                                                                                                \
    {                                                                                                                                   \
        const float e0 = pfx - Q.x, e1 = pfy - Q.y, e2 = pfz - Q.z;                                                                     \
        const float rr = e0 * e0 + e1 * e1 + e2 * e2;                                                                                   \
        if(rr < (SYM ? fmaxf(thri, Q.w) : thri)) {                                                                                      \
            PRE32_STORE(k)                                                                                                              \
        }                                                                                                                               \
    }
            switch(p_cn) {
            case 1: PRE32_BODY(0, pd0) break;
            case 2: PRE32_BODY(0, pd0) PRE32_BODY(1, pd1) break;
            case 3: PRE32_BODY(0, pd0) PRE32_BODY(1, pd1) PRE32_BODY(2, pd2) break;
            case 4: PRE32_BODY(0, pd0) PRE32_BODY(1, pd1) PRE32_BODY(2, pd2) PRE32_BODY(3, pd3) break;
            case 5: PRE32_BODY(0, pd0) PRE32_BODY(1, pd1) PRE32_BODY(2, pd2) PRE32_BODY(3, pd3) PRE32_BODY(4, pd4) break;
            case 6: PRE32_BODY(0, pd0) PRE32_BODY(1, pd1) PRE32_BODY(2, pd2) PRE32_BODY(3, pd3) PRE32_BODY(4, pd4) PRE32_BODY(5, pd5) break;
            case 7: PRE32_BODY(0, pd0) PRE32_BODY(1, pd1) PRE32_BODY(2, pd2) PRE32_BODY(3, pd3) PRE32_BODY(4, pd4) PRE32_BODY(5, pd5) PRE32_BODY(6, pd6) break;
            default: PRE32_BODY(0, pd0) PRE32_BODY(1, pd1) PRE32_BODY(2, pd2) PRE32_BODY(3, pd3) PRE32_BODY(4, pd4) PRE32_BODY(5, pd5) PRE32_BODY(6, pd6) PRE32_BODY(7, pd7) break;
            }
#undef PRE32_BODY
            fill = (int) (off >> 8); /* lane * 4 < 256 */
        }
        if(dbg)
            dbg[1] += p_cn;
        p_cn = 0;
        if(KEEP) {
            if(fill >= NL_CAP) { /* as in scan_tile; checked per leaf here (<= 8 appends, the region has 64 rows of slack) */
                ovf = true;
                fill = 0;
                mynext = -2;
            }
        } else if(shq_ballot(fill >= NL_CAP) != 0ull) {
            if(dbg)
                dbg[2] += NL_CAP;
            nl_flush(myl, fill, pair);
        }
    };

    int seg1 = -1, seg2 = -1, seg3 = -1, myend = -1;
    if(GHOSTS) {
        mynext = (valid && seg.x >= 0) ? seg.x : -2;
        seg1 = seg.y; seg2 = seg.z; seg3 = seg.w;
        if(mynext >= 0)
            myend = a.nodeC[mynext].sibling;
    }
    int cur = a.root, wbase = -(1 << 30);
    /* The loop is written for the scalar pipe (round 4; the counter pass had 69 k scalar beside 56 k vector instructions per wave, and a
     * scalar instruction holds its pipe for four cycles like an f64 one): lane conditions exist only as ballot masks, combined with
     * 64-bit scalar algebra and read back through inverse ballots; wave-uniform decisions are compare-and-branch on those masks,
     * never booleans the compiler would materialise as masks of their own; one window test; the lanes' links are set before the
     * leaf is handed on, so that nothing after the hand-over depends on it. */
    while(cur >= 0) {
#if SPH_NODE_SCALAR
        /* the node under the cursor through the scalar cache into scalar registers (as the gravity walk reads its nodes): no LDS window,
         * no refills, no read-first-lanes; the tests take the record's fields as scalar operands */
        const double4 B = make_double4(nodeBK[4 * (size_t) cur], nodeBK[4 * (size_t) cur + 1], nodeBK[4 * (size_t) cur + 2], nodeBK[4 * (size_t) cur + 3]);
        const int Csib = nodeCK[4 * (size_t) cur], Cchild = nodeCK[4 * (size_t) cur + 1];
        const int Ctype = nodeCK[4 * (size_t) cur + 2], Ccount = nodeCK[4 * (size_t) cur + 3];
        const double Hnode = SYM ? hmaxK[(size_t) cur] : 0.0;
#else
        if((unsigned int) (cur - wbase) >= (unsigned int) NW_WIN) {
            wbase = cur;
            __builtin_amdgcn_wave_barrier();
            if(lane < NW_WIN) {
                const int idx = min(cur + lane, a.npool - 1);
                const NodeB nb = a.nodeB[idx];
                const NodeC nc = a.nodeC[idx];
                winB[lane] = make_double4(nb.center[0], nb.center[1], nb.center[2], nb.len);
                winC[lane] = make_int4(nc.sibling, nc.child, nc.type, nc.count);
                if(SYM)
                    winH[lane] = a.hmax[idx];
            }
            __builtin_amdgcn_wave_barrier();
        }
        const int w = cur - wbase;
        const double4 B = winB[w];
        const int4 Cv = winC[w];
        const int Csib = __builtin_amdgcn_readfirstlane(Cv.x), Cchild = __builtin_amdgcn_readfirstlane(Cv.y);
        const int Ctype = __builtin_amdgcn_readfirstlane(Cv.z), Ccount = __builtin_amdgcn_readfirstlane(Cv.w);
        const double Hnode = SYM ? winH[w] : 0.0;
#endif
        if(dbg)
            dbg[0]++;
        /* cull_node<symmetric>, localtreewalk2.h:154-182 */
        const unsigned long long actm = shq_ballot(mynext == cur);
        const double dist = (SYM ? fmax(Hnode, h) : h) + 0.5 * B.w;
        double dx = B.x - px, dy = B.y - py, dz = B.z - pz;
        double dmax = fmax(fmax(fabs(dx), fabs(dy)), fabs(dz));
        const unsigned long long wrapm = shq_ballot(dmax > halfBox) & actm;
        if(wrapm != 0ull) {
            dx = wrapd(dx, a.Box, a.invBox);
            dy = wrapd(dy, a.Box, a.invBox);
            dz = wrapd(dz, a.Box, a.invBox);
            dmax = fmax(fmax(fabs(dx), fabs(dy)), fabs(dz));
        }
        asm volatile("" ::: "memory");
        const double r2 = dx * dx + dy * dy + dz * dz;
        const double dist2 = dist + (0.5 * (1.7320508075688772 - 1.0)) * B.w;
        const unsigned long long keepm = actm & ~(shq_ballot(dmax > dist) | shq_ballot(r2 > dist2 * dist2));
        /* the node types without control flow: every awake lane passes on to the sibling; the lanes that keep an internal node go down
         * instead, and the cursor with them; a kept leaf is the one branch */
        unsigned long long openm = Ctype == SHQ_NODE_NODE_TYPE ? keepm : 0ull;
        const unsigned long long leafm = (Ctype == SHQ_PARTICLE_NODE_TYPE && Ccount > 0) ? keepm : 0ull;
        unsigned long long downm = openm;
        if(GHOSTS) /* a lane waits at a branch below this node: go down even if nobody opens it */
            downm |= Ctype == SHQ_NODE_NODE_TYPE ? shq_ballot(mynext > cur && (Csib < 0 || mynext < Csib)) : 0ull;
        if(__builtin_amdgcn_inverse_ballot_w64(actm))
            mynext = Csib;
        if(__builtin_amdgcn_inverse_ballot_w64(openm))
            mynext = Cchild;
        const int next = downm != 0ull ? Cchild : Csib;
        asm volatile("" ::: "memory");
        if(leafm != 0ull) {
            if(!PRE32 && ncand + Ccount > 64)
                scan_tile();
            /* can a displacement from an interested lane to a particle of this leaf need the periodic wrap?  The particles lie in
             * the leaf's cell: |p - pos| <= |centre - pos| + len / 2 per coordinate; no, unless the node test itself wrapped or
             * that bound comes near Box / 2 for some interested lane (a conservative yes costs three identity wraps) */
            const unsigned long long nearm = wrapm | (shq_ballot(dmax + 0.5 * B.w > 0.999 * halfBox) & leafm);
            if(PRE32)
                process_pending(); /* leaves are scanned in walk order: the waiting one first */
            if(PRE32 && nearm == 0ull) {
                /* this leaf waits for its records while the walk goes on */
                if(LEAF_ASM) {
                    const FloatK rp = posfK + 4 * (size_t) Cchild;
                    const IntK gp = ngarbK + (size_t) Cchild;
                    asm volatile("s_load_dwordx4 %0, %9, 0x0\n\ts_load_dwordx4 %1, %9, 0x10\n\ts_load_dwordx4 %2, %9, 0x20\n\t"
                                 "s_load_dwordx4 %3, %9, 0x30\n\ts_load_dwordx4 %4, %9, 0x40\n\ts_load_dwordx4 %5, %9, 0x50\n\t"
                                 "s_load_dwordx4 %6, %9, 0x60\n\ts_load_dwordx4 %7, %9, 0x70\n\ts_load_dword %8, %10, 0x0"
                                 : "=&s"(pd0), "=&s"(pd1), "=&s"(pd2), "=&s"(pd3), "=&s"(pd4), "=&s"(pd5), "=&s"(pd6), "=&s"(pd7), "=&s"(p_ng)
                                 : "s"(rp), "s"(gp));
                } else {
                    auto ld4 = [&](const int slot) { const float4 r = ldrec(slot); f4s v; v.x = r.x; v.y = r.y; v.z = r.z; v.w = r.w; return v; };
                    pd0 = ld4(Cchild); pd1 = ld4(Cchild + 1); pd2 = ld4(Cchild + 2); pd3 = ld4(Cchild + 3);
                    pd4 = ld4(Cchild + 4); pd5 = ld4(Cchild + 5); pd6 = ld4(Cchild + 6); pd7 = ld4(Cchild + 7);
                    p_ng = ngarbK[Cchild];
                }
                p_km = leafm;
                p_sb = Cchild;
                p_cn = Ccount;
            } else {
                if(lane < Ccount)
                    tsl[ncand + lane] = Cchild + lane;
                if(lane == 0) {
                    lqm[nleafq] = leafm;
                    lqi[nleafq] = ncand | (Ccount << 8) | ((nearm != 0ull ? 1 : 0) << 16);
                }
                nleafq++;
                ncand += Ccount;
                if(PRE32)
                    scan_tile(); /* a leaf that may need the periodic wrap: the f64 scan, at once */
            }
        }
        if(GHOSTS) {
            if(shq_ballot(mynext == myend) & actm) { /* rare: some lane's branch is done */
                if(__builtin_amdgcn_inverse_ballot_w64(actm) && mynext == myend) { /* wait at the next one of the NodeList */
                    mynext = seg1 >= 0 ? seg1 : -2;
                    seg1 = seg2;
                    seg2 = seg3;
                    seg3 = -1;
                    myend = mynext >= 0 ? a.nodeC[mynext].sibling : -1;
                }
            }
        }
        cur = next;
    }
    if(PRE32)
        process_pending();
    if(ncand > 0)
        scan_tile();
    if(dbg) {
        int mf = fill;
        for(int off = 32; off > 0; off >>= 1)
            mf = max(mf, __shfl_xor(mf, off));
        dbg[2] += mf;
    }
    if(!KEEP)
        nl_flush(myl, fill, pair);
    return nint;
}

/* ---- one target, one wave --------------------------------------------------------------------------------------
 * A target whose neighbour list outgrows NL_CAP (gas next to a density caustic: the kernel-weighted count reaches its
 * target only when the support sphere already holds thousands of particles near its rim) would keep one lane busy for
 * as many rounds as it has neighbours while the other 63 idle.  Such targets are taken out of the group walks and each
 * gets a wave to itself, with the lanes on the work instead of on targets: node rounds pop up to 64 nodes of an LDS
 * stack and cull them one per lane (cull_node, localtreewalk2.h:154-182, against the one target), kept leaves queue
 * their particle slots, and candidate rounds take 64 queued particles, one per lane, through the same accept / pair
 * code; the lanes' partial sums are added across the wave at the end.  Same neighbour set and candidate count as the
 * group walk; the sum runs in a different order (rounding-level differences). */
#define HW_STACK 1024
#define HW_SOFT 512   /* above this fill the walk goes depth-first, one node per round: at most 7 more per tree level */
#define HW_UNR 4      /* candidates per lane and candidate round (round 4): the round's gathers are independent loads in flight together */
#define HW_CAND (64 * HW_UNR + 64 * 8 + 64) /* < 64 HW_UNR pending + <= 64 x 8 queued per node round */
#define HW_LDS ((HW_STACK + HW_CAND) * 4)

#define HW_ABORT 16384 /* candidates after which a wave gives its target up to a whole workgroup (heavy_block) */

template <bool SYM, class Accept, class Pair>
__device__ __forceinline__ unsigned int heavy_walk(const SphDev &a, char *lds_wave, const double px, const double py, const double pz,
                                                   const double h, Accept &&accept, Pair &&pair, bool &aborted)
{
    int *stk = reinterpret_cast<int *>(lds_wave);
    int *cq = stk + HW_STACK;
    const int lane = threadIdx.x & 63;
    unsigned int nint = 0;
    int S = 1, nc = 0, chead = 0, done = 0;
    aborted = false;
    if(lane == 0)
        stk[0] = a.root;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for(;;) {
        if(nc >= 64 * HW_UNR || (S == 0 && nc > 0)) {
            /* candidate round: up to HW_UNR candidates per lane, their records requested together (a round used to be one dependent
             * gather of 64 records: a target with 10^4 candidates spent its time waiting for 160 of them, one after the other) */
            const int m = nc < 64 * HW_UNR ? nc : 64 * HW_UNR;
            int sj[HW_UNR], flj[HW_UNR];
            double4 qj[HW_UNR];
            double hjj[HW_UNR];
#pragma unroll
            for(int j = 0; j < HW_UNR; j++) {
                const int idx = lane + 64 * j;
                sj[j] = -1;
                flj[j] = 1;
                qj[j] = make_double4(0, 0, 0, 0);
                hjj[j] = 0.0;
                if(idx < m) {
                    const int s = cq[(chead + idx) % HW_CAND];
                    sj[j] = s;
                    qj[j] = a.posm_leaf[s];
                    flj[j] = a.flag_leaf[s];
                    hjj[j] = SYM ? a.hsml_leaf[s] : 0.0;
                }
            }
#pragma unroll
            for(int j = 0; j < HW_UNR; j++)
                if(sj[j] >= 0 && !(flj[j] & 1)) {
                    nint++;
                    const double4 q = qj[j];
                    const double d0 = wrapd(px - q.x, a.Box, a.invBox), d1 = wrapd(py - q.y, a.Box, a.invBox), d2 = wrapd(pz - q.z, a.Box, a.invBox);
                    if(accept(d0 * d0 + d1 * d1 + d2 * d2, hjj[j], flj[j]))
                        pair(sj[j]);
                }
            chead = (chead + m) % HW_CAND;
            nc -= m;
            done += m;
            if(done > HW_ABORT && a.heavy2) {
                aborted = true;
                return 0;
            }
            __builtin_amdgcn_wave_barrier();
        } else if(S > 0) {
            /* node round */
            int k = S < 64 ? S : 64;
            const int room = (HW_SOFT - S) / 7;
            if(room < k)
                k = room > 1 ? room : 1;
            const bool on = lane < k;
            const int node = on ? stk[S - 1 - lane] : 0;
            S -= k;
            __builtin_amdgcn_wave_barrier();
            bool keep = false;
            NodeC nc4;
            nc4.sibling = nc4.child = -1;
            nc4.type = SHQ_PSEUDO_NODE_TYPE;
            nc4.count = 0;
            if(on) {
                const NodeB nb = a.nodeB[node];
                nc4 = a.nodeC[node];
                const double dist = (SYM ? fmax(a.hmax[node], h) : h) + 0.5 * nb.len;
                const double dx = wrapd(nb.center[0] - px, a.Box, a.invBox), dy = wrapd(nb.center[1] - py, a.Box, a.invBox),
                             dz = wrapd(nb.center[2] - pz, a.Box, a.invBox);
                const double dmax = fmax(fmax(fabs(dx), fabs(dy)), fabs(dz));
                const double r2 = dx * dx + dy * dy + dz * dz;
                const double dist2 = dist + (0.5 * (1.7320508075688772 - 1.0)) * nb.len;
                keep = !(dmax > dist) && !(r2 > dist2 * dist2);
            }
            /* kept leaves queue their particle slots */
            const bool leaf = keep && nc4.type == SHQ_PARTICLE_NODE_TYPE;
#pragma unroll
            for(int j = 0; j < SHQ_NMAXCHILD; j++) {
                const bool has = leaf && j < nc4.count;
                const unsigned long long msk = shq_ballot(has);
                if(msk == 0ull)
                    break;
                if(has)
                    cq[(chead + nc + __builtin_amdgcn_mbcnt_hi((unsigned) (msk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) msk, 0u))) % HW_CAND] =
                        nc4.child + j;
                nc += __popcll(msk);
            }
            /* kept internal nodes push their children: the first one, then along the sibling links up to the node's own sibling */
            int c = (keep && nc4.type == SHQ_NODE_NODE_TYPE) ? nc4.child : -1;
            for(int j = 0; j < 8; j++) {
                const bool has = c >= 0 && c != nc4.sibling;
                const unsigned long long msk = shq_ballot(has);
                if(msk == 0ull)
                    break;
                if(has) {
                    const int at = S + __builtin_amdgcn_mbcnt_hi((unsigned) (msk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) msk, 0u));
                    if(at < HW_STACK)
                        stk[at] = c;
                    c = a.nodeC[c].sibling;
                }
                S += __popcll(msk);
            }
            if(S > HW_STACK) { /* deeper than the slack allows (64 levels): stop rather than walk a truncated stack; the sums come out wrong and
                                  the Hsml loop reports non-convergence */
                S = 0;
                nc = 0;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else
            break;
    }
    return nint;
}

__device__ __forceinline__ double wave_sum(double v)
{
    for(int off = 32; off > 0; off >>= 1)
        v += __shfl_xor(v, off);
    return v;
}

/* ---- one target, one workgroup of 8 waves: the same walk for the few targets whose support sphere holds a good part of a
 * dense clump (10^5 - 10^6 candidates: a wave alone would need tens of milliseconds).  Stack and candidate queue are shared
 * in LDS; every position comes from prefix sums over the thread index, so which thread meets which candidate — and with it
 * the order of the sum — is fixed: results are reproducible run to run. */
#define HB_THREADS 512
#define HB_WAVES (HB_THREADS / 64)
#define HB_STACK 8192
#define HB_SOFT 4096
#define HB_UNR 4 /* candidates per thread and candidate round, as HW_UNR */
#define HB_CAND (HB_THREADS * (8 + HB_UNR) + HB_THREADS)

struct HbShared {
    int stk[HB_STACK];
    int cq[HB_CAND];
    int wtot[2][HB_WAVES];
    double red[HB_WAVES];
    int S, nc, chead;
};

/* exclusive prefix over the threads of the workgroup of a count in [0, 8]; total to all */
__device__ __forceinline__ int hb_scan8(int v, int *wtot, int &total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long b0 = shq_ballot((v & 1) != 0), b1 = shq_ballot((v & 2) != 0), b2 = shq_ballot((v & 4) != 0), b3 = shq_ballot((v & 8) != 0);
    auto mb = [](unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((unsigned) (m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) m, 0u)); };
    const int pre = mb(b0) + 2 * mb(b1) + 4 * mb(b2) + 8 * mb(b3);
    if(lane == 0)
        wtot[wv] = __popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2) + 8 * __popcll(b3);
    __syncthreads();
    int base = 0;
    total = 0;
    for(int w = 0; w < HB_WAVES; w++) {
        const int c = wtot[w];
        if(w < wv)
            base += c;
        total += c;
    }
    return base + pre;
}

__device__ __forceinline__ double hb_sum(double v, double *red)
{
    v = wave_sum(v);
    __syncthreads();
    if((threadIdx.x & 63) == 0)
        red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0;
    for(int w = 0; w < HB_WAVES; w++)
        t += red[w];
    return t;
}

__device__ __forceinline__ double hb_max(double v, double *red)
{
    for(int off = 32; off > 0; off >>= 1)
        v = fmax(v, __shfl_xor(v, off));
    __syncthreads();
    if((threadIdx.x & 63) == 0)
        red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = red[0];
    for(int w = 1; w < HB_WAVES; w++)
        t = fmax(t, red[w]);
    return t;
}

template <bool SYM, class Accept, class Pair>
__device__ __forceinline__ unsigned int heavy_block(const SphDev &a, HbShared &sh, const double px, const double py, const double pz, const double h,
                                                    Accept &&accept, Pair &&pair)
{
    const int tid = threadIdx.x;
    unsigned int nint = 0;
    __syncthreads();
    if(tid == 0) {
        sh.stk[0] = a.root;
        sh.S = 1;
        sh.nc = 0;
        sh.chead = 0;
    }
    for(;;) {
        __syncthreads();
        const int S = sh.S, nc = sh.nc, chead = sh.chead;
        __syncthreads();
        if(nc >= HB_THREADS * HB_UNR || (S == 0 && nc > 0)) {
            const int m = nc < HB_THREADS * HB_UNR ? nc : HB_THREADS * HB_UNR;
            int sj[HB_UNR], flj[HB_UNR];
            double4 qj[HB_UNR];
            double hjj[HB_UNR];
#pragma unroll
            for(int j = 0; j < HB_UNR; j++) {
                const int idx = tid + HB_THREADS * j;
                sj[j] = -1;
                flj[j] = 1;
                qj[j] = make_double4(0, 0, 0, 0);
                hjj[j] = 0.0;
                if(idx < m) {
                    const int s = sh.cq[(chead + idx) % HB_CAND];
                    sj[j] = s;
                    qj[j] = a.posm_leaf[s];
                    flj[j] = a.flag_leaf[s];
                    hjj[j] = SYM ? a.hsml_leaf[s] : 0.0;
                }
            }
#pragma unroll
            for(int j = 0; j < HB_UNR; j++)
                if(sj[j] >= 0 && !(flj[j] & 1)) {
                    nint++;
                    const double4 q = qj[j];
                    const double d0 = wrapd(px - q.x, a.Box, a.invBox), d1 = wrapd(py - q.y, a.Box, a.invBox), d2 = wrapd(pz - q.z, a.Box, a.invBox);
                    if(accept(d0 * d0 + d1 * d1 + d2 * d2, hjj[j], flj[j]))
                        pair(sj[j]);
                }
            if(tid == 0) {
                sh.chead = (chead + m) % HB_CAND;
                sh.nc = nc - m;
            }
        } else if(S > 0) {
            int k = S < HB_THREADS ? S : HB_THREADS;
            const int room = (HB_SOFT - S) / 7;
            if(room < k)
                k = room > 1 ? room : 1;
            const bool on = tid < k;
            const int node = on ? sh.stk[S - 1 - tid] : 0;
            bool keep = false;
            NodeC nc4;
            nc4.sibling = nc4.child = -1;
            nc4.type = SHQ_PSEUDO_NODE_TYPE;
            nc4.count = 0;
            if(on) {
                const NodeB nb = a.nodeB[node];
                nc4 = a.nodeC[node];
                const double dist = (SYM ? fmax(a.hmax[node], h) : h) + 0.5 * nb.len;
                const double dx = wrapd(nb.center[0] - px, a.Box, a.invBox), dy = wrapd(nb.center[1] - py, a.Box, a.invBox),
                             dz = wrapd(nb.center[2] - pz, a.Box, a.invBox);
                const double dmax = fmax(fmax(fabs(dx), fabs(dy)), fabs(dz));
                const double r2 = dx * dx + dy * dy + dz * dz;
                const double dist2 = dist + (0.5 * (1.7320508075688772 - 1.0)) * nb.len;
                keep = !(dmax > dist) && !(r2 > dist2 * dist2);
            }
            const int cl = (keep && nc4.type == SHQ_PARTICLE_NODE_TYPE) ? nc4.count : 0;
            int kid[8], nk = 0;
            {
                int c = (keep && nc4.type == SHQ_NODE_NODE_TYPE) ? nc4.child : -1;
#pragma unroll
                for(int j = 0; j < 8; j++) {
                    const bool has = c >= 0 && c != nc4.sibling;
                    kid[j] = has ? c : -1;
                    if(has) {
                        nk++;
                        c = a.nodeC[c].sibling;
                    } else
                        c = -1;
                }
            }
            int totc = 0, totk = 0;
            const int offc = hb_scan8(cl, sh.wtot[0], totc); /* its barrier also orders the pops above before the pushes below */
            const int offk = hb_scan8(nk, sh.wtot[1], totk);
#pragma unroll
            for(int j = 0; j < 8; j++) {
                if(j < cl)
                    sh.cq[(chead + nc + offc + j) % HB_CAND] = nc4.child + j;
                if(j < nk && S - k + offk + j < HB_STACK)
                    sh.stk[S - k + offk + j] = kid[j];
            }
            if(tid == 0) {
                sh.nc = nc + totc;
                sh.S = (S - k + totk > HB_STACK) ? 0 : S - k + totk; /* deeper than the slack allows: stop (see heavy_walk) */
            }
        } else
            break;
    }
    return nint;
}

/* MODE 0: fused walk + evaluation (persistent grid, one list region per resident wave; also the redo path:
 * d_nq != NULL takes the queue length from the device).  MODE 1: walk only, lists and their lengths go to
 * global memory (one region per wave of the launch).  MODE 2: evaluation only, from those lists.  MODE 3: one target
 * per wave (heavy_walk) for the targets whose list did not fit; launched with 64 threads per block.  The
 * two-kernel path lets the walk run at twice the occupancy the register-heavy evaluation allows. */
template <int KT, int MODE, bool GHOSTS>
__device__ __forceinline__ void sph_density_body(const SphDev &a, const int32_t *queue, long long nq, int WindsDecouple,
                                                 unsigned long long *nint_total, int32_t *__restrict__ nlist, long long ntasks,
                                                 int32_t *__restrict__ counts, const long long *d_nq, const int4 *qseg)
{
    __shared__ __attribute__((aligned(32))) char lds[MODE == 4 ? sizeof(HbShared) : (MODE == 3 ? HW_LDS : (MODE == 2 ? 1 : (MODE == 1 ? SPH_WALK_WPB : 4) * NW_LDS_PER_WAVE(false)))];
    const int lane = threadIdx.x & 63;
    if((MODE == 0 || MODE >= 3) && d_nq) {
        nq = *d_nq;
        ntasks = MODE >= 3 ? nq : (nq + 255) / 256;
    }
    for(long long task = xcd_block(blockIdx.x, gridDim.x); task < ntasks; task += gridDim.x) {
    const long long wave = MODE == 4 ? task : task * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int32_t *myl = nlist + (MODE == 0 ? ((size_t) blockIdx.x * 4 + (threadIdx.x >> 6)) : (size_t) wave) * (size_t) (NL_ROWS * 64) + lane;
    const long long t = MODE >= 3 ? wave : wave * 64 + lane; /* MODE 3 / 4: every lane of the wave / workgroup works for the same target */
    bool valid = t < nq;
    long long pi = 0;
    double px = 0, py = 0, pz = 0, h = 0, vx = 0, vy = 0, vz = 0;
    int type = 0;
    if(valid) {
        pi = queue ? (long long) queue[t] : t;
        const double4 p = a.posm[pi];
        px = p.x; py = p.y; pz = p.z;
        h = a.hsml[pi];
        const double4 v = a.velp[pi];
        vx = v.x; vy = v.y; vz = v.z;
        type = a.pflags[pi] >> 4;
    }
    const Kern<KT> kernel(valid ? h : 1.0);
    const double h2 = h * h, Hinv = 1.0 / kernel.H, vol = kernel.volume();
    const bool nowind = WindsDecouple && type == 5; /* wind-decoupled neighbours are invisible to black holes */
    double Ngb = 0, Rho = 0, DhsmlDensity = 0, EgyRho = 0, DhsmlEgy = 0, Div = 0;
    double R0 = 0, R1 = 0, R2 = 0, G0 = 0, G1 = 0, G2 = 0;

    /* ngbiter, densitytree2.hpp:362-423; dist points from the neighbour to the target */
    auto pair = [&](const int s) {
        const double4 q = a.posm_leaf[s];
        const double4 w = a.velp_leaf[s];
        const double d0 = wrapd(px - q.x, a.Box, a.invBox);
        const double d1 = wrapd(py - q.y, a.Box, a.invBox);
        const double d2 = wrapd(pz - q.z, a.Box, a.invBox);
        const double r2 = d0 * d0 + d1 * d1 + d2 * d2;
        /* the lists of the f32 pre-test hold a superset: the reference's test (densitytree2.hpp:362-375) decides here */
        if(!(r2 < h2) || (nowind && (a.flag_leaf[s] & 2)))
            return;
        const double r = sqrt(r2);
        const double u = r * Hinv;
        const double wk = kernel.wk(u);
        const double dwk = kernel.dwk(u);
        Ngb += wk * vol;
        const double mj = q.w;
        Rho += mj * wk;
        const double dW = -(3 * wk * Hinv + u * dwk); /* DensityKrnl::dW, densitykernel.hpp:58-61 */
        DhsmlDensity += mj * dW;
        EgyRho += mj * w.w * wk;
        DhsmlEgy += mj * w.w * dW;
        if(r > 0) {
            const double fac = mj * dwk / r;
            const double e0 = vx - w.x, e1 = vy - w.y, e2 = vz - w.z;
            Div += -fac * (d0 * e0 + d1 * e1 + d2 * e2);
            R0 += fac * (e1 * d2 - e2 * d1);
            R1 += fac * (e2 * d0 - e0 * d2);
            R2 += fac * (e0 * d1 - e1 * d0);
            G0 += fac * d0;
            G1 += fac * d1;
            G2 += fac * d2;
        }
    };

    auto accept = [&](const double r2, const double, const int fl) { return r2 < h2 && !(nowind && (fl & 2)); };
    unsigned int nint = 0;
    int fill = 0;
    bool ovf = false;
    if(MODE == 4) {
        HbShared &sh = *reinterpret_cast<HbShared *>(lds);
        nint = heavy_block<false>(a, sh, px, py, pz, h, accept, pair);
        Ngb = hb_sum(Ngb, sh.red); Rho = hb_sum(Rho, sh.red); DhsmlDensity = hb_sum(DhsmlDensity, sh.red); EgyRho = hb_sum(EgyRho, sh.red);
        DhsmlEgy = hb_sum(DhsmlEgy, sh.red); Div = hb_sum(Div, sh.red); R0 = hb_sum(R0, sh.red); R1 = hb_sum(R1, sh.red); R2 = hb_sum(R2, sh.red);
        G0 = hb_sum(G0, sh.red); G1 = hb_sum(G1, sh.red); G2 = hb_sum(G2, sh.red);
        valid = threadIdx.x == 0;
    } else if(MODE == 3) {
        bool aborted = false;
        nint = heavy_walk<false>(a, lds, px, py, pz, h, accept, pair, aborted);
        if(aborted) { /* too much for one wave: a whole workgroup takes it */
            if(lane == 0)
                a.heavy2[atomicAdd((unsigned long long *) a.nheavy2, 1ull)] = (int32_t) pi;
            continue;
        }
        Ngb = wave_sum(Ngb); Rho = wave_sum(Rho); DhsmlDensity = wave_sum(DhsmlDensity); EgyRho = wave_sum(EgyRho); DhsmlEgy = wave_sum(DhsmlEgy);
        Div = wave_sum(Div); R0 = wave_sum(R0); R1 = wave_sum(R1); R2 = wave_sum(R2); G0 = wave_sum(G0); G1 = wave_sum(G1); G2 = wave_sum(G2);
        valid = lane == 0;
    } else if(MODE != 2)
        nint = ngb_walk<false, MODE == 1, GHOSTS, true>(a, lds + (threadIdx.x >> 6) * NW_LDS_PER_WAVE(false), myl, valid, px, py, pz, h, accept, pair,
                                                        (unsigned int *) nullptr, fill, ovf,
                                                        (GHOSTS && valid) ? qseg[t] : make_int4(-1, -1, -1, -1),
                                                        valid ? pre32_bound(h, a.Box * 0x1p-20) : 0.f);
    if(MODE == 1) {
        counts[wave * 64 + lane] = ovf ? -1 : fill; /* -1: this target goes to the one-target-per-wave kernel */
        if(ovf)
            nint = 0; /* counted there */
    }
    if(MODE == 2) {
        fill = counts[wave * 64 + lane];
        if(fill < 0) { /* walked on its own (MODE 3) */
            fill = 0;
            valid = false;
        }
        nl_flush(myl, fill, pair);
    }
    if(MODE != 1 && valid) {
        /* DensityResult::reduce<PRIMARY>, densitytree2.hpp:308-343 */
        a.numngb[pi] = Ngb;
        a.dhsmldens[pi] = DhsmlDensity;
        a.rho[pi] = Rho;
        a.div[pi] = Div;
        if(type == 0 || GHOSTS) { /* an imported query returns every sum; its owner's reduce picks by type */
            a.rot[3 * pi] = R0;
            a.rot[3 * pi + 1] = R1;
            a.rot[3 * pi + 2] = R2;
            if(a.gradrho) {
                a.gradrho[3 * pi] = G0;
                a.gradrho[3 * pi + 1] = G1;
                a.gradrho[3 * pi + 2] = G2;
            }
            a.egyrho[pi] = EgyRho;
            a.dhsmlegy[pi] = DhsmlEgy;
        }
    }
    unsigned int sn = nint;
    for(int off = 32; off > 0; off >>= 1)
        sn += __shfl_xor(sn, off);
    if(lane == 0 && nint_total)
        atomicAdd(nint_total, (unsigned long long) sn);
    } /* task loop */
}

template <int KT, int MODE, bool GHOSTS = false>
__global__ __launch_bounds__(256) void sph_density_kernel(const SphDev a, const int32_t *queue, long long nq, int WindsDecouple,
                                                          unsigned long long *nint_total, int32_t *__restrict__ nlist, long long ntasks,
                                                          int32_t *__restrict__ counts, const long long *d_nq, const int4 *qseg = nullptr)
{
    sph_density_body<KT, MODE, GHOSTS>(a, queue, nq, WindsDecouple, nint_total, nlist, ntasks, counts, d_nq, qseg);
}

/* MODE 4: one target per workgroup of 512 threads */
template <int KT>
__global__ __launch_bounds__(HB_THREADS) void sph_density_block_kernel(const SphDev a, const int32_t *queue, int WindsDecouple,
                                                                       unsigned long long *nint_total, const long long *d_nq)
{
    sph_density_body<KT, 4, false>(a, queue, 0, WindsDecouple, nint_total, nullptr, 0, nullptr, d_nq, nullptr);
}

/* ---- density postprocess + Hsml update (DensityOutput::postprocess, density_check_neighbours) -- */
struct PostArgs {
    double Box, DesNumNgb, DesNumNgbBH, MinGasHsml, MaxDev;
    int update_hsml, BlackHoleOn, DoEgyDensity;
    unsigned long long *hmax_tried; /* the largest Hsml any walk of this loop has run with, as the bits of a non-negative double */
};

__global__ void sph_density_post_kernel(const SphDev a, const int32_t *queue, long long nq, const PostArgs p, int32_t *todo)
{
    const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= nq)
        return;
    const long long i = queue ? (long long) queue[t] : t;
    const int type = a.pflags[i] >> 4;
    int done = 0;
    const double density = a.rho[i];
    double hs = a.hsml[i];
    /* the radius the walk behind this call searched: a sharded caller's halo must cover the largest one TRIED, not just the one the
     * loop ends with (an intermediate guess that reaches past the imported ghosts undercounts NumNgb and steers the next guess) */
    if(p.hmax_tried && (unsigned long long) __double_as_longlong(hs) > *(volatile unsigned long long *) p.hmax_tried)
        atomicMax(p.hmax_tried, (unsigned long long) __double_as_longlong(hs));
    double DhsmlDens = a.dhsmldens[i];
    DhsmlDens *= hs / (3 * density);
    DhsmlDens = 1 / (1 + DhsmlDens);
    a.dhsmldens[i] = DhsmlDens;
    if(p.update_hsml) {
        double desnumngb = p.DesNumNgb;
        if(p.BlackHoleOn && type == 5)
            desnumngb = p.DesNumNgbBH;
        const double NumNgb = a.numngb[i];
        double L = a.left[i], R = a.right[i];
        if(NumNgb < (desnumngb - p.MaxDev) || (NumNgb > (desnumngb + p.MaxDev))) {
            if((R - L) < 1.0e-5 * R) {
                hs = R;
                done = 1;
            } else {
                if(NumNgb < desnumngb)
                    L = hs;
                else
                    R = hs;
                if((R < p.Box && L > 0) || (hs * 1.26 > 0.99 * p.Box))
                    hs = cbrt(0.5 * (L * L * L + R * R * R));
                else {
                    const double DensFac = DhsmlDens;
                    double fac = 1.26;
                    if(NumNgb > 0)
                        fac = 1 - (NumNgb - desnumngb) / (3 * NumNgb) * DensFac;
                    if(R > 0.99 * p.Box && L > 0)
                        if(DensFac <= 0 || fabs(NumNgb - desnumngb) >= 0.5 * desnumngb || fac > 1.26)
                            fac = 1.26;
                    if(R < 0.99 * p.Box && L == 0)
                        if(DensFac <= 0 || fac < 1. / 3)
                            fac = 1. / 3;
                    hs *= fac;
                }
                if(R < p.MinGasHsml) {
                    hs = p.MinGasHsml;
                    done = 1;
                } else
                    done = 0;
            }
            a.left[i] = L;
            a.right[i] = R;
        } else {
            if(hs < p.MinGasHsml)
                hs = p.MinGasHsml;
            done = 1;
        }
        a.hsml[i] = hs;
    }
    if(type == 0) {
        if(p.DoEgyDensity) {
            const double EntPred = a.velp[i].w;
            double d = a.dhsmlegy[i];
            d *= hs / (3 * a.egyrho[i]);
            d *= -DhsmlDens;
            a.dhsmlegy[i] = d;
            a.egyrho[i] = a.egyrho[i] / EntPred;
        } else
            a.dhsmlegy[i] = DhsmlDens;
        const double r0 = a.rot[3 * i], r1 = a.rot[3 * i + 1], r2 = a.rot[3 * i + 2];
        a.curl[i] = sqrt(r0 * r0 + r1 * r1 + r2 * r2) / density;
        const double dv = a.div[i] / density;
        a.div[i] = dv;
        a.dthsml[i] = (1.0 / 3) * dv * hs;
    } else if(type == 5) {
        const double dv = a.div[i] / density;
        a.div[i] = dv;
        a.dthsml[i] = (1.0 / 3) * dv * hs;
    }
    if(todo)
        todo[t] = done ? -1 : (int32_t) i;
    if(done && p.update_hsml && type == 0 && a.pfather) {
        /* update_tree_hmax_father, forcetree.cpp:1285-1313; non-negative doubles order like their bits */
        const int no = a.pfather[i];
        if(no >= 0) {
            const NodeB B = a.nodeB[no];
            const double4 P = a.posm[i];
            double nh = 0;
            nh = fmax(nh, fabs(P.x - B.center[0]) + hs - B.len / 2.);
            nh = fmax(nh, fabs(P.y - B.center[1]) + hs - B.len / 2.);
            nh = fmax(nh, fabs(P.z - B.center[2]) + hs - B.len / 2.);
            atomicMax(reinterpret_cast<unsigned long long *>(&a.hmax[no]), (unsigned long long) __double_as_longlong(nh));
        }
    }
}

/* ---- order-preserving compaction of the redo queue (three small kernels) --------------------------- */
__global__ void compact_count_kernel(const int32_t *todo, long long n, int32_t *blockcount)
{
    __shared__ int s;
    if(threadIdx.x == 0)
        s = 0;
    __syncthreads();
    const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    const bool f = (t < n) && (todo[t] >= 0);
    const unsigned long long m = shq_ballot(f);
    if((threadIdx.x & 63) == 0)
        atomicAdd(&s, __popcll(m));
    __syncthreads();
    if(threadIdx.x == 0)
        blockcount[blockIdx.x] = s;
}
__global__ void compact_scan_kernel(int32_t *blockcount, int nblocks, long long *total)
{
    /* single workgroup exclusive scan */
    __shared__ long long carry;
    __shared__ int buf[1024];
    if(threadIdx.x == 0)
        carry = 0;
    __syncthreads();
    for(int base = 0; base < nblocks; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = (i < nblocks) ? blockcount[i] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for(int off = 1; off < 1024; off <<= 1) {
            int add = (threadIdx.x >= off) ? buf[threadIdx.x - off] : 0;
            __syncthreads();
            buf[threadIdx.x] += add;
            __syncthreads();
        }
        const int incl = buf[threadIdx.x];
        if(i < nblocks)
            blockcount[i] = (int32_t) (carry + incl - v);
        __syncthreads();
        if(threadIdx.x == 1023)
            carry += incl;
        __syncthreads();
    }
    if(threadIdx.x == 0)
        *total = carry;
}
__global__ void compact_write_kernel(const int32_t *todo, long long n, const int32_t *blockoff, int32_t *out)
{
    __shared__ int wavebase[4];
    const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    const bool f = (t < n) && (todo[t] >= 0);
    const unsigned long long m = shq_ballot(f);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if(lane == 0)
        wavebase[w] = __popcll(m);
    __syncthreads();
    int base = blockoff[blockIdx.x];
    for(int k = 0; k < w; k++)
        base += wavebase[k];
    if(f) {
        const int rank = __popcll(m & ((1ull << lane) - 1ull));
        out[base + rank] = todo[t];
    }
}

/* ---- hydro walk (HydroLocalTreeWalk::ngbiter, hydratree2.hpp:253-378) ----------------------------- */
struct HydroConst {
    double hubble_a2, fac_mu, fac_vsic_fix, ArtBulkViscConst, contrast;
    int DISPH;
};

/* MODE as for sph_density_kernel */
template <int KT, int MODE, bool GHOSTS>
__device__ __forceinline__ void sph_hydro_body(const SphDev &a, const int32_t *queue, long long nq, const HydroConst &hc,
                                               unsigned long long *nint_total, int32_t *__restrict__ nlist, long long ntasks,
                                               int32_t *__restrict__ counts, const long long *d_nq, const int4 *qseg)
{
    __shared__ __attribute__((aligned(32))) char lds[MODE == 4 ? sizeof(HbShared) : (MODE == 3 ? HW_LDS : (MODE == 2 ? 1 : (MODE == 1 ? SPH_WALK_WPB : 4) * NW_LDS_PER_WAVE(true)))];
    const int lane = threadIdx.x & 63;
    if((MODE == 0 || MODE >= 3) && d_nq) {
        nq = *d_nq;
        ntasks = MODE >= 3 ? nq : (nq + 255) / 256;
    }
    for(long long task = xcd_block(blockIdx.x, gridDim.x); task < ntasks; task += gridDim.x) {
    const long long wave = MODE == 4 ? task : task * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int32_t *myl = nlist + (MODE == 0 ? ((size_t) blockIdx.x * 4 + (threadIdx.x >> 6)) : (size_t) wave) * (size_t) (NL_ROWS * 64) + lane;
    const long long t = MODE >= 3 ? wave : wave * 64 + lane; /* MODE 3 / 4: every lane of the wave / workgroup works for the same target */
    bool valid = t < nq;
    long long pi = 0;
    double px = 0, py = 0, pz = 0, mi = 0, hi = 1, vx = 0, vy = 0, vz = 0;
    double4 Ci = make_double4(1, 1, 1, 1), Di = make_double4(0, 0, 0, 0);
    if(valid) {
        pi = queue ? (long long) queue[t] : t;
        const double4 p = a.posm[pi];
        px = p.x; py = p.y; pz = p.z; mi = p.w;
        hi = a.hsml[pi];
        const double4 v = a.velp[pi];
        vx = v.x; vy = v.y; vz = v.z;
        Ci = a.hydC[pi];
        Di = a.hydD[pi];
    }
    /* HydroQuery ctor, hydratree2.hpp:165-191: for the target itself the un-drifted values apply, but a
     * target is active, so its drift factor is zero and the predicted values coincide. */
    const double iEntVarPred = Ci.x, iDensity = Ci.y, soundspeed_i = Ci.z, p_over_rho2_i = Ci.w;
    const double iDhsml = Di.x, iF1 = Di.z, idloga = Di.w;
    /* rr1 = EgyRho / Density with the contrast limit; Di.y holds exactly that for the target */
    const double rr1 = Di.y;
    const Kern<KT> kernel_i(hi);
    const double hi2 = hi * hi;
    const double si = (Kern<KT>::support / 2.) / hi, dnorm_i = kernel_i.dnorm, inv_iEVP = 1.0 / iEntVarPred;
    double A0 = 0, A1 = 0, A2 = 0, DtE = 0;
    double MaxSig = soundspeed_i; /* HydroResult ctor: sqrt(GAMMA P / EgyRho) */

    /* HydroLocalTreeWalk::ngbiter, hydratree2.hpp:253-378, for one accepted neighbour (leaf slot s) */
    auto pair = [&](const int s) {
        const HydRec *rec = a.hydrec_leaf + s;
        const double4 q = rec->posm;
        const double4 w = rec->velh;
        const double4 Cj = rec->C;
        const double4 Dj = rec->D;
        const double hj = w.w;
        const double d0 = wrapd(px - q.x, a.Box, a.invBox);
        const double d1 = wrapd(py - q.y, a.Box, a.invBox);
        const double d2 = wrapd(pz - q.z, a.Box, a.invBox);
        const double r2 = d0 * d0 + d1 * d1 + d2 * d2;
        /* the lists of the f32 pre-test hold a superset: the reference's test (hydratree2.hpp:253-262) decides here (a wind-decoupled
         * particle never reaches a list: its f32 record is at infinity, and the f64 scan's accept() knows its flag) */
        if(!(r2 > 0 && (r2 < hi2 || r2 < hj * hj)))
            return;
        /* The reference divides by r, H and the entropy variables at every use; with ~10 f64 divisions
         * (a dozen instructions each) they were half of this function.  Here 1/r comes from v_rsq_f64 + a
         * Newton step and the kernel of j from one reciprocal of h_j: same formulas, results differ from
         * the oracle's at the 1e-16 level. */
        const double EVP = Cj.x, density_j = Cj.y, soundspeed_j = Cj.z, p_over_rho2_j = Cj.w;
        double vsig = soundspeed_i + soundspeed_j;
        if(vsig > MaxSig)
            MaxSig = vsig;
        const double e0 = vx - w.x, e1 = vy - w.y, e2 = vz - w.z;
        const double vdotr = d0 * e0 + d1 * e1 + d2 * e2;
        const double vdotr2 = vdotr + hc.hubble_a2 * r2;
        const double y0 = __builtin_amdgcn_rsq(r2);
        const double ye = fma(-r2 * y0, y0, 1.0);
        const double rinv = fma(y0 * ye, fma(ye, 0.375, 0.5), y0);
        const double r = r2 * rinv;
        const double sj = (Kern<KT>::support / 2.) / hj; /* q = r * support / (2 H) */
        const double sigma = (KT == 1) ? (1 / M_PI) : ((KT == 2) ? (1 / (120 * M_PI)) : (1 / (20 * M_PI)));
        const double dnorm_j = sigma * (sj * sj) * (sj * sj);
        const double dwk_i = dnorm_i * kernel_i.dwk_int(r * si);
        const double dwk_j = dnorm_j * kernel_i.dwk_int(r * sj);
        double visc = 0;
        if(vdotr2 < 0) {
            const double mu_ij = hc.fac_mu * vdotr2 * rinv;
            const double rho_ij = 0.5 * (iDensity + density_j);
            vsig = soundspeed_i + soundspeed_j - 3 * mu_ij;
            if(vsig > MaxSig)
                MaxSig = vsig;
            visc = 0.25 * hc.ArtBulkViscConst * vsig * (-mu_ij) / rho_ij * (iF1 + Dj.z);
            const double dloga = 2 * fmax(idloga, Dj.w);
            if(dloga > 0 && (dwk_i + dwk_j) < 0) {
                if((mi + q.w) > 0)
                    visc = fmin(visc, 0.5 * hc.fac_vsic_fix * vdotr2 / (0.5 * (mi + q.w) * (dwk_i + dwk_j) * r * dloga));
            }
        }
        const double mr = q.w * rinv;
        const double hfc_visc = 0.5 * mr * visc * (dwk_i + dwk_j);
        double hfc = hfc_visc;
        if(hc.DISPH) {
            const double ratio = EVP * inv_iEVP; /* EVP_j / EVP_i */
            hfc += mr * (dwk_i * p_over_rho2_i * ratio + dwk_j * p_over_rho2_j / ratio);
        }
        hfc += mr * (p_over_rho2_i * iDhsml * dwk_i * rr1 + p_over_rho2_j * Dj.x * dwk_j * Dj.y);
        A0 += -hfc * d0;
        A1 += -hfc * d1;
        A2 += -hfc * d2;
        DtE += 0.5 * hfc_visc * vdotr2;
    };

    auto accept = [&](const double r2, const double hj, const int fl) { return r2 > 0 && (r2 < hi2 || r2 < hj * hj) && !(fl & 2); };
    unsigned int dbgc[3] = {0, 0, 0};
    unsigned int nint = 0;
    int fill = 0;
    bool ovf = false;
    if(MODE == 4) {
        HbShared &sh = *reinterpret_cast<HbShared *>(lds);
        nint = heavy_block<true>(a, sh, px, py, pz, hi, accept, pair);
        A0 = hb_sum(A0, sh.red); A1 = hb_sum(A1, sh.red); A2 = hb_sum(A2, sh.red); DtE = hb_sum(DtE, sh.red);
        MaxSig = hb_max(MaxSig, sh.red);
        valid = threadIdx.x == 0;
    } else if(MODE == 3) {
        bool aborted = false;
        nint = heavy_walk<true>(a, lds, px, py, pz, hi, accept, pair, aborted);
        if(aborted) { /* too much for one wave: a whole workgroup takes it */
            if(lane == 0)
                a.heavy2[atomicAdd((unsigned long long *) a.nheavy2, 1ull)] = (int32_t) pi;
            continue;
        }
        A0 = wave_sum(A0); A1 = wave_sum(A1); A2 = wave_sum(A2); DtE = wave_sum(DtE);
        for(int off = 32; off > 0; off >>= 1)
            MaxSig = fmax(MaxSig, __shfl_xor(MaxSig, off));
        valid = lane == 0;
    } else if(MODE != 2)
        nint = ngb_walk<true, MODE == 1, GHOSTS, true>(a, lds + (threadIdx.x >> 6) * NW_LDS_PER_WAVE(true), myl, valid, px, py, pz, hi, accept, pair,
                                                       (MODE == 0 && nint_total && !GHOSTS) ? dbgc : (unsigned int *) nullptr, fill, ovf,
                                                       (GHOSTS && valid) ? qseg[t] : make_int4(-1, -1, -1, -1),
                                                       valid ? pre32_bound(hi, a.Box * 0x1p-20) : 0.f);
    if(MODE == 1) {
        counts[wave * 64 + lane] = ovf ? -1 : fill; /* -1: this target goes to the one-target-per-wave kernel */
        if(ovf)
            nint = 0; /* counted there */
    }
    if(MODE == 2) {
        fill = counts[wave * 64 + lane];
        if(fill < 0) { /* walked on its own (MODE 3) */
            fill = 0;
            valid = false;
        }
        nl_flush(myl, fill, pair);
    }
    if(MODE != 1 && valid) {
        a.hacc[3 * pi] = A0;
        a.hacc[3 * pi + 1] = A1;
        a.hacc[3 * pi + 2] = A2;
        a.dtent[pi] = DtE;
        a.maxsig[pi] = MaxSig;
    }
    unsigned int sn = nint;
    for(int off = 32; off > 0; off >>= 1)
        sn += __shfl_xor(sn, off);
    if(lane == 0 && nint_total)
        atomicAdd(nint_total, (unsigned long long) sn);
    if(MODE == 0 && nint_total && !GHOSTS) { /* diagnostics behind SHQ_SPH_DEBUG: [1] nodes/wave [2] candidates/wave [3] pairs (lanes) [4] flush rounds/wave */
        if(lane == 0) {
            atomicAdd(nint_total + 1, (unsigned long long) dbgc[0]);
            atomicAdd(nint_total + 2, (unsigned long long) dbgc[1]);
            atomicAdd(nint_total + 4, (unsigned long long) dbgc[2]);
        }
    }
    } /* task loop */
}

template <int KT, int MODE, bool GHOSTS = false>
__global__ __launch_bounds__(256, 4) void sph_hydro_kernel(const SphDev a, const int32_t *queue, long long nq, const HydroConst hc,
                                                           unsigned long long *nint_total, int32_t *__restrict__ nlist, long long ntasks,
                                                           int32_t *__restrict__ counts, const long long *d_nq, const int4 *qseg = nullptr)
{
    sph_hydro_body<KT, MODE, GHOSTS>(a, queue, nq, hc, nint_total, nlist, ntasks, counts, d_nq, qseg);
}

/* MODE 4: one target per workgroup of 512 threads */
template <int KT>
__global__ __launch_bounds__(HB_THREADS) void sph_hydro_block_kernel(const SphDev a, const int32_t *queue, const HydroConst hc,
                                                                     unsigned long long *nint_total, const long long *d_nq)
{
    sph_hydro_body<KT, 4, false>(a, queue, 0, hc, nint_total, nullptr, 0, nullptr, d_nq, nullptr);
}

/* HydroOutput::postprocess, hydratree2.hpp:134-148 + winds_decoupled_hydro, winds.h:60-68 */
__global__ void sph_hydro_post_kernel(const SphDev a, const int32_t *queue, long long nq, const double *density,
                                      const double *delay, double hubble_a2, double atime, double WindSpeed, double WindThresh)
{
    const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= nq)
        return;
    const long long i = queue ? (long long) queue[t] : t;
    double de = a.dtent[i];
    de *= SPH_GAMMA_MINUS1 / (hubble_a2 * pow(density[i], SPH_GAMMA_MINUS1));
    if(delay && delay[i] > 0) {
        a.hacc[3 * i] = 0;
        a.hacc[3 * i + 1] = 0;
        a.hacc[3 * i + 2] = 0;
        de = 0;
        double windspeed = WindSpeed * atime;
        const double fac_mu = pow(atime, 3 * (SPH_GAMMA - 1) / 2) / atime;
        windspeed *= fac_mu;
        const double hsml_c = cbrt(WindThresh / density[i]) * atime;
        a.maxsig[i] = hsml_c * fmax(2 * windspeed, a.maxsig[i]);
    }
    a.dtent[i] = de;
}

__global__ void fill_kernel(double *x, long long n, double v)
{
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n)
        x[i] = v;
}
__global__ void gradmag_kernel(const double *g, double *out, long long n)
{
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n)
        out[i] = sqrt(g[3 * i] * g[3 * i] + g[3 * i + 1] * g[3 * i + 1] + g[3 * i + 2] * g[3 * i + 2]);
}

inline unsigned nblk(long long n, int t = 256) { return (unsigned) ((n + t - 1) / t); }

SphDev make_dev(shq_context *ctx)
{
    SphDev a;
    a.nodeB = ctx->nodeB.ptr;
    a.nodeC = ctx->nodeC.ptr;
    a.hmax = ctx->node_hmax.ptr;
    a.pfather = ctx->pfather.ptr;
    a.root = ctx->root;
    a.npool = (int) ctx->numnodes;
    a.posm_leaf = ctx->posm_leaf.ptr;
    a.velp_leaf = ctx->velp_leaf.ptr;
    a.hydrec_leaf = reinterpret_cast<const HydRec *>(ctx->hydrec_leaf.ptr);
    a.hsml_leaf = ctx->hsml_leaf.ptr;
    a.flag_leaf = ctx->flag_leaf.ptr;
    a.posf_leaf = ctx->posf_leaf.ptr;
    a.ngarb_leaf = ctx->ngarb_leaf.ptr;
    a.posm = ctx->posm.ptr;
    a.pflags = ctx->pflags.ptr;
    a.hsml = ctx->hsml.ptr;
    a.dthsml = ctx->dthsml.ptr;
    a.velp = ctx->velp.ptr;
    a.hydC = ctx->hydC.ptr;
    a.hydD = ctx->hydD.ptr;
    a.numngb = ctx->s_numngb.ptr;
    a.dhsmldens = ctx->s_dhsmldens.ptr;
    a.left = ctx->s_left.ptr;
    a.right = ctx->s_right.ptr;
    a.rho = ctx->g_density.ptr;
    a.egyrho = ctx->g_egywt.ptr;
    a.dhsmlegy = ctx->g_dhsmlegy.ptr;
    a.div = ctx->g_divvel.ptr;
    a.curl = ctx->g_curlvel.ptr;
    a.rot = ctx->s_rot.ptr;
    a.gradrho = nullptr;
    a.hacc = ctx->g_hydroaccel_out.ptr;
    a.dtent = ctx->g_dtentropy_out.ptr;
    a.maxsig = ctx->g_maxsignalvel.ptr;
    a.heavy2 = ctx->s_redo2.ptr;               /* null until a walk reserved it: the wave tier then never gives up */
    a.nheavy2 = ctx->s_counters.ptr ? ctx->s_counters.ptr + 7 : nullptr;
    a.Box = ctx->treeBox;
    a.invBox = 1.0 / ctx->treeBox;
    return a;
}

} // namespace

/* Prepass + leaf gather: fills velp (and hydC/hydD when hp != NULL) and their leaf-order copies. */
int shq_sph_prepare(shq_context *ctx, const shq_kick_factors *kf, const shq_hydro_params *hp, const double *d_evp_in)
{
    const long long n = ctx->numpart;
    SHQ_TRY(ctx->velp.reserve(n > 0 ? n : 1));
    SHQ_TRY(ctx->hydC.reserve(n > 0 ? n : 1));
    SHQ_TRY(ctx->hydD.reserve(n > 0 ? n : 1));
    const long long nl = ctx->ntreeparts + SHQ_NMAXCHILD;
    SHQ_CHECK(nl < (1ll << 30), SHQ_ERR_INVALID, "SPH walk: more than 2^30 leaf slots (the candidate tile keeps two flag bits on top of the slot)");
    SHQ_TRY(ctx->velp_leaf.reserve(nl));
    SHQ_TRY(ctx->hydrec_leaf.reserve(nl * sizeof(HydRec) + 128));
    SHQ_TRY(ctx->hsml_leaf.reserve(nl));
    SHQ_TRY(ctx->flag_leaf.reserve(nl));
    SHQ_TRY(ctx->posf_leaf.reserve(nl));
    SHQ_TRY(ctx->ngarb_leaf.reserve(nl));
    if(n == 0)
        return SHQ_OK;
    PredArgs a;
    a.n = n;
    a.pflags = ctx->pflags.ptr;
    a.vel = ctx->vel.ptr;
    a.treeacc = ctx->treeacc.ptr;
    a.gravpm = ctx->gravpm.ptr;
    a.hydroaccel = ctx->g_hydroaccel.ptr;
    a.bin_grav = ctx->bin_grav.ptr;
    a.bin_hydro = ctx->bin_hydro.ptr;
    a.entropy = ctx->g_entropy.ptr;
    a.dtentropy = ctx->g_dtentropy.ptr;
    a.hsml = ctx->hsml.ptr;
    a.density = ctx->g_density.ptr;
    a.egywt = ctx->g_egywt.ptr;
    a.dhsmlegy = ctx->g_dhsmlegy.ptr;
    a.divvel = ctx->g_divvel.ptr;
    a.curlvel = ctx->g_curlvel.ptr;
    a.velp = ctx->velp.ptr;
    a.hydC = ctx->hydC.ptr;
    a.hydD = ctx->hydD.ptr;
    a.evp_in = d_evp_in;
    a.kf = *kf;
    a.hydro = hp ? 1 : 0;
    a.DISPH = hp ? hp->DensityIndependentSphOn : 0;
    a.fac_mu = hp ? hp->fac_mu : 1;
    a.contrast = hp ? hp->DensityContrastLimit : 0;
    for(int i = 0; i <= SHQ_TIMEBINS; i++)
        a.drifts[i] = hp ? hp->drifts[i] : 0;
    sph_predict_kernel<<<dim3(nblk(n)), dim3(256), 0, ctx->stream>>>(a);
    sph_gather_leaf_kernel<<<dim3(nblk(nl)), dim3(256), 0, ctx->stream>>>(
        nl, ctx->leaf_pidx.ptr, ctx->velp.ptr, hp ? ctx->hydC.ptr : nullptr, hp ? ctx->hydD.ptr : nullptr, ctx->hsml.ptr,
        ctx->pflags.ptr, ctx->g_delaytime.ptr, ctx->velp_leaf.ptr, hp ? reinterpret_cast<HydRec *>(ctx->hydrec_leaf.ptr) : nullptr,
        ctx->hsml_leaf.ptr, ctx->flag_leaf.ptr, ctx->posm_leaf.ptr, ctx->posf_leaf.ptr, ldexp(ctx->treeBox, -20));
    if(ctx->numnodes > 0)
        sph_leaf_ngarb_kernel<<<dim3(nblk(ctx->numnodes)), dim3(256), 0, ctx->stream>>>((int) ctx->numnodes, ctx->nodeC.ptr, ctx->flag_leaf.ptr,
                                                                                      ctx->ngarb_leaf.ptr);
    SHQ_HIP(hipGetLastError());
    return SHQ_OK;
}

/* ---- launch: two kernels per chunk of targets + redo of overflowed waves ---------------------------- */
#define NL_CHUNK (1ll << 22)  /* targets per chunk: 4 Mi x NL_CAP x 4 B = 4 GB of list scratch */
#define NL_REDO_BLOCKS 1024
#define NL_BLOCK_BLOCKS 512  /* 512-thread workgroups of the one-target-per-workgroup kernel */
#define NL_HEAVY_BLOCKS 8192 /* single-wave workgroups of the one-target-per-wave kernel (it takes its queue length from the device) */

/* targets of the chunk whose lists overflowed: append them to the queue of the one-target-per-wave kernel */
__global__ void sph_collect_redo_kernel(const int32_t *__restrict__ counts, const int32_t *__restrict__ queue, long long nq, int32_t *redo,
                                        long long *nredo)
{
    const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= nq || counts[t] >= 0)
        return;
    const long long at = (long long) atomicAdd((unsigned long long *) nredo, 1ull);
    redo[at] = queue ? queue[t] : (int32_t) t;
}

static long long nl_chunk_waves(long long nq)
{
    const long long chunk = nq < NL_CHUNK ? nq : NL_CHUNK;
    return ((chunk + 255) / 256) * 4;
}

/* Walk and evaluation in a pipeline (round 4).  The walk kernel is bound by the vector and scalar pipes (tools/sph_ab.sh with the
 * SPH_PROBE builds: 3.2 ms of node tests + 1.5 ms of candidate tests for the 128^3 hydro pass), the evaluation kernel by the texture
 * addresser (eight to ten 16-byte gathers per pair from 64 different lines, 37 % VALU busy): with the targets cut into NL_PIPE pieces, the
 * evaluation of piece k runs on a second stream beside the walk of piece k + 1, two list regions in turn.  SHQ_SPH_PIPE=0: one after the
 * other, as before (the same lists, the same sums). */
#define NL_PIPE 4
#define NL_PIPE_MIN (1ll << 19)
static bool nl_pipelined(const int32_t *q, long long nq)
{
    static const bool on = getenv("SHQ_SPH_PIPE") && atoi(getenv("SHQ_SPH_PIPE")) != 0;
    return on && q != nullptr && nq > NL_PIPE_MIN;
}
static long long nl_pipe_targets(long long nq)
{
    const long long c = (((nq + NL_PIPE - 1) / NL_PIPE) + 255) / 256 * 256;
    return c < NL_CHUNK ? c : NL_CHUNK;
}

/* list scratch: one region per wave of a chunk (both layouts fit: two pieces of a quarter each, or the whole), followed by one per
 * wave of the fused redo kernel */
static int reserve_nlist(shq_context *ctx, long long nq)
{
    const long long waves = nl_chunk_waves(nq) + NL_REDO_BLOCKS * 4;
    SHQ_TRY(ctx->s_nlist.reserve((size_t) waves * NL_ROWS * 64));
    SHQ_TRY(ctx->s_ncount.reserve((size_t) (nl_chunk_waves(nq) + 8) * 64));
    SHQ_TRY(ctx->s_redo.reserve((size_t) (nq > 0 ? nq : 1)));
    SHQ_TRY(ctx->s_redo2.reserve((size_t) (nq > 0 ? nq : 1)));
    for(int i = 0; i < 4; i++)
        if(!ctx->ev_sph[i])
            SHQ_HIP(hipEventCreateWithFlags(&ctx->ev_sph[i], hipEventDisableTiming));
    return SHQ_OK;
}

/* Runs walk<1> and eval<2> over [q, q + nq) in chunks, then the one-target-per-wave kernel <3> over the targets whose
 * lists overflowed (their number is only known on the device: the kernel reads it there). */
template <class LaunchW, class LaunchP, class LaunchF, class LaunchB>
static int launch_two_kernel(shq_context *ctx, const int32_t *q, long long nq, long long nq_reserved, LaunchW &&walk, LaunchP &&eval,
                             LaunchF &&fused, LaunchB &&block)
{
    long long *d_nredo = ctx->s_counters.ptr + 6;
    SHQ_HIP(hipMemsetAsync(d_nredo, 0, 2 * sizeof(long long), ctx->stream)); /* [6] targets for a wave of their own, [7] for a workgroup */
    int32_t *lists = ctx->s_nlist.ptr;
    int32_t *fused_lists = ctx->s_nlist.ptr + (size_t) nl_chunk_waves(nq_reserved) * NL_ROWS * 64;
    SHQ_CHECK(q || nq <= NL_CHUNK, SHQ_ERR_INVALID, "SPH walk: more than %lld targets need an explicit queue", (long long) NL_CHUNK);
    const bool pipe = nl_pipelined(q, nq) && ctx->stream_pair;
    const long long chunk = pipe ? nl_pipe_targets(nq) : NL_CHUNK;
    /* pipelined: two regions of a piece's waves each (a piece is at most a quarter of what was reserved, rounded up to 256 targets) */
    const size_t region = pipe ? (size_t) ((chunk + 255) / 256 * 4) * NL_ROWS * 64 : 0;
    const size_t cregion = pipe ? (size_t) ((chunk + 255) / 256 * 4 + 4) * 64 : 0;
    static const bool hi = getenv("SHQ_SPH_PIPE") && atoi(getenv("SHQ_SPH_PIPE")) == 2; /* the evaluation on the high-priority stream */
    hipStream_t sw = ctx->stream, se = pipe ? (hi && ctx->stream_pm ? ctx->stream_pm : ctx->stream_pair) : ctx->stream;
    int k = 0;
    for(long long off = 0; off < nq; off += chunk, k++) {
        const long long m = (nq - off < chunk) ? nq - off : chunk;
        const long long ntasks = (m + 255) / 256;
        const int32_t *qc = q ? q + off : nullptr;
        const long long wtasks = (m + 64 * SPH_WALK_WPB - 1) / (64 * SPH_WALK_WPB);
        const int b = k & 1;
        int32_t *lb = lists + b * region, *cb = ctx->s_ncount.ptr + b * cregion;
        if(pipe && k >= 2)
            SHQ_HIP(hipStreamWaitEvent(sw, ctx->ev_sph[2 + b], 0)); /* the evaluation of piece k - 2 is done with this region */
        walk(sw, (unsigned) wtasks, qc, m, wtasks, lb, cb);
        if(pipe) {
            SHQ_HIP(hipEventRecord(ctx->ev_sph[b], sw));
            SHQ_HIP(hipStreamWaitEvent(se, ctx->ev_sph[b], 0));
        }
        eval(se, (unsigned) ntasks, qc, m, ntasks, lb, cb);
        sph_collect_redo_kernel<<<dim3(nblk(m)), dim3(256), 0, se>>>(cb, qc, m, ctx->s_redo.ptr, d_nredo);
        if(pipe)
            SHQ_HIP(hipEventRecord(ctx->ev_sph[2 + b], se));
    }
    if(pipe)
        for(int b = 0; b < 2 && b < k; b++)
            SHQ_HIP(hipStreamWaitEvent(sw, ctx->ev_sph[2 + b], 0));
    fused((unsigned) NL_HEAVY_BLOCKS, ctx->s_redo.ptr, fused_lists, d_nredo);
    block((unsigned) NL_BLOCK_BLOCKS, ctx->s_redo2.ptr, d_nredo + 1);
    SHQ_HIP(hipGetLastError());
    return SHQ_OK;
}

template <int KT>
static int launch_density(shq_context *ctx, const SphDev &a, const int32_t *q, long long nq, long long nq_reserved, int wd,
                          unsigned long long *nint)
{
    hipStream_t st = ctx->stream;
    return launch_two_kernel(
        ctx, q, nq, nq_reserved,
        [&](hipStream_t s, unsigned grid, const int32_t *qc, long long m, long long ntasks, int32_t *lists, int32_t *counts) {
            sph_density_kernel<KT, 1><<<dim3(grid), dim3(64 * SPH_WALK_WPB), 0, s>>>(a, qc, m, wd, nint, lists, ntasks, counts, nullptr);
        },
        [&](hipStream_t s, unsigned grid, const int32_t *qc, long long m, long long ntasks, int32_t *lists, int32_t *counts) {
            sph_density_kernel<KT, 2><<<dim3(grid), dim3(256), 0, s>>>(a, qc, m, wd, nint, lists, ntasks, counts, nullptr);
        },
        [&](unsigned grid, const int32_t *redo, int32_t *lists, const long long *d_nredo) {
            sph_density_kernel<KT, 3><<<dim3(grid), dim3(64), 0, st>>>(a, redo, 0, wd, nint, lists, 0, nullptr, d_nredo);
        },
        [&](unsigned grid, const int32_t *redo2, const long long *d_n2) {
            sph_density_block_kernel<KT><<<dim3(grid), dim3(HB_THREADS), 0, st>>>(a, redo2, wd, nint, d_n2);
        });
}
template <int KT>
static int launch_hydro(shq_context *ctx, const SphDev &a, const int32_t *q, long long nq, const HydroConst &hc, unsigned long long *nint)
{
    hipStream_t st = ctx->stream;
    return launch_two_kernel(
        ctx, q, nq, nq,
        [&](hipStream_t s, unsigned grid, const int32_t *qc, long long m, long long ntasks, int32_t *lists, int32_t *counts) {
            sph_hydro_kernel<KT, 1><<<dim3(grid), dim3(64 * SPH_WALK_WPB), 0, s>>>(a, qc, m, hc, nint, lists, ntasks, counts, nullptr);
        },
        [&](hipStream_t s, unsigned grid, const int32_t *qc, long long m, long long ntasks, int32_t *lists, int32_t *counts) {
            sph_hydro_kernel<KT, 2><<<dim3(grid), dim3(256), 0, s>>>(a, qc, m, hc, nint, lists, ntasks, counts, nullptr);
        },
        [&](unsigned grid, const int32_t *redo, int32_t *lists, const long long *d_nredo) {
            sph_hydro_kernel<KT, 3><<<dim3(grid), dim3(64), 0, st>>>(a, redo, 0, hc, nint, lists, 0, nullptr, d_nredo);
        },
        [&](unsigned grid, const int32_t *redo2, const long long *d_n2) {
            sph_hydro_block_kernel<KT><<<dim3(grid), dim3(HB_THREADS), 0, st>>>(a, redo2, hc, nint, d_n2);
        });
}

/* Device-resident density(): queue = d_queue[0..nq) of particle indices (already filtered by
 * DensityQuery::haswork).  Runs the whole Hsml loop. */
/* ---- density in phases: the set of hooks a TreeWalk backend overrides (treewalk2.cuh:212-394) --------------------
 * begin (DensityOutput ctor) -> { primary (ev_primary) -> [reduce of returned export results] -> post (ev_postprocess +
 * the do_hsml_loop bookkeeping, treewalk2.h:480-557) } until the redo queue is empty -> end. */
static SphDev density_dev(shq_context *ctx)
{
    SphDev a = make_dev(ctx);
    a.gradrho = ctx->sphrun.want_gradrho ? ctx->s_gradrho.ptr : nullptr;
    a.Box = ctx->sphrun.dp.BoxSize;
    a.invBox = 1.0 / ctx->sphrun.dp.BoxSize;
    return a;
}

int shq_sph_density_begin(shq_context *ctx, const shq_density_params *p, const int32_t *d_queue, int64_t nq, int want_gradrho)
{
    const long long n = ctx->numpart;
    const size_t cap = (size_t) (n > 0 ? n : 1);
    SHQ_TRY(ctx->s_numngb.reserve(cap));
    SHQ_TRY(ctx->s_dhsmldens.reserve(cap));
    SHQ_TRY(ctx->s_left.reserve(cap));
    SHQ_TRY(ctx->s_right.reserve(cap));
    SHQ_TRY(ctx->s_rot.reserve(3 * cap));
    SHQ_TRY(ctx->s_todo.reserve(cap));
    SHQ_TRY(ctx->s_queue2.reserve(cap));
    SHQ_TRY(ctx->s_queue3.reserve(cap));
    SHQ_TRY(ctx->s_blockcount.reserve(nblk(n) + 1));
    SHQ_TRY(ctx->s_counters.reserve(8));
    SHQ_TRY(reserve_nlist(ctx, nq));
    if(want_gradrho)
        SHQ_TRY(ctx->s_gradrho.reserve(3 * cap));
    SHQ_CHECK(p->DensityKernelType == 1 || p->DensityKernelType == 2 || p->DensityKernelType == 4, SHQ_ERR_INVALID,
              "unknown DensityKernelType %d", p->DensityKernelType);
    /* DensityOutput ctor, densitytree2.hpp:92-98 */
    if(n > 0) {
        SHQ_HIP(hipMemsetAsync(ctx->s_left.ptr, 0, sizeof(double) * n, ctx->stream));
        SHQ_HIP(hipMemsetAsync(ctx->s_numngb.ptr, 0, sizeof(double) * n, ctx->stream));
        fill_kernel<<<dim3(nblk(n)), dim3(256), 0, ctx->stream>>>(ctx->s_right.ptr, n, p->BoxSize);
    }
    SHQ_HIP(hipMemsetAsync(ctx->s_counters.ptr, 0, sizeof(long long) * 8, ctx->stream));
    shq_context::SphRun &r = ctx->sphrun;
    r.dp = *p;
    r.want_gradrho = want_gradrho;
    r.cur = d_queue;
    r.size = nq;
    r.nq0 = nq;
    r.wsel = 0;
    r.niter = 0;
    r.phase = 1;
    SHQ_HIP(hipEventRecord(ctx->ev_begin[14], ctx->stream));
    return SHQ_OK;
}

int shq_sph_density_primary(shq_context *ctx)
{
    shq_context::SphRun &r = ctx->sphrun;
    SHQ_CHECK(r.phase == 1, SHQ_ERR_STATE, "density primary: no density walk is open");
    if(r.size == 0)
        return SHQ_OK;
    const SphDev a = density_dev(ctx);
    unsigned long long *nint = reinterpret_cast<unsigned long long *>(ctx->s_counters.ptr + 1);
    switch(r.dp.DensityKernelType) {
    case 1: SHQ_TRY(launch_density<1>(ctx, a, r.cur, r.size, r.nq0, r.dp.WindsDecouple, nint)); break;
    case 2: SHQ_TRY(launch_density<2>(ctx, a, r.cur, r.size, r.nq0, r.dp.WindsDecouple, nint)); break;
    default: SHQ_TRY(launch_density<4>(ctx, a, r.cur, r.size, r.nq0, r.dp.WindsDecouple, nint)); break;
    }
    return SHQ_OK;
}

int shq_sph_density_post(shq_context *ctx, int64_t *nredo)
{
    shq_context::SphRun &r = ctx->sphrun;
    SHQ_CHECK(r.phase == 1, SHQ_ERR_STATE, "density postprocess: no density walk is open");
    const shq_density_params *p = &r.dp;
    *nredo = 0;
    if(r.size > 0) {
        const SphDev a = density_dev(ctx);
        PostArgs pa;
        pa.Box = p->BoxSize;
        pa.DesNumNgb = p->DesNumNgb;
        pa.DesNumNgbBH = p->DesNumNgbBH;
        pa.MinGasHsml = p->MinGasHsml;
        pa.MaxDev = p->MaxNumNgbDeviation;
        pa.update_hsml = p->update_hsml;
        pa.BlackHoleOn = p->BlackHoleOn;
        pa.DoEgyDensity = p->DoEgyDensity;
        pa.hmax_tried = reinterpret_cast<unsigned long long *>(ctx->s_counters.ptr + 3);
        sph_density_post_kernel<<<dim3(nblk(r.size)), dim3(256), 0, ctx->stream>>>(a, r.cur, r.size, pa, ctx->s_todo.ptr);
        SHQ_HIP(hipGetLastError());
    }
    r.niter++;
    if(!p->update_hsml || r.size == 0) {
        r.size = 0;
        return SHQ_OK;
    }
    int32_t *bufs[2] = {ctx->s_queue2.ptr, ctx->s_queue3.ptr};
    long long *total = ctx->s_counters.ptr;
    const int nb = (int) nblk(r.size);
    compact_count_kernel<<<dim3(nb), dim3(256), 0, ctx->stream>>>(ctx->s_todo.ptr, r.size, ctx->s_blockcount.ptr);
    compact_scan_kernel<<<dim3(1), dim3(1024), 0, ctx->stream>>>(ctx->s_blockcount.ptr, nb, total);
    compact_write_kernel<<<dim3(nb), dim3(256), 0, ctx->stream>>>(ctx->s_todo.ptr, r.size, ctx->s_blockcount.ptr, bufs[r.wsel]);
    long long newsize = 0;
    SHQ_HIP(hipMemcpyAsync(&newsize, total, sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    r.size = newsize;
    if(newsize == 0)
        return SHQ_OK;
    /* the neighbours' Hsml copy in leaf order is only read by hydro, so no refresh is needed here */
    r.cur = bufs[r.wsel];
    r.wsel ^= 1;
    if(r.niter > SPH_MAXITER) {
        shq_set_error("failed to converge density for %lld particles", newsize);
        return SHQ_ERR_NOCONV;
    }
    *nredo = newsize;
    return SHQ_OK;
}

int shq_sph_density_end(shq_context *ctx, shq_sph_stats *stats)
{
    shq_context::SphRun &r = ctx->sphrun;
    SHQ_CHECK(r.phase == 1, SHQ_ERR_STATE, "density end: no density walk is open");
    r.phase = 0;
    SHQ_HIP(hipEventRecord(ctx->ev_end[14], ctx->stream));
    if(stats) {
        unsigned long long h_c[3] = {0, 0, 0}; /* [0] interactions, [2] bits of the largest Hsml tried */
        SHQ_HIP(hipMemcpyAsync(h_c, ctx->s_counters.ptr + 1, sizeof(h_c), hipMemcpyDeviceToHost, ctx->stream));
        SHQ_HIP(hipStreamSynchronize(ctx->stream));
        float ms = 0;
        (void) hipEventElapsedTime(&ms, ctx->ev_begin[14], ctx->ev_end[14]);
        stats->ntargets = r.nq0;
        stats->ninteractions = (int64_t) h_c[0];
        stats->niterations = r.niter;
        stats->kernel_ms = ms;
        double hm;
        memcpy(&hm, &h_c[2], sizeof(hm));
        stats->hsml_max_tried = hm;
    }
    return SHQ_OK;
}

int shq_sph_density_device(shq_context *ctx, const shq_density_params *p, const int32_t *d_queue, int64_t nq, int want_gradrho,
                           shq_sph_stats *stats)
{
    SHQ_TRY(shq_sph_density_begin(ctx, p, d_queue, nq, want_gradrho));
    int64_t nredo = 0;
    do {
        SHQ_TRY(shq_sph_density_primary(ctx));
        SHQ_TRY(shq_sph_density_post(ctx, &nredo));
    } while(nredo > 0);
    return shq_sph_density_end(ctx, stats);
}

/* DensityResult::reduce<TREEWALK_GHOSTS> (densitytree2.hpp:308-343): one thread per table entry, the first entry of
 * a run of equal places adds the run in order (entries arrive grouped by target). */
__global__ void sph_density_reduce_kernel(const SphDev a, long long n, const int32_t *__restrict__ place, const shq_density_result *__restrict__ res)
{
    const long long k = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= n)
        return;
    const long long i = place[k];
    if(k > 0 && place[k - 1] == i)
        return;
    const int type = a.pflags[i] >> 4;
    for(long long j = k; j < n && place[j] == i; j++) {
        const shq_density_result r = res[j];
        a.numngb[i] += r.Ngb;
        a.dhsmldens[i] += r.DhsmlDensity;
        if(type == 0 || type == 5) {
            a.rho[i] += r.Rho;
            a.div[i] += r.Div;
        }
        if(type == 0) {
            a.rot[3 * i] += r.Rot[0];
            a.rot[3 * i + 1] += r.Rot[1];
            a.rot[3 * i + 2] += r.Rot[2];
            if(a.gradrho) {
                a.gradrho[3 * i] += r.GradRho[0];
                a.gradrho[3 * i + 1] += r.GradRho[1];
                a.gradrho[3 * i + 2] += r.GradRho[2];
            }
            a.egyrho[i] += r.EgyRho;
            a.dhsmlegy[i] += r.DhsmlEgyDensity;
        }
    }
}

int shq_sph_density_reduce(shq_context *ctx, const int32_t *d_place, const void *d_results, int64_t n)
{
    SHQ_CHECK(ctx->sphrun.phase == 1, SHQ_ERR_STATE, "density reduce: no density walk is open");
    if(n == 0)
        return SHQ_OK;
    const SphDev a = density_dev(ctx);
    sph_density_reduce_kernel<<<dim3(nblk(n)), dim3(256), 0, ctx->stream>>>(a, n, d_place, static_cast<const shq_density_result *>(d_results));
    SHQ_HIP(hipGetLastError());
    return SHQ_OK;
}

/* LocalNgbTreeWalk::visit<TREEWALK_GHOSTS> for imported DensityQuery records: the target-side arrays of the
 * walk kernel are swapped for query-indexed ones, the neighbour side stays the local tree.  d_q* are nq long;
 * d_out holds 12 nq doubles: Ngb, DhsmlDensity, Rho, Div, EgyRho, DhsmlEgyDensity, Rot[3], GradRho[3] (SoA). */
int shq_sph_density_secondary(shq_context *ctx, const shq_density_params *p, const double4 *d_qposm, const double *d_qhsml,
                              const double4 *d_qvelp, const uint8_t *d_qflags, const int4 *d_qseg, int64_t nq, double *d_out,
                              unsigned long long *d_nint)
{
    if(nq == 0)
        return SHQ_OK;
    SHQ_CHECK(p->DensityKernelType == 1 || p->DensityKernelType == 2 || p->DensityKernelType == 4, SHQ_ERR_INVALID,
              "unknown DensityKernelType %d", p->DensityKernelType);
    SHQ_TRY(ctx->s_nlist2.reserve((size_t) NL_REDO_BLOCKS * 4 * NL_ROWS * 64));
    SphDev a = make_dev(ctx);
    a.Box = p->BoxSize;
    a.invBox = 1.0 / p->BoxSize;
    a.posm = d_qposm;
    a.hsml = const_cast<double *>(d_qhsml);
    a.velp = d_qvelp;
    a.pflags = d_qflags;
    a.numngb = d_out;
    a.dhsmldens = d_out + nq;
    a.rho = d_out + 2 * nq;
    a.div = d_out + 3 * nq;
    a.egyrho = d_out + 4 * nq;
    a.dhsmlegy = d_out + 5 * nq;
    a.rot = d_out + 6 * nq;
    a.gradrho = d_out + 9 * nq;
    const long long ntasks = (nq + 255) / 256;
    const unsigned grid = (unsigned) (ntasks < NL_REDO_BLOCKS ? ntasks : NL_REDO_BLOCKS);
    hipStream_t st = ctx->stream;
    switch(p->DensityKernelType) {
    case 1: sph_density_kernel<1, 0, true><<<dim3(grid), dim3(256), 0, st>>>(a, nullptr, nq, p->WindsDecouple, d_nint, ctx->s_nlist2.ptr, ntasks, nullptr, nullptr, d_qseg); break;
    case 2: sph_density_kernel<2, 0, true><<<dim3(grid), dim3(256), 0, st>>>(a, nullptr, nq, p->WindsDecouple, d_nint, ctx->s_nlist2.ptr, ntasks, nullptr, nullptr, d_qseg); break;
    default: sph_density_kernel<4, 0, true><<<dim3(grid), dim3(256), 0, st>>>(a, nullptr, nq, p->WindsDecouple, d_nint, ctx->s_nlist2.ptr, ntasks, nullptr, nullptr, d_qseg); break;
    }
    SHQ_HIP(hipGetLastError());
    return SHQ_OK;
}

/* ---- hydro in phases: begin -> primary (ev_primary) -> [reduce of returned export results] -> post (ev_postprocess) -> end */
static HydroConst hydro_const(const shq_hydro_params *p)
{
    HydroConst hc;
    hc.hubble_a2 = p->hubble_a2;
    hc.fac_mu = p->fac_mu;
    hc.fac_vsic_fix = p->fac_vsic_fix;
    hc.ArtBulkViscConst = p->ArtBulkViscConst;
    hc.contrast = p->DensityContrastLimit;
    hc.DISPH = p->DensityIndependentSphOn;
    return hc;
}

static SphDev hydro_dev(shq_context *ctx, const shq_hydro_params *p)
{
    SphDev a = make_dev(ctx);
    a.Box = p->BoxSize;
    a.invBox = 1.0 / p->BoxSize;
    return a;
}

int shq_sph_hydro_begin(shq_context *ctx, const shq_hydro_params *p, const int32_t *d_queue, int64_t nq)
{
    const long long n = ctx->numpart;
    const size_t cap = (size_t) (n > 0 ? n : 1);
    SHQ_TRY(ctx->g_hydroaccel_out.reserve(3 * cap));
    SHQ_TRY(ctx->g_dtentropy_out.reserve(cap));
    SHQ_TRY(ctx->g_maxsignalvel.reserve(cap));
    SHQ_TRY(ctx->s_counters.reserve(8));
    SHQ_TRY(reserve_nlist(ctx, nq));
    SHQ_CHECK(p->DensityKernelType == 1 || p->DensityKernelType == 2 || p->DensityKernelType == 4, SHQ_ERR_INVALID,
              "unknown DensityKernelType %d", p->DensityKernelType);
    SHQ_HIP(hipMemsetAsync(ctx->s_counters.ptr, 0, sizeof(long long) * 8, ctx->stream));
    shq_context::SphRun &r = ctx->sphrun;
    r.hp = *p;
    r.cur = d_queue;
    r.size = nq;
    r.nq0 = nq;
    r.niter = 0;
    r.phase = 2;
    SHQ_HIP(hipEventRecord(ctx->ev_begin[14], ctx->stream));
    return SHQ_OK;
}

int shq_sph_hydro_primary(shq_context *ctx)
{
    shq_context::SphRun &r = ctx->sphrun;
    SHQ_CHECK(r.phase == 2, SHQ_ERR_STATE, "hydro primary: no hydro walk is open");
    if(r.size == 0)
        return SHQ_OK;
    const SphDev a = hydro_dev(ctx, &r.hp);
    const HydroConst hc = hydro_const(&r.hp);
    unsigned long long *nint = reinterpret_cast<unsigned long long *>(ctx->s_counters.ptr + 1);
    switch(r.hp.DensityKernelType) {
    case 1: SHQ_TRY(launch_hydro<1>(ctx, a, r.cur, r.size, hc, nint)); break;
    case 2: SHQ_TRY(launch_hydro<2>(ctx, a, r.cur, r.size, hc, nint)); break;
    default: SHQ_TRY(launch_hydro<4>(ctx, a, r.cur, r.size, hc, nint)); break;
    }
    return SHQ_OK;
}

int shq_sph_hydro_post(shq_context *ctx)
{
    shq_context::SphRun &r = ctx->sphrun;
    SHQ_CHECK(r.phase == 2, SHQ_ERR_STATE, "hydro postprocess: no hydro walk is open");
    if(r.size > 0) {
        const SphDev a = hydro_dev(ctx, &r.hp);
        const shq_hydro_params *p = &r.hp;
        sph_hydro_post_kernel<<<dim3(nblk(r.size)), dim3(256), 0, ctx->stream>>>(a, r.cur, r.size, ctx->g_density.ptr, ctx->g_delaytime.ptr,
                                                                                p->hubble_a2, p->atime, p->WindSpeed, p->WindFreeTravelDensThresh);
        SHQ_HIP(hipGetLastError());
    }
    r.niter = 1;
    return SHQ_OK;
}

int shq_sph_hydro_end(shq_context *ctx, shq_sph_stats *stats)
{
    shq_context::SphRun &r = ctx->sphrun;
    SHQ_CHECK(r.phase == 2, SHQ_ERR_STATE, "hydro end: no hydro walk is open");
    r.phase = 0;
    SHQ_HIP(hipEventRecord(ctx->ev_end[14], ctx->stream));
    if(stats) {
        unsigned long long *nint = reinterpret_cast<unsigned long long *>(ctx->s_counters.ptr + 1);
        unsigned long long h_nint = 0;
        SHQ_HIP(hipMemcpyAsync(&h_nint, nint, sizeof(h_nint), hipMemcpyDeviceToHost, ctx->stream));
        SHQ_HIP(hipStreamSynchronize(ctx->stream));
        float ms = 0;
        (void) hipEventElapsedTime(&ms, ctx->ev_begin[14], ctx->ev_end[14]);
        stats->ntargets = r.nq0;
        stats->ninteractions = (int64_t) h_nint;
        stats->niterations = 1;
        stats->kernel_ms = ms;
        stats->hsml_max_tried = 0;
        if(getenv("SHQ_SPH_DEBUG") && r.nq0 > 0) {
            unsigned long long d[5];
            SHQ_HIP(hipMemcpy(d, nint, sizeof(d), hipMemcpyDeviceToHost));
            const double nw = (double) ((r.nq0 + 63) / 64);
            fprintf(stderr, "[shq] hydro per wave: nodes %.1f candidates %.1f pair rounds %.1f; per target: candidates %.1f\n",
                    d[1] / nw, d[2] / nw, d[4] / nw, (double) d[0] / r.nq0);
        }
    }
    return SHQ_OK;
}

int shq_sph_hydro_device(shq_context *ctx, const shq_hydro_params *p, const int32_t *d_queue, int64_t nq, shq_sph_stats *stats)
{
    SHQ_TRY(shq_sph_hydro_begin(ctx, p, d_queue, nq));
    SHQ_TRY(shq_sph_hydro_primary(ctx));
    SHQ_TRY(shq_sph_hydro_post(ctx));
    return shq_sph_hydro_end(ctx, stats);
}

/* HydroResult::reduce<TREEWALK_GHOSTS>, hydratree2.hpp:213-227 */
__global__ void sph_hydro_reduce_kernel(const SphDev a, long long n, const int32_t *__restrict__ place, const shq_hydro_result *__restrict__ res)
{
    const long long k = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= n)
        return;
    const long long i = place[k];
    if(k > 0 && place[k - 1] == i)
        return;
    double a0 = a.hacc[3 * i], a1 = a.hacc[3 * i + 1], a2 = a.hacc[3 * i + 2], de = a.dtent[i], ms = a.maxsig[i];
    for(long long j = k; j < n && place[j] == i; j++) {
        const shq_hydro_result r = res[j];
        a0 += r.Acc[0];
        a1 += r.Acc[1];
        a2 += r.Acc[2];
        de += r.DtEntropy;
        if(ms < r.MaxSignalVel)
            ms = r.MaxSignalVel;
    }
    a.hacc[3 * i] = a0;
    a.hacc[3 * i + 1] = a1;
    a.hacc[3 * i + 2] = a2;
    a.dtent[i] = de;
    a.maxsig[i] = ms;
}

int shq_sph_hydro_reduce(shq_context *ctx, const int32_t *d_place, const void *d_results, int64_t n)
{
    SHQ_CHECK(ctx->sphrun.phase == 2, SHQ_ERR_STATE, "hydro reduce: no hydro walk is open");
    if(n == 0)
        return SHQ_OK;
    const SphDev a = hydro_dev(ctx, &ctx->sphrun.hp);
    sph_hydro_reduce_kernel<<<dim3(nblk(n)), dim3(256), 0, ctx->stream>>>(a, n, d_place, static_cast<const shq_hydro_result *>(d_results));
    SHQ_HIP(hipGetLastError());
    return SHQ_OK;
}

/* visit<TREEWALK_GHOSTS> for imported HydroQuery records; d_out: 5 nq doubles, Acc[3] (AoS, 3 nq), DtEntropy, MaxSignalVel */
int shq_sph_hydro_secondary(shq_context *ctx, const shq_hydro_params *p, const double4 *d_qposm, const double *d_qhsml,
                            const double4 *d_qvelp, const double4 *d_qC, const double4 *d_qD, const int4 *d_qseg, int64_t nq, double *d_out,
                            unsigned long long *d_nint)
{
    if(nq == 0)
        return SHQ_OK;
    SHQ_CHECK(p->DensityKernelType == 1 || p->DensityKernelType == 2 || p->DensityKernelType == 4, SHQ_ERR_INVALID,
              "unknown DensityKernelType %d", p->DensityKernelType);
    SHQ_TRY(ctx->s_nlist2.reserve((size_t) NL_REDO_BLOCKS * 4 * NL_ROWS * 64));
    SphDev a = hydro_dev(ctx, p);
    a.posm = d_qposm;
    a.hsml = const_cast<double *>(d_qhsml);
    a.velp = d_qvelp;
    a.hydC = d_qC;
    a.hydD = d_qD;
    a.hacc = d_out;
    a.dtent = d_out + 3 * nq;
    a.maxsig = d_out + 4 * nq;
    const HydroConst hc = hydro_const(p);
    const long long ntasks = (nq + 255) / 256;
    const unsigned grid = (unsigned) (ntasks < NL_REDO_BLOCKS ? ntasks : NL_REDO_BLOCKS);
    hipStream_t st = ctx->stream;
    switch(p->DensityKernelType) {
    case 1: sph_hydro_kernel<1, 0, true><<<dim3(grid), dim3(256), 0, st>>>(a, nullptr, nq, hc, d_nint, ctx->s_nlist2.ptr, ntasks, nullptr, nullptr, d_qseg); break;
    case 2: sph_hydro_kernel<2, 0, true><<<dim3(grid), dim3(256), 0, st>>>(a, nullptr, nq, hc, d_nint, ctx->s_nlist2.ptr, ntasks, nullptr, nullptr, d_qseg); break;
    default: sph_hydro_kernel<4, 0, true><<<dim3(grid), dim3(256), 0, st>>>(a, nullptr, nq, hc, d_nint, ctx->s_nlist2.ptr, ntasks, nullptr, nullptr, d_qseg); break;
    }
    SHQ_HIP(hipGetLastError());
    return SHQ_OK;
}

/* The export queries of a walk in progress, one record per table entry (DensityQuery / HydroQuery ctors,
 * densitytree2.hpp:267-278, hydratree2.hpp:165-191): everything comes from the resident arrays. */
__global__ void sph_fill_density_queries_kernel(const SphDev a, long long n, const shq_data_index *__restrict__ table, shq_density_query *out)
{
    const long long k = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= n)
        return;
    const shq_data_index e = table[k];
    const long long i = e.Index;
    shq_density_query q;
    const double4 p = a.posm[i], v = a.velp[i];
    q.Pos[0] = p.x; q.Pos[1] = p.y; q.Pos[2] = p.z;
    for(int j = 0; j < 4; j++)
        q.NodeList[j] = e.NodeList[j];
    q.Vel[0] = v.x; q.Vel[1] = v.y; q.Vel[2] = v.z;
    q.Hsml = a.hsml[i];
    q.Type = a.pflags[i] >> 4;
    q.pad_ = 0;
    out[k] = q;
}

__global__ void sph_fill_hydro_queries_kernel(const SphDev a, long long n, const shq_data_index *__restrict__ table, const double *__restrict__ density,
                                              const uint8_t *__restrict__ bin_hydro, int DISPH, double fac_mu, shq_hydro_query *out)
{
    const long long k = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= n)
        return;
    const shq_data_index e = table[k];
    const long long i = e.Index;
    shq_hydro_query q;
    const double4 p = a.posm[i], v = a.velp[i];
    q.Pos[0] = p.x; q.Pos[1] = p.y; q.Pos[2] = p.z;
    for(int j = 0; j < 4; j++)
        q.NodeList[j] = e.NodeList[j];
    const double eom = DISPH ? a.egyrho[i] : density[i]; /* SPH_EOMDensity, hydratree2.hpp:36-45 */
    q.EgyRho = eom;
    q.EntVarPred = v.w;
    q.Vel[0] = v.x; q.Vel[1] = v.y; q.Vel[2] = v.z;
    q.Hsml = a.hsml[i];
    q.Mass = p.w;
    q.Density = density[i];
    q.Pressure = pressure_predict(eom, v.w);
    q.SPH_DhsmlDensityFactor = a.dhsmlegy[i];
    const double cs = sqrt(SPH_GAMMA * q.Pressure / eom);
    q.F1 = fabs(a.div[i]) / (fabs(a.div[i]) + a.curl[i] + 0.0001 * cs / q.Hsml / fac_mu);
    q.TimeBinHydro = bin_hydro[i];
    q.pad_ = 0;
    out[k] = q;
}

int shq_sph_fill_queries_device(shq_context *ctx, const shq_data_index *d_table, int64_t n, void *d_out)
{
    const shq_context::SphRun &r = ctx->sphrun;
    SHQ_CHECK(r.phase == 1 || r.phase == 2, SHQ_ERR_STATE, "fill_queries: no SPH walk is open");
    if(n == 0)
        return SHQ_OK;
    const SphDev a = make_dev(ctx);
    if(r.phase == 1)
        sph_fill_density_queries_kernel<<<dim3(nblk(n)), dim3(256), 0, ctx->stream>>>(a, n, d_table, static_cast<shq_density_query *>(d_out));
    else
        sph_fill_hydro_queries_kernel<<<dim3(nblk(n)), dim3(256), 0, ctx->stream>>>(a, n, d_table, ctx->g_density.ptr, ctx->bin_hydro.ptr,
                                                                                   r.hp.DensityIndependentSphOn, r.hp.fac_mu,
                                                                                   static_cast<shq_hydro_query *>(d_out));
    SHQ_HIP(hipGetLastError());
    return SHQ_OK;
}

/* ---- stellar density: stellar_density2.cpp ------------------------------------------------------
 * The SPH volume weights around star particles for the metal return: a density-like walk over the gas tree that evaluates
 * NHSML = 10 trial radii per star in one pass (stellareffhsml :38-54, ngbiter :219-254), then narrows the Hsml bounds
 * (postprocess :113-154, ngb_narrow_down treewalk.c:1349-1406) until every star has DesNumNgb +- MaxNgbDeviation neighbours.
 * Same walk machinery as the density (fused variant: stars are few); the walk uses the largest trial radius throughout — the
 * reference shrinks its search radius on the way, which only skips candidates its ngbiter would reject anyway — and each
 * lane works through its neighbours in depth-first order with the reference's per-neighbour logic, including the running
 * `maxcmpte` cut. */
#define ST_NHSML 10

__device__ __forceinline__ double st_effhsml(int i, double left, double right, double Hsml, double Box)
{
    if(right > 0.99 * Box)
        right = Hsml * ((1. + ST_NHSML) / ST_NHSML);
    if(left == 0)
        left = 0.1 * Hsml;
    const double rvol = pow(right, 3), lvol = pow(left, 3);
    return pow((1. * i + 1) / (1. * ST_NHSML + 1) * (rvol - lvol) + lvol, 1. / 3);
}

struct StellarArgs {
    double Box, DesNumNgb, MaxDev;
    int SPHWeighting;
    const double *rho_leaf;   /* gas density by leaf slot */
    double *starvol;          /* by particle index */
    int32_t *todo;
};

template <int KT>
__global__ __launch_bounds__(256) void sph_stellar_kernel(const SphDev a, const int32_t *queue, long long nq, const StellarArgs sa,
                                                          unsigned long long *nint_total, int32_t *__restrict__ nlist, long long ntasks)
{
    __shared__ __attribute__((aligned(32))) char lds[4 * NW_LDS_PER_WAVE(false)];
    const int lane = threadIdx.x & 63;
    for(long long task = xcd_block(blockIdx.x, gridDim.x); task < ntasks; task += gridDim.x) {
    const long long wave = task * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int32_t *myl = nlist + ((size_t) blockIdx.x * 4 + (threadIdx.x >> 6)) * (size_t) (NL_ROWS * 64) + lane;
    const long long t = wave * 64 + lane;
    const bool valid = t < nq;
    long long pi = 0;
    double px = 0, py = 0, pz = 0, h = 1, L = 0, R = sa.Box;
    if(valid) {
        pi = queue[t];
        const double4 p = a.posm[pi];
        px = p.x; py = p.y; pz = p.z;
        h = a.hsml[pi];
        L = a.left[pi];
        R = a.right[pi];
    }
    double he[ST_NHSML], he2[ST_NHSML], hinv[ST_NHSML], wnorm[ST_NHSML], Ngb[ST_NHSML], Vol[ST_NHSML];
#pragma unroll
    for(int k = 0; k < ST_NHSML; k++) {
        he[k] = st_effhsml(k, L, R, h, sa.Box);
        he2[k] = he[k] * he[k];
        const Kern<KT> kr(he[k]);
        hinv[k] = 1.0 / he[k];
        wnorm[k] = kr.Wknorm;
        Ngb[k] = 0;
        Vol[k] = 0;
    }
    int maxcmpte = ST_NHSML;
    const Kern<KT> k0(1.0);

    /* ngbiter, stellar_density2.cpp:219-254 */
    auto pair = [&](const int s) {
        const double4 q = a.posm_leaf[s];
        const double d0 = wrapd(px - q.x, a.Box, a.invBox);
        const double d1 = wrapd(py - q.y, a.Box, a.invBox);
        const double d2 = wrapd(pz - q.z, a.Box, a.invBox);
        const double r2 = d0 * d0 + d1 * d1 + d2 * d2;
        double lim = he2[0];
#pragma unroll
        for(int k = 1; k < ST_NHSML; k++)
            lim = (k == maxcmpte - 1) ? he2[k] : lim;
        if(maxcmpte == 1)
            lim = he2[0];
        if(!(r2 < lim))
            return;
        const double r = sqrt(r2);
        const double vj = q.w / sa.rho_leaf[s];
#pragma unroll
        for(int k = 0; k < ST_NHSML; k++) {
            if(k < maxcmpte && r2 < he2[k]) {
                const double wk = wnorm[k] * k0.wk_int(r * hinv[k] * (Kern<KT>::support / 2.));
                Ngb[k] += wk * ((4.0 / 3 * M_PI) * (he[k] * he[k] * he[k]));
                Vol[k] += sa.SPHWeighting ? vj * wk : vj;
            }
        }
        int first = ST_NHSML;
#pragma unroll
        for(int k = ST_NHSML - 1; k >= 0; k--)
            first = (Ngb[k] > sa.DesNumNgb) ? k : first;
        if(first < ST_NHSML)
            maxcmpte = first + 1;
    };
    const double hwalk2 = he2[ST_NHSML - 1];
    auto accept = [&](const double r2, const double, const int) { return r2 < hwalk2; };
    int fill = 0;
    bool ovf = false;
    unsigned int nint = ngb_walk<false, false, false>(a, lds + (threadIdx.x >> 6) * NW_LDS_PER_WAVE(false), myl, valid, px, py, pz, he[ST_NHSML - 1],
                                                      accept, pair, (unsigned int *) nullptr, fill, ovf);
    if(valid) {
        /* StellarDensityOutput::postprocess (stellar_density2.cpp:113-154) with ngb_narrow_down (treewalk.c:1349-1406) */
        const int desi = (int) sa.DesNumNgb;
        int close = 0;
        double ngbdist = fabs(Ngb[0] - desi);
#pragma unroll
        for(int k = 1; k < ST_NHSML; k++) {
            const double nd = fabs(Ngb[k] - desi);
            if(k < maxcmpte && nd < ngbdist) {
                ngbdist = nd;
                close = k;
            }
        }
        bool stop = false;
#pragma unroll
        for(int k = 0; k < ST_NHSML; k++) {
            if(k < maxcmpte && !stop) {
                if(Ngb[k] < desi)
                    L = he[k];
                if(Ngb[k] > desi) {
                    R = he[k];
                    stop = true;
                }
            }
        }
        double hc = he[0], nc = Ngb[0], vc = Vol[0], rl = he[0], rl1 = he[0], nl = Ngb[0], nl1 = Ngb[0];
#pragma unroll
        for(int k = 1; k < ST_NHSML; k++) {
            if(k == close) { hc = he[k]; nc = Ngb[k]; vc = Vol[k]; }
            if(k == maxcmpte - 1) { rl = he[k]; nl = Ngb[k]; rl1 = he[k - 1]; nl1 = Ngb[k - 1]; }
        }
        double hs = hc;
        if(R > 0.99 * sa.Box) {
            double dngbdv = 0;
            if(maxcmpte > 1 && rl > rl1)
                dngbdv = (nl - nl1) / (pow(rl, 3) - pow(rl1, 3));
            double newh = 4 * hs;
            if(dngbdv > 0) {
                const double dngb = desi - nl;
                const double nv = pow(hs, 3) + dngb / dngbdv;
                if(pow(nv, 1. / 3) < newh)
                    newh = pow(nv, 1. / 3);
            }
            hs = newh;
        }
        if(hs > R)
            hs = R;
        if(L == 0) {
            double dngbdv = 0;
            if(he[1] > he[0])
                dngbdv = (Ngb[1] - Ngb[0]) / (pow(he[1], 3) - pow(he[0], 3));
            if(maxcmpte == 1 && he[0] > 0)
                dngbdv = Ngb[0] / pow(he[0], 3);
            if(dngbdv > 0) {
                const double dngb = desi - Ngb[0];
                const double nv = pow(hs, 3) + dngb / dngbdv;
                hs = pow(nv, 1. / 3);
            }
        }
        if(hs < L)
            hs = L;
        a.hsml[pi] = hs;
        a.left[pi] = L;
        a.right[pi] = R;
        a.numngb[pi] = nc;
        sa.starvol[pi] = vc;
        int redo = 0;
        if(nc < (sa.DesNumNgb - sa.MaxDev) || nc > (sa.DesNumNgb + sa.MaxDev))
            redo = ((R - L) < 1.0e-4 * L) ? 0 : 1;
        sa.todo[t] = redo ? (int32_t) pi : -1;
    }
    unsigned int sn = nint;
    for(int off = 32; off > 0; off >>= 1)
        sn += __shfl_xor(sn, off);
    if(lane == 0 && nint_total)
        atomicAdd(nint_total, (unsigned long long) sn);
    } /* task loop */
}

__global__ void gather_rho_leaf_kernel(long long nleaf, const int32_t *__restrict__ pidx, const double *__restrict__ density, double *out)
{
    const long long s = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(s < nleaf)
        out[s] = density[pidx[s]];
}

int shq_sph_stellar_density_device(shq_context *ctx, const shq_stellar_params *p, const int32_t *d_queue, int64_t nq, double *d_starvol,
                                   shq_sph_stats *stats)
{
    const long long n = ctx->numpart;
    const size_t cap = (size_t) (n > 0 ? n : 1);
    SHQ_CHECK(p->DensityKernelType == 1 || p->DensityKernelType == 2 || p->DensityKernelType == 4, SHQ_ERR_INVALID,
              "unknown DensityKernelType %d", p->DensityKernelType);
    SHQ_TRY(ctx->s_numngb.reserve(cap));
    SHQ_TRY(ctx->s_left.reserve(cap));
    SHQ_TRY(ctx->s_right.reserve(cap));
    SHQ_TRY(ctx->s_todo.reserve(cap));
    SHQ_TRY(ctx->s_queue2.reserve(cap));
    SHQ_TRY(ctx->s_queue3.reserve(cap));
    SHQ_TRY(ctx->s_blockcount.reserve(nblk(n) + 1));
    SHQ_TRY(ctx->s_counters.reserve(8));
    SHQ_TRY(ctx->s_nlist2.reserve((size_t) NL_REDO_BLOCKS * 4 * NL_ROWS * 64));
    const long long nl = ctx->ntreeparts + SHQ_NMAXCHILD;
    SHQ_TRY(ctx->hsml_leaf.reserve(nl));   /* reused as the density-by-slot array */
    SHQ_TRY(ctx->flag_leaf.reserve(nl));
    SHQ_TRY(ctx->velp_leaf.reserve(nl));
    hipStream_t st = ctx->stream;
    if(n > 0) {
        SHQ_HIP(hipMemsetAsync(ctx->s_left.ptr, 0, sizeof(double) * n, st));
        fill_kernel<<<dim3(nblk(n)), dim3(256), 0, st>>>(ctx->s_right.ptr, n, p->BoxSize);
    }
    SHQ_HIP(hipMemsetAsync(ctx->s_counters.ptr, 0, sizeof(long long) * 8, st));
    /* neighbour-side arrays in leaf order: density and the skip flags (garbage / no longer gas) */
    gather_rho_leaf_kernel<<<dim3(nblk(nl)), dim3(256), 0, st>>>(nl, ctx->leaf_pidx.ptr, ctx->g_density.ptr, ctx->hsml_leaf.ptr);
    sph_gather_leaf_kernel<<<dim3(nblk(nl)), dim3(256), 0, st>>>(nl, ctx->leaf_pidx.ptr, ctx->velp.ptr, nullptr, nullptr, ctx->hsml.ptr, ctx->pflags.ptr,
                                                                nullptr, ctx->velp_leaf.ptr, nullptr, ctx->s_evp_in.ptr, ctx->flag_leaf.ptr);
    SHQ_HIP(hipGetLastError());
    SphDev a = make_dev(ctx);
    a.Box = p->BoxSize;
    a.invBox = 1.0 / p->BoxSize;
    StellarArgs sa;
    sa.Box = p->BoxSize;
    sa.DesNumNgb = p->DesNumNgb;
    sa.MaxDev = p->MaxNgbDeviation;
    sa.SPHWeighting = p->SPHWeighting;
    sa.rho_leaf = ctx->hsml_leaf.ptr;
    sa.starvol = d_starvol;
    sa.todo = ctx->s_todo.ptr;
    unsigned long long *nint = reinterpret_cast<unsigned long long *>(ctx->s_counters.ptr + 1);
    long long *total = ctx->s_counters.ptr;
    int32_t *bufs[2] = {ctx->s_queue2.ptr, ctx->s_queue3.ptr};
    int wsel = 0, niter = 0;
    const int32_t *cur = d_queue;
    long long size = nq;
    SHQ_HIP(hipEventRecord(ctx->ev_begin[14], st));
    while(size > 0) {
        const long long ntasks = (size + 255) / 256;
        const unsigned grid = (unsigned) (ntasks < NL_REDO_BLOCKS ? ntasks : NL_REDO_BLOCKS);
        switch(p->DensityKernelType) {
        case 1: sph_stellar_kernel<1><<<dim3(grid), dim3(256), 0, st>>>(a, cur, size, sa, nint, ctx->s_nlist2.ptr, ntasks); break;
        case 2: sph_stellar_kernel<2><<<dim3(grid), dim3(256), 0, st>>>(a, cur, size, sa, nint, ctx->s_nlist2.ptr, ntasks); break;
        default: sph_stellar_kernel<4><<<dim3(grid), dim3(256), 0, st>>>(a, cur, size, sa, nint, ctx->s_nlist2.ptr, ntasks); break;
        }
        SHQ_HIP(hipGetLastError());
        niter++;
        const int nb = (int) nblk(size);
        compact_count_kernel<<<dim3(nb), dim3(256), 0, st>>>(ctx->s_todo.ptr, size, ctx->s_blockcount.ptr);
        compact_scan_kernel<<<dim3(1), dim3(1024), 0, st>>>(ctx->s_blockcount.ptr, nb, total);
        compact_write_kernel<<<dim3(nb), dim3(256), 0, st>>>(ctx->s_todo.ptr, size, ctx->s_blockcount.ptr, bufs[wsel]);
        long long newsize = 0;
        SHQ_HIP(hipMemcpyAsync(&newsize, total, sizeof(long long), hipMemcpyDeviceToHost, st));
        SHQ_HIP(hipStreamSynchronize(st));
        size = newsize;
        cur = bufs[wsel];
        wsel ^= 1;
        if(size > 0 && niter > SPH_MAXITER) {
            shq_set_error("failed to converge the stellar density for %lld stars", size);
            return SHQ_ERR_NOCONV;
        }
    }
    SHQ_HIP(hipEventRecord(ctx->ev_end[14], st));
    if(stats) {
        unsigned long long h_nint = 0;
        SHQ_HIP(hipMemcpyAsync(&h_nint, nint, sizeof(h_nint), hipMemcpyDeviceToHost, st));
        SHQ_HIP(hipStreamSynchronize(st));
        float ms = 0;
        (void) hipEventElapsedTime(&ms, ctx->ev_begin[14], ctx->ev_end[14]);
        stats->ntargets = nq;
        stats->ninteractions = (int64_t) h_nint;
        stats->niterations = niter;
        stats->kernel_ms = ms;
        stats->hsml_max_tried = 0;
    }
    return SHQ_OK;
}

/* ---- black-hole velocity dispersion: veldisp2.cpp:20-199 ----------------------------------------
 * BHVelDispLocalTreeWalk::ngbiter (:126-144): over the dark matter inside a black hole's Hsml, the count and the first and
 * second moments of the predicted DM velocity (KickFactorData::DM_VelPred, density2.h:104-111) relative to the hole's;
 * BHVelDispOutput::postprocess (:49-63) turns them into VDisp.  Same fused walk as the other neighbour operators; the tree is
 * the caller's dark-matter tree. */
struct BhVdArgs {
    const double4 *vel_leaf;  /* predicted DM velocity by leaf slot */
    const double *vel;        /* [N][3] raw velocities (the hole's own) */
    double *out;              /* [nq][5]: NumDM, V1sumDM[3], V2sumDM, by queue position */
};

__global__ __launch_bounds__(256) void bh_veldisp_kernel(const SphDev a, const int32_t *queue, long long nq, const BhVdArgs ba,
                                                         int32_t *__restrict__ nlist, long long ntasks)
{
    __shared__ __attribute__((aligned(32))) char lds[4 * NW_LDS_PER_WAVE(false)];
    const int lane = threadIdx.x & 63;
    for(long long task = xcd_block(blockIdx.x, gridDim.x); task < ntasks; task += gridDim.x) {
    const long long wave = task * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int32_t *myl = nlist + ((size_t) blockIdx.x * 4 + (threadIdx.x >> 6)) * (size_t) (NL_ROWS * 64) + lane;
    const long long t = wave * 64 + lane;
    const bool valid = t < nq;
    double px = 0, py = 0, pz = 0, h = 1, vx = 0, vy = 0, vz = 0;
    if(valid) {
        const long long pi = queue[t];
        const double4 p = a.posm[pi];
        px = p.x; py = p.y; pz = p.z;
        h = a.hsml[pi];
        vx = ba.vel[3 * pi]; vy = ba.vel[3 * pi + 1]; vz = ba.vel[3 * pi + 2];
    }
    const double h2 = h * h;
    double num = 0, s0 = 0, s1 = 0, s2 = 0, v2 = 0;
    auto pair = [&](const int s) {
        const double4 w = ba.vel_leaf[s];
        num += 1;
        const double e0 = w.x - vx, e1 = w.y - vy, e2 = w.z - vz;
        s0 += e0; v2 += e0 * e0;
        s1 += e1; v2 += e1 * e1;
        s2 += e2; v2 += e2 * e2;
    };
    auto accept = [&](const double r2, const double, const int) { return r2 > 0 && r2 < h2; };
    int fill = 0;
    bool ovf = false;
    (void) ngb_walk<false, false, false>(a, lds + (threadIdx.x >> 6) * NW_LDS_PER_WAVE(false), myl, valid, px, py, pz, h, accept, pair,
                                         (unsigned int *) nullptr, fill, ovf);
    if(valid) {
        ba.out[5 * t] = num;
        ba.out[5 * t + 1] = s0;
        ba.out[5 * t + 2] = s1;
        ba.out[5 * t + 3] = s2;
        ba.out[5 * t + 4] = v2;
    }
    } /* task loop */
}

/* neighbour-side arrays of the DM tree in leaf order: DM_VelPred and the skip flag (garbage / not dark matter any more) */
__global__ void bh_veldisp_gather_kernel(long long nleaf, const int32_t *__restrict__ pidx, const double *__restrict__ vel,
                                         const double *__restrict__ treeacc, const double *__restrict__ gravpm, const uint8_t *__restrict__ bin_grav,
                                         const uint8_t *__restrict__ pflags, shq_kick_factors kf, int typemask, double4 *vel_leaf, int32_t *flag_leaf)
{
#pragma clang fp contract(off)
    const long long s = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(s >= nleaf)
        return;
    const long long p = pidx[s];
    double v[3];
    for(int j = 0; j < 3; j++)
        v[j] = vel[3 * p + j] + kf.gravkicks[bin_grav[p]] * treeacc[3 * p + j] + gravpm[3 * p + j] * kf.FgravkickB;
    vel_leaf[s] = make_double4(v[0], v[1], v[2], 0.0);
    const unsigned f = pflags[p];
    flag_leaf[s] = ((f & 1u) || !((1 << (f >> 4)) & typemask)) ? 1 : 0;
}

int shq_bh_veldisp_device(shq_context *ctx, const shq_kick_factors *kf, double BoxSize, const int32_t *d_queue, int64_t nq, double *d_out)
{
    if(nq == 0)
        return SHQ_OK;
    const long long nl = ctx->ntreeparts + SHQ_NMAXCHILD;
    SHQ_TRY(ctx->velp_leaf.reserve(nl));
    SHQ_TRY(ctx->flag_leaf.reserve(nl));
    SHQ_TRY(ctx->s_nlist2.reserve((size_t) NL_REDO_BLOCKS * 4 * NL_ROWS * 64));
    hipStream_t st = ctx->stream;
    bh_veldisp_gather_kernel<<<dim3(nblk(nl)), dim3(256), 0, st>>>(nl, ctx->leaf_pidx.ptr, ctx->vel.ptr, ctx->treeacc.ptr, ctx->gravpm.ptr,
                                                                  ctx->bin_grav.ptr, ctx->pflags.ptr, *kf, 1 << 1, ctx->velp_leaf.ptr,
                                                                  ctx->flag_leaf.ptr);
    SHQ_HIP(hipGetLastError());
    SphDev a = make_dev(ctx);
    a.Box = BoxSize;
    a.invBox = 1.0 / BoxSize;
    BhVdArgs ba;
    ba.vel_leaf = ctx->velp_leaf.ptr;
    ba.vel = ctx->vel.ptr;
    ba.out = d_out;
    const long long ntasks = (nq + 255) / 256;
    const unsigned grid = (unsigned) (ntasks < NL_REDO_BLOCKS ? ntasks : NL_REDO_BLOCKS);
    bh_veldisp_kernel<<<dim3(grid), dim3(256), 0, st>>>(a, d_queue, nq, ba, ctx->s_nlist2.ptr, ntasks);
    SHQ_HIP(hipGetLastError());
    return SHQ_OK;
}

/* ---- wind velocity dispersion: winds_find_vel_disp, veldisp2.cpp:203-528 --------------------------
 * The 1-D velocity dispersion of the ~40 nearest dark-matter particles around star-forming gas: a density-like loop over the
 * DM tree with NWINDHSML = 5 trial radii per walk (vdispeffdmradius :216-229, ngbiter :440-479 with the Hubble flow in the
 * relative velocity), WindVDispOutput::postprocess + ngb_narrow_down (:285-320) until 40 +- 1 neighbours.  Same structure
 * as the stellar density; the reference's integer neighbour counts make the result independent of summation order. */
#define WV_NH 5
#define WV_NUMDMNGB 40
#define WV_MAXDEV 1

template <int NH> __device__ __forceinline__ double narrow_down(double &R, double &L, const double *radius, const double *numNgb, int maxcmpt,
                                                                 int desnumngb, int &close, double Box)
{
    /* ngb_narrow_down, treewalk.c:1349-1406, with the dynamic indices written as selects over the NH trial radii */
    close = 0;
    double ngbdist = fabs(numNgb[0] - desnumngb);
#pragma unroll
    for(int k = 1; k < NH; k++) {
        const double nd = fabs(numNgb[k] - desnumngb);
        if(k < maxcmpt && nd < ngbdist) {
            ngbdist = nd;
            close = k;
        }
    }
    bool stop = false;
#pragma unroll
    for(int k = 0; k < NH; k++) {
        if(k < maxcmpt && !stop) {
            if(numNgb[k] < desnumngb)
                L = radius[k];
            if(numNgb[k] > desnumngb) {
                R = radius[k];
                stop = true;
            }
        }
    }
    double hc = radius[0], rl = radius[0], rl1 = radius[0], nl = numNgb[0], nl1 = numNgb[0];
#pragma unroll
    for(int k = 1; k < NH; k++) {
        if(k == close)
            hc = radius[k];
        if(k == maxcmpt - 1) {
            rl = radius[k]; nl = numNgb[k]; rl1 = radius[k - 1]; nl1 = numNgb[k - 1];
        }
    }
    double hs = hc;
    if(R > 0.99 * Box) {
        double dngbdv = 0;
        if(maxcmpt > 1 && rl > rl1)
            dngbdv = (nl - nl1) / (pow(rl, 3) - pow(rl1, 3));
        double newh = 4 * hs;
        if(dngbdv > 0) {
            const double dngb = desnumngb - nl;
            const double nv = pow(hs, 3) + dngb / dngbdv;
            if(pow(nv, 1. / 3) < newh)
                newh = pow(nv, 1. / 3);
        }
        hs = newh;
    }
    if(hs > R)
        hs = R;
    if(L == 0) {
        double dngbdv = 0;
        if(radius[1] > radius[0])
            dngbdv = (numNgb[1] - numNgb[0]) / (pow(radius[1], 3) - pow(radius[0], 3));
        if(maxcmpt == 1 && radius[0] > 0)
            dngbdv = numNgb[0] / pow(radius[0], 3);
        if(dngbdv > 0) {
            const double dngb = desnumngb - numNgb[0];
            const double nv = pow(hs, 3) + dngb / dngbdv;
            hs = pow(nv, 1. / 3);
        }
    }
    if(hs < L)
        hs = L;
    return hs;
}

struct WindVdArgs {
    double Box, hubble_a2;     /* hubble * atime^2 */
    const double4 *vel_leaf;   /* DM_VelPred by leaf slot */
    const double *vel;         /* [N][3] */
    double *dmradius;          /* by particle index: the current DMRadius (starts as Hsml) */
    double *vdisp;             /* by particle index; < 0 where not set */
    int32_t *todo;
};

__global__ __launch_bounds__(256) void wind_veldisp_kernel(const SphDev a, const int32_t *queue, long long nq, const WindVdArgs wa,
                                                           unsigned long long *nint_total, int32_t *__restrict__ nlist, long long ntasks)
{
    __shared__ __attribute__((aligned(32))) char lds[4 * NW_LDS_PER_WAVE(false)];
    const int lane = threadIdx.x & 63;
    for(long long task = xcd_block(blockIdx.x, gridDim.x); task < ntasks; task += gridDim.x) {
    const long long wave = task * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int32_t *myl = nlist + ((size_t) blockIdx.x * 4 + (threadIdx.x >> 6)) * (size_t) (NL_ROWS * 64) + lane;
    const long long t = wave * 64 + lane;
    const bool valid = t < nq;
    long long pi = 0;
    double px = 0, py = 0, pz = 0, vx = 0, vy = 0, vz = 0, dm = 1, L = 0, R = wa.Box;
    if(valid) {
        pi = queue[t];
        const double4 p = a.posm[pi];
        px = p.x; py = p.y; pz = p.z;
        vx = wa.vel[3 * pi]; vy = wa.vel[3 * pi + 1]; vz = wa.vel[3 * pi + 2];
        dm = wa.dmradius[pi];
        L = a.left[pi];
        R = a.right[pi];
    }
    double rad[WV_NH], num[WV_NH], v1x[WV_NH], v1y[WV_NH], v1z[WV_NH], v2[WV_NH];
    {
        /* vdispeffdmradius, veldisp2.cpp:216-229 */
        double right = R, left = L;
        if(right > 0.99 * wa.Box)
            right = dm;
        if(left == 0)
            left = 0.1 * dm;
        const double rvol = pow(right, 3), lvol = pow(left, 3);
#pragma unroll
        for(int k = 0; k < WV_NH; k++) {
            rad[k] = pow((1.0 * k + 1) / (1.0 * WV_NH + 1) * (rvol - lvol) + lvol, 1. / 3);
            num[k] = 0; v1x[k] = 0; v1y[k] = 0; v1z[k] = 0; v2[k] = 0;
        }
    }
    int maxcmpte = WV_NH;
    auto pair = [&](const int s) {
        const double4 q = a.posm_leaf[s];
        const double4 w = wa.vel_leaf[s];
        const double d0 = wrapd(px - q.x, a.Box, a.invBox);
        const double d1 = wrapd(py - q.y, a.Box, a.invBox);
        const double d2 = wrapd(pz - q.z, a.Box, a.invBox);
        const double r2 = d0 * d0 + d1 * d1 + d2 * d2;
        if(r2 <= 0 || !(r2 < rad[WV_NH - 1] * rad[WV_NH - 1]))
            return;
        const double r = sqrt(r2);
        const double e0 = w.x - vx + wa.hubble_a2 * d0, e1 = w.y - vy + wa.hubble_a2 * d1, e2 = w.z - vz + wa.hubble_a2 * d2;
#pragma unroll
        for(int k = 0; k < WV_NH; k++) {
            if(k < maxcmpte && r < rad[k]) {
                num[k] += 1;
                v1x[k] += e0; v2[k] += e0 * e0;
                v1y[k] += e1; v2[k] += e1 * e1;
                v1z[k] += e2; v2[k] += e2 * e2;
            }
        }
        int first = WV_NH;
#pragma unroll
        for(int k = WV_NH - 1; k >= 0; k--)
            first = (num[k] > WV_NUMDMNGB) ? k : first;
        if(first < WV_NH)
            maxcmpte = first + 1;
    };
    const double rw2 = rad[WV_NH - 1] * rad[WV_NH - 1];
    auto accept = [&](const double r2, const double, const int) { return r2 > 0 && r2 < rw2; };
    int fill = 0;
    bool ovf = false;
    unsigned int nint = ngb_walk<false, false, false>(a, lds + (threadIdx.x >> 6) * NW_LDS_PER_WAVE(false), myl, valid, px, py, pz, rad[WV_NH - 1],
                                                      accept, pair, (unsigned int *) nullptr, fill, ovf);
    if(valid) {
        /* WindVDispOutput::postprocess, veldisp2.cpp:285-320 */
        int close = 0;
        const double newdm = narrow_down<WV_NH>(R, L, rad, num, maxcmpte, WV_NUMDMNGB, close, wa.Box);
        double nc = num[0], s0 = v1x[0], s1 = v1y[0], s2 = v1z[0], q2 = v2[0];
#pragma unroll
        for(int k = 1; k < WV_NH; k++)
            if(k == close) {
                nc = num[k]; s0 = v1x[k]; s1 = v1y[k]; s2 = v1z[k]; q2 = v2[k];
            }
        wa.dmradius[pi] = newdm;
        a.left[pi] = L;
        a.right[pi] = R;
        a.numngb[pi] = nc;
        int done = 0;
        if((nc >= (WV_NUMDMNGB - WV_MAXDEV) && nc <= (WV_NUMDMNGB + WV_MAXDEV)) || (R - L < 5e-6 * L)) {
            double vd = q2 / nc;
            vd -= (s0 / nc) * (s0 / nc);
            vd -= (s1 / nc) * (s1 / nc);
            vd -= (s2 / nc) * (s2 / nc);
            if(vd > 0)
                wa.vdisp[pi] = sqrt(vd / 3);
            done = 1;
        }
        wa.todo[t] = done ? -1 : (int32_t) pi;
    }
    unsigned int sn = nint;
    for(int off = 32; off > 0; off >>= 1)
        sn += __shfl_xor(sn, off);
    if(lane == 0 && nint_total)
        atomicAdd(nint_total, (unsigned long long) sn);
    } /* task loop */
}

int shq_wind_veldisp_device(shq_context *ctx, const shq_kick_factors *kf, double BoxSize, double hubble_a2, const int32_t *d_queue, int64_t nq,
                            double *d_dmradius, double *d_vdisp, shq_sph_stats *stats)
{
    const long long n = ctx->numpart;
    const size_t cap = (size_t) (n > 0 ? n : 1);
    SHQ_TRY(ctx->s_numngb.reserve(cap));
    SHQ_TRY(ctx->s_left.reserve(cap));
    SHQ_TRY(ctx->s_right.reserve(cap));
    SHQ_TRY(ctx->s_todo.reserve(cap));
    SHQ_TRY(ctx->s_queue2.reserve(cap));
    SHQ_TRY(ctx->s_queue3.reserve(cap));
    SHQ_TRY(ctx->s_blockcount.reserve(nblk(n) + 1));
    SHQ_TRY(ctx->s_counters.reserve(8));
    SHQ_TRY(ctx->s_nlist2.reserve((size_t) NL_REDO_BLOCKS * 4 * NL_ROWS * 64));
    const long long nl = ctx->ntreeparts + SHQ_NMAXCHILD;
    SHQ_TRY(ctx->velp_leaf.reserve(nl));
    SHQ_TRY(ctx->flag_leaf.reserve(nl));
    hipStream_t st = ctx->stream;
    if(n > 0) {
        SHQ_HIP(hipMemsetAsync(ctx->s_left.ptr, 0, sizeof(double) * n, st));
        fill_kernel<<<dim3(nblk(n)), dim3(256), 0, st>>>(ctx->s_right.ptr, n, BoxSize);
    }
    SHQ_HIP(hipMemsetAsync(ctx->s_counters.ptr, 0, sizeof(long long) * 8, st));
    bh_veldisp_gather_kernel<<<dim3(nblk(nl)), dim3(256), 0, st>>>(nl, ctx->leaf_pidx.ptr, ctx->vel.ptr, ctx->treeacc.ptr, ctx->gravpm.ptr,
                                                                  ctx->bin_grav.ptr, ctx->pflags.ptr, *kf, 1 << 1, ctx->velp_leaf.ptr,
                                                                  ctx->flag_leaf.ptr);
    SHQ_HIP(hipGetLastError());
    SphDev a = make_dev(ctx);
    a.Box = BoxSize;
    a.invBox = 1.0 / BoxSize;
    WindVdArgs wa;
    wa.Box = BoxSize;
    wa.hubble_a2 = hubble_a2;
    wa.vel_leaf = ctx->velp_leaf.ptr;
    wa.vel = ctx->vel.ptr;
    wa.dmradius = d_dmradius;
    wa.vdisp = d_vdisp;
    wa.todo = ctx->s_todo.ptr;
    unsigned long long *nint = reinterpret_cast<unsigned long long *>(ctx->s_counters.ptr + 1);
    long long *total = ctx->s_counters.ptr;
    int32_t *bufs[2] = {ctx->s_queue2.ptr, ctx->s_queue3.ptr};
    int wsel = 0, niter = 0;
    const int32_t *cur = d_queue;
    long long size = nq;
    SHQ_HIP(hipEventRecord(ctx->ev_begin[14], st));
    while(size > 0) {
        const long long ntasks = (size + 255) / 256;
        const unsigned grid = (unsigned) (ntasks < NL_REDO_BLOCKS ? ntasks : NL_REDO_BLOCKS);
        wind_veldisp_kernel<<<dim3(grid), dim3(256), 0, st>>>(a, cur, size, wa, nint, ctx->s_nlist2.ptr, ntasks);
        SHQ_HIP(hipGetLastError());
        niter++;
        const int nb = (int) nblk(size);
        compact_count_kernel<<<dim3(nb), dim3(256), 0, st>>>(ctx->s_todo.ptr, size, ctx->s_blockcount.ptr);
        compact_scan_kernel<<<dim3(1), dim3(1024), 0, st>>>(ctx->s_blockcount.ptr, nb, total);
        compact_write_kernel<<<dim3(nb), dim3(256), 0, st>>>(ctx->s_todo.ptr, size, ctx->s_blockcount.ptr, bufs[wsel]);
        long long newsize = 0;
        SHQ_HIP(hipMemcpyAsync(&newsize, total, sizeof(long long), hipMemcpyDeviceToHost, st));
        SHQ_HIP(hipStreamSynchronize(st));
        size = newsize;
        cur = bufs[wsel];
        wsel ^= 1;
        if(size > 0 && niter > SPH_MAXITER) {
            shq_set_error("failed to converge the wind velocity dispersion for %lld particles", size);
            return SHQ_ERR_NOCONV;
        }
    }
    SHQ_HIP(hipEventRecord(ctx->ev_end[14], st));
    if(stats) {
        unsigned long long h_nint = 0;
        SHQ_HIP(hipMemcpyAsync(&h_nint, nint, sizeof(h_nint), hipMemcpyDeviceToHost, st));
        SHQ_HIP(hipStreamSynchronize(st));
        float ms = 0;
        (void) hipEventElapsedTime(&ms, ctx->ev_begin[14], ctx->ev_end[14]);
        stats->ntargets = nq;
        stats->ninteractions = (int64_t) h_nint;
        stats->niterations = niter;
        stats->kernel_ms = ms;
        stats->hsml_max_tried = 0;
    }
    return SHQ_OK;
}

/* ---- black-hole repositioning and dynamical-friction sums: bhdynfric.cpp:44-295 --------------------
 * BHReposLocalTreeWalk::ngbiter (:160-174): the particle of lowest potential inside the hole's kernel radius (position and
 * velocity kept; first one met in depth-first order on ties).  BHDynFricLocalTreeWalk::ngbiter (:193-224): the same plus the
 * kernel-weighted mass, momentum (DM_VelPred) and squared velocity of the surrounding stars (and dark matter for method > 1).
 * BHDynFricOutput::postprocess (:66-82) normalises.  The tree (ALLMASK, or STARMASK + BHMASK [+ DMMASK]) is the caller's. */
struct BhDfArgs {
    const double4 *vp_leaf;    /* DM_VelPred, weight: 1 if the particle counts for the friction sums */
    const double4 *rv_leaf;    /* raw Vel, Potential */
    double *out;               /* [nq][12]: MinPot, MinPotPos[3], MinPotVel[3], Density, Vel[3], RmsVel (raw sums) */
    int dosums;
};

template <int KT>
__global__ __launch_bounds__(256) void bh_dynfric_kernel(const SphDev a, const int32_t *queue, long long nq, const BhDfArgs da,
                                                         int32_t *__restrict__ nlist, long long ntasks)
{
    __shared__ __attribute__((aligned(32))) char lds[4 * NW_LDS_PER_WAVE(false)];
    const int lane = threadIdx.x & 63;
    for(long long task = xcd_block(blockIdx.x, gridDim.x); task < ntasks; task += gridDim.x) {
    const long long wave = task * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int32_t *myl = nlist + ((size_t) blockIdx.x * 4 + (threadIdx.x >> 6)) * (size_t) (NL_ROWS * 64) + lane;
    const long long t = wave * 64 + lane;
    const bool valid = t < nq;
    double px = 0, py = 0, pz = 0, h = 1;
    if(valid) {
        const long long pi = queue[t];
        const double4 p = a.posm[pi];
        px = p.x; py = p.y; pz = p.z;
        h = a.hsml[pi];
    }
    const Kern<KT> kernel(h);
    const double h2 = kernel.H * kernel.H, Hinv = 1.0 / kernel.H;
    double minpot = 1.0e29 /* BHPOTVALUEINIT */, mp0 = -1, mp1 = -1, mp2 = -1, mv0 = 0, mv1 = 0, mv2 = 0;
    double dens = 0, sv0 = 0, sv1 = 0, sv2 = 0, rms = 0;
    auto pair = [&](const int s) {
        const double4 q = a.posm_leaf[s];
        const double4 rv = da.rv_leaf[s];
        if(rv.w < minpot) {
            minpot = rv.w;
            mp0 = q.x; mp1 = q.y; mp2 = q.z;
            mv0 = rv.x; mv1 = rv.y; mv2 = rv.z;
        }
        if(da.dosums) {
            const double4 vp = da.vp_leaf[s];
            if(vp.w != 0) {
                const double d0 = wrapd(px - q.x, a.Box, a.invBox), d1 = wrapd(py - q.y, a.Box, a.invBox), d2 = wrapd(pz - q.z, a.Box, a.invBox);
                const double u = sqrt(d0 * d0 + d1 * d1 + d2 * d2) * Hinv;
                const double mw = q.w * kernel.wk(u);
                dens += mw;
                sv0 += mw * vp.x; rms += mw * (vp.x * vp.x);
                sv1 += mw * vp.y; rms += mw * (vp.y * vp.y);
                sv2 += mw * vp.z; rms += mw * (vp.z * vp.z);
            }
        }
    };
    auto accept = [&](const double r2, const double, const int) { return r2 < h2; };
    int fill = 0;
    bool ovf = false;
    (void) ngb_walk<false, false, false>(a, lds + (threadIdx.x >> 6) * NW_LDS_PER_WAVE(false), myl, valid, px, py, pz, kernel.H, accept, pair,
                                         (unsigned int *) nullptr, fill, ovf);
    if(valid) {
        double *o = da.out + 12 * t;
        o[0] = minpot; o[1] = mp0; o[2] = mp1; o[3] = mp2; o[4] = mv0; o[5] = mv1; o[6] = mv2;
        o[7] = dens; o[8] = sv0; o[9] = sv1; o[10] = sv2; o[11] = rms;
    }
    } /* task loop */
}

__global__ void bh_dynfric_gather_kernel(long long nleaf, const int32_t *__restrict__ pidx, const double *__restrict__ vel,
                                         const double *__restrict__ treeacc, const double *__restrict__ gravpm, const uint8_t *__restrict__ bin_grav,
                                         const uint8_t *__restrict__ pflags, const double *__restrict__ potential, shq_kick_factors kf,
                                         int typemask, int method, double4 *vp_leaf, double4 *rv_leaf, int32_t *flag_leaf)
{
#pragma clang fp contract(off)
    const long long s = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(s >= nleaf)
        return;
    const long long p = pidx[s];
    double v[3];
    for(int j = 0; j < 3; j++)
        v[j] = vel[3 * p + j] + kf.gravkicks[bin_grav[p]] * treeacc[3 * p + j] + gravpm[3 * p + j] * kf.FgravkickB;
    const unsigned f = pflags[p];
    const int type = f >> 4;
    vp_leaf[s] = make_double4(v[0], v[1], v[2], (type == 4 || (type == 1 && method > 1)) ? 1.0 : 0.0);
    rv_leaf[s] = make_double4(vel[3 * p], vel[3 * p + 1], vel[3 * p + 2], potential[p]);
    flag_leaf[s] = ((f & 1u) || !((1 << type) & typemask)) ? 1 : 0;
}

int shq_bh_dynfric_device(shq_context *ctx, const shq_kick_factors *kf, double BoxSize, int kernel_type, int typemask, int method,
                          const double *d_potential, const int32_t *d_queue, int64_t nq, double *d_out)
{
    if(nq == 0)
        return SHQ_OK;
    SHQ_CHECK(kernel_type == 1 || kernel_type == 2 || kernel_type == 4, SHQ_ERR_INVALID, "unknown DensityKernelType %d", kernel_type);
    const long long nl = ctx->ntreeparts + SHQ_NMAXCHILD;
    SHQ_TRY(ctx->velp_leaf.reserve(nl));
    SHQ_TRY(ctx->hydrec_leaf.reserve((size_t) nl * sizeof(double4) + 128)); /* reused for the raw velocity + potential stream */
    SHQ_TRY(ctx->flag_leaf.reserve(nl));
    SHQ_TRY(ctx->s_nlist2.reserve((size_t) NL_REDO_BLOCKS * 4 * NL_ROWS * 64));
    hipStream_t st = ctx->stream;
    double4 *rv_leaf = reinterpret_cast<double4 *>(ctx->hydrec_leaf.ptr);
    bh_dynfric_gather_kernel<<<dim3(nblk(nl)), dim3(256), 0, st>>>(nl, ctx->leaf_pidx.ptr, ctx->vel.ptr, ctx->treeacc.ptr, ctx->gravpm.ptr,
                                                                  ctx->bin_grav.ptr, ctx->pflags.ptr, d_potential, *kf, typemask, method,
                                                                  ctx->velp_leaf.ptr, rv_leaf, ctx->flag_leaf.ptr);
    SHQ_HIP(hipGetLastError());
    SphDev a = make_dev(ctx);
    a.Box = BoxSize;
    a.invBox = 1.0 / BoxSize;
    BhDfArgs da;
    da.vp_leaf = ctx->velp_leaf.ptr;
    da.rv_leaf = rv_leaf;
    da.out = d_out;
    da.dosums = method > 0;
    const long long ntasks = (nq + 255) / 256;
    const unsigned grid = (unsigned) (ntasks < NL_REDO_BLOCKS ? ntasks : NL_REDO_BLOCKS);
    switch(kernel_type) {
    case 1: bh_dynfric_kernel<1><<<dim3(grid), dim3(256), 0, st>>>(a, d_queue, nq, da, ctx->s_nlist2.ptr, ntasks); break;
    case 2: bh_dynfric_kernel<2><<<dim3(grid), dim3(256), 0, st>>>(a, d_queue, nq, da, ctx->s_nlist2.ptr, ntasks); break;
    default: bh_dynfric_kernel<4><<<dim3(grid), dim3(256), 0, st>>>(a, d_queue, nq, da, ctx->s_nlist2.ptr, ntasks); break;
    }
    SHQ_HIP(hipGetLastError());
    return SHQ_OK;
}

int shq_sph_gradrho_mag(shq_context *ctx, double *d_out)
{
    const long long n = ctx->numpart;
    if(n > 0)
        gradmag_kernel<<<dim3(nblk(n)), dim3(256), 0, ctx->stream>>>(ctx->s_gradrho.ptr, d_out, n);
    SHQ_HIP(hipGetLastError());
    return SHQ_OK;
}

/* ---- black-hole accretion and feedback: libgadget/blackhole.cpp:373-1003 ------------------------------
 * The two legacy-API tree walks of blackhole(): symmetric neighbour search over gas + black holes (treewalk_visit_ngbiter,
 * treewalk.c:925-975: r2 <= max(Hsml_i, Hsml_j)^2), one black hole per lane on the wave-collective walk of the SPH operators.
 *   accretion (ngbiter :471-631, postprocess :373-468): merger marks (BH_SwallowID, the reference's compare-and-swap rule),
 *     stochastic gas swallowing marks (SPH_SwallowID = the largest ID + 1 that drew the particle), the kernel-weighted entropy, gas
 *     velocity and feedback weight around the hole, Bondi-Hoyle rate capped at the Eddington factor, drag, kinetic-feedback state;
 *   feedback (ngbiter :728-876, postprocess :929-965): the marked mergers and gas particles are swallowed (mass, momentum, progenitor
 *     count), thermal energy goes into the unswallowed gas inside the kernel (compare-and-swap on the entropy, temperature cap) or
 *     the accumulated kinetic energy kicks it in a random direction, the hole takes the smallest neighbour time bin.
 * Black-hole slot fields travel as one record per black hole of the particle set, in ascending particle order. */
__device__ __forceinline__ bool bh_timebin_active(int bin, long long cur) /* is_timebin_active, timestep.cpp:132-139 */
{
    if(bin <= 0 || cur <= 0)
        return true;
    return cur % (1ll << bin) == 0;
}

#define BH_ACC_NOUT 8
template <int KT>
__global__ __launch_bounds__(256) void bh_accretion_kernel(const SphDev a, const int32_t *queue, long long nq, const BhWalkArgs w, const shq_kick_factors kf,
                                                           int32_t *__restrict__ nlist, long long ntasks)
{
#pragma clang fp contract(off)
    __shared__ __attribute__((aligned(32))) char lds[4 * NW_LDS_PER_WAVE(true)];
    const int lane = threadIdx.x & 63;
    for(long long task = xcd_block(blockIdx.x, gridDim.x); task < ntasks; task += gridDim.x) {
    const long long wave = task * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int32_t *myl = nlist + ((size_t) blockIdx.x * 4 + (threadIdx.x >> 6)) * (size_t) (NL_ROWS * 64) + lane;
    const long long t = wave * 64 + lane;
    const bool valid = t < nq;
    double px = 0, py = 0, pz = 0, h = 1, imass = 0, ibhmass = 0, idens = 0, imtrack = 0;
    double iv[3] = {0, 0, 0}, ia[3] = {0, 0, 0};
    unsigned long long myid = 0;
    if(valid) {
        const long long pi = queue[t];
        const double4 p = a.posm[pi];
        px = p.x; py = p.y; pz = p.z;
        imass = p.w;
        h = a.hsml[pi];
        myid = w.ids[pi];
        const long long b = shq_bh_ordinal(w.bhp, w.nbh, (int32_t) pi);
        const BhRec &B = w.bh[b];
        ibhmass = B.Mass; idens = B.Density; imtrack = B.Mtrack;
        for(int d = 0; d < 3; d++) {
            iv[d] = w.vel[3 * pi + d];
            ia[d] = w.treeacc[3 * pi + d] + w.gravpm[3 * pi + d] + B.DFAccel[d]; /* blackhole_accretion_copy, :661-662 */
        }
    }
    const Kern<KT> kernel(h);
    const double HH = kernel.H * kernel.H, Hinv = 1.0 / kernel.H;
    const double h2 = h * h;
    int encounter = 0;
    double fws = 0, sment = 0, gv0 = 0, gv1 = 0, gv2 = 0, mgas = 0;
    const double rmerge = 2 * w.P.ForceSoftening / 2.8;
    auto pair = [&](const int s) {
        const long long p = w.leaf_pidx[s];
        const double4 q = a.posm_leaf[s];
        const int type = w.pflags[p] >> 4;
        if(q.w < 0)
            return;
        if(w.P.WindsDecoupleSph && type == 0 && w.delay[p] > 0) /* winds_is_particle_decoupled */
            return;
        if(w.ids[p] == myid)
            return;
        const double d0 = wrapd(px - q.x, a.Box, a.invBox), d1 = wrapd(py - q.y, a.Box, a.invBox), d2 = wrapd(pz - q.z, a.Box, a.invBox);
        const double r2 = d0 * d0 + d1 * d1 + d2 * d2;
        const double r = sqrt(r2);
        if(type == 5 && r < rmerge) {
            encounter = 1;
            const long long ob = shq_bh_ordinal(w.bhp, w.nbh, (int32_t) p);
            int flag = 0;
            if(w.P.RepositionEnabled == 1 || w.P.MergeGravBound == 0)
                flag = 1;
            if(w.P.MergeGravBound == 1 && w.P.RepositionEnabled == 0 && ob >= 0) {
                /* check_grav_bound, :160-180, with DM_VelPred of the other hole */
                const double dx[3] = {d0, d1, d2};
                double KE = 0, PE = 0;
                const int bg = w.bin_grav[p];
                for(int d = 0; d < 3; d++) {
                    const double vp = w.vel[3 * p + d] + kf.gravkicks[bg] * w.treeacc[3 * p + d] + w.gravpm[3 * p + d] * kf.FgravkickB;
                    const double dv = iv[d] - vp;
                    const double da = ia[d] - w.treeacc[3 * p + d] - w.gravpm[3 * p + d] - w.bh[ob].DFAccel[d];
                    KE += 0.5 * (dv * dv);
                    PE += da * dx[d];
                }
                KE /= (w.P.atime * w.P.atime);
                PE /= w.P.atime;
                flag = (PE + KE <= 0);
            }
            if(flag == 1 && ob >= 0) {
                const unsigned long long oid = w.ids[p];
                const bool oactive = bh_timebin_active(w.bin_hydro[p], w.Ti_Current);
                unsigned long long *swal = w.bh_swallow + ob;
                unsigned long long readid = atomicAdd(swal, 0ull);
                for(;;) {
                    unsigned long long newid;
                    if(readid != 0 && readid < myid)
                        newid = myid + 1;
                    else if(readid == 0 && (oid < myid || !oactive))
                        newid = myid + 1;
                    else
                        break;
                    const unsigned long long seen = atomicCAS(swal, readid, newid);
                    if(seen == readid)
                        break;
                    readid = seen;
                }
            }
        }
        if(type == 0 && r2 < HH) {
            const double u = r * Hinv;
            const double wk = kernel.wk(u);
            const double mass_j = q.w;
            sment += (mass_j * wk * w.entropy[p]);
            const double4 vp = a.velp[p]; /* SPH_VelPred */
            gv0 += (mass_j * wk * vp.x);
            gv1 += (mass_j * wk * vp.y);
            gv2 += (mass_j * wk * vp.z);
            double pacc = 0;
            double BHPartMass = imass;
            if(w.P.SeedBHDynMass > 0 && imtrack < w.P.SeedBHDynMass)
                BHPartMass = imtrack;
            if((ibhmass - BHPartMass) > 0 && idens > 0)
                pacc = (ibhmass - BHPartMass) * wk / idens;
            const double rn = w.rnd[w.ids[p] % w.rndsize];
            if(rn < pacc)
                atomicMax(w.sph_swallow + p, myid + 1); /* "prefer to be swallowed by a bigger ID" */
                if(w.touched)
                    w.touched[p] = 1;
            fws += (mass_j * wk);
            if(w.P.BlackHoleKineticOn == 1)
                mgas += mass_j;
        }
    };
    auto accept = [&](const double r2, const double hj, const int) { return r2 <= h2 || r2 <= hj * hj; };
    int fill = 0;
    bool ovf = false;
    (void) ngb_walk<true, false, false>(a, lds + (threadIdx.x >> 6) * NW_LDS_PER_WAVE(true), myl, valid, px, py, pz, h, accept, pair,
                                        (unsigned int *) nullptr, fill, ovf);
    if(valid) {
        double *o = w.out + BH_ACC_NOUT * t;
        o[0] = encounter; o[1] = fws; o[2] = sment; o[3] = gv0; o[4] = gv1; o[5] = gv2; o[6] = mgas; o[7] = 0;
    }
    } /* task loop */
}

/* blackhole_accretion_postprocess (:373-468), one thread per active black hole; out[t] then holds BH_Entropy and
 * BH_SurroundingGasVel normalised, and the hole's record Mdot, Mass, DragAccel (in out), KineticFdbkEnergy, KEflag */
__global__ void bh_accretion_post_kernel(long long nq, const int32_t *queue, const BhWalkArgs w, const shq_kick_factors kf, const double4 *posm, double *post)
{
#pragma clang fp contract(off)
    const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= nq)
        return;
    const long long i = queue[t];
    const long long b = shq_bh_ordinal(w.bhp, w.nbh, (int32_t) i);
    BhRec &B = w.bh[b];
    double *o = w.out + BH_ACC_NOUT * t;
    const shq_bh_params &P = w.P;
    double mdot = 0;
    const double meddington = P.EddingtonConst * B.Mass * P.UnitTime_in_s / P.HubbleParam;
    B.FeedbackWeightSum = o[1];
    double ent = o[2], gv[3] = {o[3], o[4], o[5]};
    if(B.Density > 0) {
        ent /= B.Density;
        for(int k = 0; k < 3; k++)
            gv[k] /= B.Density;
        double bhvel = 0;
        for(int k = 0; k < 3; k++) {
            const double dv = w.vel[3 * i + k] - gv[k];
            bhvel += dv * dv;
        }
        bhvel = sqrt(bhvel);
        bhvel /= P.atime;
        const double rho = B.Density;
        const double rho_proper = rho * P.a3inv;
        double soundspeed = 0; /* blackhole_soundspeed, :147-157 */
        if(rho > 0) {
            soundspeed = sqrt(SPH_GAMMA * ent * pow(rho, SPH_GAMMA - 1));
            soundspeed *= pow(P.atime, -1.5 * (SPH_GAMMA - 1));
        }
        const double norm = pow((soundspeed * soundspeed + bhvel * bhvel), 1.5);
        if(norm > 0)
            mdot = 4. * M_PI * P.BlackHoleAccretionFactor * P.GravInternal * P.GravInternal * B.Mass * B.Mass * rho_proper / norm;
    }
    if(P.BlackHoleEddingtonFactor > 0.0 && mdot > P.BlackHoleEddingtonFactor * meddington)
        mdot = P.BlackHoleEddingtonFactor * meddington;
    B.Mdot = mdot;
    const double dtime = kf.dloga_for_bin[w.bin_hydro[i]] / P.hubble;
    B.Mass += B.Mdot * dtime;
    double drag[3] = {0, 0, 0};
    if(P.BH_DRAG > 0) {
        double fac = 0;
        if(P.BH_DRAG == 1)
            fac = B.Mdot / posm[i].w;
        if(P.BH_DRAG == 2)
            fac = P.BlackHoleEddingtonFactor * meddington / B.Mass;
        fac *= P.atime;
        for(int k = 0; k < 3; k++)
            drag[k] = -(w.vel[3 * i + k] - gv[k]) * fac;
    }
    B.KEflag = 0;
    if(P.BlackHoleKineticOn == 1) {
        const double Edd_ratio = B.Mdot / meddington;
        double lam_thresh = P.BHKE_EddingtonThrFactor;
        const double x = P.BHKE_EddingtonMFactor * pow(B.Mass / P.BHKE_EddingtonMPivot, P.BHKE_EddingtonMIndex);
        if(lam_thresh > x)
            lam_thresh = x;
        if(Edd_ratio < lam_thresh) {
            B.KEflag = 1;
            const double rho_crit_baryon = P.OmegaBaryon * 3 * (P.Hubble * P.Hubble) / (8 * M_PI * P.GravInternal);
            const double rho_sfr = P.BHKE_SfrCritOverDensity * rho_crit_baryon;
            double epsilon = (B.Density / rho_sfr) / P.BHKE_EffRhoFactor;
            if(epsilon > P.BHKE_EffCap)
                epsilon = P.BHKE_EffCap;
            B.KineticFdbkEnergy += epsilon * (B.Mdot * dtime * (P.LightOverUnitVel * P.LightOverUnitVel));
        }
        double KE_thresh = 0.5 * B.VDisp * B.VDisp * o[6];
        KE_thresh *= P.BHKE_InjEnergyThr;
        if(B.VDisp > 0 && B.KineticFdbkEnergy > KE_thresh)
            B.KEflag = 2;
    }
    double *q = post + 8 * t;
    q[0] = ent; q[1] = gv[0]; q[2] = gv[1]; q[3] = gv[2]; q[4] = drag[0]; q[5] = drag[1]; q[6] = drag[2]; q[7] = 0;
}

__global__ void bh_gather_leaf_kernel(long long nleaf, const int32_t *__restrict__ pidx, const uint8_t *__restrict__ pflags, int32_t *flag_leaf)
{
    const long long s = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(s >= nleaf)
        return;
    const unsigned f = pflags[pidx[s]];
    const int type = f >> 4;
    flag_leaf[s] = ((f & 1u) || !(type == 0 || type == 5)) ? 1 : 0; /* IsGarbage; GASMASK + BHMASK (treewalk.c:943-949) */
}

#define BH_FB_NOUT 8
template <int KT>
__global__ __launch_bounds__(256) void bh_feedback_kernel(const SphDev a, const int32_t *queue, long long nq, const BhWalkArgs w, const shq_kick_factors kf,
                                                          int32_t *__restrict__ nlist, long long ntasks)
{
#pragma clang fp contract(off)
    __shared__ __attribute__((aligned(32))) char lds[4 * NW_LDS_PER_WAVE(true)];
    const int lane = threadIdx.x & 63;
    for(long long task = xcd_block(blockIdx.x, gridDim.x); task < ntasks; task += gridDim.x) {
    const long long wave = task * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int32_t *myl = nlist + ((size_t) blockIdx.x * 4 + (threadIdx.x >> 6)) * (size_t) (NL_ROWS * 64) + lane;
    const long long t = wave * 64 + lane;
    const bool valid = t < nq;
    double px = 0, py = 0, pz = 0, h = 1, idens = 0, imtrack = 0, fws = 0, fbenergy = 0, kefb = 0;
    int channel = 0;
    unsigned long long myid = 0;
    if(valid) {
        const long long pi = queue[t];
        const double4 p = a.posm[pi];
        px = p.x; py = p.y; pz = p.z;
        h = a.hsml[pi];
        myid = w.ids[pi];
        const long long b = shq_bh_ordinal(w.bhp, w.nbh, (int32_t) pi);
        const BhRec &B = w.bh[b];
        idens = B.Density; imtrack = B.Mtrack;
        fws = B.FeedbackWeightSum;
        /* blackhole_feedback_copy, :886-909 */
        const double dtime = kf.dloga_for_bin[w.bin_hydro[pi]] / w.P.hubble;
        fbenergy = w.P.BlackHoleFeedbackFactor * 0.1 * B.Mdot * dtime * (w.P.LightOverUnitVel * w.P.LightOverUnitVel);
        if(w.P.BlackHoleKineticOn == 1 && B.KEflag > 0) {
            channel = 1;
            if(B.KEflag == 2)
                kefb = B.KineticFdbkEnergy;
        }
    }
    const Kern<KT> kernel(h);
    const double HH = kernel.H * kernel.H, Hinv = 1.0 / kernel.H;
    const double h2 = h * h;
    int mintimebin = SHQ_TIMEBINS, countprogs = 0;
    double accmass = 0, accbh = 0, mom0 = 0, mom1 = 0, mom2 = 0;
    auto pair = [&](const int s) {
        const long long p = w.leaf_pidx[s];
        const double4 q = a.posm_leaf[s];
        const int type = w.pflags[p] >> 4;
        if(w.ids[p] == myid)
            return;
        if(w.P.WindsDecoupleSph && type == 0 && w.delay[p] > 0)
            return;
        if(type == 5) {
            const long long ob = shq_bh_ordinal(w.bhp, w.nbh, (int32_t) p);
            if(ob < 0 || w.bh_swallow[ob] == 0)
                return;
            if(w.bh_swallow[ob] != myid + 1)
                return;
            BhRec &O = w.bh[ob];
            w.bh_swallowid_out[ob] = w.bh_swallow[ob] - 1;
            atomicOr(reinterpret_cast<unsigned int *>(w.pflags + (p & ~3ll)), 2u << (8 * (p & 3))); /* Swallowed = 1 */
            countprogs += O.CountProgs;
            accbh += O.Mass;
            double othermass = q.w;
            if(w.P.SeedBHDynMass > 0 && imtrack > 0)
                if(O.Mtrack < w.P.SeedBHDynMass)
                    othermass = O.Mtrack;
            accmass += othermass;
            const int bg = w.bin_grav[p];
            const double v0 = w.vel[3 * p] + kf.gravkicks[bg] * w.treeacc[3 * p] + w.gravpm[3 * p] * kf.FgravkickB;
            const double v1 = w.vel[3 * p + 1] + kf.gravkicks[bg] * w.treeacc[3 * p + 1] + w.gravpm[3 * p + 1] * kf.FgravkickB;
            const double v2 = w.vel[3 * p + 2] + kf.gravkicks[bg] * w.treeacc[3 * p + 2] + w.gravpm[3 * p + 2] * kf.FgravkickB;
            mom0 += (othermass * v0);
            mom1 += (othermass * v1);
            mom2 += (othermass * v2);
            return;
        }
        if(type != 0)
            return;
        const double d0 = wrapd(px - q.x, a.Box, a.invBox), d1 = wrapd(py - q.y, a.Box, a.invBox), d2 = wrapd(pz - q.z, a.Box, a.invBox);
        const double r2 = d0 * d0 + d1 * d1 + d2 * d2;
        const unsigned long long mark = w.sph_swallow[p];
        if(mark == 0 && r2 < HH) {
            const int bh = w.bin_hydro[p];
            if(mintimebin > bh)
                mintimebin = bh;
            const double u = sqrt(r2) * Hinv;
            const double mass_j = q.w;
            const double wk = kernel.wk(u);
            if(fws > 0 && fbenergy > 0 && channel == 0 && mass_j > 0) {
                const double injected = fbenergy * mass_j * wk / fws;
                if(w.eeqos && w.eeqos[p])
                    w.heated[p] = 1;
                if(w.touched)
                    w.touched[p] = 1;
                const double enttou = pow(w.density[p] * w.P.a3inv, SPH_GAMMA - 1) / (SPH_GAMMA - 1);
                unsigned long long *eptr = reinterpret_cast<unsigned long long *>(w.entropy + p);
                unsigned long long oldb = atomicAdd(eptr, 0ull);
                for(;;) {
                    /* add_injected_BH_energy, :700-710 */
                    double unew = __longlong_as_double((long long) oldb) * enttou;
                    unew += injected / mass_j;
                    if(unew > w.P.MaxThermalU)
                        unew = w.P.MaxThermalU;
                    const double entnew = unew / enttou;
                    const unsigned long long seen = atomicCAS(eptr, oldb, (unsigned long long) __double_as_longlong(entnew));
                    if(seen == oldb)
                        break;
                    oldb = seen;
                }
            }
            if(kefb > 0 && channel == 1 && idens > 0) {
                const double dvel = sqrt(2 * kefb * wk / idens);
                /* get_random_dir, :712-723 */
                const double theta = acos(2 * w.rnd[(w.ids[p] + 3) % w.rndsize] - 1);
                const double phi = 2 * M_PI * w.rnd[(w.ids[p] + 4) % w.rndsize];
                const double dir[3] = {sin(theta) * cos(phi), sin(theta) * sin(phi), cos(theta)};
                for(int j = 0; j < 3; j++)
                    atomicAdd(w.velw + 3 * p + j, dvel * dir[j]);
                if(w.touched)
                    w.touched[p] = 1;
            }
        }
        if(mark == myid + 1) {
            if(w.touched)
                w.touched[p] = 1;
            accmass += q.w;
            const double4 vp = a.velp[p];
            mom0 += (q.w * vp.x);
            mom1 += (q.w * vp.y);
            mom2 += (q.w * vp.z);
            atomicOr(reinterpret_cast<unsigned int *>(w.pflags + (p & ~3ll)), 1u << (8 * (p & 3))); /* slots_mark_garbage */
        }
    };
    auto accept = [&](const double r2, const double hj, const int) { return r2 <= h2 || r2 <= hj * hj; };
    int fill = 0;
    bool ovf = false;
    (void) ngb_walk<true, false, false>(a, lds + (threadIdx.x >> 6) * NW_LDS_PER_WAVE(true), myl, valid, px, py, pz, h, accept, pair,
                                        (unsigned int *) nullptr, fill, ovf);
    if(valid) {
        double *o = w.out + BH_FB_NOUT * t;
        o[0] = accmass; o[1] = accbh; o[2] = mom0; o[3] = mom1; o[4] = mom2; o[5] = countprogs; o[6] = mintimebin; o[7] = 0;
    }
    } /* task loop */
}

/* blackhole_feedback_postprocess (:929-965), one thread per hole of the feedback queue; out[t] = accreted mass, accreted black-hole
 * mass, momentum[3], progenitors, minTimeBin from the walk */
__global__ void bh_feedback_post_kernel(long long nq, const int32_t *queue, const BhWalkArgs w, double4 *posm)
{
#pragma clang fp contract(off)
    const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= nq)
        return;
    const long long n = queue[t];
    const long long b = shq_bh_ordinal(w.bhp, w.nbh, (int32_t) n);
    BhRec &B = w.bh[b];
    const double *o = w.out + BH_FB_NOUT * t;
    B.CountProgs += (int32_t) o[5];
    if(o[1] > 0)
        B.Mass += o[1];
    if(o[0] > 0) {
        const double accmass = o[0];
        const float pm = (float) posm[n].w;
        for(int k = 0; k < 3; k++)
            w.velw[3 * n + k] = (w.velw[3 * n + k] * pm + o[2 + k]) / (pm + accmass);
        const double SeedBHDynMass = w.P.SeedBHDynMass;
        if(SeedBHDynMass > 0 && B.Mtrack + accmass < SeedBHDynMass)
            B.Mtrack += accmass;
        else if(B.Mtrack < SeedBHDynMass) {
            posm[n].w = (double) (float) (B.Mtrack + accmass);
            B.Mtrack = SeedBHDynMass;
        } else
            posm[n].w = (double) (float) (pm + accmass);
    }
    if(B.KEflag == 2)
        B.KineticFdbkEnergy = 0;
}

static int bh_launch_prep(shq_context *ctx, const shq_kick_factors *kf)
{
    SHQ_TRY(shq_sph_prepare(ctx, kf, nullptr, nullptr)); /* SPH_VelPred of every gas particle, Hsml in leaf order */
    const long long nl = ctx->ntreeparts + SHQ_NMAXCHILD;
    bh_gather_leaf_kernel<<<dim3(nblk(nl)), dim3(256), 0, ctx->stream>>>(nl, ctx->leaf_pidx.ptr, ctx->pflags.ptr, ctx->flag_leaf.ptr);
    SHQ_HIP(hipGetLastError());
    SHQ_TRY(ctx->s_nlist2.reserve((size_t) NL_REDO_BLOCKS * 4 * NL_ROWS * 64));
    return SHQ_OK;
}

int shq_bh_accretion_device(shq_context *ctx, const shq_kick_factors *kf, const BhWalkArgs *w, const int32_t *d_queue, int64_t nq, double *d_post)
{
    if(nq == 0)
        return SHQ_OK;
    const int kt = w->P.DensityKernelType;
    SHQ_CHECK(kt == 1 || kt == 2 || kt == 4, SHQ_ERR_INVALID, "unknown DensityKernelType %d", kt);
    SHQ_TRY(bh_launch_prep(ctx, kf));
    hipStream_t st = ctx->stream;
    SphDev a = make_dev(ctx);
    a.Box = w->P.BoxSize;
    a.invBox = 1.0 / w->P.BoxSize;
    const long long ntasks = (nq + 255) / 256;
    const unsigned grid = (unsigned) (ntasks < NL_REDO_BLOCKS ? ntasks : NL_REDO_BLOCKS);
    switch(kt) {
    case 1: bh_accretion_kernel<1><<<dim3(grid), dim3(256), 0, st>>>(a, d_queue, nq, *w, *kf, ctx->s_nlist2.ptr, ntasks); break;
    case 2: bh_accretion_kernel<2><<<dim3(grid), dim3(256), 0, st>>>(a, d_queue, nq, *w, *kf, ctx->s_nlist2.ptr, ntasks); break;
    default: bh_accretion_kernel<4><<<dim3(grid), dim3(256), 0, st>>>(a, d_queue, nq, *w, *kf, ctx->s_nlist2.ptr, ntasks); break;
    }
    SHQ_HIP(hipGetLastError());
    bh_accretion_post_kernel<<<dim3(nblk(nq)), dim3(256), 0, st>>>(nq, d_queue, *w, *kf, ctx->posm.ptr, d_post);
    SHQ_HIP(hipGetLastError());
    return SHQ_OK;
}

namespace {
struct Marked {
    const uint8_t *mark;
    __device__ bool operator()(const int32_t &i) const { return mark[i] != 0; }
};
__global__ void rows_gather_kernel(long long m, const int32_t *__restrict__ list, const double *__restrict__ vel, const double *__restrict__ entropy,
                                   const double *__restrict__ delay, const double4 *__restrict__ posm, const uint8_t *__restrict__ pflags,
                                   const uint8_t *__restrict__ extra, double *rows)
{
    const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= m)
        return;
    const long long p = list[t];
    double *r = rows + 8 * t;
    r[0] = vel[3 * p];
    r[1] = vel[3 * p + 1];
    r[2] = vel[3 * p + 2];
    r[3] = entropy ? entropy[p] : 0.0;
    r[4] = delay ? delay[p] : 0.0;
    r[5] = posm[p].w;
    r[6] = (double) pflags[p];
    r[7] = extra ? (double) extra[p] : 0.0;
}
} // namespace

int shq_marked_list(shq_context *ctx, const uint8_t *d_mark, int64_t n, int32_t *d_list, int64_t *m)
{
    *m = 0;
    if(n <= 0)
        return SHQ_OK;
    hipStream_t st = ctx->stream;
    SHQ_TRY(ctx->s_counters.reserve(8));
    unsigned long long *d_count = reinterpret_cast<unsigned long long *>(ctx->s_counters.ptr);
    size_t tmp = 0;
    const rocprim::counting_iterator<int32_t> all(0);
    SHQ_HIP(rocprim::select(nullptr, tmp, all, d_list, d_count, (size_t) n, Marked{d_mark}, st));
    SHQ_TRY(ctx->tb.temp.reserve(tmp + 16));
    SHQ_HIP(rocprim::select(ctx->tb.temp.ptr, tmp, all, d_list, d_count, (size_t) n, Marked{d_mark}, st));
    unsigned long long h = 0;
    SHQ_HIP(hipMemcpyAsync(&h, d_count, sizeof(h), hipMemcpyDeviceToHost, st));
    SHQ_HIP(hipStreamSynchronize(st));
    *m = (int64_t) h;
    return SHQ_OK;
}

namespace {
__global__ void u64_gather_kernel(long long m, const int32_t *__restrict__ list, const unsigned long long *__restrict__ src, unsigned long long *out)
{
    const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(t < m)
        out[t] = src[list[t]];
}
} // namespace

int shq_u64_gather(shq_context *ctx, const int32_t *d_list, int64_t m, const unsigned long long *d_src, unsigned long long *d_out)
{
    if(m <= 0)
        return SHQ_OK;
    u64_gather_kernel<<<dim3(nblk(m)), dim3(256), 0, ctx->stream>>>(m, d_list, d_src, d_out);
    SHQ_HIP(hipGetLastError());
    return SHQ_OK;
}

int shq_rows_gather(shq_context *ctx, const int32_t *d_list, int64_t m, const uint8_t *d_extra, double *d_rows)
{
    if(m <= 0)
        return SHQ_OK;
    rows_gather_kernel<<<dim3(nblk(m)), dim3(256), 0, ctx->stream>>>(m, d_list, ctx->vel.ptr, ctx->g_entropy.ptr, ctx->g_delaytime.ptr, ctx->posm.ptr,
                                                                      ctx->pflags.ptr, d_extra, d_rows);
    SHQ_HIP(hipGetLastError());
    return SHQ_OK;
}

int shq_bh_feedback_device(shq_context *ctx, const shq_kick_factors *kf, const BhWalkArgs *w, const int32_t *d_queue, int64_t nq)
{
    if(nq == 0)
        return SHQ_OK;
    const int kt = w->P.DensityKernelType;
    SHQ_CHECK(kt == 1 || kt == 2 || kt == 4, SHQ_ERR_INVALID, "unknown DensityKernelType %d", kt);
    SHQ_TRY(bh_launch_prep(ctx, kf));
    hipStream_t st = ctx->stream;
    SphDev a = make_dev(ctx);
    a.Box = w->P.BoxSize;
    a.invBox = 1.0 / w->P.BoxSize;
    const long long ntasks = (nq + 255) / 256;
    const unsigned grid = (unsigned) (ntasks < NL_REDO_BLOCKS ? ntasks : NL_REDO_BLOCKS);
    switch(kt) {
    case 1: bh_feedback_kernel<1><<<dim3(grid), dim3(256), 0, st>>>(a, d_queue, nq, *w, *kf, ctx->s_nlist2.ptr, ntasks); break;
    case 2: bh_feedback_kernel<2><<<dim3(grid), dim3(256), 0, st>>>(a, d_queue, nq, *w, *kf, ctx->s_nlist2.ptr, ntasks); break;
    default: bh_feedback_kernel<4><<<dim3(grid), dim3(256), 0, st>>>(a, d_queue, nq, *w, *kf, ctx->s_nlist2.ptr, ntasks); break;
    }
    SHQ_HIP(hipGetLastError());
    bh_feedback_post_kernel<<<dim3(nblk(nq)), dim3(256), 0, st>>>(nq, d_queue, *w, ctx->posm.ptr);
    SHQ_HIP(hipGetLastError());
    return SHQ_OK;
}

/* ---- stellar winds from new stars: libgadget/winds.cpp:227-369, 411-447, 510-565 --------------------------
 * Two asymmetric legacy-API walks over the gas tree for the new stars of the step: the total mass of the gas inside the star's
 * Hsml that is not already a wind particle (sfr_wind_weight_ngbiter), then the kick candidates: every such gas particle whose
 * draw Table[(star ID + gas ID) % size] falls below windeff * Mass / TotalWeight is appended to one list of (gas particle, distance,
 * star ID, velocity, thermal energy) — sfr_wind_feedback_ngbiter's StarKick queue.  Which candidate kicks (the nearest star, ties to
 * the smaller star ID) is resolved from the sorted list by the caller of these kernels, as the reference does after its walk. */
__global__ void wind_gather_leaf_kernel(long long nleaf, const int32_t *__restrict__ pidx, const uint8_t *__restrict__ pflags, const double *__restrict__ delay,
                                        int32_t *flag_leaf)
{
    const long long s = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(s >= nleaf)
        return;
    const int p = pidx[s];
    const unsigned f = pflags[p];
    flag_leaf[s] = ((f & 3u) || (f >> 4) != 0 || delay[p] > 0) ? 1 : 0; /* GASMASK, garbage, "skip earlier wind particles" */
}

template <bool KICK>
__global__ __launch_bounds__(256) void wind_walk_kernel(const SphDev a, const int32_t *queue, long long nq, const WindWalkArgs w, int32_t *__restrict__ nlist,
                                                        long long ntasks)
{
#pragma clang fp contract(off)
    __shared__ __attribute__((aligned(32))) char lds[4 * NW_LDS_PER_WAVE(false)];
    const int lane = threadIdx.x & 63;
    for(long long task = xcd_block(blockIdx.x, gridDim.x); task < ntasks; task += gridDim.x) {
    const long long wave = task * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int32_t *myl = nlist + ((size_t) blockIdx.x * 4 + (threadIdx.x >> 6)) * (size_t) (NL_ROWS * 64) + lane;
    const long long t = wave * 64 + lane;
    const bool valid = t < nq;
    double px = 0, py = 0, pz = 0, h = 1, imass = 0, tw = 0, vdisp = 0;
    unsigned long long myid = 0;
    if(valid) {
        const long long pi = queue[t];
        const double4 p = a.posm[pi];
        px = p.x; py = p.y; pz = p.z;
        imass = p.w;
        h = a.hsml[pi];
        if(KICK) {
            myid = w.ids[pi];
            tw = w.totalweight[t];
            vdisp = w.vdisp[t];
        }
    }
    const double h2 = h * h;
    /* get_wind_params, winds.cpp:489-507 */
    double vel = 0, windeff = 0, utherm = 0;
    if(KICK) {
        const double vphys = vdisp / w.P.Time;
        utherm = w.P.WindThermalFactor * 1.5 * vphys * vphys;
        if(w.P.WindModel & 8) {
            windeff = w.P.WindEfficiency;
            vel = w.P.WindSpeed * w.P.Time;
        } else {
            windeff = (w.P.WindSigma0 * w.P.WindSigma0) / (vphys * vphys + 2 * utherm);
            vel = w.P.WindSpeedFactor * vdisp;
        }
        if(vel < w.P.MinWindVelocity * w.P.Time)
            vel = w.P.MinWindVelocity * w.P.Time;
    }
    double sum = 0;
    unsigned int visited = 0;
    auto pair = [&](const int s) {
        const double4 q = a.posm_leaf[s];
        const double d0 = wrapd(px - q.x, a.Box, a.invBox), d1 = wrapd(py - q.y, a.Box, a.invBox), d2 = wrapd(pz - q.z, a.Box, a.invBox);
        const double r = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
        if(r > h)
            return;
        if(!KICK) {
            sum += q.w; /* wk = 1 */
            visited++;
            return;
        }
        if(tw == 0 || vdisp <= 0)
            return;
        const long long p = w.leaf_pidx[s];
        const double prob = windeff * imass / tw;
        const double rn = w.rnd[(myid + w.ids[p]) % w.rndsize];
        if(rn < prob && vel > 0) {
            const unsigned long long k = atomicAdd(w.nkicks, 1ull);
            if(k < w.maxkicks) {
                shq_wind_kick &K = w.kicks[k];
                K.part_index = (int32_t) p;
                K.pad_ = 0;
                K.StarDistance = r;
                K.StarID = myid;
                K.StarKickVelocity = vel;
                K.StarTherm = utherm;
            }
        }
    };
    auto accept = [&](const double r2, const double, const int) { return r2 <= h2; };
    int fill = 0;
    bool ovf = false;
    (void) ngb_walk<false, false, false>(a, lds + (threadIdx.x >> 6) * NW_LDS_PER_WAVE(false), myl, valid, px, py, pz, h, accept, pair,
                                         (unsigned int *) nullptr, fill, ovf);
    if(!KICK) {
        if(valid)
            w.totalweight[t] = sum;
        for(int off = 32; off > 0; off >>= 1)
            visited += __shfl_xor(visited, off);
        if(lane == 0 && visited)
            atomicAdd(w.nvisited, (unsigned long long) visited);
    }
    } /* task loop */
}

/* the StarKick resolution (winds.cpp:330-350) on the device: the candidates sorted by (particle, distance, star ID) — three stable
 * radix sorts, least significant key first — then the first candidate of every particle kicks: wind_do_kick + get_wind_dir, :449-487 */
__global__ void wind_kick_keys_kernel(long long n, const shq_wind_kick *k, int which, unsigned long long *keys, int32_t *idx, const int32_t *order)
{
    const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= n)
        return;
    const int32_t j = order ? order[t] : (int32_t) t;
    const shq_wind_kick &K = k[j];
    keys[t] = which == 0 ? K.StarID : (which == 1 ? (unsigned long long) __double_as_longlong(K.StarDistance) /* >= 0: bits order like values */
                                                   : (unsigned long long) (unsigned) K.part_index);
    idx[t] = j;
}

__global__ void wind_kick_gather_kernel(long long n, const shq_wind_kick *k, const int32_t *order, shq_wind_kick *out)
{
    const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(t < n)
        out[t] = k[order[t]];
}

__global__ void wind_do_kick_kernel(long long n, const shq_wind_kick *k, const WindWalkArgs w, double *vel, double *entropy, const double *density, double *delay,
                                    unsigned long long *napplied, int *odd)
{
#pragma clang fp contract(off)
    const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= n)
        return;
    const shq_wind_kick K = k[t];
    if(t > 0 && k[t - 1].part_index == K.part_index)
        return; /* "Only do the kick for the first particle, which is the closest" */
    const long long other = K.part_index;
    const unsigned long long id = w.ids[other];
    const double theta = acos(2 * w.rnd[(id + 3) % w.rndsize] - 1);
    const double phi = 2 * M_PI * w.rnd[(id + 4) % w.rndsize];
    const double dir[3] = {sin(theta) * cos(phi), sin(theta) * sin(phi), cos(theta)};
    const double v = K.StarKickVelocity, atime = w.P.Time;
    if(v > 0 && atime > 0) {
        for(int j = 0; j < 3; j++)
            vel[3 * other + j] += v * dir[j];
        const double enttou = pow(density[other] / pow(atime, 3), SPH_GAMMA_MINUS1) / SPH_GAMMA_MINUS1;
        entropy[other] += K.StarTherm / enttou;
        if((w.P.WindModel & 2) && w.P.MaxWindFreeTravelTime > 0) { /* winds_ever_decouple */
            double d = w.P.WindFreeTravelLength / (v / atime);
            if(d > w.P.MaxWindFreeTravelTime)
                d = w.P.MaxWindFreeTravelTime;
            delay[other] = d;
        }
    }
    if(!(v > 0) || !isfinite(v) || !isfinite(delay[other]))
        *odd = 1; /* "Odd v", winds.cpp:344 */
    atomicAdd(napplied, 1ull);
}

int shq_wind_resolve_device(shq_context *ctx, const WindWalkArgs *w, long long nk, shq_wind_kick *d_sorted, unsigned long long *d_napplied, int *d_odd, bool apply)
{
    if(nk == 0)
        return SHQ_OK;
    hipStream_t st = ctx->stream;
    SHQ_TRY(ctx->metal_keys[0].reserve((size_t) nk));
    SHQ_TRY(ctx->metal_keys[1].reserve((size_t) nk));
    SHQ_TRY(ctx->s_queue2.reserve((size_t) nk));
    SHQ_TRY(ctx->s_queue3.reserve((size_t) nk));
    int32_t *ord[2] = {ctx->s_queue2.ptr, ctx->s_queue3.ptr};
    const int32_t *cur = nullptr;
    for(int which = 0; which < 3; which++) {
        wind_kick_keys_kernel<<<dim3(nblk(nk)), dim3(256), 0, st>>>(nk, w->kicks, which, ctx->metal_keys[0].ptr, ord[0], cur);
        SHQ_HIP(hipGetLastError());
        size_t tmp = 0;
        SHQ_HIP(rocprim::radix_sort_pairs(nullptr, tmp, ctx->metal_keys[0].ptr, ctx->metal_keys[1].ptr, ord[0], ord[1], (size_t) nk, 0, 64, st));
        SHQ_TRY(ctx->hydrec_leaf.reserve(tmp + 16));
        SHQ_HIP(rocprim::radix_sort_pairs(ctx->hydrec_leaf.ptr, tmp, ctx->metal_keys[0].ptr, ctx->metal_keys[1].ptr, ord[0], ord[1], (size_t) nk, 0, 64, st));
        cur = ord[1];
        std::swap(ord[0], ord[1]); /* the next pass writes its identity-permuted indices over the old input */
    }
    wind_kick_gather_kernel<<<dim3(nblk(nk)), dim3(256), 0, st>>>(nk, w->kicks, cur, d_sorted);
    SHQ_HIP(hipGetLastError());
    if(apply) {
        wind_do_kick_kernel<<<dim3(nblk(nk)), dim3(256), 0, st>>>(nk, d_sorted, *w, ctx->vel.ptr, ctx->g_entropy.ptr, ctx->g_density.ptr, ctx->g_delaytime.ptr, d_napplied,
                                                                  d_odd);
        SHQ_HIP(hipGetLastError());
    }
    return SHQ_OK;
}

int shq_wind_walk_device(shq_context *ctx, const WindWalkArgs *w, const int32_t *d_queue, int64_t nq, bool kick)
{
    if(nq == 0)
        return SHQ_OK;
    hipStream_t st = ctx->stream;
    const long long nl = ctx->ntreeparts + SHQ_NMAXCHILD;
    SHQ_TRY(ctx->flag_leaf.reserve(nl));
    wind_gather_leaf_kernel<<<dim3(nblk(nl)), dim3(256), 0, st>>>(nl, ctx->leaf_pidx.ptr, ctx->pflags.ptr, ctx->g_delaytime.ptr, ctx->flag_leaf.ptr);
    SHQ_HIP(hipGetLastError());
    SHQ_TRY(ctx->s_nlist2.reserve((size_t) NL_REDO_BLOCKS * 4 * NL_ROWS * 64));
    SphDev a = make_dev(ctx);
    a.Box = w->P.BoxSize;
    a.invBox = 1.0 / w->P.BoxSize;
    const long long ntasks = (nq + 255) / 256;
    const unsigned grid = (unsigned) (ntasks < NL_REDO_BLOCKS ? ntasks : NL_REDO_BLOCKS);
    if(kick)
        wind_walk_kernel<true><<<dim3(grid), dim3(256), 0, st>>>(a, d_queue, nq, *w, ctx->s_nlist2.ptr, ntasks);
    else
        wind_walk_kernel<false><<<dim3(grid), dim3(256), 0, st>>>(a, d_queue, nq, *w, ctx->s_nlist2.ptr, ntasks);
    SHQ_HIP(hipGetLastError());
    return SHQ_OK;
}

/* ---- metal return to the gas around dying stars: libgadget/metal_return.cpp:582-667 -----------------------
 * metal_return_ngbiter updates every gas particle inside a star's kernel under a per-particle spin lock: the result depends on the
 * order the stars reach a particle in (float mass, the MaxGasMass cut).  Here the walk (one star per lane, asymmetric, gas tree) only
 * EMITS (gas particle, star, wk) triples; they are sorted by (particle, position of the star in the queue) and one thread per gas
 * particle applies its triples in that order with the reference's arithmetic — the serial loop over the queue, deterministic.  The
 * mass each star gave away is then summed per star in particle order. */
template <int KT, bool EMIT>
__global__ __launch_bounds__(256) void metal_emit_kernel(const SphDev a, const int32_t *queue, long long nq, const MetalWalkArgs w, int32_t *__restrict__ nlist,
                                                         long long ntasks)
{
#pragma clang fp contract(off)
    __shared__ __attribute__((aligned(32))) char lds[4 * NW_LDS_PER_WAVE(false)];
    const int lane = threadIdx.x & 63;
    for(long long task = xcd_block(blockIdx.x, gridDim.x); task < ntasks; task += gridDim.x) {
    const long long wave = task * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int32_t *myl = nlist + ((size_t) blockIdx.x * 4 + (threadIdx.x >> 6)) * (size_t) (NL_ROWS * 64) + lane;
    const long long t = wave * 64 + lane;
    const bool valid = t < nq;
    double px = 0, py = 0, pz = 0, h = 1;
    if(valid) {
        const double4 p = a.posm[queue[t]];
        px = p.x; py = p.y; pz = p.z;
        h = a.hsml[queue[t]];
    }
    const Kern<KT> kernel(h);
    const double HH = kernel.H * kernel.H, Hinv = 1.0 / kernel.H;
    unsigned int mine = 0;
    auto pair = [&](const int s) {
        if(!EMIT) {
            mine++;
            return;
        }
        const double4 q = a.posm_leaf[s];
        const double d0 = wrapd(px - q.x, a.Box, a.invBox), d1 = wrapd(py - q.y, a.Box, a.invBox), d2 = wrapd(pz - q.z, a.Box, a.invBox);
        const double r = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
        double wk = 1;
        if(w.SPHWeighting)
            wk = kernel.wk(r * Hinv);
        const unsigned long long k = atomicAdd(w.cursor, 1ull);
        if(k < w.capacity) {
            w.keys[k] = ((unsigned long long) (unsigned) w.leaf_pidx[s] << 32) | (unsigned long long) t;
            w.wk[k] = wk;
        }
    };
    auto accept = [&](const double r2, const double, const int) { return r2 > 0 && r2 < HH; };
    int fill = 0;
    bool ovf = false;
    (void) ngb_walk<false, false, false>(a, lds + (threadIdx.x >> 6) * NW_LDS_PER_WAVE(false), myl, valid, px, py, pz, h, accept, pair,
                                         (unsigned int *) nullptr, fill, ovf);
    if(!EMIT) {
        for(int off = 32; off > 0; off >>= 1)
            mine += __shfl_xor(mine, off);
        if(lane == 0 && mine)
            atomicAdd(w.cursor, (unsigned long long) mine);
    }
    } /* task loop */
}

/* one thread per run of equal gas particles in the (particle, star)-sorted list: metal_return_ngbiter's body, :622-660 */
__global__ void metal_apply_kernel(long long npairs, const unsigned long long *__restrict__ keys, const double *__restrict__ wk, const MetalWalkArgs w, double *thismass_out)
{
#pragma clang fp contract(off)
    const long long k0 = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(k0 >= npairs)
        return;
    const unsigned p = (unsigned) (keys[k0] >> 32);
    if(k0 > 0 && (unsigned) (keys[k0 - 1] >> 32) == p)
        return;
    float mass = w.gmass[p];
    double density = w.gdensity[p], metallicity = w.gmetallicity[p];
    float metals[SHQ_NMETALS];
    for(int i = 0; i < SHQ_NMETALS; i++)
        metals[i] = w.gmetals[(size_t) p * SHQ_NMETALS + i];
    for(long long k = k0; k < npairs && (unsigned) (keys[k] >> 32) == p; k++) {
        const unsigned t = (unsigned) (keys[k] & 0xffffffffull);
        const double volume = mass / density;
        const double returnfraction = wk[k] * volume / w.starvolume[t];
        const double thismass = returnfraction * w.massgenerated[t];
        if(mass + thismass > w.MaxGasMass) {
            thismass_out[k] = 0;
            continue;
        }
        for(int i = 0; i < SHQ_NMETALS; i++) {
            const double tm = returnfraction * w.speciesgenerated[(size_t) t * SHQ_NMETALS + i];
            metals[i] = (float) ((metals[i] * mass + tm) / (mass + thismass));
        }
        const double thismetal = returnfraction * w.metalgenerated[t];
        metallicity = (metallicity * mass + thismetal) / (mass + thismass);
        const double massfrac = (mass + thismass) / mass;
        mass = (float) (mass * massfrac);
        density *= massfrac;
        thismass_out[k] = thismass;
    }
    w.gmass[p] = mass;
    w.gdensity[p] = density;
    w.gmetallicity[p] = metallicity;
    for(int i = 0; i < SHQ_NMETALS; i++)
        w.gmetals[(size_t) p * SHQ_NMETALS + i] = metals[i];
    if(w.touched)
        w.touched[p] = 1;
}

__global__ void metal_rows_gather_kernel(long long m, const int32_t *__restrict__ list, const MetalWalkArgs w, double *rows)
{
    const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= m)
        return;
    const size_t p = (size_t) list[t];
    double *r = rows + (size_t) (3 + SHQ_NMETALS) * t;
    r[0] = (double) w.gmass[p];
    r[1] = w.gdensity[p];
    r[2] = w.gmetallicity[p];
    for(int i = 0; i < SHQ_NMETALS; i++)
        r[3 + i] = (double) w.gmetals[p * SHQ_NMETALS + i];
}

__global__ void metal_rekey_kernel(long long npairs, const unsigned long long *keys, unsigned long long *out)
{
    const long long k = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(k < npairs)
        out[k] = (keys[k] << 32) | (keys[k] >> 32); /* (star, particle) */
}

/* O->MassReturn += thismass over a star's neighbours, in particle order */
__global__ void metal_sum_kernel(long long npairs, const unsigned long long *__restrict__ keys_tp, const double *__restrict__ thismass, double *massreturn)
{
#pragma clang fp contract(off)
    const long long k0 = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(k0 >= npairs)
        return;
    const unsigned t = (unsigned) (keys_tp[k0] >> 32);
    if(k0 > 0 && (unsigned) (keys_tp[k0 - 1] >> 32) == t)
        return;
    double s = 0;
    for(long long k = k0; k < npairs && (unsigned) (keys_tp[k] >> 32) == t; k++)
        s += thismass[k];
    massreturn[t] = s;
}

int shq_metal_rows_gather(shq_context *ctx, const MetalWalkArgs *w, const int32_t *d_list, int64_t m, double *d_rows)
{
    if(m <= 0)
        return SHQ_OK;
    metal_rows_gather_kernel<<<dim3(nblk(m)), dim3(256), 0, ctx->stream>>>(m, d_list, *w, d_rows);
    SHQ_HIP(hipGetLastError());
    return SHQ_OK;
}

int shq_metal_return_device(shq_context *ctx, MetalWalkArgs *w, int kernel_type, double BoxSize, const int32_t *d_queue, int64_t nq, double *d_massreturn, int64_t *npairs_out)
{
    if(npairs_out)
        *npairs_out = 0;
    if(nq == 0)
        return SHQ_OK;
    SHQ_CHECK(kernel_type == 1 || kernel_type == 2 || kernel_type == 4, SHQ_ERR_INVALID, "unknown DensityKernelType %d", kernel_type);
    hipStream_t st = ctx->stream;
    const long long nl = ctx->ntreeparts + SHQ_NMAXCHILD;
    SHQ_TRY(ctx->flag_leaf.reserve(nl));
    /* GASMASK, not garbage; wind particles take metals like any other gas */
    bh_gather_leaf_kernel<<<dim3(nblk(nl)), dim3(256), 0, st>>>(nl, ctx->leaf_pidx.ptr, ctx->pflags.ptr, ctx->flag_leaf.ptr);
    SHQ_HIP(hipGetLastError());
    SHQ_TRY(ctx->s_nlist2.reserve((size_t) NL_REDO_BLOCKS * 4 * NL_ROWS * 64));
    SHQ_TRY(ctx->wind_cnt.reserve(4));
    SphDev a = make_dev(ctx);
    a.Box = BoxSize;
    a.invBox = 1.0 / BoxSize;
    const long long ntasks = (nq + 255) / 256;
    const unsigned grid = (unsigned) (ntasks < NL_REDO_BLOCKS ? ntasks : NL_REDO_BLOCKS);
    w->cursor = ctx->wind_cnt.ptr;
    w->leaf_pidx = ctx->leaf_pidx.ptr;
    unsigned long long np = 0;
    for(int pass = 0; pass < 2; pass++) {
        SHQ_HIP(hipMemsetAsync(ctx->wind_cnt.ptr, 0, sizeof(unsigned long long), st));
        if(pass == 0) {
            switch(kernel_type) {
            case 1: metal_emit_kernel<1, false><<<dim3(grid), dim3(256), 0, st>>>(a, d_queue, nq, *w, ctx->s_nlist2.ptr, ntasks); break;
            case 2: metal_emit_kernel<2, false><<<dim3(grid), dim3(256), 0, st>>>(a, d_queue, nq, *w, ctx->s_nlist2.ptr, ntasks); break;
            default: metal_emit_kernel<4, false><<<dim3(grid), dim3(256), 0, st>>>(a, d_queue, nq, *w, ctx->s_nlist2.ptr, ntasks); break;
            }
        } else {
            switch(kernel_type) {
            case 1: metal_emit_kernel<1, true><<<dim3(grid), dim3(256), 0, st>>>(a, d_queue, nq, *w, ctx->s_nlist2.ptr, ntasks); break;
            case 2: metal_emit_kernel<2, true><<<dim3(grid), dim3(256), 0, st>>>(a, d_queue, nq, *w, ctx->s_nlist2.ptr, ntasks); break;
            default: metal_emit_kernel<4, true><<<dim3(grid), dim3(256), 0, st>>>(a, d_queue, nq, *w, ctx->s_nlist2.ptr, ntasks); break;
            }
        }
        SHQ_HIP(hipGetLastError());
        unsigned long long h = 0;
        SHQ_HIP(hipMemcpyAsync(&h, ctx->wind_cnt.ptr, sizeof(h), hipMemcpyDeviceToHost, st));
        SHQ_HIP(hipStreamSynchronize(st));
        if(pass == 0) {
            np = h;
            if(np == 0)
                break;
            SHQ_TRY(ctx->metal_keys[0].reserve((size_t) np));
            SHQ_TRY(ctx->metal_keys[1].reserve((size_t) np));
            SHQ_TRY(ctx->metal_val[0].reserve((size_t) np));
            SHQ_TRY(ctx->metal_val[1].reserve((size_t) np));
            w->keys = ctx->metal_keys[0].ptr;
            w->wk = ctx->metal_val[0].ptr;
            w->capacity = np;
        } else
            SHQ_CHECK(h == np, SHQ_ERR_STATE, "metal_return: the two walks disagree on the number of pairs (%llu, %llu)", np, h);
    }
    if(npairs_out)
        *npairs_out = (int64_t) np;
    SHQ_HIP(hipMemsetAsync(d_massreturn, 0, sizeof(double) * (size_t) nq, st));
    if(np == 0)
        return SHQ_OK;
    size_t tmp = 0;
    SHQ_HIP(rocprim::radix_sort_pairs(nullptr, tmp, ctx->metal_keys[0].ptr, ctx->metal_keys[1].ptr, ctx->metal_val[0].ptr, ctx->metal_val[1].ptr, (size_t) np, 0, 64, st));
    SHQ_TRY(ctx->wind_kicks.reserve(tmp + 16));
    SHQ_HIP(rocprim::radix_sort_pairs(ctx->wind_kicks.ptr, tmp, ctx->metal_keys[0].ptr, ctx->metal_keys[1].ptr, ctx->metal_val[0].ptr, ctx->metal_val[1].ptr, (size_t) np, 0, 64,
                                      st));
    /* thismass per pair, in (particle, star) order, into metal_val[0] */
    metal_apply_kernel<<<dim3(nblk((long long) np)), dim3(256), 0, st>>>((long long) np, ctx->metal_keys[1].ptr, ctx->metal_val[1].ptr, *w, ctx->metal_val[0].ptr);
    SHQ_HIP(hipGetLastError());
    metal_rekey_kernel<<<dim3(nblk((long long) np)), dim3(256), 0, st>>>((long long) np, ctx->metal_keys[1].ptr, ctx->metal_keys[0].ptr);
    SHQ_HIP(hipGetLastError());
    SHQ_HIP(rocprim::radix_sort_pairs(nullptr, tmp, ctx->metal_keys[0].ptr, ctx->metal_keys[1].ptr, ctx->metal_val[0].ptr, ctx->metal_val[1].ptr, (size_t) np, 0, 64, st));
    SHQ_TRY(ctx->wind_kicks.reserve(tmp + 16));
    SHQ_HIP(rocprim::radix_sort_pairs(ctx->wind_kicks.ptr, tmp, ctx->metal_keys[0].ptr, ctx->metal_keys[1].ptr, ctx->metal_val[0].ptr, ctx->metal_val[1].ptr, (size_t) np, 0, 64,
                                      st));
    metal_sum_kernel<<<dim3(nblk((long long) np)), dim3(256), 0, st>>>((long long) np, ctx->metal_keys[1].ptr, ctx->metal_val[1].ptr, d_massreturn);
    SHQ_HIP(hipGetLastError());
    return SHQ_OK;
}

/* ---- the wind model's particle loops: winds_evolve (winds.cpp:370-387) and winds_subgrid / winds_make_after_sf (:272-292, 567-585) -- */
__global__ void winds_evolve_kernel(long long n, const int32_t *list, const uint8_t *pflags, const uint8_t *bin_hydro, const double *density, double *delay,
                                    double a3inv, double hubble, double DensThresh, double MaxTravelTime, shq_kick_factors kf)
{
#pragma clang fp contract(off)
    const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= n)
        return;
    const long long i = list ? (long long) list[t] : t;
    const unsigned f = pflags[i];
    if((f >> 4) != 0 || (f & 1u))
        return;
    double d = delay[i];
    if(d > 0 && density[i] * a3inv < DensThresh)
        d = 0;
    if(d > 0) {
        if(d > MaxTravelTime)
            d = MaxTravelTime;
        const double dtime = kf.dloga_for_bin[bin_hydro[i]] / hubble;
        d = fmax(d - dtime, 0);
    }
    delay[i] = d;
}

__global__ void winds_subgrid_kernel(long long n, const int32_t *list, const double *stellarmass, const double *vdisp, const double4 *posm, const WindWalkArgs w, double *vel,
                                     double *entropy, const double *density, double *delay, unsigned long long *nkicked)
{
#pragma clang fp contract(off)
    const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= n)
        return;
    const long long i = list ? (long long) list[t] : t;
    /* get_wind_params, :489-507 */
    const double time = w.P.Time;
    const double vphys = vdisp[t] / time;
    const double utherm = w.P.WindThermalFactor * 1.5 * vphys * vphys;
    double windeff, v;
    if(w.P.WindModel & 8) {
        windeff = w.P.WindEfficiency;
        v = w.P.WindSpeed * time;
    } else {
        windeff = (w.P.WindSigma0 * w.P.WindSigma0) / (vphys * vphys + 2 * utherm);
        v = w.P.WindSpeedFactor * vdisp[t];
    }
    if(v < w.P.MinWindVelocity * time)
        v = w.P.MinWindVelocity * time;
    /* winds_make_after_sf: the Springel & Hernquist 03 probability */
    const double pw = windeff * stellarmass[t] / posm[i].w;
    const double prob = 1 - exp(-pw);
    const unsigned long long id = w.ids[i];
    if(!(w.rnd[(id + 2) % w.rndsize] < prob))
        return;
    if(v > 0 && time > 0) { /* wind_do_kick */
        const double theta = acos(2 * w.rnd[(id + 3) % w.rndsize] - 1);
        const double phi = 2 * M_PI * w.rnd[(id + 4) % w.rndsize];
        const double dir[3] = {sin(theta) * cos(phi), sin(theta) * sin(phi), cos(theta)};
        for(int j = 0; j < 3; j++)
            vel[3 * i + j] += v * dir[j];
        const double enttou = pow(density[i] / pow(time, 3), SPH_GAMMA_MINUS1) / SPH_GAMMA_MINUS1;
        entropy[i] += utherm / enttou;
        if((w.P.WindModel & 2) && w.P.MaxWindFreeTravelTime > 0) {
            double d = w.P.WindFreeTravelLength / (v / time);
            if(d > w.P.MaxWindFreeTravelTime)
                d = w.P.MaxWindFreeTravelTime;
            delay[i] = d;
        }
        atomicAdd(nkicked, 1ull);
    }
}

int shq_winds_evolve_device(shq_context *ctx, const int32_t *d_list, int64_t n, double a3inv, double hubble, double DensThresh, double MaxTravelTime,
                            const shq_kick_factors *kf)
{
    if(n > 0)
        winds_evolve_kernel<<<dim3(nblk(n)), dim3(256), 0, ctx->stream>>>(n, d_list, ctx->pflags.ptr, ctx->bin_hydro.ptr, ctx->g_density.ptr, ctx->g_delaytime.ptr, a3inv,
                                                                         hubble, DensThresh, MaxTravelTime, *kf);
    SHQ_HIP(hipGetLastError());
    return SHQ_OK;
}

int shq_winds_subgrid_device(shq_context *ctx, const WindWalkArgs *w, const int32_t *d_list, int64_t n, const double *d_stellarmass, const double *d_vdisp,
                             unsigned long long *d_nkicked)
{
    if(n > 0)
        winds_subgrid_kernel<<<dim3(nblk(n)), dim3(256), 0, ctx->stream>>>(n, d_list, d_stellarmass, d_vdisp, ctx->posm.ptr, *w, ctx->vel.ptr, ctx->g_entropy.ptr,
                                                                          ctx->g_density.ptr, ctx->g_delaytime.ptr, d_nkicked);
    SHQ_HIP(hipGetLastError());
    return SHQ_OK;
}
