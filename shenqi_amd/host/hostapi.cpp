/* hostapi.cpp — flat C test/driver API over the host mirror (ctypes-friendly). */
#include "gravity.hpp"
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <random>
#include <string>

static thread_local std::string g_err;
extern "C" void shqh_set_error(const char *msg) { g_err = msg ? msg : ""; }
extern "C" const char *shqh_last_error(void) { return g_err.c_str(); }

static double g_kernels[SHQ_NGRAVTAB][5];

extern "C" {

part_manager_type *shqh_partmanager_create(particle_data *base, int64_t n, double BoxSize)
{
    part_manager_type *pm = (part_manager_type *) calloc(1, sizeof(part_manager_type));
    pm->Base = base;
    pm->NumPart = n;
    pm->MaxPart = n;
    pm->BoxSize = BoxSize;
    return pm;
}
void shqh_partmanager_free(part_manager_type *pm) { free(pm); }

ForceTree *shqh_force_tree_rebuild_mask(part_manager_type *pm, int mask, const int *active, int64_t nactive, int full)
{
    ForceTree *t = (ForceTree *) calloc(1, sizeof(ForceTree));
    ActiveParticles act;
    memset(&act, 0, sizeof(act));
    act.ActiveParticle = (int *) active;
    act.NumActiveParticle = nactive;
    int rc = force_tree_rebuild_mask(t, pm, mask, active ? &act : nullptr, 1);
    if(rc != 0) {
        char buf[128];
        snprintf(buf, sizeof(buf), "force_tree_rebuild_mask failed with code %d%s", rc,
                 rc == 2 ? " (more than NMAXCHILD particles at one position)" : "");
        shqh_set_error(buf);
        free(t);
        return nullptr;
    }
    if(full)
        t->full_particle_tree_flag = 1;
    return t;
}
void shqh_force_tree_free(ForceTree *t)
{
    if(!t)
        return;
    force_tree_free(t);
    free(t);
}
void shqh_tree_info(const ForceTree *t, int64_t out[5])
{
    out[0] = t->firstnode;
    out[1] = t->lastnode;
    out[2] = t->numnodes;
    out[3] = t->NumParticles;
    out[4] = t->full_particle_tree_flag;
}
const NODE *shqh_tree_nodes(const ForceTree *t) { return t->Nodes_base; }
const int *shqh_tree_father(const ForceTree *t) { return t->Father; }
void shqh_tree_view(const ForceTree *t, shq_tree_view *out) { *out = force_tree_view(t); }
void shqh_part_view(part_manager_type *pm, shq_part_view *out) { *out = make_part_view(pm->Base, pm->NumPart); }

int shqh_set_kernel_table(const double *table)
{
    memcpy(g_kernels, table, sizeof(g_kernels));
    return gravshort_set_kernel_table(g_kernels);
}
void shqh_set_gravshort_treepar(double ErrTolForceAcc, double BHOpeningAngle, double MaxBHOpeningAngle, int TreeUseBH,
                                double Rcut, double FractionalGravitySoftening, int windowtype)
{
    struct gravshort_tree_params p;
    memset(&p, 0, sizeof(p));
    p.ErrTolForceAcc = ErrTolForceAcc;
    p.BHOpeningAngle = BHOpeningAngle;
    p.MaxBHOpeningAngle = MaxBHOpeningAngle;
    p.TreeUseBH = TreeUseBH;
    p.Rcut = Rcut;
    p.FractionalGravitySoftening = FractionalGravitySoftening;
    p.MaxExportBufferBytes = 3584 * 1024 * 1024L;
    p.ShortRangeForceWindowType = (enum ShortRangeForceWindowType) windowtype;
    set_gravshort_treepar(p);
}
int shqh_get_TreeUseBH(void) { return get_gravshort_treepar().TreeUseBH; }
void shqh_gravshort_set_softenings(double MeanSeparation) { gravshort_set_softenings(MeanSeparation); }
double shqh_FORCE_SOFTENING(void) { return FORCE_SOFTENING(); }

int shqh_make_grav_params(double BoxSize, double Asmth, int Nmesh, double G, double rho0, shq_grav_params *out)
{
    PetaPM pm;
    gravpm_init_periodic(&pm, BoxSize, Asmth, Nmesh, G);
    return make_grav_params(&pm, BoxSize, rho0, out);
}

int shqh_grav_short_tree(shq_context *ctx, part_manager_type *pmgr, ForceTree *tree, double Asmth, int Nmesh, double G,
                         const int *active, int64_t nactive, double *AccelStore, double rho0, int UseGPU, int walk_mode,
                         shq_walk_stats *stats)
{
    PetaPM pm;
    gravpm_init_periodic(&pm, pmgr->BoxSize, Asmth, Nmesh, G);
    ActiveParticles act;
    memset(&act, 0, sizeof(act));
    act.ActiveParticle = (int *) active;
    act.NumActiveParticle = active ? nactive : pmgr->NumPart;
    return grav_short_tree(ctx, &act, &pm, tree, pmgr, (MyFloat (*)[3]) AccelStore, rho0, 0, UseGPU != 0, walk_mode, stats);
}

int shqh_gravpm_force(shq_context *ctx, part_manager_type *pmgr, double Asmth, int Nmesh, double G, int UseGPU)
{
    PetaPM pm;
    gravpm_init_periodic(&pm, pmgr->BoxSize, Asmth, Nmesh, G);
    return gravpm_force(ctx, &pm, pmgr, UseGPU != 0);
}

/* Synthetic inputs of SURVEY.md §8(d): kind 0 S-grid, 1 S-uniform, 2 S-cluster
 * (mirrors tests/test_gravity.cpp:316-341 with a 64-bit engine). n3 = particles per dimension
 * for the grid; n = total particles otherwise. */
void shqh_synth_positions(int kind, int64_t n, uint64_t seed, double L, double *pos)
{
    if(kind == 0) {
        const int64_t nc = (int64_t) llround(cbrt((double) n));
        for(int64_t i = 0; i < n; i++) {
            pos[3 * i] = (L / nc) * (i / nc / nc);
            pos[3 * i + 1] = (L / nc) * ((i / nc) % nc);
            pos[3 * i + 2] = (L / nc) * (i % nc);
        }
        return;
    }
    std::mt19937_64 gen(seed);
    auto u01 = [&]() { return (double) (gen() >> 11) * (1.0 / 9007199254740992.0); };
    for(int64_t i = 0; i < n; i++) {
        for(int j = 0; j < 3; j++) {
            double v;
            if(kind == 1 || i < n / 4)
                v = L * u01();
            else if(i < 3 * n / 4)
                v = L / 2 + L / 8 * exp(pow(u01() - 0.5, 2));
            else
                v = L * 0.1 + L / 32 * exp(pow(u01() - 0.5, 2));
            pos[3 * i + j] = v;
        }
    }
}

/* Particles [first, first + count) of ONE global synthetic set of n particles (same kinds as above): the set is made of blocks of
 * 65536 particles, each with its own engine seeded from (seed, block), so any rank can generate any range of it and the union over
 * ranks does not depend on how many ranks there are (bench.py --gpus N: one S-cluster cut across the ranks, not N of them). */
void shqh_synth_positions_range(int kind, int64_t n, int64_t first, int64_t count, uint64_t seed, double L, double *pos)
{
    const int64_t B = 65536;
    if(kind == 0) {
        const int64_t nc = (int64_t) llround(cbrt((double) n));
        for(int64_t k = 0; k < count; k++) {
            const int64_t i = first + k;
            pos[3 * k] = (L / nc) * (i / nc / nc);
            pos[3 * k + 1] = (L / nc) * ((i / nc) % nc);
            pos[3 * k + 2] = (L / nc) * (i % nc);
        }
        return;
    }
    const int64_t b0 = first / B, b1 = (first + count + B - 1) / B;
#pragma omp parallel for schedule(dynamic, 1)
    for(int64_t b = b0; b < b1; b++) {
        std::mt19937_64 gen(seed ^ ((uint64_t) (b + 1) * 0x9E3779B97F4A7C15ull));
        auto u01 = [&]() { return (double) (gen() >> 11) * (1.0 / 9007199254740992.0); };
        for(int64_t i = b * B; i < (b + 1) * B && i < n; i++)
            for(int j = 0; j < 3; j++) {
                const double u = u01();
                if(i < first || i >= first + count)
                    continue;
                double v;
                if(kind == 1 || i < n / 4)
                    v = L * u;
                else if(i < 3 * n / 4)
                    v = L / 2 + L / 8 * exp(pow(u - 0.5, 2));
                else
                    v = L * 0.1 + L / 32 * exp(pow(u - 0.5, 2));
                pos[3 * (i - first) + j] = v;
            }
    }
}

/* Sort particle indices along a Morton (Z-order) key of `bits` bits per dimension so that
 * consecutive particles are spatially close (the reference keeps particles in Peano-Hilbert
 * order, domain.cpp:268; any space-filling order gives compact target groups). */
void shqh_morton_order(const double *pos, int64_t n, double L, int32_t *order)
{
    struct KV { uint64_t k; int32_t i; };
    KV *kv = (KV *) malloc(sizeof(KV) * (size_t) (n > 0 ? n : 1));
    const double scale = (double) (1 << 21) / (L * 1.001);
#pragma omp parallel for
    for(int64_t i = 0; i < n; i++) {
        uint64_t key = 0;
        uint64_t c[3];
        for(int j = 0; j < 3; j++) {
            double v = (pos[3 * i + j] + L / 2000.) * scale;
            if(v < 0) v = 0;
            if(v > (double) ((1 << 21) - 1)) v = (double) ((1 << 21) - 1);
            c[j] = (uint64_t) v;
        }
        for(int b = 20; b >= 0; b--)
            key = (key << 3) | (((c[2] >> b) & 1) << 2) | (((c[1] >> b) & 1) << 1) | ((c[0] >> b) & 1);
        kv[i].k = key;
        kv[i].i = (int32_t) i;
    }
    qsort(kv, (size_t) n, sizeof(KV), [](const void *a, const void *b) {
        const KV *x = (const KV *) a, *y = (const KV *) b;
        if(x->k != y->k) return x->k < y->k ? -1 : 1;
        return x->i < y->i ? -1 : (x->i > y->i ? 1 : 0);
    });
    for(int64_t i = 0; i < n; i++)
        order[i] = kv[i].i;
    free(kv);
}

/* Same, along a Peano-Hilbert key (the order the reference keeps particles in, domain.cpp:268 via
 * peano.c): consecutive particles are face-adjacent in space, without the long jumps of the Z-order,
 * so a 64-particle target group is more compact and its union walk shorter.  The key is built with
 * Skilling's axes-to-transpose transform ("Programming the Hilbert curve", AIP Conf. Proc. 707, 2004). */
void shqh_hilbert_order(const double *pos, int64_t n, double L, int32_t *order)
{
    struct KV { uint64_t k; int32_t i; };
    KV *kv = (KV *) malloc(sizeof(KV) * (size_t) (n > 0 ? n : 1));
    const int bits = 21;
    const double scale = (double) (1 << bits) / (L * 1.001);
#pragma omp parallel for
    for(int64_t i = 0; i < n; i++) {
        uint32_t X[3];
        for(int j = 0; j < 3; j++) {
            double v = (pos[3 * i + j] + L / 2000.) * scale;
            if(v < 0) v = 0;
            if(v > (double) ((1 << bits) - 1)) v = (double) ((1 << bits) - 1);
            X[j] = (uint32_t) v;
        }
        const uint32_t M = 1u << (bits - 1);
        for(uint32_t Q = M; Q > 1; Q >>= 1) { /* inverse undo of the excess work */
            const uint32_t P = Q - 1;
            for(int j = 0; j < 3; j++) {
                if(X[j] & Q)
                    X[0] ^= P;
                else {
                    const uint32_t t = (X[0] ^ X[j]) & P;
                    X[0] ^= t;
                    X[j] ^= t;
                }
            }
        }
        for(int j = 1; j < 3; j++) /* Gray encode */
            X[j] ^= X[j - 1];
        uint32_t t = 0;
        for(uint32_t Q = M; Q > 1; Q >>= 1)
            if(X[2] & Q)
                t ^= Q - 1;
        for(int j = 0; j < 3; j++)
            X[j] ^= t;
        uint64_t key = 0;
        for(int b = bits - 1; b >= 0; b--)
            key = (key << 3) | ((uint64_t) ((X[0] >> b) & 1) << 2) | ((uint64_t) ((X[1] >> b) & 1) << 1) | (uint64_t) ((X[2] >> b) & 1);
        kv[i].k = key;
        kv[i].i = (int32_t) i;
    }
    qsort(kv, (size_t) n, sizeof(KV), [](const void *a, const void *b) {
        const KV *x = (const KV *) a, *y = (const KV *) b;
        if(x->k != y->k) return x->k < y->k ? -1 : 1;
        return x->i < y->i ? -1 : (x->i > y->i ? 1 : 0);
    });
    for(int64_t i = 0; i < n; i++)
        order[i] = kv[i].i;
    free(kv);
}

} /* extern "C" */

/* ---- SPH test/driver API ---------------------------------------------------------------------- */
#include "density.hpp"
extern "C" {

void shqh_set_densitypar(double eta, double MaxNumNgbDeviation, int kernel, double BlackHoleNgbFactor, double MinGasHsml)
{
    struct density_params dp;
    memset(&dp, 0, sizeof(dp));
    dp.DensityResolutionEta = eta;
    dp.MaxNumNgbDeviation = MaxNumNgbDeviation;
    dp.DensityKernelType = (enum DensityKernelType) kernel;
    dp.BlackHoleNgbFactor = BlackHoleNgbFactor;
    dp.MinGasHsml = MinGasHsml;
    set_densitypar(dp);
}
double shqh_GetNumNgb(void) { return GetNumNgb(GetDensityKernelType()); }
void shqh_set_hydropar(int DensityIndependentSphOn, double DensityContrastLimit, double ArtBulkViscConst)
{
    struct hydro_params hp;
    hp.DensityIndependentSphOn = DensityIndependentSphOn;
    hp.DensityContrastLimit = DensityContrastLimit;
    hp.ArtBulkViscConst = ArtBulkViscConst;
    set_hydropar(hp);
}
int shqh_set_init_hsml(ForceTree *tree, double MeanGasSeparation, part_manager_type *pm) { return set_init_hsml(tree, MeanGasSeparation, pm); }
void shqh_force_tree_update_hmax(ForceTree *tree, part_manager_type *pm) { force_tree_update_hmax(tree, pm); }

/* evp_out: [nsph] receives *EntVarPred (copied; the malloc'ed array is freed here) */
int shqh_density(shq_context *ctx, part_manager_type *pm, ForceTree *tree, sph_particle_data *sph, int64_t nsph,
                 bh_density_slot *bh, int64_t nbh, const int *active, int64_t nactive, int update_hsml, int DoEgyDensity,
                 int BlackHoleOn, const shq_kick_factors *kick, double *evp_out, double *GradRho_mag, int UseGPU,
                 shq_sph_stats *stats)
{
    slots_manager_type S = {sph, nsph, bh, nbh};
    ActiveParticles act;
    memset(&act, 0, sizeof(act));
    act.ActiveParticle = (int *) active;
    act.NumActiveParticle = active ? nactive : pm->NumPart;
    MyFloat *evp = nullptr;
    int rc = density(ctx, &act, update_hsml, DoEgyDensity, BlackHoleOn, kick, &evp, GradRho_mag, tree, pm, &S, UseGPU != 0, stats);
    if(rc == 0 && evp) {
        if(evp_out)
            memcpy(evp_out, evp, sizeof(double) * (size_t) nsph);
        free(evp);
    }
    return rc;
}

int shqh_hydro_force(shq_context *ctx, part_manager_type *pm, ForceTree *tree, sph_particle_data *sph, int64_t nsph,
                     const int *active, int64_t nactive, double atime, double hubble, double *EntVarPred,
                     const shq_kick_factors *kick, const double *drifts, int UseGPU, shq_sph_stats *stats)
{
    slots_manager_type S = {sph, nsph, nullptr, 0};
    ActiveParticles act;
    memset(&act, 0, sizeof(act));
    act.ActiveParticle = (int *) active;
    act.NumActiveParticle = active ? nactive : pm->NumPart;
    return hydro_force(ctx, &act, atime, hubble, EntVarPred, kick, drifts, tree, pm, &S, UseGPU != 0, stats);
}
}

/* ---- time line test/driver API ------------------------------------------------------------------ */
#include "timestep.hpp"
extern "C" {

TimeBinMgr *shqh_timebinmgr_create(const double *sync_loga, int nsync) { return nsync >= 2 ? new TimeBinMgr(sync_loga, nsync) : nullptr; }
void shqh_timebinmgr_destroy(TimeBinMgr *t) { delete t; }
void shqh_timebinmgr_set_gravkick(TimeBinMgr *t, double (*cb)(inttime_t, inttime_t, void *), void *user)
{
    t->exact_gravkick = cb;
    t->user = user;
}
inttime_t shqh_tbm_ti_from_loga(const TimeBinMgr *t, double loga) { return t->ti_from_loga(loga); }
double shqh_tbm_loga_from_ti(const TimeBinMgr *t, inttime_t ti) { return t->loga_from_ti(ti); }
inttime_t shqh_tbm_dti_from_dloga(const TimeBinMgr *t, double dloga, inttime_t Ti) { return t->dti_from_dloga(dloga, Ti); }
double shqh_tbm_dloga_from_dti(const TimeBinMgr *t, inttime_t dti, inttime_t Ti) { return t->dloga_from_dti(dti, Ti); }
double shqh_tbm_get_dloga_for_bin(const TimeBinMgr *t, int bin, inttime_t Ti) { return t->get_dloga_for_bin(bin, Ti); }
inttime_t shqh_tbm_find_next_ti_sync(const TimeBinMgr *t, inttime_t ti) { return t->find_next_ti_sync(ti); }
void shqh_tbm_timeline_at(const TimeBinMgr *t, inttime_t Ti, shq_timeline *out) { *out = t->timeline_at(Ti); }
inttime_t shqh_round_down_power_of_two(inttime_t dti) { return round_down_power_of_two(dti); }
int shqh_get_timestep_bin(inttime_t dti) { return get_timestep_bin(dti); }
int shqh_is_timebin_active(int bin, inttime_t Ti) { return is_timebin_active(bin, Ti); }

void shqh_set_timestep_params(double ErrTolIntAccuracy, int ForceEqualTimesteps, double MinSizeTimestep, double MaxSizeTimestep,
                              double MaxRMSDisplacementFac, double MaxGasVel, double CourantFac)
{
    struct timestep_params p = {ErrTolIntAccuracy, ForceEqualTimesteps, MinSizeTimestep, MaxSizeTimestep, MaxRMSDisplacementFac, MaxGasVel, CourantFac};
    set_timestep_params(p);
}

/* the test driver's cosmology: the Hubble function is one number per call (the loops evaluate it at one atime only) */
struct shqh_cosmo {
    double OmegaBaryon, OmegaCDM, OmegaNu1, RhoCrit, Omega0, Hubble, GravInternal, hubble_now;
};
static double g_hubble_now;
static double hubble_const(const Cosmology *, double) { return g_hubble_now; }
static Cosmology make_cosmo(const shqh_cosmo *c)
{
    g_hubble_now = c->hubble_now;
    Cosmology CP = {c->OmegaBaryon, c->OmegaCDM, c->OmegaNu1, c->RhoCrit, c->Omega0, c->Hubble, c->GravInternal, hubble_const};
    return CP;
}
static ActiveParticles make_act(int have_list, int64_t nactive, int64_t nactivegrav)
{
    ActiveParticles act;
    memset(&act, 0, sizeof(act));
    static int marker;
    act.ActiveParticle = have_list ? &marker : nullptr; /* only tested against NULL: the list is the resident one */
    act.NumActiveParticle = nactive;
    act.NumActiveGravity = nactivegrav;
    return act;
}

int shqh_find_timesteps(shq_context *ctx, int have_list, int64_t nactive, DriftKickTimes *times, TimeBinMgr *tbm, double atime,
                        int FastParticleType, const shqh_cosmo *c, double asmth, int isFirstTimeStep, int *bad)
{
    Cosmology CP = make_cosmo(c);
    ActiveParticles act = make_act(have_list, nactive, nactive);
    return find_timesteps(ctx, &act, times, tbm, atime, FastParticleType, &CP, asmth, isFirstTimeStep, bad);
}

int shqh_find_hydro_timesteps(shq_context *ctx, int have_list, int64_t nactive, DriftKickTimes *times, TimeBinMgr *tbm, double atime,
                              const shqh_cosmo *c, int isFirstTimeStep, int *bad)
{
    Cosmology CP = make_cosmo(c);
    ActiveParticles act = make_act(have_list, nactive, nactive);
    return find_hydro_timesteps(ctx, &act, times, tbm, atime, &CP, isFirstTimeStep, bad);
}

int shqh_hierarchical_gravity_and_timesteps(shq_context *ctx, int have_list, int64_t nactive, int64_t nactivegrav, double BoxSize, double Asmth,
                                            int Nmesh, double G, int have_stored_accel, DriftKickTimes *times, TimeBinMgr *tbm, double atime,
                                            int treemask, int FastParticleType, const shqh_cosmo *c, int walk_mode, int64_t *bad)
{
    Cosmology CP = make_cosmo(c);
    ActiveParticles act = make_act(have_list, nactive, nactivegrav);
    PetaPM pm;
    gravpm_init_periodic(&pm, BoxSize, Asmth, Nmesh, G);
    return hierarchical_gravity_and_timesteps(ctx, &act, &pm, have_stored_accel, times, tbm, atime, treemask, FastParticleType, &CP, walk_mode, bad);
}
}
