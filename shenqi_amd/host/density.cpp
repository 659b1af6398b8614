/* density.cpp — see density.hpp. */
#include "density.hpp"
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include <stdio.h>

extern "C" void shqh_set_error(const char *msg);

static struct density_params DensityParams;
static struct hydro_params HydroParams;

void set_densitypar(struct density_params dp) { DensityParams = dp; }
struct density_params get_densitypar(void) { return DensityParams; }
void set_hydropar(struct hydro_params hp) { HydroParams = hp; }
struct hydro_params get_hydropar(void) { return HydroParams; }
enum DensityKernelType GetDensityKernelType(void) { return DensityParams.DensityKernelType; }

/* density2.cpp:55-67 with DensityKrnl::desnumngb (densitykernel.hpp:36-41) */
double GetNumNgb(enum DensityKernelType KernelType)
{
    const double support = (KernelType == DENSITY_KERNEL_CUBIC_SPLINE) ? 4 : ((KernelType == DENSITY_KERNEL_QUARTIC_SPLINE) ? 5 : 6);
    return (4.0 / 3 * M_PI) * pow(support / 2. * DensityParams.DensityResolutionEta, 3);
}

int set_init_hsml(ForceTree *tree, const double MeanGasSeparation, part_manager_type *PartManager)
{
    if(!tree->Father) {
        shqh_set_error("tree Father array not allocated at initial hsml!");
        return 5;
    }
    const double DesNumNgb = GetNumNgb(GetDensityKernelType());
    particle_data *parts = PartManager->Base;
#pragma omp parallel for
    for(int64_t i = 0; i < PartManager->NumPart; i++) {
        if(parts[i].Type != 0 && parts[i].Type != 5)
            continue;
        if(parts[i].IsGarbage)
            continue;
        int64_t no = i;
        do {
            const int64_t p = (no >= tree->firstnode) ? tree->Nodes[no].father : tree->Father[no];
            if(p < tree->firstnode)
                break;
            no = p;
        } while(10 * DesNumNgb * parts[i].Mass > tree->Nodes[no].mass);
        parts[i].Hsml = MeanGasSeparation;
        if(no >= tree->firstnode) {
            const double testhsml = tree->Nodes[no].len * pow(3.0 / (4 * M_PI) * DesNumNgb * parts[i].Mass / tree->Nodes[no].mass, 1.0 / 3);
            if(testhsml < 500. * MeanGasSeparation)
                parts[i].Hsml = testhsml;
        }
    }
    return 0;
}

void force_tree_update_hmax(ForceTree *tree, const part_manager_type *PartManager)
{
    const particle_data *P = PartManager->Base;
    NODE *base = tree->Nodes_base;
    for(int64_t k = 0; k < tree->numnodes; k++)
        base[k].hmax = 0;
    for(int64_t k = 0; k < tree->numnodes; k++) {
        NODE *nd = &base[k];
        if(SHQ_NODE_CHILDTYPE(nd->flags) != SHQ_PARTICLE_NODE_TYPE)
            continue;
        for(int c = 0; c < nd->noccupied; c++) {
            const particle_data &pp = P[nd->suns[c]];
            if(pp.Type != 0 && pp.Type != 5)
                continue;
            for(int j = 0; j < 3; j++) {
                const double v = fabs(pp.Pos[j] - nd->center[j]) + pp.Hsml - nd->len / 2.;
                if(v > nd->hmax)
                    nd->hmax = v;
            }
        }
        const double h = nd->hmax;
        int64_t f = nd->father;
        while(f >= tree->firstnode && tree->Nodes[f].hmax < h) {
            tree->Nodes[f].hmax = h;
            f = tree->Nodes[f].father;
        }
    }
    tree->hmax_computed_flag = 1;
}

static shq_sph_view make_sph_view(slots_manager_type *S)
{
    shq_sph_view v;
    v.base = S->sph;
    v.elsize = sizeof(sph_particle_data);
    v.numslots = S->nsph;
    v.off_density = offsetof(sph_particle_data, Density);
    v.off_egywtdensity = offsetof(sph_particle_data, EgyWtDensity);
    v.off_entropy = offsetof(sph_particle_data, Entropy);
    v.off_dtentropy = offsetof(sph_particle_data, DtEntropy);
    v.off_maxsignalvel = offsetof(sph_particle_data, MaxSignalVel);
    v.off_hydroaccel = offsetof(sph_particle_data, HydroAccel);
    v.off_dhsmlegydensityfactor = offsetof(sph_particle_data, DhsmlEgyDensityFactor);
    v.off_divvel = offsetof(sph_particle_data, DivVel);
    v.off_curlvel = offsetof(sph_particle_data, CurlVel);
    v.off_delaytime = offsetof(sph_particle_data, DelayTime);
    return v;
}

int density(shq_context *ctx, const ActiveParticles *act, int update_hsml, int DoEgyDensity, int BlackHoleOn,
            const shq_kick_factors *kick, MyFloat **EntVarPred, MyFloat *GradRho_mag, ForceTree *tree,
            part_manager_type *PartManager, slots_manager_type *SlotsManager, bool UseGPU, shq_sph_stats *stats)
{
    if(!UseGPU) {
        shqh_set_error("density: this build has no CPU neighbour walk; UseGPU must be true");
        return 1;
    }
    if(!force_tree_allocated(tree)) {
        shqh_set_error("Tree has been freed before this treewalk.");
        return 1;
    }
    if(!(tree->mask & GASMASK)) { /* LocalNgbTreeWalk::validate_tree, localtreewalk2.h:350-366 */
        shqh_set_error("Treewalk for DENSITY needs gas particles in the tree");
        return 5;
    }
    shq_density_params dp;
    memset(&dp, 0, sizeof(dp));
    dp.BoxSize = tree->BoxSize;
    dp.DesNumNgb = GetNumNgb(DensityParams.DensityKernelType);
    dp.DesNumNgbBH = dp.DesNumNgb * DensityParams.BlackHoleNgbFactor;
    dp.MinGasHsml = DensityParams.MinGasHsml;
    dp.MaxNumNgbDeviation = DensityParams.MaxNumNgbDeviation;
    dp.update_hsml = update_hsml;
    dp.BlackHoleOn = BlackHoleOn;
    dp.DoEgyDensity = DoEgyDensity;
    dp.WindsDecouple = 0; /* winds_ever_decouple(): the wind module is outside the force path */
    dp.DensityKernelType = DensityParams.DensityKernelType;
    if(kick)
        dp.kf = *kick;
    shq_part_view pv = make_part_view(PartManager->Base, PartManager->NumPart);
    shq_sph_view sv = make_sph_view(SlotsManager);
    shq_bh_view bv;
    bv.base = SlotsManager->bh;
    bv.elsize = sizeof(bh_density_slot);
    bv.numslots = SlotsManager->nbh;
    bv.off_density = offsetof(bh_density_slot, Density);
    bv.off_divvel = offsetof(bh_density_slot, DivVel);
    shq_tree_view tv = force_tree_view(tree);
    MyFloat *evp = (MyFloat *) calloc((size_t) (SlotsManager->nsph > 0 ? SlotsManager->nsph : 1), sizeof(MyFloat));
    const int32_t *active = (act && act->ActiveParticle) ? act->ActiveParticle : nullptr;
    const int64_t nactive = active ? act->NumActiveParticle : PartManager->NumPart;
    int rc = shq_density(ctx, &tv, tree->Nodes_base, &pv, &sv, SlotsManager->bh ? &bv : nullptr, active, nactive, &dp, evp,
                         GradRho_mag, stats);
    if(rc != SHQ_OK) {
        free(evp);
        shqh_set_error(shq_last_error());
        return rc;
    }
    if(EntVarPred)
        *EntVarPred = evp; /* density2.cpp:147 */
    else
        free(evp);
    return 0;
}

int hydro_force(shq_context *ctx, const ActiveParticles *act, const double atime, const double hubble, MyFloat *EntVarPred,
                const shq_kick_factors *kick, const double *drifts, const ForceTree *tree, part_manager_type *PartManager,
                slots_manager_type *SlotsManager, bool UseGPU, shq_sph_stats *stats)
{
    if(!UseGPU) {
        shqh_set_error("hydro_force: this build has no CPU neighbour walk; UseGPU must be true");
        return 1;
    }
    if(!tree->hmax_computed_flag) { /* hydra2.cpp:79-80 */
        shqh_set_error("Hydro called before hmax computed");
        return 5;
    }
    const double GAMMA = 5.0 / 3.0;
    shq_hydro_params hp;
    memset(&hp, 0, sizeof(hp));
    hp.BoxSize = tree->BoxSize;
    hp.atime = atime;
    hp.fac_mu = pow(atime, 3 * (GAMMA - 1) / 2) / atime; /* HydroPriv ctor, hydratree2.hpp:85-88 */
    hp.fac_vsic_fix = hubble * pow(atime, 3 * (GAMMA - 1));
    hp.hubble_a2 = hubble * atime * atime;
    if(drifts)
        memcpy(hp.drifts, drifts, sizeof(hp.drifts));
    hp.ArtBulkViscConst = HydroParams.ArtBulkViscConst;
    hp.DensityContrastLimit = HydroParams.DensityContrastLimit;
    hp.DensityIndependentSphOn = HydroParams.DensityIndependentSphOn;
    hp.DensityKernelType = DensityParams.DensityKernelType;
    hp.WindSpeed = 0;
    hp.WindFreeTravelDensThresh = 0;
    if(kick)
        hp.kf = *kick;
    shq_part_view pv = make_part_view(PartManager->Base, PartManager->NumPart);
    shq_sph_view sv = make_sph_view(SlotsManager);
    shq_tree_view tv = force_tree_view(tree);
    const int32_t *active = (act && act->ActiveParticle) ? act->ActiveParticle : nullptr;
    const int64_t nactive = active ? act->NumActiveParticle : PartManager->NumPart;
    int rc = shq_hydro_force(ctx, &tv, &pv, &sv, active, nactive, &hp, EntVarPred, stats);
    if(rc != SHQ_OK) {
        shqh_set_error(shq_last_error());
        return rc;
    }
    return 0;
}
