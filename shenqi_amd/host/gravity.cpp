/* gravity.cpp — see gravity.hpp. */
#include "gravity.hpp"
#include <math.h>
#include <string.h>
#include <stdio.h>
#include <vector>

extern "C" void shqh_set_error(const char *msg);

static struct gravshort_tree_params TreeParams;
static double GravitySoftening;
static const double (*KernelTable)[5] = nullptr;

GravShortTable::GravShortTable(const enum ShortRangeForceWindowType type, const double Asmth, const double (*kernels)[5])
{
    status = 0;
    memset(shortrange_table, 0, sizeof(shortrange_table));
    memset(shortrange_table_potential, 0, sizeof(shortrange_table_potential));
    dx = 0;
    if(!kernels) {
        status = 1;
        return;
    }
    if(type == SHORTRANGE_FORCE_WINDOW_TYPE_EXACT && Asmth != 1.5) {
        status = 2; /* "calibrated for Asmth = 1.5", gravshort-tree2.cpp:42-46 */
        return;
    }
    dx = kernels[1][0];
    for(size_t i = 0; i < NGRAVTAB; i++) {
        const double u = kernels[i][0] * 0.5 / Asmth;
        switch(type) {
        case SHORTRANGE_FORCE_WINDOW_TYPE_EXACT:
            shortrange_table[i] = kernels[i][2];
            shortrange_table_potential[i] = kernels[i][1];
            break;
        case SHORTRANGE_FORCE_WINDOW_TYPE_ERFC:
            shortrange_table[i] = erfc(u) + 2.0 * u / sqrt(M_PI) * exp(-u * u);
            shortrange_table_potential[i] = erfc(u);
            break;
        }
    }
}

void set_gravshort_treepar(struct gravshort_tree_params tree_params) { TreeParams = tree_params; }
struct gravshort_tree_params get_gravshort_treepar(void) { return TreeParams; }
void gravshort_set_softenings(double MeanSeparation) { GravitySoftening = TreeParams.FractionalGravitySoftening * MeanSeparation; }
double FORCE_SOFTENING(void) { return 2.8 * GravitySoftening; }
int gravshort_set_kernel_table(const double (*kernels)[5])
{
    KernelTable = kernels;
    return 0;
}

void gravpm_init_periodic(PetaPM *pm, double BoxSize, double Asmth, int Nmesh, double G)
{
    /* petapm_init, petapm.cpp:203-216 */
    pm->BoxSize = BoxSize;
    pm->Asmth = Asmth;
    pm->Nmesh = Nmesh;
    pm->G = G;
    pm->CellSize = BoxSize / Nmesh;
}

int make_grav_params(const PetaPM *pm, double BoxSize, double rho0, shq_grav_params *out)
{
    GravShortTable gravtab(TreeParams.ShortRangeForceWindowType, pm->Asmth, KernelTable);
    if(gravtab.status == 1) {
        shqh_set_error("GravShortTable: no kernel table set (gravshort_set_kernel_table)");
        return 1;
    }
    if(gravtab.status == 2) {
        char buf[200];
        snprintf(buf, sizeof(buf), "The short range force window is calibrated for Asmth = 1.5, but running with %g", pm->Asmth);
        shqh_set_error(buf);
        return 1;
    }
    memset(out, 0, sizeof(*out));
    out->BoxSize = BoxSize;
    out->cellsize = BoxSize / pm->Nmesh;
    out->Rcut = TreeParams.Rcut * pm->Asmth * out->cellsize;
    out->G = pm->G;
    out->cbrtrho0 = pow(rho0, 1.0 / 3);
    out->ForceSoftening = FORCE_SOFTENING();
    out->ErrTolForceAcc = TreeParams.ErrTolForceAcc;
    out->BHOpeningAngle2 = TreeParams.BHOpeningAngle * TreeParams.BHOpeningAngle;
    out->TreeUseBH = TreeParams.TreeUseBH;
    if(out->TreeUseBH == 0)
        out->BHOpeningAngle2 = TreeParams.MaxBHOpeningAngle * TreeParams.MaxBHOpeningAngle;
    memcpy(out->shortrange_table, gravtab.shortrange_table, sizeof(out->shortrange_table));
    memcpy(out->shortrange_table_potential, gravtab.shortrange_table_potential, sizeof(out->shortrange_table_potential));
    out->dx = gravtab.dx;
    return 0;
}

int grav_short_tree(shq_context *ctx, const ActiveParticles *act, PetaPM *pm, ForceTree *tree,
                    part_manager_type *PartManager, MyFloat (*AccelStore)[3], double rho0,
                    inttime_t Ti_Current, bool UseGPU, int walk_mode, shq_walk_stats *stats)
{
    (void) Ti_Current;
    if(!UseGPU) {
        shqh_set_error("grav_short_tree: this build has no CPU tree walk; UseGPU must be true");
        return 1;
    }
    /* GravLocalTreeWalk::validate_tree, gravshort2.hpp:203-214 */
    if(!force_tree_allocated(tree)) {
        shqh_set_error("Tree has been freed before this treewalk.");
        return 1;
    }
    const int need = GASMASK + DMMASK + STARMASK + BHMASK;
    if((tree->mask & need) != need) {
        shqh_set_error("Gravity treewalk needs all particle types but tree mask is wrong");
        return 5;
    }
    if(!tree->moments_computed_flag) {
        shqh_set_error("Gravtree called before tree moments computed!");
        return 2;
    }
    shq_grav_params gp;
    if(make_grav_params(pm, tree->BoxSize, rho0, &gp))
        return 1;
    shq_part_view pv = make_part_view(PartManager->Base, PartManager->NumPart);
    shq_tree_view tv = force_tree_view(tree);
    std::vector<double> own;
    MyFloat (*Accel)[3] = AccelStore;
    if(!Accel) { /* GravTreeOutput ctor, gravshort2.hpp:66-73 */
        own.resize(3 * (size_t) (PartManager->NumPart > 0 ? PartManager->NumPart : 1));
        Accel = (MyFloat (*)[3]) own.data();
    }
    const int32_t *active = (act && act->ActiveParticle) ? act->ActiveParticle : nullptr;
    const int64_t nactive = active ? act->NumActiveParticle : PartManager->NumPart;
    int rc = shq_grav_short_tree(ctx, &tv, &pv, active, nactive, &gp, Accel, tree->full_particle_tree_flag, walk_mode, stats);
    if(rc != SHQ_OK) {
        shqh_set_error(shq_last_error());
        return rc;
    }
    /* gravshort-tree2.cpp:168-171 */
    if(TreeParams.TreeUseBH > 1)
        TreeParams.TreeUseBH = 0;
    return 0;
}

int gravpm_force(shq_context *ctx, PetaPM *pm, part_manager_type *PartManager, bool UseGPU)
{
    if(!UseGPU) {
        shqh_set_error("gravpm_force: this build has no CPU PM; UseGPU must be true");
        return 1;
    }
    particle_data *P = PartManager->Base;
    const int64_t n = PartManager->NumPart;
    shq_pm_params pp;
    memset(&pp, 0, sizeof(pp));
    pp.Nmesh = pm->Nmesh;
    pp.BoxSize = pm->BoxSize;
    pp.Asmth = pm->Asmth;
    pp.G = pm->G;
    shq_part_view pv = make_part_view(P, n);
    std::vector<double> g(3 * (size_t) (n > 0 ? n : 1)), pot((size_t) (n > 0 ? n : 1), 0.0);
    int rc = shq_pm_force(ctx, &pp, &pv, (double (*)[3]) g.data(), pot.data());
    if(rc != SHQ_OK) {
        shqh_set_error(shq_last_error());
        return rc;
    }
    for(int64_t i = 0; i < n; i++) {
        /* gravpm.cpp:88-92 zero + :489-500 readout */
        P[i].GravPM[0] = g[3 * i];
        P[i].GravPM[1] = g[3 * i + 1];
        P[i].GravPM[2] = g[3 * i + 2];
        P[i].Potential += pot[i];
    }
    return 0;
}
