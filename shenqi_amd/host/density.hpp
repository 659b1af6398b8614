/* density.hpp — host-side mirror of the reference SPH operator API for the force path:
 * density() / set_init_hsml() / GetNumNgb (libgadget/density2.h:16-50, density2.cpp:105-204) and
 * hydro_force() (libgadget/hydra2.h:9-22, hydra2.cpp:76-110).  Compute goes through the C-ABI;
 * there is no CPU fallback (UseGPU = false is an error). */
#ifndef SHQH_DENSITY_HPP
#define SHQH_DENSITY_HPP
#include "forcetree.hpp"

enum DensityKernelType {
    DENSITY_KERNEL_CUBIC_SPLINE = 1,
    DENSITY_KERNEL_QUINTIC_SPLINE = 2,
    DENSITY_KERNEL_QUARTIC_SPLINE = 4,
};

struct density_params {
    double DensityResolutionEta;
    double MaxNumNgbDeviation;
    enum DensityKernelType DensityKernelType;
    double BlackHoleNgbFactor;
    double BlackHoleMaxAccretionRadius;
    double MinGasHsmlFractional;
    double MinGasHsml;
};

struct hydro_params {
    int DensityIndependentSphOn;
    double DensityContrastLimit;
    double ArtBulkViscConst;
};

/* The slots the force path touches (libgadget/slotsmanager.h): SPH slots and the two BH fields
 * density() writes. */
struct bh_density_slot {
    double Density;
    double DivVel;
};
struct slots_manager_type {
    sph_particle_data *sph;
    int64_t nsph;
    bh_density_slot *bh;
    int64_t nbh;
};

void set_densitypar(struct density_params dp);
struct density_params get_densitypar(void);
double GetNumNgb(enum DensityKernelType KernelType);
enum DensityKernelType GetDensityKernelType(void);
void set_hydropar(struct hydro_params hp);
struct hydro_params get_hydropar(void);

/* density2.cpp:154-204.  Needs tree->Father and moments. Returns 0 or an error code. */
int set_init_hsml(ForceTree *tree, const double MeanGasSeparation, part_manager_type *PartManager);
/* End state of force_tree_calc_moments for hmax (forcetree.cpp:947-966,1080-1101): leaf hmax from
 * the current Hsml of the gas/BH particles it holds, maxima propagated to the ancestors. */
void force_tree_update_hmax(ForceTree *tree, const part_manager_type *PartManager);

/* *EntVarPred is allocated here (malloc) as the reference does with mymanagedmalloc and must be
 * freed by the caller after hydro_force (slots_free_sph_pred_data). */
int density(shq_context *ctx, const ActiveParticles *act, int update_hsml, int DoEgyDensity, int BlackHoleOn,
            const shq_kick_factors *kick, MyFloat **EntVarPred, MyFloat *GradRho_mag, ForceTree *tree,
            part_manager_type *PartManager, slots_manager_type *SlotsManager, bool UseGPU, shq_sph_stats *stats);
int hydro_force(shq_context *ctx, const ActiveParticles *act, const double atime, const double hubble, MyFloat *EntVarPred,
                const shq_kick_factors *kick, const double *drifts, const ForceTree *tree, part_manager_type *PartManager,
                slots_manager_type *SlotsManager, bool UseGPU, shq_sph_stats *stats);
#endif
