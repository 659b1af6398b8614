/* gravity.hpp — host-side mirror of the reference gravity operator API for the force path:
 * gravshort_tree_params / GravShortTable / grav_short_tree (libgadget/gravity.h:13-61,89;
 * gravshort-tree2.cpp:29-172) and PetaPM / gravpm_force (libgadget/petapm.h:87-112,
 * gravpm.cpp:51-119).  Same names and argument meaning; the compute goes through the C-ABI
 * (include/shenqi_hip.h).  There is no CPU fallback here: UseGPU = false is an error. */
#ifndef SHQH_GRAVITY_HPP
#define SHQH_GRAVITY_HPP
#include "forcetree.hpp"

enum ShortRangeForceWindowType {
    SHORTRANGE_FORCE_WINDOW_TYPE_EXACT = 1,
    SHORTRANGE_FORCE_WINDOW_TYPE_ERFC = 2,
};

struct gravshort_tree_params {
    double ErrTolForceAcc;
    double BHOpeningAngle;
    double MaxBHOpeningAngle;
    int TreeUseBH;
    double Rcut;
    double FractionalGravitySoftening;
    size_t MaxExportBufferBytes;
    enum ShortRangeForceWindowType ShortRangeForceWindowType;
};

/* The fields of PetaPM the force path reads (petapm.h:87-112). */
struct PetaPM {
    double Asmth;
    double BoxSize;
    double CellSize;
    int Nmesh;
    double G;
};

#define NGRAVTAB SHQ_NGRAVTAB
class GravShortTable {
  public:
    float shortrange_table[NGRAVTAB];
    float shortrange_table_potential[NGRAVTAB];
    double dx;
    /* kernels: the 512x5 calibrated table of libgadget/shortrange-kernel.c
     * (x, w_pot, w_force, w_pot_erf, w_force_erf), shipped as shenqi_amd/data/shortrange_force_kernels.f64 */
    GravShortTable(const enum ShortRangeForceWindowType type, const double Asmth, const double (*kernels)[5]);
    int status; /* 0 ok, else an error (Asmth != 1.5 with the exact window, gravshort-tree2.cpp:42-46) */
};

void set_gravshort_treepar(struct gravshort_tree_params tree_params);
struct gravshort_tree_params get_gravshort_treepar(void);
void gravshort_set_softenings(double MeanSeparation);
double FORCE_SOFTENING(void);
int gravshort_set_kernel_table(const double (*kernels)[5]); /* where the exact window comes from */

void gravpm_init_periodic(PetaPM *pm, double BoxSize, double Asmth, int Nmesh, double G);

/* Both return 0 on success; otherwise the shim's endrun() equivalent: the error text is in
 * shq_last_error().  ctx is the library context of this rank (one GPU per rank). */
int grav_short_tree(shq_context *ctx, const ActiveParticles *act, PetaPM *pm, ForceTree *tree,
                    part_manager_type *PartManager, MyFloat (*AccelStore)[3], double rho0,
                    inttime_t Ti_Current, bool UseGPU, int walk_mode, shq_walk_stats *stats);
int gravpm_force(shq_context *ctx, PetaPM *pm, part_manager_type *PartManager, bool UseGPU);

/* fill the POD the C-ABI takes from the module state (GravTreeParams ctor, gravshort2.hpp:45-54) */
int make_grav_params(const PetaPM *pm, double BoxSize, double rho0, shq_grav_params *out);
#endif
