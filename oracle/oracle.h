/* oracle.h — CPU restatement of the reference algorithms on the TreePM+SPH force path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under shenqi_amd/ may include, link or call this.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * Every function cites the reference file:line (relative to /root/reference) it restates.
 * Pinning: see oracle/README.md (reference golden values of tests/test_densitykernel.cpp, the
 * reference's own accuracy gates of tests/test_gravity.cpp / tests/test_density.cpp /
 * runtests.cpp, and the short-range table compiled from the reference's own data file into
 * oracle/_ref/).
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stdint.h>
#include <stddef.h>
#include "../include/shenqi_hip.h" /* POD layouts only (shq_node, shq_grav_params, ...) */

#ifdef __cplusplus
extern "C" {
#endif

/* libgadget/partmanager.h:99 */
static inline double orc_nearest(double x, double BoxSize)
{
    return (x > 0.5 * BoxSize) ? (x - BoxSize) : ((x < -0.5 * BoxSize) ? (x + BoxSize) : x);
}

/* ---- tree (libgadget/forcetree.cpp) ---- */
/* Build the oct-tree by sequential insertion of particles idx[0..n) (NULL => 0..n-1), then
 * compute moments, sibling threading and remove empty leaves.  nodes must hold maxnodes
 * entries.  Node index space: firstnode == numpart_total.  hsml may be NULL (hmax = 0).
 * Returns number of nodes used, or -1 if maxnodes is too small. */
int64_t orc_tree_build(const double *pos, const float *mass, const double *hsml,
                       const int32_t *idx, int64_t n, int64_t numpart_total, double BoxSize,
                       shq_node *nodes, int64_t maxnodes, int32_t *father);

/* ---- short-range gravity (libgadget/gravshort2.hpp) ---- */
/* Per-target stackless primary walk; acc_out[n][3], pot_out[n] raw (before postprocess),
 * nint_out[n].  oldacc = |FullTreeGravAccel+GravPM|/G per particle. targets NULL => all. */
void orc_grav_walk_secondary(const shq_node *nodes, int64_t firstnode, const double *pos, const float *mass,
                             const double *qpos, const int32_t *qnodelist, const double *qoldacc, int64_t nq,
                             const shq_grav_params *p, double *acc_out, double *pot_out, int64_t *nint_out);
/* Top-tree walks (toptree.cpp): GravTopTreeWalk::toptree_visit (gravshort2.hpp:362-438) and
 * TopTreeWalk::toptree_visit (localtreewalk2.h:210-259) with export_particle (:269-312), one target after the other.
 * counts[t] = exports of target t; table (capacity entries) receives them in target order; returns the total
 * (table may be NULL to count only: export_count, :315-324). */
int64_t orc_grav_toptree(const shq_node *nodes, int64_t firstnode, int64_t lastnode, const shq_topleaf *topleaves,
                         const double *pos, const double *oldacc, const int32_t *targets, int64_t ntargets,
                         const shq_grav_params *p, int32_t *counts, shq_data_index *table, int64_t capacity);
int64_t orc_ngb_toptree(const shq_node *nodes, int64_t firstnode, int64_t lastnode, const shq_topleaf *topleaves,
                        const double *pos, const double *hsml, int symmetric, double BoxSize, const int32_t *targets,
                        int64_t ntargets, int32_t *counts, shq_data_index *table, int64_t capacity);
void orc_grav_walk(const shq_node *nodes, int64_t firstnode, const double *pos,
                   const float *mass, const double *oldacc, const int32_t *targets,
                   int64_t ntargets, const shq_grav_params *p, double *acc_out,
                   double *pot_out, int64_t *nint_out);
/* GravTreeOutput::postprocess (gravshort2.hpp:88-107): in place on acc/pot for target list. */
void orc_grav_postprocess(const float *mass, const int32_t *targets, int64_t ntargets,
                          const shq_grav_params *p, int update_potential, double *acc,
                          double *pot);
/* Single interaction (apply_accn, gravshort2.hpp:326-358); returns 1 if applied. */
int orc_apply_accn(const double dx[3], double r2, double mass, const shq_grav_params *p,
                   double acc[3], double *pot);
/* Direct summation with +-repeat periodic images and the spline softening, as
 * tests/test_gravity.cpp:41-76,121-143 (force_direct / grav_force). */
void orc_force_direct(const double *pos, const float *mass, int64_t n, double BoxSize, double G,
                      double h, int repeat, double *accn);

/* ---- PM (libgadget/petapm.cpp, gravpm.cpp) ---- */
/* Full PM force: CIC deposit, r2c, potential_transfer, 4x (transfer + c2r), CIC readout.
 * fixed_point_log2scale < 0: plain f64 deposit; >= 0: deposit rounds each contribution to a
 * multiple of 2^-scale (the device's order-independent integer deposit). mesh_rho / mesh_pot
 * (Nmesh^3, may be NULL) receive the deposited mass mesh and the potential mesh.
 * use_stencil: 0 = four k-space transfers + c2r as the reference; 1 = one c2r + real-space
 * 4-point differencing (mathematically identical symbol). */
void orc_pm_force(const double *pos, const float *mass, const uint8_t *skip, int64_t n,
                  const shq_pm_params *pm, int fixed_point_log2scale, int use_stencil,
                  double *gravpm, double *potential, double *mesh_rho, double *mesh_pot);
/* Unscaled 3-D r2c / c2r on an N^3 mesh, layouts [x][y][z] <-> [x][y][z'<=N/2] complex. */
void orc_fft_r2c(int N, const double *real, double *complx);
/* install other transforms for orc_pm_force (NULL, NULL: the oracle's own again); same conventions as orc_fft_r2c / orc_fft_c2r */
void orc_set_fft(void (*r2c)(int, const double *, double *), void (*c2r)(int, const double *, double *));
void orc_fft_c2r(int N, const double *complx, double *real);

/* ---- SPH (libgadget/densitytree2.hpp, hydratree2.hpp, density2.h, densitykernel.hpp) ---- */
/* Plain SoA inputs/outputs; slot arrays are indexed by pi[i] (the reference's PI). */
typedef struct orc_sph_arrays {
    int64_t n;     /* particles */
    int64_t nsph;  /* gas slots */
    const double *pos;      /* [n][3] */
    const float *mass;      /* [n] */
    const uint8_t *type;    /* [n] */
    const uint8_t *flags;   /* [n] bit0 garbage, bit1 swallowed; may be NULL */
    const int32_t *pi;      /* [n] */
    double *hsml;           /* [n] in/out */
    double *dthsml;         /* [n] out */
    const double *vel;      /* [n][3] */
    const double *treeacc;  /* [n][3] FullTreeGravAccel */
    const double *gravpm;   /* [n][3] */
    const uint8_t *bin_grav;  /* [n] */
    const uint8_t *bin_hydro; /* [n] */
    double *density, *egywtdensity;   /* [nsph] */
    const double *entropy;            /* [nsph] */
    double *dtentropy, *maxsignalvel; /* [nsph] */
    double *hydroaccel;               /* [nsph][3] */
    double *dhsmlegydensityfactor, *divvel, *curlvel; /* [nsph] */
    const double *delaytime;          /* [nsph], may be NULL */
    double *bh_density, *bh_divvel;   /* BH slots, may be NULL when there are no BHs */
} orc_sph_arrays;

int orc_density_kernel(int type, double H, double u, double eta, double out[5]);
void orc_set_init_hsml(const shq_node *nodes, int64_t firstnode, const int32_t *father, orc_sph_arrays *a,
                       double MeanGasSeparation, double DesNumNgb);
/* Returns 0, or 1 if MAXITER was exceeded. EntVarPred: [nsph] out (NULL => no cache, values
 * computed per neighbour as the reference does for few active particles). */
int orc_density(shq_node *nodes, int64_t firstnode, const int32_t *father, orc_sph_arrays *a,
                const int32_t *active, int64_t nactive, const shq_density_params *p, double *EntVarPred,
                double *GradRho, int *niter_out, int64_t *nint_out);
void orc_update_hmax(shq_node *nodes, int64_t firstnode, int64_t numnodes, const orc_sph_arrays *a);
void orc_hydro(const shq_node *nodes, int64_t firstnode, orc_sph_arrays *a, const int32_t *active,
               int64_t nactive, const shq_hydro_params *p, const double *EntVarPred, int64_t *nint_out);

int orc_num_threads(void);

/* stellar_density() (stellar_density2.cpp:306-341) with its ten-radius ngbiter, postprocess and ngb_narrow_down: the
 * SPH volume weights of the star particles in `queue` over the gas neighbours; updates a->hsml of the stars.
 * StarVolumeSPH is indexed by particle.  No reference fixture exists for it ("parity unpinned": restated line by line
 * and exercised against brute-force sums in tests/test_oracle_cpu.py). */
int orc_stellar_density(const shq_node *nodes, int64_t firstnode, orc_sph_arrays *a, const int32_t *queue, int64_t nqueue,
                        double BoxSize, double DesNumNgb, double MaxNgbDeviation, int SPHWeighting, int ktype,
                        double *StarVolumeSPH, int *niter_out, int64_t *nint_out);

/* blackhole_veldisp() (veldisp2.cpp:164-199): out[q][5] = NumDM, V1sumDM[3], V2sumDM of the q-th black hole in `queue` over the
 * dark-matter tree; vdisp[q] set where the reference sets BHP().VDisp.  "Parity unpinned" (no reference fixture). */
void orc_bh_veldisp(const shq_node *nodes, int64_t firstnode, const orc_sph_arrays *a, const int32_t *queue, int64_t nqueue,
                    double BoxSize, const shq_kick_factors *kf, double *out, double *vdisp);

/* winds_find_vel_disp(), wind part (veldisp2.cpp:203-528): vdisp[q] / dmradius[q] of the q-th gas particle of `queue`.
 * "Parity unpinned" (no reference fixture). */
int orc_wind_veldisp(const shq_node *nodes, int64_t firstnode, const orc_sph_arrays *a, const int32_t *queue, int64_t nqueue,
                     double BoxSize, const shq_kick_factors *kf, double Time, double hubble, double *vdisp, double *dmradius,
                     int *niter_out);

/* blackhole_minpot / blackhole_dynfric treewalks (bhdynfric.cpp:44-295): raw results, 12 doubles per black hole of `queue`
 * (MinPot, MinPotPos[3], MinPotVel[3], SurroundingDensity, SurroundingVel[3], SurroundingRmsVel).  "Parity unpinned". */
void orc_bh_dynfric(const shq_node *nodes, int64_t firstnode, const orc_sph_arrays *a, const double *potential, const int32_t *queue,
                    int64_t nqueue, double BoxSize, const shq_kick_factors *kf, int method, int ktype, int typemask, double *out);

#ifdef __cplusplus
}
#endif
#endif
