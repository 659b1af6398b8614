/* timestep.hpp — host-side mirror of the reference's time-line operators over the resident particle set:
 * TimeBinMgr conversions (libgadget/timebinmgr.h:48-260), DriftKickTimes (timestep.h:10-26), TimestepParams
 * (timestep.cpp:35-50), find_timesteps / find_hydro_timesteps / hierarchical_gravity_and_timesteps (timestep.cpp:307-822).
 * Same names and argument meaning; the particle loops run on the device through include/shenqi_hip.h (shq_find_timesteps,
 * shq_hier_*), this side keeps what the reference keeps per rank: the sync points, DriftKickTimes and the few scalars the
 * loops reduce to.  `act` stands for the RESIDENT active list (shq_build_active_particles); its counts are read, its
 * pointer is not.  One rank: where the reference reduces over MPI_COMM_WORLD the caller of a multi-rank run reduces the
 * result fields itself (dist.py does) before the DriftKickTimes update. */
#ifndef SHQH_TIMESTEP_HPP
#define SHQH_TIMESTEP_HPP
#include "gravity.hpp"
#include <vector>

#define TIMEBINS SHQ_TIMEBINS
#define TIMEBASE (1Lu << TIMEBINS)

static inline inttime_t dti_from_timebin(int bin) { return bin > 0 ? (inttime_t) (1Lu << (uint64_t) bin) : 0; }
int is_timebin_active(int i, inttime_t current);
inttime_t round_down_power_of_two(inttime_t dti);
int get_timestep_bin(inttime_t dti);
inttime_t find_next_kick(inttime_t Ti_Current, int minTimeBin);

/* The sync-point table and its conversions.  Kick factors need the cosmology's Hubble function: the caller supplies
 * 1 / (H(a) a^2) integrated in log a between two integer times (get_exact_gravkick_factor, timebinmgr.h:197-206). */
class TimeBinMgr {
  public:
    std::vector<double> loga; /* SyncPoints[i].loga */
    double (*exact_gravkick)(inttime_t ti0, inttime_t ti1, void *user) = nullptr;
    void *user = nullptr;
    TimeBinMgr(const double *sync_loga, int nsync) : loga(sync_loga, sync_loga + nsync) {}
    int64_t NSyncPoints() const { return (int64_t) loga.size(); }
    double Dloga_interval_ti(inttime_t ti) const;
    double loga_from_ti(inttime_t ti) const;
    inttime_t ti_from_loga(double la) const;
    inttime_t ti_from_loga_snap(double la, inttime_t lastsnap) const;
    inttime_t dti_from_dloga(double dloga, inttime_t Ti_Current) const;
    double dloga_from_dti(inttime_t dti, inttime_t Ti_Current) const;
    double get_dloga_for_bin(int timebin, inttime_t Ti_Current) const;
    inttime_t find_next_ti_sync(inttime_t ti) const;
    double get_exact_gravkick_factor(inttime_t ti0, inttime_t ti1) const { return exact_gravkick(ti0, ti1, user); }
    /* the part of the table the device loops read at Ti_Current */
    shq_timeline timeline_at(inttime_t Ti_Current) const;
};

typedef struct {
    int mintimebin;
    int maxtimebin;
    int mingravtimebin;
    inttime_t Ti_kick[TIMEBINS + 1];
    inttime_t Ti_lastactivedrift[TIMEBINS + 1];
    inttime_t Ti_Current;
    inttime_t PM_length;
    inttime_t PM_start;
    inttime_t PM_kick;
} DriftKickTimes;

struct timestep_params {
    double ErrTolIntAccuracy;
    int ForceEqualTimesteps;
    double MinSizeTimestep, MaxSizeTimestep;
    double MaxRMSDisplacementFac;
    double MaxGasVel;
    double CourantFac;
};
void set_timestep_params(struct timestep_params p);
struct timestep_params get_timestep_params(void);

/* what the loops read from Cosmology (cosmology.h): Omega of baryons / CDM / one neutrino species for the long-range
 * criterion, RhoCrit, Omega0 / Hubble / GravInternal for rho0, and the Hubble function */
struct Cosmology {
    double OmegaBaryon, OmegaCDM, OmegaNu1, RhoCrit, Omega0, Hubble, GravInternal;
    double (*hubble_function)(const Cosmology *CP, double atime);
};

int is_PM_timestep(const DriftKickTimes *times);
/* get_long_range_timestep_dloga / get_PM_timestep_ti (timestep.cpp:1141-1233); moments from shq_velocity_moments */
int get_long_range_timestep_dloga(shq_context *ctx, double atime, const Cosmology *CP, int FastParticleType, double asmth, double *dloga);
int get_PM_timestep_ti(shq_context *ctx, const DriftKickTimes *times, const TimeBinMgr *timebinmgr, double atime, const Cosmology *CP,
                       int FastParticleType, double asmth, inttime_t *dti);

/* return 0 on success (the error text is in shq_last_error()); *badstepsizecount as the reference returns it */
int find_timesteps(shq_context *ctx, const ActiveParticles *act, DriftKickTimes *times, TimeBinMgr *timebinmgr, double atime,
                   int FastParticleType, const Cosmology *CP, double asmth, int isFirstTimeStep, int *badstepsizecount);
int find_hydro_timesteps(shq_context *ctx, const ActiveParticles *act, DriftKickTimes *times, TimeBinMgr *timebinmgr, double atime,
                         const Cosmology *CP, int isFirstTimeStep, int *badstepsizecount);
/* have_stored_accel: StoredGravAccel.GravAccel != NULL, i.e. the last walk's Accel output on the device is the longest
 * step's acceleration.  treemask: the particle types of the sub-trees (force_tree_active_moments; ALLMASK unless hybrid
 * neutrinos are excluded).  Leaves the tree of the last sub-list on the device. */
int hierarchical_gravity_and_timesteps(shq_context *ctx, const ActiveParticles *act, PetaPM *pm, int have_stored_accel, DriftKickTimes *times,
                                       TimeBinMgr *timebinmgr, double atime, int treemask, int FastParticleType, const Cosmology *CP,
                                       int walk_mode, int64_t *badstepsizecount);
#endif
