/* timestep.cpp — see timestep.hpp.  Scalar logic of the reference's time-line operators around the device loops. */
#include "timestep.hpp"
#include <cmath>
#include <cstring>
#include <cstdio>

extern "C" void shqh_set_error(const char *msg);

static struct timestep_params TimestepParams = {0.02, 0, 0.0, 0.1, 0.2, 3e5, 0.15};

void set_timestep_params(struct timestep_params p) { TimestepParams = p; }
struct timestep_params get_timestep_params(void) { return TimestepParams; }

/* timestep.cpp:132-139 */
int is_timebin_active(int i, inttime_t current)
{
    if(i <= 0 || current <= 0)
        return 1;
    if(current % dti_from_timebin(i) == 0)
        return 1;
    return 0;
}

/* timebinmgr.cpp:189-203 */
inttime_t round_down_power_of_two(inttime_t dti)
{
    inttime_t ti_min = TIMEBASE;
    int sign = 1;
    if(dti < 0) {
        dti = -dti;
        sign = -1;
    }
    while(ti_min > dti)
        ti_min >>= 1;
    return ti_min * sign;
}

/* timestep.cpp:1236-1251 */
int get_timestep_bin(inttime_t dti)
{
    int bin = -1;
    if(dti <= 1)
        return 0;
    while(dti) {
        bin++;
        dti >>= 1;
    }
    return bin;
}

inttime_t find_next_kick(inttime_t Ti_Current, int minTimeBin) { return Ti_Current + dti_from_timebin(minTimeBin); }

/* timebinmgr.h:228-243 */
double TimeBinMgr::Dloga_interval_ti(inttime_t ti) const
{
    const inttime_t lastsnap = ti >> TIMEBINS;
    if(lastsnap >= NSyncPoints() - 1)
        return 0;
    return (loga[lastsnap + 1] - loga[lastsnap]) / TIMEBASE;
}

/* timebinmgr.h:91-102 */
double TimeBinMgr::loga_from_ti(inttime_t ti) const
{
    inttime_t lastsnap = ti >> TIMEBINS;
    if(lastsnap >= NSyncPoints())
        lastsnap = NSyncPoints() - 1;
    const double last = loga[lastsnap];
    const inttime_t dti = ti & (TIMEBASE - 1);
    return last + dti * Dloga_interval_ti(ti);
}

/* timebinmgr.h:104-121 */
inttime_t TimeBinMgr::ti_from_loga(double la) const
{
    inttime_t i, ti;
    for(i = 1; i < NSyncPoints() - 1; i++)
        if(loga[i] > la)
            break;
    const double logDTime = (loga[i] - loga[i - 1]) / TIMEBASE;
    ti = (i - 1) << TIMEBINS;
    ti += (la - loga[i - 1]) / logDTime;
    return ti;
}

/* timebinmgr.h:123-132 */
inttime_t TimeBinMgr::ti_from_loga_snap(double la, inttime_t lastsnap) const
{
    const double logDTime = (loga[lastsnap + 1] - loga[lastsnap]) / TIMEBASE;
    inttime_t ti = lastsnap << TIMEBINS;
    ti += (la - loga[lastsnap]) / logDTime;
    return ti;
}

/* timebinmgr.h:135-156 */
inttime_t TimeBinMgr::dti_from_dloga(double dloga, inttime_t Ti_Current) const
{
    inttime_t lastsnap = Ti_Current >> TIMEBINS;
    if(lastsnap >= NSyncPoints() - 1)
        lastsnap = NSyncPoints() - 1;
    const inttime_t dti = Ti_Current & (TIMEBASE - 1);
    const double logDTime = Dloga_interval_ti(Ti_Current);
    const double la = loga[lastsnap] + dti * logDTime;
    if(lastsnap >= NSyncPoints() - 1)
        lastsnap = NSyncPoints() - 2;
    if(lastsnap < NSyncPoints() - 2 && loga[lastsnap + 1] <= dloga + la)
        lastsnap++;
    return ti_from_loga_snap(dloga + la, lastsnap) - Ti_Current;
}

/* timebinmgr.h:158-170 */
double TimeBinMgr::dloga_from_dti(inttime_t dti, inttime_t Ti_Current) const
{
    const double Dloga = Dloga_interval_ti(Ti_Current);
    int sign = 1;
    if(dti < 0) {
        dti = -dti;
        sign = -1;
    }
    if((uint64_t) dti > TIMEBASE)
        dti = TIMEBASE;
    return Dloga * dti * sign;
}

double TimeBinMgr::get_dloga_for_bin(int timebin, inttime_t Ti_Current) const
{
    return dti_from_timebin(timebin) * Dloga_interval_ti(Ti_Current);
}

inttime_t TimeBinMgr::find_next_ti_sync(inttime_t ti) const { return ((ti >> TIMEBINS) + 1L) << (TIMEBINS); }

shq_timeline TimeBinMgr::timeline_at(inttime_t Ti_Current) const
{
    shq_timeline tl;
    memset(&tl, 0, sizeof(tl));
    const inttime_t N = NSyncPoints();
    inttime_t lastsnap = Ti_Current >> TIMEBINS;
    if(lastsnap >= N - 1)
        lastsnap = N - 1;
    tl.Ti_Current = Ti_Current;
    tl.Dloga_interval = Dloga_interval_ti(Ti_Current);
    tl.loga_now = loga[lastsnap] + (Ti_Current & (TIMEBASE - 1)) * tl.Dloga_interval;
    if(lastsnap >= N - 1)
        lastsnap = N - 2;
    tl.seg_snap[0] = lastsnap;
    tl.seg_loga[0] = loga[lastsnap];
    tl.seg_loga[1] = loga[lastsnap + 1];
    tl.nseg = 1;
    if(lastsnap < N - 2) {
        tl.nseg = 2;
        tl.seg_snap[1] = lastsnap + 1;
        tl.seg_loga[2] = loga[lastsnap + 2];
    }
    return tl;
}

/* timestep.cpp:142-149 */
int is_PM_timestep(const DriftKickTimes *times)
{
    return times->Ti_Current == times->PM_start + times->PM_length;
}

static int fail(int rc)
{
    shqh_set_error(shq_last_error());
    return rc;
}

static shq_timestep_params make_params(const DriftKickTimes *times, const TimeBinMgr *tbm, double atime, double hubble, inttime_t dti_max,
                                       int isFirstTimeStep)
{
    shq_timestep_params p;
    memset(&p, 0, sizeof(p));
    p.ErrTolIntAccuracy = TimestepParams.ErrTolIntAccuracy;
    p.CourantFac = TimestepParams.CourantFac;
    p.MinSizeTimestep = TimestepParams.MinSizeTimestep;
    p.ForceSoftening = FORCE_SOFTENING();
    p.atime = atime;
    p.hubble = hubble;
    const double GAMMA = 5.0 / 3;
    p.fac3 = pow(atime, 3 * (1 - GAMMA) / 2.0); /* timestep.cpp:1049 */
    p.dti_max = dti_max;
    p.ForceEqualTimesteps = TimestepParams.ForceEqualTimesteps;
    p.isFirstTimeStep = isFirstTimeStep;
    p.mintimebin = times->mintimebin;
    p.mingravtimebin = times->mingravtimebin;
    p.tl = tbm->timeline_at(times->Ti_Current);
    return p;
}

/* timestep.cpp:1141-1220 */
int get_long_range_timestep_dloga(shq_context *ctx, double atime, const Cosmology *CP, int FastParticleType, double asmth, double *out)
{
    double v_sum[6], min_mass[6];
    int64_t count_sum[6];
    if(int rc = shq_velocity_moments(ctx, v_sum, min_mass, count_sum))
        return fail(rc);
    double dloga = TimestepParams.MaxSizeTimestep;
    v_sum[0] += v_sum[4];
    count_sum[0] += count_sum[4];
    v_sum[4] = v_sum[0];
    count_sum[4] = count_sum[0];
    v_sum[0] += v_sum[5];
    count_sum[0] += count_sum[5];
    v_sum[5] = v_sum[0];
    count_sum[5] = count_sum[0];
    min_mass[5] = min_mass[0];
    const double hubble = CP->hubble_function(CP, atime);
    for(int type = 0; type < 6; type++) {
        if(count_sum[type] == 0)
            continue;
        double omega;
        if(type == 0 || type == 4 || type == 5)
            omega = CP->OmegaBaryon;
        else if(type == 2)
            omega = CP->OmegaNu1;
        else
            omega = CP->OmegaCDM;
        const double dmean = pow(min_mass[type] / (omega * CP->RhoCrit), 1.0 / 3);
        const double dloga1 = TimestepParams.MaxRMSDisplacementFac * hubble * atime * atime * (asmth < dmean ? asmth : dmean) /
                              sqrt(v_sum[type] / count_sum[type]);
        if(type != FastParticleType && dloga1 < dloga)
            dloga = dloga1;
    }
    if(dloga < TimestepParams.MinSizeTimestep)
        dloga = TimestepParams.MinSizeTimestep;
    *out = dloga;
    return 0;
}

/* timestep.cpp:1221-1233 */
int get_PM_timestep_ti(shq_context *ctx, const DriftKickTimes *times, const TimeBinMgr *timebinmgr, double atime, const Cosmology *CP,
                       int FastParticleType, double asmth, inttime_t *out)
{
    double dloga;
    if(int rc = get_long_range_timestep_dloga(ctx, atime, CP, FastParticleType, asmth, &dloga))
        return rc;
    inttime_t dti = timebinmgr->dti_from_dloga(dloga, times->Ti_Current);
    dti = round_down_power_of_two(dti);
    const inttime_t dti_max = timebinmgr->find_next_ti_sync(times->Ti_Current) - times->PM_kick;
    if(dti > dti_max)
        dti = dti_max;
    *out = dti;
    return 0;
}

/* timestep.cpp:705-822 */
int find_timesteps(shq_context *ctx, const ActiveParticles *act, DriftKickTimes *times, TimeBinMgr *timebinmgr, double atime,
                   int FastParticleType, const Cosmology *CP, double asmth, int isFirstTimeStep, int *badstepsizecount)
{
    const int isPM = is_PM_timestep(times);
    inttime_t dti_max = times->PM_length;
    if(isPM) {
        if(int rc = get_PM_timestep_ti(ctx, times, timebinmgr, atime, CP, FastParticleType, asmth, &dti_max))
            return rc;
        times->PM_length = dti_max;
        times->PM_start = times->PM_kick;
    }
    const double hubble = CP->hubble_function(CP, atime);
    shq_timestep_params p = make_params(times, timebinmgr, atime, hubble, dti_max, isFirstTimeStep);
    inttime_t dti_min = TIMEBASE;
    shq_timestep_result r;
    if(TimestepParams.ForceEqualTimesteps) {
        if(int rc = shq_find_global_timestep(ctx, &p, &r))
            return fail(rc);
        dti_min = r.dti_min;
    }
    const int32_t *list = (act && act->ActiveParticle) ? SHQ_ACTIVE_RESIDENT : nullptr;
    if(int rc = shq_find_timesteps(ctx, &p, list, 0, dti_min, -1, &r))
        return fail(rc);
    /* timestep.cpp:806-821 */
    if(isPM && times->PM_length > dti_from_timebin(r.maxTimeBin))
        times->PM_length = dti_from_timebin(r.maxTimeBin);
    times->mintimebin = r.mTimeBin;
    times->maxtimebin = r.maxTimeBin;
    if(badstepsizecount)
        *badstepsizecount = r.badstepsizecount;
    return 0;
}

/* timestep.cpp:583-703 */
int find_hydro_timesteps(shq_context *ctx, const ActiveParticles *act, DriftKickTimes *times, TimeBinMgr *timebinmgr, double atime,
                         const Cosmology *CP, int isFirstTimeStep, int *badstepsizecount)
{
    const double hubble = CP->hubble_function(CP, atime);
    shq_timestep_params p = make_params(times, timebinmgr, atime, hubble, times->PM_length, isFirstTimeStep);
    shq_timestep_result r;
    const int32_t *list = (act && act->ActiveParticle) ? SHQ_ACTIVE_RESIDENT : nullptr;
    if(int rc = shq_find_hydro_timesteps(ctx, &p, list, 0, &r))
        return fail(rc);
    times->mintimebin = r.mintimebin;
    if(badstepsizecount)
        *badstepsizecount = r.badstepsizecount;
    return 0;
}

/* apply_hierarchical_grav_kick, timestep.cpp:247-287 */
static double hierarchical_gravkick_factor(const DriftKickTimes *times, const TimeBinMgr *tbm, int ti, int largest_active)
{
    const inttime_t dti = dti_from_timebin(ti);
    double gravkick = tbm->get_exact_gravkick_factor(times->Ti_kick[ti], times->Ti_kick[ti] + dti / 2);
    if(ti < largest_active) {
        const inttime_t lowerdti = dti_from_timebin(ti + 1);
        const double lowerkick = tbm->get_exact_gravkick_factor(times->Ti_kick[ti + 1], times->Ti_kick[ti + 1] + lowerdti / 2);
        gravkick -= lowerkick;
    }
    return gravkick;
}

static int hierarchical_grav_kick(shq_context *ctx, const int32_t *list, const DriftKickTimes *times, const TimeBinMgr *tbm, int from_accel_store,
                                  int ti, int largest_active)
{
    const double gravkick = hierarchical_gravkick_factor(times, tbm, ti, largest_active);
    double tab[TIMEBINS + 1];
    for(int b = 0; b <= TIMEBINS; b++)
        tab[b] = gravkick; /* one factor for the whole sub-list, whatever bin a particle has moved to */
    if(int rc = shq_kick_short(ctx, tab, list, 0, from_accel_store))
        return fail(rc);
    return 0;
}

/* timestep.cpp:305-480 */
int hierarchical_gravity_and_timesteps(shq_context *ctx, const ActiveParticles *act, PetaPM *pm, int have_stored_accel, DriftKickTimes *times,
                                       TimeBinMgr *timebinmgr, double atime, int treemask, int FastParticleType, const Cosmology *CP,
                                       int walk_mode, int64_t *badstepsizecount_out)
{
    const int isPM = is_PM_timestep(times);
    inttime_t dti_max = times->PM_length;
    if(isPM) {
        const double asmth = pm->Asmth * pm->BoxSize / pm->Nmesh;
        if(int rc = get_PM_timestep_ti(ctx, times, timebinmgr, atime, CP, FastParticleType, asmth, &dti_max))
            return rc;
        times->PM_length = dti_max;
        times->PM_start = times->PM_kick;
    }
    const double hubble = CP->hubble_function(CP, atime);
    int ti, largest_active = TIMEBINS;
    for(ti = TIMEBINS; ti >= 0; ti--)
        if(is_timebin_active(ti, times->Ti_Current) && dti_from_timebin(ti) <= times->PM_length) {
            largest_active = ti;
            break;
        }
    shq_timestep_params p = make_params(times, timebinmgr, atime, hubble, dti_max, 0);
    const int32_t *top = (act && act->ActiveParticle) ? SHQ_ACTIVE_RESIDENT : nullptr;
    if(!(act->NumActiveGravity == act->NumActiveParticle || isPM)) {
        int64_t nsub = 0;
        if(int rc = shq_build_active_sublist(ctx, largest_active, times->Ti_Current, &nsub))
            return fail(rc);
        top = SHQ_SUBLIST_RESIDENT;
    }
    const double rho0 = CP->Omega0 * 3 * CP->Hubble * CP->Hubble / (8 * M_PI * CP->GravInternal);
    shq_timestep_result r;
    if(int rc = shq_hier_gravity_bins(ctx, &p, top, 0, have_stored_accel, largest_active, &r))
        return fail(rc);
    int64_t *timebincounts = r.timebincounts;
    for(ti = largest_active; ti >= 1; ti--)
        if(timebincounts[ti] > 0) {
            largest_active = ti;
            break;
        }
    int push_down_bin = largest_active;
    if(isPM)
        for(ti = largest_active; ti >= 1; ti--) {
            if(timebincounts[ti] / 3 > timebincounts[ti - 1])
                break;
            push_down_bin = ti - 1;
            timebincounts[ti - 1] += timebincounts[ti];
        }
    if(push_down_bin == 0) {
        char buf[120];
        snprintf(buf, sizeof(buf), "Bad timestep with %ld particles inside", (long) timebincounts[push_down_bin]);
        shqh_set_error(buf);
        return 77;
    }
    if(push_down_bin != largest_active) {
        if(int rc = shq_hier_push_down(ctx, top, 0, push_down_bin))
            return fail(rc);
        largest_active = push_down_bin;
    }
    times->maxtimebin = largest_active;
    if(int rc = hierarchical_grav_kick(ctx, top, times, timebinmgr, have_stored_accel, largest_active, largest_active))
        return rc;

    shq_grav_params gp;
    if(make_grav_params(pm, pm->BoxSize, rho0, &gp))
        return 1;
    if(int rc = shq_grav_refresh_oldacc(ctx, gp.G)) /* grav_get_abs_accel of FullTreeGravAccel + GravPM, unchanged by the sub-steps */
        return fail(rc);
    int64_t badstepsizecount = 0;
    /* the levels run inside the library (shq_hier_gravity_levels); the kick factors are the host's (cosmology integrals) */
    double kick_level[TIMEBINS + 1] = {};
    for(ti = largest_active - 1; ti > 0; ti--)
        kick_level[ti] = hierarchical_gravkick_factor(times, timebinmgr, ti, largest_active);
    int mingrav = times->mingravtimebin;
    if(int rc = shq_hier_gravity_levels(ctx, &p, &gp, pm->BoxSize, treemask, times->Ti_Current, largest_active, kick_level, walk_mode, &mingrav,
                                        &badstepsizecount, nullptr, nullptr))
        return fail(rc);
    times->mingravtimebin = mingrav;
    times->mintimebin = times->mingravtimebin;
    if(badstepsizecount_out)
        *badstepsizecount_out = badstepsizecount;
    return 0;
}
