/* timestep.hip — the particle loops of the integer time line on the resident particle set.
 *
 * Restates, one thread per list entry, with the reference's operation order (IEEE sqrt / divide, no fma contraction):
 *   convert_timestep_to_ti, get_timebin_from_dti                  libgadget/timestep.cpp:157-194
 *   TimeBinMgr::dti_from_dloga, ti_from_loga_snap, get_dloga_for_bin   libgadget/timebinmgr.h:120-176
 *   round_down_power_of_two, get_timestep_bin                     libgadget/timebinmgr.cpp:189-203, timestep.cpp:1236-1251
 *   get_timestep_gravity_dloga / _hydro_dloga / _dynfric_dloga    libgadget/timestep.cpp:1012-1110
 *   find_global_timestep, find_timesteps, find_hydro_timesteps    libgadget/timestep.cpp:195-221, 733-792, 596-658
 *   hierarchical_gravity_and_timesteps (three loops)              libgadget/timestep.cpp:356-380, 407-414, 449-464
 *   get_long_range_timestep_dloga (particle loop)                 libgadget/timestep.cpp:1153-1166
 *   black-hole half of do_hydro_kick, repositioning of the drift  libgadget/timestep.cpp:973-979, drift.cpp:32-53
 * The sync-point table, cosmology and DriftKickTimes stay on the host (integration/reference_side/timestep.cpp). */
#include "common.hpp"
#include <string.h>
#include <vector>
#include <algorithm>

namespace {

inline unsigned nblk(long long n, int t = 256) { return (unsigned) ((n + t - 1) / t); }

constexpr int TB = SHQ_TIMEBINS;
constexpr long long TIMEBASE = 1ll << TB;

enum { TI_ACCEL = 0, TI_COURANT = 1, TI_ACCRETE = 2, TI_NEIGH = 3, TI_HSML = 4 }; /* enum TimeStepType, timestep.cpp:87-94 */

struct TsArgs {
    shq_timestep_params p;
    const int32_t *targets;
    long long nt;
    const uint8_t *pflags;
    uint8_t *bin_grav, *bin_hydro;
    const double *accel;      /* FullTreeGravAccel or AccelStore */
    const double *gravpm;
    const double *hsml, *dthsml, *maxsig, *vel;
    /* black holes (may be null) */
    const int32_t *bh_pidx;
    long long nbh;
    const uint8_t *bh_mintimebin;
    uint8_t *bh_dynfric;
    const double *bh_dfsurr;
    long long dti_min_global;
    int largest_active, ti;
    unsigned long long *out;  /* OutSlots */
};

/* result slots in act_counts (unsigned long long each) */
enum { O_BAD = 0, O_MIN, O_MAX, O_ACCEL, O_COURANT, O_HSML, O_ACCRETE, O_NEIGH, O_NBH, O_DYNRATIO, O_MAXDYN, O_NBADBIN, O_DTIMIN, O_NOGAS, O_COUNTS,
       O_END = O_COUNTS + TB + 1 };

/* is_timebin_active, timestep.cpp:132-139 */
__device__ __forceinline__ bool timebin_active(int bin, long long Ti)
{
    if(bin <= 0 || Ti <= 0)
        return true;
    return (Ti & ((1ll << bin) - 1)) == 0;
}

/* round_down_power_of_two, timebinmgr.cpp:189-203 */
__device__ __forceinline__ long long round_down_pow2(long long dti)
{
    long long ti_min = TIMEBASE;
    int sign = 1;
    if(dti < 0) {
        dti = -dti;
        sign = -1;
    }
    while(ti_min > dti)
        ti_min >>= 1;
    return ti_min * sign;
}

/* get_timestep_bin, timestep.cpp:1236-1251 */
__device__ __forceinline__ int timestep_bin(long long dti)
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

/* convert_timestep_to_ti (timestep.cpp:157-174) over TimeBinMgr::dti_from_dloga (timebinmgr.h:135-156) and
 * ti_from_loga_snap (:120-130): `ti += x` with a double x is ti = (inttime_t) ((double) ti + x).  A sum that does not fit an
 * int64 (the conversion is undefined in the reference; x86 yields INT64_MIN and the step is then "overflowed") gives dti_max. */
__device__ long long convert_timestep_to_ti(double dloga, const long long dti_max, const shq_timestep_params &p)
{
#pragma clang fp contract(off)
    if(dti_max == 0)
        return 0;
    if(dloga < p.MinSizeTimestep)
        dloga = p.MinSizeTimestep;
    const shq_timeline &tl = p.tl;
    const double target = dloga + tl.loga_now;
    int s = 0;
    if(tl.nseg == 2 && tl.seg_loga[1] <= target)
        s = 1;
    const double logDTime = (tl.seg_loga[s + 1] - tl.seg_loga[s]) / (double) TIMEBASE;
    const long long base = (long long) ((unsigned long long) tl.seg_snap[s] << TB);
    const double tf = (double) base + (target - tl.seg_loga[s]) / logDTime;
    if(!(tf > -9.2e18 && tf < 9.2e18))
        return dti_max;
    const long long dti = (long long) tf - tl.Ti_Current;
    if(dti > dti_max || dti < 0)
        return dti_max;
    return dti;
}

/* get_timebin_from_dti, timestep.cpp:176-192 */
__device__ int timebin_from_dti(long long dti, int binold, long long Ti)
{
    dti = round_down_pow2(dti);
    int bin = timestep_bin(dti);
    if(bin > binold)
        while(!timebin_active(bin, Ti) && bin > binold && bin > 1)
            bin--;
    return bin;
}

/* grav_acceleration2 + get_timestep_gravity_dloga, timestep.cpp:1012-1040 */
__device__ double gravity_dloga(const double *GravAccel, const double *GravPM, const shq_timestep_params &p)
{
#pragma clang fp contract(off)
    const double a2inv = 1 / (p.atime * p.atime);
    double ax = a2inv * GravAccel[0];
    double ay = a2inv * GravAccel[1];
    double az = a2inv * GravAccel[2];
    ay += a2inv * GravPM[1];
    ax += a2inv * GravPM[0];
    az += a2inv * GravPM[2];
    double ac2 = ax * ax + ay * ay + az * az;
    if(ac2 == 0)
        ac2 = 1.0e-60;
    const double ac = sqrt(ac2);
    const double dt = sqrt(2 * p.ErrTolIntAccuracy * p.atime * (p.ForceSoftening / 2.8) / ac);
    return dt * p.hubble;
}

/* get_timestep_hydro_dloga, timestep.cpp:1042-1081 */
__device__ double hydro_dloga(long long i, int type, const TsArgs &a, int *titype)
{
#pragma clang fp contract(off)
    const shq_timestep_params &p = a.p;
    double dt = 1;
    *titype = TI_ACCEL;
    if(type == 0) {
        if(!a.hsml || !a.dthsml || !a.maxsig) { /* reported by the caller as SHQ_ERR_STATE */
            atomicAdd(&a.out[O_NOGAS], 1ull);
            return p.hubble;
        }
        const double hs = a.hsml[i];
        const double dt_courant = 2 * p.CourantFac * p.atime * hs / (p.fac3 * a.maxsig[i]);
        dt = dt_courant;
        *titype = TI_COURANT;
        const double dt_hsml = p.CourantFac * p.atime * p.atime * fabs(hs / (a.dthsml[i] + 1e-20));
        if(dt_hsml < dt) {
            dt = dt_hsml;
            *titype = TI_HSML;
        }
    } else if(type == 5 && a.bh_pidx) {
        const long long k = shq_bh_ordinal(a.bh_pidx, a.nbh, (int32_t) i);
        if(k >= 0) {
            const int mb = a.bh_mintimebin[k];
            if(mb > 0 && mb + 1 < TB) {
                /* get_dloga_for_bin(minTimeBin + 1) / hubble, timebinmgr.h:172-176 */
                const double dt_limiter = ((double) (1ll << (mb + 1)) * p.tl.Dloga_interval) / p.hubble;
                dt = dt_limiter;
                *titype = TI_NEIGH;
            }
        }
    }
    return dt * p.hubble;
}

/* get_timestep_dynfric_dloga, timestep.cpp:1085-1110 */
__device__ double dynfric_dloga(long long i, long long k, const TsArgs &a)
{
#pragma clang fp contract(off)
    const shq_timestep_params &p = a.p;
    if(!a.hsml || !a.dthsml || !a.vel) {
        atomicAdd(&a.out[O_NOGAS], 1ull);
        return p.hubble;
    }
    double bhvel = 0, bhvel2 = 0;
    for(int j = 0; j < 3; j++) {
        const double d = a.vel[3 * i + j] - a.bh_dfsurr[3 * k + j];
        bhvel += d * d;
        bhvel2 += a.vel[3 * i + j] * a.vel[3 * i + j];
    }
    if(bhvel2 > bhvel)
        bhvel = bhvel2;
    bhvel = sqrt(bhvel);
    double dt = 2 * p.ErrTolIntAccuracy * p.atime * p.atime * a.hsml[i] / (bhvel + 1e-20);
    const double dt_hsml = p.CourantFac * p.atime * p.atime * fabs(a.hsml[i] / (a.dthsml[i] + 1e-20));
    if(dt_hsml < dt)
        dt = dt_hsml;
    return dt * p.hubble;
}

/* Tallies of the lanes that are active here, into the workgroup's LDS copy of the result slots.  One atomic per distinct slot and
 * wave instead of one per lane: 64 lanes adding to the same LDS word serialise (that was 2 of the kernel's 2.3 ms at 256^3). */
__device__ __forceinline__ void tally_add1(unsigned long long *out, int slot)
{
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(1);
    while(todo) {
        const int leader = __ffsll((long long) todo) - 1;
        const int ls = __shfl(slot, leader);
        const unsigned long long same = __ballot(slot == ls);
        if(lane == leader)
            atomicAdd(&out[ls], (unsigned long long) __popcll(same));
        todo &= ~same;
    }
}
/* most lanes find the slot already at or beyond their value: a plain read first */
__device__ __forceinline__ void tally_min(unsigned long long *out, int slot, long long v)
{
    if(v < (long long) ((volatile unsigned long long *) out)[slot])
        atomicMin((long long *) &out[slot], v);
}
__device__ __forceinline__ void tally_max(unsigned long long *out, int slot, long long v)
{
    if(v > (long long) ((volatile unsigned long long *) out)[slot])
        atomicMax((long long *) &out[slot], v);
}

__device__ __forceinline__ void count_titype(int titype, unsigned long long *out)
{
    const int slot = titype == TI_ACCEL ? O_ACCEL : titype == TI_COURANT ? O_COURANT : titype == TI_ACCRETE ? O_ACCRETE : titype == TI_NEIGH ? O_NEIGH : O_HSML;
    tally_add1(out, slot);
}

/* MODE 0 find_timesteps, 1 find_hydro_timesteps, 2 hierarchical first loop, 3 hierarchical refinement, 4 find_global_timestep.
 * The tallies are 64-bit atomics into a handful of slots: a few per particle at most, and the loops are far from the step's
 * critical path (one pass over 16.8 M particles moves 60 B each). */
template <int MODE>
__device__ void timestep_body(const TsArgs &a, const long long t)
{
#pragma clang fp contract(off)
    const long long i = a.targets ? (long long) a.targets[t] : t;
    const unsigned f = a.pflags[i];
    if(f & 3u)
        return;
    const int type = f >> 4;
    const shq_timestep_params &p = a.p;
    const long long Ti = p.tl.Ti_Current;
    if(MODE == 0 || MODE == 4) {
        int titype = TI_ACCEL;
        long long dti;
        double dloga_gravity = 0;
        if(MODE == 0 && p.ForceEqualTimesteps)
            dti = a.dti_min_global;
        else {
            dloga_gravity = gravity_dloga(a.accel + 3 * i, a.gravpm + 3 * i, p);
            if(MODE == 0) {
                dti = convert_timestep_to_ti(dloga_gravity, p.dti_max, p);
                if(type == 0 || type == 5) {
                    int th = TI_ACCEL;
                    const double dloga_hydro = hydro_dloga(i, type, a, &th);
                    const long long dti_hydro = convert_timestep_to_ti(dloga_hydro, p.dti_max, p);
                    if(dti_hydro < dti) {
                        dti = dti_hydro;
                        titype = th;
                    }
                }
            } else { /* find_global_timestep: min of the two dloga, one conversion */
                double dloga = dloga_gravity;
                int th = TI_ACCEL;
                const double dloga_hydro = hydro_dloga(i, type, a, &th);
                if(dloga_hydro < dloga)
                    dloga = dloga_hydro;
                dti = convert_timestep_to_ti(dloga, p.dti_max, p);
                tally_min(a.out, O_DTIMIN, dti);
            }
            if(dti <= 1 || dti > TIMEBASE)
                atomicAdd(&a.out[O_NBADBIN], 1ull);
            if(MODE == 0)
                count_titype(titype, a.out);
        }
        if(MODE == 4)
            return;
        const int bin = timebin_from_dti(dti, a.bin_hydro[i], Ti);
        if(bin < 1)
            atomicAdd(&a.out[O_BAD], 1ull);
        if(timebin_active(a.bin_hydro[i], Ti) && timebin_active(bin, Ti)) {
            a.bin_hydro[i] = (uint8_t) bin;
            a.bin_grav[i] = (uint8_t) bin;
        }
        tally_min(a.out, O_MIN, (long long) bin);
        tally_max(a.out, O_MAX, (long long) bin);
    } else if(MODE == 1) {
        if(type != 0 && type != 5)
            return;
        int titype = TI_ACCEL;
        const double dloga_hydro = hydro_dloga(i, type, a, &titype);
        const long long dti_hydro = convert_timestep_to_ti(dloga_hydro, p.dti_max, p);
        if(dti_hydro <= 1 || dti_hydro > TIMEBASE)
            atomicAdd(&a.out[O_NBADBIN], 1ull);
        int bin_hydro = timebin_from_dti(dti_hydro, a.bin_hydro[i], Ti);
        const int bg = a.bin_grav[i];
        if(bin_hydro > bg) {
            bin_hydro = bg;
            titype = TI_ACCEL;
        }
        if(bin_hydro < 1)
            atomicAdd(&a.out[O_BAD], 1ull);
        count_titype(titype, a.out);
        if(timebin_active(a.bin_hydro[i], Ti) && timebin_active(bin_hydro, Ti))
            a.bin_hydro[i] = (uint8_t) bin_hydro;
        tally_min(a.out, O_MIN, (long long) bin_hydro);
        if(type == 5 && a.bh_pidx) {
            const long long k = shq_bh_ordinal(a.bh_pidx, a.nbh, (int32_t) i);
            if(k >= 0) {
                const double dloga_dynfric = dynfric_dloga(i, k, a);
                const long long dti_dynfric = convert_timestep_to_ti(dloga_dynfric, p.dti_max, p);
                int bin_dynfric = timebin_from_dti(dti_dynfric, a.bh_dynfric[k], Ti);
                const int bh = a.bin_hydro[i];
                if(bin_dynfric > bg)
                    bin_dynfric = bg;
                if(bin_dynfric < bh)
                    bin_dynfric = bh;
                a.bh_dynfric[k] = (uint8_t) bin_dynfric;
                atomicAdd(&a.out[O_DYNRATIO], (unsigned long long) (long long) (bin_dynfric - bh));
                atomicMax((long long *) &a.out[O_MAXDYN], (long long) (bin_dynfric - bh));
                atomicAdd(&a.out[O_NBH], 1ull);
            }
        }
    } else if(MODE == 2) {
        const double dloga_gravity = gravity_dloga(a.accel + 3 * i, a.gravpm + 3 * i, p);
        long long dti_gravity = convert_timestep_to_ti(dloga_gravity, p.dti_max, p);
        dti_gravity = round_down_pow2(dti_gravity);
        if(dti_gravity <= 1 || dti_gravity > TIMEBASE)
            atomicAdd(&a.out[O_NBADBIN], 1ull);
        int bin = timestep_bin(dti_gravity);
        if(bin > a.largest_active)
            bin = a.largest_active;
        tally_add1(a.out, O_COUNTS + bin);
        a.bin_grav[i] = (uint8_t) bin;
    } else if(MODE == 3) {
        const double dloga_gravity = gravity_dloga(a.accel + 3 * i, a.gravpm + 3 * i, p);
        const long long dti_gravity = convert_timestep_to_ti(dloga_gravity, p.dti_max, p);
        if(dti_gravity < (a.ti > 0 ? (1ll << a.ti) : 0ll)) {
            a.bin_grav[i] = (uint8_t) (a.ti - 1);
            if(a.ti == 1) {
                atomicAdd(&a.out[O_BAD], 1ull);
                atomicAdd(&a.out[O_NBADBIN], 1ull);
            }
        }
    }
}

/* The tallies of a workgroup go to a copy of the result slots in LDS first and from there with one global atomic per slot: with
 * every thread on the same dozen global addresses the first version of this kernel took 207 ms at 256^3 (rocprofv3, round 2); the
 * loop itself moves ~60 B per particle. */
__device__ __forceinline__ bool slot_is_min(int k) { return k == O_MIN || k == O_DTIMIN; }
__device__ __forceinline__ bool slot_is_max(int k) { return k == O_MAX || k == O_MAXDYN; }

inline unsigned ts_grid(long long nt) { const unsigned b = nblk(nt); return b < 2048u ? b : 2048u; }

template <int MODE>
__global__ __launch_bounds__(256) void timestep_kernel(TsArgs a)
{
    __shared__ unsigned long long sh[O_END];
    for(int k = threadIdx.x; k < O_END; k += blockDim.x)
        sh[k] = k == O_MIN ? (unsigned long long) (long long) TB : (k == O_DTIMIN ? (unsigned long long) TIMEBASE : 0ull);
    __syncthreads();
    unsigned long long *const gout = a.out;
    a.out = sh;
    /* a few thousand workgroups stride over the targets: every workgroup ends with a handful of atomics on the same few global
     * words, and 65 536 workgroups' worth of those serialise in the L2 (that, not the loop, was the kernel's 2.3 ms at 256^3) */
    for(long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x; t < a.nt; t += (long long) gridDim.x * blockDim.x)
        timestep_body<MODE>(a, t);
    __syncthreads();
    for(int k = threadIdx.x; k < O_END; k += blockDim.x) {
        const unsigned long long v = sh[k];
        if(slot_is_min(k)) {
            if((long long) v < (k == O_MIN ? (long long) TB : (long long) TIMEBASE))
                atomicMin((long long *) &gout[k], (long long) v);
        } else if(slot_is_max(k)) {
            if((long long) v > 0)
                atomicMax((long long *) &gout[k], (long long) v);
        } else if(v)
            atomicAdd(&gout[k], v);
    }
}

/* the push-down loop, timestep.cpp:407-414 (no garbage test in the reference either) */
__global__ void push_down_kernel(long long nt, const int32_t *targets, uint8_t *bin_grav, int push_down_bin)
{
    const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= nt)
        return;
    const long long i = targets ? (long long) targets[t] : t;
    if(bin_grav[i] > push_down_bin)
        bin_grav[i] = (uint8_t) push_down_bin;
}

/* set_bh_first_timestep, timestep.cpp:567-578 */
__global__ void bh_first_kernel(long long n, const uint8_t *pflags, uint8_t *bin_hydro, int mTimeBin)
{
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n && (pflags[i] >> 4) == 5)
        bin_hydro[i] = (uint8_t) mTimeBin;
}

/* get_long_range_timestep_dloga, timestep.cpp:1153-1166.  part[b][18]: per block v2[6], minmass[6], count[6]; the sum of a
 * block runs over its 256 particles in index order (lane 0 after an LDS gather), the block sums are added in order on the host. */
__global__ __launch_bounds__(256) void velmom_kernel(long long n, const double4 *posm, const double *vel, const uint8_t *pflags, double *part)
{
#pragma clang fp contract(off)
    __shared__ double v2[256];
    __shared__ double ms[256];
    __shared__ int ty[256];
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    int type = -1;
    double v = 0, m = 0;
    if(i < n) {
        const unsigned f = pflags[i];
        if(!(f & 3u)) {
            type = f >> 4;
            v = vel[3 * i] * vel[3 * i] + vel[3 * i + 1] * vel[3 * i + 1] + vel[3 * i + 2] * vel[3 * i + 2];
            m = posm[i].w;
        }
    }
    v2[threadIdx.x] = v;
    ms[threadIdx.x] = m;
    ty[threadIdx.x] = type;
    __syncthreads();
    if(threadIdx.x < 6) {
        const int T = threadIdx.x;
        double s = 0, mn = 1.0e30;
        long long c = 0;
        for(int k = 0; k < 256; k++)
            if(ty[k] == T) {
                s += v2[k];
                if(ms[k] > 0 && mn > ms[k])
                    mn = ms[k];
                c++;
            }
        double *o = part + (size_t) blockIdx.x * 18;
        o[T] = s;
        o[6 + T] = mn;
        o[12 + T] = (double) c;
    }
}

/* black-hole half of do_hydro_kick, timestep.cpp:973-979, over the list */
struct KickTab { double k[SHQ_TIMEBINS + 1]; };
__global__ void kick_bh_kernel(long long nt, const int32_t *targets, double *vel, const uint8_t *pflags, const uint8_t *bin_hydro,
                               const int32_t *bh_pidx, long long nbh, const double *dfaccel, const double *dragaccel, KickTab tab)
{
#pragma clang fp contract(off)
    const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= nt)
        return;
    const long long i = targets ? (long long) targets[t] : t;
    const unsigned f = pflags[i];
    if((f & 3u) || (f >> 4) != 5)
        return;
    const long long k = shq_bh_ordinal(bh_pidx, nbh, (int32_t) i);
    if(k < 0)
        return;
    const double F = tab.k[bin_hydro[i]];
    for(int j = 0; j < 3; j++) {
        vel[3 * i + j] += dfaccel[3 * k + j] * F;
        vel[3 * i + j] += dragaccel[3 * k + j] * F;
    }
}

template <typename T> const T *cfield(const void *base, size_t elsize, int64_t i, size_t off)
{
    return reinterpret_cast<const T *>(static_cast<const char *>(base) + (size_t) i * elsize + off);
}
template <typename T> T *wfield(void *base, size_t elsize, int64_t i, size_t off)
{
    return reinterpret_cast<T *>(static_cast<char *>(base) + (size_t) i * elsize + off);
}

int fill_args(shq_context *ctx, const shq_timestep_params *p, const int32_t *active, int64_t nactive, int from_accel_store, TsArgs &a,
              const char *what)
{
    SHQ_CHECK(ctx && p, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->have_parts && ctx->bin_grav.ptr && ctx->bin_hydro.ptr, SHQ_ERR_STATE,
              "%s: upload particles and time bins (shq_dynamics_upload / shq_timebins_upload) first", what);
    SHQ_CHECK(p->atime > 0 && p->hubble > 0 && p->ForceSoftening > 0, SHQ_ERR_INVALID, "%s: atime, hubble and ForceSoftening must be > 0", what);
    SHQ_CHECK(p->tl.nseg == 1 || p->tl.nseg == 2, SHQ_ERR_INVALID, "%s: timeline nseg must be 1 or 2", what);
    SHQ_CHECK(p->dti_max >= 0 && p->dti_max <= TIMEBASE, SHQ_ERR_INVALID, "%s: dti_max out of range", what);
    SHQ_HIP(hipSetDevice(ctx->device));
    a.p = *p;
    a.accel = from_accel_store ? ctx->acc.ptr : ctx->treeacc.ptr;
    a.gravpm = ctx->gravpm.ptr;
    SHQ_CHECK(a.accel && a.gravpm, SHQ_ERR_STATE, "%s: no accelerations on the device yet (walk and PM first)", what);
    a.pflags = ctx->pflags.ptr;
    a.bin_grav = ctx->bin_grav.ptr;
    a.bin_hydro = ctx->bin_hydro.ptr;
    a.hsml = ctx->hsml.ptr;
    a.dthsml = ctx->dthsml.ptr;
    a.maxsig = ctx->g_maxsignalvel.ptr;
    a.vel = ctx->vel.ptr;
    a.bh_pidx = ctx->nbh > 0 ? ctx->bh_pidx.ptr : nullptr;
    a.nbh = ctx->nbh;
    a.bh_mintimebin = ctx->bh_u8.ptr;
    a.bh_dynfric = ctx->bh_u8.ptr ? ctx->bh_u8.ptr + ctx->nbh : nullptr;
    a.bh_dfsurr = ctx->bh_vec.ptr ? ctx->bh_vec.ptr + 3 * ctx->nbh : nullptr;
    a.dti_min_global = 0;
    a.largest_active = TB;
    a.ti = 0;
    SHQ_TRY(ctx->act_counts.reserve(std::max<size_t>(6 * (TB + 1) + 4, O_END)));
    a.out = ctx->act_counts.ptr;
    unsigned long long init[O_END];
    memset(init, 0, sizeof(init));
    init[O_MIN] = (unsigned long long) (long long) TB;
    init[O_DTIMIN] = (unsigned long long) TIMEBASE;
    SHQ_HIP(hipMemcpyAsync(a.out, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
    SHQ_HIP(hipStreamSynchronize(ctx->stream)); /* init lives on this frame */
    const int32_t *d_act = nullptr;
    int64_t nt = 0;
    SHQ_TRY(shq_resolve_active(ctx, active, nactive, ctx->numpart, &d_act, &nt));
    a.targets = d_act;
    a.nt = nt;
    return SHQ_OK;
}

int fetch(shq_context *ctx, shq_timestep_result *res)
{
    unsigned long long h[O_END];
    SHQ_HIP(hipMemcpyAsync(h, ctx->act_counts.ptr, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    /* a gas particle reads Hsml, DtHsml and MaxSignalVel; without gas on the list the arrays are never touched */
    SHQ_CHECK(h[O_NOGAS] == 0, SHQ_ERR_STATE, "time steps: %llu gas particles but no Hsml / DtHsml / MaxSignalVel on the device "
              "(shq_dynamics_upload, an SPH run or shq_maxsignalvel_upload provide them)", h[O_NOGAS]);
    if(res) {
        memset(res, 0, sizeof(*res));
        res->badstepsizecount = (int32_t) h[O_BAD];
        res->mTimeBin = (int32_t) (long long) h[O_MIN];
        res->maxTimeBin = (int32_t) (long long) h[O_MAX];
        res->mintimebin = res->mTimeBin;
        res->ntiaccel = (int64_t) h[O_ACCEL];
        res->nticourant = (int64_t) h[O_COURANT];
        res->ntihsml = (int64_t) h[O_HSML];
        res->ntiaccrete = (int64_t) h[O_ACCRETE];
        res->ntineighbour = (int64_t) h[O_NEIGH];
        res->nbh = (int64_t) h[O_NBH];
        res->dynratio = (int64_t) h[O_DYNRATIO];
        res->maxdyndiff = (int32_t) (long long) h[O_MAXDYN];
        res->nbadbin = (int32_t) h[O_NBADBIN];
        res->dti_min = (int64_t) h[O_DTIMIN];
        for(int b = 0; b <= TB; b++)
            res->timebincounts[b] = (int64_t) h[O_COUNTS + b];
    }
    return SHQ_OK;
}

} // namespace

extern "C" int shq_set_bh_first_timestep(shq_context *ctx, int mTimeBin)
{
    SHQ_CHECK(ctx && mTimeBin >= 0 && mTimeBin <= TB, SHQ_ERR_INVALID, "set_bh_first_timestep: bin %d out of range", mTimeBin);
    SHQ_CHECK(ctx->have_parts && ctx->bin_hydro.ptr, SHQ_ERR_STATE, "set_bh_first_timestep: no time bins on the device");
    SHQ_HIP(hipSetDevice(ctx->device));
    if(ctx->numpart > 0) {
        bh_first_kernel<<<dim3(nblk(ctx->numpart)), dim3(256), 0, ctx->stream>>>(ctx->numpart, ctx->pflags.ptr, ctx->bin_hydro.ptr, mTimeBin);
        SHQ_HIP(hipGetLastError());
    }
    ctx->n_act = ctx->n_sub = -1;
    return SHQ_OK;
}

extern "C" int shq_find_timesteps(shq_context *ctx, const shq_timestep_params *p, const int32_t *active, int64_t nactive, int64_t dti_min_global,
                                  int mTimeBin_global, shq_timestep_result *res)
{
    TsArgs a;
    SHQ_TRY(fill_args(ctx, p, active, nactive, 0, a, "find_timesteps"));
    a.dti_min_global = dti_min_global;
    if(a.nt > 0) {
        timestep_kernel<0><<<dim3(ts_grid(a.nt)), dim3(256), 0, ctx->stream>>>(a);
        SHQ_HIP(hipGetLastError());
    }
    shq_timestep_result r;
    SHQ_TRY(fetch(ctx, &r));
    if(p->isFirstTimeStep)
        SHQ_TRY(shq_set_bh_first_timestep(ctx, mTimeBin_global >= 0 ? mTimeBin_global : r.mTimeBin));
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    ctx->n_act = ctx->n_sub = -1; /* the bins changed: resident lists are stale (the reference rebuilds them next step too) */
    if(res)
        *res = r;
    return SHQ_OK;
}

extern "C" int shq_find_global_timestep(shq_context *ctx, const shq_timestep_params *p, shq_timestep_result *res)
{
    TsArgs a;
    SHQ_TRY(fill_args(ctx, p, nullptr, 0, 0, a, "find_global_timestep"));
    if(a.nt > 0) {
        timestep_kernel<4><<<dim3(ts_grid(a.nt)), dim3(256), 0, ctx->stream>>>(a);
        SHQ_HIP(hipGetLastError());
    }
    return fetch(ctx, res);
}

extern "C" int shq_find_hydro_timesteps(shq_context *ctx, const shq_timestep_params *p, const int32_t *active, int64_t nactive,
                                        shq_timestep_result *res)
{
    TsArgs a;
    SHQ_TRY(fill_args(ctx, p, active, nactive, 0, a, "find_hydro_timesteps"));
    if(a.nt > 0) {
        timestep_kernel<1><<<dim3(ts_grid(a.nt)), dim3(256), 0, ctx->stream>>>(a);
        SHQ_HIP(hipGetLastError());
    }
    shq_timestep_result r;
    SHQ_TRY(fetch(ctx, &r));
    /* timestep.cpp:677-696 for one rank */
    const long long Ti = p->tl.Ti_Current;
    auto active_bin = [Ti](int bin) { return bin <= 0 || Ti <= 0 || (Ti & ((1ll << bin) - 1)) == 0; };
    int mTimeBin = r.mTimeBin;
    if(!active_bin(mTimeBin)) {
        mTimeBin = p->mintimebin;
        if(active_bin(mTimeBin + 1))
            mTimeBin++;
    }
    if(p->isFirstTimeStep)
        SHQ_TRY(shq_set_bh_first_timestep(ctx, mTimeBin));
    r.mTimeBin = mTimeBin;
    r.mintimebin = mTimeBin;
    if(r.mintimebin > p->mingravtimebin && p->mingravtimebin > 0)
        r.mintimebin = p->mingravtimebin;
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    ctx->n_act = ctx->n_sub = -1;
    if(res)
        *res = r;
    return SHQ_OK;
}

extern "C" int shq_hier_gravity_bins(shq_context *ctx, const shq_timestep_params *p, const int32_t *active, int64_t nactive, int from_accel_store,
                                     int largest_active, shq_timestep_result *res)
{
    SHQ_CHECK(largest_active >= 0 && largest_active <= TB, SHQ_ERR_INVALID, "hier_gravity_bins: largest_active %d out of range", largest_active);
    TsArgs a;
    SHQ_TRY(fill_args(ctx, p, active, nactive, from_accel_store, a, "hier_gravity_bins"));
    a.largest_active = largest_active;
    if(a.nt > 0) {
        timestep_kernel<2><<<dim3(ts_grid(a.nt)), dim3(256), 0, ctx->stream>>>(a);
        SHQ_HIP(hipGetLastError());
    }
    return fetch(ctx, res);
}

extern "C" int shq_hier_push_down(shq_context *ctx, const int32_t *active, int64_t nactive, int push_down_bin)
{
    SHQ_CHECK(ctx && push_down_bin >= 0 && push_down_bin <= TB, SHQ_ERR_INVALID, "hier_push_down: bin %d out of range", push_down_bin);
    SHQ_CHECK(ctx->have_parts && ctx->bin_grav.ptr, SHQ_ERR_STATE, "hier_push_down: no time bins on the device");
    SHQ_HIP(hipSetDevice(ctx->device));
    const int32_t *d_act = nullptr;
    int64_t nt = 0;
    SHQ_TRY(shq_resolve_active(ctx, active, nactive, ctx->numpart, &d_act, &nt));
    if(nt > 0) {
        push_down_kernel<<<dim3(nblk(nt)), dim3(256), 0, ctx->stream>>>((long long) nt, d_act, ctx->bin_grav.ptr, push_down_bin);
        SHQ_HIP(hipGetLastError());
    }
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    return SHQ_OK;
}

extern "C" int shq_hier_refine(shq_context *ctx, const shq_timestep_params *p, const int32_t *active, int64_t nactive, int from_accel_store, int ti,
                               shq_timestep_result *res)
{
    SHQ_CHECK(ti >= 1 && ti <= TB, SHQ_ERR_INVALID, "hier_refine: bin %d out of range", ti);
    TsArgs a;
    SHQ_TRY(fill_args(ctx, p, active, nactive, from_accel_store, a, "hier_refine"));
    a.ti = ti;
    if(a.nt > 0) {
        timestep_kernel<3><<<dim3(ts_grid(a.nt)), dim3(256), 0, ctx->stream>>>(a);
        SHQ_HIP(hipGetLastError());
    }
    SHQ_TRY(fetch(ctx, res));
    /* the accelerations this refinement read (and the level's kick is about to use) came from a walk queued before it: fetch() has
     * waited for the stream, so the pair kernel's status of that walk is on the host */
    ctx->sp_check_pending = false;
    return shq_walk_check_status(ctx, false);
}

/* The sub-step levels of hierarchical_gravity_and_timesteps (timestep.cpp:417-476) as one resident loop: for ti = largest_active - 1
 * down to 1: the sub-list of the particles whose gravity bin is <= ti (build_active_sublist, :1373-1399), a tree of just those
 * (force_tree_rebuild_mask over the sub-list, :437-441), their short-range walk without potential into AccelStore (:446), the bin
 * refinement from that acceleration (:449-464) and the hierarchical kick of the level (:469).  Nothing but the sub-list length and
 * the refinement tallies crosses PCIe.  The walk of a level is launched with `walk_mode` (SHQ_WALK_AUTO picks the sparse-list walk
 * where a list is short against its tree). */
extern "C" int shq_hier_gravity_levels(shq_context *ctx, const shq_timestep_params *p, const shq_grav_params *gp, double BoxSize, int treemask,
                                       int64_t Ti_Current, int largest_active, const double gravkick_level[SHQ_TIMEBINS + 1], int walk_mode,
                                       int *mingravtimebin, int64_t *badstepsizecount, shq_hier_level *levels, int *nlevels)
{
    SHQ_CHECK(ctx && p && gp && gravkick_level && mingravtimebin && badstepsizecount, SHQ_ERR_INVALID, "hier_gravity_levels: null argument");
    SHQ_CHECK(largest_active >= 1 && largest_active <= TB, SHQ_ERR_INVALID, "hier_gravity_levels: largest_active %d out of range", largest_active);
    int64_t bad = 0;
    int nl = 0;
    for(int ti = largest_active - 1; ti > 0; ti--) {
        int64_t nsub = 0;
        /* build_active_sublist(lastact, ti): the predicate is monotone in ti and bins only change inside the previous sub-list,
         * so selecting from the full resident list gives the same particles in the same order */
        SHQ_TRY(shq_build_active_sublist(ctx, ti, Ti_Current, &nsub));
        if(nsub == 0) {
            *mingravtimebin = ti + 1;
            break;
        }
        shq_tree_build_stats ts;
        SHQ_TRY(shq_tree_build(ctx, BoxSize, treemask, SHQ_SUBLIST_RESIDENT, 0, &ts));
        SHQ_TRY(shq_grav_short_run(ctx, gp, SHQ_SUBLIST_RESIDENT, 0, 0, walk_mode));
        shq_timestep_result r;
        SHQ_TRY(shq_hier_refine(ctx, p, SHQ_SUBLIST_RESIDENT, 0, 1, ti, &r)); /* synchronises: the walk's events are readable */
        bad += r.badstepsizecount;
        double tab[SHQ_TIMEBINS + 1];
        for(int b = 0; b <= SHQ_TIMEBINS; b++)
            tab[b] = gravkick_level[ti]; /* one factor for the whole sub-list, whatever bin a particle has moved to */
        SHQ_TRY(shq_kick_short(ctx, tab, SHQ_SUBLIST_RESIDENT, 0, 1));
        if(levels) {
            float wms = 0;
            (void) hipEventElapsedTime(&wms, ctx->ev_begin[SHQ_NTIMERS - 1], ctx->ev_end[SHQ_NTIMERS - 1]);
            levels[nl].timebin = ti;
            levels[nl].walk_mode = ctx->last_walk_mode;
            levels[nl].nparticles = nsub;
            levels[nl].tree_nodes = ts.numnodes;
            levels[nl].tree_build_ms = ts.build_ms;
            levels[nl].walk_ms = wms;
        }
        nl++;
    }
    *badstepsizecount = bad;
    if(nlevels)
        *nlevels = nl;
    return SHQ_OK;
}

extern "C" int shq_velocity_moments(shq_context *ctx, double v2sum[6], double min_mass[6], int64_t count[6])
{
    SHQ_CHECK(ctx && v2sum && min_mass && count, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->have_parts && ctx->vel.ptr, SHQ_ERR_STATE, "velocity_moments: upload particles and velocities first");
    SHQ_HIP(hipSetDevice(ctx->device));
    const long long n = ctx->numpart;
    for(int t = 0; t < 6; t++) {
        v2sum[t] = 0;
        min_mass[t] = 1.0e30;
        count[t] = 0;
    }
    if(n == 0)
        return SHQ_OK;
    const unsigned nb = nblk(n);
    SHQ_TRY(ctx->act_temp.reserve((size_t) nb * 18 * sizeof(double) + 16));
    double *d_part = reinterpret_cast<double *>(ctx->act_temp.ptr);
    velmom_kernel<<<dim3(nb), dim3(256), 0, ctx->stream>>>(n, ctx->posm.ptr, ctx->vel.ptr, ctx->pflags.ptr, d_part);
    SHQ_HIP(hipGetLastError());
    std::vector<double> h((size_t) nb * 18);
    SHQ_HIP(hipMemcpyAsync(h.data(), d_part, sizeof(double) * h.size(), hipMemcpyDeviceToHost, ctx->stream));
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    for(unsigned b = 0; b < nb; b++)
        for(int t = 0; t < 6; t++) {
            v2sum[t] += h[(size_t) b * 18 + t];
            if(min_mass[t] > h[(size_t) b * 18 + 6 + t])
                min_mass[t] = h[(size_t) b * 18 + 6 + t];
            count[t] += (int64_t) h[(size_t) b * 18 + 12 + t];
        }
    return SHQ_OK;
}

extern "C" int shq_timebins_download(shq_context *ctx, uint8_t *bin_gravity, uint8_t *bin_hydro)
{
    SHQ_CHECK(ctx, SHQ_ERR_INVALID, "null context");
    SHQ_CHECK(ctx->have_parts && ctx->bin_grav.ptr && ctx->bin_hydro.ptr, SHQ_ERR_STATE, "timebins_download: no time bins on the device");
    SHQ_HIP(hipSetDevice(ctx->device));
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    if(ctx->numpart > 0) {
        if(bin_gravity)
            SHQ_HIP(hipMemcpy(bin_gravity, ctx->bin_grav.ptr, (size_t) ctx->numpart, hipMemcpyDeviceToHost));
        if(bin_hydro)
            SHQ_HIP(hipMemcpy(bin_hydro, ctx->bin_hydro.ptr, (size_t) ctx->numpart, hipMemcpyDeviceToHost));
    }
    return SHQ_OK;
}

extern "C" int shq_maxsignalvel_upload(shq_context *ctx, const double *maxsignalvel_by_particle)
{
    SHQ_CHECK(ctx, SHQ_ERR_INVALID, "null context");
    SHQ_CHECK(ctx->have_parts, SHQ_ERR_STATE, "maxsignalvel_upload: upload particles first");
    SHQ_HIP(hipSetDevice(ctx->device));
    const size_t cap = (size_t) std::max<int64_t>(ctx->numpart, 1);
    const bool fresh = ctx->g_maxsignalvel.ptr == nullptr || ctx->g_maxsignalvel.cap < cap;
    SHQ_TRY(ctx->g_maxsignalvel.reserve(cap));
    if(maxsignalvel_by_particle && ctx->numpart > 0)
        SHQ_HIP(hipMemcpy(ctx->g_maxsignalvel.ptr, maxsignalvel_by_particle, sizeof(double) * ctx->numpart, hipMemcpyHostToDevice));
    else if(fresh)
        SHQ_HIP(hipMemset(ctx->g_maxsignalvel.ptr, 0, sizeof(double) * cap));
    return SHQ_OK;
}

extern "C" int shq_set_bh_reposition(shq_context *ctx, int enabled)
{
    SHQ_CHECK(ctx, SHQ_ERR_INVALID, "null context");
    ctx->bh_reposition = enabled != 0;
    return SHQ_OK;
}

/* device layout: bh_pidx[nbh] ascending particle indices; bh_u8 = {minTimeBin[nbh], TimeBinDynFric[nbh], JumpToMinPot[nbh]};
 * bh_vec = {DFAccel, DF_SurroundingVel, DragAccel, MinPotPos, MinPotVel} x [nbh][3] */
extern "C" int shq_bh_dynamics_upload(shq_context *ctx, const shq_part_view *parts, const shq_bh_dyn_view *bh)
{
    SHQ_CHECK(ctx && parts && bh, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->have_parts && parts->numpart == ctx->numpart, SHQ_ERR_STATE, "bh_dynamics_upload: upload the same particles first");
    SHQ_CHECK(parts->off_type != SHQ_NOFIELD && parts->off_pi != SHQ_NOFIELD, SHQ_ERR_INVALID, "bh_dynamics_upload: the particle view needs Type and PI");
    SHQ_CHECK(bh->numslots == 0 || bh->base, SHQ_ERR_INVALID, "bh_dynamics_upload: BH slot view is NULL");
    SHQ_HIP(hipSetDevice(ctx->device));
    std::vector<int32_t> pidx;
    for(int64_t i = 0; i < parts->numpart; i++)
        if(*cfield<uint8_t>(parts->base, parts->elsize, i, parts->off_type) == 5)
            pidx.push_back((int32_t) i);
    const size_t nbh = pidx.size();
    std::vector<uint8_t> u8(3 * std::max<size_t>(nbh, 1), 0);
    std::vector<double> vec(15 * std::max<size_t>(nbh, 1), 0.0);
    const size_t offs[5] = {bh->off_dfaccel, bh->off_df_surroundingvel, bh->off_dragaccel, bh->off_minpotpos, bh->off_minpotvel};
    for(size_t k = 0; k < nbh; k++) {
        const int32_t pi = *cfield<int32_t>(parts->base, parts->elsize, pidx[k], parts->off_pi);
        SHQ_CHECK(pi >= 0 && pi < bh->numslots, SHQ_ERR_INVALID, "BH particle %d with PI %d outside the BH slot array", pidx[k], pi);
        u8[k] = bh->off_mintimebin != SHQ_NOFIELD ? *cfield<uint8_t>(bh->base, bh->elsize, pi, bh->off_mintimebin) : 0;
        u8[nbh + k] = bh->off_timebindynfric != SHQ_NOFIELD ? *cfield<uint8_t>(bh->base, bh->elsize, pi, bh->off_timebindynfric) : 0;
        u8[2 * nbh + k] = bh->off_jumptominpot != SHQ_NOFIELD ? (uint8_t) (*cfield<char>(bh->base, bh->elsize, pi, bh->off_jumptominpot) != 0) : 0;
        for(int v = 0; v < 5; v++)
            if(offs[v] != SHQ_NOFIELD) {
                const double *src = cfield<double>(bh->base, bh->elsize, pi, offs[v]);
                for(int j = 0; j < 3; j++)
                    vec[(size_t) v * 3 * nbh + 3 * k + j] = src[j];
            }
    }
    SHQ_TRY(ctx->bh_pidx.reserve(std::max<size_t>(nbh, 1)));
    SHQ_TRY(ctx->bh_u8.reserve(3 * std::max<size_t>(nbh, 1)));
    SHQ_TRY(ctx->bh_vec.reserve(15 * std::max<size_t>(nbh, 1)));
    if(nbh > 0) {
        SHQ_HIP(hipMemcpy(ctx->bh_pidx.ptr, pidx.data(), sizeof(int32_t) * nbh, hipMemcpyHostToDevice));
        SHQ_HIP(hipMemcpy(ctx->bh_u8.ptr, u8.data(), 3 * nbh, hipMemcpyHostToDevice));
        SHQ_HIP(hipMemcpy(ctx->bh_vec.ptr, vec.data(), sizeof(double) * 15 * nbh, hipMemcpyHostToDevice));
    }
    ctx->nbh = (int64_t) nbh;
    ctx->have_bh_dyn = true;
    return SHQ_OK;
}

extern "C" int shq_bh_dynamics_download(shq_context *ctx, const shq_part_view *parts, const shq_bh_dyn_view *bh)
{
    SHQ_CHECK(ctx && parts && bh, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->have_bh_dyn && parts->numpart == ctx->numpart, SHQ_ERR_STATE, "bh_dynamics_download: no BH state on the device");
    SHQ_HIP(hipSetDevice(ctx->device));
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    const size_t nbh = (size_t) ctx->nbh;
    if(nbh == 0)
        return SHQ_OK;
    std::vector<int32_t> pidx(nbh);
    std::vector<uint8_t> u8(3 * nbh);
    SHQ_HIP(hipMemcpy(pidx.data(), ctx->bh_pidx.ptr, sizeof(int32_t) * nbh, hipMemcpyDeviceToHost));
    SHQ_HIP(hipMemcpy(u8.data(), ctx->bh_u8.ptr, 3 * nbh, hipMemcpyDeviceToHost));
    for(size_t k = 0; k < nbh; k++) {
        const int32_t pi = *cfield<int32_t>(parts->base, parts->elsize, pidx[k], parts->off_pi);
        SHQ_CHECK(pi >= 0 && pi < bh->numslots, SHQ_ERR_INVALID, "BH particle %d with PI %d outside the BH slot array", pidx[k], pi);
        if(bh->off_timebindynfric != SHQ_NOFIELD)
            *wfield<uint8_t>(bh->base, bh->elsize, pi, bh->off_timebindynfric) = u8[nbh + k];
        if(bh->off_jumptominpot != SHQ_NOFIELD)
            *wfield<char>(bh->base, bh->elsize, pi, bh->off_jumptominpot) = (char) u8[2 * nbh + k];
    }
    return SHQ_OK;
}

extern "C" int shq_kick_bh(shq_context *ctx, const double gravkick[SHQ_TIMEBINS + 1], const int32_t *active, int64_t nactive)
{
    SHQ_CHECK(ctx && gravkick, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->have_parts && ctx->vel.ptr && ctx->bin_hydro.ptr, SHQ_ERR_STATE, "kick_bh: upload particles, velocities and time bins first");
    SHQ_CHECK(ctx->have_bh_dyn, SHQ_ERR_STATE, "kick_bh: shq_bh_dynamics_upload first");
    SHQ_HIP(hipSetDevice(ctx->device));
    KickTab tab;
    memcpy(tab.k, gravkick, sizeof(tab.k));
    const int32_t *d_act = nullptr;
    int64_t nt = 0;
    SHQ_TRY(shq_resolve_active(ctx, active, nactive, ctx->numpart, &d_act, &nt));
    if(nt > 0 && ctx->nbh > 0) {
        kick_bh_kernel<<<dim3(nblk(nt)), dim3(256), 0, ctx->stream>>>((long long) nt, d_act, ctx->vel.ptr, ctx->pflags.ptr, ctx->bin_hydro.ptr,
                                                                     ctx->bh_pidx.ptr, ctx->nbh, ctx->bh_vec.ptr, ctx->bh_vec.ptr + 6 * ctx->nbh, tab);
        SHQ_HIP(hipGetLastError());
    }
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    return SHQ_OK;
}
