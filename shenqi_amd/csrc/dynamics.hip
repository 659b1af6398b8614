/* dynamics.hip — drift and kick of the resident particle set.
 *
 * With positions, velocities and accelerations resident in HBM, a time step does not need to cross
 * PCIe: drift -> shq_tree_build -> shq_pm_run / shq_grav_short_run -> kick.  These kernels restate
 *   real_drift_particle / drift_all_particles   libgadget/drift.cpp:16-99
 *   do_grav_short_range_kick, apply_half_kick (gravity part)   libgadget/timestep.cpp:838-872, 962-968
 *   apply_PM_half_kick                          libgadget/timestep.cpp:937-959
 * with the reference's operation order (no fma contraction), so the state stays bit-identical to a
 * host integration.  Black-hole repositioning (drift.cpp:32-53) runs when the BH slot fields are resident
 * (shq_bh_dynamics_upload, timestep.hip) and shq_set_bh_reposition is on.  The per-particle loops of the integer time line
 * are in timestep.hip; Ti_drift / Ti_kick stay with the host. */
#include "common.hpp"
#include <string.h>
#include <cstring>
#include <vector>
#include <atomic>
#include <thread>
#include <algorithm>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

namespace {

inline unsigned nblk(long long n, int t = 256) { return (unsigned) ((n + t - 1) / t); }

struct Shift3 { double s[3]; };

struct BhJump { /* black-hole repositioning, drift.cpp:32-53; pidx == nullptr: off */
    const int32_t *pidx;
    long long nbh;
    uint8_t *jump;
    const double *minpotpos, *minpotvel;
};

__global__ void drift_kernel(long long n, double4 *posm, double *vel, double *hsml, const double *__restrict__ dthsml,
                             const uint8_t *__restrict__ pflags, double ddrift, double Box, Shift3 sh, BhJump bh, int *err)
{
#pragma clang fp contract(off)
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n)
        return;
    double4 p = posm[i];
    double x[3] = {p.x, p.y, p.z};
    const unsigned f = pflags[i];
    if(f & 3u) { /* garbage / swallowed: only follow the coordinate shift, drift.cpp:19-28 */
        for(int j = 0; j < 3; j++)
            x[j] += sh.s[j];
    } else {
        if((f >> 4) == 5 && bh.pidx) {
            const long long k = shq_bh_ordinal(bh.pidx, bh.nbh, (int32_t) i);
            if(k >= 0) {
                if(bh.jump[k]) {
                    for(int j = 0; j < 3; j++) {
                        const double d = x[j] - bh.minpotpos[3 * k + j];
                        const double dx = d > 0.5 * Box ? d - Box : (d < -0.5 * Box ? d + Box : d); /* NEAREST, partmanager.h:99 */
                        if(dx > 0.1 * Box)
                            *err = 3;
                        x[j] = bh.minpotpos[3 * k + j];
                        vel[3 * i + j] = bh.minpotvel[3 * k + j];
                    }
                }
                bh.jump[k] = 0;
            }
        } else if((f >> 4) == 0 && hsml) { /* gas: Hsml prediction, drift.cpp:55-69 */
            double h = hsml[i];
            h += dthsml[i] * ddrift;
            if(h <= 0)
                *err = 1;
            const double Maxhsml = Box / 2.;
            if(h > Maxhsml)
                h = Maxhsml;
            hsml[i] = h;
        }
        for(int j = 0; j < 3; j++) {
            x[j] += vel[3 * i + j] * ddrift + sh.s[j];
            if(!isfinite(x[j]))
                *err = 2;
        }
    }
    for(int j = 0; j < 3; j++) {
        int guard = 0; /* a finite position needs a handful of wraps; never spin on garbage input */
        while(x[j] > Box && guard++ < 64)
            x[j] -= Box;
        while(x[j] <= 0 && guard++ < 64)
            x[j] += Box;
        if(guard >= 64)
            *err = 2;
    }
    p.x = x[0];
    p.y = x[1];
    p.z = x[2];
    posm[i] = p;
}

struct KickTab { double k[SHQ_TIMEBINS + 1]; };

__global__ void kick_short_kernel(long long nt, const int32_t *__restrict__ targets, double *vel, const double *__restrict__ accel,
                                  const uint8_t *__restrict__ pflags, const uint8_t *__restrict__ bin_grav, KickTab tab, const int *walk_error)
{
#pragma clang fp contract(off)
    const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= nt)
        return;
    /* the sticky error word of the tree walk's pair kernel (grav_walk.hip): a walk that dropped pairs left incomplete accelerations, and
     * nothing is kicked with them - the host learns of it at its next synchronisation, the velocities stay as they were */
    if(walk_error && *walk_error != 0)
        return;
    const long long i = targets ? (long long) targets[t] : t;
    if(pflags[i] & 3u)
        return;
    const double F = tab.k[bin_grav[i]];
    for(int j = 0; j < 3; j++)
        vel[3 * i + j] += accel[3 * i + j] * F;
}

/* do_hydro_kick for gas (timestep.cpp:970-1003) over the active list of apply_half_kick / apply_hydro_half_kick (:875-886, :914-934) */
__global__ void kick_hydro_kernel(long long nt, const int32_t *__restrict__ targets, double *vel, const double *__restrict__ hacc,
                                  const double *__restrict__ dtent, double *entropy, const uint8_t *__restrict__ pflags,
                                  const uint8_t *__restrict__ bin_hydro, KickTab hk, KickTab dte, double atime, double MaxGasVel,
                                  unsigned long long *nlimited)
{
#pragma clang fp contract(off)
    const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= nt)
        return;
    const long long i = targets ? (long long) targets[t] : t;
    const unsigned f = pflags[i];
    if((f & 3u) || (f >> 4) != 0)
        return;
    const int bin = bin_hydro[i];
    const double F = hk.k[bin];
    double v[3];
    for(int j = 0; j < 3; j++)
        v[j] = vel[3 * i + j] + hacc[3 * i + j] * F;
    double vv = 0;
    for(int j = 0; j < 3; j++)
        vv += v[j] * v[j];
    vv = sqrt(vv);
    if(vv > 0 && vv / atime > MaxGasVel) {
        for(int j = 0; j < 3; j++)
            v[j] *= MaxGasVel * atime / vv;
        atomicAdd(nlimited, 1ull);
    }
    for(int j = 0; j < 3; j++)
        vel[3 * i + j] = v[j];
    entropy[i] += dtent[i] * dte.k[bin];
}

__global__ void kick_pm_kernel(long long n, double *vel, const double *__restrict__ gravpm, const uint8_t *__restrict__ pflags, double F)
{
#pragma clang fp contract(off)
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n || (pflags[i] & 3u))
        return;
    for(int j = 0; j < 3; j++)
        vel[3 * i + j] += gravpm[3 * i + j] * F;
}

template <typename T> const T *cfield(const shq_part_view *v, int64_t i, size_t off)
{
    return reinterpret_cast<const T *>(static_cast<const char *>(v->base) + (size_t) i * v->elsize + off);
}
template <typename T> T *wfield(const shq_part_view *v, int64_t i, size_t off)
{
    return reinterpret_cast<T *>(static_cast<char *>(v->base) + (size_t) i * v->elsize + off);
}

/* is_timebin_active, timestep.cpp:132-139 (dti_from_timebin(bin) = 1 << bin, timebinmgr.h:42-45) */
__device__ __forceinline__ bool timebin_active(int bin, long long Ti)
{
    if(bin <= 0 || Ti <= 0)
        return true;
    return (Ti & ((1ll << bin) - 1)) == 0; /* Ti > 0: Ti % 2^bin == 0 */
}

constexpr int ACT_NB = 6 * (SHQ_TIMEBINS + 1);

/* ActivePredicate (timestep.cpp:1265-1282) as a flag per particle, plus the TimeBinCountType / NumActiveGravity /
 * NumActiveHydro tallies of build_active_particles (timestep.cpp:1296-1343). counts: [ACT_NB] bins, then
 * nactivegrav, nactivehydro.  all != 0 is the PM-step branch: every particle is on the list, the tally skips
 * garbage / swallowed ones and nactivehydro counts every type-0/5 record. */
__global__ __launch_bounds__(256) void active_flag_kernel(long long n, const uint8_t *__restrict__ pflags, const uint8_t *__restrict__ bin_grav,
                                                           const uint8_t *__restrict__ bin_hydro, long long Ti, int all, uint8_t *flag,
                                                           unsigned long long *counts)
{
    __shared__ unsigned int hist[ACT_NB + 2];
    for(int k = threadIdx.x; k < ACT_NB + 2; k += blockDim.x)
        hist[k] = 0;
    __syncthreads();
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n) {
        const unsigned f = pflags[i];
        const int type = f >> 4;
        const bool dead = (f & 3u) != 0;
        const bool hydro_particle = type == 0 || type == 5;
        const int bg = bin_grav[i], bh = bin_hydro[i];
        const bool ga = timebin_active(bg, Ti);
        const bool on = all ? true : (!dead && (ga || (hydro_particle && timebin_active(bh, Ti))));
        flag[i] = on ? 1 : 0;
        if(all && hydro_particle)
            atomicAdd(&hist[ACT_NB + 1], 1u);
        if(on && !dead && type < 6) {
            if(!all) {
                if(ga)
                    atomicAdd(&hist[ACT_NB], 1u);
                if(hydro_particle)
                    atomicAdd(&hist[ACT_NB + 1], 1u);
            }
            atomicAdd(&hist[(SHQ_TIMEBINS + 1) * type + (hydro_particle ? bh : bg)], 1u);
        }
    }
    __syncthreads();
    for(int k = threadIdx.x; k < ACT_NB + 2; k += blockDim.x)
        if(hist[k])
            atomicAdd(&counts[k], (unsigned long long) hist[k]);
}

/* SubActivePredicate, timestep.cpp:1354-1371 */
struct SubActive {
    const uint8_t *pflags, *bin_grav;
    long long Ti;
    int maxtimebin;
    __device__ bool operator()(const int pi) const
    {
        const int bin = bin_grav[pi];
        if(pflags[pi] & 3u)
            return false;
        if(bin > maxtimebin)
            return false;
        return timebin_active(bin, Ti);
    }
};

template <typename In> int select_flagged(shq_context *ctx, In in, const uint8_t *flags, int32_t *out, size_t n, int64_t *count)
{
    size_t tmp = 0;
    size_t *d_count = reinterpret_cast<size_t *>(ctx->act_counts.ptr + ACT_NB + 2);
    SHQ_HIP(rocprim::select(nullptr, tmp, in, flags, out, d_count, n, ctx->stream));
    SHQ_TRY(ctx->act_temp.reserve(tmp + 16));
    SHQ_HIP(rocprim::select((void *) ctx->act_temp.ptr, tmp, in, flags, out, d_count, n, ctx->stream));
    size_t h = 0;
    SHQ_HIP(hipMemcpyAsync(&h, d_count, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    *count = (int64_t) h;
    return SHQ_OK;
}

template <typename In> int select_if(shq_context *ctx, In in, SubActive pred, int32_t *out, size_t n, int64_t *count)
{
    size_t tmp = 0;
    size_t *d_count = reinterpret_cast<size_t *>(ctx->act_counts.ptr + ACT_NB + 2);
    SHQ_HIP(rocprim::select(nullptr, tmp, in, out, d_count, n, pred, ctx->stream));
    SHQ_TRY(ctx->act_temp.reserve(tmp + 16));
    SHQ_HIP(rocprim::select((void *) ctx->act_temp.ptr, tmp, in, out, d_count, n, pred, ctx->stream));
    size_t h = 0;
    SHQ_HIP(hipMemcpyAsync(&h, d_count, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    *count = (int64_t) h;
    return SHQ_OK;
}

} // namespace

extern "C" int shq_dynamics_upload(shq_context *ctx, const shq_part_view *parts)
{
    SHQ_CHECK(ctx && parts, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->have_parts && parts->numpart == ctx->numpart, SHQ_ERR_STATE, "dynamics_upload: upload the same particles first");
    SHQ_CHECK(parts->off_vel != SHQ_NOFIELD, SHQ_ERR_INVALID, "dynamics_upload: the particle view has no Vel");
    SHQ_HIP(hipSetDevice(ctx->device));
    const int64_t n = parts->numpart;
    const size_t cap = (size_t) (n > 0 ? n : 1);
    /* packed by the host threads into the pinned staging buffer (5 doubles + 2 bytes per particle) */
    SHQ_TRY(ctx->stage.reserve(cap * (5 * sizeof(double) + 2) + 256));
    double *vel = reinterpret_cast<double *>(ctx->stage.ptr), *hsml = vel + 3 * cap, *dth = hsml + cap;
    uint8_t *bg = reinterpret_cast<uint8_t *>(dth + cap), *bh = bg + cap;
    std::atomic<int> badbin(0);
    {
        unsigned nt = std::thread::hardware_concurrency();
        nt = (nt == 0 || n < 65536) ? 1 : (nt > 32 ? 32 : nt);
        const int64_t chunk = (n + nt - 1) / nt;
        auto work = [&](int64_t lo, int64_t hi) {
            for(int64_t i = lo; i < hi; i++) {
                const double *v = cfield<double>(parts, i, parts->off_vel);
                vel[3 * i] = v[0];
                vel[3 * i + 1] = v[1];
                vel[3 * i + 2] = v[2];
                hsml[i] = parts->off_hsml != SHQ_NOFIELD ? *cfield<double>(parts, i, parts->off_hsml) : 0.0;
                dth[i] = parts->off_dthsml != SHQ_NOFIELD ? *cfield<double>(parts, i, parts->off_dthsml) : 0.0;
                bg[i] = parts->off_timebin_gravity != SHQ_NOFIELD ? *cfield<uint8_t>(parts, i, parts->off_timebin_gravity) : 0;
                bh[i] = parts->off_timebin_hydro != SHQ_NOFIELD ? *cfield<uint8_t>(parts, i, parts->off_timebin_hydro) : 0;
                if(bg[i] > SHQ_TIMEBINS || bh[i] > SHQ_TIMEBINS)
                    badbin.store(1);
            }
        };
        std::vector<std::thread> th;
        for(unsigned t = 1; t < nt; t++) {
            const int64_t lo = (int64_t) t * chunk, hi = std::min<int64_t>(n, lo + chunk);
            if(lo < hi)
                th.emplace_back(work, lo, hi);
        }
        work(0, std::min<int64_t>(n, chunk));
        for(auto &x : th)
            x.join();
    }
    const int bad = badbin.load();
    SHQ_CHECK(!bad, SHQ_ERR_INVALID, "time bin out of range (TIMEBINS = %d)", SHQ_TIMEBINS);
    SHQ_TRY(ctx->vel.reserve(3 * cap));
    SHQ_TRY(ctx->hsml.reserve(cap));
    SHQ_TRY(ctx->dthsml.reserve(cap));
    SHQ_TRY(ctx->bin_grav.reserve(cap));
    SHQ_TRY(ctx->bin_hydro.reserve(cap));
    SHQ_TRY(ctx->pm_oob.reserve(4));
    if(n > 0) {
        SHQ_HIP(hipMemcpyAsync(ctx->vel.ptr, vel, sizeof(double) * 3 * n, hipMemcpyHostToDevice, ctx->stream));
        SHQ_HIP(hipMemcpyAsync(ctx->hsml.ptr, hsml, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
        SHQ_HIP(hipMemcpyAsync(ctx->dthsml.ptr, dth, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
        SHQ_HIP(hipMemcpyAsync(ctx->bin_grav.ptr, bg, (size_t) n, hipMemcpyHostToDevice, ctx->stream));
        SHQ_HIP(hipMemcpyAsync(ctx->bin_hydro.ptr, bh, (size_t) n, hipMemcpyHostToDevice, ctx->stream));
    }
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    ctx->have_dyn = true;
    ctx->n_act = ctx->n_sub = -1;
    return SHQ_OK;
}

int shq_resolve_active(shq_context *ctx, const int32_t *active, int64_t nactive, int64_t n_all, const int32_t **d_active, int64_t *nt)
{
    *d_active = nullptr;
    *nt = n_all;
    if(!active)
        return SHQ_OK;
    if(active == SHQ_SPH_QUEUE_RESIDENT) {
        SHQ_CHECK(ctx->sphrun.phase != 0, SHQ_ERR_STATE, "no SPH walk is open");
        *d_active = ctx->sphrun.cur;
        *nt = ctx->sphrun.size;
        return SHQ_OK;
    }
    if(active == SHQ_ACTIVE_RESIDENT || active == SHQ_SUBLIST_RESIDENT) {
        const bool sub = active == SHQ_SUBLIST_RESIDENT;
        SHQ_CHECK((sub ? ctx->n_sub : ctx->n_act) >= 0, SHQ_ERR_STATE, "no resident active %s: call shq_build_active_%s first",
                  sub ? "sub-list" : "list", sub ? "sublist" : "particles");
        if(sub) {
            *d_active = ctx->act_sub.ptr;
            *nt = ctx->n_sub;
        } else if(!ctx->act_all) {
            *d_active = ctx->act_list.ptr;
            *nt = ctx->n_act;
        } else
            *nt = std::min<int64_t>(n_all, ctx->n_act); /* PM step: ActiveParticle == NULL, everything is active */
        return SHQ_OK;
    }
    SHQ_CHECK(nactive >= 0, SHQ_ERR_INVALID, "bad active list length %ld", (long) nactive);
    for(int64_t k = 0; k < nactive; k++)
        SHQ_CHECK(active[k] >= (ctx->allow_padding ? -1 : 0) && active[k] < ctx->numpart, SHQ_ERR_INVALID, "active[%ld] = %d out of range", (long) k, active[k]);
    SHQ_TRY(ctx->active.reserve((size_t) std::max<int64_t>(nactive, 1)));
    if(nactive > 0)
        SHQ_HIP(hipMemcpyAsync(ctx->active.ptr, active, sizeof(int32_t) * nactive, hipMemcpyHostToDevice, ctx->stream));
    *d_active = ctx->active.ptr;
    *nt = nactive;
    return SHQ_OK;
}

extern "C" int shq_timebins_upload(shq_context *ctx, const uint8_t *bin_gravity, const uint8_t *bin_hydro)
{
    SHQ_CHECK(ctx, SHQ_ERR_INVALID, "null context");
    SHQ_CHECK(ctx->have_parts, SHQ_ERR_STATE, "timebins_upload: upload particles first");
    SHQ_HIP(hipSetDevice(ctx->device));
    const int64_t n = ctx->numpart;
    const size_t cap = (size_t) (n > 0 ? n : 1);
    for(int w = 0; w < 2; w++) {
        const uint8_t *src = w ? bin_hydro : bin_gravity;
        DevBuf<uint8_t> &dst = w ? ctx->bin_hydro : ctx->bin_grav;
        const bool fresh = dst.ptr == nullptr || dst.cap < cap;
        if(!src && !fresh)
            continue;
        if(src)
            for(int64_t i = 0; i < n; i++)
                SHQ_CHECK(src[i] <= SHQ_TIMEBINS, SHQ_ERR_INVALID, "time bin %d of particle %ld out of range (TIMEBINS = %d)", src[i], (long) i, SHQ_TIMEBINS);
        SHQ_TRY(dst.reserve(cap));
        if(src && n > 0)
            SHQ_HIP(hipMemcpyAsync(dst.ptr, src, (size_t) n, hipMemcpyHostToDevice, ctx->stream));
        else
            SHQ_HIP(hipMemsetAsync(dst.ptr, 0, cap, ctx->stream));
    }
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    ctx->n_act = ctx->n_sub = -1;
    return SHQ_OK;
}

extern "C" int shq_build_active_particles(shq_context *ctx, int64_t Ti_Current, int is_pm_step, shq_active_info *info)
{
    SHQ_CHECK(ctx, SHQ_ERR_INVALID, "null context");
    SHQ_CHECK(ctx->have_parts && ctx->bin_grav.ptr && ctx->bin_hydro.ptr, SHQ_ERR_STATE,
              "build_active_particles: upload particles and time bins (shq_dynamics_upload / shq_timebins_upload) first");
    SHQ_HIP(hipSetDevice(ctx->device));
    const int64_t n = ctx->numpart;
    SHQ_CHECK(n < (1ll << 31) - 64, SHQ_ERR_INVALID, "build_active_particles: too many particles");
    const size_t cap = (size_t) (n > 0 ? n : 1);
    SHQ_TRY(ctx->act_counts.reserve(ACT_NB + 4));
    SHQ_TRY(ctx->act_list.reserve(cap));
    SHQ_TRY(ctx->act_flag.reserve(cap));
    SHQ_HIP(hipMemsetAsync(ctx->act_counts.ptr, 0, sizeof(unsigned long long) * (ACT_NB + 4), ctx->stream));
    ctx->n_act = ctx->n_sub = -1;
    if(n > 0) {
        active_flag_kernel<<<dim3(nblk(n)), dim3(256), 0, ctx->stream>>>(n, ctx->pflags.ptr, ctx->bin_grav.ptr, ctx->bin_hydro.ptr,
                                                                        (long long) Ti_Current, is_pm_step ? 1 : 0, ctx->act_flag.ptr,
                                                                        ctx->act_counts.ptr);
        SHQ_HIP(hipGetLastError());
    }
    int64_t nact = n;
    if(!is_pm_step) {
        nact = 0;
        if(n > 0)
            SHQ_TRY(select_flagged(ctx, rocprim::counting_iterator<int32_t>(0), (const uint8_t *) ctx->act_flag.ptr, ctx->act_list.ptr, (size_t) n, &nact));
    }
    unsigned long long h[ACT_NB + 2];
    SHQ_HIP(hipMemcpyAsync(h, ctx->act_counts.ptr, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    ctx->act_all = is_pm_step != 0;
    ctx->n_act = nact;
    if(info) {
        info->NumActiveParticle = nact;
        info->NumActiveGravity = is_pm_step ? n : (int64_t) h[ACT_NB];
        info->NumActiveHydro = (int64_t) h[ACT_NB + 1];
        for(int k = 0; k < ACT_NB; k++)
            info->TimeBinCountType[k] = (int64_t) h[k];
    }
    return SHQ_OK;
}

extern "C" int shq_build_active_sublist(shq_context *ctx, int maxtimebin, int64_t Ti_Current, int64_t *nsub)
{
    SHQ_CHECK(ctx, SHQ_ERR_INVALID, "null context");
    SHQ_CHECK(ctx->n_act >= 0, SHQ_ERR_STATE, "build_active_sublist: call shq_build_active_particles first");
    SHQ_HIP(hipSetDevice(ctx->device));
    const int64_t n = ctx->n_act;
    SHQ_TRY(ctx->act_sub.reserve((size_t) (n > 0 ? n : 1)));
    const SubActive pred{ctx->pflags.ptr, ctx->bin_grav.ptr, (long long) Ti_Current, maxtimebin};
    int64_t cnt = 0;
    ctx->n_sub = -1;
    if(n > 0) {
        if(ctx->act_all)
            SHQ_TRY(select_if(ctx, rocprim::counting_iterator<int32_t>(0), pred, ctx->act_sub.ptr, (size_t) n, &cnt));
        else
            SHQ_TRY(select_if(ctx, (const int32_t *) ctx->act_list.ptr, pred, ctx->act_sub.ptr, (size_t) n, &cnt));
    }
    ctx->n_sub = cnt;
    if(nsub)
        *nsub = cnt;
    return SHQ_OK;
}

extern "C" int shq_active_download(shq_context *ctx, int sublist, int32_t *list, int64_t capacity, int64_t *count)
{
    SHQ_CHECK(ctx, SHQ_ERR_INVALID, "null context");
    const int64_t n = sublist ? ctx->n_sub : ctx->n_act;
    SHQ_CHECK(n >= 0, SHQ_ERR_STATE, "active_download: no resident list");
    if(count)
        *count = n;
    if(!list)
        return SHQ_OK;
    SHQ_CHECK(capacity >= n, SHQ_ERR_INVALID, "active_download: capacity %ld < %ld", (long) capacity, (long) n);
    SHQ_HIP(hipSetDevice(ctx->device));
    if(!sublist && ctx->act_all) { /* the reference keeps ActiveParticle == NULL here; hand out the identity */
        for(int64_t i = 0; i < n; i++)
            list[i] = (int32_t) i;
        return SHQ_OK;
    }
    if(n > 0)
        SHQ_HIP(hipMemcpy(list, sublist ? ctx->act_sub.ptr : ctx->act_list.ptr, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    return SHQ_OK;
}

extern "C" int shq_drift(shq_context *ctx, double ddrift, double BoxSize, const double random_shift[3])
{
    if(ctx)
        ctx->inputs_current = 0; /* the resident set moves: shq_set_inputs_current ends here */
    SHQ_CHECK(ctx, SHQ_ERR_INVALID, "null context");
    SHQ_CHECK(ctx->have_parts && ctx->have_dyn, SHQ_ERR_STATE, "drift: shq_particles_upload and shq_dynamics_upload first");
    SHQ_CHECK(BoxSize > 0, SHQ_ERR_INVALID, "drift: BoxSize must be > 0");
    SHQ_HIP(hipSetDevice(ctx->device));
    SHQ_TRY(shq_join_pm(ctx));
    const long long n = ctx->numpart;
    Shift3 sh;
    for(int j = 0; j < 3; j++)
        sh.s[j] = random_shift ? random_shift[j] : 0.0;
    int *d_err = ctx->pm_oob.ptr + 1;
    SHQ_HIP(hipMemsetAsync(d_err, 0, sizeof(int), ctx->stream));
    BhJump bh = {nullptr, 0, nullptr, nullptr, nullptr};
    if(ctx->bh_reposition && ctx->have_bh_dyn && ctx->nbh > 0)
        bh = BhJump{ctx->bh_pidx.ptr, ctx->nbh, ctx->bh_u8.ptr + 2 * ctx->nbh, ctx->bh_vec.ptr + 9 * ctx->nbh, ctx->bh_vec.ptr + 12 * ctx->nbh};
    if(n > 0) {
        drift_kernel<<<dim3(nblk(n)), dim3(256), 0, ctx->stream>>>(n, ctx->posm.ptr, ctx->vel.ptr, ctx->hsml.ptr, ctx->dthsml.ptr,
                                                                 ctx->pflags.ptr, ddrift, BoxSize, sh, bh, d_err);
        SHQ_HIP(hipGetLastError());
    }
    int h_err = 0;
    SHQ_HIP(hipMemcpyAsync(&h_err, d_err, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    /* positions moved: the tree, its leaf-ordered copies and the PM result are stale */
    ctx->have_tree = false;
    ctx->have_toptree = false;
    ctx->tb_built = false;
    /* (the tree-order target list survives a drift: it is an order of the same particles, refreshed by the tree builds every few steps) */
    ctx->have_pm_result = false;
    ctx->pm_prestarted = false;
    SHQ_CHECK(h_err != 1, SHQ_ERR_INVALID, "drift: a gas particle reached Hsml <= 0 (drift.cpp:61-63)");
    SHQ_CHECK(h_err != 2, SHQ_ERR_INVALID, "drift: a particle position is not finite (drift.cpp:72-75)");
    SHQ_CHECK(h_err != 3, SHQ_ERR_INVALID, "drift: a black hole would jump further than 0.1 BoxSize to its potential minimum (drift.cpp:40-48)");
    return SHQ_OK;
}

extern "C" int shq_kick_short(shq_context *ctx, const double gravkick[SHQ_TIMEBINS + 1], const int32_t *active, int64_t nactive,
                              int from_accel_store)
{
    if(ctx)
        ctx->inputs_current = 0; /* the resident set moves: shq_set_inputs_current ends here */
    SHQ_CHECK(ctx && gravkick, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->have_parts && ctx->have_dyn, SHQ_ERR_STATE, "kick_short: shq_particles_upload and shq_dynamics_upload first");
    SHQ_CHECK(!active || nactive >= 0, SHQ_ERR_INVALID, "kick_short: bad active list");
    SHQ_HIP(hipSetDevice(ctx->device));
    const double *acc = from_accel_store ? ctx->acc.ptr : ctx->treeacc.ptr;
    SHQ_CHECK(acc, SHQ_ERR_STATE, "kick_short: no accelerations on the device yet");
    /* no kick on the accelerations of a walk whose pair kernel dropped pairs: the kernel below reads the sticky error word on the
     * device and leaves the velocities alone when it is up (a walk still in flight on the stream is covered too); here, what is known
     * on the host so far.  The call does not wait for the stream (round 4: a resident step has no host round trip between walk and kick). */
    SHQ_TRY(shq_walk_check_status(ctx, false));
    KickTab tab;
    memcpy(tab.k, gravkick, sizeof(tab.k));
    const int32_t *d_act = nullptr;
    int64_t nt = 0;
    SHQ_TRY(shq_resolve_active(ctx, active, nactive, ctx->numpart, &d_act, &nt));
    if(nt > 0) {
        kick_short_kernel<<<dim3(nblk(nt)), dim3(256), 0, ctx->stream>>>((long long) nt, d_act, ctx->vel.ptr, acc, ctx->pflags.ptr, ctx->bin_grav.ptr, tab,
                                                                        shq_walk_error_word(ctx));
        SHQ_HIP(hipGetLastError());
    }
    return SHQ_OK;
}

extern "C" int shq_kick_hydro(shq_context *ctx, const double hydrokick[SHQ_TIMEBINS + 1], const double dt_entr[SHQ_TIMEBINS + 1], double atime,
                              double MaxGasVel, const int32_t *active, int64_t nactive, int from_hydro_output, int64_t *nlimited)
{
    if(ctx)
        ctx->inputs_current = 0; /* the resident set moves: shq_set_inputs_current ends here */
    SHQ_CHECK(ctx && hydrokick && dt_entr, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->have_parts && ctx->have_sph && ctx->vel.ptr && ctx->g_entropy.ptr, SHQ_ERR_STATE,
              "kick_hydro: no SPH state on the device (shq_density / shq_hydro_force, or their phase calls, load it)");
    SHQ_CHECK(ctx->sphrun.phase == 0, SHQ_ERR_STATE, "kick_hydro: an SPH walk is still open");
    SHQ_CHECK(atime > 0 && MaxGasVel > 0, SHQ_ERR_INVALID, "kick_hydro: atime and MaxGasVel must be > 0");
    SHQ_HIP(hipSetDevice(ctx->device));
    const double *hacc = from_hydro_output ? ctx->g_hydroaccel_out.ptr : ctx->g_hydroaccel.ptr;
    const double *dtent = from_hydro_output ? ctx->g_dtentropy_out.ptr : ctx->g_dtentropy.ptr;
    SHQ_CHECK(hacc && dtent, SHQ_ERR_STATE, "kick_hydro: no hydro accelerations on the device yet");
    KickTab hk, dte;
    memcpy(hk.k, hydrokick, sizeof(hk.k));
    memcpy(dte.k, dt_entr, sizeof(dte.k));
    const int32_t *d_act = nullptr;
    int64_t nt = 0;
    SHQ_TRY(shq_resolve_active(ctx, active, nactive, ctx->numpart, &d_act, &nt));
    SHQ_TRY(ctx->act_counts.reserve(6 * (SHQ_TIMEBINS + 1) + 4));
    unsigned long long *d_n = ctx->act_counts.ptr;
    SHQ_HIP(hipMemsetAsync(d_n, 0, sizeof(unsigned long long), ctx->stream));
    if(nt > 0) {
        kick_hydro_kernel<<<dim3(nblk(nt)), dim3(256), 0, ctx->stream>>>((long long) nt, d_act, ctx->vel.ptr, hacc, dtent, ctx->g_entropy.ptr,
                                                                        ctx->pflags.ptr, ctx->bin_hydro.ptr, hk, dte, atime, MaxGasVel, d_n);
        SHQ_HIP(hipGetLastError());
    }
    unsigned long long h = 0;
    SHQ_HIP(hipMemcpyAsync(&h, d_n, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    if(nlimited)
        *nlimited = (int64_t) h;
    return SHQ_OK;
}

extern "C" int shq_entropy_download(shq_context *ctx, double *entropy_by_particle)
{
    SHQ_CHECK(ctx && entropy_by_particle, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->have_parts && ctx->have_sph && ctx->g_entropy.ptr, SHQ_ERR_STATE, "entropy_download: no SPH state on the device");
    SHQ_HIP(hipSetDevice(ctx->device));
    if(ctx->numpart > 0)
        SHQ_HIP(hipMemcpy(entropy_by_particle, ctx->g_entropy.ptr, sizeof(double) * ctx->numpart, hipMemcpyDeviceToHost));
    return SHQ_OK;
}

extern "C" int shq_kick_pm(shq_context *ctx, double Fgravkick)
{
    if(ctx)
        ctx->inputs_current = 0; /* the resident set moves: shq_set_inputs_current ends here */
    SHQ_CHECK(ctx, SHQ_ERR_INVALID, "null context");
    SHQ_CHECK(ctx->have_parts && ctx->have_dyn, SHQ_ERR_STATE, "kick_pm: shq_particles_upload and shq_dynamics_upload first");
    SHQ_HIP(hipSetDevice(ctx->device));
    SHQ_TRY(shq_join_pm(ctx));
    const long long n = ctx->numpart;
    if(n > 0) {
        kick_pm_kernel<<<dim3(nblk(n)), dim3(256), 0, ctx->stream>>>(n, ctx->vel.ptr, ctx->gravpm.ptr, ctx->pflags.ptr, Fgravkick);
        SHQ_HIP(hipGetLastError());
    }
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    return SHQ_OK;
}

extern "C" int shq_dynamics_download(shq_context *ctx, const shq_part_view *parts)
{
    SHQ_CHECK(ctx && parts && parts->base, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->have_parts && ctx->have_dyn && parts->numpart == ctx->numpart, SHQ_ERR_STATE, "dynamics_download: nothing resident for this view");
    SHQ_CHECK(parts->off_pos != SHQ_NOFIELD && parts->off_vel != SHQ_NOFIELD, SHQ_ERR_INVALID, "dynamics_download: the view needs Pos and Vel");
    SHQ_HIP(hipSetDevice(ctx->device));
    SHQ_TRY(shq_join_pm(ctx));
    const int64_t n = parts->numpart;
    const size_t cap = (size_t) (n > 0 ? n : 1);
    std::vector<double4> posm(cap);
    std::vector<double> vel(3 * cap), hsml(cap);
    if(n > 0) {
        SHQ_HIP(hipMemcpyAsync(posm.data(), ctx->posm.ptr, sizeof(double4) * n, hipMemcpyDeviceToHost, ctx->stream));
        SHQ_HIP(hipMemcpyAsync(vel.data(), ctx->vel.ptr, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, ctx->stream));
        SHQ_HIP(hipMemcpyAsync(hsml.data(), ctx->hsml.ptr, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    }
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    for(int64_t i = 0; i < n; i++) {
        double *x = wfield<double>(parts, i, parts->off_pos);
        x[0] = posm[i].x;
        x[1] = posm[i].y;
        x[2] = posm[i].z;
        double *v = wfield<double>(parts, i, parts->off_vel);
        v[0] = vel[3 * i];
        v[1] = vel[3 * i + 1];
        v[2] = vel[3 * i + 2];
        if(parts->off_hsml != SHQ_NOFIELD)
            *wfield<double>(parts, i, parts->off_hsml) = hsml[i];
    }
    return SHQ_OK;
}
