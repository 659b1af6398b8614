/* dynamics.hip — drift and kick of the resident particle set.
 *
 * With positions, velocities and accelerations resident in HBM, a time step does not need to cross
 * PCIe: drift -> shq_tree_build -> shq_pm_run / shq_grav_short_run -> kick.  These kernels restate
 *   real_drift_particle / drift_all_particles   libgadget/drift.cpp:16-99
 *   do_grav_short_range_kick, apply_half_kick (gravity part)   libgadget/timestep.cpp:838-872, 962-968
 *   apply_PM_half_kick                          libgadget/timestep.cpp:937-959
 * with the reference's operation order (no fma contraction), so the state stays bit-identical to a
 * host integration.  Not covered (need the BH / SPH slot arrays): black-hole repositioning
 * (drift.cpp:32-53) and do_hydro_kick.  The integer time line (Ti_drift, Ti_kick) stays with the host. */
#include "common.hpp"
#include <string.h>
#include <vector>

namespace {

inline unsigned nblk(long long n, int t = 256) { return (unsigned) ((n + t - 1) / t); }

struct Shift3 { double s[3]; };

__global__ void drift_kernel(long long n, double4 *posm, const double *__restrict__ vel, double *hsml, const double *__restrict__ dthsml,
                             const uint8_t *__restrict__ pflags, double ddrift, double Box, Shift3 sh, int *err)
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
        if((f >> 4) == 0 && hsml) { /* gas: Hsml prediction, drift.cpp:55-69 */
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
                                  const uint8_t *__restrict__ pflags, const uint8_t *__restrict__ bin_grav, KickTab tab)
{
#pragma clang fp contract(off)
    const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= nt)
        return;
    const long long i = targets ? (long long) targets[t] : t;
    if(pflags[i] & 3u)
        return;
    const double F = tab.k[bin_grav[i]];
    for(int j = 0; j < 3; j++)
        vel[3 * i + j] += accel[3 * i + j] * F;
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

} // namespace

extern "C" int shq_dynamics_upload(shq_context *ctx, const shq_part_view *parts)
{
    SHQ_CHECK(ctx && parts, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->have_parts && parts->numpart == ctx->numpart, SHQ_ERR_STATE, "dynamics_upload: upload the same particles first");
    SHQ_CHECK(parts->off_vel != SHQ_NOFIELD, SHQ_ERR_INVALID, "dynamics_upload: the particle view has no Vel");
    SHQ_HIP(hipSetDevice(ctx->device));
    const int64_t n = parts->numpart;
    const size_t cap = (size_t) (n > 0 ? n : 1);
    std::vector<double> vel(3 * cap), hsml(cap, 0.0), dth(cap, 0.0);
    std::vector<uint8_t> bg(cap, 0);
    int bad = 0;
    for(int64_t i = 0; i < n; i++) {
        const double *v = cfield<double>(parts, i, parts->off_vel);
        vel[3 * i] = v[0];
        vel[3 * i + 1] = v[1];
        vel[3 * i + 2] = v[2];
        if(parts->off_hsml != SHQ_NOFIELD)
            hsml[i] = *cfield<double>(parts, i, parts->off_hsml);
        if(parts->off_dthsml != SHQ_NOFIELD)
            dth[i] = *cfield<double>(parts, i, parts->off_dthsml);
        if(parts->off_timebin_gravity != SHQ_NOFIELD)
            bg[i] = *cfield<uint8_t>(parts, i, parts->off_timebin_gravity);
        if(bg[i] > SHQ_TIMEBINS)
            bad |= 1;
    }
    SHQ_CHECK(!bad, SHQ_ERR_INVALID, "time bin out of range (TIMEBINS = %d)", SHQ_TIMEBINS);
    SHQ_TRY(ctx->vel.reserve(3 * cap));
    SHQ_TRY(ctx->hsml.reserve(cap));
    SHQ_TRY(ctx->dthsml.reserve(cap));
    SHQ_TRY(ctx->bin_grav.reserve(cap));
    SHQ_TRY(ctx->pm_oob.reserve(4));
    if(n > 0) {
        SHQ_HIP(hipMemcpyAsync(ctx->vel.ptr, vel.data(), sizeof(double) * 3 * n, hipMemcpyHostToDevice, ctx->stream));
        SHQ_HIP(hipMemcpyAsync(ctx->hsml.ptr, hsml.data(), sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
        SHQ_HIP(hipMemcpyAsync(ctx->dthsml.ptr, dth.data(), sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
        SHQ_HIP(hipMemcpyAsync(ctx->bin_grav.ptr, bg.data(), n, hipMemcpyHostToDevice, ctx->stream));
    }
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    ctx->have_dyn = true;
    return SHQ_OK;
}

extern "C" int shq_drift(shq_context *ctx, double ddrift, double BoxSize, const double random_shift[3])
{
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
    if(n > 0) {
        drift_kernel<<<dim3(nblk(n)), dim3(256), 0, ctx->stream>>>(n, ctx->posm.ptr, ctx->vel.ptr, ctx->hsml.ptr, ctx->dthsml.ptr,
                                                                 ctx->pflags.ptr, ddrift, BoxSize, sh, d_err);
        SHQ_HIP(hipGetLastError());
    }
    int h_err = 0;
    SHQ_HIP(hipMemcpyAsync(&h_err, d_err, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    /* positions moved: the tree, its leaf-ordered copies and the PM result are stale */
    ctx->have_tree = false;
    ctx->tb_built = false;
    ctx->have_tree_targets = false;
    ctx->have_pm_result = false;
    SHQ_CHECK(h_err != 1, SHQ_ERR_INVALID, "drift: a gas particle reached Hsml <= 0 (drift.cpp:61-63)");
    SHQ_CHECK(h_err != 2, SHQ_ERR_INVALID, "drift: a particle position is not finite (drift.cpp:72-75)");
    return SHQ_OK;
}

extern "C" int shq_kick_short(shq_context *ctx, const double gravkick[SHQ_TIMEBINS + 1], const int32_t *active, int64_t nactive,
                              int from_accel_store)
{
    SHQ_CHECK(ctx && gravkick, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->have_parts && ctx->have_dyn, SHQ_ERR_STATE, "kick_short: shq_particles_upload and shq_dynamics_upload first");
    SHQ_CHECK(!active || nactive >= 0, SHQ_ERR_INVALID, "kick_short: bad active list");
    SHQ_HIP(hipSetDevice(ctx->device));
    const double *acc = from_accel_store ? ctx->acc.ptr : ctx->treeacc.ptr;
    SHQ_CHECK(acc, SHQ_ERR_STATE, "kick_short: no accelerations on the device yet");
    KickTab tab;
    memcpy(tab.k, gravkick, sizeof(tab.k));
    const int32_t *d_act = nullptr;
    long long nt = ctx->numpart;
    if(active) {
        nt = nactive;
        SHQ_TRY(ctx->active.reserve((size_t) (nactive > 0 ? nactive : 1)));
        if(nactive > 0)
            SHQ_HIP(hipMemcpyAsync(ctx->active.ptr, active, sizeof(int32_t) * nactive, hipMemcpyHostToDevice, ctx->stream));
        d_act = ctx->active.ptr;
    }
    if(nt > 0) {
        kick_short_kernel<<<dim3(nblk(nt)), dim3(256), 0, ctx->stream>>>(nt, d_act, ctx->vel.ptr, acc, ctx->pflags.ptr, ctx->bin_grav.ptr, tab);
        SHQ_HIP(hipGetLastError());
    }
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    return SHQ_OK;
}

extern "C" int shq_kick_pm(shq_context *ctx, double Fgravkick)
{
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
