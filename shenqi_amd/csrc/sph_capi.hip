/* sph_capi.hip — extern "C" entry points for SPH density / hydro (see include/shenqi_hip.h). */
#include "common.hpp"
#include <string.h>
#include <algorithm>
#include <atomic>
#include <thread>

namespace {

template <typename T> inline const T *pfield(const shq_part_view *v, int64_t i, size_t off)
{
    return reinterpret_cast<const T *>(static_cast<const char *>(v->base) + (size_t) i * v->elsize + off);
}
template <typename T> inline T *pfield_w(const shq_part_view *v, int64_t i, size_t off)
{
    return reinterpret_cast<T *>(static_cast<char *>(v->base) + (size_t) i * v->elsize + off);
}
inline double *sfield(const shq_sph_view *v, int64_t slot, size_t off)
{
    return reinterpret_cast<double *>(static_cast<char *>(v->base) + (size_t) slot * v->elsize + off);
}

template <typename T> int up(shq_context *ctx, DevBuf<T> &b, const std::vector<T> &h)
{
    SHQ_TRY(b.reserve(std::max<size_t>(h.size(), 1)));
    if(!h.empty())
        SHQ_HIP(hipMemcpyAsync(b.ptr, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice, ctx->stream));
    return SHQ_OK;
}
template <typename T> int down(shq_context *ctx, const DevBuf<T> &b, std::vector<T> &h, size_t n)
{
    h.resize(std::max<size_t>(n, 1));
    if(n > 0)
        SHQ_HIP(hipMemcpyAsync(h.data(), b.ptr, sizeof(T) * n, hipMemcpyDeviceToHost, ctx->stream));
    return SHQ_OK;
}

/* the particle IDs the sub-grid walks compare and record (bhw_ids); skipped when the caller vouches for the copy */
int ids_upload(shq_context *ctx, const uint64_t *ids, int64_t n)
{
    SHQ_TRY(ctx->bhw_ids.reserve((size_t) std::max<int64_t>(n, 1)));
    if((ctx->inputs_current & SHQ_CURRENT_IDS) && ctx->cur_ids == (const void *) ids && ctx->cur_ids_n == n)
        return SHQ_OK;
    if(n > 0)
        SHQ_HIP(hipMemcpyAsync(ctx->bhw_ids.ptr, ids, sizeof(uint64_t) * (size_t) n, hipMemcpyHostToDevice, ctx->stream));
    ctx->cur_ids = ids;
    ctx->cur_ids_n = n;
    return SHQ_OK;
}

/* Gather the SPH state into per-particle-index arrays (gas fields come from slot PI). */
int sph_upload(shq_context *ctx, const shq_part_view *parts, const shq_sph_view *sph)
{
    SHQ_CHECK(parts->off_hsml != SHQ_NOFIELD && parts->off_vel != SHQ_NOFIELD && parts->off_pi != SHQ_NOFIELD &&
                  parts->off_type != SHQ_NOFIELD, SHQ_ERR_INVALID, "SPH needs Hsml, Vel, PI and Type in the particle view");
    SHQ_CHECK(sph && (sph->numslots == 0 || sph->base), SHQ_ERR_INVALID, "SPH slot view is NULL");
    const int64_t n = parts->numpart;
    if((ctx->inputs_current & SHQ_CURRENT_SPH) && ctx->have_sph && ctx->cur_sph == sph->base && ctx->cur_sph_n == sph->numslots &&
       ctx->cur_parts == parts->base && ctx->numpart == n)
        return SHQ_OK; /* shq_set_inputs_current */
    /* one parallel pass into pinned staging (17 doubles + 2 bytes per particle), then one copy per array */
    const size_t cap = (size_t) std::max<int64_t>(n, 1);
    SHQ_TRY(ctx->stage.reserve(cap * (17 * sizeof(double) + 2) + 256));
    double *hsml = reinterpret_cast<double *>(ctx->stage.ptr);
    double *vel = hsml + cap, *entropy = vel + 3 * cap, *dtentropy = entropy + cap, *hacc = dtentropy + cap, *delay = hacc + 3 * cap;
    double *density = delay + cap, *egywt = density + cap, *dhsml = egywt + cap, *divvel = dhsml + cap, *curl = divvel + cap;
    double *dth = curl + cap, *msv = dth + cap;
    uint8_t *bg = reinterpret_cast<uint8_t *>(msv + cap), *bh = bg + cap;
    std::atomic<int> bad(0);
    {
        unsigned nt = std::thread::hardware_concurrency();
        nt = nt == 0 ? 1 : (nt > 32 ? 32 : nt);
        if(n < 65536)
            nt = 1;
        const int64_t chunk = (n + nt - 1) / nt;
        auto work = [&](int64_t lo, int64_t hi) {
            for(int64_t i = lo; i < hi; i++) {
                hsml[i] = *pfield<double>(parts, i, parts->off_hsml);
                const double *v = pfield<double>(parts, i, parts->off_vel);
                vel[3 * i] = v[0]; vel[3 * i + 1] = v[1]; vel[3 * i + 2] = v[2];
                bg[i] = parts->off_timebin_gravity != SHQ_NOFIELD ? *pfield<uint8_t>(parts, i, parts->off_timebin_gravity) : 0;
                bh[i] = parts->off_timebin_hydro != SHQ_NOFIELD ? *pfield<uint8_t>(parts, i, parts->off_timebin_hydro) : 0;
                if(bg[i] > SHQ_TIMEBINS || bh[i] > SHQ_TIMEBINS)
                    bad.store(1);
                dth[i] = parts->off_dthsml != SHQ_NOFIELD ? *pfield<double>(parts, i, parts->off_dthsml) : 0.0;
                double e = 0, de = 0, h0 = 0, h1 = 0, h2 = 0, dl = 0, rho = 0, eg = 0, dh = 0, dv = 0, cv = 0, ms = 0;
                if(*pfield<uint8_t>(parts, i, parts->off_type) == 0) {
                    const int32_t pi = *pfield<int32_t>(parts, i, parts->off_pi);
                    if(pi < 0 || pi >= sph->numslots)
                        bad.store(2);
                    else {
                        e = *sfield(sph, pi, sph->off_entropy);
                        de = *sfield(sph, pi, sph->off_dtentropy);
                        const double *ha = sfield(sph, pi, sph->off_hydroaccel);
                        h0 = ha[0]; h1 = ha[1]; h2 = ha[2];
                        if(sph->off_delaytime != SHQ_NOFIELD)
                            dl = *sfield(sph, pi, sph->off_delaytime);
                        rho = *sfield(sph, pi, sph->off_density);
                        eg = *sfield(sph, pi, sph->off_egywtdensity);
                        dh = *sfield(sph, pi, sph->off_dhsmlegydensityfactor);
                        dv = *sfield(sph, pi, sph->off_divvel);
                        cv = *sfield(sph, pi, sph->off_curlvel);
                        if(sph->off_maxsignalvel != SHQ_NOFIELD)
                            ms = *sfield(sph, pi, sph->off_maxsignalvel);
                    }
                }
                msv[i] = ms;
                entropy[i] = e; dtentropy[i] = de;
                hacc[3 * i] = h0; hacc[3 * i + 1] = h1; hacc[3 * i + 2] = h2;
                delay[i] = dl; density[i] = rho; egywt[i] = eg; dhsml[i] = dh; divvel[i] = dv; curl[i] = cv;
            }
        };
        std::vector<std::thread> th;
        for(unsigned t = 1; t < nt; t++) {
            const int64_t lo = (int64_t) t * chunk, hi = std::min(n, lo + chunk);
            if(lo < hi)
                th.emplace_back(work, lo, hi);
        }
        work(0, std::min(n, chunk));
        for(auto &x : th)
            x.join();
    }
    SHQ_CHECK(bad.load() != 1, SHQ_ERR_INVALID, "time bin out of range (TIMEBINS = %d)", SHQ_TIMEBINS);
    SHQ_CHECK(bad.load() != 2, SHQ_ERR_INVALID, "gas particle with PI outside the SPH slot array");
    auto upd = [&](DevBuf<double> &b, const double *h, size_t cnt) -> int {
        SHQ_TRY(b.reserve(std::max<size_t>(cnt, 1)));
        if(cnt > 0)
            SHQ_HIP(hipMemcpyAsync(b.ptr, h, sizeof(double) * cnt, hipMemcpyHostToDevice, ctx->stream));
        return SHQ_OK;
    };
    auto upb = [&](DevBuf<uint8_t> &b, const uint8_t *h, size_t cnt) -> int {
        SHQ_TRY(b.reserve(std::max<size_t>(cnt, 1)));
        if(cnt > 0)
            SHQ_HIP(hipMemcpyAsync(b.ptr, h, cnt, hipMemcpyHostToDevice, ctx->stream));
        return SHQ_OK;
    };
    SHQ_TRY(upd(ctx->hsml, hsml, n));
    SHQ_TRY(upd(ctx->vel, vel, 3 * n));
    SHQ_TRY(upb(ctx->bin_grav, bg, n));
    SHQ_TRY(upb(ctx->bin_hydro, bh, n));
    SHQ_TRY(upd(ctx->g_entropy, entropy, n));
    SHQ_TRY(upd(ctx->g_dtentropy, dtentropy, n));
    SHQ_TRY(upd(ctx->g_hydroaccel, hacc, 3 * n));
    SHQ_TRY(upd(ctx->g_delaytime, delay, n));
    SHQ_TRY(upd(ctx->g_density, density, n));
    SHQ_TRY(upd(ctx->g_egywt, egywt, n));
    SHQ_TRY(upd(ctx->g_dhsmlegy, dhsml, n));
    SHQ_TRY(upd(ctx->g_divvel, divvel, n));
    SHQ_TRY(upd(ctx->g_curlvel, curl, n));
    SHQ_TRY(upd(ctx->dthsml, dth, n));          /* Part[].DtHsml and SphP[].MaxSignalVel: what the time-step criteria read */
    SHQ_TRY(upd(ctx->g_maxsignalvel, msv, n));
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    ctx->have_sph = true;
    ctx->cur_sph = sph->base;
    ctx->cur_sph_n = sph->numslots;
    return SHQ_OK;
}

/* TreeWalkQueryBase::haswork + Density/HydroQuery::haswork: not garbage, not swallowed, type mask */
std::vector<int32_t> build_queue(const shq_part_view *parts, const int32_t *active, int64_t nactive, bool with_bh)
{
    std::vector<int32_t> q;
    const int64_t nloop = active ? nactive : parts->numpart;
    q.reserve((size_t) nloop);
    for(int64_t k = 0; k < nloop; k++) {
        const int32_t i = active ? active[k] : (int32_t) k;
        if(parts->off_flags != SHQ_NOFIELD && (*pfield<uint32_t>(parts, i, parts->off_flags) & 3u))
            continue;
        const uint8_t t = *pfield<uint8_t>(parts, i, parts->off_type);
        if(t == 0 || (with_bh && t == 5))
            q.push_back(i);
    }
    return q;
}

} // namespace

extern "C" int shq_sph_state_upload(shq_context *ctx, const shq_part_view *parts, const shq_sph_view *sph)
{
    SHQ_CHECK(ctx && parts && sph, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->have_parts && ctx->numpart == parts->numpart, SHQ_ERR_STATE, "sph_state_upload: upload these particles first (%ld resident, %ld in the view)",
              (long) (ctx->have_parts ? ctx->numpart : -1), (long) parts->numpart);
    SHQ_HIP(hipSetDevice(ctx->device));
    return sph_upload(ctx, parts, sph);
}


/* the host copy of the work queue of the open walk: close() assigns results only to the walked targets */
static std::vector<int32_t> &run_queue(shq_context *ctx) { return ctx->sph_queue_host; }

extern "C" int shq_density_open(shq_context *ctx, const shq_tree_view *tree, const shq_part_view *parts, const shq_sph_view *sph,
                                const shq_bh_view *bh, const int32_t *active, int64_t nactive, const shq_density_params *params,
                                int want_gradrho, int64_t *nqueue)
{
    SHQ_CHECK(ctx && tree && parts && sph && params, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->sphrun.phase == 0, SHQ_ERR_STATE, "density_open: another SPH walk is open");
    SHQ_HIP(hipSetDevice(ctx->device));
    const int64_t n = parts->numpart;
    if(active)
        for(int64_t k = 0; k < nactive; k++)
            SHQ_CHECK(active[k] >= 0 && active[k] < n, SHQ_ERR_INVALID, "active[%ld] = %d out of range", (long) k, active[k]);
    SHQ_TRY(shq_particles_upload(ctx, parts));
    SHQ_TRY(sph_upload(ctx, parts, sph));
    SHQ_TRY(shq_tree_upload(ctx, tree));
    std::vector<int32_t> &queue = run_queue(ctx);
    queue = build_queue(parts, active, nactive, true);
    for(int32_t i : queue) {
        if(*pfield<uint8_t>(parts, i, parts->off_type) == 5)
            SHQ_CHECK(bh && bh->base, SHQ_ERR_INVALID, "black-hole density target but no BH slot view");
    }
    const int64_t nq = (int64_t) queue.size();
    SHQ_TRY(ctx->s_queue0.reserve((size_t) std::max<int64_t>(nq, 1)));
    if(nq > 0)
        SHQ_HIP(hipMemcpyAsync(ctx->s_queue0.ptr, queue.data(), sizeof(int32_t) * nq, hipMemcpyHostToDevice, ctx->stream));
    if(!ctx->have_father) { /* no hmax updates */
        SHQ_TRY(ctx->pfather.reserve((size_t) std::max<int64_t>(n, 1)));
        if(n > 0)
            SHQ_HIP(hipMemsetAsync(ctx->pfather.ptr, 0xff, sizeof(int32_t) * n, ctx->stream));
    }
    SHQ_TRY(shq_sph_prepare(ctx, &params->kf, nullptr, nullptr));
    SHQ_TRY(shq_sph_density_begin(ctx, params, ctx->s_queue0.ptr, nq, want_gradrho));
    if(nqueue)
        *nqueue = nq;
    return SHQ_OK;
}

extern "C" int shq_density_ev_primary(shq_context *ctx)
{
    SHQ_CHECK(ctx, SHQ_ERR_INVALID, "null context");
    SHQ_HIP(hipSetDevice(ctx->device));
    return shq_sph_density_primary(ctx);
}

extern "C" int shq_density_ev_postprocess(shq_context *ctx, int64_t *nredo)
{
    SHQ_CHECK(ctx, SHQ_ERR_INVALID, "null context");
    SHQ_HIP(hipSetDevice(ctx->device));
    int64_t nr = 0;
    const int rc = shq_sph_density_post(ctx, &nr);
    if(nredo)
        *nredo = nr;
    if(rc != SHQ_OK)
        ctx->sphrun.phase = 0; /* not converged: the walk is abandoned */
    return rc;
}

extern "C" int shq_density_close(shq_context *ctx, shq_node *nodes_rw, const shq_part_view *parts, const shq_sph_view *sph,
                                 const shq_bh_view *bh, double *EntVarPred, double *GradRho_mag, shq_sph_stats *stats)
{
    SHQ_CHECK(ctx && parts && sph, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->sphrun.phase == 1, SHQ_ERR_STATE, "density_close: no density walk is open");
    SHQ_HIP(hipSetDevice(ctx->device));
    const shq_density_params *params = &ctx->sphrun.dp;
    SHQ_CHECK(!GradRho_mag || ctx->sphrun.want_gradrho, SHQ_ERR_INVALID, "density_close: GradRho_mag needs want_gradrho at open");
    SHQ_CHECK(!params->update_hsml || ctx->have_father || nodes_rw == nullptr, SHQ_ERR_INVALID,
              "density with update_hsml needs tree->father to update hmax (update_tree_hmax_father)");
    const int64_t n = parts->numpart;
    SHQ_CHECK(n == ctx->numpart, SHQ_ERR_INVALID, "density_close: not the particle array of density_open");
    SHQ_TRY(shq_sph_density_end(ctx, stats));
    const std::vector<int32_t> &queue = run_queue(ctx);
    /* results back into the caller's arrays: only the walked targets are assigned
     * (reduce<PRIMARY>, localtreewalk2.h:39) */
    /* D2H into pinned staging */
    const size_t cap = (size_t) std::max<int64_t>(n, 1);
    const size_t nnodes = (size_t) std::max<int64_t>(ctx->numnodes, 1);
    SHQ_TRY(ctx->stage.reserve(cap * (8 * sizeof(double) + sizeof(double4)) + nnodes * sizeof(double) + 256));
    double *hsml = reinterpret_cast<double *>(ctx->stage.ptr);
    double *dthsml = hsml + cap, *density = dthsml + cap, *egywt = density + cap, *dhsml = egywt + cap, *divvel = dhsml + cap,
           *curl = divvel + cap, *gmag = curl + cap;
    double4 *velp = reinterpret_cast<double4 *>(gmag + cap);
    double *hmax = reinterpret_cast<double *>(velp + cap);
    auto dn = [&](void *dst, const void *src, size_t bytes) -> int {
        if(bytes > 0)
            SHQ_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        return SHQ_OK;
    };
    SHQ_TRY(dn(hsml, ctx->hsml.ptr, sizeof(double) * n));
    SHQ_TRY(dn(dthsml, ctx->dthsml.ptr, sizeof(double) * n));
    SHQ_TRY(dn(density, ctx->g_density.ptr, sizeof(double) * n));
    SHQ_TRY(dn(egywt, ctx->g_egywt.ptr, sizeof(double) * n));
    SHQ_TRY(dn(dhsml, ctx->g_dhsmlegy.ptr, sizeof(double) * n));
    SHQ_TRY(dn(divvel, ctx->g_divvel.ptr, sizeof(double) * n));
    SHQ_TRY(dn(curl, ctx->g_curlvel.ptr, sizeof(double) * n));
    SHQ_TRY(dn(velp, ctx->velp.ptr, sizeof(double4) * n));
    if(GradRho_mag) {
        SHQ_TRY(ctx->s_evp_in.reserve(cap));
        SHQ_TRY(shq_sph_gradrho_mag(ctx, ctx->s_evp_in.ptr));
        SHQ_TRY(dn(gmag, ctx->s_evp_in.ptr, sizeof(double) * n));
    }
    SHQ_TRY(dn(hmax, ctx->node_hmax.ptr, sizeof(double) * (size_t) ctx->numnodes));
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    for(int32_t i : queue) {
        *pfield_w<double>(parts, i, parts->off_hsml) = hsml[i];
        if(parts->off_dthsml != SHQ_NOFIELD)
            *pfield_w<double>(parts, i, parts->off_dthsml) = dthsml[i];
        const uint8_t t = *pfield<uint8_t>(parts, i, parts->off_type);
        const int32_t pi = *pfield<int32_t>(parts, i, parts->off_pi);
        if(t == 0) {
            *sfield(sph, pi, sph->off_density) = density[i];
            *sfield(sph, pi, sph->off_egywtdensity) = egywt[i];
            *sfield(sph, pi, sph->off_dhsmlegydensityfactor) = dhsml[i];
            *sfield(sph, pi, sph->off_divvel) = divvel[i];
            *sfield(sph, pi, sph->off_curlvel) = curl[i];
            if(GradRho_mag)
                GradRho_mag[pi] = gmag[i];
        } else {
            SHQ_CHECK(bh && pi >= 0 && pi < bh->numslots, SHQ_ERR_INVALID, "BH particle with PI outside the BH slot array");
            char *b = static_cast<char *>(bh->base) + (size_t) pi * bh->elsize;
            *reinterpret_cast<double *>(b + bh->off_density) = density[i];
            *reinterpret_cast<double *>(b + bh->off_divvel) = divvel[i];
        }
    }
    if(EntVarPred) { /* DensityPriv ctor caches it for every gas particle, densitytree2.hpp:43-50 */
        for(int64_t i = 0; i < n; i++) {
            if(*pfield<uint8_t>(parts, i, parts->off_type) != 0)
                continue;
            if(parts->off_flags != SHQ_NOFIELD && (*pfield<uint32_t>(parts, i, parts->off_flags) & 1u))
                continue;
            EntVarPred[*pfield<int32_t>(parts, i, parts->off_pi)] = velp[i].w;
        }
    }
    if(nodes_rw && params->update_hsml) { /* update_tree_hmax_father wrote leaf hmax on the device */
        for(int64_t j = 0; j < ctx->numnodes; j++) {
            shq_node &nd = nodes_rw[ctx->node_order.empty() ? j : (int64_t) ctx->node_order[j]];
            if(hmax[j] > nd.hmax)
                nd.hmax = hmax[j];
        }
    }
    return SHQ_OK;
}

extern "C" int shq_density(shq_context *ctx, const shq_tree_view *tree, shq_node *nodes_rw, const shq_part_view *parts,
                           const shq_sph_view *sph, const shq_bh_view *bh, const int32_t *active, int64_t nactive,
                           const shq_density_params *params, double *EntVarPred, double *GradRho_mag, shq_sph_stats *stats)
{
    SHQ_CHECK(ctx && tree && parts && sph && params, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(!params->update_hsml || tree->father || nodes_rw == nullptr, SHQ_ERR_INVALID,
              "density with update_hsml needs tree->father to update hmax (update_tree_hmax_father)");
    ctx->sphrun.phase = 0; /* a one-shot call supersedes whatever an earlier, failed walk left open */
    SHQ_TRY(shq_density_open(ctx, tree, parts, sph, bh, active, nactive, params, GradRho_mag != nullptr, nullptr));
    int64_t nredo = 0;
    do {
        int rc = shq_density_ev_primary(ctx);
        if(rc == SHQ_OK)
            rc = shq_density_ev_postprocess(ctx, &nredo);
        if(rc != SHQ_OK) {
            ctx->sphrun.phase = 0;
            return rc;
        }
    } while(nredo > 0);
    return shq_density_close(ctx, nodes_rw, parts, sph, bh, EntVarPred, GradRho_mag, stats);
}

extern "C" int shq_hydro_open(shq_context *ctx, const shq_tree_view *tree, const shq_part_view *parts, const shq_sph_view *sph,
                              const int32_t *active, int64_t nactive, const shq_hydro_params *params, const double *EntVarPred,
                              int64_t *nqueue)
{
    SHQ_CHECK(ctx && tree && parts && sph && params, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->sphrun.phase == 0, SHQ_ERR_STATE, "hydro_open: another SPH walk is open");
    SHQ_HIP(hipSetDevice(ctx->device));
    const int64_t n = parts->numpart;
    if(active)
        for(int64_t k = 0; k < nactive; k++)
            SHQ_CHECK(active[k] >= 0 && active[k] < n, SHQ_ERR_INVALID, "active[%ld] = %d out of range", (long) k, active[k]);
    SHQ_TRY(shq_particles_upload(ctx, parts));
    SHQ_TRY(sph_upload(ctx, parts, sph));
    SHQ_TRY(shq_tree_upload(ctx, tree));
    std::vector<int32_t> &queue = run_queue(ctx);
    queue = build_queue(parts, active, nactive, false);
    const int64_t nq = (int64_t) queue.size();
    SHQ_TRY(ctx->s_queue0.reserve((size_t) std::max<int64_t>(nq, 1)));
    if(nq > 0)
        SHQ_HIP(hipMemcpyAsync(ctx->s_queue0.ptr, queue.data(), sizeof(int32_t) * nq, hipMemcpyHostToDevice, ctx->stream));
    const double *d_evp = nullptr;
    std::vector<double> evp_by_part;
    if(EntVarPred) { /* hydro reuses density()'s EntVarPred array (hydra2.cpp:76, HydroPriv::EntVarPred) */
        evp_by_part.assign((size_t) std::max<int64_t>(n, 1), 0.0);
        for(int64_t i = 0; i < n; i++)
            if(*pfield<uint8_t>(parts, i, parts->off_type) == 0) {
                const int32_t pi = *pfield<int32_t>(parts, i, parts->off_pi);
                if(pi >= 0 && pi < sph->numslots)
                    evp_by_part[i] = EntVarPred[pi];
            }
        SHQ_TRY(up(ctx, ctx->s_evp_in, evp_by_part));
        d_evp = ctx->s_evp_in.ptr;
    }
    SHQ_TRY(shq_sph_prepare(ctx, &params->kf, params, d_evp));
    SHQ_HIP(hipStreamSynchronize(ctx->stream)); /* evp_by_part dies here */
    SHQ_TRY(shq_sph_hydro_begin(ctx, params, ctx->s_queue0.ptr, nq));
    if(nqueue)
        *nqueue = nq;
    return SHQ_OK;
}

extern "C" int shq_hydro_ev_primary(shq_context *ctx)
{
    SHQ_CHECK(ctx, SHQ_ERR_INVALID, "null context");
    SHQ_HIP(hipSetDevice(ctx->device));
    return shq_sph_hydro_primary(ctx);
}

extern "C" int shq_hydro_ev_postprocess(shq_context *ctx)
{
    SHQ_CHECK(ctx, SHQ_ERR_INVALID, "null context");
    SHQ_HIP(hipSetDevice(ctx->device));
    return shq_sph_hydro_post(ctx);
}

extern "C" int shq_hydro_close(shq_context *ctx, const shq_part_view *parts, const shq_sph_view *sph, shq_sph_stats *stats)
{
    SHQ_CHECK(ctx && parts && sph, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->sphrun.phase == 2, SHQ_ERR_STATE, "hydro_close: no hydro walk is open");
    SHQ_HIP(hipSetDevice(ctx->device));
    const int64_t n = parts->numpart;
    SHQ_CHECK(n == ctx->numpart, SHQ_ERR_INVALID, "hydro_close: not the particle array of hydro_open");
    SHQ_TRY(shq_sph_hydro_end(ctx, stats));
    const std::vector<int32_t> &queue = run_queue(ctx);
    const size_t cap = (size_t) std::max<int64_t>(n, 1);
    SHQ_TRY(ctx->stage.reserve(cap * 5 * sizeof(double) + 256));
    double *hacc = reinterpret_cast<double *>(ctx->stage.ptr), *dtent = hacc + 3 * cap, *maxsig = dtent + cap;
    if(n > 0) {
        SHQ_HIP(hipMemcpyAsync(hacc, ctx->g_hydroaccel_out.ptr, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, ctx->stream));
        SHQ_HIP(hipMemcpyAsync(dtent, ctx->g_dtentropy_out.ptr, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
        SHQ_HIP(hipMemcpyAsync(maxsig, ctx->g_maxsignalvel.ptr, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    }
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    for(int32_t i : queue) {
        const int32_t pi = *pfield<int32_t>(parts, i, parts->off_pi);
        double *ha = sfield(sph, pi, sph->off_hydroaccel);
        ha[0] = hacc[3 * (size_t) i];
        ha[1] = hacc[3 * (size_t) i + 1];
        ha[2] = hacc[3 * (size_t) i + 2];
        *sfield(sph, pi, sph->off_dtentropy) = dtent[i];
        *sfield(sph, pi, sph->off_maxsignalvel) = maxsig[i];
    }
    return SHQ_OK;
}

extern "C" int shq_hydro_force(shq_context *ctx, const shq_tree_view *tree, const shq_part_view *parts,
                               const shq_sph_view *sph, const int32_t *active, int64_t nactive,
                               const shq_hydro_params *params, const double *EntVarPred, shq_sph_stats *stats)
{
    SHQ_CHECK(ctx && tree && parts && sph && params, SHQ_ERR_INVALID, "null argument");
    ctx->sphrun.phase = 0;
    SHQ_TRY(shq_hydro_open(ctx, tree, parts, sph, active, nactive, params, EntVarPred, nullptr));
    int rc = shq_hydro_ev_primary(ctx);
    if(rc == SHQ_OK)
        rc = shq_hydro_ev_postprocess(ctx);
    if(rc != SHQ_OK) {
        ctx->sphrun.phase = 0;
        return rc;
    }
    return shq_hydro_close(ctx, parts, sph, stats);
}

/* ---- the exchange between ev_primary and ev_postprocess ------------------------------------------------------------ */
namespace {

/* group (place, result) pairs by place, keeping the order within a place; returns pointers to use */
template <typename R> void group_by_place(const int32_t *&place, const R *&results, int64_t n, std::vector<int32_t> &hp, std::vector<R> &hr)
{
    bool grouped = true;
    for(int64_t k = 1; k < n; k++)
        if(place[k] < place[k - 1]) {
            grouped = false;
            break;
        }
    if(grouped)
        return;
    std::vector<int64_t> ord((size_t) n);
    for(int64_t k = 0; k < n; k++)
        ord[k] = k;
    std::stable_sort(ord.begin(), ord.end(), [&](int64_t x, int64_t y) { return place[x] < place[y]; });
    hp.resize((size_t) n);
    hr.resize((size_t) n);
    for(int64_t k = 0; k < n; k++) {
        hp[k] = place[ord[k]];
        hr[k] = results[ord[k]];
    }
    place = hp.data();
    results = hr.data();
}

template <typename R, class Fn> int reduce_results(shq_context *ctx, const int32_t *place, const R *results, int64_t n, Fn &&launch)
{
    SHQ_CHECK(n >= 0 && (n == 0 || (place && results)), SHQ_ERR_INVALID, "ev_reduce: bad arguments");
    if(n == 0)
        return SHQ_OK;
    for(int64_t k = 0; k < n; k++)
        SHQ_CHECK(place[k] >= 0 && place[k] < ctx->numpart, SHQ_ERR_INVALID, "ev_reduce: place[%ld] = %d out of range", (long) k, place[k]);
    std::vector<int32_t> hp;
    std::vector<R> hr;
    group_by_place(place, results, n, hp, hr);
    DevBuf<int32_t> dplace;
    DevBuf<R> dres;
    auto run = [&]() -> int {
        SHQ_TRY(dplace.reserve((size_t) n));
        SHQ_TRY(dres.reserve((size_t) n));
        SHQ_HIP(hipMemcpyAsync(dplace.ptr, place, sizeof(int32_t) * n, hipMemcpyHostToDevice, ctx->stream));
        SHQ_HIP(hipMemcpyAsync(dres.ptr, results, sizeof(R) * n, hipMemcpyHostToDevice, ctx->stream));
        SHQ_TRY(launch(dplace.ptr, (const void *) dres.ptr));
        return SHQ_OK;
    };
    const int rc = run();
    (void) hipStreamSynchronize(ctx->stream);
    dplace.release();
    dres.release();
    return rc;
}

/* NodeList (host node numbers) -> sorted packed start indices of the branches */
int query_segments(shq_context *ctx, const int32_t *nodelist, int64_t q, int4 *seg)
{
    const int64_t fn = ctx->firstnode, nn = ctx->numnodes;
    const int64_t nall = ctx->node_rank.empty() ? nn : (int64_t) ctx->node_rank.size();
    int32_t st[4];
    int ns = 0;
    for(int k = 0; k < 4 && nodelist[k] >= 0; k++) {
        const int64_t no = nodelist[k];
        SHQ_CHECK(no >= fn && no < fn + nall, SHQ_ERR_INVALID, "query %ld: NodeList entry %ld is not a local node", (long) q, (long) no);
        const int32_t r = ctx->node_rank.empty() ? (int32_t) (no - fn) : ctx->node_rank[(size_t) (no - fn)];
        SHQ_CHECK(r >= 0 && r < nn, SHQ_ERR_INVALID, "query %ld: NodeList entry %ld is not reachable from the root", (long) q, (long) no);
        st[ns++] = r;
    }
    std::sort(st, st + ns);
    *seg = make_int4(ns > 0 ? st[0] : -1, ns > 1 ? st[1] : -1, ns > 2 ? st[2] : -1, ns > 3 ? st[3] : -1);
    return SHQ_OK;
}

} // namespace

extern "C" int shq_density_ev_reduce(shq_context *ctx, const int32_t *place, const shq_density_result *results, int64_t n)
{
    SHQ_CHECK(ctx, SHQ_ERR_INVALID, "null context");
    SHQ_CHECK(ctx->sphrun.phase == 1, SHQ_ERR_STATE, "density_ev_reduce: no density walk is open");
    SHQ_HIP(hipSetDevice(ctx->device));
    return reduce_results(ctx, place, results, n, [&](const int32_t *dp, const void *dr) { return shq_sph_density_reduce(ctx, dp, dr, n); });
}

extern "C" int shq_hydro_ev_reduce(shq_context *ctx, const int32_t *place, const shq_hydro_result *results, int64_t n)
{
    SHQ_CHECK(ctx, SHQ_ERR_INVALID, "null context");
    SHQ_CHECK(ctx->sphrun.phase == 2, SHQ_ERR_STATE, "hydro_ev_reduce: no hydro walk is open");
    SHQ_HIP(hipSetDevice(ctx->device));
    return reduce_results(ctx, place, results, n, [&](const int32_t *dp, const void *dr) { return shq_sph_hydro_reduce(ctx, dp, dr, n); });
}

extern "C" int shq_density_ev_secondary(shq_context *ctx, const shq_density_params *params, const shq_density_query *queries, int64_t nq,
                                        shq_density_result *results, int64_t *ninteractions_total)
{
    SHQ_CHECK(ctx && params && (nq == 0 || (queries && results)), SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->have_parts && ctx->have_tree && ctx->have_sph, SHQ_ERR_STATE, "density_ev_secondary: no SPH state on the device (open a walk first)");
    SHQ_CHECK(nq >= 0 && nq < (1ll << 31), SHQ_ERR_INVALID, "bad query count");
    SHQ_HIP(hipSetDevice(ctx->device));
    if(ninteractions_total)
        *ninteractions_total = 0;
    if(nq == 0)
        return SHQ_OK;
    const size_t n = (size_t) nq;
    std::vector<double4> hpos(n), hvel(n);
    std::vector<double> hh(n);
    std::vector<uint8_t> hf(n);
    std::vector<int4> hseg(n);
    for(size_t q = 0; q < n; q++) {
        const shq_density_query &Q = queries[q];
        hpos[q] = make_double4(Q.Pos[0], Q.Pos[1], Q.Pos[2], 0.0);
        hvel[q] = make_double4(Q.Vel[0], Q.Vel[1], Q.Vel[2], 0.0);
        hh[q] = Q.Hsml;
        SHQ_CHECK(Q.Type >= 0 && Q.Type < 6 && Q.Hsml > 0, SHQ_ERR_INVALID, "query %ld: bad Type %d or Hsml %g", (long) q, Q.Type, Q.Hsml);
        hf[q] = (uint8_t) (Q.Type << 4);
        SHQ_TRY(query_segments(ctx, Q.NodeList, (int64_t) q, &hseg[q]));
    }
    DevBuf<double4> dpos, dvel;
    DevBuf<double> dh, dout;
    DevBuf<uint8_t> df;
    DevBuf<int4> dseg;
    DevBuf<unsigned long long> dn;
    std::vector<double> hout(12 * n);
    unsigned long long hn = 0;
    auto run = [&]() -> int {
        SHQ_TRY(dpos.reserve(n)); SHQ_TRY(dvel.reserve(n)); SHQ_TRY(dh.reserve(n)); SHQ_TRY(df.reserve(n)); SHQ_TRY(dseg.reserve(n));
        SHQ_TRY(dout.reserve(12 * n)); SHQ_TRY(dn.reserve(8));
        hipStream_t st = ctx->stream;
        SHQ_HIP(hipMemcpyAsync(dpos.ptr, hpos.data(), sizeof(double4) * n, hipMemcpyHostToDevice, st));
        SHQ_HIP(hipMemcpyAsync(dvel.ptr, hvel.data(), sizeof(double4) * n, hipMemcpyHostToDevice, st));
        SHQ_HIP(hipMemcpyAsync(dh.ptr, hh.data(), sizeof(double) * n, hipMemcpyHostToDevice, st));
        SHQ_HIP(hipMemcpyAsync(df.ptr, hf.data(), n, hipMemcpyHostToDevice, st));
        SHQ_HIP(hipMemcpyAsync(dseg.ptr, hseg.data(), sizeof(int4) * n, hipMemcpyHostToDevice, st));
        SHQ_HIP(hipMemsetAsync(dout.ptr, 0, sizeof(double) * 12 * n, st));
        SHQ_HIP(hipMemsetAsync(dn.ptr, 0, sizeof(unsigned long long) * 8, st));
        SHQ_TRY(shq_sph_density_secondary(ctx, params, dpos.ptr, dh.ptr, dvel.ptr, df.ptr, dseg.ptr, nq, dout.ptr, dn.ptr));
        SHQ_HIP(hipMemcpyAsync(hout.data(), dout.ptr, sizeof(double) * 12 * n, hipMemcpyDeviceToHost, st));
        SHQ_HIP(hipMemcpyAsync(&hn, dn.ptr, sizeof(hn), hipMemcpyDeviceToHost, st));
        SHQ_HIP(hipStreamSynchronize(st));
        return SHQ_OK;
    };
    const int rc = run();
    (void) hipStreamSynchronize(ctx->stream);
    dpos.release(); dvel.release(); dh.release(); df.release(); dseg.release(); dout.release(); dn.release();
    SHQ_TRY(rc);
    for(size_t q = 0; q < n; q++) {
        shq_density_result &R = results[q];
        R.Ngb = hout[q];
        R.DhsmlDensity = hout[n + q];
        R.Rho = hout[2 * n + q];
        R.Div = hout[3 * n + q];
        R.EgyRho = hout[4 * n + q];
        R.DhsmlEgyDensity = hout[5 * n + q];
        for(int k = 0; k < 3; k++) {
            R.Rot[k] = hout[6 * n + 3 * q + k];
            R.GradRho[k] = hout[9 * n + 3 * q + k];
        }
    }
    if(ninteractions_total)
        *ninteractions_total = (int64_t) hn;
    return SHQ_OK;
}

extern "C" int shq_hydro_ev_secondary(shq_context *ctx, const shq_hydro_params *params, const shq_hydro_query *queries, int64_t nq,
                                      shq_hydro_result *results, int64_t *ninteractions_total)
{
    SHQ_CHECK(ctx && params && (nq == 0 || (queries && results)), SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->have_parts && ctx->have_tree && ctx->have_sph, SHQ_ERR_STATE, "hydro_ev_secondary: no SPH state on the device (open a walk first)");
    SHQ_CHECK(nq >= 0 && nq < (1ll << 31), SHQ_ERR_INVALID, "bad query count");
    SHQ_HIP(hipSetDevice(ctx->device));
    if(ninteractions_total)
        *ninteractions_total = 0;
    if(nq == 0)
        return SHQ_OK;
    const size_t n = (size_t) nq;
    std::vector<double4> hpos(n), hvel(n), hC(n), hD(n);
    std::vector<double> hh(n);
    std::vector<int4> hseg(n);
    const double GAMMA = 5.0 / 3.0;
    for(size_t q = 0; q < n; q++) {
        const shq_hydro_query &Q = queries[q];
        SHQ_CHECK(Q.Hsml > 0 && Q.EgyRho > 0 && Q.TimeBinHydro >= 0 && Q.TimeBinHydro <= SHQ_TIMEBINS, SHQ_ERR_INVALID,
                  "query %ld: bad Hsml %g, EgyRho %g or TimeBinHydro %d", (long) q, Q.Hsml, Q.EgyRho, Q.TimeBinHydro);
        hpos[q] = make_double4(Q.Pos[0], Q.Pos[1], Q.Pos[2], Q.Mass);
        hvel[q] = make_double4(Q.Vel[0], Q.Vel[1], Q.Vel[2], Q.EntVarPred);
        hh[q] = Q.Hsml;
        /* HydroLocalTreeWalk ctor, hydratree2.hpp:240-245, and the rr1 of ngbiter (:353-367) */
        const double cs = sqrt(GAMMA * Q.Pressure / Q.EgyRho);
        hC[q] = make_double4(Q.EntVarPred, Q.Density, cs, Q.Pressure / (Q.EgyRho * Q.EgyRho));
        double rr1 = 1;
        if(params->DensityIndependentSphOn) {
            rr1 = 0;
            if(params->DensityContrastLimit >= 0) {
                rr1 = Q.EgyRho / Q.Density;
                if(params->DensityContrastLimit > 0)
                    rr1 = std::min(rr1, params->DensityContrastLimit);
            }
        }
        hD[q] = make_double4(Q.SPH_DhsmlDensityFactor, rr1, Q.F1, params->kf.dloga_for_bin[Q.TimeBinHydro]);
        SHQ_TRY(query_segments(ctx, Q.NodeList, (int64_t) q, &hseg[q]));
    }
    DevBuf<double4> dpos, dvel, dC, dD;
    DevBuf<double> dh, dout;
    DevBuf<int4> dseg;
    DevBuf<unsigned long long> dn;
    std::vector<double> hout(5 * n);
    unsigned long long hn = 0;
    auto run = [&]() -> int {
        SHQ_TRY(dpos.reserve(n)); SHQ_TRY(dvel.reserve(n)); SHQ_TRY(dC.reserve(n)); SHQ_TRY(dD.reserve(n)); SHQ_TRY(dh.reserve(n));
        SHQ_TRY(dseg.reserve(n)); SHQ_TRY(dout.reserve(5 * n)); SHQ_TRY(dn.reserve(8));
        hipStream_t st = ctx->stream;
        SHQ_HIP(hipMemcpyAsync(dpos.ptr, hpos.data(), sizeof(double4) * n, hipMemcpyHostToDevice, st));
        SHQ_HIP(hipMemcpyAsync(dvel.ptr, hvel.data(), sizeof(double4) * n, hipMemcpyHostToDevice, st));
        SHQ_HIP(hipMemcpyAsync(dC.ptr, hC.data(), sizeof(double4) * n, hipMemcpyHostToDevice, st));
        SHQ_HIP(hipMemcpyAsync(dD.ptr, hD.data(), sizeof(double4) * n, hipMemcpyHostToDevice, st));
        SHQ_HIP(hipMemcpyAsync(dh.ptr, hh.data(), sizeof(double) * n, hipMemcpyHostToDevice, st));
        SHQ_HIP(hipMemcpyAsync(dseg.ptr, hseg.data(), sizeof(int4) * n, hipMemcpyHostToDevice, st));
        SHQ_HIP(hipMemsetAsync(dout.ptr, 0, sizeof(double) * 5 * n, st));
        SHQ_HIP(hipMemsetAsync(dn.ptr, 0, sizeof(unsigned long long) * 8, st));
        SHQ_TRY(shq_sph_hydro_secondary(ctx, params, dpos.ptr, dh.ptr, dvel.ptr, dC.ptr, dD.ptr, dseg.ptr, nq, dout.ptr, dn.ptr));
        SHQ_HIP(hipMemcpyAsync(hout.data(), dout.ptr, sizeof(double) * 5 * n, hipMemcpyDeviceToHost, st));
        SHQ_HIP(hipMemcpyAsync(&hn, dn.ptr, sizeof(hn), hipMemcpyDeviceToHost, st));
        SHQ_HIP(hipStreamSynchronize(st));
        return SHQ_OK;
    };
    const int rc = run();
    (void) hipStreamSynchronize(ctx->stream);
    dpos.release(); dvel.release(); dC.release(); dD.release(); dh.release(); dseg.release(); dout.release(); dn.release();
    SHQ_TRY(rc);
    for(size_t q = 0; q < n; q++) {
        shq_hydro_result &R = results[q];
        R.Acc[0] = hout[3 * q];
        R.Acc[1] = hout[3 * q + 1];
        R.Acc[2] = hout[3 * q + 2];
        R.DtEntropy = hout[3 * n + q];
        R.MaxSignalVel = hout[4 * n + q];
    }
    if(ninteractions_total)
        *ninteractions_total = (int64_t) hn;
    return SHQ_OK;
}

extern "C" int shq_sph_exports(shq_context *ctx, int32_t *exportcounts, shq_data_index *table, int64_t capacity, int64_t *nexport)
{
    SHQ_CHECK(ctx, SHQ_ERR_INVALID, "null context");
    const int phase = ctx->sphrun.phase;
    SHQ_CHECK(phase == 1 || phase == 2, SHQ_ERR_STATE, "sph_exports: no SPH walk is open");
    const double Box = phase == 1 ? ctx->sphrun.dp.BoxSize : ctx->sphrun.hp.BoxSize;
    return shq_ngb_toptree_exports(ctx, phase == 2, Box, SHQ_SPH_QUEUE_RESIDENT, 0, exportcounts, table, capacity, nexport);
}

extern "C" int shq_sph_fill_queries(shq_context *ctx, const shq_data_index *table, int64_t n, void *queries)
{
    SHQ_CHECK(ctx && n >= 0 && (n == 0 || (table && queries)), SHQ_ERR_INVALID, "bad argument");
    const int phase = ctx->sphrun.phase;
    SHQ_CHECK(phase == 1 || phase == 2, SHQ_ERR_STATE, "sph_fill_queries: no SPH walk is open");
    if(n == 0)
        return SHQ_OK;
    SHQ_HIP(hipSetDevice(ctx->device));
    for(int64_t k = 0; k < n; k++)
        SHQ_CHECK(table[k].Index >= 0 && table[k].Index < ctx->numpart, SHQ_ERR_INVALID, "table[%ld].Index = %d out of range", (long) k, table[k].Index);
    const size_t rec = phase == 1 ? sizeof(shq_density_query) : sizeof(shq_hydro_query);
    DevBuf<shq_data_index> dt;
    DevBuf<char> dq;
    auto run = [&]() -> int {
        SHQ_TRY(dt.reserve((size_t) n));
        SHQ_TRY(dq.reserve((size_t) n * rec));
        SHQ_HIP(hipMemcpyAsync(dt.ptr, table, sizeof(shq_data_index) * n, hipMemcpyHostToDevice, ctx->stream));
        SHQ_TRY(shq_sph_fill_queries_device(ctx, (const shq_data_index *) dt.ptr, n, (void *) dq.ptr));
        SHQ_HIP(hipMemcpyAsync(queries, dq.ptr, (size_t) n * rec, hipMemcpyDeviceToHost, ctx->stream));
        SHQ_HIP(hipStreamSynchronize(ctx->stream));
        return SHQ_OK;
    };
    const int rc = run();
    (void) hipStreamSynchronize(ctx->stream);
    dt.release();
    dq.release();
    return rc;
}

extern "C" int shq_stellar_density(shq_context *ctx, const shq_tree_view *tree, const shq_part_view *parts, const shq_sph_view *sph,
                                   const int32_t *queue, int64_t nqueue, const shq_stellar_params *params, double *StarVolumeSPH,
                                   shq_sph_stats *stats)
{
    SHQ_CHECK(ctx && tree && parts && sph && params && StarVolumeSPH && (nqueue == 0 || queue), SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->sphrun.phase == 0, SHQ_ERR_STATE, "stellar_density: an SPH walk is open");
    SHQ_CHECK(nqueue >= 0, SHQ_ERR_INVALID, "stellar_density: bad queue length");
    SHQ_HIP(hipSetDevice(ctx->device));
    const int64_t n = parts->numpart;
    for(int64_t k = 0; k < nqueue; k++) {
        const int32_t i = queue[k];
        SHQ_CHECK(i >= 0 && i < n, SHQ_ERR_INVALID, "stellar_density: queue[%ld] = %d out of range", (long) k, i);
        SHQ_CHECK(*pfield<uint8_t>(parts, i, parts->off_type) == 4, SHQ_ERR_INVALID, "stellar_density: particle %d in the queue is not a star", i);
        SHQ_CHECK(*pfield<double>(parts, i, parts->off_hsml) > 0, SHQ_ERR_INVALID,
                  "stellar_density: star %d has Hsml <= 0 (the reference re-seeds it from its father node; do that before the call)", i);
    }
    SHQ_TRY(shq_particles_upload(ctx, parts));
    SHQ_TRY(sph_upload(ctx, parts, sph));
    SHQ_TRY(shq_tree_upload(ctx, tree));
    if(nqueue == 0) {
        if(stats)
            memset(stats, 0, sizeof(*stats));
        return SHQ_OK;
    }
    SHQ_TRY(ctx->s_queue0.reserve((size_t) nqueue));
    SHQ_HIP(hipMemcpyAsync(ctx->s_queue0.ptr, queue, sizeof(int32_t) * nqueue, hipMemcpyHostToDevice, ctx->stream));
    /* velp (only its garbage / type flags matter here) and a scratch Hsml-by-slot array for the shared leaf gather */
    SHQ_TRY(ctx->velp.reserve((size_t) std::max<int64_t>(n, 1)));
    SHQ_TRY(ctx->s_evp_in.reserve((size_t) (ctx->ntreeparts + SHQ_NMAXCHILD)));
    SHQ_TRY(ctx->s_gradrho.reserve((size_t) std::max<int64_t>(n, 1)));
    SHQ_HIP(hipMemsetAsync(ctx->s_gradrho.ptr, 0, sizeof(double) * std::max<int64_t>(n, 1), ctx->stream));
    SHQ_TRY(shq_sph_stellar_density_device(ctx, params, ctx->s_queue0.ptr, nqueue, ctx->s_gradrho.ptr, stats));
    std::vector<double> hsml, vol;
    SHQ_TRY(down(ctx, ctx->hsml, hsml, n));
    SHQ_TRY(down(ctx, ctx->s_gradrho, vol, n));
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    for(int64_t k = 0; k < nqueue; k++) {
        const int32_t i = queue[k];
        *pfield_w<double>(parts, i, parts->off_hsml) = hsml[i];
        StarVolumeSPH[*pfield<int32_t>(parts, i, parts->off_pi)] = vol[i];
    }
    return SHQ_OK;
}

extern "C" int shq_bh_veldisp(shq_context *ctx, const shq_tree_view *tree, const shq_part_view *parts, const int32_t *active, int64_t nactive,
                              const shq_kick_factors *kf, double *NumDM, double (*V1sumDM)[3], double *V2sumDM, double *VDisp)
{
    SHQ_CHECK(ctx && tree && parts && kf && VDisp, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->sphrun.phase == 0, SHQ_ERR_STATE, "bh_veldisp: an SPH walk is open");
    SHQ_CHECK(parts->off_vel != SHQ_NOFIELD && parts->off_hsml != SHQ_NOFIELD && parts->off_pi != SHQ_NOFIELD &&
                  parts->off_type != SHQ_NOFIELD && parts->off_treeacc != SHQ_NOFIELD && parts->off_gravpm != SHQ_NOFIELD,
              SHQ_ERR_INVALID, "bh_veldisp: the particle view needs Vel, Hsml, PI, Type, FullTreeGravAccel and GravPM");
    SHQ_HIP(hipSetDevice(ctx->device));
    const int64_t n = parts->numpart;
    std::vector<int32_t> queue;
    const int64_t nloop = active ? nactive : n;
    for(int64_t k = 0; k < nloop; k++) {
        const int32_t i = active ? active[k] : (int32_t) k;
        SHQ_CHECK(i >= 0 && i < n, SHQ_ERR_INVALID, "bh_veldisp: active[%ld] = %d out of range", (long) k, i);
        if(parts->off_flags != SHQ_NOFIELD && (*pfield<uint32_t>(parts, i, parts->off_flags) & 3u))
            continue;
        if(*pfield<uint8_t>(parts, i, parts->off_type) == 5) {
            SHQ_CHECK(*pfield<double>(parts, i, parts->off_hsml) > 0, SHQ_ERR_INVALID, "bh_veldisp: black hole %d has Hsml <= 0", i);
            queue.push_back(i);
        }
    }
    SHQ_TRY(shq_particles_upload(ctx, parts));
    SHQ_TRY(shq_dynamics_upload(ctx, parts));
    SHQ_TRY(shq_tree_upload(ctx, tree));
    const int64_t nq = (int64_t) queue.size();
    if(nq == 0)
        return SHQ_OK;
    DevBuf<double> dout;
    std::vector<double> hout(5 * (size_t) nq);
    auto run = [&]() -> int {
        SHQ_TRY(ctx->s_queue0.reserve((size_t) nq));
        SHQ_TRY(dout.reserve(5 * (size_t) nq));
        SHQ_HIP(hipMemcpyAsync(ctx->s_queue0.ptr, queue.data(), sizeof(int32_t) * nq, hipMemcpyHostToDevice, ctx->stream));
        SHQ_TRY(shq_bh_veldisp_device(ctx, kf, tree->BoxSize, ctx->s_queue0.ptr, nq, dout.ptr));
        SHQ_HIP(hipMemcpyAsync(hout.data(), dout.ptr, sizeof(double) * 5 * nq, hipMemcpyDeviceToHost, ctx->stream));
        SHQ_HIP(hipStreamSynchronize(ctx->stream));
        return SHQ_OK;
    };
    const int rc = run();
    (void) hipStreamSynchronize(ctx->stream);
    dout.release();
    SHQ_TRY(rc);
    for(int64_t q = 0; q < nq; q++) {
        const int32_t pi = *pfield<int32_t>(parts, queue[q], parts->off_pi);
        SHQ_CHECK(pi >= 0, SHQ_ERR_INVALID, "bh_veldisp: black hole %d has a negative slot index", queue[q]);
        const double num = hout[5 * q], *v1 = &hout[5 * q + 1], v2 = hout[5 * q + 4];
        if(NumDM)
            NumDM[pi] = num;
        if(V1sumDM)
            for(int d = 0; d < 3; d++)
                V1sumDM[pi][d] = v1[d];
        if(V2sumDM)
            V2sumDM[pi] = v2;
        if(num > 0) { /* BHVelDispOutput::postprocess, veldisp2.cpp:49-63 */
            double vdisp = v2 / num;
            for(int d = 0; d < 3; d++)
                vdisp -= pow(v1[d] / num, 2);
            if(vdisp > 0)
                VDisp[pi] = sqrt(vdisp / 3);
        }
    }
    return SHQ_OK;
}

extern "C" int shq_wind_veldisp(shq_context *ctx, const shq_tree_view *tree, const shq_part_view *parts, const int32_t *queue, int64_t nqueue,
                                const shq_kick_factors *kf, double Time, double hubble, double *VDisp, shq_sph_stats *stats)
{
    SHQ_CHECK(ctx && tree && parts && kf && VDisp && (nqueue == 0 || queue), SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->sphrun.phase == 0, SHQ_ERR_STATE, "wind_veldisp: an SPH walk is open");
    SHQ_CHECK(nqueue >= 0 && Time > 0, SHQ_ERR_INVALID, "wind_veldisp: bad queue length or Time");
    SHQ_CHECK(parts->off_vel != SHQ_NOFIELD && parts->off_hsml != SHQ_NOFIELD && parts->off_pi != SHQ_NOFIELD &&
                  parts->off_type != SHQ_NOFIELD && parts->off_treeacc != SHQ_NOFIELD && parts->off_gravpm != SHQ_NOFIELD,
              SHQ_ERR_INVALID, "wind_veldisp: the particle view needs Vel, Hsml, PI, Type, FullTreeGravAccel and GravPM");
    SHQ_HIP(hipSetDevice(ctx->device));
    const int64_t n = parts->numpart;
    for(int64_t k = 0; k < nqueue; k++) {
        const int32_t i = queue[k];
        SHQ_CHECK(i >= 0 && i < n, SHQ_ERR_INVALID, "wind_veldisp: queue[%ld] = %d out of range", (long) k, i);
        SHQ_CHECK(*pfield<uint8_t>(parts, i, parts->off_type) == 0, SHQ_ERR_INVALID, "wind_veldisp: particle %d in the queue is not gas", i);
        SHQ_CHECK(*pfield<double>(parts, i, parts->off_hsml) > 0, SHQ_ERR_INVALID, "wind_veldisp: gas particle %d has Hsml <= 0", i);
    }
    SHQ_TRY(shq_particles_upload(ctx, parts));
    SHQ_TRY(shq_dynamics_upload(ctx, parts));
    SHQ_TRY(shq_tree_upload(ctx, tree));
    if(nqueue == 0) {
        if(stats)
            memset(stats, 0, sizeof(*stats));
        return SHQ_OK;
    }
    const size_t cap = (size_t) std::max<int64_t>(n, 1);
    SHQ_TRY(ctx->s_queue0.reserve((size_t) nqueue));
    SHQ_TRY(ctx->s_gradrho.reserve(2 * cap)); /* DMRadius and VDisp by particle index */
    double *d_dm = ctx->s_gradrho.ptr, *d_vd = ctx->s_gradrho.ptr + cap;
    SHQ_HIP(hipMemcpyAsync(ctx->s_queue0.ptr, queue, sizeof(int32_t) * nqueue, hipMemcpyHostToDevice, ctx->stream));
    SHQ_HIP(hipMemcpyAsync(d_dm, ctx->hsml.ptr, sizeof(double) * n, hipMemcpyDeviceToDevice, ctx->stream)); /* DMRadius starts as Hsml, :261 */
    {
        std::vector<double> neg(cap, -1.0);
        SHQ_HIP(hipMemcpyAsync(d_vd, neg.data(), sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
        SHQ_HIP(hipStreamSynchronize(ctx->stream));
    }
    SHQ_TRY(shq_wind_veldisp_device(ctx, kf, tree->BoxSize, hubble * Time * Time, ctx->s_queue0.ptr, nqueue, d_dm, d_vd, stats));
    std::vector<double> vd(cap);
    SHQ_HIP(hipMemcpyAsync(vd.data(), d_vd, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    for(int64_t k = 0; k < nqueue; k++) {
        const int32_t i = queue[k];
        if(vd[i] >= 0)
            VDisp[*pfield<int32_t>(parts, i, parts->off_pi)] = vd[i];
    }
    return SHQ_OK;
}

extern "C" int shq_bh_dynfric(shq_context *ctx, const shq_tree_view *tree, const shq_part_view *parts, const int32_t *queue, int64_t nqueue,
                              const shq_kick_factors *kf, int BH_DynFrictionMethod, int DensityKernelType, int typemask,
                              const shq_bh_dynfric_out *out)
{
    SHQ_CHECK(ctx && tree && parts && kf && out && out->MinPot && out->MinPotPos && out->MinPotVel && (nqueue == 0 || queue), SHQ_ERR_INVALID,
              "null argument");
    SHQ_CHECK(ctx->sphrun.phase == 0, SHQ_ERR_STATE, "bh_dynfric: an SPH walk is open");
    SHQ_CHECK(BH_DynFrictionMethod == 0 || (out->DF_SurroundingDensity && out->DF_SurroundingVel && out->DF_SurroundingRmsVel), SHQ_ERR_INVALID,
              "bh_dynfric: the friction outputs are needed for BH_DynFrictionMethod > 0");
    SHQ_CHECK(parts->off_vel != SHQ_NOFIELD && parts->off_hsml != SHQ_NOFIELD && parts->off_pi != SHQ_NOFIELD && parts->off_type != SHQ_NOFIELD &&
                  parts->off_treeacc != SHQ_NOFIELD && parts->off_gravpm != SHQ_NOFIELD && parts->off_potential != SHQ_NOFIELD,
              SHQ_ERR_INVALID, "bh_dynfric: the particle view needs Vel, Hsml, PI, Type, FullTreeGravAccel, GravPM and Potential");
    SHQ_HIP(hipSetDevice(ctx->device));
    const int64_t n = parts->numpart;
    for(int64_t k = 0; k < nqueue; k++) {
        const int32_t i = queue[k];
        SHQ_CHECK(i >= 0 && i < n, SHQ_ERR_INVALID, "bh_dynfric: queue[%ld] = %d out of range", (long) k, i);
        SHQ_CHECK(*pfield<uint8_t>(parts, i, parts->off_type) == 5, SHQ_ERR_INVALID, "bh_dynfric: particle %d in the queue is not a black hole", i);
        SHQ_CHECK(*pfield<double>(parts, i, parts->off_hsml) > 0, SHQ_ERR_INVALID, "bh_dynfric: black hole %d has Hsml <= 0", i);
    }
    SHQ_TRY(shq_particles_upload(ctx, parts));
    SHQ_TRY(shq_dynamics_upload(ctx, parts));
    SHQ_TRY(shq_tree_upload(ctx, tree));
    if(nqueue == 0)
        return SHQ_OK;
    const size_t cap = (size_t) std::max<int64_t>(n, 1);
    SHQ_TRY(ctx->stage.reserve(sizeof(double) * cap + 256));
    double *hp = reinterpret_cast<double *>(ctx->stage.ptr);
    for(int64_t i = 0; i < n; i++)
        hp[i] = *pfield<double>(parts, i, parts->off_potential);
    SHQ_TRY(ctx->s_gradrho.reserve(cap + 12 * (size_t) nqueue));
    double *d_pot = ctx->s_gradrho.ptr, *d_out = ctx->s_gradrho.ptr + cap;
    SHQ_TRY(ctx->s_queue0.reserve((size_t) nqueue));
    SHQ_HIP(hipMemcpyAsync(d_pot, hp, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
    SHQ_HIP(hipMemcpyAsync(ctx->s_queue0.ptr, queue, sizeof(int32_t) * nqueue, hipMemcpyHostToDevice, ctx->stream));
    SHQ_TRY(shq_bh_dynfric_device(ctx, kf, tree->BoxSize, DensityKernelType, typemask, BH_DynFrictionMethod, d_pot, ctx->s_queue0.ptr, nqueue, d_out));
    std::vector<double> ho(12 * (size_t) nqueue);
    SHQ_HIP(hipMemcpyAsync(ho.data(), d_out, sizeof(double) * 12 * nqueue, hipMemcpyDeviceToHost, ctx->stream));
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    for(int64_t q = 0; q < nqueue; q++) {
        const int32_t pi = *pfield<int32_t>(parts, queue[q], parts->off_pi);
        SHQ_CHECK(pi >= 0, SHQ_ERR_INVALID, "bh_dynfric: black hole %d has a negative slot index", queue[q]);
        const double *o = &ho[12 * q];
        if(out->MinPot[pi] > o[0]) { /* BHReposResult::reduce, bhdynfric.cpp:106-118 */
            out->MinPot[pi] = o[0];
            for(int d = 0; d < 3; d++) {
                out->MinPotPos[pi][d] = o[1 + d];
                out->MinPotVel[pi][d] = o[4 + d];
            }
            if(out->updated)
                out->updated[pi] = 1;
        }
        if(BH_DynFrictionMethod > 0) { /* BHDynFricResult::reduce<PRIMARY> assigns, then postprocess (:66-82) */
            const double dens = o[7];
            out->DF_SurroundingDensity[pi] = dens;
            double rms = o[11];
            if(dens > 0) {
                rms = sqrt(rms / dens);
                for(int d = 0; d < 3; d++)
                    out->DF_SurroundingVel[pi][d] = o[8 + d] / dens;
            } else
                for(int d = 0; d < 3; d++)
                    out->DF_SurroundingVel[pi][d] = o[8 + d];
            out->DF_SurroundingRmsVel[pi] = rms;
        }
    }
    return SHQ_OK;
}

/* ---- black-hole accretion and feedback (blackhole.cpp:217-370) ------------------------------------------------------------- */
namespace {

template <typename T> inline T *bhfield(const shq_bh_slot_view *v, int64_t slot, size_t off)
{
    return reinterpret_cast<T *>(static_cast<char *>(v->base) + (size_t) slot * v->elsize + off);
}

struct BhSet {
    std::vector<int32_t> bhp;   /* particle indices of the type-5 particles, ascending */
    std::vector<BhRec> rec;
    std::vector<int32_t> pi;    /* their slots */
};

int bh_collect(const shq_part_view *parts, const shq_bh_slot_view *bh, const shq_bh_work *work, bool feedback, BhSet &S)
{
    const int64_t n = parts->numpart;
    for(int64_t i = 0; i < n; i++) {
        if(*pfield<uint8_t>(parts, i, parts->off_type) != 5)
            continue;
        if(*pfield<uint8_t>(parts, i, parts->off_flags) & 1u)
            continue;
        const int32_t pi = *pfield<int32_t>(parts, i, parts->off_pi);
        SHQ_CHECK(pi >= 0 && pi < bh->numslots, SHQ_ERR_INVALID, "black hole %ld has PI %d outside the slot array", (long) i, pi);
        BhRec r;
        memset(&r, 0, sizeof(r));
        r.Mass = *bhfield<double>(bh, pi, bh->off_mass);
        r.Density = *bhfield<double>(bh, pi, bh->off_density);
        r.Mtrack = *bhfield<double>(bh, pi, bh->off_mtrack);
        for(int d = 0; d < 3; d++)
            r.DFAccel[d] = bhfield<double>(bh, pi, bh->off_dfaccel)[d];
        r.VDisp = *bhfield<double>(bh, pi, bh->off_vdisp);
        r.KineticFdbkEnergy = *bhfield<double>(bh, pi, bh->off_kineticfdbkenergy);
        r.Mdot = *bhfield<double>(bh, pi, bh->off_mdot);
        r.CountProgs = *bhfield<int32_t>(bh, pi, bh->off_countprogs);
        if(feedback) {
            r.FeedbackWeightSum = work->BH_FeedbackWeightSum[pi];
            r.KEflag = work->KEflag ? work->KEflag[pi] : 0;
        }
        S.bhp.push_back((int32_t) i);
        S.rec.push_back(r);
        S.pi.push_back(pi);
    }
    return SHQ_OK;
}

int bh_check_common(shq_context *ctx, const shq_tree_view *tree, const shq_part_view *parts, const shq_sph_view *sph, const shq_bh_slot_view *bh, const uint64_t *ids,
                    const int32_t *queue, int64_t nqueue, const shq_kick_factors *kf, const shq_bh_params *params, const double *rnd_table, int64_t rnd_size,
                    const shq_bh_work *work, const char *who)
{
    SHQ_CHECK(ctx && tree && parts && sph && bh && ids && kf && params && work && rnd_table && (nqueue == 0 || queue), SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->sphrun.phase == 0, SHQ_ERR_STATE, "%s: an SPH walk is open", who);
    SHQ_CHECK(rnd_size > 0, SHQ_ERR_INVALID, "%s: empty random table", who);
    SHQ_CHECK(work->SPH_SwallowID && work->BH_SwallowID && work->BH_FeedbackWeightSum && work->BH_Entropy && work->BH_SurroundingGasVel && work->MgasEnc,
              SHQ_ERR_INVALID, "%s: the BHPriv arrays are needed", who);
    SHQ_CHECK(params->BlackHoleKineticOn != 1 || work->KEflag, SHQ_ERR_INVALID, "%s: KEflag is needed with BlackHoleKineticOn", who);
    SHQ_CHECK(parts->off_vel != SHQ_NOFIELD && parts->off_hsml != SHQ_NOFIELD && parts->off_pi != SHQ_NOFIELD && parts->off_type != SHQ_NOFIELD &&
                  parts->off_treeacc != SHQ_NOFIELD && parts->off_gravpm != SHQ_NOFIELD && parts->off_flags != SHQ_NOFIELD &&
                  parts->off_timebin_hydro != SHQ_NOFIELD && parts->off_timebin_gravity != SHQ_NOFIELD,
              SHQ_ERR_INVALID, "%s: the particle view needs Vel, Hsml, PI, Type, flags, FullTreeGravAccel, GravPM and both time bins", who);
    const int64_t n = parts->numpart;
    for(int64_t k = 0; k < nqueue; k++) {
        const int32_t i = queue[k];
        SHQ_CHECK(i >= 0 && i < n, SHQ_ERR_INVALID, "%s: queue[%ld] = %d out of range", who, (long) k, i);
        SHQ_CHECK(*pfield<uint8_t>(parts, i, parts->off_type) == 5 && !(*pfield<uint8_t>(parts, i, parts->off_flags) & 3u), SHQ_ERR_INVALID,
                  "%s: particle %d in the queue is not a live black hole (blackhole_haswork)", who, i);
        SHQ_CHECK(*pfield<double>(parts, i, parts->off_hsml) > 0, SHQ_ERR_INVALID, "%s: black hole %d has Hsml <= 0", who, i);
    }
    return SHQ_OK;
}

int bh_upload_common(shq_context *ctx, const shq_tree_view *tree, const shq_part_view *parts, const shq_sph_view *sph, const uint64_t *ids, const double *rnd_table,
                     int64_t rnd_size, const BhSet &S, const std::vector<int32_t> &queue)
{
    SHQ_TRY(shq_particles_upload(ctx, parts));
    SHQ_TRY(sph_upload(ctx, parts, sph));
    SHQ_TRY(shq_tree_upload(ctx, tree));
    const size_t n = (size_t) parts->numpart, nbh = S.bhp.size();
    hipStream_t st = ctx->stream;
    SHQ_TRY(ctx->bhw_bhp.reserve(std::max<size_t>(nbh, 1)));
    SHQ_TRY(ctx->bhw_rec.reserve(std::max<size_t>(nbh, 1) * sizeof(BhRec)));
    SHQ_TRY(ctx->bhw_ids.reserve(std::max<size_t>(n, 1)));
    SHQ_TRY(ctx->bhw_rnd.reserve((size_t) rnd_size));
    SHQ_TRY(ctx->bhw_queue.reserve(std::max<size_t>(queue.size(), 1)));
    SHQ_TRY(ctx->bhw_sphsw.reserve(std::max<size_t>(n, 1)));
    SHQ_TRY(ctx->bhw_bhsw.reserve(std::max<size_t>(nbh, 1)));
    SHQ_TRY(ctx->bhw_swid.reserve(std::max<size_t>(nbh, 1)));
    SHQ_TRY(ctx->bhw_out.reserve(16 * std::max<size_t>(queue.size(), 1)));
    if(nbh) {
        SHQ_HIP(hipMemcpyAsync(ctx->bhw_bhp.ptr, S.bhp.data(), sizeof(int32_t) * nbh, hipMemcpyHostToDevice, st));
        SHQ_HIP(hipMemcpyAsync(ctx->bhw_rec.ptr, S.rec.data(), sizeof(BhRec) * nbh, hipMemcpyHostToDevice, st));
    }
    SHQ_TRY(ids_upload(ctx, ids, (int64_t) n));
    SHQ_HIP(hipMemcpyAsync(ctx->bhw_rnd.ptr, rnd_table, sizeof(double) * (size_t) rnd_size, hipMemcpyHostToDevice, st));
    if(!queue.empty())
        SHQ_HIP(hipMemcpyAsync(ctx->bhw_queue.ptr, queue.data(), sizeof(int32_t) * queue.size(), hipMemcpyHostToDevice, st));
    return SHQ_OK;
}

void bh_fill_args(shq_context *ctx, const shq_bh_params *params, int64_t rnd_size, size_t nbh, BhWalkArgs &w)
{
    memset(&w, 0, sizeof(w));
    w.bhp = ctx->bhw_bhp.ptr;
    w.nbh = (long long) nbh;
    w.bh = reinterpret_cast<BhRec *>(ctx->bhw_rec.ptr);
    w.ids = ctx->bhw_ids.ptr;
    w.rnd = ctx->bhw_rnd.ptr;
    w.rndsize = (unsigned long long) rnd_size;
    w.vel = ctx->vel.ptr;
    w.velw = ctx->vel.ptr;
    w.treeacc = ctx->treeacc.ptr;
    w.gravpm = ctx->gravpm.ptr;
    w.delay = ctx->g_delaytime.ptr;
    w.entropy = ctx->g_entropy.ptr;
    w.density = ctx->g_density.ptr;
    w.bin_grav = ctx->bin_grav.ptr;
    w.bin_hydro = ctx->bin_hydro.ptr;
    w.pflags = ctx->pflags.ptr;
    w.leaf_pidx = ctx->leaf_pidx.ptr;
    w.sph_swallow = ctx->bhw_sphsw.ptr;
    w.bh_swallow = ctx->bhw_bhsw.ptr;
    w.bh_swallowid_out = ctx->bhw_swid.ptr;
    w.out = ctx->bhw_out.ptr;
    w.P = *params;
}

} // namespace

extern "C" int shq_bh_accretion(shq_context *ctx, const shq_tree_view *tree, const shq_part_view *parts, const shq_sph_view *sph, const shq_bh_slot_view *bh,
                                const uint64_t *ids, const int32_t *queue, int64_t nqueue, const shq_kick_factors *kf, const shq_bh_params *params,
                                int64_t Ti_Current, const double *rnd_table, int64_t rnd_size, const shq_bh_work *work)
{
    SHQ_TRY(bh_check_common(ctx, tree, parts, sph, bh, ids, queue, nqueue, kf, params, rnd_table, rnd_size, work, "bh_accretion"));
    SHQ_HIP(hipSetDevice(ctx->device));
    const int64_t n = parts->numpart;
    BhSet S;
    SHQ_TRY(bh_collect(parts, bh, work, false, S));
    const size_t nbh = S.bhp.size();
    std::vector<int32_t> q(queue, queue + nqueue);
    SHQ_TRY(bh_upload_common(ctx, tree, parts, sph, ids, rnd_table, rnd_size, S, q));
    hipStream_t st = ctx->stream;
    if(n)
        SHQ_HIP(hipMemsetAsync(ctx->bhw_sphsw.ptr, 0, sizeof(uint64_t) * (size_t) n, st));
    if(nbh)
        SHQ_HIP(hipMemsetAsync(ctx->bhw_bhsw.ptr, 0, sizeof(uint64_t) * nbh, st));
    /* priv->SPH_SwallowID / BH_SwallowID start from zero (blackhole.cpp:270-274) */
    memset(work->SPH_SwallowID, 0, sizeof(uint64_t) * (size_t) sph->numslots);
    memset(work->BH_SwallowID, 0, sizeof(uint64_t) * (size_t) bh->numslots);
    if(nqueue == 0)
        return SHQ_OK;
    BhWalkArgs w;
    bh_fill_args(ctx, params, rnd_size, nbh, w);
    w.Ti_Current = Ti_Current;
    /* the gas particles the walk marks come back as a list (a few thousand of N) */
    SHQ_TRY(ctx->bhw_touched.reserve((size_t) std::max<int64_t>(n, 1)));
    SHQ_TRY(ctx->bhw_tlist.reserve((size_t) std::max<int64_t>(n, 1)));
    SHQ_HIP(hipMemsetAsync(ctx->bhw_touched.ptr, 0, (size_t) n, st));
    w.touched = ctx->bhw_touched.ptr;
    double *d_post = ctx->bhw_out.ptr + 8 * (size_t) nqueue;
    SHQ_TRY(shq_bh_accretion_device(ctx, kf, &w, ctx->bhw_queue.ptr, nqueue, d_post));
    int64_t nt = 0;
    SHQ_TRY(shq_marked_list(ctx, ctx->bhw_touched.ptr, n, ctx->bhw_tlist.ptr, &nt));
    SHQ_TRY(ctx->bhw_trows.reserve((size_t) std::max<int64_t>(nt, 1)));
    unsigned long long *d_marks = reinterpret_cast<unsigned long long *>(ctx->bhw_trows.ptr);
    std::vector<double> out(16 * (size_t) nqueue);
    std::vector<uint64_t> marks((size_t) std::max<int64_t>(nt, 1)), bhsw(std::max<size_t>(nbh, 1));
    std::vector<int32_t> tl((size_t) std::max<int64_t>(nt, 1));
    SHQ_HIP(hipMemcpyAsync(out.data(), ctx->bhw_out.ptr, sizeof(double) * out.size(), hipMemcpyDeviceToHost, st));
    SHQ_HIP(hipMemcpyAsync(S.rec.data(), ctx->bhw_rec.ptr, sizeof(BhRec) * nbh, hipMemcpyDeviceToHost, st));
    if(nt) {
        SHQ_TRY(shq_u64_gather(ctx, ctx->bhw_tlist.ptr, nt, ctx->bhw_sphsw.ptr, d_marks));
        SHQ_HIP(hipMemcpyAsync(marks.data(), d_marks, sizeof(uint64_t) * (size_t) nt, hipMemcpyDeviceToHost, st));
        SHQ_HIP(hipMemcpyAsync(tl.data(), ctx->bhw_tlist.ptr, sizeof(int32_t) * (size_t) nt, hipMemcpyDeviceToHost, st));
    }
    SHQ_HIP(hipMemcpyAsync(bhsw.data(), ctx->bhw_bhsw.ptr, sizeof(uint64_t) * nbh, hipMemcpyDeviceToHost, st));
    SHQ_HIP(hipStreamSynchronize(st));
    for(int64_t t = 0; t < nt; t++) {
        const int64_t i = tl[(size_t) t];
        SHQ_CHECK(*pfield<uint8_t>(parts, i, parts->off_type) == 0, SHQ_ERR_STATE, "bh_accretion: a swallow mark on a particle that is not gas");
        work->SPH_SwallowID[*pfield<int32_t>(parts, i, parts->off_pi)] = marks[(size_t) t];
    }
    for(size_t b = 0; b < nbh; b++)
        if(bhsw[b])
            work->BH_SwallowID[S.pi[b]] = bhsw[b];
    for(int64_t t = 0; t < nqueue; t++) {
        const size_t b = (size_t) (std::lower_bound(S.bhp.begin(), S.bhp.end(), queue[t]) - S.bhp.begin());
        const int32_t pi = S.pi[b];
        const double *o = &out[8 * (size_t) t], *p = &out[8 * (size_t) nqueue + 8 * (size_t) t];
        const BhRec &R = S.rec[b];
        *bhfield<int8_t>(bh, pi, bh->off_encounter) = (int8_t) o[0];          /* blackhole_accretion_reduce, PRIMARY */
        work->BH_FeedbackWeightSum[pi] = o[1];
        work->MgasEnc[pi] = o[6];
        work->BH_Entropy[pi] = p[0];
        for(int d = 0; d < 3; d++) {
            work->BH_SurroundingGasVel[pi][d] = p[1 + d];
            bhfield<double>(bh, pi, bh->off_dragaccel)[d] = p[4 + d];
        }
        *bhfield<double>(bh, pi, bh->off_mdot) = R.Mdot;
        *bhfield<double>(bh, pi, bh->off_mass) = R.Mass;
        *bhfield<double>(bh, pi, bh->off_kineticfdbkenergy) = R.KineticFdbkEnergy;
        if(work->KEflag)
            work->KEflag[pi] = R.KEflag;
    }
    return SHQ_OK;
}

extern "C" int shq_bh_feedback(shq_context *ctx, const shq_tree_view *tree, const shq_part_view *parts, const shq_sph_view *sph, const shq_bh_slot_view *bh,
                               const uint64_t *ids, const int32_t *queue, int64_t nqueue, const shq_kick_factors *kf, const shq_bh_params *params, int64_t MaxPart,
                               const double *rnd_table, int64_t rnd_size, const uint8_t *eeqos, const shq_bh_work *work, int64_t *n_sph_swallowed,
                               int64_t *n_bh_swallowed)
{
    SHQ_TRY(bh_check_common(ctx, tree, parts, sph, bh, ids, queue, nqueue, kf, params, rnd_table, rnd_size, work, "bh_feedback"));
    SHQ_CHECK(work->BH_accreted_Mass && work->BH_accreted_BHMass && work->BH_accreted_momentum, SHQ_ERR_INVALID, "bh_feedback: the accreted-mass arrays are needed");
    SHQ_HIP(hipSetDevice(ctx->device));
    const int64_t n = parts->numpart;
    if(n_sph_swallowed)
        *n_sph_swallowed = 0;
    if(n_bh_swallowed)
        *n_bh_swallowed = 0;
    BhSet S;
    SHQ_TRY(bh_collect(parts, bh, work, true, S));
    const size_t nbh = S.bhp.size();
    /* blackhole_feedback_haswork (:878-883): a hole that is marked itself does not swallow or heat */
    std::vector<int32_t> q;
    for(int64_t k = 0; k < nqueue; k++)
        if(work->BH_SwallowID[*pfield<int32_t>(parts, queue[k], parts->off_pi)] == 0)
            q.push_back(queue[k]);
    SHQ_TRY(bh_upload_common(ctx, tree, parts, sph, ids, rnd_table, rnd_size, S, q));
    hipStream_t st = ctx->stream;
    std::vector<uint64_t> sphsw((size_t) std::max<int64_t>(n, 1), 0), bhsw(std::max<size_t>(nbh, 1), 0), swid(std::max<size_t>(nbh, 1));
    for(int64_t i = 0; i < n; i++)
        if(*pfield<uint8_t>(parts, i, parts->off_type) == 0 && !(*pfield<uint8_t>(parts, i, parts->off_flags) & 1u)) {
            const int32_t pi = *pfield<int32_t>(parts, i, parts->off_pi);
            SHQ_CHECK(pi >= 0 && pi < sph->numslots, SHQ_ERR_INVALID, "bh_feedback: gas particle %ld has PI %d outside the slot array", (long) i, pi);
            sphsw[i] = work->SPH_SwallowID[pi];
        }
    for(size_t b = 0; b < nbh; b++)
        bhsw[b] = work->BH_SwallowID[S.pi[b]];
    SHQ_HIP(hipMemcpyAsync(ctx->bhw_sphsw.ptr, sphsw.data(), sizeof(uint64_t) * (size_t) n, hipMemcpyHostToDevice, st));
    if(nbh) {
        SHQ_HIP(hipMemcpyAsync(ctx->bhw_bhsw.ptr, bhsw.data(), sizeof(uint64_t) * nbh, hipMemcpyHostToDevice, st));
        SHQ_HIP(hipMemsetAsync(ctx->bhw_swid.ptr, 0xff, sizeof(uint64_t) * nbh, st));
    }
    SHQ_TRY(ctx->bhw_eeqos.reserve((size_t) std::max<int64_t>(n, 1)));
    SHQ_TRY(ctx->bhw_heated.reserve((size_t) std::max<int64_t>(n, 1)));
    SHQ_TRY(ctx->bhw_touched.reserve((size_t) std::max<int64_t>(n, 1)));
    SHQ_TRY(ctx->bhw_tlist.reserve((size_t) std::max<int64_t>(n, 1)));
    /* what the walk changes on the neighbours' side comes back as rows of the particles it touched (and of the holes), not as whole
     * arrays: at 2 x 10^6 particles the five array downloads and the loop over every gas particle were most of this call */
    std::vector<double> bh0(8 * std::max<size_t>(nbh, 1));
    if(n) {
        if(eeqos)
            SHQ_HIP(hipMemcpyAsync(ctx->bhw_eeqos.ptr, eeqos, (size_t) n, hipMemcpyHostToDevice, st));
        SHQ_HIP(hipMemsetAsync(ctx->bhw_heated.ptr, 0, (size_t) n, st));
        SHQ_HIP(hipMemsetAsync(ctx->bhw_touched.ptr, 0, (size_t) n, st));
    }
    SHQ_TRY(ctx->bhw_trows.reserve(8 * std::max<size_t>(nbh, 1)));
    if(nbh) { /* the holes' flags before the walk */
        SHQ_TRY(shq_rows_gather(ctx, ctx->bhw_bhp.ptr, (int64_t) nbh, nullptr, ctx->bhw_trows.ptr));
        SHQ_HIP(hipMemcpyAsync(bh0.data(), ctx->bhw_trows.ptr, sizeof(double) * 8 * nbh, hipMemcpyDeviceToHost, st));
    }
    const int64_t nq = (int64_t) q.size();
    BhWalkArgs w;
    bh_fill_args(ctx, params, rnd_size, nbh, w);
    w.eeqos = eeqos ? ctx->bhw_eeqos.ptr : nullptr;
    w.heated = ctx->bhw_heated.ptr;
    w.touched = ctx->bhw_touched.ptr;
    SHQ_TRY(shq_bh_feedback_device(ctx, kf, &w, ctx->bhw_queue.ptr, nq));
    int64_t nt = 0;
    SHQ_TRY(shq_marked_list(ctx, ctx->bhw_touched.ptr, n, ctx->bhw_tlist.ptr, &nt));
    SHQ_TRY(ctx->bhw_trows.reserve(8 * (size_t) std::max<int64_t>(nt + (int64_t) nbh, 1)));
    std::vector<double> out(8 * (size_t) std::max<int64_t>(nq, 1)), rows(8 * (size_t) std::max<int64_t>(nt, 1)), bh1(8 * std::max<size_t>(nbh, 1));
    std::vector<int32_t> tl((size_t) std::max<int64_t>(nt, 1));
    if(nq)
        SHQ_HIP(hipMemcpyAsync(out.data(), ctx->bhw_out.ptr, sizeof(double) * 8 * (size_t) nq, hipMemcpyDeviceToHost, st));
    if(nt) {
        SHQ_TRY(shq_rows_gather(ctx, ctx->bhw_tlist.ptr, nt, ctx->bhw_heated.ptr, ctx->bhw_trows.ptr));
        SHQ_HIP(hipMemcpyAsync(rows.data(), ctx->bhw_trows.ptr, sizeof(double) * 8 * (size_t) nt, hipMemcpyDeviceToHost, st));
        SHQ_HIP(hipMemcpyAsync(tl.data(), ctx->bhw_tlist.ptr, sizeof(int32_t) * (size_t) nt, hipMemcpyDeviceToHost, st));
    }
    if(nbh) {
        SHQ_TRY(shq_rows_gather(ctx, ctx->bhw_bhp.ptr, (int64_t) nbh, nullptr, ctx->bhw_trows.ptr + 8 * (size_t) nt));
        SHQ_HIP(hipMemcpyAsync(bh1.data(), ctx->bhw_trows.ptr + 8 * (size_t) nt, sizeof(double) * 8 * nbh, hipMemcpyDeviceToHost, st));
        SHQ_HIP(hipMemcpyAsync(swid.data(), ctx->bhw_swid.ptr, sizeof(uint64_t) * nbh, hipMemcpyDeviceToHost, st));
        SHQ_HIP(hipMemcpyAsync(S.rec.data(), ctx->bhw_rec.ptr, sizeof(BhRec) * nbh, hipMemcpyDeviceToHost, st));
    }
    SHQ_HIP(hipStreamSynchronize(st));
    /* the neighbours' side of the walk, into the caller's arrays: only tree particles that were gas and no garbage are ever touched */
    int64_t nsph = 0, nbhs = 0;
    for(int64_t t = 0; t < nt; t++) {
        const int64_t i = tl[(size_t) t];
        const double *r = &rows[8 * (size_t) t];
        SHQ_CHECK(*pfield<uint8_t>(parts, i, parts->off_type) == 0, SHQ_ERR_STATE, "bh_feedback: the walk touched a particle that is not gas");
        const int32_t pi = *pfield<int32_t>(parts, i, parts->off_pi);
        *sfield(sph, pi, sph->off_entropy) = r[3];
        double *v = pfield_w<double>(parts, i, parts->off_vel);
        for(int d = 0; d < 3; d++)
            v[d] = r[d];
        if(r[7] != 0)
            *pfield_w<uint8_t>(parts, i, parts->off_flags) |= 8u; /* BHHeated: bit 3 of the flag byte */
        if(((unsigned) r[6] & 1u) && !(*pfield<uint8_t>(parts, i, parts->off_flags) & 1u)) { /* slots_mark_garbage, slotsmanager.cpp:590-599 */
            *pfield_w<uint8_t>(parts, i, parts->off_flags) |= 1u;
            *reinterpret_cast<int32_t *>(static_cast<char *>(sph->base) + (size_t) pi * sph->elsize) = (int32_t) (MaxPart + 100);
            nsph++;
        }
    }
    for(size_t b = 0; b < nbh; b++) {
        const int64_t i = S.bhp[b];
        if(((unsigned) bh1[8 * b + 6] & 2u) && !((unsigned) bh0[8 * b + 6] & 2u)) {
            const int32_t pi = S.pi[b];
            *pfield_w<uint8_t>(parts, i, parts->off_flags) |= 2u; /* Swallowed */
            *bhfield<uint64_t>(bh, pi, bh->off_swallowid) = swid[b];
            *bhfield<double>(bh, pi, bh->off_swallowtime) = params->atime;
            *bhfield<int8_t>(bh, pi, bh->off_encounter) = 0;
            nbhs++;
        }
    }
    /* blackhole_feedback_reduce (PRIMARY); blackhole_feedback_postprocess (:929-965) ran on the device: its results come back */
    for(int64_t t = 0; t < nq; t++) {
        const int32_t i = q[t];
        const size_t b = (size_t) (std::lower_bound(S.bhp.begin(), S.bhp.end(), i) - S.bhp.begin());
        const int32_t pi = S.pi[b];
        const double *o = &out[8 * (size_t) t];
        const BhRec &R = S.rec[b];
        work->BH_accreted_Mass[pi] = o[0];
        work->BH_accreted_BHMass[pi] = o[1];
        for(int d = 0; d < 3; d++)
            work->BH_accreted_momentum[pi][d] = o[2 + d];
        *bhfield<uint8_t>(bh, pi, bh->off_mintimebin) = (uint8_t) o[6];
        *bhfield<int32_t>(bh, pi, bh->off_countprogs) = R.CountProgs;
        *bhfield<double>(bh, pi, bh->off_mass) = R.Mass;
        *bhfield<double>(bh, pi, bh->off_mtrack) = R.Mtrack;
        *bhfield<double>(bh, pi, bh->off_kineticfdbkenergy) = R.KineticFdbkEnergy;
        double *v = pfield_w<double>(parts, i, parts->off_vel);
        for(int d = 0; d < 3; d++)
            v[d] = bh1[8 * b + d];
        *pfield_w<float>(parts, i, parts->off_mass) = (float) bh1[8 * b + 5];
    }
    if(n_sph_swallowed)
        *n_sph_swallowed = nsph;
    if(n_bh_swallowed)
        *n_bh_swallowed = nbhs;
    return SHQ_OK;
}

/* ---- winds_and_feedback (winds.cpp:295-369) ------------------------------------------------------------------------------------ */
namespace {
/* Vel, Entropy, DelayTime of the kicked particles (the first of every run of the sorted list) back into the caller's arrays */
int winds_kicked_back(shq_context *ctx, const shq_part_view *parts, const shq_sph_view *sph, const std::vector<shq_wind_kick> &K)
{
    hipStream_t st = ctx->stream;
    /* rows of the kicked particles only (the arrays are N long, the kicked a few per cent of the gas) */
    std::vector<int32_t> list;
    int32_t last = -1;
    for(const shq_wind_kick &k : K)
        if(k.part_index != last)
            list.push_back(last = k.part_index);
    const size_t m = list.size();
    if(m == 0)
        return SHQ_OK;
    SHQ_TRY(ctx->bhw_tlist.reserve(m));
    SHQ_TRY(ctx->bhw_trows.reserve(8 * m));
    std::vector<double> rows(8 * m);
    SHQ_HIP(hipMemcpyAsync(ctx->bhw_tlist.ptr, list.data(), sizeof(int32_t) * m, hipMemcpyHostToDevice, st));
    SHQ_TRY(shq_rows_gather(ctx, ctx->bhw_tlist.ptr, (int64_t) m, nullptr, ctx->bhw_trows.ptr));
    SHQ_HIP(hipMemcpyAsync(rows.data(), ctx->bhw_trows.ptr, sizeof(double) * 8 * m, hipMemcpyDeviceToHost, st));
    SHQ_HIP(hipStreamSynchronize(st));
    for(size_t t = 0; t < m; t++) {
        const int32_t other = list[t];
        const double *r = &rows[8 * t];
        const int32_t pi = *pfield<int32_t>(parts, other, parts->off_pi);
        double *v = pfield_w<double>(parts, other, parts->off_vel);
        for(int j = 0; j < 3; j++)
            v[j] = r[j];
        *sfield(sph, pi, sph->off_entropy) = r[3];
        *sfield(sph, pi, sph->off_delaytime) = r[4];
    }
    return SHQ_OK;
}

int winds_impl(shq_context *ctx, const shq_tree_view *tree, const shq_part_view *parts, const shq_sph_view *sph, const shq_star_view *stars,
                                      const uint64_t *ids, const int32_t *NewStars, int64_t NumNewStars, const shq_wind_params *params, const double *rnd_table,
                                      int64_t rnd_size, double *TotalWeight, shq_wind_kick *kicks, int64_t kicks_capacity, int64_t *nkicks, int64_t *nkicked, bool apply)
{
    SHQ_CHECK(ctx && tree && parts && sph && stars && ids && params && rnd_table && (NumNewStars == 0 || NewStars), SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->sphrun.phase == 0, SHQ_ERR_STATE, "winds_and_feedback: an SPH walk is open");
    SHQ_CHECK(rnd_size > 0, SHQ_ERR_INVALID, "winds_and_feedback: empty random table");
    if(nkicks)
        *nkicks = 0;
    if(nkicked)
        *nkicked = 0;
    if(params->WindModel & 1) /* "The subgrid model does nothing here" */
        return SHQ_OK;
    SHQ_CHECK((params->WindModel & 8) || (params->WindModel & 4), SHQ_ERR_INVALID, "WindModel = 0x%X is strange (winds.cpp:503)", params->WindModel);
    SHQ_CHECK(parts->off_vel != SHQ_NOFIELD && parts->off_hsml != SHQ_NOFIELD && parts->off_pi != SHQ_NOFIELD && parts->off_type != SHQ_NOFIELD &&
                  parts->off_flags != SHQ_NOFIELD,
              SHQ_ERR_INVALID, "winds_and_feedback: the particle view needs Vel, Hsml, PI, Type and the flag byte");
    SHQ_CHECK(sph->off_delaytime != SHQ_NOFIELD, SHQ_ERR_INVALID, "winds_and_feedback: the gas view needs DelayTime");
    SHQ_HIP(hipSetDevice(ctx->device));
    const int64_t n = parts->numpart, nq = NumNewStars;
    std::vector<double> vdisp((size_t) std::max<int64_t>(nq, 1));
    for(int64_t k = 0; k < nq; k++) {
        const int32_t i = NewStars[k];
        SHQ_CHECK(i >= 0 && i < n, SHQ_ERR_INVALID, "winds_and_feedback: NewStars[%ld] = %d out of range", (long) k, i);
        SHQ_CHECK(*pfield<uint8_t>(parts, i, parts->off_type) == 4, SHQ_ERR_INVALID, "Particle %d has type %d not a star (winds.cpp:402)", i,
                  (int) *pfield<uint8_t>(parts, i, parts->off_type));
        const int32_t pi = *pfield<int32_t>(parts, i, parts->off_pi);
        SHQ_CHECK(pi >= 0 && pi < stars->numslots, SHQ_ERR_INVALID, "winds_and_feedback: star %d has PI %d outside the slot array", i, pi);
        vdisp[(size_t) k] = (double) *reinterpret_cast<const float *>(static_cast<const char *>(stars->base) + (size_t) pi * stars->elsize + stars->off_vdisp);
    }
    SHQ_TRY(shq_particles_upload(ctx, parts));
    SHQ_TRY(sph_upload(ctx, parts, sph));
    SHQ_TRY(shq_tree_upload(ctx, tree));
    if(nq == 0)
        return SHQ_OK;
    hipStream_t st = ctx->stream;
    SHQ_TRY(ctx->bhw_ids.reserve((size_t) std::max<int64_t>(n, 1)));
    SHQ_TRY(ctx->bhw_rnd.reserve((size_t) rnd_size));
    SHQ_TRY(ctx->bhw_queue.reserve((size_t) nq));
    SHQ_TRY(ctx->wind_d.reserve(2 * (size_t) nq));
    SHQ_TRY(ctx->wind_cnt.reserve(4));
    SHQ_TRY(ids_upload(ctx, ids, (int64_t) n));
    SHQ_HIP(hipMemcpyAsync(ctx->bhw_rnd.ptr, rnd_table, sizeof(double) * (size_t) rnd_size, hipMemcpyHostToDevice, st));
    SHQ_HIP(hipMemcpyAsync(ctx->bhw_queue.ptr, NewStars, sizeof(int32_t) * (size_t) nq, hipMemcpyHostToDevice, st));
    SHQ_HIP(hipMemcpyAsync(ctx->wind_d.ptr + nq, vdisp.data(), sizeof(double) * (size_t) nq, hipMemcpyHostToDevice, st));
    SHQ_HIP(hipMemsetAsync(ctx->wind_cnt.ptr, 0, sizeof(unsigned long long) * 4, st));
    WindWalkArgs w;
    memset(&w, 0, sizeof(w));
    w.ids = ctx->bhw_ids.ptr;
    w.rnd = ctx->bhw_rnd.ptr;
    w.rndsize = (unsigned long long) rnd_size;
    w.leaf_pidx = ctx->leaf_pidx.ptr;
    w.totalweight = ctx->wind_d.ptr;
    w.vdisp = ctx->wind_d.ptr + nq;
    w.nvisited = ctx->wind_cnt.ptr;
    w.nkicks = ctx->wind_cnt.ptr + 1;
    w.P = *params;
    SHQ_TRY(shq_wind_walk_device(ctx, &w, ctx->bhw_queue.ptr, nq, false));
    std::vector<double> tw((size_t) nq);
    unsigned long long cnt[2] = {0, 0};
    SHQ_HIP(hipMemcpyAsync(tw.data(), ctx->wind_d.ptr, sizeof(double) * (size_t) nq, hipMemcpyDeviceToHost, st));
    SHQ_HIP(hipMemcpyAsync(cnt, ctx->wind_cnt.ptr, sizeof(cnt), hipMemcpyDeviceToHost, st));
    SHQ_HIP(hipStreamSynchronize(st));
    if(TotalWeight)
        for(int64_t k = 0; k < nq; k++) /* sfr_wind_reduce_weight, PRIMARY */
            TotalWeight[*pfield<int32_t>(parts, NewStars[k], parts->off_pi)] = tw[(size_t) k];
    /* priv->maxkicks = nvisited + 2 (:311): every candidate was a visit of the first walk */
    const unsigned long long maxkicks = cnt[0] + 2;
    SHQ_TRY(ctx->wind_kicks.reserve((size_t) maxkicks * sizeof(shq_wind_kick)));
    w.kicks = reinterpret_cast<shq_wind_kick *>(ctx->wind_kicks.ptr);
    w.maxkicks = maxkicks;
    SHQ_TRY(shq_wind_walk_device(ctx, &w, ctx->bhw_queue.ptr, nq, true));
    SHQ_HIP(hipMemcpyAsync(cnt, ctx->wind_cnt.ptr, sizeof(cnt), hipMemcpyDeviceToHost, st));
    SHQ_HIP(hipStreamSynchronize(st));
    SHQ_CHECK(cnt[1] <= maxkicks, SHQ_ERR_STATE, "Not enough room in kick queue: %llu > %llu (winds.cpp:552)", cnt[1], maxkicks);
    const long long nk = (long long) cnt[1];
    if(nkicks)
        *nkicks = nk;
    if(kicks)
        SHQ_CHECK(kicks_capacity >= nk, SHQ_ERR_INVALID, "winds_and_feedback: room for %ld kicks, %ld found", (long) kicks_capacity, (long) nk);
    int64_t applied = 0;
    if(nk > 0) {
        /* the candidates sorted with StarKick's comparison, the first of every particle kicks: on the device, into the resident
         * Vel / Entropy / DelayTime, which then go back into the caller's arrays */
        SHQ_TRY(ctx->bhw_rec.reserve((size_t) nk * sizeof(shq_wind_kick)));
        shq_wind_kick *d_sorted = reinterpret_cast<shq_wind_kick *>(ctx->bhw_rec.ptr);
        SHQ_HIP(hipMemsetAsync(ctx->wind_cnt.ptr + 2, 0, sizeof(unsigned long long) * 2, st));
        SHQ_TRY(shq_wind_resolve_device(ctx, &w, nk, d_sorted, ctx->wind_cnt.ptr + 2, reinterpret_cast<int *>(ctx->wind_cnt.ptr + 3), apply));
        unsigned long long res[2] = {0, 0};
        SHQ_HIP(hipMemcpyAsync(res, ctx->wind_cnt.ptr + 2, sizeof(res), hipMemcpyDeviceToHost, st));
        std::vector<shq_wind_kick> K((size_t) nk);
        SHQ_HIP(hipMemcpyAsync(K.data(), d_sorted, sizeof(shq_wind_kick) * (size_t) nk, hipMemcpyDeviceToHost, st));
        SHQ_HIP(hipStreamSynchronize(st));
        applied = (int64_t) res[0];
        if(kicks)
            memcpy(kicks, K.data(), sizeof(shq_wind_kick) * (size_t) nk);
        if(apply)
            SHQ_TRY(winds_kicked_back(ctx, parts, sph, K));
        SHQ_CHECK((int) (res[1] & 0xffffffffull) == 0, SHQ_ERR_STATE, "Odd v in a wind kick (winds.cpp:344)");
    }
    if(nkicked)
        *nkicked = applied;
    return SHQ_OK;
}
} // namespace

extern "C" int shq_winds_and_feedback(shq_context *ctx, const shq_tree_view *tree, const shq_part_view *parts, const shq_sph_view *sph, const shq_star_view *stars,
                                      const uint64_t *ids, const int32_t *NewStars, int64_t NumNewStars, const shq_wind_params *params, const double *rnd_table,
                                      int64_t rnd_size, double *TotalWeight, shq_wind_kick *kicks, int64_t kicks_capacity, int64_t *nkicks, int64_t *nkicked)
{
    return winds_impl(ctx, tree, parts, sph, stars, ids, NewStars, NumNewStars, params, rnd_table, rnd_size, TotalWeight, kicks, kicks_capacity, nkicks, nkicked, true);
}

extern "C" int shq_winds_candidates(shq_context *ctx, const shq_tree_view *tree, const shq_part_view *parts, const shq_sph_view *sph, const shq_star_view *stars,
                                    const uint64_t *ids, const int32_t *NewStars, int64_t NumNewStars, const shq_wind_params *params, const double *rnd_table,
                                    int64_t rnd_size, double *TotalWeight, shq_wind_kick *kicks, int64_t kicks_capacity, int64_t *nkicks)
{
    return winds_impl(ctx, tree, parts, sph, stars, ids, NewStars, NumNewStars, params, rnd_table, rnd_size, TotalWeight, kicks, kicks_capacity, nkicks, nullptr, false);
}

extern "C" int shq_winds_apply(shq_context *ctx, const shq_part_view *parts, const shq_sph_view *sph, const uint64_t *ids, const shq_wind_kick *kicks, int64_t nk,
                               const shq_wind_params *params, const double *rnd_table, int64_t rnd_size, int64_t *nkicked)
{
    SHQ_CHECK(ctx && parts && sph && ids && params && rnd_table && (nk == 0 || kicks), SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(rnd_size > 0 && nk >= 0, SHQ_ERR_INVALID, "winds_apply: empty random table or bad list length");
    SHQ_CHECK(sph->off_delaytime != SHQ_NOFIELD && parts->off_vel != SHQ_NOFIELD && parts->off_pi != SHQ_NOFIELD && parts->off_type != SHQ_NOFIELD, SHQ_ERR_INVALID,
              "winds_apply: needs Vel, PI, Type and DelayTime");
    if(nkicked)
        *nkicked = 0;
    const int64_t n = parts->numpart;
    for(int64_t k = 0; k < nk; k++) {
        SHQ_CHECK(kicks[k].part_index >= 0 && kicks[k].part_index < n && *pfield<uint8_t>(parts, kicks[k].part_index, parts->off_type) == 0, SHQ_ERR_INVALID,
                  "winds_apply: kick %ld names particle %d, which is not a gas particle of this set", (long) k, kicks[k].part_index);
        SHQ_CHECK(kicks[k].StarDistance >= 0, SHQ_ERR_INVALID, "winds_apply: negative distance in kick %ld", (long) k);
    }
    if(nk == 0)
        return SHQ_OK;
    SHQ_HIP(hipSetDevice(ctx->device));
    SHQ_TRY(shq_particles_upload(ctx, parts));
    SHQ_TRY(sph_upload(ctx, parts, sph));
    hipStream_t st = ctx->stream;
    SHQ_TRY(ctx->bhw_ids.reserve((size_t) n));
    SHQ_TRY(ctx->bhw_rnd.reserve((size_t) rnd_size));
    SHQ_TRY(ctx->wind_cnt.reserve(4));
    SHQ_TRY(ctx->wind_kicks.reserve((size_t) nk * sizeof(shq_wind_kick)));
    SHQ_TRY(ctx->bhw_rec.reserve((size_t) nk * sizeof(shq_wind_kick)));
    SHQ_TRY(ids_upload(ctx, ids, (int64_t) n));
    SHQ_HIP(hipMemcpyAsync(ctx->bhw_rnd.ptr, rnd_table, sizeof(double) * (size_t) rnd_size, hipMemcpyHostToDevice, st));
    SHQ_HIP(hipMemcpyAsync(ctx->wind_kicks.ptr, kicks, sizeof(shq_wind_kick) * (size_t) nk, hipMemcpyHostToDevice, st));
    SHQ_HIP(hipMemsetAsync(ctx->wind_cnt.ptr, 0, sizeof(unsigned long long) * 4, st));
    WindWalkArgs w;
    memset(&w, 0, sizeof(w));
    w.ids = ctx->bhw_ids.ptr;
    w.rnd = ctx->bhw_rnd.ptr;
    w.rndsize = (unsigned long long) rnd_size;
    w.kicks = reinterpret_cast<shq_wind_kick *>(ctx->wind_kicks.ptr);
    w.P = *params;
    shq_wind_kick *d_sorted = reinterpret_cast<shq_wind_kick *>(ctx->bhw_rec.ptr);
    SHQ_TRY(shq_wind_resolve_device(ctx, &w, nk, d_sorted, ctx->wind_cnt.ptr + 2, reinterpret_cast<int *>(ctx->wind_cnt.ptr + 3), true));
    unsigned long long res[2] = {0, 0};
    SHQ_HIP(hipMemcpyAsync(res, ctx->wind_cnt.ptr + 2, sizeof(res), hipMemcpyDeviceToHost, st));
    std::vector<shq_wind_kick> K((size_t) nk);
    SHQ_HIP(hipMemcpyAsync(K.data(), d_sorted, sizeof(shq_wind_kick) * (size_t) nk, hipMemcpyDeviceToHost, st));
    SHQ_HIP(hipStreamSynchronize(st));
    SHQ_TRY(winds_kicked_back(ctx, parts, sph, K));
    SHQ_CHECK((int) (res[1] & 0xffffffffull) == 0, SHQ_ERR_STATE, "Odd v in a wind kick (winds.cpp:344)");
    if(nkicked)
        *nkicked = (int64_t) res[0];
    return SHQ_OK;
}

/* ---- metal_return's treewalk (metal_return.cpp:513-530, 573-667) ---------------------------------------------------------------- */
extern "C" int shq_metal_return(shq_context *ctx, const shq_tree_view *tree, const shq_part_view *parts, const shq_gas_metal_view *gas, const int32_t *queue,
                                int64_t nqueue, const double *StarVolumeSPH, const double *MassGenerated, const double *MetalGenerated,
                                const double *MetalSpeciesGenerated, double MaxGasMass, int SPHWeighting, int DensityKernelType, double *MassReturn, int64_t *npairs)
{
    SHQ_CHECK(ctx && tree && parts && gas && (nqueue == 0 || (queue && StarVolumeSPH && MassGenerated && MetalGenerated && MetalSpeciesGenerated && MassReturn)),
              SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->sphrun.phase == 0, SHQ_ERR_STATE, "metal_return: an SPH walk is open");
    SHQ_CHECK(gas->nmetals == SHQ_NMETALS, SHQ_ERR_INVALID, "metal_return: NMETALS = %d, the library was built for %d", gas->nmetals, SHQ_NMETALS);
    SHQ_CHECK(parts->off_hsml != SHQ_NOFIELD && parts->off_pi != SHQ_NOFIELD && parts->off_type != SHQ_NOFIELD && parts->off_flags != SHQ_NOFIELD, SHQ_ERR_INVALID,
              "metal_return: the particle view needs Hsml, PI, Type and the flag byte");
    if(npairs)
        *npairs = 0;
    SHQ_HIP(hipSetDevice(ctx->device));
    const int64_t n = parts->numpart, nq = nqueue;
    for(int64_t k = 0; k < nq; k++) {
        const int32_t i = queue[k];
        SHQ_CHECK(i >= 0 && i < n, SHQ_ERR_INVALID, "metal_return: queue[%ld] = %d out of range", (long) k, i);
        SHQ_CHECK(*pfield<uint8_t>(parts, i, parts->off_type) == 4 && !(*pfield<uint8_t>(parts, i, parts->off_flags) & 1u), SHQ_ERR_INVALID,
                  "metal_return: particle %d in the queue is not a live star", i);
        /* "StarVolumeSPH %g hsml %g" (:619): the reference ends the run at the first neighbour of such a star */
        SHQ_CHECK(StarVolumeSPH[k] != 0, SHQ_ERR_INVALID, "metal_return: StarVolumeSPH = 0 for star %d (hsml %g)", i, *pfield<double>(parts, i, parts->off_hsml));
    }
    SHQ_TRY(shq_particles_upload(ctx, parts));
    SHQ_TRY(shq_dynamics_upload(ctx, parts));
    SHQ_TRY(shq_tree_upload(ctx, tree));
    if(nq == 0)
        return SHQ_OK;
    hipStream_t st = ctx->stream;
    const size_t cap = (size_t) std::max<int64_t>(n, 1);
    std::vector<float> gf(cap * (1 + SHQ_NMETALS), 0.f);
    std::vector<double> gd(cap * 2, 1.0);
    float *hm = gf.data(), *hz = gf.data() + cap;
    double *hd = gd.data(), *hy = gd.data() + cap;
    auto slot = [&](int64_t i) { return static_cast<char *>(gas->base) + (size_t) *pfield<int32_t>(parts, i, parts->off_pi) * gas->elsize; };
    for(int64_t i = 0; i < n; i++) {
        hm[i] = *pfield<float>(parts, i, parts->off_mass);
        if(*pfield<uint8_t>(parts, i, parts->off_type) != 0 || (*pfield<uint8_t>(parts, i, parts->off_flags) & 1u))
            continue;
        const int32_t pi = *pfield<int32_t>(parts, i, parts->off_pi);
        SHQ_CHECK(pi >= 0 && pi < gas->numslots, SHQ_ERR_INVALID, "metal_return: gas particle %ld has PI %d outside the slot array", (long) i, pi);
        const char *s = slot(i);
        hd[i] = *reinterpret_cast<const double *>(s + gas->off_density);
        hy[i] = *reinterpret_cast<const double *>(s + gas->off_metallicity);
        for(int m = 0; m < SHQ_NMETALS; m++)
            hz[(size_t) i * SHQ_NMETALS + m] = reinterpret_cast<const float *>(s + gas->off_metals)[m];
    }
    SHQ_TRY(ctx->metal_gf.reserve(gf.size()));
    SHQ_TRY(ctx->metal_gd.reserve(gd.size()));
    SHQ_TRY(ctx->metal_star.reserve((size_t) nq * (4 + SHQ_NMETALS)));
    SHQ_TRY(ctx->bhw_queue.reserve((size_t) nq));
    SHQ_HIP(hipMemcpyAsync(ctx->metal_gf.ptr, gf.data(), sizeof(float) * gf.size(), hipMemcpyHostToDevice, st));
    SHQ_HIP(hipMemcpyAsync(ctx->metal_gd.ptr, gd.data(), sizeof(double) * gd.size(), hipMemcpyHostToDevice, st));
    double *ds = ctx->metal_star.ptr;
    SHQ_HIP(hipMemcpyAsync(ds, StarVolumeSPH, sizeof(double) * (size_t) nq, hipMemcpyHostToDevice, st));
    SHQ_HIP(hipMemcpyAsync(ds + nq, MassGenerated, sizeof(double) * (size_t) nq, hipMemcpyHostToDevice, st));
    SHQ_HIP(hipMemcpyAsync(ds + 2 * nq, MetalGenerated, sizeof(double) * (size_t) nq, hipMemcpyHostToDevice, st));
    SHQ_HIP(hipMemcpyAsync(ds + 4 * nq, MetalSpeciesGenerated, sizeof(double) * (size_t) nq * SHQ_NMETALS, hipMemcpyHostToDevice, st));
    SHQ_HIP(hipMemcpyAsync(ctx->bhw_queue.ptr, queue, sizeof(int32_t) * (size_t) nq, hipMemcpyHostToDevice, st));
    MetalWalkArgs w;
    memset(&w, 0, sizeof(w));
    w.SPHWeighting = SPHWeighting;
    w.MaxGasMass = MaxGasMass;
    w.starvolume = ds;
    w.massgenerated = ds + nq;
    w.metalgenerated = ds + 2 * nq;
    w.speciesgenerated = ds + 4 * nq;
    w.gmass = ctx->metal_gf.ptr;
    w.gmetals = ctx->metal_gf.ptr + cap;
    w.gdensity = ctx->metal_gd.ptr;
    w.gmetallicity = ctx->metal_gd.ptr + cap;
    /* the gas the return changes comes back as rows of the particles it touched */
    SHQ_TRY(ctx->bhw_touched.reserve(cap));
    SHQ_TRY(ctx->bhw_tlist.reserve(cap));
    SHQ_HIP(hipMemsetAsync(ctx->bhw_touched.ptr, 0, cap, st));
    w.touched = ctx->bhw_touched.ptr;
    SHQ_TRY(shq_metal_return_device(ctx, &w, DensityKernelType, tree->BoxSize, ctx->bhw_queue.ptr, nq, ds + 3 * nq, npairs));
    int64_t nt = 0;
    SHQ_TRY(shq_marked_list(ctx, ctx->bhw_touched.ptr, n, ctx->bhw_tlist.ptr, &nt));
    const size_t W = 3 + SHQ_NMETALS;
    SHQ_TRY(ctx->bhw_trows.reserve(W * (size_t) std::max<int64_t>(nt, 1)));
    std::vector<double> rows(W * (size_t) std::max<int64_t>(nt, 1));
    std::vector<int32_t> tl((size_t) std::max<int64_t>(nt, 1));
    SHQ_HIP(hipMemcpyAsync(MassReturn, ds + 3 * nq, sizeof(double) * (size_t) nq, hipMemcpyDeviceToHost, st));
    if(nt) {
        SHQ_TRY(shq_metal_rows_gather(ctx, &w, ctx->bhw_tlist.ptr, nt, ctx->bhw_trows.ptr));
        SHQ_HIP(hipMemcpyAsync(rows.data(), ctx->bhw_trows.ptr, sizeof(double) * W * (size_t) nt, hipMemcpyDeviceToHost, st));
        SHQ_HIP(hipMemcpyAsync(tl.data(), ctx->bhw_tlist.ptr, sizeof(int32_t) * (size_t) nt, hipMemcpyDeviceToHost, st));
    }
    SHQ_HIP(hipStreamSynchronize(st));
    for(int64_t t = 0; t < nt; t++) {
        const int64_t i = tl[(size_t) t];
        const double *r = &rows[W * (size_t) t];
        SHQ_CHECK(*pfield<uint8_t>(parts, i, parts->off_type) == 0 && !(*pfield<uint8_t>(parts, i, parts->off_flags) & 1u), SHQ_ERR_STATE,
                  "metal_return: the walk touched a particle that is not live gas");
        char *s = slot(i);
        *pfield_w<float>(parts, i, parts->off_mass) = (float) r[0];
        *reinterpret_cast<double *>(s + gas->off_density) = r[1];
        *reinterpret_cast<double *>(s + gas->off_metallicity) = r[2];
        for(int m = 0; m < SHQ_NMETALS; m++)
            reinterpret_cast<float *>(s + gas->off_metals)[m] = (float) r[3 + m];
    }
    /* the gas masses (and densities) changed in the caller's records only: the context's particle and SPH copies are no longer the views */
    ctx->inputs_current &= ~(SHQ_CURRENT_PARTICLES | SHQ_CURRENT_SPH);
    return SHQ_OK;
}

/* ---- winds_evolve / winds_subgrid (winds.cpp:272-292, 370-387, 567-585) ----------------------------------------------------------- */
namespace {
/* Vel, Entropy and DelayTime of the listed gas particles back into the caller's arrays */
int winds_writeback(shq_context *ctx, const shq_part_view *parts, const shq_sph_view *sph, const int32_t *list, int64_t nlist, bool vel_entropy)
{
    const int64_t n = parts->numpart;
    std::vector<double> hv(vel_entropy ? 3 * (size_t) n : 1), he(vel_entropy ? (size_t) n : 1), hdl((size_t) std::max<int64_t>(n, 1));
    hipStream_t st = ctx->stream;
    if(n > 0) {
        if(vel_entropy) {
            SHQ_HIP(hipMemcpyAsync(hv.data(), ctx->vel.ptr, sizeof(double) * 3 * (size_t) n, hipMemcpyDeviceToHost, st));
            SHQ_HIP(hipMemcpyAsync(he.data(), ctx->g_entropy.ptr, sizeof(double) * (size_t) n, hipMemcpyDeviceToHost, st));
        }
        SHQ_HIP(hipMemcpyAsync(hdl.data(), ctx->g_delaytime.ptr, sizeof(double) * (size_t) n, hipMemcpyDeviceToHost, st));
    }
    SHQ_HIP(hipStreamSynchronize(st));
    const int64_t cnt = list ? nlist : n;
    for(int64_t k = 0; k < cnt; k++) {
        const int64_t i = list ? list[k] : k;
        if(*pfield<uint8_t>(parts, i, parts->off_type) != 0 || (*pfield<uint8_t>(parts, i, parts->off_flags) & 1u))
            continue;
        const int32_t pi = *pfield<int32_t>(parts, i, parts->off_pi);
        *sfield(sph, pi, sph->off_delaytime) = hdl[(size_t) i];
        if(vel_entropy) {
            double *v = pfield_w<double>(parts, i, parts->off_vel);
            for(int j = 0; j < 3; j++)
                v[j] = hv[3 * (size_t) i + j];
            *sfield(sph, pi, sph->off_entropy) = he[(size_t) i];
        }
    }
    return SHQ_OK;
}

int winds_list_check(const shq_part_view *parts, const int32_t *list, int64_t nlist, bool must_be_gas, const char *who)
{
    const int64_t n = parts->numpart;
    for(int64_t k = 0; list && k < nlist; k++) {
        SHQ_CHECK(list[k] >= 0 && list[k] < n, SHQ_ERR_INVALID, "%s: list[%ld] = %d out of range", who, (long) k, list[k]);
        if(must_be_gas)
            SHQ_CHECK(*pfield<uint8_t>(parts, list[k], parts->off_type) == 0, SHQ_ERR_INVALID, "%s: particle %d is not gas", who, list[k]);
    }
    return SHQ_OK;
}
} // namespace

extern "C" int shq_winds_evolve(shq_context *ctx, const shq_part_view *parts, const shq_sph_view *sph, const int32_t *list, int64_t nlist, double a3inv, double hubble,
                                double WindFreeTravelDensThresh, double MaxWindFreeTravelTime, const shq_kick_factors *kf)
{
    SHQ_CHECK(ctx && parts && sph && kf && (nlist == 0 || list || nlist == parts->numpart), SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(sph->off_delaytime != SHQ_NOFIELD && parts->off_timebin_hydro != SHQ_NOFIELD && parts->off_flags != SHQ_NOFIELD, SHQ_ERR_INVALID,
              "winds_evolve: needs DelayTime, the hydro time bin and the flag byte");
    SHQ_TRY(winds_list_check(parts, list, nlist, false, "winds_evolve"));
    SHQ_HIP(hipSetDevice(ctx->device));
    SHQ_TRY(shq_particles_upload(ctx, parts));
    SHQ_TRY(sph_upload(ctx, parts, sph));
    const int64_t cnt = list ? nlist : parts->numpart;
    const int32_t *d_list = nullptr;
    if(list && cnt > 0) {
        SHQ_TRY(ctx->bhw_queue.reserve((size_t) cnt));
        SHQ_HIP(hipMemcpyAsync(ctx->bhw_queue.ptr, list, sizeof(int32_t) * (size_t) cnt, hipMemcpyHostToDevice, ctx->stream));
        d_list = ctx->bhw_queue.ptr;
    }
    SHQ_TRY(shq_winds_evolve_device(ctx, d_list, cnt, a3inv, hubble, WindFreeTravelDensThresh, MaxWindFreeTravelTime, kf));
    return winds_writeback(ctx, parts, sph, list, nlist, false);
}

extern "C" int shq_winds_subgrid(shq_context *ctx, const shq_part_view *parts, const shq_sph_view *sph, size_t sph_off_vdisp, const uint64_t *ids, const int32_t *list,
                                 int64_t nlist, const double *StellarMasses, const shq_wind_params *params, const double *rnd_table, int64_t rnd_size, int64_t *nkicked)
{
    SHQ_CHECK(ctx && parts && sph && ids && params && rnd_table && (nlist == 0 || (list && StellarMasses)), SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(rnd_size > 0, SHQ_ERR_INVALID, "winds_subgrid: empty random table");
    if(nkicked)
        *nkicked = 0;
    if(!(params->WindModel & 1) || nlist == 0) /* "The non-subgrid model does nothing here" */
        return SHQ_OK;
    SHQ_CHECK((params->WindModel & 8) || (params->WindModel & 4), SHQ_ERR_INVALID, "WindModel = 0x%X is strange (winds.cpp:503)", params->WindModel);
    SHQ_CHECK(sph->off_delaytime != SHQ_NOFIELD && sph_off_vdisp + 8 <= sph->elsize, SHQ_ERR_INVALID, "winds_subgrid: needs DelayTime and VDisp of the gas slots");
    SHQ_TRY(winds_list_check(parts, list, nlist, true, "winds_subgrid"));
    SHQ_HIP(hipSetDevice(ctx->device));
    const int64_t n = parts->numpart;
    SHQ_TRY(shq_particles_upload(ctx, parts));
    SHQ_TRY(sph_upload(ctx, parts, sph));
    hipStream_t st = ctx->stream;
    std::vector<double> hd(2 * (size_t) nlist);
    for(int64_t k = 0; k < nlist; k++) {
        hd[(size_t) k] = StellarMasses[k];
        hd[(size_t) (nlist + k)] = *sfield(sph, *pfield<int32_t>(parts, list[k], parts->off_pi), sph_off_vdisp);
    }
    SHQ_TRY(ctx->bhw_ids.reserve((size_t) std::max<int64_t>(n, 1)));
    SHQ_TRY(ctx->bhw_rnd.reserve((size_t) rnd_size));
    SHQ_TRY(ctx->bhw_queue.reserve((size_t) nlist));
    SHQ_TRY(ctx->wind_d.reserve(2 * (size_t) nlist));
    SHQ_TRY(ctx->wind_cnt.reserve(4));
    SHQ_TRY(ids_upload(ctx, ids, (int64_t) n));
    SHQ_HIP(hipMemcpyAsync(ctx->bhw_rnd.ptr, rnd_table, sizeof(double) * (size_t) rnd_size, hipMemcpyHostToDevice, st));
    SHQ_HIP(hipMemcpyAsync(ctx->bhw_queue.ptr, list, sizeof(int32_t) * (size_t) nlist, hipMemcpyHostToDevice, st));
    SHQ_HIP(hipMemcpyAsync(ctx->wind_d.ptr, hd.data(), sizeof(double) * hd.size(), hipMemcpyHostToDevice, st));
    SHQ_HIP(hipMemsetAsync(ctx->wind_cnt.ptr, 0, sizeof(unsigned long long) * 4, st));
    WindWalkArgs w;
    memset(&w, 0, sizeof(w));
    w.ids = ctx->bhw_ids.ptr;
    w.rnd = ctx->bhw_rnd.ptr;
    w.rndsize = (unsigned long long) rnd_size;
    w.P = *params;
    SHQ_TRY(shq_winds_subgrid_device(ctx, &w, ctx->bhw_queue.ptr, nlist, ctx->wind_d.ptr, ctx->wind_d.ptr + nlist, ctx->wind_cnt.ptr));
    unsigned long long h = 0;
    SHQ_HIP(hipMemcpyAsync(&h, ctx->wind_cnt.ptr, sizeof(h), hipMemcpyDeviceToHost, st));
    SHQ_TRY(winds_writeback(ctx, parts, sph, list, nlist, true));
    if(nkicked)
        *nkicked = (int64_t) h;
    return SHQ_OK;
}
