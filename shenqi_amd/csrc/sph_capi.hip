/* sph_capi.hip — extern "C" entry points for SPH density / hydro (see include/shenqi_hip.h). */
#include "common.hpp"
#include <string.h>
#include <algorithm>

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

/* Gather the SPH state into per-particle-index arrays (gas fields come from slot PI). */
int sph_upload(shq_context *ctx, const shq_part_view *parts, const shq_sph_view *sph)
{
    SHQ_CHECK(parts->off_hsml != SHQ_NOFIELD && parts->off_vel != SHQ_NOFIELD && parts->off_pi != SHQ_NOFIELD &&
                  parts->off_type != SHQ_NOFIELD, SHQ_ERR_INVALID, "SPH needs Hsml, Vel, PI and Type in the particle view");
    SHQ_CHECK(sph && (sph->numslots == 0 || sph->base), SHQ_ERR_INVALID, "SPH slot view is NULL");
    const int64_t n = parts->numpart;
    std::vector<double> hsml(n), vel(3 * n), entropy(n, 0.0), dtentropy(n, 0.0), hacc(3 * n, 0.0), delay(n, 0.0);
    std::vector<double> density(n, 0.0), egywt(n, 0.0), dhsml(n, 0.0), divvel(n, 0.0), curl(n, 0.0);
    std::vector<uint8_t> bg(n, 0), bh(n, 0);
    int bad = 0;
    for(int64_t i = 0; i < n; i++) {
        hsml[i] = *pfield<double>(parts, i, parts->off_hsml);
        const double *v = pfield<double>(parts, i, parts->off_vel);
        vel[3 * i] = v[0]; vel[3 * i + 1] = v[1]; vel[3 * i + 2] = v[2];
        if(parts->off_timebin_gravity != SHQ_NOFIELD)
            bg[i] = *pfield<uint8_t>(parts, i, parts->off_timebin_gravity);
        if(parts->off_timebin_hydro != SHQ_NOFIELD)
            bh[i] = *pfield<uint8_t>(parts, i, parts->off_timebin_hydro);
        if(bg[i] > SHQ_TIMEBINS || bh[i] > SHQ_TIMEBINS)
            bad = 1;
        if(*pfield<uint8_t>(parts, i, parts->off_type) == 0) {
            const int32_t pi = *pfield<int32_t>(parts, i, parts->off_pi);
            if(pi < 0 || pi >= sph->numslots) {
                bad = 2;
                continue;
            }
            entropy[i] = *sfield(sph, pi, sph->off_entropy);
            dtentropy[i] = *sfield(sph, pi, sph->off_dtentropy);
            const double *ha = sfield(sph, pi, sph->off_hydroaccel);
            hacc[3 * i] = ha[0]; hacc[3 * i + 1] = ha[1]; hacc[3 * i + 2] = ha[2];
            if(sph->off_delaytime != SHQ_NOFIELD)
                delay[i] = *sfield(sph, pi, sph->off_delaytime);
            density[i] = *sfield(sph, pi, sph->off_density);
            egywt[i] = *sfield(sph, pi, sph->off_egywtdensity);
            dhsml[i] = *sfield(sph, pi, sph->off_dhsmlegydensityfactor);
            divvel[i] = *sfield(sph, pi, sph->off_divvel);
            curl[i] = *sfield(sph, pi, sph->off_curlvel);
        }
    }
    SHQ_CHECK(bad != 1, SHQ_ERR_INVALID, "time bin out of range (TIMEBINS = %d)", SHQ_TIMEBINS);
    SHQ_CHECK(bad != 2, SHQ_ERR_INVALID, "gas particle with PI outside the SPH slot array");
    SHQ_TRY(up(ctx, ctx->hsml, hsml));
    SHQ_TRY(up(ctx, ctx->vel, vel));
    SHQ_TRY(up(ctx, ctx->bin_grav, bg));
    SHQ_TRY(up(ctx, ctx->bin_hydro, bh));
    SHQ_TRY(up(ctx, ctx->g_entropy, entropy));
    SHQ_TRY(up(ctx, ctx->g_dtentropy, dtentropy));
    SHQ_TRY(up(ctx, ctx->g_hydroaccel, hacc));
    SHQ_TRY(up(ctx, ctx->g_delaytime, delay));
    SHQ_TRY(up(ctx, ctx->g_density, density));
    SHQ_TRY(up(ctx, ctx->g_egywt, egywt));
    SHQ_TRY(up(ctx, ctx->g_dhsmlegy, dhsml));
    SHQ_TRY(up(ctx, ctx->g_divvel, divvel));
    SHQ_TRY(up(ctx, ctx->g_curlvel, curl));
    SHQ_TRY(ctx->dthsml.reserve(std::max<int64_t>(n, 1)));
    if(n > 0)
        SHQ_HIP(hipMemsetAsync(ctx->dthsml.ptr, 0, sizeof(double) * n, ctx->stream));
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    ctx->have_sph = true;
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

extern "C" int shq_density(shq_context *ctx, const shq_tree_view *tree, shq_node *nodes_rw, const shq_part_view *parts,
                           const shq_sph_view *sph, const shq_bh_view *bh, const int32_t *active, int64_t nactive,
                           const shq_density_params *params, double *EntVarPred, double *GradRho_mag, shq_sph_stats *stats)
{
    SHQ_CHECK(ctx && tree && parts && sph && params, SHQ_ERR_INVALID, "null argument");
    SHQ_HIP(hipSetDevice(ctx->device));
    const int64_t n = parts->numpart;
    if(active)
        for(int64_t k = 0; k < nactive; k++)
            SHQ_CHECK(active[k] >= 0 && active[k] < n, SHQ_ERR_INVALID, "active[%ld] = %d out of range", (long) k, active[k]);
    SHQ_TRY(shq_particles_upload(ctx, parts));
    SHQ_TRY(sph_upload(ctx, parts, sph));
    SHQ_TRY(shq_tree_upload(ctx, tree));
    SHQ_CHECK(!params->update_hsml || ctx->have_father || nodes_rw == nullptr, SHQ_ERR_INVALID,
              "density with update_hsml needs tree->father to update hmax (update_tree_hmax_father)");
    std::vector<int32_t> queue = build_queue(parts, active, nactive, true);
    for(int32_t i : queue) {
        if(*pfield<uint8_t>(parts, i, parts->off_type) == 5)
            SHQ_CHECK(bh && bh->base, SHQ_ERR_INVALID, "black-hole density target but no BH slot view");
    }
    const int64_t nq = (int64_t) queue.size();
    SHQ_TRY(ctx->active.reserve((size_t) std::max<int64_t>(nq, 1)));
    if(nq > 0)
        SHQ_HIP(hipMemcpyAsync(ctx->active.ptr, queue.data(), sizeof(int32_t) * nq, hipMemcpyHostToDevice, ctx->stream));
    if(!ctx->have_father) { /* no hmax updates */
        SHQ_TRY(ctx->pfather.reserve((size_t) std::max<int64_t>(n, 1)));
        if(n > 0)
            SHQ_HIP(hipMemsetAsync(ctx->pfather.ptr, 0xff, sizeof(int32_t) * n, ctx->stream));
    }
    SHQ_TRY(shq_sph_prepare(ctx, &params->kf, nullptr, nullptr));
    SHQ_TRY(shq_sph_density_device(ctx, params, ctx->active.ptr, nq, GradRho_mag != nullptr, stats));

    /* results back into the caller's arrays: only the walked targets are assigned
     * (reduce<PRIMARY>, localtreewalk2.h:39) */
    std::vector<double> hsml, dthsml, density, egywt, dhsml, divvel, curl, gmag;
    std::vector<double4> velp;
    SHQ_TRY(down(ctx, ctx->hsml, hsml, n));
    SHQ_TRY(down(ctx, ctx->dthsml, dthsml, n));
    SHQ_TRY(down(ctx, ctx->g_density, density, n));
    SHQ_TRY(down(ctx, ctx->g_egywt, egywt, n));
    SHQ_TRY(down(ctx, ctx->g_dhsmlegy, dhsml, n));
    SHQ_TRY(down(ctx, ctx->g_divvel, divvel, n));
    SHQ_TRY(down(ctx, ctx->g_curlvel, curl, n));
    SHQ_TRY(down(ctx, ctx->velp, velp, n));
    if(GradRho_mag) {
        SHQ_TRY(ctx->s_evp_in.reserve((size_t) std::max<int64_t>(n, 1)));
        SHQ_TRY(shq_sph_gradrho_mag(ctx, ctx->s_evp_in.ptr));
        SHQ_TRY(down(ctx, ctx->s_evp_in, gmag, n));
    }
    std::vector<double> hmax;
    SHQ_TRY(down(ctx, ctx->node_hmax, hmax, ctx->numnodes));
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
            SHQ_CHECK(pi >= 0 && pi < bh->numslots, SHQ_ERR_INVALID, "BH particle with PI outside the BH slot array");
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
            shq_node &nd = nodes_rw[ctx->node_order[j]];
            if(hmax[j] > nd.hmax)
                nd.hmax = hmax[j];
        }
    }
    return SHQ_OK;
}

extern "C" int shq_hydro_force(shq_context *ctx, const shq_tree_view *tree, const shq_part_view *parts,
                               const shq_sph_view *sph, const int32_t *active, int64_t nactive,
                               const shq_hydro_params *params, const double *EntVarPred, shq_sph_stats *stats)
{
    SHQ_CHECK(ctx && tree && parts && sph && params, SHQ_ERR_INVALID, "null argument");
    SHQ_HIP(hipSetDevice(ctx->device));
    const int64_t n = parts->numpart;
    if(active)
        for(int64_t k = 0; k < nactive; k++)
            SHQ_CHECK(active[k] >= 0 && active[k] < n, SHQ_ERR_INVALID, "active[%ld] = %d out of range", (long) k, active[k]);
    SHQ_TRY(shq_particles_upload(ctx, parts));
    SHQ_TRY(sph_upload(ctx, parts, sph));
    SHQ_TRY(shq_tree_upload(ctx, tree));
    std::vector<int32_t> queue = build_queue(parts, active, nactive, false);
    const int64_t nq = (int64_t) queue.size();
    SHQ_TRY(ctx->active.reserve((size_t) std::max<int64_t>(nq, 1)));
    if(nq > 0)
        SHQ_HIP(hipMemcpyAsync(ctx->active.ptr, queue.data(), sizeof(int32_t) * nq, hipMemcpyHostToDevice, ctx->stream));
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
    SHQ_TRY(shq_sph_hydro_device(ctx, params, ctx->active.ptr, nq, stats));
    std::vector<double> hacc, dtent, maxsig;
    SHQ_TRY(down(ctx, ctx->g_hydroaccel_out, hacc, 3 * n));
    SHQ_TRY(down(ctx, ctx->g_dtentropy_out, dtent, n));
    SHQ_TRY(down(ctx, ctx->g_maxsignalvel, maxsig, n));
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
