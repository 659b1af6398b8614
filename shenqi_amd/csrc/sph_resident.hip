/* sph_resident.hip — the SPH operators on a gas set that is ALREADY in HBM (multi-GPU runs: local + imported ghost gas as rows of a
 * device tensor that came out of the all-to-all, shenqi_amd/dist.py DistSPHDevice).
 *
 * shq_density / shq_hydro_force (sph_capi.hip) take the reference's host arrays, pack them on the host and move them over PCIe, once
 * per call: right for a drop-in behind density() / hydro_force() of one task (density2.cpp:105-151, hydra2.cpp:76-110), wrong for the
 * sharded step, where the same records would cross PCIe twice per operator and rank.  Here the caller hands over rows of
 * SHQ_GAS_NCOL doubles per gas particle — exactly the fields the two operators read of a neighbour and write of a target — the rows
 * are spread over the context's SoA arrays by one kernel, the tree comes from shq_tree_build, the operators run on the first
 * `nlocal` rows as targets (shq_sph_density_device / shq_sph_hydro_device, the same kernels as the one-shot calls) and one kernel
 * gathers the results back into rows.  Nothing but statistics crosses PCIe.
 */
#include "common.hpp"

namespace {

inline unsigned nblk(long long n, int t = 256) { return (unsigned) ((n + t - 1) / t); }

__global__ void gas_scatter_kernel(long long n, const double *__restrict__ rows, double4 *posm, uint8_t *pflags, double *vel, double *hsml,
                                   double *treeacc, double *gravpm, double *hacc, double *entropy, double *dtentropy, double *delay,
                                   double *density, double *egywt, double *dhsmlegy, double *divvel, double *curlvel, double *maxsig,
                                   double *dthsml, uint8_t *bin_grav, uint8_t *bin_hydro, int *bad)
{
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n)
        return;
    const double *r = rows + (size_t) i * SHQ_GAS_NCOL;
    posm[i] = make_double4(r[0], r[1], r[2], r[3]);
    pflags[i] = 0; /* gas, neither garbage nor swallowed */
    for(int k = 0; k < 3; k++) {
        vel[3 * i + k] = r[4 + k];
        treeacc[3 * i + k] = r[8 + k];
        gravpm[3 * i + k] = r[11 + k];
        hacc[3 * i + k] = r[14 + k];
    }
    hsml[i] = r[7];
    entropy[i] = r[17];
    dtentropy[i] = r[18];
    delay[i] = r[19];
    density[i] = r[20];
    egywt[i] = r[21];
    dhsmlegy[i] = r[22];
    divvel[i] = r[23];
    curlvel[i] = r[24];
    maxsig[i] = r[25];
    dthsml[i] = r[26];
    const long long b = (long long) r[27];
    const int bg = (int) (b & 255), bh = (int) (b >> 8);
    if(b < 0 || bg > SHQ_TIMEBINS || bh > SHQ_TIMEBINS)
        *bad = 1;
    bin_grav[i] = (uint8_t) bg;
    bin_hydro[i] = (uint8_t) bh;
}

__global__ void gas_gather_kernel(long long n, double *rows, const double *__restrict__ hsml, const double *__restrict__ dthsml,
                                  const double *__restrict__ density, const double *__restrict__ egywt, const double *__restrict__ dhsmlegy,
                                  const double *__restrict__ divvel, const double *__restrict__ curlvel, const double *__restrict__ hacc,
                                  const double *__restrict__ dtentropy, const double *__restrict__ maxsig, int which)
{
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n)
        return;
    double *r = rows + (size_t) i * SHQ_GAS_NCOL;
    if(which & 1) { /* density() results */
        r[7] = hsml[i];
        r[26] = dthsml[i];
        r[20] = density[i];
        r[21] = egywt[i];
        r[22] = dhsmlegy[i];
        r[23] = divvel[i];
        r[24] = curlvel[i];
    }
    if(which & 2) { /* hydro_force() results */
        r[14] = hacc[3 * i];
        r[15] = hacc[3 * i + 1];
        r[16] = hacc[3 * i + 2];
        r[18] = dtentropy[i];
        r[25] = maxsig[i];
    }
}

__global__ void iota_kernel(long long n, int32_t *q)
{
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n)
        q[i] = (int32_t) i;
}

int resident_queue(shq_context *ctx, const char *who)
{
    SHQ_CHECK(ctx->have_parts && ctx->have_sph && ctx->gas_resident, SHQ_ERR_STATE, "%s: no resident gas set (shq_gas_set_device)", who);
    SHQ_CHECK(ctx->have_tree && ctx->tb_built && ctx->have_father, SHQ_ERR_STATE, "%s: build the tree of the resident set first (shq_tree_build)", who);
    SHQ_CHECK(ctx->sphrun.phase == 0, SHQ_ERR_STATE, "%s: another SPH walk is open", who);
    const long long nq = ctx->nlocal;
    SHQ_TRY(ctx->s_queue0.reserve((size_t) (nq > 0 ? nq : 1)));
    if(nq > 0) {
        iota_kernel<<<dim3(nblk(nq)), dim3(256), 0, ctx->stream>>>(nq, ctx->s_queue0.ptr);
        SHQ_HIP(hipGetLastError());
    }
    return SHQ_OK;
}

} // namespace

extern "C" int shq_gas_set_device(shq_context *ctx, const double *d_rows, int64_t n, int64_t nlocal)
{
    if(ctx)
        ctx->inputs_current = 0; /* the resident set moves: shq_set_inputs_current ends here */
    SHQ_CHECK(ctx && (d_rows || n == 0), SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(n >= 0 && n < (1ll << 31) && nlocal >= 0 && nlocal <= n, SHQ_ERR_INVALID, "gas_set_device: bad particle counts");
    SHQ_HIP(hipSetDevice(ctx->device));
    SHQ_TRY(shq_join_pm(ctx));
    const size_t cap = (size_t) std::max<int64_t>(n, 1);
    SHQ_TRY(ctx->posm.reserve(cap));
    SHQ_TRY(ctx->oldacc.reserve(cap));
    SHQ_TRY(ctx->treeacc.reserve(3 * cap));
    SHQ_TRY(ctx->gravpm.reserve(3 * cap));
    SHQ_TRY(ctx->pmpot.reserve(cap));
    SHQ_TRY(ctx->acc.reserve(3 * cap));
    SHQ_TRY(ctx->pot.reserve(cap));
    SHQ_TRY(ctx->nint.reserve(cap));
    SHQ_TRY(ctx->pflags.reserve(cap));
    SHQ_TRY(ctx->hsml.reserve(cap));
    SHQ_TRY(ctx->dthsml.reserve(cap));
    SHQ_TRY(ctx->vel.reserve(3 * cap));
    SHQ_TRY(ctx->bin_grav.reserve(cap));
    SHQ_TRY(ctx->bin_hydro.reserve(cap));
    SHQ_TRY(ctx->g_entropy.reserve(cap));
    SHQ_TRY(ctx->g_dtentropy.reserve(cap));
    SHQ_TRY(ctx->g_hydroaccel.reserve(3 * cap));
    SHQ_TRY(ctx->g_delaytime.reserve(cap));
    SHQ_TRY(ctx->g_density.reserve(cap));
    SHQ_TRY(ctx->g_egywt.reserve(cap));
    SHQ_TRY(ctx->g_dhsmlegy.reserve(cap));
    SHQ_TRY(ctx->g_divvel.reserve(cap));
    SHQ_TRY(ctx->g_curlvel.reserve(cap));
    SHQ_TRY(ctx->g_maxsignalvel.reserve(cap));
    SHQ_TRY(ctx->gas_bad.reserve(1));
    int bad = 0;
    if(n > 0) {
        SHQ_HIP(hipMemsetAsync(ctx->gas_bad.ptr, 0, sizeof(int), ctx->stream));
        SHQ_HIP(hipMemsetAsync(ctx->oldacc.ptr, 0, sizeof(double) * n, ctx->stream));
        gas_scatter_kernel<<<dim3(nblk(n)), dim3(256), 0, ctx->stream>>>(
            n, d_rows, ctx->posm.ptr, ctx->pflags.ptr, ctx->vel.ptr, ctx->hsml.ptr, ctx->treeacc.ptr, ctx->gravpm.ptr, ctx->g_hydroaccel.ptr,
            ctx->g_entropy.ptr, ctx->g_dtentropy.ptr, ctx->g_delaytime.ptr, ctx->g_density.ptr, ctx->g_egywt.ptr, ctx->g_dhsmlegy.ptr,
            ctx->g_divvel.ptr, ctx->g_curlvel.ptr, ctx->g_maxsignalvel.ptr, ctx->dthsml.ptr, ctx->bin_grav.ptr, ctx->bin_hydro.ptr, ctx->gas_bad.ptr);
        SHQ_HIP(hipGetLastError());
        SHQ_HIP(hipMemcpyAsync(&bad, ctx->gas_bad.ptr, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        SHQ_HIP(hipStreamSynchronize(ctx->stream));
    }
    ctx->have_tree = false;
    ctx->tb_built = false;
    ctx->have_tree_targets = false;
    ctx->have_dyn = false;
    ctx->have_bh_dyn = false;
    ctx->nbh = 0;
    ctx->fof_ngroups = -1;
    ctx->have_toptree = false;
    ctx->n_act = ctx->n_sub = -1;
    ctx->numpart = n;
    ctx->nlocal = nlocal;
    ctx->have_parts = true;
    ctx->have_pm_result = false;
    ctx->have_sph = bad == 0;
    ctx->gas_resident = bad == 0;
    ctx->sphrun.phase = 0;
    SHQ_CHECK(bad == 0, SHQ_ERR_INVALID, "gas_set_device: a time bin column holds a value outside 0..%d", SHQ_TIMEBINS);
    return SHQ_OK;
}

extern "C" int shq_gas_get_device(shq_context *ctx, double *d_rows, int64_t n, int which)
{
    SHQ_CHECK(ctx && (d_rows || n == 0), SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->have_parts && ctx->have_sph && ctx->gas_resident && n >= 0 && n <= ctx->numpart, SHQ_ERR_STATE,
              "gas_get_device: %ld rows asked, %ld resident", (long) n, (long) (ctx->gas_resident ? ctx->numpart : -1));
    SHQ_CHECK(which >= 1 && which <= 3, SHQ_ERR_INVALID, "gas_get_device: which = 1 (density results), 2 (hydro results) or 3");
    SHQ_CHECK(!(which & 2) || ctx->g_hydroaccel_out.ptr, SHQ_ERR_STATE, "gas_get_device: no hydro run on this set");
    SHQ_HIP(hipSetDevice(ctx->device));
    if(n == 0)
        return SHQ_OK;
    gas_gather_kernel<<<dim3(nblk(n)), dim3(256), 0, ctx->stream>>>(n, d_rows, ctx->hsml.ptr, ctx->dthsml.ptr, ctx->g_density.ptr, ctx->g_egywt.ptr,
                                                                   ctx->g_dhsmlegy.ptr, ctx->g_divvel.ptr, ctx->g_curlvel.ptr,
                                                                   (which & 2) ? ctx->g_hydroaccel_out.ptr : ctx->g_hydroaccel.ptr,
                                                                   (which & 2) ? ctx->g_dtentropy_out.ptr : ctx->g_dtentropy.ptr,
                                                                   ctx->g_maxsignalvel.ptr, which);
    SHQ_HIP(hipGetLastError());
    return SHQ_OK;
}

extern "C" int shq_density_resident(shq_context *ctx, const shq_density_params *params, shq_sph_stats *stats)
{
    SHQ_CHECK(ctx && params, SHQ_ERR_INVALID, "null argument");
    SHQ_HIP(hipSetDevice(ctx->device));
    SHQ_TRY(resident_queue(ctx, "density_resident"));
    SHQ_TRY(shq_sph_prepare(ctx, &params->kf, nullptr, nullptr));
    return shq_sph_density_device(ctx, params, ctx->s_queue0.ptr, ctx->nlocal, 0, stats);
}

extern "C" int shq_hydro_resident(shq_context *ctx, const shq_hydro_params *params, shq_sph_stats *stats)
{
    SHQ_CHECK(ctx && params, SHQ_ERR_INVALID, "null argument");
    SHQ_HIP(hipSetDevice(ctx->device));
    SHQ_TRY(resident_queue(ctx, "hydro_resident"));
    SHQ_TRY(shq_sph_prepare(ctx, &params->kf, params, nullptr)); /* EntVarPred evaluated per particle (density2.h:115-128) */
    return shq_sph_hydro_device(ctx, params, ctx->s_queue0.ptr, ctx->nlocal, stats);
}
