/* capi.hip — the extern "C" boundary of libshenqi_hip (see include/shenqi_hip.h). */
#include "common.hpp"
#include <stdarg.h>
#include <string.h>
#include <stdlib.h>
#include <math.h>
#include <thread>
#include <atomic>
#include <algorithm>

static thread_local char g_err[1024] = "";

void shq_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *shq_last_error(void) { return g_err; }
extern "C" const char *shq_version(void) { return "shenqi_hip 0.1 (gfx950)"; }

namespace {

/* simple host parallel-for (the host side packs views into the device layout) */
template <typename F> void parallel_for(int64_t n, F f)
{
    unsigned nt = std::thread::hardware_concurrency();
    if(nt == 0)
        nt = 1;
    if(nt > 32)
        nt = 32;
    if(n < 65536 || nt == 1) {
        f((int64_t) 0, n);
        return;
    }
    std::vector<std::thread> th;
    const int64_t chunk = (n + nt - 1) / nt;
    for(unsigned t = 0; t < nt; t++) {
        const int64_t lo = (int64_t) t * chunk, hi = std::min(n, lo + chunk);
        if(lo >= hi)
            break;
        th.emplace_back([=]() { f(lo, hi); });
    }
    for(auto &x : th)
        x.join();
}

template <typename T> inline const T *field(const shq_part_view *v, int64_t i, size_t off)
{
    return reinterpret_cast<const T *>(static_cast<const char *>(v->base) + (size_t) i * v->elsize + off);
}
template <typename T> inline T *field_w(const shq_part_view *v, int64_t i, size_t off)
{
    return reinterpret_cast<T *>(static_cast<char *>(v->base) + (size_t) i * v->elsize + off);
}

/* the split streams of the SPH walks (centre + len, links) and of shq_tree_download, cut from the merged walk records */
__global__ void split_nodeG_kernel(const NodeG *__restrict__ g, long long n, NodeA *A, NodeB *B, NodeC *Cc)
{
    const long long j = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(j >= n)
        return;
    const NodeG x = g[j];
    NodeA a; NodeB b; NodeC c;
    a.cofm[0] = x.cofm[0]; a.cofm[1] = x.cofm[1]; a.cofm[2] = x.cofm[2]; a.mass = x.mass;
    b.center[0] = x.center[0]; b.center[1] = x.center[1]; b.center[2] = x.center[2]; b.len = x.len;
    c.sibling = x.sibling; c.child = x.child; c.type = x.type; c.count = x.count;
    A[j] = a; B[j] = b; Cc[j] = c;
}

/* rows (Acc[3], Potential) of an active list's targets, for the one-shot call's download */
__global__ void gather_rows_kernel(long long nt, const int32_t *__restrict__ targets, const double *__restrict__ acc, const double *__restrict__ pot,
                                   double *out)
{
    const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= nt)
        return;
    const long long i = targets[t];
    out[4 * t] = acc[3 * i];
    out[4 * t + 1] = acc[3 * i + 1];
    out[4 * t + 2] = acc[3 * i + 2];
    out[4 * t + 3] = pot[i];
}

__global__ void gather_leaf_kernel(const double4 *posm, const int32_t *pidx, double4 *out, long long n)
{
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n)
        out[i] = posm[pidx[i]];
}

} // namespace

extern "C" int shq_init(int device, void *stream, shq_context **out)
{
    SHQ_CHECK(out != nullptr, SHQ_ERR_INVALID, "shq_init: out is NULL");
    *out = nullptr;
    int ndev = 0;
    SHQ_HIP(hipGetDeviceCount(&ndev));
    SHQ_CHECK(ndev > 0, SHQ_ERR_DEVICE, "shq_init: no HIP device visible");
    SHQ_CHECK(device >= 0 && device < ndev, SHQ_ERR_INVALID, "shq_init: device %d out of range (%d devices)", device, ndev);
    SHQ_HIP(hipSetDevice(device));
    shq_context *ctx = new shq_context();
    ctx->device = device;
    /* SHQ_PM_CUS = k: with the PM on its own stream (SHQ_PM_OVERLAP), give it k compute units of every XCD and the
     * main stream the other 32 - k, so that the HBM-bound PM passes and the VALU-bound walk share the chip in space
     * (the walk's waves fill every register file, nothing of the PM fits beside them on the same CU).  The bit
     * pattern picks k CUs per XCD whether mask bits run XCD-major or interleave the XCDs. */
    uint32_t mask_pm[8] = {0, 0, 0, 0, 0, 0, 0, 0}, mask_main[8];
    int pm_cus = 0;
    if(const char *v = getenv("SHQ_PM_CUS"))
        pm_cus = atoi(v);
    if(pm_cus < 0 || pm_cus > 16)
        pm_cus = 0;
    for(int a = 0; a < 8; a++)
        for(int c = 0; c < pm_cus; c++) {
            const int bit = 32 * a + ((a + c / 4) % 8) + 8 * (c % 4);
            mask_pm[bit >> 5] |= 1u << (bit & 31);
        }
    for(int w = 0; w < 8; w++)
        mask_main[w] = ~mask_pm[w];
    if(stream) {
        ctx->stream = (hipStream_t) stream;
        ctx->own_stream = false;
        pm_cus = 0;
    } else {
        /* the library's own main stream runs at the highest priority: what it queues beside the work of the two side streams (the pair
         * kernel beside the walk, an early PM beside the tree build) gets the free slots first */
        int plo = 0, phi = 0;
        (void) hipDeviceGetStreamPriorityRange(&plo, &phi);
        hipError_t e = pm_cus > 0 ? hipExtStreamCreateWithCUMask(&ctx->stream, 8, mask_main)
                                  : hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, getenv("SHQ_MAIN_PRIO_DEFAULT") ? 0 : phi);
        if(e != hipSuccess) {
            shq_set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
            delete ctx;
            return SHQ_ERR_DEVICE;
        }
        ctx->own_stream = true;
    }
    for(int i = 0; i < SHQ_NTIMERS; i++) {
        (void) hipEventCreate(&ctx->ev_begin[i]);
        (void) hipEventCreate(&ctx->ev_end[i]);
    }
    int prio_lo = 0, prio_hi = 0;
    (void) hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    if(getenv("SHQ_DEBUG_PRIO"))
        fprintf(stderr, "[shq] stream priorities: lowest %d, highest %d\n", prio_lo, prio_hi);
    if((pm_cus > 0 ? hipExtStreamCreateWithCUMask(&ctx->stream_pm, 8, mask_pm)
                   : hipStreamCreateWithPriority(&ctx->stream_pm, hipStreamNonBlocking, prio_hi)) != hipSuccess ||
       hipEventCreateWithFlags(&ctx->ev_pm_ready, hipEventDisableTiming) != hipSuccess ||
       hipEventCreateWithFlags(&ctx->ev_pm_done, hipEventDisableTiming) != hipSuccess) {
        shq_set_error("creating the PM stream failed");
        delete ctx;
        return SHQ_ERR_DEVICE;
    }
    if(hipStreamCreateWithPriority(&ctx->stream_pair, hipStreamNonBlocking, prio_lo) != hipSuccess ||
       hipEventCreateWithFlags(&ctx->ev_pair_fork, hipEventDisableTiming) != hipSuccess ||
       hipEventCreateWithFlags(&ctx->ev_pair_join, hipEventDisableTiming) != hipSuccess) {
        shq_set_error("creating the pair kernel's stream failed");
        delete ctx;
        return SHQ_ERR_DEVICE;
    }
    if(const char *v = getenv("SHQ_WALK_FREE_CUS")) {
        const int k = atoi(v);
        if(k > 0 && k <= 16) {
            uint32_t mask_walk[8];
            for(int w = 0; w < 8; w++)
                mask_walk[w] = ~0u;
            for(int a = 0; a < 8; a++)
                for(int c = 0; c < k; c++) { /* the same k-CUs-of-every-XCD pattern as SHQ_PM_CUS */
                    const int bit = 32 * a + ((a + c / 4) % 8) + 8 * (c % 4);
                    mask_walk[bit >> 5] &= ~(1u << (bit & 31));
                }
            if(hipExtStreamCreateWithCUMask(&ctx->stream_walk, 8, mask_walk) != hipSuccess ||
               hipEventCreateWithFlags(&ctx->ev_walk_in, hipEventDisableTiming) != hipSuccess ||
               hipEventCreateWithFlags(&ctx->ev_walk_out, hipEventDisableTiming) != hipSuccess) {
                shq_set_error("creating the CU-masked walk stream failed");
                delete ctx;
                return SHQ_ERR_DEVICE;
            }
        }
    }
    if(const char *v = getenv("SHQ_PM_OVERLAP"))
        ctx->pm_overlap = atoi(v) != 0;
    if(const char *v = getenv("SHQ_PM_SCRUB"))
        ctx->pm_scrub = atoi(v) != 0;
    if(const char *v = getenv("SHQ_WALK_SPARSE"))
        ctx->walk_sparse = atoi(v);
    if(const char *v = getenv("SHQ_WALK_OVERLAP"))
        ctx->walk_overlap = atoi(v);
    if(const char *v = getenv("SHQ_TREE_TARGETS_REFRESH"))
        ctx->tree_targets_refresh = atoi(v) > 0 ? atoi(v) : 1;
    if(const char *v = getenv("SHQ_FFT_TRANSPOSED"))
        ctx->fft_transposed = atoi(v) != 0;
    if(const char *v = getenv("SHQ_TREEPM_FUSE"))
        ctx->treepm_fuse = atoi(v) != 0;
    if(const char *v = getenv("SHQ_WALK_VARIANT"))
        ctx->walk_variant = atoi(v);
    if(const char *v = getenv("SHQ_WALK_STATS_GUARD"))
        ctx->stats_guard = atoi(v);
    if(const char *v = getenv("SHQ_WALK_PADDING"))
        ctx->allow_padding = atoi(v) != 0;
    if(const char *v = getenv("SHQ_WALK_STATS"))
        ctx->walk_stats = atoi(v);
    if(const char *v = getenv("SHQ_XCD_K"))
        ctx->xcd_k = atoi(v);
    if(const char *v = getenv("SHQ_WALK_PERSIST"))
        ctx->walk_persist = atoi(v);
    if(const char *v = getenv("SHQ_WALK_RING"))
        ctx->walk_ring = atoi(v);
    {
        int ncu = 0;
        if(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && ncu > 0)
            ctx->num_cus = ncu;
    }
    *out = ctx;
    return SHQ_OK;
}

extern "C" void shq_shutdown(shq_context *ctx)
{
    if(!ctx)
        return;
    (void) hipSetDevice(ctx->device);
    if(ctx->stream_walk)
        (void) hipStreamSynchronize(ctx->stream_walk);
    if(ctx->stream_pm)
        (void) hipStreamSynchronize(ctx->stream_pm);
    if(ctx->stream_pair)
        (void) hipStreamSynchronize(ctx->stream_pair);
    (void) hipStreamSynchronize(ctx->stream);
    shq_pm_destroy_plans(ctx);
    ctx->posm.release(); ctx->oldacc.release(); ctx->treeacc.release(); ctx->gravpm.release();
    ctx->pmpot.release(); ctx->acc.release(); ctx->pot.release(); ctx->nint.release(); ctx->walk_tasks.release();
    ctx->sp_items.release(); ctx->sp_stack.release(); ctx->sp_count.release(); ctx->sp_flags.release(); ctx->sp_host.release(); ctx->node_lean_bad.release();
    ctx->pflags.release(); ctx->active.release(); ctx->act_list.release(); ctx->act_sub.release(); ctx->act_counts.release(); ctx->act_temp.release(); ctx->act_flag.release();
    ctx->gq_res.release(); ctx->s_queue0.release(); ctx->s_nlist2.release(); ctx->topnodes.release(); ctx->topleaves.release(); ctx->top_counts.release(); ctx->top_table.release(); ctx->gstats.release(); ctx->nodeF.release(); ctx->walk_counters.release(); ctx->walk_pool_idx.release(); ctx->walk_pool_msk.release(); ctx->walk_chunk_cnt.release(); ctx->walk_chunk_next.release(); ctx->walk_group_head.release();
    ctx->nodeA.release(); ctx->nodeB.release(); ctx->nodeC.release(); ctx->nodeG.release();
    ctx->posm_leaf.release(); ctx->leaf_pidx.release();
    ctx->mesh.release(); ctx->sinctab.release(); ctx->dbg_rho.release(); ctx->dbg_pot.release();
    ctx->mesh_words = 0; ctx->mesh_zeroed = false; ctx->mesh_alt.release();
    ctx->gravtab.release(); ctx->stage.release();
    ctx->node_hmax.release(); ctx->pfather.release();
    ctx->hsml.release(); ctx->dthsml.release(); ctx->vel.release(); ctx->bin_grav.release(); ctx->bin_hydro.release();
    ctx->g_entropy.release(); ctx->g_dtentropy.release(); ctx->g_hydroaccel.release(); ctx->g_delaytime.release();
    ctx->g_density.release(); ctx->g_egywt.release(); ctx->g_dhsmlegy.release(); ctx->g_divvel.release(); ctx->g_curlvel.release();
    ctx->g_hydroaccel_out.release(); ctx->g_dtentropy_out.release(); ctx->g_maxsignalvel.release();
    ctx->bh_pidx.release(); ctx->bh_u8.release(); ctx->bh_vec.release();
    ctx->ex_list.release(); ctx->ex_counts.release(); ctx->ex_i64.release(); ctx->ex_bytes.release(); ctx->ex_u64.release();
    for(auto &b : ctx->ex_val) b.release();
    for(auto &b : ctx->ex_key) b.release();
    ctx->fof_parent.release(); ctx->fof_partgrnr.release(); ctx->fof_members.release(); ctx->fof_groups.release(); ctx->fof_biglist.release(); ctx->fof_partial.release();
    for(auto &b : ctx->fof_i32) b.release();
    for(auto &b : ctx->fof_g32) b.release();
    for(auto &b : ctx->fof_u64) b.release();
    for(auto &b : ctx->fof_gkey) b.release();
    for(auto &b : ctx->fof_goff) b.release();
    ctx->hilb_iota.release();
    ctx->bhw_bhp.release(); ctx->bhw_queue.release(); ctx->bhw_rec.release(); ctx->bhw_ids.release(); ctx->bhw_sphsw.release(); ctx->bhw_bhsw.release();
    ctx->bhw_swid.release(); ctx->bhw_rnd.release(); ctx->bhw_out.release(); ctx->bhw_eeqos.release(); ctx->bhw_heated.release();
    ctx->wind_kicks.release(); ctx->wind_d.release(); ctx->wind_cnt.release();
    for(auto &b : ctx->metal_keys) b.release();
    for(auto &b : ctx->metal_val) b.release();
    ctx->metal_star.release(); ctx->metal_gd.release(); ctx->metal_gf.release();
    ctx->velp.release(); ctx->hydC.release(); ctx->hydD.release(); ctx->velp_leaf.release(); ctx->hydrec_leaf.release();
    ctx->hsml_leaf.release(); ctx->flag_leaf.release(); ctx->posf_leaf.release(); ctx->ngarb_leaf.release();
    ctx->s_numngb.release(); ctx->s_dhsmldens.release(); ctx->s_left.release(); ctx->s_right.release(); ctx->s_rot.release();
    ctx->s_gradrho.release(); ctx->s_evp_in.release(); ctx->s_todo.release(); ctx->s_queue2.release(); ctx->s_queue3.release();
    ctx->tb.release(); ctx->tree_targets.release(); ctx->ps_sums.release(); ctx->ps_bintab.release(); ctx->s_blockcount.release(); ctx->s_nlist.release(); ctx->s_ncount.release(); ctx->s_redo.release(); ctx->s_redo2.release(); ctx->s_counters.release(); ctx->pm_oob.release(); ctx->fft_tw.release();
    for(int i = 0; i < SHQ_NTIMERS; i++) {
        (void) hipEventDestroy(ctx->ev_begin[i]);
        (void) hipEventDestroy(ctx->ev_end[i]);
    }
    if(ctx->own_stream)
        (void) hipStreamDestroy(ctx->stream);
    if(ctx->stream_walk)
        (void) hipStreamDestroy(ctx->stream_walk);
    if(ctx->stream_pm)
        (void) hipStreamDestroy(ctx->stream_pm);
    if(ctx->stream_pair)
        (void) hipStreamDestroy(ctx->stream_pair);
    if(ctx->ev_pair_fork)
        (void) hipEventDestroy(ctx->ev_pair_fork);
    if(ctx->ev_pair_join)
        (void) hipEventDestroy(ctx->ev_pair_join);
    for(int i = 0; i < 4; i++)
        if(ctx->ev_sph[i])
            (void) hipEventDestroy(ctx->ev_sph[i]);
    if(ctx->ev_pm_ready)
        (void) hipEventDestroy(ctx->ev_pm_ready);
    if(ctx->ev_pm_done)
        (void) hipEventDestroy(ctx->ev_pm_done);
    delete ctx;
}

extern "C" int shq_set_walk_sparse(shq_context *ctx, int enable)
{
    SHQ_CHECK(ctx, SHQ_ERR_INVALID, "null context");
    ctx->walk_sparse = enable == 2 ? 2 : enable != 0;
    return SHQ_OK;
}

extern "C" int shq_set_walk_overlap(shq_context *ctx, int mode)
{
    SHQ_CHECK(ctx, SHQ_ERR_INVALID, "null context");
    SHQ_CHECK(mode >= 0 && mode <= 2, SHQ_ERR_INVALID, "walk overlap mode %d", mode);
    ctx->walk_overlap = mode;
    return SHQ_OK;
}

extern "C" int shq_walk_pair_lean(shq_context *ctx)
{
    SHQ_CHECK(ctx, SHQ_ERR_INVALID, "null context");
    if(!ctx->node_lean_checked || !ctx->node_lean_bad.ptr)
        return 0;
    SHQ_HIP(hipSetDevice(ctx->device));
    int bad = 1;
    SHQ_HIP(hipMemcpyAsync(&bad, ctx->node_lean_bad.ptr, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    return bad == 0 ? 1 : 0;
}

extern "C" int shq_set_walk_launch(shq_context *ctx, int persist, int leaf_ring)
{
    SHQ_CHECK(ctx && persist >= 0 && persist <= 2 && (leaf_ring == 0 || leaf_ring == 1), SHQ_ERR_INVALID, "walk launch: persist 0..2, leaf_ring 0 or 1");
    ctx->walk_persist = persist;
    ctx->walk_ring = leaf_ring;
    return SHQ_OK;
}

extern "C" int shq_set_walk_stats(shq_context *ctx, int level)
{
    SHQ_CHECK(ctx && level >= 0 && level <= 2, SHQ_ERR_INVALID, "walk stats level must be 0, 1 or 2");
    ctx->walk_stats = level;
    return SHQ_OK;
}

int shq_join_pm(shq_context *ctx)
{
    if(ctx->pm_pending) {
        SHQ_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_pm_done, 0));
        ctx->pm_pending = false;
    }
    return SHQ_OK;
}

extern "C" int shq_synchronize(shq_context *ctx)
{
    SHQ_CHECK(ctx, SHQ_ERR_INVALID, "null context");
    SHQ_TRY(shq_join_pm(ctx));
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    ctx->sp_check_pending = false;
    return shq_walk_check_status(ctx, false);
}

extern "C" void *shq_stream(shq_context *ctx) { return ctx ? (void *) ctx->stream : nullptr; }

extern "C" int shq_timer_begin(shq_context *ctx, int slot)
{
    SHQ_CHECK(ctx && slot >= 0 && slot < 8, SHQ_ERR_INVALID, "bad timer slot %d", slot);
    SHQ_HIP(hipEventRecord(ctx->ev_begin[slot], ctx->stream));
    return SHQ_OK;
}
extern "C" int shq_timer_end(shq_context *ctx, int slot)
{
    SHQ_CHECK(ctx && slot >= 0 && slot < 8, SHQ_ERR_INVALID, "bad timer slot %d", slot);
    SHQ_HIP(hipEventRecord(ctx->ev_end[slot], ctx->stream));
    return SHQ_OK;
}
/* elapsed time from one recorded timer event to another (which: 0 = the slot's begin event, 1 = its end event): the slots of the library's
 * own phases (8-13 PM, 16 tree build, 19 walk) against the caller's (0-7), whatever streams they were recorded on.  Waits for event b. */
extern "C" int shq_timer_between_ms(shq_context *ctx, int slot_a, int which_a, int slot_b, int which_b, double *ms)
{
    SHQ_CHECK(ctx && ms && slot_a >= 0 && slot_a < SHQ_NTIMERS && slot_b >= 0 && slot_b < SHQ_NTIMERS, SHQ_ERR_INVALID, "bad timer slot");
    hipEvent_t a = which_a ? ctx->ev_end[slot_a] : ctx->ev_begin[slot_a], b = which_b ? ctx->ev_end[slot_b] : ctx->ev_begin[slot_b];
    SHQ_HIP(hipEventSynchronize(a));
    SHQ_HIP(hipEventSynchronize(b));
    float f = 0;
    SHQ_HIP(hipEventElapsedTime(&f, a, b));
    *ms = f;
    return SHQ_OK;
}

extern "C" int shq_timer_elapsed_ms(shq_context *ctx, int slot, double *ms)
{
    SHQ_CHECK(ctx && ms && slot >= 0 && slot < SHQ_NTIMERS, SHQ_ERR_INVALID, "bad timer slot %d", slot);
    SHQ_HIP(hipEventSynchronize(ctx->ev_end[slot]));
    float f = 0;
    SHQ_HIP(hipEventElapsedTime(&f, ctx->ev_begin[slot], ctx->ev_end[slot]));
    *ms = f;
    return SHQ_OK;
}

/* ---- uploads --------------------------------------------------------------------------- */

extern "C" int shq_particles_upload(shq_context *ctx, const shq_part_view *parts)
{
    SHQ_CHECK(ctx && parts, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(parts->numpart >= 0 && parts->numpart < (1ll << 31), SHQ_ERR_INVALID, "numpart %ld out of range (int32 particle indices, treewalk2.h:599-615)", (long) parts->numpart);
    SHQ_CHECK(parts->numpart == 0 || parts->base, SHQ_ERR_INVALID, "particle base is NULL");
    SHQ_CHECK(parts->off_pos != SHQ_NOFIELD && parts->off_mass != SHQ_NOFIELD, SHQ_ERR_INVALID, "particle view needs Pos and Mass");
    SHQ_HIP(hipSetDevice(ctx->device));
    SHQ_TRY(shq_join_pm(ctx));
    /* shq_set_inputs_current: the caller vouches that the context holds this very view's particles as they are now */
    if((ctx->inputs_current & SHQ_CURRENT_PARTICLES) && ctx->have_parts && ctx->cur_parts == parts->base && ctx->cur_parts_n == parts->numpart &&
       ctx->numpart == parts->numpart)
        return SHQ_OK;
    const int64_t n = parts->numpart;
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

    /* AoS -> SoA into pinned staging, chunk by chunk, each chunk copied while the next is packed */
    const int64_t CH = 1 << 21;
    const size_t rec = sizeof(double4) + 6 * sizeof(double) + 1;
    SHQ_TRY(ctx->stage.reserve(2 * (size_t) CH * rec + 256));
    std::atomic<int> bad(0);
    double msum = 0;
    hipEvent_t ev[2] = {ctx->ev_begin[SHQ_NTIMERS - 2], ctx->ev_end[SHQ_NTIMERS - 2]};
    int nchunk = 0;
    for(int64_t c0 = 0; c0 < n; c0 += CH, nchunk++) {
        const int64_t m = std::min<int64_t>(CH, n - c0);
        const int sl = nchunk & 1;
        char *base = static_cast<char *>(ctx->stage.ptr) + (size_t) sl * CH * rec;
        double4 *h_posm = reinterpret_cast<double4 *>(base);
        double *h_tree = reinterpret_cast<double *>(base + (size_t) CH * sizeof(double4));
        double *h_pm = h_tree + 3 * CH;
        uint8_t *h_flags = reinterpret_cast<uint8_t *>(h_pm + 3 * CH);
        if(nchunk >= 2)
            SHQ_HIP(hipEventSynchronize(ev[sl])); /* the copies that read this half of the staging buffer are done */
        std::vector<double> part(64, 0.0);
        std::atomic<int> slot(0);
        parallel_for(m, [&](int64_t lo, int64_t hi) {
            double ms = 0;
            for(int64_t k = lo; k < hi; k++) {
                const int64_t i = c0 + k;
                const double *pos = field<double>(parts, i, parts->off_pos);
                const float mf = *field<float>(parts, i, parts->off_mass);
                h_posm[k] = make_double4(pos[0], pos[1], pos[2], (double) mf);
                ms += fabs((double) mf);
                if(!(isfinite(pos[0]) && isfinite(pos[1]) && isfinite(pos[2]) && isfinite(mf)))
                    bad.store(1);
                double t0 = 0, t1 = 0, t2 = 0, g0 = 0, g1 = 0, g2 = 0;
                if(parts->off_treeacc != SHQ_NOFIELD) {
                    const double *a = field<double>(parts, i, parts->off_treeacc);
                    t0 = a[0]; t1 = a[1]; t2 = a[2];
                }
                if(parts->off_gravpm != SHQ_NOFIELD) {
                    const double *a = field<double>(parts, i, parts->off_gravpm);
                    g0 = a[0]; g1 = a[1]; g2 = a[2];
                }
                h_tree[3 * k] = t0; h_tree[3 * k + 1] = t1; h_tree[3 * k + 2] = t2;
                h_pm[3 * k] = g0; h_pm[3 * k + 1] = g1; h_pm[3 * k + 2] = g2;
                uint8_t fl = 0;
                if(parts->off_flags != SHQ_NOFIELD)
                    fl |= (uint8_t) (*field<uint32_t>(parts, i, parts->off_flags) & 3u);
                if(parts->off_type != SHQ_NOFIELD)
                    fl |= (uint8_t) ((*field<uint8_t>(parts, i, parts->off_type) & 0xf) << 4);
                h_flags[k] = fl;
            }
            part[slot.fetch_add(1) & 63] += ms; /* at most 32 workers: one slot each */
        });
        for(double x : part)
            msum += x;
        SHQ_HIP(hipMemcpyAsync(ctx->posm.ptr + c0, h_posm, sizeof(double4) * m, hipMemcpyHostToDevice, ctx->stream));
        SHQ_HIP(hipMemcpyAsync(ctx->treeacc.ptr + 3 * c0, h_tree, sizeof(double) * 3 * m, hipMemcpyHostToDevice, ctx->stream));
        SHQ_HIP(hipMemcpyAsync(ctx->gravpm.ptr + 3 * c0, h_pm, sizeof(double) * 3 * m, hipMemcpyHostToDevice, ctx->stream));
        SHQ_HIP(hipMemcpyAsync(ctx->pflags.ptr + c0, h_flags, (size_t) m, hipMemcpyHostToDevice, ctx->stream));
        SHQ_HIP(hipEventRecord(ev[sl], ctx->stream));
    }
    if(n > 0) {
        SHQ_HIP(hipMemsetAsync(ctx->oldacc.ptr, 0, sizeof(double) * n, ctx->stream));
        SHQ_HIP(hipMemsetAsync(ctx->pmpot.ptr, 0, sizeof(double) * n, ctx->stream));
        SHQ_HIP(hipMemsetAsync(ctx->acc.ptr, 0, sizeof(double) * 3 * n, ctx->stream));
        SHQ_HIP(hipMemsetAsync(ctx->pot.ptr, 0, sizeof(double) * n, ctx->stream));
        SHQ_HIP(hipMemsetAsync(ctx->nint.ptr, 0, sizeof(int32_t) * n, ctx->stream));
    }
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    SHQ_CHECK(bad.load() == 0, SHQ_ERR_INVALID, "non-finite particle position or mass");
    ctx->mass_sum = msum;
    if(ctx->pm_log2scale_user >= 0)
        ctx->pm_log2scale = ctx->pm_log2scale_user;
    else {
        int ex = 0;
        (void) frexp(msum > 0 ? msum : 1.0, &ex); /* msum < 2^ex */
        ctx->pm_log2scale = 61 - ex;
    }
    ctx->numpart = n;
    ctx->nlocal = n; /* all particles are this rank's own */
    ctx->have_parts = true;
    ctx->have_tree = false; /* leaf copy refers to the old particles */
    ctx->have_tree_targets = false;
    ctx->tb_built = false;
    ctx->have_sph = false;  /* Hsml / Vel / slot data of the previous particle set */
    ctx->gas_resident = false;
    ctx->have_dyn = false;
    ctx->have_bh_dyn = false;
    ctx->nbh = 0;
    ctx->fof_ngroups = -1;
    ctx->have_toptree = false;
    ctx->n_act = ctx->n_sub = -1;
    ctx->have_pm_result = false;
    ctx->cur_parts = parts->base;
    ctx->cur_parts_n = n;
    ctx->cur_sph = ctx->cur_tree = ctx->cur_ids = nullptr; /* they described the previous set */
    return SHQ_OK;
}

/* One-shot operators upload the views they are given: particles, SPH state, tree (and the ID array of the sub-grid walks) — at 2 x 10^6
 * particles 50-170 ms of packing and PCIe around 4-7 ms of kernels (run.cpp:621-681 calls a handful of them per step on the same
 * particles).  With a bit set here the caller vouches that the context's copy of that input IS the view it passes next (same base
 * pointer and count, contents unchanged on the host since the copy was made, or changed only by the library's own operators, which
 * write their results to the views and to the context alike): the upload is skipped.  Anything that moves the resident set (a drift, a
 * kick, an exchange, a device-side particle hand-over) clears the mask; so does mask = 0. */
extern "C" int shq_set_inputs_current(shq_context *ctx, int mask)
{
    SHQ_CHECK(ctx && mask >= 0 && mask <= 15, SHQ_ERR_INVALID, "set_inputs_current: mask is a combination of SHQ_CURRENT_*");
    ctx->inputs_current = mask;
    return SHQ_OK;
}

extern "C" int shq_tree_upload(shq_context *ctx, const shq_tree_view *tree)
{
    SHQ_CHECK(ctx && tree, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->have_parts, SHQ_ERR_STATE, "tree_upload: upload particles first");
    SHQ_CHECK(tree->nodes_base && tree->numnodes > 0, SHQ_ERR_INVALID, "tree has no nodes");
    SHQ_CHECK(tree->numnodes < (1ll << 31), SHQ_ERR_INVALID, "too many nodes");
    SHQ_CHECK(tree->rootnode >= tree->firstnode && tree->rootnode < tree->firstnode + tree->numnodes, SHQ_ERR_INVALID, "root node %d outside [%ld, %ld)", tree->rootnode, (long) tree->firstnode, (long) (tree->firstnode + tree->numnodes));
    SHQ_CHECK(tree->BoxSize > 0, SHQ_ERR_INVALID, "tree BoxSize must be > 0");
    SHQ_HIP(hipSetDevice(ctx->device));
    if((ctx->inputs_current & SHQ_CURRENT_TREE) && ctx->have_tree && !ctx->tb_built && ctx->cur_tree == (const void *) tree->nodes_base &&
       ctx->cur_tree_n == tree->numnodes && ctx->cur_tree_first == tree->firstnode && ctx->treeBox == tree->BoxSize)
        return SHQ_OK;
    const int64_t nall = tree->numnodes, fn = tree->firstnode;
    const shq_node *src = tree->nodes_base;
    /* Depth-first pre-order of the reachable nodes = the threaded walk with every node opened
     * (open -> suns[0], otherwise -> sibling; forcetree.cpp:1016-1103 sets both).  The device
     * pool is stored in this order so that the record after the current one is almost always
     * the next one visited. */
    /* The threaded walk is one long chain of dependent loads (7e6 nodes x ~60 ns).  It is cut into independent pieces: the
     * nodes of the top levels are listed sequentially, every subtree hanging below them is traversed by its own host thread
     * (from its root until the walk reaches the root's sibling), and the pieces are stitched together in pre-order. */
    std::vector<int32_t> order;
    std::vector<int32_t> newidx((size_t) nall, -1);
    std::vector<int64_t> pstart; /* leaf slot offsets: exclusive prefix sum of noccupied over leaves in pre-order */
    {
        auto valid = [&](int64_t no) { return no >= fn && no < fn + nall; };
        auto internal = [&](int64_t no) {
            const shq_node &s = src[no - fn];
            return SHQ_NODE_CHILDTYPE(s.flags) == SHQ_NODE_NODE_TYPE && valid(s.suns[0]);
        };
        struct Seg { int64_t node; bool expand; int64_t size, leafparts; };
        std::vector<Seg> segs;
        /* pieces of about 1/512 of the tree each, judged by node mass: a fixed depth gives a handful of huge pieces and
         * a million tiny ones in a clustered tree */
        const double heavy = fabs(src[tree->rootnode - fn].mass) / 512.0;
        const int DEPTH = 40;
        {
            /* explicit stack of (node, depth); children of P: the sibling chain from suns[0] up to P.sibling */
            std::vector<std::pair<int64_t, int>> stack;
            stack.push_back({tree->rootnode, 0});
            int64_t guard = 0;
            while(!stack.empty()) {
                const auto [no, depth] = stack.back();
                stack.pop_back();
                SHQ_CHECK(++guard <= nall, SHQ_ERR_INVALID, "tree threading does not terminate (cycle near the root)");
                if(depth >= DEPTH || !internal(no) || !(fabs(src[no - fn].mass) > heavy)) {
                    segs.push_back({no, internal(no), 1, 0});
                    continue;
                }
                segs.push_back({no, false, 1, 0});
                const shq_node &P = src[no - fn];
                int64_t kids[9];
                int nk = 0;
                for(int64_t c = P.suns[0]; valid(c) && c != P.sibling && nk <= 8; c = src[c - fn].sibling)
                    kids[nk++] = c;
                SHQ_CHECK(nk <= 8, SHQ_ERR_INVALID, "node %ld has more than 8 children in its sibling chain", (long) no);
                for(int q = nk - 1; q >= 0; q--)
                    stack.push_back({kids[q], depth + 1});
            }
        }
        const int64_t nseg = (int64_t) segs.size();
        std::atomic<int> bad_tree(0);
        auto leaf_count = [&](const shq_node &s, int &cnt) {
            cnt = 0;
            if(SHQ_NODE_CHILDTYPE(s.flags) == SHQ_PARTICLE_NODE_TYPE) {
                cnt = s.noccupied;
                if(cnt < 0 || cnt > SHQ_NMAXCHILD)
                    bad_tree.store(2);
            }
        };
        auto run_threads = [&](auto &&fn_) {
            unsigned nt = std::thread::hardware_concurrency();
            nt = nt == 0 ? 1 : (nt > 32 ? 32 : nt);
            std::vector<std::thread> th;
            for(unsigned t = 1; t < nt; t++)
                th.emplace_back(fn_);
            fn_();
            for(auto &x : th)
                x.join();
        };
        /* pass 1: the size of every piece (nothing is stored: no allocation per piece) */
        std::atomic<int64_t> next_seg(0);
        run_threads([&]() {
            for(;;) {
                const int64_t k = next_seg.fetch_add(1);
                if(k >= nseg)
                    break;
                Seg &sg = segs[k];
                int cnt;
                if(!sg.expand) {
                    leaf_count(src[sg.node - fn], cnt);
                    sg.size = 1;
                    sg.leafparts = cnt;
                    continue;
                }
                const int64_t end = src[sg.node - fn].sibling;
                int64_t no = sg.node, lp = 0, sz = 0;
                while(valid(no) && no != end) {
                    if(++sz > nall) { /* cannot happen in a tree: a cycle */
                        bad_tree.store(1);
                        break;
                    }
                    const shq_node &sn = src[no - fn];
                    leaf_count(sn, cnt);
                    lp += cnt;
                    no = internal(no) ? sn.suns[0] : sn.sibling;
                }
                sg.size = sz;
                sg.leafparts = lp;
            }
        });
        SHQ_CHECK(bad_tree.load() != 1, SHQ_ERR_INVALID, "tree threading revisits a node (cycle)");
        SHQ_CHECK(bad_tree.load() != 2, SHQ_ERR_INVALID, "a leaf node has noccupied outside [0, %d]", SHQ_NMAXCHILD);
        std::vector<int64_t> off((size_t) nseg + 1, 0), loff((size_t) nseg + 1, 0);
        for(int64_t k = 0; k < nseg; k++) {
            off[k + 1] = off[k] + segs[k].size;
            loff[k + 1] = loff[k] + segs[k].leafparts;
        }
        const int64_t total = off[nseg];
        SHQ_CHECK(total <= nall, SHQ_ERR_INVALID, "tree threading visits %ld nodes, the tree has %ld (a node is reached twice)", (long) total, (long) nall);
        order.resize((size_t) total);
        pstart.resize((size_t) total + 1);
        /* pass 2: the same traversals again, writing at the pieces' offsets */
        std::atomic<int> twice(0);
        next_seg.store(0);
        run_threads([&]() {
            for(;;) {
                const int64_t k = next_seg.fetch_add(1);
                if(k >= nseg)
                    break;
                const Seg &sg = segs[k];
                const int64_t end = sg.expand ? (int64_t) src[sg.node - fn].sibling : -3;
                int64_t no = sg.node, lp = loff[k];
                for(int64_t j = off[k]; j < off[k + 1]; j++) {
                    const shq_node &sn = src[no - fn];
                    int cnt;
                    leaf_count(sn, cnt);
                    order[j] = (int32_t) (no - fn);
                    if(newidx[no - fn] != -1)
                        twice.store(1);
                    newidx[no - fn] = (int32_t) j;
                    pstart[j] = lp;
                    lp += cnt;
                    no = internal(no) ? sn.suns[0] : sn.sibling;
                    (void) end;
                }
            }
        });
        pstart[total] = loff[nseg];
        SHQ_CHECK(twice.load() == 0, SHQ_ERR_INVALID, "tree threading revisits a node (cycle)");
    }
    const int64_t nn = (int64_t) order.size();
    const int64_t nleafparts = pstart[nn];
    SHQ_CHECK(nleafparts < (1ll << 31) - 16, SHQ_ERR_INVALID, "too many leaf particles");
    const int64_t npad = nleafparts + SHQ_NMAXCHILD; /* padded: the walk fetches several slots at a time */
    const int64_t np = ctx->numpart;
    SHQ_TRY(ctx->nodeA.reserve(nn + 1));
    SHQ_TRY(ctx->nodeB.reserve(nn + 1));
    SHQ_TRY(ctx->nodeC.reserve(nn + 1));
    SHQ_TRY(ctx->nodeG.reserve(nn + 1));
    SHQ_TRY(ctx->node_hmax.reserve(nn + 1));
    SHQ_TRY(ctx->leaf_pidx.reserve((size_t) npad));
    SHQ_TRY(ctx->posm_leaf.reserve((size_t) npad));
    /* Pack the merged 128-byte walk records (plus hmax and the leaves' particle indices) straight into pinned staging,
     * a chunk of nodes at a time, each chunk copied while the next is packed; the split A / B / C streams of the SPH
     * walks are cut from the merged records on the device. */
    const int64_t CH = 1 << 20;
    const size_t chunk_bytes = (size_t) CH * (sizeof(NodeG) + sizeof(double) + SHQ_NMAXCHILD * sizeof(int32_t));
    SHQ_TRY(ctx->stage.reserve(2 * chunk_bytes + 256));
    std::atomic<int> bad(0);
    hipEvent_t ev[2] = {ctx->ev_begin[SHQ_NTIMERS - 2], ctx->ev_end[SHQ_NTIMERS - 2]};
    int nchunk = 0;
    for(int64_t c0 = 0; c0 < nn + 1; c0 += CH, nchunk++) {
        const int64_t m = std::min<int64_t>(CH, nn + 1 - c0);
        const int sl = nchunk & 1;
        char *base = ctx->stage.ptr + (size_t) sl * chunk_bytes;
        NodeG *hG = reinterpret_cast<NodeG *>(base);
        double *hH = reinterpret_cast<double *>(base + (size_t) CH * sizeof(NodeG));
        int32_t *hP = reinterpret_cast<int32_t *>(hH + CH);
        const int64_t p0 = pstart[c0], p1 = pstart[std::min<int64_t>(c0 + m, nn)];
        if(nchunk >= 2)
            SHQ_HIP(hipEventSynchronize(ev[sl]));
        parallel_for(m, [&](int64_t lo, int64_t hi) {
            for(int64_t k = lo; k < hi; k++) {
                const int64_t j = c0 + k;
                NodeG g;
                memset(&g, 0, sizeof(g));
                if(j == nn) { /* pad record: speculative fetch of pool[cur + 1] at the last node */
                    g.sibling = -1; g.child = -1; g.type = SHQ_PSEUDO_NODE_TYPE; g.count = 0;
                    g.wraplim = 0.5 * tree->BoxSize;
                    hG[k] = g;
                    hH[k] = 0;
                    continue;
                }
                const shq_node &sn = src[order[j]];
                for(int d = 0; d < 3; d++) {
                    g.cofm[d] = sn.cofm[d];
                    g.center[d] = sn.center[d];
                }
                g.mass = sn.mass;
                g.len = sn.len;
                const int64_t sib = sn.sibling;
                g.sibling = (sib >= fn && sib < fn + nall) ? newidx[sib - fn] : -1;
                g.type = (int32_t) SHQ_NODE_CHILDTYPE(sn.flags);
                g.count = 0;
                g.child = -1;
                if(g.type == SHQ_PARTICLE_NODE_TYPE) {
                    const int cnt = (int) (pstart[j + 1] - pstart[j]);
                    g.count = cnt;
                    g.child = (int32_t) pstart[j];
                    for(int c = 0; c < cnt; c++) {
                        const int32_t pp = sn.suns[c];
                        if(pp < 0 || pp >= np) {
                            bad.store(1);
                            break;
                        }
                        hP[pstart[j] - p0 + c] = pp;
                    }
                } else if(g.type == SHQ_NODE_NODE_TYPE) {
                    const int64_t ch = sn.suns[0];
                    g.child = (ch >= fn && ch < fn + nall) ? newidx[ch - fn] : -1;
                    if(g.child < 0)
                        g.type = SHQ_PSEUDO_NODE_TYPE; /* never descend into an invalid link */
                }
                g.bhlim = 0; /* filled per walk, like rcuthl */
                g.mlen2 = g.mass * g.len * g.len; /* (mass * len) * len, as shall_we_open_node evaluates it */
                g.inside = 0.6 * g.len;
                g.rcut2 = 0;
                g.wraplim = std::max(0.5 * tree->BoxSize - 0.5 * g.len, 0.0);
                hG[k] = g;
                hH[k] = sn.hmax;
            }
        });
        SHQ_HIP(hipMemcpyAsync(ctx->nodeG.ptr + c0, hG, sizeof(NodeG) * m, hipMemcpyHostToDevice, ctx->stream));
        SHQ_HIP(hipMemcpyAsync(ctx->node_hmax.ptr + c0, hH, sizeof(double) * m, hipMemcpyHostToDevice, ctx->stream));
        if(p1 > p0)
            SHQ_HIP(hipMemcpyAsync(ctx->leaf_pidx.ptr + p0, hP, sizeof(int32_t) * (p1 - p0), hipMemcpyHostToDevice, ctx->stream));
        SHQ_HIP(hipEventRecord(ev[sl], ctx->stream));
    }
    SHQ_HIP(hipMemsetAsync(ctx->leaf_pidx.ptr + nleafparts, 0, sizeof(int32_t) * SHQ_NMAXCHILD, ctx->stream));
    split_nodeG_kernel<<<dim3((unsigned) ((nn + 1 + 255) / 256)), dim3(256), 0, ctx->stream>>>(ctx->nodeG.ptr, nn + 1, ctx->nodeA.ptr, ctx->nodeB.ptr,
                                                                                           ctx->nodeC.ptr);
    if(np > 0)
        gather_leaf_kernel<<<dim3((unsigned) ((npad + 255) / 256)), dim3(256), 0, ctx->stream>>>(ctx->posm.ptr, ctx->leaf_pidx.ptr, ctx->posm_leaf.ptr,
                                                                                           npad);
    else
        SHQ_HIP(hipMemsetAsync(ctx->posm_leaf.ptr, 0, sizeof(double4) * npad, ctx->stream));
    SHQ_HIP(hipGetLastError());
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    SHQ_CHECK(bad.load() == 0, SHQ_ERR_INVALID, "tree leaf refers to a particle index outside [0, numpart)");
    /* SPH extra: the leaf holding each particle (ForceTree.Father) */
    ctx->have_father = false;
    if(tree->father && np > 0) {
        SHQ_TRY(ctx->pfather.reserve((size_t) np));
        SHQ_TRY(ctx->stage.reserve(sizeof(int32_t) * (size_t) np));
        int32_t *pf = reinterpret_cast<int32_t *>(ctx->stage.ptr);
        parallel_for(np, [&](int64_t lo, int64_t hi) {
            for(int64_t i = lo; i < hi; i++) {
                const int64_t f = tree->father[i];
                pf[i] = (f >= fn && f < fn + nall) ? newidx[f - fn] : -1;
            }
        });
        SHQ_HIP(hipMemcpyAsync(ctx->pfather.ptr, pf, sizeof(int32_t) * np, hipMemcpyHostToDevice, ctx->stream));
        SHQ_HIP(hipStreamSynchronize(ctx->stream));
        ctx->have_father = true;
    }
    ctx->node_order = std::move(order);
    ctx->node_rank = std::move(newidx);
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    ctx->numnodes = nn;
    ctx->firstnode = fn;
    ctx->root = 0; /* pre-order: the root is record 0 */
    ctx->ntreeparts = nleafparts;
    ctx->treeBox = tree->BoxSize;
    ctx->have_tree = true;
    ctx->node_rcut = -1;
    ctx->have_group_aux = false;
    ctx->tb_built = false;
    SHQ_TRY(shq_walk_prereserve(ctx));
    ctx->have_tree_targets = false;
    ctx->cur_tree = tree->nodes_base;
    ctx->cur_tree_n = tree->numnodes;
    ctx->cur_tree_first = tree->firstnode;
    return SHQ_OK;
}

__global__ void fill_u8_kernel(uint8_t *x, long long n, uint8_t v)
{
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n)
        x[i] = v;
}

/* Device-side particle set for multi-GPU runs: rows (x, y, z, m) already in HBM (e.g. a torch
 * tensor holding local + imported ghost particles); the first nlocal rows are this rank's own. */
extern "C" int shq_particles_set_device(shq_context *ctx, const void *d_posm, int64_t n, int64_t nlocal, int keep_tree)
{
    if(ctx)
        ctx->inputs_current = 0; /* the resident set moves: shq_set_inputs_current ends here */
    SHQ_CHECK(ctx && (d_posm || n == 0), SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(n >= 0 && n < (1ll << 31) && nlocal >= 0 && nlocal <= n, SHQ_ERR_INVALID, "bad particle counts");
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
    if(n > 0) {
        SHQ_HIP(hipMemcpyAsync(ctx->posm.ptr, d_posm, sizeof(double4) * n, hipMemcpyDeviceToDevice, ctx->stream));
        if(ctx->numpart != n) { /* a new particle set: previous-step accelerations no longer apply */
            SHQ_HIP(hipMemsetAsync(ctx->treeacc.ptr, 0, sizeof(double) * 3 * n, ctx->stream));
            SHQ_HIP(hipMemsetAsync(ctx->gravpm.ptr, 0, sizeof(double) * 3 * n, ctx->stream));
            SHQ_HIP(hipMemsetAsync(ctx->oldacc.ptr, 0, sizeof(double) * n, ctx->stream));
            SHQ_HIP(hipMemsetAsync(ctx->pmpot.ptr, 0, sizeof(double) * n, ctx->stream));
            SHQ_HIP(hipMemsetAsync(ctx->acc.ptr, 0, sizeof(double) * 3 * n, ctx->stream));
            SHQ_HIP(hipMemsetAsync(ctx->pot.ptr, 0, sizeof(double) * n, ctx->stream));
            SHQ_HIP(hipMemsetAsync(ctx->nint.ptr, 0, sizeof(int32_t) * n, ctx->stream));
        }
        fill_u8_kernel<<<dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, ctx->stream>>>(ctx->pflags.ptr, n, (uint8_t) (1 << 4));
        SHQ_HIP(hipGetLastError());
    }
    /* new positions: the tree and its leaf copies refer to the old ones even when the count is the same (a walk without a
     * rebuild is refused, as after shq_drift), unless the caller vouches that nothing moved */
    if(!(keep_tree && ctx->numpart == n && ctx->have_tree)) {
        ctx->have_tree = false;
        ctx->tb_built = false;
    }
    ctx->have_tree_targets = false; /* nlocal may have changed */
    ctx->have_sph = false;
    ctx->gas_resident = false;
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
    ctx->pm_prestarted = false;
    return SHQ_OK;
}

/* ---- gravity ---------------------------------------------------------------------------- */

extern "C" int shq_grav_short_run(shq_context *ctx, const shq_grav_params *params, const int32_t *active,
                                  int64_t nactive, int update_potential, int walk_mode)
{
    SHQ_CHECK(ctx && params, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->have_parts && ctx->have_tree, SHQ_ERR_STATE, "grav_short_run: upload particles and tree first");
    SHQ_HIP(hipSetDevice(ctx->device));
    SHQ_TRY(shq_walk_check_status(ctx, false));
    const int32_t *d_active = nullptr;
    int64_t nt = 0;
    if((walk_mode & SHQ_WALK_TREE_ORDER) && !active) {
        /* all own particles of the tree, taken in leaf order */
        SHQ_TRY(shq_build_tree_targets(ctx));
        d_active = ctx->tree_targets.ptr;
        nt = ctx->ntree_targets;
    } else
        SHQ_TRY(shq_resolve_active(ctx, active, nactive, ctx->nlocal, &d_active, &nt));
    const bool defer = (walk_mode & SHQ_WALK_DEFER_POSTPROCESS) != 0;
    walk_mode &= 0xff;
    SHQ_TRY(shq_launch_grav_walk(ctx, params, d_active, nt, update_potential, walk_mode));
    if(!defer)
        SHQ_TRY(shq_launch_grav_postprocess(ctx, params, d_active, nt, update_potential));
    ctx->grav_raw = defer;
    ctx->last_stats.ntargets = nt;
    return SHQ_OK;
}

/* The particles [first, first + count) as targets, no list: lets a caller cut one walk into pieces it interleaves with other
 * work on the stream (shenqi_amd/dist.py starts a mesh transpose between the pieces). */
extern "C" int shq_grav_short_run_range(shq_context *ctx, const shq_grav_params *params, int64_t first, int64_t count,
                                        int update_potential, int walk_mode)
{
    SHQ_CHECK(ctx && params, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->have_parts && ctx->have_tree, SHQ_ERR_STATE, "grav_short_run_range: upload particles and tree first");
    const int64_t nown = ctx->nlocal;
    SHQ_CHECK(first >= 0 && count >= 0 && first + count <= nown, SHQ_ERR_INVALID, "grav_short_run_range: [%ld, +%ld) outside the %ld own particles",
              (long) first, (long) count, (long) nown);
    SHQ_CHECK((walk_mode & ~0xff) == 0, SHQ_ERR_INVALID, "grav_short_run_range: walk_mode flags are not supported");
    SHQ_HIP(hipSetDevice(ctx->device));
    hipStream_t main = ctx->stream;
    if(ctx->stream_walk) { /* the piece runs on the CU-masked stream, in the main stream's order on both sides */
        SHQ_HIP(hipEventRecord(ctx->ev_walk_in, main));
        SHQ_HIP(hipStreamWaitEvent(ctx->stream_walk, ctx->ev_walk_in, 0));
        ctx->stream = ctx->stream_walk;
    }
    int rc = shq_launch_grav_walk(ctx, params, nullptr, count, update_potential, walk_mode, first);
    if(rc == SHQ_OK)
        rc = shq_launch_grav_postprocess(ctx, params, nullptr, count, update_potential, first);
    ctx->stream = main;
    if(ctx->stream_walk) {
        SHQ_HIP(hipEventRecord(ctx->ev_walk_out, ctx->stream_walk));
        SHQ_HIP(hipStreamWaitEvent(main, ctx->ev_walk_out, 0));
    }
    SHQ_TRY(rc);
    ctx->grav_raw = false;
    ctx->last_stats.ntargets = first == 0 ? count : ctx->last_stats.ntargets + count;
    return SHQ_OK;
}

namespace {
/* one thread per entry; the first entry of a run of equal places adds the whole run in order */
__global__ void grav_reduce_kernel(long long n, const int32_t *__restrict__ place, const shq_grav_result *__restrict__ res, double *acc,
                                   double *pot, int update_potential)
{
    const long long k = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= n)
        return;
    const int32_t pl = place[k];
    if(k > 0 && place[k - 1] == pl)
        return;
    double a0 = acc[3 * (long long) pl], a1 = acc[3 * (long long) pl + 1], a2 = acc[3 * (long long) pl + 2];
    double p = update_potential ? pot[pl] : 0.0;
    for(long long j = k; j < n && place[j] == pl; j++) {
        a0 += res[j].Acc[0];
        a1 += res[j].Acc[1];
        a2 += res[j].Acc[2];
        p += res[j].Potential;
    }
    acc[3 * (long long) pl] = a0;
    acc[3 * (long long) pl + 1] = a1;
    acc[3 * (long long) pl + 2] = a2;
    if(update_potential)
        pot[pl] = p;
}
} // namespace

extern "C" int shq_grav_reduce_export_results(shq_context *ctx, const int32_t *place, const shq_grav_result *results, int64_t n,
                                              int update_potential)
{
    SHQ_CHECK(ctx, SHQ_ERR_INVALID, "null context");
    SHQ_CHECK(n >= 0 && (n == 0 || (place && results)), SHQ_ERR_INVALID, "reduce_export_results: bad arguments");
    SHQ_CHECK(ctx->have_parts && ctx->grav_raw, SHQ_ERR_STATE,
              "reduce_export_results: needs the raw sums of a shq_grav_short_run with SHQ_WALK_DEFER_POSTPROCESS");
    if(n == 0)
        return SHQ_OK;
    SHQ_HIP(hipSetDevice(ctx->device));
    /* group the entries by target, keeping their order within a target (the reference's table is built per
     * particle, so it normally is grouped already) */
    bool grouped = true;
    for(int64_t k = 0; k < n; k++) {
        SHQ_CHECK(place[k] >= 0 && place[k] < ctx->numpart, SHQ_ERR_INVALID, "reduce_export_results: place[%ld] = %d out of range", (long) k, place[k]);
        if(k > 0 && place[k] < place[k - 1])
            grouped = false;
    }
    std::vector<int32_t> hp;
    std::vector<shq_grav_result> hr;
    if(!grouped) {
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
    SHQ_TRY(ctx->active.reserve((size_t) n));
    SHQ_TRY(ctx->gq_res.reserve((size_t) n));
    SHQ_HIP(hipMemcpyAsync(ctx->active.ptr, place, sizeof(int32_t) * n, hipMemcpyHostToDevice, ctx->stream));
    SHQ_HIP(hipMemcpyAsync(ctx->gq_res.ptr, results, sizeof(shq_grav_result) * n, hipMemcpyHostToDevice, ctx->stream));
    grav_reduce_kernel<<<dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, ctx->stream>>>(n, ctx->active.ptr, ctx->gq_res.ptr, ctx->acc.ptr,
                                                                                         ctx->pot.ptr, update_potential);
    SHQ_HIP(hipGetLastError());
    SHQ_HIP(hipStreamSynchronize(ctx->stream)); /* the host staging vectors die here */
    return SHQ_OK;
}

extern "C" int shq_grav_postprocess(shq_context *ctx, const shq_grav_params *params, const int32_t *active, int64_t nactive,
                                    int update_potential)
{
    SHQ_CHECK(ctx && params, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->have_parts && ctx->grav_raw, SHQ_ERR_STATE, "grav_postprocess: no deferred walk result on the device");
    SHQ_HIP(hipSetDevice(ctx->device));
    const int32_t *d_active = nullptr;
    int64_t nt = 0;
    SHQ_TRY(shq_resolve_active(ctx, active, nactive, ctx->nlocal, &d_active, &nt));
    SHQ_TRY(shq_launch_grav_postprocess(ctx, params, d_active, nt, update_potential));
    ctx->grav_raw = false;
    return SHQ_OK;
}

extern "C" int shq_grav_short_download(shq_context *ctx, double (*accel)[3], double *potential,
                                       int64_t *ninteractions, shq_walk_stats *stats)
{
    SHQ_CHECK(ctx, SHQ_ERR_INVALID, "null context");
    SHQ_CHECK(ctx->have_parts, SHQ_ERR_STATE, "nothing to download");
    SHQ_HIP(hipSetDevice(ctx->device));
    SHQ_TRY(shq_walk_check_status(ctx, true)); /* a pair stack that ran full in ANY launch since the last report, not just the last one */
    const int64_t n = ctx->numpart;
    if(accel && n > 0)
        SHQ_HIP(hipMemcpyAsync(accel, ctx->acc.ptr, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, ctx->stream));
    if(potential && n > 0)
        SHQ_HIP(hipMemcpyAsync(potential, ctx->pot.ptr, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    std::vector<int32_t> h_nint;
    if(ninteractions && n > 0) {
        h_nint.resize(n);
        SHQ_HIP(hipMemcpyAsync(h_nint.data(), ctx->nint.ptr, sizeof(int32_t) * n, hipMemcpyDeviceToHost, ctx->stream));
    }
    GravStatsDev gs = {};
    if(stats && ctx->gstats.ptr)
        SHQ_HIP(hipMemcpyAsync(&gs, ctx->gstats.ptr, sizeof(gs), hipMemcpyDeviceToHost, ctx->stream));
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    if(ninteractions)
        for(int64_t i = 0; i < n; i++)
            ninteractions[i] = h_nint[i];
    if(stats) {
        stats->ntargets = ctx->last_stats.ntargets;
        stats->ninteractions = (int64_t) gs.ninteractions;
        stats->nnodes_visited = (int64_t) gs.nvisited;
        stats->nwave_interactions = (int64_t) gs.nwave_applies;
        stats->nwave_node_interactions = (int64_t) gs.nwave_node_applies;
        stats->nnode_interactions = (int64_t) gs.nnode_interactions;
        stats->min_interactions = stats->ntargets > 0 ? gs.min_int : 0;
        stats->max_interactions = gs.max_int;
        if(ctx->walk_stats == 2) { /* diagnostic: rounds by participating lanes, to stderr */
            const char *names[3] = {"visit", "node", "leaf"};
            const unsigned long long *h[3] = {gs.hist_visit, gs.hist_node, gs.hist_leaf};
            for(int k = 0; k < 3; k++) {
                fprintf(stderr, "[shq] walk rounds by lanes (1-8 .. 57-64) %-5s:", names[k]);
                for(int b = 0; b < 8; b++)
                    fprintf(stderr, " %llu", h[k][b]);
                fprintf(stderr, "\n");
            }
            const double nw = (double) ((stats->ntargets + 63) / 64);
            fprintf(stderr, "[shq] interactions in rounds of <= 8 lanes: %.1f per lane, wave max %.1f; <= 16 lanes: %.1f per lane, wave max %.1f\n",
                    gs.lonely[0] / nw / 64., gs.lonely[1] / nw, gs.lonely[2] / nw / 64., gs.lonely[3] / nw);
        }
        float ms = 0;
        if(stats->ntargets > 0 && hipEventElapsedTime(&ms, ctx->ev_begin[SHQ_NTIMERS - 1], ctx->ev_end[SHQ_NTIMERS - 1]) == hipSuccess)
            stats->kernel_ms = ms;
        else
            stats->kernel_ms = 0;
    }
    return SHQ_OK;
}

/* TreeWalk::ev_secondary (treewalk2.h:618-700) for the gravity walk: imported queries against the local tree */
extern "C" int shq_grav_short_secondary(shq_context *ctx, const shq_grav_params *params, const shq_grav_query *queries, int64_t nq,
                                        shq_grav_result *results, int64_t *ninteractions, int update_potential)
{
    SHQ_CHECK(ctx && params && (nq == 0 || (queries && results)), SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->have_parts && ctx->have_tree, SHQ_ERR_STATE, "grav_short_secondary: upload particles and tree first");
    SHQ_CHECK(nq >= 0 && nq < (1ll << 31), SHQ_ERR_INVALID, "bad query count");
    SHQ_HIP(hipSetDevice(ctx->device));
    if(nq == 0)
        return SHQ_OK;
    const size_t n = (size_t) nq;
    std::vector<double4> hpos(n);
    std::vector<double> hold(n);
    std::vector<int32_t> hstart(4 * n);
    const int64_t fn = ctx->firstnode, nn = ctx->numnodes;
    const int64_t nall = ctx->node_rank.empty() ? nn : (int64_t) ctx->node_rank.size();
    for(size_t q = 0; q < n; q++) {
        hpos[q] = make_double4(queries[q].Pos[0], queries[q].Pos[1], queries[q].Pos[2], 0.0);
        hold[q] = queries[q].OldAcc;
        int32_t st[4];
        int ns = 0;
        for(int k = 0; k < 4 && queries[q].NodeList[k] >= 0; k++) { /* -1 terminates the list, gravshort2.hpp:249-250 */
            const int64_t no = queries[q].NodeList[k];
            SHQ_CHECK(no >= fn && no < fn + nall, SHQ_ERR_INVALID, "query %ld: NodeList entry %ld is not a local node", (long) q, (long) no);
            const int32_t r = ctx->node_rank.empty() ? (int32_t) (no - fn) : ctx->node_rank[(size_t) (no - fn)];
            SHQ_CHECK(r >= 0 && r < nn, SHQ_ERR_INVALID, "query %ld: NodeList entry %ld is not reachable from the root", (long) q, (long) no);
            st[ns++] = r;
        }
        std::sort(st, st + ns); /* branches are disjoint; the wave cursor meets them in pre-order */
        for(int k = 0; k < 4; k++)
            hstart[4 * q + k] = k < ns ? st[k] : -1;
    }
    DevBuf<double4> dpos;
    DevBuf<double> dold, dacc, dpot;
    DevBuf<int32_t> dstart, dnint;
    int rc = SHQ_OK;
    auto run = [&]() -> int {
        SHQ_TRY(dpos.reserve(n));
        SHQ_TRY(dold.reserve(n));
        SHQ_TRY(dacc.reserve(3 * n));
        SHQ_TRY(dpot.reserve(n));
        SHQ_TRY(dstart.reserve(4 * n));
        SHQ_TRY(dnint.reserve(n));
        SHQ_HIP(hipMemcpyAsync(dpos.ptr, hpos.data(), sizeof(double4) * n, hipMemcpyHostToDevice, ctx->stream));
        SHQ_HIP(hipMemcpyAsync(dold.ptr, hold.data(), sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
        SHQ_HIP(hipMemcpyAsync(dstart.ptr, hstart.data(), sizeof(int32_t) * 4 * n, hipMemcpyHostToDevice, ctx->stream));
        SHQ_HIP(hipMemsetAsync(dpot.ptr, 0, sizeof(double) * n, ctx->stream));
        SHQ_TRY(shq_launch_grav_walk_ghosts(ctx, params, dpos.ptr, dold.ptr, dstart.ptr, nq, dacc.ptr, dpot.ptr, dnint.ptr, update_potential));
        std::vector<double> hacc(3 * n), hpot(n);
        std::vector<int32_t> hn(n);
        SHQ_HIP(hipMemcpyAsync(hacc.data(), dacc.ptr, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, ctx->stream));
        SHQ_HIP(hipMemcpyAsync(hpot.data(), dpot.ptr, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
        SHQ_HIP(hipMemcpyAsync(hn.data(), dnint.ptr, sizeof(int32_t) * n, hipMemcpyDeviceToHost, ctx->stream));
        SHQ_HIP(hipStreamSynchronize(ctx->stream));
        for(size_t q = 0; q < n; q++) {
            results[q].Acc[0] = hacc[3 * q];
            results[q].Acc[1] = hacc[3 * q + 1];
            results[q].Acc[2] = hacc[3 * q + 2];
            results[q].Potential = hpot[q];
            if(ninteractions)
                ninteractions[q] = hn[q];
        }
        return SHQ_OK;
    };
    rc = run();
    (void) hipStreamSynchronize(ctx->stream);
    dpos.release(); dold.release(); dacc.release(); dpot.release(); dstart.release(); dnint.release();
    return rc;
}

extern "C" int shq_grav_refresh_oldacc(shq_context *ctx, double G)
{
    SHQ_CHECK(ctx && G > 0, SHQ_ERR_INVALID, "bad argument");
    SHQ_HIP(hipSetDevice(ctx->device));
    SHQ_TRY(shq_join_pm(ctx));
    return shq_launch_oldacc(ctx, G);
}

extern "C" int shq_grav_short_tree(shq_context *ctx, const shq_tree_view *tree, const shq_part_view *parts,
                                   const int32_t *active, int64_t nactive, const shq_grav_params *params,
                                   double (*accel)[3], int update_potential, int walk_mode, shq_walk_stats *stats)
{
    SHQ_CHECK(ctx && tree && parts && params && accel, SHQ_ERR_INVALID, "null argument");
    SHQ_TRY(shq_particles_upload(ctx, parts));
    SHQ_TRY(shq_tree_upload(ctx, tree));
    SHQ_TRY(shq_grav_refresh_oldacc(ctx, params->G));
    SHQ_TRY(shq_grav_short_run(ctx, params, active, nactive, update_potential, walk_mode));
    SHQ_TRY(shq_grav_short_download(ctx, nullptr, nullptr, nullptr, stats));
    /* results back: only the walked targets are assigned (reduce<PRIMARY>, localtreewalk2.h:39).  With an active list the
     * targets' rows are gathered on the device first; then chunks of rows come down into pinned staging, each scattered into the
     * caller's arrays by the host threads while the next one is in flight. */
    const int64_t n = parts->numpart;
    const int64_t nt = active ? nactive : n;
    const double *d_acc = ctx->acc.ptr, *d_pot = ctx->pot.ptr;
    if(active && nt > 0) {
        SHQ_TRY(ctx->gq_res.reserve((size_t) nt)); /* 32 bytes per row: Acc[3], Potential */
        gather_rows_kernel<<<dim3((unsigned) ((nt + 255) / 256)), dim3(256), 0, ctx->stream>>>(nt, ctx->active.ptr, ctx->acc.ptr, ctx->pot.ptr,
                                                                                          reinterpret_cast<double *>(ctx->gq_res.ptr));
        SHQ_HIP(hipGetLastError());
    }
    const int64_t CH = 1 << 21;
    SHQ_TRY(ctx->stage.reserve(2 * (size_t) CH * 32 + 256));
    hipEvent_t ev[2] = {ctx->ev_begin[SHQ_NTIMERS - 2], ctx->ev_end[SHQ_NTIMERS - 2]};
    auto scatter = [&](int64_t c0, int64_t m, const char *base) {
        const double *rows = reinterpret_cast<const double *>(base);          /* active: [m][4]; else acc [m][3] then pot [m] */
        const double *pots = rows + 3 * CH;
        parallel_for(m, [&](int64_t lo, int64_t hi) {
            for(int64_t k = lo; k < hi; k++) {
                const int64_t i = active ? active[c0 + k] : c0 + k;
                const double *r = active ? rows + 4 * k : rows + 3 * k;
                const double p = active ? r[3] : pots[k];
                accel[i][0] = r[0]; accel[i][1] = r[1]; accel[i][2] = r[2];
                if(update_potential) {
                    if(parts->off_treeacc != SHQ_NOFIELD) {
                        double *a = field_w<double>(parts, i, parts->off_treeacc);
                        a[0] = r[0]; a[1] = r[1]; a[2] = r[2];
                    }
                    if(parts->off_potential != SHQ_NOFIELD)
                        *field_w<double>(parts, i, parts->off_potential) = p;
                }
            }
        });
    };
    int64_t prev_c0 = -1, prev_m = 0;
    int nchunk = 0;
    for(int64_t c0 = 0; c0 < nt; c0 += CH, nchunk++) {
        const int64_t m = std::min<int64_t>(CH, nt - c0);
        const int sl = nchunk & 1;
        char *base = ctx->stage.ptr + (size_t) sl * CH * 32;
        if(active)
            SHQ_HIP(hipMemcpyAsync(base, reinterpret_cast<const double *>(ctx->gq_res.ptr) + 4 * c0, 32 * (size_t) m, hipMemcpyDeviceToHost, ctx->stream));
        else {
            SHQ_HIP(hipMemcpyAsync(base, d_acc + 3 * c0, 24 * (size_t) m, hipMemcpyDeviceToHost, ctx->stream));
            if(update_potential)
                SHQ_HIP(hipMemcpyAsync(base + 24 * (size_t) CH, d_pot + c0, 8 * (size_t) m, hipMemcpyDeviceToHost, ctx->stream));
        }
        SHQ_HIP(hipEventRecord(ev[sl], ctx->stream));
        if(prev_c0 >= 0) /* the previous chunk has landed in the other half: scatter it while this one is in flight */
        {
            SHQ_HIP(hipEventSynchronize(ev[sl ^ 1]));
            scatter(prev_c0, prev_m, ctx->stage.ptr + (size_t) (sl ^ 1) * CH * 32);
        }
        prev_c0 = c0;
        prev_m = m;
    }
    if(prev_c0 >= 0) {
        SHQ_HIP(hipEventSynchronize(ev[(nchunk - 1) & 1]));
        scatter(prev_c0, prev_m, ctx->stage.ptr + (size_t) ((nchunk - 1) & 1) * CH * 32);
    }
    return SHQ_OK;
}

/* ---- PM ----------------------------------------------------------------------------------- */

extern "C" int shq_pm_run(shq_context *ctx, const shq_pm_params *pm)
{
    SHQ_CHECK(ctx && pm, SHQ_ERR_INVALID, "null argument");
    SHQ_HIP(hipSetDevice(ctx->device));
    SHQ_TRY(shq_join_pm(ctx));
    ctx->pm_prestarted = false;
    if(!ctx->pm_overlap)
        return shq_pm_execute(ctx, pm);
    return shq_pm_run_on_pm_stream(ctx, pm, false);
}

/* The PM reads only the positions (and, when its readout forms OldAcc, FullTreeGravAccel) and writes only the mesh, GravPM, the PM potential
 * and OldAcc: it runs on its own stream behind everything queued so far, and the main stream goes on.  Consumers of its results join it
 * (shq_join_pm). */
int shq_pm_run_on_pm_stream(shq_context *ctx, const shq_pm_params *pm, bool low_priority)
{
    /* low_priority (shq_pm_start): the library's lowest-priority stream, the pair kernel's, idle between two walks.  An FFT pass holds
     * 464 of a SIMD's 512 registers and 141 of a CU's 160 KB of LDS: nothing else starts on a CU while two of its workgroups are resident,
     * and on the high-priority PM stream the next pass's workgroups take every slot that frees up - the tree build beside it took 17 ms
     * instead of 5.5.  At the lowest priority the tree build's kernels get the freed slots first. */
    hipStream_t ps = low_priority && ctx->stream_pair ? ctx->stream_pair : ctx->stream_pm;
    SHQ_HIP(hipEventRecord(ctx->ev_pm_ready, ctx->stream));
    SHQ_HIP(hipStreamWaitEvent(ps, ctx->ev_pm_ready, 0));
    hipStream_t main_stream = ctx->stream;
    ctx->stream = ps;
    const int rc = shq_pm_execute(ctx, pm);
    ctx->stream = main_stream;
    if(rc != SHQ_OK) {
        (void) hipStreamSynchronize(ps);
        return rc;
    }
    SHQ_HIP(hipEventRecord(ctx->ev_pm_done, ps));
    ctx->pm_pending = true;
    return SHQ_OK;
}

/* gravpm_force started EARLY: the PM needs the drifted positions and nothing of the tree, so a resident step queues it on the library's
 * second stream and builds the tree meanwhile (force_tree_full and gravpm_force have no order between them in the reference either:
 * run.cpp:476-538 builds the tree first only because the domain decomposition comes with it).  G > 0: the readout forms OldAcc from
 * FullTreeGravAccel of the last step and the new GravPM, as shq_treepm_step does.  shq_treepm_step (same Nmesh) then joins this PM instead
 * of running its own; every other consumer of PM results joins it as well.  A drift or a particle upload discards it. */
extern "C" int shq_pm_start(shq_context *ctx, const shq_pm_params *pm, double G)
{
    SHQ_CHECK(ctx && pm, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->have_parts, SHQ_ERR_STATE, "pm_start: particles must be uploaded first");
    SHQ_HIP(hipSetDevice(ctx->device));
    SHQ_TRY(shq_join_pm(ctx));
    SHQ_TRY(shq_walk_check_status(ctx, false));
    ctx->readout_oldacc_G = (G > 0 && ctx->numpart > 0 && ctx->treeacc.ptr && ctx->oldacc.ptr) ? G : 0.0;
    const bool oldacc_done = ctx->readout_oldacc_G > 0;
    const int rc = shq_pm_run_on_pm_stream(ctx, pm, true);
    ctx->readout_oldacc_G = 0;
    SHQ_TRY(rc);
    ctx->pm_prestarted = true;
    ctx->pm_prestarted_oldacc = oldacc_done;
    ctx->pm_prestarted_nmesh = pm->Nmesh;
    return SHQ_OK;
}

/* gravpm_force followed by grav_short_tree for every particle — the force part of a PM step in the reference's own order
 * (run.cpp:518-563: "gravpm_force() needs to be run first", because the walk's opening criterion reads the new GravPM through
 * grav_get_abs_accel).  Deposit and transforms run as in shq_pm_run; the readout of GravPM / the PM potential and the OldAcc
 * refresh happen in the prologue of the walk's tasks, target by target, where their scattered loads hide behind the arithmetic
 * of the other waves (the walk leaves the memory system idle).  Falls back to the three separate launches when the walk at hand
 * cannot carry them (Barnes-Hut seeding walk, diagnostic counters, a mesh beyond 2^32 cells, ...).  Same bits either way. */
extern "C" int shq_treepm_step(shq_context *ctx, const shq_pm_params *pm, const shq_grav_params *params, int update_potential, int walk_mode)
{
    SHQ_CHECK(ctx && pm && params, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->have_parts && ctx->have_tree, SHQ_ERR_STATE, "treepm_step: upload particles and tree first");
    SHQ_CHECK((walk_mode & ~SHQ_WALK_TREE_ORDER) == SHQ_WALK_EXACT, SHQ_ERR_INVALID, "treepm_step: walk_mode is SHQ_WALK_EXACT, optionally | SHQ_WALK_TREE_ORDER");
    SHQ_HIP(hipSetDevice(ctx->device));
    SHQ_TRY(shq_walk_check_status(ctx, false)); /* an earlier step of a resident loop whose pair kernel failed: no new step on its forces */
    const bool prestarted = ctx->pm_prestarted && ctx->pm_pending && ctx->pm_prestarted_nmesh == pm->Nmesh;
    SHQ_TRY(shq_join_pm(ctx));
    ctx->pm_prestarted = false;
    int64_t n = ctx->nlocal;
    const int32_t *d_targets = nullptr;
    if(walk_mode & SHQ_WALK_TREE_ORDER) { /* every own particle of the tree, in leaf order */
        SHQ_TRY(shq_build_tree_targets(ctx));
        d_targets = ctx->tree_targets.ptr;
        n = ctx->ntree_targets;
    }
    const bool fuse = !prestarted && !ctx->pm_overlap && ctx->treepm_fuse && pm->Nmesh >= 4 &&
                      (size_t) pm->Nmesh * pm->Nmesh * (size_t) (pm->Nmesh + 10) < (1ull << 29) &&
                      shq_walk_can_fuse_readout_pre(ctx, params, n);
    /* not fused: the readout kernel forms OldAcc as it stores GravPM (the operations of shq_grav_refresh_oldacc, one pass less) */
    bool oldacc_done = ctx->pm_prestarted_oldacc;
    if(!prestarted) { /* (a PM started by shq_pm_start for these positions has just been joined: nothing to run) */
        ctx->readout_oldacc_G = (!fuse && ctx->numpart > 0 && ctx->treeacc.ptr && ctx->oldacc.ptr) ? params->G : 0.0;
        oldacc_done = ctx->readout_oldacc_G > 0;
        const int rc_pm = shq_pm_execute(ctx, pm, !fuse);
        ctx->readout_oldacc_G = 0;
        SHQ_TRY(rc_pm);
    }
    if(!fuse) {
        if(!oldacc_done)
            SHQ_TRY(shq_grav_refresh_oldacc(ctx, params->G));
    } else {
        ctx->fuse_G = params->G;
        ctx->fuse_readout = true;
    }
    const int rc = shq_launch_grav_walk(ctx, params, d_targets, n, update_potential, SHQ_WALK_EXACT);
    ctx->fuse_readout = false;
    SHQ_TRY(rc);
    SHQ_TRY(shq_launch_grav_postprocess(ctx, params, d_targets, n, update_potential));
    ctx->grav_raw = false;
    ctx->last_stats.ntargets = n;
    ctx->last_step_fused = fuse;
    return SHQ_OK;
}

extern "C" int shq_treepm_set_fuse(shq_context *ctx, int enable)
{
    SHQ_CHECK(ctx, SHQ_ERR_INVALID, "null context");
    ctx->treepm_fuse = enable != 0;
    return SHQ_OK;
}

extern "C" int shq_treepm_last_fused(shq_context *ctx, int *fused)
{
    SHQ_CHECK(ctx && fused, SHQ_ERR_INVALID, "null argument");
    *fused = ctx->last_step_fused ? 1 : 0;
    return SHQ_OK;
}

extern "C" int shq_pm_download(shq_context *ctx, double (*gravpm)[3], double *pm_potential)
{
    SHQ_CHECK(ctx, SHQ_ERR_INVALID, "null context");
    SHQ_CHECK(ctx->have_pm_result, SHQ_ERR_STATE, "pm_download before pm_run");
    SHQ_HIP(hipSetDevice(ctx->device));
    SHQ_TRY(shq_join_pm(ctx));
    const int64_t n = ctx->numpart;
    if(gravpm && n > 0)
        SHQ_HIP(hipMemcpyAsync(gravpm, ctx->gravpm.ptr, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, ctx->stream));
    if(pm_potential && n > 0)
        SHQ_HIP(hipMemcpyAsync(pm_potential, ctx->pmpot.ptr, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    return SHQ_OK;
}

extern "C" int shq_pm_force(shq_context *ctx, const shq_pm_params *pm, const shq_part_view *parts,
                            double (*gravpm)[3], double *potential)
{
    SHQ_CHECK(ctx && pm && parts && gravpm, SHQ_ERR_INVALID, "null argument");
    SHQ_TRY(shq_particles_upload(ctx, parts));
    SHQ_TRY(shq_pm_run(ctx, pm));
    /* GravPM (assigned) and the PM potential (added: readout_potential adds, gravpm.cpp:489-491) come down through pinned
     * staging in chunks; the host threads move each chunk into the caller's arrays while the next one is in flight */
    SHQ_TRY(shq_join_pm(ctx));
    const int64_t n = parts->numpart;
    const int64_t CH = 1 << 21;
    SHQ_TRY(ctx->stage.reserve(2 * (size_t) CH * 32 + 256));
    hipEvent_t ev[2] = {ctx->ev_begin[SHQ_NTIMERS - 2], ctx->ev_end[SHQ_NTIMERS - 2]};
    auto land = [&](int64_t c0, int64_t m, const char *base) {
        const double *g = reinterpret_cast<const double *>(base), *pp = g + 3 * CH;
        parallel_for(m, [&](int64_t lo, int64_t hi) {
            memcpy(&gravpm[c0 + lo][0], g + 3 * lo, sizeof(double) * 3 * (size_t) (hi - lo));
            if(potential)
                for(int64_t k = lo; k < hi; k++)
                    potential[c0 + k] += pp[k];
        });
    };
    int64_t prev_c0 = -1, prev_m = 0;
    int nchunk = 0;
    for(int64_t c0 = 0; c0 < n; c0 += CH, nchunk++) {
        const int64_t m = std::min<int64_t>(CH, n - c0);
        const int sl = nchunk & 1;
        char *base = ctx->stage.ptr + (size_t) sl * CH * 32;
        SHQ_HIP(hipMemcpyAsync(base, ctx->gravpm.ptr + 3 * c0, 24 * (size_t) m, hipMemcpyDeviceToHost, ctx->stream));
        if(potential)
            SHQ_HIP(hipMemcpyAsync(base + 24 * (size_t) CH, ctx->pmpot.ptr + c0, 8 * (size_t) m, hipMemcpyDeviceToHost, ctx->stream));
        SHQ_HIP(hipEventRecord(ev[sl], ctx->stream));
        if(prev_c0 >= 0) {
            SHQ_HIP(hipEventSynchronize(ev[sl ^ 1]));
            land(prev_c0, prev_m, ctx->stage.ptr + (size_t) (sl ^ 1) * CH * 32);
        }
        prev_c0 = c0;
        prev_m = m;
    }
    if(prev_c0 >= 0) {
        SHQ_HIP(hipEventSynchronize(ev[(nchunk - 1) & 1]));
        land(prev_c0, prev_m, ctx->stage.ptr + (size_t) ((nchunk - 1) & 1) * CH * 32);
    }
    return SHQ_OK;
}

extern "C" int shq_pm_phase_ms(shq_context *ctx, double ms[6])
{
    SHQ_CHECK(ctx && ms, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->have_pm_result, SHQ_ERR_STATE, "pm_phase_ms before pm_run");
    SHQ_HIP(hipSetDevice(ctx->device));
    SHQ_TRY(shq_join_pm(ctx));
    SHQ_HIP(hipEventSynchronize(ctx->ev_begin[13]));
    for(int i = 0; i < 5; i++) {
        float f = 0;
        SHQ_HIP(hipEventElapsedTime(&f, ctx->ev_begin[8 + i], ctx->ev_begin[9 + i]));
        ms[i] = f;
    }
    float f = 0;
    SHQ_HIP(hipEventElapsedTime(&f, ctx->ev_begin[8], ctx->ev_begin[13]));
    ms[5] = f;
    return SHQ_OK;
}

extern "C" int shq_pm_set_debug(shq_context *ctx, int keep_meshes)
{
    SHQ_CHECK(ctx, SHQ_ERR_INVALID, "null context");
    ctx->pm_keep = keep_meshes;
    return SHQ_OK;
}

extern "C" int shq_pm_set_mesh_scrub(shq_context *ctx, int enable)
{
    SHQ_CHECK(ctx, SHQ_ERR_INVALID, "null context");
    ctx->pm_scrub = enable != 0;
    return SHQ_OK;
}

extern "C" int shq_pm_mesh_prezeroed(shq_context *ctx, int *zeroed)
{
    SHQ_CHECK(ctx && zeroed, SHQ_ERR_INVALID, "null argument");
    *zeroed = ctx->mesh_zeroed ? 1 : 0;
    return SHQ_OK;
}

extern "C" int shq_pm_download_mesh(shq_context *ctx, int which, double *mesh)
{
    SHQ_CHECK(ctx && mesh, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->have_pm_result && ctx->pm_keep, SHQ_ERR_STATE, "pm_download_mesh needs shq_pm_set_debug(1) before shq_pm_run");
    SHQ_CHECK(which == 0 || which == 1, SHQ_ERR_INVALID, "which must be 0 or 1");
    const size_t tot = (size_t) ctx->pm_nmesh * ctx->pm_nmesh * ctx->pm_nmesh;
    SHQ_HIP(hipSetDevice(ctx->device));
    SHQ_TRY(shq_join_pm(ctx));
    SHQ_HIP(hipMemcpyAsync(mesh, which == 0 ? ctx->dbg_rho.ptr : ctx->dbg_pot.ptr, tot * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    return SHQ_OK;
}

extern "C" int shq_fft_r2c(shq_context *ctx, int Nmesh, const double *real, double *complx)
{
    SHQ_CHECK(ctx && real && complx, SHQ_ERR_INVALID, "null argument");
    SHQ_HIP(hipSetDevice(ctx->device));
    return shq_fft_roundtrip_r2c(ctx, Nmesh, real, complx, true);
}
extern "C" int shq_fft_c2r(shq_context *ctx, int Nmesh, const double *complx, double *real)
{
    SHQ_CHECK(ctx && real && complx, SHQ_ERR_INVALID, "null argument");
    SHQ_HIP(hipSetDevice(ctx->device));
    return shq_fft_roundtrip_c2r(ctx, Nmesh, complx, real, true);
}
extern "C" int shq_fft_r2c_xyz(shq_context *ctx, int Nmesh, const double *real, double *complx)
{
    SHQ_CHECK(ctx && real && complx, SHQ_ERR_INVALID, "null argument");
    SHQ_HIP(hipSetDevice(ctx->device));
    return shq_fft_roundtrip_r2c(ctx, Nmesh, real, complx, false);
}
extern "C" int shq_fft_c2r_xyz(shq_context *ctx, int Nmesh, const double *complx, double *real)
{
    SHQ_CHECK(ctx && real && complx, SHQ_ERR_INVALID, "null argument");
    SHQ_HIP(hipSetDevice(ctx->device));
    return shq_fft_roundtrip_c2r(ctx, Nmesh, complx, real, false);
}
