/* toptree.hip — export detection: which remote top-leaves must a target visit.
 *
 * The reference's distributed tree walk first runs a "top-tree" walk per target over the TopLevel nodes of the
 * local tree; every pseudo node (a top-level leaf owned by another rank) that the target would open becomes an
 * entry of the export table, consecutive leaves of one task coalescing into one entry's NodeList[4]:
 *   GravTopTreeWalk::toptree_visit<COUNT|EXPORT>      libgadget/gravshort2.hpp:362-438
 *   TopTreeWalk::toptree_visit (cull_node)            libgadget/localtreewalk2.h:210-259, 154-182
 *   export_particle / export_count                    libgadget/localtreewalk2.h:269-324
 *   ev_count_exports / ev_toptree                     libgadget/treewalk2.cuh:243-334 (count, inclusive scan, fill)
 *
 * The top tree has a few hundred to a few thousand nodes and sits in L2; one thread walks one target (the walks
 * are short and the export state — last task, NodeList fill — is per target and sequential).  Decisions use the
 * reference's own expressions with fp contraction off, so they equal the oracle's bit for bit.  Two passes as in
 * the reference: count, inclusive scan (rocPRIM), fill at the scanned offsets. */
#include <cstring>
#include <vector>
#include "common.hpp"
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_radix_sort.hpp>

namespace {

inline unsigned nblk(long long n, int t = 256) { return (unsigned) ((n + t - 1) / t); }

enum { TOP_INTERNAL = 0, TOP_LOCAL_LEAF = 1, TOP_PSEUDO = 2 };

__device__ __forceinline__ double nearest(double x, double Box) /* NEAREST, partmanager.h:99 */
{
    return (x > 0.5 * Box) ? (x - Box) : ((x < -0.5 * Box) ? (x + Box) : x);
}

struct GravTopArgs {
    double Box, rcut, rcut2, errtol, theta2;
    int useBH;
    const double *oldacc;
};
struct NgbTopArgs {
    double Box;
    int symmetric;
    const double *hsml;
};

/* true: do not open (discarded, or accepted as a monopole): gravshort2.hpp:392-400 */
__device__ __forceinline__ bool grav_skip(const TopNodeG &nd, const double4 p, double aold, const GravTopArgs &a)
{
#pragma clang fp contract(off)
    double dx[3] = {nearest(nd.cofm[0] - p.x, a.Box), nearest(nd.cofm[1] - p.y, a.Box), nearest(nd.cofm[2] - p.z, a.Box)};
    const double r2 = dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2];
    const double cx = fabs(nearest(nd.center[0] - p.x, a.Box)), cy = fabs(nearest(nd.center[1] - p.y, a.Box)),
                 cz = fabs(nearest(nd.center[2] - p.z, a.Box));
    /* shall_we_discard_node, gravshort2.hpp:152-167 */
    if(r2 > a.rcut2) {
        const double eff = a.rcut + 0.5 * nd.len;
        if(cx > eff || cy > eff || cz > eff)
            return true;
    }
    /* shall_we_open_node, gravshort2.hpp:172-193 */
    if(a.useBH == 0 && nd.mass * nd.len * nd.len > r2 * r2 * aold)
        return false;
    if(nd.len * nd.len / r2 > a.theta2)
        return false;
    const double inside = 0.6 * nd.len;
    if(cx < inside && cy < inside && cz < inside)
        return false;
    return true;
}

/* cull_node<symmetric> == 0, localtreewalk2.h:154-182 */
__device__ __forceinline__ bool ngb_skip(const TopNodeG &nd, const double4 p, double hsml, const NgbTopArgs &a)
{
#pragma clang fp contract(off)
    double dist = (a.symmetric ? fmax(nd.hmax, hsml) : hsml) + 0.5 * nd.len;
    double r2 = 0;
    const double pos[3] = {p.x, p.y, p.z};
    for(int d = 0; d < 3; d++) {
        const double dx = nearest(nd.center[d] - pos[d], a.Box);
        if(dx > dist || dx < -dist)
            return true;
        r2 += dx * dx;
    }
    dist += 0.5 * (1.7320508075688772 - 1.0) * nd.len;
    return r2 > dist * dist;
}

/* GRAV: gravity criteria, else cull_node.  FILL: write the table at the scanned offsets, else count. */
template <bool GRAV, bool FILL>
__global__ __launch_bounds__(256) void toptree_kernel(long long nt, const int32_t *__restrict__ targets, const double4 *__restrict__ posm,
                                                       const TopNodeG *__restrict__ top, const int2 *__restrict__ leaves,
                                                       GravTopArgs ga, NgbTopArgs na, int32_t *counts, shq_data_index *table)
{
    const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= nt)
        return;
    const int target = targets ? targets[t] : (int) t;
    const double4 p = posm[target];
    const double aux = GRAV ? ga.errtol * ga.oldacc[target] : na.hsml[target];
    shq_data_index *out = nullptr;
    if(FILL) {
        const int32_t lo = t > 0 ? counts[t - 1] : 0;
        if(counts[t] == lo) /* nothing to export: skip the walk (treewalk2.cuh:96-98) */
            return;
        out = table + lo;
    }
    int nexp = 0, lasttask = 0, nodelistindex = 0;
    int no = 0;
    while(no >= 0) {
        const TopNodeG nd = top[no];
        const bool skip = GRAV ? grav_skip(nd, p, aux, ga) : ngb_skip(nd, p, aux, na);
        if(skip || nd.kind == TOP_LOCAL_LEAF) {
            no = nd.sibling;
            continue;
        }
        if(nd.kind == TOP_PSEUDO) {
            const int2 tl = leaves[nd.leaf]; /* Task, treenode */
            /* export_particle / export_count, localtreewalk2.h:269-324 */
            if(nexp >= 1 && lasttask == tl.x && nodelistindex < 4) {
                if(FILL)
                    out[nexp - 1].NodeList[nodelistindex] = tl.y;
                nodelistindex++;
            } else {
                if(FILL) {
                    out[nexp].Task = tl.x;
                    out[nexp].Index = target;
                    out[nexp].NodeList[0] = tl.y;
                    out[nexp].NodeList[1] = out[nexp].NodeList[2] = out[nexp].NodeList[3] = -1;
                }
                nodelistindex = 1;
                lasttask = tl.x;
                nexp++;
            }
            no = nd.sibling;
            continue;
        }
        no = nd.child;
    }
    if(!FILL)
        counts[t] = nexp;
}

/* entries per destination task, and (second use) each entry's slot in the task-ordered send buffer */
__global__ void top_task_count_kernel(long long n, const shq_data_index *__restrict__ table, int ntask, unsigned long long *counts)
{
    const long long k = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(k < n) {
        const int t = table[k].Task;
        if(t >= 0 && t < ntask)
            atomicAdd(&counts[t], 1ull);
    }
}

/* GravTreeQuery records (gravshort2.hpp:123-128) of the table's entries, written at their place in task order:
 * slot[k] = offset of entry k's task + rank of k among that task's entries (table order kept: a stable counting sort) */
__global__ void top_pack_queries_kernel(long long n, const shq_data_index *__restrict__ table, const long long *__restrict__ slot,
                                        const double4 *__restrict__ posm, const double *__restrict__ oldacc, shq_grav_query *q, int32_t *place)
{
    const long long k = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= n)
        return;
    const shq_data_index e = table[k];
    const long long s = slot[k];
    const double4 p = posm[e.Index];
    shq_grav_query o;
    o.Pos[0] = p.x;
    o.Pos[1] = p.y;
    o.Pos[2] = p.z;
    for(int j = 0; j < 4; j++)
        o.NodeList[j] = e.NodeList[j];
    o.OldAcc = oldacc[e.Index];
    q[s] = o;
    place[s] = e.Index;
}

template <bool GRAV>
int run_toptree(shq_context *ctx, const GravTopArgs &ga, const NgbTopArgs &na, const int32_t *active, int64_t nactive,
                int32_t *exportcounts, shq_data_index *table, int64_t capacity, int64_t *nexport, bool resident = false,
                bool active_on_device = false)
{
    const int32_t *d_act = nullptr;
    int64_t nt = 0;
    if(active_on_device && active) {
        d_act = active;
        nt = nactive;
    } else
        SHQ_TRY(shq_resolve_active(ctx, active, nactive, ctx->nlocal, &d_act, &nt));
    ctx->top_ntargets = nt;
    ctx->top_nexport = 0;
    if(nexport)
        *nexport = 0;
    if(nt == 0)
        return SHQ_OK;
    hipStream_t st = ctx->stream;
    SHQ_TRY(ctx->top_counts.reserve((size_t) nt));
    toptree_kernel<GRAV, false><<<dim3(nblk(nt)), dim3(256), 0, st>>>(nt, d_act, ctx->posm.ptr, ctx->topnodes.ptr, ctx->topleaves.ptr, ga, na,
                                                                     ctx->top_counts.ptr, nullptr);
    SHQ_HIP(hipGetLastError());
    size_t tmp = 0;
    SHQ_HIP(rocprim::inclusive_scan(nullptr, tmp, ctx->top_counts.ptr, ctx->top_counts.ptr, (size_t) nt, rocprim::plus<int32_t>(), st));
    SHQ_TRY(ctx->act_temp.reserve(tmp + 16));
    SHQ_HIP(rocprim::inclusive_scan((void *) ctx->act_temp.ptr, tmp, ctx->top_counts.ptr, ctx->top_counts.ptr, (size_t) nt,
                                    rocprim::plus<int32_t>(), st));
    int32_t total = 0;
    SHQ_HIP(hipMemcpyAsync(&total, ctx->top_counts.ptr + (nt - 1), sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if(exportcounts)
        SHQ_HIP(hipMemcpyAsync(exportcounts, ctx->top_counts.ptr, sizeof(int32_t) * nt, hipMemcpyDeviceToHost, st));
    SHQ_HIP(hipStreamSynchronize(st));
    SHQ_CHECK(total >= 0, SHQ_ERR_INVALID, "toptree: more than 2^31 exports");
    if(nexport)
        *nexport = total;
    if((!table && !resident) || total == 0)
        return SHQ_OK;
    SHQ_CHECK(resident || capacity >= total, SHQ_ERR_NOMEM, "toptree: export table holds %ld entries, %d needed", (long) capacity, total);
    SHQ_TRY(ctx->top_table.reserve((size_t) total));
    toptree_kernel<GRAV, true><<<dim3(nblk(nt)), dim3(256), 0, st>>>(nt, d_act, ctx->posm.ptr, ctx->topnodes.ptr, ctx->topleaves.ptr, ga, na,
                                                                    ctx->top_counts.ptr, ctx->top_table.ptr);
    SHQ_HIP(hipGetLastError());
    ctx->top_nexport = total;
    if(table) {
        SHQ_HIP(hipMemcpyAsync(table, ctx->top_table.ptr, sizeof(shq_data_index) * (size_t) total, hipMemcpyDeviceToHost, st));
        SHQ_HIP(hipStreamSynchronize(st));
    }
    return SHQ_OK;
}

} // namespace

extern "C" int shq_toptree_upload(shq_context *ctx, const shq_tree_view *tree, const shq_topleaf *topleaves, int ntopleaves)
{
    SHQ_CHECK(ctx && tree && tree->nodes_base, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ntopleaves >= 0 && (topleaves || ntopleaves == 0), SHQ_ERR_INVALID, "toptree_upload: bad TopLeaves table");
    SHQ_HIP(hipSetDevice(ctx->device));
    ctx->have_toptree = false;
    const int64_t fn = tree->firstnode, nall = tree->numnodes, ln = tree->lastnode;
    const shq_node *src = tree->nodes_base;
    auto valid = [&](int64_t no) { return no >= fn && no < fn + nall; };
    SHQ_CHECK(valid(tree->rootnode), SHQ_ERR_INVALID, "toptree_upload: root %d outside the node array", tree->rootnode);
    SHQ_CHECK(SHQ_NODE_TOPLEVEL(src[tree->rootnode - fn].flags), SHQ_ERR_INVALID,
              "toptree_upload: the root is not flagged TopLevel (tree built without a domain decomposition)");
    /* the nodes toptree_visit can reach: descend from the root through internal top-level nodes only */
    std::vector<int32_t> order;
    std::vector<int32_t> newidx((size_t) nall, -1);
    {
        int64_t no = tree->rootnode;
        while(valid(no)) {
            const int64_t i = no - fn;
            SHQ_CHECK(newidx[i] < 0, SHQ_ERR_INVALID, "toptree_upload: tree threading revisits node %ld", (long) no);
            const shq_node &s = src[i];
            SHQ_CHECK(SHQ_NODE_TOPLEVEL(s.flags), SHQ_ERR_INVALID, "toptree_upload: node %ld below the root is reached without the TopLevel flag", (long) no);
            newidx[i] = (int32_t) order.size();
            order.push_back((int32_t) i);
            const unsigned ct = SHQ_NODE_CHILDTYPE(s.flags);
            const bool open = ct == SHQ_NODE_NODE_TYPE && SHQ_NODE_INTERNALTOPLEVEL(s.flags);
            no = open ? s.suns[0] : s.sibling;
        }
    }
    const size_t nn = order.size();
    std::vector<TopNodeG> h(nn);
    for(size_t j = 0; j < nn; j++) {
        const shq_node &s = src[order[j]];
        TopNodeG g;
        memset(&g, 0, sizeof(g));
        for(int k = 0; k < 3; k++) {
            g.cofm[k] = s.cofm[k];
            g.center[k] = s.center[k];
        }
        g.mass = s.mass;
        g.len = s.len;
        g.hmax = s.hmax;
        g.sibling = valid(s.sibling) ? newidx[s.sibling - fn] : -1;
        g.child = -1;
        g.leaf = -1;
        const unsigned ct = SHQ_NODE_CHILDTYPE(s.flags);
        if(ct == SHQ_PSEUDO_NODE_TYPE) {
            g.kind = TOP_PSEUDO;
            const int64_t leaf = (int64_t) s.suns[0] - ln;
            SHQ_CHECK(leaf >= 0 && leaf < ntopleaves, SHQ_ERR_INVALID, "toptree_upload: pseudo node %ld refers to top leaf %ld of %d",
                      (long) (order[j] + fn), (long) leaf, ntopleaves);
            g.leaf = (int32_t) leaf;
        } else if(ct == SHQ_NODE_NODE_TYPE && SHQ_NODE_INTERNALTOPLEVEL(s.flags)) {
            g.kind = TOP_INTERNAL;
            SHQ_CHECK(valid(s.suns[0]) && newidx[s.suns[0] - fn] >= 0, SHQ_ERR_INVALID, "toptree_upload: bad first child of node %ld", (long) (order[j] + fn));
            g.child = newidx[s.suns[0] - fn];
        } else
            g.kind = TOP_LOCAL_LEAF;
        SHQ_CHECK(g.sibling != (int32_t) j, SHQ_ERR_INVALID, "toptree_upload: node %ld is its own sibling", (long) (order[j] + fn));
        h[j] = g;
    }
    std::vector<int2> hl((size_t) (ntopleaves > 0 ? ntopleaves : 1));
    for(int k = 0; k < ntopleaves; k++)
        hl[k] = make_int2(topleaves[k].Task, topleaves[k].treenode);
    SHQ_TRY(ctx->topnodes.reserve(nn));
    SHQ_TRY(ctx->topleaves.reserve(hl.size()));
    SHQ_HIP(hipMemcpyAsync(ctx->topnodes.ptr, h.data(), sizeof(TopNodeG) * nn, hipMemcpyHostToDevice, ctx->stream));
    SHQ_HIP(hipMemcpyAsync(ctx->topleaves.ptr, hl.data(), sizeof(int2) * hl.size(), hipMemcpyHostToDevice, ctx->stream));
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    ctx->ntopnodes = (int64_t) nn;
    ctx->have_toptree = true;
    return SHQ_OK;
}

extern "C" int shq_grav_toptree_exports(shq_context *ctx, const shq_grav_params *params, const int32_t *active, int64_t nactive,
                                        int32_t *exportcounts, shq_data_index *table, int64_t capacity, int64_t *nexport)
{
    SHQ_CHECK(ctx && params, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->have_parts && ctx->have_toptree, SHQ_ERR_STATE, "grav_toptree_exports: upload particles and the top tree first");
    SHQ_HIP(hipSetDevice(ctx->device));
    GravTopArgs ga;
    ga.Box = params->BoxSize;
    ga.rcut = params->Rcut;
    ga.rcut2 = params->Rcut * params->Rcut;
    ga.errtol = params->ErrTolForceAcc;
    ga.theta2 = params->BHOpeningAngle2;
    ga.useBH = params->TreeUseBH;
    ga.oldacc = ctx->oldacc.ptr;
    NgbTopArgs na = {};
    return run_toptree<true>(ctx, ga, na, active, nactive, exportcounts, table, capacity, nexport);
}

/* Export detection that stays in HBM (treewalk2.cuh:243-334 keeps its table in managed memory for the same reason): count, scan
 * and fill as above for the targets `active` (NULL = all own particles, the resident handles, or with active_on_device a device
 * list), the table and the scanned counts left resident; back come only the total and, per destination task, how many entries go
 * there (the Send counts of ev_send_recv_export_import, treewalk2.h:618-700).  shq_grav_export_pack then writes the GravTreeQuery
 * records in task order, and the matching `place` list for shq_grav_reduce_export_results, into DEVICE buffers of the caller
 * (the payload of its all-to-all). */
extern "C" int shq_grav_toptree_exports_resident(shq_context *ctx, const shq_grav_params *params, const int32_t *active, int64_t nactive,
                                                 int active_on_device, int ntask, int64_t *nexport, int64_t *task_counts)
{
    SHQ_CHECK(ctx && params && nexport && (task_counts || ntask == 0) && ntask >= 0, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->have_parts && ctx->have_toptree, SHQ_ERR_STATE, "grav_toptree_exports: upload particles and the top tree first");
    SHQ_HIP(hipSetDevice(ctx->device));
    GravTopArgs ga;
    ga.Box = params->BoxSize;
    ga.rcut = params->Rcut;
    ga.rcut2 = params->Rcut * params->Rcut;
    ga.errtol = params->ErrTolForceAcc;
    ga.theta2 = params->BHOpeningAngle2;
    ga.useBH = params->TreeUseBH;
    ga.oldacc = ctx->oldacc.ptr;
    NgbTopArgs na = {};
    SHQ_TRY(run_toptree<true>(ctx, ga, na, active, nactive, nullptr, nullptr, 0, nexport, true, active_on_device != 0));
    for(int t = 0; t < ntask; t++)
        task_counts[t] = 0;
    if(*nexport == 0 || ntask == 0)
        return SHQ_OK;
    SHQ_TRY(ctx->top_task.reserve((size_t) ntask));
    SHQ_HIP(hipMemsetAsync(ctx->top_task.ptr, 0, sizeof(unsigned long long) * ntask, ctx->stream));
    top_task_count_kernel<<<dim3(nblk(*nexport)), dim3(256), 0, ctx->stream>>>(*nexport, ctx->top_table.ptr, ntask, ctx->top_task.ptr);
    SHQ_HIP(hipGetLastError());
    std::vector<unsigned long long> h((size_t) ntask);
    SHQ_HIP(hipMemcpyAsync(h.data(), ctx->top_task.ptr, sizeof(unsigned long long) * ntask, hipMemcpyDeviceToHost, ctx->stream));
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    unsigned long long sum = 0;
    for(int t = 0; t < ntask; t++) {
        task_counts[t] = (int64_t) h[t];
        sum += h[t];
    }
    /* an entry whose Task lies outside [0, ntask) is counted for nobody, yet shq_grav_export_pack packs every entry: the all-to-all's split
     * points would no longer match the packed buffer.  The TopLeaves table and ntask disagree: the caller's error, said out loud. */
    SHQ_CHECK(sum == (unsigned long long) ctx->top_nexport, SHQ_ERR_INVALID,
              "toptree exports: %llu of %lld entries carry a Task outside [0, %d): TopLeaves[].Task and ntask disagree",
              (unsigned long long) ctx->top_nexport - sum, (long long) ctx->top_nexport, ntask);
    return SHQ_OK;
}

namespace {
struct TaskOf {
    __device__ int operator()(const shq_data_index &e) const { return e.Task; }
};
/* slot of entry k = start of its task + number of earlier entries of the same task: per task a flag scan would cost ntask passes;
 * one stable radix sort of (task, k) pairs gives the same order */
__global__ void top_keys_kernel(long long n, const shq_data_index *__restrict__ table, int32_t *key, int32_t *val)
{
    const long long k = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(k < n) {
        key[k] = table[k].Task;
        val[k] = (int32_t) k;
    }
}
__global__ void top_slots_kernel(long long n, const int32_t *__restrict__ sorted_val, long long *slot)
{
    const long long s = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(s < n)
        slot[sorted_val[s]] = s;
}
} // namespace

extern "C" int shq_grav_export_pack(shq_context *ctx, shq_grav_query *d_queries, int32_t *d_place)
{
    SHQ_CHECK(ctx && d_queries && d_place, SHQ_ERR_INVALID, "null argument");
    const long long n = ctx->top_nexport;
    SHQ_CHECK(n >= 0 && ctx->have_toptree, SHQ_ERR_STATE, "grav_export_pack: run shq_grav_toptree_exports_resident first");
    if(n == 0)
        return SHQ_OK;
    SHQ_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    SHQ_TRY(ctx->top_sort.reserve((size_t) (4 * n)));
    SHQ_TRY(ctx->top_slot.reserve((size_t) n));
    int32_t *key = ctx->top_sort.ptr, *val = key + n, *key2 = val + n, *val2 = key2 + n;
    top_keys_kernel<<<dim3(nblk(n)), dim3(256), 0, st>>>(n, ctx->top_table.ptr, key, val);
    size_t tmp = 0;
    SHQ_HIP(rocprim::radix_sort_pairs(nullptr, tmp, key, key2, val, val2, (size_t) n, 0, 32, st));
    SHQ_TRY(ctx->act_temp.reserve(tmp + 16));
    SHQ_HIP(rocprim::radix_sort_pairs((void *) ctx->act_temp.ptr, tmp, key, key2, val, val2, (size_t) n, 0, 32, st)); /* stable */
    top_slots_kernel<<<dim3(nblk(n)), dim3(256), 0, st>>>(n, val2, ctx->top_slot.ptr);
    top_pack_queries_kernel<<<dim3(nblk(n)), dim3(256), 0, st>>>(n, ctx->top_table.ptr, ctx->top_slot.ptr, ctx->posm.ptr, ctx->oldacc.ptr, d_queries, d_place);
    SHQ_HIP(hipGetLastError());
    return SHQ_OK;
}

extern "C" int shq_ngb_toptree_exports(shq_context *ctx, int symmetric, double BoxSize, const int32_t *active, int64_t nactive,
                                       int32_t *exportcounts, shq_data_index *table, int64_t capacity, int64_t *nexport)
{
    SHQ_CHECK(ctx, SHQ_ERR_INVALID, "null context");
    SHQ_CHECK(ctx->have_parts && ctx->have_toptree, SHQ_ERR_STATE, "ngb_toptree_exports: upload particles and the top tree first");
    SHQ_CHECK((ctx->have_sph || ctx->have_dyn) && ctx->hsml.ptr, SHQ_ERR_STATE, "ngb_toptree_exports: no Hsml resident (shq_dynamics_upload / SPH upload)");
    SHQ_CHECK(BoxSize > 0, SHQ_ERR_INVALID, "ngb_toptree_exports: BoxSize must be > 0");
    SHQ_HIP(hipSetDevice(ctx->device));
    GravTopArgs ga = {};
    NgbTopArgs na;
    na.Box = BoxSize;
    na.symmetric = symmetric ? 1 : 0;
    na.hsml = ctx->hsml.ptr;
    return run_toptree<false>(ctx, ga, na, active, nactive, exportcounts, table, capacity, nexport);
}
