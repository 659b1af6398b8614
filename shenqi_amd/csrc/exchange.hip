/* exchange.hip — the pack / unpack loops of the reference's particle exchange between tasks, on opaque records
 * (libgadget/exchange.hpp).
 *
 *   build_exchange_list, build_export_buffer (counts)   exchange.hpp:158-204
 *   exchange_once: pack loop, slots_mark_garbage        exchange.hpp:369-392, slotsmanager.cpp:590-599
 *   exchange_once: PI of the arrivals                   exchange.hpp:483-511
 *
 * The reference packs with one serial loop ("watch out thread unsafe"): its buffer order is the exchange list's order inside
 * every target task, for the base records and for each slot type.  Here the list comes from a stable select, the per-task and
 * per-(task, type) positions from two stable radix sorts of the list (so the orders are the serial loop's), and records move as
 * 16-byte words, a few lanes per record.  The buffers come out byte-identical to the serial loop's. */
#include "common.hpp"
#include <string.h>
#include <vector>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

namespace {

inline unsigned nblk(long long n, int t = 256) { return (unsigned) ((n + t - 1) / t); }

struct Leaving {
    const char *parts;
    size_t elsize, off_flags;
    const int32_t *target;
    int thistask;
    __device__ bool operator()(const int32_t i) const
    {
        const unsigned f = *(const unsigned char *) (parts + (size_t) i * elsize + off_flags);
        const int t = target[i];
        return !(f & 3u) && t != thistask && t >= 0;
    }
};

/* keys of the first `last` list entries: task, and task * 6 + type; bad targets are flagged */
__global__ void ex_keys_kernel(long long last, const int32_t *list, const char *parts, size_t elsize, size_t off_type, const int32_t *target, int ntask,
                               unsigned int *key_task, unsigned int *key_tt, int32_t *val, unsigned long long *counts, int *err)
{
    const long long n = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(n >= last)
        return;
    const int i = list[n];
    const int t = target[i];
    const unsigned type = *(const unsigned char *) (parts + (size_t) i * elsize + off_type);
    if(t >= ntask || t < 0 || type >= 6) {
        *err = 1; /* "layoutfunc for %d returned unreasonable %d", exchange.hpp:196 */
        key_task[n] = 0;
        key_tt[n] = 0;
        val[n] = (int32_t) n;
        return;
    }
    key_task[n] = (unsigned) t;
    key_tt[n] = (unsigned) t * 6u + type;
    val[n] = (int32_t) n;
}

/* toGo from the (task, type)-sorted keys: the length of every run, by two binary searches (no atomics on a few dozen counters) */
__global__ void ex_count_kernel(int nkeys, long long last, const unsigned int *key_tt_sorted, unsigned long long *counts)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= nkeys)
        return;
    long long lo[2];
    for(int w = 0; w < 2; w++) {
        const unsigned want = (unsigned) k + (unsigned) w;
        long long a = 0, e = last;
        while(a < e) {
            const long long mid = a + ((e - a) >> 1);
            if(key_tt_sorted[mid] < want)
                a = mid + 1;
            else
                e = mid;
        }
        lo[w] = a;
    }
    const unsigned long long c = (unsigned long long) (lo[1] - lo[0]);
    counts[(size_t) (k / 6) * 7 + 1 + (k % 6)] = c;
    if(c)
        atomicAdd(&counts[(size_t) (k / 6) * 7], c);
}

/* one record per `lanes` threads, 16 bytes per thread and pass */
__global__ void ex_copy_base_kernel(long long last, const int32_t *order /* list positions sorted by task */, const int32_t *list, const char *parts,
                                    size_t elsize, char *buf)
{
    const long long gid = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    const int words = (int) (elsize / 16);
    const long long rec = gid / words;
    const int w = (int) (gid % words);
    if(rec >= last)
        return;
    const int i = list[order[rec]];
    const uint4 v = *reinterpret_cast<const uint4 *>(parts + (size_t) i * elsize + 16 * (size_t) w);
    *reinterpret_cast<uint4 *>(buf + (size_t) rec * elsize + 16 * (size_t) w) = v;
}

struct SlotTab {
    char *ptr[6];
    char *buf[6];
    size_t elsize[6];
    long long off[6]; /* start of type t's run inside the (task, type)-sorted order, per target task: not needed — see below */
};

/* The (task, type)-sorted order lists, for every task, its type-0 entries, then type-1 ... ; the slot buffer of type ty holds
 * task 0's type-ty entries, then task 1's ...: position in the buffer = toGoOffset[task].slots[ty] + rank inside the (task, ty)
 * run, and the run starts are the exclusive prefix of the counts in (task, type) order. */
__global__ void ex_copy_slot_kernel(long long last, const int32_t *order_tt, const unsigned int *key_tt_sorted, const long long *runstart /* [ntask * 6] */,
                                    const long long *slotoff /* toGoOffset[task].slots[ty], [ntask * 6] */, const int32_t *list, const char *parts,
                                    size_t elsize, size_t off_pi, SlotTab st)
{
    const long long pos = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(pos >= last)
        return;
    const unsigned k = key_tt_sorted[pos];
    const int ty = (int) (k % 6u);
    if(!st.elsize[ty])
        return;
    const int i = list[order_tt[pos]];
    const int pi = *reinterpret_cast<const int32_t *>(parts + (size_t) i * elsize + off_pi);
    const long long dst = slotoff[k] + (pos - runstart[k]);
    const char *src = st.ptr[ty] + (size_t) pi * st.elsize[ty];
    char *out = st.buf[ty] + (size_t) dst * st.elsize[ty];
    for(size_t b = 0; b < st.elsize[ty]; b += 4) /* slot structs are at least 4-byte aligned and sized (bare particle_data_ext) */
        *reinterpret_cast<uint32_t *>(out + b) = *reinterpret_cast<const uint32_t *>(src + b);
}

/* slots_mark_garbage, slotsmanager.cpp:590-599 */
__global__ void ex_mark_kernel(long long last, const int32_t *list, char *parts, size_t elsize, size_t off_flags, size_t off_type, size_t off_pi, SlotTab st,
                               size_t off_rl, int reverselink)
{
    const long long n = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(n >= last)
        return;
    const int i = list[n];
    char *p = parts + (size_t) i * elsize;
    *(unsigned char *) (p + off_flags) |= 1u;
    const unsigned ty = *(const unsigned char *) (p + off_type);
    if(ty < 6 && st.elsize[ty]) {
        const int pi = *reinterpret_cast<const int32_t *>(p + off_pi);
        *reinterpret_cast<int32_t *>(st.ptr[ty] + (size_t) pi * st.elsize[ty] + off_rl) = reverselink;
    }
}

/* PI of the arrivals of one source task: arrival k of type ty gets base[ty] + (number of type-ty arrivals before it) */
__global__ void ex_types_kernel(long long n, const char *parts, size_t elsize, size_t off_type, long long first, unsigned int *key, int32_t *val)
{
    const long long k = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= n)
        return;
    key[k] = *(const unsigned char *) (parts + (size_t) (first + k) * elsize + off_type);
    val[k] = (int32_t) k;
}
__global__ void ex_pi_kernel(long long n, char *parts, size_t elsize, size_t off_pi, long long first, const unsigned int *key_sorted, const int32_t *order,
                             const long long *typestart /* [6] start of each type's run in the sorted order */, const long long *newpi /* [6] */)
{
    const long long pos = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(pos >= n)
        return;
    const unsigned ty = key_sorted[pos];
    if(ty >= 6)
        return;
    const long long k = order[pos];
    *reinterpret_cast<int32_t *>(parts + (size_t) (first + k) * elsize + off_pi) = (int32_t) (newpi[ty] + (pos - typestart[ty]));
}

struct LiveParticle {
    const char *parts;
    size_t elsize, off_flags;
    __device__ bool operator()(const int32_t i) const { return !(*(const unsigned char *) (parts + (size_t) i * elsize + off_flags) & 1u); }
};
struct LiveSlot {
    const char *slots;
    size_t elsize, off_rl;
    int maxpart;
    __device__ bool operator()(const int32_t i) const { return *reinterpret_cast<const int32_t *>(slots + (size_t) i * elsize + off_rl) <= maxpart; }
};

/* records `keep[k]` of src to position k of dst, 4 bytes per thread and pass (slot structs of the bare particle_data_ext are 4 bytes) */
__global__ void gc_gather_kernel(long long nkeep, const int32_t *keep, const char *src, size_t elsize, char *dst)
{
    const long long gid = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    const int words = (int) (elsize / 4);
    const long long rec = gid / words;
    const int w = (int) (gid % words);
    if(rec >= nkeep)
        return;
    *reinterpret_cast<uint32_t *>(dst + (size_t) rec * elsize + 4 * (size_t) w) =
        *reinterpret_cast<const uint32_t *>(src + (size_t) keep[rec] * elsize + 4 * (size_t) w);
}

/* slots_gc_mark, slotsmanager.cpp:243-285 */
__global__ void gc_mark_kernel(long long numpart, const char *parts, size_t elsize, size_t off_flags, size_t off_type, size_t off_pi, SlotTab st,
                               const long long *slot_size, size_t off_rl, int invalid, int *err)
{
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= numpart)
        return;
    const char *p = parts + (size_t) i * elsize;
    const unsigned ty = *(const unsigned char *) (p + off_type);
    if(ty >= 6 || !st.elsize[ty])
        return;
    const int pi = *reinterpret_cast<const int32_t *>(p + off_pi);
    if(pi < 0 || pi >= slot_size[ty]) {
        *err = 1; /* "Particle %ld, type %d has PI index %d beyond max slot size", slotsmanager.cpp:276 */
        return;
    }
    const bool garbage = *(const unsigned char *) (p + off_flags) & 1u;
    *reinterpret_cast<int32_t *>(st.ptr[ty] + (size_t) pi * st.elsize[ty] + off_rl) = garbage ? invalid : (int32_t) i;
}

/* slots_gc_collect, slotsmanager.cpp:301-322 */
__global__ void gc_collect_kernel(long long nslots, const char *slots, size_t selsize, size_t off_rl, char *parts, size_t elsize, size_t off_pi)
{
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= nslots)
        return;
    const int rl = *reinterpret_cast<const int32_t *>(slots + (size_t) i * selsize + off_rl);
    *reinterpret_cast<int32_t *>(parts + (size_t) rl * elsize + off_pi) = (int32_t) i;
}

/* TypeKey of slots_gc_sorted (slotsmanager.cpp:441-446) */
__global__ void gcs_typekey_kernel(long long n, const char *parts, size_t elsize, size_t off_flags, size_t off_type, unsigned int *key, unsigned long long *ngarbage)
{
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n)
        return;
    const char *p = parts + (size_t) i * elsize;
    const bool garbage = *(const unsigned char *) (p + off_flags) & 1u;
    key[i] = garbage ? 255u : (unsigned) *(const unsigned char *) (p + off_type);
    if(garbage)
        atomicAdd(ngarbage, 1ull);
}
__global__ void gcs_gather_u32_kernel(long long n, const int32_t *order, const unsigned int *in, unsigned int *out)
{
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n)
        out[i] = in[order[i]];
}
__global__ void gcs_iota_kernel(long long n, int32_t *v)
{
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n)
        v[i] = (int32_t) i;
}
__global__ void gcs_rlkey_kernel(long long n, const char *slots, size_t elsize, size_t off_rl, unsigned int *key, int32_t *val)
{
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n)
        return;
    key[i] = (unsigned) *reinterpret_cast<const int32_t *>(slots + (size_t) i * elsize + off_rl);
    val[i] = (int32_t) i;
}

template <typename Pred> int select_keep(shq_context *ctx, Pred pred, size_t n, int32_t *out, int64_t *nkeep)
{
    size_t tmp = 0;
    size_t *d_n = reinterpret_cast<size_t *>(ctx->ex_counts.ptr);
    SHQ_HIP(rocprim::select(nullptr, tmp, rocprim::counting_iterator<int32_t>(0), out, d_n, n, pred, ctx->stream));
    SHQ_TRY(ctx->act_temp.reserve(tmp + 16));
    SHQ_HIP(rocprim::select((void *) ctx->act_temp.ptr, tmp, rocprim::counting_iterator<int32_t>(0), out, d_n, n, pred, ctx->stream));
    size_t h = 0;
    SHQ_HIP(hipMemcpyAsync(&h, d_n, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    *nkeep = (int64_t) h;
    return SHQ_OK;
}

int check_layout(const shq_exchange_layout *l)
{
    SHQ_CHECK(l && l->part_elsize >= 16 && l->part_elsize % 16 == 0, SHQ_ERR_INVALID, "exchange: particle records must be a multiple of 16 bytes");
    SHQ_CHECK(l->off_flags < l->part_elsize && l->off_type < l->part_elsize && l->off_pi + 4 <= l->part_elsize && l->off_pi % 4 == 0, SHQ_ERR_INVALID,
              "exchange: field offsets outside the particle record");
    for(int t = 0; t < 6; t++)
        SHQ_CHECK(l->slot_elsize[t] % 4 == 0 && (l->slot_elsize[t] == 0 || l->off_reverselink + 4 <= l->slot_elsize[t]), SHQ_ERR_INVALID,
                  "exchange: slot type %d: record size %zu must be a multiple of 4 and hold ReverseLink", t, l->slot_elsize[t]);
    return SHQ_OK;
}

template <typename K> int sort_pairs(shq_context *ctx, const K *kin, K *kout, const int32_t *vin, int32_t *vout, size_t n, int bits)
{
    size_t tmp = 0;
    SHQ_HIP(rocprim::radix_sort_pairs(nullptr, tmp, kin, kout, vin, vout, n, 0, bits, ctx->stream));
    SHQ_TRY(ctx->act_temp.reserve(tmp + 16));
    SHQ_HIP(rocprim::radix_sort_pairs((void *) ctx->act_temp.ptr, tmp, kin, kout, vin, vout, n, 0, bits, ctx->stream));
    return SHQ_OK;
}

int bits_for(unsigned long long maxkey)
{
    int b = 1;
    while(b < 32 && (1ull << b) <= maxkey)
        b++;
    return b;
}

} // namespace

extern "C" int shq_exchange_plan(shq_context *ctx, const shq_exchange_layout *layout, const void *d_parts, int64_t numpart, const int32_t *d_target,
                                 int ThisTask, int NTask, int64_t maxlast, int64_t *nexchange, int64_t *last_out, shq_exchange_entry *toGo)
{
    SHQ_CHECK(ctx && d_parts && d_target && toGo, SHQ_ERR_INVALID, "null argument");
    SHQ_TRY(check_layout(layout));
    SHQ_CHECK(NTask >= 1 && ThisTask >= 0 && ThisTask < NTask && numpart >= 0 && numpart < (1ll << 31) - 64, SHQ_ERR_INVALID, "exchange_plan: bad task / particle numbers");
    SHQ_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    ctx->ex_last = -1;
    memset(toGo, 0, sizeof(shq_exchange_entry) * (size_t) NTask);
    const size_t cap = (size_t) std::max<int64_t>(numpart, 1);
    SHQ_TRY(ctx->ex_list.reserve(cap));
    SHQ_TRY(ctx->ex_counts.reserve((size_t) NTask * 7 + 8));
    SHQ_HIP(hipMemsetAsync(ctx->ex_counts.ptr, 0, sizeof(unsigned long long) * ((size_t) NTask * 7 + 8), st));
    int64_t nex = 0;
    if(numpart > 0) {
        const Leaving pred{(const char *) d_parts, layout->part_elsize, layout->off_flags, d_target, ThisTask};
        size_t tmp = 0;
        size_t *d_n = reinterpret_cast<size_t *>(ctx->ex_counts.ptr + (size_t) NTask * 7);
        SHQ_HIP(rocprim::select(nullptr, tmp, rocprim::counting_iterator<int32_t>(0), ctx->ex_list.ptr, d_n, (size_t) numpart, pred, st));
        SHQ_TRY(ctx->act_temp.reserve(tmp + 16));
        SHQ_HIP(rocprim::select((void *) ctx->act_temp.ptr, tmp, rocprim::counting_iterator<int32_t>(0), ctx->ex_list.ptr, d_n, (size_t) numpart, pred, st));
        size_t h = 0;
        SHQ_HIP(hipMemcpyAsync(&h, d_n, sizeof(h), hipMemcpyDeviceToHost, st));
        SHQ_HIP(hipStreamSynchronize(st));
        nex = (int64_t) h;
    }
    const int64_t last = (maxlast > 0 && maxlast < nex) ? maxlast : nex;
    if(nexchange)
        *nexchange = nex;
    if(last_out)
        *last_out = last;
    const size_t lcap = (size_t) std::max<int64_t>(last, 1);
    SHQ_TRY(ctx->ex_key[0].reserve(lcap));
    SHQ_TRY(ctx->ex_key[1].reserve(lcap));
    SHQ_TRY(ctx->ex_key[2].reserve(lcap));
    SHQ_TRY(ctx->ex_key[3].reserve(lcap));
    SHQ_TRY(ctx->ex_val[0].reserve(lcap));
    SHQ_TRY(ctx->ex_val[1].reserve(lcap));
    SHQ_TRY(ctx->ex_val[2].reserve(lcap));
    if(last > 0) {
        int *d_err = reinterpret_cast<int *>(ctx->ex_counts.ptr + (size_t) NTask * 7 + 4);
        ex_keys_kernel<<<dim3(nblk(last)), dim3(256), 0, st>>>(last, ctx->ex_list.ptr, (const char *) d_parts, layout->part_elsize, layout->off_type, d_target, NTask,
                                                               ctx->ex_key[0].ptr, ctx->ex_key[1].ptr, ctx->ex_val[0].ptr, ctx->ex_counts.ptr, d_err);
        SHQ_HIP(hipGetLastError());
        /* order by task (base records) and by (task, type) (slot records); stable: list order inside every run */
        SHQ_TRY(sort_pairs(ctx, (const unsigned int *) ctx->ex_key[0].ptr, ctx->ex_key[2].ptr, (const int32_t *) ctx->ex_val[0].ptr, ctx->ex_val[1].ptr, (size_t) last,
                           bits_for((unsigned long long) NTask)));
        SHQ_TRY(sort_pairs(ctx, (const unsigned int *) ctx->ex_key[1].ptr, ctx->ex_key[3].ptr, (const int32_t *) ctx->ex_val[0].ptr, ctx->ex_val[2].ptr, (size_t) last,
                           bits_for((unsigned long long) NTask * 6)));
        ex_count_kernel<<<dim3(nblk((long long) NTask * 6)), dim3(256), 0, st>>>(NTask * 6, last, ctx->ex_key[3].ptr, ctx->ex_counts.ptr);
        SHQ_HIP(hipGetLastError());
        std::vector<unsigned long long> h((size_t) NTask * 7 + 8);
        SHQ_HIP(hipMemcpyAsync(h.data(), ctx->ex_counts.ptr, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost, st));
        SHQ_HIP(hipStreamSynchronize(st));
        int h_err = 0;
        memcpy(&h_err, &h[(size_t) NTask * 7 + 4], sizeof(int));
        SHQ_CHECK(h_err == 0, SHQ_ERR_INVALID, "exchange_plan: a target task outside [0, %d) or a particle type > 5 (exchange.hpp:196)", NTask);
        for(int t = 0; t < NTask; t++) {
            toGo[t].base = (int64_t) h[(size_t) t * 7];
            for(int ty = 0; ty < 6; ty++)
                toGo[t].slots[ty] = (int64_t) h[(size_t) t * 7 + 1 + ty];
        }
    }
    ctx->ex_last = last;
    ctx->ex_ntask = NTask;
    ctx->ex_togo.assign(toGo, toGo + NTask);
    return SHQ_OK;
}

extern "C" int shq_exchange_pack(shq_context *ctx, const shq_exchange_layout *layout, void *d_parts, void *const d_slots[6], int64_t MaxPart,
                                 const shq_exchange_entry *toGoOffset, int NTask, void *d_partbuf, void *const d_slotbuf[6])
{
    SHQ_CHECK(ctx && d_parts && toGoOffset, SHQ_ERR_INVALID, "null argument");
    SHQ_TRY(check_layout(layout));
    SHQ_CHECK(ctx->ex_last >= 0 && ctx->ex_ntask == NTask, SHQ_ERR_STATE, "exchange_pack: call shq_exchange_plan first");
    SHQ_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const long long last = ctx->ex_last;
    if(last == 0)
        return SHQ_OK;
    SHQ_CHECK(d_partbuf, SHQ_ERR_INVALID, "exchange_pack: no particle buffer");
    SlotTab tab;
    memset(&tab, 0, sizeof(tab));
    for(int ty = 0; ty < 6; ty++) {
        tab.elsize[ty] = layout->slot_elsize[ty];
        if(tab.elsize[ty]) {
            int64_t need = 0;
            for(int t = 0; t < NTask; t++)
                need += ctx->ex_togo[t].slots[ty];
            SHQ_CHECK(need == 0 || (d_slots && d_slots[ty] && d_slotbuf && d_slotbuf[ty]), SHQ_ERR_INVALID, "exchange_pack: slot type %d enabled but no arrays", ty);
            tab.ptr[ty] = d_slots ? (char *) d_slots[ty] : nullptr;
            tab.buf[ty] = d_slotbuf ? (char *) d_slotbuf[ty] : nullptr;
        }
    }
    /* the base buffer is dense in task order, so toGoOffset[t].base must be the prefix of toGo (build_export_buffer computes exactly that) */
    std::vector<long long> runstart((size_t) NTask * 6), slotoff((size_t) NTask * 6);
    long long acc = 0, basepref = 0;
    for(int t = 0; t < NTask; t++) {
        SHQ_CHECK(toGoOffset[t].base == basepref, SHQ_ERR_INVALID, "exchange_pack: toGoOffset[%d].base is not the prefix sum of toGo", t);
        basepref += ctx->ex_togo[t].base;
        for(int ty = 0; ty < 6; ty++) {
            runstart[(size_t) t * 6 + ty] = acc;
            acc += ctx->ex_togo[t].slots[ty];
            slotoff[(size_t) t * 6 + ty] = toGoOffset[t].slots[ty];
        }
    }
    SHQ_TRY(ctx->ex_i64.reserve((size_t) NTask * 12 + 16));
    SHQ_HIP(hipMemcpyAsync(ctx->ex_i64.ptr, runstart.data(), sizeof(long long) * runstart.size(), hipMemcpyHostToDevice, st));
    SHQ_HIP(hipMemcpyAsync(ctx->ex_i64.ptr + (size_t) NTask * 6, slotoff.data(), sizeof(long long) * slotoff.size(), hipMemcpyHostToDevice, st));
    const long long nthreads = last * (long long) (layout->part_elsize / 16);
    ex_copy_base_kernel<<<dim3(nblk(nthreads)), dim3(256), 0, st>>>(last, ctx->ex_val[1].ptr, ctx->ex_list.ptr, (const char *) d_parts, layout->part_elsize,
                                                                   (char *) d_partbuf);
    SHQ_HIP(hipGetLastError());
    ex_copy_slot_kernel<<<dim3(nblk(last)), dim3(256), 0, st>>>(last, ctx->ex_val[2].ptr, ctx->ex_key[3].ptr, ctx->ex_i64.ptr, ctx->ex_i64.ptr + (size_t) NTask * 6,
                                                               ctx->ex_list.ptr, (const char *) d_parts, layout->part_elsize, layout->off_pi, tab);
    SHQ_HIP(hipGetLastError()); /* a failed copy launch must not be followed by the marks that destroy its source */
    /* the copies read what the marks overwrite: same stream, in order */
    ex_mark_kernel<<<dim3(nblk(last)), dim3(256), 0, st>>>(last, ctx->ex_list.ptr, (char *) d_parts, layout->part_elsize, layout->off_flags, layout->off_type,
                                                          layout->off_pi, tab, layout->off_reverselink, (int) (MaxPart + 100));
    SHQ_HIP(hipGetLastError());
    SHQ_HIP(hipStreamSynchronize(st)); /* runstart / slotoff live on this frame */
    ctx->ex_last = -1;
    return SHQ_OK;
}

extern "C" int shq_exchange_unpack(shq_context *ctx, const shq_exchange_layout *layout, void *d_parts, int64_t numpart_old, const int64_t slot_size_old[6],
                                   const shq_exchange_entry *toGet, const shq_exchange_entry *toGetOffset, int NTask)
{
    SHQ_CHECK(ctx && d_parts && slot_size_old && toGet && toGetOffset, SHQ_ERR_INVALID, "null argument");
    SHQ_TRY(check_layout(layout));
    SHQ_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    int64_t maxn = 0;
    for(int s = 0; s < NTask; s++)
        maxn = std::max(maxn, toGet[s].base);
    if(maxn == 0)
        return SHQ_OK;
    SHQ_TRY(ctx->ex_key[0].reserve((size_t) maxn));
    SHQ_TRY(ctx->ex_key[2].reserve((size_t) maxn));
    SHQ_TRY(ctx->ex_val[0].reserve((size_t) maxn));
    SHQ_TRY(ctx->ex_val[1].reserve((size_t) maxn));
    SHQ_TRY(ctx->ex_i64.reserve((size_t) NTask * 12 + 16));
    for(int src = 0; src < NTask; src++) {
        const long long n = toGet[src].base;
        if(n == 0)
            continue;
        const long long first = numpart_old + toGetOffset[src].base;
        long long h[12], acc = 0;
        for(int ty = 0; ty < 6; ty++) {
            h[ty] = acc; /* start of the type's run in the type-sorted arrivals */
            acc += toGet[src].slots[ty];
            h[6 + ty] = slot_size_old[ty] + toGetOffset[src].slots[ty];
        }
        SHQ_CHECK(acc == n, SHQ_ERR_INVALID, "exchange_unpack: toGet[%d].slots do not add up to base (N_slots mismatched, exchange.hpp:505-509)", src);
        SHQ_HIP(hipMemcpyAsync(ctx->ex_i64.ptr, h, sizeof(h), hipMemcpyHostToDevice, st));
        ex_types_kernel<<<dim3(nblk(n)), dim3(256), 0, st>>>(n, (const char *) d_parts, layout->part_elsize, layout->off_type, first, ctx->ex_key[0].ptr, ctx->ex_val[0].ptr);
        SHQ_TRY(sort_pairs(ctx, (const unsigned int *) ctx->ex_key[0].ptr, ctx->ex_key[2].ptr, (const int32_t *) ctx->ex_val[0].ptr, ctx->ex_val[1].ptr, (size_t) n, 8));
        ex_pi_kernel<<<dim3(nblk(n)), dim3(256), 0, st>>>(n, (char *) d_parts, layout->part_elsize, layout->off_pi, first, ctx->ex_key[2].ptr, ctx->ex_val[1].ptr,
                                                          ctx->ex_i64.ptr, ctx->ex_i64.ptr + 6);
        SHQ_HIP(hipGetLastError());
        SHQ_HIP(hipStreamSynchronize(st)); /* h lives on this frame */
    }
    return SHQ_OK;
}

extern "C" int shq_slots_gc(shq_context *ctx, const shq_exchange_layout *layout, void *d_parts, int64_t *numpart, int64_t MaxPart, void *const d_slots[6],
                            int64_t slot_size[6], const int compact[6])
{
    SHQ_CHECK(ctx && d_parts && numpart && slot_size && compact, SHQ_ERR_INVALID, "null argument");
    SHQ_TRY(check_layout(layout));
    SHQ_CHECK(*numpart >= 0 && *numpart <= MaxPart && MaxPart < (1ll << 31) - 200, SHQ_ERR_INVALID, "slots_gc: bad particle numbers");
    SHQ_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    SHQ_TRY(ctx->ex_counts.reserve(64));
    const size_t esz = layout->part_elsize;
    /* slots_gc_base: squeeze the garbage out of the particle array, order kept */
    int64_t n = *numpart;
    if(n > 0) {
        SHQ_TRY(ctx->ex_list.reserve((size_t) n));
        int64_t nkeep = 0;
        SHQ_TRY(select_keep(ctx, LiveParticle{(const char *) d_parts, esz, layout->off_flags}, (size_t) n, ctx->ex_list.ptr, &nkeep));
        if(nkeep < n) {
            SHQ_TRY(ctx->ex_bytes.reserve((size_t) std::max<int64_t>(nkeep, 1) * esz));
            if(nkeep > 0) {
                gc_gather_kernel<<<dim3(nblk(nkeep * (long long) (esz / 4))), dim3(256), 0, st>>>(nkeep, ctx->ex_list.ptr, (const char *) d_parts, esz, ctx->ex_bytes.ptr);
                SHQ_HIP(hipGetLastError()); /* before the copy back: a failed gather would hand d_parts stale bytes */
                SHQ_HIP(hipMemcpyAsync(d_parts, ctx->ex_bytes.ptr, (size_t) nkeep * esz, hipMemcpyDeviceToDevice, st));
            }
            n = nkeep;
        }
    }
    /* the caller's counts are published only after everything queued here has completed */
    int64_t new_slot_size[6];
    for(int ty = 0; ty < 6; ty++)
        new_slot_size[ty] = slot_size[ty];
    bool any = false;
    SlotTab tab;
    memset(&tab, 0, sizeof(tab));
    for(int ty = 0; ty < 6; ty++) {
        any = any || compact[ty];
        tab.elsize[ty] = layout->slot_elsize[ty];
        tab.ptr[ty] = (tab.elsize[ty] && d_slots) ? (char *) d_slots[ty] : nullptr;
        SHQ_CHECK(!tab.elsize[ty] || slot_size[ty] == 0 || tab.ptr[ty], SHQ_ERR_INVALID, "slots_gc: slot type %d enabled but no array", ty);
    }
    if(!any) {
        SHQ_HIP(hipStreamSynchronize(st));
        *numpart = n;
        return SHQ_OK;
    }
    /* slots_gc_mark */
    SHQ_TRY(ctx->ex_i64.reserve(16));
    long long h_sz[6];
    for(int ty = 0; ty < 6; ty++)
        h_sz[ty] = slot_size[ty];
    SHQ_HIP(hipMemcpyAsync(ctx->ex_i64.ptr, h_sz, sizeof(h_sz), hipMemcpyHostToDevice, st));
    int *d_err = reinterpret_cast<int *>(ctx->ex_counts.ptr + 8);
    SHQ_HIP(hipMemsetAsync(d_err, 0, sizeof(int), st));
    if(n > 0) {
        gc_mark_kernel<<<dim3(nblk(n)), dim3(256), 0, st>>>(n, (const char *) d_parts, esz, layout->off_flags, layout->off_type, layout->off_pi, tab, ctx->ex_i64.ptr,
                                                            layout->off_reverselink, (int) (MaxPart + 100), d_err);
        SHQ_HIP(hipGetLastError());
    }
    int h_err = 0;
    SHQ_HIP(hipMemcpyAsync(&h_err, d_err, sizeof(int), hipMemcpyDeviceToHost, st));
    SHQ_HIP(hipStreamSynchronize(st));
    if(h_err != 0)
        *numpart = n; /* the particle array has been compacted and that has completed: its count is the new one even on this error */
    SHQ_CHECK(h_err == 0, SHQ_ERR_INVALID, "slots_gc: a particle's PI lies outside its slot array (slotsmanager.cpp:276)");
    for(int ty = 0; ty < 6; ty++) {
        if(!compact[ty] || !tab.elsize[ty] || slot_size[ty] == 0)
            continue;
        /* slots_gc_sweep: slots nobody points to go; slots_gc_collect: PI follows */
        const int64_t used = slot_size[ty];
        SHQ_TRY(ctx->ex_list.reserve((size_t) used));
        int64_t nkeep = 0;
        SHQ_TRY(select_keep(ctx, LiveSlot{tab.ptr[ty], tab.elsize[ty], layout->off_reverselink, (int) MaxPart}, (size_t) used, ctx->ex_list.ptr, &nkeep));
        if(nkeep < used && nkeep > 0) {
            SHQ_TRY(ctx->ex_bytes.reserve((size_t) nkeep * tab.elsize[ty]));
            gc_gather_kernel<<<dim3(nblk(nkeep * (long long) (tab.elsize[ty] / 4))), dim3(256), 0, st>>>(nkeep, ctx->ex_list.ptr, tab.ptr[ty], tab.elsize[ty], ctx->ex_bytes.ptr);
            SHQ_HIP(hipGetLastError());
            SHQ_HIP(hipMemcpyAsync(tab.ptr[ty], ctx->ex_bytes.ptr, (size_t) nkeep * tab.elsize[ty], hipMemcpyDeviceToDevice, st));
        }
        new_slot_size[ty] = nkeep;
        if(nkeep > 0) {
            gc_collect_kernel<<<dim3(nblk(nkeep)), dim3(256), 0, st>>>(nkeep, tab.ptr[ty], tab.elsize[ty], layout->off_reverselink, (char *) d_parts, esz, layout->off_pi);
            SHQ_HIP(hipGetLastError());
        }
    }
    SHQ_HIP(hipStreamSynchronize(st));
    *numpart = n;
    for(int ty = 0; ty < 6; ty++)
        slot_size[ty] = new_slot_size[ty];
    return SHQ_OK;
}

extern "C" int shq_slots_gc_sorted(shq_context *ctx, const shq_exchange_layout *layout, void *d_parts, int64_t *numpart, int64_t MaxPart, void *const d_slots[6],
                                   int64_t slot_size[6], const uint64_t *d_keys)
{
    SHQ_CHECK(ctx && d_parts && numpart && slot_size && d_keys, SHQ_ERR_INVALID, "null argument");
    SHQ_TRY(check_layout(layout));
    SHQ_CHECK(*numpart >= 0 && *numpart <= MaxPart && MaxPart < (1ll << 31) - 200, SHQ_ERR_INVALID, "slots_gc_sorted: bad particle numbers");
    SHQ_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t esz = layout->part_elsize;
    const int64_t n = *numpart;
    SHQ_TRY(ctx->ex_counts.reserve(64));
    SlotTab tab;
    memset(&tab, 0, sizeof(tab));
    for(int ty = 0; ty < 6; ty++) {
        tab.elsize[ty] = layout->slot_elsize[ty];
        tab.ptr[ty] = (tab.elsize[ty] && d_slots) ? (char *) d_slots[ty] : nullptr;
        SHQ_CHECK(!tab.elsize[ty] || slot_size[ty] == 0 || tab.ptr[ty], SHQ_ERR_INVALID, "slots_gc_sorted: slot type %d enabled but no array", ty);
    }
    int64_t ngarbage = 0;
    if(n > 0) {
        const size_t cap = (size_t) n;
        SHQ_TRY(ctx->ex_key[0].reserve(cap));
        SHQ_TRY(ctx->ex_key[1].reserve(cap));
        SHQ_TRY(ctx->ex_key[2].reserve(cap));
        SHQ_TRY(ctx->ex_val[0].reserve(cap));
        SHQ_TRY(ctx->ex_val[1].reserve(cap));
        SHQ_TRY(ctx->ex_val[2].reserve(cap));
        SHQ_TRY(ctx->ex_u64.reserve(cap));
        SHQ_TRY(ctx->ex_bytes.reserve(cap * esz));
        unsigned long long *d_ng = ctx->ex_counts.ptr + 16;
        SHQ_HIP(hipMemsetAsync(d_ng, 0, sizeof(unsigned long long), st));
        /* PeanoOrder::operator<: TypeKey first, then Key.  Least significant first: a stable sort by key, then by TypeKey */
        gcs_iota_kernel<<<dim3(nblk(n)), dim3(256), 0, st>>>(n, ctx->ex_val[0].ptr);
        {
            size_t tmp = 0;
            SHQ_HIP(rocprim::radix_sort_pairs(nullptr, tmp, (const unsigned long long *) d_keys, ctx->ex_u64.ptr, (const int32_t *) ctx->ex_val[0].ptr, ctx->ex_val[1].ptr, cap, 0, 64, st));
            SHQ_TRY(ctx->act_temp.reserve(tmp + 16));
            SHQ_HIP(rocprim::radix_sort_pairs((void *) ctx->act_temp.ptr, tmp, (const unsigned long long *) d_keys, ctx->ex_u64.ptr, (const int32_t *) ctx->ex_val[0].ptr,
                                              ctx->ex_val[1].ptr, cap, 0, 64, st));
        }
        gcs_typekey_kernel<<<dim3(nblk(n)), dim3(256), 0, st>>>(n, (const char *) d_parts, esz, layout->off_flags, layout->off_type, ctx->ex_key[0].ptr, d_ng);
        gcs_gather_u32_kernel<<<dim3(nblk(n)), dim3(256), 0, st>>>(n, ctx->ex_val[1].ptr, ctx->ex_key[0].ptr, ctx->ex_key[1].ptr);
        SHQ_TRY(sort_pairs(ctx, (const unsigned int *) ctx->ex_key[1].ptr, ctx->ex_key[2].ptr, (const int32_t *) ctx->ex_val[1].ptr, ctx->ex_val[2].ptr, cap, 8));
        gc_gather_kernel<<<dim3(nblk(n * (long long) (esz / 4))), dim3(256), 0, st>>>(n, ctx->ex_val[2].ptr, (const char *) d_parts, esz, ctx->ex_bytes.ptr);
        SHQ_HIP(hipMemcpyAsync(d_parts, ctx->ex_bytes.ptr, cap * esz, hipMemcpyDeviceToDevice, st));
        unsigned long long h = 0;
        SHQ_HIP(hipMemcpyAsync(&h, d_ng, sizeof(h), hipMemcpyDeviceToHost, st));
        /* slots_gc_mark over the sorted array, garbage included (:487) */
        SHQ_TRY(ctx->ex_i64.reserve(16));
        long long h_sz[6];
        for(int ty = 0; ty < 6; ty++)
            h_sz[ty] = slot_size[ty];
        SHQ_HIP(hipMemcpyAsync(ctx->ex_i64.ptr, h_sz, sizeof(h_sz), hipMemcpyHostToDevice, st));
        int *d_err = reinterpret_cast<int *>(ctx->ex_counts.ptr + 8);
        SHQ_HIP(hipMemsetAsync(d_err, 0, sizeof(int), st));
        gc_mark_kernel<<<dim3(nblk(n)), dim3(256), 0, st>>>(n, (const char *) d_parts, esz, layout->off_flags, layout->off_type, layout->off_pi, tab, ctx->ex_i64.ptr,
                                                            layout->off_reverselink, (int) (MaxPart + 100), d_err);
        int h_err = 0;
        SHQ_HIP(hipMemcpyAsync(&h_err, d_err, sizeof(int), hipMemcpyDeviceToHost, st));
        SHQ_HIP(hipStreamSynchronize(st));
        SHQ_CHECK(h_err == 0, SHQ_ERR_INVALID, "slots_gc_sorted: a particle's PI lies outside its slot array (slotsmanager.cpp:276)");
        ngarbage = (int64_t) h;
    }
    *numpart = n - ngarbage;
    for(int ty = 0; ty < 6; ty++) {
        if(!tab.elsize[ty] || slot_size[ty] == 0)
            continue;
        /* slots sorted by ReverseLink (operator< of particle_data_ext): garbage (MaxPart + 100) last; then the first garbage is the size */
        const int64_t used = slot_size[ty];
        SHQ_TRY(ctx->ex_key[0].reserve((size_t) used));
        SHQ_TRY(ctx->ex_key[1].reserve((size_t) used));
        SHQ_TRY(ctx->ex_val[0].reserve((size_t) used));
        SHQ_TRY(ctx->ex_val[1].reserve((size_t) used));
        SHQ_TRY(ctx->ex_bytes.reserve((size_t) used * tab.elsize[ty]));
        gcs_rlkey_kernel<<<dim3(nblk(used)), dim3(256), 0, st>>>(used, tab.ptr[ty], tab.elsize[ty], layout->off_reverselink, ctx->ex_key[0].ptr, ctx->ex_val[0].ptr);
        SHQ_TRY(sort_pairs(ctx, (const unsigned int *) ctx->ex_key[0].ptr, ctx->ex_key[1].ptr, (const int32_t *) ctx->ex_val[0].ptr, ctx->ex_val[1].ptr, (size_t) used, 32));
        gc_gather_kernel<<<dim3(nblk(used * (long long) (tab.elsize[ty] / 4))), dim3(256), 0, st>>>(used, ctx->ex_val[1].ptr, tab.ptr[ty], tab.elsize[ty], ctx->ex_bytes.ptr);
        SHQ_HIP(hipMemcpyAsync(tab.ptr[ty], ctx->ex_bytes.ptr, (size_t) used * tab.elsize[ty], hipMemcpyDeviceToDevice, st));
        SHQ_TRY(ctx->ex_list.reserve((size_t) used));
        int64_t nkeep = 0;
        SHQ_TRY(select_keep(ctx, LiveSlot{tab.ptr[ty], tab.elsize[ty], layout->off_reverselink, (int) MaxPart}, (size_t) used, ctx->ex_list.ptr, &nkeep));
        slot_size[ty] = nkeep;
        if(nkeep > 0) {
            gc_collect_kernel<<<dim3(nblk(nkeep)), dim3(256), 0, st>>>(nkeep, tab.ptr[ty], tab.elsize[ty], layout->off_reverselink, (char *) d_parts, esz, layout->off_pi);
            SHQ_HIP(hipGetLastError());
        }
    }
    SHQ_HIP(hipStreamSynchronize(st));
    return SHQ_OK;
}

/* ---- slots_split_particle / slots_convert for lists (slotsmanager.cpp:27-126) ----------------------------------------------------
 * What star formation (sfr_eff.cpp:344-372: NewStars / NewParents, placement = firststarslot + i), black-hole seeding
 * (blackhole.cpp:1040) and the wind spawns do one particle at a time.  The reference hands out the new particle indices and slot
 * indices with atomic counters, so their order is the threads'; here entry k of the list gets NumPart + k / size + k. */
namespace {

/* the parent's half of slots_split_particle: Generation ++ (a 4-bit field of the flag byte), Mass -= childmass */
__global__ void split_parent_kernel(long long n, const int32_t *parents, const double *childmass, char *parts, size_t elsize, size_t off_flags, int gen_shift,
                                    size_t off_mass, long long numpart, int *err)
{
    const long long k = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= n)
        return;
    const long long p = parents[k];
    if(p < 0 || p >= numpart) {
        *err = 1;
        return;
    }
    char *rec = parts + (size_t) p * elsize;
    unsigned char *fl = (unsigned char *) (rec + off_flags);
    const unsigned f = *fl;
    const unsigned g = ((f >> gen_shift) + 1u) & 15u;
    *fl = (unsigned char) ((f & ~(15u << gen_shift)) | (g << gen_shift));
    float *m = (float *) (rec + off_mass);
    *m = (float) ((double) *m - childmass[k]);
}

/* Base[child] = Base[parent], 16 bytes per thread */
__global__ void split_copy_kernel(long long n, const int32_t *parents, char *parts, size_t elsize, long long numpart)
{
    const long long gid = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    const int words = (int) (elsize / 16);
    const long long k = gid / words;
    const int w = (int) (gid % words);
    if(k >= n)
        return;
    const long long p = parents[k];
    if(p < 0 || p >= numpart)
        return;
    const uint4 *src = (const uint4 *) (parts + (size_t) p * elsize);
    uint4 *dst = (uint4 *) (parts + (size_t) (numpart + k) * elsize);
    dst[w] = src[w];
}

/* the child's half: ID carries the generation in its highest 8 bits, Mass = childmass, PI = -1 */
__global__ void split_child_kernel(long long n, const double *childmass, char *parts, size_t elsize, size_t off_flags, int gen_shift, size_t off_id, size_t off_mass,
                                   size_t off_pi, long long numpart, int32_t *children)
{
    const long long k = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= n)
        return;
    char *rec = parts + (size_t) (numpart + k) * elsize;
    const unsigned long long g = (*(const unsigned char *) (rec + off_flags) >> gen_shift) & 15u;
    unsigned long long *id = (unsigned long long *) (rec + off_id);
    *id = (*id & 0x00ffffffffffffffull) + (g << 56);
    *(float *) (rec + off_mass) = (float) childmass[k];
    *(int32_t *) (rec + off_pi) = -1;
    if(children)
        children[k] = (int32_t) (numpart + k);
}

/* slots_convert of entry k with placement first + k: the old slot becomes garbage, the new one is poisoned with 'e' (101) bytes
 * (slots_connect_new_slot), PI and Type follow.  A few threads per entry: thread w of an entry fills words w, w + T, ... */
constexpr int CONV_T = 8;
__global__ void convert_kernel(long long n, const int32_t *index, char *parts, size_t elsize, size_t off_type, size_t off_pi, SlotTab tab, size_t off_rl, int invalid,
                               int ptype, long long first, long long numpart, int *err)
{
    const long long gid = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    const long long k = gid / CONV_T;
    const int w = (int) (gid % CONV_T);
    if(k >= n)
        return;
    const long long p = index[k];
    if(p < 0 || p >= numpart) {
        *err = 1;
        return;
    }
    char *rec = parts + (size_t) p * elsize;
    const unsigned oldtype = *(const unsigned char *) (rec + off_type);
    const int oldpi = *(const int32_t *) (rec + off_pi);
    if(oldtype >= 6) {
        *err = 2;
        return;
    }
    const bool newslot = tab.elsize[ptype] != 0;
    if(newslot) {
        unsigned int *dst = (unsigned int *) (tab.ptr[ptype] + (size_t) (first + k) * tab.elsize[ptype]);
        const int words = (int) (tab.elsize[ptype] / 4);
        for(int j = w; j < words; j += CONV_T)
            dst[j] = 0x65656565u;
    }
    if(w == 0) {
        /* the old slot is another array's record unless the type stays: then it may be the array being written, at another index
         * (oldpi < first), which no other entry touches */
        if(oldpi >= 0 && tab.elsize[oldtype])
            *(int32_t *) (tab.ptr[oldtype] + (size_t) oldpi * tab.elsize[oldtype] + off_rl) = invalid;
    }
}

/* PI and Type after every slot has been written (an entry's old slot can never be another entry's new one: new ones lie past size) */
__global__ void convert_link_kernel(long long n, const int32_t *index, char *parts, size_t elsize, size_t off_type, size_t off_pi, bool newslot, int ptype, long long first,
                                    long long numpart)
{
    const long long k = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= n)
        return;
    const long long p = index[k];
    if(p < 0 || p >= numpart)
        return;
    char *rec = parts + (size_t) p * elsize;
    if(newslot)
        *(int32_t *) (rec + off_pi) = (int32_t) (first + k);
    *(unsigned char *) (rec + off_type) = (unsigned char) ptype;
}

} // namespace

extern "C" int shq_slots_split_particles(shq_context *ctx, const shq_exchange_layout *layout, const shq_spawn_layout *spawn, void *d_parts, int64_t *numpart,
                                         int64_t MaxPart, const int32_t *d_parents, const double *d_childmass, int64_t n, int32_t *d_children)
{
    SHQ_CHECK(ctx && spawn && d_parts && numpart, SHQ_ERR_INVALID, "null argument");
    SHQ_TRY(check_layout(layout));
    SHQ_CHECK(n >= 0 && (n == 0 || (d_parents && d_childmass)), SHQ_ERR_INVALID, "slots_split_particles: bad list");
    SHQ_CHECK(*numpart >= 0 && *numpart <= MaxPart && MaxPart < (1ll << 31) - 200, SHQ_ERR_INVALID, "slots_split_particles: bad particle numbers");
    const size_t esz = layout->part_elsize;
    SHQ_CHECK(spawn->generation_shift >= 0 && spawn->generation_shift <= 4 && spawn->off_id + 8 <= esz && spawn->off_id % 8 == 0 && spawn->off_mass + 4 <= esz &&
                  spawn->off_mass % 4 == 0,
              SHQ_ERR_INVALID, "slots_split_particles: bad spawn layout");
    /* "Tried to spawn: NumPart=%ld MaxPart = %ld. Sorry, no space left." (slotsmanager.cpp:107-108), before anything is touched */
    SHQ_CHECK(*numpart + n <= MaxPart, SHQ_ERR_NOMEM, "slots_split_particles: NumPart = %ld + %ld spawned > MaxPart = %ld: no space left", (long) *numpart, (long) n,
              (long) MaxPart);
    if(n == 0)
        return SHQ_OK;
    SHQ_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    SHQ_TRY(ctx->ex_counts.reserve(64));
    int *d_err = reinterpret_cast<int *>(ctx->ex_counts.ptr + 8);
    SHQ_HIP(hipMemsetAsync(d_err, 0, sizeof(int), st));
    const long long np = *numpart;
    split_parent_kernel<<<dim3(nblk(n)), dim3(256), 0, st>>>(n, d_parents, d_childmass, (char *) d_parts, esz, layout->off_flags, spawn->generation_shift,
                                                             spawn->off_mass, np, d_err);
    SHQ_HIP(hipGetLastError());
    split_copy_kernel<<<dim3(nblk(n * (long long) (esz / 16))), dim3(256), 0, st>>>(n, d_parents, (char *) d_parts, esz, np);
    SHQ_HIP(hipGetLastError());
    split_child_kernel<<<dim3(nblk(n)), dim3(256), 0, st>>>(n, d_childmass, (char *) d_parts, esz, layout->off_flags, spawn->generation_shift, spawn->off_id,
                                                            spawn->off_mass, layout->off_pi, np, d_children);
    SHQ_HIP(hipGetLastError());
    int h_err = 0;
    SHQ_HIP(hipMemcpyAsync(&h_err, d_err, sizeof(int), hipMemcpyDeviceToHost, st));
    SHQ_HIP(hipStreamSynchronize(st));
    SHQ_CHECK(h_err == 0, SHQ_ERR_INVALID, "slots_split_particles: a parent index lies outside [0, NumPart)");
    *numpart = np + n;
    return SHQ_OK;
}

extern "C" int shq_slots_convert(shq_context *ctx, const shq_exchange_layout *layout, void *d_parts, int64_t numpart, int64_t MaxPart, void *const d_slots[6],
                                 int64_t slot_size[6], const int64_t slot_maxsize[6], const int32_t *d_index, int64_t n, int ptype)
{
    SHQ_CHECK(ctx && d_parts && slot_size && slot_maxsize, SHQ_ERR_INVALID, "null argument");
    SHQ_TRY(check_layout(layout));
    SHQ_CHECK(ptype >= 0 && ptype < 6, SHQ_ERR_INVALID, "slots_convert: type %d", ptype);
    SHQ_CHECK(n >= 0 && (n == 0 || d_index), SHQ_ERR_INVALID, "slots_convert: bad list");
    SHQ_CHECK(numpart >= 0 && numpart <= MaxPart && MaxPart < (1ll << 31) - 200, SHQ_ERR_INVALID, "slots_convert: bad particle numbers");
    SlotTab tab;
    memset(&tab, 0, sizeof(tab));
    for(int ty = 0; ty < 6; ty++) {
        tab.elsize[ty] = layout->slot_elsize[ty];
        tab.ptr[ty] = (tab.elsize[ty] && d_slots) ? (char *) d_slots[ty] : nullptr;
        SHQ_CHECK(!tab.elsize[ty] || tab.ptr[ty] || (slot_size[ty] == 0 && (ty != ptype || n == 0)), SHQ_ERR_INVALID, "slots_convert: slot type %d enabled but no array",
                  ty);
    }
    const bool newslot = tab.elsize[ptype] != 0;
    const long long first = slot_size[ptype];
    /* "Tried to use non-allocated slot %d (> %ld)" (slotsmanager.cpp:76-78): growing the arrays is the caller's (sfr_reserve_slots,
     * fof_seed's slots_reserve), and has to happen before the call */
    SHQ_CHECK(!newslot || first + n <= slot_maxsize[ptype], SHQ_ERR_NOMEM, "slots_convert: %ld + %ld slots of type %d > maxsize %ld", (long) first, (long) n, ptype,
              (long) slot_maxsize[ptype]);
    if(n == 0)
        return SHQ_OK;
    SHQ_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    SHQ_TRY(ctx->ex_counts.reserve(64));
    int *d_err = reinterpret_cast<int *>(ctx->ex_counts.ptr + 8);
    SHQ_HIP(hipMemsetAsync(d_err, 0, sizeof(int), st));
    convert_kernel<<<dim3(nblk(n * CONV_T)), dim3(256), 0, st>>>(n, d_index, (char *) d_parts, layout->part_elsize, layout->off_type, layout->off_pi, tab,
                                                                 layout->off_reverselink, (int) (MaxPart + 100), ptype, first, numpart, d_err);
    SHQ_HIP(hipGetLastError());
    convert_link_kernel<<<dim3(nblk(n)), dim3(256), 0, st>>>(n, d_index, (char *) d_parts, layout->part_elsize, layout->off_type, layout->off_pi, newslot, ptype, first,
                                                             numpart);
    SHQ_HIP(hipGetLastError());
    int h_err = 0;
    SHQ_HIP(hipMemcpyAsync(&h_err, d_err, sizeof(int), hipMemcpyDeviceToHost, st));
    SHQ_HIP(hipStreamSynchronize(st));
    SHQ_CHECK(h_err != 1, SHQ_ERR_INVALID, "slots_convert: a particle index lies outside [0, NumPart)");
    SHQ_CHECK(h_err == 0, SHQ_ERR_INVALID, "slots_convert: a particle's Type is not 0..5");
    if(newslot)
        slot_size[ptype] = first + n;
    return SHQ_OK;
}

/* ---- make_particle_star (sfr_eff.cpp:604-630) and blackhole_make_one (blackhole.cpp:1029-1088) for lists -------------------------
 * slots_convert of the list + the new slots' fields, read from the parents' gas slots / the particles themselves. */
namespace {

__global__ void parent_pi_kernel(long long n, const int32_t *parents, const char *parts, size_t elsize, size_t off_type, size_t off_pi, long long numpart,
                                 long long sphsize, int32_t *pi, int *err)
{
    const long long k = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= n)
        return;
    const long long p = parents[k];
    if(p < 0 || p >= numpart) {
        *err = 1;
        pi[k] = -1;
        return;
    }
    const char *rec = parts + (size_t) p * elsize;
    const int v = *(const int32_t *) (rec + off_pi);
    if(*(const unsigned char *) (rec + off_type) != 0 || v < 0 || v >= sphsize) {
        *err = 3; /* "Only gas forms stars, what's wrong?" (sfr_eff.cpp:607) / "Only Gas turns into blackholes" (blackhole.cpp:1031) */
        pi[k] = -1;
        return;
    }
    pi[k] = v;
}

__global__ void star_init_kernel(long long n, const int32_t *children, const int32_t *sphpi, const char *parts, size_t elsize, size_t off_pi, char *star, size_t star_el,
                                 const char *sph, size_t sph_el, shq_star_spawn_layout L, float Time)
{
    const long long k = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= n || sphpi[k] < 0)
        return;
    const int pi = *(const int32_t *) (parts + (size_t) children[k] * elsize + off_pi);
    char *S = star + (size_t) pi * star_el;
    const char *G = sph + (size_t) sphpi[k] * sph_el;
    *(float *) (S + L.star_formationtime) = Time;
    *(float *) (S + L.star_lastenrichmentmyr) = 0.f;
    *(double *) (S + L.star_totalmassreturned) = 0.;
    *(float *) (S + L.star_birthdensity) = (float) *(const double *) (G + L.sph_density);
    *(float *) (S + L.star_vdisp) = (float) *(const double *) (G + L.sph_vdisp);
    *(double *) (S + L.star_metallicity) = *(const double *) (G + L.sph_metallicity);
    for(int j = 0; j < L.nmetals; j++)
        ((float *) (S + L.star_metals))[j] = ((const float *) (G + L.sph_metals))[j];
}

__global__ void bh_seed_init_kernel(long long n, const int32_t *index, const double *seedmass, char *parts, size_t elsize, size_t off_pi, char *bh, size_t bh_el,
                                    shq_bh_seed_layout L, double atime, double SeedBHDynMass)
{
    const long long k = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= n)
        return;
    char *P = parts + (size_t) index[k] * elsize;
    char *B = bh + (size_t) *(const int32_t *) (P + off_pi) * bh_el;
    const double m = seedmass[k];
    *(double *) (B + L.bh_mass) = m;
    *(double *) (B + L.bh_mseed) = m;
    *(double *) (B + L.bh_mdot) = 0;
    *(double *) (B + L.bh_formationtime) = atime;
    *(unsigned long long *) (B + L.bh_swallowid) = ~0ull;
    *(double *) (B + L.bh_density) = 0;
    *(unsigned char *) (B + L.bh_timebindynfric) = *(const unsigned char *) (P + L.part_timebin_hydro);
    for(int j = 0; j < 3; j++) {
        ((double *) (B + L.bh_minpotpos))[j] = ((const double *) (P + L.part_pos))[j];
        ((double *) (B + L.bh_dfaccel))[j] = 0;
        ((double *) (B + L.bh_df_surroundingvel))[j] = 0;
        ((double *) (B + L.bh_dragaccel))[j] = 0;
    }
    *(double *) (B + L.bh_df_surroundingrmsvel) = 0;
    *(double *) (B + L.bh_df_surroundingdensity) = 0;
    *(signed char *) (B + L.bh_jumptominpot) = 0;
    *(int32_t *) (B + L.bh_countprogs) = 1;
    if(SeedBHDynMass > 0) {
        *(double *) (B + L.bh_mtrack) = (double) *(const float *) (P + L.part_mass);
        *(float *) (P + L.part_mass) = (float) SeedBHDynMass;
    } else
        *(double *) (B + L.bh_mtrack) = -1;
    *(double *) (B + L.bh_kineticfdbkenergy) = 0;
    *(double *) (B + L.bh_vdisp) = 0;
}

} // namespace

extern "C" int shq_make_particle_stars(shq_context *ctx, const shq_exchange_layout *layout, const shq_star_spawn_layout *sl, void *d_parts, int64_t numpart,
                                       int64_t MaxPart, void *const d_slots[6], int64_t slot_size[6], const int64_t slot_maxsize[6], const int32_t *d_children,
                                       const int32_t *d_parents, int64_t n, double Time)
{
    SHQ_CHECK(ctx && sl && d_parts && slot_size && slot_maxsize && d_slots, SHQ_ERR_INVALID, "null argument");
    SHQ_TRY(check_layout(layout));
    SHQ_CHECK(layout->slot_elsize[0] && layout->slot_elsize[4] && d_slots[0] && d_slots[4], SHQ_ERR_INVALID, "make_particle_stars: gas and star slots must be enabled");
    SHQ_CHECK(n >= 0 && (n == 0 || (d_children && d_parents)), SHQ_ERR_INVALID, "make_particle_stars: bad list");
    SHQ_CHECK(sl->nmetals >= 0 && sl->star_metals + 4 * (size_t) sl->nmetals <= layout->slot_elsize[4] && sl->sph_metals + 4 * (size_t) sl->nmetals <= layout->slot_elsize[0],
              SHQ_ERR_INVALID, "make_particle_stars: bad slot layout");
    if(n == 0)
        return SHQ_OK;
    SHQ_CHECK(slot_size[4] + n <= slot_maxsize[4], SHQ_ERR_NOMEM, "make_particle_stars: %ld + %ld star slots > maxsize %ld (sfr_reserve_slots first)", (long) slot_size[4],
              (long) n, (long) slot_maxsize[4]);
    SHQ_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    SHQ_TRY(ctx->ex_counts.reserve(64));
    SHQ_TRY(ctx->ex_val[2].reserve((size_t) n));
    int *d_err = reinterpret_cast<int *>(ctx->ex_counts.ptr + 8);
    SHQ_HIP(hipMemsetAsync(d_err, 0, sizeof(int), st));
    /* oldslot = SPHP(parent), before the conversion overwrites a converted parent's PI */
    parent_pi_kernel<<<dim3(nblk(n)), dim3(256), 0, st>>>(n, d_parents, (const char *) d_parts, layout->part_elsize, layout->off_type, layout->off_pi, numpart, slot_size[0],
                                                          ctx->ex_val[2].ptr, d_err);
    SHQ_HIP(hipGetLastError());
    int h_err = 0;
    SHQ_HIP(hipMemcpyAsync(&h_err, d_err, sizeof(int), hipMemcpyDeviceToHost, st));
    SHQ_HIP(hipStreamSynchronize(st));
    SHQ_CHECK(h_err != 1, SHQ_ERR_INVALID, "make_particle_stars: a parent index lies outside [0, NumPart)");
    SHQ_CHECK(h_err == 0, SHQ_ERR_INVALID, "make_particle_stars: only gas forms stars (sfr_eff.cpp:607)");
    SHQ_TRY(shq_slots_convert(ctx, layout, d_parts, numpart, MaxPart, d_slots, slot_size, slot_maxsize, d_children, n, 4));
    star_init_kernel<<<dim3(nblk(n)), dim3(256), 0, st>>>(n, d_children, ctx->ex_val[2].ptr, (const char *) d_parts, layout->part_elsize, layout->off_pi, (char *) d_slots[4],
                                                          layout->slot_elsize[4], (const char *) d_slots[0], layout->slot_elsize[0], *sl, (float) Time);
    SHQ_HIP(hipGetLastError());
    SHQ_HIP(hipStreamSynchronize(st));
    return SHQ_OK;
}

extern "C" int shq_blackhole_make_seeds(shq_context *ctx, const shq_exchange_layout *layout, const shq_bh_seed_layout *bl, void *d_parts, int64_t numpart, int64_t MaxPart,
                                        void *const d_slots[6], int64_t slot_size[6], const int64_t slot_maxsize[6], const int32_t *d_index, const double *d_seedmass,
                                        int64_t n, double atime, double SeedBHDynMass)
{
    SHQ_CHECK(ctx && bl && d_parts && slot_size && slot_maxsize && d_slots, SHQ_ERR_INVALID, "null argument");
    SHQ_TRY(check_layout(layout));
    SHQ_CHECK(layout->slot_elsize[0] && layout->slot_elsize[5] && d_slots[5], SHQ_ERR_INVALID, "blackhole_make_seeds: gas and black-hole slots must be enabled");
    SHQ_CHECK(n >= 0 && (n == 0 || (d_index && d_seedmass)), SHQ_ERR_INVALID, "blackhole_make_seeds: bad list");
    if(n == 0)
        return SHQ_OK;
    SHQ_CHECK(slot_size[5] + n <= slot_maxsize[5], SHQ_ERR_NOMEM, "blackhole_make_seeds: %ld + %ld black-hole slots > maxsize %ld (fof_seed reserves first, fof.cpp:1343-1366)",
              (long) slot_size[5], (long) n, (long) slot_maxsize[5]);
    SHQ_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    SHQ_TRY(ctx->ex_counts.reserve(64));
    SHQ_TRY(ctx->ex_val[2].reserve((size_t) n));
    int *d_err = reinterpret_cast<int *>(ctx->ex_counts.ptr + 8);
    SHQ_HIP(hipMemsetAsync(d_err, 0, sizeof(int), st));
    parent_pi_kernel<<<dim3(nblk(n)), dim3(256), 0, st>>>(n, d_index, (const char *) d_parts, layout->part_elsize, layout->off_type, layout->off_pi, numpart, slot_size[0],
                                                          ctx->ex_val[2].ptr, d_err);
    SHQ_HIP(hipGetLastError());
    int h_err = 0;
    SHQ_HIP(hipMemcpyAsync(&h_err, d_err, sizeof(int), hipMemcpyDeviceToHost, st));
    SHQ_HIP(hipStreamSynchronize(st));
    SHQ_CHECK(h_err != 1, SHQ_ERR_INVALID, "blackhole_make_seeds: a seed index lies outside [0, NumPart)");
    SHQ_CHECK(h_err == 0, SHQ_ERR_INVALID, "blackhole_make_seeds: only gas turns into black holes (blackhole.cpp:1031)");
    SHQ_TRY(shq_slots_convert(ctx, layout, d_parts, numpart, MaxPart, d_slots, slot_size, slot_maxsize, d_index, n, 5));
    bh_seed_init_kernel<<<dim3(nblk(n)), dim3(256), 0, st>>>(n, d_index, d_seedmass, (char *) d_parts, layout->part_elsize, layout->off_pi, (char *) d_slots[5],
                                                             layout->slot_elsize[5], *bl, atime, SeedBHDynMass);
    SHQ_HIP(hipGetLastError());
    SHQ_HIP(hipStreamSynchronize(st));
    return SHQ_OK;
}
