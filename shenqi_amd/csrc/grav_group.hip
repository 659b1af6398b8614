/* grav_group.hip — short-range tree walk, source-parallel flavour (walk_mode SHQ_WALK_GROUP) for gfx950.
 *
 * Same result as grav_walk.hip: every target meets exactly the interaction set of its own reference walk
 * (GravLocalTreeWalk::visit, libgadget/gravshort2.hpp:227-322, with shall_we_discard_node / shall_we_open_node :152-193 and
 * apply_accn :326-358) — interaction counts equal the reference's as integers; only the order in which a target's
 * contributions are summed differs (forces agree to rounding, ~1e-15 relative).
 *
 * The work is the transpose of grav_walk.hip's — LANES ARE NODES / SOURCES, not targets — and it is cut in two kernels, because
 * the two halves want opposite things from the machine:
 *
 *   grav_list_kernel (traversal; latency bound, few registers, many waves).  A wave takes 64 consecutive targets as 8 groups
 *   of 8 and walks, for one group at a time, the UNION of its members' reference walks; every pending node carries the 8-bit
 *   mask of the members whose own walk reaches it.
 *     - T1 rounds: up to 64 (node, mask) entries are popped from a per-wave LDS stack, one per lane.  Each lane tests its node
 *       against the group's bounding box: from the nearest and the farthest point of the box it is usually certain that every
 *       masked member discards the node, or that none discards and none opens it (=> one source for all of them), or that
 *       all open it (=> its children / its particles inherit the mask).  The comparisons carry a relative margin far above
 *       rounding, so a certain answer is the answer of every member's own test;
 *     - T2 rounds: the nodes T1 could not settle (~15 %) wait in a small LDS pile and are tested 64 at a time member by
 *       member, with the expressions of grav_walk.hip: per-member accept and open masks;
 *     - accepted monopoles and the particles of opened leaves are appended, as (source index, member mask), to the group's
 *       interaction list in HBM: chunks of CH entries from a pool, chained per group.
 *
 *   grav_eval_kernel (arithmetic; FP64-issue bound, all lanes busy).  For every group it streams the list: each lane takes one
 *   source, gathers its 32-byte (x, y, z, m) record and applies apply_accn to the members in its mask, whose coordinates are
 *   wave-uniform; the lanes' partial sums are added across the wave once per group.  Index loads and gathers of the coming
 *   rounds are in flight while a round is evaluated.
 *
 * A node test thus costs a lane-slot instead of a wave-round, and the interaction arithmetic runs at the fill of the group's
 * union list (~70 %) instead of the 44 % of the 64-target union walk.  The lists cost 5 bytes per entry written and read
 * (~15 GB per walk at 256^3) and live in a pool the host sizes per batch of targets.
 */
#include "common.hpp"
#include <algorithm>

namespace {

constexpr int GS = 8;            /* targets per group */
constexpr int WAVES = 4;         /* waves per workgroup */
/* Pending work of a group: a LIFO stack of (node, mask) entries for T1 and a LIFO pile of entries waiting for T2.  A popped
 * entry is replaced by at most 8 (its children, directly or after its T2), so I = stack fill + 8 * pile fill grows by at
 * most 7 per popped entry and never under T2.  While I < STACK_SOFT a T1 round pops as many entries as keep I below it;
 * above, the walk proceeds strictly depth-first (one entry per round, newest first), which adds at most 8 per tree level:
 * STACK_CAP - STACK_SOFT = 256 covers 32 levels (the device tree build stops at 21); beyond that the walk reports an error
 * instead of a result. */
constexpr int STACK_CAP = 1280;
constexpr int STACK_SOFT = 1024;
constexpr int AMB_CAP = 128;     /* per-wave pile of nodes waiting for the per-member tests: < 64 + <= 64 per T1 round */
constexpr int FR_CAP = 512;      /* per-wave frontier of the shared phase: nodes the wave's box cannot settle */
constexpr int CH = 512;          /* list entries per pool chunk (a half-wave appends at most 256 at a time: never more than two chunks) */

enum { ACT_T2, ACT_T1 };

struct ListArgs {
    const NodeG *G;            /* merged node record (grav_walk.hip's): cofm, mass, centre, len, links, per-node products: T2 only */
    const float4 *F;           /* [4 * node]: one 64-byte T1 record: (cofm, mass), (centre, len) rounded to f32, then 8 ints: the children
                                  of an internal node (-1 padded); for a leaf: first leaf slot, count, ..., -2; pseudo node: ..., -3 */
    const double4 *posm;       /* by particle index */
    const double *oldacc;
    const int32_t *targets;    /* may be null */
    long long ntargets;        /* of this launch */
    int root;
    double Box, invBox;
    double rcut, rcut2, bh2;
    double errtol;
    unsigned xcdK;
    /* interaction lists */
    int32_t *pool_idx;         /* [nchunks * CH]: >= 0 node index, < 0: -1 - leaf slot */
    uint8_t *pool_msk;         /* [nchunks * CH] */
    int32_t *chunk_cnt;        /* [nchunks] entries in the chunk */
    int32_t *chunk_next;       /* [nchunks] next chunk of the group or -1 */
    int32_t *group_head;       /* [groups of this launch] first chunk */
    int32_t *wave_head;        /* [waves of this launch] first chunk of the list all 64 targets share, or -1 */
    int nchunks;
    int *counters;             /* [0] chunks handed out, [1] error flags: 1 stack overflow, 2 pool exhausted */
    unsigned long long *ntests; /* node tests, summed */
    unsigned long long *dbg;    /* SHQ_WALK_STATS=2: round counters (GravStatsDev.hist_visit) */
};

struct EvalArgs {
    const NodeA *A;            /* cofm + mass of every node */
    const double4 *posm_leaf;  /* leaf order */
    const double4 *posm;       /* by particle index */
    const int32_t *targets;
    long long ntargets;
    double *acc;
    double *pot;
    int32_t *nint;
    GravStatsDev *stats;
    double Box, invBox;
    double h2, h_inv, h3_inv;
    double inv_celldx;
    double edge;               /* a group with a member this close to a box face wraps every pair */
    unsigned xcdK;
    const int32_t *pool_idx;
    const uint8_t *pool_msk;
    const int32_t *chunk_cnt;
    const int32_t *chunk_next;
    const int32_t *group_head;
    const int32_t *wave_head;
    const float *tab_f;
    const float *tab_p;
};

__device__ __forceinline__ double readlane_d(double v, int srclane)
{
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readlane((unsigned) u, srclane);
    const unsigned hi = __builtin_amdgcn_readlane((unsigned) (u >> 32), srclane);
    return __longlong_as_double(((unsigned long long) hi << 32) | lo);
}

__device__ __forceinline__ int mbcnt64(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned) (m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) m, 0u));
}

/* NEAREST (partmanager.h:99) as in grav_walk.hip */
__device__ __forceinline__ double wrapd_g(double d, double L, double invL) { return fma(-L, rint(d * invL), d); }
__device__ __forceinline__ float wrapf_g(float d, float L, float invL) { return fmaf(-L, rintf(d * invL), d); }

__device__ __forceinline__ double rsqrt_nr(double x)
{
    const double y0 = __builtin_amdgcn_rsq(x);
    const double e = fma(-x * y0, y0, 1.0);
    return fma(y0 * e, fma(e, 0.375, 0.5), y0);
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* per-wave writer of a group's interaction list: chunks of CH entries, the next chunk always reserved ahead */
struct ListWriter {
    int cur, nxt, fill;
    __device__ __forceinline__ int alloc(const ListArgs &a, int lane)
    {
        int id = 0;
        if(lane == 0)
            id = atomicAdd(a.counters, 1);
        id = __builtin_amdgcn_readfirstlane(id);
        if(id >= a.nchunks) {
            if(lane == 0)
                atomicOr(a.counters + 1, 2);
            id = -1;
        }
        return id;
    }
    __device__ __forceinline__ void append(const ListArgs &a, int lane, bool has, int val, unsigned m)
    {
        const unsigned long long msk = shq_ballot(has);
        if(msk == 0ull)
            return;
        const int pos = fill + mbcnt64(msk);
        const int c = pos < CH ? cur : nxt;
        const int o = pos < CH ? pos : pos - CH;
        if(has && c >= 0) {
            a.pool_idx[(long long) c * CH + o] = val;
            a.pool_msk[(long long) c * CH + o] = (uint8_t) m;
        }
        fill += __popcll(msk);
        if(fill >= CH) {
            if(lane == 0 && cur >= 0) {
                a.chunk_cnt[cur] = CH;
                a.chunk_next[cur] = nxt;
            }
            cur = nxt;
            fill -= CH;
            nxt = cur >= 0 ? alloc(a, lane) : -1;
        }
    }
    /* every lane appends `cnt` (<= 8) consecutive entries val0, val0 - 1, ... with mask m: the leaf slots of an opened leaf.
     * Half a wave at a time, so that one call never spans more than two chunks. */
    __device__ __forceinline__ void append_run(const ListArgs &a, int lane, int cnt, int val0, unsigned m)
    {
#pragma unroll
        for(int half = 0; half < 2; half++) {
            const int c_ = ((lane >> 5) == half) ? cnt : 0;
            const unsigned long long b0 = shq_ballot((c_ & 1) != 0), b1 = shq_ballot((c_ & 2) != 0), b2 = shq_ballot((c_ & 4) != 0),
                                     b3 = shq_ballot((c_ & 8) != 0);
            if((b0 | b1 | b2 | b3) == 0ull)
                continue;
            const int pre = mbcnt64(b0) + 2 * mbcnt64(b1) + 4 * mbcnt64(b2) + 8 * mbcnt64(b3);
            const int total = __popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2) + 8 * __popcll(b3);
            const int most = b3 ? 8 : (b2 ? 7 : (b1 ? 3 : 1));
            for(int j = 0; j < most; j++) {
                if(j < c_) {
                    const int pos = fill + pre + j;
                    const int c = pos < CH ? cur : nxt;
                    const int o = pos < CH ? pos : pos - CH;
                    if(c >= 0) {
                        a.pool_idx[(long long) c * CH + o] = val0 - j;
                        a.pool_msk[(long long) c * CH + o] = (uint8_t) m;
                    }
                }
            }
            fill += total;
            if(fill >= CH) {
                if(lane == 0 && cur >= 0) {
                    a.chunk_cnt[cur] = CH;
                    a.chunk_next[cur] = nxt;
                }
                cur = nxt;
                fill -= CH;
                nxt = cur >= 0 ? alloc(a, lane) : -1;
            }
        }
    }
    __device__ __forceinline__ void close_group(const ListArgs &a, int lane)
    {
        if(lane == 0 && cur >= 0) {
            a.chunk_cnt[cur] = fill;
            a.chunk_next[cur] = -1;
        }
        cur = nxt;
        fill = 0;
        nxt = cur >= 0 ? alloc(a, lane) : -1;
    }
};

/* T1: where does the node stand for every target inside the box (centre c, half extents h, already widened by the f32
 * error bound)?  0 every target discards it, 1 none discards and none opens it, 2 none discards and all open it, 3 undecided.
 * The comparisons keep a relative margin of 1e-5 >> the f32 rounding of the few products behind them. */
struct BoxF {
    float cx, cy, cz, hx, hy, hz, amin, amax;
};
struct WalkF {
    float L, iL, rcut, rcut2, ibh2;
    bool BH;
};
__device__ __forceinline__ int t1_classify(const float4 fa, const float4 fb, const BoxF &b, const WalkF &c)
{
    const float d0 = fabsf(wrapf_g(fa.x - b.cx, c.L, c.iL)), d1 = fabsf(wrapf_g(fa.y - b.cy, c.L, c.iL)), d2 = fabsf(wrapf_g(fa.z - b.cz, c.L, c.iL));
    const float q0 = fabsf(wrapf_g(fb.x - b.cx, c.L, c.iL)), q1 = fabsf(wrapf_g(fb.y - b.cy, c.L, c.iL)), q2 = fabsf(wrapf_g(fb.z - b.cz, c.L, c.iL));
    const float n0 = fmaxf(d0 - b.hx, 0.f), n1 = fmaxf(d1 - b.hy, 0.f), n2 = fmaxf(d2 - b.hz, 0.f);
    const float f0 = d0 + b.hx, f1 = d1 + b.hy, f2 = d2 + b.hz;
    const float r2min = n0 * n0 + n1 * n1 + n2 * n2, r2max = f0 * f0 + f1 * f1 + f2 * f2;
    const float cnear = fmaxf(fmaxf(fmaxf(q0 - b.hx, q1 - b.hy), q2 - b.hz), 0.f);
    const float cfar = fmaxf(fmaxf(q0 + b.hx, q1 + b.hy), q2 + b.hz);
    const float len = fb.w, mlen2 = fa.w * len * len, bhlim = len * len * c.ibh2, inside = 0.6f * len, rcuthl = c.rcut + 0.5f * len;
    const float up = 1.0f + 1e-5f, dn = 1.0f - 1e-5f;
    const bool alldiscard = (r2min > c.rcut2 * up) && (cnear > rcuthl * up);
    const bool nonediscards = (r2max < c.rcut2 * dn) || (cfar < rcuthl * dn);
    const bool allopen = (!c.BH && (mlen2 * dn > r2max * r2max * b.amax)) || (r2max * up < bhlim) || (cfar < inside * dn);
    const bool noneopen = (c.BH || (mlen2 * up < r2min * r2min * b.amin)) && (r2min > bhlim * up) && (cnear > inside * up);
    if(alldiscard)
        return 0;
    if(nonediscards && noneopen)
        return 1;
    if(nonediscards && allopen)
        return 2;
    return 3;
}

__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(4, 4))) void grav_list_kernel(const ListArgs a, const int useBH)
{
    __shared__ int stack_s[WAVES][STACK_CAP];
    __shared__ unsigned char stackm_s[WAVES][STACK_CAP];
    __shared__ int2 amb_s[WAVES][AMB_CAP];
    __shared__ int front_s[WAVES][FR_CAP];

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); /* wave-uniform for the compiler too */
    int *__restrict__ stk = stack_s[wv];
    unsigned char *__restrict__ stkm = stackm_s[wv];
    int2 *__restrict__ amb = amb_s[wv];
    int *__restrict__ fr = front_s[wv];
    const long long wave = (long long) xcd_block(blockIdx.x, gridDim.x, a.xcdK) * WAVES + wv;
    const long long t = wave * 64 + lane;
    if(wave * 64 >= a.ntargets)
        return;
    const bool valid = t < a.ntargets;
    const long long tt = valid ? t : a.ntargets - 1; /* clones of the last target pad the last group: they change no box */
    const long long pi = a.targets ? (long long) a.targets[tt] : tt;
    const double4 p = a.posm[pi];
    const double aold = a.errtol * a.oldacc[pi];
    WalkF wf;
    wf.BH = useBH != 0;
    wf.L = (float) a.Box;
    wf.iL = (float) a.invBox;
    wf.rcut = (float) a.rcut;
    wf.rcut2 = (float) a.rcut2;
    wf.ibh2 = (float) (1.0 / a.bh2);
    const bool BH = wf.BH;
    /* f32 view of a box for T1: widened by dpad = 1e-6 Box, more than three times what f32 rounding of a coordinate difference
     * inside the box can amount to (4 roundings of at most 2^-24 Box each) */
    const float dpad = 1e-6f * wf.L;

    ListWriter w;
    w.cur = w.alloc(a, lane);
    w.nxt = w.cur >= 0 ? w.alloc(a, lane) : -1;
    w.fill = 0;

    unsigned long long ntests = 0;
    unsigned dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0}; /* rounds: shared T1, group T1, T2; rounds with leaf runs, with pushes; frontier size; entries */
    const long long left_w = a.ntargets - wave * 64;
    const int ngroups = (int) (((left_w < 64 ? left_w : 64) + GS - 1) / GS);
    int S = 0;

    /* children of the opened internal nodes of a round go on the stack with the opening members' mask */
    auto push_children = [&](bool openint, int4 k0, int4 k1, unsigned m) {
        if(shq_ballot(openint) == 0ull)
            return;
        if(!openint)
            k0 = k1 = make_int4(-1, -1, -1, -1);
        const int kid[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
        int nk = 0;
#pragma unroll
        for(int j = 0; j < 8; j++)
            nk += kid[j] >= 0 ? 1 : 0; /* the children fill the first nk slots */
        const unsigned long long b0 = shq_ballot((nk & 1) != 0), b1 = shq_ballot((nk & 2) != 0), b2 = shq_ballot((nk & 4) != 0),
                                 b3 = shq_ballot((nk & 8) != 0);
        const int at = S + mbcnt64(b0) + 2 * mbcnt64(b1) + 4 * mbcnt64(b2) + 8 * mbcnt64(b3);
        const int most = b3 ? 8 : (b2 ? 7 : (b1 ? 3 : 1));
#pragma unroll
        for(int j = 0; j < 8; j++) {
            if(j >= most)
                break;
            if(j < nk && at + j < STACK_CAP) {
                stk[at + j] = kid[j];
                stkm[at + j] = (unsigned char) m;
            }
        }
        S += __popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2) + 8 * __popcll(b3);
    };

    /* ---- phase A: the part of the walk all 64 targets share.  One traversal against the box of the whole wave: a node every
     * target discards is dropped once, a monopole (or the particles of a leaf) every target accepts goes once to the wave's
     * common list, a node every target opens is descended once; what the wave's box cannot settle becomes the frontier from
     * which every group starts its own traversal. */
    int F = 0;
    {
        double lo[3] = {p.x, p.y, p.z}, hi[3] = {p.x, p.y, p.z}, amn = aold, amx = aold;
#pragma unroll
        for(int off = 1; off < 64; off <<= 1) {
#pragma unroll
            for(int k = 0; k < 3; k++) {
                lo[k] = fmin(lo[k], __shfl_xor(lo[k], off));
                hi[k] = fmax(hi[k], __shfl_xor(hi[k], off));
            }
            amn = fmin(amn, __shfl_xor(amn, off));
            amx = fmax(amx, __shfl_xor(amx, off));
        }
        BoxF wb;
        wb.cx = (float) (0.5 * (lo[0] + hi[0]));
        wb.cy = (float) (0.5 * (lo[1] + hi[1]));
        wb.cz = (float) (0.5 * (lo[2] + hi[2]));
        wb.hx = (float) (0.5 * (hi[0] - lo[0])) + dpad;
        wb.hy = (float) (0.5 * (hi[1] - lo[1])) + dpad;
        wb.hz = (float) (0.5 * (hi[2] - lo[2])) + dpad;
        wb.amin = (float) amn * (1.0f - 1e-6f);
        wb.amax = (float) amx * (1.0f + 1e-6f);
        if(lane == 0) {
            a.wave_head[wave] = w.cur;
            stk[0] = a.root;
        }
        S = 1;
        bool overflow = false;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        while(S > 0) {
            int k = S < 64 ? S : 64;
            const int room = (STACK_SOFT - S) / 7;
            if(room < k)
                k = room > 1 ? room : 1;
            const bool on = lane < k;
            const int node = on ? stk[S - 1 - lane] : 0;
            S -= k;
            ntests += (unsigned) k;
            __builtin_amdgcn_wave_barrier();
            int cls = 0;
            dbg[0]++;
            int4 k0 = make_int4(-1, -1, -1, -1), k1 = make_int4(-1, -1, -1, -3);
            if(on) {
                const float4 fa = a.F[4 * (long long) node], fb = a.F[4 * (long long) node + 1];
                k0 = *reinterpret_cast<const int4 *>(&a.F[4 * (long long) node + 2]);
                k1 = *reinterpret_cast<const int4 *>(&a.F[4 * (long long) node + 3]);
                cls = t1_classify(fa, fb, wb, wf);
            }
            w.append(a, lane, cls == 1, node, 0xffu);
            const bool openleaf = cls == 2 && k1.w == -2;
            if(shq_ballot(openleaf) != 0ull) {
                dbg[3]++;
                w.append_run(a, lane, openleaf ? k0.y : 0, -1 - k0.x, 0xffu);
            }
            if(shq_ballot(cls == 2 && k1.w >= -1) != 0ull)
                dbg[4]++;
            push_children(cls == 2 && k1.w >= -1, k0, k1, 0xffu);
            const unsigned long long fm = shq_ballot(cls == 3);
            const int at = F + mbcnt64(fm);
            if(cls == 3 && at < FR_CAP)
                fr[at] = node;
            F += __popcll(fm);
            if(F > FR_CAP || S > STACK_CAP) {
                overflow = true;
                break;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        dbg[5] += (unsigned) F;
        w.close_group(a, lane);
        if(overflow) { /* the shared part does not fit (a very wide wave box): every group walks from the root, no common list */
            if(lane == 0) {
                a.wave_head[wave] = -1;
                fr[0] = a.root;
            }
            F = 1;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    /* ---- phase B: every group of 8 from the frontier on */
    for(int g = 0; g < ngroups; g++) {
        const int l0 = g * GS;
        const long long left = left_w - l0;
        const int cnt = left < GS ? (int) left : GS;
        const unsigned fullmask = (1u << cnt) - 1u;
        if(lane == 0)
            a.group_head[wave * 8 + g] = w.cur;

        /* the group's targets, wave-uniform */
        double tx[GS], ty[GS], tz[GS], ta[GS];
#pragma unroll
        for(int i = 0; i < GS; i++) {
            tx[i] = readlane_d(p.x, l0 + i);
            ty[i] = readlane_d(p.y, l0 + i);
            tz[i] = readlane_d(p.z, l0 + i);
            ta[i] = readlane_d(aold, l0 + i);
        }
        double lx = tx[0], ly = ty[0], lz = tz[0], ux = tx[0], uy = ty[0], uz = tz[0], amin = ta[0], amax = ta[0];
#pragma unroll
        for(int i = 1; i < GS; i++) {
            lx = fmin(lx, tx[i]);
            ly = fmin(ly, ty[i]);
            lz = fmin(lz, tz[i]);
            ux = fmax(ux, tx[i]);
            uy = fmax(uy, ty[i]);
            uz = fmax(uz, tz[i]);
            amin = fmin(amin, ta[i]);
            amax = fmax(amax, ta[i]);
        }
        const double cx = 0.5 * (lx + ux), cy = 0.5 * (ly + uy), cz = 0.5 * (lz + uz);
        const double hx = 0.5 * (ux - lx), hy = 0.5 * (uy - ly), hz = 0.5 * (uz - lz);
        BoxF gb;
        gb.cx = (float) cx;
        gb.cy = (float) cy;
        gb.cz = (float) cz;
        gb.hx = (float) hx + dpad;
        gb.hy = (float) hy + dpad;
        gb.hz = (float) hz + dpad;
        gb.amin = (float) amin * (1.0f - 1e-6f);
        gb.amax = (float) amax * (1.0f + 1e-6f);

        int amb_n = 0;
        for(int i = lane; i < F; i += 64) {
            stk[i] = fr[i];
            stkm[i] = (unsigned char) fullmask;
        }
        S = F;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        for(;;) {
            int action;
            if(amb_n >= 64 || (amb_n > 0 && S + 8 * amb_n >= STACK_SOFT))
                action = ACT_T2;
            else if(S > 0)
                action = ACT_T1;
            else if(amb_n > 0)
                action = ACT_T2;
            else
                break;

            /* ---- node tests.  T1: box test of stack entries; T2: per-member tests of the entries T1 left open */
            int k;
            bool on;
            int2 ent = make_int2(0, 0);
            if(action == ACT_T1) {
                k = S < 64 ? S : 64;
                const int room = (STACK_SOFT - S - 8 * amb_n) / 7;
                if(room < k)
                    k = room > 1 ? room : 1;
                on = lane < k;
                if(on)
                    ent = make_int2(stk[S - 1 - lane], (int) stkm[S - 1 - lane]);
                S -= k;
            } else {
                k = amb_n < 64 ? amb_n : 64;
                on = lane < k;
                if(on)
                    ent = amb[amb_n - 1 - lane];
                amb_n -= k;
            }
            ntests += (unsigned) k;
            dbg[action == ACT_T1 ? 1 : 2]++;
            __builtin_amdgcn_wave_barrier();
            const int node = ent.x;
            const unsigned mask = (unsigned) ent.y;
            unsigned accm = 0, openm = 0;
            int4 k0 = make_int4(-1, -1, -1, -1), k1 = make_int4(-1, -1, -1, -3); /* links of the node, as in its T1 record */
            if(action == ACT_T1) {
                int cls = 0;
                if(on) {
                    const float4 fa = a.F[4 * (long long) node], fb = a.F[4 * (long long) node + 1];
                    k0 = *reinterpret_cast<const int4 *>(&a.F[4 * (long long) node + 2]);
                    k1 = *reinterpret_cast<const int4 *>(&a.F[4 * (long long) node + 3]);
                    cls = t1_classify(fa, fb, gb, wf);
                }
                accm = cls == 1 ? mask : 0u;
                openm = cls == 2 ? mask : 0u;
                /* unsettled entries wait for the per-member tests */
                const unsigned long long am = shq_ballot(cls == 3);
                if(cls == 3)
                    amb[amb_n + mbcnt64(am)] = ent;
                amb_n += __popcll(am);
            } else {
                /* per-member tests on the f64 record with grav_walk.hip's expressions (gravshort2.hpp:152-193) */
                NodeG nd;
                if(on) {
                    nd = a.G[node];
                    k0 = *reinterpret_cast<const int4 *>(&a.F[4 * (long long) node + 2]);
                    k1 = *reinterpret_cast<const int4 *>(&a.F[4 * (long long) node + 3]);
                } else {
                    nd.cofm[0] = nd.cofm[1] = nd.cofm[2] = nd.mass = 0;
                    nd.center[0] = nd.center[1] = nd.center[2] = nd.len = 0;
                    nd.sibling = nd.child = -1;
                    nd.type = SHQ_PSEUDO_NODE_TYPE;
                    nd.count = 0;
                    nd.bhlim = nd.mlen2 = nd.inside = nd.rcut2 = nd.wraplim = nd.rcuthl = 0;
                }
                const bool anywrap = shq_ballot(on && (fmax(fmax(fabs(nd.center[0] - cx) + hx, fabs(nd.center[1] - cy) + hy),
                                                              fabs(nd.center[2] - cz) + hz) > fabs(nd.wraplim))) != 0ull; /* sign bit: interior flag of the exact walk */
#pragma unroll
                for(int i = 0; i < GS; i++) {
                    double dx = nd.cofm[0] - tx[i], dy = nd.cofm[1] - ty[i], dz = nd.cofm[2] - tz[i];
                    double ex = nd.center[0] - tx[i], ey = nd.center[1] - ty[i], ez = nd.center[2] - tz[i];
                    if(anywrap) {
                        dx = wrapd_g(dx, a.Box, a.invBox);
                        dy = wrapd_g(dy, a.Box, a.invBox);
                        dz = wrapd_g(dz, a.Box, a.invBox);
                        ex = wrapd_g(ex, a.Box, a.invBox);
                        ey = wrapd_g(ey, a.Box, a.invBox);
                        ez = wrapd_g(ez, a.Box, a.invBox);
                    }
                    const double cmax = fmax(fmax(fabs(ex), fabs(ey)), fabs(ez));
                    const double r2 = dx * dx + dy * dy + dz * dz;
                    const bool discard = (r2 > a.rcut2) && (cmax > nd.rcuthl);
                    const bool open = (!BH && (nd.mlen2 > r2 * r2 * ta[i])) || (r2 < nd.bhlim) || (cmax < nd.inside);
                    const unsigned bit = (mask >> i) & 1u;
                    accm |= (bit & (unsigned) (!discard && !open)) << i;
                    openm |= (bit & (unsigned) (!discard && open)) << i;
                    __builtin_amdgcn_sched_barrier(0);
                }
                if(!on)
                    accm = openm = 0;
            }
            /* accepted monopoles join the group's list */
            w.append(a, lane, accm != 0, node, accm);
            /* so do the particles of opened leaves (gravshort2.hpp:290-304: all of them are evaluated) */
            const bool openleaf = openm != 0 && k1.w == -2;
            if(shq_ballot(openleaf) != 0ull) {
                dbg[3]++;
                w.append_run(a, lane, openleaf ? k0.y : 0, -1 - k0.x, openm);
            }
            if(shq_ballot(openm != 0 && k1.w >= -1) != 0ull)
                dbg[4]++;
            push_children(openm != 0 && k1.w >= -1, k0, k1, openm);
            if(S > STACK_CAP) { /* cannot happen for trees of fewer than 32 levels: refuse rather than walk a truncated stack */
                if(lane == 0)
                    atomicOr(a.counters + 1, 1);
                S = 0;
                amb_n = 0;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        w.close_group(a, lane);
    }
    if(lane == 0) {
        atomicAdd(a.ntests, ntests);
        if(a.dbg)
            for(int k = 0; k < 8; k++)
                atomicAdd(a.dbg + k, (unsigned long long) dbg[k]);
    }
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* apply_accn (gravshort2.hpp:326-358) + apply_short_range_window (gravity.h:48-60), as in grav_walk.hip */
template <bool POT>
__device__ __forceinline__ void apply_accn_g(const double4 *__restrict__ tab, double dx, double dy, double dz, double r2, double mass,
                                             const EvalArgs &a, double &ax, double &ay, double &az, double &pot)
{
    const double r2c = fmax(r2, 1e-280); /* a source may be the target itself: softened branch, dx * fac = 0 */
    const double rinv = rsqrt_nr(r2c);
    const double r = r2c * rinv;
    const double mr = mass * rinv;
    double fac = mr * rinv * rinv;
    double facpot = -mr;
    if(r2 < a.h2) {
        const double u = r * a.h_inv;
        double wp;
        if(u < 0.5) {
            fac = mass * a.h3_inv * (10.666666666667 + u * u * (32.0 * u - 38.4));
            wp = -2.8 + u * u * (5.333333333333 + u * u * (6.4 * u - 9.6));
            facpot = mass * a.h_inv * wp;
        } else {
            fac = fma(mass * a.h3_inv, 21.333333333333 - 48.0 * u + 38.4 * u * u - 10.666666666667 * u * u * u, -0.066666666667 * fac);
            wp = -3.2 + u * u * (10.666666666667 + u * (-16.0 + u * (9.6 - 2.133333333333 * u)));
            facpot = fma(mass * a.h_inv, wp, 0.066666666667 * mr);
        }
    }
    const double fi = r * a.inv_celldx;
    if(fi < (double) (SHQ_NGRAVTAB - 1)) {
        const int ti = (int) fi;
        const double w1 = __builtin_amdgcn_fract(fi);
        if(POT) {
            const double4 t = tab[ti];
            fac *= fma(w1, t.y, t.x);
            pot = fma(facpot, fma(w1, t.w, t.z), pot);
        } else {
            const double2 t = *reinterpret_cast<const double2 *>(&tab[ti]);
            fac *= fma(w1, t.y, t.x);
        }
        ax = fma(dx, fac, ax);
        ay = fma(dy, fac, ay);
        az = fma(dz, fac, az);
    }
}

/* position in a group's interaction list: the chunk chain all 64 targets of the wave share, then the group's own; chunk,
 * offset of the round in it, its fill and its successor */
struct ListPos {
    int c, o, cnt, nx, then; /* then: head of the chain to continue with, -2 = none left */
    __device__ __forceinline__ void open(const EvalArgs &a, int chunk)
    {
        for(;;) {
            c = chunk;
            o = 0;
            cnt = 0;
            nx = -1;
            if(c >= 0) {
                cnt = a.chunk_cnt[c];
                nx = a.chunk_next[c];
                if(cnt > 0)
                    return;
                if(nx >= 0) { /* an empty chunk in the middle of a chain cannot happen; be safe */
                    chunk = nx;
                    continue;
                }
            }
            c = -1;
            if(then == -2)
                return;
            chunk = then;
            then = -2;
        }
    }
    __device__ __forceinline__ void step(const EvalArgs &a)
    {
        o += 64;
        if(o >= cnt)
            open(a, nx);
    }
};

template <bool POT>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(3, 3))) void grav_eval_kernel(const EvalArgs a)
{
    __shared__ double4 tab[SHQ_NGRAVTAB];
    for(int i = threadIdx.x; i < SHQ_NGRAVTAB; i += blockDim.x) {
        const int j = (i + 1 < SHQ_NGRAVTAB) ? i + 1 : i;
        const double f0 = a.tab_f[i], f1 = a.tab_f[j], p0 = a.tab_p[i], p1 = a.tab_p[j];
        tab[i] = make_double4(f0, f1 - f0, p0, p1 - p0);
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long wave = (long long) xcd_block(blockIdx.x, gridDim.x, a.xcdK) * WAVES + wv;
    const long long t = wave * 64 + lane;
    if(wave * 64 >= a.ntargets)
        return; /* whole wave: no barrier follows */
    const bool valid = t < a.ntargets;
    const long long tt = valid ? t : a.ntargets - 1;
    const long long pi = a.targets ? (long long) a.targets[tt] : tt;
    const double4 p = a.posm[pi];

    double rax = 0, ray = 0, raz = 0, rpot = 0;
    int rn = 0;
    const long long left_w = a.ntargets - wave * 64;
    const int ngroups = (int) (((left_w < 64 ? left_w : 64) + GS - 1) / GS);

    for(int g = 0; g < ngroups; g++) {
        const int l0 = g * GS;
        const long long left = left_w - l0;
        const int cnt = left < GS ? (int) left : GS;

        double tx[GS], ty[GS], tz[GS];
#pragma unroll
        for(int i = 0; i < GS; i++) {
            tx[i] = readlane_d(p.x, l0 + i);
            ty[i] = readlane_d(p.y, l0 + i);
            tz[i] = readlane_d(p.z, l0 + i);
        }
        /* A pair needs the periodic wrap only when its members lie on opposite sides of the box, and then it is within the
         * table's reach only if the target is that close to a face: groups away from the faces wrap nothing. */
        bool nearface = false;
#pragma unroll
        for(int i = 0; i < GS; i++)
            nearface = nearface || fmin(fmin(tx[i], ty[i]), tz[i]) < a.edge || fmax(fmax(tx[i], ty[i]), tz[i]) > a.Box - a.edge;
        const bool wrapall = shq_ballot(nearface) != 0ull;

        double acc[GS][4];
        int cntm[GS];
#pragma unroll
        for(int i = 0; i < GS; i++) {
            acc[i][0] = acc[i][1] = acc[i][2] = acc[i][3] = 0.0;
            cntm[i] = 0;
        }

        /* three rounds in flight: indices of round r + 2, source records of round r + 1, arithmetic of round r */
        const unsigned fullmask = (1u << cnt) - 1u;
        ListPos pa;
        pa.then = a.group_head[wave * 8 + g];
        pa.open(a, a.wave_head[wave]);
        int id1 = 0, id2 = 0;
        unsigned mk1 = 0, mk2 = 0;
        double4 s1 = make_double4(0, 0, 0, 0);
        bool live1 = false, live2 = false; /* wave-uniform: the round exists */
        auto load_idx = [&](const ListPos &q, int &id, unsigned &mk) {
            id = 0;
            mk = 0;
            if(q.o + lane < q.cnt) {
                id = a.pool_idx[(long long) q.c * CH + q.o + lane];
                mk = a.pool_msk[(long long) q.c * CH + q.o + lane] & fullmask; /* the shared list carries all eight bits */
            }
        };
        auto gather = [&](int id, unsigned mk) -> double4 {
            double4 s = make_double4(0, 0, 0, 0);
            if(mk != 0u)
                s = id >= 0 ? *reinterpret_cast<const double4 *>(&a.A[id]) : a.posm_leaf[-1 - id];
            return s;
        };
        /* prologue */
        live1 = pa.c >= 0;
        if(live1) {
            load_idx(pa, id1, mk1);
            pa.step(a);
        }
        live2 = pa.c >= 0;
        if(live2) {
            load_idx(pa, id2, mk2);
            pa.step(a);
        }
        if(live1)
            s1 = gather(id1, mk1);
        while(live1) {
            /* round r = (s1, mk1); r + 1 = (id2, mk2) still without its records; r + 2 to be fetched */
            const double4 s = s1;
            const unsigned mk = mk1;
            int id3 = 0;
            unsigned mk3 = 0;
            const bool live3 = pa.c >= 0;
            if(live3) {
                load_idx(pa, id3, mk3);
                pa.step(a);
            }
            if(live2)
                s1 = gather(id2, mk2);
            mk1 = mk2;
            live1 = live2;
            id2 = id3;
            mk2 = mk3;
            live2 = live3;
#pragma unroll
            for(int i = 0; i < GS; i++) {
                if(mk & (1u << i)) {
                    double dx = s.x - tx[i], dy = s.y - ty[i], dz = s.z - tz[i];
                    if(wrapall) {
                        dx = wrapd_g(dx, a.Box, a.invBox);
                        dy = wrapd_g(dy, a.Box, a.invBox);
                        dz = wrapd_g(dz, a.Box, a.invBox);
                    }
                    const double r2 = dx * dx + dy * dy + dz * dz;
                    apply_accn_g<POT>(tab, dx, dy, dz, r2, s.w, a, acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
                    cntm[i]++;
                }
                __builtin_amdgcn_sched_barrier(0); /* one member at a time: interleaving the eight chains costs 100+ registers */
            }
        }

        /* ---- the lanes' partial sums of every member, added across the wave; member i of the group lives in lane l0 + i */
#pragma unroll
        for(int i = 0; i < GS; i++) {
            if(i < cnt) {
                double v0 = acc[i][0], v1 = acc[i][1], v2 = acc[i][2], v3 = acc[i][3];
                int c = cntm[i];
#pragma unroll
                for(int off = 32; off > 0; off >>= 1) {
                    v0 += __shfl_xor(v0, off);
                    v1 += __shfl_xor(v1, off);
                    v2 += __shfl_xor(v2, off);
                    if(POT)
                        v3 += __shfl_xor(v3, off);
                    c += __shfl_xor(c, off);
                }
                if(lane == l0 + i) {
                    rax = v0;
                    ray = v1;
                    raz = v2;
                    rpot = v3;
                    rn = c;
                }
            }
        }
    }

    if(valid) {
        a.acc[3 * pi + 0] = rax;
        a.acc[3 * pi + 1] = ray;
        a.acc[3 * pi + 2] = raz;
        if(POT)
            a.pot[pi] = rpot;
        a.nint[pi] = rn;
    }
    long long mn = valid ? rn : 0x7fffffffffffll, mx = valid ? rn : 0, sm = valid ? rn : 0;
    for(int off = 32; off > 0; off >>= 1) {
        const long long o1 = __shfl_xor(mn, off), o2 = __shfl_xor(mx, off), o3 = __shfl_xor(sm, off);
        mn = o1 < mn ? o1 : mn;
        mx = o2 > mx ? o2 : mx;
        sm += o3;
    }
    if(lane == 0 && a.stats) {
        atomicAdd(&a.stats->ninteractions, (unsigned long long) sm);
        atomicMin(&a.stats->min_int, mn);
        atomicMax(&a.stats->max_int, mx);
    }
}

/* the children of every internal node of the pre-order pool (first child = i + 1, the next ones along the sibling links) */
/* the 64-byte T1 record of every node of the pre-order pool: f32 geometry + the children (first child = i + 1, the next ones
 * along the sibling links) or, for a leaf, its particle slots */
__global__ void node_t1_kernel(const NodeG *__restrict__ G, long long n, float4 *F)
{
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n)
        return;
    int kid[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
    const int type = G[i].type, child = G[i].child, sibling = G[i].sibling;
    if(type == SHQ_NODE_NODE_TYPE && child >= 0) {
        int ch = child;
        for(int j = 0; j < 8 && ch >= 0 && ch != sibling; j++) {
            kid[j] = ch;
            ch = G[ch].sibling;
        }
    } else if(type == SHQ_PARTICLE_NODE_TYPE) {
        kid[0] = child;
        kid[1] = G[i].count;
        kid[7] = -2;
    } else
        kid[7] = -3;
    F[4 * i] = make_float4((float) G[i].cofm[0], (float) G[i].cofm[1], (float) G[i].cofm[2], (float) G[i].mass);
    F[4 * i + 1] = make_float4((float) G[i].center[0], (float) G[i].center[1], (float) G[i].center[2], (float) G[i].len);
    *reinterpret_cast<int4 *>(&F[4 * i + 2]) = make_int4(kid[0], kid[1], kid[2], kid[3]);
    *reinterpret_cast<int4 *>(&F[4 * i + 3]) = make_int4(kid[4], kid[5], kid[6], kid[7]);
}

__global__ void add_tests_kernel(GravStatsDev *s, const unsigned long long *ntests) { s->nvisited += *ntests; }

} // namespace

int shq_launch_grav_walk_group(shq_context *ctx, const shq_grav_params *p, const int32_t *d_active, int64_t ntargets, int update_potential,
                               int64_t first)
{
    SHQ_CHECK(first >= 0 && (first == 0 || !d_active) && first + ntargets <= ctx->numpart, SHQ_ERR_INVALID,
              "grav walk: bad target range [%ld, +%ld)", (long) first, (long) ntargets);
    SHQ_CHECK(ctx->have_parts && ctx->have_tree, SHQ_ERR_STATE, "grav walk: particles and tree must be uploaded first");
    SHQ_CHECK(p->ForceSoftening > 0 && p->cellsize > 0 && p->dx > 0, SHQ_ERR_INVALID, "grav params: softening/cellsize/dx must be > 0");
    SHQ_CHECK(ntargets >= 0 && ntargets <= ctx->numpart, SHQ_ERR_INVALID, "grav walk: ntargets %ld out of range", (long) ntargets);
    SHQ_TRY(ctx->gravtab.reserve(2 * SHQ_NGRAVTAB));
    SHQ_TRY(ctx->gstats.reserve(1));
    SHQ_TRY(ctx->walk_counters.reserve(4));
    SHQ_HIP(hipMemcpyAsync(ctx->gravtab.ptr, p->shortrange_table, sizeof(float) * SHQ_NGRAVTAB, hipMemcpyHostToDevice, ctx->stream));
    SHQ_HIP(hipMemcpyAsync(ctx->gravtab.ptr + SHQ_NGRAVTAB, p->shortrange_table_potential, sizeof(float) * SHQ_NGRAVTAB,
                           hipMemcpyHostToDevice, ctx->stream));
    if(first == 0)
        shq_launch_stats_init(ctx);
    if(ntargets == 0)
        return SHQ_OK;
    shq_fill_node_walk_params(ctx, p);
    const long long nn = ctx->numnodes;
    if(!ctx->have_group_aux) { /* per tree: the T1 records */
        SHQ_TRY(ctx->nodeF.reserve(4 * (size_t) (nn + 1)));
        node_t1_kernel<<<dim3((unsigned) ((nn + 255) / 256)), dim3(256), 0, ctx->stream>>>(ctx->nodeG.ptr, nn, ctx->nodeF.ptr);
        SHQ_HIP(hipGetLastError());
        ctx->have_group_aux = true;
    }
    /* ---- the list pool: SHQ_WALK_POOL_MB of chunks (default 1/8 of the card), reused by every batch of targets */
    if(ctx->walk_pool_chunks == 0) {
        size_t free_b = 0, total_b = 0;
        SHQ_HIP(hipMemGetInfo(&free_b, &total_b));
        size_t want = total_b / 8;
        if(const char *v = getenv("SHQ_WALK_POOL_MB"))
            want = (size_t) std::max(64.0, atof(v)) << 20;
        want = std::min(want, free_b / 2);
        const size_t per_chunk = (size_t) CH * 5 + 8;
        size_t nch = std::min<size_t>(want / per_chunk, (size_t) 1 << 30);
        SHQ_CHECK(nch >= 4096, SHQ_ERR_NOMEM, "grav walk: not enough free device memory for the interaction-list pool");
        SHQ_TRY(ctx->walk_pool_idx.reserve(nch * CH));
        SHQ_TRY(ctx->walk_pool_msk.reserve(nch * CH));
        SHQ_TRY(ctx->walk_chunk_cnt.reserve(nch));
        SHQ_TRY(ctx->walk_chunk_next.reserve(nch));
        ctx->walk_pool_chunks = (int64_t) nch;
    }
    const int64_t pool = ctx->walk_pool_chunks;

    ListArgs la;
    la.G = ctx->nodeG.ptr;
    la.F = ctx->nodeF.ptr;
    la.root = ctx->root;
    la.Box = p->BoxSize;
    la.invBox = 1.0 / p->BoxSize;
    la.rcut = p->Rcut;
    la.rcut2 = p->Rcut * p->Rcut;
    la.bh2 = p->BHOpeningAngle2;
    la.errtol = p->ErrTolForceAcc;
    la.xcdK = (unsigned) ctx->xcd_k;
    la.pool_idx = ctx->walk_pool_idx.ptr;
    la.pool_msk = ctx->walk_pool_msk.ptr;
    la.chunk_cnt = ctx->walk_chunk_cnt.ptr;
    la.chunk_next = ctx->walk_chunk_next.ptr;
    la.nchunks = (int) pool;
    la.counters = ctx->walk_counters.ptr;
    la.ntests = reinterpret_cast<unsigned long long *>(ctx->walk_counters.ptr + 2);
    la.dbg = ctx->walk_stats == 2 ? ctx->gstats.ptr->hist_visit : nullptr;

    EvalArgs ea;
    ea.A = ctx->nodeA.ptr;
    ea.posm_leaf = ctx->posm_leaf.ptr;
    ea.stats = ctx->gstats.ptr;
    ea.Box = p->BoxSize;
    ea.invBox = 1.0 / p->BoxSize;
    const double h = p->ForceSoftening;
    ea.h2 = h * h;
    ea.h_inv = 1.0 / h;
    ea.h3_inv = 1.0 / h / h / h;
    ea.inv_celldx = 1.0 / (p->cellsize * p->dx);
    /* the table ends at (SHQ_NGRAVTAB - 1) * dx cells; with a table reaching over half the box every group wraps */
    ea.edge = 1.001 * SHQ_NGRAVTAB * p->dx * p->cellsize;
    if(ea.edge >= 0.499 * p->BoxSize)
        ea.edge = 2 * p->BoxSize;
    ea.xcdK = (unsigned) ctx->xcd_k;
    ea.pool_idx = la.pool_idx;
    ea.pool_msk = la.pool_msk;
    ea.chunk_cnt = la.chunk_cnt;
    ea.chunk_next = la.chunk_next;
    ea.tab_f = ctx->gravtab.ptr;
    ea.tab_p = ctx->gravtab.ptr + SHQ_NGRAVTAB;

    /* ---- batches of targets sized so that their lists fit the pool: chunks per target from the last batch (first call: a
     * cautious guess); a batch that does not fit is cut in half and listed again */
    SHQ_HIP(hipEventRecord(ctx->ev_begin[SHQ_NTIMERS - 1], ctx->stream));
    int64_t done = 0;
    double cpt = ctx->walk_chunks_per_target > 0 ? ctx->walk_chunks_per_target : 8.0 / GS;
    while(done < ntargets) {
        int64_t nb = (int64_t) (0.8 * (double) pool / (1.25 * cpt));
        nb = std::max<int64_t>(64, std::min<int64_t>(nb, ntargets - done)) / 64 * 64;
        if(nb == 0 || nb > ntargets - done)
            nb = ntargets - done;
        for(;;) {
            const long long nwaves = (nb + 63) / 64;
            const long long blocks = (nwaves + WAVES - 1) / WAVES;
            SHQ_CHECK(blocks < (1ll << 31), SHQ_ERR_INVALID, "grav walk: too many targets for one launch");
            SHQ_TRY(ctx->walk_group_head.reserve((size_t) nwaves * 9));
            la.group_head = ctx->walk_group_head.ptr;
            la.wave_head = ctx->walk_group_head.ptr + nwaves * 8;
            la.posm = ctx->posm.ptr + first;
            la.oldacc = ctx->oldacc.ptr + first;
            la.targets = d_active ? d_active + done : nullptr;
            if(!d_active) {
                la.posm += done;
                la.oldacc += done;
            }
            la.ntargets = nb;
            SHQ_HIP(hipMemsetAsync(ctx->walk_counters.ptr, 0, sizeof(int) * 4, ctx->stream));
            grav_list_kernel<<<dim3((unsigned) blocks), dim3(64 * WAVES), 0, ctx->stream>>>(la, p->TreeUseBH);
            SHQ_HIP(hipGetLastError());
            int hc[4] = {0, 0, 0, 0};
            SHQ_HIP(hipMemcpyAsync(hc, ctx->walk_counters.ptr, sizeof(hc), hipMemcpyDeviceToHost, ctx->stream));
            SHQ_HIP(hipStreamSynchronize(ctx->stream));
            SHQ_CHECK((hc[1] & 1) == 0, SHQ_ERR_DEVICE, "grav walk: node stack overflow (tree deeper than 32 levels below the soft limit)");
            if(hc[1] & 2) { /* pool exhausted: half the batch */
                SHQ_CHECK(nb > 64, SHQ_ERR_NOMEM, "grav walk: the interaction lists of 64 targets do not fit the pool (%ld chunks): raise SHQ_WALK_POOL_MB",
                          (long) pool);
                cpt = std::max(cpt * 2, 1.0 / GS);
                nb = std::max<int64_t>(64, nb / 2 / 64 * 64);
                continue;
            }
            cpt = (double) hc[0] / (double) nb;
            ctx->walk_chunks_per_target = cpt;
            add_tests_kernel<<<1, 1, 0, ctx->stream>>>(ctx->gstats.ptr, la.ntests);
            ea.group_head = la.group_head;
            ea.wave_head = la.wave_head;
            ea.posm = la.posm;
            ea.targets = la.targets;
            ea.ntargets = nb;
            ea.acc = ctx->acc.ptr + 3 * first;
            ea.pot = ctx->pot.ptr + first;
            ea.nint = ctx->nint.ptr + first;
            if(!d_active) {
                ea.acc += 3 * done;
                ea.pot += done;
                ea.nint += done;
            }
            if(update_potential)
                grav_eval_kernel<true><<<dim3((unsigned) blocks), dim3(64 * WAVES), 0, ctx->stream>>>(ea);
            else
                grav_eval_kernel<false><<<dim3((unsigned) blocks), dim3(64 * WAVES), 0, ctx->stream>>>(ea);
            SHQ_HIP(hipGetLastError());
            break;
        }
        done += nb;
    }
    SHQ_HIP(hipEventRecord(ctx->ev_end[SHQ_NTIMERS - 1], ctx->stream));
    return SHQ_OK;
}
