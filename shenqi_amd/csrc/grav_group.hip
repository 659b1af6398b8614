/* grav_group.hip — short-range tree walk, source-parallel flavour (walk_mode SHQ_WALK_GROUP) for gfx950.
 *
 * Same result as grav_walk.hip: every target meets exactly the interaction set of its own reference walk
 * (GravLocalTreeWalk::visit, libgadget/gravshort2.hpp:227-322, with shall_we_discard_node / shall_we_open_node :152-193 and
 * apply_accn :326-358) — interaction counts equal the reference's as integers; only the order in which a target's
 * contributions are summed differs (forces agree to rounding, ~1e-15 relative).
 *
 * The mapping is the transpose of grav_walk.hip's: LANES ARE SOURCES, not targets.  A wave takes 64 consecutive targets as 8
 * groups of 8 and walks, for one group at a time, the UNION of its members' reference walks; every pending node carries the
 * 8-bit mask of the members whose own walk reaches it.
 *   - T1 rounds: up to 64 (node, mask) entries are popped from a per-wave LDS stack, one per lane.  Each lane tests its node
 *     against the group's bounding box: from the nearest and the farthest point of the box it is usually certain that every
 *     masked member discards the node, or that none discards and none opens it (=> one source for all of them), or that all
 *     open it (=> its children / its particles inherit the mask).  The comparisons carry a relative margin far above
 *     rounding, so a certain answer is the answer of every member's own test;
 *   - T2 rounds: the nodes T1 could not settle (~15 %) wait in a small LDS queue and are tested 64 at a time member by
 *     member, with the expressions of grav_walk.hip: per-member accept and open masks;
 *   - evaluation rounds: whenever 64 sources (accepted monopoles or particles of opened leaves, each with its member mask)
 *     are pending, every lane takes one and applies it to the members in its mask, whose coordinates are wave-uniform; the
 *     lanes' partial sums are added across the wave once per group.
 * A node test thus costs a lane-slot instead of a wave-round, and the interaction arithmetic runs at the fill of the
 * group's union list (~70 %) instead of the 44 % of the 64-target union walk.
 */
#include "common.hpp"

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
constexpr int LIST_CAP = 128;    /* per-wave source ring: < 64 pending + <= 64 appended per round */
constexpr int AMB_CAP = 128;     /* per-wave pile of nodes waiting for the per-member tests: < 64 + <= 64 per T1 round */

enum { ACT_EVAL, ACT_LEAF, ACT_T2, ACT_T1 };

struct GroupArgs {
    const NodeG *G;            /* merged node record (grav_walk.hip's): cofm, mass, centre, len, links, per-node products */
    const int4 *K;             /* [2 * node]: the (up to 8) children of an internal node, -1 padded */
    const double4 *posm;       /* by particle index */
    const double4 *posm_leaf;  /* leaf order */
    const double *oldacc;
    const int32_t *targets;    /* may be null */
    double *acc;
    double *pot;
    int32_t *nint;
    GravStatsDev *stats;
    int *errflag;
    long long ntargets;
    int root;
    double Box, invBox, halfBox;
    double rcut, rcut2;
    double h2, h_inv, h3_inv;
    double inv_celldx;
    double errtol;
    double wraplim;            /* group half-extent from which every pair is wrapped individually */
    unsigned xcdK;
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

__device__ __forceinline__ double rsqrt_nr(double x)
{
    const double y0 = __builtin_amdgcn_rsq(x);
    const double e = fma(-x * y0, y0, 1.0);
    return fma(y0 * e, fma(e, 0.375, 0.5), y0);
}

/* apply_accn (gravshort2.hpp:326-358) + apply_short_range_window (gravity.h:48-60), as in grav_walk.hip */
template <bool POT>
__device__ __forceinline__ void apply_accn_g(const double4 *__restrict__ tab, double dx, double dy, double dz, double r2, double mass,
                                             const GroupArgs &a, double &ax, double &ay, double &az, double &pot)
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

/* shift a source by a box period to the image nearest the group's centre */
__device__ __forceinline__ void to_nearest_image(double4 &s, double cx, double cy, double cz, const GroupArgs &a)
{
    const double ex = s.x - cx, ey = s.y - cy, ez = s.z - cz;
    s.x += (ex > a.halfBox) ? -a.Box : ((ex < -a.halfBox) ? a.Box : 0.0);
    s.y += (ey > a.halfBox) ? -a.Box : ((ey < -a.halfBox) ? a.Box : 0.0);
    s.z += (ez > a.halfBox) ? -a.Box : ((ez < -a.halfBox) ? a.Box : 0.0);
}

template <bool POT, bool BH>
__global__ __launch_bounds__(64 * WAVES) void grav_walk_group_kernel(const GroupArgs a)
{
    __shared__ double4 tab[SHQ_NGRAVTAB];
    __shared__ int stack_s[WAVES][STACK_CAP];
    __shared__ unsigned char stackm_s[WAVES][STACK_CAP];
    __shared__ double4 list_s[WAVES][LIST_CAP];
    __shared__ int lmask_s[WAVES][LIST_CAP];
    __shared__ int2 amb_s[WAVES][AMB_CAP];
    for(int i = threadIdx.x; i < SHQ_NGRAVTAB; i += blockDim.x) {
        const int j = (i + 1 < SHQ_NGRAVTAB) ? i + 1 : i;
        const double f0 = a.tab_f[i], f1 = a.tab_f[j], p0 = a.tab_p[i], p1 = a.tab_p[j];
        tab[i] = make_double4(f0, f1 - f0, p0, p1 - p0);
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); /* wave-uniform for the compiler too */
    int *__restrict__ stk = stack_s[wv];
    unsigned char *__restrict__ stkm = stackm_s[wv];
    double4 *__restrict__ lst = list_s[wv];
    int *__restrict__ lmk = lmask_s[wv];
    int2 *__restrict__ amb = amb_s[wv];
    const long long wave = (long long) xcd_block(blockIdx.x, gridDim.x, a.xcdK) * WAVES + wv;
    const long long t = wave * 64 + lane;
    if(wave * 64 >= a.ntargets)
        return; /* whole wave: no barrier follows */
    const bool valid = t < a.ntargets;
    const long long tt = valid ? t : a.ntargets - 1; /* clones of the last target pad the last group: they change no box */
    const long long pi = a.targets ? (long long) a.targets[tt] : tt;
    const double4 p = a.posm[pi];
    const double aold = a.errtol * a.oldacc[pi];

    double rax = 0, ray = 0, raz = 0, rpot = 0;
    int rn = 0;
    unsigned long long ntests = 0;
    const int ngroups = (int) ((((a.ntargets - wave * 64) < 64 ? (a.ntargets - wave * 64) : 64) + GS - 1) / GS);

    for(int g = 0; g < ngroups; g++) {
        const int l0 = g * GS;
        /* members actually present (the rest of the slots repeat the last one and are masked out everywhere) */
        const long long left = a.ntargets - (wave * 64 + l0);
        const int cnt = left < GS ? (int) left : GS;
        const unsigned fullmask = (1u << cnt) - 1u;

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
        /* half extents, inflated far beyond the rounding of the lines above: T1 may only be certain when it is right */
        const double hpad = 1e-13 * a.Box;
        const double hx = 0.5 * (ux - lx) + hpad, hy = 0.5 * (uy - ly) + hpad, hz = 0.5 * (uz - lz) + hpad;
        /* Sources are stored shifted by a box period to the image nearest the group's centre.  While the group is small, a
         * pair for which that is not the nearest image is further apart than the table reaches in either image and adds
         * nothing; a group wider than that wraps every pair. */
        const bool needwrap = shq_ballot(fmax(fmax(hx, hy), hz) >= a.wraplim) != 0ull;

        double acc[GS][4];
        int cntm[GS];
#pragma unroll
        for(int i = 0; i < GS; i++) {
            acc[i][0] = acc[i][1] = acc[i][2] = acc[i][3] = 0.0;
            cntm[i] = 0;
        }

        int S = 1, list_n = 0, head = 0, amb_n = 0;
        int rem = 0, slot = 0;
        unsigned lmask = 0;
        if(lane == 0) {
            stk[0] = a.root;
            stkm[0] = (unsigned char) fullmask;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        for(;;) {
            const bool anyrem = shq_ballot(rem > 0) != 0ull;
            int action;
            if(list_n >= 64)
                action = ACT_EVAL;
            else if(anyrem)
                action = ACT_LEAF;
            else if(amb_n >= 64 || (amb_n > 0 && S + 8 * amb_n >= STACK_SOFT))
                action = ACT_T2;
            else if(S > 0)
                action = ACT_T1;
            else if(amb_n > 0)
                action = ACT_T2;
            else if(list_n > 0)
                action = ACT_EVAL;
            else
                break;

            if(action == ACT_EVAL) {
                /* ---- evaluation round: one source per lane against the members in its mask */
                const int m = list_n < 64 ? list_n : 64;
                double4 s = make_double4(cx, cy, cz, 0.0);
                unsigned mk = 0;
                if(lane < m) {
                    s = lst[(head + lane) & (LIST_CAP - 1)];
                    mk = (unsigned) lmk[(head + lane) & (LIST_CAP - 1)];
                }
                head = (head + m) & (LIST_CAP - 1);
                list_n -= m;
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for(int i = 0; i < GS; i++) {
                    if(mk & (1u << i)) {
                        double dx = s.x - tx[i], dy = s.y - ty[i], dz = s.z - tz[i];
                        if(needwrap) {
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
            } else if(action == ACT_LEAF) {
                /* ---- one particle of every opened leaf joins the sources (gravshort2.hpp:290-304: all of them are evaluated) */
                const bool has = rem > 0;
                double4 q = make_double4(0, 0, 0, 0);
                if(has) {
                    q = a.posm_leaf[slot];
                    slot++;
                    rem--;
                    to_nearest_image(q, cx, cy, cz, a);
                }
                const unsigned long long msk = shq_ballot(has);
                if(has) {
                    const int at = (head + list_n + mbcnt64(msk)) & (LIST_CAP - 1);
                    lst[at] = q;
                    lmk[at] = (int) lmask;
                }
                list_n += __popcll(msk);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            } else {
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
                __builtin_amdgcn_wave_barrier();
                const int node = ent.x;
                const unsigned mask = (unsigned) ent.y;
                unsigned accm = 0, openm = 0;
                bool ambiguous = false;
                NodeG nd;
                if(on)
                    nd = a.G[node];
                else {
                    nd.cofm[0] = nd.cofm[1] = nd.cofm[2] = nd.mass = 0;
                    nd.center[0] = nd.center[1] = nd.center[2] = nd.len = 0;
                    nd.sibling = nd.child = -1;
                    nd.type = SHQ_PSEUDO_NODE_TYPE;
                    nd.count = 0;
                    nd.bhlim = nd.mlen2 = nd.inside = nd.halflen = nd.wraplim = nd.rcuthl = 0;
                }
                if(action == ACT_T1) {
                    if(on) {
                        /* distances from the nearest and the farthest point of the group's box */
                        const double d0 = fabs(wrapd_g(nd.cofm[0] - cx, a.Box, a.invBox)), d1 = fabs(wrapd_g(nd.cofm[1] - cy, a.Box, a.invBox)),
                                     d2 = fabs(wrapd_g(nd.cofm[2] - cz, a.Box, a.invBox));
                        const double q0 = fabs(wrapd_g(nd.center[0] - cx, a.Box, a.invBox)), q1 = fabs(wrapd_g(nd.center[1] - cy, a.Box, a.invBox)),
                                     q2 = fabs(wrapd_g(nd.center[2] - cz, a.Box, a.invBox));
                        const double n0 = fmax(d0 - hx, 0.0), n1 = fmax(d1 - hy, 0.0), n2 = fmax(d2 - hz, 0.0);
                        const double f0 = d0 + hx, f1 = d1 + hy, f2 = d2 + hz;
                        const double r2min = n0 * n0 + n1 * n1 + n2 * n2, r2max = f0 * f0 + f1 * f1 + f2 * f2;
                        const double qn0 = fmax(q0 - hx, 0.0), qn1 = fmax(q1 - hy, 0.0), qn2 = fmax(q2 - hz, 0.0);
                        const double qf0 = q0 + hx, qf1 = q1 + hy, qf2 = q2 + hz;
                        const double cnear = fmax(fmax(qn0, qn1), qn2), cfar = fmax(fmax(qf0, qf1), qf2);
                        const double up = 1.0 + 1e-11, dn = 1.0 - 1e-11;
                        const bool alldiscard = (r2min > a.rcut2 * up) && (cnear > nd.rcuthl * up);
                        const bool nonediscards = (r2max < a.rcut2 * dn) || (cfar < nd.rcuthl * dn);
                        const bool allopen = (!BH && (nd.mlen2 * dn > r2max * r2max * amax)) || (r2max * up < nd.bhlim) || (cfar < nd.inside * dn);
                        const bool noneopen = (BH || (nd.mlen2 * up < r2min * r2min * amin)) && (r2min > nd.bhlim * up) && (cnear > nd.inside * up);
                        if(!alldiscard) {
                            if(nonediscards && noneopen)
                                accm = mask;
                            else if(nonediscards && allopen)
                                openm = mask;
                            else
                                ambiguous = true;
                        }
                    }
                    /* unsettled entries queue for the per-member tests */
                    const unsigned long long am = shq_ballot(ambiguous);
                    if(ambiguous)
                        amb[amb_n + mbcnt64(am)] = ent;
                    amb_n += __popcll(am);
                } else {
                    /* per-member tests with grav_walk.hip's expressions (gravshort2.hpp:152-193) */
                    const bool anywrap = shq_ballot(on && (fmax(fmax(fabs(nd.center[0] - cx) + hx, fabs(nd.center[1] - cy) + hy),
                                                                  fabs(nd.center[2] - cz) + hz) > nd.wraplim)) != 0ull;
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
                /* accepted monopoles join the sources, shifted to the image nearest the group's centre */
                {
                    const bool acc_any = accm != 0;
                    const unsigned long long msk = shq_ballot(acc_any);
                    if(acc_any) {
                        double4 s = make_double4(nd.cofm[0], nd.cofm[1], nd.cofm[2], nd.mass);
                        to_nearest_image(s, cx, cy, cz, a);
                        const int at = (head + list_n + mbcnt64(msk)) & (LIST_CAP - 1);
                        lst[at] = s;
                        lmk[at] = (int) accm;
                    }
                    list_n += __popcll(msk);
                }
                /* opened leaves queue their particles; opened internal nodes push their children with the opening members' mask */
                const bool openleaf = openm != 0 && nd.type == SHQ_PARTICLE_NODE_TYPE;
                const bool openint = openm != 0 && nd.type == SHQ_NODE_NODE_TYPE;
                if(openleaf) {
                    rem = nd.count;
                    slot = nd.child;
                    lmask = openm;
                }
                int4 k0 = make_int4(-1, -1, -1, -1), k1 = k0;
                if(openint) {
                    k0 = a.K[2 * (long long) node];
                    k1 = a.K[2 * (long long) node + 1];
                }
                const int kid[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
#pragma unroll
                for(int j = 0; j < 8; j++) {
                    const bool pj = kid[j] >= 0;
                    const unsigned long long msk = shq_ballot(pj);
                    if(msk == 0ull)
                        break;
                    const int at = S + mbcnt64(msk);
                    if(pj && at < STACK_CAP) {
                        stk[at] = kid[j];
                        stkm[at] = (unsigned char) openm;
                    }
                    S += __popcll(msk);
                }
                if(S > STACK_CAP) { /* cannot happen for trees of fewer than 32 levels: refuse rather than walk a truncated stack */
                    if(lane == 0)
                        atomicOr(a.errflag, 1);
                    S = 0;
                    rem = 0;
                    list_n = 0;
                    amb_n = 0;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
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
        atomicAdd(&a.stats->nvisited, ntests);
    }
}

/* the children of every internal node of the pre-order pool (first child = i + 1, the next ones along the sibling links) */
__global__ void node_children_kernel(const NodeG *__restrict__ G, long long n, int4 *K)
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
    }
    K[2 * i] = make_int4(kid[0], kid[1], kid[2], kid[3]);
    K[2 * i + 1] = make_int4(kid[4], kid[5], kid[6], kid[7]);
}

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
    SHQ_TRY(ctx->walk_err.reserve(1));
    SHQ_HIP(hipMemcpyAsync(ctx->gravtab.ptr, p->shortrange_table, sizeof(float) * SHQ_NGRAVTAB, hipMemcpyHostToDevice, ctx->stream));
    SHQ_HIP(hipMemcpyAsync(ctx->gravtab.ptr + SHQ_NGRAVTAB, p->shortrange_table_potential, sizeof(float) * SHQ_NGRAVTAB,
                           hipMemcpyHostToDevice, ctx->stream));
    if(first == 0) {
        shq_launch_stats_init(ctx);
        SHQ_HIP(hipMemsetAsync(ctx->walk_err.ptr, 0, sizeof(int), ctx->stream));
    }
    if(ntargets == 0)
        return SHQ_OK;
    shq_fill_node_walk_params(ctx, p);
    const long long nn = ctx->numnodes;
    if(!ctx->have_group_aux) { /* per tree: the children lists */
        SHQ_TRY(ctx->nodeK.reserve(2 * (size_t) (nn + 1)));
        node_children_kernel<<<dim3((unsigned) ((nn + 255) / 256)), dim3(256), 0, ctx->stream>>>(ctx->nodeG.ptr, nn, ctx->nodeK.ptr);
        SHQ_HIP(hipGetLastError());
        ctx->have_group_aux = true;
    }
    GroupArgs a;
    a.G = ctx->nodeG.ptr;
    a.K = ctx->nodeK.ptr;
    a.posm = ctx->posm.ptr + first;
    a.posm_leaf = ctx->posm_leaf.ptr;
    a.oldacc = ctx->oldacc.ptr + first;
    a.targets = d_active;
    a.acc = ctx->acc.ptr + 3 * first;
    a.pot = ctx->pot.ptr + first;
    a.nint = ctx->nint.ptr + first;
    a.stats = ctx->gstats.ptr;
    a.errflag = ctx->walk_err.ptr;
    a.ntargets = ntargets;
    a.root = ctx->root;
    a.Box = p->BoxSize;
    a.invBox = 1.0 / p->BoxSize;
    a.halfBox = 0.5 * p->BoxSize;
    a.rcut = p->Rcut;
    a.rcut2 = p->Rcut * p->Rcut;
    const double h = p->ForceSoftening;
    a.h2 = h * h;
    a.h_inv = 1.0 / h;
    a.h3_inv = 1.0 / h / h / h;
    a.inv_celldx = 1.0 / (p->cellsize * p->dx);
    a.errtol = p->ErrTolForceAcc;
    /* the table ends at (SHQ_NGRAVTAB - 1) * dx cells: a group narrower than Box / 2 minus that (with a margin) never needs a
     * per-pair wrap */
    a.wraplim = 0.5 * p->BoxSize - 1.001 * SHQ_NGRAVTAB * p->dx * p->cellsize;
    a.xcdK = (unsigned) ctx->xcd_k;
    a.tab_f = ctx->gravtab.ptr;
    a.tab_p = ctx->gravtab.ptr + SHQ_NGRAVTAB;

    const long long nwaves = (ntargets + 63) / 64;
    const long long blocks = (nwaves + WAVES - 1) / WAVES;
    SHQ_CHECK(blocks < (1ll << 31), SHQ_ERR_INVALID, "grav walk: too many targets for one launch");
    const dim3 grid((unsigned) blocks), block(64 * WAVES);
    SHQ_HIP(hipEventRecord(ctx->ev_begin[SHQ_NTIMERS - 1], ctx->stream));
    if(update_potential) {
        if(p->TreeUseBH)
            grav_walk_group_kernel<true, true><<<grid, block, 0, ctx->stream>>>(a);
        else
            grav_walk_group_kernel<true, false><<<grid, block, 0, ctx->stream>>>(a);
    } else {
        if(p->TreeUseBH)
            grav_walk_group_kernel<false, true><<<grid, block, 0, ctx->stream>>>(a);
        else
            grav_walk_group_kernel<false, false><<<grid, block, 0, ctx->stream>>>(a);
    }
    SHQ_HIP(hipGetLastError());
    SHQ_HIP(hipEventRecord(ctx->ev_end[SHQ_NTIMERS - 1], ctx->stream));
    return SHQ_OK;
}
