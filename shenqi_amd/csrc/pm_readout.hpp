/* pm_readout.hpp — CIC readout of the potential mesh with the force from 4-point differencing (gravpm.cpp:489-500 with the
 * real-space image of the reference's k-space difference kernel, see pm.hip), shared by pm_readout_kernel (pm.hip) and by the
 * tree walk's task prologue (grav_walk.hip, shq_treepm_step), so that both produce the same bits. */
#pragma once
#include "common.hpp"
#ifndef SHQ_READOUT_FENCE
#define SHQ_READOUT_FENCE 1 /* load grouping of the lean readout's common path, see pm_readout_corner: 2 (a stencil at a time, 64 VGPRs) measured no faster than 1 (62) */
#endif

__device__ __forceinline__ int wrapi(int i, int N) { return i >= N ? i - N : (i < 0 ? i + N : i); }
/* x-plane index into the (possibly slab-local) mesh: global plane gx -> (gx - xshift) mod N.
 * xshift = 0 for the full periodic mesh; for a slab it is the global index of local plane 0. */
__device__ __forceinline__ int xloc(int gx, int xshift, int N)
{
    int v = (gx - xshift) % N;
    return v < 0 ? v + N : v;
}

/* CIC cell + residual: petapm.cpp:1147-1160 */
__device__ __forceinline__ void cic_setup(double p, double cell, int N, int &ic, double &res)
{
    const double tmp = p / cell; /* a true divide, as petapm.cpp:1148, so cells/weights match bit for bit */
    const double fl = floor(tmp);
    res = tmp - fl;
    int i = (int) fl;
    i %= N;
    if(i < 0)
        i += N;
    ic = i;
}

/* One corner c = (a, b, e) of the CIC cube: weight, potential and the three differenced force components, accumulated in the
 * order c = 0..7 — THE operation order of the readout (pm_readout_kernel's paired-load path fetches the same values and applies
 * the same operations).  M(dx, dy, dz) returns the mesh value at cell offsets (dx, dy, dz) in -2..3 from the particle's base cell.
 * FENCE: the loads are kept apart for callers on a register budget (1: two in flight; 2: a stencil's four — with the potential,
 * five — instead of thirteen); the arithmetic is the same. */
template <int FENCE, typename MeshAt>
__device__ __forceinline__ void pm_readout_corner(int c, const double res[3], double ffac, MeshAt M, double &g0, double &g1, double &g2,
                                                  double &gp)
{
    /* no fp contraction here, nor in pm_readout_kernel's paired-load path: the "same bits on every route" of the readout must not
     * depend on which of ffac * (c1 d1 - c2 d2) and w * f the compiler happens to fuse in three differently shaped code paths */
#pragma clang fp contract(off)
    const double c1 = 2.0 / 3.0, c2 = 1.0 / 12.0;
    const int a = c & 1, b = (c >> 1) & 1, e = (c >> 2) & 1;
    /* bit ? r : 1 - r as one fma with wave-uniform operands (exactly r, exactly 1 - r): with a run-time corner index the select form
     * keeps res AND 1 - res in registers for the whole loop */
    auto side = [](int bit, double r) { return fma(bit ? 1.0 : -1.0, r, bit ? 0.0 : 1.0); };
    const double w = side(a, res[0]) * side(b, res[1]) * side(e, res[2]);
    /* (FENCE: at most three loads in flight) */
#define SHQ_PM_DIFF(P1, M1, P2, M2)                                \
    [&]() {                                                        \
        double d1 = (P1) - (M1);                                   \
        if(FENCE == 1)                                             \
            asm volatile("" : "+v"(d1)::"memory");                 \
        double d2 = (P2) - (M2);                                   \
        if(FENCE)                                                  \
            asm volatile("" : "+v"(d1), "+v"(d2)::"memory");       \
        return ffac * (c1 * d1 - c2 * d2);                         \
    }()
    double phi = M(a, b, e);
    if(FENCE == 1)
        asm volatile("" : "+v"(phi)::"memory");
    const double fz = SHQ_PM_DIFF(M(a, b, e + 1), M(a, b, e - 1), M(a, b, e + 2), M(a, b, e - 2));
    const double fx = SHQ_PM_DIFF(M(a + 1, b, e), M(a - 1, b, e), M(a + 2, b, e), M(a - 2, b, e));
    const double fy = SHQ_PM_DIFF(M(a, b + 1, e), M(a, b - 1, e), M(a, b + 2, e), M(a, b - 2, e));
#undef SHQ_PM_DIFF
    gp += w * phi;
    g0 += w * fx;
    g1 += w * fy;
    g2 += w * fz;
    if(FENCE) /* the corner is finished before the next one starts: otherwise the sums are sunk and several corners' values stay live */
        asm volatile("" : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(gp)::"memory");
}

/* The whole readout of one particle on the full periodic mesh ([N][N][zp] doubles, fewer than 2^32 BYTES of them), register-lean,
 * for callers that sit in another kernel's register budget (the tree walk's prologue); pm_readout_kernel spends registers on
 * wider loads.  `interior` (wave-uniform): no lane's 6^3 neighbourhood crosses a face of the box — then a cell offset (dx, dy, dz)
 * is the same number of bytes for every lane, the compiler folds it into the scalar base of the load, and a lane carries ONE
 * offset register through all 104 loads.  Otherwise every offset is wrapped per lane (a few per cent of the waves). */
__device__ __forceinline__ void pm_readout_lean(const double *__restrict__ mesh, int N, int zp, double cell, double ffac, double px, double py,
                                                double pz, bool live, double &g0, double &g1, double &g2, double &gp)
{
    int ic[3];
    double res[3];
    cic_setup(px, cell, N, ic[0], res[0]);
    cic_setup(py, cell, N, ic[1], res[1]);
    cic_setup(pz, cell, N, ic[2], res[2]);
    g0 = g1 = g2 = gp = 0;
    const unsigned sy = (unsigned) zp, sx = (unsigned) N * (unsigned) zp;
    const bool inner = ic[0] >= 2 && ic[0] + 3 < N && ic[1] >= 2 && ic[1] + 3 < N && ic[2] >= 2 && ic[2] + 3 < N;
    const bool interior = __builtin_amdgcn_ballot_w64(live && !inner) == 0ull;
    if(interior) {
        if(live) {
            const unsigned vb = (((unsigned) ic[0] * sx + (unsigned) ic[1] * sy + (unsigned) ic[2]) << 3);
            const char *base = reinterpret_cast<const char *>(mesh);
#pragma unroll
            for(int c = 0; c < 8; c++) {
                /* the strides are laundered once per corner: otherwise the ~45 distinct scalar bases of the eight corners are formed
                 * once, up front, and stay live (70 scalar registers spilled) */
                long long sxc = (long long) sx * 8, syc = (long long) sy * 8;
                asm volatile("" : "+s"(sxc), "+s"(syc));
                auto M = [&](int dx, int dy, int dz) {
                    /* wave-uniform part first: sb is a scalar address, vb the lane's 32-bit byte offset */
                    typedef const __attribute__((address_space(1))) char *GlobalBytes;
                    GlobalBytes sb = (GlobalBytes) (base + (dx * sxc + dy * syc + dz * 8));
                    asm volatile("" : "+s"(sb)); /* pinned in scalar registers: otherwise the offsets are re-associated into vector registers */
                    return *reinterpret_cast<const __attribute__((address_space(1))) double *>(sb + vb);
                };
                pm_readout_corner<SHQ_READOUT_FENCE>(c, res, ffac, M, g0, g1, g2, gp);
            }
        }
    } else if(live) {
#pragma unroll 1
        for(int c = 0; c < 8; c++) {
            auto M = [&](int dx, int dy, int dz) {
                /* a 32-bit BYTE offset from the wave-uniform base: one address register per load (global_load with an SGPR base) */
                const unsigned o = (unsigned) wrapi(ic[0] + dx, N) * sx + (unsigned) wrapi(ic[1] + dy, N) * sy + (unsigned) wrapi(ic[2] + dz, N);
                return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(mesh) + (o << 3));
            };
            pm_readout_corner<1>(c, res, ffac, M, g0, g1, g2, gp);
        }
    }
}
