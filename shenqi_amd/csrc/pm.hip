/* pm.hip — long-range particle-mesh force entirely on the device (gfx950).
 *
 * Replaces, for a single rank, the host loops of libgadget/petapm.cpp that stay on the CPU even
 * with UseGPU (petapm.cpp:21-28): CIC deposit (pm_iterate/put_particle_to_mesh :1132-1197,
 * :1304-1310), the transfer-function sweeps (pm_apply_transfer_function :1258-1298 with
 * potential_transfer gravpm.cpp:378-444) and the CIC readout (gravpm.cpp:489-500), plus the
 * heffte/cuFFT r2c/c2r (petapm.cpp:49-71).
 *
 * MI355X-first choices:
 *  - no regions/pencil exchange on one GPU: particles deposit straight into the global mesh;
 *  - the deposit accumulates in 64-bit fixed point with integer atomics: integer addition is
 *    associative, so the mesh is bit-identical for any particle order / GPU count (the
 *    reference's `omp atomic` f64 adds are order dependent);
 *  - one in-place r2c and ONE in-place c2r (the potential).  The reference does three more
 *    c2r's after multiplying by i*K(w), K = (8 sin w - sin 2w)/6 (gravpm.cpp:448-478); that
 *    symbol is exactly the Fourier image of the 4-point difference
 *        -(N/L) [ 2/3 (f(+1)-f(-1)) - 1/12 (f(+2)-f(-2)) ],
 *    which the readout kernel applies in real space while gathering: 2 FFTs instead of 5.
 * All arithmetic is f64.
 */
#include "common.hpp"
#include "pm_readout.hpp"
#include <math.h>
#include <stdlib.h>
#include <string.h>

namespace {

/* SHQ_PM_XCD_K / SHQ_PM_DEP_XCD_K: workgroups per XCD chunk of the readout / deposit block order (xcd_block; 0 = round robin) */
static unsigned pm_xcdk(int deposit)
{
    static const unsigned kr = getenv("SHQ_PM_XCD_K") ? (unsigned) atoi(getenv("SHQ_PM_XCD_K")) : 64u;
    static const unsigned kd = getenv("SHQ_PM_DEP_XCD_K") ? (unsigned) atoi(getenv("SHQ_PM_DEP_XCD_K")) : 0u;
    return deposit ? kd : kr;
}

__global__ void pm_zero_kernel(unsigned long long *mesh, size_t n)
{
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    for(; i < n; i += stride)
        mesh[i] = 0ull;
}

/* put_particle_to_mesh, petapm.cpp:1304-1310, fixed-point accumulate.
 *
 * One workgroup owns DEP_CHUNK consecutive particles.  Particles arrive in space-filling-curve
 * order, so a chunk is spatially compact: when the bounding box of its CIC footprints fits a
 * DEP_T^3 tile the contributions are first summed in LDS (64-bit integer ds atomics) and only
 * the touched cells are flushed with one global atomic each.  In clustered regions (hundreds of
 * particles per cell) this removes almost all global atomics and all same-address contention;
 * sparse chunks whose footprint does not fit fall back to direct global atomics.  Integer adds
 * commute, so both routes give bit-identical meshes. */
#ifndef DEP_T
#define DEP_T 16
#endif
#ifndef DEP_PPT
#define DEP_PPT 2
#endif
#define DEP_CHUNK (256 * DEP_PPT)

__global__ __launch_bounds__(256) void pm_deposit_kernel(const double4 *__restrict__ posm, const uint8_t *__restrict__ pflags,
                                                         long long n, unsigned long long *mesh, int N, int zp, double cell,
                                                         double scale, int xshift, int nxalloc, int *oob, unsigned xcdk)
{
    __shared__ unsigned long long tile[DEP_T * DEP_T * DEP_T];
    __shared__ int s_min[3], s_max[3];
    const int tid = threadIdx.x;
    const long long base = (long long) xcd_block(blockIdx.x, gridDim.x, xcdk) * DEP_CHUNK;
    if(tid < 3) {
        s_min[tid] = 0x7fffffff;
        s_max[tid] = -0x7fffffff;
    }
    __syncthreads();
    int ic[DEP_PPT][3];
    double res[DEP_PPT][3];
    double mass[DEP_PPT];
    bool ok[DEP_PPT];
    int lmin[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, lmax[3] = {-0x7fffffff, -0x7fffffff, -0x7fffffff};
#pragma unroll
    for(int j = 0; j < DEP_PPT; j++) {
        const long long i = base + (long long) j * 256 + tid;
        ok[j] = (i < n) && !(pflags && (pflags[i] & 2)); /* Swallowed: RegionInd = -2, gravpm.cpp:176-178 */
        mass[j] = 0;
        if(ok[j]) {
            const double4 p = posm[i];
            cic_setup(p.x, cell, N, ic[j][0], res[j][0]);
            cic_setup(p.y, cell, N, ic[j][1], res[j][1]);
            cic_setup(p.z, cell, N, ic[j][2], res[j][2]);
            mass[j] = p.w;
#pragma unroll
            for(int k = 0; k < 3; k++) {
                lmin[k] = min(lmin[k], ic[j][k]);
                lmax[k] = max(lmax[k], ic[j][k]);
            }
        }
    }
#pragma unroll
    for(int k = 0; k < 3; k++) {
        int mn = lmin[k], mx = lmax[k];
        for(int off = 32; off > 0; off >>= 1) {
            mn = min(mn, __shfl_xor(mn, off));
            mx = max(mx, __shfl_xor(mx, off));
        }
        if((tid & 63) == 0) {
            atomicMin(&s_min[k], mn);
            atomicMax(&s_max[k], mx);
        }
    }
    __syncthreads();
    const int m0 = s_min[0], m1 = s_min[1], m2 = s_min[2];
    const bool fits = (s_max[0] - m0 + 2 <= DEP_T) && (s_max[1] - m1 + 2 <= DEP_T) && (s_max[2] - m2 + 2 <= DEP_T);
    const size_t sy = (size_t) zp, sx = (size_t) N * zp;
    if(fits) {
        /* the tile is the chunk's own footprint, ey x ez x ex cells (a dense clump covers a few dozen cells: clearing and flushing
         * all DEP_T^3 slots for it was most of the kernel's LDS traffic) */
        const int ey = s_max[1] - m1 + 2, ez = s_max[2] - m2 + 2, vol = (s_max[0] - m0 + 2) * ey * ez;
        for(int c = tid; c < vol; c += 256)
            tile[c] = 0ull;
        __syncthreads();
#pragma unroll
        for(int j = 0; j < DEP_PPT; j++) {
            if(!ok[j])
                continue;
#pragma unroll
            for(int c = 0; c < 8; c++) {
                double w = 1.0;
                int lin = 0;
#pragma unroll
                for(int k = 0; k < 3; k++) {
                    const int off = (c >> k) & 1;
                    const int t = ic[j][k] - (k == 0 ? m0 : (k == 1 ? m1 : m2)) + off;
                    lin = lin * (k == 0 ? 1 : (k == 1 ? ey : ez)) + t;
                    w *= off ? res[j][k] : (1 - res[j][k]);
                }
                const long long q = __double2ll_rn(w * mass[j] * scale);
                atomicAdd(&tile[lin], (unsigned long long) q);
            }
        }
        __syncthreads();
        for(int c = tid; c < vol; c += 256) {
            const unsigned long long v = tile[c];
            if(v != 0ull) {
                const int tz = c % ez, ty = (c / ez) % ey, tx = c / (ez * ey);
                const int lx = xloc(m0 + tx, xshift, N);
                if(lx >= nxalloc) {
                    *oob = 1; /* a particle outside this rank's slab: refuse to write out of bounds */
                    continue;
                }
                const size_t lin = (size_t) lx * sx + (size_t) wrapi(m1 + ty, N) * sy + (size_t) wrapi(m2 + tz, N);
                atomicAdd(&mesh[lin], v);
            }
        }
    } else {
        /* Direct global atomics — but a global atomic is one 64-byte request to the memory side whatever it carries (the chip
         * completes ~24 G of them per second: this path, the sparse background of the box, is what the kernel's time is made of),
         * and two lanes of one instruction whose cells share 64 bytes share the request.  The cells z and z + 1 of a CIC column are
         * neighbours in memory, so lane pairs take them: every lane forms its particle's eight (cell, value) entries, the wave
         * trades them through LDS (the tile is idle on this path), and instruction k hands the pair (2 p, 2 p + 1) the two z cells
         * of column k & 3 of particle p + 32 (k >> 2).  8 requests per particle become ~4.5.  Integer adds: the mesh does not care. */
        static_assert(DEP_T * DEP_T * DEP_T >= 4 * 8 * 64 * 2, "the staging area is the LDS tile");
        ulonglong2 *const stg = reinterpret_cast<ulonglong2 *>(tile) + (tid >> 6) * (8 * 64);
        const int lane = tid & 63;
#pragma unroll
        for(int j = 0; j < DEP_PPT; j++) {
#pragma unroll
            for(int c = 0; c < 8; c++) {
                double w = 1.0;
                size_t lin = 0;
                bool inb = true;
#pragma unroll
                for(int k = 0; k < 3; k++) {
                    const int off = (c >> k) & 1;
                    const int t = (k == 0) ? xloc(ic[j][k] + off, xshift, N) : wrapi(ic[j][k] + off, N);
                    if(k == 0 && t >= nxalloc)
                        inb = false;
                    lin += (size_t) t * (k == 0 ? sx : (k == 1 ? sy : 1));
                    w *= off ? res[j][k] : (1 - res[j][k]);
                }
                const long long q = __double2ll_rn(w * mass[j] * scale);
                if(ok[j] && !inb)
                    *oob = 1;
                /* slot = column (x, y offsets) * 2 + z offset */
                stg[((c & 3) * 2 + (c >> 2)) * 64 + lane] = make_ulonglong2(ok[j] && inb ? (unsigned long long) lin : ~0ull, (unsigned long long) q);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for(int k = 0; k < 8; k++) {
                const ulonglong2 e = stg[((k & 3) * 2 + (lane & 1)) * 64 + (lane >> 1) + 32 * (k >> 2)];
                if(e.x != ~0ull)
                    atomicAdd(&mesh[e.x], e.y);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

__global__ void pm_convert_kernel(double *mesh, size_t n, double inv_scale)
{
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    long long *im = reinterpret_cast<long long *>(mesh);
    for(; i < n; i += stride)
        mesh[i] = (double) im[i] * inv_scale;
}

/* potential_transfer, gravpm.cpp:378-444, on the [x][y][z'] half spectrum */
__global__ __launch_bounds__(256) void pm_green_kernel(double2 *cmesh, int N, int Nc, int zpc, const double *__restrict__ sinctab,
                                                       double asmth2, double pot_factor)
{
    const size_t total = (size_t) N * N * Nc;
    size_t ip = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if(ip >= total)
        return;
    const int z = (int) (ip % Nc);
    const size_t xy = ip / Nc;
    ip = xy * (size_t) zpc + z; /* complex pitch of the half spectrum */
    const int y = (int) (xy % N);
    const int x = (int) (xy / N);
    const int kx = x <= N / 2 ? x : x - N; /* petapm_mesh_to_k, petapm.cpp:159-162 */
    const int ky = y <= N / 2 ? y : y - N;
    const int kz = z;
    const long long k2 = (long long) kx * kx + (long long) ky * ky + (long long) kz * kz;
    double2 v = cmesh[ip];
    if(k2 == 0) {
        v.x = 0;
        v.y = 0;
    } else {
        double f = 1.0;
        const double smth = exp(-(double) k2 * asmth2) / (double) k2;
        f *= sinctab[x];
        f *= sinctab[y];
        f *= sinctab[z];
        const double fac = pot_factor * smth * f * f;
        v.x *= fac;
        v.y *= fac;
    }
    cmesh[ip] = v;
}

struct __attribute__((aligned(8))) pair8 { double x, y; }; /* two z-adjacent cells, one 16-byte load */

/* readout_potential / readout_force_{x,y,z}, gravpm.cpp:489-500, with the force obtained by
 * 4-point differencing of the potential mesh (see file header). */
__global__ __launch_bounds__(256) void pm_readout_kernel(const double4 *__restrict__ posm, const uint8_t *__restrict__ pflags,
                                                         long long n, const double *__restrict__ mesh, int N, int zp,
                                                         double cell, double ffac, double *gravpm, double *pmpot, int xshift,
                                                         int nxalloc, int *oob, unsigned xcdk, const double *__restrict__ treeacc = nullptr,
                                                         double *oldacc = nullptr, double G = 0)
{
#pragma clang fp contract(off) /* see pm_readout_corner: the routes of the readout agree to the bit by construction, not by luck */
    /* consecutive workgroups (consecutive runs of the space-filling curve: neighbouring mesh lines) share an XCD's L2 */
    const long long i = (long long) xcd_block(blockIdx.x, gridDim.x, xcdk) * blockDim.x + threadIdx.x;
    if(i >= n)
        return;
    double g0 = 0, g1 = 0, g2 = 0, gp = 0;
    if(!(pflags && (pflags[i] & 2))) {
        const double4 p = posm[i];
        int ic[3];
        double res[3];
        cic_setup(p.x, cell, N, ic[0], res[0]);
        cic_setup(p.y, cell, N, ic[1], res[1]);
        cic_setup(p.z, cell, N, ic[2], res[2]);
        const size_t sy = (size_t) zp, sx = (size_t) N * zp;
        /* wrapped indices for offsets -2..3 along every axis */
        size_t ox[6], oy[6], oz[6];
#pragma unroll
        for(int d = 0; d < 6; d++) {
            const int lx = xloc(ic[0] + d - 2, xshift, N);
            if(lx >= nxalloc && oob)
                *oob = 1;
            ox[d] = (size_t) (lx < nxalloc ? lx : 0) * sx;
            oy[d] = (size_t) wrapi(ic[1] + d - 2, N) * sy;
            oz[d] = (size_t) wrapi(ic[2] + d - 2, N);
        }
        const double c1 = 2.0 / 3.0, c2 = 1.0 / 12.0;
        if(ic[2] >= 2 && ic[2] + 3 < N) {
            /* The 8 corners' 13-point stencils touch only 56 distinct cells, and along z they are
             * contiguous in memory: fetch them as 28 pairs (16-byte loads, 8-byte aligned) instead of
             * 104 scattered doubles.  The kernel is bound by the texture-address path, one cache-line
             * lookup per lane and load, so this is what sets its run time.  Same values, same
             * operation order as the generic path below. */
            const double *m = mesh + (ic[2] - 2); /* z offset 0 of the 6-wide window */
            pair8 xl[2][6], yl[2][6], zl[2][2][3];
#pragma unroll
            for(int b2 = 0; b2 < 2; b2++)
#pragma unroll
                for(int d = 0; d < 6; d++)
                    xl[b2][d] = *reinterpret_cast<const pair8 *>(m + ox[d] + oy[2 + b2] + 2); /* Z = 2, 3 */
#pragma unroll
            for(int a2 = 0; a2 < 2; a2++)
#pragma unroll
                for(int d = 0; d < 6; d++)
                    yl[a2][d] = (d == 2 || d == 3) ? xl[d - 2][2 + a2] : *reinterpret_cast<const pair8 *>(m + ox[2 + a2] + oy[d] + 2);
#pragma unroll
            for(int a2 = 0; a2 < 2; a2++)
#pragma unroll
                for(int b2 = 0; b2 < 2; b2++) {
                    zl[a2][b2][0] = *reinterpret_cast<const pair8 *>(m + ox[2 + a2] + oy[2 + b2]);
                    zl[a2][b2][1] = xl[b2][2 + a2];
                    zl[a2][b2][2] = *reinterpret_cast<const pair8 *>(m + ox[2 + a2] + oy[2 + b2] + 4);
                }
#pragma unroll
            for(int c = 0; c < 8; c++) {
                const int a = c & 1, b = (c >> 1) & 1, e = (c >> 2) & 1;
                const double w = (a ? res[0] : 1 - res[0]) * (b ? res[1] : 1 - res[1]) * (e ? res[2] : 1 - res[2]);
                const int X = 2 + a, Y = 2 + b, Z = 2 + e;
#define PZ(P) ((e) ? (P).y : (P).x)
#define ZV(K) (((K) & 1) ? zl[a][b][(K) >> 1].y : zl[a][b][(K) >> 1].x)
                const double phi = PZ(xl[b][X]);
                const double fx = ffac * (c1 * (PZ(xl[b][X + 1]) - PZ(xl[b][X - 1])) - c2 * (PZ(xl[b][X + 2]) - PZ(xl[b][X - 2])));
                const double fy = ffac * (c1 * (PZ(yl[a][Y + 1]) - PZ(yl[a][Y - 1])) - c2 * (PZ(yl[a][Y + 2]) - PZ(yl[a][Y - 2])));
                const double fz = ffac * (c1 * (ZV(Z + 1) - ZV(Z - 1)) - c2 * (ZV(Z + 2) - ZV(Z - 2)));
#undef PZ
#undef ZV
                gp += w * phi;
                g0 += w * fx;
                g1 += w * fy;
                g2 += w * fz;
            }
        } else {
            auto M = [&](int dx, int dy, int dz) { return mesh[ox[2 + dx] + oy[2 + dy] + oz[2 + dz]]; };
#pragma unroll
            for(int c = 0; c < 8; c++)
                pm_readout_corner<0>(c, res, ffac, M, g0, g1, g2, gp);
        }
    }
    gravpm[3 * i + 0] = g0;
    gravpm[3 * i + 1] = g1;
    gravpm[3 * i + 2] = g2;
    pmpot[i] = gp;
    if(oldacc) { /* shq_treepm_step: grav_get_abs_accel (gravshort2.hpp:111-121) with the new GravPM, the operations of oldacc_kernel */
        const double gv[3] = {g0, g1, g2};
        double s = 0;
        for(int j = 0; j < 3; j++) {
            const double ax = treeacc[3 * i + j] + gv[j];
            s += ax * ax;
        }
        oldacc[i] = sqrt(s) / G;
    }
}

/* copy rows of `len` doubles between pitches; as_i64: the source holds fixed-point integers */
__global__ void pm_repitch_kernel(const double *src, double *dst, size_t nrows, int len, int spitch, int dpitch, int as_i64,
                                  double inv_scale)
{
    const size_t total = nrows * (size_t) len;
    size_t ip = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if(ip >= total)
        return;
    const int z = (int) (ip % len);
    const size_t row = ip / len;
    if(as_i64)
        dst[row * dpitch + z] = (double) reinterpret_cast<const long long *>(src)[row * spitch + z] * inv_scale;
    else
        dst[row * dpitch + z] = src[row * spitch + z];
}

} // namespace

void shq_pm_destroy_plans(shq_context *ctx)
{
    if(ctx->have_plans) {
        hipfftDestroy(ctx->plan_r2c);
        hipfftDestroy(ctx->plan_c2r);
        ctx->have_plans = false;
    }
    ctx->pm_nmesh = 0;
    ctx->pm_custom_fft = false;
}

static int pm_prepare(shq_context *ctx, int N)
{
    SHQ_CHECK(N >= 4 && N % 2 == 0, SHQ_ERR_INVALID, "Nmesh must be even and >= 4 (got %d)", N);
    ctx->mesh_zeroed = false; /* every caller is about to write the mesh (shq_pm_run reads the flag first) */
    if(ctx->pm_nmesh == N && (ctx->have_plans || ctx->pm_custom_fft))
        return SHQ_OK;
    shq_pm_destroy_plans(ctx);
    ctx->pm_custom_fft = shq_fft3d_supported(N) && !getenv("SHQ_PM_ROCFFT");
    ctx->pm_zp = ctx->pm_custom_fft ? shq_fft3d_pitch(N) : N + 2;
    const size_t padded = (size_t) N * N * ctx->pm_zp;
    SHQ_TRY(ctx->mesh.reserve(padded));
    ctx->mesh_words = padded;
    SHQ_TRY(ctx->sinctab.reserve(N));
    SHQ_TRY(ctx->pm_oob.reserve(1));
    SHQ_HIP(hipMemset(ctx->pm_oob.ptr, 0, sizeof(int)));
    /* 1/sinc^2(pi k/N) per mesh index: gravpm.cpp:294-302, :398-402 */
    std::vector<double> tab(N);
    for(int i = 0; i < N; i++) {
        const int k = i <= N / 2 ? i : i - N;
        double tmp = (k * M_PI) / N;
        double s;
        if(tmp < 1e-5 && tmp > -1e-5) {
            double x2 = tmp * tmp;
            s = 1.0 - x2 / 6. + x2 * x2 / 120.;
        } else
            s = sin(tmp) / tmp;
        tab[i] = 1. / (s * s);
    }
    SHQ_HIP(hipMemcpy(ctx->sinctab.ptr, tab.data(), sizeof(double) * N, hipMemcpyHostToDevice));
    ctx->sinctab_n = N;
    ctx->pm_nmesh = N;
    if(ctx->pm_custom_fft)
        return SHQ_OK;
    hipfftResult r = hipfftPlan3d(&ctx->plan_r2c, N, N, N, HIPFFT_D2Z);
    SHQ_CHECK(r == HIPFFT_SUCCESS, SHQ_ERR_DEVICE, "hipfftPlan3d(D2Z, %d) failed: %d", N, (int) r);
    r = hipfftPlan3d(&ctx->plan_c2r, N, N, N, HIPFFT_Z2D);
    SHQ_CHECK(r == HIPFFT_SUCCESS, SHQ_ERR_DEVICE, "hipfftPlan3d(Z2D, %d) failed: %d", N, (int) r);
    hipfftSetStream(ctx->plan_r2c, ctx->stream);
    hipfftSetStream(ctx->plan_c2r, ctx->stream);
    ctx->have_plans = true;
    ctx->pm_nmesh = N;
    return SHQ_OK;
}

namespace {

/* powerspectrum_add_mode (libgadget/gravpm.cpp:323-356) through measure_power_spectrum / potential_transfer
 * (:360-376, :430): every mode of the density half spectrum adds w |delta_k|^2 f^2 (f = the CIC
 * deconvolution, w = 1 on the kz = 0 and Nmesh/2 planes, else 2), w and w |k| to bin
 * floor(binsperunit log(k2) / 2); the zero mode sets Norm.  The bin of every k2 comes from a table made on
 * the host with the reference's expression, so modes land in exactly the bins the reference puts them in.
 * Each workgroup histograms in LDS and flushes with one atomic per non-empty bin. */
__global__ __launch_bounds__(256) void pm_power_kernel(const double2 *__restrict__ cmesh, int N, int Nc, int zpc, const double *__restrict__ sinctab,
                                                       const int32_t *__restrict__ bintab, int nbins, double *power, double *kk,
                                                       unsigned long long *nmodes, double *norm)
{
    extern __shared__ double hist[]; /* [3][nbins]: power, kk, modes */
    for(int i = threadIdx.x; i < 3 * nbins; i += blockDim.x)
        hist[i] = 0;
    __syncthreads();
    const size_t total = (size_t) N * N * Nc;
    for(size_t ip = (size_t) blockIdx.x * blockDim.x + threadIdx.x; ip < total; ip += (size_t) gridDim.x * blockDim.x) {
        const int z = (int) (ip % Nc);
        const size_t xy = ip / Nc;
        const int y = (int) (xy % N), x = (int) (xy / N);
        const int kx = x <= N / 2 ? x : x - N, ky = y <= N / 2 ? y : y - N, kz = z;
        const long long k2 = (long long) kx * kx + (long long) ky * ky + (long long) kz * kz;
        const double2 v = cmesh[xy * (size_t) zpc + z];
        const double m = v.x * v.x + v.y * v.y;
        if(k2 == 0) {
            *norm = m;
            continue;
        }
        const int kint = bintab[k2];
        if(kint >= nbins)
            continue;
        const double f = sinctab[x] * sinctab[y] * sinctab[z];
        const double w = (kz == 0 || kz == N / 2) ? 1.0 : 2.0;
        atomicAdd(&hist[kint], w * m * f * f);
        atomicAdd(&hist[nbins + kint], w * sqrt((double) k2));
        atomicAdd(&hist[2 * nbins + kint], w);
    }
    __syncthreads();
    for(int i = threadIdx.x; i < nbins; i += blockDim.x) {
        if(hist[2 * nbins + i] != 0) {
            atomicAdd(&power[i], hist[i]);
            atomicAdd(&kk[i], hist[nbins + i]);
            atomicAdd(&nmodes[i], (unsigned long long) hist[2 * nbins + i]);
        }
    }
}

int pm_measure_power(shq_context *ctx, int N, int zpc)
{
    const int nbins = N; /* powerspectrum_alloc(pm->ps, pm->Nmesh, ...), gravpm.cpp:207 */
    const long long k2max = 3ll * (N / 2) * (N / 2);
    if(ctx->ps_bintab_n != N) {
        std::vector<int32_t> tab((size_t) k2max + 1, 0);
        const double binsperunit = (nbins - 1) / log(sqrt(3) * N / 2.0);
        for(long long k2 = 1; k2 <= k2max; k2++)
            tab[k2] = (int32_t) floor(binsperunit * log((double) k2) / 2.);
        SHQ_TRY(ctx->ps_bintab.reserve(tab.size()));
        SHQ_HIP(hipMemcpy(ctx->ps_bintab.ptr, tab.data(), sizeof(int32_t) * tab.size(), hipMemcpyHostToDevice));
        ctx->ps_bintab_n = N;
    }
    SHQ_TRY(ctx->ps_sums.reserve(3 * (size_t) nbins + 1));
    SHQ_HIP(hipMemsetAsync(ctx->ps_sums.ptr, 0, sizeof(double) * (3 * (size_t) nbins + 1), ctx->stream));
    double *power = ctx->ps_sums.ptr, *kk = power + nbins, *norm = power + 3 * nbins;
    unsigned long long *nmodes = reinterpret_cast<unsigned long long *>(power + 2 * nbins);
    const size_t lds = sizeof(double) * 3 * nbins;
    if(lds > 48 * 1024)
        SHQ_HIP(hipFuncSetAttribute((const void *) pm_power_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    pm_power_kernel<<<dim3(2048), dim3(256), lds, ctx->stream>>>((const double2 *) ctx->mesh.ptr, N, N / 2 + 1, zpc, ctx->sinctab.ptr,
                                                                ctx->ps_bintab.ptr, nbins, power, kk, nmodes, norm);
    SHQ_HIP(hipGetLastError());
    ctx->ps_nbins = nbins;
    ctx->have_power = true;
    return SHQ_OK;
}

} // namespace

int shq_pm_execute(shq_context *ctx, const shq_pm_params *pm, bool readout)
{
    SHQ_CHECK(ctx->have_parts, SHQ_ERR_STATE, "pm: particles must be uploaded first");
    SHQ_CHECK(pm->BoxSize > 0 && pm->Asmth > 0, SHQ_ERR_INVALID, "pm params: BoxSize and Asmth must be > 0");
    const int N = pm->Nmesh;
    /* the mesh may have been cleared in the shadow of the last tree walk (grav_walk.hip, `scrub`) */
    const bool prezeroed = ctx->mesh_zeroed && ctx->pm_nmesh == N;
    SHQ_TRY(pm_prepare(ctx, N));
    const int zp = ctx->pm_zp;
    const size_t padded = (size_t) N * N * zp;
    const int Nc = N / 2 + 1;
    const double cell = pm->BoxSize / N; /* CellSize */
    const long long n = ctx->numpart;
    /* fixed-point scale: 2^e with e chosen so that the whole mass in one cell cannot overflow */
    const int e = ctx->pm_log2scale;
    const double scale = ldexp(1.0, e);
    const int threads = 256;
    const size_t dense = (size_t) N * N * N;
    const double asmth2 = pow((2 * M_PI) * pm->Asmth / N, 2);
    const double pot_factor = -pm->G / (M_PI * pm->BoxSize);

    SHQ_HIP(hipEventRecord(ctx->ev_begin[8], ctx->stream));
    if(!(prezeroed && ctx->mesh_words == padded))
        pm_zero_kernel<<<dim3(2048), dim3(threads), 0, ctx->stream>>>((unsigned long long *) ctx->mesh.ptr, padded);
    if(n > 0)
        pm_deposit_kernel<<<dim3((unsigned) ((n + DEP_CHUNK - 1) / DEP_CHUNK)), dim3(256), 0, ctx->stream>>>(
            ctx->posm.ptr, ctx->pflags.ptr, n, (unsigned long long *) ctx->mesh.ptr, N, zp, cell, scale, 0, N, ctx->pm_oob.ptr, pm_xcdk(1));
    if(ctx->pm_keep) {
        SHQ_TRY(ctx->dbg_rho.reserve(dense));
        pm_repitch_kernel<<<dim3((unsigned) ((dense + threads - 1) / threads)), dim3(threads), 0, ctx->stream>>>(
            ctx->mesh.ptr, ctx->dbg_rho.ptr, (size_t) N * N, N, zp, N, 1, 1.0 / scale);
    }
    if(ctx->pm_custom_fft && ctx->pm_measure_power) {
        /* P(k) is taken from the density spectrum, which the fused X pass never writes out: when it is
         * wanted the forward and inverse transforms run separately (6 passes + 2 sweeps instead of 5) */
        SHQ_HIP(hipEventRecord(ctx->ev_begin[9], ctx->stream));
        SHQ_TRY(shq_fft3d_run(ctx, ctx->mesh.ptr, N, zp, 0, true, 1.0 / scale, ctx->sinctab.ptr, asmth2, pot_factor));
        SHQ_HIP(hipEventRecord(ctx->ev_begin[10], ctx->stream));
        SHQ_TRY(pm_measure_power(ctx, N, zp / 2));
        const size_t tot = (size_t) N * N * Nc;
        pm_green_kernel<<<dim3((unsigned) ((tot + threads - 1) / threads)), dim3(threads), 0, ctx->stream>>>(
            (double2 *) ctx->mesh.ptr, N, Nc, zp / 2, ctx->sinctab.ptr, asmth2, pot_factor);
        SHQ_HIP(hipEventRecord(ctx->ev_begin[11], ctx->stream));
        SHQ_TRY(shq_fft3d_run(ctx, ctx->mesh.ptr, N, zp, 1, false, 1.0, ctx->sinctab.ptr, asmth2, pot_factor));
        SHQ_HIP(hipEventRecord(ctx->ev_begin[12], ctx->stream));
    } else if(ctx->pm_custom_fft) {
        /* five fused passes: Z fwd (+ int64 -> f64), Y fwd, X fwd + potential_transfer + X inv, Y inv, Z inv */
        /* the transposing pipeline (fft3d.hip) wants a second mesh as scratch: the one shq_treepm_step keeps anyway (between two steps
         * it holds the previous step's potential, which nothing reads any more once that step's walk has gone by on this stream) */
        double *scratch = nullptr;
        static const size_t scratch_off = getenv("SHQ_FFT_SCRATCH_OFFSET") ? (size_t) atoll(getenv("SHQ_FFT_SCRATCH_OFFSET")) / 8 : 0; /* probe */
        if(ctx->fft_transposed && zp == shq_fft3d_pitch(N) && ctx->mesh_alt.reserve(padded + scratch_off) == SHQ_OK)
            scratch = ctx->mesh_alt.ptr + scratch_off;
        SHQ_HIP(hipEventRecord(ctx->ev_begin[9], ctx->stream));
        if(scratch)
            SHQ_TRY(shq_fft3d_run_transposed(ctx, ctx->mesh.ptr, scratch, N, zp, true, 1.0 / scale, ctx->sinctab.ptr, asmth2, pot_factor));
        else
            SHQ_TRY(shq_fft3d_run(ctx, ctx->mesh.ptr, N, zp, 2, true, 1.0 / scale, ctx->sinctab.ptr, asmth2, pot_factor));
        SHQ_HIP(hipEventRecord(ctx->ev_begin[10], ctx->stream));
        SHQ_HIP(hipEventRecord(ctx->ev_begin[11], ctx->stream));
        SHQ_HIP(hipEventRecord(ctx->ev_begin[12], ctx->stream));
    } else {
        pm_convert_kernel<<<dim3(2048), dim3(threads), 0, ctx->stream>>>(ctx->mesh.ptr, padded, 1.0 / scale);
        SHQ_HIP(hipGetLastError());
        SHQ_HIP(hipEventRecord(ctx->ev_begin[9], ctx->stream));
        hipfftSetStream(ctx->plan_r2c, ctx->stream);
        hipfftResult r = hipfftExecD2Z(ctx->plan_r2c, (hipfftDoubleReal *) ctx->mesh.ptr, (hipfftDoubleComplex *) ctx->mesh.ptr);
        SHQ_CHECK(r == HIPFFT_SUCCESS, SHQ_ERR_DEVICE, "hipfftExecD2Z failed: %d", (int) r);
        SHQ_HIP(hipEventRecord(ctx->ev_begin[10], ctx->stream));
        {
            const size_t tot = (size_t) N * N * Nc;
            if(ctx->pm_measure_power)
                SHQ_TRY(pm_measure_power(ctx, N, zp / 2));
            pm_green_kernel<<<dim3((unsigned) ((tot + threads - 1) / threads)), dim3(threads), 0, ctx->stream>>>(
                (double2 *) ctx->mesh.ptr, N, Nc, zp / 2, ctx->sinctab.ptr, asmth2, pot_factor);
        }
        SHQ_HIP(hipEventRecord(ctx->ev_begin[11], ctx->stream));
        hipfftSetStream(ctx->plan_c2r, ctx->stream);
        r = hipfftExecZ2D(ctx->plan_c2r, (hipfftDoubleComplex *) ctx->mesh.ptr, (hipfftDoubleReal *) ctx->mesh.ptr);
        SHQ_CHECK(r == HIPFFT_SUCCESS, SHQ_ERR_DEVICE, "hipfftExecZ2D failed: %d", (int) r);
        SHQ_HIP(hipEventRecord(ctx->ev_begin[12], ctx->stream));
    }
    if(ctx->pm_keep) {
        SHQ_TRY(ctx->dbg_pot.reserve(dense));
        pm_repitch_kernel<<<dim3((unsigned) ((dense + threads - 1) / threads)), dim3(threads), 0, ctx->stream>>>(
            ctx->mesh.ptr, ctx->dbg_pot.ptr, (size_t) N * N, N, zp, N, 0, 1.0);
    }
    ctx->fuse_cell = cell;
    ctx->fuse_ffac = -(N / pm->BoxSize);
    if(n > 0 && readout) { /* !readout: shq_treepm_step, the tree walk's prologue reads the potential mesh */
        const double ffac = -(N / pm->BoxSize);
        pm_readout_kernel<<<dim3((unsigned) ((n + threads - 1) / threads)), dim3(threads), 0, ctx->stream>>>(
            ctx->posm.ptr, ctx->pflags.ptr, n, ctx->mesh.ptr, N, zp, cell, ffac, ctx->gravpm.ptr, ctx->pmpot.ptr, 0, N, ctx->pm_oob.ptr, pm_xcdk(0),
            ctx->readout_oldacc_G > 0 ? ctx->treeacc.ptr : nullptr, ctx->readout_oldacc_G > 0 ? ctx->oldacc.ptr : nullptr, ctx->readout_oldacc_G);
    }
    SHQ_HIP(hipGetLastError());
    SHQ_HIP(hipEventRecord(ctx->ev_begin[13], ctx->stream));
    ctx->have_pm_result = true;
    return SHQ_OK;
}

extern "C" int shq_pm_set_fft_transposed(shq_context *ctx, int enable)
{
    SHQ_CHECK(ctx, SHQ_ERR_INVALID, "null context");
    ctx->fft_transposed = enable != 0;
    return SHQ_OK;
}

extern "C" int shq_pm_measure_power(shq_context *ctx, int enable)
{
    SHQ_CHECK(ctx, SHQ_ERR_INVALID, "null context");
    ctx->pm_measure_power = enable != 0;
    if(!enable)
        ctx->have_power = false;
    return SHQ_OK;
}

extern "C" int shq_pm_download_power(shq_context *ctx, int size, double *kk, double *power, int64_t *nmodes, double *norm)
{
    if(ctx)
        SHQ_TRY(shq_join_pm(ctx));
    SHQ_CHECK(ctx && kk && power && nmodes && norm, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->have_power && ctx->have_pm_result, SHQ_ERR_STATE, "pm_download_power: enable shq_pm_measure_power before the PM run");
    SHQ_CHECK(size == ctx->ps_nbins, SHQ_ERR_INVALID, "pm_download_power: size %d, the spectrum has %d bins (= Nmesh)", size, ctx->ps_nbins);
    const int nb = ctx->ps_nbins;
    std::vector<double> h(3 * (size_t) nb + 1);
    SHQ_HIP(hipMemcpyAsync(h.data(), ctx->ps_sums.ptr, sizeof(double) * h.size(), hipMemcpyDeviceToHost, ctx->stream));
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    for(int i = 0; i < nb; i++) {
        power[i] = h[i];
        kk[i] = h[nb + i];
        unsigned long long c;
        memcpy(&c, &h[2 * (size_t) nb + i], sizeof(c));
        nmodes[i] = (int64_t) c;
    }
    *norm = h[3 * (size_t) nb];
    return SHQ_OK;
}

/* complex [x][y][z'] <-> [y][z'][x] for a fixed y per blockIdx.z: a 32 x 32 tile of the (x, z') plane goes through LDS so
 * that both the reads and the writes run along the fastest index of their array */
__global__ void fourier_layout_kernel(const double2 *__restrict__ in, double2 *__restrict__ out, int N, int nz, int back)
{
    __shared__ double2 tile[32][33];
    const int y = blockIdx.z, z0 = blockIdx.x * 32, x0 = blockIdx.y * 32;
    if(!back) { /* in [x][y][z'] -> out [y][z'][x] */
        for(int j = threadIdx.y; j < 32; j += 8) {
            const int x = x0 + j, z = z0 + threadIdx.x;
            if(x < N && z < nz)
                tile[j][threadIdx.x] = in[((size_t) x * N + y) * nz + z];
        }
        __syncthreads();
        for(int j = threadIdx.y; j < 32; j += 8) {
            const int z = z0 + j, x = x0 + threadIdx.x;
            if(x < N && z < nz)
                out[((size_t) y * nz + z) * N + x] = tile[threadIdx.x][j];
        }
    } else { /* in [y][z'][x] -> out [x][y][z'] */
        for(int j = threadIdx.y; j < 32; j += 8) {
            const int z = z0 + j, x = x0 + threadIdx.x;
            if(x < N && z < nz)
                tile[threadIdx.x][j] = in[((size_t) y * nz + z) * N + x];
        }
        __syncthreads();
        for(int j = threadIdx.y; j < 32; j += 8) {
            const int x = x0 + j, z = z0 + threadIdx.x;
            if(x < N && z < nz)
                out[((size_t) x * N + y) * nz + z] = tile[j][threadIdx.x];
        }
    }
}

int shq_fft_roundtrip_r2c(shq_context *ctx, int N, const double *real, double *complx, bool ref_layout)
{
    SHQ_TRY(pm_prepare(ctx, N));
    const int zp = ctx->pm_zp;
    const size_t tot = (size_t) N * N * N, ctot = (size_t) N * N * (N + 2);
    DevBuf<double> dense;
    SHQ_TRY(dense.reserve(ctot));
    SHQ_HIP(hipMemcpyAsync(dense.ptr, real, tot * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    const int threads = 256;
    pm_repitch_kernel<<<dim3((unsigned) ((tot + threads - 1) / threads)), dim3(threads), 0, ctx->stream>>>(
        dense.ptr, ctx->mesh.ptr, (size_t) N * N, N, N, zp, 0, 1.0);
    if(ctx->pm_custom_fft)
        SHQ_TRY(shq_fft3d_run(ctx, ctx->mesh.ptr, N, zp, 0, false, 1.0, ctx->sinctab.ptr, 0, 0));
    else {
        hipfftSetStream(ctx->plan_r2c, ctx->stream);
        hipfftResult r = hipfftExecD2Z(ctx->plan_r2c, (hipfftDoubleReal *) ctx->mesh.ptr, (hipfftDoubleComplex *) ctx->mesh.ptr);
        SHQ_CHECK(r == HIPFFT_SUCCESS, SHQ_ERR_DEVICE, "hipfftExecD2Z failed: %d", (int) r);
    }
    pm_repitch_kernel<<<dim3((unsigned) ((ctot + threads - 1) / threads)), dim3(threads), 0, ctx->stream>>>(
        ctx->mesh.ptr, dense.ptr, (size_t) N * N, N + 2, zp, N + 2, 0, 1.0);
    if(ref_layout) { /* [x][y][z'] -> the reference's Fourier layout [y][z'][x], x fastest (petapm.cpp:262-270) */
        DevBuf<double> tr;
        SHQ_TRY(tr.reserve(ctot));
        const int nz = N / 2 + 1;
        fourier_layout_kernel<<<dim3((unsigned) ((nz + 31) / 32), (unsigned) ((N + 31) / 32), (unsigned) N), dim3(32, 8), 0, ctx->stream>>>(
            reinterpret_cast<const double2 *>(dense.ptr), reinterpret_cast<double2 *>(tr.ptr), N, nz, 0);
        SHQ_HIP(hipMemcpyAsync(complx, tr.ptr, ctot * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        SHQ_HIP(hipStreamSynchronize(ctx->stream));
        tr.release();
        dense.release();
        return SHQ_OK;
    }
    SHQ_HIP(hipMemcpyAsync(complx, dense.ptr, ctot * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    dense.release();
    return SHQ_OK;
}

/* pm_apply_transfer_function (petapm.cpp:1258-1298) for the transfer functions of the other petapm clients, on the dense half
 * spectrum [x][y][z'] (kpos = mesh_to_k of each index, k2 their integer norm): the factor of a mode is a function of the integer k2 -
 * tabulated by the caller with its own functions (DeltaSpec, dlogGrowth, the neutrino spline ...), so the values are the reference's -
 * times 1, or i kpos[axis], or i diff_kernel(kpos[axis] 2 pi / N): libgenic/zeldovich.cpp:271-321, libgadget/plane.cpp:283-304,
 * gravpm.cpp:464-488; the reionisation filters of libgadget/uvbg.cpp:211-250 (top-hat, k-space top-hat, Gaussian in k R; divide_by_ncell) are
 * radial factors too.  The k2 = 0 mode is left alone (`if(k2)` of the zeldovich transfers), set to zero (plane.cpp:286), or multiplied by
 * T[0] like the others (divide_by_ncell). */
__global__ void pm_transfer_kernel(double2 *spec, int N, int Nc, const double *__restrict__ table, int kind, int axis, int zero_mode)
{
#pragma clang fp contract(off)
    const size_t total = (size_t) N * N * Nc;
    const size_t ip = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if(ip >= total)
        return;
    const int z = (int) (ip % Nc), y = (int) ((ip / Nc) % N), x = (int) (ip / ((size_t) Nc * N));
    const int kpos[3] = {x <= N / 2 ? x : x - N, y <= N / 2 ? y : y - N, z <= N / 2 ? z : z - N};
    const long long k2 = (long long) kpos[0] * kpos[0] + (long long) kpos[1] * kpos[1] + (long long) kpos[2] * kpos[2];
    double2 v = spec[ip];
    if(k2 == 0 && zero_mode != 2) { /* 2: the k2 = 0 mode takes its factor T[0] like every other (uvbg.cpp's divide_by_ncell) */
        if(zero_mode == 1)
            spec[ip] = make_double2(0.0, 0.0);
        return;
    }
    double fac = table[k2];
    if(kind == SHQ_TF_RADIAL) {
        v.x *= fac;
        v.y *= fac;
    } else {
        if(kind == SHQ_TF_GRADIENT)
            fac = fac * kpos[axis];
        else { /* SHQ_TF_DIFF: diff_kernel, gravpm.cpp:448-456 */
            const double w = kpos[axis] * (2 * M_PI / N);
            fac = fac * (1 / 6.0 * (8 * sin(w) - sin(2 * w)));
        }
        const double tmp = v.x;
        v.x = -v.y * fac;
        v.y = tmp * fac;
    }
    spec[ip] = v;
}

int shq_fft_roundtrip_c2r(shq_context *ctx, int N, const double *complx, double *real, bool ref_layout, const shq_pm_transfer *tf)
{
    SHQ_TRY(pm_prepare(ctx, N));
    const int zp = ctx->pm_zp;
    const size_t tot = (size_t) N * N * N, ctot = (size_t) N * N * (N + 2);
    DevBuf<double> dense;
    SHQ_TRY(dense.reserve(ctot));
    const int threads = 256;
    if(ref_layout) {
        DevBuf<double> tr;
        SHQ_TRY(tr.reserve(ctot));
        SHQ_HIP(hipMemcpyAsync(tr.ptr, complx, ctot * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        const int nz = N / 2 + 1;
        fourier_layout_kernel<<<dim3((unsigned) ((nz + 31) / 32), (unsigned) ((N + 31) / 32), (unsigned) N), dim3(32, 8), 0, ctx->stream>>>(
            reinterpret_cast<const double2 *>(tr.ptr), reinterpret_cast<double2 *>(dense.ptr), N, nz, 1);
        SHQ_HIP(hipStreamSynchronize(ctx->stream));
        tr.release();
    } else
        SHQ_HIP(hipMemcpyAsync(dense.ptr, complx, ctot * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    DevBuf<double> dtab;
    if(tf) {
        const size_t nk2 = 3 * (size_t) (N / 2) * (N / 2) + 1;
        SHQ_TRY(dtab.reserve(nk2));
        SHQ_HIP(hipMemcpyAsync(dtab.ptr, tf->table, nk2 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        const size_t modes = (size_t) N * N * (N / 2 + 1);
        pm_transfer_kernel<<<dim3((unsigned) ((modes + threads - 1) / threads)), dim3(threads), 0, ctx->stream>>>(
            reinterpret_cast<double2 *>(dense.ptr), N, N / 2 + 1, dtab.ptr, tf->kind, tf->axis, tf->zero_mode);
        SHQ_HIP(hipGetLastError());
    }
    pm_repitch_kernel<<<dim3((unsigned) ((ctot + threads - 1) / threads)), dim3(threads), 0, ctx->stream>>>(
        dense.ptr, ctx->mesh.ptr, (size_t) N * N, N + 2, N + 2, zp, 0, 1.0);
    if(ctx->pm_custom_fft)
        SHQ_TRY(shq_fft3d_run(ctx, ctx->mesh.ptr, N, zp, 1, false, 1.0, ctx->sinctab.ptr, 0, 0));
    else {
        hipfftSetStream(ctx->plan_c2r, ctx->stream);
        hipfftResult r = hipfftExecZ2D(ctx->plan_c2r, (hipfftDoubleComplex *) ctx->mesh.ptr, (hipfftDoubleReal *) ctx->mesh.ptr);
        SHQ_CHECK(r == HIPFFT_SUCCESS, SHQ_ERR_DEVICE, "hipfftExecZ2D failed: %d", (int) r);
    }
    pm_repitch_kernel<<<dim3((unsigned) ((tot + threads - 1) / threads)), dim3(threads), 0, ctx->stream>>>(
        ctx->mesh.ptr, dense.ptr, (size_t) N * N, N, zp, N, 0, 1.0);
    SHQ_HIP(hipMemcpyAsync(real, dense.ptr, tot * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    dense.release();
    dtab.release();
    return SHQ_OK;
}

/* pm_apply_transfer_function + petapm_fft_c2r for one transfer function of a petapm client other than the gravity PM (the
 * displacement / velocity / density fields of libgenic/zeldovich.cpp:215-227, 271-321; the neutrino correction of the lensing planes,
 * libgadget/plane.cpp:326-341): complx is the half spectrum in the reference's Fourier layout [y][z'][x] (as shq_fft_r2c returns it),
 * real receives the unscaled inverse transform [x][y][z].  One resident upload / transform / download per call. */
extern "C" int shq_pm_apply(shq_context *ctx, int Nmesh, const double *complx, const shq_pm_transfer *tf, double *real)
{
    SHQ_CHECK(ctx && complx && tf && real && tf->table, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(Nmesh >= 4 && Nmesh % 2 == 0, SHQ_ERR_INVALID, "pm_apply: Nmesh must be even and >= 4");
    SHQ_CHECK(tf->kind >= SHQ_TF_RADIAL && tf->kind <= SHQ_TF_DIFF && (tf->kind == SHQ_TF_RADIAL || (tf->axis >= 0 && tf->axis <= 2)), SHQ_ERR_INVALID,
              "pm_apply: kind %d / axis %d", tf->kind, tf->axis);
    SHQ_HIP(hipSetDevice(ctx->device));
    SHQ_TRY(shq_join_pm(ctx));
    ctx->have_pm_result = false; /* the context's mesh is used as work space */
    ctx->mesh_zeroed = false;
    return shq_fft_roundtrip_c2r(ctx, Nmesh, complx, real, true, tf);
}

/* ---- slab-sharded PM (multi-GPU): local phases on caller-provided device buffers ------------------
 * Rank r owns mesh planes [plane0, plane0 + nplanes).  The deposit writes nplanes + 1 planes (CIC
 * reaches one plane to the right); the readout reads nplanes + 5 planes starting at plane0 - 2
 * (CIC + the 4-point stencil).  Ghost planes are exchanged by the caller (RCCL). */
static int slab_sinctab(shq_context *ctx, int N)
{
    if(ctx->sinctab_n == N)
        return SHQ_OK;
    SHQ_TRY(ctx->sinctab.reserve(N));
    SHQ_TRY(ctx->pm_oob.reserve(1));
    std::vector<double> tab(N);
    for(int i = 0; i < N; i++) {
        const int k = i <= N / 2 ? i : i - N;
        double tmp = (k * M_PI) / N;
        double s;
        if(tmp < 1e-5 && tmp > -1e-5) {
            double x2 = tmp * tmp;
            s = 1.0 - x2 / 6. + x2 * x2 / 120.;
        } else
            s = sin(tmp) / tmp;
        tab[i] = 1. / (s * s);
    }
    SHQ_HIP(hipMemcpy(ctx->sinctab.ptr, tab.data(), sizeof(double) * N, hipMemcpyHostToDevice));
    ctx->sinctab_n = N;
    return SHQ_OK;
}

static int check_oob(shq_context *ctx, const char *what)
{
    int h = 0;
    SHQ_HIP(hipMemcpyAsync(&h, ctx->pm_oob.ptr, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    SHQ_HIP(hipStreamSynchronize(ctx->stream));
    SHQ_CHECK(h == 0, SHQ_ERR_INVALID, "%s: a particle lies outside this rank's mesh slab", what);
    return SHQ_OK;
}

extern "C" int shq_pm_slab_deposit(shq_context *ctx, const shq_pm_params *pm, int plane0, int nplanes, void *d_mesh_i64)
{
    if(ctx)
        SHQ_TRY(shq_join_pm(ctx));
    SHQ_CHECK(ctx && pm && d_mesh_i64, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->have_parts, SHQ_ERR_STATE, "pm_slab_deposit: particles must be uploaded first");
    const int N = pm->Nmesh;
    SHQ_CHECK(N >= 4 && N % 2 == 0 && nplanes > 0 && nplanes <= N && plane0 >= 0 && plane0 < N, SHQ_ERR_INVALID, "bad slab geometry");
    SHQ_HIP(hipSetDevice(ctx->device));
    SHQ_TRY(slab_sinctab(ctx, N));
    SHQ_HIP(hipMemsetAsync(ctx->pm_oob.ptr, 0, sizeof(int), ctx->stream));
    const int nalloc = nplanes == N ? N : nplanes + 1;
    const size_t cnt = (size_t) nalloc * N * (N + 2);
    pm_zero_kernel<<<dim3(2048), dim3(256), 0, ctx->stream>>>((unsigned long long *) d_mesh_i64, cnt);
    const long long n = ctx->nlocal;
    if(n > 0)
        pm_deposit_kernel<<<dim3((unsigned) ((n + DEP_CHUNK - 1) / DEP_CHUNK)), dim3(256), 0, ctx->stream>>>(
            ctx->posm.ptr, ctx->pflags.ptr, n, (unsigned long long *) d_mesh_i64, N, N + 2, pm->BoxSize / N, ldexp(1.0, ctx->pm_log2scale),
            plane0, nalloc, ctx->pm_oob.ptr, pm_xcdk(1));
    SHQ_HIP(hipGetLastError());
    return check_oob(ctx, "pm_slab_deposit");
}

extern "C" int shq_pm_slab_readout(shq_context *ctx, const shq_pm_params *pm, int plane0, int nplanes, const void *d_phi_ext)
{
    if(ctx)
        SHQ_TRY(shq_join_pm(ctx));
    SHQ_CHECK(ctx && pm && d_phi_ext, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->have_parts, SHQ_ERR_STATE, "pm_slab_readout: particles must be uploaded first");
    const int N = pm->Nmesh;
    SHQ_CHECK(N >= 4 && N % 2 == 0 && nplanes > 0 && nplanes <= N && plane0 >= 0 && plane0 < N, SHQ_ERR_INVALID, "bad slab geometry");
    SHQ_HIP(hipSetDevice(ctx->device));
    SHQ_TRY(ctx->pm_oob.reserve(1));
    SHQ_HIP(hipMemsetAsync(ctx->pm_oob.ptr, 0, sizeof(int), ctx->stream));
    const long long n = ctx->nlocal;
    const int nalloc = nplanes == N ? N : nplanes + 5;
    const int xshift = nplanes == N ? 0 : plane0 - 2;
    if(n > 0)
        pm_readout_kernel<<<dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, ctx->stream>>>(
            ctx->posm.ptr, ctx->pflags.ptr, n, (const double *) d_phi_ext, N, N + 2, pm->BoxSize / N, -(N / pm->BoxSize), ctx->gravpm.ptr,
            ctx->pmpot.ptr, xshift, nalloc, ctx->pm_oob.ptr, pm_xcdk(0));
    SHQ_HIP(hipGetLastError());
    ctx->have_pm_result = true;
    return check_oob(ctx, "pm_slab_readout");
}

/* potential_transfer (gravpm.cpp:378-444) on the transposed spectrum of a y-slab:
 * layout [ylocal][z' <= N/2][x], x fastest — the reference's own Fourier layout (petapm.cpp:243-282). */
__global__ __launch_bounds__(256) void pm_green_slab_kernel(double2 *c, int N, int Nc, int y0, int nyl, const double *__restrict__ sinctab,
                                                            double asmth2, double pot_factor)
{
    const size_t total = (size_t) nyl * Nc * N;
    const size_t ip = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if(ip >= total)
        return;
    const int x = (int) (ip % N);
    const size_t yz = ip / N;
    const int z = (int) (yz % Nc);
    const int y = y0 + (int) (yz / Nc);
    const int kx = x <= N / 2 ? x : x - N;
    const int ky = y <= N / 2 ? y : y - N;
    const long long k2 = (long long) kx * kx + (long long) ky * ky + (long long) z * z;
    double2 v = c[ip];
    if(k2 == 0) {
        v.x = 0;
        v.y = 0;
    } else {
        double f = 1.0;
        const double smth = exp(-(double) k2 * asmth2) / (double) k2;
        f *= sinctab[x];
        f *= sinctab[y];
        f *= sinctab[z];
        const double fac = pot_factor * smth * f * f;
        v.x *= fac;
        v.y *= fac;
    }
    c[ip] = v;
}

extern "C" int shq_pm_slab_green(shq_context *ctx, const shq_pm_params *pm, int y0, int nyl, void *d_spec)
{
    if(ctx)
        SHQ_TRY(shq_join_pm(ctx));
    SHQ_CHECK(ctx && pm && d_spec, SHQ_ERR_INVALID, "null argument");
    const int N = pm->Nmesh;
    SHQ_CHECK(N >= 4 && N % 2 == 0 && nyl > 0 && y0 >= 0 && y0 + nyl <= N, SHQ_ERR_INVALID, "bad slab geometry");
    SHQ_HIP(hipSetDevice(ctx->device));
    SHQ_TRY(slab_sinctab(ctx, N));
    const int Nc = N / 2 + 1;
    const size_t tot = (size_t) nyl * Nc * N;
    pm_green_slab_kernel<<<dim3((unsigned) ((tot + 255) / 256)), dim3(256), 0, ctx->stream>>>(
        (double2 *) d_spec, N, Nc, y0, nyl, ctx->sinctab.ptr, pow((2 * M_PI) * pm->Asmth / N, 2), -pm->G / (M_PI * pm->BoxSize));
    SHQ_HIP(hipGetLastError());
    return SHQ_OK;
}

/* ---- slab entry points on the bespoke FFT passes (mesh pitch zp = shq_pm_slab_pitch) -----------------
 * The x-slab of a rank lives in ONE buffer [nalloc][N][zp] that is in turn the int64 deposit mesh, the
 * (y, z) half spectrum and the potential, with ghost planes in place: the slab's first own plane sits at
 * buffer plane `xoff` (2 when the slab has neighbours: two potential ghost planes in front, the deposit
 * ghost plane and three potential ghost planes behind; 0 for a single rank, nalloc = N). */
extern "C" int shq_pm_slab_pitch(int Nmesh) { return shq_fft3d_supported(Nmesh) ? shq_fft3d_pitch(Nmesh) : 0; }

/* nghost: planes behind the slab's own that its particles may deposit into — 1 (the CIC neighbour plane of the last own plane),
 * or 2 when the slab also holds particles of the first plane of its right-hand neighbour (dist.py, slabs cut below the plane) */
extern "C" int shq_pm_slab2_deposit_ghosts(shq_context *ctx, const shq_pm_params *pm, int plane0, int nplanes, int xoff, int nalloc,
                                           int nghost, void *d_mesh_i64)
{
    if(ctx)
        SHQ_TRY(shq_join_pm(ctx));
    SHQ_CHECK(ctx && pm && d_mesh_i64, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->have_parts, SHQ_ERR_STATE, "pm_slab2_deposit: particles must be uploaded first");
    const int N = pm->Nmesh;
    SHQ_CHECK(shq_fft3d_supported(N), SHQ_ERR_INVALID, "pm_slab2: mesh size %d has no bespoke FFT", N);
    SHQ_CHECK(nplanes > 0 && nplanes <= N && plane0 >= 0 && plane0 < N && xoff >= 0 && xoff + nplanes <= nalloc && nalloc <= N + 8 &&
                  nghost >= 1 && nghost <= 2,
              SHQ_ERR_INVALID, "bad slab geometry");
    SHQ_HIP(hipSetDevice(ctx->device));
    SHQ_TRY(slab_sinctab(ctx, N));
    SHQ_HIP(hipMemsetAsync(ctx->pm_oob.ptr, 0, sizeof(int), ctx->stream));
    const int zp = shq_fft3d_pitch(N);
    const size_t cnt = (size_t) nalloc * N * zp;
    pm_zero_kernel<<<dim3(2048), dim3(256), 0, ctx->stream>>>((unsigned long long *) d_mesh_i64, cnt);
    const long long n = ctx->nlocal;
    /* buffer plane of mesh plane ix: (ix - (plane0 - xoff)) mod N; own planes and the right ghost(s) must fit */
    const int nfit = nplanes == N ? N : xoff + nplanes + nghost;
    SHQ_CHECK(nfit <= nalloc, SHQ_ERR_INVALID, "pm_slab2_deposit: no room for the deposit ghost plane");
    if(n > 0)
        pm_deposit_kernel<<<dim3((unsigned) ((n + DEP_CHUNK - 1) / DEP_CHUNK)), dim3(256), 0, ctx->stream>>>(
            ctx->posm.ptr, ctx->pflags.ptr, n, (unsigned long long *) d_mesh_i64, N, zp, pm->BoxSize / N, ldexp(1.0, ctx->pm_log2scale),
            plane0 - xoff, nfit, ctx->pm_oob.ptr, pm_xcdk(1));
    SHQ_HIP(hipGetLastError());
    return check_oob(ctx, "pm_slab2_deposit");
}

extern "C" int shq_pm_slab2_deposit(shq_context *ctx, const shq_pm_params *pm, int plane0, int nplanes, int xoff, int nalloc,
                                    void *d_mesh_i64)
{
    return shq_pm_slab2_deposit_ghosts(ctx, pm, plane0, nplanes, xoff, nalloc, 1, d_mesh_i64);
}

/* direction 0: int64 deposit planes -> (y, z) half spectrum (Z forward, Y forward); 1: back to real space.
 * d_planes points at the first of `nplanes` planes of N x zp doubles. */
extern "C" int shq_pm_slab2_fft_yz(shq_context *ctx, int Nmesh, void *d_planes, int nplanes, int direction)
{
    SHQ_CHECK(ctx && d_planes, SHQ_ERR_INVALID, "null argument");
    SHQ_TRY(shq_join_pm(ctx));
    SHQ_HIP(hipSetDevice(ctx->device));
    SHQ_TRY(slab_sinctab(ctx, Nmesh));
    const int zp = shq_fft3d_pitch(Nmesh);
    return shq_fft3d_run_slab(ctx, (double *) d_planes, Nmesh, zp, direction == 0 ? 10 : 11, direction == 0,
                              ldexp(1.0, -ctx->pm_log2scale), ctx->sinctab.ptr, 0, 0, nplanes, 0);
}

/* the same two stages with the pack / unpack of the transposes fused into the Y pass: direction 0 leaves the (y, z) spectrum of the
 * planes in d_packed = [nranks][nplanes][N / nranks][zp / 2] (complex) — rows [destination rank][x plane], what the all-to-all sends —
 * and direction 1 starts from d_packed in that layout (what the return all-to-all delivers) */
extern "C" int shq_pm_slab2_fft_yz_packed(shq_context *ctx, int Nmesh, void *d_planes, int nplanes, int direction, void *d_packed, int nranks)
{
    SHQ_CHECK(ctx && d_planes && d_packed, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(nranks >= 1 && Nmesh % nranks == 0, SHQ_ERR_INVALID, "pm_slab2_fft_yz_packed: %d ranks do not divide the mesh", nranks);
    SHQ_TRY(shq_join_pm(ctx));
    SHQ_HIP(hipSetDevice(ctx->device));
    SHQ_TRY(slab_sinctab(ctx, Nmesh));
    const int zp = shq_fft3d_pitch(Nmesh);
    return shq_fft3d_run_slab_packed(ctx, (double *) d_planes, Nmesh, zp, direction == 0 ? 13 : 14, direction == 0, ldexp(1.0, -ctx->pm_log2scale), ctx->sinctab.ptr, 0,
                                     0, nplanes, 0, (double *) d_packed, nranks);
}

/* X forward + potential_transfer + X inverse on the y-slab [N][nyl][zp / 2] (complex) received by the transpose */
extern "C" int shq_pm_slab2_xgreen(shq_context *ctx, const shq_pm_params *pm, void *d_spec, int y0, int nyl)
{
    SHQ_CHECK(ctx && pm && d_spec, SHQ_ERR_INVALID, "null argument");
    SHQ_TRY(shq_join_pm(ctx));
    const int N = pm->Nmesh;
    SHQ_HIP(hipSetDevice(ctx->device));
    SHQ_TRY(slab_sinctab(ctx, N));
    return shq_fft3d_run_slab(ctx, (double *) d_spec, N, shq_fft3d_pitch(N), 12, false, 1.0, ctx->sinctab.ptr,
                              pow((2 * M_PI) * pm->Asmth / N, 2), -pm->G / (M_PI * pm->BoxSize), nyl, y0);
}

extern "C" int shq_pm_slab2_readout(shq_context *ctx, const shq_pm_params *pm, int plane0, int nplanes, int xoff, int nalloc,
                                    const void *d_phi)
{
    if(ctx)
        SHQ_TRY(shq_join_pm(ctx));
    SHQ_CHECK(ctx && pm && d_phi, SHQ_ERR_INVALID, "null argument");
    SHQ_CHECK(ctx->have_parts, SHQ_ERR_STATE, "pm_slab2_readout: particles must be uploaded first");
    const int N = pm->Nmesh;
    SHQ_CHECK(shq_fft3d_supported(N) && nplanes > 0 && nplanes <= N && plane0 >= 0 && plane0 < N && xoff >= 0 && xoff + nplanes <= nalloc,
              SHQ_ERR_INVALID, "bad slab geometry");
    SHQ_HIP(hipSetDevice(ctx->device));
    SHQ_TRY(ctx->pm_oob.reserve(4));
    SHQ_HIP(hipMemsetAsync(ctx->pm_oob.ptr, 0, sizeof(int), ctx->stream));
    const long long n = ctx->nlocal;
    const int zp = shq_fft3d_pitch(N);
    if(n > 0)
        pm_readout_kernel<<<dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, ctx->stream>>>(
            ctx->posm.ptr, ctx->pflags.ptr, n, (const double *) d_phi, N, zp, pm->BoxSize / N, -(N / pm->BoxSize), ctx->gravpm.ptr,
            ctx->pmpot.ptr, plane0 - xoff, nalloc, ctx->pm_oob.ptr, pm_xcdk(0));
    SHQ_HIP(hipGetLastError());
    ctx->have_pm_result = true;
    return check_oob(ctx, "pm_slab2_readout");
}

extern "C" int shq_pm_get_deposit_log2scale(shq_context *ctx) { return ctx ? ctx->pm_log2scale : -1; }
extern "C" int shq_pm_set_deposit_log2scale(shq_context *ctx, int e)
{
    SHQ_CHECK(ctx && e >= -1 && e < 62, SHQ_ERR_INVALID, "bad scale exponent");
    ctx->pm_log2scale_user = e; /* -1: back to the scale chosen from the mass sum at the next particle upload */
    if(e >= 0)
        ctx->pm_log2scale = e;
    return SHQ_OK;
}
