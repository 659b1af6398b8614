/* fft3d.hip — bespoke in-place 3-D real FFT pipeline for the PM mesh on gfx950.
 *
 * Replaces the reference's heffte/cuFFT r2c + c2r (libgadget/petapm.cpp:49-71) and the separate
 * transfer-function sweep (pm_apply_transfer_function, petapm.cpp:1258-1298) for mesh sizes
 * N = 2^a 3^b (b <= 1, N <= 1024).  rocFFT spends 6 memory passes per 3-D transform (3 FFT + 3
 * transpose kernels, 8.4 + 9.7 ms at 768^3); this pipeline needs FIVE passes for the whole
 * forward -> Green's function -> inverse sequence, each one read + one write of the mesh:
 *
 *   Z fwd : rows along z, two real rows per complex FFT (two-for-one), int64 fixed-point deposit
 *           converted on load (fuses pm_convert_kernel)            real [x][y][z] -> half spectrum
 *   Y fwd : lines along y, tiles of 4 adjacent z' columns (64-byte row segments)
 *   X     : lines along x: forward FFT, potential_transfer (gravpm.cpp:378-444) on the line while it
 *           sits in LDS, inverse FFT (fuses pm_green_kernel and saves a whole read+write pass)
 *   Y inv, Z inv (c2r, two-for-one).
 *
 * Every 1-D transform is a Stockham autosort FFT in LDS (radix 3, then radix 4, then radix 2):
 * each stage reads its butterflies' inputs into registers, barrier, writes the outputs to the same
 * buffer, barrier — one LDS buffer per line (N+1 elements: the +1 de-conflicts the four column
 * lines of a tile).  Unscaled in both directions, like FFTW/heffte.
 * The z pitch of the mesh is padded to a multiple of 4 complex values so that every 4-column row
 * segment is one aligned 64-byte chunk.
 */
#include "common.hpp"
#include <math.h>

namespace {

#define FFT_C 4        /* complex lines per workgroup */
#define FFT_T 256      /* threads per workgroup */
#define FFT_K 4        /* max butterflies per thread per stage (N <= 1024) */

struct FftPlan {
    int N;
    int nstages;
    int radix[12];
};

__device__ __forceinline__ double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double2 conj2(double2 a) { return make_double2(a.x, -a.y); }
/* multiply by -i (forward) or +i (inverse) */
template <int DIR> __device__ __forceinline__ double2 rot(double2 a) { return DIR < 0 ? make_double2(a.y, -a.x) : make_double2(-a.y, a.x); }
template <int DIR> __device__ __forceinline__ double2 tw(const double2 *__restrict__ W, int idx)
{
    const double2 w = W[idx];
    return DIR < 0 ? w : conj2(w);
}

/* One Stockham stage of radix R on C lines of length N held in LDS with line stride LS.
 * n = current sub-transform length, s = stride (product of the radices already applied). */
template <int R, int DIR>
__device__ __forceinline__ void fft_stage(double2 *buf, const int N, const int LS, const int n, const int s,
                                          const double2 *__restrict__ W)
{
    const int m = n / R;
    const int nb = N / R;               /* butterflies per line */
    const int total = FFT_C * nb;
    double2 v[FFT_K][R];
#pragma unroll
    for(int kk = 0; kk < FFT_K; kk++) {
        const int i = threadIdx.x + kk * FFT_T;
        if(i < total) {
            const int line = i / nb, b = i - line * nb;
            const int p = b / s, q = b - p * s;
            const double2 *x = buf + line * LS + q + s * p;
            double2 a[R];
#pragma unroll
            for(int j = 0; j < R; j++)
                a[j] = x[s * m * j];
            if(R == 2) {
                v[kk][0] = cadd(a[0], a[1]);
                v[kk][1] = cmul(csub(a[0], a[1]), tw<DIR>(W, p * s));
            } else if(R == 4) {
                const double2 t0 = cadd(a[0], a[2]), t1 = csub(a[0], a[2]);
                const double2 t2 = cadd(a[1], a[3]), t3 = rot<DIR>(csub(a[1], a[3]));
                v[kk][0] = cadd(t0, t2);
                v[kk][1] = cmul(cadd(t1, t3), tw<DIR>(W, p * s));
                v[kk][2] = cmul(csub(t0, t2), tw<DIR>(W, 2 * p * s));
                v[kk][3] = cmul(csub(t1, t3), tw<DIR>(W, 3 * p * s));
            } else { /* R == 3 */
                const double c = -0.5, sn = (DIR < 0 ? -1.0 : 1.0) * 0.86602540378443864676;
                const double2 t1 = cadd(a[1], a[2]);
                const double2 t2 = make_double2(a[0].x + c * t1.x, a[0].y + c * t1.y);
                const double2 d = csub(a[1], a[2]);
                const double2 t3 = make_double2(-sn * d.y, sn * d.x); /* i * sn * d */
                v[kk][0] = cadd(a[0], t1);
                v[kk][1] = cmul(cadd(t2, t3), tw<DIR>(W, p * s));
                v[kk][2] = cmul(csub(t2, t3), tw<DIR>(W, 2 * p * s));
            }
        }
    }
    __syncthreads();
#pragma unroll
    for(int kk = 0; kk < FFT_K; kk++) {
        const int i = threadIdx.x + kk * FFT_T;
        if(i < total) {
            const int line = i / nb, b = i - line * nb;
            const int p = b / s, q = b - p * s;
            double2 *y = buf + line * LS + q + s * R * p;
#pragma unroll
            for(int k = 0; k < R; k++)
                y[s * k] = v[kk][k];
        }
    }
    __syncthreads();
}

template <int DIR> __device__ __forceinline__ void fft_lines(double2 *buf, const FftPlan &pl, const int LS, const double2 *__restrict__ W)
{
    int n = pl.N, s = 1;
    for(int st = 0; st < pl.nstages; st++) {
        const int r = pl.radix[st];
        if(r == 3)
            fft_stage<3, DIR>(buf, pl.N, LS, n, s, W);
        else if(r == 4)
            fft_stage<4, DIR>(buf, pl.N, LS, n, s, W);
        else
            fft_stage<2, DIR>(buf, pl.N, LS, n, s, W);
        n /= r;
        s *= r;
    }
}

/* ---- pass Z forward: two real rows -> two half spectra, in place ---------------------------------
 * mesh: [nrows][zp] doubles (zp = pitch, >= N + 2).  Workgroup = FFT_C complex lines = 2 FFT_C rows. */
template <bool FROM_I64>
__global__ __launch_bounds__(FFT_T) void fft_pass_z_fwd(double *mesh, const long long nrows, const int zp, const FftPlan pl,
                                                        const double2 *__restrict__ W, const double inv_scale)
{
    extern __shared__ double2 buf[];
    const int N = pl.N, LS = N + 1;
    const long long row0 = (long long) blockIdx.x * (2 * FFT_C);
    for(int e = threadIdx.x; e < 2 * FFT_C * N; e += FFT_T) {
        const int r = e / N, z = e - r * N;
        const long long row = row0 + r;
        double v = 0;
        if(row < nrows) {
            if(FROM_I64)
                v = (double) reinterpret_cast<const long long *>(mesh)[row * zp + z] * inv_scale;
            else
                v = mesh[row * zp + z];
        }
        double *dst = reinterpret_cast<double *>(buf + (r >> 1) * LS + z);
        dst[r & 1] = v;
    }
    __syncthreads();
    fft_lines<-1>(buf, pl, LS, W);
    /* separate the two real transforms: XA[k] = (Z[k] + conj Z[N-k]) / 2, XB[k] = -i (Z[k] - conj Z[N-k]) / 2 */
    const int Nc = N / 2 + 1;
    double2 *cm = reinterpret_cast<double2 *>(mesh);
    const int zpc = zp / 2;
    for(int e = threadIdx.x; e < FFT_C * Nc; e += FFT_T) {
        const int l = e / Nc, k = e - l * Nc;
        const double2 zk = buf[l * LS + k];
        const double2 zn = conj2(buf[l * LS + (k == 0 ? 0 : N - k)]);
        const double2 xa = make_double2(0.5 * (zk.x + zn.x), 0.5 * (zk.y + zn.y));
        const double2 d = make_double2(0.5 * (zk.x - zn.x), 0.5 * (zk.y - zn.y));
        const double2 xb = make_double2(d.y, -d.x);
        const long long ra = row0 + 2 * l, rb = ra + 1;
        if(ra < nrows)
            cm[ra * zpc + k] = xa;
        if(rb < nrows)
            cm[rb * zpc + k] = xb;
    }
}

/* ---- pass Z inverse (c2r): two half spectra -> two real rows, in place -------------------------------- */
__global__ __launch_bounds__(FFT_T) void fft_pass_z_inv(double *mesh, const long long nrows, const int zp, const FftPlan pl,
                                                        const double2 *__restrict__ W)
{
    extern __shared__ double2 buf[];
    const int N = pl.N, LS = N + 1, Nc = N / 2 + 1;
    const long long row0 = (long long) blockIdx.x * (2 * FFT_C);
    const double2 *cm = reinterpret_cast<const double2 *>(mesh);
    const int zpc = zp / 2;
    for(int e = threadIdx.x; e < FFT_C * Nc; e += FFT_T) {
        const int l = e / Nc, k = e - l * Nc;
        const long long ra = row0 + 2 * l, rb = ra + 1;
        double2 xa = make_double2(0, 0), xb = make_double2(0, 0);
        if(ra < nrows)
            xa = cm[ra * zpc + k];
        if(rb < nrows)
            xb = cm[rb * zpc + k];
        if(k == 0 || 2 * k == N) { /* a c2r transform ignores the imaginary part of the self-conjugate modes */
            xa.y = 0;
            xb.y = 0;
        }
        /* Z[k] = XA[k] + i XB[k];  Z[N-k] = conj(XA[k]) + i conj(XB[k]) */
        buf[l * LS + k] = make_double2(xa.x - xb.y, xa.y + xb.x);
        if(k > 0 && 2 * k < N)
            buf[l * LS + N - k] = make_double2(xa.x + xb.y, -xa.y + xb.x);
    }
    __syncthreads();
    fft_lines<+1>(buf, pl, LS, W);
    for(int e = threadIdx.x; e < 2 * FFT_C * N; e += FFT_T) {
        const int r = e / N, z = e - r * N;
        const long long row = row0 + r;
        if(row < nrows) {
            const double *src = reinterpret_cast<const double *>(buf + (r >> 1) * LS + z);
            mesh[row * zp + z] = src[r & 1];
        }
    }
}

/* ---- passes Y and X: complex lines with element stride `es` (in complex units), FFT_C adjacent columns.
 * MODE 0: forward only; 1: inverse only; 2: forward, Green's function, inverse (X pass of the PM). */
struct GreenArgs {
    const double *sinctab; /* 1 / sinc^2(pi k / N) per mesh index */
    double asmth2, pot_factor;
};

template <int MODE>
__global__ __launch_bounds__(FFT_T) void fft_pass_strided(double2 *cm, const long long es, const long long outer_stride,
                                                          const int ntiles, const FftPlan pl, const double2 *__restrict__ W,
                                                          const GreenArgs ga)
{
    extern __shared__ double2 buf[];
    const int N = pl.N, LS = N + 1;
    const int outer = blockIdx.x / ntiles, tile = blockIdx.x - outer * ntiles;
    double2 *base = cm + (long long) outer * outer_stride + (long long) tile * FFT_C;
    for(int e = threadIdx.x; e < FFT_C * N; e += FFT_T) {
        const int row = e / FFT_C, col = e - row * FFT_C;
        buf[col * LS + row] = base[(long long) row * es + col];
    }
    __syncthreads();
    if(MODE == 0 || MODE == 2)
        fft_lines<-1>(buf, pl, LS, W);
    if(MODE == 2) {
        /* potential_transfer, gravpm.cpp:378-444: line index = kx, outer = y, column = z' */
        const int y = outer;
        const int ky = y <= N / 2 ? y : y - N;
        for(int e = threadIdx.x; e < FFT_C * N; e += FFT_T) {
            const int col = e / N, x = e - col * N;
            const int z = tile * FFT_C + col;
            const int kx = x <= N / 2 ? x : x - N;
            const long long k2 = (long long) kx * kx + (long long) ky * ky + (long long) z * z;
            double2 v = buf[col * LS + x];
            if(k2 == 0 || z > N / 2) {
                v.x = 0;
                v.y = 0;
            } else {
                double f = 1.0;
                const double smth = exp(-(double) k2 * ga.asmth2) / (double) k2;
                f *= ga.sinctab[x];
                f *= ga.sinctab[y];
                f *= ga.sinctab[z];
                const double fac = ga.pot_factor * smth * f * f;
                v.x *= fac;
                v.y *= fac;
            }
            buf[col * LS + x] = v;
        }
        __syncthreads();
    }
    if(MODE == 1 || MODE == 2)
        fft_lines<+1>(buf, pl, LS, W);
    for(int e = threadIdx.x; e < FFT_C * N; e += FFT_T) {
        const int row = e / FFT_C, col = e - row * FFT_C;
        base[(long long) row * es + col] = buf[col * LS + row];
    }
}

bool make_plan(int N, FftPlan *pl)
{
    if(N < 4 || N > 1024)
        return false;
    int n = N, ns = 0;
    pl->N = N;
    int threes = 0;
    while(n % 3 == 0) {
        n /= 3;
        threes++;
    }
    if(threes > 1)
        return false;
    int twos = 0;
    while(n % 2 == 0) {
        n /= 2;
        twos++;
    }
    if(n != 1)
        return false;
    if(threes) {
        if(FFT_C * (N / 3) > FFT_K * FFT_T)
            return false;
        pl->radix[ns++] = 3;
    }
    for(int i = 0; i < twos / 2; i++)
        pl->radix[ns++] = 4;
    if(twos % 2) {
        if(FFT_C * (N / 2) > FFT_K * FFT_T)
            return false;
        pl->radix[ns++] = 2;
    }
    pl->nstages = ns;
    return true;
}

} // namespace

bool shq_fft3d_supported(int N)
{
    FftPlan pl;
    return make_plan(N, &pl) && N % 2 == 0;
}

/* z pitch (in doubles) the bespoke pipeline wants: N/2+1 complex rounded up to a multiple of 4. */
int shq_fft3d_pitch(int N) { return 2 * (((N / 2 + 1) + 3) / 4 * 4); }

static int ensure_twiddles(shq_context *ctx, int N)
{
    if(ctx->fft_tw_n == N)
        return SHQ_OK;
    SHQ_TRY(ctx->fft_tw.reserve(2 * (size_t) N));
    std::vector<double> h(2 * (size_t) N);
    for(int k = 0; k < N; k++) {
        h[2 * k] = cos(2 * M_PI * k / N);
        h[2 * k + 1] = -sin(2 * M_PI * k / N);
    }
    SHQ_HIP(hipMemcpy(ctx->fft_tw.ptr, h.data(), sizeof(double) * 2 * N, hipMemcpyHostToDevice));
    ctx->fft_tw_n = N;
    return SHQ_OK;
}

/* stage: 0 forward only (r2c), 1 inverse only (c2r), 2 forward + potential_transfer + inverse.
 * d_mesh: [N][N][zp] doubles in place; from_i64: the mesh holds the int64 fixed-point deposit. */
int shq_fft3d_run(shq_context *ctx, double *d_mesh, int N, int zp, int stage, bool from_i64, double inv_scale,
                  const double *d_sinctab, double asmth2, double pot_factor)
{
    FftPlan pl;
    SHQ_CHECK(make_plan(N, &pl) && N % 2 == 0, SHQ_ERR_INVALID, "fft3d: unsupported mesh size %d", N);
    SHQ_CHECK(zp >= N + 2 && zp % 8 == 0, SHQ_ERR_INVALID, "fft3d: pitch %d must be a multiple of 8 doubles and >= N+2", zp);
    SHQ_TRY(ensure_twiddles(ctx, N));
    const double2 *W = reinterpret_cast<const double2 *>(ctx->fft_tw.ptr);
    const size_t lds = sizeof(double2) * FFT_C * (N + 1);
    const long long nrows = (long long) N * N;
    const int zpc = zp / 2;
    const int ntiles = zpc / FFT_C;
    const unsigned zblocks = (unsigned) ((nrows + 2 * FFT_C - 1) / (2 * FFT_C));
    double2 *cm = reinterpret_cast<double2 *>(d_mesh);
    GreenArgs ga;
    ga.sinctab = d_sinctab;
    ga.asmth2 = asmth2;
    ga.pot_factor = pot_factor;
    hipStream_t s = ctx->stream;
    static bool attr_done = false;
    if(!attr_done) { /* allow > 48 KB of dynamic LDS */
        (void) hipFuncSetAttribute((const void *) fft_pass_z_fwd<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        (void) hipFuncSetAttribute((const void *) fft_pass_z_fwd<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        (void) hipFuncSetAttribute((const void *) fft_pass_z_inv, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        (void) hipFuncSetAttribute((const void *) fft_pass_strided<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        (void) hipFuncSetAttribute((const void *) fft_pass_strided<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        (void) hipFuncSetAttribute((const void *) fft_pass_strided<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        attr_done = true;
    }
    if(stage == 0 || stage == 2) {
        if(from_i64)
            fft_pass_z_fwd<true><<<dim3(zblocks), dim3(FFT_T), lds, s>>>(d_mesh, nrows, zp, pl, W, inv_scale);
        else
            fft_pass_z_fwd<false><<<dim3(zblocks), dim3(FFT_T), lds, s>>>(d_mesh, nrows, zp, pl, W, 1.0);
        /* Y: outer = x plane (stride N*zpc), element stride zpc */
        fft_pass_strided<0><<<dim3((unsigned) (N * ntiles)), dim3(FFT_T), lds, s>>>(cm, zpc, (long long) N * zpc, ntiles, pl, W, ga);
    }
    /* X: outer = y (stride zpc), element stride N*zpc */
    if(stage == 0)
        fft_pass_strided<0><<<dim3((unsigned) (N * ntiles)), dim3(FFT_T), lds, s>>>(cm, (long long) N * zpc, zpc, ntiles, pl, W, ga);
    else if(stage == 1)
        fft_pass_strided<1><<<dim3((unsigned) (N * ntiles)), dim3(FFT_T), lds, s>>>(cm, (long long) N * zpc, zpc, ntiles, pl, W, ga);
    else
        fft_pass_strided<2><<<dim3((unsigned) (N * ntiles)), dim3(FFT_T), lds, s>>>(cm, (long long) N * zpc, zpc, ntiles, pl, W, ga);
    if(stage == 1 || stage == 2) {
        fft_pass_strided<1><<<dim3((unsigned) (N * ntiles)), dim3(FFT_T), lds, s>>>(cm, zpc, (long long) N * zpc, ntiles, pl, W, ga);
        fft_pass_z_inv<<<dim3(zblocks), dim3(FFT_T), lds, s>>>(d_mesh, nrows, zp, pl, W);
    }
    SHQ_HIP(hipGetLastError());
    return SHQ_OK;
}
