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
#include <stdlib.h>

namespace {

#define FFT_C 4        /* complex lines per workgroup */
#define FFT_T 256      /* threads per workgroup */

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
/* exp(DIR * 2 pi i m / 16) for the in-register radix-16 butterfly */
template <int DIR, int M> __device__ __forceinline__ double2 w16()
{
    constexpr double c[16] = {1.0, 0.92387953251128673848, 0.70710678118654752440, 0.38268343236508977173, 0.0,
                              -0.38268343236508977173, -0.70710678118654752440, -0.92387953251128673848, -1.0,
                              -0.92387953251128673848, -0.70710678118654752440, -0.38268343236508977173, 0.0,
                              0.38268343236508977173, 0.70710678118654752440, 0.92387953251128673848};
    constexpr double sn[16] = {0.0, 0.38268343236508977173, 0.70710678118654752440, 0.92387953251128673848, 1.0,
                               0.92387953251128673848, 0.70710678118654752440, 0.38268343236508977173, 0.0,
                               -0.38268343236508977173, -0.70710678118654752440, -0.92387953251128673848, -1.0,
                               -0.92387953251128673848, -0.70710678118654752440, -0.38268343236508977173};
    return make_double2(c[M & 15], (DIR < 0 ? -1.0 : 1.0) * sn[M & 15]);
}

template <int DIR> __device__ __forceinline__ void bfly4(double2 &a0, double2 &a1, double2 &a2, double2 &a3)
{
    const double2 t0 = cadd(a0, a2), t1 = csub(a0, a2);
    const double2 t2 = cadd(a1, a3), t3 = rot<DIR>(csub(a1, a3));
    a0 = cadd(t0, t2);
    a1 = cadd(t1, t3);
    a2 = csub(t0, t2);
    a3 = csub(t1, t3);
}

/* b[k] = sum_j a[j] w16^(jk), in place, output in natural order k = k1 + 4 k0 stored at a[k] */
template <int DIR> __device__ __forceinline__ void bfly16(double2 (&a)[16])
{
    /* j = j0 + 4 j1: radix-4 over j1 for every j0 -> t[j0][k1] stored at a[j0 + 4 k1] */
#pragma unroll
    for(int j0 = 0; j0 < 4; j0++)
        bfly4<DIR>(a[j0], a[j0 + 4], a[j0 + 8], a[j0 + 12]);
    /* twiddle w16^(j0 k1) */
    a[1 + 4] = cmul(a[1 + 4], w16<DIR, 1>());
    a[2 + 4] = cmul(a[2 + 4], w16<DIR, 2>());
    a[3 + 4] = cmul(a[3 + 4], w16<DIR, 3>());
    a[1 + 8] = cmul(a[1 + 8], w16<DIR, 2>());
    a[2 + 8] = cmul(a[2 + 8], w16<DIR, 4>());
    a[3 + 8] = cmul(a[3 + 8], w16<DIR, 6>());
    a[1 + 12] = cmul(a[1 + 12], w16<DIR, 3>());
    a[2 + 12] = cmul(a[2 + 12], w16<DIR, 6>());
    a[3 + 12] = cmul(a[3 + 12], w16<DIR, 9>());
    /* radix-4 over j0 for every k1: b[k1 + 4 k0] */
#pragma unroll
    for(int k1 = 0; k1 < 4; k1++)
        bfly4<DIR>(a[4 * k1], a[4 * k1 + 1], a[4 * k1 + 2], a[4 * k1 + 3]);
    /* now a[4 k1 + k0] holds b[k1 + 4 k0]: transpose the 4x4 index to natural order */
#pragma unroll
    for(int k1 = 0; k1 < 4; k1++)
#pragma unroll
        for(int k0 = k1 + 1; k0 < 4; k0++) {
            const double2 tmp = a[4 * k1 + k0];
            a[4 * k1 + k0] = a[4 * k0 + k1];
            a[4 * k0 + k1] = tmp;
        }
}

/* One Stockham stage of radix R on FFT_C lines of length N held in LDS with line stride N + 1.
 * n = current sub-transform length, s = stride (product of the radices already applied); all are
 * compile-time constants, so the index arithmetic folds into shifts and constant multiplies. */
template <int N, int n, int s, int R, int DIR> __device__ __forceinline__ void fft_stage(double2 *buf, const double2 *__restrict__ W)
{
    constexpr int LS = N + 1;
    constexpr int m = n / R;
    constexpr int nb = N / R;               /* butterflies per line */
    constexpr int total = FFT_C * nb;
    constexpr int K = (total + FFT_T - 1) / FFT_T;
    double2 v[K][R];
#pragma unroll
    for(int kk = 0; kk < K; kk++) {
        const int i = threadIdx.x + kk * FFT_T;
        if(i < total) {
            const int line = i / nb, b = i - line * nb;
            const int p = b / s, q = b - p * s;
            const double2 *x = buf + line * LS + q + s * p;
#pragma unroll
            for(int j = 0; j < R; j++)
                v[kk][j] = x[s * m * j];
            if(R == 2) {
                const double2 a0 = v[kk][0], a1 = v[kk][1];
                v[kk][0] = cadd(a0, a1);
                v[kk][1] = csub(a0, a1);
            } else if(R == 4) {
                bfly4<DIR>(v[kk][0], v[kk][1], v[kk][2], v[kk][3]);
            } else if(R == 3) {
                const double c = -0.5, sn = (DIR < 0 ? -1.0 : 1.0) * 0.86602540378443864676;
                const double2 a0 = v[kk][0];
                const double2 t1 = cadd(v[kk][1], v[kk][2]);
                const double2 t2 = make_double2(a0.x + c * t1.x, a0.y + c * t1.y);
                const double2 d = csub(v[kk][1], v[kk][2]);
                const double2 t3 = make_double2(-sn * d.y, sn * d.x); /* i * sn * d */
                v[kk][0] = cadd(a0, t1);
                v[kk][1] = cadd(t2, t3);
                v[kk][2] = csub(t2, t3);
            } else {
                bfly16<DIR>(reinterpret_cast<double2(&)[16]>(v[kk]));
            }
            if(m > 1) { /* w_n^(p k); the last stage (m == 1) has p == 0 */
#pragma unroll
                for(int k = 1; k < R; k++)
                    v[kk][k] = cmul(v[kk][k], tw<DIR>(W, k * p * s));
            }
        }
    }
    __syncthreads();
#pragma unroll
    for(int kk = 0; kk < K; kk++) {
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

/* radix sequence: 16 while possible, then 4, 2, and a final 3 */
template <int N, int n, int s, int DIR> struct Stages {
    static __device__ __forceinline__ void run(double2 *buf, const double2 *__restrict__ W)
    {
        if constexpr(n % 16 == 0) {
            fft_stage<N, n, s, 16, DIR>(buf, W);
            Stages<N, n / 16, s * 16, DIR>::run(buf, W);
        } else if constexpr(n % 4 == 0) {
            fft_stage<N, n, s, 4, DIR>(buf, W);
            Stages<N, n / 4, s * 4, DIR>::run(buf, W);
        } else if constexpr(n % 2 == 0) {
            fft_stage<N, n, s, 2, DIR>(buf, W);
            Stages<N, n / 2, s * 2, DIR>::run(buf, W);
        } else if constexpr(n % 3 == 0) {
            fft_stage<N, n, s, 3, DIR>(buf, W);
            Stages<N, n / 3, s * 3, DIR>::run(buf, W);
        }
    }
};

template <int N, int DIR> __device__ __forceinline__ void fft_lines(double2 *buf, const double2 *__restrict__ W)
{
    Stages<N, N, 1, DIR>::run(buf, W);
}

/* ---- pass Z forward: two real rows -> two half spectra, in place ---------------------------------
 * mesh: [nrows][zp] doubles (zp = pitch, >= N + 2).  Workgroup = FFT_C complex lines = 2 FFT_C rows. */
template <int N, bool FROM_I64>
__global__ __launch_bounds__(FFT_T) void fft_pass_z_fwd(double *mesh, const long long nrows, const int zp,
                                                        const double2 *__restrict__ W, const double inv_scale)
{
    extern __shared__ double2 buf[];
    constexpr int LS = N + 1;
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
    fft_lines<N, -1>(buf, W);
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
template <int N>
__global__ __launch_bounds__(FFT_T) void fft_pass_z_inv(double *mesh, const long long nrows, const int zp,
                                                        const double2 *__restrict__ W)
{
    extern __shared__ double2 buf[];
    constexpr int LS = N + 1, Nc = N / 2 + 1;
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
    fft_lines<N, +1>(buf, W);
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

template <int N, int MODE>
__global__ __launch_bounds__(FFT_T) void fft_pass_strided(double2 *cm, const long long es, const long long outer_stride,
                                                          const int ntiles, const double2 *__restrict__ W, const GreenArgs ga, const unsigned xcdk)
{
    extern __shared__ double2 buf[];
    constexpr int LS = N + 1;
    /* XCD-chunked block order: neighbouring column tiles share 128-byte lines (a tile row is 64 bytes), so
     * they should run on the same XCD at about the same time and find the other half of the line in its L2 */
    const unsigned bid = xcd_block(blockIdx.x, gridDim.x, xcdk);
    const int outer = bid / ntiles, tile = bid - outer * ntiles;
    double2 *base = cm + (long long) outer * outer_stride + (long long) tile * FFT_C;
    for(int e = threadIdx.x; e < FFT_C * N; e += FFT_T) {
        const int row = e / FFT_C, col = e - row * FFT_C;
        buf[col * LS + row] = base[(long long) row * es + col];
    }
    __syncthreads();
    if(MODE == 0 || MODE == 2)
        fft_lines<N, -1>(buf, W);
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
        fft_lines<N, +1>(buf, W);
    for(int e = threadIdx.x; e < FFT_C * N; e += FFT_T) {
        const int row = e / FFT_C, col = e - row * FFT_C;
        base[(long long) row * es + col] = buf[col * LS + row];
    }
}

template <int N>
int run_n(shq_context *ctx, double *d_mesh, int zp, int stage, bool from_i64, double inv_scale, const GreenArgs &ga)
{
    const double2 *W = reinterpret_cast<const double2 *>(ctx->fft_tw.ptr);
    constexpr size_t lds = sizeof(double2) * FFT_C * (N + 1);
    const long long nrows = (long long) N * N;
    const int zpc = zp / 2;
    const int ntiles = zpc / FFT_C;
    const unsigned zblocks = (unsigned) ((nrows + 2 * FFT_C - 1) / (2 * FFT_C));
    double2 *cm = reinterpret_cast<double2 *>(d_mesh);
    hipStream_t s = ctx->stream;
    if(lds > 48 * 1024) { /* allow > 48 KB of dynamic LDS */
        (void) hipFuncSetAttribute((const void *) fft_pass_z_fwd<N, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        (void) hipFuncSetAttribute((const void *) fft_pass_z_fwd<N, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        (void) hipFuncSetAttribute((const void *) fft_pass_z_inv<N>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        (void) hipFuncSetAttribute((const void *) fft_pass_strided<N, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        (void) hipFuncSetAttribute((const void *) fft_pass_strided<N, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        (void) hipFuncSetAttribute((const void *) fft_pass_strided<N, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
    }
    const unsigned sblocks = (unsigned) (N * ntiles);
    const unsigned xcdk = getenv("SHQ_FFT_XCD_K") ? (unsigned) atoi(getenv("SHQ_FFT_XCD_K")) : 8u;
    if(stage == 0 || stage == 2) {
        if(from_i64)
            fft_pass_z_fwd<N, true><<<dim3(zblocks), dim3(FFT_T), lds, s>>>(d_mesh, nrows, zp, W, inv_scale);
        else
            fft_pass_z_fwd<N, false><<<dim3(zblocks), dim3(FFT_T), lds, s>>>(d_mesh, nrows, zp, W, 1.0);
        /* Y: outer = x plane (stride N*zpc), element stride zpc */
        fft_pass_strided<N, 0><<<dim3(sblocks), dim3(FFT_T), lds, s>>>(cm, zpc, (long long) N * zpc, ntiles, W, ga, xcdk);
    }
    /* X: outer = y (stride zpc), element stride N*zpc */
    if(stage == 0)
        fft_pass_strided<N, 0><<<dim3(sblocks), dim3(FFT_T), lds, s>>>(cm, (long long) N * zpc, zpc, ntiles, W, ga, xcdk);
    else if(stage == 1)
        fft_pass_strided<N, 1><<<dim3(sblocks), dim3(FFT_T), lds, s>>>(cm, (long long) N * zpc, zpc, ntiles, W, ga, xcdk);
    else
        fft_pass_strided<N, 2><<<dim3(sblocks), dim3(FFT_T), lds, s>>>(cm, (long long) N * zpc, zpc, ntiles, W, ga, xcdk);
    if(stage == 1 || stage == 2) {
        fft_pass_strided<N, 1><<<dim3(sblocks), dim3(FFT_T), lds, s>>>(cm, zpc, (long long) N * zpc, ntiles, W, ga, xcdk);
        fft_pass_z_inv<N><<<dim3(zblocks), dim3(FFT_T), lds, s>>>(d_mesh, nrows, zp, W);
    }
    SHQ_HIP(hipGetLastError());
    return SHQ_OK;
}

} // namespace

/* mesh sizes with a compiled pipeline: 2^a and 3 * 2^a */
bool shq_fft3d_supported(int N)
{
    switch(N) {
    case 16: case 24: case 32: case 48: case 64: case 96: case 128: case 192: case 256: case 384: case 512: case 768: case 1024:
        return true;
    }
    return false;
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
    SHQ_CHECK(shq_fft3d_supported(N), SHQ_ERR_INVALID, "fft3d: unsupported mesh size %d", N);
    SHQ_CHECK(zp >= N + 2 && zp % 8 == 0, SHQ_ERR_INVALID, "fft3d: pitch %d must be a multiple of 8 doubles and >= N+2", zp);
    SHQ_TRY(ensure_twiddles(ctx, N));
    GreenArgs ga;
    ga.sinctab = d_sinctab;
    ga.asmth2 = asmth2;
    ga.pot_factor = pot_factor;
#define SHQ_FFT_CASE(NN) case NN: return run_n<NN>(ctx, d_mesh, zp, stage, from_i64, inv_scale, ga)
    switch(N) {
        SHQ_FFT_CASE(16); SHQ_FFT_CASE(24); SHQ_FFT_CASE(32); SHQ_FFT_CASE(48); SHQ_FFT_CASE(64); SHQ_FFT_CASE(96);
        SHQ_FFT_CASE(128); SHQ_FFT_CASE(192); SHQ_FFT_CASE(256); SHQ_FFT_CASE(384); SHQ_FFT_CASE(512); SHQ_FFT_CASE(768);
        SHQ_FFT_CASE(1024);
    }
#undef SHQ_FFT_CASE
    return SHQ_ERR_INVALID;
}
