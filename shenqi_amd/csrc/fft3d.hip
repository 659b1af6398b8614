/* fft3d.hip — bespoke in-place 3-D real FFT pipeline for the PM mesh on gfx950.
 *
 * Replaces the reference's heffte/cuFFT r2c + c2r (libgadget/petapm.cpp:49-71) and the separate
 * transfer-function sweep (pm_apply_transfer_function, petapm.cpp:1258-1298) for mesh sizes
 * N = 2^a 3^b 5^c up to 1536 (the compiled list is in shq_fft3d_supported).  rocFFT spends 6 memory passes per 3-D transform (3 FFT + 3
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
 * Every 1-D transform is a Stockham autosort FFT in LDS (radix 16 in registers while it divides,
 * then 4, 2 and a final 3): each stage reads its butterflies' inputs into registers, barrier, writes
 * the outputs to the same buffer, barrier — one LDS buffer per line (N+1 elements: the +1
 * de-conflicts the four column lines of a tile); the twiddle table sits in LDS next to the lines.
 * Workgroups are persistent and prefetch their next tile into registers (see below).
 * Unscaled in both directions, like FFTW/heffte.
 * The z pitch of the mesh is padded to a multiple of 4 complex values so that every 4-column row
 * segment is one aligned 64-byte chunk.
 */
#include "common.hpp"
#include <math.h>
#include <stdlib.h>

namespace {

#ifndef FFT_C
#define FFT_C 4        /* complex lines per workgroup */
#endif
/* threads per workgroup: 256, two workgroups per CU, up to Nmesh 768; above that one tile + tables no longer fit the LDS twice, and a
 * 256-thread workgroup would be one wave per SIMD holding 250-500 registers: 512 threads (the same two waves per SIMD, half the tile
 * elements per thread).  N is the mesh size wherever the macro is used. */
__host__ __device__ constexpr int fft_threads(int N) { return N > 768 ? 512 : 256; }
/* waves per SIMD the tile passes of the transposing pipeline are compiled for: the Y passes run three workgroups of four waves per CU up
 * to Nmesh 768 (a tile + its 48 twiddles take 53 KB of the CU's 160, and the passes fit 168 registers: 3-4 % faster than two in a
 * same-box A/B); the X pass keeps its Green's function factors in LDS and two workgroups (with the factors read from global memory and
 * three workgroups it ran 2.44 against 2.09 ms); above 768 the workgroups have eight waves and the bound stays at two per SIMD
 * (four - a 128-register budget - made the 1200 mesh spill: 208 ms for its five passes instead of 44) */
__host__ __device__ constexpr int fft_tile_waves(int N, int MODE) { return (MODE != 2 && N <= 768) ? 3 : 2; }
#ifndef FFT_T
#define FFT_T fft_threads(N)
#endif
/* Element i of a line sits at LDS slot lx(i) = i + i / 16 (SHQ_FFT_PAD): the first radix-16 stage writes its outputs 16 elements
 * apart — 256 bytes, two full sweeps of the 32 store banks, so the eight lanes a ds_write_b128 serves per cycle all meet in the same
 * banks.  With one slot of padding per 16 elements consecutive lanes land 272 bytes apart, 16 bytes further along the banks each.
 * (SQ_LDS_BANK_CONFLICT was 60 % of SQ_LDS_IDX_ACTIVE.)  The stages split every index into a per-thread part and a compile-time
 * part without a carry between them, so that lx(t + c) = lx(t) + lx(c) and the sixteen accesses of a butterfly stay base register +
 * immediate offset (with lx(t + c) computed per access the passes need 244-269 VGPRs and lose a workgroup per CU).  The line stride
 * stays = 1 (mod 8) slots, which keeps the tile's transposed landing and leaving as they were. */
#ifndef SHQ_FFT_PAD
#define SHQ_FFT_PAD 1
#endif
/* the padding is used for the mesh sizes whose every stage splits that way (768 = 16 16 3, 1024 = 16 16 4, 1536 = 16 16 2 3, ...); the
 * others (384 = 16 4 2 3, 960 = 16 4 3 5, ...) keep the plain layout */
__host__ __device__ constexpr int fft_radix(int n) { return n % 16 == 0 ? 16 : (n % 4 == 0 ? 4 : (n % 2 == 0 ? 2 : (n % 3 == 0 ? 3 : 5))); }
__host__ __device__ constexpr bool fft_pad_ok(int N)
{
    if(!SHQ_FFT_PAD)
        return false;
    int n = N, s = 1;
    while(n > 1) {
        const int R = fft_radix(n), m = n / R;
        if((s * m) % 16 != 0 || !(s % 16 == 0 || ((s * R) % 16 == 0 && s * R <= 16)))
            return false;
        n = m;
        s *= R;
    }
    return true;
}
template <int N> __host__ __device__ constexpr int lx(int i) { return fft_pad_ok(N) ? i + (i >> 4) : i; }
/* Twiddles (round 4).  A butterfly of the stage (n, s, R) at position p multiplies its output k by w_N^(k p s).  Only w_N^(p s) is read
 * from the table, the powers follow by repeated multiplication (k <= 15: a dozen rounding errors of 1e-16 on factors of modulus one,
 * far inside the transforms' own error; every transform in the library takes its twiddles this way, so the pipelines that are
 * compared bit for bit still agree) - fifteen LDS reads and thirty registers less per radix-16 butterfly, and the table shrinks from
 * N entries to max over the stages of (m - 1) s + 1 (48 at 768): with 12 KB less LDS a tile pass fits THREE workgroups per CU. */
__host__ __device__ constexpr int fft_twn(int N)
{
    int n = N, s = 1, mx = 0;
    while(n > 1) {
        const int R = fft_radix(n), m = n / R;
        if((m - 1) * s > mx)
            mx = (m - 1) * s;
        n = m;
        s *= R;
    }
    return mx + 1;
}
__host__ __device__ constexpr int fft_ls(int N)
{
    const int last = fft_pad_ok(N) ? (N - 1) + ((N - 1) >> 4) : N - 1;
    return last + 1 + (9 - (last + 1) % 8) % 8;
}

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
struct NoLoadOp {
    __device__ __forceinline__ double2 operator()(int, int, double2 v) const { return v; }
};

template <int N, int n, int s, int R, int DIR, typename StoreOp = NoLoadOp>
__device__ __forceinline__ void fft_stage(double2 *buf, const double2 *__restrict__ W, const StoreOp op = StoreOp())
{
    constexpr int LS = fft_ls(N);
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
            /* s m j is a multiple of 16 in every stage of the 16 16 3 / 16 16 ... sequences: no carry into the padding term */
            const double2 *x = buf + line * LS + ((s * m) % 16 == 0 ? lx<N>(q + s * p) : 0);
#pragma unroll
            for(int j = 0; j < R; j++)
                v[kk][j] = x[(s * m) % 16 == 0 ? lx<N>(s * m * j) : lx<N>(q + s * p + s * m * j)];
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
            } else if(R == 5) {
                /* y_k = sum_j x_j w5^(jk), w5 = exp(DIR 2 pi i / 5) */
                const double c1 = 0.30901699437494742410, c2 = -0.80901699437494742410; /* cos 72, cos 144 */
                const double s1 = 0.95105651629515357212, s2 = 0.58778525229247312917;  /* sin 72, sin 144 */
                const double2 x0 = v[kk][0];
                const double2 t1 = cadd(v[kk][1], v[kk][4]), t2 = cadd(v[kk][2], v[kk][3]);
                const double2 t3 = csub(v[kk][1], v[kk][4]), t4 = csub(v[kk][2], v[kk][3]);
                const double2 a1 = make_double2(x0.x + c1 * t1.x + c2 * t2.x, x0.y + c1 * t1.y + c2 * t2.y);
                const double2 a2 = make_double2(x0.x + c2 * t1.x + c1 * t2.x, x0.y + c2 * t1.y + c1 * t2.y);
                const double2 b1 = make_double2(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y);
                const double2 b2 = make_double2(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y);
                /* DIR < 0: y1 = a1 - i b1, y4 = a1 + i b1, y2 = a2 - i b2, y3 = a2 + i b2; conjugated for DIR > 0 */
                const double sg = DIR < 0 ? 1.0 : -1.0;
                v[kk][0] = make_double2(x0.x + t1.x + t2.x, x0.y + t1.y + t2.y);
                v[kk][1] = make_double2(a1.x + sg * b1.y, a1.y - sg * b1.x);
                v[kk][4] = make_double2(a1.x - sg * b1.y, a1.y + sg * b1.x);
                v[kk][2] = make_double2(a2.x + sg * b2.y, a2.y - sg * b2.x);
                v[kk][3] = make_double2(a2.x - sg * b2.y, a2.y + sg * b2.x);
            } else {
                bfly16<DIR>(reinterpret_cast<double2(&)[16]>(v[kk]));
            }
            if(m > 1) { /* w_n^(p k); the last stage (m == 1) has p == 0 */
                const double2 w1 = tw<DIR>(W, p * s);
                double2 wk = w1;
                v[kk][1] = cmul(v[kk][1], wk);
#pragma unroll
                for(int k = 2; k < R; k++) {
                    wk = cmul(wk, w1);
                    v[kk][k] = cmul(v[kk][k], wk);
                }
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
            /* s k is a multiple of 16 (s % 16 == 0), or q + s k stays below 16 under a thread part that is a multiple of 16 */
            constexpr bool SPLIT = s % 16 == 0 || ((s * R) % 16 == 0 && s * R <= 16);
            double2 *y = buf + line * LS + (SPLIT ? lx<N>(s % 16 == 0 ? q + s * R * p : s * R * p) : 0);
#pragma unroll
            for(int k = 0; k < R; k++)
                y[SPLIT ? (s % 16 == 0 ? lx<N>(s * k) : q + s * k) : lx<N>(q + s * R * p + s * k)] = op(line, q + s * R * p + s * k, v[kk][k]);
        }
    }
    __syncthreads();
}

/* radix sequence: 16 while possible, then 4, 2, and a final 3 */
template <int N, int n, int s, int DIR> struct Stages {
    template <typename StoreOp> static __device__ __forceinline__ void run(double2 *buf, const double2 *__restrict__ W, const StoreOp op)
    {
        constexpr int R = n % 16 == 0 ? 16 : (n % 4 == 0 ? 4 : (n % 2 == 0 ? 2 : (n % 3 == 0 ? 3 : 5)));
        static_assert(n % R == 0, "mesh size must be 2^a 3^b 5^c");
        if constexpr(n / R > 1) {
            fft_stage<N, n, s, R, DIR>(buf, W);
            Stages<N, n / R, s * R, DIR>::run(buf, W, op);
        } else
            fft_stage<N, n, s, R, DIR, StoreOp>(buf, W, op); /* the hook filters the outputs of the last stage */
    }
};

/* op(line, index, value) filters every output element as the last stage stores it */
template <int N, int DIR, typename StoreOp = NoLoadOp>
__device__ __forceinline__ void fft_lines(double2 *buf, const double2 *__restrict__ W, const StoreOp op = StoreOp())
{
#ifdef SHQ_FFT_PROBE_NOCOMPUTE /* timing probe only (wrong results): what a pass costs without its transforms */
    __syncthreads();
#else
    Stages<N, N, 1, DIR>::run(buf, W, op);
#endif
}

/* ---- persistent workgroups -------------------------------------------------------------------------
 * Every pass runs as many workgroups as fit on the chip at once; workgroup b takes tiles b, b + G,
 * b + 2G ...  The twiddle table lives in LDS behind the line buffers (loaded once per workgroup), so
 * the FFT stages touch no global memory at all, and the NEXT tile is fetched into registers right
 * after the current one has landed in LDS: its HBM latency hides behind the LDS work.  (Measured
 * before: 70-80 % of wave cycles waiting, 15 dependent twiddle loads per radix-16 stage.)
 * The prefetch of the last iteration re-reads the workgroup's own tile instead of branching around
 * the loads, which keeps the loop-carried registers free of copies (and of an early s_waitcnt). */
template <int N> __device__ __forceinline__ double2 *lds_twiddles(double2 *buf, const double2 *__restrict__ W)
{
    double2 *Wl = buf + FFT_C * fft_ls(N);
    for(int i = threadIdx.x; i < fft_twn(N); i += FFT_T)
        Wl[i] = W[i];
    return Wl;
}

/* ---- pass Z forward: two real rows -> two half spectra, in place ---------------------------------
 * mesh: [nrows][zp] doubles (zp = pitch, >= N + 2).  Workgroup tile = FFT_C complex lines = 2 FFT_C rows. */
template <int N, bool FROM_I64>
__global__ __launch_bounds__(FFT_T) void fft_pass_z_fwd(double *mesh, const int ntot, const int zp,
                                                        const double2 *__restrict__ W, const double inv_scale)
{
    extern __shared__ double2 buf[];
    constexpr int LS = fft_ls(N), H = N / 2, Nc = N / 2 + 1;
    constexpr int E = (FFT_C * H + FFT_T - 1) / FFT_T; /* a thread takes the elements z, z + 1 of BOTH real rows of a line: see fft_t_z_fwd */
    constexpr bool EXACT = E * FFT_T == FFT_C * H;
    double2 *Wl = lds_twiddles<N>(buf, W);
    double2 *cm = reinterpret_cast<double2 *>(mesh);
    const int zpc = zp / 2;
    double pa[E], pb[E], pc[E], pd[E];
#define FFT_FETCH(T_)                                                                            \
    _Pragma("unroll") for(int i = 0; i < E; i++)                                                 \
    {                                                                                            \
        const int e_ = threadIdx.x + i * FFT_T, e = (EXACT || e_ < FFT_C * H) ? e_ : 0; /* no branch around the loads: see fft_t_z_inv */ \
        { \
            const int l_ = e / H;                                                                \
            const long long row = (long long) (T_) * (2 * FFT_C) + 2 * l_;                       \
            const double2 v_ = cm[row * zpc + (e - l_ * H)], w_ = cm[(row + 1) * zpc + (e - l_ * H)]; \
            pa[i] = v_.x;                                                                        \
            pb[i] = v_.y;                                                                        \
            pc[i] = w_.x;                                                                        \
            pd[i] = w_.y;                                                                        \
        }                                                                                        \
    }
    int t = blockIdx.x;
    if(t >= ntot)
        return;
    FFT_FETCH(t)
    while(true) {
#pragma unroll
        for(int i = 0; i < E; i++) {
            const int e = threadIdx.x + i * FFT_T;
            if(EXACT || e < FFT_C * H) {
                const int l = e / H, z = 2 * (e - l * H);
                double a = pa[i], b = pb[i], c = pc[i], d = pd[i];
                if(FROM_I64) {
                    a = (double) __double_as_longlong(a) * inv_scale;
                    b = (double) __double_as_longlong(b) * inv_scale;
                    c = (double) __double_as_longlong(c) * inv_scale;
                    d = (double) __double_as_longlong(d) * inv_scale;
                }
                buf[l * LS + lx<N>(z)] = make_double2(a, c);
                buf[l * LS + lx<N>(z + 1)] = make_double2(b, d);
            }
        }
        __syncthreads();
        const int tn = t + (int) gridDim.x;
        const bool more = tn < ntot;
        const int tf = more ? tn : t;
        FFT_FETCH(tf)
        fft_lines<N, -1>(buf, Wl);
        /* separate the two real transforms: XA[k] = (Z[k] + conj Z[N-k]) / 2, XB[k] = -i (Z[k] - conj Z[N-k]) / 2 */
        const long long row0 = (long long) t * (2 * FFT_C);
        for(int e = threadIdx.x; e < FFT_C * Nc; e += FFT_T) {
            const int l = e / Nc, k = e - l * Nc;
            const double2 zk = buf[l * LS + lx<N>(k)];
            const double2 zn = conj2(buf[l * LS + lx<N>(k == 0 ? 0 : N - k)]);
            const double2 xa = make_double2(0.5 * (zk.x + zn.x), 0.5 * (zk.y + zn.y));
            const double2 d = make_double2(0.5 * (zk.x - zn.x), 0.5 * (zk.y - zn.y));
            const double2 xb = make_double2(d.y, -d.x);
            const long long ra = row0 + 2 * l;
            cm[ra * zpc + k] = xa;
            cm[(ra + 1) * zpc + k] = xb;
        }
        if(!more)
            break;
        __syncthreads(); /* everyone has read its results out of LDS before the next tile lands there */
        t = tn;
    }
#undef FFT_FETCH
}

/* ---- pass Z inverse (c2r): two half spectra -> two real rows, in place -------------------------------- */
template <int N>
__global__ __launch_bounds__(FFT_T) void fft_pass_z_inv(double *mesh, const int ntot, const int zp, const double2 *__restrict__ W)
{
    extern __shared__ double2 buf[];
    constexpr int LS = fft_ls(N), H = N / 2, Nc = N / 2 + 1;
    constexpr int E = (FFT_C * Nc + FFT_T - 1) / FFT_T;
    double2 *Wl = lds_twiddles<N>(buf, W);
    double2 *cm = reinterpret_cast<double2 *>(mesh);
    const int zpc = zp / 2;
    double ax[E], ay[E], bx[E], by[E];
#define FFT_FETCH(T_)                                                                            \
    _Pragma("unroll") for(int i = 0; i < E; i++)                                                 \
    {                                                                                            \
        /* no branch around the loads (an unused slot reads the tile's first entry): see fft_t_z_inv */ \
        const int e_ = threadIdx.x + i * FFT_T, e = e_ < FFT_C * Nc ? e_ : 0;                    \
        const int l = e / Nc, k = e - l * Nc;                                                    \
        const long long ra = (long long) (T_) * (2 * FFT_C) + 2 * l;                             \
        const double2 xa_ = cm[ra * zpc + k], xb_ = cm[(ra + 1) * zpc + k];                      \
        ax[i] = xa_.x;                                                                           \
        ay[i] = xa_.y;                                                                           \
        bx[i] = xb_.x;                                                                           \
        by[i] = xb_.y;                                                                           \
    }
    int t = blockIdx.x;
    if(t >= ntot)
        return;
    FFT_FETCH(t)
    while(true) {
#pragma unroll
        for(int i = 0; i < E; i++) {
            const int e = threadIdx.x + i * FFT_T;
            asm volatile("" ::"v"(ax[i]), "v"(ay[i]), "v"(bx[i]), "v"(by[i])); /* every loaded half stays alive up to here: see fft_t_z_inv */
            if(e < FFT_C * Nc) {
                const int l = e / Nc, k = e - l * Nc;
                double2 xa = make_double2(ax[i], ay[i]), xb = make_double2(bx[i], by[i]);
                if(k == 0 || 2 * k == N) { /* a c2r transform ignores the imaginary part of the self-conjugate modes */
                    xa.y = 0;
                    xb.y = 0;
                }
                /* Z[k] = XA[k] + i XB[k];  Z[N-k] = conj(XA[k]) + i conj(XB[k]) */
                buf[l * LS + lx<N>(k)] = make_double2(xa.x - xb.y, xa.y + xb.x);
                if(k > 0 && 2 * k < N)
                    buf[l * LS + lx<N>(N - k)] = make_double2(xa.x + xb.y, -xa.y + xb.x);
            }
        }
        __syncthreads();
        const int tn = t + (int) gridDim.x;
        const bool more = tn < ntot;
        const int tf = more ? tn : t;
        FFT_FETCH(tf)
        fft_lines<N, +1>(buf, Wl);
        const long long row0 = (long long) t * (2 * FFT_C);
        /* a thread takes the elements z, z + 1 of one line (two 16-byte LDS reads, consecutive lanes 32 bytes apart: no bank conflicts;
         * the 8-byte reads of one row's real parts, 32 bytes apart, met four to a bank) and stores them to the line's two real rows */
        for(int e = threadIdx.x; e < FFT_C * H; e += FFT_T) {
            const int l = e / H, zz = e - l * H;
            const double2 u0 = buf[l * LS + lx<N>(2 * zz)], u1 = buf[l * LS + lx<N>(2 * zz + 1)];
            cm[(row0 + 2 * l) * zpc + zz] = make_double2(u0.x, u1.x);
            cm[(row0 + 2 * l + 1) * zpc + zz] = make_double2(u0.y, u1.y);
        }
        if(!more)
            break;
        __syncthreads();
        t = tn;
    }
#undef FFT_FETCH
}

/* ---- passes Y and X: complex lines with element stride `es` (in complex units), FFT_C adjacent columns.
 * MODE 0: forward only; 1: inverse only; 2: forward, Green's function, inverse (X pass of the PM). */
struct GreenArgs {
    const double *sinctab; /* 1 / sinc^2(pi k / N) per mesh index */
    const double *gaxg;    /* transposing pipeline: exp(-k_i^2 asmth2) sinctab[i]^2 per mesh index, in global memory (fft_gax_kernel) */
    double asmth2, pot_factor;
    int y0;                /* mesh index of outer = 0 in the X pass (y-slab of a distributed mesh) */
    /* PK != 0 (Y pass of a distributed mesh): the transposed side of the pass lives in `alt`, laid out as the all-to-all wants it,
     * [dest rank q][x plane][y inside q's slab][z']: line element y sits at alt + (y / nyl) qstride + outer alt_outer + (y % nyl) es */
    double2 *alt;
    int nyl;
    long long qstride, alt_outer;
};

/* PK 0: in place.  PK 1: results stored into ga.alt in the packed (send) layout.  PK 2: input loaded from ga.alt in that layout. */
template <int N, int MODE, int PK = 0>
__global__ __launch_bounds__(FFT_T) void fft_pass_strided(double2 *cm, const long long es, const long long outer_stride,
                                                          const int ntiles, const int ntot, const double2 *__restrict__ W,
                                                          const GreenArgs ga, const unsigned xcdk, const int tile0 = 0)
{
    extern __shared__ double2 buf[];
    constexpr int LS = fft_ls(N);
    constexpr int E = (FFT_C * N + FFT_T - 1) / FFT_T; /* tile elements per thread */
    constexpr bool EXACT = E * FFT_T == FFT_C * N;
    double2 *Wl = lds_twiddles<N>(buf, W);
    double *gax = reinterpret_cast<double *>(Wl + fft_twn(N)); /* MODE 2 only: per-axis factor of the Green's function */
    if(MODE == 2)
        for(int i = threadIdx.x; i < N; i += FFT_T) {
            const int k = i <= N / 2 ? i : i - N;
            const double sc = ga.sinctab[i];
            gax[i] = exp(-(double) k * (double) k * ga.asmth2) * sc * sc;
        }
    /* XCD-chunked workgroup order: neighbouring column tiles share 128-byte lines (a tile row is 64
     * bytes), so they should run on the same XCD at about the same time and find the other half in its L2. */
    const unsigned vb = xcd_block(blockIdx.x, gridDim.x, xcdk);
    double prx[E], pry[E];
#define FFT_FETCH(BASE, ABASE)                                                                   \
    _Pragma("unroll") for(int i = 0; i < E; i++)                                                 \
    {                                                                                            \
        const int e_ = threadIdx.x + i * FFT_T, e = (EXACT || e_ < FFT_C * N) ? e_ : 0; /* no branch around the loads: see fft_t_z_inv */ \
        { \
            const int r_ = e / FFT_C, c_ = e % FFT_C;                                            \
            const double2 t_ = PK == 2 ? (ABASE)[(long long) (r_ / ga.nyl) * ga.qstride + (long long) (r_ % ga.nyl) * es + c_] \
                                       : (BASE)[(long long) r_ * es + c_];                       \
            prx[i] = t_.x;                                                                       \
            pry[i] = t_.y;                                                                       \
        }                                                                                        \
    }
    int t = (int) vb;
    if(t >= ntot)
        return;
    /* a launch covers the column tiles [tile0, tile0 + ntiles) of every outer index (all of them, or one z' chunk of the chained
     * Y -> X -> Y sequence below) */
    int outer = t / ntiles, tile = tile0 + (t - outer * ntiles);
    double2 *base = cm + (long long) outer * outer_stride + (long long) tile * FFT_C;
    double2 *abase = PK ? ga.alt + (long long) outer * ga.alt_outer + (long long) tile * FFT_C : nullptr;
    FFT_FETCH(base, abase)
    while(true) {
#pragma unroll
        for(int i = 0; i < E; i++) {
            const int e = threadIdx.x + i * FFT_T;
            if(EXACT || e < FFT_C * N)
                buf[(e % FFT_C) * LS + lx<N>(e / FFT_C)] = make_double2(prx[i], pry[i]);
        }
        __syncthreads();
        const int tn = t + (int) gridDim.x;
        const bool more = tn < ntot;
        const int outer_n = more ? tn / ntiles : outer, tile_n = more ? tile0 + (tn - outer_n * ntiles) : tile;
        double2 *base_n = cm + (long long) outer_n * outer_stride + (long long) tile_n * FFT_C;
        double2 *abase_n = PK ? ga.alt + (long long) outer_n * ga.alt_outer + (long long) tile_n * FFT_C : nullptr;
        FFT_FETCH(base_n, abase_n)
        if(MODE == 0)
            fft_lines<N, -1>(buf, Wl);
        if(MODE == 2) {
            /* potential_transfer, gravpm.cpp:378-444, applied as the forward transform stores its last
             * stage (line index = kx, outer = y, column = z'); gax[i] = exp(-k_i^2 asmth2) sinctab[i]^2, so
             * the factor exp(-k^2 asmth2) f^2 / k^2 is a product of three table entries over k^2 */
            const int y = ga.y0 + outer, z0 = tile * FFT_C;
            const int ky = y <= N / 2 ? y : y - N;
            const double gy = gax[y] * ga.pot_factor, ky2 = (double) ky * (double) ky;
            auto green = [=](int col, int x, double2 v) {
                const int z = z0 + col;
                const int kx = x <= N / 2 ? x : x - N;
                const double k2 = (double) kx * (double) kx + (ky2 + (double) z * (double) z); /* exact: < 2^53 */
                const double fac = (k2 == 0.0 || z > N / 2) ? 0.0 : gax[x] * gy * gax[z] / k2;
                return make_double2(v.x * fac, v.y * fac);
            };
            fft_lines<N, -1>(buf, Wl, green);
            fft_lines<N, +1>(buf, Wl);
        }
        if(MODE == 1)
            fft_lines<N, +1>(buf, Wl);
        for(int e = threadIdx.x; e < FFT_C * N; e += FFT_T) {
            const int row = e / FFT_C, col = e - row * FFT_C;
            if(PK == 1)
                abase[(long long) (row / ga.nyl) * ga.qstride + (long long) (row % ga.nyl) * es + col] = buf[col * LS + lx<N>(row)];
            else
                base[(long long) row * es + col] = buf[col * LS + lx<N>(row)];
        }
        if(!more)
            break;
        __syncthreads(); /* everyone has read its results out of LDS before the next tile lands there */
        t = tn;
        outer = outer_n;
        tile = tile_n;
        base = base_n;
        abase = abase_n;
    }
#undef FFT_FETCH
}

/* ---- the transposing pipeline (round 4; the undivided PM of shq_pm_run / shq_treepm_step) ------------------------------------
 * tools/copy_probe.hip moves the 768^3 mesh in the access shapes of the passes with nothing but 16-byte loads and stores:
 * contiguous 48 KB tiles both sides 1.25 ms (5.8 TB/s); one side in pieces of 512 / 256 / 128 / 64 B 1.27 / 1.38 / 1.50 / 1.57 ms;
 * BOTH sides in 64-byte pieces 6208 B apart (the Y passes above) 1.83-1.91 ms, 37 MB apart (the X pass) 2.2-2.4 ms - the passes above run
 * at those rates already (1.65 / 2.38 ms): what bounds them is the DRAM's row activations, one per 64-byte piece, not the FFT.
 * So the mesh changes its layout from pass to pass, between the mesh A and a scratch mesh B of the same size, such that one side of
 * every pass is a contiguous tile and only two of the ten sides are left with 64-byte pieces:
 *
 *   LY = [x][zb][y][4]   tile (x, zb) = the four z' columns 4 zb .. 4 zb + 3 of plane x, all y: N x 64 B contiguous
 *   LX = [y][zb][x][4]   tile (y, zb) likewise with the lines along x
 *
 *   Z fwd   A (real rows [x][y][z], int64 deposit)  -> B in LY: 8 rows in, per z' block one 512-byte piece (8 y x 4 z') out
 *   Y fwd   B tile (x, zb) contiguous               -> A in LX: row ky of the tile is a 64-byte piece
 *   X       A tile (ky, zb) contiguous: forward, potential_transfer, inverse  -> B in LY: row x is a 64-byte piece
 *   Y inv   B tile (x, zb) contiguous               -> the same tile of B, in place
 *   Z inv   B in LY, gathered in 512-byte pieces    -> A (real rows): the potential, where the deposit was
 *
 * The arithmetic - stages, twiddles, the two-for-one separation, the Green's function - is the functions above applied to the same
 * values in the same order: the potential mesh is bit-identical to the in-place pipeline's (a test compares the two). */
template <int N, bool FROM_I64>
__global__ __launch_bounds__(FFT_T) void fft_t_z_fwd(const double *mesh, double2 *__restrict__ out, const int ntot, const int zp,
                                                     const double2 *__restrict__ W, const double inv_scale)
{
    extern __shared__ double2 buf[];
    constexpr int LS = fft_ls(N), H = N / 2, Nc = N / 2 + 1;
    constexpr int E = (FFT_C * H + FFT_T - 1) / FFT_T; /* a thread takes the elements z, z + 1 of BOTH real rows of a line (the line's */
    constexpr bool EXACT = E * FFT_T == FFT_C * H;     /* entries z, z + 1 are then two 16-byte LDS stores; 8-byte stores 32 bytes apart */
    static_assert(FFT_C == 4, "the layouts are written for tiles of 4 lines"); /* met four to a bank) */
    double2 *Wl = lds_twiddles<N>(buf, W);
    const double2 *cm = reinterpret_cast<const double2 *>(mesh);
    const int zpc = zp / 2, nzb = zpc / 4;
    double pa[E], pb[E], pc[E], pd[E];
#define FFT_FETCH(T_)                                                                            \
    _Pragma("unroll") for(int i = 0; i < E; i++)                                                 \
    {                                                                                            \
        const int e_ = threadIdx.x + i * FFT_T, e = (EXACT || e_ < FFT_C * H) ? e_ : 0; /* no branch around the loads: see fft_t_z_inv */ \
        { \
            const int l_ = e / H;                                                                \
            const long long row = (long long) (T_) * (2 * FFT_C) + 2 * l_;                       \
            const double2 v_ = cm[row * zpc + (e - l_ * H)], w_ = cm[(row + 1) * zpc + (e - l_ * H)]; \
            pa[i] = v_.x;                                                                        \
            pb[i] = v_.y;                                                                        \
            pc[i] = w_.x;                                                                        \
            pd[i] = w_.y;                                                                        \
        }                                                                                        \
    }
    int t = blockIdx.x;
    if(t >= ntot)
        return;
    FFT_FETCH(t)
    while(true) {
#pragma unroll
        for(int i = 0; i < E; i++) {
            const int e = threadIdx.x + i * FFT_T;
            if(EXACT || e < FFT_C * H) {
                const int l = e / H, z = 2 * (e - l * H);
                double a = pa[i], b = pb[i], c = pc[i], d = pd[i];
                if(FROM_I64) {
                    a = (double) __double_as_longlong(a) * inv_scale;
                    b = (double) __double_as_longlong(b) * inv_scale;
                    c = (double) __double_as_longlong(c) * inv_scale;
                    d = (double) __double_as_longlong(d) * inv_scale;
                }
                buf[l * LS + lx<N>(z)] = make_double2(a, c);
                buf[l * LS + lx<N>(z + 1)] = make_double2(b, d);
            }
        }
        __syncthreads();
        const int tn = t + (int) gridDim.x;
        const bool more = tn < ntot;
        const int tf = more ? tn : t;
        FFT_FETCH(tf)
        fft_lines<N, -1>(buf, Wl);
        /* rows 8 t .. 8 t + 7 = plane x, rows y0 .. y0 + 7 (8 divides N).  Thread -> (zb, line pair l, column c): sixteen threads fill
         * one 512-byte piece [zb][y0 .. y0 + 7][4] with two stores each (rows y0 + 2 l and y0 + 2 l + 1) */
        const long long row0 = (long long) t * (2 * FFT_C);
        const int x = (int) (row0 / N), y0 = (int) (row0 - (long long) x * N);
        for(int e = threadIdx.x; e < 16 * nzb; e += FFT_T) {
            const int zb = e >> 4, l = (e >> 2) & 3, c = e & 3, k = 4 * zb + c;
            double2 xa = make_double2(0, 0), xb = make_double2(0, 0); /* the pad columns k >= Nc hold zeros */
            if(k < Nc) {
                const double2 zk = buf[l * LS + lx<N>(k)];
                const double2 zn = conj2(buf[l * LS + lx<N>(k == 0 ? 0 : N - k)]);
                xa = make_double2(0.5 * (zk.x + zn.x), 0.5 * (zk.y + zn.y));
                const double2 d = make_double2(0.5 * (zk.x - zn.x), 0.5 * (zk.y - zn.y));
                xb = make_double2(d.y, -d.x);
            }
            double2 *o = out + (((long long) x * nzb + zb) * N + (y0 + 2 * l)) * 4 + c;
            o[0] = xa;
            o[4] = xb;
        }
        if(!more)
            break;
        __syncthreads();
        t = tn;
    }
#undef FFT_FETCH
}

template <int N>
__global__ __launch_bounds__(FFT_T) void fft_t_z_inv(const double2 *__restrict__ in, double *mesh, const int ntot, const int zp, const double2 *__restrict__ W)
{
    extern __shared__ double2 buf[];
    constexpr int LS = fft_ls(N), H = N / 2, Nc = N / 2 + 1;
    double2 *Wl = lds_twiddles<N>(buf, W);
    double2 *cm = reinterpret_cast<double2 *>(mesh);
    const int zpc = zp / 2, nzb = zpc / 4;
    constexpr int EMAX = (16 * ((N / 2 + 1 + 3) / 4) + FFT_T - 1) / FFT_T; /* 16 (zb, l, c) slots per z' block of the unpadded spectrum... */
    const int nslots = 16 * nzb;                                           /* ...the pitch may hold more blocks: those are skipped */
    double ax[EMAX], ay[EMAX], bx[EMAX], by[EMAX];
#define FFT_FETCH(T_)                                                                            \
    {                                                                                            \
        const long long row0_ = (long long) (T_) * (2 * FFT_C);                                  \
        const int x_ = (int) (row0_ / N), y0_ = (int) (row0_ - (long long) x_ * N);              \
        _Pragma("unroll") for(int i = 0; i < EMAX; i++)                                          \
        {                                                                                        \
            const int e = threadIdx.x + i * FFT_T;                                               \
            const int zb = e >> 4, l = (e >> 2) & 3, c = e & 3, k = 4 * zb + c;                  \
            /* no branch around the loads: an unused slot reads the tile's first entry instead.  With the loads under the slot's  \
             * condition the compiler ended every pair with s_waitcnt vmcnt(0) (the values pass into loop-carried registers at the \
             * join) and the "prefetch" was seven round trips to memory one after the other */                                     \
            const bool ok_ = e < nslots && k < Nc;                                               \
            const double2 *p_ = in + (((long long) x_ * nzb + (ok_ ? zb : 0)) * N + (y0_ + 2 * l)) * 4 + (ok_ ? c : 0); \
            const double2 xa_ = p_[0], xb_ = p_[4];                                              \
            ax[i] = xa_.x;                                                                       \
            ay[i] = xa_.y;                                                                       \
            bx[i] = xb_.x;                                                                       \
            by[i] = xb_.y;                                                                       \
        }                                                                                        \
    }
    int t = blockIdx.x;
    if(t >= ntot)
        return;
    FFT_FETCH(t)
    while(true) {
#pragma unroll
        for(int i = 0; i < EMAX; i++) {
            const int e = threadIdx.x + i * FFT_T;
            const int zb = e >> 4, l = (e >> 2) & 3, c = e & 3, k = 4 * zb + c;
            /* all four components stay alive up to here: the last slot group's imaginary parts are never used (its only column is
             * N / 2), and the compiler handed their halves of the in-flight 16-byte loads to the first stage's temporaries - a write
             * after write that made the stage wait for the whole prefetch */
            asm volatile("" ::"v"(ax[i]), "v"(ay[i]), "v"(bx[i]), "v"(by[i]));
            if(e < nslots && k < Nc) {
                double2 xa = make_double2(ax[i], ay[i]), xb = make_double2(bx[i], by[i]);
                if(k == 0 || 2 * k == N) {
                    xa.y = 0;
                    xb.y = 0;
                }
                buf[l * LS + lx<N>(k)] = make_double2(xa.x - xb.y, xa.y + xb.x);
                if(k > 0 && 2 * k < N)
                    buf[l * LS + lx<N>(N - k)] = make_double2(xa.x + xb.y, -xa.y + xb.x);
            }
        }
        __syncthreads();
        const int tn = t + (int) gridDim.x;
        const bool more = tn < ntot;
        const int tf = more ? tn : t;
        FFT_FETCH(tf)
        fft_lines<N, +1>(buf, Wl);
        const long long row0 = (long long) t * (2 * FFT_C);
        /* a thread takes the elements z, z + 1 of one line (two 16-byte LDS reads, consecutive lanes 32 bytes apart: no bank conflicts;
         * the 8-byte reads of one row's real parts, 32 bytes apart, met four to a bank) and stores them to the line's two real rows */
        for(int e = threadIdx.x; e < FFT_C * H; e += FFT_T) {
            const int l = e / H, zz = e - l * H;
            const double2 u0 = buf[l * LS + lx<N>(2 * zz)], u1 = buf[l * LS + lx<N>(2 * zz + 1)];
            cm[(row0 + 2 * l) * zpc + zz] = make_double2(u0.x, u1.x);
            cm[(row0 + 2 * l + 1) * zpc + zz] = make_double2(u0.y, u1.y);
        }
        if(!more)
            break;
        __syncthreads();
        t = tn;
    }
#undef FFT_FETCH
}

/* Y and X passes on contiguous tiles.  Tile t = (o, zb), o = t / nzb: N rows of 4 columns at src + t * 4 N.  MODE as in fft_pass_strided.
 * SCATTER: row i of the result goes to dst + ((i * nzb + zb) * N + o) * 4 (the other layout: a 64-byte piece); otherwise the tile is
 * written back where it came from (dst may be src). */
template <int N, int MODE, bool SCATTER>
__global__ __launch_bounds__(FFT_T, fft_tile_waves(N, MODE)) void fft_t_tile(const double2 *src, double2 *dst, const int nzb, const int ntot,
                                                                       const double2 *__restrict__ W, const GreenArgs ga, const unsigned xcdk)
{
    extern __shared__ double2 buf[];
    constexpr int LS = fft_ls(N);
    constexpr int E = (FFT_C * N + FFT_T - 1) / FFT_T;
    constexpr bool EXACT = E * FFT_T == FFT_C * N;
    double2 *Wl = lds_twiddles<N>(buf, W);
    double *gax = reinterpret_cast<double *>(Wl + fft_twn(N)); /* MODE 2: the Green's function's per-axis factors (fft_gax_kernel) in LDS */
    if(MODE == 2) {
        for(int i = threadIdx.x; i < N; i += FFT_T)
            gax[i] = ga.gaxg[i];
        __syncthreads();
    }
    const unsigned vb = xcd_block(blockIdx.x, gridDim.x, xcdk);
    double prx[E], pry[E];
#define FFT_FETCH(T_)                                                                            \
    {                                                                                            \
        const double2 *b_ = src + (long long) (T_) * (FFT_C * N);                                \
        _Pragma("unroll") for(int i = 0; i < E; i++)                                             \
        {                                                                                        \
            const int e_ = threadIdx.x + i * FFT_T, e = (EXACT || e_ < FFT_C * N) ? e_ : 0; /* no branch around the loads: see fft_t_z_inv */ \
            { \
                const double2 t_ = b_[e];                                                        \
                prx[i] = t_.x;                                                                   \
                pry[i] = t_.y;                                                                   \
            }                                                                                    \
        }                                                                                        \
    }
    int t = (int) vb;
    if(t >= ntot)
        return;
    FFT_FETCH(t)
    while(true) {
#pragma unroll
        for(int i = 0; i < E; i++) {
            const int e = threadIdx.x + i * FFT_T;
            if(EXACT || e < FFT_C * N)
                buf[(e % FFT_C) * LS + lx<N>(e / FFT_C)] = make_double2(prx[i], pry[i]);
        }
        __syncthreads();
        const int tn = t + (int) gridDim.x;
        const bool more = tn < ntot;
        const int tf = more ? tn : t;
        FFT_FETCH(tf)
        const int o = t / nzb, zb = t - o * nzb;
        if(MODE == 0)
            fft_lines<N, -1>(buf, Wl);
        if(MODE == 2) { /* potential_transfer as in fft_pass_strided: line index = kx, o = ky, columns z' = 4 zb .. 4 zb + 3 */
            const int y = ga.y0 + o, z0 = zb * FFT_C;
            const int ky = y <= N / 2 ? y : y - N;
            const double gy = gax[y] * ga.pot_factor, ky2 = (double) ky * (double) ky;
            auto green = [=](int col, int x, double2 v) {
                const int z = z0 + col;
                const int kx = x <= N / 2 ? x : x - N;
                const double k2 = (double) kx * (double) kx + (ky2 + (double) z * (double) z);
                const double fac = (k2 == 0.0 || z > N / 2) ? 0.0 : gax[x] * gy * gax[z] / k2;
                return make_double2(v.x * fac, v.y * fac);
            };
            fft_lines<N, -1>(buf, Wl, green);
            fft_lines<N, +1>(buf, Wl);
        }
        if(MODE == 1)
            fft_lines<N, +1>(buf, Wl);
        if(SCATTER) {
            double2 *ob = dst + ((long long) zb * N + o) * FFT_C;
            const long long rs = (long long) nzb * N * FFT_C;
            for(int e = threadIdx.x; e < FFT_C * N; e += FFT_T) {
                const int row = e / FFT_C, col = e - row * FFT_C;
                ob[(long long) row * rs + col] = buf[col * LS + lx<N>(row)];
            }
        } else {
            double2 *ob = dst + (long long) t * (FFT_C * N);
            for(int e = threadIdx.x; e < FFT_C * N; e += FFT_T)
                ob[e] = buf[(e % FFT_C) * LS + lx<N>(e / FFT_C)];
        }
        if(!more)
            break;
        __syncthreads();
        t = tn;
    }
#undef FFT_FETCH
}

/* exp(-k_i^2 asmth2) sinctab[i]^2 per mesh index: the expression the in-place X pass fills its LDS table with */
__global__ void fft_gax_kernel(int N, const double *__restrict__ sinctab, double asmth2, double *gax)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < N) {
        const int k = i <= N / 2 ? i : i - N;
        const double sc = sinctab[i];
        gax[i] = exp(-(double) k * (double) k * asmth2) * sc * sc;
    }
}

template <int N>
int run_t(shq_context *ctx, double *d_mesh, double *d_scratch, int zp, bool from_i64, double inv_scale, const GreenArgs &ga)
{
    const double2 *W = reinterpret_cast<const double2 *>(ctx->fft_tw.ptr);
    constexpr size_t lds = sizeof(double2) * (FFT_C * fft_ls(N) + fft_twn(N)), lds_x = lds + sizeof(double) * N; /* X pass: + its factor table */
    const int ztot = (int) (((long long) N * N) / (2 * FFT_C));
    const int zpc = zp / 2, nzb = zpc / FFT_C;
    const int stot = N * nzb;
    double2 *A = reinterpret_cast<double2 *>(d_mesh), *B = reinterpret_cast<double2 *>(d_scratch);
    hipStream_t s = ctx->stream;
    static unsigned res_z = 0, res_s = 0, res_x = 0;
    if(res_s == 0) {
        const void *fns[6] = {(const void *) fft_t_z_fwd<N, true>, (const void *) fft_t_z_fwd<N, false>, (const void *) fft_t_z_inv<N>,
                              (const void *) fft_t_tile<N, 0, true>, (const void *) fft_t_tile<N, 2, true>, (const void *) fft_t_tile<N, 1, false>};
        int ncu = 0;
        if(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess || ncu < 1)
            ncu = 256;
        unsigned occ[6];
        for(int i = 0; i < 6; i++) {
            const size_t l = i == 4 ? lds_x : lds;
            if(l > 48 * 1024)
                SHQ_HIP(hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int) l));
            int per_cu = 0;
            if(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fns[i], FFT_T, l) != hipSuccess || per_cu < 1)
                per_cu = 1;
            occ[i] = (unsigned) per_cu * (unsigned) ncu;
        }
        res_z = occ[0] < occ[1] ? occ[0] : occ[1];
        res_z = res_z < occ[2] ? res_z : occ[2];
        res_s = occ[3] < occ[5] ? occ[3] : occ[5];
        res_x = occ[4];
    }
    const unsigned gmul = getenv("SHQ_FFT_GRID_MUL") ? (unsigned) atoi(getenv("SHQ_FFT_GRID_MUL")) : 8u;
    auto grid = [&](int tot, unsigned resident) {
        const unsigned cap = gmul == 0 ? (unsigned) tot : resident * gmul;
        return dim3((unsigned) tot < cap ? (unsigned) tot : cap);
    };
    const dim3 gz = grid(ztot, res_z), gs = grid(stot, res_s), gx = grid(stot, res_x);
    const unsigned xcdk = getenv("SHQ_FFT_XCD_K") ? (unsigned) atoi(getenv("SHQ_FFT_XCD_K")) : 8u;
    if(from_i64)
        fft_t_z_fwd<N, true><<<gz, dim3(FFT_T), lds, s>>>(d_mesh, B, ztot, zp, W, inv_scale);
    else
        fft_t_z_fwd<N, false><<<gz, dim3(FFT_T), lds, s>>>(d_mesh, B, ztot, zp, W, 1.0);
    fft_t_tile<N, 0, true><<<gs, dim3(FFT_T), lds, s>>>(B, A, nzb, stot, W, ga, xcdk);
    fft_t_tile<N, 2, true><<<gx, dim3(FFT_T), lds_x, s>>>(A, B, nzb, stot, W, ga, xcdk);
    fft_t_tile<N, 1, false><<<gs, dim3(FFT_T), lds, s>>>(B, B, nzb, stot, W, ga, xcdk);
    fft_t_z_inv<N><<<gz, dim3(FFT_T), lds, s>>>(B, d_mesh, ztot, zp, W);
    SHQ_HIP(hipGetLastError());
    return SHQ_OK;
}

/* stage 0 / 1 / 2 as shq_fft3d_run on a full cube (nslab = N).  Slab stages for a distributed mesh:
 * 10: Z forward + Y forward on `nslab` x-planes [nslab][N][zp];  11: Y inverse + Z inverse on them;
 * 12: X forward + potential_transfer + X inverse on a y-slab [N][nslab][zpc] (lines along the slowest axis). */
template <int N>
int run_n(shq_context *ctx, double *d_mesh, int zp, int stage, bool from_i64, double inv_scale, const GreenArgs &ga, int nslab)
{
    const double2 *W = reinterpret_cast<const double2 *>(ctx->fft_tw.ptr);
    /* FFT_C padded lines + the twiddle table + (X pass) the sinc table */
    constexpr size_t lds = sizeof(double2) * (FFT_C * fft_ls(N) + fft_twn(N)) + sizeof(double) * N;
#ifndef SHQ_FFT_RELAX /* tile-shape experiments on one mesh size */
    static_assert(N % (2 * FFT_C) == 0, "rows must tile evenly");
#endif
    const int ztot = (int) (((long long) nslab * N) / (2 * FFT_C)); /* row groups of the Z passes */
    const int zpc = zp / 2;
    const int ntiles = zpc / FFT_C;
    const int stot = nslab * ntiles;                                  /* tiles of a strided pass */
    double2 *cm = reinterpret_cast<double2 *>(d_mesh);
    hipStream_t s = ctx->stream;
    /* persistent grids: as many workgroups as are resident on the chip at once (LDS-limited) */
    static unsigned res_zf = 0, res_zi = 0, res_s = 0;
    if(res_s == 0) {
        const void *fns[8] = {(const void *) fft_pass_z_fwd<N, true>, (const void *) fft_pass_z_fwd<N, false>,
                              (const void *) fft_pass_z_inv<N>,       (const void *) fft_pass_strided<N, 0>,
                              (const void *) fft_pass_strided<N, 1>,  (const void *) fft_pass_strided<N, 2>,
                              (const void *) fft_pass_strided<N, 0, 1>, (const void *) fft_pass_strided<N, 1, 2>};
        int ncu = 0;
        if(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess || ncu < 1)
            ncu = 256;
        unsigned occ[8];
        for(int i = 0; i < 8; i++) {
            if(lds > 48 * 1024) /* allow > 48 KB of dynamic LDS */
                SHQ_HIP(hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
            int per_cu = 0;
            if(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fns[i], FFT_T, lds) != hipSuccess || per_cu < 1)
                per_cu = 1;
            occ[i] = (unsigned) per_cu * (unsigned) ncu;
        }
        res_zf = occ[0] < occ[1] ? occ[0] : occ[1];
        res_zi = occ[2];
        res_s = occ[3] < occ[4] ? occ[3] : occ[4];
        res_s = res_s < occ[5] ? res_s : occ[5];
        res_s = res_s < occ[6] ? res_s : occ[6];
        res_s = res_s < occ[7] ? res_s : occ[7];
    }
    const unsigned gmul = getenv("SHQ_FFT_GRID_MUL") ? (unsigned) atoi(getenv("SHQ_FFT_GRID_MUL")) : 8u;
    auto grid = [&](int tot, unsigned resident) {
        const unsigned cap = gmul == 0 ? (unsigned) tot : resident * gmul;
        return dim3((unsigned) tot < cap ? (unsigned) tot : cap);
    };
    const dim3 gzf = grid(ztot, res_zf), gzi = grid(ztot, res_zi), gs = grid(stot, res_s);
    const unsigned xcdk = getenv("SHQ_FFT_XCD_K") ? (unsigned) atoi(getenv("SHQ_FFT_XCD_K")) : 8u;
    if(stage == 12) { /* lines along x of a y-slab: element stride nslab * zpc, outer = local y */
        fft_pass_strided<N, 2><<<gs, dim3(FFT_T), lds, s>>>(cm, (long long) nslab * zpc, zpc, ntiles, stot, W, ga, xcdk);
        SHQ_HIP(hipGetLastError());
        return SHQ_OK;
    }
    if(stage == 13 || stage == 14) { /* stages 10 / 11 with the Y pass writing / reading the all-to-all layout in ga.alt */
        if(stage == 13) {
            if(from_i64)
                fft_pass_z_fwd<N, true><<<gzf, dim3(FFT_T), lds, s>>>(d_mesh, ztot, zp, W, inv_scale);
            else
                fft_pass_z_fwd<N, false><<<gzf, dim3(FFT_T), lds, s>>>(d_mesh, ztot, zp, W, 1.0);
            fft_pass_strided<N, 0, 1><<<gs, dim3(FFT_T), lds, s>>>(cm, zpc, (long long) N * zpc, ntiles, stot, W, ga, xcdk);
        } else {
            fft_pass_strided<N, 1, 2><<<gs, dim3(FFT_T), lds, s>>>(cm, zpc, (long long) N * zpc, ntiles, stot, W, ga, xcdk);
            fft_pass_z_inv<N><<<gzi, dim3(FFT_T), lds, s>>>(d_mesh, ztot, zp, W);
        }
        SHQ_HIP(hipGetLastError());
        return SHQ_OK;
    }
    if(stage == 10 || stage == 11) {
        if(stage == 10) {
            if(from_i64)
                fft_pass_z_fwd<N, true><<<gzf, dim3(FFT_T), lds, s>>>(d_mesh, ztot, zp, W, inv_scale);
            else
                fft_pass_z_fwd<N, false><<<gzf, dim3(FFT_T), lds, s>>>(d_mesh, ztot, zp, W, 1.0);
            fft_pass_strided<N, 0><<<gs, dim3(FFT_T), lds, s>>>(cm, zpc, (long long) N * zpc, ntiles, stot, W, ga, xcdk);
        } else {
            fft_pass_strided<N, 1><<<gs, dim3(FFT_T), lds, s>>>(cm, zpc, (long long) N * zpc, ntiles, stot, W, ga, xcdk);
            fft_pass_z_inv<N><<<gzi, dim3(FFT_T), lds, s>>>(d_mesh, ztot, zp, W);
        }
        SHQ_HIP(hipGetLastError());
        return SHQ_OK;
    }
    /* Chained middle passes (stage 2, full cube; SHQ_FFT_CHAIN = column tiles per chunk, off by default): Y forward, the fused X
     * pass and Y inverse all work on lines inside one plane of constant z', so they can run chunk by chunk over z' (a chunk of c
     * tiles is c x 38 MB at 768^3) in the hope that a chunk written by one pass is still in the 256 MB Infinity Cache when the next
     * pass reads it.  Measured at 768^3 (one box, FFT pipeline in ms): unchained 9.83; chunks of 2 / 3 / 4 / 6 / 8 tiles 11.85 /
     * 11.20 / 11.03 / 10.59 / 10.56 — the cache does not absorb the write-then-read traffic of these passes, and every chunk pays
     * the start and the tail of three more launches.  Kept as a diagnostic knob only. */
    static const int chain_env = getenv("SHQ_FFT_CHAIN") ? atoi(getenv("SHQ_FFT_CHAIN")) : 0;
    const int chain = chain_env;
    if(stage == 2 && nslab == N && chain > 0 && chain < ntiles) {
        if(from_i64)
            fft_pass_z_fwd<N, true><<<gzf, dim3(FFT_T), lds, s>>>(d_mesh, ztot, zp, W, inv_scale);
        else
            fft_pass_z_fwd<N, false><<<gzf, dim3(FFT_T), lds, s>>>(d_mesh, ztot, zp, W, 1.0);
        for(int t0 = 0; t0 < ntiles; t0 += chain) {
            const int tw = t0 + chain <= ntiles ? chain : ntiles - t0;
            const int ctot = N * tw;
            const dim3 gc = grid(ctot, res_s);
            fft_pass_strided<N, 0><<<gc, dim3(FFT_T), lds, s>>>(cm, zpc, (long long) N * zpc, tw, ctot, W, ga, xcdk, t0);
            fft_pass_strided<N, 2><<<gc, dim3(FFT_T), lds, s>>>(cm, (long long) N * zpc, zpc, tw, ctot, W, ga, xcdk, t0);
            fft_pass_strided<N, 1><<<gc, dim3(FFT_T), lds, s>>>(cm, zpc, (long long) N * zpc, tw, ctot, W, ga, xcdk, t0);
        }
        fft_pass_z_inv<N><<<gzi, dim3(FFT_T), lds, s>>>(d_mesh, ztot, zp, W);
        SHQ_HIP(hipGetLastError());
        return SHQ_OK;
    }
    if(stage == 0 || stage == 2) {
        if(from_i64)
            fft_pass_z_fwd<N, true><<<gzf, dim3(FFT_T), lds, s>>>(d_mesh, ztot, zp, W, inv_scale);
        else
            fft_pass_z_fwd<N, false><<<gzf, dim3(FFT_T), lds, s>>>(d_mesh, ztot, zp, W, 1.0);
        /* Y: outer = x plane (stride N*zpc), element stride zpc */
        fft_pass_strided<N, 0><<<gs, dim3(FFT_T), lds, s>>>(cm, zpc, (long long) N * zpc, ntiles, stot, W, ga, xcdk);
    }
    /* X: outer = y (stride zpc), element stride N*zpc */
    if(stage == 0)
        fft_pass_strided<N, 0><<<gs, dim3(FFT_T), lds, s>>>(cm, (long long) N * zpc, zpc, ntiles, stot, W, ga, xcdk);
    else if(stage == 1)
        fft_pass_strided<N, 1><<<gs, dim3(FFT_T), lds, s>>>(cm, (long long) N * zpc, zpc, ntiles, stot, W, ga, xcdk);
    else
        fft_pass_strided<N, 2><<<gs, dim3(FFT_T), lds, s>>>(cm, (long long) N * zpc, zpc, ntiles, stot, W, ga, xcdk);
    if(stage == 1 || stage == 2) {
        fft_pass_strided<N, 1><<<gs, dim3(FFT_T), lds, s>>>(cm, zpc, (long long) N * zpc, ntiles, stot, W, ga, xcdk);
        fft_pass_z_inv<N><<<gzi, dim3(FFT_T), lds, s>>>(d_mesh, ztot, zp, W);
    }
    SHQ_HIP(hipGetLastError());
    return SHQ_OK;
}

} // namespace

/* mesh sizes with a compiled pipeline (multiples of 8 of the form 2^a 3^b 5^c) */
bool shq_fft3d_supported(int N)
{
    switch(N) {
    case 16: case 24: case 32: case 40: case 48: case 64: case 80: case 96: case 128: case 192: case 256: case 384: case 512: case 768:
    case 960: case 1024: case 1152: case 1200: case 1536:
        return true;
    }
    return false;
}

/* z pitch (in doubles) the bespoke pipeline wants: N/2+1 complex rounded up to a multiple of the tile width FFT_C (4). */
int shq_fft3d_pitch(int N) { return 2 * (((N / 2 + 1) + FFT_C - 1) / FFT_C * FFT_C); }

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

/* forward + potential_transfer + inverse of a full cube through the transposing pipeline: d_scratch is a second mesh of the same size */
int shq_fft3d_run_transposed(shq_context *ctx, double *d_mesh, double *d_scratch, int N, int zp, bool from_i64, double inv_scale,
                             const double *d_sinctab, double asmth2, double pot_factor)
{
    SHQ_CHECK(shq_fft3d_supported(N) && N % (2 * FFT_C) == 0, SHQ_ERR_INVALID, "fft3d: unsupported mesh size %d", N);
    SHQ_CHECK(zp == shq_fft3d_pitch(N) && d_mesh && d_scratch && d_mesh != d_scratch, SHQ_ERR_INVALID, "fft3d: bad pitch or scratch mesh");
    SHQ_TRY(ensure_twiddles(ctx, N));
    GreenArgs ga;
    ga.sinctab = d_sinctab;
    ga.asmth2 = asmth2;
    ga.pot_factor = pot_factor;
    ga.y0 = 0;
    ga.alt = nullptr;
    ga.nyl = 1;
    ga.qstride = ga.alt_outer = 0;
    if(ctx->fft_gax_n != N || ctx->fft_gax_asmth2 != asmth2 || ctx->fft_gax_src != d_sinctab) {
        SHQ_TRY(ctx->fft_gax.reserve((size_t) N));
        fft_gax_kernel<<<dim3((unsigned) ((N + 255) / 256)), dim3(256), 0, ctx->stream>>>(N, d_sinctab, asmth2, ctx->fft_gax.ptr);
        SHQ_HIP(hipGetLastError());
        ctx->fft_gax_n = N;
        ctx->fft_gax_asmth2 = asmth2;
        ctx->fft_gax_src = d_sinctab;
    }
    ga.gaxg = ctx->fft_gax.ptr;
#define SHQ_FFT_CASE(NN) case NN: return run_t<NN>(ctx, d_mesh, d_scratch, zp, from_i64, inv_scale, ga)
    switch(N) {
        SHQ_FFT_CASE(16); SHQ_FFT_CASE(24); SHQ_FFT_CASE(32); SHQ_FFT_CASE(40); SHQ_FFT_CASE(48); SHQ_FFT_CASE(64);
        SHQ_FFT_CASE(80); SHQ_FFT_CASE(96); SHQ_FFT_CASE(128); SHQ_FFT_CASE(192); SHQ_FFT_CASE(256); SHQ_FFT_CASE(384);
        SHQ_FFT_CASE(512); SHQ_FFT_CASE(768); SHQ_FFT_CASE(960); SHQ_FFT_CASE(1024); SHQ_FFT_CASE(1152); SHQ_FFT_CASE(1200);
        SHQ_FFT_CASE(1536);
    }
#undef SHQ_FFT_CASE
    return SHQ_ERR_INVALID;
}

/* stage: 0 forward only (r2c), 1 inverse only (c2r), 2 forward + potential_transfer + inverse.
 * d_mesh: [N][N][zp] doubles in place; from_i64: the mesh holds the int64 fixed-point deposit. */
int shq_fft3d_run(shq_context *ctx, double *d_mesh, int N, int zp, int stage, bool from_i64, double inv_scale,
                  const double *d_sinctab, double asmth2, double pot_factor)
{
    return shq_fft3d_run_slab(ctx, d_mesh, N, zp, stage, from_i64, inv_scale, d_sinctab, asmth2, pot_factor, N, 0);
}

/* nslab: x-planes (stages 10, 11) or y-rows (stage 12, starting at mesh row y0) of this rank's slab */
int shq_fft3d_run_slab(shq_context *ctx, double *d_mesh, int N, int zp, int stage, bool from_i64, double inv_scale,
                       const double *d_sinctab, double asmth2, double pot_factor, int nslab, int y0)
{
    return shq_fft3d_run_slab_packed(ctx, d_mesh, N, zp, stage, from_i64, inv_scale, d_sinctab, asmth2, pot_factor, nslab, y0, nullptr, 1);
}

/* stages 13 / 14: as 10 / 11, the spectrum's transposed side in d_packed = [nranks][nslab][N / nranks][zp / 2] complex */
int shq_fft3d_run_slab_packed(shq_context *ctx, double *d_mesh, int N, int zp, int stage, bool from_i64, double inv_scale,
                              const double *d_sinctab, double asmth2, double pot_factor, int nslab, int y0, double *d_packed, int nranks)
{
    SHQ_CHECK(shq_fft3d_supported(N), SHQ_ERR_INVALID, "fft3d: unsupported mesh size %d", N);
    SHQ_CHECK(nslab > 0 && nslab <= N && y0 >= 0 && y0 + (stage == 12 ? nslab : 0) <= N, SHQ_ERR_INVALID, "fft3d: bad slab geometry");
    SHQ_CHECK(zp >= N + 2 && zp % 8 == 0, SHQ_ERR_INVALID, "fft3d: pitch %d must be a multiple of 8 doubles and >= N+2", zp);
    SHQ_TRY(ensure_twiddles(ctx, N));
    GreenArgs ga;
    ga.sinctab = d_sinctab;
    ga.gaxg = nullptr;
    ga.asmth2 = asmth2;
    ga.pot_factor = pot_factor;
    ga.y0 = y0;
    ga.alt = reinterpret_cast<double2 *>(d_packed);
    ga.nyl = 1;
    ga.qstride = ga.alt_outer = 0;
    if(stage == 13 || stage == 14) {
        SHQ_CHECK(d_packed && nranks >= 1 && N % nranks == 0, SHQ_ERR_INVALID, "fft3d: packed stages need a buffer and a rank count that divides the mesh");
        ga.nyl = N / nranks;
        ga.alt_outer = (long long) ga.nyl * (zp / 2);
        ga.qstride = (long long) nslab * ga.alt_outer;
    }
#define SHQ_FFT_CASE(NN) case NN: return run_n<NN>(ctx, d_mesh, zp, stage, from_i64, inv_scale, ga, nslab)
    switch(N) {
        SHQ_FFT_CASE(16); SHQ_FFT_CASE(24); SHQ_FFT_CASE(32); SHQ_FFT_CASE(40); SHQ_FFT_CASE(48); SHQ_FFT_CASE(64);
        SHQ_FFT_CASE(80); SHQ_FFT_CASE(96); SHQ_FFT_CASE(128); SHQ_FFT_CASE(192); SHQ_FFT_CASE(256); SHQ_FFT_CASE(384);
        SHQ_FFT_CASE(512); SHQ_FFT_CASE(768); SHQ_FFT_CASE(960); SHQ_FFT_CASE(1024); SHQ_FFT_CASE(1152); SHQ_FFT_CASE(1200);
        SHQ_FFT_CASE(1536);
    }
#undef SHQ_FFT_CASE
    return SHQ_ERR_INVALID;
}
