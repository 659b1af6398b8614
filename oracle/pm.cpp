/* oracle/pm.cpp — CPU restatement of the reference particle-mesh long-range force.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Follows libgadget/petapm.cpp (CIC :1132-1183, deposit :1304-1310, mode enumeration
 * :1258-1298, petapm_mesh_to_k :159-162, FFT convention notes :1335-1348) and
 * libgadget/gravpm.cpp (sinc :294-302, potential_transfer :378-444, diff_kernel/force_transfer
 * :448-488, readout :489-500).  The reference's FFT is heffte/fftw (absent here); this file
 * carries its own mixed-radix FFT, cross-checked against numpy.fft in tests/test_oracle_pm.py.
 */
#include "oracle.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include <complex>
#include <vector>
#include <omp.h>

typedef std::complex<double> cplx;

namespace {

struct Fft1d {
    int n;
    std::vector<cplx> tw; /* tw[k] = exp(-2 pi i k / n) */
    explicit Fft1d(int n_) : n(n_), tw(n_)
    {
        for(int k = 0; k < n; k++)
            tw[k] = cplx(cos(2 * M_PI * k / n), -sin(2 * M_PI * k / n));
    }
    static int smallest_factor(int m)
    {
        for(int p = 2; p * p <= m; p++)
            if(m % p == 0)
                return p;
        return m;
    }
    /* out[0..m) = DFT_m of in[0], in[s], in[2s], ...; sign -1 forward, +1 backward (unscaled). */
    void rec(int m, const cplx *in, int s, cplx *out, int sign) const
    {
        if(m == 1) {
            out[0] = in[0];
            return;
        }
        const int p = smallest_factor(m);
        const int q = m / p;
        for(int r = 0; r < p; r++)
            rec(q, in + (size_t) r * s, s * p, out + (size_t) r * q, sign);
        const int step = n / m; /* w_m^j = tw[j*step] */
        cplx tmp[64];
        if(p > 64)
            abort();
        for(int k = 0; k < q; k++) {
            for(int r = 0; r < p; r++) {
                int idx = (int) (((int64_t) r * k * step) % n);
                cplx w = tw[idx];
                if(sign > 0)
                    w = std::conj(w);
                tmp[r] = out[k + (size_t) r * q] * w;
            }
            for(int j = 0; j < p; j++) {
                cplx acc = tmp[0];
                for(int r = 1; r < p; r++) {
                    int idx = (int) (((int64_t) r * j * q * step) % n);
                    cplx w = tw[idx];
                    if(sign > 0)
                        w = std::conj(w);
                    acc += tmp[r] * w;
                }
                out[k + (size_t) j * q] = acc;
            }
        }
    }
};

inline int mesh_to_k(int N, int i) { return i <= N / 2 ? i : (i - N); } /* petapm.cpp:159-162 */

/* gravpm.cpp:294-302 */
inline double sinc_unnormed(double x)
{
    if(x < 1e-5 && x > -1e-5) {
        double x2 = x * x;
        return 1.0 - x2 / 6. + x2 * x2 / 120.;
    }
    return sin(x) / x;
}
/* gravpm.cpp:448-456 */
inline double diff_kernel(double w) { return 1 / 6.0 * (8 * sin(w) - sin(2 * w)); }

} // namespace

extern "C" void orc_fft_r2c(int N, const double *real, double *complx_)
{
    cplx *out = reinterpret_cast<cplx *>(complx_);
    const int Nc = N / 2 + 1;
    Fft1d f(N);
#pragma omp parallel
    {
        std::vector<cplx> a(N), b(N);
#pragma omp for collapse(2)
        for(int x = 0; x < N; x++)
            for(int y = 0; y < N; y++) {
                const double *row = real + ((size_t) x * N + y) * N;
                for(int z = 0; z < N; z++)
                    a[z] = cplx(row[z], 0);
                f.rec(N, a.data(), 1, b.data(), -1);
                cplx *o = out + ((size_t) x * N + y) * Nc;
                for(int z = 0; z < Nc; z++)
                    o[z] = b[z];
            }
#pragma omp for collapse(2)
        for(int x = 0; x < N; x++)
            for(int z = 0; z < Nc; z++) {
                cplx *base = out + (size_t) x * N * Nc + z;
                for(int y = 0; y < N; y++)
                    a[y] = base[(size_t) y * Nc];
                f.rec(N, a.data(), 1, b.data(), -1);
                for(int y = 0; y < N; y++)
                    base[(size_t) y * Nc] = b[y];
            }
#pragma omp for collapse(2)
        for(int y = 0; y < N; y++)
            for(int z = 0; z < Nc; z++) {
                cplx *base = out + (size_t) y * Nc + z;
                for(int x = 0; x < N; x++)
                    a[x] = base[(size_t) x * N * Nc];
                f.rec(N, a.data(), 1, b.data(), -1);
                for(int x = 0; x < N; x++)
                    base[(size_t) x * N * Nc] = b[x];
            }
    }
}

extern "C" void orc_fft_c2r(int N, const double *complx_, double *real)
{
    const cplx *in = reinterpret_cast<const cplx *>(complx_);
    const int Nc = N / 2 + 1;
    Fft1d f(N);
    std::vector<cplx> work((size_t) N * N * Nc);
    memcpy(work.data(), in, sizeof(cplx) * work.size());
#pragma omp parallel
    {
        std::vector<cplx> a(N), b(N);
#pragma omp for collapse(2)
        for(int y = 0; y < N; y++)
            for(int z = 0; z < Nc; z++) {
                cplx *base = work.data() + (size_t) y * Nc + z;
                for(int x = 0; x < N; x++)
                    a[x] = base[(size_t) x * N * Nc];
                f.rec(N, a.data(), 1, b.data(), +1);
                for(int x = 0; x < N; x++)
                    base[(size_t) x * N * Nc] = b[x];
            }
#pragma omp for collapse(2)
        for(int x = 0; x < N; x++)
            for(int z = 0; z < Nc; z++) {
                cplx *base = work.data() + (size_t) x * N * Nc + z;
                for(int y = 0; y < N; y++)
                    a[y] = base[(size_t) y * Nc];
                f.rec(N, a.data(), 1, b.data(), +1);
                for(int y = 0; y < N; y++)
                    base[(size_t) y * Nc] = b[y];
            }
#pragma omp for collapse(2)
        for(int x = 0; x < N; x++)
            for(int y = 0; y < N; y++) {
                const cplx *row = work.data() + ((size_t) x * N + y) * Nc;
                /* Hermitian completion; imaginary parts of the self-conjugate entries are
                 * ignored as a c2r transform does. */
                a[0] = cplx(row[0].real(), 0);
                for(int z = 1; z < Nc; z++) {
                    a[z] = row[z];
                    a[N - z] = std::conj(row[z]);
                }
                if(N % 2 == 0)
                    a[N / 2] = cplx(row[N / 2].real(), 0);
                f.rec(N, a.data(), 1, b.data(), +1);
                double *o = real + ((size_t) x * N + y) * N;
                for(int z = 0; z < N; z++)
                    o[z] = b[z].real();
            }
    }
}

/* CIC index/weights for one particle: petapm.cpp:1147-1183 with a single global periodic
 * region (offset 0, size Nmesh, wrap at Nmesh). */
static inline void cic_setup(const double *Pos, double CellSize, int N, int iCell[3], double Res[3])
{
    for(int k = 0; k < 3; k++) {
        double tmp = Pos[k] / CellSize;
        iCell[k] = (int) floor(tmp);
        Res[k] = tmp - iCell[k];
        /* periodic wrap of the base cell (regions in the reference are padded and exchanged
         * periodically: petapm.cpp:837-983) */
        iCell[k] = ((iCell[k] % N) + N) % N;
    }
}

/* The transforms orc_pm_force calls: the oracle's own (below) unless a caller has installed others.  bench.py's cpu_baseline installs
 * scipy's multi-threaded pocketfft behind the same deposit / transfer-function / readout code, so that the reported CPU baseline is
 * not priced on this file's plain mixed-radix FFT (the reference runs FFTW / heffte, absent here).  Same conventions: unscaled both
 * ways, half spectrum [x][y][z'] of N / 2 + 1 complex. */
typedef void (*orc_fft_fn)(int, const double *, double *);
static orc_fft_fn g_r2c = nullptr, g_c2r = nullptr;
extern "C" void orc_set_fft(orc_fft_fn r2c, orc_fft_fn c2r)
{
    g_r2c = r2c;
    g_c2r = c2r;
}
static void pm_r2c(int N, const double *real, double *complx) { (g_r2c ? g_r2c : orc_fft_r2c)(N, real, complx); }
static void pm_c2r(int N, const double *complx, double *real) { (g_c2r ? g_c2r : orc_fft_c2r)(N, complx, real); }

extern "C" void orc_pm_force(const double *pos, const float *mass, const uint8_t *skip, int64_t n,
                             const shq_pm_params *pm, int fixed_point_log2scale, int use_stencil,
                             double *gravpm, double *potential, double *mesh_rho, double *mesh_pot)
{
    const int N = pm->Nmesh;
    const int Nc = N / 2 + 1;
    const size_t N3 = (size_t) N * N * N;
    const double CellSize = pm->BoxSize / N;
    std::vector<double> real(N3, 0.0);
    const double scale = fixed_point_log2scale >= 0 ? ldexp(1.0, fixed_point_log2scale) : 0;
    /* fixed-point mode sums in 64-bit integers (exact, order independent) like the device */
    std::vector<long long> ireal(fixed_point_log2scale >= 0 ? N3 : 0, 0);

    /* deposit: put_particle_to_mesh, petapm.cpp:1304-1310 */
#pragma omp parallel for
    for(int64_t i = 0; i < n; i++) {
        if(skip && skip[i])
            continue;
        int iCell[3];
        double Res[3];
        cic_setup(&pos[3 * i], CellSize, N, iCell, Res);
        for(int connection = 0; connection < 8; connection++) {
            double weight = 1.0;
            size_t linear = 0;
            for(int k = 0; k < 3; k++) {
                int offset = (connection >> k) & 1;
                int tmp = (iCell[k] + offset) % N;
                linear = linear * N + tmp;
                weight *= offset ? Res[k] : (1 - Res[k]);
            }
            double v = weight * mass[i];
            if(fixed_point_log2scale >= 0) {
                const long long q = llrint(v * scale);
#pragma omp atomic update
                ireal[linear] += q;
            } else {
#pragma omp atomic update
                real[linear] += v;
            }
        }
    }
    if(fixed_point_log2scale >= 0)
        for(size_t c = 0; c < N3; c++)
            real[c] = (double) ireal[c] * (1.0 / scale);
    if(mesh_rho)
        memcpy(mesh_rho, real.data(), sizeof(double) * N3);

    std::vector<cplx> rho_k((size_t) N * N * Nc);
    pm_r2c(N, real.data(), reinterpret_cast<double *>(rho_k.data()));

    /* potential_transfer, gravpm.cpp:378-444 (no neutrinos, no P(k) side effect) */
    const double asmth2 = pow((2 * M_PI) * pm->Asmth / N, 2);
    const double pot_factor = -pm->G / (M_PI * pm->BoxSize);
#pragma omp parallel for collapse(2)
    for(int x = 0; x < N; x++)
        for(int y = 0; y < N; y++)
            for(int z = 0; z < Nc; z++) {
                int kpos[3] = {mesh_to_k(N, x), mesh_to_k(N, y), mesh_to_k(N, z)};
                int64_t k2 = (int64_t) kpos[0] * kpos[0] + (int64_t) kpos[1] * kpos[1] + (int64_t) kpos[2] * kpos[2];
                cplx &v = rho_k[((size_t) x * N + y) * Nc + z];
                if(k2 == 0) {
                    v = 0;
                    continue;
                }
                double f = 1.0;
                const double smth = exp(-k2 * asmth2) / k2;
                for(int k = 0; k < 3; k++) {
                    double tmp = (kpos[k] * M_PI) / N;
                    tmp = sinc_unnormed(tmp);
                    f *= 1. / (tmp * tmp);
                }
                const double fac = pot_factor * smth * f * f;
                v *= fac;
            }

    std::vector<cplx> work((size_t) N * N * Nc);
    std::vector<double> fmesh[4];
    const int nout = use_stencil ? 1 : 4;
    for(int out = 0; out < nout; out++) {
        fmesh[out].resize(N3);
        /* pm_apply_transfer_function, petapm.cpp:1258-1298; out 0 = potential (plain copy) */
#pragma omp parallel for collapse(2)
        for(int x = 0; x < N; x++)
            for(int y = 0; y < N; y++)
                for(int z = 0; z < Nc; z++) {
                    size_t ip = ((size_t) x * N + y) * Nc + z;
                    cplx v = rho_k[ip];
                    if(out > 0) {
                        int kpos[3] = {mesh_to_k(N, x), mesh_to_k(N, y), mesh_to_k(N, z)};
                        /* force_transfer, gravpm.cpp:464-478 */
                        double fac = -1 * diff_kernel(kpos[out - 1] * (2 * M_PI / N)) * (N / pm->BoxSize);
                        v = cplx(-v.imag() * fac, v.real() * fac);
                    }
                    work[ip] = v;
                }
        pm_c2r(N, reinterpret_cast<double *>(work.data()), fmesh[out].data());
    }
    if(mesh_pot)
        memcpy(mesh_pot, fmesh[0].data(), sizeof(double) * N3);
    if(use_stencil) {
        /* real-space image of i*K(w): (2/3)(f(+1)-f(-1)) - (1/12)(f(+2)-f(-2)), times -N/L */
        const double fac = -(N / pm->BoxSize);
        for(int d = 1; d <= 3; d++)
            fmesh[d].resize(N3);
#pragma omp parallel for collapse(2)
        for(int x = 0; x < N; x++)
            for(int y = 0; y < N; y++)
                for(int z = 0; z < N; z++) {
                    const double *P = fmesh[0].data();
                    auto at = [&](int a, int b, int c) {
                        return P[((size_t) ((a + N) % N) * N + ((b + N) % N)) * N + ((c + N) % N)];
                    };
                    size_t ip = ((size_t) x * N + y) * N + z;
                    fmesh[1][ip] = fac * ((2. / 3) * (at(x + 1, y, z) - at(x - 1, y, z)) - (1. / 12) * (at(x + 2, y, z) - at(x - 2, y, z)));
                    fmesh[2][ip] = fac * ((2. / 3) * (at(x, y + 1, z) - at(x, y - 1, z)) - (1. / 12) * (at(x, y + 2, z) - at(x, y - 2, z)));
                    fmesh[3][ip] = fac * ((2. / 3) * (at(x, y, z + 1) - at(x, y, z - 1)) - (1. / 12) * (at(x, y, z + 2) - at(x, y, z - 2)));
                }
    }

    /* readout: gravpm.cpp:489-500 via pm_iterate (petapm.cpp:1132-1197); GravPM zeroed first
     * (gravpm.cpp:88-92); Potential is added to. */
#pragma omp parallel for
    for(int64_t i = 0; i < n; i++) {
        double g[4] = {0, 0, 0, 0};
        if(!(skip && skip[i])) {
            int iCell[3];
            double Res[3];
            cic_setup(&pos[3 * i], CellSize, N, iCell, Res);
            for(int connection = 0; connection < 8; connection++) {
                double weight = 1.0;
                size_t linear = 0;
                for(int k = 0; k < 3; k++) {
                    int offset = (connection >> k) & 1;
                    int tmp = (iCell[k] + offset) % N;
                    linear = linear * N + tmp;
                    weight *= offset ? Res[k] : (1 - Res[k]);
                }
                for(int out = 0; out < 4; out++)
                    g[out] += weight * fmesh[out][linear];
            }
        }
        if(potential)
            potential[i] += g[0];
        gravpm[3 * i + 0] = g[1];
        gravpm[3 * i + 1] = g[2];
        gravpm[3 * i + 2] = g[3];
    }
}
