/* tools/copy_probe.hip — the yardstick for the PM's FFT passes: what this card moves when a kernel does nothing but read and write
 * the 768^3 mesh (3.66 GB, z pitch 388 complex) in the access shapes of the five passes, with 16-byte loads and stores.
 *
 *   hipcc --offload-arch=gfx950 -O3 tools/copy_probe.hip -o build/copy_probe && build/copy_probe [N]
 *
 * Modes (all "TB/s" = (bytes read + bytes written) / time):
 *   copy      src -> dst, contiguous, persistent grid-stride, 16 B per lane                        (MI355X_MICROARCH.md: 6.29 TB/s)
 *   copy_nt   the same with nontemporal loads and stores
 *   inplace   x -> x, contiguous (the Z passes' shape: rows along z)
 *   tileY     in place, tiles of C complex columns x N rows, row stride zpc (the Y passes' shape), through registers, workgroups
 *             persistent over an XCD-chunked tile order, next tile's loads issued before this tile's stores
 *   tileX     the same with row stride N * zpc (the X pass's shape)
 * for C = 4 (64-byte row segments, what fft3d.hip uses) and C = 8 (whole 128-byte lines).
 * Design probe: not part of the library, nothing links it. */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <string>
#include <functional>

#define CK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while(0)

typedef double2 c16;

template <bool NT> __device__ __forceinline__ c16 ld(const c16 *p)
{
    if(NT) {
        c16 v;
        v.x = __builtin_nontemporal_load(&p->x);
        v.y = __builtin_nontemporal_load(&p->y);
        return v;
    }
    return *p;
}
template <bool NT> __device__ __forceinline__ void st(c16 *p, c16 v)
{
    if(NT) {
        __builtin_nontemporal_store(v.x, &p->x);
        __builtin_nontemporal_store(v.y, &p->y);
    } else
        *p = v;
}

/* U loads in flight per lane, then U stores */
template <bool NT, int U> __global__ __launch_bounds__(256) void copy_kernel(const c16 *__restrict__ src, c16 *__restrict__ dst, long long n)
{
    const long long stride = (long long) gridDim.x * blockDim.x * U;
    for(long long i0 = (long long) blockIdx.x * blockDim.x * U + threadIdx.x; i0 < n; i0 += stride) {
        c16 v[U];
#pragma unroll
        for(int u = 0; u < U; u++)
            if(i0 + u * 256 < n)
                v[u] = ld<NT>(src + i0 + u * 256);
#pragma unroll
        for(int u = 0; u < U; u++)
            if(i0 + u * 256 < n)
                st<NT>(dst + i0 + u * 256, v[u]);
    }
}

__device__ __forceinline__ unsigned xcd_block(unsigned b, unsigned nb, unsigned K)
{
    if(K == 0)
        return b;
    const unsigned super = b / (8u * K);
    if((super + 1u) * 8u * K > nb)
        return b;
    const unsigned within = b - super * 8u * K;
    return super * 8u * K + (within & 7u) * K + (within >> 3);
}

/* tiles of C columns x N rows: element (row, col) at base + row * es + col; persistent, next tile prefetched into registers */
template <int N, int C, bool NT> __global__ __launch_bounds__(256) void tile_kernel(c16 *cm, c16 *dm, long long es, long long outer_stride, int ntiles, int ntot, unsigned xcdk)
{
    constexpr int E = C * N / 256;
    c16 cur[E], nxt[E];
    int t = (int) xcd_block(blockIdx.x, gridDim.x, xcdk);
    if(t >= ntot)
        return;
    auto base_of = [&](int tt) { const int outer = tt / ntiles, tile = tt - outer * ntiles; return cm + (long long) outer * outer_stride + (long long) tile * C; };
    c16 *base = base_of(t);
#pragma unroll
    for(int i = 0; i < E; i++) {
        const int e = threadIdx.x + i * 256;
        cur[i] = ld<NT>(base + (long long) (e / C) * es + (e % C));
    }
    while(true) {
        const int tn = t + (int) gridDim.x;
        const bool more = tn < ntot;
        c16 *base_n = base_of(more ? tn : t);
#pragma unroll
        for(int i = 0; i < E; i++) {
            const int e = threadIdx.x + i * 256;
            nxt[i] = ld<NT>(base_n + (long long) (e / C) * es + (e % C));
        }
#pragma unroll
        for(int i = 0; i < E; i++) {
            const int e = threadIdx.x + i * 256;
            c16 v = cur[i];
            v.x += 1.0; /* the stores must not be elided */
            st<NT>(dm + (base - cm) + (long long) (e / C) * es + (e % C), v);
        }
        if(!more)
            break;
#pragma unroll
        for(int i = 0; i < E; i++)
            cur[i] = nxt[i];
        t = tn;
        base = base_n;
    }
}

/* Transposing passes (the mesh changes layout between passes so that one side of every pass is a contiguous 48 KB tile):
 * tile t = 3072 elements.  SCATTER_WRITE: read the tile contiguously from src, write it to dst in pieces of PE elements, piece p of
 * tile t at p * pstride + (t % TA) * ta + (t / TA) * tb.  Otherwise the reverse: gather the pieces from src, write the tile contiguously. */
template <int PE, bool SCATTER_WRITE, bool NT> __global__ __launch_bounds__(256) void trans_kernel(const c16 *__restrict__ src, c16 *__restrict__ dst, int ntot,
                                                                                                  long long pstride, int TA, long long ta, long long tb, unsigned xcdk)
{
    constexpr int TILE = 3072, E = TILE / 256;
    c16 cur[E], nxt[E];
    int t = (int) xcd_block(blockIdx.x, gridDim.x, xcdk);
    if(t >= ntot)
        return;
    auto scat = [&](int tt, int e) { return (long long) (e / PE) * pstride + (long long) (tt % TA) * ta + (long long) (tt / TA) * tb + (e % PE); };
    auto cont = [&](int tt, int e) { return (long long) tt * TILE + e; };
#pragma unroll
    for(int i = 0; i < E; i++) {
        const int e = threadIdx.x + i * 256;
        cur[i] = ld<NT>(src + (SCATTER_WRITE ? cont(t, e) : scat(t, e)));
    }
    while(true) {
        const int tn = t + (int) gridDim.x;
        const bool more = tn < ntot;
        const int tf = more ? tn : t;
#pragma unroll
        for(int i = 0; i < E; i++) {
            const int e = threadIdx.x + i * 256;
            nxt[i] = ld<NT>(src + (SCATTER_WRITE ? cont(tf, e) : scat(tf, e)));
        }
#pragma unroll
        for(int i = 0; i < E; i++) {
            const int e = threadIdx.x + i * 256;
            c16 v = cur[i];
            v.x += 1.0;
            st<NT>(dst + (SCATTER_WRITE ? scat(t, e) : cont(t, e)), v);
        }
        if(!more)
            break;
#pragma unroll
        for(int i = 0; i < E; i++)
            cur[i] = nxt[i];
        t = tn;
    }
}

static double time_ms(hipStream_t s, int reps, const std::function<void()> &fn)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for(int i = 0; i < 2; i++)
        fn();
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s));
    for(int i = 0; i < reps; i++)
        fn();
    CK(hipEventRecord(b, s));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
    return ms / reps;
}

int main(int argc, char **argv)
{
    constexpr int N = 768;
    const int zpc = 388;
    const long long n = (long long) N * N * zpc; /* complex elements */
    const double gb = 16.0 * n / 1e9;
    c16 *x, *y;
    CK(hipMalloc(&x, 16 * n));
    CK(hipMalloc(&y, 16 * n));
    CK(hipMemset(x, 0, 16 * n));
    CK(hipMemset(y, 0, 16 * n));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    const int reps = 10;
    printf("mesh %d^3 x pitch %d complex = %.2f GB; TB/s = (read + written bytes) / time\n", N, zpc, gb);
    for(int grid : {256 * 4, 256 * 8, 256 * 16, 256 * 32}) {
        double ms = time_ms(s, reps, [&] { copy_kernel<false, 4><<<grid, 256, 0, s>>>(x, y, n); });
        printf("copy     U=4 grid %5d: %.3f ms  %.2f TB/s\n", grid, ms, 2 * gb / ms);
        ms = time_ms(s, reps, [&] { copy_kernel<true, 4><<<grid, 256, 0, s>>>(x, y, n); });
        printf("copy_nt  U=4 grid %5d: %.3f ms  %.2f TB/s\n", grid, ms, 2 * gb / ms);
        ms = time_ms(s, reps, [&] { copy_kernel<false, 8><<<grid, 256, 0, s>>>(x, y, n); });
        printf("copy     U=8 grid %5d: %.3f ms  %.2f TB/s\n", grid, ms, 2 * gb / ms);
        ms = time_ms(s, reps, [&] { copy_kernel<false, 4><<<grid, 256, 0, s>>>(x, x, n); });
        printf("inplace  U=4 grid %5d: %.3f ms  %.2f TB/s\n", grid, ms, 2 * gb / ms);
        ms = time_ms(s, reps, [&] { copy_kernel<true, 4><<<grid, 256, 0, s>>>(x, x, n); });
        printf("inpl_nt  U=4 grid %5d: %.3f ms  %.2f TB/s\n", grid, ms, 2 * gb / ms);
    }
    for(unsigned xk : {0u, 8u, 32u})
        for(int wg : {2, 4, 8}) {
            const int grid = 256 * wg;
            {
                const int ntiles = zpc / 4, ntot = N * ntiles;
                double ms = time_ms(s, reps, [&] { tile_kernel<N, 4, false><<<grid, 256, 0, s>>>(x, x, zpc, (long long) N * zpc, ntiles, ntot, xk); });
                printf("tileY C=4 xcdk %2u wg/cu %d: %.3f ms  %.2f TB/s\n", xk, wg, ms, 2 * gb / ms);
                ms = time_ms(s, reps, [&] { tile_kernel<N, 4, true><<<grid, 256, 0, s>>>(x, x, zpc, (long long) N * zpc, ntiles, ntot, xk); });
                printf("tileYntC=4 xcdk %2u wg/cu %d: %.3f ms  %.2f TB/s\n", xk, wg, ms, 2 * gb / ms);
                ms = time_ms(s, reps, [&] { tile_kernel<N, 4, false><<<grid, 256, 0, s>>>(x, x, (long long) N * zpc, zpc, ntiles, ntot, xk); });
                printf("tileX C=4 xcdk %2u wg/cu %d: %.3f ms  %.2f TB/s\n", xk, wg, ms, 2 * gb / ms);
            }
            if(wg <= 4) {
                const int ntiles = zpc / 8, ntot = N * ntiles; /* 388 / 8 = 48 tiles: the last 4 columns are left out (a probe) */
                const double frac = 8.0 * ntiles / zpc;
                double ms = time_ms(s, reps, [&] { tile_kernel<N, 8, false><<<grid, 256, 0, s>>>(x, x, zpc, (long long) N * zpc, ntiles, ntot, xk); });
                printf("tileY C=8 xcdk %2u wg/cu %d: %.3f ms  %.2f TB/s\n", xk, wg, ms, 2 * gb * frac / ms);
                ms = time_ms(s, reps, [&] { tile_kernel<N, 8, false><<<grid, 256, 0, s>>>(x, x, (long long) N * zpc, zpc, ntiles, ntot, xk); });
                printf("tileX C=8 xcdk %2u wg/cu %d: %.3f ms  %.2f TB/s\n", xk, wg, ms, 2 * gb * frac / ms);
            }
        }
    for(int wg : {2, 4}) { /* both sides strided as today, but out of place */
        const int grid = 256 * wg, ntiles = zpc / 4, ntot = N * ntiles;
        double ms = time_ms(s, reps, [&] { tile_kernel<N, 4, false><<<grid, 256, 0, s>>>(x, y, zpc, (long long) N * zpc, ntiles, ntot, 8); });
        printf("tileY C=4 out of place wg/cu %d: %.3f ms  %.2f TB/s\n", wg, ms, 2 * gb / ms);
        ms = time_ms(s, reps, [&] { tile_kernel<N, 4, false><<<grid, 256, 0, s>>>(x, y, (long long) N * zpc, zpc, ntiles, ntot, 8); });
        printf("tileX C=4 out of place wg/cu %d: %.3f ms  %.2f TB/s\n", wg, ms, 2 * gb / ms);
    }
    /* transposing passes, 96 of the 97 z' blocks (a probe): tiles of 3072 elements */
    {
        const int NB = 96;
        const double g2 = 16.0 * 3072.0 * N * NB / 1e9;
        const int ntot = N * NB;
        for(unsigned xk : {0u, 8u, 32u})
            for(int wg : {2, 4}) {
                const int grid = 256 * wg;
                /* Y forward, transposing write: tile t = (zb, x) contiguous, row ky -> [ky][zb][x][4]: neighbouring tiles fill neighbouring 64-byte pieces */
                double ms = time_ms(s, reps, [&] { trans_kernel<4, true, false><<<grid, 256, 0, s>>>(x, y, ntot, (long long) NB * N * 4, ntot, 4, 0, xk); });
                printf("Ytrans scatter-write 64B  xcdk %2u wg/cu %d: %.3f ms  %.2f TB/s\n", xk, wg, ms, 2 * g2 / ms);
                ms = time_ms(s, reps, [&] { trans_kernel<4, true, true><<<grid, 256, 0, s>>>(x, y, ntot, (long long) NB * N * 4, ntot, 4, 0, xk); });
                printf("Ytrans scatter-write 64B nt xcdk %2u wg/cu %d: %.3f ms  %.2f TB/s\n", xk, wg, ms, 2 * g2 / ms);
                ms = time_ms(s, reps, [&] { trans_kernel<4, false, false><<<grid, 256, 0, s>>>(x, y, ntot, (long long) NB * N * 4, ntot, 4, 0, xk); });
                printf("Ytrans gather-read 64B    xcdk %2u wg/cu %d: %.3f ms  %.2f TB/s\n", xk, wg, ms, 2 * g2 / ms);
                /* the same with tile order (x, zb): neighbouring tiles are NOT neighbours in the scattered layout */
                ms = time_ms(s, reps, [&] { trans_kernel<4, true, false><<<grid, 256, 0, s>>>(x, y, ntot, (long long) NB * N * 4, NB, (long long) N * 4, 4, xk); });
                printf("Ytrans scatter-write 64B far tiles xcdk %2u wg/cu %d: %.3f ms  %.2f TB/s\n", xk, wg, ms, 2 * g2 / ms);
                /* Z forward, transposing write: tile t = (x, yb) = 8 rows contiguous, z' block zb -> [x][zb][y][4]: pieces of 8 y x 64 B = 512 B */
                ms = time_ms(s, reps, [&] { trans_kernel<32, true, false><<<grid, 256, 0, s>>>(x, y, ntot, (long long) N * 4, N / 8, 32, (long long) NB * N * 4, xk); });
                printf("Ztrans scatter-write 512B xcdk %2u wg/cu %d: %.3f ms  %.2f TB/s\n", xk, wg, ms, 2 * g2 / ms);
                ms = time_ms(s, reps, [&] { trans_kernel<32, false, false><<<grid, 256, 0, s>>>(x, y, ntot, (long long) N * 4, N / 8, 32, (long long) NB * N * 4, xk); });
                printf("Ztrans gather-read 512B   xcdk %2u wg/cu %d: %.3f ms  %.2f TB/s\n", xk, wg, ms, 2 * g2 / ms);
                ms = time_ms(s, reps, [&] { trans_kernel<8, true, false><<<grid, 256, 0, s>>>(x, y, ntot, (long long) NB * N * 4, ntot / 2, 8, 0, xk); });
                printf("scatter-write 128B        xcdk %2u wg/cu %d: %.3f ms  %.2f TB/s\n", xk, wg, ms, 2 * g2 / ms);
                ms = time_ms(s, reps, [&] { trans_kernel<16, true, false><<<grid, 256, 0, s>>>(x, y, ntot, (long long) NB * N * 4, ntot / 4, 16, 0, xk); });
                printf("scatter-write 256B        xcdk %2u wg/cu %d: %.3f ms  %.2f TB/s\n", xk, wg, ms, 2 * g2 / ms);
                /* in place, contiguous tile both sides (Y inverse staying in its layout) */
                ms = time_ms(s, reps, [&] { trans_kernel<3072, true, false><<<grid, 256, 0, s>>>(x, x, ntot, 0, ntot, 3072, 0, xk); });
                printf("tile copy contiguous in place xcdk %2u wg/cu %d: %.3f ms  %.2f TB/s\n", xk, wg, ms, 2 * g2 / ms);
                /* both sides contiguous tiles, out of place (what a pass costs when neither side is scattered) */
                ms = time_ms(s, reps, [&] { trans_kernel<3072, true, false><<<grid, 256, 0, s>>>(x, y, ntot, 0, ntot, 3072, 0, xk); });
                printf("tile copy contiguous      xcdk %2u wg/cu %d: %.3f ms  %.2f TB/s\n", xk, wg, ms, 2 * g2 / ms);
            }
    }
    CK(hipFree(x));
    CK(hipFree(y));
    return 0;
}
