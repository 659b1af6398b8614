// Probe: how fast are the rocFFT routes for a 768^3 f64 real transform?  (diagnostic, not product code)
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { auto e_ = (x); if(e_ != 0) { printf("fail %s -> %d line %d\n", #x, (int) e_, __LINE__); return 1; } } while(0)
template <typename F> static double time_it(hipStream_t s, int reps, const F &f)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipStreamSynchronize(s);
    hipEventRecord(a, s);
    for(int i = 0; i < reps; i++) f();
    hipEventRecord(b, s); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 768;
    const int Nc = N / 2 + 1;
    const size_t padded = (size_t) N * N * (N + 2);
    double *a, *b;
    CK(hipMalloc(&a, padded * 8)); CK(hipMalloc(&b, padded * 8));
    CK(hipMemset(a, 0, padded * 8)); CK(hipMemset(b, 0, padded * 8));
    hipStream_t s; hipStreamCreate(&s);
    hipfftHandle r2c, c2r;
    CK(hipfftPlan3d(&r2c, N, N, N, HIPFFT_D2Z)); CK(hipfftPlan3d(&c2r, N, N, N, HIPFFT_Z2D));
    hipfftSetStream(r2c, s); hipfftSetStream(c2r, s);
    printf("N=%d\n", N);
    printf("3D in-place   r2c %.3f ms  c2r %.3f ms\n",
           time_it(s, 5, [&] { hipfftExecD2Z(r2c, a, (hipfftDoubleComplex *) a); }),
           time_it(s, 5, [&] { hipfftExecZ2D(c2r, (hipfftDoubleComplex *) a, a); }));
    printf("3D out-of-place r2c %.3f ms  c2r %.3f ms\n",
           time_it(s, 5, [&] { hipfftExecD2Z(r2c, a, (hipfftDoubleComplex *) b); }),
           time_it(s, 5, [&] { hipfftExecZ2D(c2r, (hipfftDoubleComplex *) b, a); }));
    // 2-D batched (y,z) per x-plane, then strided 1-D along x
    hipfftHandle p2, p2i, p1;
    int n2[2] = {N, N};
    int inembed[2] = {N, N + 2}, onembed[2] = {N, Nc};
    CK(hipfftPlanMany(&p2, 2, n2, inembed, 1, N * (N + 2), onembed, 1, N * Nc, HIPFFT_D2Z, N));
    CK(hipfftPlanMany(&p2i, 2, n2, onembed, 1, N * Nc, inembed, 1, N * (N + 2), HIPFFT_Z2D, N));
    int n1[1] = {N};
    int emb[1] = {N};
    CK(hipfftPlanMany(&p1, 1, n1, emb, N * Nc, 1, emb, N * Nc, 1, HIPFFT_Z2Z, N * Nc));
    hipfftSetStream(p2, s); hipfftSetStream(p2i, s); hipfftSetStream(p1, s);
    printf("2D batched in-place r2c %.3f ms c2r %.3f ms ; strided 1D z2z %.3f ms\n",
           time_it(s, 5, [&] { hipfftExecD2Z(p2, a, (hipfftDoubleComplex *) a); }),
           time_it(s, 5, [&] { hipfftExecZ2D(p2i, (hipfftDoubleComplex *) a, a); }),
           time_it(s, 5, [&] { hipfftExecZ2Z(p1, (hipfftDoubleComplex *) a, (hipfftDoubleComplex *) a, HIPFFT_FORWARD); }));
    // 1-D batched contiguous along z only (one pass reference)
    hipfftHandle pz;
    int nz[1] = {N};
    int ie[1] = {N + 2}, oe[1] = {Nc};
    CK(hipfftPlanMany(&pz, 1, nz, ie, 1, N + 2, oe, 1, Nc, HIPFFT_D2Z, N * N));
    hipfftSetStream(pz, s);
    printf("1D z r2c batched %.3f ms\n", time_it(s, 5, [&] { hipfftExecD2Z(pz, a, (hipfftDoubleComplex *) a); }));
    return 0;
}
