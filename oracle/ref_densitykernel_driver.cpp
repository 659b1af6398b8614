/* Own driver (not reference code) exposing the reference's header-only SPH kernels
 * (libgadget/densitykernel.hpp, compiled in place from /root/reference) through a C ABI so
 * tests can pin oracle/sph.cpp against the real thing.  Output goes to oracle/_ref/ only. */
#include <densitykernel.hpp>

template <class K> static void eval(double H, double u, double eta, double out[5])
{
    K kern(H);
    out[0] = K::desnumngb(eta);
    out[1] = kern.volume();
    out[2] = kern.wk(u);
    out[3] = kern.dwk(u);
    out[4] = kern.dW(u);
}

/* type: 1 cubic, 2 quintic, 4 quartic (enum DensityKernelType, libgadget/density2.h) */
extern "C" int ref_density_kernel(int type, double H, double u, double eta, double out[5])
{
    switch(type) {
    case 1: eval<CubicDensityKernel>(H, u, eta, out); return 0;
    case 2: eval<QuinticDensityKernel>(H, u, eta, out); return 0;
    case 4: eval<QuarticDensityKernel>(H, u, eta, out); return 0;
    }
    return 1;
}
