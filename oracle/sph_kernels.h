/* oracle/sph_kernels.h — Price (2012) SPH kernels as the reference defines them
 * (libgadget/densitykernel.hpp:28-178). TEST INFRASTRUCTURE ONLY. */
#ifndef ORC_SPH_KERNELS_H
#define ORC_SPH_KERNELS_H
#include <math.h>

struct OrcKernel {
    int type;      /* 1 cubic, 2 quintic, 4 quartic (enum DensityKernelType) */
    int support;   /* 2H/h: 4, 6, 5 */
    double H;
    double Wknorm;
    static int support_of(int type) { return type == 1 ? 4 : (type == 2 ? 6 : 5); }
    static double sigma_of(int type)
    {
        /* cbsigma[2], quinsigma[2], quarsigma[2]: densitykernel.hpp:86,151,120 */
        return type == 1 ? 1 / M_PI : (type == 2 ? 1 / (120 * M_PI) : 1 / (20 * M_PI));
    }
    OrcKernel(int t, double H_) : type(t), support(support_of(t)), H(H_)
    {
        Wknorm = sigma_of(t) * pow(support / 2. / H, 3); /* densitykernel.hpp:33 */
    }
    /* densitykernel.hpp:36-41 */
    static double desnumngb(int type, double eta) { return (4.0 / 3 * M_PI) * pow(support_of(type) / 2. * eta, 3); }
    double volume() const { return (4.0 / 3 * M_PI) * pow(H, 3); } /* :43-46 */
    double wk_int(double q) const
    {
        switch(type) {
        case 1: /* :92-100 */
            if(q < 1.0) return 0.25 * pow(2 - q, 3) - pow(1 - q, 3);
            if(q < 2.0) return 0.25 * pow(2 - q, 3);
            return 0.0;
        case 4: /* :126-137 */
            if(q < 0.5) return pow(2.5 - q, 4) - 5 * pow(1.5 - q, 4) + 10 * pow(0.5 - q, 4);
            if(q < 1.5) return pow(2.5 - q, 4) - 5 * pow(1.5 - q, 4);
            if(q < 2.5) return pow(2.5 - q, 4);
            return 0.0;
        default: /* quintic :157-168 */
            if(q < 1.0) return pow(3 - q, 5) - 6 * pow(2 - q, 5) + 15 * pow(1 - q, 5);
            if(q < 2.0) return pow(3 - q, 5) - 6 * pow(2 - q, 5);
            if(q < 3.0) return pow(3 - q, 5);
            return 0.0;
        }
    }
    double dwk_int(double q) const
    {
        switch(type) {
        case 1: /* :102-110 */
            if(q < 1.0) return -0.25 * 3 * pow(2 - q, 2) + 3 * pow(1 - q, 2);
            if(q < 2.0) return -0.25 * 3 * pow(2 - q, 2);
            return 0.0;
        case 4: /* :139-150 */
            if(q < 0.5) return -4 * pow(2.5 - q, 3) + 20 * pow(1.5 - q, 3) - 40 * pow(0.5 - q, 3);
            if(q < 1.5) return -4 * pow(2.5 - q, 3) + 20 * pow(1.5 - q, 3);
            if(q < 2.5) return -4 * pow(2.5 - q, 3);
            return 0.0;
        default: /* :170-182 */
            if(q < 1.0) return -5 * pow(3 - q, 4) + 30 * pow(2 - q, 4) - 75 * pow(1 - q, 4);
            if(q < 2.0) return -5 * pow(3 - q, 4) + 30 * pow(2 - q, 4);
            if(q < 3.0) return -5 * pow(3 - q, 4);
            return 0.0;
        }
    }
    double dwk(double u) const { return Wknorm * support / 2. / H * dwk_int(u * support / 2.); } /* :48-51 */
    double wk(double u) const { return Wknorm * wk_int(u * support / 2.); }                      /* :53-56 */
    double dW(double u) const { return -(3 * wk(u) / H + u * dwk(u)); }                          /* :58-61 */
};
#endif
