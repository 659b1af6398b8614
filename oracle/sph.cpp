/* oracle/sph.cpp — CPU restatement of the reference SPH kernels, density and hydro loops.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 */
#include "oracle.h"
#include "sph_kernels.h"

/* libgadget/densitykernel.hpp:28-178; golden values tests/test_densitykernel.cpp:13-36 */
extern "C" int orc_density_kernel(int type, double H, double u, double eta, double out[5])
{
    if(type != 1 && type != 2 && type != 4)
        return 1;
    OrcKernel k(type, H);
    out[0] = OrcKernel::desnumngb(type, eta);
    out[1] = k.volume();
    out[2] = k.wk(u);
    out[3] = k.dwk(u);
    out[4] = k.dW(u);
    return 0;
}
