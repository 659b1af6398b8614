"""CPU restatement (numpy, brute-force neighbours) of the treewalk of the reference's metal return.

TEST INFRASTRUCTURE ONLY.  Follows /root/reference/libgadget/metal_return.cpp:
  metal_return_ngbiter      :596-667   (asymmetric, gas only, r2 > 0 and r2 < H^2, kernel weight or 1)
  metal_return_reduce       :573-577
The per-star yields (metal_return_copy, :540-571) are inputs.  The reference serialises the updates of a gas particle with a spin lock
in thread-arrival order; this restatement takes the stars in queue order and their neighbours in particle order.
parity unpinned: the reference's tests hold a fixture for the yield integrals (test_metal_return.cpp) but none for the walk; the walk
is checked against this restatement and its conservation laws."""
import numpy as np

from blackhole import kernel_wk, nearest

NMETALS = 9


def metal_return(P, S, queue, starvolume, massgen, metalgen, speciesgen, maxgasmass, sphweighting, kt, box):
    """modifies P["Mass"] (float32), S["Density"], S["Metallicity"], S["Metals"] (float32); returns MassReturn by queue position"""
    massreturn = np.zeros(len(queue))
    live = ((P["Flags"] & 1) == 0) & (P["Type"] == 0)
    for t, i in enumerate(queue):
        H = float(P["Hsml"][i])
        d = nearest(P["Pos"][i][None, :] - P["Pos"], box)
        r2 = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]
        idx = np.flatnonzero(live & (r2 <= H * H))
        for other in idx:
            rr2 = r2[other]
            if not (rr2 > 0 and rr2 < H * H):
                continue
            wk = 1.0
            if sphweighting:
                wk = kernel_wk(np.sqrt(rr2) * (1.0 / H), H, kt)
            pi = int(P["PI"][other])
            mass = np.float32(P["Mass"][other])
            volume = float(mass) / S["Density"][pi]
            rf = wk * volume / starvolume[t]
            thismass = rf * massgen[t]
            if float(mass) + thismass > maxgasmass:
                continue
            for m in range(NMETALS):
                tm = rf * speciesgen[t][m]
                S["Metals"][pi][m] = np.float32((float(np.float32(S["Metals"][pi][m]) * mass) + tm) / (float(mass) + thismass))
            thismetal = rf * metalgen[t]
            S["Metallicity"][pi] = (S["Metallicity"][pi] * float(mass) + thismetal) / (float(mass) + thismass)
            massfrac = (float(mass) + thismass) / float(mass)
            P["Mass"][other] = np.float32(float(mass) * massfrac)
            S["Density"][pi] *= massfrac
            massreturn[t] += thismass
    return massreturn
