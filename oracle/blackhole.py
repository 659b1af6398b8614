"""CPU restatement (numpy, brute-force neighbour search) of the two tree walks of the reference's black-hole module.

TEST INFRASTRUCTURE ONLY.  Follows /root/reference/libgadget/blackhole.cpp:
  blackhole_accretion_ngbiter / _reduce / _copy        :471-692
  blackhole_accretion_postprocess                       :373-468   (blackhole_soundspeed :147-157)
  check_grav_bound                                       :160-180
  blackhole_feedback_ngbiter / _copy / _reduce          :728-927   (add_injected_BH_energy :700-710, get_random_dir :712-723)
  blackhole_feedback_haswork / _postprocess             :878-965
with the neighbour test of treewalk_visit_ngbiter (treewalk.c:925-975: gas + black holes of the tree, not garbage, symmetric:
r2 <= max(Hsml_i, Hsml_j)^2) as a loop over all particles, and the kernel of densitykernel.h.  The black holes are visited in queue
order and their neighbours in particle order (the reference's threads and tree give another order: sums differ in the last bits, and
the compare-and-swap marks of two holes with consecutive IDs can fall the other way).
parity unpinned: the reference's tests hold no fixture for this module; the checks are invariants (mass and momentum of swallowed
particles, energy injected) and this restatement."""
import numpy as np

GAMMA = 5.0 / 3.0
GAMMA_MINUS1 = GAMMA - 1
TIMEBINS = 46


def nearest(x, box):
    return np.where(x > 0.5 * box, x - box, np.where(x < -0.5 * box, x + box, x))


def kernel_wk(u, H, kt):
    """density_kernel_wk with density_kernel_init(H, type): densitykernel.h (cubic 1, quintic 2, quartic 4)"""
    support = {1: 4.0, 2: 6.0, 4: 5.0}[kt]
    sigma = {1: 1 / np.pi, 2: 1 / (120 * np.pi), 4: 1 / (20 * np.pi)}[kt]
    s = support / 2. / H
    wknorm = sigma * (s * s * s)
    q = u * support / 2
    if kt == 1:
        w = 0.25 * (2 - q) ** 3 - (1 - q) ** 3 if q < 1 else (0.25 * (2 - q) ** 3 if q < 2 else 0.0)
    elif kt == 4:
        if q < 0.5:
            w = (2.5 - q) ** 4 - 5 * (1.5 - q) ** 4 + 10 * (0.5 - q) ** 4
        elif q < 1.5:
            w = (2.5 - q) ** 4 - 5 * (1.5 - q) ** 4
        else:
            w = (2.5 - q) ** 4 if q < 2.5 else 0.0
    else:
        if q < 1:
            w = (3 - q) ** 5 - 6 * (2 - q) ** 5 + 15 * (1 - q) ** 5
        elif q < 2:
            w = (3 - q) ** 5 - 6 * (2 - q) ** 5
        else:
            w = (3 - q) ** 5 if q < 3 else 0.0
    return wknorm * w


def is_timebin_active(b, cur):
    if b <= 0 or cur <= 0:
        return True
    return cur % (1 << int(b)) == 0


def sph_velpred(P, S, i, kf):
    """SPH_VelPred, density2.h:89-98"""
    bg, bh = int(P["TimeBinGravity"][i]), int(P["TimeBinHydro"][i])
    return P["Vel"][i] + kf.gravkicks[bg] * P["FullTreeGravAccel"][i] + P["GravPM"][i] * kf.FgravkickB + kf.hydrokicks[bh] * S["HydroAccel"][P["PI"][i]]


def dm_velpred(P, i, kf):
    bg = int(P["TimeBinGravity"][i])
    return P["Vel"][i] + kf.gravkicks[bg] * P["FullTreeGravAccel"][i] + P["GravPM"][i] * kf.FgravkickB


def neighbours(P, i, box):
    """indices in particle order that treewalk_visit_ngbiter hands to the ngbiter of black hole i, with r2 and r"""
    n = len(P)
    live = ((P["Flags"] & 1) == 0) & np.isin(P["Type"], (0, 5)) & ~((P["Type"] == 5) & ((P["Flags"] & 2) != 0))
    d = nearest(P["Pos"][i][None, :] - P["Pos"], box)
    r2 = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]
    dist = np.maximum(P["Hsml"], P["Hsml"][i])
    ok = live & (r2 <= dist * dist)
    idx = np.flatnonzero(ok)
    return idx, r2[idx], d[idx]


def accretion(P, S, B, ids, queue, kf, prm, Ti_Current, rnd, work):
    """treewalk_run(tw_accretion): fills work (dict of arrays by slot) and the slots; prm is the shq_bh_params-like namespace"""
    work["SPH_SwallowID"][:] = 0
    work["BH_SwallowID"][:] = 0
    kt = prm.DensityKernelType
    for i in queue:
        pi = int(P["PI"][i])
        H = float(P["Hsml"][i])
        myid = int(ids[i])
        Ivel = P["Vel"][i].copy()
        Iacc = P["FullTreeGravAccel"][i] + P["GravPM"][i] + B["DFAccel"][pi]
        Imass, Ibh, Idens, Imtrack = float(P["Mass"][i]), float(B["Mass"][pi]), float(B["Density"][pi]), float(B["Mtrack"][pi])
        enc, fws, sment, gv, mgas = 0, 0.0, 0.0, np.zeros(3), 0.0
        idx, r2s, ds = neighbours(P, i, prm.BoxSize)
        for other, r2, dx in zip(idx, r2s, ds):
            other = int(other)
            r = np.sqrt(r2)
            if P["Mass"][other] < 0:
                continue
            t = int(P["Type"][other])
            if prm.WindsDecoupleSph and t == 0 and S["DelayTime"][P["PI"][other]] > 0:
                continue
            if int(ids[other]) == myid:
                continue
            if t == 5 and r < (2 * prm.ForceSoftening / 2.8):
                enc = 1
                flag = 0
                if prm.RepositionEnabled == 1:
                    flag = 1
                if prm.MergeGravBound == 0:
                    flag = 1
                opi = int(P["PI"][other])
                if prm.MergeGravBound == 1 and prm.RepositionEnabled == 0:
                    vp = dm_velpred(P, other, kf)
                    KE = PE = 0.0
                    for d in range(3):
                        dv = Ivel[d] - vp[d]
                        da = Iacc[d] - P["FullTreeGravAccel"][other][d] - P["GravPM"][other][d] - B["DFAccel"][opi][d]
                        KE += 0.5 * (dv * dv)
                        PE += da * dx[d]
                    KE /= (prm.atime * prm.atime)
                    PE /= prm.atime
                    flag = 1 if PE + KE <= 0 else 0
                if flag == 1:
                    readid = int(work["BH_SwallowID"][opi])
                    if readid != 0 and readid < myid:
                        work["BH_SwallowID"][opi] = myid + 1
                    elif readid == 0 and (int(ids[other]) < myid or not is_timebin_active(P["TimeBinHydro"][other], Ti_Current)):
                        work["BH_SwallowID"][opi] = myid + 1
            if t == 0 and r2 < H * H:
                u = r * (1.0 / H)
                wk = kernel_wk(u, H, kt)
                mj = float(P["Mass"][other])
                spi = int(P["PI"][other])
                sment += (mj * wk * S["Entropy"][spi])
                vp = sph_velpred(P, S, other, kf)
                gv += mj * wk * vp
                p = 0.0
                part = Imass
                if prm.SeedBHDynMass > 0 and Imtrack < prm.SeedBHDynMass:
                    part = Imtrack
                if (Ibh - part) > 0 and Idens > 0:
                    p = (Ibh - part) * wk / Idens
                w = rnd[int(ids[other]) % len(rnd)]
                if w < p and int(work["SPH_SwallowID"][spi]) < myid + 1:
                    work["SPH_SwallowID"][spi] = myid + 1
                fws += (mj * wk)
                if prm.BlackHoleKineticOn == 1:
                    mgas += mj
        B["encounter"][pi] = enc
        work["BH_FeedbackWeightSum"][pi] = fws
        work["BH_Entropy"][pi] = sment
        work["BH_SurroundingGasVel"][pi] = gv
        work["MgasEnc"][pi] = mgas
    for i in queue:
        accretion_postprocess(P, B, int(i), kf, prm, work)


def accretion_postprocess(P, B, i, kf, prm, work):
    pi = int(P["PI"][i])
    mdot = 0.0
    medd = prm.EddingtonConst * B["Mass"][pi] * prm.UnitTime_in_s / prm.HubbleParam
    if B["Density"][pi] > 0:
        work["BH_Entropy"][pi] /= B["Density"][pi]
        work["BH_SurroundingGasVel"][pi] /= B["Density"][pi]
        bhvel = 0.0
        for k in range(3):
            bhvel += (P["Vel"][i][k] - work["BH_SurroundingGasVel"][pi][k]) ** 2
        bhvel = np.sqrt(bhvel) / prm.atime
        rho = float(B["Density"][pi])
        rho_proper = rho * prm.a3inv
        cs = 0.0
        if rho > 0:
            cs = np.sqrt(GAMMA * work["BH_Entropy"][pi] * rho ** GAMMA_MINUS1) * prm.atime ** (-1.5 * GAMMA_MINUS1)
        norm = (cs * cs + bhvel * bhvel) ** 1.5
        if norm > 0:
            mdot = 4. * np.pi * prm.BlackHoleAccretionFactor * prm.GravInternal * prm.GravInternal * B["Mass"][pi] * B["Mass"][pi] * rho_proper / norm
    if prm.BlackHoleEddingtonFactor > 0.0 and mdot > prm.BlackHoleEddingtonFactor * medd:
        mdot = prm.BlackHoleEddingtonFactor * medd
    B["Mdot"][pi] = mdot
    dtime = kf.dloga_for_bin[int(P["TimeBinHydro"][i])] / prm.hubble
    B["Mass"][pi] += B["Mdot"][pi] * dtime
    if prm.BH_DRAG > 0:
        fac = 0.0
        if prm.BH_DRAG == 1:
            fac = B["Mdot"][pi] / float(P["Mass"][i])
        if prm.BH_DRAG == 2:
            fac = prm.BlackHoleEddingtonFactor * medd / B["Mass"][pi]
        fac *= prm.atime
        B["DragAccel"][pi] = -(P["Vel"][i] - work["BH_SurroundingGasVel"][pi]) * fac
    else:
        B["DragAccel"][pi] = 0
    if prm.BlackHoleKineticOn == 1:
        work["KEflag"][pi] = 0
        edd = B["Mdot"][pi] / medd
        lam = prm.BHKE_EddingtonThrFactor
        x = prm.BHKE_EddingtonMFactor * (B["Mass"][pi] / prm.BHKE_EddingtonMPivot) ** prm.BHKE_EddingtonMIndex
        if lam > x:
            lam = x
        if edd < lam:
            work["KEflag"][pi] = 1
            rho_crit_baryon = prm.OmegaBaryon * 3 * prm.Hubble ** 2 / (8 * np.pi * prm.GravInternal)
            rho_sfr = prm.BHKE_SfrCritOverDensity * rho_crit_baryon
            eps = (B["Density"][pi] / rho_sfr) / prm.BHKE_EffRhoFactor
            if eps > prm.BHKE_EffCap:
                eps = prm.BHKE_EffCap
            B["KineticFdbkEnergy"][pi] += eps * (B["Mdot"][pi] * dtime * prm.LightOverUnitVel ** 2)
        thr = 0.5 * B["VDisp"][pi] * B["VDisp"][pi] * work["MgasEnc"][pi]
        thr *= prm.BHKE_InjEnergyThr
        if B["VDisp"][pi] > 0 and B["KineticFdbkEnergy"][pi] > thr:
            work["KEflag"][pi] = 2


def feedback(P, S, B, ids, queue, kf, prm, maxpart, rnd, eeqos, work):
    """treewalk_run(tw_feedback) with haswork, then postprocess.  Returns (gas swallowed, holes swallowed)."""
    kt = prm.DensityKernelType
    todo = [int(i) for i in queue if work["BH_SwallowID"][P["PI"][i]] == 0]
    nsph = nbh = 0
    # SPH_VelPred of every gas particle before any kick lands (the reference reads Part.Vel while other threads kick: a race there)
    velpred_gas = {int(j): sph_velpred(P, S, int(j), kf) for j in np.flatnonzero((P["Type"] == 0) & ((P["Flags"] & 1) == 0))}
    for i in todo:
        pi = int(P["PI"][i])
        H = float(P["Hsml"][i])
        myid = int(ids[i])
        Idens, Imtrack = float(B["Density"][pi]), float(B["Mtrack"][pi])
        fws = float(work["BH_FeedbackWeightSum"][pi])
        dtime = kf.dloga_for_bin[int(P["TimeBinHydro"][i])] / prm.hubble
        fbe = prm.BlackHoleFeedbackFactor * 0.1 * B["Mdot"][pi] * dtime * prm.LightOverUnitVel ** 2
        channel, kefb = 0, 0.0
        if prm.BlackHoleKineticOn == 1 and work["KEflag"][pi] > 0:
            channel = 1
            if work["KEflag"][pi] == 2:
                kefb = float(B["KineticFdbkEnergy"][pi])
        accm, accbh, mom, cprog, mintb = 0.0, 0.0, np.zeros(3), 0, TIMEBINS
        idx, r2s, _ = neighbours(P, i, prm.BoxSize)
        for other, r2 in zip(idx, r2s):
            other = int(other)
            t = int(P["Type"][other])
            if int(ids[other]) == myid:
                continue
            if prm.WindsDecoupleSph and t == 0 and S["DelayTime"][P["PI"][other]] > 0:
                continue
            opi = int(P["PI"][other])
            if t == 5 and work["BH_SwallowID"][opi] != 0:
                if int(work["BH_SwallowID"][opi]) != myid + 1:
                    continue
                B["SwallowID"][opi] = int(work["BH_SwallowID"][opi]) - 1
                B["SwallowTime"][opi] = prm.atime
                P["Flags"][other] |= 2
                B["encounter"][opi] = 0
                cprog += int(B["CountProgs"][opi])
                accbh += float(B["Mass"][opi])
                om = float(P["Mass"][other])
                if prm.SeedBHDynMass > 0 and Imtrack > 0:
                    if B["Mtrack"][opi] < prm.SeedBHDynMass:
                        om = float(B["Mtrack"][opi])
                accm += om
                mom += om * dm_velpred(P, other, kf)
                nbh += 1
            if t == 0 and work["SPH_SwallowID"][opi] == 0 and r2 < H * H:
                if mintb > P["TimeBinHydro"][other]:
                    mintb = int(P["TimeBinHydro"][other])
                u = np.sqrt(r2) * (1.0 / H)
                mj = float(P["Mass"][other])
                wk = kernel_wk(u, H, kt)
                if fws > 0 and fbe > 0 and channel == 0 and mj > 0:
                    inj = fbe * mj * wk / fws
                    if eeqos is not None and eeqos[other]:
                        P["Flags"][other] |= 8
                    enttou = (S["Density"][opi] * prm.a3inv) ** GAMMA_MINUS1 / GAMMA_MINUS1
                    unew = S["Entropy"][opi] * enttou
                    unew += inj / mj
                    if unew > prm.MaxThermalU:
                        unew = prm.MaxThermalU
                    S["Entropy"][opi] = unew / enttou
                if kefb > 0 and channel == 1 and Idens > 0:
                    dvel = np.sqrt(2 * kefb * wk / Idens)
                    theta = np.arccos(2 * rnd[(int(ids[other]) + 3) % len(rnd)] - 1)
                    phi = 2 * np.pi * rnd[(int(ids[other]) + 4) % len(rnd)]
                    direc = np.array([np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi), np.cos(theta)])
                    P["Vel"][other] += dvel * direc
            if t == 0 and int(work["SPH_SwallowID"][opi]) == myid + 1:
                mj = float(P["Mass"][other])
                accm += mj
                mom += mj * velpred_gas[other]
                P["Flags"][other] |= 1                           # slots_mark_garbage
                S["ReverseLink"][opi] = maxpart + 100
                nsph += 1
        work["BH_accreted_Mass"][pi] = accm
        work["BH_accreted_BHMass"][pi] = accbh
        work["BH_accreted_momentum"][pi] = mom
        B["minTimeBin"][pi] = mintb
        B["CountProgs"][pi] += cprog
    for i in todo:
        pi = int(P["PI"][i])
        if work["BH_accreted_BHMass"][pi] > 0:
            B["Mass"][pi] += work["BH_accreted_BHMass"][pi]
        if work["BH_accreted_Mass"][pi] > 0:
            accmass = float(work["BH_accreted_Mass"][pi])
            pm = np.float32(P["Mass"][i])
            for k in range(3):
                P["Vel"][i][k] = (P["Vel"][i][k] * float(pm) + work["BH_accreted_momentum"][pi][k]) / (float(pm) + accmass)
            sd = prm.SeedBHDynMass
            if sd > 0 and B["Mtrack"][pi] + accmass < sd:
                B["Mtrack"][pi] += accmass
            elif B["Mtrack"][pi] < sd:
                P["Mass"][i] = np.float32(B["Mtrack"][pi] + accmass)
                B["Mtrack"][pi] = sd
            else:
                P["Mass"][i] = np.float32(float(pm) + accmass)
        if prm.BlackHoleKineticOn == 1 and work["KEflag"][pi] == 2:
            B["KineticFdbkEnergy"][pi] = 0
    return nsph, nbh
