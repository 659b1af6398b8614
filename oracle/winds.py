"""CPU restatement (numpy, brute-force neighbours) of the reference's wind treewalks for new stars.

TEST INFRASTRUCTURE ONLY.  Follows /root/reference/libgadget/winds.cpp:
  winds_find_weights / sfr_wind_weight_ngbiter / sfr_wind_reduce_weight      :227-269, 390-447
  winds_and_feedback, the StarKick queue and its resolution                   :171-207, 295-369
  sfr_wind_feedback_ngbiter, get_wind_params, wind_do_kick, get_wind_dir      :449-565
with the asymmetric neighbour test of treewalk_visit_ngbiter (treewalk.c:925-975: gas of the tree, not garbage, r2 <= Hsml^2).
parity unpinned: the reference's tests hold no fixture for the wind module.  The outcome does not depend on the order the stars or
their neighbours are visited in (the nearest star kicks, ties to the smaller star ID), so device and restatement must agree exactly."""
import numpy as np

GAMMA_MINUS1 = 5.0 / 3.0 - 1
WIND_SUBGRID, WIND_DECOUPLE_SPH, WIND_USE_HALO, WIND_FIXED_EFFICIENCY = 1, 2, 4, 8


def nearest(x, box):
    return np.where(x > 0.5 * box, x - box, np.where(x < -0.5 * box, x + box, x))


def gas_neighbours(P, S, i, box):
    live = ((P["Flags"] & 1) == 0) & (P["Type"] == 0)
    d = nearest(P["Pos"][i][None, :] - P["Pos"], box)
    r2 = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]
    h = P["Hsml"][i]
    idx = np.flatnonzero(live & (r2 <= h * h))
    return idx, np.sqrt(r2[idx])


def get_wind_params(vdisp, time, prm):
    vphys = vdisp / time
    utherm = prm.WindThermalFactor * 1.5 * vphys * vphys
    if prm.WindModel & WIND_FIXED_EFFICIENCY:
        windeff = prm.WindEfficiency
        vel = prm.WindSpeed * time
    elif prm.WindModel & WIND_USE_HALO:
        windeff = prm.WindSigma0 ** 2 / (vphys * vphys + 2 * utherm)
        vel = prm.WindSpeedFactor * vdisp
    else:
        raise ValueError("WindModel is strange")
    if vel < prm.MinWindVelocity * time:
        vel = prm.MinWindVelocity * time
    return vel, windeff, utherm


def candidates(P, S, ST, ids, newstars, prm, rnd):
    """the two walks: (TotalWeight by star slot, candidate kicks sorted with StarKick's comparison); nothing is applied"""
    totalweight = np.zeros(len(ST))
    if prm.WindModel & WIND_SUBGRID:
        return totalweight, []
    for i in newstars:
        assert P["Type"][i] == 4
        idx, r = gas_neighbours(P, S, i, prm.BoxSize)
        tw = 0.0
        for other, rr in zip(idx, r):
            if rr > P["Hsml"][i]:
                continue
            if S["DelayTime"][P["PI"][other]] > 0:
                continue
            tw += 1.0 * float(P["Mass"][other])
        totalweight[P["PI"][i]] = tw
    kicks = []
    for i in newstars:
        idx, r = gas_neighbours(P, S, i, prm.BoxSize)
        tw = totalweight[P["PI"][i]]
        vdisp = float(ST["VDisp"][P["PI"][i]])
        for other, rr in zip(idx, r):
            if rr > P["Hsml"][i]:
                continue
            if S["DelayTime"][P["PI"][other]] > 0:
                continue
            if tw == 0 or vdisp <= 0:
                continue
            if P["Flags"][other] & 3:
                continue
            v, windeff, utherm = get_wind_params(vdisp, prm.Time, prm)
            p = windeff * float(P["Mass"][i]) / tw
            if rnd[(int(ids[i]) + int(ids[other])) % len(rnd)] < p and v > 0:
                kicks.append((int(other), float(rr), int(ids[i]), v, utherm))
    kicks.sort(key=lambda k: (k[0], k[1], k[2]))
    return totalweight, kicks


def apply(P, S, ids, kicks, prm, rnd):
    """the resolution after the walk (winds.cpp:330-350) and wind_do_kick: kicks in any order; returns the number applied"""
    kicks = sorted(kicks, key=lambda k: (k[0], k[1], k[2]))
    last, applied = -1, 0
    decouple = (prm.WindModel & WIND_DECOUPLE_SPH) and prm.MaxWindFreeTravelTime > 0
    for other, _, _, vel, therm in kicks:
        if other == last:
            continue
        last = other
        applied += 1
        theta = np.arccos(2 * rnd[(int(ids[other]) + 3) % len(rnd)] - 1)
        phi = 2 * np.pi * rnd[(int(ids[other]) + 4) % len(rnd)]
        direc = np.array([np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi), np.cos(theta)])
        if vel > 0 and prm.Time > 0:
            P["Vel"][other] += vel * direc
            pi = P["PI"][other]
            enttou = (S["Density"][pi] / prm.Time ** 3) ** GAMMA_MINUS1 / GAMMA_MINUS1
            S["Entropy"][pi] += therm / enttou
            if decouple:
                delay = prm.WindFreeTravelLength / (vel / prm.Time)
                if delay > prm.MaxWindFreeTravelTime:
                    delay = prm.MaxWindFreeTravelTime
                S["DelayTime"][pi] = delay
    return applied


def winds_and_feedback(P, S, ST, ids, newstars, prm, rnd):
    """returns (TotalWeight by star slot, sorted kick list, number applied); P["Vel"], S["Entropy"], S["DelayTime"] are modified"""
    totalweight, kicks = candidates(P, S, ST, ids, newstars, prm, rnd)
    return totalweight, kicks, apply(P, S, ids, kicks, prm, rnd)


def winds_evolve(P, S, lst, a3inv, hubble, dens_thresh, max_travel, kf):
    """winds_evolve, winds.cpp:370-387, for the gas particles of lst"""
    for i in lst:
        if P["Type"][i] != 0 or (P["Flags"][i] & 1):
            continue
        pi = P["PI"][i]
        if S["DelayTime"][pi] > 0 and S["Density"][pi] * a3inv < dens_thresh:
            S["DelayTime"][pi] = 0
        if S["DelayTime"][pi] > 0:
            if S["DelayTime"][pi] > max_travel:
                S["DelayTime"][pi] = max_travel
            dtime = kf.dloga_for_bin[int(P["TimeBinHydro"][i])] / hubble
            S["DelayTime"][pi] = max(S["DelayTime"][pi] - dtime, 0)


def winds_subgrid(P, S, ids, lst, stellarmass, prm, rnd):
    """winds_subgrid + winds_make_after_sf, winds.cpp:272-292, 567-585; stellarmass is indexed like lst.  Returns the number kicked."""
    if not (prm.WindModel & WIND_SUBGRID):
        return 0
    n = 0
    decouple = (prm.WindModel & WIND_DECOUPLE_SPH) and prm.MaxWindFreeTravelTime > 0
    for k, i in enumerate(lst):
        pi = P["PI"][i]
        vel, windeff, utherm = get_wind_params(float(S["VDisp"][pi]), prm.Time, prm)
        pw = windeff * stellarmass[k] / float(P["Mass"][i])
        prob = 1 - np.exp(-pw)
        if not (rnd[(int(ids[i]) + 2) % len(rnd)] < prob):
            continue
        if vel > 0 and prm.Time > 0:
            theta = np.arccos(2 * rnd[(int(ids[i]) + 3) % len(rnd)] - 1)
            phi = 2 * np.pi * rnd[(int(ids[i]) + 4) % len(rnd)]
            direc = np.array([np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi), np.cos(theta)])
            P["Vel"][i] += vel * direc
            enttou = (S["Density"][pi] / prm.Time ** 3) ** GAMMA_MINUS1 / GAMMA_MINUS1
            S["Entropy"][pi] += utherm / enttou
            if decouple:
                delay = prm.WindFreeTravelLength / (vel / prm.Time)
                if delay > prm.MaxWindFreeTravelTime:
                    delay = prm.MaxWindFreeTravelTime
                S["DelayTime"][pi] = delay
            n += 1
    return n
