"""Particle set-ups and parameter sets of the black-hole walk tests (no reference fixture exists for this module)."""
from types import SimpleNamespace

import numpy as np

import shenqi_amd as sq
from shenqi_amd import capi
import common as cm

PARAMS = dict(BoxSize=cm.BOX, ForceSoftening=0.35, SeedBHDynMass=0.0, atime=0.25, a3inv=64.0, hubble=0.3, GravInternal=43.0, BlackHoleAccretionFactor=100.0,
              BlackHoleEddingtonFactor=2.1, BlackHoleFeedbackFactor=0.05, EddingtonConst=14.3, UnitTime_in_s=3.0, HubbleParam=0.7, LightOverUnitVel=3.0e2,
              MaxThermalU=1.5e4, OmegaBaryon=0.05, Hubble=0.1, BHKE_EddingtonThrFactor=1.5, BHKE_EddingtonMFactor=0.002, BHKE_EddingtonMPivot=0.05,
              BHKE_EddingtonMIndex=2.0, BHKE_EffRhoFactor=0.05, BHKE_EffCap=0.05, BHKE_InjEnergyThr=1.0, BHKE_SfrCritOverDensity=57.7, DensityKernelType=1,
              WindsDecoupleSph=0, RepositionEnabled=1, MergeGravBound=1, BH_DRAG=0, BlackHoleKineticOn=0)


def params(**kw):
    d = dict(PARAMS)
    d.update(kw)
    p = capi.BhParams()
    for k, v in d.items():
        setattr(p, k, v)
    return p, SimpleNamespace(**d)


def setup(seed, ngrid=14, nbh=36, ndm=300, bh_hsml=1.0):
    """gas on a jittered grid, some dark matter the walks must ignore, black holes: single ones, close pairs (mergers) and a triple"""
    rng = np.random.default_rng(seed)
    ngas = ngrid**3
    sp = cm.BOX / ngrid
    gas = np.mod(cm.grid_positions(ngrid) + rng.normal(size=(ngas, 3)) * 0.25 * sp, cm.BOX)
    dm = rng.random((ndm, 3)) * cm.BOX
    bh = rng.random((nbh, 3)) * cm.BOX
    for k in range(0, 16, 2):                              # pairs within the merging radius 2 * 0.35 / 2.8 = 0.25
        bh[k + 1] = bh[k] + rng.normal(size=3) * 0.05
    bh[18] = bh[16] + [0.08, 0, 0]                         # a chain: 16 - 18 - 17
    bh[17] = bh[18] + [0.08, 0, 0]
    pos = np.mod(np.concatenate([gas, dm, bh]), cm.BOX)
    n = len(pos)
    perm = rng.permutation(n)
    types = np.concatenate([np.zeros(ngas, np.uint8), np.ones(ndm, np.uint8), np.full(nbh, 5, np.uint8)])[perm]
    pos = pos[perm]
    pman = sq.PartManager(n, cm.BOX)
    P = pman.Base
    P["Pos"], P["Type"] = pos, types
    P["Mass"] = rng.uniform(0.8, 1.2, n).astype(np.float32)
    P["ID"] = (rng.permutation(n).astype(np.uint64) * 5 + 11)                   # no two consecutive IDs (see oracle/blackhole.py)
    P["Vel"] = rng.normal(size=(n, 3)) * 30
    P["FullTreeGravAccel"] = rng.normal(size=(n, 3)) * 50
    P["GravPM"] = rng.normal(size=(n, 3)) * 5
    P["TimeBinGravity"] = rng.integers(20, 24, n)
    P["TimeBinHydro"] = rng.integers(16, 21, n)
    inv = np.empty(n, dtype=np.int64)
    inv[perm] = np.arange(n)
    for k in (0, 4, 8, 12):                                # these pairs are bound: nearly the same velocity, attracting accelerations
        i, j = inv[ngas + ndm + k], inv[ngas + ndm + k + 1]
        P["Vel"][j] = P["Vel"][i] + 0.01
        P["FullTreeGravAccel"][j] = P["FullTreeGravAccel"][i] + 5000.0 * cm.nearest(P["Pos"][i] - P["Pos"][j], cm.BOX) if hasattr(cm, "nearest") else \
            P["FullTreeGravAccel"][i] + 5000.0 * ((P["Pos"][i] - P["Pos"][j] + cm.BOX / 2) % cm.BOX - cm.BOX / 2)
    isgas, isbh = types == 0, types == 5
    P["Hsml"] = sp * rng.uniform(1.2, 1.8, n)
    P["Hsml"][isbh] = sp * rng.uniform(1.8, 2.6, nbh) * bh_hsml
    P["PI"][isgas] = rng.permutation(ngas)
    P["PI"][isbh] = np.arange(nbh)[::-1]
    gi = np.flatnonzero(isgas)
    P["Flags"][gi[:5]] |= 1                                # a few garbage gas particles
    S = np.zeros(ngas, dtype=capi.SPH_DTYPE)
    S["ReverseLink"][P["PI"][gi]] = gi
    S["Entropy"] = rng.uniform(50, 150, ngas)
    S["Density"] = rng.uniform(0.5, 2.0, ngas) * ngas / cm.BOX**3
    S["HydroAccel"] = rng.normal(size=(ngas, 3)) * 20
    S["DelayTime"] = np.where(rng.random(ngas) < 0.15, 0.5, 0.0)
    B = np.zeros(nbh, dtype=capi.BH_DTYPE)
    bi = np.flatnonzero(isbh)
    B["ReverseLink"][P["PI"][bi]] = bi
    B["Mass"] = rng.uniform(0.5, 8.0, nbh)                 # some below, some above the particle mass
    B["Density"] = rng.uniform(0.6, 1.5, nbh) * ngas / cm.BOX**3
    B["Density"][3] = 0                                    # a hole that has not found its density yet
    B["Mtrack"] = rng.uniform(0.2, 3.0, nbh)
    B["DFAccel"] = rng.normal(size=(nbh, 3)) * 3
    B["VDisp"] = rng.uniform(0, 40, nbh)
    B["KineticFdbkEnergy"] = rng.uniform(0, 4e4, nbh)
    B["CountProgs"] = rng.integers(1, 4, nbh)
    B["SwallowID"] = np.uint64(0xffffffffffffffff)
    B["encounter"] = 7
    B["minTimeBin"] = 3
    kf = sq.KickFactors()
    kf.FgravkickB = 2e-3
    for b in range(47):
        kf.gravkicks[b] = 1e-4 * (b - 10)
        kf.hydrokicks[b] = 5e-5 * (b - 8)
        kf.dloga_for_bin[b] = 1e-7 * 2.0 ** (b - 16) if b > 0 else 0.0
    rnd = rng.random(4099)
    return pman, S, B, kf, rnd, bi


def make_work(ngas, nbh):
    w = dict(SPH_SwallowID=np.full(ngas, 99, dtype=np.uint64), BH_SwallowID=np.full(nbh, 99, dtype=np.uint64), BH_FeedbackWeightSum=np.zeros(nbh),
             BH_Entropy=np.zeros(nbh), BH_SurroundingGasVel=np.zeros((nbh, 3)), MgasEnc=np.zeros(nbh), KEflag=np.zeros(nbh, dtype=np.int32),
             BH_accreted_Mass=np.zeros(nbh), BH_accreted_BHMass=np.zeros(nbh), BH_accreted_momentum=np.zeros((nbh, 3)))
    cw = capi.BhWork(*[w[k].ctypes.data for k in ("SPH_SwallowID", "BH_SwallowID", "BH_FeedbackWeightSum", "BH_Entropy", "BH_SurroundingGasVel", "MgasEnc", "KEflag",
                                                   "BH_accreted_Mass", "BH_accreted_BHMass", "BH_accreted_momentum")])
    return w, cw
