"""ctypes bindings of the C-ABI (include/shenqi_hip.h) and of the host mirror library.

The compute lives in shenqi_amd/lib/libshenqi_hip.so (hand-written HIP for gfx950).  There is
no Python or CPU fallback: if the library is missing, importing this module raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIBDIR = os.environ.get("SHQ_LIBDIR") or os.path.join(_HERE, "lib")   # SHQ_LIBDIR: an alternative build of the two libraries (tuning experiments)
DATADIR = os.path.join(_HERE, "data")

NGRAVTAB = 512
NMAXCHILD = 8
NOFIELD = C.c_size_t(-1).value

WALK_EXACT = 0
WALK_GROUP = 1
WALK_AUTO = 2
WALK_TREE_ORDER = 0x100
WALK_DEFER_POSTPROCESS = 0x200


class ShqError(RuntimeError):
    """Non-zero status from the library (the shim's endrun() equivalent)."""


def _load(name):
    path = os.path.join(LIBDIR, name)
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). shenqi_amd has no fallback path."
        )
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


hip = _load("libshenqi_hip.so")
host = _load("libshenqi_host.so")


# ---- POD mirrors ---------------------------------------------------------------------------
class PartView(C.Structure):
    _fields_ = [
        ("base", C.c_void_p), ("elsize", C.c_size_t), ("numpart", C.c_int64),
        ("off_pos", C.c_size_t), ("off_mass", C.c_size_t), ("off_type", C.c_size_t),
        ("off_flags", C.c_size_t), ("off_pi", C.c_size_t), ("off_vel", C.c_size_t),
        ("off_treeacc", C.c_size_t), ("off_gravpm", C.c_size_t), ("off_potential", C.c_size_t),
        ("off_hsml", C.c_size_t), ("off_dthsml", C.c_size_t),
        ("off_timebin_hydro", C.c_size_t), ("off_timebin_gravity", C.c_size_t),
    ]


class Node(C.Structure):
    _fields_ = [
        ("sibling", C.c_int32), ("father", C.c_int32), ("len", C.c_double),
        ("center", C.c_double * 3), ("cofm", C.c_double * 3), ("mass", C.c_double),
        ("hmax", C.c_double), ("suns", C.c_int32 * NMAXCHILD), ("noccupied", C.c_int32),
        ("flags", C.c_uint32),
    ]


NODE_DTYPE = np.dtype(
    {
        "names": ["sibling", "father", "len", "center", "cofm", "mass", "hmax", "suns", "noccupied", "flags"],
        "formats": ["<i4", "<i4", "<f8", ("<f8", 3), ("<f8", 3), "<f8", "<f8", ("<i4", 8), "<i4", "<u4"],
        "offsets": [0, 4, 8, 16, 40, 64, 72, 80, 112, 116],
        "itemsize": 120,
    }
)
assert C.sizeof(Node) == 120

# struct particle_data, libgadget/partmanager.h:9-71 (160 B)
PARTICLE_DTYPE = np.dtype(
    {
        "names": ["Pos", "TopLeaf", "Mass", "PI", "Flags", "TimeBinHydro", "TimeBinGravity", "Type", "Vel",
                  "FullTreeGravAccel", "GravPM", "Ti_drift", "Hsml", "DtHsml", "ID", "GrNr", "Potential"],
        "formats": [("<f8", 3), "<i4", "<f4", "<i4", "u1", "u1", "u1", "u1", ("<f8", 3), ("<f8", 3), ("<f8", 3),
                    "<i8", "<f8", "<f8", "<u8", "<i8", "<f8"],
        "offsets": [0, 24, 28, 32, 36, 37, 38, 39, 40, 64, 88, 112, 120, 128, 136, 144, 152],
        "itemsize": 160,
    }
)

# struct sph_particle_data, libgadget/slotsmanager.h:97-131 (176 B)
SPH_DTYPE = np.dtype(
    {
        "names": ["ReverseLink", "Density", "EgyWtDensity", "Entropy", "DtEntropy", "MaxSignalVel", "HydroAccel",
                  "DhsmlEgyDensityFactor", "DivVel", "CurlVel", "Sfr", "Ne", "VDisp", "DelayTime", "Metallicity", "Metals"],
        "formats": ["<i4", "<f8", "<f8", "<f8", "<f8", "<f8", ("<f8", 3), "<f8", "<f8", "<f8", "<f8", "<f8", "<f8",
                    "<f8", "<f8", ("<f4", 9)],
        "offsets": [0, 8, 16, 24, 32, 40, 48, 72, 80, 88, 96, 104, 112, 120, 128, 136],
        "itemsize": 176,
    }
)


class TreeView(C.Structure):
    _fields_ = [
        ("nodes_base", C.c_void_p), ("firstnode", C.c_int64), ("lastnode", C.c_int64),
        ("numnodes", C.c_int64), ("rootnode", C.c_int32), ("full_particle_tree_flag", C.c_int32),
        ("BoxSize", C.c_double), ("father", C.c_void_p),
    ]


class GravParams(C.Structure):
    _fields_ = [
        ("BoxSize", C.c_double), ("cellsize", C.c_double), ("Rcut", C.c_double), ("G", C.c_double),
        ("cbrtrho0", C.c_double), ("ForceSoftening", C.c_double), ("ErrTolForceAcc", C.c_double),
        ("BHOpeningAngle2", C.c_double), ("TreeUseBH", C.c_int32), ("pad_", C.c_int32),
        ("shortrange_table", C.c_float * NGRAVTAB), ("shortrange_table_potential", C.c_float * NGRAVTAB),
        ("dx", C.c_double),
    ]


class TreeBuildStats(C.Structure):
    _fields_ = [("nparticles", C.c_int64), ("numnodes", C.c_int64), ("maxdepth", C.c_int32), ("build_ms", C.c_float)]


TIMEBINS = 46


class ActiveInfo(C.Structure):
    _fields_ = [("NumActiveParticle", C.c_int64), ("NumActiveGravity", C.c_int64), ("NumActiveHydro", C.c_int64),
                ("TimeBinCountType", C.c_int64 * (6 * (TIMEBINS + 1)))]


ACTIVE_RESIDENT = C.c_void_p(2**64 - 1)    # SHQ_ACTIVE_RESIDENT
SUBLIST_RESIDENT = C.c_void_p(2**64 - 2)   # SHQ_SUBLIST_RESIDENT


class WalkStats(C.Structure):
    _fields_ = [
        ("ntargets", C.c_int64), ("ninteractions", C.c_int64), ("min_interactions", C.c_int64),
        ("max_interactions", C.c_int64), ("nnodes_visited", C.c_int64), ("nwave_interactions", C.c_int64),
        ("nwave_node_interactions", C.c_int64), ("nnode_interactions", C.c_int64),
        ("kernel_ms", C.c_double),
    ]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}




class KickFactors(C.Structure):
    _fields_ = [("FgravkickB", C.c_double), ("gravkicks", C.c_double * (TIMEBINS + 1)),
                ("hydrokicks", C.c_double * (TIMEBINS + 1)), ("dloga_kick", C.c_double * (TIMEBINS + 1)),
                ("dloga_for_bin", C.c_double * (TIMEBINS + 1))]


class DensityParams(C.Structure):
    _fields_ = [("BoxSize", C.c_double), ("DesNumNgb", C.c_double), ("DesNumNgbBH", C.c_double),
                ("MinGasHsml", C.c_double), ("MaxNumNgbDeviation", C.c_double), ("update_hsml", C.c_int32),
                ("BlackHoleOn", C.c_int32), ("DoEgyDensity", C.c_int32), ("WindsDecouple", C.c_int32),
                ("DensityKernelType", C.c_int32), ("pad_", C.c_int32), ("kf", KickFactors)]


class HydroParams(C.Structure):
    _fields_ = [("BoxSize", C.c_double), ("atime", C.c_double), ("fac_mu", C.c_double), ("fac_vsic_fix", C.c_double),
                ("hubble_a2", C.c_double), ("drifts", C.c_double * (TIMEBINS + 1)), ("ArtBulkViscConst", C.c_double),
                ("DensityContrastLimit", C.c_double), ("WindSpeed", C.c_double), ("WindFreeTravelDensThresh", C.c_double),
                ("DensityIndependentSphOn", C.c_int32), ("DensityKernelType", C.c_int32), ("kf", KickFactors)]


class SphView(C.Structure):
    _fields_ = [("base", C.c_void_p), ("elsize", C.c_size_t), ("numslots", C.c_int64),
                ("off_density", C.c_size_t), ("off_egywtdensity", C.c_size_t), ("off_entropy", C.c_size_t),
                ("off_dtentropy", C.c_size_t), ("off_maxsignalvel", C.c_size_t), ("off_hydroaccel", C.c_size_t),
                ("off_dhsmlegydensityfactor", C.c_size_t), ("off_divvel", C.c_size_t), ("off_curlvel", C.c_size_t),
                ("off_delaytime", C.c_size_t)]


# struct bh_particle_data, libgadget/slotsmanager.h:35-73 (248 B)
BH_DTYPE = np.dtype(
    {
        "names": ["ReverseLink", "minTimeBin", "encounter", "TimeBinDynFric", "JumpToMinPot", "Mass", "Mdot", "Density", "DivVel", "Mtrack",
                  "KineticFdbkEnergy", "VDisp", "DFAccel", "DF_SurroundingVel", "DF_SurroundingRmsVel", "DF_SurroundingDensity", "DragAccel",
                  "SwallowTime", "SwallowID", "Mseed", "FormationTime", "MinPot", "MinPotPos", "MinPotVel", "CountProgs"],
        "formats": ["<i4", "u1", "i1", "u1", "i1", "<f8", "<f8", "<f8", "<f8", "<f8", "<f8", "<f8", ("<f8", 3), ("<f8", 3), "<f8", "<f8",
                    ("<f8", 3), "<f8", "<u8", "<f8", "<f8", "<f8", ("<f8", 3), ("<f8", 3), "<i4"],
        "offsets": [0, 4, 5, 6, 7, 8, 16, 24, 32, 40, 48, 56, 64, 88, 112, 120, 128, 152, 160, 168, 176, 184, 192, 216, 240],
        "itemsize": 248,
    }
)


class BhDynView(C.Structure):
    _fields_ = [("base", C.c_void_p), ("elsize", C.c_size_t), ("numslots", C.c_int64),
                ("off_mintimebin", C.c_size_t), ("off_timebindynfric", C.c_size_t), ("off_jumptominpot", C.c_size_t),
                ("off_dfaccel", C.c_size_t), ("off_df_surroundingvel", C.c_size_t), ("off_dragaccel", C.c_size_t),
                ("off_minpotpos", C.c_size_t), ("off_minpotvel", C.c_size_t)]


def bh_dyn_view(BhP):
    """shq_bh_dyn_view of a numpy array of BH_DTYPE records."""
    v = BhDynView()
    f = BhP.dtype.fields
    v.base, v.elsize, v.numslots = BhP.ctypes.data, BhP.dtype.itemsize, len(BhP)
    v.off_mintimebin, v.off_timebindynfric, v.off_jumptominpot = f["minTimeBin"][1], f["TimeBinDynFric"][1], f["JumpToMinPot"][1]
    v.off_dfaccel, v.off_df_surroundingvel, v.off_dragaccel = f["DFAccel"][1], f["DF_SurroundingVel"][1], f["DragAccel"][1]
    v.off_minpotpos, v.off_minpotvel = f["MinPotPos"][1], f["MinPotVel"][1]
    return v


class ExchangeLayout(C.Structure):
    _fields_ = [("part_elsize", C.c_size_t), ("off_flags", C.c_size_t), ("off_type", C.c_size_t), ("off_pi", C.c_size_t),
                ("slot_elsize", C.c_size_t * 6), ("off_reverselink", C.c_size_t)]


class SpawnLayout(C.Structure):
    _fields_ = [("off_id", C.c_size_t), ("off_mass", C.c_size_t), ("generation_shift", C.c_int), ("pad_", C.c_int)]


class BhParams(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("BoxSize", "ForceSoftening", "SeedBHDynMass", "atime", "a3inv", "hubble", "GravInternal", "BlackHoleAccretionFactor",
                                         "BlackHoleEddingtonFactor", "BlackHoleFeedbackFactor", "EddingtonConst", "UnitTime_in_s", "HubbleParam",
                                         "LightOverUnitVel", "MaxThermalU", "OmegaBaryon", "Hubble", "BHKE_EddingtonThrFactor", "BHKE_EddingtonMFactor",
                                         "BHKE_EddingtonMPivot", "BHKE_EddingtonMIndex", "BHKE_EffRhoFactor", "BHKE_EffCap", "BHKE_InjEnergyThr",
                                         "BHKE_SfrCritOverDensity")] + \
               [(k, C.c_int) for k in ("DensityKernelType", "WindsDecoupleSph", "RepositionEnabled", "MergeGravBound", "BH_DRAG", "BlackHoleKineticOn")]


class BhSlotView(C.Structure):
    _fields_ = [("base", C.c_void_p), ("elsize", C.c_size_t), ("numslots", C.c_int64)] + \
               [(k, C.c_size_t) for k in ("off_mass", "off_mdot", "off_density", "off_mtrack", "off_dfaccel", "off_vdisp", "off_kineticfdbkenergy", "off_dragaccel",
                                          "off_encounter", "off_countprogs", "off_mintimebin", "off_swallowid", "off_swallowtime")]


class BhWork(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("SPH_SwallowID", "BH_SwallowID", "BH_FeedbackWeightSum", "BH_Entropy", "BH_SurroundingGasVel", "MgasEnc", "KEflag",
                                          "BH_accreted_Mass", "BH_accreted_BHMass", "BH_accreted_momentum")]


def bh_slot_view(arr):
    """shq_bh_slot_view of a numpy array of BH_DTYPE records"""
    f = BH_DTYPE.fields
    v = BhSlotView()
    v.base, v.elsize, v.numslots = arr.ctypes.data, BH_DTYPE.itemsize, len(arr)
    for key, name in (("off_mass", "Mass"), ("off_mdot", "Mdot"), ("off_density", "Density"), ("off_mtrack", "Mtrack"), ("off_dfaccel", "DFAccel"),
                      ("off_vdisp", "VDisp"), ("off_kineticfdbkenergy", "KineticFdbkEnergy"), ("off_dragaccel", "DragAccel"), ("off_encounter", "encounter"),
                      ("off_countprogs", "CountProgs"), ("off_mintimebin", "minTimeBin"), ("off_swallowid", "SwallowID"), ("off_swallowtime", "SwallowTime")):
        setattr(v, key, f[name][1])
    return v


class GasMetalView(C.Structure):
    _fields_ = [("base", C.c_void_p), ("elsize", C.c_size_t), ("numslots", C.c_int64), ("off_density", C.c_size_t), ("off_metallicity", C.c_size_t),
                ("off_metals", C.c_size_t), ("nmetals", C.c_int), ("pad_", C.c_int)]


class WindParams(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("BoxSize", "Time", "WindFreeTravelLength", "MaxWindFreeTravelTime", "WindEfficiency", "WindSpeed", "WindSigma0",
                                         "WindSpeedFactor", "MinWindVelocity", "WindThermalFactor")] + [("WindModel", C.c_int), ("pad_", C.c_int)]


WIND_KICK_DTYPE = np.dtype([("part_index", "<i4"), ("pad_", "<i4"), ("StarDistance", "<f8"), ("StarID", "<u8"), ("StarKickVelocity", "<f8"), ("StarTherm", "<f8")])


class StarView(C.Structure):
    _fields_ = [("base", C.c_void_p), ("elsize", C.c_size_t), ("numslots", C.c_int64), ("off_vdisp", C.c_size_t)]


class StarSpawnLayout(C.Structure):
    _fields_ = [(k, C.c_size_t) for k in ("star_formationtime", "star_lastenrichmentmyr", "star_totalmassreturned", "star_birthdensity", "star_vdisp",
                                         "star_metallicity", "star_metals", "sph_density", "sph_vdisp", "sph_metallicity", "sph_metals")] + [("nmetals", C.c_int), ("pad_", C.c_int)]


class BhSeedLayout(C.Structure):
    _fields_ = [(k, C.c_size_t) for k in ("bh_mass", "bh_mseed", "bh_mdot", "bh_formationtime", "bh_swallowid", "bh_density", "bh_timebindynfric", "bh_minpotpos",
                                         "bh_dfaccel", "bh_df_surroundingvel", "bh_dragaccel", "bh_df_surroundingrmsvel", "bh_df_surroundingdensity", "bh_jumptominpot",
                                         "bh_countprogs", "bh_mtrack", "bh_kineticfdbkenergy", "bh_vdisp", "part_pos", "part_mass", "part_timebin_hydro")]


class ExchangeEntry(C.Structure):
    _fields_ = [("base", C.c_int64), ("slots", C.c_int64 * 6)]


# struct star_particle_data, libgadget/slotsmanager.h:77-92 (72 B)
STAR_DTYPE = np.dtype({"names": ["ReverseLink", "LastEnrichmentMyr", "TotalMassReturned", "Metallicity", "Metals", "VDisp", "BirthDensity", "FormationTime"],
                       "formats": ["<i4", "<f4", "<f8", "<f8", ("<f4", 9), "<f4", "<f4", "<f4"], "offsets": [0, 4, 8, 16, 24, 60, 64, 68], "itemsize": 72})


class FofParams(C.Structure):
    _fields_ = [("BoxSize", C.c_double), ("LinkingLength", C.c_double), ("PrimaryLinkTypes", C.c_int32), ("SecondaryLinkTypes", C.c_int32),
                ("HaloMinLength", C.c_int32), ("WindsDecoupleSph", C.c_int32)]


FOF_GROUP_DTYPE = np.dtype([("MinID", "<u8"), ("Length", "<i4"), ("GrNr", "<i4"), ("LenType", "<i4", 6), ("FirstPos", "<f4", 3), ("seed_index", "<i4"),
                            ("MassType", "<f8", 6), ("Mass", "<f8"), ("CM", "<f8", 3), ("Vel", "<f8", 3), ("Imom", "<f8", (3, 3)), ("Jmom", "<f8", 3),
                            ("MaxDens", "<f8"), ("first_member", "<i8")])


class TopNodeGeo(C.Structure):
    _fields_ = [("daughter", C.c_int32 * 8), ("leaf", C.c_int32), ("pad_", C.c_int32)]


class TopLeafMoments(C.Structure):
    _fields_ = [("s", C.c_double * 3), ("mass", C.c_double), ("hmax", C.c_double)]


TOPNODE_GEO_DTYPE = np.dtype([("daughter", "<i4", 8), ("leaf", "<i4"), ("pad_", "<i4")])
TOPLEAF_MOMENTS_DTYPE = np.dtype([("s", "<f8", 3), ("mass", "<f8"), ("hmax", "<f8")])


class Timeline(C.Structure):
    _fields_ = [("Ti_Current", C.c_int64), ("loga_now", C.c_double), ("Dloga_interval", C.c_double), ("nseg", C.c_int32), ("pad_", C.c_int32),
                ("seg_snap", C.c_int64 * 2), ("seg_loga", C.c_double * 3)]


class TimestepParams(C.Structure):
    _fields_ = [("ErrTolIntAccuracy", C.c_double), ("CourantFac", C.c_double), ("MinSizeTimestep", C.c_double), ("ForceSoftening", C.c_double),
                ("atime", C.c_double), ("hubble", C.c_double), ("fac3", C.c_double), ("dti_max", C.c_int64), ("ForceEqualTimesteps", C.c_int32),
                ("isFirstTimeStep", C.c_int32), ("mintimebin", C.c_int32), ("mingravtimebin", C.c_int32), ("tl", Timeline)]


class TimestepResult(C.Structure):
    _fields_ = [("badstepsizecount", C.c_int32), ("mTimeBin", C.c_int32), ("maxTimeBin", C.c_int32), ("mintimebin", C.c_int32),
                ("ntiaccel", C.c_int64), ("nticourant", C.c_int64), ("ntihsml", C.c_int64), ("ntiaccrete", C.c_int64), ("ntineighbour", C.c_int64),
                ("nbh", C.c_int64), ("dynratio", C.c_int64), ("maxdyndiff", C.c_int32), ("nbadbin", C.c_int32), ("dti_min", C.c_int64),
                ("timebincounts", C.c_int64 * (TIMEBINS + 1))]


class HierLevel(C.Structure):
    """shq_hier_level"""
    _fields_ = [("timebin", C.c_int32), ("walk_mode", C.c_int32), ("nparticles", C.c_int64), ("tree_nodes", C.c_int64),
                ("tree_build_ms", C.c_double), ("walk_ms", C.c_double)]


class DriftKickTimes(C.Structure):
    """DriftKickTimes, libgadget/timestep.h:10-26"""
    _fields_ = [("mintimebin", C.c_int), ("maxtimebin", C.c_int), ("mingravtimebin", C.c_int), ("Ti_kick", C.c_int64 * (TIMEBINS + 1)),
                ("Ti_lastactivedrift", C.c_int64 * (TIMEBINS + 1)), ("Ti_Current", C.c_int64), ("PM_length", C.c_int64), ("PM_start", C.c_int64),
                ("PM_kick", C.c_int64)]


class HostCosmo(C.Structure):
    _fields_ = [("OmegaBaryon", C.c_double), ("OmegaCDM", C.c_double), ("OmegaNu1", C.c_double), ("RhoCrit", C.c_double), ("Omega0", C.c_double),
                ("Hubble", C.c_double), ("GravInternal", C.c_double), ("hubble_now", C.c_double)]


class BhView(C.Structure):
    _fields_ = [("base", C.c_void_p), ("elsize", C.c_size_t), ("numslots", C.c_int64),
                ("off_density", C.c_size_t), ("off_divvel", C.c_size_t)]


def sph_view(SphP):
    """shq_sph_view of a numpy array of SPH_DTYPE records."""
    v = SphView()
    f = SphP.dtype.fields
    v.base, v.elsize, v.numslots = SphP.ctypes.data, SphP.dtype.itemsize, len(SphP)
    v.off_density, v.off_egywtdensity, v.off_entropy = f["Density"][1], f["EgyWtDensity"][1], f["Entropy"][1]
    v.off_dtentropy, v.off_maxsignalvel, v.off_hydroaccel = f["DtEntropy"][1], f["MaxSignalVel"][1], f["HydroAccel"][1]
    v.off_dhsmlegydensityfactor, v.off_divvel, v.off_curlvel = f["DhsmlEgyDensityFactor"][1], f["DivVel"][1], f["CurlVel"][1]
    v.off_delaytime = f["DelayTime"][1]
    return v


def bh_view(BhP):
    v = BhView()
    f = BhP.dtype.fields
    v.base, v.elsize, v.numslots = BhP.ctypes.data, BhP.dtype.itemsize, len(BhP)
    v.off_density, v.off_divvel = f["Density"][1], f["DivVel"][1]
    return v


DENSITY_QUERY_DTYPE = np.dtype({"names": ["Pos", "NodeList", "Vel", "Hsml", "Type"], "formats": [("<f8", 3), ("<i4", 4), ("<f8", 3), "<f8", "<i4"],
                                "offsets": [0, 24, 40, 64, 72], "itemsize": 80})
DENSITY_RESULT_DTYPE = np.dtype([("EgyRho", "<f8"), ("DhsmlEgyDensity", "<f8"), ("Rho", "<f8"), ("DhsmlDensity", "<f8"), ("Ngb", "<f8"),
                                 ("Div", "<f8"), ("Rot", "<f8", 3), ("GradRho", "<f8", 3)])
HYDRO_QUERY_DTYPE = np.dtype({"names": ["Pos", "NodeList", "EgyRho", "EntVarPred", "Vel", "Hsml", "Mass", "Density", "Pressure", "F1",
                                        "SPH_DhsmlDensityFactor", "TimeBinHydro"],
                              "formats": [("<f8", 3), ("<i4", 4), "<f8", "<f8", ("<f8", 3), "<f8", "<f8", "<f8", "<f8", "<f8", "<f8", "<i4"],
                              "offsets": [0, 24, 40, 48, 56, 80, 88, 96, 104, 112, 120, 128], "itemsize": 136})
HYDRO_RESULT_DTYPE = np.dtype([("Acc", "<f8", 3), ("DtEntropy", "<f8"), ("MaxSignalVel", "<f8")])
assert DENSITY_RESULT_DTYPE.itemsize == 96 and HYDRO_RESULT_DTYPE.itemsize == 40


class StellarParams(C.Structure):
    _fields_ = [("BoxSize", C.c_double), ("DesNumNgb", C.c_double), ("MaxNgbDeviation", C.c_double), ("SPHWeighting", C.c_int32),
                ("DensityKernelType", C.c_int32)]


class SphStats(C.Structure):
    _fields_ = [("ntargets", C.c_int64), ("ninteractions", C.c_int64), ("niterations", C.c_int32), ("pad_", C.c_int32),
                ("kernel_ms", C.c_double), ("hsml_max_tried", C.c_double)]


class PMParams(C.Structure):
    _fields_ = [("Nmesh", C.c_int32), ("pad_", C.c_int32), ("BoxSize", C.c_double), ("Asmth", C.c_double),
                ("G", C.c_double)]


# ---- prototypes ------------------------------------------------------------------------------
_vp = C.c_void_p
hip.shq_last_error.restype = C.c_char_p
hip.shq_version.restype = C.c_char_p
hip.shq_init.argtypes = [C.c_int, _vp, C.POINTER(_vp)]
hip.shq_shutdown.argtypes = [_vp]
hip.shq_shutdown.restype = None
hip.shq_synchronize.argtypes = [_vp]
hip.shq_stream.argtypes = [_vp]
hip.shq_stream.restype = _vp
hip.shq_timer_begin.argtypes = [_vp, C.c_int]
hip.shq_timer_end.argtypes = [_vp, C.c_int]
hip.shq_timer_elapsed_ms.argtypes = [_vp, C.c_int, C.POINTER(C.c_double)]
hip.shq_particles_upload.argtypes = [_vp, C.POINTER(PartView)]
hip.shq_tree_upload.argtypes = [_vp, C.POINTER(TreeView)]
hip.shq_tree_build.argtypes = [_vp, C.c_double, C.c_int, _vp, C.c_int64, C.POINTER(TreeBuildStats)]
hip.shq_tree_build.restype = C.c_int
hip.shq_tree_download.argtypes = [_vp, C.c_int64, _vp, C.c_int64, _vp, C.POINTER(C.c_int64)]
hip.shq_tree_download.restype = C.c_int
GRAV_QUERY_DTYPE = np.dtype({"names": ["Pos", "NodeList", "OldAcc"], "formats": [("<f8", 3), ("<i4", 4), "<f8"],
                             "offsets": [0, 24, 40], "itemsize": 48})
GRAV_RESULT_DTYPE = np.dtype({"names": ["Acc", "Potential"], "formats": [("<f8", 3), "<f8"], "offsets": [0, 24], "itemsize": 32})
hip.shq_grav_short_secondary.argtypes = [_vp, C.POINTER(GravParams), _vp, C.c_int64, _vp, _vp, C.c_int]
hip.shq_grav_short_secondary.restype = C.c_int
TOPLEAF_DTYPE = np.dtype([("Task", "<i4"), ("topnode", "<i4"), ("treenode", "<i4")])
DATA_INDEX_DTYPE = np.dtype([("Task", "<i4"), ("Index", "<i4"), ("NodeList", "<i4", 4)])
hip.shq_grav_reduce_export_results.argtypes = [_vp, _vp, _vp, C.c_int64, C.c_int]
hip.shq_grav_reduce_export_results.restype = C.c_int
hip.shq_grav_postprocess.argtypes = [_vp, C.POINTER(GravParams), _vp, C.c_int64, C.c_int]
hip.shq_grav_postprocess.restype = C.c_int
hip.shq_toptree_upload.argtypes = [_vp, C.POINTER(TreeView), _vp, C.c_int]
hip.shq_toptree_upload.restype = C.c_int
hip.shq_grav_toptree_exports.argtypes = [_vp, C.POINTER(GravParams), _vp, C.c_int64, _vp, _vp, C.c_int64, C.POINTER(C.c_int64)]
hip.shq_grav_toptree_exports.restype = C.c_int
hip.shq_ngb_toptree_exports.argtypes = [_vp, C.c_int, C.c_double, _vp, C.c_int64, _vp, _vp, C.c_int64, C.POINTER(C.c_int64)]
hip.shq_ngb_toptree_exports.restype = C.c_int
hip.shq_grav_toptree_exports_resident.argtypes = [_vp, C.POINTER(GravParams), _vp, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64), _vp]
hip.shq_grav_toptree_exports_resident.restype = C.c_int
hip.shq_grav_export_pack.argtypes = [_vp, _vp, _vp]
hip.shq_grav_export_pack.restype = C.c_int
GAS_NCOL = 28
hip.shq_gas_set_device.argtypes = [_vp, _vp, C.c_int64, C.c_int64]
hip.shq_gas_get_device.argtypes = [_vp, _vp, C.c_int64, C.c_int]
hip.shq_density_resident.argtypes = [_vp, C.POINTER(DensityParams), C.POINTER(SphStats)]
hip.shq_hydro_resident.argtypes = [_vp, C.POINTER(HydroParams), C.POINTER(SphStats)]
for _f in ("shq_gas_set_device", "shq_gas_get_device", "shq_density_resident", "shq_hydro_resident"):
    getattr(hip, _f).restype = C.c_int
hip.shq_set_walk_stats.argtypes = [_vp, C.c_int]
hip.shq_set_walk_stats.restype = C.c_int
hip.shq_set_walk_launch.argtypes = [_vp, C.c_int, C.c_int]
hip.shq_set_walk_launch.restype = C.c_int
hip.shq_set_walk_sparse.argtypes = [_vp, C.c_int]
hip.shq_set_walk_sparse.restype = C.c_int
hip.shq_walk_pair_lean.argtypes = [_vp]
hip.shq_walk_pair_lean.restype = C.c_int
hip.shq_set_walk_overlap.argtypes = [_vp, C.c_int]
hip.shq_set_walk_overlap.restype = C.c_int
hip.shq_pm_set_fft_transposed.argtypes = [_vp, C.c_int]
hip.shq_pm_set_fft_transposed.restype = C.c_int
hip.shq_walk_pair_status.argtypes = [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
hip.shq_walk_pair_status.restype = C.c_int
hip.shq_set_walk_debug.argtypes = [_vp, C.c_int, C.c_int]
hip.shq_set_walk_debug.restype = C.c_int
hip.shq_direct_force_sample.argtypes = [_vp, _vp, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_int, _vp]
hip.shq_direct_force_sample.restype = C.c_int
hip.shq_exchange_plan.argtypes = [_vp, C.POINTER(ExchangeLayout), _vp, C.c_int64, _vp, C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _vp]
hip.shq_exchange_pack.argtypes = [_vp, C.POINTER(ExchangeLayout), _vp, _vp, C.c_int64, _vp, C.c_int, _vp, _vp]
hip.shq_exchange_unpack.argtypes = [_vp, C.POINTER(ExchangeLayout), _vp, C.c_int64, _vp, _vp, _vp, C.c_int]
hip.shq_slots_gc.argtypes = [_vp, C.POINTER(ExchangeLayout), _vp, C.POINTER(C.c_int64), C.c_int64, _vp, _vp, _vp]
hip.shq_slots_gc_sorted.argtypes = [_vp, C.POINTER(ExchangeLayout), _vp, C.POINTER(C.c_int64), C.c_int64, _vp, _vp, _vp]
hip.shq_slots_split_particles.argtypes = [_vp, C.POINTER(ExchangeLayout), C.POINTER(SpawnLayout), _vp, C.POINTER(C.c_int64), C.c_int64, _vp, _vp, C.c_int64, _vp]
hip.shq_slots_convert.argtypes = [_vp, C.POINTER(ExchangeLayout), _vp, C.c_int64, C.c_int64, _vp, _vp, _vp, _vp, C.c_int64, C.c_int]
hip.shq_make_particle_stars.argtypes = [_vp, C.POINTER(ExchangeLayout), C.POINTER(StarSpawnLayout), _vp, C.c_int64, C.c_int64, _vp, _vp, _vp, _vp, _vp, C.c_int64, C.c_double]
hip.shq_blackhole_make_seeds.argtypes = [_vp, C.POINTER(ExchangeLayout), C.POINTER(BhSeedLayout), _vp, C.c_int64, C.c_int64, _vp, _vp, _vp, _vp, _vp, C.c_int64,
                                         C.c_double, C.c_double]
hip.shq_bh_accretion.argtypes = [_vp, C.POINTER(TreeView), C.POINTER(PartView), C.POINTER(SphView), C.POINTER(BhSlotView), _vp, _vp, C.c_int64, C.POINTER(KickFactors),
                                 C.POINTER(BhParams), C.c_int64, _vp, C.c_int64, C.POINTER(BhWork)]
hip.shq_bh_feedback.argtypes = [_vp, C.POINTER(TreeView), C.POINTER(PartView), C.POINTER(SphView), C.POINTER(BhSlotView), _vp, _vp, C.c_int64, C.POINTER(KickFactors),
                                C.POINTER(BhParams), C.c_int64, _vp, C.c_int64, _vp, C.POINTER(BhWork), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
hip.shq_bh_accretion.restype = hip.shq_bh_feedback.restype = C.c_int
hip.shq_winds_and_feedback.argtypes = [_vp, C.POINTER(TreeView), C.POINTER(PartView), C.POINTER(SphView), C.POINTER(StarView), _vp, _vp, C.c_int64, C.POINTER(WindParams),
                                       _vp, C.c_int64, _vp, _vp, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
hip.shq_winds_and_feedback.restype = C.c_int
hip.shq_winds_candidates.argtypes = [_vp, C.POINTER(TreeView), C.POINTER(PartView), C.POINTER(SphView), C.POINTER(StarView), _vp, _vp, C.c_int64, C.POINTER(WindParams),
                                     _vp, C.c_int64, _vp, _vp, C.c_int64, C.POINTER(C.c_int64)]
hip.shq_winds_apply.argtypes = [_vp, C.POINTER(PartView), C.POINTER(SphView), _vp, _vp, C.c_int64, C.POINTER(WindParams), _vp, C.c_int64, C.POINTER(C.c_int64)]
hip.shq_winds_candidates.restype = hip.shq_winds_apply.restype = C.c_int
hip.shq_metal_return.argtypes = [_vp, C.POINTER(TreeView), C.POINTER(PartView), C.POINTER(GasMetalView), _vp, C.c_int64, _vp, _vp, _vp, _vp, C.c_double, C.c_int, C.c_int, _vp,
                                 C.POINTER(C.c_int64)]
hip.shq_metal_return.restype = C.c_int
hip.shq_domain_maintain_topleaf.argtypes = [_vp, C.c_int, C.c_int64, _vp, _vp, C.POINTER(C.c_int64)]
hip.shq_domain_maintain_topleaf.restype = C.c_int
hip.shq_winds_evolve.argtypes = [_vp, C.POINTER(PartView), C.POINTER(SphView), _vp, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double, C.POINTER(KickFactors)]
hip.shq_winds_subgrid.argtypes = [_vp, C.POINTER(PartView), C.POINTER(SphView), C.c_size_t, _vp, _vp, C.c_int64, _vp, C.POINTER(WindParams), _vp, C.c_int64,
                                  C.POINTER(C.c_int64)]
hip.shq_winds_evolve.restype = hip.shq_winds_subgrid.restype = C.c_int
hip.shq_sph_state_upload.argtypes = [_vp, C.POINTER(PartView), C.POINTER(SphView)]
hip.shq_sph_state_upload.restype = C.c_int
hip.shq_fof_group_sums.argtypes = [_vp, _vp, C.c_int, _vp]
hip.shq_fof_group_sums.restype = C.c_int
hip.shq_fof_seed_select.argtypes = [_vp, C.c_double, C.c_double, _vp, C.c_int64, C.POINTER(C.c_int64)]
for _f in ("shq_make_particle_stars", "shq_blackhole_make_seeds", "shq_fof_seed_select"):
    getattr(hip, _f).restype = C.c_int
for _f in ("shq_exchange_plan", "shq_exchange_pack", "shq_exchange_unpack", "shq_slots_gc", "shq_slots_gc_sorted", "shq_slots_split_particles", "shq_slots_convert"):
    getattr(hip, _f).restype = C.c_int
hip.shq_fof.argtypes = [_vp, C.POINTER(FofParams), _vp, _vp, _vp, C.POINTER(C.c_int64)]
hip.shq_fof.restype = C.c_int
hip.shq_fof_groups_download.argtypes = [_vp, _vp, C.c_int64]
hip.shq_fof_groups_download.restype = C.c_int
hip.shq_fof_members.argtypes = [_vp, _vp, C.c_int64, C.POINTER(C.c_int64)]
hip.shq_fof_members.restype = C.c_int
hip.shq_tree_build_domain.argtypes = [_vp, C.c_double, C.c_int, _vp, C.c_int64, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int64, _vp, _vp]
hip.shq_tree_build_domain.restype = C.c_int
hip.shq_tree_set_topleaf_moments.argtypes = [_vp, _vp, C.c_int]
hip.shq_tree_set_topleaf_moments.restype = C.c_int
_tsp, _tsr = C.POINTER(TimestepParams), C.POINTER(TimestepResult)
hip.shq_find_timesteps.argtypes = [_vp, _tsp, _vp, C.c_int64, C.c_int64, C.c_int, _tsr]
hip.shq_find_global_timestep.argtypes = [_vp, _tsp, _tsr]
hip.shq_find_hydro_timesteps.argtypes = [_vp, _tsp, _vp, C.c_int64, _tsr]
hip.shq_set_bh_first_timestep.argtypes = [_vp, C.c_int]
hip.shq_hier_gravity_bins.argtypes = [_vp, _tsp, _vp, C.c_int64, C.c_int, C.c_int, _tsr]
hip.shq_hier_push_down.argtypes = [_vp, _vp, C.c_int64, C.c_int]
hip.shq_hier_refine.argtypes = [_vp, _tsp, _vp, C.c_int64, C.c_int, C.c_int, _tsr]
hip.shq_velocity_moments.argtypes = [_vp, _vp, _vp, _vp]
hip.shq_timebins_download.argtypes = [_vp, _vp, _vp]
hip.shq_maxsignalvel_upload.argtypes = [_vp, _vp]
hip.shq_bh_dynamics_upload.argtypes = [_vp, C.POINTER(PartView), C.POINTER(BhDynView)]
hip.shq_bh_dynamics_download.argtypes = [_vp, C.POINTER(PartView), C.POINTER(BhDynView)]
hip.shq_set_bh_reposition.argtypes = [_vp, C.c_int]
hip.shq_kick_bh.argtypes = [_vp, _vp, _vp, C.c_int64]
for _f in ("shq_find_timesteps", "shq_find_global_timestep", "shq_find_hydro_timesteps", "shq_set_bh_first_timestep", "shq_hier_gravity_bins",
           "shq_hier_push_down", "shq_hier_refine", "shq_velocity_moments", "shq_timebins_download", "shq_maxsignalvel_upload",
           "shq_bh_dynamics_upload", "shq_bh_dynamics_download", "shq_set_bh_reposition", "shq_kick_bh"):
    getattr(hip, _f).restype = C.c_int
hip.shq_pm_slab_pitch.argtypes = [C.c_int]
hip.shq_pm_slab_pitch.restype = C.c_int
hip.shq_pm_slab2_deposit.argtypes = [_vp, C.POINTER(PMParams), C.c_int, C.c_int, C.c_int, C.c_int, _vp]
hip.shq_pm_slab2_deposit.restype = C.c_int
hip.shq_pm_slab2_deposit_ghosts.argtypes = [_vp, C.POINTER(PMParams), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]
hip.shq_pm_slab2_deposit_ghosts.restype = C.c_int
hip.shq_set_inputs_current.argtypes = [_vp, C.c_int]
hip.shq_set_inputs_current.restype = C.c_int
CURRENT_PARTICLES, CURRENT_SPH, CURRENT_TREE, CURRENT_IDS = 1, 2, 4, 8
hip.shq_pm_slab2_fft_yz.argtypes = [_vp, C.c_int, _vp, C.c_int, C.c_int]
hip.shq_pm_slab2_fft_yz_packed.argtypes = [_vp, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
hip.shq_pm_slab2_fft_yz_packed.restype = C.c_int
hip.shq_pm_slab2_fft_yz.restype = C.c_int
hip.shq_pm_slab2_xgreen.argtypes = [_vp, C.POINTER(PMParams), _vp, C.c_int, C.c_int]
hip.shq_pm_slab2_xgreen.restype = C.c_int
hip.shq_pm_slab2_readout.argtypes = [_vp, C.POINTER(PMParams), C.c_int, C.c_int, C.c_int, C.c_int, _vp]
hip.shq_pm_slab2_readout.restype = C.c_int
hip.shq_pm_measure_power.argtypes = [_vp, C.c_int]
hip.shq_pm_measure_power.restype = C.c_int
hip.shq_pm_download_power.argtypes = [_vp, C.c_int, _vp, _vp, _vp, _vp]
hip.shq_pm_download_power.restype = C.c_int
hip.shq_dynamics_upload.argtypes = [_vp, C.POINTER(PartView)]
hip.shq_dynamics_upload.restype = C.c_int
hip.shq_drift.argtypes = [_vp, C.c_double, C.c_double, _vp]
hip.shq_drift.restype = C.c_int
hip.shq_kick_short.argtypes = [_vp, _vp, _vp, C.c_int64, C.c_int]
hip.shq_kick_short.restype = C.c_int
hip.shq_timebins_upload.argtypes = [_vp, _vp, _vp]
hip.shq_timebins_upload.restype = C.c_int
hip.shq_build_active_particles.argtypes = [_vp, C.c_int64, C.c_int, C.POINTER(ActiveInfo)]
hip.shq_build_active_particles.restype = C.c_int
hip.shq_build_active_sublist.argtypes = [_vp, C.c_int, C.c_int64, C.POINTER(C.c_int64)]
hip.shq_build_active_sublist.restype = C.c_int
hip.shq_hier_gravity_levels.argtypes = [_vp, _tsp, C.POINTER(GravParams), C.c_double, C.c_int, C.c_int64, C.c_int, _vp, C.c_int, C.POINTER(C.c_int),
                                        C.POINTER(C.c_int64), C.POINTER(HierLevel), C.POINTER(C.c_int)]
hip.shq_hier_gravity_levels.restype = C.c_int
hip.shq_active_download.argtypes = [_vp, C.c_int, _vp, C.c_int64, C.POINTER(C.c_int64)]
hip.shq_active_download.restype = C.c_int
hip.shq_kick_hydro.argtypes = [_vp, _vp, _vp, C.c_double, C.c_double, _vp, C.c_int64, C.c_int, C.POINTER(C.c_int64)]
hip.shq_kick_hydro.restype = C.c_int
hip.shq_entropy_download.argtypes = [_vp, _vp]
hip.shq_entropy_download.restype = C.c_int
hip.shq_kick_pm.argtypes = [_vp, C.c_double]
hip.shq_kick_pm.restype = C.c_int
hip.shq_dynamics_download.argtypes = [_vp, C.POINTER(PartView)]
hip.shq_dynamics_download.restype = C.c_int
hip.shq_grav_short_run.argtypes = [_vp, C.POINTER(GravParams), _vp, C.c_int64, C.c_int, C.c_int]
hip.shq_grav_short_run_range.argtypes = [_vp, C.POINTER(GravParams), C.c_int64, C.c_int64, C.c_int, C.c_int]
hip.shq_hilbert_order.argtypes = [_vp, C.c_void_p, C.c_int64, C.c_double, C.c_void_p]
hip.shq_grav_short_download.argtypes = [_vp, _vp, _vp, _vp, C.POINTER(WalkStats)]
hip.shq_grav_refresh_oldacc.argtypes = [_vp, C.c_double]
hip.shq_grav_short_tree.argtypes = [_vp, C.POINTER(TreeView), C.POINTER(PartView), _vp, C.c_int64,
                                    C.POINTER(GravParams), _vp, C.c_int, C.c_int, C.POINTER(WalkStats)]
hip.shq_pm_force.argtypes = [_vp, C.POINTER(PMParams), C.POINTER(PartView), _vp, _vp]
hip.shq_pm_run.argtypes = [_vp, C.POINTER(PMParams)]
hip.shq_pm_download.argtypes = [_vp, _vp, _vp]
hip.shq_treepm_step.argtypes = [_vp, C.POINTER(PMParams), C.POINTER(GravParams), C.c_int, C.c_int]
class PMTransfer(C.Structure):
    _fields_ = [("kind", C.c_int32), ("axis", C.c_int32), ("zero_mode", C.c_int32), ("pad_", C.c_int32), ("table", C.c_void_p)]


hip.shq_pm_apply.argtypes = [_vp, C.c_int, _vp, C.POINTER(PMTransfer), _vp]
hip.shq_pm_apply.restype = C.c_int
hip.shq_timer_between_ms.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
hip.shq_timer_between_ms.restype = C.c_int
hip.shq_pm_start.argtypes = [_vp, _vp, C.c_double]
hip.shq_pm_start.restype = C.c_int
hip.shq_treepm_last_fused.argtypes = [_vp, C.POINTER(C.c_int)]
hip.shq_treepm_set_fuse.argtypes = [_vp, C.c_int]
hip.shq_pm_phase_ms.argtypes = [_vp, C.POINTER(C.c_double * 6)]
hip.shq_pm_set_debug.argtypes = [_vp, C.c_int]
hip.shq_pm_set_mesh_scrub.argtypes = [_vp, C.c_int]
hip.shq_pm_mesh_prezeroed.argtypes = [_vp, C.POINTER(C.c_int)]
hip.shq_pm_download_mesh.argtypes = [_vp, C.c_int, _vp]
hip.shq_fft_r2c.argtypes = [_vp, C.c_int, _vp, _vp]
hip.shq_fft_r2c_xyz.argtypes = [_vp, C.c_int, _vp, _vp]
hip.shq_fft_c2r_xyz.argtypes = [_vp, C.c_int, _vp, _vp]
hip.shq_fft_c2r.argtypes = [_vp, C.c_int, _vp, _vp]

host.shqh_last_error.restype = C.c_char_p
GRAVKICK_CB = C.CFUNCTYPE(C.c_double, C.c_int64, C.c_int64, C.c_void_p)
host.shqh_timebinmgr_create.argtypes = [_vp, C.c_int]
host.shqh_timebinmgr_create.restype = _vp
host.shqh_timebinmgr_destroy.argtypes = [_vp]
host.shqh_timebinmgr_destroy.restype = None
host.shqh_timebinmgr_set_gravkick.argtypes = [_vp, GRAVKICK_CB, _vp]
host.shqh_timebinmgr_set_gravkick.restype = None
host.shqh_tbm_ti_from_loga.argtypes = [_vp, C.c_double]
host.shqh_tbm_ti_from_loga.restype = C.c_int64
host.shqh_tbm_loga_from_ti.argtypes = [_vp, C.c_int64]
host.shqh_tbm_loga_from_ti.restype = C.c_double
host.shqh_tbm_dti_from_dloga.argtypes = [_vp, C.c_double, C.c_int64]
host.shqh_tbm_dti_from_dloga.restype = C.c_int64
host.shqh_tbm_dloga_from_dti.argtypes = [_vp, C.c_int64, C.c_int64]
host.shqh_tbm_dloga_from_dti.restype = C.c_double
host.shqh_tbm_get_dloga_for_bin.argtypes = [_vp, C.c_int, C.c_int64]
host.shqh_tbm_get_dloga_for_bin.restype = C.c_double
host.shqh_tbm_find_next_ti_sync.argtypes = [_vp, C.c_int64]
host.shqh_tbm_find_next_ti_sync.restype = C.c_int64
host.shqh_tbm_timeline_at.argtypes = [_vp, C.c_int64, C.POINTER(Timeline)]
host.shqh_tbm_timeline_at.restype = None
host.shqh_round_down_power_of_two.argtypes = [C.c_int64]
host.shqh_round_down_power_of_two.restype = C.c_int64
host.shqh_get_timestep_bin.argtypes = [C.c_int64]
host.shqh_is_timebin_active.argtypes = [C.c_int, C.c_int64]
host.shqh_set_timestep_params.argtypes = [C.c_double, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double]
host.shqh_set_timestep_params.restype = None
host.shqh_find_timesteps.argtypes = [_vp, C.c_int, C.c_int64, C.POINTER(DriftKickTimes), _vp, C.c_double, C.c_int, C.POINTER(HostCosmo), C.c_double,
                                     C.c_int, C.POINTER(C.c_int)]
host.shqh_find_hydro_timesteps.argtypes = [_vp, C.c_int, C.c_int64, C.POINTER(DriftKickTimes), _vp, C.c_double, C.POINTER(HostCosmo), C.c_int,
                                           C.POINTER(C.c_int)]
host.shqh_hierarchical_gravity_and_timesteps.argtypes = [_vp, C.c_int, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_int, C.c_double, C.c_int,
                                                         C.POINTER(DriftKickTimes), _vp, C.c_double, C.c_int, C.c_int, C.POINTER(HostCosmo), C.c_int,
                                                         C.POINTER(C.c_int64)]
host.shqh_partmanager_create.argtypes = [_vp, C.c_int64, C.c_double]
host.shqh_partmanager_create.restype = _vp
host.shqh_partmanager_free.argtypes = [_vp]
host.shqh_partmanager_free.restype = None
host.shqh_force_tree_rebuild_mask.argtypes = [_vp, C.c_int, _vp, C.c_int64, C.c_int]
host.shqh_force_tree_rebuild_mask.restype = _vp
host.shqh_force_tree_free.argtypes = [_vp]
host.shqh_force_tree_free.restype = None
host.shqh_tree_info.argtypes = [_vp, C.POINTER(C.c_int64 * 5)]
host.shqh_tree_info.restype = None
host.shqh_tree_nodes.argtypes = [_vp]
host.shqh_tree_nodes.restype = _vp
host.shqh_tree_father.argtypes = [_vp]
host.shqh_tree_father.restype = _vp
host.shqh_tree_view.argtypes = [_vp, C.POINTER(TreeView)]
host.shqh_tree_view.restype = None
host.shqh_part_view.argtypes = [_vp, C.POINTER(PartView)]
host.shqh_part_view.restype = None
host.shqh_set_kernel_table.argtypes = [_vp]
host.shqh_set_gravshort_treepar.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int, C.c_double, C.c_double, C.c_int]
host.shqh_set_gravshort_treepar.restype = None
host.shqh_gravshort_set_softenings.argtypes = [C.c_double]
host.shqh_gravshort_set_softenings.restype = None
host.shqh_FORCE_SOFTENING.restype = C.c_double
host.shqh_make_grav_params.argtypes = [C.c_double, C.c_double, C.c_int, C.c_double, C.c_double, C.POINTER(GravParams)]
host.shqh_grav_short_tree.argtypes = [_vp, _vp, _vp, C.c_double, C.c_int, C.c_double, _vp, C.c_int64, _vp,
                                      C.c_double, C.c_int, C.c_int, C.POINTER(WalkStats)]
host.shqh_gravpm_force.argtypes = [_vp, _vp, C.c_double, C.c_int, C.c_double, C.c_int]
host.shqh_synth_positions.argtypes = [C.c_int, C.c_int64, C.c_uint64, C.c_double, _vp]
host.shqh_synth_positions.restype = None
host.shqh_synth_positions_range.argtypes = [C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_uint64, C.c_double, _vp]
host.shqh_synth_positions_range.restype = None
host.shqh_morton_order.argtypes = [_vp, C.c_int64, C.c_double, _vp]
host.shqh_morton_order.restype = None
host.shqh_hilbert_order.argtypes = [_vp, C.c_int64, C.c_double, _vp]
host.shqh_hilbert_order.restype = None


def check(rc, where="shq"):
    if rc != 0:
        msg = hip.shq_last_error().decode() or host.shqh_last_error().decode()
        raise ShqError(f"{where} failed with status {rc}: {msg}")


def check_host(rc, where="shqh"):
    if rc != 0:
        msg = host.shqh_last_error().decode() or hip.shq_last_error().decode()
        raise ShqError(f"{where} failed with status {rc}: {msg}")


def ptr(a):
    """void* of a C-contiguous numpy array (None -> NULL)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def load_kernel_table():
    tab = np.fromfile(os.path.join(DATADIR, "shortrange_force_kernels.f64"), dtype="<f8").reshape(NGRAVTAB, 5)
    return np.ascontiguousarray(tab)


_KERNELS = load_kernel_table()
host.shqh_set_kernel_table(ptr(_KERNELS))

# ---- SPH host-mirror prototypes --------------------------------------------------------------
hip.shq_density.argtypes = [_vp, C.POINTER(TreeView), _vp, C.POINTER(PartView), C.POINTER(SphView), C.POINTER(BhView),
                            _vp, C.c_int64, C.POINTER(DensityParams), _vp, _vp, C.POINTER(SphStats)]
_i64p = C.POINTER(C.c_int64)
hip.shq_density_open.argtypes = [_vp, C.POINTER(TreeView), C.POINTER(PartView), C.POINTER(SphView), C.POINTER(BhView), _vp, C.c_int64,
                                 C.POINTER(DensityParams), C.c_int, _i64p]
hip.shq_density_ev_primary.argtypes = [_vp]
hip.shq_density_ev_secondary.argtypes = [_vp, C.POINTER(DensityParams), _vp, C.c_int64, _vp, _i64p]
hip.shq_density_ev_reduce.argtypes = [_vp, _vp, _vp, C.c_int64]
hip.shq_density_ev_postprocess.argtypes = [_vp, _i64p]
hip.shq_density_close.argtypes = [_vp, _vp, C.POINTER(PartView), C.POINTER(SphView), C.POINTER(BhView), _vp, _vp, C.POINTER(SphStats)]
hip.shq_hydro_open.argtypes = [_vp, C.POINTER(TreeView), C.POINTER(PartView), C.POINTER(SphView), _vp, C.c_int64, C.POINTER(HydroParams),
                               _vp, _i64p]
hip.shq_hydro_ev_primary.argtypes = [_vp]
hip.shq_hydro_ev_secondary.argtypes = [_vp, C.POINTER(HydroParams), _vp, C.c_int64, _vp, _i64p]
hip.shq_hydro_ev_reduce.argtypes = [_vp, _vp, _vp, C.c_int64]
hip.shq_hydro_ev_postprocess.argtypes = [_vp]
hip.shq_hydro_close.argtypes = [_vp, C.POINTER(PartView), C.POINTER(SphView), C.POINTER(SphStats)]
hip.shq_sph_exports.argtypes = [_vp, _vp, _vp, C.c_int64, _i64p]
hip.shq_sph_fill_queries.argtypes = [_vp, _vp, C.c_int64, _vp]
for _f in ("shq_density_open", "shq_density_ev_primary", "shq_density_ev_secondary", "shq_density_ev_reduce", "shq_density_ev_postprocess",
           "shq_density_close", "shq_hydro_open", "shq_hydro_ev_primary", "shq_hydro_ev_secondary", "shq_hydro_ev_reduce",
           "shq_hydro_ev_postprocess", "shq_hydro_close", "shq_sph_exports", "shq_sph_fill_queries"):
    getattr(hip, _f).restype = C.c_int
hip.shq_stellar_density.argtypes = [_vp, C.POINTER(TreeView), C.POINTER(PartView), C.POINTER(SphView), _vp, C.c_int64,
                                    C.POINTER(StellarParams), _vp, C.POINTER(SphStats)]
hip.shq_stellar_density.restype = C.c_int
hip.shq_bh_veldisp.argtypes = [_vp, C.POINTER(TreeView), C.POINTER(PartView), _vp, C.c_int64, C.POINTER(KickFactors), _vp, _vp, _vp, _vp]
hip.shq_bh_veldisp.restype = C.c_int
hip.shq_wind_veldisp.argtypes = [_vp, C.POINTER(TreeView), C.POINTER(PartView), _vp, C.c_int64, C.POINTER(KickFactors), C.c_double, C.c_double,
                                 _vp, C.POINTER(SphStats)]
hip.shq_wind_veldisp.restype = C.c_int
class BhDynFricOut(C.Structure):
    _fields_ = [("MinPot", _vp), ("MinPotPos", _vp), ("MinPotVel", _vp), ("updated", _vp), ("DF_SurroundingDensity", _vp),
                ("DF_SurroundingVel", _vp), ("DF_SurroundingRmsVel", _vp)]


hip.shq_bh_dynfric.argtypes = [_vp, C.POINTER(TreeView), C.POINTER(PartView), _vp, C.c_int64, C.POINTER(KickFactors), C.c_int, C.c_int, C.c_int,
                               C.POINTER(BhDynFricOut)]
hip.shq_bh_dynfric.restype = C.c_int
hip.shq_hydro_force.argtypes = [_vp, C.POINTER(TreeView), C.POINTER(PartView), C.POINTER(SphView), _vp, C.c_int64,
                                C.POINTER(HydroParams), _vp, C.POINTER(SphStats)]
host.shqh_set_densitypar.argtypes = [C.c_double, C.c_double, C.c_int, C.c_double, C.c_double]
host.shqh_set_densitypar.restype = None
host.shqh_GetNumNgb.restype = C.c_double
host.shqh_set_hydropar.argtypes = [C.c_int, C.c_double, C.c_double]
host.shqh_set_hydropar.restype = None
host.shqh_set_init_hsml.argtypes = [_vp, C.c_double, _vp]
host.shqh_force_tree_update_hmax.argtypes = [_vp, _vp]
host.shqh_force_tree_update_hmax.restype = None
host.shqh_density.argtypes = [_vp, _vp, _vp, _vp, C.c_int64, _vp, C.c_int64, _vp, C.c_int64, C.c_int, C.c_int, C.c_int,
                              C.POINTER(KickFactors), _vp, _vp, C.c_int, C.POINTER(SphStats)]
host.shqh_hydro_force.argtypes = [_vp, _vp, _vp, _vp, C.c_int64, _vp, C.c_int64, C.c_double, C.c_double, _vp,
                                  C.POINTER(KickFactors), _vp, C.c_int, C.POINTER(SphStats)]

# ---- multi-GPU slab entry points -------------------------------------------------------------
hip.shq_particles_set_device.argtypes = [_vp, _vp, C.c_int64, C.c_int64, C.c_int]
hip.shq_pm_slab_deposit.argtypes = [_vp, C.POINTER(PMParams), C.c_int, C.c_int, _vp]
hip.shq_pm_slab_green.argtypes = [_vp, C.POINTER(PMParams), C.c_int, C.c_int, _vp]
hip.shq_pm_slab_readout.argtypes = [_vp, C.POINTER(PMParams), C.c_int, C.c_int, _vp]
hip.shq_pm_get_deposit_log2scale.argtypes = [_vp]
hip.shq_pm_set_deposit_log2scale.argtypes = [_vp, C.c_int]
