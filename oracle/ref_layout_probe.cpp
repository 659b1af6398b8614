/* ref_layout_probe.cpp — own driver (test infrastructure): compiles the reference's partmanager.h and slotsmanager.h where they
 * lie under /root/reference (no stand-ins: they need only <mpi.h>, which the image has under /opt/conda/include) and prints, as
 * JSON, the layout facts the C-ABI relies on: the views INTEGRATION.md builds (the same initialisers, verbatim), the struct
 * sizes, and where the IsGarbage / Swallowed bits really sit.  tests/test_layout_probe.py compares them with
 * shenqi_amd/capi.py's dtypes.  Built by `make -C oracle ref` into oracle/_ref/. */
#include <stddef.h>
#include <stdio.h>
#include <string.h>
#include "partmanager.h"
#include "slotsmanager.h"
#include "shenqi_hip.h"

/* ---- the reference-side binding of INTEGRATION.md, verbatim ---- */
static shq_part_view part_view(particle_data * P, int64_t n) {
    shq_part_view v = { P, sizeof(particle_data), n,
        offsetof(particle_data, Pos), offsetof(particle_data, Mass),
        offsetof(particle_data, Type), offsetof(particle_data, TimeBinHydro) - 1 /* the byte holding IsGarbage, Swallowed */,
        offsetof(particle_data, PI), offsetof(particle_data, Vel),
        offsetof(particle_data, FullTreeGravAccel), offsetof(particle_data, GravPM),
        offsetof(particle_data, Potential), offsetof(particle_data, Hsml), offsetof(particle_data, DtHsml),
        offsetof(particle_data, TimeBinHydro), offsetof(particle_data, TimeBinGravity) };
    return v;
}
static shq_sph_view sph_view(sph_particle_data * S, int64_t n) {
    shq_sph_view sv = { S, sizeof(sph_particle_data), n,
        offsetof(sph_particle_data, Density), offsetof(sph_particle_data, EgyWtDensity),
        offsetof(sph_particle_data, Entropy), offsetof(sph_particle_data, DtEntropy),
        offsetof(sph_particle_data, MaxSignalVel), offsetof(sph_particle_data, HydroAccel),
        offsetof(sph_particle_data, DhsmlEgyDensityFactor), offsetof(sph_particle_data, DivVel),
        offsetof(sph_particle_data, CurlVel), offsetof(sph_particle_data, DelayTime) };
    return sv;
}
static shq_bh_view bh_view(bh_particle_data * B, int64_t n) {
    shq_bh_view bv = { B, sizeof(bh_particle_data), n,
        offsetof(bh_particle_data, Density), offsetof(bh_particle_data, DivVel) };
    return bv;
}

static int first_set_bit(const void *p, size_t n)
{
    const unsigned char *b = (const unsigned char *) p;
    for(size_t i = 0; i < n; i++)
        for(int k = 0; k < 8; k++)
            if(b[i] >> k & 1)
                return (int) (8 * i + k);
    return -1;
}

int main()
{
    particle_data P;
    const shq_part_view pv = part_view(&P, 1);
    const shq_sph_view sv = sph_view(nullptr, 0);
    const shq_bh_view bv = bh_view(nullptr, 0);
    memset(&P, 0, sizeof(P));
    P.IsGarbage = 1;
    const int bit_garbage = first_set_bit(&P, sizeof(P));
    memset(&P, 0, sizeof(P));
    P.Swallowed = 1;
    const int bit_swallowed = first_set_bit(&P, sizeof(P));
    memset(&P, 0, sizeof(P));
    P.Generation = 1;
    const int bit_generation = first_set_bit(&P, sizeof(P));
    P.Generation = 15;
    P.Generation++; /* the 4-bit field wraps: slots_split_particle's "generation wrapped" test can never fire */
    const int generation_after_15 = P.Generation;
    printf("{\"sizeof_star_particle_data\": %zu, \"star_particle_data\": {\"ReverseLink\": %zu, \"Metals\": %zu, \"FormationTime\": %zu},\n",
           sizeof(star_particle_data), offsetof(star_particle_data, ReverseLink), offsetof(star_particle_data, Metals), offsetof(star_particle_data, FormationTime));
    printf(" \"sizeof_particle_data\": %zu, \"sizeof_sph_particle_data\": %zu, \"sizeof_bh_particle_data\": %zu,\n", sizeof(particle_data),
           sizeof(sph_particle_data), sizeof(bh_particle_data));
    printf(" \"part_view\": {\"elsize\": %zu, \"off_pos\": %zu, \"off_mass\": %zu, \"off_type\": %zu, \"off_flags\": %zu, \"off_pi\": %zu, \"off_vel\": %zu, "
           "\"off_treeacc\": %zu, \"off_gravpm\": %zu, \"off_potential\": %zu, \"off_hsml\": %zu, \"off_dthsml\": %zu, \"off_timebin_hydro\": %zu, "
           "\"off_timebin_gravity\": %zu},\n",
           pv.elsize, pv.off_pos, pv.off_mass, pv.off_type, pv.off_flags, pv.off_pi, pv.off_vel, pv.off_treeacc, pv.off_gravpm, pv.off_potential,
           pv.off_hsml, pv.off_dthsml, pv.off_timebin_hydro, pv.off_timebin_gravity);
    printf(" \"particle_data\": {\"TopLeaf\": %zu, \"Ti_drift\": %zu, \"ID\": %zu, \"GrNr\": %zu},\n", offsetof(particle_data, TopLeaf),
           offsetof(particle_data, Ti_drift), offsetof(particle_data, ID), offsetof(particle_data, GrNr));
    printf(" \"bit_IsGarbage\": %d, \"bit_Swallowed\": %d, \"bit_Generation\": %d, \"generation_after_15\": %d,\n", bit_garbage, bit_swallowed, bit_generation,
           generation_after_15);
    printf(" \"sph_view\": {\"elsize\": %zu, \"off_density\": %zu, \"off_egywtdensity\": %zu, \"off_entropy\": %zu, \"off_dtentropy\": %zu, "
           "\"off_maxsignalvel\": %zu, \"off_hydroaccel\": %zu, \"off_dhsmlegydensityfactor\": %zu, \"off_divvel\": %zu, \"off_curlvel\": %zu, "
           "\"off_delaytime\": %zu},\n",
           sv.elsize, sv.off_density, sv.off_egywtdensity, sv.off_entropy, sv.off_dtentropy, sv.off_maxsignalvel, sv.off_hydroaccel,
           sv.off_dhsmlegydensityfactor, sv.off_divvel, sv.off_curlvel, sv.off_delaytime);
    printf(" \"sph_particle_data\": {\"ReverseLink\": %zu, \"Sfr\": %zu, \"Ne\": %zu, \"VDisp\": %zu, \"Metallicity\": %zu, \"Metals\": %zu},\n",
           offsetof(sph_particle_data, ReverseLink), offsetof(sph_particle_data, Sfr), offsetof(sph_particle_data, Ne),
           offsetof(sph_particle_data, VDisp), offsetof(sph_particle_data, Metallicity), offsetof(sph_particle_data, Metals));
    printf(" \"bh_view\": {\"elsize\": %zu, \"off_density\": %zu, \"off_divvel\": %zu},\n", bv.elsize, bv.off_density, bv.off_divvel);
    /* shq_bh_dyn_view as the reference-side shim fills it */
    const shq_bh_dyn_view dv = { nullptr, sizeof(bh_particle_data), 0,
        offsetof(bh_particle_data, minTimeBin), offsetof(bh_particle_data, TimeBinDynFric), offsetof(bh_particle_data, JumpToMinPot),
        offsetof(bh_particle_data, DFAccel), offsetof(bh_particle_data, DF_SurroundingVel), offsetof(bh_particle_data, DragAccel),
        offsetof(bh_particle_data, MinPotPos), offsetof(bh_particle_data, MinPotVel) };
    printf(" \"bh_dyn_view\": {\"elsize\": %zu, \"off_mintimebin\": %zu, \"off_timebindynfric\": %zu, \"off_jumptominpot\": %zu, \"off_dfaccel\": %zu, "
           "\"off_df_surroundingvel\": %zu, \"off_dragaccel\": %zu, \"off_minpotpos\": %zu, \"off_minpotvel\": %zu},\n",
           dv.elsize, dv.off_mintimebin, dv.off_timebindynfric, dv.off_jumptominpot, dv.off_dfaccel, dv.off_df_surroundingvel, dv.off_dragaccel,
           dv.off_minpotpos, dv.off_minpotvel);
    printf(" \"bh_particle_data\": {\"ReverseLink\": %zu, \"Mass\": %zu, \"Mdot\": %zu, \"Density\": %zu, \"DivVel\": %zu, \"VDisp\": %zu, "
           "\"SwallowID\": %zu, \"MinPot\": %zu, \"CountProgs\": %zu}}\n",
           offsetof(bh_particle_data, ReverseLink), offsetof(bh_particle_data, Mass), offsetof(bh_particle_data, Mdot), offsetof(bh_particle_data, Density),
           offsetof(bh_particle_data, DivVel), offsetof(bh_particle_data, VDisp), offsetof(bh_particle_data, SwallowID), offsetof(bh_particle_data, MinPot),
           offsetof(bh_particle_data, CountProgs));
    return 0;
}
