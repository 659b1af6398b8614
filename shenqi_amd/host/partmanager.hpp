/* partmanager.hpp — host-side mirror of the reference particle store layouts so that a view of
 * PartManager->Base can be handed to the C-ABI without copies.
 * Mirrors `struct particle_data` (libgadget/partmanager.h:9-71, 160 bytes in the default
 * LOW_PRECISION=double, non-DEBUG build) and `struct sph_particle_data`
 * (libgadget/slotsmanager.h:97-131, 176 bytes). */
#ifndef SHQH_PARTMANAGER_HPP
#define SHQH_PARTMANAGER_HPP
#include <stdint.h>
#include <stddef.h>
#include "../../include/shenqi_hip.h"

typedef double MyFloat;
typedef int64_t inttime_t;
typedef uint64_t MyIDType;

struct particle_data {
    double Pos[3];
    int TopLeaf;
    float Mass;
    int PI;
    struct {
        unsigned int IsGarbage : 1;
        unsigned int Swallowed : 1;
        unsigned int HeIIIionized : 1;
        unsigned int BHHeated : 1;
        unsigned char Generation : 4;
        unsigned char TimeBinHydro;
        unsigned char TimeBinGravity;
        unsigned char Type;
    };
    MyFloat Vel[3];
    MyFloat FullTreeGravAccel[3];
    MyFloat GravPM[3];
    inttime_t Ti_drift;
    MyFloat Hsml;
    MyFloat DtHsml;
    MyIDType ID;
    int64_t GrNr;
    MyFloat Potential;
};
static_assert(sizeof(particle_data) == 160, "particle_data must match the reference layout (160 B)");
static_assert(offsetof(particle_data, Mass) == 28 && offsetof(particle_data, Vel) == 40 &&
              offsetof(particle_data, FullTreeGravAccel) == 64 && offsetof(particle_data, GravPM) == 88 &&
              offsetof(particle_data, Hsml) == 120 && offsetof(particle_data, ID) == 136 &&
              offsetof(particle_data, Potential) == 152, "particle_data field offsets");

#define SHQH_NMETALS 9
struct sph_particle_data {
    /* particle_data_ext base (libgadget/slotsmanager.h:19-32; ID only under -DDEBUG) */
    int ReverseLink;
    MyFloat Density;
    MyFloat EgyWtDensity;
    MyFloat Entropy;
    MyFloat DtEntropy;
    MyFloat MaxSignalVel;
    MyFloat HydroAccel[3];
    MyFloat DhsmlEgyDensityFactor;
    MyFloat DivVel;
    MyFloat CurlVel;
    MyFloat Sfr;
    MyFloat Ne;
    MyFloat VDisp;
    MyFloat DelayTime;
    MyFloat Metallicity;
    float Metals[SHQH_NMETALS];
};
static_assert(sizeof(sph_particle_data) == 176, "sph_particle_data must match the reference layout (176 B)");

struct part_manager_type {
    particle_data *Base;
    int64_t NumPart;
    int64_t MaxPart;
    double CurrentParticleOffset[3];
    double BoxSize;
};

/* ActiveParticles (libgadget/timestep.h:10-38): NULL list means "all particles". */
struct ActiveParticles {
    int64_t MaxActiveParticle;
    int64_t NumActiveParticle;
    int *ActiveParticle;
    int64_t NumActiveGravity;
    int64_t NumActiveHydro;
};

static inline shq_part_view make_part_view(particle_data *base, int64_t numpart)
{
    shq_part_view v;
    v.base = base;
    v.elsize = sizeof(particle_data);
    v.numpart = numpart;
    v.off_pos = offsetof(particle_data, Pos);
    v.off_mass = offsetof(particle_data, Mass);
    v.off_type = offsetof(particle_data, PI) + 4 + 3; /* Type: 4th byte of the bitfield word */
    v.off_flags = offsetof(particle_data, PI) + 4;
    v.off_pi = offsetof(particle_data, PI);
    v.off_vel = offsetof(particle_data, Vel);
    v.off_treeacc = offsetof(particle_data, FullTreeGravAccel);
    v.off_gravpm = offsetof(particle_data, GravPM);
    v.off_potential = offsetof(particle_data, Potential);
    v.off_hsml = offsetof(particle_data, Hsml);
    v.off_dthsml = offsetof(particle_data, DtHsml);
    v.off_timebin_hydro = offsetof(particle_data, PI) + 4 + 1;
    v.off_timebin_gravity = offsetof(particle_data, PI) + 4 + 2;
    return v;
}
#endif
