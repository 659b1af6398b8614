"""Generate the committed golden fixtures under tests/golden/ from the (pinned) CPU oracle.

The reference cannot be built in this image, so these vectors are outputs of the oracle
after it passed the reference's golden values and accuracy gates (oracle/README.md); they freeze it
against drift and let the GPU tests run against data files.  Run in the build container:
    python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import shenqi_amd as sq  # noqa: E402
import orc  # noqa: E402
import common as cm  # noqa: E402

out = os.path.join(ROOT, "tests", "golden")
os.makedirs(out, exist_ok=True)

# gravity: 12^3 clustered particles (tests/test_gravity.cpp:316-341 pattern), Nmesh 36
n = 12**3
pos = cm.random_positions(orc.boost_mt19937_uniform(0, 3 * n), n)
mass = np.ones(n, dtype=np.float32)
cm.reference_treepar(ErrTolForceAcc=0.005, MaxBHOpeningAngle=0.9, Rcut=6.0, TreeUseBH=0)
sq.gravshort_set_softenings(cm.BOX / np.cbrt(n))
gp = sq.make_grav_params(cm.BOX, 1.5, 36, cm.G, cm.RHO0)
nodes, first, _ = orc.tree_build(pos, mass, cm.BOX)
gpm, ppot, rho, phi = orc.pm_force(pos, mass, 36, cm.BOX, 1.5, cm.G, want_mesh=True)
cm.reference_treepar(ErrTolForceAcc=0.005, MaxBHOpeningAngle=0.9, Rcut=6.0, TreeUseBH=1)
gp_bh = sq.make_grav_params(cm.BOX, 1.5, 36, cm.G, cm.RHO0)
a1, p1, n1 = orc.grav_walk(nodes, first, pos, mass, np.zeros(n), gp_bh)
oldacc = np.linalg.norm(a1 * cm.G + gpm, axis=1) / cm.G
a2, p2, n2 = orc.grav_walk(nodes, first, pos, mass, oldacc, gp)
np.savez_compressed(os.path.join(out, "treepm_12cube.npz"), pos=pos, gravpm=gpm, pm_potential=ppot, oldacc=oldacc,
                    acc_bh=a1, nint_bh=n1, acc_rel=a2, pot_rel=p2, nint_rel=n2, nmesh=36, box=cm.BOX, G=cm.G,
                    errtol=0.005, note="raw walk outputs before postprocess (x G not applied)")

# SPH: 10^3 random gas, cubic kernel, pressure-entropy
n = 10**3
u = orc.boost_mt19937_uniform(7, 3 * n)
pos = cm.BOX * u.reshape(n, 3)
pman, SphP, BhP = cm.make_gas(pos, np.full(n, cm.BOX / 10))
rng = np.random.default_rng(11)
pman.Base["Vel"] = rng.normal(size=(n, 3))
SphP["Entropy"] = rng.uniform(0.5, 2.0, size=n)
dp = cm.density_params(DoEgyDensity=1)
st = orc.SphState(pman.Base, SphP, BhP)
nodes, first, father = orc.tree_build(pos, pman.Base["Mass"], cm.BOX)
orc.set_init_hsml(nodes, first, father, st, cm.BOX, dp.DesNumNgb)
hsml0 = st.hsml.copy()
nodes, first, father = orc.tree_build(pos, pman.Base["Mass"], cm.BOX)
rc, evp, _, niter, nint = orc.density(nodes, first, father, st, dp)
assert rc == 0
orc.update_hmax(nodes, first, st)
hp = cm.hydro_params()
nint_h = orc.hydro(nodes, first, st, hp, evp)
np.savez_compressed(os.path.join(out, "sph_10cube.npz"), pos=pos, vel=pman.Base["Vel"], entropy=SphP["Entropy"], hsml0=hsml0,
                    hsml=st.hsml, density=st.density, egywtdensity=st.egywtdensity, dhsml=st.dhsmlegydensityfactor,
                    divvel=st.divvel, curlvel=st.curlvel, entvarpred=evp, niter=niter, nint_density=nint,
                    hydroaccel=st.hydroaccel, dtentropy=st.dtentropy, maxsignalvel=st.maxsignalvel, nint_hydro=nint_h, box=cm.BOX)
print("wrote", os.listdir(out))

# distributed walk: export table of a fabricated three-task domain over the 12^3 gravity fixture, and the stellar
# density of 200 stars in a 12^3 gas box (oracle/toptree.cpp, oracle/sph.cpp)
g = np.load(os.path.join(out, "treepm_12cube.npz"))
pos = g["pos"]
n = len(pos)
pman = cm.make_partmanager(pos)
dom = sq.force_tree_full(pman)
tl = cm.make_domain(dom, ntask=3, me=1, depth=2)
cm.reference_treepar(ErrTolForceAcc=0.005, MaxBHOpeningAngle=0.9, Rcut=6.0, TreeUseBH=0)
sq.gravshort_set_softenings(cm.BOX / np.cbrt(n))
gp = sq.make_grav_params(cm.BOX, 1.5, 36, cm.G, cm.RHO0)
counts, table = orc.grav_toptree(dom.Nodes_base, dom.firstnode, dom.lastnode, tl, pos, g["oldacc"], gp)
hs = 0.08 * cm.BOX * (0.5 + np.random.default_rng(3).random(n))
ncounts, ntable = orc.ngb_toptree(dom.Nodes_base, dom.firstnode, dom.lastnode, tl, pos, hs, 0, cm.BOX)
np.savez_compressed(os.path.join(out, "toptree_12cube.npz"), topleaves=tl, counts=counts, table=table, hsml=hs, ngb_counts=ncounts,
                    ngb_table=ntable, ntask=3, me=1, depth=2)

sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_oracle_cpu as toc  # noqa: E402
pman, SphP, ng, nstar = toc._stars_in_gas(n1=12, nstar=200, seed=2)
tree = sq.force_tree_rebuild_mask(pman, sq.GASMASK)
st = orc.SphState(pman.Base, SphP)
des = 4.0 / 3 * np.pi * 2.0**3
queue = np.arange(ng, ng + nstar, dtype=np.int32)
hs0 = pman.Base["Hsml"].copy()
rc, vol, niter, nint = orc.stellar_density(tree.Nodes_base, tree.firstnode, st, queue, cm.BOX, des, 2.0, 1, 1)
assert rc == 0
np.savez_compressed(os.path.join(out, "stellar_12cube.npz"), pos=pman.Base["Pos"], mass=pman.Base["Mass"], hsml0=hs0, density=SphP["Density"],
                    ng=ng, nstar=nstar, hsml=st.hsml[ng:], starvol=vol[ng:], niter=niter, des=des)
print("wrote", os.listdir(out))
