"""CPU stand-ins (numpy) for the local PM phases, with the device kernels' semantics, so that the
multi-rank orchestration in shenqi_amd/dist.py can be exercised with gloo where there is no GPU.
Test infrastructure only."""
import math

import numpy as np
import torch


class CpuOps:
    def __init__(self, Nmesh, BoxSize, Asmth, G):
        self.N, self.L, self.Asmth, self.G = Nmesh, BoxSize, Asmth, G
        self.device = torch.device("cpu")

    def empty(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype)

    def set_particles(self, posm_all, nlocal):
        self.posm = posm_all.numpy().copy()
        self.nlocal = nlocal

    def set_deposit_scale(self, total_mass):
        self.log2scale = 61 - math.frexp(total_mass if total_mass > 0 else 1.0)[1]

    def _cic(self):
        p = self.posm[: self.nlocal]
        cell = self.L / self.N
        t = p[:, :3] / cell
        ic = np.floor(t).astype(np.int64)
        res = t - ic
        return ic % self.N, res, p[:, 3]

    def deposit(self, plane0, nxl, ghosts=1):
        N = self.N
        nalloc = nxl if nxl == N else nxl + ghosts
        mesh = np.zeros((nalloc, N, N + 2), dtype=np.int64)
        ic, res, m = self._cic()
        scale = 2.0 ** self.log2scale
        for c in range(8):
            off = [(c >> k) & 1 for k in range(3)]
            w = np.ones(len(m))
            for k in range(3):
                w = w * (res[:, k] if off[k] else (1 - res[:, k]))
            lx = (ic[:, 0] + off[0] - plane0) % N
            assert lx.max(initial=0) < nalloc, "particle outside the slab"
            q = np.rint(w * m * scale).astype(np.int64)
            np.add.at(mesh, (lx, (ic[:, 1] + off[1]) % N, (ic[:, 2] + off[2]) % N), q)
        return torch.from_numpy(mesh)

    def to_real(self, mesh_i):
        return mesh_i.to(torch.float64) * (1.0 / 2.0 ** self.log2scale)

    def green(self, spec_t, y0, nyl):
        """potential_transfer, gravpm.cpp:378-444, on [y_l][z'][x]."""
        N = self.N
        a = spec_t.numpy()
        k = np.arange(N)
        kk = np.where(k <= N // 2, k, k - N)
        tmp = kk * np.pi / N
        with np.errstate(invalid="ignore", divide="ignore"):
            s = np.where(np.abs(tmp) < 1e-5, 1.0 - tmp**2 / 6 + tmp**4 / 120, np.sin(tmp) / np.where(tmp == 0, 1, tmp))
        sinc = 1.0 / (s * s)
        ky = kk[y0:y0 + nyl][:, None, None]
        kz = np.arange(N // 2 + 1)[None, :, None]
        kx = kk[None, None, :]
        k2 = (kx**2 + ky**2 + kz**2).astype(np.float64)
        asmth2 = ((2 * np.pi) * self.Asmth / N) ** 2
        f = sinc[y0:y0 + nyl][:, None, None] * sinc[: N // 2 + 1][None, :, None] * sinc[None, None, :]
        with np.errstate(divide="ignore", invalid="ignore"):
            fac = (-self.G / (np.pi * self.L)) * np.exp(-k2 * asmth2) / k2 * f * f
        fac[k2 == 0] = 0.0
        a *= fac

    def readout(self, ext, plane0, nxl, pot_right=3):
        N = self.N
        phi = ext.numpy()
        assert nxl == N or phi.shape[0] == nxl + 2 + pot_right
        xshift = 0 if nxl == N else plane0 - 2
        ic, res, m = self._cic()
        ffac = -(N / self.L)
        c1, c2 = 2.0 / 3.0, 1.0 / 12.0
        g = np.zeros((self.nlocal, 3))
        pot = np.zeros(self.nlocal)

        def at(dx, dy, dz):
            return phi[(ic[:, 0] + dx - xshift) % N, (ic[:, 1] + dy) % N, (ic[:, 2] + dz) % N]

        for c in range(8):
            a, b, e = c & 1, (c >> 1) & 1, (c >> 2) & 1
            w = (res[:, 0] if a else 1 - res[:, 0]) * (res[:, 1] if b else 1 - res[:, 1]) * (res[:, 2] if e else 1 - res[:, 2])
            pot += w * at(a, b, e)
            g[:, 0] += w * (ffac * (c1 * (at(a + 1, b, e) - at(a - 1, b, e)) - c2 * (at(a + 2, b, e) - at(a - 2, b, e))))
            g[:, 1] += w * (ffac * (c1 * (at(a, b + 1, e) - at(a, b - 1, e)) - c2 * (at(a, b + 2, e) - at(a, b - 2, e))))
            g[:, 2] += w * (ffac * (c1 * (at(a, b, e + 1) - at(a, b, e - 1)) - c2 * (at(a, b, e + 2) - at(a, b, e - 2))))
        self.gravpm, self.pmpot = g, pot

    def results(self, nlocal):
        return self.gravpm, self.pmpot
