"""Multi-GPU TreePM: one process per GPU, x-slab domain decomposition, RCCL over xGMI.

How the path shards:
  * particles and the PM mesh are cut into equal x-slabs, one per rank (the reference cuts its
    real-space PM pencils over a 2-D rank grid np0 x np1, petapm.cpp:217-257; here np1 = 1);
  * PM: local CIC deposit (HIP kernel, 64-bit fixed point) -> one ghost plane to the right
    neighbour (integer add: bit-identical for any rank count) -> 2-D r2c over (y, z) per local
    plane -> all-to-all transpose (x-slabs -> y-slabs; the reference's heffte reshape /
    petapm.cpp:1036 alltoallv) -> 1-D FFT along x -> Green's function on the transposed spectrum
    [y][z'][x] (the reference's own Fourier layout) -> inverse 1-D -> all-to-all back -> 2-D c2r
    -> 2+3 potential ghost planes from the neighbours -> CIC readout with the 4-point stencil;
  * tree: instead of exporting queries to remote trees and importing results
    (treewalk2.h:618-812, two alltoallv per walk and a second remote walk), every rank imports
    the neighbours' particles within `halo` of its slab (one alltoallv of 32-byte records),
    the host builds the local tree over local + ghost particles with the global root cell, and
    the walk runs for the local targets only.  Nodes beyond TreeRcut are discarded by the walk,
    so nothing farther than the halo can contribute.
The FFT stages use rocFFT through torch.fft; torch.distributed (backend "nccl" = RCCL, or
"gloo" with host staging for tests) carries every exchange.  The local compute is abstracted as
`ops` so that the orchestration can be exercised on CPU (tests/cpu_ops.py) with gloo.
"""
import ctypes as C
import math
import os

import numpy as np
import torch
import torch.distributed as dist

from . import capi


class _Done:
    def __init__(self, recv):
        self.recv = recv

    def wait(self):
        return self.recv


class _Pending:
    """A started all-to-all: keeps the buffers alive until the communicator's stream is done with them."""

    def __init__(self, works, send, recv, cplx):
        self.works, self.send, self.recv, self.cplx = works, send, recv, cplx

    def wait(self):
        for w in self.works:
            w.wait()           # the current stream waits for the communicator's; the host does not block
        self.send = None
        return torch.view_as_complex(self.recv) if self.cplx else self.recv


class Comm:
    """Thin wrapper over torch.distributed for the exchanges the path needs."""

    def __init__(self, group=None):
        self.group = group
        self.on = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if self.on else 0
        self.size = dist.get_world_size(group) if self.on else 1
        self.backend = dist.get_backend(group) if self.on else "none"
        # SHQ_COMM_FORCE=1: a one-rank group still goes through the collectives (particle / ghost exchange, the two
        # mesh transposes, the scalar all-reduce): rehearses the RCCL calls of an N-GPU run on a one-GPU box.
        self.multi = self.size > 1 or (self.on and os.environ.get("SHQ_COMM_FORCE", "0") == "1")

    def _stage(self, t):
        """gloo cannot move device tensors: stage through the host (tests only)."""
        if self.backend == "gloo" and t.is_cuda:
            return t.cpu(), t.device
        return t, None

    # RCCL moved a single 3.6 GB peer message wrongly (768^3 spectrum of a one-rank rehearsal: PM force off by 10^2 while
    # the same code is exact at 48^3); messages are therefore kept below 1 GiB: larger ones go as K row-chunked rounds.
    MAX_MSG_BYTES = int(os.environ.get("SHQ_COMM_MAX_MSG", str(1 << 30)))   # the tests lower it to reach the chunked rounds

    def all_to_all_rows(self, send, send_counts):
        """Variable all-to-all of the rows of a 2-D+ tensor: rows [sum(c[:d]), sum(c[:d+1])) go to rank d.
        Returns (recv, recv_counts)."""
        if not self.multi:
            return send, list(send_counts)
        cplx = send.is_complex()
        s, dev = self._stage(torch.view_as_real(send.contiguous()) if cplx else send.contiguous())
        row_bytes = (s[0].numel() if s.shape[0] else int(np.prod(s.shape[1:]))) * s.element_size()
        # counts and the largest message any rank sends travel together, so that every rank derives the same K
        cnt = torch.tensor([[c, max(send_counts) * row_bytes] for c in send_counts], dtype=torch.int64)
        rcnt = torch.empty_like(cnt)
        cdev = send.device if self.backend == "nccl" else torch.device("cpu")
        cnt_d, rcnt_d = cnt.to(cdev), rcnt.to(cdev)
        dist.all_to_all_single(rcnt_d, cnt_d, group=self.group)
        rc = rcnt_d.cpu()
        recv_counts = [int(x) for x in rc[:, 0]]
        K = max(1, -(-int(rc[:, 1].max()) // self.MAX_MSG_BYTES)) if self.backend == "nccl" else 1
        r = torch.empty((sum(recv_counts),) + tuple(s.shape[1:]), dtype=s.dtype, device=s.device)
        if K == 1:
            dist.all_to_all_single(r, s, output_split_sizes=recv_counts, input_split_sizes=list(send_counts), group=self.group)
        else:
            soff = np.concatenate([[0], np.cumsum(send_counts)])
            roff = np.concatenate([[0], np.cumsum(recv_counts)])
            for k in range(K):   # chunk k of every peer segment: rows [c k / K, c (k + 1) / K) of its c rows, views, no copies
                ins = [s[soff[d] + send_counts[d] * k // K:soff[d] + send_counts[d] * (k + 1) // K] for d in range(self.size)]
                outs = [r[roff[d] + recv_counts[d] * k // K:roff[d] + recv_counts[d] * (k + 1) // K] for d in range(self.size)]
                dist.all_to_all(outs, ins, group=self.group)
        r = r.to(dev) if dev is not None else r
        return (torch.view_as_complex(r) if cplx else r), recv_counts

    def all_to_all_rows_start(self, send, send_counts, recv_counts):
        """all_to_all_rows with the receive counts known to the caller (the mesh transposes), started without waiting:
        returns a handle whose wait() gives the received rows.  Over RCCL the rounds run on the communicator's stream and
        the caller may queue independent kernels in the meantime; the other backends complete here."""
        if not self.multi or self.backend != "nccl":
            recv, _ = self.all_to_all_rows(send, send_counts)
            return _Done(recv)
        cplx = send.is_complex()
        s = torch.view_as_real(send.contiguous()) if cplx else send.contiguous()
        row_bytes = int(np.prod(s.shape[1:])) * s.element_size()
        # every rank sees the same counts matrix through (send_counts, recv_counts) only if the caller's counts are global
        # knowledge (slab widths): K follows from the largest width, which all ranks know
        K = max(1, -(-max(max(send_counts), max(recv_counts)) * row_bytes // self.MAX_MSG_BYTES))
        r = torch.empty((sum(recv_counts),) + tuple(s.shape[1:]), dtype=s.dtype, device=s.device)
        works = []
        if K == 1:
            works.append(dist.all_to_all_single(r, s, output_split_sizes=list(recv_counts), input_split_sizes=list(send_counts),
                                                group=self.group, async_op=True))
        else:
            soff = np.concatenate([[0], np.cumsum(send_counts)])
            roff = np.concatenate([[0], np.cumsum(recv_counts)])
            for k in range(K):
                ins = [s[soff[d] + send_counts[d] * k // K:soff[d] + send_counts[d] * (k + 1) // K] for d in range(self.size)]
                outs = [r[roff[d] + recv_counts[d] * k // K:roff[d] + recv_counts[d] * (k + 1) // K] for d in range(self.size)]
                works.append(dist.all_to_all(outs, ins, group=self.group, async_op=True))
        return _Pending(works, s, r, cplx)

    def all_to_all_equal(self, send):
        """Equal-split all-to-all along dim 0 (dim 0 must be a multiple of size)."""
        if not self.multi:
            return send
        cplx = send.is_complex()
        s, dev = self._stage(torch.view_as_real(send.contiguous()) if cplx else send.contiguous())
        r = torch.empty_like(s)
        dist.all_to_all_single(r, s, group=self.group)
        r = r.to(dev) if dev is not None else r
        return torch.view_as_complex(r) if cplx else r

    def shift(self, t, direction):
        """Send `t` to rank + direction (periodic), receive the same-shaped tensor from rank - direction."""
        if not self.multi:
            return t.clone()
        dst = (self.rank + direction) % self.size
        counts = [0] * self.size
        counts[dst] = t.shape[0]
        recv, _ = self.all_to_all_rows(t, counts)
        return recv

    def allreduce_sum(self, x):
        if not self.multi:
            return x
        t = torch.tensor([x], dtype=torch.float64)
        if self.backend == "nccl":
            t = t.cuda()
        dist.all_reduce(t, group=self.group)
        return float(t.item())

    def barrier(self):
        if self.size > 1:
            dist.barrier(group=self.group)


class SlabDecomp:
    """x-slabs of the box = runs of mesh planes, one per rank.  `bounds` (P+1 plane indices, 0 ... Nmesh)
    may be unequal: balanced_bounds() picks them so that every rank owns about the same number of
    particles (the reference balances work through its Peano top-leaf assignment, domain.cpp:620).
    `ycuts` (P+1 lengths, the first and last 0) refine a boundary below the plane: the particles of plane bounds[r] with
    y < ycuts[r] belong to rank r - 1 (its walk targets, its deposit and readout) while the MESH plane stays rank r's —
    a plane through the core of a cluster carries several per cent of all the work, more than the share of a rank at 64
    ranks.  The left rank then deposits into two planes of its right neighbour instead of one and reads four of its
    potential planes instead of three (dep_ghosts, pot_right)."""

    MIN_PLANES = 3   # the readout needs 3 planes from the right neighbour and 2 from the left

    def __init__(self, comm, Nmesh, BoxSize, bounds=None, ycuts=None):
        P = comm.size
        if bounds is None:
            if Nmesh % P != 0:
                raise ValueError("Nmesh %d must be divisible by the number of ranks %d (or pass bounds)" % (Nmesh, P))
            bounds = [r * (Nmesh // P) for r in range(P + 1)]
        bounds = [int(b) for b in bounds]
        if len(bounds) != P + 1 or bounds[0] != 0 or bounds[-1] != Nmesh:
            raise ValueError("bounds must run from 0 to Nmesh with one entry per rank + 1")
        ycuts = [0.0] * (P + 1) if ycuts is None else [float(c) for c in ycuts]
        if len(ycuts) != P + 1 or ycuts[0] != 0.0 or ycuts[-1] != 0.0 or min(ycuts) < 0.0 or max(ycuts) > BoxSize:
            raise ValueError("ycuts must hold one length in [0, BoxSize] per boundary, none at the box's own edge")
        self.split = max(ycuts) > 0.0
        self.dep_ghosts, self.pot_right = (2, 4) if self.split else (1, 3)
        self.min_planes = self.MIN_PLANES + (1 if self.split else 0)
        widths = [bounds[r + 1] - bounds[r] for r in range(P)]
        if P > 1 and min(widths) < self.min_planes:
            raise ValueError("every slab needs at least %d mesh planes (got %s)" % (self.min_planes, widths))
        if Nmesh % P != 0:
            raise ValueError("Nmesh %d must be divisible by the number of ranks %d (equal y-slabs of the spectrum)" % (Nmesh, P))
        self.comm, self.N, self.L, self.bounds, self.widths, self.ycuts = comm, Nmesh, BoxSize, bounds, widths, ycuts
        self.cell = BoxSize / Nmesh
        self.nxl = widths[comm.rank]
        self.plane0 = bounds[comm.rank]
        self.x0, self.x1 = self.slab_range(comm.rank)

    def owner_of(self, x, y=None):
        """rank owning positions x (tensor), by the mesh plane of the CIC base cell (floor(x / cell)); with sub-plane cuts
        the y coordinates decide inside a boundary plane."""
        plane = torch.floor(x / self.cell).to(torch.int64) % self.N
        inner = torch.tensor(self.bounds[1:-1], dtype=torch.int64, device=x.device)
        owner = torch.searchsorted(inner, plane, right=True)
        if self.split:
            if y is None:
                raise ValueError("this decomposition cuts planes by y: owner_of needs the y coordinates")
            first = torch.tensor(self.bounds[:-1], dtype=torch.int64, device=x.device)[owner]
            cut = torch.tensor(self.ycuts[:-1], dtype=y.dtype, device=x.device)[owner]
            owner = owner - ((plane == first) & (y < cut)).to(owner.dtype)
        return owner

    def slab_range(self, r):
        """x extent of what rank r owns: its planes, and the boundary plane behind them when part of that is its own too"""
        return self.bounds[r] * self.cell, (self.bounds[r + 1] + (1 if self.ycuts[r + 1] > 0.0 else 0)) * self.cell


def balanced_bounds(comm, Nmesh, BoxSize, x, weights=None, plane_cost=0.0, y=None):
    """Plane boundaries giving every rank about the same share of the work (x: positions held by this rank, any
    distribution).  Work of a plane = the sum of `weights` over its particles (1 each when None: equal counts) +
    `plane_cost` (what a mesh plane costs whoever owns it, in the units of the weights).  One all-reduce of an
    Nmesh-long histogram.
    With y (the same particles' y coordinates) the cut is refined below the plane and (bounds, ycuts) come back for
    SlabDecomp: the plane in which a rank's share ends goes to the right-hand rank as a mesh plane, and its particles
    with y < ycut — a whole number of mesh cells — to the left-hand one, so that the shares meet to within one row of
    cells instead of one plane.  A second all-reduce, of the cut planes' y histograms."""
    P = comm.size
    cell = BoxSize / Nmesh
    plane = (torch.floor(x / cell).to(torch.int64) % Nmesh).cpu()
    w = None if weights is None else torch.as_tensor(weights).to(torch.float64).cpu()

    def allsum(h):
        if P > 1:
            g = h.cuda() if comm.backend == "nccl" else h
            dist.all_reduce(g, group=comm.group)
            h = g.cpu()
        return h

    hist = allsum(torch.bincount(plane, weights=w, minlength=Nmesh).to(torch.float64))
    cum = torch.cumsum(hist + float(plane_cost), 0).numpy()
    total = cum[-1]
    minp = SlabDecomp.MIN_PLANES + (0 if y is None else 1)
    bounds, want = [0], [0.0]
    for r in range(1, P):
        target = total * r / P
        i = int(np.searchsorted(cum, target, side="left"))          # first plane whose cumulated work reaches the target
        if y is None:
            # cut before or after that plane, whichever leaves the left ranks closer to their share (a plane through the
            # cluster's core carries several per cent of the work)
            b = i if (i > 0 and target - cum[i - 1] < cum[min(i, Nmesh - 1)] - target) else i + 1
        else:
            b = min(i, Nmesh - 1)                                     # the share ends inside plane i: cut that plane
        lo, hi = bounds[-1] + minp, Nmesh - minp * (P - r)
        want.append(target - (cum[b - 1] if b > 0 else 0.0) if lo <= b <= hi and y is not None else 0.0)
        bounds.append(min(max(b, lo), hi))
    bounds.append(Nmesh)
    if y is None:
        return bounds
    # the particles of the cut planes by their row of cells in y
    row = (torch.floor(y / cell).to(torch.int64) % Nmesh).cpu()
    yh = torch.zeros((P - 1, Nmesh), dtype=torch.float64)
    for r in range(1, P):
        sel = plane == bounds[r]
        yh[r - 1] = torch.bincount(row[sel], weights=None if w is None else w[sel], minlength=Nmesh).to(torch.float64)
    yh = allsum(yh).numpy() if P > 1 else yh.numpy()
    ycuts = [0.0]
    for r in range(1, P):
        cy = np.cumsum(yh[r - 1])
        k = int(np.searchsorted(cy, want[r], side="left"))            # rows 0 .. k reach the share
        if want[r] <= 0.0:
            k = 0
        elif k < Nmesh and want[r] - (cy[k - 1] if k > 0 else 0.0) >= cy[k] - want[r]:
            k = k + 1                                                 # with row k the left rank comes closer to its share
        ycuts.append(min(k * cell, float(BoxSize)))
    ycuts.append(0.0)
    return bounds, ycuts


# Cost model of one force step on an MI355X, from the one-GPU bench: the walk takes 39.5 ms for 8.5e9 interactions,
# deposit + readout 4.2 ms for 1.7e7 particles, the four (y, z) FFT passes 7.9 ms for 768^3 cells.
COST_MS_PER_INTERACTION = 39.5 / 8.5e9
COST_MS_PER_PARTICLE = 4.2 / 16777216
COST_MS_PER_CELL = 7.9 / 768.0**3


def cost_balanced_bounds(comm, drv, subplane=False):
    """Slab boundaries that even out the measured work instead of the particle count: every local particle weighs its
    interaction count of the last walk (plus its deposit / readout), every mesh plane its share of the (y, z) passes.
    The reference balances its domains the same way, by the work counted in the previous step (domain.cpp:620-700,
    GravCost).  Call after a drv.step(); all ranks get the same list — with subplane the pair (bounds, ycuts)."""
    n = int(drv.allp.shape[0])
    nint = np.zeros(n, dtype=np.int64)
    capi.check(capi.hip.shq_grav_short_download(drv.ctx.h, None, None, capi.ptr(nint), None))
    w = COST_MS_PER_INTERACTION * nint[: drv.nloc].astype(np.float64) + COST_MS_PER_PARTICLE
    return balanced_bounds(comm, drv.N, drv.L, drv.local[:, 0], weights=torch.from_numpy(w),
                           plane_cost=COST_MS_PER_CELL * float(drv.N) ** 2, y=drv.local[:, 1] if subplane else None)


def exchange_to_owner(comm, decomp, posm):
    """Domain exchange: send every particle (rows x, y, z, m) to the rank owning its slab."""
    owner = decomp.owner_of(posm[:, 0], posm[:, 1])
    order = torch.argsort(owner, stable=True)
    counts = torch.bincount(owner, minlength=comm.size).tolist()
    recv, _ = comm.all_to_all_rows(posm[order], counts)
    return recv


def ghost_exchange(comm, decomp, posm, halo):
    """Import every other rank's particles within `halo` (periodic) of this rank's slab.  Returns
    the ghost rows (x, y, z, m) in source-rank order; a particle goes at most once to a rank."""
    if not comm.multi:
        return posm[:0]
    L = decomp.L
    x = posm[:, 0]
    parts, counts = [], [0] * comm.size
    for d in range(comm.size):
        if d == comm.rank:
            continue
        a, b = decomp.slab_range(d)
        span = (b - a) + 2 * halo
        if span >= L:
            sel = posm
        else:
            sel = posm[torch.remainder(x - (a - halo), L) < span]
        parts.append(sel)
        counts[d] = int(sel.shape[0])
    send = torch.cat(parts, dim=0) if parts else posm[:0]
    recv, _ = comm.all_to_all_rows(send, counts)
    return recv


class SlabPM:
    """Distributed PM force for the particles this rank owns."""

    def __init__(self, comm, Nmesh, BoxSize, Asmth, G, ops, bounds=None, ycuts=None):
        self.comm, self.ops = comm, ops
        self.N, self.L, self.Asmth, self.G = Nmesh, BoxSize, Asmth, G
        self.d = SlabDecomp(comm, Nmesh, BoxSize, bounds, ycuts)

    def force(self, hooks=()):
        """Runs one PM step for the particles loaded in `ops`; results stay in ops (gravpm, potential).
        hooks: up to two callables that queue work independent of the PM (pieces of the tree walk); the bespoke pipeline
        calls them right after starting its first and its second mesh transpose, so that work runs while the spectrum
        travels."""
        if getattr(self.ops, "pitch", None) is not None and self.ops.pitch() > 0 and os.environ.get("SHQ_SLAB_TORCH_FFT", "0") != "1":
            return self._force_bespoke(list(hooks))
        self._force_torch()
        for h in hooks:
            h()

    def _force_bespoke(self, hooks=()):
        """The slab pipeline on the library's own FFT passes (csrc/fft3d.hip): ONE buffer [nalloc][N][zp] is the
        int64 deposit mesh, the (y, z) half spectrum and the potential, ghost planes in place; the blocks the
        all-to-all delivers are transformed along x as they are (x slowest), fused with the Green's function.
        Passes over the slab: zero, deposit, Z, Y, pack | x+Green+x | unpack, Y, Z, readout."""
        c, N, nxl, P = self.comm, self.N, self.d.nxl, self.comm.size
        ops = self.ops
        zp = ops.pitch()
        zpc = zp // 2
        multi = c.multi   # SHQ_COMM_FORCE: one rank keeps its periodic geometry but still runs the transposes as collectives
        dg, pr = self.d.dep_ghosts, self.d.pot_right     # deposit ghost planes behind the slab, potential planes read from there
        xoff, nalloc = (0, N) if P == 1 else (2, nxl + 2 + pr)
        buf = ops.mesh_buffer(nalloc, N, zp)                               # int64 [nalloc, N, zp]
        ops.deposit2(buf, self.d.plane0, nxl, xoff, nalloc, dg)
        if P > 1:
            ghost = c.shift(buf[xoff + nxl:xoff + nxl + dg].contiguous(), +1)
            buf[xoff:xoff + dg] += ghost
        own = buf[xoff:xoff + nxl]
        nyl = N // P
        # the pack / unpack around the transposes: fused into the Y pass of the FFT (SHQ_DIST_FUSED_PACK=0: torch permute copies)
        fused = multi and hasattr(ops, "fft_yz_packed") and os.environ.get("SHQ_DIST_FUSED_PACK", "1") != "0"
        if fused:
            send = torch.empty((P * nxl, nyl, zpc), dtype=torch.complex128, device=buf.device)      # rows [dest q][x_l]
            ops.fft_yz_packed(own, nxl, 0, send, P)
        else:
            ops.fft_yz(own, nxl, 0)
        spec = own.view(torch.float64).view(torch.complex128)               # [nxl, N, zpc]
        hooks = list(hooks) + [None, None]
        if multi:
            if not fused:
                send = spec.reshape(nxl, P, nyl, zpc).permute(1, 0, 2, 3).reshape(P * nxl, nyl, zpc)   # rows [dest q][x_l]
            pend = c.all_to_all_rows_start(send, [nxl] * P, self.d.widths)                         # [x (all)][y_l][z']
            del send
            if hooks[0]:
                hooks[0]()
            spec_t = pend.wait().contiguous()
        else:
            if hooks[0]:
                hooks[0]()
            spec_t = spec
        ops.xgreen(spec_t, c.rank * nyl, nyl)
        if multi:
            pend = c.all_to_all_rows_start(spec_t, self.d.widths, [nxl] * P)                       # rows [src q][x_l]
            if hooks[1]:
                hooks[1]()
            recv = pend.wait()                                                                     # rows [src q][x_l]
            if fused:
                ops.fft_yz_packed(own, nxl, 1, recv.contiguous(), P)
            else:
                spec.copy_(recv.reshape(P, nxl, nyl, zpc).permute(1, 0, 2, 3).reshape(nxl, N, zpc))
            del recv, pend
        elif hooks[1]:
            hooks[1]()
        if not fused:
            ops.fft_yz(own, nxl, 1)
        phi = buf.view(torch.float64)
        if P > 1:
            phi[0:2] = c.shift(phi[xoff + nxl - 2:xoff + nxl].contiguous(), +1)    # my last 2 -> right rank's left ghosts
            phi[xoff + nxl:xoff + nxl + pr] = c.shift(phi[xoff:xoff + pr].contiguous(), -1)  # my first 3 (4) -> left rank's right ghosts
        ops.readout2(phi, self.d.plane0, nxl, xoff, nalloc)

    def _force_torch(self):
        """The same pipeline with torch.fft (rocFFT) for mesh sizes without a bespoke transform, and on the
        CPU stand-ins of the gloo tests."""
        c, N, nxl, P = self.comm, self.N, self.d.nxl, self.comm.size
        Nc = N // 2 + 1
        widths = self.d.widths
        # 1. deposit + ghost plane to the right neighbour (integer add)
        dg, pr = self.d.dep_ghosts, self.d.pot_right
        mesh_i = self.ops.deposit(self.d.plane0, nxl, dg)             # int64 [nxl(+dg), N, N+2]
        if P > 1:
            ghost = c.shift(mesh_i[nxl:nxl + dg].contiguous(), +1)
            mesh_i[0:dg] += ghost
        real = self.ops.to_real(mesh_i[:nxl])                          # f64 [nxl, N, N+2]
        # 2. forward: 2-D r2c over (y, z), transpose (x-slabs -> equal y-slabs), 1-D along x
        spec = torch.fft.rfft2(real[..., :N], dim=(1, 2))              # [nxl, N, Nc], unscaled
        nyl = N // P
        send = spec.reshape(nxl, P, nyl, Nc).permute(1, 0, 2, 3).reshape(P * nxl, nyl, Nc)   # rows [dest q][x_l]
        recv, _ = c.all_to_all_rows(send, [nxl] * P)                                         # rows [src p][x_l] = all x
        spec_t = recv.reshape(N, nyl, Nc).permute(1, 2, 0).contiguous()                      # [y_l][z][x]
        spec_t = torch.fft.fft(spec_t, dim=2)
        # 3. Green's function / CIC deconvolution on the transposed spectrum
        self.ops.green(spec_t, c.rank * nyl, nyl)
        # 4. inverse: 1-D along x, transpose back, 2-D c2r
        spec_t = torch.fft.ifft(spec_t, dim=2, norm="forward")
        send = spec_t.permute(2, 0, 1).contiguous()                                          # rows = x planes, in order
        recv, _ = c.all_to_all_rows(send, widths)                                            # rows [src q][x_l]
        spec = recv.reshape(P, nxl, nyl, Nc).permute(1, 0, 2, 3).reshape(nxl, N, Nc)
        phi = torch.fft.irfft2(spec, s=(N, N), dim=(1, 2), norm="forward")                   # [nxl, N, N], unscaled
        # 5. potential ghost planes (2 from the left neighbour, 3 from the right) and readout
        if P > 1:
            ext = self.ops.empty((nxl + 2 + pr, N, N + 2), torch.float64)
            ext[2:2 + nxl, :, :N] = phi
            ext[0:2, :, :N] = c.shift(phi[nxl - 2:nxl].contiguous(), +1)          # my last 2 -> right rank's left ghosts
            ext[2 + nxl:, :, :N] = c.shift(phi[0:pr].contiguous(), -1)            # my first 3 (4) -> left rank's right ghosts
        else:
            ext = self.ops.empty((N, N, N + 2), torch.float64)
            ext[:, :, :N] = phi
        self.ops.readout(ext, self.d.plane0, nxl, pr)


class GpuOps:
    """Local compute phases on the device through the C-ABI (libshenqi_hip.so)."""

    def __init__(self, ctx, Nmesh, BoxSize, Asmth, G, device):
        self.ctx, self.device = ctx, device
        self.pm = capi.PMParams(Nmesh, 0, BoxSize, Asmth, G)
        self.N = Nmesh

    def empty(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype, device=self.device)

    def _shared(self):
        """The context was created on torch's current stream: library kernels and torch operators are ordered by the stream
        itself (and the RCCL rounds join it through their work handles), so no host synchronisation is needed anywhere in a
        step.  With a stream of its own (the tests) every hand-over is a full synchronisation of both streams."""
        if os.environ.get("SHQ_DIST_SYNC", "0") == "1":      # diagnostic: synchronise around every phase anyway
            return False
        s = getattr(self.ctx, "stream", None)      # 0 is the null stream: shq_init then made a stream of its own
        return bool(s) and int(s) == int(torch.cuda.current_stream(self.device).cuda_stream)

    def _before(self):
        if not self._shared():
            torch.cuda.current_stream(self.device).synchronize()

    def _after(self):
        if not self._shared():
            self.ctx.synchronize()

    def set_particles(self, posm_all, nlocal, keep_tree=False):
        """posm_all: device tensor [n, 4] (x, y, z, m), the first nlocal rows are this rank's own.  keep_tree: these are the
        positions the resident tree was built from (nothing moved since)."""
        t = posm_all.contiguous()
        self._before()                  # torch produced t on its stream; the library copies on its own
        capi.check(capi.hip.shq_particles_set_device(self.ctx.h, C.c_void_p(t.data_ptr()), t.shape[0], nlocal, int(keep_tree)))
        self._after()
        self._keep = t

    # ---- phases on the bespoke FFT passes (shq_pm_slab2_*) ----
    def pitch(self):
        return int(capi.hip.shq_pm_slab_pitch(self.N))

    def mesh_buffer(self, nalloc, N, zp):
        key = (nalloc, N, zp)
        if getattr(self, "_mesh_key", None) != key:
            self._mesh = torch.empty((nalloc, N, zp), dtype=torch.int64, device=self.device)
            self._mesh_key = key
        return self._mesh

    def _call(self, fn, *args):
        self._before()
        capi.check(fn(self.ctx.h, *args))
        self._after()

    def deposit2(self, buf, plane0, nxl, xoff, nalloc, ghosts=1):
        self._call(capi.hip.shq_pm_slab2_deposit_ghosts, C.byref(self.pm), plane0, nxl, xoff, nalloc, ghosts, C.c_void_p(buf.data_ptr()))

    def fft_yz(self, planes, nplanes, direction):
        assert planes.is_contiguous()
        self._call(capi.hip.shq_pm_slab2_fft_yz, self.N, C.c_void_p(planes.data_ptr()), nplanes, direction)

    def fft_yz_packed(self, planes, nplanes, direction, packed, nranks):
        """fft_yz with the pack (direction 0) / unpack (direction 1) of the transposes fused into the Y pass: `packed` is the
        all-to-all buffer [nranks * nplanes, N / nranks, zpc] complex"""
        assert planes.is_contiguous() and packed.is_contiguous() and packed.dtype == torch.complex128
        self._call(capi.hip.shq_pm_slab2_fft_yz_packed, self.N, C.c_void_p(planes.data_ptr()), nplanes, direction, C.c_void_p(packed.data_ptr()), nranks)

    def xgreen(self, spec_t, y0, nyl):
        assert spec_t.is_contiguous() and spec_t.dtype == torch.complex128
        self._call(capi.hip.shq_pm_slab2_xgreen, C.byref(self.pm), C.c_void_p(spec_t.data_ptr()), y0, nyl)

    def readout2(self, phi, plane0, nxl, xoff, nalloc):
        assert phi.is_contiguous()
        self._call(capi.hip.shq_pm_slab2_readout, C.byref(self.pm), plane0, nxl, xoff, nalloc, C.c_void_p(phi.data_ptr()))

    def hilbert_sorted(self, posm, L):
        """rows of posm (x, y, z, m) along the Peano-Hilbert curve, ordered on the device (shq_hilbert_order)"""
        posm = posm.contiguous()
        n = int(posm.shape[0])
        if n == 0:
            return posm
        order = torch.empty(n, dtype=torch.int64, device=posm.device)
        self._call(capi.hip.shq_hilbert_order, posm.data_ptr(), n, float(L), order.data_ptr())
        return posm[order].contiguous()

    def set_deposit_scale(self, total_mass):
        e = 61 - math.frexp(total_mass if total_mass > 0 else 1.0)[1]
        capi.check(capi.hip.shq_pm_set_deposit_log2scale(self.ctx.h, e))
        self.log2scale = e

    def deposit(self, plane0, nxl, ghosts=1):
        if ghosts != 1:
            raise NotImplementedError("slabs cut below the plane need a mesh size with a bespoke transform (shq_pm_slab2_*)")
        nalloc = nxl if nxl == self.N else nxl + 1
        mesh = torch.empty((nalloc, self.N, self.N + 2), dtype=torch.int64, device=self.device)
        self._call(capi.hip.shq_pm_slab_deposit, C.byref(self.pm), plane0, nxl, C.c_void_p(mesh.data_ptr()))
        return mesh

    def to_real(self, mesh_i):
        return mesh_i.to(torch.float64) * (1.0 / 2.0 ** self.log2scale)

    def green(self, spec_t, y0, nyl):
        assert spec_t.is_contiguous() and spec_t.dtype == torch.complex128
        self._call(capi.hip.shq_pm_slab_green, C.byref(self.pm), y0, nyl, C.c_void_p(spec_t.data_ptr()))

    def readout(self, ext, plane0, nxl, pot_right=3):
        assert ext.is_contiguous() and pot_right == 3
        self._call(capi.hip.shq_pm_slab_readout, C.byref(self.pm), plane0, nxl, C.c_void_p(ext.data_ptr()))

    def results(self, nlocal):
        g = np.zeros((self._keep.shape[0], 3))
        p = np.zeros(self._keep.shape[0])
        capi.check(capi.hip.shq_pm_download(self.ctx.h, capi.ptr(g), capi.ptr(p)))
        return g[:nlocal], p[:nlocal]


class DistTreePM:
    """One rank of the sharded TreePM force: PM over x-slabs + tree walk over local + ghost particles."""

    def __init__(self, comm, ctx, Nmesh, BoxSize, Asmth, G, device, halo_factor=1.5, bounds=None, ycuts=None):
        import shenqi_amd as sq
        self.sq = sq
        self.comm, self.ctx, self.device = comm, ctx, device
        self.N, self.L, self.Asmth, self.G = Nmesh, BoxSize, Asmth, G
        self.ops = GpuOps(ctx, Nmesh, BoxSize, Asmth, G, device)
        self.pm = SlabPM(comm, Nmesh, BoxSize, Asmth, G, self.ops, bounds, ycuts)
        self.decomp = self.pm.d
        self.halo_factor = halo_factor
        self.tree = None

    def setup(self, posm_local, Rcut):
        """posm_local: device tensor [nloc, 4] of the particles this rank owns (already exchanged to
        their owner).  Orders them along a space-filling curve, imports ghosts, builds the tree on the device."""
        self.local = self.ops.hilbert_sorted(posm_local, self.L)
        self.nloc = int(self.local.shape[0])
        self.halo = self.halo_factor * Rcut
        self.ops.set_deposit_scale(self.comm.allreduce_sum(float(self.local[:, 3].sum().item())))
        self._load_particles()
        self._build_tree()

    def _build_tree(self):
        """local + ghost particles are resident: build their tree (global root cell) on the device"""
        sq = self.sq
        try:
            self.tree = sq.tree_build_device(self.ctx, self.L)
        except sq.ShqError as e:
            if "deeper" not in str(e) and "levels" not in str(e):
                raise       # out of memory, invalid state ...: not something a host build would cure
            # deeper than the device build's 21 levels (more than 8 particles within L / 2^21): host build + upload
            allh = self.allp.cpu().numpy()
            pman = sq.PartManager(allh.shape[0], self.L)
            pman.Base["Pos"] = allh[:, :3]
            pman.Base["Mass"] = allh[:, 3]
            pman.Base["Type"] = 1
            self.pman = pman
            self.tree = sq.force_tree_full(pman)
            tv = self.tree.view()
            capi.check(capi.hip.shq_tree_upload(self.ctx.h, C.byref(tv)))

    def _load_particles(self, keep_tree=False):
        ghosts = ghost_exchange(self.comm, self.decomp, self.local, self.halo)
        self.allp = torch.cat([self.local, ghosts], dim=0).contiguous()
        self.nghost = int(ghosts.shape[0])
        self.ops.set_particles(self.allp, self.nloc, keep_tree)

    def step(self, gp, update_potential=1, walk_mode=0, overlap=None, moved=False):
        """One force evaluation: ghost import, PM, walk for the local targets, OldAcc refresh.
        moved: self.local changed since the tree was built (a drift): the ghosts are imported for the new positions and the
        tree is rebuilt.  Without it the step repeats the evaluation on the positions of setup(): the ghost exchange still
        runs (it is part of a step), the tree is kept.
        overlap (default: whenever the transposes are collectives; SHQ_DIST_OVERLAP=0 turns it off): the walk does not need
        the PM result of its own step (OldAcc is the previous step's), so it is cut in two pieces that are queued behind the
        start of the two mesh transposes: the walk computes while the spectrum travels over xGMI."""
        self._load_particles(keep_tree=not moved)
        if moved:
            self._build_tree()
        if overlap is None:
            overlap = self.comm.multi and os.environ.get("SHQ_DIST_OVERLAP", "1") != "0"
        if overlap and self.nloc >= 512:
            half = (self.nloc // 2) // 256 * 256

            def piece(first, count):
                return lambda: capi.check(capi.hip.shq_grav_short_run_range(self.ctx.h, C.byref(gp), first, count,
                                                                           int(update_potential), walk_mode))
            self.pm.force([piece(0, half), piece(half, self.nloc - half)])
        else:
            self.pm.force()
            capi.check(capi.hip.shq_grav_short_run(self.ctx.h, C.byref(gp), None, 0, int(update_potential), walk_mode))
        capi.check(capi.hip.shq_grav_refresh_oldacc(self.ctx.h, self.G))

    def download(self):
        n = int(self.allp.shape[0])
        acc = np.zeros((n, 3))
        pot = np.zeros(n)
        capi.check(capi.hip.shq_grav_short_download(self.ctx.h, capi.ptr(acc), capi.ptr(pot), None, None))
        gpm, ppot = self.ops.results(self.nloc)
        return acc[: self.nloc], pot[: self.nloc], gpm, ppot


# ---------------------------------------------------------------------------------------------------------------------
# Sharded SPH (density with the Hsml loop, hydro force): the same ghost-import design as the tree.
# The reference exports queries to the ranks whose top-leaves a target's search sphere touches, walks them there and
# imports the partial results (treewalk2.h:480-557 do_hsml_loop with exports, DensityQuery / HydroQuery wire formats,
# densitytree2.hpp:260-344, hydratree2.hpp:151-228), once per Hsml iteration.  Here every rank imports, once per operator,
# the records of the other ranks' gas within reach of its slab and runs the whole operator locally for its own targets:
#   density: a neighbour j matters to target i if r_ij < Hsml_i           -> import within halo = hfac * max local Hsml;
#            the Hsml loop may grow a target's Hsml past the halo: then the import is repeated with a larger one;
#   hydro:   symmetric, r_ij < max(Hsml_i, Hsml_j) (hydratree2.hpp:258-259; the reference propagates hmax for this,
#            run.cpp:493)                                                 -> a particle goes to rank d if it lies within
#            max(its own Hsml, max Hsml of d's targets) of d's slab; its record carries the density results of its owner.
# Records travel whole (particle_data 160 B + sph_particle_data 176 B), so predicted velocities and entropies of ghosts are
# evaluated from the same fields as on their owner.


def ghost_records(comm, decomp, x, reach, recs, reach_of_rank):
    """Import the records (uint8 rows) of other ranks' particles: a particle at x with own reach `reach` goes to rank d if it
    lies within max(reach, reach_of_rank[d]) of d's slab (periodic).  Returns the received rows in source-rank order."""
    if not comm.multi:
        return recs[:0]
    L = decomp.L
    parts, counts = [], [0] * comm.size
    for d in range(comm.size):
        if d == comm.rank:
            continue
        a, b = decomp.slab_range(d)
        halo = torch.clamp(reach, min=float(reach_of_rank[d]))
        span = (b - a) + 2 * halo
        sel = (span >= L) | (torch.remainder(x - (a - halo), L) < span)
        parts.append(recs[sel])
        counts[d] = int(sel.sum())
    send = torch.cat(parts, dim=0) if parts else recs[:0]
    recv, _ = comm.all_to_all_rows(send, counts)
    return recv


class DistSPH:
    """One rank of the sharded SPH operators.  `P` / `SphP`: numpy arrays (capi.PARTICLE_DTYPE / capi.SPH_DTYPE) of the gas this
    rank owns (PI = slot index).  `ops` runs an operator on local + ghost particles for the first `nloc` of them:
    ops.density(P, SphP, nloc, **kw) updates Hsml / DtHsml / the density fields of the targets in place;
    ops.hydro(P, SphP, nloc, **kw) likewise HydroAccel / DtEntropy / MaxSignalVel (GpuSphOps below; the tests use the oracle)."""

    def __init__(self, comm, decomp, ops, hfac=1.3):
        self.comm, self.d, self.ops, self.hfac = comm, decomp, ops, hfac
        self.nghost = 0

    def _allmax(self, v):
        if not self.comm.multi:
            return [float(v)] * max(1, self.comm.size)
        t = torch.zeros(self.comm.size, dtype=torch.float64)
        t[self.comm.rank] = float(v)
        if self.comm.backend == "nccl":
            t = t.cuda()
        dist.all_reduce(t, group=self.comm.group)
        return [float(x) for x in t.cpu()]

    def _import(self, P, SphP, reach, reach_of_rank):
        """local + ghost records: ghosts get consecutive slots after the local ones"""
        nloc = len(P)
        rec = np.concatenate([P.view(np.uint8).reshape(nloc, -1), SphP[P["PI"]].view(np.uint8).reshape(nloc, -1)], axis=1)
        got = ghost_records(self.comm, self.d, torch.from_numpy(np.ascontiguousarray(P["Pos"][:, 0])), torch.from_numpy(reach),
                            torch.from_numpy(rec), reach_of_rank).numpy()
        ng = len(got)
        self.nghost = ng
        psz = P.dtype.itemsize
        # not np.concatenate: it repacks structured dtypes (sph_particle_data's padding would go, 176 -> 168 bytes)
        Pall = np.empty(nloc + ng, dtype=P.dtype)
        Sall = np.empty(nloc + ng, dtype=SphP.dtype)
        Pall[:nloc] = P
        Sall[:nloc] = SphP[P["PI"]]
        Pall[nloc:] = np.ascontiguousarray(got[:, :psz]).view(P.dtype).reshape(ng)
        Sall[nloc:] = np.ascontiguousarray(got[:, psz:]).view(SphP.dtype).reshape(ng)
        Pall["PI"] = np.arange(nloc + ng)
        return Pall, Sall

    def density(self, P, SphP, **kw):
        """density() for the local gas (Hsml loop included).  Returns the number of import rounds it took.
        The repeat decision looks at the largest Hsml the loop TRIED (shq_sph_stats.hsml_max_tried; round 4): an intermediate guess
        that reaches past the halo undercounts NumNgb, which changes the next guess, so a loop whose guesses left the halo is run
        again on a wider one even if it came back inside.  (An operator that does not report it - the CPU stand-in of the tests -
        falls back to the final Hsml: with hfac = 1.3 the first guess, at most 1.26 x the start value, densitytree2.hpp:233-246,
        cannot leave the halo.)  The sharded tests assert Hsml against the undivided result to 1e-9."""
        nloc = len(P)
        rounds = 0
        halo = self.hfac * (float(P["Hsml"].max()) if nloc else 0.0)
        h0 = P["Hsml"].copy()
        while True:
            rounds += 1
            halos = self._allmax(halo)
            Pall, Sall = self._import(P, SphP, np.zeros(nloc), halos)
            Pall["Hsml"][:nloc] = h0
            tried = self.ops.density(Pall, Sall, nloc, **kw)
            # the largest Hsml the loop TRIED where the operator reports it (the device library does), else the one it ended with
            hmax = max(float(Pall["Hsml"][:nloc].max()), float(tried or 0.0)) if nloc else 0.0
            # every rank must agree on repeating: a target whose sphere outgrew the halo may have missed neighbours
            worst = max(h / max(hl, 1e-300) for h, hl in zip(self._allmax(hmax), halos)) if self.comm.multi else 0.0
            if worst <= 1.0 or not self.comm.multi:
                break
            halo = self.hfac * max(hmax, halo)
        for k in ("Hsml", "DtHsml"):
            P[k] = Pall[k][:nloc]
        for k in ("Density", "EgyWtDensity", "DhsmlEgyDensityFactor", "DivVel", "CurlVel"):
            SphP[k][P["PI"]] = Sall[k][:nloc]
        return rounds

    def hydro(self, P, SphP, **kw):
        """hydro_force() for the local gas; needs the density fields of density() above on every rank"""
        nloc = len(P)
        hmax = self._allmax(float(P["Hsml"].max()) if nloc else 0.0)
        Pall, Sall = self._import(P, SphP, P["Hsml"].astype(np.float64), hmax)
        self.ops.hydro(Pall, Sall, nloc, **kw)
        for k in ("HydroAccel", "DtEntropy", "MaxSignalVel"):
            SphP[k][P["PI"]] = Sall[k][:nloc]


def gas_rows_from_records(P, SphP):
    """[n, capi.GAS_NCOL] float64 rows (the layout of shq_gas_set_device, include/shenqi_hip.h) of gas records
    (capi.PARTICLE_DTYPE / capi.SPH_DTYPE, slot = PI)"""
    n = len(P)
    S = SphP[P["PI"]]
    r = np.zeros((n, capi.GAS_NCOL))
    r[:, 0:3], r[:, 3], r[:, 4:7], r[:, 7] = P["Pos"], P["Mass"], P["Vel"], P["Hsml"]
    r[:, 8:11], r[:, 11:14], r[:, 14:17] = P["FullTreeGravAccel"], P["GravPM"], S["HydroAccel"]
    r[:, 17], r[:, 18], r[:, 19] = S["Entropy"], S["DtEntropy"], S["DelayTime"]
    r[:, 20], r[:, 21], r[:, 22], r[:, 23], r[:, 24] = S["Density"], S["EgyWtDensity"], S["DhsmlEgyDensityFactor"], S["DivVel"], S["CurlVel"]
    r[:, 25], r[:, 26] = S["MaxSignalVel"], P["DtHsml"]
    r[:, 27] = P["TimeBinGravity"].astype(np.float64) + 256.0 * P["TimeBinHydro"].astype(np.float64)
    return r


def gas_rows_to_records(rows, P, SphP):
    """the result columns of rows (density and hydro) back into the records"""
    pi = P["PI"]
    P["Hsml"], P["DtHsml"] = rows[:, 7], rows[:, 26]
    for c, k in ((20, "Density"), (21, "EgyWtDensity"), (22, "DhsmlEgyDensityFactor"), (23, "DivVel"), (24, "CurlVel"), (18, "DtEntropy"),
                 (25, "MaxSignalVel")):
        SphP[k][pi] = rows[:, c]
    SphP["HydroAccel"][pi] = rows[:, 14:17]


class DistSPHDevice:
    """One rank of the sharded SPH operators with the gas RESIDENT on the device: `rows` is a float64 device tensor
    [nloc, capi.GAS_NCOL] of the gas this rank owns (layout: shq_gas_set_device).  Ghost rows travel as device tensors through the
    all-to-all (RCCL; gloo stages them through the host inside Comm), local + ghost rows become the context's particle set by one
    scatter kernel, the tree of that set is built on the device (shq_tree_build over the gas), density / hydro run on the first
    nloc rows (shq_density_resident / shq_hydro_resident) and one gather kernel writes the results back into the rows: no host
    numpy and no PCIe copy of particle data inside an operator (the import rule is DistSPH's, see above)."""

    def __init__(self, comm, decomp, ctx, BoxSize, hfac=1.3):
        import shenqi_amd as sq
        self.sq, self.comm, self.d, self.ctx, self.L, self.hfac = sq, comm, decomp, ctx, BoxSize, hfac
        self.nghost = 0
        self.stats = {}

    def _allmax(self, v):
        if not self.comm.multi:
            return [float(v)] * max(1, self.comm.size)
        t = torch.zeros(self.comm.size, dtype=torch.float64)
        t[self.comm.rank] = float(v)
        if self.comm.backend == "nccl":
            t = t.cuda()
        dist.all_reduce(t, group=self.comm.group)
        return [float(x) for x in t.cpu()]

    def _shared(self, device):
        """the context runs on torch's current stream (then the stream orders library kernels and torch operators); with a stream of
        its own (the tests) every hand-over between torch and the library is a synchronisation of the producing stream"""
        s = getattr(self.ctx, "stream", None)
        return bool(s) and int(s) == int(torch.cuda.current_stream(device).cuda_stream)

    def _before(self, device):
        if not self._shared(device):
            torch.cuda.current_stream(device).synchronize()

    def _after(self, device):
        if not self._shared(device):
            self.ctx.synchronize()

    def _load(self, rows, reach, reach_of_rank):
        """import the ghosts, make local + ghost rows the resident set, build their tree"""
        ghosts = ghost_records(self.comm, self.d, rows[:, 0], reach, rows, reach_of_rank)
        allrows = torch.cat([rows, ghosts.to(rows.device)], dim=0).contiguous()
        self.nghost = int(ghosts.shape[0])
        self._before(rows.device)        # torch produced allrows on its stream
        capi.check(capi.hip.shq_gas_set_device(self.ctx.h, allrows.data_ptr(), int(allrows.shape[0]), int(rows.shape[0])), "shq_gas_set_device")
        self.sq.tree_build_device(self.ctx, self.L, mask=self.sq.GASMASK)
        return allrows

    def _results(self, allrows, nloc, which):
        capi.check(capi.hip.shq_gas_get_device(self.ctx.h, allrows.data_ptr(), nloc, which), "shq_gas_get_device")
        self._after(allrows.device)      # the library wrote allrows on its stream

    def density(self, rows, dp):
        """density() for the local gas (Hsml loop included); rows are updated in place.  Returns the number of import rounds."""
        nloc = int(rows.shape[0])
        h0 = rows[:, 7].clone()
        halo = self.hfac * (float(h0.max().item()) if nloc else 0.0)
        rounds = 0
        st = capi.SphStats()
        while True:
            rounds += 1
            halos = self._allmax(halo)
            rows[:, 7] = h0
            allrows = self._load(rows, torch.zeros(nloc, dtype=torch.float64, device=rows.device), halos)
            capi.check(capi.hip.shq_density_resident(self.ctx.h, C.byref(dp), C.byref(st)), "shq_density_resident")
            self._results(allrows, nloc, 1)
            # the halo must cover the largest Hsml the loop TRIED, not the one it ended with: a guess that reached past the imported
            # ghosts undercounted NumNgb and steered the guesses after it (the reference runs every guess against all ranks,
            # treewalk2.h:480-557)
            hmax = max(float(st.hsml_max_tried), float(allrows[:nloc, 7].max().item())) if nloc else 0.0
            worst = max(h / max(hl, 1e-300) for h, hl in zip(self._allmax(hmax), halos)) if self.comm.multi else 0.0
            if worst <= 1.0 or not self.comm.multi:
                break
            halo = self.hfac * max(hmax, halo)
        rows.copy_(allrows[:nloc])
        self.stats["density"] = dict(kernel_ms=float(st.kernel_ms), iterations=int(st.niterations), ninteractions=int(st.ninteractions), nghost=self.nghost)
        return rounds

    def hydro(self, rows, hp):
        """hydro_force() for the local gas; needs the density fields of density() above on every rank"""
        nloc = int(rows.shape[0])
        hmax = self._allmax(float(rows[:, 7].max().item()) if nloc else 0.0)
        allrows = self._load(rows, rows[:, 7].contiguous(), hmax)
        st = capi.SphStats()
        capi.check(capi.hip.shq_hydro_resident(self.ctx.h, C.byref(hp), C.byref(st)), "shq_hydro_resident")
        self._results(allrows, nloc, 2)
        rows.copy_(allrows[:nloc])
        self.stats["hydro"] = dict(kernel_ms=float(st.kernel_ms), ninteractions=int(st.ninteractions), nghost=self.nghost)


class GpuSphOps:
    """DistSPH's operators on the device library, through the host mirror of the reference API (one-shot calls: the records
    cross PCIe once per operator)."""

    def __init__(self, ctx, BoxSize):
        import shenqi_amd as sq
        self.sq, self.ctx, self.L = sq, ctx, BoxSize

    def _pman(self, Pall):
        pman = self.sq.PartManager(len(Pall), self.L)
        pman.Base[:] = Pall
        return pman

    def density(self, Pall, Sall, nloc, DoEgyDensity=1, kick=None, **_):
        sq = self.sq
        pman = self._pman(Pall)
        tree = sq.force_tree_rebuild_mask(pman, sq.GASMASK)
        BhP = np.zeros(2, dtype=sq.BH_SLOT_DTYPE)
        _, st = sq.density(self.ctx, np.arange(nloc, dtype=np.int32), 1, DoEgyDensity, 0, kick, tree, pman, Sall, BhP)
        Pall[:] = pman.Base
        return float(st.hsml_max_tried)     # the largest Hsml any walk of the loop searched with

    def hydro(self, Pall, Sall, nloc, atime=0.1, hubble=0.1, kick=None, **_):
        sq = self.sq
        pman = self._pman(Pall)
        tree = sq.force_tree_rebuild_mask(pman, sq.GASMASK)
        sq.force_tree_update_hmax(tree, pman)
        sq.hydro_force(self.ctx, np.arange(nloc, dtype=np.int32), atime, hubble, None, kick, tree, pman, Sall)


class DistFOF:
    """One rank of the friends-of-friends finder over x-slabs (fof.cpp's multi-task part, re-cut for ghost imports).

    The reference links across tasks by exporting primary particles to the tasks whose top leaves they touch and lowering MinID
    on both sides until nothing changes (fof_label_primary's do-while with the ghost branch of fof_primary_ngbiter, fof.cpp:404-470,
    565-580), then reduces the groups that span tasks onto the task of their MinID particle (fof_reduce_groups).  Here every rank
    imports the particles within `halo` of its slab once, labels local + ghost particles (ops.labels: shq_fof on the device, the
    oracle in the CPU tests), and the ranks then lower labels through the particles they share — the owner's label goes to its
    ghost copies, each rank takes the minimum over its local components — until no label changes anywhere.  The secondary types are
    attached by a second labelling call that carries the final labels of the primaries as their "IDs".  Groups are summed per rank
    over the particles it owns and reduced by MinID; FirstPos is the position of the group's MinID particle on every rank (the
    reference uses whichever member its unstable sort put first on the prime task).

    `P`: numpy records (capi.PARTICLE_DTYPE) of the particles this rank owns, all inside its slab.
    ops.labels(Pall, ids, linkl, primary_mask, secondary_mask) -> MinID per particle of Pall (np.uint64)."""

    def __init__(self, comm, decomp, ops):
        self.comm, self.d, self.ops = comm, decomp, ops
        self.rounds = 0
        self.nghost = 0

    def _allsum(self, v):
        return int(round(self.comm.allreduce_sum(float(v)))) if self.comm.multi else int(v)

    def _selections(self, x, halo):
        """per destination rank: which of my particles it holds as ghosts (the same masks give the order of every later label
        message, so labels travel as bare uint64 rows)"""
        L = self.d.L
        sels = []
        for dd in range(self.comm.size):
            if dd == self.comm.rank or not self.comm.multi:
                sels.append(None)
                continue
            a, b = self.d.slab_range(dd)
            span = (b - a) + 2 * halo
            sels.append(np.ones(len(x), dtype=bool) if span >= L else (np.remainder(x - (a - halo), L) < span))
        return sels

    def _send(self, rows, sels):
        parts, counts = [], [0] * self.comm.size
        for dd, s in enumerate(sels):
            if s is None:
                continue
            parts.append(rows[s])
            counts[dd] = int(s.sum())
        send = np.concatenate(parts, axis=0) if parts else rows[:0]
        recv, _ = self.comm.all_to_all_rows(torch.from_numpy(np.ascontiguousarray(send)), counts)
        return recv.numpy()

    def fof(self, P, linkl, minlength, primary_mask=2, secondary_mask=1 + 16 + 32):
        """Returns (MinID per local particle, groups hosted by this rank as a list of dicts ordered by MinID, GrNr per local
        particle).  A group is hosted by the rank that owns its MinID particle; GrNr is global (by decreasing length, then MinID)."""
        comm, L = self.comm, self.d.L
        nloc = len(P)
        ids = P["ID"].astype(np.uint64)
        # the largest radius the secondary search can reach: the first of 0.4 L 2^k (or 0.5 Hsml 2^k) that is >= 4 linking lengths
        hs = float(P["Hsml"].max()) if nloc else 0.0
        halo = max(8.0 * linkl, hs if self.comm.multi else 0.0)
        if comm.multi:
            t = torch.tensor([halo], dtype=torch.float64)
            t = t.cuda() if comm.backend == "nccl" else t
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=comm.group)
            halo = float(t.item())
        x = np.ascontiguousarray(P["Pos"][:, 0])
        sels = self._selections(x, halo)
        got = self._send(P.view(np.uint8).reshape(nloc, -1), sels) if comm.multi else P.view(np.uint8).reshape(nloc, -1)[:0]
        ng = len(got)
        self.nghost = ng
        Pall = np.empty(nloc + ng, dtype=P.dtype)
        Pall[:nloc] = P
        Pall[nloc:] = np.ascontiguousarray(got).view(P.dtype).reshape(ng)
        idall = Pall["ID"].astype(np.uint64)
        prim = (((1 << Pall["Type"].astype(np.int64)) & primary_mask) != 0) & ((Pall["Flags"] & 3) == 0)
        # 1. local components of the primaries (label = smallest ID inside local + ghost)
        lab = self.ops.labels(Pall, idall, linkl, primary_mask, 0)
        comp_key, comp = np.unique(lab, return_inverse=True)
        # 2. lower the labels through the shared particles until nothing changes anywhere
        self.rounds = 0
        while comm.multi:
            self.rounds += 1
            # my owners' labels -> their ghost copies elsewhere (as int64 bit patterns: gloo / NCCL carry no uint64)
            recv = np.ascontiguousarray(self._send(lab[:nloc].view(np.int64).reshape(-1, 1), sels).reshape(-1)).view(np.uint64)
            new = lab.copy()
            new[nloc:] = np.minimum(new[nloc:], recv)
            low = np.full(len(comp_key), np.iinfo(np.uint64).max, dtype=np.uint64)
            np.minimum.at(low, comp, new)
            new = np.where(prim, low[comp], new)
            changed = int((new != lab).sum())
            lab = new
            if self._allsum(changed) == 0:
                break
        # 3. secondary types: nearest primary; the primaries carry their final labels as IDs, everybody else its own ID
        ids2 = np.where(prim, lab, idall)
        lab2 = self.ops.labels(Pall, ids2, linkl, primary_mask, secondary_mask)
        minid = lab2[:nloc].copy()
        # 4. catalogue: lengths first (groups below minlength go), then the sums of the owned members, reduced by MinID
        key, inv = np.unique(minid, return_inverse=True)
        cnt = np.bincount(inv, minlength=len(key)).astype(np.int64)
        allk, allc = self._gather_pairs(key, cnt)
        tot_key, tinv = np.unique(allk, return_inverse=True)
        tot_len = np.bincount(tinv, weights=allc, minlength=len(tot_key)).astype(np.int64)
        keep = tot_len >= minlength
        kept_key, kept_len = tot_key[keep], tot_len[keep]
        order = np.lexsort((kept_key, -kept_len))
        grnr_of = np.empty(len(kept_key), dtype=np.int64)
        grnr_of[order] = np.arange(1, len(kept_key) + 1)
        pos_in_kept = np.searchsorted(kept_key, minid)
        pos_in_kept = np.minimum(pos_in_kept, max(len(kept_key) - 1, 0))
        ingroup = (kept_key[pos_in_kept] == minid) if len(kept_key) else np.zeros(nloc, dtype=bool)
        part_grnr = np.where(ingroup, grnr_of[pos_in_kept] if len(kept_key) else -1, -1)
        groups = self._catalogue(P, minid, ingroup, pos_in_kept, kept_key, kept_len, grnr_of, ids)
        return minid, groups, part_grnr

    def _gather_pairs(self, key, val):
        """all ranks' (key, value) rows"""
        if not self.comm.multi:
            return key, val
        rows = np.stack([key.astype(np.uint64).view(np.int64), val.astype(np.int64)], axis=1)
        counts = [len(rows)] * self.comm.size
        send = np.concatenate([rows] * self.comm.size, axis=0) if len(rows) else rows
        recv, _ = self.comm.all_to_all_rows(torch.from_numpy(np.ascontiguousarray(send)), counts)
        r = recv.numpy()
        return r[:, 0].copy().view(np.uint64), r[:, 1].copy()

    def _catalogue(self, P, minid, ingroup, gidx, kept_key, kept_len, grnr_of, ids):
        """add_particle_to_group over the owned members of every kept group, fof_reduce_groups by allgather, then
        fof_finish_group_properties on the host rank of each group (fof.cpp:583-705, 903-1040)"""
        comm, L = self.comm, self.d.L
        ngk = len(kept_key)
        # FirstPos: the position (as float, BaseGroup.FirstPos) of the MinID particle, known to its owner, gathered
        fp = np.zeros((ngk, 4))
        mine = np.flatnonzero(ingroup & (ids == minid))
        fp[gidx[mine], :3] = P["Pos"][mine].astype(np.float32).astype(np.float64)
        fp[gidx[mine], 3] = 1.0
        fp = self._sum_rows(fp)
        host = fp[:, 3] > 0                    # exactly one owner per kept group holds the MinID particle
        first = fp[:, :3]
        hosted = np.zeros(ngk, dtype=bool)
        hosted[gidx[mine]] = True
        nf = 1 + 6 + 6 + 3 + 3 + 9 + 3          # Mass, LenType, MassType, CM, Vel, Imom, Jmom
        S = np.zeros((ngk, nf))
        m = np.flatnonzero(ingroup)
        if len(m):
            g = gidx[m]
            mass = P["Mass"][m].astype(np.float64)
            ty = P["Type"][m].astype(np.int64)
            d = P["Pos"][m] - first[g]
            rel = np.where(d > 0.5 * L, d - L, np.where(d < -0.5 * L, d + L, d))
            xyz = rel + first[g]
            vel = P["Vel"][m]
            jm = np.stack([rel[:, 1] * vel[:, 2] - vel[:, 1] * rel[:, 2], rel[:, 2] * vel[:, 0] - vel[:, 2] * rel[:, 0],
                           rel[:, 0] * vel[:, 1] - vel[:, 0] * rel[:, 1]], axis=1)
            np.add.at(S[:, 0], g, mass)
            np.add.at(S, (g, 1 + ty), 1.0)
            np.add.at(S, (g, 7 + ty), mass)
            for k in range(3):
                np.add.at(S[:, 13 + k], g, mass * xyz[:, k])
                np.add.at(S[:, 16 + k], g, mass * vel[:, k])
                np.add.at(S[:, 28 + k], g, mass * jm[:, k])
                for k2 in range(3):
                    np.add.at(S[:, 19 + 3 * k + k2], g, mass * rel[:, k] * rel[:, k2])
        S = self._sum_rows(S)
        groups = []
        for gi in np.flatnonzero(hosted):
            s = S[gi]
            M = s[0]
            G = dict(MinID=int(kept_key[gi]), Length=int(kept_len[gi]), GrNr=int(grnr_of[gi]), LenType=[int(round(v)) for v in s[1:7]],
                     MassType=[float(v) for v in s[7:13]], Mass=float(M), FirstPos=first[gi].astype(np.float32))
            vcm = s[16:19] / M
            cm = s[13:16] / M
            dcm = cm - first[gi]
            rel = np.where(dcm > 0.5 * L, dcm - L, np.where(dcm < -0.5 * L, dcm + L, dcm))
            G["CM"] = np.mod(cm, L)
            G["Vel"] = vcm
            jcm = np.array([rel[1] * vcm[2] - vcm[1] * rel[2], rel[2] * vcm[0] - vcm[2] * rel[0], rel[0] * vcm[1] - vcm[0] * rel[1]])
            G["Jmom"] = s[28:31] - jcm * M
            G["Imom"] = s[19:28].reshape(3, 3) - M * np.outer(rel, rel)
            groups.append(G)
        groups.sort(key=lambda G: G["MinID"])
        return groups

    def _sum_rows(self, a):
        if not self.comm.multi:
            return a
        t = torch.from_numpy(np.ascontiguousarray(a))
        t = t.cuda() if self.comm.backend == "nccl" else t
        dist.all_reduce(t, group=self.comm.group)
        return t.cpu().numpy()


class GpuFofOps:
    """DistFOF's labelling on the device library (shq_fof: tree of the primary types, linking, secondary attachment)."""

    def __init__(self, ctx, BoxSize):
        import shenqi_amd as sq
        self.sq, self.ctx, self.L = sq, ctx, BoxSize

    def labels(self, Pall, ids, linkl, primary_mask, secondary_mask):
        sq = self.sq
        n = len(Pall)
        pman = sq.PartManager(n, self.L)
        pman.Base[:] = Pall
        pv = pman.view()
        capi.check(capi.hip.shq_particles_upload(self.ctx.h, C.byref(pv)))
        sq.dynamics_upload(self.ctx, pman)
        fp = capi.FofParams(self.L, linkl, primary_mask, secondary_mask, 1 << 30, 0)     # no catalogue wanted here
        out = np.zeros(n, dtype=np.uint64)
        idarr = np.ascontiguousarray(ids, dtype=np.uint64)
        ng = C.c_int64()
        capi.check(capi.hip.shq_fof(self.ctx.h, C.byref(fp), capi.ptr(idarr), capi.ptr(out), None, C.byref(ng)))
        return out


# ---------------------------------------------------------------------------------------------------------------------
# Particle exchange between ranks on full records: ExchangePlan::domain_exchange
# (libgadget/exchange.hpp:242-333) with the loops on the device (shq_exchange_plan / _pack / _unpack, shq_slots_gc) and the
# collectives through Comm — RCCL when the group is nccl, so the payloads cross xGMI from device buffer to device buffer.
class DistExchange:
    """One rank of a domain exchange.  parts: device uint8 tensor of MaxPart records (layout.part_elsize bytes each), slots: per type a
    device uint8 tensor of slot records or None; layoutfn(parts, numpart) -> device int32 tensor with one target task per particle
    (ExchangePlan::layoutfunc; under shenqi's decomposition what shq_domain_maintain_topleaf writes), evaluated every round as the
    reference does.  maxlast caps the list entries sent per round (find_iter_space's role); a capped or memory-short round runs the
    garbage collection between pack and receive (exchange_once, exchange.hpp:398-406, with shall_we_compact_slots)."""

    def __init__(self, comm, ctx, layout):
        self.comm, self.ctx, self.L = comm, ctx, layout
        self.rounds = 0

    @staticmethod
    def _entries(arr):
        e = (capi.ExchangeEntry * len(arr))()
        for k, row in enumerate(arr):
            e[k].base = int(row[0])
            for t in range(6):
                e[k].slots[t] = int(row[1 + t])
        return e

    @staticmethod
    def _offsets(c):
        o = np.zeros_like(c)
        o[1:] = np.cumsum(c[:-1], axis=0)
        return o

    def _allsum(self, v):
        """element-wise sum of a small int64 vector over the ranks"""
        if not self.comm.multi:
            return np.asarray(v, dtype=np.int64)
        t = torch.tensor(np.asarray(v, dtype=np.int64))
        if self.comm.backend == "nccl":
            t = t.cuda()
        dist.all_reduce(t, group=self.comm.group)
        return t.cpu().numpy()

    def exchange(self, parts, numpart, slots, slot_size, layoutfn, maxlast=0, maxrounds=10000):
        comm, L, h = self.comm, self.L, self.ctx.h
        esz = int(L.part_elsize)
        maxpart = parts.numel() // esz
        ssz = [int(L.slot_elsize[t]) for t in range(6)]
        cap = [0 if slots[t] is None else slots[t].numel() // ssz[t] for t in range(6)]
        slot_size = [int(x) for x in slot_size]
        dev = parts.device
        ntask = comm.size
        self.rounds = 0
        sp = (C.c_void_p * 6)(*[None if s is None else s.data_ptr() for s in slots])
        while True:
            if self.rounds >= maxrounds:
                raise RuntimeError("DistExchange: no end after %d rounds" % maxrounds)
            target = layoutfn(parts, numpart).to(torch.int32).contiguous()
            tg = (capi.ExchangeEntry * ntask)()
            nex, last = C.c_int64(), C.c_int64()
            torch.cuda.current_stream(dev).synchronize()
            capi.check(capi.hip.shq_exchange_plan(h, C.byref(L), parts.data_ptr(), numpart, target.data_ptr(), comm.rank, ntask, int(maxlast), C.byref(nex),
                                                  C.byref(last), tg))
            togo = np.array([[tg[t].base] + list(tg[t].slots) for t in range(ntask)], dtype=np.int64)
            if int(self._allsum([nex.value])[0]) == 0:          # nobody has anything to send
                break
            self.rounds += 1
            # MPI_Alltoall of the plan (exchange.hpp:217-219)
            if comm.multi:
                toget = comm.all_to_all_equal(torch.from_numpy(togo.copy()).to(dev if comm.backend == "nccl" else "cpu")).cpu().numpy()
            else:
                toget = togo.copy()
            soff, goff = self._offsets(togo), self._offsets(toget)
            nsend, nrecv = togo.sum(axis=0), toget.sum(axis=0)
            # Capacity first, before anything is packed or marked (the pack turns the leavers into garbage, and a rank that raised
            # alone would leave the others blocked in the next collective): what cannot fit even once every leaver's record has been
            # collected fails here, on every rank together, as the reference's endrun ends the whole job (exchange.hpp:286-296).
            short = [int(numpart - int(nsend[0]) + int(nrecv[0]) > maxpart)]
            short += [int(slots[t] is not None and slot_size[t] - int(nsend[1 + t]) + int(nrecv[1 + t]) > cap[t]) for t in range(6)]
            gshort = self._allsum(short)
            if gshort.sum() > 0:
                what = "MaxPart %d" % maxpart if gshort[0] else "the slot array of type %d" % int(np.flatnonzero(gshort[1:])[0])
                raise MemoryError("DistExchange: the arrivals of this round do not fit %s on %d rank(s); nothing was packed" % (what, int(gshort.max())))
            partbuf = torch.empty(max(int(nsend[0]), 1) * esz, dtype=torch.uint8, device=dev)
            slotbuf = [None if slots[t] is None else torch.empty(max(int(nsend[1 + t]), 1) * ssz[t], dtype=torch.uint8, device=dev) for t in range(6)]
            bp = (C.c_void_p * 6)(*[None if b is None else b.data_ptr() for b in slotbuf])
            capi.check(capi.hip.shq_exchange_pack(h, C.byref(L), parts.data_ptr(), sp, maxpart, self._entries(soff), ntask, partbuf.data_ptr(), bp))
            # shall_we_gc on any task (:398-402), the slot types to compact on any task (shall_we_compact_slots, :300-317)
            mine = [int(last.value < nex.value or numpart + int(nrecv[0]) > maxpart)]
            for t in range(6):
                c = 0
                if slots[t] is not None and (slot_size[t] + int(nrecv[1 + t]) > 0.95 * cap[t] or int(nsend[1 + t]) > 0.1 * slot_size[t]):
                    c = 1
                mine.append(c)
            glob = self._allsum(mine)
            if glob[0] > 0:
                n = C.c_int64(numpart)
                sz = (C.c_int64 * 6)(*slot_size)
                compact = (C.c_int * 6)(*[int(glob[1 + t] > 0) for t in range(6)])
                capi.check(capi.hip.shq_slots_gc(h, C.byref(L), parts.data_ptr(), C.byref(n), maxpart, sp, sz, compact))
                numpart, slot_size = int(n.value), [int(x) for x in sz]
            self.ctx.synchronize()
            # second line (garbage that was there before this round may or may not have been collected): still collective
            if int(self._allsum([int(numpart + int(nrecv[0]) > maxpart)])[0]) > 0:
                raise MemoryError("DistExchange: %d + %d particles do not fit MaxPart %d on some rank" % (numpart, int(nrecv[0]), maxpart))
            # the alltoallv of the base records and of every slot type (:409-480): device buffers, over RCCL when the group is nccl
            rows, _ = comm.all_to_all_rows(partbuf[:int(nsend[0]) * esz].view(-1, esz), [int(x) for x in togo[:, 0]])
            parts[numpart * esz:(numpart + int(nrecv[0])) * esz] = rows.reshape(-1)
            for t in range(6):
                if slots[t] is None:
                    continue
                if int(self._allsum([int(slot_size[t] + int(nrecv[1 + t]) > cap[t])])[0]) > 0:
                    raise MemoryError("DistExchange: slot array of type %d is full on some rank" % t)
                rows, _ = comm.all_to_all_rows(slotbuf[t][:int(nsend[1 + t]) * ssz[t]].view(-1, ssz[t]), [int(x) for x in togo[:, 1 + t]])
                slots[t][slot_size[t] * ssz[t]:(slot_size[t] + int(nrecv[1 + t])) * ssz[t]] = rows.reshape(-1)
            torch.cuda.current_stream(dev).synchronize()
            so = (C.c_int64 * 6)(*slot_size)
            capi.check(capi.hip.shq_exchange_unpack(h, C.byref(L), parts.data_ptr(), numpart, so, self._entries(toget), self._entries(goff), ntask))
            numpart += int(nrecv[0])
            for t in range(6):
                if slots[t] is not None:
                    slot_size[t] += int(nrecv[1 + t])
            self.ctx.synchronize()
            if int(self._allsum([int(last.value < nex.value)])[0]) == 0:
                break
        return numpart, slot_size


# ---------------------------------------------------------------------------------------------------------------------
# Winds from new stars across ranks (libgadget/winds.cpp:295-369).  The reference exports every new star's query to the ranks whose
# top leaves its Hsml touches and merges the StarKick queues afterwards; here, as for the SPH operators, every rank imports the gas
# records within reach of its slab, runs the two walks for its own new stars on local + ghost gas, and sends the kick candidates
# that fell on ghosts to the ghosts' owners, who resolve all candidates of their own particles (nearest star, then smaller star
# ID: the same outcome whatever the decomposition) and kick.
class DistWinds:
    """P: numpy PARTICLE_DTYPE array of everything this rank owns (gas PI -> SphP, star PI -> StarP), IDs in P["ID"].
    ops.candidates(Pall, Sall, StarP, newstars, prm, rnd) -> (TotalWeight by star slot, kicks: numpy array of capi.WIND_KICK_DTYPE),
    ops.apply(P, SphP, kicks, prm, rnd) -> number kicked (GpuWindOps below; the CPU tests use the restatement)."""

    def __init__(self, comm, decomp, ops):
        self.comm, self.d, self.ops = comm, decomp, ops
        self.nghost = self.nkicks = 0

    def run(self, P, SphP, StarP, newstars, prm, rnd):
        comm = self.comm
        nloc = len(P)
        newstars = np.asarray(newstars, dtype=np.int32)
        reach = float(P["Hsml"][newstars].max()) if len(newstars) else 0.0
        if comm.multi:
            t = torch.tensor([reach], dtype=torch.float64)
            if comm.backend == "nccl":
                t = t.cuda()
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=comm.group)
            reach = float(t.item())
        gas = np.flatnonzero((P["Type"] == 0) & ((P["Flags"] & 1) == 0))
        # ghost gas: the record carries its owner and its index there in fields the walks do not read
        G = P[gas].copy()
        G["TopLeaf"] = comm.rank
        G["GrNr"] = gas
        rec = np.concatenate([G.view(np.uint8).reshape(len(G), -1), SphP[P["PI"][gas]].view(np.uint8).reshape(len(G), -1)], axis=1)
        got = ghost_records(comm, self.d, torch.from_numpy(np.ascontiguousarray(G["Pos"][:, 0])), torch.zeros(len(G), dtype=torch.float64),
                            torch.from_numpy(rec), [reach] * max(1, comm.size)).numpy()
        ng = self.nghost = len(got)
        psz = P.dtype.itemsize
        Pall = np.empty(nloc + ng, dtype=P.dtype)
        Pall[:nloc] = P
        Pall[nloc:] = np.ascontiguousarray(got[:, :psz]).view(P.dtype).reshape(ng)
        Sall = np.empty(len(SphP) + ng, dtype=SphP.dtype)
        Sall[:len(SphP)] = SphP
        Sall[len(SphP):] = np.ascontiguousarray(got[:, psz:]).view(SphP.dtype).reshape(ng)
        Pall["PI"][nloc:] = len(SphP) + np.arange(ng)
        tw, kicks = self.ops.candidates(Pall, Sall, StarP, newstars, prm, rnd)
        # a candidate goes to the owner of its particle, with the particle's index there
        part = kicks["part_index"].astype(np.int64)
        ghost = part >= nloc
        owner = np.full(len(kicks), comm.rank, dtype=np.int64)
        owner[ghost] = Pall["TopLeaf"][part[ghost]]
        kicks = kicks.copy()
        kicks["part_index"][ghost] = Pall["GrNr"][part[ghost]]
        if comm.multi:
            order = np.argsort(owner, kind="stable")
            counts = np.bincount(owner, minlength=comm.size).tolist()
            rows = np.ascontiguousarray(kicks[order]).view(np.int64).reshape(len(kicks), -1)      # 40-byte records as five int64 words
            recv, _ = comm.all_to_all_rows(torch.from_numpy(rows), counts)
            kicks = np.ascontiguousarray(recv.numpy()).view(kicks.dtype).reshape(-1)
        self.nkicks = len(kicks)
        applied = self.ops.apply(P, SphP, kicks, prm, rnd)
        return tw, applied


class GpuWindOps:
    """DistWinds' two steps on the device library (shq_winds_candidates / shq_winds_apply) through the host mirror's tree builder."""

    def __init__(self, ctx, BoxSize):
        import shenqi_amd as sq
        self.sq, self.ctx, self.L = sq, ctx, BoxSize

    def _pman(self, Pall):
        pman = self.sq.PartManager(len(Pall), self.L)
        pman.Base[:] = Pall
        return pman

    def _params(self, prm):
        p = capi.WindParams()
        for k, _ in capi.WindParams._fields_:
            if k != "pad_":
                setattr(p, k, getattr(prm, k))
        return p

    def candidates(self, Pall, Sall, StarP, newstars, prm, rnd):
        sq = self.sq
        pman = self._pman(Pall)
        tree = sq.force_tree_rebuild_mask(pman, sq.GASMASK)
        ids = np.ascontiguousarray(Pall["ID"])
        new = np.ascontiguousarray(newstars, dtype=np.int32)
        rnd = np.ascontiguousarray(rnd, dtype=np.float64)
        tw = np.zeros(len(StarP))
        pv, tv, sv = pman.view(), tree.view(), capi.sph_view(Sall)
        stv = capi.StarView(StarP.ctypes.data, StarP.dtype.itemsize, len(StarP), StarP.dtype.fields["VDisp"][1])
        cp = self._params(prm)
        nk = C.c_int64()
        capi.check(capi.hip.shq_winds_candidates(self.ctx.h, C.byref(tv), C.byref(pv), C.byref(sv), C.byref(stv), capi.ptr(ids), capi.ptr(new), len(new), C.byref(cp),
                                                 capi.ptr(rnd), len(rnd), capi.ptr(tw), None, 0, C.byref(nk)))
        kicks = np.zeros(max(nk.value, 1), dtype=capi.WIND_KICK_DTYPE)
        capi.check(capi.hip.shq_winds_candidates(self.ctx.h, C.byref(tv), C.byref(pv), C.byref(sv), C.byref(stv), capi.ptr(ids), capi.ptr(new), len(new), C.byref(cp),
                                                 capi.ptr(rnd), len(rnd), capi.ptr(tw), capi.ptr(kicks), len(kicks), C.byref(nk)))
        return tw, kicks[:nk.value]

    def apply(self, P, SphP, kicks, prm, rnd):
        pman = self._pman(P)
        ids = np.ascontiguousarray(P["ID"])
        rnd = np.ascontiguousarray(rnd, dtype=np.float64)
        kicks = np.ascontiguousarray(kicks, dtype=capi.WIND_KICK_DTYPE)
        pv, sv = pman.view(), capi.sph_view(SphP)
        cp = self._params(prm)
        na = C.c_int64()
        capi.check(capi.hip.shq_winds_apply(self.ctx.h, C.byref(pv), C.byref(sv), capi.ptr(ids), capi.ptr(kicks), len(kicks), C.byref(cp), capi.ptr(rnd), len(rnd),
                                            C.byref(na)))
        P["Vel"] = pman.Base["Vel"]
        return na.value
