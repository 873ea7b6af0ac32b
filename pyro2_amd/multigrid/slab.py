"""V-cycle of MG.CellCenterMG2d with the fine levels split into x slabs across the
GPUs of one node and the coarse levels collapsed onto one GPU (north_star: "the
multigrid coarse levels collapse to one GPU").

The reference's multigrid is single-block (pyro/multigrid/MG.py:623-778 walks
whole levels).  Here every rank holds the full level hierarchy of a
`device.DeviceMG` but owns, on the levels above `collapse_n`, only its slab of
rows; what it reads beyond the slab arrives as halo rows from its two x
neighbours:

* smoothing (MG.py:544-621): the smoothers advance K red-black iterations per
  launch by temporal blocking -- K = 10, a whole V-cycle leg, where the row-marching
  kernel (levels >= 2048^2) or the deep-apron band kernel (levels <= 512^2) runs, 5
  in between (`pyrohip_mg_rows_kmax`) -- which needs the 2K rows beyond the slab as
  they were BEFORE the launch: one exchange of 2K rows per launch, after which the
  slab holds, bit for bit, what the whole-level launch computes there;
* residual + restriction (MG.py:529-542, patch.py:640-676): one halo row of v;
  the restriction is local because slabs start on odd rows and hold an even
  number of them; the new coarse right-hand side then needs its own 2K halo rows
  (the smoother's apron cells are relaxed with THEIR right-hand side);
* prolongation (patch.py:678-736, MG.py:745-748) rides on the first smoothing
  launch of the up leg: K + 1 halo rows of the coarse solution;
* solve() (MG.py:623-697): per cycle one more halo row of v, the slab's share of the
  two sums (residual, relative change: `pyrohip_mg_diag_rows`) and one all-reduce of the
  two numbers; the convergence test and the cycle count are those of the single-domain
  loop (the sums differ from it in the order of the additions only);
* levels of `collapse_n`^2 and below (0.5 MB at 256^2): the right-hand side is
  gathered to rank 0, which runs the ordinary single-GPU V-cycle from there down
  (`pyrohip_mg_vcycle`), and the rows of the solution each rank prolongs from are
  scattered back.

The communicator only has to move rows (`exchange`, `gather_rows`,
`scatter_rows`): the CPU tests stage them through the host over a
torch.distributed group (tests/host_comm.py: HostRowComm, gloo); on GPUs the same three moves are RCCL
send / recv pairs on the level arrays (`RcclRowComm`; it needs two GPUs and has
not run yet, tests/test_zz_comm.py::test_rccl_multigrid_slabs_two_ranks skips on
one: DESIGN.md 6).
"""
KMAX = 5          # iterations per launch of the tile smoother (multigrid.hip MGW_KMAX)
KDEEP = 10        # ... of the row-marching / deep-apron band kernels


class RcclRowComm:
    """the same three moves device to device over RCCL (csrc/comm.hip); the context of
    the DeviceMG must carry a communicator (Context.comm_init)"""

    def __init__(self, rank, nranks):
        self.rank, self.nranks = rank, nranks

    def allreduce_sum(self, mg, values):
        return [float(x) for x in mg.ctx.allreduce_sum(values)]

    def allgather_rows(self, mg, level, var, rows_of):
        calls = []
        a, b = rows_of(self.rank)
        for r in range(self.nranks):
            if r == self.rank:
                continue
            ra, rb = rows_of(r)
            calls.append(("pyrohip_mg_send_rows", (level, var, a, b - a + 1, r)))
            calls.append(("pyrohip_mg_recv_rows", (level, var, ra, rb - ra + 1, r)))
        self._grouped(mg, calls)

    def exchange(self, mg, level, var, r0, r1, h):
        lo = self.rank - 1 if self.rank > 0 else -1
        hi = self.rank + 1 if self.rank < self.nranks - 1 else -1
        mg._call("pyrohip_mg_exchange_rows", int(level), int(var), int(r0), int(r1), int(h), lo, hi)

    def _grouped(self, mg, calls):
        from .._lib import check, lib
        check(lib().pyrohip_comm_group(1))
        try:
            for fn, args in calls:
                mg._call(fn, *args)
        finally:
            check(lib().pyrohip_comm_group(0))

    def gather_rows(self, mg, level, var, rows_of):
        if self.rank == 0:
            calls = [("pyrohip_mg_recv_rows", (level, var, rows_of(r)[0],
                                               rows_of(r)[1] - rows_of(r)[0] + 1, r))
                     for r in range(1, self.nranks)]
        else:
            a, b = rows_of(self.rank)
            calls = [("pyrohip_mg_send_rows", (level, var, a, b - a + 1, 0))]
        self._grouped(mg, calls)

    def scatter_rows(self, mg, level, var, rows_of):
        if self.rank == 0:
            calls = [("pyrohip_mg_send_rows", (level, var, rows_of(r)[0],
                                               rows_of(r)[1] - rows_of(r)[0] + 1, r))
                     for r in range(1, self.nranks)]
        else:
            a, b = rows_of(self.rank)
            calls = [("pyrohip_mg_recv_rows", (level, var, a, b - a + 1, 0))]
        self._grouped(mg, calls)


class SlabMG:
    """one rank's part of the decomposed V-cycle on a `device.DeviceMG`"""

    def __init__(self, mg, comm, rank, nranks, collapse_n=256, nsmooth=10):
        self.mg, self.comm, self.rank, self.R = mg, comm, rank, nranks
        self.nsmooth = nsmooth
        self.Lf = mg.nlevels - 1
        if collapse_n < 64:
            raise ValueError("levels of 64^2 and below live in single-workgroup kernels")
        self.Lc = max(l for l in range(mg.nlevels) if 2 ** (l + 1) <= collapse_n)
        if self.Lc >= self.Lf:
            raise ValueError("nothing to decompose: the finest level is below the collapse size")
        # iterations one launch does per level (the halo is twice that deep)
        self.kcap = {l: max(1, min(mg.rows_kmax(l), KDEEP)) for l in range(self.Lc + 1, self.Lf + 1)}
        self.kcap[self.Lc] = KDEEP            # (the collapsed level: depth of the scattered rows)
        per = 2 ** (self.Lc + 2) // nranks
        need = 2 * max(self.kcap[self.Lc + 1], KMAX)
        if per * nranks != 2 ** (self.Lc + 2) or per % 2 or per < need:
            raise ValueError("the coarsest decomposed level needs an even number of at least "
                             f"{need} rows per rank")

    def rows(self, level, rank=None):
        """interior rows (1-based, inclusive) of a rank's slab on a level"""
        rank = self.rank if rank is None else rank
        per = 2 ** (level + 1) // self.R
        return 1 + rank * per, (rank + 1) * per

    def _k_first(self, level):
        """iterations of the first smoothing launch on a level"""
        return min(self.nsmooth, self.kcap[level])

    def _smooth(self, level, v_is_zero=False, prolong=False):
        r0, r1 = self.rows(level)
        left, first = self.nsmooth, True
        while left > 0:
            k = min(left, self.kcap[level])
            if not (first and v_is_zero):       # a zero solution needs no halo
                self.comm.exchange(self.mg, level, 0, r0, r1, 2 * k)
            self.mg.smooth_rows(level, k, r0, r1, prolong=prolong and first)
            left -= k
            first = False

    def _cycle(self, level):
        mg = self.mg
        r0, r1 = self.rows(level)
        self._smooth(level, v_is_zero=level < self.Lf)                      # MG.py:722
        self.comm.exchange(mg, level, 0, r0, r1, 1)
        mg.residual_restrict_rows(level, *self.rows(level - 1))             # :724-732
        if level - 1 > self.Lc:
            # the smoother recomputes its 2K apron rows, right-hand side included: the
            # neighbours' rows of the new coarse right-hand side, once per visit
            self.comm.exchange(mg, level - 1, 1, *self.rows(level - 1), 2 * self.kcap[level - 1])
            self._cycle(level - 1)                                          # :735
            c0, c1 = self.rows(level - 1)
            self.comm.exchange(mg, level - 1, 0, c0, c1, self._k_first(level) + 1)
        else:                                                               # collapse
            lc = level - 1
            self.comm.gather_rows(mg, lc, 1, lambda r: self.rows(lc, r))
            if self.rank == 0:
                mg.vcycle(lc)
            n = 2 ** (lc + 1)
            h = self._k_first(level) + 1
            self.comm.scatter_rows(mg, lc, 0, lambda r: (max(self.rows(lc, r)[0] - h, 0),
                                                         min(self.rows(lc, r)[1] + h, n + 1)))
        self._smooth(level, prolong=True)                                   # :745-758

    # ---- decomposition every MG.CellCenterMG2d made from here on uses (the callers --
    # diffusion, incompressible -- construct their solver objects themselves) ----
    _default = None

    @classmethod
    def set_decomposition(cls, comm, rank, nranks, collapse_n=256):
        """one process per GPU: `comm` moves rows between the ranks (HostRowComm /
        RcclRowComm); None switches the decomposition off again"""
        cls._default = None if comm is None else (comm, int(rank), int(nranks), int(collapse_n))

    def vcycle(self):
        """one V-cycle from the finest level (MG.py:699-778), coarse solutions zeroed
        first like MG.solve does (MG.py:658-659)"""
        for l in range(self.Lf):
            self.mg.mark_zero(l)
        self._cycle(self.Lf)

    def solve(self, rtol=1.e-11, max_cycles=100):
        """MG.CellCenterMG2d.solve (MG.py:623-697) on the slabs: V-cycles until the residual
        norm relative to the source norm drops below rtol; returns (cycles, residual_error,
        relative_error).  Every rank ends up with the whole finest solution."""
        import math
        mg, Lf = self.mg, self.Lf
        r0, r1 = self.rows(Lf)
        dx2 = mg.dx * mg.dx
        mg.save_old()                                                      # :647
        res = rel = 1.e33
        cycle = 1
        while res > rtol and cycle <= max_cycles:                         # :652
            self.vcycle()
            self.comm.exchange(mg, Lf, 0, r0, r1, 1)
            s, s2 = self.comm.allreduce_sum(mg, mg.diag_rows(r0, r1))     # :670-678
            rel = math.sqrt(dx2 * s)
            rn = math.sqrt(dx2 * s2)
            res = rn / mg.source_norm if mg.source_norm != 0.0 else rn    # :682-685
            cycle += 1
        self.gather_solution()
        return cycle - 1, res, rel

    def gather_solution(self):
        """the slabs of the finest solution to every rank; ghost cells refreshed (:697)"""
        self.comm.allgather_rows(self.mg, self.Lf, 0, lambda r: self.rows(self.Lf, r))
        self.mg.fill_bc(self.Lf, 0)

    def solution_rows(self):
        """this rank's rows of the finest solution, ghost columns included"""
        r0, r1 = self.rows(self.Lf)
        return self.mg.get_rows(self.Lf, 0, r0, r1 - r0 + 1)
