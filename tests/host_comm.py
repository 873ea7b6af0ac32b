"""Host-staged transports of the CPU multi-process tests (TEST INFRASTRUCTURE -- moved out of
the product package in round 5: the product's data path is RCCL inside libpyrohip,
pyro2_amd.decomp.RcclComm / pyro2_amd.multigrid.slab.RcclRowComm).

`HostStagedComm`: the contract of RcclComm over a torch.distributed process group (gloo) with
the halo rows staged through host memory; also what `bench.py` uses when it is ASKED for the
host-staged debug path (PYRO_BENCH_COMM=host: flagged in the line, never a scaling result).
`HostRowComm`: rows of multigrid level arrays between ranks, the same way."""
import numpy as np


class HostStagedComm:
    """same contract as RcclComm over a torch.distributed process group
    (gloo) with the halo rows staged through host memory.  NOT the product
    data path: used by the CPU test-suite (no RCCL without GPUs) and as the
    loudly reported fallback of bench.py when the RCCL communicator cannot
    be created."""

    # with the emulated backend set_neighbours only selects the launch order of the
    # row-marching kernel (boundary strips first): same results, exercised on CPU
    overlap = True

    def __init__(self, td):
        self.td = td

    def halo_exchange(self, state, lo, hi):
        import torch
        ng, nxl = state.ng, state.nx
        reqs, recvs = [], []
        # same pairing as csrc/comm.hip: low rows -> lo, hi ghosts <- hi,
        # high rows -> hi, lo ghosts <- lo
        if lo >= 0:
            t = torch.from_numpy(state.download_rows(ng, ng).copy())
            reqs.append(self.td.isend(t, lo, tag=1))
        if hi >= 0:
            buf = torch.empty((ng, state.qy, state.nvar), dtype=torch.float64)
            reqs.append(self.td.irecv(buf, hi, tag=1))
            recvs.append((nxl + ng, buf))
        if hi >= 0:
            t = torch.from_numpy(state.download_rows(nxl, ng).copy())
            reqs.append(self.td.isend(t, hi, tag=2))
        if lo >= 0:
            buf = torch.empty((ng, state.qy, state.nvar), dtype=torch.float64)
            reqs.append(self.td.irecv(buf, lo, tag=2))
            recvs.append((0, buf))
        for r in reqs:
            r.wait()
        for row, buf in recvs:
            state.upload_rows(row, buf.numpy())

    def allreduce_min(self, x):
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        self.td.all_reduce(t, op=self.td.ReduceOp.MIN)
        return float(t[0])

    def gather(self, state, dec):
        """same contract as RcclComm.gather: every rank's slab -> rank 0's (qx_global, qy, nvar)"""
        import torch
        ng = state.ng
        mine = state.download()
        if dec.rank != 0:
            self.td.send(torch.from_numpy(np.ascontiguousarray(mine)), 0, tag=7)
            return None
        out = np.empty((dec.nx + 2 * ng, state.qy, state.nvar))
        out[:state.qx] = mine
        for r in range(1, dec.nranks):
            buf = torch.empty((dec.counts[r] + 2 * ng, state.qy, state.nvar), dtype=torch.float64)
            self.td.recv(buf, r, tag=7)
            i0 = sum(dec.counts[:r])
            out[i0 + ng:i0 + buf.shape[0]] = buf.numpy()[ng:]
        return out


class HostRowComm:
    """rows of level arrays between ranks, staged through the host (torch.distributed)"""

    def __init__(self, td, rank, nranks):
        self.td, self.rank, self.nranks = td, rank, nranks

    def _send(self, a, dst, tag):
        import torch
        return self.td.isend(torch.from_numpy(np.ascontiguousarray(a)), dst, tag=tag)

    def _recv(self, shape, src, tag):
        import torch
        buf = torch.empty(shape, dtype=torch.float64)
        return buf, self.td.irecv(buf, src, tag=tag)

    def exchange(self, mg, level, var, r0, r1, h):
        """h halo rows on either side of the slab [r0, r1] of `var` on `level`"""
        lo = self.rank - 1 if self.rank > 0 else -1
        hi = self.rank + 1 if self.rank < self.nranks - 1 else -1
        q = mg._n(level)
        reqs, recvs = [], []
        if lo >= 0:
            reqs.append(self._send(mg.get_rows(level, var, r0, h), lo, 1))
            buf, rq = self._recv((h, q), lo, 2)
            reqs.append(rq)
            recvs.append((r0 - h, buf))
        if hi >= 0:
            reqs.append(self._send(mg.get_rows(level, var, r1 - h + 1, h), hi, 2))
            buf, rq = self._recv((h, q), hi, 1)
            reqs.append(rq)
            recvs.append((r1 + 1, buf))
        for r in reqs:
            r.wait()
        for i0, buf in recvs:
            mg.set_rows(level, var, i0, buf.numpy())

    def gather_rows(self, mg, level, var, rows_of):
        """every rank's slab rows_of(rank) of `var` on `level` -> rank 0's array"""
        q = mg._n(level)
        if self.rank == 0:
            pend = []
            for r in range(1, self.nranks):
                a, b = rows_of(r)
                buf, rq = self._recv((b - a + 1, q), r, 3)
                pend.append((a, buf, rq))
            for a, buf, rq in pend:
                rq.wait()
                mg.set_rows(level, var, a, buf.numpy())
        else:
            a, b = rows_of(self.rank)
            self._send(mg.get_rows(level, var, a, b - a + 1), 0, 3).wait()

    def scatter_rows(self, mg, level, var, rows_of):
        """rows rows_of(rank) (array rows, ghost rows allowed) of rank 0's `var` -> each rank"""
        q = mg._n(level)
        if self.rank == 0:
            reqs = []
            for r in range(1, self.nranks):
                a, b = rows_of(r)
                reqs.append(self._send(mg.get_rows(level, var, a, b - a + 1), r, 4))
            for rq in reqs:
                rq.wait()
        else:
            a, b = rows_of(self.rank)
            buf, rq = self._recv((b - a + 1, q), 0, 4)
            rq.wait()
            mg.set_rows(level, var, a, buf.numpy())


    def allreduce_sum(self, mg, values):
        import torch
        t = torch.tensor(list(values), dtype=torch.float64)
        self.td.all_reduce(t, op=self.td.ReduceOp.SUM)
        return [float(x) for x in t]

    def allgather_rows(self, mg, level, var, rows_of):
        """every rank's slab rows_of(rank) of `var` on `level` -> every rank's array"""
        q = mg._n(level)
        a, b = rows_of(self.rank)
        mine = mg.get_rows(level, var, a, b - a + 1)
        reqs, recvs = [], []
        for r in range(self.nranks):
            if r == self.rank:
                continue
            ra, rb = rows_of(r)
            buf, rq = self._recv((rb - ra + 1, q), r, 5)
            recvs.append((ra, buf, rq))
            reqs.append(self._send(mine, r, 5))
        for ra, buf, rq in recvs:
            rq.wait()
            mg.set_rows(level, var, ra, buf.numpy())
        for rq in reqs:
            rq.wait()
