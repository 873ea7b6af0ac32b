"""HIP multigrid (MG.CellCenterMG2d core) vs the reference's golden vectors.

Tolerance: north_star asks 1e-10 rtol for multigrid.  The smoother, residual,
restriction and prolongation keep the reference's operation order and are
built without FMA contraction: bit-identical on the emulated backend and
<= 1e-13 on the GPU.  Norms are sums and differ from NumPy's pairwise
summation at the 1e-15 level.
"""
import numpy as np
import pytest

from conftest import max_rel_err
from pyro2_amd import device

TOL = 1e-13


def _mk(dev, g, k):
    nx, alpha, beta, ns, nb, inhom = g[f"m{k}_meta"]
    bcs = [str(b) for b in g[f"m{k}_bc"]]
    m = device.DeviceMG(dev, int(nx), bcs=bcs, alpha=alpha, beta=beta, nsmooth=int(ns),
                        nsmooth_bottom=int(nb))
    if int(inhom):
        for s in range(4):
            m.set_bcval(s, g[f"m{k}_bcvals"][s])
    return m


def test_mg_operators(dev, golden):
    g = golden("mg_ops")
    tol = 0.0 if dev.kind == "emu" else TOL
    for k in range(int(g["ncases"])):
        m = _mk(dev, g, k)
        L = m.nlevels - 1
        m.set(L, 0, g[f"m{k}_v0"])
        m.set(L, 1, g[f"m{k}_f0"])
        m.smooth(L, 2)
        m.fill_bc(L, 0)   # corners (the reference's last fill_BC sets them)
        assert max_rel_err(m.get(L, 0), g[f"m{k}_v_smooth"]) <= tol, k
        m.residual(L)
        assert max_rel_err(m.get(L, 2)[1:-1, 1:-1], g[f"m{k}_r"][1:-1, 1:-1]) <= tol, k
        np.testing.assert_allclose(m.norm(L, 2), g[f"m{k}_rnorm"], rtol=1e-13)
        m.restrict(L)
        assert max_rel_err(m.get(L - 1, 1)[1:-1, 1:-1], g[f"m{k}_restrict"][1:-1, 1:-1]) <= tol
        m.set(L - 1, 0, g[f"m{k}_cv"])
        m.zero(L, 0)
        m.prolong_add(L)
        assert max_rel_err(m.get(L, 0)[1:-1, 1:-1], g[f"m{k}_prolong"][1:-1, 1:-1]) <= tol

        m = _mk(dev, g, k)
        m.set(L, 0, g[f"m{k}_v0"])
        m.set(L, 1, g[f"m{k}_f1"])
        src = m.init_rhs_norm()
        info = g[f"m{k}_solve_info"]
        np.testing.assert_allclose(src, info[3], rtol=1e-13)
        for lev in range(L):
            m.zero(lev, 0)
        m.vcycle()
        m.fill_bc(L, 0)
        assert max_rel_err(m.get(L, 0), g[f"m{k}_v_vcycle"]) <= tol * 10, k
        nc, res, rel = m.solve(rtol=1e-10, max_cycles=6)
        assert nc == int(info[0])
        assert max_rel_err(m.get(L, 0), g[f"m{k}_v_solve"]) <= tol * 100, k
        np.testing.assert_allclose(res, info[1], rtol=1e-8)
        np.testing.assert_allclose(rel, info[2], rtol=1e-8)


def _poisson(dev, g, nx):
    m = device.DeviceMG(dev, nx)
    L = m.nlevels - 1
    return m, L


@pytest.mark.gpu
def test_mg_reference_regression_poisson_dirichlet(hip, golden):
    """pyro/test.py:138-140 -- mg_test_simple 256^2 vs mg_poisson_dirichlet.h5,
    7 V-cycles, L2 error 1.60408e-06 (mg_convergence.txt:7)"""
    g = golden("mg_poisson_dirichlet_256")
    m = device.DeviceMG(hip, 256)
    L = m.nlevels - 1
    m.zero(L, 0)
    m.set(L, 1, g["rhs"])
    m.init_rhs_norm()
    nc, res, rel = m.solve(rtol=1e-11)
    assert nc == int(g["ncycles"]) == 7
    v = m.get(L, 0)
    assert max_rel_err(v[1:-1, 1:-1], g["gold"]) <= 1e-12
    x = (np.arange(258) - 0.5) / 256
    X, Y = np.meshgrid(x, x, indexing="ij")
    e = (v - (X ** 2 - X ** 4) * (Y ** 4 - Y ** 2))[1:-1, 1:-1]
    assert abs(np.sqrt(np.sum(e ** 2) / 256 ** 2) - 1.60408e-06) < 1e-11


def test_mg_poisson_64_vs_oracle(dev):
    """same problem at 64^2 (small enough for the emulated backend)"""
    from oracle import orc
    nx = 64
    x = (np.arange(nx + 2) - 0.5) / nx
    X, Y = np.meshgrid(x, x, indexing="ij")
    rhs = -2.0 * ((1.0 - 6.0 * X ** 2) * Y ** 2 * (1.0 - Y ** 2) +
                  (1.0 - 6.0 * Y ** 2) * X ** 2 * (1.0 - X ** 2))
    o = orc.MG(nx)
    o.init_rhs(rhs)
    o.solve(rtol=1e-11, max_cycles=3)
    m = device.DeviceMG(dev, nx)
    L = m.nlevels - 1
    m.zero(L, 0)
    m.set(L, 1, rhs)
    m.init_rhs_norm()
    nc, res, rel = m.solve(rtol=1e-11, max_cycles=3)
    assert nc == o.num_cycles == 3
    tol = 0.0 if dev.kind == "emu" else 1e-12
    assert max_rel_err(m.get(L, 0), o.arr(L, 0)) <= tol
    np.testing.assert_allclose(res, o.residual_error, rtol=1e-9)


@pytest.mark.gpu
def test_mg_4096_vcycle_properties(hip):
    """BASELINE config 4 size: residual norm contracts by > 5x per V-cycle and
    the discrete solution converges to the analytic one (2nd order)"""
    nx = 4096
    x = (np.arange(nx + 2) - 0.5) / nx
    X, Y = np.meshgrid(x, x, indexing="ij")
    rhs = -2.0 * ((1.0 - 6.0 * X ** 2) * Y ** 2 * (1.0 - Y ** 2) +
                  (1.0 - 6.0 * Y ** 2) * X ** 2 * (1.0 - X ** 2))
    m = device.DeviceMG(hip, nx)
    L = m.nlevels - 1
    m.zero(L, 0)
    m.set(L, 1, rhs)
    src = m.init_rhs_norm()
    prev = src
    for c in range(4):
        for lev in range(L):
            m.zero(lev, 0)
        m.vcycle()
        m.residual(L)
        rn = m.norm(L, 2)
        assert rn < prev / 5.0, (c, rn, prev)
        prev = rn
    # (round-off floor of the residual at 4096^2 is ~1e-10 of the source norm)
    nc, res, rel = m.solve(rtol=1e-9)
    assert res <= 1e-9 and nc <= 6
    v = m.get(L, 0)
    e = (v - (X ** 2 - X ** 4) * (Y ** 4 - Y ** 2))[1:-1, 1:-1]
    l2 = np.sqrt(np.sum(e ** 2) / nx ** 2)
    assert l2 < 1.60408e-06 / 200.0   # (4096/256)^2 = 256x smaller than at 256^2


@pytest.mark.parametrize("bcs", [("dirichlet",) * 4, ("periodic",) * 4,
                                 ("neumann", "dirichlet", "periodic", "periodic"),
                                 ("periodic", "periodic", "dirichlet", "neumann")])
def test_mg_tile_smoother_multi_tile(dev, bcs):
    """128^2 is 4 x 2 tiles of the LDS tile smoother (incl. wrapped staging on
    periodic sides and the K = 5 + 2 launch split): must equal both the
    one-launch-per-colour kernel and the oracle bit for bit"""
    from oracle import orc
    nx = 128
    rng = np.random.default_rng(11)
    v0 = rng.standard_normal((nx + 2, nx + 2))
    f0 = rng.standard_normal((nx + 2, nx + 2))
    o = orc.MG(nx, bcs=bcs, alpha=0.3, beta=-1.1)
    L = o.nlevels - 1
    o.arr(L, 0)[:, :] = v0
    o.init_rhs(f0)
    o.smooth(L, 7)
    res = {}
    for kind in (0, 1, 11, 13):     # 10 + k: tile smoother, k iterations / launch
        m = device.DeviceMG(dev, nx, bcs=bcs, alpha=0.3, beta=-1.1)
        m.set_smoother(kind)
        m.set(L, 0, v0)
        m.set(L, 1, f0)
        m.smooth(L, 7)
        m.fill_bc(L, 0)
        res[kind] = m.get(L, 0)
    tol = 0.0 if dev.kind == "emu" else TOL
    assert max_rel_err(res[0], o.arr(L, 0)) <= tol
    for kind in (1, 11, 13):
        assert np.array_equal(res[0], res[kind]), kind
    # and a whole V-cycle through the tile smoother
    o.vcycle()
    m.vcycle()
    m.fill_bc(L, 0)
    assert max_rel_err(m.get(L, 0), o.arr(L, 0)) <= tol * 10


def test_mg_variable_coefficient(dev, golden):
    """VarCoeffCCMG2d on the device against the reference (edge coefficients,
    smoother, residual, 5-cycle solve); the reference's own vc goldens are
    missing from the checkout, the vectors come from running it"""
    g = golden("mg_vc")
    tol = 0.0 if dev.kind == "emu" else TOL
    for k in range(int(g["ncases"])):
        nx = int(g[f"v{k}_nx"])
        bcs = [str(b) for b in g[f"v{k}_bc"]]
        m = device.DeviceMG(dev, nx, bcs=bcs, alpha=0.0, beta=0.0, nsmooth=4, nsmooth_bottom=9)
        m.set_coeffs(g[f"v{k}_c"], [str(b) for b in g[f"v{k}_cbc"]])
        L = m.nlevels - 1
        for lev in (L, L - 1, 0):
            n = 2 ** (lev + 1)
            assert max_rel_err(m.get(lev, 3), g[f"v{k}_c_l{lev}"]) <= tol, (k, lev)
            assert max_rel_err(m.get(lev, 4)[1:n + 2, 1:n + 1],
                               g[f"v{k}_ex_l{lev}"][1:n + 2, 1:n + 1]) <= tol
            assert max_rel_err(m.get(lev, 5)[1:n + 1, 1:n + 2],
                               g[f"v{k}_ey_l{lev}"][1:n + 1, 1:n + 2]) <= tol
        m.set(L, 0, g[f"v{k}_v0"])
        m.set(L, 1, g[f"v{k}_f0"])
        m.init_rhs_norm()
        m.smooth(L, 3)
        m.fill_bc(L, 0)
        assert max_rel_err(m.get(L, 0), g[f"v{k}_v_smooth"]) <= tol, k
        m.residual(L)
        assert max_rel_err(m.get(L, 2)[1:-1, 1:-1], g[f"v{k}_r"][1:-1, 1:-1]) <= tol, k
        m.set(L, 0, g[f"v{k}_v0"])
        nc, res, rel = m.solve(rtol=1e-10, max_cycles=5)
        info = g[f"v{k}_info"]
        assert nc == int(info[0])
        assert max_rel_err(m.get(L, 0), g[f"v{k}_v_solve"]) <= tol * 100, k
        np.testing.assert_allclose(res, info[1], rtol=1e-8)


def test_vc_class_surface(dev, tmp_path, monkeypatch):
    """mg_test_vc_dirichlet (multigrid/examples/mg_test_vc_dirichlet.py) through
    the VarCoeffCCMG2d class: converges to the analytic solution"""
    from pyro2_amd import device as devmod
    from pyro2_amd.mesh import boundary as bnd
    from pyro2_amd.mesh import patch
    from pyro2_amd.multigrid import variable_coeff_MG as VC
    monkeypatch.setattr(devmod.Context, "_default", dev)
    nx = 64 if dev.kind == "hip" else 32
    gr = patch.Grid2d(nx, nx, ng=1)
    c = gr.scratch_array()
    c[:, :] = 2.0 + np.cos(2.0 * np.pi * gr.x2d) * np.cos(2.0 * np.pi * gr.y2d)
    bc_c = bnd.BC(xlb="neumann", xrb="neumann", ylb="neumann", yrb="neumann")
    a = VC.VarCoeffCCMG2d(nx, nx, coeffs=c, coeffs_bc=bc_c, verbose=0)
    a.init_zeros()
    rhs = -16.0 * np.pi**2 * (np.cos(2 * np.pi * a.x2d) * np.cos(2 * np.pi * a.y2d) + 1) * \
        np.sin(2 * np.pi * a.x2d) * np.sin(2 * np.pi * a.y2d)
    a.init_RHS(rhs)
    a.solve(rtol=1.e-11)
    v = a.get_solution()
    e = v - np.sin(2.0 * np.pi * a.x2d) * np.sin(2.0 * np.pi * a.y2d)
    assert e.norm() < (3e-3 if nx == 32 else 8e-4) and a.residual_error < 1e-11
    assert a.edge_coeffs[a.nlevels - 1].x.shape == v.shape


@pytest.mark.gpu
def test_mg_4096_vs_reference_samples(hip, golden):
    """BASELINE config 4 at full size against the REFERENCE itself
    (CellCenterMG2d(4096^2), 10 V-cycles; oracle/gen_golden.py mg_4096): a 64x64
    lattice of the solution, its row / column sums and the residual norm,
    north_star tolerance 1e-10 (measured ~1e-14)"""
    g = golden("mg_4096_samples")
    nx = 4096
    x = (np.arange(nx + 2) - 0.5) / nx
    X, Y = np.meshgrid(x, x, indexing="ij")
    rhs = -2.0 * ((1.0 - 6.0 * X ** 2) * Y ** 2 * (1.0 - Y ** 2) +
                  (1.0 - 6.0 * Y ** 2) * X ** 2 * (1.0 - X ** 2))
    m = device.DeviceMG(hip, nx)
    L = m.nlevels - 1
    m.zero(L, 0)
    m.set(L, 1, rhs)
    src = m.init_rhs_norm()
    nc, res, rel = m.solve(rtol=0.0, max_cycles=10)
    v = m.get(L, 0)[1:-1, 1:-1]
    assert abs(src / float(g["source_norm"]) - 1) < 1e-13
    assert nc == int(g["num_cycles"]) == 10
    assert abs(res / float(g["residual_error"]) - 1) < 1e-10
    step = nx // 64
    # ELEMENT-WISE: the solution vanishes toward the Dirichlet boundary (the lattice's first
    # row / column are the cells next to it, |v| ~ 1e-11 there), every sample is held to the
    # relative tolerance of its own magnitude
    ref = g["samples"]
    assert (np.abs(ref) > 0).all()
    assert (np.abs(v[::step, ::step] - ref) / np.abs(ref)).max() <= 1e-10
    assert np.abs(v.sum(axis=1) - g["row_sums"]).max() <= 1e-10 * np.abs(g["row_sums"]).max()
    assert np.abs(v.sum(axis=0) - g["col_sums"]).max() <= 1e-10 * np.abs(g["col_sums"]).max()


def test_mg_general(dev, golden, tmp_path, monkeypatch):
    """GeneralMG2d (alpha phi + div(beta grad phi) + gamma . grad phi = f) on the
    device against the reference: coefficient hierarchy, smoother, residual,
    5-cycle solve -- through the C ABI and through the GeneralMG2d class"""
    from pyro2_amd import device as devmod
    from pyro2_amd.mesh import boundary as bnd
    from pyro2_amd.mesh import patch
    from pyro2_amd.multigrid import general_MG as GM
    monkeypatch.setattr(devmod.Context, "_default", dev)
    g = golden("mg_general")
    tol = 0.0 if dev.kind == "emu" else TOL
    for k in range(int(g["ncases"])):
        pre = f"g{k}_"
        nx = int(g[pre + "nx"])
        bcs = [str(b) for b in g[pre + "bc"]]
        cbcs = [str(b) for b in g[pre + "cbc"]]
        m = device.DeviceMG(dev, nx, bcs=bcs, alpha=0.0, beta=0.0, nsmooth=4, nsmooth_bottom=9)
        m.set_general_coeffs(g[pre + "alpha"], g[pre + "beta"], g[pre + "gamma_x"],
                             g[pre + "gamma_y"], [cbcs] * 4)
        L = m.nlevels - 1
        for lev in (L, L - 1, 0):
            n = 2 ** (lev + 1)
            for var, nm in ((6, "alpha"), (7, "gamma_x"), (8, "gamma_y")):
                assert max_rel_err(m.get(lev, var), g[pre + f"{nm}_l{lev}"]) <= tol, (k, lev, nm)
            assert max_rel_err(m.get(lev, 4)[1:n + 2, 1:n + 1],
                               g[pre + f"ex_l{lev}"][1:n + 2, 1:n + 1]) <= tol
            assert max_rel_err(m.get(lev, 5)[1:n + 1, 1:n + 2],
                               g[pre + f"ey_l{lev}"][1:n + 1, 1:n + 2]) <= tol
        m.set(L, 0, g[pre + "v0"])
        m.set(L, 1, g[pre + "f0"])
        m.init_rhs_norm()
        m.smooth(L, 3)
        m.fill_bc(L, 0)
        assert max_rel_err(m.get(L, 0), g[pre + "v_smooth"]) <= tol, k
        m.residual(L)
        assert max_rel_err(m.get(L, 2)[1:-1, 1:-1], g[pre + "r"][1:-1, 1:-1]) <= tol, k
        m.set(L, 0, g[pre + "v0"])
        nc, res, rel = m.solve(rtol=1e-10, max_cycles=5)
        info = g[pre + "info"]
        assert nc == int(info[0])
        assert max_rel_err(m.get(L, 0), g[pre + "v_solve"]) <= tol * 100, k
        np.testing.assert_allclose(res, info[1], rtol=1e-8)
    # class surface (last case): coeffs as a CellCenterData2d
    gr = patch.Grid2d(nx, nx, ng=1)
    d = patch.CellCenterData2d(gr)
    bc_c = bnd.BC(xlb=cbcs[0], xrb=cbcs[1], ylb=cbcs[2], yrb=cbcs[3])
    for nm in ("alpha", "beta", "gamma_x", "gamma_y"):
        d.register_var(nm, bc_c)
    d.create()
    for nm in ("alpha", "beta", "gamma_x", "gamma_y"):
        d.get_var(nm)[:, :] = g[pre + nm]
    a = GM.GeneralMG2d(nx, nx, xl_BC_type=bcs[0], xr_BC_type=bcs[1], yl_BC_type=bcs[2],
                       yr_BC_type=bcs[3], nsmooth=4, nsmooth_bottom=9, coeffs=d, verbose=0)
    a.init_solution(g[pre + "v0"])
    a.init_RHS(g[pre + "f0"])
    a.max_cycles = 5
    a.solve(rtol=1.e-10)
    assert a.num_cycles == int(g[pre + "info"][0])
    assert max_rel_err(np.asarray(a.get_solution()), g[pre + "v_solve"]) <= tol * 100
    assert max_rel_err(np.asarray(a.beta_edge[L].x)[1:nx + 2, 1:nx + 1],
                       g[pre + f"ex_l{L}"][1:nx + 2, 1:nx + 1]) <= tol
    assert max_rel_err(np.asarray(a.grids[L].get_var("alpha")), g[pre + f"alpha_l{L}"]) <= tol


class _LockstepRows:
    """the three row moves of multigrid/slab.py between `nranks` SlabMG instances that
    run as threads of ONE process on one device: a mailbox and a barrier; library calls
    never overlap (every thread holds `lock` except while it waits)"""

    class Shared:
        def __init__(self, nranks):
            import threading
            self.lock, self.barrier, self.box = threading.Lock(), threading.Barrier(nranks), {}

    def __init__(self, shared, rank, nranks):
        self.s, self.rank, self.nranks = shared, rank, nranks

    def _sync(self):
        self.s.lock.release()
        try:
            self.s.barrier.wait(timeout=120)
        finally:
            self.s.lock.acquire()

    def exchange(self, mg, level, var, r0, r1, h):
        lo = self.rank - 1 if self.rank > 0 else -1
        hi = self.rank + 1 if self.rank < self.nranks - 1 else -1
        self.s.box[(self.rank, "lo")] = mg.get_rows(level, var, r0, h)
        self.s.box[(self.rank, "hi")] = mg.get_rows(level, var, r1 - h + 1, h)
        self._sync()
        if lo >= 0:
            mg.set_rows(level, var, r0 - h, self.s.box[(lo, "hi")])
        if hi >= 0:
            mg.set_rows(level, var, r1 + 1, self.s.box[(hi, "lo")])
        self._sync()

    def gather_rows(self, mg, level, var, rows_of):
        if self.rank:
            a, b = rows_of(self.rank)
            self.s.box[(self.rank, "g")] = mg.get_rows(level, var, a, b - a + 1)
        self._sync()
        if self.rank == 0:
            for r in range(1, self.nranks):
                mg.set_rows(level, var, rows_of(r)[0], self.s.box[(r, "g")])
        self._sync()

    def scatter_rows(self, mg, level, var, rows_of):
        if self.rank == 0:
            for r in range(1, self.nranks):
                a, b = rows_of(r)
                self.s.box[(r, "s")] = mg.get_rows(level, var, a, b - a + 1)
        self._sync()
        if self.rank:
            mg.set_rows(level, var, rows_of(self.rank)[0], self.s.box[(self.rank, "s")])
        self._sync()


    def allreduce_sum(self, mg, values):
        self.s.box[(self.rank, "sum")] = list(values)
        self._sync()
        # the same order of additions on every rank
        tot = [sum(self.s.box[(r, "sum")][k] for r in range(self.nranks)) for k in range(len(values))]
        self._sync()
        return tot

    def allgather_rows(self, mg, level, var, rows_of):
        a, b = rows_of(self.rank)
        self.s.box[(self.rank, "ag")] = mg.get_rows(level, var, a, b - a + 1)
        self._sync()
        for r in range(self.nranks):
            if r != self.rank:
                mg.set_rows(level, var, rows_of(r)[0], self.s.box[(r, "ag")])
        self._sync()


@pytest.mark.parametrize("nranks,nx,collapse,march", [(2, 256, 64, 0), (2, 256, 64, 256), (4, 512, 128, 0),
                                                      (2, 512, 64, 256), (2, 4096, 256, 2048)])
def test_mg_slab_solve_through_class(dev, nranks, nx, collapse, march):
    """MG.CellCenterMG2d.solve() on x slabs (SlabMG.solve behind the class: the slab option
    of the constructor): V-cycles with the fine levels decomposed -- ten iterations per
    launch on the levels of the row-marching kernel (switched on from `march`^2 here, 2048^2
    in production) and of the deep-apron band kernel, five in between -- the residual and
    relative-change sums all-reduced, against the single-domain solve(): same number of
    cycles, the same solution bit for bit on every rank, norms to the order of the additions."""
    import threading
    from pyro2_amd.multigrid import MG
    if dev.kind == "emu" and nx > 256:
        pytest.skip("size for the GPU")
    rtol = 1.e-6 if dev.kind == "emu" else 1.e-9
    tun = dict(march_min=march, march_waves=16) if march and nx <= 512 else None

    def make(slab=None):
        a = MG.CellCenterMG2d(nx, nx, verbose=0, ctx=dev, slab=slab)
        if tun:
            a._dev.set_tuning(**tun)
            if a._slab is not None:
                a._slab.kcap = {l: max(1, min(a._dev.rows_kmax(l), 10)) if l > a._slab.Lc else 10
                                for l in a._slab.kcap}
        a.init_zeros()
        a.init_RHS(-2.0 * ((1 - 6 * a.x2d**2) * a.y2d**2 * (1 - a.y2d**2) +
                           (1 - 6 * a.y2d**2) * a.x2d**2 * (1 - a.x2d**2)))
        return a
    ref = make()
    ref.solve(rtol=rtol)
    want = np.asarray(ref.get_solution())
    assert 1 < ref.num_cycles < 20

    shared = _LockstepRows.Shared(nranks)
    out, errs = {}, []

    def rank_main(rank):
        with shared.lock:
            try:
                a = make(slab=(_LockstepRows(shared, rank, nranks), rank, nranks, collapse))
                assert a._slab is not None
                if march:
                    assert a._slab.kcap[a.nlevels - 1] == 10
                a.solve(rtol=rtol)
                out[rank] = (a.num_cycles, a.residual_error, a.relative_error, np.asarray(a.get_solution()))
            except BaseException as e:          # noqa: BLE001 - reported below
                errs.append((rank, e))
                shared.barrier.abort()
    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(nranks)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for r in range(nranks):
        nc, res, rel, v = out[r]
        assert nc == ref.num_cycles, r
        assert abs(res / ref.residual_error - 1) < 1e-10 and abs(rel / ref.relative_error - 1) < 1e-10, r
        assert np.array_equal(v[1:-1, 1:-1], want[1:-1, 1:-1]), r


@pytest.mark.parametrize("nranks,nx,collapse", [(2, 256, 64), (4, 512, 128), (2, 1024, 256)])
def test_mg_slab_vcycle_bit_identical(dev, nranks, nx, collapse):
    """the V-cycle with its fine levels in x slabs and the coarse ones collapsed onto
    slab 0 (multigrid/slab.py: row-window launches of the tile smoother and of the
    residual+restriction, halo rows of v, of the coarse right-hand side and of the
    coarse solution) is, bit for bit, the single-domain V-cycle.  The slabs are separate
    DeviceMG hierarchies on one device, the rows move through the host; between
    processes: tests/test_decomp_gloo.py (gloo) and tests/test_zz_comm.py (RCCL)."""
    import threading
    from pyro2_amd.multigrid.slab import SlabMG
    if dev.kind == "emu" and nx > 512:
        pytest.skip("size for the GPU")
    x = (np.arange(nx + 2) - 0.5) / nx
    X, Y = np.meshgrid(x, x, indexing="ij")
    rhs = -2.0 * ((1 - 6 * X**2) * Y**2 * (1 - Y**2) + (1 - 6 * Y**2) * X**2 * (1 - X**2))
    v0 = np.zeros((nx + 2, nx + 2))
    v0[1:-1, 1:-1] = 0.01 * np.random.default_rng(3).standard_normal((nx, nx))
    ref = device.DeviceMG(dev, nx)
    L = ref.nlevels - 1
    ref.set(L, 0, v0)
    ref.set(L, 1, rhs)
    for _ in range(2):
        for l in range(L):
            ref.mark_zero(l)
        ref.vcycle(L)
    want = ref.get(L, 0)

    shared = _LockstepRows.Shared(nranks)
    out, errs = {}, []

    def rank_main(rank):
        with shared.lock:
            try:
                m = device.DeviceMG(dev, nx)
                m.set(L, 0, v0)
                m.set(L, 1, rhs)
                sm = SlabMG(m, _LockstepRows(shared, rank, nranks), rank, nranks, collapse_n=collapse)
                for _ in range(2):
                    sm.vcycle()
                out[rank] = (sm.rows(L), sm.solution_rows())
            except BaseException as e:          # noqa: BLE001 - reported below
                errs.append((rank, e))
                shared.barrier.abort()
    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(nranks)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for r in range(nranks):
        (a, b), v = out[r]
        assert np.array_equal(v[:, 1:-1], want[a:b + 1, 1:-1]), r
    assert np.abs(want[1:-1, 1:-1]).max() > 0


@pytest.mark.parametrize("bcs", [("dirichlet",) * 4, ("periodic",) * 4,
                                 ("neumann", "dirichlet", "periodic", "periodic"),
                                 ("periodic", "periodic", "dirichlet", "neumann")])
@pytest.mark.parametrize("coef", [(0.3, -1.1), (0.0, -1.0)])
def test_mg_march_smoother(dev, bcs, coef, monkeypatch):
    """the row-marching smoother of the large levels (csrc/mg_march.hip), switched on
    for a 256^2 level (column strips, row chunks with aprons, wrapped loads on the
    periodic sides; general and power-of-two coefficients): ten and twenty iterations
    (one and two launches) and a whole V-cycle (prolongation fused into the
    load, zero start on the way down) must equal the one-launch-per-colour kernels bit
    for bit"""
    nx = 256
    alpha, beta = coef
    rng = np.random.default_rng(12)
    v0 = rng.standard_normal((nx + 2, nx + 2))
    f0 = rng.standard_normal((nx + 2, nx + 2))
    res = {}
    for march in (0, 1):
        m = device.DeviceMG(dev, nx, bcs=bcs, alpha=alpha, beta=beta,
                            tuning=dict(march_min=256 if march else 0,
                                        march_waves=24))            # 8 row chunks of 32
        L = m.nlevels - 1
        m.set_smoother(0 if not march else 1)
        out = []
        for nsm in (10, 20):
            m.set(L, 0, v0)
            m.set(L, 1, f0)
            m.smooth(L, nsm)
            m.fill_bc(L, 0)
            out.append(m.get(L, 0))
        m.set(L, 0, v0)
        m.set(L, 1, f0)
        m.vcycle()
        m.fill_bc(L, 0)
        out.append(m.get(L, 0))
        res[march] = out
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)


def test_mg_solve_lazy_residual_and_old_copy(dev, monkeypatch):
    """inside solve() the residual arrays are not stored by the fused residual passes and
    the copy of the previous solution is a buffer rotation (csrc/multigrid.hip:
    pyrohip_mg::lazy_r, capture_old): cycles, both error norms, the solution and the
    finest level's residual array read afterwards equal those of the storing variant"""
    nx = 128
    x = (np.arange(nx + 2) - 0.5) / nx
    X, Y = np.meshgrid(x, x, indexing="ij")
    rhs = -2.0 * ((1.0 - 6.0 * X ** 2) * Y ** 2 * (1.0 - Y ** 2) +
                  (1.0 - 6.0 * Y ** 2) * X ** 2 * (1.0 - X ** 2))
    out = []
    for eager in (True, False):
        m = device.DeviceMG(dev, nx, tuning=dict(lazy_residual=0 if eager else 1))
        L = m.nlevels - 1
        m.zero(L, 0)
        m.set(L, 1, rhs)
        m.init_rhs_norm()
        r1 = m.solve(rtol=1e-11, max_cycles=4)
        r2 = m.solve(rtol=0.0, max_cycles=2)           # a second solve starts from the first one's state
        out.append((r1, r2, m.get(L, 0), m.get(L, 2)))
    assert out[0][0] == out[1][0] and out[0][1] == out[1][1]
    assert np.array_equal(out[0][2], out[1][2])
    assert np.array_equal(out[0][3][1:-1, 1:-1], out[1][3][1:-1, 1:-1])


@pytest.mark.parametrize("nx", [128, 256])
def test_mg_solve_speculative_cycle(dev, nx, monkeypatch):
    """solve() puts the next V-cycle on the stream before the norms of the current one have
    reached the host and undoes it when it was one too many (csrc/multigrid.hip,
    pyrohip_mg_solve): with the speculation forced on for every cycle, the cycle count, both
    norms, the solution and the residual array equal those of the loop that waits -- for a
    solve that converges (the last cycle is undone), one that runs into max_cycles, and a
    second solve on the same object"""
    if dev.kind == "emu" and nx > 128:
        pytest.skip("the larger grid on the GPU only (asynchronous there)")
    import subprocess, sys, pickle
    code = r"""
import sys, pickle, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from conftest import _get_ctx
from pyro2_amd import device
dev = _get_ctx(%r)
nx = %d
x = (np.arange(nx + 2) - 0.5) / nx
X, Y = np.meshgrid(x, x, indexing="ij")
rhs = -2.0 * ((1.0 - 6.0 * X ** 2) * Y ** 2 * (1.0 - Y ** 2) + (1.0 - 6.0 * Y ** 2) * X ** 2 * (1.0 - X ** 2))
m = device.DeviceMG(dev, nx, tuning=dict(speculate=%d))
L = m.nlevels - 1
m.zero(L, 0); m.set(L, 1, rhs); m.init_rhs_norm()
out = [m.solve(rtol=1e-7, max_cycles=30), m.get(L, 0)]
out += [m.solve(rtol=0.0, max_cycles=3), m.get(L, 0), m.get(L, 2)[1:-1, 1:-1]]
out += [m.solve(rtol=1e-11, max_cycles=30), m.get(L, 0)]
pickle.dump(out, sys.stdout.buffer)
"""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for mode in ("0", "2", "1"):
        p = subprocess.run([sys.executable, "-c", code % (root, os.path.join(root, "tests"), dev.kind, nx,
                                                          int(mode))],
                           env=dict(os.environ), capture_output=True, timeout=900)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        res[mode] = pickle.loads(p.stdout)
    assert 1 < res["0"][0][0] < 30            # converged: the forced speculation had a cycle to undo
    for mode in ("2", "1"):
        for a, b in zip(res["0"], res[mode]):
            if isinstance(a, tuple):
                assert a == b, mode
            else:
                assert np.array_equal(a, b), mode


@pytest.mark.parametrize("nx,bcs,coef", [
    (256, ("dirichlet",) * 4, (0.0, -1.0)),
    (256, ("periodic",) * 4, (0.3, -1.1)),
    (256, ("neumann", "dirichlet", "periodic", "periodic"), (0.0, -1.0)),
    (256, ("periodic", "periodic", "dirichlet", "neumann"), (0.3, -1.1)),
    (512, ("dirichlet", "neumann", "neumann", "dirichlet"), (0.0, -1.0)),
])
def test_mg_march_tails(dev, nx, bcs, coef):
    """inside solve() the marching smoother's launches carry what follows them (csrc/mg_march.hip,
    MGMarch::tail): the down leg's residual + restriction (every marching level; 512^2: two of
    them, the coarser one from a zero start) and, on the finest level's last launch, the two
    sums of MG.py:670-686.  Cycle count, solution and the residual array read afterwards equal
    those of the separate passes bit for bit; the two norms to rounding (other summation order)"""
    alpha, beta = coef
    rng = np.random.default_rng(5)
    f0 = rng.standard_normal((nx + 2, nx + 2))
    if alpha == 0.0 and "dirichlet" not in bcs:
        f0[1:-1, 1:-1] -= f0[1:-1, 1:-1].mean()
    out = {}
    for tail in (0, 1):
        m = device.DeviceMG(dev, nx, bcs=bcs, alpha=alpha, beta=beta,
                            tuning=dict(march_min=256, march_waves=24 if nx == 256 else 48,
                                        march_tail=tail, speculate=0))
        L = m.nlevels - 1
        m.zero(L, 0)
        m.set(L, 1, f0)
        m.init_rhs_norm()
        r1 = m.solve(rtol=1e-30, max_cycles=2)
        v1 = m.get(L, 0)
        r2 = m.solve(rtol=1e-5, max_cycles=6)
        ncyc = r1[0] + r2[0]
        assert m.tail_counts() == ((ncyc * (2 if nx == 512 else 1), ncyc) if tail else (0, 0))
        out[tail] = (r1, v1, r2, m.get(L, 0), m.get(L, 2)[1:-1, 1:-1])
    a, b = out[0], out[1]
    for k in (0, 2):
        assert a[k][0] == b[k][0]
        assert a[k][1] == pytest.approx(b[k][1], rel=1e-14)
        assert a[k][2] == pytest.approx(b[k][2], rel=1e-14)
    assert a[0][1] > 0 and a[0][2] > 0
    for k in (1, 3, 4):
        assert np.array_equal(a[k], b[k])


@pytest.mark.parametrize("bcs", [("dirichlet",) * 4, ("periodic",) * 4, ("neumann",) * 4,
                                 ("neumann", "dirichlet", "periodic", "periodic"),
                                 ("periodic", "periodic", "dirichlet", "neumann")])
@pytest.mark.parametrize("coef", [(0.3, -1.1), (0.0, -1.0)])
@pytest.mark.parametrize("nx", [4, 8, 16, 32, 64, 128])
def test_mg_coarse_wave_levels(dev, nx, bcs, coef):
    """the levels up to 32^2 of the coarse V-cycle kernel run on one wavefront with a level's
    cells in registers (csrc/multigrid.hip: mgw_vcycle) -- mirror and periodic sides, general
    and power-of-two coefficients, every depth of the recursion (a 4^2 finest level has only
    the 2^2 level below it, 128^2 enters the kernel at 64^2): v, f and r of every level after
    a V-cycle equal those of the workgroup's LDS sweeps bit for bit"""
    alpha, beta = coef
    rng = np.random.default_rng(nx)
    v0 = rng.standard_normal((nx + 2, nx + 2))
    f0 = rng.standard_normal((nx + 2, nx + 2))
    if alpha == 0.0 and "dirichlet" not in bcs:
        f0[1:-1, 1:-1] -= f0[1:-1, 1:-1].mean()
    out = []
    # (the workgroup's LDS sweeps on every level; + the 64^2 level's sweeps in registers, rows
    # through LDS: mgc_sweeps_band64; + the wavefront-resident levels)
    for wave, band in ((0, 0), (0, 1), (1, 1)):
        m = device.DeviceMG(dev, nx, bcs=bcs, alpha=alpha, beta=beta,
                            tuning=dict(coarse_wave=wave, coarse_band64=band))
        L = m.nlevels - 1
        m.set(L, 0, v0)
        m.set(L, 1, f0)
        m.vcycle()
        m.vcycle()
        out.append([m.get(l, var) for l in range(m.nlevels) for var in (0, 1, 2)])
    for other in out[1:]:
        for k, (a, b) in enumerate(zip(out[0], other)):
            l, var = divmod(k, 3)
            if var == 2:
                a, b = a[1:-1, 1:-1], b[1:-1, 1:-1]
            if var == 2 and l in (0, m.nlevels - 1):
                continue                      # r of the bottom / finest level: nobody's output
            assert np.array_equal(a, b), (l, var)


@pytest.mark.gpu
@pytest.mark.parametrize("nx,bcs,coef", [
    (2048, ("periodic",) * 4, (0.0, -1.0)),
    (2048, ("neumann", "dirichlet", "periodic", "periodic"), (0.3, -1.1)),
    (8192, ("dirichlet",) * 4, (0.0, -1.0)),
])
def test_mg_march_tails_production_sizes(hip, nx, bcs, coef):
    """test_mg_march_tails on the levels the marching smoother runs on in production (2048^2:
    one marching level; 8192^2: three, 2048 wavefronts each): inside solve() with and without
    the work that rides on the marching launches -- cycle count, solution and residual array
    bit for bit, norms to rounding"""
    alpha, beta = coef
    rng = np.random.default_rng(nx)
    f0 = rng.standard_normal((nx + 2, nx + 2))
    if alpha == 0.0 and "dirichlet" not in bcs:
        f0[1:-1, 1:-1] -= f0[1:-1, 1:-1].mean()
    out = {}
    for tail in (0, 1):
        m = device.DeviceMG(hip, nx, bcs=bcs, alpha=alpha, beta=beta, tuning=dict(march_tail=tail, speculate=0))
        L = m.nlevels - 1
        m.zero(L, 0)
        m.set(L, 1, f0)
        m.init_rhs_norm()
        r = m.solve(rtol=1e-30, max_cycles=2)
        nmarch = {2048: 1, 8192: 3}[nx]
        assert m.tail_counts() == ((2 * nmarch, 2) if tail else (0, 0))
        out[tail] = (r, m.get(L, 0), m.get(L, 2)[1:-1, 1:-1])
        del m
    assert out[0][0][0] == out[1][0][0]
    assert out[0][0][1] == pytest.approx(out[1][0][1], rel=1e-13)
    assert out[0][0][2] == pytest.approx(out[1][0][2], rel=1e-13)
    assert np.array_equal(out[0][1], out[1][1])
    assert np.array_equal(out[0][2], out[1][2])


@pytest.mark.parametrize("nx", [256, 2048])
def test_mg_solve_speculation_with_tails(dev, nx):
    """the cycle launched ahead and undone, with the sums riding on the marching launches:
    relative_error is then taken once after the loop from the solution and the one before it,
    which a fourth finest-level buffer keeps through the speculative cycle (csrc/multigrid.hip:
    pyrohip_mg::older).  Speculation forced on every cycle against the loop that waits: cycle
    count, both norms and the solution identical -- for a solve that converges (the last cycle is
    undone), one that runs into max_cycles, and a second solve on the same object"""
    if (dev.kind == "emu") != (nx == 256):
        pytest.skip("256^2 on the emulator (marching kernel switched on there), 2048^2 on the GPU")
    x = (np.arange(nx + 2) - 0.5) / nx
    X, Y = np.meshgrid(x, x, indexing="ij")
    rhs = -2.0 * ((1.0 - 6.0 * X ** 2) * Y ** 2 * (1.0 - Y ** 2) + (1.0 - 6.0 * Y ** 2) * X ** 2 * (1.0 - X ** 2))
    res = {}
    for spec in (0, 2, 1):
        tun = dict(speculate=spec)
        if nx == 256:
            tun.update(march_min=256, march_waves=24)
        m = device.DeviceMG(dev, nx, tuning=tun)
        L = m.nlevels - 1
        m.zero(L, 0)
        m.set(L, 1, rhs)
        m.init_rhs_norm()
        out = [m.solve(rtol=1e-7, max_cycles=30), m.get(L, 0)]
        out += [m.solve(rtol=0.0, max_cycles=3), m.get(L, 0)]
        out += [m.solve(rtol=1e-10, max_cycles=30), m.get(L, 0), m.get(L, 2)[1:-1, 1:-1]]
        assert m.tail_counts()[1] > 0
        res[spec] = out
    assert 1 < res[0][0][0] < 30
    for spec in (2, 1):
        for a, b in zip(res[0], res[spec]):
            if isinstance(a, tuple):
                assert a == b, spec
            else:
                assert np.array_equal(a, b), spec
