"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (pyro2) itself.

TEST INFRASTRUCTURE ONLY.  Run in the build container (the reference does not
exist on the GPU box):

    cd /tmp && MPLBACKEND=Agg \
      PYTHONPATH=/root/repo/oracle/shim:/root/reference \
      /opt/conda/bin/python3.9 /root/repo/oracle/gen_golden.py

The shim replaces numba.njit by the identity (numba is not importable here;
the decorated functions are plain Python/NumPy) and stubs
the setuptools_scm-generated pyro/_version.py.  Nothing is copied from the
reference: its functions are *called* and their inputs/outputs are stored.
"""
import os
import sys
import tempfile

import h5py
import numpy as np

os.chdir(tempfile.mkdtemp())  # Pyro writes inputs.auto into cwd

import pyro.compressible as comp                       # noqa: E402
import pyro.compressible.interface as ifc              # noqa: E402
import pyro.compressible.unsplit_fluxes as flx         # noqa: E402
import pyro.mesh.array_indexer as ai                   # noqa: E402
import pyro.mesh.boundary as bnd                       # noqa: E402
import pyro.multigrid.MG as MG                         # noqa: E402
from pyro.advection import advective_fluxes            # noqa: E402
from pyro.advection.interface import linear_interface  # noqa: E402
from pyro.compressible import riemann                  # noqa: E402
from pyro.mesh import patch, reconstruction            # noqa: E402
from pyro.pyro_sim import Pyro                         # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
REF = "/root/reference/pyro"
os.makedirs(OUT, exist_ok=True)


def save(name, **kw):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **kw)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


# --------------------------------------------------------------------------
# ghost fill (a1): every BC type, ng = 4 and ng = 1, non-square grid
# --------------------------------------------------------------------------
def gen_fill_bc():
    rng = np.random.default_rng(1)
    out = {}
    cases = [("outflow",) * 4, ("periodic",) * 4,
             ("reflect-even", "reflect-odd", "reflect-odd", "reflect-even"),
             ("outflow", "reflect-even", "periodic", "periodic"),
             ("reflect-odd", "outflow", "reflect-even", "outflow")]
    for ng in (4, 1):
        for k, bcs in enumerate(cases):
            g = patch.Grid2d(6 if ng == 1 else 9, 8 if ng == 1 else 7, ng=ng)
            d = patch.CellCenterData2d(g)
            bc = bnd.BC(xlb=bcs[0], xrb=bcs[1], ylb=bcs[2], yrb=bcs[3])
            d.register_var("a", bc)
            d.create()
            a = d.get_var("a")
            a[:, :] = rng.standard_normal(a.shape)
            out[f"in_ng{ng}_{k}"] = np.array(a)
            d.fill_BC("a")
            out[f"out_ng{ng}_{k}"] = np.array(d.get_var("a"))
            out[f"bc_ng{ng}_{k}"] = np.array(bcs)
    save("fill_bc", **out)


# --------------------------------------------------------------------------
# advection (a2, a4-a6)
# --------------------------------------------------------------------------
def gen_advection():
    # (i) stage dumps of one step on small grids, all velocity sign combos
    out = {}
    rng = np.random.default_rng(2)
    k = 0
    for (nx, ny) in ((12, 9), (8, 16)):
        for (u, v) in ((1.0, 1.0), (-0.7, 0.4), (0.3, -1.2), (-1.0, -0.5), (0.0, 1.0)):
            for limiter in (0, 1, 2):
                if limiter != 2 and (u, v) != (-0.7, 0.4):
                    continue
                p = Pyro("advection")
                p.initialize_problem("smooth", inputs_dict={
                    "mesh.nx": nx, "mesh.ny": ny, "advection.u": u,
                    "advection.v": v, "advection.limiter": limiter,
                    "particles.do_particles": 0,
                    "mesh.xlboundary": "outflow" if k % 2 else "periodic",
                    "mesh.xrboundary": "outflow" if k % 2 else "periodic"})
                sim = p.sim
                dens = sim.cc_data.get_var("density")
                dens[:, :] += 0.3 * rng.standard_normal(dens.shape)  # kinks
                sim.cc_data.fill_BC_all()
                sim.compute_timestep()
                myg = sim.cc_data.grid
                a0 = np.array(dens)
                lx = reconstruction.limit(dens, myg, 1, limiter)
                ly = reconstruction.limit(dens, myg, 2, limiter)
                _, _, a_x, a_y = linear_interface(dens, myg, sim.rp, sim.dt)
                Fx, Fy = advective_fluxes.unsplit_fluxes(
                    sim.cc_data, sim.rp, sim.dt, "density", linear_interface)
                sim.evolve()
                pre = f"s{k}_"
                out[pre + "meta"] = np.array([nx, ny, 4, myg.dx, myg.dy, u, v,
                                              sim.dt, limiter])
                out[pre + "a0"] = a0
                out[pre + "ldx"] = np.array(lx)
                out[pre + "ldy"] = np.array(ly)
                out[pre + "ax"] = np.array(a_x)
                out[pre + "ay"] = np.array(a_y)
                out[pre + "Fx"] = np.array(Fx)
                out[pre + "Fy"] = np.array(Fy)
                out[pre + "a1"] = np.array(sim.cc_data.get_var("density"))
                k += 1
    out["ncases"] = np.array(k)
    save("adv_stages", **out)

    # (ii) the reference's own regression: advection smooth inputs.smooth
    # (32^2, 40 steps) vs pyro/advection/tests/smooth_0040.h5 (test.py:93)
    p = Pyro("advection")
    p.initialize_problem("smooth", inputs_dict={"particles.do_particles": 0})
    ic = np.array(p.sim.cc_data.get_var("density"))
    dts = []
    while not p.sim.finished():
        p.single_step()
        dts.append(p.sim.dt)
    with h5py.File(REF + "/advection/tests/smooth_0040.h5", "r") as f:
        gold = f["state/density/data"][...]
        assert f.attrs["nsteps"] == p.sim.n == 40
        tgold = f.attrs["time"]
    mine = p.sim.cc_data.get_var("density").v()
    err = np.abs(mine - gold).max()
    print("advection smooth 32^2 x40: reference-run vs stored golden, max abs err", err)
    assert err < 1e-13
    save("adv_smooth_0040", ic=ic, gold=gold, run=np.array(mine),
         dts=np.array(dts), t=np.array(tgold))

    # (iii) 64^2 run to tmax (81 steps): convergence-table known answer
    # pyro/advection/tests/advection_convergence.txt:8
    p = Pyro("advection")
    p.initialize_problem("smooth", inputs_dict={"mesh.nx": 64, "mesh.ny": 64,
                                                "particles.do_particles": 0})
    ic = np.array(p.sim.cc_data.get_var("density"))
    dts = []
    while not p.sim.finished():
        p.single_step()
        dts.append(p.sim.dt)
    fin = np.array(p.sim.cc_data.get_var("density"))
    save("adv_smooth_64", ic=ic, final=fin, dts=np.array(dts), n=np.array(p.sim.n))


# --------------------------------------------------------------------------
# compressible (a3, a7-a12)
# --------------------------------------------------------------------------
def comp_stage_dump(sim):
    """replicate Simulation.evolve (compressible/simulation.py:290-450) by
    CALLING the reference's functions, keeping every intermediate array."""
    rp, ivars, myg, tc = sim.rp, sim.ivars, sim.cc_data.grid, sim.tc
    dt = sim.dt
    gamma = rp.get_param("eos.gamma")
    st = {}
    sim.clean_state(sim.cc_data.data)
    U0 = np.array(sim.cc_data.data)
    st["U0"] = U0
    q = comp.cons_to_prim(sim.cc_data.data, gamma, ivars, myg)
    st["q"] = np.array(q)
    if rp.get_param("compressible.use_flattening"):
        xi_x = reconstruction.flatten(myg, q, 1, ivars, rp)
        xi_y = reconstruction.flatten(myg, q, 2, ivars, rp)
        xi = reconstruction.flatten_multid(myg, q, xi_x, xi_y, ivars)
        st["xi"] = np.array(xi)
    else:
        xi = 1.0
        st["xi"] = np.ones((myg.qx, myg.qy))
    limiter = rp.get_param("compressible.limiter")
    ldx = myg.scratch_array(nvar=ivars.nvar)
    ldy = myg.scratch_array(nvar=ivars.nvar)
    for n in range(ivars.nvar):
        ldx[:, :, n] = xi * reconstruction.limit(q[:, :, n], myg, 1, limiter)
        ldy[:, :, n] = xi * reconstruction.limit(q[:, :, n], myg, 2, limiter)
    st["ldx"], st["ldy"] = np.array(ldx), np.array(ldy)

    U_xl, U_xr, U_yl, U_yr = flx.interface_states(sim.cc_data, rp, ivars, tc, dt)
    U_xl, U_xr, U_yl, U_yr = flx.apply_source_terms(
        U_xl, U_xr, U_yl, U_yr, sim.cc_data, sim.aux_data, rp, ivars, tc, dt,
        problem_source=sim.problem_source)
    for nm, a in (("Uxl0", U_xl), ("Uxr0", U_xr), ("Uyl0", U_yl), ("Uyr0", U_yr)):
        st[nm] = np.array(a)
    FxT = riemann.riemann_flux(1, U_xl, U_xr, sim.cc_data, rp, ivars,
                               sim.solid.xl, sim.solid.xr, tc)
    FyT = riemann.riemann_flux(2, U_yl, U_yr, sim.cc_data, rp, ivars,
                               sim.solid.yl, sim.solid.yr, tc)
    st["FxT"], st["FyT"] = np.array(FxT), np.array(FyT)
    U_xl, U_xr, U_yl, U_yr = flx.apply_transverse_flux(
        U_xl, U_xr, U_yl, U_yr, sim.cc_data, rp, ivars, sim.solid, tc, dt)
    for nm, a in (("Uxl", U_xl), ("Uxr", U_xr), ("Uyl", U_yl), ("Uyr", U_yr)):
        st[nm] = np.array(a)
    F_x = riemann.riemann_flux(1, U_xl, U_xr, sim.cc_data, rp, ivars,
                               sim.solid.xl, sim.solid.xr, tc)
    F_y = riemann.riemann_flux(2, U_yl, U_yr, sim.cc_data, rp, ivars,
                               sim.solid.yl, sim.solid.yr, tc)
    st["Fx0"], st["Fy0"] = np.array(F_x), np.array(F_y)
    cvisc = rp.get_param("compressible.cvisc")
    _ax, _ay = ifc.artificial_viscosity(
        myg.ng, myg.dx, myg.dy, myg.Lx, myg.Ly, myg.xmin, myg.ymin,
        myg.coord_type, cvisc, q.v(n=ivars.iu, buf=myg.ng),
        q.v(n=ivars.iv, buf=myg.ng))
    st["avx"], st["avy"] = np.array(_ax), np.array(_ay)
    F_x, F_y = flx.apply_artificial_viscosity(F_x, F_y, q, sim.cc_data, rp, ivars)
    st["Fx"], st["Fy"] = np.array(F_x), np.array(F_y)
    # and now the reference's own evolve for the end state
    sim.evolve()
    st["U1"] = np.array(sim.cc_data.data)
    return st


def bc_names(rp):
    return np.array([rp.get_param("mesh." + k) for k in
                     ("xlboundary", "xrboundary", "ylboundary", "yrboundary")])


def comp_meta(sim):
    rp, g = sim.rp, sim.cc_data.grid
    return np.array([g.nx, g.ny, g.ng, g.dx, g.dy, rp.get_param("eos.gamma"),
                     rp.get_param("compressible.limiter"),
                     rp.get_param("compressible.use_flattening"),
                     rp.get_param("compressible.z0"), rp.get_param("compressible.z1"),
                     rp.get_param("compressible.delta"),
                     rp.get_param("compressible.cvisc"),
                     rp.get_param("compressible.grav"),
                     rp.get_param("driver.cfl")])


def gen_compressible_stages():
    cases = [
        ("sedov", None, {"mesh.nx": 24, "mesh.ny": 24, "sedov.r_init": 0.12}, 9),
        ("sedov", None, {"mesh.nx": 20, "mesh.ny": 28, "sedov.r_init": 0.15,
                         "compressible.limiter": 1}, 7),
        ("quad", None, {"mesh.nx": 24, "mesh.ny": 20}, 12),
        ("quad", None, {"mesh.nx": 16, "mesh.ny": 16,
                        "compressible.use_flattening": 0,
                        "compressible.limiter": 0}, 6),
        ("sod", "inputs.sod.x", {"mesh.nx": 32, "mesh.ny": 8}, 10),
        ("sod", "inputs.sod.y", {"mesh.nx": 8, "mesh.ny": 32,
                                 "compressible.limiter": 2}, 10),
        ("kh", None, {"mesh.nx": 16, "mesh.ny": 24}, 8),
        ("sedov", None, {"mesh.nx": 16, "mesh.ny": 16, "sedov.r_init": 0.2,
                         "mesh.xlboundary": "reflect", "mesh.xrboundary": "reflect",
                         "mesh.ylboundary": "reflect", "mesh.yrboundary": "reflect",
                         "compressible.grav": -0.5}, 6),
    ]
    out = {"ncases": np.array(len(cases))}
    for k, (prob, inp, d, nsteps) in enumerate(cases):
        p = Pyro("compressible")
        p.initialize_problem(prob, inputs_file=inp, inputs_dict=d)
        sim = p.sim
        dts = []
        for _ in range(nsteps):
            p.single_step()
            dts.append(sim.dt)
        sim.cc_data.fill_BC_all()
        sim.compute_timestep()
        pre = f"c{k}_"
        out[pre + "meta"] = comp_meta(sim)
        out[pre + "bc"] = bc_names(sim.rp)
        out[pre + "dt"] = np.array(sim.dt)
        out[pre + "dt_method"] = np.array(_raw_cfl_dt(sim))
        st = comp_stage_dump(sim)
        for nm, a in st.items():
            out[pre + nm] = a
        print("compressible stage case", k, prob, d, "dt", sim.dt)
    save("comp_stages", **out)


def gen_compressible_f2():
    """rows f2: CGF Riemann solver and sponge; same dump format as comp_stages"""
    sp = {"sponge.do_sponge": 1, "sponge.sponge_rho_begin": 1.05,
          "sponge.sponge_rho_full": 0.3, "sponge.sponge_timescale": 0.02}
    cases = [
        ("sedov", None, {"mesh.nx": 20, "mesh.ny": 24, "sedov.r_init": 0.15,
                         "compressible.riemann": "CGF"}, 8),
        ("sod", "inputs.sod.x", {"mesh.nx": 32, "mesh.ny": 8, "compressible.riemann": "CGF",
                                 "mesh.xlboundary": "reflect", "mesh.xrboundary": "reflect"}, 12),
        ("sod", "inputs.sod.y", {"mesh.nx": 8, "mesh.ny": 32, "compressible.riemann": "CGF",
                                 "compressible.limiter": 2}, 9),
        ("kh", None, {"mesh.nx": 16, "mesh.ny": 24, "compressible.riemann": "CGF"}, 7),
        ("sedov", None, dict({"mesh.nx": 20, "mesh.ny": 20, "sedov.r_init": 0.15}, **sp), 8),
        ("quad", None, dict({"mesh.nx": 16, "mesh.ny": 24, "compressible.riemann": "CGF"}, **sp), 10),
    ]
    out = {"ncases": np.array(len(cases))}
    for k, (prob, inp, d, nsteps) in enumerate(cases):
        p = Pyro("compressible")
        p.initialize_problem(prob, inputs_file=inp, inputs_dict=d)
        sim = p.sim
        for _ in range(nsteps):
            p.single_step()
        sim.cc_data.fill_BC_all()
        sim.compute_timestep()
        pre = f"c{k}_"
        out[pre + "meta"] = comp_meta(sim)
        out[pre + "bc"] = bc_names(sim.rp)
        out[pre + "riemann"] = np.array(sim.rp.get_param("compressible.riemann"))
        out[pre + "sponge"] = np.array([sim.rp.get_param("sponge.do_sponge"),
                                        sim.rp.get_param("sponge.sponge_rho_begin"),
                                        sim.rp.get_param("sponge.sponge_rho_full"),
                                        sim.rp.get_param("sponge.sponge_timescale")])
        out[pre + "dt"] = np.array(sim.dt)
        st = comp_stage_dump(sim)
        for nm in ("U0", "FxT", "FyT", "Fx0", "Fy0", "Fx", "Fy", "U1"):
            out[pre + nm] = st[nm]
        print("f2 case", k, prob, d.get("compressible.riemann", "HLLC"),
              "sponge" if d.get("sponge.do_sponge") else "", "dt", sim.dt)
    save("comp_stages_f2", **out)


def gen_compressible_hse():
    """row f2: gravity with the hse / ambient user boundaries
    (compressible/BC.py).  Whole short runs (IC -> final, dt sequence) plus a
    stage dump of the step after the last one."""
    cases = [
        ("rt", None, {"mesh.nx": 16, "mesh.ny": 48}, 25, None),
        ("rt", None, {"mesh.nx": 12, "mesh.ny": 40, "mesh.xlboundary": "outflow",
                      "mesh.xrboundary": "reflect", "rt.amp": 0.4,
                      "compressible.riemann": "CGF"}, 20, None),
        ("hse", None, {"mesh.nx": 8, "mesh.ny": 32}, 12, None),
        ("rt", None, {"mesh.nx": 16, "mesh.ny": 32, "mesh.yrboundary": "ambient",
                      "rt.amp": 0.5}, 20, (0.8, 0.1, -0.2, 6.0)),
    ]
    out = {"ncases": np.array(len(cases))}
    for k, (prob, inp, d, nsteps, amb) in enumerate(cases):
        p = Pyro("compressible")
        if amb is not None:
            # the aux values must exist before the first fill_BC_all
            import pyro.mesh.patch as _patch
            orig = _patch.CellCenterData2d.create

            def create(self, _orig=orig, _amb=amb):
                _orig(self)
                for nm, v in zip(("ambient_rho", "ambient_u", "ambient_v", "ambient_p"), _amb):
                    self.set_aux(nm, v)
            _patch.CellCenterData2d.create = create
        p.initialize_problem(prob, inputs_file=inp, inputs_dict=d)
        if amb is not None:
            _patch.CellCenterData2d.create = orig
        sim = p.sim
        pre = f"c{k}_"
        out[pre + "ic"] = np.array(sim.cc_data.data)
        dts = []
        for _ in range(nsteps):
            p.single_step()
            dts.append(sim.dt)
        out[pre + "final"] = np.array(sim.cc_data.data)
        out[pre + "dts"] = np.array(dts)
        sim.cc_data.fill_BC_all()
        out[pre + "filled"] = np.array(sim.cc_data.data)
        sim.compute_timestep()
        out[pre + "meta"] = comp_meta(sim)
        out[pre + "bc"] = bc_names(sim.rp)
        out[pre + "ambient"] = np.array(amb if amb is not None else (0.0,) * 4)
        out[pre + "drv"] = np.array([p.rp.get_param("driver.init_tstep_factor"),
                                     p.rp.get_param("driver.max_dt_change")])
        out[pre + "dt"] = np.array(sim.dt)
        st = comp_stage_dump(sim)
        for nm in ("U0", "FxT", "FyT", "Fx", "Fy", "U1"):
            out[pre + nm] = st[nm]
        print("hse case", k, prob, d, "dt", sim.dt)
    save("comp_hse", **out)


def gen_compressible_rt():
    """reference regression compressible rt (inputs.rt, 64x192, 945 steps) vs
    pyro/compressible/tests/rt_0945.h5 (test.py:102): the reference's IC and
    its stored end state (like quad: too slow to re-run interpreted)."""
    names = ["density", "energy", "x-momentum", "y-momentum"]
    p = Pyro("compressible")
    p.initialize_problem("rt", inputs_file="inputs.rt")
    ic = np.array(p.sim.cc_data.data)
    with h5py.File(REF + "/compressible/tests/rt_0945.h5", "r") as f:
        nsteps = int(f.attrs["nsteps"])
        tfin = float(f.attrs["time"])
        g = np.stack([f["state/" + nm + "/data"][...] for nm in names], axis=-1)
    save("comp_rt_0945", ic=ic, gold=g, nsteps=np.array(nsteps), t=np.array(tfin),
         meta=comp_meta(p.sim), bc=bc_names(p.sim.rp), tmax=np.array(p.sim.tmax),
         drv=np.array([p.rp.get_param("driver.init_tstep_factor"),
                       p.rp.get_param("driver.max_dt_change")]))


def _planes(cc):
    """(nvar, qx, qy) planar copy of a CellCenterData2d"""
    return np.ascontiguousarray(np.moveaxis(np.array(cc.data), -1, 0))


def gen_incompressible():
    """rows f1/f4: burgers (the CTU predictor shared with incompressible) and
    the incompressible projection solver on top of the multigrid solver."""
    import pyro.burgers.burgers_interface as bi
    import pyro.incompressible.incomp_interface as ii
    import pyro.incompressible.simulation as isim
    from pyro.mesh import reconstruction as rec

    out = {}
    # ---- burgers: test problem (diagonal shock, outflow), no particles
    for k, (nx, ny, lim, nsteps) in enumerate([(24, 24, 2, 10), (20, 28, 1, 8)]):
        p = Pyro("burgers")
        p.initialize_problem("test", inputs_dict={"mesh.nx": nx, "mesh.ny": ny,
                                                  "advection.limiter": lim,
                                                  "particles.do_particles": 0})
        sim = p.sim
        pre = f"b{k}_"
        out[pre + "ic"] = _planes(sim.cc_data)
        dts = []
        for _ in range(nsteps):
            p.single_step()
            dts.append(sim.dt)
        out[pre + "final"] = _planes(sim.cc_data)
        out[pre + "dts"] = np.array(dts)
        g = sim.cc_data.grid
        out[pre + "meta"] = np.array([g.nx, g.ny, g.ng, g.dx, g.dy, lim,
                                      sim.rp.get_param("driver.cfl")])
        out[pre + "bc"] = bc_names(sim.rp)
        # edge states of the next step
        sim.cc_data.fill_BC_all()
        sim.compute_timestep()
        u, v = sim.cc_data.get_var("x-velocity"), sim.cc_data.get_var("y-velocity")
        ld = [rec.limit(a, g, d, lim) for a, d in ((u, 1), (v, 1), (u, 2), (v, 2))]
        E = bi.get_interface_states(g, sim.dt, u, v, *ld)
        E = bi.apply_transverse_corrections(g, sim.dt, *E)
        out[pre + "U0"] = _planes(sim.cc_data)
        out[pre + "dt"] = np.array(sim.dt)
        out[pre + "E"] = np.array([np.array(e) for e in E])
        print("burgers case", k, nx, ny, "dt", sim.dt)
    out["nburgers"] = np.array(2)

    # ---- incompressible shear: IC before preevolve, state after it, a run,
    #      and the pieces of one more step
    store = {}
    orig_pre = isim.Simulation.preevolve

    def preevolve(self):
        store["ic"] = _planes(self.cc_data)
        orig_pre(self)
        store["after_pre"] = _planes(self.cc_data)
    isim.Simulation.preevolve = preevolve
    for k, (nx, lim, proj, nsteps) in enumerate([(32, 2, 2, 6), (16, 1, 1, 5)]):
        p = Pyro("incompressible")
        p.initialize_problem("shear", inputs_dict={"mesh.nx": nx, "mesh.ny": nx,
                                                   "incompressible.limiter": lim,
                                                   "incompressible.proj_type": proj})
        sim = p.sim
        pre = f"i{k}_"
        out[pre + "ic"] = store["ic"]
        out[pre + "after_pre"] = store["after_pre"]
        dts = []
        for _ in range(nsteps):
            p.single_step()
            dts.append(sim.dt)
        out[pre + "final"] = _planes(sim.cc_data)
        out[pre + "dts"] = np.array(dts)
        g = sim.cc_data.grid
        out[pre + "meta"] = np.array([g.nx, g.ng, lim, proj, sim.rp.get_param("driver.cfl"),
                                      sim.rp.get_param("driver.init_tstep_factor"),
                                      sim.rp.get_param("driver.max_dt_change")])
        # one more step with the MAC velocities captured
        sim.cc_data.fill_BC_all()
        sim.compute_timestep()
        out[pre + "U0"] = _planes(sim.cc_data)
        out[pre + "dt"] = np.array(sim.dt)
        cap = {}
        orig_states = ii.states

        def states(grid, dt, u, v, a, b, c, d, gx, gy, u_MAC, v_MAC, sx=None, sy=None):
            cap["umac"], cap["vmac"] = np.array(u_MAC), np.array(v_MAC)
            return orig_states(grid, dt, u, v, a, b, c, d, gx, gy, u_MAC, v_MAC, sx, sy)
        ii.states = states
        sim.evolve()
        ii.states = orig_states
        out[pre + "umac"], out[pre + "vmac"] = cap["umac"], cap["vmac"]
        out[pre + "U1"] = _planes(sim.cc_data)
        print("incompressible case", k, nx, "dt", sim.dt)
    out["nincomp"] = np.array(2)
    save("incomp", **out)

    # ---- reference regression: incompressible shear inputs.shear (128^2, 216
    # steps) vs pyro/incompressible/tests/shear_128_0216.h5 (test.py:110)
    p = Pyro("incompressible")
    p.initialize_problem("shear", inputs_file="inputs.shear")
    ic = store["ic"]
    while not p.sim.finished():
        p.single_step()
    names = ["x-velocity", "y-velocity", "phi-MAC", "phi", "gradp_x", "gradp_y"]
    with h5py.File(REF + "/incompressible/tests/shear_128_0216.h5", "r") as f:
        assert int(f.attrs["nsteps"]) == p.sim.n, (f.attrs["nsteps"], p.sim.n)
        gold = np.array([f["state/" + nm + "/data"][...] for nm in names])
        tfin = float(f.attrs["time"])
    run = np.array([np.array(p.sim.cc_data.get_var(nm).v()) for nm in names])
    print("shear_128: reference-run vs stored golden, max abs err per var",
          np.abs(run - gold).max(axis=(1, 2)))
    isim.Simulation.preevolve = orig_pre
    save("incomp_shear_128_0216", ic=ic[:2], gold=gold[:2].astype(np.float64),
         gold_gp=gold[4:6], run=run[:2], run_gp=run[4:6], nsteps=np.array(p.sim.n),
         t=np.array(tfin), tmax=np.array(p.sim.tmax),
         meta=np.array([128, 4, 2, 2, p.rp.get_param("driver.cfl"),
                        p.rp.get_param("driver.init_tstep_factor"),
                        p.rp.get_param("driver.max_dt_change")]))


def gen_compressible_rk():
    """row f4: compressible_rk (method of lines + RK integrators): short runs
    and the right-hand side k = -div F + S of the state after them"""
    sp = {"sponge.do_sponge": 1, "sponge.sponge_rho_begin": 1.05,
          "sponge.sponge_rho_full": 0.3, "sponge.sponge_timescale": 0.02}
    cases = [
        ("sedov", None, {"mesh.nx": 16, "mesh.ny": 16, "sedov.r_init": 0.2}, 4),
        ("rt", None, {"mesh.nx": 12, "mesh.ny": 36, "rt.amp": 0.4,
                      "compressible.temporal_method": "TVD3"}, 5),
        ("sod", "inputs.sod.x", {"mesh.nx": 32, "mesh.ny": 8, "compressible.riemann": "CGF",
                                 "compressible.temporal_method": "RK2"}, 6),
        ("quad", None, dict({"mesh.nx": 16, "mesh.ny": 16,
                             "compressible.temporal_method": "TVD2",
                             "compressible.limiter": 1}, **sp), 5),
    ]
    out = {"ncases": np.array(len(cases))}
    for k, (prob, inp, d, nsteps) in enumerate(cases):
        p = Pyro("compressible_rk")
        p.initialize_problem(prob, inputs_file=inp, inputs_dict=d)
        sim = p.sim
        pre = f"c{k}_"
        out[pre + "ic"] = np.array(sim.cc_data.data)
        dts = []
        for _ in range(nsteps):
            p.single_step()
            dts.append(sim.dt)
        out[pre + "final"] = np.array(sim.cc_data.data)
        out[pre + "dts"] = np.array(dts)
        out[pre + "meta"] = comp_meta(sim)
        out[pre + "bc"] = bc_names(sim.rp)
        out[pre + "method"] = np.array(sim.rp.get_param("compressible.temporal_method"))
        out[pre + "riemann"] = np.array(sim.rp.get_param("compressible.riemann"))
        out[pre + "sponge"] = np.array([sim.rp.get_param("sponge.do_sponge"),
                                        sim.rp.get_param("sponge.sponge_rho_begin"),
                                        sim.rp.get_param("sponge.sponge_rho_full"),
                                        sim.rp.get_param("sponge.sponge_timescale")])
        out[pre + "drv"] = np.array([p.rp.get_param("driver.init_tstep_factor"),
                                     p.rp.get_param("driver.max_dt_change")])
        sim.cc_data.fill_BC_all()
        sim.compute_timestep()
        out[pre + "U0"] = np.array(sim.cc_data.data)
        out[pre + "dt"] = np.array(sim.dt)
        out[pre + "k"] = np.array(sim.substep(sim.cc_data))
        print("compressible_rk case", k, prob, d, "dt", sim.dt)
    save("comp_rk", **out)


def gen_swe():
    """row f4: shallow-water solver (pyro/swe): short runs and the stages of
    one more step (face states before / after the transverse terms, fluxes)"""
    import pyro.swe as swe
    import pyro.swe.interface as sifc
    from pyro.mesh import reconstruction as rec
    cases = [
        ("dam", "inputs.dam.x", {"mesh.nx": 32, "mesh.ny": 8}, 10),
        ("dam", "inputs.dam.y", {"mesh.nx": 8, "mesh.ny": 32, "swe.riemann": "HLLC",
                                 "swe.limiter": 2}, 10),
        ("quad", None, {"mesh.nx": 16, "mesh.ny": 16, "swe.riemann": "HLLC"}, 8),
        ("quad", None, {"mesh.nx": 16, "mesh.ny": 20, "swe.riemann": "Roe",
                        "swe.limiter": 0, "swe.grav": 2.0}, 6),
        ("kh", None, {"mesh.nx": 16, "mesh.ny": 16}, 6),
    ]
    out = {"ncases": np.array(len(cases))}
    for k, (prob, inp, d, nsteps) in enumerate(cases):
        p = Pyro("swe")
        p.initialize_problem(prob, inputs_file=inp, inputs_dict=d)
        sim = p.sim
        rp, ivars, myg = sim.rp, sim.ivars, sim.cc_data.grid
        pre = f"c{k}_"
        out[pre + "ic"] = np.array(sim.cc_data.data)
        dts = []
        for _ in range(nsteps):
            p.single_step()
            dts.append(sim.dt)
        out[pre + "final"] = np.array(sim.cc_data.data)
        out[pre + "dts"] = np.array(dts)
        g = rp.get_param("swe.grav")
        limiter = rp.get_param("swe.limiter")
        out[pre + "meta"] = np.array([myg.nx, myg.ny, myg.ng, myg.dx, myg.dy, g, limiter,
                                      rp.get_param("driver.cfl")])
        out[pre + "riemann"] = np.array(rp.get_param("swe.riemann"))
        out[pre + "bc"] = bc_names(rp)
        out[pre + "drv"] = np.array([p.rp.get_param("driver.init_tstep_factor"),
                                     p.rp.get_param("driver.max_dt_change")])
        # stages of one more step
        sim.cc_data.fill_BC_all()
        sim.compute_timestep()
        dt = sim.dt
        out[pre + "U0"] = np.array(sim.cc_data.data)
        out[pre + "dt"] = np.array(dt)
        q = swe.cons_to_prim(sim.cc_data.data, ivars, myg)
        ldx = myg.scratch_array(nvar=ivars.nvar)
        ldy = myg.scratch_array(nvar=ivars.nvar)
        for n in range(ivars.nvar):
            ldx[:, :, n] = rec.limit(q[:, :, n], myg, 1, limiter)
            ldy[:, :, n] = rec.limit(q[:, :, n], myg, 2, limiter)
        args = (ivars.ih, ivars.iu, ivars.iv, ivars.ix, ivars.naux, g)
        V_l, V_r = sifc.states(1, myg.ng, myg.dx, dt, *args, q, ldx)
        out[pre + "Uxl0"] = np.array(swe.prim_to_cons(V_l, ivars, myg))
        out[pre + "Uxr0"] = np.array(swe.prim_to_cons(V_r, ivars, myg))
        V_l, V_r = sifc.states(2, myg.ng, myg.dy, dt, *args, q, ldy)
        out[pre + "Uyl0"] = np.array(swe.prim_to_cons(ai.ArrayIndexer(d=V_l, grid=myg), ivars, myg))
        out[pre + "Uyr0"] = np.array(swe.prim_to_cons(ai.ArrayIndexer(d=V_r, grid=myg), ivars, myg))
        rf = sifc.riemann_hllc if rp.get_param("swe.riemann") == "HLLC" else sifc.riemann_roe
        rargs = (ivars.ih, ivars.ixmom, ivars.iymom, ivars.ihx, ivars.naux)
        out[pre + "FxT"] = np.array(rf(1, myg.ng, *rargs, sim.solid.xl, sim.solid.xr, g,
                                       out[pre + "Uxl0"], out[pre + "Uxr0"]))
        out[pre + "FyT"] = np.array(rf(2, myg.ng, *rargs, sim.solid.yl, sim.solid.yr, g,
                                       out[pre + "Uyl0"], out[pre + "Uyr0"]))
        import pyro.swe.unsplit_fluxes as sflx
        Fx, Fy = sflx.unsplit_fluxes(sim.cc_data, rp, ivars, sim.solid, sim.tc, dt)
        out[pre + "Fx"], out[pre + "Fy"] = np.array(Fx), np.array(Fy)
        sim.evolve()
        out[pre + "U1"] = np.array(sim.cc_data.data)
        print("swe case", k, prob, d, "dt", dt)
    save("swe", **out)

    # reference regression: swe dam inputs.dam.x (128x10, 81 steps) vs
    # pyro/swe/tests/dam_x_0081.h5 (test.py:113)
    p = Pyro("swe")
    p.initialize_problem("dam", inputs_file="inputs.dam.x")
    ic = np.array(p.sim.cc_data.data)
    dts = []
    while not p.sim.finished():
        p.single_step()
        dts.append(p.sim.dt)
    names = ["height", "x-momentum", "y-momentum", "fuel"]
    with h5py.File(REF + "/swe/tests/dam_x_0081.h5", "r") as f:
        assert int(f.attrs["nsteps"]) == p.sim.n, (f.attrs["nsteps"], p.sim.n)
        gold = np.stack([f["state/" + nm + "/data"][...] for nm in names], axis=-1)
    run = np.stack([np.array(p.sim.cc_data.get_var(nm).v()) for nm in names], axis=-1)
    print("dam_x: reference-run vs stored golden, max abs err", np.abs(run - gold).max())
    myg = p.sim.cc_data.grid
    save("swe_dam_x_0081", ic=ic, gold=gold, run=run, dts=np.array(dts),
         meta=np.array([myg.nx, myg.ny, myg.ng, myg.dx, myg.dy, p.rp.get_param("swe.grav"),
                        p.rp.get_param("swe.limiter"), p.rp.get_param("driver.cfl")]),
         bc=bc_names(p.rp), tmax=np.array(p.sim.tmax), riemann=np.array("Roe"))


def gen_mg_4096():
    """BASELINE config 4 at full size: the reference's CellCenterMG2d(4096^2),
    10 V-cycles.  The solution does not fit a fixture, so
    a 64x64 lattice of samples, row / column checksums and the norms are kept."""
    nx = 4096
    a = MG.CellCenterMG2d(nx, nx, xl_BC_type="dirichlet", yl_BC_type="dirichlet",
                          xr_BC_type="dirichlet", yr_BC_type="dirichlet", verbose=0)
    a.init_zeros()
    x, y = a.x2d, a.y2d
    rhs = -2.0 * ((1.0 - 6.0 * x**2) * y**2 * (1.0 - y**2) +
                  (1.0 - 6.0 * y**2) * x**2 * (1.0 - x**2))
    a.init_RHS(rhs)
    a.max_cycles = 10
    a.solve(rtol=0.0)
    v = np.array(a.get_solution())
    step = nx // 64
    save("mg_4096_samples", samples=v[1:-1:step, 1:-1:step].copy(),
         row_sums=v[1:-1, 1:-1].sum(axis=1), col_sums=v[1:-1, 1:-1].sum(axis=0),
         source_norm=np.array(a.source_norm), residual_error=np.array(a.residual_error),
         relative_error=np.array(a.relative_error), num_cycles=np.array(a.num_cycles),
         vnorm=np.array(a.get_solution().norm()))
    print("mg 4096: residual_error", a.residual_error, "cycles", a.num_cycles)


def gen_compressible_lm():
    """row f2: the low-Mach HLLC variant (compressible.riemann = HLLC_lm,
    riemann_hllc_lowspeed); same dump format as comp_stages_f2"""
    cases = [
        ("sedov", None, {"mesh.nx": 20, "mesh.ny": 24, "sedov.r_init": 0.15,
                         "compressible.riemann": "HLLC_lm"}, 8),
        ("kh", None, {"mesh.nx": 16, "mesh.ny": 24, "compressible.riemann": "HLLC_lm"}, 7),
        ("sod", "inputs.sod.y", {"mesh.nx": 8, "mesh.ny": 32, "compressible.riemann": "HLLC_lm",
                                 "compressible.limiter": 2}, 9),
        ("rt", None, {"mesh.nx": 12, "mesh.ny": 36, "rt.amp": 0.4,
                      "compressible.riemann": "HLLC_lm"}, 10),
    ]
    out = {"ncases": np.array(len(cases))}
    for k, (prob, inp, d, nsteps) in enumerate(cases):
        p = Pyro("compressible")
        p.initialize_problem(prob, inputs_file=inp, inputs_dict=d)
        sim = p.sim
        for _ in range(nsteps):
            p.single_step()
        sim.cc_data.fill_BC_all()
        sim.compute_timestep()
        pre = f"c{k}_"
        out[pre + "meta"] = comp_meta(sim)
        out[pre + "bc"] = bc_names(sim.rp)
        out[pre + "dt"] = np.array(sim.dt)
        st = comp_stage_dump(sim)
        for nm in ("U0", "FxT", "FyT", "Fx0", "Fy0", "Fx", "Fy", "U1"):
            out[pre + nm] = st[nm]
        print("HLLC_lm case", k, prob, "dt", sim.dt)
    save("comp_stages_lm", **out)


def gen_compressible_ramp():
    """row f2: the double Mach reflection problem with its time-dependent
    "ramp" boundary (compressible/BC.py:178-296): a short run, and the ghost
    fill at the end time"""
    p = Pyro("compressible")
    p.initialize_problem("ramp", inputs_dict={"mesh.nx": 48, "mesh.ny": 12})
    sim = p.sim
    out = {"ic": np.array(sim.cc_data.data)}
    dts = []
    for _ in range(12):
        p.single_step()
        dts.append(sim.dt)
    out["final"] = np.array(sim.cc_data.data)
    out["dts"] = np.array(dts)
    out["t"] = np.array(sim.cc_data.t)
    sim.cc_data.fill_BC_all()
    out["filled"] = np.array(sim.cc_data.data)
    out["meta"] = comp_meta(sim)
    out["bc"] = bc_names(sim.rp)
    g = sim.cc_data.grid
    out["domain"] = np.array([g.xmin, g.xmax, g.ymin, g.ymax])
    out["drv"] = np.array([p.rp.get_param("driver.init_tstep_factor"),
                           p.rp.get_param("driver.max_dt_change")])
    print("ramp: t", sim.cc_data.t, "dt", sim.dt)
    save("comp_ramp", **out)


def gen_compressible_heating():
    """row f2: problem source terms of the heating / plume / convection
    problems (S[energy] += rho * e_rate * exp(-(dist/r)^2)): short runs, the
    profile the reference evaluates, and the stages of one more step"""
    cases = [
        ("heating", {"mesh.nx": 24, "mesh.ny": 24, "heating.r_src": 0.15, "heating.e_rate": 5.0}, 12),
        ("plume", {"mesh.nx": 16, "mesh.ny": 32, "plume.r_pert": 0.6}, 12),
        ("convection", {"mesh.nx": 16, "mesh.ny": 48, "convection.thickness": 0.5}, 12),
    ]
    out = {"ncases": np.array(len(cases))}
    for k, (prob, d, nsteps) in enumerate(cases):
        p = Pyro("compressible")
        p.initialize_problem(prob, inputs_dict=d)
        sim = p.sim
        rp, myg = sim.rp, sim.cc_data.grid
        pre = f"c{k}_"
        out[pre + "ic"] = np.array(sim.cc_data.data)
        dts = []
        for _ in range(nsteps):
            p.single_step()
            dts.append(sim.dt)
        out[pre + "final"] = np.array(sim.cc_data.data)
        out[pre + "dts"] = np.array(dts)
        out[pre + "meta"] = comp_meta(sim)
        out[pre + "bc"] = bc_names(rp)
        out[pre + "small_dens"] = np.array(rp.get_param("compressible.small_dens"))
        out[pre + "sponge"] = np.array([rp.get_param("sponge.do_sponge"),
                                        rp.get_param("sponge.sponge_rho_begin"),
                                        rp.get_param("sponge.sponge_rho_full"),
                                        rp.get_param("sponge.sponge_timescale")])
        amb = [sim.cc_data.aux.get(nm, 0.0) for nm in
               ("ambient_rho", "ambient_u", "ambient_v", "ambient_p")]
        out[pre + "ambient"] = np.array(amb, dtype=np.float64)
        out[pre + "drv"] = np.array([p.rp.get_param("driver.init_tstep_factor"),
                                     p.rp.get_param("driver.max_dt_change")])
        # the source of a unit-density state: rate * profile as the reference evaluates it
        ones = myg.scratch_array(nvar=sim.ivars.nvar)
        ones[:, :, sim.ivars.idens] = 1.0
        out[pre + "unit_source"] = np.array(sim.problem_source(myg, ones, sim.ivars, rp)[:, :, sim.ivars.iener])
        out[pre + "e_rate"] = np.array(rp.get_param(prob + ".e_rate"))
        # the bare profile exp(-(dist/r)^2), evaluated with this NumPy build's exp
        if prob == "heating":
            dist = np.sqrt((myg.x2d - 0.5 * (myg.xmin + myg.xmax))**2 +
                           (myg.y2d - 0.5 * (myg.ymin + myg.ymax))**2)
            r = rp.get_param("heating.r_src")
        elif prob == "plume":
            dist = np.sqrt((myg.x2d - rp.get_param("plume.x_pert"))**2 +
                           (myg.y2d - rp.get_param("plume.y_pert"))**2)
            r = rp.get_param("plume.r_pert")
        else:
            dist = np.abs(myg.y2d - rp.get_param("convection.y_height"))
            r = rp.get_param("convection.thickness")
        prof = np.exp(-(dist / r)**2)
        assert np.array_equal(1.0 * out[pre + "e_rate"] * prof, out[pre + "unit_source"])
        out[pre + "prof"] = np.array(prof)
        sim.cc_data.fill_BC_all()
        sim.compute_timestep()
        out[pre + "dt"] = np.array(sim.dt)
        st = comp_stage_dump(sim)
        for nm in ("U0", "Uxl0", "Uyr0", "Fx", "Fy", "U1"):
            out[pre + nm] = st[nm]
        print("source case", k, prob, "dt", sim.dt)
    save("comp_heating", **out)


def gen_compressible_general_source():
    """an arbitrary problem source (tests/general_source.py: all four
    components, state dependent) through the reference's predictor / corrector and
    interface-state source terms -- outflow, hse (+ gravity), ambient + sponge, and
    reflecting walls + gravity"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    from general_source import source_terms
    cases = [
        ("sedov", {"mesh.nx": 24, "mesh.ny": 20, "driver.tmax": 100.0}, 10),
        ("plume", {"mesh.nx": 16, "mesh.ny": 32, "plume.r_pert": 0.6}, 10),
        ("convection", {"mesh.nx": 16, "mesh.ny": 48, "convection.thickness": 0.5}, 10),
        ("rt", {"mesh.nx": 16, "mesh.ny": 48, "mesh.ylboundary": "reflect",
                "mesh.yrboundary": "reflect", "mesh.xlboundary": "reflect",
                "mesh.xrboundary": "outflow"}, 10),
    ]
    out = {"ncases": np.array(len(cases))}
    for k, (prob, d, nsteps) in enumerate(cases):
        p = Pyro("compressible")
        p.initialize_problem(prob, inputs_dict=d)
        sim = p.sim
        sim.problem_source = source_terms
        pre = f"c{k}_"
        out[pre + "ic"] = np.array(sim.cc_data.data)
        dts = []
        for _ in range(nsteps):
            p.single_step()
            dts.append(sim.dt)
        out[pre + "final"] = np.array(sim.cc_data.data)
        out[pre + "dts"] = np.array(dts)
        print("general source case", k, prob, "t", sim.cc_data.t)
    save("comp_general_source", **out)


def _regression(solver, problem, inputs, h5name, names, save_as, ng):
    """run the reference (pyro/test.py entry), check it against its stored regression file
    and keep IC, dt sequence, the run's end state and the stored one"""
    p = Pyro(solver)
    p.initialize_problem(problem, inputs_file=inputs)
    ic = np.array(p.sim.cc_data.data)
    dts = []
    while not p.sim.finished():
        p.single_step()
        dts.append(p.sim.dt)
    with h5py.File(REF + "/" + h5name, "r") as f:
        assert int(f.attrs["nsteps"]) == p.sim.n, (f.attrs["nsteps"], p.sim.n)
        gold = np.stack([f["state/" + nm + "/data"][...] for nm in names], axis=-1)
    run = np.stack([np.array(p.sim.cc_data.get_var(nm).v()) for nm in names], axis=-1)
    print(save_as, ": reference run vs stored golden, max abs err", np.abs(run - gold).max(),
          "steps", p.sim.n)
    save(save_as, ic=ic, gold=gold, run=run, dts=np.array(dts), nsteps=np.array(p.sim.n))


def gen_regressions2():
    """pyro/test.py:99 burgers test (test_0051.h5) and :103 compressible_rk rt (rt_1835.h5)"""
    _regression("burgers", "test", "inputs.test", "burgers/tests/test_0051.h5",
                ["x-velocity", "y-velocity"], "burgers_test_0051", 4)
    # compressible_rk rt: 1835 RK steps take hours with the njit kernels stubbed; the
    # stored regression file alone is the target (the initial condition is that of the
    # compressible rt problem, pinned by comp_rt_0945)
    names = ["density", "energy", "x-momentum", "y-momentum"]
    with h5py.File(REF + "/compressible_rk/tests/rt_1835.h5", "r") as f:
        gold = np.stack([f["state/" + nm + "/data"][...] for nm in names], axis=-1)
        save("comp_rk_rt_1835", gold=gold, nsteps=np.array(int(f.attrs["nsteps"])),
             time=np.array(float(f.attrs["time"])))


def gen_mesh_utils():
    """the general mesh utilities next to the hot path: CellCenterData2d.restrict (by 2 and
    by 4) / prolong on a rectangular grid with ng = 2, EdgeCoeffs and its restriction"""
    import pyro.multigrid.edge_coeffs as ec
    rng = np.random.default_rng(11)
    g = patch.Grid2d(12, 8, ng=2, xmax=1.5, ymax=1.0)
    d = patch.CellCenterData2d(g)
    bc = bnd.BC(xlb="outflow", xrb="outflow", ylb="periodic", yrb="periodic")
    d.register_var("a", bc)
    d.create()
    a = d.get_var("a")
    a[:, :] = rng.standard_normal(a.shape)
    out = {"a": np.array(a), "r2": np.array(d.restrict("a")), "r4": np.array(d.restrict("a", N=4)),
           "p": np.array(d.prolong("a"))}
    ge = patch.Grid2d(8, 12, ng=1, xmax=2.0, ymax=3.0)
    eta = ge.scratch_array()
    eta[:, :] = 1.0 + rng.random(eta.shape)
    e = ec.EdgeCoeffs(ge, eta)
    c = e.restrict()
    out.update(eta=np.array(eta), ex=np.array(e.x), ey=np.array(e.y), cx=np.array(c.x),
               cy=np.array(c.y))
    save("mesh_utils", **out)


def gen_problem_ics():
    """initial conditions of the remaining compressible problem set-ups"""
    cases = {"acoustic_pulse": {"mesh.nx": 24, "mesh.ny": 24},
             "advect": {"mesh.nx": 16, "mesh.ny": 20},
             "bubble": {"mesh.nx": 32, "mesh.ny": 64},
             "gresho": {"mesh.nx": 20, "mesh.ny": 20},
             "rt2": {"mesh.nx": 24, "mesh.ny": 48},
             "rt_multimode": {"mesh.nx": 24, "mesh.ny": 48}}
    out = {}
    for prob, d in cases.items():
        p = Pyro("compressible")
        p.initialize_problem(prob, inputs_dict=d)
        out[prob] = np.array(p.sim.cc_data.data)
        out[prob + "_bc"] = bc_names(p.sim.rp)
        print("ic", prob, out[prob].shape)
    save("comp_problem_ics", **out)


def gen_incompressible_viscous():
    """row f4: incompressible_viscous (lid-driven cavity, moving_lid boundary,
    two Helmholtz solves per step) -- short runs, one instrumented step, and
    the reference regression cavity_n64_Re400_0025.h5 (test.py:111)"""
    import pyro.incompressible.incomp_interface as ii
    import pyro.incompressible.simulation as isim
    store = {}
    orig_pre = isim.Simulation.preevolve

    def preevolve(self):
        store["ic"] = _planes(self.cc_data)
        orig_pre(self)
        store["after_pre"] = _planes(self.cc_data)
    isim.Simulation.preevolve = preevolve
    out = {}
    cases = [("cavity", {"mesh.nx": 16, "mesh.ny": 16}, 5),
             ("cavity", {"mesh.nx": 32, "mesh.ny": 32, "incompressible.proj_type": 1,
                         "incompressible_viscous.viscosity": 0.01}, 4),
             ("shear", {"mesh.nx": 16, "mesh.ny": 16, "incompressible.limiter": 1}, 4)]
    for k, (prob, d, nsteps) in enumerate(cases):
        p = Pyro("incompressible_viscous")
        p.initialize_problem(prob, inputs_dict=d)
        sim = p.sim
        pre = f"i{k}_"
        out[pre + "ic"], out[pre + "after_pre"] = store["ic"], store["after_pre"]
        dts = []
        for _ in range(nsteps):
            p.single_step()
            dts.append(sim.dt)
        out[pre + "final"] = _planes(sim.cc_data)
        out[pre + "dts"] = np.array(dts)
        g = sim.cc_data.grid
        out[pre + "meta"] = np.array([g.nx, g.ng, sim.rp.get_param("incompressible.limiter"),
                                      sim.rp.get_param("incompressible.proj_type"),
                                      sim.rp.get_param("driver.cfl"),
                                      sim.rp.get_param("driver.init_tstep_factor"),
                                      sim.rp.get_param("driver.max_dt_change"),
                                      sim.rp.get_param("incompressible_viscous.viscosity")])
        out[pre + "bc"] = bc_names(sim.rp)
        sim.cc_data.fill_BC_all()
        sim.compute_timestep()
        out[pre + "U0"] = _planes(sim.cc_data)
        out[pre + "dt"] = np.array(sim.dt)
        cap = {}
        orig_states = ii.states

        def states(grid, dt, u, v, a, b, c, dd, gx, gy, u_MAC, v_MAC, sx=None, sy=None):
            cap["umac"], cap["vmac"] = np.array(u_MAC), np.array(v_MAC)
            return orig_states(grid, dt, u, v, a, b, c, dd, gx, gy, u_MAC, v_MAC, sx, sy)
        ii.states = states
        sim.evolve()
        ii.states = orig_states
        out[pre + "umac"], out[pre + "vmac"] = cap["umac"], cap["vmac"]
        out[pre + "U1"] = _planes(sim.cc_data)
        print("viscous case", k, prob, d, "dt", sim.dt)
    out["ncases"] = np.array(len(cases))
    save("incomp_viscous", **out)

    p = Pyro("incompressible_viscous")
    p.initialize_problem("cavity", inputs_file="inputs.cavity")
    ic = store["ic"]
    while not p.sim.finished():
        p.single_step()
    names = ["x-velocity", "y-velocity"]
    with h5py.File(REF + "/incompressible_viscous/tests/cavity_n64_Re400_0025.h5", "r") as f:
        assert int(f.attrs["nsteps"]) == p.sim.n, (f.attrs["nsteps"], p.sim.n)
        gold = np.array([f["state/" + nm + "/data"][...] for nm in names])
        tfin = float(f.attrs["time"])
    run = np.array([np.array(p.sim.cc_data.get_var(nm).v()) for nm in names])
    print("cavity: reference-run vs stored golden, max abs err", np.abs(run - gold).max(axis=(1, 2)))
    isim.Simulation.preevolve = orig_pre
    save("incomp_cavity_0025", ic=ic[:2], gold=gold, run=run, nsteps=np.array(p.sim.n),
         t=np.array(tfin), tmax=np.array(p.sim.tmax), bc=bc_names(p.rp),
         meta=np.array([64, 4, 2, 2, p.rp.get_param("driver.cfl"),
                        p.rp.get_param("driver.init_tstep_factor"),
                        p.rp.get_param("driver.max_dt_change"),
                        p.rp.get_param("incompressible_viscous.viscosity")]))


def _raw_cfl_dt(sim):
    dt_keep, dto_keep = sim.dt, sim.dt_old
    sim.method_compute_timestep()
    raw = sim.dt
    sim.dt, sim.dt_old = dt_keep, dto_keep
    return raw


def gen_compressible_runs():
    # (i) sedov 64^2, 20 steps: SURVEY 8(c) fingerprint
    p = Pyro("compressible")
    p.initialize_problem("sedov", inputs_dict={"mesh.nx": 64, "mesh.ny": 64,
                                               "driver.max_steps": 20})
    ic = np.array(p.sim.cc_data.data)
    dts = []
    while not p.sim.finished():
        p.single_step()
        dts.append(p.sim.dt)
    save("comp_sedov_64_020", ic=ic, final=np.array(p.sim.cc_data.data),
         dts=np.array(dts), meta=comp_meta(p.sim), bc=bc_names(p.sim.rp),
         t=np.array(p.sim.cc_data.t))

    # (ii) reference regression: compressible sod inputs.sod.x (128x10, 76
    # steps) vs pyro/compressible/tests/sod_x_0076.h5 (test.py:101)
    p = Pyro("compressible")
    p.initialize_problem("sod", inputs_file="inputs.sod.x")
    ic = np.array(p.sim.cc_data.data)
    dts = []
    while not p.sim.finished():
        p.single_step()
        dts.append(p.sim.dt)
    gold = {}
    with h5py.File(REF + "/compressible/tests/sod_x_0076.h5", "r") as f:
        assert f.attrs["nsteps"] == p.sim.n
        for nm in ("density", "energy", "x-momentum", "y-momentum"):
            gold[nm] = f["state/" + nm + "/data"][...]
    names = ["density", "energy", "x-momentum", "y-momentum"]
    g = np.stack([gold[nm] for nm in names], axis=-1)
    run = np.stack([np.array(p.sim.cc_data.get_var(nm).v()) for nm in names], axis=-1)
    print("sod_x: reference-run vs stored golden, max abs err", np.abs(run - g).max())
    assert np.abs(run - g).max() < 1e-12
    save("comp_sod_x_0076", ic=ic, gold=g, run=run, dts=np.array(dts),
         meta=comp_meta(p.sim), bc=bc_names(p.sim.rp), tmax=np.array(p.sim.tmax))

    # (iii) reference regression: compressible quad inputs.quad (256^2, 606
    # steps) vs quad_unsplit_0606.h5 (test.py:100).  Too slow to re-run with
    # interpreted njit kernels; we store the reference's IC (its init_data is
    # run here) and its stored golden end state.  The C oracle is then
    # required to reproduce the golden from the IC (tests/test_oracle_golden).
    p = Pyro("compressible")
    p.initialize_problem("quad", inputs_file="inputs.quad")
    ic = np.array(p.sim.cc_data.data)
    with h5py.File(REF + "/compressible/tests/quad_unsplit_0606.h5", "r") as f:
        nsteps = int(f.attrs["nsteps"])
        tfin = float(f.attrs["time"])
        g = np.stack([f["state/" + nm + "/data"][...] for nm in names], axis=-1)
    save("comp_quad_0606", ic=ic.astype(np.float64), gold=g, nsteps=np.array(nsteps),
         t=np.array(tfin), meta=comp_meta(p.sim), bc=bc_names(p.sim.rp),
         tmax=np.array(p.sim.tmax),
         drv=np.array([p.rp.get_param("driver.init_tstep_factor"),
                       p.rp.get_param("driver.max_dt_change")]))


# --------------------------------------------------------------------------
# multigrid (a14-a18)
# --------------------------------------------------------------------------
def gen_mg():
    import pyro.multigrid.examples.mg_test_simple as mts

    # (i) reference regression mg_poisson_dirichlet 256^2 (test.py:138-140)
    nx = 256
    a = MG.CellCenterMG2d(nx, nx, xl_BC_type="dirichlet", yl_BC_type="dirichlet",
                          xr_BC_type="dirichlet", yr_BC_type="dirichlet", verbose=0)
    a.init_zeros()
    rhs = mts.f(a.x2d, a.y2d)
    a.init_RHS(rhs)
    a.solve(rtol=1.e-11)
    v = a.get_solution()
    with h5py.File(REF + "/multigrid/tests/mg_poisson_dirichlet.h5", "r") as f:
        gv = f["state/v/data"][...]
    print("mg 256: reference-run vs stored golden max abs", np.abs(v.v() - gv).max(),
          "cycles", a.num_cycles)
    assert np.abs(v.v() - gv).max() == 0.0
    save("mg_poisson_dirichlet_256", rhs=np.array(rhs), gold=gv,
         ncycles=np.array(a.num_cycles), source_norm=np.array(a.source_norm),
         residual_error=np.array(a.residual_error),
         relative_error=np.array(a.relative_error))

    # (ii) per-operator vectors on 16^2 / 32^2 for every BC flavour
    out = {}
    rng = np.random.default_rng(3)
    k = 0
    for nx in (16, 32):
        for bcs, inhom in ((("dirichlet",) * 4, False), (("dirichlet",) * 4, True),
                           (("neumann",) * 4, False), (("neumann", "neumann", "dirichlet", "dirichlet"), True),
                           (("periodic",) * 4, False),
                           (("periodic", "periodic", "dirichlet", "neumann"), False)):
            for (alpha, beta) in ((0.0, -1.0), (2.5, 0.3)):
                if (alpha, beta) != (0.0, -1.0) and nx != 16:
                    continue
                kw = {}
                if inhom:
                    kw = dict(xl_BC=lambda y: np.sin(y) + 0.2, xr_BC=lambda y: 0.5 * y - 0.1,
                              yl_BC=lambda x: x * x, yr_BC=lambda x: np.cos(3 * x))
                a = MG.CellCenterMG2d(nx, nx, xl_BC_type=bcs[0], xr_BC_type=bcs[1],
                                      yl_BC_type=bcs[2], yr_BC_type=bcs[3],
                                      alpha=alpha, beta=beta, nsmooth=3,
                                      nsmooth_bottom=7, verbose=0, **kw)
                L = a.nlevels - 1
                v0 = rng.standard_normal((nx + 2, nx + 2))
                f0 = rng.standard_normal((nx + 2, nx + 2))
                a.init_solution(v0)
                a.init_RHS(f0)
                pre = f"m{k}_"
                out[pre + "meta"] = np.array([nx, alpha, beta, 3, 7, int(inhom)])
                out[pre + "bc"] = np.array(bcs)
                out[pre + "v0"], out[pre + "f0"] = v0, f0
                if inhom:
                    bc_obj = a.grids[L].BCs["v"]
                    out[pre + "bcvals"] = np.stack([bc_obj.xl_value, bc_obj.xr_value,
                                                    bc_obj.yl_value, bc_obj.yr_value])
                a.smooth(L, 2)
                out[pre + "v_smooth"] = np.array(a.grids[L].get_var("v"))
                a._compute_residual(L)
                out[pre + "r"] = np.array(a.grids[L].get_var("r"))
                out[pre + "rnorm"] = np.array(a.grids[L].get_var("r").norm())
                out[pre + "restrict"] = np.array(a.grids[L].restrict("r"))
                # prolong the (filled) fine v of level L-1 source: use coarse grid data
                cp = a.grids[L - 1]
                cv = cp.get_var("v")
                cv[:, :] = rng.standard_normal(cv.shape)
                cp.fill_BC("v")
                out[pre + "cv"] = np.array(cv)
                out[pre + "prolong"] = np.array(cp.prolong("v"))
                # a full V-cycle and then a solve from a fresh object
                b = MG.CellCenterMG2d(nx, nx, xl_BC_type=bcs[0], xr_BC_type=bcs[1],
                                      yl_BC_type=bcs[2], yr_BC_type=bcs[3],
                                      alpha=alpha, beta=beta, nsmooth=3,
                                      nsmooth_bottom=7, verbose=0, **kw)
                b.init_solution(v0)
                f1 = f0.copy()
                if bcs[0] in ("periodic", "neumann") and alpha == 0.0 and \
                        all(x in ("periodic", "neumann") for x in bcs):
                    f1[1:-1, 1:-1] -= f1[1:-1, 1:-1].mean()   # solvability
                b.init_RHS(f1)
                out[pre + "f1"] = f1
                for lev in range(L):
                    b.grids[lev].zero("v")
                b.v_cycle(L)
                out[pre + "v_vcycle"] = np.array(b.grids[L].get_var("v"))
                b.max_cycles = 6
                b.solve(rtol=1.e-10)
                out[pre + "v_solve"] = np.array(b.grids[L].get_var("v"))
                out[pre + "solve_info"] = np.array([b.num_cycles, b.residual_error,
                                                    b.relative_error, b.source_norm])
                k += 1
    out["ncases"] = np.array(k)
    save("mg_ops", **out)


# --------------------------------------------------------------------------
# diffusion: a caller of MG
# --------------------------------------------------------------------------
def gen_diffusion():
    # (i) reference regression: diffusion gaussian inputs.gaussian (128^2) vs
    # pyro/diffusion/tests/gaussian_0164.h5 (test.py "diffusion gaussian")
    p = Pyro("diffusion")
    p.initialize_problem("gaussian")
    ic = np.array(p.sim.cc_data.get_var("phi"))
    dts = []
    while not p.sim.finished():
        p.single_step()
        dts.append(p.sim.dt)
    with h5py.File(REF + "/diffusion/tests/gaussian_0164.h5", "r") as f:
        gold = f["state/phi/data"][...]
        assert int(f.attrs["nsteps"]) == p.sim.n
    run = np.array(p.sim.cc_data.get_var("phi").v())
    print("diffusion gaussian: reference-run vs stored golden max abs", np.abs(run - gold).max(),
          "steps", p.sim.n)
    save("diff_gaussian_0164", ic=ic, gold=gold, run=run, dts=np.array(dts),
         n=np.array(p.sim.n))
    # (ii) small cases with other boundary types, 6 steps
    out = {}
    cases = [("periodic",) * 4, ("dirichlet",) * 4, ("neumann", "neumann", "periodic", "periodic")]
    for k, b in enumerate(cases):
        p = Pyro("diffusion")
        p.initialize_problem("gaussian", inputs_dict={
            "mesh.nx": 32, "mesh.ny": 32, "driver.max_steps": 6, "driver.cfl": 1.5,
            "mesh.xlboundary": b[0], "mesh.xrboundary": b[1],
            "mesh.ylboundary": b[2], "mesh.yrboundary": b[3], "gaussian.t_0": 0.002})
        out[f"d{k}_ic"] = np.array(p.sim.cc_data.get_var("phi"))
        out[f"d{k}_bc"] = np.array(b)
        while not p.sim.finished():
            p.single_step()
        out[f"d{k}_final"] = np.array(p.sim.cc_data.get_var("phi"))
        out[f"d{k}_dt"] = np.array(p.sim.dt)
    out["ncases"] = np.array(len(cases))
    save("diff_small", **out)


def gen_diffusion_2048(nx=2048, nsteps=2):
    """the diffusion solver at the size its bench leg is timed at: the
    REFERENCE itself (pure NumPy + its multigrid, ~12 s per step), gaussian, 2 steps; a 64 x 64
    lattice of phi, row / column sums, dt (tests/fullsize_ics.py: lattice)"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    from fullsize_ics import lattice
    p = Pyro("diffusion")
    p.initialize_problem("gaussian", inputs_dict={"mesh.nx": nx, "mesh.ny": nx, "driver.max_steps": nsteps})
    dts = []
    while not p.sim.finished():
        p.single_step()
        dts.append(p.sim.dt)
    phi = np.array(p.sim.cc_data.get_var("phi").v())
    save(f"diff_gaussian_{nx}", dts=np.array(dts), nsteps=np.array(p.sim.n), **lattice(phi[:, :, None]))


# --------------------------------------------------------------------------
# variable-coefficient multigrid.  The reference's stored
# goldens mg_vc_poisson_*.h5 are missing from this checkout
# (.MISSING_LARGE_BLOBS), so the reference itself is run.
# --------------------------------------------------------------------------
def gen_mg_vc():
    import pyro.multigrid.variable_coeff_MG as VC
    out = {}
    cases = [(("dirichlet",) * 4, ("neumann",) * 4, 32),
             (("periodic",) * 4, ("periodic",) * 4, 32),
             (("neumann", "dirichlet", "periodic", "periodic"),
              ("neumann", "neumann", "periodic", "periodic"), 16)]
    rng = np.random.default_rng(5)
    for k, (bcs, cbcs, nx) in enumerate(cases):
        g = patch.Grid2d(nx, nx, ng=1)
        d = patch.CellCenterData2d(g)
        bc_c = bnd.BC(xlb=cbcs[0], xrb=cbcs[1], ylb=cbcs[2], yrb=cbcs[3])
        d.register_var("c", bc_c)
        d.create()
        c = d.get_var("c")
        c[:, :] = 2.0 + np.cos(2.0 * np.pi * g.x2d) * np.cos(2.0 * np.pi * g.y2d) + \
            0.1 * rng.random(c.shape)
        a = VC.VarCoeffCCMG2d(nx, nx, xl_BC_type=bcs[0], xr_BC_type=bcs[1],
                              yl_BC_type=bcs[2], yr_BC_type=bcs[3], nsmooth=4,
                              nsmooth_bottom=9, coeffs=c, coeffs_bc=bc_c, verbose=0)
        L = a.nlevels - 1
        pre = f"v{k}_"
        out[pre + "bc"] = np.array(bcs)
        out[pre + "cbc"] = np.array(cbcs)
        out[pre + "nx"] = np.array(nx)
        out[pre + "c"] = np.array(c)
        for lev in (L, L - 1, 0):
            out[pre + f"c_l{lev}"] = np.array(a.grids[lev].get_var("coeffs"))
            out[pre + f"ex_l{lev}"] = np.array(a.edge_coeffs[lev].x)
            out[pre + f"ey_l{lev}"] = np.array(a.edge_coeffs[lev].y)
        v0 = rng.standard_normal((nx + 2, nx + 2))
        f0 = rng.standard_normal((nx + 2, nx + 2))
        if all(b in ("periodic", "neumann") for b in bcs):
            f0[1:-1, 1:-1] -= f0[1:-1, 1:-1].mean()
        out[pre + "v0"], out[pre + "f0"] = v0, f0
        a.init_solution(v0)
        a.init_RHS(f0)
        a.smooth(L, 3)
        out[pre + "v_smooth"] = np.array(a.grids[L].get_var("v"))
        a._compute_residual(L)
        out[pre + "r"] = np.array(a.grids[L].get_var("r"))
        b = VC.VarCoeffCCMG2d(nx, nx, xl_BC_type=bcs[0], xr_BC_type=bcs[1],
                              yl_BC_type=bcs[2], yr_BC_type=bcs[3], nsmooth=4,
                              nsmooth_bottom=9, coeffs=c, coeffs_bc=bc_c, verbose=0)
        b.init_solution(v0)
        b.init_RHS(f0)
        b.max_cycles = 5
        b.solve(rtol=1.e-10)
        out[pre + "v_solve"] = np.array(b.grids[L].get_var("v"))
        out[pre + "info"] = np.array([b.num_cycles, b.residual_error, b.relative_error,
                                      b.source_norm])
    out["ncases"] = np.array(len(cases))
    save("mg_vc", **out)


def gen_mg_general():
    """row f1: general_MG.GeneralMG2d (alpha phi + div(beta grad phi) +
    gamma . grad phi = f): coefficient hierarchy, smoother, residual, solve"""
    import pyro.multigrid.general_MG as GM
    out = {}
    cases = [(("dirichlet",) * 4, ("neumann",) * 4, 32),
             (("periodic",) * 4, ("periodic",) * 4, 32),
             (("neumann", "dirichlet", "periodic", "periodic"),
              ("neumann", "neumann", "periodic", "periodic"), 16)]
    rng = np.random.default_rng(11)
    for k, (bcs, cbcs, nx) in enumerate(cases):
        g = patch.Grid2d(nx, nx, ng=1)
        d = patch.CellCenterData2d(g)
        bc_c = bnd.BC(xlb=cbcs[0], xrb=cbcs[1], ylb=cbcs[2], yrb=cbcs[3])
        for nm in ("alpha", "beta", "gamma_x", "gamma_y"):
            d.register_var(nm, bc_c)
        d.create()
        X, Y = 2.0 * np.pi * g.x2d, 2.0 * np.pi * g.y2d
        d.get_var("alpha")[:, :] = -(8.0 + np.cos(X) + 0.1 * rng.random(X.shape))
        d.get_var("beta")[:, :] = 2.0 + np.cos(X) * np.cos(Y) + 0.1 * rng.random(X.shape)
        d.get_var("gamma_x")[:, :] = 0.5 * np.sin(X) + 0.05 * rng.random(X.shape)
        d.get_var("gamma_y")[:, :] = 0.5 * np.sin(Y) + 0.05 * rng.random(X.shape)

        def make():
            return GM.GeneralMG2d(nx, nx, xl_BC_type=bcs[0], xr_BC_type=bcs[1],
                                  yl_BC_type=bcs[2], yr_BC_type=bcs[3], nsmooth=4,
                                  nsmooth_bottom=9, coeffs=d, verbose=0)
        a = make()
        L = a.nlevels - 1
        pre = f"g{k}_"
        out[pre + "bc"], out[pre + "cbc"], out[pre + "nx"] = np.array(bcs), np.array(cbcs), np.array(nx)
        for nm in ("alpha", "beta", "gamma_x", "gamma_y"):
            out[pre + nm] = np.array(d.get_var(nm))
        for lev in (L, L - 1, 0):
            for nm in ("alpha", "gamma_x", "gamma_y"):
                out[pre + f"{nm}_l{lev}"] = np.array(a.grids[lev].get_var(nm))
            out[pre + f"ex_l{lev}"] = np.array(a.beta_edge[lev].x)
            out[pre + f"ey_l{lev}"] = np.array(a.beta_edge[lev].y)
        v0 = rng.standard_normal((nx + 2, nx + 2))
        f0 = rng.standard_normal((nx + 2, nx + 2))
        out[pre + "v0"], out[pre + "f0"] = v0, f0
        a.init_solution(v0)
        a.init_RHS(f0)
        a.smooth(L, 3)
        out[pre + "v_smooth"] = np.array(a.grids[L].get_var("v"))
        a._compute_residual(L)
        out[pre + "r"] = np.array(a.grids[L].get_var("r"))
        b = make()
        b.init_solution(v0)
        b.init_RHS(f0)
        b.max_cycles = 5
        b.solve(rtol=1.e-10)
        out[pre + "v_solve"] = np.array(b.grids[L].get_var("v"))
        out[pre + "info"] = np.array([b.num_cycles, b.residual_error, b.relative_error,
                                      b.source_norm])
        print("general MG case", k, bcs, "cycles", b.num_cycles, "res", b.residual_error)
    out["ncases"] = np.array(len(cases))
    save("mg_general", **out)


def gen_burgers_viscous():
    """burgers_viscous (another multigrid caller: unsplit Burgers fluxes + one
    Crank-Nicolson Helmholtz solve per velocity component): short runs of its
    three problems and one more step from the end state"""
    cases = [
        ("converge", "inputs.converge.32", {"mesh.nx": 16, "mesh.ny": 16}, 5),
        ("test", None, {"mesh.nx": 32, "mesh.ny": 32, "driver.max_dt_change": 2.0,
                        "driver.init_tstep_factor": 0.5}, 5),
        ("tophat", None, {"mesh.nx": 16, "mesh.ny": 16, "advection.limiter": 1,
                          "diffusion.eps": 0.01}, 6),
    ]
    out = {"ncases": np.array(len(cases))}
    for k, (prob, inp, d, nsteps) in enumerate(cases):
        p = Pyro("burgers_viscous")
        import importlib
        mod = importlib.import_module(f"pyro.burgers_viscous.problems.{prob}")
        if not hasattr(mod, "PROBLEM_PARAMS"):
            # shipped without it (converge, tophat): register init_data by hand
            p.add_problem(prob, mod.init_data)
        p.initialize_problem(prob, inputs_file=inp or f"inputs.{prob}",
                             inputs_dict=dict(d, **{"particles.do_particles": 0}))
        sim = p.sim
        pre = f"v{k}_"
        out[pre + "ic"] = np.moveaxis(np.array(sim.cc_data.data), -1, 0)
        dts = []
        for _ in range(nsteps):
            p.single_step()
            dts.append(sim.dt)
        out[pre + "dts"] = np.array(dts)
        out[pre + "final"] = np.moveaxis(np.array(sim.cc_data.data), -1, 0)
        sim.cc_data.fill_BC_all()
        sim.compute_timestep()
        out[pre + "U0"] = np.moveaxis(np.array(sim.cc_data.data), -1, 0)
        out[pre + "dt"] = np.array(sim.dt)
        sim.evolve()
        out[pre + "U1"] = np.moveaxis(np.array(sim.cc_data.data), -1, 0)
        g = sim.cc_data.grid
        out[pre + "meta"] = np.array([g.nx, g.ng, sim.rp.get_param("advection.limiter"),
                                      sim.rp.get_param("diffusion.eps"),
                                      sim.rp.get_param("driver.cfl"),
                                      sim.rp.get_param("driver.init_tstep_factor"),
                                      sim.rp.get_param("driver.max_dt_change"),
                                      sim.rp.get_param("driver.fix_dt"),
                                      sim.rp.get_param("driver.tmax")])
        out[pre + "bc"] = bc_names(sim.rp)
        out[pre + "problem"] = np.array(prob)
        print("burgers_viscous case", k, prob, "dt", sim.dt, "max u", np.abs(out[pre + "U1"][0]).max())
    save("burgers_viscous", **out)


def sph_geometry(g):
    """the arrays of patch.SphericalPolar + the sines artificial_viscosity
    evaluates (interface.py:345-347), with the reference's own expressions"""
    out = {n: np.array(getattr(g, n)) for n in ("Lx", "Ly", "Ax", "Ay", "V", "dlogAx", "dlogAy",
                                                 "x2d", "y2d")}
    j = np.arange(g.qy)
    out["sint"] = np.array([np.sin((jj + 0.5 - g.ng) * g.dy + g.ymin) for jj in j])
    out["sinb"] = np.array([np.sin((jj - 0.5 - g.ng) * g.dy + g.ymin) for jj in j])
    out["sinc"] = np.array([np.sin((jj - g.ng) * g.dy + g.ymin) for jj in j])
    out["domain"] = np.array([g.xmin, g.xmax, g.ymin, g.ymax])
    return out


def gen_compressible_spherical():
    """row f4: the compressible solver on a SphericalPolar grid (mesh.grid_type
    = SphericalPolar; CGF solver): short runs of the reference's two spherical
    set-ups and the stages of one more step"""
    cases = [
        ("sedov", "inputs.sedov.spherical", {"mesh.nx": 48, "mesh.ny": 24}, 8),
        ("advect", "inputs.advect.spherical.64", {"mesh.nx": 40, "mesh.ny": 20}, 6),
        ("sedov", "inputs.sedov.spherical", {"mesh.nx": 40, "mesh.ny": 16, "compressible.grav": -0.3,
                                             "compressible.limiter": 1, "mesh.ymin": 0.4,
                                             "mesh.ymax": 1.2}, 7),
    ]
    out = {"ncases": np.array(len(cases))}
    for k, (prob, inp, d, nsteps) in enumerate(cases):
        p = Pyro("compressible")
        p.initialize_problem(prob, inputs_file=inp, inputs_dict=d)
        sim = p.sim
        pre = f"c{k}_"
        out[pre + "ic"] = np.array(sim.cc_data.data)
        dts = []
        for _ in range(nsteps):
            p.single_step()
            dts.append(sim.dt)
        out[pre + "dts"] = np.array(dts)
        out[pre + "after"] = np.array(sim.cc_data.data)
        sim.cc_data.fill_BC_all()
        sim.compute_timestep()
        out[pre + "meta"] = comp_meta(sim)
        out[pre + "bc"] = bc_names(sim.rp)
        out[pre + "problem"] = np.array(prob)
        out[pre + "dt"] = np.array(sim.dt)
        out[pre + "drv"] = np.array([p.rp.get_param("driver.init_tstep_factor"),
                                     p.rp.get_param("driver.max_dt_change")])
        for nm, a in sph_geometry(sim.cc_data.grid).items():
            out[pre + "g_" + nm] = a
        st = comp_stage_dump(sim)
        for nm in ("U0", "q", "ldx", "Uxl0", "Uxr0", "Uyl0", "Uyr0", "FxT", "FyT", "Uxl", "Uxr",
                   "Uyl", "Uyr", "Fx0", "Fy0", "avx", "avy", "Fx", "Fy", "U1"):
            out[pre + nm] = st[nm]
        print("spherical case", k, prob, "dt", sim.dt, type(sim.cc_data.grid).__name__)
    save("comp_spherical", **out)


if __name__ == "__main__":
    if "burgers_viscous" in sys.argv[1:]:
        gen_burgers_viscous()
    if "comp_spherical" in sys.argv[1:]:
        gen_compressible_spherical()
    if "mg_general" in sys.argv[1:]:
        gen_mg_general()
    if "comp_lm" in sys.argv[1:]:
        gen_compressible_lm()
    if "comp_ramp" in sys.argv[1:]:
        gen_compressible_ramp()
    if "comp_heating" in sys.argv[1:]:
        gen_compressible_heating()
    if "regressions2" in sys.argv[1:]:
        gen_regressions2()
    if "mesh_utils" in sys.argv[1:]:
        gen_mesh_utils()
    if "comp_general_source" in sys.argv[1:]:
        gen_compressible_general_source()
    if "problem_ics" in sys.argv[1:]:
        gen_problem_ics()
    if "incomp_viscous" in sys.argv[1:]:
        gen_incompressible_viscous()
    if "mg_vc" in sys.argv[1:]:
        gen_mg_vc()
    if "comp_f2" in sys.argv[1:]:
        gen_compressible_f2()
    if "comp_hse" in sys.argv[1:]:
        gen_compressible_hse()
    if "comp_rt" in sys.argv[1:]:
        gen_compressible_rt()
    if "incomp" in sys.argv[1:]:
        gen_incompressible()
    if "comp_rk" in sys.argv[1:]:
        gen_compressible_rk()
    if "swe" in sys.argv[1:]:
        gen_swe()
    if "mg_4096" in sys.argv[1:]:
        gen_mg_4096()
    if "diff_2048" in sys.argv[1:]:
        gen_diffusion_2048()
    which = sys.argv[1:] or ["bc", "adv", "comp_stages", "comp_runs", "mg", "diffusion"]
    if "diffusion" in which:
        gen_diffusion()
    if "bc" in which:
        gen_fill_bc()
    if "adv" in which:
        gen_advection()
    if "comp_stages" in which:
        gen_compressible_stages()
    if "comp_runs" in which:
        gen_compressible_runs()
    if "mg" in which:
        gen_mg()
