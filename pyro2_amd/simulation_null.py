"""Solver-independent Simulation base class and grid / BC set-up helpers with
the surface of pyro/simulation_null.py:10-300.  Host side: this is the caller
of the device hot path, its time-step policy (compute_timestep) is part of the
boundary that must be preserved exactly."""
import numpy as np

from .mesh import boundary as bnd
from .mesh import patch
from .util import msg
from .util import profile_pyro as profile


def _param(rp, key, default, what):
    try:
        return rp.get_param(key)
    except KeyError:
        msg.warning(f"{key} not set, defaulting to {what}")
        return default


_warned_replicas = False


def grid_setup(rp, ng=1, spherical_ok=False, decomposable=False):
    """build the Grid2d described by the [mesh] parameters
    (simulation_null.py:10-69).  spherical_ok: the calling solver has the
    geometry terms (only the compressible solver does, as in the reference).
    decomposable: the calling solver steps x-slabs (compressible, advection): with one process
    per GPU (decomp.active_decomposition: a launcher's RANK / WORLD_SIZE, gpu.decompose) the
    grid handed out is THIS rank's slab of the [mesh] grid -- global xmin / xmax / dx, local nx,
    the coordinates of its rows of the whole grid (mesh/patch.py Grid2d(slab=...))"""
    nx = rp.get_param("mesh.nx")
    ny = rp.get_param("mesh.ny")
    xmin = _param(rp, "mesh.xmin", 0.0, "0.0")
    xmax = _param(rp, "mesh.xmax", 1.0, "1.0")
    ymin = _param(rp, "mesh.ymin", 0.0, "0.0")
    ymax = _param(rp, "mesh.ymax", 1.0, "1.0")
    grid_type = _param(rp, "mesh.grid_type", "Cartesian2d", "Cartesian2D")
    if grid_type == "SphericalPolar":
        if not spherical_ok:
            raise ValueError("mesh.grid_type = SphericalPolar is implemented by the "
                             "compressible solver only")
        return patch.SphericalPolar(nx, ny, xmin=xmin, xmax=xmax, ymin=ymin, ymax=ymax, ng=ng)
    if grid_type != "Cartesian2d":
        raise ValueError("Unsupported grid type!")
    from . import decomp
    dec = decomp.active_decomposition(rp)
    if dec is not None and not decomposable:
        global _warned_replicas
        if not _warned_replicas:
            _warned_replicas = True
            msg.warning(f"{dec.nranks} processes: this solver is not domain-decomposed, every "
                        "process runs the whole problem")
        dec = None
    slab = None
    if dec is not None:
        xlb = _param(rp, "mesh.xlboundary", "periodic", "periodic")
        slab = decomp.SlabDecomp(nx, dec.nranks, dec.rank, periodic=(xlb == "periodic"))
    return patch.Cartesian2d(nx, ny, xmin=xmin, xmax=xmax, ymin=ymin, ymax=ymax, ng=ng, slab=slab)


def bc_setup(rp):
    """(bc, bc_xodd, bc_yodd): scalars reflect evenly, the normal velocity
    component oddly (simulation_null.py:72-112)"""
    sides = [_param(rp, "mesh." + k, "periodic", "periodic")
             for k in ("xlboundary", "xrboundary", "ylboundary", "yrboundary")]
    kw = dict(xlb=sides[0], xrb=sides[1], ylb=sides[2], yrb=sides[3])
    return (bnd.BC(**kw), bnd.BC(odd_reflect_dir="x", **kw), bnd.BC(odd_reflect_dir="y", **kw))


class NullSimulation:
    def __init__(self, solver_name, problem_name, problem_func, rp, *,
                 problem_finalize_func=None, problem_source_func=None,
                 timers=None, data_class=patch.CellCenterData2d):
        self.n = 0
        self.dt = -1.e33
        self.dt_old = -1.e33
        self.data_class = data_class
        self.rp = rp
        self.tmax = self._opt("driver.tmax")
        self.max_steps = self._opt("driver.max_steps")
        self.cc_data = None
        self.particles = None
        self.SMALL = 1.e-12
        self.solver_name = solver_name
        self.problem_name = problem_name
        self.problem_func = problem_func
        self.problem_finalize = problem_finalize_func
        self.problem_source = problem_source_func
        self.tc = timers if timers is not None else profile.TimerCollection()
        v = self._opt("driver.verbose")
        self.verbose = 0 if v is None else v
        self.n_num_out = 0
        self.cm = "viridis"

    def _opt(self, key):
        try:
            return self.rp.get_param(key)
        except (AttributeError, KeyError):
            return None

    def _rp_opt(self, key, default):
        """a runtime parameter that a hand-built RuntimeParameters (the reference's
        unit tests) may not carry"""
        v = self._opt(key)
        return default if v is None else v

    def __str__(self):
        return f"pyro Simulation:\n  solver: {self.solver_name}\n  problem: {self.problem_name}\n"

    # ---- tracer particles (pyro/particles; host-side diagnostic) ---------
    def setup_particles(self, bc):
        """particles.do_particles = 1: seed the tracers the way every solver's
        initialize() does (e.g. advection/simulation.py:30-33)"""
        if self.rp.get_param("particles.do_particles") == 1:
            if getattr(self.cc_data, "slab", None) is not None:
                msg.fail("ERROR: tracer particles are not carried by a decomposed run")
            from .particles import particles
            self.particles = particles.Particles(
                self.cc_data, bc, self.rp.get_param("particles.n_particles"),
                self.rp.get_param("particles.particle_generator"))

    def advance_particles(self, u=None, v=None):
        """move the tracers over self.dt with the cell-centred velocity: the
        arrays given, else the stored x-/y-velocity, else the derived
        "velocity" (read from the device once, only when particles exist)"""
        if self.particles is None:
            return
        cc = self.cc_data
        if u is None and v is None:
            if "x-velocity" in cc.names:
                u, v = cc.get_var_readonly("x-velocity"), cc.get_var_readonly("y-velocity")
            else:
                u, v = cc.get_var_readonly("velocity")
        self.particles.update_particles(self.dt, u, v)

    def finished(self):
        return self.cc_data.t >= self.tmax or self.n >= self.max_steps

    def do_output(self):
        dt_out = self.rp.get_param("io.dt_out")
        n_out = self.rp.get_param("io.n_out")
        do_io = self.rp.get_param("io.do_io")
        due = self.cc_data.t >= (self.n_num_out + 1) * dt_out or self.n % n_out == 0
        if due and do_io == 1:
            self.n_num_out += 1
            return True
        return False

    def initialize(self):
        pass

    def method_compute_timestep(self):
        """solver specific CFL step -> self.dt"""

    def compute_timestep(self):
        """driver policy around the solver's CFL step: fixed dt, small first
        step, bounded growth, and landing exactly on tmax
        (simulation_null.py:222-244)"""
        init_tstep_factor = self.rp.get_param("driver.init_tstep_factor")
        max_dt_change = self.rp.get_param("driver.max_dt_change")
        fix_dt = self.rp.get_param("driver.fix_dt")
        if fix_dt > 0.0:
            self.dt = fix_dt
        else:
            self.method_compute_timestep()
            if self.n == 0:
                self.dt = init_tstep_factor * self.dt
            else:
                self.dt = min(max_dt_change * self.dt_old, self.dt)
            self.dt_old = self.dt
        if self.cc_data.t + self.dt > self.tmax:
            self.dt = self.tmax - self.cc_data.t

    def preevolve(self):
        pass

    def evolve(self):
        self.cc_data.t += self.dt
        self.n += 1

    def dovis(self):
        pass

    def finalize(self):
        if self.problem_finalize:
            self.problem_finalize()

    def write(self, filename):
        """HDF5 dump in the reference layout (simulation_null.py:270-290);
        this is a device -> host synchronisation point"""
        from .util import h5lite
        cc = self.cc_data
        if getattr(cc, "slab", None) is not None:
            # COLLECTIVE: the slabs are gathered, rank 0 writes the ONE file of the reference's
            # layout (state/<var>/data is the whole nx x ny array, patch.py:750-788)
            if self.particles is not None:
                msg.fail("ERROR: tracer particles are not carried by a decomposed run")
            whole = cc.gather()
            if whole is None:
                return
            keep, self.cc_data = self.cc_data, whole
            try:
                self._write_one(filename)
            finally:
                self.cc_data = keep
            return
        self._write_one(filename)

    def _write_one(self, filename):
        from .util import h5lite
        with h5lite.open_file(filename, "w") as f:
            f.attrs["solver"] = self.solver_name
            f.attrs["problem"] = self.problem_name
            f.attrs["time"] = self.cc_data.t
            f.attrs["nsteps"] = self.n
            # not in the reference's files: lets a restart reproduce the
            # max_dt_change limiter of the next step exactly
            f.attrs["dt"] = self.dt
            f.attrs["dt_old"] = self.dt_old
            try:      # which arithmetic wrote the file (0: bit-faithful, 1: contracted)
                f.attrs["gpu_fast_math"] = int(self.rp.get_param("gpu.fast_math"))
            except (KeyError, RuntimeError, ValueError):
                pass
            self.cc_data.write_data(f)
            if self.particles is not None:
                self.particles.write_particles(f)
            self.rp.write_params(f)
            self.write_extras(f)

    def write_extras(self, f):
        pass

    def read_extras(self, f):
        pass
