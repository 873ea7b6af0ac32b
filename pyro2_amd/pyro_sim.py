#!/usr/bin/env python3
"""Driver of the device solvers, with the call surface of pyro's own driver
(pyro/pyro_sim.py:33-456): `Pyro(solver)`, `.initialize_problem(problem, ...)`,
`.run_sim()`, `.single_step()`, `PyroBenchmark`, and the command line
`pyro_sim.py solver problem inputs [section.option=value ...]`.

What a step is (pyro_sim.py:241-281): boundary fill -> time step -> evolve.  All three
run on the GPU; dt (8 bytes) is what comes back.  Beyond the reference: user problems
(`add_problem`), restart from an output file (`restart_problem`), and batches of
steps handed to the device at once when nothing is printed, plotted or written in
between (`Simulation.evolve_many`).
"""
import argparse
import functools
import importlib
import os
import sys
from collections import namedtuple

from .util import msg
from .util import profile_pyro as profile
from .util.runparams import RuntimeParameters, _get_val

# the solvers this package implements on the device; the reference's
# other solvers are not in scope
valid_solvers = ["advection", "burgers", "compressible", "compressible_rk", "diffusion", "swe",
                 "incompressible", "incompressible_viscous", "burgers_viscous"]
# a solver that keeps its inputs files in another solver's problem directory
problem_home = {"compressible_rk": "compressible"}

_PACKAGE = __package__ or "pyro2_amd"
_HERE = os.path.dirname(os.path.realpath(__file__)) + "/"

# what a problem contributes: the initial condition, its parameters, optional hooks
_Problem = namedtuple("_Problem", "init params finalize source heating inputs")


def _needs_problem(method):
    """methods that only make sense once initialize_problem() has run"""
    @functools.wraps(method)
    def guarded(self, *a, **kw):
        if not self.is_initialized:
            msg.fail("ERROR: problem has not been initialized")
        return method(self, *a, **kw)
    return guarded


class Pyro:
    def __init__(self, solver_name, *, from_commandline=False):
        self.from_commandline = from_commandline
        if from_commandline:
            msg.bold("pyro ...")
        name = solver_name[len(_PACKAGE) + 1:] if solver_name.startswith(_PACKAGE + ".") else solver_name
        if name not in valid_solvers:
            msg.fail(f"ERROR: {name} is not a valid solver")
        self.solver_name = name
        self.pyro_home = _HERE
        self.solver = importlib.import_module(f"{_PACKAGE}.{name}")
        self.custom_problems = {}
        self.problem_name = None
        self._problem = None
        self.sim = None
        self.is_initialized = False
        self._quiet = False
        self.tc = profile.TimerCollection()
        # defaults of the package, then of the solver
        self.rp = RuntimeParameters()
        for defaults in (_HERE + "_defaults", f"{_HERE}{name}/_defaults"):
            self.rp.load_params(defaults)

    # ---- problems ---------------------------------------------------------------------

    def add_problem(self, name, problem_func, *, problem_params=None):
        """register a user problem: problem_func(cc_data, rp) fills the state"""
        self.custom_problems[name] = (problem_func, problem_params or {})

    def _find_problem(self, problem_name):
        if problem_name in self.custom_problems:
            init, params = self.custom_problems[problem_name]
            return _Problem(init, params, None, None, None, None)
        mod = importlib.import_module(f"{_PACKAGE}.{self.solver_name}.problems.{problem_name}")
        source = getattr(mod, "source_terms", None)
        # A source of the form S[energy] = rho * e_rate * profile(x, y) (the reference's
        # three heating problems) runs on the device: the problem module then also has
        # heating_profile(grid, rp) -> (e_rate, profile).  Any other source_terms() is
        # evaluated on the host twice per step by the compressible solver
        # (Simulation._evolve_host_source); the other solvers refuse it.
        heating = None
        if source is not None:
            heating = getattr(sys.modules.get(source.__module__), "heating_profile", None)
            if heating is None and self.solver_name != "compressible":
                msg.fail("ERROR: this problem's source_terms() has no heating_profile() "
                         "companion; only sources of the form rho * e_rate * profile(x, y) "
                         "are carried by the device path")
        return _Problem(mod.init_data, mod.PROBLEM_PARAMS, mod.finalize, source, heating,
                        mod.DEFAULT_INPUTS)

    def _inputs_path(self, inputs_file):
        """the file as given, or the one of that name in the solver's problem directory"""
        if os.path.isfile(inputs_file):
            return inputs_file
        home = problem_home.get(self.solver_name, self.solver_name)
        shipped = f"{self.pyro_home}{home}/problems/{inputs_file}"
        if not os.path.isfile(shipped):
            msg.fail("ERROR: inputs file does not exist")
        return shipped

    # properties the solvers / older callers read
    problem_func = property(lambda self: self._problem.init if self._problem else None)
    problem_params = property(lambda self: self._problem.params if self._problem else None)
    problem_finalize = property(lambda self: self._problem.finalize if self._problem else None)
    problem_source = property(lambda self: self._problem.source if self._problem else None)
    problem_heating = property(lambda self: self._problem.heating if self._problem else None)

    def initialize_problem(self, problem_name, *, inputs_file=None, inputs_dict=None):
        prob = self._find_problem(problem_name)
        self._problem, self.problem_name = prob, problem_name
        # parameter layers, weakest first: problem defaults, inputs file, library-use
        # quieting, explicit overrides
        for key, value in prob.params.items():
            self.rp.set_param(key, value, no_new=False)
        chosen = inputs_file if inputs_file is not None else prob.inputs
        if chosen is not None:
            self.rp.load_params(self._inputs_path(chosen), no_new=1)
        if not self.from_commandline:       # notebook / library use: no output, no windows
            for key in ("vis.dovis", "driver.verbose", "io.do_io"):
                self.rp.set_param(key, 0)
        for key, value in (inputs_dict or {}).items():
            self.rp.set_param(key, value)
        # one process per GPU (a launcher's RANK / WORLD_SIZE, gpu.decompose): every process
        # builds the Simulation of ITS x-slab of the one problem (simulation_null.grid_setup);
        # rank 0 speaks and writes for all of them
        from . import decomp
        dec = decomp.active_decomposition(self.rp)
        self._quiet = dec is not None and dec.rank != 0
        if not self._quiet:
            self.rp.print_paramfile()
        self.verbose = self.rp.get_param("driver.verbose")
        self.dovis = self.rp.get_param("vis.dovis")

        sim = self.solver.Simulation(self.solver_name, problem_name, prob.init, self.rp,
                                     problem_finalize_func=prob.finalize,
                                     problem_source_func=prob.source, timers=self.tc)
        sim.problem_heating = prob.heating
        sim.initialize()
        sim.preevolve()
        sim.cc_data.t = 0.0
        self.sim = sim
        self.is_initialized = True

    def restart_problem(self, filename, *, inputs_dict=None):
        """continue a run from an output file written by Simulation.write()
        (the reference has no restart path).  The runtime
        parameters stored in the file are re-applied (inputs_dict overrides,
        e.g. driver.max_steps / driver.tmax), the problem is set up as usual
        and the interior of every variable, the time, the step count and the
        previous time step are taken from the file.  Ghost cells are refilled
        by the first step, so with the standard boundary types the continued
        run is bit-identical to the uninterrupted one."""
        from .util import io_pyro

        def plain(x):
            x = x.decode() if isinstance(x, bytes) else x
            return x.item() if hasattr(x, "item") else x

        chk = io_pyro.read(filename)
        info = getattr(chk, "restart_info", None)
        if info is None or chk.solver_name != self.solver_name:
            msg.fail(f"ERROR: {filename} is not an output file of the {self.solver_name} solver")
        params = {k: plain(v) for k, v in info["params"].items()}
        params.update(inputs_dict or {})
        # parameters only the problem module knows must exist before they can be overridden
        for key, value in params.items():
            try:
                self.rp.get_param(key)
            except (KeyError, RuntimeError):
                self.rp.set_param(key, value, no_new=False)
        self.initialize_problem(plain(chk.problem_name), inputs_dict=params)

        new, old = self.sim.cc_data, chk.cc_data
        if (new.grid.nx, new.grid.ny) != (old.grid.nx, old.grid.ny):
            msg.fail("ERROR: the grid of the restart file does not match the inputs")
        for var in old.names:
            new.get_var(var).v()[:, :] = old.get_var(var).v()
        new.t = float(old.t)
        self.sim.n = int(chk.n)
        if info["dt"] is None:      # a file written by pyro itself carries no dt history
            self.sim.n = max(self.sim.n, 1)
            self.sim.dt_old = 1.e33
        else:
            self.sim.dt, self.sim.dt_old = float(info["dt"]), float(info["dt_old"])
        # do_output()'s counter (simulation_null.py:270-290) would otherwise call every
        # step after the restart "due" until it has caught up
        every = self.rp.get_param("io.dt_out")
        if every > 0.0:
            self.sim.n_num_out = int(new.t / every)

    # ---- running ----------------------------------------------------------------------

    def _output_name(self):
        return "%s%04d" % (self.rp.get_param("io.basename"), self.sim.n)

    def _write_output(self, announce=True):
        if announce and self.verbose > 0 and not self._quiet:
            msg.warning("outputting...")
        self.sim.write(self._output_name())

    def _can_batch(self, writing):
        """nothing happens between the steps: the device may run several in a row"""
        if self.verbose > 0 or self.dovis or writing:
            return False
        probe = getattr(self.sim, "can_evolve_many", None)
        return bool(probe and probe())

    @_needs_problem
    def run_sim(self):
        clock = self.tc.timer("main")
        clock.begin()
        writing = self.rp.get_param("io.do_io")
        if writing:
            self._write_output(announce=False)
        if self.dovis:
            import matplotlib.pyplot as plt
            plt.ion()
            plt.figure(num=1, figsize=(8, 6), dpi=100, facecolor="w")
            self.sim.dovis()
        while not self.sim.finished():
            if self._can_batch(writing):
                batch = getattr(self.sim, "batch_steps", 64)
                if not len(self.sim.evolve_many(min(batch, self.sim.max_steps - self.sim.n))):
                    # nothing the device would take (a zero time step, a refusal of the
                    # batched path): the plain step, which always advances n
                    self.single_step()
            else:
                self.single_step()
        if writing or self.rp.get_param("io.force_final_output"):
            self._write_output()
        clock.end()
        if self.verbose > 0 and not self._quiet:
            self.rp.print_unused_params()
            self.tc.report()
        if not self._quiet:
            self.sim.finalize()

    @_needs_problem
    def single_step(self):
        sim = self.sim
        sim.cc_data.fill_BC_all()       # the step of pyro_sim.py:250-256, in its order
        sim.compute_timestep()
        sim.evolve()
        if self.verbose > 0 and not self._quiet:
            print("%5d %10.5f %10.5f" % (sim.n, sim.cc_data.t, sim.dt))
        if sim.do_output():
            self._write_output()
        if self.dovis:
            clock = self.tc.timer("vis")
            clock.begin()
            sim.dovis()
            if self.rp.get_param("vis.store_images") == 1:
                import matplotlib.pyplot as plt
                plt.savefig(self._output_name() + ".png")
            clock.end()

    # ---- inspection -------------------------------------------------------------------

    def __repr__(self):
        return f"Pyro('{self.solver_name}')"

    def __str__(self):
        head = [f"Solver = {self.solver_name}"]
        if self.is_initialized:
            head += [f"Problem = {self.sim.problem_name}",
                     f"Simulation time = {self.sim.cc_data.t}",
                     f"Simulation step number = {self.sim.n}"]
        return "\n".join(head) + "\n\nRuntime Parameters\n------------------\n" + str(self.rp)

    @_needs_problem
    def get_var(self, v):
        return self.sim.cc_data.get_var(v)

    @_needs_problem
    def get_grid(self):
        return self.sim.cc_data.grid

    def get_sim(self):
        return self.sim


class PyroBenchmark(Pyro):
    """a Pyro whose end state is compared with / stored as a benchmark file
    (pyro/pyro_sim.py:322-408)"""

    def __init__(self, solver_name, *, comp_bench=False, reset_bench_on_fail=False,
                 make_bench=False, bench_dir=None):
        super().__init__(solver_name)
        self.comp_bench, self.make_bench = comp_bench, make_bench
        self.reset_bench_on_fail = reset_bench_on_fail
        self.bench_dir = bench_dir if bench_dir else f"{self.pyro_home}{self.solver_name}/tests/"

    def _bench_file(self):
        return self.bench_dir + self._output_name()

    def initialize_problem(self, problem_name, *, inputs_file=None, inputs_dict=None):
        """benchmark files are compared at the reference's rtol = 1e-12 (pyro_sim.py:353):
        a benchmark run takes the bit-faithful arithmetic (gpu.fast_math = 0) unless the
        caller names the build himself -- the product default (the contracted build) is
        parity-tested to 1e-10 / 1e-12 of the reference, not to the last bits of a stored
        file.  The build is recorded in the output file (attribute gpu_fast_math and the
        runtime parameters), so a mismatch can be traced to it."""
        # at the strength of a default: an inputs file or inputs_dict that names the build wins
        self.rp.set_param("gpu.fast_math", 0)
        super().initialize_problem(problem_name, inputs_file=inputs_file, inputs_dict=inputs_dict)

    def run_sim(self, rtol=1.e-12):
        super().run_sim()
        if not self.comp_bench:
            if self.make_bench:
                self.store_as_benchmark()
            return self.sim
        verdict = self.compare_to_benchmark(rtol)
        if self.make_bench or (verdict != 0 and self.reset_bench_on_fail):
            self.store_as_benchmark()
        return verdict

    def compare_to_benchmark(self, rtol):
        from .util import compare, io_pyro
        stored = self._bench_file()
        msg.warning(f"comparing to: {stored} ")
        try:
            reference = io_pyro.read(stored)
        except OSError:
            msg.warning("ERROR opening compare file")
            return "ERROR opening compare file"
        code = compare.compare(self.sim.cc_data, reference.cc_data, rtol)
        if code:
            msg.warning("ERROR: " + compare.errors[code] + "\n")
            info = getattr(reference, "restart_info", None) or {}
            theirs = (info.get("params") or {}).get("gpu.fast_math")
            try:
                mine = self.rp.get_param("gpu.fast_math")
            except (KeyError, RuntimeError):
                mine = None
            if theirs is not None and mine is not None and int(theirs) != int(mine):
                msg.warning(f"(the stored file was written with gpu.fast_math = {int(theirs)}, "
                            f"this run used gpu.fast_math = {int(mine)}: the two builds agree "
                            "to 1e-10 / 1e-12, not to the last bit)\n")
        else:
            msg.success(f"results match benchmark to within relative tolerance of {rtol}\n")
        return code

    def store_as_benchmark(self):
        os.makedirs(self.bench_dir, exist_ok=True)
        target = self._bench_file()
        msg.warning(f"storing new benchmark: {target}\n")
        self.sim.write(target)


def parse_args():
    ap = argparse.ArgumentParser(description="pyro hot path on MI355X")
    for flag, text in (("--make_benchmark", "create a new benchmark file for regression testing"),
                       ("--compare_benchmark", "compare the end result to the stored benchmark")):
        ap.add_argument(flag, action="store_true", help=text)
    ap.add_argument("solver", metavar="solver-name", choices=valid_solvers)
    ap.add_argument("problem", metavar="problem-name")
    ap.add_argument("param", metavar="inputs-file")
    ap.add_argument("other", metavar="runtime-parameters", nargs="*",
                    help="section.option=value overrides")
    return ap.parse_args()


def main():
    args = parse_args()
    overrides = dict((k, _get_val(v)) for k, v in (item.split("=") for item in args.other))
    benchmarking = args.compare_benchmark or args.make_benchmark      # pyro_sim.py:453-461
    if benchmarking:
        run = PyroBenchmark(args.solver, comp_bench=args.compare_benchmark,
                            make_bench=args.make_benchmark)
    else:
        run = Pyro(args.solver, from_commandline=True)
    run.initialize_problem(args.problem, inputs_file=args.param, inputs_dict=overrides)
    outcome = run.run_sim()
    if args.compare_benchmark and outcome != 0:
        sys.exit(1)


if __name__ == "__main__":
    main()
