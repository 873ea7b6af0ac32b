#!/usr/bin/env python3
"""The driver: `Pyro(solver).initialize_problem(problem, ...)`, `run_sim()`,
`single_step()` and the `pyro_sim.py solver problem inputs` command line, with
the surface of pyro/pyro_sim.py:33-456.

single_step() is the hot path (pyro_sim.py:241-281):
    cc_data.fill_BC_all()  ->  compute_timestep()  ->  evolve()
all three run on the GPU; only dt (8 bytes) comes back per step.
"""
import argparse
import importlib
import os
import sys

from .util import msg
from .util import profile_pyro as profile
from .util.runparams import RuntimeParameters, _get_val

# solvers with a device implementation in this package; the reference's other
# solvers are out of scope
valid_solvers = ["advection", "burgers", "compressible", "compressible_rk", "diffusion", "swe",
                 "incompressible", "incompressible_viscous", "burgers_viscous"]
# solvers that share another solver's problem directory (inputs files)
problem_home = {"compressible_rk": "compressible"}


class Pyro:
    def __init__(self, solver_name, *, from_commandline=False):
        if from_commandline:
            msg.bold("pyro ...")
        if solver_name.startswith("pyro2_amd."):
            solver_name = solver_name[len("pyro2_amd."):]
        if solver_name not in valid_solvers:
            msg.fail(f"ERROR: {solver_name} is not a valid solver")
        self.from_commandline = from_commandline
        self.pyro_home = os.path.dirname(os.path.realpath(__file__)) + "/"
        self.solver = importlib.import_module("pyro2_amd." + solver_name)
        self.solver_name = solver_name
        self.problem_name = None
        self.problem_func = None
        self.problem_source = None
        self.problem_params = None
        self.problem_finalize = None
        self.custom_problems = {}
        self.rp = RuntimeParameters()
        self.rp.load_params(self.pyro_home + "_defaults")
        self.rp.load_params(self.pyro_home + self.solver_name + "/_defaults")
        self.tc = profile.TimerCollection()
        self.is_initialized = False

    def add_problem(self, name, problem_func, *, problem_params=None):
        """register a user problem: problem_func(cc_data, rp) fills the state"""
        self.custom_problems[name] = (problem_func, problem_params or {})

    def initialize_problem(self, problem_name, *, inputs_file=None, inputs_dict=None):
        if problem_name in self.custom_problems:
            self.problem_func, self.problem_params = self.custom_problems[problem_name]
            self.problem_finalize = None
            self.problem_source = None
        else:
            problem = importlib.import_module(
                f"pyro2_amd.{self.solver_name}.problems.{problem_name}")
            self.problem_func = problem.init_data
            self.problem_params = problem.PROBLEM_PARAMS
            self.problem_finalize = problem.finalize
            self.problem_source = getattr(problem, "source_terms", None)
            if inputs_file is None:
                inputs_file = problem.DEFAULT_INPUTS
        self.problem_name = problem_name
        # Problem sources run on the device when they have the form of the
        # reference's three heating problems, S[energy] = rho * e_rate *
        # profile(x, y): the problem module then also provides
        # heating_profile(grid, rp) -> (e_rate, profile).  Any other source_terms()
        # is evaluated on the host, twice per step, by the compressible solver
        # (Simulation._evolve_host_source: two PCIe round trips of the state per step).
        self.problem_heating = None
        if self.problem_source is not None:
            self.problem_heating = getattr(
                sys.modules.get(self.problem_source.__module__), "heating_profile", None)
            if self.problem_heating is None and self.solver_name != "compressible":
                msg.fail("ERROR: this problem's source_terms() has no heating_profile() "
                         "companion; only sources of the form rho * e_rate * profile(x, y) "
                         "are carried by the device path")

        for k, v in self.problem_params.items():
            self.rp.set_param(k, v, no_new=False)
        if inputs_file is not None:
            if not os.path.isfile(inputs_file):
                base = problem_home.get(self.solver_name, self.solver_name)
                inputs_file = self.pyro_home + base + "/problems/" + inputs_file
                if not os.path.isfile(inputs_file):
                    msg.fail("ERROR: inputs file does not exist")
            self.rp.load_params(inputs_file, no_new=1)
        if not self.from_commandline:
            # library / notebook use: quiet, no files, no windows
            self.rp.set_param("vis.dovis", 0)
            self.rp.set_param("driver.verbose", 0)
            self.rp.set_param("io.do_io", 0)
        if inputs_dict is not None:
            for k, v in inputs_dict.items():
                self.rp.set_param(k, v)
        self.rp.print_paramfile()
        self.verbose = self.rp.get_param("driver.verbose")
        self.dovis = self.rp.get_param("vis.dovis")

        self.sim = self.solver.Simulation(
            self.solver_name, self.problem_name, self.problem_func, self.rp,
            problem_finalize_func=self.problem_finalize,
            problem_source_func=self.problem_source, timers=self.tc)
        self.sim.problem_heating = getattr(self, "problem_heating", None)
        self.sim.initialize()
        self.sim.preevolve()
        self.sim.cc_data.t = 0.0
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
        chk = io_pyro.read(filename)
        info = getattr(chk, "restart_info", None)
        if info is None or chk.solver_name != self.solver_name:
            msg.fail(f"ERROR: {filename} is not an output file of the {self.solver_name} solver")
        params = {}
        for k, v in info["params"].items():
            if isinstance(v, bytes):
                v = v.decode()
            params[k] = v.item() if hasattr(v, "item") else v
        params.update(inputs_dict or {})
        problem = chk.problem_name.decode() if isinstance(chk.problem_name, bytes) \
            else chk.problem_name
        for k, v in list(params.items()):      # parameters of the problem module come first
            try:
                self.rp.get_param(k)
            except (KeyError, RuntimeError):
                self.rp.set_param(k, v, no_new=False)
        self.initialize_problem(problem, inputs_dict=params)
        dst, src = self.sim.cc_data, chk.cc_data
        if (dst.grid.nx, dst.grid.ny) != (src.grid.nx, src.grid.ny):
            msg.fail("ERROR: the grid of the restart file does not match the inputs")
        for name in src.names:
            dst.get_var(name).v()[:, :] = src.get_var(name).v()
        dst.t = float(src.t)
        self.sim.n = int(chk.n)
        if info["dt"] is not None:
            self.sim.dt, self.sim.dt_old = float(info["dt"]), float(info["dt_old"])
        else:       # a file written by pyro itself: no dt history
            self.sim.n = max(self.sim.n, 1)
            self.sim.dt_old = 1.e33
        # the output counter of do_output() (simulation_null.py:270-290): without it
        # every step after a restart is "due" until the counter has caught up
        dt_out = self.rp.get_param("io.dt_out")
        if dt_out > 0.0:
            self.sim.n_num_out = int(dst.t / dt_out)

    def run_sim(self):
        if not self.is_initialized:
            msg.fail("ERROR: problem has not been initialized")
        tm_main = self.tc.timer("main")
        tm_main.begin()
        basename = self.rp.get_param("io.basename")
        do_io = self.rp.get_param("io.do_io")
        if do_io:
            self.sim.write(f"{basename}{self.sim.n:04d}")
        if self.dovis:
            import matplotlib.pyplot as plt
            plt.ion()
            plt.figure(num=1, figsize=(8, 6), dpi=100, facecolor="w")
            self.sim.dovis()
        while not self.sim.finished():
            # nothing to print, plot or write between the steps: hand a batch of them to
            # the device at once (the dt policy runs there; solvers that can do it)
            if not (self.verbose > 0 or self.dovis or do_io) and \
                    getattr(self.sim, "can_evolve_many", lambda: False)():
                self.sim.evolve_many(min(64, self.sim.max_steps - self.sim.n))
                continue
            self.single_step()
        if do_io or self.rp.get_param("io.force_final_output"):
            if self.verbose > 0:
                msg.warning("outputting...")
            self.sim.write(f"{basename}{self.sim.n:04d}")
        tm_main.end()
        if self.verbose > 0:
            self.rp.print_unused_params()
            self.tc.report()
        self.sim.finalize()

    def single_step(self):
        if not self.is_initialized:
            msg.fail("ERROR: problem has not been initialized")
        self.sim.cc_data.fill_BC_all()
        self.sim.compute_timestep()
        self.sim.evolve()
        if self.verbose > 0:
            print("%5d %10.5f %10.5f" % (self.sim.n, self.sim.cc_data.t, self.sim.dt))
        if self.sim.do_output():
            if self.verbose > 0:
                msg.warning("outputting...")
            basename = self.rp.get_param("io.basename")
            self.sim.write(f"{basename}{self.sim.n:04d}")
        if self.dovis:
            tm_vis = self.tc.timer("vis")
            tm_vis.begin()
            self.sim.dovis()
            if self.rp.get_param("vis.store_images") == 1:
                import matplotlib.pyplot as plt
                basename = self.rp.get_param("io.basename")
                plt.savefig(f"{basename}{self.sim.n:04d}.png")
            tm_vis.end()

    def __repr__(self):
        return f"Pyro('{self.solver_name}')"

    def __str__(self):
        s = f"Solver = {self.solver_name}\n"
        if self.is_initialized:
            s += f"Problem = {self.sim.problem_name}\n"
            s += f"Simulation time = {self.sim.cc_data.t}\n"
            s += f"Simulation step number = {self.sim.n}\n"
        return s + "\nRuntime Parameters\n------------------\n" + str(self.rp)

    def get_var(self, v):
        if not self.is_initialized:
            msg.fail("ERROR: problem has not been initialized")
        return self.sim.cc_data.get_var(v)

    def get_grid(self):
        if not self.is_initialized:
            msg.fail("ERROR: problem has not been initialized")
        return self.sim.cc_data.grid

    def get_sim(self):
        return self.sim


class PyroBenchmark(Pyro):
    """Pyro that compares its end state with a stored HDF5 output / stores one
    (pyro/pyro_sim.py:322-408)"""

    def __init__(self, solver_name, *, comp_bench=False, reset_bench_on_fail=False,
                 make_bench=False, bench_dir=None):
        super().__init__(solver_name)
        self.comp_bench = comp_bench
        self.reset_bench_on_fail = reset_bench_on_fail
        self.make_bench = make_bench
        self.bench_dir = bench_dir or (self.pyro_home + self.solver_name + "/tests/")

    def _bench_file(self):
        basename = self.rp.get_param("io.basename")
        return f"{self.bench_dir}{basename}{self.sim.n:04d}"

    def run_sim(self, rtol=1.e-12):
        super().run_sim()
        result = 0
        if self.comp_bench:
            result = self.compare_to_benchmark(rtol)
        if self.make_bench or (result != 0 and self.reset_bench_on_fail):
            self.store_as_benchmark()
        return result if self.comp_bench else self.sim

    def compare_to_benchmark(self, rtol):
        from .util import compare, io_pyro
        compare_file = self._bench_file()
        msg.warning(f"comparing to: {compare_file} ")
        try:
            sim_bench = io_pyro.read(compare_file)
        except OSError:
            msg.warning("ERROR opening compare file")
            return "ERROR opening compare file"
        result = compare.compare(self.sim.cc_data, sim_bench.cc_data, rtol)
        if result == 0:
            msg.success(f"results match benchmark to within relative tolerance of {rtol}\n")
        else:
            msg.warning("ERROR: " + compare.errors[result] + "\n")
        return result

    def store_as_benchmark(self):
        os.makedirs(self.bench_dir, exist_ok=True)
        bench_file = self._bench_file()
        msg.warning(f"storing new benchmark: {bench_file}\n")
        self.sim.write(bench_file)


def parse_args():
    p = argparse.ArgumentParser(description="pyro hot path on MI355X")
    p.add_argument("--make_benchmark", action="store_true",
                   help="create a new benchmark file for regression testing")
    p.add_argument("--compare_benchmark", action="store_true",
                   help="compare the end result to the stored benchmark")
    p.add_argument("solver", metavar="solver-name", choices=valid_solvers)
    p.add_argument("problem", metavar="problem-name")
    p.add_argument("param", metavar="inputs-file")
    p.add_argument("other", metavar="runtime-parameters", nargs="*",
                   help="section.option=value overrides")
    return p.parse_args()


def main():
    args = parse_args()
    other = {}
    for param_string in args.other:
        k, v = param_string.split("=")
        other[k] = _get_val(v)
    if args.compare_benchmark or args.make_benchmark:     # pyro_sim.py:453-461
        pyro = PyroBenchmark(args.solver, comp_bench=args.compare_benchmark,
                             make_bench=args.make_benchmark)
    else:
        pyro = Pyro(args.solver, from_commandline=True)
    pyro.initialize_problem(args.problem, inputs_file=args.param, inputs_dict=other)
    result = pyro.run_sim()
    if args.compare_benchmark and result != 0:
        sys.exit(1)


if __name__ == "__main__":
    main()
