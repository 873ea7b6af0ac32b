#!/usr/bin/env python3
"""bench.py -- throughput of the pyro2 hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Headline workload (BASELINE.json metric "cell-updates/s ... at 1/2/4/8 GPUs",
configs[4]): compressible Sedov, inputs.sedov physics, 16384 x 16384, x-slab
decomposed over the N GPUs of one node (strong scaling), one process per GPU
(torch.distributed.run launches us; RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from
the env), halo exchange + dt all-reduce over RCCL inside libpyrohip.

A "step" is one full Pyro.single_step of the hot path: ghost fill (halo
exchange), CFL time step (device reduction + driver dt policy), evolve.
State is resident in HBM before the timed region starts.

Rank 0 prints ONE JSON line.  At N=1 it also carries
  roofline      HBM roofline of the update kernels (HIP events, per launch),
                with the FP64-VALU figures that actually bind this kernel
  cpu_baseline  the oracle (CPU port of the reference) on a bounded sample
  also          sedov_developed: the SAME kernel on developed flow (a 1024^2
                Sedov blast run to t = 0.1 on the GPU and tiled over the grid:
                the shocked region covers ~30 % of the cells; the headline state
                is 99.9 % ambient gas), >= 200 timed steps, both builds;
                sedov_exact: the headline workload in the bit-faithful build;
                advection 2048^2 (configs[1]), multigrid 4096^2 V-cycles/s
                (configs[3]) and the incompressible solver, measured the same way

`python bench.py --gpus N` without a launcher spawns its own N ranks
(127.0.0.1 rendezvous); under `python -m torch.distributed.run` it uses the
RANK / WORLD_SIZE it is given.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec
SEDOV_BYTES_PER_CELL = 64   # read 4 + write 4 conserved doubles
ADV_BYTES_PER_CELL = 16     # read a + write a
MG_BYTES_PER_CELL_VCYCLE = 720      # SURVEY 8(d)'s one-pass-per-iteration model (kept as a labelled extra)
# The floor under temporal blocking: a V-cycle visits every level twice, and a
# visit cannot move less than  down-leg: read v, f + write v (24 B) + the restricted residual
# (8 B / 4) ; up-leg: read v, f + the coarse correction (8 B / 4) + write v  = 52 B per level cell,
# x 4/3 for the level pyramid = 69.3 B per finest cell per V-cycle.
MG_FLOOR_BYTES_PER_LEVEL_CELL = 52
MG_FLOOR_BYTES_PER_CELL_VCYCLE = MG_FLOOR_BYTES_PER_LEVEL_CELL * 4.0 / 3.0
FP64_PEAK_FLOPS = 78.6e12           # MI355X FP64 vector peak (FMA = 2 flop), MI355X_MICROARCH.md
VALU_ISSUE_PEAK = 1024 * 2.4e9 / 4  # wave-instructions/s: 1024 SIMDs, 4 cycles per FP64 wave-instruction
# arithmetic minimum of one CTU + HLLC cell update (operation count of the
# reference's formulas with every shared quantity computed once; FMA counted as 2)
SEDOV_MIN_FLOPS_PER_CELL = 880


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--nx", type=int, default=16384)
    ap.add_argument("--fast-math", type=int, default=None,
                    help="1: contracted arithmetic (parity 1e-10), 0: bit-faithful")
    ap.add_argument("--kernel-set", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true")
    ap.add_argument("--cpu-sample-nx", type=int, default=1024)
    ap.add_argument("--developed-steps", type=int, default=250)
    ap.add_argument("--no-developed", action="store_true")
    ap.add_argument("--scale-check", action="store_true",
                    help="before timing: compressible Sedov 2048^2, 12 steps, on 1 rank vs the N ranks "
                         "of this run, must agree bit for bit (SURVEY 8(d).5); the result goes "
                         "into config.scale_check, a mismatch is fatal")
    ap.add_argument("--host-dt", action="store_true",
                    help="step from the host (one dt read-back per step) instead of "
                         "pyrohip_comp_evolve")
    return ap.parse_args()


def self_spawn(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: start N ranks of this
    script with a 127.0.0.1 rendezvous, relay rank 0's JSON line"""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PYRO_BENCH_SPAWNED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                      env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    if any(rcs):
        sys.exit(f"bench.py: rank exit codes {rcs}")


class Dist:
    """process-group plumbing (torch.distributed, gloo on CPU tensors);
    the data path (halos, dt) goes through RCCL inside libpyrohip."""

    def __init__(self, world):
        self.world = world
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.td = None
        if world > 1:
            import torch.distributed as td
            td.init_process_group("gloo", rank=self.rank, world_size=world)
            self.td = td

    def barrier(self):
        if self.td:
            self.td.barrier()

    def max(self, x):
        if not self.td:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        self.td.all_reduce(t, op=self.td.ReduceOp.MAX)
        return float(t[0])

    def bcast_bytes(self, b, n):
        if not self.td:
            return b
        import torch
        t = torch.zeros(n, dtype=torch.uint8)
        if self.rank == 0:
            t = torch.frombuffer(bytearray(b), dtype=torch.uint8).clone()
        self.td.broadcast(t, 0)
        return bytes(t.numpy().tobytes())


def also_traffic(section, key):
    """measured fabric bytes of the advection step / the multigrid V-cycle from the committed
    PMC passes (tools/pmc_also.sh -> profiles/r*_also_traffic.json); None when absent"""
    import glob
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_also_traffic.json")))
    if not fs:
        return None
    try:
        return json.load(open(fs[-1]))[section][key]
    except Exception:
        return None


def kernel_table(prof, nlaunch_unit):
    return {k: {"launches": n, "avg_ms": ms / max(n, 1)} for k, (n, ms) in prof.items()}


def developed_tile(ctx, device, n=1024, tmax=0.1):
    """Sedov (inputs.sedov physics) at n x n run on this GPU to t = tmax with the
    bit-faithful build: the blast has grown to r ~ 0.3 and is far from the outflow
    boundary, so copies of the tile can sit side by side without seams.
    Returns the interior (n, n, 4) and the fraction of cells the blast has reached."""
    from pyro2_amd.compressible.problems.sedov import sedov_state
    from pyro2_amd.decomp import DtPolicy
    ng = 4
    st = device.DeviceState(ctx, n, n, ng, [["outflow"] * 4] * 4)
    st.upload(np.nan_to_num(sedov_state(n, n, ng, 0.0, 1.0, 0.0, 1.0, 1.4, 0.01, 4)))
    P = device.make_comp_params(1.0 / n, 1.0 / n, fast_math=0, kernel_set=-1)
    pol = DtPolicy(tmax)
    while pol.t < tmax and pol.n < 100000:
        st.fill_bc()
        dt = pol(st.comp_dt(P, 0.8))
        st.comp_step(P, dt)
        pol.advance(dt)
    U = st.download()[ng:-ng, ng:-ng].copy()
    frac = float((np.abs(U[..., 0] - 1.0) > 1e-8).mean())
    return U, frac, pol.n


def bench_sedov(args, dist, ctx, device, defaults, steps=None, warmup=None, tile=None, nx=None,
                collect=False):
    """tile = None: the Sedov initial condition at t = 0 (BASELINE config);
    tile = (n, n, 4) array: that state repeated over the whole grid;
    collect: also return this rank's final slab interior (scale check)"""
    from pyro2_amd.compressible.problems.sedov import sedov_state
    from pyro2_amd.decomp import DtPolicy, NoComm, RcclComm, SlabCompressible, SlabDecomp
    nx = ny = args.nx if nx is None else nx
    ng = 4
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    dec = SlabDecomp(nx, dist.world, dist.rank, periodic=False)
    if dist.world == 1:
        comm = NoComm()
    elif dist.comm_kind == "rccl":
        comm = RcclComm(ctx)
    else:
        from pyro2_amd.decomp import HostStagedComm
        comm = HostStagedComm(dist.td)
    kw = dict(dx=1.0 / nx, dy=1.0 / ny, fast_math=defaults["fast_math"],
              kernel_set=defaults["kernel_set"])
    slab = SlabCompressible(ctx, dec, ny, ["outflow"] * 4, kw, comm, ng=ng)
    st = slab.state
    # initial condition: generated slab by slab on the host (never more than a few
    # hundred rows in memory), resident in HBM before the timed region
    chunk = 512 if tile is None else 256
    if tile is not None:
        n = tile.shape[0]
        cols = (np.arange(ny + 2 * ng) - ng) % n
        tile_cols = np.ascontiguousarray(tile[:, cols, :])      # (n, qy, 4)
    for r0 in range(0, dec.nx_local + 2 * ng, chunk):
        nr = min(chunk, dec.nx_local + 2 * ng - r0)
        if tile is None:
            st.upload_rows(r0, sedov_state(nx, ny, ng, 0.0, 1.0, 0.0, 1.0, 1.4, 0.01, 4,
                                           i0=dec.i0 + r0, ni=nr))
        else:
            rows = (np.arange(dec.i0 + r0, dec.i0 + r0 + nr) - ng) % n
            st.upload_rows(r0, tile_cols[rows])
    pol = DtPolicy(tmax=0.1 if tile is None else 1.0e9)
    # a step = ghost fill (halo exchange), CFL time step with the driver's policy, evolve.
    # Default: the steps are enqueued on the device back to back (pyrohip_comp_evolve, the
    # dt policy runs in a kernel); --host-dt: the Python loop with one read-back per step
    device_dt = not args.host_dt and isinstance(comm, (NoComm, RcclComm))

    def run(n):
        if n <= 0:
            return
        if device_dt:
            assert len(slab.evolve(pol, 0.8, n)) == n
        else:
            for _ in range(n):
                slab.step(pol, 0.8)
    run(warmup)
    ctx.sync()
    dist.barrier()
    ctx.timer_start()
    t0 = time.perf_counter()
    run(steps)
    ctx.sync()
    t1 = time.perf_counter()
    ev_ms = ctx.timer_stop()
    dist.barrier()
    elapsed = dist.max(t1 - t0)
    # the kernels' own durations: more steps with the library's HIP events around every
    # launch, OUTSIDE the timed region (the events cost ~4 us per launch: nothing against the
    # 9.5 ms kernel of the headline, 2-7 % of a 4096^2 step)
    nprof = steps if r_short(elapsed, steps) else min(steps, 5)
    ctx.prof_enable(True)
    run(nprof)
    ctx.sync()
    prof = ctx.prof_report()
    ctx.prof_enable(False)
    dist.barrier()
    res = {"elapsed": elapsed, "cells": float(nx) * ny, "prof": prof, "event_ms": ev_ms,
           "t": pol.t, "dt": pol.dt_old, "local_cells": float(dec.nx_local) * ny, "steps": steps,
           "prof_steps": nprof, "dt_policy": "device" if device_dt else "host"}
    if collect:
        res["interior"] = st.download()[ng:-ng, ng:-ng].copy()
        res["rows"] = (dec.i0, dec.nx_local)
    del slab, st
    return res


def r_short(elapsed, steps):
    """steps shorter than 5 ms: the instrumented pass takes as many steps as the timed one (a
    handful of short launches after the idle moment of the synchronisation run at lower clocks:
    4096^2 0.92 ms per launch over 5 steps against 0.68 ms by rocprofv3)"""
    return steps > 0 and elapsed / steps < 5.0e-3


def pmc_counts(fast_math):
    """VALU instruction / traffic counts of the dominant update kernel from the committed
    rocprofv3 PMC passes (profiles/traffic.json, tools/gpu_round.sh + tools/make_traffic.py)"""
    tr = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        return json.load(open(tr))[f"fast_math_{fast_math}"]
    except Exception:
        return None


def fp64_roofline(cells_per_s_kernel, fast_math, dom):
    """the roof that binds the CTU kernel is the FP64 vector unit, not HBM:
    arithmetic minimum x cell rate against the FMA peak, and the instruction-issue figure"""
    out = {
        "bound": "fp64_valu", "peak": FP64_PEAK_FLOPS / 1e12, "unit": "TFLOP/s",
        "achieved": SEDOV_MIN_FLOPS_PER_CELL * cells_per_s_kernel / 1e12,
        "frac": SEDOV_MIN_FLOPS_PER_CELL * cells_per_s_kernel / FP64_PEAK_FLOPS,
        "flops_per_cell_update": SEDOV_MIN_FLOPS_PER_CELL,
        "basis": "arithmetic minimum of one CTU + 4 x HLLC cell update (DESIGN.md 3, FMA = 2 "
                 "flop) x cell rate of the update kernel; the kernel is mostly non-FMA, so "
                 "the instruction-issue figures below are the tighter statement"}
    t = pmc_counts(fast_math)
    if t and t.get("kernel") == dom:
        ipc = t["valu_insts_per_cell_update"]        # lane-instructions per cell update
        out["valu_issue"] = {
            "valu_lane_insts_per_cell_update": ipc,
            "executed_flops_per_cell_update": t.get("flops_per_cell_update"),
            "achieved_wave_insts_per_s": ipc / 64.0 * cells_per_s_kernel,
            "peak_wave_insts_per_s": VALU_ISSUE_PEAK,
            "frac": ipc / 64.0 * cells_per_s_kernel / VALU_ISSUE_PEAK,
            "valu_busy_frac_of_kernel_time": t["valu_busy_ms"] / t["kernel_ms"],
            "source": "SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU of the same kernel (counted at 16384^2; "
                      "per cell update, the strip geometry differs slightly by size), " + t["measured_at"]}
    return out


def sedov_leg(r, defaults, nx, extra=None):
    """summary of a secondary Sedov measurement for the `also` block"""
    upd = r["prof"]
    tot_ms = sum(ms for (_, ms) in upd.values()) / max(r["prof_steps"], 1)
    timer = "events per launch (instrumented pass after the timed one)"
    if r_short(r["elapsed"], r["steps"]):
        # sub-millisecond launches: the events around every launch stretch them (4096^2: 0.82-0.92 ms
        # against 0.68 ms by rocprofv3 and a 0.74 ms STEP); the event pair around the timed
        # region / steps -- all launches of a step, the three small ones included -- is the bound
        tot_ms = r["event_ms"] / r["steps"]
        timer = "event pair over the timed region / steps (every launch of a step)"
    gbs = SEDOV_BYTES_PER_CELL * r["local_cells"] / (tot_ms * 1e-3) / 1e9 if tot_ms else 0.0
    dom = max(upd, key=lambda k: upd[k][1]) if upd else None
    out = {"value": r["cells"] * r["steps"] / r["elapsed"], "unit": "cell-updates/s",
           "ms_per_step": r["elapsed"] / r["steps"] * 1e3, "steps": r["steps"],
           "timed_seconds": r["elapsed"], "fast_math": defaults["fast_math"],
           "kernel_set": defaults["kernel_set"],
           "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": gbs / HBM_PEAK_GBS, "kernel_ms_per_step": tot_ms, "kernel_timer": timer,
                        "dominant_kernel": dom,
                        "kernels": {k: {"launches": n, "avg_ms": ms / n} for k, (n, ms) in upd.items()}},
           "roofline_fp64": fp64_roofline(r["local_cells"] / (tot_ms * 1e-3) if tot_ms else 0.0,
                                          defaults["fast_math"], dom)}
    if extra:
        out.update(extra)
    return out


class _Solo:
    """a one-rank stand-in for Dist (scale check: the single-domain run on this rank's GPU)"""
    world, rank, local_rank, td, comm_kind = 1, 0, 0, None, "rccl"

    def barrier(self):
        pass

    def max(self, x):
        return x


def scale_check(args, dist, ctx, device, defaults, nx=2048, steps=12):
    """Sedov nx^2 on the N ranks of this run against the single-domain run
    (every rank repeats it on its own GPU and compares its slab's rows): bit-identical or
    fatal.  Both runs use the row-marching kernel (kernel_set 2) and this run's arithmetic."""
    d = dict(defaults, kernel_set=2)
    rn = bench_sedov(args, dist, ctx, device, d, steps=steps, warmup=0, nx=nx, collect=True)
    r1 = bench_sedov(args, _Solo(), ctx, device, d, steps=steps, warmup=0, nx=nx, collect=True)
    i0, n = rn["rows"]
    same = bool(np.array_equal(rn["interior"], r1["interior"][i0:i0 + n])) and rn["t"] == r1["t"]
    bad = dist.max(0.0 if same else 1.0)
    if bad > 0.0:
        sys.exit(f"bench.py rank {dist.rank}: FATAL: scale check failed: sedov {nx}^2 x {steps} steps on "
                 f"{dist.world} ranks differs from the single-domain run (this rank: "
                 f"{'identical' if same else 'DIFFERENT'})")
    return {"workload": f"sedov {nx}x{nx}, {steps} steps, {dist.world} ranks vs 1 rank, kernel_set 2, "
                        f"fast_math {d['fast_math']}", "bit_identical": True, "sim_time": rn["t"]}


def sedov_size_leg(args, dist, ctx, device, defaults, nx, steps, warmup=5):
    """the same workload at another BASELINE size (configs[2] = 4096^2, north_star's target
    8192^2), both builds, each with the HBM roofline and the FP64 issue figures"""
    r = bench_sedov(args, dist, ctx, device, defaults, steps=steps, warmup=warmup, nx=nx)
    leg = sedov_leg(r, defaults, nx, {
        "workload": f"compressible sedov {nx}x{nx} (inputs.sedov physics), 1 GPU, steps "
                    f"{warmup}-{warmup + steps} from t = 0"})
    d2 = dict(defaults, fast_math=1 - defaults["fast_math"])
    r2 = bench_sedov(args, dist, ctx, device, d2, steps=max(5, steps // 2), warmup=2, nx=nx)
    leg["other_build"] = sedov_leg(r2, d2, nx)
    return leg


def reference_baseline(section, key):
    """the REFERENCE ITSELF timed on host cores (oracle/time_reference.py ->
    profiles/cpu_reference.json; measured in the build container: the GPU box has no copy of
    the reference) as the cpu_baseline of a leg"""
    try:
        ref = json.load(open(os.path.join(ROOT, "profiles", "cpu_reference.json")))
        r = ref[section][key]
    except Exception:
        return None
    return {"value": r["value"], "unit": r["unit"], "cores": r["cores"], "kind": "reference",
            "sample": f"{r['workload']}; " +
                      (f"{r['steps']} steps, {r['seconds_per_step']:.3f} s/step" if "steps" in r else
                       f"{r['cycles']} V-cycles, {r['seconds_per_vcycle']:.2f} s/V-cycle") +
                      f"; host {ref['cpu']} ({ref['host_cores']} cores, 1 used: the reference is "
                      f"single-threaded), {ref['date']}, oracle/time_reference.py -- measured in the "
                      "build container, not on the GPU box"}


def bench_advection(ctx, device, nx=2048, steps=100, warmup=10, fast_math=1, other=True):
    x = (np.arange(nx + 8) - 3.5) / nx
    X, Y = np.meshgrid(x, x, indexing="ij")
    ic = 1.0 + np.exp(-60.0 * ((X - 0.5) ** 2 + (Y - 0.5) ** 2))
    st = device.DeviceState(ctx, nx, nx, 4, [["periodic"] * 4])
    st.upload(ic)
    dt = 0.8 * min((1 / nx) / 1.0, (1 / nx) / 1.0)     # advection/simulation.py:38-54, u = v = 1, cfl 0.8

    def step():      # the ghost fill is folded into the step kernel (one launch per step)
        st.adv_step(0, 1 / nx, 1 / nx, 1.0, 1.0, dt, 2, fill=True, fast_math=fast_math)
    for _ in range(warmup):
        step()
    ctx.sync()
    # a HIP-event pair on the library's stream brackets the timed region: `steps` launches of
    # the one kernel a step is, back to back -- elapsed / steps is the average launch duration
    # including the gap to the next launch.  (Events around EVERY launch, the second pass
    # below, stretch a 21 us kernel to 25 us: rocprofv3 says 21.35, profiles/r03_adv2048_*.)
    ctx.timer_start()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ev_ms = ctx.timer_stop()
    ctx.sync()
    t1 = time.perf_counter()
    ctx.prof_enable(True)
    for _ in range(steps):
        step()
    ctx.sync()
    prof = ctx.prof_report()
    ctx.prof_enable(False)
    n, ms = prof["k_adv_step"]
    # launches of >= 100 us: the events around every launch (their ~4 us are < 4 % there, and
    # they leave out the write-back between two launches, as rocprofv3 does: 8192^2 254-266 us
    # against 280 us from launch to launch); shorter ones: the pair around the timed region
    per_launch = ms / n >= 0.1
    kern_s = (ms / n if per_launch else ev_ms / steps) * 1e-3
    traffic = also_traffic("adv_summary", "bytes_per_step") if nx == 2048 else None
    out_other = None
    if other:      # the other arithmetic (fast_math = 0: bit-faithful, the audit build)
        out_other = bench_advection(ctx, device, nx, steps, warmup, 1 - fast_math, other=False)
    return {"workload": f"advection smooth {nx}x{nx} periodic, limiter 2",
            "fast_math": fast_math, "other_build": out_other,
            "value": nx * nx * steps / (t1 - t0), "unit": "cell-updates/s",
            "ms_per_step": (t1 - t0) / steps * 1e3, "steps": steps,
            "roofline": {"bound": "hbm", "kernel": "k_adv_step",
                         "achieved": ADV_BYTES_PER_CELL * nx * nx / kern_s / 1e9,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ADV_BYTES_PER_CELL * nx * nx / kern_s / 1e9 / HBM_PEAK_GBS,
                         "kernel_avg_ms": kern_s * 1e3, "kernel_avg_ms_events_per_launch": ms / n,
                         "kernel_avg_ms_events_over_region": ev_ms / steps,
                         "kernel_timer": "events per launch" if per_launch else "event pair over the timed region",
                         "traffic": traffic,
                         "step_frac": ADV_BYTES_PER_CELL * nx * nx * steps / (t1 - t0) / 1e9 / HBM_PEAK_GBS,
                         "launches_per_step": 1},
            "cpu_baseline": reference_baseline("advection", str(nx)) if other else None}


def _mg_vcycles(ctx, device, nx, cycles):
    """seconds for `cycles` V-cycles of the Poisson test problem of MG.py's tests at nx^2 (the
    solve loop with its norms and convergence test, rtol 0), and the residual it ends with"""
    x = (np.arange(nx + 2) - 0.5) / nx
    X, Y = np.meshgrid(x, x, indexing="ij")
    rhs = -2.0 * ((1.0 - 6.0 * X ** 2) * Y ** 2 * (1.0 - Y ** 2) +
                  (1.0 - 6.0 * Y ** 2) * X ** 2 * (1.0 - X ** 2))
    m = device.DeviceMG(ctx, nx)
    L = m.nlevels - 1
    m.zero(L, 0)
    m.set(L, 1, rhs)
    m.init_rhs_norm()
    m.solve(rtol=0.0, max_cycles=2)     # warm-up
    m.zero(L, 0)
    ctx.sync()
    t0 = time.perf_counter()
    nc, res, rel = m.solve(rtol=0.0, max_cycles=cycles)
    ctx.sync()
    return time.perf_counter() - t0, res


def bench_mg(ctx, device, nx=4096, cycles=10, small_sizes=True):
    dt_, res = _mg_vcycles(ctx, device, nx, cycles)
    t0, t1 = 0.0, dt_
    vps = cycles / (t1 - t0)
    # the sizes the multigrid callers (incompressible, diffusion) solve on: launch-latency bound
    small = {}
    if nx == 4096 and small_sizes:
        for n2 in (512, 1024, 2048):
            d2, _ = _mg_vcycles(ctx, device, n2, 20)
            small[str(n2)] = d2 / 20 * 1e6
    model_gbs = MG_BYTES_PER_CELL_VCYCLE * nx * nx * vps / 1e9
    traffic = also_traffic("mg_summary", "bytes_per_vcycle") if nx == 4096 else None
    # the roofline entry is priced with the ALGORITHMIC floor of a temporally blocked V-cycle
    # (two visits per level, 52 B per level cell, x 4/3: MG_FLOOR_BYTES_PER_CELL_VCYCLE); the
    # measured fabric bytes (PMC, committed under profiles/) ride along as `traffic`, the
    # 720 B one-pass-per-iteration model of SURVEY 8(d) as `model_equivalent_gbs`
    gbs = MG_FLOOR_BYTES_PER_CELL_VCYCLE * nx * nx * vps / 1e9
    return {"workload": f"multigrid constant-coeff Poisson {nx}x{nx} dirichlet, "
                        f"{cycles} V-cycles (nsmooth 10, bottom 50)",
            "value": vps, "unit": "V-cycles/s", "ms_per_vcycle": (t1 - t0) / cycles * 1e3,
            "residual_error_after": res, "us_per_vcycle_by_size": small,
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": traffic,
                         "model_equivalent_gbs": model_gbs,
                         "traffic_gbs": traffic * vps / 1e9 if traffic else None,
                         "algorithmic_bytes_per_vcycle": MG_FLOOR_BYTES_PER_CELL_VCYCLE * nx * nx,
                         "basis": "floor of a temporally blocked V-cycle: every level is visited twice, "
                                  "down-leg read v, f + write v + restricted residual, up-leg read v, f + "
                                  "coarse correction + write v = 52 B per level cell, x 4/3 for the pyramid = "
                                  "69.3 B per finest cell per V-cycle, x V-cycles/s; traffic = measured fabric "
                                  "bytes per V-cycle (profiles/*_also_traffic.json, rocprofv3 --pmc FETCH_SIZE "
                                  "x 2 + WRITE_SIZE over every multigrid kernel); model_equivalent_gbs = the "
                                  "720 B one-pass-per-iteration model of SURVEY 8(d) x V-cycles/s"},
            "cpu_baseline": reference_baseline("multigrid", str(nx))}


def bench_incompressible(ctx, device, nx=2048, steps=5):
    """the solver north_star names as the caller of the multigrid V-cycle:
    incompressible shear layer, two MG solves (rtol 1e-12) per step"""
    import contextlib
    import io
    import tempfile
    from pyro2_amd.pyro_sim import Pyro
    cwd = os.getcwd()
    os.chdir(tempfile.mkdtemp())          # Pyro writes inputs.auto
    try:
        old = device.Context._default
        device.Context._default = ctx
        with contextlib.redirect_stdout(io.StringIO()):
            p = Pyro("incompressible")
            p.initialize_problem("shear", inputs_dict={"mesh.nx": nx, "mesh.ny": nx,
                                                       "driver.max_steps": steps + 1})
            p.single_step()
        ctx.sync()
        t0 = time.perf_counter()
        cyc = []
        for _ in range(steps):
            p.single_step()
            cyc.append(sum(p.sim.mg_cycles))
        ctx.sync()
        t1 = time.perf_counter()
    finally:
        device.Context._default = old
        os.chdir(cwd)
    ms = (t1 - t0) / steps * 1e3
    return {"workload": f"incompressible shear {nx}x{nx} (limiter 2, proj_type 2), {steps} steps",
            "value": nx * nx / (ms * 1e-3), "unit": "cell-updates/s", "ms_per_step": ms,
            "vcycles_per_step": sum(cyc) / steps}


def bench_small_grids(ctx, device, sizes=(64, 256, 512), steps=400):
    """the grid sizes the reference's own problems use: time per step with the steps
    enqueued on the device (pyrohip_comp_evolve: the tile kernel applies the boundary
    rules and writes the ghost frame itself, the dt policy takes the CFL minimum: two
    launches per step)"""
    from pyro2_amd.compressible.problems.sedov import sedov_state
    from pyro2_amd.decomp import DtPolicy
    out = {}
    for nx in sizes:
        st = device.DeviceState(ctx, nx, nx, 4, [["outflow"] * 4] * 4)
        st.upload(sedov_state(nx, nx, 4, 0.0, 1.0, 0.0, 1.0, 1.4, 0.01, 4))
        P = device.make_comp_params(1.0 / nx, 1.0 / nx, fast_math=1, kernel_set=-1)
        pol = DtPolicy(1.0e9)
        st.comp_evolve(P, 0.8, pol, 50)
        ctx.sync()
        t0 = time.perf_counter()
        st.comp_evolve(P, 0.8, pol, steps)
        ctx.sync()
        us = (time.perf_counter() - t0) / steps * 1e6
        out[f"{nx}x{nx}"] = {"us_per_step": us, "value": nx * nx / (us * 1e-6), "unit": "cell-updates/s"}
    return {"workload": f"compressible sedov on small grids, {steps} steps enqueued on the device, "
                        "fast build", "sizes": out}


def cpu_baseline_sedov(sample_nx, max_seconds=25.0):
    """the oracle (single-threaded C port of the reference) on a bounded sample
    of the same workload: Sedov, same physics, sample_nx^2, from t = 0"""
    from oracle import orc
    from helpers import DtPolicy, meta_to_params
    from sedov_ic import sedov_ic
    ic, meta, bcs = sedov_ic(sample_nx)
    P, cfl = meta_to_params(meta, bcs)
    U = ic.copy()
    pol = DtPolicy(0.1)
    n = 0
    t0 = time.perf_counter()
    while n < 50:
        orc.comp_fill_bc(U, P.nx, P.ny, P.ng, bcs)
        dt = pol(orc.comp_dt(U, P.nx, P.ny, P.ng, P.dx, P.dy, P.gamma, cfl))
        orc.comp_step(U, P, dt)
        pol.advance(dt)
        n += 1
        if time.perf_counter() - t0 > max_seconds:
            break
    el = time.perf_counter() - t0
    return {"value": sample_nx * sample_nx * n / el, "unit": "cell-updates/s",
            "cores": 1, "kind": "port",
            # the reference's own time next to the port.  numba is not
            # installable here, so what can be timed of the reference itself is its
            # NumPy stages with the njit kernels stubbed out: a LOWER bound on its step
            # time, measured once in the survey container (1 core, 2.1 GHz Xeon), carried
            # here as a labelled constant -- not re-measured on this host
            "reference_numpy_stages_only": {
                "value_upper_bound": [0.55e6, 0.36e6, 0.26e6], "at_nx": [512, 1024, 2048],
                "unit": "cell-updates/s", "source": "SURVEY.md 6 (compressible step, NumPy-only "
                "stages, njit kernels stubbed): real reference <= these rates"},
            "sample": f"oracle/pyro_oracle.c (gcc -O2, 1 thread), compressible sedov "
                      f"{sample_nx}x{sample_nx}, {n} steps from t=0, {el:.1f} s; host has "
                      f"{os.cpu_count()} cores; the reference itself is single-threaded "
                      f"NumPy/numba (SURVEY 8(d))"}


def main():
    args = parse()
    # multi-process GPU work on this image: dmabuf IPC only (already exported on
    # the GPU boxes; kept here so that a bare environment behaves the same), and
    # all ranks of this bench live on ONE node: let RCCL's bootstrap use the
    # loopback interface unless the caller chose one
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus > 1 and "RANK" not in os.environ:
        return self_spawn(args)
    # stdout carries exactly ONE JSON line: libraries that announce themselves on the
    # C stdout (RCCL's version banner at communicator creation, gloo) are sent to
    # stderr for the whole run, the line itself goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    dist = Dist(world)
    from pyro2_amd import device
    ndev = device.device_count()
    dist.oversubscribed = world > 1 and world > ndev
    if dist.oversubscribed and dist.rank == 0:
        print(f"[bench] WARNING: --gpus {world} on a box with {ndev} GPU(s): ranks share GPUs; "
              "this run checks the multi-rank path, its number is NOT a scaling result "
              "(flagged as config.oversubscribed)", file=sys.stderr)
    # one rank per GPU; several ranks on one GPU only happens when debugging
    # the launcher on a smaller box and is flagged in the output
    ctx = device.Context(dist.local_rank % ndev)
    dist.comm_kind, dist.comm_note = "rccl", None
    want_comm = os.environ.get("PYRO_BENCH_COMM", "rccl")
    if world > 1:
        # data path: RCCL inside libpyrohip.  A communicator that cannot be created on
        # every rank is FATAL: a host-staged (PCIe-bound) number must never pass for a
        # scaling result.  PYRO_BENCH_COMM=host asks for the host-staged path on purpose
        # (debugging the launcher; flagged in config.halo).
        err = None
        if want_comm == "rccl":
            try:
                uid = device.Context.comm_unique_id() if dist.rank == 0 else b""
                uid = dist.bcast_bytes(uid, 128)
                ctx.comm_init(world, dist.rank, uid)
                assert ctx.allreduce_min(float(dist.rank + 1)) == 1.0
            except Exception as e:    # noqa: BLE001
                err = f"{type(e).__name__}: {e}"
            else:
                dist.rccl_ranks = ctx.comm_size()
        else:
            err = "PYRO_BENCH_COMM=%s: host-staged halos requested" % want_comm
        if dist.max(1.0 if err else 0.0) > 0.0:
            note = err or "RCCL initialisation failed on another rank"
            if want_comm == "rccl":
                sys.exit(f"bench.py rank {dist.rank}: FATAL: RCCL communicator over {world} ranks could "
                         f"not be created ({note}).  No number is printed: a host-staged run is not a "
                         "scaling result (PYRO_BENCH_COMM=host runs that path on purpose).")
            dist.comm_kind = "host-staged"
            dist.comm_note = note
            print(f"[bench rank {dist.rank}] WARNING: {note}; halo exchange staged through the "
                  "host over gloo (NOT a scaling result)", file=sys.stderr)
    # default: the contracted / reciprocal-division build, parity-tested to the
    # north_star tolerance (1e-10); --fast-math 0 times the bit-faithful build.
    # kernel_set -1: the library picks (row-marching wavefront kernel from 2048^2 on)
    defaults = {"fast_math": 1 if args.fast_math is None else args.fast_math,
                "kernel_set": -1 if args.kernel_set is None else args.kernel_set}
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    except Exception:
        pass

    check = scale_check(args, dist, ctx, device, defaults) if (args.scale_check and world > 1) else None
    r = bench_sedov(args, dist, ctx, device, defaults)
    value = r["cells"] * args.steps / r["elapsed"]
    out = {
        "metric": "cell-updates/s", "value": value, "unit": "cell-updates/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": r["elapsed"] / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"compressible sedov {args.nx}x{args.nx} (inputs.sedov physics: "
                               "HLLC, limiter 2, flattening, cvisc 0.1, cfl 0.8, outflow), "
                               f"x-slab decomposed over {world} GPU(s), "
                               + ("RCCL halo exchange" if dist.comm_kind == "rccl"
                                  else f"HOST-STAGED halo exchange ({dist.comm_note or 'debug path'}): "
                                       "NOT a scaling result"),
                   "parallelism": f"slab{world}",
                   "halo": dist.comm_kind if world > 1 else "none",
                   "rccl_ranks": getattr(dist, "rccl_ranks", None) if world > 1 else None,
                   "fast_math": defaults["fast_math"],
                   "kernel_set": defaults["kernel_set"], "sim_time": r["t"],
                   "dt_policy": r["dt_policy"] + (" (pyrohip_comp_evolve: no host round trip per step)"
                                                  if r["dt_policy"] == "device" else ""),
                   "state": "steps %d-%d from t = 0 (blast radius << grid: > 99 %% of the cells are "
                            "ambient gas; see also.sedov_developed)" % (args.warmup, args.warmup + args.steps)},
    }
    if dist.comm_note:
        out["config"]["halo_note"] = dist.comm_note
    if check:
        out["config"]["scale_check"] = check
    if dist.oversubscribed:
        out["config"]["oversubscribed"] = "several ranks share one GPU (debug run, not a result)"
    if dist.rank == 0:
        # roofline of the update kernels: algorithmic bytes of ONE rank's slab
        # per step / HIP-event time of that rank's kernels per step
        prof = r["prof"]
        upd = {k: v for k, v in prof.items()}
        tot_ms = sum(ms for (_, ms) in upd.values()) / max(r["prof_steps"], 1)
        dom = max(upd, key=lambda k: upd[k][1]) if upd else None
        gbs = SEDOV_BYTES_PER_CELL * r["local_cells"] / (tot_ms * 1e-3) / 1e9 if tot_ms else 0.0
        cells_per_s_kernel = r["local_cells"] / (tot_ms * 1e-3) if tot_ms else 0.0
        out["roofline"] = {
            "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": gbs / HBM_PEAK_GBS, "traffic": None,
            "basis": "64 B/cell-update (SURVEY 8(d)) x cells of this rank / sum of the "
                     "update kernels' HIP-event durations per step (events around every launch, "
                     "over up to five more steps after the timed ones: the timed region carries none)",
            "update_kernels_ms_per_step": tot_ms, "dominant_kernel": dom,
            "kernels": {k: {"launches": n, "avg_ms": ms / n} for k, (n, ms) in upd.items()},
            "stream_event_ms_per_step": r["event_ms"] / args.steps,
        }
        out["roofline_fp64"] = fp64_roofline(cells_per_s_kernel, defaults["fast_math"], dom)
        # traffic: from the committed rocprofv3 PMC passes of this same default command
        # (profiles/traffic.json), scaled to this rank's cells
        t = pmc_counts(defaults["fast_math"])
        if t and t.get("kernel") == dom:
            out["roofline"]["traffic"] = t["bytes_per_cell_update"] * r["local_cells"]
            out["roofline"]["traffic_source"] = "profiles/traffic.json: " + t["measured_at"]
        if world == 1:
            also = {}
            if not args.no_also:
                # the headline workload in the other build
                d2 = dict(defaults, fast_math=1 - defaults["fast_math"])
                r2 = bench_sedov(args, dist, ctx, device, d2, steps=max(5, args.steps // 2), warmup=2)
                also["sedov_exact" if d2["fast_math"] == 0 else "sedov_fast"] = sedov_leg(r2, d2, args.nx)
            if not args.no_developed and not args.no_also:
                tile, frac, nst = developed_tile(ctx, device)
                info = {"workload": f"compressible sedov {args.nx}x{args.nx}, DEVELOPED flow: a 1024x1024 "
                                    f"Sedov blast at t = 0.1 ({nst} steps on this GPU) tiled over the grid",
                        "shocked_cell_fraction": frac}
                rd = bench_sedov(args, dist, ctx, device, defaults, steps=args.developed_steps,
                                 warmup=5, tile=tile)
                also["sedov_developed"] = sedov_leg(rd, defaults, args.nx, info)
                also["sedov_developed"]["ratio_to_headline"] = also["sedov_developed"]["value"] / value
                d2 = dict(defaults, fast_math=1 - defaults["fast_math"])
                rd2 = bench_sedov(args, dist, ctx, device, d2, steps=max(20, args.developed_steps // 5),
                                  warmup=3, tile=tile)
                also["sedov_developed_exact" if d2["fast_math"] == 0 else "sedov_developed_fast"] = \
                    sedov_leg(rd2, d2, args.nx, info)
                del tile
            if not args.no_also and args.nx == 16384:
                # BASELINE configs[2] and north_star's target size, both builds
                also["sedov_4096"] = sedov_size_leg(args, dist, ctx, device, defaults, 4096, 100)
                also["sedov_8192"] = sedov_size_leg(args, dist, ctx, device, defaults, 8192, 40)
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline_sedov(args.cpu_sample_nx)
            if not args.no_also:
                also.update({"sedov_small_grids": bench_small_grids(ctx, device),
                             "advection": bench_advection(ctx, device, fast_math=defaults["fast_math"]),
                             "advection_8192": bench_advection(ctx, device, nx=8192, steps=30, warmup=5,
                                                               fast_math=defaults["fast_math"]),
                             "multigrid": bench_mg(ctx, device),
                             "incompressible": bench_incompressible(ctx, device)})
                out["also"] = also
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    dist.barrier()


if __name__ == "__main__":
    main()
