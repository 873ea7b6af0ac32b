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
                is 99.9 % ambient gas), 120 timed steps, both builds;
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

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # as pyro2_amd/__init__.py (before any HIP call)
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6290.0 # ... 6.29 TB/s measured (float4 copy, 79 %)
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


# DESIGN.md 6 "Predicted ... step at 16384^2": the committed prediction a first hardware
# SCALE line is read against (ms per step; slab kernel = single-GPU kernel / N x the tail of
# its ~2 rounds of wavefronts, + ~0.1 ms of dt all-reduce and small launches on the critical
# path; the halo exchange runs beside the interior strips)
PREDICTED_MS_16384 = {1: 9.42, 2: 4.75, 4: 2.45, 8: 1.40}


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
    ap.add_argument("--cpu-seconds", type=float, default=10.0,
                    help="bound of the cpu_baseline sample (seconds of one host core)")
    ap.add_argument("--also-div", type=int, default=1,
                    help="> 1: the secondary legs run on grids this many times smaller and a few steps "
                         "(tests/test_bench_line.py runs the whole script on the host emulator)")
    ap.add_argument("--developed-steps", type=int, default=120)
    ap.add_argument("--no-developed", action="store_true")
    ap.add_argument("--scale-check", action="store_true", default=None,
                    help="before timing: compressible Sedov 2048^2, 12 steps, on 1 rank vs the N ranks "
                         "of this run, must agree bit for bit (SURVEY 8(d).5); the result goes "
                         "into config.scale_check, a mismatch is fatal.  On by default at --gpus > 1")
    ap.add_argument("--no-scale-check", dest="scale_check", action="store_false",
                    help="skip the bit-identity check of the decomposed run (--gpus > 1)")
    ap.add_argument("--slab-of", type=int, default=0,
                    help="ONE GPU: time one rank's slab of a --slab-of N rank run of the headline grid "
                         "(nx / N rows x nx columns, both x neighbours = the rank itself over a 1-rank RCCL "
                         "communicator: boundary strips first, halo exchange of the new rows on the second "
                         "stream beside the interior strips, all-reduced CFL minimum, device-side stepping) "
                         "-- the critical path of the N-GPU run, measurable on one GPU; prints its own line")
    ap.add_argument("--slab-rank", type=int, default=None,
                    help="--slab-of: which rank's rows of the Sedov initial condition (default: 0, ambient gas "
                         "only, and N/2, with the blast)")
    ap.add_argument("--march-rows", type=int, default=0,
                    help="--slab-of: rows per strip of the row-marching kernel (0: the library's choice)")
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

    def gather(self, values):
        """list of floats of every rank -> list (over ranks) of lists, on every rank"""
        if not self.td:
            return [list(values)]
        import torch
        mine = torch.tensor(list(values), dtype=torch.float64)
        out = [torch.zeros_like(mine) for _ in range(self.world)]
        self.td.all_gather(out, mine)
        return [[float(v) for v in t] for t in out]

    def bcast_bytes(self, b, n):
        if not self.td:
            return b
        import torch
        t = torch.zeros(n, dtype=torch.uint8)
        if self.rank == 0:
            t = torch.frombuffer(bytearray(b), dtype=torch.uint8).clone()
        self.td.broadcast(t, 0)
        return bytes(t.numpy().tobytes())


def device_identity(index):
    """an integer naming the physical GPU behind HIP device `index` on this host (its PCI
    domain:bus:device.function), or None.  bench.py --gpus N compares the ranks' identities:
    a first SCALE line must not silently be N ranks on one device"""
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(index)) != 0:
            return None
        dom, bus, rest = buf.value.decode().split(":")
        dev, fn = rest.split(".")
        return (int(dom, 16) << 24) | (int(bus, 16) << 16) | (int(dev, 16) << 8) | int(fn, 16)
    except Exception:      # noqa: BLE001 -- the host emulator has no HIP runtime
        return None


def check_rank_devices(dist, ident, rccl_ranks):
    """every rank of a --gpus N run on a GPU of its own and all N in the communicator, else
    the run is either flagged (oversubscribed debug run) or refused"""
    ids = [int(row[0]) for row in dist.gather([float(ident if ident is not None else -1 - dist.rank)])]
    distinct = len(set(ids)) == len(ids)
    if dist.world > 1 and dist.comm_kind == "rccl" and rccl_ranks != dist.world:
        sys.exit(f"bench.py rank {dist.rank}: FATAL: the RCCL communicator has {rccl_ranks} ranks, "
                 f"--gpus says {dist.world}")
    if dist.world > 1 and not distinct and not dist.oversubscribed:
        sys.exit(f"bench.py rank {dist.rank}: FATAL: {dist.world} ranks but only {len(set(ids))} distinct "
                 f"GPUs (PCI ids {ids}): not a scaling run.  (A box with fewer GPUs than ranks is "
                 "detected and flagged as config.oversubscribed; this is a launcher / visibility problem.)")
    return ids, distinct


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


def also_traffic_source():
    """which committed PMC file the `traffic` entries of the advection / multigrid legs come
    from, with the commit / box / date recorded in it"""
    import glob
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_also_traffic.json")))
    if not fs:
        return None
    try:
        prov = json.load(open(fs[-1])).get("provenance")
    except Exception:
        prov = None
    return {"file": "profiles/" + os.path.basename(fs[-1]), "provenance": prov,
            "note": "counters of a separate rocprofv3 --pmc run of the same leg, not of this run"}


def adv_traffic(nx, steps_per_launch):
    """fabric bytes per launch of the advection kernel at this size from the committed PMC passes
    (tools/pmc_adv.sh -> profiles/r*_adv_pmc.json), if they were counted for the same number of
    steps per launch; None otherwise"""
    import glob
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_adv_pmc.json")))
    if not fs:
        return None
    try:
        e = json.load(open(fs[-1]))[str(nx)]
        if int(e["steps_per_launch"]) != int(round(steps_per_launch)):
            return None
        return e["read_bytes"] + e["write_bytes"]
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
        from host_comm import HostStagedComm      # tests/host_comm.py: the debug transport, asked for by name
        comm = HostStagedComm(dist.td)
    kw = dict(dx=1.0 / nx, dy=1.0 / ny, fast_math=defaults["fast_math"],
              kernel_set=defaults["kernel_set"], march_rows=getattr(args, "march_rows", 0))
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
    if collect:
        # scale check: the state after exactly `steps` steps (the profiling steps below are as
        # many as the elapsed time allows -- a count that differs between the N-rank run and the
        # single-domain one; found by running two ranks on one GPU)
        snap = {"interior": st.download()[ng:-ng, ng:-ng].copy(), "rows": (dec.i0, dec.nx_local),
                "t": pol.t}
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
           "rows_local": dec.nx_local,
           "prof_steps": nprof, "dt_policy": "device" if device_dt else "host"}
    if collect:
        res.update(snap)
    del slab, st
    return res


def r_short(elapsed, steps):
    """steps shorter than 5 ms: the instrumented pass takes as many steps as the timed one (a
    handful of short launches after the idle moment of the synchronisation run at lower clocks:
    4096^2 0.92 ms per launch over 5 steps against 0.68 ms by rocprofv3)"""
    return steps > 0 and elapsed / steps < 5.0e-3


def pmc_counts(fast_math, nx=16384):
    """VALU instruction / traffic counts of the dominant update kernel from the committed
    rocprofv3 PMC passes (profiles/traffic.json: tools/profile_round.sh + tools/make_traffic.py),
    counted at this grid size when the file has it, else at 16384^2 (entry "counted_at_nx"
    says which); every entry carries the commit, box and date of the session that counted it"""
    tr = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        tab = json.load(open(tr))
    except Exception:
        return None
    for key, at in ((f"nx{nx}_fast_math_{fast_math}", nx), (f"fast_math_{fast_math}", 16384)):
        if key in tab and (at == nx or key.startswith("fast_math")):
            e = dict(tab[key])
            e["counted_at_nx"] = e.get("nx", 16384)
            e.setdefault("provenance", tab.get("provenance"))
            return e
    return None


def fp64_roofline(cells_per_s_kernel, fast_math, dom, nx=16384):
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
    t = pmc_counts(fast_math, nx)
    if t and t.get("kernel") == dom:
        ipc = t["valu_insts_per_cell_update"]        # lane-instructions per cell update
        out["valu_issue"] = {
            "valu_lane_insts_per_cell_update": ipc,
            "executed_flops_per_cell_update": t.get("flops_per_cell_update"),
            "achieved_wave_insts_per_s": ipc / 64.0 * cells_per_s_kernel,
            "peak_wave_insts_per_s": VALU_ISSUE_PEAK,
            "frac": ipc / 64.0 * cells_per_s_kernel / VALU_ISSUE_PEAK,
            "valu_busy_frac_of_kernel_time": t["valu_busy_ms"] / t["kernel_ms"],
            "counted_at_nx": t["counted_at_nx"], "provenance": t.get("provenance"),
            "source": "SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU of the same kernel in a separate rocprofv3 "
                      "--pmc session (profiles/traffic.json; NOT counters of this run), " + t["measured_at"]}
    return out


def sedov_leg(r, defaults, nx, extra=None):
    """summary of a secondary Sedov measurement for the `also` block"""
    upd = {k: v for k, v in r["prof"].items() if not k.startswith("comm:")}
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
                                          defaults["fast_math"], dom, nx)}
    out["roofline"]["frac_of_achievable"] = gbs / HBM_ACHIEVABLE_GBS
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


def bench_slab(args, ctx, device, defaults, nranks, ranks=None, nx=None, steps=20, warmup=5, single_ms=None,
               march_rows=0):
    """the critical path of the N-GPU headline run on ONE GPU.  One rank's slab
    (nx / nranks rows x nx columns) steps exactly as it does in the decomposed run -- halo sides on
    both cuts, boundary strips first, the exchange of the NEW boundary rows posted on the halo
    stream / second communicator beside the interior strips, CFL minimum all-reduced on the device,
    fill + dt policy + update enqueued by pyrohip_comp_evolve -- over a ONE-rank RCCL communicator
    whose both neighbours are the rank itself (tests/test_zz_comm.py::
    test_rccl_overlapped_halo_self_neighbour: that slab is a periodic problem).  What it cannot
    show is the xGMI transfer time (2.1 MB per neighbour and direction) and the latency of an
    8-rank all-reduce; everything a rank's own GPU does per step it does.
    speedup_upper_bound = single-GPU ms per step of the whole grid / slab ms per step."""
    from pyro2_amd.compressible.problems.sedov import sedov_state
    from pyro2_amd.decomp import DtPolicy, RcclComm, SlabCompressible, SlabDecomp
    nx = args.nx if nx is None else nx
    ng = 4
    had_comm = ctx.comm_size() > 0
    if not had_comm:
        ctx.comm_init(1, 0, device.Context.comm_unique_id())
    out = {"workload": f"one rank's slab of compressible sedov {nx}x{nx} on {nranks} ranks: "
                       f"{nx // nranks} rows x {nx} columns on ONE GPU, both x neighbours = the rank itself "
                       "(1-rank RCCL communicator): boundary strips first, halo exchange on the second "
                       "stream beside the interior strips, all-reduced CFL minimum, device-side stepping",
           "ranks": {}}
    try:
        for r in (ranks if ranks is not None else (0, nranks // 2)):
            dec = SlabDecomp(nx, nranks, r, periodic=False)
            dec.lo = dec.hi = 0              # this GPU plays both neighbours
            comm = RcclComm(ctx)
            kw = dict(dx=1.0 / nx, dy=1.0 / nx, fast_math=defaults["fast_math"],
                      kernel_set=defaults["kernel_set"], march_rows=march_rows)
            slab = SlabCompressible(ctx, dec, nx, ["outflow"] * 4, kw, comm, ng=ng)
            st = slab.state
            for r0 in range(0, dec.nx_local + 2 * ng, 512):
                nr = min(512, dec.nx_local + 2 * ng - r0)
                st.upload_rows(r0, sedov_state(nx, nx, ng, 0.0, 1.0, 0.0, 1.0, 1.4, 0.01, 4,
                                               i0=dec.i0 + r0, ni=nr))
            pol = DtPolicy(tmax=1.0e9)      # (a slab of ambient gas alone takes large steps)
            assert len(slab.evolve(pol, 0.8, warmup)) == warmup
            ctx.sync()
            t0 = time.perf_counter()
            assert len(slab.evolve(pol, 0.8, steps)) == steps
            ctx.sync()
            ms = (time.perf_counter() - t0) / steps * 1e3
            ctx.prof_enable(True)
            slab.evolve(pol, 0.8, steps)
            ctx.sync()
            prof = ctx.prof_report()
            ctx.prof_enable(False)
            geo = device.comp_wave_geometry(dec.nx_local, nx, ng, ctx.info().get("compute_units", 0),
                                            march_rows)
            e = {"slab_ms_per_step": ms, "rows": dec.nx_local, "ic_rows_from": dec.i0,
                 "kernel_ms": sum(v[1] for k, v in prof.items() if not k.startswith("comm:")) / steps,
                 "halo_wait_ms": prof.get("comm:halo_wait", (0, 0.0))[1] / steps,
                 "halo_sync_ms": prof.get("comm:halo_sync", (0, 0.0))[1] / steps,
                 "allreduce_ms": prof.get("comm:allreduce_dt", (0, 0.0))[1] / steps,
                 "kernels": {k: {"launches": n, "avg_ms": t / max(n, 1)} for k, (n, t) in prof.items()},
                 "geometry": geo, "rounds": geo["wavefronts"] / max(geo["slots"], 1),
                 "overlapped": bool(geo["overlap"]), "sim_time": pol.t}
            if single_ms:
                e["speedup_upper_bound"] = single_ms / ms
            out["ranks"][str(r)] = e
            del slab, st
    finally:
        ctx.comm_set_global_dt(False)
        if not had_comm:
            ctx.comm_destroy()
    worst = max(v["slab_ms_per_step"] for v in out["ranks"].values())
    out["slab_ms_per_step"] = worst
    out["of_ranks"] = nranks
    if single_ms:
        out["single_gpu_ms_per_step"] = single_ms
        out["speedup_upper_bound"] = single_ms / worst
    out["note"] = ("slowest of the measured slabs; an upper bound of the strong-scaling factor: the xGMI "
                   "transfers and the N-rank all-reduce latency are not in it")
    return out


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
        what = "identical"
        if not same:     # where, and by how much: the first thing anyone will ask
            a, b = rn["interior"], r1["interior"][i0:i0 + n]
            w = np.argwhere(a != b)
            what = f"DIFFERENT: t {rn['t']!r} vs {r1['t']!r}; {len(w)} of {a.size} values differ"
            if len(w):
                r, c, v = (int(x) for x in w[0])
                rl, cl, vl = (int(x) for x in w[-1])
                what += (f", rows {int(w[:, 0].min())}..{int(w[:, 0].max())} of the slab's {n} (global row "
                         f"{i0} + that), columns {int(w[:, 1].min())}..{int(w[:, 1].max())}; first at "
                         f"[{r}, {c}, {v}]: {a[r, c, v]!r} vs {b[r, c, v]!r}; last at [{rl}, {cl}, {vl}]: "
                         f"{a[rl, cl, vl]!r} vs {b[rl, cl, vl]!r}")
        sys.exit(f"bench.py rank {dist.rank}: FATAL: scale check failed: sedov {nx}^2 x {steps} steps on "
                 f"{dist.world} ranks differs from the single-domain run (this rank: {what})")
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
                      f"single-threaded), {ref[section].get('date', ref['date'])}, oracle/time_reference.py -- measured in the "
                      "build container, not on the GPU box"}


def reference_numpy_stages():
    """the reference's compressible step with its njit kernels stubbed (an upper bound of its
    rate), as timed by oracle/time_reference.py in the build container"""
    try:
        ref = json.load(open(os.path.join(ROOT, "profiles", "cpu_reference.json")))
        rows = ref["compressible_numpy_stages_only"]
    except Exception:
        return None
    sizes = sorted(rows, key=int)
    return {"value_upper_bound": [rows[k]["value_upper_bound"] for k in sizes],
            "at_nx": [int(k) for k in sizes], "seconds_per_step": [rows[k]["seconds_per_step"] for k in sizes],
            "unit": "cell-updates/s", "cores": 1,
            "source": f"profiles/cpu_reference.json ({ref['date']}, {ref['cpu']}, build container): "
                      "Pyro.single_step of the reference with interface.states / riemann_hllc / "
                      "artificial_viscosity replaced by allocate-only stubs; real reference <= these rates"}


def bench_advection(ctx, device, nx=2048, steps=600, warmup=30, fast_math=1, other=True, multi_k=0):
    """advection smooth nx^2 periodic through pyrohip_adv_evolve: the driver's loop body
    (fill_BC_all + evolve) `steps` times in one call, several time steps per pass over the
    grid (k_adv_multi; multi_k = 0: the library's choice, 2).  A step moves 16 algorithmic
    bytes per cell (read a, write a) whatever the launch fuses: step_frac = 16 B x cells x
    steps / wall time; the kernel entry prices one launch = K steps."""
    x = (np.arange(nx + 8) - 3.5) / nx
    X, Y = np.meshgrid(x, x, indexing="ij")
    ic = 1.0 + np.exp(-60.0 * ((X - 0.5) ** 2 + (Y - 0.5) ** 2))
    st = device.DeviceState(ctx, nx, nx, 4, [["periodic"] * 4])
    st.upload(ic)
    dt = 0.8 * min((1 / nx) / 1.0, (1 / nx) / 1.0)     # advection/simulation.py:38-54, u = v = 1, cfl 0.8

    def run(n, k=multi_k):
        if k < 0:        # one launch per step of the single-step kernel (round 3's leg)
            for _ in range(n):
                st.adv_step(0, 1 / nx, 1 / nx, 1.0, 1.0, dt, 2, fill=True, fast_math=fast_math)
        else:
            st.adv_evolve(0, 1 / nx, 1 / nx, 1.0, 1.0, [dt] * n, 2, fast_math=fast_math, multi_k=k)
    run(warmup)
    ctx.sync()
    # a HIP-event pair on the library's stream brackets the timed region: the launches of
    # `steps` steps back to back -- elapsed / launches is the average launch duration including
    # the gap to the next launch.  (Events around EVERY launch, the second pass below, stretch
    # a short kernel by ~4 us.)
    ctx.timer_start()
    t0 = time.perf_counter()
    run(steps)
    ev_ms = ctx.timer_stop()
    ctx.sync()
    t1 = time.perf_counter()
    ctx.prof_enable(True)
    run(steps)
    ctx.sync()
    prof = ctx.prof_report()
    ctx.prof_enable(False)
    kname = "k_adv_multi" if "k_adv_multi" in prof else "k_adv_step"
    n, ms = prof[kname]
    spl = steps / n                                     # time steps per launch
    # launches of >= 100 us: the events around every launch (their ~4 us are < 4 % there, and
    # they leave out the write-back between two launches, as rocprofv3 does); shorter ones:
    # the pair around the timed region
    per_launch = ms / n >= 0.1
    kern_s = (ms / n if per_launch else ev_ms / n) * 1e-3
    traffic = adv_traffic(nx, spl)
    # the same steps one launch each (the single-step kernel: what round 3 timed)
    ctx.sync()
    s0 = time.perf_counter()
    run(steps, -1)
    ctx.sync()
    single_ms = (time.perf_counter() - s0) / steps * 1e3
    out_other = None
    if other:      # the other arithmetic (fast_math = 0: bit-faithful, the audit build)
        out_other = bench_advection(ctx, device, nx, steps, warmup, 1 - fast_math, other=False, multi_k=multi_k)
    abytes = ADV_BYTES_PER_CELL * nx * nx
    return {"workload": f"advection smooth {nx}x{nx} periodic, limiter 2, {steps} steps in one "
                        f"pyrohip_adv_evolve call ({spl:g} time steps per launch)",
            "fast_math": fast_math, "other_build": out_other,
            "value": nx * nx * steps / (t1 - t0), "unit": "cell-updates/s",
            "ms_per_step": (t1 - t0) / steps * 1e3, "steps": steps, "steps_per_launch": spl,
            "ms_per_step_one_launch_per_step": single_ms,
            "roofline": {"bound": "hbm", "kernel": kname,
                         "achieved": abytes * spl / kern_s / 1e9,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": abytes * spl / kern_s / 1e9 / HBM_PEAK_GBS,
                         "frac_of_achievable": abytes * spl / kern_s / 1e9 / HBM_ACHIEVABLE_GBS,
                         "algorithmic_bytes_per_launch": abytes * spl,
                         "kernel_avg_ms": kern_s * 1e3, "kernel_avg_ms_events_per_launch": ms / n,
                         "kernel_avg_ms_events_over_region": ev_ms / n,
                         "kernel_timer": "events per launch" if per_launch else "event pair over the timed region",
                         "traffic": traffic,
                         "traffic_source": {"file": "profiles/r*_adv_pmc.json (the last one)",
                                            "note": "counters of a separate rocprofv3 --pmc session "
                                                    "(tools/pmc_adv.sh), with its provenance in the file"},
                         "step_frac": abytes * steps / (t1 - t0) / 1e9 / HBM_PEAK_GBS,
                         "launches_per_step": 1.0 / spl,
                         "basis": "16 B per cell and time step (read a, write a) x the time steps one launch "
                                  "takes / the launch's duration; traffic = measured fabric bytes per launch "
                                  "(one read and one write of the grid per launch: below the algorithmic "
                                  "bytes from two steps per launch on)"},
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
    # warm-up: ~50 ms of V-cycles (the clock after the host-side set-up: tools/mg_warm.py; 2 cycles when
    # the caller asks for a token run)
    m.solve(rtol=0.0, max_cycles=2 if cycles < 5 else (80 if nx >= 2048 else 250))
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


def bench_pyro_run(ctx, device, solver, problem, inputs, steps, warm, model=None, inputs_file=None):
    """`steps` time steps THROUGH THE CLASS SURFACE: Pyro(solver).initialize_problem(...) and
    Pyro.run_sim() (pyro_sim.py:197-281: verbose 0, no output, no plot -- the loop the reference's
    driver runs), after `warm` untimed steps of the same object.  model = (bytes per cell and
    step, text): the algorithmic bytes the step is priced with for an HBM figure."""
    import contextlib
    import io
    import tempfile
    from pyro2_amd.pyro_sim import Pyro
    cwd = os.getcwd()
    os.chdir(tempfile.mkdtemp())          # Pyro writes inputs.auto
    old = device.Context._default
    try:
        device.Context._default = ctx
        with contextlib.redirect_stdout(io.StringIO()):
            p = Pyro(solver)
            p.initialize_problem(problem, inputs_file=inputs_file,
                                 inputs_dict=dict(inputs, **{"driver.max_steps": warm, "driver.tmax": 1.0e9}))
            p.run_sim()
            ctx.sync()
            n0, p.sim.max_steps = p.sim.n, warm + steps
            t0 = time.perf_counter()
            p.run_sim()
            ctx.sync()
            t1 = time.perf_counter()
        done = p.sim.n - n0
        g = p.sim.cc_data.grid
        cells = float(g.nx) * g.ny
        extra = {}
        if hasattr(p.sim, "mg_cycles"):
            c = p.sim.mg_cycles
            extra["vcycles_last_step"] = int(sum(c)) if hasattr(c, "__len__") else int(c)
        del p
    finally:
        device.Context._default = old
        os.chdir(cwd)
    ms = (t1 - t0) / max(done, 1) * 1e3
    out = {"workload": f"Pyro('{solver}') {problem} {g.nx}x{g.ny} through run_sim(), {done} steps "
                       f"after {warm} untimed ones", "value": cells * done / (t1 - t0),
           "unit": "cell-updates/s", "ms_per_step": ms, "steps": done}
    out.update(extra)
    if model:
        gbs = model[0] * cells / (ms * 1e-3) / 1e9
        out["roofline"] = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": gbs / HBM_PEAK_GBS, "bytes_per_cell_step": model[0], "basis": model[1]}
    return out


def bench_pyro_driver(ctx, device, bare, n_=lambda n, lo=32: n, k_=lambda k, few=3: k):
    """the two hyperbolic solvers timed through Pyro.run_sim() -- the call
    surface north_star keeps -- next to the bare C-ABI legs of this line (`bare`: ms per step
    of also.sedov_4096 / also.advection), and one-line legs of the SURVEY 8(f) solvers with
    the bytes model each is priced with (and, where oracle/time_reference.py has timed it, the
    reference itself on one host core as `cpu_baseline`)."""
    out = {}

    def leg(name, *a, bare_ms=None, ref=None, note=None, **kw):
        try:
            r = bench_pyro_run(ctx, device, *a, **kw)
        except Exception as e:      # noqa: BLE001 -- one solver's failure must not hide the others
            out[name] = {"error": f"{type(e).__name__}: {e}"}
            return
        if bare_ms:
            r["bare_c_abi_ms_per_step"] = bare_ms
            r["ratio_to_bare"] = bare_ms / r["ms_per_step"]
        if note:
            r["note"] = note
        cb = reference_baseline(*ref) if ref else None
        if cb:
            r["cpu_baseline"] = cb
        out[name] = r

    # (the same steps of the same run as the bare leg also.sedov_4096: 100 after 5 untimed ones --
    # later steps cost more, the blast grows)
    leg("compressible_sedov_4096", "compressible", "sedov", {"mesh.nx": n_(4096), "mesh.ny": n_(4096)},
        k_(100), k_(5), (SEDOV_BYTES_PER_CELL, "64 B per cell update (SURVEY 8(d))"),
        bare_ms=bare.get("sedov_4096"))
    leg("advection_smooth_2048", "advection", "smooth",
        {"mesh.nx": n_(2048), "mesh.ny": n_(2048), "particles.do_particles": 0}, k_(768, 6), k_(3072, 6),
        (ADV_BYTES_PER_CELL, "16 B per cell update"), bare_ms=bare.get("advection"),
        note="inputs.smooth carries 100 tracer particles (host-side NumPy, two grid-sized velocity "
             "arrays per call): switched off here, the leg times the grid update")
    # SURVEY 8(f) rows
    leg("diffusion_gaussian_2048", "diffusion", "gaussian", {"mesh.nx": n_(2048), "mesh.ny": n_(2048)},
        k_(10, 1), k_(2, 1),
        (16 + 24, "read + write phi (16 B) and the Crank-Nicolson right-hand side pass (24 B) per cell and "
                  "step; the multigrid solve on top is priced in also.multigrid (V-cycles per step "
                  "reported beside it)"), ref=("diffusion", "2048"))
    # (20 untimed steps in front of these legs, not 3: the problem set-up on the host leaves the GPU idle for
    # hundreds of ms, the first ~10 ms of work after that run at a lower clock -- tools/runsim_overhead.py:
    # swe 4096^2 20 steps 11.5 / 10.7 / 10.5 ms in three consecutive run_sim() calls)
    leg("swe_dam_4096", "swe", "dam", {"mesh.nx": n_(4096), "mesh.ny": n_(4096)}, k_(20), k_(20),
        (48, "read 3 + write 3 conserved doubles per cell update"), inputs_file="inputs.dam.x",
        ref=("swe", "interpreted"))
    leg("compressible_rk_sedov_2048", "compressible_rk", "sedov", {"mesh.nx": n_(2048), "mesh.ny": n_(2048)},
        k_(20), k_(3),
        (416, "RK4 in four launches (pyrohip_comp_rk_step): 64 + 96 + 96 + 160 B per cell and step, see the "
              "4096^2 leg"), ref=("compressible_rk", "interpreted"))
    leg("compressible_rk_sedov_4096", "compressible_rk", "sedov", {"mesh.nx": n_(4096), "mesh.ny": n_(4096)},
        k_(20), k_(10),
        (416, "RK4 with the stage folded into the right-hand side's load (pyrohip_comp_rk_step): stage 0 reads "
              "y_0 and writes k_0 (64 B), stages 1-2 read y_0 + one k and write a k (96 B each), the last reads "
              "y_0 + k_0..k_2 and writes the new state (160 B): 416 B per cell and step"))
    leg("compressible_sedov_spherical_2048", "compressible", "sedov",
        {"mesh.nx": n_(2048, 64), "mesh.ny": n_(2048, 64)}, k_(20), k_(20),
        (SEDOV_BYTES_PER_CELL, "64 B per cell update (round 6: the row-marching kernel with the geometry terms, "
                               "k_sph_wave: row / column factor tables instead of 8 geometry planes)"),
        inputs_file="inputs.sedov.spherical", ref=("compressible_spherical", "interpreted"))
    return out


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
        if steps >= 100:      # the clock up first (bench legs above): ~50 ms of the same steps on a twin state
            tw = device.DeviceState(ctx, nx, nx, 4, [["outflow"] * 4] * 4)
            tw.upload(sedov_state(nx, nx, 4, 0.0, 1.0, 0.0, 1.0, 1.4, 0.01, 4))
            polw = DtPolicy(1.0e9)
            for _ in range(5):
                tw.comp_evolve(P, 0.8, polw, 400)
            del tw
        st.comp_evolve(P, 0.8, pol, 50)
        ctx.sync()
        t0 = time.perf_counter()
        st.comp_evolve(P, 0.8, pol, steps)
        ctx.sync()
        us = (time.perf_counter() - t0) / steps * 1e6
        out[f"{nx}x{nx}"] = {"us_per_step": us, "value": nx * nx / (us * 1e-6), "unit": "cell-updates/s"}
    return {"workload": f"compressible sedov on small grids, {steps} steps enqueued on the device, "
                        "fast build", "sizes": out}


def cpu_baseline_sedov(sample_nx, max_seconds=12.0):
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
            # installable here, so what can be timed of the reference itself is its step with
            # the three njit kernels stubbed out (oracle/time_reference.py
            # time_compressible_numpy_stages -> profiles/cpu_reference.json): the real
            # reference is slower than that
            "reference_numpy_stages_only": reference_numpy_stages(),
            "sample": f"oracle/pyro_oracle.c (gcc -O2, 1 thread), compressible sedov "
                      f"{sample_nx}x{sample_nx}, {n} steps from t=0, {el:.1f} s; host has "
                      f"{os.cpu_count()} cores; the reference itself is single-threaded "
                      f"NumPy/numba (SURVEY 8(d))"}


LINE_LIMIT = 4096       # the driver keeps a bounded tail of stdout: the ONE line must fit it whole


def _rnd(x, sig=6):
    """floats to `sig` significant digits (the full-precision record goes to the side file)"""
    if isinstance(x, float):
        return float(f"{x:.{sig}g}")
    if isinstance(x, dict):
        return {k: _rnd(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_rnd(v, sig) for v in x]
    return x


def _pick(d, *keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_line(out, full_path=None):
    """The ONE stdout line of the bench contract, <= LINE_LIMIT bytes: headline, roofline,
    cpu_baseline, and `targets` = one figure per secondary leg (round 4's
    30 KB line did not survive the driver's tail buffer).  Everything else -- every `also` leg
    with its notes, per-kernel tables, provenance -- goes to the side file / stderr."""
    line = _pick(out, "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                 "higher_is_better", "scaling", "dtype", "data")
    line["vs_baseline"] = out.get("vs_baseline")
    cfg = out.get("config", {})
    c = _pick(cfg, "workload", "parallelism", "halo", "rccl_ranks", "fast_math", "kernel_set",
              "sim_time", "oversubscribed", "halo_note")
    if len(c.get("workload", "")) > 200:
        c["workload"] = c["workload"][:197] + "..."
    c["dt_policy"] = str(cfg.get("dt_policy", "")).split(" (")[0]
    c["state"] = str(cfg.get("state", "")).split(" (")[0]
    sc = cfg.get("scale_check")
    if isinstance(sc, dict):
        c["scale_check"] = _pick(sc, "ok", "bit_identical", "nx", "steps", "max_abs_diff", "ranks")
    elif sc is not None:
        c["scale_check"] = sc
    line["config"] = c
    rf = out.get("roofline")
    if rf:
        dom = rf.get("dominant_kernel")
        r = _pick(rf, "bound", "achieved", "peak", "unit", "frac", "frac_of_achievable", "traffic",
                  "update_kernels_ms_per_step", "stream_event_ms_per_step")
        r["kernel"] = dom
        r["kernel_avg_ms"] = (rf.get("kernels", {}).get(dom) or {}).get("avg_ms")
        r["kernel_timer"] = "HIP events per launch, instrumented pass after the timed steps"
        ts = rf.get("traffic_source")
        if ts:
            r["traffic_source"] = {"file": ts.get("file"), "counted_at_nx": ts.get("counted_at_nx"),
                                   "commit": (ts.get("provenance") or {}).get("commit"),
                                   "note": "separate rocprofv3 --pmc session, not counters of this run"}
        r.setdefault("traffic", None)
        line["roofline"] = r
    f64 = out.get("roofline_fp64")
    if f64:
        vi = f64.get("valu_issue") or {}
        line["roofline_fp64"] = {"bound": "fp64_valu_issue", "frac": vi.get("frac"),
                                 "valu_lane_insts_per_cell_update": vi.get("valu_lane_insts_per_cell_update"),
                                 "valu_busy": vi.get("valu_busy_frac_of_kernel_time"),
                                 "flop_frac": f64.get("frac"), "counted_at_nx": vi.get("counted_at_nx"),
                                 "commit": (vi.get("provenance") or {}).get("commit")}
    cb = out.get("cpu_baseline")
    if cb:
        b = _pick(cb, "value", "unit", "cores", "kind")
        b["sample"] = str(cb.get("sample", ""))[:160]
        ref = cb.get("reference_numpy_stages_only")
        if ref:
            b["reference_upper_bound"] = {"value": ref["value_upper_bound"][-1], "at_nx": ref["at_nx"][-1],
                                          "cores": ref.get("cores"), "kind": "reference, njit kernels stubbed"}
        line["cpu_baseline"] = b
    rk = out.get("ranks")
    if rk:
        t = rk.get("per_rank", {})
        line["ranks"] = {k: t[k] for k in ("kernel_ms", "halo_wait_ms", "halo_sync_ms", "allreduce_ms",
                                           "stream_ms_per_step") if k in t}
        line["ranks"]["wavefronts"] = rk.get("max", {}).get("wavefronts")
        line["ranks"]["predicted_ms_per_step"] = rk.get("predicted_ms_per_step")
        if "gpu_unique_ids_distinct" in rk:
            line["ranks"]["gpu_unique_ids_distinct"] = rk["gpu_unique_ids_distinct"]
    pdn = out.get("pyro_driver")
    if isinstance(pdn, dict):
        line["pyro_driver"] = _pick(pdn, "ms_per_step", "value", "ratio_to_bare") if "error" not in pdn \
            else {"error": str(pdn["error"])[:80]}
    also = out.get("also")
    if also:
        line["targets"] = targets_of(also)
    if full_path:
        line["full_record"] = full_path
    line = _rnd(line)
    s = json.dumps(line, separators=(",", ":"))
    # belt and braces: shed the optional parts until the line fits
    for k in ("targets", "ranks", "roofline_fp64"):
        if len(s) < LINE_LIMIT:
            break
        if k == "targets" and isinstance(line.get("targets"), dict):
            line["targets"] = {a: _pick(b, "value", "frac", "step_frac", "ms_per_step")
                               for a, b in line["targets"].items()}
        else:
            line.pop(k, None)
        s = json.dumps(line, separators=(",", ":"))
    assert len(s) < LINE_LIMIT, len(s)
    return s


def targets_of(also):
    """one compact entry per secondary leg: the figure VERDICT / north_star quote it by"""
    t = {}

    def leg(name, d, *keys):
        if isinstance(d, dict) and "error" not in d:
            e = _pick(d, *keys)
            rf = d.get("roofline") or {}
            for k in ("frac", "step_frac", "kernel_avg_ms"):
                if k in rf and rf[k] is not None:
                    e[k] = rf[k]
            if e:
                t[name] = e
        elif isinstance(d, dict):
            t[name] = {"error": str(d["error"])[:80]}

    for name in ("sedov_exact", "sedov_fast", "sedov_developed", "sedov_4096", "sedov_8192"):
        leg(name, also.get(name), "value", "ms_per_step")
    for name in ("sedov_developed", "sedov_4096", "sedov_8192"):
        ob = (also.get(name) or {}).get("other_build")
        if isinstance(ob, dict) and name in t and "error" not in t[name]:
            t[name]["exact_value" if ob.get("fast_math") == 0 else "fast_value"] = ob.get("value")
    leg("advection_2048", also.get("advection"), "value", "ms_per_step")
    leg("advection_8192", also.get("advection_8192"), "value", "ms_per_step")
    mg = also.get("multigrid")
    if isinstance(mg, dict):
        leg("mg_4096", mg, "value", "unit", "ms_per_vcycle", "ms_per_step")
    leg("incompressible_2048", also.get("incompressible"), "value", "ms_per_step")
    sl = also.get("slab_of_8")
    if isinstance(sl, dict):
        if "error" in sl:
            t["slab_of_8"] = {"error": str(sl["error"])[:80]}
        else:
            w = max(sl["ranks"].values(), key=lambda v: v["slab_ms_per_step"])
            t["slab_of_8"] = dict(_pick(sl, "slab_ms_per_step", "speedup_upper_bound"),
                                  **_pick(w, "halo_wait_ms", "kernel_ms", "rounds"))
    pd = also.get("pyro_driver")
    if isinstance(pd, dict):
        if "error" in pd:
            t["pyro_driver"] = {"error": str(pd["error"])[:80]}
        else:
            t["pyro_driver"] = {k: _pick(v, "ms_per_step", "value", "ratio_to_bare", "frac", "step_frac")
                                for k, v in pd.items() if isinstance(v, dict)}
    return t


def emit(out, json_fd):
    """full record -> bench_full.json (under gpurun_out/ when that exists or can be made) and,
    leg by leg, stderr; the compact line -> stdout, last"""
    full_path = None
    for d in (os.path.join(ROOT, "gpurun_out"), ROOT, "/tmp"):
        try:
            os.makedirs(d, exist_ok=True)
            full_path = os.path.join(d, "bench_full.json")
            with open(full_path, "w") as f:
                json.dump(out, f, indent=1)
            break
        except OSError:
            full_path = None
    for k, v in (out.get("also") or {}).items():
        print(f"[bench also] {k}: {json.dumps(_rnd(v, 5))}", file=sys.stderr)
    sys.stderr.flush()
    rel = os.path.relpath(full_path, ROOT) if full_path and full_path.startswith(ROOT) else full_path
    os.write(json_fd, (compact_line(out, rel) + "\n").encode())


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
    # several ranks on one GPU?  Decided by the devices the ranks actually get (PCI ids gathered
    # over the process group), not by the device count one rank sees: a launcher that gives every
    # rank ONE visible device (ROCR / HIP_VISIBLE_DEVICES per rank) is a proper one-rank-per-GPU
    # run.  Without PCI ids (no HIP runtime answer) the device count decides, as before.
    my_ident = device_identity(dist.local_rank % max(ndev, 1))
    all_ids = [int(r[0]) for r in dist.gather([float(my_ident if my_ident is not None else -1)])]
    if world > 1 and all(i >= 0 for i in all_ids):
        dist.oversubscribed = len(set(all_ids)) < world
    else:
        dist.oversubscribed = world > 1 and world > ndev
    if dist.oversubscribed and dist.rank == 0:
        print(f"[bench] WARNING: --gpus {world} on a box with {ndev} GPU(s): ranks share GPUs; "
              "this run checks the multi-rank path, its number is NOT a scaling result "
              "(flagged as config.oversubscribed)", file=sys.stderr)
    # one rank per GPU; several ranks on one GPU only happens when debugging
    # the launcher on a smaller box and is flagged in the output.  RCCL refuses two ranks on
    # one device of one host ("Duplicate GPU detected"): each rank then names itself a host of
    # its own (NCCL_HOSTID), and the communicator runs over the socket transport on lo -- the
    # same RCCL calls, streams and events as on a node, none of its bandwidth
    if dist.oversubscribed:
        os.environ.setdefault("NCCL_HOSTID", f"pyro2amd-rank{dist.rank}")
        os.environ.setdefault("NCCL_IB_DISABLE", "1")
        os.environ.setdefault("NCCL_P2P_DISABLE", "1")
    ctx = device.Context(dist.local_rank % ndev)
    ident = my_ident
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
    # N ranks = N distinct GPUs in one communicator, checked BEFORE anything is timed
    dev_ids, dev_distinct = check_rank_devices(dist, ident, getattr(dist, "rccl_ranks", None))
    # default: the contracted / reciprocal-division build, parity-tested to the
    # north_star tolerance (1e-10); --fast-math 0 times the bit-faithful build.
    # kernel_set -1: the library picks (row-marching wavefront kernel from 2048^2 on)
    defaults = {"fast_math": 1 if args.fast_math is None else args.fast_math,
                "kernel_set": -1 if args.kernel_set is None else args.kernel_set}
    if world > 1 and dist.comm_kind == "rccl":
        # pyro's class surface steps its slab of ONE problem over this communicator
        # (pyro2_amd.decomp: grid_setup hands out slabs, fill_BC_all exchanges halo rows)
        from pyro2_amd import decomp
        device.Context._default = ctx
        _rc = decomp.RcclComm(ctx)
        ctx.comm_set_global_dt(False)
        decomp.set_decomposition(_rc, dist.rank, world)
    if args.slab_of > 1:
        if world != 1:
            sys.exit("bench.py: --slab-of measures one slab on ONE GPU (--gpus 1)")
        rs = None if args.slab_rank is None else (args.slab_rank,)
        r1 = None
        if not args.no_also:       # the whole grid on this GPU: the numerator of the bound
            r1 = bench_sedov(args, dist, ctx, device, defaults)
        o = bench_slab(args, ctx, device, defaults, args.slab_of, rs, steps=args.steps, warmup=args.warmup,
                       single_ms=(r1["elapsed"] / args.steps * 1e3) if r1 else None,
                       march_rows=args.march_rows)
        os.write(json_fd, (json.dumps(_rnd(o), separators=(",", ":")) + "\n").encode())
        return

    # a decomposed run is checked against the single-domain one BEFORE anything is timed,
    # unless the caller says --no-scale-check
    do_check = world > 1 and args.scale_check is not False
    check = scale_check(args, dist, ctx, device, defaults) if do_check else None
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
                               + ("one GPU, single domain (no decomposition, no halo exchange)" if world == 1
                                  else f"x-slab decomposed over {world} GPUs, "
                                  + ("RCCL halo exchange" if dist.comm_kind == "rccl"
                                     else f"HOST-STAGED halo exchange ({dist.comm_note or 'debug path'}): "
                                          "NOT a scaling result")),
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
    # every rank's own figures (kernel time, what the main stream waited for), gathered: a
    # first hardware scaling line must say WHERE the time of a step went
    geo = device.comp_wave_geometry(r["rows_local"], args.nx, 4, ctx.info().get("compute_units", 0))
    pr, nst = r["prof"], max(r["prof_steps"], 1)
    per_rank = dist.gather([
        sum(ms for k, (_, ms) in pr.items() if not k.startswith("comm:")) / nst,
        pr.get("comm:halo_wait", (0, 0.0))[1] / nst, pr.get("comm:halo_sync", (0, 0.0))[1] / nst,
        pr.get("comm:allreduce_dt", (0, 0.0))[1] / nst, r["event_ms"] / max(args.steps, 1),
        float(r["rows_local"]), float(geo["col_strips"]), float(geo["rows_per_strip"]),
        float(geo["row_strips"]), float(geo["overlap"]), float(geo["wavefronts"])])
    if world > 1:
        cols = ("kernel_ms", "halo_wait_ms", "halo_sync_ms", "allreduce_ms", "stream_ms_per_step",
                "rows", "col_strips", "rows_per_strip", "row_strips", "overlap", "wavefronts")
        table = {c: [row[i] for row in per_rank] for i, c in enumerate(cols)}
        out["ranks"] = {
            "per_rank": table,
            "min": {c: min(v) for c, v in table.items()}, "max": {c: max(v) for c, v in table.items()},
            "note": "per step, from the library's HIP events in the instrumented pass after the timed "
                    "one: kernel_ms = this rank's kernels; halo_wait_ms = the main stream standing "
                    "still for the posted halo exchange (0 when it finished beside the interior "
                    "strips); halo_sync_ms = exchanges on the main stream (no overlap possible); "
                    "allreduce_ms = the dt all-reduce; stream_ms_per_step = event pair over the timed "
                    "region / steps; the rest = pyrohip_comp_wave_geometry of the rank's slab",
            "gpu_pci_ids": ["%x" % i if i >= 0 else None for i in dev_ids],
            "gpu_unique_ids_distinct": dev_distinct,
            "predicted_ms_per_step": PREDICTED_MS_16384.get(world) if args.nx == 16384 else None,
            "predicted_source": "DESIGN.md 6 (kernel / N x tail + ~0.1 ms all-reduce and small launches)"}
    if dist.rank == 0:
        # roofline of the update kernels: algorithmic bytes of ONE rank's slab
        # per step / HIP-event time of that rank's kernels per step
        prof = r["prof"]
        upd = {k: v for k, v in prof.items() if not k.startswith("comm:")}
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
        out["roofline"]["frac_of_achievable"] = gbs / HBM_ACHIEVABLE_GBS
        out["roofline_fp64"] = fp64_roofline(cells_per_s_kernel, defaults["fast_math"], dom, args.nx)
        # traffic: from the committed rocprofv3 PMC passes of this same command
        # (profiles/traffic.json), scaled to this rank's cells -- counters of a SEPARATE
        # session, whose commit / box / date ride along
        t = pmc_counts(defaults["fast_math"], args.nx)
        if t and t.get("kernel") == dom:
            out["roofline"]["traffic"] = t["bytes_per_cell_update"] * r["local_cells"]
            out["roofline"]["traffic_source"] = {
                "file": "profiles/traffic.json", "counted_at_nx": t["counted_at_nx"],
                "measured_at": t["measured_at"], "provenance": t.get("provenance"),
                "note": "fabric bytes per cell update of a separate rocprofv3 --pmc session x the "
                        "cells of this run; not counters of this run"}
        if world == 1:
            also, legs_s = {}, {}
            D = max(1, args.also_div)      # > 1: every secondary leg on a grid D times smaller,
                                           # a few steps (the CPU test of this script's output)

            def n_(nx, lo=32):
                return nx if D == 1 else max(lo, nx // D)

            def k_(steps, few=3):
                return steps if D == 1 else min(steps, few)

            def leg(name, thunk):
                """a secondary leg must not cost the headline: errors are recorded, not raised"""
                t0 = time.perf_counter()
                try:
                    also[name] = thunk()
                except Exception as e:      # noqa: BLE001
                    also[name] = {"error": f"{type(e).__name__}: {e}"}
                legs_s[name] = round(time.perf_counter() - t0, 2)

            if not args.no_cpu_baseline:
                t0 = time.perf_counter()
                out["cpu_baseline"] = cpu_baseline_sedov(args.cpu_sample_nx, args.cpu_seconds)
                legs_s["cpu_baseline"] = round(time.perf_counter() - t0, 2)
            if not args.no_also:
                d2 = dict(defaults, fast_math=1 - defaults["fast_math"])
                other = "sedov_exact" if d2["fast_math"] == 0 else "sedov_fast"
                # the headline workload in the other build
                leg(other, lambda: sedov_leg(bench_sedov(args, dist, ctx, device, d2, steps=max(5, args.steps // 2),
                                                         warmup=2), d2, args.nx))
                if not args.no_developed:
                    def developed():
                        tile, frac, nst = developed_tile(ctx, device, n_(1024), 0.1 if D == 1 else 0.004)
                        info = {"workload": f"compressible sedov {args.nx}x{args.nx}, DEVELOPED flow: a "
                                            f"{tile.shape[0]}x{tile.shape[0]} Sedov blast ({nst} steps on this "
                                            "GPU) tiled over the grid", "shocked_cell_fraction": frac}
                        rd = bench_sedov(args, dist, ctx, device, defaults, steps=k_(args.developed_steps),
                                         warmup=k_(5), tile=tile)
                        a = sedov_leg(rd, defaults, args.nx, info)
                        a["ratio_to_headline"] = a["value"] / value
                        rd2 = bench_sedov(args, dist, ctx, device, d2, steps=k_(max(10, args.developed_steps // 5)),
                                          warmup=k_(3), tile=tile)
                        a["other_build"] = sedov_leg(rd2, d2, args.nx, info)
                        return a
                    leg("sedov_developed", developed)
                if args.nx == 16384 or D > 1:
                    # BASELINE configs[2] and north_star's target size, both builds
                    leg("sedov_4096", lambda: sedov_size_leg(args, dist, ctx, device, defaults, n_(4096), k_(100)))
                    leg("sedov_8192", lambda: sedov_size_leg(args, dist, ctx, device, defaults, n_(8192), k_(40)))
                leg("sedov_small_grids", lambda: bench_small_grids(
                    ctx, device, sizes=(64, 256, 512) if D == 1 else (32, 64), steps=k_(400)))
                # (~50 ms of untimed steps in front of the memory-bound legs: after the host-side set-up of a
                # leg the GPU's clock takes tens of ms of work to come back up -- tools/adv_warm.py, mg_warm.py:
                # advection 2048^2 17.7 / 17.4 / 16.7 / 15.7 us per step after 6 / 60 / 600 / 3000 untimed steps,
                # 8192^2 0.197 -> 0.165 ms, a 4096^2 V-cycle 0.719 -> 0.681 ms; a run of thousands of steps
                # sees the warm figure.  The FP64-bound Sedov legs go the other way under sustained load
                # (profiles/r06_sedov4096_by_launch.txt) and keep their windows.)
                leg("advection", lambda: bench_advection(ctx, device, nx=n_(2048), steps=k_(600, 6), warmup=k_(3000, 6),
                                                         fast_math=defaults["fast_math"]))
                leg("advection_8192", lambda: bench_advection(ctx, device, nx=n_(8192), steps=k_(60, 6),
                                                              warmup=k_(300, 6), fast_math=defaults["fast_math"]))
                leg("multigrid", lambda: bench_mg(ctx, device, nx=n_(4096), cycles=k_(10), small_sizes=D == 1))
                leg("incompressible", lambda: bench_incompressible(ctx, device, nx=n_(2048), steps=k_(5, 1)))
                bare = {"sedov_4096": (also.get("sedov_4096") or {}).get("ms_per_step"),
                        "advection": (also.get("advection") or {}).get("ms_per_step")}
                leg("pyro_driver", lambda: bench_pyro_driver(ctx, device, bare, n_, k_))
                if args.nx == 16384 or D > 1:
                    # the 8-GPU run's critical path, measured on this one GPU (last: it brings up
                    # a communicator and takes it down again)
                    leg("slab_of_8", lambda: bench_slab(args, ctx, device, defaults, 8, nx=n_(args.nx, 256),
                                                        steps=k_(20), warmup=k_(5),
                                                        single_ms=out["ms_per_step"] if D == 1 else None))
                out["also"] = also
            out["seconds_by_leg"] = legs_s
    if world > 1 and dist.comm_kind == "rccl" and not args.no_also:
        # the SAME workload through pyro's class surface -- every rank
        # runs Pyro("compressible").initialize_problem("sedov") + run_sim() and steps ITS slab
        # (COLLECTIVE: all ranks take part).  The headline is measured and its record complete by
        # now; a watchdog prints it without this leg if the leg does not come back (a collective
        # that hangs cannot be interrupted from Python: the threads of a blocked rank still run)
        import threading

        def give_up():
            if dist.rank == 0:
                out["pyro_driver"] = {"error": "no result after 240 s: leg abandoned, headline kept"}
                emit(out, json_fd)
            os._exit(0)
        dog = threading.Timer(240.0, give_up)
        dog.daemon = True
        dog.start()
        try:
            pr_ = bench_pyro_run(ctx, device, "compressible", "sedov",
                                 {"mesh.nx": args.nx, "mesh.ny": args.nx, "gpu.fast_math": defaults["fast_math"],
                                  "gpu.kernel_set": defaults["kernel_set"]}, args.steps, args.warmup)
            ms_ = dist.max(pr_["ms_per_step"])
            out["pyro_driver"] = {
                "workload": f"Pyro('compressible') sedov {args.nx}x{args.nx} through run_sim() on {world} "
                            f"processes: ONE problem in x-slabs behind the class surface, {args.steps} steps "
                            f"after {args.warmup} untimed ones", "ms_per_step": ms_,
                "value": float(args.nx) * args.nx / (ms_ * 1e-3), "unit": "cell-updates/s",
                "bare_c_abi_ms_per_step": out["ms_per_step"], "ratio_to_bare": out["ms_per_step"] / ms_}
        except Exception as e:      # noqa: BLE001 -- recorded, never fatal for the headline
            out["pyro_driver"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        dog.cancel()
    if dist.rank == 0:
        emit(out, json_fd)
    dist.barrier()


if __name__ == "__main__":
    main()
