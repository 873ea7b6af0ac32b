/*
 * pyrohip.h -- C ABI of libpyrohip.so, the MI355X (gfx950) implementation of
 * pyro2's per-timestep hot path.
 *
 * pyro2 has no FFI/plugin layer of its own: its operator API
 * is the Python class surface that pyro_sim.py drives.  Each entry point below
 * therefore cites the reference method whose arithmetic it replaces; the
 * Python-side binding (ctypes) that keeps pyro's class surface lives in
 * pyro2_amd/ and is described in INTEGRATION.md.
 *
 * Conventions
 *   - every function returns int: 0 = ok, otherwise a hipError_t / own code;
 *     pyrohip_last_error() returns the text of the last failure (per thread).
 *   - opaque handles; the library owns all device memory.
 *   - host pointers are borrowed for the duration of the call only.
 *   - all floating point data is IEEE double (the reference is float64).
 *   - host arrays use the reference layouts: scalar fields (qx,qy) C order
 *     (j fastest); cell data (qx,qy,nvar) C order (patch.py:450-452).  On the
 *     device every variable is a separate plane (SoA), j fastest.
 *   - calls on one context are serialised by the caller (one stream per ctx).
 *   - no torch types, no C++ types: plain pointers and sizes.
 */
#ifndef PYROHIP_H
#define PYROHIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pyrohip_ctx pyrohip_ctx;
typedef struct pyrohip_state pyrohip_state;
typedef struct pyrohip_mg pyrohip_mg;

/* boundary-condition codes (pyro/mesh/boundary.py:10-17 names) */
enum {
    PYROHIP_BC_OUTFLOW = 0,      /* "outflow", homogeneous "neumann"          */
    PYROHIP_BC_REFLECT_EVEN = 1, /* "reflect-even"                            */
    PYROHIP_BC_REFLECT_ODD = 2,  /* "reflect-odd", homogeneous "dirichlet"    */
    PYROHIP_BC_PERIODIC = 3,     /* "periodic"                                */
    PYROHIP_BC_HALO = 4,         /* interior slab interface: filled by        */
                                 /* pyrohip_halo_exchange, not by fill_bc     */
    PYROHIP_BC_HSE = 5,          /* compressible "hse" user boundary, y sides */
    PYROHIP_BC_AMBIENT = 6,      /* compressible "ambient" user boundary, yr  */
    PYROHIP_BC_RAMP = 7,         /* compressible "ramp" (double Mach           */
                                 /* reflection) user boundary: xl, yl, yr     */
    PYROHIP_BC_CONST = 8         /* ghost cells = a constant: "moving_lid" of */
                                 /* incompressible_viscous/BC.py:9-50 (yr);   */
                                 /* in a pyrohip_mg the constant is 0         */
};

/* own status codes (hipError_t values are passed through unchanged) */
enum {
    PYROHIP_OK = 0,
    PYROHIP_ERR_ARG = 10001,
    PYROHIP_ERR_STATE = 10002,    /* e.g. negative density / internal energy */
    PYROHIP_ERR_UNSUPPORTED = 10003,
    PYROHIP_ERR_COMM = 10004
};

/* ---- context ---------------------------------------------------------- */
int pyrohip_device_count(int *count);   /* visible HIP devices */
int pyrohip_init(int device_id, pyrohip_ctx **out);
int pyrohip_shutdown(pyrohip_ctx *ctx);
int pyrohip_sync(pyrohip_ctx *ctx);
const char *pyrohip_last_error(void);
/* "hip-gfx950" for the real library.  The host-side binding refuses anything
   else unless a test explicitly injects another backend. */
const char *pyrohip_backend(void);
int pyrohip_device_info(pyrohip_ctx *ctx, char *name, int name_len,
                        size_t *free_bytes, size_t *total_bytes,
                        int *compute_units);
/* HIP-event timing on the context's stream (bench.py roofline leg) */
int pyrohip_timer_start(pyrohip_ctx *ctx);
int pyrohip_timer_stop(pyrohip_ctx *ctx, double *elapsed_ms);

/* per-kernel HIP-event timing of the solver kernels.  enable(1) starts
   recording one event pair per launch; report() synchronises and writes
   lines "kernel_name launches total_ms\n" into buf, then clears. */
int pyrohip_prof_enable(pyrohip_ctx *ctx, int on);
int pyrohip_prof_report(pyrohip_ctx *ctx, char *buf, int buf_len);

/* ---- cell-centred data: CellCenterData2d storage (patch.py:315-794) ---- */
/* bc: nvar*4 codes, order per variable: xl, xr, yl, yr                     */
int pyrohip_state_create(pyrohip_ctx *ctx, int nx, int ny, int ng, int nvar,
                         const int *bc, pyrohip_state **out);
int pyrohip_state_destroy(pyrohip_state *s);
/* whole (qx,qy,nvar) array, reference layout (CellCenterData2d.data) */
int pyrohip_state_upload(pyrohip_state *s, const double *host_aos);
int pyrohip_state_download(pyrohip_state *s, double *host_aos);
/* one variable, (qx,qy) */
int pyrohip_state_upload_var(pyrohip_state *s, int n, const double *host);
int pyrohip_state_download_var(pyrohip_state *s, int n, double *host);
/* rows [i0, i0+ni) of every variable <-> host (ni,qy,nvar); used to stream
   large initial conditions and slab gathers */
int pyrohip_state_upload_rows(pyrohip_state *s, int i0, int ni,
                              const double *host_aos);
int pyrohip_state_download_rows(pyrohip_state *s, int i0, int ni,
                                double *host_aos);
/* ArrayIndexer.fill_ghost / CellCenterData2d.fill_BC(_all)
   (array_indexer.py:150-274, patch.py:575-624); n = -1: all variables */
int pyrohip_fill_bc(pyrohip_state *s, int n);
/* Parameters of the compressible solver's user boundaries, compressible/BC.py:
   21-176 ("hse": hydrostatic pressure integrated into the y ghost cells at
   constant density; "ambient": fixed rho,u,v,p above the upper y boundary).
   The state must hold the 4 conserved variables in pyro's order.  With these
   codes fill_bc(-1) fills variable after variable like fill_BC_all
   (patch.py:575-624): the hse energy ghosts are computed from the momenta's x
   ghost columns as the PREVIOUS fill left them.  ambient = rho,u,v,p or NULL */
int pyrohip_state_set_user_bc(pyrohip_state *s, double gamma, double grav,
                              double dy, const double *ambient);
/* Heating profile exp(-(dist/r)^2) of the problem source above: (qx, qy) host
   array on the whole grid (ghost coordinates included), or NULL to remove it.
   The ghost cells of the device copy are refilled like the reference's E_src
   (the boundary types of the energy, plain copies for hse / ambient).        */
int pyrohip_state_set_heating(pyrohip_state *s, const double *profile);
/* ghost value of variable n on its PYROHIP_BC_CONST side (upper y side only:
   "moving_lid", incompressible_viscous/BC.py:31-42; default 0)             */
int pyrohip_state_set_const_bc(pyrohip_state *s, int n, double value);
/* SphericalPolar grid (pyro/mesh/patch.py:242-312; x = r, y = theta) for the
   compressible solver: the grid's arrays Lx, Ly, Ax, Ay, V, dlogAx, dlogAy, x2d,
   each (qx, qy) row-major and evaluated by the caller with the reference's
   expressions, and the qy sines that artificial_viscosity evaluates
   (compressible/interface.py:345-347): sint[j] = sin((j + 1/2 - ng) dy + ymin),
   sinb[j] = sin((j - 1/2 - ng) dy + ymin), sinc[j] = sin((j - ng) dy + ymin).
   All arrays are copied.  With a geometry set, pyrohip_comp_dt / _step follow the
   coord_type == 1 branches of compressible/simulation.py:117-147, 284-288,
   330-398, unsplit_fluxes.py:411-488, interface.py:215-234, 331-376 and
   riemann.py:1156-1171 (CGF solver only, like the reference); NULL removes it.
   pyrohip_comp_step then is one launch (the 2-d tile kernel with the geometry terms) where
   the sides are outflow / reflect / periodic, the staged kernel set with its stage dumps
   for kernel_set 0 and any other boundary (same arithmetic: bit-identical).            */
typedef struct pyrohip_geom {
    const double *Lx, *Ly, *Ax, *Ay, *V, *dlogAx, *dlogAy, *x2d;
    const double *sint, *sinb, *sinc;
    double xmin, ymin;
    /* optional (may be NULL): the 1-d factors the 2-d arrays above are products of, evaluated by
       the caller with the reference's expressions (mesh/patch.py:262-312) so that the kernels can
       rebuild every geometry value in registers, bit for bit, instead of reading eight planes:
         rowf: 7 arrays of qx doubles  A = -2 pi xl^2, D = xr^2 - xl^2, F = xr - xl,
               G = xr^2 + xl^2 + xr xl, Ly = x dy, dlogAx = 2 / x, x
         colf: 4 arrays of qy doubles  B = cos(yr) - cos(yl), C = pi sin(yl), E = (-2 pi / 3) B, T = tan(y)
       with Ax = |A B|, Ay = |C D|, V = |(E F) G|, dlogAy = 1 / (T x), Lx = dx.               */
    const double *rowf, *colf;
} pyrohip_geom;
int pyrohip_state_set_geometry(pyrohip_state *s, const pyrohip_geom *g);
/* Parameters of the "ramp" boundary of the double Mach reflection problem
   (compressible/BC.py:178-296).  x: the qx cell-centre coordinates of the grid
   (copied); cxoff = 0.5 dx sqrt(3); post / pre: post- and pre-shock values of
   the 4 conserved variables in the state's order; sf_down / sf_up: the shock
   front positions (BC.py:240-243) of the ng ghost rows above the upper y
   boundary at the CURRENT time -- call again whenever t changes.  All
   transcendental factors are evaluated by the caller, so the fill is
   bit-identical to the reference.                                          */
int pyrohip_state_set_ramp_bc(pyrohip_state *s, const double *x, double cxoff,
                              const double *post, const double *pre,
                              const double *sf_down, const double *sf_up);
/* min / max over the valid region grown by buf (patch.py:626-638) */
int pyrohip_state_minmax(pyrohip_state *s, int n, int buf, double *vmin,
                         double *vmax);

/* ---- advection: Simulation.evolve (advection/simulation.py:56-94) with
        advective_fluxes.unsplit_fluxes (:1-92), interface.linear_interface
        (:4-43) and reconstruction.limit (reconstruction.py:9-120) fused ---- */
int pyrohip_adv_step(pyrohip_state *s, int n, double dx, double dy, double u,
                     double v, double dt, int limiter);
/* the same with the ghost fill of variable n (ArrayIndexer.fill_ghost,
   array_indexer.py:150-274; outflow / reflect-even / reflect-odd / periodic
   sides only) folded into the step when fill != 0: ONE launch does what
   pyrohip_fill_bc(s, n) + pyrohip_adv_step do, the ghost cells of the result
   hold the filled values of the old time level like after the in-place update
   of the reference.  fill = 0: ghost cells are used as they are. */
int pyrohip_adv_step_fill(pyrohip_state *s, int n, double dx, double dy, double u,
                          double v, double dt, int limiter, int fill);
/* (after a step with the fill folded in -- here and in pyrohip_comp_step with
   pyrohip_comp_params.fuse_fill -- the ghost cells IN MEMORY hold what the fill at the
   START of that step gave them, like the reference's array after evolve(); they are not
   the fill of the new state: call pyrohip_fill_bc before reading ghost cells on the host) */
/* the same with the parameters in a struct.
   fast_math   0: bit-faithful arithmetic (the reference's operation order, no
               contraction: results identical to NumPy, what the two functions above
               run); 1: the contracted build (fused multiply-adds; north_star's
               tolerance for advection, 1e-12 relative, parity-tested)
   march_rows  rows per strip of the row-marching kernel (0: chosen by the library
               from the grid size and the number of compute units)            */
typedef struct {
    double dx, dy, u, v;
    int limiter;       /* 0 none, 1 MC 2nd order, 2 MC 4th order (advection.limiter) */
    int fill;          /* fold the ghost fill into the step (see above) */
    int fast_math;
    int march_rows;
    int multi_k;       /* pyrohip_adv_evolve: time steps per launch on periodic grids
                          (0: the library's choice = 3 for u > 0, 2 for u < 0; 1: one step per launch of
                          the several-steps kernel; at most 3)                          */
    int multi_prio;    /* ... its wavefronts take turns at the priority levels (0: the
                          library's choice = yes, -1: no)                           */
} pyrohip_adv_params;
int pyrohip_adv_step_p(pyrohip_state *s, int n, const pyrohip_adv_params *p, double dt);
/* nsteps iterations of the driver's loop body for the advection solver
   (pyro_sim.py:250-256: fill_BC_all, then Simulation.evolve, advection/simulation.py:56-94)
   with the time steps dts[0 .. nsteps) the driver's policy gave (simulation_null.py:222-244;
   the advective CFL step is closed-form, advection/simulation.py:38-54, so the caller knows
   them all beforehand).  Nothing reads the data between the steps, so on periodic grids
   several steps are taken in ONE pass over the grid (time-skewed row march: intermediate
   time levels stay in registers); the bit-faithful build gives, bit for bit, what nsteps
   calls of pyrohip_adv_step_fill(fill = 1) give, ghost frame included.  Outflow / reflect
   sides, u = 0 or v = 0 and slabs with neighbours take one launch per step.  p->fill is
   ignored (every step fills).                                                          */
int pyrohip_adv_evolve(pyrohip_state *s, int n, const pyrohip_adv_params *p,
                       const double *dts, int nsteps);

/* ---- compressible ---------------------------------------------------- */
/* conserved order: density(0) energy(1) x-momentum(2) y-momentum(3)
   (compressible/simulation.py:223-226)                                    */
typedef struct {
    double dx, dy;
    double gamma;          /* eos.gamma                                     */
    int limiter;           /* compressible.limiter 0/1/2                    */
    int use_flattening;    /* compressible.use_flattening                   */
    double z0, z1, delta;  /* compressible.z0/z1/delta                      */
    double cvisc;          /* compressible.cvisc                            */
    double grav;           /* compressible.grav (y direction)               */
    double small_dens;     /* compressible.small_dens                       */
    /* slab decomposition: compute the artificial-viscosity
       coefficient on the upper x (y) boundary face because it is an interior
       interface of the global grid (reference leaves the physical one 0,
       compressible/interface.py:366-367) */
    int avisc_xhi_interior, avisc_yhi_interior;
    /* 0 = bit-faithful arithmetic (no FMA contraction, true divisions);
       1 = contracted / reciprocal arithmetic, parity-tested to 1e-10      */
    int fast_math;
    /* kernel set: -1 = chosen by the library from the grid size (2 from
       2048^2 cells on, 1 below); 0 = staged kernels with global intermediates (debuggable,
       supports pyrohip_comp_stage_dump); 1 = one fused kernel on 2-d LDS
       tiles (best below ~1024^2); 2 = one fused kernel of autonomous
       wavefronts marching along the rows (x windows in registers, y exchange
       by DPP lane rotation; the fastest on large grids)                     */
    int kernel_set;
    /* compressible.riemann: 0 = HLLC (riemann.py:681-860), 1 = CGF (:8-310),
       2 = HLLC_lm (riemann_hllc_lowspeed, :863-1020).
       solid_xl / solid_yl: the lower x / y mesh boundary is a solid wall
       (boundary.bc_is_solid), used by CGF only */
    int riemann, solid_xl, solid_yl;
    /* sponge (compressible/simulation.py:164-184,427-441) */
    int do_sponge;
    double sponge_rho_begin, sponge_rho_full, sponge_timescale;
    /* e_rate of the problem source S[energy] += rho e_rate profile(x, y) of the
       heating / plume / convection problems (compressible/simulation.py:156-159,
       problems/{heating,plume,convection}.py source_terms); only used when the
       state carries a profile (pyrohip_state_set_heating) */
    double heat_rate;
    /* kernel_set 2: rows per strip of the row-marching kernel; 0 = chosen by
       the library from the grid size and the CU count (tuning / test knob)  */
    int march_rows;
    /* 1: pyrohip_comp_step / pyrohip_comp_evolve take the state with UNFILLED ghost cells
       and apply the boundary rules themselves (CellCenterData2d.fill_BC_all, patch.py:
       582-624, folded into the step).  The 2-d tile kernel (kernel_set 1, grids below
       2048^2) then reads every ghost cell from the cell the rule copies from -- two
       launches less per step, which is what a step on a 64^2 ... 512^2 grid consists
       of -- wherever the boundaries are outflow / reflect / periodic / halo and there
       are no source terms; in every other case the step runs the ordinary fill first.
       Either way the ghost cells of the new state hold the filled ghost cells of the old
       one, like the reference's array after evolve(). */
    int fuse_fill;
    /* pyrohip_comp_evolve with the row-marching kernel (kernel_set 2): 0 = the library's choice
       (today 3); 3 = boundary fill, dt policy and step kernel as three launches per step; 1 =
       ONE launch per step on a single domain with outflow / reflect / periodic sides: the
       kernel reads ghost cells through the boundary rules and every wavefront derives the
       step's dt from the CFL minima of the previous launch by the driver's policy
       (simulation_null.py:222-244) -- bit-identical, measured no faster */
    int step_launches;
} pyrohip_comp_params;

/* method_compute_timestep (compressible/simulation.py:267-288 +
   derives.py:19-25,59): cfl * min over the whole array incl. ghost cells */
int pyrohip_comp_dt(pyrohip_state *s, const pyrohip_comp_params *p, double cfl,
                    double *dt_out);
/* Simulation.evolve (compressible/simulation.py:290-450) = interface_states,
   apply_source_terms, apply_transverse_flux, riemann_flux x4 (HLLC),
   apply_artificial_viscosity, conservative update, source corrector
   (unsplit_fluxes.py:134-549, interface.py:5-378, riemann.py:596-860).
   Ghost cells must be filled.  PYROHIP_ERR_STATE mirrors the reference's
   positivity assert (simulation.py:68-71). */
int pyrohip_comp_step(pyrohip_state *s, const pyrohip_comp_params *p,
                      double dt);
/* the launch geometry kernel_set 2 (the row-marching kernel) takes for an nx x ny grid or
   slab on a device of num_cus compute units (<= 0: 256), without launching anything:
   out6 = { column strips, rows per strip, row strips, 1 if a slab's first and last strip go
   first with the halo exchange (pyrohip_state_set_neighbours) beside the interior strips,
   wavefronts per launch, wavefronts resident at once }.  For callers that decompose a grid
   (decomp.py; bench.py's scaling line records it per rank).                            */
int pyrohip_comp_wave_geometry(int nx, int ny, int ng, int num_cus, int march_rows, int *out6);
/* the driver's time-step policy, NullSimulation.compute_timestep
   (simulation_null.py:222-244): dt = cfl * min, scaled by init_tstep_factor on
   the first step (n == 0), growth capped by max_dt_change * dt_old, fix_dt > 0
   overrides, the last step lands on tmax.  t, dt_old, n are in/out. */
typedef struct {
    double tmax, init_tstep_factor, max_dt_change, fix_dt;
    double t, dt_old;
    long long n;
} pyrohip_dt_policy;
/* up to max_steps iterations of Pyro.single_step (pyro_sim.py:241-281: ghost fill
   [+ halo exchange of a slab], compute_timestep, evolve) WITHOUT a host round trip
   per step: the policy above runs in a one-thread kernel on the CFL minimum the
   previous step left in device memory, the step kernels take dt from device
   memory.  Steps past tmax do nothing.  One synchronisation at the end returns
   the number of steps taken, the policy state and (dts_out, may be NULL, max_steps
   doubles) the dt of each step (-1 for the ones not taken).  Cartesian grids,
   outflow / reflect / periodic / halo boundaries, kernel sets -1 / 1 / 2, no
   sponge.  PYROHIP_ERR_STATE: a step found an invalid state; the state left
   behind is the one before that step, like after a failed pyrohip_comp_step. */
int pyrohip_comp_evolve(pyrohip_state *s, const pyrohip_comp_params *p, double cfl,
                        pyrohip_dt_policy *policy, int max_steps, int *steps_done,
                        double *dts_out);
/* debug: copy an intermediate of the LAST staged step to the host.
   stage ids: 0 q(4) 1 xi(1) 2 XM(4) 3 XP(4) 4 YM(4) 5 YP(4) 6 FxT(4)
   7 FyT(4) 8 Fx(4) 9 Fy(4).  out: (qx,qy,ncomp) reference layout.
   XM/XP (YM/YP) are the lower/upper face states of each CELL:
   XM[i,j] = U_xr[i,j], XP[i,j] = U_xl[i+1,j] before the transverse
   correction (unsplit_fluxes.py:207-242). */
int pyrohip_comp_stage_dump(pyrohip_state *s, int stage_id, double *out);

/* ---- multigrid: MG.CellCenterMG2d (multigrid/MG.py:85-778) ------------- */
/* bc: xl,xr,yl,yr with PYROHIP_BC_REFLECT_ODD = dirichlet,
   PYROHIP_BC_OUTFLOW = neumann, PYROHIP_BC_PERIODIC; PYROHIP_BC_CONST: ghost
   cells = 0 (what incompressible_viscous/BC.py:39-42 does to the variable "v"
   of the velocity solves)                                                   */
int pyrohip_mg_create(pyrohip_ctx *ctx, int nx, double xmin, double xmax,
                      double ymin, double ymax, const int *bc, double alpha,
                      double beta, int nsmooth, int nsmooth_bottom,
                      pyrohip_mg **out);
int pyrohip_mg_destroy(pyrohip_mg *m);
/* new alpha, beta of (alpha - beta L) phi = f for an existing solver: the
   reference builds a new MG object for every solve with beta = dt nu / 2
   (incompressible_viscous/simulation.py:103-111); the levels are reused here */
int pyrohip_mg_set_helmholtz(pyrohip_mg *m, double alpha, double beta);
int pyrohip_mg_nlevels(pyrohip_mg *m, int *nlevels);
/* smoother implementation: 1 (default) = LDS tile kernel running up to 5
   red-black iterations per launch; 0 = one launch per colour.  Results are
   bit-identical. */
int pyrohip_mg_set_smoother(pyrohip_mg *m, int kind);
/* Tuning of the multigrid kernels.  Every setting gives bit-identical results; the
   defaults (what pyrohip_mg_create leaves, read them with pyrohip_mg_get_tuning) are the
   measured winners on MI355X.  Tests use this to run the large-level
   kernels on small levels; it replaces the PYRO_MG_* environment knobs of round 2.
     kmax             red-black iterations per launch of the tile / band smoother (1..5)
     kmax_small       ... on levels up to nsmall^2 (0..10; 10: a whole leg in one launch)
     nsmall           see kmax_small (512)
     march_min        row-marching smoother (mg_march.hip) on levels >= march_min^2
                      (2048; 0: never)
     march_waves      wavefronts a marching launch is cut into (0: what the device holds)
     march_side       the first / last column strip gets 1 / march_side of the rows (1.5)
     march_minrows    rows a wavefront stores, at least (32)
     fuse_res_restrict  residual + restriction in one pass on the way down (1)
     lazy_residual    inside solve() residual arrays are computed on demand (1)
     allow_pow2       scaled right-hand side where the coefficients are powers of two (1)
     small_tiles      workgroups aimed at on the small levels (-1: built-in default)
     band_maxn        band smoother up to this level size (2048)
     band_genedge     band smoother: the general edge instance everywhere (0)
     coarse_band64    coarse V-cycle kernel: the 64^2 level's sweeps in registers (1)
     march_tail       inside solve(): the down leg's residual + restriction and the cycle's
                      two sums ride on the marching smoother's launches (1)
     coarse_wave      coarse V-cycle kernel: the levels up to 32^2 on one wavefront, in
                      registers (1)
     speculate        solve(): launch the next V-cycle while the norms travel to the host:
                      0 never, 1 when a further cycle is likely (default), 2 always
     trace, spec_debug  developer aids (phase clocks of the band / coarse kernels; solve()
                      prints cycles / launched ahead / undone)                              */
typedef struct {
    int kmax, kmax_small, nsmall;
    int march_min, march_waves;
    double march_side;
    int march_minrows;
    int fuse_res_restrict, lazy_residual, allow_pow2;
    int small_tiles, band_maxn, band_genedge, coarse_band64;
    int speculate, trace, spec_debug;
    int march_tail, coarse_wave;
} pyrohip_mg_tuning;
int pyrohip_mg_get_tuning(pyrohip_mg *m, pyrohip_mg_tuning *t);
int pyrohip_mg_set_tuning(pyrohip_mg *m, const pyrohip_mg_tuning *t);
/* marching launches so far that carried the down leg's residual + restriction / a solve
   cycle's two sums (march_tail above; the tests make sure the path they test is the one taken) */
int pyrohip_mg_tail_counts(pyrohip_mg *m, int *restrictions, int *diagnostics);
/* var: 0 = v, 1 = f, 2 = r; arrays are (n+2, n+2) with ng = 1 */
int pyrohip_mg_set(pyrohip_mg *m, int level, int var, const double *host);
int pyrohip_mg_get(pyrohip_mg *m, int level, int var, double *host);
/* inhomogeneous boundary values on the finest level (boundary.py:196-211);
   side 0..3 = xl,xr,yl,yr; vals has n+2 entries; NULL clears */
int pyrohip_mg_set_bcval(pyrohip_mg *m, int side, const double *vals);
int pyrohip_mg_zero(pyrohip_mg *m, int level, int var);        /* patch.py:562 */
int pyrohip_mg_fill_bc(pyrohip_mg *m, int level, int var);
int pyrohip_mg_smooth(pyrohip_mg *m, int level, int nsmooth);  /* MG.py:544-621 */
/* Row windows: the building blocks of a V-cycle whose levels are split into x
   slabs across GPUs (pyro2_amd/multigrid/slab.py; constant coefficients, levels
   above 64^2).  Rows are 1-based interior rows of the (n+2, n+2) level arrays.
   rows_kmax: how many red-black iterations ONE launch does on that level: 10 (a whole
   V-cycle leg) where the row-marching kernel (levels >= 2048^2) or the deep-apron band
   kernel (levels <= 512^2) runs, 5 in between; 0: no row windows (variable coefficients).
   smooth_rows: ONE launch, nsweeps <= rows_kmax red-black
   iterations on rows [row0, row1]; the 2*nsweeps rows beyond them must hold the
   neighbours' current values, v and f (physical boundaries are refreshed by the kernel);
   prolong != 0: add the prolongation of the coarser level's solution while
   staging (needs nsweeps + 1 coarse halo rows).  Identical, bit for bit, to what
   the whole-level launch computes on those rows.
   residual_restrict_rows: residual of the fine rows 2*crow0-1 .. 2*crow1 and its
   restriction into the coarse right-hand side rows [crow0, crow1].
   get_rows / set_rows: rows [i0, i0+ni) of a level array incl. ghost columns
   (host staging of the halos; var 0 = v, 1 = f, 2 = r).
   mark_zero: the level's solution is zero from here on (MG.py:658-659). */
int pyrohip_mg_rows_kmax(pyrohip_mg *m, int level, int *k);
int pyrohip_mg_smooth_rows(pyrohip_mg *m, int level, int nsweeps, int row0, int row1,
                           int prolong);
/* the diagnostics of one cycle of solve() (MG.py:670-686) over rows [row0, row1] of the
   finest level: sums[0] = sum ((v - old) / (v + 1e-16))^2, sums[1] = sum r^2 (one current
   halo row of v on either side); old <- v on those rows.  save_old: old <- v everywhere
   (MG.py:647).  The slabs' sums are added over the ranks (pyrohip_allreduce_sum). */
int pyrohip_mg_diag_rows(pyrohip_mg *m, int row0, int row1, double *sums);
int pyrohip_mg_save_old(pyrohip_mg *m);
int pyrohip_mg_residual_restrict_rows(pyrohip_mg *m, int fine, int crow0, int crow1);
int pyrohip_mg_get_rows(pyrohip_mg *m, int level, int var, int i0, int ni, double *host);
int pyrohip_mg_set_rows(pyrohip_mg *m, int level, int var, int i0, int ni, const double *host);
int pyrohip_mg_mark_zero(pyrohip_mg *m, int level);
/* the same row moves over RCCL, device to device (pyrohip_comm_init first):
   exchange_rows: h halo rows on either side of the slab [row0, row1] with the x
   neighbours (-1 = none), one grouped send / recv pair per neighbour;
   send_rows / recv_rows: a block of rows to / from one peer (gather of the
   right-hand side to, scatter of the solution from the rank that owns the
   collapsed levels); several of them form one step between
   pyrohip_comm_group(1) and pyrohip_comm_group(0). */
int pyrohip_mg_exchange_rows(pyrohip_mg *m, int level, int var, int row0, int row1, int h,
                             int rank_lo, int rank_hi);
int pyrohip_mg_send_rows(pyrohip_mg *m, int level, int var, int i0, int ni, int peer);
int pyrohip_mg_recv_rows(pyrohip_mg *m, int level, int var, int i0, int ni, int peer);
int pyrohip_comm_group(int begin);
int pyrohip_mg_residual(pyrohip_mg *m, int level);             /* MG.py:529-542 */
int pyrohip_mg_restrict(pyrohip_mg *m, int fine_level);        /* patch.py:640-676 */
int pyrohip_mg_prolong_add(pyrohip_mg *m, int fine_level);     /* patch.py:678-736 */
int pyrohip_mg_norm(pyrohip_mg *m, int level, int var, double *out); /* array_indexer.py:98-111 */
int pyrohip_mg_vcycle(pyrohip_mg *m, int level);               /* MG.py:699-778 */
/* init_RHS bookkeeping: source_norm = ||f|| on the finest level (MG.py:521) */
int pyrohip_mg_init_rhs_norm(pyrohip_mg *m, double *source_norm);
/* variable-coefficient mode, VarCoeffCCMG2d (multigrid/variable_coeff_MG.py:
   23-213): solve div(eta grad phi) = f.  coeffs: (n+2, n+2) cell-centred eta on
   the finest level (interior is used), coeffs_bc: its 4 BC codes.  Builds the
   edge coefficients on every level (edge_coeffs.py:1-54); smoothing and the
   residual then use them.  pyrohip_mg_get accepts var 3 = eta, 4 = eta_x,
   5 = eta_y afterwards. */
int pyrohip_mg_set_coeffs(pyrohip_mg *m, const double *coeffs, const int *coeffs_bc);
/* general mode, GeneralMG2d (multigrid/general_MG.py:22-242): solve
   alpha phi + div(beta grad phi) + gamma . grad phi = f.  Four (n+2, n+2)
   cell-centred arrays on the finest level and their BC codes (alpha, beta,
   gamma_x, gamma_y; 4 each).  pyrohip_mg_get then also accepts var 6 = alpha,
   7 = gamma_x, 8 = gamma_y (3..5 = beta, beta_x, beta_y).                    */
int pyrohip_mg_set_general_coeffs(pyrohip_mg *m, const double *alpha,
                                  const double *beta, const double *gamma_x,
                                  const double *gamma_y, const int *coeffs_bc);
/* callers that keep their field on the device (pyro/diffusion/simulation.py:
   92-118): f <- phi + coef * Laplacian(phi) on the finest level from variable
   n of a (nx, nx, ng = 1) state, and the solution back into that variable */
int pyrohip_mg_set_rhs_cn(pyrohip_mg *m, pyrohip_state *s, int n, double coef,
                          double *source_norm);
int pyrohip_mg_copy_solution(pyrohip_mg *m, pyrohip_state *s, int n);
/* MG.py:623-697 */
int pyrohip_mg_solve(pyrohip_mg *m, double rtol, int max_cycles,
                     int *num_cycles, double *residual_error,
                     double *relative_error);

/* ---- compressible_rk (method of lines) -------------
   Simulation.substep (pyro/compressible_rk/simulation.py:10-44) with
   fluxes.fluxes (compressible_rk/fluxes.py:28-180): k = -div F + S of the
   stage state y (ghost cells filled; the density floor is applied to y in
   place) into planes 4*slot .. 4*slot+3 of the state k (same nx, ny, ng).   */
int pyrohip_comp_rk_rhs(pyrohip_state *y, const pyrohip_comp_params *p,
                        pyrohip_state *k, int slot);
/* its CFL step (compressible_rk/simulation.py:46-56)                        */
/* Arbitrary problem source terms (the `source_terms(myg, U, ivars, rp)` callback of
   a problem module, compressible/simulation.py:157-159) are evaluated by the HOST;
   the device applies them where the reference does.  Per step:
     1. src_old := S_h(U^n) uploaded into a 4-variable state with the boundary types
        of the reference's aux data (simulation.py:248-253; hse / ambient as outflow,
        BC.py:56-63,146-151), ghost-filled (pyrohip_state_fill_bc);
        pyrohip_state_set_source(s, 0, src_old);
     2. pyrohip_comp_step: interface states with dt/2 (S_grav + S_h) (unsplit_fluxes.py:
        295-328), fluxes, conservative update and the predictor U* = U + dt S(U^n)
        (simulation.py:406-412) -- staged kernels, any kernel_set;
     3. download U*, src_new := S_h(U*); pyrohip_state_set_source(s, 1, src_new);
     4. pyrohip_comp_source_correct: U = U* + dt/2 (S(U*) - S(U^n)) with the
        time-centred gravity of simulation.py:126-155, then the sponge.
   The states passed in are borrowed (keep them alive); NULL removes the source.
   Cartesian grids; excludes the heating profile and the ramp boundary. */
int pyrohip_state_set_source(pyrohip_state *s, int which, pyrohip_state *src);
int pyrohip_comp_source_correct(pyrohip_state *s, const pyrohip_comp_params *p, double dt);

int pyrohip_comp_rk_dt(pyrohip_state *s, const pyrohip_comp_params *p,
                       double cfl, double *dt_out);
/* The WHOLE step of compressible_rk.Simulation.evolve (pyro/compressible_rk/simulation.py:
   58-104) with the RKIntegrator of pyro/mesh/integration.py:76-129 in nstages launches:
   stage 0 = pyrohip_comp_rk_rhs of the ghost-filled state; every later stage builds its start
   y_0 + dt sum_j a[s][j] k_j (interior; ghost cells = the images the boundary rules give) as
   the rows enter the kernel, the last one stores y_0 + dt sum_s b[s] k_s as the new state and
   leaves the CFL minimum pyrohip_comp_rk_dt then answers from.  a: nstages x nstages row-major
   (strictly lower triangular), b: nstages, 2 <= nstages <= 4; k: a state with >= 4 nstages
   planes (scratch).  Needs pyrohip_comp_rk_can_fuse (single Cartesian domain, outflow /
   reflect / periodic sides, no sponge / heating / host source, kernel_set 2 or >= 2048^2
   cells); everything else goes stage by stage (pyrohip_comp_rk_rhs + pyrohip_state_lincomb).
   PYROHIP_ERR_STATE: an invalid stage state; y is left as it was (floored).                */
int pyrohip_comp_rk_can_fuse(pyrohip_state *y, const pyrohip_comp_params *p,
                             pyrohip_state *k, int nstages, int *flag);
int pyrohip_comp_rk_step(pyrohip_state *y, const pyrohip_comp_params *p,
                         pyrohip_state *k, double dt, int nstages,
                         const double *a, const double *b);
/* ... and up to max_steps of them with the driver's dt policy on the device, as
   pyrohip_comp_evolve (no host round trip per step; one synchronisation at the end)      */
int pyrohip_comp_rk_evolve(pyrohip_state *y, const pyrohip_comp_params *p,
                           pyrohip_state *k, int nstages, const double *a,
                           const double *b, double cfl, pyrohip_dt_policy *policy,
                           int max_steps, int *steps_done, double *dts_out);
/* RKIntegrator.get_stage_start / compute_final_update (pyro/mesh/
   integration.py:84-113): dst <- src everywhere (clone), then on the interior
   dst += coef[0] k_0; dst += coef[1] k_1; ... in this order.  dst may be src. */
int pyrohip_state_lincomb(pyrohip_state *dst, const pyrohip_state *src,
                          const pyrohip_state *k, const double *coef, int ncoef);

/* ---- shallow water (pyro/swe) -------------------------
   state: 4 variables height, x-momentum, y-momentum, fuel (ng >= 4).
   riemann: 0 Roe, 1 HLLC (swe/interface.py:216-554).
   swe_dt: Simulation.method_compute_timestep (swe/simulation.py:143-153);
   swe_step: Simulation.evolve (:155-193) with unsplit_fluxes.unsplit_fluxes
   (swe/unsplit_fluxes.py:132-380) and interface.states (:5-213)            */
int pyrohip_swe_dt(pyrohip_state *s, double dx, double dy, double grav,
                   double cfl, double *dt_out);
int pyrohip_swe_step(pyrohip_state *s, double dx, double dy, double grav,
                     int limiter, int riemann, double dt);
/* the same with the kernel set named: 0 the staged kernels (every stage dumpable through
   pyrohip_swe_stage_dump), 1 the whole step in one launch (row-marching wavefronts, state
   read once and written once; bit-identical to the staged set), -1 the library's choice
   (= 1; what pyrohip_swe_step runs) */
int pyrohip_swe_step_ks(pyrohip_state *s, double dx, double dy, double grav, int limiter,
                        int riemann, double dt, int kernel_set);
/* ... and the arithmetic (one-launch kernel): fast_math 0 the reference's operation order
   (bit-identical to the staged set), 1 contracted, reciprocal-based quotients, the
   characteristic sums without their structural zeros: <= 1e-10 element-wise of the former     */
int pyrohip_swe_step_ex(pyrohip_state *s, double dx, double dy, double grav, int limiter,
                        int riemann, double dt, int kernel_set, int fast_math);
/* up to max_steps iterations of the swe driver loop (pyro_sim.py:241-281 with swe/simulation.py:
   143-193) without a host round trip per step, as pyrohip_comp_evolve: ghost fill, the driver's
   dt policy in a kernel on the CFL minimum the previous step's kernel left, the one-launch step.
   Single Cartesian domain, standard boundary types.                                          */
int pyrohip_swe_evolve(pyrohip_state *s, double dx, double dy, double grav, int limiter,
                       int riemann, int fast_math, double cfl, pyrohip_dt_policy *policy,
                       int max_steps, int *steps_done, double *dts_out);
/* test hook: 0-3 U_xl U_xr U_yl U_yr before the transverse terms, 4 FxT 5 FyT
   (transverse fluxes), 6 Fx 7 Fy -> host (qx, qy, 4)                        */
int pyrohip_swe_stage_dump(pyrohip_state *s, int stage, double *out);

/* ---- burgers / incompressible (the solvers on top of the multigrid solver;
        SURVEY.md 8 rows f1, f4) ------------------------------------------- */
/* burgers Simulation.evolve (pyro/burgers/simulation.py:53-117 with
   burgers_interface.py:4-312): one unsplit CTU step of (u, v) = variables
   iu, iv of the state (ng >= 4, ghost cells filled)                        */
int pyrohip_bg_step(pyrohip_state *s, int iu, int iv, double dx, double dy,
                    double dt, int limiter);
/* incompressible Simulation.evolve (pyro/incompressible/simulation.py:200-
   330), the four device pieces around the two MG solves.  mg: a
   pyrohip_mg with the state's nx (= ny) and the BCs of phi.
   1. mac_rhs: edge states (incomp_interface.mac_vels :4-63), MAC velocities,
      mg.f = div(U_MAC), mg.v = 0 (init_zeros), source norm (init_RHS).
      nu > 0: the viscous source nu L(U) of incompressible_viscous
      (simulation.py:24-41, incomp_interface.py:186-254) enters the edge
      states                                                                */
int pyrohip_inc_mac_rhs(pyrohip_state *s, pyrohip_mg *mg, int iu, int iv,
                        int igpx, int igpy, double dx, double dy, double dt,
                        int limiter, double nu, double *source_norm);
/* 2. after mg solve: phi-MAC <- solution (buf 1), MAC correction, states()
      (:66-136), advective terms and provisional velocity update (:286-304);
      proj_type 0: advective terms only (the viscous update follows)        */
int pyrohip_inc_advect(pyrohip_state *s, pyrohip_mg *mg, int iu, int iv,
                       int iphimac, int igpx, int igpy, double dx, double dy,
                       double dt, int proj_type);
/* 3. mg.f = cell-centred div(U) [/ dt], mg.v = phi on buf 1 (iphi >= 0) or 0
      (:306-325; :99-103 for the initial projection of preevolve)           */
int pyrohip_inc_proj_rhs(pyrohip_state *s, pyrohip_mg *mg, int iu, int iv,
                         int iphi, double dx, double dy, double dt,
                         int divide_by_dt, double *source_norm);
/* 4. after mg solve: phi <- solution (buf 1, 0 elsewhere), (u,v) -= fac *
      grad(solution); gp_mode 0: grad p untouched, 1: +=, 2: = (:327-339)   */
int pyrohip_inc_proj_update(pyrohip_state *s, pyrohip_mg *mg, int iu, int iv,
                            int iphi, int igpx, int igpy, double dx, double dy,
                            double fac, int gp_mode);
/* incompressible_viscous do_other_update_velocity (pyro/incompressible_viscous/
   simulation.py:43-176), one velocity component w = variable iw (comp 0: u,
   1: v) per call.  visc_rhs: mg.f = w + dt nu / 2 L(w) - dt (advect [+ grad p,
   proj_type 1]), mg.v = w on buf 1 (the guess); mg must have the BCs of w and
   alpha = 1, beta = dt nu / 2 (pyrohip_mg_set_helmholtz).  visc_store after the
   solve: interior of w <- solution                                          */
int pyrohip_inc_visc_rhs(pyrohip_state *s, pyrohip_mg *mg, int iw, int comp,
                         int igp, double dx, double dy, double dt, double nu,
                         int proj_type, double *source_norm);
int pyrohip_inc_visc_store(pyrohip_state *s, pyrohip_mg *mg, int iw);
/* burgers_viscous Simulation.evolve (pyro/burgers_viscous/simulation.py:9-89):
   bgv_predict: edge states with the diffusion correction eps dt / 2 L(U) applied
   before the transverse terms (burgers_viscous/interface.py:94-171), MAC
   velocities.  bgv_rhs, once per component (comp 0: u, 1: v; iw its variable):
   mg.f = w + dt eps / 2 L(w) - dt A with A from the unsplit fluxes, mg.v = 0
   (interface.diffuse :27-91; mg: BCs of w, alpha = 1, beta = dt eps / 2); after
   the solve pyrohip_inc_visc_store writes the solution back                  */
int pyrohip_bgv_predict(pyrohip_state *s, int iu, int iv, double dx, double dy,
                        double dt, int limiter, double eps);
int pyrohip_bgv_rhs(pyrohip_state *s, pyrohip_mg *mg, int iw, int comp, double dx,
                    double dy, double dt, double eps, double *source_norm);
/* test hook: which 0-7 edge states u_xl u_xr u_yl u_yr v_xl v_xr v_yl v_yr,
   8 u_MAC, 9 v_MAC, 10 advect_x, 11 advect_y -> host (qx, qy)              */
int pyrohip_inc_stage_dump(pyrohip_state *s, int which, double *host);

/* ---- multi-GPU: x-slab decomposition, one process per GPU, RCCL -------- */
#define PYROHIP_UNIQUE_ID_BYTES 128
int pyrohip_comm_unique_id(char *out_id /* PYROHIP_UNIQUE_ID_BYTES */);
int pyrohip_comm_init(pyrohip_ctx *ctx, int nranks, int rank,
                      const char *unique_id);
int pyrohip_comm_destroy(pyrohip_ctx *ctx);
/* number of ranks RCCL itself reports for the context's communicator
   (ncclCommCount; 0 without a communicator) -- bench.py prints it */
int pyrohip_comm_size(pyrohip_ctx *ctx, int *nranks);
/* on: pyrohip_comp_step all-reduces (min) the CFL minimum of the new state
   over the communicator on the device, inside the step -- every rank must then
   call comp_step in lock step.  pyrohip_comp_dt_is_global tells whether the next
   pyrohip_comp_dt comes from that reduced value (no pyrohip_allreduce_min
   needed) or from a local reduction (first step, after an upload).          */
int pyrohip_comm_set_global_dt(pyrohip_ctx *ctx, int on);
/* flag = 1: pyrohip_comp_dt will answer from the CFL minimum the last step kernel left
   behind, without reading the state (so its ghost cells need not be filled: fuse_fill) */
int pyrohip_comp_dt_is_cached(pyrohip_state *s, int *flag);
int pyrohip_comp_dt_is_global(pyrohip_state *s, int *flag);
/* flag = 1: pyrohip_comp_rk_dt will answer from the minimum the last stage of pyrohip_comp_rk_step
   left (compressible_rk's CFL quantity), without reading the state or its ghost cells */
int pyrohip_comp_rk_dt_is_cached(pyrohip_state *s, int *flag);
/* exchange ng ghost rows of every variable with the x neighbours
   (rank_lo / rank_hi, -1 = none).  Replaces the single-domain x ghost fill
   for PYROHIP_BC_HALO sides; y ghost fill must follow (fill order of
   array_indexer.py:150-274). */
int pyrohip_halo_exchange(pyrohip_state *s, int rank_lo, int rank_hi);
/* tell a slab's state who its x neighbours are (-1 = none).  pyrohip_comp_step
   (kernel_set 2) then updates the first and the last strip of rows first and
   posts the exchange of the NEW boundary rows on a second stream / communicator,
   so that it overlaps the update of the interior strips; the next
   pyrohip_halo_exchange(s, same neighbours) only waits for it.  Any write to
   the state in between (upload, fill of another kind) makes that call exchange
   again.  Without this call every exchange is synchronous. */
int pyrohip_state_set_neighbours(pyrohip_state *s, int rank_lo, int rank_hi);
/* 1 if the last step posted the halo exchange of the state it produced and
   nothing has touched the state since (diagnostics / tests) */
int pyrohip_state_halo_pending(pyrohip_state *s, int *flag);
/* whole rows [i0, i0 + ni) of every variable of a state to / from one peer rank (ghost columns
   included: a row is `pitch` contiguous doubles per variable) -- how the slabs of a decomposed
   run are gathered for output (CellCenterData2d.write_data / Simulation.write of a decomposed
   run, pyro/mesh/patch.py:750-788 writes ONE array per variable).  The calls of one collective
   step are bracketed by pyrohip_comm_group(1) ... pyrohip_comm_group(0); sender and receiver
   must name the same number of rows of states with the same ny and ng. */
int pyrohip_state_send_rows(pyrohip_state *s, int i0, int ni, int peer);
int pyrohip_state_recv_rows(pyrohip_state *s, int i0, int ni, int peer);
int pyrohip_allreduce_min(pyrohip_ctx *ctx, double *value);
int pyrohip_allreduce_max(pyrohip_ctx *ctx, double *value);
/* sum of n doubles over the ranks, in place (n <= 16) */
int pyrohip_allreduce_sum(pyrohip_ctx *ctx, double *values, int n);

#ifdef __cplusplus
}
#endif
#endif /* PYROHIP_H */
