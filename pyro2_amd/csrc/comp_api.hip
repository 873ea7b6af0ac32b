// extern "C" entry points of the compressible solver: argument checking and
// dispatch between the bit-faithful (exact) and contracted (fastm) builds of
// compressible.hip / comp_fused.hip.
#include "common.h"
#include "stencil.h"

namespace pyro {
namespace exact {
int comp_dt(pyrohip_state *, const pyrohip_comp_params *, double, double *);
int comp_step_staged(pyrohip_state *, const pyrohip_comp_params *, double);
int comp_step_fused(pyrohip_state *, const pyrohip_comp_params *, double);
int comp_step_wave(pyrohip_state *, const pyrohip_comp_params *, double);
int comp_step_fused_ex(pyrohip_state *, const pyrohip_comp_params *, double, const StepScalars *,
                       const double **);
int comp_step_wave_ex(pyrohip_state *, const pyrohip_comp_params *, double, const StepScalars *,
                      const double **);
int comp_cfl_min_device(pyrohip_state *, const pyrohip_comp_params *, const double **);
int comp_step_sph(pyrohip_state *, const pyrohip_comp_params *, double);
int comp_step_wave_sph(pyrohip_state *, const pyrohip_comp_params *, double);
int comp_step_wave_sph_ex(pyrohip_state *, const pyrohip_comp_params *, double, const StepScalars *,
                          const double **);
int comp_step_fused_sph(pyrohip_state *, const pyrohip_comp_params *, double);
int comp_step_fused_sph_ex(pyrohip_state *, const pyrohip_comp_params *, double, const StepScalars *,
                           const double **);
int comp_cfl_min_device_sph(pyrohip_state *, const pyrohip_comp_params *, const double **);
int comp_dt_sph(pyrohip_state *, const pyrohip_comp_params *, double, double *);
int comp_stage_dump(pyrohip_state *, int, double *);
int comp_sponge(pyrohip_state *, const pyrohip_comp_params *, double);
int comp_source_correct(pyrohip_state *, const pyrohip_comp_params *, double);
int comp_rk_dt(pyrohip_state *, const pyrohip_comp_params *, double, double *);
int comp_rk_rhs(pyrohip_state *, const pyrohip_comp_params *, pyrohip_state *, int);
int comp_rk_rhs_wave(pyrohip_state *, const pyrohip_comp_params *, pyrohip_state *, int);
int comp_rk_step_wave(pyrohip_state *, const pyrohip_comp_params *, pyrohip_state *, int, const double *,
                      const double *, double, const StepScalars *, const double **);
int comp_rk_cfl_min_device(pyrohip_state *, const pyrohip_comp_params *, const double **);
int comp_wave_geometry(int, int, int, int, int, int *);
}
namespace fastm {
int comp_rk_rhs(pyrohip_state *, const pyrohip_comp_params *, pyrohip_state *, int);
int comp_rk_rhs_wave(pyrohip_state *, const pyrohip_comp_params *, pyrohip_state *, int);
int comp_rk_step_wave(pyrohip_state *, const pyrohip_comp_params *, pyrohip_state *, int, const double *,
                      const double *, double, const StepScalars *, const double **);
int comp_dt(pyrohip_state *, const pyrohip_comp_params *, double, double *);
int comp_step_staged(pyrohip_state *, const pyrohip_comp_params *, double);
int comp_step_fused(pyrohip_state *, const pyrohip_comp_params *, double);
int comp_step_wave(pyrohip_state *, const pyrohip_comp_params *, double);
int comp_step_fused_ex(pyrohip_state *, const pyrohip_comp_params *, double, const StepScalars *,
                       const double **);
int comp_step_wave_ex(pyrohip_state *, const pyrohip_comp_params *, double, const StepScalars *,
                      const double **);
int comp_cfl_min_device(pyrohip_state *, const pyrohip_comp_params *, const double **);
int comp_step_sph(pyrohip_state *, const pyrohip_comp_params *, double);
int comp_step_wave_sph(pyrohip_state *, const pyrohip_comp_params *, double);
int comp_step_wave_sph_ex(pyrohip_state *, const pyrohip_comp_params *, double, const StepScalars *,
                          const double **);
int comp_step_fused_sph(pyrohip_state *, const pyrohip_comp_params *, double);
int comp_step_fused_sph_ex(pyrohip_state *, const pyrohip_comp_params *, double, const StepScalars *,
                           const double **);
int comp_cfl_min_device_sph(pyrohip_state *, const pyrohip_comp_params *, const double **);
int comp_dt_sph(pyrohip_state *, const pyrohip_comp_params *, double, double *);
}
}  // namespace pyro

using namespace pyro;

// kernel_set -1: the row-marching wavefront kernel needs >= ~2 wavefronts per SIMD
// of 56 columns x >= 32 rows to fill the chip; measured crossover with the tile
// kernel at 2048^2 (profiles/r02_kernel_sets_by_size.txt)
static bool wave_kernel_pays(const Geom &g)
{
    return (double)g.nx * (double)g.ny >= 2048.0 * 2048.0;
}
// ... the Cartesian CTU step of the contracted build from 1024^2 cells on (round 6: with strips short
// enough for ONE round of resident wavefronts -- wave_rows down to 8 rows -- the row-marching kernel
// passes the tile kernel there: 1024^2 78.3 -> 71.0 us per step, 1152^2 90.9 -> 78.4, 1536^2 148.6 -> 136.6,
// 1792^2 190.4 -> 155.7; 896^2 a tie, below it and in the bit-faithful build the tile kernel stays ahead
// up to 2048^2: 1024^2 104.8 vs 114.6 us, 1536^2 203.5 vs 226.9)
static bool wave_kernel_pays_ctu(const Geom &g, const pyrohip_comp_params *p)
{
    const double cells = (double)g.nx * (double)g.ny;
    return cells >= (p->fast_math ? 1024.0 * 1024.0 : 2048.0 * 2048.0) && g.nx >= 512 && g.ny >= 512;
}

// May the tile kernel read the ghost cells through the boundary rules instead of from
// memory (pyrohip_comp_params.fuse_fill)?  Index maps exist for outflow / reflect /
// periodic sides (halo rows are data); all four variables must follow the same kind of
// rule on a side (they do for bc / bc_xodd / bc_yodd, simulation_null.py:72-112); the
// source terms read ghost cells of their own.
static bool comp_can_fuse_fill(const pyrohip_state *s, const pyrohip_comp_params *p, bool wave)
{
    if (wave || p->kernel_set == 0 || s->sph || s->user_bc || s->ramp_bc || s->ext_old ||
        p->grav != 0.0 || s->heat != nullptr)
        return false;
    for (int sd = 0; sd < 4; sd++) {
        int kind0 = -1;
        for (int n = 0; n < 4; n++) {
            const int b = s->bc[n * 4 + sd];
            int kind;
            if (b == PYROHIP_BC_OUTFLOW) kind = 0;
            else if (b == PYROHIP_BC_REFLECT_EVEN || b == PYROHIP_BC_REFLECT_ODD) kind = 1;
            else if (b == PYROHIP_BC_PERIODIC) kind = 2;
            else if (b == PYROHIP_BC_HALO) kind = 3;
            else return false;
            if (n == 0) kind0 = kind;
            else if (kind != kind0) return false;
        }
    }
    return true;
}

// SphericalPolar grid: one launch (k_ctu_fused_sph) where the boundaries are index maps -- the
// same kind (outflow / reflect / periodic) for the four variables on every side; the staged set
// (kernel_set 0: stage dumps) everywhere else
static bool comp_can_fuse_sph(const pyrohip_state *s, const pyrohip_comp_params *p)
{
    // (single domain; the tile kernel's 4-cell apron needs ng >= 4 and as many interior cells)
    if (!s->sph || p->kernel_set == 0 || s->nb_set || s->g.ng < 4 || s->g.nx < 4 || s->g.ny < 4 ||
        s->user_bc || s->ramp_bc || s->heat || s->ext_old || p->riemann != 1)
        return false;
    for (int sd = 0; sd < 4; sd++) {
        int kind0 = -1;
        for (int n = 0; n < 4; n++) {
            const int b = s->bc[n * 4 + sd];
            const int kind = (b == PYROHIP_BC_OUTFLOW) ? 0
                             : (b == PYROHIP_BC_REFLECT_EVEN || b == PYROHIP_BC_REFLECT_ODD) ? 1
                             : (b == PYROHIP_BC_PERIODIC) ? 2 : -1;
            if (kind < 0 || (n > 0 && kind != kind0)) return false;
            kind0 = kind;
        }
    }
    return true;
}

static int check_comp(pyrohip_state *s, const pyrohip_comp_params *p)
{
    PYRO_REQUIRE(s && p, "NULL argument");
    PYRO_REQUIRE(s->nvar == 4, "compressible state must have 4 variables "
                               "(density, energy, x-momentum, y-momentum)");
    PYRO_REQUIRE(s->g.ng >= 4, "compressible needs ng >= 4 (compressible/simulation.py:194)");
    PYRO_REQUIRE(p->limiter >= 0 && p->limiter <= 2, "limiter must be 0, 1 or 2");
    PYRO_REQUIRE(p->dx > 0 && p->dy > 0 && p->gamma > 1.0, "bad dx/dy/gamma");
    return 0;
}

// Boundary fill of all four variables AND the ghost frame of the other state buffer in one
// launch (device-side stepping with the row-marching kernel: pyrohip_fill_bc is two launches,
// the copy of the ghost frame into the new buffer a third -- 19 us of kernels and two gaps per
// step, 2.5 % of a 4096^2 step, 8 % at 2048^2).  A ghost cell's value goes through the x rule
// and then the y rule (array_indexer.py:163-274 fills x over all columns first, so a corner
// takes its value from an x ghost cell): for outflow / reflect / periodic sides both are index
// maps with a sign, and their composition is what the two passes leave.  One thread per
// cell of the frame.
// (b: piece of 256 threads, t: thread in the piece -- a workgroup of k_fill_frame2, or a quarter
// of one of k_fill_frame2_policy)
__device__ __forceinline__ void fill_frame2_piece(const double *src, double *cur, double *alt,
                                                  const Geom &g, const int *__restrict__ bc, int b, int t)
{   // src: the buffer whose interior the images are taken from (cur itself, or -- at the end of a
    // run of one-launch steps -- the buffer that holds the previous state); alt may be nullptr
    // 1-d grid: first the 2 ng full ghost rows in pieces of 256 columns, then the ghost
    // columns of the interior rows, 256 / (2 ng) rows per piece
    const int ng = g.ng;
    const int nxb = (g.qy + 255) / 256, nrowblk = 2 * ng * nxb;
    int i, j;
    if (b < nrowblk) {                              // a piece of a full ghost row
        const int rr = b / nxb;
        i = (rr < ng) ? rr : g.ihi + 1 + (rr - ng);
        j = (b - rr * nxb) * 256 + t;
        if (j >= g.qy) return;
    } else {                                        // ghost columns of interior rows
        const int rows_per_block = 256 / (2 * ng);
        const int r = (b - nrowblk) * rows_per_block + t / (2 * ng);
        const int kx = t % (2 * ng);
        if (r >= g.nx || t >= rows_per_block * 2 * ng) return;
        i = g.ilo + r;
        j = (kx < ng) ? kx : g.jhi + 1 + (kx - ng);
    }
    const size_t k = (size_t)i * g.pitch + j;
#pragma unroll
    for (int n = 0; n < 4; n++) {
        const pyro::BcMap mx = pyro::bc_map(g.ilo, g.ihi, ng, bc[n * 4 + 0], bc[n * 4 + 1], true);
        const pyro::BcMap my = pyro::bc_map(g.jlo, g.jhi, ng, bc[n * 4 + 2], bc[n * 4 + 3], true);
        const int si = pyro::bc_src(mx, i, g.ilo, g.ihi), sj = pyro::bc_src(my, j, g.jlo, g.jhi);
        const bool neg = ((i < g.ilo && mx.odd_lo) || (i > g.ihi && mx.odd_hi)) !=
                         ((j < g.jlo && my.odd_lo) || (j > g.jhi && my.odd_hi));
        const double v = src[n * g.plane + (size_t)si * g.pitch + sj];
        const double w = neg ? -v : v;
        cur[n * g.plane + k] = w;
        if (alt) alt[n * g.plane + k] = w;
    }
}
__global__ __launch_bounds__(256) void k_fill_frame2(const double *src, double *cur, double *alt,
                                                     Geom g, const int *__restrict__ bc)
{
    fill_frame2_piece(src, cur, alt, g, bc, (int)blockIdx.x, (int)threadIdx.x);
}

// (x sides of a slab that are cuts -- PYROHIP_BC_HALO -- are identity maps: the halo rows are data
// that arrived with the exchange, the y rule runs along them like along an interior row and the
// other buffer's frame takes a copy, which the exchange posted by the coming step overwrites)
static bool frame_fill_ok(const pyrohip_state *s, bool halo_ok = false, bool sph_ok = false)
{
    if (s->nvar != 4 || (s->nb_set && !halo_ok) || s->user_bc || s->ramp_bc || (s->sph && !sph_ok) || !s->alt_base)
        return false;
    for (int k = 0; k < 16; k++) {
        const int b = s->bc[k];
        if (b != PYROHIP_BC_OUTFLOW && b != PYROHIP_BC_REFLECT_EVEN && b != PYROHIP_BC_REFLECT_ODD &&
            b != PYROHIP_BC_PERIODIC && !(halo_ok && (k % 4) < 2 && b == PYROHIP_BC_HALO))
            return false;
    }
    return true;
}

constexpr int kPolicyThreads = 1024;
// The driver's compute_timestep (simulation_null.py:222-244) between two steps of a run that
// advances on the device (dt_policy_apply, common.h), from the CFL minimum the previous step
// kernel left in device memory.
__device__ __forceinline__ void dt_policy_block(StepScalars *S, const double *cflmin,
                                                const int *flag, double *dts, int slot,
                                                int final_call, const double *part, int nparts,
                                                double *minout, int flag_mask)
{
    // the CFL minimum of the previous step: already reduced (cflmin), or still the
    // per-workgroup partials of the tile kernel (part: reduced here, kept in minout)
    // (1024 threads, four loads in flight each: 40 000 partials at 16384^2 -- with 256 threads
    // and one dependent load after the other this took 36 us at 8192^2, 1.4 % of the step)
    __shared__ double red[kPolicyThreads];
    __shared__ double cmin_s;
    if (part != nullptr) {
        double m0 = INFINITY, m1 = INFINITY, m2 = INFINITY, m3 = INFINITY;
        const int nt = blockDim.x;
        int i = threadIdx.x;
        for (; i + 3 * nt < nparts; i += 4 * nt) {
            const double a = part[i], b = part[i + nt], c2 = part[i + 2 * nt], d = part[i + 3 * nt];
            m0 = fmin(m0, a); m1 = fmin(m1, b); m2 = fmin(m2, c2); m3 = fmin(m3, d);
        }
        for (; i < nparts; i += nt) m0 = fmin(m0, part[i]);
        const double m = fmin(fmin(m0, m1), fmin(m2, m3));
        red[threadIdx.x] = m;
        __syncthreads();
        for (int w = blockDim.x / 2; w > 0; w >>= 1) {
            if ((int)threadIdx.x < w) red[threadIdx.x] = fmin(red[threadIdx.x], red[threadIdx.x + w]);
            __syncthreads();
        }
        if (threadIdx.x == 0) { cmin_s = red[0]; *minout = red[0]; }
    } else if (threadIdx.x == 0) {
        cmin_s = *cflmin;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    // (raised by the step that just ran: it does not count)
    pyro::dt_policy_apply(S, cmin_s, (*flag & flag_mask) != 0, dts, slot, final_call);
}
__global__ __launch_bounds__(kPolicyThreads) void k_dt_policy(StepScalars *S, const double *cflmin,
                                                   const int *flag, double *dts, int slot,
                                                   int final_call, const double *part, int nparts,
                                                   double *minout, int flag_mask)
{
    dt_policy_block(S, cflmin, flag, dts, slot, final_call, part, nparts, minout, flag_mask);
}
// The two small launches between two steps of a device-side run in ONE (round 6): the ghost
// frames of both buffers (k_fill_frame2: reads the state the last step left) and the driver's dt
// policy (k_dt_policy: reads that step's CFL partials) do not depend on each other.  Workgroups
// of 1024 threads: the first nfill hold four 256-thread pieces of the fill each, the last one
// runs the policy.  One launch and its gap less per step (8 us of a 0.68 ms step at 4096^2).
__global__ __launch_bounds__(kPolicyThreads) void k_fill_frame2_policy(
    const double *src, double *cur, double *alt, Geom g, const int *__restrict__ bc, int npieces,
    StepScalars *S, const double *cflmin, const int *flag, double *dts, int slot, const double *part,
    int nparts, double *minout)
{
    if (blockIdx.x + 1 == gridDim.x) {
        dt_policy_block(S, cflmin, flag, dts, slot, 0, part, nparts, minout, 1);
        return;
    }
    const int piece = (int)blockIdx.x * 4 + (int)threadIdx.x / 256;
    if (piece < npieces) fill_frame2_piece(src, cur, alt, g, bc, piece, (int)threadIdx.x % 256);
}

namespace pyro {
// the small launches of a device-side run, for the other solvers' stepping loops (swe.hip)
int launch_fill_frame2(pyrohip_state *s, bool *done)
{
    *done = false;
    if (!frame_fill_ok(s)) return pyrohip_fill_bc(s, -1);
    pyrohip_ctx *c = s->ctx;
    const Geom &g = s->g;
    const int rows_per_block = 256 / (2 * g.ng);
    const int nblk = 2 * g.ng * ((g.qy + 255) / 256) + (g.nx + rows_per_block - 1) / rows_per_block;
    PYRO_LAUNCH(c, "k_fill_frame2", k_fill_frame2, dim3(nblk), dim3(256), 0, (const double *)s->d, s->d,
                s->alt_base + geom_lead(g), g, (const int *)s->d_bc);
    PYRO_CHECK_HIP(hipGetLastError());
    *done = true;
    return 0;
}

// A device-side run whose last iterations were inactive (past tmax, after an invalid state) has
// kept filling / copying ghost frames between the two buffers on those iterations: the frame of
// the buffer that holds the final state then depends on their parity.  What a single step leaves
// there is the filled ghost frame of the state BEFORE the last step that advanced -- whose
// interior sits untouched in the other buffer (inactive launches store nothing): rebuild it.
int restore_frame_after_inactive(pyrohip_state *s, int steps, int max_steps, bool halo_ok, bool sph_ok)
{
    if (steps < 1 || steps >= max_steps || !s->alt_base || !frame_fill_ok(s, halo_ok, sph_ok)) return 0;
    pyrohip_ctx *c = s->ctx;
    const Geom &g = s->g;
    const int rows_per_block = 256 / (2 * g.ng);
    const int nblk = 2 * g.ng * ((g.qy + 255) / 256) + (g.nx + rows_per_block - 1) / rows_per_block;
    PYRO_LAUNCH(c, "k_fill_frame2", k_fill_frame2, dim3(nblk), dim3(256), 0,
                (const double *)(s->alt_base + geom_lead(g)), s->d, (double *)nullptr, g, (const int *)s->d_bc);
    PYRO_CHECK_HIP(hipGetLastError());
    PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

// both in ONE launch (k_fill_frame2_policy) where the frame fill is an index map; *merged = false and
// nothing launched otherwise (the caller takes the two launches above)
int launch_fill_frame2_policy(pyrohip_state *s, StepScalars *S, const double *cflmin, const int *flag, double *dts,
                              int slot, const double *part, int nparts, double *minout, bool *merged)
{
    *merged = false;
    if (!frame_fill_ok(s)) return 0;
    pyrohip_ctx *c = s->ctx;
    const Geom &g = s->g;
    const int rows_per_block = 256 / (2 * g.ng);
    const int nblk = 2 * g.ng * ((g.qy + 255) / 256) + (g.nx + rows_per_block - 1) / rows_per_block;
    PYRO_LAUNCH(c, "k_fill_frame2_policy", k_fill_frame2_policy, dim3((nblk + 3) / 4 + 1), dim3(kPolicyThreads), 0,
                (const double *)s->d, s->d, s->alt_base + geom_lead(g), g, (const int *)s->d_bc, nblk, S, cflmin,
                flag, dts, slot, part, nparts, minout);
    PYRO_CHECK_HIP(hipGetLastError());
    *merged = true;
    return 0;
}

int launch_dt_policy(pyrohip_ctx *c, StepScalars *S, const double *cflmin, const int *flag, double *dts,
                     int slot, int final_call, const double *part, int nparts, double *minout)
{
    PYRO_LAUNCH(c, "k_dt_policy", k_dt_policy, dim3(1), dim3(kPolicyThreads), 0, S, cflmin, flag, dts, slot,
                final_call, part, nparts, minout, 1);
    PYRO_CHECK_HIP(hipGetLastError());
    return 0;
}
}  // namespace pyro

extern "C" {

int pyrohip_comp_evolve(pyrohip_state *s, const pyrohip_comp_params *p, double cfl,
                        pyrohip_dt_policy *pol, int max_steps, int *steps_done, double *dts_out)
{
    PYRO_TRY(check_comp(s, p));
    PYRO_REQUIRE(pol && steps_done, "NULL argument");
    PYRO_REQUIRE(max_steps >= 1, "max_steps must be positive");
    PYRO_REQUIRE(p->kernel_set != 0, "the staged kernel set steps from the host (kernel_set 0)");
    PYRO_REQUIRE(p->riemann >= 0 && p->riemann <= 2, "riemann must be 0 (HLLC), 1 (CGF) or 2 (HLLC_lm)");
    // (a SphericalPolar grid steps on the device where its step is one launch: comp_can_fuse_sph)
    const bool sphf = s->sph != nullptr && comp_can_fuse_sph(s, p);
    PYRO_REQUIRE((!s->sph || sphf) && !s->user_bc && !s->ramp_bc && !p->do_sponge && !s->ext_old,
                 "device-side stepping: standard boundaries, no sponge, no host-evaluated source; "
                 "a SphericalPolar grid needs CGF and outflow / reflect / periodic sides "
                 "(use pyrohip_comp_dt / pyrohip_comp_step)");
    pyrohip_ctx *c = s->ctx;
    if (!s->d_scal) PYRO_CHECK_HIP(hipMalloc((void **)&s->d_scal, sizeof(StepScalars)));
    if (s->dts_cap < max_steps + 1) {
        if (s->d_dts) PYRO_CHECK_HIP(hipFree(s->d_dts));
        s->d_dts = nullptr;
        // (not less than 1024: a run that asks for more steps call by call must not free + allocate every time)
        const int cap = max_steps + 1 > 1024 ? max_steps + 1 : 1024;
        PYRO_CHECK_HIP(hipMalloc((void **)&s->d_dts, (size_t)cap * sizeof(double)));
        s->dts_cap = cap;
    }
    StepScalars H;
    memset(&H, 0, sizeof(H));
    H.t = pol->t; H.dt_old = pol->dt_old; H.n = pol->n;
    H.tmax = pol->tmax; H.f0 = pol->init_tstep_factor; H.mx = pol->max_dt_change;
    H.fix_dt = pol->fix_dt; H.cfl = cfl; H.dx = p->dx; H.dy = p->dy;
    // the CFL minimum of the state as handed over: the one the last step of the previous call left, where
    // nothing has touched the state since (what pyrohip_comp_dt answers from as well) -- a pass over the
    // whole array otherwise (1.55 ms at 16384^2, 0.14 ms at 4096^2 per call)
    bool min_cached = cfl_min_cached(s, 0, p->gamma, p->dx, p->dy) && (!c->global_cfl || s->cfl_is_global);
    H.min0 = min_cached ? s->next_cfl_min : 0.0;
    H.keep0 = min_cached ? 1.0 : 0.0;
    PYRO_CHECK_HIP(hipMemcpyAsync(s->d_scal, &H, sizeof(H), hipMemcpyHostToDevice, c->stream));
    PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));      // H is on this stack frame
    if (c->global_cfl && c->comm != nullptr) {
        // decomposed run: all ranks keep their (global) minimum or none does -- a rank whose slab was written
        // since must reduce its array, and the others' kept minimum still counts that slab's OLD cells.  One
        // small all-reduce + read-back per call instead of a pass over the slab (0.9 ms at 2048 x 16384)
        PYRO_TRY(comm_allreduce_min_device(c, &s->d_scal->keep0));
        PYRO_CHECK_HIP(hipMemcpyAsync(c->reduce_host, &s->d_scal->keep0, sizeof(double), hipMemcpyDeviceToHost,
                                      c->stream));
        PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));
        min_cached = ((double *)c->reduce_host)[0] == 1.0;
    }
    PYRO_CHECK_HIP(hipMemsetAsync(s->d_flag, 0, sizeof(int), c->stream));
    const bool wave = !sphf && ((p->kernel_set == 2) ||
                                (p->kernel_set == -1 && wave_kernel_pays_ctu(s->g, p)));
    // SphericalPolar grid on the row-marching kernel (comp_sph_wave.hip: reads a filled frame)
    const bool sphw = sphf && ((p->kernel_set == 2) || (p->kernel_set == -1 && wave_kernel_pays(s->g)));
    const double *dmin = nullptr;
    bool first = true;
    int rc = 0;
    s->pend_part = nullptr;
    // steps after the first: the tile kernel applies the boundary rules itself where it can
    // (the first one needs filled ghost cells for the CFL minimum over the whole array)
    // (the spherical kernel reads every ghost cell through the boundary rules anyway)
    const bool fuse = (sphf && !sphw) || comp_can_fuse_fill(s, p, wave);
    pyrohip_comp_params pf = *p;
    // step_launches 1: the row-marching kernel as the ONLY launch of a step (single domain;
    // outflow / reflect / periodic sides, the same kind for the four variables): it reads ghost
    // cells through the boundary rules instead of a filled frame, and every wavefront derives
    // the step's dt from the CFL minima the previous launch left (k_ctu_wave<.., ONE>, common.h:
    // StepPolicy).  The first step keeps its fill + CFL + policy launches (the CFL minimum of
    // the state as handed over), the closing policy call is a launch, and the ghost cells of
    // the final state are filled once at the end.  Bit-identical to the three launches, and
    // measured no faster (profiles/r04_one_launch_step.txt: every wavefront of that instance starts
    // with ~5 us of dependent latency, the two small launches cost 1-2.5 % of a step; -1.5 % at
    // 16384^2): not the default.
    bool one_launch = wave && p->step_launches == 1 && !s->nb_set && !c->global_cfl &&
                      comp_can_fuse_fill(s, p, false);
    for (int k = 0; k < 16 && one_launch; k++) one_launch = (s->bc[k] != PYROHIP_BC_HALO);
    pyro::StepPolicy *d_pol = nullptr;
    if (one_launch) {
        // three sets of slots (+inf), the reduced minimum, then the StepPolicy of this call
        using namespace pyro;
        const size_t nw = 3 * (size_t)kPolSetWords + 1;
        const size_t bytes = nw * 8 + sizeof(StepPolicy);
        if (!s->d_polmem) PYRO_CHECK_HIP(hipMalloc((void **)&s->d_polmem, bytes));
        std::vector<unsigned long long> init(nw + (sizeof(StepPolicy) + 7) / 8, 0ull);
        const double inf = INFINITY;
        for (size_t k = 0; k < nw; k++) memcpy(&init[k], &inf, 8);
        StepPolicy sp;
        memset(&sp, 0, sizeof(sp));
        sp.S[0] = H;
        sp.slots = s->d_polmem;
        sp.dts = s->d_dts;
        memcpy(&init[nw], &sp, sizeof(sp));
        PYRO_CHECK_HIP(hipMemcpy(s->d_polmem, init.data(), bytes, hipMemcpyHostToDevice));
        d_pol = (StepPolicy *)(s->d_polmem + nw);
    }
    // (the step scalars of a one-launch run live in the StepPolicy: two copies, by step parity)
    StepScalars *const d_scal0 = one_launch ? &d_pol->S[0] : s->d_scal;
    for (int m = 0; m < max_steps && rc == 0; m++) {
        if (one_launch && m > 0) {
            s->pol_next = d_pol;
            s->pol_m = m;
            rc = p->fast_math ? fastm::comp_step_wave_ex(s, p, 0.0, d_scal0, &dmin)
                              : exact::comp_step_wave_ex(s, p, 0.0, d_scal0, &dmin);
            continue;
        }
        // ghost cells: halos of a slab first, then the boundary fill (pyro_sim.py:250-256)
        if (s->nb_set && c->comm != nullptr) rc = pyrohip_halo_exchange(s, s->nb_lo, s->nb_hi);
        pf.fuse_fill = (fuse && !first) ? 1 : 0;
        s->frame_prefilled = false;
        bool policy_done = false;
        if (rc == 0 && !pf.fuse_fill) {
            if ((wave && frame_fill_ok(s, true)) || (sphw && frame_fill_ok(s, false, true))) {      // fill + the other buffer's ghost frame: one launch
                const Geom &g = s->g;
                const int rows_per_block = 256 / (2 * g.ng);
                const int nblk = 2 * g.ng * ((g.qy + 255) / 256) + (g.nx + rows_per_block - 1) / rows_per_block;
                if (!first) {
                    // ... and the dt policy of this step in the same launch (k_fill_frame2_policy)
                    PYRO_LAUNCH(c, "k_fill_frame2_policy", k_fill_frame2_policy, dim3((nblk + 3) / 4 + 1),
                                dim3(kPolicyThreads), 0, (const double *)s->d, s->d, s->alt_base + geom_lead(g), g,
                                (const int *)s->d_bc, nblk, d_scal0, dmin, (const int *)s->d_flag, s->d_dts, m,
                                (const double *)s->pend_part, s->pend_n, const_cast<double *>(dmin));
                    policy_done = true;
                } else
                PYRO_LAUNCH(c, "k_fill_frame2", k_fill_frame2, dim3(nblk), dim3(256), 0, (const double *)s->d,
                            s->d, s->alt_base + geom_lead(g), g, (const int *)s->d_bc);
                const hipError_t e = hipGetLastError();
                if (e != hipSuccess) {
                    set_error(std::string("k_fill_frame2: ") + hipGetErrorString(e));
                    rc = (int)e;
                }
                s->frame_prefilled = (rc == 0);
            } else
                rc = pyrohip_fill_bc(s, -1);
        }
        if (rc) break;
        if (first) {   // CFL minimum of the state as handed over (full array, ghost cells filled)
            if (min_cached)
                dmin = &d_scal0->min0;      // (... the one the previous call's last step left)
            else if (sphf)
                rc = p->fast_math ? fastm::comp_cfl_min_device_sph(s, p, &dmin)
                                  : exact::comp_cfl_min_device_sph(s, p, &dmin);
            else
                rc = p->fast_math ? fastm::comp_cfl_min_device(s, p, &dmin)
                                  : exact::comp_cfl_min_device(s, p, &dmin);
            if (rc) break;
            // decomposed run: every rank steps with the minimum over ALL slabs -- also in the
            // first step of a call (found by running four ranks on one GPU: the slabs far from
            // the blast started every call with their own, larger dt; with two ranks the two
            // local minima are equal by symmetry and nothing showed)
            // (a kept minimum is the global one already: every rank kept it, see above)
            if (c->global_cfl && !min_cached) {
                rc = comm_allreduce_min_device(c, const_cast<double *>(dmin));
                if (rc) break;
                s->cfl_is_global = true;
            }
            first = false;
        }
        // (the minimum of the previous tile-kernel launch is taken here: pend_part)
        if (!policy_done)
        PYRO_LAUNCH(c, "k_dt_policy", k_dt_policy, dim3(1), dim3(kPolicyThreads), 0, d_scal0, dmin,
                    (const int *)s->d_flag, s->d_dts, m, 0, (const double *)s->pend_part,
                    s->pend_n, const_cast<double *>(dmin), 1);
        s->pend_part = nullptr;
        s->next_cfl_min = 1.0;      // "cached on the device": keeps a posted halo exchange valid
        s->cfl_kind = 0;
        s->pol_next = d_pol;    // (one launch per step: this is step 0, its dt is in S[0])
        s->pol_m = 0;
        if (sphw)
            rc = p->fast_math ? fastm::comp_step_wave_sph_ex(s, &pf, 0.0, s->d_scal, &dmin)
                              : exact::comp_step_wave_sph_ex(s, &pf, 0.0, s->d_scal, &dmin);
        else if (sphf)
            rc = p->fast_math ? fastm::comp_step_fused_sph_ex(s, &pf, 0.0, s->d_scal, &dmin)
                              : exact::comp_step_fused_sph_ex(s, &pf, 0.0, s->d_scal, &dmin);
        else if (wave)
            rc = p->fast_math ? fastm::comp_step_wave_ex(s, p, 0.0, d_scal0, &dmin)
                              : exact::comp_step_wave_ex(s, p, 0.0, d_scal0, &dmin);
        else
            rc = p->fast_math ? fastm::comp_step_fused_ex(s, &pf, 0.0, s->d_scal, &dmin)
                              : exact::comp_step_fused_ex(s, &pf, 0.0, s->d_scal, &dmin);
    }
    s->frame_prefilled = false;      // (an iteration that stopped between the fill and its step)
    s->pol_next = nullptr;
    PYRO_TRY(rc);
    // the closing policy call (one launch per step: on the scalars of the last step's parity,
    // with the minimum of the slots the last launch filled -- unused words hold +inf)
    StepScalars *const d_scalN = one_launch ? &d_pol->S[(max_steps - 1) & 1] : s->d_scal;
    if (one_launch) {
        s->pend_part = (double *)(s->d_polmem + (size_t)((max_steps - 1) % 3) * pyro::kPolSetWords);
        s->pend_n = pyro::kPolSetWords;
        dmin = (const double *)(s->d_polmem + 3 * (size_t)pyro::kPolSetWords);
    }
    hipLaunchKernelGGL(k_dt_policy, dim3(1), dim3(kPolicyThreads), 0, c->stream, d_scalN, dmin,
                       (const int *)s->d_flag, s->d_dts, max_steps, 1, (const double *)s->pend_part,
                       s->pend_n, const_cast<double *>(dmin),
                       one_launch ? (2 << ((max_steps - 1) & 1)) : 1);
    s->pend_part = nullptr;
    PYRO_CHECK_HIP(hipGetLastError());
    // the last step's halo exchange (posted on the halo stream) must have landed before
    // the call returns: the buffers may be read, written or freed by the caller next
    PYRO_TRY(comm_wait_halo(s));
    // the one round trip of the call: scalars, flag, last CFL minimum, the dt sequence
    char *hb = (char *)c->reduce_host;                       // 256 pinned bytes
    static_assert(sizeof(StepScalars) + 16 <= 256, "pinned scratch");
    PYRO_CHECK_HIP(hipMemcpyAsync(hb, d_scalN, sizeof(StepScalars), hipMemcpyDeviceToHost, c->stream));
    PYRO_CHECK_HIP(hipMemcpyAsync(hb + sizeof(StepScalars), s->d_flag, sizeof(int),
                                  hipMemcpyDeviceToHost, c->stream));
    PYRO_CHECK_HIP(hipMemcpyAsync(hb + sizeof(StepScalars) + 8, dmin, sizeof(double),
                                  hipMemcpyDeviceToHost, c->stream));
    if (dts_out)
        PYRO_CHECK_HIP(hipMemcpyAsync(dts_out, s->d_dts, (size_t)max_steps * sizeof(double),
                                      hipMemcpyDeviceToHost, c->stream));
    PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));
    memcpy(&H, hb, sizeof(H));
    // (a one-launch run raises per-launch bits of the flag; the policy keeps the verdict)
    const int flagv = (*(int *)(hb + sizeof(StepScalars)) & 1) | (H.dead ? 1 : 0);
    const double lastmin = *(double *)(hb + sizeof(StepScalars) + 8);
    // max_steps swaps were made; the last state that advanced sits H.steps swaps from the start
    if ((max_steps - H.steps) % 2) {
        double *old_base = s->base;
        s->base = s->alt_base;
        s->alt_base = old_base;
        s->d = s->base + geom_lead(s->g);
    }
    s->halo_pending = false;
    if (one_launch && H.steps >= 1) {
        // the steps read their ghost cells through the boundary rules and wrote none: the final
        // state's ghost cells hold the filled ghost cells of the state before its last step, like
        // the reference's array after evolve() -- that state sits untouched in the other buffer
        // (after an invalid step the other buffer is that step's debris: the state's own images)
        const Geom &g = s->g;
        const int rows_per_block = 256 / (2 * g.ng);
        const int nblk = 2 * g.ng * ((g.qy + 255) / 256) + (g.nx + rows_per_block - 1) / rows_per_block;
        const double *src = (flagv & 1) ? s->d : s->alt_base + geom_lead(g);
        PYRO_LAUNCH(c, "k_fill_frame2", k_fill_frame2, dim3(nblk), dim3(256), 0, src, s->d,
                    (double *)nullptr, g, (const int *)s->d_bc);
        PYRO_CHECK_HIP(hipGetLastError());
    }
    // (the row-marching kernels work on filled frames: the iterations past the last step that
    // advanced kept filling them -- rebuild the final state's, restore_frame_after_inactive)
    if (!one_launch && (wave || sphw) && !(flagv & 1))
        PYRO_TRY(restore_frame_after_inactive(s, H.steps, max_steps, true, sphw));
    // the minimum of the last launch belongs to the state only if that launch advanced it
    s->next_cfl_min = (H.steps == max_steps && !(flagv & 1)) ? lastmin : -1.0;
    s->cfl_kind = 0;
    s->cfl_par[0] = p->gamma; s->cfl_par[1] = p->dx; s->cfl_par[2] = p->dy;
    s->ghost_by_rules = false;      // a new time level: its ghost cells are stale until the next fill
    if (s->next_cfl_min <= 0.0) s->cfl_is_global = false;
    pol->t = H.t; pol->dt_old = H.dt_old; pol->n = H.n;
    *steps_done = H.steps;
    if (flagv & 1) {
        set_error("invalid state: min(rho) <= 0 or min(e) <= 0 on the interior "
                  "(compressible/simulation.py:68-71); the state is the one before that step");
        return PYROHIP_ERR_STATE;
    }
    return 0;
}

int pyrohip_comp_dt(pyrohip_state *s, const pyrohip_comp_params *p, double cfl, double *dt_out)
{
    PYRO_TRY(check_comp(s, p));
    PYRO_REQUIRE(dt_out, "dt_out is NULL");
    if (s->sph) {
        // (cached by the one-launch spherical step: whole-array minimum of the new state; every
        // other path that touches the state, the staged spherical set included, resets it)
        if (s->next_cfl_min > 0.0 && s->cfl_kind == 0) { *dt_out = cfl * s->next_cfl_min; return 0; }
        return p->fast_math ? fastm::comp_dt_sph(s, p, cfl, dt_out)
                            : exact::comp_dt_sph(s, p, cfl, dt_out);
    }
    return p->fast_math ? fastm::comp_dt(s, p, cfl, dt_out) : exact::comp_dt(s, p, cfl, dt_out);
}

int pyrohip_comp_dt_is_cached(pyrohip_state *s, int *flag)
{
    PYRO_REQUIRE(s && flag, "NULL argument");
    *flag = (s->next_cfl_min > 0.0 && s->cfl_kind == 0 && !s->user_bc && !s->ramp_bc) ? 1 : 0;
    return 0;
}

int pyrohip_comp_rk_dt_is_cached(pyrohip_state *s, int *flag)
{
    PYRO_REQUIRE(s && flag, "NULL argument");
    *flag = (s->next_cfl_min > 0.0 && s->cfl_kind == 1) ? 1 : 0;
    return 0;
}

int pyrohip_comp_dt_is_global(pyrohip_state *s, int *flag)
{
    PYRO_REQUIRE(s && flag, "NULL argument");
    *flag = (s->next_cfl_min > 0.0 && s->cfl_kind == 0 && s->cfl_is_global && !s->user_bc && !s->ramp_bc) ? 1 : 0;
    return 0;
}

int pyrohip_comp_step(pyrohip_state *s, const pyrohip_comp_params *p, double dt)
{
    PYRO_TRY(check_comp(s, p));
    PYRO_REQUIRE(dt > 0.0, "dt must be positive");
    PYRO_REQUIRE(p->kernel_set >= -1 && p->kernel_set <= 2, "kernel_set must be -1 (automatic), 0, 1 or 2");
    PYRO_REQUIRE(p->riemann >= 0 && p->riemann <= 2, "riemann must be 0 (HLLC), 1 (CGF) or 2 (HLLC_lm)");
    int rc;
    PYRO_REQUIRE(!s->ext_pending, "the corrector of the host-evaluated source has not run "
                                  "(pyrohip_comp_source_correct)");
    pyrohip_comp_params pf = *p;
    if (p->fuse_fill) {
        // ghost cells not filled by the caller: folded into the tile kernel where that
        // works, the ordinary fill first everywhere else
        // (the spherical one-launch kernel reads every ghost cell through the boundary rules)
        const bool wave = !s->sph && (p->kernel_set == 2 || (p->kernel_set == -1 && wave_kernel_pays_ctu(s->g, p)));
        if (!comp_can_fuse_fill(s, p, wave) && !comp_can_fuse_sph(s, p)) {
            PYRO_TRY(pyrohip_fill_bc(s, -1));
            pf.fuse_fill = 0;
        }
        p = &pf;
    }
    if (s->sph) {
        // compressible/simulation.py:206-208: no HLLC on a SphericalPolar grid
        PYRO_REQUIRE(p->riemann == 1, "a SphericalPolar grid needs the CGF Riemann solver");
        PYRO_REQUIRE(!s->user_bc && !s->ramp_bc && !s->heat && !s->ext_old,
                     "hse / ambient / ramp boundaries and heating are Cartesian-only");
        // the one-launch kernel reads EVERY ghost cell through the boundary rules: only where the
        // ghost cells are known to be those images -- the caller left the fill to the kernel, or the
        // last thing that wrote the state was the library's own full fill.  Ghost cells a host-side
        // boundary callback wrote (uploaded afterwards) are read from memory by the staged set.
        const bool fuse_sph = comp_can_fuse_sph(s, p) && (p->fuse_fill || s->ghost_by_rules);
        if (fuse_sph && (p->kernel_set == 2 || (p->kernel_set == -1 && wave_kernel_pays(s->g)))) {
            // the row-marching kernel reads the state's ghost cells from memory: filled here if
            // the caller left the fill to the step
            if (p->fuse_fill) PYRO_TRY(pyrohip_fill_bc(s, -1));
            rc = p->fast_math ? fastm::comp_step_wave_sph(s, p, dt) : exact::comp_step_wave_sph(s, p, dt);
        } else if (fuse_sph)
            rc = p->fast_math ? fastm::comp_step_fused_sph(s, p, dt) : exact::comp_step_fused_sph(s, p, dt);
        else
            rc = p->fast_math ? fastm::comp_step_sph(s, p, dt) : exact::comp_step_sph(s, p, dt);
    } else if (s->ext_old) {
        // host-evaluated source: staged kernels up to the predictor U* = U + dt S(U^n);
        // the caller evaluates S_h(U*) and finishes with pyrohip_comp_source_correct
        PYRO_REQUIRE(!s->heat && !s->ramp_bc, "a host-evaluated source excludes the heating "
                     "profile and the ramp boundary (which zeroes the source arrays, BC.py:198-200)");
        return p->fast_math ? fastm::comp_step_staged(s, p, dt) : exact::comp_step_staged(s, p, dt);
    } else if (p->kernel_set == 2 || (p->kernel_set == -1 && wave_kernel_pays_ctu(s->g, p)))
        rc = p->fast_math ? fastm::comp_step_wave(s, p, dt) : exact::comp_step_wave(s, p, dt);
    else if (p->kernel_set == 1 || p->kernel_set == -1)
        rc = p->fast_math ? fastm::comp_step_fused(s, p, dt) : exact::comp_step_fused(s, p, dt);
    else
        rc = p->fast_math ? fastm::comp_step_staged(s, p, dt) : exact::comp_step_staged(s, p, dt);
    s->ghost_by_rules = false;      // a new time level: its ghost cells are stale until the next fill
    if (rc == 0 && p->do_sponge) {
        PYRO_REQUIRE(p->sponge_rho_begin > p->sponge_rho_full,
                     "sponge_rho_begin must exceed sponge_rho_full (simulation.py:172)");
        rc = exact::comp_sponge(s, p, dt);
    }
    return rc;
}

int pyrohip_state_set_source(pyrohip_state *s, int which, pyrohip_state *src)
{
    PYRO_REQUIRE(s && (which == 0 || which == 1), "NULL state / which must be 0 (old) or 1 (new)");
    if (src) {
        PYRO_REQUIRE(src->nvar == 4 && s->nvar == 4 && src->g.nx == s->g.nx &&
                     src->g.ny == s->g.ny && src->g.ng == s->g.ng && src->ctx == s->ctx,
                     "the source state must match the 4-variable state it acts on");
    }
    if (which == 0) {
        PYRO_REQUIRE(!s->ext_pending, "the corrector of the previous step has not run");
        s->ext_old = src ? src->d : nullptr;
    } else {
        s->ext_new = src ? src->d : nullptr;
    }
    return 0;
}

int pyrohip_comp_source_correct(pyrohip_state *s, const pyrohip_comp_params *p, double dt)
{
    PYRO_TRY(check_comp(s, p));
    PYRO_REQUIRE(s->ext_pending && s->ext_old && s->ext_new,
                 "needs a predictor step (pyrohip_comp_step with a source set) and S_h(U*)");
    int rc = exact::comp_source_correct(s, p, dt);
    if (rc == 0 && p->do_sponge) {
        PYRO_REQUIRE(p->sponge_rho_begin > p->sponge_rho_full,
                     "sponge_rho_begin must exceed sponge_rho_full (simulation.py:172)");
        rc = exact::comp_sponge(s, p, dt);
    }
    return rc;
}

int pyrohip_comp_rk_dt(pyrohip_state *s, const pyrohip_comp_params *p, double cfl, double *dt_out)
{
    PYRO_TRY(check_comp(s, p));
    PYRO_REQUIRE(dt_out, "dt_out is NULL");
    // (left by the last stage of pyrohip_comp_rk_step: the minimum over the new interior, which
    // is the whole-array minimum of simulation.py:46-56 once the ghost cells are images)
    if (s->next_cfl_min > 0.0 && s->cfl_kind == 1) { *dt_out = cfl * s->next_cfl_min; return 0; }
    return exact::comp_rk_dt(s, p, cfl, dt_out);
}

// The whole Runge-Kutta step in nstages launches of the row-marching kernel (comp_wave.hip:
// comp_rk_step_wave)?  Single Cartesian domain, outflow / reflect / periodic sides (the same kind
// for the four variables: the stage states' ghost cells are read through index maps), no sponge,
// no heating profile, no host-evaluated source; kernel_set 2 or the library's choice from
// 2048^2 cells on.
static bool comp_rk_can_fuse(const pyrohip_state *y, const pyrohip_comp_params *p, const pyrohip_state *k,
                             int nstages)
{
    if (y->nvar != 4 || y->sph || y->nb_set || y->user_bc || y->ramp_bc || y->heat || y->ext_old ||
        p->do_sponge || y->g.ng < 4 || nstages < 2 || nstages > 4 || !k || k->nvar < 4 * nstages)
        return false;
    if (!(p->kernel_set == 2 || (p->kernel_set < 0 && wave_kernel_pays(y->g)))) return false;
    for (int sd = 0; sd < 4; sd++) {
        int kind0 = -1;
        for (int n = 0; n < 4; n++) {
            const int b = y->bc[n * 4 + sd];
            const int kind = (b == PYROHIP_BC_OUTFLOW) ? 0
                             : (b == PYROHIP_BC_REFLECT_EVEN || b == PYROHIP_BC_REFLECT_ODD) ? 1
                             : (b == PYROHIP_BC_PERIODIC) ? 2 : -1;
            if (kind < 0 || (n > 0 && kind != kind0)) return false;
            kind0 = kind;
        }
    }
    return true;
}

static int check_rk(pyrohip_state *y, const pyrohip_comp_params *p, pyrohip_state *k, int nstages,
                    const double *a, const double *b)
{
    PYRO_TRY(check_comp(y, p));
    PYRO_REQUIRE(k && a && b && k->ctx == y->ctx, "NULL argument / k state on another context");
    PYRO_REQUIRE(nstages >= 2 && nstages <= 4, "2 to 4 stages (RK2, TVD2, TVD3, RK4)");
    PYRO_REQUIRE(k->g.nx == y->g.nx && k->g.ny == y->g.ny && k->g.ng == y->g.ng && k->nvar >= 4 * nstages,
                 "the k state must have the geometry of the state and 4 planes per stage");
    PYRO_REQUIRE(p->riemann >= 0 && p->riemann <= 2, "riemann must be 0 (HLLC), 1 (CGF) or 2 (HLLC_lm)");
    for (int s = 0; s < nstages; s++)
        for (int j = s; j < nstages; j++)
            PYRO_REQUIRE(a[s * nstages + j] == 0.0, "explicit methods only (strictly lower triangular a)");
    return 0;
}

int pyrohip_comp_rk_can_fuse(pyrohip_state *y, const pyrohip_comp_params *p, pyrohip_state *k, int nstages,
                             int *flag)
{
    PYRO_REQUIRE(y && p && flag, "NULL argument");
    *flag = comp_rk_can_fuse(y, p, k, nstages) ? 1 : 0;
    return 0;
}

int pyrohip_comp_rk_step(pyrohip_state *y, const pyrohip_comp_params *p, pyrohip_state *k, double dt,
                         int nstages, const double *a, const double *b)
{
    PYRO_TRY(check_rk(y, p, k, nstages, a, b));
    PYRO_REQUIRE(dt > 0.0, "dt must be positive");
    PYRO_REQUIRE(comp_rk_can_fuse(y, p, k, nstages),
                 "pyrohip_comp_rk_step: single Cartesian domain, outflow / reflect / periodic sides, no sponge / "
                 "heating / host source, kernel_set 2 or a grid of >= 2048^2 cells (pyrohip_comp_rk_can_fuse; "
                 "otherwise stage by stage: pyrohip_comp_rk_rhs + pyrohip_state_lincomb)");
    const int rc = p->fast_math ? fastm::comp_rk_step_wave(y, p, k, nstages, a, b, dt, nullptr, nullptr)
                                : exact::comp_rk_step_wave(y, p, k, nstages, a, b, dt, nullptr, nullptr);
    y->ghost_by_rules = false;
    return rc;
}

// Up to max_steps steps of the compressible_rk driver loop (pyro_sim.py:241-281 with
// compressible_rk/simulation.py:46-104) without a host round trip per step: as pyrohip_comp_evolve,
// with the Runge-Kutta step above between the policy calls.
int pyrohip_comp_rk_evolve(pyrohip_state *y, const pyrohip_comp_params *p, pyrohip_state *k, int nstages,
                           const double *a, const double *b, double cfl, pyrohip_dt_policy *pol,
                           int max_steps, int *steps_done, double *dts_out)
{
    PYRO_TRY(check_rk(y, p, k, nstages, a, b));
    PYRO_REQUIRE(pol && steps_done && max_steps >= 1, "NULL argument / max_steps must be positive");
    PYRO_REQUIRE(comp_rk_can_fuse(y, p, k, nstages),
                 "device-side stepping: compressible_rk needs the conditions of pyrohip_comp_rk_step "
                 "(pyrohip_comp_rk_can_fuse)");
    pyrohip_state *s = y;
    pyrohip_ctx *c = s->ctx;
    PYRO_REQUIRE(!c->global_cfl, "device-side stepping: compressible_rk runs on a single domain");
    if (!s->d_scal) PYRO_CHECK_HIP(hipMalloc((void **)&s->d_scal, sizeof(StepScalars)));
    if (s->dts_cap < max_steps + 1) {
        if (s->d_dts) PYRO_CHECK_HIP(hipFree(s->d_dts));
        s->d_dts = nullptr;
        // (not less than 1024: a run that asks for more steps call by call must not free + allocate every time)
        const int cap = max_steps + 1 > 1024 ? max_steps + 1 : 1024;
        PYRO_CHECK_HIP(hipMalloc((void **)&s->d_dts, (size_t)cap * sizeof(double)));
        s->dts_cap = cap;
    }
    StepScalars H;
    memset(&H, 0, sizeof(H));
    H.t = pol->t; H.dt_old = pol->dt_old; H.n = pol->n;
    H.tmax = pol->tmax; H.f0 = pol->init_tstep_factor; H.mx = pol->max_dt_change;
    H.fix_dt = pol->fix_dt; H.cfl = cfl; H.dx = p->dx; H.dy = p->dy;
    const bool min_cached = cfl_min_cached(s, 1, p->gamma, p->dx, p->dy);      // (pyrohip_comp_evolve)
    H.min0 = min_cached ? s->next_cfl_min : 0.0;
    PYRO_CHECK_HIP(hipMemcpyAsync(s->d_scal, &H, sizeof(H), hipMemcpyHostToDevice, c->stream));
    PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));      // H is on this stack frame
    PYRO_CHECK_HIP(hipMemsetAsync(s->d_flag, 0, sizeof(int), c->stream));
    s->pend_part = nullptr;
    const double *dmin = nullptr;
    int rc = 0;
    for (int m = 0; m < max_steps && rc == 0; m++) {
        if (m == 0) {
            // the CFL minimum of the state as handed over: whole array, ghost cells filled
            // (or the one the previous call's last stage left)
            rc = pyrohip_fill_bc(s, -1);
            if (min_cached) dmin = &s->d_scal->min0;
            else if (rc == 0) rc = exact::comp_rk_cfl_min_device(s, p, &dmin);
            if (rc) break;
        }
        // steps after the first: the ghost frames of both buffers, the minimum of the last stage's
        // CFL partials and the dt policy in ONE launch (k_fill_frame2_policy, round 6) -- they were
        // k_fill_x + k_fill_y inside the step, k_copy_frame4, k_min_one and k_dt_policy: five
        // launches, 30 us of a 0.64 ms step at 2048^2
        s->frame_prefilled = false;
        if (m > 0 && frame_fill_ok(s)) {
            const Geom &g = s->g;
            const int rows_per_block = 256 / (2 * g.ng);
            const int nblk = 2 * g.ng * ((g.qy + 255) / 256) + (g.nx + rows_per_block - 1) / rows_per_block;
            PYRO_LAUNCH(c, "k_fill_frame2_policy", k_fill_frame2_policy, dim3((nblk + 3) / 4 + 1),
                        dim3(kPolicyThreads), 0, (const double *)s->d, s->d, s->alt_base + geom_lead(g), g,
                        (const int *)s->d_bc, nblk, s->d_scal, dmin, (const int *)s->d_flag, s->d_dts, m,
                        (const double *)s->pend_part, s->pend_n, const_cast<double *>(dmin));
            s->frame_prefilled = true;
        } else
        PYRO_LAUNCH(c, "k_dt_policy", k_dt_policy, dim3(1), dim3(kPolicyThreads), 0, s->d_scal, dmin,
                    (const int *)s->d_flag, s->d_dts, m, 0, (const double *)s->pend_part, s->pend_n,
                    const_cast<double *>(dmin), 1);
        s->pend_part = nullptr;
        rc = p->fast_math ? fastm::comp_rk_step_wave(s, p, k, nstages, a, b, 0.0, s->d_scal, &dmin)
                          : exact::comp_rk_step_wave(s, p, k, nstages, a, b, 0.0, s->d_scal, &dmin);
    }
    s->frame_prefilled = false;
    PYRO_TRY(rc);
    hipLaunchKernelGGL(k_dt_policy, dim3(1), dim3(kPolicyThreads), 0, c->stream, s->d_scal, dmin,
                       (const int *)s->d_flag, s->d_dts, max_steps, 1, (const double *)s->pend_part, s->pend_n,
                       const_cast<double *>(dmin), 1);
    s->pend_part = nullptr;
    PYRO_CHECK_HIP(hipGetLastError());
    char *hb = (char *)c->reduce_host;                       // 256 pinned bytes
    PYRO_CHECK_HIP(hipMemcpyAsync(hb, s->d_scal, sizeof(StepScalars), hipMemcpyDeviceToHost, c->stream));
    PYRO_CHECK_HIP(hipMemcpyAsync(hb + sizeof(StepScalars), s->d_flag, sizeof(int),
                                  hipMemcpyDeviceToHost, c->stream));
    PYRO_CHECK_HIP(hipMemcpyAsync(hb + sizeof(StepScalars) + 8, dmin, sizeof(double),
                                  hipMemcpyDeviceToHost, c->stream));
    if (dts_out)
        PYRO_CHECK_HIP(hipMemcpyAsync(dts_out, s->d_dts, (size_t)max_steps * sizeof(double),
                                      hipMemcpyDeviceToHost, c->stream));
    PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));
    memcpy(&H, hb, sizeof(H));
    const int flagv = (*(int *)(hb + sizeof(StepScalars)) & 1) | (H.dead ? 1 : 0);
    const double lastmin = *(double *)(hb + sizeof(StepScalars) + 8);
    // max_steps swaps were made; the last state that advanced sits H.steps swaps from the start
    if ((max_steps - H.steps) % 2) {
        double *old_base = s->base;
        s->base = s->alt_base;
        s->alt_base = old_base;
        s->d = s->base + geom_lead(s->g);
    }
    if (!(flagv & 1)) PYRO_TRY(restore_frame_after_inactive(s, H.steps, max_steps, false, false));
    s->next_cfl_min = (H.steps == max_steps && !(flagv & 1)) ? lastmin : -1.0;
    s->cfl_kind = 1;
    s->cfl_par[0] = p->gamma; s->cfl_par[1] = p->dx; s->cfl_par[2] = p->dy;
    s->cfl_is_global = false;
    s->ghost_by_rules = false;
    pol->t = H.t; pol->dt_old = H.dt_old; pol->n = H.n;
    *steps_done = H.steps;
    if (flagv & 1) {
        set_error("invalid state: min(rho) <= 0 or min(e) <= 0 on the interior "
                  "(compressible/simulation.py:68-71); the state is the one before that step");
        return PYROHIP_ERR_STATE;
    }
    return 0;
}

int pyrohip_comp_rk_rhs(pyrohip_state *y, const pyrohip_comp_params *p, pyrohip_state *k, int slot)
{
    PYRO_TRY(check_comp(y, p));
    PYRO_REQUIRE(k && k->ctx == y->ctx, "k state missing or on another context");
    PYRO_REQUIRE(k->g.nx == y->g.nx && k->g.ny == y->g.ny && k->g.ng == y->g.ng,
                 "k state must have the geometry of the stage state");
    PYRO_REQUIRE(slot >= 0 && 4 * (slot + 1) <= k->nvar, "slot outside the k state");
    PYRO_REQUIRE(p->riemann >= 0 && p->riemann <= 2, "riemann must be 0 (HLLC), 1 (CGF) or 2 (HLLC_lm)");
    if (p->do_sponge)
        PYRO_REQUIRE(p->sponge_rho_begin > p->sponge_rho_full,
                     "sponge_rho_begin must exceed sponge_rho_full (simulation.py:172)");
    // kernel_set 2, or the library's choice from 2048^2 cells on: one launch of the row-marching
    // kernel's method-of-lines instance; the staged kernels otherwise (small grids, the sponge)
    const bool wave = !p->do_sponge && y->nvar == 4 &&
                      (p->kernel_set == 2 || (p->kernel_set < 0 && wave_kernel_pays(y->g)));
    if (wave)
        return p->fast_math ? fastm::comp_rk_rhs_wave(y, p, k, slot) : exact::comp_rk_rhs_wave(y, p, k, slot);
    return p->fast_math ? fastm::comp_rk_rhs(y, p, k, slot) : exact::comp_rk_rhs(y, p, k, slot);
}

int pyrohip_comp_stage_dump(pyrohip_state *s, int stage_id, double *out)
{
    PYRO_REQUIRE(s && out, "NULL argument");
    return exact::comp_stage_dump(s, stage_id, out);
}

}  // extern "C"

extern "C" int pyrohip_comp_wave_geometry(int nx, int ny, int ng, int num_cus, int march_rows, int *out6)
{
    PYRO_REQUIRE(out6 && nx > 0 && ny > 0 && ng >= 0, "bad argument");
    return pyro::exact::comp_wave_geometry(nx, ny, ng, num_cus, march_rows, out6);
}
