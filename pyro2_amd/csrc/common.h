// Internal declarations shared by the libpyrohip.so translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pyrohip.h"

namespace pyro {

void set_error(const std::string &msg);

#define PYRO_CHECK_HIP(expr)                                                  \
    do {                                                                      \
        hipError_t _e = (expr);                                               \
        if (_e != hipSuccess) {                                               \
            ::pyro::set_error(std::string(#expr) + ": " +                     \
                              hipGetErrorString(_e));                         \
            return (int)_e;                                                   \
        }                                                                     \
    } while (0)

#define PYRO_REQUIRE(cond, msg)                                               \
    do {                                                                      \
        if (!(cond)) {                                                        \
            ::pyro::set_error(std::string(__func__) + ": " + (msg));          \
            return PYROHIP_ERR_ARG;                                           \
        }                                                                     \
    } while (0)

#define PYRO_TRY(expr)                                                        \
    do {                                                                      \
        int _rc = (expr);                                                     \
        if (_rc != 0) return _rc;                                             \
    } while (0)

// Geometry of one plane.  Rows are 128-B aligned at the FIRST INTERIOR cell
// (j = ng), so interior row loads/stores of a wave are aligned; `pitch` is a
// multiple of 16 doubles.
struct Geom {
    int nx, ny, ng;
    int qx, qy;
    int pitch;        // doubles between consecutive i rows
    size_t plane;     // doubles between consecutive variables
    int ilo, ihi, jlo, jhi;
};

inline Geom make_geom(int nx, int ny, int ng)
{
    Geom g;
    g.nx = nx; g.ny = ny; g.ng = ng;
    g.qx = nx + 2 * ng; g.qy = ny + 2 * ng;
    int lead = (16 - ng % 16) % 16;            // pad so that j = ng is aligned
    g.pitch = ((lead + g.qy + 15) / 16) * 16;
    g.plane = (size_t)g.qx * g.pitch;
    g.ilo = ng; g.ihi = ng + nx - 1; g.jlo = ng; g.jhi = ng + ny - 1;
    return g;
}
inline int geom_lead(const Geom &g) { return (16 - g.ng % 16) % 16; }

// scratch buffer grown on demand (device)
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need);
    void release();
};

}  // namespace pyro

namespace pyro {
// per-kernel HIP-event timing (bench.py roofline leg); off by default
struct ProfRec { const char *name; hipEvent_t a, b; };
struct Prof {
    bool on = false;
    std::vector<ProfRec> recs;
};
}  // namespace pyro

struct pyrohip_ctx {
    int device = 0;
    pyro::Prof prof;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    pyro::DevBuf staging;     // AoS <-> planar staging
    pyro::DevBuf reduce;      // reduction scratch (device)
    void *reduce_host = nullptr;  // pinned host words for scalar results
    void *comm = nullptr;     // ncclComm_t (scalar all-reduces, synchronous halo exchange)
    // overlapped halo exchange (comm.hip): its own communicator (split off `comm`)
    // on its own stream, so that the point-to-point traffic of step n+1's halos
    // runs beside the interior strips of step n and never interleaves with the
    // dt all-reduce on `stream`
    void *comm_halo = nullptr;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_boundary = nullptr, ev_halo = nullptr, ev_bdone = nullptr;
    int nranks = 1, rank = 0;
    bool global_cfl = false;  // all-reduce the step kernels' CFL minimum on the device
    int num_cus = 0;
    // one-round launches of the row-marching kernels: the two wavefronts of a SIMD tell each other how many
    // rows they have left (comp_wave.hip: wave_prio_feedback); [SIMD of the chip][wavefront slot], tagged per launch
    pyro::DevBuf prio_board;
    unsigned launch_seq = 0;
};

namespace pyro {
struct ProfScope {
    pyrohip_ctx *c;
    hipEvent_t b = nullptr;
    hipStream_t st;
    // (on: another stream of the context -- the boundary strips of a slab run on the halo stream)
    ProfScope(pyrohip_ctx *c_, const char *name, hipStream_t on = nullptr) : c(c_), st(on ? on : c_->stream)
    {
        if (!c->prof.on) return;
        hipEvent_t a;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { b = nullptr; return; }
        (void)hipEventRecord(a, st);
        c->prof.recs.push_back(ProfRec{name, a, b});
    }
    ~ProfScope() { if (b) (void)hipEventRecord(b, st); }
};
}  // namespace pyro

namespace pyro {
// in-place min all-reduce of one device double over the context's communicator
// on its stream; no-op without a communicator (comm.hip / tests/emu/comm_emu.cpp)
int comm_allreduce_min_device(pyrohip_ctx *c, double *d);
// can the halo exchange of a state be posted beside a running kernel?
bool comm_can_overlap(const pyrohip_state *s);
// post the exchange of the ng boundary rows of the planes at `d` (laid out like
// the state's) with the state's neighbours on the halo stream: it starts when
// everything queued on the context's stream so far is done and signals ev_halo
int comm_post_halo(pyrohip_state *s, double *d);
// The boundary strips of a slab's step run on the halo stream, BESIDE the interior strips on the
// context's stream (two launches in a row on one stream cost the first one's under-occupancy: 586
// wavefronts on 2048 slots for a whole strip time, 0.21 of 1.5 ms on a 2048 x 16384 slab,
// profiles/r06_slab2048x16384_*):
//   comm_fork_boundary: the halo stream waits for everything queued on the context's stream;
//     *bs = the stream to launch the boundary strips on (the context's own when there is no halo
//     stream: the emulator);
//   comm_post_halo_here: after that launch -- marks the boundary strips done (ev_bdone), posts the
//     exchange of the new boundary rows of the planes at d behind them, signals ev_halo;
//   comm_join_boundary: the context's stream waits for the boundary strips (their CFL partials).
int comm_fork_boundary(pyrohip_state *s, hipStream_t *bs);
int comm_post_halo_here(pyrohip_state *s, double *d);
int comm_join_boundary(pyrohip_state *s);
// a posted halo exchange writes the state's ghost rows from the halo stream: whoever
// reads or writes the state's memory on the context's stream (accessors, ghost fill,
// destroy) orders that stream behind it first.  No-op when nothing is pending.
int comm_wait_halo(pyrohip_state *s);
// plain ghost fill (x sides, then y sides) of cnt planes laid out like the
// state's planes n0.. with the boundary types of those variables (ctx.hip)
int fill_bc_planes(pyrohip_state *s, double *planes, int n0, int cnt);
}  // namespace pyro

// launch on the context's stream, bracketed by events when profiling is on
#define PYRO_LAUNCH(c, name, kern, grid, block, shmem, ...)                       \
    do {                                                                          \
        ::pyro::ProfScope _ps((c), (name));                                       \
        hipLaunchKernelGGL(kern, (grid), (block), (shmem), (c)->stream, __VA_ARGS__); \
    } while (0)
// ... on another stream of the context
#define PYRO_LAUNCH_ON(c, strm, name, kern, grid, block, shmem, ...)              \
    do {                                                                          \
        ::pyro::ProfScope _ps((c), (name), (strm));                               \
        hipLaunchKernelGGL(kern, (grid), (block), (shmem), (strm), __VA_ARGS__);  \
    } while (0)

namespace pyro {
// device copy of a SphericalPolar grid's geometry (pyrohip_state_set_geometry):
// 8 planes laid out like a state plane, 3 arrays of qy sines
struct SphGeom {
    double *base = nullptr;
    const double *Lx, *Ly, *Ax, *Ay, *V, *dlAx, *dlAy, *x2d;
    const double *sint, *sinb, *sinc;
    double xmin, ymin;
    // the 1-d factors of the planes (pyrohip_geom::rowf / colf), nullptr when the caller gave none:
    // 7 arrays of qxp doubles (A D F G Ly dlogAx x), 4 of qyp doubles (B C E T)
    const double *rowf = nullptr, *colf = nullptr;
    size_t qxp = 0, qyp = 0;
};
// row factors on the device: kSphRowStride doubles per ROW (the nine factors of a row side by side, 128-byte
// rows: a wavefront fetches a row's factors with one or two wide scalar loads instead of one load --
// and one wait -- per factor); column factors stay one array per factor (stride qyp: per-lane reads)
constexpr int kSphRowStride = 16;
}  // namespace pyro

namespace pyro {
// Per-step scalars of a run that advances on the device (pyrohip_comp_evolve):
// written by the one-thread kernel k_dt_policy (the driver's compute_timestep,
// simulation_null.py:222-244), read by the step kernels instead of their
// by-value parameters.  One instance per state, in device memory.
struct StepScalars {
    double dt, dtdx, dtdy, hdtV, dtdV;   // this step's dt and the quotients the kernels use
    double t, dt_old;                    // simulation time before this step, previous dt
    double tmax, f0, mx, fix_dt, cfl;    // policy parameters
    double dx, dy;
    long long n;                         // steps taken so far (driver's counter)
    int active;                          // this step advances the state (else: identity copy)
    int steps;                           // steps that advanced, this call
    int dead;                            // an invalid state was met: nothing runs any more
    double min0;                         // the CFL minimum the call starts from where the last call left it (device-side runs)
    double keep0;                        // decomposed runs: 1 if this rank kept it, min-reduced over the ranks before it is used
};

// The driver's compute_timestep (simulation_null.py:222-244) for a run that advances on the
// device: ends the previous step (t, n), decides whether the next one runs (t < tmax, state
// still valid) and derives its dt from the CFL minimum `cmin` of the state the previous step
// left.  ONE thread calls it: k_dt_policy (comp_api.hip) between two steps, or the last
// wavefront of the row-marching step kernel to finish (comp_wave.hip: one launch per step).
// IEEE operations in the reference's order, never contracted.
__device__ inline void dt_policy_apply(StepScalars *S, double cmin, bool invalid, double *dts,
                                       int slot, int final_call)
{
#pragma clang fp contract(off)
    invalid = invalid || S->dead != 0;       // (once invalid, always: the run has ended)
    S->dead = invalid ? 1 : 0;
    if (S->active && !invalid) { S->t += S->dt; S->n += 1; S->steps += 1; }
    S->active = 0;
    if (final_call) return;
    double dt = 0.0;
    const bool go = !invalid && (S->t < S->tmax);
    if (go) {
        if (S->fix_dt > 0.0) {
            dt = S->fix_dt;
        } else {
            dt = S->cfl * cmin;
            if (S->n == 0) dt = S->f0 * dt;
            else dt = fmin(S->mx * S->dt_old, dt);
            S->dt_old = dt;
        }
        if (S->t + dt > S->tmax) dt = S->tmax - S->t;
    }
    S->active = go ? 1 : 0;
    S->dt = dt;
    S->dtdx = dt / S->dx; S->dtdy = dt / S->dy;           // interface.py:106
    S->hdtV = (0.5 * dt) / (S->dx * S->dy);              // unsplit_fluxes.py:444-445
    S->dtdV = dt / (S->dx * S->dy);                      // simulation.py:375
    dts[slot] = go ? dt : -1.0;
}

// The row-marching step kernel as the ONLY launch of a step (pyrohip_comp_evolve).  Nothing in
// it waits for another wavefront or for the return of an atomic (a device-scope atomic with
// return costs the wavefront tens of microseconds at its end, a release fence writes the whole
// L2 back: both measured, profiles/r04_one_launch_step.txt):
//  - every wavefront of launch m folds its CFL minimum into one of 64 slots of set m % 3 by a
//    fire-and-forget unsigned 64-bit atomic minimum (positive doubles order like their bit
//    patterns: exact, order-independent);
//    (and raises bit 2 << (m & 1) of the positivity flag on an invalid state: its own bit, so
//    that a wavefront of the same launch that starts later does not take it for the previous
//    step's -- the policy remembers in StepScalars.dead);
//  - every wavefront of launch m + 1 starts by taking the minimum of set m % 3 and running the
//    driver's dt policy (dt_policy_apply) on a copy of S[m & 1] in its registers -- the same
//    IEEE operations in every wavefront; the wavefront of unit 0 also stores the result as
//    S[(m + 1) & 1] and dts[m + 1], and re-arms set (m + 2) % 3 for the launch after.
// The first policy call of a run (the CFL minimum of the state as handed over) and the closing
// one are launches of k_dt_policy.  One instance per state in device memory, set up per call.
constexpr int kPolSlots = 64, kPolStride = 16;          // a 128-byte line per slot
constexpr int kPolSetWords = kPolSlots * kPolStride;    // words of a set (the unused ones hold +inf)
struct StepPolicy {
    StepScalars S[2];
    unsigned long long *slots;   // 3 sets of kPolSetWords bit patterns
    double *dts;                 // dt sequence of the call
};
}  // namespace pyro

struct pyrohip_state {
    pyrohip_ctx *ctx = nullptr;
    pyro::Geom g;
    int nvar = 0;
    std::vector<int> bc;      // nvar*4
    double *base = nullptr;   // allocation
    double *d = nullptr;      // base + lead: plane n, row i: d + n*plane + i*pitch
    double *alt_base = nullptr;  // second buffer (fused kernels write the new
                                 // time level here, then the two are swapped)
    int *d_bc = nullptr;      // device copy of bc
    // compressible user boundaries (pyrohip_state_set_user_bc)
    bool user_bc = false;     // any HSE / AMBIENT code in bc
    bool user_bc_set = false;
    double ubc_gamma = 0.0, ubc_grav = 0.0, ubc_dy = 0.0, ubc_amb[4] = {0, 0, 0, 0};
    double *heat_base = nullptr, *heat = nullptr;   // heating profile plane (set_heating)
    // host-evaluated problem source (pyrohip_state_set_source): planes of two other
    // states (not owned), S_h(U^n) ghost-filled and S_h(U*); ext_pending: the
    // predictor ran, pyrohip_comp_source_correct has not yet
    const double *ext_old = nullptr, *ext_new = nullptr;
    int ext_pending = 0;
    // "ramp" boundary (pyrohip_state_set_ramp_bc)
    bool ramp_bc = false, ramp_set = false;
    double *d_x = nullptr;    // cell-centre x coordinates (qx)
    double r_cxoff = 0.0, r_post[4] = {0, 0, 0, 0}, r_pre[4] = {0, 0, 0, 0};
    double r_sfd[8] = {0}, r_sfu[8] = {0};
    // compressible work space (allocated on first use)
    double *work = nullptr;
    size_t work_planes = 0;
    int *d_flag = nullptr;    // positivity flag
    pyro::StepScalars *d_scal = nullptr;   // pyrohip_comp_evolve
    double *d_dts = nullptr;  // ... dt of every step of a call
    int dts_cap = 0;
    // CFL partials of the last tile-kernel launch whose minimum the next policy kernel
    // takes itself (device-side stepping of small grids: one launch less per step)
    double *pend_part = nullptr;
    int pend_n = 0;
    double *d_cval = nullptr; // per-variable ghost value of PYROHIP_BC_CONST sides
    pyro::SphGeom *sph = nullptr;   // SphericalPolar geometry (compressible solver)
    // x neighbours of a slab (pyrohip_state_set_neighbours, -1 = none) and
    // "the halo rows of this state are already on their way" (posted by the step
    // that produced it; only meaningful while next_cfl_min is still cached, i.e.
    // nothing touched the state since)
    int nb_lo = -1, nb_hi = -1;
    bool nb_set = false, halo_pending = false;
    // the ghost frame of the OTHER buffer already holds this step's boundary fill (written
    // together with this buffer's by k_fill_frame2, comp_api.hip): the step need not copy it
    bool frame_prefilled = false;
    // ... or the next launch of the row-marching kernel is the whole step (pyrohip_comp_evolve):
    // it reads ghost cells through the boundary rules and ends with the dt policy (StepPolicy)
    pyro::StepPolicy *pol_next = nullptr;
    int pol_m = 0;            // ... step of the call the next launch is
    unsigned long long *d_polmem = nullptr;   // 3 sets of slots, the reduced minimum, the StepPolicy
    double next_cfl_min = -1.0;  // min over interior of dx/(|u|+c) etc. of the
                                 // state after the last step (-1: unknown)
    bool cfl_is_global = false;  // ... already reduced over all ranks
    int cfl_kind = 0;            // ... 0: the CTU solver's quantity; 1: compressible_rk's; 2: swe's
    double cfl_par[3] = {0.0, 0.0, 0.0};   // ... of a device-side run: the (gamma | g, dx, dy) it was taken with
                                 // min 1 / ((|u|+c)/dx + (|v|+c)/dy) (comp_rk_step_wave)
    // the ghost cells hold exactly what the boundary rules (outflow / reflect / periodic index
    // maps) give for the current interior: set by a full pyrohip_fill_bc, dropped by anything
    // that writes the state (uploads -- a host-side boundary callback comes back as one --,
    // steps, linear combinations).  Kernels that read ghost cells THROUGH the rules instead of
    // from memory (k_ctu_fused_sph) need this or fuse_fill (a SphericalPolar run with
    // a host-side define_bc lost its ghost values)
    bool ghost_by_rules = false;
    bool stages_valid = false;   // swe: the work planes hold the stages of the LAST step (staged set)
};

// the rows-left board of the SIMD pairs for a row-marching launch whose wavefronts all sit on the chip at once or in
// two rounds (comp_wave.hip: the one with more rows left takes the priority): 2^16 SIMD numbers (XCC, SE, SH, CU,
// SIMD fields of the hardware id) x 2 slots, zeroed once, tagged per launch (1 .. 32767: never the zeroed board's 0)
inline int prio_board_acquire(pyrohip_ctx *c, int **board, int *tag)
{
    const size_t bytes = (size_t)2 * 65536 * sizeof(int);
    if (c->prio_board.bytes < bytes) {
        PYRO_TRY(c->prio_board.ensure(bytes));
        PYRO_CHECK_HIP(hipMemsetAsync(c->prio_board.p, 0, bytes, c->stream));
    }
    c->launch_seq = (c->launch_seq % 32767u) + 1u;
    *board = (int *)c->prio_board.p;
    *tag = (int)c->launch_seq;
    return 0;
}
// one-round launches: as many strips as wavefront slots -- the first n_extra of the ncb column strips are cut into
// nsb + 1 row strips of equal length instead of nsb (a SIMD left with ONE wavefront gets little out of it and
// the launch lasts as long as its last wavefront: tools/wave_timeline.py)
inline int wave_fill_extra(int ncb, int nsb, int nx, int slots)
{
    const int nreg = ncb * nsb;
    if (nsb < 2 || nreg >= slots || nx / (nsb + 1) < 8) return 0;
    return slots - nreg < ncb ? slots - nreg : ncb;
}
// the CFL minimum the last step of a device-side run left still describes the state: same kind of
// quantity, same (gamma | g, dx, dy), nothing has written the state since (every writer resets it)
inline bool cfl_min_cached(const pyrohip_state *s, int kind, double a, double dx, double dy)
{
    return s->next_cfl_min > 0.0 && s->cfl_kind == kind && s->cfl_par[0] == a && s->cfl_par[1] == dx &&
           s->cfl_par[2] == dy;
}

