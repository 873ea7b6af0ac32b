// Constant-coefficient Helmholtz multigrid (alpha - beta L) phi = f on a
// cell-centred 2^k x 2^k grid, V-cycles with red-black Gauss-Seidel.
//
// Replaces (reference file:line)
//   pyro/multigrid/MG.py:85-295    level hierarchy
//   pyro/multigrid/MG.py:529-542   _compute_residual
//   pyro/multigrid/MG.py:544-621   smooth (4 groups = 2 colours)
//   pyro/multigrid/MG.py:623-697   solve
//   pyro/multigrid/MG.py:699-778   v_cycle
//   pyro/mesh/patch.py:640-676     restrict
//   pyro/mesh/patch.py:678-736     prolong
//   pyro/mesh/array_indexer.py:98-111,150-274  norm, fill_ghost (ng = 1)
//
// Device layout: per level three planes v, f, r of (n+2) rows, ng = 1, row
// pitch a multiple of 16 doubles with the first interior cell 128-B aligned.
//
// Ghost cells: the reference refills ghosts after each colour (groups (0,0),
// (1,1) | fill | (1,0),(0,1) | fill).  Within a colour pass a ghost cell is
// read only by the one interior cell next to it, and it mirrors either that
// cell or a cell of the other colour, so the thread that updates a
// boundary-adjacent cell rewrites the dependent ghost itself right after its
// own read: same values as the reference at every read, no extra launch.
// Corner ghosts (never read by a 5-point stencil) are made exact by the
// explicit fill kernel, which runs wherever the reference calls fill_BC
// outside the colour loop.
#include "common.h"
#include "mg_internal.h"
#include "mg_march.h"
#include "reduce.h"
#include "stencil.h"
#include <cmath>
#include <type_traits>

namespace pyro {

constexpr int MG_MAXLEV = 24;

struct MGLevel {
    int n;          // interior cells per side
    int pitch;
    double dx;
    double *v, *f, *r;
    double *v2;     // second solution buffer (tile smoother ping-pong)
    double *c = nullptr, *ex = nullptr, *ey = nullptr;   // variable-coefficient mode
    double *a = nullptr, *gx = nullptr, *gy = nullptr;   // general mode: alpha, gamma_x, gamma_y
};

struct MGBC {
    int code[4];            // xl xr yl yr
    const double *val[4];   // inhomogeneous values (finest level) or nullptr
};

}  // namespace pyro

struct pyrohip_mg {
    pyrohip_ctx *ctx = nullptr;
    int nlevels = 0, nx = 0;
    double alpha = 0, beta = 0;
    int nsmooth = 0, nsmooth_bottom = 0;
    int bc[4] = {0, 0, 0, 0};
    pyro::MGLevel lev[pyro::MG_MAXLEV];
    double *pool = nullptr;       // all level planes
    double *old_phi = nullptr;    // finest-level copy for relative_error
    // solve() with the sums on the marching launch's tail: relative_error is taken ONCE, after
    // the last cycle, from the solution and the one before it -- which must then survive a
    // speculative cycle that is undone: a fourth finest-level buffer keeps it (the buffer the
    // capture would have turned into scratch).  134 MB of old solution and a division per cell
    // less in every cycle's last launch: 745 -> 707 us per 4096^2 V-cycle.
    double *older = nullptr, *older_base = nullptr;
    bool lazy_rel[2] = {false, false};      // per result slot: this cycle left relative_error for later
    double *bcval[4] = {nullptr, nullptr, nullptr, nullptr};  // device
    double source_norm = 0.0;
    int smoother = 1;             // 0: one launch per colour, 1: LDS tile smoother
    int kmax = 5;                 // red-black iterations fused per tile launch
    // ... on levels <= nsmall^2: one launch lasts as long as its slowest workgroup and
    // costs ~5 us before it starts; 10 iterations in one launch instead of 2 x 5 (the
    // apron doubles, which is free while most CUs idle).  Measured per V-cycle at
    // 512^2 / 2048^2 / 4096^2 (tools/mg_ab.sh): 5 everywhere 309 / 700 / 1590 us,
    // 10 up to 512^2 284 / 668 / 1529, 10 up to 1024^2 286 / 714 / 1578.
    int kmax_small = 10, kmax_small_tuned = 10;
    int nsmall = 512;
    // levels >= march_min^2 (0: none): the row-marching smoother (mg_march.hip), cut into
    // at most march_waves wavefronts (pyrohip_mg_set_tuning: the tests exercise the kernel
    // on small levels that way)
    int march_min = 2048;
    int march_waves = 0;   // 0: what the device holds
    // the first / last column strip (a select more per update) in shorter chunks: their
    // wavefronts are given 1 / march_side of the steps of the others (<= 1: cut like the others).
    // Measured per V-cycle at 2048^2 / 4096^2: 1.0 562 / 1041 us, 1.17 547 / 1031, 1.3 532 / 1020,
    // 1.5 535 / 1018, 1.8 527 / 1032
    double march_side = 1.5;
    int march_minrows = 32;   // rows a part stores, at least
    int coarse_kernel = 1;        // levels <= 64^2 in one LDS-resident workgroup
    int fuse_res_restrict = 1;   // down leg: residual + restriction in one pass
    // ... and inside solve(), where nobody reads r: on the tail of the marching smoother's
    // launch (MGMarch::tail: no pass of its own at all); likewise the two sums after a cycle
    int march_tail = 1;
    int tail_done = 0;            // the tail the last smoothing call carried
    int n_tail[3] = {0, 0, 0};    // launches with tail 1 / 2 so far (pyrohip_mg_tail_counts)
    bool diag_req = false, diag_done = false;   // solve(): the sums are wanted / were taken
    int diag_nb = 0;              // ... partials per sum
    double *diag_part = nullptr;
    int vc = 0;                   // 1: div(eta grad phi) = f; 2: general (alpha, beta, gamma)
    double *vc_pool = nullptr;
    double *gen_pool = nullptr;
    bool corners_stale[pyro::MG_MAXLEV] = {};   // v: corner ghosts not refreshed yet
    // v of the level is to be taken as 0 by its next smoothing launch (set by
    // the solve loop instead of a memset, consumed inside the same V-cycle)
    bool v_is_zero[pyro::MG_MAXLEV] = {};
    // Inside pyrohip_mg_solve the residual arrays are scratch: the passes that only need the
    // residual on the way to something else (its restriction on the way down, its norm after
    // the cycle) do not store it -- a level's worth of writes each -- and mark r of the level
    // stale; whoever asks for the array (plane(), mg_finest) gets it computed from the level's
    // current v and f first.  For the finest level that is what MG.py:668-671 leaves there;
    // the coarser levels' r (the reference keeps the down leg's there) is nobody's input.
    // (pyrohip_mg_tuning.lazy_residual = 0: always store.)
    bool lazy_r = true;
    // frozen choices that used to be environment knobs (pyrohip_mg_set_tuning)
    bool allow_pow2 = true;       // scaled right-hand side where the coefficients are powers of two
    int small_tiles = -1;         // workgroups aimed at on the small levels (-1: MG_SMALL_TILES_DEFAULT)
    int band_maxn = 2048;         // band smoother up to this level size
    bool band_genedge = false;    // band smoother: the general edge instance everywhere (tests)
    bool coarse_band64 = true;    // coarse kernel: the 64^2 level's sweeps in registers
    bool coarse_wave = true;      // coarse kernel: the levels up to 32^2 on one wavefront
    int speculate = 1;            // solve loop: 0 never launch ahead, 1 when likely needed, 2 always
    bool trace = false;           // developer aid: phase clocks of the band / coarse kernels
    bool spec_debug = false;      // developer aid: print cycles / launched ahead / undone per solve
    bool in_solve = false;
    bool r_stale[pyro::MG_MAXLEV] = {};
    // the solve loop's copy of the solution before the cycle (relative_error): the first
    // smoothing launch of the cycle on the finest level reads v from one buffer and writes
    // another -- the buffer it read IS that copy, the previous copy becomes the scratch buffer
    bool capture_old = false, old_captured = false;
    hipEvent_t ev[2] = {nullptr, nullptr};   // solve loop: a cycle's norms have reached the host
};

namespace pyro {

__device__ __forceinline__ double ghost_lo(int code, double inner, const double *val, int idx,
                                           double dx)
{
    switch (code) {
    case PYROHIP_BC_OUTFLOW: return val ? inner - dx * val[idx] : inner;          // neumann
    case PYROHIP_BC_REFLECT_ODD: return val ? 2 * val[idx] - inner : -inner;      // dirichlet
    case PYROHIP_BC_CONST: return 0.0;   // incompressible_viscous/BC.py:39-42 on "v"
    default: return inner;                                                       // reflect-even
    }
}
__device__ __forceinline__ double ghost_hi(int code, double inner, const double *val, int idx,
                                           double dx)
{
    switch (code) {
    case PYROHIP_BC_OUTFLOW: return val ? inner + dx * val[idx] : inner;
    case PYROHIP_BC_REFLECT_ODD: return val ? 2 * val[idx] - inner : -inner;
    case PYROHIP_BC_CONST: return 0.0;
    default: return inner;
    }
}

// homogeneous ghost value (val == nullptr above) as a select, not a branch:
// +inner, -inner or 0.  A taken branch costs more than the dozen fp64
// operations of a cell update (no branch prediction); measured on the bottom
// solve: 0.48 -> 0.17 us per colour sweep (gpurun_out/mgc_probe3/4.log).
__device__ __forceinline__ double ghost_h(int code, double inner)
{
    const double g = (code == PYROHIP_BC_REFLECT_ODD) ? -inner : inner;
    return (code == PYROHIP_BC_CONST) ? 0.0 : g;
}

// explicit ghost fill, x sides (all j) -- array_indexer.py:163-221 with ng=1
__global__ void k_mg_fill_x(double *__restrict__ a, int n, int pitch, double dx, MGBC bc)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > n + 1) return;
    const size_t lo = (size_t)1 * pitch + j, hi = (size_t)n * pitch + j;
    if (bc.code[0] == PYROHIP_BC_PERIODIC) a[j] = a[hi];
    else a[j] = ghost_lo(bc.code[0], a[lo], bc.val[0], j, dx);
    if (bc.code[1] == PYROHIP_BC_PERIODIC) a[(size_t)(n + 1) * pitch + j] = a[lo];
    else a[(size_t)(n + 1) * pitch + j] = ghost_hi(bc.code[1], a[hi], bc.val[1], j, dx);
}
// y sides (all i, including the x ghosts just filled) -- :223-274
__global__ void k_mg_fill_y(double *__restrict__ a, int n, int pitch, double dx, MGBC bc)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n + 1) return;
    double *row = a + (size_t)i * pitch;
    if (bc.code[2] == PYROHIP_BC_PERIODIC) row[0] = row[n];
    else row[0] = ghost_lo(bc.code[2], row[1], bc.val[2], i, dx);
    if (bc.code[3] == PYROHIP_BC_PERIODIC) row[n + 1] = row[1];
    else row[n + 1] = ghost_hi(bc.code[3], row[n], bc.val[3], i, dx);
}

// one colour of red-black Gauss-Seidel (MG.py:591-599).
// colour 0: groups (0,0),(1,1)  -> (i-1)+(j-1) even;  colour 1: (1,0),(0,1)
// thread (t, i): j = 1 + 2t + ((i - 1 + colour) & 1)
__global__ __launch_bounds__(256) void k_mg_smooth(double *__restrict__ v,
                                                   const double *__restrict__ f, int n, int pitch,
                                                   double dx, double xcoeff, double ycoeff,
                                                   double denom, int colour, MGBC bc)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = 1 + blockIdx.y;
    const int j = 1 + 2 * t + ((i - 1 + colour) & 1);
    if (j > n) return;
    const size_t k = (size_t)i * pitch + j;
    const double vn = (f[k] + xcoeff * (v[k + pitch] + v[k - pitch]) +
                       ycoeff * (v[k + 1] + v[k - 1])) / denom;
    v[k] = vn;
    // dependent ghost cells
    if (i == 1) {
        if (bc.code[0] == PYROHIP_BC_PERIODIC) v[(size_t)(n + 1) * pitch + j] = vn;
        else v[j] = ghost_lo(bc.code[0], vn, bc.val[0], j, dx);
    }
    if (i == n) {
        if (bc.code[1] == PYROHIP_BC_PERIODIC) v[j] = vn;
        else v[(size_t)(n + 1) * pitch + j] = ghost_hi(bc.code[1], vn, bc.val[1], j, dx);
    }
    if (j == 1) {
        if (bc.code[2] == PYROHIP_BC_PERIODIC) v[(size_t)i * pitch + n + 1] = vn;
        else v[(size_t)i * pitch] = ghost_lo(bc.code[2], vn, bc.val[2], i, dx);
    }
    if (j == n) {
        if (bc.code[3] == PYROHIP_BC_PERIODIC) v[(size_t)i * pitch] = vn;
        else v[(size_t)i * pitch + n + 1] = ghost_hi(bc.code[3], vn, bc.val[3], i, dx);
    }
}


// ---------------------------------------------------------------------------
// LDS tile smoother: K complete red-black iterations per launch.
//
// A workgroup stages its TI x TJ tile of v and f plus an apron of H = 2K cells
// in LDS, runs 2K colour passes there (after pass s the outermost s apron
// rings are stale and no longer read), and writes the tile to the second
// solution buffer.  Every cell update uses the reference's expression on the
// same operands as the one-launch-per-colour kernel, so results are
// bit-identical; HBM traffic per iteration drops from 48 B/cell to
// (16*apron_factor + 8)/K B/cell (8.4 B at K = 5).
//
// Physical boundaries inside the staged region: the ghost ring is refreshed
// from its interior neighbour after every colour pass (the reference's
// fill_BC after groups (1,1) and (0,1), MG.py:598-599), so no ring is lost on
// that side.  Periodic sides of multi-tile levels are staged through wrapped
// indices and treated like interior.  Levels that fit one tile (n <= 64) run
// single = 1: the whole level incl. ghosts is staged once and ANY number of
// iterations (nsmooth, nsmooth_bottom) runs in one launch.
// ---------------------------------------------------------------------------
// Two instantiations:
//   <256, 0>    generic: region pitch = region width (run time); used for the
//               levels that fit one tile ("single": any number of iterations).
//   <1024, 128> wide: the staged region is (TI+4K) x (TJ+4K) <= 64 x 128 cells
//               with a fixed LDS pitch of 128.  Thread (wave w, lane h) owns
//               the region cells (w + 16 m, 2h + q), m < 4, q < 2 for the whole
//               launch; their right-hand sides stay in registers, so LDS holds
//               v only (64 KB: two 16-wave workgroups = 8 waves/SIMD per CU).
//               Measured at 4096^2, smooth(10): 651 us with v and f in LDS
//               (32 x 128 region, K = 3) -> 491 us with f in registers
//               -> ~350 us with the 64-row region and K = 5 (two launches).
constexpr int MG_SMALL_TILES_DEFAULT = 192;         // workgroups aimed at on small levels (0: off;
                                                   // env PYRO_MG_SMALL_TILES; measured per V-cycle at
                                                   // 512^2 / 2048^2: 64 232 / 450 us, 128 233 / 452,
                                                   // 192 229 / 448, 256 245 / 491 (the up-leg launch holds
                                                   // one workgroup per CU: more than 256 is two rounds))
constexpr int MGS_CELLS = 66 * 66;                 // single-tile levels: n <= 64
constexpr size_t MGS_LDS = (size_t)2 * MGS_CELLS * sizeof(double);
#ifndef PYRO_MGW_RI
#define PYRO_MGW_RI 64
#endif
constexpr int MGW_RI = PYRO_MGW_RI, MGW_LP = 128, MGW_NT = 16 * MGW_RI, MGW_KMAX = 5;
constexpr size_t MGW_LDS = (size_t)MGW_RI * MGW_LP * sizeof(double);   // v only; f in registers

struct MGTile {
    const double *vin, *f;
    double *vout;
    int n, pitch;
    double dx, xc, yc, denom, rdenom;   // rdenom = RN(1 / denom), see div_by
    double kx, ky;                      // xc * rdenom, yc * rdenom (POW2 variants, see mg_pow2)
    int K, TI, TJ, ntj, ntiles, single;
    MGBC bc;
    // up leg: v += prolong(coarse v) while staging (patch.py:678-736 + MG.py:
    // 745-748), instead of a separate pass over the level; nullptr: plain smooth
    const double *cv;
    int cpitch;
    int vin_zero;   // 1: take vin as 0 (down leg: MG.py:658-659 zeroes the coarse solutions)
    long long *trace;   // developer aid (PYRO_MG_TRACE): clock64() of workgroup 0 at phase marks
    int row0, row1; // interior rows the launch updates (whole level: 1, n; a slab of a
                    // decomposed level: its rows -- the 2K apron rows beyond them are read)
};

// Power-of-two coefficients (the Poisson problems: alpha = 0, beta = +-1 on a unit
// square give xc = yc = +-4^k and denom = 4 xc): scaling by a power of two
// commutes with rounding, so the reference's
//     v = ((f + xc (a + b)) + yc (c + d)) / denom                  [9 operations]
// is, bit for bit,
//     v = fma(ky, c + d, fma(kx, a + b, f rd)),  kx = xc rd, ky = yc rd, rd = 1 / denom
// -- every product is exact, each fma rounds once where the reference's addition
// rounds, and the final division only scales (barring overflow / underflow of the
// scaled values: |f rd| < 2^-1022 with f != 0 does not occur on these grids).
// Four operations instead of nine, with f rd prepared once per launch; the
// smoother is VALU bound.  mg_pow2: are xc, yc, 1 / denom all powers of two?
static bool mg_is_pow2(double x)
{
    int e;
    return x != 0.0 && std::isfinite(x) && std::fabs(std::frexp(x, &e)) == 0.5;
}
static bool mg_pow2(double xc, double yc, double denom, bool allow = true)
{
    return allow && mg_is_pow2(xc) && mg_is_pow2(yc) && mg_is_pow2(denom) &&
           (1.0 / denom) * denom == 1.0;
}

// a / b for a divisor that is the same in every cell, with rb = RN(1 / b)
// evaluated once on the host: Markstein's sequence q = a rb; e = a - b q
// (exact, one FMA); q + e rb.  With a correctly rounded reciprocal and a
// faithful first quotient the result is the correctly rounded quotient, i.e.
// the same bits as the IEEE division of the reference (barring over/underflow
// of the intermediates, far from the values on these grids), at 3 instead of
// ~14 VALU instructions -- the smoother is VALU bound (profiles/r01c_mg4096).
__device__ __forceinline__ double div_by(double a, double b, double rb)
{
    const double q = a * rb;
    const double e = fma(-b, q, a);
    return fma(e, rb, q);
}

__device__ __forceinline__ int mg_wrap(int g, int n)   // periodic image in [1, n]
{
    int w = (g - 1) % n;
    if (w < 0) w += n;
    return w + 1;
}

template <int NT, int LPC, bool POW2 = false>
__global__ __launch_bounds__(NT, LPC ? 8 : 1) void k_mg_smooth_tile(MGTile A)
{
    HIP_DYNAMIC_SHARED(double, lds)
    const int n = A.n;
    const bool per_i = (A.bc.code[0] == PYROHIP_BC_PERIODIC);
    const bool per_j = (A.bc.code[2] == PYROHIP_BC_PERIODIC);
    const int H = A.single ? 0 : 2 * A.K;
    int ti0, ti1, tj0, tj1, gi0, gi1, gj0, gj1;
    if (A.single) {
        ti0 = 1; ti1 = n; tj0 = 1; tj1 = n;
        gi0 = 0; gi1 = n + 1; gj0 = 0; gj1 = n + 1;
    } else {
        const int tile = xcd_tile(blockIdx.x, A.ntiles);
        ti0 = A.row0 + (tile / A.ntj) * A.TI; ti1 = min(ti0 + A.TI - 1, A.row1);
        tj0 = 1 + (tile % A.ntj) * A.TJ; tj1 = min(tj0 + A.TJ - 1, n);
        gi0 = ti0 - H; gi1 = ti1 + H; gj0 = tj0 - H; gj1 = tj1 + H;
        if (!per_i) { gi0 = max(gi0, 0); gi1 = min(gi1, n + 1); }
        if (!per_j) { gj0 = max(gj0, 0); gj1 = min(gj1, n + 1); }
    }
    const int RI = gi1 - gi0 + 1, RJ = gj1 - gj0 + 1;
    const int LP = LPC ? LPC : RJ;                     // LDS row pitch (generic variant)
    double *V = lds, *F = lds + (LPC ? MGW_RI * LPC : RI * RJ);
    // LDS index of region cell (r, c).  Wide variant: the two checkerboard
    // classes of the region are stored separately (class, row, c/2), so the 64
    // lanes of a wave, which work on ONE class of one row, touch consecutive
    // doubles -- and so do their four neighbours, which all belong to the other
    // class.  The plain row-major layout made every access stride-2 (measured:
    // 2.2 bank-conflict cycles per LDS-active cycle).
    constexpr int HALF = LPC ? MGW_RI * (LPC / 2) : 0;
    auto at = [&](int r, int c) -> int {
        return LPC ? ((r + c) & 1) * HALF + r * (LPC / 2) + (c >> 1) : r * LP + c;
    };
    const int tid = threadIdx.x;
    const bool wrap_i = per_i && !A.single, wrap_j = per_j && !A.single;

    // wide variant: thread (wave w, lane h) owns the region cells (w + 8m, 2h + q),
    // m < 4, q < 2, for the whole launch; their right-hand sides stay in
    // registers, so LDS only holds v (32 KB: four workgroups per CU instead of two)
    static_assert(LPC == 0 || (MGW_RI * 64) / NT == 4, "four region rows per wave");
    constexpr int WSTEP = NT >> 6;             // waves per workgroup = row stride
    double f00 = 0, f01 = 0, f10 = 0, f11 = 0, f20 = 0, f21 = 0, f30 = 0, f31 = 0;
    const int wv = tid >> 6, ln = tid & 63;
    if (LPC) {
        // explicit scalars (not an array): they must live in VGPRs
        auto stage = [&](int m, double &fa, double &fb) {
            const int r = wv + WSTEP * m;
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int c = 2 * ln + q;
                double vv = 0.0, ff = 0.0;
                if (r < RI && c < RJ) {
                    const int gi = wrap_i ? mg_wrap(gi0 + r, n) : gi0 + r;
                    const int gj = wrap_j ? mg_wrap(gj0 + c, n) : gj0 + c;
                    const size_t k = (size_t)gi * A.pitch + gj;
                    if (!A.vin_zero) vv = A.vin[k];
                    ff = A.f[k];
                    if (A.cv && gi >= 1 && gi <= n && gj >= 1 && gj <= n) {
                        // k_mg_prolong_add's expression for fine cell (gi-1, gj-1)
                        const int fi = gi - 1, fj = gj - 1;
                        const size_t ck = (size_t)(1 + (fi >> 1)) * A.cpitch + 1 + (fj >> 1);
                        const double c0 = A.cv[ck];
                        const double m_x = 0.5 * (A.cv[ck + A.cpitch] - A.cv[ck - A.cpitch]);
                        const double m_y = 0.5 * (A.cv[ck + 1] - A.cv[ck - 1]);
                        double e;
                        if (fi & 1) e = (fj & 1) ? c0 + 0.25 * m_x + 0.25 * m_y : c0 + 0.25 * m_x - 0.25 * m_y;
                        else        e = (fj & 1) ? c0 - 0.25 * m_x + 0.25 * m_y : c0 - 0.25 * m_x - 0.25 * m_y;
                        vv += e;
                    }
                }
                V[at(r, c)] = vv;
                if (q) fb = ff; else fa = ff;
            }
        };
        stage(0, f00, f01); stage(1, f10, f11); stage(2, f20, f21); stage(3, f30, f31);
    } else {
        for (int idx = tid; idx < RI * RJ; idx += NT) {
            const int r = idx / RJ, c = idx - r * RJ;
            const size_t k = (size_t)(gi0 + r) * A.pitch + (gj0 + c);
            V[idx] = A.vin[k];
            F[idx] = A.f[k];
        }
    }
    __syncthreads();

    // sides on which the staged region ends at the level's ghost ring
    const bool plo_i = (gi0 == 0) && !wrap_i, phi_i = (gi1 == n + 1) && !wrap_i;
    const bool plo_j = (gj0 == 0) && !wrap_j, phi_j = (gj1 == n + 1) && !wrap_j;
    const bool any_phys = plo_i || phi_i || plo_j || phi_j;   // uniform per workgroup
    const int npass = 2 * A.K;
    // Wide variant, per-thread sweep schedule (all of it fixed for the launch):
    // odd passes (colour 0) write checkerboard class P1, even passes the other
    // one; of the thread's two columns 2*ln + q the one in the written class is
    // qA on odd and qB = 1 - qA on even passes, the same for its four rows.  A
    // cell may be relaxed in pass s while all four neighbours are still valid,
    // i.e. while s <= its distance to the nearest non-physical edge of the
    // staged region (and only inside the level on physical sides): one integer
    // per cell, compared with s -- instead of re-deriving the updatable
    // rectangle and the lane predicates in every pass (that bookkeeping was
    // ~40 % of the VALU instructions of a pass, and the smoother is VALU bound).
    constexpr int HP = LPC ? LPC / 2 : 1;                   // row pitch of one class
    const int P1 = (gi0 + gj0) & 1;
    const int qA = (P1 + wv) & 1, qB = qA ^ 1;
    int sA0 = -1, sA1 = -1, sA2 = -1, sA3 = -1, sB0 = -1, sB1 = -1, sB2 = -1, sB3 = -1;
    int b0 = 0, b1 = 0, b2 = 0, b3 = 0;
    double fA0 = 0, fA1 = 0, fA2 = 0, fA3 = 0, fB0 = 0, fB1 = 0, fB2 = 0, fB3 = 0;
    if (LPC) {
        constexpr int BIG = 1 << 20;
        auto last_j = [&](int q) {
            const int c = 2 * ln + q, gj = gj0 + c;
            if (c >= RJ) return -1;
            const int lo = plo_j ? (gj >= 1 ? BIG : -1) : c;
            const int hi = phi_j ? (gj <= n ? BIG : -1) : gj1 - gj;
            return min(lo, hi);
        };
        const int sjA = last_j(qA), sjB = last_j(qB);
        auto plan = [&](int m, double fa, double fb, int &sa, int &sb, int &b, double &ga,
                        double &gb) {
            const int r = wv + WSTEP * m, gi = gi0 + r;
            int si = -1;
            if (r < RI) {
                const int lo = plo_i ? (gi >= 1 ? BIG : -1) : r;
                const int hi = phi_i ? (gi <= n ? BIG : -1) : gi1 - gi;
                si = min(lo, hi);
            }
            sa = min(si, sjA); sb = min(si, sjB);
            b = r * HP + ln;
            ga = qA ? fb : fa; gb = qA ? fa : fb;
            if (POW2) { ga *= A.rdenom; gb *= A.rdenom; }     // exact: f rd
        };
        plan(0, f00, f01, sA0, sB0, b0, fA0, fB0);
        plan(1, f10, f11, sA1, sB1, b1, fA1, fB1);
        plan(2, f20, f21, sA2, sB2, b2, fA2, fB2);
        plan(3, f30, f31, sA3, sB3, b3, fA3, fB3);
    }
    // pass 0 only refreshes the staged ghost cells from their interior
    // neighbours (the fill_BC that opens MG.smooth, MG.py:565), so the input
    // buffer's ghosts need not be current
    for (int s = 0; s <= npass; s++) {
        const int colour = (s - 1) & 1;
        // updatable cells (global indices): all four neighbours still valid
        const int ulo_i = plo_i ? 1 : gi0 + s, uhi_i = phi_i ? n : gi1 - s;
        const int ulo_j = plo_j ? 1 : gj0 + s, uhi_j = phi_j ? n : gj1 - s;
        const int ni = uhi_i - ulo_i + 1, nj = uhi_j - ulo_j + 1;
        if (s == 0) {
            // nothing to update
        } else if (LPC) {
            // class written / read in this pass; neighbours of (r, c): (r+-1, c)
            // at b +- HP and (r, c+1), (r, c-1) at b + q, b + q - 1 of the other class
            const bool odd = s & 1;
            double *Vo = V + (odd ? P1 : P1 ^ 1) * HALF;
            const double *Vn = V + (odd ? P1 ^ 1 : P1) * HALF;
            const double *Vq = Vn + (odd ? qA : qB);
            auto relax = [&](int b, int last, double fc) {
                if (s <= last) {
                    if (POW2)
                        Vo[b] = fma(A.ky, Vq[b] + Vq[b - 1], fma(A.kx, Vn[b + HP] + Vn[b - HP], fc));
                    else
                        Vo[b] = div_by(fc + A.xc * (Vn[b + HP] + Vn[b - HP]) +
                                       A.yc * (Vq[b] + Vq[b - 1]), A.denom, A.rdenom);
                }
            };
            if (odd) { relax(b0, sA0, fA0); relax(b1, sA1, fA1); relax(b2, sA2, fA2); relax(b3, sA3, fA3); }
            else     { relax(b0, sB0, fB0); relax(b1, sB1, fB1); relax(b2, sB2, fB2); relax(b3, sB3, fB3); }
        } else {
            const int half = (nj + 1) >> 1;
            for (int idx = tid; idx < ni * half; idx += NT) {
                const int ri = idx / half, h = idx - ri * half;
                const int gi = ulo_i + ri;
                const int off = (gi - 1 + ulo_j - 1 + colour) & 1;
                const int gj = ulo_j + off + 2 * h;
                if (gj > uhi_j) continue;
                const int c = (gi - gi0) * LP + (gj - gj0);
                V[c] = (F[c] + A.xc * (V[c + LP] + V[c - LP]) + A.yc * (V[c + 1] + V[c - 1])) /
                       A.denom;
            }
        }
        if (s > 0) __syncthreads();
        if (!any_phys) continue;
        // ghost refresh on the physical sides (edge cells only; corners are
        // never read by the 5-point stencil).  Homogeneous boundaries (every
        // level but a finest one with boundary values): selects instead of the
        // switch of ghost_lo / ghost_hi -- branches are what this phase costs.
        const bool hom = !(A.bc.val[0] || A.bc.val[1] || A.bc.val[2] || A.bc.val[3]);
        if (plo_i || phi_i)
            for (int gj = ulo_j + tid; gj <= uhi_j; gj += NT) {
                const int c = gj - gj0;
                if (plo_i) {
                    const double in = V[at(1, c)];
                    V[at(0, c)] = (A.bc.code[0] == PYROHIP_BC_PERIODIC) ? V[at(n, c)]
                                  : hom ? ghost_h(A.bc.code[0], in)
                                        : ghost_lo(A.bc.code[0], in, A.bc.val[0], gj, A.dx);
                }
                if (phi_i) {
                    const int rl = (n + 1) - gi0;
                    const double in = V[at(rl - 1, c)];
                    V[at(rl, c)] = (A.bc.code[1] == PYROHIP_BC_PERIODIC) ? V[at(1 - gi0, c)]
                                   : hom ? ghost_h(A.bc.code[1], in)
                                         : ghost_hi(A.bc.code[1], in, A.bc.val[1], gj, A.dx);
                }
            }
        if (plo_j || phi_j)
            for (int gi = ulo_i + tid; gi <= uhi_i; gi += NT) {
                const int r = gi - gi0;
                if (plo_j) {
                    const double in = V[at(r, 1)];
                    V[at(r, 0)] = (A.bc.code[2] == PYROHIP_BC_PERIODIC) ? V[at(r, n)]
                                  : hom ? ghost_h(A.bc.code[2], in)
                                        : ghost_lo(A.bc.code[2], in, A.bc.val[2], gi, A.dx);
                }
                if (phi_j) {
                    const int cl = (n + 1) - gj0;
                    const double in = V[at(r, cl - 1)];
                    V[at(r, cl)] = (A.bc.code[3] == PYROHIP_BC_PERIODIC) ? V[at(r, 1 - gj0)]
                                   : hom ? ghost_h(A.bc.code[3], in)
                                         : ghost_hi(A.bc.code[3], in, A.bc.val[3], gi, A.dx);
                }
            }
        __syncthreads();
    }

    // tile (plus the ghost cells next to it on physical sides) -> vout
    const int oi0 = (plo_i && ti0 == 1) ? 0 : ti0, oi1 = (phi_i && ti1 == n) ? n + 1 : ti1;
    const int oj0 = (plo_j && tj0 == 1) ? 0 : tj0, oj1 = (phi_j && tj1 == n) ? n + 1 : tj1;
    if (LPC) {
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int r = wv + WSTEP * m;
            const int gi = gi0 + r;
            if (gi < oi0 || gi > oi1) continue;
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int gj = gj0 + 2 * ln + q;
                if (gj >= oj0 && gj <= oj1) A.vout[(size_t)gi * A.pitch + gj] = V[at(r, 2 * ln + q)];
            }
        }
    } else {
        const int oni = oi1 - oi0 + 1, onj = oj1 - oj0 + 1;
        for (int idx = tid; idx < oni * onj; idx += NT) {
            const int r = idx / onj, c = idx - r * onj;
            const int gi = oi0 + r, gj = oj0 + c;
            A.vout[(size_t)gi * A.pitch + gj] = V[at(gi - gi0, gj - gj0)];
        }
    }
    // periodic sides staged through wrapped indices: the tile that owns row /
    // column n (1) also writes the ghost row / column 0 (n+1), so the other
    // kernels (residual, prolongation) find current edge ghosts
    if (wrap_i && (ti1 == n || ti0 == 1))
        for (int gj = tj0 + tid; gj <= tj1; gj += NT) {
            if (ti1 == n) A.vout[gj] = V[at(n - gi0, gj - gj0)];
            if (ti0 == 1) A.vout[(size_t)(n + 1) * A.pitch + gj] = V[at(1 - gi0, gj - gj0)];
        }
    if (wrap_j && (tj1 == n || tj0 == 1))
        for (int gi = ti0 + tid; gi <= ti1; gi += NT) {
            if (tj1 == n) A.vout[(size_t)gi * A.pitch] = V[at(gi - gi0, n - gj0)];
            if (tj0 == 1) A.vout[(size_t)gi * A.pitch + n + 1] = V[at(gi - gi0, 1 - gj0)];
        }
}


// ---------------------------------------------------------------------------
// Band variant of the wide tile smoother (same region, same staging rules, same
// expression on the same operands: bit-identical).  Measured on the kernel above:
// 56 cycles per 64 cell updates and SIMD whatever the arithmetic costs (9 or 4
// operations) -- it is bound by its LDS traffic, five 512-B reads and one write per
// wavefront update against 128-256 B per clock and CU.  Here wave w keeps R = 4
// CONSECUTIVE region rows R w ... R w + R - 1 (lane h: columns 2h, 2h+1) in
// registers for the whole launch: of the four neighbours of a cell the rows are the
// thread's own registers, one column neighbour is the thread's other cell and the
// other one comes from the neighbouring lane by a whole-wave DPP rotation.  LDS
// only carries the two band-edge rows to the neighbouring waves: 2 reads + 2 writes
// per pass and thread instead of 20 + 4.
// ---------------------------------------------------------------------------
#if !defined(PYRO_EMU)
template <int CTRL> __device__ __forceinline__ double mgb_dpp(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double mgb_from_lower(double v) { return mgb_dpp<0x13C>(v); }   // wave_ror:1
__device__ __forceinline__ double mgb_from_upper(double v) { return mgb_dpp<0x134>(v); }   // wave_rol:1
#define MGB_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define MGB_SCHED_FENCE() ((void)0)
__device__ __forceinline__ double mgb_from_lower(double v) { return __shfl_up(v, 1, 64); }
__device__ __forceinline__ double mgb_from_upper(double v) { return __shfl_down(v, 1, 64); }
#endif

// Ghost cells are not kept at all: a cell next to a physical boundary takes
// ghost_h(its own value) where the stencil asks for the ghost cell -- what the
// reference's fill_BC after every colour sweep stores there (MG.py:598-599) -- and
// the ghost cells next to the tile are written from their mirror cells at the end.
// No cell is excluded from a pass either: a cell beyond the still-valid part of the
// apron (or at a ghost position) computes something that no cell of the tile ever
// reads.  Homogeneous boundaries (a finest level with boundary VALUES: k_mg_smooth_tile).
// R region rows per wavefront, 64 / R wavefronts per workgroup.  R = 4 (1024 threads) is
// what runs; R = 8 (512 threads, 102 VGPRs without the prolongation) measured 22.1 ->
// 23.5 us per launch on the small levels, 29.9 -> 31.1 us at 2048^2.
// PROL: the launch that opens the up leg adds the prolonged coarse correction while
// staging.  That code (five coarse reads per cell, eight cells) is what needs 128 VGPRs;
// the other launches fit 64 and share a CU two workgroups at a time.
// EDGE: 0 no physical side on the level; 2 physical sides, a select of ghost_h(own value) per
// side and update (28 instructions around the 6 of the update: 4.5 us per launch, measured by
// running the EDGE = 0 code on a Dirichlet problem); 1 the same for sides whose ghost cell is
// +-(its mirror cell) (everything but the value-0 ghosts of PYROHIP_BC_CONST): the tiles at the
// top / right side are staged from a region that ENDS at the ghost row / column, so that the
// rows / columns next to a physical side sit at fixed places of the register layout (row 1:
// wavefront 0, register row 1; row n: wavefront 15, register row 2; column 1: lane 0, cell 1;
// column n: lane 63, cell 0), and x + s me in one fma stands for x + ghost -- one select per
// update and direction instead of two selects of a three-way rule.
template <bool POW2, int EDGE, bool PROL, int R>
__global__ __launch_bounds__(64 * (MGW_RI / R), (PROL || R == 8) ? 4 : 8) void k_mg_smooth_band(MGTile A)
{
    HIP_DYNAMIC_SHARED(double, lds)
    static_assert(MGW_RI == 64 && MGW_LP == 128 && (R == 4 || R == 8), "64 / R waves x R rows x 128 columns");
    constexpr int HP = MGW_LP / 2, HALF = MGW_RI * HP;
    const int n = A.n;
    const bool per_i = (A.bc.code[0] == PYROHIP_BC_PERIODIC);
    const bool per_j = (A.bc.code[2] == PYROHIP_BC_PERIODIC);
    const int H = 2 * A.K;
    const int tile = xcd_tile(blockIdx.x, A.ntiles);
    const int ti0 = A.row0 + (tile / A.ntj) * A.TI, ti1 = min(ti0 + A.TI - 1, A.row1);
    const int tj0 = 1 + (tile % A.ntj) * A.TJ, tj1 = min(tj0 + A.TJ - 1, n);
    int gi0 = ti0 - H, gi1 = ti1 + H, gj0 = tj0 - H, gj1 = tj1 + H;
    if (!per_i) { gi0 = max(gi0, 0); gi1 = min(gi1, n + 1); }
    if (!per_j) { gj0 = max(gj0, 0); gj1 = min(gj1, n + 1); }
    if (EDGE == 1) {
        static_assert(EDGE != 1 || R == 4, "register row of the top row");
        if (!per_i && gi1 == n + 1 && gi0 > 0) gi0 = n + 1 - (MGW_RI - 1);
        if (!per_j && gj1 == n + 1 && gj0 > 0) gj0 = n + 1 - (MGW_LP - 1);
    }
    const int RI = gi1 - gi0 + 1;
    double *V = lds;
    // LDS index of region cell (r, c): the two checkerboard classes apart, so that the
    // lanes of a wave (one class of one row) touch consecutive doubles
    auto at = [&](int r, int c) -> int { return ((r + c) & 1) * HALF + r * HP + (c >> 1); };
    const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
    // sides on which the staged region ends at the level's ghost ring
    const bool plo_i = (gi0 == 0) && !per_i, phi_i = (gi1 == n + 1) && !per_i;
    const bool plo_j = (gj0 == 0) && !per_j, phi_j = (gj1 == n + 1) && !per_j;
    const int c0 = A.bc.code[0], c1 = A.bc.code[1], c2 = A.bc.code[2], c3 = A.bc.code[3];
#ifndef PYRO_EMU
#define MGB_MARK(k) do { if (A.trace && blockIdx.x == (unsigned)A.ntiles / 2 && tid == 0) A.trace[k] = clock64(); } while (0)
#else
#define MGB_MARK(k) ((void)0)
#endif
    MGB_MARK(0);

    // ---- stage (prolongation of the up leg on the way).  Rows / columns beyond the
    // region load its last row / column again: inside the array, never read by the tile ----
    double v[R][2], f[R][2];
    int gi_[R], gj_[2];             // array row / column of the thread's cells (wrapped, clamped)
#pragma unroll
    for (int m = 0; m < R; m++) {
        const int g = gi0 + R * wv + m;
        const int w = g + (g < 1 ? n : 0) - (g > n ? n : 0);      // |apron| <= n: one correction wraps
        gi_[m] = per_i ? w : min(g, n + 1);
    }
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int g = gj0 + 2 * ln + q;
        int w = g + (g < 1 ? n : 0);
        w -= (w > n ? n : 0);
        w -= (w > n ? n : 0);                                     // lanes far beyond a narrow region
        gj_[q] = per_j ? w : min(g, n + 1);
    }
    {
        const double *src = A.vin_zero ? A.f : A.vin;             // zero solution: cache hits, dropped
#pragma unroll
        for (int m = 0; m < R; m++)
#pragma unroll
            for (int q = 0; q < 2; q++) {
                // 32-bit element offsets (a level is < 2^25 elements): scalar base + one VGPR
                const unsigned k = (unsigned)(gi_[m] * A.pitch + gj_[q]);
                const double x = src[k], y = A.f[k];
                v[m][q] = A.vin_zero ? 0.0 : x;
                f[m][q] = POW2 ? y * A.rdenom : y;                // exact: see mg_pow2
            }
        MGB_SCHED_FENCE();
    }
    if (PROL) {
        // k_mg_prolong_add's expression for the thread's R x 2 fine cells.  They lie over 2-3
        // coarse rows and 1-2 coarse columns, which of them depends only on the parity of the
        // tile's first row / column: the coarse values are loaded once into a small array that
        // is indexed at compile time (16 loads instead of 5 per cell = 40; the launch that opens
        // the up leg was 6 us slower than the others), cells outside the level drop the result
        const int gu0 = gi0 + R * wv, gv0 = gj0 + 2 * ln;     // the thread's first cell, unwrapped
        const int nc = n >> 1;
        auto prol = [&](auto pic, auto pjc) __attribute__((always_inline)) {
            constexpr int PI = decltype(pic)::value, PJ = decltype(pjc)::value;   // parities of gu0, gv0
            constexpr int NR = ((R - PI) >> 1) + 3, NC = ((2 - PJ) >> 1) + 3;
            const int C0 = ((gu0 + 1) >> 1) - 1, D0 = ((gv0 + 1) >> 1) - 1;      // first coarse row / column read
            double cc[NR][NC];
            unsigned ccol[NC];
#pragma unroll
            for (int b = 0; b < NC; b++) {
                const int d = D0 + b;
                int w = d + (d < 1 ? nc : 0);
                w -= (w > nc ? nc : 0);
                w -= (w > nc ? nc : 0);
                ccol[b] = (unsigned)(per_j ? w : min(max(d, 0), nc + 1));
            }
#pragma unroll
            for (int a = 0; a < NR; a++) {
                const int cr = C0 + a;
                const int w = cr + (cr < 1 ? nc : 0) - (cr > nc ? nc : 0);
                const unsigned base = (unsigned)((per_i ? w : min(max(cr, 0), nc + 1)) * A.cpitch);
#pragma unroll
                for (int b = 0; b < NC; b++) cc[a][b] = A.cv[base + ccol[b]];      // unused corners: dropped
            }
#pragma unroll
            for (int m = 0; m < R; m++) {
                const int g = gu0 + m;
                const bool rin = per_i || (g >= 1 && g <= n);
                const int ro = (m + 1 - PI) >> 1;
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int h = gv0 + q;
                    const bool in = rin && (per_j || (h >= 1 && h <= n));
                    const int co = (q + 1 - PJ) >> 1;
                    const double q0 = cc[ro + 1][co + 1];
                    const double m_x = 0.5 * (cc[ro + 2][co + 1] - cc[ro][co + 1]);
                    const double m_y = 0.5 * (cc[ro + 1][co + 2] - cc[ro + 1][co]);
                    const double tx = 0.25 * m_x, ty = 0.25 * m_y;    // x - y == x + (-y)
                    // fine cell (fi, fj) = (g - 1, h - 1): + for odd fi / fj
                    const bool fio = ((PI + m + 1) & 1) != 0, fjo = ((PJ + q + 1) & 1) != 0;
                    const double e = (q0 + (fio ? tx : -tx)) + (fjo ? ty : -ty);
                    v[m][q] += in ? e : 0.0;
                }
            }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        const int par = (gi0 & 1) * 2 + (gj0 & 1);            // the same for the whole tile
        if (par == 0) prol(I0{}, I0{});
        else if (par == 1) prol(I0{}, I1{});
        else if (par == 2) prol(I1{}, I0{});
        else prol(I1{}, I1{});
    }
    // LDS slots of the band-edge rows (this band's rows 0 and 3, both columns) and of
    // the rows next to the band (clamped: an edge band's outer rows are never read by
    // a cell that matters)
    const int r0 = R * wv;
    const int w00 = at(r0, 2 * ln), w01 = at(r0, 2 * ln + 1);
    const int w30 = at(r0 + R - 1, 2 * ln), w31 = at(r0 + R - 1, 2 * ln + 1);
    const int rb = max(r0 - 1, 0), rt = min(r0 + R, MGW_RI - 1);
    const int b0 = at(rb, 2 * ln), b1 = at(rb, 2 * ln + 1), t0 = at(rt, 2 * ln), t1 = at(rt, 2 * ln + 1);
    V[w00] = v[0][0]; V[w01] = v[0][1]; V[w30] = v[R - 1][0]; V[w31] = v[R - 1][1];
    // cells whose neighbour is a ghost cell (EDGE tiles)
    bool isTop[R], isBot[R], isE[2], isW[2];
#pragma unroll
    for (int m = 0; m < R; m++) {
        isTop[m] = EDGE == 2 && phi_i && gi0 + r0 + m == n;
        isBot[m] = EDGE == 2 && plo_i && gi0 + r0 + m == 1;
    }
#pragma unroll
    for (int q = 0; q < 2; q++) {
        isE[q] = EDGE == 2 && phi_j && gj0 + 2 * ln + q == n;
        isW[q] = EDGE == 2 && plo_j && gj0 + 2 * ln + q == 1;
    }
    // EDGE == 1: the fixed places (see above) and the ghost cells' signs (1 where there is none)
    const bool botW = EDGE == 1 && plo_i && wv == 0, topW = EDGE == 1 && phi_i && wv == MGW_RI / R - 1;
    const bool isWl = EDGE == 1 && plo_j && ln == 0, isEl = EDGE == 1 && phi_j && ln == 63;
    const double sBw = botW ? (c0 == PYROHIP_BC_REFLECT_ODD ? -1.0 : 1.0) : 1.0;
    const double sTw = topW ? (c1 == PYROHIP_BC_REFLECT_ODD ? -1.0 : 1.0) : 1.0;
    const double sWl = isWl ? (c2 == PYROHIP_BC_REFLECT_ODD ? -1.0 : 1.0) : 1.0;
    const double sEl = isEl ? (c3 == PYROHIP_BC_REFLECT_ODD ? -1.0 : 1.0) : 1.0;
    // the band is part of the region; beyond pass wave_last none of its rows is valid
    // any more (rows only: one scalar for the wave, the whole wave skips the pass)
    int wave_last = -1;
#pragma unroll
    for (int m = 0; m < R; m++) {
        const int r = r0 + m;
        if (r < RI) wave_last = max(wave_last, min(plo_i ? (1 << 20) : r, phi_i ? (1 << 20) : RI - 1 - r));
    }
#ifndef PYRO_EMU
    wave_last = __builtin_amdgcn_readfirstlane(wave_last);
#endif
    __syncthreads();
    MGB_MARK(1);
    MGB_MARK(2);

    // ---- 2K colour passes ----
    // class relaxed in pass s: (gi + gj + colour) even <=> (r + c + par) even; the
    // thread's cell of that class in row 4w + m is its column (par + m) & 1
    auto pass = [&](auto par_c) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_c)::value;
        // band-edge rows of the other class from the neighbouring waves
        const double below = V[(PAR & 1) ? b1 : b0];             // (row below the band, column of row 0's cell)
        const double above = V[((PAR + R - 1) & 1) ? t1 : t0];   // (row above the band, column of the last row's cell)
        // a pass only reads cells of the other class: every result goes straight back
#pragma unroll
        for (int m = 0; m < R; m++) {
            const int q = (PAR + m) & 1;
            const double me = v[m][q];
            double up = (m < R - 1) ? v[m < R - 1 ? m + 1 : R - 1][q] : above;     // row r + 1
            double dn = (m > 0) ? v[m > 0 ? m - 1 : 0][q] : below;     // row r - 1
            // the thread's other cell and the neighbouring lane's
            const double own = v[m][q ^ 1];
            const double oth = q ? mgb_from_upper(v[m][0]) : mgb_from_lower(v[m][1]);
            double e = q ? oth : own, w = q ? own : oth;               // columns c + 1, c - 1
            if (EDGE == 2) {
                up = isTop[m] ? ghost_h(c1, me) : up;
                dn = isBot[m] ? ghost_h(c0, me) : dn;
                e = isE[q] ? ghost_h(c3, me) : e;
                w = isW[q] ? ghost_h(c2, me) : w;
            }
            double si, sj;                                             // up + dn, e + w
            if (EDGE == 1 && m == 1) si = fma(botW ? me : dn, sBw, up);        // row 1: ghost below
            else if (EDGE == 1 && m == 2) si = fma(topW ? me : up, sTw, dn);   // row n: ghost above
            else si = up + dn;
            if (EDGE == 1 && q == 1) sj = fma(isWl ? me : w, sWl, e);          // column 1: ghost left
            else if (EDGE == 1) sj = fma(isEl ? me : e, sEl, w);               // column n: ghost right
            else sj = e + w;
            if (POW2)
                v[m][q] = fma(A.ky, sj, fma(A.kx, si, f[m][q]));
            else
                v[m][q] = div_by(f[m][q] + A.xc * si + A.yc * sj, A.denom, A.rdenom);
            if (m % 4 == 3) MGB_SCHED_FENCE();             // four rows interleaved, not R (VGPR budget)
        }
        V[(PAR & 1) ? w01 : w00] = v[0][PAR & 1];
        V[((PAR + R - 1) & 1) ? w31 : w30] = v[R - 1][(PAR + R - 1) & 1];
    };
    const int npass = 2 * A.K;
    for (int s = 1; s <= npass; s++) {
        const int par = (gi0 + gj0 + ((s - 1) & 1)) & 1;
        if (s <= wave_last) {
            if (par) pass(std::integral_constant<int, 1>{});
            else pass(std::integral_constant<int, 0>{});
        }
        __syncthreads();
        MGB_MARK(2 + s);
    }

    // ---- tile -> vout; on physical sides the ghost cells next to it from their mirror
    // cells; on periodic sides the tile that owns row / column n (1) also writes ghost
    // row / column 0 (n + 1), so that the other kernels find current edge ghosts ----
#pragma unroll
    for (int m = 0; m < R; m++) {
        const int gi = gi0 + r0 + m;                      // unwrapped
        if (gi < ti0 || gi > ti1) continue;
        const unsigned row = (unsigned)(gi * A.pitch);
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int gj = gj0 + 2 * ln + q;
            if (gj < tj0 || gj > tj1) continue;
            const double x = v[m][q];
            A.vout[row + gj] = x;
            if (per_i) {
                if (gi == n) A.vout[gj] = x;
                if (gi == 1) A.vout[(unsigned)((n + 1) * A.pitch + gj)] = x;
            } else {
                if (gi == 1) A.vout[gj] = ghost_h(c0, x);
                if (gi == n) A.vout[(unsigned)((n + 1) * A.pitch + gj)] = ghost_h(c1, x);
            }
            if (per_j) {
                if (gj == n) A.vout[row] = x;
                if (gj == 1) A.vout[row + n + 1] = x;
            } else {
                if (gj == 1) A.vout[row] = ghost_h(c2, x);
                if (gj == n) A.vout[row + n + 1] = ghost_h(c3, x);
            }
        }
    }
    MGB_MARK(23);
}


// ---------------------------------------------------------------------------
// Coarse sub-V-cycle: every level with n <= 64 (v and f of 2^2 ... 64^2, 96 KB)
// lives in the LDS of ONE workgroup, which runs the whole recursion below the
// 64^2 level -- smoothing, residual, restriction, bottom solve, prolongation --
// without leaving the kernel.  These levels are pure launch latency otherwise
// (~550 us of ~100 dependent launches per V-cycle at 4096^2, measured).
// Arithmetic and operation order are those of the per-level kernels.
// ---------------------------------------------------------------------------
constexpr int MGC_TOP = 5;                       // level index of n = 64
constexpr int MGC_NT = 1024;
struct MGCoarse {
    double *v[MGC_TOP + 1], *f[MGC_TOP + 1], *r[MGC_TOP + 1];
    int pitch[MGC_TOP + 1];
    double dx[MGC_TOP + 1];
    int top;                                     // run levels top .. 0
    int finest;                                  // 1: `top` is the finest level (bc values apply)
    double alpha, beta;
    int nsmooth, nsmooth_bottom;
    MGBC bc;                                     // val[] only meaningful when finest
    unsigned zero_mask;                          // bit l: take v of level l as 0 (no memset before)
    int allow_pow2;                              // 0: PYRO_MG_NOPOW2 (see mg_pow2)
    int band64;                                  // 1: the 64^2 level's sweeps in registers (mgc_sweeps_band64)
    int wave_levels;                             // 1: the levels up to 32^2 on one wavefront (mgw_vcycle)
    long long *trace;                            // developer aid: clock64() at the phase marks
};
#ifdef PYRO_EMU
#define MGC_MARK(k) ((void)0)
#else
#define MGC_MARK(k) do { if (A.trace && tid == 0) A.trace[k] = clock64(); } while (0)
#endif
__host__ __device__ inline int mgc_off(int l)    // LDS offset (doubles) of level l's v
{
    int o = 0;
    for (int k = 0; k < l; k++) o += 2 * ((2 << k) + 2) * ((2 << k) + 2);
    return o;
}
constexpr int MGC_LDS_DOUBLES = 2 * (16 + 36 + 100 + 324 + 1156 + 4356);
constexpr int MGC_EDGE_DOUBLES = 32 * 2 * 64;     // mgc_sweeps_band64: first / last row of every half wavefront
constexpr size_t MGC_LDS = (size_t)(MGC_LDS_DOUBLES + MGC_EDGE_DOUBLES) * sizeof(double);

// Synchronisation inside the coarse kernel: workgroup barriers.  Running the smallest levels
// on wavefront 0 alone with wave-level barriers (within one wave LDS operations complete in
// program order) was measured twice and never paid: round 1 with the LDS sweeps (325 / 333 /
// 329 / 320 / 333 / 353 us on a 64^2 V-cycle for the levels up to none / 2^2 / 4^2 / 8^2 / 16^2 /
// 32^2), round 2 with sweeps of one wavefront in registers (a B x B block of cells per lane,
// neighbours by ds_bpermute, ghost values synthesised: 295 instead of 390 cycles per colour
// sweep on the 8^2 level, eaten up by the single-wave residual / restriction / prolongation
// phases: 129 us against 125 us).  A colour sweep of a tiny level is an LDS-latency round trip
// (~150 cycles) plus ~60 dependent-issue-bound instructions either way.  The code is gone: its
// second instantiation of every level function made the kernel 78 KB, more than the 64 KB
// instruction cache holds.
template <int NT>
__device__ __forceinline__ void mgc_sync()
{
    static_assert(NT == MGC_NT, "the coarse kernel runs every level with the whole workgroup");
    __syncthreads();
}

template <int NT>
__device__ inline void mgc_fill(double *V, int n, double dx, const MGBC &bc, bool use_val, int tid)
{
    const int q = n + 2;
    // x sides over all j, then y sides over all i (corners from x-filled data)
    for (int j = tid; j < q; j += NT) {
        const double *v0 = use_val ? bc.val[0] : nullptr, *v1 = use_val ? bc.val[1] : nullptr;
        V[j] = (bc.code[0] == PYROHIP_BC_PERIODIC) ? V[n * q + j]
                                                    : ghost_lo(bc.code[0], V[q + j], v0, j, dx);
        V[(n + 1) * q + j] = (bc.code[1] == PYROHIP_BC_PERIODIC)
                                 ? V[q + j]
                                 : ghost_hi(bc.code[1], V[n * q + j], v1, j, dx);
    }
    mgc_sync<NT>();
    for (int i = tid; i < q; i += NT) {
        const double *v2 = use_val ? bc.val[2] : nullptr, *v3 = use_val ? bc.val[3] : nullptr;
        double *row = V + i * q;
        row[0] = (bc.code[2] == PYROHIP_BC_PERIODIC) ? row[n] : ghost_lo(bc.code[2], row[1], v2, i, dx);
        row[n + 1] = (bc.code[3] == PYROHIP_BC_PERIODIC) ? row[1]
                                                          : ghost_hi(bc.code[3], row[n], v3, i, dx);
    }
    mgc_sync<NT>();
}

// colour sweeps of mgc_smooth (below); HOM: homogeneous boundaries, branch-free
template <int NT, bool HOM>
__device__ __forceinline__ void mgc_sweeps(double *V, const double *F, int n, int lg, double dx,
                                           double xc, double yc, double denom, double rdenom,
                                           int iters, int c0, int c1, int c2, int c3,
                                           const double *v0, const double *v1, const double *v2,
                                           const double *v3, int tid)
{
    const int q = n + 2, half = n >> 1;
    const bool p0 = (c0 == PYROHIP_BC_PERIODIC), p1 = (c1 == PYROHIP_BC_PERIODIC);
    const bool p2 = (c2 == PYROHIP_BC_PERIODIC), p3 = (c3 == PYROHIP_BC_PERIODIC);
    for (int it = 0; it < 2 * iters; it++) {
        const int colour = it & 1;
        for (int idx = tid; idx < n * half; idx += NT) {
            const int ri = (lg > 1) ? (idx >> (lg - 1)) : idx, h = idx - ri * half;
            const int i = 1 + ri;
            const int j = 1 + 2 * h + ((ri + colour) & 1);
            const int c = i * q + j;
            const double vn = div_by(F[c] + xc * (V[c + q] + V[c - q]) + yc * (V[c + 1] + V[c - 1]),
                                     denom, rdenom);
            V[c] = vn;
            if (HOM) {
                // four unconditional stores with selected target and value (the
                // cell itself again when it is not on that side): no branches
                const bool a0 = (i == 1), a1 = (i == n), a2 = (j == 1), a3 = (j == n);
                V[a0 ? (p0 ? (n + 1) * q + j : j) : c] = a0 ? (p0 ? vn : ghost_h(c0, vn)) : vn;
                V[a1 ? (p1 ? j : (n + 1) * q + j) : c] = a1 ? (p1 ? vn : ghost_h(c1, vn)) : vn;
                V[a2 ? (p2 ? i * q + n + 1 : i * q) : c] = a2 ? (p2 ? vn : ghost_h(c2, vn)) : vn;
                V[a3 ? (p3 ? i * q : i * q + n + 1) : c] = a3 ? (p3 ? vn : ghost_h(c3, vn)) : vn;
            } else {
                if (i == 1) {
                    if (p0) V[(n + 1) * q + j] = vn;
                    else V[j] = ghost_lo(c0, vn, v0, j, dx);
                }
                if (i == n) {
                    if (p1) V[j] = vn;
                    else V[(n + 1) * q + j] = ghost_hi(c1, vn, v1, j, dx);
                }
                if (j == 1) {
                    if (p2) V[i * q + n + 1] = vn;
                    else V[i * q] = ghost_lo(c2, vn, v2, i, dx);
                }
                if (j == n) {
                    if (p3) V[i * q] = vn;
                    else V[i * q + n + 1] = ghost_hi(c3, vn, v3, i, dx);
                }
            }
        }
        mgc_sync<NT>();
    }
}

// The same sweeps with everything that does not change from sweep to sweep taken out
// of the loop: a thread keeps the (at most two) cells it relaxes per colour for the
// whole smoothing -- LDS index, and for cells next to a boundary the index and the
// rule of the ghost cell(s) that mirror them (-1: none) -- so that a sweep is five
// LDS reads, the update, one write and, for boundary cells only, the ghost writes.
// (mgc_sweeps re-derives cell and ghost indices in every sweep and issues four
// predicated-by-select ghost stores per cell: ~60 instructions around a 9-operation
// update.)  Homogeneous boundaries; n * n / 2 <= 2 NT.
template <int NT, bool POW2>
__device__ __forceinline__ void mgc_sweeps_lean(double *V, const double *F, int n, int lg,
                                                double xc, double yc, double denom,
                                                double rdenom, int iters, int c0, int c1, int c2,
                                                int c3, int tid)
{
    const int q = n + 2, half = n >> 1, ncell = n * half;
    const bool p0 = (c0 == PYROHIP_BC_PERIODIC), p1 = (c1 == PYROHIP_BC_PERIODIC);
    const bool p2 = (c2 == PYROHIP_BC_PERIODIC), p3 = (c3 == PYROHIP_BC_PERIODIC);
    const double kx = xc * rdenom, ky = yc * rdenom;
    int cell[2][2], gi_t[2][2], gj_t[2][2], gi_c[2][2], gj_c[2][2];
#pragma unroll
    for (int colour = 0; colour < 2; colour++)
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int idx = tid + k * NT;
            int c = -1, ti = -1, tj = -1, ci = 0, cj = 0;
            if (idx < ncell) {
                const int ri = (lg > 1) ? (idx >> (lg - 1)) : idx, h = idx - ri * half;
                const int i = 1 + ri, j = 1 + 2 * h + ((ri + colour) & 1);
                c = i * q + j;
                // ghost cell mirrored across an x side / a y side (n >= 2: at most one each)
                if (i == 1) { ti = p0 ? (n + 1) * q + j : j; ci = p0 ? -1 : c0; }
                if (i == n) { ti = p1 ? j : (n + 1) * q + j; ci = p1 ? -1 : c1; }
                if (j == 1) { tj = p2 ? i * q + n + 1 : i * q; cj = p2 ? -1 : c2; }
                if (j == n) { tj = p3 ? i * q : i * q + n + 1; cj = p3 ? -1 : c3; }
            }
            cell[colour][k] = c; gi_t[colour][k] = ti; gj_t[colour][k] = tj;
            gi_c[colour][k] = ci; gj_c[colour][k] = cj;
        }
    auto relax = [&](int c, int ti, int tj, int ci, int cj) __attribute__((always_inline)) {
        if (c < 0) return;
        double vn;
        if (POW2)
            vn = fma(ky, V[c + 1] + V[c - 1], fma(kx, V[c + q] + V[c - q], F[c] * rdenom));
        else
            vn = div_by(F[c] + xc * (V[c + q] + V[c - q]) + yc * (V[c + 1] + V[c - 1]), denom, rdenom);
        V[c] = vn;
        if (ti >= 0) V[ti] = (ci < 0) ? vn : ghost_h(ci, vn);
        if (tj >= 0) V[tj] = (cj < 0) ? vn : ghost_h(cj, vn);
    };
    for (int it = 0; it < iters; it++) {
        relax(cell[0][0], gi_t[0][0], gj_t[0][0], gi_c[0][0], gj_c[0][0]);
        relax(cell[0][1], gi_t[0][1], gj_t[0][1], gi_c[0][1], gj_c[0][1]);
        mgc_sync<NT>();
        relax(cell[1][0], gi_t[1][0], gj_t[1][0], gi_c[1][0], gj_c[1][0]);
        relax(cell[1][1], gi_t[1][1], gj_t[1][1], gi_c[1][1], gj_c[1][1]);
        mgc_sync<NT>();
    }
}

// The sweeps of the 64^2 level in the layout of k_mg_smooth_band: rows in registers for the
// whole smoothing, lane h of a half wavefront the columns 2h+1, 2h+2, column neighbours by a
// whole-wave DPP rotation, the edge rows through LDS (E), ghost values as +-(own value) in one
// fma (mirror sides; a side with value-0 ghosts takes mgc_sweeps_lean).
#if !defined(PYRO_EMU)
__device__ __forceinline__ double mgc_rot_from_lower(double v) { return mgb_from_lower(v); }   // lane 0 <- 63
__device__ __forceinline__ double mgc_rot_from_upper(double v) { return mgb_from_upper(v); }
#else
__device__ __forceinline__ double mgc_rot_from_lower(double v) { return __shfl(v, (int)((threadIdx.x + 63) & 63), 64); }
__device__ __forceinline__ double mgc_rot_from_upper(double v) { return __shfl(v, (int)((threadIdx.x + 1) & 63), 64); }
#endif
// the value lane L (a constant) holds, in every lane
__device__ __forceinline__ double mgc_lane_value(double v, int L)
{
#if defined(PYRO_EMU)
    return __shfl(v, L, 64);
#else
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), L);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), L);
    return __hiloint2double(hi, lo);
#endif
}

template <bool POW2>
__device__ __forceinline__ void mgc_sweeps_band64(double *V, const double *F, double *E, double xc,
                                                  double yc, double denom, double rdenom, int iters,
                                                  int c0, int c1, int c2, int c3, int tid)
{
    // Round 3: a wavefront's two halves (lanes 0 .. 31, 32 .. 63) hold different rows -- no
    // lane computes a copy: slab sl = 2 w + half keeps rows 2 sl + 1, 2 sl + 2, lane h of the
    // half the columns 2h + 1, 2h + 2.  The level's 2048 updates per colour are 32 per SIMD
    // lane instead of 64 (the kernel's one CU is VALU bound on this level: sixteen wavefronts
    // on four SIMDs, 1600 cycles per colour sweep before).  The whole-wave rotation hands lane
    // 0 / 32 the other half's last column and lane 31 / 63 its first one: never read next to
    // a mirror side (the ghost value is taken instead), read back from the two lanes that got it
    // on periodic ones.  Every row is a slab's first or last one: E holds the level, a row
    // per (slab, first / last).
    constexpr int N = 64, Q = N + 2, R = 2, NS = 32;
    const int wv = tid >> 6, ln = tid & 63, hl = ln & 31, sl = 2 * wv + (ln >> 5);
    const bool per_i = (c0 == PYROHIP_BC_PERIODIC), per_j = (c2 == PYROHIP_BC_PERIODIC);
    const bool botW = (sl == 0) && !per_i, topW = (sl == NS - 1) && !per_i;   // rows 1 / 64
    const bool isWl = (hl == 0) && !per_j, isEl = (hl == 31) && !per_j;       // columns 1 / 64
    const double sBw = (botW && c0 == PYROHIP_BC_REFLECT_ODD) ? -1.0 : 1.0;
    const double sTw = (topW && c1 == PYROHIP_BC_REFLECT_ODD) ? -1.0 : 1.0;
    const double sWl = (isWl && c2 == PYROHIP_BC_REFLECT_ODD) ? -1.0 : 1.0;
    const double sEl = (isEl && c3 == PYROHIP_BC_REFLECT_ODD) ? -1.0 : 1.0;
    const double kx = xc * rdenom, ky = yc * rdenom;
    const int sb = (sl + NS - 1) & (NS - 1), sa = (sl + 1) & (NS - 1);
    auto ei = [](int s_, int which, int col) -> int { return (s_ * 2 + which) * N + col; };
    double v[R][2], fs[R][2];
#pragma unroll
    for (int m = 0; m < R; m++)
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int c = (R * sl + m + 1) * Q + 2 * hl + 1 + q;
            v[m][q] = V[c];
            fs[m][q] = POW2 ? F[c] * rdenom : F[c];
        }
    E[ei(sl, 0, 2 * hl)] = v[0][0]; E[ei(sl, 0, 2 * hl + 1)] = v[0][1];
    E[ei(sl, 1, 2 * hl)] = v[1][0]; E[ei(sl, 1, 2 * hl + 1)] = v[1][1];
    __syncthreads();
    // cell (i, j) = (2 sl + m + 1, 2h + 1 + q) is relaxed in sweep s iff i + j + s is even:
    // the thread's column q = (PAR + m) & 1 with PAR = s & 1
    auto pass = [&](auto par_c) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_c)::value;
        const double below = E[ei(sb, 1, 2 * hl + (PAR & 1))];
        const double above = E[ei(sa, 0, 2 * hl + ((PAR + 1) & 1))];
#pragma unroll
        for (int m = 0; m < R; m++) {
            const int q = (PAR + m) & 1;
            const double me = v[m][q];
            const double up = (m == 0) ? v[1][q] : above;
            const double dn = (m == 0) ? below : v[0][q];
            const double own = v[m][q ^ 1];
            double oth = q ? mgc_rot_from_upper(v[m][0]) : mgc_rot_from_lower(v[m][1]);
            if (per_j) {                      // the level's first / last column: the other half got it
                // (two lanes read, two selects: no trip through the LDS crossbar in the chain)
                const double x0 = mgc_lane_value(oth, q ? 31 : 0), x1 = mgc_lane_value(oth, q ? 63 : 32);
                oth = (ln == (q ? 31 : 0)) ? x1 : (ln == (q ? 63 : 32)) ? x0 : oth;
            }
            const double e = q ? oth : own, w = q ? own : oth;
            double si, sj;
            if (m == 0) si = fma(botW ? me : dn, sBw, up);
            else si = fma(topW ? me : up, sTw, dn);
            if (q == 0) sj = fma(isWl ? me : w, sWl, e);
            else sj = fma(isEl ? me : e, sEl, w);
            if (POW2)
                v[m][q] = fma(ky, sj, fma(kx, si, fs[m][q]));
            else
                v[m][q] = div_by(fs[m][q] + xc * si + yc * sj, denom, rdenom);
        }
        E[ei(sl, 0, 2 * hl + (PAR & 1))] = v[0][PAR & 1];
        E[ei(sl, 1, 2 * hl + ((PAR + 1) & 1))] = v[1][(PAR + 1) & 1];
    };
    for (int s = 0; s < 2 * iters; s++) {
        if (s & 1) pass(std::integral_constant<int, 1>{});
        else pass(std::integral_constant<int, 0>{});
        __syncthreads();
    }
#pragma unroll
    for (int m = 0; m < R; m++)
#pragma unroll
        for (int q = 0; q < 2; q++) V[(R * sl + m + 1) * Q + 2 * hl + 1 + q] = v[m][q];
    __syncthreads();
}

// Red-black sweeps of one LDS-resident level.  The thread that updates a cell
// next to a boundary also refreshes the ghost cell(s) that mirror it (like
// k_mg_smooth): during a colour sweep a ghost cell is only read by the cell it
// mirrors (or, on periodic sides, by a cell of the other colour), so this is
// the reference's fill_BC after the sweep (MG.py:598-599) without a separate
// pass and its two extra synchronisations.  Corner ghosts are not read by the
// 5-point stencil; the closing fill makes them exact again.
// Measured (tools/mgc_probe.py, gpurun_out/mgc_probe*.log): a colour sweep
// costs ~0.6 us on the 2 x 2 level and ~0.8 us averaged over the levels,
// whether ghosts are refreshed here or in two extra passes and whether one
// wave or the workgroup runs it -- the kernel (~250 us of the V-cycle, 300
// sweeps at nsmooth 10 / bottom 50) is bound by the dependent chain LDS read
// -> 8 fp64 operations -> LDS write -> barrier of each sweep, not by the
// number of barriers.
__device__ __forceinline__ bool mgc_is_pow2(double x)     // device twin of mg_is_pow2
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    const unsigned e = (unsigned)(b >> 52) & 0x7ffu;
    return (b & 0x000fffffffffffffull) == 0 && e != 0 && e != 0x7ffu;
}

template <int NT>
__device__ inline void mgc_smooth(double *V, const double *F, int n, int lg, double dx,
                                  double alpha, double beta, int iters, const MGBC &bc,
                                  bool use_val, int tid, bool allow_pow2 = true, double *E = nullptr)
{
    const int q = n + 2;
    const double xc = beta / (dx * dx), yc = beta / (dx * dx);
    const double denom = alpha + 2.0 * xc + 2.0 * yc;
    const double rdenom = 1.0 / denom;                          // correctly rounded: div_by
    const int c0 = bc.code[0], c1 = bc.code[1], c2 = bc.code[2], c3 = bc.code[3];
    const double *v0 = use_val ? bc.val[0] : nullptr, *v1 = use_val ? bc.val[1] : nullptr;
    const double *v2 = use_val ? bc.val[2] : nullptr, *v3 = use_val ? bc.val[3] : nullptr;
    const bool mirror = c0 != PYROHIP_BC_CONST && c1 != PYROHIP_BC_CONST && c2 != PYROHIP_BC_CONST &&
                        c3 != PYROHIP_BC_CONST;
    // (the sweeps in registers read no ghost cell: no fill before them)
    const bool band = NT == MGC_NT && n == 64 && E && mirror && !(v0 || v1 || v2 || v3) && iters > 0;
    if (!band) mgc_fill<NT>(V, n, dx, bc, use_val, tid);        // MG.py:565
    if (iters <= 0) return;
    const bool pow2 = allow_pow2 && mgc_is_pow2(xc) && mgc_is_pow2(yc) && mgc_is_pow2(denom) &&
                      rdenom * denom == 1.0;                    // see mg_pow2
    if (band) {
        if (pow2) mgc_sweeps_band64<true>(V, F, E, xc, yc, denom, rdenom, iters, c0, c1, c2, c3, tid);
        else mgc_sweeps_band64<false>(V, F, E, xc, yc, denom, rdenom, iters, c0, c1, c2, c3, tid);
    } else if (!(v0 || v1 || v2 || v3) && n * (n >> 1) <= 2 * NT) {
        if (pow2) mgc_sweeps_lean<NT, true>(V, F, n, lg, xc, yc, denom, rdenom, iters, c0, c1, c2, c3, tid);
        else mgc_sweeps_lean<NT, false>(V, F, n, lg, xc, yc, denom, rdenom, iters, c0, c1, c2, c3, tid);
    } else if (!(v0 || v1 || v2 || v3))   // boundary values only on a finest level <= 64^2
        mgc_sweeps<NT, true>(V, F, n, lg, dx, xc, yc, denom, rdenom, iters, c0, c1, c2, c3, v0, v1,
                             v2, v3, tid);
    else
        mgc_sweeps<NT, false>(V, F, n, lg, dx, xc, yc, denom, rdenom, iters, c0, c1, c2, c3, v0, v1,
                              v2, v3, tid);
    mgc_fill<NT>(V, n, dx, bc, use_val, tid);                   // corners
}

// Bottom solve (the 2 x 2 level, nsmooth_bottom = 50 iterations = 100 colour
// sweeps, MG.py:776-778) by ONE thread with the 4 x 4 array in registers: the
// same updates in the same order as mgc_smooth (within a colour the two cells
// do not read anything the other one writes), without 100 LDS round trips.
__device__ inline void mgc_bottom_regs(double *V, const double *F, double dx, double alpha,
                                       double beta, int iters, const MGBC &bc, bool use_val)
{
    const double xc = beta / (dx * dx), yc = beta / (dx * dx);
    const double denom = alpha + 2.0 * xc + 2.0 * yc;
    const double rdenom = 1.0 / denom;
    const int c0 = bc.code[0], c1 = bc.code[1], c2 = bc.code[2], c3 = bc.code[3];
    const bool p0 = (c0 == PYROHIP_BC_PERIODIC), p1 = (c1 == PYROHIP_BC_PERIODIC);
    const bool p2 = (c2 == PYROHIP_BC_PERIODIC), p3 = (c3 == PYROHIP_BC_PERIODIC);
    const double *v0 = use_val ? bc.val[0] : nullptr, *v1 = use_val ? bc.val[1] : nullptr;
    const double *v2 = use_val ? bc.val[2] : nullptr, *v3 = use_val ? bc.val[3] : nullptr;
    const bool inhom = v0 || v1 || v2 || v3;   // boundary values: only if this is the finest level
    double v[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) v[i][j] = V[i * 4 + j];
    const double f11 = F[5], f12 = F[6], f21 = F[9], f22 = F[10];
    // cell (I, J) with literal indices; ghost refresh as in mgc_smooth (n = 2).
    // The boundary rules are selects, not branches: a taken branch costs more
    // than the dozen fp64 operations of the update (no branch prediction).
#define MGC_RELAX(I, J, FF, GLO, GHI)                                                        \
    {                                                                                        \
        const double vn = div_by(FF + xc * (v[I + 1][J] + v[I - 1][J]) +                     \
                                     yc * (v[I][J + 1] + v[I][J - 1]), denom, rdenom);        \
        v[I][J] = vn;                                                                        \
        if (I == 1) { v[3][J] = p0 ? vn : v[3][J]; v[0][J] = p0 ? v[0][J] : GLO(c0, vn, v0, J); } \
        if (I == 2) { v[0][J] = p1 ? vn : v[0][J]; v[3][J] = p1 ? v[3][J] : GHI(c1, vn, v1, J); } \
        if (J == 1) { v[I][3] = p2 ? vn : v[I][3]; v[I][0] = p2 ? v[I][0] : GLO(c2, vn, v2, I); } \
        if (J == 2) { v[I][0] = p3 ? vn : v[I][0]; v[I][3] = p3 ? v[I][3] : GHI(c3, vn, v3, I); } \
    }
#define MGC_GH(code, x, val, idx) ghost_h(code, x)
#define MGC_GLO(code, x, val, idx) ghost_lo(code, x, val, idx, dx)
#define MGC_GHI(code, x, val, idx) ghost_hi(code, x, val, idx, dx)
    if (!inhom)
        for (int it = 0; it < iters; it++) {
            MGC_RELAX(1, 1, f11, MGC_GH, MGC_GH) MGC_RELAX(2, 2, f22, MGC_GH, MGC_GH)   // colour 0
            MGC_RELAX(1, 2, f12, MGC_GH, MGC_GH) MGC_RELAX(2, 1, f21, MGC_GH, MGC_GH)   // colour 1
        }
    else
        for (int it = 0; it < iters; it++) {
            MGC_RELAX(1, 1, f11, MGC_GLO, MGC_GHI) MGC_RELAX(2, 2, f22, MGC_GLO, MGC_GHI)
            MGC_RELAX(1, 2, f12, MGC_GLO, MGC_GHI) MGC_RELAX(2, 1, f21, MGC_GLO, MGC_GHI)
        }
#undef MGC_RELAX
#undef MGC_GH
#undef MGC_GLO
#undef MGC_GHI
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) V[i * 4 + j] = v[i][j];
}

// The same bottom solve for homogeneous, non-periodic boundaries without the ghost
// cells: every cell of the 2 x 2 level has a ghost neighbour on two sides, and that
// ghost cell mirrors the cell itself -- ghost_h(own value) where the stencil asks
// for it (what the refresh after every colour stores there).  One wave issues a VALU
// instruction every ~5 cycles whatever the dependencies, so the ~220 instructions per
// iteration of the general form above cost ~0.4 us; here an iteration is 4 x (update +
// two ghost values).  The closing fill restores the stored ghost cells.
template <bool POW2>
__device__ inline void mgc_bottom_fast(double *V, const double *F, double dx, double alpha,
                                       double beta, int iters, const MGBC &bc)
{
    const double xc = beta / (dx * dx), yc = beta / (dx * dx);
    const double denom = alpha + 2.0 * xc + 2.0 * yc;
    const double rdenom = 1.0 / denom;
    const double kx = xc * rdenom, ky = yc * rdenom;
    const int c0 = bc.code[0], c1 = bc.code[1], c2 = bc.code[2], c3 = bc.code[3];
    double a = V[5], b = V[6], c = V[9], d = V[10];            // (1,1) (1,2) (2,1) (2,2)
    const double fa = POW2 ? F[5] * rdenom : F[5], fb = POW2 ? F[6] * rdenom : F[6];
    const double fc = POW2 ? F[9] * rdenom : F[9], fd = POW2 ? F[10] * rdenom : F[10];
    auto relax = [&](double f, double up, double dn, double e, double w) __attribute__((always_inline)) {
        return POW2 ? fma(ky, e + w, fma(kx, up + dn, f))
                    : div_by(f + xc * (up + dn) + yc * (e + w), denom, rdenom);
    };
    for (int it = 0; it < iters; it++) {
        a = relax(fa, c, ghost_h(c0, a), b, ghost_h(c2, a));   // colour 0: (1,1), (2,2)
        d = relax(fd, ghost_h(c1, d), b, ghost_h(c3, d), c);
        b = relax(fb, d, ghost_h(c0, b), ghost_h(c3, b), a);   // colour 1: (1,2), (2,1)
        c = relax(fc, ghost_h(c1, c), a, d, ghost_h(c2, c));
    }
    V[5] = a; V[6] = b; V[9] = c; V[10] = d;
}

// ---------------------------------------------------------------------------
// The levels up to 32^2 of the coarse V-cycle on ONE wavefront.
//
// A colour sweep of a tiny level by the whole workgroup is an LDS round trip and a
// 16-wavefront barrier around a dozen operations: ~850 cycles whatever the level's
// size (trace: 19-20 thousand cycles per level and leg, 136 of the kernel's 265
// thousand for the levels 2^2 ... 16^2).  Here wavefront 0 takes a level's cells into
// registers in the layout of the band / marching kernels -- a lane holds two
// neighbouring columns of up to eight rows; the column pairs of an N^2 level sit 32 / N
// lanes apart in a row of 16 lanes so that a DPP rotation of that row by 32 / N lanes
// wraps at the level's width (the lanes in between, and the other rows of 16, hold
// copies); the 16^2 level's upper eight rows are in the second row of lanes, the 32^2 level
// takes all four, eight rows each, and the rows of lanes trade their edge rows by a
// shuffle -- and smooths without LDS and
// without a barrier: no ghost cells, a cell next to a mirror boundary takes +-(its own
// value) (what the fill after every colour stores there), the rotation's wrap is the
// periodic neighbour.  Between levels the data goes through the levels' LDS arrays
// (lane-private reads and writes; the four fine cells under a coarse cell are one
// lane's): down a level = load, smooth, store, residual -> global r, restriction -> the
// coarser level's f; up = load, add the prolonged correction, smooth, store, fill the
// ghost cells (corners too: the arrays are written back to global as they are).
// Arithmetic and operation order are those of mgc_sweeps_lean / mgc_down / mgc_up.
// Mirror / periodic sides, levels below the kernel's top level (no boundary values).
// ---------------------------------------------------------------------------
template <int N> struct MGWave {
    static constexpr int R = N < 8 ? N : 8;   // rows a lane holds
    static constexpr int G = N / R;           // rows of 16 lanes the level's rows are spread over (1, 2, 4)
    static constexpr int S = 32 / N;          // lanes between neighbouring column pairs
    static constexpr int Q = N + 2;
};

__device__ __forceinline__ void mgw_sync()    // LDS written by one lane, read by another of the wavefront
{
#if defined(PYRO_EMU)
    hipemu::wave_barrier();
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}
// the value of the lane S lanes below / above in the row of 16 lanes, wrapping (S = 16: itself)
template <int S> __device__ __forceinline__ double mgw_from_lower(double v, int ln)
{
    if constexpr (S >= 16) return v;
#if defined(PYRO_EMU)
    else return __shfl(v, (ln & ~15) | ((ln - S) & 15), 64);
#else
    else return mgb_dpp<0x120 + S>(v);          // row_ror:S
#endif
}
template <int S> __device__ __forceinline__ double mgw_from_upper(double v, int ln)
{
    if constexpr (S >= 16) return v;
#if defined(PYRO_EMU)
    else return __shfl(v, (ln & ~15) | ((ln + S) & 15), 64);
#else
    else return mgb_dpp<0x120 + 16 - S>(v);     // row_ror:(16 - S)
#endif
}

struct MGWSide {        // what a lane is next to, and the mirror signs
    bool botW, topW, isW, isE;
    double sB, sT, sW, sE;
    int grp, h;         // row of lanes that holds the lane's part of the level, column pair
};
template <int N> __device__ __forceinline__ MGWSide mgw_side(const MGBC &bc, int ln)
{
    using W = MGWave<N>;
    MGWSide s;
    const bool per_i = (bc.code[0] == PYROHIP_BC_PERIODIC), per_j = (bc.code[2] == PYROHIP_BC_PERIODIC);
    s.grp = (ln >> 4) & (W::G - 1);
    s.h = (ln & 15) / W::S;
    s.botW = !per_i && s.grp == 0;
    s.topW = !per_i && s.grp == W::G - 1;
    s.isW = !per_j && s.h == 0;
    s.isE = !per_j && s.h == N / 2 - 1;
    s.sB = (s.botW && bc.code[0] == PYROHIP_BC_REFLECT_ODD) ? -1.0 : 1.0;
    s.sT = (s.topW && bc.code[1] == PYROHIP_BC_REFLECT_ODD) ? -1.0 : 1.0;
    s.sW = (s.isW && bc.code[2] == PYROHIP_BC_REFLECT_ODD) ? -1.0 : 1.0;
    s.sE = (s.isE && bc.code[3] == PYROHIP_BC_REFLECT_ODD) ? -1.0 : 1.0;
    return s;
}
// the rows beyond the lane's first / last one (the other half's edge rows at 16^2; the
// periodic image otherwise -- not read where the row is next to a mirror boundary)
template <int N> __device__ __forceinline__ double mgw_below(const double (&v)[MGWave<N>::R][2], int q, int ln)
{
    constexpr int G = MGWave<N>::G, R = MGWave<N>::R;
    if constexpr (G == 1) return v[R - 1][q];
    else if constexpr (G == 2) return __shfl_xor(v[R - 1][q], 16);
    else return __shfl(v[R - 1][q], (ln + 48) & 63, 64);      // the row of lanes below, wrapping
}
template <int N> __device__ __forceinline__ double mgw_above(const double (&v)[MGWave<N>::R][2], int q, int ln)
{
    constexpr int G = MGWave<N>::G;
    if constexpr (G == 1) return v[0][q];
    else if constexpr (G == 2) return __shfl_xor(v[0][q], 16);
    else return __shfl(v[0][q], (ln + 16) & 63, 64);
}

template <int N>
__device__ __forceinline__ void mgw_load(const double *V, double (&v)[MGWave<N>::R][2], const MGWSide &sd)
{
    using W = MGWave<N>;
#pragma unroll
    for (int r = 0; r < W::R; r++) {
        const int c = (1 + W::R * sd.grp + r) * W::Q + 1 + 2 * sd.h;
        v[r][0] = V[c]; v[r][1] = V[c + 1];
    }
}
template <int N>
__device__ __forceinline__ void mgw_store(double *V, const double (&v)[MGWave<N>::R][2], const MGWSide &sd, int ln)
{
    using W = MGWave<N>;
    if ((ln & 15) % W::S == 0 && (ln >> 4) < W::G) {
#pragma unroll
        for (int r = 0; r < W::R; r++) {
            const int c = (1 + W::R * sd.grp + r) * W::Q + 1 + 2 * sd.h;
            V[c] = v[r][0]; V[c + 1] = v[r][1];
        }
    }
    mgw_sync();
}
// mgc_fill by one wavefront (homogeneous): x sides over all j, then y sides over all i
__device__ inline void mgw_fill(double *V, int n, const MGBC &bc, int ln)
{
    const int q = n + 2;
    for (int j = ln; j < q; j += 64) {
        V[j] = (bc.code[0] == PYROHIP_BC_PERIODIC) ? V[n * q + j] : ghost_lo(bc.code[0], V[q + j], nullptr, j, 0.0);
        V[(n + 1) * q + j] = (bc.code[1] == PYROHIP_BC_PERIODIC) ? V[q + j] : ghost_hi(bc.code[1], V[n * q + j], nullptr, j, 0.0);
    }
    mgw_sync();
    for (int i = ln; i < q; i += 64) {
        double *row = V + i * q;
        row[0] = (bc.code[2] == PYROHIP_BC_PERIODIC) ? row[n] : ghost_lo(bc.code[2], row[1], nullptr, i, 0.0);
        row[n + 1] = (bc.code[3] == PYROHIP_BC_PERIODIC) ? row[1] : ghost_hi(bc.code[3], row[n], nullptr, i, 0.0);
    }
    mgw_sync();
}

// red-black sweeps on the registers; fs: f (POW2: f / denom)
template <int N, bool POW2>
__device__ __forceinline__ void mgw_sweeps(double (&v)[MGWave<N>::R][2], const double (&fs)[MGWave<N>::R][2],
                                           double xc, double yc, double denom, double rdenom, int iters,
                                           const MGWSide &sd, int ln)
{
    using W = MGWave<N>;
    constexpr int R = W::R, S = W::S;
    const double kx = xc * rdenom, ky = yc * rdenom;
    // cell (i, j) is relaxed in colour c iff j - 1 = (i - 1 + c) mod 2 (mgc_sweeps): in the
    // lane's row r (the rows before it are an even number) its column q = (r + c) & 1
    auto pass = [&](auto cc) __attribute__((always_inline)) {
        constexpr int C = decltype(cc)::value;
        const double below = mgw_below<N>(v, C & 1, ln);             // for row 0, column (0 + C) & 1
        const double above = mgw_above<N>(v, (R - 1 + C) & 1, ln);
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int q = (r + C) & 1;
            const double me = v[r][q], own = v[r][q ^ 1];
            const double oth = q ? mgw_from_upper<S>(v[r][0], ln) : mgw_from_lower<S>(v[r][1], ln);
            const double e = q ? oth : own, w = q ? own : oth;
            const double up = (r < R - 1) ? v[r < R - 1 ? r + 1 : r][q] : above;
            const double dn = (r > 0) ? v[r > 0 ? r - 1 : 0][q] : below;
            double si, sj;
            if (R == 1) si = up + dn;
            else if (r == 0) si = fma(sd.botW ? me : dn, sd.sB, up);
            else if (r == R - 1) si = fma(sd.topW ? me : up, sd.sT, dn);
            else si = up + dn;
            if (q == 0) sj = fma(sd.isW ? me : w, sd.sW, e);
            else sj = fma(sd.isE ? me : e, sd.sE, w);
            if (POW2) v[r][q] = fma(ky, sj, fma(kx, si, fs[r][q]));
            else v[r][q] = div_by(fs[r][q] + xc * si + yc * sj, denom, rdenom);
        }
    };
    for (int it = 0; it < iters; it++) {
        pass(std::integral_constant<int, 0>{});
        pass(std::integral_constant<int, 1>{});
    }
}

struct MGWCoef { double xc, yc, denom, rdenom; bool pow2; };
__device__ __forceinline__ MGWCoef mgw_coef(const MGCoarse &A, int l)
{
    MGWCoef c;
    c.xc = A.beta / (A.dx[l] * A.dx[l]); c.yc = c.xc;
    c.denom = A.alpha + 2.0 * c.xc + 2.0 * c.yc;
    c.rdenom = 1.0 / c.denom;
    c.pow2 = A.allow_pow2 && mgc_is_pow2(c.xc) && mgc_is_pow2(c.yc) && mgc_is_pow2(c.denom) &&
             c.rdenom * c.denom == 1.0;                  // mgc_smooth
    return c;
}

// level l = log2(N) - 1 on the way down (MG.py:722-735)
template <int N, bool POW2>
__device__ inline void mgw_down_t(const MGCoarse &A, double *lds, const MGWCoef &cf, int ln)
{
    using W = MGWave<N>;
    constexpr int R = W::R, S = W::S, Q = W::Q, l = (N == 32) ? 4 : (N == 16) ? 3 : (N == 8) ? 2 : 1, QC = N / 2 + 2;
    double *V = lds + mgc_off(l), *F = V + Q * Q, *Fc = lds + mgc_off(l - 1) + QC * QC;
    const MGWSide sd = mgw_side<N>(A.bc, ln);
    double v[R][2], fs[R][2];
    mgw_load<N>(V, v, sd);
    mgw_load<N>(F, fs, sd);
    if (POW2) {
#pragma unroll
        for (int r = 0; r < R; r++) { fs[r][0] *= cf.rdenom; fs[r][1] *= cf.rdenom; }
    }
    mgw_sweeps<N, POW2>(v, fs, cf.xc, cf.yc, cf.denom, cf.rdenom, A.nsmooth, sd, ln);
    mgw_store<N>(V, v, sd, ln);
    // residual (mgc_down's expression; the divisions by dx^2 as div_by: the same bits), its
    // restriction: the lane's rows 2m, 2m + 1 and its two columns are one coarse cell
    const double dx2 = A.dx[l] * A.dx[l], rdx2 = 1.0 / dx2;
    const bool primary = (ln & 15) % S == 0 && (ln >> 4) < W::G;
    const double below0 = mgw_below<N>(v, 0, ln), below1 = mgw_below<N>(v, 1, ln);
    const double above0 = mgw_above<N>(v, 0, ln), above1 = mgw_above<N>(v, 1, ln);
#pragma unroll
    for (int m = 0; m < R / 2; m++) {
        double rr[2][2];
#pragma unroll
        for (int rl = 0; rl < 2; rl++) {
            const int r = 2 * m + rl;
            const double fromW = mgw_from_lower<S>(v[r][1], ln), fromE = mgw_from_upper<S>(v[r][0], ln);
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const double me = v[r][q];
                double dn = (r > 0) ? v[r > 0 ? r - 1 : 0][q] : (q ? below1 : below0);
                double up = (r < R - 1) ? v[r < R - 1 ? r + 1 : r][q] : (q ? above1 : above0);
                if (r == 0 && sd.botW) dn = sd.sB * me;
                if (r == R - 1 && sd.topW) up = sd.sT * me;
                double w = q ? v[r][0] : fromW, e = q ? fromE : v[r][1];
                if (q == 0 && sd.isW) w = sd.sW * me;
                if (q == 1 && sd.isE) e = sd.sE * me;
                const double f = POW2 ? fs[r][q] * cf.denom : fs[r][q];
                rr[rl][q] = f - A.alpha * me +
                            A.beta * (div_by(dn + up - 2 * me, dx2, rdx2) + div_by(w + e - 2 * me, dx2, rdx2));
            }
        }
        if (primary) {
#pragma unroll
            for (int rl = 0; rl < 2; rl++) {
                const size_t g = (size_t)(1 + R * sd.grp + 2 * m + rl) * A.pitch[l] + 1 + 2 * sd.h;
                A.r[l][g] = rr[rl][0]; A.r[l][g + 1] = rr[rl][1];
            }
            // patch.py:660-662: (i,j) + (i+1,j) + (i,j+1) + (i+1,j+1)
            Fc[(1 + (R / 2) * sd.grp + m) * QC + 1 + sd.h] = 0.25 * (rr[0][0] + rr[1][0] + rr[0][1] + rr[1][1]);
        }
    }
    mgw_sync();
}

// ... on the way up (MG.py:745-758): the coarser level's array has its ghost cells
template <int N, bool POW2>
__device__ inline void mgw_up_t(const MGCoarse &A, double *lds, const MGWCoef &cf, int ln)
{
    using W = MGWave<N>;
    constexpr int R = W::R, Q = W::Q, l = (N == 32) ? 4 : (N == 16) ? 3 : (N == 8) ? 2 : 1, QC = N / 2 + 2;
    double *V = lds + mgc_off(l), *F = V + Q * Q;
    const double *Vc = lds + mgc_off(l - 1);
    const MGWSide sd = mgw_side<N>(A.bc, ln);
    double v[R][2], fs[R][2];
    mgw_load<N>(V, v, sd);
    mgw_load<N>(F, fs, sd);
#pragma unroll
    for (int m = 0; m < R / 2; m++) {        // the coarse cell under the lane's rows 2m, 2m + 1
        const int ck = (1 + (R / 2) * sd.grp + m) * QC + 1 + sd.h;
        const double c0 = Vc[ck];
        const double m_x = 0.5 * (Vc[ck + QC] - Vc[ck - QC]);
        const double m_y = 0.5 * (Vc[ck + 1] - Vc[ck - 1]);
        v[2 * m][0] += c0 - 0.25 * m_x - 0.25 * m_y;        // mgc_up: fi even / odd, fj even / odd
        v[2 * m][1] += c0 - 0.25 * m_x + 0.25 * m_y;
        v[2 * m + 1][0] += c0 + 0.25 * m_x - 0.25 * m_y;
        v[2 * m + 1][1] += c0 + 0.25 * m_x + 0.25 * m_y;
    }
    if (POW2) {
#pragma unroll
        for (int r = 0; r < R; r++) { fs[r][0] *= cf.rdenom; fs[r][1] *= cf.rdenom; }
    }
    mgw_sweeps<N, POW2>(v, fs, cf.xc, cf.yc, cf.denom, cf.rdenom, A.nsmooth, sd, ln);
    mgw_store<N>(V, v, sd, ln);
    mgw_fill(V, N, A.bc, ln);
}

// the 2^2 level (MG.py:776-778)
template <bool POW2>
__device__ inline void mgw_bottom_t(const MGCoarse &A, double *lds, const MGWCoef &cf, int ln)
{
    constexpr int N = 2;
    double *V = lds + mgc_off(0), *F = V + 16;
    const MGWSide sd = mgw_side<N>(A.bc, ln);
    double v[2][2], fs[2][2];
    mgw_load<N>(V, v, sd);
    mgw_load<N>(F, fs, sd);
    if (POW2) { fs[0][0] *= cf.rdenom; fs[0][1] *= cf.rdenom; fs[1][0] *= cf.rdenom; fs[1][1] *= cf.rdenom; }
    mgw_sweeps<N, POW2>(v, fs, cf.xc, cf.yc, cf.denom, cf.rdenom, A.nsmooth_bottom, sd, ln);
    mgw_store<N>(V, v, sd, ln);
    mgw_fill(V, N, A.bc, ln);
}

template <int N> __device__ __forceinline__ void mgw_down(const MGCoarse &A, double *lds, int ln)
{
    const MGWCoef cf = mgw_coef(A, (N == 32) ? 4 : (N == 16) ? 3 : (N == 8) ? 2 : 1);
    if (cf.pow2) mgw_down_t<N, true>(A, lds, cf, ln);
    else mgw_down_t<N, false>(A, lds, cf, ln);
}
template <int N> __device__ __forceinline__ void mgw_up(const MGCoarse &A, double *lds, int ln)
{
    const MGWCoef cf = mgw_coef(A, (N == 32) ? 4 : (N == 16) ? 3 : (N == 8) ? 2 : 1);
    if (cf.pow2) mgw_up_t<N, true>(A, lds, cf, ln);
    else mgw_up_t<N, false>(A, lds, cf, ln);
}

// the sub-V-cycle below level `from` + 1 (from <= 3): wavefront 0 of the workgroup
__device__ inline void mgw_vcycle(const MGCoarse &A, double *lds, int from, int ln)
{
    const int tid = ln;
    if (from >= 4) { mgw_down<32>(A, lds, ln); MGC_MARK(2 + (A.top - 4)); }
    if (from >= 3) { mgw_down<16>(A, lds, ln); MGC_MARK(2 + (A.top - 3)); }
    if (from >= 2) { mgw_down<8>(A, lds, ln); MGC_MARK(2 + (A.top - 2)); }
    if (from >= 1) { mgw_down<4>(A, lds, ln); MGC_MARK(2 + (A.top - 1)); }
    {
        const MGWCoef cf = mgw_coef(A, 0);
        if (cf.pow2) mgw_bottom_t<true>(A, lds, cf, ln);
        else mgw_bottom_t<false>(A, lds, cf, ln);
    }
    MGC_MARK(8);
    if (from >= 1) { mgw_up<4>(A, lds, ln); MGC_MARK(9); }
    if (from >= 2) { mgw_up<8>(A, lds, ln); MGC_MARK(10); }
    if (from >= 3) { mgw_up<16>(A, lds, ln); MGC_MARK(11); }
    if (from >= 4) { mgw_up<32>(A, lds, ln); MGC_MARK(12); }
}

// one level of the down leg (MG.py:722-735): smooth, residual -> global r,
// its restriction -> f of the next coarser level
template <int NT>
__device__ inline void mgc_down(const MGCoarse &A, int l, double *lds, int tid)
{
    const int n = 2 << l, q = n + 2, nc = n >> 1, qc = nc + 2;
    double *V = lds + mgc_off(l), *F = V + q * q;
    double *Fc = lds + mgc_off(l - 1) + qc * qc;
    const bool uv = A.finest && l == A.top;
    if (l == A.top) MGC_MARK(14);
    mgc_smooth<NT>(V, F, n, l + 1, A.dx[l], A.alpha, A.beta, A.nsmooth, A.bc, uv, tid, A.allow_pow2,
                   A.band64 ? lds + MGC_LDS_DOUBLES : nullptr);
    if (l == A.top) MGC_MARK(15);
    const double dx2 = A.dx[l] * A.dx[l];
    for (int idx = tid; idx < nc * nc; idx += NT) {
        const int ci = idx / nc, cj = idx - ci * nc;
        double rr[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = 1 + 2 * ci + (k & 1), j = 1 + 2 * cj + (k >> 1);
            const int c = i * q + j;
            rr[k] = F[c] - A.alpha * V[c] +
                    A.beta * ((V[c - q] + V[c + q] - 2 * V[c]) / dx2 +
                              (V[c - 1] + V[c + 1] - 2 * V[c]) / dx2);
            A.r[l][(size_t)i * A.pitch[l] + j] = rr[k];
        }
        // patch.py:660-662: (i,j) + (i+1,j) + (i,j+1) + (i+1,j+1)
        Fc[(1 + ci) * qc + (1 + cj)] = 0.25 * (rr[0] + rr[1] + rr[2] + rr[3]);
    }
    mgc_sync<NT>();
}

// one level of the up leg (MG.py:745-758): prolong the coarse correction, smooth
template <int NT>
__device__ inline void mgc_up(const MGCoarse &A, int l, double *lds, int tid)
{
    const int n = 2 << l, q = n + 2, nc = n >> 1, qc = nc + 2;
    double *V = lds + mgc_off(l), *F = V + q * q;
    const double *Vc = lds + mgc_off(l - 1);
    for (int idx = tid; idx < n * n; idx += NT) {
        const int fi = idx / n, fj = idx - fi * n;
        const int ck = (1 + (fi >> 1)) * qc + 1 + (fj >> 1);
        const double c0 = Vc[ck];
        const double m_x = 0.5 * (Vc[ck + qc] - Vc[ck - qc]);
        const double m_y = 0.5 * (Vc[ck + 1] - Vc[ck - 1]);
        double e;
        if (fi & 1) e = (fj & 1) ? c0 + 0.25 * m_x + 0.25 * m_y : c0 + 0.25 * m_x - 0.25 * m_y;
        else        e = (fj & 1) ? c0 - 0.25 * m_x + 0.25 * m_y : c0 - 0.25 * m_x - 0.25 * m_y;
        V[(1 + fi) * q + (1 + fj)] += e;
    }
    mgc_sync<NT>();
    mgc_smooth<NT>(V, F, n, l + 1, A.dx[l], A.alpha, A.beta, A.nsmooth, A.bc,
                   A.finest && l == A.top, tid, A.allow_pow2, A.band64 ? lds + MGC_LDS_DOUBLES : nullptr);
}

__global__ __launch_bounds__(MGC_NT) void k_mg_coarse_vcycle(MGCoarse A)
{
    HIP_DYNAMIC_SHARED(double, lds)
    const int tid = threadIdx.x;
    MGC_MARK(0);
    // stage v and f of every level
    for (int l = 0; l <= A.top; l++) {
        const int n = 2 << l, q = n + 2;
        double *V = lds + mgc_off(l), *F = V + q * q;
        const bool vz = (A.zero_mask >> l) & 1u;
        for (int idx = tid; idx < q * q; idx += MGC_NT) {
            const int i = idx / q, j = idx - i * q;
            V[idx] = vz ? 0.0 : A.v[l][(size_t)i * A.pitch[l] + j];
            F[idx] = A.f[l][(size_t)i * A.pitch[l] + j];
        }
    }
    __syncthreads();
    MGC_MARK(1);
    // the levels up to 32^2 (below the top level) on wavefront 0: mgw_vcycle
    const bool mirror_bc = A.bc.code[0] != PYROHIP_BC_CONST && A.bc.code[1] != PYROHIP_BC_CONST &&
                           A.bc.code[2] != PYROHIP_BC_CONST && A.bc.code[3] != PYROHIP_BC_CONST;
    const int wave_from = (A.wave_levels && mirror_bc) ? (A.top - 1 < 4 ? A.top - 1 : 4) : -1;
    // down leg
    for (int l = A.top; l >= 1 && l > wave_from; l--) { mgc_down<MGC_NT>(A, l, lds, tid); MGC_MARK(2 + (A.top - l)); }
    if (wave_from >= 0) {
        if (tid < 64) mgw_vcycle(A, lds, wave_from, tid);
        __syncthreads();
    } else {
        // MG.py:565 fill, the sweeps in registers of thread 0, closing fill (corners)
        double *V = lds + mgc_off(0), *F = V + 16;
        const bool uv = A.finest && A.top == 0;
        mgc_fill<MGC_NT>(V, 2, A.dx[0], A.bc, uv, tid);
        const bool plain = !uv && A.bc.code[0] != PYROHIP_BC_PERIODIC && A.bc.code[2] != PYROHIP_BC_PERIODIC;
        if (tid == 0 && A.nsmooth_bottom > 0) {
            const double xc0 = A.beta / (A.dx[0] * A.dx[0]), den0 = A.alpha + 2.0 * xc0 + 2.0 * xc0;
            const bool pow2 = A.allow_pow2 && mgc_is_pow2(xc0) && mgc_is_pow2(den0) &&
                              (1.0 / den0) * den0 == 1.0;
            if (plain && pow2)
                mgc_bottom_fast<true>(V, F, A.dx[0], A.alpha, A.beta, A.nsmooth_bottom, A.bc);
            else if (plain)
                mgc_bottom_fast<false>(V, F, A.dx[0], A.alpha, A.beta, A.nsmooth_bottom, A.bc);
            else
                mgc_bottom_regs(V, F, A.dx[0], A.alpha, A.beta, A.nsmooth_bottom, A.bc, uv);
        }
        __syncthreads();
        if (A.nsmooth_bottom > 0) mgc_fill<MGC_NT>(V, 2, A.dx[0], A.bc, uv, tid);
    }
    if (wave_from < 0) MGC_MARK(8);
    for (int l = (wave_from >= 0 ? wave_from + 1 : 1); l <= A.top; l++) { mgc_up<MGC_NT>(A, l, lds, tid); MGC_MARK(8 + l); }
    // write back: v of every level, f of the levels below the top
    for (int l = 0; l <= A.top; l++) {
        const int n = 2 << l, q = n + 2;
        const double *V = lds + mgc_off(l), *F = V + q * q;
        for (int idx = tid; idx < q * q; idx += MGC_NT) {
            const int i = idx / q, j = idx - i * q;
            A.v[l][(size_t)i * A.pitch[l] + j] = V[idx];
            if (l < A.top) A.f[l][(size_t)i * A.pitch[l] + j] = F[idx];
        }
    }
}


// ---------------------------------------------------------------------------
// Variable-coefficient mode  div(eta grad phi) = f
// pyro/multigrid/variable_coeff_MG.py:23-213, edge_coeffs.py:1-54.
// eta_x[i,j] = eta_{i-1/2,j}/dx^2, eta_y[i,j] = eta_{i,j-1/2}/dy^2.
// ---------------------------------------------------------------------------
__global__ void k_vc_edges(const double *__restrict__ c, double *__restrict__ ex,
                           double *__restrict__ ey, int n, int pitch, double dx2)
{
    const int j = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    const int i = 1 + blockIdx.y;
    if (j > n + 1 || i > n + 1) return;
    const size_t k = (size_t)i * pitch + j;
    double e = 0.5 * (c[k - pitch] + c[k]);   // edge_coeffs.py:21-25
    ex[k] = e / dx2;
    e = 0.5 * (c[k - 1] + c[k]);
    ey[k] = e / dx2;
}

// EdgeCoeffs.restrict, edge_coeffs.py:29-54
__global__ void k_vc_edges_restrict(const double *__restrict__ fx, const double *__restrict__ fy,
                                    int fpitch, double *__restrict__ cx, double *__restrict__ cy,
                                    int cpitch, int nc, double fdx2, double cdx2)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;   // 0 .. nc
    const int i = blockIdx.y;                              // 0 .. nc
    if (j > nc || i > nc) return;
    const size_t fk = (size_t)(1 + 2 * i) * fpitch + (1 + 2 * j);
    const size_t ck = (size_t)(1 + i) * cpitch + (1 + j);
    if (j < nc) cx[ck] = 0.5 * (fx[fk] + fx[fk + 1]) * fdx2 / cdx2;
    if (i < nc) cy[ck] = 0.5 * (fy[fk] + fy[fk + fpitch]) * fdx2 / cdx2;
}

// general mode (general_MG.py:107-242): alpha phi + div(beta grad phi) +
// gamma . grad phi = f with beta on the edges (ex, ey) and cell-centred alpha,
// gamma_x, gamma_y
struct MGGen { const double *a, *gx, *gy; };

// one colour of the variable-coefficient smoother, variable_coeff_MG.py:131-147
// (GEN: general_MG.py:130-160)
template <bool GEN>
__global__ __launch_bounds__(256) void k_vc_smooth(double *__restrict__ v,
                                                   const double *__restrict__ f,
                                                   const double *__restrict__ ex,
                                                   const double *__restrict__ ey, int n, int pitch,
                                                   double dx, int colour, MGBC bc, MGGen G)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = 1 + blockIdx.y;
    const int j = 1 + 2 * t + ((i - 1 + colour) & 1);
    if (j > n) return;
    const size_t k = (size_t)i * pitch + j;
    double vn;
    if (GEN) {
        const double gxc = 0.5 * G.gx[k] / dx, gyc = 0.5 * G.gy[k] / dx;
        const double denom = G.a[k] - ex[k + pitch] - ex[k] - ey[k + 1] - ey[k];
        vn = (f[k] - (ex[k + pitch] + gxc) * v[k + pitch] - (ex[k] - gxc) * v[k - pitch] -
              (ey[k + 1] + gyc) * v[k + 1] - (ey[k] - gyc) * v[k - 1]) / denom;
    } else {
        const double denom = ex[k + pitch] + ex[k] + ey[k + 1] + ey[k];
        vn = (-f[k] + ex[k + pitch] * v[k + pitch] + ex[k] * v[k - pitch] +
              ey[k + 1] * v[k + 1] + ey[k] * v[k - 1]) / denom;
    }
    v[k] = vn;
    if (i == 1) {
        if (bc.code[0] == PYROHIP_BC_PERIODIC) v[(size_t)(n + 1) * pitch + j] = vn;
        else v[j] = ghost_lo(bc.code[0], vn, bc.val[0], j, dx);
    }
    if (i == n) {
        if (bc.code[1] == PYROHIP_BC_PERIODIC) v[j] = vn;
        else v[(size_t)(n + 1) * pitch + j] = ghost_hi(bc.code[1], vn, bc.val[1], j, dx);
    }
    if (j == 1) {
        if (bc.code[2] == PYROHIP_BC_PERIODIC) v[(size_t)i * pitch + n + 1] = vn;
        else v[(size_t)i * pitch] = ghost_lo(bc.code[2], vn, bc.val[2], i, dx);
    }
    if (j == n) {
        if (bc.code[3] == PYROHIP_BC_PERIODIC) v[(size_t)i * pitch] = vn;
        else v[(size_t)i * pitch + n + 1] = ghost_hi(bc.code[3], vn, bc.val[3], i, dx);
    }
}

// variable_coeff_MG.py:191-213 (GEN: general_MG.py:196-242)
template <bool GEN>
__global__ __launch_bounds__(256) void k_vc_residual(const double *__restrict__ v,
                                                     const double *__restrict__ f,
                                                     const double *__restrict__ ex,
                                                     const double *__restrict__ ey,
                                                     double *__restrict__ r, int n, int pitch,
                                                     double dx, MGGen G)
{
    const int j = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    const int i = 1 + blockIdx.y;
    if (j > n) return;
    const size_t k = (size_t)i * pitch + j;
    double L;
    if (GEN) {
        const double gxc = 0.5 * G.gx[k] / dx, gyc = 0.5 * G.gy[k] / dx;
        L = G.a[k] * v[k] + ex[k + pitch] * (v[k + pitch] - v[k]) - ex[k] * (v[k] - v[k - pitch]) +
            ey[k + 1] * (v[k + 1] - v[k]) - ey[k] * (v[k] - v[k - 1]) +
            gxc * (v[k + pitch] - v[k - pitch]) + gyc * (v[k + 1] - v[k - 1]);
    } else {
        L = ex[k + pitch] * (v[k + pitch] - v[k]) - ex[k] * (v[k] - v[k - pitch]) +
            ey[k + 1] * (v[k + 1] - v[k]) - ey[k] * (v[k] - v[k - 1]);
    }
    r[k] = f[k] - L;
}

// MG.py:529-542
__global__ __launch_bounds__(256) void k_mg_residual(const double *__restrict__ v,
                                                     const double *__restrict__ f,
                                                     double *__restrict__ r, int n, int pitch,
                                                     double alpha, double beta, double dx2)
{
    const int j = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    const int i = 1 + blockIdx.y;
    if (j > n) return;
    const size_t k = (size_t)i * pitch + j;
    r[k] = f[k] - alpha * v[k] +
           beta * ((v[k - pitch] + v[k + pitch] - 2 * v[k]) / dx2 +
                   (v[k - 1] + v[k + 1] - 2 * v[k]) / dx2);
}

// patch.py:640-676: coarse f <- restrict(fine r)
__global__ __launch_bounds__(256) void k_mg_restrict(const double *__restrict__ fr, int fpitch,
                                                     double *__restrict__ cf, int cpitch, int nc)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= nc) return;
    const size_t fk = (size_t)(1 + 2 * i) * fpitch + (1 + 2 * j);
    cf[(size_t)(1 + i) * cpitch + (1 + j)] =
        0.25 * (fr[fk] + fr[fk + fpitch] + fr[fk + 1] + fr[fk + fpitch + 1]);
}

// Down leg of the V-cycle, constant coefficients: residual of the fine level
// (MG.py:529-542) AND its restriction into the coarse right-hand side
// (patch.py:640-676) in one pass -- one thread per COARSE cell evaluates the
// residual of its 2 x 2 fine cells with k_mg_residual's expression, stores
// them (r stays observable) and averages them in k_mg_restrict's order.  Saves
// re-reading r and one launch per level.
template <bool STORE_R>
__global__ __launch_bounds__(256) void k_mg_residual_restrict(
    const double *__restrict__ v, const double *__restrict__ f, double *__restrict__ r, int fpitch,
    double *__restrict__ cf, int cpitch, int nc, double alpha, double beta, double dx2,
    int ci0 = 0)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = ci0 + blockIdx.y;   // 0-based coarse row (ci0: first row of a slab)
    if (j >= nc) return;
    const size_t k00 = (size_t)(1 + 2 * i) * fpitch + (1 + 2 * j);
    auto res = [&](size_t k) {
        return f[k] - alpha * v[k] +
               beta * ((v[k - fpitch] + v[k + fpitch] - 2 * v[k]) / dx2 +
                       (v[k - 1] + v[k + 1] - 2 * v[k]) / dx2);
    };
    const double r00 = res(k00), r10 = res(k00 + fpitch), r01 = res(k00 + 1),
                 r11 = res(k00 + fpitch + 1);
    if (STORE_R) { r[k00] = r00; r[k00 + fpitch] = r10; r[k00 + 1] = r01; r[k00 + fpitch + 1] = r11; }
    cf[(size_t)(1 + i) * cpitch + (1 + j)] = 0.25 * (r00 + r10 + r01 + r11);
}

// patch.py:678-736 + MG.py:745-748: fine v += prolong(coarse v)
// one thread per FINE cell
__global__ __launch_bounds__(256) void k_mg_prolong_add(const double *__restrict__ cv, int cpitch,
                                                        double *__restrict__ fv, int fpitch,
                                                        int nf)
{
    const int fj = blockIdx.x * blockDim.x + threadIdx.x;   // 0-based fine interior index
    const int fi = blockIdx.y;
    if (fj >= nf) return;
    const int ci = 1 + (fi >> 1), cj = 1 + (fj >> 1);
    const size_t ck = (size_t)ci * cpitch + cj;
    const double c0 = cv[ck];
    const double m_x = 0.5 * (cv[ck + cpitch] - cv[ck - cpitch]);
    const double m_y = 0.5 * (cv[ck + 1] - cv[ck - 1]);
    double e;
    if (fi & 1) e = (fj & 1) ? c0 + 0.25 * m_x + 0.25 * m_y : c0 + 0.25 * m_x - 0.25 * m_y;
    else        e = (fj & 1) ? c0 - 0.25 * m_x + 0.25 * m_y : c0 - 0.25 * m_x - 0.25 * m_y;
    fv[(size_t)(1 + fi) * fpitch + (1 + fj)] += e;
}

// Crank-Nicolson right-hand side of the diffusion solver,
// pyro/diffusion/simulation.py:104-110:  f = phi + coef * L(phi)
__global__ __launch_bounds__(256) void k_mg_rhs_cn(const double *__restrict__ phi,
                                                   double *__restrict__ f, int n, int pitch,
                                                   double coef, double dx2, double dy2)
{
    const int j = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    const int i = 1 + blockIdx.y;
    if (j > n) return;
    const size_t k = (size_t)i * pitch + j;
    f[k] = phi[k] + coef * ((phi[k + pitch] + phi[k - pitch] - 2.0 * phi[k]) / dx2 +
                            (phi[k + 1] + phi[k - 1] - 2.0 * phi[k]) / dy2);
}

__global__ __launch_bounds__(256) void k_mg_copy_interior(const double *__restrict__ src,
                                                          double *__restrict__ dst, int n,
                                                          int pitch)
{
    const int j = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    const int i = 1 + blockIdx.y;
    if (j > n) return;
    dst[(size_t)i * pitch + j] = src[(size_t)i * pitch + j];
}

// sum over the interior of a^2 (mode 0) or ((a-b)/(a+small))^2 (mode 1)
__global__ __launch_bounds__(256) void k_mg_sumsq(const double *__restrict__ a,
                                                  const double *__restrict__ b, int n, int pitch,
                                                  int mode, double small,
                                                  double *__restrict__ partial)
{
    double s = 0.0;
    for (int i = 1 + blockIdx.y; i <= n; i += gridDim.y)
        for (int j = 1 + blockIdx.x * blockDim.x + threadIdx.x; j <= n;
             j += gridDim.x * blockDim.x) {
            const size_t k = (size_t)i * pitch + j;
            double d = a[k];
            if (mode == 1) d = (a[k] - b[k]) / (a[k] + small);
            s += d * d;
        }
    s = block_reduce_sum(s);
    if (threadIdx.x == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = s;
}

// The per-cycle diagnostics of MG.solve (MG.py:670-686) in ONE pass over the
// finest level: relative change of the solution (sum of ((v-old)/(v+small))^2,
// then old <- v) and the residual r = f - (alpha - beta L) v with the sum of
// r^2.  Replaces four kernels / two host syncs per V-cycle by two / one.
template <bool STORE_R, bool COPY_OLD>
__global__ __launch_bounds__(256) void k_mg_solve_diag(const double *__restrict__ v,
                                                       const double *__restrict__ f,
                                                       double *__restrict__ r,
                                                       double *__restrict__ old, int n, int pitch,
                                                       double alpha, double beta, double dx2,
                                                       double small, double *__restrict__ partial,
                                                       int row0, int row1)
{
    double srel = 0.0, sres = 0.0;
    for (int i = row0 + blockIdx.y; i <= row1; i += gridDim.y)
        for (int j = 1 + blockIdx.x * blockDim.x + threadIdx.x; j <= n;
             j += gridDim.x * blockDim.x) {
            const size_t k = (size_t)i * pitch + j;
            const double vk = v[k];
            const double d = (vk - old[k]) / (vk + small);
            srel += d * d;
            if (COPY_OLD) old[k] = vk;
            const double rr = f[k] - alpha * vk +
                              beta * ((v[k - pitch] + v[k + pitch] - 2 * vk) / dx2 +
                                      (v[k - 1] + v[k + 1] - 2 * vk) / dx2);
            if (STORE_R) r[k] = rr;
            sres += rr * rr;
        }
    srel = block_reduce_sum(srel);
    sres = block_reduce_sum(sres);
    if (threadIdx.x == 0) {
        const int b = blockIdx.y * gridDim.x + blockIdx.x, nb = gridDim.x * gridDim.y;
        partial[b] = srel;
        partial[nb + b] = sres;
    }
}

__global__ void k_sum_final2(const double *__restrict__ partial, int nb, double *__restrict__ out)
{
    double a = 0.0, b = 0.0;
    for (int k = threadIdx.x; k < nb; k += blockDim.x) { a += partial[k]; b += partial[nb + k]; }
    a = block_reduce_sum(a);
    b = block_reduce_sum(b);
    if (threadIdx.x == 0) { out[0] = a; out[1] = b; }
}

__global__ void k_sum_final(const double *__restrict__ partial, int nb, double *__restrict__ out)
{
    double s = 0.0;
    for (int b = threadIdx.x; b < nb; b += blockDim.x) s += partial[b];
    s = block_reduce_sum(s);
    if (threadIdx.x == 0) out[0] = s;
}

static MGBC make_bc(const pyrohip_mg *m, int level, bool for_v)
{
    MGBC b;
    for (int s = 0; s < 4; s++) {
        b.code[s] = m->bc[s];
        b.val[s] = (for_v && level == m->nlevels - 1) ? m->bcval[s] : nullptr;
    }
    return b;
}

static int mg_residual(pyrohip_mg *m, int level);

static double *plane(pyrohip_mg *m, int level, int var)
{
    MGLevel &L = m->lev[level];
    if (var == 2 && m->r_stale[level]) mg_residual(m, level);   // see pyrohip_mg::lazy_r
    switch (var) {
    case 0: return L.v;
    case 1: return L.f;
    case 2: return L.r;
    case 3: return L.c;
    case 4: return L.ex;
    case 5: return L.ey;
    case 6: return L.a;
    case 7: return L.gx;
    default: return L.gy;
    }
}

static int mg_fill(pyrohip_mg *m, int level, int var)
{
    MGLevel &L = m->lev[level];
    MGBC bc = make_bc(m, level, var == 0);
    double *a = plane(m, level, var);
    const int nt = L.n + 2;
    hipLaunchKernelGGL(k_mg_fill_x, dim3((nt + 255) / 256), dim3(256), 0, m->ctx->stream, a, L.n,
                       L.pitch, L.dx, bc);
    hipLaunchKernelGGL(k_mg_fill_y, dim3((nt + 255) / 256), dim3(256), 0, m->ctx->stream, a, L.n,
                       L.pitch, L.dx, bc);
    return 0;
}

static int mg_smooth_colour_launches(pyrohip_mg *m, int level, int nsmooth)
{
    MGLevel &L = m->lev[level];
    MGBC bc = make_bc(m, level, true);
    const double xcoeff = m->beta / (L.dx * L.dx);        // :567-568
    const double ycoeff = m->beta / (L.dx * L.dx);
    const double denom = m->alpha + 2.0 * xcoeff + 2.0 * ycoeff;
    const int half = (L.n + 1) / 2;
    const int bx = (half >= 256) ? 256 : 64;
    dim3 grid((half + bx - 1) / bx, L.n), block(bx);
    for (int it = 0; it < nsmooth; it++)
        for (int colour = 0; colour < 2; colour++)
            PYRO_LAUNCH(m->ctx, "k_mg_smooth", k_mg_smooth, grid, block, 0, L.v,
                        (const double *)L.f, L.n, L.pitch, L.dx, xcoeff, ycoeff, denom, colour, bc);
    return 0;
}

static int mg_zero(pyrohip_mg *m, int level, int var);

// can the prolongation of level-1's correction ride on the first smoothing
// launch of `level`?  (wide tile kernel only)
static bool mg_prolong_fusable(pyrohip_mg *m, int level, int nsmooth)
{
    const MGLevel &L = m->lev[level];
    return !m->vc && m->smoother != 0 && nsmooth > 0 && level > 0 &&
           (L.n + 2) * (L.n + 2) > MGS_CELLS;
}

// a smoothing launch read L.v and wrote L.v2: L.v2 is the solution now.  The first such
// launch of a solve cycle on the finest level leaves the solution before the cycle in the
// buffer it read: that buffer becomes old_phi (pyrohip_mg::capture_old), the previous
// old_phi the scratch buffer.  (Whole-level launches only: a slab's launches share buffers.)
static void mg_swap_solution(pyrohip_mg *m, int level)
{
    MGLevel &L = m->lev[level];
    double *read = L.v;
    L.v = L.v2;
    if (m->capture_old && level == m->nlevels - 1) {
        if (m->older) { L.v2 = m->older; m->older = m->old_phi; }   // keep the one before, too
        else L.v2 = m->old_phi;
        m->old_phi = read;
        m->capture_old = false;
        m->old_captured = true;
    } else
        L.v2 = read;
}

static int mg_smooth_tiles(pyrohip_mg *m, int level, int nsmooth, bool prolong = false,
                           int row0 = 1, int row1 = -1, int *nlaunch = nullptr, int tail = 0)
{
    MGLevel &L = m->lev[level];
    m->tail_done = 0;
    if (row1 < 0) row1 = L.n;
    const int nrows = row1 - row0 + 1;
    int launches = 0;
#ifndef PYRO_EMU
    static bool attr_set = false;
    if (!attr_set) {
        PYRO_CHECK_HIP(hipFuncSetAttribute((const void *)k_mg_smooth_tile<256, 0>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)MGS_LDS));
        PYRO_CHECK_HIP(hipFuncSetAttribute((const void *)k_mg_smooth_tile<MGW_NT, MGW_LP>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)MGW_LDS));
        PYRO_CHECK_HIP(hipFuncSetAttribute((const void *)k_mg_smooth_tile<MGW_NT, MGW_LP, true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)MGW_LDS));
        const void *bands[] = {
#define MGB_INST(P2, E) (const void *)k_mg_smooth_band<P2, E, false, 4>, (const void *)k_mg_smooth_band<P2, E, true, 4>
            MGB_INST(false, 0), MGB_INST(false, 1), MGB_INST(false, 2), MGB_INST(true, 0), MGB_INST(true, 1), MGB_INST(true, 2)
#undef MGB_INST
        };
        for (const void *fn : bands)
            PYRO_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)MGW_LDS));
        attr_set = true;
    }
#endif
    MGTile A;
    A.f = L.f; A.n = L.n; A.pitch = L.pitch; A.dx = L.dx;
    A.xc = m->beta / (L.dx * L.dx);
    A.yc = m->beta / (L.dx * L.dx);
    A.denom = m->alpha + 2.0 * A.xc + 2.0 * A.yc;
    A.rdenom = 1.0 / A.denom;
    A.kx = A.xc * A.rdenom; A.ky = A.yc * A.rdenom;
    const bool pow2 = mg_pow2(A.xc, A.yc, A.denom, m->allow_pow2);
    A.bc = make_bc(m, level, true);
    A.single = ((L.n + 2) * (L.n + 2) <= MGS_CELLS) ? 1 : 0;   // whole level in one tile
    A.cv = nullptr; A.cpitch = 0;
    if (prolong) { A.cv = m->lev[level - 1].v; A.cpitch = m->lev[level - 1].pitch; }
    A.vin_zero = 0;
    A.row0 = row0; A.row1 = row1;
    A.trace = nullptr;
#ifndef PYRO_EMU
    const bool tracing = m->trace;
    static long long *d_trace = nullptr;
    if (tracing) {
        if (!d_trace) PYRO_CHECK_HIP(hipMalloc((void **)&d_trace, 32 * sizeof(long long)));
        A.trace = d_trace;
    }
#endif
    if (m->v_is_zero[level]) {
        if (A.single) PYRO_TRY(mg_zero(m, level, 0));   // generic variant: materialise
        else A.vin_zero = 1;
        m->v_is_zero[level] = false;
    }
    int kmax = (m->kmax >= 1 && m->kmax <= MGW_KMAX) ? m->kmax : MGW_KMAX;
    // levels up to 1024^2 live in L2 / Infinity Cache and are launch-latency
    // bound: fuse as many iterations per launch as the 32-row region allows
    if (L.n <= m->nsmall && m->kmax_small > kmax) kmax = m->kmax_small;
    int left = nsmooth;
    // the large levels: the row-marching smoother (mg_march.hip), the ten iterations of a
    // V-cycle leg in one launch
    const int march_min = m->march_min, march_waves = m->march_waves;
    const bool hom_bc = !(A.bc.val[0] || A.bc.val[1] || A.bc.val[2] || A.bc.val[3]);
    while (!A.single && hom_bc && march_min > 0 && L.n >= march_min) {
        const int MK = 10;
        if (left < MK) break;
        MGMarch M;
        M.vin = L.v; M.f = L.f; M.vout = L.v2; M.n = L.n; M.pitch = L.pitch;
        M.xc = A.xc; M.yc = A.yc; M.denom = A.denom; M.rdenom = A.rdenom; M.kx = A.kx; M.ky = A.ky;
        for (int s = 0; s < 4; s++) M.code[s] = A.bc.code[s];
        M.cv = A.cv; M.cpitch = A.cpitch; M.vin_zero = A.vin_zero;
        M.row0 = row0; M.row1 = row1;
        // what rides on the launch that finishes the call (mg_march.h: MGMarch::tail)
        M.tail = (left == MK && row0 == 1 && row1 == L.n && m->march_tail) ? tail : 0;
        if (M.tail == 1 && !(level > 0 && !M.cv)) M.tail = 0;
        if (M.tail == 2 && !(m->old_captured && M.cv && level == m->nlevels - 1)) M.tail = 0;
        M.alpha = m->alpha; M.beta = m->beta; M.dx2 = L.dx * L.dx; M.rdx2 = 1.0 / M.dx2; M.small = 1.e-16;
        M.cf = nullptr; M.cfpitch = 0; M.old = nullptr; M.partial = nullptr;
        // row chunks: as many wavefronts as the device holds at once, not one more (two per
        // SIMD at 256 registers: a wavefront too many would run alone after all the others); the
        // parts that end at the top boundary start up to mgm_align rows lower (mg_march.hip:
        // mgm_part), so the last chunk is made that much shorter
        const int slots = march_waves > 0 ? march_waves : 8 * (m->ctx->num_cus > 0 ? m->ctx->num_cus : 256);
        const int pad = (M.code[0] != PYROHIP_BC_PERIODIC && row1 == L.n) ? mgm_align(MK) : 0;
        auto cut = [&]() {
            M.TJ = mgm_tj(MK, M.tail); M.ncs = (L.n + M.TJ - 1) / M.TJ;
            const bool sides = M.code[2] != PYROHIP_BC_PERIODIC && M.ncs >= 3 && m->march_side > 1.0 &&
                               row0 == 1 && row1 == L.n;
            auto chunks = [&](int rows, int least, int &cr) {   // chunks of about `rows` rows -> count
                cr = rows < least ? least : rows;
                if (M.tail) cr += cr & 1;                     // whole coarse rows
                return (nrows + cr - 1) / cr;
            };
            M.nchunks_side = 0; M.CR_side = 0;
            for (int nch = slots / M.ncs > 2 ? slots / M.ncs : 2; nch >= 2; nch--) {
                M.nchunks = chunks((nrows + pad + nch - 1) / nch, m->march_minrows, M.CR);
                if (!sides) break;
                // a wavefront of a side strip needs march_side times as long per row: fewer rows
                const int steps = M.CR + 6 * MK;              // apron below and above, 2K steps to drain
                M.nchunks_side = chunks((int)(steps / m->march_side) - 6 * MK, 8, M.CR_side);
                if ((M.ncs - 2) * M.nchunks + 2 * M.nchunks_side <= slots || nch == 2) break;
            }
            return mg_march_usable(M, MK);
        };
        bool ok = cut();
        if (!ok && M.tail) { M.tail = 0; ok = cut(); }       // without the tail, then
        if (!ok) break;
        if (M.tail == 1) { M.cf = m->lev[level - 1].f; M.cfpitch = m->lev[level - 1].pitch; }
        if (M.tail == 2) {
            const int nb = mg_march_blocks(M);
            PYRO_TRY(m->ctx->reduce.ensure((2 * (size_t)nb + 4) * sizeof(double)));
            M.old = nullptr; M.partial = (double *)m->ctx->reduce.p;
            m->diag_nb = nb; m->diag_part = M.partial;
        }
        PYRO_TRY(mg_march_launch(m->ctx, M, pow2, MK));
        m->tail_done = M.tail;
        m->n_tail[M.tail]++;
        launches++;
        mg_swap_solution(m, level);
        left -= MK;
        A.cv = nullptr; A.vin_zero = 0;
    }
    while (left > 0) {
        const int K = A.single ? left : (left < kmax ? left : kmax);
        A.K = K;
        A.vin = L.v; A.vout = L.v2;
        if (A.single) {
            A.TI = L.n; A.TJ = L.n; A.ntj = 1; A.ntiles = 1;
            PYRO_LAUNCH(m->ctx, "k_mg_smooth_tile", (k_mg_smooth_tile<256, 0>), dim3(1), dim3(256),
                        MGS_LDS, A);
        } else {
            A.TI = MGW_RI - 4 * K; A.TJ = MGW_LP - 4 * K;
            A.ntj = (L.n + A.TJ - 1) / A.TJ;
            // Small levels: a launch lasts as long as ONE workgroup needs for its
            // region (2K sweeps over up to 64 x 128 cells on one CU), while most of
            // the 256 CUs idle.  Shorter tiles (fewer region rows per workgroup,
            // more workgroups) cut that latency; the extra apron rows are free here.
            const int target = m->small_tiles >= 0 ? m->small_tiles : MG_SMALL_TILES_DEFAULT;
            if (target > 0) {
                const int want = (target + A.ntj - 1) / A.ntj;
                int ti = (nrows + want - 1) / want;
                if (ti < 4) ti = 4;
                if (ti < A.TI) A.TI = ti;
            }
            const int nti = (nrows + A.TI - 1) / A.TI;
            A.ntiles = nti * A.ntj;
            // measured per V-cycle at 2048^2 / 4096^2 (tools/mg_ab.sh): band kernel up to
            // 1024^2: 625 / 1497 us, up to 2048^2: 611 / 1485, everywhere: 615 / 1501
            const bool band = L.n <= m->band_maxn;
            // the band kernel: homogeneous boundaries; its EDGE instance (ghost values
            // synthesised at the physical sides) wherever a tile can touch one
            const bool hom = !(A.bc.val[0] || A.bc.val[1] || A.bc.val[2] || A.bc.val[3]);
            // 0: no physical side; 1: mirror ghosts (+-own value); 2: value-0 ghosts among them
            const bool gen_edge = m->band_genedge;
            int edge = (A.bc.code[0] != PYROHIP_BC_PERIODIC || A.bc.code[2] != PYROHIP_BC_PERIODIC) ? 1 : 0;
            for (int sd = 0; sd < 4; sd++)
                if (A.bc.code[sd] == PYROHIP_BC_CONST || (edge && gen_edge)) edge = 2;
            if (band && hom) {
                using BandT = void (*)(MGTile);
#define MGB_ROW(P2, E) {k_mg_smooth_band<P2, E, false, 4>, k_mg_smooth_band<P2, E, true, 4>}
                static const BandT inst[2][3][2] = {{MGB_ROW(false, 0), MGB_ROW(false, 1), MGB_ROW(false, 2)},
                                                    {MGB_ROW(true, 0), MGB_ROW(true, 1), MGB_ROW(true, 2)}};
#undef MGB_ROW
                PYRO_LAUNCH(m->ctx, "k_mg_smooth_band", inst[pow2 ? 1 : 0][edge][A.cv ? 1 : 0],
                            dim3(A.ntiles), dim3(1024), MGW_LDS, A);
            }
            else if (pow2)
                PYRO_LAUNCH(m->ctx, "k_mg_smooth_tile", (k_mg_smooth_tile<MGW_NT, MGW_LP, true>),
                            dim3(A.ntiles), dim3(MGW_NT), MGW_LDS, A);
            else
                PYRO_LAUNCH(m->ctx, "k_mg_smooth_tile", (k_mg_smooth_tile<MGW_NT, MGW_LP>),
                            dim3(A.ntiles), dim3(MGW_NT), MGW_LDS, A);
        }
#ifndef PYRO_EMU
        if (tracing && !A.single) {
            long long h[32];
            PYRO_CHECK_HIP(hipMemcpy(h, d_trace, sizeof(h), hipMemcpyDeviceToHost));
            fprintf(stderr, "band trace n=%d K=%d tiles=%d prolong=%d: stage %lld ghosts0 %lld passes", L.n, K,
                    A.ntiles, A.cv != nullptr, h[1] - h[0], h[2] - h[1]);
            for (int s = 1; s <= 2 * K; s++) fprintf(stderr, " %lld", h[2 + s] - h[1 + s]);
            fprintf(stderr, " store %lld total %lld\n", h[23] - h[2 + 2 * K], h[23] - h[0]);
        }
#endif
        if (A.single) { double *t = L.v; L.v = L.v2; L.v2 = t; }
        else mg_swap_solution(m, level);
        launches++;
        left -= K;
        A.cv = nullptr;   // only the first launch carries the prolongation
        A.vin_zero = 0;
    }
    if (nlaunch) *nlaunch = launches;
    return 0;
}

// `corners`: also make the corner ghosts exact (needed only when the array is
// handed to the host; no kernel reads corners)
static int mg_smooth(pyrohip_mg *m, int level, int nsmooth, bool corners = true,
                     bool prolong = false, int tail = 0)
{
    m->tail_done = 0;
    if (m->vc) {   // variable coefficients: one launch per colour
        PYRO_TRY(mg_fill(m, level, 0));
        MGLevel &L = m->lev[level];
        MGBC bc = make_bc(m, level, true);
        const int half = (L.n + 1) / 2;
        const int bx = (half >= 256) ? 256 : 64;
        dim3 grid((half + bx - 1) / bx, L.n), block(bx);
        const MGGen G{L.a, L.gx, L.gy};
        for (int it = 0; it < nsmooth; it++)
            for (int colour = 0; colour < 2; colour++) {
                if (m->vc == 2)
                    PYRO_LAUNCH(m->ctx, "k_vc_smooth", k_vc_smooth<true>, grid, block, 0, L.v,
                                (const double *)L.f, (const double *)L.ex, (const double *)L.ey,
                                L.n, L.pitch, L.dx, colour, bc, G);
                else
                    PYRO_LAUNCH(m->ctx, "k_vc_smooth", k_vc_smooth<false>, grid, block, 0, L.v,
                                (const double *)L.f, (const double *)L.ex, (const double *)L.ey,
                                L.n, L.pitch, L.dx, colour, bc, G);
            }
        return 0;
    }
    if (m->smoother == 0 || nsmooth <= 0) {
        PYRO_TRY(mg_fill(m, level, 0));                   // MG.py:565
        return nsmooth > 0 ? mg_smooth_colour_launches(m, level, nsmooth) : 0;
    }
    // the tile kernel refreshes the edge ghosts itself on load (= the fill_BC
    // of MG.py:565) and leaves them current on exit
    PYRO_TRY(mg_smooth_tiles(m, level, nsmooth, prolong, 1, -1, nullptr, tail));
    m->corners_stale[level] = !corners;
    return corners ? mg_fill(m, level, 0) : 0;
}

static int mg_residual(pyrohip_mg *m, int level)
{
    MGLevel &L = m->lev[level];
    const int bx = (L.n >= 256) ? 256 : 64;
    m->r_stale[level] = false;
    if (m->vc) {
        const MGGen G{L.a, L.gx, L.gy};
        if (m->vc == 2)
            hipLaunchKernelGGL(k_vc_residual<true>, dim3((L.n + bx - 1) / bx, L.n), dim3(bx), 0,
                               m->ctx->stream, (const double *)L.v, (const double *)L.f,
                               (const double *)L.ex, (const double *)L.ey, L.r, L.n, L.pitch, L.dx,
                               G);
        else
            hipLaunchKernelGGL(k_vc_residual<false>, dim3((L.n + bx - 1) / bx, L.n), dim3(bx), 0,
                               m->ctx->stream, (const double *)L.v, (const double *)L.f,
                               (const double *)L.ex, (const double *)L.ey, L.r, L.n, L.pitch, L.dx,
                               G);
        return 0;
    }
    hipLaunchKernelGGL(k_mg_residual, dim3((L.n + bx - 1) / bx, L.n), dim3(bx), 0, m->ctx->stream,
                       (const double *)L.v, (const double *)L.f, L.r, L.n, L.pitch, m->alpha,
                       m->beta, L.dx * L.dx);
    return 0;
}

static int mg_restrict(pyrohip_mg *m, int fine)
{
    MGLevel &F = m->lev[fine], &Cc = m->lev[fine - 1];
    const int bx = (Cc.n >= 256) ? 256 : 64;
    PYRO_LAUNCH(m->ctx, "k_mg_restrict", k_mg_restrict, dim3((Cc.n + bx - 1) / bx, Cc.n), dim3(bx), 0, (const double *)F.r, F.pitch, Cc.f, Cc.pitch, Cc.n);
    return 0;
}

static int mg_prolong_add(pyrohip_mg *m, int fine)
{
    MGLevel &F = m->lev[fine], &Cc = m->lev[fine - 1];
    const int bx = (F.n >= 256) ? 256 : 64;
    PYRO_LAUNCH(m->ctx, "k_mg_prolong_add", k_mg_prolong_add, dim3((F.n + bx - 1) / bx, F.n), dim3(bx), 0, (const double *)Cc.v, Cc.pitch, F.v, F.pitch, F.n);
    return 0;
}

// (hipMemset runs at ~1.2 TB/s on this part: 110 us for the 4096^2 level)
__global__ __launch_bounds__(256) void k_mg_zero_plane(double *__restrict__ a, size_t n)
{
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x)
        a[k] = 0.0;
}

static int mg_zero(pyrohip_mg *m, int level, int var)
{
    MGLevel &L = m->lev[level];
    // patch.py:562-573 zeroes the whole array incl. ghosts
    const size_t n = (size_t)(L.n + 2) * L.pitch;
    double *a = plane(m, level, var);
    if (n >= ((size_t)1 << 20)) {
        hipLaunchKernelGGL(k_mg_zero_plane, dim3(4096), dim3(256), 0, m->ctx->stream, a, n);
        PYRO_CHECK_HIP(hipGetLastError());
    } else
        PYRO_CHECK_HIP(hipMemsetAsync(a, 0, n * sizeof(double), m->ctx->stream));
    return 0;
}

// returns sum (not sqrt) into host *out
static int mg_sumsq(pyrohip_mg *m, const double *a, const double *b, int level, int mode,
                    double *out)
{
    pyrohip_ctx *c = m->ctx;
    MGLevel &L = m->lev[level];
    // ~2048 workgroups on the large levels, as many across as a row has pieces of 256 columns
    // (512 workgroups with long dependent sums: 84 us for the 4096^2 level, 1.6 TB/s)
    const int gx = L.n >= 4096 ? 16 : (L.n >= 256 ? L.n / 256 : 1);
    const int gy = L.n >= 64 ? (2048 / gx < L.n ? 2048 / gx : L.n) : 1;
    dim3 grid(gx, gy), block(256);
    const int nb = grid.x * grid.y;
    PYRO_TRY(c->reduce.ensure((nb + 2) * sizeof(double)));
    double *part = (double *)c->reduce.p;
    hipLaunchKernelGGL(k_mg_sumsq, grid, block, 0, c->stream, a, b, L.n, L.pitch, mode, 1.e-16,
                       part);
    hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(256), 0, c->stream, (const double *)part, nb,
                       part + nb);
    PYRO_CHECK_HIP(hipGetLastError());
    PYRO_CHECK_HIP(hipMemcpyAsync(c->reduce_host, part + nb, sizeof(double),
                                  hipMemcpyDeviceToHost, c->stream));
    PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));
    *out = ((double *)c->reduce_host)[0];
    return 0;
}

static int mg_coarse_vcycle(pyrohip_mg *m, int top)
{
#ifndef PYRO_EMU
    static bool attr_set = false;
    if (!attr_set) {
        PYRO_CHECK_HIP(hipFuncSetAttribute((const void *)k_mg_coarse_vcycle,
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)MGC_LDS));
        attr_set = true;
    }
#endif
    MGCoarse A;
    for (int l = 0; l <= MGC_TOP; l++) {
        MGLevel &L = m->lev[l <= top ? l : top];
        A.v[l] = L.v; A.f[l] = L.f; A.r[l] = L.r; A.pitch[l] = L.pitch; A.dx[l] = L.dx;
    }
    A.top = top;
    A.finest = (top == m->nlevels - 1) ? 1 : 0;
    A.alpha = m->alpha; A.beta = m->beta;
    A.nsmooth = m->nsmooth; A.nsmooth_bottom = m->nsmooth_bottom;
    A.allow_pow2 = m->allow_pow2 ? 1 : 0;
    A.band64 = m->coarse_band64 ? 1 : 0;
    A.wave_levels = m->coarse_wave ? 1 : 0;
    A.zero_mask = 0;
    for (int l = 0; l <= top; l++)
        if (m->v_is_zero[l]) { A.zero_mask |= 1u << l; m->v_is_zero[l] = false; }
    A.bc = make_bc(m, top, true);
    A.trace = nullptr;
#ifndef PYRO_EMU
    const bool tracing = m->trace;
    static long long *d_trace = nullptr;
    if (tracing) {
        if (!d_trace) PYRO_CHECK_HIP(hipMalloc((void **)&d_trace, 16 * sizeof(long long)));
        A.trace = d_trace;
    }
#endif
    PYRO_LAUNCH(m->ctx, "k_mg_coarse_vcycle", k_mg_coarse_vcycle, dim3(1), dim3(MGC_NT), MGC_LDS,
                A);
#ifndef PYRO_EMU
    if (tracing) {
        long long h[16];
        PYRO_CHECK_HIP(hipMemcpy(h, d_trace, sizeof(h), hipMemcpyDeviceToHost));
        fprintf(stderr, "mgc trace (cycles): stage %lld | down", h[1] - h[0]);
        for (int k = 2; k < 2 + top; k++) fprintf(stderr, " %lld", h[k] - h[k - 1]);
        fprintf(stderr, " | bottom %lld | up", h[8] - h[1 + top]);
        for (int l = 1; l <= top; l++) fprintf(stderr, " %lld", h[8 + l] - h[8 + l - 1]);
        fprintf(stderr, " | top level down: smooth %lld\n", h[15] - h[14]);
    }
#endif
    for (int l = 0; l <= top; l++) m->corners_stale[l] = false;   // full fills inside
    return 0;
}

static int mg_vcycle(pyrohip_mg *m, int level)
{
    if (!m->vc && m->smoother != 0 && m->coarse_kernel && level <= MGC_TOP)
        return mg_coarse_vcycle(m, level);
    if (level > 0) {
        // inside solve() nobody reads r: residual and restriction ride on the smoothing launch
        const bool no_r = m->lazy_r && m->in_solve;
        PYRO_TRY(mg_smooth(m, level, m->nsmooth, false, false,
                           (!m->vc && m->fuse_res_restrict && no_r) ? 1 : 0)); // MG.py:722
        if (m->tail_done == 1) m->r_stale[level] = true;  // :724 + :731-732 done there
        else if (!m->vc && m->fuse_res_restrict) {        // :724 + :731-732 in one pass
            MGLevel &F = m->lev[level], &Cc = m->lev[level - 1];
            const int bx = (Cc.n >= 256) ? 256 : 64;
            const bool store = !(m->lazy_r && m->in_solve);
            using RRT = void (*)(const double *, const double *, double *, int, double *, int, int,
                                 double, double, double, int);
            const RRT rr = store ? (RRT)k_mg_residual_restrict<true> : (RRT)k_mg_residual_restrict<false>;
            PYRO_LAUNCH(m->ctx, "k_mg_residual_restrict", rr, dim3((Cc.n + bx - 1) / bx, Cc.n), dim3(bx), 0, (const double *)F.v,
                        (const double *)F.f, F.r, F.pitch, Cc.f, Cc.pitch, Cc.n, m->alpha, m->beta,
                        F.dx * F.dx, 0);
            m->r_stale[level] = !store;
        } else {
            PYRO_TRY(mg_residual(m, level));              // :724
            PYRO_TRY(mg_restrict(m, level));              // :731-732
        }
        PYRO_TRY(mg_vcycle(m, level - 1));                // :735
        const bool fuse = mg_prolong_fusable(m, level, m->nsmooth);
        if (!fuse) PYRO_TRY(mg_prolong_add(m, level));    // :745-748 (else: while staging below)
        if (m->smoother == 0 || m->vc) PYRO_TRY(mg_fill(m, level, 0));   // :751 (tile smoother: on load)
        const bool diag = m->diag_req && level == m->nlevels - 1;
        PYRO_TRY(mg_smooth(m, level, m->nsmooth, false, fuse, diag ? 2 : 0)); // :758
        if (diag && m->tail_done == 2) m->diag_done = true;
    } else {
        PYRO_TRY(mg_smooth(m, level, m->nsmooth_bottom, false)); // :776
        if (m->smoother == 0 || m->vc) PYRO_TRY(mg_fill(m, level, 0));     // :778
    }
    return 0;
}

}  // namespace pyro

using namespace pyro;

#define MG_CHECK_LEVEL(m, level)                                                   \
    PYRO_REQUIRE((m) != nullptr, "NULL mg");                                       \
    PYRO_REQUIRE((level) >= 0 && (level) < (m)->nlevels, "level out of range")

extern "C" {

int pyrohip_mg_create(pyrohip_ctx *c, int nx, double xmin, double xmax, double ymin, double ymax,
                      const int *bc, double alpha, double beta, int nsmooth, int nsmooth_bottom,
                      pyrohip_mg **out)
{
    PYRO_REQUIRE(c && bc && out, "NULL argument");
    PYRO_REQUIRE(nx >= 2 && (nx & (nx - 1)) == 0, "nx must be a power of two >= 2");
    PYRO_REQUIRE((xmax - xmin) == (ymax - ymin),
                 "multigrid requires a square domain (MG.py:197-198)");
    for (int s = 0; s < 4; s++)
        PYRO_REQUIRE(bc[s] == PYROHIP_BC_OUTFLOW || bc[s] == PYROHIP_BC_REFLECT_ODD ||
                         bc[s] == PYROHIP_BC_PERIODIC || bc[s] == PYROHIP_BC_REFLECT_EVEN ||
                         bc[s] == PYROHIP_BC_CONST,
                     "bad BC code");
    PYRO_REQUIRE((bc[0] == PYROHIP_BC_PERIODIC) == (bc[1] == PYROHIP_BC_PERIODIC) &&
                     (bc[2] == PYROHIP_BC_PERIODIC) == (bc[3] == PYROHIP_BC_PERIODIC),
                 "periodic BCs must be paired (boundary.py:186-192)");
    PYRO_CHECK_HIP(hipSetDevice(c->device));
    pyrohip_mg *m = new pyrohip_mg();
    m->ctx = c;
    m->nx = nx;
    m->alpha = alpha; m->beta = beta;
    m->nsmooth = nsmooth; m->nsmooth_bottom = nsmooth_bottom;
    memcpy(m->bc, bc, sizeof(int) * 4);
    int nl = 0;
    for (int t = nx; t > 1; t >>= 1) nl++;   // == int(log(nx)/log(2)) for powers of two
    m->nlevels = nl;
    PYRO_REQUIRE(nl <= MG_MAXLEV, "too many levels");
    size_t total = 16;
    int nt = 2;
    for (int l = 0; l < nl; l++) {
        Geom g = make_geom(nt, nt, 1);
        total += 4 * g.plane + 16;
        nt *= 2;
    }
    Geom gf = make_geom(nx, nx, 1);
    total += gf.plane + 16;
    PYRO_CHECK_HIP(hipMalloc((void **)&m->pool, total * sizeof(double)));
    PYRO_CHECK_HIP(hipMemsetAsync(m->pool, 0, total * sizeof(double), c->stream));
    double *p = m->pool;
    nt = 2;
    for (int l = 0; l < nl; l++) {
        Geom g = make_geom(nt, nt, 1);
        MGLevel &L = m->lev[l];
        L.n = nt;
        L.pitch = g.pitch;
        L.dx = (xmax - xmin) / nt;   // patch.py:121
        L.v = p + geom_lead(g); p += g.plane;
        L.f = p + geom_lead(g); p += g.plane;
        L.r = p + geom_lead(g); p += g.plane;
        L.v2 = p + geom_lead(g); p += g.plane;
        p += 16;
        nt *= 2;
    }
    m->old_phi = p + geom_lead(gf);
    *out = m;
    return 0;
}

int pyrohip_mg_destroy(pyrohip_mg *m)
{
    if (!m) return 0;
    (void)hipSetDevice(m->ctx->device);
    (void)hipStreamSynchronize(m->ctx->stream);
    if (m->pool) (void)hipFree(m->pool);
    if (m->older_base) (void)hipFree(m->older_base);
    if (m->vc_pool) (void)hipFree(m->vc_pool);
    if (m->gen_pool) (void)hipFree(m->gen_pool);
    for (int s = 0; s < 4; s++)
        if (m->bcval[s]) (void)hipFree(m->bcval[s]);
    for (int k = 0; k < 2; k++)
        if (m->ev[k]) (void)hipEventDestroy(m->ev[k]);
    delete m;
    return 0;
}

int pyrohip_mg_set_helmholtz(pyrohip_mg *m, double alpha, double beta)
{
    PYRO_REQUIRE(m, "NULL mg");
    PYRO_REQUIRE(m->vc == 0, "constant-coefficient solvers only");
    m->alpha = alpha; m->beta = beta;   // read at every launch
    return 0;
}

int pyrohip_mg_get_tuning(pyrohip_mg *m, pyrohip_mg_tuning *t)
{
    PYRO_REQUIRE(m && t, "NULL argument");
    t->kmax = m->kmax; t->kmax_small = m->kmax_small_tuned; t->nsmall = m->nsmall;
    t->march_min = m->march_min; t->march_waves = m->march_waves; t->march_side = m->march_side;
    t->march_minrows = m->march_minrows; t->fuse_res_restrict = m->fuse_res_restrict;
    t->lazy_residual = m->lazy_r ? 1 : 0; t->allow_pow2 = m->allow_pow2 ? 1 : 0;
    t->small_tiles = m->small_tiles; t->band_maxn = m->band_maxn;
    t->band_genedge = m->band_genedge ? 1 : 0; t->coarse_band64 = m->coarse_band64 ? 1 : 0;
    t->speculate = m->speculate; t->trace = m->trace ? 1 : 0; t->spec_debug = m->spec_debug ? 1 : 0;
    t->march_tail = m->march_tail; t->coarse_wave = m->coarse_wave ? 1 : 0;
    return 0;
}

int pyrohip_mg_set_tuning(pyrohip_mg *m, const pyrohip_mg_tuning *t)
{
    PYRO_REQUIRE(m && t, "NULL argument");
    PYRO_REQUIRE(t->kmax >= 1 && t->kmax <= MGW_KMAX, "kmax: 1..5 iterations per tile launch");
    PYRO_REQUIRE(t->kmax_small >= 0 && t->kmax_small <= 10, "kmax_small: 0..10");
    PYRO_REQUIRE(t->speculate >= 0 && t->speculate <= 2, "speculate: 0, 1 or 2");
    m->kmax = t->kmax; m->kmax_small_tuned = t->kmax_small; m->nsmall = t->nsmall;
    if (m->kmax_small != 0) m->kmax_small = t->kmax_small;      // (0: switched off by set_smoother)
    m->march_min = t->march_min; m->march_waves = t->march_waves; m->march_side = t->march_side;
    m->march_minrows = t->march_minrows; m->fuse_res_restrict = t->fuse_res_restrict;
    m->lazy_r = t->lazy_residual != 0; m->allow_pow2 = t->allow_pow2 != 0;
    m->small_tiles = t->small_tiles; m->band_maxn = t->band_maxn;
    m->band_genedge = t->band_genedge != 0; m->coarse_band64 = t->coarse_band64 != 0;
    m->speculate = t->speculate; m->trace = t->trace != 0; m->spec_debug = t->spec_debug != 0;
    m->march_tail = t->march_tail; m->coarse_wave = t->coarse_wave != 0;
    return 0;
}

int pyrohip_mg_tail_counts(pyrohip_mg *m, int *restrictions, int *diagnostics)
{
    PYRO_REQUIRE(m && restrictions && diagnostics, "NULL argument");
    *restrictions = m->n_tail[1];
    *diagnostics = m->n_tail[2];
    return 0;
}

int pyrohip_mg_set_smoother(pyrohip_mg *m, int kind)
{
    PYRO_REQUIRE(m, "NULL mg");
    PYRO_REQUIRE(kind >= 0 && kind <= 25, "smoother must be 0, 1, 10+kmax or 20+kmax");
    // 10 + k selects the tile smoother with k fused iterations (tuning knob)
    // 20 + k: the same without the single-workgroup coarse V-cycle kernel
    m->coarse_kernel = 1;
    m->kmax_small = m->kmax_small_tuned;
    if (kind >= 20) { m->smoother = 1; m->kmax = kind - 20; m->coarse_kernel = 0; m->kmax_small = 0; }
    else if (kind >= 10) { m->smoother = 1; m->kmax = kind - 10; }
    else m->smoother = kind;
    return 0;
}

int pyrohip_mg_nlevels(pyrohip_mg *m, int *nlevels)
{
    PYRO_REQUIRE(m && nlevels, "NULL argument");
    *nlevels = m->nlevels;
    return 0;
}

int pyrohip_mg_set(pyrohip_mg *m, int level, int var, const double *host)
{
    MG_CHECK_LEVEL(m, level);
    PYRO_REQUIRE(var >= 0 && var <= 2 && host, "bad var / NULL host");
    MGLevel &L = m->lev[level];
    const int q = L.n + 2;
    PYRO_CHECK_HIP(hipMemcpy2DAsync(plane(m, level, var), L.pitch * sizeof(double), host,
                                    q * sizeof(double), q * sizeof(double), q,
                                    hipMemcpyHostToDevice, m->ctx->stream));
    PYRO_CHECK_HIP(hipStreamSynchronize(m->ctx->stream));
    return 0;
}

int pyrohip_mg_get(pyrohip_mg *m, int level, int var, double *host)
{
    MG_CHECK_LEVEL(m, level);
    PYRO_REQUIRE(var >= 0 && var <= (m->vc == 2 ? 8 : m->vc ? 5 : 2) && host,
                 "bad var / NULL host");
    if (var == 0 && m->corners_stale[level]) {   // index-for-index incl. corner ghosts
        PYRO_TRY(mg_fill(m, level, 0));
        m->corners_stale[level] = false;
    }
    MGLevel &L = m->lev[level];
    const int q = L.n + 2;
    PYRO_CHECK_HIP(hipMemcpy2DAsync(host, q * sizeof(double), plane(m, level, var),
                                    L.pitch * sizeof(double), q * sizeof(double), q,
                                    hipMemcpyDeviceToHost, m->ctx->stream));
    PYRO_CHECK_HIP(hipStreamSynchronize(m->ctx->stream));
    return 0;
}

int pyrohip_mg_set_bcval(pyrohip_mg *m, int side, const double *vals)
{
    PYRO_REQUIRE(m, "NULL mg");
    PYRO_REQUIRE(side >= 0 && side < 4, "side out of range");
    PYRO_CHECK_HIP(hipStreamSynchronize(m->ctx->stream));
    if (m->bcval[side]) { PYRO_CHECK_HIP(hipFree(m->bcval[side])); m->bcval[side] = nullptr; }
    if (vals) {
        size_t n = (size_t)m->nx + 2;
        PYRO_CHECK_HIP(hipMalloc((void **)&m->bcval[side], n * sizeof(double)));
        PYRO_CHECK_HIP(hipMemcpy(m->bcval[side], vals, n * sizeof(double), hipMemcpyHostToDevice));
    }
    return 0;
}

int pyrohip_mg_zero(pyrohip_mg *m, int level, int var)
{
    MG_CHECK_LEVEL(m, level);
    PYRO_REQUIRE(var >= 0 && var <= 2, "bad var");
    return mg_zero(m, level, var);
}

int pyrohip_mg_fill_bc(pyrohip_mg *m, int level, int var)
{
    MG_CHECK_LEVEL(m, level);
    PYRO_REQUIRE(var >= 0 && var <= 2, "bad var");
    PYRO_TRY(mg_fill(m, level, var));
    PYRO_CHECK_HIP(hipGetLastError());
    return 0;
}

int pyrohip_mg_smooth(pyrohip_mg *m, int level, int nsmooth)
{
    MG_CHECK_LEVEL(m, level);
    PYRO_REQUIRE(nsmooth >= 0, "nsmooth < 0");
    PYRO_TRY(mg_smooth(m, level, nsmooth));
    PYRO_CHECK_HIP(hipGetLastError());
    return 0;
}

int pyrohip_mg_residual(pyrohip_mg *m, int level)
{
    MG_CHECK_LEVEL(m, level);
    PYRO_TRY(mg_residual(m, level));
    PYRO_CHECK_HIP(hipGetLastError());
    return 0;
}

int pyrohip_mg_restrict(pyrohip_mg *m, int fine)
{
    MG_CHECK_LEVEL(m, fine);
    PYRO_REQUIRE(fine >= 1, "no coarser level");
    PYRO_TRY(mg_restrict(m, fine));
    PYRO_CHECK_HIP(hipGetLastError());
    return 0;
}

int pyrohip_mg_prolong_add(pyrohip_mg *m, int fine)
{
    MG_CHECK_LEVEL(m, fine);
    PYRO_REQUIRE(fine >= 1, "no coarser level");
    PYRO_TRY(mg_prolong_add(m, fine));
    PYRO_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---- row windows: building blocks of a V-cycle on x slabs (multigrid/slab.py) ----
int pyrohip_mg_rows_kmax(pyrohip_mg *m, int level, int *k)
{
    MG_CHECK_LEVEL(m, level);
    PYRO_REQUIRE(k, "NULL argument");
    const MGLevel &L = m->lev[level];
    // ten iterations (a whole V-cycle leg) in one launch: the row-marching kernel on the
    // large levels, the band kernel with its deep apron on the small ones; five in between
    const bool hom = !(level == m->nlevels - 1 && (m->bcval[0] || m->bcval[1] || m->bcval[2] || m->bcval[3]));
    bool cst = false;
    for (int s = 0; s < 4; s++) cst = cst || m->bc[s] == PYROHIP_BC_CONST;
    const bool march = hom && !cst && m->march_min > 0 && L.n >= m->march_min && L.n >= 2 * MGM_COLS &&
                       m->bc[0] != PYROHIP_BC_PERIODIC;
    const bool small10 = L.n <= m->nsmall && m->kmax_small_tuned >= 10;
    *k = (m->vc || m->smoother == 0) ? 0 : ((march || small10) ? 10 : MGW_KMAX);
    return 0;
}

int pyrohip_mg_smooth_rows(pyrohip_mg *m, int level, int nsweeps, int row0, int row1, int prolong)
{
    MG_CHECK_LEVEL(m, level);
    MGLevel &L = m->lev[level];
    PYRO_REQUIRE(!m->vc && m->smoother != 0, "row windows: constant coefficients, tile smoother");
    PYRO_REQUIRE((L.n + 2) * (L.n + 2) > MGS_CELLS, "row windows: levels above 64^2 only");
    PYRO_REQUIRE(nsweeps >= 1 && nsweeps <= 10, "one launch: 1..10 iterations");
    int kcap = 0;
    PYRO_TRY(pyrohip_mg_rows_kmax(m, level, &kcap));
    PYRO_REQUIRE(nsweeps <= kcap, "more iterations than one launch does on this level "
                                  "(pyrohip_mg_rows_kmax)");
    PYRO_REQUIRE(row0 >= 1 && row1 <= L.n && row0 <= row1, "rows outside the level");
    PYRO_REQUIRE(!prolong || level > 0, "no coarser level to prolong from");
    // exactly `nsweeps` iterations in ONE launch (no halo exchange could happen between
    // two launches of a split call): the tuning values do not apply to row windows
    const int ks = m->kmax_small, km = m->kmax;
    m->kmax_small = nsweeps > MGW_KMAX ? nsweeps : 0;
    m->kmax = nsweeps > MGW_KMAX ? MGW_KMAX : nsweeps;
    int nl = 0;
    const int rc = mg_smooth_tiles(m, level, nsweeps, prolong != 0, row0, row1, &nl);
    m->kmax_small = ks;
    m->kmax = km;
    PYRO_REQUIRE(rc != 0 || nl == 1, "the window is too short for one launch of that many iterations");
    m->corners_stale[level] = true;
    PYRO_CHECK_HIP(hipGetLastError());
    return rc;
}

// the per-cycle diagnostics of solve() (MG.py:670-686) over rows [row0, row1] of the finest
// level: sums[0] = sum of ((v - old) / (v + small))^2, sums[1] = sum of r^2 with
// r = f - (alpha - beta L) v (one halo row of v on either side must be current); old <- v on
// those rows.  The caller adds the slabs' sums (all-reduce) and takes the norms.
int pyrohip_mg_diag_rows(pyrohip_mg *m, int row0, int row1, double *sums)
{
    PYRO_REQUIRE(m && sums, "NULL argument");
    PYRO_REQUIRE(!m->vc, "row windows: constant coefficients");
    pyrohip_ctx *c = m->ctx;
    const int Lf = m->nlevels - 1;
    MGLevel &F = m->lev[Lf];
    PYRO_REQUIRE(row0 >= 1 && row1 <= F.n && row0 <= row1, "rows outside the level");
    const int nrows = row1 - row0 + 1;
    const int gx = F.n >= 4096 ? 16 : (F.n >= 256 ? F.n / 256 : 1);
    const int gy = nrows >= 64 ? (2048 / gx < nrows ? 2048 / gx : nrows) : 1;
    const dim3 grid(gx, gy), block(256);
    const int nb = grid.x * grid.y;
    PYRO_TRY(c->reduce.ensure((2 * nb + 4) * sizeof(double)));
    double *part = (double *)c->reduce.p;
    PYRO_LAUNCH(c, "k_mg_solve_diag", (k_mg_solve_diag<true, true>), grid, block, 0,
                (const double *)F.v, (const double *)F.f, F.r, m->old_phi, F.n, F.pitch, m->alpha,
                m->beta, F.dx * F.dx, 1.e-16, part, row0, row1);
    // r holds the residual on the rows of the window only: a whole-level consumer (get r, the
    // residual norm, the restriction) must not take the rows outside it for current
    if (row0 == 1 && row1 == F.n) m->r_stale[Lf] = false;
    hipLaunchKernelGGL(k_sum_final2, dim3(1), dim3(256), 0, c->stream, (const double *)part, nb,
                       part + 2 * nb);
    PYRO_CHECK_HIP(hipGetLastError());
    PYRO_CHECK_HIP(hipMemcpyAsync(c->reduce_host, part + 2 * nb, 2 * sizeof(double),
                                  hipMemcpyDeviceToHost, c->stream));
    PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));
    sums[0] = ((double *)c->reduce_host)[0];
    sums[1] = ((double *)c->reduce_host)[1];
    return 0;
}

// old <- v on the finest level (MG.py:647: the copy solve() keeps for relative_error)
int pyrohip_mg_save_old(pyrohip_mg *m)
{
    PYRO_REQUIRE(m, "NULL mg");
    pyrohip_ctx *c = m->ctx;
    MGLevel &F = m->lev[m->nlevels - 1];
    PYRO_CHECK_HIP(hipMemcpyAsync(m->old_phi, F.v, (size_t)(F.n + 2) * F.pitch * sizeof(double),
                                  hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

int pyrohip_mg_residual_restrict_rows(pyrohip_mg *m, int fine, int crow0, int crow1)
{
    MG_CHECK_LEVEL(m, fine);
    PYRO_REQUIRE(fine >= 1 && !m->vc, "needs a coarser level, constant coefficients");
    MGLevel &F = m->lev[fine], &Cc = m->lev[fine - 1];
    PYRO_REQUIRE(crow0 >= 1 && crow1 <= Cc.n && crow0 <= crow1, "coarse rows outside the level");
    const int bx = (Cc.n >= 256) ? 256 : 64;
    PYRO_LAUNCH(m->ctx, "k_mg_residual_restrict", k_mg_residual_restrict<true>,
                dim3((Cc.n + bx - 1) / bx, crow1 - crow0 + 1), dim3(bx), 0, (const double *)F.v,
                (const double *)F.f, F.r, F.pitch, Cc.f, Cc.pitch, Cc.n, m->alpha, m->beta,
                F.dx * F.dx, crow0 - 1);
    PYRO_CHECK_HIP(hipGetLastError());
    return 0;
}

// rows [i0, i0 + ni) of an (n+2, n+2) level array (ghost rows / columns included)
int pyrohip_mg_get_rows(pyrohip_mg *m, int level, int var, int i0, int ni, double *host)
{
    MG_CHECK_LEVEL(m, level);
    MGLevel &L = m->lev[level];
    PYRO_REQUIRE(var >= 0 && var <= 2 && host, "bad var / NULL host");
    PYRO_REQUIRE(i0 >= 0 && ni >= 1 && i0 + ni <= L.n + 2, "rows outside the array");
    const int q = L.n + 2;
    PYRO_CHECK_HIP(hipMemcpy2DAsync(host, q * sizeof(double),
                                    plane(m, level, var) + (size_t)i0 * L.pitch,
                                    L.pitch * sizeof(double), q * sizeof(double), ni,
                                    hipMemcpyDeviceToHost, m->ctx->stream));
    PYRO_CHECK_HIP(hipStreamSynchronize(m->ctx->stream));
    return 0;
}

int pyrohip_mg_set_rows(pyrohip_mg *m, int level, int var, int i0, int ni, const double *host)
{
    MG_CHECK_LEVEL(m, level);
    MGLevel &L = m->lev[level];
    PYRO_REQUIRE(var >= 0 && var <= 2 && host, "bad var / NULL host");
    PYRO_REQUIRE(i0 >= 0 && ni >= 1 && i0 + ni <= L.n + 2, "rows outside the array");
    const int q = L.n + 2;
    if (var == 0 && m->v_is_zero[level]) {   // rows are about to be written: materialise the zeros
        PYRO_TRY(mg_zero(m, level, 0));
        m->v_is_zero[level] = false;
    }
    PYRO_CHECK_HIP(hipMemcpy2DAsync(plane(m, level, var) + (size_t)i0 * L.pitch,
                                    L.pitch * sizeof(double), host, q * sizeof(double),
                                    q * sizeof(double), ni, hipMemcpyHostToDevice,
                                    m->ctx->stream));
    PYRO_CHECK_HIP(hipStreamSynchronize(m->ctx->stream));
    return 0;
}

// the solution of `level` counts as zero from here on (the lazily zeroed coarse
// solutions of MG.py:658-659: the first smoothing launch takes v = 0 while staging)
int pyrohip_mg_mark_zero(pyrohip_mg *m, int level)
{
    MG_CHECK_LEVEL(m, level);
    MGLevel &L = m->lev[level];
    if (!m->vc && m->smoother != 0 && m->nsmooth > 0 &&
        ((L.n + 2) * (L.n + 2) > MGS_CELLS || (m->coarse_kernel && level <= MGC_TOP)))
        m->v_is_zero[level] = true;
    else
        PYRO_TRY(mg_zero(m, level, 0));
    return 0;
}

int pyrohip_mg_norm(pyrohip_mg *m, int level, int var, double *out)
{
    MG_CHECK_LEVEL(m, level);
    PYRO_REQUIRE(var >= 0 && var <= 2 && out, "bad var / NULL out");
    double s = 0.0;
    PYRO_TRY(mg_sumsq(m, plane(m, level, var), nullptr, level, 0, &s));
    const double dx = m->lev[level].dx;
    *out = sqrt(dx * dx * s);   // array_indexer.py:104-111
    return 0;
}

int pyrohip_mg_vcycle(pyrohip_mg *m, int level)
{
    MG_CHECK_LEVEL(m, level);
    PYRO_TRY(mg_vcycle(m, level));
    PYRO_CHECK_HIP(hipGetLastError());
    return 0;
}

int pyrohip_mg_init_rhs_norm(pyrohip_mg *m, double *source_norm)
{
    PYRO_REQUIRE(m, "NULL mg");
    double nrm = 0.0;
    PYRO_TRY(pyrohip_mg_norm(m, m->nlevels - 1, 1, &nrm));
    m->source_norm = nrm;
    if (source_norm) *source_norm = nrm;
    return 0;
}

int pyrohip_mg_set_coeffs(pyrohip_mg *m, const double *coeffs, const int *coeffs_bc)
{
    PYRO_REQUIRE(m && coeffs && coeffs_bc, "NULL argument");
    pyrohip_ctx *c = m->ctx;
    PYRO_CHECK_HIP(hipSetDevice(c->device));
    if (!m->vc_pool) {
        size_t total = 16;
        for (int l = 0; l < m->nlevels; l++) {
            Geom g = make_geom(m->lev[l].n, m->lev[l].n, 1);
            total += 3 * g.plane + 16;
        }
        PYRO_CHECK_HIP(hipMalloc((void **)&m->vc_pool, total * sizeof(double)));
        PYRO_CHECK_HIP(hipMemsetAsync(m->vc_pool, 0, total * sizeof(double), c->stream));
        double *p = m->vc_pool;
        for (int l = 0; l < m->nlevels; l++) {
            Geom g = make_geom(m->lev[l].n, m->lev[l].n, 1);
            MGLevel &L = m->lev[l];
            L.c = p + geom_lead(g); p += g.plane;
            L.ex = p + geom_lead(g); p += g.plane;
            L.ey = p + geom_lead(g); p += g.plane;
            p += 16;
        }
    }
    MGBC cbc;
    for (int s = 0; s < 4; s++) { cbc.code[s] = coeffs_bc[s]; cbc.val[s] = nullptr; }
    const int Lf = m->nlevels - 1;
    {   // finest: c.v() = coeffs.v(); fill_BC; EdgeCoeffs (variable_coeff_MG.py:72-84)
        MGLevel &F = m->lev[Lf];
        const int q = F.n + 2;
        PYRO_CHECK_HIP(hipMemcpy2DAsync(F.c, F.pitch * sizeof(double), coeffs, q * sizeof(double),
                                        q * sizeof(double), q, hipMemcpyHostToDevice, c->stream));
    }
    for (int l = Lf; l >= 0; l--) {
        MGLevel &L = m->lev[l];
        if (l < Lf) {   // coeffs_c.v() = f_patch.restrict("coeffs").v()  (:86-93)
            MGLevel &F = m->lev[l + 1];
            const int bx = (L.n >= 256) ? 256 : 64;
            hipLaunchKernelGGL(k_mg_restrict, dim3((L.n + bx - 1) / bx, L.n), dim3(bx), 0,
                               c->stream, (const double *)F.c, F.pitch, L.c, L.pitch, L.n);
        }
        const int nt = L.n + 2;
        hipLaunchKernelGGL(k_mg_fill_x, dim3((nt + 255) / 256), dim3(256), 0, c->stream, L.c, L.n,
                           L.pitch, L.dx, cbc);
        hipLaunchKernelGGL(k_mg_fill_y, dim3((nt + 255) / 256), dim3(256), 0, c->stream, L.c, L.n,
                           L.pitch, L.dx, cbc);
        if (l == Lf) {
            hipLaunchKernelGGL(k_vc_edges, dim3((L.n + 1 + 63) / 64, L.n + 1), dim3(64), 0,
                               c->stream, (const double *)L.c, L.ex, L.ey, L.n, L.pitch,
                               L.dx * L.dx);
        } else {
            MGLevel &F = m->lev[l + 1];
            hipLaunchKernelGGL(k_vc_edges_restrict, dim3((L.n + 1 + 63) / 64, L.n + 1), dim3(64), 0,
                               c->stream, (const double *)F.ex, (const double *)F.ey, F.pitch,
                               L.ex, L.ey, L.pitch, L.n, F.dx * F.dx, L.dx * L.dx);
        }
    }
    PYRO_CHECK_HIP(hipGetLastError());
    PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));
    m->vc = 1;
    return 0;
}

int pyrohip_mg_set_general_coeffs(pyrohip_mg *m, const double *alpha, const double *beta,
                                  const double *gamma_x, const double *gamma_y,
                                  const int *coeffs_bc)
{
    PYRO_REQUIRE(m && alpha && beta && gamma_x && gamma_y && coeffs_bc, "NULL argument");
    // beta: cell values restricted down the hierarchy, then onto the edges,
    // exactly like the variable-coefficient solver (general_MG.py:84-105)
    PYRO_TRY(pyrohip_mg_set_coeffs(m, beta, coeffs_bc + 4));
    pyrohip_ctx *c = m->ctx;
    if (!m->gen_pool) {
        size_t total = 16;
        for (int l = 0; l < m->nlevels; l++) total += 3 * make_geom(m->lev[l].n, m->lev[l].n, 1).plane + 16;
        PYRO_CHECK_HIP(hipMalloc((void **)&m->gen_pool, total * sizeof(double)));
        PYRO_CHECK_HIP(hipMemsetAsync(m->gen_pool, 0, total * sizeof(double), c->stream));
        double *p = m->gen_pool;
        for (int l = 0; l < m->nlevels; l++) {
            Geom g = make_geom(m->lev[l].n, m->lev[l].n, 1);
            MGLevel &L = m->lev[l];
            L.a = p + geom_lead(g); p += g.plane;
            L.gx = p + geom_lead(g); p += g.plane;
            L.gy = p + geom_lead(g); p += g.plane;
            p += 16;
        }
    }
    const int Lf = m->nlevels - 1;
    const double *src[3] = {alpha, gamma_x, gamma_y};
    const int *sbc[3] = {coeffs_bc, coeffs_bc + 8, coeffs_bc + 12};
    for (int w = 0; w < 3; w++) {   // general_MG.py:66-82
        MGBC cbc;
        for (int s = 0; s < 4; s++) { cbc.code[s] = sbc[w][s]; cbc.val[s] = nullptr; }
        auto arr = [&](int l) { MGLevel &L = m->lev[l]; return w == 0 ? L.a : w == 1 ? L.gx : L.gy; };
        {
            MGLevel &F = m->lev[Lf];
            const int q = F.n + 2;
            PYRO_CHECK_HIP(hipMemcpy2DAsync(arr(Lf), F.pitch * sizeof(double), src[w],
                                            q * sizeof(double), q * sizeof(double), q,
                                            hipMemcpyHostToDevice, c->stream));
        }
        for (int l = Lf; l >= 0; l--) {
            MGLevel &L = m->lev[l];
            if (l < Lf) {
                MGLevel &F = m->lev[l + 1];
                const int bx = (L.n >= 256) ? 256 : 64;
                hipLaunchKernelGGL(k_mg_restrict, dim3((L.n + bx - 1) / bx, L.n), dim3(bx), 0,
                                   c->stream, (const double *)arr(l + 1), F.pitch, arr(l), L.pitch,
                                   L.n);
            }
            const int nt = L.n + 2;
            hipLaunchKernelGGL(k_mg_fill_x, dim3((nt + 255) / 256), dim3(256), 0, c->stream, arr(l),
                               L.n, L.pitch, L.dx, cbc);
            hipLaunchKernelGGL(k_mg_fill_y, dim3((nt + 255) / 256), dim3(256), 0, c->stream, arr(l),
                               L.n, L.pitch, L.dx, cbc);
        }
    }
    PYRO_CHECK_HIP(hipGetLastError());
    PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));
    m->vc = 2;
    return 0;
}

int pyrohip_mg_set_rhs_cn(pyrohip_mg *m, pyrohip_state *s, int n, double coef,
                          double *source_norm)
{
    PYRO_REQUIRE(m && s, "NULL argument");
    PYRO_REQUIRE(n >= 0 && n < s->nvar, "variable index out of range");
    PYRO_REQUIRE(s->ctx == m->ctx, "state and multigrid live on different contexts");
    PYRO_REQUIRE(s->g.ng == 1 && s->g.nx == m->nx && s->g.ny == m->nx,
                 "state must be nx x nx with ng = 1 like the finest multigrid level");
    MGLevel &F = m->lev[m->nlevels - 1];
    PYRO_REQUIRE(F.pitch == s->g.pitch, "pitch mismatch");
    const int bx = (F.n >= 256) ? 256 : 64;
    hipLaunchKernelGGL(k_mg_rhs_cn, dim3((F.n + bx - 1) / bx, F.n), dim3(bx), 0, m->ctx->stream,
                       (const double *)(s->d + (size_t)n * s->g.plane), F.f, F.n, F.pitch, coef,
                       F.dx * F.dx, F.dx * F.dx);
    PYRO_CHECK_HIP(hipGetLastError());
    return pyrohip_mg_init_rhs_norm(m, source_norm);
}

int pyrohip_mg_copy_solution(pyrohip_mg *m, pyrohip_state *s, int n)
{
    PYRO_REQUIRE(m && s, "NULL argument");
    PYRO_REQUIRE(n >= 0 && n < s->nvar, "variable index out of range");
    PYRO_REQUIRE(s->ctx == m->ctx, "state and multigrid live on different contexts");
    PYRO_REQUIRE(s->g.ng == 1 && s->g.nx == m->nx && s->g.ny == m->nx,
                 "state must be nx x nx with ng = 1 like the finest multigrid level");
    MGLevel &F = m->lev[m->nlevels - 1];
    const int bx = (F.n >= 256) ? 256 : 64;
    hipLaunchKernelGGL(k_mg_copy_interior, dim3((F.n + bx - 1) / bx, F.n), dim3(bx), 0,
                       m->ctx->stream, (const double *)F.v, s->d + (size_t)n * s->g.plane, F.n,
                       F.pitch);
    PYRO_CHECK_HIP(hipGetLastError());
    s->next_cfl_min = -1.0;
    return 0;
}

int pyrohip_mg_solve(pyrohip_mg *m, double rtol, int max_cycles, int *num_cycles,
                     double *residual_error, double *relative_error)
{
    PYRO_REQUIRE(m, "NULL mg");
    pyrohip_ctx *c = m->ctx;
    const int Lf = m->nlevels - 1;
    MGLevel &F = m->lev[Lf];
    const size_t fbytes = (size_t)(F.n + 2) * F.pitch * sizeof(double);
    // the solution before the first cycle, for its relative change -- unless the cycle's first
    // smoothing launch leaves it behind anyway (capture_old: the finest level's launches read
    // one buffer and write another; 87 us of copy per solve at 4096^2)
    const bool first_launch_keeps_old = !m->vc && m->smoother != 0 && m->nsmooth > 0 && Lf > MGC_TOP &&
                                        (F.n + 2) * (F.n + 2) > MGS_CELLS;
    if (!first_launch_keeps_old)
        PYRO_CHECK_HIP(hipMemcpyAsync(m->old_phi, F.v, fbytes, hipMemcpyDeviceToDevice, c->stream));
    // (the fourth buffer: wherever a cycle's sums may ride on the marching launch)
    if (first_launch_keeps_old && m->lazy_r && m->march_tail && m->march_min > 0 && F.n >= m->march_min &&
        !m->older) {
        const Geom gf = make_geom(F.n, F.n, 1);
        PYRO_CHECK_HIP(hipMalloc((void **)&m->older_base, (gf.plane + 16) * sizeof(double)));
        PYRO_CHECK_HIP(hipMemsetAsync(m->older_base, 0, (gf.plane + 16) * sizeof(double), c->stream));
        m->older = m->older_base + geom_lead(gf);
    }
    double res = 1.e33, rel = 1.e33;
    int cycle = 1;
    // one cycle on the stream: zeroed coarse solutions, V-cycle, both norms.  Constant
    // coefficients: the norms land in host slot `slot` when event ev[slot] has passed;
    // variable coefficients (sync = true): the old blocking sequence, norms returned at once
    auto enqueue = [&](int slot, bool sync, double *s_out, double *s2_out) -> int {
        for (int l = 0; l < Lf; l++) {                    // :658-659 (zero the coarse solutions)
            // levels the wide tile smoother visits first: no memset (hipMemset runs
            // at ~270 GB/s: 123 us for the 2048^2 level), the staging takes v = 0
            const MGLevel &Lc = m->lev[l];
            const bool lazy = !m->vc && m->smoother != 0 && m->nsmooth > 0 &&
                              ((Lc.n + 2) * (Lc.n + 2) > MGS_CELLS ||
                               (m->coarse_kernel && l <= MGC_TOP));   // staged as 0 there
            if (lazy) m->v_is_zero[l] = true;
            else PYRO_TRY(mg_zero(m, l, 0));
        }
        m->in_solve = true;
        m->capture_old = !m->vc;
        m->old_captured = false;
        m->diag_req = !sync && m->lazy_r;   // the sums below on the tail of the last launch
        m->diag_done = false;
        const int vrc = mg_vcycle(m, Lf);
        m->in_solve = false;
        m->capture_old = false;
        m->diag_req = false;
        m->lazy_rel[slot & 1] = m->diag_done && !sync;
        PYRO_TRY(vrc);
        PYRO_REQUIRE(!first_launch_keeps_old || m->old_captured,
                     "internal: the cycle did not leave the solution before it behind");
        if (sync) {                                       // :673-678
            PYRO_TRY(mg_sumsq(m, F.v, m->old_phi, Lf, 1, s_out));
            PYRO_CHECK_HIP(hipMemcpyAsync(m->old_phi, F.v, fbytes, hipMemcpyDeviceToDevice,
                                          c->stream));
            PYRO_TRY(mg_residual(m, Lf));
            PYRO_TRY(mg_sumsq(m, F.r, nullptr, Lf, 0, s2_out));
            return 0;
        }
        // fused: relative change, old <- v (unless captured), residual and its norm in one pass
        // a workgroup covers 256 columns of a row: as many workgroups across as the row has such
        // pieces (16 across on a 2048^2 level left half of them idle: 31 -> 21 us), ~2048 in all
        const int gx = F.n >= 4096 ? 16 : (F.n >= 256 ? F.n / 256 : 1);
        const int gy = F.n >= 64 ? (2048 / gx < F.n ? 2048 / gx : F.n) : 1;
        const dim3 grid(gx, gy), block(256);
        const int nb = m->diag_done ? m->diag_nb : grid.x * grid.y;
        PYRO_TRY(c->reduce.ensure((2 * nb + 4) * sizeof(double)));
        double *part = (double *)c->reduce.p;
        PYRO_REQUIRE(!m->diag_done || part == m->diag_part, "internal: the partial sums moved");
        using DiagT = void (*)(const double *, const double *, double *, double *, int, int, double,
                               double, double, double, double *, int, int);
        static const DiagT diag[2][2] = {{k_mg_solve_diag<false, false>, k_mg_solve_diag<false, true>},
                                        {k_mg_solve_diag<true, false>, k_mg_solve_diag<true, true>}};
        const bool store = !m->lazy_r;
        if (!m->diag_done)
            PYRO_LAUNCH(c, "k_mg_solve_diag", diag[store ? 1 : 0][m->old_captured ? 0 : 1], grid, block, 0,
                        (const double *)F.v, (const double *)F.f, F.r, m->old_phi, F.n, F.pitch, m->alpha,
                        m->beta, F.dx * F.dx, 1.e-16, part, 1, F.n);
        m->r_stale[Lf] = !store;
        hipLaunchKernelGGL(k_sum_final2, dim3(1), dim3(256), 0, c->stream,
                           (const double *)part, nb, part + 2 * nb + 2 * slot);
        PYRO_CHECK_HIP(hipGetLastError());
        PYRO_CHECK_HIP(hipMemcpyAsync((double *)c->reduce_host + 2 * slot, part + 2 * nb + 2 * slot,
                                      2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        if (!m->ev[slot]) PYRO_CHECK_HIP(hipEventCreateWithFlags(&m->ev[slot], hipEventDisableTiming));
        PYRO_CHECK_HIP(hipEventRecord(m->ev[slot], c->stream));
        return 0;
    };
    // Between a cycle's norms and the decision they feed (MG.py:652) the device would idle for
    // the read-back and the next launch (~13 us per cycle, measured).  While that answer is on
    // its way the NEXT cycle is already put on the stream when it is likely to be needed; the
    // first smoothing launch of a cycle leaves the solution before the cycle untouched in
    // old_phi (capture_old), so a cycle that turns out to be one too many is undone by taking
    // that buffer back -- cycles, norms and solution are those of the loop that waits.  (The
    // coarser levels' arrays, scratch between solves, then hold the undone cycle's values.)
    // pyrohip_mg_tuning.speculate: 0 never, 2 whenever a further cycle is allowed (tests).
    const int spec_mode = m->speculate;
    const bool can_undo = first_launch_keeps_old;          // the finest level's launches ping-pong
    double res_prev = -1.0, res_pprev = -1.0;
    bool pending = false;                                  // cycle `cycle` is already on the stream
    bool rel_pending = false;                              // the last cycle left relative_error to the end
    const bool spec_debug = m->spec_debug;   // developer aid
    int n_spec = 0, n_undo = 0;
    while (res > rtol && cycle <= max_cycles) {           // MG.py:652
        double s = 0.0, s2 = 0.0;
        if (m->vc) {
            PYRO_TRY(enqueue(0, true, &s, &s2));
        } else {
            const int slot = cycle & 1;
            if (!pending) PYRO_TRY(enqueue(slot, false, nullptr, nullptr));
            pending = false;
            // would the loop run another cycle if this one's residual were what the trend says?
            bool spec = false;
            if (can_undo && spec_mode != 0 && cycle + 1 <= max_cycles) {
                const double ratio = (res_pprev > 0.0 && res_prev > 0.0) ? fmin(1.0, fmax(0.02, res_prev / res_pprev)) : 0.05;
                const double guess = (res_prev > 0.0) ? res_prev * ratio : 1.e33;
                spec = spec_mode == 2 || rtol <= 0.0 || guess > 3.0 * rtol;
            }
            double *undo_v = nullptr;
            if (spec) {
                n_spec++;
                PYRO_TRY(enqueue(slot ^ 1, false, nullptr, nullptr));
                PYRO_REQUIRE(m->old_captured, "internal: a speculative cycle must leave the solution before it");
                undo_v = m->old_phi;                       // the solution after cycle `cycle`
            }
            PYRO_CHECK_HIP(hipEventSynchronize(m->ev[slot]));
            s = ((double *)c->reduce_host)[2 * slot];
            s2 = ((double *)c->reduce_host)[2 * slot + 1];
            if (spec) {
                const double rn_ = sqrt(F.dx * F.dx * s2);
                const double res_ = (m->source_norm != 0.0) ? rn_ / m->source_norm : rn_;
                if (res_ > rtol) pending = true;           // it was needed
                else {                                     // undo: the buffers trade places again
                    n_undo++;
                    PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));
                    if (m->older) {                         // ... and the solution before THAT cycle is back
                        double *spec_result = F.v;
                        F.v = undo_v;
                        m->old_phi = m->older;
                        m->older = spec_result;
                    } else {
                        m->old_phi = F.v;
                        F.v = undo_v;
                    }
                    m->r_stale[Lf] = true;
                    m->corners_stale[Lf] = true;
                }
            }
        }
        rel = sqrt(F.dx * F.dx * s);
        rel_pending = !m->vc && m->lazy_rel[cycle & 1];
        double rn = sqrt(F.dx * F.dx * s2);
        res = (m->source_norm != 0.0) ? rn / m->source_norm : rn;   // :682-685
        res_pprev = res_prev; res_prev = res;
        cycle++;
    }
    if (rel_pending) {       // MG.py:673-678 for the cycle that ended the loop: F.v against old_phi
        double s = 0.0;
        PYRO_TRY(mg_sumsq(m, F.v, m->old_phi, Lf, 1, &s));
        rel = sqrt(F.dx * F.dx * s);
    }
    PYRO_TRY(mg_fill(m, Lf, 0));                          // :697
    m->corners_stale[Lf] = false;
    PYRO_CHECK_HIP(hipGetLastError());
    PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));
    if (spec_debug)
        fprintf(stderr, "mg solve n=%d rtol=%g: %d cycles, %d launched ahead, %d undone, residual %g (before: %g)\n",
                F.n, rtol, cycle - 1, n_spec, n_undo, res, res_pprev);
    if (num_cycles) *num_cycles = cycle - 1;
    if (residual_error) *residual_error = res;
    if (relative_error) *relative_error = rel;
    return 0;
}

}  // extern "C"

// ---- mg_internal.h ----------------------------------------------------------
namespace pyro {

int mg_finest(pyrohip_mg *m, MgFinest *out)
{
    PYRO_REQUIRE(m && out, "NULL argument");
    const int Lf = m->nlevels - 1;
    const MGLevel &F = m->lev[Lf];
    *out = MgFinest{m->ctx, Lf, F.n, F.pitch, F.dx, F.v, F.f, plane(m, Lf, 2)};
    return 0;
}

int mg_rows_ptr(pyrohip_mg *m, int level, int var, int i0, int ni, double **ptr, int *pitch,
                pyrohip_ctx **ctx)
{
    MG_CHECK_LEVEL(m, level);
    MGLevel &L = m->lev[level];
    PYRO_REQUIRE(var >= 0 && var <= 2, "bad var");
    PYRO_REQUIRE(i0 >= 0 && ni >= 1 && i0 + ni <= L.n + 2, "rows outside the array");
    if (var == 0 && m->v_is_zero[level]) {
        PYRO_TRY(mg_zero(m, level, 0));
        m->v_is_zero[level] = false;
    }
    *ptr = plane(m, level, var) + (size_t)i0 * L.pitch;
    *pitch = L.pitch;
    *ctx = m->ctx;
    return 0;
}

int mg_solution_written(pyrohip_mg *m)
{
    PYRO_REQUIRE(m, "NULL mg");
    m->corners_stale[m->nlevels - 1] = false;
    return 0;
}

int mg_solution_ghosts(pyrohip_mg *m)
{
    PYRO_REQUIRE(m, "NULL mg");
    const int Lf = m->nlevels - 1;
    if (m->corners_stale[Lf]) {
        PYRO_TRY(mg_fill(m, Lf, 0));
        m->corners_stale[Lf] = false;
        PYRO_CHECK_HIP(hipGetLastError());
    }
    return 0;
}

}  // namespace pyro
