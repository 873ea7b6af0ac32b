// Burgers and incompressible solvers: the unsplit CTU predictor for (u, v) and
// the pieces of the approximate-projection step that sit between the two
// multigrid solves (north_star: "the multigrid
// V-cycle that the diffusion and incompressible solvers sit on").
//
//   pyro/burgers/burgers_interface.py:4-312   edge states, transverse terms,
//                                             riemann / upwind
//   pyro/incompressible/incomp_interface.py   mac_vels, states, gradp terms
//   pyro/incompressible/simulation.py:200-330 evolve
//
// All arrays are planes of the solver's pyrohip_state (ng = 4) or of its work
// area; the elliptic solves run in the pyrohip_mg (ng = 1) that is passed in,
// and the right-hand sides / guesses / solutions move between the two on the
// device.  The reference evaluates everything on the valid region grown by 2
// ("B2"); the values that reach the interior only need the region grown by 1
// ("B1"), which is what is computed here (edge states of a B1 cell need slopes
// of B1 cells only, so the limiter's lda2 = 0 rule outside B2 never enters).
// Compiled with -ffp-contract=off, reference operation order: bit-identical to
// the oracle.
#include "common.h"
#include "mg_internal.h"
#include "stencil.h"

namespace pyro {

// work planes
enum {
    W_UXL, W_UXR, W_UYL, W_UYR, W_VXL, W_VXR, W_VYL, W_VYR,   // uncorrected ("hat") states
    C_UXL, C_UXR, C_UYL, C_UYR, C_VXL, C_VXR, C_VYL, C_VYR,   // + transverse + grad p terms
    W_UMAC, W_VMAC, W_ADVX, W_ADVY,
    W_NPL
};

// burgers_interface.py:265-290
__device__ __forceinline__ double bg_riemann(double ql, double qr)
{
    if (ql <= 0.0 && qr >= 0.0) return 0.0;
    return (ql > 0.0 && ql + qr > 0.0) ? ql : qr;
}
// burgers_interface.py:236-262
__device__ __forceinline__ double bg_upwind(double ql, double qr, double s)
{
    if (s == 0.0) return 0.5 * (ql + qr);
    return (s > 0.0) ? ql : qr;
}

struct BP {   // kernel parameters
    double dx, dy, dt, dtdx, dtdy;
    int limiter;
    double nu;   // > 0: viscous source in the predictor (incompressible_viscous)
    double eps;  // != 0: diffusion correction of the uncorrected states (burgers_viscous)
};

// one thread per cell of B1
#define BG_CELL_B1()                                                   \
    const int j = g.jlo - 1 + blockIdx.x * blockDim.x + threadIdx.x;   \
    const int i = g.ilo - 1 + blockIdx.y;                              \
    if (j > g.jhi + 1) return;                                         \
    const int p = g.pitch;                                             \
    const size_t pl = g.plane;                                         \
    const size_t k = (size_t)i * p + j;

// get_interface_states, burgers_interface.py:4-86 (cells of B1)
__global__ __launch_bounds__(256) void k_bg_hat(const double *__restrict__ u,
                                                const double *__restrict__ v,
                                                double *__restrict__ W, Geom g, BP P)
{
    BG_CELL_B1()
    const double uc = u[k], vc = v[k];
    const double ldux = limited_slope(u[k - 2 * p], u[k - p], uc, u[k + p], u[k + 2 * p], P.limiter);
    const double ldvx = limited_slope(v[k - 2 * p], v[k - p], vc, v[k + p], v[k + 2 * p], P.limiter);
    const double lduy = limited_slope(u[k - 2], u[k - 1], uc, u[k + 1], u[k + 2], P.limiter);
    const double ldvy = limited_slope(v[k - 2], v[k - 1], vc, v[k + 1], v[k + 2], P.limiter);
    double uxl = uc + 0.5 * (1.0 - P.dtdx * uc) * ldux;
    double uxr = uc - 0.5 * (1.0 + P.dtdx * uc) * ldux;
    double vxl = vc + 0.5 * (1.0 - P.dtdx * uc) * ldvx;
    double vxr = vc - 0.5 * (1.0 + P.dtdx * uc) * ldvx;
    double uyl = uc + 0.5 * (1.0 - P.dtdy * vc) * lduy;
    double uyr = uc - 0.5 * (1.0 + P.dtdy * vc) * lduy;
    double vyl = vc + 0.5 * (1.0 - P.dtdy * vc) * ldvy;
    double vyr = vc - 0.5 * (1.0 + P.dtdy * vc) * ldvy;
    if (P.eps != 0.0) {   // apply_diffusion_corrections, burgers_viscous/interface.py:94-171
        const double dx2 = P.dx * P.dx, dy2 = P.dy * P.dy;
        const double lu = (u[k + p] - 2.0 * uc + u[k - p]) / dx2 + (u[k + 1] - 2.0 * uc + u[k - 1]) / dy2;
        const double lv = (v[k + p] - 2.0 * vc + v[k - p]) / dx2 + (v[k + 1] - 2.0 * vc + v[k - 1]) / dy2;
        const double cu = 0.5 * P.eps * P.dt * lu, cv = 0.5 * P.eps * P.dt * lv;
        uxl += cu; uyl += cu; uxr += cu; uyr += cu;
        vxl += cv; vyl += cv; vxr += cv; vyr += cv;
    }
    W[W_UXL * pl + k + p] = uxl;
    W[W_UXR * pl + k] = uxr;
    W[W_VXL * pl + k + p] = vxl;
    W[W_VXR * pl + k] = vxr;
    W[W_UYL * pl + k + 1] = uyl;
    W[W_UYR * pl + k] = uyr;
    W[W_VYL * pl + k + 1] = vyl;
    W[W_VYR * pl + k] = vyr;
}

// apply_transverse_corrections (burgers_interface.py:89-175) and
// apply_gradp_corrections (incomp_interface.py:139-183) for the states that
// belong to cell (i, j): x states for j interior, y states for i interior
// (the others would need hat states outside B1 and are never used).
// With P.nu > 0 the other source terms follow (apply_other_source_terms,
// incomp_interface.py:186-254): + dt/2 nu L(u) with the source evaluated on
// the interior only (incompressible_viscous/simulation.py:24-41).
__global__ __launch_bounds__(256) void k_bg_trans(const double *__restrict__ u,
                                                  const double *__restrict__ v,
                                                  const double *__restrict__ gpx,
                                                  const double *__restrict__ gpy,
                                                  double *__restrict__ W, Geom g, BP P)
{
    BG_CELL_B1()
    const double *uxl = W + W_UXL * pl, *uxr = W + W_UXR * pl, *uyl = W + W_UYL * pl,
                 *uyr = W + W_UYR * pl, *vxl = W + W_VXL * pl, *vxr = W + W_VXR * pl,
                 *vyl = W + W_VYL * pl, *vyr = W + W_VYR * pl;
    const bool jin = (j >= g.jlo && j <= g.jhi), iin = (i >= g.ilo && i <= g.ihi);
    double gx = 0.0, gy = 0.0;
    if (gpx) { gx = -0.5 * P.dt * gpx[k]; gy = -0.5 * P.dt * gpy[k]; }
    const bool src = (P.nu > 0.0) && jin && iin;
    double sx = 0.0, sy = 0.0;
    if (src) {
        const double dx2 = P.dx * P.dx, dy2 = P.dy * P.dy;
        const double lu = P.nu * ((u[k + p] + u[k - p] - 2.0 * u[k]) / dx2 +
                                  (u[k + 1] + u[k - 1] - 2.0 * u[k]) / dy2);
        const double lv = P.nu * ((v[k + p] + v[k - p] - 2.0 * v[k]) / dx2 +
                                  (v[k + 1] + v[k - 1] - 2.0 * v[k]) / dy2);
        sx = 0.5 * P.dt * lu;
        sy = 0.5 * P.dt * lv;
    }
    if (jin) {   // x states: transverse (y) derivative over the cell
        const double vh0 = bg_riemann(vyl[k], vyr[k]), vh1 = bg_riemann(vyl[k + 1], vyr[k + 1]);
        const double vbar = 0.5 * (vh0 + vh1);
        const double uy0 = bg_upwind(uyl[k], uyr[k], vh0), uy1 = bg_upwind(uyl[k + 1], uyr[k + 1], vh1);
        const double vy0 = bg_upwind(vyl[k], vyr[k], vh0), vy1 = bg_upwind(vyl[k + 1], vyr[k + 1], vh1);
        const double tu = -0.5 * P.dtdy * vbar * (uy1 - uy0);
        const double tv = -0.5 * P.dtdy * vbar * (vy1 - vy0);
        double a = uxl[k + p] + tu, b = uxr[k] + tu, c = vxl[k + p] + tv, d = vxr[k] + tv;
        if (gpx) { a += gx; b += gx; c += gy; d += gy; }
        if (src) { a += sx; b += sx; c += sy; d += sy; }
        W[C_UXL * pl + k + p] = a; W[C_UXR * pl + k] = b;
        W[C_VXL * pl + k + p] = c; W[C_VXR * pl + k] = d;
    }
    if (iin) {   // y states: transverse (x) derivative over the cell
        const double uh0 = bg_riemann(uxl[k], uxr[k]), uh1 = bg_riemann(uxl[k + p], uxr[k + p]);
        const double ubar = 0.5 * (uh0 + uh1);
        const double vx0 = bg_upwind(vxl[k], vxr[k], uh0), vx1 = bg_upwind(vxl[k + p], vxr[k + p], uh1);
        const double ux0 = bg_upwind(uxl[k], uxr[k], uh0), ux1 = bg_upwind(uxl[k + p], uxr[k + p], uh1);
        const double sv = -0.5 * P.dtdx * ubar * (vx1 - vx0);
        const double su = -0.5 * P.dtdx * ubar * (ux1 - ux0);
        double a = vyl[k + 1] + sv, b = vyr[k] + sv, c = uyl[k + 1] + su, d = uyr[k] + su;
        if (gpx) { a += gy; b += gy; c += gx; d += gx; }
        if (src) { a += sy; b += sy; c += sx; d += sx; }
        W[C_VYL * pl + k + 1] = a; W[C_VYR * pl + k] = b;
        W[C_UYL * pl + k + 1] = c; W[C_UYR * pl + k] = d;
    }
}

// one thread per cell of the interior grown by one on the HIGH side (faces)
#define BG_FACE()                                                      \
    const int j = g.jlo + blockIdx.x * blockDim.x + threadIdx.x;       \
    const int i = g.ilo + blockIdx.y;                                  \
    if (j > g.jhi + 1) return;                                         \
    const int p = g.pitch;                                             \
    const size_t pl = g.plane;                                         \
    const size_t k = (size_t)i * p + j;

// riemann_and_upwind (burgers_interface.py:293-312) on the faces the update
// needs: u_MAC on x faces [ilo, ihi+1] x [jlo, jhi], v_MAC on y faces
__global__ __launch_bounds__(256) void k_bg_mac(double *__restrict__ W, Geom g)
{
    BG_FACE()
    if (j <= g.jhi) {
        const double l = W[C_UXL * pl + k], r = W[C_UXR * pl + k];
        W[W_UMAC * pl + k] = bg_upwind(l, r, bg_riemann(l, r));
    }
    if (i <= g.ihi) {
        const double l = W[C_VYL * pl + k], r = W[C_VYR * pl + k];
        W[W_VMAC * pl + k] = bg_upwind(l, r, bg_riemann(l, r));
    }
}

// burgers update: construct_unsplit_fluxes (:178-233) + simulation.py:96-101
__global__ __launch_bounds__(256) void k_bg_update(double *__restrict__ u, double *__restrict__ v,
                                                   const double *__restrict__ W, Geom g, BP P)
{
    const int j = g.jlo + blockIdx.x * blockDim.x + threadIdx.x;
    const int i = g.ilo + blockIdx.y;
    if (j > g.jhi) return;
    const int p = g.pitch;
    const size_t pl = g.plane;
    const size_t k = (size_t)i * p + j;
    const double *um = W + W_UMAC * pl, *vm = W + W_VMAC * pl;
    auto fx = [&](int plane_l, int plane_r, size_t kk) {
        return 0.5 * bg_upwind(W[plane_l * pl + kk], W[plane_r * pl + kk], um[kk]) * um[kk];
    };
    auto fy = [&](int plane_l, int plane_r, size_t kk) {
        return 0.5 * bg_upwind(W[plane_l * pl + kk], W[plane_r * pl + kk], vm[kk]) * vm[kk];
    };
    u[k] = u[k] + P.dtdx * (fx(C_UXL, C_UXR, k) - fx(C_UXL, C_UXR, k + p)) +
           P.dtdy * (fy(C_UYL, C_UYR, k) - fy(C_UYL, C_UYR, k + 1));
    v[k] = v[k] + P.dtdx * (fx(C_VXL, C_VXR, k) - fx(C_VXL, C_VXR, k + p)) +
           P.dtdy * (fy(C_VYL, C_VYR, k) - fy(C_VYL, C_VYR, k + 1));
}

// ---- incompressible ------------------------------------------------------
// RHS of the MAC projection, incompressible/simulation.py:243-247, written
// into the finest multigrid level (interior); v of that level is zeroed
// elsewhere by the caller (init_zeros)
__global__ __launch_bounds__(256) void k_inc_div_mac(const double *__restrict__ W, Geom g,
                                                     double *__restrict__ f, int mpitch, BP P)
{
    const int jj = blockIdx.x * blockDim.x + threadIdx.x, ii = blockIdx.y;
    if (jj >= g.ny) return;
    const size_t k = (size_t)(g.ilo + ii) * g.pitch + g.jlo + jj;
    const double *um = W + W_UMAC * g.plane, *vm = W + W_VMAC * g.plane;
    f[(size_t)(ii + 1) * mpitch + jj + 1] =
        (um[k + g.pitch] - um[k]) / P.dx + (vm[k + 1] - vm[k]) / P.dy;
}

// phi-MAC <- solution on B1 (simulation.py:253-255)
__global__ __launch_bounds__(256) void k_inc_copy_b1(const double *__restrict__ mv, int mpitch,
                                                     double *__restrict__ dst, Geom g,
                                                     int zero_outside)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= g.qy) return;
    const bool in = (i >= g.ilo - 1 && i <= g.ihi + 1 && j >= g.jlo - 1 && j <= g.jhi + 1);
    const size_t k = (size_t)i * g.pitch + j;
    if (in) dst[k] = mv[(size_t)(i - g.ilo + 1) * mpitch + (j - g.jlo + 1)];
    else if (zero_outside) dst[k] = 0.0;
}

// MAC correction (:259-262) stored for the faces, then states() with the MAC
// velocities (incomp_interface.py:66-136), the advective terms (:286-296)
// and the provisional velocity update (:298-304).  Two kernels: the face
// velocities are shared by neighbouring cells.
__global__ __launch_bounds__(256) void k_inc_mac_project(double *__restrict__ W,
                                                         const double *__restrict__ phiM, Geom g,
                                                         BP P)
{
    BG_FACE()
    (void)pl;
    if (j <= g.jhi) W[W_UMAC * g.plane + k] -= (phiM[k] - phiM[k - p]) / P.dx;
    if (i <= g.ihi) W[W_VMAC * g.plane + k] -= (phiM[k] - phiM[k - 1]) / P.dy;
}

__global__ __launch_bounds__(256) void k_inc_advect(double *__restrict__ u, double *__restrict__ v,
                                                    const double *__restrict__ gpx,
                                                    const double *__restrict__ gpy,
                                                    double *__restrict__ W, Geom g, BP P,
                                                    int proj_type)
{
    const int j = g.jlo + blockIdx.x * blockDim.x + threadIdx.x;
    const int i = g.ilo + blockIdx.y;
    if (j > g.jhi) return;
    const int p = g.pitch;
    const size_t pl = g.plane;
    const size_t k = (size_t)i * p + j;
    const double *um = W + W_UMAC * pl, *vm = W + W_VMAC * pl;
    auto xs = [&](int L, int R, size_t kk) { return bg_upwind(W[L * pl + kk], W[R * pl + kk], um[kk]); };
    auto ys = [&](int L, int R, size_t kk) { return bg_upwind(W[L * pl + kk], W[R * pl + kk], vm[kk]); };
    const double ubar = 0.5 * (um[k] + um[k + p]), vbar = 0.5 * (vm[k] + vm[k + 1]);
    const double ax = ubar * (xs(C_UXL, C_UXR, k + p) - xs(C_UXL, C_UXR, k)) / P.dx +
                      vbar * (ys(C_UYL, C_UYR, k + 1) - ys(C_UYL, C_UYR, k)) / P.dy;
    const double ay = ubar * (xs(C_VXL, C_VXR, k + p) - xs(C_VXL, C_VXR, k)) / P.dx +
                      vbar * (ys(C_VYL, C_VYR, k + 1) - ys(C_VYL, C_VYR, k)) / P.dy;
    W[W_ADVX * pl + k] = ax;
    W[W_ADVY * pl + k] = ay;
    if (proj_type == 0) return;   // viscous: the parabolic solves do the update
    if (proj_type == 1) {
        u[k] -= (P.dt * ax + P.dt * gpx[k]);
        v[k] -= (P.dt * ay + P.dt * gpy[k]);
    } else {
        u[k] -= P.dt * ax;
        v[k] -= P.dt * ay;
    }
}

// cell-centred divergence (:318-319 and :99-100) -> f, optionally / dt;
// guess = phi on B1 (:322-324) or zeros (init_zeros)
__global__ __launch_bounds__(256) void k_inc_div_cc(const double *__restrict__ u,
                                                    const double *__restrict__ v,
                                                    const double *__restrict__ phi, Geom g,
                                                    double *__restrict__ f, double *__restrict__ mv,
                                                    int mpitch, BP P, int divide_by_dt)
{
    const int jj = blockIdx.x * blockDim.x + threadIdx.x, ii = blockIdx.y;   // MG indices
    if (jj > g.ny + 1) return;
    const size_t mk = (size_t)ii * mpitch + jj;
    const size_t k = (size_t)(g.ilo + ii - 1) * g.pitch + g.jlo + jj - 1;
    mv[mk] = phi ? phi[k] : 0.0;
    if (ii >= 1 && ii <= g.nx && jj >= 1 && jj <= g.ny) {
        double d = 0.5 * (u[k + g.pitch] - u[k - g.pitch]) / P.dx + 0.5 * (v[k + 1] - v[k - 1]) / P.dy;
        if (divide_by_dt) d = d / P.dt;
        f[mk] = d;
    }
}

// incompressible_viscous/simulation.py:113-135 (and :152-170 for v): RHS and
// guess of the Helmholtz solve of one velocity component w
__global__ __launch_bounds__(256) void k_inc_visc_rhs(const double *__restrict__ w,
                                                      const double *__restrict__ adv,
                                                      const double *__restrict__ gp, Geom g,
                                                      double *__restrict__ f,
                                                      double *__restrict__ mv, int mpitch, BP P,
                                                      int proj_type)
{
    const int jj = blockIdx.x * blockDim.x + threadIdx.x, ii = blockIdx.y;   // MG indices
    if (jj > g.ny + 1) return;
    const size_t mk = (size_t)ii * mpitch + jj;
    const int p = g.pitch;
    const size_t k = (size_t)(g.ilo + ii - 1) * p + g.jlo + jj - 1;
    const double wc = w[k];
    mv[mk] = wc;
    if (ii >= 1 && ii <= g.nx && jj >= 1 && jj <= g.ny) {
        double r = wc + 0.5 * P.dt * P.nu *
                            ((w[k + p] + w[k - p] - 2.0 * wc) / (P.dx * P.dx) +
                             (w[k + 1] + w[k - 1] - 2.0 * wc) / (P.dy * P.dy));
        if (proj_type == 1) r -= P.dt * (adv[k] + gp[k]);
        else r -= P.dt * adv[k];
        f[mk] = r;
    }
}

// u.v()[:, :] = mg.get_solution().v()  (:141, :176)
__global__ __launch_bounds__(256) void k_inc_visc_store(double *__restrict__ w,
                                                        const double *__restrict__ mv, int mpitch,
                                                        Geom g)
{
    const int jj = blockIdx.x * blockDim.x + threadIdx.x, ii = blockIdx.y;
    if (jj >= g.ny) return;
    w[(size_t)(g.ilo + ii) * g.pitch + g.jlo + jj] = mv[(size_t)(ii + 1) * mpitch + jj + 1];
}

// burgers_viscous: advective term from the unsplit fluxes (simulation.py:64-73,
// construct_unsplit_fluxes burgers_interface.py:178-233) and the right-hand side
// of interface.diffuse (burgers_viscous/interface.py:66-72) for component w
__global__ __launch_bounds__(256) void k_bgv_rhs(const double *__restrict__ w,
                                                 const double *__restrict__ W, Geom g,
                                                 double *__restrict__ f, int mpitch, BP P, int comp)
{
    const int jj = blockIdx.x * blockDim.x + threadIdx.x, ii = blockIdx.y;
    if (jj >= g.ny) return;
    const int p = g.pitch;
    const size_t pl = g.plane;
    const size_t k = (size_t)(g.ilo + ii) * p + g.jlo + jj;
    const double *um = W + W_UMAC * pl, *vm = W + W_VMAC * pl;
    const int XL = comp ? C_VXL : C_UXL, XR = comp ? C_VXR : C_UXR;
    const int YL = comp ? C_VYL : C_UYL, YR = comp ? C_VYR : C_UYR;
    auto fx = [&](size_t kk) { return 0.5 * bg_upwind(W[XL * pl + kk], W[XR * pl + kk], um[kk]) * um[kk]; };
    auto fy = [&](size_t kk) { return 0.5 * bg_upwind(W[YL * pl + kk], W[YR * pl + kk], vm[kk]) * vm[kk]; };
    const double A = (fx(k + p) - fx(k)) / P.dx + (fy(k + 1) - fy(k)) / P.dy;
    const double a = w[k];
    const double lap = (w[k + p] - 2.0 * a + w[k - p]) / (P.dx * P.dx) +
                       (w[k + 1] - 2.0 * a + w[k - 1]) / (P.dy * P.dy);
    f[(size_t)(ii + 1) * mpitch + jj + 1] = a + 0.5 * P.dt * P.eps * lap - P.dt * A;
}

// solution gradient (MG.py:439-469) and the velocity / grad p update
// (simulation.py:329-339; :108-109 with fac = 1 and gp_mode 0)
__global__ __launch_bounds__(256) void k_inc_proj_update(double *__restrict__ u,
                                                         double *__restrict__ v,
                                                         double *__restrict__ gpx,
                                                         double *__restrict__ gpy,
                                                         const double *__restrict__ mv, int mpitch,
                                                         Geom g, BP P, double fac, int gp_mode)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= g.qy) return;
    const size_t k = (size_t)i * g.pitch + j;
    const bool in = (i >= g.ilo && i <= g.ihi && j >= g.jlo && j <= g.jhi);
    if (!in) {
        if (gp_mode == 2) { gpx[k] = 0.0; gpy[k] = 0.0; }   // gradp[:, :] = gradphi[:, :]
        return;
    }
    const size_t mk = (size_t)(i - g.ilo + 1) * mpitch + (j - g.jlo + 1);
    const double gx = 0.5 * (mv[mk + mpitch] - mv[mk - mpitch]) / P.dx;
    const double gy = 0.5 * (mv[mk + 1] - mv[mk - 1]) / P.dy;
    u[k] -= fac * gx;
    v[k] -= fac * gy;
    if (gp_mode == 1) { gpx[k] += gx; gpy[k] += gy; }
    else if (gp_mode == 2) { gpx[k] = gx; gpy[k] = gy; }
}

static int bg_work(pyrohip_state *s)
{
    if (s->work_planes >= (size_t)W_NPL) return 0;
    if (s->work) PYRO_CHECK_HIP(hipFree(s->work));
    s->work = nullptr; s->work_planes = 0;
    const size_t n = s->g.plane * W_NPL + 16;
    PYRO_CHECK_HIP(hipMalloc((void **)&s->work, n * sizeof(double)));
    // zero once: positions the kernels never write are read as 0, like the
    // reference's scratch arrays
    PYRO_CHECK_HIP(hipMemsetAsync(s->work, 0, n * sizeof(double), s->ctx->stream));
    s->work_planes = W_NPL;
    return 0;
}

static BP make_bp(double dx, double dy, double dt, int limiter, double nu = 0.0, double eps = 0.0)
{
    BP P;
    P.dx = dx; P.dy = dy; P.dt = dt; P.dtdx = dt / dx; P.dtdy = dt / dy; P.limiter = limiter;
    P.nu = nu; P.eps = eps;
    return P;
}

// edge states + MAC velocities of the current (u, v[, grad p])
static int bg_predict(pyrohip_state *s, int iu, int iv, int igpx, int igpy, const BP &P)
{
    pyrohip_ctx *c = s->ctx;
    const Geom &g = s->g;
    PYRO_REQUIRE(g.ng >= 4, "the CTU predictor needs ng >= 4");
    PYRO_TRY(bg_work(s));
    double *W = s->work + geom_lead(g);
    const double *u = s->d + (size_t)iu * g.plane, *v = s->d + (size_t)iv * g.plane;
    const double *gpx = igpx >= 0 ? s->d + (size_t)igpx * g.plane : nullptr;
    const double *gpy = igpy >= 0 ? s->d + (size_t)igpy * g.plane : nullptr;
    const dim3 block(256);
    const dim3 gridB1((g.ny + 2 + 255) / 256, g.nx + 2), gridF((g.ny + 1 + 255) / 256, g.nx + 1);
    PYRO_LAUNCH(c, "k_bg_hat", k_bg_hat, gridB1, block, 0, u, v, W, g, P);
    PYRO_LAUNCH(c, "k_bg_trans", k_bg_trans, gridB1, block, 0, u, v, gpx, gpy, W, g, P);
    PYRO_LAUNCH(c, "k_bg_mac", k_bg_mac, gridF, block, 0, W, g);
    PYRO_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace pyro

using namespace pyro;

#define BG_CHECK_VARS(s, ...)                                                      \
    do {                                                                           \
        const int _v[] = {__VA_ARGS__};                                            \
        for (int _x : _v) PYRO_REQUIRE(_x >= 0 && _x < (s)->nvar, "variable index out of range"); \
    } while (0)

#define INC_CHECK_MG(s, m, F)                                                                  \
    PYRO_REQUIRE((s) && (m), "NULL argument");                                                 \
    MgFinest F;                                                                                \
    PYRO_TRY(mg_finest((m), &F));                                                              \
    PYRO_REQUIRE(F.ctx == (s)->ctx, "state and multigrid live on different contexts");         \
    PYRO_REQUIRE((s)->g.nx == F.n && (s)->g.ny == F.n, "state and multigrid sizes differ")

extern "C" {

int pyrohip_bg_step(pyrohip_state *s, int iu, int iv, double dx, double dy, double dt, int limiter)
{
    PYRO_REQUIRE(s, "NULL state");
    BG_CHECK_VARS(s, iu, iv);
    const BP P = make_bp(dx, dy, dt, limiter);
    PYRO_TRY(bg_predict(s, iu, iv, -1, -1, P));
    const Geom &g = s->g;
    PYRO_LAUNCH(s->ctx, "k_bg_update", k_bg_update, dim3((g.ny + 255) / 256, g.nx), dim3(256), 0,
                s->d + (size_t)iu * g.plane, s->d + (size_t)iv * g.plane,
                (const double *)(s->work + geom_lead(g)), g, P);
    PYRO_CHECK_HIP(hipGetLastError());
    return 0;
}

int pyrohip_inc_mac_rhs(pyrohip_state *s, pyrohip_mg *m, int iu, int iv, int igpx, int igpy,
                        double dx, double dy, double dt, int limiter, double nu,
                        double *source_norm)
{
    INC_CHECK_MG(s, m, F);
    BG_CHECK_VARS(s, iu, iv, igpx, igpy);
    PYRO_REQUIRE(nu >= 0.0, "negative viscosity");
    const BP P = make_bp(dx, dy, dt, limiter, nu);
    PYRO_TRY(bg_predict(s, iu, iv, igpx, igpy, P));
    const Geom &g = s->g;
    PYRO_TRY(pyrohip_mg_zero(m, F.level, 0));   // init_zeros
    PYRO_TRY(pyrohip_mg_zero(m, F.level, 1));
    PYRO_LAUNCH(s->ctx, "k_inc_div_mac", k_inc_div_mac, dim3((g.ny + 255) / 256, g.nx), dim3(256),
                0, (const double *)(s->work + geom_lead(g)), g, F.f, F.pitch, P);
    PYRO_CHECK_HIP(hipGetLastError());
    return pyrohip_mg_init_rhs_norm(m, source_norm);
}

int pyrohip_inc_advect(pyrohip_state *s, pyrohip_mg *m, int iu, int iv, int iphimac, int igpx,
                       int igpy, double dx, double dy, double dt, int proj_type)
{
    INC_CHECK_MG(s, m, F);
    BG_CHECK_VARS(s, iu, iv, iphimac, igpx, igpy);
    PYRO_REQUIRE(s->work_planes >= (size_t)W_NPL, "call pyrohip_inc_mac_rhs first");
    const BP P = make_bp(dx, dy, dt, 0);
    const Geom &g = s->g;
    pyrohip_ctx *c = s->ctx;
    double *W = s->work + geom_lead(g);
    double *phiM = s->d + (size_t)iphimac * g.plane;
    const dim3 block(256);
    PYRO_LAUNCH(c, "k_inc_copy_b1", k_inc_copy_b1, dim3((g.qy + 255) / 256, g.qx), block, 0,
                (const double *)F.v, F.pitch, phiM, g, 0);
    PYRO_LAUNCH(c, "k_inc_mac_project", k_inc_mac_project, dim3((g.ny + 1 + 255) / 256, g.nx + 1),
                block, 0, W, (const double *)phiM, g, P);
    PYRO_LAUNCH(c, "k_inc_advect", k_inc_advect, dim3((g.ny + 255) / 256, g.nx), block, 0,
                s->d + (size_t)iu * g.plane, s->d + (size_t)iv * g.plane,
                (const double *)(s->d + (size_t)igpx * g.plane),
                (const double *)(s->d + (size_t)igpy * g.plane), W, g, P, proj_type);
    PYRO_CHECK_HIP(hipGetLastError());
    s->next_cfl_min = -1.0;
    return 0;
}

int pyrohip_inc_proj_rhs(pyrohip_state *s, pyrohip_mg *m, int iu, int iv, int iphi, double dx,
                         double dy, double dt, int divide_by_dt, double *source_norm)
{
    INC_CHECK_MG(s, m, F);
    BG_CHECK_VARS(s, iu, iv);
    PYRO_REQUIRE(iphi >= -1 && iphi < s->nvar, "variable index out of range");
    const BP P = make_bp(dx, dy, dt, 0);
    const Geom &g = s->g;
    PYRO_TRY(pyrohip_mg_zero(m, F.level, 1));
    PYRO_LAUNCH(s->ctx, "k_inc_div_cc", k_inc_div_cc, dim3((g.ny + 2 + 255) / 256, g.nx + 2),
                dim3(256), 0, (const double *)(s->d + (size_t)iu * g.plane),
                (const double *)(s->d + (size_t)iv * g.plane),
                iphi >= 0 ? (const double *)(s->d + (size_t)iphi * g.plane) : nullptr, g, F.f, F.v,
                F.pitch, P, divide_by_dt);
    PYRO_CHECK_HIP(hipGetLastError());
    PYRO_TRY(mg_solution_written(m));
    return pyrohip_mg_init_rhs_norm(m, source_norm);
}

int pyrohip_inc_proj_update(pyrohip_state *s, pyrohip_mg *m, int iu, int iv, int iphi, int igpx,
                            int igpy, double dx, double dy, double fac, int gp_mode)
{
    INC_CHECK_MG(s, m, F);
    BG_CHECK_VARS(s, iu, iv, iphi);
    PYRO_REQUIRE(gp_mode >= 0 && gp_mode <= 2, "gp_mode: 0 none, 1 +=, 2 =");
    if (gp_mode) BG_CHECK_VARS(s, igpx, igpy);
    const BP P = make_bp(dx, dy, 0.0, 0);
    const Geom &g = s->g;
    pyrohip_ctx *c = s->ctx;
    const dim3 block(256), gridA((g.qy + 255) / 256, g.qx);
    PYRO_TRY(mg_solution_ghosts(m));   // solve() ends with fill_BC(v)
    PYRO_LAUNCH(c, "k_inc_copy_b1", k_inc_copy_b1, gridA, block, 0, (const double *)F.v, F.pitch,
                s->d + (size_t)iphi * g.plane, g, 1);
    PYRO_LAUNCH(c, "k_inc_proj_update", k_inc_proj_update, gridA, block, 0,
                s->d + (size_t)iu * g.plane, s->d + (size_t)iv * g.plane,
                gp_mode ? s->d + (size_t)igpx * g.plane : nullptr,
                gp_mode ? s->d + (size_t)igpy * g.plane : nullptr, (const double *)F.v, F.pitch, g,
                P, fac, gp_mode);
    PYRO_CHECK_HIP(hipGetLastError());
    s->next_cfl_min = -1.0;
    return 0;
}

int pyrohip_inc_visc_rhs(pyrohip_state *s, pyrohip_mg *m, int iw, int comp, int igp, double dx,
                         double dy, double dt, double nu, int proj_type, double *source_norm)
{
    INC_CHECK_MG(s, m, F);
    BG_CHECK_VARS(s, iw, igp);
    PYRO_REQUIRE(comp == 0 || comp == 1, "comp: 0 = u, 1 = v");
    PYRO_REQUIRE(proj_type == 1 || proj_type == 2, "proj_type must be 1 or 2");
    PYRO_REQUIRE(s->work_planes >= (size_t)W_NPL, "call pyrohip_inc_advect first");
    const BP P = make_bp(dx, dy, dt, 0, nu);
    const Geom &g = s->g;
    PYRO_TRY(pyrohip_mg_zero(m, F.level, 1));
    PYRO_LAUNCH(s->ctx, "k_inc_visc_rhs", k_inc_visc_rhs, dim3((g.ny + 2 + 255) / 256, g.nx + 2),
                dim3(256), 0, (const double *)(s->d + (size_t)iw * g.plane),
                (const double *)(s->work + geom_lead(g) +
                                 (size_t)(comp ? W_ADVY : W_ADVX) * g.plane),
                (const double *)(s->d + (size_t)igp * g.plane), g, F.f, F.v, F.pitch, P,
                proj_type);
    PYRO_CHECK_HIP(hipGetLastError());
    PYRO_TRY(mg_solution_written(m));
    return pyrohip_mg_init_rhs_norm(m, source_norm);
}

int pyrohip_bgv_predict(pyrohip_state *s, int iu, int iv, double dx, double dy, double dt,
                        int limiter, double eps)
{
    PYRO_REQUIRE(s, "NULL state");
    BG_CHECK_VARS(s, iu, iv);
    return bg_predict(s, iu, iv, -1, -1, make_bp(dx, dy, dt, limiter, 0.0, eps));
}

int pyrohip_bgv_rhs(pyrohip_state *s, pyrohip_mg *m, int iw, int comp, double dx, double dy,
                    double dt, double eps, double *source_norm)
{
    INC_CHECK_MG(s, m, F);
    BG_CHECK_VARS(s, iw);
    PYRO_REQUIRE(comp == 0 || comp == 1, "comp: 0 = u, 1 = v");
    PYRO_REQUIRE(s->work_planes >= (size_t)W_NPL, "call pyrohip_bgv_predict first");
    const BP P = make_bp(dx, dy, dt, 0, 0.0, eps);
    const Geom &g = s->g;
    PYRO_TRY(pyrohip_mg_zero(m, F.level, 0));   // init_zeros
    PYRO_TRY(pyrohip_mg_zero(m, F.level, 1));
    PYRO_LAUNCH(s->ctx, "k_bgv_rhs", k_bgv_rhs, dim3((g.ny + 255) / 256, g.nx), dim3(256), 0,
                (const double *)(s->d + (size_t)iw * g.plane),
                (const double *)(s->work + geom_lead(g)), g, F.f, F.pitch, P, comp);
    PYRO_CHECK_HIP(hipGetLastError());
    return pyrohip_mg_init_rhs_norm(m, source_norm);
}

int pyrohip_inc_visc_store(pyrohip_state *s, pyrohip_mg *m, int iw)
{
    INC_CHECK_MG(s, m, F);
    BG_CHECK_VARS(s, iw);
    const Geom &g = s->g;
    PYRO_LAUNCH(s->ctx, "k_inc_visc_store", k_inc_visc_store, dim3((g.ny + 255) / 256, g.nx),
                dim3(256), 0, s->d + (size_t)iw * g.plane, (const double *)F.v, F.pitch, g);
    PYRO_CHECK_HIP(hipGetLastError());
    s->next_cfl_min = -1.0;
    return 0;
}

// which: 0-7 corrected edge states (u_xl u_xr u_yl u_yr v_xl v_xr v_yl v_yr),
// 8 u_MAC, 9 v_MAC, 10 advect_x, 11 advect_y; host: (qx, qy)
int pyrohip_inc_stage_dump(pyrohip_state *s, int which, double *host)
{
    PYRO_REQUIRE(s && host, "NULL argument");
    PYRO_REQUIRE(which >= 0 && which < 12, "which out of range");
    PYRO_REQUIRE(s->work_planes >= (size_t)W_NPL, "no predictor work area yet");
    const Geom &g = s->g;
    const int plane = (which < 8) ? C_UXL + which : W_UMAC + (which - 8);
    PYRO_CHECK_HIP(hipMemcpy2DAsync(host, g.qy * sizeof(double),
                                    s->work + geom_lead(g) + (size_t)plane * g.plane,
                                    g.pitch * sizeof(double), g.qy * sizeof(double), g.qx,
                                    hipMemcpyDeviceToHost, s->ctx->stream));
    PYRO_CHECK_HIP(hipStreamSynchronize(s->ctx->stream));
    return 0;
}

}  // extern "C"
