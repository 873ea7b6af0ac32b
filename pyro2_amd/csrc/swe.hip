// Shallow-water solver: the unsplit CTU scheme of
// pyro/swe on the device, staged through global work planes like kernel_set 0
// of the compressible solver.
//
//   pyro/swe/simulation.py:48-80, 143-193    cons/prim, CFL, evolve
//   pyro/swe/unsplit_fluxes.py:132-380       orchestration, transverse terms
//   pyro/swe/interface.py:5-578              states, riemann_roe, riemann_hllc,
//                                            consFlux
//
// State: 4 planes height, x-momentum, y-momentum, fuel (= h X); primitive
// h, u, v, X.  Compiled with -ffp-contract=off and the reference's operation
// order (4-term in-order dot products included): bit-identical to the oracle.
#include "common.h"
#include "reduce.h"
#include "stencil.h"

namespace pyro {

struct SW {   // kernel parameters
    double dx, dy, dt, g;
    int limiter, riemann;   // riemann: 0 Roe, 1 HLLC
};

enum {   // work planes
    SW_Q = 0,                                // 4: h u v X
    SW_XM = 4, SW_XP = 8, SW_YM = 12, SW_YP = 16,   // face states of the cell (conserved)
    SW_FXT = 20, SW_FYT = 24, SW_FX = 28, SW_FY = 32,
    SW_NPL = 36
};

struct V4 { double a[4]; };   // conserved: h, mx, my, hX

__device__ __forceinline__ V4 ld4(const double *__restrict__ p, size_t pl, size_t k)
{
    return V4{{p[k], p[pl + k], p[2 * pl + k], p[3 * pl + k]}};
}
__device__ __forceinline__ void st4(double *__restrict__ p, size_t pl, size_t k, const V4 &v)
{
    p[k] = v.a[0]; p[pl + k] = v.a[1]; p[2 * pl + k] = v.a[2]; p[3 * pl + k] = v.a[3];
}

// simulation.py:65-80
__device__ __forceinline__ V4 sw_prim_to_cons(const double q[4])
{
    V4 U;
    U.a[0] = q[0];
    U.a[1] = q[1] * U.a[0];
    U.a[2] = q[2] * U.a[0];
    U.a[3] = q[3] * q[0];
    return U;
}

// interface.py:557-578; x: idir == 1
__device__ __forceinline__ V4 sw_cons_flux(const V4 &U, double g, bool x)
{
    const double u = U.a[1] / U.a[0], v = U.a[2] / U.a[0];
    const double w = x ? u : v;
    V4 F;
    F.a[0] = U.a[0] * w;
    F.a[1] = U.a[1] * w;
    F.a[2] = U.a[2] * w;
    const double pr = 0.5 * g * (U.a[0] * U.a[0]);
    if (x) F.a[1] = F.a[1] + pr; else F.a[2] = F.a[2] + pr;
    F.a[3] = U.a[3] * w;
    return F;
}

// characteristic tracing of one cell in one direction, interface.py:5-213:
// primitive states on the cell's lower face (q_r[face]) and upper face
// (q_l[face+1])
__device__ __forceinline__ void sw_trace(const double q[4], const double dq[4], double g,
                                         double dtdx, bool x, double lo[4], double hi[4])
{
    const int in = x ? 1 : 2, it = x ? 2 : 1;
    const double cs = sqrt(g * q[0]);
    const double dtdx3 = 0.33333 * dtdx;   // sic, interface.py:100
    double lvec[4][4] = {}, rvec[4][4] = {}, e_val[4], betal[4], betar[4];
    e_val[0] = q[in] - cs; e_val[1] = q[in]; e_val[2] = q[in] + cs; e_val[3] = q[in];
    lvec[0][0] = cs;   lvec[0][in] = -q[0];
    lvec[1][it] = 1.0;
    lvec[2][0] = cs;   lvec[2][in] = q[0];
    rvec[0][0] = q[0]; rvec[0][in] = -cs;
    rvec[1][it] = 1.0;
    rvec[2][0] = q[0]; rvec[2][in] = cs;
    lvec[3][3] = 1.0; rvec[3][3] = 1.0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        lvec[0][k] = lvec[0][k] * 0.50 / (cs * q[0]);
        lvec[2][k] = -lvec[2][k] * 0.50 / (cs * q[0]);
    }
    double factor = 0.5 * (1.0 - dtdx * fmax(e_val[2], 0.0));
#pragma unroll
    for (int m = 0; m < 4; m++) hi[m] = q[m] + factor * dq[m];
    factor = 0.5 * (1.0 + dtdx * fmin(e_val[0], 0.0));
#pragma unroll
    for (int m = 0; m < 4; m++) lo[m] = q[m] - factor * dq[m];
#pragma unroll
    for (int m = 0; m < 4; m++) {
        double asum = 0.0;
#pragma unroll
        for (int k = 0; k < 4; k++) asum += lvec[m][k] * dq[k];
        betal[m] = dtdx3 * (e_val[2] - e_val[m]) * (copysign(1.0, e_val[m]) + 1.0) * asum;
        betar[m] = dtdx3 * (e_val[0] - e_val[m]) * (1.0 - copysign(1.0, e_val[m])) * asum;
    }
#pragma unroll
    for (int m = 0; m < 4; m++) {
        double sum_l = 0.0, sum_r = 0.0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            sum_l += betal[k] * rvec[k][m];
            sum_r += betar[k] * rvec[k][m];
        }
        hi[m] = hi[m] + sum_l;
        lo[m] = lo[m] + sum_r;
    }
}

// interface.py:216-385
__device__ __forceinline__ V4 sw_roe(const V4 &Ul, const V4 &Ur, double g, bool x)
{
    const double smallc = 1.e-10, tol = 0.1e-1;
    const int im = x ? 1 : 2, it = x ? 2 : 1;
    const double h_l = Ul.a[0], un_l = Ul.a[im] / h_l;
    const double h_r = Ur.a[0], un_r = Ur.a[im] / h_r;
    const double c_l = fmax(smallc, sqrt(g * h_l)), c_r = fmax(smallc, sqrt(g * h_r));
    double U_roe[4], delta[4], lambda[4], alpha[4], K[4][4] = {};
#pragma unroll
    for (int n = 0; n < 4; n++) {
        U_roe[n] = (Ul.a[n] / sqrt(h_l) + Ur.a[n] / sqrt(h_r)) / (sqrt(h_l) + sqrt(h_r));
        delta[n] = Ur.a[n] / h_r - Ul.a[n] / h_l;
    }
    U_roe[0] = sqrt(h_l * h_r);
    const double c_roe = sqrt(0.5 * (c_l * c_l + c_r * c_r));
    delta[0] = h_r - h_l;
    const double un_roe = U_roe[im];
    lambda[0] = un_roe - c_roe; lambda[1] = un_roe; lambda[2] = un_roe + c_roe; lambda[3] = un_roe;
    alpha[0] = 0.5 * (delta[0] - U_roe[0] / c_roe * delta[im]);
    alpha[1] = U_roe[0] * delta[it];
    alpha[2] = 0.5 * (delta[0] + U_roe[0] / c_roe * delta[im]);
    alpha[3] = U_roe[0] * delta[3];
    K[0][0] = 1.0; K[0][im] = un_roe - c_roe; K[0][it] = U_roe[it];
    K[1][it] = 1.0;
    K[2][0] = 1.0; K[2][im] = un_roe + c_roe; K[2][it] = U_roe[it];
    K[3][3] = 1.0;
    const V4 Fl = sw_cons_flux(Ul, g, x), Fr = sw_cons_flux(Ur, g, x);
    V4 F;
#pragma unroll
    for (int n = 0; n < 4; n++) F.a[n] = 0.5 * (Fl.a[n] + Fr.a[n]);
    const double hs = 0.5 * (c_l + c_r) + 0.25 * (un_l - un_r);
    const double h_star = 1.0 / g * (hs * hs);
    const double u_star = 0.5 * (un_l + un_r) + c_l - c_r;
    const double c_star = sqrt(g * h_star);
    if (fabs(lambda[0]) < tol)
        lambda[0] = lambda[0] * (u_star - c_star - lambda[0]) / (u_star - c_star - (un_l - c_l));
    if (fabs(lambda[2]) < tol)
        lambda[2] = lambda[2] * (u_star + c_star - lambda[2]) / (u_star + c_star - (un_r + c_r));
#pragma unroll
    for (int n = 0; n < 4; n++)
#pragma unroll
        for (int m = 0; m < 4; m++) F.a[n] -= 0.5 * alpha[m] * fabs(lambda[m]) * K[m][n];
    return F;
}

// interface.py:388-554
__device__ __forceinline__ V4 sw_hllc(const V4 &Ul, const V4 &Ur, double g, bool x)
{
    const double smallc = 1.e-10;
    const int im = x ? 1 : 2, it = x ? 2 : 1;
    const double h_l = Ul.a[0], un_l = Ul.a[im] / h_l, ut_l = Ul.a[it] / h_l;
    const double h_r = Ur.a[0], un_r = Ur.a[im] / h_r, ut_r = Ur.a[it] / h_r;
    const double c_l = fmax(smallc, sqrt(g * h_l)), c_r = fmax(smallc, sqrt(g * h_r));
    const double h_avg = 0.5 * (h_l + h_r), c_avg = 0.5 * (c_l + c_r);
    const double hstar = h_avg - 0.25 * (un_r - un_l) * h_avg / c_avg;
    const double S_l = (hstar <= h_l) ? un_l - c_l
                                      : un_l - c_l * sqrt(0.5 * (hstar + h_l) * hstar) / h_l;
    const double S_r = (hstar <= h_r) ? un_r + c_r
                                      : un_r + c_r * sqrt(0.5 * (hstar + h_r) * hstar) / h_r;
    const double S_c = (S_l * h_r * (un_r - S_r) - S_r * h_l * (un_l - S_l)) /
                       (h_r * (un_r - S_r) - h_l * (un_l - S_l));
    V4 Us, F;
    if (S_r <= 0.0) return sw_cons_flux(Ur, g, x);
    if (S_c <= 0.0 && 0.0 < S_r) {
        const double fac = h_r * (S_r - un_r) / (S_r - S_c);
        Us.a[0] = fac; Us.a[im] = fac * S_c; Us.a[it] = fac * ut_r;
        Us.a[3] = fac * Ur.a[3] / h_r;
        F = sw_cons_flux(Ur, g, x);
#pragma unroll
        for (int n = 0; n < 4; n++) F.a[n] = F.a[n] + S_r * (Us.a[n] - Ur.a[n]);
        return F;
    }
    if (S_l < 0.0 && 0.0 < S_c) {
        const double fac = h_l * (S_l - un_l) / (S_l - S_c);
        Us.a[0] = fac; Us.a[im] = fac * S_c; Us.a[it] = fac * ut_l;
        Us.a[3] = fac * Ul.a[3] / h_l;
        F = sw_cons_flux(Ul, g, x);
#pragma unroll
        for (int n = 0; n < 4; n++) F.a[n] = F.a[n] + S_l * (Us.a[n] - Ul.a[n]);
        return F;
    }
    return sw_cons_flux(Ul, g, x);
}

__device__ __forceinline__ V4 sw_riemann(const V4 &Ul, const V4 &Ur, const SW &P, bool x)
{
    return P.riemann == 1 ? sw_hllc(Ul, Ur, P.g, x) : sw_roe(Ul, Ur, P.g, x);
}

// ---- stage 0: primitives over the whole array (simulation.py:48-63) ------
__global__ __launch_bounds__(256) void k_sw_prim(const double *__restrict__ U,
                                                 double *__restrict__ W, Geom g)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= g.qy) return;
    const size_t k = (size_t)i * g.pitch + j, pl = g.plane;
    const V4 Uc = ld4(U, pl, k);
    double *Q = W + (size_t)SW_Q * pl;
    Q[k] = Uc.a[0];
    Q[pl + k] = Uc.a[1] / Uc.a[0];
    Q[2 * pl + k] = Uc.a[2] / Uc.a[0];
    Q[3 * pl + k] = Uc.a[3] / Uc.a[0];
}

// ---- stage 1: limited slopes + tracing for the cells of R(1) --------------
__global__ __launch_bounds__(256) void k_sw_states(double *__restrict__ W, Geom g, SW P)
{
    const int j = g.jlo - 1 + blockIdx.x * blockDim.x + threadIdx.x;
    const int i = g.ilo - 1 + blockIdx.y;
    if (j > g.jhi + 1) return;
    const int p = g.pitch;
    const size_t k = (size_t)i * p + j, pl = g.plane;
    const double *Q = W + (size_t)SW_Q * pl;
    double q[4], dqx[4], dqy[4];
#pragma unroll
    for (int n = 0; n < 4; n++) {
        const double *a = Q + (size_t)n * pl;
        q[n] = a[k];
        // xi = 1.0 (unsplit_fluxes.py:175-177; no flattening for swe)
        dqx[n] = 1.0 * limited_slope(a[k - 2 * p], a[k - p], a[k], a[k + p], a[k + 2 * p],
                                     P.limiter);
        dqy[n] = 1.0 * limited_slope(a[k - 2], a[k - 1], a[k], a[k + 1], a[k + 2], P.limiter);
    }
    double lo[4], hi[4];
    sw_trace(q, dqx, P.g, P.dt / P.dx, true, lo, hi);
    st4(W + (size_t)SW_XM * pl, pl, k, sw_prim_to_cons(lo));
    st4(W + (size_t)SW_XP * pl, pl, k, sw_prim_to_cons(hi));
    sw_trace(q, dqy, P.g, P.dt / P.dy, false, lo, hi);
    st4(W + (size_t)SW_YM * pl, pl, k, sw_prim_to_cons(lo));
    st4(W + (size_t)SW_YP * pl, pl, k, sw_prim_to_cons(hi));
}

// ---- stage 2: transverse Riemann problems on the lower faces of R(1) ------
__global__ __launch_bounds__(256) void k_sw_riemann_t(double *__restrict__ W, Geom g, SW P)
{
    const int j = g.jlo - 1 + blockIdx.x * blockDim.x + threadIdx.x;
    const int i = g.ilo - 1 + blockIdx.y;
    if (j > g.jhi + 1) return;
    const int p = g.pitch;
    const size_t k = (size_t)i * p + j, pl = g.plane;
    if (i >= g.ilo)
        st4(W + (size_t)SW_FXT * pl, pl, k,
            sw_riemann(ld4(W + (size_t)SW_XP * pl, pl, k - p), ld4(W + (size_t)SW_XM * pl, pl, k),
                       P, true));
    if (j >= g.jlo)
        st4(W + (size_t)SW_FYT * pl, pl, k,
            sw_riemann(ld4(W + (size_t)SW_YP * pl, pl, k - 1), ld4(W + (size_t)SW_YM * pl, pl, k),
                       P, false));
}

__device__ __forceinline__ V4 sw_corrected(const V4 &U, const V4 &Fhi, const V4 &Flo, double c)
{
    // U += -0.5*dtdy*(F_hi - F_lo), unsplit_fluxes.py:336-352 (c = 0.5*dt/d)
    V4 r;
#pragma unroll
    for (int n = 0; n < 4; n++) r.a[n] = U.a[n] + (-c * (Fhi.a[n] - Flo.a[n]));
    return r;
}

// ---- stage 3: transverse correction + final Riemann problems --------------
// thread (i,j) in [ilo, ihi+1] x [jlo, jhi+1]
__global__ __launch_bounds__(256) void k_sw_final(double *__restrict__ W, Geom g, SW P)
{
    const int j = g.jlo + blockIdx.x * blockDim.x + threadIdx.x;
    const int i = g.ilo + blockIdx.y;
    if (j > g.jhi + 1) return;
    const int p = g.pitch;
    const size_t k = (size_t)i * p + j, pl = g.plane;
    const double *FXT = W + (size_t)SW_FXT * pl, *FYT = W + (size_t)SW_FYT * pl;
    const double hdtdx = 0.5 * (P.dt / P.dx), hdtdy = 0.5 * (P.dt / P.dy);
    if (j <= g.jhi) {
        const V4 Uxl = sw_corrected(ld4(W + (size_t)SW_XP * pl, pl, k - p), ld4(FYT, pl, k - p + 1),
                                    ld4(FYT, pl, k - p), hdtdy);
        const V4 Uxr = sw_corrected(ld4(W + (size_t)SW_XM * pl, pl, k), ld4(FYT, pl, k + 1),
                                    ld4(FYT, pl, k), hdtdy);
        st4(W + (size_t)SW_FX * pl, pl, k, sw_riemann(Uxl, Uxr, P, true));
    }
    if (i <= g.ihi) {
        const V4 Uyl = sw_corrected(ld4(W + (size_t)SW_YP * pl, pl, k - 1), ld4(FXT, pl, k + p - 1),
                                    ld4(FXT, pl, k - 1), hdtdx);
        const V4 Uyr = sw_corrected(ld4(W + (size_t)SW_YM * pl, pl, k), ld4(FXT, pl, k + p),
                                    ld4(FXT, pl, k), hdtdx);
        st4(W + (size_t)SW_FY * pl, pl, k, sw_riemann(Uyl, Uyr, P, false));
    }
}

// ---- stage 4: conservative update (simulation.py:172-181) -----------------
__global__ __launch_bounds__(256) void k_sw_update(double *__restrict__ U,
                                                   const double *__restrict__ W, Geom g, SW P)
{
    const int j = g.jlo + blockIdx.x * blockDim.x + threadIdx.x;
    const int i = g.ilo + blockIdx.y;
    if (j > g.jhi) return;
    const int p = g.pitch;
    const size_t k = (size_t)i * p + j, pl = g.plane;
    const double dtdx = P.dt / P.dx, dtdy = P.dt / P.dy;
#pragma unroll
    for (int n = 0; n < 4; n++) {
        const double *fx = W + (size_t)(SW_FX + n) * pl, *fy = W + (size_t)(SW_FY + n) * pl;
        U[(size_t)n * pl + k] += dtdx * (fx[k] - fx[k + p]) + dtdy * (fy[k] - fy[k + 1]);
    }
}

// simulation.py:143-153: min over the whole array of dx/(|u|+c), dy/(|v|+c)
__global__ __launch_bounds__(256) void k_sw_cfl(const double *__restrict__ U, Geom g, double grav,
                                                double dx, double dy, double *__restrict__ partial)
{
    double m = INFINITY;
    for (int i = blockIdx.y; i < g.qx; i += gridDim.y)
        for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < g.qy; j += gridDim.x * blockDim.x) {
            const size_t k = (size_t)i * g.pitch + j;
            const double h = U[k], u = U[g.plane + k] / h, v = U[2 * g.plane + k] / h;
            const double cs = sqrt(grav * h);
            m = fmin(m, fmin(dx / (fabs(u) + cs), dy / (fabs(v) + cs)));
        }
    m = block_reduce_min(m);
    if (threadIdx.x == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = m;
}

static int sw_work(pyrohip_state *s)
{
    if (s->work_planes >= (size_t)SW_NPL) return 0;
    if (s->work) PYRO_CHECK_HIP(hipFree(s->work));
    s->work = nullptr; s->work_planes = 0;
    const size_t n = s->g.plane * SW_NPL + 16;
    PYRO_CHECK_HIP(hipMalloc((void **)&s->work, n * sizeof(double)));
    PYRO_CHECK_HIP(hipMemsetAsync(s->work, 0, n * sizeof(double), s->ctx->stream));
    s->work_planes = SW_NPL;
    return 0;
}

}  // namespace pyro

using namespace pyro;

static int sw_check(pyrohip_state *s, double dx, double dy, double grav, int limiter, int riemann)
{
    PYRO_REQUIRE(s, "NULL state");
    PYRO_REQUIRE(s->nvar == 4, "swe state must have 4 variables (height, x-momentum, y-momentum, fuel)");
    PYRO_REQUIRE(s->g.ng >= 4, "swe needs ng >= 4 (swe/simulation.py:98)");
    PYRO_REQUIRE(dx > 0 && dy > 0 && grav > 0, "bad dx / dy / grav");
    PYRO_REQUIRE(limiter >= 0 && limiter <= 2, "limiter must be 0, 1 or 2");
    PYRO_REQUIRE(riemann == 0 || riemann == 1, "riemann must be 0 (Roe) or 1 (HLLC)");
    return 0;
}

extern "C" {

int pyrohip_swe_dt(pyrohip_state *s, double dx, double dy, double grav, double cfl, double *dt_out)
{
    PYRO_TRY(sw_check(s, dx, dy, grav, 0, 0));
    PYRO_REQUIRE(dt_out, "dt_out is NULL");
    pyrohip_ctx *c = s->ctx;
    const dim3 grid(8, 128), block(256);
    const int nb = grid.x * grid.y;
    PYRO_TRY(c->reduce.ensure((nb + kMinStageBlocks + 2) * sizeof(double)));
    double *part = (double *)c->reduce.p;
    hipLaunchKernelGGL(k_sw_cfl, grid, block, 0, c->stream, (const double *)s->d, s->g, grav, dx, dy,
                       part);
    const double *dmin = launch_min_reduce(c->stream, part, nb);
    PYRO_CHECK_HIP(hipGetLastError());
    PYRO_CHECK_HIP(hipMemcpyAsync(c->reduce_host, dmin, sizeof(double), hipMemcpyDeviceToHost,
                                  c->stream));
    PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));
    *dt_out = cfl * ((double *)c->reduce_host)[0];
    return 0;
}

int pyrohip_swe_step(pyrohip_state *s, double dx, double dy, double grav, int limiter, int riemann,
                     double dt)
{
    PYRO_TRY(sw_check(s, dx, dy, grav, limiter, riemann));
    PYRO_REQUIRE(dt > 0.0, "dt must be positive");
    PYRO_TRY(sw_work(s));
    pyrohip_ctx *c = s->ctx;
    const Geom &g = s->g;
    const SW P{dx, dy, dt, grav, limiter, riemann};
    double *W = s->work + geom_lead(g);
    const dim3 block(256);
    const dim3 gridA((g.qy + 255) / 256, g.qx), gridR1((g.ny + 2 + 255) / 256, g.nx + 2),
        gridF((g.ny + 1 + 255) / 256, g.nx + 1), gridI((g.ny + 255) / 256, g.nx);
    PYRO_LAUNCH(c, "k_sw_prim", k_sw_prim, gridA, block, 0, (const double *)s->d, W, g);
    PYRO_LAUNCH(c, "k_sw_states", k_sw_states, gridR1, block, 0, W, g, P);
    PYRO_LAUNCH(c, "k_sw_riemann_t", k_sw_riemann_t, gridR1, block, 0, W, g, P);
    PYRO_LAUNCH(c, "k_sw_final", k_sw_final, gridF, block, 0, W, g, P);
    PYRO_LAUNCH(c, "k_sw_update", k_sw_update, gridI, block, 0, s->d, (const double *)W, g, P);
    PYRO_CHECK_HIP(hipGetLastError());
    s->next_cfl_min = -1.0;
    return 0;
}

// stage: 0 Uxl0 1 Uxr0 2 Uyl0 3 Uyr0 (face states before the transverse
// terms, reference face indexing), 4 FxT 5 FyT 6 Fx 7 Fy -> host (qx, qy, 4)
int pyrohip_swe_stage_dump(pyrohip_state *s, int stage, double *out)
{
    PYRO_REQUIRE(s && out, "NULL argument");
    PYRO_REQUIRE(stage >= 0 && stage < 8, "stage out of range");
    PYRO_REQUIRE(s->work_planes >= (size_t)SW_NPL, "no swe step has been run");
    static const int first[8] = {SW_XP, SW_XM, SW_YP, SW_YM, SW_FXT, SW_FYT, SW_FX, SW_FY};
    const Geom &g = s->g;
    pyrohip_ctx *c = s->ctx;
    std::vector<double> tmp((size_t)g.qx * g.qy);
    // XP / YP are stored at the cell whose upper face they sit on: shift by one
    // cell so that out[i, j] is the reference's U_xl[i, j] / U_yl[i, j]
    const int si = (stage == 0) ? 1 : 0, sj = (stage == 2) ? 1 : 0;
    for (int n = 0; n < 4; n++) {
        PYRO_CHECK_HIP(hipMemcpy2DAsync(tmp.data(), g.qy * sizeof(double),
                                        s->work + geom_lead(g) + (size_t)(first[stage] + n) * g.plane,
                                        g.pitch * sizeof(double), g.qy * sizeof(double), g.qx,
                                        hipMemcpyDeviceToHost, c->stream));
        PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));
        for (int i = 0; i < g.qx; i++)
            for (int j = 0; j < g.qy; j++) {
                const int ii = i - si, jj = j - sj;
                out[((size_t)i * g.qy + j) * 4 + n] =
                    (ii >= 0 && jj >= 0) ? tmp[(size_t)ii * g.qy + jj] : 0.0;
            }
    }
    return 0;
}

}  // extern "C"
