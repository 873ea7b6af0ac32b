// Linear advection, 2nd-order unsplit CTU update, one fused LDS-tiled kernel.
//
// Replaces (reference file:line)
//   pyro/advection/simulation.py:56-94        Simulation.evolve
//   pyro/advection/advective_fluxes.py:1-92   unsplit_fluxes
//   pyro/advection/interface.py:4-43          linear_interface
//   pyro/mesh/reconstruction.py:9-120         limit / limit2 / limit4
//
// Roofline: HBM bound, 16 B per cell update (read a, write a).  The tile
// (TI x TJ interior + 3-cell apron) is staged once in LDS; interface states
// a_x / a_y are built in LDS; the conservative update reads them from LDS, so
// every input cell is fetched from HBM once (aprons are L2 hits: tiles are
// dealt to XCDs in contiguous bands).
//
// The result is written to the state's second buffer (neighbouring tiles read
// the old apron while we write), then the buffers are swapped.
#include "common.h"
#include "stencil.h"

namespace pyro {

#ifndef PYRO_ADV_TI
#define PYRO_ADV_TI 16
#endif
constexpr int ADV_TI = PYRO_ADV_TI;   // tile rows   (i, slow axis)
constexpr int ADV_TJ = 64;   // tile columns (j, fast axis) = one wave
constexpr int ADV_H = 3;     // apron
constexpr int ADV_AW = ADV_TJ + 2 * ADV_H;       // 70
constexpr int ADV_AH = ADV_TI + 2 * ADV_H;       // 22
constexpr int ADV_XW = ADV_TJ + 2;               // a_x: j in [j0-1, j0+TJ]
constexpr int ADV_XH = ADV_TI + 1;               //      i in [i0, i0+TI]
constexpr int ADV_YW = ADV_TJ + 1;               // a_y: j in [j0, j0+TJ]
constexpr int ADV_YH = ADV_TI + 2;               //      i in [i0-1, i0+TI]
#ifndef PYRO_ADV_THREADS
#define PYRO_ADV_THREADS 256
#endif
constexpr int ADV_THREADS = PYRO_ADV_THREADS;

struct AdvParams {
    double u, v, dt, dx, dy;
    int limiter;
    // uniform quotients, evaluated once on the host with the reference's
    // expressions (IEEE double on both sides: same bits); a division is ~14
    // VALU instructions per thread otherwise
    double cx, cy;          // u*dt/dx, v*dt/dy          interface.py:10-11
    double dtdx2, dtdy2;    // 0.5*dt/dx, 0.5*dt/dy      advective_fluxes.py:60-61
    double dtdx, dtdy;      // dt/dx, dt/dy              simulation.py:63-64
};

// LIM: limiter (0 none, 1 MC2, 2 MC4); UNEG / VNEG: u < 0 / v < 0 (upwind side)
//
// Thread layout: 4 waves; wave w, lane l.  Every phase walks rows w, w+4, ...
// with the lane as the column (no integer division, one LDS address add per
// row); the few columns beyond 64 (apron of A, the two extra face columns of
// a_x, the extra one of a_y) are a short second pass of a partial wave.
template <int LIM, bool UNEG, bool VNEG>
__global__ __launch_bounds__(ADV_THREADS) void k_adv_step(const double *__restrict__ ain,
                                                          double *__restrict__ aout, Geom g,
                                                          AdvParams P, int ntj, int ntiles)
{
    static_assert(ADV_THREADS == 256 && ADV_TJ == 64, "4 waves, lane = column");
    // rows padded to 4 * NR so that every wave stores all the rows it loaded
    // (no predicate the compiler could sink a load under)
    __shared__ double A[((ADV_AH + 3) / 4) * 4][ADV_AW];
    __shared__ double AX[ADV_XH][ADV_XW];
    __shared__ double AY[ADV_YH][ADV_YW];
    const int tile = xcd_tile(blockIdx.x, ntiles);
    const int i0 = g.ilo + (tile / ntj) * ADV_TI;
    const int j0 = g.jlo + (tile % ntj) * ADV_TJ;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int p = g.pitch;

    // ---- phase 0: stage a (tile + apron) in LDS -------------------------
    // Every load of the thread (up to 6 tile rows + 1 apron cell) is issued
    // before the first LDS store: written as load/store loops, each iteration
    // waited for its own load -- seven dependent HBM round trips per workgroup.
    constexpr int NR = (ADV_AH + 3) / 4;          // rows per wave
    double av[NR];
#pragma unroll
    for (int n = 0; n < NR; n++) {
        // columns H .. H+63 (the tile's own, 512-byte aligned rows) by all lanes
        int i = i0 - ADV_H + w + 4 * n, j = j0 + l;
        i = (i < g.qx) ? i : g.qx - 1;   // partial tiles / rows beyond the apron: clamp (unused)
        j = (j < g.qy) ? j : g.qy - 1;
        av[n] = ain[(size_t)i * p + j];
    }
    // the 2 x H apron columns: ADV_AH rows x 6 columns, one cell per thread
    static_assert(ADV_AH * 2 * ADV_H <= ADV_THREADS, "one apron cell per thread");
    static_assert(ADV_THREADS <= 2 * ADV_AH * 2 * ADV_H, "apron cell index wraps once");
    // threads beyond the apron cells repeat one of them (same value to the same
    // LDS word): no predicate the compiler could sink the load under
    const int at = (tid < ADV_AH * 2 * ADV_H) ? tid : tid - ADV_AH * 2 * ADV_H;
    const int apr = at / (2 * ADV_H), apk = at - apr * (2 * ADV_H);
    const int apc = (apk < ADV_H) ? apk : ADV_TJ + apk;          // 0..2, 67..69
    double apv;
    {
        int i = i0 - ADV_H + apr, j = j0 - ADV_H + apc;
        i = (i < g.qx) ? i : g.qx - 1;
        j = (j < g.qy) ? j : g.qy - 1;
        apv = ain[(size_t)i * p + j];
    }
#pragma unroll
    for (int n = 0; n < NR; n++) {
        A[w + 4 * n][ADV_H + l] = av[n];
    }
    A[apr][apc] = apv;
    __syncthreads();

    const double u = P.u, v = P.v;
    const double cx = P.cx, cy = P.cy;

    // ---- phase 1: upwind interface states (interface.py:25-41) ----------
    // a_x at faces i in [i0, i0+TI], j in [j0-1, j0+TJ]: AX[r][c], c = 0..65
    auto ax_state = [&](int r, int c) {
        // A-tile coordinates of the upwind cell of face (i0 + r, j0 - 1 + c)
        const int br = r + ADV_H - (UNEG ? 0 : 1), ac = c - 1 + ADV_H;
        const double a0 = A[br][ac];
        const double ld = limited_slope(A[br - 2][ac], A[br - 1][ac], a0, A[br + 1][ac],
                                        A[br + 2][ac], LIM);
        AX[r][c] = UNEG ? a0 - 0.5 * (1.0 + cx) * ld : a0 + 0.5 * (1.0 - cx) * ld;
    };
    for (int r = w; r < ADV_XH; r += 4) ax_state(r, l + 1);          // columns 1..64
    if (tid < 2 * ADV_XH) ax_state(tid >> 1, (tid & 1) ? ADV_XW - 1 : 0);   // columns 0, 65
    // a_y at faces i in [i0-1, i0+TI], j in [j0, j0+TJ]: AY[r][c], c = 0..64
    auto ay_state = [&](int r, int c) {
        const int ar = r - 1 + ADV_H, bc = c + ADV_H - (VNEG ? 0 : 1);
        const double a0 = A[ar][bc];
        const double ld = limited_slope(A[ar][bc - 2], A[ar][bc - 1], a0, A[ar][bc + 1],
                                        A[ar][bc + 2], LIM);
        AY[r][c] = VNEG ? a0 - 0.5 * (1.0 + cy) * ld : a0 + 0.5 * (1.0 - cy) * ld;
    };
    for (int r = w; r < ADV_YH; r += 4) ay_state(r, l);              // columns 0..63
    if (tid >= 64 && tid < 64 + ADV_YH) ay_state(tid - 64, ADV_YW - 1);   // column 64
    __syncthreads();

    // ---- phase 2: transverse-corrected fluxes + conservative update -----
    const int mx = (u <= 0) ? 0 : -1;   // advective_fluxes.py:71-79
    const int my = (v <= 0) ? 0 : -1;
    const double dtdx2 = P.dtdx2, dtdy2 = P.dtdy2;
    const double dtdx = P.dtdx, dtdy = P.dtdy;      // simulation.py:63-64

    const int c = l;
    for (int r = w; r < ADV_TI; r += 4) {
        const int i = i0 + r, j = j0 + c;
        if (i > g.ihi || j > g.jhi) continue;
        // AX[r][c+1] is a_x at (i, j);  AY[r+1][c] is a_y at (i, j)
        // F_x[i,j] = u*(a_x[i,j] - dtdy2*(F_yt[i+mx,j+1] - F_yt[i+mx,j]))
        double Fx0 = u * (AX[r][c + 1] -
                          dtdy2 * (v * AY[r + 1 + mx][c + 1] - v * AY[r + 1 + mx][c]));
        double Fx1 = u * (AX[r + 1][c + 1] -
                          dtdy2 * (v * AY[r + 2 + mx][c + 1] - v * AY[r + 2 + mx][c]));
        // F_y[i,j] = v*(a_y[i,j] - dtdx2*(F_xt[i+1,j+my] - F_xt[i,j+my]))
        double Fy0 = v * (AY[r + 1][c] -
                          dtdx2 * (u * AX[r + 1][c + 1 + my] - u * AX[r][c + 1 + my]));
        double Fy1 = v * (AY[r + 1][c + 1] -
                          dtdx2 * (u * AX[r + 1][c + 2 + my] - u * AX[r][c + 2 + my]));
        double a = A[r + ADV_H][c + ADV_H];
        aout[(size_t)i * p + j] = a + dtdx * (Fx0 - Fx1) + dtdy * (Fy0 - Fy1);
    }
}

template <int LIM>
static void adv_launch(pyrohip_ctx *c, bool uneg, bool vneg, int ntiles, const double *cur,
                       double *nxt, const Geom &g, const AdvParams &P, int ntj)
{
    const dim3 grid(ntiles), block(ADV_THREADS);
    if (uneg && vneg)
        PYRO_LAUNCH(c, "k_adv_step", (k_adv_step<LIM, true, true>), grid, block, 0, cur, nxt, g, P, ntj, ntiles);
    else if (uneg)
        PYRO_LAUNCH(c, "k_adv_step", (k_adv_step<LIM, true, false>), grid, block, 0, cur, nxt, g, P, ntj, ntiles);
    else if (vneg)
        PYRO_LAUNCH(c, "k_adv_step", (k_adv_step<LIM, false, true>), grid, block, 0, cur, nxt, g, P, ntj, ntiles);
    else
        PYRO_LAUNCH(c, "k_adv_step", (k_adv_step<LIM, false, false>), grid, block, 0, cur, nxt, g, P, ntj, ntiles);
}

// copy the ghost frame of one plane from src to dst (keeps the "stale ghost"
// semantics of the reference's in-place update).  O(perimeter): blockIdx.y
// enumerates the 2*ng ghost rows (all j), then groups of interior rows (only
// their 2*ng ghost columns).
__global__ void k_copy_frame(const double *__restrict__ src, double *__restrict__ dst, Geom g)
{
    const int ng = g.ng;
    const int b = blockIdx.y;
    int i, j;
    if (b < 2 * ng) {
        i = (b < ng) ? b : g.ihi + 1 + (b - ng);
        j = blockIdx.x * blockDim.x + threadIdx.x;
        if (j >= g.qy) return;
    } else {
        if (blockIdx.x != 0) return;
        const int t = threadIdx.x;
        const int rows_per_block = 256 / (2 * ng);
        const int r = (b - 2 * ng) * rows_per_block + t / (2 * ng);
        const int kx = t % (2 * ng);
        if (r >= g.nx || t >= rows_per_block * 2 * ng) return;
        i = g.ilo + r;
        j = (kx < ng) ? kx : g.jhi + 1 + (kx - ng);
    }
    dst[(size_t)i * g.pitch + j] = src[(size_t)i * g.pitch + j];
}

}  // namespace pyro

using namespace pyro;

extern "C" int pyrohip_adv_step(pyrohip_state *s, int n, double dx, double dy, double u, double v,
                                double dt, int limiter)
{
    PYRO_REQUIRE(s, "NULL state");
    PYRO_REQUIRE(n >= 0 && n < s->nvar, "variable index out of range");
    PYRO_REQUIRE(s->g.ng >= 4, "advection needs ng >= 4 (advection/simulation.py:20)");
    PYRO_REQUIRE(limiter >= 0 && limiter <= 2, "limiter must be 0, 1 or 2");
    pyrohip_ctx *c = s->ctx;
    const Geom &g = s->g;
    // scratch plane for the new time level
    if (s->work_planes < 1) {
        if (s->work) PYRO_CHECK_HIP(hipFree(s->work));
        s->work = nullptr;
        PYRO_CHECK_HIP(hipMalloc((void **)&s->work, (g.plane + 16) * sizeof(double)));
        s->work_planes = 1;
    }
    double *cur = s->d + (size_t)n * g.plane;
    double *nxt = s->work + geom_lead(g);
    AdvParams P;
    P.u = u; P.v = v; P.dt = dt; P.dx = dx; P.dy = dy; P.limiter = limiter;
    P.cx = u * dt / dx; P.cy = v * dt / dy;
    P.dtdx2 = 0.5 * dt / dx; P.dtdy2 = 0.5 * dt / dy;
    P.dtdx = dt / dx; P.dtdy = dt / dy;
    const int nti = (g.nx + ADV_TI - 1) / ADV_TI, ntj = (g.ny + ADV_TJ - 1) / ADV_TJ;
    const int ntiles = nti * ntj;
    const bool uneg = (u < 0), vneg = (v < 0);   // interface.py:28,38
    if (limiter == 0) adv_launch<0>(c, uneg, vneg, ntiles, cur, nxt, g, P, ntj);
    else if (limiter == 1) adv_launch<1>(c, uneg, vneg, ntiles, cur, nxt, g, P, ntj);
    else adv_launch<2>(c, uneg, vneg, ntiles, cur, nxt, g, P, ntj);
    // interior back into the state plane: swap roles by copying the interior
    // is avoided -- instead copy the (tiny) ghost frame into the new buffer
    // and exchange the two planes' contents by pointer where possible.
    {
        const int rows_per_block = 256 / (2 * g.ng);
        const int nby = 2 * g.ng + (g.nx + rows_per_block - 1) / rows_per_block;
        hipLaunchKernelGGL(k_copy_frame, dim3((g.qy + 255) / 256, nby), dim3(256), 0, c->stream,
                           (const double *)cur, nxt, g);
    }
    PYRO_CHECK_HIP(hipGetLastError());
    if (s->nvar == 1) {
        // single-variable state: swap the two allocations
        double *old_base = s->base;
        s->base = s->work;
        s->work = old_base;
        s->d = s->base + geom_lead(g);
    } else {
        PYRO_CHECK_HIP(hipMemcpyAsync(cur, nxt, g.plane * sizeof(double),
                                      hipMemcpyDeviceToDevice, c->stream));
    }
    return 0;
}
