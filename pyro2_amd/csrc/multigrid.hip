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
#include "reduce.h"

namespace pyro {

constexpr int MG_MAXLEV = 24;

struct MGLevel {
    int n;          // interior cells per side
    int pitch;
    double dx;
    double *v, *f, *r;
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
    double *bcval[4] = {nullptr, nullptr, nullptr, nullptr};  // device
    double source_norm = 0.0;
};

namespace pyro {

__device__ __forceinline__ double ghost_lo(int code, double inner, const double *val, int idx,
                                           double dx)
{
    switch (code) {
    case PYROHIP_BC_OUTFLOW: return val ? inner - dx * val[idx] : inner;          // neumann
    case PYROHIP_BC_REFLECT_ODD: return val ? 2 * val[idx] - inner : -inner;      // dirichlet
    default: return inner;                                                       // reflect-even
    }
}
__device__ __forceinline__ double ghost_hi(int code, double inner, const double *val, int idx,
                                           double dx)
{
    switch (code) {
    case PYROHIP_BC_OUTFLOW: return val ? inner + dx * val[idx] : inner;
    case PYROHIP_BC_REFLECT_ODD: return val ? 2 * val[idx] - inner : -inner;
    default: return inner;
    }
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

static double *plane(pyrohip_mg *m, int level, int var)
{
    MGLevel &L = m->lev[level];
    return var == 0 ? L.v : var == 1 ? L.f : L.r;
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

static int mg_smooth(pyrohip_mg *m, int level, int nsmooth)
{
    MGLevel &L = m->lev[level];
    MGBC bc = make_bc(m, level, true);
    PYRO_TRY(mg_fill(m, level, 0));                       // MG.py:565
    const double xcoeff = m->beta / (L.dx * L.dx);        // :567-568
    const double ycoeff = m->beta / (L.dx * L.dx);
    const double denom = m->alpha + 2.0 * xcoeff + 2.0 * ycoeff;
    const int half = (L.n + 1) / 2;
    const int bx = (half >= 256) ? 256 : 64;
    dim3 grid((half + bx - 1) / bx, L.n), block(bx);
    for (int it = 0; it < nsmooth; it++)
        for (int colour = 0; colour < 2; colour++)
            PYRO_LAUNCH(m->ctx, "k_mg_smooth", k_mg_smooth, grid, block, 0, L.v,
                               (const double *)L.f, L.n, L.pitch, L.dx, xcoeff, ycoeff, denom,
                               colour, bc);
    return 0;
}

static int mg_residual(pyrohip_mg *m, int level)
{
    MGLevel &L = m->lev[level];
    const int bx = (L.n >= 256) ? 256 : 64;
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

static int mg_zero(pyrohip_mg *m, int level, int var)
{
    MGLevel &L = m->lev[level];
    // patch.py:562-573 zeroes the whole array incl. ghosts
    PYRO_CHECK_HIP(hipMemsetAsync(plane(m, level, var), 0,
                                  (size_t)(L.n + 2) * L.pitch * sizeof(double), m->ctx->stream));
    return 0;
}

// returns sum (not sqrt) into host *out
static int mg_sumsq(pyrohip_mg *m, const double *a, const double *b, int level, int mode,
                    double *out)
{
    pyrohip_ctx *c = m->ctx;
    MGLevel &L = m->lev[level];
    dim3 grid(L.n >= 2048 ? 8 : 1, L.n >= 64 ? 64 : 1), block(256);
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

static int mg_vcycle(pyrohip_mg *m, int level)
{
    if (level > 0) {
        PYRO_TRY(mg_smooth(m, level, m->nsmooth));       // MG.py:722
        PYRO_TRY(mg_residual(m, level));                  // :724
        PYRO_TRY(mg_restrict(m, level));                  // :731-732
        PYRO_TRY(mg_vcycle(m, level - 1));                // :735
        PYRO_TRY(mg_prolong_add(m, level));               // :745-748
        PYRO_TRY(mg_fill(m, level, 0));                   // :751
        PYRO_TRY(mg_smooth(m, level, m->nsmooth));        // :758
    } else {
        PYRO_TRY(mg_smooth(m, level, m->nsmooth_bottom)); // :776
        PYRO_TRY(mg_fill(m, level, 0));                   // :778
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
                         bc[s] == PYROHIP_BC_PERIODIC || bc[s] == PYROHIP_BC_REFLECT_EVEN,
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
        total += 3 * g.plane + 16;
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
    for (int s = 0; s < 4; s++)
        if (m->bcval[s]) (void)hipFree(m->bcval[s]);
    delete m;
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
    PYRO_REQUIRE(var >= 0 && var <= 2 && host, "bad var / NULL host");
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

int pyrohip_mg_solve(pyrohip_mg *m, double rtol, int max_cycles, int *num_cycles,
                     double *residual_error, double *relative_error)
{
    PYRO_REQUIRE(m, "NULL mg");
    pyrohip_ctx *c = m->ctx;
    const int Lf = m->nlevels - 1;
    MGLevel &F = m->lev[Lf];
    const size_t fbytes = (size_t)(F.n + 2) * F.pitch * sizeof(double);
    PYRO_CHECK_HIP(hipMemcpyAsync(m->old_phi, F.v, fbytes, hipMemcpyDeviceToDevice, c->stream));
    double res = 1.e33, rel = 1.e33;
    int cycle = 1;
    while (res > rtol && cycle <= max_cycles) {           // MG.py:652
        for (int l = 0; l < Lf; l++) PYRO_TRY(mg_zero(m, l, 0));   // :658-659
        PYRO_TRY(mg_vcycle(m, Lf));
        double s = 0.0;                                   // :673-676
        PYRO_TRY(mg_sumsq(m, F.v, m->old_phi, Lf, 1, &s));
        rel = sqrt(F.dx * F.dx * s);
        PYRO_CHECK_HIP(hipMemcpyAsync(m->old_phi, F.v, fbytes, hipMemcpyDeviceToDevice,
                                      c->stream));
        PYRO_TRY(mg_residual(m, Lf));                     // :678
        PYRO_TRY(mg_sumsq(m, F.r, nullptr, Lf, 0, &s));
        double rn = sqrt(F.dx * F.dx * s);
        res = (m->source_norm != 0.0) ? rn / m->source_norm : rn;   // :682-685
        cycle++;
    }
    PYRO_TRY(mg_fill(m, Lf, 0));                          // :697
    PYRO_CHECK_HIP(hipGetLastError());
    PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));
    if (num_cycles) *num_cycles = cycle - 1;
    if (residual_error) *residual_error = res;
    if (relative_error) *relative_error = rel;
    return 0;
}

}  // extern "C"
