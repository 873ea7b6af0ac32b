// Context, cell-centred data storage, host<->device transfers, ghost fill,
// reductions.  Replaces the storage / data-movement half of
// pyro/mesh/patch.py (CellCenterData2d) and ArrayIndexer.fill_ghost
// (pyro/mesh/array_indexer.py:150-274).
#include "common.h"
#include "reduce.h"

namespace pyro {

static thread_local std::string g_last_error;
void set_error(const std::string &msg) { g_last_error = msg; }

int DevBuf::ensure(size_t need)
{
    if (need <= bytes) return 0;
    if (p) PYRO_CHECK_HIP(hipFree(p));
    p = nullptr; bytes = 0;
    PYRO_CHECK_HIP(hipMalloc(&p, need));
    bytes = need;
    return 0;
}
void DevBuf::release()
{
    if (p) (void)hipFree(p);
    p = nullptr; bytes = 0;
}

// ---------------------------------------------------------------------------
// AoS (rows, qy, nvar) staging  <->  planar SoA
// one thread per (row, j); it moves the nvar contiguous doubles of a cell
// ---------------------------------------------------------------------------
__global__ void k_aos_to_planar(const double *__restrict__ aos, double *__restrict__ d,
                                int i0, int ni, int qy, int nvar, int pitch, size_t plane)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    int r = blockIdx.y;
    if (j >= qy || r >= ni) return;
    const double *src = aos + ((size_t)r * qy + j) * nvar;
    double *dst = d + (size_t)(i0 + r) * pitch + j;
    for (int n = 0; n < nvar; n++) dst[n * plane] = src[n];
}

__global__ void k_planar_to_aos(const double *__restrict__ d, double *__restrict__ aos,
                                int i0, int ni, int qy, int nvar, int pitch, size_t plane)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    int r = blockIdx.y;
    if (j >= qy || r >= ni) return;
    double *dst = aos + ((size_t)r * qy + j) * nvar;
    const double *src = d + (size_t)(i0 + r) * pitch + j;
    for (int n = 0; n < nvar; n++) dst[n] = src[n * plane];
}

// ---------------------------------------------------------------------------
// ghost fill.  x pass: rows i < ilo and i > ihi over ALL j (ghost columns
// included); y pass afterwards over ALL i, so corners come from x-filled data
// exactly as in array_indexer.py:163-274.
// grid.z = variable.  bc = nvar*4 codes on the device.
// ---------------------------------------------------------------------------
__global__ void k_fill_x(double *__restrict__ d, Geom g, const int *__restrict__ bc, int n0)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    int n = n0 + blockIdx.z;
    if (j >= g.qy) return;
    double *a = d + (size_t)n * g.plane;
    const int bl = bc[n * 4 + 0], br = bc[n * 4 + 1];
    const int ng = g.ng, ilo = g.ilo, ihi = g.ihi;
    const int p = g.pitch;
    if (bl != PYROHIP_BC_HALO && bl != PYROHIP_BC_RAMP)   // ramp: k_fill_ramp
        for (int i = 0; i < ilo; i++) {
            double v;
            switch (bl) {
            case PYROHIP_BC_OUTFLOW: v = a[(size_t)ilo * p + j]; break;
            case PYROHIP_BC_REFLECT_EVEN: v = a[(size_t)(2 * ng - i - 1) * p + j]; break;
            case PYROHIP_BC_REFLECT_ODD: v = -a[(size_t)(2 * ng - i - 1) * p + j]; break;
            default: v = a[(size_t)(ihi - ng + i + 1) * p + j]; break;  // periodic
            }
            a[(size_t)i * p + j] = v;
        }
    if (br != PYROHIP_BC_HALO)
        for (int k = 0; k < ng; k++) {
            int i = ihi + 1 + k;
            double v;
            switch (br) {
            case PYROHIP_BC_OUTFLOW: v = a[(size_t)ihi * p + j]; break;
            case PYROHIP_BC_REFLECT_EVEN: v = a[(size_t)(ihi - k) * p + j]; break;
            case PYROHIP_BC_REFLECT_ODD: v = -a[(size_t)(ihi - k) * p + j]; break;
            default: v = a[(size_t)(i - ihi - 1 + ng) * p + j]; break;  // periodic
            }
            a[(size_t)i * p + j] = v;
        }
}

// threads: x = ghost index k in [0, 2*ng) (contiguous in memory per side),
//          y = row i
__global__ void k_fill_y(double *__restrict__ d, Geom g, const int *__restrict__ bc, int n0,
                         const double *__restrict__ cval)
{
    int k = threadIdx.x;
    int i = blockIdx.x * blockDim.y + threadIdx.y;
    int n = n0 + blockIdx.z;
    if (i >= g.qx || k >= 2 * g.ng) return;
    double *a = d + (size_t)n * g.plane + (size_t)i * g.pitch;
    const int ng = g.ng, jlo = g.jlo, jhi = g.jhi;
    if (k < ng) {
        int j = k;
        double v;
        switch (bc[n * 4 + 2]) {
        case PYROHIP_BC_HSE:   // BC.py:54-62 (energy is overwritten by k_fill_y_user)
        case PYROHIP_BC_OUTFLOW: v = a[jlo]; break;
        case PYROHIP_BC_REFLECT_EVEN: v = a[2 * ng - j - 1]; break;
        case PYROHIP_BC_REFLECT_ODD: v = -a[2 * ng - j - 1]; break;
        case PYROHIP_BC_PERIODIC: v = a[jhi - ng + j + 1]; break;
        default: return;
        }
        a[j] = v;
    } else {
        int kk = k - ng, j = jhi + 1 + kk;
        double v;
        switch (bc[n * 4 + 3]) {
        case PYROHIP_BC_HSE:      // BC.py:87-93
        case PYROHIP_BC_AMBIENT:  // BC.py:159-160
        case PYROHIP_BC_OUTFLOW: v = a[jhi]; break;
        case PYROHIP_BC_REFLECT_EVEN: v = a[jhi - kk]; break;
        case PYROHIP_BC_REFLECT_ODD: v = -a[jhi - kk]; break;
        case PYROHIP_BC_PERIODIC: v = a[j - jhi - 1 + ng]; break;
        case PYROHIP_BC_CONST: v = cval[n]; break;   // incompressible_viscous/BC.py:31-42
        default: return;
        }
        a[j] = v;
    }
}

// User boundaries of the compressible solver (compressible/BC.py:21-176) for
// variable n of a 4-plane conserved state; one thread per row i and side.
//   hse, energy only (:64-84 lower, :95-115 upper): integrate dp = rho g dy
//   away from the last interior cell at constant density, keep its kinetic
//   energy.  ambient (:147-176, upper side): constant state.
struct UserBc {
    double gamma, grav, dy, amb_val;
};
__global__ void k_fill_y_user(double *__restrict__ d, Geom g, const int *__restrict__ bc, int n,
                              UserBc ub)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int side = blockIdx.y;
    if (i >= g.qx) return;
    const int code = bc[n * 4 + 2 + side];
    const size_t row = (size_t)i * g.pitch;
    if (code == PYROHIP_BC_AMBIENT && side == 1) {
        double *a = d + (size_t)n * g.plane + row;
        for (int j = g.jhi + 1; j < g.qy; j++) a[j] = ub.amb_val;
        return;
    }
    if (code != PYROHIP_BC_HSE || n != 1) return;
    const int jb = side ? g.jhi : g.jlo;
    const double dens_base = d[row + jb];
    const double ener = d[(size_t)g.plane + row + jb];
    const double xm = d[2 * (size_t)g.plane + row + jb], ym = d[3 * (size_t)g.plane + row + jb];
    const double ke_base = 0.5 * (xm * xm + ym * ym) / dens_base;
    const double eint_base = (ener - ke_base) / dens_base;
    double pres_base = dens_base * eint_base * (ub.gamma - 1.0);   // eos.pres
    double *e = d + (size_t)g.plane + row;
    for (int k = 1; k <= g.ng; k++) {
        const int j = side ? g.jhi + k : g.jlo - k;
        const double pres_next = side ? pres_base + ub.grav * dens_base * ub.dy
                                      : pres_base - ub.grav * dens_base * ub.dy;
        e[j] = pres_next / (ub.gamma - 1.0) + ke_base;             // eos.rhoe
        pres_base = pres_next;
    }
}

// Runge-Kutta stage combination (pyro/mesh/integration.py:84-113): dst <- src
// on the whole array (cell_center_data_clone), then on the interior
// dst += c[0] k_0; dst += c[1] k_1; ... in this order, k_s = planes
// nvar*s .. nvar*s + nvar - 1 of the k state.  dst may be src.
struct LinComb { double c[8]; int n; };
__global__ __launch_bounds__(256) void k_lincomb(double *__restrict__ dst,
                                                 const double *__restrict__ src,
                                                 const double *__restrict__ K, Geom g, int nvar,
                                                 LinComb lc)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y, n = blockIdx.z;
    if (j >= g.qy) return;
    const size_t k = (size_t)i * g.pitch + j;
    double v = src[(size_t)n * g.plane + k];
    if (i >= g.ilo && i <= g.ihi && j >= g.jlo && j <= g.jhi)
        for (int s = 0; s < lc.n; s++) v += lc.c[s] * K[(size_t)(nvar * s + n) * g.plane + k];
    dst[(size_t)n * g.plane + k] = v;
}

// "ramp" boundary of the double Mach reflection problem, compressible/BC.py:
// 178-296, for conserved variable n.  pass 0: lower x side = post-shock inflow
// for all j (:186-191).  pass 1: lower y side (:199-214: inflow for x < 1/6,
// reflecting wall beyond -- odd for the y-momentum) and upper y side (:224-246:
// the moving shock, sub-sampled at four points per cell), for all i.
struct RampBc {
    double post, pre, cxoff;
    double sfd[8], sfu[8];
    int odd;   // variable is the y-momentum
};
__global__ void k_fill_ramp(double *__restrict__ a, Geom g, const double *__restrict__ x,
                            int bxl, int byl, int byr, int pass, RampBc rb)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (pass == 0) {
        if (bxl != PYROHIP_BC_RAMP || t >= g.qy) return;
        for (int i = 0; i < g.ilo; i++) a[(size_t)i * g.pitch + t] = rb.post;
        return;
    }
    if (t >= g.qx) return;
    double *row = a + (size_t)t * g.pitch;
    const double xc = x[t];
    if (byl == PYROHIP_BC_RAMP)
        for (int j = g.jlo - 1; j >= 0; j--) {
            if (xc < 1.0 / 6.0) row[j] = rb.post;
            else {
                const double m = row[2 * g.ng - j - 1];   // row jlo + jj
                row[j] = rb.odd ? -1.0 * m : m;
            }
        }
    if (byr == PYROHIP_BC_RAMP) {
        const double cx[2] = {xc - rb.cxoff, xc + rb.cxoff};
        for (int k = 0; k < g.ng; k++) {
            const double sf[2] = {rb.sfd[k], rb.sfu[k]};
            double v = 0.0;
#pragma unroll
            for (int s = 0; s < 2; s++)
#pragma unroll
                for (int c = 0; c < 2; c++) v = v + 0.25 * ((cx[c] < sf[s]) ? rb.post : rb.pre);
            row[g.jhi + 1 + k] = v;
        }
    }
}

// ---------------------------------------------------------------------------
// min / max over a rectangular region of one plane (two-pass, deterministic)
// ---------------------------------------------------------------------------
__global__ void k_minmax(const double *__restrict__ a, int pitch, int i0, int i1, int j0, int j1,
                         double *__restrict__ partial)
{
    double mn = INFINITY, mx = -INFINITY;
    for (int i = i0 + blockIdx.y; i <= i1; i += gridDim.y)
        for (int j = j0 + blockIdx.x * blockDim.x + threadIdx.x; j <= j1;
             j += gridDim.x * blockDim.x) {
            double v = a[(size_t)i * pitch + j];
            mn = fmin(mn, v);
            mx = fmax(mx, v);
        }
    mn = block_reduce_min(mn);
    mx = block_reduce_max(mx);
    if (threadIdx.x == 0) {
        int b = blockIdx.y * gridDim.x + blockIdx.x;
        partial[2 * b] = mn;
        partial[2 * b + 1] = mx;
    }
}

__global__ void k_minmax_final(const double *__restrict__ partial, int nb, double *__restrict__ out)
{
    double mn = INFINITY, mx = -INFINITY;
    for (int b = threadIdx.x; b < nb; b += blockDim.x) {
        mn = fmin(mn, partial[2 * b]);
        mx = fmax(mx, partial[2 * b + 1]);
    }
    mn = block_reduce_min(mn);
    mx = block_reduce_max(mx);
    if (threadIdx.x == 0) { out[0] = mn; out[1] = mx; }
}

int fill_bc_planes(pyrohip_state *s, double *planes, int n0, int cnt)
{
    pyrohip_ctx *c = s->ctx;
    const Geom &g = s->g;
    hipLaunchKernelGGL(k_fill_x, dim3((g.qy + 255) / 256, 1, cnt), dim3(256), 0, c->stream, planes,
                       g, (const int *)s->d_bc, n0);
    hipLaunchKernelGGL(k_fill_y, dim3((g.qx + 15) / 16, 1, cnt), dim3(16, 16), 0, c->stream,
                       planes, g, (const int *)s->d_bc, n0, (const double *)s->d_cval);
    PYRO_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace pyro

using namespace pyro;

extern "C" {

const char *pyrohip_last_error(void) { return g_last_error.c_str(); }

#ifndef PYRO_BACKEND_NAME
#define PYRO_BACKEND_NAME "hip-gfx950"
#endif
const char *pyrohip_backend(void) { return PYRO_BACKEND_NAME; }

int pyrohip_device_count(int *count)
{
    PYRO_REQUIRE(count != nullptr, "count is NULL");
    PYRO_CHECK_HIP(hipGetDeviceCount(count));
    return 0;
}

int pyrohip_init(int device_id, pyrohip_ctx **out)
{
    PYRO_REQUIRE(out != nullptr, "out is NULL");
    int ndev = 0;
    PYRO_CHECK_HIP(hipGetDeviceCount(&ndev));
    if (ndev <= 0) { set_error("no HIP device visible"); return PYROHIP_ERR_UNSUPPORTED; }
    PYRO_REQUIRE(device_id >= 0 && device_id < ndev, "device_id out of range");
    PYRO_CHECK_HIP(hipSetDevice(device_id));
    pyrohip_ctx *c = new pyrohip_ctx();
    c->device = device_id;
    PYRO_CHECK_HIP(hipStreamCreate(&c->stream));
    PYRO_CHECK_HIP(hipEventCreate(&c->ev0));
    PYRO_CHECK_HIP(hipEventCreate(&c->ev1));
    PYRO_CHECK_HIP(hipHostMalloc(&c->reduce_host, 256, 0));
    hipDeviceProp_t prop;
    PYRO_CHECK_HIP(hipGetDeviceProperties(&prop, device_id));
    c->num_cus = prop.multiProcessorCount;
    *out = c;
    return 0;
}

int pyrohip_shutdown(pyrohip_ctx *c)
{
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    pyrohip_comm_destroy(c);
    c->staging.release();
    c->reduce.release();
    c->prio_board.release();
    if (c->reduce_host) (void)hipHostFree(c->reduce_host);
    (void)hipEventDestroy(c->ev0);
    (void)hipEventDestroy(c->ev1);
    (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

int pyrohip_sync(pyrohip_ctx *c)
{
    PYRO_REQUIRE(c, "ctx is NULL");
    PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int pyrohip_device_info(pyrohip_ctx *c, char *name, int name_len, size_t *free_bytes,
                        size_t *total_bytes, int *compute_units)
{
    PYRO_REQUIRE(c, "ctx is NULL");
    hipDeviceProp_t prop;
    PYRO_CHECK_HIP(hipGetDeviceProperties(&prop, c->device));
    if (name && name_len > 0) {
        std::string s = std::string(prop.name) + " (" + prop.gcnArchName + ")";
        strncpy(name, s.c_str(), name_len - 1);
        name[name_len - 1] = 0;
    }
    size_t f = 0, t = 0;
    PYRO_CHECK_HIP(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    if (compute_units) *compute_units = prop.multiProcessorCount;
    return 0;
}

int pyrohip_timer_start(pyrohip_ctx *c)
{
    PYRO_REQUIRE(c, "ctx is NULL");
    PYRO_CHECK_HIP(hipEventRecord(c->ev0, c->stream));
    return 0;
}

int pyrohip_timer_stop(pyrohip_ctx *c, double *elapsed_ms)
{
    PYRO_REQUIRE(c && elapsed_ms, "NULL argument");
    PYRO_CHECK_HIP(hipEventRecord(c->ev1, c->stream));
    PYRO_CHECK_HIP(hipEventSynchronize(c->ev1));
    float ms = 0.f;
    PYRO_CHECK_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *elapsed_ms = (double)ms;
    return 0;
}

int pyrohip_prof_enable(pyrohip_ctx *c, int on)
{
    PYRO_REQUIRE(c, "ctx is NULL");
    c->prof.on = (on != 0);
    return 0;
}

int pyrohip_prof_report(pyrohip_ctx *c, char *buf, int buf_len)
{
    PYRO_REQUIRE(c && buf && buf_len > 0, "bad argument");
    PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));
    struct Agg { const char *name; int n; double ms; };
    std::vector<Agg> agg;
    for (auto &r : c->prof.recs) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, r.a, r.b);
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
        bool found = false;
        for (auto &a : agg)
            if (strcmp(a.name, r.name) == 0) { a.n++; a.ms += ms; found = true; break; }
        if (!found) agg.push_back(Agg{r.name, 1, (double)ms});
    }
    c->prof.recs.clear();
    std::string out;
    char line[256];
    for (auto &a : agg) {
        snprintf(line, sizeof line, "%s %d %.6f\n", a.name, a.n, a.ms);
        out += line;
    }
    strncpy(buf, out.c_str(), buf_len - 1);
    buf[buf_len - 1] = 0;
    return 0;
}

// --------------------------------------------------------------------------
int pyrohip_state_create(pyrohip_ctx *c, int nx, int ny, int ng, int nvar, const int *bc,
                         pyrohip_state **out)
{
    PYRO_REQUIRE(c && out && bc, "NULL argument");
    PYRO_REQUIRE(nx > 0 && ny > 0 && ng >= 1 && ng <= 8 && nvar >= 1, "bad dimensions");
    PYRO_REQUIRE(nx >= ng && ny >= ng, "grid smaller than the ghost width");
    for (int k = 0; k < nvar * 4; k++)
        PYRO_REQUIRE(bc[k] >= 0 && bc[k] <= PYROHIP_BC_CONST, "bad BC code");
    for (int k = 0; k < nvar * 4; k++)   // incompressible_viscous/BC.py:44-45
        PYRO_REQUIRE(bc[k] != PYROHIP_BC_CONST || (k & 3) == 3,
                     "the constant-value boundary is only defined on the upper y side");
    bool user_bc = false, ramp_bc = false;
    for (int k = 0; k < nvar * 4; k++) {
        if (bc[k] != PYROHIP_BC_RAMP) continue;
        PYRO_REQUIRE(nvar == 4, "the ramp boundary needs the 4-variable compressible state");
        PYRO_REQUIRE((k & 3) != 1, "the ramp boundary is not defined on the upper x side");
        ramp_bc = true;
    }
    for (int k = 0; k < nvar * 4; k++) {
        if (bc[k] != PYROHIP_BC_HSE && bc[k] != PYROHIP_BC_AMBIENT) continue;
        // BC.py:116-117, 176-177: hse on the y sides only, ambient on the upper y side only
        PYRO_REQUIRE(nvar == 4, "hse / ambient boundaries need the 4-variable compressible state");
        PYRO_REQUIRE((k & 3) >= 2, "hse / ambient boundaries are not supported on the x sides");
        PYRO_REQUIRE(bc[k] == PYROHIP_BC_HSE || (k & 3) == 3,
                     "the ambient boundary is only supported on the upper y side");
        user_bc = true;
    }
    PYRO_CHECK_HIP(hipSetDevice(c->device));
    pyrohip_state *s = new pyrohip_state();
    s->user_bc = user_bc;
    s->ramp_bc = ramp_bc;
    s->ctx = c;
    s->g = make_geom(nx, ny, ng);
    s->nvar = nvar;
    s->bc.assign(bc, bc + nvar * 4);
    size_t n = s->g.plane * nvar + 16;
    PYRO_CHECK_HIP(hipMalloc((void **)&s->base, n * sizeof(double)));
    PYRO_CHECK_HIP(hipMemsetAsync(s->base, 0, n * sizeof(double), c->stream));
    s->d = s->base + geom_lead(s->g);
    PYRO_CHECK_HIP(hipMalloc((void **)&s->d_bc, sizeof(int) * nvar * 4));
    PYRO_CHECK_HIP(hipMemcpy(s->d_bc, bc, sizeof(int) * nvar * 4, hipMemcpyHostToDevice));
    PYRO_CHECK_HIP(hipMalloc((void **)&s->d_cval, sizeof(double) * nvar));
    PYRO_CHECK_HIP(hipMemsetAsync(s->d_cval, 0, sizeof(double) * nvar, c->stream));
    PYRO_CHECK_HIP(hipMalloc((void **)&s->d_flag, sizeof(int) * 4));
    PYRO_CHECK_HIP(hipMemsetAsync(s->d_flag, 0, sizeof(int) * 4, c->stream));
    *out = s;
    return 0;
}

int pyrohip_state_destroy(pyrohip_state *s)
{
    if (!s) return 0;
    (void)hipSetDevice(s->ctx->device);
    (void)comm_wait_halo(s);          // a receive may still be writing the ghost rows
    (void)hipStreamSynchronize(s->ctx->stream);
    if (s->d_scal) (void)hipFree(s->d_scal);
    if (s->d_polmem) (void)hipFree(s->d_polmem);
    if (s->d_dts) (void)hipFree(s->d_dts);
    if (s->base) (void)hipFree(s->base);
    if (s->alt_base) (void)hipFree(s->alt_base);
    if (s->d_bc) (void)hipFree(s->d_bc);
    if (s->d_x) (void)hipFree(s->d_x);
    if (s->heat_base) (void)hipFree(s->heat_base);
    if (s->d_flag) (void)hipFree(s->d_flag);
    if (s->d_cval) (void)hipFree(s->d_cval);
    if (s->sph) { if (s->sph->base) (void)hipFree(s->sph->base); delete s->sph; }
    if (s->work) (void)hipFree(s->work);
    delete s;
    return 0;
}

static const size_t kStageBytes = (size_t)256 << 20;

static int rows_per_chunk(const pyrohip_state *s)
{
    size_t row = (size_t)s->g.qy * s->nvar * sizeof(double);
    size_t r = kStageBytes / row;
    if (r < 1) r = 1;
    return (int)r;
}

int pyrohip_state_upload_rows(pyrohip_state *s, int i0, int ni, const double *host)
{
    PYRO_REQUIRE(s && host, "NULL argument");
    PYRO_REQUIRE(i0 >= 0 && ni >= 0 && i0 + ni <= s->g.qx, "row range out of bounds");
    pyrohip_ctx *c = s->ctx;
    PYRO_CHECK_HIP(hipSetDevice(c->device));
    PYRO_TRY(comm_wait_halo(s));
    const Geom &g = s->g;
    const int chunk = rows_per_chunk(s);
    const size_t row = (size_t)g.qy * s->nvar;
    for (int r0 = 0; r0 < ni; r0 += chunk) {
        int nr = (ni - r0 < chunk) ? ni - r0 : chunk;
        PYRO_TRY(c->staging.ensure(nr * row * sizeof(double)));
        PYRO_CHECK_HIP(hipMemcpyAsync(c->staging.p, host + (size_t)r0 * row,
                                      nr * row * sizeof(double), hipMemcpyHostToDevice,
                                      c->stream));
        dim3 grid((g.qy + 255) / 256, nr), block(256);
        hipLaunchKernelGGL(k_aos_to_planar, grid, block, 0, c->stream,
                           (const double *)c->staging.p, s->d, i0 + r0, nr, g.qy, s->nvar,
                           g.pitch, g.plane);
        PYRO_CHECK_HIP(hipGetLastError());
        PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));
    }
    s->next_cfl_min = -1.0;
    s->ghost_by_rules = false;
    return 0;
}

int pyrohip_state_download_rows(pyrohip_state *s, int i0, int ni, double *host)
{
    PYRO_REQUIRE(s && host, "NULL argument");
    PYRO_REQUIRE(i0 >= 0 && ni >= 0 && i0 + ni <= s->g.qx, "row range out of bounds");
    pyrohip_ctx *c = s->ctx;
    PYRO_CHECK_HIP(hipSetDevice(c->device));
    PYRO_TRY(comm_wait_halo(s));
    const Geom &g = s->g;
    const int chunk = rows_per_chunk(s);
    const size_t row = (size_t)g.qy * s->nvar;
    for (int r0 = 0; r0 < ni; r0 += chunk) {
        int nr = (ni - r0 < chunk) ? ni - r0 : chunk;
        PYRO_TRY(c->staging.ensure(nr * row * sizeof(double)));
        dim3 grid((g.qy + 255) / 256, nr), block(256);
        hipLaunchKernelGGL(k_planar_to_aos, grid, block, 0, c->stream, (const double *)s->d,
                           (double *)c->staging.p, i0 + r0, nr, g.qy, s->nvar, g.pitch,
                           g.plane);
        PYRO_CHECK_HIP(hipGetLastError());
        PYRO_CHECK_HIP(hipMemcpyAsync(host + (size_t)r0 * row, c->staging.p,
                                      nr * row * sizeof(double), hipMemcpyDeviceToHost,
                                      c->stream));
        PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));
    }
    return 0;
}

int pyrohip_state_upload(pyrohip_state *s, const double *host)
{
    PYRO_REQUIRE(s, "NULL state");
    return pyrohip_state_upload_rows(s, 0, s->g.qx, host);
}

int pyrohip_state_download(pyrohip_state *s, double *host)
{
    PYRO_REQUIRE(s, "NULL state");
    return pyrohip_state_download_rows(s, 0, s->g.qx, host);
}

int pyrohip_state_upload_var(pyrohip_state *s, int n, const double *host)
{
    PYRO_REQUIRE(s && host, "NULL argument");
    PYRO_REQUIRE(n >= 0 && n < s->nvar, "variable index out of range");
    pyrohip_ctx *c = s->ctx;
    PYRO_CHECK_HIP(hipSetDevice(c->device));
    PYRO_TRY(comm_wait_halo(s));
    const Geom &g = s->g;
    PYRO_CHECK_HIP(hipMemcpy2DAsync(s->d + (size_t)n * g.plane, g.pitch * sizeof(double), host,
                                    g.qy * sizeof(double), g.qy * sizeof(double), g.qx,
                                    hipMemcpyHostToDevice, c->stream));
    PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));
    s->next_cfl_min = -1.0;
    s->ghost_by_rules = false;
    return 0;
}

int pyrohip_state_download_var(pyrohip_state *s, int n, double *host)
{
    PYRO_REQUIRE(s && host, "NULL argument");
    PYRO_REQUIRE(n >= 0 && n < s->nvar, "variable index out of range");
    pyrohip_ctx *c = s->ctx;
    PYRO_CHECK_HIP(hipSetDevice(c->device));
    PYRO_TRY(comm_wait_halo(s));
    const Geom &g = s->g;
    PYRO_CHECK_HIP(hipMemcpy2DAsync(host, g.qy * sizeof(double), s->d + (size_t)n * g.plane,
                                    g.pitch * sizeof(double), g.qy * sizeof(double), g.qx,
                                    hipMemcpyDeviceToHost, c->stream));
    PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int pyrohip_state_set_user_bc(pyrohip_state *s, double gamma, double grav, double dy,
                              const double *ambient)
{
    PYRO_REQUIRE(s, "NULL state");
    PYRO_REQUIRE(s->nvar == 4, "user boundaries need the 4-variable compressible state");
    s->ubc_gamma = gamma;
    s->ubc_grav = grav;
    s->ubc_dy = dy;
    for (int k = 0; k < 4; k++) s->ubc_amb[k] = ambient ? ambient[k] : 0.0;
    s->user_bc_set = true;
    return 0;
}

int pyrohip_state_set_heating(pyrohip_state *s, const double *profile)
{
    PYRO_REQUIRE(s, "NULL state");
    PYRO_REQUIRE(s->nvar == 4, "the heating source needs the 4-variable compressible state");
    pyrohip_ctx *c = s->ctx;
    const Geom &g = s->g;
    if (!profile) {
        PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));
        if (s->heat_base) PYRO_CHECK_HIP(hipFree(s->heat_base));
        s->heat_base = s->heat = nullptr;
        return 0;
    }
    if (!s->heat_base) {
        PYRO_CHECK_HIP(hipMalloc((void **)&s->heat_base, (g.plane + 16) * sizeof(double)));
        PYRO_CHECK_HIP(hipMemsetAsync(s->heat_base, 0, (g.plane + 16) * sizeof(double), c->stream));
        s->heat = s->heat_base + geom_lead(g);
    }
    PYRO_CHECK_HIP(hipMemcpy2DAsync(s->heat, g.pitch * sizeof(double), profile,
                                    g.qy * sizeof(double), g.qy * sizeof(double), g.qx,
                                    hipMemcpyHostToDevice, c->stream));
    // ghost cells like my_aux.fill_BC("E_src") (unsplit_fluxes.py:303-306):
    // the boundary types of the energy = row 1 of the BC table, used as "variable 0"
    // of this one-plane array
    hipLaunchKernelGGL(k_fill_x, dim3((g.qy + 255) / 256, 1, 1), dim3(256), 0, c->stream, s->heat, g,
                       (const int *)(s->d_bc + 4), 0);
    hipLaunchKernelGGL(k_fill_y, dim3((g.qx + 15) / 16, 1, 1), dim3(16, 16), 0, c->stream, s->heat,
                       g, (const int *)(s->d_bc + 4), 0, (const double *)nullptr);
    PYRO_CHECK_HIP(hipGetLastError());
    PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));   // profile is borrowed for the call only
    return 0;
}

int pyrohip_state_set_ramp_bc(pyrohip_state *s, const double *x, double cxoff, const double *post,
                              const double *pre, const double *sf_down, const double *sf_up)
{
    PYRO_REQUIRE(s && x && post && pre && sf_down && sf_up, "NULL argument");
    PYRO_REQUIRE(s->nvar == 4, "the ramp boundary needs the 4-variable compressible state");
    if (!s->d_x) PYRO_CHECK_HIP(hipMalloc((void **)&s->d_x, sizeof(double) * s->g.qx));
    PYRO_CHECK_HIP(hipMemcpyAsync(s->d_x, x, sizeof(double) * s->g.qx, hipMemcpyHostToDevice,
                                  s->ctx->stream));
    PYRO_CHECK_HIP(hipStreamSynchronize(s->ctx->stream));   // x is borrowed for the call only
    s->r_cxoff = cxoff;
    for (int n = 0; n < 4; n++) { s->r_post[n] = post[n]; s->r_pre[n] = pre[n]; }
    for (int k = 0; k < s->g.ng; k++) { s->r_sfd[k] = sf_down[k]; s->r_sfu[k] = sf_up[k]; }
    s->ramp_set = true;
    return 0;
}

static void launch_ramp(pyrohip_state *s, int n, int pass)
{
    const Geom &g = s->g;
    RampBc rb;
    rb.post = s->r_post[n]; rb.pre = s->r_pre[n]; rb.cxoff = s->r_cxoff;
    for (int k = 0; k < 8; k++) { rb.sfd[k] = s->r_sfd[k]; rb.sfu[k] = s->r_sfu[k]; }
    rb.odd = (n == 3);
    const int len = pass == 0 ? g.qy : g.qx;
    hipLaunchKernelGGL(k_fill_ramp, dim3((len + 63) / 64), dim3(64), 0, s->ctx->stream,
                       s->d + (size_t)n * g.plane, g, (const double *)s->d_x, s->bc[n * 4 + 0],
                       s->bc[n * 4 + 2], s->bc[n * 4 + 3], pass, rb);
}

// fill_BC for variables n0 .. n0+cnt-1: x sides, then y sides, then the
// user boundaries of each variable
static int fill_bc_range(pyrohip_state *s, int n0, int cnt)
{
    pyrohip_ctx *c = s->ctx;
    const Geom &g = s->g;
    {
        dim3 grid((g.qy + 255) / 256, 1, cnt), block(256);
        hipLaunchKernelGGL(k_fill_x, grid, block, 0, c->stream, s->d, g, (const int *)s->d_bc, n0);
    }
    if (s->ramp_bc)
        for (int n = n0; n < n0 + cnt; n++) launch_ramp(s, n, 0);
    {
        dim3 block(16, 16);
        dim3 grid((g.qx + 15) / 16, 1, cnt);
        hipLaunchKernelGGL(k_fill_y, grid, block, 0, c->stream, s->d, g, (const int *)s->d_bc, n0,
                           (const double *)s->d_cval);
    }
    if (s->ramp_bc)
        for (int n = n0; n < n0 + cnt; n++) launch_ramp(s, n, 1);
    if (s->user_bc) {
        for (int n = n0; n < n0 + cnt; n++) {
            const int yl = s->bc[n * 4 + 2], yr = s->bc[n * 4 + 3];
            const bool hse = (n == 1) && (yl == PYROHIP_BC_HSE || yr == PYROHIP_BC_HSE);
            const bool amb = (yr == PYROHIP_BC_AMBIENT);
            if (!hse && !amb) continue;
            const double *A = s->ubc_amb;   // rho, u, v, p
            UserBc ub{s->ubc_gamma, s->ubc_grav, s->ubc_dy, 0.0};
            // BC.py:162-176; n: density, energy, x-momentum, y-momentum
            if (n == 0) ub.amb_val = A[0];
            else if (n == 2) ub.amb_val = A[0] * A[1];
            else if (n == 3) ub.amb_val = A[0] * A[2];
            else {
                const double ke = 0.5 * A[0] * (A[1] * A[1] + A[2] * A[2]);
                ub.amb_val = A[3] / (s->ubc_gamma - 1.0) + ke;
            }
            hipLaunchKernelGGL(k_fill_y_user, dim3((g.qx + 63) / 64, 2), dim3(64), 0, c->stream,
                               s->d, g, (const int *)s->d_bc, n, ub);
        }
    }
    PYRO_CHECK_HIP(hipGetLastError());
    return 0;
}

int pyrohip_fill_bc(pyrohip_state *s, int n)
{
    PYRO_REQUIRE(s, "NULL state");
    PYRO_REQUIRE(n >= -1 && n < s->nvar, "variable index out of range");
    PYRO_TRY(comm_wait_halo(s));
    if (n >= 0) s->ghost_by_rules = false;     // (one variable: the others may be anything)
    if (s->ramp_bc) {
        PYRO_REQUIRE(s->ramp_set, "ramp boundary: call pyrohip_state_set_ramp_bc first");
        if (!s->user_bc) return fill_bc_range(s, n < 0 ? 0 : n, n < 0 ? s->nvar : 1);
    }
    if (s->user_bc) {
        PYRO_REQUIRE(s->user_bc_set, "hse / ambient boundary: call pyrohip_state_set_user_bc first");
        if (n >= 0) return fill_bc_range(s, n, 1);
        // variable after variable, like fill_BC_all: the hse energy (variable 1)
        // must see the momenta's x ghosts of the previous fill.  (0,1) and (2,3)
        // can still share launches: energy only reads row jlo/jhi of the others.
        PYRO_TRY(fill_bc_range(s, 0, 2));
        return fill_bc_range(s, 2, 2);
    }
    pyrohip_ctx *c = s->ctx;
    const Geom &g = s->g;
    int n0 = (n < 0) ? 0 : n, cnt = (n < 0) ? s->nvar : 1;
    {
        dim3 grid((g.qy + 255) / 256, 1, cnt), block(256);
        hipLaunchKernelGGL(k_fill_x, grid, block, 0, c->stream, s->d, g, (const int *)s->d_bc, n0);
    }
    {
        dim3 block(16, 16);
        dim3 grid((g.qx + 15) / 16, 1, cnt);
        hipLaunchKernelGGL(k_fill_y, grid, block, 0, c->stream, s->d, g, (const int *)s->d_bc, n0,
                           (const double *)s->d_cval);
    }
    PYRO_CHECK_HIP(hipGetLastError());
    if (n < 0) {
        // every ghost cell is now the image its side's rule gives (index maps only)
        bool rules = !s->nb_set;
        for (size_t k = 0; k < s->bc.size(); k++)
            rules = rules && (s->bc[k] == PYROHIP_BC_OUTFLOW || s->bc[k] == PYROHIP_BC_REFLECT_EVEN ||
                              s->bc[k] == PYROHIP_BC_REFLECT_ODD || s->bc[k] == PYROHIP_BC_PERIODIC);
        s->ghost_by_rules = rules;
    }
    return 0;
}

int pyrohip_state_set_geometry(pyrohip_state *s, const pyrohip_geom *hg)
{
    PYRO_REQUIRE(s, "NULL state");
    pyrohip_ctx *c = s->ctx;
    PYRO_CHECK_HIP(hipSetDevice(c->device));
    PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));
    if (s->sph) {
        if (s->sph->base) PYRO_CHECK_HIP(hipFree(s->sph->base));
        delete s->sph;
        s->sph = nullptr;
    }
    s->next_cfl_min = -1.0;
    if (!hg) return 0;
    const double *src[8] = {hg->Lx, hg->Ly, hg->Ax, hg->Ay, hg->V, hg->dlogAx, hg->dlogAy, hg->x2d};
    for (int k = 0; k < 8; k++) PYRO_REQUIRE(src[k], "NULL geometry array");
    PYRO_REQUIRE(hg->sint && hg->sinb && hg->sinc, "NULL geometry array");
    const Geom &g = s->g;
    const size_t qyp = ((size_t)g.qy + 7) & ~(size_t)7, qxp = ((size_t)g.qx + 7) & ~(size_t)7;
    SphGeom *G = new SphGeom();
    const bool fac = hg->rowf && hg->colf;
    // (+ reciprocals the contracted builds multiply by: rows 1 / Ly, 1 / (F G); columns 1 / |E|, 1 / T)
    const size_t n = 8 * g.plane + 3 * qyp + 16 + (fac ? (size_t)pyro::kSphRowStride * qxp + 6 * qyp : 0);
    PYRO_CHECK_HIP(hipMalloc((void **)&G->base, n * sizeof(double)));
    PYRO_CHECK_HIP(hipMemsetAsync(G->base, 0, n * sizeof(double), c->stream));
    double *planes = G->base + geom_lead(g);
    for (int k = 0; k < 8; k++)
        PYRO_CHECK_HIP(hipMemcpy2DAsync(planes + (size_t)k * g.plane, g.pitch * sizeof(double), src[k],
                                        g.qy * sizeof(double), g.qy * sizeof(double), g.qx,
                                        hipMemcpyHostToDevice, c->stream));
    double *sines = G->base + 8 * g.plane;
    const double *ssrc[3] = {hg->sint, hg->sinb, hg->sinc};
    for (int k = 0; k < 3; k++)
        PYRO_CHECK_HIP(hipMemcpyAsync(sines + k * qyp, ssrc[k], g.qy * sizeof(double),
                                      hipMemcpyHostToDevice, c->stream));
    if (fac) {
        constexpr size_t RS = pyro::kSphRowStride;
        double *rf = G->base + 8 * g.plane + 3 * qyp + 16, *cf = rf + RS * qxp;
        std::vector<double> hr(RS * qxp, 0.0), hc(6 * qyp, 0.0);
        for (int k = 0; k < 7; k++)
            for (int i = 0; i < g.qx; i++) hr[(size_t)i * RS + k] = hg->rowf[(size_t)k * g.qx + i];
        for (int k = 0; k < 4; k++)
            for (int j = 0; j < g.qy; j++) hc[k * qyp + j] = hg->colf[(size_t)k * g.qy + j];
        for (int i = 0; i < g.qx; i++) {
            hr[(size_t)i * RS + 7] = 1.0 / hr[(size_t)i * RS + 4];                                   // 1 / Ly
            hr[(size_t)i * RS + 8] = 1.0 / (hr[(size_t)i * RS + 2] * hr[(size_t)i * RS + 3]);       // 1 / (F G)
        }
        for (int j = 0; j < g.qy; j++) {
            hc[4 * qyp + j] = 1.0 / fabs(hc[2 * qyp + j]);                      // 1 / |E|
            hc[5 * qyp + j] = 1.0 / hc[3 * qyp + j];                            // 1 / T
        }
        PYRO_CHECK_HIP(hipMemcpyAsync(rf, hr.data(), hr.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
        PYRO_CHECK_HIP(hipMemcpyAsync(cf, hc.data(), hc.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
        PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));      // (hr / hc are locals)
        G->rowf = rf; G->colf = cf; G->qxp = qxp; G->qyp = qyp;
    }
    PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));   // the host arrays are borrowed
    G->Lx = planes; G->Ly = planes + g.plane; G->Ax = planes + 2 * g.plane;
    G->Ay = planes + 3 * g.plane; G->V = planes + 4 * g.plane; G->dlAx = planes + 5 * g.plane;
    G->dlAy = planes + 6 * g.plane; G->x2d = planes + 7 * g.plane;
    G->sint = sines; G->sinb = sines + qyp; G->sinc = sines + 2 * qyp;
    G->xmin = hg->xmin; G->ymin = hg->ymin;
    s->sph = G;
    return 0;
}

int pyrohip_state_set_const_bc(pyrohip_state *s, int n, double value)
{
    PYRO_REQUIRE(s, "NULL state");
    PYRO_REQUIRE(n >= 0 && n < s->nvar, "variable index out of range");
    PYRO_CHECK_HIP(hipSetDevice(s->ctx->device));
    PYRO_CHECK_HIP(hipMemcpyAsync(s->d_cval + n, &value, sizeof(double), hipMemcpyHostToDevice,
                                  s->ctx->stream));
    PYRO_CHECK_HIP(hipStreamSynchronize(s->ctx->stream));   // value is a stack temporary
    return 0;
}

int pyrohip_state_lincomb(pyrohip_state *dst, const pyrohip_state *src, const pyrohip_state *k,
                          const double *coef, int ncoef)
{
    PYRO_REQUIRE(dst && src && k && coef, "NULL argument");
    PYRO_REQUIRE(ncoef >= 0 && ncoef <= 8, "at most 8 increments");
    PYRO_REQUIRE(dst->nvar == src->nvar && k->nvar >= ncoef * src->nvar, "variable counts differ");
    PYRO_REQUIRE(dst->g.plane == src->g.plane && dst->g.plane == k->g.plane &&
                     dst->g.nx == src->g.nx && dst->g.ny == src->g.ny && dst->g.ng == src->g.ng,
                 "geometries differ");
    PYRO_REQUIRE(dst->ctx == src->ctx && dst->ctx == k->ctx, "states live on different contexts");
    LinComb lc;
    lc.n = ncoef;
    for (int s = 0; s < ncoef; s++) lc.c[s] = coef[s];
    const Geom &g = dst->g;
    hipLaunchKernelGGL(k_lincomb, dim3((g.qy + 255) / 256, g.qx, dst->nvar), dim3(256), 0,
                       dst->ctx->stream, dst->d, (const double *)src->d, (const double *)k->d, g,
                       dst->nvar, lc);
    PYRO_CHECK_HIP(hipGetLastError());
    dst->next_cfl_min = -1.0;
    dst->ghost_by_rules = false;
    return 0;
}

int pyrohip_state_minmax(pyrohip_state *s, int n, int buf, double *vmin, double *vmax)
{
    PYRO_REQUIRE(s, "NULL state");
    PYRO_REQUIRE(n >= 0 && n < s->nvar, "variable index out of range");
    PYRO_REQUIRE(buf >= 0 && buf <= s->g.ng, "buf out of range");
    pyrohip_ctx *c = s->ctx;
    const Geom &g = s->g;
    dim3 grid(8, 64), block(256);
    int nb = grid.x * grid.y;
    PYRO_TRY(c->reduce.ensure((2 * nb + 2) * sizeof(double)));
    double *part = (double *)c->reduce.p;
    hipLaunchKernelGGL(k_minmax, grid, block, 0, c->stream,
                       (const double *)(s->d + (size_t)n * g.plane), g.pitch, g.ilo - buf,
                       g.ihi + buf, g.jlo - buf, g.jhi + buf, part);
    hipLaunchKernelGGL(k_minmax_final, dim3(1), dim3(256), 0, c->stream, (const double *)part, nb,
                       part + 2 * nb);
    PYRO_CHECK_HIP(hipGetLastError());
    PYRO_CHECK_HIP(hipMemcpyAsync(c->reduce_host, part + 2 * nb, 2 * sizeof(double),
                                  hipMemcpyDeviceToHost, c->stream));
    PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));
    if (vmin) *vmin = ((double *)c->reduce_host)[0];
    if (vmax) *vmax = ((double *)c->reduce_host)[1];
    return 0;
}

}  // extern "C"
