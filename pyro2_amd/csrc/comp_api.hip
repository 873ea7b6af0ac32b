// extern "C" entry points of the compressible solver: argument checking and
// dispatch between the bit-faithful (exact) and contracted (fastm) builds of
// compressible.hip / comp_fused.hip.
#include "common.h"

namespace pyro {
namespace exact {
int comp_dt(pyrohip_state *, const pyrohip_comp_params *, double, double *);
int comp_step_staged(pyrohip_state *, const pyrohip_comp_params *, double);
int comp_step_fused(pyrohip_state *, const pyrohip_comp_params *, double);
int comp_step_wave(pyrohip_state *, const pyrohip_comp_params *, double);
int comp_step_sph(pyrohip_state *, const pyrohip_comp_params *, double);
int comp_dt_sph(pyrohip_state *, const pyrohip_comp_params *, double, double *);
int comp_stage_dump(pyrohip_state *, int, double *);
int comp_sponge(pyrohip_state *, const pyrohip_comp_params *, double);
int comp_rk_dt(pyrohip_state *, const pyrohip_comp_params *, double, double *);
int comp_rk_rhs(pyrohip_state *, const pyrohip_comp_params *, pyrohip_state *, int);
}
namespace fastm {
int comp_rk_rhs(pyrohip_state *, const pyrohip_comp_params *, pyrohip_state *, int);
int comp_dt(pyrohip_state *, const pyrohip_comp_params *, double, double *);
int comp_step_staged(pyrohip_state *, const pyrohip_comp_params *, double);
int comp_step_fused(pyrohip_state *, const pyrohip_comp_params *, double);
int comp_step_wave(pyrohip_state *, const pyrohip_comp_params *, double);
int comp_step_sph(pyrohip_state *, const pyrohip_comp_params *, double);
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

extern "C" {

int pyrohip_comp_dt(pyrohip_state *s, const pyrohip_comp_params *p, double cfl, double *dt_out)
{
    PYRO_TRY(check_comp(s, p));
    PYRO_REQUIRE(dt_out, "dt_out is NULL");
    if (s->sph)
        return p->fast_math ? fastm::comp_dt_sph(s, p, cfl, dt_out)
                            : exact::comp_dt_sph(s, p, cfl, dt_out);
    return p->fast_math ? fastm::comp_dt(s, p, cfl, dt_out) : exact::comp_dt(s, p, cfl, dt_out);
}

int pyrohip_comp_dt_is_global(pyrohip_state *s, int *flag)
{
    PYRO_REQUIRE(s && flag, "NULL argument");
    *flag = (s->next_cfl_min > 0.0 && s->cfl_is_global && !s->user_bc && !s->ramp_bc) ? 1 : 0;
    return 0;
}

int pyrohip_comp_step(pyrohip_state *s, const pyrohip_comp_params *p, double dt)
{
    PYRO_TRY(check_comp(s, p));
    PYRO_REQUIRE(dt > 0.0, "dt must be positive");
    PYRO_REQUIRE(p->kernel_set >= -1 && p->kernel_set <= 2, "kernel_set must be -1 (automatic), 0, 1 or 2");
    PYRO_REQUIRE(p->riemann >= 0 && p->riemann <= 2, "riemann must be 0 (HLLC), 1 (CGF) or 2 (HLLC_lm)");
    int rc;
    if (s->sph) {
        // compressible/simulation.py:206-208: no HLLC on a SphericalPolar grid
        PYRO_REQUIRE(p->riemann == 1, "a SphericalPolar grid needs the CGF Riemann solver");
        PYRO_REQUIRE(!s->user_bc && !s->ramp_bc && !s->heat,
                     "hse / ambient / ramp boundaries and heating are Cartesian-only");
        rc = p->fast_math ? fastm::comp_step_sph(s, p, dt) : exact::comp_step_sph(s, p, dt);
    } else if (p->kernel_set == 2 || (p->kernel_set == -1 && wave_kernel_pays(s->g)))
        rc = p->fast_math ? fastm::comp_step_wave(s, p, dt) : exact::comp_step_wave(s, p, dt);
    else if (p->kernel_set == 1 || p->kernel_set == -1)
        rc = p->fast_math ? fastm::comp_step_fused(s, p, dt) : exact::comp_step_fused(s, p, dt);
    else
        rc = p->fast_math ? fastm::comp_step_staged(s, p, dt) : exact::comp_step_staged(s, p, dt);
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
    return exact::comp_rk_dt(s, p, cfl, dt_out);
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
    return p->fast_math ? fastm::comp_rk_rhs(y, p, k, slot) : exact::comp_rk_rhs(y, p, k, slot);
}

int pyrohip_comp_stage_dump(pyrohip_state *s, int stage_id, double *out)
{
    PYRO_REQUIRE(s && out, "NULL argument");
    return exact::comp_stage_dump(s, stage_id, out);
}

}  // extern "C"
