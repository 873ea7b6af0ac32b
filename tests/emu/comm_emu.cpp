// host-emu build has no RCCL: the comm entry points report "unsupported".
// (CPU multi-process tests exchange halos over torch.distributed/gloo through
// download_rows / upload_rows instead -- tests/test_decomp_gloo.py.)
#include "../../pyro2_amd/csrc/common.h"
extern "C" {
int pyrohip_comm_unique_id(char *) { pyro::set_error("host-emu: no RCCL"); return PYROHIP_ERR_UNSUPPORTED; }
int pyrohip_comm_init(pyrohip_ctx *, int, int, const char *) { pyro::set_error("host-emu: no RCCL"); return PYROHIP_ERR_UNSUPPORTED; }
int pyrohip_comm_destroy(pyrohip_ctx *) { return 0; }
int pyrohip_comm_size(pyrohip_ctx *, int *n) { if (n) *n = 0; return 0; }
// Test hooks: a communicator made of CALLBACKS (tests/test_decomp_gloo.py: four gloo processes
// drive pyrohip_comp_evolve on their slabs; the halo rows and the CFL minimum travel through
// torch.distributed in the callbacks).  The emulator runs everything synchronously, so the
// exchange a real step POSTS for the new state (comm_post_halo) is carried out when the next
// step asks for it -- the same rows, the same order of events on every rank.
typedef int (*emu_halo_fn)(pyrohip_state *, int, int);
typedef double (*emu_min_fn)(double);
static emu_halo_fn g_halo_fn = nullptr;
static emu_min_fn g_min_fn = nullptr;
int pyrohip_emu_set_comm(emu_halo_fn h, emu_min_fn m) { g_halo_fn = h; g_min_fn = m; return 0; }
int pyrohip_halo_exchange(pyrohip_state *s, int lo, int hi)
{
    if (!g_halo_fn) { pyro::set_error("host-emu: no RCCL"); return PYROHIP_ERR_UNSUPPORTED; }
    s->halo_pending = false;
    return g_halo_fn(s, lo, hi);
}
int pyrohip_allreduce_min(pyrohip_ctx *, double *) { return 0; }
int pyrohip_allreduce_max(pyrohip_ctx *, double *) { return 0; }
int pyrohip_allreduce_sum(pyrohip_ctx *, double *, int) { return 0; }
// Test hook: "the other ranks' CFL minimum".  With it set, the device-side all-reduce of the
// step kernels' minimum (comm_allreduce_min_device) folds this value in, so a CPU test can see
// whether EVERY dt of a device-side run -- the first one of a call included -- comes from the
// global minimum (tests/test_device_compressible.py::test_comp_evolve_global_minimum_every_step;
// the bug this pins was found on hardware in round 4).
static double g_peer_min = -1.0;
int pyrohip_emu_set_peer_min(double v) { g_peer_min = v; return 0; }
int pyrohip_comm_set_global_dt(pyrohip_ctx *c, int on)
{
    if (on && !(g_peer_min > 0.0) && !g_min_fn) { pyro::set_error("host-emu: no RCCL"); return PYROHIP_ERR_UNSUPPORTED; }
    c->global_cfl = on != 0;
    if (g_halo_fn) c->comm = on ? (void *)&g_halo_fn : nullptr;     // "there is a communicator"
    return 0;
}
int pyrohip_mg_exchange_rows(pyrohip_mg *, int, int, int, int, int, int, int) { pyro::set_error("host-emu: no RCCL"); return PYROHIP_ERR_UNSUPPORTED; }
int pyrohip_mg_send_rows(pyrohip_mg *, int, int, int, int, int) { pyro::set_error("host-emu: no RCCL"); return PYROHIP_ERR_UNSUPPORTED; }
int pyrohip_mg_recv_rows(pyrohip_mg *, int, int, int, int, int) { pyro::set_error("host-emu: no RCCL"); return PYROHIP_ERR_UNSUPPORTED; }
int pyrohip_state_send_rows(pyrohip_state *, int, int, int) { pyro::set_error("host-emu: no RCCL"); return PYROHIP_ERR_UNSUPPORTED; }
int pyrohip_state_recv_rows(pyrohip_state *, int, int, int) { pyro::set_error("host-emu: no RCCL"); return PYROHIP_ERR_UNSUPPORTED; }
int pyrohip_comm_group(int) { pyro::set_error("host-emu: no RCCL"); return PYROHIP_ERR_UNSUPPORTED; }
int pyrohip_state_halo_pending(pyrohip_state *s, int *flag) { *flag = 0; (void)s; return 0; }
int pyrohip_state_set_neighbours(pyrohip_state *s, int lo, int hi)
{
    // the emulated backend has no communicator: the neighbours only select the
    // boundary-strips-first launch order of the row-marching kernel (the halos
    // still travel through the host, tests/test_decomp_gloo.py)
    s->nb_lo = lo; s->nb_hi = hi; s->nb_set = (lo >= 0 || hi >= 0);
    return 0;
}
}
namespace pyro {
int comm_allreduce_min_device(pyrohip_ctx *, double *d)
{
    if (g_peer_min > 0.0 && g_peer_min < *d) *d = g_peer_min;    // (device memory is host memory here)
    if (g_min_fn) *d = g_min_fn(*d);
    return 0;
}
bool comm_can_overlap(const pyrohip_state *) { return true; }
int comm_post_halo(pyrohip_state *, double *) { return 0; }     // nothing to post: see above
int comm_fork_boundary(pyrohip_state *s, hipStream_t *bs) { *bs = s->ctx->stream; return 0; }
int comm_post_halo_here(pyrohip_state *, double *) { return 0; }
int comm_join_boundary(pyrohip_state *) { return 0; }
int comm_wait_halo(pyrohip_state *) { return 0; }
}
