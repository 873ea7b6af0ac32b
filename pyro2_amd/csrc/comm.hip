// Multi-GPU plumbing: x-slab halo exchange and scalar all-reduce over RCCL
// (xGMI), one process per GPU.  The reference has no distributed path at all
// (SURVEY.md 5, 8(e)); this is the spatial domain decomposition of the
// explicit update: ng ghost rows per variable are exchanged with the two x
// neighbours once per time step and dt is min-reduced.
//
// Per-link sizing: 4 variables x ng(4) rows x 16400 doubles = 2.1 MB per
// neighbour per direction at 16384^2 -- one grouped send/recv per neighbour,
// not a ring collective, so each transfer rides a single xGMI link.
#include <rccl/rccl.h>

#include "common.h"
#include "mg_internal.h"

namespace pyro {

#define PYRO_CHECK_NCCL(expr)                                                  \
    do {                                                                       \
        ncclResult_t _r = (expr);                                              \
        if (_r != ncclSuccess) {                                               \
            ::pyro::set_error(std::string(#expr) + ": " + ncclGetErrorString(_r)); \
            return PYROHIP_ERR_COMM;                                           \
        }                                                                      \
    } while (0)

}  // namespace pyro

using namespace pyro;

extern "C" {

int pyrohip_comm_unique_id(char *out_id)
{
    PYRO_REQUIRE(out_id, "out_id is NULL");
    static_assert(sizeof(ncclUniqueId) <= PYROHIP_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId id;
    PYRO_CHECK_NCCL(ncclGetUniqueId(&id));
    memset(out_id, 0, PYROHIP_UNIQUE_ID_BYTES);
    memcpy(out_id, &id, sizeof(id));
    return 0;
}

int pyrohip_comm_init(pyrohip_ctx *c, int nranks, int rank, const char *unique_id)
{
    PYRO_REQUIRE(c && unique_id, "NULL argument");
    PYRO_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank / nranks");
    PYRO_REQUIRE(c->comm == nullptr, "communicator already initialised");
    PYRO_CHECK_HIP(hipSetDevice(c->device));
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t comm;
    PYRO_CHECK_NCCL(ncclCommInitRank(&comm, nranks, id, rank));
    c->comm = (void *)comm;
    c->nranks = nranks;
    c->rank = rank;
    // second communicator + stream for the overlapped halo exchange; without
    // them (old RCCL, split refused) the exchange stays on the main stream
    // (the halo stream at the highest priority: the boundary strips of a step and the exchange
    // behind them are dispatched ahead of the interior strips they run beside)
    ncclComm_t comm2 = nullptr;
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    int prio_mode = 2;   // 2: a stream of normal priority (hipStreamCreateWithFlags); 1 / 0: highest / lowest (developer experiments)
    if (const char *e = getenv("PYRO_HALO_PRIO")) prio_mode = atoi(e);
    if (prio_mode == 0) prio_greatest = prio_least;
    if (ncclCommSplit(comm, 0, rank, &comm2, nullptr) == ncclSuccess && comm2 != nullptr &&
        (prio_mode == 2 ? hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking)
                        : hipStreamCreateWithPriority(&c->comm_stream, hipStreamNonBlocking, prio_greatest)) == hipSuccess &&
        hipEventCreateWithFlags(&c->ev_boundary, hipEventDisableTiming) == hipSuccess &&
        hipEventCreateWithFlags(&c->ev_bdone, hipEventDisableTiming) == hipSuccess &&
        hipEventCreateWithFlags(&c->ev_halo, hipEventDisableTiming) == hipSuccess) {
        c->comm_halo = (void *)comm2;
    } else {
        if (comm2) ncclCommDestroy(comm2);
        c->comm_halo = nullptr;
    }
    // every rank must take the same path through the exchange protocol: the halo
    // communicator is used only if ALL ranks have it
    double have = c->comm_halo ? 1.0 : 0.0;
    PYRO_TRY(pyrohip_allreduce_min(c, &have));
    if (have < 1.0 && c->comm_halo) {
        ncclCommDestroy((ncclComm_t)c->comm_halo);
        c->comm_halo = nullptr;
    }
    return 0;
}

int pyrohip_comm_size(pyrohip_ctx *c, int *nranks)
{
    PYRO_REQUIRE(c && nranks, "NULL argument");
    *nranks = 0;
    if (c->comm) PYRO_CHECK_NCCL(ncclCommCount((ncclComm_t)c->comm, nranks));
    return 0;
}

int pyrohip_comm_destroy(pyrohip_ctx *c)
{
    if (!c || !c->comm) return 0;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
    if (c->comm_halo) ncclCommDestroy((ncclComm_t)c->comm_halo);
    c->comm_halo = nullptr;
    if (c->comm_stream) { (void)hipStreamDestroy(c->comm_stream); c->comm_stream = nullptr; }
    if (c->ev_boundary) { (void)hipEventDestroy(c->ev_boundary); c->ev_boundary = nullptr; }
    if (c->ev_halo) { (void)hipEventDestroy(c->ev_halo); c->ev_halo = nullptr; }
    if (c->ev_bdone) { (void)hipEventDestroy(c->ev_bdone); c->ev_bdone = nullptr; }
    ncclCommDestroy((ncclComm_t)c->comm);
    c->comm = nullptr;
    return 0;
}

// one grouped send / recv pair per neighbour and variable for the planes at d
static int post_halo(pyrohip_state *s, double *d, int rank_lo, int rank_hi, ncclComm_t comm,
                     hipStream_t stream)
{
    const Geom &g = s->g;
    const size_t cnt = (size_t)g.ng * g.pitch;   // ng whole rows, contiguous
    PYRO_CHECK_NCCL(ncclGroupStart());
    // order (per variable): send low rows -> lo, recv hi ghosts <- hi,
    // send high rows -> hi, recv lo ghosts <- lo.  With lo == hi (two ranks,
    // periodic) the per-peer FIFO matching pairs my hi ghosts with the peer's
    // low rows and my lo ghosts with its high rows, as required.
    for (int n = 0; n < s->nvar; n++) {
        double *a = d + (size_t)n * g.plane;
        if (rank_lo >= 0)
            PYRO_CHECK_NCCL(ncclSend(a + (size_t)g.ilo * g.pitch, cnt, ncclDouble, rank_lo, comm,
                                     stream));
        if (rank_hi >= 0)
            PYRO_CHECK_NCCL(ncclRecv(a + (size_t)(g.ihi + 1) * g.pitch, cnt, ncclDouble, rank_hi,
                                     comm, stream));
        if (rank_hi >= 0)
            PYRO_CHECK_NCCL(ncclSend(a + (size_t)(g.ihi - g.ng + 1) * g.pitch, cnt, ncclDouble,
                                     rank_hi, comm, stream));
        if (rank_lo >= 0)
            PYRO_CHECK_NCCL(ncclRecv(a, cnt, ncclDouble, rank_lo, comm, stream));
    }
    PYRO_CHECK_NCCL(ncclGroupEnd());
    return 0;
}

int pyrohip_halo_exchange(pyrohip_state *s, int rank_lo, int rank_hi)
{
    PYRO_REQUIRE(s, "NULL state");
    pyrohip_ctx *c = s->ctx;
    PYRO_REQUIRE(c->comm != nullptr, "communicator not initialised");
    PYRO_REQUIRE(rank_lo >= -1 && rank_lo < c->nranks && rank_hi >= -1 && rank_hi < c->nranks,
                 "neighbour rank out of range");
    if (rank_lo < 0 && rank_hi < 0) return 0;
    PYRO_REQUIRE(s->g.nx >= s->g.ng, "slab thinner than the ghost width");
    if (s->halo_pending) {
        // the step that produced this state posted the exchange already (beside
        // its interior strips): the main stream only has to wait for it.
        s->halo_pending = false;
        {   // (profiling: how long the main stream stands still for the posted exchange)
            ProfScope ps(c, "comm:halo_wait");
            PYRO_CHECK_HIP(hipStreamWaitEvent(c->stream, c->ev_halo, 0));
        }
        // Any write to the state since then dropped the cached CFL minimum: the rows the
        // neighbours hold are stale.  Exchanging again would need the neighbours to take
        // part, and they cannot know (this is per-rank state): refuse instead of hanging.
        PYRO_REQUIRE(s->next_cfl_min > 0.0 && rank_lo == s->nb_lo && rank_hi == s->nb_hi,
                     "the state was written (or other neighbours named) while the halo exchange "
                     "its last step posted was pending: call pyrohip_state_set_neighbours(s, -1, -1) "
                     "before modifying a slab between steps");
        return 0;
    }
    ProfScope ps(c, "comm:halo_sync");       // exchange on the main stream, nothing beside it
    return post_halo(s, s->d, rank_lo, rank_hi, (ncclComm_t)c->comm, c->stream);
}

int pyrohip_state_halo_pending(pyrohip_state *s, int *flag)
{
    PYRO_REQUIRE(s && flag, "NULL argument");
    *flag = (s->halo_pending && s->next_cfl_min > 0.0) ? 1 : 0;
    return 0;
}

int pyrohip_state_set_neighbours(pyrohip_state *s, int rank_lo, int rank_hi)
{
    PYRO_REQUIRE(s, "NULL state");
    // (no upper bound: without a communicator -- halos staged through the host --
    // the neighbours only select the launch order, see comp_step_wave)
    PYRO_REQUIRE(rank_lo >= -1 && rank_hi >= -1, "neighbour rank out of range");
    if (s->halo_pending && (rank_lo != s->nb_lo || rank_hi != s->nb_hi)) {
        // a posted exchange still belongs to the old neighbours: let it land, then forget it
        PYRO_TRY(comm_wait_halo(s));
        s->halo_pending = false;
    }
    s->nb_lo = rank_lo; s->nb_hi = rank_hi;
    s->nb_set = (rank_lo >= 0 || rank_hi >= 0);
    return 0;
}

// ---- row moves of a slab-decomposed multigrid V-cycle (multigrid/slab.py) ----
// Whole rows of a level array are contiguous (pitch doubles each), so a block of
// rows is one message.
int pyrohip_mg_exchange_rows(pyrohip_mg *m, int level, int var, int row0, int row1, int h,
                             int rank_lo, int rank_hi)
{
    double *p = nullptr;
    int pitch = 0;
    pyrohip_ctx *c = nullptr;
    PYRO_REQUIRE(h >= 1 && row1 - row0 + 1 >= h, "halo deeper than the slab");
    PYRO_TRY(mg_rows_ptr(m, level, var, row0 - (rank_lo >= 0 ? h : 0),
                         (row1 - row0 + 1) + (rank_lo >= 0 ? h : 0) + (rank_hi >= 0 ? h : 0), &p,
                         &pitch, &c));
    PYRO_REQUIRE(c->comm != nullptr, "communicator not initialised");
    double *slab = p + (size_t)(rank_lo >= 0 ? h : 0) * pitch;      // row0
    const size_t cnt = (size_t)h * pitch;
    const size_t nrows = (size_t)(row1 - row0 + 1);
    ncclComm_t comm = (ncclComm_t)c->comm;
    PYRO_CHECK_NCCL(ncclGroupStart());
    if (rank_lo >= 0) {
        PYRO_CHECK_NCCL(ncclSend(slab, cnt, ncclDouble, rank_lo, comm, c->stream));
        PYRO_CHECK_NCCL(ncclRecv(slab - cnt, cnt, ncclDouble, rank_lo, comm, c->stream));
    }
    if (rank_hi >= 0) {
        PYRO_CHECK_NCCL(ncclSend(slab + (nrows - h) * pitch, cnt, ncclDouble, rank_hi, comm, c->stream));
        PYRO_CHECK_NCCL(ncclRecv(slab + nrows * pitch, cnt, ncclDouble, rank_hi, comm, c->stream));
    }
    PYRO_CHECK_NCCL(ncclGroupEnd());
    return 0;
}

// rows [i0, i0 + ni) of a level array to / from one peer (gather to and scatter from
// the rank that owns the collapsed levels); calls of one collective step are grouped
// by the caller with pyrohip_comm_group(1) ... pyrohip_comm_group(0)
int pyrohip_mg_send_rows(pyrohip_mg *m, int level, int var, int i0, int ni, int peer)
{
    double *p = nullptr;
    int pitch = 0;
    pyrohip_ctx *c = nullptr;
    PYRO_TRY(mg_rows_ptr(m, level, var, i0, ni, &p, &pitch, &c));
    PYRO_REQUIRE(c->comm != nullptr && peer >= 0 && peer < c->nranks, "bad peer / no communicator");
    PYRO_CHECK_NCCL(ncclSend(p, (size_t)ni * pitch, ncclDouble, peer, (ncclComm_t)c->comm, c->stream));
    return 0;
}

int pyrohip_mg_recv_rows(pyrohip_mg *m, int level, int var, int i0, int ni, int peer)
{
    double *p = nullptr;
    int pitch = 0;
    pyrohip_ctx *c = nullptr;
    PYRO_TRY(mg_rows_ptr(m, level, var, i0, ni, &p, &pitch, &c));
    PYRO_REQUIRE(c->comm != nullptr && peer >= 0 && peer < c->nranks, "bad peer / no communicator");
    PYRO_CHECK_NCCL(ncclRecv(p, (size_t)ni * pitch, ncclDouble, peer, (ncclComm_t)c->comm, c->stream));
    return 0;
}

// whole rows of every variable of a state to / from one peer (gather of the slabs for output)
static int state_rows(pyrohip_state *s, int i0, int ni, int peer, bool send)
{
    PYRO_REQUIRE(s, "NULL state");
    pyrohip_ctx *c = s->ctx;
    const Geom &g = s->g;
    PYRO_REQUIRE(c->comm != nullptr && peer >= 0 && peer < c->nranks, "bad peer / no communicator");
    PYRO_REQUIRE(i0 >= 0 && ni >= 1 && i0 + ni <= g.qx, "rows out of range");
    PYRO_TRY(comm_wait_halo(s));
    const size_t cnt = (size_t)ni * g.pitch;
    for (int n = 0; n < s->nvar; n++) {
        double *a = s->d + (size_t)n * g.plane + (size_t)i0 * g.pitch;
        if (send)
            PYRO_CHECK_NCCL(ncclSend(a, cnt, ncclDouble, peer, (ncclComm_t)c->comm, c->stream));
        else
            PYRO_CHECK_NCCL(ncclRecv(a, cnt, ncclDouble, peer, (ncclComm_t)c->comm, c->stream));
    }
    return 0;
}

int pyrohip_state_send_rows(pyrohip_state *s, int i0, int ni, int peer)
{
    return state_rows(s, i0, ni, peer, true);
}

int pyrohip_state_recv_rows(pyrohip_state *s, int i0, int ni, int peer)
{
    PYRO_TRY(state_rows(s, i0, ni, peer, false));
    s->next_cfl_min = -1.0;      // the data changed: no cached CFL minimum
    s->ghost_by_rules = false;
    return 0;
}

int pyrohip_comm_group(int begin)
{
    if (begin) PYRO_CHECK_NCCL(ncclGroupStart());
    else PYRO_CHECK_NCCL(ncclGroupEnd());
    return 0;
}

int pyrohip_comm_set_global_dt(pyrohip_ctx *c, int on)
{
    PYRO_REQUIRE(c, "NULL context");
    PYRO_REQUIRE(!on || c->comm != nullptr, "communicator not initialised");
    c->global_cfl = on != 0;
    return 0;
}

static int allreduce_scalar(pyrohip_ctx *c, double *value, ncclRedOp_t op)
{
    PYRO_REQUIRE(c && value, "NULL argument");
    if (c->comm == nullptr) return 0;   // single process, no communicator
    PYRO_TRY(c->reduce.ensure(64));
    double *d = (double *)c->reduce.p;
    *(double *)c->reduce_host = *value;
    PYRO_CHECK_HIP(hipMemcpyAsync(d, c->reduce_host, sizeof(double), hipMemcpyHostToDevice,
                                  c->stream));
    PYRO_CHECK_NCCL(ncclAllReduce(d, d + 1, 1, ncclDouble, op, (ncclComm_t)c->comm, c->stream));
    PYRO_CHECK_HIP(hipMemcpyAsync(c->reduce_host, d + 1, sizeof(double), hipMemcpyDeviceToHost,
                                  c->stream));
    PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));
    *value = *(double *)c->reduce_host;
    return 0;
}

int pyrohip_allreduce_min(pyrohip_ctx *c, double *value)
{
    return allreduce_scalar(c, value, ncclMin);
}

int pyrohip_allreduce_max(pyrohip_ctx *c, double *value)
{
    return allreduce_scalar(c, value, ncclMax);
}

int pyrohip_allreduce_sum(pyrohip_ctx *c, double *values, int n)
{
    PYRO_REQUIRE(c && values && n >= 1 && n <= 16, "NULL argument / 1..16 values");
    if (c->comm == nullptr) return 0;   // single process, no communicator
    PYRO_TRY(c->reduce.ensure(512));
    double *d = (double *)c->reduce.p;
    memcpy(c->reduce_host, values, n * sizeof(double));
    PYRO_CHECK_HIP(hipMemcpyAsync(d, c->reduce_host, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    PYRO_CHECK_NCCL(ncclAllReduce(d, d + 16, n, ncclDouble, ncclSum, (ncclComm_t)c->comm, c->stream));
    PYRO_CHECK_HIP(hipMemcpyAsync(c->reduce_host, d + 16, n * sizeof(double), hipMemcpyDeviceToHost,
                                  c->stream));
    PYRO_CHECK_HIP(hipStreamSynchronize(c->stream));
    memcpy(values, c->reduce_host, n * sizeof(double));
    return 0;
}

}  // extern "C"

namespace pyro {
bool comm_can_overlap(const pyrohip_state *s)
{
    const pyrohip_ctx *c = s->ctx;
    return c->comm_halo != nullptr && c->comm_stream != nullptr;
}

int comm_post_halo(pyrohip_state *s, double *d)
{
    pyrohip_ctx *c = s->ctx;
    PYRO_CHECK_HIP(hipEventRecord(c->ev_boundary, c->stream));
    PYRO_CHECK_HIP(hipStreamWaitEvent(c->comm_stream, c->ev_boundary, 0));
    PYRO_TRY(post_halo(s, d, s->nb_lo, s->nb_hi, (ncclComm_t)c->comm_halo, c->comm_stream));
    PYRO_CHECK_HIP(hipEventRecord(c->ev_halo, c->comm_stream));
    return 0;
}

int comm_fork_boundary(pyrohip_state *s, hipStream_t *bs)
{
    pyrohip_ctx *c = s->ctx;
    PYRO_CHECK_HIP(hipEventRecord(c->ev_boundary, c->stream));
    PYRO_CHECK_HIP(hipStreamWaitEvent(c->comm_stream, c->ev_boundary, 0));
    *bs = c->comm_stream;
    return 0;
}

int comm_post_halo_here(pyrohip_state *s, double *d)
{
    pyrohip_ctx *c = s->ctx;
    PYRO_CHECK_HIP(hipEventRecord(c->ev_bdone, c->comm_stream));
    PYRO_TRY(post_halo(s, d, s->nb_lo, s->nb_hi, (ncclComm_t)c->comm_halo, c->comm_stream));
    PYRO_CHECK_HIP(hipEventRecord(c->ev_halo, c->comm_stream));
    return 0;
}

int comm_join_boundary(pyrohip_state *s)
{
    pyrohip_ctx *c = s->ctx;
    PYRO_CHECK_HIP(hipStreamWaitEvent(c->stream, c->ev_bdone, 0));
    return 0;
}

int comm_wait_halo(pyrohip_state *s)
{
    pyrohip_ctx *c = s->ctx;
    if (s->halo_pending && c->ev_halo) {
        ProfScope ps(c, "comm:halo_wait");
        PYRO_CHECK_HIP(hipStreamWaitEvent(c->stream, c->ev_halo, 0));
    }
    return 0;
}

int comm_allreduce_min_device(pyrohip_ctx *c, double *d)
{
    if (c->comm == nullptr) return 0;
    ProfScope ps(c, "comm:allreduce_dt");
    PYRO_CHECK_NCCL(ncclAllReduce(d, d, 1, ncclDouble, ncclMin, (ncclComm_t)c->comm, c->stream));
    return 0;
}
}  // namespace pyro
