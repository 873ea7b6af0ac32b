// placeholder until comm.hip (RCCL halo exchange) lands
#include "common.h"
extern "C" {
int pyrohip_comm_unique_id(char *) { pyro::set_error("comm not built"); return PYROHIP_ERR_UNSUPPORTED; }
int pyrohip_comm_init(pyrohip_ctx *, int, int, const char *) { pyro::set_error("comm not built"); return PYROHIP_ERR_UNSUPPORTED; }
int pyrohip_comm_destroy(pyrohip_ctx *) { return 0; }
int pyrohip_halo_exchange(pyrohip_state *, int, int) { pyro::set_error("comm not built"); return PYROHIP_ERR_UNSUPPORTED; }
int pyrohip_allreduce_min(pyrohip_ctx *, double *) { return 0; }
int pyrohip_allreduce_max(pyrohip_ctx *, double *) { return 0; }
}
