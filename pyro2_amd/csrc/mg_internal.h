// Access to the finest multigrid level for the solvers that build right-hand
// sides / read solutions on the device (incompressible.hip).  Defined in
// multigrid.hip.
#pragma once
#include "common.h"

namespace pyro {

struct MgFinest {
    pyrohip_ctx *ctx;
    int level;          // index of the finest level
    int n, pitch;       // interior cells per side, row pitch (ng = 1)
    double dx;
    double *v, *f, *r;
};

int mg_finest(pyrohip_mg *m, MgFinest *out);
// device address of row i0 of a level array (var 0 = v, 1 = f, 2 = r), its row
// pitch in doubles and the context; for the row moves of a slab-decomposed
// V-cycle (comm.hip).  A lazily zeroed solution is materialised first.
int mg_rows_ptr(pyrohip_mg *m, int level, int var, int i0, int ni, double **ptr, int *pitch,
                pyrohip_ctx **ctx);
// the caller overwrote v of the finest level completely (ghosts and corners
// included): nothing is stale any more
int mg_solution_written(pyrohip_mg *m);
// make every ghost cell of the finest v current (MG.solve ends with fill_BC)
int mg_solution_ghosts(pyrohip_mg *m);

}  // namespace pyro
