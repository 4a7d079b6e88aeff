// qe_pernode.h -- kernel-per-expression-node executor (QE_EXEC_PER_NODE).
#pragma once
#include "qe_internal.h"

namespace qe {
qe_result *run_per_node(qe_ctx *ctx, const qe_batch *batch, const qe_expr *filter, const qe_expr *const *projs,
                        int32_t nproj);
}
