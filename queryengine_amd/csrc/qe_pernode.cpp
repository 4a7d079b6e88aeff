#include "qe_pernode.h"
namespace qe {
qe_result *run_per_node(qe_ctx *, const qe_batch *, const qe_expr *, const qe_expr *const *, int32_t) {
    fail(QE_ERR_UNSUPPORTED, "QE_EXEC_PER_NODE is not built yet");
}
}
