// Reference-side binding of libqe_hip.so (include/qe_hip.h) through the Java Foreign Function &
// Memory API (JDK >= 22).  NOT compiled in this repository: the build image has no JDK/kotlinc
// (SURVEY.md 8c).  It shows exactly which symbols the Kotlin side binds and how; a JNI shim with
// the same calls is the alternative for the reference's Java 11 target (pom.xml:13-15).
package net.jhorstmann.queryengine.gpu

import java.lang.foreign.*
import java.lang.foreign.ValueLayout.*
import java.lang.invoke.MethodHandle

internal object QeNative {
    private val linker = Linker.nativeLinker()
    private val lib = SymbolLookup.libraryLookup(System.getProperty("qe.hip.library", "libqe_hip.so"), Arena.global())

    private fun handle(name: String, res: MemoryLayout?, vararg args: MemoryLayout): MethodHandle {
        val fd = if (res == null) FunctionDescriptor.ofVoid(*args) else FunctionDescriptor.of(res, *args)
        return linker.downcallHandle(lib.find(name).orElseThrow { UnsatisfiedLinkError(name) }, fd)
    }

    val qe_last_error = handle("qe_last_error", ADDRESS, ADDRESS)
    val qe_ctx_create = handle("qe_ctx_create", JAVA_INT, JAVA_INT, ADDRESS, ADDRESS)
    val qe_ctx_destroy = handle("qe_ctx_destroy", null, ADDRESS)
    val qe_dict_create = handle("qe_dict_create", JAVA_INT, ADDRESS, JAVA_INT, ADDRESS, ADDRESS)
    val qe_dict_size = handle("qe_dict_size", JAVA_INT, ADDRESS)
    val qe_dict_entry = handle("qe_dict_entry", ADDRESS, ADDRESS, JAVA_INT)
    val qe_batch_create = handle("qe_batch_create", JAVA_INT, ADDRESS, JAVA_LONG, JAVA_INT, ADDRESS, ADDRESS)
    val qe_batch_free = handle("qe_batch_free", null, ADDRESS, ADDRESS)
    val qe_expr_compile = handle("qe_expr_compile", JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, ADDRESS)
    val qe_expr_free = handle("qe_expr_free", null, ADDRESS, ADDRESS)
    val qe_filter_project = handle("qe_filter_project", JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, JAVA_INT, ADDRESS)
    val qe_filter_project_prepare = handle("qe_filter_project_prepare", JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, JAVA_INT)
    val qe_filter_aggregate = handle("qe_filter_aggregate", JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, JAVA_INT,
            ADDRESS, ADDRESS, ADDRESS)
    // ctx, batch, filter|NULL, qe_expr*[nkeys], nkeys, qe_expr*[nagg], int32 fns[nagg], nagg, qe_result** -> status
    val qe_filter_groupby = handle("qe_filter_groupby", JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, JAVA_INT, ADDRESS, ADDRESS,
            JAVA_INT, ADDRESS)
    val qe_result_count = handle("qe_result_count", JAVA_LONG, ADDRESS)
    val qe_result_ncols = handle("qe_result_ncols", JAVA_INT, ADDRESS)
    val qe_result_column = handle("qe_result_column", JAVA_INT, ADDRESS, JAVA_INT, ADDRESS)
    val qe_result_column_to_host = handle("qe_result_column_to_host", JAVA_INT, ADDRESS, ADDRESS, JAVA_INT, ADDRESS, ADDRESS)
    val qe_result_free = handle("qe_result_free", null, ADDRESS, ADDRESS)
    // result -> PINNED host memory owned by the library (pooled per context), on the context's copy stream:
    // qe_result_to_host STARTS the copies and returns; wait blocks; column gives HOST pointers in a qe_col_view
    val qe_result_to_host = handle("qe_result_to_host", JAVA_INT, ADDRESS, ADDRESS, ADDRESS)          // ctx, result, qe_host_result**
    val qe_host_result_wait = handle("qe_host_result_wait", JAVA_INT, ADDRESS, ADDRESS)
    val qe_host_result_count = handle("qe_host_result_count", JAVA_LONG, ADDRESS)
    val qe_host_result_ncols = handle("qe_host_result_ncols", JAVA_INT, ADDRESS)
    val qe_host_result_column = handle("qe_host_result_column", JAVA_INT, ADDRESS, JAVA_INT, ADDRESS) // host, col, qe_col_view*
    val qe_host_result_free = handle("qe_host_result_free", null, ADDRESS, ADDRESS)
    // ctx, qe_result*[nparts], nparts, qe_result** -> status: results of consecutive batches as ONE result (input order)
    val qe_result_concat = handle("qe_result_concat", JAVA_INT, ADDRESS, ADDRESS, JAVA_INT, ADDRESS)
    // ctx, result, column (0-based), qe_result** -> status: OrderByOperator.kt:9-15 on the device
    val qe_result_order_by = handle("qe_result_order_by", JAVA_INT, ADDRESS, ADDRESS, JAVA_INT, ADDRESS)

    // ---- the exchange step of a row-range sharded scan: one JVM process (one qe_ctx) per GPU ----
    // rank 0: qe_comm_unique_id(ctx, id128) -> the 128 bytes travel to the other ranks over the host's own channel
    val qe_comm_unique_id = handle("qe_comm_unique_id", JAVA_INT, ADDRESS, ADDRESS)
    val qe_comm_init = handle("qe_comm_init", JAVA_INT, ADDRESS, JAVA_INT, JAVA_INT, ADDRESS)       // ctx, nranks, rank, id128
    val qe_comm_rank = handle("qe_comm_rank", JAVA_INT, ADDRESS)
    val qe_comm_nranks = handle("qe_comm_nranks", JAVA_INT, ADDRESS)
    val qe_comm_destroy = handle("qe_comm_destroy", null, ADDRESS)
    // collective: ctx, local result, root, qe_result** (non-NULL on root only) -> status (QE_ERR_COMM = 7 on an RCCL failure)
    val qe_gather = handle("qe_gather", JAVA_INT, ADDRESS, ADDRESS, JAVA_INT, ADDRESS)
    // collective: ctx, batch, filter|NULL, qe_expr*[nproj], nproj, root, nslices (0 = 8), qe_result** -> scan + exchange overlapped
    val qe_filter_project_gather = handle("qe_filter_project_gather", JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, JAVA_INT, JAVA_INT, JAVA_INT, ADDRESS)
    // collective: ctx, send, nbytes, recv[nranks * nbytes] (host buffers): aggregate partials, counts
    val qe_comm_allgather_host = handle("qe_comm_allgather_host", JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, ADDRESS)

    // ---- CSV text -> columns (replaces CsvTable / UnivocityCsvTable scan leaves) ----
    // ctx, path, nfields, char*[names], int32[types], qe_csv_table** -> status
    val qe_csv_parse_file = handle("qe_csv_parse_file", JAVA_INT, ADDRESS, ADDRESS, JAVA_INT, ADDRESS, ADDRESS, ADDRESS)
    val qe_csv_nrows = handle("qe_csv_nrows", JAVA_LONG, ADDRESS)
    val qe_csv_column = handle("qe_csv_column", JAVA_INT, ADDRESS, JAVA_INT, ADDRESS)                  // -> qe_col_desc (host pointers)
    val qe_csv_pin = handle("qe_csv_pin", JAVA_INT, ADDRESS, ADDRESS, ADDRESS)                          // -> qe_batch (HBM)
    val qe_csv_free = handle("qe_csv_free", null, ADDRESS, ADDRESS)

    /** struct qe_col_desc { int32 type; int32 reserved; const void* data; const uint64* validity; const qe_dict* dict; } */
    val COL_DESC: StructLayout = MemoryLayout.structLayout(
            JAVA_INT.withName("type"), JAVA_INT.withName("reserved"),
            ADDRESS.withName("data"), ADDRESS.withName("validity"), ADDRESS.withName("dict"))

    /** struct qe_col_view { int32 type; int32 nullable; const void* data; const uint64* validity; int64 count; const qe_dict* dict; } */
    val COL_VIEW: StructLayout = MemoryLayout.structLayout(
            JAVA_INT.withName("type"), JAVA_INT.withName("nullable"),
            ADDRESS.withName("data"), ADDRESS.withName("validity"), JAVA_LONG.withName("count"), ADDRESS.withName("dict"))

    fun check(ctx: MemorySegment, status: Int) {
        if (status != 0) {
            val msg = (qe_last_error.invokeExact(ctx) as MemorySegment).reinterpret(4096).getString(0)
            throw when (status) {
                1 -> IllegalArgumentException(msg)
                2 -> net.jhorstmann.queryengine.evaluator.TypeCheckException(msg)
                else -> IllegalStateException("libqe_hip status $status: $msg")
            }
        }
    }
}
