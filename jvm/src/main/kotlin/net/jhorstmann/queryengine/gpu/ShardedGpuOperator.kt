// Row-range sharding across the GPUs of one node: one JVM process per GPU, each with its own qe_ctx and its shard of the
// table pinned to its HBM.  The scan needs no communication; only a plan whose root hands rows to the caller
// (Main.kt:18 `physicalPlan.map { it }`) materialises the result on one rank: qe_gather (RCCL inside libqe_hip.so:
// ncclAllGather of the row counts, then grouped ncclSend / ncclRecv straight to the final offsets, rank order = row order).
// NOT compiled in this repository (no JDK in the build image).
package net.jhorstmann.queryengine.gpu

import net.jhorstmann.queryengine.operator.Operator
import java.lang.foreign.Arena
import java.lang.foreign.MemorySegment
import java.lang.foreign.ValueLayout.ADDRESS
import java.lang.foreign.ValueLayout.JAVA_BYTE

/** rows [begin, end) of rank `rank`: equal contiguous shares, boundaries aligned to 64 rows (bitmap words never straddle) */
fun shardRange(nrows: Long, rank: Int, world: Int): LongRange {
    var per = (nrows + world - 1) / world
    per = (per + 63) / 64 * 64
    val begin = minOf(nrows, rank * per)
    return begin until minOf(nrows, begin + per)
}

/** rank 0 creates the id and publishes it through `broadcast` (the host application's own RPC), every rank joins */
fun initComm(ctx: MemorySegment, nranks: Int, rank: Int, broadcast: (ByteArray?) -> ByteArray) = Arena.ofConfined().use { a ->
    val id = a.allocate(128)
    if (rank == 0) QeNative.check(ctx, QeNative.qe_comm_unique_id.invokeExact(ctx, id) as Int)
    val bytes = broadcast(if (rank == 0) id.toArray(JAVA_BYTE) else null)
    MemorySegment.copy(bytes, 0, id, JAVA_BYTE, 0, 128)
    QeNative.check(ctx, QeNative.qe_comm_init.invokeExact(ctx, nranks, rank, id) as Int)
}

/** wraps a shard-local GPU operator; on `root` it yields the rows of ALL shards in input order, elsewhere nothing */
class GatheringOperator(private val ctx: MemorySegment, private val local: GpuFilterProjectOperator, private val root: Int = 0) : Operator() {
    private val arena = Arena.ofConfined()
    private var gathered: MemorySegment = MemorySegment.NULL      // qe_result* on root, NULL elsewhere
    private var columns: Array<HostColumn>? = null
    private var idx = 0L
    private var count = 0L

    override fun open() {
        local.open()                                             // qe_filter_project on this rank's shard
        val out = arena.allocate(ADDRESS)
        QeNative.check(ctx, QeNative.qe_gather.invokeExact(ctx, local.result(), root, out) as Int)   // collective
        gathered = out.get(ADDRESS, 0)
        local.close()
        count = if (gathered == MemorySegment.NULL) 0L else QeNative.qe_result_count.invokeExact(gathered) as Long
        columns = null
        idx = 0
    }

    override fun next(): Array<Any?>? {                           // rows boxed lazily from the gathered columns
        if (idx >= count) return null
        val n = QeNative.qe_result_ncols.invokeExact(gathered) as Int
        val cols = columns ?: Array(n) { HostColumn.fetch(ctx, gathered, it, arena) }.also { columns = it }
        val i = idx++
        return Array(n) { cols[it].box(i) }
    }

    override fun close() {
        if (gathered != MemorySegment.NULL) QeNative.qe_result_free.invokeExact(ctx, gathered)
        gathered = MemorySegment.NULL
        columns = null
    }
}
