// qe_test_transport.cpp -- TEST INFRASTRUCTURE ONLY.  A stand-in for librccl that lets qe_gather / qe_comm_* (the product's
// exchange step, queryengine_amd/csrc/qe_comm.cpp) run with MORE THAN ONE RANK ON ONE GPU: RCCL itself refuses two ranks on
// one device, and a GPU box of this pool has one.  It exports the nine nccl* symbols qe_comm.cpp binds (same signatures as
// rccl.h) and moves the bytes between the rank processes over Unix-domain sockets with host staging:
//
//   send:  stream sync -> hipMemcpy D2H -> socket          recv:  socket -> hipMemcpy H2D
//
// Semantics kept from NCCL: operations between ncclGroupStart / ncclGroupEnd are issued together and may complete in any
// order (a poll() loop progresses every queue, so a grouped exchange never deadlocks on socket buffers); sends and receives
// between one pair of ranks match in program order; an operation outside a group completes before the call returns (NCCL:
// before later work on the stream -- the shim synchronises instead); a peer that never shows up is an error, not a hang
// (QE_TEST_TRANSPORT_TIMEOUT_S, default 60).
//
// Selected with QE_RCCL_LIBRARY=<path of this .so> (qe_comm.cpp honours that variable for exactly this purpose).  Nothing
// under queryengine_amd/ references this file; tests/test_abi.py enforces it.
#include <errno.h>
#include <fcntl.h>
#include <poll.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/un.h>
#include <time.h>
#include <unistd.h>

#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

namespace {

constexpr int kOk = 0, kSystemError = 2, kInternalError = 3, kInvalidArgument = 4;

struct UniqueId { char internal[128]; };

double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

double timeout_s() {
    const char *e = getenv("QE_TEST_TRANSPORT_TIMEOUT_S");
    return e && *e ? atof(e) : 60.0;
}

struct Op {
    bool send;
    int peer;
    void *dev;
    size_t nbytes, done;
    std::vector<char> host;
};

struct Comm {
    int nranks = 0, rank = -1;
    std::string base;
    int listen_fd = -1;
    std::vector<int> fd;          // per peer
};

// group state is per thread in NCCL; one thread per process drives a qe_ctx
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
thread_local Comm *g_comm = nullptr;
thread_local hipStream_t g_stream = nullptr;
thread_local std::string g_err;
thread_local bool g_group_failed = false;   // an operation of the open group was refused: its ncclGroupEnd discards the group

// QE_TEST_TRANSPORT_HOSTMEM=1: buffers are host memory (the CPU self-test of this file, tests/test_transport_cpu.py)
bool hostmem() {
    static const bool on = [] { const char *e = getenv("QE_TEST_TRANSPORT_HOSTMEM"); return e && *e == '1'; }();
    return on;
}
bool copy_bytes(void *dst, const void *src, size_t n, hipMemcpyKind kind) {
    if (hostmem()) { memcpy(dst, src, n); return true; }
    return hipMemcpy(dst, src, n, kind) == hipSuccess;
}
bool sync_stream(hipStream_t s) { return hostmem() || hipStreamSynchronize(s) == hipSuccess; }

int fail(int code, const std::string &msg) {
    g_err = msg;
    if (getenv("QE_TEST_TRANSPORT_VERBOSE")) fprintf(stderr, "[qe_test_transport] %s\n", msg.c_str());
    return code;
}

void set_nonblocking(int fd, bool on) {
    int fl = fcntl(fd, F_GETFL, 0);
    fcntl(fd, F_SETFL, on ? (fl | O_NONBLOCK) : (fl & ~O_NONBLOCK));
}

bool write_all(int fd, const void *p, size_t n) {
    const char *c = (const char *)p;
    while (n) {
        ssize_t w = ::send(fd, c, n, MSG_NOSIGNAL);
        if (w < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        c += w;
        n -= (size_t)w;
    }
    return true;
}

bool read_all(int fd, void *p, size_t n, double deadline) {
    char *c = (char *)p;
    while (n) {
        pollfd pf{fd, POLLIN, 0};
        int pr = poll(&pf, 1, 200);
        if (pr < 0 && errno != EINTR) return false;
        if (pr <= 0) {
            if (now_s() > deadline) return false;
            continue;
        }
        ssize_t r = ::recv(fd, c, n, 0);
        if (r == 0) return false;
        if (r < 0) {
            if (errno == EINTR || errno == EAGAIN) continue;
            return false;
        }
        c += r;
        n -= (size_t)r;
    }
    return true;
}

sockaddr_un addr_of(const std::string &path) {
    sockaddr_un a{};
    a.sun_family = AF_UNIX;
    snprintf(a.sun_path, sizeof a.sun_path, "%s", path.c_str());
    return a;
}

// run every queued operation to completion: sends were staged to the host already, receives land in host buffers
int progress(Comm *c, std::vector<Op> &ops) {
    const double deadline = now_s() + timeout_s();
    const int n = c->nranks;
    // per peer and direction: indices of the operations in program order
    std::vector<std::vector<size_t>> sq((size_t)n), rq((size_t)n);
    std::vector<size_t> si((size_t)n, 0), ri((size_t)n, 0);
    for (size_t i = 0; i < ops.size(); i++) (ops[i].send ? sq : rq)[(size_t)ops[i].peer].push_back(i);
    for (int p = 0; p < n; p++)
        if (c->fd[(size_t)p] >= 0) set_nonblocking(c->fd[(size_t)p], true);
    int rc = kOk;
    for (;;) {
        std::vector<pollfd> pfs;
        std::vector<int> who;
        for (int p = 0; p < n; p++) {
            short ev = 0;
            while (si[p] < sq[p].size() && ops[sq[p][si[p]]].done == ops[sq[p][si[p]]].nbytes) si[p]++;
            while (ri[p] < rq[p].size() && ops[rq[p][ri[p]]].done == ops[rq[p][ri[p]]].nbytes) ri[p]++;
            if (si[p] < sq[p].size()) ev |= POLLOUT;
            if (ri[p] < rq[p].size()) ev |= POLLIN;
            if (ev) {
                pfs.push_back({c->fd[(size_t)p], ev, 0});
                who.push_back(p);
            }
        }
        if (pfs.empty()) break;
        int pr = poll(pfs.data(), (nfds_t)pfs.size(), 200);
        if (pr < 0 && errno != EINTR) { rc = fail(kSystemError, std::string("poll: ") + strerror(errno)); break; }
        if (now_s() > deadline) {
            rc = fail(kSystemError, "rank " + std::to_string(c->rank) + ": exchange timed out (a peer did not post its matching operations)");
            break;
        }
        if (pr <= 0) continue;
        for (size_t k = 0; k < pfs.size() && rc == kOk; k++) {
            const int p = who[k];
            if (pfs[k].revents & POLLOUT) {
                Op &o = ops[sq[p][si[p]]];
                ssize_t w = ::send(pfs[k].fd, o.host.data() + o.done, o.nbytes - o.done, MSG_NOSIGNAL);
                if (w > 0) o.done += (size_t)w;
                else if (w < 0 && errno != EAGAIN && errno != EINTR) rc = fail(kSystemError, std::string("send to rank ") + std::to_string(p) + ": " + strerror(errno));
            }
            if (pfs[k].revents & POLLIN) {
                Op &o = ops[rq[p][ri[p]]];
                ssize_t r = ::recv(pfs[k].fd, o.host.data() + o.done, o.nbytes - o.done, 0);
                if (r > 0) o.done += (size_t)r;
                else if (r == 0) rc = fail(kSystemError, "rank " + std::to_string(p) + " closed its connection");
                else if (errno != EAGAIN && errno != EINTR) rc = fail(kSystemError, std::string("recv from rank ") + std::to_string(p) + ": " + strerror(errno));
            } else if (pfs[k].revents & (POLLERR | POLLHUP)) {
                rc = fail(kSystemError, "connection to rank " + std::to_string(p) + " broke");
            }
        }
        if (rc != kOk) break;
    }
    for (int p = 0; p < n; p++)
        if (c->fd[(size_t)p] >= 0) set_nonblocking(c->fd[(size_t)p], false);
    return rc;
}

int flush() {
    std::vector<Op> ops;
    ops.swap(g_ops);
    Comm *c = g_comm;
    g_comm = nullptr;
    if (ops.empty() || !c) return kOk;
    int rc = progress(c, ops);
    if (rc != kOk) return rc;
    for (Op &o : ops)
        if (!o.send && o.nbytes)
            if (!copy_bytes(o.dev, o.host.data(), o.nbytes, hipMemcpyHostToDevice)) return fail(kInternalError, "hipMemcpy H2D failed");
    return kOk;
}

int refuse(int code, const std::string &msg) {
    if (g_depth > 0) g_group_failed = true;
    return fail(code, msg);
}

int enqueue(bool send, void *buf, size_t nbytes, int peer, Comm *c, hipStream_t s) {
    if (!c || peer < 0 || peer >= c->nranks) return refuse(kInvalidArgument, "bad communicator / peer");
    if (peer == c->rank) return refuse(kInvalidArgument, "send / recv to self is not supported by the test transport");
    if (g_comm && g_comm != c) return refuse(kInvalidArgument, "one communicator per group");
    g_comm = c;
    g_stream = s;
    Op o;
    o.send = send;
    o.peer = peer;
    o.dev = buf;
    o.nbytes = nbytes;
    o.done = 0;
    o.host.resize(nbytes);
    if (send && nbytes) {   // everything queued on the stream before this call has produced the bytes
        if (!sync_stream(s)) return refuse(kInternalError, "hipStreamSynchronize failed");
        if (!copy_bytes(o.host.data(), buf, nbytes, hipMemcpyDeviceToHost)) return refuse(kInternalError, "hipMemcpy D2H failed");
    }
    g_ops.push_back(std::move(o));
    return g_depth > 0 ? kOk : flush();
}

}  // namespace

extern "C" {

int ncclGetUniqueId(UniqueId *id) {
    if (!id) return kInvalidArgument;
    memset(id, 0, sizeof *id);
    const char *dir = getenv("TMPDIR");
    unsigned long long r = (unsigned long long)(now_s() * 1e9) ^ ((unsigned long long)getpid() << 32);
    snprintf(id->internal, sizeof id->internal, "%s/qe_tt_%d_%llx", dir && *dir && strlen(dir) < 60 ? dir : "/tmp", (int)getpid(), r);
    return kOk;
}

int ncclCommInitRank(void **out, int nranks, UniqueId id, int rank) {
    if (!out || nranks < 1 || rank < 0 || rank >= nranks) return kInvalidArgument;
    id.internal[sizeof id.internal - 1] = 0;
    Comm *c = new Comm();
    c->nranks = nranks;
    c->rank = rank;
    c->base = id.internal;
    c->fd.assign((size_t)nranks, -1);
    const double deadline = now_s() + timeout_s();
    auto bail = [&](const std::string &m) {
        for (int f : c->fd) if (f >= 0) close(f);
        if (c->listen_fd >= 0) close(c->listen_fd);
        unlink((c->base + "." + std::to_string(rank)).c_str());
        delete c;
        return fail(kSystemError, m);
    };
    if (nranks > 1) {
        const std::string mine = c->base + "." + std::to_string(rank);
        c->listen_fd = socket(AF_UNIX, SOCK_STREAM, 0);
        sockaddr_un a = addr_of(mine);
        unlink(mine.c_str());
        if (c->listen_fd < 0 || bind(c->listen_fd, (sockaddr *)&a, sizeof a) != 0 || listen(c->listen_fd, nranks) != 0)
            return bail("bind/listen " + mine + ": " + strerror(errno));
        // a higher rank connects to every lower one
        for (int p = 0; p < rank; p++) {
            sockaddr_un pa = addr_of(c->base + "." + std::to_string(p));
            int f = -1;
            for (;;) {
                f = socket(AF_UNIX, SOCK_STREAM, 0);
                if (f >= 0 && connect(f, (sockaddr *)&pa, sizeof pa) == 0) break;
                if (f >= 0) close(f);
                f = -1;
                if (now_s() > deadline) return bail("rank " + std::to_string(p) + " did not come up");
                usleep(20000);
            }
            int32_t me = rank;
            if (!write_all(f, &me, 4)) { close(f); return bail("handshake with rank " + std::to_string(p) + " failed"); }
            c->fd[(size_t)p] = f;
        }
        for (int k = rank + 1; k < nranks; k++) {
            pollfd pf{c->listen_fd, POLLIN, 0};
            for (;;) {
                int pr = poll(&pf, 1, 200);
                if (pr > 0) break;
                if (now_s() > deadline) return bail("not every higher rank connected");
            }
            int f = accept(c->listen_fd, nullptr, nullptr);
            int32_t who = -1;
            if (f < 0 || !read_all(f, &who, 4, deadline) || who <= rank || who >= nranks || c->fd[(size_t)who] >= 0) {
                if (f >= 0) close(f);
                return bail("bad handshake");
            }
            c->fd[(size_t)who] = f;
        }
    }
    *out = c;
    return kOk;
}

int ncclCommDestroy(void *comm) {
    Comm *c = (Comm *)comm;
    if (!c) return kOk;
    for (int f : c->fd) if (f >= 0) close(f);
    if (c->listen_fd >= 0) {
        close(c->listen_fd);
        unlink((c->base + "." + std::to_string(c->rank)).c_str());
    }
    delete c;
    return kOk;
}

int ncclGroupStart() {
    g_depth++;
    return kOk;
}

int ncclGroupEnd() {
    if (g_depth <= 0) return fail(kInvalidArgument, "ncclGroupEnd without ncclGroupStart");
    if (--g_depth > 0) return kOk;
    if (g_group_failed) {
        g_group_failed = false;
        g_ops.clear();
        g_comm = nullptr;
        return fail(kInvalidArgument, "an operation of this group failed: " + g_err);
    }
    return flush();
}

static size_t dtype_size(int dtype) {   // rccl.h: int8 0, uint8 1, int32 2, uint32 3, int64 4, uint64 5, half 6, float 7, double 8
    static const size_t w[] = {1, 1, 4, 4, 8, 8, 2, 4, 8};
    return dtype >= 0 && dtype < 9 ? w[dtype] : 0;
}

int ncclSend(const void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t s) {
    if (!dtype_size(dtype)) return kInvalidArgument;
    return enqueue(true, (void *)buf, count * dtype_size(dtype), peer, (Comm *)comm, s);
}

int ncclRecv(void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t s) {
    if (!dtype_size(dtype)) return kInvalidArgument;
    return enqueue(false, buf, count * dtype_size(dtype), peer, (Comm *)comm, s);
}

int ncclAllGather(const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t s) {
    Comm *c = (Comm *)comm;
    const size_t nb = count * dtype_size(dtype);
    if (!c || !nb) return kInvalidArgument;
    if (!sync_stream(s)) return fail(kInternalError, "hipStreamSynchronize failed");
    if (!copy_bytes((char *)recv + nb * (size_t)c->rank, send, nb, hipMemcpyDeviceToDevice)) return fail(kInternalError, "hipMemcpy D2D failed");
    if (c->nranks == 1) return kOk;
    int rc = ncclGroupStart();
    for (int p = 0; p < c->nranks && rc == kOk; p++) {
        if (p == c->rank) continue;
        rc = enqueue(true, (void *)send, nb, p, c, s);
        if (rc == kOk) rc = enqueue(false, (char *)recv + nb * (size_t)p, nb, p, c, s);
    }
    const int end = ncclGroupEnd();   // discards the group if an operation was refused
    return rc != kOk ? rc : end;
}

const char *ncclGetErrorString(int r) {
    if (r == kOk) return "no error";
    return g_err.empty() ? "test transport error" : g_err.c_str();
}

}  // extern "C"
