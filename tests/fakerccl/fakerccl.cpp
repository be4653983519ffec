// fakerccl.cpp -- TEST INFRASTRUCTURE, never shipped: a stand-in for the 12 RCCL entry points libpte resolves
// (pigeons.jl_amd/csrc/pte_comm.hpp) that lets >= 2 ranks share ONE GPU.  Real RCCL refuses that ("Duplicate GPU
// detected"), and the development / CI box has one GPU, so without this the production transport
// (RcclShard -> pte_comm_init -> pte_run_scans: ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on the engine's stream)
// would never meet a peer before the 8-GPU node does.  Loaded through libpte's documented override: $PTE_RCCL_LIB.
//
// Semantics kept from RCCL: every call is ENQUEUED on the caller's stream and returns at once; data moves when the stream
// gets there; send/recv inside a group do not deadlock whatever their order; collectives are collective.
// Mechanism: the ranks (separate processes) share a POSIX shm segment named by the unique id.  A send is
//   hipMemcpyAsync(device -> pinned staging) ; hipLaunchHostFunc(copy staging -> ring slot in shm, publish sequence number)
// and a receive is
//   hipLaunchHostFunc(wait for the sequence number, copy slot -> pinned staging, free the slot) ; hipMemcpyAsync(staging -> device).
// Collectives (all-reduce of doubles MAX / SUM, all-gather of bytes) use a per-rank slot area with two-phase sequence numbers.
// Every wait has a timeout and aborts the process loudly: a hang here is a test failure, not a stuck box.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <thread>
#include <vector>

namespace {

constexpr int MAX_RANKS = 8;
constexpr int RING = 4;                               // in-flight messages per directed pair
constexpr size_t SLOT_BYTES = 256u << 10;             // one message (libpte: 8 (sw + 8) bytes; 32 KiB at d = 4096)
constexpr size_t COLL_BYTES = 16u << 20;              // per-rank contribution of one collective
constexpr double TIMEOUT_S = 120.0;

struct Channel {                                      // src -> dst
    std::atomic<uint64_t> head;                       // messages published by src
    std::atomic<uint64_t> tail;                       // messages consumed by dst
    uint64_t bytes[RING];
    alignas(64) unsigned char slot[RING][SLOT_BYTES];
};
struct Shm {
    std::atomic<uint32_t> magic;
    std::atomic<int32_t> arrived, departed;
    int32_t nranks;
    std::atomic<uint64_t> coll_posted[MAX_RANKS];     // collective number this rank has posted its contribution for
    std::atomic<uint64_t> coll_done[MAX_RANKS];       // collective number this rank has finished reading
    alignas(64) Channel ch[MAX_RANKS][MAX_RANKS];
    alignas(64) unsigned char coll[MAX_RANKS][COLL_BYTES];
};

[[noreturn]] void die(const char *what) {
    std::fprintf(stderr, "fakerccl: %s (timeout %.0f s or fatal); aborting this rank\n", what, TIMEOUT_S);
    std::fflush(stderr);
    _exit(97);
}
template <typename F> void wait_until(F ok, const char *what) {
    const auto t0 = std::chrono::steady_clock::now();
    int spins = 0;
    while (!ok()) {
        if (++spins > 2000) { std::this_thread::sleep_for(std::chrono::microseconds(50)); }
        if ((spins & 1023) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > TIMEOUT_S) die(what);
    }
}

struct Op { bool send; void *dev; size_t bytes; int peer; ncclComm_t comm; hipStream_t stream; };
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

size_t dtype_size(ncclDataType_t t) {
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

}  // namespace

struct ncclComm {
    Shm *shm = nullptr;
    char name[64] = {0};
    int rank = 0, nranks = 1;
    uint64_t n_coll = 0;                              // collectives issued on this communicator (same on every rank)
    std::map<int, unsigned char *> stage_send, stage_recv;     // pinned, per peer
    unsigned char *stage_coll = nullptr;              // pinned, COLL_BYTES * (nranks + 1)
    uint64_t sent[MAX_RANKS] = {0}, received[MAX_RANKS] = {0};
    unsigned char *stage(std::map<int, unsigned char *> &m, int peer) {
        auto it = m.find(peer);
        if (it != m.end()) return it->second;
        void *p = nullptr;
        if (hipHostMalloc(&p, SLOT_BYTES, hipHostMallocDefault) != hipSuccess) die("hipHostMalloc staging");
        m[peer] = (unsigned char *)p;
        return (unsigned char *)p;
    }
};

namespace {

struct HostArg { ncclComm_t c; int peer; size_t bytes; unsigned char *stage; uint64_t seq; int kind; ncclRedOp_t op; size_t count; };

void host_send(void *p) {
    HostArg *a = (HostArg *)p;
    Channel &ch = a->c->shm->ch[a->c->rank][a->peer];
    wait_until([&] { return ch.head.load(std::memory_order_relaxed) - ch.tail.load(std::memory_order_acquire) < RING; }, "send: ring full");
    const uint64_t h = ch.head.load(std::memory_order_relaxed);
    std::memcpy(ch.slot[h % RING], a->stage, a->bytes);
    ch.bytes[h % RING] = a->bytes;
    ch.head.store(h + 1, std::memory_order_release);
    delete a;
}
void host_recv(void *p) {
    HostArg *a = (HostArg *)p;
    Channel &ch = a->c->shm->ch[a->peer][a->c->rank];
    const uint64_t t = ch.tail.load(std::memory_order_relaxed);
    wait_until([&] { return ch.head.load(std::memory_order_acquire) > t; }, "recv: no message from the peer");
    if (ch.bytes[t % RING] != a->bytes) { std::fprintf(stderr, "fakerccl: recv of %zu bytes met a message of %llu bytes\n", a->bytes, (unsigned long long)ch.bytes[t % RING]); die("size mismatch"); }
    std::memcpy(a->stage, ch.slot[t % RING], a->bytes);
    ch.tail.store(t + 1, std::memory_order_release);
    delete a;
}
// collective number a->seq: post my contribution, wait for everyone's, combine into the staging area, mark done
void host_coll(void *p) {
    HostArg *a = (HostArg *)p;
    ncclComm_t c = a->c;
    Shm *s = c->shm;
    const int R = c->nranks, me = c->rank;
    // my slot is free once every rank has finished reading the previous collective
    wait_until([&] { for (int r = 0; r < R; ++r) if (s->coll_done[r].load(std::memory_order_acquire) + 1 < a->seq) return false; return true; }, "collective: previous one unfinished");
    std::memcpy(s->coll[me], a->stage, a->bytes);
    s->coll_posted[me].store(a->seq, std::memory_order_release);
    wait_until([&] { for (int r = 0; r < R; ++r) if (s->coll_posted[r].load(std::memory_order_acquire) < a->seq) return false; return true; }, "collective: a rank did not arrive");
    unsigned char *out = a->stage + COLL_BYTES;
    if (a->kind == 0) {                               // all-reduce of doubles
        double *o = (double *)out;
        for (size_t i = 0; i < a->count; ++i) {
            double v = ((const double *)s->coll[0])[i];
            for (int r = 1; r < R; ++r) { const double w = ((const double *)s->coll[r])[i]; v = a->op == ncclMax ? (w > v ? w : v) : v + w; }
            o[i] = v;
        }
    } else {                                          // all-gather
        for (int r = 0; r < R; ++r) std::memcpy(out + (size_t)r * a->bytes, s->coll[r], a->bytes);
    }
    s->coll_done[me].store(a->seq, std::memory_order_release);
    delete a;
}

ncclResult_t enqueue(const Op &o) {
    ncclComm_t c = o.comm;
    if (o.peer < 0 || o.peer >= c->nranks || o.peer == c->rank || o.bytes > SLOT_BYTES) return ncclInvalidArgument;
    if (o.send) {
        unsigned char *st = c->stage(c->stage_send, o.peer);
        if (hipMemcpyAsync(st, o.dev, o.bytes, hipMemcpyDeviceToHost, o.stream) != hipSuccess) return ncclUnhandledCudaError;
        if (hipLaunchHostFunc(o.stream, host_send, new HostArg{c, o.peer, o.bytes, st, 0, 0, ncclSum, 0}) != hipSuccess) return ncclUnhandledCudaError;
    } else {
        unsigned char *st = c->stage(c->stage_recv, o.peer);
        if (hipLaunchHostFunc(o.stream, host_recv, new HostArg{c, o.peer, o.bytes, st, 0, 0, ncclSum, 0}) != hipSuccess) return ncclUnhandledCudaError;
        if (hipMemcpyAsync(o.dev, st, o.bytes, hipMemcpyHostToDevice, o.stream) != hipSuccess) return ncclUnhandledCudaError;
    }
    return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetVersion(int *version) { if (version) *version = 99999; return ncclSuccess; }      // (no RCCL has this version)
const char *ncclGetErrorString(ncclResult_t r) {
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "fakerccl: HIP call failed";
    case ncclInvalidArgument: return "fakerccl: invalid argument";
    case ncclSystemError: return "fakerccl: shm / system error";
    default: return "fakerccl: error";
    }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    static std::atomic<int> counter{0};
    std::memset(id->internal, 0, sizeof id->internal);
    const auto now = std::chrono::steady_clock::now().time_since_epoch().count();
    std::snprintf(id->internal, sizeof id->internal, "/fakerccl_%d_%d_%llx", (int)getpid(), counter++, (unsigned long long)now);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks || id.internal[0] != '/') return ncclInvalidArgument;
    ncclComm *c = new ncclComm;
    std::strncpy(c->name, id.internal, sizeof c->name - 1);
    c->rank = rank; c->nranks = nranks;
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)sizeof(Shm)) != 0) { std::perror("fakerccl: shm_open / ftruncate"); return ncclSystemError; }
    void *m = mmap(nullptr, sizeof(Shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);       // tmpfs: zero-filled, pages appear on touch
    close(fd);
    if (m == MAP_FAILED) { std::perror("fakerccl: mmap"); return ncclSystemError; }
    c->shm = (Shm *)m;
    c->shm->nranks = nranks;
    c->shm->arrived.fetch_add(1, std::memory_order_acq_rel);
    wait_until([&] { return c->shm->arrived.load(std::memory_order_acquire) >= nranks; }, "ncclCommInitRank: not every rank arrived");
    void *p = nullptr;
    if (hipHostMalloc(&p, COLL_BYTES * (size_t)(nranks + 1), hipHostMallocDefault) != hipSuccess) return ncclUnhandledCudaError;
    c->stage_coll = (unsigned char *)p;
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclSuccess;
    (void)hipDeviceSynchronize();
    if (c->shm->departed.fetch_add(1, std::memory_order_acq_rel) + 1 == c->nranks) shm_unlink(c->name);     // the last one out removes the name
    munmap(c->shm, sizeof(Shm));
    for (auto &kv : c->stage_send) (void)hipHostFree(kv.second);
    for (auto &kv : c->stage_recv) (void)hipHostFree(kv.second);
    (void)hipHostFree(c->stage_coll);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t c, int *count) { if (!c || !count) return ncclInvalidArgument; *count = c->nranks; return ncclSuccess; }

ncclResult_t ncclGroupStart() { ++g_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
    if (g_depth <= 0) return ncclInvalidArgument;
    if (--g_depth > 0) return ncclSuccess;
    ncclResult_t rc = ncclSuccess;
    for (int pass = 0; pass < 2 && rc == ncclSuccess; ++pass)          // every send of the group before any receive: no order can deadlock
        for (const Op &o : g_ops)
            if (o.send == (pass == 0) && rc == ncclSuccess) rc = enqueue(o);
    g_ops.clear();
    return rc;
}
ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s) {
    if (!c || !buf || !dtype_size(t)) return ncclInvalidArgument;
    Op o{true, const_cast<void *>(buf), count * dtype_size(t), peer, c, s};
    if (g_depth > 0) { g_ops.push_back(o); return ncclSuccess; }
    return enqueue(o);
}
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s) {
    if (!c || !buf || !dtype_size(t)) return ncclInvalidArgument;
    Op o{false, buf, count * dtype_size(t), peer, c, s};
    if (g_depth > 0) { g_ops.push_back(o); return ncclSuccess; }
    return enqueue(o);
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t s) {
    if (!c || !send || !recv || t != ncclFloat64 || (op != ncclMax && op != ncclSum) || count * 8 > COLL_BYTES) return ncclInvalidArgument;
    const size_t bytes = count * 8;
    if (hipMemcpyAsync(c->stage_coll, send, bytes, hipMemcpyDeviceToHost, s) != hipSuccess) return ncclUnhandledCudaError;
    if (hipLaunchHostFunc(s, host_coll, new HostArg{c, -1, bytes, c->stage_coll, ++c->n_coll, 0, op, count}) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpyAsync(recv, c->stage_coll + COLL_BYTES, bytes, hipMemcpyHostToDevice, s) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}
ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t s) {
    const size_t bytes = count * dtype_size(t);
    if (!c || !send || !recv || !bytes || bytes > COLL_BYTES) return ncclInvalidArgument;
    if (hipMemcpyAsync(c->stage_coll, send, bytes, hipMemcpyDeviceToHost, s) != hipSuccess) return ncclUnhandledCudaError;
    if (hipLaunchHostFunc(s, host_coll, new HostArg{c, -1, bytes, c->stage_coll, ++c->n_coll, 1, ncclSum, 0}) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpyAsync(recv, c->stage_coll + COLL_BYTES, bytes * (size_t)c->nranks, hipMemcpyHostToDevice, s) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

}  // extern "C"
