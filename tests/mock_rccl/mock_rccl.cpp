// mock_rccl.cpp — a loopback transport with RCCL's C signatures, for TESTS ONLY.
//
// liborama_hip.so binds nine RCCL symbols with dlopen (oramacore_amd/csrc/shard_group.hip, load_rccl).  A 1-GPU box
// cannot run real RCCL with more than one rank (RCCL refuses two ranks on one device), so the code that only runs
// with world > 1 — slot_of() with rank0 > 0, the GroupStart/GroupEnd loop over several communicators, the index-wide
// df / min-max / count reductions across real processes — would never execute before an 8-GPU job.  This library
// exports the same nine symbols and moves the bytes through a POSIX shared-memory segment + hipMemcpy, so that
// world-2 / world-3 jobs (separate PROCESSES, each with its own HIP context, all on GPU 0) and one-process groups of
// several communicators run the product's exchange code unchanged.  Selected with ORAMA_RCCL_LIB=<path to this .so>.
//
// Semantics kept: collectives are ordered with the stream they are issued on (the mock drains the stream, then does
// the exchange on the host — slower than a device collective, equivalent in ordering); every rank of a communicator
// must issue the same collectives in the same order; calls between ncclGroupStart/End are deferred to ncclGroupEnd
// (that is what lets ONE thread drive several ranks without deadlock); in-place all-gather / all-reduce.
// Not kept: performance, asynchrony, any data type or operator the product does not use.
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

constexpr int kMaxWorld = 16;
constexpr size_t kSlotBytes = 4u << 20;  // per-rank payload window (largest product block: 256 x 100 x 12 B = 307 KB)
constexpr double kTimeoutSeconds = 120.0;

enum { kOk = 0, kUnhandledCuda = 1, kSystem = 2, kInternal = 3, kInvalidArgument = 4, kInvalidUsage = 5 };

struct ShmHeader {
    std::atomic<uint32_t> attached;   // ranks that mapped the segment
    std::atomic<uint32_t> detached;
    std::atomic<uint32_t> aborted;
    std::atomic<uint64_t> arrivals;   // barrier counter: phase p is complete when arrivals >= (p + 1) * world
};

struct Segment {
    ShmHeader* hdr = nullptr;
    char* data = nullptr;  // world windows of kSlotBytes
    size_t bytes = 0;
    char name[64] = {0};
    int users = 0;         // communicators of THIS process on the segment
};

}  // namespace

struct ncclComm {
    Segment* seg = nullptr;
    int rank = 0, world = 1, device = 0;
    uint64_t phase = 0;  // barrier phases this rank has passed
};
typedef ncclComm* ncclComm_t;
struct ncclUniqueId {
    char internal[128];
};

namespace {

// MOCK_RCCL_HOST_BUFFERS=1: the buffers are host memory and streams are ignored — lets the transport itself (segment,
// barriers, groups, reductions) be tested on a box without a GPU (tests/test_mock_rccl.py).
bool host_buffers() {
    static const bool v = [] {
        const char* e = getenv("MOCK_RCCL_HOST_BUFFERS");
        return e && *e == '1';
    }();
    return v;
}
bool copy_out(void* dst_host, const void* src, size_t bytes) {
    if (host_buffers()) {
        memcpy(dst_host, src, bytes);
        return true;
    }
    return hipMemcpy(dst_host, src, bytes, hipMemcpyDeviceToHost) == hipSuccess;
}
bool copy_in(void* dst, const void* src_host, size_t bytes) {
    if (host_buffers()) {
        memcpy(dst, src_host, bytes);
        return true;
    }
    return hipMemcpy(dst, src_host, bytes, hipMemcpyHostToDevice) == hipSuccess;
}

double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

Segment* open_segment(const char* name, int world) {
    const size_t bytes = 4096 + (size_t)world * kSlotBytes;
    const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return nullptr;
    if (ftruncate(fd, (off_t)bytes) != 0) {
        close(fd);
        return nullptr;
    }
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return nullptr;
    Segment* s = new Segment();
    s->hdr = static_cast<ShmHeader*>(p);  // a fresh segment is zero-filled: every counter starts at 0
    s->data = static_cast<char*>(p) + 4096;
    s->bytes = bytes;
    snprintf(s->name, sizeof(s->name), "%s", name);
    return s;
}

void close_segment(Segment* s) {
    if (!s) return;
    shm_unlink(s->name);  // idempotent; the mapping stays valid for whoever still holds it
    munmap(s->hdr, s->bytes);
    delete s;
}

// One barrier phase for the `m` ranks this thread drives (they all sit at the same phase).
int barrier(ncclComm_t const* comms, int m) {
    Segment* s = comms[0]->seg;
    const uint64_t target = (comms[0]->phase + 1) * (uint64_t)comms[0]->world;
    s->hdr->arrivals.fetch_add((uint64_t)m, std::memory_order_acq_rel);
    const double t0 = now_s();
    unsigned spins = 0;
    while (s->hdr->arrivals.load(std::memory_order_acquire) < target) {
        if (s->hdr->aborted.load(std::memory_order_relaxed)) return kSystem;
        if ((++spins & 1023) == 0) {
            if (now_s() - t0 > kTimeoutSeconds) {
                s->hdr->aborted.store(1);
                fprintf(stderr, "[mock_rccl] rank %d: barrier timed out (a rank is missing or issued another collective)\n",
                        comms[0]->rank);
                return kSystem;
            }
            usleep(50);
        }
    }
    for (int i = 0; i < m; ++i) comms[i]->phase++;
    return kOk;
}

struct Op {
    bool reduce = false;
    const void* send = nullptr;
    void* recv = nullptr;
    size_t count = 0;  // elements (all-reduce) / elements per rank (all-gather)
    int dtype = 0, op = 0;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
};

size_t dtype_size(int dt) {
    switch (dt) {
        case 0: case 1: return 1;          // int8 / uint8
        case 2: case 3: case 7: return 4;  // int32 / uint32 / float32
        case 4: case 5: case 8: return 8;  // int64 / uint64 / float64
        default: return 0;
    }
}

template <typename T>
void reduce_typed(const char* windows, int world, size_t count, int op, T* out) {
    for (size_t i = 0; i < count; ++i) {
        T acc = reinterpret_cast<const T*>(windows)[i];
        for (int r = 1; r < world; ++r) {
            const T v = reinterpret_cast<const T*>(windows + (size_t)r * kSlotBytes)[i];
            if (op == 0) acc = acc + v;
            else if (op == 2) acc = v > acc ? v : acc;
            else acc = v < acc ? v : acc;
        }
        out[i] = acc;
    }
}

// The deferred calls of one group (or one ungrouped call): every communicator must have queued the same number of
// operations; round r executes the r-th operation of each.
int run_ops(std::vector<Op>& ops) {
    if (ops.empty()) return kOk;
    std::vector<ncclComm_t> comms;
    for (const Op& o : ops) {
        bool seen = false;
        for (ncclComm_t c : comms) seen |= c == o.comm;
        if (!seen) comms.push_back(o.comm);
    }
    const size_t rounds = ops.size() / comms.size();
    if (rounds * comms.size() != ops.size()) return kInvalidUsage;
    int saved_dev = 0;
    if (!host_buffers()) (void)hipGetDevice(&saved_dev);
    int rc = kOk;
    std::vector<size_t> cursor(comms.size(), 0);
    for (size_t r = 0; r < rounds && rc == kOk; ++r) {
        std::vector<Op*> cur;
        for (size_t ci = 0; ci < comms.size(); ++ci) {
            size_t seen = 0;
            for (Op& o : ops)
                if (o.comm == comms[ci] && seen++ == r) cur.push_back(&o);
        }
        if (cur.size() != comms.size()) return kInvalidUsage;
        // publish: stream drained (ordering), payload into the rank's window
        for (Op* o : cur) {
            const size_t bytes = o->count * dtype_size(o->dtype);
            if (bytes == 0 || bytes > kSlotBytes) return kInvalidArgument;
            if (!host_buffers()) {
                (void)hipSetDevice(o->comm->device);
                if (hipStreamSynchronize(o->stream) != hipSuccess) return kUnhandledCuda;
            }
            if (!copy_out(o->comm->seg->data + (size_t)o->comm->rank * kSlotBytes, o->send, bytes)) return kUnhandledCuda;
        }
        if ((rc = barrier(comms.data(), (int)comms.size())) != kOk) break;
        // collect
        for (Op* o : cur) {
            const size_t esz = dtype_size(o->dtype), bytes = o->count * esz;
            const char* win = o->comm->seg->data;
            if (!host_buffers()) (void)hipSetDevice(o->comm->device);
            if (!o->reduce) {
                for (int rk = 0; rk < o->comm->world; ++rk)
                    if (!copy_in(static_cast<char*>(o->recv) + (size_t)rk * bytes, win + (size_t)rk * kSlotBytes, bytes))
                        return kUnhandledCuda;
            } else {
                std::vector<char> tmp(bytes);
                switch (o->dtype) {
                    case 2: reduce_typed<int32_t>(win, o->comm->world, o->count, o->op, reinterpret_cast<int32_t*>(tmp.data())); break;
                    case 4: reduce_typed<int64_t>(win, o->comm->world, o->count, o->op, reinterpret_cast<int64_t*>(tmp.data())); break;
                    case 5: reduce_typed<uint64_t>(win, o->comm->world, o->count, o->op, reinterpret_cast<uint64_t*>(tmp.data())); break;
                    case 7: reduce_typed<float>(win, o->comm->world, o->count, o->op, reinterpret_cast<float*>(tmp.data())); break;
                    case 8: reduce_typed<double>(win, o->comm->world, o->count, o->op, reinterpret_cast<double*>(tmp.data())); break;
                    default: return kInvalidArgument;
                }
                if (!copy_in(o->recv, tmp.data(), bytes)) return kUnhandledCuda;
            }
        }
        // nobody may overwrite a window before every rank has read it
        rc = barrier(comms.data(), (int)comms.size());
    }
    if (!host_buffers()) (void)hipSetDevice(saved_dev);
    return rc;
}

thread_local int g_group_depth = 0;
thread_local std::vector<Op> g_deferred;

int submit(const Op& op) {
    if (!op.comm || !op.comm->seg) return kInvalidArgument;
    if (g_group_depth > 0) {
        g_deferred.push_back(op);
        return kOk;
    }
    std::vector<Op> one{op};
    return run_ops(one);
}

}  // namespace

extern "C" {

int ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return kInvalidArgument;
    memset(id->internal, 0, sizeof(id->internal));
    timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    static std::atomic<unsigned> seq{0};
    snprintf(id->internal, sizeof(id->internal), "/orama_mock_rccl_%d_%ld_%u", (int)getpid(), (long)ts.tv_nsec, seq++);
    return kOk;
}

int ncclCommInitRank(ncclComm_t* comm, int world, ncclUniqueId id, int rank) {
    if (!comm || world < 1 || world > kMaxWorld || rank < 0 || rank >= world || id.internal[0] != '/') return kInvalidArgument;
    Segment* s = open_segment(id.internal, world);
    if (!s) return kSystem;
    ncclComm* c = new ncclComm();
    c->seg = s;
    s->users = 1;
    c->rank = rank;
    c->world = world;
    if (!host_buffers()) (void)hipGetDevice(&c->device);
    s->hdr->attached.fetch_add(1);
    const double t0 = now_s();
    while (s->hdr->attached.load() < (uint32_t)world) {  // like the real call: returns once every rank has joined
        if (now_s() - t0 > kTimeoutSeconds) {
            fprintf(stderr, "[mock_rccl] rank %d: only %u of %d ranks joined\n", rank, s->hdr->attached.load(), world);
            close_segment(s);
            delete c;
            return kSystem;
        }
        usleep(200);
    }
    *comm = c;
    return kOk;
}

// One process, `n` communicators.  Real RCCL refuses a device that appears twice; the mock accepts it (that is the
// point: several ranks on the one GPU of the test box).
int ncclCommInitAll(ncclComm_t* comms, int n, const int* devices) {
    if (!comms || n < 1 || n > kMaxWorld) return kInvalidArgument;
    ncclUniqueId id;
    ncclGetUniqueId(&id);
    Segment* s = open_segment(id.internal, n);
    if (!s) return kSystem;
    s->users = n;
    s->hdr->attached.store((uint32_t)n);
    for (int i = 0; i < n; ++i) {
        ncclComm* c = new ncclComm();
        c->seg = s;
        c->rank = i;
        c->world = n;
        c->device = devices ? devices[i] : i;
        comms[i] = c;
    }
    return kOk;
}

int ncclCommDestroy(ncclComm_t comm) {
    if (!comm) return kOk;
    Segment* s = comm->seg;
    if (s) {
        s->hdr->detached.fetch_add(1);
        if (--s->users == 0) close_segment(s);
    }
    delete comm;
    return kOk;
}

int ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, int datatype, ncclComm_t comm, hipStream_t stream) {
    Op o;
    o.reduce = false;
    o.send = sendbuff;
    o.recv = recvbuff;
    o.count = sendcount;
    o.dtype = datatype;
    o.comm = comm;
    o.stream = stream;
    return submit(o);
}

int ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, int datatype, int op, ncclComm_t comm, hipStream_t stream) {
    if (op != 0 && op != 2 && op != 3) return kInvalidArgument;
    Op o;
    o.reduce = true;
    o.send = sendbuff;
    o.recv = recvbuff;
    o.count = count;
    o.dtype = datatype;
    o.op = op;
    o.comm = comm;
    o.stream = stream;
    return submit(o);
}

int ncclGroupStart() {
    ++g_group_depth;
    return kOk;
}

int ncclGroupEnd() {
    if (g_group_depth <= 0) return kInvalidUsage;
    if (--g_group_depth > 0) return kOk;
    std::vector<Op> ops;
    ops.swap(g_deferred);
    return run_ops(ops);
}

const char* ncclGetErrorString(int e) {
    switch (e) {
        case kOk: return "no error";
        case kUnhandledCuda: return "mock_rccl: HIP call failed";
        case kSystem: return "mock_rccl: shared-memory transport failed or a peer never arrived";
        case kInvalidArgument: return "mock_rccl: invalid argument (payload over the 4 MiB window, unknown type or operator)";
        case kInvalidUsage: return "mock_rccl: invalid usage (unbalanced group)";
        default: return "mock_rccl: internal error";
    }
}

}  // extern "C"
