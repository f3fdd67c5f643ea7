// Test double for librccl.so (test infrastructure): the handful of entry points
// libopty_hip.so binds (csrc/comm.cpp: load_rccl), implemented over files in
// /dev/shm so that SEVERAL RANKS CAN SHARE ONE GPU -- RCCL itself refuses
// duplicate devices, and no multi-GPU box is available to the build.  It lets
// the GPU tests drive opty_hip_bcast_free / opty_hip_gather_v with 2 and 3
// ranks (peer loops, staging offsets, strided copies, unequal shards) before
// the real library ever carries more than one rank.
//
// Semantics kept: the k-th ncclSend of rank a to rank b matches the k-th
// ncclRecv of b from a; calls between ncclGroupStart / ncclGroupEnd progress
// together (all sends are published before any receive blocks); results are
// ordered with the stream (this double synchronises it).
//
//   hipcc -shared -fPIC -O1 tests/fake_rccl/fake_rccl.cpp -o <dir>/libfake_rccl.so
//   OPTY_HIP_RCCL_LIBRARY=<dir>/libfake_rccl.so
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

struct Comm {
    std::string base;
    int rank = 0, world = 1;
    std::map<int, long> sent, received;     // per peer message counters
};

struct Op {
    bool send;
    void *buf;
    size_t bytes;
    int peer;
    Comm *comm;
    hipStream_t stream;
};

thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
const char *g_error = "fake rccl: unspecified failure";

std::string mailbox(const Comm *c, int src, int dst, long seq) {
    char name[256];
    snprintf(name, sizeof name, "%s_%d_%d_%ld", c->base.c_str(), src, dst,
             seq);
    return name;
}

int publish(const std::string &path, const void *data, size_t bytes) {
    const std::string tmp = path + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) { g_error = "fake rccl: cannot create a mailbox"; return 1; }
    const size_t n = bytes ? fwrite(data, 1, bytes, f) : 0;
    fclose(f);
    if (n != bytes) { g_error = "fake rccl: short write"; return 1; }
    return rename(tmp.c_str(), path.c_str()) == 0 ? 0 : 1;
}

int collect(const std::string &path, void *data, size_t bytes) {
    const auto t0 = std::chrono::steady_clock::now();
    FILE *f = nullptr;
    while (!(f = fopen(path.c_str(), "rb"))) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(90)) {
            g_error = "fake rccl: timed out waiting for a message";
            return 1;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    const size_t n = bytes ? fread(data, 1, bytes, f) : 0;
    fclose(f);
    unlink(path.c_str());
    if (n != bytes) { g_error = "fake rccl: message of another size"; return 1; }
    return 0;
}

int run(const std::vector<Op> &ops) {
    std::vector<char> host;
    // everything the streams hold must be done before the buffers are read
    for (const Op &op : ops)
        if (hipStreamSynchronize(op.stream) != hipSuccess) return 1;
    for (const Op &op : ops) {
        if (!op.send) continue;
        host.resize(op.bytes);
        if (op.bytes && hipMemcpy(host.data(), op.buf, op.bytes,
                                  hipMemcpyDeviceToHost) != hipSuccess)
            return 1;
        const long seq = op.comm->sent[op.peer]++;
        if (publish(mailbox(op.comm, op.comm->rank, op.peer, seq),
                    host.data(), op.bytes))
            return 1;
    }
    for (const Op &op : ops) {
        if (op.send) continue;
        host.resize(op.bytes);
        const long seq = op.comm->received[op.peer]++;
        if (collect(mailbox(op.comm, op.peer, op.comm->rank, seq),
                    host.data(), op.bytes))
            return 1;
        if (op.bytes && hipMemcpy(op.buf, host.data(), op.bytes,
                                  hipMemcpyHostToDevice) != hipSuccess)
            return 1;
    }
    return 0;
}

int submit(const Op &op) {
    if (g_depth > 0) {
        g_ops.push_back(op);
        return 0;
    }
    return run({op});
}

}  // namespace

extern "C" {

int ncclGetUniqueId(void *id) {
    unsigned char *p = static_cast<unsigned char *>(id);
    memset(p, 0, 128);
    FILE *f = fopen("/dev/urandom", "rb");
    if (!f || fread(p, 1, 16, f) != 16) return 1;
    fclose(f);
    return 0;
}

struct FakeId { char b[128]; };

int ncclCommInitRank(void **comm, int nranks, FakeId id, int rank) {
    auto *c = new Comm;
    char hex[40];
    for (int k = 0; k < 16; ++k)
        snprintf(hex + 2*k, 3, "%02x", (unsigned char)id.b[k]);
    c->base = std::string("/dev/shm/opty_fake_rccl_") + hex;
    c->rank = rank;
    c->world = nranks;
    // rendezvous: everybody says hello to everybody (as the real call blocks
    // until all ranks have arrived)
    std::vector<Op> ops;
    static char token;
    for (int g = 0; g < nranks; ++g) {
        if (g == rank) continue;
        const long s = c->sent[g]++;
        if (publish(mailbox(c, rank, g, s), &token, 0)) return 1;
    }
    for (int g = 0; g < nranks; ++g) {
        if (g == rank) continue;
        const long s = c->received[g]++;
        if (collect(mailbox(c, g, rank, s), &token, 0)) return 1;
    }
    *comm = c;
    return 0;
}

int ncclCommDestroy(void *comm) {
    delete static_cast<Comm *>(comm);
    return 0;
}

int ncclGroupStart() {
    ++g_depth;
    return 0;
}

int ncclGroupEnd() {
    if (--g_depth > 0) return 0;
    std::vector<Op> ops;
    ops.swap(g_ops);
    return run(ops);
}

int ncclSend(const void *buf, size_t count, int dtype, int peer, void *comm,
             hipStream_t stream) {
    if (dtype != 8) { g_error = "fake rccl: float64 only"; return 1; }
    return submit(Op{true, const_cast<void *>(buf), count*8, peer,
                     static_cast<Comm *>(comm), stream});
}

int ncclRecv(void *buf, size_t count, int dtype, int peer, void *comm,
             hipStream_t stream) {
    if (dtype != 8) { g_error = "fake rccl: float64 only"; return 1; }
    return submit(Op{false, buf, count*8, peer, static_cast<Comm *>(comm),
                     stream});
}

int ncclBroadcast(const void *send, void *recv, size_t count, int dtype,
                  int root, void *comm, hipStream_t stream) {
    if (dtype != 8) { g_error = "fake rccl: float64 only"; return 1; }
    Comm *c = static_cast<Comm *>(comm);
    std::vector<Op> ops;
    if (c->rank == root) {
        for (int g = 0; g < c->world; ++g)
            if (g != root)
                ops.push_back(Op{true, const_cast<void *>(send), count*8, g,
                                 c, stream});
        if (recv != send &&
            hipMemcpyAsync(recv, send, count*8, hipMemcpyDeviceToDevice,
                           stream) != hipSuccess)
            return 1;
    } else {
        ops.push_back(Op{false, recv, count*8, root, c, stream});
    }
    return run(ops);
}

const char *ncclGetErrorString(int) { return g_error; }

}  // extern "C"
