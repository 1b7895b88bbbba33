// sc_comm.hip -- the exchange of trial-sharded accumulator records as C calls (SURVEY section 8(b) `sc_allreduce`, 8(e)).
//
// The reference has no multi-device path (its CuPy backend is one GPU: transforms.py:405-439); the contract's multi-GPU row
// shards the trials over one process per GPU and sums the un-normalised records (connectivity.py:67-75, :489: the expectation
// is a plain sum).  The PyTorch host does that exchange through torch.distributed (parallel.py); these entry points put the
// same steps behind the C ABI for a host that has no torch: RCCL over xGMI, one communicator per process.
//
//   sc_comm_exchange_blocks_f32   block j of the local record buffer -> rank j, block i of the receive buffer <- rank i
//                                 (N - 1 concurrent point-to-point transfers, one per xGMI link: the "direct" reduce-scatter
//                                 of parallel.py, whose sum over the received blocks sc_measure_multi_parts takes in rank order)
//   sc_comm_allreduce_f32         in-place sum of whole records (the ring form)
//   sc_comm_gather_f32            the measures of the owned bins -> one root
//
// RCCL is looked up at run time (dlopen of librccl.so.1: the copy PyTorch loaded first when the PyTorch host is in the
// process, /opt/rocm's otherwise) so that libsc_hip.so itself has no link-time dependency on it: without RCCL the calls
// return SC_EUNSUPPORTED and everything else works.
#include <dlfcn.h>
#include <cstring>
#include <mutex>
#include "sc_common.h"
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// Built without RCCL's headers: the handful of declarations the run-time lookup needs (the values are RCCL's public ABI);
// without librccl.so.1 at run time every entry point returns SC_EUNSUPPORTED either way.
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclFloat = 7 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId*);
ncclResult_t ncclCommInitRank(ncclComm_t*, int, ncclUniqueId, int);
ncclResult_t ncclCommDestroy(ncclComm_t);
ncclResult_t ncclAllReduce(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
ncclResult_t ncclSend(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
ncclResult_t ncclRecv(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
ncclResult_t ncclGroupStart(void);
ncclResult_t ncclGroupEnd(void);
const char* ncclGetErrorString(ncclResult_t);
}
#endif

namespace {
struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};
RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
        }
        if (!api.handle) return;
#define SC_RCCL_SYM(field, symbol) api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.handle, symbol))
        SC_RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
        SC_RCCL_SYM(CommInitRank, "ncclCommInitRank");
        SC_RCCL_SYM(CommDestroy, "ncclCommDestroy");
        SC_RCCL_SYM(AllReduce, "ncclAllReduce");
        SC_RCCL_SYM(Send, "ncclSend");
        SC_RCCL_SYM(Recv, "ncclRecv");
        SC_RCCL_SYM(GroupStart, "ncclGroupStart");
        SC_RCCL_SYM(GroupEnd, "ncclGroupEnd");
        SC_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef SC_RCCL_SYM
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce && api.Send && api.Recv && api.GroupStart &&
                 api.GroupEnd && api.GetErrorString;
    });
    return api;
}
}  // namespace

struct sc_comm {
    ncclComm_t comm;
    int n_ranks, rank;
};

#define SC_NEED_RCCL()                                                                                   \
    RcclApi& api = rccl();                                                                               \
    if (!api.ok) {                                                                                       \
        sc_set_error("RCCL (librccl.so.1) could not be loaded: the exchange entry points are unavailable"); \
        return SC_EUNSUPPORTED;                                                                          \
    }
#define SC_CHECK_RCCL(expr)                                                                              \
    do {                                                                                                 \
        const ncclResult_t r_ = (expr);                                                                  \
        if (r_ != ncclSuccess) {                                                                         \
            sc_set_error("%s failed: %s", #expr, api.GetErrorString(r_));                                \
            return SC_EHIP;                                                                              \
        }                                                                                                \
    } while (0)

extern "C" int sc_comm_available(void) { return rccl().ok ? 1 : 0; }

extern "C" int sc_comm_unique_id(void* id128) {
    SC_REQUIRE(id128 != nullptr, "NULL argument");
    SC_NEED_RCCL();
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    SC_CHECK_RCCL(api.GetUniqueId(reinterpret_cast<ncclUniqueId*>(id128)));
    return SC_OK;
}

extern "C" int sc_comm_create(const void* id128, int n_ranks, int rank, sc_comm** out) {
    SC_REQUIRE(id128 != nullptr && out != nullptr, "NULL argument");
    SC_REQUIRE(n_ranks >= 1 && rank >= 0 && rank < n_ranks, "rank outside the communicator");
    *out = nullptr;
    SC_NEED_RCCL();
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t c = nullptr;
    SC_CHECK_RCCL(api.CommInitRank(&c, n_ranks, id, rank));           // (on the device that is current)
    *out = new sc_comm{c, n_ranks, rank};
    return SC_OK;
}

extern "C" int sc_comm_destroy(sc_comm* c) {
    if (!c) return SC_OK;
    RcclApi& api = rccl();
    if (api.ok && c->comm) (void)api.CommDestroy(c->comm);
    delete c;
    return SC_OK;
}

extern "C" int sc_comm_size(const sc_comm* c, int* n_ranks, int* rank) {
    SC_REQUIRE(c != nullptr, "NULL argument");
    if (n_ranks) *n_ranks = c->n_ranks;
    if (rank) *rank = c->rank;
    return SC_OK;
}

extern "C" int sc_comm_allreduce_f32(sc_comm* c, float* d_buf, int64_t n, void* stream) {
    SC_REQUIRE(c != nullptr && d_buf != nullptr && n >= 1, "bad argument");
    SC_NEED_RCCL();
    SC_CHECK_RCCL(api.AllReduce(d_buf, d_buf, (size_t)n, ncclFloat, ncclSum, c->comm, (hipStream_t)stream));
    return SC_OK;
}

// d_send: n_ranks blocks of `block` floats (block j is what rank j owns); d_recv: n_ranks blocks, block i = rank i's block for
// THIS rank (the own block is copied).  Their sum in rank order is this rank's share of the total record.
extern "C" int sc_comm_exchange_blocks_f32(sc_comm* c, const float* d_send, float* d_recv, int64_t block, void* stream) {
    SC_REQUIRE(c != nullptr && d_send != nullptr && d_recv != nullptr && block >= 1, "bad argument");
    SC_NEED_RCCL();
    hipStream_t s = (hipStream_t)stream;
    SC_CHECK_RCCL(api.GroupStart());
    // Inside the bracket nothing returns: a failed call is remembered (the first one), the group is always closed -- an open group
    // would swallow every later call on this communicator -- and the error is reported after ncclGroupEnd.
    ncclResult_t first = ncclSuccess;
    const char* what = "";
    for (int peer = 0; peer < c->n_ranks && first == ncclSuccess; ++peer) {
        if (peer == c->rank) continue;
        first = api.Send(d_send + (int64_t)peer * block, (size_t)block, ncclFloat, peer, c->comm, s);
        if (first != ncclSuccess) { what = "ncclSend"; break; }
        first = api.Recv(d_recv + (int64_t)peer * block, (size_t)block, ncclFloat, peer, c->comm, s);
        if (first != ncclSuccess) what = "ncclRecv";
    }
    const ncclResult_t end = api.GroupEnd();
    if (first != ncclSuccess) { sc_set_error("%s failed inside the exchange group: %s", what, api.GetErrorString(first)); return SC_EHIP; }
    SC_CHECK_RCCL(end);
    SC_CHECK_HIP(hipMemcpyAsync(d_recv + (int64_t)c->rank * block, d_send + (int64_t)c->rank * block, (size_t)block * sizeof(float),
                                hipMemcpyDeviceToDevice, s));
    return SC_OK;
}

// n floats of every rank -> d_recv[rank][n] on `root` (d_recv may be NULL elsewhere)
extern "C" int sc_comm_gather_f32(sc_comm* c, const float* d_send, float* d_recv, int64_t n, int root, void* stream) {
    SC_REQUIRE(c != nullptr && d_send != nullptr && n >= 1 && root >= 0 && root < c->n_ranks, "bad argument");
    SC_REQUIRE(c->rank != root || d_recv != nullptr, "the root needs a receive buffer");
    SC_NEED_RCCL();
    hipStream_t s = (hipStream_t)stream;
    if (c->rank == root) {
        SC_CHECK_RCCL(api.GroupStart());
        ncclResult_t first = ncclSuccess;                       // (as in sc_comm_exchange_blocks_f32: the group is always closed)
        for (int peer = 0; peer < c->n_ranks && first == ncclSuccess; ++peer)
            if (peer != root) first = api.Recv(d_recv + (int64_t)peer * n, (size_t)n, ncclFloat, peer, c->comm, s);
        const ncclResult_t end = api.GroupEnd();
        if (first != ncclSuccess) { sc_set_error("ncclRecv failed inside the gather group: %s", api.GetErrorString(first)); return SC_EHIP; }
        SC_CHECK_RCCL(end);
        SC_CHECK_HIP(hipMemcpyAsync(d_recv + (int64_t)root * n, d_send, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, s));
    } else {
        SC_CHECK_RCCL(api.Send(d_send, (size_t)n, ncclFloat, root, c->comm, s));
    }
    return SC_OK;
}
