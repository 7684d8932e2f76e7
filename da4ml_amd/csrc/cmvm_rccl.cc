// cmvm_rccl.cc -- see cmvm_rccl.h.  The five RCCL entry points used are resolved with dlsym from librccl.so (ROCm's NCCL:
// `ncclGetUniqueId`, `ncclCommInitRank`, `ncclAllReduce`, `ncclCommDestroy`, `ncclGetErrorString`); their prototypes are the
// published NCCL API, restated here so that neither the RCCL headers nor the library are needed to build.

#include "cmvm_rccl.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>

namespace da {
namespace gpu {
namespace {

constexpr int ID_BYTES = 128;  // NCCL_UNIQUE_ID_BYTES
struct UniqueId {
    char internal[ID_BYTES];
};
using Comm = void *;                  // ncclComm_t
constexpr int NCCL_INT32 = 2, NCCL_SUM = 0;  // ncclDataType_t ncclInt32, ncclRedOp_t ncclSum

struct Api {
    int (*get_unique_id)(UniqueId *) = nullptr;
    int (*comm_init_rank)(Comm *, int, UniqueId, int) = nullptr;
    int (*all_reduce)(const void *, void *, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*comm_destroy)(Comm) = nullptr;
    const char *(*error_string)(int) = nullptr;
};

const Api &api() {
    static Api a;
    static std::once_flag once;
    static std::string err;
    std::call_once(once, [] {
        // The RCCL that belongs to the HIP runtime THIS library runs on: librccl.so from the directory of the libamdhip64.so that
        // provides hipStreamSynchronize here.  A process may hold another pair (PyTorch ships its own HIP runtime and RCCL): a communicator
        // of that copy cannot work on this library's streams ("unhandled cuda error" from ncclCommInitRank, measured).  RTLD_LOCAL:
        // the copy's symbols must not become visible to the other one (opened RTLD_GLOBAL before torch initialised its own, the
        // process ended in `double free or corruption` at exit).
        void *h = nullptr;
        {
            Dl_info info;
            std::memset(&info, 0, sizeof info);
            hipError_t (*probe)(hipStream_t) = &hipStreamSynchronize;  // a function of the HIP runtime this library is bound to
            if (dladdr(reinterpret_cast<const void *>(probe), &info) && info.dli_fname) {
                std::string dir(info.dli_fname);
                const size_t slash = dir.rfind('/');
                if (slash != std::string::npos) {
                    dir.resize(slash + 1);
                    for (const char *name : {"librccl.so", "librccl.so.1"}) {
                        h = dlopen((dir + name).c_str(), RTLD_NOW | RTLD_LOCAL);
                        if (h) break;
                    }
                }
            }
        }
        if (!h)
            for (const char *name : {"/opt/rocm/lib/librccl.so", "librccl.so", "librccl.so.1"}) {
                h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (h) break;
            }
        if (!h) {
            err = std::string("librccl.so cannot be loaded: ") + (dlerror() ? dlerror() : "not found");
            return;
        }
        auto sym = [&](const char *n) {
            void *p = dlsym(h, n);
            if (!p && err.empty()) err = std::string("librccl.so lacks ") + n;
            return p;
        };
        a.get_unique_id = reinterpret_cast<decltype(a.get_unique_id)>(sym("ncclGetUniqueId"));
        a.comm_init_rank = reinterpret_cast<decltype(a.comm_init_rank)>(sym("ncclCommInitRank"));
        a.all_reduce = reinterpret_cast<decltype(a.all_reduce)>(sym("ncclAllReduce"));
        a.comm_destroy = reinterpret_cast<decltype(a.comm_destroy)>(sym("ncclCommDestroy"));
        a.error_string = reinterpret_cast<decltype(a.error_string)>(sym("ncclGetErrorString"));
    });
    if (!err.empty()) throw std::runtime_error(err);
    return a;
}

std::string nccl_message(const char *what, int rc) {
    const Api &a = api();
    return std::string(what) + ": " + (a.error_string ? a.error_string(rc) : "RCCL error") + " (" + std::to_string(rc) + ")";
}

class Transport : public RcclTransport {
  public:
    Transport(Comm comm, hipStream_t stream) : comm_(comm), stream_(stream) {}
    ~Transport() override {
        // the communicator lives as long as the process uses its id (cache below); destroying communicators during interpreter
        // shutdown can hang in the network teardown, so it is left to process exit -- or to an explicit rccl_shutdown() while the
        // process group is still alive
        if (stage_) (void)hipFree(stage_);
    }
    // explicit release (rccl_shutdown): the communicator is destroyed, the transport refuses further use
    void destroy() {
        if (comm_) (void)api().comm_destroy(comm_);
        comm_ = nullptr;
        if (stage_) (void)hipFree(stage_);
        stage_ = nullptr;
        stage_bytes_ = 0;
    }
    bool allreduce(void *buf, int64_t count, bool on_device) override {
        const Api &a = api();
        if (count <= 0) return true;
        if (!comm_) return fail("the RCCL communicator was shut down (rccl_shutdown)");
        if (on_device) {
            const int rc = a.all_reduce(buf, buf, (size_t)count, NCCL_INT32, NCCL_SUM, comm_, stream_);
            ++dev_calls_;
            if (rc != 0) return fail(nccl_message("ncclAllReduce", rc));
            return true;  // stream-ordered: the kernels that consume `buf` are queued behind it on the same stream
        }
        const size_t bytes = (size_t)count * 4;
        if (bytes > stage_bytes_) {
            if (stage_) (void)hipFree(stage_);
            stage_ = nullptr;
            stage_bytes_ = 0;
            if (hipMalloc(&stage_, bytes + bytes / 2 + 4096) != hipSuccess) return fail("hipMalloc of the staging buffer failed");
            stage_bytes_ = bytes + bytes / 2 + 4096;
        }
        if (hipMemcpyAsync(stage_, buf, bytes, hipMemcpyHostToDevice, stream_) != hipSuccess) return fail("staging copy failed");
        const int rc = a.all_reduce(stage_, stage_, (size_t)count, NCCL_INT32, NCCL_SUM, comm_, stream_);
        ++host_calls_;
        if (rc != 0) return fail(nccl_message("ncclAllReduce", rc));
        if (hipMemcpyAsync(buf, stage_, bytes, hipMemcpyDeviceToHost, stream_) != hipSuccess || hipStreamSynchronize(stream_) != hipSuccess)
            return fail("staging copy back failed");
        return true;
    }
    const char *last_error() const override { return err_.c_str(); }
    long long device_calls() const override { return dev_calls_; }
    long long host_calls() const override { return host_calls_; }

  private:
    bool fail(std::string m) {
        err_ = std::move(m);
        return false;
    }
    Comm comm_;
    hipStream_t stream_;
    void *stage_ = nullptr;
    size_t stage_bytes_ = 0;
    long long dev_calls_ = 0, host_calls_ = 0;
    std::string err_;
};

}  // namespace

void rccl_unique_id(void *out128) {
    UniqueId id;
    std::memset(&id, 0, sizeof id);
    const int rc = api().get_unique_id(&id);
    if (rc != 0) throw std::runtime_error(nccl_message("ncclGetUniqueId", rc));
    std::memcpy(out128, id.internal, ID_BYTES);
}

namespace {
// one communicator per (unique id, rank, device): creating one costs tens of milliseconds and a rendezvous of all ranks.  Callers
// re-use ONE id per process group (da4ml_amd.multi_gpu caches the broadcast id), so the cache holds one entry per group; without an
// explicit rccl_shutdown() the entries are leaked on purpose: at process exit the HIP runtime and RCCL are torn down in an order this
// library does not control, and a communicator or staging buffer released then is released twice.
std::mutex &cache_mutex() {
    static std::mutex &mu = *new std::mutex();
    return mu;
}
std::map<std::string, std::shared_ptr<RcclTransport>> &comm_cache() {
    static std::map<std::string, std::shared_ptr<RcclTransport>> &cache = *new std::map<std::string, std::shared_ptr<RcclTransport>>();
    return cache;
}
}  // namespace

int rccl_shutdown() {
    std::lock_guard<std::mutex> lk(cache_mutex());
    auto &cache = comm_cache();
    const int n = (int)cache.size();
    for (auto &kv : cache) static_cast<Transport *>(kv.second.get())->destroy();
    cache.clear();
    return n;
}

std::shared_ptr<RcclTransport> RcclTransport::open(const void *id128, int rank, int world, int device, void *stream) {
    std::mutex &mu = cache_mutex();
    auto &cache = comm_cache();
    std::string key(static_cast<const char *>(id128), ID_BYTES);
    key += ":" + std::to_string(rank) + ":" + std::to_string(world) + ":" + std::to_string(device) + ":" + std::to_string(reinterpret_cast<uintptr_t>(stream));
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    if (hipSetDevice(device) != hipSuccess) throw std::runtime_error("hipSetDevice failed");
    UniqueId id;
    std::memcpy(id.internal, id128, ID_BYTES);
    Comm comm = nullptr;
    const int rc = api().comm_init_rank(&comm, world, id, rank);
    if (rc != 0) throw std::runtime_error(nccl_message("ncclCommInitRank", rc));
    auto t = std::shared_ptr<RcclTransport>(new Transport(comm, static_cast<hipStream_t>(stream)));
    cache.emplace(std::move(key), t);
    return t;
}

}  // namespace gpu
}  // namespace da
