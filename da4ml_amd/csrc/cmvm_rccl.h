// cmvm_rccl.h -- RCCL transport of the column-sharded chain (cmvm_shard.h): the all-reduce(sum, int32) the chain needs, issued
// by the library itself on the backend's HIP stream -- `ncclAllReduce`, in place, stream-ordered with the kernels that produce
// and consume the buffers: no Python callback, no host synchronisation per exchange for device buffers.  librccl.so is opened
// at run time (dlopen), so libda4ml_hip.so has no link-time dependency on it and loads where RCCL is absent.
// (The other transport is the caller-supplied callback of da_solve_sharded: torch.distributed, any backend -- gloo in the CPU
// tests.)
#pragma once

#include <cstdint>
#include <memory>

#include "cmvm_shard.h"

namespace da {
namespace gpu {

class RcclTransport {
  public:
    // communicator of `world` ranks for the 128-byte unique id every rank received from rank 0 (rccl_unique_id); collectives
    // run on `stream` of `device`.  Throws std::runtime_error when librccl.so cannot be loaded or the communicator cannot be made.
    static std::shared_ptr<RcclTransport> open(const void *id128, int rank, int world, int device, void *stream);
    virtual ~RcclTransport() = default;
    // all-reduce(sum) of `count` int32 at `buf`, in place.  Device buffers: enqueued on the stream, returns at once (the
    // producers and consumers run on the same stream).  Host buffers: staged through device memory, returns when the result is
    // back.  Returns false after an RCCL / HIP error (message in last_error()).
    virtual bool allreduce(void *buf, int64_t count, bool on_device) = 0;
    virtual const char *last_error() const = 0;
    virtual long long device_calls() const = 0;
    virtual long long host_calls() const = 0;
};

// 128 bytes for rank 0 to hand to every rank (ncclGetUniqueId)
void rccl_unique_id(void *out128);
// destroys every cached communicator (ncclCommDestroy) and frees its staging buffer; returns how many there were.  Collective in
// effect: every rank calls it, while the process group is still alive and no solve is running.
int rccl_shutdown();

}  // namespace gpu
}  // namespace da
