// comm.h — RCCL communicator of the island-sharded solve (comm.hip).  One per process / GPU.
#pragma once

#include "common.h"

namespace phx {

class Comm {
public:
    explicit Comm(int device) : device_(device) {}
    ~Comm();
    Comm(const Comm&) = delete;
    Comm& operator=(const Comm&) = delete;
    static int unique_id(void* out);                  // PHX_COMM_ID_BYTES bytes (ncclGetUniqueId), to be handed to every rank out of band
    int init(const void* unique_id, int rank, int nranks);
    int all_gather(const void* d_send, void* d_recv, size_t bytes_per_rank, hipStream_t stream);
    int all_reduce_max_int(int* d_word, hipStream_t stream);
    int barrier(hipStream_t stream);
    int barrier_async(hipStream_t stream);
    int async_error(int* out);
    int agree(int status, long long bytes, int* worst_status, long long* min_bytes, long long* max_bytes, hipStream_t stream);
    int wait_stream(hipStream_t stream, const char* what);      // hipStreamSynchronize with the communicator's time bound
    static int version();                             // ncclGetVersion (0 if RCCL is not available)
    int rank() const { return rank_; }
    int size() const { return nranks_; }
    int device() const { return device_; }

private:
    struct Impl;
    Impl* impl_ = nullptr;
    int device_, rank_ = 0, nranks_ = 1;
    int* flag_ = nullptr;
};

} // namespace phx

struct phx_comm {
    phx::Comm impl;
    explicit phx_comm(int device) : impl(device) {}
};
