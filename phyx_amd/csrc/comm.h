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
    // The agreement of a step whose collective's size depends on the step: one 16-byte all-reduce (max) of {status, bytes, -bytes}.
    // agree_post only QUEUES it on `stream` (no host wait, no host copy: the words are kernel arguments); agree_read waits for it
    // (bounded) and reads the result.  A rank that failed posts {1, 0, -(2^31 - 1)}: neutral for the sizes, so that it learns its
    // healthy peers' size from the result.  agree = post + read.
    int agree_post(int status, long long bytes, hipStream_t stream);
    int agree_read(int* worst_status, long long* min_bytes, long long* max_bytes, hipStream_t stream);
    int agree(int status, long long bytes, int* worst_status, long long* min_bytes, long long* max_bytes, hipStream_t stream);
    int wait_stream(hipStream_t stream, const char* what);      // hipStreamSynchronize with the communicator's time bound
    static int version();                             // ncclGetVersion (0 if RCCL is not available)
    static double timeout_s();                        // PHX_COMM_TIMEOUT_S: the bound of every host wait on a stream that carries collectives
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
