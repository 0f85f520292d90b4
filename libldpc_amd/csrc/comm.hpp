// comm.hpp — the one exchange step of the multi-GPU path: small all-gathers between the ranks of a
// sharded simulation (one process per GPU).  Two transports behind one interface:
//
//   RcclComm  RCCL (ncclAllGather over xGMI) on device buffers, librccl.so loaded on first use.  The communicator is
//             built from an ncclUniqueId the launcher distributes (torch.distributed in bench.py, pipes in
//             `ldpcsim --devices`), exactly like ncclCommInitRank.
//   ShmComm   a POSIX shared-memory segment on the host: rehearsals in which several ranks share one GPU (RCCL
//             refuses two ranks on one device) and tests without a GPU.
//   EchoComm  no transport at all: one process plays one rank of a world of any size (cost probes).
//
// The reference has no communication at all (OpenMP threads share counters, ldpcsim.cpp:175-252); what crosses
// ranks here is what those shared counters carried, plus the accepted-pair counts that place every rank in the
// one noise stream.
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace ldpc_amd
{

constexpr size_t kCommIdBytes = 128; // NCCL_UNIQUE_ID_BYTES

class Comm
{
  public:
    virtual ~Comm() = default;
    int rank() const { return rank_; }
    int world() const { return world_; }
    // recv[q*bytes .. (q+1)*bytes) = rank q's send[0..bytes), host buffers, blocking; bytes <= kMaxBytes.  Never waits
    // without a bound: a rank that does not arrive within the deadline (LDPC_AMD_COMM_TIMEOUT_S, default 60 s; shared
    // memory: 600 s) makes the call throw on every rank that waits for it.  The host time of every call is recorded.
    void all_gather(const void *send, void *recv, size_t bytes);
    virtual const char *transport() const = 0;
    // "rccl 2.22.3", "shm", "echo"
    virtual std::string describe() const { return transport(); }
    static constexpr size_t kMaxBytes = 256;
    // host microseconds inside all_gather since the last reset: {calls, min, median, max}
    void exchange_stats(double out[4], bool reset);
    static double timeout_seconds(double fallback);

    // sum of n int64 values over all ranks (an all-gather and a local sum: the payloads are a few words)
    void all_reduce_sum(int64_t *values, size_t n);

  protected:
    virtual void exchange(const void *send, void *recv, size_t bytes) = 0;
    int rank_ = 0, world_ = 1;

  private:
    std::vector<float> exchange_us_;
};

// fills id[kCommIdBytes] with a fresh ncclUniqueId (rank 0 calls this, every rank gets the bytes)
void rccl_unique_id(uint8_t *id);
std::unique_ptr<Comm> make_rccl_comm(int rank, int world, int device, const uint8_t *id);
// name: shared-memory object name, the same on every rank and unique to the job (e.g. "/ldpc_amd_<port>")
std::unique_ptr<Comm> make_shm_comm(int rank, int world, const std::string &name);
// ONE process standing in for rank `rank` of `world`: all_gather answers every rank's slot with this rank's own payload
// (every piece of a sharded step then holds as many pairs as this one).  For measuring what a rank of a world-sized job
// costs on a single GPU (tools/shard_probe.py); results are frames of valid noise, not the frames of the real stream.
std::unique_ptr<Comm> make_echo_comm(int rank, int world);

} // namespace ldpc_amd
