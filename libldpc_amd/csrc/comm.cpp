// comm.cpp — see comm.hpp.  Compiled by hipcc (RcclComm stages its payload in device memory).
#include "comm.hpp"

#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstring>
#include <stdexcept>
#include <thread>

namespace ldpc_amd
{

void Comm::all_reduce_sum(int64_t *values, size_t n)
{
    if (world_ == 1)
        return;
    if (n * sizeof(int64_t) > kMaxBytes)
        throw std::runtime_error("Comm::all_reduce_sum: payload too large");
    std::vector<int64_t> all(n * static_cast<size_t>(world_));
    all_gather(values, all.data(), n * sizeof(int64_t));
    for (size_t i = 0; i < n; ++i)
    {
        int64_t s = 0;
        for (int q = 0; q < world_; ++q)
            s += all[static_cast<size_t>(q) * n + i];
        values[i] = s;
    }
}

namespace
{
// ---------------------------------------------------------------------------------------------------------------
// RCCL, bound at run time: the single-GPU product path never loads librccl.so
// ---------------------------------------------------------------------------------------------------------------
struct Rccl
{
    using comm_t = void *;
    struct UniqueId
    {
        char internal[kCommIdBytes];
    };
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(comm_t *, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, comm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;

    static Rccl &get()
    {
        static Rccl r;
        if (!r.GetUniqueId)
        {
            void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!h)
                h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (!h)
                throw std::runtime_error(std::string("cannot load librccl.so: ") + dlerror());
            auto sym = [&](const char *n) {
                void *p = dlsym(h, n);
                if (!p)
                    throw std::runtime_error(std::string("librccl.so lacks ") + n);
                return p;
            };
            r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
            r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
            r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
            r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
            r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        }
        return r;
    }
    void check(int rc, const char *what) const
    {
        if (rc != 0)
            throw std::runtime_error(std::string("RCCL error in ") + what + ": " + (GetErrorString ? GetErrorString(rc) : "?"));
    }
};

void hip_check(hipError_t e, const char *what)
{
    if (e != hipSuccess)
        throw std::runtime_error(std::string("HIP error in ") + what + ": " + hipGetErrorString(e));
}

class RcclComm final : public Comm
{
  public:
    RcclComm(int rank, int world, int device, const uint8_t *id) : device_(device)
    {
        rank_ = rank, world_ = world;
        Rccl &r = Rccl::get();
        hip_check(hipSetDevice(device_), "hipSetDevice");
        Rccl::UniqueId uid;
        std::memcpy(uid.internal, id, kCommIdBytes);
        r.check(r.CommInitRank(&comm_, world, uid, rank), "ncclCommInitRank");
        hip_check(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking), "hipStreamCreate");
        hip_check(hipMalloc(&dev_, kMaxBytes * static_cast<size_t>(world + 1)), "hipMalloc");
    }
    ~RcclComm() override
    {
        if (comm_)
            (void)Rccl::get().CommDestroy(comm_);
        if (dev_)
            (void)hipFree(dev_);
        if (stream_)
            (void)hipStreamDestroy(stream_);
    }
    void all_gather(const void *send, void *recv, size_t bytes) override
    {
        if (bytes == 0 || bytes > kMaxBytes || bytes % 8)
            throw std::runtime_error("RcclComm::all_gather: payload must be a multiple of 8 bytes, at most 256");
        Rccl &r = Rccl::get();
        hip_check(hipSetDevice(device_), "hipSetDevice");
        char *d = static_cast<char *>(dev_);
        hip_check(hipMemcpyAsync(d, send, bytes, hipMemcpyHostToDevice, stream_), "copy in");
        // the one collective of the path: world x bytes over xGMI, latency-bound
        r.check(r.AllGather(d, d + kMaxBytes, bytes / 8, /*ncclUint64*/ 5, comm_, stream_), "ncclAllGather");
        hip_check(hipMemcpyAsync(recv, d + kMaxBytes, bytes * static_cast<size_t>(world_), hipMemcpyDeviceToHost, stream_), "copy out");
        hip_check(hipStreamSynchronize(stream_), "sync");
    }
    const char *transport() const override { return "rccl"; }

  private:
    int device_;
    Rccl::comm_t comm_ = nullptr;
    hipStream_t stream_ = nullptr;
    void *dev_ = nullptr;
};

// ---------------------------------------------------------------------------------------------------------------
// host shared memory: per rank a sequence word and two payload slots (calls alternate between them)
// ---------------------------------------------------------------------------------------------------------------
struct ShmSeg
{
    static constexpr int kMaxRanks = 64;
    static constexpr uint64_t kMagic = 0x6c6470635f616d64ull; // "ldpc_amd"
    std::atomic<uint64_t> seq[kMaxRanks];
    std::atomic<uint64_t> magic; // set by rank 0 once the object is sized and zero-filled
    std::atomic<uint32_t> attached;
    std::atomic<uint32_t> go;    // set by rank 0 when every rank has attached to THIS object (then the name is unlinked)
    char data[2][kMaxRanks][Comm::kMaxBytes];
};

class ShmComm final : public Comm
{
  public:
    ShmComm(int rank, int world, const std::string &name) : name_(name)
    {
        if (world < 1 || world > ShmSeg::kMaxRanks || rank < 0 || rank >= world)
            throw std::runtime_error("ShmComm: bad rank / world size");
        rank_ = rank, world_ = world;
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(120);
        auto map = [&](int fd) {
            void *p = mmap(nullptr, sizeof(ShmSeg), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            close(fd);
            if (p == MAP_FAILED)
                throw std::runtime_error("ShmComm: mmap failed");
            return static_cast<ShmSeg *>(p);
        };
        if (rank == 0)
        {
            (void)shm_unlink(name.c_str()); // an object of this name left by an aborted job (names get reused)
            const int fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
            if (fd < 0 || ftruncate(fd, sizeof(ShmSeg)) != 0)
                throw std::runtime_error("ShmComm: cannot create " + name);
            seg_ = map(fd); // a fresh object is zero-filled: every seq starts at 0
            seg_->magic.store(ShmSeg::kMagic, std::memory_order_release);
            seg_->attached.fetch_add(1);
            while (seg_->attached.load() < static_cast<uint32_t>(world)) // everyone is in before the name goes away
            {
                if (std::chrono::steady_clock::now() > deadline)
                {
                    (void)shm_unlink(name.c_str());
                    throw std::runtime_error("ShmComm: timed out waiting for the other ranks");
                }
                std::this_thread::sleep_for(std::chrono::milliseconds(1));
            }
            seg_->go.store(1, std::memory_order_release);
            (void)shm_unlink(name.c_str());
            return;
        }
        // A rank may find an object of the same name that an ABORTED earlier job left behind (its rank 0 never unlinked it)
        // before this job's rank 0 has replaced it.  Such an object never gets the go-ahead: after a short wait the rank lets
        // go of it and opens the name again, until it sits on the object this job's rank 0 created.
        for (;;)
        {
            if (std::chrono::steady_clock::now() > deadline)
                throw std::runtime_error("ShmComm: timed out waiting for " + name);
            const int fd = shm_open(name.c_str(), O_RDWR, 0600);
            struct stat st;
            if (fd < 0 || fstat(fd, &st) != 0 || st.st_size < static_cast<off_t>(sizeof(ShmSeg)))
            {
                if (fd >= 0)
                    close(fd);
                std::this_thread::sleep_for(std::chrono::milliseconds(2));
                continue;
            }
            ShmSeg *seg = map(fd);
            bool ok = false;
            if (seg->magic.load(std::memory_order_acquire) == ShmSeg::kMagic)
            {
                seg->attached.fetch_add(1);
                const auto patience = std::chrono::steady_clock::now() + std::chrono::seconds(3);
                while (!(ok = seg->go.load(std::memory_order_acquire) != 0) && std::chrono::steady_clock::now() < patience &&
                       std::chrono::steady_clock::now() < deadline)
                    std::this_thread::sleep_for(std::chrono::milliseconds(1));
            }
            if (ok)
            {
                seg_ = seg;
                return;
            }
            if (seg->magic.load(std::memory_order_acquire) == ShmSeg::kMagic)
            {
                // take the attachment back — unless the go-ahead arrived in between (rank 0 counts attachments)
                uint32_t cur = seg->attached.load();
                while (seg->go.load(std::memory_order_acquire) == 0 && !seg->attached.compare_exchange_weak(cur, cur - 1))
                {
                }
                if (seg->go.load(std::memory_order_acquire) != 0)
                {
                    seg_ = seg;
                    return;
                }
            }
            munmap(seg, sizeof(ShmSeg));
            std::this_thread::sleep_for(std::chrono::milliseconds(5));
        }
    }
    ~ShmComm() override
    {
        if (seg_)
            munmap(seg_, sizeof(ShmSeg));
    }
    void all_gather(const void *send, void *recv, size_t bytes) override
    {
        if (bytes == 0 || bytes > kMaxBytes)
            throw std::runtime_error("ShmComm::all_gather: payload too large");
        const uint64_t k = calls_++;
        std::memcpy(seg_->data[k & 1][rank_], send, bytes);
        seg_->seq[rank_].store(k + 1, std::memory_order_release);
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(600);
        for (int q = 0; q < world_; ++q)
        {
            unsigned spins = 0;
            while (seg_->seq[q].load(std::memory_order_acquire) < k + 1)
                if (++spins > 2000)
                {
                    if (std::chrono::steady_clock::now() > deadline)
                        throw std::runtime_error("ShmComm::all_gather: a rank did not arrive");
                    std::this_thread::yield();
                }
            std::memcpy(static_cast<char *>(recv) + static_cast<size_t>(q) * bytes, seg_->data[k & 1][q], bytes);
        }
        // slot k&1 is rewritten in call k+2, which a rank enters only after every rank has published call k+1,
        // i.e. after every rank has finished reading call k
    }
    const char *transport() const override { return "shm"; }

  private:
    std::string name_;
    ShmSeg *seg_ = nullptr;
    uint64_t calls_ = 0;
};
class EchoComm final : public Comm
{
  public:
    EchoComm(int rank, int world)
    {
        if (world < 1 || rank < 0 || rank >= world)
            throw std::runtime_error("EchoComm: bad rank / world size");
        rank_ = rank, world_ = world;
    }
    void all_gather(const void *send, void *recv, size_t bytes) override
    {
        if (bytes == 0 || bytes > kMaxBytes)
            throw std::runtime_error("EchoComm::all_gather: payload too large");
        for (int q = 0; q < world_; ++q)
            std::memcpy(static_cast<char *>(recv) + static_cast<size_t>(q) * bytes, send, bytes);
    }
    const char *transport() const override { return "echo"; }
};
} // namespace

std::unique_ptr<Comm> make_echo_comm(int rank, int world) { return std::make_unique<EchoComm>(rank, world); }

void rccl_unique_id(uint8_t *id)
{
    Rccl &r = Rccl::get();
    Rccl::UniqueId uid;
    r.check(r.GetUniqueId(&uid), "ncclGetUniqueId");
    std::memcpy(id, uid.internal, kCommIdBytes);
}

std::unique_ptr<Comm> make_rccl_comm(int rank, int world, int device, const uint8_t *id)
{
    return std::make_unique<RcclComm>(rank, world, device, id);
}

std::unique_ptr<Comm> make_shm_comm(int rank, int world, const std::string &name)
{
    return std::make_unique<ShmComm>(rank, world, name);
}

} // namespace ldpc_amd
