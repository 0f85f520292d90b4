// comm.cpp — see comm.hpp.  Compiled by hipcc (RcclComm stages its payload in device memory).
#include "comm.hpp"

#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <thread>

namespace ldpc_amd
{

void Comm::all_gather(const void *send, void *recv, size_t bytes)
{
    const auto t0 = std::chrono::steady_clock::now();
    exchange(send, recv, bytes);
    const float us = std::chrono::duration<float, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (exchange_us_.size() < (1u << 20))
        exchange_us_.push_back(us);
}

void Comm::exchange_stats(double out[4], bool reset)
{
    std::vector<float> v = exchange_us_;
    std::sort(v.begin(), v.end());
    out[0] = static_cast<double>(v.size());
    out[1] = v.empty() ? 0.0 : v.front();
    out[2] = v.empty() ? 0.0 : v[v.size() / 2];
    out[3] = v.empty() ? 0.0 : v.back();
    if (reset)
        exchange_us_.clear();
}

double Comm::timeout_seconds(double fallback)
{
    if (const char *e = std::getenv("LDPC_AMD_COMM_TIMEOUT_S"))
    {
        const double v = std::strtod(e, nullptr);
        if (v > 0)
            return v;
    }
    return fallback;
}

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
    int (*CommAbort)(comm_t) = nullptr;
    int (*GetVersion)(int *) = nullptr;
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
            r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(h, "ncclCommAbort"));
            r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(dlsym(h, "ncclGetVersion"));
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
    // ncclCommInitRank is a collective: it returns when every rank of the world has called it.  It runs on a thread of its
    // own so that a rank whose peers never arrive (a process that died before the rendezvous) gives up after the deadline
    // instead of blocking for ever: the constructor throws, the caller exits non-zero, the launcher ends the job.  (The
    // thread itself cannot be cancelled; it is left behind in a process that is about to end.)
    RcclComm(int rank, int world, int device, const uint8_t *id) : device_(device)
    {
        rank_ = rank, world_ = world;
        Rccl &r = Rccl::get();
        hip_check(hipSetDevice(device_), "hipSetDevice");
        struct Init
        {
            std::mutex m;
            std::condition_variable cv;
            bool done = false;
            int rc = 0;
            Rccl::comm_t comm = nullptr;
        };
        auto st = std::make_shared<Init>();
        Rccl::UniqueId uid;
        std::memcpy(uid.internal, id, kCommIdBytes);
        std::thread([st, uid, world, rank, device, &r] {
            Rccl::comm_t c = nullptr;
            int rc = static_cast<int>(hipSetDevice(device));
            if (rc == 0)
                rc = r.CommInitRank(&c, world, uid, rank);
            std::lock_guard<std::mutex> g(st->m);
            st->rc = rc, st->comm = c, st->done = true;
            st->cv.notify_all();
        }).detach();
        {
            std::unique_lock<std::mutex> lk(st->m);
            const double limit = timeout_seconds(60.0);
            if (!st->cv.wait_for(lk, std::chrono::duration<double>(limit), [&] { return st->done; }))
                throw std::runtime_error("RCCL: ncclCommInitRank did not return within " + std::to_string(static_cast<int>(limit)) +
                                         " s (rank " + std::to_string(rank) + " of " + std::to_string(world) +
                                         "): a peer never reached the rendezvous (LDPC_AMD_COMM_TIMEOUT_S)");
            r.check(st->rc, "ncclCommInitRank");
            comm_ = st->comm;
        }
        hip_check(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking), "hipStreamCreate");
        hip_check(hipMalloc(&dev_, kMaxBytes * static_cast<size_t>(world + 1)), "hipMalloc");
    }
    ~RcclComm() override
    {
        if (comm_) // (a communicator whose collective timed out is aborted: ncclCommDestroy would wait for it)
            (void)((broken_ && Rccl::get().CommAbort) ? Rccl::get().CommAbort(comm_) : Rccl::get().CommDestroy(comm_));
        if (dev_ && !broken_)
            (void)hipFree(dev_);
        if (stream_ && !broken_)
            (void)hipStreamDestroy(stream_);
    }
    std::string describe() const override
    {
        int v = 0;
        if (Rccl::get().GetVersion && Rccl::get().GetVersion(&v) == 0 && v > 0)
        {
            // NCCL_VERSION_CODE: major * 10000 + minor * 100 + patch from 2.9 on (major * 1000 + ... before)
            const int major = v >= 20000 ? v / 10000 : v / 1000, minor = v >= 20000 ? (v / 100) % 100 : (v / 100) % 10, patch = v % 100;
            return "rccl " + std::to_string(major) + "." + std::to_string(minor) + "." + std::to_string(patch);
        }
        return "rccl";
    }
    const char *transport() const override { return "rccl"; }

  protected:
    void exchange(const void *send, void *recv, size_t bytes) override
    {
        if (bytes == 0 || bytes > kMaxBytes || bytes % 8)
            throw std::runtime_error("RcclComm::all_gather: payload must be a multiple of 8 bytes, at most 256");
        if (broken_)
            throw std::runtime_error("RcclComm::all_gather: the communicator is unusable after a collective that timed out");
        Rccl &r = Rccl::get();
        hip_check(hipSetDevice(device_), "hipSetDevice");
        char *d = static_cast<char *>(dev_);
        hip_check(hipMemcpyAsync(d, send, bytes, hipMemcpyHostToDevice, stream_), "copy in");
        // the one collective of the path: world x bytes over xGMI, latency-bound
        r.check(r.AllGather(d, d + kMaxBytes, bytes / 8, /*ncclUint64*/ 5, comm_, stream_), "ncclAllGather");
        hip_check(hipMemcpyAsync(recv, d + kMaxBytes, bytes * static_cast<size_t>(world_), hipMemcpyDeviceToHost, stream_), "copy out");
        // bounded wait: a peer that died never joins the collective, and hipStreamSynchronize would wait for it for ever
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(timeout_seconds(60.0));
        unsigned spins = 0;
        for (;;)
        {
            const hipError_t q = hipStreamQuery(stream_);
            if (q == hipSuccess)
                return;
            if (q != hipErrorNotReady)
                hip_check(q, "all-gather stream");
            if (++spins > 4000) // (the exchange normally completes within tens of microseconds: spin first, then yield)
            {
                if (std::chrono::steady_clock::now() > deadline)
                {
                    broken_ = true;
                    throw std::runtime_error("RCCL: the all-gather did not complete within the deadline (rank " + std::to_string(rank_) + " of " +
                                             std::to_string(world_) + "): a peer left the job (LDPC_AMD_COMM_TIMEOUT_S)");
                }
                std::this_thread::yield();
            }
        }
    }

  private:
    int device_;
    Rccl::comm_t comm_ = nullptr;
    hipStream_t stream_ = nullptr;
    void *dev_ = nullptr;
    bool broken_ = false;
};

// ---------------------------------------------------------------------------------------------------------------
// host shared memory: per rank a sequence word and two payload slots (calls alternate between them)
// ---------------------------------------------------------------------------------------------------------------
struct ShmSeg
{
    static constexpr int kMaxRanks = 64;
    static constexpr uint64_t kMagic = 0x6c6470635f616d64ull; // "ldpc_amd"
    static constexpr uint32_t kSealed = 0x80000000u;          // in `attached`: rank 0 found every rank attached
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
            // everyone is in before the name goes away.  The count is SEALED in the same atomic step that finds it complete:
            // a rank whose patience runs out at that very moment (it lets go of objects that never get the go-ahead, below)
            // either takes its attachment back before the seal — then the count is not complete and rank 0 keeps waiting for
            // it to come back — or finds the seal and stays.
            for (;;)
            {
                uint32_t expect = static_cast<uint32_t>(world);
                if (seg_->attached.compare_exchange_strong(expect, static_cast<uint32_t>(world) | ShmSeg::kSealed))
                    break;
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
                // take the attachment back — unless rank 0 has sealed the count in between: then this IS the job's object
                // and the go-ahead follows at once
                uint32_t cur = seg->attached.load();
                bool sealed = false;
                while (!(sealed = (cur & ShmSeg::kSealed) != 0) && !seg->attached.compare_exchange_weak(cur, cur - 1))
                {
                }
                if (sealed)
                {
                    while (seg->go.load(std::memory_order_acquire) == 0 && std::chrono::steady_clock::now() < deadline)
                        std::this_thread::sleep_for(std::chrono::milliseconds(1));
                    if (seg->go.load(std::memory_order_acquire) != 0)
                    {
                        seg_ = seg;
                        return;
                    }
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
    const char *transport() const override { return "shm"; }

  protected:
    void exchange(const void *send, void *recv, size_t bytes) override
    {
        if (bytes == 0 || bytes > kMaxBytes)
            throw std::runtime_error("ShmComm::all_gather: payload too large");
        const uint64_t k = calls_++;
        std::memcpy(seg_->data[k & 1][rank_], send, bytes);
        seg_->seq[rank_].store(k + 1, std::memory_order_release);
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(timeout_seconds(600.0));
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
    const char *transport() const override { return "echo"; }

  protected:
    void exchange(const void *send, void *recv, size_t bytes) override
    {
        if (bytes == 0 || bytes > kMaxBytes)
            throw std::runtime_error("EchoComm::all_gather: payload too large");
        for (int q = 0; q < world_; ++q)
            std::memcpy(static_cast<char *>(recv) + static_cast<size_t>(q) * bytes, send, bytes);
    }
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
