// ldpcsim — command line front end, drop-in for the reference's `ldpcsim` (src/sim_cpu.cpp:5-85):
//
//   ldpcsim codefile output-file MIN MAX STEP [-G file] [-i 50] [-s 0] [-t 1] [--channel AWGN|BSC|BEC]
//           [--decoding BP|BP_MS] [--max-frames N] [--frame-error-count 50] [--no-early-term]
//
// Same positional arguments, flags, defaults, console table and result file; the frames are decoded on
// the GPU in batches of the reference's single noise stream (seed as given; -t is accepted and ignored:
// the batch replaces the OpenMP threads).  Extra flags: --device N (GPU index); --devices LIST (e.g. 0-7 or 0,2,4:
// one process per listed GPU, forked before anything touches a GPU, the frames of every step shared out over them and
// the counters exchanged over RCCL — results are those of the one-GPU run; --comm shm puts the exchange on host
// shared memory instead, for rehearsals with a repeated device such as --devices 0,0); --bec-compat (reproduce the
// reference's out-of-bounds read for erased degree-1 variable nodes).
#include <fcntl.h>
#include <signal.h>
#include <sys/prctl.h>
#include <sys/wait.h>
#include <unistd.h>

#include <atomic>
#include <cctype>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ldpc_amd.h"

namespace
{
const char *kUsage =
    "Usage: ldpc [options] codefile output-file snr-range \n\n"
    "Positional arguments:\n"
    "codefile            \tLDPC parity-check matrix file containing all non-zero entries.\n"
    "output-file         \tResults output file.\n"
    "snr-range           \t{MIN} {MAX} {STEP}\n\n"
    "Optional arguments:\n"
    "-h --help           \tshows help message and exits\n"
    "-G --gen-matrix     \tGenerator matrix file.\n"
    "-i --num-iterations \tNumber of iterations for decoding. (Default: 50)\n"
    "-s --seed           \tRNG seed. (Default: 0)\n"
    "-t --num-threads    \tNumber of frames to be decoded in parallel. (Default: 1; ignored: GPU batches)\n"
    "--channel           \tSpecifies channel: \"AWGN\", \"BSC\", \"BEC\" (Default: AWGN)\n"
    "--decoding          \tSpecifies decoding algorithm: \"BP\", \"BP_MS\" (Default: BP)\n"
    "--max-frames        \tLimit number of decoded frames.\n"
    "--frame-error-count \tMaximum frame errors for given simulation point.\n"
    "--no-early-term     \tDisable early termination for decoding.\n"
    "--device            \tGPU index. (Default: 0)\n"
    "--devices           \tGPUs to share the frames over, one process each: \"0-7\", \"0,1,2\".\n"
    "--comm              \tExchange between those processes: \"rccl\" (default) or \"shm\".\n"
    "--bec-compat        \tBEC: erased degree-1 variable nodes emit 0 as the reference build does.\n";

std::vector<int> parse_devices(const std::string &spec)
{
    std::vector<int> out;
    size_t i = 0;
    while (i < spec.size())
    {
        size_t j = spec.find(',', i);
        const std::string part = spec.substr(i, j == std::string::npos ? std::string::npos : j - i);
        const size_t dash = part.find('-');
        if (dash == std::string::npos)
            out.push_back(std::stoi(part));
        else
            for (int d = std::stoi(part.substr(0, dash)); d <= std::stoi(part.substr(dash + 1)); ++d)
                out.push_back(d);
        if (j == std::string::npos)
            break;
        i = j + 1;
    }
    if (out.empty() || out.size() > 64)
        throw std::runtime_error("--devices: expected a list such as 0-7 or 0,1,2");
    return out;
}

[[noreturn]] void fail(const std::string &msg)
{
    std::cout << msg << std::endl << kUsage;
    std::exit(EXIT_FAILURE);
}
} // namespace

int main(int argc, char *argv[])
{
    std::vector<std::string> pos;
    std::string gen, channel = "AWGN", decoding = "BP";
    unsigned iterations = 50, threads = 1;
    unsigned long seed = 0, max_frames = static_cast<unsigned long>(10e9), fec = 50;
    bool no_early = false, bec_compat = false;
    int device = 0;
    std::vector<int> devices;
    std::string comm_kind = "rccl";
    try
    {
        for (int i = 1; i < argc; ++i)
        {
            std::string a = argv[i];
            auto value = [&]() -> std::string {
                if (i + 1 >= argc)
                    throw std::runtime_error("missing value for " + a);
                return argv[++i];
            };
            if (a == "-h" || a == "--help")
            {
                std::cout << kUsage;
                return 0;
            }
            else if (a == "-G" || a == "--gen-matrix")
                gen = value();
            else if (a == "-i" || a == "--num-iterations")
                iterations = static_cast<unsigned>(std::stoul(value()));
            else if (a == "-s" || a == "--seed")
                seed = std::stoul(value());
            else if (a == "-t" || a == "--num-threads")
                threads = static_cast<unsigned>(std::stoul(value()));
            else if (a == "--channel")
                channel = value();
            else if (a == "--decoding")
                decoding = value();
            else if (a == "--max-frames")
                max_frames = std::stoul(value());
            else if (a == "--frame-error-count")
                fec = std::stoul(value());
            else if (a == "--no-early-term")
                no_early = true;
            else if (a == "--device")
                device = std::stoi(value());
            else if (a == "--devices")
                devices = parse_devices(value());
            else if (a == "--comm")
                comm_kind = value();
            else if (a == "--bec-compat")
                bec_compat = true;
            else if (a.size() > 1 && a[0] == '-' && !(std::isdigit(static_cast<unsigned char>(a[1])) || a[1] == '.'))
                throw std::runtime_error("Unknown argument: " + a);
            else
                pos.push_back(a); // negative numbers are positional (snr range)
        }
        if (pos.size() != 5)
            throw std::runtime_error("expected: codefile output-file MIN MAX STEP");
        if (comm_kind != "rccl" && comm_kind != "shm")
            throw std::runtime_error("--comm: rccl or shm");
    }
    catch (const std::exception &e)
    {
        fail(e.what());
    }
    double range[3];
    try
    {
        for (int i = 0; i < 3; ++i)
            range[i] = std::stod(pos[2 + i]);
    }
    catch (const std::exception &)
    {
        fail("snr-range must be three numbers");
    }
    if (range[0] > range[1])
        fail("snr min > snr max"); // sim_cpu.cpp:29

    // ---- several GPUs: one process per device, forked before this process makes any HIP call ----
    const int world = devices.empty() ? 1 : static_cast<int>(devices.size());
    int rank = 0;
    std::vector<pid_t> children;
    std::vector<int> id_pipe_w; // parent -> child r: the RCCL unique id
    int id_pipe_r = -1;
    const std::string shm_name = "/ldpc_amd_" + std::to_string(static_cast<long>(getpid()));
    if (world > 1)
    {
        std::fflush(stdout);
        for (int r = 1; r < world; ++r)
        {
            int fds[2];
            if (pipe(fds) != 0)
                fail("pipe() failed");
            const pid_t pid = fork();
            if (pid < 0)
                fail("fork() failed");
            if (pid == 0)
            {
                rank = r;
                close(fds[1]);
                id_pipe_r = fds[0];
                for (int w : id_pipe_w)
                    close(w);
                id_pipe_w.clear();
                children.clear();
                const int devnull = open("/dev/null", O_WRONLY); // rank 0 alone prints the banner, table and result file
                if (devnull >= 0)
                    dup2(devnull, STDOUT_FILENO);
                break;
            }
            close(fds[0]);
            id_pipe_w.push_back(fds[1]);
            children.push_back(pid);
        }
        device = devices[rank];
    }
    // Rank 0 watches its rank processes while it works: a rank that dies (no usable device, a HIP error, a signal) would
    // otherwise leave the others — and this process — waiting in the next collective for ever.  The watcher ends the
    // remaining ranks and the whole run with a non-zero status; the ranks themselves die with their parent.
    static std::atomic<bool> watch_on{false};
    static std::vector<pid_t> watched;
    static std::mutex watch_mu;
    std::thread watcher;
    if (world > 1 && rank == 0)
    {
        watched = children;
        watch_on = true;
        watcher = std::thread([] {
            while (watch_on.load())
            {
                {
                    std::lock_guard<std::mutex> lk(watch_mu);
                    for (pid_t &c : watched)
                    {
                        if (c <= 0)
                            continue;
                        int st = 0;
                        const pid_t r = waitpid(c, &st, WNOHANG);
                        if (r == c)
                        {
                            const bool ok = WIFEXITED(st) && WEXITSTATUS(st) == 0;
                            c = ok ? 0 : -1;
                            if (!ok && watch_on.load())
                            {
                                std::fprintf(stderr, "Error: a rank process ended abnormally (status 0x%x): stopping the run\n", st);
                                for (pid_t o : watched)
                                    if (o > 0)
                                        kill(o, SIGTERM);
                                std::fflush(nullptr);
                                _exit(EXIT_FAILURE);
                            }
                        }
                    }
                }
                std::this_thread::sleep_for(std::chrono::milliseconds(50));
            }
        });
    }
    else if (world > 1)
        prctl(PR_SET_PDEATHSIG, SIGTERM);
    auto reap = [&](int rc) {
        if (watcher.joinable())
        {
            watch_on = false;
            watcher.join();
        }
        std::lock_guard<std::mutex> lk(watch_mu);
        for (size_t i = 0; i < children.size(); ++i)
        {
            const pid_t c = children[i];
            if (i < watched.size() && watched[i] <= 0) // already collected by the watcher
            {
                if (watched[i] < 0)
                    rc = rc ? rc : EXIT_FAILURE;
                continue;
            }
            if (rc != 0)
                kill(c, SIGTERM); // this rank failed: the others would wait for it
            int st = 0;
            if (waitpid(c, &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0)
                rc = rc ? rc : EXIT_FAILURE;
        }
        return rc;
    };

    ldpc_hip_ctx *ctx = ldpc_hip_create(pos[0].c_str(), gen.c_str(), device);
    if (!ctx)
    {
        std::cout << "Error: ldpc_code(): " << ldpc_hip_last_error() << std::endl; // ldpc.cpp:16-20
        return reap(EXIT_FAILURE);
    }
    ldpc_hip_set_bec_compat(ctx, bec_compat);
    int64_t info[10];
    ldpc_hip_code_info(ctx, info);

    const char *bar = "========================================================================================";
    std::cout << bar << std::endl;
    std::cout << "Parity-Check Matrix: " << pos[0] << std::endl;
    std::cout << "Generator Matrix: " << gen << std::endl;
    std::cout << ldpc_hip_describe(ctx) << std::endl;
    std::cout << bar << std::endl;
    std::cout << "== Decoder Parameters\n";
    std::cout << " Type: " << decoding << "\n Iterations: " << iterations << "\n Early Termination: " << !no_early << "\n";
    std::cout << "== Channel Parameters\n";
    std::cout << " Type: " << channel << "\n Seed: " << seed << "\n Range: Min: " << range[0] << ", Max: " << range[1]
              << ", Step: " << range[2] << "\n";
    std::cout << "== Simulation Parameters\n";
    std::cout << " Threads: " << threads << "\n FEC: " << fec << "\n Max Frames: " << max_frames
              << "\n Output File: " << pos[1] << "\n";
    std::cout << std::endl << bar << std::endl;

    decoder_param dp{!no_early, iterations, decoding.c_str()};
    channel_param cp{seed, {range[0], range[1], range[2]}, channel.c_str()};
    simulation_param sp{threads, max_frames, fec, pos[1].c_str()};
    bool stop = false;
    ldpc_hip_comm *comm = nullptr;
    if (world > 1)
    {
        if (comm_kind == "rccl")
        {
            uint8_t id[128] = {0};
            bool ok = true;
            if (rank == 0)
            {
                ok = ldpc_hip_comm_unique_id(id) == 0;
                for (int w : id_pipe_w)
                {
                    ok = ok && write(w, id, sizeof id) == static_cast<ssize_t>(sizeof id);
                    close(w);
                }
            }
            else
                ok = read(id_pipe_r, id, sizeof id) == static_cast<ssize_t>(sizeof id);
            if (ok)
                comm = ldpc_hip_comm_create(rank, world, device, id);
        }
        else
        {
            for (int w : id_pipe_w)
                close(w);
            comm = ldpc_hip_comm_create_shm(rank, world, shm_name.c_str());
        }
        if (!comm)
        {
            std::fprintf(stderr, "Error: rank %d: communicator: %s\n", rank, ldpc_hip_last_error());
            ldpc_hip_destroy(ctx);
            return reap(EXIT_FAILURE);
        }
    }
    int rc = comm ? ldpc_hip_simulate_sharded(ctx, comm, dp, cp, sp, nullptr, nullptr, &stop, /*cli_output=*/1)
                  : ldpc_hip_simulate(ctx, dp, cp, sp, nullptr, nullptr, &stop, /*cli_output=*/1);
    if (rc < 0)
    {
        std::cout << "Error: ldpc_sim::ldpc_sim() " << ldpc_hip_last_error() << std::endl;
        std::fprintf(stderr, "Error: rank %d: %s\n", rank, ldpc_hip_last_error());
        ldpc_hip_destroy(ctx);
        return reap(EXIT_FAILURE);
    }
    if (comm)
        ldpc_hip_comm_destroy(comm);
    ldpc_hip_destroy(ctx);
    return reap(0);
}
