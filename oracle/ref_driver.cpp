// ref_driver.cpp — golden-vector dumper that drives the UNMODIFIED reference classes.
//
// TEST INFRASTRUCTURE ONLY.  This file is our own code; it is compiled by oracle/Makefile
// together with the reference sources where they lie under /root/reference (nothing is
// copied) into oracle/_ref/ref_dump.  It exists to pin oracle/ldpc_oracle.c: the reference
// has no decoder vectors of its own, so tests/golden/make_golden.py runs this
// dumper and commits its output as fixtures.
//
// It performs exactly the per-frame call sequence of ldpc_sim::start (ldpcsim.cpp:158-188)
// on one channel object and records what the reference computed.
//
// usage: ref_dump H G|- AWGN|BSC|BEC BP|BP_MS iters early seed x skip count out.bin
// record per decoded frame (little endian):
//   i32 iters, i32 bit_errors, u8 hard[nc], f64 llr_in[nc], f64 llr_out[nc], u8 codeword[nc]
// header: i32 nc, i32 count

#include <algorithm>
#include <chrono>
#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <forward_list>
#include <fstream>
#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <random>
#include <sstream>
#include <string>
#include <unordered_map>
#include <variant>
#include <vector>

// the dumper needs the decoder's input LLR vector, which the reference keeps protected
#define protected public
#define private public
#include "sim/ldpcsim.h"
#undef protected
#undef private

using namespace ldpc;

int main(int argc, char **argv)
{
    if (argc != 12)
    {
        std::fprintf(stderr, "usage: %s H G|- CH DEC iters early seed x skip count out.bin\n", argv[0]);
        return 2;
    }
    std::string hfile = argv[1], gfile = argv[2], ch = argv[3], dec = argv[4];
    if (gfile == "-")
        gfile = "";
    decoder_param dp;
    dp.iterations = static_cast<u32>(std::strtoul(argv[5], nullptr, 10));
    dp.earlyTerm = std::atoi(argv[6]) != 0;
    dp.type = dec.c_str();
    u64 seed = std::strtoull(argv[7], nullptr, 10);
    double x = std::atof(argv[8]);
    u64 skip = std::strtoull(argv[9], nullptr, 10), count = std::strtoull(argv[10], nullptr, 10);

    auto code = std::make_shared<ldpc_code>(hfile, gfile);
    const int nc = code->nc();

    std::shared_ptr<channel> chan;
    if (ch == "AWGN")
        chan = std::make_shared<channel_awgn>(code, dp, seed, 1.);
    else if (ch == "BSC")
        chan = std::make_shared<channel_bsc>(code, dp, seed, 0.);
    else
        chan = std::make_shared<channel_bec>(code, dp, seed, 0.);
    chan->set_channel_param(x);

    FILE *fp = std::fopen(argv[11], "wb");
    if (!fp)
        return 3;
    int hdr[2] = {nc, static_cast<int>(count)};
    std::fwrite(hdr, 4, 2, fp);

    std::vector<double> in(nc), out(nc);
    std::vector<unsigned char> hard(nc), cw(nc);
    for (u64 f = 0; f < skip + count; ++f)
    {
        if (!code->G().empty())
            chan->encode_and_map();
        chan->simulate();
        if (f < skip)
            continue;
        chan->calculate_llrs();
        if (ch == "BEC")
        {
            auto *c = static_cast<channel_bec *>(chan.get());
            for (int i = 0; i < nc; ++i)
                in[i] = c->mLdpcDecoder->mLLRIn[i];
        }
        else
        {
            for (int i = 0; i < nc; ++i)
                in[i] = chan->mLdpcDecoder->mLLRIn[i];
        }
        int it = chan->decode();
        int be = 0;
        for (auto ci : code->bit_pos())
            be += (chan->estimate()[ci] != chan->codeword()[ci]);
        for (int i = 0; i < nc; ++i)
        {
            hard[i] = chan->estimate()[i].value;
            cw[i] = chan->codeword()[i].value;
        }
        if (ch == "BEC")
        {
            auto *c = static_cast<channel_bec *>(chan.get());
            for (int i = 0; i < nc; ++i)
                out[i] = c->mLdpcDecoder->llr_out()[i];
        }
        else
        {
            for (int i = 0; i < nc; ++i)
                out[i] = chan->mLdpcDecoder->llr_out()[i];
        }
        std::fwrite(&it, 4, 1, fp);
        std::fwrite(&be, 4, 1, fp);
        std::fwrite(hard.data(), 1, nc, fp);
        std::fwrite(in.data(), 8, nc, fp);
        std::fwrite(out.data(), 8, nc, fp);
        std::fwrite(cw.data(), 1, nc, fp);
    }
    std::fclose(fp);
    return 0;
}
