// (host only) where the time of frisk_fasta_load goes before the upload: the native reader's parse and the 2-bit packer on a FASTA file.
// g++ -std=c++17 -O3 -Ifrisk_amd/csrc tools/exp/load_split.cpp -o /tmp/load_split -lpthread -lz && /tmp/load_split <fasta> [threads]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include "fasta_reader.h"
#include "seq_pack2.h"
int main(int argc, char** argv) {
    const int threads = argc > 2 ? std::atoi(argv[2]) : 32;
    for (int round = 0; round < 3; ++round) {
        frisk_fasta::Records rec;
        std::string err;
        auto t0 = std::chrono::steady_clock::now();
        if (!frisk_fasta::parse(argv[1], rec, err, threads)) { std::printf("parse failed: %s\n", err.c_str()); return 1; }
        auto t1 = std::chrono::steady_clock::now();
        const int64_t P = frisk_pack2::padded_len(rec.lens.data(), int32_t(rec.lens.size()));
        std::vector<uint32_t, frisk_fasta::NoInitAlloc<uint32_t>> codes;
        codes.resize(size_t(P / 16));
        auto t2 = std::chrono::steady_clock::now();
        frisk_pack2::Runs R;
        frisk_pack2::pack_stage(rec.stage.data(), rec.lens.data(), int32_t(rec.lens.size()), codes.data(), R, threads);
        auto t3 = std::chrono::steady_clock::now();
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        std::printf("threads %d: parse %.1f ms, alloc %.1f ms, pack %.1f ms (%zu records, %ld positions, runs %zu / %zu, avx512 %d)\n", threads, ms(t0, t1),
                    ms(t1, t2), ms(t2, t3), rec.lens.size(), long(P), R.inv.size() / 2, R.low.size() / 2, int(frisk_pack2::have_avx512()));
    }
}
