// (CPU) the library's host-only headers under AddressSanitizer + UBSan (GPU sanitizers are not available on the pool): random batches
// through the 2-bit packer (against a letter-by-letter restatement), bitmap -> runs, the HMM fit / Viterbi on random series of awkward
// lengths.  build + run:  g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -Ifrisk_amd/csrc
//                         tools/exp/san_host.cpp -o build/san/san_host -lpthread -lz && build/san/san_host
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include "seq_pack2.h"
#include "hmm_host.h"
#include "fasta_index.h"
#include "fasta_pack2.h"
#include <fstream>
#include <unistd.h>

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s:%d %s\n", __FILE__, __LINE__, #c); ++fails; } } while (0)

int main() {
    std::mt19937_64 rng(12345);
    const char alphabet[] = "ACGTacgtNnRYKM-*xACGTACGTACGT";
    for (int round = 0; round < 300; ++round) {
        const int n_seq = int(rng() % 6);
        std::vector<std::string> seqs;
        std::vector<int64_t> lens;
        for (int s = 0; s < n_seq; ++s) {
            const int64_t n = (round % 10 == 0 && s == 0) ? int64_t(1) << 22 : int64_t(rng() % 3000);      // (one batch in ten takes the threaded path)
            std::string q(size_t(n), 'A');
            for (auto& ch : q) ch = alphabet[rng() % (sizeof(alphabet) - 1)];
            for (int k = 0; k < 4 && n > 64; ++k) {                     // long runs: run merging across words and thread cuts
                const size_t a = rng() % size_t(n - 40), ln = rng() % 40 + (k == 0 ? 3000 % size_t(n - a) : 0);
                for (size_t i = a; i < std::min(size_t(n), a + ln); ++i) q[i] = (k & 1) ? 'N' : 'c';
            }
            seqs.push_back(q); lens.push_back(n);
        }
        std::vector<const uint8_t*> ptr;
        for (auto& q : seqs) ptr.push_back(reinterpret_cast<const uint8_t*>(q.data()));
        const int64_t P = frisk_pack2::padded_len(lens.data(), n_seq);
        std::vector<uint32_t> codes(size_t(P / 16), 0xDEADBEEFu);
        frisk_pack2::Runs R;
        frisk_pack2::pack_batch(ptr.data(), lens.data(), n_seq, codes.data(), R, 1 + int(rng() % 7), (round & 1) != 0);      // (odd rounds: the AVX-512 path where the host has it)
        // letter by letter
        std::vector<uint32_t> want(size_t(P / 16), 0u), inv(size_t(P / 32), 0u), low(size_t(P / 32), 0u);
        int64_t pos = 0;
        for (int s = 0; s < n_seq; ++s) {
            for (int64_t i = 0; i < lens[size_t(s)]; ++i, ++pos) {
                const uint8_t v = frisk_pack2::lut().t[uint8_t(seqs[size_t(s)][size_t(i)])];
                want[size_t(pos >> 4)] |= uint32_t(v & 3u) << (30 - 2 * int(pos & 15));
                if (v & 4u) inv[size_t(pos >> 5)] |= 0x80000000u >> (pos & 31);
                if (v & 8u) low[size_t(pos >> 5)] |= 0x80000000u >> (pos & 31);
            }
            ++pos;
        }
        CHECK(codes == want);
        for (int m = 0; m < 2; ++m) {
            const std::vector<int64_t>& runs = m ? R.low : R.inv;
            std::vector<uint32_t> got(size_t(P / 32), 0u);
            int64_t prev_end = -1;
            for (size_t k = 0; k + 1 < runs.size(); k += 2) {
                CHECK(runs[k] > prev_end && runs[k] < runs[k + 1] && runs[k + 1] <= P);
                prev_end = runs[k + 1];
                for (int64_t p = runs[k]; p < runs[k + 1]; ++p) got[size_t(p >> 5)] |= 0x80000000u >> (p & 31);
            }
            CHECK(got == (m ? low : inv));
            std::vector<int64_t> back;
            frisk_pack2::bitmap_runs((m ? low : inv).data(), lens.data(), n_seq, back);
            CHECK(back == runs);
        }
    }
    // HMM: series of awkward lengths (fewer windows than pieces, one window, constants)
    for (int64_t n : {int64_t(1), int64_t(2), int64_t(3), int64_t(255), int64_t(256), int64_t(257), int64_t(1000), int64_t(70001)}) {
        std::vector<double> x(static_cast<size_t>(n));
        std::normal_distribution<double> g0(-2.0, 0.3), g1(-0.5, 0.6);
        for (int64_t i = 0; i < n; ++i) x[size_t(i)] = ((i / 37) & 1) ? g1(rng) : g0(rng);
        if (n == 3) x[0] = x[1] = x[2] = 0.25;
        const frisk_hmm::Fit f = frisk_hmm::fit(x.data(), n, 10, 1e-2, 1e-3, 1e-2);
        CHECK(f.m.covars[0] > 0 && f.m.covars[1] > 0);
        std::vector<int64_t> off{0, n / 3, n / 3, n};               // (an empty segment in the middle)
        std::vector<int8_t> path(static_cast<size_t>(n), int8_t(9));
        frisk_hmm::viterbi_segments(x.data(), off.data(), 3, f.m, path.data());
        for (int8_t s : path) CHECK(s == 0 || s == 1);
    }
    // FASTA reader + seek index: files of every awkward form the Python mirror accepts; where an index can be built, every record read
    // back through it must equal the parser's bytes
    {
        const std::string path = "/tmp/frisk_san_host.fa";
        const std::vector<std::string> files = {
            "", ">only_header", ">a\nACGT\n>b\n\n>c\nAC\nGT", ">a desc\r\nACGT\r\nAC\r\n>b\r\nTT\r\n", "ACGT\n>late\nAAA\n", ">x\nACGTACGT\nACGTACGT\nAC\n>y\nAAAA\nAAAA\nAAAA",
            ">r\nACGT\nACG\nACGT\n", ">b\nAC GT\n\nAC\n", ">t\n" + std::string(200000, 'G') + "\n>u\n" + std::string(77, 'a'), "\n\n>z\nNNNN\n\n", ">\nAC\n>",
            ">w\nACGTAC\nACGTAC\nACGTAC\n\n>v\nAC\n"};
        std::vector<std::string> all(files);
        for (int big = 0; big < 2; ++big) {             // > 16 MB: the readers' multi-threaded path, runs across lines, blocks and chunk cuts
            std::string t;
            const char letters[] = "ACGTACGTACGTacgtN";
            while (t.size() < (size_t(20) << 20)) {
                t += ">rec" + std::to_string(t.size()) + " x\n";
                size_t n = (rng() % 7 == 0) ? rng() % 50 : rng() % 3000000;
                const size_t width = big ? 61 : 60;
                std::string q(n, 'A');
                for (auto& ch : q) ch = letters[rng() % (sizeof(letters) - 1)];
                for (int k = 0; k < 6 && n > 5000; ++k) {
                    const size_t a = rng() % (n - 4000), ln = rng() % 4000;
                    for (size_t i = a; i < a + ln; ++i) q[i] = (k & 1) ? 'N' : char(q[i] | 0x20);
                }
                for (size_t i = 0; i < n; i += width) { t.append(q, i, std::min(width, n - i)); t += big ? "\r\n" : "\n"; }
            }
            all.push_back(t);
        }
        for (const std::string& text : all) {
            { std::ofstream fh(path, std::ios::binary); fh << text; }
            frisk_fasta::Records rec;
            std::string err;
            const bool ok = frisk_fasta::parse(path.c_str(), rec, err, 1 + int(rng() % 5));
            if (!ok) continue;
            int64_t total = 0;
            for (int64_t n : rec.lens) total += n + 1;
            CHECK(int64_t(rec.stage.size()) >= total && rec.names.size() == rec.lens.size());
            {   // the fused reader (no staging buffer) against parse() + pack_stage(), both letter widths
                for (int wide = 0; wide < 2; ++wide) {
                    frisk_fasta::Records r2;
                    frisk_fasta::CodeVec c2;
                    frisk_pack2::Runs R2;
                    bool fused = false;
                    CHECK(frisk_fasta::parse_pack(path.c_str(), r2, c2, R2, err, 1 + int(rng() % 5), &fused, wide != 0));
                    CHECK(r2.names == rec.names && r2.lens == rec.lens);
                    const int64_t P = frisk_pack2::padded_len(rec.lens.data(), int32_t(rec.lens.size()));
                    std::vector<uint32_t> c1(size_t(P / 16));
                    frisk_pack2::Runs R1;
                    frisk_pack2::pack_stage(rec.stage.data(), rec.lens.data(), int32_t(rec.lens.size()), c1.data(), R1, 3, false);
                    CHECK(c2.size() == c1.size() && std::equal(c1.begin(), c1.end(), c2.begin()));
                    CHECK(R1.inv == R2.inv && R1.low == R2.low);
                }
            }
            frisk_fasta::MappedFile f(path.c_str());
            std::vector<frisk_fasta::FaiEntry> idx;
            std::string why;
            if (!frisk_fasta::build_index(f, idx, why)) continue;
            CHECK(idx.size() == rec.lens.size());
            int64_t off = 0;
            for (size_t r = 0; r < idx.size() && r < rec.lens.size(); ++r) {
                CHECK(idx[r].len == rec.lens[r]);
                std::vector<uint8_t> got(size_t(idx[r].len) + 1, 0);
                frisk_fasta::read_range_mt(f, idx[r], 0, idx[r].len, got.data(), 3);
                CHECK(std::memcmp(got.data(), rec.stage.data() + off, size_t(idx[r].len)) == 0);
                if (idx[r].len > 5) {
                    frisk_fasta::read_range(f, idx[r], 3, idx[r].len - 5, got.data());
                    CHECK(std::memcmp(got.data(), rec.stage.data() + off + 3, size_t(idx[r].len - 5)) == 0);
                }
                off += rec.lens[r] + 1;
            }
            const std::string ip = path + ".fai";
            CHECK(frisk_fasta::write_index(ip.c_str(), f, idx, why));
            std::vector<frisk_fasta::FaiEntry> back;
            const bool rd = frisk_fasta::read_index(ip.c_str(), f, back, why);
            if (!(rd && back.size() == idx.size())) std::printf("read_index: %s (file of %zu bytes, %zu records)\n", why.c_str(), text.size(), idx.size());
            CHECK(rd && back.size() == idx.size());
            ::unlink(ip.c_str());
        }
        ::unlink(path.c_str());
    }
    std::printf("%s (%d failed checks; AVX-512 packer %s)\n", fails ? "FAILED" : "ok", fails, frisk_pack2::have_avx512() ? "exercised" : "not available on this host");
    return fails ? 1 : 0;
}
