// fasta_pack2.h - a plain FASTA file straight into the 0.25 B/base form (seq_pack2.h), without the one-byte-per-base staging
// buffer in between.  frisk_fasta_load used to parse into that buffer (3.3 GB for a GRCh38-sized assembly), pack from it and
// give it back: the give-back alone - 800 k pages unmapped - cost 0.16 s of a 0.27 s load, with the process's mm lock held.
// Here every reader thread keeps ONE block of a megabyte (cache-resident) that its chunk's sequence lines are appended to, and
// packs the block whenever it is full: same record table (fasta_reader.h's plan), same code words and run lists as
// parse() + pack_stage() - tests/test_pack2_cpu.py compares the two on files of every awkward form.
// gzip streams and files that cannot be mapped take the staged path (fused = false).
#pragma once
#include "fasta_reader.h"
#include "seq_pack2.h"

namespace frisk_fasta {

using CodeVec = std::vector<uint32_t, NoInitAlloc<uint32_t>>;

inline bool parse_pack_plain(const char* base, size_t size, Records& out, CodeVec& codes, frisk_pack2::Runs& runs, std::string& err,
                             int max_threads = 32, bool wide = true) {
    PlainPlan pl;
    if (!plan_plain(base, size, out, pl, err, max_threads)) return false;
    const int64_t P = frisk_pack2::padded_len(out.lens.data(), int32_t(out.lens.size()));
    codes.resize(size_t(P / 16));
    uint32_t* const cw = codes.data();
    const int T = pl.T;
    run_threads(T, [&](int t) {                         // zeroed by the threads that will fill it (first touch)
        const size_t n = codes.size(), a = n * size_t(t) / size_t(T), b = n * size_t(t + 1) / size_t(T);
        std::memset(cw + a, 0, (b - a) * sizeof(uint32_t));
    });
    constexpr size_t CAP = size_t(1) << 20;
    std::vector<frisk_pack2::Runs> part{size_t(T)};
    run_threads(T, [&](int t) {
        std::vector<uint8_t> buf(CAP + 64);
        uint8_t* const bp = buf.data();
        int64_t pos0 = -1;                              // destination of bp[0]
        size_t fill = 0;
        frisk_pack2::Runs& R = part[size_t(t)];
        auto flush_all = [&]() {
            if (fill) frisk_pack2::pack_piece(bp, pos0, pos0 + int64_t(fill), cw, R, wide, true);
            fill = 0;
        };
        auto flush_aligned = [&]() {                    // up to the last code-word boundary; the few letters behind it stay
            const int64_t end = pos0 + int64_t(fill), stop = end & ~int64_t(15);
            if (stop <= pos0) return;
            frisk_pack2::pack_piece(bp, pos0, stop, cw, R, wide, true);
            const size_t rest = size_t(end - stop);
            std::memmove(bp, bp + (stop - pos0), rest);
            pos0 = stop; fill = rest;
        };
        walk_plain(base, size, pl, t, [&](int64_t dst, const char* b, size_t n) {
            if (pos0 < 0 || dst != pos0 + int64_t(fill)) { flush_all(); pos0 = dst; }      // another record
            while (n > 0) {
                const size_t take = std::min(n, CAP - fill);
                std::memcpy(bp + fill, b, take);
                fill += take; b += take; n -= take;
                if (fill == CAP) flush_aligned();
            }
        });
        flush_all();
    });
    size_t ni = 0, nl = 0;
    for (const frisk_pack2::Runs& r : part) { ni += r.inv.size(); nl += r.low.size(); }
    runs.inv.clear(); runs.low.clear();
    runs.inv.reserve(ni); runs.low.reserve(nl);
    for (const frisk_pack2::Runs& r : part) {           // (a run that crosses a thread's or a block's edge: merged by push_run)
        for (size_t k = 0; k + 1 < r.inv.size(); k += 2) frisk_pack2::push_run(runs.inv, r.inv[k], r.inv[k + 1]);
        for (size_t k = 0; k + 1 < r.low.size(); k += 2) frisk_pack2::push_run(runs.low, r.low[k], r.low[k + 1]);
    }
    return true;
}

// `path` -> record table + 0.25 B/base form.  *fused: the file went through parse_pack_plain (no staging buffer).
inline bool parse_pack(const char* path, Records& out, CodeVec& codes, frisk_pack2::Runs& runs, std::string& err, int max_threads = 32,
                       bool* fused = nullptr, bool wide = true) {
    if (fused) *fused = false;
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) { err = std::string("cannot open FASTA file: ") + path; return false; }
    struct stat sb;
    unsigned char magic[2] = {0, 0};
    const bool ok_stat = ::fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode);
    const ssize_t got = ::pread(fd, magic, 2, 0);
    const bool gz = got == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
    void* m = MAP_FAILED;
    if (!gz && ok_stat && sb.st_size > 0) m = ::mmap(nullptr, size_t(sb.st_size), PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (m != MAP_FAILED) {
        ::madvise(m, size_t(sb.st_size), MADV_SEQUENTIAL);
        const bool ok = parse_pack_plain(static_cast<const char*>(m), size_t(sb.st_size), out, codes, runs, err, max_threads, wide);
        ::munmap(m, size_t(sb.st_size));
        if (!ok && err.find(path) == std::string::npos) err += std::string(": ") + path;
        if (ok && fused) *fused = true;
        return ok;
    }
    // gzip, pipes, empty files: the staged reader, then the packer over its buffer
    if (!parse(path, out, err, max_threads)) return false;
    const int64_t P = frisk_pack2::padded_len(out.lens.data(), int32_t(out.lens.size()));
    codes.resize(size_t(P / 16));
    frisk_pack2::pack_stage(out.stage.data(), out.lens.data(), int32_t(out.lens.size()), codes.data(), runs, max_threads, wide);
    ByteVec().swap(out.stage);
    return true;
}

}  // namespace frisk_fasta
