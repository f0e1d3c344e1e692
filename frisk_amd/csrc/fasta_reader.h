// fasta_reader.h - native FASTA / FASTA.gz reader (host code) with the record semantics of the reference's iterFasta
// (frisk/__init__.py L139-164): lines are stripped of surrounding whitespace, blank lines skipped, a line starting with
// '>' opens a record whose name is the first whitespace-delimited token of the header with '>' stripped from both ends,
// every other line of a record is sequence (case preserved), text before the first header is dropped.
// Output = the upload layout of frisk_device.h: record bytes, then one PAD byte, record after record.
//
// Plain files are memory-mapped and parsed by all host threads: pass 1 counts (per chunk of the file: bytes before its
// first header, then name + bytes of every record that starts in it), a sequential prefix turns the counts into record
// lengths and destinations, pass 2 copies the stripped lines to where they belong.  3.3 GB of 60-column FASTA in well under
// a second instead of 4 s; gzip streams stay sequential (inflate is), at zlib's ~0.3 GB/s of output.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace frisk_fasta {

// std::vector<uint8_t> whose resize() does not zero the new bytes (they are all overwritten, by several threads)
// Large blocks (the sequence of a multi-gigabase FASTA) come straight from mmap with transparent huge pages asked for: 2 MB
// pages take the first-touch faults of the placing threads and the unmapping at the end from 800 k page operations per 3 GB
// (0.2 s of unmapping alone, and the mm lock held against every other thread) to 1 600.
template <class T>
struct NoInitAlloc : std::allocator<T> {
    static constexpr size_t BIG = size_t(64) << 20, HUGE_PAGE = size_t(2) << 20;
    template <class U> struct rebind { using other = NoInitAlloc<U>; };
    template <class U, class... A>
    void construct(U* p, A&&... a) {
        if constexpr (sizeof...(A) == 0) ::new (static_cast<void*>(p)) U;
        else ::new (static_cast<void*>(p)) U(std::forward<A>(a)...);
    }
    static size_t mapped_bytes(size_t n) { return (n * sizeof(T) + HUGE_PAGE - 1) / HUGE_PAGE * HUGE_PAGE; }
    T* allocate(size_t n) {
        if (n * sizeof(T) < BIG) return std::allocator<T>::allocate(n);
        void* p = mmap(nullptr, mapped_bytes(n), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (p == MAP_FAILED) throw std::bad_alloc();
        (void)madvise(p, mapped_bytes(n), MADV_HUGEPAGE);            // (advice only: plain pages where huge pages are off)
        return static_cast<T*>(p);
    }
    void deallocate(T* p, size_t n) {
        if (n * sizeof(T) < BIG) std::allocator<T>::deallocate(p, n);
        else (void)munmap(p, mapped_bytes(n));
    }
};
using ByteVec = std::vector<uint8_t, NoInitAlloc<uint8_t>>;

struct Records {
    std::vector<std::string> names;
    std::vector<int64_t> lens;
    ByteVec stage;                  // sum(len + 1) bytes
};

inline bool is_space(unsigned char ch) { return ch == ' ' || (ch >= 9 && ch <= 13); }       // str.strip() on ASCII

// one line [b, e) without its '\n': 0 = blank, 1 = sequence (b, e narrowed to the stripped text), 2 = header (name set),
// -1 = header without a name (IndexError in the reference)
inline int classify(const char*& b, const char*& e, std::string* name) {
    while (b < e && is_space((unsigned char)*b)) ++b;
    while (e > b && is_space((unsigned char)e[-1])) --e;
    if (b == e) return 0;                                                   // L150-151
    if (*b != '>') return 1;                                                // L158-160
    const char* hb = b;
    const char* he = e;
    while (hb < he && *hb == '>') ++hb;                                     // line.strip('>')
    while (he > hb && he[-1] == '>') --he;
    while (hb < he && is_space((unsigned char)*hb)) ++hb;                   // .split()[0]
    const char* te = hb;
    while (te < he && !is_space((unsigned char)*te)) ++te;
    if (te == hb) return -1;
    if (name) name->assign(hb, te);
    return 2;
}

struct ChunkCount {
    int64_t pre = 0;                                   // sequence bytes before the chunk's first header
    std::vector<std::string> names;                    // records that start in the chunk ...
    std::vector<int64_t> bytes;                        // ... and their sequence bytes inside the chunk
    bool bad = false;
};

// lines that START in [b, e) of the mapping [base, base + size)
template <class OnSeq, class OnHeader>
inline bool for_lines(const char* base, size_t size, size_t b, size_t e, OnSeq&& on_seq, OnHeader&& on_header) {
    const char* end = base + size;
    const char* p = base + b;
    const char* stop = base + e;
    std::string name;
    while (p < stop) {
        const char* nl = static_cast<const char*>(std::memchr(p, '\n', size_t(end - p)));
        const char* le = nl ? nl : end;
        const char* lb = p;
        const char* lend = le;
        const int kind = classify(lb, lend, &name);
        if (kind == 1) on_seq(lb, lend);
        else if (kind == 2) on_header(name);
        else if (kind < 0) return false;
        p = le + 1;
    }
    return true;
}

// The front half of reading a mapped plain file: the chunks the threads take (cut at line starts), what every chunk holds
// (pass 1), the record table, and where every chunk's bases go - destinations are BYTE offsets into the staging layout
// (records back to back, one PAD byte behind each) = padded positions of the resident batch.
struct PlainPlan {
    int T = 1;
    std::vector<size_t> cut;                           // chunk t = file bytes [cut[t], cut[t + 1])
    std::vector<int64_t> pre_dst;                      // destination of the chunk's leading bytes (-1: dropped, no record open)
    std::vector<size_t> first_rec;                     // index of the first record that starts in the chunk
    std::vector<int64_t> start;                        // destination of record r's first base; start[n] = total
};

template <class Fn>
inline void run_threads(int T, Fn&& fn) {
    if (T == 1) { fn(0); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back(fn, t);
    for (auto& x : th) x.join();
}

inline bool plan_plain(const char* base, size_t size, Records& out, PlainPlan& pl, std::string& err, int max_threads = 32) {
    unsigned hw = std::thread::hardware_concurrency();
    int T = int(hw ? hw : 1);
    if (T > 32) T = 32;
    if (T > max_threads) T = max_threads < 1 ? 1 : max_threads;      // (N ranks of one node parse the same file at the same time)
    if (size < (size_t(1) << 24)) T = 1;
    pl.T = T;
    std::vector<size_t>& cut = pl.cut;
    cut.assign(size_t(T) + 1, size);
    cut[0] = 0;
    for (int t = 1; t < T; ++t) {                      // chunk t starts at the first line start at or after t * size / T
        size_t p = size * size_t(t) / size_t(T);
        if (p < cut[size_t(t) - 1]) p = cut[size_t(t) - 1];
        const char* nl = p < size ? static_cast<const char*>(std::memchr(base + p, '\n', size - p)) : nullptr;
        cut[size_t(t)] = nl ? size_t(nl - base) + 1 : size;
    }
    std::vector<ChunkCount> cc{size_t(T)};
    auto pass1 = [&](int t) {
        ChunkCount& c = cc[size_t(t)];
        c.bad = !for_lines(base, size, cut[size_t(t)], cut[size_t(t) + 1],
                           [&](const char* b, const char* e) { if (c.bytes.empty()) c.pre += e - b; else c.bytes.back() += e - b; },
                           [&](const std::string& nm) { c.names.push_back(nm); c.bytes.push_back(0); });
    };
    run_threads(T, pass1);
    for (auto& c : cc) if (c.bad) { err = "FASTA header without a name"; return false; }
    // sequential prefix: record lengths, and where every chunk's pieces go
    std::vector<int64_t>& pre_dst = pl.pre_dst;
    std::vector<size_t>& first_rec = pl.first_rec;
    pre_dst.assign(size_t(T), -1);
    first_rec.assign(size_t(T), 0);
    bool open = false;
    for (int t = 0; t < T; ++t) {
        ChunkCount& c = cc[size_t(t)];
        if (open) { pre_dst[size_t(t)] = -2; out.lens.back() += c.pre; }   // (-2: resolved below, once starts are known)
        first_rec[size_t(t)] = out.names.size();
        for (size_t k = 0; k < c.names.size(); ++k) {
            out.names.push_back(c.names[k]);
            out.lens.push_back(c.bytes[k]);
            open = true;
        }
    }
    if (out.lens.size() > size_t(0x7FFFFFFF)) { err = "too many FASTA records"; return false; }
    std::vector<int64_t>& start = pl.start;
    start.assign(out.lens.size() + 1, 0);
    for (size_t r = 0; r < out.lens.size(); ++r) start[r + 1] = start[r] + out.lens[r] + 1;
    // leading bytes of chunk t continue the record that was open when the chunk began: behind what earlier chunks gave it
    std::vector<int64_t> filled(out.lens.size(), 0);
    for (int t = 0; t < T; ++t) {
        ChunkCount& c = cc[size_t(t)];
        if (pre_dst[size_t(t)] == -2) {
            const size_t r = first_rec[size_t(t)] - 1;
            pre_dst[size_t(t)] = start[r] + filled[r];
            filled[r] += c.pre;
        }
        for (size_t k = 0; k < c.names.size(); ++k) filled[first_rec[size_t(t)] + k] += c.bytes[k];
    }
    return true;
}

// pass 2 of chunk t: every sequence line's stripped text with its destination, in file order: on_piece(destination, bytes, n)
template <class OnPiece>
inline void walk_plain(const char* base, size_t size, const PlainPlan& pl, int t, OnPiece&& on_piece) {
    int64_t dst = pl.pre_dst[size_t(t)];               // (< 0: text before the first header - dropped, L158-160)
    size_t rec = pl.first_rec[size_t(t)];
    for_lines(base, size, pl.cut[size_t(t)], pl.cut[size_t(t) + 1],
              [&](const char* b, const char* e) { if (dst >= 0) { on_piece(dst, b, size_t(e - b)); dst += e - b; } },
              [&](const std::string&) { dst = pl.start[rec]; ++rec; });
}

inline bool parse_plain(const char* base, size_t size, Records& out, std::string& err, int max_threads = 32) {
    PlainPlan pl;
    if (!plan_plain(base, size, out, pl, err, max_threads)) return false;
    const std::vector<int64_t>& start = pl.start;
    out.stage.reserve(size_t(start[out.lens.size()]) + 64);          // (room for the caller's padding to a multiple of 32: no reallocation later)
    out.stage.resize(size_t(start[out.lens.size()]));
    uint8_t* const stage = out.stage.data();
    run_threads(pl.T, [&](int t) {
        walk_plain(base, size, pl, t, [&](int64_t dst, const char* b, size_t n) { std::memcpy(stage + dst, b, n); });
    });
    for (size_t r = 0; r < out.lens.size(); ++r) out.stage[size_t(start[r] + out.lens[r])] = 0;      // PAD behind every record
    return true;
}

inline bool parse_gz(const char* path, Records& out, std::string& err) {
    gzFile fh = gzopen(path, "rb");
    if (!fh) { err = std::string("cannot open FASTA file: ") + path; return false; }
    gzbuffer(fh, 1 << 20);
    std::string carry, name;
    std::vector<char> buf(1 << 22);
    bool in_record = false, bad = false;
    auto handle_line = [&](const char* b, const char* e) {
        const int kind = classify(b, e, &name);
        if (kind == 1) {
            if (in_record) { out.stage.insert(out.stage.end(), reinterpret_cast<const uint8_t*>(b), reinterpret_cast<const uint8_t*>(e)); out.lens.back() += e - b; }
        } else if (kind == 2) {
            if (in_record) out.stage.push_back(0);
            out.names.push_back(name);
            out.lens.push_back(0);
            in_record = true;
        } else if (kind < 0) bad = true;
    };
    for (;;) {
        const int got = gzread(fh, buf.data(), unsigned(buf.size()));
        if (got < 0) { gzclose(fh); err = std::string("read error in FASTA file: ") + path; return false; }
        if (got == 0) break;
        const char* p = buf.data();
        const char* end = p + got;
        while (p < end) {
            const char* nl = static_cast<const char*>(std::memchr(p, '\n', size_t(end - p)));
            if (!nl) { carry.append(p, end); break; }
            if (!carry.empty()) { carry.append(p, nl); handle_line(carry.data(), carry.data() + carry.size()); carry.clear(); }
            else handle_line(p, nl);
            p = nl + 1;
        }
    }
    gzclose(fh);
    if (!carry.empty()) handle_line(carry.data(), carry.data() + carry.size());
    if (in_record) out.stage.push_back(0);
    out.stage.reserve(out.stage.size() + 64);                        // (as parse_plain: room for the caller's padding)
    if (bad) { err = "FASTA header without a name"; return false; }
    if (out.lens.size() > size_t(0x7FFFFFFF)) { err = "too many FASTA records"; return false; }
    return true;
}

inline bool parse(const char* path, Records& out, std::string& err, int max_threads = 32) {
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) { err = std::string("cannot open FASTA file: ") + path; return false; }
    struct stat sb;
    unsigned char magic[2] = {0, 0};
    const bool ok_stat = ::fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode);
    const ssize_t got = ::pread(fd, magic, 2, 0);
    const bool gz = got == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
    bool ok;
    if (!gz && ok_stat && sb.st_size > 0) {
        void* m = ::mmap(nullptr, size_t(sb.st_size), PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { ::close(fd); ok = parse_gz(path, out, err); }
        else {
            ::madvise(m, size_t(sb.st_size), MADV_SEQUENTIAL);
            ok = parse_plain(static_cast<const char*>(m), size_t(sb.st_size), out, err, max_threads);
            ::munmap(m, size_t(sb.st_size));
            ::close(fd);
        }
    } else {
        ::close(fd);
        ok = parse_gz(path, out, err);          // gzip, pipes, empty files
    }
    if (!ok && err.find(path) == std::string::npos) err += std::string(": ") + path;
    return ok;
}

}  // namespace frisk_fasta
