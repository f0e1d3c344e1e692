// fasta_index.h - a seek index over a plain FASTA file (host code), so that a rank of a multi-GPU job reads the bytes of ITS
// tiles instead of parsing the whole file (frisk_fasta_load_shard_indexed; the reference reads every record on every run,
// frisk/__init__.py L139-164 / L1442 / L1478).
//
// The index is the five columns of a `samtools faidx` line per record - NAME, LENGTH, OFFSET (byte of the first base),
// LINEBASES, LINEWIDTH - behind one stamp line "#frisk-fai 1 <size of the FASTA> <mtime in ns>".  It exists only for files
// that are REGULAR in faidx's sense and on which the reference's reader (fasta_reader.h: lines stripped, blank lines skipped,
// text before the first header dropped) and plain byte arithmetic agree: nothing but blank lines before the first header,
// no blank line and no surrounding blanks inside a record, every sequence line of a record but its last of one length and
// one width ("\n" or "\r\n").  Anything else - gzip included - has no index, and the caller parses (FRISK_E_INDEX).
// Names follow fasta_reader.h's classify(), i.e. the reference's header rule, not faidx's.
// An index is trusted only for the file it was made from (size + mtime in the stamp) and is checked against the file where
// it is used: the byte before every record's first base must end a header line that gives the record's name, and the byte
// behind its last base must end the line.  A foreign `<fasta>.fai` (no stamp) is accepted on those checks when it is not
// older than the FASTA.
#pragma once
#include <cinttypes>
#include <cstdio>

#include "fasta_reader.h"

namespace frisk_fasta {

struct FaiEntry {
    std::string name;
    int64_t len = 0, offset = 0;
    int64_t linebases = 0, linewidth = 0;
};

// the read-only mapping of a regular file (empty files map to nothing: ok() with size 0)
struct MappedFile {
    const char* base = nullptr;
    size_t size = 0;
    int64_t mtime_ns = 0;
    bool good = false;
    explicit MappedFile(const char* path) {
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) return;
        struct stat sb;
        if (::fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode)) {
            size = size_t(sb.st_size);
            mtime_ns = int64_t(sb.st_mtim.tv_sec) * 1000000000ll + int64_t(sb.st_mtim.tv_nsec);
            if (size == 0) good = true;
            else {
                void* m = ::mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
                if (m != MAP_FAILED) { base = static_cast<const char*>(m); good = true; }
            }
        }
        ::close(fd);
    }
    ~MappedFile() { if (base) ::munmap(const_cast<char*>(base), size); }
    MappedFile(const MappedFile&) = delete;
    MappedFile& operator=(const MappedFile&) = delete;
    bool gzip() const { return size >= 2 && (unsigned char)base[0] == 0x1f && (unsigned char)base[1] == 0x8b; }
};

// One pass over the mapped file.  false + why: the file is not regular (no index for it).
inline bool build_index(const MappedFile& f, std::vector<FaiEntry>& out, std::string& why) {
    out.clear();
    if (!f.good) { why = "cannot map the FASTA file"; return false; }
    if (f.gzip()) { why = "gzip stream"; return false; }
    const char* const base = f.base;
    const char* const end = base + f.size;
    const char* p = base;
    bool in_record = false, short_seen = false, blank_seen = false;
    std::string name;
    while (p < end) {
        const char* nl = static_cast<const char*>(std::memchr(p, '\n', size_t(end - p)));
        const char* le = nl ? nl : end;                 // the line without its '\n'
        const char* b = p;
        const char* e = le;
        const int kind = classify(b, e, &name);
        if (kind < 0) { why = "FASTA header without a name"; return false; }
        if (kind == 0) { if (in_record) blank_seen = true; }
        else if (kind == 2) {
            FaiEntry r;
            r.name = name;
            r.offset = int64_t((nl ? nl + 1 : end) - base);
            out.push_back(r);
            in_record = true; short_seen = false; blank_seen = false;
        } else {
            if (!in_record) { why = "text before the first header"; return false; }
            if (blank_seen) { why = "blank line inside record " + out.back().name; return false; }
            // the raw line must be its stripped text plus nothing but the terminator ("\n", "\r\n", or the end of the file)
            const int64_t term = int64_t(le - e) + (nl ? 1 : 0);
            if (b != p || (le - e) > 1 || (le - e == 1 && *e != '\r')) { why = "blanks around a sequence line of record " + out.back().name; return false; }
            FaiEntry& r = out.back();
            const int64_t n = int64_t(e - b);
            if (short_seen) { why = "lines of different length in record " + r.name; return false; }
            if (r.len == 0) {
                if (r.offset != int64_t(p - base)) { why = "blank line inside record " + r.name; return false; }
                r.linebases = n; r.linewidth = n + term;
            } else if (n > r.linebases) { why = "lines of different length in record " + r.name; return false; }
            // a line shorter than the first, or one whose terminator differs (the file's last line may have none), must be the last
            if (n < r.linebases || n + term != r.linewidth) short_seen = true;
            r.len += n;
        }
        p = le + 1;
    }
    if (out.size() > size_t(0x7FFFFFFF)) { why = "too many FASTA records"; return false; }
    // a one-line record whose only line lacks the final newline: width as if it were there (never used: the line is the last)
    for (FaiEntry& r : out) if (r.linewidth <= r.linebases && r.len > 0) r.linewidth = r.linebases + 1;
    return true;
}

inline bool write_index(const char* path, const MappedFile& f, const std::vector<FaiEntry>& idx, std::string& why) {
    const std::string tmp = std::string(path) + ".tmp" + std::to_string(long(::getpid()));
    FILE* fh = std::fopen(tmp.c_str(), "wb");
    if (!fh) { why = "cannot write " + tmp; return false; }
    std::fprintf(fh, "#frisk-fai 1 %zu %" PRId64 "\n", f.size, f.mtime_ns);
    for (const FaiEntry& r : idx)
        std::fprintf(fh, "%s\t%" PRId64 "\t%" PRId64 "\t%" PRId64 "\t%" PRId64 "\n", r.name.c_str(), r.len, r.offset, r.linebases, r.linewidth);
    const bool ok = std::fflush(fh) == 0 && !std::ferror(fh);
    std::fclose(fh);
    if (!ok || std::rename(tmp.c_str(), path) != 0) { std::remove(tmp.c_str()); why = std::string("cannot write ") + path; return false; }
    return true;
}

// Read an index and check it against the file.  false + why: unusable (missing, stale, malformed, or not this file's).
inline bool read_index(const char* path, const MappedFile& f, std::vector<FaiEntry>& out, std::string& why) {
    out.clear();
    if (!f.good) { why = "cannot map the FASTA file"; return false; }
    if (f.gzip()) { why = "gzip stream"; return false; }
    FILE* fh = std::fopen(path, "rb");
    if (!fh) { why = std::string("no index at ") + path; return false; }
    struct stat sb;
    const bool have_stat = ::fstat(::fileno(fh), &sb) == 0;
    std::vector<char> line(1 << 16);
    bool stamped = false, first = true, bad = false;
    while (std::fgets(line.data(), int(line.size()), fh)) {
        const size_t n = std::strlen(line.data());
        if (n == 0) continue;
        if (line[n - 1] != '\n' && !std::feof(fh)) { bad = true; break; }            // (a line beyond the buffer)
        if (first && line[0] == '#') {
            size_t sz = 0; int64_t mt = 0; int ver = 0;
            if (std::sscanf(line.data(), "#frisk-fai %d %zu %" SCNd64, &ver, &sz, &mt) != 3 || ver != 1) { bad = true; break; }
            if (sz != f.size || mt != f.mtime_ns) { std::fclose(fh); why = "the index was made from another version of the file"; return false; }
            stamped = true; first = false;
            continue;
        }
        first = false;
        char* tab = std::strchr(line.data(), '\t');
        if (!tab) { bad = true; break; }
        FaiEntry r;
        r.name.assign(line.data(), tab);
        if (std::sscanf(tab + 1, "%" SCNd64 "\t%" SCNd64 "\t%" SCNd64 "\t%" SCNd64, &r.len, &r.offset, &r.linebases, &r.linewidth) != 4) { bad = true; break; }
        out.push_back(r);
    }
    std::fclose(fh);
    if (bad) { why = std::string("malformed index ") + path; return false; }
    if (!stamped) {         // a foreign .fai: not older than the file it describes
        const int64_t mt = have_stat ? int64_t(sb.st_mtim.tv_sec) * 1000000000ll + int64_t(sb.st_mtim.tv_nsec) : 0;
        if (mt <= f.mtime_ns) { why = "the index is not newer than the FASTA file"; return false; }
    }
    if (out.size() > size_t(0x7FFFFFFF)) { why = "too many FASTA records"; return false; }
    // every record where the index says it is: a header line that gives its name ends right before the first base, and the
    // last base ends its line; records follow one another in file order
    const int64_t size = int64_t(f.size);
    int64_t prev_end = 0;
    std::string name;
    for (const FaiEntry& r : out) {
        const bool shape = r.len >= 0 && r.offset > 0 && r.offset <= size && r.offset >= prev_end &&
                           (r.len == 0 || (r.linebases > 0 && r.linewidth > r.linebases && r.linewidth <= r.linebases + 2));
        if (!shape) { why = "index entry out of shape: " + r.name; return false; }
        const int64_t full = r.len == 0 ? 0 : (r.len - 1) / r.linebases;                 // complete lines before the last
        const int64_t last = r.offset + full * r.linewidth + (r.len - full * r.linebases);  // byte behind the last base
        if (last > size) { why = "index entry beyond the end of the file: " + r.name; return false; }
        // (an empty record whose header is the file's last line and lacks the newline: build_index puts it at the end of the file)
        const bool eof_header = r.len == 0 && r.offset == size && f.base[size - 1] != '\n';
        if (!eof_header && f.base[r.offset - 1] != '\n') { why = "no line start where the index puts record " + r.name; return false; }
        const char* he = f.base + r.offset - (eof_header ? 0 : 1);                       // the header line: back to its start
        const char* hb = he;
        while (hb > f.base && hb[-1] != '\n') --hb;
        const char* b = hb;
        const char* e = he;
        if (classify(b, e, &name) != 2 || name != r.name) { why = "no header of record " + r.name + " where the index puts it"; return false; }
        if (r.len > 0 && last < size && f.base[last] != '\n' && !(f.base[last] == '\r' && last + 1 < size && f.base[last + 1] == '\n')) {
            why = "record " + r.name + " does not end where the index says"; return false;
        }
        if (r.len > 0 && (is_space((unsigned char)f.base[r.offset]) || is_space((unsigned char)f.base[last - 1]))) {
            why = "record " + r.name + " does not hold bases where the index says"; return false;
        }
        if (!stamped && full > 0) {
            // a foreign index says nothing about blanks inside lines (samtools counts graphic characters only, the reference's
            // reader strips whole lines): the terminator behind the first line must be exactly what linewidth - linebases says,
            // and so must a sample of the interior line ends (and no line may start or end on a blank)
            const int64_t term = r.linewidth - r.linebases;
            const int64_t samples = std::min<int64_t>(full, 64);
            for (int64_t k = 0; k < samples; ++k) {
                const int64_t ln = k == 0 ? 0 : (full - 1) * k / (samples - 1 > 0 ? samples - 1 : 1);
                const char* le = f.base + r.offset + ln * r.linewidth + r.linebases;      // behind the line's last base
                const bool ok_term = term == 1 ? le[0] == '\n' : (le[0] == '\r' && le[1] == '\n');
                bool blank = false;                                  // (the sampled lines are read in full: a blank anywhere in them)
                for (int64_t q = 1; q <= r.linebases && !blank; ++q) blank = is_space((unsigned char)le[-q]);
                if (!ok_term || blank) { why = "record " + r.name + ": a line does not hold bases and end where the index says"; return false; }
            }
        }
        prev_end = last;
    }
    return true;
}

// bases [pos0, pos0 + n) of record r, line terminators skipped, to dst
inline void read_range(const MappedFile& f, const FaiEntry& r, int64_t pos0, int64_t n, uint8_t* dst) {
    if (n <= 0) return;
    int64_t line = pos0 / r.linebases, col = pos0 - line * r.linebases;
    const char* src = f.base + r.offset + line * r.linewidth + col;
    while (n > 0) {
        const int64_t take = std::min<int64_t>(n, r.linebases - col);
        std::memcpy(dst, src, size_t(take));
        dst += take; n -= take;
        src += take + (r.linewidth - r.linebases);
        col = 0;
    }
}

// ... by several threads (a tile of a chromosome is hundreds of megabytes)
inline void read_range_mt(const MappedFile& f, const FaiEntry& r, int64_t pos0, int64_t n, uint8_t* dst, int threads) {
    const int64_t piece = int64_t(8) << 20;
    int T = int(std::min<int64_t>(threads, (n + piece - 1) / piece));
    if (T <= 1) { read_range(f, r, pos0, n, dst); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) {
        const int64_t a = n * t / T, b = n * (t + 1) / T;
        th.emplace_back([&f, &r, pos0, dst, a, b] { read_range(f, r, pos0 + a, b - a, dst + a); });
    }
    for (auto& x : th) x.join();
}

}  // namespace frisk_fasta
