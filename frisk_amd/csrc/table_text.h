// table_text.h - the score table as text, natively (host code; no GPU involved).
//
// Reference: the scan loop writes one row per window with str() of every field (frisk/__init__.py L1487-1494), under
// Python 2, whose str(float) is '%.12g' with '.0' appended to integral values.  At 3 M rows (GRCh38, w = 5000 i = 1000) a
// Python loop over rows costs tens of seconds; this formats them with std::to_chars on all host threads.
#pragma once
#include <charconv>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace frisk_text {

// str(x) of a Python 2 float: repr at 12 significant digits
inline char* put_py2_float(char* p, double x) {
    if (x != x) { std::memcpy(p, "nan", 3); return p + 3; }
    if (x > 1.7976931348623157e308) { std::memcpy(p, "inf", 3); return p + 3; }
    if (x < -1.7976931348623157e308) { std::memcpy(p, "-inf", 4); return p + 4; }
    auto r = std::to_chars(p, p + 40, x, std::chars_format::general, 12);     // == printf("%.12g")
    bool plain = true;                                                          // digits only (and a sign): an integral value
    for (char* q = p; q < r.ptr; ++q)
        if (*q == '.' || *q == 'e' || *q == 'n' || *q == 'i') { plain = false; break; }
    if (plain) { r.ptr[0] = '.'; r.ptr[1] = '0'; return r.ptr + 2; }
    return r.ptr;
}

inline char* put_int(char* p, int64_t v) { return std::to_chars(p, p + 24, v).ptr; }

struct Columns {
    int64_t n;
    const char* const* names;       // per scaffold
    const int32_t* seq_index;
    const int64_t* start;
    const int64_t* stop;
    const uint8_t* kld_is_int0;     // nullable: rows whose KLD is the int 0 (no max-mer: empty sum, L465)
    const double* kld;
    const double* gc;
    const double* pi;               // nullable (with si, cri): the three RIP columns
    const double* si;
    const double* cri;
};

inline void format_range(const Columns& c, int64_t r0, int64_t r1, std::string& out) {
    out.clear();
    out.reserve(size_t(r1 - r0) * 72);
    char buf[512];
    for (int64_t r = r0; r < r1; ++r) {
        const char* nm = c.names[c.seq_index[r]];
        out.append(nm);
        char* p = buf;
        *p++ = '\t'; p = put_int(p, c.start[r]);
        *p++ = '\t'; p = put_int(p, c.stop[r]);
        *p++ = '\t';
        if (c.kld_is_int0 && c.kld_is_int0[r]) *p++ = '0'; else p = put_py2_float(p, c.kld[r]);
        *p++ = '\t'; p = put_py2_float(p, c.gc[r]);
        if (c.pi) {
            *p++ = '\t'; p = put_py2_float(p, c.pi[r]);
            *p++ = '\t'; p = put_py2_float(p, c.si[r]);
            *p++ = '\t'; p = put_py2_float(p, c.cri[r]);
        }
        *p++ = '\n';
        out.append(buf, size_t(p - buf));
    }
}

// all rows, in order, as one malloc'd buffer
inline char* format_all(const Columns& c, int64_t* out_len) {
    unsigned hw = std::thread::hardware_concurrency();
    int nt = int(hw ? hw : 1);
    if (nt > 32) nt = 32;
    if (c.n < 20000) nt = 1;
    std::vector<std::string> parts;
    parts.resize(size_t(nt));
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) {
        const int64_t r0 = c.n * t / nt, r1 = c.n * (t + 1) / nt;
        if (nt == 1) format_range(c, r0, r1, parts[0]);
        else th.emplace_back([&c, r0, r1, &parts, t]() { format_range(c, r0, r1, parts[size_t(t)]); });
    }
    for (auto& x : th) x.join();
    size_t total = 0;
    for (auto& s : parts) total += s.size();
    char* res = static_cast<char*>(std::malloc(total + 1));
    if (!res) return nullptr;
    size_t o = 0;
    for (auto& s : parts) { std::memcpy(res + o, s.data(), s.size()); o += s.size(); }
    res[total] = 0;
    if (out_len) *out_len = int64_t(total);
    return res;
}

}  // namespace frisk_text
