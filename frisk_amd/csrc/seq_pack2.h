// seq_pack2.h - host side of the 0.25 B/base upload form ("2-bit codes; N / lowercase as sparse interval lists").
//
// The resident layout of frisk_device.h needs three bit arrays per padded position: codes (2 bits), inv, low (1 bit each).
// Over PCIe only the codes travel densely; the two masks travel as lists of half-open runs [begin, end) of padded positions
// (N runs and soft-masked runs are few and long in assemblies) and are expanded to the bitmaps on the device
// (expand_runs_kernel, profile_kernels.h).  PAD positions are known from the scaffold lengths and never travel.
// Classification of a byte = classify_byte of profile_kernels.h (the device packer): ACGT / acgt -> code, everything else
// invalid; lowercase acgt -> soft-masked (case-sensitive tallies of the reference: countN L106-118, calcGC L120-137).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace frisk_pack2 {

// per byte: bits 0-1 code, bit 2 inv, bit 3 low
struct Lut {
    uint8_t t[256];
    Lut() {
        for (int c = 0; c < 256; ++c) t[c] = 4;                 // not one of ACGTacgt
        const char* up = "ATGC";                                // A=0, T=1, G=2, C=3 (reference L70)
        for (int d = 0; d < 4; ++d) {
            t[uint8_t(up[d])] = uint8_t(d);
            t[uint8_t(up[d] | 0x20)] = uint8_t(d | 8);          // lowercase: counted as a k-mer letter, soft-masked
        }
    }
};
inline const Lut& lut() { static const Lut L; return L; }

inline int64_t padded_len(const int64_t* lens, int32_t n_seq) {
    int64_t pos = 0;
    for (int32_t s = 0; s < n_seq; ++s) pos += lens[s] + 1;
    int64_t p = (pos + 31) / 32 * 32;
    return p ? p : 32;
}

struct Runs {
    std::vector<int64_t> inv, low;          // pairs begin, end (padded positions), ascending, disjoint, non-adjacent
};

// append [b, e) to a run list, merging with a run that ends at b
inline void push_run(std::vector<int64_t>& v, int64_t b, int64_t e) {
    if (!v.empty() && v.back() == b) v.back() = e;
    else { v.push_back(b); v.push_back(e); }
}

// eight letters at once (one 64-bit load, first letter in the low byte): their 2-bit codes as 16 bits with the FIRST letter most
// significant, and the two masks as 8 bits (bit i = letter i).  Branch-free word arithmetic; classification as the table above.
struct Eight { uint32_t code16, inv8, low8; };
inline Eight classify8(uint64_t w) {
    constexpr uint64_t K01 = 0x0101010101010101ull, K7F = 0x7F7F7F7F7F7F7F7Full, K80 = 0x8080808080808080ull;
    const uint64_t fold = w & 0xDFDFDFDFDFDFDFDFull;                       // ASCII case folded
    auto zero_bytes = [&](uint64_t t) -> uint64_t { return ~(((t & K7F) + K7F) | t | K7F); };          // 0x80 in every zero byte, exactly
    const uint64_t valid = zero_bytes(fold ^ (K01 * 'A')) | zero_bytes(fold ^ (K01 * 'C')) | zero_bytes(fold ^ (K01 * 'G')) |
                           zero_bytes(fold ^ (K01 * 'T'));
    const uint64_t lower = (w & 0x2020202020202020ull) << 2;               // 0x80 where the letter is lowercase
    // (c >> 1) & 3: A 0, C 1, T 2, G 3  ->  A 0, T 1, G 2, C 3:  low' = low ^ high, high' = low
    const uint64_t x = fold >> 1, lo = x & K01, hi = (x >> 1) & K01;
    uint64_t cb = (((lo << 1) | (lo ^ hi)) & ((valid >> 7) * 3u));         // one code per byte, 0 where invalid
    cb = __builtin_bswap64(cb);                                            // first letter in the top byte
    cb = (cb | (cb >> 6)) & 0x000F000F000F000Full;                         // pairs -> nibbles
    cb = (cb | (cb >> 12)) & 0x000000FF000000FFull;                        // -> bytes
    cb = (cb | (cb >> 24)) & 0xFFFFull;                                    // -> 16 bits
    auto bits8 = [](uint64_t m80) -> uint32_t { return uint32_t(((m80 >> 7) * 0x0102040810204080ull) >> 56); };     // byte i -> bit i
    return Eight{uint32_t(cb), bits8(~valid & K80), bits8(valid & lower)};
}

// one mask of 16 positions from p on (bit i = position p + i) into a run list with an open run carried along
inline void track16(uint32_t m, int64_t p, int64_t& open, std::vector<int64_t>& out) {
    if ((open < 0 && m == 0u) || (open >= 0 && m == 0xFFFFu)) return;      // nothing changes inside these 16: nearly always
    for (int i = 0; i < 16; ++i) {
        const bool on = (m >> i) & 1u;
        if (on) { if (open < 0) open = p + i; }
        else if (open >= 0) { push_run(out, open, p + i); open = -1; }
    }
}

// one mask of 64 positions from p on, FIRST position in bit 63 (bit 63 - i = position p + i), into a run list with an open run carried along
inline void track64_be(uint64_t m, int64_t p, int64_t& open, std::vector<int64_t>& out) {
    if ((open < 0 && m == 0ull) || (open >= 0 && m == ~0ull)) return;      // nothing changes inside these 64: nearly always
    bool on = open >= 0;
    int pos = 0;
    while (pos < 64) {
        const uint64_t y = (on ? ~m : m) << pos;                           // the next position whose bit differs from the state
        if (y == 0ull) break;
        pos += __builtin_clzll(y);
        if (!on) { open = p + pos; on = true; }
        else { push_run(out, open, p + pos); open = -1; on = false; }
    }
}

// Sixty-four letters per step where the host has AVX-512BW + BMI2 (every x86 server the MI355X ships in): the letters' order is
// reversed inside the vector first, so that the byte masks come out with the FIRST letter in bit 63 - the order of the code
// words and of track64_be -; validity is four byte compares of the case-folded letters, the two code planes are two bit tests
// (classify8's arithmetic: (c >> 1) & 3 = A 0, C 1, T 2, G 3 -> high' = low, low' = low ^ high), and PDEP interleaves the planes
// into the 2-bit codes, thirty-two letters per instruction.  Same words and runs as the 8-letter form (tests/test_pack2_cpu.py
// runs both).  The device pass of the HIP compiler sees a stub: this is host code.
#if (defined(__x86_64__) || defined(_M_X64)) && !defined(__HIP_DEVICE_COMPILE__) && !defined(FRISK_PACK_NO_AVX512)
#define FRISK_PACK_AVX512 1
}  // namespace frisk_pack2
#include <immintrin.h>
namespace frisk_pack2 {
inline bool have_avx512() {
    static const bool ok = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("bmi2");
    return ok;
}
// positions [p, p + 64 k) with p a multiple of 16 and as many whole steps as fit below p1; returns the new p (s advances alike)
__attribute__((target("avx512f,avx512bw,bmi2"))) inline int64_t pack64_avx512(const uint8_t*& s, int64_t p, int64_t p1, uint32_t* codes,
                                                                              int64_t& inv_b, int64_t& low_b, Runs& out) {
    const __m512i rev_in_lane = _mm512_set_epi8(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15,
                                                0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    const __m512i rev_lanes = _mm512_set_epi64(1, 0, 3, 2, 5, 4, 7, 6);
    const __m512i fold_m = _mm512_set1_epi8(char(0xDF)), cA = _mm512_set1_epi8('A'), cC = _mm512_set1_epi8('C'), cG = _mm512_set1_epi8('G'),
                  cT = _mm512_set1_epi8('T'), b02 = _mm512_set1_epi8(0x02), b04 = _mm512_set1_epi8(0x04), b20 = _mm512_set1_epi8(0x20);
    for (; p + 64 <= p1; p += 64, s += 64) {
        __m512i v = _mm512_loadu_si512(reinterpret_cast<const void*>(s));
        v = _mm512_permutexvar_epi64(rev_lanes, _mm512_shuffle_epi8(v, rev_in_lane));           // byte 63 = the first letter
        const __m512i f = _mm512_and_si512(v, fold_m);
        const uint64_t valid = _mm512_cmpeq_epi8_mask(f, cA) | _mm512_cmpeq_epi8_mask(f, cC) | _mm512_cmpeq_epi8_mask(f, cG) | _mm512_cmpeq_epi8_mask(f, cT);
        const uint64_t lo = _mm512_test_epi8_mask(f, b02), hi = _mm512_test_epi8_mask(f, b04);
        const uint64_t H = lo & valid, L = (lo ^ hi) & valid;                                     // code planes, first letter in bit 63
        const uint64_t w01 = _pdep_u64(H >> 32, 0xAAAAAAAAAAAAAAAAull) | _pdep_u64(L >> 32, 0x5555555555555555ull);
        const uint64_t w23 = _pdep_u64(H & 0xFFFFFFFFull, 0xAAAAAAAAAAAAAAAAull) | _pdep_u64(L & 0xFFFFFFFFull, 0x5555555555555555ull);
        uint32_t* c = codes + (p >> 4);
        c[0] |= uint32_t(w01 >> 32); c[1] |= uint32_t(w01); c[2] |= uint32_t(w23 >> 32); c[3] |= uint32_t(w23);
        track64_be(~valid, p, inv_b, out.inv);
        track64_be(valid & _mm512_test_epi8_mask(v, b20), p, low_b, out.low);
    }
    return p;
}
#else
inline bool have_avx512() { return false; }
inline int64_t pack64_avx512(const uint8_t*&, int64_t p, int64_t, uint32_t*, int64_t&, int64_t&, Runs&) { return p; }
#endif

// positions [p0, p1) of one scaffold piece: `src` = its bytes, codes written to the words of `codes` (2 bits per position,
// big-endian in the word; the caller zeroed them), runs appended.  Words shared by two pieces are written by ONE thread only
// because the threads' ranges are cut at multiples of 32.  Whole 16-position words go eight letters at a time (classify8);
// sixty-four where the host has AVX-512, `wide`), the ragged head and tail of a piece letter by letter.
// (shared_edges: the piece's first and last code word may be shared with a piece that ANOTHER thread packs at the same time - they
//  are OR-ed in atomically; every word in between belongs to this piece alone)
inline void pack_piece(const uint8_t* src, int64_t p0, int64_t p1, uint32_t* codes, Runs& out, bool wide = true, bool shared_edges = false) {
    const uint8_t* T = lut().t;
    int64_t inv_b = -1, low_b = -1;
    int64_t p = p0;
    const uint8_t* s = src;
    auto letters = [&](int64_t upto) {
        while (p < upto) {
            const int64_t wend = std::min<int64_t>(upto, (p | 15) + 1);   // positions of one code word
            uint32_t word = 0;
            for (; p < wend; ++p, ++s) {
                const uint32_t v = T[*s];
                word |= (v & 3u) << (30 - 2 * int(p & 15));
                if (v & 4u) { if (inv_b < 0) inv_b = p; }
                else if (inv_b >= 0) { push_run(out.inv, inv_b, p); inv_b = -1; }
                if (v & 8u) { if (low_b < 0) low_b = p; }
                else if (low_b >= 0) { push_run(out.low, low_b, p); low_b = -1; }
            }
            if (shared_edges) __atomic_fetch_or(&codes[(p - 1) >> 4], word, __ATOMIC_RELAXED);
            else codes[(p - 1) >> 4] |= word;
        }
    };
    letters(std::min<int64_t>(p1, (p0 + 15) & ~int64_t(15)));              // up to the first word boundary
    if (wide && have_avx512()) p = pack64_avx512(s, p, p1, codes, inv_b, low_b, out);
    for (; p + 16 <= p1; p += 16, s += 16) {
        uint64_t a, b;
        std::memcpy(&a, s, 8); std::memcpy(&b, s + 8, 8);
        const Eight x = classify8(a), y = classify8(b);
        codes[p >> 4] |= (x.code16 << 16) | y.code16;
        track16(x.inv8 | (y.inv8 << 8), p, inv_b, out.inv);
        track16(x.low8 | (y.low8 << 8), p, low_b, out.low);
    }
    letters(p1);
    if (inv_b >= 0) push_run(out.inv, inv_b, p1);
    if (low_b >= 0) push_run(out.low, low_b, p1);
}

// The whole batch: scaffold s at padded positions [off[s], off[s] + lens[s]).  `codes`: 2 * P / 32 words, zeroed here.
// Work is cut at multiples of 32 positions and dealt to `threads` workers; runs that cross a cut are merged afterwards.
inline void pack_batch(const uint8_t* const* seqs, const int64_t* lens, int32_t n_seq, uint32_t* codes, Runs& out, int threads, bool wide = true) {
    const int64_t P = padded_len(lens, n_seq);
    std::vector<int64_t> off(size_t(n_seq) + 1, 0);
    for (int32_t s = 0; s < n_seq; ++s) off[size_t(s) + 1] = off[size_t(s)] + lens[s] + 1;
    const int64_t words32 = P / 32;
    int T = std::max(1, threads);
    if (P < (int64_t(1) << 22)) T = 1;
    T = int(std::min<int64_t>(T, std::max<int64_t>(1, words32)));
    std::vector<Runs> part{size_t(T)};
    auto work = [&](int t) {
        const int64_t a = words32 * t / T * 32, b = words32 * (t + 1) / T * 32;
        std::memset(codes + a / 16, 0, size_t(b - a) / 16 * 4);
        // scaffolds that meet [a, b)
        int32_t s = int32_t(std::upper_bound(off.begin(), off.end(), a) - off.begin()) - 1;
        if (s < 0) s = 0;
        for (; s < n_seq && off[size_t(s)] < b; ++s) {
            const int64_t q0 = std::max<int64_t>(a, off[size_t(s)]), q1 = std::min<int64_t>(b, off[size_t(s)] + lens[s]);
            if (q1 > q0) pack_piece(seqs[s] + (q0 - off[size_t(s)]), q0, q1, codes, part[size_t(t)], wide);
        }
    };
    if (T == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 1; t < T; ++t) th.emplace_back(work, t);
        work(0);
        for (auto& x : th) x.join();
    }
    size_t ni = 0, nl = 0;
    for (const Runs& r : part) { ni += r.inv.size(); nl += r.low.size(); }
    out.inv.clear(); out.low.clear();
    out.inv.reserve(ni); out.low.reserve(nl);
    for (const Runs& r : part) {
        for (size_t k = 0; k + 1 < r.inv.size(); k += 2) push_run(out.inv, r.inv[k], r.inv[k + 1]);
        for (size_t k = 0; k + 1 < r.low.size(); k += 2) push_run(out.low, r.low[k], r.low[k + 1]);
    }
}

// The same from the parser's buffer (fasta_reader.h: records back to back, one PAD byte behind each): one source pointer.
inline void pack_stage(const uint8_t* stage, const int64_t* lens, int32_t n_seq, uint32_t* codes, Runs& out, int threads, bool wide = true) {
    std::vector<const uint8_t*> ptr(size_t(std::max(n_seq, 1)));
    int64_t pos = 0;
    for (int32_t s = 0; s < n_seq; ++s) { ptr[size_t(s)] = stage + pos; pos += lens[s] + 1; }
    pack_batch(ptr.data(), lens, n_seq, codes, out, threads, wide);
}

// Runs of set bits of a bitmap over positions [0, P) (big-endian in the word), with the PAD positions - behind every
// scaffold and behind the batch - taken out: the run lists of a batch that is resident as bitmaps (sequence cache).
inline void bitmap_runs(const uint32_t* bits, const int64_t* lens, int32_t n_seq, std::vector<int64_t>& out) {
    out.clear();
    int64_t off = 0;
    for (int32_t s = 0; s < n_seq; ++s) {
        const int64_t a = off, b = off + lens[s];
        int64_t p = a, run_b = -1;
        while (p < b) {
            const uint32_t w = bits[p >> 5];
            const int r = int(p & 31);
            const int64_t wend = std::min<int64_t>(b, (p | 31) + 1);
            if (r == 0 && wend - p == 32 && (w == 0u || w == 0xFFFFFFFFu)) {           // whole words at once
                if (w) { if (run_b < 0) run_b = p; }
                else if (run_b >= 0) { push_run(out, run_b, p); run_b = -1; }
                p = wend;
                continue;
            }
            for (; p < wend; ++p) {
                const bool set = (w >> (31 - int(p & 31))) & 1u;
                if (set) { if (run_b < 0) run_b = p; }
                else if (run_b >= 0) { push_run(out, run_b, p); run_b = -1; }
            }
        }
        if (run_b >= 0) push_run(out, run_b, b);
        off = b + 1;
    }
}

}  // namespace frisk_pack2
