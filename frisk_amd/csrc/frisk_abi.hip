// frisk_abi.hip - host side of libfrisk_hip.so: the C ABI declared in include/frisk_hip.h.
// gfx950 (MI355X) only.  Build: see __graft_entry__.build().
#include <hip/hip_runtime.h>
#include <zlib.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "frisk_hip.h"
#include "frisk_device.h"
#include "profile_kernels.h"
#include "scan_kernel.h"
#include "scan8_kernel.h"
#include "scan_big_kernel.h"
#include "synth_kernel.h"
#include "table_text.h"
#include "fasta_index.h"
#include "fasta_reader.h"
#include "fasta_pack2.h"
#include "seq_pack2.h"
#include "hmm_host.h"

#ifndef FRISK_K7_WPS
#define FRISK_K7_WPS 4              // waves per SIMD (= 256-thread workgroups per CU) of the K = 6, 7 narrow-counter kernels
#endif
#ifndef FRISK_SIDE_SHARE
#define FRISK_SIDE_SHARE 0.06       // 4-bit bulk takes the side-table form when the plain form would hand on more than this share of the sample
#endif
#ifndef FRISK_K8_WIDTH
#define FRISK_K8_WIDTH 0            // order-8 counters of the default K = 8 path (scan8_kernel.h): 0 = adaptive 4/8 bits, 4, 8, 16 = off
#endif

namespace {

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;     // elements
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        size_t want = n + n / 8 + 64;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), want * sizeof(T));
        if (e != hipSuccess) { p = nullptr; return e; }
        cap = want;
        return hipSuccess;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

}  // namespace

struct frisk_ctx {
    int device = 0;
    int kmin = 1, kmax = 8;
    int64_t nprof = 0;
    int num_cu = 256;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double ms[3] = {-1.0, -1.0, -1.0};
    std::string err;

    // sequence batches: bat[cur] is the resident one; the other slot receives the next batch while the resident one is
    // being scanned (frisk_seq_stage / frisk_seq_commit)
    struct Batch {
        int32_t n_seq = 0;
        std::vector<int64_t> seq_off, seq_len;
        std::vector<std::string> seq_name;
        int64_t padded_len = 0;
        bool have_seq = false;
        DevBuf<uint8_t> d_ascii;
        DevBuf<uint32_t> d_codes, d_inv, d_low;
        // A TILED batch (frisk_fasta_load_shard) holds, per scaffold, only the bases one rank needs: the windows of its
        // candidate range plus the positions whose k-mers it counts.  "Sequences" of the layout above are then tiles.
        struct Tile {
            int32_t scaf;               // index of the scaffold in the FASTA (what seq_index reports)
            int64_t size;               // length of the whole scaffold
            int64_t base0;              // position inside the scaffold of the tile's first base
            int64_t j0, ncand;          // first window index inside the scaffold and number of windows scored here
            int32_t kind;               // as ScafDesc::kind
            int64_t own0, own1;         // [own0, own1): scaffold positions whose k-mers THIS rank counts in phase A
        };
        int width_hint = 0, hint_w = 0, hint_inc = 0, hint_side = 0;   // counter width the last sampled scan of this batch chose (0: none yet)
        bool tiled = false;
        std::vector<Tile> tiles;
        int32_t tile_w = 0, tile_inc = 0;
        uint32_t tile_flags = 0;
        int64_t cand_begin = 0, cand_end = 0;       // the rank's range of the job's candidate numbering
        std::vector<std::string> g_name;            // all records of the FASTA
        std::vector<int64_t> g_len;
        // A STREAMED batch (frisk_seq_stage_2bit): the codes arrive in pieces on the copy stream; piece i - bitmap words
        // [piece_end[i-1], piece_end[i]) and their code words - is complete when piece_ev[i] has passed.  `streaming`: the compute
        // stream has not waited for them yet (frisk_profile_add follows the pieces one by one; anything else waits for the last).
        std::vector<int64_t> piece_end;
        std::vector<hipEvent_t> piece_ev;
        bool streaming = false;
        DevBuf<int64_t> d_runs;                     // the run lists of the two masks and of the PADs, as uploaded
        // frisk_fasta_load packs on the host and keeps the 0.25 B/base form until the next load: frisk_seq_export_2bit (the CLI's
        // sequence cache) hands it out without touching the device
        std::vector<uint32_t, frisk_fasta::NoInitAlloc<uint32_t>> h_codes;
        frisk_pack2::Runs h_runs;
        bool have_host2 = false;
        int64_t* h_pads = nullptr;                  // page-locked: the PAD runs on their way to the device
        size_t h_pads_cap = 0;
        void release() {
            d_ascii.release(); d_codes.release(); d_inv.release(); d_low.release(); d_runs.release();
            if (h_pads) (void)hipHostFree(h_pads);
            h_pads = nullptr; h_pads_cap = 0;
            for (hipEvent_t e : piece_ev) (void)hipEventDestroy(e);
            piece_ev.clear();
        }
    };
    Batch bat[2];
    int cur = 0;
    Batch& b() { return bat[cur]; }
    const Batch& b() const { return bat[cur]; }
    hipStream_t copy_stream = nullptr;      // uploads + packing of the staged batch
    // h2d(): pageable host memory goes through these page-locked pieces (4 worker threads x 2), not through the runtime's own staging
    static constexpr int PIN_N = 8;
    static constexpr size_t PIN_BYTES = size_t(16) << 20;
    void* pin_buf[PIN_N] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t pin_ev[PIN_N] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool pin_busy[PIN_N] = {false, false, false, false, false, false, false, false};
    hipStream_t tail_stream = nullptr;      // frisk_scan: the last sixteenth of a long scan, while the rows of the rest go to the host
    hipEvent_t ev_fork = nullptr, ev_tail_kernels = nullptr, ev_tail_done = nullptr;
    hipEvent_t staged_ev = nullptr;         // recorded behind the staged batch's last operation
    hipEvent_t slot_free_ev = nullptr;      // recorded on `stream` at every commit: everything queued there before it may still read the
    bool slot_ev_set = false;               // batch slot that the NEXT stage overwrites (profile kernels and scans are asynchronous)
    bool staged = false;

    // profile
    DevBuf<int64_t> d_raw, d_cnt, d_sym;
    DevBuf<double> d_ig, d_logtab, d_logtab64, d_logtab32, d_rctab;
    DevBuf<int64_t> d_ovf_list, d_ovf_list2;   // windows handed from 4-bit to 8-bit counters, and from there to the 16-bit form
    uint64_t ig_gen = 1, ring_gen = 0;         // the genome table's generation, and the one whose copy heads d_ig_ring
    DevBuf<double> d_ig_ring;                  // scan8_kernel: per-workgroup ring of genome-side values by position (40 KB each at 20 positions per lane)
    DevBuf<unsigned int> d_verdict;            // the adaptive width's verdict, written on the device by scan8_decide_kernel: {form, handed, would-have, scored}
    DevBuf<unsigned int> d_ovf_count;          // per row segment 32 counters: [0], [1] the lists' lengths, [8..15] the bulk launch's chunk queues, [16] the 8-bit launch's
    int64_t scan_stat[5] = {0, 0, 0, 0, 0};    // most recent scan: counter width of the bulk launch, windows handed 4->8, ->16, row segments,
                                               // 1 = the bulk launch had the side table for period-4 max-mers beside its 4-bit counters
    DevBuf<uint32_t> d_big;      // 32-bit tables of the long-window path (scan_big_kernel.h), one slice per workgroup
    DevBuf<int64_t> d_meta;          // {totalLen, exMax, nnTotal} of the finalised profile, on the device
    int64_t* h_meta = nullptr;       // page-locked mirror, valid after the stream has been synchronised
    bool profile_final = false;
    bool ms_pending[3] = {false, false, false};     // events recorded, elapsed time not read yet (frisk_last_kernel_ms)
    hipEvent_t evp0 = nullptr, evp1 = nullptr;      // profile_add kernel

    // scan plan + outputs
    DevBuf<ScafDesc> d_desc;
    std::vector<ScafDesc> h_desc;
    int32_t plan_w = -1, plan_inc = -1;
    uint32_t plan_flags = 0;
    int64_t plan_ncand = 0;
    int64_t plan_maxwin = 0;
    DevBuf<int32_t> o_seq;
    DevBuf<int64_t> o_start, o_stop, o_meta;
    DevBuf<uint32_t> o_status, o_counts;
    DevBuf<double> o_kld, o_gc, o_pi, o_si, o_cri, o_sw, o_sg, o_ivom;
    // a SHORT scan's row columns sit in one device block and travel to the host as ONE copy into this page-locked block (the
    // runtime spaces small device-to-host copies ~14 us apart: seven of them were 0.1 ms behind a 0.3 ms scan)
    DevBuf<double> o_block;
    void* h_block = nullptr;
    size_t h_block_cap = 0;
    double* want_ivom = nullptr;     // frisk_scan_ivom: host buffer of the per-max-mer dump requested from the next debug scan
};

namespace {

int fail(frisk_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}

#define HIPC(ctx, call)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail((ctx), FRISK_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

int64_t profile_len(int kmin, int kmax) { return table_offset(kmin, kmax + 1); }

int grid_for(int64_t items, int per_block, int max_blocks) {
    int64_t b = (items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return int(b);
}

// layout of a batch in padded coordinates (frisk_device.h): scaffold s at [off, off+len), then at least one PAD
int layout_batch(frisk_ctx* c, frisk_ctx::Batch& B, const int64_t* lens, int32_t n_seq) {
    if (n_seq < 0) return fail(c, FRISK_E_ARG, "n_seq < 0");
    B.seq_off.assign(size_t(n_seq), 0);
    B.seq_len.assign(lens, lens + n_seq);
    B.seq_name.assign(size_t(n_seq), std::string());
    int64_t pos = 0;
    for (int32_t s = 0; s < n_seq; ++s) {
        if (lens[s] < 0) return fail(c, FRISK_E_ARG, "negative scaffold length");
        B.seq_off[size_t(s)] = pos;
        pos += lens[s] + 1;                       // at least one PAD position after every scaffold
    }
    B.padded_len = (pos + 31) / 32 * 32;
    if (B.padded_len == 0) B.padded_len = 32;
    B.n_seq = n_seq;
    B.have_seq = false;
    B.width_hint = 0;
    B.tiled = false;
    B.tiles.clear(); B.g_name.clear(); B.g_len.clear();
    B.streaming = false;        // (piece_end is kept: frisk_seq_stage_2bit looks at the slot's previous upload)
    if (B.have_host2) {
        B.have_host2 = false;
        decltype(B.h_codes)().swap(B.h_codes);
        B.h_runs = frisk_pack2::Runs();
    }
    return FRISK_OK;
}
int layout_batch(frisk_ctx* c, const int64_t* lens, int32_t n_seq) { return layout_batch(c, c->b(), lens, n_seq); }

// packed arrays of a batch: allocate, and mark the words past the batch
int alloc_packed(frisk_ctx* c, frisk_ctx::Batch& B, hipStream_t st) {
    const size_t w32 = size_t(B.padded_len / 32);
    HIPC(c, B.d_codes.reserve(2 * w32 + 8));
    HIPC(c, B.d_inv.reserve(w32 + 8));
    HIPC(c, B.d_low.reserve(w32 + 8));
    // tail words past the batch: codes 0, inv/low all ones (= PAD), so a run can never extend past the end
    HIPC(c, hipMemsetAsync(B.d_codes.p + 2 * w32, 0, 8 * sizeof(uint32_t), st));
    HIPC(c, hipMemsetAsync(B.d_inv.p + w32, 0xFF, 8 * sizeof(uint32_t), st));
    HIPC(c, hipMemsetAsync(B.d_low.p + w32, 0xFF, 8 * sizeof(uint32_t), st));
    return FRISK_OK;
}
int alloc_packed(frisk_ctx* c) { return alloc_packed(c, c->b(), c->stream); }

// ASCII -> packed, on stream `st`, no host synchronisation
int enqueue_pack(frisk_ctx* c, frisk_ctx::Batch& B, hipStream_t st) {
    const int64_t w32 = B.padded_len / 32;
    if (w32 > 0)
        pack_kernel<<<grid_for(w32, 256, c->num_cu * 8), 256, 0, st>>>(B.d_ascii.p, w32, B.d_codes.p, B.d_inv.p, B.d_low.p);
    HIPC(c, hipGetLastError());
    return FRISK_OK;
}

// common tail of frisk_seq_load / frisk_seq_synth / frisk_fasta_load: pack the resident batch, timed, synchronous
int run_pack(frisk_ctx* c) {
    HIPC(c, hipEventRecord(c->ev0, c->stream));
    int rc = enqueue_pack(c, c->b(), c->stream);
    if (rc) return rc;
    HIPC(c, hipEventRecord(c->ev1, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    float ms = 0;
    HIPC(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->ms[2] = ms;
    c->b().have_seq = true;
    c->plan_w = -1;
    return FRISK_OK;
}

// Host -> device copy on stream `st`.  Page-locked sources go straight to the copy engine (asynchronous).  A large PAGEABLE source -
// the parser's buffer of a multi-gigabase FASTA, a Python bytes object - would be staged by the runtime at ~3 GB/s (measured: 1.03 s
// for the 3.3 Gb assembly); here four threads copy it through page-locked 16 MB pieces of the context's, two per thread, and the
// copy engine follows them at PCIe rate.  On return every byte of `src` has been read (the DMAs may still be in flight, from the
// context's buffers).
int h2d(frisk_ctx* c, void* dst, const void* src, size_t n, hipStream_t st) {
    hipPointerAttribute_t attr;
    const bool pinned = hipPointerGetAttributes(&attr, src) == hipSuccess && attr.type == hipMemoryTypeHost;
    if (!pinned) (void)hipGetLastError();                           // (an unregistered host pointer is reported as an error: expected)
    if (pinned || n < 4 * frisk_ctx::PIN_BYTES) {
        HIPC(c, hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, st));
        return FRISK_OK;
    }
    for (int i = 0; i < frisk_ctx::PIN_N; ++i) {
        if (!c->pin_buf[i]) {
            HIPC(c, hipHostMalloc(&c->pin_buf[i], frisk_ctx::PIN_BYTES, hipHostMallocDefault));
            HIPC(c, hipEventCreateWithFlags(&c->pin_ev[i], hipEventDisableTiming));
        }
    }
    constexpr int WORKERS = frisk_ctx::PIN_N / 2;
    std::atomic<size_t> next{0};
    std::atomic<int> bad{0};
    auto work = [&](int wk) {
        if (hipSetDevice(c->device) != hipSuccess) { bad = 1; return; }
        for (int it = 0; !bad; ++it) {
            const size_t off = next.fetch_add(1) * frisk_ctx::PIN_BYTES;
            if (off >= n) break;
            const int b = wk * 2 + (it & 1);
            if (c->pin_busy[b] && hipEventSynchronize(c->pin_ev[b]) != hipSuccess) { bad = 1; break; }   // the DMA that last read this piece
            const size_t len = std::min(frisk_ctx::PIN_BYTES, n - off);
            std::memcpy(c->pin_buf[b], static_cast<const char*>(src) + off, len);
            if (hipMemcpyAsync(static_cast<char*>(dst) + off, c->pin_buf[b], len, hipMemcpyHostToDevice, st) != hipSuccess ||
                hipEventRecord(c->pin_ev[b], st) != hipSuccess) { bad = 1; break; }
            c->pin_busy[b] = true;
        }
    };
    std::vector<std::thread> th;
    for (int wk = 1; wk < WORKERS; ++wk) th.emplace_back(work, wk);
    work(0);
    for (auto& t : th) t.join();
    if (bad) return fail(c, FRISK_E_HIP, "host-to-device copy through the page-locked staging buffers failed");
    return FRISK_OK;
}

// ASCII scaffolds -> B.d_ascii on stream `st` (asynchronous for page-locked sources)
int enqueue_ascii_upload(frisk_ctx* c, frisk_ctx::Batch& B, const uint8_t* const* seqs, const int64_t* lens, int32_t n_seq,
                         hipStream_t st) {
    HIPC(c, B.d_ascii.reserve(size_t(B.padded_len)));
    // PAD everywhere first: the gaps between scaffolds and the tail.  (One memset of the whole buffer costs 0.1 ms per 400 MB.)
    HIPC(c, hipMemsetAsync(B.d_ascii.p, FRISK_PAD_BYTE, size_t(B.padded_len), st));
    for (int32_t s = 0; s < n_seq; ++s)
        if (lens[s] > 0) {
            const int rc = h2d(c, B.d_ascii.p + B.seq_off[size_t(s)], seqs[s], size_t(lens[s]), st);
            if (rc) return rc;
        }
    return FRISK_OK;
}

// A streamed batch (frisk_seq_stage_2bit) whose pieces the compute stream has not followed: wait, on the device, for the last
int settle_stream(frisk_ctx* c) {
    frisk_ctx::Batch& B = c->b();
    if (!B.streaming) return FRISK_OK;
    if (!B.piece_end.empty()) HIPC(c, hipStreamWaitEvent(c->stream, B.piece_ev[B.piece_end.size() - 1], 0));
    B.streaming = false;
    return FRISK_OK;
}

// candidate windows of one scaffold (crawlGenome, L194-251)
void plan_scaffold(int64_t size, int32_t w, int32_t inc, bool all, int64_t& ncand, int32_t& kind) {
    const double limit = double(w) + ((double(w) * 0.75) - double(inc));      // L211 / L222, evaluated as CPython does
    if (double(size) <= limit) {
        kind = 1;
        ncand = all ? 1 : 0;
    } else {
        kind = 0;
        ncand = (size - inc + 1 > 0) ? (size - inc) / inc + 1 : 0;            // len(range(0, size-inc+1, inc)), L228
    }
}

template <int NT, bool K8, int ITS, bool DEBUG>
hipError_t launch_scan(const ScanParams& P, int grid, size_t lds, hipStream_t st) {
    auto kern = scan_kernel<NT, K8, ITS, DEBUG>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       int(lds));
    if (e != hipSuccess) return e;
    kern<<<grid, NT, lds, st>>>(P);
    return hipGetLastError();
}

template <int KMAX, int NT, int ITS, int BITS, int LOGN, int WPS, bool DEBUG, int ROLE = 0, bool SIDE = false>
hipError_t launch_scan8(const ScanParams& P, int num_cu, int64_t work_items, hipStream_t st) {
    constexpr int wg_per_cu = WPS * 256 / NT;
    static_assert(Lds8<KMAX, BITS, LOGN, NT, SIDE>::granules * 1280 * wg_per_cu <= 160 * 1024, "the workgroups meant to share a CU must fit its LDS (allocated in pieces of 1280 bytes)");
    int grid = int(std::max<int64_t>(1, std::min<int64_t>(work_items, int64_t(num_cu) * wg_per_cu)));
    if (grid >= 8) grid &= ~7;
    scan8_kernel<KMAX, NT, ITS, BITS, LOGN, WPS, DEBUG, ROLE, SIDE><<<grid, NT, 0, st>>>(P);      // LDS is static (Lds8)
    return hipGetLastError();
}

// one launch of the narrow-counter K = 8 kernel: counter width, window class (<= 2048 / <= 5120 bases), debug dump
// (side: 4-bit counters with the side table for the period-4 max-mers)
hipError_t launch_narrow(int kmax, int bits, bool small_w, bool debug, const ScanParams& P, int num_cu, int64_t work_items, hipStream_t st,
                         bool sample = false, bool side = false) {
    const bool slides = P.slide_pp > 0 && P.in_list == nullptr;        // (else: the instantiation without the ring, ROLE bit 1)
    if (sample) {           // the sample of the adaptive width: 4-bit counters, its own name in kernel statistics; it runs the
                            // side-table form and counts what the plain form would have handed on as well
        if (small_w) return slides ? launch_scan8<8, 256, 8, 4, 64, 3, false, 1, true>(P, num_cu, work_items, st)
                                   : launch_scan8<8, 256, 8, 4, 64, 3, false, 3, true>(P, num_cu, work_items, st);
        return slides ? launch_scan8<8, 256, 20, 4, 64, 3, false, 1, true>(P, num_cu, work_items, st)
                      : launch_scan8<8, 256, 20, 4, 64, 3, false, 3, true>(P, num_cu, work_items, st);
    }
    if (side && P.in_list == nullptr && bits == 4 && kmax == 8 && !debug) {
        if (small_w) return slides ? launch_scan8<8, 256, 8, 4, 64, 3, false, 0, true>(P, num_cu, work_items, st)
                                   : launch_scan8<8, 256, 8, 4, 64, 3, false, 2, true>(P, num_cu, work_items, st);
        return slides ? launch_scan8<8, 256, 20, 4, 64, 3, false, 0, true>(P, num_cu, work_items, st)
                      : launch_scan8<8, 256, 20, 4, 64, 3, false, 2, true>(P, num_cu, work_items, st);
    }
#define FRISK_L7(K_, ITS_, DBG_) return launch_scan8<K_, 256, ITS_, 8, 64, FRISK_K7_WPS, DBG_>(P, num_cu, work_items, st)
    if (kmax == 7) {        // K = 6, 7: the 8-bit table is 16 / 4 KiB - registers, not LDS, bound the workgroups per CU
        if (debug) { if (small_w) FRISK_L7(7, 8, true); else FRISK_L7(7, 20, true); }
        if (small_w) FRISK_L7(7, 8, false);
        FRISK_L7(7, 20, false);
    }
    if (kmax == 6) {
        if (debug) { if (small_w) FRISK_L7(6, 8, true); else FRISK_L7(6, 20, true); }
        if (small_w) FRISK_L7(6, 8, false);
        FRISK_L7(6, 20, false);
    }
#undef FRISK_L7
#define FRISK_L8(ITS_, BITS_, WPS_, DBG_) return launch_scan8<8, 256, ITS_, BITS_, 64, WPS_, DBG_>(P, num_cu, work_items, st)
    if (bits == 4) {
        if (debug) { if (small_w) FRISK_L8(8, 4, 3, true); else FRISK_L8(20, 4, 3, true); }
        if (!slides) {
            if (small_w) return launch_scan8<8, 256, 8, 4, 64, 3, false, 2>(P, num_cu, work_items, st);
            return launch_scan8<8, 256, 20, 4, 64, 3, false, 2>(P, num_cu, work_items, st);
        }
        if (small_w) FRISK_L8(8, 4, 3, false);
        FRISK_L8(20, 4, 3, false);
    }
    if (debug) { if (small_w) FRISK_L8(8, 8, 2, true); else FRISK_L8(20, 8, 2, true); }
    if (small_w) FRISK_L8(8, 8, 2, false);
    FRISK_L8(20, 8, 2, false);
#undef FRISK_L8
}

// Tuning knobs are read from the environment only in experiment builds (-DFRISK_TUNE); the product library has none.
inline const char* tune_env(const char* name) {
#ifdef FRISK_TUNE
    return std::getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

int build_genome_table(frisk_ctx* c) {
    const int64_t n = int64_t(1) << (2 * c->kmax);
    genome_ivom_kernel<<<grid_for(n, 256, 1 << 20), 256, 0, c->stream>>>(c->d_sym.p, c->kmin, c->kmax, c->d_meta.p, c->d_ig.p);
    HIPC(c, hipGetLastError());
    HIPC(c, hipMemcpyAsync(c->h_meta, c->d_meta.p, 3 * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    ++c->ig_gen;                    // (the copy of the table that heads the scan's ring buffer is stale)
    c->profile_final = true;        // in stream order: everything that follows on the context's stream sees the table
    return FRISK_OK;
}

// [lo, hi) of the rank-th of `world` near-equal contiguous parts of range(n)
void split_range(int64_t n, int rank, int world, int64_t& lo, int64_t& hi) {
    const int64_t base = n / world, extra = n % world;
    lo = rank * base + std::min<int64_t>(rank, extra);
    hi = lo + base + (rank < extra ? 1 : 0);
}

// Window-tile sharding: the job's candidate windows are numbered in output order over ALL scaffolds and cut
// into `world` contiguous ranges; a rank keeps, per scaffold, the bases of its windows - [j0*i, (j1-1)*i + w), reaching back
// to size-w where its range includes the scaffold's jumpback windows (L230-243) - and counts (phase A) the k-mers that start
// in [j0*i, j1*i), or up to the scaffold's end behind its last window, so that every base of the genome is counted by
// exactly one rank; K-1 bases beyond that range are kept for the words that start in it.  Scaffolds without windows go to
// the rank whose range holds their position in the numbering.
std::vector<frisk_ctx::Batch::Tile> plan_tiles(const std::vector<int64_t>& lens, int32_t w, int32_t inc, bool all, int kmax,
                                               int rank, int world, int64_t& c0, int64_t& c1) {
    std::vector<int64_t> ncand(lens.size()), first(lens.size());
    std::vector<int32_t> kind(lens.size());
    int64_t total = 0;
    for (size_t s = 0; s < lens.size(); ++s) {
        plan_scaffold(lens[s], w, inc, all, ncand[s], kind[s]);
        first[s] = total;
        total += ncand[s];
    }
    split_range(total, rank, world, c0, c1);
    std::vector<frisk_ctx::Batch::Tile> tiles;
    for (size_t s = 0; s < lens.size(); ++s) {
        const int64_t size = lens[s];
        const int64_t ja = std::max<int64_t>(c0, first[s]) - first[s], jb = std::min<int64_t>(c1, first[s] + ncand[s]) - first[s];
        frisk_ctx::Batch::Tile t;
        t.scaf = int32_t(s); t.size = size; t.kind = kind[s];
        if (jb > ja) {
            t.j0 = ja; t.ncand = jb - ja;
            if (kind[s] == 1) { t.base0 = 0; t.own0 = 0; t.own1 = size; tiles.push_back(t); continue; }
            int64_t a = ja * inc;
            if ((jb - 1) * inc + w > size) a = std::min<int64_t>(a, std::max<int64_t>(0, size - w));    // jumpback windows
            t.base0 = a;
            t.own0 = ja * inc;
            t.own1 = (jb == ncand[s]) ? size : jb * inc;
            tiles.push_back(t);             // (what it keeps resident ends at tile_end())
        } else if (ncand[s] == 0) {
            // no windows at all: counted by the rank whose candidate range holds this scaffold's place in the numbering
            const bool mine = (first[s] >= c0 && first[s] < c1) || (first[s] == total && rank == world - 1) ||
                              (total == 0 && rank == world - 1);
            if (!mine) continue;
            t.j0 = 0; t.ncand = 0; t.base0 = 0; t.own0 = 0; t.own1 = size;
            tiles.push_back(t);
        }
    }
    return tiles;
}

// one past the last scaffold position a tile keeps resident
int64_t tile_end(const frisk_ctx::Batch::Tile& t, int32_t w, int32_t inc, int kmax) {
    if (t.ncand == 0 || t.kind == 1) return t.size;
    int64_t b = (t.j0 + t.ncand - 1) * inc + w;
    if (b > t.size) b = t.size;
    return std::max<int64_t>(b, std::min<int64_t>(t.size, t.own1 + kmax - 1));
}

}  // namespace

extern "C" {

const char* frisk_version(void) { return "frisk_hip 0.1 (gfx950)"; }

int frisk_supported(int kmin, int kmax, int64_t max_window) {
    return (kmin >= 1 && kmin <= kmax && kmax <= FRISK_MAX_K && max_window >= 1 && max_window <= 0x7FFFFFFF) ? 1 : 0;
}

int frisk_create(int device, int kmin, int kmax, frisk_ctx** out) {
    if (!out) return FRISK_E_ARG;
    *out = nullptr;
    frisk_ctx* c = new (std::nothrow) frisk_ctx();
    if (!c) return FRISK_E_HIP;
    *out = c;       // returned even on failure so that frisk_last_error() can be read; caller destroys it
    if (kmin < 1 || kmin > kmax || kmax > FRISK_MAX_K)
        return fail(c, FRISK_E_ARG, "word sizes must satisfy 1 <= kmin <= kmax <= 12");
    c->device = device;
    c->kmin = kmin;
    c->kmax = kmax;
    c->nprof = profile_len(kmin, kmax);
    int ndev = 0;
    HIPC(c, hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(c, FRISK_E_HIP, "no such HIP device");
    HIPC(c, hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPC(c, hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(c, FRISK_E_HIP, std::string("libfrisk_hip is built for gfx950 only; device is ") + prop.gcnArchName);
    c->num_cu = prop.multiProcessorCount;
    HIPC(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIPC(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    HIPC(c, hipEventCreateWithFlags(&c->staged_ev, hipEventDisableTiming));
    HIPC(c, hipEventCreateWithFlags(&c->slot_free_ev, hipEventDisableTiming));
    HIPC(c, hipStreamCreateWithFlags(&c->tail_stream, hipStreamNonBlocking));
    HIPC(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIPC(c, hipEventCreateWithFlags(&c->ev_tail_kernels, hipEventDisableTiming));
    HIPC(c, hipEventCreateWithFlags(&c->ev_tail_done, hipEventDisableTiming));
    HIPC(c, hipEventCreate(&c->ev0));
    HIPC(c, hipEventCreate(&c->ev1));
    HIPC(c, c->d_raw.reserve(size_t(c->nprof) + 4));
    HIPC(c, c->d_cnt.reserve(size_t(c->nprof) + 4));
    HIPC(c, c->d_sym.reserve(size_t(c->nprof)));
    HIPC(c, c->d_ig.reserve(size_t(1) << (2 * kmax)));
    HIPC(c, c->d_meta.reserve(4));
    HIPC(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_meta), 4 * sizeof(int64_t), hipHostMallocDefault));
    HIPC(c, hipEventCreate(&c->evp0));
    HIPC(c, hipEventCreate(&c->evp1));
    HIPC(c, hipMemsetAsync(c->d_raw.p, 0, (size_t(c->nprof) + 4) * sizeof(int64_t), c->stream));
    {   // range-reduction table of the scan kernel's logarithm (scan_kernel.h: log_tab_pos)
        double tab[2 * FRISK_LOGTAB_N];
        for (int nbin : {FRISK_LOGTAB_N, 64, 32}) {
            for (int i = 0; i < nbin; ++i) {
                const double ci = 0.5 + (double(i) + 0.5) / (2.0 * nbin);
                const double u = 1.0 / ci;
                tab[2 * i] = u;
                tab[2 * i + 1] = double(-logl((long double)u));      // -ln of the ROUNDED reciprocal: the identity stays exact
            }
            DevBuf<double>& dst = nbin == 64 ? c->d_logtab64 : (nbin == 32 ? c->d_logtab32 : c->d_logtab);
            HIPC(c, dst.reserve(2 * size_t(nbin)));
            HIPC(c, hipMemcpyAsync(dst.p, tab, 2 * size_t(nbin) * sizeof(double), hipMemcpyHostToDevice, c->stream));
            HIPC(c, hipStreamSynchronize(c->stream));                // `tab` is reused
        }
        double rc[256];                                          // scan8_kernel.h: weight 1/c of a position whose max-mer occurs c times
        rc[0] = 0.0;
        for (int i = 1; i < 256; ++i) rc[i] = 1.0 / double(i);
        HIPC(c, c->d_rctab.reserve(256));
        HIPC(c, hipMemcpyAsync(c->d_rctab.p, rc, sizeof(rc), hipMemcpyHostToDevice, c->stream));
        HIPC(c, c->d_ovf_count.reserve(64));
        HIPC(c, c->d_verdict.reserve(8));
        HIPC(c, hipStreamSynchronize(c->stream));
    }
    HIPC(c, hipStreamSynchronize(c->stream));
    return FRISK_OK;
}

void frisk_destroy(frisk_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    c->bat[0].release(); c->bat[1].release();
    c->d_raw.release(); c->d_cnt.release(); c->d_sym.release(); c->d_ig.release(); c->d_logtab.release(); c->d_logtab64.release(); c->d_logtab32.release(); c->d_rctab.release(); c->d_ig_ring.release(); c->d_ovf_list.release(); c->d_ovf_list2.release(); c->d_ovf_count.release(); c->d_verdict.release(); c->d_big.release(); c->d_desc.release();
    c->o_seq.release(); c->o_start.release(); c->o_stop.release(); c->o_meta.release();
    c->o_status.release(); c->o_counts.release();
    c->o_ivom.release(); c->o_kld.release(); c->o_gc.release(); c->o_sw.release(); c->o_sg.release(); c->o_pi.release(); c->o_si.release(); c->o_cri.release();
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    if (c->staged_ev) (void)hipEventDestroy(c->staged_ev);
    if (c->slot_free_ev) (void)hipEventDestroy(c->slot_free_ev);
    for (int i = 0; i < frisk_ctx::PIN_N; ++i) {
        if (c->pin_ev[i]) { (void)hipEventSynchronize(c->pin_ev[i]); (void)hipEventDestroy(c->pin_ev[i]); }
        if (c->pin_buf[i]) (void)hipHostFree(c->pin_buf[i]);
    }
    if (c->tail_stream) { (void)hipStreamSynchronize(c->tail_stream); (void)hipStreamDestroy(c->tail_stream); }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_tail_kernels) (void)hipEventDestroy(c->ev_tail_kernels);
    if (c->ev_tail_done) (void)hipEventDestroy(c->ev_tail_done);
    if (c->h_meta) (void)hipHostFree(c->h_meta);
    if (c->h_block) (void)hipHostFree(c->h_block);
    c->o_block.release();
    if (c->evp0) (void)hipEventDestroy(c->evp0);
    if (c->evp1) (void)hipEventDestroy(c->evp1);
    c->d_meta.release();
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* frisk_last_error(const frisk_ctx* c) { return c ? c->err.c_str() : "null context"; }
int64_t frisk_profile_len(const frisk_ctx* c) { return c ? c->nprof : 0; }
int64_t frisk_profile_raw_len(const frisk_ctx* c) { return c ? c->nprof + 4 : 0; }
int64_t frisk_seq_padded_len(const frisk_ctx* c) { return (c && c->b().have_seq) ? c->b().padded_len : 0; }
void* frisk_host_alloc(frisk_ctx* c, int64_t bytes) {
    if (!c || bytes <= 0) return nullptr;
    void* p = nullptr;
    if (hipSetDevice(c->device) != hipSuccess) return nullptr;
    if (hipHostMalloc(&p, size_t(bytes), hipHostMallocDefault) != hipSuccess) { c->err = "hipHostMalloc failed"; return nullptr; }
    return p;
}
void frisk_host_free(frisk_ctx* c, void* ptr) {
    if (c && ptr) { (void)hipSetDevice(c->device); (void)hipHostFree(ptr); }
}
double frisk_last_kernel_ms(const frisk_ctx* cc, int which) {
    frisk_ctx* c = const_cast<frisk_ctx*>(cc);
    if (!c || which < 0 || which >= 3) return -1.0;
    if (which == 1 && c->ms_pending[1]) {           // the profile kernel's events are read on demand: no sync in profile_add
        float ms = 0;
        if (hipEventSynchronize(c->evp1) == hipSuccess && hipEventElapsedTime(&ms, c->evp0, c->evp1) == hipSuccess) c->ms[1] = ms;
        c->ms_pending[1] = false;
    }
    return c->ms[which];
}

// ------------------------------------------------------------------------------------- sequences
int frisk_seq_load(frisk_ctx* c, const uint8_t* const* seqs, const int64_t* lens, int32_t n_seq) {
    if (!c) return FRISK_E_ARG;
    if (n_seq > 0 && (!seqs || !lens)) return fail(c, FRISK_E_ARG, "null sequence table");
    HIPC(c, hipSetDevice(c->device));
    int rc = layout_batch(c, lens, n_seq);
    if (rc) return rc;
    if (n_seq <= 64) {
        rc = enqueue_ascii_upload(c, c->b(), seqs, lens, n_seq, c->stream);
        if (rc) return rc;
    } else {            // many small scaffolds: assemble once on the host, one copy
        HIPC(c, c->b().d_ascii.reserve(size_t(c->b().padded_len)));
        std::vector<uint8_t> stage(size_t(c->b().padded_len), uint8_t(FRISK_PAD_BYTE));
        for (int32_t s = 0; s < n_seq; ++s)
            if (lens[s] > 0) std::memcpy(stage.data() + c->b().seq_off[size_t(s)], seqs[s], size_t(lens[s]));
        rc = h2d(c, c->b().d_ascii.p, stage.data(), stage.size(), c->stream);
        if (rc) return rc;
        HIPC(c, hipStreamSynchronize(c->stream));
    }
    rc = alloc_packed(c);
    if (rc) return rc;
    return run_pack(c);
}

static int upload_2bit(frisk_ctx* c, frisk_ctx::Batch& B, const uint32_t* codes, const int64_t* inv_runs, int64_t n_inv,
                       const int64_t* low_runs, int64_t n_low, int64_t piece_bases, hipStream_t st);

int frisk_fasta_load(frisk_ctx* c, const char* path, int32_t* n_seq_out, int64_t* total_len_out) {
    if (!c || !path) return FRISK_E_ARG;
    frisk_fasta::Records rec;
    std::string err;
#ifdef FRISK_TUNE
    const auto tt0 = std::chrono::steady_clock::now();
#endif
    // Read and packed on the HOST, by the reader's threads, straight into the 0.25 B/base form (fasta_pack2.h: no one-byte-per-base
    // buffer in between; seq_pack2.h: 2-bit codes + run lists of the two masks).  PCIe then carries a quarter of the bytes (3.3 GB
    // of ASCII -> 0.82 GB for a GRCh38-sized assembly), and the host copy stays until the next load: the CLI's sequence cache is
    // written from it without touching the device.
    const unsigned hw = std::thread::hardware_concurrency();
    frisk_fasta::CodeVec packed;
    frisk_pack2::Runs packed_runs;
    if (!frisk_fasta::parse_pack(path, rec, packed, packed_runs, err, int(std::min(32u, hw ? hw : 1u)))) return fail(c, FRISK_E_ARG, err);
#ifdef FRISK_TUNE
    const auto tt1 = std::chrono::steady_clock::now();
#endif
    HIPC(c, hipSetDevice(c->device));
    int rc = layout_batch(c, rec.lens.data(), int32_t(rec.lens.size()));
    if (rc) return rc;
    frisk_ctx::Batch& B = c->b();
    B.seq_name = rec.names;
    if (packed.size() != size_t(B.padded_len / 32) * 2) return fail(c, FRISK_E_STATE, "frisk_fasta_load: packed length and batch layout disagree");
    B.h_codes.swap(packed);
    B.h_runs = std::move(packed_runs);
    B.have_host2 = true;
#ifdef FRISK_TUNE
    const auto tt2 = std::chrono::steady_clock::now();
#endif
    HIPC(c, hipEventRecord(c->ev0, c->stream));
    rc = upload_2bit(c, B, B.h_codes.data(), B.h_runs.inv.data(), int64_t(B.h_runs.inv.size() / 2), B.h_runs.low.data(),
                     int64_t(B.h_runs.low.size() / 2), 0, c->stream);
    if (rc) return rc;
    HIPC(c, hipEventRecord(c->ev1, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    float ms = 0;
    HIPC(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->ms[2] = ms;                              // (upload + mask expansion: there is no device-side packing on this path)
    B.have_seq = true;
    c->plan_w = -1;
#ifdef FRISK_TUNE
    {
        const auto tt3 = std::chrono::steady_clock::now();
        auto msf = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        if (std::getenv("FRISK_LOAD_SPLIT")) std::fprintf(stderr, "[load] read + pack %.1f ms, layout %.1f ms, upload %.1f ms\n", msf(tt0, tt1), msf(tt1, tt2), msf(tt2, tt3));
    }
#endif
    int64_t total = 0;
    for (int64_t v : rec.lens) total += v;
    if (n_seq_out) *n_seq_out = int32_t(rec.lens.size());
    if (total_len_out) *total_len_out = total;
    return FRISK_OK;
}

int frisk_fasta_pack_2bit(const char* path, int32_t* n_seq, int64_t** lens, uint32_t** codes, int64_t* n_code_words, int64_t** inv_runs,
                          int64_t* n_inv, int64_t** low_runs, int64_t* n_low) {
    if (!path || !n_seq || !lens || !codes || !n_code_words || !inv_runs || !n_inv || !low_runs || !n_low) return FRISK_E_ARG;
    frisk_fasta::Records rec;
    frisk_fasta::CodeVec packed;
    frisk_pack2::Runs R;
    std::string err;
    const unsigned hw = std::thread::hardware_concurrency();
    if (!frisk_fasta::parse_pack(path, rec, packed, R, err, int(std::min(32u, hw ? hw : 1u)))) return FRISK_E_ARG;
    auto give = [](const void* src, size_t bytes, void** out) -> bool {
        *out = std::malloc(std::max<size_t>(bytes, 16));
        if (!*out) return false;
        if (bytes) std::memcpy(*out, src, bytes);
        return true;
    };
    void *pl = nullptr, *pc = nullptr, *pi = nullptr, *pw = nullptr;
    const bool ok = give(rec.lens.data(), rec.lens.size() * 8, &pl) && give(packed.data(), packed.size() * 4, &pc) &&
                    give(R.inv.data(), R.inv.size() * 8, &pi) && give(R.low.data(), R.low.size() * 8, &pw);
    if (!ok) { std::free(pl); std::free(pc); std::free(pi); std::free(pw); return FRISK_E_HIP; }
    *n_seq = int32_t(rec.lens.size());
    *lens = static_cast<int64_t*>(pl); *codes = static_cast<uint32_t*>(pc); *n_code_words = int64_t(packed.size());
    *inv_runs = static_cast<int64_t*>(pi); *n_inv = int64_t(R.inv.size() / 2);
    *low_runs = static_cast<int64_t*>(pw); *n_low = int64_t(R.low.size() / 2);
    return FRISK_OK;
}

int frisk_fasta_digest(const char* path, int32_t* n_seq, int64_t* total_len, uint64_t* digest) {
    if (!path) return FRISK_E_ARG;
    frisk_fasta::Records rec;
    std::string err;
    if (!frisk_fasta::parse(path, rec, err)) return FRISK_E_ARG;
    uint64_t h = 1469598103934665603ull;                       // FNV-1a over name \0 sequence \0, record after record
    auto eat = [&h](const uint8_t* p, size_t n) { for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; } };
    const uint8_t zero = 0;
    int64_t total = 0, off = 0;
    for (size_t r = 0; r < rec.lens.size(); ++r) {
        eat(reinterpret_cast<const uint8_t*>(rec.names[r].data()), rec.names[r].size());
        eat(&zero, 1);
        eat(rec.stage.data() + off, size_t(rec.lens[r]));
        eat(&zero, 1);
        if (rec.stage[size_t(off + rec.lens[r])] != FRISK_PAD_BYTE) return FRISK_E_STATE;       // the PAD behind every record
        off += rec.lens[r] + 1;
        total += rec.lens[r];
    }
    if (size_t(off) != rec.stage.size()) return FRISK_E_STATE;
    if (n_seq) *n_seq = int32_t(rec.lens.size());
    if (total_len) *total_len = total;
    if (digest) *digest = h;
    return FRISK_OK;
}

// common tail of the two shard loaders: the tiles' ASCII is on the device; pack it and remember what the tiles are
static int finish_tiled(frisk_ctx* c, const std::vector<frisk_ctx::Batch::Tile>& tiles, const std::vector<std::string>& names,
                        const std::vector<int64_t>& lens, int32_t w, int32_t inc, uint32_t flags, int64_t c0, int64_t c1,
                        int32_t* n_seq_out, int64_t* total_len_out, int64_t* cand_begin, int64_t* cand_end) {
    frisk_ctx::Batch& B = c->b();
    int rc = alloc_packed(c);
    if (rc) return rc;
    rc = run_pack(c);
    if (rc) return rc;
    B.tiled = true;
    B.tiles = tiles;
    B.tile_w = w; B.tile_inc = inc; B.tile_flags = flags & FRISK_SCAN_SCAFFOLDS_ALL;
    B.cand_begin = c0; B.cand_end = c1;
    B.g_name = names;
    B.g_len = lens;
    int64_t total = 0;
    for (int64_t v : lens) total += v;
    if (n_seq_out) *n_seq_out = int32_t(lens.size());
    if (total_len_out) *total_len_out = total;
    if (cand_begin) *cand_begin = c0;
    if (cand_end) *cand_end = c1;
    return FRISK_OK;
}

int frisk_fasta_load_shard(frisk_ctx* c, const char* path, int32_t w, int32_t inc, uint32_t flags, int32_t rank, int32_t world,
                           int32_t* n_seq_out, int64_t* total_len_out, int64_t* cand_begin, int64_t* cand_end) {
    if (!c || !path) return FRISK_E_ARG;
    if (w < 1 || inc < 1) return fail(c, FRISK_E_ARG, "window length and increment must be >= 1");
    if (world < 1 || rank < 0 || rank >= world) return fail(c, FRISK_E_ARG, "rank outside [0, world)");
    frisk_fasta::Records rec;
    std::string err;
    // (the ranks of a job parse the same file at the same time, normally on one node: each takes its share of the host's threads)
    const unsigned hw = std::thread::hardware_concurrency();
    if (!frisk_fasta::parse(path, rec, err, int(std::max(1u, (hw ? hw : 1u) / unsigned(world))))) return fail(c, FRISK_E_ARG, err);
    const std::vector<int64_t>& lens = rec.lens;
    const std::vector<std::string>& names = rec.names;
    const frisk_fasta::ByteVec& stage = rec.stage;
    HIPC(c, hipSetDevice(c->device));
    int64_t c0 = 0, c1 = 0;
    std::vector<frisk_ctx::Batch::Tile> tiles = plan_tiles(lens, w, inc, (flags & FRISK_SCAN_SCAFFOLDS_ALL) != 0, c->kmax, rank, world, c0, c1);
    std::vector<int64_t> rec_off(lens.size());          // where record s starts in `stage`
    int64_t pos = 0;
    for (size_t s = 0; s < lens.size(); ++s) { rec_off[s] = pos; pos += lens[s] + 1; }
    std::vector<int64_t> tlen(tiles.size());
    for (size_t t = 0; t < tiles.size(); ++t) tlen[t] = tile_end(tiles[t], w, inc, c->kmax) - tiles[t].base0;
    int rc = layout_batch(c, tlen.data(), int32_t(tiles.size()));
    if (rc) return rc;
    frisk_ctx::Batch& B = c->b();
    // the rank's tiles straight from the parser's buffer into the batch layout on the device (no second host copy of them):
    // PAD everywhere first, then one upload per tile
    HIPC(c, B.d_ascii.reserve(size_t(B.padded_len)));
    HIPC(c, hipMemsetAsync(B.d_ascii.p, FRISK_PAD_BYTE, size_t(B.padded_len), c->stream));
    for (size_t t = 0; t < tiles.size(); ++t) {
        if (tlen[t] <= 0) continue;
        rc = h2d(c, B.d_ascii.p + B.seq_off[t], stage.data() + rec_off[size_t(tiles[t].scaf)] + tiles[t].base0, size_t(tlen[t]), c->stream);
        if (rc) return rc;
    }
    HIPC(c, hipStreamSynchronize(c->stream));          // (the parser's buffer goes away with this call)
    return finish_tiled(c, tiles, names, lens, w, inc, flags, c0, c1, n_seq_out, total_len_out, cand_begin, cand_end);
}

// ---- the same from a seek index: the rank maps the file and copies the bytes of ITS tiles, nothing else (fasta_index.h) ----
int frisk_fasta_index_build(const char* fasta_path, const char* index_path, int32_t* n_seq, char* why, int32_t why_cap) {
    if (!fasta_path || !index_path) return FRISK_E_ARG;
    auto say = [&](const std::string& m) { if (why && why_cap > 0) { std::strncpy(why, m.c_str(), size_t(why_cap) - 1); why[why_cap - 1] = 0; } };
    say("");
    frisk_fasta::MappedFile f(fasta_path);
    if (!f.good) { say(std::string("cannot open FASTA file: ") + fasta_path); return FRISK_E_ARG; }
    std::vector<frisk_fasta::FaiEntry> idx;
    std::string msg;
    if (!frisk_fasta::build_index(f, idx, msg)) { say(msg); return FRISK_E_INDEX; }
    if (!frisk_fasta::write_index(index_path, f, idx, msg)) { say(msg); return FRISK_E_ARG; }
    if (n_seq) *n_seq = int32_t(idx.size());
    return FRISK_OK;
}

int frisk_fasta_index_read(const char* fasta_path, const char* index_path, int32_t seq_index, int64_t pos0, int64_t n, uint8_t* out,
                           int32_t* n_seq, int64_t* seq_len, char* name, int32_t name_cap, char* why, int32_t why_cap) {
    if (!fasta_path || !index_path) return FRISK_E_ARG;
    auto say = [&](const std::string& m) { if (why && why_cap > 0) { std::strncpy(why, m.c_str(), size_t(why_cap) - 1); why[why_cap - 1] = 0; } };
    say("");
    frisk_fasta::MappedFile f(fasta_path);
    if (!f.good) { say(std::string("cannot open FASTA file: ") + fasta_path); return FRISK_E_ARG; }
    std::vector<frisk_fasta::FaiEntry> idx;
    std::string msg;
    if (!frisk_fasta::read_index(index_path, f, idx, msg)) { say(msg); return FRISK_E_INDEX; }
    if (n_seq) *n_seq = int32_t(idx.size());
    if (seq_index < 0) return FRISK_OK;                 // (the count alone)
    if (size_t(seq_index) >= idx.size()) { say("record index out of range"); return FRISK_E_ARG; }
    const frisk_fasta::FaiEntry& r = idx[size_t(seq_index)];
    if (seq_len) *seq_len = r.len;
    if (name && name_cap > 0) { std::strncpy(name, r.name.c_str(), size_t(name_cap) - 1); name[name_cap - 1] = 0; }
    if (n > 0) {
        if (!out || pos0 < 0 || pos0 + n > r.len) { say("range outside the record"); return FRISK_E_ARG; }
        frisk_fasta::read_range(f, r, pos0, n, out);
    }
    return FRISK_OK;
}

int frisk_fasta_load_shard_indexed(frisk_ctx* c, const char* path, const char* index_path, int32_t w, int32_t inc, uint32_t flags,
                                   int32_t rank, int32_t world, int32_t* n_seq_out, int64_t* total_len_out, int64_t* cand_begin,
                                   int64_t* cand_end) {
    if (!c || !path || !index_path) return FRISK_E_ARG;
    if (w < 1 || inc < 1) return fail(c, FRISK_E_ARG, "window length and increment must be >= 1");
    if (world < 1 || rank < 0 || rank >= world) return fail(c, FRISK_E_ARG, "rank outside [0, world)");
    frisk_fasta::MappedFile f(path);
    if (!f.good) return fail(c, FRISK_E_ARG, std::string("cannot open FASTA file: ") + path);
    std::vector<frisk_fasta::FaiEntry> idx;
    std::string msg;
    if (!frisk_fasta::read_index(index_path, f, idx, msg)) return fail(c, FRISK_E_INDEX, msg);
    std::vector<int64_t> lens(idx.size());
    std::vector<std::string> names(idx.size());
    for (size_t s = 0; s < idx.size(); ++s) { lens[s] = idx[s].len; names[s] = idx[s].name; }
    HIPC(c, hipSetDevice(c->device));
    int64_t c0 = 0, c1 = 0;
    std::vector<frisk_ctx::Batch::Tile> tiles = plan_tiles(lens, w, inc, (flags & FRISK_SCAN_SCAFFOLDS_ALL) != 0, c->kmax, rank, world, c0, c1);
    std::vector<int64_t> tlen(tiles.size());
    for (size_t t = 0; t < tiles.size(); ++t) tlen[t] = tile_end(tiles[t], w, inc, c->kmax) - tiles[t].base0;
    int rc = layout_batch(c, tlen.data(), int32_t(tiles.size()));
    if (rc) return rc;
    frisk_ctx::Batch& B = c->b();
    // the batch as it will sit on the device, assembled on the host from the mapping (PAD between the tiles), one upload
    frisk_fasta::ByteVec host;
    host.resize(size_t(B.padded_len));
    const unsigned hw = std::thread::hardware_concurrency();
    const int threads = int(std::max(1u, std::min(32u, (hw ? hw : 1u) / unsigned(world))));
    int64_t pos = 0;
    for (size_t t = 0; t < tiles.size(); ++t) {
        const int64_t off = B.seq_off[t];
        std::memset(host.data() + pos, FRISK_PAD_BYTE, size_t(off - pos));
        if (tlen[t] > 0) frisk_fasta::read_range_mt(f, idx[size_t(tiles[t].scaf)], tiles[t].base0, tlen[t], host.data() + off, threads);
        pos = off + (tlen[t] > 0 ? tlen[t] : 0);
    }
    std::memset(host.data() + pos, FRISK_PAD_BYTE, size_t(B.padded_len - pos));
    HIPC(c, B.d_ascii.reserve(size_t(B.padded_len)));
    rc = h2d(c, B.d_ascii.p, host.data(), size_t(B.padded_len), c->stream);
    if (rc) return rc;
    HIPC(c, hipStreamSynchronize(c->stream));          // (the host copy goes away with this call)
    return finish_tiled(c, tiles, names, lens, w, inc, flags, c0, c1, n_seq_out, total_len_out, cand_begin, cand_end);
}

int32_t frisk_seq_count(const frisk_ctx* c) {
    if (!c || !c->b().have_seq) return 0;
    return c->b().tiled ? int32_t(c->b().g_len.size()) : c->b().n_seq;
}
const char* frisk_seq_name(const frisk_ctx* c, int32_t s) {
    if (!c || !c->b().have_seq || s < 0) return "";
    if (c->b().tiled) return size_t(s) < c->b().g_name.size() ? c->b().g_name[size_t(s)].c_str() : "";
    return s < c->b().n_seq ? c->b().seq_name[size_t(s)].c_str() : "";
}
int64_t frisk_seq_len(const frisk_ctx* c, int32_t s) {
    if (!c || !c->b().have_seq || s < 0) return -1;
    if (c->b().tiled) return size_t(s) < c->b().g_len.size() ? c->b().g_len[size_t(s)] : -1;
    return s < c->b().n_seq ? c->b().seq_len[size_t(s)] : -1;
}

// ---- double-buffered residency: upload the NEXT batch while the resident one is being profiled / scanned ----------
int frisk_seq_stage(frisk_ctx* c, const uint8_t* const* seqs, const int64_t* lens, int32_t n_seq) {
    if (!c) return FRISK_E_ARG;
    if (n_seq > 0 && (!seqs || !lens)) return fail(c, FRISK_E_ARG, "null sequence table");
    HIPC(c, hipSetDevice(c->device));
    frisk_ctx::Batch& B = c->bat[c->cur ^ 1];
    c->staged = false;
    // work queued on the compute stream before the last commit may still be reading this slot
    if (c->slot_ev_set) HIPC(c, hipStreamWaitEvent(c->copy_stream, c->slot_free_ev, 0));
    int rc = layout_batch(c, B, lens, n_seq);
    if (rc) return rc;
    rc = enqueue_ascii_upload(c, B, seqs, lens, n_seq, c->copy_stream);
    if (rc) return rc;
    rc = alloc_packed(c, B, c->copy_stream);
    if (rc) return rc;
    rc = enqueue_pack(c, B, c->copy_stream);
    if (rc) return rc;
    HIPC(c, hipEventRecord(c->staged_ev, c->copy_stream));
    c->staged = true;
    return FRISK_OK;
}

int frisk_seq_stage_packed(frisk_ctx* c, const uint32_t* codes, const uint32_t* inv, const uint32_t* low, const int64_t* lens,
                           int32_t n_seq) {
    if (!c) return FRISK_E_ARG;
    if (!codes || !inv || !low || (n_seq > 0 && !lens)) return fail(c, FRISK_E_ARG, "null packed arrays");
    HIPC(c, hipSetDevice(c->device));
    frisk_ctx::Batch& B = c->bat[c->cur ^ 1];
    c->staged = false;
    if (c->slot_ev_set) HIPC(c, hipStreamWaitEvent(c->copy_stream, c->slot_free_ev, 0));      // (as frisk_seq_stage)
    int rc = layout_batch(c, B, lens, n_seq);
    if (rc) return rc;
    rc = alloc_packed(c, B, c->copy_stream);
    if (rc) return rc;
    const size_t w32 = size_t(B.padded_len / 32);
    // (h2d: page-locked sources asynchronously; pageable ones - the memory-mapped sequence cache - through the context's
    //  page-locked pieces at PCIe rate instead of the runtime's own staging)
    rc = h2d(c, B.d_codes.p, codes, 2 * w32 * 4, c->copy_stream);
    if (rc) return rc;
    rc = h2d(c, B.d_inv.p, inv, w32 * 4, c->copy_stream);
    if (rc) return rc;
    rc = h2d(c, B.d_low.p, low, w32 * 4, c->copy_stream);
    if (rc) return rc;
    HIPC(c, hipEventRecord(c->staged_ev, c->copy_stream));
    c->staged = true;
    return FRISK_OK;
}

// ---- the 0.25 B/base upload form: 2-bit codes densely, the two masks as run lists, the codes in pieces --------------------
int64_t frisk_padded_len_of(const int64_t* lens, int32_t n_seq) {
    if (n_seq < 0 || (n_seq > 0 && !lens)) return -1;
    for (int32_t s = 0; s < n_seq; ++s) if (lens[s] < 0) return -1;
    return frisk_pack2::padded_len(lens, n_seq);
}

int frisk_pack_2bit(const uint8_t* const* seqs, const int64_t* lens, int32_t n_seq, uint32_t* codes, int64_t** inv_runs,
                    int64_t* n_inv, int64_t** low_runs, int64_t* n_low) {
    if (n_seq < 0 || (n_seq > 0 && (!seqs || !lens)) || !codes || !inv_runs || !n_inv || !low_runs || !n_low) return FRISK_E_ARG;
    for (int32_t s = 0; s < n_seq; ++s) if (lens[s] < 0 || (lens[s] > 0 && !seqs[s])) return FRISK_E_ARG;
    frisk_pack2::Runs R;
    const unsigned hw = std::thread::hardware_concurrency();
    frisk_pack2::pack_batch(seqs, lens, n_seq, codes, R, int(std::min(32u, hw ? hw : 1u)));
    auto give = [](const std::vector<int64_t>& v, int64_t** out, int64_t* n) -> bool {
        *n = int64_t(v.size() / 2);
        *out = static_cast<int64_t*>(std::malloc(std::max<size_t>(v.size(), 2) * sizeof(int64_t)));
        if (!*out) return false;
        if (!v.empty()) std::memcpy(*out, v.data(), v.size() * sizeof(int64_t));
        return true;
    };
    if (!give(R.inv, inv_runs, n_inv)) return FRISK_E_HIP;
    if (!give(R.low, low_runs, n_low)) { std::free(*inv_runs); *inv_runs = nullptr; return FRISK_E_HIP; }
    return FRISK_OK;
}

// one mask of a streamed batch: zero the bitmap, then the caller's run list (or the caller's dense bitmap), then the PADs
static int enqueue_mask(frisk_ctx* c, frisk_ctx::Batch& B, uint32_t* d_bits, const int64_t* runs, int64_t n_runs, int64_t* d_runs,
                        const int64_t* d_pads, int64_t n_pads, hipStream_t st) {
    const size_t w32 = size_t(B.padded_len / 32);
    if (n_runs < 0) {               // dense: P / 32 words in the library's layout (real bases only; PADs are added here)
        int rc = h2d(c, d_bits, runs, w32 * 4, st);
        if (rc) return rc;
    } else if (size_t(n_runs) * 16 > w32 * 4 + (size_t(1) << 20)) {
        // more bytes of runs than of bitmap (a sequence that changes case every few bases): expand here, upload densely
        std::vector<uint32_t> bits(w32, 0u);
        for (int64_t r = 0; r < n_runs; ++r)
            for (int64_t p = runs[2 * r]; p < runs[2 * r + 1]; ++p) bits[size_t(p >> 5)] |= 0x80000000u >> (p & 31);
        int rc = h2d(c, d_bits, bits.data(), w32 * 4, st);
        if (rc) return rc;
        HIPC(c, hipStreamSynchronize(st));              // (`bits` is a local)
    } else {
        HIPC(c, hipMemsetAsync(d_bits, 0, w32 * 4, st));
        if (n_runs > 0) {
            int rc = h2d(c, d_runs, runs, size_t(n_runs) * 16, st);
            if (rc) return rc;
            expand_runs_kernel<<<grid_for(n_runs, 4, c->num_cu * 16), 256, 0, st>>>(d_runs, n_runs, d_bits);
            HIPC(c, hipGetLastError());
        }
    }
    expand_runs_kernel<<<grid_for(n_pads, 4, c->num_cu * 16), 256, 0, st>>>(d_pads, n_pads, d_bits);
    HIPC(c, hipGetLastError());
    return FRISK_OK;
}

// the 0.25 B/base form of batch B (laid out by the caller) onto the device, on stream `st`: masks from the run lists, the codes
// in pieces with an event behind each
static int upload_2bit(frisk_ctx* c, frisk_ctx::Batch& B, const uint32_t* codes, const int64_t* inv_runs, int64_t n_inv,
                       const int64_t* low_runs, int64_t n_low, int64_t piece_bases, hipStream_t st) {
    const int64_t P = B.padded_len;
    const int32_t n_seq = B.n_seq;
    for (const auto& L : {std::make_pair(inv_runs, n_inv), std::make_pair(low_runs, n_low)})
        for (int64_t r = 0; r < L.second; ++r)
            if (L.first[2 * r] < 0 || L.first[2 * r] > L.first[2 * r + 1] || L.first[2 * r + 1] > P)
                return fail(c, FRISK_E_ARG, "frisk_seq_stage_2bit: a run outside the batch");
    int rc = alloc_packed(c, B, st);
    if (rc) return rc;
    // PAD runs: the position behind every scaffold, and the batch's tail (inv AND low, frisk_device.h); adjacent ones merged
    // (the page-locked buffer they travel from may still feed the copy of this slot's previous upload: wait for that one)
    if (!B.piece_end.empty() && !B.piece_ev.empty()) HIPC(c, hipEventSynchronize(B.piece_ev[std::min(B.piece_end.size(), B.piece_ev.size()) - 1]));
    std::vector<int64_t> pads;
    for (int32_t s = 0; s < n_seq; ++s) frisk_pack2::push_run(pads, B.seq_off[size_t(s)] + B.seq_len[size_t(s)], B.seq_off[size_t(s)] + B.seq_len[size_t(s)] + 1);
    {
        const int64_t tail = n_seq > 0 ? B.seq_off[size_t(n_seq) - 1] + B.seq_len[size_t(n_seq) - 1] + 1 : 0;
        if (tail < P) frisk_pack2::push_run(pads, tail, P);
    }
    const int64_t n_pads = int64_t(pads.size() / 2);
    const size_t need = 2 * size_t(std::max<int64_t>(n_inv, 0) + std::max<int64_t>(n_low, 0) + n_pads) + 8;
    HIPC(c, B.d_runs.reserve(need));
    int64_t* d_pads = B.d_runs.p;
    int64_t* d_inv_runs = d_pads + 2 * n_pads;
    int64_t* d_low_runs = d_inv_runs + 2 * std::max<int64_t>(n_inv, 0);
    if (B.h_pads_cap < pads.size()) {
        if (B.h_pads) HIPC(c, hipHostFree(B.h_pads));
        B.h_pads = nullptr; B.h_pads_cap = 0;
        HIPC(c, hipHostMalloc(reinterpret_cast<void**>(&B.h_pads), (pads.size() + pads.size() / 4 + 16) * sizeof(int64_t), hipHostMallocDefault));
        B.h_pads_cap = pads.size() + pads.size() / 4 + 16;
    }
    std::memcpy(B.h_pads, pads.data(), pads.size() * sizeof(int64_t));
    HIPC(c, hipMemcpyAsync(d_pads, B.h_pads, pads.size() * sizeof(int64_t), hipMemcpyHostToDevice, st));
    rc = enqueue_mask(c, B, B.d_inv.p, inv_runs, n_inv, d_inv_runs, d_pads, n_pads, st);
    if (rc) return rc;
    rc = enqueue_mask(c, B, B.d_low.p, low_runs, n_low, d_low_runs, d_pads, n_pads, st);
    if (rc) return rc;
    // the codes, piece by piece: an event behind each, so that phase A can follow the copies (frisk_profile_add)
    const int64_t w32 = P / 32;
    // default: 256 Mbases = 64 MB of codes (measured on the whole C5 shape, ms per streamed job: pieces of 8 MB 71.6, 16 MB 69.2,
    // 32 MB 67.5, 64 MB 66.6, 128 MB 66.5 - the copy engine leaves ~20 us between two copies, the last piece's profile kernel
    // is what follows the upload: tools/exp/piece_sweep.py)
    int64_t piece_words = (piece_bases ? piece_bases : (int64_t(1) << 28)) / 32;
    if (piece_words < 1) piece_words = 1;
    const int64_t n_pieces = std::max<int64_t>(1, (w32 + piece_words - 1) / piece_words);
    if (n_pieces > 4096) piece_words = (w32 + 4095) / 4096;
    B.piece_end.clear();
    for (int64_t a = 0; a < w32; a += piece_words) {
        const int64_t b = std::min(w32, a + piece_words);
        rc = h2d(c, B.d_codes.p + 2 * a, codes + 2 * a, size_t(b - a) * 8, st);
        if (rc) return rc;
        const size_t i = B.piece_end.size();
        if (B.piece_ev.size() <= i) {
            hipEvent_t ev = nullptr;
            HIPC(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            B.piece_ev.push_back(ev);
        }
        HIPC(c, hipEventRecord(B.piece_ev[i], st));
        B.piece_end.push_back(b);
    }
    return FRISK_OK;
}

int frisk_seq_stage_2bit(frisk_ctx* c, const uint32_t* codes, const int64_t* inv_runs, int64_t n_inv, const int64_t* low_runs,
                         int64_t n_low, const int64_t* lens, int32_t n_seq, int64_t piece_bases) {
    if (!c) return FRISK_E_ARG;
    if (!codes || (n_seq > 0 && !lens) || (n_inv != 0 && !inv_runs) || (n_low != 0 && !low_runs))
        return fail(c, FRISK_E_ARG, "frisk_seq_stage_2bit: null array");
    if (piece_bases < 0) return fail(c, FRISK_E_ARG, "frisk_seq_stage_2bit: piece_bases < 0");
    HIPC(c, hipSetDevice(c->device));
    frisk_ctx::Batch& B = c->bat[c->cur ^ 1];
    c->staged = false;
    if (c->slot_ev_set) HIPC(c, hipStreamWaitEvent(c->copy_stream, c->slot_free_ev, 0));      // (as frisk_seq_stage)
    int rc = layout_batch(c, B, lens, n_seq);
    if (rc) return rc;
    rc = upload_2bit(c, B, codes, inv_runs, n_inv, low_runs, n_low, piece_bases, c->copy_stream);
    if (rc) return rc;
    HIPC(c, hipEventRecord(c->staged_ev, c->copy_stream));
    B.streaming = true;
    c->staged = true;
    return FRISK_OK;
}

int frisk_seq_commit(frisk_ctx* c) {
    if (!c) return FRISK_E_ARG;
    if (!c->staged) return fail(c, FRISK_E_STATE, "frisk_seq_commit: no staged batch");
    HIPC(c, hipSetDevice(c->device));
    // the slot that stops being resident here is the one the next stage fills: that upload waits for what is queued so far
    HIPC(c, hipEventRecord(c->slot_free_ev, c->stream));
    c->slot_ev_set = true;
    // the compute stream waits on the device; the host does not.  A streamed batch is waited for piece by piece, by its first user
    if (!c->bat[c->cur ^ 1].streaming) HIPC(c, hipStreamWaitEvent(c->stream, c->staged_ev, 0));
    c->cur ^= 1;
    c->b().have_seq = true;
    c->staged = false;
    c->plan_w = -1;
    return FRISK_OK;
}

int frisk_seq_export_2bit(frisk_ctx* c, uint32_t* codes, int64_t** inv_runs, int64_t* n_inv, int64_t** low_runs, int64_t* n_low) {
    if (!c || !codes || !inv_runs || !n_inv || !low_runs || !n_low) return FRISK_E_ARG;
    if (!c->b().have_seq) return fail(c, FRISK_E_STATE, "no resident sequence batch");
    auto give = [](const std::vector<int64_t>& v, int64_t** out, int64_t* n) -> bool {
        *n = int64_t(v.size() / 2);
        *out = static_cast<int64_t*>(std::malloc(std::max<size_t>(v.size(), 2) * sizeof(int64_t)));
        if (!*out) return false;
        if (!v.empty()) std::memcpy(*out, v.data(), v.size() * sizeof(int64_t));
        return true;
    };
    if (c->b().have_host2) {            // packed on the host by frisk_fasta_load: the device is not involved
        const frisk_ctx::Batch& HB = c->b();
        const size_t nbytes = HB.h_codes.size() * 4;
        const int T = int(std::max<size_t>(1, std::min<size_t>(8, nbytes >> 26)));
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) {
            const size_t a = nbytes / 64 * size_t(t) / size_t(T) * 64, b = t + 1 == T ? nbytes : nbytes / 64 * size_t(t + 1) / size_t(T) * 64;
            th.emplace_back([&, a, b] { std::memcpy(reinterpret_cast<char*>(codes) + a, reinterpret_cast<const char*>(HB.h_codes.data()) + a, b - a); });
        }
        for (auto& x : th) x.join();
        if (!give(HB.h_runs.inv, inv_runs, n_inv)) return fail(c, FRISK_E_HIP, "out of host memory");
        if (!give(HB.h_runs.low, low_runs, n_low)) { std::free(*inv_runs); *inv_runs = nullptr; return fail(c, FRISK_E_HIP, "out of host memory"); }
        return FRISK_OK;
    }
    HIPC(c, hipSetDevice(c->device));
    int rc = settle_stream(c);
    if (rc) return rc;
    const frisk_ctx::Batch& B = c->b();
    const size_t w32 = size_t(B.padded_len / 32);
    std::vector<uint32_t> inv(w32), low(w32);
    HIPC(c, hipMemcpyAsync(codes, B.d_codes.p, 2 * w32 * 4, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipMemcpyAsync(inv.data(), B.d_inv.p, w32 * 4, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipMemcpyAsync(low.data(), B.d_low.p, w32 * 4, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    std::vector<int64_t> ri, rl;
    std::thread t([&] { frisk_pack2::bitmap_runs(inv.data(), B.seq_len.data(), B.n_seq, ri); });
    frisk_pack2::bitmap_runs(low.data(), B.seq_len.data(), B.n_seq, rl);
    t.join();
    if (!give(ri, inv_runs, n_inv)) return fail(c, FRISK_E_HIP, "out of host memory");
    if (!give(rl, low_runs, n_low)) { std::free(*inv_runs); *inv_runs = nullptr; return fail(c, FRISK_E_HIP, "out of host memory"); }
    return FRISK_OK;
}

int frisk_seq_export_packed(frisk_ctx* c, uint32_t* codes, uint32_t* inv, uint32_t* low) {
    if (!c || !codes || !inv || !low) return FRISK_E_ARG;
    if (!c->b().have_seq) return fail(c, FRISK_E_STATE, "no resident sequence batch");
    HIPC(c, hipSetDevice(c->device));
    if (int rs = settle_stream(c)) return rs;
    const size_t w32 = size_t(c->b().padded_len / 32);
    HIPC(c, hipMemcpyAsync(codes, c->b().d_codes.p, 2 * w32 * 4, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipMemcpyAsync(inv, c->b().d_inv.p, w32 * 4, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipMemcpyAsync(low, c->b().d_low.p, w32 * 4, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    return FRISK_OK;
}

int frisk_seq_set_names(frisk_ctx* c, const char* const* names, int32_t n_seq) {
    if (!c || (n_seq > 0 && !names)) return FRISK_E_ARG;
    if (!c->b().have_seq || n_seq != c->b().n_seq) return fail(c, FRISK_E_ARG, "frisk_seq_set_names: one name per resident scaffold");
    for (int32_t s = 0; s < n_seq; ++s) c->b().seq_name[size_t(s)] = names[s] ? names[s] : "";
    return FRISK_OK;
}

int frisk_seq_synth(frisk_ctx* c, const int64_t* lens, int32_t n_seq, uint64_t seed, double island_frac,
                    double n_frac, double lower_frac, double repeats_per_kb) {
    return frisk_seq_synth2(c, lens, n_seq, seed, island_frac, n_frac, lower_frac, repeats_per_kb, 0.0, 0.0);
}

int frisk_seq_synth2(frisk_ctx* c, const int64_t* lens, int32_t n_seq, uint64_t seed, double island_frac,
                     double n_frac, double lower_frac, double repeats_per_kb, double period_mix, double sat_frac) {
    if (!c) return FRISK_E_ARG;
    if (n_seq > 0 && !lens) return fail(c, FRISK_E_ARG, "null length table");
    HIPC(c, hipSetDevice(c->device));
    int rc = layout_batch(c, lens, n_seq);
    if (rc) return rc;
    HIPC(c, c->b().d_ascii.reserve(size_t(c->b().padded_len)));
    HIPC(c, hipMemsetAsync(c->b().d_ascii.p, FRISK_PAD_BYTE, size_t(c->b().padded_len), c->stream));
    SynthTables tabs;
    synth_make_tables(seed, tabs);
    const uint32_t thr_island = synth_frac_to_u32(island_frac), thr_nbig = synth_frac_to_u32(n_frac * 0.8),
                   thr_nsmall = synth_frac_to_u32(n_frac * 0.2), thr_low = synth_frac_to_u32(lower_frac),
                   thr_rep = synth_frac_to_u32(repeats_per_kb * (SYNTH_REP / 1000.0)), thr_mix = synth_frac_to_u32(period_mix),
                   thr_sat = synth_frac_to_u32(sat_frac * (2.0 * SYNTH_SAT_GROUP / (SYNTH_SAT_GROUP + 1.0))),
                   thr_div = synth_frac_to_u32(SYNTH_SAT_DIV);
    for (int32_t s = 0; s < n_seq; ++s) {
        if (lens[s] <= 0) continue;
        const int64_t nblk = (lens[s] + SYNTH_BLOCK - 1) / SYNTH_BLOCK;
        synth_kernel<<<grid_for(nblk, 64, 1 << 20), 64, 0, c->stream>>>(c->b().d_ascii.p + c->b().seq_off[size_t(s)], lens[s],
                                                                       seed, uint32_t(s), tabs, thr_island, thr_nbig,
                                                                       thr_nsmall, thr_low, thr_rep, thr_mix, thr_sat, thr_div);
        HIPC(c, hipGetLastError());
    }
    rc = alloc_packed(c);
    if (rc) return rc;
    return run_pack(c);
}

int frisk_seq_read(frisk_ctx* c, int32_t s, int64_t offset, int64_t n, uint8_t* out) {
    if (!c || !out) return FRISK_E_ARG;
    if (!c->b().have_seq) return fail(c, FRISK_E_STATE, "no resident sequence batch");
    if (c->b().tiled) return fail(c, FRISK_E_STATE, "frisk_seq_read: the resident batch holds tiles, not whole scaffolds");
    if (s < 0 || s >= c->b().n_seq) return fail(c, FRISK_E_ARG, "sequence index out of range");
    if (offset < 0 || n < 0 || offset + n > c->b().seq_len[size_t(s)]) return fail(c, FRISK_E_ARG, "range outside the scaffold");
    if (n == 0) return FRISK_OK;
    HIPC(c, hipSetDevice(c->device));
    if (int rs = settle_stream(c)) return rs;
    DevBuf<uint8_t> tmp;
    HIPC(c, tmp.reserve(size_t(n)));
    unpack_kernel<<<grid_for(n, 256, c->num_cu * 8), 256, 0, c->stream>>>(c->b().d_codes.p, c->b().d_inv.p, c->b().d_low.p,
                                                                           c->b().seq_off[size_t(s)] + offset, n, tmp.p);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, tmp.p, size_t(n), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    tmp.release();
    HIPC(c, e);
    return FRISK_OK;
}

// --------------------------------------------------------------------------------------- phase A
int frisk_profile_reset(frisk_ctx* c) {
    if (!c) return FRISK_E_ARG;
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipMemsetAsync(c->d_raw.p, 0, (size_t(c->nprof) + 4) * sizeof(int64_t), c->stream));
    c->profile_final = false;
    return FRISK_OK;
}

static int profile_add_range(frisk_ctx* c, int mask_host, int64_t p0, int64_t p1, bool first = true, bool last = true);

int frisk_profile_add(frisk_ctx* c, int mask_host, int64_t p0, int64_t p1) {
    if (!c) return FRISK_E_ARG;
    if (!c->b().have_seq) return fail(c, FRISK_E_STATE, "frisk_profile_add: no resident sequence batch");
    if (c->b().tiled) {
        // a tiled batch: "the whole batch" means the positions this rank OWNS (every base of the genome is owned by exactly
        // one rank; the K-1 bases kept behind an owned range belong to the words that start inside it)
        if (!(p0 < 0 && p1 < 0)) return fail(c, FRISK_E_ARG, "frisk_profile_add: a tiled batch is profiled as a whole (-1, -1)");
        const frisk_ctx::Batch& B = c->b();
        for (size_t t = 0; t < B.tiles.size(); ++t) {
            const frisk_ctx::Batch::Tile& T = B.tiles[t];
            if (T.own1 <= T.own0) continue;
            int rc = profile_add_range(c, mask_host, B.seq_off[t] + (T.own0 - T.base0), B.seq_off[t] + (T.own1 - T.base0));
            if (rc) return rc;
        }
        return FRISK_OK;
    }
    frisk_ctx::Batch& SB = c->b();
    if (SB.streaming && p0 < 0 && p1 < 0 && !SB.piece_end.empty()) {
        // a streamed batch counted as a whole: phase A follows the copies piece by piece - the kernel of piece i runs while piece
        // i + 1 crosses PCIe, only the last piece's kernel is left when the upload ends.  A lane that counts the positions of
        // bitmap word q reads word q + 1 of the masks and code word 2 q + 2, so piece i's kernel stops one word short of its end.
        HIPC(c, hipSetDevice(c->device));
        const size_t np = SB.piece_end.size();
        int64_t done = 0;
        for (size_t i = 0; i < np; ++i) {
            HIPC(c, hipStreamWaitEvent(c->stream, SB.piece_ev[i], 0));
            const int64_t upto = i + 1 == np ? SB.padded_len : (SB.piece_end[i] - 1) * 32;
            if (upto > done || i + 1 == np || i == 0) {
                int rc = profile_add_range(c, mask_host, done, std::max(done, upto), i == 0, i + 1 == np);
                if (rc) return rc;
                done = std::max(done, upto);
            }
        }
        SB.streaming = false;
        return FRISK_OK;
    }
    if (int rs = settle_stream(c)) return rs;
    return profile_add_range(c, mask_host, p0, p1);
}

static int profile_add_range(frisk_ctx* c, int mask_host, int64_t p0, int64_t p1, bool first, bool last) {
    if (p0 < 0 && p1 < 0) { p0 = 0; p1 = c->b().padded_len; }
    if (p0 < 0 || p1 > c->b().padded_len || p0 > p1) return fail(c, FRISK_E_ARG, "position range outside the batch");
    HIPC(c, hipSetDevice(c->device));
    c->profile_final = false;
    if (c->kmax > 8) {          // 4^K counters do not fit a CU: one global atomic per position (profile_kernels.h)
        const int64_t nw = p1 > p0 ? ((p1 + 31) >> 5) - (p0 >> 5) : 0;
        if (first) HIPC(c, hipEventRecord(c->evp0, c->stream));
        if (nw > 0)
            profile_add_big_kernel<<<grid_for(nw, 256, c->num_cu * 8), 256, 0, c->stream>>>(
                c->b().d_codes.p, c->b().d_inv.p, c->b().d_low.p, p0, p1, c->kmin, c->kmax, mask_host ? 1 : 0, int(c->nprof),
                reinterpret_cast<unsigned long long*>(c->d_raw.p));
        HIPC(c, hipGetLastError());
        if (last) { HIPC(c, hipEventRecord(c->evp1, c->stream)); c->ms_pending[1] = true; }
        return FRISK_OK;
    }
    // K = 8, a long range (a whole resident genome): 16-bit counters, ONE walk over the sequence (profile_add16_kernel; a wrapped
    // counter sends its workgroup to the two-half form).  The walk halves, the flush doubles (65 536 fields per workgroup instead of
    // 32 768 bins): measured 2.42 -> 1.33 ms on the 3.29 Gb shape, 0.345 -> 0.55 ms on the 410 Mb shard (tools/exp/prof_ab.py) - so from
    // 2^30 positions per launch on.  (mask_host bit 1 = FRISK_PROFILE_ONE_PASS forces it: the tests' way to the overflow path.)
#ifndef FRISK_PROF16
#define FRISK_PROF16 1
#endif
    const bool one_pass = c->kmax == 8 && FRISK_PROF16 && ((mask_host & 2) || p1 - p0 >= (int64_t(1) << 30));
    mask_host &= 1;
    if (one_pass) {
        const int64_t span16 = p1 - p0;
        const int64_t nwords16 = span16 > 0 ? ((p1 + 31) >> 5) - (p0 >> 5) : 0;
        int64_t nch = std::min<int64_t>(std::max<int64_t>(1, span16 / 65536), int64_t(c->num_cu));
        const int64_t chunk_len16 = (nwords16 + nch - 1) / std::max<int64_t>(nch, 1);
        nch = chunk_len16 > 0 ? (nwords16 + chunk_len16 - 1) / chunk_len16 : 0;
        if (first) HIPC(c, hipEventRecord(c->evp0, c->stream));
        if (span16 > 0) {
            HIPC(c, hipFuncSetAttribute(reinterpret_cast<const void*>(profile_add16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
            profile_add16_kernel<<<int(nch), FRISK_PROF_NT, 131072, c->stream>>>(c->b().d_codes.p, c->b().d_inv.p, c->b().d_low.p, p0, p1, c->kmin,
                                                                                 mask_host ? 1 : 0, int(c->nprof), chunk_len16,
                                                                                 reinterpret_cast<unsigned long long*>(c->d_raw.p));
        }
        HIPC(c, hipGetLastError());
        if (last) { HIPC(c, hipEventRecord(c->evp1, c->stream)); c->ms_pending[1] = true; }
        return FRISK_OK;
    }
    // order-K table privatised in LDS (u32): split in two halves at K = 8 (256 KiB does not fit a CU)
    const int halves = (c->kmax == 8) ? 2 : 1;
    const size_t lds = (size_t(1) << (2 * c->kmax)) / size_t(halves) * 4;
    const int64_t span = p1 - p0;
    const int64_t nwords = span > 0 ? ((p1 + 31) >> 5) - (p0 >> 5) : 0;     // a lane takes one 32-position bitmap word
#ifndef FRISK_PROF_WG_PER_CU
#define FRISK_PROF_WG_PER_CU 1      // measured on the 410 Mb shard: 1 -> 0.34 ms, 2 -> 0.41, 4 -> 0.51 (the flush of the private tables dominates)
#endif
    int64_t nchunks = std::min<int64_t>(std::max<int64_t>(1, span / 65536), int64_t(c->num_cu) * FRISK_PROF_WG_PER_CU / halves);
    const int64_t chunk_len = (nwords + nchunks - 1) / std::max<int64_t>(nchunks, 1);   // in words
    nchunks = chunk_len > 0 ? (nwords + chunk_len - 1) / chunk_len : 0;
    if (first) HIPC(c, hipEventRecord(c->evp0, c->stream));
    if (span > 0) {
        auto raw = reinterpret_cast<unsigned long long*>(c->d_raw.p);
        HIPC(c, hipFuncSetAttribute(reinterpret_cast<const void*>(profile_add_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, int(std::max<size_t>(lds, 16))));
        profile_add_kernel<<<int(nchunks) * halves, FRISK_PROF_NT, std::max<size_t>(lds, 16), c->stream>>>(
            c->b().d_codes.p, c->b().d_inv.p, c->b().d_low.p, p0, p1, c->kmin, c->kmax, mask_host ? 1 : 0, int(c->nprof), halves,
            chunk_len, raw);
    }
    HIPC(c, hipGetLastError());
    if (last) {
        HIPC(c, hipEventRecord(c->evp1, c->stream));
        c->ms_pending[1] = true;    // asynchronous: the elapsed time is read when frisk_last_kernel_ms(1) asks for it
    }
    return FRISK_OK;
}

int frisk_profile_export_device(frisk_ctx* c, void* dst) {
    if (!c || !dst) return FRISK_E_ARG;
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipMemcpyAsync(dst, c->d_raw.p, (size_t(c->nprof) + 4) * 8, hipMemcpyDeviceToDevice, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    return FRISK_OK;
}
int frisk_profile_import_device(frisk_ctx* c, const void* src) {
    if (!c || !src) return FRISK_E_ARG;
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipMemcpyAsync(c->d_raw.p, src, (size_t(c->nprof) + 4) * 8, hipMemcpyDeviceToDevice, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    c->profile_final = false;
    return FRISK_OK;
}
int frisk_profile_device_view(frisk_ctx* c, void** raw, void** stream) {
    if (!c || !raw || !stream) return FRISK_E_ARG;
    *raw = c->d_raw.p;
    *stream = reinterpret_cast<void*>(c->stream);
    c->profile_final = false;
    return FRISK_OK;
}
// The one collective of a multi-GPU job, without torch: RCCL's all-reduce(sum, int64) IN PLACE on the raw profile, on the context's
// stream.  librccl is not linked: the copy already in the process (PyTorch-ROCm brings its own under the same SONAME) or the
// system one is resolved on first use, so that a one-GPU process never loads it.
int frisk_profile_allreduce(frisk_ctx* c, void* rccl_comm) {
    if (!c) return FRISK_E_ARG;
    if (!rccl_comm) return FRISK_OK;                    // one GPU: nothing to sum
    typedef int (*allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
    typedef const char* (*errstr_fn)(int);
    static allreduce_fn all_reduce = nullptr;
    static errstr_fn err_string = nullptr;
    if (!all_reduce) {
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);           // the instance the caller's communicator came from
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return fail(c, FRISK_E_HIP, std::string("frisk_profile_allreduce: librccl not found: ") + dlerror());
        all_reduce = reinterpret_cast<allreduce_fn>(dlsym(h, "ncclAllReduce"));
        err_string = reinterpret_cast<errstr_fn>(dlsym(h, "ncclGetErrorString"));
        if (!all_reduce) return fail(c, FRISK_E_HIP, "frisk_profile_allreduce: ncclAllReduce not found in librccl");
    }
    HIPC(c, hipSetDevice(c->device));
    const int rc = all_reduce(c->d_raw.p, c->d_raw.p, size_t(c->nprof) + 4, /* ncclInt64 */ 4, /* ncclSum */ 0, rccl_comm, c->stream);
    if (rc != 0) return fail(c, FRISK_E_HIP, std::string("ncclAllReduce: ") + (err_string ? err_string(rc) : "error"));
    c->profile_final = false;
    return FRISK_OK;
}

int frisk_profile_export_host(frisk_ctx* c, int64_t* dst) {
    if (!c || !dst) return FRISK_E_ARG;
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipMemcpyAsync(dst, c->d_raw.p, (size_t(c->nprof) + 4) * 8, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    return FRISK_OK;
}
int frisk_profile_import_host(frisk_ctx* c, const int64_t* src) {
    if (!c || !src) return FRISK_E_ARG;
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipMemcpyAsync(c->d_raw.p, src, (size_t(c->nprof) + 4) * 8, hipMemcpyHostToDevice, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    c->profile_final = false;
    return FRISK_OK;
}

int frisk_profile_finalize(frisk_ctx* c) {
    if (!c) return FRISK_E_ARG;
    HIPC(c, hipSetDevice(c->device));
    const size_t nraw = size_t(c->nprof) + 4;
    HIPC(c, hipMemcpyAsync(c->d_cnt.p, c->d_raw.p, nraw * 8, hipMemcpyDeviceToDevice, c->stream));
    int x = c->kmax - 1;
    for (; x >= c->kmin && x > 7; --x) {                // the wide levels (K > 8): one launch per order
        const int64_t n = int64_t(1) << (2 * x);
        marginalize_kernel<<<grid_for(n, 256, 1 << 20), 256, 0, c->stream>>>(c->d_cnt.p, c->kmin, x);
    }
    if (x >= c->kmin) marginalize_low_kernel<<<1, 1024, 0, c->stream>>>(c->d_cnt.p, c->kmin, x);       // orders <= 7: one launch
    symmetrize_kernel<<<grid_for(c->nprof, 256, 1 << 20), 256, 0, c->stream>>>(c->d_cnt.p, c->d_sym.p, c->kmin, c->kmax);
    HIPC(c, hipGetLastError());
    // metadata of L356-359 (totalLen, exMax, nnTotal): computed on the device, mirrored to the host asynchronously
    profile_meta_kernel<<<1, 1024, 0, c->stream>>>(c->d_cnt.p, c->d_raw.p + c->nprof, c->kmin, c->kmax, c->d_meta.p);
    HIPC(c, hipGetLastError());
    return build_genome_table(c);
}

int frisk_profile_get(frisk_ctx* c, int64_t* sym, int64_t* total_len, int64_t* ex_max, int64_t* nn_total) {
    if (!c) return FRISK_E_ARG;
    if (!c->profile_final) return fail(c, FRISK_E_STATE, "profile not finalised");
    HIPC(c, hipSetDevice(c->device));
    if (sym) HIPC(c, hipMemcpyAsync(sym, c->d_sym.p, size_t(c->nprof) * 8, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    if (total_len) *total_len = c->h_meta[0];
    if (ex_max) *ex_max = c->h_meta[1];
    if (nn_total) *nn_total = c->h_meta[2];
    return FRISK_OK;
}

int frisk_profile_set(frisk_ctx* c, const int64_t* sym, int64_t total_len, int64_t ex_max, int64_t nn_total) {
    if (!c || !sym) return FRISK_E_ARG;
    HIPC(c, hipSetDevice(c->device));
    HIPC(c, hipStreamSynchronize(c->stream));                   // h_meta may still be the target of an earlier mirror copy
    HIPC(c, hipMemcpyAsync(c->d_sym.p, sym, size_t(c->nprof) * 8, hipMemcpyHostToDevice, c->stream));
    c->h_meta[0] = total_len; c->h_meta[1] = ex_max; c->h_meta[2] = nn_total;
    HIPC(c, hipMemcpyAsync(c->d_meta.p, c->h_meta, 3 * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));                   // `sym` is the caller's: the copy must have left it
    return build_genome_table(c);
}

// --------------------------------------------------------------------------------------- phase B
int frisk_scan_plan(frisk_ctx* c, int32_t w, int32_t inc, uint32_t flags, int64_t* n_candidates) {
    if (!c) return FRISK_E_ARG;
    if (!c->b().have_seq) return fail(c, FRISK_E_STATE, "frisk_scan_plan: no resident sequence batch");
    if (w < 1 || inc < 1) return fail(c, FRISK_E_ARG, "window length and increment must be >= 1");
    const bool all = (flags & FRISK_SCAN_SCAFFOLDS_ALL) != 0;
    if (c->plan_w == w && c->plan_inc == inc && ((c->plan_flags ^ flags) & FRISK_SCAN_SCAFFOLDS_ALL) == 0) {
        if (n_candidates) *n_candidates = c->plan_ncand;
        return FRISK_OK;
    }
    HIPC(c, hipSetDevice(c->device));
    const frisk_ctx::Batch& B = c->b();
    if (B.tiled && (w != B.tile_w || inc != B.tile_inc || (flags & FRISK_SCAN_SCAFFOLDS_ALL) != B.tile_flags))
        return fail(c, FRISK_E_ARG, "the resident batch holds the tiles of another window geometry (frisk_fasta_load_shard)");
    c->h_desc.assign(size_t(B.n_seq) + 1, ScafDesc());
    int64_t cand = 0, maxwin = 0;
    for (int32_t s = 0; s < B.n_seq; ++s) {
        ScafDesc& d = c->h_desc[size_t(s)];
        d.off = B.seq_off[size_t(s)];
        d.cand0 = cand;
        d.pad_ = 0;
        if (B.tiled) {              // candidates are numbered from the rank's first one; the windows were chosen at load time
            const frisk_ctx::Batch::Tile& t = B.tiles[size_t(s)];
            d.size = t.size; d.base0 = t.base0; d.j0 = t.j0; d.ncand = t.ncand; d.kind = t.kind;
        } else {
            d.size = B.seq_len[size_t(s)];
            d.base0 = 0; d.j0 = 0;
            plan_scaffold(d.size, w, inc, all, d.ncand, d.kind);
        }
        if (d.ncand > 0) maxwin = std::max<int64_t>(maxwin, d.kind == 1 ? d.size : w);
        cand += d.ncand;
    }
    ScafDesc& sentinel = c->h_desc[size_t(B.n_seq)];       // keeps the binary search in range
    sentinel.off = B.padded_len; sentinel.size = 0; sentinel.cand0 = cand; sentinel.ncand = 0; sentinel.kind = 0;
    sentinel.base0 = 0; sentinel.j0 = 0; sentinel.pad_ = 0;
    if (maxwin > 0x7FFFFFFF) return fail(c, FRISK_E_ARG, "a window longer than 2^31-1 bases");
    HIPC(c, c->d_desc.reserve(c->h_desc.size()));
    HIPC(c, hipMemcpyAsync(c->d_desc.p, c->h_desc.data(), c->h_desc.size() * sizeof(ScafDesc), hipMemcpyHostToDevice,
                           c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    c->plan_w = w; c->plan_inc = inc; c->plan_flags = flags; c->plan_ncand = cand; c->plan_maxwin = maxwin;
    if (n_candidates) *n_candidates = cand;
    return FRISK_OK;
}

static int scan_impl(frisk_ctx* c, int32_t w, int32_t inc, uint32_t flags, int64_t c0, int64_t c1, int64_t cap,
                     int32_t* seq_index, int64_t* start, int64_t* stop, uint32_t* status, double* kld, double* gc,
                     double* pi, double* si, double* cri, uint32_t* dbg_counts, int64_t* dbg_meta);

int frisk_scan(frisk_ctx* c, int32_t w, int32_t inc, uint32_t flags, int64_t c0, int64_t c1, int64_t cap,
               int32_t* seq_index, int64_t* start, int64_t* stop, uint32_t* status, double* kld, double* gc,
               double* pi, double* si, double* cri, uint32_t* dbg_counts, int64_t* dbg_meta) {
    const int rc = scan_impl(c, w, inc, flags, c0, c1, cap, seq_index, start, stop, status, kld, gc, pi, si, cri, dbg_counts, dbg_meta);
    if (rc != FRISK_OK && c) {
        // a failure after work was queued: copies may still target the caller's row buffers (and locals of the call), kernels of
        // the tail segment may still run - nothing of this call is in flight once it has returned
        const std::string msg = c->err;
        if (hipSetDevice(c->device) == hipSuccess) {
            (void)hipStreamSynchronize(c->stream);
            (void)hipStreamSynchronize(c->tail_stream);
            (void)hipGetLastError();
        }
        c->err = msg;
    }
    return rc;
}

static int scan_impl(frisk_ctx* c, int32_t w, int32_t inc, uint32_t flags, int64_t c0, int64_t c1, int64_t cap,
                     int32_t* seq_index, int64_t* start, int64_t* stop, uint32_t* status, double* kld, double* gc,
                     double* pi, double* si, double* cri, uint32_t* dbg_counts, int64_t* dbg_meta) {
    if (!c) return FRISK_E_ARG;
    if (!c->profile_final) return fail(c, FRISK_E_STATE, "frisk_scan: genome profile not finalised");
    int64_t ncand_all = 0;
    int rc = frisk_scan_plan(c, w, inc, flags, &ncand_all);
    if (rc) return rc;
    if (c1 < 0) c1 = ncand_all;
    if (c0 < 0 || c0 > c1 || c1 > ncand_all) return fail(c, FRISK_E_ARG, "candidate range outside [0, n_candidates]");
    const int64_t n = c1 - c0;
    if (cap < n) return fail(c, FRISK_E_CAP, "output capacity too small: need " + std::to_string(n));
    const bool rip = (flags & FRISK_SCAN_RIP) != 0;
    if (rip && !(c->kmin <= 2 && c->kmax >= 2))
        return fail(c, FRISK_E_ARG, "RIP indices need dinucleotide counts: kmin <= 2 <= kmax (reference L478)");
    if (rip && (!pi || !si || !cri)) return fail(c, FRISK_E_ARG, "FRISK_SCAN_RIP needs pi/si/cri buffers");
    if (!seq_index || !start || !stop || !status || !kld || !gc) return fail(c, FRISK_E_ARG, "null output buffer");
    c->ms[0] = 0.0;
    if (n == 0) return FRISK_OK;
    HIPC(c, hipSetDevice(c->device));
    if (int rs = settle_stream(c)) return rs;
    const bool debug = dbg_counts || dbg_meta;
    const size_t N = size_t(n);
    // short scans: the row columns as consecutive pieces of one block - [start | stop | kld | gc | pi si cri | seq_index | status] - so
    // that one copy brings them to the host
    const bool packed_rows = n < (int64_t(1) << 17);
    const size_t Np = (N + 1) / 2 * 2;                          // (the two 4-byte columns end on a multiple of 8 bytes)
    const size_t blk_words = packed_rows ? (4 + (rip ? 3 : 0)) * Np + Np : 0;      // in doubles
    if (packed_rows) {
        HIPC(c, c->o_block.reserve(blk_words));
        if (c->h_block_cap < blk_words * 8) {
            if (c->h_block) HIPC(c, hipHostFree(c->h_block));
            c->h_block = nullptr; c->h_block_cap = 0;
            HIPC(c, hipHostMalloc(&c->h_block, blk_words * 8 + blk_words, hipHostMallocDefault));
            c->h_block_cap = blk_words * 8 + blk_words;
        }
    } else {
        HIPC(c, c->o_seq.reserve(N)); HIPC(c, c->o_start.reserve(N)); HIPC(c, c->o_stop.reserve(N));
        HIPC(c, c->o_status.reserve(N)); HIPC(c, c->o_kld.reserve(N)); HIPC(c, c->o_gc.reserve(N));
        if (rip) { HIPC(c, c->o_pi.reserve(N)); HIPC(c, c->o_si.reserve(N)); HIPC(c, c->o_cri.reserve(N)); }
    }
    HIPC(c, c->o_sw.reserve(N)); HIPC(c, c->o_sg.reserve(N));
    if (dbg_counts) {
        HIPC(c, c->o_counts.reserve(N * size_t(c->nprof)));
        HIPC(c, hipMemsetAsync(c->o_counts.p, 0, N * size_t(c->nprof) * 4, c->stream));
    }
    if (dbg_meta) {
        HIPC(c, c->o_meta.reserve(N * 3));
        HIPC(c, hipMemsetAsync(c->o_meta.p, 0, N * 3 * 8, c->stream));
    }

    ScanParams P;
    P.codes = c->b().d_codes.p; P.inv = c->b().d_inv.p; P.low = c->b().d_low.p;
    P.descs = c->d_desc.p; P.ig = c->d_ig.p; P.log_tab = c->d_logtab.p; P.log_tab64 = c->d_logtab64.p; P.log_tab32 = c->d_logtab32.p;
    P.n_desc = c->b().n_seq + 1;
    P.kmin = c->kmin; P.kmax = c->kmax; P.w = w; P.inc = inc; P.flags = flags; P.c0 = c0; P.c1 = c1;
    P.orphan_cap = int32_t(c->plan_maxwin / 8 + 2);
    P.nprof = int32_t(c->nprof);
    if (packed_rows) {
        double* b = c->o_block.p;
        P.start = reinterpret_cast<int64_t*>(b); P.stop = reinterpret_cast<int64_t*>(b + Np); P.kld = b + 2 * Np; P.gc = b + 3 * Np;
        double* q = b + 4 * Np;
        P.pi = rip ? q : nullptr; P.si = rip ? q + Np : nullptr; P.cri = rip ? q + 2 * Np : nullptr;
        q += rip ? 3 * Np : 0;
        P.seq_index = reinterpret_cast<int32_t*>(q); P.status = reinterpret_cast<uint32_t*>(q) + Np;
    } else {
        P.seq_index = c->o_seq.p; P.start = c->o_start.p; P.stop = c->o_stop.p; P.status = c->o_status.p;
        P.kld = c->o_kld.p; P.gc = c->o_gc.p;
        P.pi = rip ? c->o_pi.p : nullptr; P.si = rip ? c->o_si.p : nullptr; P.cri = rip ? c->o_cri.p : nullptr;
    }
    P.sw = c->o_sw.p; P.sg = c->o_sg.p;
    P.dbg_counts = dbg_counts ? c->o_counts.p : nullptr;
    P.dbg_meta = dbg_meta ? c->o_meta.p : nullptr;
    P.dbg_ivom = nullptr;
    const size_t nk = size_t(1) << (2 * c->kmax);
    if (c->want_ivom) {
        HIPC(c, c->o_ivom.reserve(N * 2 * nk));
        HIPC(c, hipMemsetAsync(c->o_ivom.p, 0, N * 2 * nk * sizeof(double), c->stream));
        P.dbg_ivom = c->o_ivom.p;
    }
    P.stamps = nullptr;
    P.rc_tab = c->d_rctab.p; P.in_list = nullptr; P.in_count = nullptr; P.out_list = nullptr; P.out_count = nullptr;
    P.sel_mode = 0; P.sel_mod = 16; P.queue = nullptr; P.queue_n = 1; P.slide_pp = 0; P.ig_ring = nullptr;
    P.verdict = nullptr; P.my_form = 0u;
    c->scan_stat[0] = 16; c->scan_stat[1] = 0; c->scan_stat[2] = 0; c->scan_stat[3] = 1; c->scan_stat[4] = 0;
#ifdef FRISK_STAMPS
    DevBuf<unsigned long long> d_stamps;
    HIPC(c, d_stamps.reserve(4 * 16 * 12));
    HIPC(c, hipMemsetAsync(d_stamps.p, 0, 4 * 16 * 12 * 8, c->stream));
    P.stamps = d_stamps.p;
#endif

    const bool k8 = (c->kmax == 8);
    // LDS budget: 160 KB per workgroup.  Long windows at K = 8 need a long orphan list; the shared prefix tables
    // (12 KB, an optimisation only) make room for it.
    P.lv = shared_level(c->kmin, c->kmax);
    LdsLayout L = make_layout(c->kmin, c->kmax, P.orphan_cap, P.lv);
    if (L.total > 160 * 1024 && P.lv) { P.lv = 0; L = make_layout(c->kmin, c->kmax, P.orphan_cap, 0); }
    if (c->kmax <= 8 && c->plan_maxwin <= 65535 && L.total > 160 * 1024)
        return fail(c, FRISK_E_ARG, "window too long for the 160 KB LDS of one workgroup");
    const int wg_per_cu = std::max(1, std::min(2, int(160 * 1024 / L.total)));
    int grid = int(std::min<int64_t>(n, int64_t(c->num_cu) * wg_per_cu));
    if (grid >= 8) grid &= ~7;
    int64_t chunk = n / (int64_t(grid) * 8);
    chunk = std::max<int64_t>(1, std::min<int64_t>(chunk, 8));     // measured: 8 is best, 1..64 within 3 %
    if (const char* ev = tune_env("FRISK_SCAN_CHUNK")) chunk = std::max<int64_t>(1, std::atoll(ev));   // tuning knob
    P.chunk = int32_t(chunk);
    // fast paths: 512-thread workgroups, per-position loops unrolled ITS = 4 / 10 / 16 times (windows up to 2048 /
    // 5120 / 8192 bases); anything longer (up to 65535): generic 1024-thread kernel with runtime loops
    const int64_t need = (c->plan_maxwin + 511) / 512;
    const int its = need <= 4 ? 4 : need <= 10 ? 10 : need <= 16 ? 16 : 0;

    const bool force_one = tune_env("FRISK_ONE_WG") != nullptr;      // tuning knob: never two workgroups per CU
    // narrow-counter form (scan8_kernel.h): K = 8, kmin <= 5 (shared prefix level), windows of at most 256 x 20 bases.
    // width 0 = adaptive (the default), 4 / 8 = fixed, anything else = off (scan_kernel.h's 16-bit form for everything)
    int width = FRISK_K8_WIDTH;
    if (const char* ev = tune_env("FRISK_K8_BITS")) width = std::atoi(ev);
    // (decided by -w alone: rescued small scaffolds beyond the kernel's reach are handed on per window, see scan8_kernel.h)
    const bool narrow8 = k8 && c->kmin <= 5 && w <= 5120 && c->plan_maxwin <= 65535 && (width == 0 || width == 4 || width == 8);
    // K = 6, 7: the same kernel with 8-bit counters (a K-mer must occur 256 times in a window to wrap one)
    const bool narrow7 = (c->kmax == 6 || c->kmax == 7) && c->kmin <= c->kmax - 3 && w <= 5120 && c->plan_maxwin <= 65535 && width != 16;
    // (the per-max-mer IVOM dump of frisk_scan_ivom is written by scan_kernel.h's debug instantiation only)
    const bool narrow = (narrow8 || narrow7) && !c->want_ivom;
    // the 16-bit form (scan_kernel.h) over the candidates that PP names, by window class
    auto launch16 = [&](const ScanParams& PP, int g, hipStream_t st) -> hipError_t {
        hipError_t le;
#define FRISK_LAUNCH16(NT_, K8_, ITS_, DBG_) le = launch_scan<NT_, K8_, ITS_, DBG_>(PP, g, L.total, st)
        if (k8) {
            if (debug) { if (its) FRISK_LAUNCH16(512, true, 16, true); else FRISK_LAUNCH16(1024, true, 0, true); }
            else if (its == 4) FRISK_LAUNCH16(512, true, 4, false);
            else if (its == 10) FRISK_LAUNCH16(512, true, 10, false);
            else if (its == 16) FRISK_LAUNCH16(512, true, 16, false);
            else FRISK_LAUNCH16(1024, true, 0, false);
        } else {
            if (debug) { if (its) FRISK_LAUNCH16(512, false, 16, true); else FRISK_LAUNCH16(1024, false, 0, true); }
            else if (its == 4) FRISK_LAUNCH16(512, false, 4, false);
            else if (its == 10) FRISK_LAUNCH16(512, false, 10, false);
            else if (its == 16) FRISK_LAUNCH16(512, false, 16, false);
            else FRISK_LAUNCH16(1024, false, 0, false);
        }
#undef FRISK_LAUNCH16
        return le;
    };
    // rows [r0, r1) to the caller's buffers
    auto copy_rows = [&](int64_t r0, int64_t r1, hipStream_t st) -> int {
        const size_t m = size_t(r1 - r0);
        if (packed_rows) {              // (always the whole scan: short scans run in one segment) - unpacked behind the final wait
            HIPC(c, hipMemcpyAsync(c->h_block, c->o_block.p, blk_words * 8, hipMemcpyDeviceToHost, st));
            return FRISK_OK;
        }
        HIPC(c, hipMemcpyAsync(seq_index + r0, P.seq_index + r0, m * 4, hipMemcpyDeviceToHost, st));
        HIPC(c, hipMemcpyAsync(start + r0, P.start + r0, m * 8, hipMemcpyDeviceToHost, st));
        HIPC(c, hipMemcpyAsync(stop + r0, P.stop + r0, m * 8, hipMemcpyDeviceToHost, st));
        HIPC(c, hipMemcpyAsync(status + r0, P.status + r0, m * 4, hipMemcpyDeviceToHost, st));
        HIPC(c, hipMemcpyAsync(kld + r0, P.kld + r0, m * 8, hipMemcpyDeviceToHost, st));
        HIPC(c, hipMemcpyAsync(gc + r0, P.gc + r0, m * 8, hipMemcpyDeviceToHost, st));
        if (rip) {
            HIPC(c, hipMemcpyAsync(pi + r0, P.pi + r0, m * 8, hipMemcpyDeviceToHost, st));
            HIPC(c, hipMemcpyAsync(si + r0, P.si + r0, m * 8, hipMemcpyDeviceToHost, st));
            HIPC(c, hipMemcpyAsync(cri + r0, P.cri + r0, m * 8, hipMemcpyDeviceToHost, st));
        }
        return FRISK_OK;
    };
    bool rows_sent = false;                 // the narrow-counter path finishes and ships its rows itself, in two segments
    bool verdict_pending = false;           // the adaptive width's verdict was taken on the device in this scan: read back at the end
    unsigned int novf[34] = {0}, novf_sample[2] = {0, 0};
    HIPC(c, hipEventRecord(c->ev0, c->stream));
    hipError_t e = hipSuccess;
    if (c->plan_maxwin > 65535 || c->kmax > 8) {
        // windows beyond the 16-bit LDS counters, and every window at K > 8: 32-bit tables of all orders in a global scratch
        // slice per workgroup
        const int big_grid = int(std::min<int64_t>(n, c->num_cu));
        const int64_t stride = (c->nprof + 3) / 4 * 4;
        HIPC(c, c->d_big.reserve(size_t(big_grid) * size_t(stride)));
        HIPC(c, hipMemsetAsync(c->d_big.p, 0, size_t(big_grid) * size_t(stride) * 4, c->stream));
        if (debug) scan_big_kernel<true><<<big_grid, FRISK_BIG_NT, 0, c->stream>>>(P, c->d_big.p, stride);
        else scan_big_kernel<false><<<big_grid, FRISK_BIG_NT, 0, c->stream>>>(P, c->d_big.p, stride);
        e = hipGetLastError();
    } else
#define FRISK_LAUNCH(NT_, K8_, ITS_, DBG_) e = launch_scan<NT_, K8_, ITS_, DBG_>(P, grid, L.total, c->stream)
    if (narrow) {
        // K = 8 default (scan8_kernel.h): narrow order-8 counters, three (4-bit) or two (8-bit) independent 256-thread
        // workgroups per CU.  A window with a max-mer that occurs 16+ (256+) times - poly-A, microsatellites, satellite arrays -
        // wraps a 4-bit (8-bit) counter; the kernel notices and hands it to the next wider form through a device-side list:
        //     4-bit bulk -> list 1 -> 8-bit -> list 2 -> 16-bit (scan_kernel.h)        or        8-bit bulk -> list 2 -> 16-bit.
        // Which width suits the bulk depends on the sequence, so (width 0) every 16th chunk of 8 windows is scanned with 4-bit
        // counters first, and the share of it that had to be handed on decides the width for the other fifteen.  All three
        // forms give the same bits for a window (same arithmetic; 16-bit only ever sees the windows that wrap 8 bits), so
        // results do not depend on the choice, on the grid, or on the candidate range.
        P.ig_ring = nullptr;        // (the ring of scan8_kernel.h: allocated below, where the launch's shape is known)
        HIPC(c, c->d_ovf_list.reserve(N));
        HIPC(c, c->d_ovf_list2.reserve(N));
        HIPC(c, hipMemsetAsync(c->d_ovf_count.p, 0, 64 * sizeof(unsigned int), c->stream));
        const bool small_w = w <= 2048;
        // chunks of 16 consecutive windows where the tables slide and the genome-side values travel through the ring (one window
        // in 16 is counted - and gathered - afresh; measured on the bench shard: 8: 6.71 ms, 12: 6.65, 16: 6.61, 24: 6.79), of 8 otherwise
        const bool can_slide = 2 * int64_t(inc) <= int64_t(w) - (c->kmax - 1) && !tune_env("FRISK_NO_SLIDE");
#ifndef FRISK8_CHUNK_LONG
#define FRISK8_CHUNK_LONG 16
#endif
        // (a long scan can afford longer chunks - fewer windows counted and gathered afresh - while every workgroup still gets FRISK8_CHUNK_ROUNDS of them)
        int64_t chunk_cap = can_slide ? 16 : 8;
        if (can_slide && FRISK8_CHUNK_LONG > 16) chunk_cap = std::max<int64_t>(16, std::min<int64_t>(FRISK8_CHUNK_LONG, n / (int64_t(c->num_cu) * 3 * 64)));
        int64_t chunk8 = std::max<int64_t>(1, std::min<int64_t>(n / (int64_t(c->num_cu) * 3 * 8), chunk_cap));
        // A SHORT scan (fewer than 2 x 16 windows per workgroup: BASELINE's C3, a rank's share of a small genome) is dealt statically in
        // TWO rounds of the launch's workgroups: chunks of ceil(n / (2 x workgroups)) windows, tables sliding and the ring inside a chunk.
        // Measured (tools/exp/c3_sweep.py, us per scan of the first n windows of the shard; window by window / the best chunk):
        // 1 500: 55 / 55 (1);  3 000: 87 / 86 (2);  6 000: 160 / 137 (4);  12 063: 303 / 243 (8);  24 000: 550 / 451 (16) - one round
        // of longer chunks puts every workgroup through the same stage at the same time (12 063 in chunks of 16: 323), chunks dealt by
        // counters cost such a scan an atomic's round trip per chunk (12 063 in chunks of 8: 269 dealt, 243 static).
        const int64_t wgs3 = int64_t(c->num_cu) * (narrow8 ? 3 : FRISK_K7_WPS);
        const bool short_scan = can_slide && n < wgs3 * 2 * 16 && !(flags & FRISK_SCAN_CHUNKS) && !tune_env("FRISK_SCAN_CHUNK");
        if (short_scan) chunk8 = std::max<int64_t>(1, (n + wgs3 * 2 - 1) / (wgs3 * 2));
        if (flags & FRISK_SCAN_CHUNKS) chunk8 = 8;
        if (const char* ev = tune_env("FRISK_SCAN_CHUNK")) chunk8 = std::max<int64_t>(1, std::atoll(ev));
        // inside a chunk the order-K table slides from window to window where two windows share more than half their bases
        // (2 inc updates instead of w - K + 1 and a cleared table; scan8_kernel.h)
        if (can_slide && chunk8 >= 2) P.slide_pp = int32_t((inc + 255) / 256);
        // the ring through which genome-side values travel from window to window (scan8_kernel.h): a copy of the genome table (one
        // base address for both) followed by one slice of 20 rows x 512 columns per workgroup launched.  Only the K = 8 / 4-bit
        // instantiations with the ring read it: launches whose windows slide, and the debug form
        if (narrow8 && (P.slide_pp > 0 || debug)) {
            const size_t slices = size_t(std::min<int64_t>(std::max<int64_t>((n + chunk8 - 1) / chunk8, 1), int64_t(c->num_cu) * 4));
            const double* had = c->d_ig_ring.p;
            HIPC(c, c->d_ig_ring.reserve(nk + slices * (20 * FRISK8_RING_COLS + FRISK8_RING_PAD)));
            if (c->d_ig_ring.p != had || c->ring_gen != c->ig_gen) {        // (once per genome table, not once per scan)
                HIPC(c, hipMemcpyAsync(c->d_ig_ring.p, c->d_ig.p, nk * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
                c->ring_gen = c->ig_gen;
            }
            P.ig_ring = c->d_ig_ring.p;
        }
        const int64_t nchunks = (n + chunk8 - 1) / chunk8;
        // (the sample of the adaptive width: every 16th chunk, every 32nd or fewer of a long scan - a short launch runs at two thirds
        //  of a long one's rate, tools/exp/launch_size.py, and 12 000 windows tell the shares as well as 25 000
        //  ... and no more chunks than the launch has workgroups - one round: a second chunk for a few of them doubled its time)
        if (nchunks >= 64 * 32) P.sel_mod = int32_t(std::max<int64_t>(32, (nchunks + int64_t(c->num_cu) * 3 - 1) / (int64_t(c->num_cu) * 3)));
        // chunks dealt by counters (scan8_kernel.h) where a chunk is long enough to pay for the exchange: a short scan keeps the static deal
        const bool dealt = chunk8 >= 4 && !short_scan && !tune_env("FRISK_NO_DEAL");
        int bulk = (width == 4 && narrow8) ? 4 : 8;
        bool side = false;              // 4-bit bulk with the side table (scan8_kernel.h, SIDE)
        const bool side_ok = narrow8 && !debug;
        int sel_mode = 0;
        bool undecided = false;         // the sample's verdict is on the device: all three bulk forms are queued, one of them runs
        frisk_ctx::Batch& RB = c->b();
        const bool hinted = RB.width_hint != 0 && RB.hint_w == w && RB.hint_inc == inc;
        if (narrow7) { /* 8-bit bulk, no sample */ }
        else if ((flags & (FRISK_SCAN_BITS4 | FRISK_SCAN_SIDE4)) && narrow8) { bulk = 4; side = (flags & FRISK_SCAN_SIDE4) && side_ok; }
        else if (width == 0 && !debug && hinted) { bulk = RB.width_hint; side = RB.hint_side && side_ok; }   // same batch, same geometry: the earlier sample still holds
        else if (width == 0 && !debug && nchunks >= 64 * P.sel_mod) {
            ScanParams S = P;                   // the sample
            S.chunk = int32_t(chunk8);
            S.sel_mode = 1; S.out_list = c->d_ovf_list.p; S.out_count = c->d_ovf_count.p;
            if (dealt) { S.queue = c->d_ovf_count.p + 8; S.queue_n = 8; }
            const int64_t nsample = (nchunks + S.sel_mod - 1) / S.sel_mod;
            HIPC(c, launch_narrow(c->kmax, 4, small_w, false, S, c->num_cu, nsample, c->stream, true));
            // The verdict is taken ON THE DEVICE (scan8_decide_kernel, one thread behind the sample): the host queues all three bulk forms
            // behind it, each with the verdict's address and its own number, and two of them return at once - no host synchronisation
            // in the first scan of a batch (round 3: sample -> copy -> hipStreamSynchronize -> decide -> launch).  The rule:
            // 4-bit pays while fewer than about three windows in ten have to be redone (round 3, bench shard with simple repeats at
            // 0.05 / 0.1 / 0.2 / 0.3 per kb = 10 / 20 / 37 / 51 % of the scored windows handed on: 4-bit bulk 7.27 / 7.96 / 8.97 /
            // 9.98 ms, 8-bit bulk 8.51 / 8.57 / 8.71 / 8.78 ms - tools/exp/width_sweep.sh); the side table pays when the plain form
            // would hand on more than FRISK_SIDE_SHARE of the windows that are scored (it costs a scored window 1.0 ns - ten
            // instructions per position: 6.97 against 6.59 ms per scan -, a window handed on 18 ns: tools/exp/side_rate.py)
            scan8_decide_kernel<<<1, 1, 0, c->stream>>>(c->d_ovf_count.p, static_cast<unsigned int>(nsample * chunk8), double(FRISK_SIDE_SHARE),
                                                        side_ok ? 1 : 0, c->d_verdict.p);
            HIPC(c, hipGetLastError());
            undecided = true;
            verdict_pending = true;
            bulk = 4;                           // (what the launch shapes below assume until the verdict is read back)
            sel_mode = 2;
            // the sample's own hand-overs now (list 1 -> 8-bit -> list 2 -> 16-bit), so that lists and counters are free for
            // the bulk segments and no later pass touches rows of another segment
            ScanParams H = P;
            H.in_list = c->d_ovf_list.p; H.in_count = c->d_ovf_count.p;
            H.out_list = c->d_ovf_list2.p; H.out_count = c->d_ovf_count.p + 1;
            if (dealt) H.queue = c->d_ovf_count.p + 16;
            HIPC(c, launch_narrow(c->kmax, 8, small_w, false, H, c->num_cu, nsample * chunk8, c->stream));
            ScanParams H2 = P;
            H2.in_list = c->d_ovf_list2.p; H2.in_count = c->d_ovf_count.p + 1;
            int g16 = int(std::min<int64_t>(n, int64_t(c->num_cu)));
            if (g16 >= 8) g16 &= ~7;
            HIPC(c, launch16(H2, g16, c->stream));
            HIPC(c, hipMemcpyAsync(novf_sample, c->d_ovf_count.p, sizeof(novf_sample), hipMemcpyDeviceToHost, c->stream));
            HIPC(c, hipMemsetAsync(c->d_ovf_count.p, 0, 64 * sizeof(unsigned int), c->stream));
        }
        c->scan_stat[0] = bulk;
        c->scan_stat[4] = side ? 1 : 0;
        // Rows [r0, r1) of this scan on stream `st`: bulk launch, the two hand-over launches, the rows' scalar tail, and the
        // rows' scalar tail.  Segment `seg` has its own slice of the two lists (from entry r0) and its own 32 counters: [0], [1]
        // the lists' lengths, [8..15] the bulk launch's chunk queues (one per XCD), [16] the 8-bit launch's.
        auto run_rows = [&](int seg, int64_t r0, int64_t r1, hipStream_t st, bool fork_tail) -> int {
            const int64_t m = r1 - r0;
            ScanParams R = P;
            R.c0 = P.c0 + r0; R.c1 = P.c0 + r1;
            R.seq_index += r0; R.start += r0; R.stop += r0; R.status += r0; R.kld += r0; R.gc += r0; R.sw += r0; R.sg += r0;
            if (rip) { R.pi += r0; R.si += r0; R.cri += r0; }
            if (R.dbg_counts) R.dbg_counts += r0 * int64_t(c->nprof);
            if (R.dbg_meta) R.dbg_meta += r0 * 3;
            if (R.dbg_ivom) R.dbg_ivom += r0 * 2 * int64_t(nk);
            unsigned int* cnt = c->d_ovf_count.p + 32 * seg;
            int64_t* list1 = c->d_ovf_list.p + r0;
            int64_t* list2 = c->d_ovf_list2.p + r0;
            ScanParams B = R;                       // the bulk launch
            B.chunk = int32_t(chunk8);
            B.sel_mode = sel_mode;
            if (bulk == 4) { B.out_list = list1; B.out_count = cnt; }
            else { B.out_list = list2; B.out_count = cnt + 1; }
            if (dealt) { B.queue = cnt + 8; B.queue_n = 8; }
            const int64_t mchunks = (m + chunk8 - 1) / chunk8;
            const int64_t bulk_chunks = sel_mode == 2 ? mchunks - (mchunks + B.sel_mod - 1) / B.sel_mod : mchunks;
            if (undecided) {                        // plain 4-bit / 4-bit + side table / 8-bit: the device's verdict lets one of them run
                B.verdict = c->d_verdict.p;
                B.out_list = list1; B.out_count = cnt;
                B.my_form = 1u;
                HIPC(c, launch_narrow(c->kmax, 4, small_w, false, B, c->num_cu, bulk_chunks, st, false, false));
                B.my_form = 2u;
                HIPC(c, launch_narrow(c->kmax, 4, small_w, false, B, c->num_cu, bulk_chunks, st, false, true));
                B.my_form = 3u;
                B.out_list = list2; B.out_count = cnt + 1;
                HIPC(c, launch_narrow(c->kmax, 8, small_w, false, B, c->num_cu, bulk_chunks, st));
            } else
            HIPC(c, launch_narrow(c->kmax, bulk, small_w, debug, B, c->num_cu, bulk_chunks, st, false, side));
            if (bulk == 4) {                        // list 1 (4-bit hand-overs) -> 8-bit -> list 2
                ScanParams H = R;
                H.in_list = list1; H.in_count = cnt;
                H.out_list = list2; H.out_count = cnt + 1;
                if (dealt) H.queue = cnt + 16;
                HIPC(c, launch_narrow(c->kmax, 8, small_w, debug, H, c->num_cu, m, st));
            }
            // list 2 -> 16-bit counters, one window per workgroup at a time (a no-op when the list is empty)
            R.in_list = list2; R.in_count = cnt + 1;
            int g16 = int(std::min<int64_t>(m, int64_t(c->num_cu)));
            if (g16 >= 8) g16 &= ~7;
            HIPC(c, launch16(R, g16, st));
            if (fork_tail) {        // the tail segment starts here: beside this segment's scalar tail and its rows' way to the host
                HIPC(c, hipEventRecord(c->ev_fork, st));
                HIPC(c, hipStreamWaitEvent(c->tail_stream, c->ev_fork, 0));
            }
            finish_rows_kernel<<<grid_for(m, 256, 1 << 20), 256, 0, st>>>(m, R.status, R.kld, R.gc, R.sw, R.sg);
            HIPC(c, hipGetLastError());
            return FRISK_OK;
        };
        // The last sixteenth of a long scan goes to a second stream and starts when the kernels of the first fifteen are done:
        // it runs while their rows travel to the host (16 MB per 410 k windows: 0.36 ms that used to follow the scan).  The
        // cut is a multiple of 16 chunks: chunk numbering and the sample's stride stay aligned across it.
        const int64_t unit = chunk8 * P.sel_mod;
        int64_t cut = n;
        // (worth a second launch only when the rows' way to the host is long against a launch: 40 B x 128 K rows ~ 0.1 ms)
        if (!debug && !c->want_ivom && n >= (int64_t(1) << 17) && n >= 64 * unit && !tune_env("FRISK_ONE_SEGMENT")) {
            cut = (n / unit - std::max<int64_t>(1, n / unit / 16)) * unit;
            // ... and the tail is a launch of its own: about a sixteenth of the windows is two chunks per workgroup - 1 616 chunks on 768
            // workgroups left a tenth of them a third chunk and the others idle (0.78 ms under the profiler for 0.40 ms of work).  So
            // the tail takes whole rounds: the largest number of chunks <= rounds x workgroups that the cut's alignment allows.
            const int64_t wgs = int64_t(c->num_cu) * (bulk == 4 ? 3 : 2);
            const int64_t tail_chunks = (n - cut + chunk8 - 1) / chunk8;
            if (tail_chunks >= wgs && !tune_env("FRISK_TAIL_ANY")) {
                const int64_t rounds = (tail_chunks + wgs / 2) / wgs;
                // (a cut is a whole number of units - the kernels number a segment's chunks from its first candidate - and leaves a tail)
                cut = std::min((n / unit - 1) * unit, (n - rounds * wgs * chunk8 + unit - 1) / unit * unit);
            }
            if (cut <= 0 || cut >= n || cut % unit != 0) return fail(c, FRISK_E_STATE, "frisk_scan: row segments cut off a unit boundary");
        }
        rc = run_rows(0, 0, cut, c->stream, cut < n);
        if (rc) return rc;
        if (cut < n) {
            rc = run_rows(1, cut, n, c->tail_stream, false);
            if (rc) return rc;
            HIPC(c, hipEventRecord(c->ev_tail_kernels, c->tail_stream));
            rc = copy_rows(cut, n, c->tail_stream);
            if (rc) return rc;
            HIPC(c, hipEventRecord(c->ev_tail_done, c->tail_stream));
            rc = copy_rows(0, cut, c->stream);
            if (rc) return rc;
            HIPC(c, hipStreamWaitEvent(c->stream, c->ev_tail_kernels, 0));
            HIPC(c, hipEventRecord(c->ev1, c->stream));                     // every scan kernel of this call has finished
            HIPC(c, hipStreamWaitEvent(c->stream, c->ev_tail_done, 0));
        } else {
            HIPC(c, hipEventRecord(c->ev1, c->stream));
            rc = copy_rows(0, n, c->stream);
            if (rc) return rc;
        }
        rows_sent = true;
        c->scan_stat[3] = cut < n ? 2 : 1;
    } else if (k8) {
        e = launch16(P, grid, c->stream);
    } else if (!debug && !force_one && c->plan_maxwin <= 5120 && L.total <= 80 * 1024) {
        // K <= 7: the tables of a window take < 60 KB, so TWO independent 256-thread workgroups fit a CU.  The two waves of
        // a SIMD then belong to different windows in different stages, and the LDS phases of one overlap the VALU phases of
        // the other: measured -23 % (K = 7) and -26 % (K = 6) against one 512-thread workgroup with the same code.
        grid = int(std::min<int64_t>(n, int64_t(c->num_cu) * 2));
        if (grid >= 8) grid &= ~7;
        P.chunk = int32_t(std::max<int64_t>(1, std::min<int64_t>(n / (int64_t(grid) * 8), 8)));
        if (c->plan_maxwin <= 2048) FRISK_LAUNCH(256, false, 8, false); else FRISK_LAUNCH(256, false, 20, false);
    } else {
        e = launch16(P, grid, c->stream);
    }
#undef FRISK_LAUNCH
    HIPC(c, e);
#ifdef FRISK_STAMPS
    {
        std::vector<unsigned long long> h(4 * 16 * 12);
        HIPC(c, hipMemcpyAsync(h.data(), d_stamps.p, h.size() * 8, hipMemcpyDeviceToHost, c->stream));
        HIPC(c, hipStreamSynchronize(c->stream));
        static const char* names16[9] = {"stage1", "barrier1", "stage2", "stage3", "barrier3", "stage4", "blocksum", "cleanup+store", "endbarrier"};
        static const char* names8[9] = {"stage1", "barrier1", "stage2a+b", "stage2b+b", "stage3+b", "stage4", "sums+b", "clear+store", "endbarrier"};
        const char* const* names = narrow ? names8 : names16;
        double acc[9] = {0}; int cnt = 0;
        for (int b = 0; b < 4; ++b) for (int w = 4; w < 16; ++w) {
            const unsigned long long* t = &h[(b * 16 + w) * 12];
            if (!t[0] || !t[9]) continue;
            for (int i = 0; i < 9; ++i) acc[i] += double(t[i + 1] - t[i]);
            ++cnt;
        }
        if (cnt) { std::fprintf(stderr, "[stamps] windows %d:", cnt); for (int i = 0; i < 9; ++i) std::fprintf(stderr, " %s %.0f", names[i], acc[i] / cnt); std::fprintf(stderr, "\n"); }
        d_stamps.release();
    }
#endif
    if (!rows_sent) {
        if (c->plan_maxwin <= 65535 && c->kmax <= 8 && n > 0) {     // the LDS kernels leave the rows' scalar tail to one thread per row
            finish_rows_kernel<<<grid_for(n, 256, 1 << 20), 256, 0, c->stream>>>(n, P.status, P.kld, P.gc, P.sw, P.sg);
            HIPC(c, hipGetLastError());
        }
        HIPC(c, hipEventRecord(c->ev1, c->stream));
        rc = copy_rows(0, n, c->stream);
        if (rc) return rc;
    }
    if (dbg_counts)
        HIPC(c, hipMemcpyAsync(dbg_counts, c->o_counts.p, N * size_t(c->nprof) * 4, hipMemcpyDeviceToHost, c->stream));
    if (dbg_meta) HIPC(c, hipMemcpyAsync(dbg_meta, c->o_meta.p, N * 3 * 8, hipMemcpyDeviceToHost, c->stream));
    if (c->want_ivom) HIPC(c, hipMemcpyAsync(c->want_ivom, c->o_ivom.p, N * 2 * nk * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (narrow) HIPC(c, hipMemcpyAsync(novf, c->d_ovf_count.p, sizeof(novf), hipMemcpyDeviceToHost, c->stream));
    unsigned int verdict_host[4] = {0, 0, 0, 0};
    if (verdict_pending) HIPC(c, hipMemcpyAsync(verdict_host, c->d_verdict.p, sizeof(verdict_host), hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    if (verdict_pending) {              // what the device decided: remembered per batch and geometry (later scans launch that form alone)
        frisk_ctx::Batch& VB = c->b();
        VB.width_hint = verdict_host[0] == 3u ? 8 : 4; VB.hint_w = w; VB.hint_inc = inc; VB.hint_side = verdict_host[0] == 2u ? 1 : 0;
        c->scan_stat[0] = VB.width_hint;
        c->scan_stat[4] = VB.hint_side;
    }
    if (packed_rows) {
        const double* b = static_cast<const double*>(c->h_block);
        std::memcpy(start, b, N * 8); std::memcpy(stop, b + Np, N * 8); std::memcpy(kld, b + 2 * Np, N * 8); std::memcpy(gc, b + 3 * Np, N * 8);
        const double* q = b + 4 * Np;
        if (rip) { std::memcpy(pi, q, N * 8); std::memcpy(si, q + Np, N * 8); std::memcpy(cri, q + 2 * Np, N * 8); q += 3 * Np; }
        std::memcpy(seq_index, q, N * 4); std::memcpy(status, reinterpret_cast<const uint32_t*>(q) + Np, N * 4);
    }
    c->scan_stat[1] = novf[0] + novf[32] + novf_sample[0];
    c->scan_stat[2] = novf[1] + novf[33] + novf_sample[1];
    if (c->b().tiled)               // descriptor index -> index of the scaffold in the FASTA
        for (size_t r = 0; r < N; ++r) seq_index[r] = c->b().tiles[size_t(seq_index[r])].scaf;
    float ms = 0;
    HIPC(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->ms[0] = ms;
    return FRISK_OK;
}

int frisk_scan_ivom(frisk_ctx* c, int32_t w, int32_t inc, uint32_t flags, int64_t c0, int64_t c1, int64_t cap,
                    double* window_ivom, double* genome_ivom) {
    if (!c || !window_ivom || !genome_ivom) return FRISK_E_ARG;
    if (c->kmax > 6) return fail(c, FRISK_E_ARG, "frisk_scan_ivom: the per-max-mer dump exists for kmax <= 6 only");
    int64_t ncand = 0;
    int rc = frisk_scan_plan(c, w, inc, flags, &ncand);
    if (rc) return rc;
    if (c1 < 0) c1 = ncand;
    if (c0 < 0 || c0 > c1 || c1 > ncand) return fail(c, FRISK_E_ARG, "candidate range outside [0, n_candidates]");
    const int64_t n = c1 - c0;
    if (cap < n) return fail(c, FRISK_E_CAP, "output capacity too small: need " + std::to_string(n));
    if (c->plan_maxwin > 65535) return fail(c, FRISK_E_ARG, "frisk_scan_ivom: windows of at most 65535 bases");
    const size_t nk = size_t(1) << (2 * c->kmax), N = size_t(std::max<int64_t>(n, 1));
    std::vector<double> raw(N * 2 * nk), kld(N), gc(N), pi(N), si(N), cri(N);
    std::vector<int32_t> seq(N);
    std::vector<int64_t> start(N), stop(N), meta(N * 3);
    std::vector<uint32_t> status(N);
    c->want_ivom = raw.data();
    rc = frisk_scan(c, w, inc, flags & ~FRISK_SCAN_RIP, c0, c1, int64_t(N), seq.data(), start.data(), stop.data(), status.data(),
                    kld.data(), gc.data(), nullptr, nullptr, nullptr, nullptr, meta.data());
    c->want_ivom = nullptr;
    if (rc) return rc;
    for (int64_t r = 0; r < n; ++r) {           // normalise over the window's present max-mers (L450-454), in index order
        const double* iw = raw.data() + size_t(r) * 2 * nk;
        const double* ig = iw + nk;
        double sw = 0.0, sg = 0.0;
        for (size_t k = 0; k < nk; ++k) { sw += iw[k]; sg += ig[k]; }
        for (size_t k = 0; k < nk; ++k) {
            window_ivom[size_t(r) * nk + k] = iw[k] != 0.0 ? iw[k] / sw : 0.0;
            genome_ivom[size_t(r) * nk + k] = iw[k] != 0.0 ? ig[k] / sg : 0.0;
        }
    }
    return FRISK_OK;
}

char* frisk_format_rows(int64_t n, const char* const* names, const int32_t* seq_index, const int64_t* start, const int64_t* stop,
                        const uint8_t* kld_is_int0, const double* kld, const double* gc, const double* pi, const double* si,
                        const double* cri, int64_t* out_len) {
    if (out_len) *out_len = 0;
    if (n < 0 || (n > 0 && (!names || !seq_index || !start || !stop || !kld || !gc))) return nullptr;
    if ((pi || si || cri) && !(pi && si && cri)) return nullptr;
    frisk_text::Columns c{n, names, seq_index, start, stop, kld_is_int0, kld, gc, pi, si, cri};
    return frisk_text::format_all(c, out_len);
}
void frisk_free(void* p) { std::free(p); }

// ---- host-native 2-state Gaussian HMM (hmm_host.h): the model frisk_amd/hmm.py documents, for millions of windows ----------
int frisk_hmm_fit(const double* x, int64_t n, int32_t n_iter, double tol, double min_covar, double covars_prior, double* means,
                  double* covars, double* startprob, double* transmat, double* loglik, int32_t* iters) {
    if (!x || n < 1 || n_iter < 0 || !means || !covars || !startprob || !transmat) return FRISK_E_ARG;
    for (int64_t t = 0; t < n; ++t) if (!std::isfinite(x[t])) return FRISK_E_ARG;
    const frisk_hmm::Fit F = frisk_hmm::fit(x, n, n_iter, tol, min_covar, covars_prior);
    for (int i = 0; i < 2; ++i) { means[i] = F.m.means[i]; covars[i] = F.m.covars[i]; startprob[i] = F.m.startprob[i]; }
    for (int i = 0; i < 4; ++i) transmat[i] = F.m.transmat[i];
    if (loglik) *loglik = F.loglik;
    if (iters) *iters = F.iters;
    return FRISK_OK;
}

int frisk_hmm_viterbi(const double* x, const int64_t* seg_off, int32_t n_seg, const double* means, const double* covars,
                      const double* startprob, const double* transmat, int8_t* states) {
    if (n_seg < 0 || !seg_off || !means || !covars || !startprob || !transmat) return FRISK_E_ARG;
    for (int32_t s = 0; s < n_seg; ++s) if (seg_off[s + 1] < seg_off[s]) return FRISK_E_ARG;
    if (n_seg == 0 || seg_off[n_seg] == seg_off[0]) return FRISK_OK;
    if (!x || !states) return FRISK_E_ARG;
    frisk_hmm::Model m;
    for (int i = 0; i < 2; ++i) { m.means[i] = means[i]; m.covars[i] = covars[i]; m.startprob[i] = startprob[i]; }
    for (int i = 0; i < 4; ++i) m.transmat[i] = transmat[i];
    frisk_hmm::viterbi_segments(x, seg_off, n_seg, m, states);
    return FRISK_OK;
}

int64_t frisk_last_scan_stat(const frisk_ctx* c, int which) { return (c && which >= 0 && which < 5) ? c->scan_stat[which] : -1; }

}  // extern "C"
