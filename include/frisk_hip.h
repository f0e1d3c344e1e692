/* frisk_hip.h - C ABI of libfrisk_hip.so: the MI355X (gfx950) implementation of frisk's hot path.
 *
 * The reference (Adamtaranto/frisk, one Python-2 module) has no FFI, plugin or operator
 * interface; its only stable boundary is the call pattern inside main().  Each entry point
 * below names the reference lines it replaces (citations are to /root/reference/frisk/__init__.py):
 *
 *   phase A  genome k-mer profile      computeKmers(genomeMode=True)   L1442  (L280-367)
 *   phase B  window scan               the loop L1478-1494: crawlGenome (L194-251) ->
 *            computeKmers(window) (L280-367) -> IvomBuild x2 (L369-457) -> KLD (L459-472)
 *            -> calcGC (L120-137) [-> calcRIP (L474-495)]
 *
 * Conventions: plain C, plain pointers and sizes, no exceptions, no torch types.  Every
 * function returning int returns FRISK_OK (0) or a negative FRISK_E_* code; the message is
 * available from frisk_last_error().  The caller owns every host buffer it passes; the
 * library never keeps a caller pointer after the call returns.  A context is bound to one
 * device and one HIP stream and is not thread-safe; use one context per device per thread
 * (multi-GPU = one process per GPU, one context each).  K-mer index convention (reference
 * L70, L253-274): digits A=0,T=1,G=2,C=3, first base most significant; tables for orders
 * kmin..kmax are concatenated in ascending order ("profile layout", frisk_profile_len()).
 */
#ifndef FRISK_HIP_H
#define FRISK_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct frisk_ctx frisk_ctx;

enum {
    FRISK_OK = 0,
    FRISK_E_ARG = -1,         /* bad argument / unsupported geometry                                   */
    FRISK_E_HIP = -2,         /* a HIP runtime call failed (no device, out of memory, launch failure)  */
    FRISK_E_STATE = -3,       /* call order violated (e.g. scan before a profile is finalised)         */
    FRISK_E_CAP = -4,         /* caller buffer too small; the needed size is reported                  */
    FRISK_E_ZERO_WEIGHT = -5, /* reserved: ZeroDivisionError cases are reported per row (FRISK_ROW_ZERO_WEIGHT)  */
    FRISK_E_INDEX = -6        /* no usable seek index for this FASTA file: the caller parses it instead  */
};

/* frisk_scan flags */
#define FRISK_SCAN_RIP           1u   /* fill pi/si/cri (--RIP, L1485-1486); needs kmin <= 2 <= kmax  */
#define FRISK_SCAN_SCAFFOLDS_ALL 2u   /* --scaffoldsAll: small scaffolds become one window (L211-221) */
#define FRISK_SCAN_CHUNKS       256u /* schedule the windows in chunks of 8 consecutive candidates whatever the scan's size (a short
                                        scan is otherwise dealt window by window): the path long scans take - tables sliding from
                                        window to window inside a chunk - on inputs of any size.  Results are the same bits. */
#define FRISK_SCAN_BITS4        512u /* K = 8: the bulk launch with 4-bit counters whatever the scan's size, overflowing windows handed on
                                        to the 8- and 16-bit forms (a short scan otherwise starts at 8 bits; a long one samples first).
                                        Results are the same bits. */
#define FRISK_SCAN_SIDE4       1024u /* K = 8: as FRISK_SCAN_BITS4, with the side table for the max-mers of period <= 4 (poly-A, (CA)n, (AAAT)n
                                        ...) beside the 4-bit counters - the form a long scan of repeat-rich sequence picks by itself.
                                        Results are the same bits. */

/* per-row status bits written to `status` by frisk_scan */
#define FRISK_ROW_KEPT        1u      /* window passed the < 30 % non-ACGT filter (L237-241)          */
#define FRISK_ROW_ZERO_WEIGHT 2u      /* reference raises ZeroDivisionError for this window (kld = NaN) */
#define FRISK_ROW_JUMPBACK    4u      /* end-of-scaffold "jumpback" window (0-based start, L230-243)  */
#define FRISK_ROW_NO_MAXMER   8u      /* no valid max-mer: the reference's KLD is the int 0 (L465)    */

const char* frisk_version(void);

/* Library limits for (kmin,kmax,window length); 0 = unsupported.  kmax <= 12 (the reference's -k is unbounded, L1197-1206,
 * but its own cost grows with 4^K); kmax <= 8 and windows up to 65535 bases run in LDS, everything else (windows up to
 * 2^31-1 bases, kmax 9..12) on a slower path with 32-bit tables in global memory. */
int frisk_supported(int kmin, int kmax, int64_t max_window);

/* Context: device ordinal, word sizes -m/-k (L1197-1206). */
int  frisk_create(int device, int kmin, int kmax, frisk_ctx** out);
void frisk_destroy(frisk_ctx* ctx);
const char* frisk_last_error(const frisk_ctx* ctx);   /* library-owned, valid until the next call on ctx */
int64_t frisk_profile_len(const frisk_ctx* ctx);      /* sum_{x=kmin..kmax} 4^x                          */

/* ---- sequence residency ------------------------------------------------------------------
 * Replaces the hand-off of scaffold strings from iterFasta (L139-164) to the counters
 * (L297, L203).  Uploads n_seq scaffolds (ASCII, any case, any IUPAC letter), packs them on
 * the device to 2 bits/base + validity and soft-mask bitmaps, and keeps them resident until the
 * next load.  Empty scaffolds (len 0) are allowed. */
int frisk_seq_load(frisk_ctx* ctx, const uint8_t* const* seqs, const int64_t* lens, int32_t n_seq);

/* Double-buffered residency - the "streamed to HBM" of the north star.  frisk_seq_stage uploads and packs the NEXT batch on
 * a second HIP stream while the resident batch is being profiled / scanned; frisk_seq_commit makes it the resident batch
 * (the compute stream waits for the upload on the device, the host does not block).  The copies are asynchronous when the
 * source buffers are page-locked (frisk_host_alloc); the caller keeps them alive until the next call after the commit that
 * synchronises with the device - frisk_scan, frisk_profile_get, frisk_profile_export_host, frisk_seq_export_packed,
 * frisk_seq_read (frisk_profile_reset / _add / _finalize only enqueue).  A stage also waits, on the device, for work
 * queued before the last commit: the slot it fills may still be read by that work.  frisk_seq_stage_packed takes the three bit-packed arrays in the
 * library's own layout (as frisk_seq_export_packed returns them for the same `lens`): 0.5 bytes per base over PCIe instead
 * of 1, and no parsing - the sequence-cache path of the CLI. */
int frisk_seq_stage(frisk_ctx* ctx, const uint8_t* const* seqs, const int64_t* lens, int32_t n_seq);
int frisk_seq_stage_packed(frisk_ctx* ctx, const uint32_t* codes, const uint32_t* inv, const uint32_t* low,
                           const int64_t* lens, int32_t n_seq);
/* The 0.25 B/base upload form - the north star's "2-bit-packed and streamed to HBM", SURVEY.md 8(d)'s algorithmic bytes: only
 * the 2-bit codes travel densely (2 * P / 32 words, P = frisk_padded_len_of(lens, n_seq), library layout: scaffold s at padded
 * positions [off, off + len), one PAD behind it, 16 bases per word, first base most significant); the two masks travel as run
 * lists - n_inv / n_low pairs [begin, end) of padded positions, ascending and disjoint: inv = letters other than ACGTacgt,
 * low = lowercase acgt - and are expanded to the bitmaps on the device; PADs are the library's business.  n_inv (n_low) < 0:
 * the argument is the dense bitmap instead (P / 32 words, PAD bits clear) - for an assembly with more runs than bitmap words.
 * The codes cross PCIe in pieces of piece_bases positions (0: the library's default, 256 Mbases = 64 MB) with an event behind
 * each: frisk_seq_commit does not wait for them, and frisk_profile_add(-1, -1) on the committed batch counts piece i while
 * piece i + 1 is on its way, so that of phase A only the last piece's kernel follows the upload.  Any other use of the batch
 * waits (on the device) for the last piece.  Replaces, with frisk_pack_2bit, the hand-off of scaffold strings from iterFasta
 * (L139-164) to computeKmers (L297) and crawlGenome (L203).  Caller keeps the arrays alive as for frisk_seq_stage. */
int frisk_seq_stage_2bit(frisk_ctx* ctx, const uint32_t* codes, const int64_t* inv_runs, int64_t n_inv, const int64_t* low_runs,
                         int64_t n_low, const int64_t* lens, int32_t n_seq, int64_t piece_bases);
/* Host-only (multi-threaded): scaffolds as ASCII -> that form.  codes: caller's buffer of 2 * P / 32 words (page-locked memory
 * makes the upload asynchronous); *inv_runs / *low_runs: malloc'd by the library (frisk_free), n pairs each. */
int frisk_pack_2bit(const uint8_t* const* seqs, const int64_t* lens, int32_t n_seq, uint32_t* codes, int64_t** inv_runs,
                    int64_t* n_inv, int64_t** low_runs, int64_t* n_low);
int64_t frisk_padded_len_of(const int64_t* lens, int32_t n_seq);   /* P: sum(len + 1) rounded up to 32 (32 for an empty batch); -1: bad lengths */
/* The resident batch in that form (the CLI's sequence cache): codes into the caller's buffer, run lists malloc'd (frisk_free). */
int frisk_seq_export_2bit(frisk_ctx* ctx, uint32_t* codes, int64_t** inv_runs, int64_t* n_inv, int64_t** low_runs, int64_t* n_low);
int frisk_seq_commit(frisk_ctx* ctx);
/* Packed arrays of the resident batch: codes 2 * P / 32 words, inv and low P / 32 words each, P = frisk_seq_padded_len(). */
int frisk_seq_export_packed(frisk_ctx* ctx, uint32_t* codes, uint32_t* inv, uint32_t* low);
/* Record names for a batch that did not come from frisk_fasta_load (e.g. from the sequence cache). */
int frisk_seq_set_names(frisk_ctx* ctx, const char* const* names, int32_t n_seq);

/* Native FASTA reader: parse `path` (plain or gzip) with the record semantics of the reference's iterFasta
 * (L139-164: name = first whitespace-delimited token of the header with '>' stripped from both ends, lines stripped of
 * surrounding whitespace, blank lines skipped, case preserved) straight into the upload layout, and make its records
 * the resident batch.  Record names / lengths are then available from frisk_seq_name / frisk_seq_len. */
int frisk_fasta_load(frisk_ctx* ctx, const char* path, int32_t* n_seq, int64_t* total_len);
/* Host-only test utility: parse `path` with the native reader and return the number of records, their total length and an
 * FNV-1a digest over (name, 0, sequence, 0) of every record - what the CPU test-suite compares with the Python reader. */
int frisk_fasta_digest(const char* path, int32_t* n_seq, int64_t* total_len, uint64_t* digest);
/* Host-only: a FASTA file (plain or gzip) straight into the 0.25 B/base form - what frisk_fasta_load does before its upload (iterFasta
 * L139-164 -> frisk_pack_2bit in one pass over the mapped file, no one-byte-per-base buffer in between).  All five arrays are
 * malloc'd by the library (frisk_free): *lens (n_seq values), *codes (*n_code_words = 2 * P / 32 words), the two run lists. */
int frisk_fasta_pack_2bit(const char* path, int32_t* n_seq, int64_t** lens, uint32_t** codes, int64_t* n_code_words, int64_t** inv_runs,
                          int64_t* n_inv, int64_t** low_runs, int64_t* n_low);

/* Multi-GPU form of frisk_fasta_load (window-tile sharding with halo): every rank parses the file, but keeps
 * resident only what it needs for ITS share of the job - the candidate windows [*cand_begin, *cand_end) of the job's
 * numbering (an equal contiguous share) and the positions whose k-mers it counts in phase A (every base of the genome
 * belongs to exactly one rank; K-1 bases of halo behind each owned range).  The window geometry is fixed here; frisk_scan
 * on the batch then numbers candidates from 0 = *cand_begin, frisk_profile_add(-1,-1) counts the owned positions, and
 * seq_index / frisk_seq_name / frisk_seq_len / frisk_seq_count refer to the records of the FASTA, as on one GPU. */
int frisk_fasta_load_shard(frisk_ctx* ctx, const char* path, int32_t w, int32_t inc, uint32_t flags, int32_t rank,
                           int32_t world, int32_t* n_seq, int64_t* total_len, int64_t* cand_begin, int64_t* cand_end);
/* The same without parsing the file: a seek index (frisk_amd/csrc/fasta_index.h: one `samtools faidx` line per record behind a
 * stamp line with the FASTA's size and modification time; a foreign <fasta>.fai that is not older than the file is accepted
 * too) tells the rank where the bases of ITS tiles are, and it copies those bytes from the mapped file - 1/world of the
 * file + halos instead of all of it.  The index is checked against the file where it is used (a header line that gives the
 * record's name ends right before its first base, its last base ends a line).  FRISK_E_INDEX: no usable index (missing,
 * made from another version of the file, gzip, a record the byte arithmetic cannot address) - call frisk_fasta_load_shard.
 * The reference reads every record of the file on every run (iterFasta, L139-164). */
int frisk_fasta_load_shard_indexed(frisk_ctx* ctx, const char* path, const char* index_path, int32_t w, int32_t inc,
                                   uint32_t flags, int32_t rank, int32_t world, int32_t* n_seq, int64_t* total_len,
                                   int64_t* cand_begin, int64_t* cand_end);
/* Host-only: write the seek index of `fasta_path` to `index_path` (one pass over the mapped file).  FRISK_E_INDEX when the
 * file is not regular - text before the first header, blank lines or surrounding blanks inside a record, lines of different
 * length before a record's last, gzip - with the reason in `why` (nullable). */
int frisk_fasta_index_build(const char* fasta_path, const char* index_path, int32_t* n_seq, char* why, int32_t why_cap);
/* Host-only test / extraction utility: bases [pos0, pos0 + n) of record seq_index through the index (seq_index < 0: the number
 * of records alone; n = 0: name and length alone).  Any output pointer may be NULL. */
int frisk_fasta_index_read(const char* fasta_path, const char* index_path, int32_t seq_index, int64_t pos0, int64_t n,
                           uint8_t* out, int32_t* n_seq, int64_t* seq_len, char* name, int32_t name_cap, char* why,
                           int32_t why_cap);
int32_t frisk_seq_count(const frisk_ctx* ctx);
const char* frisk_seq_name(const frisk_ctx* ctx, int32_t seq_index);   /* "" for batches not loaded from FASTA */
int64_t frisk_seq_len(const frisk_ctx* ctx, int32_t seq_index);

/* Bench/test utility: fill the resident batch with synthetic scaffolds generated ON the device
 * (order-3 Markov background + compositional islands + N runs + soft-masked runs + simple repeats - poly-A / poly-T
 * tails and microsatellites, repeats_per_kb of them per 1000 bases, soft-masked; the generator is specified in
 * frisk_amd/synth.py, which reproduces it bit-for-bit on the host). */
int frisk_seq_synth(frisk_ctx* ctx, const int64_t* lens, int32_t n_seq, uint64_t seed,
                    double island_frac, double n_frac, double lower_frac, double repeats_per_kb);
/* ... with repeat content no table of the scan kernel is shaped after: a share period_mix of the simple repeats is of period 3, 5
 * or 6 ((CAG)n, (AAT)n, (AAAAT)n, (TTAGGG)n, up to 210 bases), and a share sat_frac of the bases lies in satellite arrays - tandem
 * copies of a 171-base monomer, 3 % divergence between copies, 0.13 .. 1.05 Mb each (frisk_amd/synth.py is the specification). */
int frisk_seq_synth2(frisk_ctx* ctx, const int64_t* lens, int32_t n_seq, uint64_t seed, double island_frac, double n_frac,
                     double lower_frac, double repeats_per_kb, double period_mix, double sat_frac);

/* Copy bases [offset, offset+n) of resident scaffold seq_index back as ASCII (canonical letters:
 * A/T/G/C, a/t/g/c, N for every non-ACGT letter) - test / bench-sampling utility. */
int frisk_seq_read(frisk_ctx* ctx, int32_t seq_index, int64_t offset, int64_t n, uint8_t* out);

/* ---- phase A: genome profile (computeKmers genomeMode=True, L1442) -------------------------
 * reset -> add (once per resident batch; forward counts only) -> [all-reduce across GPUs]
 * -> finalize (adds the reverse complement, L350-351, and builds the genome-side IVOM table). */
int frisk_profile_reset(frisk_ctx* ctx);
/* Count the k-mers that START in padded positions [pos_begin,pos_end) of the resident batch;
 * pos_begin = pos_end = -1 means the whole batch.  mask_host: bit 0 = --maskHost (L336-337); bit 1 = FRISK_PROFILE_ONE_PASS, a test
 * hook: at kmax = 8 take the one-pass form with 16-bit LDS counters (what ranges of 2^30+ positions take by themselves; a wrapped
 * counter falls back to the two-pass form) whatever the size.  Same counts either way. */
#define FRISK_PROFILE_ONE_PASS 2
int frisk_profile_add(frisk_ctx* ctx, int mask_host, int64_t pos_begin, int64_t pos_end);
int64_t frisk_seq_padded_len(const frisk_ctx* ctx);
/* Raw (linear, summable) profile state: int64[frisk_profile_raw_len()] on the device.  To
 * all-reduce across GPUs the caller exports it into its own device buffer (e.g. a torch
 * tensor), runs ONE RCCL all-reduce(sum) on that, and imports the result. */
int64_t frisk_profile_raw_len(const frisk_ctx* ctx);
int frisk_profile_export_device(frisk_ctx* ctx, void* dst_device_int64);
int frisk_profile_import_device(frisk_ctx* ctx, const void* src_device_int64);
/* The same without copies and without the host: *raw = the raw profile's own device buffer (int64[frisk_profile_raw_len()]),
 * *stream = the context's HIP stream (a hipStream_t).  A collective library sums the buffer IN PLACE ON THAT STREAM - RCCL's
 * all-reduce over xGMI, issued under torch.cuda.ExternalStream(*stream) - so that profile_add -> all-reduce -> finalize is
 * one chain of enqueued work.  The call marks the profile as changed (as frisk_profile_import_device does). */
int frisk_profile_device_view(frisk_ctx* ctx, void** raw, void** stream);
/* The one exchange step of a multi-GPU job without any framework: RCCL all-reduce(sum, int64) of the raw profile
 * IN PLACE over `rccl_comm` (an ncclComm_t of the caller's making: one rank per GPU, created with ncclCommInitRank on this
 * context's device), enqueued on the context's stream - profile_add -> all-reduce -> finalize stay one chain, the host does
 * not wait.  rccl_comm = NULL: a single GPU, nothing to do.  librccl is resolved at first use (the instance already in the
 * process if there is one); it is not a link-time dependency of the library. */
int frisk_profile_allreduce(frisk_ctx* ctx, void* rccl_comm);
int frisk_profile_export_host(frisk_ctx* ctx, int64_t* dst_host);
int frisk_profile_import_host(frisk_ctx* ctx, const int64_t* src_host);
int frisk_profile_finalize(frisk_ctx* ctx);
/* The finished profile in the reference's terms: symmetric counts in profile layout and the
 * three metadata values of L356-359.  Any output pointer may be NULL. */
int frisk_profile_get(frisk_ctx* ctx, int64_t* sym_counts, int64_t* total_len, int64_t* ex_max,
                      int64_t* nn_total);
/* Install a finished profile (e.g. from a cache file; replaces pickle.load at L1437-1439). */
int frisk_profile_set(frisk_ctx* ctx, const int64_t* sym_counts, int64_t total_len, int64_t ex_max,
                      int64_t nn_total);

/* ---- phase B: window scan (loop L1478-1494) -------------------------------------------------
 * Candidate windows of the resident batch are numbered in output order (scaffold order, then j
 * ascending, L228); frisk_scan_plan returns their number.  frisk_scan scores candidates
 * [c0,c1) (c1 = -1: to the end) and writes one entry per candidate, index (c - c0); entries
 * whose status lacks FRISK_ROW_KEPT are windows the reference drops (no row).  cap = length of
 * the output arrays (FRISK_E_CAP if < c1-c0).  Nullable: pi/si/cri (required with
 * FRISK_SCAN_RIP), dbg_counts (cap x frisk_profile_len() uint32: the window's k-mer counts),
 * dbg_meta (cap x 3: totalLen, exMax, nnTotal of L356-359). */
int frisk_scan_plan(frisk_ctx* ctx, int32_t w, int32_t inc, uint32_t flags, int64_t* n_candidates);
int frisk_scan(frisk_ctx* ctx, int32_t w, int32_t inc, uint32_t flags, int64_t c0, int64_t c1,
               int64_t cap, int32_t* seq_index, int64_t* start, int64_t* stop, uint32_t* status,
               double* kld, double* gc, double* pi, double* si, double* cri,
               uint32_t* dbg_counts, int64_t* dbg_meta);

/* Test utility (kmax <= 6): the two distributions IvomBuild returns for each candidate window of [c0,c1) (L1481-1482, L369-457) -
 * window-side and genome-side interpolated probabilities of the window's present max-mers, each normalised to sum 1 -
 * as dense vectors of 4^kmax doubles per candidate (0 for absent max-mers; all 0 for windows the reference drops). */
int frisk_scan_ivom(frisk_ctx* ctx, int32_t w, int32_t inc, uint32_t flags, int64_t c0, int64_t c1, int64_t cap,
                    double* window_ivom, double* genome_ivom);

/* Diagnostics of the most recent frisk_scan (results never depend on them).  The default K = 8 kernel counts max-mers in
 * 4- or 8-bit counters and hands a window with a more frequent max-mer (low-complexity sequence) to the next wider form:
 * which = 0: counter width of the bulk launch (4 or 8; 16 = the narrow kernel was not used),
 *         1: windows handed from 4-bit to 8-bit counters,   2: windows handed on to 16-bit counters,
 *         3: row segments (2: the last sixteenth of a long scan ran on a second stream while the rows of the rest went to the host),
 *         4: 1 = the 4-bit bulk launch counted the max-mers of period <= 4 in its side table (FRISK_SCAN_SIDE4 / picked by the sample). */
int64_t frisk_last_scan_stat(const frisk_ctx* ctx, int which);

/* The rows of the score table as text, exactly as the reference's scan loop writes them (L1487-1494): tab-separated
 * name, start, stop, windowKLD, GC[, PI, SI, CRI], one line per row, every value as Python 2's str() prints it (floats:
 * 12 significant digits, '.0' on integral values, nan / inf).  Host-side, multi-threaded; replaces a per-row Python loop
 * that costs tens of seconds at 3 M rows.  names: one per scaffold; kld_is_int0 (nullable): rows whose KLD is the int 0
 * (L465: no max-mer); pi/si/cri: all three or none.  Returns a malloc'd, NUL-terminated buffer (frisk_free) and its length. */
char* frisk_format_rows(int64_t n_rows, const char* const* names, const int32_t* seq_index, const int64_t* start,
                        const int64_t* stop, const uint8_t* kld_is_int0, const double* kld, const double* gc,
                        const double* pi, const double* si, const double* cri, int64_t* out_len);
void frisk_free(void* ptr);

/* Host-only: the 2-state, one-feature Gaussian HMM that segments the KLD track - what the reference asks of hmmlearn's
 * GaussianHMM(n_components=2, covariance_type="full") at L1539-1541 (fit on all non-NaN window scores stacked as one sequence)
 * and at L769 inside hmm2BED (Viterbi path per scaffold).  The model is the one frisk_amd/hmm.py documents (Baum-Welch with
 * hmmlearn's default priors, deterministic 2-means start, state 0 = the lower mean); this is its multi-threaded form for the
 * 3 M windows of a GRCh38-sized run.  frisk_hmm_fit: x[n] finite; n_iter / tol / min_covar / covars_prior as hmmlearn's
 * arguments (10, 1e-2, 1e-3, 1e-2); outputs means[2], covars[2], startprob[2], transmat[4] (row-major), the log-likelihood of
 * the last E step and the number of EM rounds run (nullable).  frisk_hmm_viterbi: n_seg sequences x[seg_off[s] .. seg_off[s+1])
 * decoded independently into states[] (0 / 1), one task per sequence. */
int frisk_hmm_fit(const double* x, int64_t n, int32_t n_iter, double tol, double min_covar, double covars_prior, double* means,
                  double* covars, double* startprob, double* transmat, double* loglik, int32_t* iters);
int frisk_hmm_viterbi(const double* x, const int64_t* seg_off, int32_t n_seg, const double* means, const double* covars,
                      const double* startprob, const double* transmat, int8_t* states);

/* Page-locked host memory for result buffers: D2H copies into it are asynchronous and run at PCIe rate
 * (pageable buffers work too, at a fraction of it).  Free with frisk_host_free before frisk_destroy. */
void* frisk_host_alloc(frisk_ctx* ctx, int64_t bytes);
void  frisk_host_free(frisk_ctx* ctx, void* ptr);

/* Timing of the most recent launches on the context's stream, measured with HIP events:
 * which = 0 scan kernel, 1 profile_add kernel, 2 pack kernel.  Returns milliseconds, <0 if none. */
double frisk_last_kernel_ms(const frisk_ctx* ctx, int which);

#ifdef __cplusplus
}
#endif
#endif /* FRISK_HIP_H */
