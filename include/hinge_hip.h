/* libhinge_hip - MI355X (gfx950) implementation of the HINGE filter / maximal / layout hot path.
 *
 * Drop-in boundary.  The reference has no plugin or FFI seam: its three stage programs
 * (src/filter/filter.cpp, src/maximal/maximal.cpp, src/layout/hinging.cpp) call a C++ class,
 * LAInterface (src/include/LAInterface.h:113-233), and a handful of LOverlap methods, and talk to each
 * other through files.  This C ABI is what those call sites bind instead; every entry point names the
 * reference code it replaces.  Plain pointers and sizes only, `int` status (0 = ok, <0 = HINGE_E_*),
 * no exceptions across the boundary, caller-owned host buffers, library-owned device buffers.
 * One context per GPU; a context is not thread-safe; contexts are independent.
 *
 * Data layout handed in (built by the host ingest from the .las in one pass - the replacement of
 * LAInterface::getOverlap, src/lib/LAInterface.cpp:1519-1634):
 *   row_ptr[n_reads+1]  int64   CSR by A read over the arrays below
 *   a_span[n][2]        int32   (abpos, aepos)
 *   b_span[n][2]        int32   (bbpos, bepos) on the FORWARD strand of B, i.e. after the flip of
 *                               LAInterface.cpp:1619-1626
 *   b_flag[n]           uint32  bread | comp << 31
 * holding, for every A read, its overlaps in .las order with SELF-overlaps (aread == bread) removed -
 * the reference excludes them from every pile-up (filter.cpp:537-548).
 */
#ifndef HINGE_HIP_H
#define HINGE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HINGE_OK 0
#define HINGE_E_ARG (-1)       /* bad argument / call order                                   */
#define HINGE_E_DEVICE (-2)    /* HIP runtime error (no GPU, OOM, launch failure)              */
#define HINGE_E_CAPACITY (-3)  /* an internal device buffer overflowed even after regrowing    */
#define HINGE_E_UNDEFINED (-4) /* input on which the reference itself is undefined (e.g. no
                                  read >= 5000 bp in a part: filter.cpp:660-666)               */
#define HINGE_E_RANGE (-5)     /* coordinate outside the read (bin beyond the LDS histogram)   */

typedef struct hinge_ctx hinge_ctx;

/* [filter] keys of nominal.ini as filter.cpp:377-406 reads them (plus reso = 40, filter.cpp:386). */
typedef struct hinge_filter_params {
    int32_t reso;
    int32_t cut_off;
    int32_t min_cov;
    int32_t est_cov;
    int32_t theta;
    int32_t coverage_fraction;
    int32_t min_repeat_annotation;
    int32_t max_repeat_annotation;
    int32_t repeat_annotation_gap;
    int32_t no_hinge_region;
    int32_t hinge_min_support;
    int32_t hinge_bin_pileup;
    int32_t hinge_unbridged;
    int32_t hinge_tolerance;
    int32_t use_qv_mask;       /* already AND-ed with "qual track present" (filter.cpp:409)    */
    int32_t use_coverage_mask;
    int32_t delete_telomere;
} hinge_filter_params;

typedef struct hinge_cov_estimate {
    int32_t cov_est;           /* median of per-read mean coverage, filter.cpp:660-664         */
    int32_t n_long;            /* reads with len >= 5000 in the part                           */
    int64_t total_cov;         /* filter.cpp:652                                               */
    int64_t num_slot;          /* filter.cpp:653                                               */
} hinge_cov_estimate;

/* ---- context ------------------------------------------------------------------------------- */
int hinge_device_count(void);       /* visible HIP devices (0 if the runtime cannot be initialised) */
int hinge_ctx_create(int device, hinge_ctx** out);
/* free / total device memory of the context's GPU in bytes (round 6: `hinge pipeline` keeps a finished stage's buffers for speed and
 * gives them back only when the device is running short; no counterpart in the reference, which has no device)                    */
int hinge_ctx_device_memory(hinge_ctx* ctx, int64_t* free_bytes, int64_t* total_bytes);
void hinge_ctx_destroy(hinge_ctx* ctx);
const char* hinge_last_error(const hinge_ctx* ctx);
/* Run all launches on this hipStream_t (e.g. torch's current stream); NULL = the null stream. */
int hinge_set_stream(hinge_ctx* ctx, void* hip_stream);
int hinge_synchronize(hinge_ctx* ctx);

/* ---- inputs -------------------------------------------------------------------------------- */
/* Read table: replaces LAInterface::openDB/getRead lengths (LAInterface.cpp:133-185,1195-1286) and
 * the QV mask of filter.cpp:340-369 (qv_mask may be NULL = no qual track). Host pointers.        */
int hinge_set_reads(hinge_ctx* ctx, int32_t n_reads, const int32_t* rlen, const int32_t* qv_mask);
/* Pile-ups of one part (one .las file): A reads r_begin..r_end inclusive (filter.cpp:516-517).
 * on_device = 0: host pointers, copied; on_device = 1: device pointers, adopted (caller keeps them
 * alive).  row_ptr always has n_reads+1 entries (empty rows outside the part).                   */
int hinge_set_pileups(hinge_ctx* ctx, int32_t r_begin, int32_t r_end, int64_t n_ovl, const int64_t* row_ptr,
                      const int32_t* a_span, const int32_t* b_span, const uint32_t* b_flag, int on_device);
/* The same, for an ingest that hands over what it saw while it touched every record (the replacement of the per-record work of
 * LAInterface::getOverlap, LAInterface.cpp:1553-1634): span16[n_ovl + hinge_span16_pad()] = abpos | aepos << 16 (the stream of
 * the two coverage passes; NULL if some read is >= 65536 bp), max_pile = the largest pile-up, spans_in_range = 1 if every
 * (abpos, aepos) lies in [0, rlen[A]].  No device sweep over the spans runs before the first pass then.  The pad elements
 * behind span16[n_ovl] may be read by a kernel (their values are never used).                                             */
int hinge_set_pileups_packed(hinge_ctx* ctx, int32_t r_begin, int32_t r_end, int64_t n_ovl, const int64_t* row_ptr,
                             const int32_t* a_span, const int32_t* b_span, const uint32_t* b_flag, const uint32_t* span16,
                             uint32_t max_pile, int spans_in_range, int on_device);
#define HINGE_SPAN16_PAD 256   /* = hinge_span16_pad() */
int hinge_span16_pad(void);
/* The two facts about the current part, whoever produced them (the caller or the library's own sweep). */
int hinge_get_pileup_facts(hinge_ctx* ctx, uint32_t* max_pile, int* spans_in_range);
/* One more fact of the pile-ups an ingest can hand over (round 4: the one-sweep pass then needs NO device sweep before it):
 * nbins[k], k = 0 .. r_end - r_begin: the bins of read r_begin + k's plain coverage profile at `reso`,
 *   0 for an empty pile-up, else max over its overlaps of max(abpos, aepos) / reso + 2   (K of profileCoverage, LAInterface.cpp:4298-4320),
 *   or -1 when a coordinate lies outside [0, rlen] or the pile-up has 65 536+ overlaps (the general kernel then takes the read).
 * Without it the library makes them itself, once per hinge_set_pileups, with k_cov_stats.  After hinge_set_pileups[_packed].        */
int hinge_set_pile_bins(hinge_ctx* ctx, int32_t reso, const int32_t* nbins, int on_device);
/* Optional: use a caller-owned DEVICE buffer int32[n_reads][2] as the all-read mask table (so a
 * collective can fill other ranks' rows in place).  NULL returns to the library-owned table.      */
int hinge_attach_mask_table(hinge_ctx* ctx, int32_t* d_mask_all);
int hinge_attach_mean_cov(hinge_ctx* ctx, int32_t* d_mean_cov);
/* Rows r0..r1 of the mask table from the host (masks another context computed: the parts of a --mlas run on several GPUs see
 * the masks of the parts before them, filter.cpp:778-787 in the reference's sequential loop).                                */
int hinge_set_mask_rows(hinge_ctx* ctx, int32_t r0, int32_t r1, const int32_t* rows);
/* Reset the mask table to (0,0) (the state maskvec has before the first part, filter.cpp:534).   */
int hinge_clear_masks(hinge_ctx* ctx);

/* ---- filter: coverage -> mask -> repeat annotation -> hinges ----------------------------------- */
/* K1: per-read sum of the cutoff-0 coverage bins and the bin count (profileCoverage,
 * LAInterface.cpp:4298-4320, as used by filter.cpp:642-656). Fills mean_cov[i] for len >= 5000.   */
int hinge_filter_stats(hinge_ctx* ctx, const hinge_filter_params* p);
/* Median of mean_cov over [lo, hi] (nth_element, filter.cpp:660-664) and the MIN_COV update of
 * filter.cpp:671-678 done on the device scalar. If `out` is non-NULL the estimate is copied back
 * (synchronises).                                                                                 */
int hinge_filter_median(hinge_ctx* ctx, const hinge_filter_params* p, int32_t lo, int32_t hi, hinge_cov_estimate* out);
/* hinge_filter_stats followed by the median of the part's OWN reads [r_begin, r_end] (filter.cpp:642-678), no host round trip in
 * between.  hist_dev == NULL: as hinge_filter_median (MIN_COV updated on the device; `out` non-NULL copies the estimate back and
 * synchronises).  hist_dev = device uint32[4096 + 2]: as hinge_filter_median_hist, the part's histogram for an all-reduce over
 * ranks and hinge_filter_median_from_hist (`out` ignored).                                                                        */
int hinge_filter_stats_median(hinge_ctx* ctx, const hinge_filter_params* p, uint32_t* hist_dev, hinge_cov_estimate* out);
/* Sharded runs (one context per GPU, reads split by block): the global median without gathering 4 bytes per read.
 * median_hist() histograms the mean coverages of this rank's reads lo..hi into hist_dev[4096 + 2] (device memory:
 * bins, number of values, 1 if a value fell outside [0, 4096)); the caller sums hist_dev over ranks (one 16 KiB
 * all-reduce); median_from_hist() takes the element of rank n/2 and applies the MIN_COV update (filter.cpp:660-678).
 * An out-of-range value makes the next status check fail with HINGE_E_RANGE: gather the means and use
 * hinge_filter_median.                                                                                          */
int hinge_filter_median_hist(hinge_ctx* ctx, const hinge_filter_params* p, int32_t lo, int32_t hi, uint32_t* hist_dev);
int hinge_filter_median_from_hist(hinge_ctx* ctx, const hinge_filter_params* p, const uint32_t* hist_dev);
/* median_from_hist for the n resident parts of a rank in ONE launch (after one all-reduce over all their histograms): part k's
 * histogram is hist_dev + k * row_stride (uint32 words, row_stride >= 4096 + 2), its context ctxs[k].  The contexts share one
 * device and one stream; n <= 16.                                                                                              */
int hinge_filter_median_from_hist_batch(hinge_ctx** ctxs, int32_t n, const hinge_filter_params* p, const uint32_t* hist_dev, int64_t row_stride);
/* The median of each of n resident parts over its OWN reads [r_begin, r_end] in ONE launch (the kernel is a chain of dependent
 * round trips, ~16 us whatever the part's size: paid once for all parts).  hist_dev == NULL: as hinge_filter_median without
 * `out` for every part.  hist_dev != NULL: as hinge_filter_median_hist, part k's histogram at hist_dev + k * row_stride.
 * The contexts share one device and one stream; n <= 16.                                                                    */
int hinge_filter_median_batch(hinge_ctx** ctxs, int32_t n, const hinge_filter_params* p, uint32_t* hist_dev, int64_t row_stride);
/* --restrictreads (filter.cpp:680-694,767-773): keep[n_reads], 0 = the read's coverage and QV masks are emptied
 * (maxend = maxstart, QV.second = QV.first) before the mask is formed; NULL = no restriction.  The caller builds the
 * set (the listed reads plus every B read they overlap).                                                             */
int hinge_set_read_restriction(hinge_ctx* ctx, const uint8_t* keep);
/* Get / set the running MIN_COV (it carries across parts, filter.cpp:677-678).                     */
int hinge_filter_set_min_cov(hinge_ctx* ctx, int32_t min_cov);
int hinge_filter_get_min_cov(hinge_ctx* ctx, int32_t* min_cov);
/* K2: coverage mask (filter.cpp:696-789) + repeat annotation and merge (filter.cpp:796-829) + the
 * fp32 end-coverage gate (filter.cpp:842-865) for reads r_begin..r_end; writes their mask rows.  */
int hinge_filter_mask_annotate(hinge_ctx* ctx, const hinge_filter_params* p);
/* K3: hinge calling (filter.cpp:867-1068) for the reads that passed the gate; reads the whole mask
 * table (masks of B reads).                                                                       */
int hinge_filter_hinges(hinge_ctx* ctx, const hinge_filter_params* p);
/* All of the above for one single-GPU part, asynchronously on the stream, no host round trip.    */
int hinge_filter_run(hinge_ctx* ctx, const hinge_filter_params* p);
/* ---- the one-sweep pass (round 4) --------------------------------------------------------------------------------------
 * filter.cpp sweeps every pile-up twice because the coverage mask needs MIN_COV = max(MIN_COV, median coverage / 3), a
 * whole-part barrier (filter.cpp:642-678) that lies between profileCoverage (filter.cpp:597-598) and the mask loop
 * (filter.cpp:696-829).  Here ONE sweep does both: sweep_batch() predicts MIN_COV from a sample of each part, runs K2 with it -
 * exact for every MIN_COV within +-band (default 1) of the prediction; the ~1 % of reads with a coverage bin or an annotation
 * threshold inside that band emit nothing and are listed - while K2's prefix scan yields the per-read coverage sums, and then
 * runs the exact median on those sums as VERIFICATION (MIN_COV is updated as by hinge_filter_median).  finish_batch() runs
 * the listed reads with the exact MIN_COV (all reads if it fell outside the band).  Results are those of
 * stats + median + mask_annotate, bit for bit.  n <= 16 parts (contexts on one device and one stream), each over its own
 * reads [r_begin, r_end]; asynchronous; status through hinge_filter_check / the getters.  The sweep of up to 8 parts is ONE
 * kernel launch (k_mask_annotate_q20_batch: the chip sweeps part after part without a launch boundary; parts the fast kernel
 * does not take, or in different kernel variants, get a launch each; HINGE_K2_BATCH=0: always).
 *   hist_dev == NULL: the median is finished on this GPU.  hist_dev != NULL (sharded runs): part k's histogram goes to
 *   hist_dev + k * row_stride as by hinge_filter_median_hist; the caller all-reduces and calls
 *   hinge_filter_median_from_hist_batch (which verifies) before finish_batch().
 * With delete_telomere != 0 (the telomere test sums max(cov, MIN_COV): no band) or HINGE_ONE_SWEEP=0 the two calls run the
 * two-sweep pass: sweep_batch = stats + median, finish_batch = mask_annotate.                                                */
int hinge_filter_sweep_batch_async(hinge_ctx** ctxs, int32_t n, const hinge_filter_params* p, uint32_t* hist_dev, int64_t row_stride);
int hinge_filter_finish_batch_async(hinge_ctx** ctxs, int32_t n, const hinge_filter_params* p);
/* The same for one part, synchronously: sweep + verification + guard-band reads, the annotation buffer regrown and the pass
 * repeated when it overflows; `out` (may be NULL) as from hinge_filter_median.  Replaces the sequence
 * hinge_filter_stats_median + hinge_filter_mask_annotate (profileCoverage .. annotation merge, filter.cpp:597-829).          */
int hinge_filter_sweep(hinge_ctx* ctx, const hinge_filter_params* p, hinge_cov_estimate* out);
/* out[0] one-sweep passes verified so far, [1] of which the exact MIN_COV differed from the prediction, [2] of which it fell
 * outside the band (whole part redone); last pass: [3] reads on the guard-band list (-1: it was a two-sweep pass),
 * [4] predicted MIN_COV, [5] exact MIN_COV.  Synchronises.                                                                   */
int hinge_filter_spec_stats(hinge_ctx* ctx, int64_t out[6]);
/* Tests: band (>= 0; -1 keeps), sample size (> 0; else keeps), bias added to every prediction (forces mispredictions).        */
int hinge_debug_spec(hinge_ctx* ctx, int band, int sample, int bias);

/* ---- filter results (host buffers; these synchronise) ---------------------------------------- */
/* mask/cmask: int32[n][2] for reads r_begin..r_end (n = r_end-r_begin+1); flags: bit0 = .cov.flag
 * condition (filter.cpp:758). Any pointer may be NULL.                                            */
int hinge_filter_get_masks(hinge_ctx* ctx, int32_t* mask, int32_t* cmask, uint8_t* flags);
/* Repeat annotations and hinges as CSR over reads r_begin..r_end: off[n+1]; pos/type[off[n]];
 * is_hinge[off[n]] marks the annotations promoted to hinges (same order as .repeat.txt).
 * Call with pos == NULL to get only `off` (to size the arrays).                                   */
int hinge_filter_get_annotations(hinge_ctx* ctx, int64_t* off, int32_t* pos, int32_t* type, uint8_t* is_hinge);
/* Cutoff-`cutoff` coverage bins (the .coverage.txt payload, filter.cpp:599-602) for reads
 * r0..r1 inclusive: nbins[r1-r0+1]; if cov != NULL, bins of read i start at sum(nbins[<i]).      */
int hinge_filter_coverage_bins(hinge_ctx* ctx, int32_t r0, int32_t r1, int32_t reso, int32_t cutoff, int32_t* nbins,
                               int32_t* cov, int64_t cov_cap);
/* .coverage.txt without a second sweep: with coverage_out(1), K2 (hinge_filter_mask_annotate*) also stores the cutoff-0 coverage
 * bins it holds in LDS (profileCoverage(cutoff 0), LAInterface.cpp:4298-4320 as printed by filter.cpp:599-602).
 * get_coverage: off[n + 1] (always filled; host-known layout: bins of read r_begin + k start at cov[off[k]], off[n] = ints
 * needed), nbins[n] = bins of each read, cov[off[n]]; nbins and cov may be NULL (to size the buffer first).            */
int hinge_filter_coverage_out(hinge_ctx* ctx, int enable);
int hinge_filter_get_coverage(hinge_ctx* ctx, int64_t* off, int32_t* nbins, int32_t* cov, int64_t cov_cap);
/* Counters of the last hinge pass: [0] reads that reached hinge calling, [1] annotations resolved on
 * the exact (std::sort-replaying) path, [2] total annotations, [3] total hinges.                  */
int hinge_filter_counters(hinge_ctx* ctx, int64_t out[4]);

/* ---- maximal / layout: overlap trim + classify ------------------------------------------------- */
/* Trace points of the pile-up overlaps set by hinge_set_pileups (same order): trace = concatenated
 * trace bytes as they are in the .las (tbytes = 1 if tspace <= 125 else 2: LAInterface.cpp:607-614),
 * trace_off[n_ovl] = byte offset of each overlap's trace, tlen[n_ovl] = Path.tlen (align.h:126-132).  */
int hinge_set_traces(hinge_ctx* ctx, const uint8_t* trace, int64_t trace_bytes, const int64_t* trace_off, const int32_t* tlen, int tbytes,
                     int on_device);
/* The part form (hinge_trim_classify_part[_full]) straight from the .las image: `image` = the file's bytes as they lie on disk
 * (records - align.h:126-146 without the trace pointer, 40 bytes - and traces interleaved).  The overlaps set by
 * hinge_set_pileups (storage order; self-overlaps are not among them, their records are simply stepped over) are cut into windows
 * of 64: win_base[(n_ovl + 63) / 64 + 1] = byte offset of every window's first record (last entry: where the last overlap's trace
 * ends), rec_rel[n_ovl] = offset of every overlap's record behind its window's win_base.  The kernel then reads tlen, the spans,
 * the strand flag, A and B from the image itself (getOverlap's strand flip, LAInterface.cpp:1619-1626, included): no
 * hinge_set_traces, and 4 instead of 32 column bytes per overlap.  Replaces the source of trim_overlap's arguments in
 * maximal.cpp:780-850; results are those of hinge_set_traces + hinge_trim_classify_part.  Needs hinge_set_reads.
 * hinge_trim_classify / hinge_matching_position (list forms) still need hinge_set_traces.                                    */
int hinge_set_las_image(hinge_ctx* ctx, const uint8_t* image, int64_t image_bytes, const int64_t* win_base, const uint32_t* rec_rel, int tbytes,
                        int on_device);
/* effective_start / effective_end of every read (the .mas file: maximal.cpp:524-531, hinging.cpp:867-874) */
int hinge_set_eff_reads(hinge_ctx* ctx, const int32_t* eff);
/* ProcessAlignment's `trim` argument (maximal.cpp:799-804, hinging.cpp:542-549): 1 (default) with a DAZZ_DB / .las,
 * 0 for FASTA + PAF input, where overlaps have no trace points and the match is classified as it stands.           */
int hinge_set_trim(hinge_ctx* ctx, int trim);
/* ProcessAlignment(match, A, B, ALN_THRESHOLD, THETA, THETA2, trim=true) (maximal.cpp:65-134 ==
 * hinging.cpp:78-147 -> LOverlap::trim_overlap LAInterface.cpp:4552-4683, AddTypesAsymmetric :4721-4806) for
 * n_sel overlaps: sel[j] indexes the pile-up arrays, a_of[j] is its A read.  out[j][10] = eff_ab, eff_ae,
 * eff_bb, eff_be, match type (enum of LAInterface.h:30-33), active, weight, length, start idx, end idx.       */
int hinge_trim_classify(hinge_ctx* ctx, int64_t n_sel, const int64_t* sel, const int32_t* a_of, int32_t aln_threshold, int32_t theta,
                        int32_t theta2, int32_t* out);
/* The same, returning only the match type (one byte per overlap): all `hinge maximal` reads of the result is
 * `type == BCOVERA` (maximal.cpp:805-857), and 1 byte instead of 40 per overlap comes back over PCIe.            */
int hinge_trim_classify_types(hinge_ctx* ctx, int64_t n_sel, const int64_t* sel, const int32_t* a_of, int32_t aln_threshold, int32_t theta,
                              int32_t theta2, uint8_t* type_out);
/* The match type of EVERY overlap of the current part (type_out[n_ovl], storage order), for callers that classify most of
 * them anyway (`hinge maximal`: the best one or two overlaps of every (A, B) pair): one wavefront per A read streams the part
 * with coalesced loads instead of gathering a list (a list in hash-map order costs six scattered cache lines per overlap).   */
int hinge_trim_classify_part(hinge_ctx* ctx, int32_t aln_threshold, int32_t theta, int32_t theta2, uint8_t* type_out);
/* The same with all ten fields of hinge_trim_classify per overlap: out[n_ovl][10], storage order.                          */
int hinge_trim_classify_part_full(hinge_ctx* ctx, int32_t aln_threshold, int32_t theta, int32_t theta2, int32_t* out);
/* Sequential containment resolution of `hinge maximal` (maximal.cpp:780-858), host side, no device work: reads in
 * ascending id; a read that is still active is removed if one of its containers is active at that moment (containers
 * of lower id have their final state by then, those of higher id their initial one).  pairs = n_pairs (a, b) int32 rows,
 * one per selected overlap that classified as BCOVERA, grouped by ascending a; inside a group in the reference's iteration
 * order (only the LAST b of a group is order dependent: it is what .contained.txt prints).  active[n_reads]: in = reads
 * whose mask is at least length_threshold long, out = the maximal-read mask.  containing[n_reads] (may be NULL): the
 * container printed for a removed read, -1 for the others.  With shards, every rank resolves the all-gathered pairs.  */
int hinge_resolve_containment(int32_t n_reads, uint8_t* active, int64_t n_pairs, const int32_t* pairs, int32_t* containing);
/* Host side, no device work: perm[] = where `std::sort(v.begin(), v.end(), compare_overlap)` - run n_sorts times in a row, as
 * maximal.cpp:790-805 does twice and hinging.cpp:560-570 once - leaves the elements of a pair's overlap vector whose keys
 * (aepos - abpos + bepos - bbpos, LAInterface.cpp:4884-4889) are key[0..n): descending keys, equal keys where libstdc++'s
 * introsort puts them (up to 16 elements it is an insertion sort, i.e. stable; beyond that it is not).                       */
int hinge_sort_order_desc(int32_t n, const int64_t* key, int32_t n_sorts, int32_t* perm);
/* Host side, no device work: the overlaps `hinge maximal` (maximal.cpp:615-654, 780-850) / `hinge layout` (hinging.cpp:478-602)
 * hand to ProcessAlignment for the reads [lo, hi), in the reference's order: reads ascending; per read its (A, B) pairs in the
 * ITERATION ORDER of the reference's std::unordered_map<int, std::vector<LOverlap*>> (same container, same insertion sequence);
 * per pair the first one or two (two_matches) elements after std::sort(compare_overlap) was run n_sorts times (2 / 1).
 * Arrays as in hinge_set_pileups (host pointers, row_ptr over all n_reads); self_before[a] (may be NULL) = -1, or the number of
 * read a's kept overlaps that lie in front of its first A == B record in the .las (that record's key takes part in the map's
 * insertion order); accept_a / accept_b (may be NULL): reads whose pairs are walked / B reads that are inserted at all
 * (layout: both = the active reads; maximal: accept_a = the initially active reads).  Returns the number of picks (writes
 * min(that, cap) of them to sel[] = overlap indices and a_of[]; sel == NULL only counts), or a negative HINGE_E_*.            */
int64_t hinge_pick_pairs(int32_t n_reads, const int64_t* row_ptr, const int32_t* a_span, const int32_t* b_span, const uint32_t* b_flag,
                         const int32_t* self_before, const uint8_t* accept_a, const uint8_t* accept_b, int32_t lo, int32_t hi,
                         int32_t two_matches, int32_t n_sorts, int64_t* sel, int32_t* a_of, int64_t cap);
/* LOverlap::GetMatchingPosition (LAInterface.cpp:4498-4546) for nq (overlap, position on A) queries.        */
int hinge_matching_position(hinge_ctx* ctx, int64_t nq, const int64_t* q_ovl, const int32_t* q_pos, int32_t* out);

/* Best-overlap edge selection of `hinge layout` (hinging.cpp:1911-2148): for every active read one walk over its forward and
 * one over its backward matches, each keeping at most one edge.  All arrays are host pointers.
 *   read_active[n_reads]             reads[i]->active after the hinge bookkeeping
 *   match_rec[n_matches][9]          b, reverse_complement_match_, match_type_, active, weight, eff_read_B_match_start_,
 *                                    eff_read_B_match_end_, read_B_match_start_, read_B_match_end_ (LAInterface.h:76-110)
 *   off_fwd / off_bwd[n_reads + 1]   matches_forward[i] / matches_backward[i] = match_rec[off[i] .. off[i + 1]), in
 *                                    compare_overlap_weight order (hinging.cpp:1244-1256)
 *   h_off[n_reads + 1], h_rec[][3]   hinges_vec[i]: pos, type, active        (hinging.cpp:1200-1240, 1650-1675)
 *   k_off[n_reads + 1], k_rec[][2]   new_killed_hinges_vec[i]: pos, type     (hinging.cpp:1262-1321)
 * Out: chosen[2][n_reads] = index into match_rec of the edge kept by the forward ([0][i]) and backward ([1][i]) walk, -1 = dead
 * end or inactive read; chosen_hinge_pos[2][n_reads] = the hinge_pos PrintOverlapToFile2 prints with it; poison_hits[n_matches] =
 * how many killed hinges poisoned the match (= its lines in .edges.skipped).                                                    */
int hinge_select_edges(hinge_ctx* ctx, int32_t n_reads, const uint8_t* read_active, int64_t n_matches, const int64_t* off_fwd,
                       const int64_t* off_bwd, const int32_t* match_rec, const int64_t* h_off, const int32_t* h_rec, const int64_t* k_off,
                       const int32_t* k_rec, int32_t hinge_tolerance, int32_t hinge_slack, int32_t* chosen, int32_t* chosen_hinge_pos,
                       int32_t* poison_hits);

/* Staged launches with no host round trip, for pipelines that put a collective between the stages
 * (multi-GPU).  A pass starts at hinge_filter_stats (which clears the per-pass device scalars and is
 * itself asynchronous); check reports the HINGE_E_* flags raised since then.                       */
int hinge_filter_mask_annotate_async(hinge_ctx* ctx, const hinge_filter_params* p);
int hinge_filter_hinges_async(hinge_ctx* ctx, const hinge_filter_params* p);
/* hinge_filter_hinges_async for n resident parts (contexts on one device and one stream, n <= 8) in one launch per kernel:
 * hinge calling touches 1-2 % of the reads through chains of dependent look-ups, ~19 us per kernel however small the part.   */
int hinge_filter_hinges_batch_async(hinge_ctx** ctxs, int32_t n, const hinge_filter_params* p);
int hinge_filter_check(hinge_ctx* ctx);

/* ---- RCCL between the contexts of one process (round 4) ---------------------------------------------------------------------
 * The executables run a --mlas set on one rank (host thread + context) per visible GPU; what the
 * reference's sequential part loop carries from part to part on the device side - the mask table, filter.cpp:534 / :778-787 -
 * then moves between the GPUs as ONE ncclAllGather over xGMI per wave instead of n (n - 1) host-staged copies.
 * hinge_comm_create: one communicator over the n contexts' devices (ncclCommInitAll).  HINGE_E_DEVICE when two contexts share
 * a device (RCCL takes one rank per device), librccl.so cannot be loaded, or HINGE_HOST_EXCHANGE=1: the caller then keeps
 * exchanging through the host (hinge_set_mask_rows).
 * hinge_comm_exchange_mask_rows: context k owns rows [lo[k], hi[k]] of its mask table (hi < lo: none).  phase 0: the all-gather,
 * after which context k's table also holds the rows of the contexts BEFORE it (the state in which part k's hinges are called);
 * phase 1: the rows of the contexts after it (no further collective).  Synchronises every context's stream.                   */
int hinge_comm_create(hinge_ctx** ctxs, int32_t n);
int hinge_comm_exchange_mask_rows(hinge_ctx** ctxs, int32_t n, const int32_t* lo, const int32_t* hi, int32_t phase);
/* hinge_comm_allgather_rows (round 6): rank k holds counts[k] rows of row_bytes bytes at rows[k] (HOST memory: what its host half made of
 * its device's results - the containment candidates of `hinge maximal`, maximal.cpp:805-857; the classified matches / hinge rows of
 * `hinge layout`, hinging.cpp:917-936, :1694-1704).  Two grouped ncclAllGathers over xGMI on the ranks' streams (the counts, then the
 * rows padded to the largest count) give EVERY rank's device all rows; `out` (host, room for out_cap_rows rows) receives them in
 * rank order - what the reference's sequential part loop would have accumulated - and out_counts[k] (may be NULL) rank k's count as
 * delivered.  The order-dependent resolution stays host code on the gathered rows.  Needs hinge_comm_create first; the
 * executables fall back to plain host concatenation where that was refused (HINGE_HOST_EXCHANGE=1, ranks sharing a device).      */
int hinge_comm_allgather_rows(hinge_ctx** ctxs, int32_t n, const void* const* rows, const int64_t* counts, int32_t row_bytes, void* out, int64_t out_cap_rows,
                              int64_t* out_counts);

/* ---- hinge consensus (consensus/consensus.cpp:77-288) --------------------------------------------------
 * The per-contig pile-up vote over base-level realignments.  Replaces, for the alignments the caller selected:
 *   LAInterface::recoverAlignment -> computeTracePTS -> iter_np   lib/LAInterface.cpp:4125-4244, :3410-3506, :3152-3404
 *     (Myers' O(np) waves between successive trace points, one GPU lane per ~100 x ~100-base segment, the reference's
 *     trace-back with re-sliding: the SAME indel list, entry for entry)
 *   LAInterface::getAlignmentTags + chop_end(100)                  lib/LAInterface.cpp:3709-3905, consensus.cpp:27-45
 *   the column vote and the base calls                             consensus.cpp:163-283
 * What stays with the caller (hinge_amd/host/consensus_main.cpp): reading the two DBs and the .las, the per-contig
 * std::sort(compare_overlap_aln) (hinge_sort_order_desc gives libstdc++'s order) and remove_multialign's count
 * (consensus.cpp:126-150), the FASTA / stdout text.                                                                         */
typedef struct hinge_cns_alignment {
    int32_t aread, bread;                 /* contig in the draft DB, read in the read DB (trimmed ids, as in the .las)          */
    int32_t comp;                         /* flags & 1                                                                          */
    int32_t abpos, aepos, bbpos, bepos;   /* as in the .las record: B in the complemented frame when comp (NOT flipped)         */
    int32_t tlen;                         /* trace values of this alignment: (diffs, B advance) pairs, 16 bit each              */
    int64_t trace_off;                    /* its first value in the trace array                                                  */
} hinge_cns_alignment;
typedef struct hinge_cns_stats {          /* what consensus.cpp:272-278 prints per contig                                        */
    int64_t sum_coverage;
    int32_t contig_length, good_bases, insertions, deletions, low_coverage_bases, consensus_length;
} hinge_cns_stats;
/* which: 0 = draft DB (the A reads), 1 = read DB.  rlen[n], boff[n] (byte offset of read i in bps), bps = the .bps file's bytes
 * (2 bits per base, first base in a byte's top bits: DB.c Compress_Read).  Host buffers; copied.                               */
int hinge_consensus_set_db(hinge_ctx* ctx, int32_t which, int32_t n, const int32_t* rlen, const int64_t* boff, const uint8_t* bps, int64_t bps_bytes);
/* Realign, vote and call: every alignment given takes part (in any order).  trace: 16-bit values (byte traces widened, as
 * Decompress_TraceTo16 does).  HINGE_E_RANGE when a segment needs more edit operations than the largest `diffs` its
 * alignment's trace records - where the reference overruns the arrays it sized from that number (LAInterface.cpp:3444-3466). */
int hinge_consensus_run(hinge_ctx* ctx, int64_t n_aln, const hinge_cns_alignment* alns, const uint16_t* trace, int64_t n_trace, int32_t tspace);
/* The consensus string of one contig (no header, no newline; lower case = coverage below 3, consensus.cpp:232-238).
 * out may be NULL to ask for *len only.                                                                                        */
int hinge_consensus_get_contig(hinge_ctx* ctx, int32_t contig, char* out, int64_t cap, int64_t* len, hinge_cns_stats* stats);
/* chop_end's return value per alignment, in the order given to run() (consensus.cpp:176-177 prints them).                     */
int hinge_consensus_get_offsets(hinge_ctx* ctx, int32_t* offsets);
/* Tests: the recovered indel list of one alignment (LAlignment::trace after recoverAlignment).                                */
int hinge_consensus_get_indels(hinge_ctx* ctx, int64_t aln, int32_t* out, int64_t cap, int64_t* n);

/* ---- hinge draft (consensus/draft.cpp:125-715) ----------------------------------------------------------
 * The base-level work of stitching a contig from the reads along a path of the layout graph.  Both DB slots of
 * hinge_consensus_set_db hold the READ DB (a draft aligns reads with reads).  What stays with the caller
 * (hinge_amd/host/draft_main.cpp): the .edges.list path file, which alignment belongs to which edge (draft.cpp:162-178, :262-275),
 * the way points and lanes carried from read to read through the mappings (:415-495), the ladders (:540-556), the coverage
 * profile that picks a ladder's template (:573-590), prefix / suffix / overhang and the cuts (:520-531, :700-712), the FASTA.
 *
 * hinge_draft_mappings: LAInterface::recoverAlignment + getAlignmentTags + get_mapping (LAInterface.cpp:4125-4244, :3709-3905,
 *   draft.cpp:70-87, :213-216, :394-397) for n_aln alignments (as hinge_consensus_run takes them).  mapping[map_off[i] + k], k <
 *   aepos - abpos (map_off[i + 1] - map_off[i] must be that): the number of B bases in the columns in front of the column of A
 *   base abpos + k - the forward-strand map; bit 31 is set when that column holds a gap in B (with it the caller derives the
 *   map of the reverse-complemented tags of a strand-1 edge: draft.cpp:292-298).  HINGE_E_RANGE as hinge_consensus_run. */
int hinge_draft_mappings(hinge_ctx* ctx, int64_t n_aln, const hinge_cns_alignment* alns, const uint16_t* trace, int64_t n_trace, int32_t tspace,
                         const int64_t* map_off, uint32_t* mapping);
/* One member of a ladder: bases [start, end) of read `read` in its strand frame (strand 1: of its reverse complement). */
typedef struct hinge_draft_rung { int32_t read, strand, start, end; } hinge_draft_rung;
/* hinge_draft_ladders: for every ladder (members rungs[rung_off[l] .. rung_off[l + 1]), 1 .. 64 of them) the consensus of
 *   draft.cpp:597-691: every member aligned to member template_rung[l] with falcon's banded O(ND) aligner (lib/DW_banded.c:97-311,
 *   band_tolerance = 150 in the reference), alignment tags with a leading 'T' column (lib/falcon.c:68-125), get_cns_from_align_tags
 *   over template length + 1 positions with min_cov 1 (lib/falcon.c:246-517) - ties, the link-index quirk of its last base and the
 *   lower case of thinly covered bases included.  out_off[l] = where ladder l's string goes in `out` (caller-laid-out slots of at
 *   least 2 * (template length + 1) bytes: out_off[n_ladders] = the buffer's size), out_len[l] = its length.
 *   Limits (the reference has none; draft.cpp's ladders are `[draft] tspace` ~ 900-base windows at the data set's coverage):
 *   HINGE_E_CAPACITY: more than 64 members (one lane per member in the vote), a member of 32768+ bases, or member + template beyond
 *   ~61 000 bases together (the aligner's V / U arrays - 16-bit cells, 8 x 0.3 x (q + t) bytes - and both sequences at 2 bits per base
 *   live in the CU's 160 KB of LDS); HINGE_E_RANGE: 255+
 *   inserted bases in a row (the reference's tags are undefined there, falcon.c:96); HINGE_E_UNDEFINED: its assert(g_best_score != -1).
 *   Any of them fails the CALL (all ladders): `draft_assembly` stops as the reference does on its own asserts. */
int hinge_draft_ladders(hinge_ctx* ctx, int64_t n_ladders, const int64_t* rung_off, const hinge_draft_rung* rungs, const int32_t* template_rung, int32_t band_tolerance,
                        const int64_t* out_off, char* out, int32_t* out_len);

/* Per-kernel timing with HIP events recorded around every launch on the context's stream.
 * enable(max_launches > 0) starts a fresh recording; report() synchronises and returns total ms and
 * launch count per kernel id in [0, hinge_profile_kernels()).  select() restricts the events to the
 * kernel ids whose bit is set (default: all); two events per launch cost a few microseconds of stream
 * time each, so a timed region brackets only the kernel it prices.                                 */
int hinge_profile_enable(hinge_ctx* ctx, int max_launches);
int hinge_profile_select(hinge_ctx* ctx, uint32_t kernel_mask);
int hinge_profile_kernels(void);
const char* hinge_profile_kernel_name(int id);
int hinge_profile_report(hinge_ctx* ctx, double* total_ms, int64_t* count);

/* ---- device event timing helper for bench.py (HIP events on the ctx stream) -------------------- */
int hinge_timer_start(hinge_ctx* ctx);
int hinge_timer_stop_ms(hinge_ctx* ctx, float* ms);

/* ---- test hooks (tests/ only; no product code calls them) ---------------------------------------- */
/* 1 = every scanned annotation through the serial k_hinge_exact, 2 = always replay the pile-up's std::sort order in LDS. */
int hinge_debug_force_exact(hinge_ctx* ctx, int mode);
/* Run the general k_mask_annotate even where k_mask_annotate_q20 applies. */
int hinge_debug_force_general_mask(hinge_ctx* ctx, int on);
/* out[0] = reads the last K2 pass handed from the fast kernel to the general one. */
int hinge_debug_fallback_reads(hinge_ctx* ctx, int64_t* out);
/* out[0..1] = undecided annotations the last hinge pass put through the half-size / full-size k_hinge_call. */
int hinge_debug_heavy_items(hinge_ctx* ctx, int64_t* out);
/* pos_out[k] = position of element k after std::sort(compare_overlap) of n keys, through the wavefront-parallel replay. */
int hinge_debug_pileup_order(hinge_ctx* ctx, int32_t n, const int32_t* keys, int32_t* pos_out);

#ifdef __cplusplus
}
#endif
#endif /* HINGE_HIP_H */
