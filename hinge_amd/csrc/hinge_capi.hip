// C ABI of libhinge_hip (include/hinge_hip.h): context, device memory, launch plumbing.
// No CPU fallback lives here: without a usable HIP device hinge_ctx_create fails with
// HINGE_E_DEVICE and every other entry point needs a context.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstddef>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/hinge_hip.h"
#include "filter_kernels.h"
#include "hinge_call_kernel.h"
#include "align_kernels.h"
#include "select_kernel.h"
#include "consensus_kernels.h"
#include "draft_kernels.h"

using namespace hinge;

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    bool owned = true;
};

struct hinge_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    int n_cu = 256;

    int32_t n_reads = 0, max_rlen = 0;
    DevBuf rlen, qv_mask;   // int32[n], int2[n]
    bool has_qv = false;

    int32_t r_begin = 0, r_end = -1;
    int64_t n_ovl = 0;
    DevBuf row_ptr, a_span, b_span, b_flag;

    DevBuf mask_own;          // int2[n_reads]
    int2* mask = nullptr;     // active table (own or attached)
    bool mask_attached = false;   // mask points at a caller-owned table (hinge_attach_mask_table)
    DevBuf mean_own;
    int* mean_cov = nullptr;
    bool mean_attached = false;
    DevBuf cmask, rflags, nbins0, keep;
    int nbins0_reso = -1;               // reso k_cov_stats last filled nbins0[] at for the current pile-ups (-1: not yet)
    DevBuf span16;             // 16|16-bit copy of a_span (abpos | aepos << 16) for the two streaming kernels
    bool use_span16 = false;   // every read < 65536 bp and every coordinate inside its read (k_pileup_facts)
    int no_span16 = 0;         // HINGE_NO_SPAN16=1: keep the streaming kernels on the int32 spans
    bool has_keep = false;
    DevBuf anno_buf, anno_off, anno_cnt, hinge_flag, work_list, heavy_list, heavy2_list, fallback_list, bucket_list, k2_heads;
    DevBuf rd_head, rd_next2, rl2;      // round 6: the second-tier items chained by read (HingePart::rd_head ...)
    unsigned pass_stamp = 0;            // ... and the stamp of the last hinge launch (rd_head entries of other launches are stale)
    int group_reads = 1;                // HINGE_CALL_GROUP=0: k_hinge_call<CAP> draws items one by one (round 5)
    int hinge_lean = 1;                 // HINGE_CALL_LEAN=0: the full-size k_hinge_call<2048> behind the light kernel instead of the LEAN one
    unsigned anno_cap = 0;               // annotation slots (a multiple of N_SHARD: every shard allocates from its own 1 / N_SHARD of them)
    unsigned work_cap = 0;               // work-list slots (the list is interleaved over the shards: n_reads + room for their imbalance)
    DevBuf exact_queue;
    unsigned exact_cap = 0;
    DevBuf arena;
    unsigned long long arena_cap = 0;
    DevBuf scalars;   // see Scalars
    // one-sweep pass (round 4, see MODE_SPEC in filter_kernels.h)
    DevBuf final_batch;                       // k_mask_final_batch's arguments (of the batch's first context)
    MaskFinalBatch final_batch_host;
    bool final_batch_valid = false;
    size_t lds_attr_final = 0;
    int final_batched = 1;                    // HINGE_FINAL_BATCH=0: one MODE_FINAL launch per part
    int count_waves = 2;                      // wavefronts per work-list read in k_hinge_count (HINGE_COUNT_WAVES=4: rounds 1-3)
    int hinge_light = 1;                      // HINGE_CALL_LIGHT=0: every open annotation straight to k_hinge_call<CAP> (rounds 1-3)
    int hinge_mini = 0;                       // HINGE_CALL_MINI=1: a quarter-size (1024-overlap) instance behind the light kernel, four workgroups per CU (measured: no gain - the replay is bound by its slowest item, not by the workgroups in flight)
    int light_occ = 0;
    int k2_batch = 1;                         // HINGE_K2_BATCH=0: one k_mask_annotate_q20 launch per part of a batched sweep
    int k2_steal = 2;                         // the persistent workgroups of a batched launch: 0 stay with their own part, 1 go round the parts from their own, 2 all sweep part 0, 1, ... (HINGE_K2_STEAL)
    DevBuf cov_tot, redo_list, spec_sample;   // int[n_reads] coverage sums, int[n_reads] guard-band list, int[spec_ns] sample means
    int spec_band = 1;        // the sweep is exact for every MIN_COV within +- this of the prediction (HINGE_SPEC_BAND)
    int spec_ns = 1024;       // reads k_spec_predict samples per part (HINGE_SPEC_SAMPLE; round 5: 1024 instead of 4096 - the sample median moves by +-0.6 of a coverage unit, MIN_COV is a third of it and the band is +-1: 21.5 -> 16.7 us, no miss)
    int spec_bias = 0;        // tests: added to the prediction (hinge_debug_spec)
    int pass_mode = 0;        // what the current pass's first sweep was: 0 classic, 1 one-sweep through k_mask_annotate_q20<SPEC>, 2 one-sweep through the general kernel
    DevBuf med;       // median histogram scratch (k_median_hist)
    DevBuf wave_totals;   // k_cov_stats per-wave (total_cov, num_slot) partials
    int n_wave_totals = 0;
    size_t lds_attr_set = 0;
    int force_exact = 0;
    int trim = 1;                 // ProcessAlignment's trim flag: 0 for PAF input (no trace points)
    int force_general_mask = 0;
    int one_sweep = 1;            // HINGE_ONE_SWEEP=0: hinge_filter_sweep_batch_async always runs the two-sweep pass
    bool debug_paths = false;     // HINGE_DEBUG_PATHS: path counters (same-line global atomics, ~12 ns each: off by default)
    bool min_cov_pending = false;   // hinge_filter_set_min_cov is applied by the next launch that needs it
    int min_cov_value = 0;
    std::vector<int> h_rlen;      // host copy of the read lengths (length buckets of K2)
    unsigned max_pile = 0;        // facts about the current part's pile-ups (k_pileup_facts)
    bool spans_in_range = false;
    // tests: run the general K2 kernel where the q20 kernel would be chosen

    // trim / classify (maximal, layout)
    DevBuf trace, trace_off, tlen, eff_reads, pair_sel, pair_a, pair_out;
    DevBuf bspan16;             // 16|16 copy of b_span for k_hinge_count / k_hinge_call_light, made before the part's first hinge pass
    int bspan16_state = 0;      // 0 not tried for the current pile-ups, 1 usable, -1 not usable
    DevBuf img_row_base, img_rec_rel;   // hinge_set_las_image: the part form reads the .las image itself (k_trim_classify_image)
    bool image_set = false;
    int k2_wgs = 0;              // workgroups of k_mask_annotate_q20 (0: as many as the GPU holds at once, capped by the part's reads); HINGE_K2_WGS
    int k2_order_bp = 1024;                   // bucket width of the longest-first order of the one-slot reads (HINGE_K2_ORDER_BP)
    int k2_occ_lds = -1, k2_occ = 0;          // occupancy calculator: workgroups per CU at that many bytes of dynamic LDS
    std::vector<int> k2_list;                 // host copy of bucket_list (the upload is asynchronous)
    std::vector<int> k2_c1;                   // the one-slot reads in storage order (for the XCD-contiguous deal, see launch_mask_annotate)
    std::vector<unsigned char> k2_heavy;      // per read of the part: its pile-up is far deeper than the part's mean (see hinge_set_pileups)
    int k2_heavy_mode = 2;                    // HINGE_K2_HEAVY: 2 = the deep pile-ups at even intervals over the first 60 % of an XCD's sequence, 1 = first, 0 = where the storage order puts them
    int k2_deal = 1;                          // 1: deal every XCD a contiguous eighth of the one-slot reads, in storage order (HINGE_K2_DEAL=0: longest first, round 2's order)
    int k2_deal_heads = -1, k2_deal_rot = -1; // what the list on the device was dealt for
    int n_class[3] = {0, 0, 0};               // bucket_list = [reads needing 1 (longest first) | 2 | 4 LDS slots of a K2 workgroup] of the current part
    unsigned k2_head_base[K2_MAX_HEADS] = {}; // value of every item counter of k_mask_annotate_q20 (DevBuf k2_heads) before its next launch
    DevBuf k2c;                  // K2Const of k_mask_annotate_q20 in device memory
    K2Const k2c_host;            // what was uploaded last
    bool k2c_valid = false;
    bool trace_padded = false;   // the trace buffer is the library's own copy with 8 spare bytes behind it
    int64_t trace_bytes = 0;
    int tbytes = 1;

    // K2's optional coverage-bin output (hinge_filter_coverage_out)
    bool cov_out_on = false;
    DevBuf cov_buf, cov_off_d, cov_nb;
    std::vector<int64_t> h_cov_off;          // [nr + 1] over the reads of the current part
    int cov_key[4] = {-1, -1, -1, -1};       // r_begin, r_end, reso, cut_off the layout was made for
    bool cov_valid = false;                  // the last K2 pass wrote the bins

    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    // per-kernel HIP-event timing (bench.py roofline): pairs recorded around every launch
    bool prof_on = false;
    uint32_t prof_mask = 0xffffffffu;   // kernel ids that get events (hinge_profile_select)
    std::vector<hipEvent_t> prof_pool;
    size_t prof_used = 0;
    std::vector<int> prof_kid;

    struct CnsState* cns = nullptr;     // `hinge consensus` (consensus_capi.inc)
    struct DraftState* draft = nullptr; // `hinge draft` (draft_capi.inc)
    void* comm = nullptr;               // ncclComm_t of this context among the contexts of its process (comm_capi.inc)
    int comm_rank = -1, comm_size = 0;
    DevBuf comm_stage;                  // all-gathered mask rows [comm_size + 1][S][2]
    DevBuf comm_rows;                   // hinge_comm_allgather_rows: [comm_size + 1][S] rows (the last block is this rank's send block) + the counts
    int64_t comm_staged_S = 0;          // rows per rank of the last phase-0 exchange (0: nothing staged), and every rank's rows
    std::vector<int32_t> comm_staged_lo, comm_staged_hi;   // in it: phase 1 places from that stage and must be given the same layout
};

enum KernelId { KID_STATS = 0, KID_MEDIAN, KID_MASK_ANNOTATE, KID_MASK_FALLBACK, KID_HINGE_COUNT, KID_HINGE_CALL, KID_HINGE_EXACT, KID_COVERAGE_BINS, KID_TRIM_CLASSIFY,
                KID_PILEUP_FACTS, KID_MATCHING_POSITION, KID_SELECT_EDGES, KID_SPEC_PREDICT, KID_MASK_FINAL, KID_CNS_REALIGN, KID_CNS_COLUMNS, KID_CNS_VOTE, KID_CNS_CALL, KID_DRAFT_ALIGN, KID_DRAFT_CNS, KID_COUNT };
static const char* const KERNEL_NAMES[KID_COUNT] = {"k_cov_stats", "k_median_hist", "k_mask_annotate", "k_mask_annotate_fallback", "k_hinge_count", "k_hinge_call", "k_hinge_exact",
                                                     "k_coverage_bins", "k_trim_classify", "k_pileup_facts", "k_matching_position", "k_select_edges", "k_spec_predict",
                                                     "k_mask_annotate_final", "k_cns_realign", "k_cns_columns", "k_cns_vote", "k_cns_call", "k_draft_align", "k_draft_cns"};

struct ProfScope {
    hinge_ctx* c;
    bool on;
    ProfScope(hinge_ctx* ctx, int kid) : c(ctx), on(false) {
        if (c->prof_on && ((c->prof_mask >> kid) & 1u) && c->prof_used + 2 <= c->prof_pool.size()) {
            on = true;
            c->prof_kid.push_back(kid);
            (void)hipEventRecord(c->prof_pool[c->prof_used], c->stream);
        }
    }
    void stop() {
        if (on) {
            (void)hipEventRecord(c->prof_pool[c->prof_used + 1], c->stream);
            c->prof_used += 2;
            on = false;
        }
    }
    ~ProfScope() { stop(); }
};

static_assert(HINGE_SPAN16_PAD == (LOADS_IN_FLIGHT / 2) * WAVE, "elements behind span16[n_ovl] a kernel may read (never uses)");
static const int K2_SHORT_RLEN = 19000;   // 20-bp bins of a 19 kb read + hot words + pads = 1268 ints per wavefront, 19.8 KiB per workgroup: eight
                                          // workgroups per CU, the kernel's register budget (99 % of the bench part's reads need one slot)

// words of LDS per wavefront slot of k_mask_annotate_q20: 20-bp bins of the longest "short" read + the hot words
static int k2_slot_ints(const hinge_ctx* ctx) {
    const int len = std::min(ctx->max_rlen, K2_SHORT_RLEN);
    return (((len / 20 + 1 + 3) & ~3) + 4) + 4 * WAVE;
}

// device scalars, one allocation
struct Scalars {
    // ---- cleared by ONE memset at the start of every pass (SCALARS_RESET_BYTES) ----
    unsigned long long totals[2];       // total_cov, num_slot
    unsigned long long arena_used;
    unsigned shards[2 * N_SHARD * SHARD_STRIDE];   // [0][s]: annotation allocator of shard s, [1][s]: its work-list length (filter_kernels.h N_SHARD; counters 128 bytes apart)
    unsigned fallback_count;            // reads k_mask_annotate_q20 handed back to the general kernel
    unsigned exact_count;
    unsigned work_next;                 // k_hinge_call's work-list cursor
    unsigned heavy_count;               // annotations the count-only sweep could not decide, pile-up <= PO_CAP_SMALL (front of heavy_list)
    unsigned work_next_big;             // the same two for the pile-ups beyond PO_CAP_SMALL (back of heavy_list, PO_CAP instance)
    unsigned heavy_count_big;
    unsigned work_next_light;           // k_hinge_call_light's cursor over both ends of heavy_list
    unsigned heavy2_count;              // what it passed on to k_hinge_call<CAP>: front / back of heavy2_list
    unsigned heavy2_count_big;
    unsigned work_next_small;           // the 1024-overlap instance's cursor over the front of heavy2_list (round 5)
    unsigned rl2_count;                 // round 6: reads with second-tier items (front / back of rl2, like heavy2_list): k_hinge_call<CAP> draws reads
    unsigned rl2_count_big;
    unsigned redo_count;                // one-sweep pass: length of the guard-band list
    int spec_state;                     // one-sweep pass: 1 = the exact MIN_COV fell outside the band (SpecVerify)
    int status;
    // ---- persistent across passes ----
    int est[2];                         // cov_est, n_long
    int min_cov;
    int pad;
    unsigned facts[2];                  // k_pileup_facts: largest pile-up, out-of-range flag
    int bins_status;                    // hinge_filter_coverage_bins' own range flag
    int pad2;
    unsigned dbg[24];                   // k_hinge_call path counters (cumulative; diagnostics only); [8..] HINGE_TIMING builds ([16..]: the replay's phases)
    int spec_min_cov;                   // one-sweep pass: the MIN_COV the sweep ran with (k_spec_predict)
    unsigned spec_ticket;               // k_spec_predict's workgroup counter (zero between launches)
    unsigned spec_stats[3];             // cumulative: passes verified, exact != predicted, exact outside the band
};
static const size_t SCALARS_RESET_BYTES = offsetof(Scalars, est);

#define CK(call)                                                                                         \
    do {                                                                                                 \
        hipError_t _e = (call);                                                                          \
        if (_e != hipSuccess) {                                                                          \
            ctx->err = std::string(#call) + ": " + hipGetErrorString(_e);                                \
            return HINGE_E_DEVICE;                                                                       \
        }                                                                                                \
    } while (0)

// totals of the sharded counters of a Scalars copy
static unsigned shard_sum(const unsigned* shards, int which) { unsigned t = 0; for (int k = 0; k < N_SHARD; k++) t += shards[(which * N_SHARD + k) * SHARD_STRIDE]; return t; }

static int fail(hinge_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    return code;
}

static int ensure(hinge_ctx* ctx, DevBuf& b, size_t bytes) {
    if (b.owned && b.p && b.bytes >= bytes) return HINGE_OK;
    if (b.owned && b.p) { (void)hipFree(b.p); b.p = nullptr; }
    b.owned = true;
    b.bytes = std::max<size_t>(bytes, 16);
    CK(hipMalloc(&b.p, b.bytes));
    return HINGE_OK;
}

static void release(DevBuf& b) {
    if (b.owned && b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
    b.owned = true;
}

static Scalars* sc(hinge_ctx* ctx) { return (Scalars*)ctx->scalars.p; }

static FilterDev to_dev(const hinge_filter_params* p) {
    FilterDev d;
    d.reso = p->reso; d.cut_off = p->cut_off; d.theta = p->theta;
    d.cov_frac = p->coverage_fraction; d.min_ra = p->min_repeat_annotation; d.max_ra = p->max_repeat_annotation;
    d.ra_gap = p->repeat_annotation_gap; d.nhr = p->no_hinge_region;
    d.sup = p->hinge_min_support; d.pil = p->hinge_bin_pileup; d.unb = p->hinge_unbridged; d.tol = p->hinge_tolerance;
    d.bin_len = 2 * p->hinge_tolerance;   // filter.cpp:405
    d.use_qv = p->use_qv_mask; d.use_cov = p->use_coverage_mask; d.del_telo = p->delete_telomere;
    d.est_cov = p->est_cov;
    d.ablate = 0;
#ifdef HINGE_ABLATE
    if (const char* a = getenv("HINGE_ABLATE_PHASE")) d.ablate = atoi(a);   // ablation builds only (tools/ablate_k2.sh)
#endif
    return d;
}

static int check_params(hinge_ctx* ctx, const hinge_filter_params* p) {
    if (!p) return fail(ctx, HINGE_E_ARG, "params == NULL");
    if (p->reso <= 0) return fail(ctx, HINGE_E_ARG, "reso must be > 0");
    if (p->coverage_fraction == 0) return fail(ctx, HINGE_E_ARG, "coverage_frac_repeat_annotation == 0 divides by zero in the reference");
    return HINGE_OK;
}

extern "C" {

int hinge_ctx_create(int device, hinge_ctx** out) {
    if (!out) return HINGE_E_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return HINGE_E_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return HINGE_E_DEVICE;
    hinge_ctx* ctx = new hinge_ctx();
    ctx->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->n_cu = prop.multiProcessorCount;
    if (hipMalloc(&ctx->scalars.p, sizeof(Scalars)) != hipSuccess) { delete ctx; return HINGE_E_DEVICE; }
    ctx->scalars.bytes = sizeof(Scalars);
    (void)hipMemset(ctx->scalars.p, 0, sizeof(Scalars));
    if (const char* g = getenv("HINGE_DEBUG_GENERAL_MASK")) ctx->force_general_mask = atoi(g);
    if (const char* g = getenv("HINGE_NO_SPAN16")) ctx->no_span16 = atoi(g);
    if (const char* g = getenv("HINGE_K2_WGS")) ctx->k2_wgs = std::max(1, atoi(g));
    if (const char* g = getenv("HINGE_K2_ORDER_BP")) ctx->k2_order_bp = std::max(1, atoi(g));
    if (const char* g = getenv("HINGE_K2_DEAL")) ctx->k2_deal = atoi(g);
    if (const char* g = getenv("HINGE_K2_HEAVY")) ctx->k2_heavy_mode = atoi(g);
    if (const char* g = getenv("HINGE_DEBUG_FORCE_EXACT")) ctx->force_exact = atoi(g);   // 1: serial exact kernel, 2: exact replay in LDS (tests)
    ctx->debug_paths = getenv("HINGE_DEBUG_PATHS") != nullptr;
    if (const char* g = getenv("HINGE_ONE_SWEEP")) ctx->one_sweep = atoi(g);
    if (const char* g = getenv("HINGE_FINAL_BATCH")) ctx->final_batched = atoi(g);
    if (const char* g = getenv("HINGE_K2_BATCH")) ctx->k2_batch = atoi(g);
    if (const char* g = getenv("HINGE_CALL_LIGHT")) ctx->hinge_light = atoi(g);
    if (const char* g = getenv("HINGE_CALL_MINI")) ctx->hinge_mini = atoi(g);
    if (const char* g = getenv("HINGE_CALL_GROUP")) ctx->group_reads = atoi(g);
    if (const char* g = getenv("HINGE_CALL_LEAN")) ctx->hinge_lean = atoi(g);
    if (const char* g = getenv("HINGE_COUNT_WAVES")) ctx->count_waves = atoi(g) == 4 ? 4 : 2;
    if (const char* g = getenv("HINGE_K2_STEAL")) ctx->k2_steal = atoi(g);
    if (const char* g = getenv("HINGE_SPEC_BAND")) ctx->spec_band = std::max(0, atoi(g));
    if (const char* g = getenv("HINGE_SPEC_SAMPLE")) ctx->spec_ns = std::max(1, atoi(g));
    if (hipMalloc(&ctx->med.p, sizeof(unsigned) * MED_WORDS) != hipSuccess) { (void)hipFree(ctx->scalars.p); delete ctx; return HINGE_E_DEVICE; }
    ctx->med.bytes = sizeof(unsigned) * MED_WORDS;
    (void)hipMemset(ctx->med.p, 0, ctx->med.bytes);
    (void)hipEventCreate(&ctx->ev0);
    (void)hipEventCreate(&ctx->ev1);
    *out = ctx;
    return HINGE_OK;
}

static void cns_release(hinge_ctx* ctx);
static void draft_release(hinge_ctx* ctx);
static void comm_release(hinge_ctx* ctx);
void hinge_ctx_destroy(hinge_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    cns_release(ctx);
    draft_release(ctx);
    comm_release(ctx);
    DevBuf* all[] = {&ctx->rlen, &ctx->qv_mask, &ctx->row_ptr, &ctx->a_span, &ctx->b_span, &ctx->b_flag, &ctx->mask_own, &ctx->mean_own,
                     &ctx->cmask, &ctx->rflags, &ctx->nbins0, &ctx->anno_buf, &ctx->anno_off, &ctx->anno_cnt, &ctx->hinge_flag,
                     &ctx->work_list, &ctx->heavy_list, &ctx->fallback_list, &ctx->bucket_list, &ctx->k2_heads, &ctx->keep, &ctx->span16, &ctx->exact_queue, &ctx->arena, &ctx->scalars, &ctx->med, &ctx->wave_totals, &ctx->trace, &ctx->trace_off, &ctx->tlen,
                     &ctx->eff_reads, &ctx->pair_sel, &ctx->pair_a, &ctx->pair_out, &ctx->cov_buf, &ctx->cov_off_d, &ctx->cov_nb, &ctx->k2c,
                     &ctx->cov_tot, &ctx->redo_list, &ctx->spec_sample, &ctx->final_batch, &ctx->heavy2_list, &ctx->rd_head, &ctx->rd_next2, &ctx->rl2, &ctx->img_row_base, &ctx->img_rec_rel, &ctx->bspan16};
    for (DevBuf* b : all) release(*b);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    for (hipEvent_t e : ctx->prof_pool) (void)hipEventDestroy(e);
    delete ctx;
}

const char* hinge_last_error(const hinge_ctx* ctx) { return ctx ? ctx->err.c_str() : "no context"; }

int hinge_set_stream(hinge_ctx* ctx, void* s) {
    if (!ctx) return HINGE_E_ARG;
    ctx->stream = (hipStream_t)s;
    return HINGE_OK;
}

int hinge_synchronize(hinge_ctx* ctx) {
    if (!ctx) return HINGE_E_ARG;
    CK(hipStreamSynchronize(ctx->stream));
    return HINGE_OK;
}

int hinge_set_reads(hinge_ctx* ctx, int32_t n_reads, const int32_t* rlen, const int32_t* qv_mask) {
    if (!ctx || n_reads <= 0 || !rlen) return fail(ctx, HINGE_E_ARG, "hinge_set_reads: bad arguments");
    CK(hipSetDevice(ctx->device));
    ctx->n_reads = n_reads;
    int rc;
    if ((rc = ensure(ctx, ctx->rlen, sizeof(int) * (size_t)n_reads))) return rc;
    if ((rc = ensure(ctx, ctx->qv_mask, sizeof(int2) * (size_t)n_reads))) return rc;
    CK(hipMemcpyAsync(ctx->rlen.p, rlen, sizeof(int) * (size_t)n_reads, hipMemcpyHostToDevice, ctx->stream));
    ctx->has_qv = qv_mask != nullptr;
    if (qv_mask) CK(hipMemcpyAsync(ctx->qv_mask.p, qv_mask, sizeof(int2) * (size_t)n_reads, hipMemcpyHostToDevice, ctx->stream));
    else CK(hipMemsetAsync(ctx->qv_mask.p, 0, sizeof(int2) * (size_t)n_reads, ctx->stream));
    ctx->max_rlen = 0;
    for (int i = 0; i < n_reads; i++) ctx->max_rlen = std::max(ctx->max_rlen, rlen[i]);
    ctx->h_rlen.assign(rlen, rlen + n_reads);
    // the coverage-output layout (slot offsets, buffer size) was made from the OLD read lengths: forget it
    ctx->cov_key[0] = ctx->cov_key[1] = ctx->cov_key[2] = ctx->cov_key[3] = -1;
    ctx->cov_valid = false;
    size_t n = (size_t)n_reads;
    if ((rc = ensure(ctx, ctx->mask_own, sizeof(int2) * n))) return rc;
    if ((rc = ensure(ctx, ctx->mean_own, sizeof(int) * n))) return rc;
    if ((rc = ensure(ctx, ctx->cmask, sizeof(int2) * n))) return rc;
    if ((rc = ensure(ctx, ctx->rflags, n))) return rc;
    if ((rc = ensure(ctx, ctx->nbins0, sizeof(int) * n))) return rc;
    if ((rc = ensure(ctx, ctx->anno_off, sizeof(unsigned) * n))) return rc;
    if ((rc = ensure(ctx, ctx->anno_cnt, sizeof(int) * n))) return rc;
    if ((rc = ensure(ctx, ctx->rd_head, sizeof(unsigned long long) * n))) return rc;
    CK(hipMemsetAsync(ctx->rd_head.p, 0, sizeof(unsigned long long) * n, ctx->stream));   // stamp 0 = no launch
    ctx->work_cap = (unsigned)std::min<size_t>(n + (size_t)N_SHARD * 256, 0x7fffffffu);
    if ((rc = ensure(ctx, ctx->work_list, sizeof(WorkItem) * (size_t)ctx->work_cap))) return rc;
    if ((rc = ensure(ctx, ctx->fallback_list, sizeof(int) * n))) return rc;
    if ((rc = ensure(ctx, ctx->cov_tot, sizeof(int) * n))) return rc;
    if ((rc = ensure(ctx, ctx->redo_list, sizeof(int) * n))) return rc;
    // (the own tables may just have been reallocated: re-point unless a caller table is attached)
    if (!ctx->mask_attached) ctx->mask = (int2*)ctx->mask_own.p;
    if (!ctx->mean_attached) ctx->mean_cov = (int*)ctx->mean_own.p;
    CK(hipMemsetAsync(ctx->mask_own.p, 0, sizeof(int2) * n, ctx->stream));
    CK(hipMemsetAsync(ctx->anno_cnt.p, 0, sizeof(int) * n, ctx->stream));
    CK(hipMemsetAsync(ctx->anno_off.p, 0, sizeof(unsigned) * n, ctx->stream));
    CK(hipMemsetAsync(ctx->cmask.p, 0, sizeof(int2) * n, ctx->stream));
    CK(hipMemsetAsync(ctx->rflags.p, 0, n, ctx->stream));
    {   // mean_cov defaults to "not in the median"
        std::vector<int> init(n, MEAN_SENTINEL);
        CK(hipMemcpyAsync(ctx->mean_own.p, init.data(), sizeof(int) * n, hipMemcpyHostToDevice, ctx->stream));
        CK(hipStreamSynchronize(ctx->stream));
    }
    return HINGE_OK;
}

static int adopt(hinge_ctx* ctx, DevBuf& b, const void* src, size_t bytes, int on_device) {
    if (on_device) {
        if (b.owned && b.p) (void)hipFree(b.p);
        b.p = const_cast<void*>(src);
        b.bytes = bytes;
        b.owned = false;
        return HINGE_OK;
    }
    if (!b.owned) { b.p = nullptr; b.bytes = 0; b.owned = true; }
    int rc = ensure(ctx, b, bytes);
    if (rc) return rc;
    if (bytes) CK(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return HINGE_OK;
}

static int set_pileups_impl(hinge_ctx* ctx, int32_t r_begin, int32_t r_end, int64_t n_ovl, const int64_t* row_ptr, const int32_t* a_span,
                            const int32_t* b_span, const uint32_t* b_flag, const uint32_t* span16, bool facts_given, uint32_t max_pile,
                            int spans_in_range, int on_device) {
    if (!ctx || ctx->n_reads <= 0) return fail(ctx, HINGE_E_ARG, "hinge_set_pileups: call hinge_set_reads first");
    if (r_begin < 0 || r_end >= ctx->n_reads || r_end < r_begin || n_ovl < 0 || !row_ptr) return fail(ctx, HINGE_E_ARG, "hinge_set_pileups: bad range");
    CK(hipSetDevice(ctx->device));
    ctx->r_begin = r_begin; ctx->r_end = r_end; ctx->n_ovl = n_ovl;
    ctx->image_set = false;   // (a .las image belongs to the pile-ups it was set for)
    ctx->bspan16_state = 0;
    ctx->nbins0_reso = -1;
    ctx->cov_valid = false;
    int rc;
    if ((rc = adopt(ctx, ctx->row_ptr, row_ptr, sizeof(int64_t) * ((size_t)ctx->n_reads + 1), on_device))) return rc;
    if ((rc = adopt(ctx, ctx->a_span, a_span, sizeof(int2) * (size_t)n_ovl, on_device))) return rc;
    if ((rc = adopt(ctx, ctx->b_span, b_span, sizeof(int2) * (size_t)n_ovl, on_device))) return rc;
    if ((rc = adopt(ctx, ctx->b_flag, b_flag, sizeof(unsigned) * (size_t)n_ovl, on_device))) return rc;
    // annotation storage: grows on overflow
    if (ctx->anno_cap == 0) {
        ctx->anno_cap = (unsigned)((std::max<int64_t>(N_SHARD * 32, 4LL * (r_end - r_begin + 1)) + N_SHARD - 1) / N_SHARD * N_SHARD);
        if ((rc = ensure(ctx, ctx->anno_buf, sizeof(int2) * (size_t)ctx->anno_cap))) return rc;
        if ((rc = ensure(ctx, ctx->hinge_flag, (size_t)ctx->anno_cap))) return rc;
        if ((rc = ensure(ctx, ctx->heavy_list, sizeof(HeavyItem) * (size_t)ctx->anno_cap))) return rc;
        if ((rc = ensure(ctx, ctx->heavy2_list, sizeof(HeavyItem) * (size_t)ctx->anno_cap))) return rc;
        if ((rc = ensure(ctx, ctx->rd_next2, sizeof(unsigned) * (size_t)ctx->anno_cap))) return rc;
        if ((rc = ensure(ctx, ctx->rl2, sizeof(unsigned) * (size_t)ctx->anno_cap))) return rc;
    }
    if (ctx->exact_cap == 0) {
        ctx->exact_cap = 4096;
        if ((rc = ensure(ctx, ctx->exact_queue, sizeof(int2) * (size_t)ctx->exact_cap))) return rc;
    }
    if (ctx->arena_cap == 0) {
        ctx->arena_cap = 1ull << 22;   // ints
        if ((rc = ensure(ctx, ctx->arena, sizeof(int) * (size_t)ctx->arena_cap))) return rc;
    }
    {   // K2 length classes: reads that fit one, two or four LDS slots of a workgroup (see k_mask_annotate_q20)
        const int nr = r_end - r_begin + 1;
        const int slot = k2_slot_ints(ctx);
        const int len1 = (slot - 4 * WAVE - 1) * 20 + 19, len2 = (2 * slot - 4 * WAVE - 1) * 20 + 19;   // longest read per class
        ctx->k2_list.assign((size_t)std::max(nr, 1), 0);
        std::vector<int>& lst = ctx->k2_list;   // (lives in the context: the upload is asynchronous)
        int n1 = 0, n2 = 0, n4 = 0;
        for (int i = r_begin; i <= r_end; i++) { const int l = ctx->h_rlen[(size_t)i]; n1 += l <= len1; n2 += l > len1 && l <= len2; }
        n4 = nr - n1 - n2;
        // class 1 longest first (counting sort on rlen / K2_ORDER_BP, reads of one bucket in storage order so that neighbours in the
        // list still share cache lines of the per-read tables): the wavefronts of k_mask_annotate_q20 draw these items one at a time,
        // and a launch whose last items are its cheapest ends with all wavefronts within a short read's time of each other
        const int ob = ctx->k2_order_bp;
        std::vector<int> at((size_t)(std::max(len1, 0) / ob + 2), 0);
        for (int i = r_begin; i <= r_end; i++) { const int l = ctx->h_rlen[(size_t)i]; if (l <= len1) at[(size_t)(std::max(l, 0) / ob)]++; }
        for (int b = (int)at.size() - 1, run = 0; b >= 0; b--) { const int c = at[(size_t)b]; at[(size_t)b] = run; run += c; }
        int p2 = n1, p4 = n1 + n2;
        for (int i = r_begin; i <= r_end; i++) {
            const int l = ctx->h_rlen[(size_t)i];
            if (l <= len1) lst[(size_t)at[(size_t)(std::max(l, 0) / ob)]++] = i; else if (l <= len2) lst[(size_t)p2++] = i; else lst[(size_t)p4++] = i;
        }
        ctx->n_class[0] = n1; ctx->n_class[1] = n2; ctx->n_class[2] = n4;
        ctx->k2_heavy.assign((size_t)std::max(nr, 1), 0);
        if (ctx->k2_heavy_mode && nr > 0) {
            std::vector<int64_t> rp_copy;
            const int64_t* rp = row_ptr;
            if (on_device) {   // (once per part, outside any pass)
                rp_copy.resize((size_t)nr + 1);
                CK(hipMemcpy(rp_copy.data(), row_ptr + r_begin, sizeof(int64_t) * ((size_t)nr + 1), hipMemcpyDeviceToHost));
                rp = rp_copy.data() - r_begin;
            }
            const int64_t deep = 2 * ((rp[r_end + 1] - rp[r_begin]) / nr) + 64;   // deep = more than twice the part's mean
            for (int i = r_begin; i <= r_end; i++) ctx->k2_heavy[(size_t)(i - r_begin)] = rp[i + 1] - rp[i] > deep;
        }
        ctx->k2_deal_heads = ctx->k2_deal_rot = -1;
        ctx->k2_c1.clear();
        if (ctx->k2_deal) {
            ctx->k2_c1.reserve((size_t)n1);
            for (int i = r_begin; i <= r_end; i++) if (ctx->h_rlen[(size_t)i] <= len1) ctx->k2_c1.push_back(i);
        }
        if ((rc = ensure(ctx, ctx->bucket_list, sizeof(int) * (size_t)std::max(nr, 1)))) return rc;
        if (nr > 0) CK(hipMemcpyAsync(ctx->bucket_list.p, lst.data(), sizeof(int) * (size_t)nr, hipMemcpyHostToDevice, ctx->stream));
        const bool pack_ok = ctx->max_rlen < 65536 && n_ovl > 0 && !ctx->no_span16;
        if (facts_given) {
            // The ingest touched every record anyway: it hands over the facts and (when every coordinate fits 16 bits and
            // lies inside its read) the 16|16 copy of the spans, so no device sweep is needed before the first pass.
            ctx->max_pile = max_pile;
            ctx->spans_in_range = spans_in_range != 0;
            ctx->use_span16 = false;
            if (pack_ok && span16 && ctx->spans_in_range) {
                // (+ half a batch of elements: k_mask_annotate_q20 reads its last batch without clamping the index)
                const size_t bytes = sizeof(unsigned) * ((size_t)n_ovl + HINGE_SPAN16_PAD);
                if ((rc = adopt(ctx, ctx->span16, span16, bytes, on_device))) return rc;
                ctx->use_span16 = true;
            }
        } else {
            // one sweep over the spans, once per part: largest pile-up, any coordinate outside [0, rlen], the 16|16 copy
            unsigned* facts = sc(ctx)->facts;
            CK(hipMemsetAsync(facts, 0, 2 * sizeof(unsigned), ctx->stream));
            if (pack_ok) {
                if (!ctx->span16.owned) { ctx->span16.p = nullptr; ctx->span16.bytes = 0; ctx->span16.owned = true; }
                if ((rc = ensure(ctx, ctx->span16, sizeof(unsigned) * ((size_t)n_ovl + HINGE_SPAN16_PAD)))) return rc;
            }
            {
                ProfScope _ps(ctx, KID_PILEUP_FACTS);
                hipLaunchKernelGGL(k_pileup_facts, dim3(std::max(1, std::min((nr + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK, ctx->n_cu * 8))), dim3(BLOCK), 0, ctx->stream, r_begin, r_end,
                                   (const int64_t*)ctx->row_ptr.p, (const int2*)ctx->a_span.p, (const int*)ctx->rlen.p, facts,
                                   pack_ok ? (unsigned*)ctx->span16.p : (unsigned*)nullptr);
            }
            CK(hipGetLastError());
            unsigned h[2] = {0, 0};
            CK(hipMemcpyAsync(h, facts, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
            CK(hipStreamSynchronize(ctx->stream));
            ctx->max_pile = h[0];
            ctx->spans_in_range = h[1] == 0;
            ctx->use_span16 = pack_ok && ctx->spans_in_range;
        }
    }
    if (!on_device) CK(hipStreamSynchronize(ctx->stream));
    return HINGE_OK;
}

int hinge_set_pileups(hinge_ctx* ctx, int32_t r_begin, int32_t r_end, int64_t n_ovl, const int64_t* row_ptr, const int32_t* a_span,
                      const int32_t* b_span, const uint32_t* b_flag, int on_device) {
    return set_pileups_impl(ctx, r_begin, r_end, n_ovl, row_ptr, a_span, b_span, b_flag, nullptr, false, 0, 0, on_device);
}

int hinge_set_pileups_packed(hinge_ctx* ctx, int32_t r_begin, int32_t r_end, int64_t n_ovl, const int64_t* row_ptr, const int32_t* a_span,
                             const int32_t* b_span, const uint32_t* b_flag, const uint32_t* span16, uint32_t max_pile, int spans_in_range,
                             int on_device) {
    return set_pileups_impl(ctx, r_begin, r_end, n_ovl, row_ptr, a_span, b_span, b_flag, span16, true, max_pile, spans_in_range, on_device);
}

int hinge_span16_pad(void) { return HINGE_SPAN16_PAD; }

int hinge_get_pileup_facts(hinge_ctx* ctx, uint32_t* max_pile, int* spans_in_range) {
    if (!ctx || ctx->r_end < ctx->r_begin) return fail(ctx, HINGE_E_ARG, "hinge_get_pileup_facts: no pile-ups set");
    if (max_pile) *max_pile = ctx->max_pile;
    if (spans_in_range) *spans_in_range = ctx->spans_in_range ? 1 : 0;
    return HINGE_OK;
}

int hinge_attach_mask_table(hinge_ctx* ctx, int32_t* d) {
    if (!ctx) return HINGE_E_ARG;
    ctx->mask = d ? (int2*)d : (int2*)ctx->mask_own.p;
    ctx->mask_attached = d != nullptr;
    return HINGE_OK;
}
int hinge_attach_mean_cov(hinge_ctx* ctx, int32_t* d) {
    if (!ctx) return HINGE_E_ARG;
    ctx->mean_cov = d ? d : (int*)ctx->mean_own.p;
    ctx->mean_attached = d != nullptr;
    return HINGE_OK;
}
int hinge_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int hinge_ctx_device_memory(hinge_ctx* ctx, int64_t* free_bytes, int64_t* total_bytes) {
    if (!ctx || !free_bytes || !total_bytes) return HINGE_E_ARG;
    CK(hipSetDevice(ctx->device));
    size_t fr = 0, tot = 0;
    CK(hipMemGetInfo(&fr, &tot));
    *free_bytes = (int64_t)fr; *total_bytes = (int64_t)tot;
    return HINGE_OK;
}

int hinge_set_pile_bins(hinge_ctx* ctx, int32_t reso, const int32_t* nbins, int on_device) {
    if (!ctx || reso <= 0 || !nbins || ctx->r_end < ctx->r_begin || !ctx->nbins0.p) return fail(ctx, HINGE_E_ARG, "hinge_set_pile_bins: bad arguments (call hinge_set_pileups first)");
    CK(hipSetDevice(ctx->device));
    const size_t nr = (size_t)(ctx->r_end - ctx->r_begin + 1);
    CK(hipMemcpyAsync((int*)ctx->nbins0.p + ctx->r_begin, nbins, sizeof(int) * nr, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
    if (!on_device) CK(hipStreamSynchronize(ctx->stream));
    ctx->nbins0_reso = reso;
    return HINGE_OK;
}

int hinge_set_mask_rows(hinge_ctx* ctx, int32_t r0, int32_t r1, const int32_t* rows) {
    if (!ctx || !ctx->mask || !rows || r0 < 0 || r1 >= ctx->n_reads || r1 < r0) return fail(ctx, HINGE_E_ARG, "hinge_set_mask_rows: bad arguments");
    CK(hipSetDevice(ctx->device));
    CK(hipMemcpyAsync(ctx->mask + r0, rows, sizeof(int2) * (size_t)(r1 - r0 + 1), hipMemcpyHostToDevice, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    return HINGE_OK;
}

int hinge_clear_masks(hinge_ctx* ctx) {
    if (!ctx || !ctx->mask) return HINGE_E_ARG;
    CK(hipMemsetAsync(ctx->mask, 0, sizeof(int2) * (size_t)ctx->n_reads, ctx->stream));
    return HINGE_OK;
}

// test hook: std::sort(compare_overlap) order of n keys through the wavefront-parallel replay
__global__ __launch_bounds__(256) void k_debug_pileup_order(const int* __restrict__ keys, int n, int* __restrict__ pos_out) {
    __shared__ WaveSortLds po;
    const int tid = threadIdx.x;
    for (int k = tid; k < n; k += 256) po.key[k] = keys[k];
    __syncthreads();
    block_std_sort_desc(po, n, tid);
    for (int k = tid; k < n; k += 256) pos_out[k] = po.pl[k];
}

int hinge_debug_pileup_order(hinge_ctx* ctx, int32_t n, const int32_t* keys, int32_t* pos_out) {
    if (!ctx || n < 0 || n > PO_CAP || !keys || !pos_out) return fail(ctx, HINGE_E_ARG, "hinge_debug_pileup_order: bad arguments");
    CK(hipSetDevice(ctx->device));
    int *dk = nullptr, *dp = nullptr;
    CK(hipMalloc(&dk, sizeof(int) * (size_t)std::max(n, 1)));
    CK(hipMalloc(&dp, sizeof(int) * (size_t)std::max(n, 1)));
    CK(hipMemcpy(dk, keys, sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_debug_pileup_order, dim3(1), dim3(256), 0, ctx->stream, (const int*)dk, n, dp);
    CK(hipGetLastError());
    CK(hipStreamSynchronize(ctx->stream));
    CK(hipMemcpy(pos_out, dp, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
    (void)hipFree(dk);
    (void)hipFree(dp);
    return HINGE_OK;
}

// hidden knob for tests: run the general K2 kernel even where k_mask_annotate_q20 applies
// (also: environment HINGE_DEBUG_GENERAL_MASK=1 when the context is created, for the executables)
int hinge_debug_force_general_mask(hinge_ctx* ctx, int on) {
    if (!ctx) return HINGE_E_ARG;
    ctx->force_general_mask = on;
    return HINGE_OK;
}

// hidden: reads the last K2 pass handed from k_mask_annotate_q20 back to the general kernel
int hinge_debug_fallback_reads(hinge_ctx* ctx, int64_t* out) {
    if (!ctx || !out) return HINGE_E_ARG;
    unsigned v = 0;
    CK(hipStreamSynchronize(ctx->stream));
    CK(hipMemcpy(&v, &sc(ctx)->fallback_count, sizeof(unsigned), hipMemcpyDeviceToHost));
    *out = v;
    return HINGE_OK;
}

// tests: undecided annotations the last hinge pass sent through the half-size / the full-size instance of k_hinge_call
int hinge_debug_heavy_items(hinge_ctx* ctx, int64_t* out) {
    if (!ctx || !out) return HINGE_E_ARG;
    unsigned v[3] = {0, 0, 0};   // heavy_count, work_next_big, heavy_count_big
    CK(hipStreamSynchronize(ctx->stream));
    CK(hipMemcpy(v, &sc(ctx)->heavy_count, sizeof(v), hipMemcpyDeviceToHost));
    out[0] = v[0];
    out[1] = v[2];
    return HINGE_OK;
}

// hidden knob for tests: 1 = route every scanned annotation through k_hinge_exact,
// 2 = force the in-kernel exact pile-up order for every scanned annotation
int hinge_debug_force_exact(hinge_ctx* ctx, int on) {
    if (!ctx) return HINGE_E_ARG;
    ctx->force_exact = on;
    return HINGE_OK;
}

static int grid_for_reads(hinge_ctx* ctx, int n_reads_in_part, int waves_per_block) {
    int blocks = (n_reads_in_part + waves_per_block - 1) / waves_per_block;
    int cap = ctx->n_cu * 8;
    return std::max(1, std::min(blocks, cap));
}

static int kcap_for(hinge_ctx* ctx, const hinge_filter_params* p) {
    // bins a read can touch: (rlen + cut_off) / reso + 1 (bin_of) + 2 slack, rounded to 4
    int k = (ctx->max_rlen + std::max(p->cut_off, 0)) / p->reso + 4;
    return (k + 3) & ~3;
}

// one memset clears every per-pass device scalar (totals, counters, exact queue, arena, status)
// a pending hinge_filter_set_min_cov becomes a stream-ordered 4-byte set (k_cov_stats applies it for free)
static int flush_min_cov(hinge_ctx* ctx) {
    if (ctx->min_cov_pending) {
        CK(hipMemsetD32Async((hipDeviceptr_t)&sc(ctx)->min_cov, ctx->min_cov_value, 1, ctx->stream));
        ctx->min_cov_pending = false;
    }
    return HINGE_OK;
}

static int launch_stats(hinge_ctx* ctx, const hinge_filter_params* p) {
    const int nr = ctx->r_end - ctx->r_begin + 1;
    const int grid = grid_for_reads(ctx, nr, WAVES_PER_BLOCK);
    ctx->n_wave_totals = grid * WAVES_PER_BLOCK;
    {
        int rc = ensure(ctx, ctx->wave_totals, sizeof(unsigned long long) * 2 * (size_t)ctx->n_wave_totals);
        if (rc) return rc;
    }
    // the kernel also clears the pass scalars and applies a pending MIN_COV (a pass always starts here)
    static_assert(SCALARS_RESET_BYTES % sizeof(int) == 0, "reset region is whole ints");
    const int n_reset = (int)(SCALARS_RESET_BYTES / sizeof(int));
    const int set_mc = ctx->min_cov_pending ? 1 : 0, mc = ctx->min_cov_value;
    ctx->min_cov_pending = false;
    ProfScope _ps(ctx, KID_STATS);
#define LAUNCH_COV_STATS(RESO, PACKED)                                                                                                 \
    hipLaunchKernelGGL((k_cov_stats<RESO, PACKED>), dim3(grid), dim3(BLOCK), 0, ctx->stream, ctx->r_begin, ctx->r_end,                   \
                       (const int64_t*)ctx->row_ptr.p, (const int2*)ctx->a_span.p, (const unsigned*)ctx->span16.p, (const int*)ctx->rlen.p, \
                       p->reso, ctx->mean_cov, (int*)ctx->nbins0.p, (unsigned long long*)ctx->wave_totals.p, (int*)ctx->scalars.p, n_reset, \
                       &sc(ctx)->min_cov, set_mc, mc)
    if (p->reso == 40 && ctx->use_span16) LAUNCH_COV_STATS(40, true);
    else if (p->reso == 40) LAUNCH_COV_STATS(40, false);
    else if (ctx->use_span16) LAUNCH_COV_STATS(0, true);
    else LAUNCH_COV_STATS(0, false);
    CK(hipGetLastError());
    ctx->nbins0_reso = p->reso;   // nbins0[] now describes these pile-ups at this reso
    ctx->pass_mode = 0;           // a two-sweep pass: MIN_COV is exact when K2 starts
    return HINGE_OK;
}

int hinge_filter_stats(hinge_ctx* ctx, const hinge_filter_params* p) {
    int rc = check_params(ctx, p);
    if (rc) return rc;
    if (ctx->r_end < ctx->r_begin) return fail(ctx, HINGE_E_ARG, "no pile-ups set");
    CK(hipSetDevice(ctx->device));
    return launch_stats(ctx, p);   // a pass always starts here: the kernel clears the pass scalars
}

static int fetch_estimate(hinge_ctx* ctx, hinge_cov_estimate* out) {
    Scalars h;
    CK(hipMemcpyAsync(&h, ctx->scalars.p, sizeof(Scalars), hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    out->cov_est = h.est[0];
    out->n_long = h.est[1];
    out->total_cov = (int64_t)h.totals[0];
    out->num_slot = (int64_t)h.totals[1];
    if (h.status & ST_NO_LONG_READ) return fail(ctx, HINGE_E_UNDEFINED, "no read >= 5000 bp in this part: the reference is undefined here (filter.cpp:660-666)");
    return HINGE_OK;
}

int hinge_filter_stats_median(hinge_ctx* ctx, const hinge_filter_params* p, uint32_t* hist_dev, hinge_cov_estimate* out) {
    int rc = check_params(ctx, p);
    if (rc) return rc;
    if (ctx->r_end < ctx->r_begin) return fail(ctx, HINGE_E_ARG, "no pile-ups set");
    CK(hipSetDevice(ctx->device));
    if ((rc = launch_stats(ctx, p))) return rc;
    return hist_dev ? hinge_filter_median_hist(ctx, p, ctx->r_begin, ctx->r_end, hist_dev) : hinge_filter_median(ctx, p, ctx->r_begin, ctx->r_end, out);
}

static SpecVerify spec_verify_of(hinge_ctx* ctx) {
    SpecVerify v;
    memset(&v, 0, sizeof(v));
    if (ctx->pass_mode != 0) {
        v.spec_min_cov = &sc(ctx)->spec_min_cov; v.band = ctx->spec_band; v.spec_state = &sc(ctx)->spec_state;
        v.shards = sc(ctx)->shards; v.stats = sc(ctx)->spec_stats;
    }
    return v;
}
static MedianPart median_part_of(hinge_ctx* ctx, int32_t lo, int32_t hi, uint32_t* hist_dev) {
    MedianPart a;
    memset(&a, 0, sizeof(a));
    a.mean_cov = (const int*)ctx->mean_cov; a.lo = lo; a.hi = hi;
    a.med = (unsigned*)ctx->med.p; a.est = sc(ctx)->est; a.min_cov = &sc(ctx)->min_cov; a.status = &sc(ctx)->status;
    a.wave_totals = (const unsigned long long*)ctx->wave_totals.p; a.n_wave_totals = ctx->n_wave_totals; a.totals = sc(ctx)->totals;
    a.hist_out = (unsigned*)hist_dev;
    if (ctx->pass_mode == 1) {   // one-sweep pass through k_mask_annotate_q20<SPEC>: the means are derived here, from the sweep's sums
        a.cov_tot = (const int*)ctx->cov_tot.p; a.nbins0 = (const int*)ctx->nbins0.p; a.rlen = (const int*)ctx->rlen.p; a.mean_out = ctx->mean_cov;
    }
    a.spec = spec_verify_of(ctx);
    return a;
}
// k_median_hist over n parts (contexts on one device and one stream), each over its own [lo, hi]; hist_dev[k] as in median_hist
static int launch_median_batch(hinge_ctx** ctxs, int n, const hinge_filter_params* p, const int32_t* lo, const int32_t* hi, uint32_t* const* hist_dev) {
    hinge_ctx* ctx = ctxs[0];
    MedianHistBatch B;
    memset(&B, 0, sizeof(B));
    B.n = n;
    int blocks = 1;
    for (int k = 0; k < n; k++) {
        int rc = flush_min_cov(ctxs[k]);
        if (rc) return rc;
        if (ctxs[k]->pass_mode == 1 && (lo[k] < ctxs[k]->r_begin || hi[k] > ctxs[k]->r_end))
            return fail(ctxs[k], HINGE_E_ARG, "median of a one-sweep pass: the range must lie inside the part's own reads (the means of other reads do not exist)");
        B.part[k] = median_part_of(ctxs[k], lo[k], hi[k], hist_dev ? hist_dev[k] : nullptr);
        if (hist_dev) memset(&B.part[k].spec, 0, sizeof(SpecVerify));   // the histogram form: verified where the median is finished (k_median_from_hist)
        blocks = std::max(blocks, std::min((hi[k] - lo[k] + 1 + 1023) / 1024, MED_MAX_BLOCKS));
    }
    ProfScope _ps(ctx, KID_MEDIAN);
    hipLaunchKernelGGL(k_median_hist, dim3(blocks * n), dim3(256), 0, ctx->stream, B, p->est_cov);
    CK(hipGetLastError());
    return HINGE_OK;
}

int hinge_filter_median(hinge_ctx* ctx, const hinge_filter_params* p, int32_t lo, int32_t hi, hinge_cov_estimate* out) {
    int rc = check_params(ctx, p);
    if (rc) return rc;
    if (lo < 0 || hi >= ctx->n_reads || hi < lo) return fail(ctx, HINGE_E_ARG, "median range");
    CK(hipSetDevice(ctx->device));
    if ((rc = launch_median_batch(&ctx, 1, p, &lo, &hi, nullptr))) return rc;
    if (out) return fetch_estimate(ctx, out);
    return HINGE_OK;
}

// ---- sharded median: local histogram -> (all-reduce by the caller) -> median -------------------------------------
int hinge_filter_median_hist(hinge_ctx* ctx, const hinge_filter_params* p, int32_t lo, int32_t hi, uint32_t* hist_dev) {
    int rc = check_params(ctx, p);
    if (rc) return rc;
    if (lo < 0 || hi >= ctx->n_reads || hi < lo || !hist_dev) return fail(ctx, HINGE_E_ARG, "median_hist: bad arguments");
    CK(hipSetDevice(ctx->device));
    return launch_median_batch(&ctx, 1, p, &lo, &hi, &hist_dev);
}

int hinge_filter_median_batch(hinge_ctx** ctxs, int32_t n, const hinge_filter_params* p, uint32_t* hist_dev, int64_t row_stride) {
    if (!ctxs || n <= 0 || n > MED_BATCH_MAX || (hist_dev && row_stride < MED_BINS + 2)) return HINGE_E_ARG;
    for (int k = 0; k < n; k++)
        if (!ctxs[k] || ctxs[k]->device != ctxs[0]->device || ctxs[k]->stream != ctxs[0]->stream || ctxs[k]->r_end < ctxs[k]->r_begin)
            return fail(ctxs[0], HINGE_E_ARG, "median_batch: the contexts must share one device and one stream and have pile-ups set");
    hinge_ctx* ctx = ctxs[0];
    int rc = check_params(ctx, p);
    if (rc) return rc;
    CK(hipSetDevice(ctx->device));
    int32_t lo[MED_BATCH_MAX], hi[MED_BATCH_MAX];
    uint32_t* hd[MED_BATCH_MAX];
    for (int k = 0; k < n; k++) { lo[k] = ctxs[k]->r_begin; hi[k] = ctxs[k]->r_end; hd[k] = hist_dev ? hist_dev + (int64_t)k * row_stride : nullptr; }
    return launch_median_batch(ctxs, n, p, lo, hi, hist_dev ? hd : nullptr);
}

int hinge_filter_median_from_hist(hinge_ctx* ctx, const hinge_filter_params* p, const uint32_t* hist_dev) {
    int rc = check_params(ctx, p);
    if (rc) return rc;
    if (!hist_dev) return fail(ctx, HINGE_E_ARG, "median_from_hist: bad arguments");
    CK(hipSetDevice(ctx->device));
    if ((rc = flush_min_cov(ctx))) return rc;
    ProfScope _ps(ctx, KID_MEDIAN);
    hipLaunchKernelGGL(k_median_from_hist, dim3(1), dim3(256), 0, ctx->stream, (const unsigned*)hist_dev, p->est_cov, sc(ctx)->est,
                       &sc(ctx)->min_cov, &sc(ctx)->status, spec_verify_of(ctx));
    CK(hipGetLastError());
    return HINGE_OK;
}

int hinge_filter_median_from_hist_batch(hinge_ctx** ctxs, int32_t n, const hinge_filter_params* p, const uint32_t* hist_dev, int64_t row_stride) {
    if (!ctxs || n <= 0 || n > MED_BATCH_MAX || !hist_dev || row_stride < MED_BINS + 2) return HINGE_E_ARG;
    for (int k = 0; k < n; k++)
        if (!ctxs[k] || ctxs[k]->device != ctxs[0]->device || ctxs[k]->stream != ctxs[0]->stream)
            return fail(ctxs[0], HINGE_E_ARG, "median_from_hist_batch: the contexts must share one device and one stream");
    hinge_ctx* ctx = ctxs[0];
    int rc = check_params(ctx, p);
    if (rc) return rc;
    CK(hipSetDevice(ctx->device));
    MedianBatch B;
    memset(&B, 0, sizeof(B));
    for (int k = 0; k < n; k++) {
        if ((rc = flush_min_cov(ctxs[k]))) return rc;
        B.est[k] = sc(ctxs[k])->est;
        B.min_cov[k] = &sc(ctxs[k])->min_cov;
        B.status[k] = &sc(ctxs[k])->status;
        B.spec[k] = spec_verify_of(ctxs[k]);
    }
    ProfScope _ps(ctx, KID_MEDIAN);
    hipLaunchKernelGGL(k_median_from_hist_batch, dim3(n), dim3(256), 0, ctx->stream, (const unsigned*)hist_dev, (long long)row_stride, p->est_cov, B);
    CK(hipGetLastError());
    return HINGE_OK;
}

int hinge_set_read_restriction(hinge_ctx* ctx, const uint8_t* keep) {
    if (!ctx || ctx->n_reads <= 0) return fail(ctx, HINGE_E_ARG, "hinge_set_read_restriction: call hinge_set_reads first");
    CK(hipSetDevice(ctx->device));
    ctx->has_keep = keep != nullptr;
    if (!keep) return HINGE_OK;
    int rc = ensure(ctx, ctx->keep, (size_t)ctx->n_reads);
    if (rc) return rc;
    CK(hipMemcpyAsync(ctx->keep.p, keep, (size_t)ctx->n_reads, hipMemcpyHostToDevice, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    return HINGE_OK;
}

int hinge_filter_set_min_cov(hinge_ctx* ctx, int32_t v) {
    if (!ctx) return HINGE_E_ARG;
    ctx->min_cov_pending = true;   // applied, in stream order, by the next launch that reads or updates MIN_COV
    ctx->min_cov_value = v;
    return HINGE_OK;
}
int hinge_filter_get_min_cov(hinge_ctx* ctx, int32_t* v) {
    if (!ctx || !v) return HINGE_E_ARG;
    int rc0 = flush_min_cov(ctx);
    if (rc0) return rc0;
    CK(hipMemcpyAsync(v, &sc(ctx)->min_cov, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    return HINGE_OK;
}

static AnnoOut anno_out(hinge_ctx* ctx) {
    AnnoOut o;
    o.qv_mask = ctx->has_qv ? (const int2*)ctx->qv_mask.p : (const int2*)nullptr;
    o.keep = ctx->has_keep ? (const unsigned char*)ctx->keep.p : (const unsigned char*)nullptr;
    o.mask = ctx->mask;
    o.cmask = (int2*)ctx->cmask.p;
    o.rflags = (unsigned char*)ctx->rflags.p;
    o.anno_buf = (int2*)ctx->anno_buf.p;
    o.hinge_flag = (unsigned char*)ctx->hinge_flag.p;
    o.anno_off = (unsigned*)ctx->anno_off.p;
    o.anno_cnt = (int*)ctx->anno_cnt.p;
    o.anno_shard = sc(ctx)->shards; o.work_shard = sc(ctx)->shards + N_SHARD * SHARD_STRIDE;
    o.anno_region = ctx->anno_cap / N_SHARD; o.work_cap = ctx->work_cap;
    o.anno_cap = ctx->anno_cap;
    o.work_list = (WorkItem*)ctx->work_list.p;
    o.status = &sc(ctx)->status;
    o.cov_out = ctx->cov_out_on ? (int*)ctx->cov_buf.p : (int*)nullptr;
    o.cov_off = (const long long*)ctx->cov_off_d.p;
    o.cov_nbins = (int*)ctx->cov_nb.p;
    o.cov_base = ctx->r_begin;
    return o;
}

#define LAUNCH_MASK_ANNOTATE(RESO, GRID, LIST, COUNT, SA)                                                                     \
    hipLaunchKernelGGL(k_mask_annotate<RESO>, dim3(GRID), dim3(BLOCK), lds, ctx->stream, to_dev(p), ctx->r_begin, ctx->r_end,  \
                       (const int64_t*)ctx->row_ptr.p, (const int2*)ctx->a_span.p, (const int*)ctx->rlen.p,                     \
                       (const int*)((SA).mode == MODE_SPEC ? &sc(ctx)->spec_min_cov : &sc(ctx)->min_cov), kcap, anno_out(ctx), LIST, COUNT, SA)

// arguments of the general kernel's one-sweep modes (MODE_CLASSIC: all unused); MODE_SPEC launches write one (sum, slots) pair per wavefront
static int spec_args_of(hinge_ctx* ctx, int mode, int grid, SpecArgs* out) {
    SpecArgs a;
    memset(&a, 0, sizeof(a));
    a.mode = mode; a.band = ctx->spec_band;
    a.mean_cov = ctx->mean_cov;
    a.spec_state = &sc(ctx)->spec_state;
    a.redo_list = (int*)ctx->redo_list.p; a.redo_count = &sc(ctx)->redo_count; a.redo_cap = (unsigned)ctx->n_reads;
    if (mode == MODE_SPEC) {
        ctx->n_wave_totals = grid * WAVES_PER_BLOCK;
        int rc = ensure(ctx, ctx->wave_totals, sizeof(unsigned long long) * 2 * (size_t)ctx->n_wave_totals);
        if (rc) return rc;
    }
    a.wave_totals = (unsigned long long*)ctx->wave_totals.p;
    if (mode == MODE_SPEC && ctx->pass_mode == 1) { a.cov_tot = (int*)ctx->cov_tot.p; a.nbins0 = (const int*)ctx->nbins0.p; }
    *out = a;
    return HINGE_OK;
}

// Layout of K2's coverage-bin output: read i of the part gets (rlen + cut_off) / reso + 3 slots, the most bins a profile the
// kernels accept can have (more raises ST_RANGE).  Host-computable, so no device prefix sum and no second launch.
// ST_ANNO_CAP: a shard ran out of annotation slots or the interleaved work list of room - both buffers double, the pass is repeated
static int grow_annotations(hinge_ctx* ctx) {
    int rc;
    ctx->anno_cap = ctx->anno_cap * 2;       // (stays a multiple of N_SHARD)
    if ((rc = ensure(ctx, ctx->anno_buf, sizeof(int2) * (size_t)ctx->anno_cap))) return rc;
    if ((rc = ensure(ctx, ctx->hinge_flag, (size_t)ctx->anno_cap))) return rc;
    if ((rc = ensure(ctx, ctx->heavy_list, sizeof(HeavyItem) * (size_t)ctx->anno_cap))) return rc;
    if ((rc = ensure(ctx, ctx->heavy2_list, sizeof(HeavyItem) * (size_t)ctx->anno_cap))) return rc;
    if ((rc = ensure(ctx, ctx->rd_next2, sizeof(unsigned) * (size_t)ctx->anno_cap))) return rc;
    if ((rc = ensure(ctx, ctx->rl2, sizeof(unsigned) * (size_t)ctx->anno_cap))) return rc;
    ctx->work_cap = ctx->work_cap * 2;
    return ensure(ctx, ctx->work_list, sizeof(WorkItem) * (size_t)ctx->work_cap);
}

static int prepare_cov_out(hinge_ctx* ctx, const hinge_filter_params* p) {
    if (!ctx->cov_out_on) return HINGE_OK;
    const int key[4] = {ctx->r_begin, ctx->r_end, p->reso, p->cut_off};
    const int nr = ctx->r_end - ctx->r_begin + 1;
    if (memcmp(key, ctx->cov_key, sizeof(key)) != 0 || ctx->h_cov_off.size() != (size_t)nr + 1) {
        ctx->h_cov_off.assign((size_t)nr + 1, 0);
        for (int k = 0; k < nr; k++)
            ctx->h_cov_off[(size_t)k + 1] = ctx->h_cov_off[(size_t)k] + ((int64_t)std::max(ctx->h_rlen[(size_t)(ctx->r_begin + k)], 0) + std::max(p->cut_off, 0)) / p->reso + 6;   // (the bins, + room for what k_mask_annotate_q20's 16-byte stores write past the last one)
        int rc;
        if ((rc = ensure(ctx, ctx->cov_off_d, sizeof(int64_t) * ((size_t)nr + 1)))) return rc;
        if ((rc = ensure(ctx, ctx->cov_nb, sizeof(int) * (size_t)nr))) return rc;
        if ((rc = ensure(ctx, ctx->cov_buf, sizeof(int) * (size_t)std::max<int64_t>(ctx->h_cov_off[(size_t)nr], 1)))) return rc;
        CK(hipMemcpyAsync(ctx->cov_off_d.p, ctx->h_cov_off.data(), sizeof(int64_t) * ((size_t)nr + 1), hipMemcpyHostToDevice, ctx->stream));
        CK(hipStreamSynchronize(ctx->stream));
        memcpy(ctx->cov_key, key, sizeof(key));
    }
    ctx->cov_valid = true;
    return HINGE_OK;
}

// ---- K2 (k_mask_annotate_q20): what a launch needs, worked out per part; then one launch per part or one for all of them ----
struct K2Prep {
    int n1, n2, n4, g, n_heads, slot, mulpath_thr, cov_mask_off, variant;
    size_t lds;
    int* cov_out;
    unsigned base[K2_MAX_HEADS], next_base[K2_MAX_HEADS];   // the host mirror of the device counters moves on only once the launch is known to be queued
};
static bool k2_applies(hinge_ctx* ctx, const hinge_filter_params* p, int* mulpath_thr_out) {
    // shipped configuration (reso 40, cut_off = 300): one 20-bp begin|end histogram per read
    // (it takes the bin count and the well-formedness of each pile-up from k_cov_stats<40> of this pass)
    // ... and the division-free annotation test of mask_gate_annotate applies: then |gradient| > min(MIN_RA, MAX_RA) is necessary for an
    // annotation, which is what the fast kernel's scan flags its 64-bin words with (the threshold is added to a count there: < 2^28)
    const int mulpath_thr = (p->coverage_fraction > 0 && p->coverage_fraction < 8192 && p->min_repeat_annotation >= 0 && p->max_repeat_annotation >= 0)
                                ? std::min(p->min_repeat_annotation, p->max_repeat_annotation) : -1;
    *mulpath_thr_out = mulpath_thr;
    return p->reso == 40 && p->cut_off >= 0 && p->cut_off % 20 == 0 && p->cut_off <= 1200 && ctx->force_general_mask == 0 && ctx->nbins0_reso == 40 &&
           mulpath_thr >= 0 && mulpath_thr < (1 << 28);
}
// `share`: how many parts divide the chip's resident workgroups between them (1: a launch of its own)
static int k2_prepare(hinge_ctx* ctx, const hinge_filter_params* p, int mode, int share, K2Prep* q) {
    {
    // bins + hot words (the read classes of hinge_set_pileups are cut for this much) + the zero / total pads of this cut_off
    const int SH = p->cut_off / 20;
    const int slot = k2_slot_ints(ctx) + ((SH + 2 + 3) & ~3) + ((2 * SH + 4 + 3) & ~3);
    const size_t lds20 = (size_t)WAVES_PER_BLOCK * slot * sizeof(int);   // 19.8 KiB for reads of up to 19 kb: eight workgroups (32 wavefronts) per CU
    const size_t lds_all = lds20;
    if (ctx->k2_occ_lds != (int)lds_all) {
        int nb = 0;
        if (ctx->use_span16) CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_mask_annotate_q20<true, true, 15, 0>, BLOCK, lds_all));
        else CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_mask_annotate_q20<false, true, 15, 0>, BLOCK, lds_all));
        ctx->k2_occ = std::max(nb, 1);
        ctx->k2_occ_lds = (int)lds_all;
    }
    const int n1 = ctx->n_class[0], n2 = ctx->n_class[1], n4 = ctx->n_class[2];
    // persistent workgroups for the short reads: as many as are resident at once (HINGE_K2_WGS overrides), in a multiple of the
    // number of item counters so that every counter serves the same number of workgroups
    int gp = n1 > 0 ? std::min((n1 + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK, std::max((ctx->k2_wgs > 0 ? ctx->k2_wgs : ctx->k2_occ * ctx->n_cu) / share, 1)) : 0;
    int n_heads = 1;
    if (gp >= 2 * K2_MAX_HEADS) { n_heads = K2_MAX_HEADS; gp -= gp % K2_MAX_HEADS; }
    else for (int h = std::min(gp, K2_MAX_HEADS); h >= 1; h--) if (gp % h == 0) { n_heads = h; break; }
    const int g = std::max(1, n4 + (n2 + 1) / 2 + gp);
    // XCD-contiguous deal (HINGE_K2_DEAL=1): workgroups go round-robin to the 8 XCDs, each with its own L2; head h is served by
    // the workgroups g4 + g2 + h, + n_heads, ... - all on XCD (g4 + g2 + h) % 8 when n_heads is a multiple of 8 - and takes the
    // list positions h, h + n_heads, ...  So the LIST is arranged such that the positions of one XCD's heads hold one contiguous
    // eighth of the reads in storage order: that L2 then sees one eighth of the span copy and of the per-read tables, and the
    // rows it fetches next to each other in memory are worked on next to each other in time.
    if (ctx->k2_deal && n1 > 0 && n_heads % 8 == 0 && (ctx->k2_deal_heads != n_heads || ctx->k2_deal_rot != (n4 + (n2 + 1) / 2) % 8)) {
        const int rot = (n4 + (n2 + 1) / 2) % 8;
        int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, at[8];
        for (int pz = 0; pz < n1; pz++) cnt[((pz % n_heads) + rot) % 8]++;
        for (int x = 0, run = 0; x < 8; x++) { at[x] = run; run += cnt[x]; }
        std::vector<int> c1(ctx->k2_c1);
        if (ctx->k2_heavy_mode) {
            // Where an XCD's DEEP pile-ups go in its sequence.  A read's time is ~1.3 us + 12 ns per overlap (per-read time stamps,
            // tools/k2_trace.py: 5 us on average, 20-30 us for the 2 000 overlaps of a read inside a repeat), and one that is drawn in
            // the last third of the launch ends long after everything else: the launch's last 7 us ran at falling occupancy behind
            // a handful of them.  All of them FIRST is far worse (66 -> 90 us): hundreds of their overlaps begin or end in the same
            // 20-bp bin, a same-address LDS atomic costs 0.83 ns of the CU's LDS pipe per lane that shares the word
            // (tools/probes/lds_atomic_probe.cpp), and a CU full of them stalls on it; already over the first 50 % of the
            // sequence they are too dense (69 us).  At even intervals over the first 60 %: 67.2 -> 64.4 us.
            auto heavy = [&](int i) { return ctx->k2_heavy[(size_t)(i - ctx->r_begin)] != 0; };
            for (int x = 0; x < 8; x++) {
                std::vector<int> seg(c1.begin() + at[x], c1.begin() + at[x] + cnt[x]), hv, rest;
                for (int i : seg) (heavy(i) ? hv : rest).push_back(i);
                size_t o = (size_t)at[x], ih = 0, ir = 0;
                const size_t span = ctx->k2_heavy_mode == 2 ? (size_t)(0.6 * seg.size()) : hv.size();
                for (size_t k = 0; k < seg.size(); k++) {
                    const bool take_h = ih < hv.size() && (ir >= rest.size() || (k < span && ih * span <= k * hv.size()));
                    c1[o + k] = take_h ? hv[ih++] : rest[ir++];
                }
            }
        }
        for (int pz = 0; pz < n1; pz++) ctx->k2_list[(size_t)pz] = c1[(size_t)at[((pz % n_heads) + rot) % 8]++];
        CK(hipMemcpyAsync(ctx->bucket_list.p, ctx->k2_list.data(), sizeof(int) * (size_t)n1, hipMemcpyHostToDevice, ctx->stream));
        ctx->k2_deal_heads = n_heads; ctx->k2_deal_rot = rot;
    }
    {
        if (!ctx->k2_heads.p) {
            int rc = ensure(ctx, ctx->k2_heads, sizeof(unsigned) * 32 * K2_MAX_HEADS);
            if (rc) return rc;
            CK(hipMemsetAsync(ctx->k2_heads.p, 0, sizeof(unsigned) * 32 * K2_MAX_HEADS, ctx->stream));
        }
        const int gp_run = g - n4 - (n2 + 1) / 2;   // (g >= 1: an empty part still launches one workgroup)
        for (int h = 0; h < K2_MAX_HEADS; h++) {
            q->base[h] = q->next_base[h] = ctx->k2_head_base[h];
            if (h >= n_heads) continue;
            // one draw per wavefront of the head's workgroups + one per item of the head
            const unsigned wgs_h = (unsigned)(gp_run / n_heads + (h < gp_run % n_heads));
            const unsigned items_h = (unsigned)(n1 / n_heads + (h < n1 % n_heads));
            q->next_base[h] += wgs_h * WAVES_PER_BLOCK + items_h;
        }
    }
    {   // the kernel's constants (parameters, output pointers): a 200-byte block in device memory, uploaded when it changes
        K2Const hc;
        memset(&hc, 0, sizeof(hc));
        hc.P = to_dev(p);
        hc.o = anno_out(ctx);
        hc.fallback_list = (int*)ctx->fallback_list.p;
        hc.fallback_count = &sc(ctx)->fallback_count;
        hc.redo_list = (int*)ctx->redo_list.p;
        hc.redo_count = &sc(ctx)->redo_count;
        hc.redo_cap = (unsigned)ctx->n_reads;
        int rc = ensure(ctx, ctx->k2c, sizeof(K2Const));
        if (rc) return rc;
        if (!ctx->k2c_valid || memcmp(&hc, &ctx->k2c_host, sizeof(K2Const)) != 0) {
            ctx->k2c_host = hc;
            CK(hipMemcpyAsync(ctx->k2c.p, &ctx->k2c_host, sizeof(K2Const), hipMemcpyHostToDevice, ctx->stream));
            ctx->k2c_valid = true;
        }
    }
    q->n1 = n1; q->n2 = n2; q->n4 = n4; q->g = g; q->n_heads = n_heads; q->slot = slot; q->lds = lds_all;
    q->cov_out = ctx->cov_out_on ? (int*)ctx->cov_buf.p : (int*)nullptr;
    q->cov_mask_off = p->use_coverage_mask != 0 ? INT_MIN : (1 << 29);
    const int spec = mode == MODE_SPEC ? (ctx->spec_band == 1 ? 1 : 2) : 0;
    q->variant = (ctx->use_span16 ? 1 : 0) | (q->cov_out ? 2 : 0) | (p->cut_off == 300 ? 4 : 0) | (spec << 3);
    }
    return HINGE_OK;
}
// the template instance of a variant word: PACKED | COVOUT << 1 | (cut_off == 300) << 2 | SPEC << 3
#define K2_DISPATCH(variant, CALL)                                                                                                            \
    do {                                                                                                                                      \
        switch (variant) {                                                                                                                    \
            K2_CASES(0, CALL) K2_CASES(1, CALL) K2_CASES(2, CALL)                                                                             \
            default: return fail(ctx, HINGE_E_ARG, "k_mask_annotate_q20: unknown variant");                                              \
        }                                                                                                                                     \
    } while (0)
#define K2_CASES(SPEC, CALL)                                                                                                                  \
    case ((SPEC) << 3) | 0: CALL(false, false, -1, SPEC); break; case ((SPEC) << 3) | 1: CALL(true, false, -1, SPEC); break;                  \
    case ((SPEC) << 3) | 2: CALL(false, true, -1, SPEC); break;  case ((SPEC) << 3) | 3: CALL(true, true, -1, SPEC); break;                   \
    case ((SPEC) << 3) | 4: CALL(false, false, 15, SPEC); break; case ((SPEC) << 3) | 5: CALL(true, false, 15, SPEC); break;                  \
    case ((SPEC) << 3) | 6: CALL(false, true, 15, SPEC); break;  case ((SPEC) << 3) | 7: CALL(true, true, 15, SPEC); break;
static int k2_launch_one(hinge_ctx* ctx, const hinge_filter_params* p, const K2Prep& q) {
    K2Heads bases;
    memcpy(bases.base, q.base, sizeof(bases.base));
    const void* spans = ctx->use_span16 ? (const void*)ctx->span16.p : (const void*)ctx->a_span.p;
    const int* min_cov = (q.variant >> 3) != 0 ? &sc(ctx)->spec_min_cov : &sc(ctx)->min_cov;
    ProfScope _ps(ctx, KID_MASK_ANNOTATE);
#define K2_ONE(PACKED, COVOUT, CUT20, SPEC)                                                                                                                   \
    hipLaunchKernelGGL((k_mask_annotate_q20<PACKED, COVOUT, CUT20, SPEC>), dim3(q.g), dim3(BLOCK), q.lds, ctx->stream, (const K2Const*)ctx->k2c.p, p->cut_off, \
                       q.mulpath_thr, p->no_hinge_region, q.cov_mask_off, (const int*)ctx->bucket_list.p, q.n1, q.n2, q.n4, (const int64_t*)ctx->row_ptr.p,   \
                       (const typename SpanLoad<PACKED>::raw*)spans, (const int*)ctx->rlen.p, (const int*)ctx->nbins0.p, min_cov, q.slot, q.cov_out,           \
                       (const long long*)ctx->cov_off_d.p, (int*)ctx->cov_nb.p, ctx->r_begin, (unsigned*)ctx->k2_heads.p, q.n_heads, bases,                    \
                       (int*)ctx->cov_tot.p, ctx->spec_band)
    K2_DISPATCH(q.variant, K2_ONE);
#undef K2_ONE
    CK(hipGetLastError());
    memcpy(ctx->k2_head_base, q.next_base, sizeof(ctx->k2_head_base));
    return HINGE_OK;
}
// reads handed back by K2 (65536+ overlaps, coordinates outside [0, rlen], longer than four LDS slots) go through the general
// kernel: the launch is skipped when the part's facts rule all three out
static int k2_fallback(hinge_ctx* ctx, const hinge_filter_params* p, int mode, int grid) {
    const bool no_handback = ctx->max_pile < 65536u && ctx->spans_in_range && ctx->max_rlen / 20 < WAVES_PER_BLOCK * k2_slot_ints(ctx) - 4 * WAVE;   // (bins-only slots: conservative)
    if (no_handback) return HINGE_OK;
    SpecArgs sa;
    int rc = spec_args_of(ctx, mode, std::min(grid, 64), &sa);
    if (rc) return rc;
    const int kcap = kcap_for(ctx, p);
    const size_t lds = (size_t)WAVES_PER_BLOCK * 2 * kcap * sizeof(int);
    if (lds > 160 * 1024) return fail(ctx, HINGE_E_RANGE, "read too long for the LDS histogram (max ~200 kb)");
    if (lds > 48 * 1024 && lds > ctx->lds_attr_set) {
        CK(hipFuncSetAttribute((const void*)k_mask_annotate<40>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CK(hipFuncSetAttribute((const void*)k_mask_annotate<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        ctx->lds_attr_set = lds;
    }
    ProfScope _ps2(ctx, KID_MASK_FALLBACK);
    LAUNCH_MASK_ANNOTATE(40, std::min(grid, 64), (const int*)ctx->fallback_list.p, (const unsigned*)&sc(ctx)->fallback_count, sa);
    CK(hipGetLastError());
    return HINGE_OK;
}

static int launch_mask_annotate(hinge_ctx* ctx, const hinge_filter_params* p, int mode = MODE_CLASSIC) {
    {
        int rc = flush_min_cov(ctx);
        if (rc) return rc;
        if ((rc = prepare_cov_out(ctx, p))) return rc;
    }
    if (mode == MODE_SPEC) ctx->n_wave_totals = 0;   // (set again below if the general kernel takes part in the sweep)
    const int kcap = kcap_for(ctx, p);
    const size_t lds = (size_t)WAVES_PER_BLOCK * 2 * kcap * sizeof(int);
    if (lds > 160 * 1024) return fail(ctx, HINGE_E_RANGE, "read too long for the LDS histogram (max ~200 kb)");
    if (lds > 48 * 1024 && lds > ctx->lds_attr_set) {
        CK(hipFuncSetAttribute((const void*)k_mask_annotate<40>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CK(hipFuncSetAttribute((const void*)k_mask_annotate<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        ctx->lds_attr_set = lds;
    }
    const int nr = ctx->r_end - ctx->r_begin + 1;
    // the general kernel: one read per wavefront, no grid cap - the hardware dispatcher balances uneven pile-ups better than a
    // grid-stride loop does (measured 137 -> 119 us at 87 k reads; the fast kernel below draws its reads instead)
    const int grid = std::max(1, std::min((nr + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK, 1 << 20));
    int mulpath_thr;
    const bool q20 = k2_applies(ctx, p, &mulpath_thr);
    if (mode == MODE_FINAL) {
        // the guard-band list of a one-sweep pass (about 1 % of the part's reads; every read if the prediction missed the band),
        // one read per wavefront through the general kernel
        SpecArgs sa;
        const int gfin = std::max(64, std::min(grid, std::max(nr / 128, 1)));
        int rc = spec_args_of(ctx, MODE_FINAL, gfin, &sa);
        if (rc) return rc;
        ProfScope _ps(ctx, KID_MASK_FINAL);
        if (p->reso == 40) LAUNCH_MASK_ANNOTATE(40, gfin, (const int*)ctx->redo_list.p, (const unsigned*)&sc(ctx)->redo_count, sa);
        else LAUNCH_MASK_ANNOTATE(0, gfin, (const int*)ctx->redo_list.p, (const unsigned*)&sc(ctx)->redo_count, sa);
        CK(hipGetLastError());
        return HINGE_OK;
    }
    if (mode == MODE_SPEC) ctx->pass_mode = q20 ? 1 : 2;
    if (q20) {
        K2Prep q;
        int rc = k2_prepare(ctx, p, mode, 1, &q);
        if (rc) return rc;
        q.mulpath_thr = mulpath_thr;
        if ((rc = k2_launch_one(ctx, p, q))) return rc;
        return k2_fallback(ctx, p, mode, grid);
    }
    SpecArgs sa;
    {
        int rc = spec_args_of(ctx, mode, grid, &sa);
        if (rc) return rc;
    }
    ProfScope _ps(ctx, KID_MASK_ANNOTATE);
    if (p->reso == 40) LAUNCH_MASK_ANNOTATE(40, grid, (const int*)nullptr, (const unsigned*)nullptr, sa);
    else LAUNCH_MASK_ANNOTATE(0, grid, (const int*)nullptr, (const unsigned*)nullptr, sa);
    CK(hipGetLastError());
    return HINGE_OK;
}

// The mask / annotation sweep over n resident parts (contexts on one device and one stream): ONE k_mask_annotate_q20_batch launch
// when the fast kernel takes all of them in the same variant, else a launch per part.
static int launch_mask_annotate_parts(hinge_ctx** ctxs, int n, const hinge_filter_params* p, int mode) {
    hinge_ctx* ctx = ctxs[0];
    int rc, thr = -1;
    if (n > K2_BATCH_MAX) {
        for (int k0 = 0; k0 < n; k0 += K2_BATCH_MAX)
            if ((rc = launch_mask_annotate_parts(ctxs + k0, std::min(n - k0, (int)K2_BATCH_MAX), p, mode))) return rc;
        return HINGE_OK;
    }
    bool batch = ctx->k2_batch != 0 && n > 1 && mode != MODE_FINAL;
    for (int k = 0; k < n && batch; k++) batch = k2_applies(ctxs[k], p, &thr);
    if (!batch) {
        for (int k = 0; k < n; k++)
            if ((rc = launch_mask_annotate(ctxs[k], p, mode))) return rc;
        return HINGE_OK;
    }
    K2Prep q[K2_BATCH_MAX];
    bool same = true;
    for (int k = 0; k < n; k++) {
        hinge_ctx* c = ctxs[k];
        if ((rc = flush_min_cov(c))) return rc;
        if ((rc = prepare_cov_out(c, p))) return rc;
        if (mode == MODE_SPEC) { c->n_wave_totals = 0; c->pass_mode = 1; }
        if ((rc = k2_prepare(c, p, mode, n, &q[k]))) return rc;
        q[k].mulpath_thr = thr;
        same = same && q[k].variant == q[0].variant && c->spec_band == ctx->spec_band;
    }
    if (!same) {
        for (int k = 0; k < n; k++)
            if ((rc = k2_launch_one(ctxs[k], p, q[k]))) return rc;
    } else {
        K2Batch B;
        memset(&B, 0, sizeof(B));
        B.n = n;
        int blocks = 0, slot = 0;
        for (int k = 0; k < n; k++) slot = std::max(slot, q[k].slot);
        const size_t lds = (size_t)WAVES_PER_BLOCK * slot * sizeof(int);
        const int visit = ctx->k2_steal ? n : 1;
        for (int k = 0; k < n; k++) {
            hinge_ctx* c = ctxs[k];
            K2Part& a = B.part[k];
            a.C = (const K2Const*)c->k2c.p;
            a.read_list = (const int*)c->bucket_list.p; a.row_ptr = (const int64_t*)c->row_ptr.p;
            a.a_span = c->use_span16 ? (const void*)c->span16.p : (const void*)c->a_span.p;
            a.rlen = (const int*)c->rlen.p; a.nbins0 = (const int*)c->nbins0.p;
            a.d_min_cov = (q[k].variant >> 3) != 0 ? &sc(c)->spec_min_cov : &sc(c)->min_cov;
            a.cov_out = q[k].cov_out; a.cov_off = (const long long*)c->cov_off_d.p; a.cov_nbins = (int*)c->cov_nb.p;
            a.heads = (unsigned*)c->k2_heads.p; a.cov_tot = (int*)c->cov_tot.p;
            a.n1 = q[k].n1; a.n2 = q[k].n2; a.n4 = q[k].n4; a.slot_ints = slot; a.cov_base = c->r_begin; a.n_heads = q[k].n_heads;
            a.first_block = blocks; a.n_blocks = q[k].g;
            memcpy(a.head_bases, q[k].base, sizeof(a.head_bases));
            blocks += (q[k].g + 7) & ~7;
            // where the part's counters stand after the launch: one draw per item + one failing draw per wavefront that visits the head
            const int nh = q[k].n_heads, fixed_k = q[k].n4 + (q[k].n2 + 1) / 2;
            for (int h = 0; h < nh; h++) q[k].next_base[h] = q[k].base[h] + (unsigned)(q[k].n1 / nh + (h < q[k].n1 % nh));
            for (int t = 0; t < visit; t++) {
                const int from = (k - t + n) % n, fixed_f = q[from].n4 + (q[from].n2 + 1) / 2;
                for (int j = 0; j < q[from].g - fixed_f; j++)
                    q[k].next_base[t == 0 ? j % nh : k2_visit_head(j, fixed_f & 7, fixed_k & 7, nh)] += WAVES_PER_BLOCK;
            }
        }
        ProfScope _ps(ctx, KID_MASK_ANNOTATE);
#define K2_ALL(PACKED, COVOUT, CUT20, SPEC)                                                                                                             \
        hipLaunchKernelGGL((k_mask_annotate_q20_batch<PACKED, COVOUT, CUT20, SPEC>), dim3(blocks), dim3(BLOCK), lds, ctx->stream, B, p->cut_off, thr, \
                           p->no_hinge_region, q[0].cov_mask_off, ctx->spec_band, visit | (ctx->k2_steal == 2 && visit > 1 ? 256 : 0))
        K2_DISPATCH(q[0].variant, K2_ALL);
#undef K2_ALL
        CK(hipGetLastError());
        for (int k = 0; k < n; k++) memcpy(ctxs[k]->k2_head_base, q[k].next_base, sizeof(ctxs[k]->k2_head_base));
    }
    for (int k = 0; k < n; k++) {
        hinge_ctx* c = ctxs[k];
        const int nr = c->r_end - c->r_begin + 1;
        if ((rc = k2_fallback(c, p, mode, std::max(1, std::min((nr + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK, 1 << 20))))) return rc;
    }
    return HINGE_OK;
}

int hinge_filter_mask_annotate(hinge_ctx* ctx, const hinge_filter_params* p) {
    int rc = check_params(ctx, p);
    if (rc) return rc;
    if (ctx->r_end < ctx->r_begin) return fail(ctx, HINGE_E_ARG, "no pile-ups set");
    CK(hipSetDevice(ctx->device));
    for (int attempt = 0; attempt < 8; attempt++) {
        CK(hipMemsetAsync(&sc(ctx)->status, 0, sizeof(int), ctx->stream));
        CK(hipMemsetAsync(sc(ctx)->shards, 0, sizeof(sc(ctx)->shards) + sizeof(unsigned), ctx->stream));   // + fallback_count (directly behind them)
        if ((rc = launch_mask_annotate(ctx, p))) return rc;
        // the annotation buffer is sized optimistically; grow + rerun on overflow (rare)
        Scalars h;
        CK(hipMemcpyAsync(&h, ctx->scalars.p, sizeof(Scalars), hipMemcpyDeviceToHost, ctx->stream));
        CK(hipStreamSynchronize(ctx->stream));
        if (h.status & ST_RANGE) return fail(ctx, HINGE_E_RANGE, "overlap coordinate beyond read length + cut_off");
        if (!(h.status & ST_ANNO_CAP)) return HINGE_OK;
        if ((rc = grow_annotations(ctx))) return rc;
    }
    return fail(ctx, HINGE_E_CAPACITY, "annotation buffer kept overflowing");
}

static HingePart hinge_part_of(hinge_ctx* ctx) {
    HingePart a;
    memset(&a, 0, sizeof(a));
    a.row_ptr = (const int64_t*)ctx->row_ptr.p; a.a_span = (const int2*)ctx->a_span.p; a.b_span = (const int2*)ctx->b_span.p;
    a.b_flag = (const unsigned*)ctx->b_flag.p; a.mask = (const int2*)ctx->mask;
    a.anno_buf = (const int2*)ctx->anno_buf.p; a.anno_off = (const unsigned*)ctx->anno_off.p; a.anno_cnt = (const int*)ctx->anno_cnt.p;
    a.work_list = (const WorkItem*)ctx->work_list.p; a.work_cap = ctx->work_cap; a.work_shard = (const unsigned*)(sc(ctx)->shards + N_SHARD * SHARD_STRIDE);
    a.hinge_flag = (unsigned char*)ctx->hinge_flag.p;
    a.heavy = (HeavyItem*)ctx->heavy_list.p; a.heavy_count = &sc(ctx)->heavy_count; a.heavy_count_big = &sc(ctx)->heavy_count_big; a.heavy_cap = ctx->anno_cap;
    a.heavy2 = (HeavyItem*)ctx->heavy2_list.p; a.heavy2_count = &sc(ctx)->heavy2_count; a.heavy2_count_big = &sc(ctx)->heavy2_count_big; a.work_next_light = &sc(ctx)->work_next_light;
    if (ctx->group_reads && ctx->rd_head.p && ctx->rd_next2.p && ctx->rl2.p) {
        a.rd_head = (unsigned long long*)ctx->rd_head.p; a.rd_next2 = (unsigned*)ctx->rd_next2.p; a.rl2 = (unsigned*)ctx->rl2.p;
        a.rl2_count = &sc(ctx)->rl2_count; a.rl2_count_big = &sc(ctx)->rl2_count_big;
        if (++ctx->pass_stamp == 0) ctx->pass_stamp = 1;
        a.stamp = ctx->pass_stamp;
    }
    a.exact_queue = (int2*)ctx->exact_queue.p; a.exact_count = &sc(ctx)->exact_count; a.exact_cap = ctx->exact_cap;
    a.status = &sc(ctx)->status; a.work_next = &sc(ctx)->work_next; a.work_next_big = &sc(ctx)->work_next_big; a.work_next_small = &sc(ctx)->work_next_small;
    a.dbg = ctx->debug_paths ? sc(ctx)->dbg : (unsigned*)nullptr;
    a.force_exact = ctx->force_exact;
    const bool packed = ctx->use_span16 && ctx->bspan16_state == 1;
    a.span16 = packed ? (const unsigned*)ctx->span16.p : (const unsigned*)nullptr;
    a.bspan16 = packed ? (const unsigned*)ctx->bspan16.p : (const unsigned*)nullptr;
    return a;
}

// The hinge kernels over n resident parts (contexts on one device and one stream) in one launch each; n = 1 is the single part.
// the 16|16 copy of b_span, once per set of pile-ups, before their first hinge pass (one sweep of 12 bytes per overlap and a
// four-byte read-back; the first pass of a part pays it - in bench.py that is a warm-up step, in `hinge filter` ~0.1 ms of 280)
static int ensure_bspan16(hinge_ctx* ctx) {
    if (ctx->bspan16_state != 0) return HINGE_OK;
    ctx->bspan16_state = -1;
    if (!ctx->use_span16 || ctx->n_ovl <= 0 || getenv("HINGE_COUNT_INT32")) return HINGE_OK;
    int rc = ensure(ctx, ctx->bspan16, sizeof(unsigned) * (size_t)ctx->n_ovl);
    if (rc) return rc;
    unsigned* bad = sc(ctx)->facts;     // (a scratch word outside the pass scalars' reset region: k_pileup_facts' own)
    CK(hipMemsetAsync(bad, 0, sizeof(unsigned), ctx->stream));
    hipLaunchKernelGGL(k_pack_bspan, dim3((unsigned)std::min<long long>((ctx->n_ovl + BLOCK - 1) / BLOCK, (long long)ctx->n_cu * 16)), dim3(BLOCK), 0, ctx->stream,
                       (long long)ctx->n_ovl, (const int2*)ctx->b_span.p, (unsigned*)ctx->bspan16.p, bad);
    CK(hipGetLastError());
    unsigned h = 1;
    CK(hipMemcpyAsync(&h, bad, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    if (h == 0) ctx->bspan16_state = 1;
    return HINGE_OK;
}

static int launch_hinges_batch(hinge_ctx** ctxs, int n, const hinge_filter_params* p) {
    hinge_ctx* ctx = ctxs[0];
    for (int k = 0; k < n; k++) { const int rc = ensure_bspan16(ctxs[k]); if (rc) return rc; }
    HingeBatch B;
    memset(&B, 0, sizeof(B));
    B.n = n;
    bool any_big = false;
    for (int k = 0; k < n; k++) {
        B.part[k] = hinge_part_of(ctxs[k]);
        any_big = any_big || ctxs[k]->max_pile > (unsigned)PO_CAP_SMALL;
    }
#ifdef HINGE_PROBE_SORTED_WORK
    // probe builds only (tools/probes/sorted_work_probe.sh): the work lists rewritten in STORAGE order (dense, sorted by row) before
    // k_hinge_count reads them - does the order in which the pile-ups are visited matter (address translation, DRAM pages)?
    if (getenv("HINGE_PROBE_SORTED_WORK")) {
        for (int k = 0; k < n; k++) {
            hinge_ctx* c = ctxs[k];
            Scalars h;
            CK(hipMemcpyAsync(&h, c->scalars.p, sizeof(Scalars), hipMemcpyDeviceToHost, c->stream));
            CK(hipStreamSynchronize(c->stream));
            std::vector<WorkItem> all((size_t)c->work_cap), live;
            CK(hipMemcpy(all.data(), c->work_list.p, sizeof(WorkItem) * all.size(), hipMemcpyDeviceToHost));
            const unsigned* cnt = h.shards + N_SHARD * SHARD_STRIDE;
            for (size_t w = 0; w < all.size(); w++)
                if ((unsigned)(w / N_SHARD) < cnt[(w % N_SHARD) * SHARD_STRIDE]) live.push_back(all[w]);
            const int mode = atoi(getenv("HINGE_PROBE_SORTED_WORK"));
            if (mode == 1) std::sort(live.begin(), live.end(), [](const WorkItem& a, const WorkItem& b) { return a.row < b.row; });
            if (mode == 2) std::sort(live.begin(), live.end(), [](const WorkItem& a, const WorkItem& b) { return a.n * (long long)((a.cnt + 3) / 4) > b.n * (long long)((b.cnt + 3) / 4); });
            const size_t M = live.size();
            for (size_t w = 0; w < M; w++) all[w] = live[w];
            for (int sh = 0; sh < N_SHARD; sh++) h.shards[(N_SHARD + sh) * SHARD_STRIDE] = (unsigned)((M + N_SHARD - 1 - sh) / N_SHARD);
            CK(hipMemcpy(c->work_list.p, all.data(), sizeof(WorkItem) * all.size(), hipMemcpyHostToDevice));
            CK(hipMemcpy((char*)c->scalars.p + offsetof(Scalars, shards), h.shards, sizeof(h.shards), hipMemcpyHostToDevice));
        }
    }
#endif
    { ProfScope _ps(ctx, KID_HINGE_COUNT);
    bool all_packed = true;      // (a batch with a part that has no packed copies runs the int32 form for all: the copies are both or neither per part)
    for (int k = 0; k < n; k++) all_packed = all_packed && B.part[k].span16 != nullptr;
    if (!all_packed) for (int k = 0; k < n; k++) { B.part[k].span16 = nullptr; B.part[k].bspan16 = nullptr; }
    if (ctx->count_waves == 2) {
        if (all_packed) hipLaunchKernelGGL((k_hinge_count<2, true>), dim3(std::max(n, (ctx->n_cu * 16 / n) * n)), dim3(2 * WAVE), 0, ctx->stream, to_dev(p), B);
        else hipLaunchKernelGGL((k_hinge_count<2, false>), dim3(std::max(n, (ctx->n_cu * 16 / n) * n)), dim3(2 * WAVE), 0, ctx->stream, to_dev(p), B);
    } else {
        if (all_packed) hipLaunchKernelGGL((k_hinge_count<4, true>), dim3(std::max(n, (ctx->n_cu * 8 / n) * n)), dim3(BLOCK), 0, ctx->stream, to_dev(p), B);
        else hipLaunchKernelGGL((k_hinge_count<4, false>), dim3(std::max(n, (ctx->n_cu * 8 / n) * n)), dim3(BLOCK), 0, ctx->stream, to_dev(p), B);
    } }
    CK(hipGetLastError());
    // Undecided annotations: pile-ups of up to PO_CAP_SMALL overlaps go through the 72 KiB instance, two workgroups per CU (one
    // round instead of two on the E. coli restatement); larger ones through the 144 KiB instance, launched only if a part has
    // such a pile-up (k_pileup_facts).
    // Open annotations: first the order-independent evaluation in 16 KiB of LDS for all of them at once (k_hinge_call_light); what
    // it passes on - the tie order decides, or more than LIGHT_CAP supporters - goes through the instances that can replay the sort
    bool light = ctx->hinge_light != 0;
    for (int k = 0; k < n; k++) light = light && ctxs[k]->force_exact == 0;
    { ProfScope _ps(ctx, KID_HINGE_CALL);
    if (light) {
        if (ctx->light_occ == 0) {
            int nb = 0;
            CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_hinge_call_light, BLOCK, 0));
            ctx->light_occ = std::max(nb, 1);
            if (const char* g = getenv("HINGE_LIGHT_WGS_PER_CU")) ctx->light_occ = std::max(atoi(g), 1);
            if (getenv("HINGE_DEBUG_PATHS")) fprintf(stderr, "[hinge] k_hinge_call_light: %d workgroups per CU (occupancy calculator: %d)\n", ctx->light_occ, nb);
        }
        hipLaunchKernelGGL(k_hinge_call_light, dim3(std::max(n, (ctx->light_occ * ctx->n_cu / n) * n)), dim3(BLOCK), 0, ctx->stream, to_dev(p), B);
    }
    // HINGE_CALL_MINI=1 (round 5, measured, off): items whose pile-up has at most PO_CAP_MINI overlaps through a 37 KiB instance, FOUR
    // workgroups per CU, the instance behind it only takes the rest.  No gain on the repeat-rich config (the replay is bound by its
    // slowest item, and an item gets slower with more workgroups on its CU), +5 us of empty launch on config 2.
    const bool mini = light && ctx->hinge_mini != 0;
    if (mini) hipLaunchKernelGGL(k_hinge_call<PO_CAP_MINI>, dim3(std::max(n, (4 * ctx->n_cu / n) * n)), dim3(BLOCK), 0, ctx->stream, to_dev(p), B, 0, 1, 0, PO_CAP_MINI, 1);
    bool any_mid = false;
    for (int k = 0; k < n; k++) any_mid = any_mid || ctxs[k]->max_pile > (unsigned)PO_CAP_MINI;
    const int n_min = mini ? PO_CAP_MINI + 1 : 0;
    if (mini && !any_mid) {
        // every pile-up fits the small instance: nothing is left for another launch
    } else if (light && any_big) {
        // behind the light kernel ONE second-tier launch: the full-size instance takes both ends of the list (an empty launch is 5 us)
        hipLaunchKernelGGL(k_hinge_call<PO_CAP>, dim3(std::max(n, (ctx->n_cu / n) * n)), dim3(BLOCK), 0, ctx->stream, to_dev(p), B, 1, 2, n_min, INT_MAX, 0);
    } else {
        // behind the light kernel: the LEAN instance (52 KiB of LDS: three workgroups per CU; HINGE_CALL_LEAN=0: the full one, two per CU)
        if (light && ctx->hinge_lean) hipLaunchKernelGGL((k_hinge_call<PO_CAP_SMALL, true>), dim3(std::max(n, (3 * ctx->n_cu / n) * n)), dim3(BLOCK), 0, ctx->stream, to_dev(p), B, 0, 1, n_min, INT_MAX, 0);
        else hipLaunchKernelGGL(k_hinge_call<PO_CAP_SMALL>, dim3(std::max(n, (2 * ctx->n_cu / n) * n)), dim3(BLOCK), 0, ctx->stream, to_dev(p), B, 0, light ? 1 : 0, n_min, INT_MAX, 0);
        if (any_big) hipLaunchKernelGGL(k_hinge_call<PO_CAP>, dim3(std::max(n, (ctx->n_cu / n) * n)), dim3(BLOCK), 0, ctx->stream, to_dev(p), B, 1, 0, 0, INT_MAX, 0);
    } }
    CK(hipGetLastError());
    // the serial exact path takes pile-ups or supporter lists beyond PO_CAP (and everything under force_exact == 1): per part
    for (int k = 0; k < n; k++) {
        hinge_ctx* c = ctxs[k];
        if (c->max_pile <= (unsigned)PO_CAP && c->force_exact != 1) continue;
        ProfScope _ps2(ctx, KID_HINGE_EXACT);
        hipLaunchKernelGGL(k_hinge_exact, dim3(64), dim3(64), 0, ctx->stream, to_dev(p), (const int64_t*)c->row_ptr.p,
                           (const int2*)c->a_span.p, (const int2*)c->b_span.p, (const unsigned*)c->b_flag.p, (const int2*)c->mask,
                           (const int2*)c->anno_buf.p, (const unsigned*)c->anno_off.p, (const int2*)c->exact_queue.p,
                           (const unsigned*)&sc(c)->exact_count, c->exact_cap, (int*)c->arena.p, &sc(c)->arena_used, c->arena_cap,
                           (unsigned char*)c->hinge_flag.p, &sc(c)->status);
        CK(hipGetLastError());
    }
    return HINGE_OK;
}

static int launch_hinges(hinge_ctx* ctx, const hinge_filter_params* p) { return launch_hinges_batch(&ctx, 1, p); }

// checks the capacity flags of the last hinge pass; grows what overflowed. 1 = rerun needed.
static int hinges_settle(hinge_ctx* ctx, int* rerun) {
    Scalars h;
    CK(hipMemcpyAsync(&h, ctx->scalars.p, sizeof(Scalars), hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    *rerun = 0;
    int rc;
    if (h.status & ST_QUEUE_CAP) {
        ctx->exact_cap = std::max(ctx->exact_cap * 2, h.exact_count + 1024);
        if ((rc = ensure(ctx, ctx->exact_queue, sizeof(int2) * (size_t)ctx->exact_cap))) return rc;
        *rerun = 1;
    }
    if (h.status & ST_ARENA_CAP) {
        ctx->arena_cap = std::max(ctx->arena_cap * 2, h.arena_used + 1024);
        if ((rc = ensure(ctx, ctx->arena, sizeof(int) * (size_t)ctx->arena_cap))) return rc;
        *rerun = 1;
    }
    return HINGE_OK;
}

int hinge_filter_hinges(hinge_ctx* ctx, const hinge_filter_params* p) {
    int rc = check_params(ctx, p);
    if (rc) return rc;
    CK(hipSetDevice(ctx->device));
    for (int attempt = 0; attempt < 16; attempt++) {
        CK(hipMemsetAsync(&sc(ctx)->status, 0, sizeof(int), ctx->stream));
        CK(hipMemsetAsync(&sc(ctx)->exact_count, 0, 11 * sizeof(unsigned), ctx->stream));   // + work_next, heavy_count, work_next_big, heavy_count_big, work_next_light, heavy2_count, heavy2_count_big, work_next_small, rl2_count, rl2_count_big
        CK(hipMemsetAsync(&sc(ctx)->arena_used, 0, sizeof(unsigned long long), ctx->stream));
        if ((rc = launch_hinges(ctx, p))) return rc;
        int rerun = 0;
        if ((rc = hinges_settle(ctx, &rerun))) return rc;
        if (!rerun) return HINGE_OK;
    }
    return fail(ctx, HINGE_E_CAPACITY, "exact-path buffers kept overflowing");
}

int hinge_filter_run(hinge_ctx* ctx, const hinge_filter_params* p) {
    // asynchronous single-part pipeline; capacity overflow is reported by the next result getter
    int rc = check_params(ctx, p);
    if (rc) return rc;
    if (ctx->r_end < ctx->r_begin) return fail(ctx, HINGE_E_ARG, "no pile-ups set");
    CK(hipSetDevice(ctx->device));
    ctx->min_cov_pending = true;   // single part: MIN_COV starts at the ini value (applied by the pass's first kernel)
    ctx->min_cov_value = p->min_cov;
    if ((rc = hinge_filter_sweep_batch_async(&ctx, 1, p, nullptr, 0))) return rc;
    if ((rc = hinge_filter_finish_batch_async(&ctx, 1, p))) return rc;
    if ((rc = launch_hinges(ctx, p))) return rc;
    return HINGE_OK;
}

static int check_status(hinge_ctx* ctx) {
    Scalars h;
    CK(hipMemcpyAsync(&h, ctx->scalars.p, sizeof(Scalars), hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    if (h.status & ST_NO_LONG_READ) return fail(ctx, HINGE_E_UNDEFINED, "no read >= 5000 bp in this part");
    if (h.status & ST_RANGE) return fail(ctx, HINGE_E_RANGE, "overlap coordinate beyond read length + cut_off");
    if (h.status & ST_MEDIAN_RANGE) return fail(ctx, HINGE_E_RANGE, "a mean coverage lies outside [0, 4096): all-gather the means and call hinge_filter_median instead of the histogram exchange");
    if (h.status & ST_REDO_CAP) return fail(ctx, HINGE_E_CAPACITY, "guard-band list overflow (cannot happen: it has a slot per read)");
    if (h.status & (ST_ANNO_CAP | ST_QUEUE_CAP | ST_ARENA_CAP))
        return fail(ctx, HINGE_E_CAPACITY, "device buffer overflow in hinge_filter_run: use the staged calls (they regrow)");
    return HINGE_OK;
}

int hinge_filter_get_masks(hinge_ctx* ctx, int32_t* mask, int32_t* cmask, uint8_t* flags) {
    if (!ctx) return HINGE_E_ARG;
    int rc = check_status(ctx);
    if (rc) return rc;
    const size_t n = (size_t)(ctx->r_end - ctx->r_begin + 1);
    if (mask) CK(hipMemcpyAsync(mask, ctx->mask + ctx->r_begin, sizeof(int2) * n, hipMemcpyDeviceToHost, ctx->stream));
    if (cmask) CK(hipMemcpyAsync(cmask, (int2*)ctx->cmask.p + ctx->r_begin, sizeof(int2) * n, hipMemcpyDeviceToHost, ctx->stream));
    if (flags) CK(hipMemcpyAsync(flags, (unsigned char*)ctx->rflags.p + ctx->r_begin, n, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    return HINGE_OK;
}

int hinge_filter_get_annotations(hinge_ctx* ctx, int64_t* off, int32_t* pos, int32_t* type, uint8_t* is_hinge) {
    if (!ctx || !off) return HINGE_E_ARG;
    int rc = check_status(ctx);
    if (rc) return rc;
    const size_t n = (size_t)(ctx->r_end - ctx->r_begin + 1);
    std::vector<unsigned> aoff(n);
    std::vector<int> acnt(n);
    CK(hipMemcpyAsync(aoff.data(), (unsigned*)ctx->anno_off.p + ctx->r_begin, sizeof(unsigned) * n, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipMemcpyAsync(acnt.data(), (int*)ctx->anno_cnt.p + ctx->r_begin, sizeof(int) * n, hipMemcpyDeviceToHost, ctx->stream));
    Scalars h;
    CK(hipMemcpyAsync(&h, ctx->scalars.p, sizeof(Scalars), hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    off[0] = 0;
    for (size_t i = 0; i < n; i++) off[i + 1] = off[i] + acnt[i];
    if (!pos) return HINGE_OK;
    size_t tot = 0;                                  // (the shards' regions are not contiguous: copy up to the last slot in use)
    for (size_t i = 0; i < n; i++) if (acnt[i] > 0) tot = std::max(tot, (size_t)aoff[i] + (size_t)acnt[i]);
    std::vector<int2> buf(tot);
    std::vector<unsigned char> hf(tot);
    if (tot) {
        CK(hipMemcpyAsync(buf.data(), ctx->anno_buf.p, sizeof(int2) * tot, hipMemcpyDeviceToHost, ctx->stream));
        CK(hipMemcpyAsync(hf.data(), ctx->hinge_flag.p, tot, hipMemcpyDeviceToHost, ctx->stream));
        CK(hipStreamSynchronize(ctx->stream));
    }
    for (size_t i = 0; i < n; i++)
        for (int t = 0; t < acnt[i]; t++) {
            const size_t d = (size_t)off[i] + t, s = (size_t)aoff[i] + t;
            pos[d] = buf[s].x;
            if (type) type[d] = buf[s].y;
            if (is_hinge) is_hinge[d] = hf[s];
        }
    return HINGE_OK;
}

int hinge_filter_coverage_out(hinge_ctx* ctx, int enable) {
    if (!ctx) return HINGE_E_ARG;
    ctx->cov_out_on = enable != 0;
    if (!enable) ctx->cov_valid = false;
    return HINGE_OK;
}

int hinge_filter_get_coverage(hinge_ctx* ctx, int64_t* off, int32_t* nbins, int32_t* cov, int64_t cov_cap) {
    if (!ctx || !off) return HINGE_E_ARG;
    if (!ctx->cov_out_on || !ctx->cov_valid) return fail(ctx, HINGE_E_ARG, "hinge_filter_get_coverage: no K2 pass with hinge_filter_coverage_out(1) on these pile-ups");
    const size_t nr = (size_t)(ctx->r_end - ctx->r_begin + 1);
    memcpy(off, ctx->h_cov_off.data(), sizeof(int64_t) * (nr + 1));
    if (!nbins && !cov) return HINGE_OK;
    CK(hipSetDevice(ctx->device));
    int rc = check_status(ctx);
    if (rc) return rc;
    if (cov && cov_cap < off[nr]) return fail(ctx, HINGE_E_ARG, "hinge_filter_get_coverage: cov_cap too small");
    if (nbins) CK(hipMemcpyAsync(nbins, ctx->cov_nb.p, sizeof(int) * nr, hipMemcpyDeviceToHost, ctx->stream));
    if (cov && off[nr] > 0) CK(hipMemcpyAsync(cov, ctx->cov_buf.p, sizeof(int) * (size_t)off[nr], hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    return HINGE_OK;
}

int hinge_filter_coverage_bins(hinge_ctx* ctx, int32_t r0, int32_t r1, int32_t reso, int32_t cutoff, int32_t* nbins, int32_t* cov,
                               int64_t cov_cap) {
    if (!ctx || !nbins || r0 < 0 || r1 >= ctx->n_reads || r1 < r0 || reso <= 0) return fail(ctx, HINGE_E_ARG, "coverage_bins: bad arguments");
    CK(hipSetDevice(ctx->device));
    const size_t n = (size_t)(r1 - r0 + 1);
    struct Tmp {   // freed on every return path
        int* nb = nullptr; int64_t* off = nullptr; int* cov = nullptr;
        ~Tmp() { if (nb) (void)hipFree(nb); if (off) (void)hipFree(off); if (cov) (void)hipFree(cov); }
    } d;
    CK(hipMalloc(&d.nb, sizeof(int) * n));
    const int kcap = ((ctx->max_rlen + std::max(cutoff, 0)) / reso + 4 + 3) & ~3;
    const size_t lds = (size_t)WAVES_PER_BLOCK * kcap * sizeof(int);
    if (lds > 160 * 1024) return fail(ctx, HINGE_E_RANGE, "coverage_bins: read too long for the LDS histogram");
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k_coverage_bins, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int grid = grid_for_reads(ctx, (int)n, WAVES_PER_BLOCK);
    // its own status word: the pass status (capacity / median flags of an asynchronous pass) is not this function's to clear
    int* st = &sc(ctx)->bins_status;
    CK(hipMemsetAsync(st, 0, sizeof(int), ctx->stream));
    {
        ProfScope _ps(ctx, KID_COVERAGE_BINS);
        hipLaunchKernelGGL(k_coverage_bins, dim3(grid), dim3(BLOCK), lds, ctx->stream, r0, r1, (const int64_t*)ctx->row_ptr.p,
                           (const int2*)ctx->a_span.p, reso, cutoff, kcap, d.nb, (const int64_t*)nullptr, (int*)nullptr, st);
    }
    CK(hipGetLastError());
    CK(hipMemcpyAsync(nbins, d.nb, sizeof(int) * n, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    if (!cov) return HINGE_OK;
    std::vector<int64_t> off(n + 1, 0);
    for (size_t i = 0; i < n; i++) off[i + 1] = off[i] + nbins[i];
    if (off[n] > cov_cap) return fail(ctx, HINGE_E_ARG, "coverage_bins: cov_cap too small");
    if (off[n] == 0) return HINGE_OK;
    CK(hipMalloc(&d.off, sizeof(int64_t) * (n + 1)));
    CK(hipMalloc(&d.cov, sizeof(int) * (size_t)off[n]));
    CK(hipMemcpyAsync(d.off, off.data(), sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice, ctx->stream));
    {
        ProfScope _ps(ctx, KID_COVERAGE_BINS);
        hipLaunchKernelGGL(k_coverage_bins, dim3(grid), dim3(BLOCK), lds, ctx->stream, r0, r1, (const int64_t*)ctx->row_ptr.p,
                           (const int2*)ctx->a_span.p, reso, cutoff, kcap, d.nb, (const int64_t*)d.off, d.cov, st);
    }
    CK(hipGetLastError());
    CK(hipMemcpyAsync(cov, d.cov, sizeof(int) * (size_t)off[n], hipMemcpyDeviceToHost, ctx->stream));
    int stv = 0;
    CK(hipMemcpyAsync(&stv, st, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    if (stv & ST_RANGE) return fail(ctx, HINGE_E_RANGE, "coverage_bins: bins exceed LDS capacity");
    return HINGE_OK;
}

int hinge_filter_counters(hinge_ctx* ctx, int64_t out[4]) {
    if (!ctx || !out) return HINGE_E_ARG;
    Scalars h;
    CK(hipMemcpyAsync(&h, ctx->scalars.p, sizeof(Scalars), hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    out[0] = shard_sum(h.shards, 1);
    out[1] = std::min(h.exact_count, ctx->exact_cap);
    out[2] = shard_sum(h.shards, 0);
    int64_t nh = 0;
    {   // hinges = flags set in the slots the shards handed out
        const size_t R = ctx->anno_cap / N_SHARD;
        std::vector<unsigned char> hf((size_t)ctx->anno_cap);
        if (out[2]) CK(hipMemcpy(hf.data(), ctx->hinge_flag.p, hf.size(), hipMemcpyDeviceToHost));
        for (int sh = 0; sh < N_SHARD; sh++) {
            const size_t used = std::min<size_t>(h.shards[(size_t)sh * SHARD_STRIDE], R);
            for (size_t k = 0; k < used; k++) nh += hf[(size_t)sh * R + k];
        }
    }
    out[3] = nh;
    if (getenv("HINGE_DEBUG_PATHS"))
        fprintf(stderr, "[hinge] cumulative hinge-call paths: none=%u lds=%u exact=%u shortcut=%u | pile-up sorts=%u ordered=%u max_sup=%u max_anno=%u | last pass: %u items in the half-size instance, %u in the full-size one\n",
                h.dbg[0], h.dbg[1], h.dbg[2], h.dbg[3], h.dbg[4], h.dbg[5], h.dbg[6], h.dbg[7], h.heavy_count, h.heavy_count_big);
    if (getenv("HINGE_DEBUG_PATHS"))
        fprintf(stderr, "[hinge] last pass: k_hinge_call_light drew %u times and passed %u + %u items on to k_hinge_call<CAP> (cursors there: %u, %u)\n", h.work_next_light, h.heavy2_count, h.heavy2_count_big, h.work_next, h.work_next_big);
    if (getenv("HINGE_DEBUG_PATHS") && h.dbg[15])
        fprintf(stderr, "[hinge] k_hinge_count per read (HINGE_TIMING builds): reads=%u mean %.1f us max %.1f us, largest pile-up %u\n", h.dbg[5], h.dbg[4] * 0.01 / std::max(1u, h.dbg[5]), h.dbg[15] * 0.01, h.dbg[7]);
    if (getenv("HINGE_DEBUG_PATHS") && h.dbg[10] && h.dbg[15])
        fprintf(stderr, "[hinge] k_hinge_call_light (HINGE_TIMING builds): slowest item %.1f us, largest pile-up among the items %u, last item done %.1f us after its workgroup started\n", h.dbg[15] * 0.01, h.dbg[7], h.dbg[6] * 0.01);
    if (getenv("HINGE_DEBUG_PATHS") && h.dbg[10])
        fprintf(stderr, "[hinge] timing (10 ns ticks, cumulative): items=%u gather=%u (mean %.1f us) eval=%u (mean %.1f us) mean_n=%.0f mean_sup=%.0f | bin %.1f us scan %.1f us\n", h.dbg[10],
                h.dbg[8], h.dbg[8] * 0.01 / h.dbg[10], h.dbg[11], h.dbg[11] * 0.01 / h.dbg[10], (double)h.dbg[9] / h.dbg[10], (double)h.dbg[12] / h.dbg[10], h.dbg[13] * 0.01 / h.dbg[10], h.dbg[14] * 0.01 / h.dbg[10]);
    if (getenv("HINGE_DEBUG_PATHS") && h.dbg[20])
        fprintf(stderr, "[hinge] replay (HINGE_TIMING builds, cumulative): items=%u mean_n=%.0f mean_sup=%.0f | keys + pile-up sort %.1f us, supporters into pile-up order %.1f us, supporter sort %.1f us, scan %.1f us per item; slowest item (draw to flag) %.1f us\n",
                h.dbg[20], (double)h.dbg[21] / h.dbg[20], (double)h.dbg[22] / h.dbg[20], h.dbg[16] * 0.01 / h.dbg[20], h.dbg[17] * 0.01 / h.dbg[20], h.dbg[18] * 0.01 / h.dbg[20],
                h.dbg[19] * 0.01 / h.dbg[20], h.dbg[23] * 0.01);
    return HINGE_OK;
}

// staged launches without the host round trip (multi-GPU pipeline: collectives sit between them);
// capacity / range flags accumulate in the status word and are reported by hinge_filter_check.
int hinge_filter_mask_annotate_async(hinge_ctx* ctx, const hinge_filter_params* p) {
    int rc = check_params(ctx, p);
    if (rc) return rc;
    return launch_mask_annotate(ctx, p);
}
int hinge_filter_hinges_async(hinge_ctx* ctx, const hinge_filter_params* p) {
    int rc = check_params(ctx, p);
    if (rc) return rc;
    return launch_hinges(ctx, p);
}
static int same_device_and_stream(hinge_ctx** ctxs, int32_t n, int max_n, const char* who) {
    if (!ctxs || n <= 0 || n > max_n) return HINGE_E_ARG;
    for (int k = 0; k < n; k++)
        if (!ctxs[k] || ctxs[k]->device != ctxs[0]->device || ctxs[k]->stream != ctxs[0]->stream)
            return fail(ctxs[0], HINGE_E_ARG, (std::string(who) + ": the contexts must share one device and one stream").c_str());
    return HINGE_OK;
}
// every context of a batched launch must have what the kernel dereferences (a null table pointer faults on the device)
static int batch_parts_ready(hinge_ctx** ctxs, int32_t n, const char* who) {
    for (int k = 0; k < n; k++) {
        hinge_ctx* c = ctxs[k];
        if (c->r_end < c->r_begin || !c->row_ptr.p) return fail(ctxs[0], HINGE_E_ARG, (std::string(who) + ": a context has no pile-ups set").c_str());
        if (!c->mask) return fail(ctxs[0], HINGE_E_ARG, (std::string(who) + ": a context has no mask table (hinge_set_reads / hinge_attach_mask_table)").c_str());
        if (!c->anno_buf.p || !c->anno_off.p || !c->anno_cnt.p || !c->work_list.p) return fail(ctxs[0], HINGE_E_ARG, (std::string(who) + ": a context has no annotations yet (run the mask / annotate pass first)").c_str());
    }
    return HINGE_OK;
}
// (batched kernels record their HIP-event time on the batch's FIRST context: hinge_profile_report of the others does not include them)
int hinge_filter_hinges_batch_async(hinge_ctx** ctxs, int32_t n, const hinge_filter_params* p) {
    int rc = same_device_and_stream(ctxs, n, HINGE_BATCH_MAX, "hinge_filter_hinges_batch_async");
    if (rc) return rc;
    if ((rc = batch_parts_ready(ctxs, n, "hinge_filter_hinges_batch_async"))) return rc;
    hinge_ctx* ctx = ctxs[0];
    if ((rc = check_params(ctx, p))) return rc;
    CK(hipSetDevice(ctx->device));
    return launch_hinges_batch(ctxs, n, p);
}
int hinge_filter_check(hinge_ctx* ctx) {
    if (!ctx) return HINGE_E_ARG;
    return check_status(ctx);
}

// ---- the one-sweep pass (round 4; filter_kernels.h "the one-sweep pass") ------------------------------------------------
// per-read bin counts / well-formedness of the current pile-ups (nbins0[]): a fact of the pile-ups like the 16|16 span copy,
// made once per hinge_set_pileups and reso (k_cov_stats is the kernel that knows how)
static int ensure_pile_bins(hinge_ctx* ctx, const hinge_filter_params* p) {
    if (ctx->nbins0_reso == p->reso) return HINGE_OK;
    return launch_stats(ctx, p);
}
static int launch_spec_predict(hinge_ctx** ctxs, int n, const hinge_filter_params* p) {
    hinge_ctx* ctx = ctxs[0];
    static_assert(SCALARS_RESET_BYTES % sizeof(int) == 0, "reset region is whole ints");
    for (int packed = 0; packed < 2; packed++) {   // (one launch per span format; a rank's parts normally share one)
        SpecBatch B;
        memset(&B, 0, sizeof(B));
        B.ns = ctx->spec_ns;
        int max_nr = 1;
        for (int k = 0; k < n; k++) {
            hinge_ctx* c = ctxs[k];
            if ((c->use_span16 ? 1 : 0) != packed) continue;
            int rc = ensure(c, c->spec_sample, sizeof(int) * (size_t)ctx->spec_ns);
            if (rc) return rc;
            SpecPart& a = B.part[B.n++];
            a.r_begin = c->r_begin; a.r_end = c->r_end;
            a.row_ptr = (const int64_t*)c->row_ptr.p; a.a_span = (const int2*)c->a_span.p; a.span16 = (const unsigned*)c->span16.p;
            a.rlen = (const int*)c->rlen.p; a.nbins0 = (const int*)c->nbins0.p;
            a.pass_scalars = (int*)c->scalars.p; a.n_pass_scalars = (int)(SCALARS_RESET_BYTES / sizeof(int));
            a.min_cov = &sc(c)->min_cov; a.set_min_cov = c->min_cov_pending ? 1 : 0; a.min_cov_value = c->min_cov_value;
            c->min_cov_pending = false;
            a.spec_min_cov = &sc(c)->spec_min_cov; a.sample = (int*)c->spec_sample.p; a.ticket = &sc(c)->spec_ticket; a.bias = c->spec_bias;
            max_nr = std::max(max_nr, c->r_end - c->r_begin + 1);
        }
        if (B.n == 0) continue;
        const int bpp = (std::min(B.ns, max_nr) + SPEC_READS_PER_BLOCK - 1) / SPEC_READS_PER_BLOCK;
        ProfScope _ps(ctx, KID_SPEC_PREDICT);
        if (p->reso == 40) { if (packed) hipLaunchKernelGGL((k_spec_predict<40, true>), dim3(bpp * B.n), dim3(SPEC_BLOCK), 0, ctx->stream, B, p->reso, p->est_cov);
                             else hipLaunchKernelGGL((k_spec_predict<40, false>), dim3(bpp * B.n), dim3(SPEC_BLOCK), 0, ctx->stream, B, p->reso, p->est_cov); }
        else { if (packed) hipLaunchKernelGGL((k_spec_predict<0, true>), dim3(bpp * B.n), dim3(SPEC_BLOCK), 0, ctx->stream, B, p->reso, p->est_cov);
               else hipLaunchKernelGGL((k_spec_predict<0, false>), dim3(bpp * B.n), dim3(SPEC_BLOCK), 0, ctx->stream, B, p->reso, p->est_cov); }
        CK(hipGetLastError());
    }
    return HINGE_OK;
}

int hinge_filter_sweep_batch_async(hinge_ctx** ctxs, int32_t n, const hinge_filter_params* p, uint32_t* hist_dev, int64_t row_stride) {
    int rc = same_device_and_stream(ctxs, n, MED_BATCH_MAX, "hinge_filter_sweep_batch_async");
    if (rc) return rc;
    hinge_ctx* ctx = ctxs[0];
    if ((rc = check_params(ctx, p))) return rc;
    if (hist_dev && row_stride < MED_BINS + 2) return fail(ctx, HINGE_E_ARG, "sweep_batch: a histogram row has MED_BINS + 2 words");
    for (int k = 0; k < n; k++)
        if (ctxs[k]->r_end < ctxs[k]->r_begin) return fail(ctxs[k], HINGE_E_ARG, "no pile-ups set");
    CK(hipSetDevice(ctx->device));
    int32_t lo[MED_BATCH_MAX], hi[MED_BATCH_MAX];
    uint32_t* hd[MED_BATCH_MAX];
    for (int k = 0; k < n; k++) { lo[k] = ctxs[k]->r_begin; hi[k] = ctxs[k]->r_end; hd[k] = hist_dev ? hist_dev + (int64_t)k * row_stride : nullptr; }
    // the telomere test (filter.cpp:731-760) sums max(cov, MIN_COV): not constant over a band of MIN_COV values - two sweeps
    const bool one_sweep = ctx->one_sweep != 0 && p->delete_telomere == 0;
    if (!one_sweep) {
        for (int k = 0; k < n; k++)
            if ((rc = launch_stats(ctxs[k], p))) return rc;
        return launch_median_batch(ctxs, n, p, lo, hi, hist_dev ? hd : nullptr);
    }
    for (int k = 0; k < n; k++)
        if ((rc = ensure_pile_bins(ctxs[k], p))) return rc;
    if ((rc = launch_spec_predict(ctxs, n, p))) return rc;
    if ((rc = launch_mask_annotate_parts(ctxs, n, p, MODE_SPEC))) return rc;
    return launch_median_batch(ctxs, n, p, lo, hi, hist_dev ? hd : nullptr);
}

int hinge_filter_finish_batch_async(hinge_ctx** ctxs, int32_t n, const hinge_filter_params* p) {
    int rc = same_device_and_stream(ctxs, n, MED_BATCH_MAX, "hinge_filter_finish_batch_async");
    if (rc) return rc;
    hinge_ctx* ctx = ctxs[0];
    if ((rc = check_params(ctx, p))) return rc;
    CK(hipSetDevice(ctx->device));
    bool all_final = true;
    for (int k = 0; k < n; k++) all_final = all_final && ctxs[k]->pass_mode != 0;
    if (!all_final || ctx->final_batched == 0) {
        for (int k = 0; k < n; k++)
            if ((rc = launch_mask_annotate(ctxs[k], p, ctxs[k]->pass_mode != 0 ? MODE_FINAL : MODE_CLASSIC))) return rc;
        return HINGE_OK;
    }
    // the guard-band lists of all parts in one launch per MASK_FINAL_BATCH_MAX parts (k_mask_final_batch)
    for (int k0 = 0; k0 < n; k0 += MASK_FINAL_BATCH_MAX) {
        const int nb = std::min(n - k0, (int)MASK_FINAL_BATCH_MAX);
        MaskFinalBatch B;
        memset(&B, 0, sizeof(B));
        B.n = nb;
        int kcap = 0, nr_max = 1;
        for (int k = 0; k < nb; k++) {
            hinge_ctx* c = ctxs[k0 + k];
            if ((rc = flush_min_cov(c))) return rc;
            if ((rc = prepare_cov_out(c, p))) return rc;
            kcap = std::max(kcap, kcap_for(c, p));
            nr_max = std::max(nr_max, c->r_end - c->r_begin + 1);
            MaskFinalPart& a = B.part[k];
            a.r_begin = c->r_begin; a.r_end = c->r_end;
            a.row_ptr = (const int64_t*)c->row_ptr.p; a.a_span = (const int2*)c->a_span.p; a.rlen = (const int*)c->rlen.p;
            a.d_min_cov = &sc(c)->min_cov;
            a.o = anno_out(c);
            a.read_list = (const int*)c->redo_list.p; a.list_count = &sc(c)->redo_count;
            if ((rc = spec_args_of(c, MODE_FINAL, 0, &a.sa))) return rc;
        }
        const size_t lds = (size_t)WAVES_PER_BLOCK * 2 * kcap * sizeof(int);
        if (lds > 160 * 1024) return fail(ctx, HINGE_E_RANGE, "read too long for the LDS histogram (max ~200 kb)");
        if (lds > 48 * 1024 && lds > ctx->lds_attr_final) {
            CK(hipFuncSetAttribute((const void*)k_mask_final_batch<40>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            CK(hipFuncSetAttribute((const void*)k_mask_final_batch<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            ctx->lds_attr_final = lds;
        }
        if ((rc = ensure(ctx, ctx->final_batch, sizeof(MaskFinalBatch)))) return rc;
        if (!ctx->final_batch_valid || memcmp(&B, &ctx->final_batch_host, sizeof(B)) != 0) {
            ctx->final_batch_host = B;
            CK(hipMemcpyAsync(ctx->final_batch.p, &ctx->final_batch_host, sizeof(B), hipMemcpyHostToDevice, ctx->stream));
            ctx->final_batch_valid = true;
        }
        // (as many workgroups per part as the single launch takes: the lists are ~1 % of the reads, a whole part if the band was missed)
        const int gper = std::max(64, std::min((nr_max + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK, std::max(nr_max / 128, 1)));
        ProfScope _ps(ctx, KID_MASK_FINAL);
        if (p->reso == 40) hipLaunchKernelGGL(k_mask_final_batch<40>, dim3(gper * nb), dim3(BLOCK), lds, ctx->stream, to_dev(p), (const MaskFinalBatch*)ctx->final_batch.p, kcap);
        else hipLaunchKernelGGL(k_mask_final_batch<0>, dim3(gper * nb), dim3(BLOCK), lds, ctx->stream, to_dev(p), (const MaskFinalBatch*)ctx->final_batch.p, kcap);
        CK(hipGetLastError());
    }
    return HINGE_OK;
}

int hinge_filter_sweep(hinge_ctx* ctx, const hinge_filter_params* p, hinge_cov_estimate* out) {
    int rc = check_params(ctx, p);
    if (rc) return rc;
    if (ctx->r_end < ctx->r_begin) return fail(ctx, HINGE_E_ARG, "no pile-ups set");
    CK(hipSetDevice(ctx->device));
    for (int attempt = 0; attempt < 8; attempt++) {
        // (the pass's first kernel clears the pass scalars: status, annotation allocator, work list, guard-band list)
        if ((rc = hinge_filter_sweep_batch_async(&ctx, 1, p, nullptr, 0))) return rc;
        if ((rc = hinge_filter_finish_batch_async(&ctx, 1, p))) return rc;
        Scalars h;
        CK(hipMemcpyAsync(&h, ctx->scalars.p, sizeof(Scalars), hipMemcpyDeviceToHost, ctx->stream));
        CK(hipStreamSynchronize(ctx->stream));
        if (out) { out->cov_est = h.est[0]; out->n_long = h.est[1]; out->total_cov = (int64_t)h.totals[0]; out->num_slot = (int64_t)h.totals[1]; }
        if (h.status & ST_NO_LONG_READ) return fail(ctx, HINGE_E_UNDEFINED, "no read >= 5000 bp in this part: the reference is undefined here (filter.cpp:660-666)");
        if (h.status & ST_RANGE) return fail(ctx, HINGE_E_RANGE, "overlap coordinate beyond read length + cut_off");
        if (h.status & ST_REDO_CAP) return fail(ctx, HINGE_E_CAPACITY, "guard-band list overflow (cannot happen: it has a slot per read)");
        if (!(h.status & ST_ANNO_CAP)) return HINGE_OK;
        // the annotation buffer is sized optimistically; grow + rerun the pass on overflow (rare; MIN_COV's update is idempotent)
        if ((rc = grow_annotations(ctx))) return rc;
    }
    return fail(ctx, HINGE_E_CAPACITY, "annotation buffer kept overflowing");
}

int hinge_filter_spec_stats(hinge_ctx* ctx, int64_t out[6]) {
    if (!ctx || !out) return HINGE_E_ARG;
    Scalars h;
    CK(hipMemcpyAsync(&h, ctx->scalars.p, sizeof(Scalars), hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    out[0] = h.spec_stats[0]; out[1] = h.spec_stats[1]; out[2] = h.spec_stats[2];
    out[3] = ctx->pass_mode != 0 ? (int64_t)h.redo_count : -1;
    out[4] = h.spec_min_cov; out[5] = h.min_cov;
    return HINGE_OK;
}

int hinge_debug_spec(hinge_ctx* ctx, int band, int sample, int bias) {
    if (!ctx) return HINGE_E_ARG;
    if (band >= 0) ctx->spec_band = band;
    if (sample > 0) ctx->spec_ns = sample;
    ctx->spec_bias = bias;
    return HINGE_OK;
}

int hinge_profile_enable(hinge_ctx* ctx, int max_launches) {
    if (!ctx) return HINGE_E_ARG;
    CK(hipSetDevice(ctx->device));
    ctx->prof_on = max_launches > 0;
    ctx->prof_used = 0;
    ctx->prof_kid.clear();
    while (ctx->prof_pool.size() < 2 * (size_t)std::max(max_launches, 0)) {
        hipEvent_t e;
        CK(hipEventCreate(&e));
        ctx->prof_pool.push_back(e);
    }
    return HINGE_OK;
}
// total milliseconds and launch count per kernel since hinge_profile_enable; arrays of hinge_profile_kernels() entries
int hinge_profile_select(hinge_ctx* ctx, uint32_t kernel_mask) {
    if (!ctx) return HINGE_E_ARG;
    ctx->prof_mask = kernel_mask;
    return HINGE_OK;
}

int hinge_profile_kernels(void) { return KID_COUNT; }
const char* hinge_profile_kernel_name(int id) { return (id >= 0 && id < KID_COUNT) ? KERNEL_NAMES[id] : ""; }
int hinge_profile_report(hinge_ctx* ctx, double* total_ms, int64_t* count) {
    if (!ctx || !total_ms || !count) return HINGE_E_ARG;
    CK(hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < KID_COUNT; k++) { total_ms[k] = 0; count[k] = 0; }
    for (size_t i = 0; i < ctx->prof_kid.size(); i++) {
        float ms = 0;
        CK(hipEventElapsedTime(&ms, ctx->prof_pool[2 * i], ctx->prof_pool[2 * i + 1]));
        total_ms[ctx->prof_kid[i]] += ms;
        count[ctx->prof_kid[i]] += 1;
    }
    return HINGE_OK;
}

int hinge_timer_start(hinge_ctx* ctx) {
    if (!ctx) return HINGE_E_ARG;
    CK(hipEventRecord(ctx->ev0, ctx->stream));
    return HINGE_OK;
}
int hinge_timer_stop_ms(hinge_ctx* ctx, float* ms) {
    if (!ctx || !ms) return HINGE_E_ARG;
    CK(hipEventRecord(ctx->ev1, ctx->stream));
    CK(hipEventSynchronize(ctx->ev1));
    CK(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return HINGE_OK;
}

}  // extern "C"

#include "align_capi.inc"
#include "consensus_capi.inc"
#include "draft_capi.inc"
#include "comm_capi.inc"

#ifdef HINGE_K2_TRACE
// Trace builds only (tools/k2_trace.py): a device buffer of 5 * n_items time stamps for k_mask_annotate_q20.
extern "C" int hinge_debug_k2_trace_begin(long long n_items) {
    unsigned long long* p = nullptr;
    if (hipMalloc(&p, (size_t)n_items * 5 * 8) != hipSuccess) return -1;
    hipMemset(p, 0, (size_t)n_items * 5 * 8);
    return hipMemcpyToSymbol(HIP_SYMBOL(hinge::g_k2_trace), &p, sizeof p) == hipSuccess ? 0 : -1;
}
extern "C" int hinge_debug_k2_trace_end(unsigned long long* out, long long n_items) {
    unsigned long long* p = nullptr;
    hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(&p, HIP_SYMBOL(hinge::g_k2_trace), sizeof p) != hipSuccess || !p) return -1;
    hipMemcpy(out, p, (size_t)n_items * 5 * 8, hipMemcpyDeviceToHost);
    unsigned long long* z = nullptr;
    hipMemcpyToSymbol(HIP_SYMBOL(hinge::g_k2_trace), &z, sizeof z);
    hipFree(p);
    return 0;
}
#endif
