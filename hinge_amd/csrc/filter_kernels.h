// HIP kernels (gfx950 / CDNA4) of the `hinge filter` hot path.  Integer / indexing work, HBM-bound:
// no MFMA.  One 64-lane wavefront owns one A read; its pile-up is streamed with coalesced 8-byte
// loads, turned into difference histograms with LDS atomics, prefix-scanned with DPP row shifts,
// and the mask / annotation / gate logic runs on the scanned bins while they are still in LDS.
//
// Reference semantics restated per kernel (file:line under /root/reference/src):
//   k_cov_stats       profileCoverage(cutoff 0) sums as used by filter/filter.cpp:642-656
//   k_median_hist     nth_element median + MIN_COV update, filter/filter.cpp:660-678
//   k_mask_annotate   filter/filter.cpp:696-829 and the gate of :842-865
//   k_hinge_call      filter/filter.cpp:867-1068, tie order of std::sort replayed in LDS
//   k_hinge_exact     the same for the rare cases that need the whole pile-up's std::sort order
//                     (filter.cpp:565-567) or overflow the LDS lists
//   k_coverage_bins   lib/LAInterface.cpp:4298-4320 (materialised bins for .coverage.txt)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>
#include "stdsort_emul.h"

namespace hinge {

constexpr int WAVE = 64;
constexpr int WAVES_PER_BLOCK = 4;
constexpr int BLOCK = WAVE * WAVES_PER_BLOCK;
constexpr int MEAN_SENTINEL = INT_MIN;  // mean_cov of reads that do not enter the median

constexpr int MED_BINS = 4096;  // one-pass median histogram range
constexpr int LOADS_IN_FLIGHT = 8;   // 8-byte pile-up loads a lane issues back to back (4 KiB per wave)

// status word bits (device -> host)
constexpr int ST_RANGE = 1;        // bin index beyond the LDS histogram
constexpr int ST_ANNO_CAP = 2;     // annotation buffer full
constexpr int ST_QUEUE_CAP = 4;    // exact queue full
constexpr int ST_ARENA_CAP = 8;    // exact-path scratch arena full
constexpr int ST_NO_LONG_READ = 16;
constexpr int ST_MEDIAN_RANGE = 32;   // sharded median: a mean coverage outside [0, MED_BINS), the histogram exchange cannot be used
// The annotation allocator and the work list are SHARDED (round 4): a returning device-scope atomic on ONE word serialises at
// ~12 ns, and both used to be single words - nothing at the 1.5 % of reads with annotations of the E. coli set, but a repeat-rich
// part (BASELINE config 3: a quarter of the reads have annotations) spent 0.36 of K2's 0.42 ms queueing on them
// (profiles/r4l_cfg3_* against r4n).  N_SHARD counters 128 bytes apart, the shard picked by the workgroup: shard s allocates
// annotation slots from ITS region [s R, (s + 1) R) of the buffer (R = anno_cap / N_SHARD), and files work items at positions
// s, s + N_SHARD, s + 2 N_SHARD ... of the work list, so that k_hinge_count still walks one array: slot w is in use iff
// w / N_SHARD < count[w % N_SHARD].  (The heavy list k_hinge_count appends to and k_hinge_call's cursor were sharded the same
// way and measured: no gain - on config 3 those kernels are at 5 TB/s of pile-up re-reads, not in the atomic queue - not kept.)
constexpr int N_SHARD = 64;
constexpr int SHARD_STRIDE = 32;      // unsigneds between two shard counters (128 bytes)
constexpr int ST_REDO_CAP = 64;       // one-sweep pass: the guard-band list is full (cannot happen: it has room for every read)

// Publication before a ticket (k_median_hist, k_spec_predict): everything a workgroup publishes before it takes its ticket is a
// device-scope ATOMIC (fire-and-forget adds / atomic stores, performed at the device's coherence point), so what the protocol needs
// is only that they are acknowledged first.  On gfx9 / CDNA non-returning global atomics count on vmcnt: `s_waitcnt vmcnt(0)`
// is that acknowledgement, and it is NOT an L2 write-back + invalidate like __threadfence() (~1 us per workgroup, serialised
// per XCD: 281 vs 32 us on a 2 048-workgroup sweep).  It is not a formal release: a PLAIN store published this
// way would need the fence - keep such data atomic.  Other targets (vscnt on gfx10+) get the fence.
#if defined(__gfx90a__) || defined(__gfx940__) || defined(__gfx941__) || defined(__gfx942__) || defined(__gfx950__)
#define HINGE_ATOMICS_ACKNOWLEDGED() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define HINGE_ATOMICS_ACKNOWLEDGED() __threadfence()
#endif

// ---- the one-sweep pass (round 4) -----------------------------------------------------------------------------------
// filter.cpp needs the whole part's median coverage (a GLOBAL BARRIER, filter.cpp:642-678) before it can cut a single mask:
// MIN_COV = max(MIN_COV, cov_est / 3).  Rounds 1-3 therefore swept every pile-up twice (k_cov_stats for the means, then K2).
// Now K2 runs FIRST, with a MIN_COV predicted from a sample of the part (k_spec_predict), and produces the exact per-read
// coverage sums as a by-product of its prefix scan.  It is exact for every MIN_COV in [pred - band, pred + band]:
//   * a read for which some MIN_COV-dependent predicate (a bin of the cutoff profile above MIN_COV, filter.cpp:703; an
//     annotation threshold (cov + MIN_COV) / F, filter.cpp:803) is NOT constant over that band emits nothing and goes on the
//     guard-band list (about 1 % of the reads of an E. coli 160x part at band 1);
//   * the exact median then runs as VERIFICATION (k_median_hist on the sums of this same sweep), and the listed reads are
//     run with the exact MIN_COV (k_mask_annotate in MODE_FINAL); if the exact value falls outside the band (not seen with
//     4096 sample reads) spec_state = 1 and that launch takes every read of the part instead, after the verification reset
//     the annotation allocator.
// Nothing is taken from an earlier pass over the same data: prediction, sweep and verification are one pass.
constexpr int MODE_CLASSIC = 0;   // MIN_COV is exact (k_cov_stats + median ran before)
constexpr int MODE_SPEC = 1;      // first sweep of a one-sweep pass: per-read sums out, guard band, deferral
constexpr int MODE_FINAL = 2;     // after the verification: the guard-band list (or everything) with the exact MIN_COV
struct SpecVerify {               // per part; spec_min_cov == nullptr: a classic pass, nothing to verify
    const int* spec_min_cov;      // MIN_COV the sweep ran with
    int band;
    int* spec_state;              // out: 0 = the exact MIN_COV lies inside the band, 1 = outside (everything is redone)
    unsigned* shards;             // annotation allocators and work-list lengths ([2][N_SHARD] counters, SHARD_STRIDE apart): reset when everything is redone
    unsigned* stats;              // cumulative: [0] passes verified, [1] exact != predicted, [2] outside the band
};
__device__ __forceinline__ void spec_verify(const SpecVerify& v, int exact) {
    if (!v.spec_min_cov) return;
    const int pred = *v.spec_min_cov;
    const int miss = (exact < pred - v.band || exact > pred + v.band) ? 1 : 0;
    *v.spec_state = miss;
    atomicAdd(&v.stats[0], 1u);
    if (exact != pred) atomicAdd(&v.stats[1], 1u);
    if (miss) {
        atomicAdd(&v.stats[2], 1u);
        for (int k = 0; k < 2 * N_SHARD; k++) v.shards[k * SHARD_STRIDE] = 0u;
    }
}

#ifdef HINGE_ABLATE
#define HINGE_ABLATE_POINT(k) if (P.ablate == (k)) continue;
#define HINGE_ABLATE_RETURN_V(k) if (P.ablate == (k) || ((k) == 2 && P.ablate >= 6)) return 0;
#else
#define HINGE_ABLATE_POINT(k)
#define HINGE_ABLATE_RETURN_V(k)
#endif

struct FilterDev {   // device copy of hinge_filter_params + derived values
    int reso, cut_off, theta;
    int cov_frac, min_ra, max_ra, ra_gap, nhr;
    int sup, pil, unb, tol, bin_len;
    int use_qv, use_cov, del_telo;
    int est_cov;
    int ablate;   // ablation builds only (-DHINGE_ABLATE): stop k_mask_annotate after phase k
};

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }
// ballot of a predicate as the compiler holds it (a lane mask in SGPRs): __ballot(int) materialises 0/1 per lane and compares again
__device__ __forceinline__ unsigned long long ballot_of(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// Per-read table look-ups with a 32-bit BYTE offset (tables of at most 4 GiB: read ids below 2^28).  A uniform offset becomes the
// scalar load's offset register and a per-lane one the store's 32-bit offset register, next to the table's base pointer - instead
// of a sign extension, a 64-bit shift and a 64-bit add on the scalar unit per access (it is the busiest unit of
// k_mask_annotate_q20: one per CU, 75 % occupied).
template <typename T> __device__ __forceinline__ T load_at32(const T* p, unsigned byte_off) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(p) + byte_off);
}
template <typename T> __device__ __forceinline__ void store_at32(T* p, unsigned byte_off, const T& v) {
    *reinterpret_cast<T*>(reinterpret_cast<char*>(p) + byte_off) = v;
}
__device__ __forceinline__ unsigned in_vgpr(unsigned x) {   // a uniform value the compiler must treat as per-lane: what is computed from it runs on the vector unit
    unsigned r;
    asm("v_mov_b32 %0, %1" : "=v"(r) : "s"(x));
    return r;
}

template <int RESO>
__device__ __forceinline__ int bin_of(int v, int reso) {
    // index of the first bin k with v < k*reso  (profileCoverage consumes events `< i*reso`)
    // reso is 40 in the reference (filter.cpp:386): constant division, branch-free ((v + R) / R is 0 on [-R, -1])
    if constexpr (RESO > 0) return (int)((unsigned)(max(v, -RESO) + RESO) / (unsigned)RESO);
    else return v < 0 ? 0 : v / reso + 1;
}

// ---- wavefront primitives on DPP (no LDS round trips) ------------------------------------------
// dpp_ctrl: row_shr:n = 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_or_old(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, false);
}

__device__ __forceinline__ int wave_incl_scan(int v) {   // inclusive +scan over the 64 lanes
    v += dpp_or_old<0x111, 0xf>(0, v);
    v += dpp_or_old<0x112, 0xf>(0, v);
    v += dpp_or_old<0x114, 0xf>(0, v);
    v += dpp_or_old<0x118, 0xf>(0, v);
    v += dpp_or_old<0x142, 0xa>(0, v);
    v += dpp_or_old<0x143, 0xc>(0, v);
    return v;
}
__device__ __forceinline__ int wave_incl_max_scan(int v) {
    v = max(v, dpp_or_old<0x111, 0xf>(INT_MIN, v));
    v = max(v, dpp_or_old<0x112, 0xf>(INT_MIN, v));
    v = max(v, dpp_or_old<0x114, 0xf>(INT_MIN, v));
    v = max(v, dpp_or_old<0x118, 0xf>(INT_MIN, v));
    v = max(v, dpp_or_old<0x142, 0xa>(INT_MIN, v));
    v = max(v, dpp_or_old<0x143, 0xc>(INT_MIN, v));
    return v;
}
__device__ __forceinline__ int wave_last(int v) { return __builtin_amdgcn_readlane(v, WAVE - 1); }
__device__ __forceinline__ int wave_sum(int v) { return wave_last(wave_incl_scan(v)); }
__device__ __forceinline__ int wave_max(int v) { return wave_last(wave_incl_max_scan(v)); }
__device__ __forceinline__ long long wave_sum64(long long v) {
    // 64-bit sum as two 32-bit scans with carry: low words summed as unsigned halves
    const unsigned lo = (unsigned)v;
    const int hi = (int)(v >> 32);
    const int s_lo16 = wave_sum((int)(lo & 0xffffu));
    const int s_hi16 = wave_sum((int)(lo >> 16));
    const int s_hi = wave_sum(hi);
    return ((long long)s_hi << 32) + ((long long)(unsigned)s_hi16 << 16) + (long long)(unsigned)s_lo16;
}
__device__ __forceinline__ long long wave_max64(long long v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { long long t = __shfl_xor(v, d); v = t > v ? t : v; }
    return v;
}
__device__ __forceinline__ int shfl_up1(int v, int fill) {   // lane l gets lane l-1, lane 0 gets fill
    int t = __builtin_amdgcn_update_dpp(fill, v, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
    return t;
}

template <int RESO>
__device__ __forceinline__ int nbins_of(int n_ovl, int max_ev, int reso) {
    // K of profileCoverage: 0 for an empty pile-up, else one bin past the one that consumes max_ev
    if (n_ovl == 0) return 0;
    return bin_of<RESO>(max_ev, reso) + 1;
}

// ------------------------------------------------------------------------------------------------
// K1: per-read cutoff-0 coverage sum and bin count without materialising the bins:
//     sum_k cov[k] = sum_o (bin_of(aepos) - bin_of(abpos)),  K = bin_of(max event) + 1.
// ------------------------------------------------------------------------------------------------
// PACKED: the spans come from the 16|16-bit copy (abpos | aepos << 16, written by k_pileup_facts): half the bytes of
// the int32 pairs, usable when every read of the part is shorter than 65536 bp and every coordinate lies in its read.
// K1 and the histogram phase of K2 sit at the HBM floor with int32 spans, so the bytes are what is left to cut.
template <bool PACKED> struct SpanLoad;
template <> struct SpanLoad<false> {
    typedef int2 raw;
    static __device__ __forceinline__ int2 get(raw v) { return v; }
};
template <> struct SpanLoad<true> {
    typedef unsigned raw;
    static __device__ __forceinline__ int2 get(raw v) { return make_int2((int)(v & 0xffffu), (int)(v >> 16)); }
};

template <int RESO, bool PACKED>
__global__ __launch_bounds__(BLOCK) void k_cov_stats(int r_begin, int r_end, const int64_t* __restrict__ row_ptr,
                                                     const int2* __restrict__ a_span, const unsigned* __restrict__ span16,
                                                     const int* __restrict__ rlen, int reso,
                                                     int* __restrict__ mean_cov, int* __restrict__ nbins0,
                                                     unsigned long long* __restrict__ wave_totals /*[2 * nwaves]*/,
                                                     int* __restrict__ pass_scalars, int n_pass_scalars, int* __restrict__ d_min_cov,
                                                     int set_min_cov, int min_cov_value) {
    // This is the first kernel of a pass and touches none of the pass scalars itself, so workgroup 0 clears
    // them (and applies a pending MIN_COV) instead of two 4-us memset launches in front of it.
    if (blockIdx.x == 0) {
        for (int t = threadIdx.x; t < n_pass_scalars; t += BLOCK) pass_scalars[t] = 0;
        if (threadIdx.x == 0 && set_min_cov) *d_min_cov = min_cov_value;
    }
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * BLOCK + threadIdx.x) >> 6);   // tell the compiler it is wave-uniform: row bounds become scalar loads
    const int nwaves = (gridDim.x * BLOCK) >> 6;
    long long blk_cov = 0, blk_slot = 0;
    // (no hand-written prefetch of the next row: measured 7x slower - it serialises the wave's loads)
    for (int i = r_begin + wave; i <= r_end; i += nwaves) {
        const int64_t s = row_ptr[i], e = row_ptr[i + 1];
        const int rl = rlen[i];
        long long tot;
        int mx = INT_MIN;
        bool q20_ok = false;     // every coordinate in [0, rl] and fewer than 65536 overlaps: k_mask_annotate_q20 may take the read
        if (e - s < 65536) {
            // common case: 32-bit lane offsets from the scalar row base, unconditional loads from a clamped index (no
            // exec-mask branch and no 64-bit address arithmetic per load), one 32-bit sum (n * K < 2^32 for n < 65536)
            const int n = (int)(e - s);
            typedef SpanLoad<PACKED> SL;
            const typename SL::raw* __restrict__ row = (PACKED ? (const typename SL::raw*)(const void*)span16 : (const typename SL::raw*)(const void*)a_span) + s;
            const unsigned last = n > 0 ? (unsigned)(n - 1) : 0u;
            unsigned sum = 0, umx = 0;   // max as unsigned: a negative coordinate shows up as a huge one
            for (int base = 0; base < n; base += LOADS_IN_FLIGHT * WAVE) {
                typename SL::raw v[LOADS_IN_FLIGHT];
#pragma unroll
                for (int u = 0; u < LOADS_IN_FLIGHT; u++) v[u] = row[min((unsigned)(base + u * WAVE + lane), last)];
#pragma unroll
                for (int u = 0; u < LOADS_IN_FLIGHT; u++) {
                    if (base + u * WAVE >= n) break;   // wave-uniform
                    if (base + u * WAVE + lane < n) {
                        const int2 w = SL::get(v[u]);
                        sum += (unsigned)(bin_of<RESO>(w.y, reso) - bin_of<RESO>(w.x, reso));
                        umx = max(umx, max((unsigned)w.x, (unsigned)w.y));
                    }
                }
            }
            const int cmx = wave_max((int)min(umx, 0x7fffffffu));
            if (n > 0) {
                mx = cmx;
                q20_ok = rl >= 0 && cmx <= rl;
                if (cmx == 0x7fffffff) {   // a negative (or absurd) coordinate: the signed maximum needs its own sweep
                    int m2 = INT_MIN;
                    for (int k = lane; k < n; k += WAVE) { const int2 w = SL::get(row[k]); m2 = max(m2, max(w.x, w.y)); }
                    mx = wave_max(m2);
                }
            } else {
                q20_ok = rl >= 0;
            }
            // |bin difference| <= bin_of(largest coordinate) when none is negative: the 32-bit (modular) sum is exact while n
            // times that stays below 2^31; otherwise (absurd coordinates) the row is summed again in 64 bits
            tot = (long long)(int)wave_sum((int)sum);
            if (cmx == 0x7fffffff || (long long)n * (long long)(bin_of<RESO>(mx, reso) + 1) >= (1LL << 31)) {
                long long s64 = 0;
                for (int k = lane; k < n; k += WAVE) { const int2 w = SL::get(row[k]); s64 += bin_of<RESO>(w.y, reso) - bin_of<RESO>(w.x, reso); }
                tot = wave_sum64(s64);
            }
        } else {
            int sum = 0;
            for (int64_t base = s; base < e; base += LOADS_IN_FLIGHT * WAVE) {   // all loads of a batch are issued before the first use
                int2 v[LOADS_IN_FLIGHT];
#pragma unroll
                for (int u = 0; u < LOADS_IN_FLIGHT; u++) {
                    const int64_t k = base + u * WAVE + lane;
                    v[u] = k < e ? a_span[k] : make_int2(0, 0);
                }
#pragma unroll
                for (int u = 0; u < LOADS_IN_FLIGHT; u++) {
                    if (base + u * WAVE + lane < e) {
                        sum += bin_of<RESO>(v[u].y, reso) - bin_of<RESO>(v[u].x, reso);
                        mx = max(mx, max(v[u].x, v[u].y));
                    }
                }
            }
            // per-lane partial sums fit 32 bits (<= 2^31 / 64 bins*overlaps per lane); widen for the total
            tot = wave_sum64((long long)sum);
            mx = wave_max(mx);
        }
        if (lane == 0) {
            const int K = nbins_of<RESO>((int)(e - s), mx, reso);
            // bins of the plain profile for k_mask_annotate_q20, -1 = not a read for that kernel (a coordinate outside the read, or
            // 65536+ overlaps: its 16-bit counts would overflow)
            nbins0[i] = (q20_ok && e - s < 65536) ? K : -1;
            if (rl >= 5000) {
                const long long m = tot / (long long)max(1, K);   // C division, filter.cpp:654
                mean_cov[i] = (int)m;
                blk_cov += tot;
                blk_slot += K;
            } else {
                mean_cov[i] = MEAN_SENTINEL;
            }
        }
    }
    // one slot per wave: thousands of atomics on one address cost ~12 ns each (they would dominate the kernel)
    if (lane == 0) {
        wave_totals[2 * wave] = (unsigned long long)blk_cov;
        wave_totals[2 * wave + 1] = (unsigned long long)blk_slot;
    }
}

// General median (any int32 values): one workgroup, 4-pass 8-bit radix select.  Run by the last block of
// k_median_hist when some mean coverage falls outside [0, MED_BINS).
template <typename VF>
__device__ void median_radix_select(VF value_of, int lo, int hi, int est_cov_override, int* __restrict__ est,
                                    int* __restrict__ min_cov, int* __restrict__ status, const SpecVerify& spec) {
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix, s_rank, s_nvalid;
    const int tid = threadIdx.x;
    if (tid == 0) s_nvalid = 0;
    __syncthreads();
    unsigned cnt = 0;
    for (int i = lo + tid; i <= hi; i += blockDim.x) cnt += (value_of(i) != MEAN_SENTINEL);
    atomicAdd(&s_nvalid, cnt);
    __syncthreads();
    const unsigned nvalid = s_nvalid;
    if (nvalid == 0) {
        if (tid == 0) { est[0] = 0; est[1] = 0; atomicOr(status, ST_NO_LONG_READ); spec_verify(spec, *min_cov); }
        return;
    }
    if (tid == 0) { s_prefix = 0; s_rank = nvalid / 2; }
    __syncthreads();
    for (int pass = 3; pass >= 0; --pass) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const unsigned prefix = s_prefix;
        const unsigned himask = pass == 3 ? 0u : (0xFFFFFFFFu << (8 * (pass + 1)));
        for (int i = lo + tid; i <= hi; i += blockDim.x) {
            const int v = value_of(i);
            if (v == MEAN_SENTINEL) continue;
            const unsigned u = (unsigned)v ^ 0x80000000u;   // order-preserving map to unsigned
            if ((u & himask) == prefix) atomicAdd(&hist[(u >> (8 * pass)) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned r = s_rank, d = 0;
            for (; d < 256; ++d) {
                if (r < hist[d]) break;
                r -= hist[d];
            }
            s_rank = r;
            s_prefix = prefix | (d << (8 * pass));
        }
        __syncthreads();
    }
    if (tid == 0) {
        int cov_est = (int)(s_prefix ^ 0x80000000u);
        est[0] = cov_est;
        est[1] = (int)nvalid;
        if (est_cov_override != 0) cov_est = est_cov_override;
        if (*min_cov < cov_est / 3) *min_cov = cov_est / 3;
        spec_verify(spec, *min_cov);
    }
}

// ------------------------------------------------------------------------------------------------
// Median, fast path: one multi-block pass.  Every block histograms its slice of mean_cov into LDS
// (values 0..MED_BINS-1) and merges the occupied range into one of MED_REPLICAS global histograms; the
// last block to finish sums the replicas, walks to the element of rank n/2, applies the MIN_COV update
// and clears the scratch for the next launch.  If a value lies outside the range the last block runs
// the radix select instead.
// Same-cache-line global atomics cost ~12 ns each and serialise (a first version with 169 blocks adding
// ~40 bins each to ONE histogram spent 27 us doing only that), hence few fat blocks, replicated
// histograms 16 KiB apart and one header word per 128-byte line.
// med layout: [MED_REPLICAS][MED_BINS] histograms, then one 64-bit word at MED_HDR: valid values (bits 0-39),
//   blocks done (40-51), blocks that saw an out-of-range value (52-63)
// ------------------------------------------------------------------------------------------------
constexpr int MED_REPLICAS = 8;
constexpr int MED_HDR = MED_REPLICAS * MED_BINS;
constexpr int MED_WORDS = MED_HDR + 32;
constexpr int MED_MAX_BLOCKS = 64;
constexpr int MED_BATCH_MAX = 16;

// One part's arguments; a launch takes up to MED_BATCH_MAX parts (round 3: the kernel is a 16-us chain of dependent round trips
// whatever the part's size, so the resident parts of a GPU go through it together - workgroup b works for part b % n).
struct MedianPart {
    const int* mean_cov; int lo, hi;
    unsigned* med; int* est; int* min_cov; int* status;
    const unsigned long long* wave_totals; int n_wave_totals; unsigned long long* totals;
    unsigned* hist_out;   // nullptr, or [MED_BINS + 2]: histogram, valid, out of range
    // one-sweep pass (cov_tot != nullptr): the means do not exist yet.  Read i has nbins0[i] bins and the coverage sum cov_tot[i]
    // (stored by k_mask_annotate_q20<SPEC>); nbins0[i] < 0: a read the general kernel took, which stored mean_out[i] itself and
    // accounted for its sums in wave_totals.  This kernel then also writes the means (mean_out = mean_cov).
    const int* cov_tot; const int* nbins0; const int* rlen; int* mean_out;
    SpecVerify spec;
};
// mean coverage of read i of a one-sweep pass (filter.cpp:642-656: C division of the sum by max(1, bins), reads >= 5000 bp only)
__device__ __forceinline__ int fused_mean(int K, int tot, int rl, int stored) {
    if (K < 0) return stored;
    if (rl < 5000) return MEAN_SENTINEL;
    return tot / max(1, K);
}
struct MedianHistBatch {
    MedianPart part[MED_BATCH_MAX];
    int n;
};
__global__ __launch_bounds__(256) void k_median_hist(MedianHistBatch B, int est_cov_override) {
    const unsigned n_parts = (unsigned)B.n;
    const MedianPart& A = B.part[blockIdx.x % n_parts];
    const unsigned bx = blockIdx.x / n_parts, gx = gridDim.x / n_parts;
    if (bx >= gx) return;
    const int* __restrict__ mean_cov = A.mean_cov; const int lo = A.lo, hi = A.hi;
    unsigned* __restrict__ med = A.med; int* __restrict__ est = A.est; int* __restrict__ min_cov = A.min_cov; int* __restrict__ status = A.status;
    const unsigned long long* __restrict__ wave_totals = A.wave_totals; const int n_wave_totals = A.n_wave_totals;
    unsigned long long* __restrict__ totals = A.totals; unsigned* __restrict__ hist_out = A.hist_out;
    const int* __restrict__ cov_tot = A.cov_tot; const int* __restrict__ nb0 = A.nbins0; const int* __restrict__ rlen = A.rlen;
    const bool fused = cov_tot != nullptr;
    __shared__ unsigned hist[MED_BINS];
    __shared__ unsigned s_valid, s_oor, s_last, s_lo, s_hi, s_general;
    __shared__ unsigned long long s_tc, s_ts, s_ticket;
    const int tid = threadIdx.x;
#ifdef HINGE_TIMING
    const unsigned long long tq0 = wall_clock64();
#endif
    for (int b = tid; b < MED_BINS; b += blockDim.x) hist[b] = 0;
    if (tid == 0) { s_valid = 0; s_oor = 0; s_last = 0; s_lo = MED_BINS - 1; s_hi = 0; s_tc = 0; s_ts = 0; s_general = 0; }
    __syncthreads();
    // mean coverages cluster in a few dozen values: only the occupied range [vlo, vhi] is merged and read back
    unsigned nv = 0, oor = 0, vlo = MED_BINS - 1, vhi = 0;
    unsigned long long tc = 0, ts = 0;
    {
        constexpr int U = 8;   // independent loads in flight per thread: one memory round trip per 8 values
        const int stride = (int)(gx * blockDim.x);
        // total_cov / num_slot of the part (only logged by the reference, filter.cpp:666,672): every block sums a slice of
        // k_cov_stats' per-wave partials; loaded together with the first batch of values
        for (int w = (int)(bx * blockDim.x) + tid; w < n_wave_totals; w += stride) { tc += wave_totals[2 * w]; ts += wave_totals[2 * w + 1]; }
        for (int i0 = lo + (int)(bx * blockDim.x) + tid; i0 <= hi; i0 += U * stride) {
            int vv[U];
            if (!fused) {
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const long long i = (long long)i0 + (long long)u * stride;
                    vv[u] = i <= hi ? mean_cov[i] : MEAN_SENTINEL;
                }
            } else {
                int kk[U], tt[U], rr[U];
#pragma unroll
                for (int u = 0; u < U; u++) {   // (all loads of the batch in flight together)
                    const long long i = min((long long)i0 + (long long)u * stride, (long long)hi);
                    kk[u] = nb0[i]; tt[u] = cov_tot[i]; rr[u] = rlen[i]; vv[u] = mean_cov[i];
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const long long i = (long long)i0 + (long long)u * stride;
                    if (i > hi) { vv[u] = MEAN_SENTINEL; continue; }
                    const int v = fused_mean(kk[u], tt[u], rr[u], vv[u]);
                    if (kk[u] >= 0) {
                        A.mean_out[i] = v;
                        if (v != MEAN_SENTINEL) { tc += (unsigned long long)(long long)tt[u]; ts += (unsigned long long)kk[u]; }
                    }
                    vv[u] = v;
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int v = vv[u];
                if (v == MEAN_SENTINEL) continue;
                nv++;
                if (v >= 0 && v < MED_BINS) {
                    atomicAdd(&hist[v], 1u);
                    vlo = min(vlo, (unsigned)v);
                    vhi = max(vhi, (unsigned)v);
                } else oor = 1;
            }
        }
    }
    {   // one LDS atomic per wavefront and quantity, not one per thread
        const int lane = lane_id();
        nv = (unsigned)wave_sum((int)nv);
        vlo = (unsigned)(MED_BINS - 1) - (unsigned)wave_max((int)((unsigned)(MED_BINS - 1) - vlo));
        vhi = (unsigned)wave_max((int)vhi);
        oor = __ballot(oor != 0) != 0ull;
        tc = (unsigned long long)wave_sum64((long long)tc);
        ts = (unsigned long long)wave_sum64((long long)ts);
        if (lane == 0) {
            if (nv) { atomicAdd(&s_valid, nv); atomicMin(&s_lo, vlo); atomicMax(&s_hi, vhi); }
            if (oor) atomicOr(&s_oor, 1u);
            if (tc) atomicAdd(&s_tc, tc);
            if (ts) atomicAdd(&s_ts, ts);
        }
    }
    __syncthreads();
#ifdef HINGE_TIMING
    const unsigned long long tq1 = wall_clock64();
#endif
    unsigned* my_hist = med + (size_t)(bx % MED_REPLICAS) * MED_BINS;
    for (unsigned b = s_lo + tid; b <= s_hi; b += blockDim.x)
        if (hist[b]) atomicAdd(&my_hist[b], hist[b]);
    switch (tid) {   // fire-and-forget
        case 4: if (s_tc) atomicAdd(&totals[0], s_tc); break;
        case 5: if (s_ts) atomicAdd(&totals[1], s_ts); break;
        default: break;
    }
    // What this block publishes went out as device-scope atomics, which are performed at the device's coherence point: waiting
    // for their acknowledgement orders them before the ticket.  NOT __threadfence(): a release fence at device scope writes
    // back and invalidates the XCD's whole L2 (~1 us, serialised per XCD): 64 blocks doing that cost this kernel 2.5 of its
    // 19 us, and 2 048 workgroups doing it inside k_cov_stats (a fused variant, round 3) took that sweep from 32 to 281 us.
    HINGE_ATOMICS_ACKNOWLEDGED();
    __syncthreads();
    if (tid == 0) {   // ONE returning atomic publishes this block and tells the last one everything it needs:
                      // bits 0-39 valid values, 40-51 blocks done, 52-63 blocks that saw an out-of-range value
        const unsigned long long mine = (unsigned long long)s_valid + (1ull << 40) + (s_oor ? (1ull << 52) : 0ull);
        const unsigned long long t = atomicAdd(reinterpret_cast<unsigned long long*>(&med[MED_HDR]), mine);
        s_ticket = t + mine;
        s_last = (((t >> 40) & 0xfffull) == gx - 1);
    }
    __syncthreads();
#ifdef HINGE_TIMING
    const unsigned long long tq2 = wall_clock64();
#endif
    if (!s_last) return;
    __threadfence();   // acquire: the loads below must not be served from a stale vector L1 line
    // last block: sum the replicas, 16 consecutive bins per thread, all loads independent (one memory round trip),
    // and leave the scratch clean for the next launch
    {
        int acc[16];
#pragma unroll
        for (int k = 0; k < 16; k++) acc[k] = 0;
#pragma unroll
        for (int r = 0; r < MED_REPLICAS; r++) {
            int4* src = reinterpret_cast<int4*>(med + (size_t)r * MED_BINS + (size_t)tid * 16);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int4 v = src[q];
                acc[4 * q] += v.x; acc[4 * q + 1] += v.y; acc[4 * q + 2] += v.z; acc[4 * q + 3] += v.w;
                src[q] = make_int4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int k = 0; k < 16; k++) hist[tid * 16 + k] = (unsigned)acc[k];
    }
    if (tid == 0) { med[MED_HDR] = 0; med[MED_HDR + 1] = 0; }
    if (hist_out) {   // sharded runs: the block's histogram goes out to be summed over ranks (k_median_from_hist finishes)
        __syncthreads();
        for (int b = tid; b < MED_BINS; b += blockDim.x) hist_out[b] = hist[b];
        if (tid == 0) {
            hist_out[MED_BINS] = (unsigned)(s_ticket & ((1ull << 40) - 1ull));   // < 2^32 values per rank
            hist_out[MED_BINS + 1] = (unsigned)(s_ticket >> 52);
        }
        return;
    }
    const unsigned glo = 0, ghi = MED_BINS - 1;
    if (tid == 0) {
        const unsigned nvalid = (unsigned)(s_ticket & ((1ull << 40) - 1ull));
        const unsigned bad = (unsigned)(s_ticket >> 52);
        if (nvalid == 0) {
            est[0] = 0; est[1] = 0;
            atomicOr(status, ST_NO_LONG_READ);
            spec_verify(A.spec, *min_cov);
        } else if (bad) {
            s_general = 1;
        } else {
            s_general = 2;   // in range: wavefront 0 walks the histogram below
        }
    }
    __syncthreads();
    if (s_general == 2 && tid < WAVE) {
        // element of rank n/2 (median_id = size/2, filter.cpp:660): first bin whose inclusive prefix exceeds it
        const unsigned nvalid = (unsigned)(s_ticket & ((1ull << 40) - 1ull));
        const int r = (int)(nvalid / 2);
        int carry = 0, found = -1;
        for (unsigned base = glo; base <= ghi && found < 0; base += WAVE) {
            const unsigned b = base + tid;
            const int incl = wave_incl_scan(b <= ghi ? (int)hist[b] : 0) + carry;
            const unsigned long long hit = __ballot(incl > r);
            if (hit) found = (int)base + __ffsll((long long)hit) - 1;
            carry = wave_last(incl);
        }
        if (tid == 0) {
            int cov_est = found;
            est[0] = cov_est;
            est[1] = (int)nvalid;
            if (est_cov_override != 0) cov_est = est_cov_override;   // filter.cpp:671
            const int before = atomicMax(min_cov, cov_est / 3);       // filter.cpp:677-678: if (MIN_COV < cov_est/3) MIN_COV = cov_est/3
            spec_verify(A.spec, max(before, cov_est / 3));
        }
    }
#ifdef HINGE_TIMING
#endif
    __syncthreads();
    if (s_general == 1) {
        // (the means other workgroups stored in this launch are not visible here: a one-sweep pass derives them again)
        if (fused) median_radix_select([&](int i) { return fused_mean(nb0[i], cov_tot[i], rlen[i], mean_cov[i]); }, lo, hi, est_cov_override, est, min_cov, status, A.spec);
        else median_radix_select([&](int i) { return mean_cov[i]; }, lo, hi, est_cov_override, est, min_cov, status, A.spec);
    }
}

// Median from a histogram that was summed over ranks (hist[MED_BINS + 2]: bins, valid values, ranks that saw a value
// outside [0, MED_BINS)).  One workgroup; same walk and MIN_COV update as the tail of k_median_hist.
__global__ __launch_bounds__(256) void k_median_from_hist(const unsigned* __restrict__ hist_in, int est_cov_override, int* __restrict__ est,
                                                          int* __restrict__ min_cov, int* __restrict__ status, SpecVerify spec) {
    __shared__ unsigned hist[MED_BINS];
    const int tid = threadIdx.x;
    for (int b = tid; b < MED_BINS; b += blockDim.x) hist[b] = hist_in[b];
    const unsigned nvalid = hist_in[MED_BINS], bad = hist_in[MED_BINS + 1];
    __syncthreads();
    if (nvalid == 0) {
        if (tid == 0) { est[0] = 0; est[1] = 0; atomicOr(status, ST_NO_LONG_READ); spec_verify(spec, *min_cov); }
        return;
    }
    if (bad) {   // needs the values themselves: the caller has to all-gather the means and use k_median_hist
        if (tid == 0) atomicOr(status, ST_MEDIAN_RANGE);
        return;
    }
    if (tid < WAVE) {
        const int r = (int)(nvalid / 2);
        int carry = 0, found = -1;
        for (int base = 0; base < MED_BINS && found < 0; base += WAVE) {
            const int incl = wave_incl_scan((int)hist[base + tid]) + carry;
            const unsigned long long hit = __ballot(incl > r);
            if (hit) found = base + __ffsll((long long)hit) - 1;
            carry = wave_last(incl);
        }
        if (tid == 0) {
            int cov_est = found;
            est[0] = cov_est;
            est[1] = (int)nvalid;
            if (est_cov_override != 0) cov_est = est_cov_override;   // filter.cpp:671
            const int before = atomicMax(min_cov, cov_est / 3);       // filter.cpp:677-678
            spec_verify(spec, max(before, cov_est / 3));
        }
    }
}

// The same for several parts at once (a rank's resident parts after ONE all-reduce over all their histograms): block b
// finishes the median of part b.  One launch instead of one per part.
struct MedianBatch {
    int* est[MED_BATCH_MAX];
    int* min_cov[MED_BATCH_MAX];
    int* status[MED_BATCH_MAX];
    SpecVerify spec[MED_BATCH_MAX];
};
__global__ __launch_bounds__(256) void k_median_from_hist_batch(const unsigned* __restrict__ hist_in, long long row_stride, int est_cov_override,
                                                                MedianBatch B) {
    __shared__ unsigned hist[MED_BINS];
    const int tid = threadIdx.x, b = blockIdx.x;
    const unsigned* __restrict__ h = hist_in + (long long)b * row_stride;
    for (int k = tid; k < MED_BINS; k += blockDim.x) hist[k] = h[k];
    const unsigned nvalid = h[MED_BINS], bad = h[MED_BINS + 1];
    __syncthreads();
    if (nvalid == 0) {
        if (tid == 0) { B.est[b][0] = 0; B.est[b][1] = 0; atomicOr(B.status[b], ST_NO_LONG_READ); spec_verify(B.spec[b], *B.min_cov[b]); }
        return;
    }
    if (bad) {
        if (tid == 0) atomicOr(B.status[b], ST_MEDIAN_RANGE);
        return;
    }
    if (tid < WAVE) {
        const int r = (int)(nvalid / 2);
        int carry = 0, found = -1;
        for (int base = 0; base < MED_BINS && found < 0; base += WAVE) {
            const int incl = wave_incl_scan((int)hist[base + tid]) + carry;
            const unsigned long long hit = __ballot(incl > r);
            if (hit) found = base + __ffsll((long long)hit) - 1;
            carry = wave_last(incl);
        }
        if (tid == 0) {
            int cov_est = found;
            B.est[b][0] = cov_est;
            B.est[b][1] = (int)nvalid;
            if (est_cov_override != 0) cov_est = est_cov_override;   // filter.cpp:671
            const int before = atomicMax(B.min_cov[b], cov_est / 3);  // filter.cpp:677-678
            spec_verify(B.spec[b], max(before, cov_est / 3));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// One-sweep pass, step 1: the MIN_COV the sweep will run with.  ns reads of the part, evenly spaced over its id range, get
// their exact mean coverage (the closed form of k_cov_stats over their pile-ups: 1/20 of the part's spans at 4096 of 87 k
// reads); the sample's median stands in for the part's (filter.cpp:660-678).  Two reads per wavefront (32 lanes each, all of a
// read's loads in flight together), 32 reads per 1024-thread workgroup; the means go out as device-scope atomic stores, the
// workgroups count themselves on a ticket and the last one histograms the sample and walks to its median - the same
// publication pattern as k_median_hist.  Workgroup 0 of a part also clears the pass scalars and applies a pending MIN_COV
// (what k_cov_stats did for the two-sweep pass).  Up to MED_BATCH_MAX parts per launch (workgroup b works for part b % n).
// A prediction only: whatever it says, the verification after the sweep decides (SpecVerify); `bias` (tests) shifts it.
// ------------------------------------------------------------------------------------------------
struct SpecPart {
    int r_begin, r_end;
    const int64_t* row_ptr; const int2* a_span; const unsigned* span16; const int* rlen; const int* nbins0;
    int* pass_scalars; int n_pass_scalars;
    int* min_cov; int set_min_cov, min_cov_value;
    int* spec_min_cov;
    int* sample;                 // [ns] scratch
    unsigned* ticket;            // one word, zero between launches
    int bias;
};
struct SpecBatch {
    SpecPart part[MED_BATCH_MAX];
    int n, ns;
};
constexpr int SPEC_BLOCK = 1024;
constexpr int SPEC_READS_PER_BLOCK = SPEC_BLOCK / 32;
template <int RESO, bool PACKED>
__global__ __launch_bounds__(SPEC_BLOCK) void k_spec_predict(SpecBatch B, int reso, int est_cov_override) {
    const unsigned n_parts = (unsigned)B.n;
    const SpecPart& A = B.part[blockIdx.x % n_parts];
    const unsigned bx = blockIdx.x / n_parts, gx = gridDim.x / n_parts;
    if (bx >= gx) return;
    const int tid = threadIdx.x;
    const int cur_min_cov = A.set_min_cov ? A.min_cov_value : *A.min_cov;   // (nobody writes *min_cov in this launch unless set_min_cov)
    if (bx == 0) {
        for (int t = tid; t < A.n_pass_scalars; t += SPEC_BLOCK) A.pass_scalars[t] = 0;
        if (tid == 0 && A.set_min_cov) *A.min_cov = A.min_cov_value;
    }
    if (est_cov_override != 0) {   // `ec` in the ini (filter.cpp:671): MIN_COV does not depend on the data
        if (bx == 0 && tid == 0) *A.spec_min_cov = max(cur_min_cov, est_cov_override / 3) + A.bias;
        return;
    }
    const long long nr = (long long)A.r_end - A.r_begin + 1;
    const int ns = (int)min((long long)B.ns, max(nr, 0ll));
    __shared__ unsigned hist[MED_BINS];
    __shared__ unsigned s_last, s_n;
    const int lane = tid & 63, half = lane >> 5, l32 = lane & 31;
    const int wib = __builtin_amdgcn_readfirstlane(tid >> 6);
    {
        const int k = (int)bx * SPEC_READS_PER_BLOCK + wib * 2 + half;   // sample index of this half wavefront
        int mean = MEAN_SENTINEL;
        if (k < ns) {
            const int i = A.r_begin + (int)(((long long)k * nr) / ns);
            const int64_t s = A.row_ptr[i], e = A.row_ptr[i + 1];
            const int rl = A.rlen[i], K = A.nbins0[i];
            typedef SpanLoad<PACKED> SL;
            const typename SL::raw* __restrict__ row = (PACKED ? (const typename SL::raw*)(const void*)A.span16 : (const typename SL::raw*)(const void*)A.a_span) + s;
            const long long n = e - s;
            long long sum = 0;
            if (rl >= 5000 && K >= 0) {   // (K < 0: 65536+ overlaps or a coordinate outside the read - not sampled)
                for (long long base = 0; base < n; base += LOADS_IN_FLIGHT * 32) {
                    typename SL::raw v[LOADS_IN_FLIGHT];
#pragma unroll
                    for (int u = 0; u < LOADS_IN_FLIGHT; u++) v[u] = row[min(base + u * 32 + l32, n - 1)];
#pragma unroll
                    for (int u = 0; u < LOADS_IN_FLIGHT; u++)
                        if (base + u * 32 + l32 < n) { const int2 w = SL::get(v[u]); sum += bin_of<RESO>(w.y, reso) - bin_of<RESO>(w.x, reso); }
                }
            }
            // the two halves of the wavefront are summed apart: xor butterflies stay inside 32 lanes
#pragma unroll
            for (int d = 16; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
            if (rl >= 5000 && K >= 0) mean = (int)(sum / (long long)max(1, K));
            if (l32 == 0) __hip_atomic_store(&A.sample[k], mean, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    HINGE_ATOMICS_ACKNOWLEDGED();   // (see k_median_hist: the stores above are performed at device scope)
    __syncthreads();
    if (tid == 0) s_last = (atomicAdd(A.ticket, 1u) == gx - 1) ? 1u : 0u;
    for (int b = tid; b < MED_BINS; b += SPEC_BLOCK) hist[b] = 0;
    if (tid == 0) s_n = 0;
    __syncthreads();
    if (!s_last) return;
    unsigned cnt = 0;
    for (int k = tid; k < ns; k += SPEC_BLOCK) {
        const int v = __hip_atomic_load(&A.sample[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v == MEAN_SENTINEL) continue;
        atomicAdd(&hist[min(max(v, 0), MED_BINS - 1)], 1u);
        cnt++;
    }
    if (cnt) atomicAdd(&s_n, cnt);
    __syncthreads();
    if (tid < WAVE) {
        const unsigned nvalid = s_n;
        int found = 0;
        if (nvalid) {
            const int r = (int)(nvalid / 2);
            int carry = 0;
            found = -1;
            for (int base = 0; base < MED_BINS && found < 0; base += WAVE) {
                const int incl = wave_incl_scan((int)hist[base + tid]) + carry;
                const unsigned long long hit = __ballot(incl > r);
                if (hit) found = base + __ffsll((long long)hit) - 1;
                carry = wave_last(incl);
            }
        }
        if (tid == 0) {
            *A.spec_min_cov = max(cur_min_cov, found / 3) + A.bias;
            *A.ticket = 0u;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Facts about a part's pile-ups that stay true for every pass over it (run once by hinge_set_pileups):
// facts[0] = largest pile-up, facts[1] = 1 if some coordinate lies outside [0, rlen].  With them the host
// knows when the hand-back launch of K2 and the serial exact-path kernel of K3 cannot have work.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_pileup_facts(int r_begin, int r_end, const int64_t* __restrict__ row_ptr,
                                                        const int2* __restrict__ a_span, const int* __restrict__ rlen,
                                                        unsigned* __restrict__ facts, unsigned* __restrict__ span16 /*nullptr or [n_ovl]*/) {
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * BLOCK + threadIdx.x) >> 6);
    const int nwaves = (gridDim.x * BLOCK) >> 6;
    unsigned max_pile = 0, bad = 0;
    for (int i = r_begin + wave; i <= r_end; i += nwaves) {
        const int64_t s = row_ptr[i], e = row_ptr[i + 1];
        const unsigned rl = (unsigned)max(rlen[i], 0);
        max_pile = max(max_pile, (unsigned)min<int64_t>(e - s, 0x7fffffff));
        for (int64_t k = s + lane; k < e; k += WAVE) {
            const int2 v = a_span[k];
            bad |= ((unsigned)v.x > rl) || ((unsigned)v.y > rl);   // unsigned: negative coordinates are "too large"
            if (span16) span16[k] = ((unsigned)v.x & 0xffffu) | ((unsigned)v.y << 16);   // only used if the facts allow it
        }
    }
    if (__ballot(bad != 0) && lane == 0) atomicOr(&facts[1], 1u);
    if (lane == 0 && max_pile) atomicMax(&facts[0], max_pile);
}

// ------------------------------------------------------------------------------------------------
// K2: coverage mask + repeat annotation + gate.  Two kernels share everything after the binning:
//   k_mask_annotate_q20  the shipped configuration (reso 40, cut_off a multiple of 20): ONE 20-bp
//                        begin|end histogram per read from which both 40-bp profiles derive
//   k_mask_annotate      any reso / cut_off / pile-up size: two 40-bp difference histograms; also runs
//                        the reads the fast kernel hands back (fallback list)
// ------------------------------------------------------------------------------------------------
// A read that goes on to hinge calling, with everything k_hinge_count needs to start streaming its pile-up after ONE
// dependent load (instead of work list -> row_ptr / mask / anno_off / anno_cnt -> anno_buf -> spans).
struct alignas(16) WorkItem {
    int read, n;          // read id, pile-up size
    long long row;        // row_ptr[read]
    int mask_lo, mask_hi; // its own mask
    unsigned off;         // anno_off[read]
    int cnt;              // anno_cnt[read]
    int2 anno[4];         // the first four annotations (pos, type); further ones are read from anno_buf
};

struct AnnoOut {   // per-part outputs of K2
    const int2* qv_mask;
    const unsigned char* keep;   // --restrictreads (filter.cpp:680-694,767-773): nullptr, or 0 for reads whose masks are emptied
    int2* mask;
    int2* cmask;
    unsigned char* rflags;
    int2* anno_buf;
    unsigned char* hinge_flag;
    unsigned* anno_off;
    int* anno_cnt;
    unsigned* anno_shard; // N_SHARD annotation allocators (SHARD_STRIDE apart): shard s owns slots [s * anno_region, (s + 1) * anno_region)
    unsigned* work_shard; // N_SHARD work-list lengths: shard s files its items at s, s + N_SHARD, ...
    unsigned anno_region; // anno_cap / N_SHARD
    unsigned work_cap;    // slots of work_list
    unsigned anno_cap;
    WorkItem* work_list;
    int* status;
    // optional: the cutoff-0 coverage bins themselves (the .coverage.txt payload, filter.cpp:599-602), written while they are
    // in LDS anyway: bins of read i at cov_out[cov_off[i - cov_base] ..], their number at cov_nbins[i - cov_base]
    int* cov_out;
    const long long* cov_off;
    int* cov_nbins;
    int cov_base;
};

// Longest run of bins with coverage > MIN_COV (filter.cpp:696-728), fed 64 bins at a time as a ballot.
// All of it is scalar: runs are closed by the set bits of C below (a handful per read).
struct RunState {
    int last_np;                   // index of the last bin with c <= MIN_COV seen so far (0 before any: start = 0)
    unsigned long long prev_pos;   // was the previous bin positive?
    int best_len, best_j;          // first longest run wins (strict > in ascending j)
};
__device__ __forceinline__ void run_feed(RunState& r, int base, unsigned long long M /*positive bins*/, unsigned long long V /*valid bins*/,
                                         int reso) {
    const unsigned long long N = ~M & V;
    unsigned long long C = N & ((M << 1) | r.prev_pos);   // non-positive bins that close a positive run
    while (C) {
        const int jj = __ffsll((long long)C) - 1;
        C &= C - 1ull;
        const unsigned long long below = N & ((1ull << jj) - 1ull);
        const int zb = below ? base + 63 - __clzll((long long)below) : r.last_np;   // last non-positive bin before the run
        const int len = reso * (base + jj - 1) - reso * zb - reso;
        if (len > r.best_len) { r.best_len = len; r.best_j = base + jj; }
    }
    if (N) r.last_np = base + 63 - __clzll((long long)N);
    r.prev_pos = M >> 63;
}

// Everything after the coverage profiles exist: mask, telomere flag, gate sums, annotation candidates, merge,
// outputs.  z(j) = cutoff-0 coverage of bin j (j < K0), c(j) = cutoff coverage; cand = LDS scratch for the
// packed candidates (pos << 1 | (type == +1)): slot t is written only after z(j) was read for every j <= t.
// PT / OT: FilterDev / AnnoOut, possibly qualified with the constant address space (k_mask_annotate_q20 passes them in device
// memory: every field is then a scalar load at its point of use instead of an SGPR that is live - or spilled - across the read loop).
constexpr int SPEC_DEFERRED = 1 << 30;   // mask_gate_annotate's return value: the read emitted nothing and waits for the exact MIN_COV
template <typename PT, typename OT, typename ZF, typename CF>
__device__ __forceinline__ int mask_gate_annotate(const PT& P, const int reso, const int MIN_COV, const int i, const int lane,
                                                   const int K0, const RunState& run, ZF z, CF c, int* cand, const OT& o,
                                                   const long long row, const int n_pile, const bool cov_done = false,
                                                   const bool cand_in_profile = true /*cand[] overwrites what z() reads*/,
                                                   const unsigned long long flag_words = ~0ull /*bit w clear: no bin of [64 w, 64 w + 63] can be an annotation*/,
                                                   const int band = 0 /*MODE_SPEC: MIN_COV is only known to lie in [MIN_COV - band, MIN_COV + band]*/) {
    // What every read needs of the parameters and output pointers, looked up TOGETHER: where P and o are in device memory
    // (k_mask_annotate_q20) each look-up at its point of use is a scalar-load round trip of its own in the read's dependency
    // chain - six of them, one behind the other, before this.
    const int del_telo = P.del_telo, use_qv = P.use_qv, use_cov = P.use_cov;
    const int2* const qv_maskp = o.qv_mask;
    const unsigned char* const keepp = o.keep;
    int2* const maskp = o.mask;
    int2* const cmaskp = o.cmask;
    unsigned char* const rflagsp = o.rflags;
    unsigned* const anno_offp = o.anno_off;
    int* const anno_cntp = o.anno_cnt;
    asm volatile("" :: "s"(del_telo), "s"(use_qv), "s"(use_cov), "s"(qv_maskp), "s"(keepp), "s"(maskp), "s"(cmaskp), "s"(rflagsp), "s"(anno_offp), "s"(anno_cntp));
    if (o.cov_out && !cov_done) {   // before anything reuses the profile's LDS (cand)
        int* __restrict__ dst = o.cov_out + o.cov_off[i - o.cov_base];
        for (int j = lane; j < K0; j += WAVE) dst[j] = z(j);
        if (lane == 0) o.cov_nbins[i - o.cov_base] = K0;
    }
    int maxstart = 0, maxend = 0, msc = 0, mec = 0;
    if (run.best_len > 0) {
        mec = run.best_j - 1;
        maxend = reso * mec;
        maxstart = maxend - run.best_len;   // = reso*z + reso
        msc = maxstart / reso;              // = z + 1
    }
    unsigned char fl = 0;
    if (del_telo && lane == 0) {   // filter.cpp:731-760 on cutoff_cov + MIN_COV = max(cov, MIN_COV)
        int sc = 0, ec = 0;
        if (mec - msc + 1 > 20) {
            for (int d = 0; d < 10; d++) { sc += max(c(msc + d), MIN_COV); ec += max(c(mec - d), MIN_COV); }
            sc /= 10; ec /= 10;
        } else {
            const int limit = (mec - msc) / 2;
            for (int d = 0; d < limit; d++) { sc += max(c(msc + d), MIN_COV); ec += max(c(mec - d), MIN_COV); }
            if (limit == 0) { sc = 0; ec = 0; } else { sc /= limit; ec /= limit; }
        }
        if ((sc >= 10 * ec) || (ec >= 10 * sc)) fl |= 1;
    }
    const unsigned iv = in_vgpr((unsigned)i);   // the read's index for the lane-0 stores below (vector-side address arithmetic)
    int2 mk;
    {
        int2 q = qv_maskp ? load_at32(qv_maskp, (unsigned)i << 3) : make_int2(0, 0);
        if (keepp && !load_at32(keepp, (unsigned)i)) { maxend = maxstart; q.y = q.x; }   // filter.cpp:767-773
        if (use_qv && use_cov) mk = make_int2(max(maxstart, q.x), min(maxend, q.y));
        else if (use_cov && !use_qv) mk = make_int2(maxstart, maxend);
        else mk = q;
    }
    if (lane == 0) {
        store_at32(maskp, iv << 3, mk);
        store_at32(cmaskp, iv << 3, make_int2(msc, mec));
        store_at32(rflagsp, iv, fl);
    }
    HINGE_ABLATE_RETURN_V(2)
    // ---- gate sums over the two NO_HINGE_REGION windows only (filter.cpp:842-865) -----------------
    // Only a read that keeps an annotation after the merge needs them (about 3 % of the reads): they are taken lazily, after the
    // merge, unless the candidate list shares its LDS with the profile (then the windows may be overwritten by it: taken first).
    int ncand = 0;
    int S = 0, nS = 0, E = 0, nE = 0;
    auto gate_sums = [&]() {
        // bins j with lo <= reso*j <= hi, clipped to [0, K0)
        auto jfirst = [&](int lo) { return lo <= 0 ? 0 : (lo + reso - 1) / reso; };
        auto jlast = [&](int hi) { return hi < 0 ? -1 : min(hi / reso, K0 - 1); };
        const int s0 = jfirst(mk.x), s1 = jlast(mk.x + P.nhr);
        const int e0 = jfirst(mk.y - P.nhr), e1 = jlast(mk.y);
        nS = max(s1 - s0 + 1, 0);
        nE = max(e1 - e0 + 1, 0);
        if (nS <= 32 && nE <= 32) {
            // both windows are a dozen bins: lanes 0-31 take the start window, lanes 32-63 the end window, ONE scan gives both
            // sums (integer adds: the order of summation is immaterial)
            const int half = lane & 31;
            const int j = lane < 32 ? s0 + half : e0 + half;
            const bool in = lane < 32 ? half < nS : half < nE;
            const int incl = wave_incl_scan(in ? z(j) : 0);
            S = __builtin_amdgcn_readlane(incl, 31);
            E = wave_last(incl) - S;
        } else {
            for (int j = s0 + lane; j <= s1; j += WAVE) S += z(j);
            for (int j = e0 + lane; j <= e1; j += WAVE) E += z(j);
            S = wave_sum(S); E = wave_sum(E);
        }
    };
    if (cand_in_profile && flag_words != 0ull) gate_sums();   // (no flagged word: no candidate, no gate)
    HINGE_ABLATE_RETURN_V(3)
    // annotation window in bins: reso*j in [mk.x + nhr, mk.y - nhr], j < K0 - 2
    // (nothing of this - window bounds, parameter loads - when no word can hold an annotation: the usual case in k_mask_annotate_q20)
    if (flag_words != 0ull) {
        const int wlo = mk.x + P.nhr, whi = mk.y - P.nhr;
        int jlo = wlo <= 0 ? 0 : (wlo + reso - 1) / reso;
        int jhi = whi < 0 ? -1 : whi / reso;
        jhi = min(jhi, K0 - 3);
        // |g| > min(max(x / F, lo), hi) with x = c + MIN_COV  <=>  |g| > hi  ||  (|g| > lo && |g| > x / F), and for
        // x >= 0, F > 0:  |g| > x / F  <=>  |g| * F > x  -- no division on the common path
        const bool mulpath = P.cov_frac > 0 && P.cov_frac < 8192 && P.min_ra >= 0 && P.max_ra >= 0;
        for (int base = (jlo / WAVE) * WAVE; base <= jhi; base += WAVE) {
            if (base < 64 * WAVE && !((flag_words >> (base / WAVE)) & 1ull)) continue;   // (words beyond the 64th: always looked at)
            const int j = base + lane;
            int code = -1;
            const bool in = j >= jlo && j <= jhi;
            int cv = 0, g = 0;
            if (in) { cv = z(j); g = z(j + 1) - cv; }
            const int G = g < 0 ? -g : g;
            // an annotation needs |g| > min(lo, hi) on the division-free path: most 64-bin words have none at all
            if (mulpath && !ballot_of(in && G > min(P.min_ra, P.max_ra))) continue;
            bool near = false;   // the test's outcome is not the same for every MIN_COV of the band
            if (in) {
                const int x = cv + MIN_COV;
                if (mulpath && x >= band && (unsigned)G < 131072u) {   // thr >= 0 here: the sign of g picks the type, g == 0 never passes
                    const int gf = G * P.cov_frac;
                    if ((G > P.max_ra) || ((G > P.min_ra) && (gf > x))) code = ((reso * j) << 1) | (g > 0 ? 1 : 0);
                    if (band > 0) near = G <= P.max_ra && G > P.min_ra && ((gf > x - band) != (gf > x + band));
                } else {
                    const int thr = min(max(x / P.cov_frac, P.min_ra), P.max_ra);
                    if (g > thr) code = ((reso * j) << 1) | 1;
                    else if (g < -thr) code = ((reso * j) << 1) | 0;
                    near = band > 0;   // (the division path: not analysed, the read waits for the exact MIN_COV)
                }
            }
            if (band > 0 && ballot_of(near)) return ncand | SPEC_DEFERRED;   // nothing was emitted: the read goes on the guard-band list
            const unsigned long long bal = ballot_of(code != -1);
            if (code != -1) cand[ncand + __popcll(bal & ((1ull << lane) - 1ull))] = code;
            ncand += __popcll(bal);
        }
    }
    HINGE_ABLATE_RETURN_V(4)
    if (ncand == 0) {   // (wave-uniform; most reads: no merge, no gate, no work item)
        if (lane == 0) {
            store_at32(anno_offp, iv << 2, 0u);
            store_at32(anno_cntp, iv << 2, 0);
        }
        return 0;
    }
    // merge (filter.cpp:817-829) - sequential on a short list, in place
    int m = 0;
    if (lane == 0) {
        int cur = cand[0];
        for (int t = 1; t < ncand; t++) {
            const int nx = cand[t];
            const bool close = ((nx >> 1) - (cur >> 1)) < P.ra_gap;
            if ((cur & 1) && (nx & 1) && close) {
                // (+,+): drop the later one
            } else if (!(cur & 1) && !(nx & 1) && close) {
                cur = nx;   // (-,-): drop the earlier one
            } else {
                cand[m++] = cur;
                cur = nx;
            }
        }
        cand[m++] = cur;
    }
    m = __builtin_amdgcn_readfirstlane(m);
    if (m > 0 && !cand_in_profile) gate_sums();
    // gate: fp32, IEEE divide, NaN compares false (filter.cpp:861-865)
    bool gate_skip = true;
    if (m > 0) {   // (only read below when the read keeps an annotation: two IEEE divisions, 27 vector instructions)
        const float avg_end = __fdiv_rn((float)E, (float)nE);
        const float avg_start = __fdiv_rn((float)S, (float)nS);
        gate_skip = fabsf(avg_end - avg_start) < 10.0f;
    }
    unsigned off = 0;
    if (lane == 0) {
        const unsigned shard = (unsigned)blockIdx.x & (unsigned)(N_SHARD - 1);
        if (m > 0) {
            const unsigned local = atomicAdd(&o.anno_shard[shard * SHARD_STRIDE], (unsigned)m);
            off = shard * o.anno_region + local;
            if (local + (unsigned)m > o.anno_region) { atomicOr(o.status, ST_ANNO_CAP); m = 0; off = 0; }
        }
        store_at32(anno_offp, iv << 2, off);
        store_at32(anno_cntp, iv << 2, m);
        if (m > 0 && !gate_skip) {
            const unsigned w = shard + (unsigned)N_SHARD * atomicAdd(&o.work_shard[shard * SHARD_STRIDE], 1u);
            if (w >= o.work_cap) atomicOr(o.status, ST_ANNO_CAP);   // (the host grows both buffers and repeats the pass)
            else {
                WorkItem it;
                it.read = i; it.n = n_pile; it.row = row; it.mask_lo = mk.x; it.mask_hi = mk.y; it.off = off; it.cnt = m;
#pragma unroll
                for (int t = 0; t < 4; t++) { const int cd = t < m ? cand[t] : 0; it.anno[t] = make_int2(cd >> 1, (cd & 1) ? 1 : -1); }
                o.work_list[w] = it;
            }
        }
    }
    m = __builtin_amdgcn_readfirstlane(m);
    off = __builtin_amdgcn_readfirstlane(off);
    for (int t = lane; t < m; t += WAVE) {
        const int cd = cand[t];
        o.anno_buf[off + t] = make_int2(cd >> 1, (cd & 1) ? 1 : -1);
        o.hinge_flag[off + t] = 0;
    }
    return ncand;   // slots of cand[] that were written
}

// General kernel.  LDS per wave: h0[kcap] (cutoff-0 difference histogram -> coverage), hc[kcap] (cutoff
// CUT_OFF; reused as the candidate list once the mask is known).  With read_list != nullptr it runs the
// *list_count reads of that list (the fast kernel's hand-backs) instead of [r_begin, r_end].
// One-sweep pass (SpecArgs.mode): MODE_SPEC - d_min_cov is the predicted MIN_COV; every read's mean coverage goes to mean_cov
// (the sums to wave_totals, one slot per wavefront, as k_cov_stats writes them) and a read that is not decided for the whole band
// goes to redo_list instead of emitting.  MODE_FINAL - d_min_cov is exact; the reads of read_list (the guard-band list), or every
// read of [r_begin, r_end] when *spec_state says the prediction missed the band; the coverage bins were stored by the first sweep.
struct SpecArgs {
    int mode, band;
    int* mean_cov;
    unsigned long long* wave_totals;
    const int* spec_state;
    int* redo_list;
    unsigned* redo_count;
    unsigned redo_cap;
    // MODE_SPEC behind the fast kernel (its hand-backs): a read with nbins0[i] >= 0 - well formed, only too long for the fast
    // kernel's LDS slots - is one k_median_hist derives the mean of (from cov_tot[i]) and adds to the totals itself, like every
    // read the fast kernel kept; its sum goes to cov_tot and nothing to mean_cov / wave_totals.  nullptr: every read is this kernel's.
    int* cov_tot;
    const int* nbins0;
};
template <int RESO>
__device__ __forceinline__ void mask_annotate_body(const FilterDev& P, int r_begin, int r_end, const int64_t* __restrict__ row_ptr,
                                                   const int2* __restrict__ a_span, const int* __restrict__ rlen,
                                                   const int* __restrict__ d_min_cov, int kcap, const AnnoOut& o,
                                                   const int* __restrict__ read_list, const unsigned* __restrict__ list_count, const SpecArgs& sa,
                                                   int block_index, int n_blocks) {
    extern __shared__ int lds[];
    const int lane = lane_id();
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform by construction; lets the per-read control flow go scalar
    int* h0 = lds + (size_t)wib * 2 * kcap;
    int* hc = h0 + kcap;
    const int wave = block_index * WAVES_PER_BLOCK + wib;
    const int nwaves = n_blocks * WAVES_PER_BLOCK;
    const int MIN_COV = *d_min_cov;
    const int reso = RESO > 0 ? RESO : P.reso;   // compile-time 40 in the shipped configuration: no runtime divisions
    if (sa.mode == MODE_FINAL && *sa.spec_state != 0) read_list = nullptr;   // the prediction missed the band: everything again
    const int n_items = read_list ? (int)*list_count : r_end - r_begin + 1;
    const int band = sa.mode == MODE_SPEC ? sa.band : 0;
    long long blk_cov = 0, blk_slot = 0;   // MODE_SPEC: this wavefront's share of total_cov / num_slot (filter.cpp:666,672)

    for (int item = wave; item < n_items; item += nwaves) {
        const int i = read_list ? read_list[item] : r_begin + item;
        const int64_t s = row_ptr[i], e = row_ptr[i + 1];
        const int rl = rlen[i];
        const int n = (int)(e - s);
        const int2* __restrict__ row = a_span + s;
        // bins this read can touch: events are <= rlen + cut_off for well-formed input
        int kb = bin_of<RESO>(rl + max(P.cut_off, 0), reso) + 2;
        kb = min(kb, kcap);
        const int kclamp = kb - 1;
        int mx0 = INT_MIN, mxc = INT_MIN;
        bool cleared = false;
        // Hot bins.  Pile-up events are concentrated where overlaps reach the read's ends: about half of all
        // begins fall in the first bin and half of all ends in the last one or two, and 32 lanes adding to one
        // LDS word serialise.  Events in these seven bins are counted in registers (two 16-bit counters per
        // VGPR; pile-ups of 65536+ overlaps take the plain path) and added once per read; integer adds
        // commute, so the histogram is the one a per-event update would give.
        const bool hot = n < 65536;
        const int g0b = hot ? bin_of<RESO>(0, reso) : -1;
        const int g0e = hot ? min(bin_of<RESO>(rl, reso), kclamp) : -1, g0e1 = hot ? g0e - 1 : -1;
        const int gcb = hot ? min(bin_of<RESO>(P.cut_off, reso), kclamp) : -1, gcb1 = hot ? gcb + 1 : -1;
        const int gce = hot ? min(bin_of<RESO>(rl - P.cut_off, reso), kclamp) : -1, gce1 = hot ? gce - 1 : -1;
        int c_b0_e0 = 0, c_e01_bc = 0, c_bc1_ec = 0, c_ec1 = 0;
        for (int base = 0; base < n || !cleared; base += LOADS_IN_FLIGHT * WAVE) {
            int2 v[LOADS_IN_FLIGHT];
#pragma unroll
            for (int u = 0; u < LOADS_IN_FLIGHT; u++) {
                const int k = base + u * WAVE + lane;
                v[u] = k < n ? row[k] : make_int2(0, 0);
            }
            if (!cleared) {   // the histograms are cleared (16 bytes per lane and store) while the first batch is in flight
                int4* z0 = reinterpret_cast<int4*>(h0);
                int4* zc = reinterpret_cast<int4*>(hc);
                for (int t = lane; t < (kb + 3) / 4; t += WAVE) { z0[t] = make_int4(0, 0, 0, 0); zc[t] = make_int4(0, 0, 0, 0); }
                cleared = true;
            }
#pragma unroll
            for (int u = 0; u < LOADS_IN_FLIGHT; u++) {
                if (base + u * WAVE >= n) break;   // wave-uniform: dead slots of the last batch cost nothing
                if (base + u * WAVE + lane < n) {
                    const int2 w = v[u];
                    // an event past rlen + cut_off is malformed input: clamp keeps the LDS write in range, the
                    // maxima below flag it
                    const int b0 = min(bin_of<RESO>(w.x, reso), kclamp), e0 = min(bin_of<RESO>(w.y, reso), kclamp);
                    const int bc = min(bin_of<RESO>(w.x + P.cut_off, reso), kclamp), ec = min(bin_of<RESO>(w.y - P.cut_off, reso), kclamp);
                    if (b0 == g0b) c_b0_e0 += 1; else atomicAdd(&h0[b0], 1);
                    if (e0 == g0e) c_b0_e0 += 0x10000; else if (e0 == g0e1) c_e01_bc += 1; else atomicAdd(&h0[e0], -1);
                    if (bc == gcb) c_e01_bc += 0x10000; else if (bc == gcb1) c_bc1_ec += 1; else atomicAdd(&hc[bc], 1);
                    if (ec == gce) c_bc1_ec += 0x10000; else if (ec == gce1) c_ec1 += 1; else atomicAdd(&hc[ec], -1);
                    mx0 = max(mx0, max(w.x, w.y));
                    mxc = max(mxc, max(w.x + P.cut_off, w.y - P.cut_off));
                }
            }
        }
        if (hot) {   // wave-uniform; wave totals are < 65536 each, so the packed halves cannot carry into each other
            c_b0_e0 = wave_sum(c_b0_e0); c_e01_bc = wave_sum(c_e01_bc); c_bc1_ec = wave_sum(c_bc1_ec); c_ec1 = wave_sum(c_ec1);
            int* hh = nullptr; int idx = -1, val = 0;
            switch (lane) {
                case 0: hh = h0; idx = g0b; val = c_b0_e0 & 0xffff; break;
                case 1: hh = h0; idx = g0e; val = -(int)((unsigned)c_b0_e0 >> 16); break;
                case 2: hh = h0; idx = g0e1; val = -(c_e01_bc & 0xffff); break;
                case 3: hh = hc; idx = gcb; val = (int)((unsigned)c_e01_bc >> 16); break;
                case 4: hh = hc; idx = gcb1; val = c_bc1_ec & 0xffff; break;
                case 5: hh = hc; idx = gce; val = -(int)((unsigned)c_bc1_ec >> 16); break;
                case 6: hh = hc; idx = gce1; val = -(c_ec1 & 0xffff); break;
                default: break;
            }
            if (val != 0) atomicAdd(&hh[idx], val);   // a counter is non-zero only if some event had that (valid) bin
        }
        HINGE_ABLATE_POINT(1)
        mx0 = wave_max(mx0);
        mxc = wave_max(mxc);
        const int K0 = nbins_of<RESO>(n, mx0, reso);
        const int KC = nbins_of<RESO>(n, mxc, reso);
        if (max(K0, KC) > kb) {
            if (lane == 0) atomicOr(o.status, ST_RANGE);
            continue;
        }

        // ---- both prefix scans in one sweep; coverage mask on the cutoff bins (filter.cpp:696-728) ----
        int carry = 0, carry0 = 0;   // running coverage (cutoff / cutoff-0)
        RunState run{0, 0ull, 0, 0};
        unsigned long long near = 0ull;   // MODE_SPEC: bins whose `c > MIN_COV` is not the same for every MIN_COV of the band
        const int Kmax = max(K0, KC);
        const bool packed = n < 32768;     // cutoff-0 prefix in [0, n], cutoff prefix in [-n, n]: one 16|16 scan does both
        for (int base = 0; base < Kmax; base += WAVE) {
            const int j = base + lane;
            int c = j < KC ? hc[j] : 0;
            int z = j < K0 ? h0[j] : 0;
            if (packed) {
                const int pk = wave_incl_scan(z + c * 65536) + carry;
                carry = wave_last(pk);
                z = pk & 0xffff;
                c = pk >> 16;
            } else {
                c = wave_incl_scan(c) + carry;
                z = wave_incl_scan(z) + carry0;
                carry = wave_last(c);
                carry0 = wave_last(z);
            }
            if (j < KC) hc[j] = c;
            if (j < K0) h0[j] = z;
            if (base >= KC) continue;   // wave-uniform
            const int left = KC - base;
            const unsigned long long V = left >= 64 ? ~0ull : ((1ull << left) - 1ull);
            run_feed(run, base, __ballot(c > MIN_COV) & V, V, reso);   // c[j] > 0 after subtracting MIN_COV
            if (band > 0) near |= (__ballot(c > MIN_COV - band) ^ __ballot(c > MIN_COV + band)) & V;
        }
        if (sa.mode == MODE_SPEC) {   // the read's mean coverage, as k_cov_stats has it (filter.cpp:642-656)
            long long t = 0;
            for (int j = lane; j < K0; j += WAVE) t += h0[j];
            t = wave_sum64(t);
            if (lane == 0 && sa.cov_tot && sa.nbins0[i] >= 0) {
                sa.cov_tot[i] = (int)t;            // (K0 == nbins0[i]: both are profileCoverage's K of the same pile-up)
            } else if (lane == 0) {
                if (rl >= 5000) {
                    sa.mean_cov[i] = (int)(t / (long long)max(1, K0));
                    blk_cov += t;
                    blk_slot += K0;
                } else {
                    sa.mean_cov[i] = MEAN_SENTINEL;
                }
            }
        }
        int used = 0;
        if (near == 0ull)
            used = mask_gate_annotate(P, reso, MIN_COV, i, lane, K0, run, [&](int j) { return h0[j]; }, [&](int j) { return hc[j]; }, hc, o, (long long)s, n,
                                      sa.mode == MODE_FINAL /*the first sweep stored the bins*/,
                                      false /*the candidates go where the cutoff profile was: the gate reads the plain one*/, ~0ull, band);
        else if (o.cov_out) {   // (a deferred read still owes its coverage bins: the final launch does not store them)
            int* __restrict__ dst = o.cov_out + o.cov_off[i - o.cov_base];
            for (int j = lane; j < K0; j += WAVE) dst[j] = h0[j];
            if (lane == 0) o.cov_nbins[i - o.cov_base] = K0;
        }
        if ((near != 0ull || (used & SPEC_DEFERRED)) && lane == 0) {
            const unsigned at = atomicAdd(sa.redo_count, 1u);
            if (at < sa.redo_cap) sa.redo_list[at] = i; else atomicOr(o.status, ST_REDO_CAP);
        }
    }
    if (sa.mode == MODE_SPEC && lane == 0) {
        sa.wave_totals[2 * wave] = (unsigned long long)blk_cov;
        sa.wave_totals[2 * wave + 1] = (unsigned long long)blk_slot;
    }
}
template <int RESO>
__global__ __launch_bounds__(BLOCK) void k_mask_annotate(FilterDev P, int r_begin, int r_end, const int64_t* __restrict__ row_ptr,
                                                         const int2* __restrict__ a_span, const int* __restrict__ rlen,
                                                         const int* __restrict__ d_min_cov, int kcap, AnnoOut o,
                                                         const int* __restrict__ read_list, const unsigned* __restrict__ list_count, SpecArgs sa) {
    mask_annotate_body<RESO>(P, r_begin, r_end, row_ptr, a_span, rlen, d_min_cov, kcap, o, read_list, list_count, sa, (int)blockIdx.x, (int)gridDim.x);
}
// MODE_FINAL for up to MASK_FINAL_BATCH_MAX resident parts in ONE launch (block b works for part b % n): the guard-band lists are
// ~1 % of a part's reads each, the launch is a chain of a few dependent look-ups whose length does not depend on how many
// parts share it (four launches: 4 x 17 us; one: 18 us - profiles/r4a vs r4b).
constexpr int MASK_FINAL_BATCH_MAX = 8;
struct MaskFinalPart {
    int r_begin, r_end;
    const int64_t* row_ptr; const int2* a_span; const int* rlen; const int* d_min_cov;
    AnnoOut o;
    const int* read_list; const unsigned* list_count;
    SpecArgs sa;
};
struct MaskFinalBatch { int n; MaskFinalPart part[MASK_FINAL_BATCH_MAX]; };
template <int RESO>
__global__ __launch_bounds__(BLOCK) void k_mask_final_batch(FilterDev P, const MaskFinalBatch* __restrict__ B, int kcap) {
    typedef const MaskFinalBatch __attribute__((address_space(4))) BatchK;   // constant address space: scalar loads
    BatchK& b = *(BatchK*)(unsigned long long)B;
    const int n = b.n;
    const int part = (int)blockIdx.x % n;
    const MaskFinalPart a = const_cast<const MaskFinalBatch*>(B)->part[part];
    mask_annotate_body<RESO>(P, a.r_begin, a.r_end, a.row_ptr, a.a_span, a.rlen, a.d_min_cov, kcap, a.o, a.read_list, a.list_count, a.sa,
                             (int)blockIdx.x / n, (int)gridDim.x / n);
}

// Fast kernel for reso = 40 and cut_off = 20 * SH >= 0.  Per wave: Pq[qcap] = begin|end counts per 20-bp bin
// (begins in the low 16 bits, ends in the high 16), then their inclusive prefixes PB|PE.  From those
//     cov0[k] = PB[2k-1] - PE[2k-1]               (bin_of(v) <= k  <=>  v/20 <= 2k-1)
//     covc[k] = PB[2k-1-SH] - PE[2k-1+SH]         ((v +- cut_off)/20 = v/20 +- SH exactly)
// with PB[<0] = 0 and P[> last] = P[last]: 2 bin computations and 2 LDS adds per overlap instead of 4 + 4.
// Events in the four hot bins (begins in bins 0-1, ends in the read's last two) go to lane-private LDS
// words (hot[4][64]) so they never collide; they are summed once per read.
// A read goes to the fallback list (run by k_mask_annotate afterwards) when its pile-up has 65536+ overlaps
// or any coordinate lies outside [0, rlen], or the read is too long even for a whole workgroup's LDS.
// The constants of the read's last phase (FilterDev, the output pointers) are passed in device memory behind one pointer and read
// through the constant address space where they are used (scalar loads, a field at a time): as by-value kernel arguments they are
// live across the whole read loop and cost 58 SGPR spills to VGPR lanes - the reloads alone were ~90 v_readlane per read,
// 15 % of the kernel's time (ablation build, tools/ablate_k2.sh).
struct K2Const {
    FilterDev P;
    AnnoOut o;
    int* fallback_list;         // reads handed back to the general kernel (hardly ever: the pointers are looked up when one is)
    unsigned* fallback_count;
    int* redo_list;             // SPEC: the guard-band list (reads that wait for the exact MIN_COV), its length and capacity
    unsigned* redo_count;
    unsigned redo_cap;
};
constexpr int K2_MAX_HEADS = 64;
struct K2Heads { unsigned base[K2_MAX_HEADS]; };   // value of every item counter before this launch
#ifdef HINGE_K2_TRACE
// Trace builds (-DHINGE_ABLATE -DHINGE_K2_TRACE): per-read time stamps (100 MHz s_memrealtime) of k_mask_annotate_q20, five per
// list item: read start, histogram done, scan done, mask pass done, read done (tools/k2_trace.py).  Not part of the plain ablation
// build: the stamps change the kernel's register allocation and instruction counts.
__device__ unsigned long long* g_k2_trace = nullptr;
#define HINGE_K2_STAMP(k) do { if (k2tr && lane == 0) k2tr[5 * (size_t)item + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define HINGE_K2_STAMP(k) do { } while (0)
#endif

// (eight wavefronts per SIMD: with the scalar and the vector unit both ~60 % busy the kernel is latency-sensitive again - 84.7 us at
// six (77 VGPRs, the compiler's choice), 79.7 at seven, 78.3 at eight with a one-VGPR scratch spill and 52 SGPR spills)
// COVOUT: the .coverage.txt bins are written too (a template parameter, like the flags below that became launch conditions: a
// run-time flag of this kernel is a lane mask or a scalar that lives - spilled - across the whole read loop).
// CUT20: cut_off / 20 when it is the shipped 300 (the pads, the profile accessors' offsets and the bounds below are then immediates), else -1.
// SPEC: the first sweep of a one-sweep pass (see MODE_SPEC above): *d_min_cov is the PREDICTED MIN_COV, the read's coverage sum
// goes to cov_tot (it falls out of the prefix scan: cov0[k] = PB[2k-1] - PE[2k-1] is zero behind the last event, so the sum of the
// scan's values at the odd indices is the sum over the read's bins), and a read that is not decided for every MIN_COV in
// [pred - band, pred + band] emits nothing and goes on the guard-band list.
template <bool PACKED, bool COVOUT, int CUT20, int SPEC /*0: a pass with the exact MIN_COV; 1: first sweep of a one-sweep pass, band 1 (a constant: MIN_COV +- band cost no registers); 2: any band*/>
__device__ __forceinline__ void k2_q20_body(const int vblock /*the workgroup's index among its PART's workgroups*/, const K2Const* __restrict__ C, int cut_off_arg, int mulpath_thr /*min(MIN_RA, MAX_RA) >= 0: the
                                                             division-free annotation test applies (a launch condition)*/,
                                                             int nhr /*NO_HINGE_REGION*/, int cov_mask_off /*INT_MIN if the coverage mask takes part in the mask, else 1 << 29*/,
                                                             const int* __restrict__ read_list, int n1, int n2, int n4,
                                                             const int64_t* __restrict__ row_ptr,
                                                             const typename SpanLoad<PACKED>::raw* __restrict__ a_span, const int* __restrict__ rlen,
                                                             const int* __restrict__ nbins0, const int* __restrict__ d_min_cov, int slot_ints,
                                                             int* __restrict__ cov_out /*COVOUT: the coverage-bin output*/,
                                                             const long long* __restrict__ cov_off, int* __restrict__ cov_nbins, int cov_base,
                                                             unsigned* __restrict__ heads, int n_heads, const unsigned* __restrict__ head_bases /*value of every item counter before this launch*/,
                                                             int* __restrict__ cov_tot /*SPEC*/, int band_arg /*SPEC == 2*/) {
    const int band = SPEC == 1 ? 1 : band_arg;
    extern __shared__ int lds[];
    constexpr int HOT = 4;    // words per lane the slot has room for behind the profile (candidate list of the last phase)
    constexpr int HOTW = 2;   // of which hot words: W0 = begins in bin 0 | ends in bin qe - 1 << 16, W1 = begins in bin 1 | ends in bin qe << 16
    const int lane = lane_id();
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // A workgroup has four LDS slots of slot_ints words.  read_list = [n1 reads that fit one slot, longest first | n2 reads that
    // need two | n4 reads that need all four].  The grid is [n4 workgroups that run one long read each | ceil(n2/2) that run two
    // (wavefronts 0 and 2, each over two slots) | the PERSISTENT workgroups of the n1 short reads, as many as the GPU holds at
    // once]: the long reads - the most expensive items of the launch - start first, and the wavefronts of the persistent
    // workgroups take the short reads one at a time, the next one fetched while the current one is worked on.
    // Where from: ONE device counter serialises - a returning device-scope atomic on one word takes 11.7 ns, 1.1 ms for the
    // 87 k reads of the bench part (measured; the guide's "one word saturates at 88 dequeues / us").  So there are n_heads (<= 64)
    // counters, 128 bytes apart; persistent workgroup pb draws from head pb % n_heads, whose items are h, h + n_heads, ... of the
    // sorted list (every head the same mix of lengths, every head the same number of workgroups: n_heads divides their count).
    // The counters only ever grow: a launch advances head h by its wavefronts + its items, the host keeps the sums and passes
    // where this launch starts (`bases`).
    // Round 1 gave every wavefront a fixed number of reads and every four such wavefronts a workgroup, in list order with the long
    // reads last: the per-read time stamps of an ablation build (tools/k2_trace.py) showed 18 % of the resident wavefront slots
    // idle behind their workgroup's slowest wavefront and the last 15 % of the launch at falling occupancy.  Dealing the reads to
    // the workgroups on the host by estimated cost (least-loaded-first, dynamic only inside a workgroup) was worse than that: a
    // read's time is not predictable enough from its length (117 us; the workgroups ended between 50 % and 100 % of the launch).
    const int g4 = n4, g2 = (n2 + 1) / 2;
    int width, item, item_end;
    bool dyn = false;
    unsigned grab = 0, head_base = 0;
    int head = 0;
    unsigned* head_ptr = nullptr;
    if (vblock < g4) { width = 4; if (wib != 0) return; item = n1 + n2 + vblock; item_end = item + 1; }
    else if (vblock < g4 + g2) { width = 2; if (wib & 1) return; item = (vblock - g4) * 2 + (wib >> 1); if (item >= n2) return; item += n1; item_end = item + 1; }
    else {
        width = 1; dyn = true; item = 0; item_end = n1;
        head = (vblock - g4 - g2) % n_heads;
        head_ptr = heads + head * 32;
        head_base = head_bases[head];
        if (lane == 0) grab = atomicAdd(head_ptr, 1u);   // (its latency is covered by the set-up below)
    }
    constexpr int reso = 40;
    const int SH = CUT20 >= 0 ? CUT20 : cut_off_arg / 20;
    const int cut_off = CUT20 >= 0 ? CUT20 * 20 : cut_off_arg;
    // Zero words in front of the prefix array and copies of the totals behind it make PB[q < 0] = 0 and P[q > last] = P[last]
    // plain loads: the profile accessors below need no clamps and issue their LDS reads back to back.
    const int PADF = (SH + 2 + 3) & ~3, PADT = (2 * SH + 4 + 3) & ~3;
    const int qcap = width * slot_ints - HOT * WAVE - PADF - PADT;
    int* Pq = lds + (size_t)wib * slot_ints + PADF;
    int* hot = Pq + qcap + PADT;                   // lane-private words for the four hot bins (see below)
#pragma unroll
    for (int h = 0; h < HOTW; h++) hot[h * WAVE + lane] = 0;
    int* const hot_l = hot + lane;                 // begins: + q * 64 for q in {0, 1}; ends: + (q - (qe - 1)) * 64 for q in {qe - 1, qe}
    const int MIN_COV = *d_min_cov;
#ifdef HINGE_ABLATE
    struct { int ablate; } P = {C->P.ablate};
#endif
#ifdef HINGE_K2_TRACE
    unsigned long long* const k2tr = g_k2_trace;
#endif
    for (int t = lane; t < PADF; t += WAVE) Pq[t - PADF] = 0;
    auto drawn = [&]() { return head + n_heads * (int)((unsigned)__builtin_amdgcn_readfirstlane((int)grab) - head_base); };
    if (dyn) item = drawn();
    for (; (unsigned)item < (unsigned)item_end; item = dyn ? drawn() : item_end) {   // `continue` leaves a read
        if (dyn && lane == 0) grab = atomicAdd(head_ptr, 1u);   // the item after this one
        HINGE_K2_STAMP(0);
        const int i = load_at32(read_list, (unsigned)item << 2);
        struct Bounds { int64_t s, e; };
        const Bounds rb = load_at32(reinterpret_cast<const Bounds*>(row_ptr), (unsigned)i << 3);   // row_ptr[i], row_ptr[i + 1]
        const int64_t s = rb.s, e = rb.e;
        const int rl = load_at32(rlen, (unsigned)i << 2);
        const int K0 = load_at32(nbins0, (unsigned)i << 2);   // k_cov_stats: bins of the plain profile, or -1 if a coordinate leaves [0, rl]
        const long long cov_at = COVOUT ? load_at32(cov_off, (unsigned)(i - cov_base) << 3) : 0;   // (fetched with the row bounds, used after phase 1)
        // (all five look-ups in flight together: left alone the compiler moves the row bounds behind the hand-back test below - a
        // third dependent round trip per read, 69.1 -> 70.3 us)
        asm volatile("" :: "s"(rb.s), "s"(rb.e), "s"(cov_at));
        const int64_t n64 = e - s;
        const int qe = rl / 20;                       // last bin an event can fall in
        // k_cov_stats says no (16-bit counts would overflow, malformed), or too long for the LDS (then also: bins < 16384 for the keys
        // of the run search below): general kernel
        if (K0 < 0 || qe >= qcap) {
            if (lane == 0) { int* const fl = C->fallback_list; fl[atomicAdd(C->fallback_count, 1u)] = i; }
            continue;
        }
        const int n = (int)n64;
        typedef SpanLoad<PACKED> SL;
        const typename SL::raw* __restrict__ row = a_span + s;
        const int Qn = qe + 1;                        // bins in use
        const unsigned last = n > 0 ? (unsigned)(n - 1) : 0u;
        // the ends' hot word from the bin itself: hot_l + (qd - (qe - 1)) * 64 (one shift-add per event off a per-read base)
        // (LDS byte addresses the compiler must take as they are: with the pads' sizes known at compile time it splits every base
        // into a register and a constant and adds the constant per event)
        typedef __attribute__((address_space(3))) int lds_int;
        unsigned pq_a = (unsigned)(uintptr_t)(lds_int*)Pq, hot_b = (unsigned)(uintptr_t)(lds_int*)hot_l;
        asm("" : "+s"(pq_a));
        asm("" : "+v"(hot_b));
        unsigned hot_e = hot_b - (unsigned)((qe - 1) * WAVE * 4);
        asm("" : "+v"(hot_e));
        auto event = [&](typename SL::raw raw) {
            // k_cov_stats vouches for 0 <= abpos, aepos <= rl: the bins need no clamp
            const int2 w = SL::get(raw);
            unsigned qb = (unsigned)w.x / 20u, qd = (unsigned)w.y / 20u;
            // (the tests on the BINS: left to itself the compiler tests the begin's position - an extra mask of the packed word)
            asm("" : "+v"(qb));
            asm("" : "+v"(qd));
            const bool hb = qb < 2u, he = (int)qd > qe - 2;
            __builtin_amdgcn_sched_barrier(0);   // (the compares first: a select right behind its compare costs wait states)
            lds_int* pb = (lds_int*)(uintptr_t)(hb ? hot_b + qb * (WAVE * 4) : pq_a + qb * 4);
            lds_int* pe = (lds_int*)(uintptr_t)(he ? hot_e + qd * (WAVE * 4) : pq_a + qd * 4);
            __hip_atomic_fetch_add(pb, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(pe, 0x10000, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        auto clear_bins = [&]() {   // (under the first batch's loads)
            int4* z4 = reinterpret_cast<int4*>(Pq);
            for (int t = lane; t < (Qn + 3) / 4; t += WAVE) z4[t] = make_int4(0, 0, 0, 0);
        };
        if constexpr (PACKED) {
            // The 16|16 copy is padded by half a batch, so the loads need neither a clamp nor a branch each: scalar row base + one
            // lane offset + immediates.  As many loads as the row has 64-overlap batches left, in eight tiers (three uniform
            // branches): the reads are not visited in storage order, so what a wavefront loads past its row's end is fetched from HBM
            // for nothing.  A tier knows how many of its batches are FULL: those run without a lane test and without the uniform
            // "any overlap left?" test - per batch an exec-mask save / restore, two branches and a compare less on the scalar side.
            static_assert(LOADS_IN_FLIGHT == 8, "load tiers below");
            int base = 0;
            do {
                const typename SL::raw* __restrict__ p = row + (unsigned)(base + lane);
                const int left = n - base;
                const int rem = left - lane;       // this lane has an overlap in batch u iff u * 64 < rem
                typename SL::raw v[LOADS_IN_FLIGHT];
#define HINGE_K2_TIER(CNT, FULL)                                                         \
                {                                                                        \
                    _Pragma("unroll") for (int u = 0; u < (CNT); u++) v[u] = p[u * WAVE]; \
                    if (base == 0) clear_bins();                                         \
                    _Pragma("unroll") for (int u = 0; u < (FULL); u++) event(v[u]);      \
                    _Pragma("unroll") for (int u = (FULL); u < (CNT); u++) if (u * WAVE < rem) event(v[u]); \
                }
                if (left > 4 * WAVE) {
                    if (left >= 8 * WAVE) HINGE_K2_TIER(8, 8)
                    else if (left > 6 * WAVE) HINGE_K2_TIER(8, 6)
                    else if (left > 5 * WAVE) HINGE_K2_TIER(6, 5)
                    else HINGE_K2_TIER(5, 4)
                } else if (left > 2 * WAVE) {
                    if (left > 3 * WAVE) HINGE_K2_TIER(4, 3) else HINGE_K2_TIER(3, 2)
                } else if (left > WAVE) HINGE_K2_TIER(2, 1)
                else if (left > 0) HINGE_K2_TIER(1, 0)
                else if (base == 0) clear_bins();
#undef HINGE_K2_TIER
                base += LOADS_IN_FLIGHT * WAVE;
            } while (base < n);
        } else {
            bool cleared = false;
            for (int base = 0; base < n || !cleared; base += LOADS_IN_FLIGHT * WAVE) {
                typename SL::raw v[LOADS_IN_FLIGHT];
                if (n > 0) {   // (the int32 spans may be the caller's buffer: clamped index)
#pragma unroll
                    for (int u = 0; u < LOADS_IN_FLIGHT; u++) v[u] = row[min((unsigned)(base + u * WAVE + lane), last)];
                }
                if (!cleared) { clear_bins(); cleared = true; }
                const int rem = n - base - lane;
#pragma unroll
                for (int u = 0; u < LOADS_IN_FLIGHT; u++) {
                    if (base + u * WAVE >= n) break;   // wave-uniform
                    if (u * WAVE < rem) event(v[u]);
                }
            }
        }
        HINGE_K2_STAMP(1);
        HINGE_ABLATE_POINT(9)    // (ablation builds: 9 = stop after the histogram, before the hot-word fold)
        {   // fold the lane-private hot words into their bins (and zero them for the next read)
            const int h0 = hot[lane], h1 = hot[WAVE + lane];
            hot[lane] = 0; hot[WAVE + lane] = 0;
            // per-lane counts are < 65536, wave totals too (n < 65536): the two halves of a word never meet
            const int t0 = wave_incl_scan(h0);   // last lane: begins in bin 0 | ends in bin qe - 1 << 16
            const int t1 = wave_incl_scan(h1);   // last lane: begins in bin 1 | ends in bin qe << 16
            // The last lane adds all four where it holds them (no broadcast, no lane masks to keep across the loop).  Plain ds_add
            // instructions: from atomicAdd on a uniform address the compiler builds a wave-aggregated atomic (mbcnt, bcnt, multiply).
            // A zero count adds zero; with qe = 0 the third add goes to the zero pad in front of the bins, with zero.
            if (lane == WAVE - 1) {
                const unsigned a0 = (unsigned)(size_t)(__attribute__((address_space(3))) int*)Pq;
                const unsigned a1 = a0 + (unsigned)(qe * 4);
                asm volatile("ds_add_u32 %0, %2\n\tds_add_u32 %0, %3 offset:4\n\tds_add_u32 %1, %4\n\tds_add_u32 %1, %5 offset:4"
                             :: "v"(a0), "v"(a1 - 4u), "v"(t0 & 0xffff), "v"(t1 & 0xffff), "v"(t0 & (int)0xffff0000), "v"(t1 & (int)0xffff0000) : "memory");
            }
        }
        HINGE_ABLATE_POINT(1)
        // The cutoff profile is zero from its last bin on (every event consumed: begins - ends = 0), and a zero bin that
        // follows a zero bin changes nothing in the run search, so any bound >= the reference's K works: the largest a
        // well-formed pile-up can have needs no reduction.
        const int KC = nbins_of<40>(n, rl + cut_off, reso);

        // ---- inclusive prefixes of begins|ends, 8 consecutive bins per lane (512 per step: one step for reads of up to 10 kb) -----
        int carry = 0;
        int tot_l = 0;   // SPEC: this lane's share of the read's coverage sum
        int* __restrict__ const cov_dst = COVOUT ? cov_out + cov_at : (int*)nullptr;
        // Which 64-bin words of the plain profile can hold an annotation at all: an annotation needs |cov0[k+1] - cov0[k]| above
        // min(MIN_RA, MAX_RA), and that difference is simply the begins minus the ends of the two 20-bp bins 2k, 2k+1 - the raw
        // counts this loop holds before it sums them.  A typical read has no such bin between its ends' pile-ups.
        unsigned long long flag_words = 0ull;
        // ... and only between the bounds every annotation window respects: 40 j >= mask.first + NHR >= NHR, j <= K0 - 3, and with
        // the coverage mask 40 j <= mask.second - NHR <= rl - cut_off - NHR (a bin of the cutoff profile is positive only while an
        // overlap still has cut_off bases to go, so the mask ends at rl - cut_off at the latest).  The pile-ups of begins at the
        // read's start and of ends at its end - which exceed the threshold in every read - lie outside them.
        const int jlo_b = max(nhr, 0) / reso;
        const int jhi_b = min(K0 - 3, max(rl - cut_off - nhr < 0 ? -1 : (rl - cut_off - nhr) / reso, cov_mask_off));
        // as one unsigned range test per lane: a lane's four 40-bp bins are hb + 4 lane .. + 3 (hb = first 40-bp bin of the step),
        // some of them inside [jlo_b, jhi_b] iff 4 lane - (b_lo - hb) <= b_span (no bin at all: b_lo beyond every lane)
        const bool no_bins = jhi_b - jlo_b + 1 <= 0;
        const int b_lo = no_bins ? (1 << 30) : jlo_b - 3;
        const unsigned b_span = no_bins ? 0u : (unsigned)(jhi_b - jlo_b + 3);
#ifdef HINGE_ABLATE
        if (P.ablate != 6 && P.ablate != 8)
#endif
        for (int base = 0; base < Qn; base += 8 * WAVE) {
            const int t = base + 8 * lane;
            // (lanes past the last bin read the zero pad in front of the bins: an address select instead of an exec-mask save /
            // restore and four register clears per load)
            int4 v = *reinterpret_cast<const int4*>(t < Qn ? Pq + t : Pq - PADF);
            int4 w = *reinterpret_cast<const int4*>(t + 4 < Qn ? Pq + t + 4 : Pq - PADF);
            v.y += v.x;                                        // (= begins|ends of this lane's first 40-bp bin, both halves below 65536)
            w.y += w.x;
            {
                // |begins - ends| > thr  <=>  (unsigned)(begins - ends + thr) > 2 thr (thr < 2^28, the host sees to it)
                const int s1 = v.z + v.w, s3 = w.z + w.w;
                auto excess = [&](int pair) { return (unsigned)((pair & 0xffff) - (int)((unsigned)pair >> 16) + mulpath_thr); };
                const unsigned um = max(max(excess(v.y), excess(s1)), max(excess(w.y), excess(s3)));
                const int hb = base >> 1;
                // (two ballots and a scalar AND: the ballot of `a && b` re-materialises the predicate on the vector side)
                const unsigned long long bal = ballot_of((unsigned)(4 * lane - (b_lo - hb)) <= b_span) & ballot_of(um > 2u * (unsigned)mulpath_thr);
                // a step covers four 64-bin words (16 lanes each); flagged two at a time: lanes 0-31 -> words w, w + 1, lanes 32-63 ->
                // w + 2, w + 3.  Scalar minima written as such: from `x != 0` or from min(x, 1) the compiler makes a zero-extended
                // boolean, selects it on the vector side and drags the whole flag word into vector registers (25 vector instructions
                // per step, seen in the ISA)
                unsigned f0, f1;
                asm("s_min_u32 %0, %1, 1" : "=s"(f0) : "s"((unsigned)bal) : "scc");
                asm("s_min_u32 %0, %1, 1" : "=s"(f1) : "s"((unsigned)(bal >> 32)) : "scc");
                const unsigned f = f0 * 3u + f1 * 12u;
                // (w = base / 128 is a multiple of 4; beyond word 63 the shift wraps and flags low words for nothing, which is only
                // a look too many: the candidate pass looks at every word from the 64th on anyway)
                flag_words |= (unsigned long long)f << ((base >> 7) & 63);
            }
            v.z += v.y; v.w += v.z;
            w.x += v.w; w.y += v.w; w.z += w.y; w.w += w.z;
            const int incl = wave_incl_scan(w.w);
            const int excl = incl - w.w + carry;
            v.x += excl; v.y += excl; v.z += excl; v.w += excl;
            w.x += excl; w.y += excl; w.z += excl; w.w += excl;
            if (t < Qn) *reinterpret_cast<int4*>(Pq + t) = v;
            if (t + 4 < Qn) *reinterpret_cast<int4*>(Pq + t + 4) = w;
            if constexpr (SPEC != 0 && !COVOUT) {
                auto cv = [&](int pre) { return (pre & 0xffff) - (int)((unsigned)pre >> 16); };
                tot_l += cv(v.y) + cv(v.w) + cv(w.y) + cv(w.w);
            }
            if constexpr (COVOUT) {
                // the .coverage.txt bins straight from the registers of the scan: cov0[k] = PB[2k-1] - PE[2k-1], and this lane holds
                // the prefixes of bins t .. t+7, i.e. 2k-1 = t+1, t+3, t+5, t+7 (k = t/2 + 1 .. t/2 + 4): one 16-byte store (4-byte
                // aligned; a read's area has room for the up to three values it writes past the read's last bin).  Stored here - not
                // with the other outputs at the end of the read - they also have the rest of the read's work to drain: stores count
                // in vmcnt like loads on this architecture, so the next read's first wait for its spans waits for every store issued
                // before it.
                const int k1 = (t >> 1) + 1;
                auto cv = [&](int pre) { return (pre & 0xffff) - (int)((unsigned)pre >> 16); };
                struct __attribute__((packed, aligned(4))) Bins4 { int a, b, c, d; };
                const Bins4 b4{cv(v.y), cv(v.w), cv(w.y), cv(w.w)};
                if (k1 < K0) *reinterpret_cast<Bins4*>(cov_dst + k1) = b4;
                if constexpr (SPEC != 0) tot_l += (b4.a + b4.b) + (b4.c + b4.d);   // (lanes past the last event hold begins = ends: zero)
            }
            carry += wave_last(incl);
        }
        if (COVOUT && lane == 0) {
            if (K0 > 0) cov_dst[0] = 0;                      // cov0[0]: nothing is consumed before position 0
            store_at32(cov_nbins, in_vgpr((unsigned)(i - cov_base)) << 2, K0);
        }
        if constexpr (SPEC != 0) {   // the last lane of the scan holds the wavefront's sum: it stores it (no broadcast)
            const int tot = wave_incl_scan(tot_l);
            if (lane == WAVE - 1) store_at32(cov_tot, in_vgpr((unsigned)i) << 2, tot);
        }
        {   // copies of the totals behind the scanned bins (the scan ran over [0, round-up-to-4 of Qn))
            const int Qs = (Qn + 3) & ~3;
            if (PADT <= WAVE) { if (lane < PADT) Pq[Qs + lane] = carry; }   // (cut_off <= 600)
            else for (int t = lane; t < PADT; t += WAVE) Pq[Qs + t] = carry;
        }
        HINGE_K2_STAMP(2);
        auto cov0 = [&](int k) { const int p = Pq[2 * k - 1]; return (p & 0xffff) - (int)((unsigned)p >> 16); };
        auto covc = [&](int k) { return (Pq[2 * k - 1 - SH] & 0xffff) - (int)((unsigned)Pq[2 * k - 1 + SH] >> 16); };
        // ---- coverage mask on the cutoff profile: the first longest run of bins above MIN_COV (filter.cpp:696-728) ------------
        // On the VECTOR side.  run_feed (the general kernel's form) walks the set bits of a ballot on the scalar unit, ~55 scalar
        // instructions for a word that is not all ones - and a read's first and last word never are.  The scalar unit is what
        // bounds this kernel (one per CU: 455 scalar against 290 vector instructions per read, 75 % against 47 % busy), so here a
        // lane owns bin j = base + lane:
        //   z(j)  = largest non-positive bin <= j, 0 if none      (inclusive max-scan of `positive ? 0 : j`, carried across words)
        //   bin j closes a run iff it is valid, non-positive and bin j - 1 is positive; the run has j - z(j - 1) - 2 bins
        //   key   = bins << 14 | (16383 - j): the largest key is the longest run and, among equals, the first (j < 16384: reads
        //           beyond 600 kb are handed back above); every lane keeps the largest key it saw, ONE wave maximum at the end.
        // A run still open at the last valid bin is not counted, as in the reference's loop.
        int key_best = 0;
        bool near_band = false;   // SPEC: some bin's `covc > MIN_COV` depends on where in the band the exact MIN_COV lies
        {
            int zc = 0, pc = 0;                        // carries: z and positivity of the previous word's last bin
            const int* pcb = Pq + 2 * lane - 1 - SH;   // covc(base + lane) = begins below pcb[2 base] - ends below pce[2 base]
            const int* pce = Pq + 2 * lane - 1 + SH;
            // a word that is not all positive (a read's first and last word, hardly another)
            auto word = [&](int base, bool p, unsigned long long M) {
                const int j = base + lane;
                const int z = max(wave_incl_max_scan(p ? 0 : j), zc);   // (only valid bins reach here as non-positive ones: see the tail below)
                const int zprev = shfl_up1(z, zc);
                const int pprev = shfl_up1(p ? 1 : 0, pc);
                const int bins = j - zprev - 2;
                const int key = (bins << 14) | (16383 - j);
                key_best = max(key_best, (!p && pprev != 0 && bins > 0) ? key : 0);
                zc = wave_last(z);
                pc = (int)(M >> 63);
            };
            // A word in which no run ends (no non-positive bin behind a positive one: a read's FIRST word, where the coverage rises
            // once) only moves the carries - a dozen scalar instructions instead of the ~45 of the vector form.
            auto closes_none = [&](int base, unsigned long long M) {
                const unsigned long long N = ~M;
                if (N & ((M << 1) | (unsigned long long)pc)) return false;
                if (N) zc = base + 63 - __builtin_clzll(N);   // the word's last non-positive bin (bins ascend: it is the largest so far)
                pc = (int)(M >> 63);
                return true;
            };
            int base = 0;
#ifdef HINGE_ABLATE
            if (P.ablate == 7 || P.ablate == 8) base = KC;
#endif
            // (loaded by every lane: LDS reads do not fault)
            for (; base + WAVE <= KC; base += WAVE, pcb += 2 * WAVE, pce += 2 * WAVE) {
                const int cvv = (*pcb & 0xffff) - (int)((unsigned)*pce >> 16);
                if constexpr (SPEC != 0) {
                    // 64 bins above the band's upper end: positive whatever the exact MIN_COV turns out to be
                    const unsigned long long MH = ballot_of(cvv > MIN_COV + band);
                    if (MH == ~0ull) { pc = 1; continue; }
                    if (ballot_of(cvv > MIN_COV - band) != MH) near_band = true;   // a bin inside the band: the read waits
                }
                const bool p = cvv > MIN_COV;
                const unsigned long long M = ballot_of(p);
                // 64 bins above MIN_COV (the interior of nearly every read) open or continue a run and close none
                if (M == ~0ull) { pc = 1; continue; }
                if (closes_none(base, M)) continue;
                word(base, p, M);
            }
            if (base < KC) {   // the last, partial word: the bins from KC on count as positive ones that belong to no run (they close
                               // none, and a run that is still open at the last valid bin is not counted, as in the reference's loop)
                const bool valid = base + lane < KC;
                const int cvv = (*pcb & 0xffff) - (int)((unsigned)*pce >> 16);
                if constexpr (SPEC != 0) {
                    if (ballot_of(valid && cvv > MIN_COV - band) != ballot_of(valid && cvv > MIN_COV + band)) near_band = true;
                }
                const bool p = !valid || cvv > MIN_COV;
                const unsigned long long M = ballot_of(p);
                if (!closes_none(base, M)) word(base, p, M);
            }
        }
        if constexpr (SPEC != 0) {
            if (near_band) {   // (wave-uniform) nothing emitted: the guard-band list takes the read
                if (lane == 0) {
                    const unsigned at = atomicAdd(C->redo_count, 1u);
                    if (at < C->redo_cap) C->redo_list[at] = i; else atomicOr(C->o.status, ST_REDO_CAP);
                }
                continue;
            }
        }
        RunState run{0, 0ull, 0, 0};
        {
            const int kmax = wave_max(key_best);
            if (kmax > 0) { run.best_len = reso * (kmax >> 14); run.best_j = 16383 - (kmax & 16383); }
        }
        // the candidate list goes to the (now free) hot words when it is sure to fit - at most K0 - 2 candidates - so that the
        // gate sums need not be taken before it is known that the read keeps an annotation; else it overwrites the profile in place
        HINGE_K2_STAMP(3);
        const bool cand_apart = K0 - 2 <= HOT * WAVE;
        const K2Const* c = C;
        asm volatile("" : "+s"(c));   // (hides the pointer from the hoisting passes: the loads below stay inside this phase)
        typedef const K2Const __attribute__((address_space(4))) K2ConstK;   // constant address space: scalar loads
        K2ConstK& kc = *(K2ConstK*)(unsigned long long)c;
        int used = __builtin_amdgcn_readfirstlane(mask_gate_annotate(kc.P, reso, MIN_COV, i, lane, K0, run, cov0, covc, cand_apart ? hot : Pq, kc.o, (long long)s, n,
                                                                     true, !cand_apart, flag_words, SPEC != 0 ? band : 0));
        if constexpr (SPEC != 0) {
            if (used & SPEC_DEFERRED) {   // an annotation threshold inside the band: the guard-band list takes the read
                used &= ~SPEC_DEFERRED;
                if (lane == 0) {
                    const unsigned at = atomicAdd(C->redo_count, 1u);
                    if (at < C->redo_cap) C->redo_list[at] = i; else atomicOr(C->o.status, ST_REDO_CAP);
                }
            }
        }
        if (cand_apart && used > 0) {   // the hot words start every read at zero
#pragma unroll
            for (int h = 0; h < HOTW; h++) hot[h * WAVE + lane] = 0;
        }
        HINGE_K2_STAMP(4);
    }
}

// the kernel for ONE part: everything in kernel arguments
template <bool PACKED, bool COVOUT, int CUT20, int SPEC>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_mask_annotate_q20(const K2Const* __restrict__ C, int cut_off_arg, int mulpath_thr, int nhr, int cov_mask_off,
                                                             const int* __restrict__ read_list, int n1, int n2, int n4, const int64_t* __restrict__ row_ptr,
                                                             const typename SpanLoad<PACKED>::raw* __restrict__ a_span, const int* __restrict__ rlen,
                                                             const int* __restrict__ nbins0, const int* __restrict__ d_min_cov, int slot_ints, int* __restrict__ cov_out,
                                                             const long long* __restrict__ cov_off, int* __restrict__ cov_nbins, int cov_base,
                                                             unsigned* __restrict__ heads, int n_heads, K2Heads bases, int* __restrict__ cov_tot, int band_arg) {
    k2_q20_body<PACKED, COVOUT, CUT20, SPEC>((int)blockIdx.x, C, cut_off_arg, mulpath_thr, nhr, cov_mask_off, read_list, n1, n2, n4, row_ptr, a_span, rlen, nbins0, d_min_cov,
                                             slot_ints, cov_out, cov_off, cov_nbins, cov_base, heads, n_heads, bases.base, cov_tot, band_arg);
}
// ... and for up to K2_BATCH_MAX resident parts in ONE launch (round 4): the parts' workgroups lie one range after the other in the
// grid, every part with its share of the persistent workgroups (all resident at once), its own lists, counters and outputs.  Four
// launches pay four ramps, four tails behind a part's slowest reads and three launch boundaries; one launch pays one of each.
// The parts' arguments travel BY VALUE (3 KB of kernel arguments for eight parts: scalar loads, wave-uniform like any argument,
// and nothing to upload between launches although the counters' base values change with every launch).
constexpr int K2_BATCH_MAX = 8;
struct K2Part {
    const K2Const* C;
    const int* read_list; const int64_t* row_ptr; const void* a_span; const int* rlen; const int* nbins0; const int* d_min_cov;
    int* cov_out; const long long* cov_off; int* cov_nbins; unsigned* heads; int* cov_tot;
    int n1, n2, n4, slot_ints, cov_base, n_heads;
    int first_block, n_blocks;      // the part's workgroups are [first_block, first_block + n_blocks) (first_block a multiple of 8: the XCD of a workgroup is the same as in a launch of its own)
    unsigned head_bases[K2_MAX_HEADS];
};
struct K2Batch { int n; int pad; K2Part part[K2_BATCH_MAX]; };
// A part's persistent workgroups do not stop when their own part's short reads run out: wavefront by wavefront they move on to the
// next part of the batch (ring order) and draw from ITS counters - the whole launch has one tail, not one per part at a fraction of
// the chip (without this the batched launch is SLOWER than the launches it replaces: 296 against 4 x 68 us).  Persistent workgroup j
// of part p serves head (j + rot_p - rot_q) mod n_heads_q of part q, rot = (g4 + g2) % 8: the XCD its own deal put it on is the XCD
// of that head's slice of q's span copy.  Every wavefront visits every part and makes exactly one failing draw per visit, so the
// host still knows where each counter stands after the launch.  All parts use ONE slot size (the batch's largest): wavefronts of one
// workgroup work on different parts at the same time.
__host__ __device__ inline int k2_visit_head(int j, int rot_from, int rot_to, int n_heads_to) {
    int m = (j + rot_from - rot_to) % n_heads_to;
    return m < 0 ? m + n_heads_to : m;
}
template <bool PACKED, bool COVOUT, int CUT20, int SPEC>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_mask_annotate_q20_batch(const K2Batch B, int cut_off_arg, int mulpath_thr, int nhr, int cov_mask_off, int band_arg,
                                                                                                             int visit /*1: own part only*/) {
    int p = 0;
    for (int q = 1; q < B.n; q++) p += (int)blockIdx.x >= B.part[q].first_block ? 1 : 0;     // (uniform; n <= 8)
    const int vblock = (int)blockIdx.x - B.part[p].first_block;
    if (vblock >= B.part[p].n_blocks) return;                                                  // (the padding to the next multiple of 8)
    const int fixed_p = B.part[p].n4 + (B.part[p].n2 + 1) / 2;
    const int j = vblock - fixed_p;                                                            // >= 0: a persistent workgroup
    const int first = (visit & 256) && j >= 0 ? 0 : p;      // bit 8: every persistent workgroup starts with part 0 (the chip sweeps one part after the other)
    for (int t = 0; t < (visit & 255); t++) {
        int q = first + t;
        if (q >= B.n) q -= B.n;
        const K2Part& A = B.part[q];
        const int fixed_q = A.n4 + (A.n2 + 1) / 2;
        const int vb = q == p ? vblock : fixed_q + k2_visit_head(j, fixed_p & 7, fixed_q & 7, A.n_heads);
        k2_q20_body<PACKED, COVOUT, CUT20, SPEC>(vb, A.C, cut_off_arg, mulpath_thr, nhr, cov_mask_off, A.read_list, A.n1, A.n2, A.n4, A.row_ptr,
                                                 (const typename SpanLoad<PACKED>::raw*)A.a_span, A.rlen, A.nbins0, A.d_min_cov, A.slot_ints, A.cov_out, A.cov_off, A.cov_nbins,
                                                 A.cov_base, A.heads, A.n_heads, A.head_bases, A.cov_tot, band_arg);
        if (j < 0) return;                                                                     // (a long read's workgroup: that read only)
    }
}

// ------------------------------------------------------------------------------------------------
// Hinge scan over the supporters in their final (std::sort) order.  f[] = other-end coordinate
// mapped so that ascending f is the scan order (abpos for type -1, -aepos for type +1);
// m0 = mask.first resp. -mask.second.  `ord` is an optional indirection (position -> element).
// Returns 1 = bridged, 0 = not bridged.   filter.cpp:920-963 / 1019-1062.
// ------------------------------------------------------------------------------------------------
template <typename OrdT>
HINGE_HD inline int hinge_scan(const int* f, const int* sec, const OrdT* ord, int s, int m0, int BIN, int TH, int UNB, int PIL) {
    int considered = 0, to_end = 0;
    const int f0 = f[ord ? ord[0] : 0];
    for (int id = 0; id < s; ++id) {
        const int x = ord ? ord[id] : id;
        const int fx = f[x];
        if (fx - m0 < BIN) {
            considered++;
            to_end++;
            if ((to_end > UNB) || ((considered > UNB) && (fx - f0 > BIN))) return 0;
        } else if (sec[x] < TH) {
            considered++;
            if ((to_end > UNB) || ((considered > UNB) && (fx - f0 > BIN))) return 0;
        } else if (sec[x] > TH) {
            considered++;
            int id1 = id + 1;
            int pile = 1;
            while (id1 < s) {
                if (f[ord ? ord[id1] : id1] - fx < BIN) { pile++; id1++; }
                else break;
            }
            if (pile > PIL) return 1;
        }
    }
    return 1;   // `bridged` starts true
}

__device__ __forceinline__ void overhangs(int2 bs, int comp, int2 mb, int& L, int& R) {
    // filter.cpp:883-890: overhang of B past the alignment, inside B's mask, on A's left / right
    if (comp == 0) { R = max(mb.y - bs.y, 0); L = max(bs.x - mb.x, 0); }
    else { R = max(bs.x - mb.x, 0); L = max(mb.y - bs.y, 0); }
}

// The 16|16 copy of b_span for the hinge kernels (round 5): bad[0] is set when a coordinate does not fit (the copy is then not used)
__global__ __launch_bounds__(BLOCK) void k_pack_bspan(long long n_ovl, const int2* __restrict__ b_span, unsigned* __restrict__ out, unsigned* __restrict__ bad) {
    bool any = false;
    for (long long k = (long long)blockIdx.x * BLOCK + threadIdx.x; k < n_ovl; k += (long long)gridDim.x * BLOCK) {
        const int2 v = b_span[k];
        any = any || ((unsigned)v.x > 65535u) || ((unsigned)v.y > 65535u);
        out[k] = ((unsigned)v.x & 0xffffu) | ((unsigned)v.y << 16);
    }
    if (__any(any) && lane_id() == 0) atomicOr(bad, 1u);
}

// ------------------------------------------------------------------------------------------------
// K3 exact path: one thread per queued (read, annotation).  Replays the reference literally:
// std::sort of the pile-up by compare_overlap, supporters collected in that order, std::sort by
// pairAscend / pairDescend, scan.  Scratch comes from a bump-allocated global arena.
// ------------------------------------------------------------------------------------------------
__global__ void k_hinge_exact(FilterDev P, const int64_t* __restrict__ row_ptr, const int2* __restrict__ a_span,
                              const int2* __restrict__ b_span, const unsigned* __restrict__ b_flag, const int2* __restrict__ mask,
                              const int2* __restrict__ anno_buf, const unsigned* __restrict__ anno_off,
                              const int2* __restrict__ exact_queue, const unsigned* __restrict__ exact_count, unsigned exact_cap,
                              int* __restrict__ arena, unsigned long long* __restrict__ arena_used, unsigned long long arena_cap,
                              unsigned char* __restrict__ hinge_flag, int* __restrict__ status) {
    const unsigned nq = min(*exact_count, exact_cap);
    for (unsigned q = blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += gridDim.x * blockDim.x) {
        const int i = exact_queue[q].x, a = exact_queue[q].y;
        const int64_t s = row_ptr[i], e = row_ptr[i + 1];
        const int n = (int)(e - s);
        const unsigned long long need = 5ull * (unsigned long long)n;
        const unsigned long long base = atomicAdd(arena_used, need);
        if (base + need > arena_cap) { atomicOr(status, ST_ARENA_CAP); continue; }
        int* perm = arena + base;
        int* key = perm + n;
        int* sf = key + n;
        int* ss = sf + n;
        int* sidx = ss + n;
        for (int k = 0; k < n; k++) {
            const int2 av = a_span[s + k];
            const int2 bs = b_span[s + k];
            key[k] = av.y - av.x + bs.y - bs.x;
            perm[k] = k;
        }
        hinge_sort::std_sort(perm, n, key, 1);   // compare_overlap: descending length sum
        const int2 an = anno_buf[anno_off[i] + a];
        const int pos = an.x, type = an.y;
        int support = 0;
        for (int t = 0; t < n; t++) {
            const int64_t k = s + perm[t];
            const int2 av = a_span[k];
            const int near_c = type == -1 ? av.y : av.x;
            if (!((near_c > pos - P.tol) && (near_c < pos + P.tol))) continue;
            const unsigned bf = b_flag[k];
            const int2 bs = b_span[k];
            const int2 mb = mask[bf & 0x7fffffffu];
            int L, R;
            overhangs(bs, (int)(bf >> 31), mb, L, R);
            if (type == -1) {
                if (R > P.theta) { sf[support] = av.x; ss[support] = L; support++; }
            } else {
                if (L > P.theta) { sf[support] = -av.y; ss[support] = R; support++; }
            }
        }
        unsigned char res = 0;
        if (support > P.sup) {
            for (int t = 0; t < support; t++) sidx[t] = t;
            // pairAscend on abpos == ascending f; pairDescend on aepos == ascending f = -aepos
            hinge_sort::std_sort(sidx, support, sf, 0);
            const int2 mk = mask[i];
            const int m0 = type == -1 ? mk.x : -mk.y;
            const int r = hinge_scan(sf, ss, sidx, support, m0, P.bin_len, P.theta, P.unb, P.pil);
            res = r == 0 ? 1 : 0;
        }
        hinge_flag[anno_off[i] + a] = res;
    }
}

// ------------------------------------------------------------------------------------------------
// Materialised coverage bins of one cutoff for reads r0..r1 (for .coverage.txt / debugging).
// Two modes: count (cov == nullptr) writes nbins only; fill writes bins at out_off[i - r0].
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_coverage_bins(int r0, int r1, const int64_t* __restrict__ row_ptr, const int2* __restrict__ a_span,
                                                         int reso, int cutoff, int kcap, int* __restrict__ nbins,
                                                         const int64_t* __restrict__ out_off, int* __restrict__ cov, int* __restrict__ status) {
    extern __shared__ int lds[];
    const int lane = lane_id();
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform by construction; lets the per-read control flow go scalar
    int* h = lds + (size_t)wib * kcap;
    const int wave = blockIdx.x * WAVES_PER_BLOCK + wib;
    const int nwaves = gridDim.x * WAVES_PER_BLOCK;
    for (int i = r0 + wave; i <= r1; i += nwaves) {
        const int64_t s = row_ptr[i], e = row_ptr[i + 1];
        int mx = INT_MIN;
        for (int64_t k = s + lane; k < e; k += WAVE) {
            const int2 v = a_span[k];
            mx = max(mx, max(v.x + cutoff, v.y - cutoff));
        }
        mx = wave_max(mx);
        const int K = nbins_of<0>((int)(e - s), mx, reso);
        if (lane == 0) nbins[i - r0] = K;
        if (cov == nullptr) continue;
        if (K > kcap) { if (lane == 0) atomicOr(status, ST_RANGE); continue; }
        for (int k = lane; k < K; k += WAVE) h[k] = 0;
        for (int64_t k = s + lane; k < e; k += WAVE) {
            const int2 v = a_span[k];
            atomicAdd(&h[bin_of<0>(v.x + cutoff, reso)], 1);
            atomicAdd(&h[bin_of<0>(v.y - cutoff, reso)], -1);
        }
        int carry = 0;
        const int64_t o = out_off[i - r0];
        for (int base = 0; base < K; base += WAVE) {
            const int j = base + lane;
            int v = j < K ? h[j] : 0;
            v = wave_incl_scan(v) + carry;
            if (j < K) cov[o + j] = v;
            carry = wave_last(v);
        }
    }
}

}  // namespace hinge
