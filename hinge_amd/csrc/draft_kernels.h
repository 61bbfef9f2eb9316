// `hinge draft` on the GPU (second half; reference: src/consensus/draft.cpp:125-715 with lib/DW_banded.c and
// lib/falcon.c under it).  Included by hinge_capi.hip after consensus_kernels.h (the realignment between trace points is the same
// kernel, k_cns_realign: recoverAlignment, LAInterface.cpp:4125-4244).
//
//   k_draft_map     one thread per trace-point segment: its indel list -> get_mapping (draft.cpp:70-87) of the alignment's forward
//                   tags: for every A base of the alignment the number of B bases in front of its column (+ bit 31: its column
//                   holds a gap in B - what the host needs to turn the map around for a strand-1 edge)
//   k_draft_align   one WAVEFRONT per (ladder, member): falcon's banded O(ND) alignment of the member against the ladder's
//                   template (DW_banded.c:97-311) - the diagonals of one d across the lanes, V / U in LDS, the (d, k) records 4 bytes
//                   each in HBM, the trace-back by lane 0, then the alignment tags (falcon.c:68-125, with the leading 'T' column of
//                   draft.cpp:646-655) written run by run, a run's columns across the lanes
//   k_draft_cns     one WAVEFRONT per ladder, one LANE per member: falcon's consensus over the members' tags (falcon.c:246-517) -
//                   the columns (t_pos, delta, base) are visited in the reference's order, the members that share a column vote with
//                   ballots (a link = the member's previous column; equal links are found with readlane + ballot, in member order =
//                   the reference's insertion order, so ties fall the same way), scores as doubled integers (the reference adds
//                   link counts and halves of the coverage: exact in either form), the best-predecessor table in HBM, the
//                   trace-back by lane 0 - including the reference's habit of deciding the LAST base by a link index
//
// Everything is integer work on 2-bit bases; nothing here is GEMM-shaped.  Bound: latency (dependent look-ups along one
// alignment path), hidden by running thousands of ladders' wavefronts side by side.
#pragma once

namespace hinge {

struct DraftSeq { long long boff; int rlen, strand, start, len; };     // bases [start, start + len) of the read in its strand frame

__device__ __forceinline__ int draft_base(const unsigned char* __restrict__ bps, const DraftSeq& s, int x) {
    const int p = s.start + x;
    return s.strand ? 3 - cns_base(bps, s.boff, s.rlen - 1 - p) : cns_base(bps, s.boff, p);
}

// 16 consecutive bases x .. x + 15 of a sequence in ITS frame as one word, the first base in the top two bits (cns_window's form; the
// complemented strand as CnsPair::winB makes it).  What lies behind the sequence's end is whatever follows: callers bound their use.
__device__ __forceinline__ unsigned draft_window(const unsigned char* __restrict__ bps, const DraftSeq& s, int x) {
    const int p0 = s.start + x;
    if (!s.strand) return cns_window(bps, s.boff, p0);
    const int p = s.rlen - 1 - p0;            // base x + t is the complement of read base p - t
    unsigned v;
    if (p >= 15) v = cns_window(bps, s.boff, p - 15);
    else { v = 0; for (int t = 0; t <= p; t++) v |= (unsigned)cns_base(bps, s.boff, t) << (2 * (p - t)); }
    v = __brev(v);
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    return ~v;
}
// words of a staged sequence: bases 16 i .. 16 i + 15 in word i, one spare word behind the last base's word
__host__ __device__ inline int draft_words(int len) { return (len + 15) / 16 + 1; }
// bases x .. x + 15 out of the staged words (x < len)
__device__ __forceinline__ unsigned draft_lds_window(const unsigned* W, int x) {
    const int i = x >> 4;
    const unsigned long long two = ((unsigned long long)W[i] << 32) | W[i + 1];
    return (unsigned)((two << (2 * (x & 15))) >> 32);
}
__device__ __forceinline__ int draft_lds_base(const unsigned* W, int x) { return (int)((W[x >> 4] >> (30 - 2 * (x & 15))) & 3u); }

constexpr unsigned DRAFT_GAP = 0x80000000u;

__global__ __launch_bounds__(CNS_BLOCK) void k_draft_map(const CnsAln* __restrict__ alns, const CnsSeg* __restrict__ segs, int n_seg, const int* __restrict__ indels,
                                                         const int* __restrict__ n_indel, const long long* __restrict__ map_off, unsigned* __restrict__ mapping) {
    const int s = blockIdx.x * CNS_BLOCK + threadIdx.x;
    if (s >= n_seg) return;
    const CnsSeg g = segs[s];
    const CnsAln al = alns[g.aln];
    unsigned* __restrict__ m = mapping + map_off[g.aln];
    cns_walk(g, indels + g.out_off, n_indel[s], [&](int kind, int i, int j, int cnt) {
        if (kind == 0) { for (int t = 0; t < cnt; t++) m[i - 1 + t - al.ab] = (unsigned)(j - 1 + t - al.bb); }
        else if (kind == 2) m[i - 1 - al.ab] = (unsigned)(j - 1 - al.bb) | DRAFT_GAP;
        return true;
    });
}

// ---- falcon's aligner, one wavefront per job ---------------------------------------------------------------------------------
struct DraftJob {
    DraftSeq q, t;              // query = the member, target = the ladder's template
    long long ent_off;          // its (d, k) records: ent_cap words
    long long dtab_off;         // per d: first record, min_k  (2 * (max_d + 1) ints)
    long long tag_off;          // its tags (q.len + t.len + 2 words) ...
    int ent_cap, max_d;
};
constexpr int DRAFT_ST_CAP = 1;      // a record / tag buffer too small (host sizing bug)
constexpr int DRAFT_ST_DELTA = 2;    // a run of 255+ inserted bases: the reference's tags are undefined there (falcon.c:96)
constexpr int DRAFT_ST_BASE = 4;

// a tag: t_pos << 11 | delta << 3 | base (0-3 A C G T, 4 '-')
__device__ __forceinline__ unsigned draft_tag(int t_pos, int delta, int base) { return ((unsigned)t_pos << 11) | ((unsigned)delta << 3) | (unsigned)base; }

__global__ __launch_bounds__(64) void k_draft_align(const unsigned char* __restrict__ bps, const DraftJob* __restrict__ jobs, int n_jobs, int band_tol,
                                                    unsigned* __restrict__ ents, int* __restrict__ dtab, unsigned* __restrict__ tags, int* __restrict__ n_tags,
                                                    int* __restrict__ status) {
    extern __shared__ int lds[];
    const int lane = threadIdx.x;
    // jobs are DRAWN (status[1] is the cursor, zero at launch): an alignment's time follows its edit distance - the template against
    // itself is over at d = 0, a diverged member takes 250 rounds - and with a fixed stride the launch waited for its unluckiest
    // wavefront at twice the mean (SQ_WAVE_CYCLES / waves = 51 % of the launch, profiles/r6f_draft_*)
    while (true) {
        int jb = 0;
        if (lane == 0) jb = atomicAdd(status + 1, 1);
        jb = __builtin_amdgcn_readfirstlane(jb);
        if (jb >= n_jobs) break;
        const DraftJob J = jobs[jb];
        const int q_len = J.q.len, t_len = J.t.len, max_d = J.max_d;
        // V / U as 16-bit cells (x < 32768: a member has fewer bases; x + y < 65536): half the LDS of 32-bit cells, i.e. twice the
        // wavefronts per CU for a kernel whose wavefronts wait half their cycles (a latency chain per round)
        unsigned short* V = reinterpret_cast<unsigned short*>(lds);                 // [2 * max_d + 1]
        unsigned short* U = V + (2 * max_d + 2);
        // Round 6: both sequences staged in LDS, 16 bases per word in their own frame (strand applied once here, not per base), so a
        // snake compares 16 base pairs per step - two LDS words per side, one XOR, one count of leading zeros - instead of two
        // dependent byte loads from global memory per base pair (12 % errors: a snake is ~8 bases, i.e. ONE step).
        unsigned* Wq = reinterpret_cast<unsigned*>(lds + (2 * max_d + 2));
        unsigned* Wt = Wq + draft_words(q_len);
        for (int i = lane; i < 2 * max_d + 2; i += 64) lds[i] = 0;
        for (int i = lane; i < draft_words(q_len); i += 64) Wq[i] = draft_window(bps, J.q, 16 * i);
        for (int i = lane; i < draft_words(t_len); i += 64) Wt[i] = draft_window(bps, J.t, 16 * i);
        __syncthreads();
        const int k_off = max_d, band_size = band_tol * 2;
        unsigned* __restrict__ E = ents + J.ent_off;
        int* __restrict__ DT = dtab + J.dtab_off;
        unsigned* __restrict__ TG = tags + J.tag_off;
        int best_m = -1, min_k = 0, max_k = 0, n_ent = 0;
        int fin_d = -1, fin_k = 0;
        bool overflow = false;
        for (int d = 0; d < max_d; d++) {
            if (max_k - min_k > band_size) break;
            const int nk = (max_k - min_k) / 2 + 1;
            if (n_ent + nk > J.ent_cap) { overflow = true; break; }
            if (lane == 0) { DT[2 * d] = n_ent; DT[2 * d + 1] = min_k; }
            int my_best = -1;
            unsigned long long done_any = 0ull;
            int done_at = 0;
            for (int it = 0; it * 64 < nk && !done_any; it++) {
                const int idx = it * 64 + lane;
                const bool on = idx < nk;
                const int k = min_k + 2 * idx;
                int x = 0, y = 0;
                bool fin = false;
                if (on) {
                    unsigned pre_minus;
                    if (k == min_k || (k != max_k && V[k - 1 + k_off] < V[k + 1 + k_off])) { pre_minus = 0u; x = (int)V[k + 1 + k_off]; }
                    else { pre_minus = 1u; x = (int)V[k - 1 + k_off] + 1; }
                    y = x - k;
                    const int x1 = x;
                    while (true) {          // the snake, 16 base pairs at a time
                        const int rem = min(q_len - x, t_len - y);
                        if (rem <= 0) break;
                        const unsigned df = draft_lds_window(Wq, x) ^ draft_lds_window(Wt, y);
                        const int m = min(df ? (int)(__clz(df) >> 1) : 16, rem);
                        x += m; y += m;
                        if (m < 16) break;
                    }
                    E[n_ent + idx] = ((unsigned)x1 << 17) | (pre_minus << 16) | (unsigned)x;      // x1 (15 bits) | came from k - 1 | x2 (16 bits)
                    fin = x >= q_len || y >= t_len;
                }
                __syncthreads();            // (one wavefront: orders the LDS reads above before the writes below)
                if (on) { V[k + k_off] = (unsigned short)x; U[k + k_off] = (unsigned short)(x + y); my_best = max(my_best, x + y); }
                done_any = __ballot(on && fin);
                if (done_any) done_at = it * 64 + (int)__builtin_ctzll(done_any);
                __syncthreads();
            }
            if (done_any) {                 // the reference breaks at the first k (ascending) that reaches an end
                fin_d = d; fin_k = min_k + 2 * done_at;
                n_ent += done_at + 1;
                break;
            }
            n_ent += nk;
            best_m = max(best_m, wave_max(my_best));          // (DPP reduction; the kernel issues 0.71 of the vector peak: instructions count)
            // the band's new ends: first / last diagonal within band_tol of the best anti-diagonal - one ballot per 64 diagonals, the
            // ends read off the mask on the scalar side (was: two 64-lane shuffle reductions per round)
            int lo = max_k, hi = min_k;
            for (int c = 0; c * 64 < nk; c++) {
                const int idx = c * 64 + lane;
                const bool ok = idx < nk && (int)U[min_k + 2 * idx + k_off] >= best_m - band_tol;
                const unsigned long long Bm = __ballot(ok);
                if (Bm) {
                    lo = min(lo, min_k + 2 * (c * 64 + (int)__builtin_ctzll(Bm)));
                    hi = max(hi, min_k + 2 * (c * 64 + 63 - (int)__builtin_clzll(Bm)));
                }
            }
            max_k = hi + 1; min_k = lo - 1;
            __syncthreads();
        }
        if (overflow && lane == 0) atomicOr(status, DRAFT_ST_CAP);
        // ---- tags: the leading 'T' column, then the path's columns (falcon.c:68-125 on the rows of DW_banded.c:245-300) ---------------
        if (lane == 0) TG[0] = draft_tag(0, 0, 3);
        int n_col = 1;
        if (fin_d >= 0 && !overflow) {
            // trace-back: record index of every d on the path, kept in the V / U space of LDS (no longer needed; d <= 2 * max_d ints)
            int* path = lds;
            __syncthreads();
            if (lane == 0) {
                int ck = fin_k;
                for (int cd = fin_d; cd >= 0; cd--) {
                    const int at = DT[2 * cd] + (ck - DT[2 * cd + 1]) / 2;
                    path[cd] = at;
                    ck = ((E[at] >> 16) & 1u) ? ck - 1 : ck + 1;
                }
            }
            __syncthreads();
            int jj = 0;                     // consecutive inserted bases in front of the next column
            int py = 0;                     // target bases consumed so far
            int px = 0;
            bool bad_delta = false;
            for (int cd = 0; cd <= fin_d; cd++) {
                const unsigned e = E[path[cd]];
                const int x1 = (int)(e >> 17), x2 = (int)(e & 0xffffu);
                const int k = x1 - 0;       // (y1 follows from the step kind below)
                (void)k;
                if (cd > 0) {               // the edit step from (px, py): one column
                    const bool x_step = (e >> 16) & 1u;          // came from k - 1: a query base against a gap
                    if (n_col + 1 > q_len + t_len + 2) { overflow = true; break; }
                    if (x_step) {
                        jj += 1;
                        if (jj >= 255) bad_delta = true;
                        if (lane == 0) TG[n_col] = draft_tag(py, jj & 255, draft_lds_base(Wq, px));
                        px += 1;
                    } else {
                        jj = 0;
                        py += 1;
                        if (lane == 0) TG[n_col] = draft_tag(py, 0, 4);
                    }
                    n_col += 1;
                }
                const int run = x2 - x1;    // the snake: run matched pairs
                if (run > 0) {
                    if (n_col + run > q_len + t_len + 2) { overflow = true; break; }
                    for (int t = lane; t < run; t += 64) TG[n_col + t] = draft_tag(py + 1 + t, 0, draft_lds_base(Wq, px + t));
                    n_col += run; px += run; py += run; jj = 0;
                }
            }
            if (lane == 0 && bad_delta) atomicOr(status, DRAFT_ST_DELTA);
            if (lane == 0 && overflow) atomicOr(status, DRAFT_ST_CAP);
        }
        if (lane == 0) n_tags[jb] = n_col;
        __syncthreads();
    }
}

// ---- falcon's consensus, one wavefront per ladder ---------------------------------------------------------------------------
struct DraftLadder {
    int job0, n;                // its jobs (members), n <= 64
    int t_len;                  // template length + 1
    long long col_off;          // best-predecessor table: col_cap words
    long long tb_off;           // per t: first (t, delta) slot, coverage  (2 * t_len ints)
    long long out_off;          // its consensus: up to 2 * t_len characters
    int col_cap;
};
constexpr unsigned DRAFT_NONE = 0xffffffffu;
constexpr int DRAFT_S2_LDS = 32;     // deltas (consecutive inserted bases in front of a template position) whose column scores live in LDS

__global__ __launch_bounds__(64) void k_draft_cns(const DraftJob* __restrict__ jobs, const DraftLadder* __restrict__ ladders, int n_ladders, const unsigned* __restrict__ tags,
                                                  const int* __restrict__ n_tags, unsigned* __restrict__ cols, int* __restrict__ tbase, char* __restrict__ out,
                                                  int* __restrict__ out_len, unsigned min_cov, int* __restrict__ status, int* __restrict__ s2_far) {
    // doubled scores of the columns of the current and the previous template position.  Round 6: the columns of the first
    // DRAFT_S2_LDS deltas in LDS (1.3 instead of 10 KiB per wavefront: 32 instead of 16 wavefronts per CU for a kernel that is one
    // dependent chain per ladder), the rest - 32+ inserted bases in a row - in a per-workgroup stretch of s2_far
    __shared__ int S2[2][DRAFT_S2_LDS][5];
    int* __restrict__ const S2F = s2_far + (size_t)blockIdx.x * (2 * 256 * 5);
    const int lane = threadIdx.x;
    while (true) {                           // ladders are drawn (status[2], zero at launch): their times differ with their members
        int ld = 0;
        if (lane == 0) ld = atomicAdd(status + 2, 1);
        ld = __builtin_amdgcn_readfirstlane(ld);
        if (ld >= n_ladders) break;
        const DraftLadder L = ladders[ld];
        const bool member = lane < L.n;
        const unsigned* __restrict__ TG = member ? tags + jobs[L.job0 + lane].tag_off : tags;
        const int nt = member ? n_tags[L.job0 + lane] : 0;
        unsigned* __restrict__ C = cols + L.col_off;
        int* __restrict__ TB = tbase + L.tb_off;
        int p = 0;                           // next tag of this lane
        unsigned prev = DRAFT_NONE;          // this lane's previous column (a tag word), DRAFT_NONE in front of its first
        int slot = 0;                        // (t, delta) slots so far
        int g_best = -2, g_ck = 0, g_t = 0;
        unsigned g_col = DRAFT_NONE;         // slot * 5 + base of the best column
        int best_ck = -1;
        bool cap_hit = false;
        for (int t = 0; t < L.t_len; t++) {
            int cov = 0;
            if (lane == 0) TB[2 * t] = slot;
            for (int delta = 0;; delta++) {
                const unsigned tg = p < nt ? TG[p] : DRAFT_NONE;
                const bool has = tg != DRAFT_NONE && (int)(tg >> 11) == t && (int)((tg >> 3) & 255u) == delta;
                const unsigned long long H = __ballot(has);
                if (delta == 0) { cov = __popcll(H); if (lane == 0) TB[2 * t + 1] = cov; }
                if (!H) break;
                if ((slot + 1) * 5 > L.col_cap || delta > 255) { cap_hit = true; break; }
                const int base = (int)(tg & 7u);
                // this lane's predecessor score (doubled); a first tag has none
                int ps = 0;
                if (has && prev != DRAFT_NONE) {
                    const unsigned pt = (prev >> 11) & 1u, pd = (prev >> 3) & 255u, pb = prev & 7u;
                    ps = pd < (unsigned)DRAFT_S2_LDS ? S2[pt][pd][pb] : S2F[(pt * 256 + pd) * 5 + pb];
                }
                for (int kk = 0; kk < 5; kk++) {
                    unsigned long long M = __ballot(has && base == kk);
                    int best = -2;
                    unsigned best_p = 0u;            // (the reference leaves best_p_* of a column without links at zero)
                    int ck = 0;
                    while (M) {
                        const int lead = (int)__builtin_ctzll(M);
                        const unsigned lp = __shfl(prev, lead);
                        const int ls = __shfl(ps, lead);
                        const unsigned long long same = __ballot(has && base == kk && prev == lp);
                        const int score = (lp == DRAFT_NONE ? 0 : ls) + 2 * __popcll(same) - cov;
                        if (score > best) { best = score; best_p = lp; best_ck = ck; }
                        ck++;
                        M &= ~same;
                    }
                    if (lane == 0) {
                        if (delta < DRAFT_S2_LDS) S2[t & 1][delta][kk] = best; else S2F[((t & 1) * 256 + delta) * 5 + kk] = best;
                        C[slot * 5 + kk] = best_p;
                    }
                    if (best > g_best) { g_best = best; g_col = (unsigned)(slot * 5 + kk); g_ck = best_ck; g_t = t; }
                }
                __syncthreads();
                if (has) { prev = tg; p++; }
                slot++;
            }
            if (cap_hit) break;
        }
        if (cap_hit && lane == 0) atomicOr(status, DRAFT_ST_CAP);
        // ---- the sequence, back to front (falcon.c:440-478), then turned around ----------------------------------------------------
        int len = 0;
        if (lane == 0) {
            char* __restrict__ o = out + L.out_off;
            if (g_col != DRAFT_NONE && !cap_hit) {
                char bb = '$';
                int ck = g_ck, i = g_t;
                unsigned col = g_col;
                while (true) {
                    if (ck >= 0 && ck < 5) bb = (unsigned)TB[2 * i + 1] > min_cov ? "ACGT-"[ck] : "acgt-"[ck];
                    const unsigned bp = C[col];
                    if (bp == DRAFT_NONE || len >= 2 * L.t_len) break;
                    i = (int)(bp >> 11);
                    const int j = (int)((bp >> 3) & 255u);
                    ck = (int)(bp & 7u);
                    col = (unsigned)((TB[2 * i] + j) * 5 + ck);
                    if (bb != '-') o[len++] = bb;
                }
                for (int a = 0, b = len - 1; a < b; a++, b--) { const char c = o[a]; o[a] = o[b]; o[b] = c; }
            } else if (!cap_hit) {
                atomicOr(status, DRAFT_ST_BASE);       // (the reference's assert(g_best_score != -1))
            }
            out_len[ld] = len;
        }
        __syncthreads();
    }
}

}  // namespace hinge
