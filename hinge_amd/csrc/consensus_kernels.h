// `hinge consensus` on the GPU: per-contig pile-up vote over base-level realignments.
// Reference: consensus/consensus.cpp:77-288 and, under it, LAInterface::recoverAlignment (lib/LAInterface.cpp:4125-4244) ->
// computeTracePTS (:3410-3506) -> iter_np (:3152-3404), LAInterface::getAlignmentTags (:3709-3905), chop_end (consensus.cpp:27-45).
//
// Roofline: NOT HBM.  The work is Myers' O(np) wave algorithm between successive trace points - a serial recurrence along each
// wave (a diagonal needs its neighbour of the SAME wave) - so the unit of parallelism is the trace-point segment (~100 x ~100
// bases), one LANE per segment, tens of thousands of segments per contig.  Bound by instruction issue and the latency of the
// wave arrays in L2-resident scratch; bytes moved from HBM are a few hundred per segment.
//
// Kernels:
//   k_cns_realign   one lane per segment: forward waves, the reference's trace-back with re-sliding, the indel list
//   k_cns_columns   one thread per alignment: column offsets of its segments, chop_end's start / offset / end
//   k_cns_vote_tiles one workgroup per tile of contig positions: its segments replay their columns and vote with LDS atomics
//                   (k_cns_tile_count / k_cns_tile_fill bin the segments; k_cns_vote: the same with global atomics, the fallback)
//   k_cns_call      one thread per contig position: the reference's base calls (0-2 characters), block sums
//   k_cns_scan      exclusive scan of the block sums;  k_cns_emit  writes the characters at their final offsets
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hinge {

struct CnsSeqs {                   // one DAZZ_DB with bases: 2 bits per base, four per byte, first base in the top bits (DB.c Compress_Read)
    const unsigned char* bps;
    const long long* boff;         // per (trimmed) read: first byte
    const int* rlen;
};
struct CnsAln {
    int a, b, comp;
    int ab, ae, bb, be;            // as in the .las record: B in the complemented frame when comp
    int blen;
    int seg0, nseg;                // its segments
    int dcap;                      // waves its segments may need: the largest recorded `diffs` of its trace (what the reference sizes its arrays from, LAInterface.cpp:3444-3456)
    int tlen;                      // its trace: tlen 16-bit values from toff on ((diffs, B advance) pairs)
    long long toff;
};
struct CnsSeg {
    int aln;
    int a0, m;                     // A bases [a0, a0 + m)
    int b0, n;                     // B bases [b0, b0 + n) (complemented frame when comp)
    unsigned out_off;              // first slot of its indel list
    int out_cap;
};
constexpr int CNS_ST_WAVES = 1;    // a segment needed more waves than its alignment's recorded diffs allow (the reference overruns its arrays there)
constexpr int CNS_ST_INDELS = 2;   // ... or more indel slots
constexpr int CNS_ST_RANGE = 4;    // a coordinate outside its sequence
constexpr int CNS_ST_TRACE = 8;    // trace points inconsistent with the alignment's coordinates (k_cns_segments)

__device__ __forceinline__ int cns_base(const unsigned char* __restrict__ bps, long long boff, int p) {
    const unsigned b = bps[boff + (p >> 2)];
    return (int)((b >> (6 - 2 * (p & 3))) & 3u);
}
// 16 consecutive bases of a packed sequence as one word, the FIRST base in the top two bits: two aligned 32-bit loads around
// the (byte-unaligned) start, composed big-endian.  Reads up to 8 bytes behind the last base's byte (the .bps copy has spare bytes).
__device__ __forceinline__ unsigned cns_window(const unsigned char* __restrict__ bps, long long boff, int x) {
    const unsigned long long at = (unsigned long long)(bps + boff + (x >> 2));
    const unsigned* __restrict__ q = reinterpret_cast<const unsigned*>(at & ~3ull);
    const unsigned lo = q[0], hi = q[1];
    const unsigned long long be = ((unsigned long long)__builtin_bswap32(lo) << 32) | __builtin_bswap32(hi);
    return (unsigned)((be << (8u * (unsigned)(at & 3ull) + 2u * (unsigned)(x & 3))) >> 32);
}
struct CnsPair {                   // the two sequences of one alignment, addressed as the reference's aseq / bseq
    const unsigned char* abps; long long aoff;
    const unsigned char* bbps; long long boff;
    int comp, blen;
    __device__ __forceinline__ int A(int x) const { return cns_base(abps, aoff, x); }
    __device__ __forceinline__ int B(int x) const { return comp ? 3 - cns_base(bbps, boff, blen - 1 - x) : cns_base(bbps, boff, x); }
    // bases x .. x + 15 of aseq / bseq, first base on top (what lies behind a sequence's end is whatever follows it in memory: callers bound their use)
    __device__ __forceinline__ unsigned winA(int x) const { return cns_window(abps, aoff, x); }
    __device__ __forceinline__ unsigned winB(int x) const {
        if (!comp) return cns_window(bbps, boff, x);
        // the complemented strand: bseq[x + t] = 3 - read[blen - 1 - x - t]: the read's bases p - 15 .. p (p = blen - 1 - x), reversed and complemented
        const int p = blen - 1 - x;
        unsigned v;
        if (p >= 15) v = cns_window(bbps, boff, p - 15);
        else { v = 0; for (int t = 0; t <= p; t++) v |= (unsigned)cns_base(bbps, boff, t) << (2 * (p - t)); }   // (near the read's first base: bases 0 .. p at the bottom)
        v = __brev(v);                                                   // group order reversed, and the two bits of every group
        v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);         // ... swapped back
        return ~v;
    }
};

// ---- wave storage of one lane ---------------------------------------------------------------------------------------------
// Row D (= -2, -1, 0, 1, ...) holds the diagonals k in [lo0 - ext(D) - 1, hi0 + ext(D) + 1], ext(D) = max(D, 0) / 2, lo0 = min(0, del),
// hi0 = max(0, del): exactly what iter_np's wave D touches, sentinels included (the reference's rows are tspace + nmax + 3 wide
// and never read outside this range either - checked with a probe in the test restatement of the same loop).  A cell is furthest << 8 | move.
// Layout: cell (D, k) of lane l of a wavefront sits at ((D + 2) * width + (k - klo(D))) * 64 + l of the WAVEFRONT's stretch, `width`
// one row's cells for the whole launch.  The 64 lanes walk their waves in lock step and their columns k - klo(D) differ only by
// their segments' |M - N| - a few cells -, so one wavefront access touches a handful of 256-byte lines instead of 64 (each lane a
// stretch of its own: 64 lines per access, every one an L2 round trip).
struct CnsWaves {
    int* W; int width, lo0;
    __device__ __forceinline__ int klo(int D) const { return lo0 - (D > 0 ? (D >> 1) : 0) - 1; }
    __device__ __forceinline__ int* at(int D, int k) const { return W + (((D + 2) * width + (k - klo(D))) << 6); }
    __device__ __forceinline__ int* row(int D) const { return at(D, 0); }                   // row(D)[k << 6]
    __device__ __forceinline__ int v(int D, int k) const { return *at(D, k) >> 8; }
    __device__ __forceinline__ int h(int D, int k) const { return (int)(signed char)(*at(D, k) & 0xff); }
    __device__ __forceinline__ void set_v(int D, int k, int val) { int* p = at(D, k); *p = (int)((unsigned)val << 8) | (*p & 0xff); }
    __device__ __forceinline__ void set_h(int D, int k, int e) { int* p = at(D, k); *p = (*p & ~0xff) | (e & 0xff); }
};
__host__ __device__ inline int cns_row_width(int dcap, int del_abs) { return del_abs + 2 * ((dcap + 1) / 2) + 3; }   // cells of the widest row (D = dcap)
__host__ __device__ inline long long cns_cells(int dcap, int width) { return (long long)(dcap + 3) * width; }            // rows -2 .. dcap

// iter_np (LAInterface.cpp:3152-3404) for the segment A[a0, a0 + M) x B[b0, b0 + N).  Writes the indel list (1-based absolute
// positions: +B position for a gap in B, -(A position) for a gap in A) to out[0..), returns its length, or -1 / -2 on overflow.
// Round 6: LA / LB (may be nullptr) = the segment's two sequences staged in LDS by the caller, 16 bases per word, word j of this lane
// at L[j * CNS_BLOCK] (bases a0 + 16 j .. of aseq, b0 + 16 j .. of bseq; CNS_LDS_WORDS words each, i.e. M, N <= CNS_LDS_BASES).  The
// slide then reads two LDS words per side instead of two global words per side for every 16 base pairs it compares, and the
// trace-back's re-slides single bases out of the same words.
#ifndef HINGE_CNS_LDS_WORDS
#define HINGE_CNS_LDS_WORDS 12
#endif
constexpr int CNS_LDS_WORDS = HINGE_CNS_LDS_WORDS;      // 12 words per side: segments of up to 176 bases, 24 KiB per workgroup (9 / 10 / 12 words measure the same: profiles/EXPERIMENTS.md C6)
constexpr int CNS_LDS_BASES = 16 * (CNS_LDS_WORDS - 1);
__device__ __forceinline__ unsigned cns_lds_window(const unsigned* L, int x) {
    const unsigned* q = L + (x >> 4) * 256;
    const unsigned long long two = ((unsigned long long)q[0] << 32) | q[256];
    return (unsigned)((two << (2 * (x & 15))) >> 32);
}
__device__ __forceinline__ int cns_lds_base(const unsigned* L, int x) { return (int)((L[(x >> 4) * 256] >> (30 - 2 * (x & 15))) & 3u); }

__device__ inline int cns_iter_np(const CnsPair& S, int a0, int M, int b0, int N, CnsWaves w, int dcap, int* __restrict__ out, int out_cap, int& n_ins,
                                  const unsigned* LA = nullptr, const unsigned* LB = nullptr) {
    const int del = M - N;
    int low = del >= 0 ? 0 : del, hgh = del >= 0 ? del : 0;
    w.lo0 = low;
    {
        int* r2 = w.row(-2); int* r1 = w.row(-1);
        for (int k = low - 1; k <= hgh + 1; k++) { r2[k << 6] = -512; r1[k << 6] = -512; }
        r1[0] = -256;
    }
    low += 1; hgh -= 1;
    int D;
    for (D = 0;; D++) {
        if (D > dcap) return -1;
        if ((D & 1) == 0) { low -= 1; hgh += 1; }
        int* __restrict__ F0 = w.row(D);
        const int* __restrict__ F1 = w.row(D - 1);
        const int* __restrict__ F2 = w.row(D - 2);
        F0[(hgh + 1) << 6] = -512; F0[(low - 1) << 6] = -512;
        auto move = [&](int k, int am, int ap, int mdir, int pdir) {
            const int ac = (F1[k << 6] >> 8) + 1;
            int j, hc;
            if (ac < am) { if (ap < am) { hc = mdir; j = am; } else { hc = pdir; j = ap; } }
            else { if (ap < ac) { hc = 0; j = ac; } else { hc = pdir; j = ap; } }
            const int i = M - k;
            const int lim = N < i ? N : i;
            // (j >= 0 always: every diagonal of wave D is reachable from (0, 0) - the oracle counts the exceptions: none)
            // the slide, 16 bases per step: XOR of the two packed windows, the leading equal pairs counted
            if (j >= 0)
                while (j < lim) {
                    const unsigned x = LA ? (cns_lds_window(LA, j + k) ^ cns_lds_window(LB, j)) : (S.winA(a0 + j + k) ^ S.winB(b0 + j));
                    const int eq = x ? (__clz((int)x) >> 1) : 16;
                    const int room = lim - j;
                    j += eq < room ? eq : room;
                    if (eq < 16) break;
                }
            F0[k << 6] = (int)((unsigned)j << 8) | (hc & 0xff);
            return j;
        };
        int j = -2;
        for (int k = hgh; k > del; k--) j = move(k, F2[(k - 1) << 6] >> 8, j + 1, -1, 4);
        j = -2;
        for (int k = low; k < del; k++) j = move(k, j, (F2[(k + 1) << 6] >> 8) + 1, 2, 1);
        j = move(del, j, (F0[(del + 1) << 6] >> 8) + 1, 2, 4);
        if (j >= N) break;
    }
    // trace-back with re-sliding (LAInterface.cpp:3285-3352)
    {
        w.set_h(0, 0, 3);
        int c = N, k = del;
        int e = w.h(D, k);
        w.set_h(D, k, 3);
        while (e != 3) {
            int h = k + e;
            if (e > 1) h -= 3;
            else if (e == 0) D -= 1;
            else D -= 2;
            if (h < k) {
                int m = k < 0 ? -k : 0;
                const int vh = w.v(D, h);
                if (vh <= c) c = vh - 1;
                if (LA) { while (c >= m && cns_lds_base(LA, c + k) == cns_lds_base(LB, c)) c -= 1; }
                else while (c >= m && S.A(a0 + c + k) == S.B(b0 + c)) c -= 1;
                if (e < 1) {
                    if (c <= w.v(D + 2, k + 1)) { e = 4; h = k + 1; D = D + 2; }
                    else if (c == w.v(D + 1, k)) { e = 0; h = k; D = D + 1; }
                    else w.set_v(D, h, c + 1);
                } else {
                    m = (k == del) ? D : D - 2;
                    if (c <= w.v(m, k + 1)) { e = (k == del) ? 4 : 1; h = k + 1; D = m; }
                    else if (c == w.v(D - 1, k)) { e = 0; h = k; D = D - 1; }
                    else w.set_v(D, h, c + 1);
                }
            }
            const int m2 = w.h(D, h);
            w.set_h(D, h, e);
            e = m2;
            k = h;
        }
    }
    // forward along the reversed chain: one entry per indel (LAInterface.cpp:3354-3371)
    int cnt = 0;
    n_ins = 0;
    {
        const int ap = -a0 - 1, bp = b0 + 1;
        int k = 0, DD = 0;
        int e = w.h(DD, k);
        while (e != 3) {
            int h = k - e;
            const int c = w.v(DD, k);
            if (e > 1) h += 3;
            else if (e == 0) DD += 1;
            else DD += 2;
            if (h != k) {
                if (cnt >= out_cap) return -2;
                if (h > k) out[cnt++] = bp + c;
                else { out[cnt++] = ap - (c + k); n_ins++; }
            }
            k = h;
            e = w.h(DD, h);
        }
    }
    return cnt;
}

constexpr int CNS_BLOCK = 256;

__global__ __launch_bounds__(CNS_BLOCK) void k_cns_realign(CnsSeqs SA, CnsSeqs SB, const CnsAln* __restrict__ alns, const CnsSeg* __restrict__ segs, int n_seg,
                                                           int* __restrict__ scratch, int row_width, int rows, int* __restrict__ indels,
                                                           int* __restrict__ n_indel, int* __restrict__ n_ins_out, int* __restrict__ status) {
    const long long lane_g = (long long)blockIdx.x * CNS_BLOCK + threadIdx.x;
    const long long n_lanes = (long long)gridDim.x * CNS_BLOCK;
    static_assert(CNS_BLOCK == 256, "cns_lds_window's word stride");
    __shared__ unsigned LW[2][CNS_LDS_WORDS][CNS_BLOCK];   // 2 x CNS_LDS_WORDS KiB: this lane's segment, both sequences, 16 bases per word (lane-private columns: no barrier)
    CnsWaves w;
    w.W = scratch + (lane_g >> 6) * ((long long)rows * row_width * 64) + (lane_g & 63); w.width = row_width; w.lo0 = 0;
    for (long long s = lane_g; s < n_seg; s += n_lanes) {
        const CnsSeg g = segs[s];
        const CnsAln al = alns[g.aln];
        CnsPair S;
        S.abps = SA.bps; S.aoff = SA.boff[al.a]; S.bbps = SB.bps; S.boff = SB.boff[al.b]; S.comp = al.comp; S.blen = al.blen;
        const int del_abs = g.m >= g.n ? g.m - g.n : g.n - g.m;
        int dcap = al.dcap;
        if (cns_row_width(dcap, del_abs) > row_width || dcap + 3 > rows) { atomicOr(status, CNS_ST_WAVES); n_indel[s] = 0; n_ins_out[s] = 0; continue; }
        int nins = 0;
        // the segment's bases to LDS (independent loads, issued together) unless it is longer than the staging area
        const bool staged = g.m <= CNS_LDS_BASES && g.n <= CNS_LDS_BASES;
        if (staged) {
            const int na = (g.m + 15) / 16 + 1, nb = (g.n + 15) / 16 + 1;
            for (int j = 0; j < na; j++) LW[0][j][threadIdx.x] = S.winA(g.a0 + 16 * j);
            for (int j = 0; j < nb; j++) LW[1][j][threadIdx.x] = S.winB(g.b0 + 16 * j);
        }
        const int cnt = cns_iter_np(S, g.a0, g.m, g.b0, g.n, w, dcap, indels + g.out_off, g.out_cap, nins,
                                    staged ? &LW[0][0][threadIdx.x] : (const unsigned*)nullptr, staged ? &LW[1][0][threadIdx.x] : (const unsigned*)nullptr);
        if (cnt < 0) { atomicOr(status, cnt == -1 ? CNS_ST_WAVES : CNS_ST_INDELS); n_indel[s] = 0; n_ins_out[s] = 0; continue; }
        n_indel[s] = cnt;
        n_ins_out[s] = nins;
    }
}

// The columns of a segment, as getAlignmentTags lays them out (LAInterface.cpp:3822-3866): runs of aligned pairs separated by
// single gap columns.  f(kind, i, j, cnt): kind 0 = cnt aligned pairs starting at A position i, B position j (1-based);
// 1 = one column with a gap in A (B base j);  2 = one column with a gap in B (A base i).  f returns false to stop.
template <typename F>
__device__ __forceinline__ void cns_walk(const CnsSeg& g, const int* __restrict__ ind, int cnt, F f) {
    int i = g.a0 + 1, j = g.b0 + 1;
    for (int t = 0; t < cnt; t++) {
        const int p = ind[t];
        if (p < 0) {
            const int run = -p - i;
            if (run > 0) { if (!f(0, i, j, run)) return; i += run; j += run; }
            if (!f(1, i, j, 1)) return;
            j += 1;
        } else {
            const int run = p - j;
            if (run > 0) { if (!f(0, i, j, run)) return; i += run; j += run; }
            if (!f(2, i, j, 1)) return;
            i += 1;
        }
    }
    const int run = g.a0 + g.m + 1 - i;
    if (run > 0) f(0, i, j, run);
}

struct CnsCols { int start, end, offset; };   // columns [start, end) of the alignment vote; offset = chop_end's return value

// one thread per alignment (an alignment has ~100 segments; the walk to column `chop` touches two or three of them)
__global__ __launch_bounds__(CNS_BLOCK) void k_cns_columns(const CnsAln* __restrict__ alns, int n_aln, const CnsSeg* __restrict__ segs, const int* __restrict__ indels,
                                                           const int* __restrict__ n_indel, const int* __restrict__ n_ins, int* __restrict__ col_base,
                                                           CnsCols* __restrict__ cols, int chop) {
    const int x = blockIdx.x * CNS_BLOCK + threadIdx.x;
    if (x >= n_aln) return;
    const CnsAln al = alns[x];
    int len = 0;
    for (int s = al.seg0; s < al.seg0 + al.nseg; s++) { col_base[s] = len; len += segs[s].m + n_ins[s]; }
    CnsCols c;
    c.start = 0; c.end = len; c.offset = 0;
    if (len >= chop * 2 + 10) {   // chop_end (consensus.cpp:27-45): the first non-gap column of A at or behind column `chop`
        int col = 0, abases = 0, start = -1;
        for (int s = al.seg0; s < al.seg0 + al.nseg && start < 0; s++) {
            const CnsSeg g = segs[s];
            cns_walk(g, indels + g.out_off, n_indel[s], [&](int kind, int, int, int cnt) {
                if (kind == 1) { col += 1; return true; }                 // a gap in A is never the start
                if (col + cnt > chop) {                                     // the run reaches column `chop` or lies behind it
                    const int skip = chop > col ? chop - col : 0;
                    start = col + skip; abases += skip;
                    return false;
                }
                col += cnt; abases += cnt;
                return true;
            });
        }
        c.start = start; c.offset = abases; c.end = len - chop;
    }
    cols[x] = c;
}

// counters: nine int32 planes over the concatenated contig positions: A C G T '-' of the aligned columns, A C G T of the
// inserted ones (consensus.cpp:163-212: contig_base_scores, insertion_base_scores; cov_depth and insertion_score are their sums)
__global__ __launch_bounds__(CNS_BLOCK) void k_cns_vote(CnsSeqs SB, const CnsAln* __restrict__ alns, const CnsSeg* __restrict__ segs, int n_seg,
                                                        const int* __restrict__ indels, const int* __restrict__ n_indel, const int* __restrict__ col_base,
                                                        const CnsCols* __restrict__ cols, const long long* __restrict__ cbase, int* __restrict__ counts,
                                                        long long plane) {
    const int s = blockIdx.x * CNS_BLOCK + threadIdx.x;
    if (s >= n_seg) return;
    const CnsSeg g = segs[s];
    const CnsAln al = alns[g.aln];
    const CnsCols c = cols[g.aln];
    const unsigned char* __restrict__ bbps = SB.bps;
    const long long boff = SB.boff[al.b];
    auto Bb = [&](int j1) { return al.comp ? 3 - cns_base(bbps, boff, al.blen - j1) : cns_base(bbps, boff, j1 - 1); };   // 1-based position
    int* __restrict__ cnt0 = counts + cbase[al.a];
    const long long alen_a = cbase[al.a + 1] - cbase[al.a];
    int col = col_base[s];
    if (col >= c.end || col + g.m + 2 * g.out_cap < c.start) return;   // (cheap reject; the exact test is per column)
    cns_walk(g, indels + g.out_off, n_indel[s], [&](int kind, int i, int j, int cnt) {
        if (kind == 0) {
            int lo = c.start > col ? c.start - col : 0;
            int hi = c.end - col < cnt ? c.end - col : cnt;
            for (int t = lo; t < hi; t++) atomicAdd(cnt0 + (long long)Bb(j + t) * plane + (i - 1 + t), 1);
            col += cnt;
        } else {
            if (col >= c.start && col < c.end) {
                // (an inserted base behind the contig's LAST base - i - 1 == alen, only where an unchopped alignment ends at the
                // contig's end - has no position: the reference indexes insertion_score[alen] out of bounds there, the tiled vote
                // drops the slot, and so does this one instead of voting on the next contig's first position)
                if (kind == 1) { if (i - 1 < alen_a) atomicAdd(cnt0 + (long long)(5 + Bb(j)) * plane + (i - 1), 1); }
                else atomicAdd(cnt0 + 4ll * plane + (i - 1), 1);
            }
            col += 1;
        }
        return col < c.end;
    });
}

// ---- the vote with the counters in LDS (the default) -----------------------------------------------------------------------
// The contigs are cut into tiles of CNS_TILE(tspace) positions, a multiple of tspace: a segment never straddles a multiple of
// tspace (computeTracePTS cuts there), so every vote of a segment lands in ITS tile - except an inserted base behind the
// segment's last A base, whose position is the first one of the next tile: the halo slot, written to `halo` and added by
// k_cns_call.  One workgroup per tile: the tile's segments (binned by k_cns_tile_count / k_cns_tile_fill) vote with LDS
// atomics into 16|16-packed counters (a contig with 65 536+ voting alignments takes the global-atomics kernel above), then
// the workgroup stores its tile's nine planes with plain, coalesced stores: no global atomic, no memset of the planes.
__host__ __device__ inline int cns_tile_len(int tspace) { return tspace >= 2048 ? tspace : (2048 / tspace) * tspace; }
constexpr int CNS_TILE_MAX = 4096;       // positions a tile may have (LDS: 5 words per position)

// One thread per ALIGNMENT: its segments ascend along the contig, so the segments of one tile are a run - one atomic per run
// (a segment-per-thread form took 1.33 M atomics on ~2 k words: 0.30 + 0.44 ms; this one ~0.1 M).
__global__ __launch_bounds__(CNS_BLOCK) void k_cns_tile_count(const CnsAln* __restrict__ alns, int n_aln, const CnsSeg* __restrict__ segs, const int* __restrict__ tile_base,
                                                              int tile, unsigned* __restrict__ tile_cnt) {
    const int x = blockIdx.x * CNS_BLOCK + threadIdx.x;
    if (x >= n_aln) return;
    const CnsAln al = alns[x];
    const int tb = tile_base[al.a];
    int cur = -1, run = 0;
    for (int s = al.seg0; s < al.seg0 + al.nseg; s++) {
        const int t = tb + segs[s].a0 / tile;
        if (t != cur) { if (run) atomicAdd(&tile_cnt[cur], (unsigned)run); cur = t; run = 0; }
        run++;
    }
    if (run) atomicAdd(&tile_cnt[cur], (unsigned)run);
}
// tile_ptr = exclusive scan of the counts (k_cns_scan); cursor starts as a copy of it
__global__ __launch_bounds__(CNS_BLOCK) void k_cns_tile_fill(const CnsAln* __restrict__ alns, int n_aln, const CnsSeg* __restrict__ segs, const int* __restrict__ tile_base,
                                                             int tile, unsigned* __restrict__ cursor, int* __restrict__ order) {
    const int x = blockIdx.x * CNS_BLOCK + threadIdx.x;
    if (x >= n_aln) return;
    const CnsAln al = alns[x];
    const int tb = tile_base[al.a];
    int cur = -1, first = al.seg0;
    auto flush = [&](int end) {
        if (cur < 0 || end == first) return;
        const unsigned at = atomicAdd(&cursor[cur], (unsigned)(end - first));
        for (int s = first; s < end; s++) order[at + (unsigned)(s - first)] = s;
    };
    for (int s = al.seg0; s < al.seg0 + al.nseg; s++) {
        const int t = tb + segs[s].a0 / tile;
        if (t != cur) { flush(s); cur = t; first = s; }
    }
    flush(al.seg0 + al.nseg);
}

__global__ __launch_bounds__(CNS_BLOCK) void k_cns_vote_tiles(CnsSeqs SB, const CnsAln* __restrict__ alns, const CnsSeg* __restrict__ segs, const int* __restrict__ indels,
                                                              const int* __restrict__ n_indel, const int* __restrict__ col_base, const CnsCols* __restrict__ cols,
                                                              const long long* __restrict__ cbase, const int* __restrict__ tile_base, const int* __restrict__ contig_of_tile,
                                                              const unsigned* __restrict__ tile_ptr, const int* __restrict__ order, int tile, int* __restrict__ counts,
                                                              long long plane, int* __restrict__ halo) {
    extern __shared__ unsigned cnt_lds[];            // [tile + 1][5]: A|C, G|T, '-', iA|iC, iG|iT (16 bits each)
    const int t = blockIdx.x;
    const int cg = contig_of_tile[t];
    const int p0 = (t - tile_base[cg]) * tile;       // first position of the tile in its contig
    const long long base = cbase[cg];
    const int alen = (int)(cbase[cg + 1] - base);
    const int np = min(tile, alen - p0);
    for (int x = threadIdx.x; x < (tile + 1) * 5; x += CNS_BLOCK) cnt_lds[x] = 0u;
    __syncthreads();
    const unsigned s_lo = tile_ptr[t], s_hi = tile_ptr[t + 1];
    const unsigned char* __restrict__ bbps = SB.bps;
    for (unsigned x = s_lo + threadIdx.x; x < s_hi; x += CNS_BLOCK) {
        const int s = order[x];
        const CnsSeg g = segs[s];
        const CnsAln al = alns[g.aln];
        const CnsCols c = cols[g.aln];
        const long long boff = SB.boff[al.b];
        auto Bb = [&](int j1) { return al.comp ? 3 - cns_base(bbps, boff, al.blen - j1) : cns_base(bbps, boff, j1 - 1); };
        CnsPair SP;                                  // (only its B side is used: bseq[x] = Bb(x + 1))
        SP.abps = bbps; SP.aoff = 0; SP.bbps = bbps; SP.boff = boff; SP.comp = al.comp; SP.blen = al.blen;
        auto vote = [&](int pos, int slot) {         // slot 0-3 aligned base, 4 '-', 5-8 inserted base
            const int w = slot < 4 ? (slot >> 1) : slot == 4 ? 2 : 3 + ((slot - 5) >> 1);
            const unsigned inc = (slot < 4 ? (slot & 1) : slot == 4 ? 0 : ((slot - 5) & 1)) ? 0x10000u : 1u;
            atomicAdd(&cnt_lds[(pos - p0) * 5 + w], inc);
        };
        int col = col_base[s];
        if (col >= c.end || col + g.m + g.out_cap < c.start) continue;
        cns_walk(g, indels + g.out_off, n_indel[s], [&](int kind, int i, int j, int cnt) {
            if (kind == 0) {
                const int lo = c.start > col ? c.start - col : 0;
                const int hi = c.end - col < cnt ? c.end - col : cnt;
                // (round 6) the run's B bases 16 at a time out of one packed window instead of one byte load per base
                for (int u = lo; u < hi;) {
                    const unsigned w = SP.winB(j - 1 + u);
                    const int m = hi - u < 16 ? hi - u : 16;
                    for (int t = 0; t < m; t++) vote(i - 1 + u + t, (int)((w >> (30 - 2 * t)) & 3u));
                    u += m;
                }
                col += cnt;
            } else {
                if (col >= c.start && col < c.end) vote(i - 1, kind == 1 ? 5 + Bb(j) : 4);
                col += 1;
            }
            return col < c.end;
        });
    }
    __syncthreads();
    int* __restrict__ out = counts + base + p0;
    for (int p = threadIdx.x; p < np; p += CNS_BLOCK) {
        const unsigned w0 = cnt_lds[p * 5], w1 = cnt_lds[p * 5 + 1], w2 = cnt_lds[p * 5 + 2], w3 = cnt_lds[p * 5 + 3], w4 = cnt_lds[p * 5 + 4];
        out[p] = (int)(w0 & 0xffff); out[plane + p] = (int)(w0 >> 16);
        out[2 * plane + p] = (int)(w1 & 0xffff); out[3 * plane + p] = (int)(w1 >> 16);
        out[4 * plane + p] = (int)(w2 & 0xffff);
        out[5 * plane + p] = (int)(w3 & 0xffff); out[6 * plane + p] = (int)(w3 >> 16);
        out[7 * plane + p] = (int)(w4 & 0xffff); out[8 * plane + p] = (int)(w4 >> 16);
    }
    if (threadIdx.x < 4) {   // the halo slot: inserted bases in front of the NEXT tile's first position
        const unsigned w = cnt_lds[tile * 5 + 3 + (threadIdx.x >> 1)];
        halo[(long long)t * 4 + threadIdx.x] = (int)((threadIdx.x & 1) ? (w >> 16) : (w & 0xffff));
    }
}

struct CnsStats { long long sum_cov; int good, insertions, deletions, low_cov, clen, pad; };

// one thread per contig position (consensus.cpp:228-270): packed = count | c0 << 8 | c1 << 16; per-block character counts
constexpr int CNS_CALL_ITEMS = 8;   // positions per thread in k_cns_call / k_cns_emit (a block covers CNS_BLOCK * CNS_CALL_ITEMS)
__global__ __launch_bounds__(CNS_BLOCK) void k_cns_call(CnsSeqs SA, const int* __restrict__ counts, long long plane, long long n_pos,
                                                        const int2* __restrict__ contig_of_block /*(contig, its first block)*/, const long long* __restrict__ cbase,
                                                        unsigned* __restrict__ packed, unsigned* __restrict__ block_sum, CnsStats* __restrict__ stats,
                                                        const int* __restrict__ halo /*nullptr: the global-atomics vote*/, const int* __restrict__ tile_base, int tile) {
    // blocks never straddle contigs: block b works on contig contig_of_block[b], positions from its own first position on
    __shared__ int red[8];
    __shared__ long long redl;
    const int2 cb = contig_of_block[blockIdx.x];
    const int cg = cb.x, b0 = cb.y;   // (the contig's first block comes from the host's table: a walk back over the block list is
                                      // quadratic per contig - 2.4e9 dependent loads for a 100 Mb contig)
    const long long base = cbase[cg], alen = cbase[cg + 1] - base;
    const long long first = (long long)(blockIdx.x - b0) * CNS_BLOCK * CNS_CALL_ITEMS;
    const unsigned char* __restrict__ abps = SA.bps;
    const long long aoff = SA.boff[cg];
    int chars = 0, good = 0, ins = 0, dels = 0, low = 0;
    long long sum = 0;
    for (int u = 0; u < CNS_CALL_ITEMS; u++) {
        const long long j = first + (long long)u * CNS_BLOCK + threadIdx.x;
        if (j >= alen) continue;
        const long long gp = base + j;
        int sc[5], ib[4];
#pragma unroll
        for (int b = 0; b < 5; b++) sc[b] = counts[(long long)b * plane + gp];
#pragma unroll
        for (int b = 0; b < 4; b++) ib[b] = counts[(long long)(5 + b) * plane + gp];
        if (halo && j > 0 && j % tile == 0) {        // inserted bases the previous tile's segments put in front of this position
            const int* __restrict__ h = halo + (long long)(tile_base[cg] + (int)(j / tile) - 1) * 4;
#pragma unroll
            for (int b = 0; b < 4; b++) ib[b] += h[b];
        }
        const int depth = sc[0] + sc[1] + sc[2] + sc[3] + sc[4];
        const int iscore = ib[0] + ib[1] + ib[2] + ib[3];
        sum += depth;
        unsigned pk;
        if (depth < 3) {
            low++;
            pk = 1u | ((unsigned)("acgt"[cns_base(abps, aoff, (int)j)]) << 8);
        } else {
            int n = 0; unsigned c0 = 0, c1 = 0;
            if (iscore > depth / 2) {
                int mb = 0;
                for (int b = 1; b < 4; b++) if (ib[b] > ib[mb]) mb = b;
                c0 = (unsigned)"ACGT"[mb]; n = 1; ins++;
            }
            int mb = 0;
            for (int b = 1; b < 5; b++) if (sc[b] > sc[mb]) mb = b;
            if (mb < 4) { if (n == 0) c0 = (unsigned)"ACGT"[mb]; else c1 = (unsigned)"ACGT"[mb]; n++; good++; }
            else dels++;
            pk = (unsigned)n | (c0 << 8) | (c1 << 16);
        }
        packed[gp] = pk;
        chars += (int)(pk & 0xff);
    }
    // block reductions (six small sums): wave shuffles, then the four waves through LDS
    auto wsum = [&](int v) { for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d); return v; };
    auto wsuml = [&](long long v) { for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d); return v; };
    if (threadIdx.x < 8) red[threadIdx.x] = 0;
    if (threadIdx.x == 0) redl = 0;
    __syncthreads();
    const int v0 = wsum(chars), v1 = wsum(good), v2 = wsum(ins), v3 = wsum(dels), v4 = wsum(low);
    const long long v5 = wsuml(sum);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&red[0], v0); atomicAdd(&red[1], v1); atomicAdd(&red[2], v2); atomicAdd(&red[3], v3); atomicAdd(&red[4], v4);
        atomicAdd((unsigned long long*)&redl, (unsigned long long)v5);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        block_sum[blockIdx.x] = (unsigned)red[0];
        CnsStats* st = stats + cg;
        atomicAdd(&st->good, red[1]); atomicAdd(&st->insertions, red[2]); atomicAdd(&st->deletions, red[3]); atomicAdd(&st->low_cov, red[4]);
        atomicAdd(&st->clen, red[1] + red[2]);
        atomicAdd((unsigned long long*)&st->sum_cov, (unsigned long long)redl);
    }
}

// The segment table (computeTracePTS's loop bounds, LAInterface.cpp:3470-3500) on the device (round 5; the host built and uploaded it
// before: 1.33 M segments = 37 MB over PCIe and 5 ms of a single thread per call, more than the kernels took).  One thread per
// alignment, two walks over its trace: the largest recorded `diffs` (-> dcap), then the segments [ab, ae) x [bb, be) - A advances to
// the next multiple of tspace, B by the trace's advance, the last segment ends at (aepos, bepos).  A segment's indel slots are
// dcap + |m - n|; their per-alignment sums go to aln_slots for the scan that k_cns_seg_offsets turns into out_off.
// gmax: [0] widest row of the launch, [1] largest dcap (both size k_cns_realign's wave storage: the host reads them back).
__global__ __launch_bounds__(CNS_BLOCK) void k_cns_segments(CnsAln* __restrict__ alns, int n_aln, const unsigned short* __restrict__ trace, int tspace,
                                                           const int* __restrict__ rlen_a, CnsSeg* __restrict__ segs, unsigned* __restrict__ aln_slots,
                                                           int* __restrict__ gmax, int* __restrict__ status) {
    const int x = blockIdx.x * CNS_BLOCK + threadIdx.x;
    if (x >= n_aln) return;
    CnsAln al = alns[x];
    const unsigned short* __restrict__ pts = trace + al.toff;
    const int alen = rlen_a[al.a], blen = al.blen;
    int dmax = 0;
    for (int d = 0; d < al.tlen; d += 2) dmax = max(dmax, (int)pts[d]);
    alns[x].dcap = dmax;
    int wmax = 4, at = al.seg0;
    unsigned slots = 0;
    bool ok = true;
    auto add = [&](int ab, int ae, int bb, int be) {
        if (ae < ab || be < bb || ae > alen || be > blen) { ok = false; return; }
        CnsSeg g;
        g.aln = x; g.a0 = ab; g.m = ae - ab; g.b0 = bb; g.n = be - bb;
        const int del_abs = abs(g.m - g.n);
        g.out_cap = dmax + del_abs;
        g.out_off = 0u;
        slots += (unsigned)g.out_cap;
        wmax = max(wmax, cns_row_width(dmax, del_abs));
        segs[at++] = g;
    };
    int ab = al.ab, ae = (ab / tspace) * tspace, bb = al.bb;
    const int tl = al.tlen - 2;
    for (int i = 1; i < tl && ok; i += 2) {
        ae += tspace;
        const int be = bb + (int)pts[i];
        add(ab, ae, bb, be);
        ab = ae; bb = be;
    }
    if (ok) add(ab, al.ae, bb, al.be);
    if (!ok) {   // the rest of its slots stay empty segments: nothing of a refused call is used
        atomicOr(status, CNS_ST_TRACE);
        for (; at < al.seg0 + al.nseg; at++) { CnsSeg g; g.aln = x; g.a0 = al.ab; g.m = 0; g.b0 = al.bb; g.n = 0; g.out_off = 0u; g.out_cap = 0; segs[at] = g; }
        slots = 0;
    }
    aln_slots[x] = slots;
    atomicMax(&gmax[0], wmax);
    atomicMax(&gmax[1], dmax);
}
// out_off of every segment: the alignment's base (the exclusive scan of aln_slots) + the slots of its earlier segments
__global__ __launch_bounds__(CNS_BLOCK) void k_cns_seg_offsets(const CnsAln* __restrict__ alns, int n_aln, const unsigned* __restrict__ aln_base, CnsSeg* __restrict__ segs) {
    const int x = blockIdx.x * CNS_BLOCK + threadIdx.x;
    if (x >= n_aln) return;
    const int s0 = alns[x].seg0, s1 = s0 + alns[x].nseg;
    unsigned run = aln_base[x];
    for (int s = s0; s < s1; s++) { segs[s].out_off = run; run += (unsigned)segs[s].out_cap; }
}

// exclusive scan of n values in place, one workgroup of 1024 threads; total to *total
__global__ __launch_bounds__(1024) void k_cns_scan(unsigned* __restrict__ v, int n, unsigned long long* __restrict__ total) {
    __shared__ unsigned long long part[1024];
    const int per = (n + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = min(lo + per, n);
    unsigned long long s = 0;
    for (int k = lo; k < hi; k++) s += v[k];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        unsigned long long t = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += t;
        __syncthreads();
    }
    unsigned long long run = part[threadIdx.x] - s;
    for (int k = lo; k < hi; k++) { const unsigned x = v[k]; v[k] = (unsigned)run; run += x; }
    if (threadIdx.x == 1023) *total = part[1023];
}

// characters to their final places: block b's output starts at block_off[b] (exclusive scan of the block sums)
__global__ __launch_bounds__(CNS_BLOCK) void k_cns_emit(const unsigned* __restrict__ packed, const int2* __restrict__ contig_of_block, const long long* __restrict__ cbase,
                                                        const unsigned* __restrict__ block_off, char* __restrict__ out) {
    __shared__ unsigned wsum_[4];
    const int2 cb = contig_of_block[blockIdx.x];
    const int cg = cb.x, b0 = cb.y;
    const long long base = cbase[cg], alen = cbase[cg + 1] - base;
    const long long first = (long long)(blockIdx.x - b0) * CNS_BLOCK * CNS_CALL_ITEMS;
    unsigned run = block_off[blockIdx.x];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int u = 0; u < CNS_CALL_ITEMS; u++) {
        const long long j = first + (long long)u * CNS_BLOCK + threadIdx.x;
        const unsigned pk = j < alen ? packed[base + j] : 0u;
        const unsigned n = pk & 0xff;
        unsigned inc = n;                                   // inclusive scan inside the wavefront
        for (int d = 1; d < 64; d <<= 1) { const unsigned t = __shfl_up(inc, d); if (lane >= d) inc += t; }
        if (lane == 63) wsum_[wv] = inc;
        __syncthreads();
        unsigned before = 0, all = 0;
        for (int q = 0; q < 4; q++) { if (q < wv) before += wsum_[q]; all += wsum_[q]; }
        const unsigned at = run + before + inc - n;
        if (n >= 1) out[at] = (char)((pk >> 8) & 0xff);
        if (n >= 2) out[at + 1] = (char)((pk >> 16) & 0xff);
        run += all;
        __syncthreads();
    }
}

}  // namespace hinge
