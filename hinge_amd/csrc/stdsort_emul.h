// Exact emulation of libstdc++'s std::sort (GCC 11 <bits/stl_algo.h>, <bits/stl_heap.h>) on an index
// array, usable from host and device code.
//
// Why: the reference's hinge scan (src/filter/filter.cpp:914,1010) and its pile-up order
// (src/filter/filter.cpp:565-567) go through std::sort with comparators that look at one field only
// (pairAscend / pairDescend / compare_overlap, src/lib/LAInterface.cpp:4875-4889).  std::sort is not
// stable, so which of two equal-key elements comes first is a property of the introsort algorithm
// itself.  In the rare case where that order can change a hinge call (see k_hinge_call), the device
// replays the very same sequence of comparisons and moves: introsort_loop (median-of-3 to first,
// unguarded Hoare partition, recurse right / loop left, depth limit 2*floor(log2 n) then heapsort)
// followed by the final insertion sort with the 16-element threshold.
//
// The sort permutes `idx[0..n)`; ordering is by `key[idx[k]]`, descending if `desc` else ascending,
// i.e. comp(x, y) = desc ? key[x] > key[y] : key[x] < key[y].
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define HINGE_HD __host__ __device__
#else
#define HINGE_HD
#endif

namespace hinge_sort {

struct KeyCmp {
    const int* key;
    int desc;
    HINGE_HD bool operator()(int x, int y) const { return desc ? key[x] > key[y] : key[x] < key[y]; }
};

HINGE_HD inline void swap_i(int* a, int i, int j) { int t = a[i]; a[i] = a[j]; a[j] = t; }

HINGE_HD inline int floor_log2(unsigned n) {
    int l = -1;
    while (n) { n >>= 1; ++l; }
    return l;
}

// ---- heap part (reached only when the introsort depth limit hits) --------------------------------
// T = element type of the array, C = comparator on its elements (KeyCmp over an index array, PackedCmp over key | element words)
template <typename T, typename C>
HINGE_HD inline void push_heap_(T* a, int first, int hole, int top, T value, const C& c) {
    int parent = (hole - 1) / 2;
    while (hole > top && c(a[first + parent], value)) {
        a[first + hole] = a[first + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    a[first + hole] = value;
}

template <typename T, typename C>
HINGE_HD inline void adjust_heap_(T* a, int first, int hole, int len, T value, const C& c) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (c(a[first + child], a[first + (child - 1)])) child--;
        a[first + hole] = a[first + child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        a[first + hole] = a[first + (child - 1)];
        hole = child - 1;
    }
    push_heap_(a, first, hole, top, value, c);
}

template <typename T, typename C>
HINGE_HD inline void heapsort_(T* a, int first, int last, const C& c) {   // __partial_sort(first,last,last)
    int len = last - first;
    if (len >= 2) {                                                              // __make_heap
        int parent = (len - 2) / 2;
        while (true) {
            T value = a[first + parent];
            adjust_heap_(a, first, parent, len, value, c);
            if (parent == 0) break;
            parent--;
        }
    }
    while (last - first > 1) {                                                   // __sort_heap
        --last;
        T value = a[last];                                                       // __pop_heap(first,last,last)
        a[last] = a[first];
        adjust_heap_(a, first, 0, last - first, value, c);
    }
}

// key | element in one word (pileup_order.h, the packed replay): comp(x, y) = key(x) > key(y) - the element bits never take part
struct PackedCmp {
    int shift;
    HINGE_HD bool operator()(unsigned x, unsigned y) const { return (x >> shift) > (y >> shift); }
};

// ---- introsort -----------------------------------------------------------------------------------
HINGE_HD inline void move_median_to_first_(int* a, int result, int x, int y, int z, const KeyCmp& c) {
    if (c(a[x], a[y])) {
        if (c(a[y], a[z])) swap_i(a, result, y);
        else if (c(a[x], a[z])) swap_i(a, result, z);
        else swap_i(a, result, x);
    } else if (c(a[x], a[z])) swap_i(a, result, x);
    else if (c(a[y], a[z])) swap_i(a, result, z);
    else swap_i(a, result, y);
}

HINGE_HD inline int unguarded_partition_(int* a, int first, int last, int pivot, const KeyCmp& c) {
    while (true) {
        while (c(a[first], a[pivot])) ++first;
        --last;
        while (c(a[pivot], a[last])) --last;
        if (!(first < last)) return first;
        swap_i(a, first, last);
        ++first;
    }
}

HINGE_HD inline void unguarded_linear_insert_(int* a, int last, const KeyCmp& c) {
    int val = a[last];
    int next = last - 1;
    while (c(val, a[next])) {
        a[last] = a[next];
        last = next;
        --next;
    }
    a[last] = val;
}

HINGE_HD inline void insertion_sort_(int* a, int first, int last, const KeyCmp& c) {
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        if (c(a[i], a[first])) {
            int val = a[i];
            for (int k = i; k > first; --k) a[k] = a[k - 1];
            a[first] = val;
        } else {
            unguarded_linear_insert_(a, i, c);
        }
    }
}

// __introsort_loop(idx + first, idx + last, depth, comp).  The explicit stack replaces the recursion on
// the right part: the recursive call on [cut,last) runs BEFORE the loop continues on [first,cut), but the
// two ranges are disjoint, so deferring one of them performs the same operations on each range.
template <int STACK>
HINGE_HD inline void introsort_loop(int* idx, int first0, int last0, int depth0, const KeyCmp& c) {
    const int THRESH = 16;
    int stack_first[STACK], stack_last[STACK], stack_depth[STACK];
    int sp = 0;
    stack_first[0] = first0; stack_last[0] = last0; stack_depth[0] = depth0;
    sp = 1;
    while (sp > 0) {
        --sp;
        int first = stack_first[sp], last = stack_last[sp], depth = stack_depth[sp];
        while (last - first > THRESH) {
            if (depth == 0) {
                heapsort_(idx, first, last, c);
                break;
            }
            --depth;
            int mid = first + (last - first) / 2;
            move_median_to_first_(idx, first, first + 1, mid, last - 1, c);
            int cut = unguarded_partition_(idx, first + 1, last, first, c);
            // defer the left part, continue with the right part (any order gives the same result)
            stack_first[sp] = first; stack_last[sp] = cut; stack_depth[sp] = depth;
            ++sp;
            first = cut;
        }
    }
}

// std::sort(idx, idx + n, comp)
HINGE_HD inline void std_sort(int* idx, int n, const int* key, int desc) {
    if (n <= 0) return;
    KeyCmp c{key, desc};
    const int THRESH = 16;
    introsort_loop<64>(idx, 0, n, floor_log2((unsigned)n) * 2, c);
    // __final_insertion_sort
    if (n > THRESH) {
        insertion_sort_(idx, 0, THRESH, c);
        for (int i = THRESH; i != n; ++i) unguarded_linear_insert_(idx, i, c);
    } else {
        insertion_sort_(idx, 0, n, c);
    }
}

}  // namespace hinge_sort
