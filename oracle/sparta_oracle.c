/*
 * sparta_oracle.c -- CPU ORACLE (test infrastructure only; see sparta_oracle.h).
 *
 * Plain-C restatement of the reference's hot path.  Deliberately LITERAL: column-level cluster patterns,
 * the sequential lossy merge, one integer division per pattern element, the triple loop -- i.e. the
 * reference's algorithm, not the product's (which pre-reduces rows to block lists, uses a closed-form
 * merge and MFMA kernels).  Agreement between the two is therefore meaningful.
 *
 * Build: gcc -O2 -std=c99 -ffp-contract=off (oracle/Makefile).  -ffp-contract=off keeps `c += a*b` an
 * unfused multiply and add, as in the reference's x86-64 build (its makefile:2 has no -march, so no FMA).
 *
 * Third-party algorithm restated here: std::sort of libstdc++ (GCC 11.4, bits/stl_algo.h / stl_heap.h:
 * introsort with median-of-3 pivot, threshold 16, depth limit 2*floor(log2 n), heap-sort fallback, final
 * insertion sort).  The reference's get_permutation (src/general/utilities.cpp:8-20) sorts row indices with
 * it under a comparator that only looks at the group id, so the order of rows INSIDE a group -- and with it
 * the row order of every VBS block -- is whatever that exact algorithm leaves.
 * Likewise std::set of libstdc++ (bits/stl_tree.h, src/c++98/tree.cc: red-black tree with a header node):
 * IterativeBlockingKeeper (blocking.cpp:509-511) increments an iterator past end(), so the elements it erases
 * depend on the tree's shape; the insert / erase rebalancing and the iterator increment are restated below.
 */
#include "sparta_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------ */
/* distances                                                                                        */
/* ------------------------------------------------------------------------------------------------ */

static long lmax(long a, long b) { return a > b ? a : b; }

/* src/general/blocking.cpp:859-921 */
float oracle_hamming_distance_group(const long* row_A, long size_A, long group_size_A, const long* row_B, long size_B,
                                    long group_size_B, long block_size)
{
    if (size_A == 0 && size_B == 0) return 0;                                               /* :863 */
    if (size_A == 0 || size_B == 0) return (float)lmax(size_A * group_size_A, size_B * group_size_B);   /* :864 */
    /* count_zeros = 1 (:861): a block only in A weighs group_size_B, a block only in B weighs group_size_A */
    const long add_to_count_A = group_size_B, add_to_count_B = group_size_A;                /* :867-871 */
    long i = 0, j = 0, count = 0;
    while (i < size_A && j < size_B) {                                                      /* :884-904 */
        long pos_A = row_A[i] / block_size, pos_B = row_B[j] / block_size;
        if (pos_A < pos_B) {
            count += add_to_count_A;
            while (i < size_A && row_A[i] / block_size == pos_A) i++;
        } else if (pos_A > pos_B) {
            count += add_to_count_B;
            while (j < size_B && row_B[j] / block_size == pos_B) j++;
        } else {
            while (i < size_A && row_A[i] / block_size == pos_A) i++;
            while (j < size_B && row_B[j] / block_size == pos_B) j++;
        }
    }
    while (i < size_A) {                                                                    /* :906-911 */
        long pos_A = row_A[i] / block_size;
        count += add_to_count_A;
        while (i < size_A && row_A[i] / block_size == pos_A) i++;
    }
    while (j < size_B) {                                                                    /* :913-918 */
        long pos_B = row_B[j] / block_size;
        count += add_to_count_B;
        while (j < size_B && row_B[j] / block_size == pos_B) j++;
    }
    return (float)count;                                                                    /* :920 */
}

/* src/general/blocking.cpp:923-994 */
float oracle_jaccard_distance_group(const long* row_A, long size_A, long group_size_A, const long* row_B, long size_B,
                                    long group_size_B, long block_size)
{
    if (size_A == 0 && size_B == 0) return 0;                                               /* :926 */
    if (size_A == 0 || size_B == 0) return 1;                                               /* :927 */
    const long add_to_count_A = group_size_B, add_to_count_B = group_size_A;                /* :932-936 */
    long i = 0, j = 0, count = 0, block_size_A = 0, block_size_B = 0;
    while (i < size_A && j < size_B) {                                                      /* :951-975 */
        long pos_A = row_A[i] / block_size, pos_B = row_B[j] / block_size;
        if (pos_A < pos_B) {
            count += add_to_count_A;
            block_size_A++;
            while (i < size_A && row_A[i] / block_size == pos_A) i++;
        } else if (pos_A > pos_B) {
            count += add_to_count_B;
            block_size_B++;
            while (j < size_B && row_B[j] / block_size == pos_B) j++;
        } else {
            block_size_A++;
            block_size_B++;
            while (i < size_A && row_A[i] / block_size == pos_A) i++;
            while (j < size_B && row_B[j] / block_size == pos_B) j++;
        }
    }
    while (i < size_A) {                                                                    /* :977-983 */
        long pos_A = row_A[i] / block_size;
        count += add_to_count_A;
        block_size_A++;
        while (i < size_A && row_A[i] / block_size == pos_A) i++;
    }
    while (j < size_B) {                                                                    /* :985-991 */
        long pos_B = row_B[j] / block_size;
        count += add_to_count_B;
        block_size_B++;
        while (j < size_B && row_B[j] / block_size == pos_B) j++;
    }
    /* :993 -- double division, then narrowed to the float return type */
    return (float)((2.0 * (double)count) / (double)(block_size_A * group_size_A + block_size_B * group_size_B + count));
}

/* ------------------------------------------------------------------------------------------------ */
/* merge_rows: src/general/utilities.cpp:145-173 (lossy "union")                                    */
/* ------------------------------------------------------------------------------------------------ */

/* first position in [lo, hi) whose element is >= val */
static long lower_bound_l(const long* a, long lo, long hi, long val)
{
    while (lo < hi) {
        long mid = lo + (hi - lo) / 2;
        if (a[mid] < val) lo = mid + 1; else hi = mid;
    }
    return lo;
}

long oracle_merge_rows(const long* A, long size_A, const long* B, long size_B, long* result)
{
    long n = 0, i = 0, j = 0;
    while (j < size_B) {                                          /* :152 */
        long B_val = B[j];
        long new_i = lower_bound_l(A, i, size_A, B_val);          /* :157 */
        if (new_i == size_A) break;                               /* :159 -- A[i..) is NOT copied */
        for (long t = i; t < new_i; t++) result[n++] = A[t];      /* :163 */
        result[n++] = B_val;                                      /* :164 */
        if (A[new_i] == B_val) new_i++;                           /* :165 */
        i = new_i;
        j++;
    }
    for (; j < size_B; j++) result[n++] = B[j];                   /* :171 -- A's tail is never appended */
    return n;
}

/* ------------------------------------------------------------------------------------------------ */
/* std::sort (libstdc++ 11 introsort) on an index array with comparator key[a] < key[b]             */
/* ------------------------------------------------------------------------------------------------ */

typedef struct { const long* key; } cmp_t;
/* the reference's lambda takes `int i, int j` (utilities.cpp:10): the indices pass through int */
static int comp(const cmp_t* c, long a, long b) { return c->key[(int)a] < c->key[(int)b]; }

static void swap_l(long* a, long* b) { long t = *a; *a = *b; *b = t; }

/* bits/stl_algo.h: __move_median_to_first(result, a, b, c) */
static void move_median_to_first(long* v, long result, long a, long b, long c, const cmp_t* k)
{
    if (comp(k, v[a], v[b])) {
        if (comp(k, v[b], v[c])) swap_l(&v[result], &v[b]);
        else if (comp(k, v[a], v[c])) swap_l(&v[result], &v[c]);
        else swap_l(&v[result], &v[a]);
    } else if (comp(k, v[a], v[c])) swap_l(&v[result], &v[a]);
    else if (comp(k, v[b], v[c])) swap_l(&v[result], &v[c]);
    else swap_l(&v[result], &v[b]);
}

/* __unguarded_partition(first, last, pivot) */
static long unguarded_partition(long* v, long first, long last, long pivot, const cmp_t* k)
{
    for (;;) {
        while (comp(k, v[first], v[pivot])) ++first;
        --last;
        while (comp(k, v[pivot], v[last])) --last;
        if (!(first < last)) return first;
        swap_l(&v[first], &v[last]);
        ++first;
    }
}

/* bits/stl_heap.h: __push_heap / __adjust_heap / __make_heap / __pop_heap / __sort_heap on v[first..first+len) */
static void push_heap_(long* v, long first, long hole, long top, long value, const cmp_t* k)
{
    long parent = (hole - 1) / 2;
    while (hole > top && comp(k, v[first + parent], value)) {
        v[first + hole] = v[first + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    v[first + hole] = value;
}
static void adjust_heap_(long* v, long first, long hole, long len, long value, const cmp_t* k)
{
    const long top = hole;
    long child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (comp(k, v[first + child], v[first + (child - 1)])) child--;
        v[first + hole] = v[first + child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        v[first + hole] = v[first + (child - 1)];
        hole = child - 1;
    }
    push_heap_(v, first, hole, top, value, k);
}
static void heap_sort_(long* v, long first, long last, const cmp_t* k)
{
    /* __partial_sort(first, last, last) = __heap_select (== __make_heap here) + __sort_heap */
    long len = last - first;
    if (len >= 2) {
        long parent = (len - 2) / 2;
        for (;;) {
            long value = v[first + parent];
            adjust_heap_(v, first, parent, len, value, k);
            if (parent == 0) break;
            parent--;
        }
    }
    while (last - first > 1) {
        --last;
        long value = v[last];
        v[last] = v[first];
        adjust_heap_(v, first, 0, last - first, value, k);
    }
}

static void introsort_loop(long* v, long first, long last, long depth_limit, const cmp_t* k)
{
    while (last - first > 16) {                                   /* _S_threshold */
        if (depth_limit == 0) { heap_sort_(v, first, last, k); return; }
        --depth_limit;
        long mid = first + (last - first) / 2;
        move_median_to_first(v, first, first + 1, mid, last - 1, k);
        long cut = unguarded_partition(v, first + 1, last, first, k);
        introsort_loop(v, cut, last, depth_limit, k);
        last = cut;
    }
}
static void unguarded_linear_insert(long* v, long last, const cmp_t* k)
{
    long val = v[last];
    long next = last - 1;
    while (comp(k, val, v[next])) { v[last] = v[next]; last = next; --next; }
    v[last] = val;
}
static void insertion_sort_(long* v, long first, long last, const cmp_t* k)
{
    if (first == last) return;
    for (long i = first + 1; i != last; ++i) {
        if (comp(k, v[i], v[first])) {
            long val = v[i];
            memmove(&v[first + 1], &v[first], (size_t)(i - first) * sizeof(long));
            v[first] = val;
        } else unguarded_linear_insert(v, i, k);
    }
}
static void std_sort_indices(long* v, long n, const long* key)
{
    cmp_t k = { key };
    if (n <= 0) return;
    long lg = 0;
    for (long t = n; t > 1; t >>= 1) lg++;                        /* std::__lg(n) */
    introsort_loop(v, 0, n, lg * 2, &k);
    if (n > 16) {                                                 /* __final_insertion_sort */
        insertion_sort_(v, 0, 16, &k);
        for (long i = 16; i < n; ++i) unguarded_linear_insert(v, i, &k);
    } else insertion_sort_(v, 0, n, &k);
}

/* src/general/utilities.cpp:8-20 */
void oracle_get_permutation(const long* grouping, long n, long* perm)
{
    for (long i = 0; i < n; i++) perm[i] = i;                     /* iota (:16) */
    std_sort_indices(perm, n, grouping);                          /* :17 */
}

static int cmp_long(const void* a, const void* b)
{
    long x = *(const long*)a, y = *(const long*)b;
    return (x > y) - (x < y);
}

/* src/general/utilities.cpp:22-43 (sorting VALUES: any correct sort gives the same array) */
long oracle_get_partition(const long* grouping, long n, long* partition)
{
    long* s = (long*)malloc(sizeof(long) * (size_t)(n > 0 ? n : 1));
    memcpy(s, grouping, sizeof(long) * (size_t)n);
    qsort(s, (size_t)n, sizeof(long), cmp_long);                  /* :27 */
    long np = 0, current_group = -1;                              /* :29 */
    for (long i = 0; i < n; i++) {
        if (current_group != s[i]) { current_group = s[i]; partition[np++] = i; }   /* :34-38 */
    }
    partition[np++] = n;                                          /* :41 */
    free(s);
    return np;
}

/* src/general/utilities.cpp:45-54 */
void oracle_get_fixed_size_grouping(const long* grouping, long n, long row_block_size, long* result)
{
    long* perm = (long*)malloc(sizeof(long) * (size_t)(n > 0 ? n : 1));
    oracle_get_permutation(grouping, n, perm);
    for (long i = 0; i < n; i++) result[i] = -1;
    for (long i = 0; i < n; i++) result[perm[i]] = i / row_block_size;   /* :51 */
    free(perm);
}

/* ------------------------------------------------------------------------------------------------ */
/* blocking algorithms                                                                              */
/* ------------------------------------------------------------------------------------------------ */

typedef float (*dist_fn)(const long*, long, long, const long*, long, long, long);

typedef struct {
    long* data;
    long size, cap;
} lvec;
static void lvec_reserve(lvec* v, long need)
{
    if (need <= v->cap) return;
    long cap = v->cap ? v->cap : 16;
    while (cap < need) cap *= 2;
    v->data = (long*)realloc(v->data, sizeof(long) * (size_t)cap);
    v->cap = cap;
}

/* `float distances[cmat.rows] = {-1};` (blocking.cpp:159,255): element 0 is -1, the rest are 0 */
static float* make_distances(long rows)
{
    float* d = (float*)calloc((size_t)(rows > 0 ? rows : 1), sizeof(float));
    if (rows > 0) d[0] = -1.0f;
    return d;
}

/* IterativeBlockingPatternCLOCKED, src/general/blocking.cpp:156-243 */
static void blocking_clocked(long rows, const long* rowptr, const long* colidx, float tau, dist_fn distance, long block_size,
                             int use_size, int use_pattern, long* grouping, long* comparison_counter, long* merge_counter)
{
    for (long i = 0; i < rows; i++) grouping[i] = -1;                                       /* :158 */
    float* distances = make_distances(rows);                                                /* :159 */
    lvec pattern = {0, 0, 0}, merged = {0, 0, 0};
    for (long i = 0; i < rows; i++) {                                                       /* :166 */
        if (grouping[i] != -1) continue;                                                    /* :168 */
        long current_group_size = 1;
        grouping[i] = i;                                                                    /* :172 */
        long ni = rowptr[i + 1] - rowptr[i];
        lvec_reserve(&pattern, ni);
        memcpy(pattern.data, colidx + rowptr[i], sizeof(long) * (size_t)ni);                /* :173 */
        pattern.size = ni;
        for (long j = i + 1; j < rows; j++) {                                               /* :180 */
            if (distances[i] != -1 && distances[j] != -1 && fabsf(distances[i] - distances[j]) > tau) {   /* :192 */
                distances[j] = -1;
                continue;
            }
            if (grouping[j] == -1) {                                                        /* :199 */
                (*comparison_counter)++;
                const long* row_j = colidx + rowptr[j];
                long nj = rowptr[j + 1] - rowptr[j];
                float dist = distance(pattern.data, pattern.size, current_group_size, row_j, nj, 1, block_size);   /* :203 */
                distances[j] = dist;                                                        /* :205 */
                if (dist <= tau) {                                                          /* :207 */
                    (*merge_counter)++;
                    grouping[j] = i;                                                        /* :213 */
                    if (use_pattern) {                                                      /* :214-220 */
                        lvec_reserve(&merged, pattern.size + nj);
                        merged.size = oracle_merge_rows(pattern.data, pattern.size, row_j, nj, merged.data);
                        lvec t = pattern; pattern = merged; merged = t;
                    }
                    if (use_size) current_group_size++;                                     /* :221-224 */
                }
            }
        }
    }
    free(distances); free(pattern.data); free(merged.data);
}

/* IterativeBlockingQueue, src/general/blocking.cpp:245-338 */
static void blocking_queue(long rows, const long* rowptr, const long* colidx, float tau, dist_fn distance, long block_size,
                           int use_size, int use_pattern, long* grouping, long* comparison_counter, long* merge_counter)
{
    for (long i = 0; i < rows; i++) grouping[i] = -1;
    long* row_queue = (long*)malloc(sizeof(long) * (size_t)(rows > 0 ? rows : 1));
    long* inner_queue = (long*)malloc(sizeof(long) * (size_t)(rows > 0 ? rows : 1));
    long qn = rows;
    for (long i = 0; i < rows; i++) row_queue[i] = i;                                       /* :250-253 */
    float* distances = make_distances(rows);                                                /* :255 */
    lvec pattern = {0, 0, 0}, merged = {0, 0, 0};
    while (qn > 0) {                                                                        /* :262 */
        long i = row_queue[0];                                                              /* :264-266 */
        long current_group_size = 1;
        grouping[i] = i;
        long ni = rowptr[i + 1] - rowptr[i];
        lvec_reserve(&pattern, ni);
        memcpy(pattern.data, colidx + rowptr[i], sizeof(long) * (size_t)ni);
        pattern.size = ni;
        long inner_n = 0;
        for (long q = 1; q < qn; q++) {                                                     /* :277 */
            long j = row_queue[q];
            if (distances[i] != -1 && distances[j] != -1 && fabsf(distances[i] - distances[j]) > tau) {   /* :284 */
                distances[j] = -1;
                inner_queue[inner_n++] = j;
                continue;
            }
            (*comparison_counter)++;
            const long* row_j = colidx + rowptr[j];
            long nj = rowptr[j + 1] - rowptr[j];
            float dist = distance(pattern.data, pattern.size, current_group_size, row_j, nj, 1, block_size);   /* :293 */
            distances[j] = dist;
            if (dist > tau) {                                                               /* :297 */
                inner_queue[inner_n++] = j;
            } else {
                (*merge_counter)++;
                grouping[j] = i;
                if (use_pattern) {
                    lvec_reserve(&merged, pattern.size + nj);
                    merged.size = oracle_merge_rows(pattern.data, pattern.size, row_j, nj, merged.data);
                    lvec t = pattern; pattern = merged; merged = t;
                }
                if (use_size) current_group_size++;
            }
        }
        long* t = row_queue; row_queue = inner_queue; inner_queue = t;                      /* :324 swap */
        qn = inner_n;
    }
    free(distances); free(pattern.data); free(merged.data); free(row_queue); free(inner_queue);
}

/* IterativeBlockingPattern, src/general/blocking.cpp:89-154: strict `<` (:124) and -- the un-braced
 * `if (use_pattern)` guards only a timer macro -- the merge ALWAYS runs (:128-132). */
static void blocking_plain(long rows, const long* rowptr, const long* colidx, float tau, dist_fn distance, long block_size,
                           int use_size, long* grouping, long* comparison_counter, long* merge_counter)
{
    for (long i = 0; i < rows; i++) grouping[i] = -1;
    lvec pattern = {0, 0, 0}, merged = {0, 0, 0};
    for (long i = 0; i < rows; i++) {
        if (grouping[i] != -1) continue;
        long current_group_size = 1;
        grouping[i] = i;
        long ni = rowptr[i + 1] - rowptr[i];
        lvec_reserve(&pattern, ni);
        memcpy(pattern.data, colidx + rowptr[i], sizeof(long) * (size_t)ni);
        pattern.size = ni;
        for (long j = i + 1; j < rows; j++) {
            if (grouping[j] != -1) continue;
            (*comparison_counter)++;
            const long* row_j = colidx + rowptr[j];
            long nj = rowptr[j + 1] - rowptr[j];
            float dist = distance(pattern.data, pattern.size, current_group_size, row_j, nj, 1, block_size);
            if (dist < tau) {
                (*merge_counter)++;
                grouping[j] = i;
                lvec_reserve(&merged, pattern.size + nj);
                merged.size = oracle_merge_rows(pattern.data, pattern.size, row_j, nj, merged.data);
                lvec t = pattern; pattern = merged; merged = t;
                if (use_size) current_group_size++;
            }
        }
    }
    free(pattern.data); free(merged.data);
}

/* IterativeBlockingPatternMN, src/general/blocking.cpp:19-87 with check_structured_sparsity / update_structured_sparsity,
 * src/general/utilities.cpp:56-129: the plain algorithm plus an m:n guard -- inside every run of structured_n merged rows no
 * column may be hit more than structured_m times.  Note `if (use_pattern)` guards the merge here (it does not in the plain one). */
static int mn_check(const lvec* pat, const lvec* cnt, const long* row, long row_len, int m)
{
    long i = 0, j = 0;
    while (i < pat->size && j < row_len) {
        if (pat->data[i] < row[j]) i++;
        else if (pat->data[i] > row[j]) j++;
        else { if (cnt->data[i] >= m) return 0; i++; j++; }
    }
    return 1;
}
static void mn_update(lvec* pat, lvec* cnt, lvec* np_, lvec* nc_, const long* row, long row_len)
{
    lvec_reserve(np_, pat->size + row_len); lvec_reserve(nc_, pat->size + row_len);
    long i = 0, j = 0, k = 0;
    while (i < pat->size && j < row_len) {
        if (pat->data[i] < row[j]) { np_->data[k] = pat->data[i]; nc_->data[k++] = cnt->data[i]; i++; }
        else if (pat->data[i] > row[j]) { np_->data[k] = row[j]; nc_->data[k++] = 1; j++; }
        else { np_->data[k] = pat->data[i]; nc_->data[k++] = cnt->data[i] + 1; i++; j++; }
    }
    while (i < pat->size) { np_->data[k] = pat->data[i]; nc_->data[k++] = cnt->data[i]; i++; }
    while (j < row_len) { np_->data[k] = row[j]; nc_->data[k++] = 1; j++; }
    np_->size = nc_->size = k;
    lvec t = *pat; *pat = *np_; *np_ = t;
    t = *cnt; *cnt = *nc_; *nc_ = t;
}
static void blocking_mn(long rows, const long* rowptr, const long* colidx, float tau, dist_fn distance, long block_size, int use_size,
                        int use_pattern, int structured_m, int structured_n, long* grouping, long* comparison_counter, long* merge_counter)
{
    for (long i = 0; i < rows; i++) grouping[i] = -1;
    lvec pattern = {0, 0, 0}, merged = {0, 0, 0}, sp = {0, 0, 0}, sc = {0, 0, 0}, t1 = {0, 0, 0}, t2 = {0, 0, 0};
    for (long i = 0; i < rows; i++) {
        if (grouping[i] != -1) continue;
        long current_group_size = 1;
        grouping[i] = i;
        long ni = rowptr[i + 1] - rowptr[i];
        lvec_reserve(&pattern, ni);
        memcpy(pattern.data, colidx + rowptr[i], sizeof(long) * (size_t)ni);
        pattern.size = ni;
        int row_counter = 1;                                           /* :33 */
        lvec_reserve(&sp, ni); lvec_reserve(&sc, ni);
        memcpy(sp.data, pattern.data, sizeof(long) * (size_t)ni);
        for (long k = 0; k < ni; k++) sc.data[k] = 1;
        sp.size = sc.size = ni;
        for (long j = i + 1; j < rows; j++) {
            if (grouping[j] != -1) continue;
            (*comparison_counter)++;
            const long* row_j = colidx + rowptr[j];
            long nj = rowptr[j + 1] - rowptr[j];
            float dist = distance(pattern.data, pattern.size, current_group_size, row_j, nj, 1, block_size);
            if (dist < tau) {
                int ok = 1;
                if (row_counter % structured_n == 0) { row_counter = 0; sp.size = sc.size = 0; }      /* :50-56 */
                else ok = mn_check(&sp, &sc, row_j, nj, structured_m);
                if (ok) {
                    (*merge_counter)++;
                    grouping[j] = i;
                    if (use_pattern) {
                        lvec_reserve(&merged, pattern.size + nj);
                        merged.size = oracle_merge_rows(pattern.data, pattern.size, row_j, nj, merged.data);
                        lvec t = pattern; pattern = merged; merged = t;
                    }
                    if (use_size) current_group_size++;
                    mn_update(&sp, &sc, &t1, &t2, row_j, nj);
                    row_counter++;
                }
            }
        }
    }
    free(pattern.data); free(merged.data); free(sp.data); free(sc.data); free(t1.data); free(t2.data);
}

/* ------------------------------------------------------------------------------------------------ */
/* std::set<std::pair<float,long>> of libstdc++ 11 (bits/stl_tree.h, src/c++98/tree.cc), restated   */
/* ------------------------------------------------------------------------------------------------ */
/* IterativeBlockingKeeper trims its candidate set with
 *       auto it = best.end(); advance(it, k); best.erase(it, best.end());     (blocking.cpp:509-511)
 * i.e. it increments an iterator PAST end().  That is undefined by the standard but deterministic on
 * libstdc++: incrementing the header node lands on the right-most node (or its left child) and the walk wraps.
 * Which elements get erased depends on the SHAPE of the red-black tree, so the oracle has to rebuild the tree
 * exactly: _Rb_tree_insert_and_rebalance, _Rb_tree_rebalance_for_erase, _Rb_tree_increment as published. */
typedef struct rbn {
    int red;
    struct rbn *parent, *left, *right;
    float d;
    long row;
} rbn;
typedef struct { rbn header; long count; rbn* pool; long pool_n, pool_cap; rbn* free_list; } rbset;

static int key_less(float d1, long r1, float d2, long r2) { return d1 < d2 || (!(d2 < d1) && r1 < r2); }   /* std::pair operator< */

static void rb_init(rbset* s, long cap)
{
    s->header.red = 1; s->header.parent = 0; s->header.left = &s->header; s->header.right = &s->header;
    s->count = 0; s->pool = (rbn*)malloc(sizeof(rbn) * (size_t)(cap + 1)); s->pool_n = 0; s->pool_cap = cap + 1; s->free_list = 0;
}
static void rb_clear(rbset* s)
{
    s->header.parent = 0; s->header.left = &s->header; s->header.right = &s->header; s->count = 0; s->pool_n = 0; s->free_list = 0;
}
static rbn* rb_alloc(rbset* s)
{
    if (s->free_list) { rbn* n = s->free_list; s->free_list = n->parent; return n; }
    return &s->pool[s->pool_n++];
}
static void rb_free_node(rbset* s, rbn* n) { n->parent = s->free_list; s->free_list = n; }

static rbn* rb_increment(rbn* x)                         /* _Rb_tree_increment; also what ++end() does */
{
    if (x->right != 0) { x = x->right; while (x->left != 0) x = x->left; }
    else { rbn* y = x->parent; while (x == y->right) { x = y; y = y->parent; } if (x->right != y) x = y; }
    return x;
}
static rbn* rb_decrement(rbn* x)                         /* _Rb_tree_decrement */
{
    if (x->red && x->parent->parent == x) x = x->right;  /* x is the header */
    else if (x->left != 0) { rbn* y = x->left; while (y->right != 0) y = y->right; x = y; }
    else { rbn* y = x->parent; while (x == y->left) { x = y; y = y->parent; } x = y; }
    return x;
}
static void rb_rotate_left(rbn* x, rbn** root)
{
    rbn* y = x->right;
    x->right = y->left;
    if (y->left != 0) y->left->parent = x;
    y->parent = x->parent;
    if (x == *root) *root = y; else if (x == x->parent->left) x->parent->left = y; else x->parent->right = y;
    y->left = x; x->parent = y;
}
static void rb_rotate_right(rbn* x, rbn** root)
{
    rbn* y = x->left;
    x->left = y->right;
    if (y->right != 0) y->right->parent = x;
    y->parent = x->parent;
    if (x == *root) *root = y; else if (x == x->parent->right) x->parent->right = y; else x->parent->left = y;
    y->right = x; x->parent = y;
}
static void rb_insert_and_rebalance(int insert_left, rbn* x, rbn* p, rbn* header)
{
    rbn** root = &header->parent;
    x->parent = p; x->left = 0; x->right = 0; x->red = 1;
    if (insert_left) {
        p->left = x;
        if (p == header) { header->parent = x; header->right = x; }
        else if (p == header->left) header->left = x;
    } else {
        p->right = x;
        if (p == header->right) header->right = x;
    }
    while (x != *root && x->parent->red) {
        rbn* const xpp = x->parent->parent;
        if (x->parent == xpp->left) {
            rbn* const y = xpp->right;
            if (y && y->red) { x->parent->red = 0; y->red = 0; xpp->red = 1; x = xpp; }
            else {
                if (x == x->parent->right) { x = x->parent; rb_rotate_left(x, root); }
                x->parent->red = 0; xpp->red = 1; rb_rotate_right(xpp, root);
            }
        } else {
            rbn* const y = xpp->left;
            if (y && y->red) { x->parent->red = 0; y->red = 0; xpp->red = 1; x = xpp; }
            else {
                if (x == x->parent->left) { x = x->parent; rb_rotate_right(x, root); }
                x->parent->red = 0; xpp->red = 1; rb_rotate_left(xpp, root);
            }
        }
    }
    (*root)->red = 0;
}
static rbn* rb_rebalance_for_erase(rbn* const z, rbn* header)
{
    rbn** root = &header->parent;
    rbn** leftmost = &header->left;
    rbn** rightmost = &header->right;
    rbn* y = z; rbn* x = 0; rbn* x_parent = 0;
    if (y->left == 0) x = y->right;
    else if (y->right == 0) x = y->left;
    else { y = y->right; while (y->left != 0) y = y->left; x = y->right; }
    if (y != z) {
        z->left->parent = y; y->left = z->left;
        if (y != z->right) {
            x_parent = y->parent;
            if (x) x->parent = y->parent;
            y->parent->left = x;
            y->right = z->right; z->right->parent = y;
        } else x_parent = y;
        if (*root == z) *root = y; else if (z->parent->left == z) z->parent->left = y; else z->parent->right = y;
        y->parent = z->parent;
        { int t = y->red; y->red = z->red; z->red = t; }
        y = z;
    } else {
        x_parent = y->parent;
        if (x) x->parent = y->parent;
        if (*root == z) *root = x; else if (z->parent->left == z) z->parent->left = x; else z->parent->right = x;
        if (*leftmost == z) {
            if (z->right == 0) *leftmost = z->parent;
            else { rbn* m = x; while (m->left != 0) m = m->left; *leftmost = m; }
        }
        if (*rightmost == z) {
            if (z->left == 0) *rightmost = z->parent;
            else { rbn* m = x; while (m->right != 0) m = m->right; *rightmost = m; }
        }
    }
    if (!y->red) {
        while (x != *root && (x == 0 || !x->red)) {
            if (x == x_parent->left) {
                rbn* w = x_parent->right;
                if (w->red) { w->red = 0; x_parent->red = 1; rb_rotate_left(x_parent, root); w = x_parent->right; }
                if ((w->left == 0 || !w->left->red) && (w->right == 0 || !w->right->red)) { w->red = 1; x = x_parent; x_parent = x_parent->parent; }
                else {
                    if (w->right == 0 || !w->right->red) { w->left->red = 0; w->red = 1; rb_rotate_right(w, root); w = x_parent->right; }
                    w->red = x_parent->red; x_parent->red = 0;
                    if (w->right) w->right->red = 0;
                    rb_rotate_left(x_parent, root);
                    break;
                }
            } else {
                rbn* w = x_parent->left;
                if (w->red) { w->red = 0; x_parent->red = 1; rb_rotate_right(x_parent, root); w = x_parent->left; }
                if ((w->right == 0 || !w->right->red) && (w->left == 0 || !w->left->red)) { w->red = 1; x = x_parent; x_parent = x_parent->parent; }
                else {
                    if (w->left == 0 || !w->left->red) { w->right->red = 0; w->red = 1; rb_rotate_left(w, root); w = x_parent->left; }
                    w->red = x_parent->red; x_parent->red = 0;
                    if (w->left) w->left->red = 0;
                    rb_rotate_right(x_parent, root);
                    break;
                }
            }
        }
        if (x) x->red = 0;
    }
    return y;
}
/* std::set::insert (unique): _M_get_insert_unique_pos + _M_insert_ */
static void rb_insert_unique(rbset* s, float d, long row)
{
    rbn* header = &s->header;
    rbn* x = header->parent;
    rbn* y = header;
    int comp = 1;
    while (x != 0) { y = x; comp = key_less(d, row, x->d, x->row); x = comp ? x->left : x->right; }
    rbn* j = y;
    if (comp) {
        if (j == header->left) goto do_insert;           /* j == begin() */
        j = rb_decrement(j);
    }
    if (!key_less(j->d, j->row, d, row)) return;          /* equivalent key present */
do_insert:
    {
        const int insert_left = (y == header) || key_less(d, row, y->d, y->row);
        rbn* z = rb_alloc(s);
        z->d = d; z->row = row;
        rb_insert_and_rebalance(insert_left, z, y, header);
        s->count++;
    }
}
/* std::set::erase(first, end()) : _M_erase_aux(first, last) */
static void rb_erase_to_end(rbset* s, rbn* first)
{
    rbn* header = &s->header;
    if (first == header->left && s->count > 0 && first != header) {   /* first == begin() && last == end() -> clear() */
        rb_clear(s);
        return;
    }
    while (first != header) {
        rbn* cur = first;
        first = rb_increment(first);
        rbn* y = rb_rebalance_for_erase(cur, header);
        rb_free_node(s, y);
        s->count--;
    }
}

/* IterativeBlockingKeeper, src/general/blocking.cpp:433-549 (what `-a 5` runs: dispatch at :655) */
static void blocking_keeper(long rows, const long* rowptr, const long* colidx, float tau, dist_fn distance, long col_block_size,
                            long max_row_block_size, int use_pattern, long* grouping, long* comparison_counter, long* merge_counter)
{
    for (long i = 0; i < rows; i++) grouping[i] = -1;                                       /* :435 */
    float* distances = make_distances(rows);                                                /* :436 */
    lvec pattern = {0, 0, 0}, merged = {0, 0, 0}, merged_rows = {0, 0, 0};
    rbset best;
    rb_init(&best, rows);
    for (long i = 0; i < rows; i++) {                                                       /* :443 */
        if (grouping[i] != -1) continue;
        rb_clear(&best);                                                                    /* :447 fresh set per seed */
        merged_rows.size = 0;
        const long group_number = i + rows;                                                 /* :450 */
        long current_group_size = 1;
        grouping[i] = group_number;
        lvec_reserve(&merged_rows, 1); merged_rows.data[merged_rows.size++] = i;
        long ni = rowptr[i + 1] - rowptr[i];
        lvec_reserve(&pattern, ni);
        memcpy(pattern.data, colidx + rowptr[i], sizeof(long) * (size_t)ni);
        pattern.size = ni;
        for (long j = i + 1; j < rows; j++) {                                               /* :460 */
            if (current_group_size == max_row_block_size) break;                            /* :463 */
            if (distances[i] != -1 && distances[j] != -1 && fabsf(distances[i] - distances[j]) > tau) {   /* :469 */
                distances[j] = -1;
                continue;
            }
            if (grouping[j] != -1) continue;                                                /* :476 */
            (*comparison_counter)++;
            const long* row_j = colidx + rowptr[j];
            long nj = rowptr[j + 1] - rowptr[j];
            float dist = distance(pattern.data, pattern.size, current_group_size, row_j, nj, 1, col_block_size);   /* :480 */
            distances[j] = dist;
            if (dist <= tau) {                                                              /* :484 */
                (*merge_counter)++;
                grouping[j] = group_number;
                lvec_reserve(&merged_rows, merged_rows.size + 1); merged_rows.data[merged_rows.size++] = j;
                if (use_pattern) {
                    lvec_reserve(&merged, pattern.size + nj);
                    merged.size = oracle_merge_rows(pattern.data, pattern.size, row_j, nj, merged.data);
                    lvec t = pattern; pattern = merged; merged = t;
                }
                current_group_size++;                                                       /* :501 */
            } else {
                rb_insert_unique(&best, dist, j);                                           /* :505-506 */
                const unsigned long room = (unsigned long)max_row_block_size - (unsigned long)merged_rows.size;
                if ((unsigned long)best.count > room) {                                     /* :507 */
                    rbn* it = &best.header;                                                 /* end() */
                    for (unsigned long t = 0; t < room; t++) it = rb_increment(it);         /* :510 advance PAST end() */
                    rb_erase_to_end(&best, it);                                             /* :511 */
                }
            }
        }
        if (current_group_size < max_row_block_size) {                                      /* :517-525 */
            for (rbn* it = best.header.left; best.count > 0 && it != &best.header && current_group_size != max_row_block_size; it = rb_increment(it)) {
                grouping[it->row] = group_number;
                lvec_reserve(&merged_rows, merged_rows.size + 1); merged_rows.data[merged_rows.size++] = it->row;
                current_group_size++;
            }
        }
        if (current_group_size == max_row_block_size)                                       /* :527-533 */
            for (long t = 0; t < merged_rows.size; t++) grouping[merged_rows.data[t]] -= rows;
    }
    free(distances); free(pattern.data); free(merged.data); free(merged_rows.data); free(best.pool);
}

/* BlockingEngine::GetGrouping, src/general/blocking.cpp:633-676 */
int oracle_get_grouping(long rows, const long* rowptr, const long* colidx, int blocking_algo, int sim_measure, float tau,
                        long col_block_size, long row_block_size, int use_groups, int use_pattern, int force_fixed_size,
                        long* grouping, long* counters)
{
    return oracle_get_grouping_mn(rows, rowptr, colidx, blocking_algo, sim_measure, tau, col_block_size, row_block_size, use_groups,
                                  use_pattern, force_fixed_size, 2, 4, grouping, counters);     /* include/blocking.h:20-21 defaults */
}

int oracle_get_grouping_mn(long rows, const long* rowptr, const long* colidx, int blocking_algo, int sim_measure, float tau,
                           long col_block_size, long row_block_size, int use_groups, int use_pattern, int force_fixed_size,
                           int structured_m, int structured_n, long* grouping, long* counters)
{
    long cmp = 0, mrg = 0;
    /* SetComparator (blocking.cpp:699-717): 0/2 Hamming, 1/3 Jaccard (the "OPENMP" twins compute the same) */
    dist_fn distance = (sim_measure & 1) ? oracle_jaccard_distance_group : oracle_hamming_distance_group;
    switch (blocking_algo) {
        case 3: blocking_clocked(rows, rowptr, colidx, tau, distance, col_block_size, use_groups, use_pattern, grouping, &cmp, &mrg); break;
        case 4: blocking_queue(rows, rowptr, colidx, tau, distance, col_block_size, use_groups, use_pattern, grouping, &cmp, &mrg); break;
        case 0: blocking_plain(rows, rowptr, colidx, tau, distance, col_block_size, use_groups, grouping, &cmp, &mrg); break;
        case 2: for (long i = 0; i < rows; i++) grouping[i] = i / row_block_size; break;   /* FixedBlocking :554-562 */
        case 5: blocking_keeper(rows, rowptr, colidx, tau, distance, col_block_size, row_block_size, use_pattern, grouping, &cmp, &mrg); break;
        case 1: blocking_mn(rows, rowptr, colidx, tau, distance, col_block_size, use_groups, use_pattern, structured_m, structured_n, grouping, &cmp, &mrg); break;
        default: return -1;
    }
    if (force_fixed_size && blocking_algo != 2) {                                           /* :670-673 */
        long* tmp = (long*)malloc(sizeof(long) * (size_t)(rows > 0 ? rows : 1));
        oracle_get_fixed_size_grouping(grouping, rows, row_block_size, tmp);
        memcpy(grouping, tmp, sizeof(long) * (size_t)rows);
        free(tmp);
    }
    if (counters) { counters[0] = cmp; counters[1] = mrg; }
    return 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* VBS builder: VBR::fill_from_CSR_inplace, src/general/vbr.cpp:135-237                             */
/* ------------------------------------------------------------------------------------------------ */
int oracle_vbr_fill_inplace(long cmat_rows, long cmat_cols, const long* rowptr, const long* colidx, const float* vals,
                            const long* grouping, long col_block_size, long row_block_size, int force_fixed_size,
                            long* dims_out, long* row_part, long* nzcount, long* jab, float* mab)
{
    long* row_partition = (long*)malloc(sizeof(long) * (size_t)(cmat_rows + 2));
    long npart = oracle_get_partition(grouping, cmat_rows, row_partition);                  /* :139 */
    long rows, cols;
    if (force_fixed_size) {                                                                 /* :143-148 */
        rows = ((cmat_rows - 1) / row_block_size + 1) * row_block_size;
        cols = ((cmat_cols - 1) / col_block_size + 1) * col_block_size;
        row_partition[npart - 1] = rows;
    } else { rows = cmat_rows; cols = cmat_cols; }
    long* row_permutation = (long*)malloc(sizeof(long) * (size_t)(rows > cmat_rows ? rows : cmat_rows) + sizeof(long));
    oracle_get_permutation(grouping, cmat_rows, row_permutation);                           /* :140 */
    for (long i = cmat_rows; i < rows; i++) row_permutation[i] = i;                         /* :147 */

    const long block_cols = (cols - 1) / col_block_size + 1;                                /* :156 */
    const long block_rows = npart - 1;                                                      /* :157 */
    char* nonzero_flags = (char*)malloc((size_t)block_cols);
    long* before = (long*)malloc(sizeof(long) * (size_t)(block_cols + 1));
    long jab_n = 0, mab_n = 0;

    for (long ib = 0; ib < block_rows; ib++) {                                              /* :175 */
        memset(nonzero_flags, 0, (size_t)block_cols);                                       /* :177 */
        const long h = row_partition[ib + 1] - row_partition[ib];                           /* :178 */
        for (long ir = row_partition[ib]; ir < row_partition[ib + 1]; ir++) {               /* :181-192 */
            long i = row_permutation[ir];
            if (i >= cmat_rows) continue;
            for (long nz = rowptr[i]; nz < rowptr[i + 1]; nz++) nonzero_flags[colidx[nz] / col_block_size] = 1;
        }
        long count = 0;
        for (long jb = 0; jb < block_cols; jb++) {                                          /* :195-201 */
            before[jb] = count;                 /* == std::count(flags.begin(), flags.begin() + jb, true), :222 */
            if (nonzero_flags[jb]) {
                if (jab) jab[jab_n] = jb;
                jab_n++;
                count++;
            }
        }
        if (nzcount) nzcount[ib] = count;
        const long current_mab_size = mab_n;                                                /* :205 */
        mab_n += count * h * col_block_size;                                                /* :206 */
        if (mab) {
            for (long t = current_mab_size; t < mab_n; t++) mab[t] = 0.0f;
            for (long ir = row_partition[ib]; ir < row_partition[ib + 1]; ir++) {           /* :207-228 */
                long i = row_permutation[ir];
                if (i >= cmat_rows) continue;
                for (long nz = rowptr[i]; nz < rowptr[i + 1]; nz++) {
                    long j = colidx[nz];
                    float d = vals ? vals[nz] : 1.0f;                                       /* :216-218 */
                    long pos = current_mab_size + before[j / col_block_size] * col_block_size * h +
                               h * (j % col_block_size) + (ir - row_partition[ib]);         /* :224 column-major */
                    mab[pos] = d;
                }
            }
        }
    }
    if (row_part) memcpy(row_part, row_partition, sizeof(long) * (size_t)(block_rows + 1));
    if (dims_out) {
        dims_out[0] = rows; dims_out[1] = cols; dims_out[2] = block_rows; dims_out[3] = block_cols;
        dims_out[4] = mab_n; dims_out[5] = jab_n;
    }
    free(row_partition); free(row_permutation); free(nonzero_flags); free(before);
    return 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* multiplies                                                                                       */
/* ------------------------------------------------------------------------------------------------ */

/* VBR::multiply, src/general/vbr.cpp:323-372, block-rows [ib0, ib1) */
void oracle_vbr_multiply_range(long rows, long cols, long block_col_size, const long* row_part, const long* nzcount,
                               const long* jab, const float* mab, long ib0, long ib1, const float* B, int B_cols, float* C)
{
    const long B_rows = cols, C_rows = rows;                                                /* :331-332 */
    long vbmat_idx = 0, jab_idx = 0;
    for (long ib = 0; ib < ib0; ib++) {             /* advance to the range (the reference walks from 0) */
        vbmat_idx += (row_part[ib + 1] - row_part[ib]) * block_col_size * nzcount[ib];
        jab_idx += nzcount[ib];
    }
    for (long ib = ib0; ib < ib1; ib++) {                                                   /* :342 */
        const long rows_in_block = row_part[ib + 1] - row_part[ib];                         /* :344 */
        for (long nzs = 0; nzs < nzcount[ib]; nzs++) {                                      /* :346 */
            const long jb = jab[jab_idx];                                                   /* :349 */
            const float* d_B_block = B + block_col_size * jb;                               /* :351 */
            const float* d_A_block = mab + vbmat_idx;                                       /* :354 */
            float* d_C_block = C + row_part[ib];                                            /* :355 */
            for (long i = 0; i < rows_in_block; i++)                                        /* :358-363 */
                for (long j = 0; j < B_cols; j++)
                    for (long k = 0; k < block_col_size; k++) {
                        /* the reference reads B out of bounds when jb*w + k >= cols; the stored A value there is 0 */
                        const float b = (block_col_size * jb + k < B_rows) ? d_B_block[k + j * B_rows] : 0.0f;
                        d_C_block[i + C_rows * j] += d_A_block[i + k * rows_in_block] * b;
                    }
            vbmat_idx += rows_in_block * block_col_size;                                    /* :366 */
            jab_idx++;                                                                      /* :367 */
        }
    }
}

void oracle_vbr_multiply(long rows, long cols, long block_rows, long block_col_size, const long* row_part, const long* nzcount,
                         const long* jab, const float* mab, const float* B, int B_cols, float* C)
{
    oracle_vbr_multiply_range(rows, cols, block_col_size, row_part, nzcount, jab, mab, 0, block_rows, B, B_cols, C);
}

/* CSR::multiply, src/general/csr.cpp:49-65 */
void oracle_csr_multiply(long rows, const long* rowptr, const long* colidx, const float* vals, const float* B, long ldb,
                         long B_cols, float* C)
{
    for (long i = 0; i < rows; i++)                                                         /* :53 */
        for (long nz = rowptr[i]; nz < rowptr[i + 1]; nz++) {                               /* :55 */
            const long col = colidx[nz];
            const float val = vals ? vals[nz] : 1.0f;                                       /* :58 */
            for (long j = 0; j < B_cols; j++) C[i + j * rows] += val * B[col + j * ldb];    /* :59-62 (ldb == rows there) */
        }
}

/* ------------------------------------------------------------------------------------------------ */
/* BlockingEngine::CollectBlockingInfo, src/general/blocking.cpp:576-631                            */
/* ------------------------------------------------------------------------------------------------ */
void oracle_collect_blocking_info(long rows, long cols, const long* rowptr, const long* colidx, const long* grouping,
                                  long col_block_size, long* info_out, float* avg_height_out)
{
    long* part = (long*)malloc(sizeof(long) * (size_t)(rows + 2));
    long* perm = (long*)malloc(sizeof(long) * (size_t)(rows + 1));
    long npart = oracle_get_partition(grouping, rows, part);                                /* :583 */
    oracle_get_permutation(grouping, rows, perm);                                           /* :584 */
    const long block_cols = (long)ceilf(((float)cols) / col_block_size);                    /* :589 */
    const long block_rows = npart - 1;
    char* flags = (char*)malloc((size_t)(block_cols > 0 ? block_cols : 1));
    long VBR_nzcount = 0, VBR_nzblocks_count = 0, VBR_longest_row = 0, total_blocks_height = 0;
    for (long ib = 0; ib < block_rows; ib++) {                                              /* :595 */
        memset(flags, 0, (size_t)block_cols);
        const long row_block_size = part[ib + 1] - part[ib];
        for (long ir = part[ib]; ir < part[ib + 1]; ir++) {                                 /* :601-609 */
            long i = perm[ir];
            for (long nz = rowptr[i]; nz < rowptr[i + 1]; nz++) flags[colidx[nz] / col_block_size] = 1;
        }
        long cnt = 0;
        for (long jb = 0; jb < block_cols; jb++)                                            /* :613-622 */
            if (flags[jb]) {
                cnt++;
                VBR_nzcount += col_block_size * row_block_size;
                VBR_nzblocks_count++;
                total_blocks_height += row_block_size;
            }
        if (cnt > VBR_longest_row) VBR_longest_row = cnt;                                   /* :612 */
        if (cols % col_block_size != 0 && flags[block_cols - 1])                            /* :624-627 */
            VBR_nzcount -= row_block_size * (col_block_size - cols % col_block_size);
    }
    info_out[0] = VBR_nzcount; info_out[1] = VBR_nzblocks_count; info_out[2] = VBR_longest_row;
    if (avg_height_out) *avg_height_out = ((float)total_blocks_height) / VBR_nzblocks_count;   /* :630 */
    free(part); free(perm); free(flags);
}
