/* gosort_check.c -- TEST INFRASTRUCTURE (part of the oracle; nothing on the product path links or runs it).
 *
 * A second, independent restatement of Go 1.18's sort.Sort (src/sort/sort.go of the go1.18 release: Sort -> quickSort ->
 * doPivot / heapSort / insertionSort, medianOfThree, the ShellSort pass with gap 6 for <= 12 elements) in C, written from the
 * published source without looking at open-simulator_amd/gosort.py.  The Go standard library is not under /root/reference; the
 * reference pins go1.18.3 (Dockerfile:1).  tests/test_ref_parity.py runs both restatements over thousands of flag vectors (the
 * queues of pkg/algo compare with a Less(i, j) that looks at element i only: affinity.go:21-23, toleration.go:19-21) and pins
 * a few committed vectors (tests/golden/gosort_vectors.json, lengths 13 ... 60: the quickSort / ninther / duplicate-protection
 * branches), so that a slip in either restatement shows as a difference.  Parity with a Go BINARY stays unpinned until
 * oracle/run_ref.sh runs on a box with go1.18.
 *
 * usage: gosort_check <flags>      flags = a string of '0' / '1' (element i is "first" when flags[i] == '1': Less(i, j) = flag[i])
 *        prints the resulting permutation (original indices, space separated) on one line.
 *        gosort_check -k <keys>    keys = comma separated integers, Less(i, j) = key[i] < key[j] (a strict order, for the sanity test) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int n_el;
static long long* key;    /* sort key (or flag) per slot */
static int* ident;        /* original index per slot */
static int unary;         /* 1: Less(i, j) = key[i] != 0 */

static int less_(int i, int j) { return unary ? key[i] != 0 : key[i] < key[j]; }
static void swap_(int i, int j) {
    long long k = key[i]; key[i] = key[j]; key[j] = k;
    int t = ident[i]; ident[i] = ident[j]; ident[j] = t;
}

/* insertionSort sorts data[a:b] */
static void insertion_sort(int a, int b) {
    for (int i = a + 1; i < b; i++)
        for (int j = i; j > a && less_(j, j - 1); j--) swap_(j, j - 1);
}

/* siftDown implements the heap property on data[lo:hi]; first is an offset into the array where the root of the heap lies */
static void sift_down(int lo, int hi, int first) {
    int root = lo;
    for (;;) {
        int child = 2 * root + 1;
        if (child >= hi) return;
        if (child + 1 < hi && less_(first + child, first + child + 1)) child++;
        if (!less_(first + root, first + child)) return;
        swap_(first + root, first + child);
        root = child;
    }
}

static void heap_sort(int a, int b) {
    int first = a, lo = 0, hi = b - a;
    for (int i = (hi - 1) / 2; i >= 0; i--) sift_down(i, hi, first);       /* build heap with greatest element at top */
    for (int i = hi - 1; i >= 0; i--) { swap_(first, first + i); sift_down(lo, i, first); }   /* pop elements, largest first */
}

/* medianOfThree moves the median of the three values data[m0], data[m1], data[m2] into data[m1] */
static void median_of_three(int m1, int m0, int m2) {
    if (less_(m1, m0)) swap_(m1, m0);
    if (less_(m2, m1)) {
        swap_(m2, m1);
        if (less_(m1, m0)) swap_(m1, m0);
    }
}

static void do_pivot(int lo, int hi, int* midlo, int* midhi) {
    int m = (int)((unsigned)(lo + hi) >> 1);
    if (hi - lo > 40) {                                   /* Tukey's "Ninther": median of three medians of three */
        int s = (hi - lo) / 8;
        median_of_three(lo, lo + s, lo + 2 * s);
        median_of_three(m, m - s, m + s);
        median_of_three(hi - 1, hi - 1 - s, hi - 1 - 2 * s);
    }
    median_of_three(lo, m, hi - 1);
    int pivot = lo;
    int a = lo + 1, c = hi - 1;
    for (; a < c && less_(a, pivot); a++) {}
    int b = a;
    for (;;) {
        for (; b < c && !less_(pivot, b); b++) {}          /* data[b] <= pivot */
        for (; b < c && less_(pivot, c - 1); c--) {}       /* data[c-1] > pivot */
        if (b >= c) break;
        swap_(b, c - 1);                                   /* data[b] > pivot; data[c-1] <= pivot */
        b++; c--;
    }
    int protect = hi - c < 5;                              /* "if hi-c<3 then there are duplicates"; border 5 */
    if (!protect && hi - c < (hi - lo) / 4) {
        int dups = 0;
        if (!less_(pivot, hi - 1)) { swap_(c, hi - 1); c++; dups++; }     /* data[hi-1] = pivot */
        if (!less_(b - 1, pivot)) { b--; dups++; }                         /* data[b-1] = pivot */
        if (!less_(m, pivot)) { swap_(m, b - 1); b--; dups++; }            /* data[m] = pivot */
        protect = dups > 1;
    }
    if (protect) {                                         /* protect against a lot of duplicates */
        for (;;) {
            for (; a < b && !less_(b - 1, pivot); b--) {}  /* data[b] == pivot */
            for (; a < b && less_(a, pivot); a++) {}       /* data[a] < pivot */
            if (a >= b) break;
            swap_(a, b - 1);
            a++; b--;
        }
    }
    swap_(pivot, b - 1);                                   /* swap pivot into middle */
    *midlo = b - 1; *midhi = c;
}

static void quick_sort(int a, int b, int max_depth) {
    while (b - a > 12) {                                   /* ShellSort for slices <= 12 elements */
        if (max_depth == 0) { heap_sort(a, b); return; }
        max_depth--;
        int mlo, mhi;
        do_pivot(a, b, &mlo, &mhi);
        if (mlo - a < b - mhi) { quick_sort(a, mlo, max_depth); a = mhi; }     /* recurse on the smaller side */
        else { quick_sort(mhi, b, max_depth); b = mlo; }
    }
    if (b - a > 1) {
        for (int i = a + 6; i < b; i++)                    /* ShellSort pass with gap 6 */
            if (less_(i, i - 6)) swap_(i, i - 6);
        insertion_sort(a, b);
    }
}

/* maxDepth returns a threshold at which quicksort should switch to heapsort: 2*ceil(lg(n+1)) */
static int max_depth_of(int n) {
    int depth = 0;
    for (int i = n; i > 0; i >>= 1) depth++;
    return depth * 2;
}

int main(int argc, char** argv) {
    if (argc == 3 && !strcmp(argv[1], "-k")) {
        unary = 0;
        char* s = argv[2];
        int cap = (int)strlen(s) / 2 + 2;
        key = malloc(sizeof(long long) * cap); ident = malloc(sizeof(int) * cap);
        for (char* tok = strtok(s, ","); tok; tok = strtok(NULL, ",")) { key[n_el] = atoll(tok); ident[n_el] = n_el; n_el++; }
    } else if (argc == 2) {
        unary = 1;
        n_el = (int)strlen(argv[1]);
        key = malloc(sizeof(long long) * (n_el + 1)); ident = malloc(sizeof(int) * (n_el + 1));
        for (int i = 0; i < n_el; i++) { key[i] = argv[1][i] == '1'; ident[i] = i; }
    } else {
        fprintf(stderr, "usage: gosort_check <flags of 0/1> | -k <k0,k1,...>\n");
        return 2;
    }
    quick_sort(0, n_el, max_depth_of(n_el));
    for (int i = 0; i < n_el; i++) printf(i ? " %d" : "%d", ident[i]);
    printf("\n");
    return 0;
}
