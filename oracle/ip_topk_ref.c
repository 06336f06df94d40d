/* ORACLE (test infrastructure — never linked into libwise_hip.so or loaded by wise_amd/).
 *
 * Plain-C restatement of faiss IndexFlatIP::search as the reference calls it at
 * /root/reference/src/index/feature_search_index.py:113 and /root/reference/api/routes.py:1407:
 * for each query a sequential pass over the fp32 rows (faiss's nq<20 path is a dot-product loop,
 * not BLAS), keeping the k best in a binary min-heap, then sorting descending.
 * Ties: lower row first (faiss: unspecified).  Padding: (-FLT_MAX, -1).
 * Also the scalar CPU baseline bench.py times ("port", 1 core).
 */
#include <float.h>
#include <stdint.h>
#include <stdlib.h>

typedef struct { float s; int64_t r; } ent;

/* a is worse than b: lower score, or equal score and higher row */
static int worse(ent a, ent b) { return a.s < b.s || (a.s == b.s && a.r > b.r); }

static void sift_down(ent* h, int n, int i) {
    for (;;) {
        int l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && worse(h[l], h[m])) m = l;
        if (r < n && worse(h[r], h[m])) m = r;
        if (m == i) return;
        ent t = h[i]; h[i] = h[m]; h[m] = t;
        i = m;
    }
}

static int cmp_desc(const void* pa, const void* pb) {
    ent a = *(const ent*)pa, b = *(const ent*)pb;
    if (worse(a, b)) return 1;
    if (worse(b, a)) return -1;
    return 0;
}

/* X [N,d], Q [nq,d]; ids may be NULL (id = id_base + row). outD/outI [nq,k]. */
void wise_oracle_ip_topk(const float* X, int64_t N, int d, const float* Q, int nq, int k, const int64_t* ids,
                         int64_t id_base, float* outD, int64_t* outI) {
    ent* heap = (ent*)malloc(sizeof(ent) * (size_t)k);
    for (int q = 0; q < nq; ++q) {
        const float* qv = Q + (size_t)q * d;
        int n = 0;
        for (int64_t r = 0; r < N; ++r) {
            const float* x = X + (size_t)r * d;
            float s = 0.f;
            for (int j = 0; j < d; ++j) s += x[j] * qv[j];
            ent e = {s, r};
            if (n < k) {
                heap[n++] = e;
                if (n == k) for (int i = k / 2 - 1; i >= 0; --i) sift_down(heap, k, i);
            } else if (worse(heap[0], e)) {
                heap[0] = e;
                sift_down(heap, k, 0);
            }
        }
        qsort(heap, (size_t)n, sizeof(ent), cmp_desc);
        for (int i = 0; i < k; ++i) {
            if (i < n) {
                outD[(size_t)q * k + i] = heap[i].s;
                outI[(size_t)q * k + i] = ids ? ids[heap[i].r] : id_base + heap[i].r;
            } else {
                outD[(size_t)q * k + i] = -FLT_MAX;
                outI[(size_t)q * k + i] = -1;
            }
        }
    }
    free(heap);
}
