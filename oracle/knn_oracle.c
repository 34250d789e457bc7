/* TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's exact kNN
 *   /root/reference/utils/nearest_neighbors/knn_.cxx:22-135  (cpp_knn / cpp_knn_batch[_omp])
 * which queries a nanoflann 1.2.3 KD-tree (leaf 10, eps 0 => exact search).
 * What it pins:
 *   - distance arithmetic: float32 squared L2 accumulated x -> y -> z starting from 0,
 *     one rounding per multiply and per add (nanoflann.hpp:323-347, tail loop for dim<4;
 *     groups of four summed left-to-right before being added for dim>=4);
 *   - result order: ascending distance (nanoflann.hpp:115-138 insertion sort);
 *   - output type: signed 64-bit indices, [B, Nq, K] row-major (knn_.cxx:78-99).
 * Tie rule: the reference keeps whichever equal-distance candidate its tree
 * traversal meets first (nanoflann.hpp:122 strict '>' and :1361 strict '<'), which
 * is a property of the tree, not of the data. This oracle (and the HIP kernel)
 * break ties towards the LOWER point index; on tie-free inputs both agree and
 * that is where index parity is asserted. Parity is pinned against the compiled
 * reference (oracle/_ref/libref_knn.so) and the committed fixtures in tests/golden.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.
 * Build: see oracle/Makefile (-O2 -ffp-contract=off: no FMA contraction).
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

static inline float sqdist(const float* q, const float* p, size_t dim) {
    float r = 0.0f;
    size_t d = 0;
    while (d + 3 < dim) { /* nanoflann.hpp:332-343: groups of four */
        float d0 = q[d] - p[d], d1 = q[d + 1] - p[d + 1];
        float d2 = q[d + 2] - p[d + 2], d3 = q[d + 3] - p[d + 3];
        r += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
        d += 4;
    }
    for (; d < dim; ++d) { /* nanoflann.hpp:344-347 */
        float t = q[d] - p[d];
        r += t * t;
    }
    return r;
}

/* One cloud: brute force, top-K by (distance, index). */
static void knn_one(const float* pts, size_t npts, size_t dim, const float* queries,
                    size_t nq, size_t K, int64_t* out) {
    float* bd = (float*)malloc(sizeof(float) * K);
    int64_t* bi = (int64_t*)malloc(sizeof(int64_t) * K);
    for (size_t i = 0; i < nq; ++i) {
        size_t cnt = 0;
        const float* q = queries + i * dim;
        for (size_t j = 0; j < npts; ++j) {
            float dist = sqdist(q, pts + j * dim, dim);
            if (cnt == K && !(dist < bd[K - 1])) continue; /* j ascending: later equal loses */
            size_t pos = cnt < K ? cnt : K - 1;
            while (pos > 0 && bd[pos - 1] > dist) {
                bd[pos] = bd[pos - 1];
                bi[pos] = bi[pos - 1];
                --pos;
            }
            bd[pos] = dist;
            bi[pos] = (int64_t)j;
            if (cnt < K) ++cnt;
        }
        for (size_t k = 0; k < K; ++k) out[i * K + k] = k < cnt ? bi[k] : 0;
    }
    free(bd);
    free(bi);
}

/* Mirrors cpp_knn_batch (knn_.cxx:72-102): per-cloud independent search. */
void oracle_knn_batch(const float* pts, size_t B, size_t npts, size_t dim,
                      const float* queries, size_t nq, size_t K, int64_t* out) {
#pragma omp parallel for schedule(dynamic, 1)
    for (long b = 0; b < (long)B; ++b)
        knn_one(pts + (size_t)b * npts * dim, npts, dim, queries + (size_t)b * nq * dim, nq, K,
                out + (size_t)b * nq * K);
}

/* Mirrors cpp_knn (knn_.cxx:22-44). */
void oracle_knn(const float* pts, size_t npts, size_t dim, const float* queries, size_t nq,
                size_t K, int64_t* out) {
    knn_one(pts, npts, dim, queries, nq, K, out);
}

/* Sorted squared distances of the K nearest (for tie-heavy inputs where only the
 * distance multiset is well defined). */
void oracle_knn_dists(const float* pts, size_t npts, size_t dim, const float* queries,
                      size_t nq, const int64_t* idx, size_t K, float* out) {
    for (size_t i = 0; i < nq; ++i)
        for (size_t k = 0; k < K; ++k)
            out[i * K + k] = sqdist(queries + i * dim, pts + (size_t)idx[i * K + k] * dim, dim);
}
