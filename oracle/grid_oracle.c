/* TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's voxel-grid subsampling
 *   /root/reference/utils/cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp:5-106
 *   (+ SampledData, grid_subsampling.h:10-80; PointXYZ / floor / min_point / max_point,
 *      cpp_utils/cloud/cloud.h:40-143, cloud.cpp:27-66).
 * Arithmetic followed line by line (all float32 unless noted):
 *   origin      = floor(min * (1/dl)) * dl                            (grid_subsampling.cpp:27)
 *   NX, NY      = (size_t)floor((max - origin)/dl) + 1                (:30-31)
 *   iX,iY,iZ    = (size_t)floor((p - origin)/dl)                      (:53-55)
 *   key         = iX + NX*iY + NX*NY*iZ                               (:56)
 *   per voxel   : count, sum xyz, sum features in ARRIVAL order        (grid_subsampling.h:42-47)
 *                 per-label-column histogram                          (:48-53)
 *   barycentre  = sum * (float)(1.0/count)   (double reciprocal, then narrowed) (:87)
 *   feature     = sum / (float)count                                  (:90-95)
 *   label       = a most frequent value per column                    (:98-102)
 * Differences by construction (documented, asserted nowhere as index parity):
 *   - output ROW ORDER: the reference emits libstdc++ unordered_map iteration order
 *     (:85); this restatement and the HIP kernel emit ascending voxel key. Parity is
 *     on the voxel-keyed set of rows (tests re-key both sides by voxel).
 *   - label ties: the reference returns the first maximum in hash-map iteration
 *     order; here the SMALLEST label value among the maxima. Equal on tie-free input.
 * Pinned against the compiled reference core (oracle/_ref/libref_grid.so) and the
 * fixtures in tests/golden.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint64_t key;
    int64_t idx;
} KeyIdx;

static int cmp_keyidx(const void* a, const void* b) {
    const KeyIdx* x = (const KeyIdx*)a;
    const KeyIdx* y = (const KeyIdx*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}

static int cmp_int(const void* a, const void* b) {
    int x = *(const int*)a, y = *(const int*)b;
    return x < y ? -1 : (x > y);
}

/* Computes the voxel key of every point (also used by tests to re-key reference rows). */
void oracle_grid_keys(const float* pts, int64_t N, float dl, uint64_t* keys) {
    float mn[3] = {pts[0], pts[1], pts[2]}, mx[3] = {pts[0], pts[1], pts[2]};
    for (int64_t i = 0; i < N; ++i)
        for (int c = 0; c < 3; ++c) {
            float v = pts[3 * i + c];
            if (v < mn[c]) mn[c] = v;
            if (v > mx[c]) mx[c] = v;
        }
    float inv = 1 / dl;
    float org[3];
    for (int c = 0; c < 3; ++c) org[c] = floorf(mn[c] * inv) * dl;
    uint64_t NX = (uint64_t)floorf((mx[0] - org[0]) / dl) + 1;
    uint64_t NY = (uint64_t)floorf((mx[1] - org[1]) / dl) + 1;
    for (int64_t i = 0; i < N; ++i) {
        uint64_t iX = (uint64_t)floorf((pts[3 * i + 0] - org[0]) / dl);
        uint64_t iY = (uint64_t)floorf((pts[3 * i + 1] - org[1]) / dl);
        uint64_t iZ = (uint64_t)floorf((pts[3 * i + 2] - org[2]) / dl);
        keys[i] = iX + NX * iY + NX * NY * iZ;
    }
}

/* Returns M (number of occupied voxels); rows ascending by voxel key.
 * feats/classes may be NULL. out_keys (optional) receives the key of each row. */
int64_t oracle_grid_subsample(const float* pts, int64_t N, const float* feats, int fdim,
                              const int32_t* classes, int ldim, float dl, float* out_pts,
                              float* out_feats, int32_t* out_classes, uint64_t* out_keys) {
    if (N <= 0) return 0;
    uint64_t* keys = (uint64_t*)malloc(sizeof(uint64_t) * N);
    oracle_grid_keys(pts, N, dl, keys);
    KeyIdx* ki = (KeyIdx*)malloc(sizeof(KeyIdx) * N);
    for (int64_t i = 0; i < N; ++i) {
        ki[i].key = keys[i];
        ki[i].idx = i;
    }
    qsort(ki, N, sizeof(KeyIdx), cmp_keyidx);
    int* lab = (int*)malloc(sizeof(int) * N);
    int64_t M = 0;
    for (int64_t s = 0; s < N;) {
        int64_t e = s;
        while (e < N && ki[e].key == ki[s].key) ++e;
        int count = (int)(e - s);
        float sx = 0, sy = 0, sz = 0;
        for (int64_t t = s; t < e; ++t) { /* arrival order == ascending original index */
            const float* p = pts + 3 * ki[t].idx;
            sx += p[0];
            sy += p[1];
            sz += p[2];
        }
        float r = (float)(1.0 / count);
        out_pts[3 * M + 0] = sx * r;
        out_pts[3 * M + 1] = sy * r;
        out_pts[3 * M + 2] = sz * r;
        if (feats) {
            float fc = (float)count;
            for (int f = 0; f < fdim; ++f) {
                float acc = 0;
                for (int64_t t = s; t < e; ++t) acc += feats[(size_t)ki[t].idx * fdim + f];
                out_feats[(size_t)M * fdim + f] = acc / fc;
            }
        }
        if (classes) {
            for (int l = 0; l < ldim; ++l) {
                for (int64_t t = s; t < e; ++t) lab[t - s] = classes[(size_t)ki[t].idx * ldim + l];
                qsort(lab, count, sizeof(int), cmp_int);
                int best = lab[0], bestc = 0;
                for (int a = 0; a < count;) {
                    int b = a;
                    while (b < count && lab[b] == lab[a]) ++b;
                    if (b - a > bestc) {
                        bestc = b - a;
                        best = lab[a];
                    }
                    a = b;
                }
                out_classes[(size_t)M * ldim + l] = best;
            }
        }
        if (out_keys) out_keys[M] = ki[s].key;
        ++M;
        s = e;
    }
    free(lab);
    free(ki);
    free(keys);
    return M;
}
