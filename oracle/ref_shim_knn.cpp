// TEST INFRASTRUCTURE ONLY -- never linked into the product library.
//
// extern "C" doorway onto the reference's own kNN translation unit
// (/root/reference/utils/nearest_neighbors/knn_.cxx, declarations in knn_.h:2-19).
// The reference exports C++-mangled symbols and is normally reached through a
// Cython module (knn.pyx) that does not build under Cython 3; this shim just
// forwards, so ctypes can call the reference's compiled code unchanged.
// Built by oracle/Makefile into oracle/_ref/ (git-ignored); no reference source
// is copied into this repository.
#include <cstddef>
#include "knn_.h"

extern "C" {

void ref_knn(const float* pts, size_t npts, size_t dim, const float* queries,
             size_t nq, size_t K, long* out) {
    cpp_knn(pts, npts, dim, queries, nq, K, out);
}

void ref_knn_omp(const float* pts, size_t npts, size_t dim, const float* queries,
                 size_t nq, size_t K, long* out) {
    cpp_knn_omp(pts, npts, dim, queries, nq, K, out);
}

void ref_knn_batch(const float* pts, size_t B, size_t npts, size_t dim,
                   const float* queries, size_t nq, size_t K, long* out) {
    cpp_knn_batch(pts, B, npts, dim, queries, nq, K, out);
}

void ref_knn_batch_omp(const float* pts, size_t B, size_t npts, size_t dim,
                       const float* queries, size_t nq, size_t K, long* out) {
    cpp_knn_batch_omp(pts, B, npts, dim, queries, nq, K, out);
}

}  // extern "C"
