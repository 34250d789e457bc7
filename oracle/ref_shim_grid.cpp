// TEST INFRASTRUCTURE ONLY -- never linked into the product library.
//
// extern "C" doorway onto the reference's voxel-grid subsampling core
// (/root/reference/utils/cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp:5-106,
//  declared in grid_subsampling.h:84-91). The reference's CPython wrapper
// (wrapper.cpp) does not compile against the NumPy 2 C-API, so ctypes drives the
// core through this forwarder instead. Built by oracle/Makefile into oracle/_ref/.
#include <cstdint>
#include <cstring>
#include <vector>
#include "grid_subsampling/grid_subsampling.h"

extern "C" {

// Returns the number of voxels M. Output buffers must hold N rows (M <= N).
// feats / classes may be NULL (fdim / ldim then ignored). Row order is the
// reference's own (libstdc++ unordered_map iteration order).
long ref_grid_subsampling(const float* pts, long N, const float* feats, int fdim,
                          const int* classes, int ldim, float sampleDl,
                          float* out_pts, float* out_feats, int* out_classes) {
    std::vector<PointXYZ> op((const PointXYZ*)pts, (const PointXYZ*)pts + N);
    std::vector<float> of;
    std::vector<int> oc;
    if (feats) of.assign(feats, feats + (size_t)N * fdim);
    if (classes) oc.assign(classes, classes + (size_t)N * ldim);
    std::vector<PointXYZ> sp;
    std::vector<float> sf;
    std::vector<int> sc;
    grid_subsampling(op, sp, of, sf, oc, sc, sampleDl, 0);
    long M = (long)sp.size();
    std::memcpy(out_pts, sp.data(), sizeof(float) * 3 * M);
    if (feats) std::memcpy(out_feats, sf.data(), sizeof(float) * sf.size());
    if (classes) std::memcpy(out_classes, sc.data(), sizeof(int) * sc.size());
    return M;
}

}  // extern "C"
