// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).  PARITY UNPINNED.
//
// Restatement of the PCL 1.10 / FLANN 1.9 calls on the reference's hot path (un-vendored
// dependencies, README.md:25): pcl::removeNaNFromPointCloud, pcl::VoxelGrid<PointXYZI>::filter
// (filters/impl/voxel_grid.hpp applyFilter, downsample_all_data = true, min_points_per_voxel = 0)
// and pcl::KdTreeFLANN<PointXYZI>::nearestKSearch (exact search, flann::L2_Simple<float> on x,y,z).
//
// Canonicalisation of behaviour the reference leaves implementation-defined:
//  * VoxelGrid sorts (voxel idx, point) pairs with std::sort, whose order among equal idx is
//    unspecified; the f32 centroid sum then runs in that order.  The oracle uses a STABLE sort
//    (input order within a voxel).  Build with -DORC_STD_SORT to get libstdc++'s std::sort order
//    instead (tests check the two agree to 1 ulp-level tolerance).
//  * kNN ties on equal f32 distance: lowest index wins.
#pragma once
#include <cstdint>
#include <cmath>
#include <vector>
#include <algorithm>

namespace orc {

struct PointXYZI {  // == the reference's PointType (common.h:42) minus PCL's SSE padding
  float x, y, z, intensity;
};
typedef std::vector<PointXYZI> Cloud;

// VoxelGrid::filter.  Returns the filtered cloud (or a copy of the input when the int32 index
// space would overflow, as PCL does with a warning).
Cloud voxel_grid(const Cloud& in, float leaf);

// Exact k-nearest-neighbour index over x,y,z.
class KdTree {
 public:
  void build(const Cloud& pts);
  // Fills idx[k], d2[k] ascending by (d2, idx).  Returns number found (min(k, n)).
  int knn(const float q[3], int k, int* idx, float* d2) const;
  size_t size() const { return pts_ ? pts_->size() : 0; }

 private:
  struct Node {
    int lo, hi;          // index range into order_
    int left, right;     // children (-1 = leaf)
    float bmin[3], bmax[3];
  };
  const Cloud* pts_ = nullptr;
  std::vector<int> order_;
  std::vector<Node> nodes_;
  int build_rec(int lo, int hi);
};

// flann::L2_Simple<float>: result += diff*diff over the 3 dims, in f32, in x,y,z order.
inline float l2_simple(const float* a, const PointXYZI& p) {
  float r = 0.0f;
  float d0 = a[0] - p.x; r += d0 * d0;
  float d1 = a[1] - p.y; r += d1 * d1;
  float d2 = a[2] - p.z; r += d2 * d2;
  return r;
}

// brute-force reference for tests
int knn_brute(const Cloud& pts, const float q[3], int k, int* idx, float* d2);

}  // namespace orc
