// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).  PARITY UNPINNED.
#include "orc_pcl.h"
#include <limits>

namespace orc {

// pcl::VoxelGrid<PointXYZI>::applyFilter (PCL 1.10 filters/impl/voxel_grid.hpp), as called at
// scan_registration.cpp:433-437 (leaf 0.2) and laser_mapping.cpp:433-439,689-702 (0.4 / 0.8).
Cloud voxel_grid(const Cloud& in, float leaf) {
  Cloud out;
  if (in.empty()) return out;
  const float inv = 1.0f / leaf;  // inverse_leaf_size_ = Array4f::Ones() / leaf_size_.array()
  // getMinMax3D (dense cloud)
  float mn[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
  float mx[3] = {-mn[0], -mn[1], -mn[2]};
  for (const auto& p : in) {
    mn[0] = std::min(mn[0], p.x); mn[1] = std::min(mn[1], p.y); mn[2] = std::min(mn[2], p.z);
    mx[0] = std::max(mx[0], p.x); mx[1] = std::max(mx[1], p.y); mx[2] = std::max(mx[2], p.z);
  }
  int64_t dx = static_cast<int64_t>((mx[0] - mn[0]) * inv) + 1;
  int64_t dy = static_cast<int64_t>((mx[1] - mn[1]) * inv) + 1;
  int64_t dz = static_cast<int64_t>((mx[2] - mn[2]) * inv) + 1;
  if ((dx * dy * dz) > static_cast<int64_t>(std::numeric_limits<int32_t>::max())) return in;  // PCL warns, copies input

  int min_b[3], max_b[3], div_b[3];
  for (int a = 0; a < 3; a++) {
    min_b[a] = static_cast<int>(std::floor(mn[a] * inv));
    max_b[a] = static_cast<int>(std::floor(mx[a] * inv));
    div_b[a] = max_b[a] - min_b[a] + 1;
  }
  const int mul[3] = {1, div_b[0], div_b[0] * div_b[1]};

  struct Item { unsigned idx; unsigned pt; };
  std::vector<Item> items;
  items.reserve(in.size());
  for (unsigned i = 0; i < in.size(); i++) {
    const auto& p = in[i];
    int ijk0 = static_cast<int>(std::floor(p.x * inv) - static_cast<float>(min_b[0]));
    int ijk1 = static_cast<int>(std::floor(p.y * inv) - static_cast<float>(min_b[1]));
    int ijk2 = static_cast<int>(std::floor(p.z * inv) - static_cast<float>(min_b[2]));
    int idx = ijk0 * mul[0] + ijk1 * mul[1] + ijk2 * mul[2];
    items.push_back({static_cast<unsigned>(idx), i});
  }
  auto less = [](const Item& a, const Item& b) { return a.idx < b.idx; };
#ifdef ORC_STD_SORT
  std::sort(items.begin(), items.end(), less);         // what PCL literally calls (order within a voxel unspecified)
#else
  std::stable_sort(items.begin(), items.end(), less);  // canonical: input order within a voxel
#endif
  size_t index = 0;
  while (index < items.size()) {
    size_t i = index + 1;
    while (i < items.size() && items[i].idx == items[index].idx) ++i;
    // CentroidPoint<PointXYZI>: AccumulatorXYZ (Vector3f sum) + AccumulatorIntensity (float sum), get = sum / n
    float sx = 0.0f, sy = 0.0f, sz = 0.0f, si = 0.0f;
    for (size_t li = index; li < i; ++li) {
      const auto& p = in[items[li].pt];
      sx += p.x; sy += p.y; sz += p.z; si += p.intensity;
    }
    const float n = static_cast<float>(i - index);
    out.push_back({sx / n, sy / n, sz / n, si / n});
    index = i;
  }
  return out;
}

// ---------------------------------------------------------------- exact kNN
int knn_brute(const Cloud& pts, const float q[3], int k, int* idx, float* d2) {
  int found = 0;
  for (int i = 0; i < (int)pts.size(); i++) {
    float d = l2_simple(q, pts[i]);
    if (found == k && !(d < d2[k - 1])) continue;  // ties: lower index (already present) wins
    int pos = found < k ? found : k - 1;
    while (pos > 0 && d < d2[pos - 1]) { d2[pos] = d2[pos - 1]; idx[pos] = idx[pos - 1]; pos--; }
    d2[pos] = d; idx[pos] = i;
    if (found < k) found++;
  }
  return found;
}

void KdTree::build(const Cloud& pts) {
  pts_ = &pts;
  order_.resize(pts.size());
  for (size_t i = 0; i < pts.size(); i++) order_[i] = (int)i;
  nodes_.clear();
  nodes_.reserve(pts.size() / 4 + 8);
  if (!pts.empty()) build_rec(0, (int)pts.size());
}

int KdTree::build_rec(int lo, int hi) {
  Node nd;
  nd.lo = lo; nd.hi = hi; nd.left = nd.right = -1;
  for (int a = 0; a < 3; a++) { nd.bmin[a] = std::numeric_limits<float>::max(); nd.bmax[a] = -std::numeric_limits<float>::max(); }
  for (int i = lo; i < hi; i++) {
    const PointXYZI& p = (*pts_)[order_[i]];
    const float c[3] = {p.x, p.y, p.z};
    for (int a = 0; a < 3; a++) { nd.bmin[a] = std::min(nd.bmin[a], c[a]); nd.bmax[a] = std::max(nd.bmax[a], c[a]); }
  }
  int me = (int)nodes_.size();
  nodes_.push_back(nd);
  if (hi - lo > 16) {
    int ax = 0;
    float ext = nd.bmax[0] - nd.bmin[0];
    for (int a = 1; a < 3; a++) if (nd.bmax[a] - nd.bmin[a] > ext) { ext = nd.bmax[a] - nd.bmin[a]; ax = a; }
    if (ext > 0.0f) {
      int mid = (lo + hi) / 2;
      const Cloud& P = *pts_;
      std::nth_element(order_.begin() + lo, order_.begin() + mid, order_.begin() + hi, [&](int a, int b) {
        float ca = ax == 0 ? P[a].x : (ax == 1 ? P[a].y : P[a].z);
        float cb = ax == 0 ? P[b].x : (ax == 1 ? P[b].y : P[b].z);
        return ca < cb || (ca == cb && a < b);
      });
      int l = build_rec(lo, mid);
      int r = build_rec(mid, hi);
      nodes_[me].left = l;
      nodes_[me].right = r;
    }
  }
  return me;
}

int KdTree::knn(const float q[3], int k, int* idx, float* d2) const {
  if (!pts_ || pts_->empty()) return 0;
  int found = 0;
  const Cloud& P = *pts_;
  auto box_bound = [&](const Node& nd) {
    double b = 0.0;
    for (int a = 0; a < 3; a++) {
      double g = 0.0;
      if (q[a] < nd.bmin[a]) g = (double)nd.bmin[a] - q[a];
      else if (q[a] > nd.bmax[a]) g = (double)q[a] - nd.bmax[a];
      b += g * g;
    }
    return b * (1.0 - 1e-6);  // slack: the f32 metric rounds; never prune a cell that could tie or win
  };
  // explicit stack DFS, nearer child first
  int stack[128];
  int sp = 0;
  stack[sp++] = 0;
  while (sp > 0) {
    const Node& nd = nodes_[stack[--sp]];
    if (found == k && box_bound(nd) > (double)d2[k - 1]) continue;
    if (nd.left < 0) {
      for (int i = nd.lo; i < nd.hi; i++) {
        const int pi = order_[i];
        const float d = l2_simple(q, P[pi]);
        if (found == k) {
          if (d > d2[k - 1] || (d == d2[k - 1] && pi > idx[k - 1])) continue;
        }
        int pos = found < k ? found : k - 1;
        while (pos > 0 && (d < d2[pos - 1] || (d == d2[pos - 1] && pi < idx[pos - 1]))) {
          d2[pos] = d2[pos - 1]; idx[pos] = idx[pos - 1]; pos--;
        }
        d2[pos] = d; idx[pos] = pi;
        if (found < k) found++;
      }
    } else {
      const Node& L = nodes_[nd.left];
      const Node& R = nodes_[nd.right];
      double bl = box_bound(L), br = box_bound(R);
      if (bl <= br) { stack[sp++] = nd.right; stack[sp++] = nd.left; }
      else { stack[sp++] = nd.left; stack[sp++] = nd.right; }
    }
  }
  return found;
}

}  // namespace orc
