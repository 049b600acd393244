// ORACLE — test infrastructure.  Sanitizer self-test of the CPU restatement ("race detection / sanitizers"):
// built by `make -C oracle asan` with -fsanitize=address,undefined and run by tests/test_oracle_sanitizers.py.  Drives the whole
// pipeline (scan registration -> odometry -> mapping), the image front-end and the trust-region solver through the C API on a small
// synthetic scene — a 16-line sensor inside a box room with two pillars, moving 0.4 m per sweep — so that every container the
// restatement indexes (ring tables, sector pick lists, kd-trees, voxel maps, cube grid, Jacobian blocks) sees real traffic under the
// sanitizers.  Exit code 0 and no sanitizer report == pass.  No reference source is involved.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

extern "C" {
struct orc_handle;
orc_handle* orc_create(int scan_line, double minimum_range, float line_res, float plane_res, int mapping_skip_frame, int detach_vo_lo, int with_mapping);
void orc_destroy(orc_handle* h);
int orc_process(orc_handle* h, const float* xyz_pad4, int n);
int orc_get_cloud(orc_handle* h, int which, float* buf, int cap);
void orc_get_lo_pose(orc_handle* h, double* q_w, double* t_w, double* q_lc, double* t_lc);
int orc_solve(const double* factors, int nf, int quaternion, double huber_a, int max_iters, double* p0, double* p1, double* trace, int cap_iters,
              int* n_iters, double* H0, double* g0, int* termination, double* costs2);
int orc_good_features(const unsigned char* img, int w, int h, int max_corners, double quality, double min_distance, int block_size, float* xy_out, int cap,
                      float* eig_out);
}

// first hit of a ray from `o` along `d` with the inside of an axis-aligned room and two square pillars
static double cast(const double o[3], const double d[3]) {
  const double lo[3] = {-12, -9, -1.7}, hi[3] = {14, 8, 3.5};
  double best = 1e9;
  for (int a = 0; a < 3; a++) {
    if (std::fabs(d[a]) < 1e-12) continue;
    const double t = ((d[a] > 0 ? hi[a] : lo[a]) - o[a]) / d[a];
    if (t > 0 && t < best) best = t;
  }
  const double px[2] = {5.0, -4.0}, py[2] = {3.0, -4.5};
  for (int p = 0; p < 2; p++)
    for (int a = 0; a < 2; a++) {
      if (std::fabs(d[a]) < 1e-12) continue;
      const double c = a == 0 ? px[p] : py[p];
      for (int s = -1; s <= 1; s += 2) {
        const double t = (c + 0.4 * s - o[a]) / d[a];
        if (t <= 0 || t >= best) continue;
        const double u = o[1 - a] + t * d[1 - a], cu = a == 0 ? py[p] : px[p];
        if (std::fabs(u - cu) <= 0.4) best = t;
      }
    }
  return best;
}

int main() {
  orc_handle* h = orc_create(16, 1.0, 0.4f, 0.8f, 1, 1, 1);
  const int rings = 16, cols = 360;
  std::vector<float> cloud;
  for (int sweep = 0; sweep < 4; sweep++) {
    cloud.clear();
    const double o[3] = {0.4 * sweep, 0.05 * sweep, 0.0};
    for (int c = 0; c < cols; c++)
      for (int r = 0; r < rings; r++) {
        const double az = -2.0 * M_PI * (c + 0.37 * r / rings) / cols, el = (-15.0 + 2.0 * r) * M_PI / 180.0;
        const double d[3] = {std::cos(el) * std::cos(az), std::cos(el) * std::sin(az), std::sin(el)};
        const double t = cast(o, d);
        if (t > 80) continue;
        cloud.push_back((float)(t * d[0])); cloud.push_back((float)(t * d[1])); cloud.push_back((float)(t * d[2])); cloud.push_back(0.f);
      }
    if (orc_process(h, cloud.data(), (int)(cloud.size() / 4)) != 0) { std::fprintf(stderr, "orc_process failed at sweep %d\n", sweep); return 2; }
  }
  double qw[4], tw[3], ql[4], tl[3];
  orc_get_lo_pose(h, qw, tw, ql, tl);
  std::vector<float> buf(4 * 100000);
  const int n_map = orc_get_cloud(h, 9, buf.data(), 100000) + orc_get_cloud(h, 10, buf.data(), 100000);
  std::printf("selftest: odometry t = %.3f %.3f %.3f, map points in the valid block = %d\n", tw[0], tw[1], tw[2], n_map);
  orc_destroy(h);
  if (!(std::fabs(tw[0] - 1.2) < 0.3) || n_map < 100) { std::fprintf(stderr, "implausible result\n"); return 3; }

  // the trust-region solver on a handful of edge / plane / plane-norm factors
  std::vector<double> f(16 * 12, 0.0);
  for (int i = 0; i < 12; i++) {
    double* r = &f[16 * i];
    const double p[3] = {3.0 * std::sin(1.3 * i), 2.0 * std::cos(0.7 * i), 1.0 + 0.3 * i};
    r[0] = i % 3; r[1] = p[0]; r[2] = p[1]; r[3] = p[2];
    if (i % 3 == 0) { r[4] = p[0] + 0.5; r[5] = p[1]; r[6] = p[2] + 0.02; r[7] = p[0] - 0.5; r[8] = p[1] + 0.01; r[9] = p[2]; }
    else if (i % 3 == 1) { r[4] = p[0] + 0.4; r[5] = p[1]; r[6] = p[2]; r[7] = p[0]; r[8] = p[1] + 0.5; r[9] = p[2]; r[10] = p[0] - 0.3; r[11] = p[1] - 0.4; r[12] = p[2] + 0.01; }
    else { r[4] = 0.0; r[5] = 0.6; r[6] = 0.8; r[7] = -(0.6 * p[1] + 0.8 * p[2]) + 0.01; }
  }
  double p0[4] = {0.02, -0.01, 0.03, 0.9993}, p1[3] = {0.1, -0.05, 0.02}, trace[8 * 64], H0[36], g0[6], costs[2];
  int ni = 0, term = 0;
  if (orc_solve(f.data(), 12, 1, 0.1, 8, p0, p1, trace, 64, &ni, H0, g0, &term, costs) != 0 || ni < 2) { std::fprintf(stderr, "orc_solve failed\n"); return 4; }

  // the image front-end on a synthetic texture
  const int W = 160, H = 96;
  std::vector<unsigned char> img((size_t)W * H);
  for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) img[(size_t)y * W + x] = (unsigned char)(128 + 60 * std::sin(0.31 * x) * std::cos(0.23 * y) + 40 * (((x / 9) + (y / 7)) & 1));
  std::vector<float> xy(2 * 1024);
  const int nc = orc_good_features(img.data(), W, H, 1024, 0.03, 7.5, 5, xy.data(), 1024, nullptr);
  std::printf("selftest: solver iterations %d (termination %d), corners %d\n", ni, term, nc);
  return nc > 0 ? 0 : 5;
}
