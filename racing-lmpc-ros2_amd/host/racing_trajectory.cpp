// racing_trajectory.cpp -- see racing_trajectory.hpp.
#include "racing_trajectory.hpp"

#include <cmath>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace lmpc {
namespace vehicle_model {
namespace racing_trajectory {

namespace {
double align_abscissa(double s1, double s2, double s_total) {  // lmpc_utils/utils.hpp:35-41
  const double k = std::fabs(s2 - s1) + s_total / 2.0;
  const double l = k - std::fmod(k, s_total);
  const double d = s2 - s1;
  return s1 + l * ((d > 0) - (d < 0));
}
double align_yaw(double yaw_1, double yaw_2) {  // utils.hpp:25-31
  const double d = yaw_1 - yaw_2;
  return std::atan2(std::sin(d), std::cos(d)) + yaw_2;
}
}  // namespace

NotAKnotCubic::NotAKnotCubic(const std::vector<double>& x, const std::vector<double>& y) : x_(x) {
  const std::size_t n = x.size();
  if (n < 4 || y.size() != n) throw std::invalid_argument("NotAKnotCubic: need >= 4 points");
  std::vector<double> h(n - 1), dl(n - 1);
  for (std::size_t i = 0; i + 1 < n; ++i) {
    h[i] = x[i + 1] - x[i];
    if (!(h[i] > 0)) throw std::invalid_argument("NotAKnotCubic: abscissae must increase");
    dl[i] = (y[i + 1] - y[i]) / h[i];
  }
  // unknowns: second derivatives m_i.  Interior rows: C1 continuity; end rows: third derivative continuous across
  // x_1 and x_{n-2}.  Dense Gaussian elimination with partial pivoting (n ~ 160).
  std::vector<double> A(n * n, 0.0), r(n, 0.0);
  for (std::size_t i = 1; i + 1 < n; ++i) {
    A[i * n + i - 1] = h[i - 1];
    A[i * n + i] = 2.0 * (h[i - 1] + h[i]);
    A[i * n + i + 1] = h[i];
    r[i] = 6.0 * (dl[i] - dl[i - 1]);
  }
  A[0] = h[1]; A[1] = -(h[0] + h[1]); A[2] = h[0];
  A[(n - 1) * n + n - 3] = h[n - 2]; A[(n - 1) * n + n - 2] = -(h[n - 3] + h[n - 2]); A[(n - 1) * n + n - 1] = h[n - 3];
  for (std::size_t c = 0; c < n; ++c) {
    std::size_t p = c;
    for (std::size_t i = c + 1; i < n; ++i)
      if (std::fabs(A[i * n + c]) > std::fabs(A[p * n + c])) p = i;
    if (p != c) {
      for (std::size_t j = 0; j < n; ++j) std::swap(A[p * n + j], A[c * n + j]);
      std::swap(r[p], r[c]);
    }
    for (std::size_t i = c + 1; i < n; ++i) {
      const double f = A[i * n + c] / A[c * n + c];
      if (f == 0.0) continue;
      for (std::size_t j = c; j < n; ++j) A[i * n + j] -= f * A[c * n + j];
      r[i] -= f * r[c];
    }
  }
  std::vector<double> m(n);
  for (std::size_t ii = n; ii-- > 0;) {
    double t = r[ii];
    for (std::size_t j = ii + 1; j < n; ++j) t -= A[ii * n + j] * m[j];
    m[ii] = t / A[ii * n + ii];
  }
  a_.resize(n - 1); b_.resize(n - 1); c_.resize(n - 1); d_.resize(n - 1);
  for (std::size_t i = 0; i + 1 < n; ++i) {
    a_[i] = y[i];
    b_[i] = dl[i] - h[i] * (2.0 * m[i] + m[i + 1]) / 6.0;
    c_[i] = m[i] / 2.0;
    d_[i] = (m[i + 1] - m[i]) / (6.0 * h[i]);
  }
}

double NotAKnotCubic::operator()(double xq, int nu) const {
  std::size_t lo = 0, hi = x_.size() - 1;  // last interval whose left end is <= xq (end pieces extrapolate)
  while (hi - lo > 1) {
    const std::size_t mid = (lo + hi) / 2;
    if (x_[mid] <= xq) lo = mid; else hi = mid;
  }
  const double h = xq - x_[lo];
  if (nu == 0) return a_[lo] + h * (b_[lo] + h * (c_[lo] + h * d_[lo]));
  if (nu == 1) return b_[lo] + h * (2.0 * c_[lo] + 3.0 * h * d_[lo]);
  return 2.0 * c_[lo] + 6.0 * h * d_[lo];
}

static DM read_table_transposed(const std::string& file_name) {
  std::ifstream f(file_name);
  if (!f) throw std::runtime_error("cannot open " + file_name);
  std::vector<std::vector<double>> rows;
  std::string line;
  while (std::getline(f, line)) {
    std::istringstream ls(line);
    std::vector<double> r;
    double v;
    while (ls >> v) r.push_back(v);
    if (!r.empty()) rows.push_back(r);
  }
  if (rows.empty() || rows[0].size() != 17) throw std::runtime_error(file_name + ": not a 17-column track table");
  DM m(17, rows.size());
  for (std::size_t j = 0; j < rows.size(); ++j) {
    if (rows[j].size() != 17) throw std::runtime_error(file_name + ": ragged table");
    for (std::size_t i = 0; i < 17; ++i) m(i, j) = rows[j][i];
  }
  return m;
}

RacingTrajectory::RacingTrajectory(const std::string& file_name) : RacingTrajectory(read_table_transposed(file_name)) {}

RacingTrajectory::RacingTrajectory(const DM& traj) : traj_(traj) {
  if (traj_.rows != 17 || traj_.cols < 8) throw std::invalid_argument("RacingTrajectory: 17 x n table expected");
  const std::size_t n = traj_.cols;
  total_length_ = traj_(DIST_TO_SF_FWD, 0);
  // the last three waypoints in front (-L), all of them, the first four behind (+L)   (racing_trajectory.cpp:48-59)
  std::vector<std::size_t> idx;
  std::vector<double> shift;
  for (std::size_t j = n - 3; j < n; ++j) { idx.push_back(j); shift.push_back(-total_length_); }
  for (std::size_t j = 0; j < n; ++j) { idx.push_back(j); shift.push_back(0.0); }
  for (std::size_t j = 0; j < 4; ++j) { idx.push_back(j); shift.push_back(total_length_); }
  std::vector<double> s, px, py, v, l, r;
  for (std::size_t k = 0; k < idx.size(); ++k) {
    const std::size_t j = idx[k];
    s.push_back(traj_(DIST_TO_SF_BWD, j) + shift[k]);
    px.push_back(traj_(PX, j));
    py.push_back(traj_(PY, j));
    v.push_back(traj_(SPEED, j));
    l.push_back(std::hypot(traj_(PX, j) - traj_(LEFT_BOUND_X, j), traj_(PY, j) - traj_(LEFT_BOUND_Y, j)));
    r.push_back(-std::hypot(traj_(PX, j) - traj_(RIGHT_BOUND_X, j), traj_(PY, j) - traj_(RIGHT_BOUND_Y, j)));
  }
  x_ = NotAKnotCubic(s, px);
  y_ = NotAKnotCubic(s, py);
  vel_ = NotAKnotCubic(s, v);
  left_ = NotAKnotCubic(s, l);
  right_ = NotAKnotCubic(s, r);
}

double RacingTrajectory::mod(double s) const { return align_abscissa(s, total_length_ / 2.0, total_length_); }
double RacingTrajectory::x_interpolation(double s) const { return x_(mod(s)); }
double RacingTrajectory::y_interpolation(double s) const { return y_(mod(s)); }
double RacingTrajectory::velocity_interpolation(double s) const { return vel_(mod(s)); }
double RacingTrajectory::left_boundary_interpolation(double s) const { return left_(mod(s)); }
double RacingTrajectory::right_boundary_interpolation(double s) const { return right_(mod(s)); }
double RacingTrajectory::yaw_interpolation(double s) const {
  const double sm = mod(s);
  return std::atan2(y_(sm, 1), x_(sm, 1));
}
double RacingTrajectory::curvature_interpolation(double s) const {
  const double sm = mod(s);
  const double dx = x_(sm, 1), dy = y_(sm, 1), d2x = x_(sm, 2), d2y = y_(sm, 2);
  return dx * d2y - dy * d2x / std::sqrt(std::pow(dx * dx + dy * dy, 3));  // as written upstream
}

void RacingTrajectory::frenet_to_global(const FrenetPose2D& fp, Pose2D& gp) const {
  const double yaw0 = yaw_interpolation(fp.position.s);
  gp.position.x = x_interpolation(fp.position.s) - std::sin(yaw0) * fp.position.t;
  gp.position.y = y_interpolation(fp.position.s) + std::cos(yaw0) * fp.position.t;
  gp.yaw = align_yaw(yaw0 + fp.yaw, 0.0);
}

void RacingTrajectory::global_to_frenet(const Pose2D& gp, FrenetPose2D& fp, const bool& initialize_with_previous) const {
  const double x = gp.position.x, y = gp.position.y;
  double s0;
  if (initialize_with_previous) {
    s0 = fp.position.s;
  } else {  // closest waypoint (the kd-tree lookup upstream, racing_trajectory.cpp:213-216)
    std::size_t best = 0;
    double bd = INFINITY;
    for (std::size_t j = 0; j < traj_.cols; ++j) {
      const double d = (traj_(PX, j) - x) * (traj_(PX, j) - x) + (traj_(PY, j) - y) * (traj_(PY, j) - y);
      if (d < bd) { bd = d; best = j; }
    }
    s0 = traj_(DIST_TO_SF_BWD, best);
  }
  // minimise the squared distance to the centre line over the abscissa (the reference: CasADi sqpmethod on the same
  // scalar objective, :144-169): safeguarded Newton
  double s = mod(s0);
  auto f = [&](double sv) { return (x_(sv) - x) * (x_(sv) - x) + (y_(sv) - y) * (y_(sv) - y); };
  for (int it = 0; it < 50; ++it) {
    const double ex = x_(s) - x, ey = y_(s) - y, dx = x_(s, 1), dy = y_(s, 1), d2x = x_(s, 2), d2y = y_(s, 2);
    const double g = 2.0 * (ex * dx + ey * dy), H = 2.0 * (dx * dx + dy * dy + ex * d2x + ey * d2y);
    const double step = H > 1e-12 ? -g / H : -g / (2.0 * (dx * dx + dy * dy));
    const double f0 = f(s);
    double a = 1.0;
    while (f(s + a * step) > f0 && a > 1e-6) a *= 0.5;
    s += a * step;
    if (std::fabs(a * step) < 1e-12) break;
  }
  const double so = mod(s), xo = x_(so), yo = y_(so), yawo = std::atan2(y_(so, 1), x_(so, 1));
  const double sg = std::cos(yawo) * (y - yo) - std::sin(yawo) * (x - xo);  // lateral_sign (utils.hpp)
  fp.position.s = so;
  fp.position.t = std::hypot(x - xo, y - yo) * ((sg > 0) - (sg < 0));
  fp.yaw = align_yaw(gp.yaw, yawo) - yawo;
}

void RacingTrajectory::to_track_table(std::size_t M, std::vector<double>& curvature, std::vector<double>& bound_left,
                                      std::vector<double>& bound_right, std::vector<double>& vel) const {
  curvature.resize(M); bound_left.resize(M); bound_right.resize(M); vel.resize(M);
  for (std::size_t j = 0; j < M; ++j) {
    const double s = total_length_ * double(j) / double(M);
    curvature[j] = curvature_interpolation(s);
    bound_left[j] = left_boundary_interpolation(s);
    bound_right[j] = right_boundary_interpolation(s);
    vel[j] = velocity_interpolation(s);
  }
}

}  // namespace racing_trajectory
}  // namespace vehicle_model
}  // namespace lmpc
