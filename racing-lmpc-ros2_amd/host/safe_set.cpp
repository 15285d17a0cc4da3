// safe_set.cpp -- see safe_set.hpp.  Plain C++17; links the C ABI only.
#include "safe_set.hpp"

#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>

namespace lmpc {
namespace vehicle_model {
namespace racing_trajectory {

DM read_txt(const std::string& path) {
  std::ifstream f(path);
  if (!f) throw std::runtime_error("cannot open " + path);
  std::vector<std::vector<double>> rows;
  std::string line;
  while (std::getline(f, line)) {
    std::istringstream ls(line);
    std::vector<double> r;
    double v;
    while (ls >> v) r.push_back(v);
    if (!r.empty()) rows.push_back(r);
  }
  if (rows.empty()) return DM();
  DM m(rows.size(), rows[0].size());
  for (std::size_t i = 0; i < rows.size(); ++i) {
    if (rows[i].size() != m.cols) throw std::runtime_error("ragged matrix in " + path);
    for (std::size_t j = 0; j < m.cols; ++j) m(i, j) = rows[i][j];
  }
  return m;
}

void write_txt(const DM& m, const std::string& path) {
  std::FILE* f = std::fopen(path.c_str(), "w");
  if (!f) throw std::runtime_error("cannot write " + path);
  for (std::size_t i = 0; i < m.rows; ++i) {
    for (std::size_t j = 0; j < m.cols; ++j) std::fprintf(f, j ? "  % .16e" : "% .16e", m(i, j));
    std::fputc('\n', f);
  }
  std::fclose(f);
}

static DM transpose(const DM& a) {
  DM t(a.cols, a.rows);
  for (std::size_t i = 0; i < a.rows; ++i)
    for (std::size_t j = 0; j < a.cols; ++j) t(j, i) = a(i, j);
  return t;
}

SafeSetManager::SafeSetManager(lmpc_handle* handle, const std::size_t& max_lap_stored)
    : h_(handle), max_lap_stored_(max_lap_stored) {}

void SafeSetManager::add_lap(const DM& x, const DM& u, const DM& k, const DM& t, const double& total_length) {
  if (x.rows != 6 || x.cols < 1) throw std::invalid_argument("SafeSetManager::add_lap: x must be 6 x n");
  laps_.push_back(Lap{x, u, k, t});
  while (max_lap_stored_ > 0 && laps_.size() > max_lap_stored_) laps_.pop_front();  // circular_buffer (safe_set.cpp:139-151)
  std::vector<int32_t> n_pts;
  std::vector<double> flat;  // [n][6] row-major = the 6 x n column-major lap as it is
  for (const auto& lap : laps_) {
    n_pts.push_back(static_cast<int32_t>(lap.x.cols));
    flat.insert(flat.end(), lap.x.data.begin(), lap.x.data.end());
  }
  if (lmpc_set_safe_set(h_, static_cast<int32_t>(laps_.size()), n_pts.data(), flat.data(), total_length) != LMPC_OK)
    throw std::runtime_error(std::string("SafeSetManager::add_lap: ") + lmpc_last_error(h_));
}

SSResult SafeSetManager::query(const SSQuery& query) {
  SSResult res;
  if (laps_.empty() || query.max_num_total == 0) return res;
  const std::size_t S = query.max_num_total;
  std::vector<double> sx(6 * S), sj(S);
  int32_t n_found = 0;
  double j0 = 0.0;
  const double q[2] = {query.x(0, 0), query.x(1, 0)};  // only (s, e_y) enter the search (safe_set.cpp:47-50)
  if (lmpc_ss_query_host(h_, q, sx.data(), sj.data(), &n_found, &j0) != LMPC_OK)
    throw std::runtime_error(std::string("SafeSetManager::query: ") + lmpc_last_error(h_));
  res.x = DM(6, static_cast<std::size_t>(n_found));
  res.J = DM(1, static_cast<std::size_t>(n_found));
  for (int32_t j = 0; j < n_found; ++j) {
    for (int k = 0; k < 6; ++k) res.x(k, j) = sx[(std::size_t)j * 6 + k];
    res.J(0, j) = sj[j] + j0;  // the device result is J - J[0] (racing_mpc.cpp:280); the query result is J
  }
  return res;
}

SafeSetRecorder::SafeSetRecorder(SafeSetManager& manager, const bool& to_file, const std::string& file_prefix)
    : manager_(manager), to_file_(to_file), file_prefix_(file_prefix) {}

// Lap files (safe_set.cpp:260-276 upstream): `<stem>_{x,u,k,t}.txt`, one sample per row.  A lap that cannot be read is
// reported and skipped; the others still load.
void SafeSetRecorder::load(const std::vector<std::string>& from_files, const double& total_length) {
  static const char* const kinds[4] = {"_x.txt", "_u.txt", "_k.txt", "_t.txt"};
  for (const std::string& stem : from_files) {
    std::cout << "Loading lap from " << stem << std::endl;
    try {
      DM part[4];
      for (int i = 0; i < 4; ++i) part[i] = transpose(read_txt(stem + kinds[i]));
      manager_.add_lap(part[0], part[1], part[2], part[3], total_length);
      ++lap_count_;
    } catch (const std::exception& err) {
      std::cout << "Failed to load lap from " << stem << std::endl << err.what() << std::endl;
    }
  }
}

void SafeSetRecorder::Lap::restart(const DM& x0, const DM& u0, const DM& k0, const DM& t0) {
  x = x0;
  u = u0;
  k = k0;
  t = t0;
}

void SafeSetRecorder::Lap::push(const DM& xi, const DM& ui, const DM& ki, const DM& ti) {
  x.append_column(xi);
  u.append_column(ui);
  k.append_column(ki);
  t.append_column(ti);
}

void SafeSetRecorder::finish_lap(double total_length) {
  std::cout << "Lap " << lap_count_ << " completed. Adding to safe set." << std::endl;
  manager_.add_lap(lap_.x, lap_.u, lap_.k, lap_.t, total_length);
  if (!to_file_) return;
  const std::string stem = file_prefix_ + "lap_" + std::to_string(lap_count_);
  std::cout << "Saving lap to " << stem << std::endl;
  write_txt(transpose(lap_.x), stem + "_x.txt");
  write_txt(transpose(lap_.u), stem + "_u.txt");
  write_txt(transpose(lap_.t), stem + "_t.txt");
  write_txt(transpose(lap_.k), stem + "_k.txt");
}

// One control period (safe_set.cpp:278-322 upstream).  Laps are cut where the abscissa falls by more than half the track
// length from one sample to the next.  The stretch before the first crossing is a partial lap and is dropped; the lap
// counter counts crossings, so the first complete lap is number 1 -- file names and console lines as upstream.
void SafeSetRecorder::step(const DM& x, const DM& u, const DM& k, const DM& t, const double& total_length) {
  if (!seeded_) {  // the first sample only provides the abscissa to compare the second one with
    lap_.x = x;
    seeded_ = true;
    return;
  }
  const bool crossed_line = lap_.last_abscissa() - x(0, 0) > 0.5 * total_length;
  if (!crossed_line) {
    lap_.push(x, u, k, t);
    return;
  }
  if (past_first_line_) finish_lap(total_length);
  past_first_line_ = true;
  ++lap_count_;
  lap_.restart(x, u, k, t);
}

}  // namespace racing_trajectory
}  // namespace vehicle_model
}  // namespace lmpc
