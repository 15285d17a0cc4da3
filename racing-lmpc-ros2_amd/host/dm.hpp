// dm.hpp -- the dense matrix the facade passes around in place of casadi::DM (CasADi is not a dependency).
#ifndef LMPC_HOST_DM_HPP_
#define LMPC_HOST_DM_HPP_

#include <cstddef>
#include <map>
#include <string>
#include <vector>

namespace lmpc {

// Dense column-major matrix of doubles: element (r, c) at data[c * rows + r], as casadi::DM stores it.
struct DM {
  std::size_t rows = 0, cols = 0;
  std::vector<double> data;
  DM() = default;
  DM(std::size_t r, std::size_t c, double fill = 0.0) : rows(r), cols(c), data(r * c, fill) {}
  explicit DM(double scalar) : rows(1), cols(1), data(1, scalar) {}
  double& operator()(std::size_t r, std::size_t c) { return data[c * rows + r]; }
  double operator()(std::size_t r, std::size_t c) const { return data[c * rows + r]; }
  std::size_t size1() const { return rows; }
  std::size_t size2() const { return cols; }
  explicit operator double() const { return data.at(0); }
  // casadi::DM::horzcat({*this, col}) for a column of matching height (an empty matrix takes the column's height)
  void append_column(const DM& col) {
    if (cols == 0) rows = col.rows;
    data.insert(data.end(), col.data.begin(), col.data.begin() + rows);
    ++cols;
  }
};
typedef std::map<std::string, DM> DMDict;
typedef std::map<std::string, double> Dict;  // the node reads only stats["iter_count"] (racing_mpc_node.cpp:355-357)

}  // namespace lmpc
#endif
