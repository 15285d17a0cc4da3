// lmpc_ss_kernel.hip -- LMPC safe-set query on gfx950: per-lap k-nearest neighbours in (s, e_y)
// + cost-to-go gather.  One wavefront per query.
//
// Restates SafeSetManager::query(SSQuery) (safe_set.cpp:153-180) over SSTrajectory::query
// (:42-54) and TrajectoryKDTree::find_closest_waypoint_indices (trajectory_kd_tree.cpp:53-63),
// with the +-L unrolling and cost-to-go of SSTrajectory::process_lap_data (:116-137) done
// arithmetically instead of being stored, and the pad / truncate / J - J[0] post-processing of
// RacingMPC::solve (racing_mpc.cpp:263-280) fused in.
//   laps newest -> oldest while fewer than S points are collected; per lap the K points of the
//   3n-point unrolled lap [x - L e_0, x, x + L e_0] nearest to the query in (s, e_y), nearest
//   first (ties: lower unrolled index; CGAL's order for exact ties is unspecified); J of
//   unrolled index c = rep * n + j is (n-1-j) + (1-rep)(n-1)  (:122,128).
// The brute-force scan replaces CGAL's kd-tree: 3n distances per lap live in LDS, each lane keeps
// the best of its strided share, and K rounds of a wave-wide arg-min pick the neighbours in order.
#include <hip/hip_runtime.h>

#include <limits.h>

// one DPP step of the arg-min: lanes outside ROW_MASK see the identity (+inf, INT_MAX)
#define ARGMIN_STEP(CTRL, ROW_MASK)                                                                                      \
  {                                                                                                                      \
    const int od_lo = __builtin_amdgcn_update_dpp(0, __double2loint(d), CTRL, ROW_MASK, 0xf, false);                     \
    const int od_hi = __builtin_amdgcn_update_dpp(0x7ff00000, __double2hiint(d), CTRL, ROW_MASK, 0xf, false);            \
    const int oi = __builtin_amdgcn_update_dpp(INT_MAX, i, CTRL, ROW_MASK, 0xf, false);                                  \
    const double od = __hiloint2double(od_hi, od_lo);                                                                    \
    if (od < d || (od == d && oi < i)) {                                                                                 \
      d = od;                                                                                                            \
      i = oi;                                                                                                            \
    }                                                                                                                    \
  }

__global__ __launch_bounds__(64) void lmpc_ss_query_kernel(int B, int n_laps, int S, int K,
                                                           const int* __restrict__ npts, const int* __restrict__ off,
                                                           const double* __restrict__ x, double Lt,
                                                           const double* __restrict__ query, double* __restrict__ ss_x,
                                                           double* __restrict__ ss_j, int* __restrict__ n_found, double* __restrict__ j0_out) {
#pragma clang fp contract(off)
  extern __shared__ __attribute__((aligned(16))) double dist[];
  const int b = blockIdx.x, lane = threadIdx.x;
  const double qs = query[b], qe = query[(size_t)B + b];
  int tot = 0;
  double last = 0.0;  // lane k < 6: component k of the last point written; lane 6: its J - J0
  double j0 = 0.0;
  for (int l = n_laps - 1; l >= 0 && tot < S; --l) {
    const int n = npts[l], n3 = 3 * n;
    const double* xl = x + (size_t)off[l] * 6;
    __syncthreads();
    for (int c = lane; c < n3; c += 64) {
      const int rep = c / n, j = c - rep * n;
      const double s = xl[(size_t)j * 6] + (rep - 1) * Lt;
      const double ds = s - qs, de = xl[(size_t)j * 6 + 1] - qe;
      dist[c] = ds * ds + de * de;
    }
    __syncthreads();
    double bestd = INFINITY;
    int besti = INT_MAX;
    for (int c = lane; c < n3; c += 64) {
      const double d = dist[c];
      if (d < bestd) {
        bestd = d;
        besti = c;
      }
    }
    const int take = K < n3 ? K : n3;
    for (int q = 0; q < take && tot < S; ++q, ++tot) {
      // wave-wide lexicographic arg-min of (distance, unrolled index) on the VALU: four row_ror steps give every
      // lane of a 16-lane row the row's winner, row_bcast15 / row_bcast31 fold the rows into lane 63
      double d = bestd;
      int i = besti;
      ARGMIN_STEP(0x128, 0xf)
      ARGMIN_STEP(0x124, 0xf)
      ARGMIN_STEP(0x122, 0xf)
      ARGMIN_STEP(0x121, 0xf)
      ARGMIN_STEP(0x142, 0xa)
      ARGMIN_STEP(0x143, 0xc)
      d = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(d), 63), __builtin_amdgcn_readlane(__double2loint(d), 63));
      i = __builtin_amdgcn_readlane(i, 63);
      const int rep = i / n, j = i - rep * n;
      const double jv = (double)(n - 1 - j) + (1 - rep) * (double)(n - 1);
      if (tot == 0) j0 = jv;
      if (lane < 6) {
        last = xl[(size_t)j * 6 + lane] + (lane == 0 ? (rep - 1) * Lt : 0.0);
        ss_x[((size_t)lane * S + tot) * B + b] = last;
      } else if (lane == 6) {
        last = jv - j0;
        ss_j[(size_t)tot * B + b] = last;
      }
      if ((i & 63) == lane) {  // the winner's owner retires it and rescans its share
        dist[i] = INFINITY;
        bestd = INFINITY;
        besti = INT_MAX;
        for (int c = lane; c < n3; c += 64) {
          const double dd = dist[c];
          if (dd < bestd) {
            bestd = dd;
            besti = c;
          }
        }
      }
    }
  }
  if (lane == 0) {
    n_found[b] = tot;
    if (j0_out) j0_out[b] = j0;  // cost-to-go of the first point, subtracted from ss_j (racing_mpc.cpp:280)
  }
  if (tot > 0)  // pad with the last point (racing_mpc.cpp:263-272)
    for (int q = tot; q < S; ++q) {
      if (lane < 6)
        ss_x[((size_t)lane * S + q) * B + b] = last;
      else if (lane == 6)
        ss_j[(size_t)q * B + b] = last;
    }
}
