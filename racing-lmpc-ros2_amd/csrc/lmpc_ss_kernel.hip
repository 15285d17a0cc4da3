// lmpc_ss_kernel.hip -- LMPC safe-set query on gfx950: per-lap k-nearest neighbours in (s, e_y)
// + cost-to-go gather.  One wavefront per query.
//
// Restates SafeSetManager::query(SSQuery) (safe_set.cpp:153-180) over SSTrajectory::query
// (:42-54) and TrajectoryKDTree::find_closest_waypoint_indices (trajectory_kd_tree.cpp:53-63),
// with the +-L unrolling and cost-to-go of SSTrajectory::process_lap_data (:116-137) done
// arithmetically instead of being stored, and the pad / truncate / J - J[0] post-processing of
// RacingMPC::solve (racing_mpc.cpp:263-280) fused in.
// Index mode (ss_idx != NULL, round 5): instead of the 7 S doubles of (ss_x, ss_j) a query leaves S int32 codes
//   code = ((row of the point in the concatenated lap store) << 2) | rep        (rep = 0, 1, 2: the -L, 0, +L copy; -1: no point)
// and the learning kernel's prologue gathers the points from the L2-resident store itself (lmpc_solve_kernel.hip): 640 B per
// query instead of 8960 B written here and read back there, and 4-byte stores that merge 16 to a line instead of 8.
//   laps newest -> oldest while fewer than S points are collected; per lap the K points of the
//   3n-point unrolled lap [x - L e_0, x, x + L e_0] nearest to the query in (s, e_y), nearest
//   first (ties: lower unrolled index; CGAL's order for exact ties is unspecified); J of
//   unrolled index c = rep * n + j is (n-1-j) + (1-rep)(n-1)  (:122,128).
// The brute-force scan replaces CGAL's kd-tree: one pass over the 3n unrolled points of a lap leaves each lane with
// the two nearest of its strided share (all distances stay in LDS); the 64 lane minima are sorted across the wave
// (bitonic network, lexicographic in (distance, index)) and, unless some lane's runner-up beats the K-th of them, lanes
// 0..K-1 hold the lap's neighbours nearest first and write their points in parallel.  When a lane owns two winners
// (a lap revisiting a place within 64 samples) K rounds of a wave-wide arg-min pick them one at a time instead.
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
                                                           double* __restrict__ ss_j, int* __restrict__ n_found, double* __restrict__ j0_out,
                                                           int* __restrict__ ss_idx) {
#pragma clang fp contract(off)
  extern __shared__ __attribute__((aligned(16))) double dist[];
  // XCD-aware query assignment (as in the QP kernel): consecutive workgroups go round-robin to the 8 XCDs, so workgroup
  // w takes query (w mod 8) * ceil(B / 8) + w / 8 and the 8-byte results of neighbouring queries, which share 64-byte
  // lines of the [field][point][batch] arrays, are merged in one XCD's L2 instead of reaching HBM as partial lines
  const int b = (int)(blockIdx.x & 7) * ((B + 7) >> 3) + (int)(blockIdx.x >> 3), lane = threadIdx.x;
  if (b >= B) return;
  const double qs = query[b], qe = query[(size_t)B + b];
  int tot = 0;
  double last = 0.0;  // lane k < 6: component k of the last point written; lane 6: its J - J0
  double j0 = 0.0;
  int last_code = -1;  // (index mode) the code of the last point taken: what the padding repeats
  for (int l = n_laps - 1; l >= 0 && tot < S; --l) {
    const int n = npts[l], n3 = 3 * n;
    const double* xl = x + (size_t)off[l] * 6;
    __syncthreads();
    // one pass: distance of every unrolled point of the lane's strided share, kept in LDS for the (rare) rescan, and
    // the share's two smallest in registers.  Neighbours are close in index and the share is strided by 64, so a lane
    // seldom owns more than one of the K winners: its runner-up is promoted without touching LDS again.
    double bestd = INFINITY, secd = INFINITY;
    int besti = INT_MAX, seci = INT_MAX;  // seci: INT_MAX = the share has no further point, -1 = not known (rescan)
    // four points of the share per trip, their loads issued together (one point per trip left the pass waiting on a dependent
    // L2 round trip per point, behind an integer division for (rep, j): 62 us per query wave), (rep, j) stepped, not divided
    {
      int j = lane, rep = 0;
      while (j >= n) {
        j -= n;
        ++rep;
      }
      for (int c = lane; c < n3; c += 256) {
        double sv[4], ev[4];
        int rp[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const bool in = c + 64 * t < n3;
          const int jj = in ? j : 0;
          sv[t] = xl[(size_t)jj * 6];
          ev[t] = xl[(size_t)jj * 6 + 1];
          rp[t] = rep;
          j += 64;
          while (j >= n) {
            j -= n;
            ++rep;
          }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int ct = c + 64 * t;
          if (ct < n3) {
            const double s = sv[t] + (rp[t] - 1) * Lt;
            const double ds = s - qs, de = ev[t] - qe;
            const double d = ds * ds + de * de;
            dist[ct] = d;
            if (d < bestd) {
              secd = bestd;
              seci = besti;
              bestd = d;
              besti = ct;
            } else if (d < secd) {
              secd = d;
              seci = ct;
            }
          }
        }
      }
    }
    int take = K < n3 ? K : n3;
    if (take > S - tot) take = S - tot;
    // Fast path.  Sort the 64 lane minima (bitonic network over the lanes, lexicographic in (distance, index)): if no
    // lane's runner-up beats the take-th of them, the first `take` lanes now hold the lap's neighbours nearest first,
    // and every lane writes its own point.  Otherwise (a lane owning two of the winners: laps with repeated or
    // crawling samples) the rounds below pick them one at a time.
    if (take <= 64) {
      double d = bestd;
      int i = besti;
#pragma unroll
      for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int jj = k >> 1; jj > 0; jj >>= 1) {
          const double od = __shfl_xor(d, jj, 64);
          const int oi = __shfl_xor(i, jj, 64);
          const bool other_less = od < d || (od == d && oi < i);
          const bool keep_min = ((lane & jj) == 0) == ((lane & k) == 0);
          if (keep_min ? other_less : !other_less) {
            d = od;
            i = oi;
          }
        }
      }
      const double td = __shfl(d, take - 1, 64);
      const int ti = __shfl(i, take - 1, 64);
      const bool beaten = seci >= 0 && seci != INT_MAX && (secd < td || (secd == td && seci < ti));
      if (!__any(beaten) && ti != INT_MAX) {
        const bool mine = lane < take;
        const int ii = mine ? i : 0;
        const int rep = ii / n, j = ii - rep * n;
        const double jv = (double)(n - 1 - j) + (1 - rep) * (double)(n - 1);
        if (tot == 0) j0 = __shfl(jv, 0, 64);
        if (mine) {
          if (ss_idx) {
            ss_idx[(size_t)(tot + lane) * B + b] = ((off[l] + j) << 2) | rep;
          } else {
#pragma unroll
            for (int k = 0; k < 6; ++k)
              ss_x[((size_t)k * S + tot + lane) * B + b] = xl[(size_t)j * 6 + k] + (k == 0 ? (rep - 1) * Lt : 0.0);
            ss_j[(size_t)(tot + lane) * B + b] = jv - j0;
          }
        }
        // the last point written, as the padding below wants it: component k on lane k < 6, J - J0 on lane 6
        const int il = __shfl(i, take - 1, 64);
        const int repl = il / n, jl = il - repl * n;
        if (lane < 6)
          last = xl[(size_t)jl * 6 + lane] + (lane == 0 ? (repl - 1) * Lt : 0.0);
        else if (lane == 6)
          last = ((double)(n - 1 - jl) + (1 - repl) * (double)(n - 1)) - j0;
        last_code = ((off[l] + jl) << 2) | repl;
        tot += take;
        continue;
      }
    }
    for (int q = 0; q < take; ++q, ++tot) {
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
      if (i == INT_MAX) {  // no finite distance left (a NaN query from a diverged car state): nothing more to take
        break;
      }
      const int rep = i / n, j = i - rep * n;
      const double jv = (double)(n - 1 - j) + (1 - rep) * (double)(n - 1);
      if (tot == 0) j0 = jv;
      last_code = ((off[l] + j) << 2) | rep;
      if (ss_idx) {
        if (lane == 0) ss_idx[(size_t)tot * B + b] = last_code;
      } else if (lane < 6) {
        last = xl[(size_t)j * 6 + lane] + (lane == 0 ? (rep - 1) * Lt : 0.0);
        ss_x[((size_t)lane * S + tot) * B + b] = last;
      } else if (lane == 6) {
        last = jv - j0;
        ss_j[(size_t)tot * B + b] = last;
      }
      if ((i & 63) == lane) {  // the winner's owner retires it and moves on to its runner-up
        dist[i] = INFINITY;
        if (seci >= 0) {
          bestd = secd;
          besti = seci;
          secd = INFINITY;
          seci = besti == INT_MAX ? INT_MAX : -1;
        } else {  // second win in a row without a known runner-up: rescan the share for its two smallest
          bestd = secd = INFINITY;
          besti = seci = INT_MAX;
          for (int c = lane; c < n3; c += 64) {
            const double dd = dist[c];
            if (dd < bestd) {
              secd = bestd;
              seci = besti;
              bestd = dd;
              besti = c;
            } else if (dd < secd) {
              secd = dd;
              seci = c;
            }
          }
        }
      }
    }
  }
  if (lane == 0) {
    n_found[b] = tot;
    if (j0_out) j0_out[b] = j0;  // cost-to-go of the first point, subtracted from ss_j (racing_mpc.cpp:280)
  }
  // pad with the last point (racing_mpc.cpp:263-272).  With nothing found (no lap stored, or a NaN query) the reference
  // keeps its previous parameter values; a batch has no "previous", so the outputs are zero-filled -- defined data -- and
  // n_found = 0 tells the caller not to solve on them.
  if (tot == 0) last = 0.0;
  if (ss_idx) {
    for (int q = tot + lane; q < S; q += 64) ss_idx[(size_t)q * B + b] = last_code;
    return;
  }
  for (int q = tot; q < S; ++q) {
    if (lane < 6)
      ss_x[((size_t)lane * S + q) * B + b] = last;
    else if (lane == 6)
      ss_j[(size_t)q * B + b] = last;
  }
}
