// Geometry kernels (fp64): fisheye undistortion, depth lookup + unprojection, two-view DLT.
// One lane per keypoint (pair); the work per frame is tens of points, so these are latency-bound
// and exist to keep the whole frame->3D path on the device (no host round trip between NMS
// and the all-gather of 3D keypoints).
//
// Restated algorithms (the reference calls OpenCV, which is not vendored in the reference tree;
// its env pins opencv=3.4.2, corner_net_lite/conda_packagelist.txt:53):
//   cv2.fisheye.undistortPoints(P=K): normalise, clip theta_d to [-pi/2, pi/2], 10 fixed-point
//     iterations theta <- theta_d / (1 + k1 t^2 + k2 t^4 + k3 t^6 + k4 t^8), scale = tan(theta)/theta_d
//   cv2.triangulatePoints: per point the 4x4 system rows x*P[2]-P[0], y*P[2]-P[1] for both views,
//     solution = right singular vector of the smallest singular value (here: one-sided Jacobi SVD)
//   cv2.correctMatches: Hartley-Sturm optimal correction (Hartley & Zisserman alg. 12.1)
#include "okp_internal.h"

namespace {

__device__ __forceinline__ void fisheye_undistort(const okp_camera& cam, double x, double y, double& xu, double& yu) {
  const double pwx = (x - cam.cx) / cam.fx, pwy = (y - cam.cy) / cam.fy;
  double theta_d = sqrt(pwx * pwx + pwy * pwy);
  const double half_pi = 1.5707963267948966;
  theta_d = fmin(fmax(-half_pi, theta_d), half_pi);
  double scale = 1.0;
  if (theta_d > 1e-8) {
    double theta = theta_d;
    for (int j = 0; j < 10; ++j) {
      const double t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t6 * t2;
      theta = theta_d / (1.0 + cam.d[0] * t2 + cam.d[1] * t4 + cam.d[2] * t6 + cam.d[3] * t8);
    }
    scale = tan(theta) / theta_d;
  }
  xu = cam.fx * (pwx * scale) + cam.cx;   // R = I, P = K
  yu = cam.fy * (pwy * scale) + cam.cy;
}

// cv2.undistortPoints(xy, K, D, P=K) of OpenCV 3.4 for the plumb-bob model with D = (k1, k2, p1, p2): five fixed-point iterations
//   x <- (x0 - (2 p1 x y + p2 (r^2 + 2 x^2))) / (1 + k1 r^2 + k2 r^4),  y likewise with p1 (r^2 + 2 y^2) + 2 p2 x y
__device__ __forceinline__ void radtan_undistort(const okp_camera& cam, double u, double v, double& xu, double& yu) {
  const double x0 = (u - cam.cx) / cam.fx, y0 = (v - cam.cy) / cam.fy;
  const double k1 = cam.d[0], k2 = cam.d[1], p1 = cam.d[2], p2 = cam.d[3];
  double x = x0, y = y0;
  for (int j = 0; j < 5; ++j) {
    const double r2 = x * x + y * y;
    const double icdist = 1.0 / (1.0 + (k2 * r2 + k1) * r2);
    const double dx = 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x);
    const double dy = p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y;
    x = (x0 - dx) * icdist;
    y = (y0 - dy) * icdist;
  }
  xu = cam.fx * x + cam.cx;
  yu = cam.fy * y + cam.cy;
}

__device__ __forceinline__ void camera_undistort(const okp_camera& cam, double x, double y, double& xu, double& yu) {
  if (cam.model == OKP_CAM_RADTAN) radtan_undistort(cam, x, y, xu, yu);
  else fisheye_undistort(cam, x, y, xu, yu);
}

__global__ void okp_undistort_kernel(okp_camera cam, const float* __restrict__ xy, int m, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  double xu, yu;
  camera_undistort(cam, (double)xy[2 * i], (double)xy[2 * i + 1], xu, yu);
  out[2 * i] = xu;
  out[2 * i + 1] = yu;
}

__global__ void okp_unproject_kernel(okp_camera cam, const float* __restrict__ xy, const int* __restrict__ map_id, int m,
                                     const float* __restrict__ depth, int H, int W, int max_x, int max_y,
                                     double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const int map = map_id[i];
  if (map < 0) {
    const double nan = __builtin_nan("");
    out[3 * i] = nan; out[3 * i + 1] = nan; out[3 * i + 2] = nan;
    return;
  }
  // cv2 returns the dtype it was given: the reference passes float32 peaks, so the undistorted
  // point is rounded to float32 before it is used (pipeline.py:167-171).
  double xu, yu;
  camera_undistort(cam, (double)xy[2 * i], (double)xy[2 * i + 1], xu, yu);
  const float xf = (float)xu, yf = (float)yu;
  int xi = (int)rintf(xf), yi = (int)rintf(yf);        // numpy round = half to even
  xi = min(max(xi, 0), max_x);
  yi = min(max(yi, 0), max_y);
  const double z = (double)depth[((size_t)map * H + yi) * W + xi];
  // K^-1 [x y 1]^T * z for an upper-triangular K without skew
  out[3 * i + 0] = ((double)xf - cam.cx) / cam.fx * z;
  out[3 * i + 1] = ((double)yf - cam.cy) / cam.fy * z;
  out[3 * i + 2] = z;
}

__global__ void okp_lift_peaks_kernel(okp_camera cam, const int* __restrict__ count, const float* __restrict__ xyc, int n_maps,
                                      int cap, const float* __restrict__ depth, int H, int W, int max_x, int max_y,
                                      double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_maps * cap) return;
  const int map = i / cap, j = i - map * cap;
  double* o = out + (size_t)i * 4;
  if (j >= count[map]) {
    const double nan = __builtin_nan("");
    o[0] = nan; o[1] = nan; o[2] = nan; o[3] = nan;
    return;
  }
  const float* pk = xyc + (size_t)i * 3;
  double xu, yu;
  camera_undistort(cam, (double)pk[0], (double)pk[1], xu, yu);
  const float xf = (float)xu, yf = (float)yu;
  int xi = (int)rintf(xf), yi = (int)rintf(yf);
  xi = min(max(xi, 0), max_x);
  yi = min(max(yi, 0), max_y);
  const double z = (double)depth[((size_t)map * H + yi) * W + xi];
  o[0] = ((double)xf - cam.cx) / cam.fx * z;
  o[1] = ((double)yf - cam.cy) / cam.fy * z;
  o[2] = z;
  o[3] = (double)pk[2];
}

struct GroupParams {
  const int* count; const float* xyc; const float* centers;
  int n, K, cap, H, W, max_obj, max_sel;
  int type_count[8];
  float max_dist;
  int* n_obj; int* sel; int* n_votes; int* assign; double* pred;
  float* reduced;          // [n][max_obj][K-1][max_sel][2] (x, y) or NULL: the k-means reduction of surplus votes (reduce_surplus below)
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);        // butterfly: every lane ends with the same bits
  return v;
}

// Surplus votes of a multi-instance type -> `want` cluster centres (ObjectExtraction, perception/pipeline.py:135-148: the reference runs an
// unseeded sklearn KMeans(init='random', n_init = 10 restarts) and keeps the run of least inertia; its result is reproducible only as a
// set).  Device form: Lloyd's algorithm, one lane per vote (the first 64 votes of the object, in peak order), distances and means in fp64,
// farthest-point initialisation (the start that separates well-separated groups at once) seeded in turn by each of the up to KM_RESTARTS most
// confident votes; the run of least inertia wins, earlier seeds win ties.  Deterministic: same votes, same centres, on every launch.
constexpr int KM_RESTARTS = 4, KM_ITERS = 32, KM_MAXK = 8;
__device__ void reduce_surplus(int np, double px, double py, float conf, int want, int lane, float* out /* [max_sel][2] */) {
  const bool live = lane < np;
  // rank of this vote by (confidence descending, peak order ascending)
  int rank = 0;
  for (int i = 0; i < np; ++i) {
    const float ci = __shfl(conf, i);
    rank += (ci > conf || (ci == conf && i < lane)) ? 1 : 0;
  }
  double best_inertia = 1e300, bx[KM_MAXK], by[KM_MAXK];
#pragma unroll
  for (int t = 0; t < KM_MAXK; ++t) { bx[t] = 0.0; by[t] = 0.0; }
  const int restarts = np < KM_RESTARTS ? np : KM_RESTARTS;
  for (int r = 0; r < restarts; ++r) {
    double cx[KM_MAXK], cy[KM_MAXK];
#pragma unroll
    for (int t = 0; t < KM_MAXK; ++t) { cx[t] = 0.0; cy[t] = 0.0; }
    const int seed = __ffsll((long long)__ballot(live && rank == r)) - 1;
    cx[0] = __shfl(px, seed); cy[0] = __shfl(py, seed);
    double dmin = live ? (px - cx[0]) * (px - cx[0]) + (py - cy[0]) * (py - cy[0]) : -1.0;
#pragma unroll
    for (int t = 1; t < KM_MAXK; ++t) {
      if (t < want) {                                     // the vote farthest from the centres chosen so far (lowest lane wins ties)
        double d = dmin; int l = lane;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          const double d2 = __shfl_xor(d, off); const int l2 = __shfl_xor(l, off);
          if (d2 > d || (d2 == d && l2 < l)) { d = d2; l = l2; }
        }
        cx[t] = __shfl(px, l); cy[t] = __shfl(py, l);
        const double dn = (px - cx[t]) * (px - cx[t]) + (py - cy[t]) * (py - cy[t]);
        if (live && dn < dmin) dmin = dn;
      }
    }
    double inertia = 0.0;
    for (int it = 0; it < KM_ITERS; ++it) {
      int a = 0; double da = 1e300;
#pragma unroll
      for (int t = 0; t < KM_MAXK; ++t) {
        if (t < want) {
          const double d = (px - cx[t]) * (px - cx[t]) + (py - cy[t]) * (py - cy[t]);
          if (d < da) { da = d; a = t; }
        }
      }
      inertia = wave_sum(live ? da : 0.0);
      bool moved = false;
#pragma unroll
      for (int t = 0; t < KM_MAXK; ++t) {
        if (t < want) {
          const bool in = live && a == t;
          const double cnt = wave_sum(in ? 1.0 : 0.0);
          if (cnt > 0.0) {                                // an empty cluster keeps its centre
            const double nx = wave_sum(in ? px : 0.0) / cnt, ny = wave_sum(in ? py : 0.0) / cnt;
            moved = moved || nx != cx[t] || ny != cy[t];
            cx[t] = nx; cy[t] = ny;
          }
        }
      }
      if (!moved) break;                                  // (wave-uniform: every lane holds the same centres)
    }
    if (inertia < best_inertia) {
      best_inertia = inertia;
#pragma unroll
      for (int t = 0; t < KM_MAXK; ++t) { bx[t] = cx[t]; by[t] = cy[t]; }
    }
  }
#pragma unroll
  for (int t = 0; t < KM_MAXK; ++t)
    if (t < want && lane == t) { out[2 * t] = (float)bx[t]; out[2 * t + 1] = (float)by[t]; }
}

// One wave per frame, one lane per peak.  The reference walks the peaks of a keypoint type in order and lets each
// vote for the nearest object centre (pipeline.py:121-133); here the votes of up to 64 peaks are taken at once and
// the order-dependent parts are rebuilt from ballots: a peak's slot among the votes of its object is the number of
// lower lanes voting for the same object, and the "most confident vote, first one wins ties" rule is a wave-wide
// (max confidence, lowest lane) reduction.  Lane o keeps the running vote count / best vote of object o.
// Distances are evaluated in fp64 like the reference (NumPy promotes float32 centres + float64 pixel grid).
__global__ __launch_bounds__(64) void okp_group_objects_kernel(const GroupParams p) {
  __shared__ float oc[64 * 2];
  const int f = blockIdx.x, lane = threadIdx.x;
  const int* cnt = p.count + (size_t)f * p.K;
  const float* pk = p.xyc + (size_t)f * p.K * p.cap * 3;
  const int nobj = min(min(cnt[0], p.cap), p.max_obj);
  if (lane == 0) p.n_obj[f] = nobj;
  int* sel = p.sel + (size_t)f * p.max_obj * (p.K - 1) * p.max_sel;
  int* votes = p.n_votes + (size_t)f * p.max_obj * (p.K - 1);
  for (int i = lane; i < p.max_obj * (p.K - 1) * p.max_sel; i += 64) sel[i] = -1;
  for (int i = lane; i < p.max_obj * (p.K - 1); i += 64) votes[i] = 0;
  int* assign = p.assign + (size_t)f * p.K * p.cap;
  double* pred = p.pred + (size_t)f * p.K * p.cap * 2;
  for (int i = lane; i < p.K * p.cap; i += 64) { assign[i] = -1; pred[2 * i] = 0.0; pred[2 * i + 1] = 0.0; }   // unused slots are defined output (callers hand over uninitialised buffers)
  float* reduced = p.reduced ? p.reduced + (size_t)f * p.max_obj * (p.K - 1) * p.max_sel * 2 : nullptr;
  if (reduced)
    for (int i = lane; i < p.max_obj * (p.K - 1) * p.max_sel * 2; i += 64) reduced[i] = __builtin_nanf("");
  if (nobj == 0) return;
  if (lane < nobj) { oc[2 * lane] = pk[lane * 3 + 0]; oc[2 * lane + 1] = pk[lane * 3 + 1]; }
  __syncthreads();
  const unsigned long long below = (1ull << lane) - 1ull;
  for (int k = 1; k < p.K; ++k) {
    const int want = p.type_count[k - 1];
    const float* cmap = p.centers + ((size_t)f * (p.K - 1) + (k - 1)) * 2 * p.H * p.W;
    const int npk = min(cnt[k], p.cap);
    int my_votes = 0, my_best_j = -1;                    // state of object `lane`
    float my_best_conf = 0.f;
    for (int base = 0; base < npk; base += 64) {
      const int j = base + lane;
      const bool active = j < npk;
      int best = -1;
      float conf = 0.f;
      bool ok = false;
      if (active) {
        const float* q = pk + ((size_t)k * p.cap + j) * 3;
        int xi = (int)rintf(q[0]), yi = (int)rintf(q[1]);
        xi = min(max(xi, 0), p.W - 1);
        yi = min(max(yi, 0), p.H - 1);
        const double cx = ((double)xi + 0.5) + (double)cmap[(size_t)yi * p.W + xi];
        const double cy = ((double)yi + 0.5) + (double)cmap[(size_t)p.H * p.W + (size_t)yi * p.W + xi];
        double bestd = 1e300;
        best = 0;
        for (int o = 0; o < nobj; ++o) {
          const double dx = (double)oc[2 * o] - cx, dy = (double)oc[2 * o + 1] - cy;
          const double d = sqrt(dx * dx + dy * dy);
          if (d < bestd) { bestd = d; best = o; }
        }
        pred[((size_t)k * p.cap + j) * 2 + 0] = cx;
        pred[((size_t)k * p.cap + j) * 2 + 1] = cy;
        ok = !(bestd > (double)p.max_dist);
        if (ok) assign[k * p.cap + j] = best;
        conf = q[2];
      }
      for (int o = 0; o < nobj; ++o) {
        const bool mine = ok && best == o;
        const unsigned long long m = __ballot(mine);
        if (m == 0ull) continue;
        const int v0 = __shfl(my_votes, o);              // votes object o had before this batch of peaks
        if (want == 1) {                                 // most confident vote; the first one wins ties, as argmax does
          float c = mine ? conf : -__builtin_inff();
          int l = mine ? lane : 64;
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) {
            const float c2 = __shfl_xor(c, off);
            const int l2 = __shfl_xor(l, off);
            if (c2 > c || (c2 == c && l2 < l)) { c = c2; l = l2; }
          }
          if (l == 64) l = __ffsll((long long)m) - 1;     // every confidence compared false (NaN): first voter
          if (lane == o && (v0 == 0 || c > my_best_conf)) { my_best_conf = c; my_best_j = base + l; }
        } else if (mine) {
          const int v = v0 + __popcll(m & below);
          if (v < p.max_sel) sel[((size_t)o * (p.K - 1) + (k - 1)) * p.max_sel + v] = j;   // peak order; the caller resolves v > want
        }
        if (lane == o) my_votes += __popcll(m);
      }
    }
    if (lane < nobj) {
      votes[lane * (p.K - 1) + (k - 1)] = my_votes;
      if (want == 1 && my_votes > 0) sel[((size_t)lane * (p.K - 1) + (k - 1)) * p.max_sel] = my_best_j;
    }
    // multi-instance type with more votes than instances: the reference's k-means reduction, on the device (wave-uniform loop over the objects)
    if (reduced && want > 1 && want <= p.max_sel) {
      const unsigned long long surplus = __ballot(lane < nobj && my_votes > want);
      if (surplus) {
        __syncthreads();                                  // the `assign` entries of this type, written by other lanes, are visible
        for (unsigned long long rest = surplus; rest; rest &= rest - 1ull) {
          const int o = __ffsll((long long)rest) - 1;
          int np = 0;
          double px = 0.0, py = 0.0; float cf = 0.f;
          for (int base = 0; base < npk && np < 64; base += 64) {        // the object's votes in peak order, compacted onto lanes 0 .. np - 1
            const int j = base + lane;
            const bool mine = j < npk && assign[k * p.cap + j] == o;
            const unsigned long long m = __ballot(mine);
            float x = 0.f, y = 0.f, c = 0.f;
            if (mine) { const float* q = pk + ((size_t)k * p.cap + j) * 3; x = q[0]; y = q[1]; c = q[2]; }
            for (unsigned long long mm = m; mm; mm &= mm - 1ull) {
              const int src = __ffsll((long long)mm) - 1;
              const int dst = np + __popcll(m & ((1ull << src) - 1ull));
              const float sx = __shfl(x, src), sy = __shfl(y, src), sc = __shfl(c, src);
              if (lane == dst) { px = (double)sx; py = (double)sy; cf = sc; }
            }
            np += __popcll(m);
          }
          np = np < 64 ? np : 64;
          reduce_surplus(np, px, py, cf, want, lane, reduced + ((size_t)o * (p.K - 1) + (k - 1)) * p.max_sel * 2);
        }
      }
    }
  }
}

constexpr int OKP_TRI_OCTET_MAX = 256;

struct TriParams {
  okp_camera left, right;
  double T[12];
  double F[9];
  int correct;
};

// ---- Hartley-Sturm correction (cv2.correctMatches) -------------------------------------------
__device__ void mat3_mul(const double (&A)[3][3], const double (&B)[3][3], double (&C)[3][3]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += A[i][k] * B[k][j]; C[i][j] = s; }
}

// Right null vector of a 3x3 matrix (rank 2): the largest cross product of two rows.
__device__ void null3(const double (&M)[3][3], double (&e)[3]) {
  double best = -1;
  for (int a = 0; a < 3; ++a) {
    const int b = (a + 1) % 3;
    const double c0 = M[a][1] * M[b][2] - M[a][2] * M[b][1];
    const double c1 = M[a][2] * M[b][0] - M[a][0] * M[b][2];
    const double c2 = M[a][0] * M[b][1] - M[a][1] * M[b][0];
    const double n = c0 * c0 + c1 * c1 + c2 * c2;
    if (n > best) { best = n; e[0] = c0; e[1] = c1; e[2] = c2; }
  }
}

__device__ double hs_cost(double t, double a, double b, double c, double d, double f1, double f2) {
  const double u = a * t + b, v = c * t + d;
  return t * t / (1.0 + f1 * f1 * t * t) + v * v / (u * u + f2 * f2 * v * v);
}

// ---- one thread per pair: the form for MANY pairs (every lane busy; rounds 1-3).  okp_triangulate_dlt launches it above
// OKP_TRI_OCTET_MAX pairs, the octet-cooperative wavefront kernel below for fewer (a camera rig's tens of points per tick: latency) ----
// One-sided (Hestenes) Jacobi on the columns of a 4x4 matrix; returns the column of V that
// belongs to the smallest singular value, i.e. argmin |A v| over unit v.
__device__ void null_vector4(double (&A)[4][4], double (&v)[4]) {
  double V[4][4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < 3; ++p)
      for (int q = p + 1; q < 4; ++q) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int i = 0; i < 4; ++i) { alpha += A[i][p] * A[i][p]; beta += A[i][q] * A[i][q]; gamma += A[i][p] * A[i][q]; }
        if (gamma == 0.0) continue;
        off = fmax(off, fabs(gamma) / sqrt(alpha * beta + 1e-300));
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int i = 0; i < 4; ++i) {
          const double ap = A[i][p], aq = A[i][q];
          A[i][p] = c * ap - s * aq; A[i][q] = s * ap + c * aq;
          const double vp = V[i][p], vq = V[i][q];
          V[i][p] = c * vp - s * vq; V[i][q] = s * vp + c * vq;
        }
      }
    if (off < 1e-15) break;
  }
  int best = 0;
  double bestn = 1e300;
  for (int j = 0; j < 4; ++j) {
    double n = 0;
    for (int i = 0; i < 4; ++i) n += A[i][j] * A[i][j];
    if (n < bestn) { bestn = n; best = j; }
  }
  for (int i = 0; i < 4; ++i) v[i] = V[i][best];
}

// Real roots of a degree-6 polynomial g[0] + g[1] t + ... + g[6] t^6 by Durand-Kerner in complex fp64.
__device__ int poly6_roots(const double (&g)[7], double (&re)[6], double (&im)[6]) {
  int deg = 6;
  double scale = 0;
  for (int i = 0; i <= 6; ++i) scale = fmax(scale, fabs(g[i]));
  while (deg > 0 && fabs(g[deg]) <= 1e-14 * scale) --deg;
  if (deg == 0) return 0;
  double a[7];
  for (int i = 0; i <= deg; ++i) a[i] = g[i] / g[deg];
  double radius = 0;
  for (int i = 0; i < deg; ++i) radius = fmax(radius, fabs(a[i]));
  radius = 1.0 + radius;
  radius = fmin(radius, 1e6);
  for (int k = 0; k < deg; ++k) {
    const double ang = 2.0 * 3.141592653589793 * k / deg + 0.4;
    const double r = radius * (0.5 + 0.07 * k);
    re[k] = r * cos(ang); im[k] = r * sin(ang);
  }
  for (int it = 0; it < 500; ++it) {
    double change = 0;
    for (int k = 0; k < deg; ++k) {
      double pr = 1.0, pi = 0.0;       // p(z) by Horner, monic
      for (int i = deg - 1; i >= 0; --i) { const double nr = pr * re[k] - pi * im[k] + a[i]; pi = pr * im[k] + pi * re[k]; pr = nr; }
      double qr = 1.0, qi = 0.0;       // prod (z_k - z_j)
      for (int j = 0; j < deg; ++j) if (j != k) {
        const double dr = re[k] - re[j], di = im[k] - im[j];
        const double nr = qr * dr - qi * di; qi = qr * di + qi * dr; qr = nr;
      }
      const double den = qr * qr + qi * qi + 1e-300;
      const double sr = (pr * qr + pi * qi) / den, si = (pi * qr - pr * qi) / den;
      re[k] -= sr; im[k] -= si;
      change = fmax(change, fabs(sr) + fabs(si));
    }
    if (change < 1e-14 * radius) break;
  }
  return deg;
}

__device__ void hartley_sturm(const double (&F)[9], double& x1, double& y1, double& x2, double& y2) {
  // translate both points to the origin: F' = T2^-T F T1^-1 with T^-1 = [[1,0,x],[0,1,y],[0,0,1]]
  double Fm[3][3], T1i[3][3] = {{1, 0, x1}, {0, 1, y1}, {0, 0, 1}}, T2it[3][3] = {{1, 0, 0}, {0, 1, 0}, {x2, y2, 1}};
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Fm[i][j] = F[3 * i + j];
  double tmp[3][3], Fp[3][3];
  mat3_mul(T2it, Fm, tmp);
  mat3_mul(tmp, T1i, Fp);
  double e1[3], e2[3], Ft[3][3];
  null3(Fp, e1);                                  // F' e1 = 0
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Ft[i][j] = Fp[j][i];
  null3(Ft, e2);                                  // e2^T F' = 0
  const double n1 = sqrt(e1[0] * e1[0] + e1[1] * e1[1]), n2 = sqrt(e2[0] * e2[0] + e2[1] * e2[1]);
  if (n1 < 1e-300 || n2 < 1e-300) return;
  for (int i = 0; i < 3; ++i) { e1[i] /= n1; e2[i] /= n2; }
  double R1[3][3] = {{e1[0], e1[1], 0}, {-e1[1], e1[0], 0}, {0, 0, 1}};
  double R2[3][3] = {{e2[0], e2[1], 0}, {-e2[1], e2[0], 0}, {0, 0, 1}};
  double R1t[3][3];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R1t[i][j] = R1[j][i];
  double Fpp[3][3];
  mat3_mul(R2, Fp, tmp);
  mat3_mul(tmp, R1t, Fpp);
  const double f1 = e1[2], f2 = e2[2];
  const double a = Fpp[1][1], b = Fpp[1][2], c = Fpp[2][1], d = Fpp[2][2];
  // g(t) = t((at+b)^2 + f2^2 (ct+d)^2)^2 - (ad-bc)(1+f1^2 t^2)^2 (at+b)(ct+d)
  double g[7] = {0, 0, 0, 0, 0, 0, 0};
  {
    // q(t) = (at+b)^2 + f2^2 (ct+d)^2 = q2 t^2 + q1 t + q0
    const double q2 = a * a + f2 * f2 * c * c, q1 = 2 * (a * b + f2 * f2 * c * d), q0 = b * b + f2 * f2 * d * d;
    // q^2
    const double s4 = q2 * q2, s3 = 2 * q2 * q1, s2 = 2 * q2 * q0 + q1 * q1, s1 = 2 * q1 * q0, s0 = q0 * q0;
    g[5] += s4; g[4] += s3; g[3] += s2; g[2] += s1; g[1] += s0;       // t * q^2
    const double k = a * d - b * c;
    // r(t) = (1 + f1^2 t^2)^2 = 1 + 2 f1^2 t^2 + f1^4 t^4 ; w(t) = (at+b)(ct+d) = ac t^2 + (ad+bc) t + bd
    const double r0 = 1, r2 = 2 * f1 * f1, r4 = f1 * f1 * f1 * f1;
    const double w2 = a * c, w1 = a * d + b * c, w0 = b * d;
    g[0] -= k * (r0 * w0);
    g[1] -= k * (r0 * w1);
    g[2] -= k * (r0 * w2 + r2 * w0);
    g[3] -= k * (r2 * w1);
    g[4] -= k * (r2 * w2 + r4 * w0);
    g[5] -= k * (r4 * w1);
    g[6] -= k * (r4 * w2);
  }
  double re[6], im[6];
  const int nroots = poly6_roots(g, re, im);
  // t = infinity: s = 1/f1^2 + c^2/(a^2 + f2^2 c^2)
  double best_t = 0, best_s = 1e300;
  bool inf_best = false;
  {
    const double den = a * a + f2 * f2 * c * c;
    if (f1 != 0.0 && den != 0.0) { best_s = 1.0 / (f1 * f1) + c * c / den; inf_best = true; }
  }
  for (int k = 0; k < nroots; ++k) {
    const double t = re[k];                       // OpenCV evaluates the cost at the real part of every root
    const double s = hs_cost(t, a, b, c, d, f1, f2);
    if (s < best_s) { best_s = s; best_t = t; inf_best = false; }
  }
  // closest points on l1 = (t f1, 1, -t), l2 = (-f2 (ct+d), at+b, ct+d) to the origin
  double p1[3], p2[3];
  if (inf_best) {
    p1[0] = f1; p1[1] = 0; p1[2] = f1 * f1;       // limit t -> inf of (t^2 f1, t, t^2 f1^2 + 1) / t^2
    p2[0] = f2 * c * c; p2[1] = -a * c; p2[2] = f2 * f2 * c * c + a * a;
  } else {
    const double t = best_t, u = a * t + b, v = c * t + d;
    p1[0] = t * t * f1; p1[1] = t; p1[2] = t * t * f1 * f1 + 1.0;
    p2[0] = f2 * v * v; p2[1] = -u * v; p2[2] = f2 * f2 * v * v + u * u;
  }
  // back: x = T^-1 R^T p
  double q1v[3], q2v[3];
  for (int i = 0; i < 3; ++i) {
    q1v[i] = R1[0][i] * p1[0] + R1[1][i] * p1[1] + R1[2][i] * p1[2];
    q2v[i] = R2[0][i] * p2[0] + R2[1][i] * p2[1] + R2[2][i] * p2[2];
  }
  if (fabs(q1v[2]) < 1e-300 || fabs(q2v[2]) < 1e-300) return;
  x1 = q1v[0] / q1v[2] + x1; y1 = q1v[1] / q1v[2] + y1;
  x2 = q2v[0] / q2v[2] + x2; y2 = q2v[1] / q2v[2] + y2;
}

__global__ void okp_triangulate_thread_kernel(TriParams tp, const float* __restrict__ lxy, const float* __restrict__ rxy, int m,
                                       double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  double xl, yl, xr, yr;
  camera_undistort(tp.left, (double)lxy[2 * i], (double)lxy[2 * i + 1], xl, yl);
  camera_undistort(tp.right, (double)rxy[2 * i], (double)rxy[2 * i + 1], xr, yr);
  // cv2 hands float32 back for float32 input (camera_utils.py:93-97)
  xl = (double)(float)xl; yl = (double)(float)yl; xr = (double)(float)xr; yr = (double)(float)yr;
  if (tp.correct) {
    double F[9];
    for (int k = 0; k < 9; ++k) F[k] = tp.F[k];
    hartley_sturm(F, xl, yl, xr, yr);
    xl = (double)(float)xl; yl = (double)(float)yl; xr = (double)(float)xr; yr = (double)(float)yr;
  }
  // P1 = K_l [I 0]; P2 = K_r T_RL[:3]
  double P1[3][4] = {{tp.left.fx, 0, tp.left.cx, 0}, {0, tp.left.fy, tp.left.cy, 0}, {0, 0, 1, 0}};
  double P2[3][4];
  for (int j = 0; j < 4; ++j) {
    P2[0][j] = tp.right.fx * tp.T[j] + tp.right.cx * tp.T[8 + j];
    P2[1][j] = tp.right.fy * tp.T[4 + j] + tp.right.cy * tp.T[8 + j];
    P2[2][j] = tp.T[8 + j];
  }
  double A[4][4];
  for (int j = 0; j < 4; ++j) {
    A[0][j] = xl * P1[2][j] - P1[0][j];
    A[1][j] = yl * P1[2][j] - P1[1][j];
    A[2][j] = xr * P2[2][j] - P2[0][j];
    A[3][j] = yr * P2[2][j] - P2[1][j];
  }
  double v[4];
  null_vector4(A, v);
  out[3 * i + 0] = v[0] / v[3];
  out[3 * i + 1] = v[1] / v[3];
  out[3 * i + 2] = v[2] / v[3];
}


// ------------------------------------------------------------------------------------------------------------------------
// Triangulation as a WAVEFRONT kernel: eight lanes (an octet) cooperate on one keypoint pair, eight pairs per wave.  The three
// serial pieces of the per-pair work are spread over the octet and reduced with width-8 shuffles:
//   * undistortion: lane 0 the left point, lane 1 the right point;
//   * the degree-6 polynomial of Hartley-Sturm: lane k owns root k of a Durand-Kerner iteration (all roots stepped together from
//     the previous iterate; the other roots arrive by shuffle), the cost of the seven candidates (six roots + t = infinity) is
//     evaluated one per lane and the minimum taken by an octet reduction with OpenCV's order of preference (infinity, then root order);
//   * the 4x4 null vector of the DLT system: one-sided Jacobi with the two disjoint column rotations of a round in parallel -
//     lanes (rotation r, row i) - dot products by group reductions, rotated columns exchanged between the groups.
// What is uniform per pair (3x3 algebra, polynomial coefficients, back-transformation) is computed redundantly by the eight lanes.
// One thread per pair (rounds 1-3) walked six roots x up to 500 iterations and six rotations x up to 30 sweeps in sequence.
// ------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double oct_bcast(double v, int src) { return __shfl(v, src, 8); }
__device__ __forceinline__ double oct_max(double v) {
#pragma unroll
  for (int off = 4; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 8));
  return v;
}
__device__ __forceinline__ double quad_sum(double v) {       // over the four lanes of a rotation group (lane bits 0-1)
  v += __shfl_xor(v, 1, 8);
  v += __shfl_xor(v, 2, 8);
  return v;
}

// Real parts of the roots of g[0] + ... + g[6] t^6: lane g (< degree) of the octet returns its root in `re`; returns the degree.
__device__ int poly6_roots_oct(const double (&gc)[7], int g, double& re) {
  int deg = 6;
  double scale = 0;
  for (int i = 0; i <= 6; ++i) scale = fmax(scale, fabs(gc[i]));
  while (deg > 0 && fabs(gc[deg]) <= 1e-14 * scale) --deg;            // (uniform over the octet)
  re = 0.0;
  if (__all(deg == 0)) return deg;
  double a[7];
  const double lead = deg > 0 ? gc[deg] : 1.0;
  for (int i = 0; i <= 6; ++i) a[i] = i <= deg ? gc[i] / lead : 0.0;
  double radius = 0;
  for (int i = 0; i < deg; ++i) radius = fmax(radius, fabs(a[i]));
  radius = fmin(1.0 + radius, 1e6);
  const bool mine = g < deg;
  double im = 0.0;
  if (mine) {
    const double ang = 2.0 * 3.141592653589793 * g / deg + 0.4;
    const double r = radius * (0.5 + 0.07 * g);
    re = r * cos(ang); im = r * sin(ang);
  }
  bool done = deg == 0;
  for (int it = 0; it < 500; ++it) {
    double pr = 1.0, pi = 0.0;                                         // p(z) by Horner, monic, at this lane's root
    for (int i = 5; i >= 0; --i)
      if (i < deg) { const double nr = pr * re - pi * im + a[i]; pi = pr * im + pi * re; pr = nr; }
    double qr = 1.0, qi = 0.0;                                         // prod over the other roots (z - z_j)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const double rj = oct_bcast(re, j), ij = oct_bcast(im, j);
      if (j < deg && j != g) {
        const double dr = re - rj, di = im - ij;
        const double nr = qr * dr - qi * di; qi = qr * di + qi * dr; qr = nr;
      }
    }
    const double den = qr * qr + qi * qi + 1e-300;
    const double sr = (pr * qr + pi * qi) / den, si = (pi * qr - pr * qi) / den;
    if (mine && !done) { re -= sr; im -= si; }
    const double change = oct_max(mine ? fabs(sr) + fabs(si) : 0.0);
    done = done || change < 1e-14 * radius;
    if (__all(done)) break;                                            // every octet of the wave has converged
  }
  return deg;
}

// cv2.correctMatches for one pair, by the octet (g = lane within it)
__device__ void hartley_sturm_oct(const double (&F)[9], int g, double& x1, double& y1, double& x2, double& y2) {
  // translate both points to the origin: F' = T2^-T F T1^-1 with T^-1 = [[1,0,x],[0,1,y],[0,0,1]]
  double Fm[3][3], T1i[3][3] = {{1, 0, x1}, {0, 1, y1}, {0, 0, 1}}, T2it[3][3] = {{1, 0, 0}, {0, 1, 0}, {x2, y2, 1}};
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Fm[i][j] = F[3 * i + j];
  double tmp[3][3], Fp[3][3];
  mat3_mul(T2it, Fm, tmp);
  mat3_mul(tmp, T1i, Fp);
  double e1[3], e2[3], Ft[3][3];
  null3(Fp, e1);                                  // F' e1 = 0
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Ft[i][j] = Fp[j][i];
  null3(Ft, e2);                                  // e2^T F' = 0
  const double n1 = sqrt(e1[0] * e1[0] + e1[1] * e1[1]), n2 = sqrt(e2[0] * e2[0] + e2[1] * e2[1]);
  const bool degenerate = n1 < 1e-300 || n2 < 1e-300;           // (the shuffles below need every lane: no early return)
  const double s1 = degenerate ? 1.0 : n1, s2 = degenerate ? 1.0 : n2;
  for (int i = 0; i < 3; ++i) { e1[i] /= s1; e2[i] /= s2; }
  double R1[3][3] = {{e1[0], e1[1], 0}, {-e1[1], e1[0], 0}, {0, 0, 1}};
  double R2[3][3] = {{e2[0], e2[1], 0}, {-e2[1], e2[0], 0}, {0, 0, 1}};
  double R1t[3][3];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R1t[i][j] = R1[j][i];
  double Fpp[3][3];
  mat3_mul(R2, Fp, tmp);
  mat3_mul(tmp, R1t, Fpp);
  const double f1 = e1[2], f2 = e2[2];
  const double a = Fpp[1][1], b = Fpp[1][2], c = Fpp[2][1], d = Fpp[2][2];
  // g(t) = t((at+b)^2 + f2^2 (ct+d)^2)^2 - (ad-bc)(1+f1^2 t^2)^2 (at+b)(ct+d)
  double gp[7] = {0, 0, 0, 0, 0, 0, 0};
  {
    const double q2 = a * a + f2 * f2 * c * c, q1 = 2 * (a * b + f2 * f2 * c * d), q0 = b * b + f2 * f2 * d * d;
    const double s4 = q2 * q2, s3 = 2 * q2 * q1, sq2 = 2 * q2 * q0 + q1 * q1, sq1 = 2 * q1 * q0, s0 = q0 * q0;
    gp[5] += s4; gp[4] += s3; gp[3] += sq2; gp[2] += sq1; gp[1] += s0;       // t * q^2
    const double k = a * d - b * c;
    const double r0 = 1, r2 = 2 * f1 * f1, r4 = f1 * f1 * f1 * f1;
    const double w2 = a * c, w1 = a * d + b * c, w0 = b * d;
    gp[0] -= k * (r0 * w0);
    gp[1] -= k * (r0 * w1);
    gp[2] -= k * (r0 * w2 + r2 * w0);
    gp[3] -= k * (r2 * w1);
    gp[4] -= k * (r2 * w2 + r4 * w0);
    gp[5] -= k * (r4 * w1);
    gp[6] -= k * (r4 * w2);
  }
  double root;
  const int nroots = poly6_roots_oct(gp, g, root);
  // one candidate per lane: lanes 0..5 the real parts of the roots (OpenCV evaluates the cost there), lane 6 t = infinity:
  // s = 1/f1^2 + c^2/(a^2 + f2^2 c^2).  Preference among equal costs as in the sequential scan: infinity first, then root order.
  double cs = 1e300, ct = 0.0;
  int key = 99;
  if (g < nroots) { cs = hs_cost(root, a, b, c, d, f1, f2); ct = root; key = g + 1; }
  else if (g == 6) {
    const double den = a * a + f2 * f2 * c * c;
    if (f1 != 0.0 && den != 0.0) { cs = 1.0 / (f1 * f1) + c * c / den; key = 0; }
  }
  if (!(cs < 1e300)) { cs = 1e300; ct = 0.0; key = 99; }         // (NaN / overflow: never preferred; all of them: t = 0 as the sequential default)
#pragma unroll
  for (int off = 4; off > 0; off >>= 1) {
    const double s2o = __shfl_xor(cs, off, 8), t2o = __shfl_xor(ct, off, 8);
    const int k2o = __shfl_xor(key, off, 8);
    if (s2o < cs || (s2o == cs && k2o < key)) { cs = s2o; ct = t2o; key = k2o; }
  }
  const bool inf_best = key == 0;
  const double best_t = key == 99 ? 0.0 : ct;
  // closest points on l1 = (t f1, 1, -t), l2 = (-f2 (ct+d), at+b, ct+d) to the origin
  double p1[3], p2[3];
  if (inf_best) {
    p1[0] = f1; p1[1] = 0; p1[2] = f1 * f1;       // limit t -> inf of (t^2 f1, t, t^2 f1^2 + 1) / t^2
    p2[0] = f2 * c * c; p2[1] = -a * c; p2[2] = f2 * f2 * c * c + a * a;
  } else {
    const double t = best_t, u = a * t + b, v = c * t + d;
    p1[0] = t * t * f1; p1[1] = t; p1[2] = t * t * f1 * f1 + 1.0;
    p2[0] = f2 * v * v; p2[1] = -u * v; p2[2] = f2 * f2 * v * v + u * u;
  }
  // back: x = T^-1 R^T p
  double q1v[3], q2v[3];
  for (int i = 0; i < 3; ++i) {
    q1v[i] = R1[0][i] * p1[0] + R1[1][i] * p1[1] + R1[2][i] * p1[2];
    q2v[i] = R2[0][i] * p2[0] + R2[1][i] * p2[1] + R2[2][i] * p2[2];
  }
  if (degenerate || fabs(q1v[2]) < 1e-300 || fabs(q2v[2]) < 1e-300) return;
  x1 = q1v[0] / q1v[2] + x1; y1 = q1v[1] / q1v[2] + y1;
  x2 = q2v[0] / q2v[2] + x2; y2 = q2v[1] / q2v[2] + y2;
}

// Null vector of a 4x4 matrix (argmin |A v| over unit v) by one-sided (Hestenes) Jacobi, by the octet: lane (r = g >> 2, i = g & 3)
// holds row i of A and of V (a0..a3, v0..v3; both groups hold every row) and rotates the r-th column pair of the round.
__device__ void null_vector4_oct(int g, double a0, double a1, double a2, double a3, double (&nv)[4]) {
  const int r = g >> 2, i = g & 3;
  double v0 = i == 0, v1 = i == 1, v2 = i == 2, v3 = i == 3;
  auto get = [&](int col, double c0, double c1, double c2, double c3) { return col == 0 ? c0 : col == 1 ? c1 : col == 2 ? c2 : c3; };
  bool done = false;
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0.0;
#pragma unroll
    for (int round = 0; round < 3; ++round) {
      // round 0: (0,1) | (2,3);  round 1: (0,2) | (1,3);  round 2: (0,3) | (1,2)
      const int p = r == 0 ? 0 : (round == 0 ? 2 : 1);
      const int q = r == 0 ? round + 1 : (round == 2 ? 2 : 3);
      const double ap = get(p, a0, a1, a2, a3), aq = get(q, a0, a1, a2, a3);
      const double vp = get(p, v0, v1, v2, v3), vq = get(q, v0, v1, v2, v3);
      const double alpha = quad_sum(ap * ap), beta = quad_sum(aq * aq), gamma = quad_sum(ap * aq);
      double nap = ap, naq = aq, nvp = vp, nvq = vq;
      if (gamma != 0.0 && !done) {
        off = fmax(off, fabs(gamma) / sqrt(alpha * beta + 1e-300));
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        nap = c * ap - s * aq; naq = s * ap + c * aq;
        nvp = c * vp - s * vq; nvq = s * vp + c * vq;
      }
      // every lane takes its own group's two columns from itself and the other two from its partner in the other group (lane ^ 4)
      const double oap = __shfl_xor(nap, 4, 8), oaq = __shfl_xor(naq, 4, 8), ovp = __shfl_xor(nvp, 4, 8), ovq = __shfl_xor(nvq, 4, 8);
      const int op = r == 0 ? (round == 0 ? 2 : 1) : 0;                  // the partner's pair
      const int oq = r == 0 ? (round == 2 ? 2 : 3) : round + 1;
      auto put = [&](int col, double x, double& c0, double& c1, double& c2, double& c3) {
        if (col == 0) c0 = x; else if (col == 1) c1 = x; else if (col == 2) c2 = x; else c3 = x;
      };
      put(p, nap, a0, a1, a2, a3); put(q, naq, a0, a1, a2, a3); put(op, oap, a0, a1, a2, a3); put(oq, oaq, a0, a1, a2, a3);
      put(p, nvp, v0, v1, v2, v3); put(q, nvq, v0, v1, v2, v3); put(op, ovp, v0, v1, v2, v3); put(oq, ovq, v0, v1, v2, v3);
    }
    done = done || oct_max(off) < 1e-15;
    if (__all(done)) break;
  }
  // the column of the smallest norm (the first one among equals)
  const double n0 = quad_sum(a0 * a0), n1 = quad_sum(a1 * a1), n2 = quad_sum(a2 * a2), n3 = quad_sum(a3 * a3);
  int best = 0;
  double bestn = n0;
  if (n1 < bestn) { bestn = n1; best = 1; }
  if (n2 < bestn) { bestn = n2; best = 2; }
  if (n3 < bestn) { bestn = n3; best = 3; }
  const double mine = get(best, v0, v1, v2, v3);        // V[i][best] of this lane's row
#pragma unroll
  for (int k = 0; k < 4; ++k) nv[k] = oct_bcast(mine, k);
}

__global__ __launch_bounds__(64) void okp_triangulate_kernel(TriParams tp, const float* __restrict__ lxy, const float* __restrict__ rxy, int m,
                                                           double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int g = lane & 7;
  const int pair = blockIdx.x * 8 + (lane >> 3);
  const bool live = pair < m;
  const int i = live ? pair : 0;                  // idle octets of the last wave work on pair 0 (the wave-wide votes need every lane) and store nothing
  double ux = 0.0, uy = 0.0;
  if (g == 0) camera_undistort(tp.left, (double)lxy[2 * i], (double)lxy[2 * i + 1], ux, uy);
  else if (g == 1) camera_undistort(tp.right, (double)rxy[2 * i], (double)rxy[2 * i + 1], ux, uy);
  // cv2 hands float32 back for float32 input (camera_utils.py:93-97)
  double xl = (double)(float)oct_bcast(ux, 0), yl = (double)(float)oct_bcast(uy, 0);
  double xr = (double)(float)oct_bcast(ux, 1), yr = (double)(float)oct_bcast(uy, 1);
  if (tp.correct) {
    double F[9];
    for (int k = 0; k < 9; ++k) F[k] = tp.F[k];
    hartley_sturm_oct(F, g, xl, yl, xr, yr);
    xl = (double)(float)xl; yl = (double)(float)yl; xr = (double)(float)xr; yr = (double)(float)yr;
  }
  // P1 = K_l [I 0]; P2 = K_r T_RL[:3]; row i of the DLT system for this lane
  const int row = g & 3;
  double a[4];
  for (int j = 0; j < 4; ++j) {
    const double P1_0 = j == 0 ? tp.left.fx : j == 2 ? tp.left.cx : 0.0, P1_1 = j == 1 ? tp.left.fy : j == 2 ? tp.left.cy : 0.0, P1_2 = j == 2 ? 1.0 : 0.0;
    const double P2_0 = tp.right.fx * tp.T[j] + tp.right.cx * tp.T[8 + j];
    const double P2_1 = tp.right.fy * tp.T[4 + j] + tp.right.cy * tp.T[8 + j];
    const double P2_2 = tp.T[8 + j];
    a[j] = row == 0 ? xl * P1_2 - P1_0 : row == 1 ? yl * P1_2 - P1_1 : row == 2 ? xr * P2_2 - P2_0 : yr * P2_2 - P2_1;
  }
  double v[4];
  null_vector4_oct(g, a[0], a[1], a[2], a[3], v);
  if (live && g == 0) {
    out[3 * pair + 0] = v[0] / v[3];
    out[3 * pair + 1] = v[1] / v[3];
    out[3 * pair + 2] = v[2] / v[3];
  }
}

}  // namespace

extern "C" int okp_fisheye_undistort(const okp_camera* cam, const float* xy, int32_t m, double* out, void* stream) {
  return okp_camera_undistort(cam, xy, m, out, stream);
}

extern "C" int okp_camera_undistort(const okp_camera* cam, const float* xy, int32_t m, double* out, void* stream) {
  if (!cam || (m > 0 && (!xy || !out))) { okp_set_error("okp_camera_undistort: null argument"); return OKP_EINVAL; }
  if (cam->model != OKP_CAM_EQUIDISTANT && cam->model != OKP_CAM_RADTAN) { okp_set_error("okp_camera_undistort: unknown camera model %d", cam->model); return OKP_EINVAL; }
  if (m <= 0) return OKP_OK;
  hipLaunchKernelGGL(okp_undistort_kernel, dim3((m + 63) / 64), dim3(64), 0, (hipStream_t)stream, *cam, xy, m, out);
  return okp_check_hip(hipGetLastError(), "okp_camera_undistort launch");
}

extern "C" int okp_unproject_depth(const okp_camera* cam, const float* xy, const int32_t* map_id, int32_t m,
                                   const float* depth, int32_t h, int32_t w, int32_t max_x, int32_t max_y,
                                   double* out, void* stream) {
  if (!cam || (m > 0 && (!xy || !map_id || !depth || !out))) { okp_set_error("okp_unproject_depth: null argument"); return OKP_EINVAL; }
  if (m <= 0) return OKP_OK;
  if (max_x < 0 || max_x >= w || max_y < 0 || max_y >= h) {
    okp_set_error("okp_unproject_depth: clip box (%d,%d) outside the %dx%d depth map", max_x, max_y, h, w);
    return OKP_EINVAL;
  }
  hipLaunchKernelGGL(okp_unproject_kernel, dim3((m + 63) / 64), dim3(64), 0, (hipStream_t)stream, *cam, xy, map_id, m, depth, h, w, max_x, max_y, out);
  return okp_check_hip(hipGetLastError(), "okp_unproject_depth launch");
}

extern "C" int okp_lift_peaks(const okp_camera* cam, const int32_t* count, const float* xyc, int32_t n_maps, int32_t cap,
                              const float* depth, int32_t h, int32_t w, int32_t max_x, int32_t max_y, double* out, void* stream) {
  if (!cam || !count || !xyc || !depth || !out) { okp_set_error("okp_lift_peaks: null argument"); return OKP_EINVAL; }
  if (n_maps <= 0 || cap <= 0) return OKP_OK;
  if (max_x < 0 || max_x >= w || max_y < 0 || max_y >= h) {
    okp_set_error("okp_lift_peaks: clip box (%d,%d) outside the %dx%d depth map", max_x, max_y, h, w);
    return OKP_EINVAL;
  }
  const int total = n_maps * cap;
  hipLaunchKernelGGL(okp_lift_peaks_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, *cam, count, xyc, n_maps, cap, depth, h, w, max_x, max_y, out);
  return okp_check_hip(hipGetLastError(), "okp_lift_peaks launch");
}

__global__ void okp_capacity_overflow_kernel(const int* __restrict__ count, int n_maps, int K, int cap, int max_obj, const int* __restrict__ range_flag, int* __restrict__ flag) {
  // one workgroup: any map with more peaks than `cap`, or any centre map (map 0 of a frame) with more than `max_obj`
  __shared__ int any;
  if (threadIdx.x == 0) any = 0;
  __syncthreads();
  int bad = 0;
  for (int i = threadIdx.x; i < n_maps; i += blockDim.x) {
    const int c = count[i];
    bad |= (c > cap) || ((i % K) == 0 && c > max_obj);
  }
  if (bad) atomicOr(&any, 1);
  __syncthreads();
  if (threadIdx.x == 0) *flag = any | ((range_flag && *range_flag) ? 2 : 0);      // bit 1: the network's fp16-range flag (okp_conv_set_range_flag)
}

extern "C" int okp_capacity_overflow(const int32_t* count, int32_t n_maps, int32_t K, int32_t cap, int32_t max_obj, const int32_t* range_flag, int32_t* flag, void* stream) {
  if (!count || !flag || K < 1) { okp_set_error("okp_capacity_overflow: bad argument"); return OKP_EINVAL; }
  hipLaunchKernelGGL(okp_capacity_overflow_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, count, n_maps, K, cap, max_obj, range_flag, flag);
  return okp_check_hip(hipGetLastError(), "okp_capacity_overflow launch");
}

extern "C" int okp_group_objects(const int32_t* count, const float* xyc, const float* centers, int32_t n, int32_t K, int32_t cap,
                                 int32_t h, int32_t w, const int32_t* type_count, float max_dist, int32_t max_obj, int32_t max_sel,
                                 int32_t* n_obj, int32_t* sel, int32_t* n_votes, int32_t* assign, double* pred, float* reduced, void* stream) {
  if (!count || !xyc || !centers || !type_count || !n_obj || !sel || !n_votes || !assign || !pred) { okp_set_error("okp_group_objects: null argument"); return OKP_EINVAL; }
  if (K < 2 || K > 8 || max_sel < 1 || max_sel > 8 || max_obj < 1 || max_obj > 64 || cap < 1) { okp_set_error("okp_group_objects: K in [2,8], max_sel in [1,8], max_obj in [1,64] required"); return OKP_EINVAL; }
  if (n <= 0) return OKP_OK;
  GroupParams p;
  p.count = count; p.xyc = xyc; p.centers = centers; p.n = n; p.K = K; p.cap = cap; p.H = h; p.W = w;
  p.max_obj = max_obj; p.max_sel = max_sel; p.max_dist = max_dist; p.n_obj = n_obj; p.sel = sel; p.n_votes = n_votes; p.assign = assign; p.pred = pred; p.reduced = reduced;
  for (int k = 0; k < 8; ++k) p.type_count[k] = k < K - 1 ? type_count[k] : 0;
  hipLaunchKernelGGL(okp_group_objects_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, p);
  return okp_check_hip(hipGetLastError(), "okp_group_objects launch");
}

extern "C" int okp_triangulate_dlt(const okp_camera* left, const okp_camera* right, const double* T_RL, const double* F,
                                   int correct_matches, const float* lxy, const float* rxy, int32_t m, double* out,
                                   void* stream) {
  if (!left || !right || !T_RL || (correct_matches && !F) || (m > 0 && (!lxy || !rxy || !out))) {
    okp_set_error("okp_triangulate_dlt: null argument"); return OKP_EINVAL;
  }
  if (m <= 0) return OKP_OK;
  TriParams tp;
  tp.left = *left; tp.right = *right;
  for (int i = 0; i < 12; ++i) tp.T[i] = T_RL[i];
  for (int i = 0; i < 9; ++i) tp.F[i] = F ? F[i] : 0.0;
  tp.correct = correct_matches;
  // measured (scripts/probe_triangulate.py, launch + host): 4 / 20 pairs 23-24 us (octets) against 50 us (threads); 256 pairs 50 / 48 us;
  // 4096 pairs 64 / 48 us - an octet keeps six of eight lanes busy in the root iteration, a thread all of one
  if (m <= OKP_TRI_OCTET_MAX) hipLaunchKernelGGL(okp_triangulate_kernel, dim3((m + 7) / 8), dim3(64), 0, (hipStream_t)stream, tp, lxy, rxy, m, out);     // eight pairs per wave
  else hipLaunchKernelGGL(okp_triangulate_thread_kernel, dim3((m + 63) / 64), dim3(64), 0, (hipStream_t)stream, tp, lxy, rxy, m, out);
  return okp_check_hip(hipGetLastError(), "okp_triangulate_dlt launch");
}
