// Host-side tile bookkeeping of the zonal loop, bit-exact with the reference's float64 arithmetic.
//
//   ffa_slice_grid   <- flair_zonal_detection/slicing.py:51-112 (generate_patches_from_reference core loop)
//   ffa_write_window <- flair_zonal_detection/inference.py:318-335 (window placement and clipping)
//
// What has to be reproduced to get the same bits (verified against the
// reference through tests/golden/gen_goldens.py):
//   * np.arange(start, stop, step): n = ceil((stop - start) / step), element i = start + i * delta with
//     delta = (start + step) - start  (NOT start + i * step)
//   * duplicate filter on round(v, 6): numpy scalars round as rint(v * 1e6) / 1e6, Python floats round
//     correctly (decimal) -- which one applies depends on whether the coordinate came from np.arange
//     (np.float64) or from the clamp expression (Python float), and min(a, b) keeps a unless b < a
//   * `//` is the fmod-based floor division of CPython / numpy, not floor(a / b)
//   * int(round(q)) is round-half-to-even on the float64 quotient
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <set>
#include <tuple>
#include <vector>

#include "ffa_common_host.h"

namespace {

struct Num {
  double v;
  bool np;  // true: numpy float64 scalar, false: Python float
};

inline Num add(Num a, Num b) { return {a.v + b.v, a.np || b.np}; }
inline Num sub(Num a, Num b) { return {a.v - b.v, a.np || b.np}; }
inline Num pymin(Num a, Num b) { return (b.v < a.v) ? b : a; }  // Python min(a, b)

double round6_numpy(double v) { return nearbyint(v * 1e6) / 1e6; }

double round6_python(double v) {
  // correctly rounded decimal, as float.__round__(6) (dtoa mode 3 + strtod); glibc printf is exact
  if (!isfinite(v)) return v;
  char buf[512];
  snprintf(buf, sizeof(buf), "%.6f", v);
  return strtod(buf, nullptr);
}

double round6(Num x) { return x.np ? round6_numpy(x.v) : round6_python(x.v); }

double py_floor_div(double vx, double wx) {
  double mod = fmod(vx, wx);
  double div = (vx - mod) / wx;
  if (mod != 0.0) {
    if ((wx < 0) != (mod < 0)) div -= 1.0;
  }
  double floordiv;
  if (div != 0.0) {
    floordiv = floor(div);
    if (div - floordiv > 0.5) floordiv += 1.0;
  } else {
    floordiv = copysign(0.0, vx / wx);
  }
  return floordiv;
}

long long arange_len(double start, double stop, double step) {
  const double n = ceil((stop - start) / step);
  if (!(n > 0)) return 0;
  return (long long)n;
}

// numpy fills element 0 and 1 explicitly (start, start + step) and the rest as start + i * delta
inline double arange_at(double start, double step, double delta, long long i) {
  if (i == 0) return start;
  if (i == 1) return start + step;
  return start + (double)i * delta;
}

}  // namespace

extern "C" long long ffa_slice_grid(double min_x, double min_y, double max_x, double max_y, double ref_left,
                                    double ref_bottom, int patch_size, int margin, double resolution, ffa_tile_t* out,
                                    long long capacity) {
  if (patch_size <= 0 || margin < 0 || !(resolution > 0) || patch_size - 2 * margin <= 0) {
    ffa_set_error("slice_grid: bad patch/margin/resolution");
    return FFA_ERR_ARG;
  }
  const double size = patch_size * resolution;                 // geo_output_size
  const double gm = margin * resolution;                       // geo_margin
  const double step = (patch_size - 2 * margin) * resolution;  // geo_step
  const double x_start = min_x - gm, x_stop = max_x + gm;
  const double y_start = min_y - gm, y_stop = max_y + gm;
  const long long nx = arange_len(x_start, x_stop, step);
  const long long ny = arange_len(y_start, y_stop, step);
  const double dx = (x_start + step) - x_start;
  const double dy = (y_start + step) - y_start;

  std::set<std::tuple<double, double, double, double>> seen;
  long long count = 0;
  for (long long ix = 0; ix < nx; ++ix) {
    Num xc{arange_at(x_start, step, dx, ix), true};
    for (long long iy = 0; iy < ny; ++iy) {
      Num yc{arange_at(y_start, step, dy, iy), true};
      // clamp inside the zone (the clamped x sticks for the rest of this column, slicing.py:73-74)
      if (xc.v + size > max_x + gm) xc = Num{max_x + gm - size, false};
      if (yc.v + size > max_y + gm) yc = Num{max_y + gm - size, false};
      const Num py_gm{gm, false}, py_size{size, false};
      const Num left = add(xc, py_gm);
      const Num right = pymin(sub(add(xc, py_size), py_gm), Num{max_x, false});
      const Num bottom = add(yc, py_gm);
      const Num top = pymin(sub(add(yc, py_size), py_gm), Num{max_y, false});
      auto key = std::make_tuple(round6(left), round6(bottom), round6(right), round6(top));
      if (!seen.insert(key).second) continue;
      const long long col = (long long)py_floor_div(xc.v - ref_left, resolution) + 1;
      const long long row = (long long)py_floor_div(yc.v - ref_bottom, resolution) + 1;
      if (right.v - left.v > 0 && top.v - bottom.v > 0) {
        if (count < capacity && out) {
          ffa_tile_t& t = out[count];
          t.left = left.v; t.bottom = bottom.v; t.right = right.v; t.top = top.v;
          t.x0 = xc.v; t.y0 = yc.v; t.x1 = xc.v + size; t.y1 = yc.v + size;
          t.row = row; t.col = col;
        }
        ++count;
      }
    }
  }
  return count;
}

extern "C" int ffa_write_window(double left, double top, double img_left, double img_bottom, double img_right,
                                double img_top, double out_res, int pred_h, int pred_w, ffa_window_t* out) {
  if (!out || !(out_res > 0)) {
    ffa_set_error("write_window: bad arguments");
    return FFA_ERR_ARG;
  }
  const long long left_px = (long long)nearbyint((left - img_left) / out_res);
  const long long top_px = (long long)nearbyint((img_top - top) / out_res);
  long long h = pred_h, w = pred_w;
  const long long img_h = (long long)nearbyint((img_top - img_bottom) / out_res);
  const long long img_w = (long long)nearbyint((img_right - img_left) / out_res);
  if (top_px + h > img_h) h = img_h - top_px;
  if (left_px + w > img_w) w = img_w - left_px;
  out->col_off = (int)left_px;
  out->row_off = (int)top_px;
  out->width = (int)w;
  out->height = (int)h;
  out->skip = (h <= 0 || w <= 0) ? 1 : 0;
  return FFA_OK;
}
