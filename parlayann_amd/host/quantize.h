// quantize.h -- host mirror of the translating PointRange constructor (algorithms/utils/point_range.h:54-72) for
// the two scalar quantisers the drivers use with `-quantize_bits 8` (bench/neighborsTime.C:157-164, 190-197):
//   Euclidian_Point<uint8_t>  generate_parameters euclidian_point.h:211-235, translate_point :182-209
//   Quantized_Mips_Point<8>   generate_parameters mips_point.h:433-486,      translate_point :416-430
// Host-side preprocessing (one pass over the float slab); the quantised range is an ordinary
// PointRange<Euclidian_Point<uint8_t>> / PointRange<Mips_Point<int8_t>> that is mirrored to the device like any other.
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

#include "point_range.h"

namespace parlayANN {

struct euclid_u8_parameters {           // Euclidian_Point<uint8_t>::parameters (euclidian_point.h:100-110)
  float slope = 1.0f; int32_t offset = 0; int dims = 0;
  bool identity() const { return slope == 1.0f && offset == 0; }
};

// min / max over every coordinate, both starting at 0; all-integer non-negative data below 256 keeps its values
template <class FloatRange>
euclid_u8_parameters generate_parameters_u8(const FloatRange& pr) {
  float lo = 0.0f, hi = 0.0f;
  bool integral = true;
  const long d = pr.dimension();
  for (size_t i = 0; i < pr.size(); i++) {
    const float* v = (const float*)pr.location((long)i);
    for (long j = 0; j < d; j++) {
      const float x = v[j];
      integral = integral && x >= 0 && (x - (float)(long)x) == 0;
      lo = std::min(lo, x); hi = std::max(hi, x);
    }
  }
  if (integral) { if (hi < 256) hi = 255; lo = 0; }
  euclid_u8_parameters p;
  p.slope = 255 / (hi - lo);
  p.offset = (int32_t)std::round(lo * p.slope);
  p.dims = (int)d;
  std::cout << "scalar quantization: min value = " << lo << ", max value = " << hi << std::endl;
  return p;
}

template <class FloatRange>
PointRange<Euclidian_Point<uint8_t>> quantize_u8(const FloatRange& pr, const euclid_u8_parameters& p) {
  const long d = pr.dimension();
  std::vector<uint8_t> q(pr.size() * (size_t)d);
  for (size_t i = 0; i < pr.size(); i++) {
    const float* v = (const float*)pr.location((long)i);
    uint8_t* o = q.data() + i * (size_t)d;
    for (long j = 0; j < d; j++) {
      if (p.identity()) { o[j] = (uint8_t)v[j]; continue; }
      int64_t r = (int64_t)std::round(v[j] * p.slope) - p.offset;
      o[j] = (uint8_t)std::min<int64_t>(std::max<int64_t>(r, 0), 255);
    }
  }
  return PointRange<Euclidian_Point<uint8_t>>(q.data(), pr.size(), (unsigned int)d);
}

// largest magnitude (or the 1e-4 / 1-1e-4 quantiles when trim) over every coordinate
template <class FloatRange>
float generate_max_val_mips_i8(const FloatRange& pr, bool trim) {
  const long d = pr.dimension();
  std::vector<float> vals;
  vals.reserve(pr.size() * (size_t)d);
  for (size_t i = 0; i < pr.size(); i++) {
    const float* v = (const float*)pr.location((long)i);
    vals.insert(vals.end(), v, v + d);
  }
  const long len = (long)vals.size();
  float lo, hi;
  if (trim) {
    const float cutoff = .0001f;
    const long a = (long)(cutoff * len), b = (long)((1.0 - cutoff) * (len - 1));
    std::nth_element(vals.begin(), vals.begin() + a, vals.end()); lo = vals[a];
    std::nth_element(vals.begin(), vals.begin() + b, vals.end()); hi = vals[b];
  } else {
    const auto mm = std::minmax_element(vals.begin(), vals.end());
    lo = *mm.first; hi = *mm.second;
  }
  const float mv = std::max(hi, -lo);
  std::cout << "scalar quantization: min value = " << lo << ", max value = " << hi << std::endl;
  return mv;
}

template <class FloatRange>
PointRange<Mips_Point<int8_t>> quantize_mips_i8(const FloatRange& pr, float max_val) {
  const long d = pr.dimension();
  const int half = 255 / 2;
  const float scale = half / max_val;
  std::vector<int8_t> q(pr.size() * (size_t)d);
  for (size_t i = 0; i < pr.size(); i++) {
    const float* v = (const float*)pr.location((long)i);
    int8_t* o = q.data() + i * (size_t)d;
    for (long j = 0; j < d; j++) {
      const float x = v[j];
      if (x < -max_val) o[j] = (int8_t)(-half);
      else if (x > max_val) o[j] = (int8_t)half;
      else o[j] = (int8_t)(int32_t)std::round(x * scale);
    }
  }
  return PointRange<Mips_Point<int8_t>>(q.data(), pr.size(), (unsigned int)d);
}

// Point::normalize for every row of a float range (mips_point.h:113-122, euclidian_point.h:150-158; `-normalize`,
// neighborsTime.C:147-153): norm accumulated in double over float products, inverse taken in float
template <class FloatRange>
void normalize_range(FloatRange& pr) {
  const long d = pr.dimension();
  for (size_t i = 0; i < pr.size(); i++) {
    float* v = (float*)pr.location((long)i);
    double norm = 0.0;
    for (long j = 0; j < d; j++) norm += v[j] * v[j];
    norm = std::sqrt(norm);
    if (norm == 0) norm = 1.0;
    const float inv_norm = (float)(1.0 / norm);
    for (long j = 0; j < d; j++) v[j] = v[j] * inv_norm;
  }
  pr.touch();      // coordinates changed in place: device mirrors of this range are stale
}

}  // namespace parlayANN
