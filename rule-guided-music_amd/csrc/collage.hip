// collage.hip -- DiffCollage window split / conditional-independence merge for long sequences.
//
// Reference: diff_collage/w_img.py:8-24 (split_wimg: unfold into n windows of width 128, stride 128-overlap),
// :26-48 (avg_merge_wimg: fold-sum [/ coverage]); diff_collage/condind_long.py:24-51 and
// condind_circle.py:41-84 (eps = sum_i full_i - sum_{i<n-1} half_i on the right overlaps; circular variant
// appends the first `overlap` columns and averages the seam).
// Pure gather / scatter-free sums (each output element reads its <=2 covering windows): bandwidth-trivial.
#include "common.h"

namespace rgm {
constexpr int BASE = 128;

// wins[(b*n + i)][c][y][x] = src(b, c, y, i*stride + x), src = long image, circularly extended when wrap > 0
__global__ void collage_split_kernel(const float* __restrict__ img, float* __restrict__ wins, float* __restrict__ halves,
                                     int B, int Cc, int h, int W, int n, int stride, int ov) {
  const long long total = (long long)B * n * Cc * h * BASE;
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % BASE);
  long long r = i / BASE;
  const int y = (int)(r % h);
  r /= h;
  const int c = (int)(r % Cc);
  r /= Cc;
  const int wi = (int)(r % n);
  const int b = (int)(r / n);
  int X = wi * stride + x;
  if (X >= W) X -= W;   // circular extension (only reached when the caller split an extended image)
  const float v = img[(((long long)b * Cc + c) * h + y) * W + X];
  wins[i] = v;
  if (halves && x >= BASE - ov)
    halves[((((long long)b * n + wi) * Cc + c) * h + y) * ov + (x - (BASE - ov))] = v;
}

__device__ __forceinline__ float long_eps_at(const float* __restrict__ full, const float* __restrict__ half, int b, int c,
                                             int y, int X, int Cc, int h, int n, int stride, int ov, int is_avg) {
  // windows covering column X: i in [ceil((X-127)/stride), floor(X/stride)]
  float s = 0.f;
  int cnt = 0;
  int i1 = X / stride;
  if (i1 > n - 1) i1 = n - 1;
  for (int i = i1; i >= 0 && X - i * stride < BASE; --i) {
    const int x = X - i * stride;
    const long long row = (((long long)b * n + i) * Cc + c) * h + y;
    float v = full[row * BASE + x];
    if (half && i != n - 1 && x >= BASE - ov) v -= half[row * ov + (x - (BASE - ov))];
    s += v;
    ++cnt;
  }
  return is_avg ? s / (float)cnt : s;
}

// out (B,C,h,Wout): linear: Wout = n*128 - (n-1)*ov ; circle: Wout = n*(128-ov), seam averaged
__global__ void collage_merge_kernel(const float* __restrict__ full, const float* __restrict__ half, float* __restrict__ out,
                                     int B, int Cc, int h, int n, int ov, int circle, int is_avg) {
  const int stride = BASE - ov;
  const int Wl = n * BASE - (n - 1) * ov;            // width of the (extended) long image
  const int Wout = circle ? Wl - ov : Wl;
  const long long total = (long long)B * Cc * h * Wout;
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int X = (int)(i % Wout);
  long long r = i / Wout;
  const int y = (int)(r % h);
  r /= h;
  const int c = (int)(r % Cc);
  const int b = (int)(r / Cc);
  float v = long_eps_at(full, half, b, c, y, X, Cc, h, n, stride, ov, is_avg);
  if (circle && X < ov) v = (v + long_eps_at(full, half, b, c, y, Wl - ov + X, Cc, h, n, stride, ov, is_avg)) / 2.0f;
  out[i] = v;
}
}  // namespace rgm

using namespace rgm;

extern "C" int rgm_collage_split(const float* img, float* wins, float* halves, int B, int C, int h, int W, int n, int overlap,
                                 void* stream) {
  RGM_REQUIRE(img && wins && B > 0 && C > 0 && h > 0 && n > 0 && overlap >= 0 && overlap < BASE, "collage_split: bad arguments");
  const int stride = BASE - overlap;
  RGM_REQUIRE((n - 1) * stride + BASE <= W + overlap, "collage_split: %d windows do not fit width %d", n, W);
  const long long total = (long long)B * n * C * h * BASE;
  hipLaunchKernelGGL(collage_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, img, wins, halves,
                     B, C, h, W, n, stride, overlap);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}

extern "C" int rgm_collage_merge(const float* full, const float* half, float* out, int B, int C, int h, int n, int overlap,
                                 int circle, int is_avg, void* stream) {
  RGM_REQUIRE(full && out && B > 0 && C > 0 && h > 0 && n > 0 && overlap >= 0 && overlap < BASE, "collage_merge: bad arguments");
  const int Wl = n * BASE - (n - 1) * overlap;
  const long long total = (long long)B * C * h * (circle ? Wl - overlap : Wl);
  hipLaunchKernelGGL(collage_merge_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, full, half, out, B,
                     C, h, n, overlap, circle, is_avg);
  RGM_LAUNCH_CHECK();
  return RGM_OK;
}
