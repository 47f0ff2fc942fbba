// homography.h -- device side of homography adaptation (python/src/homographies.py:250-324):
// perspective warps of frames and probability maps, erosion of the validity masks, count-normalised aggregation.
// All of it is HBM-bound element-wise work; the 1 + num network passes in between are the path itself.
//
// The reference delegates the arithmetic to libraries that are absent from this image:
//   * homography_transform = torchvision.transforms.functional_tensor.perspective (homographies.py:215-216):
//     restated here from torchvision 0.10's `_perspective_grid` + torch's grid_sample(padding zeros,
//     align_corners = False): output pixel (x, y) samples the input at
//         X = x + 0.5, Y = y + 0.5,  den = c6 X + c7 Y + 1,
//         gx = (c0 X + c1 Y + c2) / (0.5 W) / den - 1      (all in fp32, in this order)
//         ix = ((gx + 1) W - 1) / 2                         (grid_sample's un-normalisation)
//     bilinear with zero padding, or nearest (nearbyint).  The sampling arithmetic is pinned against torch itself
//     (fixture F8: torch.nn.functional.grid_sample on a grid built by this formula); the formula is from memory of
//     torchvision's source -- PARITY UNPINNED for it.
//   * erode = cv2.erode with cv2.getStructuringElement(MORPH_ELLIPSE, (2r, 2r)), constant border 0
//     (homographies.py:238-247): OpenCV's published ellipse rasterisation (row i: dy = i - r,
//     dx = round(r sqrt((r^2 - dy^2) / r^2)), columns [r - dx, r + dx], anchor (r, r)) -- PARITY UNPINNED.
#pragma once
#include <hip/hip_runtime.h>

namespace fpc {

struct WarpCoeffs {
  float c[8];
};

__device__ __forceinline__ void warp_source(const WarpCoeffs& k, int x, int y, int W, int H, float* ix, float* iy) {
  const float X = (float)x + 0.5f, Y = (float)y + 0.5f;
  const float hw = 0.5f * (float)W, hh = 0.5f * (float)H;
  // base_grid . (theta1^T / [0.5 W, 0.5 H]) and base_grid . theta2^T: three-term dot products in fp32
  const float nx = __fmaf_rn(1.0f, k.c[2] / hw, __fmaf_rn(Y, k.c[1] / hw, X * (k.c[0] / hw)));
  const float ny = __fmaf_rn(1.0f, k.c[5] / hh, __fmaf_rn(Y, k.c[4] / hh, X * (k.c[3] / hh)));
  const float den = __fmaf_rn(1.0f, 1.0f, __fmaf_rn(Y, k.c[7], X * k.c[6]));
  const float gx = nx / den - 1.0f, gy = ny / den - 1.0f;
  *ix = ((gx + 1.0f) * (float)W - 1.0f) / 2.0f;
  *iy = ((gy + 1.0f) * (float)H - 1.0f) / 2.0f;
}

__device__ __forceinline__ float sample_plane(const float* p, int W, int H, float ix, float iy, int nearest) {
  if (nearest) {
    const float rx = nearbyintf(ix), ry = nearbyintf(iy);
    if (!(rx >= 0.f && rx <= (float)(W - 1) && ry >= 0.f && ry <= (float)(H - 1))) return 0.f;
    return p[(size_t)(int)ry * W + (int)rx];
  }
  const float fx = floorf(ix), fy = floorf(iy);
  if (!(fx >= -1.f && fx <= (float)W && fy >= -1.f && fy <= (float)H)) return 0.f;  // also rejects NaN / huge values
  const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
  // ATen's grid_sampler_2d: nw = (ix_se - ix) * (iy_se - iy), ne = (ix - ix_sw) * (iy_sw - iy), ...
  const float wx0 = (float)x1 - ix, wx1 = ix - (float)x0, wy0 = (float)y1 - iy, wy1 = iy - (float)y0;
  const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
  float v = 0.f;
  if (vy0 && vx0) v += p[(size_t)y0 * W + x0] * (wx0 * wy0);
  if (vy0 && vx1) v += p[(size_t)y0 * W + x1] * (wx1 * wy0);
  if (vy1 && vx0) v += p[(size_t)y1 * W + x0] * (wx0 * wy1);
  if (vy1 && vx1) v += p[(size_t)y1 * W + x1] * (wx1 * wy1);
  return v;
}

// out[p][y][x] = sample(in[p], source(x, y)) for `planes` planes of H x W (frames: planes = n * C; maps: planes = n).
// `src_is_ones`: the input is torch.ones (the validity masks): no plane is read.
__global__ __launch_bounds__(256) void warp_perspective_kernel(const float* in, float* out, int planes, int H, int W,
                                                               WarpCoeffs k, int nearest, int src_is_ones) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)H * W) return;
  const int y = i / W, x = i - (size_t)y * W;
  float ix, iy;
  warp_source(k, x, y, W, H, &ix, &iy);
  if (src_is_ones) {
    float v;
    if (nearest) {
      const float rx = nearbyintf(ix), ry = nearbyintf(iy);
      v = (rx >= 0.f && rx <= (float)(W - 1) && ry >= 0.f && ry <= (float)(H - 1)) ? 1.f : 0.f;
    } else {
      v = 0.f;  // not used by the reference
    }
    out[i] = v;
    return;
  }
  for (int p = 0; p < planes; ++p) out[(size_t)p * H * W + i] = sample_plane(in + (size_t)p * H * W, W, H, ix, iy, nearest);
}

// cv2.erode(img, getStructuringElement(MORPH_ELLIPSE, (2r, 2r)), borderType = CONSTANT, borderValue = 0) on one plane
__global__ __launch_bounds__(256) void erode_ellipse_kernel(const float* in, float* out, int H, int W, int r) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)H * W) return;
  const int y = i / W, x = i - (size_t)y * W;
  const double inv_r2 = r ? 1.0 / ((double)r * r) : 0.0;
  float m = INFINITY;
  for (int a = 0; a < 2 * r; ++a) {
    const int dy = a - r;
    const int dx = (int)rint((double)r * sqrt(((double)r * r - (double)dy * dy) * inv_r2));
    const int j1 = max(r - dx, 0), j2 = min(r + dx + 1, 2 * r);
    const int yy = y + dy;
    for (int j = j1; j < j2; ++j) {
      const int xx = x + j - r;
      const float v = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? in[(size_t)yy * W + xx] : 0.f;
      m = fminf(m, v);
    }
  }
  out[i] = m;
}

// maps[p] *= mask   (warped_prob * mask, homographies.py:298)
__global__ __launch_bounds__(256) void mul_mask_kernel(float* maps, const float* mask, int planes, size_t HW) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= HW) return;
  const float m = mask[i];
  for (int p = 0; p < planes; ++p) maps[(size_t)p * HW + i] *= m;
}

// One view: proj = perspective(warped_prob, H_inv, bilinear) * count; sum += proj; mx = max(mx, proj); cnt += count
// (homographies.py:299-305 and the reductions :311-313).  first != 0 initialises with the un-warped view
// (probs = net(image), counts = 1: :269-270).
__global__ __launch_bounds__(256) void unwarp_accumulate_kernel(const float* wprob, WarpCoeffs kinv, const float* count,
                                                                float* sum, float* mx, float* cnt, int planes, int H,
                                                                int W, int first) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t HW = (size_t)H * W;
  if (i >= HW) return;
  if (first) {
    for (int p = 0; p < planes; ++p) {
      const float v = wprob[(size_t)p * HW + i];
      sum[(size_t)p * HW + i] = v;
      mx[(size_t)p * HW + i] = v;
    }
    cnt[i] = 1.f;
    return;
  }
  const int y = i / W, x = i - (size_t)y * W;
  float ix, iy;
  warp_source(kinv, x, y, W, H, &ix, &iy);
  const float c = count[i];
  for (int p = 0; p < planes; ++p) {
    const float v = sample_plane(wprob + (size_t)p * HW, W, H, ix, iy, 0) * c;
    sum[(size_t)p * HW + i] += v;
    mx[(size_t)p * HW + i] = fmaxf(mx[(size_t)p * HW + i], v);
  }
  cnt[i] += c;
}

// prob = where(counts >= num // 3, aggregation == max ? max : sum / counts, 0)   (homographies.py:311-324)
__global__ __launch_bounds__(256) void aggregate_kernel(const float* sum, const float* mx, const float* cnt, float* out,
                                                        int planes, size_t HW, float min_count, int use_max) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= HW) return;
  const float c = cnt[i];
  for (int p = 0; p < planes; ++p) {
    const float v = use_max ? mx[(size_t)p * HW + i] : sum[(size_t)p * HW + i] / c;
    out[(size_t)p * HW + i] = c >= min_count ? v : 0.f;
  }
}

}  // namespace fpc
