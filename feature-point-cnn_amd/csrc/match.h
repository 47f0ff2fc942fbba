// match.h -- brute-force descriptor matching, the step right after the path in both demos:
//   Python: cv2.BFMatcher(cv2.NORM_L2, crossCheck=True).match(query, train)
//           (python/src/inference.py:88-96)
//   C++   : SearchKeyFrameCorrespondence -- the FIRST descriptor of the current frame whose L2
//           distance to a key-frame descriptor is below the tolerance (cpp/src/main.cc:9-29,79-92)
//
// S = Q . T^T is a [nq x nt x D] GEMM on the fp32 matrix cores; both operands are already in the
// per-lane order MFMA wants (a row of 128 floats per descriptor, lane = row, 16 bytes per step), so
// they are read straight from global memory.  d^2 = |q|^2 + |t|^2 - 2 q.t; the row / column
// arg-minima are merged across workgroups with 64-bit atomicMin on (float bits of d^2, index):
// exact, order-independent, ties go to the lower index (OpenCV scans the train set in ascending
// order and keeps the first minimum).
#pragma once
#include "conv_mfma.h"

namespace fpc {

struct MatchArgs {
  const float* q;   // [nq][D]
  const float* t;   // [nt][D]
  int nq, nt;
  int D;            // descriptor length, a multiple of 8: 128 (python net) / 256 (C++ net)
  unsigned long long* rowbest;  // [nq]  (d^2 bits << 32 | t index), pre-filled with ~0
  unsigned long long* colbest;  // [nt]  (d^2 bits << 32 | q index), pre-filled with ~0
  unsigned int* first;          // [nq]  lowest t index with d < tol (first-within mode), pre-filled with ~0
  float tol2;                   // tol^2; < 0: arg-min mode
};

__global__ __launch_bounds__(256) void match_gemm_kernel(const MatchArgs a) {
  // per-workgroup arg-min tables: the four waves of a 128 x 128 tile reduce into LDS first, then ONE global 64-bit
  // atomicMin per row and per column of the tile
  __shared__ unsigned long long s_row[128], s_col[128];
  __shared__ unsigned int s_first[128];
  // each wave's 64 x 64 tile of squared distances (pitch 68 floats): the arg-min over a row / a column is then a
  // plain loop of one lane over LDS instead of a 64-bit butterfly of cross-lane shuffles per accumulator row
  __shared__ __attribute__((aligned(16))) float s_d2[4][64 * 68];
  if (threadIdx.x < 128) {
    s_row[threadIdx.x] = ~0ull;
    s_col[threadIdx.x] = ~0ull;
    s_first[threadIdx.x] = ~0u;
  }
  __syncthreads();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int q0 = blockIdx.x * 128 + (wave >> 1) * 64, t0 = blockIdx.y * 128 + (wave & 1) * 64;
  // operand rows of this lane (clamped; out-of-range rows / columns are masked in the epilogue)
  const float* qrow[2];
  const float* trow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    qrow[i] = a.q + (size_t)min(q0 + i * 32 + l31, a.nq - 1) * a.D + half * 4;
    trow[i] = a.t + (size_t)min(t0 + i * 32 + l31, a.nt - 1) * a.D + half * 4;
  }
  f32x16 acc[2][2];
  float qn[2] = {0.f, 0.f}, tn[2] = {0.f, 0.f};  // partial squared norms of this lane's rows (its half of each k8)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float4 qa[2], ta[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    qa[i] = *reinterpret_cast<const float4*>(qrow[i]);
    ta[i] = *reinterpret_cast<const float4*>(trow[i]);
  }
  const int K8 = a.D / 8;
  for (int k8 = 0; k8 < K8; ++k8) {
    float4 qc[2], tc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      qc[i] = qa[i];
      tc[i] = ta[i];
      const int kn = k8 + 1 < K8 ? k8 + 1 : k8;
      qa[i] = *reinterpret_cast<const float4*>(qrow[i] + kn * 8);
      ta[i] = *reinterpret_cast<const float4*>(trow[i] + kn * 8);
      qn[i] += qc[i].x * qc[i].x + qc[i].y * qc[i].y + qc[i].z * qc[i].z + qc[i].w * qc[i].w;
      tn[i] += tc[i].x * tc[i].x + tc[i].y * tc[i].y + tc[i].z * tc[i].z + tc[i].w * tc[i].w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const float af = j == 0 ? qc[mi].x : j == 1 ? qc[mi].y : j == 2 ? qc[mi].z : qc[mi].w;
          const float bf = j == 0 ? tc[ni].x : j == 1 ? tc[ni].y : j == 2 ? tc[ni].z : tc[ni].w;
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[mi][ni], 0, 0, 0);
        }
  }
  // full squared norms: lanes l and l+32 hold the two halves of the same row
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    qn[i] += __shfl_xor(qn[i], 32);
    tn[i] += __shfl_xor(tn[i], 32);
  }
  // C/D map: column (t) = lane & 31, row (q) = (r&3) + 8*(r>>2) + 4*half.  d2 -> this wave's LDS tile
  float* tile = s_d2[wave];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rowl = (r & 3) + 8 * (r >> 2) + 4 * half;
      const float qnr = __shfl(qn[mi], rowl);                         // |q|^2 of that row lives in lane `rowl`
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        float d2 = qnr + tn[ni] - 2.f * acc[mi][ni][r];
        tile[(mi * 32 + rowl) * 68 + ni * 32 + l31] = d2 > 0.f ? d2 : 0.f;
      }
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the tile is private to this wave: no workgroup barrier needed
  {  // lane = row of the tile: arg-min (lowest column index on ties) and first-within over its 64 columns
    const int qi = q0 + lane;
    const int ncol = min(64, a.nt - t0);                 // valid columns of this tile (may be <= 0)
    float best = INFINITY;
    int bj = -1;
    unsigned int firstj = ~0u;
    const float4* rowp = reinterpret_cast<const float4*>(tile + lane * 68);
#pragma unroll 4
    for (int j4 = 0; j4 < 16; ++j4) {
      const float4 v = rowp[j4];
      const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int j = j4 * 4 + k;
        if (j < ncol) {
          if (e[k] < best) { best = e[k]; bj = j; }
          if (a.tol2 >= 0.f && e[k] < a.tol2 && firstj == ~0u) firstj = (unsigned)(t0 + j);
        }
      }
    }
    if (qi < a.nq && bj >= 0) {
      const int rl = (wave >> 1) * 64 + lane;
      if (a.tol2 < 0.f) atomicMin(&s_row[rl], ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)(t0 + bj));
      else if (firstj != ~0u) atomicMin(&s_first[rl], firstj);
    }
  }
  if (a.tol2 < 0.f) {  // lane = column of the tile: arg-min over its 64 rows (for the cross check)
    const int tj = t0 + lane;
    const int nrow = min(64, a.nq - q0);
    float best = INFINITY;
    int bi = -1;
#pragma unroll 8
    for (int i = 0; i < 64; ++i) {
      const float e = tile[i * 68 + lane];
      if (i < nrow && e < best) { best = e; bi = i; }
    }
    if (tj < a.nt && bi >= 0)
      atomicMin(&s_col[(wave & 1) * 64 + lane], ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)(q0 + bi));
  }
  __syncthreads();
  if (tid < 128) {
    const int qi = blockIdx.x * 128 + tid;
    if (qi < a.nq) {
      if (a.tol2 < 0.f) {
        if (s_row[tid] != ~0ull) atomicMin(a.rowbest + qi, s_row[tid]);
      } else if (s_first[tid] != ~0u) {
        atomicMin(a.first + qi, s_first[tid]);
      }
    }
  } else if (a.tol2 < 0.f) {
    const int tj = blockIdx.y * 128 + tid - 128;
    if (tj < a.nt && s_col[tid - 128] != ~0ull) atomicMin(a.colbest + tj, s_col[tid - 128]);
  }
}

__global__ __launch_bounds__(256) void match_finalize_kernel(const unsigned long long* rowbest, const unsigned long long* colbest,
                                                             int nq, int cross_check, float max_dist, int32_t* match,
                                                             float* dist) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nq) return;
  const unsigned long long rb = rowbest[i];
  const int j = (int)(rb & 0xffffffffu);
  const float d = sqrtf(__uint_as_float((unsigned)(rb >> 32)));
  bool ok = rb != ~0ull;
  if (ok && cross_check) ok = (int)(colbest[j] & 0xffffffffu) == i;
  if (ok && max_dist > 0.f) ok = d < max_dist;
  match[i] = ok ? j : -1;
  if (dist) dist[i] = d;
}

__global__ __launch_bounds__(256) void first_finalize_kernel(const unsigned int* first, int nq, int32_t* out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < nq) out[i] = first[i] == ~0u ? -1 : (int)first[i];
}

}  // namespace fpc
