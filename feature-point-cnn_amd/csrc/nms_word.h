// nms_word.h -- the NMS state word of a candidate (shared by kernels_misc.h and the fused softmax epilogue of
// block_bf16.h).  State map word: 0 = empty / suppressed, float bits + 1 (> 0) = undecided candidate, that | 0x80000000
// = kept.  +1 so that an undecided candidate is never the word 0 -- with conf_thresh == 0 a candidate may have
// p == +0.0 -- and never has the sign bit; -0.0 counts as +0.0.  Bits of non-negative floats order like the floats,
// and so do bits + 1.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fpc {
__device__ __forceinline__ uint32_t nms_state_word(float p) { return (p == 0.f ? 0u : __float_as_uint(p)) + 1u; }
__device__ __forceinline__ float nms_state_conf(uint32_t word) { return __uint_as_float((word & 0x7fffffffu) - 1u); }

// The arithmetic of the exp-softmax (superpoint.py:111-112), shared by softmax_d2s_kernel (kernels_misc.h) and the fused
// epilogue of FPC_BF16's detector.layer.1 (block_bf16.h).
// FAST (FPC_BF16 only, round 3): the exponential as v_exp_f32 of x * log2(e) and ONE reciprocal per cell instead of 64
// IEEE divisions -- in the fused epilogue of that mode's detector.layer.1 (block_bf16.h) the libm expf and the division
// sequences were ~1 250 of a lane's instructions, 17 k of a tile's 39 k cycles.  Relative error ~1e-6 on probabilities
// whose logits carry bf16's eight bits; the fp32 modes keep expf and the division (FAST = false).  The epilogue and
// softmax_d2s_kernel<true> use these two functions and the same order of operations: bit-identical maps.
template <bool FAST>
__device__ __forceinline__ float sm_exp(float x) {
  if constexpr (FAST) return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f);
  else return expf(x);
}
template <bool FAST>
__device__ __forceinline__ float sm_scale(float den) {   // what a cell's exps are multiplied with (FAST) / divided by
  if constexpr (FAST) return __builtin_amdgcn_rcpf(den);
  else return den;
}
template <bool FAST>
__device__ __forceinline__ float sm_prob(float e, float scale) {
  if constexpr (FAST) return e * scale;
  else return e / scale;
}

}  // namespace fpc
