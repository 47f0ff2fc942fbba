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
}  // namespace fpc
