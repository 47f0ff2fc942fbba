// fpc_api.hip -- the C-ABI of include/fpc.h: context, weight packing, launch plan.
//
// Data layout in HBM (all fp32): activations NHWC with the channel count padded to a
// multiple of 8 (65 -> 72); one buffer per tensor of the network sized for max_batch
// frames; the transposed-conv output and the encoder features share ONE 256-channel
// buffer (`cat`), so torch.cat (python/src/superpoint.py:59) costs nothing; weights
// live in one packed blob (BN folded, MFMA fragment order) that stays L2-resident.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <chrono>
#include <string>
#include <vector>

#include "../../include/fpc.h"
#include "block_mfma.h"
#include "block_bf16.h"
#include "convt_bf16.h"
#include "block_x3.h"
#include "stem_bf16.h"
#include "wblock_mfma.h"
#include "wblock16_mfma.h"
#include "wblock36_mfma.h"
#include "wblock36p_mfma.h"
#include "conv_mfma.h"
#include "kernels_misc.h"
#include "queue_map.h"
#include "match.h"
#include "homography.h"
#include "weights.h"

namespace fpc {

constexpr int BLOB_HEADER_FLOATS = 16;
constexpr uint32_t BLOB_MAGIC = 0x57435046u;   // "FPCW"

static thread_local std::string g_hip_err;

#define HIPCHECK(expr)                                                                   \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) {                                                              \
      g_hip_err = std::string(#expr) + ": " + hipGetErrorString(e_);                     \
      return FPC_E_HIP;                                                                  \
    }                                                                                    \
  } while (0)

// ------------------------------------------------------------------------------------
// Kernel instances.  KIND(name, TH,TW, S,EXT, KC, WM,WN, MB,NB)
// ------------------------------------------------------------------------------------
#define FPC_KINDS(X)                                                        \
  X(T816_3x3_K64_N64, 8, 16, 1, 3, 64, 4, 1, 1, 2)                          \
  X(T816_1x1_K64_N64, 8, 16, 1, 1, 64, 4, 1, 1, 2)                          \
  X(T620_3x3s2_K32_N128, 6, 20, 2, 3, 32, 2, 2, 2, 2)                       \
  X(T620_3x3_K64_N128, 6, 20, 1, 3, 64, 2, 2, 2, 2)                         \
  X(T620_1x1_K64_N128, 6, 20, 1, 1, 64, 2, 2, 2, 2)                         \
  X(T620_2x2_K64_N128, 6, 20, 1, 2, 64, 2, 2, 2, 2)                         \
  X(T320_2x2_K64_N128, 3, 20, 1, 2, 64, 1, 4, 2, 1)                         \
  X(T320_1x1_K64_N128, 3, 20, 1, 1, 64, 1, 4, 2, 1)                         \
  X(T620_3x3_K64_N96, 6, 20, 1, 3, 64, 4, 1, 1, 3)                          \
  X(T620_1x1_K64_N96, 6, 20, 1, 1, 64, 4, 1, 1, 3)                          \
  X(T620_3x3_K72_N96, 6, 20, 1, 3, 72, 4, 1, 1, 3)                          \
  X(T620_1x1_K72_N96, 6, 20, 1, 1, 72, 4, 1, 1, 3)

enum Kind {
#define X(name, ...) K_##name,
  FPC_KINDS(X)
#undef X
      K_COUNT
};

struct KindInfo {
  const char* name;
  const char* symbol;   // as rocprofv3 prints it
  int TH, TW, S, EXT, KC, WM, WN, MB, NB;
  int lds_bytes, threads;
  const void* fn;
  void (*launch)(const ConvArgs&, dim3, hipStream_t);
};

#define X(name, TH, TW, S, EXT, KC, WM, WN, MB, NB)                                                    \
  static void launch_##name(const ConvArgs& a, dim3 grid, hipStream_t st) {                            \
    using C = ConvCfg<TH, TW, S, EXT, KC, WM, WN, MB, NB>;                                             \
    hipLaunchKernelGGL((conv_mfma_kernel<TH, TW, S, EXT, KC, WM, WN, MB, NB>), grid, dim3(C::NT),      \
                       C::LDS_BYTES, st, a);                                                           \
  }
FPC_KINDS(X)
#undef X

static const KindInfo g_kinds[K_COUNT] = {
#define X(name, TH, TW, S, EXT, KC, WM, WN, MB, NB)                                                    \
  {#name, "conv_mfma_kernel<" #TH ", " #TW ", " #S ", " #EXT ", " #KC ", " #WM ", " #WN ", " #MB ", " #NB ">", TH, TW, S, EXT, KC, WM, WN, MB, NB, ConvCfg<TH, TW, S, EXT, KC, WM, WN, MB, NB>::LDS_BYTES,   \
   WM * WN * 64, (const void*)conv_mfma_kernel<TH, TW, S, EXT, KC, WM, WN, MB, NB>, launch_##name},
    FPC_KINDS(X)
#undef X
};

// Round 5: conv2_mfma_kernel (conv_mfma.h: the same convolution with a quarter of the VALU instructions) for the two
// instances of the default fp32 plan -- the ConvTranspose phases and descriptor.layer_in.1's 1x1; launches it does not
// cover (a second K source, workgroups that store only part of their N channels) and every other instance keep
// conv_mfma_kernel.
#define FPC_LEAN_KINDS(X)                          \
  X(T320_2x2_K64_N128, 3, 20, 1, 2, 64, 1, 4, 2, 1) \
  X(T320_1x1_K64_N128, 3, 20, 1, 1, 64, 1, 4, 2, 1)
struct LeanKindInfo {
  Kind kind;
  const char* symbol;
  int lds_bytes;
  const void* fn;
  void (*launch)(const ConvArgs&, dim3, hipStream_t);
};
#define X(name, TH, TW, S, EXT, KC, WM, WN, MB, NB)                                                              \
  static void launch_lean_##name(const ConvArgs& a, dim3 grid, hipStream_t st) {                                 \
    using C2 = Conv2Cfg<TH, TW, S, EXT, KC, WM, WN, MB, NB>;                                                    \
    hipLaunchKernelGGL((conv2_mfma_kernel<TH, TW, S, EXT, KC, WM, WN, MB, NB>), grid, dim3(WM* WN * 64),         \
                       C2::LDS_BYTES, st, a);                                                                    \
  }
FPC_LEAN_KINDS(X)
#undef X
static const LeanKindInfo g_lean_kinds[] = {
#define X(name, TH, TW, S, EXT, KC, WM, WN, MB, NB)                                                                       \
  {K_##name, "conv2_mfma_kernel<" #TH ", " #TW ", " #S ", " #EXT ", " #KC ", " #WM ", " #WN ", " #MB ", " #NB ">",          \
   Conv2Cfg<TH, TW, S, EXT, KC, WM, WN, MB, NB>::LDS_BYTES, (const void*)conv2_mfma_kernel<TH, TW, S, EXT, KC, WM, WN, MB, NB>, \
   launch_lean_##name},
    FPC_LEAN_KINDS(X)
#undef X
};
static const LeanKindInfo* lean_kind(Kind k) {
  for (const LeanKindInfo& l : g_lean_kinds)
    if (l.kind == k) return &l;
  return nullptr;
}

// Fused ResNetBlock instances.  BKIND(name, TH,TW, S, KC, WM,WN, MB,NB, CMIDP)
#define FPC_BLOCK_KINDS(X)                                   \
  X(B816_s1_K64_C64, 8, 16, 1, 64, 4, 1, 1, 2, 64)           \
  X(B1616_s1_K32_C64, 16, 16, 1, 32, 4, 1, 2, 2, 64)         \
  X(B620_s2_K32_C128, 6, 20, 2, 32, 2, 2, 2, 2, 128)         \
  X(B620_s1_K64_C128, 6, 20, 1, 64, 2, 2, 2, 2, 128)         \
  X(B620_s1_K64_C72, 6, 20, 1, 64, 4, 1, 1, 3, 72)           \
  X(B620_s1_K72_C72, 6, 20, 1, 72, 4, 1, 1, 3, 72)           \
  X(B320_s2_K32_C256, 3, 20, 2, 32, 1, 4, 2, 2, 256)         \
  X(B310_s2_K32_C256, 3, 10, 2, 32, 1, 4, 1, 2, 256)         \
  X(B320_s2_K32_C128, 3, 20, 2, 32, 1, 4, 2, 1, 128)         \
  X(B320_s1_K64_C256, 3, 20, 1, 64, 1, 4, 2, 2, 256)

enum BKind {
#define X(name, ...) BK_##name,
  FPC_BLOCK_KINDS(X)
#undef X
      BK_COUNT
};

struct BKindInfo {
  const char* name;
  const char* symbol;
  int TH, TW, S, KC, WM, WN, MB, NB, CMIDP;
  int lds_bytes;
  const void* fn;
  void (*launch)(const BlockArgs&, dim3, hipStream_t);
};

#define X(name, TH, TW, S, KC, WM, WN, MB, NB, CMIDP)                                                     \
  static void launchb_##name(const BlockArgs& a, dim3 grid, hipStream_t st) {                             \
    using BC = BlockCfg<TH, TW, S, KC, WM, WN, MB, NB, CMIDP>;                                            \
    hipLaunchKernelGGL((block_mfma_kernel<TH, TW, S, KC, WM, WN, MB, NB, CMIDP>), grid, dim3(WM* WN * 64), \
                       BC::LDS_BYTES, st, a);                                                             \
  }
FPC_BLOCK_KINDS(X)
#undef X

static const BKindInfo g_bkinds[BK_COUNT] = {
#define X(name, TH, TW, S, KC, WM, WN, MB, NB, CMIDP)                                                      \
  {#name, "block_mfma_kernel<" #TH ", " #TW ", " #S ", " #KC ", " #WM ", " #WN ", " #MB ", " #NB ", " #CMIDP ">", \
   TH, TW, S, KC, WM, WN, MB, NB, CMIDP, BlockCfg<TH, TW, S, KC, WM, WN, MB, NB, CMIDP>::LDS_BYTES,         \
   (const void*)block_mfma_kernel<TH, TW, S, KC, WM, WN, MB, NB, CMIDP>, launchb_##name},
    FPC_BLOCK_KINDS(X)
#undef X
};

// Winograd ResNetBlock instances.  WKIND(name, KC, NBT, CMID): wblock_mfma_kernel (generation 1: 32x32x2 blocks, a wave
// owns 4 positions x 64 channels, phases in lockstep); W16KIND(name, NCG): wblock16_kernel (generation 2: 16x16x4 blocks,
// a wave owns 16 channels x all positions, input side pipelined into the GEMM), N = 16 NCG output channels.
#define FPC_WBLOCK_KINDS(X)     \
  X(W816_K32_C64, 32, 2, 64)    \
  X(W816_K32_C128, 32, 4, 128)  \
  X(W816_K32_C72, 32, 3, 72)    \
  X(W816_K24_C72, 24, 3, 72)
#define FPC_W16_KINDS(X) \
  X(W16_C64, 4, 8)       \
  X(W16_C128, 8, 8)      \
  X(W16_C128H, 8, 4)
// W36KIND(name, NB, TYT, TXT): wblock36_kernel (generation 3: Winograd F(4x4,3x3), one wave per SIMD, 16 Winograd tiles
// of 4x4 pixels arranged TYT x TXT per workgroup), N = 64 NB output channels.
#define FPC_W36_KINDS(X) \
  X(W36_C64_4x4, 1, 4, 4)  \
  X(W36_C64_2x8, 1, 2, 8)  \
  X(W36_C128_4x4, 2, 4, 4) \
  X(W36_C128_2x8, 2, 2, 8)

// W36PKIND(name, TYT, TXT): wblock36p_kernel (round 5: the 64-channel instance at TWO waves per SIMD -- 512 threads, the 36
// positions split between the two waves of a SIMD; same packed weights as W36_C64_*)
#define FPC_W36P_KINDS(X) \
  X(W36P_C64_4x4, 4, 4)   \
  X(W36P_C64_2x8, 2, 8)

// W36PDKIND(name, TYT, TXT): wblock36p_dust_kernel -- the paired instance + the detector's 65th channel (same dust block)
#define FPC_W36PD_KINDS(X) \
  X(W36PD_C65_4x4, 4, 4)   \
  X(W36PD_C65_2x8, 2, 8)

// W36DKIND(name, TYT, TXT): wblock36_dust_kernel -- the 64-channel instance + the detector's 65th channel on the VALU
#define FPC_W36D_KINDS(X) \
  X(W36_C65_4x4, 4, 4)    \
  X(W36_C65_2x8, 2, 8)

enum WKind {
#define X(name, ...) WK_##name,
  FPC_WBLOCK_KINDS(X)
  FPC_W16_KINDS(X)
  FPC_W36_KINDS(X)
  FPC_W36D_KINDS(X)
  FPC_W36P_KINDS(X)
  FPC_W36PD_KINDS(X)
#undef X
      WK_COUNT
};

struct WKindInfo {
  const char* name;
  const char* symbol;
  int gen;               // 1: wblock_mfma_kernel, 2: wblock16_kernel, 3: wblock36_kernel
  int KC, NBT, CMID, lds_bytes;
  int NCG;               // generations 2, 3: channel groups of 16
  int TH;                // tile height in pixels (8; 4 for the latency instance; generation 3: 16 or 8)
  int TW;                // tile width in pixels (16; generation 3: 16 or 32)
  int threads;           // workgroup size (512; generation 3: 256)
  const void* fn;
  void (*launch)(const WBlockArgs&, dim3, hipStream_t);
  bool dust = false;     // generation 3: the 65-channel (64 + dustbin) instance
};

#define X(name, KC, NBT, CMID)                                                                            \
  static void launchw_##name(const WBlockArgs& a, dim3 grid, hipStream_t st) {                            \
    constexpr int lds = WBlockCfg<KC, NBT, CMID>::LDS_BYTES;                                              \
    hipLaunchKernelGGL((wblock_mfma_kernel<KC, NBT, CMID>), grid, dim3(512), lds, st, a);                 \
  }
FPC_WBLOCK_KINDS(X)
#undef X
#define X(name, NCG, TH)                                                                                  \
  static void launchw_##name(const WBlockArgs& a, dim3 grid, hipStream_t st) {                            \
    constexpr int lds = W16Cfg<NCG, TH>::LDS_BYTES;                                                       \
    hipLaunchKernelGGL((wblock16_kernel<NCG, TH>), grid, dim3(512), lds, st, a);                          \
  }
FPC_W16_KINDS(X)
#undef X
#define X(name, NB, TYT, TXT)                                                                             \
  static void launchw_##name(const WBlockArgs& a, dim3 grid, hipStream_t st) {                            \
    constexpr int lds = W36Cfg<NB, TYT, TXT>::LDS_BYTES;                                                  \
    hipLaunchKernelGGL((wblock36_kernel<NB, TYT, TXT>), grid, dim3(256), lds, st, a);                     \
  }
FPC_W36_KINDS(X)
#undef X
#define X(name, TYT, TXT)                                                                                  \
  static void launchw_##name(const WBlockArgs& a, dim3 grid, hipStream_t st) {                            \
    constexpr int lds = W36Cfg<1, TYT, TXT, true>::LDS_BYTES;                                             \
    hipLaunchKernelGGL((wblock36_dust_kernel<TYT, TXT>), grid, dim3(256), lds, st, a);                    \
  }
FPC_W36D_KINDS(X)
#undef X
#define X(name, TYT, TXT)                                                                                  \
  static void launchw_##name(const WBlockArgs& a, dim3 grid, hipStream_t st) {                            \
    constexpr int lds = W36PCfg<TYT, TXT>::LDS_BYTES;                                                     \
    hipLaunchKernelGGL((wblock36p_kernel<TYT, TXT>), grid, dim3(512), lds, st, a);                        \
  }
FPC_W36P_KINDS(X)
#undef X
#define X(name, TYT, TXT)                                                                                  \
  static void launchw_##name(const WBlockArgs& a, dim3 grid, hipStream_t st) {                            \
    constexpr int lds = W36PCfg<TYT, TXT, true>::LDS_BYTES;                                               \
    hipLaunchKernelGGL((wblock36p_dust_kernel<TYT, TXT>), grid, dim3(512), lds, st, a);                   \
  }
FPC_W36PD_KINDS(X)
#undef X

static const WKindInfo g_wkinds[WK_COUNT] = {
#define X(name, KC, NBT, CMID)                                                                            \
  {#name, "wblock_mfma_kernel<" #KC ", " #NBT ", " #CMID ">", 1, KC, NBT, CMID,                           \
   WBlockCfg<KC, NBT, CMID>::LDS_BYTES, 0, 8, 16, 512, (const void*)wblock_mfma_kernel<KC, NBT, CMID>, launchw_##name},
    FPC_WBLOCK_KINDS(X)
#undef X
#define X(name, NCG, TH)                                                                                  \
  {#name, "wblock16_kernel<" #NCG ", " #TH ">", 2, 16, NCG / 2, NCG * 16,                                  \
   W16Cfg<NCG, TH>::LDS_BYTES, NCG, TH, 16, 512, (const void*)wblock16_kernel<NCG, TH>, launchw_##name},
    FPC_W16_KINDS(X)
#undef X
#define X(name, NB, TYT, TXT)                                                                             \
  {#name, "wblock36_kernel<" #NB ", " #TYT ", " #TXT ">", 3, 16, 2 * NB, 64 * NB,                          \
   W36Cfg<NB, TYT, TXT>::LDS_BYTES, 4 * NB, 4 * TYT, 4 * TXT, 256, (const void*)wblock36_kernel<NB, TYT, TXT>, launchw_##name},
    FPC_W36_KINDS(X)
#undef X
#define X(name, TYT, TXT)                                                                                 \
  {#name, "wblock36_dust_kernel<" #TYT ", " #TXT ">", 3, 16, 2, 64,                                        \
   W36Cfg<1, TYT, TXT, true>::LDS_BYTES, 4, 4 * TYT, 4 * TXT, 256, (const void*)wblock36_dust_kernel<TYT, TXT>, launchw_##name, true},
    FPC_W36D_KINDS(X)
#undef X
#define X(name, TYT, TXT)                                                                                 \
  {#name, "wblock36p_kernel<" #TYT ", " #TXT ", 3>", 3, 16, 2, 64,                                            \
   W36PCfg<TYT, TXT>::LDS_BYTES, 4, 4 * TYT, 4 * TXT, 512, (const void*)wblock36p_kernel<TYT, TXT>, launchw_##name},
    FPC_W36P_KINDS(X)
#undef X
#define X(name, TYT, TXT)                                                                                 \
  {#name, "wblock36p_dust_kernel<" #TYT ", " #TXT ", 3>", 3, 16, 2, 64,                                    \
   W36PCfg<TYT, TXT, true>::LDS_BYTES, 4, 4 * TYT, 4 * TXT, 512, (const void*)wblock36p_dust_kernel<TYT, TXT>, launchw_##name, true},
    FPC_W36PD_KINDS(X)
#undef X
};

// blob floats of a generation-2 / generation-3 kernel's conv1 fragments: [channel group][chunk of 16][position][64 lanes]
// float4 + zero positions behind every group (the fragment rings read ahead)
static size_t w1_floats(const WKindInfo& k, int nchunk) {
  if (k.gen == 3) return (size_t)k.NCG * ((size_t)nchunk * 36 + (k.NCG == 8 ? W36Cfg<2, 4, 4>::WPAD : W36Cfg<1, 4, 4>::WPAD)) * 256;
  return (size_t)k.NCG * ((size_t)nchunk * 16 + W16Cfg<8>::WPAD) * 256;
}
// MFMA FLOPs one frame's tiles issue: positions x Winograd tiles x channels x K for the 3x3, pixels x channels x K for the 1x1
static double wkind_mfma_flops(const WKindInfo& k, int tiles, int nchunk, int k8_1x1) {
  const double px = (double)k.TH * k.TW, n = k.NBT * 32.0;
  if (k.gen == 3) return 2.0 * tiles * n * (36.0 * (px / 16) * nchunk * k.KC + px * k8_1x1 * 8.0);
  return 2.0 * tiles * n * (16.0 * (px / 4) * nchunk * k.KC + px * k8_1x1 * 8.0);
}
// Generation 3 addresses its tensors with signed 32-bit byte offsets below W36_MARKER (wblock36_mfma.h).  A launch is
// given pointers to ITS first frame (round 4; round 3 addressed from the batch's frame 0 and sent a layer whose whole
// tensor did not fit -- the C++ network's full-resolution layers at 32 frames -- to generation 2) and is split into
// several launches where even its own frames do not fit; only a layer whose single frame is too large falls back.
static bool w36_fits(int /*B*/, int H, int W, int cs_in, int cs_out) {
  return (unsigned long long)H * W * (unsigned long long)std::max(cs_in, cs_out) * sizeof(float) < (unsigned long long)W36_MARKER;
}
static int w36_frames_per_launch(int H, int W, int cs_in, int cs_out) {
  const unsigned long long fb = (unsigned long long)H * W * (unsigned long long)std::max(cs_in, cs_out) * sizeof(float);
  return (int)std::max<unsigned long long>(1, ((unsigned long long)W36_MARKER - 1) / fb);
}
// conv-only layers wider than one instance: two 128-channel parts or four 64-channel ones (NB = 1 tiles take 0.53 of an
// NB = 2 tile's time) -- whichever needs fewer tile times on `cus` CUs for `tiles` tiles per part
static int w36_conv_part(int cout, int tiles, int cus) {
  if (cout < 128) return 64;
  if (cout == 128) return 128;
  const double t2 = std::ceil((double)tiles * (cout / 128) / cus) * 1.0, t1 = std::ceil((double)tiles * (cout / 64) / cus) * 0.53;
  return t1 < t2 ? 64 : 128;
}
// Generation 3: the arrangement of the 16 Winograd tiles (4 x 4 or 2 x 8) that covers the map with fewer tiles
// (cout == 64, `paired`: the two-waves-per-SIMD instance, wblock36p_mfma.h -- the default since round 5; a block's
// projection must then be a multiple of 64 channels wide, which every 64-channel layer of both networks is)
static WKind w36_kind(int cout, int H, int W, bool paired = false) {
  const int t44 = ((H + 15) / 16) * ((W + 15) / 16), t28 = ((H + 7) / 8) * ((W + 31) / 32);
  if (cout == 65 && paired) return t28 < t44 ? WK_W36PD_C65_2x8 : WK_W36PD_C65_4x4;
  if (cout == 65) return t28 < t44 ? WK_W36_C65_2x8 : WK_W36_C65_4x4;
  if (cout == 64 && paired) return t28 < t44 ? WK_W36P_C64_2x8 : WK_W36P_C64_4x4;
  if (cout == 64) return t28 < t44 ? WK_W36_C64_2x8 : WK_W36_C64_4x4;
  return t28 < t44 ? WK_W36_C128_2x8 : WK_W36_C128_4x4;
}

// bf16 / split-operand instances.  FKIND(name, KERNEL, CFG, PLANES, TH,TW, S,EXT, KC, WM,WN, MB,NB, CMIDP)
//   PLANES 1: block_bf16_kernel (dtype = FPC_BF16); 3: block_x3_kernel (dtype = FPC_F32_SPLIT)
#define FPC_BF16_KINDS(X)                                                                   \
  X(F816_s1_K64_C64, block_bf16_one_kernel, BlockBfCfg, 1, 8, 32, 1, 3, 64, 2, 2, 4, 1, 64) \
  X(F816_s1_K64_C128, block_bf16_kernel, BlockBfCfg, 1, 8, 16, 1, 3, 64, 1, 4, 4, 1, 128)   \
  X(F816_s1_K80_C80, block_bf16_one_kernel, BlockBfCfg, 1, 8, 16, 1, 3, 80, 4, 1, 1, 3, 80) \
  X(F816_s1_K128_C128, block_bf16_one_kernel, BlockBfCfg, 1, 8, 16, 1, 3, 128, 1, 4, 4, 1, 128) \
  X(F816_s1_K128_C80w, block_bf16_one_kernel, BlockBfCfg, 1, 8, 16, 1, 3, 128, 1, 4, 4, 1, 80) \
  X(F620_s2_K64_C128, block_bf16_one_kernel, BlockBfCfg, 1, 6, 20, 2, 3, 64, 2, 2, 2, 2, 128) \
  X(F320_s2_K128_C256, block_bf16_one_kernel, BlockBfCfg, 1, 3, 20, 2, 3, 128, 1, 4, 2, 2, 256) \
  X(F416_s1_K128_C256, block_bf16_two_kernel, BlockBfCfg, 1, 4, 16, 1, 3, 128, 1, 4, 2, 2, 256) \
  X(F816_ct_K64_C128, block_bf16_kernel, BlockBfCfg, 1, 8, 16, 1, 2, 64, 1, 4, 4, 1, 128)   \
  X(F816_ctf_K64_C128, convt_bf16_kernel, ConvTBfCfg, 1, 8, 16, 1, 2, 64, 2, 2, 2, 1, 128)  \
  X(S816_s1_K64_C64, block_x3_kernel, BlockX3Cfg, 3, 8, 16, 1, 3, 64, 2, 2, 2, 1, 64)       \
  X(S620_s2_K16_C128, block_x3_kernel, BlockX3Cfg, 3, 6, 20, 2, 3, 16, 2, 2, 2, 2, 128)     \
  X(S620_s1_K64_C128, block_x3_kernel, BlockX3Cfg, 3, 6, 20, 1, 3, 64, 2, 2, 2, 2, 128)     \
  X(S620_s1_K64_C80, block_x3_kernel, BlockX3Cfg, 3, 6, 20, 1, 3, 64, 4, 1, 1, 3, 80)       \
  X(S620_s1_K16_C80, block_x3_kernel, BlockX3Cfg, 3, 6, 20, 1, 3, 16, 4, 1, 1, 3, 80)       \
  X(S320_s2_K16_C256, block_x3_kernel, BlockX3Cfg, 3, 3, 20, 2, 3, 16, 1, 4, 2, 2, 256)     \
  X(S320_s1_K64_C256, block_x3_kernel, BlockX3Cfg, 3, 3, 20, 1, 3, 64, 1, 4, 2, 2, 256)     \
  X(S620_ct_K64_C128, block_x3_kernel, BlockX3Cfg, 3, 6, 20, 1, 2, 64, 2, 2, 2, 2, 128)     \
  X(S620_1x1_K64_C80, block_x3_kernel, BlockX3Cfg, 3, 6, 20, 1, 1, 64, 4, 1, 1, 3, 80)      \
  X(S320_1x1_K64_C256, block_x3_kernel, BlockX3Cfg, 3, 3, 20, 1, 1, 64, 1, 4, 2, 2, 256)     \
  X(H816_s1_K64_C64, block_h2_kernel, BlockH2Cfg, 2, 8, 16, 1, 3, 64, 2, 2, 2, 1, 64)     \
  X(H620_s2_K16_C128, block_h2_kernel, BlockH2Cfg, 2, 6, 20, 2, 3, 16, 2, 2, 2, 2, 128)     \
  X(H620_s1_K64_C128, block_h2_kernel, BlockH2Cfg, 2, 6, 20, 1, 3, 64, 2, 2, 2, 2, 128)     \
  X(H620_s1_K64_C80, block_h2_kernel, BlockH2Cfg, 2, 6, 20, 1, 3, 64, 4, 1, 1, 3, 80)     \
  X(H620_s1_K16_C80, block_h2_kernel, BlockH2Cfg, 2, 6, 20, 1, 3, 16, 4, 1, 1, 3, 80)     \
  X(H320_s2_K16_C256, block_h2_kernel, BlockH2Cfg, 2, 3, 20, 2, 3, 16, 1, 4, 2, 2, 256)     \
  X(H320_s1_K64_C256, block_h2_kernel, BlockH2Cfg, 2, 3, 20, 1, 3, 64, 1, 4, 2, 2, 256)     \
  X(H620_ct_K64_C128, block_h2_kernel, BlockH2Cfg, 2, 6, 20, 1, 2, 64, 2, 2, 2, 2, 128)     \
  X(H620_1x1_K64_C80, block_h2_kernel, BlockH2Cfg, 2, 6, 20, 1, 1, 64, 4, 1, 1, 3, 80)     \
  X(H320_1x1_K64_C256, block_h2_kernel, BlockH2Cfg, 2, 3, 20, 1, 1, 64, 1, 4, 2, 2, 256)

enum FKind {
#define X(name, ...) FK_##name,
  FPC_BF16_KINDS(X)
#undef X
      FK_COUNT
};

struct FKindInfo {
  const char* name;
  const char* symbol;
  int planes, TH, TW, S, EXT, KC, WM, WN, MB, NB, CMIDP;
  int lds_bytes;
  const void* fn;
  void (*launch)(const BlockBfArgs&, dim3, hipStream_t);
};

#define X(name, KERN, CFG, PL, TH, TW, S, EXT, KC, WM, WN, MB, NB, CMIDP)                                  \
  static void launchf_##name(const BlockBfArgs& a, dim3 grid, hipStream_t st) {                            \
    constexpr int lds = CFG<TH, TW, S, EXT, KC, WM, WN, MB, NB, CMIDP>::LDS_BYTES;                         \
    hipLaunchKernelGGL((KERN<TH, TW, S, EXT, KC, WM, WN, MB, NB, CMIDP>), grid, dim3(WM* WN * 64), lds, st, a); \
  }
FPC_BF16_KINDS(X)
#undef X

static const FKindInfo g_fkinds[FK_COUNT] = {
#define X(name, KERN, CFG, PL, TH, TW, S, EXT, KC, WM, WN, MB, NB, CMIDP)                                   \
  {#name, #KERN "<" #TH ", " #TW ", " #S ", " #EXT ", " #KC ", " #WM ", " #WN ", " #MB ", " #NB ", " #CMIDP ">", \
   PL, TH, TW, S, EXT, KC, WM, WN, MB, NB, CMIDP, CFG<TH, TW, S, EXT, KC, WM, WN, MB, NB, CMIDP>::LDS_BYTES,   \
   (const void*)KERN<TH, TW, S, EXT, KC, WM, WN, MB, NB, CMIDP>, launchf_##name},
    FPC_BF16_KINDS(X)
#undef X
};

// ------------------------------------------------------------------------------------
// Launch plan
// ------------------------------------------------------------------------------------
enum OpType { OP_STEM, OP_POOL, OP_CONV, OP_BLOCK, OP_WBLOCK, OP_BF16, OP_SOFTMAX, OP_NMS, OP_DESC,
              OP_VCONV0, OP_POOL2, OP_L2NORM };

struct Op {
  OpType type;
  std::string name;
  Kind kind = K_COUNT;
  ConvArgs args{};
  BKind bkind = BK_COUNT;
  BlockArgs bargs{};
  WKind wkind = WK_COUNT;
  WBlockArgs wargs{};
  FKind fkind = FK_COUNT;
  BlockBfArgs fargs{};
  int phase = -1;              // ConvTranspose output-parity phase of a bf16 conv-only op
  int when = 0;                // 0: always; 1: only in calls of a few frames; 2: only in larger calls
  bool shadow = false;         // packed like any op but never launched: the second half of a merged two-half launch (see add_wconv)
  bool wconv = false;          // conv-only Winograd launch: 3x3 conv + (BN or bias) + ReLU, output channels n0 .. n0+127
  int n0 = 0;
  bool plain_conv = false;     // a Conv2d + bias (+ ReLU) of the C++ network: weights `prefix`.weight / .bias, no BN
  bool lean = false;           // OP_CONV on conv2_mfma_kernel (round 5; add_conv decides)
  int ksize = 0;
  const float* pin = nullptr;  // OP_POOL2 / OP_L2NORM / OP_VCONV0 operands
  float* pout = nullptr;
  int pH = 0, pW = 0, pC = 0;
  std::string prefix;          // checkpoint prefix of a fused block
  int cin = 0, cout = 0;       // real channel counts of a fused block
  int grid_y = 1, grid_z = 1;
  double flops_per_frame = 0;  // algorithmic: 2 * MACs of the real (unpadded) convolution
  double mfma_flops_per_frame = 0;  // issued on the matrix cores (padding included; Winograd: 16/36 of the 3x3)
  bool fused_softmax_capable = false;   // FPC_BF16's detector.layer.1 (block_bf16.h: fused exp-softmax epilogue)
  double bytes_per_frame = 0;       // algorithmic HBM bytes: the launch's input tensor(s) read once + its output written once
  bool descriptor_branch = false;
};

struct Timing {
  hipEvent_t start, stop;
  int op;
  int frames;  // frames covered by this launch (a sub-batch)
};

}  // namespace fpc

using namespace fpc;

struct fpc_ctx {
  fpc_config cfg{};
  int H = 0, W = 0, B = 0, Hc = 0, Wc = 0;
  bool vgg = false;                  // cfg.arch == FPC_ARCH_VGG: superpoint::SPModel (cpp/src/model.cc)
  int D = 128;                       // descriptor length: 128 (python net) / 256 (C++ net, settings.h:25)
  std::vector<float*> vbuf;          // VGG activation buffers
  size_t vconv0_off = 0;
  bool bf16 = false;                 // cfg.dtype == FPC_BF16
  bool split = false;                // cfg.dtype == FPC_F32_SPLIT or FPC_F32_SPLIT_F16
  bool split_f16 = false;            // ... the latter: two fp16 terms per operand, three MFMAs per product
  int lgcs = 72;                     // channel stride of the logits buffer (80 in bf16 mode)
  int cin = 3;                       // 3: [n,3,H,W] frames (the reference's layout); 1: gray [n,1,H,W]
  int cap = 0, sort_cap = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipStream_t upload = nullptr;       // fpc_upload_stream: the caller's copy stream, on a hardware queue of its own
  bool queue_probe = true;            // fpc_create placed the streams on hardware queues of their own (queue_map.h; FPC_QUEUE_PROBE)
  int probe_rounds = 0;               // ... with this many probe rounds, in this much host time (fpc_stream_report)
  double placement_ms = 0.0;
  std::vector<std::pair<hipStream_t, int>> caller_stream_memo;   // fpc_set_stream: queue classes of the caller's streams seen before (at most 4)
  std::vector<hipStream_t> aux;      // extra streams for sub-batches
  std::vector<hipEvent_t> ev_join;
  std::vector<hipStream_t> side;     // per sub-batch: detector head + NMS next to the descriptor head
  std::vector<hipEvent_t> ev_enc, ev_det;
  bool nms_aside = true;             // FPC_NMS_ASIDE=0: NMS in line on the sub-batch stream
  bool split_heads = false;          // FPC_SPLIT_HEADS=1: detector head + NMS of a sub-batch on a side stream next to its descriptor head
  hipEvent_t ev_fork = nullptr;
  int min_sub = 8;                   // smallest sub-batch worth its own stream (FPC_MIN_SUB); calls below twice this take the latency plan
  int num_cus = 256;
  int fkind_blocks_per_cu[64] = {};   // resident workgroups per CU of every bf16 / split-operand instance (fpc_create)
  int persist_min_tiles = 1;         // FPC_PERSIST_MIN: tiles per CU from which the Winograd kernel runs persistent (0 = never)
  bool xcd_order = true;             // FPC_XCD_ORDER=0: plain tile order in the persistent Winograd kernel
  int nms_passes = 2;
  bool fuse_softmax = true;        // !FPC_PLAN_NO_FUSED_SOFTMAX (FPC_BF16's detector.layer.1, block_bf16.h)
  bool convt_fused = true;         // !FPC_PLAN_CONVT_PHASES (FPC_BF16's ConvTranspose as ONE launch, convt_bf16.h)
  bool logits_valid = false;       // the last call wrote the logits ("det.1" of fpc_read_activation): false after a fused-softmax fpc_detect
  bool nms_one_workgroup = false;  // FPC_PLAN_NMS_ONE_WORKGROUP: round 1's sort (one workgroup per frame) for every frame
  int nms_g = 0;                     // FPC_NMS_G: workgroups per frame of the NMS rounds kernel (0 = 512 / frames, at most 16)
  bool fuse_stem_pool = true;        // conv1+bn1+relu+max_pool in one launch (FPC_FUSE_STEM=0: two)
  bool conv_lean = true;             // OP_CONV launches on conv2_mfma_kernel where it applies (FPC_PLAN_CONV_ROUND1 / FPC_CONV_LEAN=0: conv_mfma_kernel)
  bool stem_lean = true;             // ... on stem_pool2_kernel where the conv map is whole 16 x 16 tiles (FPC_PLAN_STEM_ROUND3 / FPC_STEM_LEAN=0: stem_pool_kernel)
  bool stem2 = false;                // decided in fpc_create: stem_lean, fp32 MFMA mode, fused stem + pool, whole tiles
  bool layer1_t816 = false;          // direct (non-Winograd) layer1 blocks on 8x16 tiles instead of 16x16
#ifdef FPC_DIAG
  unsigned long long* diag_stamps = nullptr;
  int diag_n = 0;
#endif
  bool winograd_in1 = true;          // descriptor.layer_in.1 (256 ch): conv-only Winograd x2 + 1x1 (FPC_WINOGRAD_IN1=0: fused direct block)
  bool winograd_det = true;          // ... also the detector's 65-channel blocks (FPC_WINOGRAD_DET=0: direct)
  int w36_cus = 0;                   // FPC_W36_CUS: CUs a generation-3 launch may take (0 = all): A/B knob for contexts that share the GPU
  bool w36_paired = true;            // the 64-channel F(4x4,3x3) layers on wblock36p_kernel, two waves per SIMD (FPC_PLAN_W36_ONE_WAVE / FPC_W36_PAIRED=0: wblock36_kernel<1, ..>)
  bool winograd_det_gen3 = true;     // ... on wblock36_dust_kernel in batch calls (FPC_PLAN_DETECTOR_GEN1 / FPC_WINOGRAD_DET_GEN=1: round 1's kernel)
  bool winograd = true;              // stride-1 blocks with <= 128 channels: Winograd F(2x2,3x3) (FPC_WINOGRAD=0: direct)
  bool latency_tiles = true;         // calls of a few frames run the 128-channel Winograd blocks on 4 x 16 tiles (FPC_LATENCY_TILES=0: 8 x 16)
  int winograd_gen = 3;              // 64- and 128-channel Winograd layers on wblock36_kernel (3: F(4x4,3x3)), wblock16_kernel (2) or wblock_mfma_kernel (1; FPC_WINOGRAD_GEN)
  bool fuse_blocks = true;           // one launch per ResNetBlock (FPC_FUSE=0: conv1 / conv2 launches)
  bool weights_loaded = false;
  bool plan_error = false;           // a layer asked for a kernel instance that does not exist (fpc_create -> FPC_E_INVALID)

  // one slab for all activations / results; carved below
  char* slab = nullptr;
  size_t slab_bytes = 0;
  bool guard_zones = false;                                    // FPC_PLAN_GUARD_ZONES
  std::vector<std::pair<size_t, size_t>> guards;               // (offset, bytes) of the canary zones in the slab
  float *stem_out, *x0, *h4, *x1, *x2, *h8, *x3, *cat, *dh, *dproj, *d0, *lg;
  float *h16, *y16a, *y16b, *lo_h, *lo0, *desc_map, *desc_in_nhwc;
  float* prob;
  uint32_t *nmsmap, *cand;
  int32_t *ncand, *count, *xy, *status;
  float* rowmax = nullptr;          // [B][H / 8]: the largest logit of every row of cells of the last call
  uint32_t* range = nullptr;        // [B][FPC_RANGE_WORDS]: largest logit / descriptor-map value, non-finite input flag, per frame
  float *conf, *desc_out;
  unsigned long long* sort_scratch;
  int32_t* nms_aux = nullptr;  // [B][NMS_AUX_INTS] (kernels_misc.h: nms_chunk_sort_kernel)
  unsigned long long *rowbest, *colbest;  // descriptor matching workspace, `cap` entries each

  float* u8stage = nullptr;          // fpc_detect_u8: converted frames [B,cin,H,W], allocated on first use
  char* ha_ws = nullptr;             // fpc_homography_adaptation: workspace, allocated on first use

  // packed weights
  float* blob = nullptr;
  size_t blob_floats = 0;
  uint32_t* bcast_tag = nullptr;   // 64 bytes of the slab: fpc_broadcast_weights' first collective (allocated here so that it cannot fail there)
  std::vector<float> host_blob;

  std::vector<Op> ops;
  StemArgs stem{};
  size_t stem_w_off = 0, stem_b_off = 0;
  struct ConvW {
    size_t w_off[4] = {0, 0, 0, 0}, b_off = 0, b2_off = 0;
  };
  std::vector<ConvW> convw;  // parallel to ops (unused entries for non-conv ops)

  bool timing = false;               // this call is timed (LaunchTimer)
  int timing_every = 0, timing_calls = 0;   // fpc_set_timing(n): one fpc_detect call in n is timed (0 = off)
  std::vector<Timing> timings;
  std::vector<hipEvent_t> event_pool;
  size_t events_used = 0;
};

namespace fpc {

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// FPC_PLAN_GUARD_ZONES (a test facility, include/fpc.h): `guard` bytes of a canary pattern behind EVERY carved buffer and
// GUARD_TAIL bytes behind the last one -- the out-of-range store this exists for ("a fault worth
// recording") landed W36_MARKER = 2 GiB behind its tensor, so the tail zone spans that distance from any buffer of a
// small configuration.  fpc_check_guards counts the words that no longer hold the pattern.
constexpr size_t GUARD_BYTES = 64 << 10;
constexpr size_t GUARD_TAIL = (size_t)0x80000000u + (4u << 20);
constexpr uint32_t GUARD_PATTERN = 0xA5C3F00Du;

struct Carver {
  size_t off = 0;
  size_t guard = 0;
  std::vector<std::pair<size_t, size_t>>* zones = nullptr;   // (offset, bytes) of every guard zone
  template <typename T>
  size_t take(size_t n) {
    const size_t o = off;
    off = align_up(off + n * sizeof(T), 256);
    if (guard) {
      zones->push_back({off, guard});
      off += guard;
    }
    return o;
  }
  void finish() {
    if (guard) {
      zones->push_back({off, GUARD_TAIL});
      off += GUARD_TAIL;
    }
  }
};


__global__ void guard_fill_kernel(uint32_t* p, size_t n, uint32_t v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void guard_count_kernel(const uint32_t* p, size_t n, uint32_t v, unsigned long long* bad) {
  unsigned long long mine = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) mine += p[i] != v;
  if (mine) atomicAdd(bad, mine);
}
static int fill_guards(fpc_ctx* c) {
  // (the hipMemset of the workspace in front of this call runs on the null stream and need not have finished when it
  // returns; c->stream is non-blocking, so nothing orders the two -- the first version lost 2 MB of pattern to it.
  // Waited for in every context: the first call's kernels must not race the zeroing either.)
  HIPCHECK(hipDeviceSynchronize());
  if (c->guards.empty()) return FPC_OK;
  for (const auto& z : c->guards) {
    const size_t n = z.second / 4;
    guard_fill_kernel<<<(unsigned)std::min<size_t>(4096, (n + 255) / 256), 256, 0, c->stream>>>(reinterpret_cast<uint32_t*>(c->slab + z.first), n, GUARD_PATTERN);
  }
  if (!c->guards.empty()) HIPCHECK(hipStreamSynchronize(c->stream));
  return FPC_OK;
}

// ---- architecture walk ---------------------------------------------------------------
// Builds ctx->ops with geometry and buffer pointers, and assigns blob offsets.  The
// weight VALUES are filled by pack_all() when a checkpoint arrives.

struct ConvSpec {
  std::string name;
  Kind kind;
  int ksize;             // 3, 1, or 2 (= ConvTranspose phase set)
  int stride;
  const float* in0;      // source 0
  int cs0, cin0, cin0_pad, H0, W0;
  const float* in1 = nullptr;  // optional second K source (1x1)
  int cs1 = 0, cin1 = 0, cin1_pad = 0, s1 = 1, H1 = 0, W1 = 0;
  const float* res = nullptr;
  int csr = 0;
  float* out;
  int cso, cout, nstore, Ho, Wo;
  int relu;
  bool desc_branch;
};

// conv2_mfma_kernel (round 5) takes a launch with one K source whose workgroups store all of their N channels as float4s
static bool conv_is_lean(const fpc_ctx* c, const Op& op) {
  const KindInfo& k = g_kinds[op.kind];
  const ConvArgs& a = op.args;
  return c->conv_lean && lean_kind(op.kind) && a.nchunk1 == 0 && a.nstore == op.grid_y * (k.WN * k.NB * 32) && a.cso % 4 == 0 &&
         (!a.res || a.csr % 4 == 0);
}

static void add_conv(fpc_ctx* c, const ConvSpec& s, size_t* blob_off) {
  const KindInfo& k = g_kinds[s.kind];
  Op op;
  op.type = OP_CONV;
  op.name = s.name;
  op.kind = s.kind;
  op.descriptor_branch = s.desc_branch;
  ConvArgs& a = op.args;
  const int N = k.WN * k.NB * 32;
  const int nbt = (s.cout + N - 1) / N * (N / 32);
  op.grid_y = nbt * 32 / N;
  a.in0 = s.in0;
  a.cs0 = s.cs0;
  a.nchunk0 = s.cin0_pad / k.KC;
  a.in1 = s.in1;
  a.cs1 = s.cs1;
  a.nchunk1 = s.in1 ? s.cin1_pad / k.KC : 0;
  a.s1 = s.s1;
  a.H1 = s.H1;
  a.W1 = s.W1;
  a.H = s.H0;
  a.W = s.W0;
  a.pad = s.ksize == 3 ? 1 : 0;
  a.res = s.res;
  a.csr = s.csr;
  a.out = s.out;
  a.cso = s.cso;
  a.nstore = s.nstore;
  a.relu = s.relu;
  a.nbt = nbt;
  const int HWp = (k.TW - 1) * k.S + k.EXT, ROW4 = k.KC / 4 + 1;
  fpc_ctx::ConvW cw;
  const int K8 = k.KC / 8;
  if (s.ksize == 2) {  // ConvTranspose2d(k3, s2, p1, op1): 4 output-parity phases over the INPUT grid
    op.grid_z = 4;
    a.Ho = s.H0;
    a.Wo = s.W0;
    a.OH = s.Ho;
    a.OW = s.Wo;
    a.oys = a.oxs = 2;
    double macs = 0;
    for (int ph = 0; ph < 4; ++ph) {
      const int py = ph >> 1, px = ph & 1;
      ConvSub& sp = a.sub[ph];
      sp.oy0 = py;
      sp.ox0 = px;
      sp.ntaps = 0;
      for (int iy = 0; iy < (py ? 2 : 1); ++iy)
        for (int ix = 0; ix < (px ? 2 : 1); ++ix) {
          const int dy = py ? 1 - iy : 0, dx = px ? 1 - ix : 0;  // py=1: (dy=1,ky=0), (dy=0,ky=2)
          sp.tapoff4[sp.ntaps++] = (dy * HWp + dx) * ROW4;
        }
      cw.w_off[ph] = *blob_off;
      *blob_off += ((size_t)a.nchunk0 * sp.ntaps * K8 + 2) * nbt * 64 * 4;
      macs += (double)sp.ntaps;
    }
    op.flops_per_frame = 2.0 * macs * s.H0 * s.W0 * s.cin0 * s.cout;
    op.mfma_flops_per_frame = 2.0 * macs * ((s.H0 + k.TH - 1) / k.TH) * ((s.W0 + k.TW - 1) / k.TW) * (k.WM * k.MB * 32.0) * (a.nchunk0 * k.KC) * (nbt * 32.0);
  } else {
    a.Ho = a.OH = s.Ho;
    a.Wo = a.OW = s.Wo;
    a.oys = a.oxs = 1;
    ConvSub& sp = a.sub[0];
    sp.oy0 = sp.ox0 = 0;
    sp.ntaps = s.ksize * s.ksize;
    for (int ky = 0; ky < s.ksize; ++ky)
      for (int kx = 0; kx < s.ksize; ++kx) sp.tapoff4[ky * s.ksize + kx] = (ky * HWp + kx) * ROW4;
    cw.w_off[0] = *blob_off;
    *blob_off += ((size_t)(a.nchunk0 * sp.ntaps + a.nchunk1) * K8 + 2) * nbt * 64 * 4;
    op.flops_per_frame = 2.0 * s.Ho * s.Wo * s.cout * ((double)s.cin0 * sp.ntaps + s.cin1);
    op.mfma_flops_per_frame = 2.0 * ((s.Ho + k.TH - 1) / k.TH) * ((s.Wo + k.TW - 1) / k.TW) * (k.WM * k.MB * 32.0) * (nbt * 32.0) *
                              ((double)a.nchunk0 * k.KC * sp.ntaps + (double)a.nchunk1 * k.KC);
  }
  cw.b_off = *blob_off;
  *blob_off += (size_t)nbt * 32;
  // conv2_mfma_kernel: one K source, every workgroup stores all of its N channels as float4s
  op.lean = conv_is_lean(c, op);
  a.tiles_x = (a.Wo + k.TW - 1) / k.TW;
  a.tiles_y = (a.Ho + k.TH - 1) / k.TH;
  op.bytes_per_frame = 4.0 * ((double)s.cin0 * s.H0 * s.W0 + (s.in1 ? (double)s.cin1 * s.H1 * s.W1 : 0.0) +
                              (s.res ? (double)s.cout * s.Ho * s.Wo : 0.0) + (double)s.cout * s.Ho * s.Wo);
  c->ops.push_back(op);
  c->convw.push_back(cw);
}

struct BlockSpec {
  std::string prefix;
  BKind kind;
  const float* x;
  int csx, cin, cin_pad, H, W;
  float* out;
  int cso, cout, cout_pad;
  bool proj, desc_branch;
};

static void add_block(fpc_ctx* c, const BlockSpec& s, size_t* blob_off) {
  const BKindInfo& k = g_bkinds[s.kind];
  Op op;
  op.type = OP_BLOCK;
  op.name = s.prefix + (s.proj ? " [conv1+bn1+relu+conv2+bn2+proj+relu]" : " [conv1+bn1+relu+conv2+bn2+identity+relu]");
  op.prefix = s.prefix;
  op.bkind = s.kind;
  op.cin = s.cin;
  op.cout = s.cout;
  op.descriptor_branch = s.desc_branch;
  BlockArgs& a = op.bargs;
  const int nbt = k.WN * k.NB, K8 = k.KC / 8;
  a.x = s.x;
  a.csx = s.csx;
  a.x_bytes = (unsigned)std::min<size_t>((size_t)c->B * s.H * s.W * s.csx * 4, 0xffffff00u);
  a.nchunk = s.cin_pad / k.KC;
  a.H = s.H;
  a.W = s.W;
  const int HWp = (k.TW - 1) * k.S + 3, ROW4 = k.KC / 4 + 1;
  for (int ky = 0; ky < 3; ++ky)
    for (int kx = 0; kx < 3; ++kx) a.tapoff4[ky * 3 + kx] = (ky * HWp + kx) * ROW4;
  a.k8_h = k.CMIDP / 8;
  a.k8_x = s.proj ? s.cin_pad / 8 : 0;   // the kernel walks the projection eight K steps at a time
  if (a.k8_x % 8) { fprintf(stderr, "fpc: fused block %s: projection over %d channels is not a multiple of 64\n", s.prefix.c_str(), s.cin_pad); abort(); }
  a.out = s.out;
  a.cso = s.cso;
  a.Ho = s.H / k.S;
  a.Wo = s.W / k.S;
  a.tiles_x = (a.Wo + k.TW - 1) / k.TW;
  a.tiles_y = (a.Ho + k.TH - 1) / k.TH;
  a.nstore = s.cout_pad;
  fpc_ctx::ConvW cw;
  cw.w_off[0] = *blob_off;
  *blob_off += ((size_t)a.nchunk * 9 * K8 + 2) * nbt * 64 * 4;
  cw.b_off = *blob_off;
  *blob_off += (size_t)nbt * 32;
  cw.w_off[1] = *blob_off;
  *blob_off += ((size_t)(a.k8_h + a.k8_x) + 2) * nbt * 64 * 4;
  cw.b2_off = *blob_off;
  *blob_off += (size_t)nbt * 32;
  op.flops_per_frame = 2.0 * a.Ho * a.Wo * s.cout * ((double)s.cin * 9 + s.cout + (s.proj ? s.cin : 0));
  op.mfma_flops_per_frame = 2.0 * a.tiles_x * a.tiles_y * (k.WM * k.MB * 32.0) * (nbt * 32.0) *
                            ((double)a.nchunk * k.KC * 9 + (a.k8_h + a.k8_x) * 8.0);
  op.bytes_per_frame = 4.0 * ((double)s.cin * s.H * s.W + (double)s.cout * a.Ho * a.Wo);
  c->ops.push_back(op);
  c->convw.push_back(cw);
}

// The op just added moves to a finer tile (an instance that consumes the same fragments: same KC and channel blocking).
// Half-size tiles double the workgroups of the stride-2 blocks, the transposed convolution and layer_in.1's 1x1: the
// latency of a single frame's layer is one tile's time (20 - 40 tiles per frame on 256 CUs), and a 32-frame batch runs
// 3 - 10 % faster on them too (shorter tail, more workgroups per CU in flight) -- measured, so they serve every call.
// FPC_PLAN_NO_LATENCY_TILES keeps round 1's tiles.
static void retile_last(fpc_ctx* c, BKind small) {
  if (!c->latency_tiles || c->ops.empty() || c->ops.back().type != OP_BLOCK) return;
  const BKindInfo &k0 = g_bkinds[c->ops.back().bkind], &k = g_bkinds[small];
  if (k.KC != k0.KC || k.WN * k.NB != k0.WN * k0.NB || k.S != k0.S || k.CMIDP != k0.CMIDP) { c->plan_error = true; return; }
  Op& o = c->ops.back();
  o.bkind = small;
  BlockArgs& a = o.bargs;
  const int HWp = (k.TW - 1) * k.S + 3, ROW4 = k.KC / 4 + 1;
  for (int ky = 0; ky < 3; ++ky)
    for (int kx = 0; kx < 3; ++kx) a.tapoff4[ky * 3 + kx] = (ky * HWp + kx) * ROW4;
  a.tiles_x = (a.Wo + k.TW - 1) / k.TW;
  a.tiles_y = (a.Ho + k.TH - 1) / k.TH;
  o.mfma_flops_per_frame = 2.0 * a.tiles_x * a.tiles_y * (k.WM * k.MB * 32.0) * (k.WN * k.NB * 32.0) *
                           ((double)a.nchunk * k.KC * 9 + (a.k8_h + a.k8_x) * 8.0);
}

static void retile_last(fpc_ctx* c, Kind small) {
  if (!c->latency_tiles || c->ops.empty() || c->ops.back().type != OP_CONV) return;
  const KindInfo &k0 = g_kinds[c->ops.back().kind], &k = g_kinds[small];
  if (k.KC != k0.KC || k.WN * k.NB != k0.WN * k0.NB || k.S != k0.S || k.EXT != k0.EXT) { c->plan_error = true; return; }
  Op& o = c->ops.back();
  o.kind = small;
  ConvArgs& a = o.args;
  const int HWp = (k.TW - 1) * k.S + k.EXT, HWp0 = (k0.TW - 1) * k0.S + k0.EXT, ROW4 = k.KC / 4 + 1;
  for (int z = 0; z < o.grid_z; ++z)
    for (int t = 0; t < a.sub[z].ntaps; ++t) {   // offsets were (dy * HWp0 + dx) * ROW4 with dx < EXT <= HWp0
      const int lin = a.sub[z].tapoff4[t] / ROW4, dy = lin / HWp0, dx = lin - dy * HWp0;
      a.sub[z].tapoff4[t] = (dy * HWp + dx) * ROW4;
    }
  a.tiles_x = (a.Wo + k.TW - 1) / k.TW;
  a.tiles_y = (a.Ho + k.TH - 1) / k.TH;
  o.lean = conv_is_lean(c, o);
}

static void add_wblock(fpc_ctx* c, const BlockSpec& s, WKind wk, size_t* blob_off, bool small_only = false) {
  const WKindInfo& k = g_wkinds[wk];
  Op op;
  op.type = OP_WBLOCK;
  op.name = s.prefix + (s.proj ? " [winograd conv1+bn1+relu+conv2+bn2+proj+relu]" : " [winograd conv1+bn1+relu+conv2+bn2+identity+relu]");
  op.prefix = s.prefix;
  op.wkind = wk;
  op.cin = s.cin;
  op.cout = s.cout;
  op.descriptor_branch = s.desc_branch;
  WBlockArgs& a = op.wargs;
  const int K8 = k.KC / 8;
  a.x = s.x;
  a.csx = s.csx;
  a.nchunk = s.cin_pad / k.KC;
  a.dust = nullptr;
  a.dust_in = 0;
  if (k.dust) {
    // the detector's 65 channels: 64 on the MFMAs (this instance's 4 channel groups), channel 64 beside them (W36Dust).
    // Input: 64 k channels through the chunk loop (+ a projection over them), or 65 = 4 chunks + input channel 64
    // (detector.layer.1, identity shortcut).
    if (s.cout != 65 || !(s.cin == 65 ? !s.proj : (s.cin == s.cin_pad && s.cin_pad % 64 == 0 && s.cin_pad <= 256))) { c->plan_error = true; return; }
    a.dust_in = s.cin == 65;
    a.nchunk = a.dust_in ? 4 : s.cin_pad / 16;
  } else if (s.cin_pad % k.KC) { c->plan_error = true; return; }
  a.H = s.H;
  a.W = s.W;
  a.k8_h = k.CMID / 8;
  a.k8_x = s.proj ? s.cin_pad / 8 : 0;   // the kernel walks the projection four K steps at a time
  if (a.k8_x % 4) { fprintf(stderr, "fpc: winograd block %s: projection over %d channels is not a multiple of 32\n", s.prefix.c_str(), s.cin_pad); abort(); }
  a.out = s.out;
  a.cso = s.cso;
  a.tiles_x = (s.W + k.TW - 1) / k.TW;
  a.tiles_y = (s.H + k.TH - 1) / k.TH;
  fpc_ctx::ConvW cw;
  cw.w_off[0] = *blob_off;
  if (k.gen >= 2) {   // [channel group][chunk of 16][position][64 lanes] float4 + zero pad per group (wblock16_mfma.h, wblock36_mfma.h)
    if (a.nchunk < 4 || (a.nchunk & 1)) { c->plan_error = true; return; }
    *blob_off += w1_floats(k, a.nchunk);
  } else {
    *blob_off += ((size_t)a.nchunk * 16 * K8 + 16 * K8 + 2) * k.NBT * 64 * 4;
  }
  cw.b_off = *blob_off;
  *blob_off += (size_t)k.NBT * 32;
  cw.w_off[1] = *blob_off;
  if (k.gen >= 2) *blob_off += (size_t)k.NCG * ((size_t)(a.k8_h + a.k8_x) / 2 + W16Cfg<8>::WPAD) * 256;
  else *blob_off += ((size_t)(a.k8_h + a.k8_x) + 2) * k.NBT * 64 * 4;
  cw.b2_off = *blob_off;
  *blob_off += (size_t)k.NBT * 32;
  if (k.dust) {
    cw.w_off[2] = *blob_off;
    *blob_off += (size_t)W36Dust::floats(a.nchunk);
  }
  op.flops_per_frame = 2.0 * s.H * s.W * s.cout * ((double)s.cin * 9 + s.cout + (s.proj ? s.cin : 0));
  // 16 (36) GEMMs over the tile's Winograd tiles instead of 9 taps over its pixels; then the 1x1 on its pixels
  op.mfma_flops_per_frame = wkind_mfma_flops(k, a.tiles_x * a.tiles_y, a.nchunk, a.k8_h + a.k8_x);
  if (k.dust && a.dust_in) op.mfma_flops_per_frame += 2.0 * a.tiles_x * a.tiles_y * 64.0 * 36 * 16 * 4;   // input channel 64: one K = 4 MFMA per position
  op.bytes_per_frame = 4.0 * ((double)s.cin * s.H * s.W + (double)s.cout * s.H * s.W);
  if (k.dust) {
    // calls of a few frames keep the round-1 instance (its own fragments), as the 64- / 128-channel layers keep generation 2
    op.when = c->latency_tiles ? 2 : 0;
    c->ops.push_back(op);
    c->convw.push_back(cw);
    if (c->latency_tiles) add_wblock(c, s, s.cin == 65 ? WK_W816_K24_C72 : WK_W816_K32_C72, blob_off, true);
    return;
  }
  if (k.gen == 3 && c->latency_tiles) {
    // Generation 3's tile is 256 pixels: a frame has 20 of them at 60 x 80, and the latency of a single frame's layer is
    // one tile's time.  Calls of a few frames keep generation 2 (its own fragments: 16 positions instead of 36), on its
    // 4 x 16 latency tiles where that instance exists.
    op.when = 2;
    c->ops.push_back(op);
    c->convw.push_back(cw);
    add_wblock(c, s, s.cout == 64 ? WK_W16_C64 : WK_W16_C128, blob_off, true);
    return;
  }
  if (small_only && wk == WK_W16_C128) {   // (the caller's large-call op exists already: only the 4 x 16 instance)
    const WKindInfo& kh = g_wkinds[WK_W16_C128H];
    op.when = 1;
    op.wkind = WK_W16_C128H;
    op.wargs.tiles_y = (s.H + kh.TH - 1) / kh.TH;
    op.mfma_flops_per_frame = wkind_mfma_flops(kh, op.wargs.tiles_x * op.wargs.tiles_y, a.nchunk, a.k8_h + a.k8_x);
    c->ops.push_back(op);
    c->convw.push_back(cw);
    return;
  }
  if (small_only) op.when = 1;
  c->ops.push_back(op);
  c->convw.push_back(cw);
  if (wk == WK_W16_C128 && c->latency_tiles) {
    // the same layer on 4 x 16 tiles for calls of a few frames (same fragments: the blob offsets are shared)
    const WKindInfo& kh = g_wkinds[WK_W16_C128H];
    c->ops.back().when = 2;
    Op oh = c->ops.back();
    oh.when = 1;
    oh.wkind = WK_W16_C128H;
    oh.wargs.tiles_y = (s.H + kh.TH - 1) / kh.TH;
    oh.mfma_flops_per_frame = wkind_mfma_flops(kh, oh.wargs.tiles_x * oh.wargs.tiles_y, a.nchunk, a.k8_h + a.k8_x);
    c->ops.push_back(oh);
    c->convw.push_back(cw);
  }
}

// A 3x3 stride-1 convolution + (folded BN | bias) + ReLU on the Winograd kernel in conv-only form.  Output channels
// n0 .. n0 + CMID - 1 of the layer (a 256-wide layer takes two launches).  `bn`: weights are `prefix`.conv1.weight +
// `prefix`.bn1.* (a ResNetBlock's first half); otherwise `prefix`.weight + `prefix`.bias (the C++ network).
static void add_wconv(fpc_ctx* c, const std::string& prefix, bool bn, WKind wk, const float* x, int csx, int cin, int H,
                      int W, float* out, int cso, int cout, int n0, bool desc_branch, size_t* blob_off) {
  const WKindInfo& k = g_wkinds[wk];
  Op op;
  op.type = OP_WBLOCK;
  op.name = prefix + (bn ? ".conv1+bn1+relu [winograd" : " [winograd conv+bias+relu") +
            (cout > k.CMID ? ", channels " + std::to_string(n0) + ".." + std::to_string(n0 + k.CMID - 1) : std::string()) + "]";
  op.prefix = prefix;
  op.wkind = wk;
  op.wconv = true;
  op.plain_conv = !bn;
  op.n0 = n0;
  op.cin = cin;
  op.cout = cout;
  op.descriptor_branch = desc_branch;
  WBlockArgs& a = op.wargs;
  const int K8 = k.KC / 8;
  a.x = x;
  a.csx = csx;
  a.nchunk = cin / k.KC;
  a.H = H;
  a.W = W;
  a.k8_h = a.k8_x = 0;
  a.conv_only = 1;
  a.out = out + n0;
  a.cso = cso;
  a.tiles_x = (W + k.TW - 1) / k.TW;
  a.tiles_y = (H + k.TH - 1) / k.TH;
  fpc_ctx::ConvW cw;
  cw.w_off[0] = *blob_off;
  if (k.gen >= 2) {
    if (a.nchunk < 4 || (a.nchunk & 1)) { c->plan_error = true; return; }
    *blob_off += w1_floats(k, a.nchunk);
  } else {
    *blob_off += ((size_t)a.nchunk * 16 * K8 + 16 * K8 + 2) * k.NBT * 64 * 4;
  }
  cw.b_off = *blob_off;
  *blob_off += (size_t)k.NBT * 32;
  op.flops_per_frame = 2.0 * H * W * (double)k.CMID * cin * 9;
  op.mfma_flops_per_frame = wkind_mfma_flops(k, a.tiles_x * a.tiles_y, a.nchunk, 0);
  op.bytes_per_frame = 4.0 * ((double)cin * H * W + (double)std::min(k.CMID, cout - n0) * H * W);
  const int part = n0 / k.CMID;
  if (k.gen >= 2 && part > 0 && n0 % k.CMID == 0 && cout % k.CMID == 0 && (int)c->ops.size() >= part) {
    // part 1 .. G - 1 of a layer wider than the instance (256 channels as 2 x 128 or 4 x 64): ONE launch computes all
    // parts (gridDim.y = G; the kernel finds part y's fragments and bias y * ysplit_floats behind part 0's and its
    // outputs y * CMID channels behind `out`); this op only carries the part's checkpoint -> fragment packing
    Op& first = c->ops[c->ops.size() - part];
    fpc_ctx::ConvW& cf = c->convw[c->convw.size() - part];
    if (first.wconv && first.prefix == prefix && first.n0 == 0 && first.wkind == wk && first.grid_y == part) {
      const int ys = (int)((cw.w_off[0] - cf.w_off[0]) / part);
      if (part == 1) first.wargs.ysplit_floats = ys;
      else if (first.wargs.ysplit_floats != ys) { c->plan_error = true; return; }
      first.grid_y = part + 1;
      first.flops_per_frame += op.flops_per_frame;
      first.mfma_flops_per_frame += op.mfma_flops_per_frame;
      first.bytes_per_frame = 4.0 * ((double)cin * H * W + (double)std::min(cout, (part + 1) * k.CMID) * H * W);
      const std::string parts = std::to_string(part + 1) + " parts of " + std::to_string(k.CMID) + " channels";
      first.name = prefix + (bn ? ".conv1+bn1+relu [winograd, " : " [winograd conv+bias+relu, ") + parts + "]";
      op.shadow = true;
    }
  }
  c->ops.push_back(op);
  c->convw.push_back(cw);
}

// ---- bf16 plan (block_bf16.h) ---------------------------------------------------------------
struct FBlockSpec {
  std::string prefix;
  FKind kind;
  const void* x;
  int csx, in_f32, cin, cin_pad, H, W;
  void* out;
  int cso, out_f32, cout;
  bool proj, desc_branch;
};

static void add_fblock(fpc_ctx* c, const FBlockSpec& s, size_t* blob_off) {
  if (s.kind >= FK_COUNT) { c->plan_error = true; return; }
  const FKindInfo& k = g_fkinds[s.kind];
  Op op;
  op.type = OP_BF16;
  const std::string tag = k.planes == 3 ? " [3xbf16 " : k.planes == 2 ? " [2xfp16 " : " [bf16 ";
  op.name = s.prefix + tag + (s.proj ? "conv1+bn1+relu+conv2+bn2+proj+relu]" : "conv1+bn1+relu+conv2+bn2+identity+relu]");
  op.prefix = s.prefix;
  op.fkind = s.kind;
  op.cin = s.cin;
  op.cout = s.cout;
  op.descriptor_branch = s.desc_branch;
  BlockBfArgs& a = op.fargs;
  const int nbt = k.WN * k.NB, K16 = k.KC / 16;
  a.x = s.x;
  a.csx = s.csx;
  a.in_f32 = s.in_f32;
  a.x_bytes = (unsigned)std::min<size_t>((size_t)c->B * s.H * s.W * s.csx * 2, 0xffffff00u);
  a.nchunk = s.cin_pad / k.KC;
  // block_bf16_one_kernel: the single chunk is a compile-time fact of the instance (block_bf16.h)
  if (strstr(k.symbol, "block_bf16_one_kernel") && a.nchunk != 1) { c->plan_error = true; return; }
  if (strstr(k.symbol, "block_bf16_two_kernel") && a.nchunk != 2) { c->plan_error = true; return; }
  a.H = s.H;
  a.W = s.W;
  a.ntaps = 9;
  a.pad = 1;
  // (halo row pitch: block_bf16_kernel pads it for its bank-conflict-free pixel order, block_bf16.h)
  const int HWp = k.planes == 1 ? bf_halo_pitch(k.TH, k.TW, k.S, k.EXT) : (k.TW - 1) * k.S + k.EXT, ROW16 = k.KC / 8 + 1;
  for (int ky = 0; ky < 3; ++ky)
    for (int kx = 0; kx < 3; ++kx) a.tapoff16[ky * 3 + kx] = (ky * HWp + kx) * ROW16;
  a.k16_h = k.CMIDP / 16;
  a.k16_x = s.proj ? s.cin_pad / 16 : 0;
  a.conv_only = 0;
  a.out = s.out;
  a.cso = s.cso;
  a.out_f32 = s.out_f32;
  a.Ho = a.OH = s.H / k.S;
  a.Wo = a.OW = s.W / k.S;
  a.oys = a.oxs = 1;
  a.oy0 = a.ox0 = 0;
  a.tiles_x = (a.Wo + k.TW - 1) / k.TW;
  a.tiles_y = (a.Ho + k.TH - 1) / k.TH;
  fpc_ctx::ConvW cw;
  // FPC_BF16 (planes == 1): the shortcut's fragments (projection, or the identity as a unit matrix) follow each chunk's
  // conv1 fragments in the w1 stream as a tenth tap (block_bf16.h), and w2 holds conv2 only
  const bool sc_in_w1 = k.planes == 1;
  cw.w_off[0] = *blob_off;
  *blob_off += ((size_t)a.nchunk * (sc_in_w1 ? 10 : 9) * K16 + 2) * k.planes * nbt * 64 * 4;
  cw.b_off = *blob_off;
  *blob_off += (size_t)nbt * 32;
  cw.w_off[1] = *blob_off;
  *blob_off += ((size_t)(a.k16_h + (sc_in_w1 ? 0 : a.k16_x)) + 2) * k.planes * nbt * 64 * 4;
  cw.b2_off = *blob_off;
  *blob_off += (size_t)nbt * 32;
  op.flops_per_frame = 2.0 * a.Ho * a.Wo * s.cout * ((double)s.cin * 9 + s.cout + (s.proj ? s.cin : 0));
  // split operands: six bf16 MFMAs per product
  op.mfma_flops_per_frame = (k.planes == 3 ? 6.0 : k.planes == 2 ? 3.0 : 1.0) * 2.0 * a.tiles_x * a.tiles_y * (k.WM * k.MB * 32.0) * (nbt * 32.0) *
                            ((double)a.nchunk * k.KC * 9 + (a.k16_h + (sc_in_w1 ? s.cin_pad / 16 : a.k16_x)) * 16.0);
  op.bytes_per_frame = (double)s.cin * s.H * s.W * ((s.in_f32 || k.planes > 1) ? 4.0 : 2.0) +
                       (double)s.cout * a.Ho * a.Wo * ((s.out_f32 || k.planes > 1) ? 4.0 : 2.0);
  c->ops.push_back(op);
  c->convw.push_back(cw);
}

// ConvTranspose2d(k3, s2, p1, op1) + bn + relu as four output-parity phases, one launch each
static void add_fconvT(fpc_ctx* c, FKind kind, const void* x, int csx, int cin, int H, int W, void* out, int cso,
                       int cout, size_t* blob_off) {
  if (kind >= FK_COUNT) { c->plan_error = true; return; }
  const FKindInfo& k = g_fkinds[kind];
  const int nbt = k.WN * k.NB, K16 = k.KC / 16;
  if (strstr(k.symbol, "convt_bf16_kernel")) {
    // round 5: ONE launch (convt_bf16.h) -- a workgroup stages a tile's halo chunk once and issues all nine taps on it, the
    // four parities' accumulators side by side; gridDim.y = the output-channel parts
    if (cin % k.KC != 0 || cout != k.CMIDP) { c->plan_error = true; return; }
    Op op;
    op.type = OP_BF16;
    op.name = "descriptor.up_sample+bn+relu [bf16, four parities in one launch]";
    op.prefix = "descriptor.up_sample";
    op.fkind = kind;
    op.phase = 4;   // all four
    op.cin = cin;
    op.cout = cout;
    op.descriptor_branch = true;
    op.grid_y = k.CMIDP / (k.WN * 32);
    BlockBfArgs& a = op.fargs;
    a.x = x;
    a.csx = csx;
    a.in_f32 = 0;
    a.x_bytes = (unsigned)std::min<size_t>((size_t)c->B * H * W * csx * 2, 0xffffff00u);
    a.nchunk = cin / k.KC;
    a.H = H;
    a.W = W;
    a.pad = 0;
    a.ntaps = 9;
    a.conv_only = 1;
    a.out = out;
    a.cso = cso;
    a.out_f32 = 0;
    a.Ho = H;
    a.Wo = W;
    a.OH = 2 * H;
    a.OW = 2 * W;
    a.oys = a.oxs = 2;
    a.tiles_x = (W + k.TW - 1) / k.TW;
    a.tiles_y = (H + k.TH - 1) / k.TH;
    fpc_ctx::ConvW cw;
    cw.w_off[0] = *blob_off;
    *blob_off += ((size_t)(cin / 16) * 9 + 2) * (k.CMIDP / 32) * 64 * 4;   // pack_conv_bf16 with KC = 16: [step of 16 channels][tap][32-channel block][lane]
    cw.b_off = *blob_off;
    *blob_off += (size_t)k.CMIDP;
    op.flops_per_frame = 2.0 * 9 * H * W * cin * cout;
    op.mfma_flops_per_frame = 2.0 * 9 * a.tiles_x * a.tiles_y * (k.TH * k.TW) * (double)cin * k.CMIDP;
    op.bytes_per_frame = 2.0 * ((double)cin * H * W + 4.0 * cout * H * W);
    c->ops.push_back(op);
    c->convw.push_back(cw);
    return;
  }
  const int HWp = k.planes == 1 ? bf_halo_pitch(k.TH, k.TW, k.S, k.EXT) : (k.TW - 1) * k.S + k.EXT, ROW16 = k.KC / 8 + 1;
  for (int ph = 0; ph < 4; ++ph) {
    const int py = ph >> 1, px = ph & 1;
    Op op;
    op.type = OP_BF16;
    op.name = std::string("descriptor.up_sample+bn+relu [") + (k.planes == 3 ? "3xbf16" : k.planes == 2 ? "2xfp16" : "bf16") + " phase " + std::to_string(ph) + "]";
    op.prefix = "descriptor.up_sample";
    op.fkind = kind;
    op.phase = ph;
    op.cin = cin;
    op.cout = cout;
    op.descriptor_branch = true;
    BlockBfArgs& a = op.fargs;
    a.x = x;
    a.csx = csx;
    a.in_f32 = 0;
    a.x_bytes = (unsigned)std::min<size_t>((size_t)c->B * H * W * csx * 2, 0xffffff00u);
    a.nchunk = cin / k.KC;
    a.H = H;
    a.W = W;
    a.pad = 0;
    a.ntaps = 0;
    for (int iy = 0; iy < (py ? 2 : 1); ++iy)
      for (int ix = 0; ix < (px ? 2 : 1); ++ix) {
        const int dy = py ? 1 - iy : 0, dx = px ? 1 - ix : 0;
        a.tapoff16[a.ntaps++] = (dy * HWp + dx) * ROW16;
      }
    a.conv_only = 1;
    a.out = out;
    a.cso = cso;
    a.out_f32 = 0;
    a.Ho = H;
    a.Wo = W;
    a.OH = 2 * H;
    a.OW = 2 * W;
    a.oys = a.oxs = 2;
    a.oy0 = py;
    a.ox0 = px;
    a.tiles_x = (W + k.TW - 1) / k.TW;
    a.tiles_y = (H + k.TH - 1) / k.TH;
    fpc_ctx::ConvW cw;
    cw.w_off[0] = *blob_off;
    *blob_off += ((size_t)a.nchunk * a.ntaps * K16 + 2) * k.planes * nbt * 64 * 4;
    cw.b_off = *blob_off;
    *blob_off += (size_t)nbt * 32;
    op.flops_per_frame = 2.0 * a.ntaps * H * W * cin * cout;
    op.mfma_flops_per_frame = (k.planes == 3 ? 6.0 : k.planes == 2 ? 3.0 : 1.0) * 2.0 * a.ntaps * a.tiles_x * a.tiles_y * (k.WM * k.MB * 32.0) * (a.nchunk * k.KC) * (nbt * 32.0);
    // the four phases read the same input: a quarter of it is attributed to each, plus the phase's quarter of the output
    op.bytes_per_frame = (k.planes > 1 ? 4.0 : 2.0) * ((double)cin * H * W / 4.0 + (double)cout * H * W);
    c->ops.push_back(op);
    c->convw.push_back(cw);
  }
}

static void build_bf16_ops(fpc_ctx* c, size_t* bo) {
  const int H = c->H, W = c->W;
  const int H4 = H / 4, W4 = W / 4, Hc = H / 8, Wc = W / 8, H16 = H / 16, W16 = W / 16;
  const bool de = c->cfg.descriptor_enabled != 0;
  bf16_t* feat = reinterpret_cast<bf16_t*>(c->cat) + 128;
  // (layer1 measured the same 0.53 ms per 64 HD frames on 8 x 16 tiles of 2 x 2 waves x (2 x 1) blocks, 16 x 16 tiles of
  // 4 x 1 waves x (2 x 2) and 8 x 16 tiles of 2 x 1 waves x (2 x 2): not a matter of the blocking;
  // single-wave workgroups (4 x 16 tiles, 1 x 1 waves x (2 x 2) blocks: no barrier waits at all) measured 0.55 ms: not
  // the barriers either.  What these kernels run into is operand delivery: at full MFMA rate a CU's four SIMDs would
  // pull 64 B/clk of weight fragments from L2 and 64 B/clk of pixels from LDS for a 2 x 2 register blocking.)
  add_fblock(c, {"encoder.layer1.0", FK_F816_s1_K64_C64, c->x0, 64, 0, 64, 64, H4, W4, c->x1, 64, 0, 64, true, false}, bo);
  add_fblock(c, {"encoder.layer1.1", FK_F816_s1_K64_C64, c->x1, 64, 0, 64, 64, H4, W4, c->x2, 64, 0, 64, false, false}, bo);
  add_fblock(c, {"encoder.layer2.0", FK_F620_s2_K64_C128, c->x2, 64, 0, 64, 64, H4, W4, c->x3, 128, 0, 128, true, false}, bo);
  add_fblock(c, {"encoder.layer2.1", FK_F816_s1_K128_C128, c->x3, 128, 0, 128, 128, Hc, Wc, feat, 256, 0, 128, false, false}, bo);
  // detector.layer.0 on the 2 x 2 blocking of the 128-wide layers (N = 128 for 65 channels: a quarter of the MFMAs on
  // zeros, but four MFMAs per four operand fetches instead of three per four: 0.47 -> 0.40 ms per 64 HD frames);
  // layer.1 (K = 80) measured the same on both shapes and keeps the narrower one
  add_fblock(c, {"detector.layer.0", FK_F816_s1_K128_C80w, feat, 256, 0, 128, 128, Hc, Wc, c->d0, 80, 0, 65, true, false}, bo);
  add_fblock(c, {"detector.layer.1", FK_F816_s1_K80_C80, c->d0, 80, 0, 65, 80, Hc, Wc, c->lg, 80, 1, 65, false, false}, bo);
  c->ops.back().fused_softmax_capable = c->fuse_softmax;
  {
    Op op;
    op.type = OP_SOFTMAX;
    op.name = "exp-softmax+depth_to_space+threshold";
    c->ops.push_back(op);
    c->convw.push_back({});
  }
  if (de) {
    add_fblock(c, {"descriptor.layer_in.0", FK_F320_s2_K128_C256, feat, 256, 0, 128, 128, Hc, Wc, c->y16a, 256, 0, 256, true, true}, bo);
    add_fblock(c, {"descriptor.layer_in.1", FK_F416_s1_K128_C256, c->y16a, 256, 0, 256, 256, H16, W16, c->y16b, 256, 0, 256, false, true}, bo);
    add_fconvT(c, c->convt_fused ? FK_F816_ctf_K64_C128 : FK_F816_ct_K64_C128, c->y16b, 256, 256, H16, W16, c->cat, 256, 128, bo);
    add_fblock(c, {"descriptor.layer_out.0", FK_F816_s1_K64_C128, c->cat, 256, 0, 256, 256, Hc, Wc, c->lo0, 128, 0, 128, true, true}, bo);
    // (the descriptor map is bf16 as well -- round 3: descriptor16_kernel<8, true> reads it, fpc_forward / the tap convert it)
    add_fblock(c, {"descriptor.layer_out.1", FK_F816_s1_K128_C128, c->lo0, 128, 0, 128, 128, Hc, Wc, c->desc_map, 128, 0, 128, false, true}, bo);
  }
}

// dtype = FPC_F32_SPLIT: the fp32 plan's buffers (all fp32), every ResNetBlock / ConvTranspose on block_x3_kernel
// the fp16 twin of a bf16x3 instance (rows H* follow rows S* in the same order in FPC_BF16_KINDS)
static FKind split_kind(const fpc_ctx* c, FKind s);

static void build_x3_ops(fpc_ctx* c, size_t* bo) {
  const int H = c->H, W = c->W;
  const int H4 = H / 4, W4 = W / 4, Hc = H / 8, Wc = W / 8, H16 = H / 16, W16 = W / 16;
  const bool de = c->cfg.descriptor_enabled != 0;
  float* feat = c->cat + 128;
  add_fblock(c, {"encoder.layer1.0", split_kind(c, FK_S816_s1_K64_C64), c->x0, 64, 1, 64, 64, H4, W4, c->x1, 64, 1, 64, true, false}, bo);
  add_fblock(c, {"encoder.layer1.1", split_kind(c, FK_S816_s1_K64_C64), c->x1, 64, 1, 64, 64, H4, W4, c->x2, 64, 1, 64, false, false}, bo);
  add_fblock(c, {"encoder.layer2.0", split_kind(c, FK_S620_s2_K16_C128), c->x2, 64, 1, 64, 64, H4, W4, c->x3, 128, 1, 128, true, false}, bo);
  add_fblock(c, {"encoder.layer2.1", split_kind(c, FK_S620_s1_K64_C128), c->x3, 128, 1, 128, 128, Hc, Wc, feat, 256, 1, 128, false, false}, bo);
  add_fblock(c, {"detector.layer.0", split_kind(c, FK_S620_s1_K64_C80), feat, 256, 1, 128, 128, Hc, Wc, c->d0, 80, 1, 65, true, false}, bo);
  add_fblock(c, {"detector.layer.1", split_kind(c, FK_S620_s1_K16_C80), c->d0, 80, 1, 65, 80, Hc, Wc, c->lg, 80, 1, 65, false, false}, bo);
  {
    Op op;
    op.type = OP_SOFTMAX;
    op.name = "exp-softmax+depth_to_space+threshold";
    c->ops.push_back(op);
    c->convw.push_back({});
  }
  if (de) {
    add_fblock(c, {"descriptor.layer_in.0", split_kind(c, FK_S320_s2_K16_C256), feat, 256, 1, 128, 128, Hc, Wc, c->y16a, 256, 1, 256, true, true}, bo);
    add_fblock(c, {"descriptor.layer_in.1", split_kind(c, FK_S320_s1_K64_C256), c->y16a, 256, 1, 256, 256, H16, W16, c->y16b, 256, 1, 256, false, true}, bo);
    add_fconvT(c, split_kind(c, FK_S620_ct_K64_C128), c->y16b, 256, 256, H16, W16, c->cat, 256, 128, bo);
    add_fblock(c, {"descriptor.layer_out.0", split_kind(c, FK_S620_s1_K64_C128), c->cat, 256, 1, 256, 256, Hc, Wc, c->lo0, 128, 1, 128, true, true}, bo);
    add_fblock(c, {"descriptor.layer_out.1", split_kind(c, FK_S620_s1_K64_C128), c->lo0, 128, 1, 128, 128, Hc, Wc, c->desc_map, 128, 1, 128, false, true}, bo);
  }
}

// ---- the reference's C++ network (superpoint::SPModel, cpp/src/model.cc) -----------------------------
// A plain Conv2d + bias (+ ReLU) on the split-operand kernel (conv_only) -- dtype FPC_F32_SPLIT
static void add_fconv(fpc_ctx* c, FKind kind, const std::string& prefix, const float* x, int csx, int cin, int cin_pad,
                      int H, int W, float* out, int cso, int cout, int ksize, bool relu, bool desc_branch,
                      size_t* blob_off) {
  if (kind >= FK_COUNT) { c->plan_error = true; return; }
  const FKindInfo& k = g_fkinds[kind];
  const int nbt = k.WN * k.NB, K16 = k.KC / 16;
  const int HWp = k.planes == 1 ? bf_halo_pitch(k.TH, k.TW, k.S, k.EXT) : (k.TW - 1) * k.S + k.EXT, ROW16 = k.KC / 8 + 1;
  Op op;
  op.type = OP_BF16;
  op.name = prefix + (k.planes == 2 ? " [2xfp16 conv+bias" : " [3xbf16 conv+bias") + (relu ? "+relu]" : "]");
  op.prefix = prefix;
  op.fkind = kind;
  op.plain_conv = true;
  op.ksize = ksize;
  op.cin = cin;
  op.cout = cout;
  op.descriptor_branch = desc_branch;
  BlockBfArgs& a = op.fargs;
  a.x = x;
  a.csx = csx;
  a.in_f32 = 1;
  a.nchunk = cin_pad / k.KC;
  a.H = H;
  a.W = W;
  a.pad = ksize / 2;
  a.ntaps = ksize * ksize;
  for (int ky = 0; ky < ksize; ++ky)
    for (int kx = 0; kx < ksize; ++kx) a.tapoff16[ky * ksize + kx] = (ky * HWp + kx) * ROW16;
  a.conv_only = 1;
  a.norelu = relu ? 0 : 1;
  a.out = out;
  a.cso = cso;
  a.out_f32 = 1;
  a.Ho = a.OH = H;
  a.Wo = a.OW = W;
  a.oys = a.oxs = 1;
  a.tiles_x = (W + k.TW - 1) / k.TW;
  a.tiles_y = (H + k.TH - 1) / k.TH;
  fpc_ctx::ConvW cw;
  cw.w_off[0] = *blob_off;
  *blob_off += ((size_t)a.nchunk * a.ntaps * K16 + 2) * k.planes * nbt * 64 * 4;
  cw.b_off = *blob_off;
  *blob_off += (size_t)nbt * 32;
  op.flops_per_frame = 2.0 * a.ntaps * H * W * (double)cin * cout;
  op.mfma_flops_per_frame = (k.planes == 2 ? 3.0 : 6.0) * 2.0 * a.ntaps * a.tiles_x * a.tiles_y * (k.WM * k.MB * 32.0) * (a.nchunk * k.KC) * (nbt * 32.0);
  op.bytes_per_frame = 4.0 * ((double)cin * H * W + (double)cout * H * W);
  c->ops.push_back(op);
  c->convw.push_back(cw);
}

static int build_vgg_plan(fpc_ctx* c) {
  const int H = c->H, W = c->W, B = c->B, Hc = H / 8, Wc = W / 8;
  const size_t npix8 = (size_t)B * Hc * Wc;
  Carver cv;
  cv.zones = &c->guards;
  cv.guard = c->guard_zones ? GUARD_BYTES : 0;
  // activations: one buffer per tensor (sub-batches of one call run different layers at the same time)
  const int ah[4] = {H, H / 2, H / 4, Hc}, aw[4] = {W, W / 2, W / 4, Wc}, ac[4] = {64, 64, 128, 128};
  size_t o_a[4], o_b[4], o_p[3];
  for (int i = 0; i < 4; ++i) {
    o_a[i] = cv.take<float>((size_t)B * ah[i] * aw[i] * ac[i]);
    o_b[i] = cv.take<float>((size_t)B * ah[i] * aw[i] * ac[i]);
    if (i < 3) o_p[i] = cv.take<float>((size_t)B * ah[i + 1] * aw[i + 1] * ac[i]);
  }
  const size_t o_pa = cv.take<float>(npix8 * 256), o_da = cv.take<float>(npix8 * 256);
  const size_t o_lg = cv.take<float>(npix8 * 80), o_desc = cv.take<float>(npix8 * 256), o_descin = cv.take<float>(npix8 * 256);
  const size_t o_prob = cv.take<float>((size_t)B * H * W);
  const size_t o_map = cv.take<uint32_t>((size_t)B * H * W), o_cand = cv.take<uint32_t>((size_t)B * H * W);
  const size_t o_ncand = cv.take<int32_t>(B), o_count = cv.take<int32_t>(B), o_status = cv.take<int32_t>(4 + BLOB_HEADER_FLOATS);   // + the 64-byte tag of fpc_broadcast_weights
  const size_t o_range = cv.take<uint32_t>((size_t)B * FPC_RANGE_WORDS);   // per-frame range words (kernels_misc.h)
  const size_t o_rowmax = cv.take<float>((size_t)B * (H / 8));             // largest logit per row of cells (softmax_d2s_kernel)
  const size_t o_xy = cv.take<int32_t>((size_t)B * c->cap * 2), o_conf = cv.take<float>((size_t)B * c->cap);
  const size_t o_dout = cv.take<float>((size_t)B * c->cap * 256);
  const size_t o_sort = cv.take<unsigned long long>((size_t)B * c->sort_cap);
  const size_t o_aux = cv.take<int32_t>((size_t)B * NMS_AUX_INTS);
  const size_t o_rowbest = cv.take<unsigned long long>(c->cap), o_colbest = cv.take<unsigned long long>(c->cap);
  cv.finish();
  c->slab_bytes = cv.off;
  if (hipMalloc((void**)&c->slab, c->slab_bytes) != hipSuccess) {
    g_hip_err = "hipMalloc(workspace " + std::to_string(c->slab_bytes >> 20) + " MiB) failed";
    return FPC_E_HIP;
  }
  HIPCHECK(hipMemset(c->slab, 0, c->slab_bytes));
  if (int grc = fill_guards(c)) return grc;
  auto F = [&](size_t o) { return reinterpret_cast<float*>(c->slab + o); };
  c->lg = F(o_lg); c->desc_map = F(o_desc); c->desc_in_nhwc = F(o_descin); c->prob = F(o_prob);
  c->nmsmap = reinterpret_cast<uint32_t*>(c->slab + o_map);
  c->cand = reinterpret_cast<uint32_t*>(c->slab + o_cand);
  c->ncand = reinterpret_cast<int32_t*>(c->slab + o_ncand);
  c->count = reinterpret_cast<int32_t*>(c->slab + o_count);
  c->status = reinterpret_cast<int32_t*>(c->slab + o_status);
  c->bcast_tag = reinterpret_cast<uint32_t*>(c->status + 4);
  c->range = reinterpret_cast<uint32_t*>(c->slab + o_range);
  c->rowmax = reinterpret_cast<float*>(c->slab + o_rowmax);
  c->xy = reinterpret_cast<int32_t*>(c->slab + o_xy);
  c->conf = F(o_conf);
  c->desc_out = F(o_dout);
  c->sort_scratch = reinterpret_cast<unsigned long long*>(c->slab + o_sort);
  c->nms_aux = reinterpret_cast<int32_t*>(c->slab + o_aux);
  c->rowbest = reinterpret_cast<unsigned long long*>(c->slab + o_rowbest);
  c->colbest = reinterpret_cast<unsigned long long*>(c->slab + o_colbest);

  size_t bo = BLOB_HEADER_FLOATS;
  c->ops.clear();
  c->convw.clear();
  auto conv = [&](const std::string& prefix, const float* x, int cin, int Hx, int Wx, float* out, int cso, int cout,
                  int ksize, bool relu, bool desc) {
    if (c->split) {
      FKind fk;
      if (ksize == 3) fk = cout == 64 ? split_kind(c, FK_S816_s1_K64_C64) : cout == 128 ? split_kind(c, FK_S620_s1_K64_C128) : split_kind(c, FK_S320_s1_K64_C256);
      else fk = cout == 65 ? split_kind(c, FK_S620_1x1_K64_C80) : split_kind(c, FK_S320_1x1_K64_C256);
      add_fconv(c, fk, prefix, x, cin, cin, cin, Hx, Wx, out, cso, cout, ksize, relu, desc, &bo);
      return;
    }
    if (ksize == 3 && relu && c->winograd) {  // Winograd F(2x2,3x3), conv-only; 256 outputs = two 128-channel launches
      const WKind wk2 = c->winograd_gen >= 2 ? (cout == 64 ? WK_W16_C64 : WK_W16_C128) : (cout == 64 ? WK_W816_K32_C64 : WK_W816_K32_C128);
      auto add = [&](WKind wk, int when) {
        const size_t i0 = c->ops.size();
        for (int n0 = 0; n0 < cout; n0 += g_wkinds[wk].CMID)
          add_wconv(c, prefix, false, wk, x, cin, cin, Hx, Wx, out, cso, cout, n0, desc, &bo);
        for (size_t k = i0; k < c->ops.size(); ++k) c->ops[k].when = when;
      };
      if (c->winograd_gen == 3 && w36_fits(c->B, Hx, Wx, cin, cso) && cin % 32 == 0 && cin >= 64) {
        // round 4: F(4x4,3x3) for the C++ network's 3x3 layers too (the conv-only form of wblock36_kernel; 256 outputs in
        // parts), with generation 2 -- fragments of its own -- for calls of a few frames, as the Python network's blocks
        // (priced for a 32-frame call whatever max_batch is: the packed layout must not depend on it -- contexts of
        // different max_batch exchange blobs)
        const int part = w36_conv_part(cout, ((Hx + 15) / 16) * ((Wx + 15) / 16) * 32, 256);
        add(w36_kind(part, Hx, Wx, c->w36_paired), c->latency_tiles ? 2 : 0);
        if (c->latency_tiles) add(wk2, 1);
      } else {
        add(wk2, 0);
      }
      return;
    }
    ConvSpec s{};
    s.name = prefix + (relu ? " [conv+bias+relu]" : " [conv+bias]");
    if (ksize == 3) s.kind = cout == 64 ? K_T816_3x3_K64_N64 : K_T620_3x3_K64_N128;
    else s.kind = cout == 65 ? K_T620_1x1_K64_N96 : K_T620_1x1_K64_N128;
    s.ksize = ksize; s.stride = 1;
    s.in0 = x; s.cs0 = cin; s.cin0 = cin; s.cin0_pad = cin; s.H0 = Hx; s.W0 = Wx;
    s.out = out; s.cso = cso; s.cout = cout; s.nstore = cout == 65 ? cso : cout; s.Ho = Hx; s.Wo = Wx; s.relu = relu ? 1 : 0;
    s.desc_branch = desc;
    add_conv(c, s, &bo);
    c->ops.back().prefix = prefix;
    c->ops.back().plain_conv = true;
    c->ops.back().ksize = ksize;
    c->ops.back().cin = cin;
    c->ops.back().cout = cout;
  };
  const float* x = nullptr;
  for (int i = 0; i < 4; ++i) {
    const std::string pa = "encoder_conv" + std::to_string(i) + "_a", pb = "encoder_conv" + std::to_string(i) + "_b";
    float* a = F(o_a[i]);
    float* b = F(o_b[i]);
    if (i == 0) {  // Conv2d(1, 64): K = 9, plain FMAs
      Op op;
      op.type = OP_VCONV0;
      op.name = pa + " [conv+bias+relu, 1 input channel]";
      op.prefix = pa;
      op.pout = a;
      op.pH = H; op.pW = W;
      op.flops_per_frame = 2.0 * H * W * 64 * 9;
      op.bytes_per_frame = 4.0 * ((double)H * W + 64.0 * H * W);
      c->ops.push_back(op);
      c->convw.push_back({});
      c->vconv0_off = bo;
      bo += 640;
    } else {
      conv(pa, x, ac[i - 1], ah[i], aw[i], a, ac[i], ac[i], 3, true, false);
    }
    conv(pb, a, ac[i], ah[i], aw[i], b, ac[i], ac[i], 3, true, false);
    if (i < 3) {
      Op op;
      op.type = OP_POOL2;
      op.name = "max_pool2d(2,2) after " + pb;
      op.pin = b;
      op.pout = F(o_p[i]);
      op.pH = ah[i]; op.pW = aw[i]; op.pC = ac[i];
      op.bytes_per_frame = 4.0 * 1.25 * ac[i] * ah[i] * aw[i];
      c->ops.push_back(op);
      c->convw.push_back({});
      x = op.pout;
    } else {
      x = b;
    }
  }
  const int lgcs = c->lgcs;
  conv("detector_conv_a", x, 128, Hc, Wc, F(o_pa), 256, 256, 3, true, false);
  conv("detector_conv_b", F(o_pa), 256, Hc, Wc, c->lg, lgcs, 65, 1, false, false);
  {
    Op op;
    op.type = OP_SOFTMAX;
    op.name = "exp-softmax+depth_to_space+threshold";
    c->ops.push_back(op);
    c->convw.push_back({});
  }
  const bool de = c->cfg.descriptor_enabled != 0;
  if (de) {
    conv("descriptor_conv_a", x, 128, Hc, Wc, F(o_da), 256, 256, 3, true, true);
    conv("descriptor_conv_b", F(o_da), 256, Hc, Wc, c->desc_map, 256, 256, 1, false, true);
    Op op;
    op.type = OP_L2NORM;
    op.name = "descriptor L2 normalisation over channels";
    op.descriptor_branch = true;
    op.pout = c->desc_map;
    op.bytes_per_frame = 4.0 * 2 * 256.0 * Hc * Wc;
    c->ops.push_back(op);
    c->convw.push_back({});
  }
  {
    Op op;
    op.type = OP_NMS;
    op.name = "nms+sort+border_crop";
    c->ops.push_back(op);
    c->convw.push_back({});
    if (de) {
      op.type = OP_DESC;
      op.name = "descriptor_sample+l2norm";
      c->ops.push_back(op);
      c->convw.push_back({});
    }
  }
  for (Op& op : c->ops)
    if (op.type == OP_SOFTMAX) op.bytes_per_frame = 4.0 * (65.0 * Hc * Wc + 2.0 * H * W);
  c->blob_floats = bo;
  HIPCHECK(hipMalloc((void**)&c->blob, c->blob_floats * sizeof(float)));
  HIPCHECK(hipMemset(c->blob, 0, c->blob_floats * sizeof(float)));
  for (size_t i = 0; i < c->ops.size(); ++i) {
    Op& op = c->ops[i];
    if (op.type == OP_BF16) {
      op.fargs.w1 = reinterpret_cast<const uint4*>(c->blob + c->convw[i].w_off[0]);
      op.fargs.b1 = c->blob + c->convw[i].b_off;
    }
    if (op.type == OP_CONV) {
      op.args.sub[0].wfrag = reinterpret_cast<const float4*>(c->blob + c->convw[i].w_off[0]);
      op.args.bias = c->blob + c->convw[i].b_off;
    }
    if (op.type == OP_WBLOCK) {
      op.wargs.w1 = reinterpret_cast<const float4*>(c->blob + c->convw[i].w_off[0]);
      op.wargs.b1 = c->blob + c->convw[i].b_off;
    }
  }
  return FPC_OK;
}

// Rows H* of FPC_BF16_KINDS must mirror rows S* one for one.  (Round 1, gdb.log: with the two H*_1x1 rows still
// missing, the C++ network's 1x1 layers in FPC_F32_SPLIT_F16 mode indexed past the end of g_fkinds, read TW = KC = 0 and
// build_vgg_plan died with SIGFPE in `(W + TW - 1) / TW`.)
static_assert((int)FK_COUNT - (int)FK_H816_s1_K64_C64 == (int)FK_H816_s1_K64_C64 - (int)FK_S816_s1_K64_C64,
              "every S* (bf16x3) instance needs its H* (fp16x2) twin, in the same order");
static FKind split_kind(const fpc_ctx* c, FKind s) {
  if (!c->split_f16) return s;
  const int h = (int)s + ((int)FK_H816_s1_K64_C64 - (int)FK_S816_s1_K64_C64);
  return h >= (int)FK_H816_s1_K64_C64 && h < (int)FK_COUNT ? (FKind)h : FK_COUNT;
}

static int build_plan(fpc_ctx* c) {
  if (c->vgg) return build_vgg_plan(c);
  const int H = c->H, W = c->W, B = c->B;
  const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4, Hc = H / 8, Wc = W / 8, H16 = H / 16, W16 = W / 16;
  const bool de = c->cfg.descriptor_enabled != 0;
  // ---- carve the slab
  Carver cv;
  cv.zones = &c->guards;
  cv.guard = c->guard_zones ? GUARD_BYTES : 0;
  const size_t o_stem = cv.take<float>((size_t)B * H2 * W2 * 64);
  const size_t o_x0 = cv.take<float>((size_t)B * H4 * W4 * 64), o_h4 = cv.take<float>((size_t)B * H4 * W4 * 64);
  const size_t o_x1 = cv.take<float>((size_t)B * H4 * W4 * 64), o_x2 = cv.take<float>((size_t)B * H4 * W4 * 64);
  const size_t npix8 = (size_t)B * Hc * Wc, npix16 = (size_t)B * H16 * W16;
  const size_t o_h8 = cv.take<float>(npix8 * 128), o_x3 = cv.take<float>(npix8 * 128);
  const size_t o_cat = cv.take<float>(npix8 * 256);
  const size_t o_dh = cv.take<float>(npix8 * 72), o_dproj = cv.take<float>(npix8 * 72);
  const size_t o_d0 = cv.take<float>(npix8 * 80), o_lg = cv.take<float>(npix8 * 80);
  const size_t o_h16 = cv.take<float>(npix16 * 256), o_y16a = cv.take<float>(npix16 * 256);
  const size_t o_y16b = cv.take<float>(npix16 * 256);
  const size_t o_loh = cv.take<float>(npix8 * 128), o_lo0 = cv.take<float>(npix8 * 128);
  const size_t o_desc = cv.take<float>(npix8 * 128), o_descin = cv.take<float>(npix8 * 128);
  const size_t o_prob = cv.take<float>((size_t)B * H * W);
  const size_t o_map = cv.take<uint32_t>((size_t)B * H * W), o_cand = cv.take<uint32_t>((size_t)B * H * W);
  const size_t o_ncand = cv.take<int32_t>(B), o_count = cv.take<int32_t>(B), o_status = cv.take<int32_t>(4 + BLOB_HEADER_FLOATS);   // + the 64-byte tag of fpc_broadcast_weights
  const size_t o_range = cv.take<uint32_t>((size_t)B * FPC_RANGE_WORDS);   // per-frame range words (kernels_misc.h)
  const size_t o_rowmax = cv.take<float>((size_t)B * (H / 8));             // largest logit per row of cells (softmax_d2s_kernel)
  const size_t o_xy = cv.take<int32_t>((size_t)B * c->cap * 2), o_conf = cv.take<float>((size_t)B * c->cap);
  const size_t o_dout = cv.take<float>(de ? (size_t)B * c->cap * 128 : 64);
  const size_t o_sort = cv.take<unsigned long long>((size_t)B * c->sort_cap);
  const size_t o_aux = cv.take<int32_t>((size_t)B * NMS_AUX_INTS);
  const size_t o_rowbest = cv.take<unsigned long long>(c->cap), o_colbest = cv.take<unsigned long long>(c->cap);
  cv.finish();
  c->slab_bytes = cv.off;
  if (hipMalloc((void**)&c->slab, c->slab_bytes) != hipSuccess) {
    g_hip_err = "hipMalloc(workspace " + std::to_string(c->slab_bytes >> 20) + " MiB) failed";
    return FPC_E_HIP;
  }
  HIPCHECK(hipMemset(c->slab, 0, c->slab_bytes));
  if (int grc = fill_guards(c)) return grc;
  auto F = [&](size_t o) { return reinterpret_cast<float*>(c->slab + o); };
  c->stem_out = F(o_stem); c->x0 = F(o_x0); c->h4 = F(o_h4); c->x1 = F(o_x1); c->x2 = F(o_x2);
  c->h8 = F(o_h8); c->x3 = F(o_x3); c->cat = F(o_cat); c->dh = F(o_dh); c->dproj = F(o_dproj);
  c->d0 = F(o_d0); c->lg = F(o_lg); c->h16 = F(o_h16); c->y16a = F(o_y16a); c->y16b = F(o_y16b);
  c->lo_h = F(o_loh); c->lo0 = F(o_lo0); c->desc_map = F(o_desc); c->desc_in_nhwc = F(o_descin);
  c->prob = F(o_prob);
  c->nmsmap = reinterpret_cast<uint32_t*>(c->slab + o_map);
  c->cand = reinterpret_cast<uint32_t*>(c->slab + o_cand);
  c->ncand = reinterpret_cast<int32_t*>(c->slab + o_ncand);
  c->count = reinterpret_cast<int32_t*>(c->slab + o_count);
  c->status = reinterpret_cast<int32_t*>(c->slab + o_status);
  c->bcast_tag = reinterpret_cast<uint32_t*>(c->status + 4);
  c->range = reinterpret_cast<uint32_t*>(c->slab + o_range);
  c->rowmax = reinterpret_cast<float*>(c->slab + o_rowmax);
  c->xy = reinterpret_cast<int32_t*>(c->slab + o_xy);
  c->conf = F(o_conf);
  c->desc_out = F(o_dout);
  c->sort_scratch = reinterpret_cast<unsigned long long*>(c->slab + o_sort);
  c->nms_aux = reinterpret_cast<int32_t*>(c->slab + o_aux);
  c->rowbest = reinterpret_cast<unsigned long long*>(c->slab + o_rowbest);
  c->colbest = reinterpret_cast<unsigned long long*>(c->slab + o_colbest);

  // ---- ops
  size_t bo = BLOB_HEADER_FLOATS;  // blob offset in floats (the tag of the packed format comes first)
  c->ops.clear();
  c->convw.clear();
  {
    Op op;
    op.type = OP_STEM;
    op.name = "encoder.conv1+bn1+relu";
    op.flops_per_frame = 2.0 * H2 * W2 * 64 * 147;  // of the reference's 3-channel convolution, also for gray frames
    op.mfma_flops_per_frame = 2.0 * ((H2 + 15) / 16) * ((W2 + 15) / 16) * 256.0 * 64 * 152;
    op.bytes_per_frame = 4.0 * (double)c->cin * H * W + (c->bf16 ? 2.0 : 4.0) * 64.0 * H4 * W4;   // frame in, pooled map out
    if (c->split || c->bf16)  // stem_pool_x3_kernel: (rows + 1) / 2 K16 steps of six bf16 MFMAs
      op.mfma_flops_per_frame = (c->bf16 ? 1 : c->split_f16 ? 3 : 6) * 2.0 * ((H2 + 15) / 16) * ((W2 + 15) / 16) * 256.0 * 64 * 16.0 * ((c->cin * 7 + 1) / 2);
    if (c->bf16)              // stem_bf16_kernel: 8 x 7 pooled pixels per tile, 8 blocks of 32 pixels x 64 channels x K16 steps
      op.mfma_flops_per_frame = 2.0 * ((H4 + SB2_PH - 1) / SB2_PH) * ((W4 + SB2_PW - 1) / SB2_PW) * 256.0 * 64 * 16.0 * ((c->cin * 7 + 1) / 2);
    c->ops.push_back(op);
    c->convw.push_back({});
    c->stem_w_off = bo;
    // + two zero groups: the fused kernel prefetches ahead; the split-operand stem needs 13 steps x 3 planes
    bo += std::max<size_t>((size_t)(STEM_KG + 2) * 2 * 64 * 4, (size_t)13 * 3 * 2 * 64 * 4);
    c->stem_b_off = bo;
    bo += 64;
    op = Op();
    op.type = OP_POOL;
    op.name = "encoder.max_pool";
    c->ops.push_back(op);
    c->convw.push_back({});
  }
  auto block = [&](const std::string& p, Kind k3, Kind k1, int stride, const float* x, int csx, int cin, int cinp,
                   int Hx, int Wx, float* h, int csh, int cout, int coutp, float* y, int csy, bool proj,
                   bool desc, BKind bk = BK_COUNT) {
    if (c->fuse_blocks && bk != BK_COUNT) {
      const BlockSpec bs{p, bk, x, csx, cin, cinp, Hx, Wx, y, csy, cout, coutp, proj, desc};
      WKind wk = WK_COUNT;
      if (c->winograd && stride == 1) {
        // generation 3's projection pass walks x in 16-channel steps, 4 or 8 per pass (wblock36_mfma.h: gemm_over advances
        // by 4): k8_x = cinp / 8 must be a multiple of 8, i.e. cinp of 64 -- true of every layer of both networks; a
        // layer that is not falls back to generation 2 instead of running extra K steps over stale staging rows
        const int gen = c->winograd_gen == 3 && (!w36_fits(c->B, Hx, Wx, csx, csy) || (proj && cinp % 64 != 0)) ? 2 : c->winograd_gen;
        if (cinp % 32 == 0 && cout == 64) wk = gen == 3 ? w36_kind(64, Hx, Wx, c->w36_paired) : gen == 2 ? WK_W16_C64 : WK_W816_K32_C64;
        else if (cinp % 32 == 0 && cout == 128) wk = gen == 3 ? w36_kind(128, Hx, Wx) : gen == 2 ? WK_W16_C128 : WK_W816_K32_C128;
        else if (c->winograd_det && cout == 65 && c->winograd_det_gen3 && c->winograd_gen == 3 && w36_fits(c->B, Hx, Wx, csx, csy) &&
                 (cin == 65 ? !proj : (cin == cinp && cinp % 64 == 0 && cinp <= 256)))
          wk = w36_kind(65, Hx, Wx, c->w36_paired);
        else if (c->winograd_det && cinp % 32 == 0 && coutp == 72) wk = WK_W816_K32_C72;
        else if (c->winograd_det && cinp == 72 && coutp == 72) wk = WK_W816_K24_C72;
      }
      if (wk != WK_COUNT)
        add_wblock(c, bs, wk, &bo);
      else
        add_block(c, bs, &bo);
      return;
    }
    const int Ho = Hx / stride, Wo = Wx / stride;
    ConvSpec s{};
    s.name = p + ".conv1+bn1+relu";
    s.kind = k3; s.ksize = 3; s.stride = stride;
    s.in0 = x; s.cs0 = csx; s.cin0 = cin; s.cin0_pad = cinp; s.H0 = Hx; s.W0 = Wx;
    s.out = h; s.cso = csh; s.cout = cout; s.nstore = coutp; s.Ho = Ho; s.Wo = Wo; s.relu = 1;
    s.desc_branch = desc;
    add_conv(c, s, &bo);
    ConvSpec t{};
    t.name = p + (proj ? ".conv2+bn2+proj+relu" : ".conv2+bn2+identity+relu");
    t.kind = k1; t.ksize = 1; t.stride = 1;
    t.in0 = h; t.cs0 = csh; t.cin0 = cout; t.cin0_pad = coutp; t.H0 = Ho; t.W0 = Wo;
    if (proj) {
      t.in1 = x; t.cs1 = csx; t.cin1 = cin; t.cin1_pad = cinp; t.s1 = stride; t.H1 = Hx; t.W1 = Wx;
    } else {
      t.res = x; t.csr = csx;
    }
    t.out = y; t.cso = csy; t.cout = cout; t.nstore = coutp; t.Ho = Ho; t.Wo = Wo; t.relu = 1;
    t.desc_branch = desc;
    add_conv(c, t, &bo);
  };
  float* feat = c->cat + 128;  // encoder output lives in channels 128..255 of `cat`
  if (c->bf16 || c->split) {
    if (c->bf16) build_bf16_ops(c, &bo);
    else build_x3_ops(c, &bo);
    goto postproc;
  }
  {
  const BKind l1kind = c->layer1_t816 ? BK_B816_s1_K64_C64 : BK_B1616_s1_K32_C64;
  block("encoder.layer1.0", K_T816_3x3_K64_N64, K_T816_1x1_K64_N64, 1, c->x0, 64, 64, 64, H4, W4, c->h4, 64, 64, 64,
        c->x1, 64, true, false, l1kind);
  block("encoder.layer1.1", K_T816_3x3_K64_N64, K_T816_1x1_K64_N64, 1, c->x1, 64, 64, 64, H4, W4, c->h4, 64, 64, 64,
        c->x2, 64, false, false, l1kind);
  block("encoder.layer2.0", K_T620_3x3s2_K32_N128, K_T620_1x1_K64_N128, 2, c->x2, 64, 64, 64, H4, W4, c->h8, 128,
        128, 128, c->x3, 128, true, false, BK_B620_s2_K32_C128);
  if (c->fuse_blocks) retile_last(c, BK_B320_s2_K32_C128);
  block("encoder.layer2.1", K_T620_3x3_K64_N128, K_T620_1x1_K64_N128, 1, c->x3, 128, 128, 128, Hc, Wc, c->h8, 128,
        128, 128, feat, 256, false, false, BK_B620_s1_K64_C128);
  if (c->fuse_blocks) {
    const BlockSpec bs{"detector.layer.0", BK_B620_s1_K64_C72, feat, 256, 128, 128, Hc, Wc, c->d0, 72, 65, 72, true, false};
    if (c->winograd && c->winograd_det)
      add_wblock(c, bs, c->winograd_det_gen3 && c->winograd_gen == 3 && w36_fits(c->B, Hc, Wc, 256, 72) ? w36_kind(65, Hc, Wc, c->w36_paired) : WK_W816_K32_C72, &bo);
    else add_block(c, bs, &bo);
  } else {  // detector.layer.0: the projection shortcut has K = 128 while conv2 has K = 72 (65 padded):
     // run the shortcut as its own 1x1 and add it as the residual of conv2
    ConvSpec s{};
    s.name = "detector.layer.0.conv1+bn1+relu";
    s.kind = K_T620_3x3_K64_N96; s.ksize = 3; s.stride = 1;
    s.in0 = feat; s.cs0 = 256; s.cin0 = 128; s.cin0_pad = 128; s.H0 = Hc; s.W0 = Wc;
    s.out = c->dh; s.cso = 72; s.cout = 65; s.nstore = 72; s.Ho = Hc; s.Wo = Wc; s.relu = 1;
    add_conv(c, s, &bo);
    ConvSpec p{};
    p.name = "detector.layer.0.identity_downsample";
    p.kind = K_T620_1x1_K64_N96; p.ksize = 1; p.stride = 1;
    p.in0 = feat; p.cs0 = 256; p.cin0 = 128; p.cin0_pad = 128; p.H0 = Hc; p.W0 = Wc;
    p.out = c->dproj; p.cso = 72; p.cout = 65; p.nstore = 72; p.Ho = Hc; p.Wo = Wc; p.relu = 0;
    add_conv(c, p, &bo);
    ConvSpec t{};
    t.name = "detector.layer.0.conv2+bn2+shortcut+relu";
    t.kind = K_T620_1x1_K72_N96; t.ksize = 1; t.stride = 1;
    t.in0 = c->dh; t.cs0 = 72; t.cin0 = 65; t.cin0_pad = 72; t.H0 = Hc; t.W0 = Wc;
    t.res = c->dproj; t.csr = 72;
    t.out = c->d0; t.cso = 72; t.cout = 65; t.nstore = 72; t.Ho = Hc; t.Wo = Wc; t.relu = 1;
    add_conv(c, t, &bo);
  }
  block("detector.layer.1", K_T620_3x3_K72_N96, K_T620_1x1_K72_N96, 1, c->d0, 72, 65, 72, Hc, Wc, c->dh, 72, 65, 72,
        c->lg, 72, false, false, BK_B620_s1_K72_C72);
  {
    Op op;
    op.type = OP_SOFTMAX;
    op.name = "exp-softmax+depth_to_space+threshold";
    c->ops.push_back(op);
    c->convw.push_back({});
  }
  if (de) {
    block("descriptor.layer_in.0", K_T620_3x3s2_K32_N128, K_T620_1x1_K64_N128, 2, feat, 256, 128, 128, Hc, Wc, c->h16,
          256, 256, 256, c->y16a, 256, true, true, BK_B320_s2_K32_C256);
    if (c->fuse_blocks) retile_last(c, BK_B310_s2_K32_C256);
    if (c->fuse_blocks && c->winograd && c->winograd_in1) {
      // 256 channels are too wide for the fused Winograd block (accumulators): conv1 as two conv-only Winograd
      // launches of 128 output channels each, then conv2 + identity + ReLU as a 1x1 launch (h makes one round trip)
      const std::string p = "descriptor.layer_in.1";
      // Generation 3: FOUR parts of 64 channels (the NB = 1 instance, gridDim.y = 4).  A 30 x 40 map is 6 tiles per
      // frame: 192 tiles x 2 halves were 384 one-tile workgroups = 1.5 rounds of the 256 CUs at the 128-channel tile's
      // time; 192 x 4 = 768 half-as-long tiles are 3 full rounds (64 workgroups per part walk 3 tiles each).
      const bool g3 = c->winograd_gen == 3 && w36_fits(c->B, H16, W16, 256, 256);
      const int in1_parts = g3 ? 4 : 2;
      for (int n0 = 0; n0 < 256; n0 += 256 / in1_parts)
        add_wconv(c, p, true, g3 ? w36_kind(64, H16, W16, c->w36_paired) : c->winograd_gen >= 2 ? WK_W16_C128 : WK_W816_K32_C128, c->y16a, 256, 256, H16, W16, c->h16, 256, 256, n0, true, &bo);
      const size_t in1_ops = in1_parts + 1;      // [the launch, its shadows, the 1x1 below]
      ConvSpec t{};
      t.name = p + ".conv2+bn2+identity+relu";
      t.kind = K_T620_1x1_K64_N128; t.ksize = 1; t.stride = 1;
      t.in0 = c->h16; t.cs0 = 256; t.cin0 = 256; t.cin0_pad = 256; t.H0 = H16; t.W0 = W16;
      t.res = c->y16a; t.csr = 256;
      t.out = c->y16b; t.cso = 256; t.cout = 256; t.nstore = 256; t.Ho = H16; t.Wo = W16; t.relu = 1;
      t.desc_branch = true;
      add_conv(c, t, &bo);
      retile_last(c, K_T320_1x1_K64_N128);
      if (g_wkinds[c->ops[c->ops.size() - in1_ops].wkind].gen == 2 && c->latency_tiles) {
        // calls of a few frames: the same two-half launch on 4 x 16 tiles (24 tiles x 2 halves per frame instead of
        // 12 x 2), then the same 1x1 -- 0.10 ms for one frame against 0.16 ms for the fused direct block on 20 tiles
        const size_t ia = c->ops.size() - in1_ops;   // [two-half launch, its shadow, 1x1]
        c->ops[ia].when = 2;
        Op oh = c->ops[ia];
        oh.when = 1;
        oh.wkind = WK_W16_C128H;
        oh.wargs.tiles_y = (H16 + 3) / 4;
        oh.mfma_flops_per_frame = 2 * 2.0 * oh.wargs.tiles_x * oh.wargs.tiles_y * 128.0 * (16.0 * 16 * 256);
        const fpc_ctx::ConvW cwa = c->convw[ia];
        c->ops.insert(c->ops.begin() + ia + 1, oh);
        c->convw.insert(c->convw.begin() + ia + 1, cwa);
      } else if (g_wkinds[c->ops[c->ops.size() - in1_ops].wkind].gen == 3 && c->latency_tiles) {
        // generation 3's fragments have 36 positions: the small-call variant (generation 2 on 4 x 16 tiles, then the same
        // 1x1) gets fragments of its own
        for (size_t k = c->ops.size() - in1_ops; k < c->ops.size(); ++k) c->ops[k].when = 2;
        const size_t ib = c->ops.size();
        for (int n0 = 0; n0 < 256; n0 += 128)
          add_wconv(c, p, true, WK_W16_C128, c->y16a, 256, 256, H16, W16, c->h16, 256, 256, n0, true, &bo);
        add_conv(c, t, &bo);
        retile_last(c, K_T320_1x1_K64_N128);
        Op& oh = c->ops[ib];
        oh.wkind = WK_W16_C128H;
        oh.wargs.tiles_y = (H16 + 3) / 4;
        oh.mfma_flops_per_frame = 2 * 2.0 * oh.wargs.tiles_x * oh.wargs.tiles_y * 128.0 * (16.0 * 16 * 256);
        for (size_t k = ib; k < c->ops.size(); ++k) c->ops[k].when = 1;
      } else if (c->latency_tiles || c->winograd_gen == 1) {
      for (size_t k = c->ops.size() - in1_ops; k < c->ops.size(); ++k) c->ops[k].when = 2;
      // a call of a few frames has 12 tiles per frame here: three dependent launches cost more latency than they save
      // work, so small calls take the fused direct block instead (its weights sit in the blob next to the others)
      block("descriptor.layer_in.1", K_T620_3x3_K64_N128, K_T620_1x1_K64_N128, 1, c->y16a, 256, 256, 256, H16, W16,
            c->h16, 256, 256, 256, c->y16b, 256, false, true, BK_B320_s1_K64_C256);
      c->ops.back().when = 1;
      }
    } else
    block("descriptor.layer_in.1", K_T620_3x3_K64_N128, K_T620_1x1_K64_N128, 1, c->y16a, 256, 256, 256, H16, W16,
          c->h16, 256, 256, 256, c->y16b, 256, false, true, BK_B320_s1_K64_C256);
    ConvSpec u{};
    u.name = "descriptor.up_sample+bn+relu";
    u.kind = K_T620_2x2_K64_N128; u.ksize = 2; u.stride = 1;
    u.in0 = c->y16b; u.cs0 = 256; u.cin0 = 256; u.cin0_pad = 256; u.H0 = H16; u.W0 = W16;
    u.out = c->cat; u.cso = 256; u.cout = 128; u.nstore = 128; u.Ho = Hc; u.Wo = Wc; u.relu = 1;
    u.desc_branch = true;
    add_conv(c, u, &bo);
    retile_last(c, K_T320_2x2_K64_N128);
    block("descriptor.layer_out.0", K_T620_3x3_K64_N128, K_T620_1x1_K64_N128, 1, c->cat, 256, 256, 256, Hc, Wc,
          c->lo_h, 128, 128, 128, c->lo0, 128, true, true, BK_B620_s1_K64_C128);
    block("descriptor.layer_out.1", K_T620_3x3_K64_N128, K_T620_1x1_K64_N128, 1, c->lo0, 128, 128, 128, Hc, Wc,
          c->lo_h, 128, 128, 128, c->desc_map, 128, false, true, BK_B620_s1_K64_C128);
  }
  }
postproc:
  {
    Op op;
    op.type = OP_NMS;
    op.name = "nms+sort+border_crop";
    c->ops.push_back(op);
    c->convw.push_back({});
    if (de) {
      op.type = OP_DESC;
      op.name = "descriptor_sample+l2norm";
      c->ops.push_back(op);
      c->convw.push_back({});
    }
  }
  for (Op& op : c->ops)
    if (op.type == OP_SOFTMAX) op.bytes_per_frame = 4.0 * (65.0 * Hc * Wc + 2.0 * H * W);   // logits in; prob + NMS state map out
  c->blob_floats = bo;
  HIPCHECK(hipMalloc((void**)&c->blob, c->blob_floats * sizeof(float)));
  HIPCHECK(hipMemset(c->blob, 0, c->blob_floats * sizeof(float)));
  // resolve weight pointers
  for (size_t i = 0; i < c->ops.size(); ++i) {
    Op& op = c->ops[i];
    if (op.type == OP_WBLOCK) {
      op.wargs.w1 = reinterpret_cast<const float4*>(c->blob + c->convw[i].w_off[0]);
      op.wargs.b1 = c->blob + c->convw[i].b_off;
      op.wargs.w2 = reinterpret_cast<const float4*>(c->blob + c->convw[i].w_off[1]);
      op.wargs.b2 = c->blob + c->convw[i].b2_off;
      op.wargs.dust = g_wkinds[op.wkind].dust ? c->blob + c->convw[i].w_off[2] : nullptr;
    }
    if (op.type == OP_BF16) {
      op.fargs.w1 = reinterpret_cast<const uint4*>(c->blob + c->convw[i].w_off[0]);
      op.fargs.b1 = c->blob + c->convw[i].b_off;
      op.fargs.w2 = reinterpret_cast<const uint4*>(c->blob + c->convw[i].w_off[1]);
      op.fargs.b2 = c->blob + c->convw[i].b2_off;
    }
    if (op.type == OP_BLOCK) {
      op.bargs.w1 = reinterpret_cast<const float4*>(c->blob + c->convw[i].w_off[0]);
      op.bargs.b1 = c->blob + c->convw[i].b_off;
      op.bargs.w2 = reinterpret_cast<const float4*>(c->blob + c->convw[i].w_off[1]);
      op.bargs.b2 = c->blob + c->convw[i].b2_off;
    }
    if (op.type != OP_CONV) continue;
    for (int z = 0; z < op.grid_z; ++z)
      op.args.sub[z].wfrag = reinterpret_cast<const float4*>(c->blob + c->convw[i].w_off[z]);
    op.args.bias = c->blob + c->convw[i].b_off;
  }
  c->stem.wfrag = reinterpret_cast<const float4*>(c->blob + c->stem_w_off);
  c->stem.bias = c->blob + c->stem_b_off;
  c->stem.out = c->stem_out;
  c->stem.H = H; c->stem.W = W; c->stem.Ho = H2; c->stem.Wo = W2;
  c->stem.tiles_x = (W2 + STEM_T - 1) / STEM_T;
  c->stem.tiles_y = (H2 + STEM_T - 1) / STEM_T;
  // round 5: the fp32 stem + pool on stem_pool2_kernel where the conv map's width is whole tiles (VGA, HD, QVGA).  The fragments carry the bias as a K step in EVERY geometry (pack layout revision 4; stem_pool_kernel and
  // stem_kernel skip that step), so the choice is a launch-time one and blobs stay exchangeable between geometries
  c->stem2 = c->stem_lean && !c->bf16 && !c->split && !c->vgg && (c->fuse_stem_pool || c->cin == 1) && H2 % 2 == 0 && W2 % STEM_T == 0;
  return FPC_OK;
}

// ---- the packed blob's tag ----------------------------------------------------------------
// The first 16 words of the blob describe what packed it, so that a blob that travels (fpc_export_packed ->
// fpc_import_packed, the RCCL broadcast) is never interpreted with another layout: magic, ABI version, dtype, arch,
// a hash of the launch plan (kernel instances and fragment offsets of every layer), the blob size.
static uint64_t plan_hash(const fpc_ctx* c) {
  uint64_t h = 1469598103934665603ull;
  auto mix = [&](uint64_t v) {
    for (int i = 0; i < 8; ++i) {
      h ^= (v >> (8 * i)) & 0xff;
      h *= 1099511628211ull;
    }
  };
  mix(FPC_ABI_VERSION); mix(FPC_PACK_LAYOUT_REVISION); mix((uint64_t)c->cfg.dtype); mix((uint64_t)c->cfg.arch); mix((uint64_t)c->cin);
  mix((uint64_t)(c->cfg.descriptor_enabled != 0)); mix(c->blob_floats); mix(c->stem_w_off); mix(c->stem_b_off);
  mix(c->vconv0_off);
  for (size_t i = 0; i < c->ops.size(); ++i) {
    const Op& op = c->ops[i];
    mix((uint64_t)op.type); mix((uint64_t)op.kind); mix((uint64_t)op.bkind); mix((uint64_t)op.wkind); mix((uint64_t)op.fkind);
    mix((uint64_t)(op.phase + 1)); mix((uint64_t)op.n0);
    for (char ch : op.prefix) mix((uint64_t)(unsigned char)ch);
    const fpc_ctx::ConvW& cw = c->convw[i];
    for (int z = 0; z < 4; ++z) mix(cw.w_off[z]);
    mix(cw.b_off); mix(cw.b2_off);
  }
  return h;
}

static void fill_blob_header(const fpc_ctx* c, uint32_t* h) {
  memset(h, 0, BLOB_HEADER_FLOATS * sizeof(uint32_t));
  const uint64_t ph = plan_hash(c);
  h[0] = BLOB_MAGIC; h[1] = FPC_ABI_VERSION; h[2] = (uint32_t)c->cfg.dtype; h[3] = (uint32_t)c->cfg.arch;
  h[4] = (uint32_t)ph; h[5] = (uint32_t)(ph >> 32);
  h[6] = (uint32_t)c->blob_floats; h[7] = (uint32_t)((uint64_t)c->blob_floats >> 32);
  h[8] = 1;  // holds weights
  h[9] = FPC_PACK_LAYOUT_REVISION;
}

static bool check_blob_header(const fpc_ctx* c, const uint32_t* h, std::string* why) {
  uint32_t want[BLOB_HEADER_FLOATS];
  fill_blob_header(c, want);
  static const char* what[10] = {"magic", "ABI version", "dtype", "arch", "launch plan", "launch plan", "size", "size", "weights-present flag",
                                 "fragment-layout revision"};
  for (int i = 0; i < 10; ++i)
    if (h[i] != want[i]) {
      *why = std::string("packed weights do not match this context: ") + what[i] + " differs (the blob was packed by another "
             "build, dtype, arch or plan, or holds no weights)";
      return false;
    }
  return true;
}

// ---- Winograd-domain filters U = G g G^T (x the folded BN scale), in double ------------------------------------
// F(2x2,3x3): 16 positions (generations 1, 2); F(4x4,3x3) on the points 0, +-1, +-2, inf: 36 positions (generation 3).
// U[(n * ci + cc) * P + i * R + j] for output channels n0 .. n0 + nn - 1 of w1 [co][ci][3][3].
static std::vector<double> winograd_filters(const float* w1, int n0, int nn, int ci, const std::vector<double>& scale, int P) {
  static const double G2[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
  static const double G4[6][3] = {{1.0 / 4, 0, 0},          {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                  {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
  const int R = P == 36 ? 6 : 4;
  const double(*G)[3] = P == 36 ? G4 : G2;
  std::vector<double> U((size_t)nn * ci * P);
  for (int n = 0; n < nn; ++n)
    for (int cc = 0; cc < ci; ++cc) {
      const float* g = w1 + ((size_t)(n0 + n) * ci + cc) * 9;
      double t[6][3];
      for (int i = 0; i < R; ++i)
        for (int j = 0; j < 3; ++j) t[i][j] = G[i][0] * g[0 * 3 + j] + G[i][1] * g[1 * 3 + j] + G[i][2] * g[2 * 3 + j];
      for (int i = 0; i < R; ++i)
        for (int j = 0; j < R; ++j)
          U[((size_t)n * ci + cc) * P + i * R + j] = (t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2]) * scale[n0 + n];
    }
  return U;
}

// ---- generation-3 Winograd fragments (wblock36_mfma.h): as generation 2's with 36 positions per chunk --------------
static void pack_w36_winograd(const std::vector<double>& U, int nn, int ci, const WKindInfo& k, int nchunk, float* dst) {
  const size_t gstride = w1_floats(k, nchunk) / k.NCG;
  for (int cg = 0; cg < k.NCG; ++cg)
    for (int ch = 0; ch < nchunk; ++ch)
      for (int xi = 0; xi < 36; ++xi)
        for (int lane = 0; lane < 64; ++lane) {
          const int n = 16 * cg + (lane & 15), kq = lane >> 4;
          float* d4 = dst + cg * gstride + (((size_t)ch * 36 + xi) * 64 + lane) * 4;
          for (int j = 0; j < 4; ++j) {
            const int cc = 16 * ch + 4 * kq + j;
            d4[j] = (n < nn && cc < ci) ? (float)U[((size_t)n * ci + cc) * 36 + xi] : 0.f;
          }
        }
}

// ---- generation-2 Winograd fragments (wblock16_mfma.h) ---------------------------------------------
// U: [nn][ci][16] Winograd-domain filters (scaled); dst: [channel group][chunk of 16][position][lane] float4 with
// lane (n = l & 15, kq = l >> 4) holding { U[16 cg + n][16 chunk + 4 kq + j][pos] }, j = 0..3; WPAD zero steps per group.
static void pack_w16_winograd(const std::vector<double>& U, int nn, int ci, int ncg, int nchunk, float* dst) {
  const size_t gstride = ((size_t)nchunk * 16 + W16Cfg<8>::WPAD) * 256;
  for (int cg = 0; cg < ncg; ++cg)
    for (int ch = 0; ch < nchunk; ++ch)
      for (int xi = 0; xi < 16; ++xi)
        for (int lane = 0; lane < 64; ++lane) {
          const int n = 16 * cg + (lane & 15), kq = lane >> 4;
          float* d4 = dst + cg * gstride + (((size_t)ch * 16 + xi) * 64 + lane) * 4;
          for (int j = 0; j < 4; ++j) {
            const int cc = 16 * ch + 4 * kq + j;
            d4[j] = (n < nn && cc < ci) ? (float)U[((size_t)n * ci + cc) * 16 + xi] : 0.f;
          }
        }
}

// 1x1 over [h | x]: sources concatenated along K (each a multiple of 16 channels); dst [channel group][step of 16][lane] float4
static void pack_w16_1x1(const std::vector<PackSource>& srcs, int cout, int ncg, float* dst) {
  size_t steps = 0;
  for (auto& s : srcs) steps += (size_t)s.cin_pad / 16;
  const size_t gstride = (steps + W16Cfg<8>::WPAD) * 256;
  size_t step0 = 0;
  for (auto& s : srcs) {
    for (int g = 0; g < s.cin_pad / 16; ++g)
      for (int cg = 0; cg < ncg; ++cg)
        for (int lane = 0; lane < 64; ++lane) {
          const int n = 16 * cg + (lane & 15), kq = lane >> 4;
          float* d4 = dst + cg * gstride + ((step0 + g) * 64 + lane) * 4;
          for (int j = 0; j < 4; ++j) {
            const int cc = 16 * g + 4 * kq + j;
            d4[j] = (n < cout && cc < s.cin) ? (float)(s.w(n, cc, 0) * (*s.scale)[n]) : 0.f;
          }
        }
    step0 += (size_t)s.cin_pad / 16;
  }
}

// ---- checkpoint -> blob -----------------------------------------------------------------
static int pack_all_impl(fpc_ctx* c, const TensorMap& m, std::string* missing, bool& range_bad);
static int pack_all(fpc_ctx* c, const TensorMap& m, std::string* missing) {
  bool range_bad = false;
  const int rc = pack_all_impl(c, m, missing, range_bad);
  if (rc == FPC_OK && range_bad) {
    *missing = "a BatchNorm-folded weight exceeds fp16's range (|w| > 65504): FPC_F32_SPLIT_F16 cannot represent it";
    return FPC_E_RANGE;
  }
  if (rc == FPC_OK) fill_blob_header(c, reinterpret_cast<uint32_t*>(c->host_blob.data()));
  return rc;
}
static int pack_all_impl(fpc_ctx* c, const TensorMap& m, std::string* missing, bool& range_bad) {
  std::vector<float>& blob = c->host_blob;
  blob.assign(c->blob_floats, 0.f);
  auto need = [&](const std::string& k, std::initializer_list<int64_t> shp) -> const float* {
    if (!has_shape(m, k, shp)) {
      if (missing->empty()) *missing = k;
      return nullptr;
    }
    return m.at(k).data;
  };
  if (c->vgg) {  // superpoint::SPModel: Conv2d weights + biases, no BatchNorm (cpp/src/model.cc:4-58)
    static const std::vector<double> ones(256, 1.0);
    for (size_t i = 0; i < c->ops.size(); ++i) {
      const Op& op = c->ops[i];
      if (op.type == OP_VCONV0) {
        const float* w = need(op.prefix + ".weight", {64, 1, 3, 3});
        const float* bv = need(op.prefix + ".bias", {64});
        if (!w || !bv) return FPC_E_MISSING_KEY;
        float* dst = blob.data() + c->vconv0_off;
        for (int t = 0; t < 9; ++t)
          for (int n = 0; n < 64; ++n) dst[t * 64 + n] = w[n * 9 + t];
        for (int n = 0; n < 64; ++n) dst[576 + n] = bv[n];
        continue;
      }
      if (!op.plain_conv || op.type == OP_WBLOCK) continue;   // Winograd conv-only launches: generic loop below
      const int ci = op.cin, co = op.cout, kk = op.ksize * op.ksize;
      const float* w = need(op.prefix + ".weight", {co, ci, op.ksize, op.ksize});
      const float* bv = need(op.prefix + ".bias", {co});
      if (!w || !bv) return FPC_E_MISSING_KEY;
      const fpc_ctx::ConvW& cw = c->convw[i];
      PackSource src{ci, ci, kk, [&](int n, int cc, int t) { return (double)w[((size_t)n * ci + cc) * kk + t]; }, &ones};
      std::vector<float> frag;
      if (op.type == OP_BF16) {
        const FKindInfo& k = g_fkinds[op.fkind];
        frag = pack_conv_bf16({src}, co, k.WN * k.NB, k.KC, k.planes, &range_bad);
      } else {
        frag = pack_conv({src}, co, op.args.nbt, g_kinds[op.kind].KC);
      }
      memcpy(blob.data() + cw.w_off[0], frag.data(), frag.size() * sizeof(float));
      for (int n = 0; n < co; ++n) blob[cw.b_off + n] = bv[n];
    }
  }
  // stem
  if (!c->vgg) {
    const float* w = need("encoder.conv1.weight", {64, 3, 7, 7});
    Fold f;
    if (!w || !fold_bn(m, "encoder.bn1", 64, &f, missing)) return FPC_E_MISSING_KEY;
    float* dst = blob.data() + c->stem_w_off;
    const int kreal = c->cin * 49, kg = (kreal + 7) / 8;
    if (c->split || c->bf16) {  // stem_pool_x3_kernel: [step][plane][nb][lane] x 8 bf16; k = (row = 2*step + half, kx = j)
      uint16_t* d16 = reinterpret_cast<uint16_t*>(dst);
      const int rows = c->cin * 7, steps = (rows + 1) / 2;
      for (int st = 0; st < steps; ++st)
        for (int nb = 0; nb < 2; ++nb)
          for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
              const int row = 2 * st + (lane >> 5), n = nb * 32 + (lane & 31);
              double v = 0.0;
              if (row < rows && j > 0) {     // j = 0: the zero-weight pad (LDS column 0 is one pixel left of the window)
                const int k = row * 7 + j - 1;  // (c, ky, kx = j - 1) flattened exactly as conv1.weight[n]
                if (c->cin == 3) v = (double)w[n * 147 + k];
                else v = (double)w[n * 147 + k] + (double)w[n * 147 + 49 + k] + (double)w[n * 147 + 98 + k];
              }
              const float wv = (float)(v * f.s[n]);
              if (c->bf16) {
                d16[(((size_t)st * 2 + nb) * 64 + lane) * 8 + j] = host_f2bf(wv);
              } else if (c->split_f16) {
                if (!(std::fabs(wv) <= 65504.f)) range_bad = true;
                uint16_t t2[2];
                host_split2_f16(wv, t2);
                for (int pl = 0; pl < 2; ++pl) d16[((((size_t)st * 2 + pl) * 2 + nb) * 64 + lane) * 8 + j] = t2[pl];
              } else {
                uint16_t t3[3];
                host_split3(wv, t3);
                for (int pl = 0; pl < 3; ++pl) d16[((((size_t)st * 3 + pl) * 2 + nb) * 64 + lane) * 8 + j] = t3[pl];
              }
            }
    } else
    for (int g = 0; g < kg; ++g)
      for (int nb = 0; nb < 2; ++nb)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 4; ++j) {
            const StemPair sp = stem_pair2(c->cin, g * 4 + j);      // the kernels' K order (kernels_misc.h), bias step included
            const int k = (lane >> 5) ? sp.wb : sp.wa, n = nb * 32 + (lane & 31);
            if (k == STEM_BIAS_TAP) {   // the folded-BN bias as a K step (stem_pool2_kernel: A operand 1.0; the other stem kernels skip the step)
              dst[(((size_t)g * 2 + nb) * 64 + lane) * 4 + j] = (float)f.t[n];
              continue;
            }
            double v = 0.0;
            if (k >= 0 && k < kreal) {
              if (c->cin == 3) v = (double)w[n * 147 + k];
              else  // gray frame == the same plane in all three channels: sum the three filters
                v = (double)w[n * 147 + k] + (double)w[n * 147 + 49 + k] + (double)w[n * 147 + 98 + k];
            }
            dst[(((size_t)g * 2 + nb) * 64 + lane) * 4 + j] = (float)(v * f.s[n]);
          }
    for (int n = 0; n < 64; ++n) blob[c->stem_b_off + n] = (float)f.t[n];
  }
  for (size_t i = 0; i < c->ops.size(); ++i) {
    const Op& op = c->ops[i];
    if (c->vgg && op.type != OP_WBLOCK) continue;  // packed above
    if (op.type == OP_WBLOCK) {
      const WKindInfo& k = g_wkinds[op.wkind];
      const WBlockArgs& a = op.wargs;
      const fpc_ctx::ConvW& cw = c->convw[i];
      const std::string& p = op.prefix;
      const int ci = op.cin, co = op.cout, nbt = k.NBT, K8 = k.KC / 8;
      if (op.wconv) {  // conv-only launch: U = G g G^T of output channels n0 .. n0 + CMID - 1
        const float* w1 = need(p + (op.plain_conv ? ".weight" : ".conv1.weight"), {co, ci, 3, 3});
        Fold f1;
        if (!w1) return FPC_E_MISSING_KEY;
        if (op.plain_conv) {
          const float* bv = need(p + ".bias", {co});
          if (!bv) return FPC_E_MISSING_KEY;
          f1.s.assign(co, 1.0);
          f1.t.assign(bv, bv + co);
        } else if (!fold_bn(m, p + ".bn1", co, &f1, missing)) {
          return FPC_E_MISSING_KEY;
        }
        float* dst = blob.data() + cw.w_off[0];
        const int nn = std::min(k.CMID, co - op.n0);
        const std::vector<double> U = winograd_filters(w1, op.n0, nn, ci, f1.s, k.gen == 3 ? 36 : 16);
        if (k.gen == 3) pack_w36_winograd(U, nn, ci, k, a.nchunk, dst);
        else if (k.gen == 2) pack_w16_winograd(U, nn, ci, k.NCG, a.nchunk, dst);
        else
        for (int ch = 0; ch < a.nchunk; ++ch)
          for (int xi = 0; xi < 16; ++xi)
            for (int k8 = 0; k8 < K8; ++k8)
              for (int nb = 0; nb < nbt; ++nb)
                for (int lane = 0; lane < 64; ++lane) {
                  const int n = nb * 32 + (lane & 31), half = lane >> 5;
                  float* d4 = dst + (((((size_t)ch * 16 + xi) * K8 + k8) * nbt + nb) * 64 + lane) * 4;
                  for (int j = 0; j < 4; ++j) {
                    const int cc = ch * k.KC + k8 * 8 + 4 * half + j;
                    d4[j] = (n < nn && cc < ci) ? (float)U[((size_t)n * ci + cc) * 16 + xi] : 0.f;
                  }
                }
        for (int n = 0; n < nn; ++n) blob[cw.b_off + n] = (float)f1.t[op.n0 + n];
        continue;
      }
      const float* w1 = need(p + ".conv1.weight", {co, ci, 3, 3});
      const float* w2 = need(p + ".conv2.weight", {co, co, 1, 1});
      Fold f1, f2, fp;
      if (!w1 || !w2 || !fold_bn(m, p + ".bn1", co, &f1, missing) || !fold_bn(m, p + ".bn2", co, &f2, missing))
        return FPC_E_MISSING_KEY;
      // U = G (g * s) G^T in double, rounded once; generation 1: [chunk][xi][k8][nb][lane] float4
      const std::vector<double> U = winograd_filters(w1, 0, co, ci, f1.s, k.gen == 3 ? 36 : 16);
      float* dst = blob.data() + cw.w_off[0];
      if (k.gen == 3) pack_w36_winograd(U, co, ci, k, a.nchunk, dst);
      else if (k.gen == 2) pack_w16_winograd(U, co, ci, k.NCG, a.nchunk, dst);
      else
      for (int ch = 0; ch < a.nchunk; ++ch)
        for (int xi = 0; xi < 16; ++xi)
          for (int k8 = 0; k8 < K8; ++k8)
            for (int nb = 0; nb < nbt; ++nb)
              for (int lane = 0; lane < 64; ++lane) {
                const int n = nb * 32 + (lane & 31), half = lane >> 5;
                float* d4 = dst + (((((size_t)ch * 16 + xi) * K8 + k8) * nbt + nb) * 64 + lane) * 4;
                for (int j = 0; j < 4; ++j) {
                  const int cc = ch * k.KC + k8 * 8 + 4 * half + j;
                  d4[j] = (n < co && cc < ci) ? (float)U[((size_t)n * ci + cc) * 16 + xi] : 0.f;
                }
              }
      for (int n = 0; n < std::min(co, nbt * 32); ++n) blob[cw.b_off + n] = (float)f1.t[n];
      std::vector<PackSource> srcs;
      srcs.push_back({co, a.k8_h * 8, 1, [&](int n, int c_, int) { return (double)w2[(size_t)n * co + c_]; }, &f2.s});
      std::vector<double> bias(f2.t);
      const float* wp = nullptr;
      if (a.k8_x > 0) {
        wp = need(p + ".identity_downsample.0.weight", {co, ci, 1, 1});
        if (!wp || !fold_bn(m, p + ".identity_downsample.1", co, &fp, missing)) return FPC_E_MISSING_KEY;
        srcs.push_back({ci, a.k8_x * 8, 1, [&](int n, int c_, int) { return (double)wp[(size_t)n * ci + c_]; }, &fp.s});
        for (int n = 0; n < co; ++n) bias[n] += fp.t[n];
      }
      if (k.gen >= 2) {
        pack_w16_1x1(srcs, co, k.NCG, blob.data() + cw.w_off[1]);
      } else {
        std::vector<float> frag = pack_conv(srcs, co, nbt, 8);
        memcpy(blob.data() + cw.w_off[1], frag.data(), frag.size() * sizeof(float));
      }
      for (int n = 0; n < std::min(co, nbt * 32); ++n) blob[cw.b2_off + n] = (float)bias[n];
      if (k.dust) {   // the 65th channel's weights (W36Dust, wblock36_mfma.h)
        float* d = blob.data() + cw.w_off[2];
        for (int n = 0; n < 64; ++n) d[W36Dust::W2ROW + n] = (float)((double)w2[(size_t)n * co + 64] * f2.s[n]);
        for (int kk = 0; kk < 65; ++kk) d[W36Dust::W2COL + kk] = (float)((double)w2[(size_t)64 * co + kk] * f2.s[64]);
        if (wp)
          for (int cc = 0; cc < ci; ++cc) d[W36Dust::WPCOL + cc] = (float)((double)wp[(size_t)64 * ci + cc] * fp.s[64]);
        d[W36Dust::B1] = (float)f1.t[64];
        d[W36Dust::B2] = (float)bias[64];
        if (a.dust_in) {
          for (int pp = 0; pp < 36; ++pp) d[W36Dust::UIN64 + pp] = (float)U[((size_t)64 * ci + 64) * 36 + pp];
          for (int cg = 0; cg < 4; ++cg)
            for (int pp = 0; pp < 36; ++pp)
              for (int n16 = 0; n16 < 16; ++n16)
                d[W36Dust::UIN + (cg * 36 + pp) * 16 + n16] = (float)U[((size_t)(16 * cg + n16) * ci + 64) * 36 + pp];
        }
        for (int ch = 0; ch < a.nchunk; ++ch)
          for (int pp = 0; pp < 36; ++pp)
            for (int kk = 0; kk < 16; ++kk)
              d[W36Dust::UOUT + (ch * 36 + pp) * 16 + kk] = 16 * ch + kk < ci ? (float)U[((size_t)64 * ci + 16 * ch + kk) * 36 + pp] : 0.f;
      }
      continue;
    }
    if (op.type == OP_BF16) {
      const FKindInfo& k = g_fkinds[op.fkind];
      const BlockBfArgs& a = op.fargs;
      const fpc_ctx::ConvW& cw = c->convw[i];
      const std::string& p = op.prefix;
      const int ci = op.cin, co = op.cout, nbt = k.WN * k.NB;
      if (op.phase >= 0) {  // ConvTranspose phase
        const float* w = need("descriptor.up_sample.weight", {256, 128, 3, 3});
        const float* bct = need("descriptor.up_sample.bias", {128});
        Fold f;
        if (!w || !bct || !fold_bn(m, "descriptor.bn", 128, &f, missing)) return FPC_E_MISSING_KEY;
        if (op.phase == 4) {   // convt_bf16_kernel: all nine taps per step of 16 channels, in ConvTTaps' order
          PackSource s{ci, ci, 9,
                       [&](int n, int cc, int t) { return (double)w[((cc * 128 + n) * 3 + ConvTTaps::ky(t)) * 3 + ConvTTaps::kx(t)]; },
                       &f.s};
          std::vector<float> frag = pack_conv_bf16({s}, co, k.CMIDP / 32, 16, 1, &range_bad);
          memcpy(blob.data() + cw.w_off[0], frag.data(), frag.size() * sizeof(float));
          for (int n = 0; n < co; ++n) blob[cw.b_off + n] = (float)((double)bct[n] * f.s[n] + f.t[n]);
          continue;
        }
        const int py = op.phase >> 1, px = op.phase & 1;
        std::vector<std::pair<int, int>> taps;  // (ky, kx) in the order add_fconvT laid the taps out
        for (int iy = 0; iy < (py ? 2 : 1); ++iy)
          for (int ix = 0; ix < (px ? 2 : 1); ++ix) taps.push_back({py ? (iy == 0 ? 0 : 2) : 1, px ? (ix == 0 ? 0 : 2) : 1});
        PackSource s{ci, a.nchunk * k.KC, (int)taps.size(),
                     [&](int n, int cc, int t) { return (double)w[((cc * 128 + n) * 3 + taps[t].first) * 3 + taps[t].second]; },
                     &f.s};
        std::vector<float> frag = pack_conv_bf16({s}, co, nbt, k.KC, k.planes, &range_bad);
        memcpy(blob.data() + cw.w_off[0], frag.data(), frag.size() * sizeof(float));
        for (int n = 0; n < co; ++n) blob[cw.b_off + n] = (float)((double)bct[n] * f.s[n] + f.t[n]);
        continue;
      }
      const float* w1 = need(p + ".conv1.weight", {co, ci, 3, 3});
      const float* w2 = need(p + ".conv2.weight", {co, co, 1, 1});
      Fold f1, f2, fp;
      if (!w1 || !w2 || !fold_bn(m, p + ".bn1", co, &f1, missing) || !fold_bn(m, p + ".bn2", co, &f2, missing))
        return FPC_E_MISSING_KEY;
      const bool sc_in_w1 = k.planes == 1;   // (add_fblock)
      std::vector<double> bias(f2.t);
      const float* wp = nullptr;
      if (a.k16_x > 0) {
        wp = need(p + ".identity_downsample.0.weight", {co, ci, 1, 1});
        if (!wp || !fold_bn(m, p + ".identity_downsample.1", co, &fp, missing)) return FPC_E_MISSING_KEY;
        for (int n = 0; n < co; ++n) bias[n] += fp.t[n];
      }
      const std::vector<double> ones(co, 1.0);
      // taps 0..8: conv1 x bn1's scale; tap 9 (FPC_BF16): the shortcut on the same channels -- the projection x its bn's
      // scale, or the unit matrix
      PackSource s1{ci, a.nchunk * k.KC, sc_in_w1 ? 10 : 9,
                    [&](int n, int c_, int t) {
                      if (t < 9) return (double)w1[((size_t)(n * ci + c_)) * 9 + t] * f1.s[n];
                      return wp ? (double)wp[(size_t)n * ci + c_] * fp.s[n] : (n == c_ ? 1.0 : 0.0);
                    },
                    &ones};
      std::vector<float> frag = pack_conv_bf16({s1}, co, nbt, k.KC, k.planes, &range_bad);
      memcpy(blob.data() + cw.w_off[0], frag.data(), frag.size() * sizeof(float));
      for (int n = 0; n < co; ++n) blob[cw.b_off + n] = (float)f1.t[n];
      std::vector<PackSource> srcs;
      srcs.push_back({co, a.k16_h * 16, 1, [&](int n, int c_, int) { return (double)w2[(size_t)n * co + c_]; }, &f2.s});
      if (wp && !sc_in_w1)
        srcs.push_back({ci, a.k16_x * 16, 1, [&](int n, int c_, int) { return (double)wp[(size_t)n * ci + c_]; }, &fp.s});
      frag = pack_conv_bf16(srcs, co, nbt, 16, k.planes, &range_bad);
      memcpy(blob.data() + cw.w_off[1], frag.data(), frag.size() * sizeof(float));
      for (int n = 0; n < co; ++n) blob[cw.b2_off + n] = (float)bias[n];
      continue;
    }
    if (op.type == OP_BLOCK) {
      const BKindInfo& k = g_bkinds[op.bkind];
      const BlockArgs& a = op.bargs;
      const fpc_ctx::ConvW& cw = c->convw[i];
      const std::string& p = op.prefix;
      const int ci = op.cin, co = op.cout, nbt = k.WN * k.NB;
      const float* w1 = need(p + ".conv1.weight", {co, ci, 3, 3});
      const float* w2 = need(p + ".conv2.weight", {co, co, 1, 1});
      Fold f1, f2, fp;
      if (!w1 || !w2 || !fold_bn(m, p + ".bn1", co, &f1, missing) || !fold_bn(m, p + ".bn2", co, &f2, missing))
        return FPC_E_MISSING_KEY;
      PackSource s1{ci, a.nchunk * k.KC, 9, [&](int n, int c_, int t) { return (double)w1[((size_t)(n * ci + c_)) * 9 + t]; }, &f1.s};
      std::vector<float> frag = pack_conv({s1}, co, nbt, k.KC);
      memcpy(blob.data() + cw.w_off[0], frag.data(), frag.size() * sizeof(float));
      for (int n = 0; n < co; ++n) blob[cw.b_off + n] = (float)f1.t[n];
      std::vector<PackSource> srcs;
      srcs.push_back({co, a.k8_h * 8, 1, [&](int n, int c_, int) { return (double)w2[(size_t)n * co + c_]; }, &f2.s});
      std::vector<double> bias(f2.t);
      const float* wp = nullptr;
      if (a.k8_x > 0) {
        wp = need(p + ".identity_downsample.0.weight", {co, ci, 1, 1});
        if (!wp || !fold_bn(m, p + ".identity_downsample.1", co, &fp, missing)) return FPC_E_MISSING_KEY;
        srcs.push_back({ci, a.k8_x * 8, 1, [&](int n, int c_, int) { return (double)wp[(size_t)n * ci + c_]; }, &fp.s});
        for (int n = 0; n < co; ++n) bias[n] += fp.t[n];
      }
      frag = pack_conv(srcs, co, nbt, 8);
      memcpy(blob.data() + cw.w_off[1], frag.data(), frag.size() * sizeof(float));
      for (int n = 0; n < co; ++n) blob[cw.b2_off + n] = (float)bias[n];
      continue;
    }
    if (op.type != OP_CONV) continue;
    const KindInfo& k = g_kinds[op.kind];
    const ConvArgs& a = op.args;
    const fpc_ctx::ConvW& cw = c->convw[i];
    // recover the checkpoint prefix and role from the op name
    const std::string& nm = op.name;
    const size_t dot = nm.find(".conv1+");
    const size_t dot2 = nm.find(".conv2+");
    const int cin0p = a.nchunk0 * k.KC, cin1p = a.nchunk1 * k.KC;
    Fold f, fp;
    std::vector<double> bias(a.nbt * 32, 0.0);
    if (nm == "descriptor.up_sample+bn+relu") {
      const float* w = need("descriptor.up_sample.weight", {256, 128, 3, 3});
      const float* bct = need("descriptor.up_sample.bias", {128});
      if (!w || !bct || !fold_bn(m, "descriptor.bn", 128, &f, missing)) return FPC_E_MISSING_KEY;
      for (int ph = 0; ph < 4; ++ph) {
        const int py = ph >> 1, px = ph & 1;
        std::vector<std::pair<int, int>> taps;  // (ky, kx) in the order add_conv laid the taps out
        for (int iy = 0; iy < (py ? 2 : 1); ++iy)
          for (int ix = 0; ix < (px ? 2 : 1); ++ix) taps.push_back({py ? (iy == 0 ? 0 : 2) : 1, px ? (ix == 0 ? 0 : 2) : 1});
        PackSource s{256, cin0p, (int)taps.size(),
                     [&](int n, int ci, int t) { return (double)w[((ci * 128 + n) * 3 + taps[t].first) * 3 + taps[t].second]; },
                     &f.s};
        std::vector<float> frag = pack_conv({s}, 128, a.nbt, k.KC);
        memcpy(blob.data() + cw.w_off[ph], frag.data(), frag.size() * sizeof(float));
      }
      for (int n = 0; n < 128; ++n) bias[n] = (double)bct[n] * f.s[n] + f.t[n];
    } else if (nm == "detector.layer.0.identity_downsample") {
      const float* w = need("detector.layer.0.identity_downsample.0.weight", {65, 128, 1, 1});
      if (!w || !fold_bn(m, "detector.layer.0.identity_downsample.1", 65, &f, missing)) return FPC_E_MISSING_KEY;
      PackSource s{128, cin0p, 1, [&](int n, int ci, int) { return (double)w[n * 128 + ci]; }, &f.s};
      std::vector<float> frag = pack_conv({s}, 65, a.nbt, k.KC);
      memcpy(blob.data() + cw.w_off[0], frag.data(), frag.size() * sizeof(float));
      for (int n = 0; n < 65; ++n) bias[n] = f.t[n];
    } else if (dot != std::string::npos) {  // 3x3 conv1 + bn1
      const std::string p = nm.substr(0, dot);
      auto it = m.find(p + ".conv1.weight");
      if (it == m.end() || it->second.shape.size() != 4 || it->second.shape[2] != 3) {
        if (missing->empty()) *missing = p + ".conv1.weight";
        return FPC_E_MISSING_KEY;
      }
      const int co = (int)it->second.shape[0], ci = (int)it->second.shape[1];
      const float* w = it->second.data;
      if (ci > cin0p || co > a.nbt * 32 || !w) {
        *missing = p + ".conv1.weight";
        return FPC_E_MISSING_KEY;
      }
      if (!fold_bn(m, p + ".bn1", co, &f, missing)) return FPC_E_MISSING_KEY;
      PackSource s{ci, cin0p, 9, [&](int n, int c_, int t) { return (double)w[((size_t)(n * ci + c_)) * 9 + t]; }, &f.s};
      std::vector<float> frag = pack_conv({s}, co, a.nbt, k.KC);
      memcpy(blob.data() + cw.w_off[0], frag.data(), frag.size() * sizeof(float));
      for (int n = 0; n < co; ++n) bias[n] = f.t[n];
    } else if (dot2 != std::string::npos) {  // 1x1 conv2 + bn2 (+ projection shortcut as a second K source)
      const std::string p = nm.substr(0, dot2);
      auto it = m.find(p + ".conv2.weight");
      if (it == m.end() || it->second.shape.size() != 4 || !it->second.data) {
        if (missing->empty()) *missing = p + ".conv2.weight";
        return FPC_E_MISSING_KEY;
      }
      const int co = (int)it->second.shape[0];
      const float* w = it->second.data;
      if ((int)it->second.shape[1] != co || co > cin0p || co > a.nbt * 32) {
        *missing = p + ".conv2.weight";
        return FPC_E_MISSING_KEY;
      }
      if (!fold_bn(m, p + ".bn2", co, &f, missing)) return FPC_E_MISSING_KEY;
      std::vector<PackSource> srcs;
      srcs.push_back({co, cin0p, 1, [&, co](int n, int c_, int) { return (double)w[(size_t)n * co + c_]; }, &f.s});
      for (int n = 0; n < co; ++n) bias[n] = f.t[n];
      const float* wp = nullptr;
      int cip = 0;
      if (a.in1) {
        auto ip = m.find(p + ".identity_downsample.0.weight");
        if (ip == m.end() || ip->second.shape.size() != 4 || (int)ip->second.shape[0] != co || !ip->second.data ||
            (int)ip->second.shape[1] > cin1p) {
          if (missing->empty()) *missing = p + ".identity_downsample.0.weight";
          return FPC_E_MISSING_KEY;
        }
        wp = ip->second.data;
        cip = (int)ip->second.shape[1];
        if (!fold_bn(m, p + ".identity_downsample.1", co, &fp, missing)) return FPC_E_MISSING_KEY;
        srcs.push_back({cip, cin1p, 1, [&, cip](int n, int c_, int) { return (double)wp[(size_t)n * cip + c_]; }, &fp.s});
        for (int n = 0; n < co; ++n) bias[n] += fp.t[n];
      }
      std::vector<float> frag = pack_conv(srcs, co, a.nbt, k.KC);
      memcpy(blob.data() + cw.w_off[0], frag.data(), frag.size() * sizeof(float));
    } else {
      *missing = "internal: unknown op " + nm;
      return FPC_E_INVALID;
    }
    for (int n = 0; n < a.nbt * 32; ++n) blob[cw.b_off + n] = (float)bias[n];
  }
  return FPC_OK;
}

// ---- execution -------------------------------------------------------------------------------
// A call's frames are split into sub-batches that run the whole launch sequence on separate
// HIP streams: workgroups of one sub-batch fill the CUs another leaves idle in the tail of each
// (short) launch, and the latency-bound NMS of one overlaps the MFMA-bound convolutions of the
// other.  Frames are independent, so sub-batches share nothing but the weights.
static hipEvent_t next_event(fpc_ctx* c) {
  if (c->events_used == c->event_pool.size()) {
    hipEvent_t e;
    hipEventCreate(&e);
    c->event_pool.push_back(e);
  }
  return c->event_pool[c->events_used++];
}

struct Sub {
  int f0, n;
  bool small = false;              // the whole call is a few frames: the latency plan (heads side by side, fused layer_in.1)
  bool fuse_softmax = false;       // run_network(which = 1): the detector's last block also does exp-softmax / depth-to-space / threshold
  hipStream_t st;                  // encoder, descriptor head, descriptor sampling
  hipStream_t side = nullptr;      // detector head + NMS, concurrent with the descriptor head
  hipEvent_t ev_enc = nullptr, ev_det = nullptr;
};

struct LaunchTimer {
  fpc_ctx* c;
  int op;
  hipStream_t st;
  int frames;
  hipEvent_t s{}, e{};
  LaunchTimer(fpc_ctx* c_, int op_, hipStream_t st_, int frames_) : c(c_), op(op_), st(st_), frames(frames_) {
    if (c->timing) {
      s = next_event(c);
      e = next_event(c);
      hipEventRecord(s, st);
    }
  }
  ~LaunchTimer() {
    if (c->timing) {
      hipEventRecord(e, st);
      c->timings.push_back({s, e, op, frames});
    }
  }
};

static int op_index(const fpc_ctx* c, OpType t) {
  for (size_t i = 0; i < c->ops.size(); ++i)
    if (c->ops[i].type == t) return (int)i;
  return -1;
}

// which: 0 = encoder, 1 = detector head, 2 = descriptor head
static void run_network(fpc_ctx* c, const float* frames, const Sub& sb0, int which, hipStream_t st) {
  Sub sb = sb0;
  sb.st = st;
  const int H = c->H, W = c->W, n = sb.n, f0 = sb.f0;
  for (size_t i = 0; i < c->ops.size(); ++i) {
    const Op& op = c->ops[i];
    const int br = op.descriptor_branch ? 2 : (op.name.compare(0, 8, "detector") == 0 ? 1 : 0);
    if (br != which) continue;
    if (op.shadow) continue;
    if (op.when == 1 && !sb.small) continue;   // variants of a layer for small / large calls
    if (op.when == 2 && sb.small) continue;
    switch (op.type) {
      case OP_STEM: {
        if (c->fuse_stem_pool || c->cin == 1 || c->bf16) {
          // the pooled map: fp32, completed across tiles with atomicMax (hence zero-filled); FPC_BF16 writes it once, as bf16
          float* x0 = c->bf16 ? reinterpret_cast<float*>(reinterpret_cast<unsigned short*>(c->x0) + (size_t)f0 * (H / 4) * (W / 4) * 64)
                              : c->x0 + (size_t)f0 * (H / 4) * (W / 4) * 64;
          // (only the cells that tiles complete for each other with atomicMax need the zero: kernels_misc.h)
          if (!c->bf16) stem_border_clear_kernel<<<(n * (H / 4) + 7) / 8, 256, 0, sb.st>>>(reinterpret_cast<float4*>(x0), H / 4, W / 4, n * (H / 4));
          LaunchTimer t(c, (int)i, sb.st, n);
          StemPoolArgs a{};
          a.in = frames + (size_t)f0 * c->cin * H * W;
          a.wfrag = c->stem.wfrag;
          a.bias = c->stem.bias;
          a.out = x0;
          a.H = H; a.W = W; a.Ho = H / 2; a.Wo = W / 2; a.Hp = H / 4; a.Wp = W / 4;
          a.tiles_x = c->stem.tiles_x; a.tiles_y = c->stem.tiles_y;
          a.range = c->range + (size_t)f0 * FPC_RANGE_WORDS;
          if (c->split || c->bf16) {
            StemX3Args x{};
            x.in = a.in; x.wfrag = reinterpret_cast<const uint4*>(a.wfrag); x.bias = a.bias; x.out = a.out;
            x.H = H; x.W = W; x.Ho = a.Ho; x.Wo = a.Wo; x.Hp = a.Hp; x.Wp = a.Wp; x.tiles_x = a.tiles_x; x.tiles_y = a.tiles_y;
            x.range_flag = c->split_f16 ? c->status + 1 : nullptr;
#ifdef FPC_DIAG
            x.stamps = nullptr;
            if (const char* e = getenv("FPC_STAMP_OP"))
              if (op.name.find(e) != std::string::npos) {
                if (!c->diag_stamps) hipHostMalloc((void**)&c->diag_stamps, (size_t)65536 * 8 * sizeof(unsigned long long));
                memset(c->diag_stamps, 0, (size_t)65536 * 8 * sizeof(unsigned long long));
                x.stamps = c->diag_stamps;
                c->diag_n = a.tiles_x * a.tiles_y * n;
              }
#endif
            const dim3 grid(a.tiles_x * a.tiles_y * n);
            if (c->split_f16) {
              if (c->cin == 1) hipLaunchKernelGGL((stem_pool_x3_kernel<1, 2>), grid, dim3(256), 0, sb.st, x);
              else hipLaunchKernelGGL((stem_pool_x3_kernel<3, 2>), grid, dim3(256), 0, sb.st, x);
            } else if (c->bf16) {  // bf16 mode: bf16 operands like every other layer of the mode, bf16 out, no atomics
              x.frames = n;
              x.tiles_x = (x.Wp + SB2_PW - 1) / SB2_PW;   // stem_bf16.h: 8 x 7 pooled pixels per tile
              x.tiles_y = (x.Hp + SB2_PH - 1) / SB2_PH;
              const int T = x.tiles_x * x.tiles_y * n;
              const dim3 pg(std::min((T + 7) / 8 * 8, std::max(8, 2 * c->num_cus / 8 * 8)));  // two workgroups per CU, a multiple of 8
              if (c->cin == 1) hipLaunchKernelGGL(stem_bf16_kernel<1>, pg, dim3(SB2_THREADS), StemB2Cfg<1>::LDS_BYTES, sb.st, x);
              else hipLaunchKernelGGL(stem_bf16_kernel<3>, pg, dim3(SB2_THREADS), StemB2Cfg<3>::LDS_BYTES, sb.st, x);
            } else {
              if (c->cin == 1) hipLaunchKernelGGL((stem_pool_x3_kernel<1, 3>), grid, dim3(256), 0, sb.st, x);
              else hipLaunchKernelGGL((stem_pool_x3_kernel<3, 3>), grid, dim3(256), 0, sb.st, x);
            }
          } else if (c->stem2) {
            if (c->cin == 1) hipLaunchKernelGGL(stem_pool2_kernel<1>, dim3(a.tiles_x * a.tiles_y * n), dim3(256), 0, sb.st, a);
            else hipLaunchKernelGGL(stem_pool2_kernel<3>, dim3(a.tiles_x * a.tiles_y * n), dim3(256), 0, sb.st, a);
          } else if (c->cin == 1)
            hipLaunchKernelGGL(stem_pool_kernel<1>, dim3(a.tiles_x * a.tiles_y * n), dim3(256), 0, sb.st, a);
          else
            hipLaunchKernelGGL(stem_pool_kernel<3>, dim3(a.tiles_x * a.tiles_y * n), dim3(256), 0, sb.st, a);
          break;
        }
        LaunchTimer t(c, (int)i, sb.st, n);
        StemArgs a = c->stem;
        a.in = frames + (size_t)f0 * 3 * H * W;
        a.out = c->stem_out + (size_t)f0 * (H / 2) * (W / 2) * 64;
        hipLaunchKernelGGL(stem_kernel, dim3(a.tiles_x * a.tiles_y * n), dim3(256), 0, sb.st, a);
        break;
      }
      case OP_POOL: {
        if (c->fuse_stem_pool || c->cin == 1 || c->bf16) break;
        LaunchTimer t(c, (int)i, sb.st, n);
        const size_t total = (size_t)n * (H / 4) * (W / 4) * 16;
        hipLaunchKernelGGL(maxpool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, sb.st,
                           reinterpret_cast<const float4*>(c->stem_out + (size_t)f0 * (H / 2) * (W / 2) * 64),
                           reinterpret_cast<float4*>(c->x0 + (size_t)f0 * (H / 4) * (W / 4) * 64), n, H / 2, W / 2, H / 4,
                           W / 4);
        break;
      }
      case OP_WBLOCK: {
        LaunchTimer t(c, (int)i, sb.st, n);
        WBlockArgs a = op.wargs;
        a.frame0 = f0;
#ifdef FPC_DIAG
        a.stamps = nullptr;
        if (const char* e = getenv("FPC_STAMP_OP"))
          if (op.name.find(e) != std::string::npos) {
            if (!c->diag_stamps) hipHostMalloc((void**)&c->diag_stamps, (size_t)65536 * 8 * sizeof(unsigned long long));
            memset(c->diag_stamps, 0, (size_t)65536 * 8 * sizeof(unsigned long long));
            a.stamps = c->diag_stamps;
            c->diag_n = a.tiles_x * a.tiles_y * n;
          }
#endif
        a.total = a.tiles_x * a.tiles_y * n;
        a.xcd_order = c->xcd_order ? 1 : 0;
        {  // range of the input's buffer descriptor: frames 0 .. f0 + n - 1 of the tensor, from `x` (already offset to its first channel)
          const unsigned long long xb = (unsigned long long)(f0 + n) * a.H * a.W * a.csx * sizeof(float);
          a.x_bytes = xb > 0xfffffff0ull ? 0xfffffff0u : (unsigned)xb;
        }
        // persistent (one workgroup per CU walking the tiles) when a workgroup gets enough tiles to
        // amortise; otherwise one workgroup per tile
        int grid = (c->persist_min_tiles > 0 && a.total >= c->persist_min_tiles * c->num_cus) ? c->num_cus : a.total;
        if (g_wkinds[op.wkind].gen == 3) {
          // Pointers to the launch's first frame, as many frames per launch as stay below W36_MARKER bytes (all of them,
          // except at the C++ network's full-resolution layers: 78.6 MB per VGA frame and tensor).
          const int fpl = w36_frames_per_launch(a.H, a.W, a.csx, a.cso);
          const float* x0 = a.x;
          float* o0 = a.out;
          for (int g0 = 0; g0 < n; g0 += fpl) {
            const int gn = std::min(fpl, n - g0);
            a.frame0 = 0;
            a.x = x0 + (size_t)(f0 + g0) * a.H * a.W * a.csx;
            a.out = o0 + (size_t)(f0 + g0) * a.H * a.W * a.cso;
            a.total = a.tiles_x * a.tiles_y * gn;
            a.x_bytes = (unsigned)((unsigned long long)gn * a.H * a.W * a.csx * sizeof(float));
            // One workgroup per CU (137 KB of LDS), so a launch lasts ceil(tiles / CUs) tile times whatever the grid: take
            // the SMALLEST grid that still finishes in that many rounds (a multiple of 8: the tile walk is per XCD) and
            // leave the other CUs to the launches of the other sub-batch's stream -- 640 tiles: 216 workgroups x 3 rounds
            // instead of 256 of which 128 idle through the third; 320 tiles: 160 x 2.
            // (a launch of gridDim.y parts shares the CUs between them: each part's tile walk gets CUs / parts)
            const int cus_cap = c->w36_cus > 0 ? std::min(c->num_cus, c->w36_cus) : c->num_cus;
            const int cus8 = std::max(1, cus_cap / 8 / std::max(1, op.grid_y)), per_xcd = (a.total + 7) / 8;
            const int rounds = (per_xcd + cus8 - 1) / cus8;
            const int grid3 = 8 * std::min(cus8, (per_xcd + rounds - 1) / rounds);
            g_wkinds[op.wkind].launch(a, dim3(std::min(a.total, grid3), op.grid_y), sb.st);
          }
          break;
        }
        g_wkinds[op.wkind].launch(a, dim3(std::min(a.total, grid), op.grid_y), sb.st);
        break;
      }
      case OP_BLOCK: {
        LaunchTimer t(c, (int)i, sb.st, n);
        BlockArgs a = op.bargs;
        a.frame0 = f0;
#ifdef FPC_DIAG
        a.stamps = nullptr;
        if (const char* e = getenv("FPC_STAMP_OP"))
          if (op.name.find(e) != std::string::npos) {
            if (!c->diag_stamps) hipHostMalloc((void**)&c->diag_stamps, (size_t)65536 * 8 * sizeof(unsigned long long));
            memset(c->diag_stamps, 0, (size_t)65536 * 8 * sizeof(unsigned long long));
            a.stamps = c->diag_stamps;
            c->diag_n = a.tiles_x * a.tiles_y * n;
          }
#endif
        g_bkinds[op.bkind].launch(a, dim3(a.tiles_x * a.tiles_y * n), sb.st);
        break;
      }
      case OP_BF16: {
        LaunchTimer t(c, (int)i, sb.st, n);
        BlockBfArgs a = op.fargs;
        a.frame0 = f0;
        a.range_flag = c->split_f16 ? c->status + 1 : nullptr;
#ifdef FPC_DIAG
        a.stamps = nullptr;
        if (const char* e = getenv("FPC_STAMP_OP"))
          if (op.name.find(e) != std::string::npos) {
            if (!c->diag_stamps) hipHostMalloc((void**)&c->diag_stamps, (size_t)65536 * 8 * sizeof(unsigned long long));
            memset(c->diag_stamps, 0, (size_t)65536 * 8 * sizeof(unsigned long long));
            a.stamps = c->diag_stamps;
            c->diag_n = a.tiles_x * a.tiles_y * n;
          }
#endif
        if (g_fkinds[op.fkind].planes == 1) {   // block_bf16_kernel: persistent grid, a multiple of 8 (block_bf16.h)
          a.total_tiles = a.tiles_x * a.tiles_y * n;
          if (sb.fuse_softmax && op.fused_softmax_capable) {
            const size_t HW = (size_t)c->H * c->W;
            hipMemsetAsync(c->ncand + f0, 0, sizeof(int32_t) * n, sb.st);
            a.softmax = 1;
            a.thresh = c->cfg.conf_thresh;
            a.nmsmap = c->nmsmap + f0 * HW;
            a.cand = c->cand + f0 * HW;
            a.ncand = c->ncand + f0;
          }
          // (op.grid_y parts -- convt_bf16_kernel's output-channel halves -- share the CUs)
          const int g = std::min((a.total_tiles + 7) / 8 * 8, std::max(8, c->fkind_blocks_per_cu[op.fkind] * c->num_cus / op.grid_y / 8 * 8));
          g_fkinds[op.fkind].launch(a, dim3(g, op.grid_y), sb.st);
        } else {
          g_fkinds[op.fkind].launch(a, dim3(a.tiles_x * a.tiles_y * n), sb.st);
        }
        break;
      }
      case OP_VCONV0: {
        LaunchTimer t(c, (int)i, sb.st, n);
        VggConv0Args a{};
        a.in = frames;
        a.w = c->blob + c->vconv0_off;
        a.out = op.pout;
        a.H = H; a.W = W; a.frame0 = f0;
        hipLaunchKernelGGL(vgg_conv0_kernel, dim3((unsigned)((size_t)n * H * W / 64)), dim3(256), 0, sb.st, a);
        break;
      }
      case OP_POOL2: {
        LaunchTimer t(c, (int)i, sb.st, n);
        const size_t tot = (size_t)n * (op.pH / 2) * (op.pW / 2) * (op.pC / 4);
        hipLaunchKernelGGL(maxpool2_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, sb.st,
                           reinterpret_cast<const float4*>(op.pin), reinterpret_cast<float4*>(op.pout), n, op.pH, op.pW,
                           op.pC / 4, f0);
        break;
      }
      case OP_L2NORM: {
        LaunchTimer t(c, (int)i, sb.st, n);
        const size_t npix = (size_t)n * c->Hc * c->Wc;
        hipLaunchKernelGGL(l2norm256_kernel, dim3((unsigned)((npix + 3) / 4)), dim3(256), 0, sb.st,
                           op.pout + (size_t)f0 * c->Hc * c->Wc * 256, npix);
        break;
      }
      case OP_CONV: {
        LaunchTimer t(c, (int)i, sb.st, n);
        ConvArgs a = op.args;
        a.frame0 = f0;
        if (op.lean) lean_kind(op.kind)->launch(a, dim3(a.tiles_x * a.tiles_y * n, op.grid_y, op.grid_z), sb.st);
        else g_kinds[op.kind].launch(a, dim3(a.tiles_x * a.tiles_y * n, op.grid_y, op.grid_z), sb.st);
        break;
      }
      default:
        break;  // post-processing ops are issued by the callers
    }
  }
}

static Sub on(const Sub& sb, hipStream_t st) {
  Sub r = sb;
  r.st = st;
  return r;
}

static void run_softmax(fpc_ctx* c, const Sub& sb, bool dense_map = true) {
  const size_t HW = (size_t)c->H * c->W;
  hipMemsetAsync(c->ncand + sb.f0, 0, sizeof(int32_t) * sb.n, sb.st);
  LaunchTimer t(c, op_index(c, OP_SOFTMAX), sb.st, sb.n);
  // (FPC_BF16: the arithmetic of that mode's fused epilogue -- v_exp_f32, one reciprocal per cell; nms_word.h)
  const auto kern = c->bf16 ? softmax_d2s_kernel<true> : softmax_d2s_kernel<false>;
  hipLaunchKernelGGL(kern, dim3(sb.n * c->Hc), dim3(256), (size_t)12 * c->W * sizeof(float), sb.st,
                     c->lg + (size_t)sb.f0 * c->Hc * c->Wc * c->lgcs, c->lgcs, sb.n, c->Hc, c->Wc, c->cfg.conf_thresh,
                     dense_map ? c->prob + sb.f0 * HW : nullptr, c->nmsmap + sb.f0 * HW, c->cand + sb.f0 * HW, c->ncand + sb.f0,
                     c->rowmax + (size_t)sb.f0 * c->Hc);
}

static void run_nms(fpc_ctx* c, const Sub& sb) {
  LaunchTimer t(c, op_index(c, OP_NMS), sb.st, sb.n);
  const size_t HW = (size_t)c->H * c->W;
  const int n = sb.n, f0 = sb.f0;
  NmsArgs a{};
  a.nmsmap = c->nmsmap + f0 * HW; a.cand = c->cand + f0 * HW; a.ncand = c->ncand + f0;
  a.sort_scratch = c->sort_scratch + (size_t)f0 * c->sort_cap;
  a.sort_cap = c->sort_cap;
  const bool chunked = !c->nms_one_workgroup;
  a.aux = chunked ? c->nms_aux + (size_t)f0 * NMS_AUX_INTS : nullptr;
  a.H = c->H; a.W = c->W; a.r = c->cfg.nms_dist; a.border = c->cfg.border_remove; a.cap = c->cap;
  a.count = c->count + f0; a.xy = c->xy + (size_t)f0 * c->cap * 2; a.conf = c->conf + (size_t)f0 * c->cap;
  a.status = c->status;
  a.max_rounds = c->H * c->W;
  // enough workgroups that a typical frame (a few thousand candidates) has about one candidate
  // per thread; the second launch normally finds nothing left to do
  const int G = c->nms_g > 0 ? c->nms_g : std::max(1, std::min(16, 512 / n));
  for (int pass = 0; pass < c->nms_passes; ++pass) {
    if (c->cfg.nms_dist == 4)
      hipLaunchKernelGGL(nms_rounds_kernel<4>, dim3(G, n), dim3(NMS_ROUNDS_THREADS), 0, sb.st, a);
    else
      hipLaunchKernelGGL(nms_rounds_kernel<0>, dim3(G, n), dim3(NMS_ROUNDS_THREADS), 0, sb.st, a);
  }
  if (chunked) {
    // the leftovers of the round launches settled by one workgroup per frame, then slices of the candidate list sorted
    // by one workgroup each and merged by rank (kernels_misc.h); nms_sort_kernel leaves at once unless a slice had
    // more survivors than LDS holds
    const bool big = HW > 400000;
    const int GC = big ? 8 : 4;  // a frame of a few thousand candidates has this many slices
    hipLaunchKernelGGL(nms_finish_kernel, dim3(n), dim3(NMS_FINISH_THREADS), 0, sb.st, a);
    if (big) {
      hipLaunchKernelGGL(nms_chunk_sort_kernel<4096>, dim3(GC, n), dim3(1024), 4096 * sizeof(unsigned long long), sb.st, a);
      hipLaunchKernelGGL(nms_merge_kernel<4096>, dim3(GC, n), dim3(1024), 4096 * sizeof(unsigned long long), sb.st, a);
    } else {
      hipLaunchKernelGGL(nms_chunk_sort_kernel<2048>, dim3(GC, n), dim3(1024), 2048 * sizeof(unsigned long long), sb.st, a);
      hipLaunchKernelGGL(nms_merge_kernel<2048>, dim3(GC, n), dim3(1024), 2048 * sizeof(unsigned long long), sb.st, a);
    }
  }
  hipLaunchKernelGGL(nms_sort_kernel, dim3(n), dim3(1024), chunked ? 0 : NMS_LDS_KEYS * sizeof(unsigned long long), sb.st, a);
}

static void run_desc(fpc_ctx* c, const Sub& sb, const float* dmap_nhwc) {
  LaunchTimer t(c, op_index(c, OP_DESC), sb.st, sb.n);
  // persistent grid, a multiple of 8 (kernels_misc.h): eight workgroups of four waves per CU
  const int by_xcd = sb.n >= 8;
  const int G = std::max(8, c->num_cus * 8 / 8 * 8);
  if (c->D == 256)
    hipLaunchKernelGGL(descriptor16_kernel<16>, dim3(G), dim3(256), 0, sb.st,
                       dmap_nhwc + (size_t)sb.f0 * c->Hc * c->Wc * 256, 256, c->Hc, c->Wc, c->H, c->W, c->count + sb.f0,
                       c->xy + (size_t)sb.f0 * c->cap * 2, c->cap, c->desc_out + (size_t)sb.f0 * c->cap * 256, sb.n, by_xcd);
  else if (c->bf16 && dmap_nhwc == c->desc_map)   // FPC_BF16's own map is bf16: half the elements' bytes per frame
    hipLaunchKernelGGL((descriptor16_kernel<8, true>), dim3(G), dim3(256), 0, sb.st,
                       reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(dmap_nhwc) + (size_t)sb.f0 * c->Hc * c->Wc * 128),
                       128, c->Hc, c->Wc, c->H, c->W, c->count + sb.f0,
                       c->xy + (size_t)sb.f0 * c->cap * 2, c->cap, c->desc_out + (size_t)sb.f0 * c->cap * 128, sb.n, by_xcd);
  else
    hipLaunchKernelGGL(descriptor16_kernel<8>, dim3(G), dim3(256), 0, sb.st,
                       dmap_nhwc + (size_t)sb.f0 * c->Hc * c->Wc * 128, 128, c->Hc, c->Wc, c->H, c->W, c->count + sb.f0,
                       c->xy + (size_t)sb.f0 * c->cap * 2, c->cap, c->desc_out + (size_t)sb.f0 * c->cap * 128, sb.n, by_xcd);
}

// Splits [0,n) over the ctx's streams; aux streams fork from / join into the main stream.
template <typename F>
static int for_each_sub(fpc_ctx* c, int n, F&& body) {
  // fpc_set_timing(n > 1): every n-th pass over a batch carries the events -- decided HERE, for every entry point alike
  // (fpc_detect, fpc_forward, fpc_detect_u8*, each network pass of fpc_homography_adaptation)
  if (c->timing_every > 1) c->timing = c->timing_calls++ % c->timing_every == 0;
  int parts = std::min<int>((int)c->aux.size() + 1, std::max(1, n / c->min_sub));
  if (parts > 1) HIPCHECK(hipEventRecord(c->ev_fork, c->stream));
  int f0 = 0;
  for (int p = 0; p < parts; ++p) {
    const int cnt = n / parts + (p < n % parts ? 1 : 0);
    Sub sb;
    sb.f0 = f0;
    sb.n = cnt;
    sb.small = n < 2 * c->min_sub;
    sb.st = p == 0 ? c->stream : c->aux[p - 1];
    sb.side = c->side[p];
    // The runtime multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES) in creation order, and two
    // streams on one queue run in order: with main, aux1, aux2, side0 the side stream shared the main stream's queue
    // and a single frame's heads no longer overlapped (0.81 instead of 0.62 ms in the split mode).  A call of a few
    // frames has one sub-batch, so the first aux stream -- created right behind the main stream -- is idle: use it.
    if (sb.small && !c->aux.empty()) sb.side = c->aux[0];
    sb.ev_enc = c->ev_enc[p];
    sb.ev_det = c->ev_det[p];
    if (p) HIPCHECK(hipStreamWaitEvent(sb.st, c->ev_fork, 0));
    body(sb);
    if (p) {
      HIPCHECK(hipEventRecord(c->ev_join[p - 1], sb.st));
      HIPCHECK(hipStreamWaitEvent(c->stream, c->ev_join[p - 1], 0));
    }
    f0 += cnt;
  }
  HIPCHECK(hipGetLastError());
  return FPC_OK;
}

// The whole path for one sub-batch.  The detector head and its post-processing (latency-bound,
// few CUs) run on a side stream next to the descriptor head (MFMA-bound); both only need the
// encoder output.  `upto`: 0 = dense maps only (fpc_forward), 1 = keypoints + descriptors.
static void run_path(fpc_ctx* c, const float* frames, const Sub& sb0, bool de, int upto) {
  // fpc_detect in FPC_BF16: the detector's last block writes the NMS state map and the candidate lists itself (the
  // logits and the dense probability map, which only fpc_forward's callers see, are not produced)
  Sub sb = sb0;
  for (const Op& op : c->ops) sb.fuse_softmax |= upto && op.fused_softmax_capable;
  c->logits_valid = !sb.fuse_softmax;
  hipMemsetAsync(c->range + (size_t)sb.f0 * FPC_RANGE_WORDS, 0, sizeof(uint32_t) * FPC_RANGE_WORDS * sb.n, sb.st);
  run_network(c, frames, sb, 0, sb.st);
  if (de && upto && c->nms_aside && !sb.small && sb.side) {
    // detector head and softmax in line; the (latency-bound, few-CU) NMS on the side stream next to the descriptor head
    run_network(c, frames, sb, 1, sb.st);
    if (!sb.fuse_softmax) run_softmax(c, sb, !upto);
    hipEventRecord(sb.ev_enc, sb.st);
    hipStreamWaitEvent(sb.side, sb.ev_enc, 0);
    run_nms(c, on(sb, sb.side));
    hipEventRecord(sb.ev_det, sb.side);
    run_network(c, frames, sb, 2, sb.st);
    hipStreamWaitEvent(sb.st, sb.ev_det, 0);
    run_desc(c, sb, c->desc_map);
    return;
  }
  if (!de || !(c->split_heads || sb.small) || !sb.side) {
    run_network(c, frames, sb, 1, sb.st);
    if (!sb.fuse_softmax) run_softmax(c, sb, !upto);
    if (upto) run_nms(c, sb);
    if (de) {
      run_network(c, frames, sb, 2, sb.st);
      if (upto) run_desc(c, sb, c->desc_map);
    }
    return;
  }
  hipEventRecord(sb.ev_enc, sb.st);
  hipStreamWaitEvent(sb.side, sb.ev_enc, 0);
  const Sub det = on(sb, sb.side);
  run_network(c, frames, sb, 1, sb.side);
  if (!sb.fuse_softmax) run_softmax(c, det, !upto);
  if (upto) run_nms(c, det);
  hipEventRecord(sb.ev_det, sb.side);
  run_network(c, frames, sb, 2, sb.st);
  hipStreamWaitEvent(sb.st, sb.ev_det, 0);
  if (upto) run_desc(c, sb, c->desc_map);
}

}  // namespace fpc

// ====================================================================================
// C-ABI
// ====================================================================================
extern "C" {

int fpc_abi_version(void) { return FPC_ABI_VERSION; }
int fpc_pack_layout_revision(void) { return FPC_PACK_LAYOUT_REVISION; }

// What this binary was compiled with: the target, whether it is the diagnostic build (in-kernel stamps: never the
// product), and that no experiment switch of earlier rounds reached it (they are gone from the headers; a build
// that defines one of their names anyway does not compile).
#if defined(STEMB_SKIP_LOAD) || defined(STEMB_SKIP_K) || defined(STEMB_SKIP_TILE) || defined(STEMB_SKIP_POOL) || \
    defined(STEMB_SKIP_STORE) || defined(W16_YOUNG_PRIO) || defined(W16_PRIO_SPLIT) || defined(FPC_NO_PIN)
#error "ablation switches are not part of the library: see experiments/harness/"
#endif
const char* fpc_build_flags(void) {
  return "arch=gfx950"
#ifdef FPC_DIAG
         ";diag=1"
#else
         ";diag=0"
#endif
         ";ablations=none";
}

const char* fpc_strerror(int code) {
  switch (code) {
    case FPC_OK: return "ok";
    case FPC_E_INVALID: return "invalid argument or unsupported geometry";
    case FPC_E_NO_DEVICE: return "no usable HIP device (this library has no CPU fallback)";
    case FPC_E_HIP: return "HIP runtime error (see fpc_last_hip_error)";
    case FPC_E_NO_WEIGHTS: return "weights not loaded";
    case FPC_E_MISSING_KEY: return "checkpoint entry missing or of the wrong shape (see fpc_last_hip_error)";
    case FPC_E_CAPACITY: return "caller buffer too small";
    case FPC_E_NOT_CONVERGED: return "NMS round limit hit";
    case FPC_E_RANGE: return "value outside fp16's range in FPC_F32_SPLIT_F16 mode (use FPC_F32_SPLIT or FPC_F32)";
    case FPC_E_NONFINITE: return "a frame of the call holds a NaN or Inf pixel (see fpc_last_hip_error for which)";
    default: return "unknown error";
  }
}

const char* fpc_last_hip_error(void) { return g_hip_err.c_str(); }

int fpc_default_config(fpc_config* cfg) {
  if (!cfg) return FPC_E_INVALID;
  memset(cfg, 0, sizeof(*cfg));
  cfg->device = 0;
  cfg->height = 480;
  cfg->width = 640;
  cfg->max_batch = 1;
  cfg->cell = 8;
  cfg->nms_dist = 4;
  cfg->conf_thresh = 0.015f;
  cfg->border_remove = 4;
  cfg->descriptor_enabled = 1;
  cfg->max_keypoints = 0;
  return FPC_OK;
}

// ------------------------------------------------------------------------------------
// Streams on distinct hardware queues: queue_map.h (anchors, device timestamps, a registry of the live contexts' streams)
// ------------------------------------------------------------------------------------
enum { SLOT_MAIN = 0, SLOT_AUX = 1, SLOT_SIDE = 100, SLOT_UPLOAD = 200 };

// Dynamic-LDS limits of every kernel instance and the occupancy table of the bf16 / split-operand instances.
static int prepare_kernels(fpc_ctx* c, const fpc_config* cfg) {
  for (int k = 0; k < K_COUNT; ++k)
    HIPCHECK(hipFuncSetAttribute(g_kinds[k].fn, hipFuncAttributeMaxDynamicSharedMemorySize, g_kinds[k].lds_bytes));
  for (const LeanKindInfo& l : g_lean_kinds) HIPCHECK(hipFuncSetAttribute(l.fn, hipFuncAttributeMaxDynamicSharedMemorySize, l.lds_bytes));
  for (int k = 0; k < WK_COUNT; ++k)
    HIPCHECK(hipFuncSetAttribute(g_wkinds[k].fn, hipFuncAttributeMaxDynamicSharedMemorySize, g_wkinds[k].lds_bytes));
  for (int k = 0; k < BK_COUNT; ++k)
    HIPCHECK(hipFuncSetAttribute(g_bkinds[k].fn, hipFuncAttributeMaxDynamicSharedMemorySize, g_bkinds[k].lds_bytes));
  static_assert(FK_COUNT <= 64, "fpc_ctx::fkind_blocks_per_cu");
  for (int k = 0; k < FK_COUNT; ++k) {
    HIPCHECK(hipFuncSetAttribute(g_fkinds[k].fn, hipFuncAttributeMaxDynamicSharedMemorySize, g_fkinds[k].lds_bytes));
    int nb = 0;
    HIPCHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, g_fkinds[k].fn, g_fkinds[k].WM * g_fkinds[k].WN * 64, g_fkinds[k].lds_bytes));
    c->fkind_blocks_per_cu[k] = std::max(1, nb);
#ifdef FPC_DIAG
    hipFuncAttributes fa{};
    hipFuncGetAttributes(&fa, g_fkinds[k].fn);
    fprintf(stderr, "[diag] %s: lds %d B, regs %d, scratch %zu, occupancy %d blocks/CU\n", g_fkinds[k].name, g_fkinds[k].lds_bytes, fa.numRegs, fa.localSizeBytes, nb);
#endif
  }
  HIPCHECK(hipFuncSetAttribute((const void*)softmax_d2s_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               12 * cfg->width * (int)sizeof(float)));
  HIPCHECK(hipFuncSetAttribute((const void*)softmax_d2s_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               12 * cfg->width * (int)sizeof(float)));
  HIPCHECK(hipFuncSetAttribute((const void*)stem_bf16_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, StemB2Cfg<1>::LDS_BYTES));
  HIPCHECK(hipFuncSetAttribute((const void*)stem_bf16_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, StemB2Cfg<3>::LDS_BYTES));
  HIPCHECK(hipFuncSetAttribute((const void*)nms_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                               NMS_LDS_KEYS * (int)sizeof(unsigned long long)));
#ifdef FPC_DIAG
  for (int k = 0; k < BK_COUNT; ++k) {
    int nb = -1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, g_bkinds[k].fn, g_bkinds[k].WM * g_bkinds[k].WN * 64, g_bkinds[k].lds_bytes);
    hipFuncAttributes fa{};
    hipFuncGetAttributes(&fa, g_bkinds[k].fn);
    fprintf(stderr, "[diag] %s: lds %d B, regs %d, static lds %zu, occupancy %d blocks/CU\n", g_bkinds[k].name, g_bkinds[k].lds_bytes, fa.numRegs, fa.sharedSizeBytes, nb);
  }
  for (int k = 0; k < K_COUNT; ++k) {
    int nb = -1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, g_kinds[k].fn, g_kinds[k].threads, g_kinds[k].lds_bytes);
    fprintf(stderr, "[diag] %s: lds %d B, occupancy %d blocks/CU\n", g_kinds[k].name, g_kinds[k].lds_bytes, nb);
  }
#endif
  return FPC_OK;
}

int fpc_create(fpc_ctx** out, const fpc_config* cfg) {
  if (!out || !cfg) return FPC_E_INVALID;
  *out = nullptr;
  // the descriptor head halves the 1/8 map and doubles it again (superpoint.py:43-59): odd
  // H/8 or W/8 breaks its concat, so frames must be multiples of 16 unless it is disabled
  const int mult = (cfg->descriptor_enabled && cfg->arch != FPC_ARCH_VGG) ? 16 : 8;
  if (cfg->in_channels != 0 && cfg->in_channels != 1 && cfg->in_channels != 3) return FPC_E_INVALID;
  if (cfg->dtype < FPC_F32 || cfg->dtype > FPC_F32_SPLIT_F16) return FPC_E_INVALID;
  if (cfg->arch != FPC_ARCH_RESNET && cfg->arch != FPC_ARCH_VGG) return FPC_E_INVALID;
  // the C++ network takes one gray plane (cpp/src/settings.h:19) and has no bf16 plan
  if (cfg->arch == FPC_ARCH_VGG && (cfg->in_channels != 1 || cfg->dtype == FPC_BF16)) return FPC_E_INVALID;
  // conf_thresh: finite and >= 0.  The maps are probabilities (>= 0), so a negative threshold means the same as 0; it
  // is refused rather than silently clamped.  (The reference compares `prob >= thresh` with any float: netutils.py:59.)
  if (!(cfg->conf_thresh >= 0.f) || !(cfg->conf_thresh <= 3.0e38f)) return FPC_E_INVALID;
  if (cfg->cell != 8 || cfg->height < 16 || cfg->width < 16 || cfg->height % mult || cfg->width % mult ||
      cfg->max_batch < 1 || cfg->nms_dist < 0 || cfg->nms_dist > 64 || cfg->border_remove < 0 ||
      (long long)cfg->height * cfg->width >= (1ll << 30) ||
      // the kernels address a tensor of the whole batch with 32-bit byte offsets (the widest: 16 bytes per frame pixel)
      (long long)cfg->max_batch * cfg->height * cfg->width >= (1ll << 28) ||
      cfg->width > 3328)  // softmax_d2s_kernel keeps an 8 x W strip (48 W bytes) in the 160 KB of LDS
    return FPC_E_INVALID;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) {
    g_hip_err = "hipGetDeviceCount: no device";
    return FPC_E_NO_DEVICE;
  }
  HIPCHECK(hipSetDevice(cfg->device));
  std::unique_ptr<fpc_ctx> c(new fpc_ctx());
  {
    hipDeviceProp_t prop;
    HIPCHECK(hipGetDeviceProperties(&prop, cfg->device));
    c->num_cus = std::max(1, prop.multiProcessorCount);
  }
  c->cfg = *cfg;
  c->H = cfg->height;
  c->W = cfg->width;
  c->B = cfg->max_batch;
  c->Hc = c->H / 8;
  c->Wc = c->W / 8;
  c->cin = cfg->in_channels == 1 ? 1 : 3;
  c->vgg = cfg->arch == FPC_ARCH_VGG;
  c->D = c->vgg ? 256 : 128;
  c->bf16 = cfg->dtype == FPC_BF16;
  c->split = cfg->dtype == FPC_F32_SPLIT || cfg->dtype == FPC_F32_SPLIT_F16;
  c->split_f16 = cfg->dtype == FPC_F32_SPLIT_F16;
  c->lgcs = (c->bf16 || c->split) ? 80 : 72;
  // kept points are pairwise > nms_dist apart (infinity norm): at most one per (r+1)^2 cell
  const int r1 = cfg->nms_dist + 1;
  const int worst = ((c->H + r1 - 1) / r1) * ((c->W + r1 - 1) / r1);
  c->cap = cfg->max_keypoints > 0 ? cfg->max_keypoints : worst;
  c->sort_cap = 1;
  while (c->sort_cap < worst) c->sort_cap <<= 1;
  // Kernel attributes first: the process's first call loads the code object here (~150 ms), so that the stream placement
  // below -- whose first probe launch would otherwise trigger that load -- reports its own cost (fpc_stream_report).
  {
    const int prc = prepare_kernels(c.get(), cfg);
    if (prc != FPC_OK) return prc;      // (nothing to give back yet: no stream, no event, no allocation)
  }
  // Streams: placed on hardware queues of their own (queue_map.h).  The registry's lock is held while this context's
  // streams are chosen and registered, and RELEASED before anything that can fail and call fpc_destroy (round 4 held it
  // to the end of the function: a failing build_plan then deadlocked on fpc_destroy's own lock).
  const bool qprobe = qmap::probe_mode() != 0;
  c->queue_probe = qprobe;
  const auto create_t0 = std::chrono::steady_clock::now();
  std::unique_lock<std::mutex> queue_lock(qmap::state().mu);
  const int probe_rounds0 = qmap::state().dev[cfg->device].rounds;
  c->stream = qmap::acquire(cfg->device, c.get(), SLOT_MAIN, true, qprobe, true);
  if (!c->stream) { qmap::release_owner(c.get()); g_hip_err = "hipStreamCreateWithFlags"; return FPC_E_HIP; }
  c->own_stream = true;
  {
    // Launch-plan knobs: fpc_config fields (include/fpc.h, FPC_PLAN_*) first, then the FPC_* environment variables as
    // overrides for A/B runs of an unmodified caller.
    int nsub = cfg->num_streams > 0 ? std::min(8, cfg->num_streams) : (c->split ? 3 : 2);   // measured: 2 sub-batches for the fp32-MFMA kernels, 3 for the (shorter) split-operand ones
    const unsigned pf = cfg->plan_flags;
    c->fuse_blocks = !(pf & FPC_PLAN_NO_FUSED_BLOCKS);
    c->guard_zones = (pf & FPC_PLAN_GUARD_ZONES) != 0;
    if (const char* e = getenv("FPC_GUARD_ZONES")) c->guard_zones = atoi(e) != 0;      // (a whole test run under the canary zones)
    c->winograd_det_gen3 = !(pf & FPC_PLAN_DETECTOR_GEN1);
    c->w36_paired = !(pf & FPC_PLAN_W36_ONE_WAVE);
    if (const char* e = getenv("FPC_W36_PAIRED")) c->w36_paired = atoi(e) != 0;
    if (const char* e = getenv("FPC_W36_CUS")) c->w36_cus = std::max(0, atoi(e));
    if (const char* e = getenv("FPC_WINOGRAD_DET_GEN")) c->winograd_det_gen3 = atoi(e) >= 3;
    c->winograd = !(pf & FPC_PLAN_NO_WINOGRAD);
    c->winograd_det = !(pf & FPC_PLAN_NO_WINOGRAD_DETECTOR);
    c->winograd_in1 = !(pf & FPC_PLAN_NO_WINOGRAD_LAYER_IN1);
    c->xcd_order = !(pf & FPC_PLAN_NO_XCD_ORDER);
    c->fuse_stem_pool = !(pf & FPC_PLAN_NO_FUSED_STEM_POOL);
    c->stem_lean = !(pf & FPC_PLAN_STEM_ROUND3);
    c->conv_lean = !(pf & FPC_PLAN_CONV_ROUND1);
    if (const char* e = getenv("FPC_CONV_LEAN")) c->conv_lean = atoi(e) != 0;
    if (const char* e = getenv("FPC_STEM_LEAN")) c->stem_lean = atoi(e) != 0;
    // Round 4, with the streams on hardware queues of their own: one context, two sub-batches, Python network, fp32 MFMA:
    // 10 740 -> 10 845 frames/s (steady 10 790 -> 10 930; two runs each); the C++ network loses 2 %, bf16 HD and the QVGA
    // detector do not move, one-sub-batch contexts (the bench's two in turn) +0.1 %: the default where it was measured to pay.
    c->split_heads = (pf & FPC_PLAN_SPLIT_HEADS) != 0;   // (and by default where it pays: below, once nsub is final)
    if (pf & FPC_PLAN_NO_PERSISTENT_GRID) c->persist_min_tiles = 0;
    c->layer1_t816 = (pf & FPC_PLAN_LAYER1_TILE_8x16) != 0;
    c->winograd_gen = (pf & FPC_PLAN_WINOGRAD_GEN1) ? 1 : (pf & FPC_PLAN_WINOGRAD_GEN2) ? 2 : 3;
    if (const char* e = getenv("FPC_WINOGRAD_GEN")) c->winograd_gen = std::min(3, std::max(1, atoi(e)));
    c->latency_tiles = !(pf & FPC_PLAN_NO_LATENCY_TILES);
    if (const char* e = getenv("FPC_LATENCY_TILES")) c->latency_tiles = atoi(e) != 0;
    c->fuse_softmax = !(pf & FPC_PLAN_NO_FUSED_SOFTMAX);
    c->convt_fused = !(pf & FPC_PLAN_CONVT_PHASES);
    if (const char* e = getenv("FPC_CONVT_FUSED")) c->convt_fused = atoi(e) != 0;
    if (const char* e = getenv("FPC_FUSE_SOFTMAX")) c->fuse_softmax = atoi(e) != 0;
    c->nms_one_workgroup = (pf & FPC_PLAN_NMS_ONE_WORKGROUP) != 0;
    if (const char* e = getenv("FPC_NMS_CHUNKED")) c->nms_one_workgroup = atoi(e) == 0;
    if (cfg->min_sub_batch > 0) c->min_sub = cfg->min_sub_batch;
    if (cfg->nms_round_launches > 0) c->nms_passes = std::min(64, cfg->nms_round_launches);
    else if (cfg->nms_round_launches < 0) c->nms_passes = 0;
    if (const char* e = getenv("FPC_STREAMS")) nsub = std::max(1, std::min(8, atoi(e)));
    if (!(pf & FPC_PLAN_HEADS_IN_LINE) && nsub >= 2 && !c->vgg && !c->bf16 && !c->split) c->split_heads = true;
    if (const char* e = getenv("FPC_FUSE")) c->fuse_blocks = atoi(e) != 0;
    if (const char* e = getenv("FPC_WINOGRAD")) c->winograd = atoi(e) != 0;
    if (const char* e = getenv("FPC_XCD_ORDER")) c->xcd_order = atoi(e) != 0;
    if (const char* e = getenv("FPC_MIN_SUB")) c->min_sub = std::max(1, atoi(e));
    if (const char* e = getenv("FPC_PERSIST_MIN")) c->persist_min_tiles = atoi(e);
    if (const char* e = getenv("FPC_WINOGRAD_DET")) c->winograd_det = atoi(e) != 0;
    if (const char* e = getenv("FPC_WINOGRAD_IN1")) c->winograd_in1 = atoi(e) != 0;
    if (const char* e = getenv("FPC_FUSE_STEM")) c->fuse_stem_pool = atoi(e) != 0;
    if (const char* e = getenv("FPC_NMS_PASSES")) c->nms_passes = std::max(0, std::min(64, atoi(e)));
    if (const char* e = getenv("FPC_NMS_G")) c->nms_g = std::max(0, std::min(64, atoi(e)));
    if (getenv("FPC_L1_T816")) c->layer1_t816 = true;
    // (from here to the end of the stream block: a failure leaves through `fail`, which drops the lock first)
    auto fail = [&](const char* what) {
      g_hip_err = what;
      queue_lock.unlock();
      fpc_destroy(c.release());
      return FPC_E_HIP;
    };
    if (hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess) return fail("hipEventCreateWithFlags");
    for (int i = 1; i < nsub; ++i) {
      hipStream_t st = qmap::acquire(cfg->device, c.get(), SLOT_AUX + (i - 1), true, qprobe, true);
      hipEvent_t ev;
      if (!st) return fail("hipStreamCreateWithFlags");
      c->aux.push_back(st);
      if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return fail("hipEventCreateWithFlags");
      c->ev_join.push_back(ev);
    }
    // FPC_SPLIT_HEADS (detector head + NMS of a sub-batch on a side stream next to its descriptor head): the default of
    // the Python network in FPC_F32 with two or more sub-batches since round 4 (above: +1 % with the streams on queues of
    // their own; the C++ network loses 2 % with it); the variable overrides either way.
    if (const char* e = getenv("FPC_SPLIT_HEADS")) c->split_heads = atoi(e) != 0;
    c->nms_aside = !c->split;  // measured: +1.5 % with two sub-batches (fp32-MFMA kernels), nothing with three (split modes)
    if (pf & FPC_PLAN_NMS_IN_LINE) c->nms_aside = false;
    if (const char* e = getenv("FPC_NMS_ASIDE")) c->nms_aside = atoi(e) != 0;
    if (c->split_heads) c->nms_aside = false;
    // one side stream per sub-batch only while main + aux + side <= 4 streams, otherwise just the first
    const int nside = 2 * nsub <= 4 ? nsub : 1;
    for (int i = 0; i < nsub; ++i) {
      hipStream_t st = nullptr;
      hipEvent_t e1, e2;
      if (i < nside) {
        st = qmap::acquire(cfg->device, c.get(), SLOT_SIDE + i, false, qprobe, true);
        if (!st) return fail("hipStreamCreateWithFlags");
      }
      c->side.push_back(st);
      if (hipEventCreateWithFlags(&e1, hipEventDisableTiming) != hipSuccess) return fail("hipEventCreateWithFlags");
      c->ev_enc.push_back(e1);
      if (hipEventCreateWithFlags(&e2, hipEventDisableTiming) != hipSuccess) return fail("hipEventCreateWithFlags");
      c->ev_det.push_back(e2);
    }
    c->probe_rounds = qmap::state().dev[cfg->device].rounds - probe_rounds0;
    c->placement_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - create_t0).count();
    queue_lock.unlock();
  }
  int rc = FPC_OK;
  // (test hook: the failure path below -- streams, events, registry entries and whatever build_plan had allocated given
  // back, the error code returned -- cannot be reached on this pool's 288 GB otherwise)
  if (rc == FPC_OK && getenv("FPC_TEST_FAIL_CREATE")) {
    g_hip_err = "FPC_TEST_FAIL_CREATE is set: fpc_create's failure path was asked for";
    rc = FPC_E_HIP;
  }
  if (rc == FPC_OK) rc = build_plan(c.get());
  if (rc == FPC_OK && c->plan_error) {
    g_hip_err = "no kernel instance for a layer of this dtype / arch combination";
    rc = FPC_E_INVALID;
  }
  if (rc != FPC_OK) {
    fpc_destroy(c.release());  // streams, events and whatever build_plan had allocated
    return rc;
  }
  *out = c.release();
  return FPC_OK;
}

void fpc_destroy(fpc_ctx* c) {
  if (!c) return;
  hipSetDevice(c->cfg.device);
  hipDeviceSynchronize();
  for (auto e : c->event_pool) hipEventDestroy(e);
  for (auto e : c->ev_join) hipEventDestroy(e);
  for (auto e : c->ev_enc) hipEventDestroy(e);
  for (auto e : c->ev_det) hipEventDestroy(e);
  if (c->ev_fork) hipEventDestroy(c->ev_fork);
  {
    // the ctx's own streams (idle: the device was waited for above) go back to the device's spare list with the queue
    // they were found on -- the next fpc_create takes them from there (queue_map.h); a caller's stream is not ours
    std::lock_guard<std::mutex> queue_lock(qmap::state().mu);
    qmap::DeviceQueues& q = qmap::state().dev[c->cfg.device];
    auto give_back = [&](hipStream_t st, int slot) {
      if (!st) return;
      const qmap::Placed* p = qmap::find(c, slot);
      qmap::retire(q, st, p && p->st == st ? p->qclass : qmap::Q_UNKNOWN);
    };
    for (size_t i = 0; i < c->aux.size(); ++i) give_back(c->aux[i], SLOT_AUX + (int)i);
    for (size_t i = 0; i < c->side.size(); ++i) give_back(c->side[i], SLOT_SIDE + (int)i);
    give_back(c->upload, SLOT_UPLOAD);
    if (c->own_stream) give_back(c->stream, SLOT_MAIN);
    qmap::release_owner(c);
  }
  if (c->slab) hipFree(c->slab);
  if (c->blob) hipFree(c->blob);
  if (c->u8stage) hipFree(c->u8stage);
  if (c->ha_ws) hipFree(c->ha_ws);
  delete c;
}

int fpc_load_weights(fpc_ctx* c, const fpc_tensor* tensors, int n) {
  if (!c || !tensors || n <= 0) return FPC_E_INVALID;
  TensorMap m;
  for (int i = 0; i < n; ++i) {
    if (!tensors[i].name || tensors[i].ndim < 0 || tensors[i].ndim > 4) return FPC_E_INVALID;
    HostTensor t;
    t.data = tensors[i].data;
    t.shape.assign(tensors[i].shape, tensors[i].shape + tensors[i].ndim);
    m[tensors[i].name] = t;
  }
  std::string missing;
  const int rc = pack_all(c, m, &missing);
  if (rc != FPC_OK) {
    g_hip_err = "checkpoint entry: " + missing;
    return rc;
  }
  HIPCHECK(hipSetDevice(c->cfg.device));
  HIPCHECK(hipMemcpy(c->blob, c->host_blob.data(), c->blob_floats * sizeof(float), hipMemcpyHostToDevice));
  c->weights_loaded = true;
  return FPC_OK;
}

size_t fpc_packed_size(const fpc_ctx* c) { return c ? c->blob_floats * sizeof(float) : 0; }
void* fpc_packed_device_ptr(fpc_ctx* c) { return c ? c->blob : nullptr; }

int fpc_export_packed(fpc_ctx* c, void* dst, size_t cap) {
  if (!c || !dst) return FPC_E_INVALID;
  if (!c->weights_loaded) return FPC_E_NO_WEIGHTS;
  if (cap < c->blob_floats * sizeof(float)) return FPC_E_CAPACITY;
  HIPCHECK(hipSetDevice(c->cfg.device));
  HIPCHECK(hipMemcpy(dst, c->blob, c->blob_floats * sizeof(float), hipMemcpyDeviceToHost));
  return FPC_OK;
}

int fpc_import_packed(fpc_ctx* c, const void* src, size_t n) {
  if (!c || !src) return FPC_E_INVALID;
  if (n != c->blob_floats * sizeof(float)) {
    g_hip_err = "packed weights: " + std::to_string(n) + " bytes, this context's plan packs " + std::to_string(c->blob_floats * sizeof(float));
    return FPC_E_INVALID;
  }
  if (!check_blob_header(c, static_cast<const uint32_t*>(src), &g_hip_err)) return FPC_E_INVALID;
  HIPCHECK(hipSetDevice(c->cfg.device));
  HIPCHECK(hipMemcpy(c->blob, src, n, hipMemcpyHostToDevice));
  c->weights_loaded = true;
  return FPC_OK;
}

int fpc_import_packed_device(fpc_ctx* c, const void* src_dev, size_t n) {
  if (!c || !src_dev) return FPC_E_INVALID;
  if (n != c->blob_floats * sizeof(float)) {
    g_hip_err = "packed weights: " + std::to_string(n) + " bytes, this context's plan packs " + std::to_string(c->blob_floats * sizeof(float));
    return FPC_E_INVALID;
  }
  HIPCHECK(hipSetDevice(c->cfg.device));
  // the 64-byte tag is checked on the host BEFORE the blob is touched; the bytes themselves never leave the device
  uint32_t h[BLOB_HEADER_FLOATS];
  HIPCHECK(hipMemcpy(h, src_dev, sizeof(h), hipMemcpyDeviceToHost));
  if (!check_blob_header(c, h, &g_hip_err)) return FPC_E_INVALID;
  if (src_dev != (const void*)c->blob) {
    c->weights_loaded = false;
    HIPCHECK(hipMemcpy(c->blob, src_dev, n, hipMemcpyDeviceToDevice));
    HIPCHECK(hipDeviceSynchronize());
  }
  c->weights_loaded = true;
  return FPC_OK;
}

int fpc_mark_weights_loaded(fpc_ctx* c) {
  if (!c) return FPC_E_INVALID;
  // the blob was written in place (a collective into fpc_packed_device_ptr): it must carry this context's tag
  uint32_t h[BLOB_HEADER_FLOATS];
  HIPCHECK(hipSetDevice(c->cfg.device));
  HIPCHECK(hipMemcpy(h, c->blob, sizeof(h), hipMemcpyDeviceToHost));
  if (!check_blob_header(c, h, &g_hip_err)) {
    c->weights_loaded = false;
    return FPC_E_INVALID;
  }
  c->weights_loaded = true;
  return FPC_OK;
}

uint64_t fpc_plan_hash(const fpc_ctx* c) { return c ? plan_hash(c) : 0; }

int fpc_check_guards(fpc_ctx* c, long long* bad_words) {
  if (!c || !bad_words) return FPC_E_INVALID;
  *bad_words = 0;
  if (!c->guard_zones || c->guards.empty()) return FPC_E_INVALID;      // the context was not created with FPC_PLAN_GUARD_ZONES
  HIPCHECK(hipSetDevice(c->cfg.device));
  HIPCHECK(hipDeviceSynchronize());
  unsigned long long* d = nullptr;
  HIPCHECK(hipMalloc((void**)&d, sizeof(*d)));
  {
    const hipError_t e0 = hipMemset(d, 0, sizeof(*d));
    if (e0 != hipSuccess) hipFree(d);
    HIPCHECK(e0);
  }
  for (const auto& z : c->guards) {
    const size_t n = z.second / 4;
    guard_count_kernel<<<(unsigned)std::min<size_t>(4096, (n + 255) / 256), 256, 0, c->stream>>>(reinterpret_cast<const uint32_t*>(c->slab + z.first), n, GUARD_PATTERN, d);
  }
  unsigned long long h = 0;
  const hipError_t e1 = hipStreamSynchronize(c->stream);
  const hipError_t e2 = hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);
  hipFree(d);
  HIPCHECK(e1);
  HIPCHECK(e2);
  *bad_words = (long long)h;
  return FPC_OK;
}

// RCCL is resolved at run time: first among the libraries the process already holds (a torch host has
// torch/lib/librccl.so mapped; the communicator the caller passes came from THAT copy), then the system's.
namespace {
typedef int (*nccl_bcast_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*nccl_rank_fn)(void*, int*);
struct RcclApi {
  nccl_bcast_fn bcast = nullptr;
  nccl_rank_fn user_rank = nullptr;
  bool tried = false;
};
static RcclApi g_rccl;
static bool load_rccl() {
  if (g_rccl.tried) return g_rccl.bcast && g_rccl.user_rank;
  g_rccl.tried = true;
  void* h = nullptr;
  void* f = dlsym(RTLD_DEFAULT, "ncclBroadcast");
  if (!f) {
    h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return false;
    f = dlsym(h, "ncclBroadcast");
  }
  g_rccl.bcast = reinterpret_cast<nccl_bcast_fn>(f);
  void* r = h ? dlsym(h, "ncclCommUserRank") : dlsym(RTLD_DEFAULT, "ncclCommUserRank");
  g_rccl.user_rank = reinterpret_cast<nccl_rank_fn>(r);
  return g_rccl.bcast && g_rccl.user_rank;
}
}  // namespace

int fpc_broadcast_weights(fpc_ctx* c, void* nccl_comm, int root) {
  if (!c || !nccl_comm || root < 0) return FPC_E_INVALID;
  if (!load_rccl()) {
    g_hip_err = "librccl.so (ncclBroadcast / ncclCommUserRank) could not be resolved";
    return FPC_E_HIP;
  }
  HIPCHECK(hipSetDevice(c->cfg.device));
  int rank = -1;
  if (g_rccl.user_rank(nccl_comm, &rank) != 0 || rank < 0) {
    g_hip_err = "ncclCommUserRank failed";
    return FPC_E_HIP;
  }
  const bool is_root = rank == root;
  // Step 1: the 64-byte tag alone, into scratch -- every rank learns what the root is about to send BEFORE the
  // collective whose size depends on it.  A root without weights sends a tag with the weights-present flag clear.
  // (the tag's device buffer is 64 bytes of the slab carved at fpc_create: an allocation HERE could fail on one rank,
  // which would return early and leave the others waiting in the collective)
  uint32_t* tag_dev = c->bcast_tag;
  uint32_t tag[BLOB_HEADER_FLOATS];
  if (is_root) {
    fill_blob_header(c, tag);
    if (!c->weights_loaded) tag[8] = 0;
    hipMemcpyAsync(tag_dev, tag, sizeof(tag), hipMemcpyHostToDevice, c->stream);
  }
  int nrc = g_rccl.bcast(tag_dev, tag_dev, sizeof(tag), /*ncclUint8*/ 1, root, nccl_comm, c->stream);
  hipError_t he = hipMemcpyAsync(tag, tag_dev, sizeof(tag), hipMemcpyDeviceToHost, c->stream);
  if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
  if (nrc != 0 || he != hipSuccess) {
    g_hip_err = nrc ? "ncclBroadcast(tag) failed: ncclResult " + std::to_string(nrc) : std::string("tag copy: ") + hipGetErrorString(he);
    return FPC_E_HIP;
  }
  if (tag[0] != BLOB_MAGIC || tag[8] != 1) {   // every rank sees the same tag: all return here, no rank is left in step 2
    g_hip_err = "the root rank holds no packed weights (load them on the root before fpc_broadcast_weights)";
    return FPC_E_NO_WEIGHTS;
  }
  // Step 2: the blob, `root_bytes` long on EVERY rank.  A rank whose own plan differs still takes part (into scratch), so
  // that no rank is left waiting in the collective; it then reports the mismatch.
  const size_t root_bytes = (((size_t)tag[7] << 32) | tag[6]) * sizeof(float);
  std::string why;
  const bool match = check_blob_header(c, tag, &why);
  void* dst = c->blob;
  void* scratch = nullptr;
  bool slab_used = false;
  if (!match) {
    // Somewhere to receive root_bytes.  What this rank must NOT do is return before the collective (the others would wait
    // in ncclBroadcast until the backend's time-out), so three places are tried: scratch memory; this rank's own blob
    // when it is large enough (its weights are declared gone BEFORE the bytes arrive -- a broadcast that fails half way
    // leaves the blob undefined); the activation workspace, which holds nothing between calls and is re-zeroed after.
    if (hipMalloc(&scratch, root_bytes) == hipSuccess) {
      dst = scratch;
    } else {
      (void)hipGetLastError();
      scratch = nullptr;
      if (root_bytes <= c->blob_floats * sizeof(float)) {
        c->weights_loaded = false;
      } else if (root_bytes <= c->slab_bytes) {
        dst = c->slab;
        slab_used = true;
      } else {
        g_hip_err = "no memory to take part in the weight broadcast (" + std::to_string(root_bytes) + " bytes: no scratch, and neither this "
                    "context's blob nor its workspace is that large); the other ranks of this communicator will wait for this one";
        return FPC_E_HIP;
      }
    }
  }
  nrc = g_rccl.bcast(dst, dst, root_bytes, 1, root, nccl_comm, c->stream);
  he = hipStreamSynchronize(c->stream);
  if (scratch) hipFree(scratch);
  if (slab_used) {      // foreign bytes in the workspace: back to the state fpc_create left (zeros, canary zones)
    hipMemsetAsync(c->slab, 0, std::min(c->slab_bytes, align_up(root_bytes, 256)), c->stream);
    hipStreamSynchronize(c->stream);
    if (c->guard_zones) fill_guards(c);
  }
  if (nrc != 0 || he != hipSuccess) {
    if (match) c->weights_loaded = false;      // the blob may hold part of the new bytes
    g_hip_err = nrc ? "ncclBroadcast(weights) failed: ncclResult " + std::to_string(nrc) : std::string("broadcast: ") + hipGetErrorString(he);
    return FPC_E_HIP;
  }
  if (!match) {
    g_hip_err = why;
    return FPC_E_INVALID;
  }
  c->weights_loaded = true;
  return FPC_OK;
}

int fpc_set_stream(fpc_ctx* c, void* s) {
  if (!c) return FPC_E_INVALID;
  // the stream the ctx already runs on -- the caller's from an earlier call, or the ctx's OWN one handed back (fpc_get_stream):
  // nothing to do, and ownership stays as it is (round 5's first form destroyed the ctx's own stream in the second case and
  // went on with the dangling handle)
  if ((hipStream_t)s == c->stream) return FPC_OK;
  HIPCHECK(hipSetDevice(c->cfg.device));
  std::lock_guard<std::mutex> queue_lock(qmap::state().mu);
  if (c->own_stream && c->stream) {
    hipStreamSynchronize(c->stream);
    hipStreamDestroy(c->stream);
  }
  c->stream = (hipStream_t)s;
  c->own_stream = false;
  // The registry's entry is THIS context's main slot -- never another context's that happens to name the same caller
  // stream.  The caller's stream is a foreign object: its queue class is looked up in a small per-context memo of handles
  // seen before, and otherwise found with ONE probe round (a one-thread kernel of ~40 us on it) only if that is safe --
  // probing on, not the null stream, not being captured into a graph, and idle.  Anything else: class unknown, the
  // context's other streams stay as they are.
  qmap::Placed* mainp = qmap::find(c, SLOT_MAIN);
  int k = qmap::Q_UNKNOWN;
  qmap::DeviceQueues& q = qmap::state().dev[c->cfg.device];
  bool memo_hit = false;
  for (const auto& m : c->caller_stream_memo)
    if (m.first == c->stream) { k = m.second; memo_hit = true; }
  if (!memo_hit && c->queue_probe && c->stream && !q.futile && q.anchor.size() >= 2) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(c->stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone;
    if (!capturing && hipStreamQuery(c->stream) == hipSuccess) {
      k = qmap::classify(q, c->stream);
      if (k == qmap::Q_INCONCLUSIVE) k = qmap::Q_UNKNOWN;
      if (c->caller_stream_memo.size() >= 4) c->caller_stream_memo.erase(c->caller_stream_memo.begin());
      c->caller_stream_memo.push_back({c->stream, k});
    }
    (void)hipGetLastError();
  }
  if (mainp) {
    mainp->st = c->stream;
    mainp->qclass = k;
  }
  // a sub-batch or side stream of this context on the caller's queue would run in a row with it: exchanged
  if (k >= 0) {
    auto settle = [&](hipStream_t& st, int slot, bool heavy) {
      qmap::Placed* p = qmap::find(c, slot);
      if (!st || !p || p->qclass != k) return;
      const qmap::Placed old = *p;
      for (size_t i = 0; i < qmap::state().placed.size(); ++i)      // (out of the table while its replacement is chosen)
        if (&qmap::state().placed[i] == p) { qmap::state().placed.erase(qmap::state().placed.begin() + i); break; }
      hipStream_t fresh = qmap::acquire(c->cfg.device, c, slot, heavy, true, true);
      if (fresh) {
        hipStreamSynchronize(st);
        hipStreamDestroy(st);
        st = fresh;
      } else {
        qmap::state().placed.push_back(old);
      }
    };
    for (size_t i = 0; i < c->aux.size(); ++i) settle(c->aux[i], SLOT_AUX + (int)i, true);
    for (size_t i = 0; i < c->side.size(); ++i) settle(c->side[i], SLOT_SIDE + (int)i, false);
  }
  return FPC_OK;
}

void* fpc_get_stream(fpc_ctx* c) { return c ? (void*)c->stream : nullptr; }

void* fpc_upload_stream(fpc_ctx* c) {
  if (!c) return nullptr;
  if (c->upload) return (void*)c->upload;
  if (hipSetDevice(c->cfg.device) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> queue_lock(qmap::state().mu);
  // beside this context's main / sub-batch streams (its side streams carry the few-CU NMS launches: sharing a queue with
  // one of them is the lesser evil when all four queues are taken) and, where a queue is left, other contexts' too
  c->upload = qmap::acquire(c->cfg.device, c, SLOT_UPLOAD, false, c->queue_probe, false);
  return (void*)c->upload;
}

int fpc_stream_report(fpc_ctx* c, fpc_stream_report_t* out) {
  if (!c || !out) return FPC_E_INVALID;
  memset(out, 0, sizeof(*out));
  std::lock_guard<std::mutex> queue_lock(qmap::state().mu);
  const qmap::DeviceQueues& q = qmap::state().dev[c->cfg.device];
  out->probing = c->queue_probe && !q.futile ? 1 : 0;
  out->hw_queues_found = (int)q.anchor.size();
  out->process_probe_rounds = q.rounds;
  out->process_probe_launches = q.launches;
  out->process_probe_ms = (float)q.ms;
  out->process_inconclusive_rounds = q.inconclusive;
  out->create_probe_rounds = c->probe_rounds;
  out->create_placement_ms = (float)c->placement_ms;
  out->process_registered_streams = (int)qmap::state().placed.size();
  for (const qmap::Placed& p : qmap::state().placed) {
    if (p.owner != c || out->n_streams >= 16) continue;
    out->slot[out->n_streams] = p.slot;
    out->queue[out->n_streams] = p.qclass;
    out->n_streams += 1;
  }
  return FPC_OK;
}

int fpc_sync(fpc_ctx* c) {
  if (!c) return FPC_E_INVALID;
  HIPCHECK(hipStreamSynchronize(c->stream));
#ifdef FPC_DIAG
  if (c->diag_stamps && c->diag_n) {
    if (FILE* f = fopen(getenv("FPC_STAMP_FILE") ? getenv("FPC_STAMP_FILE") : "/tmp/fpc_stamps.bin", "wb")) {
      fwrite(c->diag_stamps, sizeof(unsigned long long) * 8, getenv("FPC_STAMP_FULL") ? 65536 : c->diag_n, f);
      fclose(f);
    }
  }
#endif
  return FPC_OK;
}

int fpc_forward(fpc_ctx* c, const float* frames, int n, float* prob, float* desc, float* logits) {
  if (!c) return FPC_E_INVALID;
  if (!c->weights_loaded) return FPC_E_NO_WEIGHTS;
  if (n < 1 || n > c->B || !frames) return FPC_E_INVALID;
  HIPCHECK(hipSetDevice(c->cfg.device));
  const bool de = c->cfg.descriptor_enabled != 0;
  if (c->split_f16) HIPCHECK(hipMemsetAsync(c->status + 1, 0, sizeof(int32_t), c->stream));
  int rc = for_each_sub(c, n, [&](const Sub& sb) { run_path(c, frames, sb, de, 0); });
  if (rc != FPC_OK) return rc;
  const int HWc = c->Hc * c->Wc;
  if (prob) HIPCHECK(hipMemcpyAsync(prob, c->prob, (size_t)n * c->H * c->W * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
  if (logits) {
    const size_t tot = (size_t)n * 65 * HWc;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream, c->lg, c->lgcs, 65,
                       HWc, n, logits, (uint32_t*)nullptr, 0);
  }
  if (desc) {
    const size_t tot = (size_t)n * c->D * HWc;
    if (de && c->bf16)
      hipLaunchKernelGGL(nhwc_bf16_to_nchw_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream,
                         reinterpret_cast<const unsigned short*>(c->desc_map), c->D, c->D, HWc, n, desc);
    else if (de)
      hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream,
                         c->desc_map, c->D, c->D, HWc, n, desc, c->range, (int)RANGE_MAX_DESC);
    else  // superpoint.py:106-109: zeros when the descriptor head is disabled
      HIPCHECK(hipMemsetAsync(desc, 0, tot * sizeof(float), c->stream));
  }
  HIPCHECK(hipGetLastError());
  return FPC_OK;
}

// Tensors the last forward left in the workspace, by the reference module that produced them.
int fpc_read_activation(fpc_ctx* c, const char* name, int frame0, int n, float* out, int* channels, int* height, int* width) {
  if (!c || !name || frame0 < 0 || n < 0 || frame0 + n > c->B) return FPC_E_INVALID;
  if (c->vgg) return FPC_E_INVALID;   // the C++ network's layers are read through fpc_forward only
  const int H4 = c->H / 4, W4 = c->W / 4, Hc = c->Hc, Wc = c->Wc, H16 = c->H / 16, W16 = c->W / 16;
  const bool lowp = c->bf16;          // bf16 tensors (all but the logits)
  struct Tap { const char* name; const void* p; int cs, C, H, W; bool bf; int off; };
  const Tap taps[] = {
      {"pool", c->x0, 64, 64, H4, W4, lowp, 0},
      {"layer1.0", c->x1, 64, 64, H4, W4, lowp, 0},
      {"layer1.1", c->x2, 64, 64, H4, W4, lowp, 0},
      {"layer2.0", c->x3, 128, 128, Hc, Wc, lowp, 0},
      {"layer2.1", c->cat, 256, 128, Hc, Wc, lowp, 128},
      {"det.0", c->d0, (c->bf16 || c->split) ? 80 : 72, 65, Hc, Wc, lowp, 0},
      {"det.1", c->lg, c->lgcs, 65, Hc, Wc, false, 0},
      {"desc_in.0", c->y16a, 256, 256, H16, W16, lowp, 0},
      {"desc_in.1", c->y16b, 256, 256, H16, W16, lowp, 0},
      {"up", c->cat, 256, 128, Hc, Wc, lowp, 0},
      {"desc_out.0", c->lo0, 128, 128, Hc, Wc, lowp, 0},
      {"desc_out.1", c->desc_map, 128, 128, Hc, Wc, lowp, 0},
  };
  for (const Tap& t : taps) {
    if (strcmp(t.name, name)) continue;
    if (!c->cfg.descriptor_enabled && (!strncmp(name, "desc", 4) || !strcmp(name, "up"))) return FPC_E_INVALID;
    // FPC_BF16's fpc_detect fuses exp-softmax into detector.layer.1: that call writes NMS state and candidates, no logits
    if (!strcmp(name, "det.1") && !c->logits_valid) return FPC_E_INVALID;
    if (channels) *channels = t.C;
    if (height) *height = t.H;
    if (width) *width = t.W;
    if (!out || n == 0) return FPC_OK;
    HIPCHECK(hipSetDevice(c->cfg.device));
    const int HW = t.H * t.W;
    const size_t tot = (size_t)n * t.C * HW;
    const dim3 grid((unsigned)((tot + 255) / 256));
    if (t.bf) {
      const unsigned short* src = static_cast<const unsigned short*>(t.p) + t.off + (size_t)frame0 * HW * t.cs;
      hipLaunchKernelGGL(nhwc_bf16_to_nchw_kernel, grid, dim3(256), 0, c->stream, src, t.cs, t.C, HW, n, out);
    } else {
      const float* src = static_cast<const float*>(t.p) + t.off + (size_t)frame0 * HW * t.cs;
      hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid, dim3(256), 0, c->stream, src, t.cs, t.C, HW, n, out, (uint32_t*)nullptr, 0);
    }
    HIPCHECK(hipGetLastError());
    return FPC_OK;
  }
  return FPC_E_INVALID;
}

int fpc_detect(fpc_ctx* c, const float* frames, int n) {
  if (!c) return FPC_E_INVALID;
  if (!c->weights_loaded) return FPC_E_NO_WEIGHTS;
  if (n < 1 || n > c->B || !frames) return FPC_E_INVALID;
  HIPCHECK(hipSetDevice(c->cfg.device));
  const bool de = c->cfg.descriptor_enabled != 0;
  if (c->split_f16) HIPCHECK(hipMemsetAsync(c->status + 1, 0, sizeof(int32_t), c->stream));
  return for_each_sub(c, n, [&](const Sub& sb) { run_path(c, frames, sb, de, 1); });
}

int fpc_detect_u8(fpc_ctx* c, const uint8_t* frames, int n, int layout) {
  if (!c || !frames || n < 1 || n > c->B || layout < FPC_U8_GRAY || layout > FPC_U8_BGR_HWC_GRAY) return FPC_E_INVALID;
  if (!c->weights_loaded) return FPC_E_NO_WEIGHTS;
  const int cout = (layout == FPC_U8_GRAY || layout == FPC_U8_BGR_HWC_GRAY) ? 1 : 3;
  if (cout != c->cin) return FPC_E_INVALID;  // gray layouts feed an in_channels = 1 ctx, colour layouts a 3-channel one
  HIPCHECK(hipSetDevice(c->cfg.device));
  const int HW = c->H * c->W;  // multiple of 64 (H, W multiples of 8)
  if (!c->u8stage) HIPCHECK(hipMalloc((void**)&c->u8stage, (size_t)c->B * c->cin * HW * sizeof(float)));
  const size_t quads = (size_t)n * (HW / 4);
  hipLaunchKernelGGL(u8_to_float_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, c->stream, frames,
                     c->u8stage, n, HW, layout);
  return fpc_detect(c, c->u8stage, n);
}

const float* fpc_u8_staging(fpc_ctx* c) { return c ? c->u8stage : nullptr; }

int fpc_detect_u8_resized(fpc_ctx* c, const uint8_t* frames, int n, int src_h, int src_w, int layout) {
  if (!c || !frames || n < 1 || n > c->B || src_h < 2 || src_w < 2 || src_h > 16384 || src_w > 16384 ||
      (layout != FPC_U8_RGB_HWC && layout != FPC_U8_BGR_HWC) || c->cin != 3)
    return FPC_E_INVALID;
  if (!c->weights_loaded) return FPC_E_NO_WEIGHTS;
  HIPCHECK(hipSetDevice(c->cfg.device));
  const size_t HW = (size_t)c->H * c->W;
  if (!c->u8stage) HIPCHECK(hipMalloc((void**)&c->u8stage, (size_t)c->B * c->cin * HW * sizeof(float)));
  ResizeArgs a{};
  a.in = frames;
  a.out = c->u8stage;
  a.n = n; a.src_h = src_h; a.src_w = src_w; a.H = c->H; a.W = c->W;
  // make_query_image, python/src/inference.py:72-85 (Python float = double; int() truncates)
  const double scale_h = (double)c->H / src_h, scale_w = (double)c->W / src_w;
  const double scale_max = scale_h > scale_w ? scale_h : scale_w;
  a.new_w = (int)(src_w * scale_max);
  a.new_h = (int)(src_h * scale_max);
  if (a.new_w < c->W || a.new_h < c->H) return FPC_E_INVALID;  // the reference's crop would come out short
  a.x0 = a.new_w / 2 - c->W / 2;
  a.y0 = a.new_h / 2 - c->H / 2;
  a.scale_x = 1.0 / ((double)a.new_w / src_w);   // cv::resize: inv_scale = dsize / ssize, scale = 1 / inv_scale
  a.scale_y = 1.0 / ((double)a.new_h / src_h);
  a.swap_rb = layout == FPC_U8_BGR_HWC;
  const size_t tot = (size_t)n * HW;
  hipLaunchKernelGGL(resize_crop_u8_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream, a);
  return fpc_detect(c, c->u8stage, n);
}

// 3x3 inverse of a flat homography (h[8], implicit 1), normalised back to flat form -- invert_homography,
// python/src/homographies.py:185-209 (torch.linalg.inv in fp32 there, double here)
static bool invert_flat_homography(const float* h, float* out) {
  const double m[9] = {h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], 1.0};
  const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[2] * m[7] - m[1] * m[8], c02 = m[1] * m[5] - m[2] * m[4];
  const double c10 = m[5] * m[6] - m[3] * m[8], c11 = m[0] * m[8] - m[2] * m[6], c12 = m[2] * m[3] - m[0] * m[5];
  const double c20 = m[3] * m[7] - m[4] * m[6], c21 = m[1] * m[6] - m[0] * m[7], c22 = m[0] * m[4] - m[1] * m[3];
  const double det = m[0] * c00 + m[1] * c10 + m[2] * c20;
  if (!(std::fabs(det) > 1e-300) || !(std::fabs(c22) > 0.0)) return false;
  const double inv[9] = {c00, c01, c02, c10, c11, c12, c20, c21, c22};  // adjugate; the 1/det cancels in mat2flat
  for (int i = 0; i < 8; ++i) out[i] = (float)(inv[i] / inv[8]);
  return true;
}

int fpc_homography_adaptation(fpc_ctx* c, const float* frames, int n, const float* homographies, const float* inverses,
                              int num, int erosion_radius, int aggregation, float* prob_out) {
  if (!c || !frames || !prob_out || n < 1 || n > c->B || num < 0 || (num > 0 && !homographies) || erosion_radius < 0 ||
      erosion_radius > 64 || (aggregation != 0 && aggregation != 1))
    return FPC_E_INVALID;
  if (!c->weights_loaded) return FPC_E_NO_WEIGHTS;
  HIPCHECK(hipSetDevice(c->cfg.device));
  const bool de = c->cfg.descriptor_enabled != 0;
  const size_t HW = (size_t)c->H * c->W, frame_elems = (size_t)c->cin * HW;
  // workspace: warped frames [B,cin,HW] | warped prob [B,HW] | sum [B,HW] | max [B,HW] | cnt, count, mask, tmp [HW] each
  if (!c->ha_ws)
    HIPCHECK(hipMalloc((void**)&c->ha_ws, ((size_t)c->B * frame_elems + 3 * (size_t)c->B * HW + 4 * HW) * sizeof(float)));
  float* wfr = reinterpret_cast<float*>(c->ha_ws);
  float* wprob = wfr + (size_t)c->B * frame_elems;
  float* sum = wprob + (size_t)c->B * HW;
  float* mx = sum + (size_t)c->B * HW;
  float* cnt = mx + (size_t)c->B * HW;
  float* count = cnt + HW;
  float* mask = count + HW;
  float* tmp = mask + HW;
  const dim3 gp((unsigned)((HW + 255) / 256)), bp(256);
  hipStream_t st = c->stream;
  auto forward = [&](const float* f) -> int {  // prob maps of n frames -> c->prob (the path up to depth-to-space)
    return for_each_sub(c, n, [&](const Sub& sb) { run_path(c, f, sb, de, 0); });
  };
  int rc = forward(frames);  // all_probs = net(image), all_counts = 1   (homographies.py:269-270)
  if (rc != FPC_OK) return rc;
  WarpCoeffs none{};
  hipLaunchKernelGGL(unwarp_accumulate_kernel, gp, bp, 0, st, c->prob, none, count, sum, mx, cnt, n, c->H, c->W, 1);
  for (int i = 0; i < num; ++i) {
    WarpCoeffs k{}, kinv{};
    memcpy(k.c, homographies + (size_t)i * 8, sizeof(k.c));
    if (inverses) memcpy(kinv.c, inverses + (size_t)i * 8, sizeof(kinv.c));
    else if (!invert_flat_homography(k.c, kinv.c)) return FPC_E_INVALID;
    // warped = transform(image, H); count = transform(ones, H_inv, nearest); mask = transform(ones, H, nearest)  :288-292
    hipLaunchKernelGGL(warp_perspective_kernel, gp, bp, 0, st, frames, wfr, n * c->cin, c->H, c->W, k, 0, 0);
    float* cdst = erosion_radius ? tmp : count;
    hipLaunchKernelGGL(warp_perspective_kernel, gp, bp, 0, st, (const float*)nullptr, cdst, 1, c->H, c->W, kinv, 1, 1);
    if (erosion_radius) hipLaunchKernelGGL(erode_ellipse_kernel, gp, bp, 0, st, tmp, count, c->H, c->W, erosion_radius);
    float* mdst = erosion_radius ? tmp : mask;
    hipLaunchKernelGGL(warp_perspective_kernel, gp, bp, 0, st, (const float*)nullptr, mdst, 1, c->H, c->W, k, 1, 1);
    if (erosion_radius) hipLaunchKernelGGL(erode_ellipse_kernel, gp, bp, 0, st, tmp, mask, c->H, c->W, erosion_radius);
    rc = forward(wfr);  // warped_prob = net(warped)   :297
    if (rc != FPC_OK) return rc;
    HIPCHECK(hipMemcpyAsync(wprob, c->prob, (size_t)n * HW * sizeof(float), hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(mul_mask_kernel, gp, bp, 0, st, wprob, mask, n, HW);                         // :298
    hipLaunchKernelGGL(unwarp_accumulate_kernel, gp, bp, 0, st, wprob, kinv, count, sum, mx, cnt, n, c->H, c->W, 0);  // :299-305
  }
  hipLaunchKernelGGL(aggregate_kernel, gp, bp, 0, st, sum, mx, cnt, prob_out, n, HW, (float)(num / 3), aggregation);
  HIPCHECK(hipGetLastError());
  return FPC_OK;
}

int fpc_get_points(fpc_ctx* c, const float* prob, const float* desc_nchw, int n) {
  if (!c || !prob || n < 1 || n > c->B) return FPC_E_INVALID;
  HIPCHECK(hipSetDevice(c->cfg.device));
  const int HW = c->H * c->W;
  HIPCHECK(hipMemsetAsync(c->ncand, 0, sizeof(int32_t) * c->B, c->stream));
  // (the per-frame range words belong to the last CALL: a caller-provided map has no input frames to flag, and an earlier
  // fpc_detect's FPC_E_NONFINITE must not come back from this call's fpc_get_counts)
  HIPCHECK(hipMemsetAsync(c->range, 0, sizeof(uint32_t) * FPC_RANGE_WORDS * c->B, c->stream));
  HIPCHECK(hipMemsetAsync(c->rowmax, 0, sizeof(float) * c->Hc * c->B, c->stream));
  const int per = (HW + 255) / 256;
  hipLaunchKernelGGL(threshold_kernel, dim3(per * n), dim3(256), 0, c->stream, prob, n, HW, c->cfg.conf_thresh,
                     c->nmsmap, c->cand, c->ncand);
  Sub all;
  all.f0 = 0;
  all.n = n;
  all.st = c->stream;
  run_nms(c, all);
  if (desc_nchw && c->cfg.descriptor_enabled) {
    const size_t tot = (size_t)n * c->D * c->Hc * c->Wc;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream, desc_nchw,
                       c->D, c->Hc * c->Wc, n, c->desc_in_nhwc);
    run_desc(c, all, c->desc_in_nhwc);
  }
  HIPCHECK(hipGetLastError());
  return FPC_OK;
}

int fpc_sample_descriptors(fpc_ctx* c, const float* desc_nchw, const double* xy, int k, float* out) {
  if (!c || !desc_nchw || k < 0 || (k > 0 && (!xy || !out))) return FPC_E_INVALID;
  if (k == 0) return FPC_OK;
  HIPCHECK(hipSetDevice(c->cfg.device));
  const size_t tot = (size_t)c->D * c->Hc * c->Wc;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream, desc_nchw, c->D,
                     c->Hc * c->Wc, 1, c->desc_in_nhwc);
  if (c->D == 256)
    hipLaunchKernelGGL(descriptor_at_points_kernel<4>, dim3((k + 3) / 4), dim3(256), 0, c->stream, c->desc_in_nhwc, 256,
                       c->Hc, c->Wc, c->H, c->W, xy, k, out);
  else
    hipLaunchKernelGGL(descriptor_at_points_kernel<2>, dim3((k + 3) / 4), dim3(256), 0, c->stream, c->desc_in_nhwc, 128,
                       c->Hc, c->Wc, c->H, c->W, xy, k, out);
  HIPCHECK(hipGetLastError());
  return FPC_OK;
}

int fpc_match(fpc_ctx* c, const float* q, int nq, const float* t, int nt, int cross_check, float max_dist,
              int32_t* match, float* dist) {
  if (!c || nq < 0 || nt < 0 || nq > c->cap || nt > c->cap || (!q && nq) || (!t && nt) || (!match && nq))
    return FPC_E_INVALID;
  HIPCHECK(hipSetDevice(c->cfg.device));
  if (nq == 0) return FPC_OK;
  if (nt == 0) {
    HIPCHECK(hipMemsetAsync(match, 0xff, sizeof(int32_t) * nq, c->stream));  // -1
    return FPC_OK;
  }
  HIPCHECK(hipMemsetAsync(c->rowbest, 0xff, sizeof(unsigned long long) * nq, c->stream));
  HIPCHECK(hipMemsetAsync(c->colbest, 0xff, sizeof(unsigned long long) * nt, c->stream));
  MatchArgs a{};
  a.q = q; a.t = t; a.nq = nq; a.nt = nt; a.D = c->D; a.rowbest = c->rowbest; a.colbest = c->colbest; a.first = nullptr; a.tol2 = -1.f;
  hipLaunchKernelGGL(match_gemm_kernel, dim3((nq + 127) / 128, (nt + 127) / 128), dim3(256), 0, c->stream, a);
  hipLaunchKernelGGL(match_finalize_kernel, dim3((nq + 255) / 256), dim3(256), 0, c->stream, c->rowbest, c->colbest, nq,
                     cross_check, max_dist, match, dist);
  HIPCHECK(hipGetLastError());
  return FPC_OK;
}

int fpc_first_within(fpc_ctx* c, const float* key, int nk, const float* cur, int nc, float tolerance, int32_t* first) {
  if (!c || nk < 0 || nc < 0 || nk > c->cap || nc > c->cap || !(tolerance >= 0.f) || (!key && nk) || (!cur && nc) ||
      (!first && nk))
    return FPC_E_INVALID;
  HIPCHECK(hipSetDevice(c->cfg.device));
  if (nk == 0) return FPC_OK;
  if (nc == 0) {
    HIPCHECK(hipMemsetAsync(first, 0xff, sizeof(int32_t) * nk, c->stream));
    return FPC_OK;
  }
  unsigned int* ws = reinterpret_cast<unsigned int*>(c->rowbest);
  HIPCHECK(hipMemsetAsync(ws, 0xff, sizeof(unsigned int) * nk, c->stream));
  MatchArgs a{};
  a.q = key; a.t = cur; a.nq = nk; a.nt = nc; a.D = c->D; a.rowbest = nullptr; a.colbest = nullptr; a.first = ws; a.tol2 = tolerance * tolerance;
  hipLaunchKernelGGL(match_gemm_kernel, dim3((nk + 127) / 128, (nc + 127) / 128), dim3(256), 0, c->stream, a);
  hipLaunchKernelGGL(first_finalize_kernel, dim3((nk + 255) / 256), dim3(256), 0, c->stream, ws, nk, first);
  HIPCHECK(hipGetLastError());
  return FPC_OK;
}

int fpc_results(fpc_ctx* c, fpc_device_results* out) {
  if (!c || !out) return FPC_E_INVALID;
  out->count = c->count;
  out->n_candidates = c->ncand;
  out->xy = c->xy;
  out->conf = c->conf;
  out->desc = c->cfg.descriptor_enabled ? c->desc_out : nullptr;
  out->capacity = c->cap;
  out->desc_dim = c->D;
  return FPC_OK;
}

int fpc_get_counts(fpc_ctx* c, int n, int32_t* count, int32_t* ncand) {
  if (!c || n < 1 || n > c->B) return FPC_E_INVALID;
  HIPCHECK(hipSetDevice(c->cfg.device));
  HIPCHECK(hipStreamSynchronize(c->stream));
  int32_t st[2] = {0, 0};
  HIPCHECK(hipMemcpy(st, c->status, sizeof(st), hipMemcpyDeviceToHost));
  if (st[0]) return FPC_E_NOT_CONVERGED;
  if (st[1]) return FPC_E_RANGE;
  if (count) HIPCHECK(hipMemcpy(count, c->count, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
  if (ncand) HIPCHECK(hipMemcpy(ncand, c->ncand, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
  // a frame with a NaN / Inf pixel (FPC_F32's stem looks at every pixel it stages): the counts above are delivered, the
  // call is flagged -- include/fpc.h, FPC_E_NONFINITE
  std::vector<uint32_t> rw((size_t)n * FPC_RANGE_WORDS);
  HIPCHECK(hipMemcpy(rw.data(), c->range, rw.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
  for (int b = 0; b < n; ++b)
    if (rw[(size_t)b * FPC_RANGE_WORDS + RANGE_BAD_INPUT]) {
      g_hip_err = "frame " + std::to_string(b) + " of the last call holds a NaN or Inf pixel";
      return FPC_E_NONFINITE;
    }
  return FPC_OK;
}

int fpc_output_range(fpc_ctx* c, int n, float* max_logit, float* max_desc, int32_t* nonfinite_input) {
  if (!c || n < 1 || n > c->B) return FPC_E_INVALID;
  HIPCHECK(hipSetDevice(c->cfg.device));
  HIPCHECK(hipStreamSynchronize(c->stream));
  std::vector<uint32_t> rw((size_t)n * FPC_RANGE_WORDS);
  HIPCHECK(hipMemcpy(rw.data(), c->range, rw.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
  std::vector<float> rm((size_t)n * c->Hc);
  if (max_logit) HIPCHECK(hipMemcpy(rm.data(), c->rowmax, rm.size() * sizeof(float), hipMemcpyDeviceToHost));
  for (int b = 0; b < n; ++b) {
    const uint32_t* w = rw.data() + (size_t)b * FPC_RANGE_WORDS;
    float f;
    if (max_logit) {
      f = 0.f;
      for (int i = 0; i < c->Hc; ++i) f = std::max(f, rm[(size_t)b * c->Hc + i]);
      max_logit[b] = f;
    }
    if (max_desc) { memcpy(&f, w + RANGE_MAX_DESC, 4); max_desc[b] = f; }
    if (nonfinite_input) nonfinite_input[b] = (int32_t)w[RANGE_BAD_INPUT];
  }
  return FPC_OK;
}

int fpc_get_keypoints(fpc_ctx* c, int frame, int cap, int32_t* xy, float* conf, float* desc) {
  if (!c || frame < 0 || frame >= c->B || cap < 0) return FPC_E_INVALID;
  HIPCHECK(hipSetDevice(c->cfg.device));
  HIPCHECK(hipStreamSynchronize(c->stream));
  int32_t k = 0;
  HIPCHECK(hipMemcpy(&k, c->count + frame, sizeof(k), hipMemcpyDeviceToHost));
  if (k > cap || k > c->cap) return FPC_E_CAPACITY;
  if (k == 0) return 0;
  if (xy) HIPCHECK(hipMemcpy(xy, c->xy + (size_t)frame * c->cap * 2, sizeof(int32_t) * 2 * k, hipMemcpyDeviceToHost));
  if (conf) HIPCHECK(hipMemcpy(conf, c->conf + (size_t)frame * c->cap, sizeof(float) * k, hipMemcpyDeviceToHost));
  if (desc) {
    if (!c->cfg.descriptor_enabled) return FPC_E_INVALID;
    HIPCHECK(hipMemcpy(desc, c->desc_out + (size_t)frame * c->cap * c->D, sizeof(float) * c->D * k, hipMemcpyDeviceToHost));
  }
  return k;
}

int fpc_set_timing(fpc_ctx* c, int enable) {
  if (!c) return FPC_E_INVALID;
  c->timing_every = enable > 0 ? enable : 0;
  c->timing_calls = 0;
  c->timing = enable != 0;
  c->timings.clear();   // records accumulate over calls from here on
  c->events_used = 0;
  return FPC_OK;
}

int fpc_get_timings(fpc_ctx* c, int cap, const char** names, const char** kernels, float* ms, double* flops,
                    double* mfma_flops, double* bytes) {
  if (!c) return FPC_E_INVALID;
  const int n = (int)c->timings.size();
  for (int i = 0; i < n && i < cap; ++i) {
    const Timing& t = c->timings[i];
    float v = 0.f;
    if (hipEventElapsedTime(&v, t.start, t.stop) != hipSuccess) v = -1.f;
    const Op* op = t.op >= 0 ? &c->ops[t.op] : nullptr;
    if (names) names[i] = op ? op->name.c_str() : "?";
    if (kernels) {
      const char* k = "?";
      if (op) switch (op->type) {
          case OP_STEM: k = c->bf16 ? "stem_bf16_kernel" : c->split ? "stem_pool_x3_kernel" : c->stem2 ? "stem_pool2_kernel" : c->fuse_stem_pool ? "stem_pool_kernel" : "stem_kernel"; break;
          case OP_POOL: k = "maxpool_kernel"; break;
          case OP_CONV: k = op->lean ? lean_kind(op->kind)->symbol : g_kinds[op->kind].symbol; break;
          case OP_BLOCK: k = g_bkinds[op->bkind].symbol; break;
          case OP_WBLOCK: k = g_wkinds[op->wkind].symbol; break;
          case OP_BF16: k = g_fkinds[op->fkind].symbol; break;
          case OP_SOFTMAX: k = "softmax_d2s_kernel"; break;
          case OP_NMS: k = c->nms_one_workgroup ? "nms_rounds_kernel+nms_sort_kernel" : "nms_rounds_kernel+nms_finish_kernel+nms_chunk_sort_kernel+nms_merge_kernel"; break;
          case OP_DESC: k = "descriptor16_kernel"; break;
          case OP_VCONV0: k = "vgg_conv0_kernel"; break;
          case OP_POOL2: k = "maxpool2_kernel"; break;
          case OP_L2NORM: k = "l2norm256_kernel"; break;
        }
      kernels[i] = k;
    }
    if (ms) ms[i] = v;
    if (flops) flops[i] = op ? op->flops_per_frame * t.frames : 0.0;
    if (mfma_flops) mfma_flops[i] = op ? op->mfma_flops_per_frame * t.frames : 0.0;
    if (bytes) bytes[i] = op ? op->bytes_per_frame * t.frames : 0.0;
  }
  return n;
}

}  // extern "C"
