// convt_bf16_kernel (csrc/convt_bf16.h, round 5: the bf16 ConvTranspose as ONE launch) stand-alone, beside the four phase
// launches of block_bf16_kernel's conv-only path it replaces: `check` compares both with a double-precision CPU transposed
// convolution on bf16-rounded operands (small and odd maps), `time [B H W]` times them on an HD/16-sized input.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o ct convt_bf16_bench.hip && ./ct check && ./ct time 64 60 80
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <array>
#include <vector>
#include <algorithm>
#include "../../feature-point-cnn_amd/csrc/block_mfma.h"
#include "../../feature-point-cnn_amd/csrc/block_bf16.h"
#include "../../scratch/convt_exp.h"
#include "../../feature-point-cnn_amd/csrc/weights.h"
using namespace fpc;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

static float bf2f_h(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

constexpr int CIN = 256, COUT = 128, CSO = 256;
#define CT_INST 8, 16, 1, 2, 64, 2, 2, 2, 1, 128
#define PH_INST 8, 16, 1, 2, 64, 1, 4, 4, 1, 128
using CTC = ConvTBfCfg<CT_INST>;
using PHC = BlockBfCfg<PH_INST>;

struct Problem {
  int B, H, W;
  std::vector<uint16_t> x;        // [B, H, W, 256] bf16
  std::vector<float> w;           // [256][128][3][3]
  std::vector<double> scale;      // [128]
  std::vector<float> bias;        // [128]
};

static Problem make_problem(int B, int H, int W, unsigned seed) {
  Problem p{B, H, W};
  srand(seed);
  p.x.resize((size_t)B * H * W * CIN);
  for (auto& v : p.x) v = host_f2bf((float)rand() / (float)RAND_MAX - 0.35f);
  p.w.resize((size_t)CIN * COUT * 9);
  for (auto& v : p.w) v = ((float)rand() / (float)RAND_MAX - 0.5f) * 0.08f;
  p.scale.resize(COUT);
  for (auto& v : p.scale) v = 0.75 + 0.5 * rand() / RAND_MAX;
  p.bias.resize(COUT);
  for (auto& v : p.bias) v = (float)rand() / (float)RAND_MAX - 0.5f;
  return p;
}

// out[b, oy, ox, n] = relu(bias[n] + sum_{ci, ky, kx: oy = 2 iy - 1 + ky, ox = 2 ix - 1 + kx} x[b, iy, ix, ci] bf16(w[ci, n, ky, kx] scale[n]))
static std::vector<float> cpu_convt(const Problem& p, int b) {
  const int H = p.H, W = p.W, OH = 2 * H, OW = 2 * W;
  std::vector<float> wq((size_t)9 * CIN * COUT);   // [ky][kx][ci][n] bf16-rounded
  for (int ci = 0; ci < CIN; ++ci)
    for (int n = 0; n < COUT; ++n)
      for (int k = 0; k < 9; ++k)
        wq[((size_t)k * CIN + ci) * COUT + n] = bf2f_h(host_f2bf((float)((double)p.w[((size_t)ci * COUT + n) * 9 + k] * p.scale[n])));
  std::vector<double> acc((size_t)OH * OW * COUT, 0.0);
  for (int iy = 0; iy < H; ++iy)
    for (int ix = 0; ix < W; ++ix) {
      const uint16_t* xp = &p.x[(((size_t)b * H + iy) * W + ix) * CIN];
      for (int ky = 0; ky < 3; ++ky) {
        const int oy = 2 * iy - 1 + ky;
        if (oy < 0 || oy >= OH) continue;
        for (int kx = 0; kx < 3; ++kx) {
          const int ox = 2 * ix - 1 + kx;
          if (ox < 0 || ox >= OW) continue;
          double* o = &acc[((size_t)oy * OW + ox) * COUT];
          for (int ci = 0; ci < CIN; ++ci) {
            const double xv = bf2f_h(xp[ci]);
            const float* wr = &wq[((size_t)(ky * 3 + kx) * CIN + ci) * COUT];
            for (int n = 0; n < COUT; ++n) o[n] += xv * wr[n];
          }
        }
      }
    }
  std::vector<float> out(acc.size());
  for (size_t i = 0; i < acc.size(); ++i) {
    const double v = acc[i] + p.bias[i % COUT];
    out[i] = (float)(v > 0 ? v : 0);
  }
  return out;
}

struct Dev {
  uint16_t* x; uint16_t* out_f; uint16_t* out_p; uint4* wf; float* bf; uint4* wp[4]; float* bp;
};

int main(int argc, char** argv) {
  const bool check = argc > 1 && !strcmp(argv[1], "check");
  std::vector<std::array<int, 3>> cases;
  if (check) cases = {{2, 16, 32}, {3, 30, 40}, {1, 7, 21}, {2, 60, 80}, {1, 4, 8}};
  else cases = {{argc > 2 ? atoi(argv[2]) : 64, argc > 3 ? atoi(argv[3]) : 60, argc > 4 ? atoi(argv[4]) : 80}};
  CK(hipFuncSetAttribute((const void*)convt_bf16_kernel<CT_INST>, hipFuncAttributeMaxDynamicSharedMemorySize, CTC::LDS_BYTES));
  CK(hipFuncSetAttribute((const void*)block_bf16_kernel<PH_INST>, hipFuncAttributeMaxDynamicSharedMemorySize, PHC::LDS_BYTES));
  int nb_ct = 0, nb_ph = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb_ct, (const void*)convt_bf16_kernel<CT_INST>, 256, CTC::LDS_BYTES));
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb_ph, (const void*)block_bf16_kernel<PH_INST>, 256, PHC::LDS_BYTES));
  hipFuncAttributes fa;
  CK(hipFuncGetAttributes(&fa, (const void*)convt_bf16_kernel<CT_INST>));
  printf("convt_bf16_kernel: lds %d B, regs %d, scratch %zu B, %d workgroups/CU; phase kernel %d/CU\n", CTC::LDS_BYTES, fa.numRegs, fa.localSizeBytes, nb_ct, nb_ph);
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  int bad = 0;
  for (auto& cs : cases) {
    const int B = cs[0], H = cs[1], W = cs[2], OH = 2 * H, OW = 2 * W;
    Problem p = make_problem(B, H, W, 7 + H);
    const size_t nx = p.x.size(), no = (size_t)B * OH * OW * CSO;
    Dev d{};
    CK(hipMalloc(&d.x, nx * 2)); CK(hipMalloc(&d.out_f, no * 2)); CK(hipMalloc(&d.out_p, no * 2));
    CK(hipMemcpy(d.x, p.x.data(), nx * 2, hipMemcpyHostToDevice));
    CK(hipMemset(d.out_f, 0x7f, no * 2)); CK(hipMemset(d.out_p, 0x7f, no * 2));
    // fused kernel's fragments: 16 channels per step, taps in ConvTTaps' order
    {
      PackSource s{CIN, CIN, 9, [&](int n, int c, int t) { return (double)p.w[((size_t)c * COUT + n) * 9 + ConvTTaps::ky(t) * 3 + ConvTTaps::kx(t)]; }, &p.scale};
      std::vector<float> frag = pack_conv_bf16({s}, COUT, CTC::NBLK, 16, 1);
      CK(hipMalloc(&d.wf, frag.size() * 4));
      CK(hipMemcpy(d.wf, frag.data(), frag.size() * 4, hipMemcpyHostToDevice));
      CK(hipMalloc(&d.bf, COUT * 4));
      CK(hipMemcpy(d.bf, p.bias.data(), COUT * 4, hipMemcpyHostToDevice));
    }
    BlockBfArgs af{};
    af.x = d.x; af.csx = CIN; af.nchunk = CIN / 64; af.x_bytes = (unsigned)(nx * 2); af.H = H; af.W = W;
    af.w1 = d.wf; af.b1 = d.bf; af.out = d.out_f; af.cso = CSO; af.OH = OH; af.OW = OW;
    af.tiles_x = (W + 15) / 16; af.tiles_y = (H + 7) / 8; af.frame0 = 0; af.total_tiles = af.tiles_x * af.tiles_y * B;
    const int gf = std::min((af.total_tiles + 7) / 8 * 8, std::max(8, nb_ct * cus / CTC::NPARTS / 8 * 8));
    // the four phase launches (fpc_api.hip: add_fconvT)
    BlockBfArgs ap[4];
    const int HWp = bf_halo_pitch(8, 16, 1, 2), ROW16 = 64 / 8 + 1;
    for (int ph = 0; ph < 4; ++ph) {
      const int py = ph >> 1, px = ph & 1;
      BlockBfArgs& a = ap[ph];
      a = BlockBfArgs{};
      a.x = d.x; a.csx = CIN; a.nchunk = CIN / 64; a.x_bytes = (unsigned)(nx * 2); a.H = H; a.W = W; a.pad = 0;
      std::vector<std::pair<int, int>> taps;
      for (int iy = 0; iy < (py ? 2 : 1); ++iy)
        for (int ix = 0; ix < (px ? 2 : 1); ++ix) {
          const int dy = py ? 1 - iy : 0, dx = px ? 1 - ix : 0;
          a.tapoff16[a.ntaps++] = (dy * HWp + dx) * ROW16;
          taps.push_back({py ? (iy == 0 ? 0 : 2) : 1, px ? (ix == 0 ? 0 : 2) : 1});
        }
      PackSource s{CIN, CIN, (int)taps.size(), [&](int n, int c, int t) { return (double)p.w[((size_t)c * COUT + n) * 9 + taps[t].first * 3 + taps[t].second]; }, &p.scale};
      std::vector<float> frag = pack_conv_bf16({s}, COUT, 4, 64, 1);
      CK(hipMalloc(&d.wp[ph], frag.size() * 4));
      CK(hipMemcpy(d.wp[ph], frag.data(), frag.size() * 4, hipMemcpyHostToDevice));
      a.w1 = d.wp[ph]; a.b1 = d.bf; a.conv_only = 1; a.out = d.out_p; a.cso = CSO; a.Ho = H; a.Wo = W; a.OH = OH; a.OW = OW;
      a.oys = a.oxs = 2; a.oy0 = py; a.ox0 = px; a.tiles_x = af.tiles_x; a.tiles_y = af.tiles_y; a.total_tiles = af.total_tiles;
    }
    const int gp = std::min((af.total_tiles + 7) / 8 * 8, std::max(8, nb_ph * cus / 8 * 8));
    auto run_fused = [&]() { hipLaunchKernelGGL((convt_bf16_kernel<CT_INST>), dim3(gf, CTC::NPARTS), dim3(256), CTC::LDS_BYTES, 0, af); };
    auto run_phases = [&]() { for (int ph = 0; ph < 4; ++ph) hipLaunchKernelGGL((block_bf16_kernel<PH_INST>), dim3(gp), dim3(256), PHC::LDS_BYTES, 0, ap[ph]); };
    run_fused();
    run_phases();
    CK(hipDeviceSynchronize());
    if (check) {
      std::vector<uint16_t> of(no), op(no);
      CK(hipMemcpy(of.data(), d.out_f, no * 2, hipMemcpyDeviceToHost));
      CK(hipMemcpy(op.data(), d.out_p, no * 2, hipMemcpyDeviceToHost));
      double worst_f = 0, worst_p = 0; size_t flips = 0, untouched_bad = 0, diff_fp = 0;
      for (int b = 0; b < B; ++b) {
        std::vector<float> want = cpu_convt(p, b);
        for (size_t px = 0; px < (size_t)OH * OW; ++px)
          for (int n = 0; n < CSO; ++n) {
            const size_t i = ((size_t)b * OH * OW + px) * CSO + n;
            if (n >= COUT) { untouched_bad += of[i] != 0x7f7f; continue; }   // channels 128..255 of the cat buffer are not this layer's
            const double w_ = want[px * COUT + n], gf_ = bf2f_h(of[i]), gp_ = bf2f_h(op[i]);
            const double ef = std::fabs(gf_ - w_) / std::max(std::fabs(w_), 1.0), ep = std::fabs(gp_ - w_) / std::max(std::fabs(w_), 1.0);
            worst_f = std::max(worst_f, ef); worst_p = std::max(worst_p, ep);
            flips += ef > 1e-4 && std::fabs(gf_ - bf2f_h(host_f2bf((float)w_))) > 0;
            diff_fp += of[i] != op[i];
          }
      }
      const bool ok = worst_f < 1.25 / 128 && untouched_bad == 0;
      printf("B %d H %d W %d: fused max rel %.3g (phases %.3g), %zu values not the rounded CPU value, %zu differ from the phase launches, %zu foreign bytes touched -> %s\n",
             B, H, W, worst_f, worst_p, flips, diff_fp, untouched_bad, ok ? "ok" : "FAILED");
      bad += !ok;
    } else {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      float bf_ = 1e9, bp_ = 1e9;
      for (int rep = 0; rep < 8; ++rep) {
        float ms;
        hipEventRecord(e0); run_fused(); hipEventRecord(e1); CK(hipDeviceSynchronize());
        hipEventElapsedTime(&ms, e0, e1); bf_ = std::min(bf_, ms);
        hipEventRecord(e0); run_phases(); hipEventRecord(e1); CK(hipDeviceSynchronize());
        hipEventElapsedTime(&ms, e0, e1); bp_ = std::min(bp_, ms);
      }
#ifdef FPC_DIAG
      {
        unsigned long long* st;
        const size_t nst = (size_t)gf * CTC::NPARTS * 16;
        CK(hipMalloc(&st, nst * 8)); CK(hipMemset(st, 0, nst * 8));
        af.stamps = st; run_fused(); CK(hipDeviceSynchronize()); af.stamps = nullptr;
        std::vector<unsigned long long> h(nst);
        CK(hipMemcpy(h.data(), st, nst * 8, hipMemcpyDeviceToHost));
        double sum[12] = {0}; int n = 0;
        for (size_t w = 0; w < nst / 16; ++w) {
          if (!h[w * 16 + 11] || !h[w * 16]) continue;
          for (int i = 1; i < 12; ++i) sum[i] += (double)(h[w * 16 + i] - h[w * 16 + i - 1]);
          ++n;
        }
        printf("stamps over %d workgroups (cycles, third tile): ", n);
        const char* nm[12] = {"", "stage0", "mfma0", "stage1", "mfma1", "stage2", "mfma2", "stage3", "mfma3", "to-epilogue", "acc->lds", "stores"};
        double tot = 0;
        for (int i = 1; i < 12; ++i) { printf("%s %.0f  ", nm[i], sum[i] / std::max(n, 1)); tot += sum[i] / std::max(n, 1); }
        printf("| tile %.0f\n", tot);
        hipFree(st);
      }
#endif
      const double flops = 2.0 * 9 * B * H * W * (double)CIN * COUT;
      printf("B %d H %d W %d: fused %.4f ms (%.0f TFLOP/s algorithmic, grid %d x %d), four phase launches %.4f ms (grid %d)\n", B, H, W, bf_,
             flops / bf_ * 1e-9, gf, CTC::NPARTS, bp_, gp);
    }
    hipFree(d.x); hipFree(d.out_f); hipFree(d.out_p); hipFree(d.wf); hipFree(d.bf);
    for (int ph = 0; ph < 4; ++ph) hipFree(d.wp[ph]);
  }
  return bad ? 1 : 0;
}
