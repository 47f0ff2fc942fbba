/*
 * fpc_oracle.c -- CPU restatement of the reference's SuperPoint inference path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the *checker* for the HIP path: only
 * tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load it.
 * Nothing under feature-point-cnn_amd/ links, imports or falls back to it.
 *
 * Parity pin: the restatement is checked against golden vectors produced by
 * importing the reference's own python modules (tests/golden/make_golden.py,
 * fixtures under tests/golden/); see tests/test_oracle_vs_golden.py.  The few pieces whose arithmetic lives in
 * libraries absent from the reference tree and from this image (OpenCV's matcher / cvtColor / resize / erode,
 * torchvision's perspective grid) say PARITY UNPINNED at their definition, with what they ARE checked against.
 *
 * Every function cites the reference lines (relative to /root/reference) whose
 * arithmetic it restates.  Tensors are NCHW float32 exactly as the reference
 * holds them.  Convolutions accumulate each output element in double over
 * (ci, ky, kx) in that order and round once to float: the reference's own fp32
 * result (PyTorch / oneDNN) and the HIP path's fp32 MFMA chain both sit within a
 * few 1e-7 relative of it.
 *
 * Build: see oracle/Makefile (gcc -O3 -fopenmp -shared).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define BN_EPS 1e-5f /* torch.nn.BatchNorm2d default: python/src/resnet_blocks.py:8,10,35 */

void oracle_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int oracle_get_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------------- */
/* nn.Conv2d(cin, cout, k, stride, padding=pad, bias=False)                   */
/* call sites: python/src/superpoint.py:12, python/src/resnet_blocks.py:7,9,34 */
/* in  [B,cin,H,W]   w [cout,cin,k,k]   out [B,cout,Ho,Wo]                    */
/* ------------------------------------------------------------------------- */
void oracle_conv2d(const float* in, const float* w, float* out, int B, int cin, int H, int W,
                   int cout, int k, int stride, int pad) {
  const int Ho = (H + 2 * pad - k) / stride + 1;
  const int Wo = (W + 2 * pad - k) / stride + 1;
#pragma omp parallel
  {
    double* acc = (double*)malloc(sizeof(double) * (size_t)Wo);
#pragma omp for collapse(3) schedule(static)
    for (int b = 0; b < B; ++b)
      for (int co = 0; co < cout; ++co)
        for (int y = 0; y < Ho; ++y) {
          for (int x = 0; x < Wo; ++x) acc[x] = 0.0;
          for (int ci = 0; ci < cin; ++ci)
            for (int ky = 0; ky < k; ++ky) {
              const int iy = y * stride + ky - pad;
              if (iy < 0 || iy >= H) continue;
              const float* row = in + (((size_t)b * cin + ci) * H + iy) * W;
              for (int kx = 0; kx < k; ++kx) {
                const double wv = (double)w[(((size_t)co * cin + ci) * k + ky) * k + kx];
                /* valid x: 0 <= x*stride + kx - pad < W */
                int x0 = 0, x1 = Wo;
                while (x0 < Wo && x0 * stride + kx - pad < 0) ++x0;
                while (x1 > x0 && (x1 - 1) * stride + kx - pad >= W) --x1;
                const float* src = row + kx - pad;
                if (stride == 1) {
                  for (int x = x0; x < x1; ++x) acc[x] += wv * (double)src[x];
                } else {
                  for (int x = x0; x < x1; ++x) acc[x] += wv * (double)src[x * stride];
                }
              }
            }
          float* dst = out + (((size_t)b * cout + co) * Ho + y) * Wo;
          for (int x = 0; x < Wo; ++x) dst[x] = (float)acc[x];
        }
    free(acc);
  }
}

/* ------------------------------------------------------------------------- */
/* nn.ConvTranspose2d(cin, cout, 3, stride=2, padding=1, output_padding=1)    */
/* with bias -- python/src/superpoint.py:45.  w is [cin,cout,3,3].            */
/* out[b,co,oy,ox] = bias[co] + sum_{ci,ky,kx : oy = 2*iy - 1 + ky, ...}      */
/*                   in[b,ci,iy,ix] * w[ci,co,ky,kx];  Ho = 2H, Wo = 2W.      */
/* ------------------------------------------------------------------------- */
void oracle_conv_transpose2d(const float* in, const float* w, const float* bias, float* out,
                             int B, int cin, int H, int W, int cout) {
  const int Ho = 2 * H, Wo = 2 * W;
#pragma omp parallel for collapse(3) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int co = 0; co < cout; ++co)
      for (int oy = 0; oy < Ho; ++oy) {
        float* dst = out + (((size_t)b * cout + co) * Ho + oy) * Wo;
        for (int ox = 0; ox < Wo; ++ox) {
          double acc = 0.0;
          for (int ci = 0; ci < cin; ++ci)
            for (int ky = 0; ky < 3; ++ky) {
              const int ty = oy + 1 - ky;
              if (ty < 0 || (ty & 1)) continue;
              const int iy = ty >> 1;
              if (iy >= H) continue;
              for (int kx = 0; kx < 3; ++kx) {
                const int tx = ox + 1 - kx;
                if (tx < 0 || (tx & 1)) continue;
                const int ix = tx >> 1;
                if (ix >= W) continue;
                acc += (double)in[(((size_t)b * cin + ci) * H + iy) * W + ix] *
                       (double)w[(((size_t)ci * cout + co) * 3 + ky) * 3 + kx];
              }
            }
          dst[ox] = (float)(acc + (double)bias[co]);
        }
      }
}

/* ------------------------------------------------------------------------- */
/* nn.BatchNorm2d in .eval() (python/src/inferencewrapper.py:26) optionally   */
/* followed by "+= identity" and nn.ReLU (python/src/resnet_blocks.py:16-27). */
/* PyTorch's inference form: alpha = gamma/sqrt(var+eps); beta = b - mean*alpha;*/
/* y = x*alpha + beta, all in float.  bn = {weight,bias,running_mean,running_var}*/
/* ------------------------------------------------------------------------- */
void oracle_bn_add_relu(float* x, const float* const* bn, const float* add, int relu, int B, int C,
                        int HW) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c) {
      const float invstd = 1.0f / sqrtf(bn[3][c] + BN_EPS);
      const float alpha = bn[0][c] * invstd;
      const float beta = bn[1][c] - bn[2][c] * alpha;
      float* p = x + ((size_t)b * C + c) * HW;
      const float* a = add ? add + ((size_t)b * C + c) * HW : 0;
      for (int i = 0; i < HW; ++i) {
        float v = p[i] * alpha + beta;
        if (a) v += a[i];
        if (relu && !(v > 0.0f)) v = 0.0f;
        p[i] = v;
      }
    }
}

/* nn.MaxPool2d(kernel_size=3, stride=2, padding=1) -- python/src/superpoint.py:15,23 */
void oracle_maxpool3s2(const float* in, float* out, int B, int C, int H, int W) {
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
#pragma omp parallel for collapse(2) schedule(static)
  for (int bc = 0; bc < B * C; ++bc)
    for (int y = 0; y < Ho; ++y) {
      const float* src = in + (size_t)bc * H * W;
      float* dst = out + ((size_t)bc * Ho + y) * Wo;
      for (int x = 0; x < Wo; ++x) {
        float m = -INFINITY;
        for (int ky = 0; ky < 3; ++ky) {
          const int iy = 2 * y + ky - 1;
          if (iy < 0 || iy >= H) continue;
          for (int kx = 0; kx < 3; ++kx) {
            const int ix = 2 * x + kx - 1;
            if (ix < 0 || ix >= W) continue;
            const float v = src[(size_t)iy * W + ix];
            if (v > m) m = v;
          }
        }
        dst[x] = m;
      }
    }
}

/* ------------------------------------------------------------------------- */
/* Weight cursor: the 163 checkpoint entries in state_dict order              */
/* (python/src/saveutils.py:57-62).  BatchNorm consumes 5  */
/* entries (weight, bias, running_mean, running_var, num_batches_tracked).    */
/* ------------------------------------------------------------------------- */
typedef struct {
  const float* const* t;
  int pos;
} cursor_t;

static const float* next_tensor(cursor_t* c) { return c->t[c->pos++]; }
static const float* const* next_bn(cursor_t* c) {
  const float* const* p = c->t + c->pos;
  c->pos += 5;
  return p;
}

/* ResNetBlock.forward -- python/src/resnet_blocks.py:14-27.
 * x [B,cin,H,W] -> returns malloc'd [B,cout,Ho,Wo]; proj = identity_downsample present. */
static float* resnet_block(cursor_t* cur, const float* x, int B, int cin, int H, int W, int cout,
                           int stride, int proj, int* Ho_, int* Wo_) {
  const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  const size_t n = (size_t)B * cout * Ho * Wo;
  float* h = (float*)malloc(n * sizeof(float));
  float* y = (float*)malloc(n * sizeof(float));
  float* idt = 0;
  oracle_conv2d(x, next_tensor(cur), h, B, cin, H, W, cout, 3, stride, 1); /* conv1 :16 */
  oracle_bn_add_relu(h, next_bn(cur), 0, 1, B, cout, Ho * Wo);             /* bn1, relu :17-18 */
  oracle_conv2d(h, next_tensor(cur), y, B, cout, Ho, Wo, cout, 1, 1, 0);   /* conv2 :19 */
  const float* const* bn2 = next_bn(cur);
  if (proj) { /* identity_downsample = Sequential(conv1x1(stride), bn) :22-23, :33-35 */
    idt = (float*)malloc(n * sizeof(float));
    oracle_conv2d(x, next_tensor(cur), idt, B, cin, H, W, cout, 1, stride, 0);
    oracle_bn_add_relu(idt, next_bn(cur), 0, 0, B, cout, Ho * Wo);
  }
  /* bn2, += identity, relu :20,25-26 (identity is x itself when there is no projection) */
  oracle_bn_add_relu(y, bn2, proj ? idt : x, 1, B, cout, Ho * Wo);
  free(h);
  free(idt);
  *Ho_ = Ho;
  *Wo_ = Wo;
  return y;
}

static void tap(float** taps, int i, const float* src, size_t n) {
  if (taps && taps[i]) memcpy(taps[i], src, n * sizeof(float));
}

/* restore_prob_map -- python/src/netutils.py:64-75: drop the dustbin channel,
 * prob[b, 8i + c/8, 8j + c%8] = p[b, c, i, j].  p is [B,65,Hc,Wc]. */
void oracle_restore_prob_map(const float* p, float* prob, int B, int Hc, int Wc, int cell) {
  const int C = cell * cell + 1, H = Hc * cell, W = Wc * cell;
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < cell * cell; ++c)
      for (int i = 0; i < Hc; ++i)
        for (int j = 0; j < Wc; ++j)
          prob[((size_t)b * H + (size_t)i * cell + c / cell) * W + (size_t)j * cell + c % cell] =
              p[(((size_t)b * C + c) * Hc + i) * Wc + j];
}

/* ------------------------------------------------------------------------- */
/* SuperPoint.forward -- python/src/superpoint.py:91-115 (Encoder :19-26,     */
/* Detector :34-36, Descriptor :52-61).                                       */
/* image [B,3,H,W]; weights = 163 pointers in state_dict order (int64 entries */
/* may be NULL); outputs prob_map [B,H,W], desc [B,128,H/8,W/8], logits       */
/* [B,65,H/8,W/8]; taps (optional) = 13 buffers, see tests/golden/README.     */
/* descriptor_enabled=0 reproduces :103-109 (desc = zeros).                   */
/* ------------------------------------------------------------------------- */
int oracle_forward(const float* image, const float* const* weights, int B, int H, int W,
                   int descriptor_enabled, float* prob_map, float* desc, float* logits,
                   float** taps) {
  if (H % 16 || W % 16) return -1;
  cursor_t cur = {weights, 0};
  int h = H / 2, w = W / 2, ho, wo;
  /* Encoder: conv1 7x7 s2 p3 -> bn1 -> relu -> max_pool :20-23 */
  float* s = (float*)malloc((size_t)B * 64 * h * w * sizeof(float));
  oracle_conv2d(image, next_tensor(&cur), s, B, 3, H, W, 64, 7, 2, 3);
  oracle_bn_add_relu(s, next_bn(&cur), 0, 1, B, 64, h * w);
  tap(taps, 0, s, (size_t)B * 64 * h * w);
  float* x = (float*)malloc((size_t)B * 64 * (h / 2) * (w / 2) * sizeof(float));
  oracle_maxpool3s2(s, x, B, 64, h, w);
  free(s);
  h /= 2;
  w /= 2;
  tap(taps, 1, x, (size_t)B * 64 * h * w);
  /* layer1: 2 blocks @64, stride 1; layer2: 2 blocks @128, stride 2 :16-17,24-25 */
  float* y = resnet_block(&cur, x, B, 64, h, w, 64, 1, 1, &ho, &wo);
  free(x);
  tap(taps, 2, y, (size_t)B * 64 * ho * wo);
  x = resnet_block(&cur, y, B, 64, ho, wo, 64, 1, 0, &ho, &wo);
  free(y);
  tap(taps, 3, x, (size_t)B * 64 * ho * wo);
  y = resnet_block(&cur, x, B, 64, ho, wo, 128, 2, 1, &ho, &wo);
  free(x);
  tap(taps, 4, y, (size_t)B * 128 * ho * wo);
  float* feat = resnet_block(&cur, y, B, 128, ho, wo, 128, 1, 0, &ho, &wo);
  free(y);
  tap(taps, 5, feat, (size_t)B * 128 * ho * wo);
  const int Hc = ho, Wc = wo;
  /* Detector: 2 blocks 128 -> 65 -> 65 :32,35 */
  y = resnet_block(&cur, feat, B, 128, Hc, Wc, 65, 1, 1, &ho, &wo);
  tap(taps, 6, y, (size_t)B * 65 * Hc * Wc);
  float* lg = resnet_block(&cur, y, B, 65, Hc, Wc, 65, 1, 0, &ho, &wo);
  free(y);
  tap(taps, 7, lg, (size_t)B * 65 * Hc * Wc);
  memcpy(logits, lg, (size_t)B * 65 * Hc * Wc * sizeof(float));
  /* Descriptor :52-61 */
  const size_t nd = (size_t)B * 128 * Hc * Wc;
  if (descriptor_enabled) {
    int h2, w2;
    y = resnet_block(&cur, feat, B, 128, Hc, Wc, 256, 2, 1, &h2, &w2);
    tap(taps, 8, y, (size_t)B * 256 * h2 * w2);
    x = resnet_block(&cur, y, B, 256, h2, w2, 256, 1, 0, &h2, &w2);
    free(y);
    tap(taps, 9, x, (size_t)B * 256 * h2 * w2);
    float* cat = (float*)malloc(2 * nd * sizeof(float));
    float* up = (float*)malloc(nd * sizeof(float));
    const float* upw = next_tensor(&cur);
    const float* upb = next_tensor(&cur);
    oracle_conv_transpose2d(x, upw, upb, up, B, 256, h2, w2, 128); /* up_sample :55 */
    free(x);
    oracle_bn_add_relu(up, next_bn(&cur), 0, 1, B, 128, Hc * Wc);   /* bn, relu :56-57 */
    tap(taps, 10, up, nd);
    /* torch.cat([out, feature_embeddings], dim=1) :59 -- embeddings == encoder output :36 */
    for (int b = 0; b < B; ++b) {
      memcpy(cat + (size_t)b * 256 * Hc * Wc, up + (size_t)b * 128 * Hc * Wc,
             (size_t)128 * Hc * Wc * sizeof(float));
      memcpy(cat + ((size_t)b * 256 + 128) * Hc * Wc, feat + (size_t)b * 128 * Hc * Wc,
             (size_t)128 * Hc * Wc * sizeof(float));
    }
    free(up);
    y = resnet_block(&cur, cat, B, 256, Hc, Wc, 128, 1, 1, &ho, &wo);
    free(cat);
    tap(taps, 11, y, nd);
    x = resnet_block(&cur, y, B, 128, Hc, Wc, 128, 1, 0, &ho, &wo);
    free(y);
    tap(taps, 12, x, nd);
    memcpy(desc, x, nd * sizeof(float));
    free(x);
  } else {
    memset(desc, 0, nd * sizeof(float));
  }
  free(feat);
  /* softmax_result = exp(prob) / (sum(exp(prob), dim=1) + .00001) :111-112 */
  float* sm = (float*)malloc((size_t)B * 65 * Hc * Wc * sizeof(float));
  const size_t hw = (size_t)Hc * Wc;
  for (int b = 0; b < B; ++b)
    for (size_t i = 0; i < hw; ++i) {
      float sum = 0.0f;
      for (int c = 0; c < 65; ++c) {
        const float e = expf(lg[((size_t)b * 65 + c) * hw + i]);
        sm[((size_t)b * 65 + c) * hw + i] = e;
        sum += e;
      }
      const float den = sum + .00001f;
      for (int c = 0; c < 65; ++c) sm[((size_t)b * 65 + c) * hw + i] /= den;
    }
  oracle_restore_prob_map(sm, prob_map, B, Hc, Wc, 8); /* :114 */
  free(sm);
  free(lg);
  return 0;
}

/* ------------------------------------------------------------------------- */
/* get_points -- python/src/netutils.py:78-100 with get_points_coordinates    */
/* :56-61 and corners_nms python/src/nms.py:4-53, for ONE frame.              */
/*                                                                           */
/* Order on confidence ties: the reference sorts with numpy's default         */
/* (unstable) argsort (nms.py:17,51; netutils.py:92), so its tie order is     */
/* unspecified; this restatement -- and the HIP path -- define it as          */
/* (confidence descending, then row-major pixel index y*W+x ascending).       */
/*                                                                           */
/* prob [H,W]; outputs xs, ys (int32), conf (float32) with capacity `cap`;    */
/* returns the number of points K (even if K > cap; only cap are written).    */
/* n_candidates (optional) receives the count that passed the threshold.      */
/* ------------------------------------------------------------------------- */
typedef struct {
  float conf;
  int32_t idx;
} cand_t;

static int cand_cmp(const void* a, const void* b) {
  const cand_t* p = (const cand_t*)a;
  const cand_t* q = (const cand_t*)b;
  if (p->conf > q->conf) return -1;
  if (p->conf < q->conf) return 1;
  return (p->idx > q->idx) - (p->idx < q->idx);
}

int oracle_get_points(const float* prob, int H, int W, float conf_thresh, int nms_dist,
                      int border_remove, int32_t* xs, int32_t* ys, float* conf, int cap,
                      int* n_candidates) {
  /* np.where(prob_map >= confidence_thresh) -- netutils.py:59 (row-major order) */
  size_t n = 0;
  for (size_t i = 0; i < (size_t)H * W; ++i) n += prob[i] >= conf_thresh;
  if (n_candidates) *n_candidates = (int)n;
  if (n == 0) return 0; /* netutils.py:81-82 */
  cand_t* c = (cand_t*)malloc(n * sizeof(cand_t));
  n = 0;
  for (size_t i = 0; i < (size_t)H * W; ++i)
    if (prob[i] >= conf_thresh) {
      c[n].conf = prob[i];
      c[n].idx = (int32_t)i;
      ++n;
    }
  /* argsort(-conf) nms.py:17 */
  qsort(c, n, sizeof(cand_t), cand_cmp);
  /* grid: 1 = to be processed, 0 = empty/suppressed, -1 = kept; padded by nms_dist nms.py:26-35 */
  const int pad = nms_dist, GW = W + 2 * pad, GH = H + 2 * pad;
  int8_t* grid = (int8_t*)calloc((size_t)GW * GH, 1);
  for (size_t i = 0; i < n; ++i)
    grid[(size_t)(c[i].idx / W + pad) * GW + (c[i].idx % W + pad)] = 1;
  size_t kept = 0;
  if (n == 1) { /* nms.py:23-25: a single corner is returned as is */
    kept = 1;
  } else {
    for (size_t i = 0; i < n; ++i) { /* nms.py:37-44 */
      const int gx = c[i].idx % W + pad, gy = c[i].idx / W + pad;
      if (grid[(size_t)gy * GW + gx] == 1) {
        for (int yy = gy - pad; yy <= gy + pad; ++yy)
          memset(grid + (size_t)yy * GW + gx - pad, 0, (size_t)(2 * pad + 1));
        grid[(size_t)gy * GW + gx] = -1;
        c[kept++] = c[i]; /* survivors stay in descending order: nms.py:46-53, netutils.py:92-93 */
      }
    }
  }
  free(grid);
  /* remove points along the border -- netutils.py:95-99 (after NMS) */
  int K = 0;
  for (size_t i = 0; i < kept; ++i) {
    const int x = c[i].idx % W, y = c[i].idx / W;
    if (x < border_remove || x >= W - border_remove || y < border_remove ||
        y >= H - border_remove)
      continue;
    if (K < cap) {
      xs[K] = x;
      ys[K] = y;
      conf[K] = c[i].conf;
    }
    ++K;
  }
  free(c);
  return K;
}

/* ------------------------------------------------------------------------- */
/* get_descriptors -- python/src/netutils.py:103-121 for ONE frame.           */
/* desc_map [D,Hc,Wc]; points xs, ys (K);  out [K,D] (row k = the reference's  */
/* column k of its [D,K] result).                                             */
/* grid coordinate: x / (W/2) - 1 computed in double then cast to float       */
/* (:111-116); F.grid_sample(bilinear, zeros padding, align_corners=True)     */
/* (:118) un-normalises in float as ((g + 1) / 2) * (size - 1); the four      */
/* corner weights and the nw,ne,sw,se accumulation follow ATen's              */
/* grid_sampler_2d; then each column is divided by its L2 norm, no epsilon    */
/* (:120; an all-zero descriptor gives NaN, as in the reference).             */
/* ------------------------------------------------------------------------- */
/* One body for both entry points: `xs_i/ys_i` (integer pixels, what get_points returns) or `xy_d` ([K][2] float64, the
 * reference's `points` array as it stands -- netutils.py:110-112 normalises in float64, :115 rounds to float once). */
static void get_descriptors_impl(const float* desc_map, int D, int Hc, int Wc, int H, int W, const int32_t* xs,
                                 const int32_t* ys, const double* xy_d, int K, float* out) {
#pragma omp parallel for schedule(static)
  for (int k = 0; k < K; ++k) {
    const double px = xy_d ? xy_d[2 * (size_t)k] : (double)xs[k], py = xy_d ? xy_d[2 * (size_t)k + 1] : (double)ys[k];
    const float gx = (float)(px / ((double)W / 2.) - 1.);
    const float gy = (float)(py / ((double)H / 2.) - 1.);
    const float ix = ((gx + 1.f) / 2.f) * (float)(Wc - 1);
    const float iy = ((gy + 1.f) / 2.f) * (float)(Hc - 1);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float wnw = ((float)x1 - ix) * ((float)y1 - iy);
    const float wne = (ix - (float)x0) * ((float)y1 - iy);
    const float wsw = ((float)x1 - ix) * (iy - (float)y0);
    const float wse = (ix - (float)x0) * (iy - (float)y0);
    const int vx0 = x0 >= 0 && x0 < Wc, vx1 = x1 >= 0 && x1 < Wc;
    const int vy0 = y0 >= 0 && y0 < Hc, vy1 = y1 >= 0 && y1 < Hc;
    float ss = 0.0f;
    for (int d = 0; d < D; ++d) {
      const float* m = desc_map + (size_t)d * Hc * Wc;
      float v = 0.0f;
      if (vy0 && vx0) v += m[(size_t)y0 * Wc + x0] * wnw;
      if (vy0 && vx1) v += m[(size_t)y0 * Wc + x1] * wne;
      if (vy1 && vx0) v += m[(size_t)y1 * Wc + x0] * wsw;
      if (vy1 && vx1) v += m[(size_t)y1 * Wc + x1] * wse;
      out[(size_t)k * D + d] = v;
      ss += v * v;
    }
    const float nrm = sqrtf(ss);
    for (int d = 0; d < D; ++d) out[(size_t)k * D + d] /= nrm;
  }
}

void oracle_get_descriptors(const float* desc_map, int D, int Hc, int Wc, int H, int W,
                            const int32_t* xs, const int32_t* ys, int K, float* out) {
  get_descriptors_impl(desc_map, D, Hc, Wc, H, W, xs, ys, 0, K, out);
}

/* get_descriptors at arbitrary (fractional, possibly out-of-frame) float64 points xy [K][2] */
void oracle_get_descriptors_at(const float* desc_map, int D, int Hc, int Wc, int H, int W, const double* xy, int K,
                               float* out) {
  get_descriptors_impl(desc_map, D, Hc, Wc, H, W, 0, 0, xy, K, out);
}

/* ------------------------------------------------------------------------- */
/* Descriptor matching.                        */
/* PARITY UNPINNED for this block: the Python flavour is OpenCV's             */
/* cv2.BFMatcher(cv2.NORM_L2, crossCheck=True).match(query, train)            */
/* (python/src/inference.py:88-96) and cv2 is not installed here, so no        */
/* golden vectors could be produced; the semantics below restate OpenCV's     */
/* documented behaviour (nearest train descriptor per query, first minimum,   */
/* cross check = the query is also the nearest of its match).  The C++        */
/* flavour restates cpp/src/main.cc:9-29 (DescriptorDist,                     */
/* SearchKeyFrameCorrespondence), which cannot be built here either           */
/* (OpenCV / TRTorch).                                                        */
/* ------------------------------------------------------------------------- */
static double l2_dist(const float* a, const float* b, int D) {
  double sum = 0; /* cpp/src/main.cc:11-15: float differences, double accumulation */
  for (int i = 0; i < D; ++i) sum += (double)((a[i] - b[i]) * (a[i] - b[i]));
  return sqrt(sum);
}

void oracle_match(const float* q, int nq, const float* t, int nt, int D, int cross_check,
                  float max_dist, int32_t* match, float* dist) {
  int32_t* tbest = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nt > 0 ? nt : 1));
#pragma omp parallel for schedule(static)
  for (int j = 0; j < nt; ++j) {
    double bd = 1e300;
    int bi = -1;
    for (int i = 0; i < nq; ++i) {
      const double d = l2_dist(q + (size_t)i * D, t + (size_t)j * D, D);
      if (d < bd) { bd = d; bi = i; }
    }
    tbest[j] = bi;
  }
#pragma omp parallel for schedule(static)
  for (int i = 0; i < nq; ++i) {
    double bd = 1e300;
    int bj = -1;
    for (int j = 0; j < nt; ++j) {
      const double d = l2_dist(q + (size_t)i * D, t + (size_t)j * D, D);
      if (d < bd) { bd = d; bj = j; }
    }
    int ok = bj >= 0;
    if (ok && cross_check) ok = tbest[bj] == i;
    if (ok && max_dist > 0.f) ok = bd < (double)max_dist;
    match[i] = ok ? bj : -1;
    if (dist) dist[i] = bj >= 0 ? (float)bd : 0.f;
  }
  free(tbest);
}

/* SearchKeyFrameCorrespondence, cpp/src/main.cc:18-29: first current-frame point below the tolerance */
void oracle_first_within(const float* key, int nk, const float* cur, int nc, int D, double tolerance,
                         int32_t* first) {
#pragma omp parallel for schedule(static)
  for (int k = 0; k < nk; ++k) {
    first[k] = -1;
    for (int j = 0; j < nc; ++j)
      if (l2_dist(cur + (size_t)j * D, key + (size_t)k * D, D) < tolerance) {
        first[k] = j;
        break;
      }
  }
}

/* ---------------------------------------------------------------------------------------------
 * 8-bit frames -> the float frames the network takes (the step in front of the path; the checker of
 * fpc_detect_u8).  layout 0: gray [n,HW] -> [n,1,HW];  1: RGB HWC -> planar [n,3,HW]
 * (python/src/inferencewrapper.py:70-81 transposes, python/src/camera.py:31 divides: float32(u8) / 255.0);
 * 2: BGR HWC with the swap of cv2.COLOR_BGR2RGB (python/src/inference.py:79);  3: BGR HWC -> gray as
 * cv::cvtColor(COLOR_BGR2GRAY) on 8-bit data followed by convertTo(CV_32FC1, 1.0/255.0)
 * (cpp/src/camera.cc:17-18).  OpenCV is absent from /root/reference and from this image: layout 3 restates
 * OpenCV 4.x's published 8-bit algorithm (color_yuv: B2Y 1868, G2Y 9617, R2Y 4899, shift 14 with rounding;
 * convertScale 8u->32f: float multiply by (float)alpha) -- PARITY UNPINNED for layout 3.
 * Layouts 0-2 are pinned by fixture F6 (tests/golden/f6_u8_to_float.npz: numpy's own evaluation of the
 * reference's expression for all 256 byte values).
 * ------------------------------------------------------------------------------------------- */
void oracle_u8_to_float(const uint8_t* in, int n, int hw, int layout, float* out) {
  for (int f = 0; f < n; ++f)
    for (int p = 0; p < hw; ++p) {
      if (layout == 0) {
        out[(size_t)f * hw + p] = (float)in[(size_t)f * hw + p] / 255.0f;
      } else if (layout == 3) {
        const uint8_t* s = in + ((size_t)f * hw + p) * 3;
        const unsigned y = (s[0] * 1868u + s[1] * 9617u + s[2] * 4899u + 8192u) >> 14;
        out[(size_t)f * hw + p] = (float)y * (float)(1.0 / 255.0);
      } else {
        const uint8_t* s = in + ((size_t)f * hw + p) * 3;
        for (int c = 0; c < 3; ++c) out[((size_t)f * 3 + c) * hw + p] = (float)s[layout == 2 ? 2 - c : c] / 255.0f;
      }
    }
}

/* ---------------------------------------------------------------------------------------------
 * The reference's OTHER network: superpoint::SPModelImpl (cpp/src/model.cc:4-94, dims
 * cpp/src/settings.h:19-25) -- the VGG-style SuperPoint its cpp/ frontend runs: 1-channel input, four
 * pairs of 3x3 convolutions (bias, ReLU) with 2x2 max-pools between them, two heads of 3x3 + 1x1
 * convolutions, 256-D descriptors L2-normalised over channels (model.cc:61-93).  The probability map is
 * SuperPoint::GetPoints' exp / (sum + 1e-5), dustbin dropped, depth-to-space (cpp/src/superpoint.cc:154-172)
 * -- the same arithmetic as python/src/superpoint.py:111-114.
 * PINNED against the reference itself: oracle/_ref/ref_vgg_forward is cpp/src/model.cc compiled unmodified
 * (oracle/Makefile.ref); tests/golden/f7_vgg_*.npz hold its outputs (tests/golden/make_golden_vgg.py).
 * weights: 12 convolutions x (weight, bias) in named_parameters() order: encoder_conv{0..3}_{a,b},
 * detector_conv_{a,b}, descriptor_conv_{a,b}.
 * ------------------------------------------------------------------------------------------- */
static void bias_act(float* x, const float* bias, int relu, int B, int C, size_t hw) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c) {
      float* p = x + ((size_t)b * C + c) * hw;
      const float bv = bias[c];
      for (size_t i = 0; i < hw; ++i) {
        const float v = p[i] + bv;
        p[i] = (relu && v < 0.0f) ? 0.0f : v;
      }
    }
}

/* torch::max_pool2d(x, 2, 2) -- model.cc:74 */
static void maxpool2(const float* in, float* out, int B, int C, int H, int W) {
  const int Ho = H / 2, Wo = W / 2;
#pragma omp parallel for schedule(static)
  for (int bc = 0; bc < B * C; ++bc)
    for (int y = 0; y < Ho; ++y)
      for (int x = 0; x < Wo; ++x) {
        const float* p = in + ((size_t)bc * H + 2 * y) * W + 2 * x;
        float m = p[0];
        if (p[1] > m) m = p[1];
        if (p[W] > m) m = p[W];
        if (p[W + 1] > m) m = p[W + 1];
        out[((size_t)bc * Ho + y) * Wo + x] = m;
      }
}

static float* vgg_conv(const float* x, const float* w, const float* b, int B, int cin, int H, int W, int cout, int k,
                       int relu) {
  float* y = (float*)malloc((size_t)B * cout * H * W * sizeof(float));
  oracle_conv2d(x, w, y, B, cin, H, W, cout, k, 1, k / 2);
  bias_act(y, b, relu, B, cout, (size_t)H * W);
  return y;
}

int oracle_vgg_forward(const float* image, const float* const* weights, int B, int H, int W, float* prob_map,
                       float* desc, float* logits) {
  if (H % 8 || W % 8) return -1;
  static const int dims[4][2] = {{1, 64}, {64, 64}, {64, 128}, {128, 128}}; /* settings.h:19-22 */
  int h = H, w = W, wi = 0;
  float* x = (float*)malloc((size_t)B * H * W * sizeof(float));
  memcpy(x, image, (size_t)B * H * W * sizeof(float));
  for (int i = 0; i < 4; ++i) { /* model.cc:66-76 */
    float* a = vgg_conv(x, weights[wi], weights[wi + 1], B, dims[i][0], h, w, dims[i][1], 3, 1);
    free(x);
    float* b2 = vgg_conv(a, weights[wi + 2], weights[wi + 3], B, dims[i][1], h, w, dims[i][1], 3, 1);
    free(a);
    wi += 4;
    if (i != 3) {
      float* p = (float*)malloc((size_t)B * dims[i][1] * (h / 2) * (w / 2) * sizeof(float));
      maxpool2(b2, p, B, dims[i][1], h, w);
      free(b2);
      h /= 2;
      w /= 2;
      x = p;
    } else {
      x = b2;
    }
  }
  const int Hc = h, Wc = w;
  const size_t hw = (size_t)Hc * Wc;
  float* pa = vgg_conv(x, weights[16], weights[17], B, 128, Hc, Wc, 256, 3, 1); /* model.cc:79-82 */
  float* lg = vgg_conv(pa, weights[18], weights[19], B, 256, Hc, Wc, 65, 1, 0);
  free(pa);
  memcpy(logits, lg, (size_t)B * 65 * hw * sizeof(float));
  float* da = vgg_conv(x, weights[20], weights[21], B, 128, Hc, Wc, 256, 3, 1); /* model.cc:85-88 */
  free(x);
  float* dd = vgg_conv(da, weights[22], weights[23], B, 256, Hc, Wc, 256, 1, 0);
  free(da);
  for (int b = 0; b < B; ++b) /* model.cc:90-91: desc / norm(desc, 2, dim=1), no epsilon */
    for (size_t i = 0; i < hw; ++i) {
      double s = 0.0;
      for (int c = 0; c < 256; ++c) {
        const double v = dd[((size_t)b * 256 + c) * hw + i];
        s += v * v;
      }
      const float nrm = (float)sqrt(s);
      for (int c = 0; c < 256; ++c) desc[((size_t)b * 256 + c) * hw + i] = dd[((size_t)b * 256 + c) * hw + i] / nrm;
    }
  free(dd);
  float* sm = (float*)malloc((size_t)B * 65 * hw * sizeof(float)); /* superpoint.cc:157-172 */
  for (int b = 0; b < B; ++b)
    for (size_t i = 0; i < hw; ++i) {
      float sum = 0.0f;
      for (int c = 0; c < 65; ++c) {
        const float e = expf(lg[((size_t)b * 65 + c) * hw + i]);
        sm[((size_t)b * 65 + c) * hw + i] = e;
        sum += e;
      }
      const float den = sum + .00001f;
      for (int c = 0; c < 65; ++c) sm[((size_t)b * 65 + c) * hw + i] /= den;
    }
  oracle_restore_prob_map(sm, prob_map, B, Hc, Wc, 8);
  free(sm);
  free(lg);
  return 0;
}

/* ---------------------------------------------------------------------------------------------
 * Homography adaptation pieces (python/src/homographies.py:215-216, 238-247, 250-324) -- the checker of
 * fpc_homography_adaptation.  The reference delegates to torchvision.transforms.functional_tensor.perspective and
 * cv2.erode, neither present in /root/reference nor in this image:
 *   - the warp restates torchvision 0.10's `_perspective_grid` (pixel centres at +0.5, coefficients rescaled by half
 *     the output size, projective division, -1) followed by torch.nn.functional.grid_sample(mode, padding zeros,
 *     align_corners False).  Its sampling arithmetic is pinned by fixture F8 (torch's own grid_sample run on a grid
 *     built by this formula, tests/golden/make_golden.py: golden_warp); the grid formula itself is restated from the
 *     library's published source -- PARITY UNPINNED for it;
 *   - the erosion restates OpenCV 4.x getStructuringElement(MORPH_ELLIPSE) + erode(BORDER_CONSTANT, 0) --
 *     PARITY UNPINNED.
 * ------------------------------------------------------------------------------------------- */
static void warp_source(const float* k, int x, int y, int W, int H, float* ix, float* iy) {
  const float X = (float)x + 0.5f, Y = (float)y + 0.5f;
  const float hw = 0.5f * (float)W, hh = 0.5f * (float)H;
  const float nx = fmaf(1.0f, k[2] / hw, fmaf(Y, k[1] / hw, X * (k[0] / hw)));
  const float ny = fmaf(1.0f, k[5] / hh, fmaf(Y, k[4] / hh, X * (k[3] / hh)));
  const float den = fmaf(1.0f, 1.0f, fmaf(Y, k[7], X * k[6]));
  const float gx = nx / den - 1.0f, gy = ny / den - 1.0f;
  *ix = ((gx + 1.0f) * (float)W - 1.0f) / 2.0f;
  *iy = ((gy + 1.0f) * (float)H - 1.0f) / 2.0f;
}

void oracle_warp_perspective(const float* in, int planes, int H, int W, const float* coeffs, int nearest, float* out) {
#pragma omp parallel for schedule(static)
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      float ix, iy;
      warp_source(coeffs, x, y, W, H, &ix, &iy);
      for (int p = 0; p < planes; ++p) {
        const float* src = in + (size_t)p * H * W;
        float v = 0.0f;
        if (nearest) {
          const float rx = nearbyintf(ix), ry = nearbyintf(iy);
          if (rx >= 0.f && rx <= (float)(W - 1) && ry >= 0.f && ry <= (float)(H - 1)) v = src[(size_t)(int)ry * W + (int)rx];
        } else {
          const float fx = floorf(ix), fy = floorf(iy);
          if (fx >= -1.f && fx <= (float)W && fy >= -1.f && fy <= (float)H) {
            const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
            const float wx0 = (float)x1 - ix, wx1 = ix - (float)x0, wy0 = (float)y1 - iy, wy1 = iy - (float)y0;
            if (y0 >= 0 && y0 < H && x0 >= 0 && x0 < W) v += src[(size_t)y0 * W + x0] * (wx0 * wy0);
            if (y0 >= 0 && y0 < H && x1 >= 0 && x1 < W) v += src[(size_t)y0 * W + x1] * (wx1 * wy0);
            if (y1 >= 0 && y1 < H && x0 >= 0 && x0 < W) v += src[(size_t)y1 * W + x0] * (wx0 * wy1);
            if (y1 >= 0 && y1 < H && x1 >= 0 && x1 < W) v += src[(size_t)y1 * W + x1] * (wx1 * wy1);
          }
        }
        out[(size_t)p * H * W + (size_t)y * W + x] = v;
      }
    }
}

void oracle_erode_ellipse(const float* in, int H, int W, int r, float* out) {
  const double inv_r2 = r ? 1.0 / ((double)r * r) : 0.0;
#pragma omp parallel for schedule(static)
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      float m = INFINITY;
      for (int a = 0; a < 2 * r; ++a) {
        const int dy = a - r;
        const int dx = (int)rint((double)r * sqrt(((double)r * r - (double)dy * dy) * inv_r2));
        const int j1 = r - dx > 0 ? r - dx : 0, j2 = r + dx + 1 < 2 * r ? r + dx + 1 : 2 * r;
        const int yy = y + dy;
        for (int j = j1; j < j2; ++j) {
          const int xx = x + j - r;
          const float v = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? in[(size_t)yy * W + xx] : 0.0f;
          if (v < m) m = v;
        }
      }
      out[(size_t)y * W + x] = m;
    }
}

/* ---------------------------------------------------------------------------------------------
 * make_query_image (python/src/inference.py:72-85) applied to a camera.py:31 frame: the checker of
 * fpc_detect_u8_resized.  u8 HWC frame -> /255 -> optional BGR->RGB -> cv2.resize(INTER_LINEAR) to
 * (new_w, new_h) = (int(w s), int(h s)), s = max(H/h, W/w) -> centre crop H x W -> CHW.
 * cv2.resize on float32 (OpenCV 4.x resize.cpp: coordinates from the size ratio in double, floor, both edges clamped
 * with weight 0, horizontal pass then vertical pass in fp32) is restated; OpenCV is absent here, so the restatement
 * is pinned against torch.nn.functional.interpolate(bilinear, align_corners=False) -- the same sampling rule --
 * within 1e-5 (fixture F9) and PARITY WITH cv2 ITSELF IS UNPINNED.
 * ------------------------------------------------------------------------------------------- */
int oracle_resize_crop_u8(const uint8_t* in, int n, int src_h, int src_w, int H, int W, int swap_rb, float* out) {
  const double scale_h = (double)H / src_h, scale_w = (double)W / src_w;
  const double scale_max = scale_h > scale_w ? scale_h : scale_w;
  const int new_w = (int)(src_w * scale_max), new_h = (int)(src_h * scale_max);
  if (new_w < W || new_h < H) return -1;
  const int x0 = new_w / 2 - W / 2, y0 = new_h / 2 - H / 2;
  const double sx_ = 1.0 / ((double)new_w / src_w), sy_ = 1.0 / ((double)new_h / src_h);
  for (int f = 0; f < n; ++f)
    for (int oy = 0; oy < H; ++oy)
      for (int ox = 0; ox < W; ++ox) {
        float fx = (float)(((double)(ox + x0) + 0.5) * sx_ - 0.5), fy = (float)(((double)(oy + y0) + 0.5) * sy_ - 0.5);
        int sx = (int)floorf(fx), sy = (int)floorf(fy);
        fx -= (float)sx;
        fy -= (float)sy;
        if (sx < 0) { fx = 0.f; sx = 0; }
        if (sx >= src_w - 1) { fx = 0.f; sx = src_w - 1; }
        if (sy < 0) { fy = 0.f; sy = 0; }
        if (sy >= src_h - 1) { fy = 0.f; sy = src_h - 1; }
        const int sx1 = sx + 1 < src_w ? sx + 1 : sx, sy1 = sy + 1 < src_h ? sy + 1 : sy;
        const uint8_t* base = in + (size_t)f * src_h * src_w * 3;
        const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
        for (int c = 0; c < 3; ++c) {
          const int sc = swap_rb ? 2 - c : c;
          const float v00 = (float)base[((size_t)sy * src_w + sx) * 3 + sc] / 255.0f;
          const float v01 = (float)base[((size_t)sy * src_w + sx1) * 3 + sc] / 255.0f;
          const float v10 = (float)base[((size_t)sy1 * src_w + sx) * 3 + sc] / 255.0f;
          const float v11 = (float)base[((size_t)sy1 * src_w + sx1) * 3 + sc] / 255.0f;
          const float r0 = fmaf(v01, a1, v00 * a0), r1 = fmaf(v11, a1, v10 * a0);
          out[(((size_t)f * 3 + c) * H + oy) * W + ox] = fmaf(r1, b1, r0 * b0);
        }
      }
  return 0;
}
