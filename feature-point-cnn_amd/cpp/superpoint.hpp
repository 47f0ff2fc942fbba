// superpoint.hpp -- C++ host mirror of the reference's entry point for this path:
//
//     reference: cpp/src/superpoint.h:12-36   class superpoint::SuperPoint
//                cpp/src/torchutis.h:11-18    struct FeaturePoint, DescriptorType
//     here:      the same names and call shapes, implemented over the C-ABI of
//                include/fpc.h (libfpc.so) -- no libtorch, OpenCV or TRTorch types.
//
//   superpoint::SuperPoint net("snapshots/super_point.pt", /*load_script=*/false);
//   std::vector<superpoint::FeaturePoint> pts = net.ProcessFrame(gray, rows, cols);
//
// The file is the reference trainer's checkpoint (python/src/saveutils.py:57-62) or the
// flat {name: tensor} export (python/src/inferencewrapper.py:89-91), read by
// pt_reader.hpp.  `load_script` (a TorchScript module compiled through TRTorch in the
// reference, cpp/src/superpoint.cc:11-26) has no meaning here: the kernels ARE the
// compiled model; passing true throws.
//
// Frame: CV_32FC1, values in [0,1], rows x cols, borrowed for the call
// (cpp/src/camera.cc:16-18, cpp/src/torchutis.cc:5-10).  The snapshot network takes 3
// channels; a gray frame is replicated as the reference does (python/src/dataset_utils.py:19-20).
// Errors are C++ exceptions (std::runtime_error), as in the reference's libtorch calls;
// nothing calls exit().  One call at a time per object (members are reused, superpoint.h:31-35).
#pragma once
#include <hip/hip_runtime_api.h>

#include <array>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/fpc.h"
#include "pt_reader.hpp"

namespace superpoint {

using DescriptorType = std::array<float, 256>;  // torchutis.h:11; the snapshot net fills the first 128, the C++ net all 256

struct FeaturePoint {  // torchutis.h:13-18
  int x = 0;
  int y = 0;
  float confidence = 0;
  DescriptorType descriptor{};
};

struct Settings {  // inference fields of cpp/src/settings.h:27-31
  int nms_dist = 4;
  float confidence_thresh = 0.015f;
  float nn_thresh = 0.7f;
  int cell = 8;
  int border_remove = 4;
};

class SuperPoint {
 public:
  SuperPoint(const std::string& file_name, bool load_script, int device = 0) : device_(device) {
    if (load_script)
      throw std::runtime_error("SuperPoint: TorchScript/TRTorch loading is not part of this build; pass the "
                               "checkpoint file with load_script=false");
    ckpt_ = fpc_pt::load_checkpoint(file_name);
    // the flat dict of the reference's own C++ network (cpp/src/model.cc, names as cpp/src/superpoint.cc:27-55
    // copies them) selects that architecture; the trainer's snapshot selects the Python network
    for (auto& kv : ckpt_.tensors)
      if (kv.first == "encoder_conv0_a.weight") vgg_ = true;
  }
  SuperPoint(const SuperPoint&) = delete;
  SuperPoint& operator=(const SuperPoint&) = delete;
  ~SuperPoint() { release(); }

  int descriptor_len() const { return vgg_ ? 256 : 128; }
  bool is_cpp_network() const { return vgg_; }
  // Changes take effect at the NEXT frame: ProcessFrame compares the settings it built its context with against these
  // and rebuilds when nms_dist / confidence_thresh / border_remove differ (round 3 applied them only when the frame size
  // changed and silently ignored them otherwise).
  Settings& settings() { return settings_; }

  // frame: rows x cols floats (gray)
  std::vector<FeaturePoint> ProcessFrame(const float* frame, int rows, int cols) {
    ensure(rows, cols, 1);   // gray plane as is: the library sums the stem filters over the input channels
    hip(hipMemcpy(frame_dev_, frame, (size_t)rows * cols * sizeof(float), hipMemcpyHostToDevice), "upload");
    return run();
  }
  // frame: 3 x rows x cols planar RGB
  std::vector<FeaturePoint> ProcessFrameRGB(const float* chw, int rows, int cols) {
    if (vgg_) throw std::runtime_error("SuperPoint: the C++ network (cpp/src/model.cc) takes one gray plane");
    ensure(rows, cols, 3);
    hip(hipMemcpy(frame_dev_, chw, (size_t)3 * rows * cols * sizeof(float), hipMemcpyHostToDevice), "upload");
    return run();
  }
  // any cv::Mat-like (CV_32FC1, continuous): .rows, .cols, .data
  template <class Mat>
  std::vector<FeaturePoint> ProcessFrame(const Mat& frame) {
    return ProcessFrame(reinterpret_cast<const float*>(frame.data), frame.rows, frame.cols);
  }

 private:
  static void hip(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string("SuperPoint: ") + what + ": " + hipGetErrorString(e));
  }
  static void chk(int rc, const char* what) {
    if (rc < 0) throw std::runtime_error(std::string("SuperPoint: ") + what + ": " + fpc_strerror(rc) + " -- " + fpc_last_hip_error());
  }
  void release() {
    if (ctx_) fpc_destroy(ctx_);
    ctx_ = nullptr;
    if (frame_dev_) (void)hipFree(frame_dev_);
    frame_dev_ = nullptr;
  }
  void ensure(int rows, int cols, int channels) {
    if (ctx_ && rows == rows_ && cols == cols_ && channels == channels_ && settings_.nms_dist == applied_.nms_dist &&
        settings_.confidence_thresh == applied_.confidence_thresh && settings_.border_remove == applied_.border_remove)
      return;
    release();
    fpc_config cfg;
    chk(fpc_default_config(&cfg), "fpc_default_config");
    cfg.device = device_;
    cfg.height = rows;
    cfg.width = cols;
    cfg.max_batch = 1;
    cfg.in_channels = channels;
    cfg.arch = vgg_ ? FPC_ARCH_VGG : FPC_ARCH_RESNET;
    cfg.nms_dist = settings_.nms_dist;
    cfg.conf_thresh = settings_.confidence_thresh;
    cfg.border_remove = settings_.border_remove;
    chk(fpc_create(&ctx_, &cfg), "fpc_create");
    std::vector<fpc_tensor> table;
    for (auto& kv : ckpt_.tensors) {
      if (kv.second.dtype != "float32" || kv.second.shape.size() > 4) continue;  // int64 num_batches_tracked
      fpc_tensor t{};
      t.name = kv.first.c_str();
      t.data = static_cast<const float*>(kv.second.data);
      t.ndim = (int)kv.second.shape.size();
      for (int d = 0; d < t.ndim; ++d) t.shape[d] = kv.second.shape[d];
      table.push_back(t);
    }
    chk(fpc_load_weights(ctx_, table.data(), (int)table.size()), "fpc_load_weights");
    hip(hipSetDevice(device_), "hipSetDevice");
    hip(hipMalloc((void**)&frame_dev_, (size_t)3 * rows * cols * sizeof(float)), "hipMalloc");
    rows_ = rows;
    cols_ = cols;
    channels_ = channels;
    applied_ = settings_;
  }
  std::vector<FeaturePoint> run() {
    chk(fpc_detect(ctx_, frame_dev_, 1), "fpc_detect");
    int32_t k = 0;
    chk(fpc_get_counts(ctx_, 1, &k, nullptr), "fpc_get_counts");
    xy_.resize((size_t)2 * k);
    conf_.resize(k);
    const int D = descriptor_len();
    desc_.resize((size_t)D * k);
    chk(fpc_get_keypoints(ctx_, 0, k, xy_.data(), conf_.data(), desc_.data()), "fpc_get_keypoints");
    feature_points_.resize(k);
    for (int i = 0; i < k; ++i) {
      FeaturePoint& fp = feature_points_[i];
      fp.x = xy_[2 * i];
      fp.y = xy_[2 * i + 1];
      fp.confidence = conf_[i];
      fp.descriptor.fill(0.f);
      std::copy(desc_.begin() + (size_t)D * i, desc_.begin() + (size_t)D * (i + 1), fp.descriptor.begin());
    }
    return feature_points_;  // by-value copy, as cpp/src/superpoint.cc:95
  }

  Settings settings_, applied_;   // as the caller last set them / as the current context was built
  int device_ = 0, rows_ = 0, cols_ = 0, channels_ = 0;
  bool vgg_ = false;
  fpc_pt::Checkpoint ckpt_;
  fpc_ctx* ctx_ = nullptr;
  float* frame_dev_ = nullptr;
  // memory management buffers (superpoint.h:31-35)
  std::vector<float> conf_, desc_;
  std::vector<int32_t> xy_;
  std::vector<FeaturePoint> feature_points_;
};

}  // namespace superpoint
