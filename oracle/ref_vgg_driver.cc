// ref_vgg_driver.cc -- test infrastructure: runs the REFERENCE's own C++ network, superpoint::SPModel
// (/root/reference/cpp/src/model.cc + model.h + settings.h, compiled from where they lie by oracle/Makefile.ref
// into oracle/_ref/), on inputs and parameters read from flat binary files, and writes its outputs.
// Only this driver is ours; the network code is the reference's, unmodified, linked against the image's libtorch.
//
//   ref_vgg_forward <params.bin> <input.bin> <output.bin> [repeats]
// With `repeats` the forward pass is run that many more times and "forward_seconds <mean>" and "threads <n>" go to
// stderr (bench.py's cpu_baseline of kind "reference" for --arch vgg).
// params.bin : int32 count, then per entry: int32 name_len, name bytes, int32 ndim, int64 shape[ndim], float data
//              (names as named_parameters() of SPModel gives them -- cpp/src/superpoint.cc:27-55 copies a flat
//              {name: tensor} dict into exactly those)
// input.bin  : int32 n, h, w, then float [n,1,h,w]
// output.bin : int32 n, hc, wc, then float point [n,65,hc,wc], float desc [n,256,hc,wc]   (model.cc:61-93)
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <map>
#include <string>
#include <vector>

#include "model.h"

static bool read_exact(std::ifstream& f, void* p, size_t n) { return (bool)f.read(reinterpret_cast<char*>(p), (std::streamsize)n); }

int main(int argc, char** argv) {
  if (argc != 4 && argc != 5) {
    std::cerr << "usage: ref_vgg_forward params.bin input.bin output.bin\n";
    return 2;
  }
  torch::NoGradGuard no_grad;
  superpoint::Settings settings;
  superpoint::SPModel model(settings);
  model->eval();
  std::map<std::string, at::Tensor> params;
  {
    std::ifstream f(argv[1], std::ios::binary);
    int32_t count = 0;
    if (!read_exact(f, &count, 4)) return 3;
    for (int i = 0; i < count; ++i) {
      int32_t nl = 0, nd = 0;
      read_exact(f, &nl, 4);
      std::string name(nl, '\0');
      read_exact(f, &name[0], nl);
      read_exact(f, &nd, 4);
      std::vector<int64_t> shape(nd);
      read_exact(f, shape.data(), 8 * nd);
      at::Tensor t = torch::empty(shape, torch::kFloat32);
      if (!read_exact(f, t.data_ptr<float>(), sizeof(float) * t.numel())) return 3;
      params[name] = t;
    }
  }
  size_t copied = 0;
  for (auto& p : model->named_parameters()) {  // as cpp/src/superpoint.cc:40-52
    auto it = params.find(p.key());
    if (it == params.end()) {
      std::cerr << "missing parameter " << p.key() << "\n";
      return 4;
    }
    p.value().copy_(it->second);
    ++copied;
  }
  if (copied != params.size()) {
    std::cerr << "unused parameters in file\n";
    return 4;
  }
  std::ifstream fi(argv[2], std::ios::binary);
  int32_t dims[3];
  if (!read_exact(fi, dims, 12)) return 5;
  at::Tensor x = torch::empty({dims[0], 1, dims[1], dims[2]}, torch::kFloat32);
  if (!read_exact(fi, x.data_ptr<float>(), sizeof(float) * x.numel())) return 5;
  auto out = model->forward(x);
  if (argc == 5) {
    const int reps = std::atoi(argv[4]);
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) out = model->forward(x);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::cerr << "forward_seconds " << (reps > 0 ? dt / reps : 0.0) << "\nthreads " << at::get_num_threads() << "\n";
  }
  at::Tensor point = out.first.contiguous(), desc = out.second.contiguous();
  std::ofstream fo(argv[3], std::ios::binary);
  int32_t od[3] = {(int32_t)point.size(0), (int32_t)point.size(2), (int32_t)point.size(3)};
  fo.write(reinterpret_cast<const char*>(od), 12);
  fo.write(reinterpret_cast<const char*>(point.data_ptr<float>()), sizeof(float) * point.numel());
  fo.write(reinterpret_cast<const char*>(desc.data_ptr<float>()), sizeof(float) * desc.numel());
  // the parameter names, one per line, on stdout: the loader's key list comes from the reference itself
  for (auto& p : model->named_parameters()) {
    std::cout << p.key();
    for (auto d : p.value().sizes()) std::cout << " " << d;
    std::cout << "\n";
  }
  return 0;
}
