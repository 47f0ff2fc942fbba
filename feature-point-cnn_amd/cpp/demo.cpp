// demo.cpp -- the shape of the reference's cpp/src/main.cc:60,77 without camera / GUI:
//   demo <checkpoint.pt> <frame.f32> <rows> <cols> [out.txt [nms_dist]]
// (a sixth argument runs the frame a second time after settings().nms_dist = nms_dist -- same object, same frame size --
// and prints that count too: settings changed between frames take effect at the next frame)
// frame.f32 = rows*cols raw float32 gray values in [0,1].  Prints the keypoint count and
// writes "x y confidence d0 d1 d2 d3" per keypoint.  `demo --list <checkpoint.pt>` only
// parses the checkpoint (no GPU needed) and prints name, dtype, shape, sum of every tensor.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "superpoint.hpp"

int main(int argc, char** argv) {
  try {
    if (argc >= 3 && std::string(argv[1]) == "--list") {
      auto ck = fpc_pt::load_checkpoint(argv[2]);
      for (auto& kv : ck.tensors) {
        double sum = 0;
        if (kv.second.dtype == "float32")
          for (size_t i = 0; i < kv.second.numel; ++i) sum += static_cast<const float*>(kv.second.data)[i];
        std::printf("%s %s", kv.first.c_str(), kv.second.dtype.c_str());
        for (auto d : kv.second.shape) std::printf(" %lld", (long long)d);
        std::printf(" | %.9g\n", sum);
      }
      return 0;
    }
    if (argc < 5) {
      std::fprintf(stderr, "usage: %s <checkpoint.pt> <frame.f32> <rows> <cols> [out.txt]\n", argv[0]);
      return 2;
    }
    const int rows = std::atoi(argv[3]), cols = std::atoi(argv[4]);
    std::vector<float> frame((size_t)rows * cols);
    std::ifstream in(argv[2], std::ios::binary);
    if (!in.read(reinterpret_cast<char*>(frame.data()), frame.size() * sizeof(float)))
      throw std::runtime_error(std::string("cannot read frame ") + argv[2]);
    superpoint::SuperPoint net(argv[1], false);
    auto pts = net.ProcessFrame(frame.data(), rows, cols);
    std::printf("%zu feature points\n", pts.size());
    if (argc > 5) {
      std::ofstream out(argv[5]);
      out.precision(9);
      for (auto& p : pts)
        out << p.x << ' ' << p.y << ' ' << p.confidence << ' ' << p.descriptor[0] << ' ' << p.descriptor[1] << ' '
            << p.descriptor[2] << ' ' << p.descriptor[127] << ' ' << p.descriptor[255] << '\n';
    }
    if (argc > 6) {
      net.settings().nms_dist = std::atoi(argv[6]);
      auto again = net.ProcessFrame(frame.data(), rows, cols);
      std::printf("%zu feature points with nms_dist %d\n", again.size(), net.settings().nms_dist);
    }
    return 0;
  } catch (const std::exception& e) {  // cpp/src/main.cc:146-149
    std::cerr << e.what() << std::endl;
    return -1;
  }
}
