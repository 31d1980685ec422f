// Test infrastructure (oracle/): a driver around the REFERENCE's own header-only bundle recorder
// (include/analysis/recorder.hpp, included from where it lies under /root/reference; never copied).  It feeds
// caller-provided float32 stereo taps through AnalysisRecorder::capture / tick / write_bundle so that the tap files
// and meta.json under tests/golden/bundle/ are bytes the reference itself wrote.  Built only by oracle/ref_bundle/Makefile
// into oracle/_ref/; only tests/golden/make_bundle_fixture.py runs it.
//
//   ref_bundle_driver <in.f32> <frames> <block_size> <out_dir> <tap name>...
//   in.f32: for each tap in argument order, frames * 2 interleaved float32 (L, R)
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "analysis/recorder.hpp"

int main(int argc, char** argv) {
  if (argc < 6) {
    std::fprintf(stderr, "usage: %s in.f32 frames block_size out_dir tap...\n", argv[0]);
    return 2;
  }
  const long frames = std::atol(argv[2]);
  const int block = std::atoi(argv[3]);
  const int ntaps = argc - 5;
  if (frames <= 0 || block <= 0 || frames % block != 0) return 2;
  std::vector<float> in((size_t)ntaps * frames * 2);
  FILE* f = std::fopen(argv[1], "rb");
  if (!f || std::fread(in.data(), sizeof(float), in.size(), f) != in.size()) return 3;
  std::fclose(f);

  AnalysisRecorder rec;
  rec.set_path(argv[4]);
  rec.begin((int)(frames / block), block, 48000);
  for (long b = 0; b < frames / block; ++b) {
    for (int i = 0; i < block; ++i) {
      const long n = b * block + i;
      for (int t = 0; t < ntaps; ++t) {
        const float* p = in.data() + ((size_t)t * frames + n) * 2;
        ANALYSE_TAP(rec, argv[5 + t], p[0], p[1]);
      }
    }
    if (rec.tick()) rec.write_bundle();
  }
  return rec.finished ? 0 : 4;
}
