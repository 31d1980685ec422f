// native ingest of the recorder's tap files (RIFF/WAVE, 16-bit PCM, mono or stereo; written
// by the reference's C++ recorder, include/analysis/recorder.hpp:55-90) without the Python WAV stack:
//   ira_wav_probe / ira_wav_read_pcm16   host-side: walk the RIFF chunks, read the interleaved int16 payload
//   ira_pcm16_to_channels                device: int16 interleaved -> float32 planar channels with the reference's
//                                        conversion (x / 32768, clip to [-1, 1]; reference analyse/io.py:46-64, :98-113)
//                                        and, for stereo, the optional mono downmix 0.5 * (L + R) in float32
//                                        (analyse/io.py:85-91).  The H2D copy carries 2 bytes per sample instead of 4.
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "ira_common.h"

namespace {

struct WavInfo {
  int32_t sample_rate = 0, channels = 0, bits = 0, format = 0;
  int64_t frames = 0, data_offset = 0;
};

// Returns IRA_OK, IRA_E_UNSUPPORTED (a valid WAV that is not 16-bit PCM), IRA_E_FORMAT (not RIFF/WAVE) or IRA_E_IO
// (truncated).
int32_t parse_wav(FILE* f, WavInfo* w) {
  unsigned char hdr[12];
  if (std::fread(hdr, 1, 12, f) != 12) return IRA_E_FORMAT;
  if (std::memcmp(hdr, "RIFF", 4) != 0 || std::memcmp(hdr + 8, "WAVE", 4) != 0) return IRA_E_FORMAT;
  bool have_fmt = false;
  uint16_t block_align = 0;
  for (;;) {
    unsigned char ch[8];
    if (std::fread(ch, 1, 8, f) != 8) return IRA_E_IO;
    uint32_t size;
    std::memcpy(&size, ch + 4, 4);
    if (std::memcmp(ch, "fmt ", 4) == 0) {
      unsigned char fm[16];
      if (size < 16 || std::fread(fm, 1, 16, f) != 16) return IRA_E_IO;
      uint16_t fmt, chans, bits;
      uint32_t rate;
      std::memcpy(&fmt, fm, 2); std::memcpy(&chans, fm + 2, 2); std::memcpy(&rate, fm + 4, 4);
      std::memcpy(&block_align, fm + 12, 2); std::memcpy(&bits, fm + 14, 2);
      if (fmt == 0xFFFE && size >= 40) {                       // WAVE_FORMAT_EXTENSIBLE: the sub-format's first two bytes
        unsigned char ext[24];
        if (std::fread(ext, 1, 24, f) != 24) return IRA_E_IO;
        std::memcpy(&fmt, ext + 8, 2);
        if (std::fseek(f, (long)(size - 40 + (size & 1)), SEEK_CUR) != 0) return IRA_E_IO;
      } else if (std::fseek(f, (long)(size - 16 + (size & 1)), SEEK_CUR) != 0) {
        return IRA_E_IO;
      }
      w->format = fmt; w->channels = chans; w->sample_rate = (int32_t)rate; w->bits = bits;
      have_fmt = true;
    } else if (std::memcmp(ch, "data", 4) == 0) {
      if (!have_fmt) return IRA_E_FORMAT;
      w->data_offset = std::ftell(f);
      if (w->format != 1 || w->bits != 16 || w->channels < 1 || w->channels > 2) {
        w->frames = block_align ? (int64_t)size / block_align : 0;           // header facts for the caller's validation
        return IRA_E_UNSUPPORTED;
      }
      w->frames = (int64_t)size / (2 * w->channels);
      return IRA_OK;
    } else {
      if (std::fseek(f, (long)(size + (size & 1)), SEEK_CUR) != 0) return IRA_E_IO;   // LIST, fact, ... (word aligned)
    }
  }
}

constexpr int PCM_THREADS = 256;

// mode 0: out[c * frames + i] = conv(pcm[i * channels + c]);  mode 1 (stereo only): out[i] = 0.5f * (conv(L) + conv(R))
__global__ __launch_bounds__(PCM_THREADS) void pcm16_kernel(const int16_t* __restrict__ pcm, long long frames,
                                                            int channels, int mode, float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * PCM_THREADS + threadIdx.x;
  if (i >= frames) return;
  if (channels == 1) {
    out[i] = fminf(fmaxf((float)pcm[i] / 32768.0f, -1.0f), 1.0f);
    return;
  }
  const int32_t both = reinterpret_cast<const int32_t*>(pcm)[i];             // one 4-byte load: L | R << 16
  const float l = fminf(fmaxf((float)(int16_t)(both & 0xFFFF) / 32768.0f, -1.0f), 1.0f);
  const float r = fminf(fmaxf((float)(int16_t)(both >> 16) / 32768.0f, -1.0f), 1.0f);
  if (mode == 1) {
    out[i] = 0.5f * (l + r);
  } else {
    out[i] = l;
    out[frames + i] = r;
  }
}

}  // namespace

extern "C" int32_t ira_wav_probe(const char* path, int32_t* sample_rate, int32_t* channels, int64_t* frames,
                                 int64_t* data_offset) {
  IRA_CHECK_PTR(path); IRA_CHECK_PTR(sample_rate); IRA_CHECK_PTR(channels); IRA_CHECK_PTR(frames);
  IRA_CHECK_PTR(data_offset);
  FILE* f = std::fopen(path, "rb");
  if (!f) return IRA_E_IO;
  WavInfo w;
  const int32_t rc = parse_wav(f, &w);
  std::fclose(f);
  *sample_rate = w.sample_rate; *channels = w.channels; *frames = w.frames; *data_offset = w.data_offset;
  return rc;
}

extern "C" int32_t ira_wav_read_pcm16(const char* path, int64_t data_offset, int64_t frames, int32_t channels,
                                      int16_t* dst_host) {
  IRA_CHECK_PTR(path); IRA_CHECK_PTR(dst_host);
  if (frames < 0 || channels < 1 || channels > 2 || data_offset < 0) return IRA_E_SIZE;
  FILE* f = std::fopen(path, "rb");
  if (!f) return IRA_E_IO;
  int32_t rc = IRA_OK;
  if (std::fseek(f, (long)data_offset, SEEK_SET) != 0) rc = IRA_E_IO;
  const size_t want = (size_t)frames * (size_t)channels;
  if (rc == IRA_OK && std::fread(dst_host, sizeof(int16_t), want, f) != want) rc = IRA_E_IO;
  std::fclose(f);
  return rc;
}

extern "C" int32_t ira_pcm16_to_channels(const int16_t* pcm_dev, int64_t frames, int32_t channels, int32_t mono_downmix,
                                         float* out_dev, void* stream) {
  IRA_CHECK_PTR(pcm_dev); IRA_CHECK_PTR(out_dev);
  if (frames < 0 || channels < 1 || channels > 2) return IRA_E_SIZE;
  if (mono_downmix && channels != 2) return IRA_E_UNSUPPORTED;
  if (frames == 0) return IRA_OK;
  if (channels == 2 && (reinterpret_cast<uintptr_t>(pcm_dev) & 3u) != 0) return IRA_E_SIZE;   // 4-byte frame loads
  const long long blocks = (frames + PCM_THREADS - 1) / PCM_THREADS;
  pcm16_kernel<<<(unsigned)blocks, PCM_THREADS, 0, (hipStream_t)stream>>>(pcm_dev, frames, channels, mono_downmix ? 1 : 0,
                                                                         out_dev);
  IRA_RETURN_LAUNCH();
}
