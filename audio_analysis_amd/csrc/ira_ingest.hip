// native ingest of the recorder's tap files (RIFF/WAVE, 16-bit PCM, mono or stereo; written
// by the reference's C++ recorder, include/analysis/recorder.hpp:55-90) without the Python WAV stack:
//   ira_wav_probe / ira_wav_read_pcm16   host-side: walk the RIFF chunks, read the interleaved int16 payload
//   ira_pcm16_to_channels                device: int16 interleaved -> float32 planar channels with the reference's
//                                        conversion (x / 32768, clip to [-1, 1]; reference analyse/io.py:46-64, :98-113)
//                                        and, for stereo, the optional mono downmix 0.5 * (L + R) in float32
//                                        (analyse/io.py:85-91).  The H2D copy carries 2 bytes per sample instead of 4.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

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
      // A payload shorter than the header says: the reference's reader (scipy.io.wavfile.read behind analyse/io.py:200)
      // warns "Reached EOF prematurely" and returns the samples the file holds (numpy.fromfile reads what is there); the
      // reference then analyses them.  Whole samples that do not make whole stereo frames fail its reshape(-1, channels)
      // with a ValueError -> IRA_E_FORMAT here.
      int64_t samples = (int64_t)size / 2;
      const long here = std::ftell(f);
      if (std::fseek(f, 0, SEEK_END) == 0) {
        const int64_t avail = ((int64_t)std::ftell(f) - (int64_t)here) / 2;
        if (avail < samples) samples = avail < 0 ? 0 : avail;
      }
      (void)std::fseek(f, here, SEEK_SET);
      if (samples % w->channels != 0) return IRA_E_FORMAT;
      w->frames = samples / w->channels;
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

// The same conversion for a whole GROUP of tap files in ONE launch: grid (frame blocks, files), every file described by a
// row of the job table.  One launch per tap (round 2) cost 32 launches + as many event pairs per 32-tap step -- 1.7 ms of
// a 6.2 ms step for 46 MB of traffic; one launch over the table is bandwidth-bound.
constexpr int PCM_PER_THREAD = 4;

__global__ __launch_bounds__(PCM_THREADS) void pcm16_jobs_kernel(const int16_t* __restrict__ pcm, const long long* __restrict__ src_off,
                                                                 const long long* __restrict__ nframes,
                                                                 const int32_t* __restrict__ nchannels,
                                                                 const int32_t* __restrict__ modes,
                                                                 const long long* __restrict__ dst_off, float* __restrict__ out) {
  const int f = blockIdx.y;
  const long long frames = ira::uniform(nframes[f]);
  const long long i0 = ((long long)blockIdx.x * PCM_THREADS) * PCM_PER_THREAD + threadIdx.x;
  if ((long long)blockIdx.x * PCM_THREADS * PCM_PER_THREAD >= frames) return;
  const int channels = ira::uniform(nchannels[f]), mode = ira::uniform(modes[f]);
  const int16_t* src = pcm + ira::uniform(src_off[f]);
  float* dst = out + ira::uniform(dst_off[f]);
#pragma unroll
  for (int u = 0; u < PCM_PER_THREAD; ++u) {
    const long long i = i0 + (long long)u * PCM_THREADS;
    if (i >= frames) break;
    if (channels == 1) {
      dst[i] = fminf(fmaxf((float)src[i] / 32768.0f, -1.0f), 1.0f);
      continue;
    }
    const int32_t both = reinterpret_cast<const int32_t*>(src)[i];           // one 4-byte load: L | R << 16
    const float l = fminf(fmaxf((float)(int16_t)(both & 0xFFFF) / 32768.0f, -1.0f), 1.0f);
    const float r = fminf(fmaxf((float)(int16_t)(both >> 16) / 32768.0f, -1.0f), 1.0f);
    if (mode == 1) {
      dst[i] = 0.5f * (l + r);
    } else {
      dst[i] = l;
      dst[frames + i] = r;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// ira_host_pull: the batch upload as a KERNEL that reads pinned host memory through the PCIe link and writes HBM -- an
// alternative to hipMemcpyAsync that converts PCM16 while it copies (no int16 staging buffer in HBM) and leaves the copy
// engines alone.  Each lane keeps PULL_U 16-byte reads of host memory in flight; a PCIe round trip is ~1.5-2 us, so
// ~100 KB in flight fill a Gen5 x16 link: 8 workgroups x 256 lanes x 4 x 16 B = 131 KB, 46-47 GB/s measured.  MORE is
// worse: with 32-48 workgroups the outstanding host reads crowd the fabric queues every other kernel's HBM reads go
// through (full report, 64 x 10 s: peak pick 0.09 -> 1.2 ms, step 6.9 -> 8.4 ms).  Against the copy engine it is a draw
// (tools/upload_ab.py, alternating order in one process: 10.37 k vs 9.4-10.8 k IRs/s at 256 x 10 s per step), so the
// feed uses it only when asked (audio_analysis_amd.feed.DeviceFeed(pull=True)).
// ------------------------------------------------------------------------------------------------
constexpr int PULL_THREADS = 256;
constexpr int PULL_U = 4;
typedef int pull_i4 __attribute__((ext_vector_type(4)));
typedef float pull_f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float pcm_conv(int v) { return fminf(fmaxf((float)v / 32768.0f, -1.0f), 1.0f); }

// format 0: float32 -> float32 (n16 = number of 16-byte pieces); format 1: mono int16 -> float32 (8 samples per piece)
template <int FORMAT>
__global__ __launch_bounds__(PULL_THREADS) void host_pull_kernel(const pull_i4* __restrict__ src, float* __restrict__ dst,
                                                                 long long n16) {
  const long long stride = (long long)gridDim.x * PULL_THREADS;
  for (long long base = (long long)blockIdx.x * PULL_THREADS + threadIdx.x; base < n16; base += stride * PULL_U) {
    pull_i4 v[PULL_U];
#pragma unroll
    for (int u = 0; u < PULL_U; ++u) {
      const long long i = base + stride * u;
      v[u] = i < n16 ? __builtin_nontemporal_load(src + i) : pull_i4{0, 0, 0, 0};
    }
#pragma unroll
    for (int u = 0; u < PULL_U; ++u) {
      const long long i = base + stride * u;
      if (i >= n16) continue;
      if (FORMAT == 0) {
        reinterpret_cast<pull_i4*>(dst)[i] = v[u];
      } else {
        pull_f4 a, b;
        a.x = pcm_conv((int)(short)(v[u].x & 0xFFFF)); a.y = pcm_conv(v[u].x >> 16);
        a.z = pcm_conv((int)(short)(v[u].y & 0xFFFF)); a.w = pcm_conv(v[u].y >> 16);
        b.x = pcm_conv((int)(short)(v[u].z & 0xFFFF)); b.y = pcm_conv(v[u].z >> 16);
        b.z = pcm_conv((int)(short)(v[u].w & 0xFFFF)); b.w = pcm_conv(v[u].w >> 16);
        reinterpret_cast<pull_f4*>(dst)[2 * i] = a;
        reinterpret_cast<pull_f4*>(dst)[2 * i + 1] = b;
      }
    }
  }
}

template <int FORMAT>
__global__ void host_pull_tail_kernel(const void* __restrict__ src, float* __restrict__ dst, long long first, long long count) {
  const long long i = first + (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  if (FORMAT == 0) dst[i] = static_cast<const float*>(src)[i];
  else dst[i] = pcm_conv((int)static_cast<const int16_t*>(src)[i]);
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

// ---- a whole group of taps in ONE call: a Python caller's interpreter lock is released once, for the whole group, instead of
// being taken and dropped around every file by sixteen pool threads (which starved the thread that drives the GPU: at 128
// taps per step the analysis thread spent more time waiting for the lock than enqueueing work).  The threads live inside the
// call (created and joined here: the library keeps no state).
namespace {
template <typename F>
void for_each_file(int32_t n, int32_t threads, F work) {
  if (threads < 1) threads = 1;
  if (threads > n) threads = n;
  if (threads > 64) threads = 64;
  if (threads <= 1) {
    for (int32_t i = 0; i < n; ++i) work(i);
    return;
  }
  std::atomic<int32_t> next{0};
  auto drain = [&]() {
    for (int32_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) work(i);
  };
  std::vector<std::thread> pool;
  try {                                     // no exception may cross the C boundary: a thread that cannot be started
    pool.reserve((size_t)threads);          // (resource limits) just means fewer readers -- the caller's thread drains too
    for (int32_t t = 1; t < threads; ++t) pool.emplace_back(drain);
  } catch (...) {
  }
  drain();
  for (auto& th : pool) th.join();
}
}  // namespace

extern "C" int32_t ira_wav_probe_batch(const char* const* paths, int32_t n, int32_t threads, int32_t* status,
                                       int32_t* sample_rate, int32_t* channels, int64_t* frames, int64_t* data_offset) {
  IRA_CHECK_PTR(paths); IRA_CHECK_PTR(status); IRA_CHECK_PTR(sample_rate); IRA_CHECK_PTR(channels); IRA_CHECK_PTR(frames);
  IRA_CHECK_PTR(data_offset);
  if (n < 0) return IRA_E_SIZE;
  for_each_file(n, threads, [&](int32_t i) {
    status[i] = paths[i] ? ira_wav_probe(paths[i], sample_rate + i, channels + i, frames + i, data_offset + i) : IRA_E_NULL;
  });
  return IRA_OK;
}

extern "C" int32_t ira_wav_read_pcm16_batch(const char* const* paths, const int64_t* data_offset, const int64_t* frames,
                                            const int32_t* channels, const int64_t* dst_off, int16_t* dst_host, int32_t n,
                                            int32_t threads, int32_t* status) {
  IRA_CHECK_PTR(paths); IRA_CHECK_PTR(data_offset); IRA_CHECK_PTR(frames); IRA_CHECK_PTR(channels); IRA_CHECK_PTR(dst_off);
  IRA_CHECK_PTR(dst_host); IRA_CHECK_PTR(status);
  if (n < 0) return IRA_E_SIZE;
  for_each_file(n, threads, [&](int32_t i) {
    status[i] = (paths[i] && dst_off[i] >= 0)
                    ? ira_wav_read_pcm16(paths[i], data_offset[i], frames[i], channels[i], dst_host + dst_off[i])
                    : IRA_E_NULL;
  });
  return IRA_OK;
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

extern "C" int32_t ira_pcm16_to_channels_jobs(const int16_t* pcm_dev, const int64_t* src_off_dev, const int64_t* frames_dev,
                                              const int32_t* channels_dev, const int32_t* mode_dev,
                                              const int64_t* dst_off_dev, int32_t nfiles, int64_t max_frames, float* out_dev,
                                              void* stream) {
  IRA_CHECK_PTR(pcm_dev); IRA_CHECK_PTR(src_off_dev); IRA_CHECK_PTR(frames_dev); IRA_CHECK_PTR(channels_dev);
  IRA_CHECK_PTR(mode_dev); IRA_CHECK_PTR(dst_off_dev); IRA_CHECK_PTR(out_dev);
  if (nfiles < 0 || max_frames < 0 || nfiles > 65535) return IRA_E_SIZE;
  if (nfiles == 0 || max_frames == 0) return IRA_OK;
  if ((reinterpret_cast<uintptr_t>(pcm_dev) & 3u) != 0) return IRA_E_SIZE;                     // 4-byte frame loads
  const long long per_block = (long long)PCM_THREADS * PCM_PER_THREAD;
  const long long blocks = (max_frames + per_block - 1) / per_block;
  if (blocks > 0x7fffffffll) return IRA_E_SIZE;
  static_assert(sizeof(long long) == sizeof(int64_t), "job table element size");
  pcm16_jobs_kernel<<<dim3((unsigned)blocks, (unsigned)nfiles), PCM_THREADS, 0, (hipStream_t)stream>>>(
      pcm_dev, reinterpret_cast<const long long*>(src_off_dev), reinterpret_cast<const long long*>(frames_dev), channels_dev,
      mode_dev, reinterpret_cast<const long long*>(dst_off_dev), out_dev);
  IRA_RETURN_LAUNCH();
}

extern "C" int32_t ira_host_pull(const void* host_src, int64_t count, int32_t format, float* out_dev,
                                 int32_t workgroups, void* stream) {
  IRA_CHECK_PTR(host_src); IRA_CHECK_PTR(out_dev);
  if (count < 0 || (format != 0 && format != 1)) return IRA_E_SIZE;
  if (count == 0) return IRA_OK;
  // the kernel dereferences the DEVICE view of the pinned allocation; a pointer that is not mapped host memory is refused
  void* dev_view = nullptr;
  if (hipHostGetDevicePointer(&dev_view, const_cast<void*>(host_src), 0) != hipSuccess || dev_view == nullptr) {
    (void)hipGetLastError();
    return IRA_E_UNSUPPORTED;
  }
  if ((reinterpret_cast<uintptr_t>(dev_view) & 15u) != 0 || (reinterpret_cast<uintptr_t>(out_dev) & 15u) != 0) return IRA_E_SIZE;
  hipStream_t st = (hipStream_t)stream;
  const int per = format == 0 ? 4 : 8;                        // samples per 16-byte piece
  const long long n16 = count / per;
  int wg = workgroups > 0 ? workgroups : 8;
  if (wg > 1024) wg = 1024;
  if (n16 > 0) {
    const long long need = (n16 + (long long)PULL_THREADS * PULL_U - 1) / ((long long)PULL_THREADS * PULL_U);
    if (need < wg) wg = (int)need;
    if (format == 0) host_pull_kernel<0><<<wg, PULL_THREADS, 0, st>>>(static_cast<const pull_i4*>(dev_view), out_dev, n16);
    else host_pull_kernel<1><<<wg, PULL_THREADS, 0, st>>>(static_cast<const pull_i4*>(dev_view), out_dev, n16);
  }
  if (n16 * per < count) {
    if (format == 0) host_pull_tail_kernel<0><<<1, 64, 0, st>>>(dev_view, out_dev, n16 * per, count);
    else host_pull_tail_kernel<1><<<1, 64, 0, st>>>(dev_view, out_dev, n16 * per, count);
  }
  IRA_RETURN_LAUNCH();
}
