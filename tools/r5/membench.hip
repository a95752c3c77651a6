// HBM read+write throughput of chunked streams on MI355X: every wave copies CHUNK-byte pieces (src -> dst) in one of several visiting orders.
//   hipcc --offload-arch=gfx950 -O3 tools/r5/membench.hip -o tools/scratch/membench && tools/scratch/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// pattern 0: sweep      -- chunk index = k * NW + w            (all waves advance through memory together)
// pattern 1: own region -- chunk index = w * K + k             (every wave streams through its own contiguous region)
// pattern 2: strided    -- workgroups at unrelated places: chunk index = (wg * K + k) * WPG' ... see code: the sepconv kernel's order
//                          (a workgroup's 8 waves = 2 adjacent chunks x 4 rows, rows `row` chunks apart, next phase 4 rows down)
// pattern 3: random     -- chunk index = hash(w, k)
template <int CHUNK>
__global__ __launch_bounds__(512) void copyk(const char* __restrict__ src, char* __restrict__ dst, int K, int pattern, long long nchunks, int row) {
  const int lane = threadIdx.x & 63, wl = threadIdx.x >> 6, wg = blockIdx.x, NWG = gridDim.x;
  const long long NW = (long long)NWG * 8, w = (long long)wg * 8 + wl;
  constexpr int NI = CHUNK / 1024;
  for (int k = 0; k < K; ++k) {
    long long ci;
    if (pattern == 0) ci = (long long)k * NW + w;
    else if (pattern == 1) ci = w * K + k;
    else if (pattern == 2) {
      // workgroup wg owns K phases of a strip: strip s = wg % S at phases [ (wg / S) * K, ... ) ; unit (y, xu): ci = y * row + xu
      const int S = row / 2;
      const int s = wg % S; const long long ph = (long long)(wg / S) * K + k;
      const long long y = ph * 4 + (wl >> 1);
      ci = y * row + 2 * s + (wl & 1);
    } else {
      unsigned long long h = (unsigned long long)w * 0x9E3779B97F4A7C15ull + (unsigned long long)k * 0xC2B2AE3D27D4EB4Full;
      h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
      ci = (long long)(h % (unsigned long long)nchunks);
    }
    if (ci >= nchunks) ci %= nchunks;
    const char* s = src + ci * CHUNK + lane * 16;
    char* d = dst + ci * CHUNK + lane * 16;
    f32x4 v[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) v[i] = *reinterpret_cast<const f32x4*>(s + i * 1024);
#pragma unroll
    for (int i = 0; i < NI; ++i) *reinterpret_cast<f32x4*>(d + i * 1024) = v[i];
  }
}
template <int CHUNK>
void run(const char* src, char* dst, size_t bytes, int pattern, const char* name) {
  const int NWG = 256; const long long NW = NWG * 8;
  const long long nchunks = bytes / CHUNK;
  const int K = (int)(nchunks / NW);
  const int row = 28;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(copyk<CHUNK>, dim3(NWG), dim3(512), 0, 0, src, dst, K, pattern, nchunks, row);
  hipEventRecord(a);
  const int R = 5;
  for (int i = 0; i < R; ++i) hipLaunchKernelGGL(copyk<CHUNK>, dim3(NWG), dim3(512), 0, 0, src, dst, K, pattern, nchunks, row);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= R;
  const double moved = 2.0 * (double)K * NW * CHUNK;
  printf("chunk %5d  %-10s  %8.1f us  %7.0f GB/s read+write\n", CHUNK, name, 1e3 * ms, moved / ms / 1e6);
}
int main() {
  const size_t bytes = 768ull << 20;
  char *src, *dst; hipMalloc(&src, bytes); hipMalloc(&dst, bytes); hipMemset(src, 1, bytes); hipMemset(dst, 0, bytes);
  const char* names[4] = {"sweep", "own-region", "sepconv", "random"};
  for (int p = 0; p < 4; ++p) { run<1024>(src, dst, bytes, p, names[p]); run<2048>(src, dst, bytes, p, names[p]); run<4096>(src, dst, bytes, p, names[p]); run<8192>(src, dst, bytes, p, names[p]); run<16384>(src, dst, bytes, p, names[p]); }
  return 0;
}
