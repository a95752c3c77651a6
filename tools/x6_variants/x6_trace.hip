// Section timing of sepconv_bwd_x6 (wave cycles, s_memtime): hipcc --offload-arch=gfx950 -O3 -DX6_TRACE -I include -I meta-interpolation_amd/csrc
//   -fno-slp-vectorize tools/x6_variants/x6_trace.hip -o /tmp/x6_trace   (builds the half-phase-skewed variant next to it, which carries the X6_T probes)
#include "sepconv_x6_skewed_schedule.hip.txt"
#include <cstdio>
#include <vector>
int main() {
  const int B = 8, Ho = 256, Wo = 448, K = 51, C = 3, Hi = Ho + K - 1, Wi = Wo + K - 1;
  size_t nin = (size_t)B * C * Hi * Wi, nt = (size_t)B * K * Ho * Wo, ng = (size_t)B * C * Ho * Wo;
  std::vector<float> hin(nin), ht(nt), hg(ng);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / (1 << 24) - 0.5f; };
  for (auto& x : hin) x = rnd();
  for (auto& x : ht) x = rnd() * 0.3f;
  for (auto& x : hg) x = rnd();
  float *in, *v, *h, *gO, *gV, *gH;
  hipMalloc(&in, nin * 4); hipMalloc(&v, nt * 4); hipMalloc(&h, nt * 4); hipMalloc(&gO, ng * 4); hipMalloc(&gV, nt * 4); hipMalloc(&gH, nt * 4);
  hipMemcpy(in, hin.data(), nin * 4, hipMemcpyHostToDevice); hipMemcpy(v, ht.data(), nt * 4, hipMemcpyHostToDevice);
  hipMemcpy(h, ht.data(), nt * 4, hipMemcpyHostToDevice); hipMemcpy(gO, hg.data(), ng * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) {
    hipEventRecord(e0);
    int rc = savfi_sepconv_bwd_x6_launch(in, v, h, gO, gV, gH, B, Ho, Wo, 256, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("launch rc=%d %.1f us\n", rc, ms * 1e3);
  }
  unsigned long long t[8 * 16];
  hipMemcpyFromSymbol(t, HIP_SYMBOL(x6_trace_buf), sizeof(t));
  const char* names[12] = {"loop top", "pos/slide-load/readlanes", "bq(h)+prefetch h,g", "v-table write", "prefetch v + gV tail", "gV MFMA loop",
                           "gV epilogue", "bq(v)+gH tail prep", "gH MFMA loop", "gH epilogue", "h-table write", "slide write"};
  for (int w = 0; w < 8; ++w) {
    unsigned long long tot = 0;
    for (int k = 0; k < 12; ++k) tot += t[w * 16 + k];
    printf("wave %d total %llu:", w, tot);
    for (int k = 0; k < 12; ++k) printf(" %llu", t[w * 16 + k]);
    printf("\n");
  }
  for (int k = 0; k < 12; ++k) printf("%2d %s\n", k, names[k]);
  return 0;
}
