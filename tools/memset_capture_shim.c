/* LD_PRELOAD shim: reports every hipMemset*Async issued on a stream that is being captured into a hipGraph.
 *
 * On ROCm 7.2 a memset NODE of an instantiated hipGraph clears its range only in the graph's first launch
 * (tools/graph_memset_probe.py), so any library op that zero-fills through hipMemsetAsync inside a capture (ATen's multi-workgroup
 * reductions, MIOpen solvers that accumulate with atomics) is wrong from the second replay on.  tools/memset_capture_audit.py runs
 * every plugin's captured inner loop under this shim and lists the callers.
 *
 *   gcc -O1 -shared -fPIC -o /tmp/memset_capture_shim.so tools/memset_capture_shim.c -ldl
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>

typedef int (*is_capturing_fn)(void*, int*);

/* the HIP runtime arrives through python's dlopen of torch (not in the global search order RTLD_NEXT walks): take the handle of
 * the copy that is already loaded */
static void* real_symbol(const char* name) {
  static void* hip = NULL;
  if (!hip) hip = dlopen("libamdhip64.so", RTLD_NOW | RTLD_NOLOAD);
  void* fn = hip ? dlsym(hip, name) : dlsym(RTLD_NEXT, name);
  if (!fn) {
    fprintf(stderr, "[memset-in-capture] cannot resolve %s\n", name);
    abort();
  }
  return fn;
}

static int capturing(void* stream) {
  static is_capturing_fn fn = NULL;
  if (!fn) fn = (is_capturing_fn)real_symbol("hipStreamIsCapturing");
  int status = 0;
  if (!fn || fn(stream, &status) != 0) return 0;
  return status == 1; /* hipStreamCaptureStatusActive */
}

static void report(const char* what, size_t bytes) {
  void* frames[24];
  const int n = backtrace(frames, 24);
  char** names = backtrace_symbols(frames, n);
  fprintf(stderr, "[memset-in-capture] %s %zu bytes\n", what, bytes);
  for (int i = 2; names && i < n && i < 12; ++i) fprintf(stderr, "[memset-in-capture]    %s\n", names[i]);
  free(names);
}

#define FORWARD(name, proto, args, bytes)                      \
  int name proto {                                             \
    static int (*real) proto = NULL;                           \
    if (!real) real = (int(*) proto)real_symbol(#name);        \
    if (capturing(stream)) report(#name, (size_t)(bytes));     \
    return real args;                                          \
  }

FORWARD(hipMemsetAsync, (void* dst, int value, size_t n, void* stream), (dst, value, n, stream), n)
FORWARD(hipMemsetD8Async, (void* dst, unsigned char value, size_t n, void* stream), (dst, value, n, stream), n)
FORWARD(hipMemsetD16Async, (void* dst, unsigned short value, size_t n, void* stream), (dst, value, n, stream), 2 * n)
FORWARD(hipMemsetD32Async, (void* dst, int value, size_t n, void* stream), (dst, value, n, stream), 4 * n)
FORWARD(hipMemset2DAsync, (void* dst, size_t pitch, int value, size_t w, size_t h, void* stream), (dst, pitch, value, w, h, stream), w * h)
