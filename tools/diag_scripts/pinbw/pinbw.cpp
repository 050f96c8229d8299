// How fast do T host threads fill a pinned staging buffer? (gfbe_batch_upload packs 0.44 MB per window into hipHostMalloc memory on 24
// threads: 8.3 ms per 1024 windows = 54 GB/s.)  hipcc -O2 -pthread pinbw.cpp -o pinbw ; ./pinbw [threads] [MB]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double fill(char *buf, size_t bytes, int T, int mode) {
  const size_t win = 440 * 1024;
  const size_t nwin = bytes / win;
  auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  for (int t = 0; t < T; t++)
    th.emplace_back([=]() {
      for (size_t w = t; w < nwin; w += T) {
        double *o = (double *)(buf + w * win);
        const size_t n = win / 8;
        if (mode == 0) memset(o, 1, win);
        else {      // scattered 16-byte records: 10 streams advancing in turn (a landmark's factors go to ten pair regions)
          const size_t per = n / 10 / 2;
          for (size_t i = 0; i < per; i++)
            for (int s = 0; s < 10; s++) { double *f = o + ((size_t)s * per + i) * 2; f[0] = (double)i; f[1] = (double)s; }
        }
      }
    });
  for (auto &x : th) x.join();
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
int main(int argc, char **argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 24;
  const size_t bytes = (size_t)(argc > 2 ? atoi(argv[2]) : 451) << 20;
  struct { const char *name; unsigned flags; int kind; } kinds[] = {
    {"malloc (pageable)", 0, 0}, {"hipHostMalloc default", hipHostMallocDefault, 1}, {"hipHostMalloc non-coherent", hipHostMallocNonCoherent, 1},
    {"hipHostMalloc portable|mapped", hipHostMallocPortable | hipHostMallocMapped, 1}, {"hipHostMalloc numa-user", hipHostMallocNumaUser, 1},
    {"malloc + hipHostRegister", hipHostRegisterDefault, 2}};
  for (auto &k : kinds) {
    char *p = nullptr;
    if (k.kind == 0) p = (char *)aligned_alloc(4096, bytes);
    else if (k.kind == 1) { if (hipHostMalloc((void **)&p, bytes, k.flags) != hipSuccess) { printf("%-32s failed\n", k.name); (void)hipGetLastError(); continue; } }
    else { p = (char *)aligned_alloc(4096, bytes); memset(p, 0, bytes); if (hipHostRegister(p, bytes, k.flags) != hipSuccess) { printf("%-32s register failed\n", k.name); continue; } }
    fill(p, bytes, T, 0);      // first touch
    double m0 = 1e9, m1 = 1e9;
    for (int r = 0; r < 5; r++) { m0 = std::min(m0, fill(p, bytes, T, 0)); m1 = std::min(m1, fill(p, bytes, T, 1)); }
    printf("%-32s %2d threads, %zu MB: memset %.2f ms (%.0f GB/s), scattered 16 B records %.2f ms (%.0f GB/s)\n", k.name, T, bytes >> 20, m0, bytes / m0 / 1e6, m1, bytes / m1 / 1e6);
    if (k.kind == 0) free(p); else if (k.kind == 1) (void)hipHostFree(p); else { (void)hipHostUnregister(p); free(p); }
  }
  return 0;
}
