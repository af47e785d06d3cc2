// PCIe copy rates on the GPU box: pageable vs pinned host memory, and what pinning a caller's buffer costs.
// Build: hipcc -O3 tools/pcie_bench.hip -o tools/pcie_bench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t sizes[] = {(size_t)8 << 20, (size_t)32 << 20, (size_t)128 << 20};
  void* dev; CK(hipMalloc(&dev, (size_t)128 << 20));
  for (size_t bytes : sizes) {
    char* pageable = (char*)malloc(bytes); memset(pageable, 1, bytes);
    char* pinned; CK(hipHostMalloc((void**)&pinned, bytes, hipHostMallocDefault)); memset(pinned, 1, bytes);
    auto rate = [&](void* dst, const void* src, hipMemcpyKind k) { double best = 1e9; for (int r = 0; r < 4; ++r) { double t = now(); hipMemcpy(dst, src, bytes, k); best = std::min(best, now() - t); } return bytes / best / 1e9; };
    printf("%4zu MiB  pageable H2D %6.1f GB/s  D2H %6.1f GB/s | pinned H2D %6.1f GB/s  D2H %6.1f GB/s", bytes >> 20,
           rate(dev, pageable, hipMemcpyHostToDevice), rate(pageable, dev, hipMemcpyDeviceToHost),
           rate(dev, pinned, hipMemcpyHostToDevice), rate(pinned, dev, hipMemcpyDeviceToHost));
    double t = now(); CK(hipHostRegister(pageable, bytes, hipHostRegisterDefault)); double treg = now() - t;
    double r1 = rate(dev, pageable, hipMemcpyHostToDevice);
    t = now(); CK(hipHostUnregister(pageable)); double tun = now() - t;
    printf(" | hipHostRegister %.2f ms, unregister %.2f ms, registered H2D %.1f GB/s", treg * 1e3, tun * 1e3, r1);
    // CPU memcpy pageable -> pinned, 1 and 4 threads
    for (int th : {1, 4}) {
      double best = 1e9;
      for (int r = 0; r < 3; ++r) {
        double t0 = now();
        std::vector<std::thread> ts;
        for (int i = 0; i < th; ++i) ts.emplace_back([&, i] { size_t per = bytes / th; memcpy(pinned + i * per, pageable + i * per, per); });
        for (auto& x : ts) x.join();
        best = std::min(best, now() - t0);
      }
      printf(" | memcpy x%d %.1f GB/s", th, bytes / best / 1e9);
    }
    printf("\n");
    free(pageable); CK(hipHostFree(pinned));
  }
  return 0;
}
