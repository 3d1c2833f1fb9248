// ubench_hostreg.cpp -- what it costs to get 400 MB of final values into a caller's std::vector:
// page-faulting the vector, hipHostRegister / hipHostUnregister of it, and the D2H copy into
// pageable vs registered memory.  Decides the pinning policy of smmc_engine_simulate_to_host.
// build: hipcc -O2 -o /tmp/ubench_hostreg tools/ubench_hostreg.cpp ; run on the GPU box
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include <sys/mman.h>
#include <unistd.h>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
  const size_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 100000000ull;
  const size_t bytes = n * sizeof(float);
  double t0 = now();
  CK(hipFree(nullptr));
  std::printf("hip runtime init            %8.2f ms\n", (now() - t0) * 1e3);
  float *d = nullptr;
  CK(hipMalloc(reinterpret_cast<void **>(&d), bytes));
  CK(hipMemset(d, 0x3f, bytes));
  CK(hipDeviceSynchronize());
  for (int rep = 0; rep < 2; ++rep) {
    t0 = now();
    std::vector<float> v(n);  // value-initialised like totals.resize(N): every page touched
    std::printf("[%d] vector(%zu) zero-fill     %8.2f ms\n", rep, n, (now() - t0) * 1e3);
    t0 = now();
    CK(hipMemcpy(v.data(), d, bytes, hipMemcpyDeviceToHost));
    double dt = now() - t0;
    std::printf("[%d] D2H pageable             %8.2f ms  %.1f GB/s\n", rep, dt * 1e3, bytes / dt / 1e9);
    t0 = now();
    CK(hipHostRegister(v.data(), bytes, hipHostRegisterDefault));
    double treg = now() - t0;
    std::printf("[%d] hipHostRegister          %8.2f ms  %.1f GB/s\n", rep, treg * 1e3, bytes / treg / 1e9);
    t0 = now();
    CK(hipMemcpy(v.data(), d, bytes, hipMemcpyDeviceToHost));
    dt = now() - t0;
    std::printf("[%d] D2H registered           %8.2f ms  %.1f GB/s\n", rep, dt * 1e3, bytes / dt / 1e9);
    t0 = now();
    CK(hipHostUnregister(v.data()));
    std::printf("[%d] hipHostUnregister        %8.2f ms\n", rep, (now() - t0) * 1e3);
    // registration in 64 MiB pieces (what a chunk pipeline would do)
    const size_t piece = 64ull << 20;
    t0 = now();
    size_t k = 0;
    for (size_t off = 0; off < bytes; off += piece, ++k)
      CK(hipHostRegister(reinterpret_cast<char *>(v.data()) + off, std::min(piece, bytes - off), hipHostRegisterDefault));
    treg = now() - t0;
    std::printf("[%d] register in %zu x 64 MiB  %8.2f ms  (%.2f ms per piece)\n", rep, k, treg * 1e3, treg * 1e3 / k);
    for (size_t off = 0; off < bytes; off += piece) CK(hipHostUnregister(reinterpret_cast<char *>(v.data()) + off));
  }
  // sizing a fresh vector: plain resize vs reserve + MADV_POPULATE_WRITE from T threads + resize
  for (int threads : {0, 1, 2, 4, 8, 16}) {
    t0 = now();
    std::vector<float> v;
    double t_pop = 0;
    if (threads) {
      v.reserve(n);
      const long page = sysconf(_SC_PAGESIZE);
      const uintptr_t lo = (reinterpret_cast<uintptr_t>(v.data()) + page - 1) / page * page;
      const uintptr_t hi = reinterpret_cast<uintptr_t>(v.data() + n) / page * page;
      const size_t per = ((hi - lo) / threads + page - 1) / page * page;
      std::vector<std::thread> w;
      int rc_all = 0;
      for (int t = 0; t < threads; ++t) {
        const uintptr_t a = lo + t * per, b = std::min<uintptr_t>(hi, a + per);
        if (a < b) w.emplace_back([a, b, &rc_all] { if (madvise(reinterpret_cast<void *>(a), b - a, MADV_POPULATE_WRITE)) rc_all = 1; });
      }
      for (auto &x : w) x.join();
      t_pop = now() - t0;
      if (rc_all) std::printf("   (madvise failed)\n");
    }
    v.resize(n);
    std::printf("size vector: %2d populate threads: populate %7.2f ms, total %7.2f ms\n", threads, t_pop * 1e3, (now() - t0) * 1e3);
  }
  // untouched memory: register first, let the copy be the first touch
  t0 = now();
  float *raw = static_cast<float *>(std::malloc(bytes));
  CK(hipHostRegister(raw, bytes, hipHostRegisterDefault));
  double treg = now() - t0;
  std::printf("malloc + register untouched %8.2f ms\n", treg * 1e3);
  t0 = now();
  CK(hipMemcpy(raw, d, bytes, hipMemcpyDeviceToHost));
  std::printf("D2H into it                 %8.2f ms\n", (now() - t0) * 1e3);
  CK(hipHostUnregister(raw));
  std::free(raw);
  float *pinned = nullptr;
  t0 = now();
  CK(hipHostMalloc(reinterpret_cast<void **>(&pinned), bytes, hipHostMallocDefault));
  std::printf("hipHostMalloc               %8.2f ms\n", (now() - t0) * 1e3);
  t0 = now();
  CK(hipMemcpy(pinned, d, bytes, hipMemcpyDeviceToHost));
  double dt = now() - t0;
  std::printf("D2H hipHostMalloc'd         %8.2f ms  %.1f GB/s\n", dt * 1e3, bytes / dt / 1e9);
  CK(hipHostFree(pinned));
  CK(hipFree(d));
  return 0;
}
