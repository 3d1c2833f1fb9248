// ubench_store_pattern.hip -- what HBM write rate does keepdata's access pattern allow?
// (development tool)  Every wave owns 64 consecutive rows of `row_len` floats and writes them
//   A: tile by tile, `tile` aligned floats per row per visit (keepdata's pattern), rows 4*row_len B apart
//   B: row after row, each row start to end (contiguous per wave)
// Pure stores of a constant: no compute, no LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int TILE>
__global__ __launch_bounds__(256) void pattern_a(float *out, unsigned long long n_rows, unsigned row_len) {
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long n_groups = (n_rows + 63) / 64;
  constexpr int RPS = 64 / TILE;
  const unsigned sub = lane / TILE, col = lane % TILE;
  const unsigned long long base_f = reinterpret_cast<uintptr_t>(out) >> 2;
  for (unsigned long long g = (unsigned long long)blockIdx.x * 4 + wave; g < n_groups; g += (unsigned long long)gridDim.x * 4) {
    const unsigned long long row0 = g * 64;
    const unsigned n_tiles = (row_len + 2 * (TILE - 1)) / TILE;
    for (unsigned t = 0; t < n_tiles; ++t) {
      unsigned long long row_off = (row0 + sub) * row_len;
      for (unsigned r = 0; r < 64; r += RPS) {
        const unsigned phi = (unsigned)(base_f + row_off) & (TILE - 1);
        const unsigned s = t * TILE + col - phi;
        if (row0 + r + sub < n_rows && s < row_len) out[row_off + s] = 1.0f + s;
        row_off += (unsigned long long)RPS * row_len;
      }
    }
  }
}

// C: keepdata WITHOUT per-lane phase: tile t of every row is the 32 floats [1 + 32 t, 33 + 32 t)
// of that row, wherever they fall (each piece straddles two lines; the leading part completes the
// line the previous tile left partial).  `spin` emulates the compute between two tiles of a wave
// and `lds_bytes` (dynamic LDS) limits the waves per CU: does L2 merge the partial lines when the
// in-flight footprint (waves x 64 rows x 128 B) fits?
__global__ __launch_bounds__(256) void pattern_c(float *out, unsigned long long n_rows, unsigned row_len, int spin) {
  extern __shared__ float lds[];
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long n_groups = (n_rows + 63) / 64;
  const unsigned sub = lane / 32, col = lane % 32;
  float acc = 1.0f + lane;
  for (unsigned long long g = (unsigned long long)blockIdx.x * 4 + wave; g < n_groups; g += (unsigned long long)gridDim.x * 4) {
    const unsigned long long row0 = g * 64;
    const unsigned n_tiles = (row_len + 31) / 32;
    for (unsigned t = 0; t < n_tiles; ++t) {
      for (int i = 0; i < spin; ++i) acc = acc * 1.0000001f + 1e-7f;  // "compute"
      unsigned long long row_off = (row0 + sub) * row_len;
      const unsigned s = t * 32 + col;
      for (unsigned r = 0; r < 64; r += 2) {
        if (row0 + r + sub < n_rows && s < row_len) out[row_off + s] = acc;
        row_off += 2ull * row_len;
      }
    }
  }
  if (acc == 12345.678f) lds[threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void pattern_b(float *out, unsigned long long n_rows, unsigned row_len) {
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long n_groups = (n_rows + 63) / 64;
  for (unsigned long long g = (unsigned long long)blockIdx.x * 4 + wave; g < n_groups; g += (unsigned long long)gridDim.x * 4) {
    const unsigned long long first = g * 64 * row_len;
    unsigned long long last = (g * 64 + 64) * row_len;
    if (last > n_rows * row_len) last = n_rows * row_len;
    for (unsigned long long i = first + lane; i < last; i += 64) out[i] = 1.0f + (float)lane;
  }
}

// D: the COMB pattern.  A wave owns 64 streams of K consecutive rows; stream l starts at row
// super * 2048 + 32 l + w K (w = the wave's slot in its 2048-row super-chunk), so all 64 streams have
// the same phase modulo a 128-byte line.  Each stream is written window by window (one aligned 128-byte
// line per visit, 8 streams x 128 B per wave-wide 16-byte store), starting at its first WHOLE line and
// running through the line in which its last row ends (which holds the head of the next row): every
// line is written whole, exactly once, no line is shared between two writers.
__global__ __launch_bounds__(256) void pattern_d(float *out, unsigned long long n_rows, unsigned row_len, unsigned K) {
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned sub = lane / 8, quad = lane % 8;
  const unsigned long long base_f = reinterpret_cast<uintptr_t>(out) >> 2;
  const unsigned waves_per_super = 32 / K;
  const unsigned long long n_chunks = (n_rows / 2048) * waves_per_super;
  for (unsigned long long c = (unsigned long long)blockIdx.x * 4 + wave; c < n_chunks; c += (unsigned long long)gridDim.x * 4) {
    const unsigned long long super = c / waves_per_super;
    const unsigned w = (unsigned)(c % waves_per_super);
    const unsigned long long row0 = super * 2048 + (unsigned long long)w * K;  // stream 0's first row
    const unsigned phi = (unsigned)(base_f + row0 * row_len) & 31u;             // the same for every stream
    const unsigned first_t = (phi != 0 && row0 != 0) ? 1u : 0u;
    const unsigned n_t = (phi + K * row_len + 31) / 32;
    for (unsigned t = first_t; t < n_t; ++t) {
#pragma unroll
      for (unsigned it = 0; it < 8; ++it) {
        const unsigned l = sub + 8 * it;
        const long long a = (long long)((row0 + 32ull * l) * row_len) - phi + 32ll * t + 4 * quad;
        if (a >= 0 && (unsigned long long)a + 4 <= n_rows * row_len)
          *reinterpret_cast<float4 *>(out + a) = make_float4(1.0f + t, 2.0f, 3.0f, 4.0f + l);
      }
    }
  }
}

// E: the comb pattern with LINES consecutive 128-byte lines per stream per visit (a tile of 32 * LINES columns): does the
// memory system prefer longer contiguous pieces per stream?  64 / (8 * LINES) streams per wave-wide 16-byte store.
template <int LINES>
__global__ __launch_bounds__(256) void pattern_e(float *out, unsigned long long n_rows, unsigned row_len) {
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr unsigned LPS = 8 * LINES;          // lanes per stream per store
  constexpr unsigned SPI = 64 / LPS;           // streams per store instruction
  const unsigned sub = lane / LPS, quad = lane % LPS;
  const unsigned long long base_f = reinterpret_cast<uintptr_t>(out) >> 2;
  const unsigned long long n_chunks = (n_rows / 2048) * 32;
  for (unsigned long long c = (unsigned long long)blockIdx.x * 4 + wave; c < n_chunks; c += (unsigned long long)gridDim.x * 4) {
    const unsigned long long row0 = (c / 32) * 2048 + (c % 32);
    const unsigned phi = (unsigned)(base_f + row0 * row_len) & 31u;
    const unsigned first_t = (phi != 0 && row0 != 0) ? 1u : 0u;
    const unsigned n_t = (phi + row_len + 31) / 32;
    for (unsigned t = first_t; t < n_t; t += LINES) {
#pragma unroll
      for (unsigned it = 0; it < 64 / SPI; ++it) {
        const unsigned l = sub + SPI * it;
        const long long a = (long long)((row0 + 32ull * l) * row_len) - phi + 32ll * t + 4 * quad;
        if (t + quad / 8 < n_t && a >= 0 && (unsigned long long)a + 4 <= n_rows * row_len)
          *reinterpret_cast<float4 *>(out + a) = make_float4(1.0f + t, 2.0f, 3.0f, 4.0f + l);
      }
    }
  }
}

// H: the comb pattern cut in TIME: one launch per tile column t, every wave stores ONE tile (one line of each of its 64
// streams: 8 wave-wide stores) and ends -- what a trajectory kernel would look like if a wave handed its 64 running totals
// on to the next launch instead of living for a whole row.  Round 4: do short-lived waves stream better here too?
__global__ __launch_bounds__(256) void pattern_h(float *out, unsigned long long n_rows, unsigned row_len, unsigned t) {
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned sub = lane / 8, quad = lane % 8;
  const unsigned long long base_f = reinterpret_cast<uintptr_t>(out) >> 2;
  const unsigned long long n_chunks = (n_rows / 2048) * 32;
  const unsigned long long c = (unsigned long long)blockIdx.x * 4 + wave;
  if (c >= n_chunks) return;
  const unsigned long long row0 = (c / 32) * 2048 + (c % 32);
  const unsigned phi = (unsigned)(base_f + row0 * row_len) & 31u;
  const unsigned first_t = (phi != 0 && row0 != 0) ? 1u : 0u;
  const unsigned n_t = (phi + row_len + 31) / 32;
  if (t < first_t || t >= n_t) return;
#pragma unroll
  for (unsigned it = 0; it < 8; ++it) {
    const unsigned l = sub + 8 * it;
    const long long a = (long long)((row0 + 32ull * l) * row_len) - phi + 32ll * t + 4 * quad;
    if (a >= 0 && (unsigned long long)a + 4 <= n_rows * row_len)
      *reinterpret_cast<float4 *>(out + a) = make_float4(1.0f + t, 2.0f, 3.0f, 4.0f + l);
  }
}

// I: the comb pattern in HALF lines: a visit writes 64 aligned bytes of each stream (4 lanes x 16 B, 16 streams per store), the
// other half of the line comes with the same wave's next visit -- what a tile of 16 columns (half the LDS, twice the waves)
// would store.  Every line is still written whole and by one wave, but in two pieces some microseconds apart.
__global__ __launch_bounds__(256) void pattern_i(float *out, unsigned long long n_rows, unsigned row_len, int spin) {
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned sub = lane / 4, quad = lane % 4;
  const unsigned long long base_f = reinterpret_cast<uintptr_t>(out) >> 2;
  const unsigned long long n_chunks = (n_rows / 2048) * 32;
  float acc = 1.0f + lane;
  for (unsigned long long c = (unsigned long long)blockIdx.x * 4 + wave; c < n_chunks; c += (unsigned long long)gridDim.x * 4) {
    const unsigned long long row0 = (c / 32) * 2048 + (c % 32);
    const unsigned phi = (unsigned)(base_f + row0 * row_len) & 31u;
    const unsigned first_h = (phi != 0 && row0 != 0) ? 2u : 0u;      // halves: skip the whole partial first line
    const unsigned n_h = 2u * ((phi + row_len + 31) / 32);
    for (unsigned h = first_h; h < n_h; ++h) {
      for (int i = 0; i < spin; ++i) acc = __builtin_fmaf(acc, 1.0000001f, 0.5f);  // the compute between two visits
#pragma unroll
      for (unsigned it = 0; it < 4; ++it) {
        const unsigned l = sub + 16 * it;
        const long long a = (long long)((row0 + 32ull * l) * row_len) - phi + 16ll * h + 4 * quad;
        if (a >= 0 && (unsigned long long)a + 4 <= n_rows * row_len)
          *reinterpret_cast<float4 *>(out + a) = make_float4(acc, 2.0f, 3.0f, 4.0f + l);
      }
    }
  }
}

// F: a plain fill, 16 bytes per lane, a wave's store = 1 KiB contiguous, the grid strides over the array (ATen's shape)
__global__ __launch_bounds__(256) void pattern_f(float4 *out, unsigned long long n4) {
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (unsigned long long)gridDim.x * 256)
    out[i] = make_float4(1.0f, 2.0f, 3.0f, 4.0f);
}
// G: the same, one workgroup per 4 KiB piece and no loop (a grid of n4 / 256 workgroups)
__global__ __launch_bounds__(256) void pattern_g(float4 *out, unsigned long long n4) {
  const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) out[i] = make_float4(1.0f, 2.0f, 3.0f, 4.0f);
}

int main(int argc, char **argv) {
  const unsigned long long n_rows = 4000000;
  const unsigned row_len = argc > 1 ? atoi(argv[1]) : 361;
  float *d;
  CK(hipMalloc(&d, n_rows * row_len * 4ull + 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double bytes = 4.0 * n_rows * row_len;
  for (int bpc : {4, 8, 16}) {
    for (int v = 0; v < 3; ++v) {
      float ms = 0;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        if (v == 0) hipLaunchKernelGGL(pattern_a<32>, dim3(256 * bpc), dim3(256), 0, 0, d, n_rows, row_len);
        if (v == 1) hipLaunchKernelGGL(pattern_a<64>, dim3(256 * bpc), dim3(256), 0, 0, d, n_rows, row_len);
        if (v == 2) hipLaunchKernelGGL(pattern_b, dim3(256 * bpc), dim3(256), 0, 0, d, n_rows, row_len);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
      }
      printf("row_len=%u bpc=%d %s: %.3f ms  %.0f GB/s\n", row_len, bpc,
             v == 0 ? "A tile=32 (128 B aligned pieces per row)" : v == 1 ? "A tile=64 (256 B aligned pieces per row)" : "B contiguous per wave", ms,
             bytes / ms / 1e6);
    }
  }
  for (unsigned K : {1u, 2u, 4u, 8u, 16u}) {
    for (int bpc : {4, 8, 16}) {
      float ms = 0;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(pattern_d, dim3(256 * bpc), dim3(256), 0, 0, d, n_rows, row_len, K);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
      }
      printf("row_len=%u bpc=%d D comb, %u rows per stream (whole lines only): %.3f ms  %.0f GB/s\n", row_len, bpc, K, ms,
             4.0 * (n_rows / 2048 * 2048) * row_len / ms / 1e6);
    }
  }
  for (int lines : {1, 2, 4, 8}) {
    for (int bpc : {4, 8}) {
      float ms = 0;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        if (lines == 1) hipLaunchKernelGGL(pattern_e<1>, dim3(256 * bpc), dim3(256), 0, 0, d, n_rows, row_len);
        if (lines == 2) hipLaunchKernelGGL(pattern_e<2>, dim3(256 * bpc), dim3(256), 0, 0, d, n_rows, row_len);
        if (lines == 4) hipLaunchKernelGGL(pattern_e<4>, dim3(256 * bpc), dim3(256), 0, 0, d, n_rows, row_len);
        if (lines == 8) hipLaunchKernelGGL(pattern_e<8>, dim3(256 * bpc), dim3(256), 0, 0, d, n_rows, row_len);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
      }
      printf("row_len=%u bpc=%d E comb, %d lines per stream per visit: %.3f ms  %.0f GB/s\n", row_len, bpc, lines, ms,
             4.0 * (n_rows / 2048 * 2048) * row_len / ms / 1e6);
    }
  }
  for (int bpc : {4, 8}) {
    for (int spin : {0, 100, 400}) {
      float ms = 0;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(pattern_i, dim3(256 * bpc), dim3(256), 0, 0, d, n_rows, row_len, spin);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
      }
      printf("row_len=%u bpc=%d I comb in half lines, %d fma between visits: %.3f ms  %.0f GB/s\n", row_len, bpc, spin, ms,
             4.0 * (n_rows / 2048 * 2048) * row_len / ms / 1e6);
    }
  }
  {
    const unsigned long long n_chunks = (n_rows / 2048) * 32;
    const unsigned n_t = (31 + row_len + 31) / 32;
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      for (unsigned t = 0; t < n_t; ++t)
        hipLaunchKernelGGL(pattern_h, dim3((unsigned)((n_chunks + 3) / 4)), dim3(256), 0, 0, d, n_rows, row_len, t);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("row_len=%u H comb cut in time, %u launches of one-tile waves: %.3f ms  %.0f GB/s\n", row_len, n_t, ms,
           4.0 * (n_rows / 2048 * 2048) * row_len / ms / 1e6);
  }
  {
    const unsigned long long n4 = n_rows * row_len / 4;
    for (int bpc : {4, 8, 16, 0}) {
      float ms = 0;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        if (bpc) hipLaunchKernelGGL(pattern_f, dim3(256 * bpc), dim3(256), 0, 0, reinterpret_cast<float4 *>(d), n4);
        else hipLaunchKernelGGL(pattern_g, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, reinterpret_cast<float4 *>(d), n4);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
      }
      printf("row_len=%u bpc=%d %s: %.3f ms  %.0f GB/s\n", row_len, bpc, bpc ? "F fill, 16 B per lane, grid stride" : "G fill, one 4 KiB piece per workgroup", ms, bytes / ms / 1e6);
    }
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      CK(hipMemsetAsync(d, 0, n_rows * row_len * 4ull, 0));
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("row_len=%u hipMemsetAsync: %.3f ms  %.0f GB/s\n", row_len, ms, bytes / ms / 1e6);
  }
  if (argc > 2) return 0;  // a second argument: skip the unaligned-piece sweeps
  for (int lds_kb : {8, 19, 39, 79}) {     // 160 KiB / lds -> 20(cap 8), 8, 4, 2 workgroups per CU
    for (int spin : {0, 2000, 8000}) {
      float ms = 0;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(pattern_c, dim3(256 * 16), dim3(256), lds_kb * 1024, 0, d, n_rows, row_len, spin);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
      }
      printf("row_len=%u C unaligned 128 B pieces, lds=%d KiB/wg spin=%d: %.3f ms  %.0f GB/s\n", row_len, lds_kb, spin, ms, bytes / ms / 1e6);
    }
  }
  return 0;
}
