// smmc_ref_kernels.hip -- the reference CPU engine's OWN random stream on gfx950 (SMMC_FLAG_STREAM_REF).
//
// The loop of mc_simulations (reference src/simulations.cpp:240-252), one path per lane:
//     std::mt19937 rng(seed);  std::uniform_int_distribution<int> uni(0, T - 1);
//     P x { total = update_fund(total, table[uni(rng)]); }
// with the seed interposed as (uint32_t)(seed0 + path id) where the reference reads a fresh
// std::random_device per path (it has no seed argument): oracle engine (R), oracle/smmc_oracle.c:143-180.
// uniform_int_distribution on mt19937 is libstdc++-11's Lemire map with rejection
// (/usr/include/c++/11/bits/uniform_int_dist.h:241-268), restated here as in the oracle and pinned
// to the real library by tests/golden/libstdcxx_random.json.
//
// mt19937 keeps 624 words of state per generator: 2496 bytes per PATH, 160 KiB per wave -- a CU's whole
// LDS.  But the state a fresh generator starts from is itself a cheap recurrence,
//     x[0] = seed,  x[i] = 1812433253 (x[i-1] ^ (x[i-1] >> 30)) + i          (i < 624)
// and output j is temper(x[624 + j]) with  x[n + 624] = x[n + 397] ^ tw(x[n], x[n + 1]).  So:
//
//   ref_windowed_kernel (paths of at most 454 outputs: the reference's 360-period runs): NO state in
//     memory.  Outputs j < 227 need x[j], x[j+1] and x[j+397], all seed words: two seed chains 397
//     apart, advanced one step per output (the far one is first run forward 397 steps).  Outputs
//     227 <= j < 454 need x[624 + (j - 227)], an earlier OUTPUT: it is generated again from two more
//     seed chains (at j - 227 and j + 170) rather than kept -- three chain steps per output, still
//     nothing stored, 30-odd VGPRs, full occupancy.
//   ref_tree_kernel (455 .. 1816 outputs: BASELINE configs[4]'s 1000 periods): the same idea carried on as the
//     recursion it is -- every operand that stops being a seed word is generated again from seed chains, shared
//     between the operands that walk the same words: six chains and seven made words per output at 1000 periods,
//     nine and seventeen at 1816, still nothing stored (see the section below).
//   ref_generic_kernel (any length): the classic circular 624-word state, held in a global-memory
//     workspace laid out [word][lane] so that every access of a wave is one coalesced 256-byte
//     line -- but only for GENERATED words: the seed words still come from chains (two below output
//     227, one up to output 623), so a path stores one word per output and loads none, one or two.
//     From output 624 on the words a batch of 8 outputs needs are loaded before the batch (none of
//     them is written inside it), so the loads' latency is paid once per 8 outputs.  One workgroup per
//     CU keeps the workspace (160 KiB per wave) inside the 256 MiB Infinity Cache.
//
// Rejections (T / 2^32 per draw: 2.6e-7 for the 1127-entry table) shift a path's later draws by one
// output.  Generation stays wave-uniform -- every lane generates output j at step j.  In the generic
// kernel a lane that rejected simply does not compound at that step and goes on past output P; the
// windowed and tree kernels only flag it: the path goes on the redo list that a small generic launch works
// off (so does a path that leaves the checked divide's window).
//
// Bound: VALU issue, like paths_kernel (DESIGN.md section 5 has the counts).  HBM sees 4 B per path.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "smmc_device.h"
#include "smmc_internal.h"

namespace smmc {
namespace {

using namespace dev;

constexpr uint32_t kMtN = 624, kMtM = 397, kMtLag = kMtN - kMtM;  // 227

// x[i] from x[i-1]: the seeding recurrence of ISO C++ [rand.eng.mers] (oracle mt_seed)
//
// i is wave-uniform.  The multiply-add is ONE v_mad_u64_u32 -- multiplier in a VGPR (`k`, set up once per kernel by
// mt_multiplier(): an instruction reads one scalar operand, and that is the 64-bit addend i), the low word of the
// result taken -- where hipcc, which knows that only the low word is wanted, emits v_mul_lo_u32 + v_add.  On gfx950
// that pair is much the slower: with 1250 chain steps per 360-period path the exchange took the windowed kernel
// from 27.4 to 22.5 ms (the same exchange for v_mul_hi_u32 in paths_kernel's table draw changed nothing).
__device__ __forceinline__ uint32_t mt_multiplier() {
  uint32_t k = 1812433253u;
  asm volatile("" : "+v"(k));
  return k;
}
__device__ __forceinline__ uint32_t mt_seed_step(uint32_t x, uint32_t i, uint32_t k) {
  const uint32_t t = x ^ (x >> 30);
  uint64_t d, carry;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry) : "v"(t), "v"(k), "s"(static_cast<uint64_t>(i)));
  return static_cast<uint32_t>(d);
}

// x[397] from x[0]: the 397-step run-up of the far chain, eight steps per trip (the trip count is hidden from the
// compiler, which otherwise unrolls all 397 steps: 16 KB of code and an SGPR pair per step's index).
__device__ __forceinline__ uint32_t mt_run_up(uint32_t seed, uint32_t k) {
  uint32_t x = seed, idx = 1, last = kMtM;
  asm("" : "+s"(last));
  for (; idx + 7u <= last; idx += 8u) {
#pragma unroll
    for (uint32_t t = 0; t < 8u; ++t) x = mt_seed_step(x, idx + t, k);
  }
  for (; idx <= last; ++idx) x = mt_seed_step(x, idx, k);
  return x;
}

// The three masks of the twist and the tempering, as VGPR operands of their v_bitop3_b32 (vgpr_const, smmc_device.h):
// set up once per kernel beside the multiplier.
struct MtConsts {
  uint32_t magic, temper_b, temper_c;
};
__device__ __forceinline__ MtConsts mt_consts() {
  MtConsts c;
  c.magic = vgpr_const(0x9908b0dfu);
  c.temper_b = vgpr_const(0x9d2c5680u);
  c.temper_c = vgpr_const(0xefc60000u);
  return c;
}

// tw(x[n], x[n+1]) of the twist  x[n + 624] = x[n + 397] ^ tw(x[n], x[n+1])  (oracle mt_twist):
// y = top bit of x[n] with the low 31 of x[n+1]; (y >> 1) ^ (y odd ? 0x9908b0df : 0).  The callers keep
// the halves h = x >> 1 of the words they walk over (each word is x[n+1] once and x[n] the step after):
// y >> 1 is then h[n+1] with its bit 30 taken from h[n] -- one v_bfi_b32, whose mask 0x40000000 is the
// inline constant 2.0.
__device__ __forceinline__ uint32_t mt_twist_term(uint32_t hn, uint32_t hn1, uint32_t xn1, const MtConsts &mc) {
  uint32_t ysh;  // (hn & 0x40000000) | (hn1 & ~0x40000000); left to itself hipcc takes three instructions
  asm("v_bfi_b32 %0, 2.0, %1, %2" : "=v"(ysh) : "v"(hn), "v"(hn1));
  const uint32_t odd = static_cast<uint32_t>(static_cast<int32_t>(xn1 << 31) >> 31);  // v_bfe_i32: all ones when odd
  return __builtin_amdgcn_bitop3_b32(odd, mc.magic, ysh, 0x6a);                       // (odd & magic) ^ (y >> 1)
}

// tempering (oracle mt_next); a ^ (b & c) is one v_bitop3_b32 (truth table 0x78)
__device__ __forceinline__ uint32_t mt_temper(uint32_t y, const MtConsts &mc) {
  y ^= y >> 11;
  y = __builtin_amdgcn_bitop3_b32(y, y << 7, mc.temper_b, 0x78);
  y = __builtin_amdgcn_bitop3_b32(y, y << 15, mc.temper_c, 0x78);
  y ^= y >> 18;
  return y;
}

// One generator output `g` (x[624 + j], untempered) offered to a path that still needs `need` draws: the
// Lemire map picks the table entry or rejects the output; an accepted one is a period
// (src/simulations.cpp:250).
template <bool kExactDiv>
__device__ __forceinline__ bool offer(const RefArgs &k, const float *lds_table, const MtConsts &mc, uint32_t g, float &total, uint32_t &need) {
  const uint32_t y = mt_temper(g, mc);
  const uint64_t prod = static_cast<uint64_t>(y) * k.table_len;
  const bool take = static_cast<uint32_t>(prod) >= k.reject_below && need != 0u;
  const float next = compound<kExactDiv>(total, lds_table[static_cast<uint32_t>(prod >> 32)]);
  total = take ? next : total;
  need -= take ? 1u : 0u;
  return take;
}

// The seed words a path's next output needs, as chains advanced in step with the output index j.
struct MtWindow {
  uint32_t seed, x397;  // x[0], x[397]: where the chains of the second stretch start
  uint32_t k;           // mt_multiplier()
  MtConsts mc;          // mt_consts()
  uint32_t ah, a1, a1h; // x[j] >> 1, x[j + 1], x[j + 1] >> 1
  uint32_t b;           // j < 227: x[j + 397];  j >= 227: x[j + 170]
  uint32_t ch, c1, c1h; // j >= 227: x[j - 227] >> 1, x[j - 226], x[j - 226] >> 1
};

__device__ __forceinline__ void window_enter_a(MtWindow &w) {
  w.ah = w.seed >> 1;
  w.a1 = mt_seed_step(w.seed, 1u, w.k);
  w.a1h = w.a1 >> 1;
  w.b = w.x397;
}
// `far` = j + 398, the index of the far chain's next word, is the caller's own scalar counter: formed
// as j + 398 it is a second v_add per step (398 is no inline constant), as far + t it is part of a v_add3.
__device__ __forceinline__ uint32_t window_next_a(MtWindow &w, uint32_t j, uint32_t far) {  // outputs 0 .. 226
  const uint32_t g = w.b ^ mt_twist_term(w.ah, w.a1h, w.a1, w.mc);
  w.ah = w.a1h;
  w.a1 = mt_seed_step(w.a1, j + 2u, w.k);
  w.a1h = w.a1 >> 1;
  w.b = mt_seed_step(w.b, far, w.k);
  return g;
}
__device__ __forceinline__ void window_enter_b(MtWindow &w) {
  w.ch = w.seed >> 1;
  w.c1 = mt_seed_step(w.seed, 1u, w.k);
  w.c1h = w.c1 >> 1;
  w.b = w.x397;
}
// `near` = j - 225 and `far` = j + 171: the next words of the two replayed chains (own counters, as above)
__device__ __forceinline__ uint32_t window_next_b(MtWindow &w, uint32_t j, uint32_t near, uint32_t far) {  // outputs 227 .. 453
  // x[j + 397] = x[624 + (j - 227)] = x[j + 170] ^ tw(x[j - 227], x[j - 226]): output j - 227 again
  const uint32_t g = xor3(w.b, mt_twist_term(w.ch, w.c1h, w.c1, w.mc), mt_twist_term(w.ah, w.a1h, w.a1, w.mc));
  w.ah = w.a1h;
  w.a1 = mt_seed_step(w.a1, j + 2u, w.k);
  w.a1h = w.a1 >> 1;
  w.ch = w.c1h;
  w.c1 = mt_seed_step(w.c1, near, w.k);
  w.c1h = w.c1 >> 1;
  w.b = mt_seed_step(w.b, far, w.k);
  return g;
}

// A path that rejects an output would use one output more than it has periods, and every later draw
// would come from the next output: its lane is flagged instead (one v_cmp whose result is OR-ed into a
// scalar mask) and the path is left to the generic kernel -- 1e-4 of the paths for the 1127-entry table
// at 360 periods.  That keeps the loop free of per-lane state: no accept/select, no draw counter.
//
// kTraj (mc_simulations_keepdata, src/simulations.cpp:139-202: the same generator per path, every value kept).
// Lanes own paths, so one period's values of a wave are a COLUMN of the path-major output, and rows are
// 4 (P + 1) bytes apart: they begin at arbitrary offsets inside a 128-byte line.  Round 3 wrote 32-period
// pieces of 64 ADJACENT rows wherever they fell: every line reached HBM as two partial writes 32 periods apart
// (measured, profiles/r04/bench_ref_traj_before.jsonl: 1.65 TB/s at 4e6 x 361 values, WRITE_SIZE 1.32 x the
// bytes stored).  Round 4 gives a wave the comb of keepdata_comb_kernel (smmc_kernels.hip):
//   * lane l of wave w of a 2048-row super-chunk runs the K CONSECUTIVE rows 2048 s + 32 l + K w ...
//     one after the other -- a stream of K (P + 1) contiguous floats.  32 rows are 32 (P + 1) floats, a whole
//     number of lines, so all 64 streams of a wave start at the SAME offset phi inside a line: where a line ends
//     is wave-uniform (a scalar column cursor, a scalar branch), no per-lane delay or predicate;
//   * a stream's values go to a column-major LDS tile ([32 columns][64 lanes + 1]: conflict-free both ways) and
//     every 32 columns the tile leaves as WHOLE aligned lines, 16 bytes per lane, 8 streams per store;
//   * only the first and the last line of a stream are partial (one junction per K rows instead of two
//     per row); a new generator per row costs nothing extra here -- every path starts one anyway (the Philox
//     kernel has to recompute the head of the following row to close its last line);
//   * 32 / K waves share a super-chunk: 8 / K workgroups of four waves.
constexpr uint32_t kTrajSuper = 2048;              // rows per super-chunk: 64 lanes x 32
// K, the consecutive rows per stream (RefArgs::traj_rows: 8, 4, 2 or 1), is the host's choice: 32 / K waves share a
// super-chunk, i.e. 8 / K workgroups of four waves -- fewer rows per stream are more, shorter work units (a launch
// of 4e6 rows is 1953 units at K = 8: not quite two rounds of the 1024 resident workgroups) against one more
// junction line per K rows.
constexpr uint32_t kTrajColStride = 65;            // words between two columns of a tile
// A line is checked for completion once per TRIP of the period loops (four outputs), not per value -- the store
// phase is ~150 instructions, and inlined behind every value of every stretch of the tree kernel it made 365 KB
// of code and spilled -- so a tile holds up to 3 columns of the next line as well.
constexpr uint32_t kTrajTrip = 4;
constexpr uint32_t kTrajTileCols = 32 + kTrajTrip - 1;
constexpr uint32_t kTrajTileWords = kTrajTileCols * kTrajColStride;  // per wave

struct TrajWriter {
  // wave-uniform
  uint32_t col = 0;      // column of the next value: its float offset inside its 128-byte line
  uint32_t win = 0;      // lines of the stream already stored
  uint32_t phi = 0;      // column of the stream's first value
  uint32_t stream0 = 0;  // first row of stream 0 (lane 0)
  uint32_t n_rows = 0, row_len = 0, stream_len = 0, K = 0;
  bool whole = false;    // every stream of this wave lies inside the launch
  char *line0 = nullptr; // window 0 of stream 0
  // per lane
  float *mine = nullptr, *wp = nullptr;  // this lane's word of column 0 / of the next column
  const float *take = nullptr;           // store phase: columns 4 quad .. + 3 of stream sub (+ 8 per iteration)
  uint32_t lane_off = 0, sub = 0, quad = 0;

  __device__ __forceinline__ void init(float *tile, uint32_t n_rows_, uint32_t n_periods, uint32_t rows_per_stream) {
    const uint32_t lane = threadIdx.x & 63u;
    sub = lane >> 3;
    quad = lane & 7u;
    mine = tile + lane;
    take = tile + (4u * quad) * kTrajColStride + sub;
    n_rows = n_rows_;
    row_len = n_periods + 1u;
    K = rows_per_stream;
    stream_len = K * row_len;
    lane_off = sub * (128u * row_len) + 16u * quad;  // 32 rows of 4 row_len bytes between two streams
  }
  // the wave's 64 streams begin at rows first_row + 32 lane
  __device__ __forceinline__ void begin(float *d_traj, uint32_t first_row) {
    stream0 = first_row;
    const uint64_t first_f = (reinterpret_cast<uintptr_t>(d_traj) >> 2) + static_cast<uint64_t>(first_row) * row_len;
    phi = static_cast<uint32_t>(first_f) & 31u;
    line0 = reinterpret_cast<char *>(d_traj) + (static_cast<int64_t>(static_cast<uint64_t>(first_row) * row_len) - phi) * 4;
    whole = static_cast<uint64_t>(first_row) + 32u * 63u + K <= n_rows;
    col = phi;
    win = 0;
    wp = mine + col * kTrajColStride;
  }
  __device__ __forceinline__ void flush() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    char *line_t = line0 + win * 128u;
    const uint64_t it_step = 1024ull * row_len;  // 8 streams further
    const uint32_t o_first = 32u * win - phi;    // stream offset of column 0 (window 0: wraps below zero)
    if (whole && win != 0u && o_first + 32u <= stream_len) {  // uniform: a whole line of every stream
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const float *src = take + 8 * it;
        const float4 v = make_float4(src[0], src[kTrajColStride], src[2 * kTrajColStride], src[3 * kTrajColStride]);
        *reinterpret_cast<float4 *>(line_t + it * it_step + lane_off) = v;
      }
    } else {
      // a stream's first line (columns from phi on: the columns before belong to the stream of the rows before),
      // its last line (up to its last value), streams that reach past the launch's last row
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const uint32_t r = sub + 8u * it;
        const uint64_t first_row = static_cast<uint64_t>(stream0) + 32u * r;
        const uint32_t rows_here = first_row >= n_rows ? 0u : (n_rows - first_row < K ? static_cast<uint32_t>(n_rows - first_row) : K);
        const uint32_t valid = rows_here * row_len;
        const float *src = take + 8 * it;
#pragma unroll
        for (uint32_t e = 0; e < 4; ++e) {
          const uint32_t o = o_first + 4u * quad + e;  // unsigned: columns before the stream wrap to huge values
          if (o < valid) *reinterpret_cast<float *>(line_t + it * it_step + lane_off + 4u * e) = src[e * kTrajColStride];
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    win += 1u;
  }
  // one value of the stream; at most kTrajTrip of them between two calls of line_check()
  __device__ __forceinline__ void put(float v) {
    *wp = v;
    wp += kTrajColStride;
    col += 1u;
  }
  // after a trip: a complete line goes out, the columns beyond it (at most kTrajTrip - 1) open the next one
  __device__ __forceinline__ void line_check() {
    if (col >= 32u) {  // uniform
      flush();
#pragma unroll
      for (uint32_t c = 0; c + 1u < kTrajTrip; ++c) mine[c * kTrajColStride] = mine[(32u + c) * kTrajColStride];
      col -= 32u;
      wp = mine + col * kTrajColStride;
    }
  }
  __device__ __forceinline__ void end() {
    line_check();
    if (col != 0u) flush();
  }
};

template <int kDiv, bool kTraj>
__global__ __launch_bounds__(kBlock) void ref_windowed_kernel(const RefArgs k) {
  extern __shared__ __align__(16) float lds_table[];
  for (uint32_t i = threadIdx.x; i < k.table_len; i += kBlock) lds_table[i] = k.table_a[i];
  __syncthreads();
  constexpr bool kExactDiv = kDiv == kDivExact;
  const uint32_t kmul = mt_multiplier();
  const MtConsts mconst = mt_consts();
  const uint32_t P = k.n_periods;  // <= ref_windowed_max_outputs()
  TrajWriter tw;
  if constexpr (kTraj) tw.init(lds_table + ((k.table_len + 3u) & ~3u) + (threadIdx.x >> 6) * kTrajTileWords, k.n_paths, P, k.traj_rows);

  // one path: generator of seed0 + i, P periods, the final value (or the redo list)
  auto run_path = [&](const uint32_t i) {
    MtWindow w;
    w.k = kmul;
    w.mc = mconst;
    w.seed = k.seed0 + i;
    w.x397 = w.seed;
    w.x397 = mt_run_up(w.seed, w.k);
    window_enter_a(w);
    w.ch = w.c1 = w.c1h = 0u;
    float total = k.initial_capital;
    // lanes whose path rejected an output or left the checked divide's window: a scalar mask (v_cmp
    // writes the wave's 64 results to an SGPR pair, the OR is scalar: one VALU instruction per period)
    uint64_t redo_mask = 0;
    auto period = [&](uint32_t g, uint32_t j) {
      const uint32_t y = mt_temper(g, mconst);
      const uint64_t prod = static_cast<uint64_t>(y) * k.table_len;
      redo_mask |= __ballot(static_cast<uint32_t>(prod) < k.reject_below);
      total = compound<kExactDiv>(total, lds_table[static_cast<uint32_t>(prod >> 32)]);
      if constexpr (kDiv == kDivChecked) {  // the window of divide_kind() (smmc_capi.cpp): at least every 8 periods
        if ((j & 7u) == 7u) redo_mask |= __ballot(!(total > k.chk_lo && total < k.chk_hi));
      }
      if constexpr (kTraj) tw.put(total);  // value j + 1 of the row
    };
    auto trip_done = [&]() {
      if constexpr (kTraj) tw.line_check();
    };
    static_assert(kTrajTrip == 4, "the period loops below make trips of four outputs");
    if constexpr (kTraj) {
      tw.put(total);  // values[0]: the initial capital
      tw.line_check();
    }
    // four outputs per trip, written out (a loop holding a ballot is not unrolled with a remainder)
    uint32_t j = 0;
    const uint32_t first = P < kMtLag ? P : kMtLag;
    for (; j + 4u <= first; j += 4u) {
      uint32_t far = j + kMtM + 1u;
      asm("" : "+s"(far));  // a scalar of its own (see window_next_a)
#pragma unroll
      for (uint32_t t = 0; t < 4u; ++t) period(window_next_a(w, j + t, far + t), j + t);
      trip_done();
    }
    for (; j < first; ++j) {
      period(window_next_a(w, j, j + kMtM + 1u), j);
      trip_done();
    }
    if (P > kMtLag) {
      window_enter_b(w);
      for (; j + 4u <= P; j += 4u) {
        uint32_t near = j - kMtLag + 2u, far = j + kMtM - kMtLag + 1u;
        asm("" : "+s"(near));
        asm("" : "+s"(far));
#pragma unroll
        for (uint32_t t = 0; t < 4u; ++t) period(window_next_b(w, j + t, near + t, far + t), j + t);
        trip_done();
      }
      for (; j < P; ++j) {
        period(window_next_b(w, j, j - kMtLag + 2u, j + kMtM - kMtLag + 1u), j);
        trip_done();
      }
    }
    if (i < k.n_paths) {
      if ((redo_mask >> (threadIdx.x & 63u)) & 1u) {
        k.redo_list[atomicAdd(k.redo_count, 1u)] = i;  // finished by ref_generic_kernel with the IEEE divide
      } else {
        k.d_final[i] = total;
      }
    }
  };

  if constexpr (kTraj) {
    const uint32_t K = k.traj_rows, per_super = 8u / K;  // workgroups per super-chunk
    const uint32_t n_units = ((k.n_paths + kTrajSuper - 1u) / kTrajSuper) * per_super;  // n_paths <= 2^31
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    for (uint32_t unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
      const uint32_t first_row = (unit / per_super) * kTrajSuper + K * (kWaves * (unit % per_super) + wave);  // stream 0; lane l: + 32 l
      tw.begin(k.d_traj, first_row);
      for (uint32_t kk = 0; kk < K; ++kk) run_path(first_row + 32u * lane + kk);
      tw.end();
    }
  } else {
    const uint32_t n_chunks = (k.n_paths + kBlock - 1u) / kBlock;
    for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) run_path(chunk * kBlock + threadIdx.x);
  }
}

// ---- the windowed recurrence carried on: paths of 455 .. 1816 outputs (ref_tree_kernel) -----------------------
//
// Output n is temper(new[n]),  new[k] = word(k + 397) ^ tw(word(k), word(k + 1)),  word(j) = x[j] for j < 624 (a
// seed word: a chain) and new[j - 624] from there on.  The windowed kernel above is the first two stretches of
// that recursion written out by hand: new[n] from three seed chains, then (n >= 227) with word(n + 397) itself
// a new[], generated again from its own seed chains.  Written as the recursion it is, the same idea goes on:
// at output 454 word(n + 397)'s own far operand becomes a new[], at 623 word(n + 1) does, and so on -- every
// operand that stops being a seed word is generated again, from seed chains of its own, and nothing is ever
// stored.  What a path carries is one word per STREAM:
//
//   stream W  =  word(n + W) at output n:  the seed chain x[n + W] while n + W < 624 ("seed form"), afterwards
//                new[n + W - 624], made from the streams W - 623 (its word(k + 1)) and W - 227 (its word(k + 397))
//                and the half of its own previous word(k); the output itself is stream 624.
//
// Streams are shared: word(n - 453) is the far operand of one node and the near operand of another, and one
// chain serves both (a tree that gave every operand its own chain would carry 9 chains and 8 twists per output
// from output 908, 42 and 41 from output 1869; shared, 6 and 7, 10 and 19).  The set of streams, the output at
// which each begins (its first user's first output: every chain begins at x[1] or x[397]) and the output 624 - W
// at which it changes form are compile-time tables (tree_tables()); between two such outputs the loop body is one
// fixed expression -- tree_run_stretch<S> is instantiated per stretch, `if constexpr` picks every stream's form,
// evaluated in ascending W (a stream's operands have smaller W) -- and at a stretch's first output the streams that
// begin or change form there are set up.  A made word travels as a pair p ^ q where that saves the XOR (a user
// that adds its own twist term folds all three in one v_bitop3).  Per output: 3 VALU per live chain, 4-5 per
// made word, 14 for tempering, the Lemire product, the rejection flag and the step.
// Two instantiations: paths of up to 1077 periods carry 13 words of stream state (45 VGPRs, eight waves per SIMD),
// paths of up to 1816 periods 26 (70 VGPRs, seven waves); longer ones go to ref_generic_kernel.
constexpr int kTreeMaxShort = 1077, kTreeMaxLong = 1816;
constexpr int kTreeCap = 64;     // room in the tables

struct TreeTables {
  int n = 0;                 // streams, ascending W; the last one is W = 624, the output
  int w[kTreeCap] = {};      // W
  int first[kTreeCap] = {};  // the output at which the stream begins
  int n_stretch = 0;
  int stretch[kTreeCap] = {};  // first outputs of the stretches: 0 and every 624 - W inside (0, kTreeMax)
};

constexpr TreeTables tree_tables(int kTreeMax) {
  TreeTables t;
  // worklist of streams: a stream W made (from output 624 - W on) needs the streams W - 623 and W - 227 from then on
  int work_w[4 * kTreeCap] = {}, work_first[4 * kTreeCap] = {}, n_work = 0;
  work_w[n_work] = int(kMtN);
  work_first[n_work++] = 0;
  for (int i = 0; i < n_work; ++i) {
    const int w = work_w[i], first = work_first[i];
    int at = -1;
    for (int j = 0; j < t.n; ++j)
      if (t.w[j] == w) at = j;
    if (at >= 0) {  // known: its operands are on the list already (with the same outputs: 624 - W does not depend on the user)
      if (first < t.first[at]) t.first[at] = first;
      continue;
    }
    t.w[t.n] = w;
    t.first[t.n++] = first;
    const int made_from = int(kMtN) - w;  // the output at which this stream changes form
    if (made_from < kTreeMax) {
      const int from = made_from < 0 ? 0 : made_from;
      work_w[n_work] = w - int(kMtN) + 1;
      work_first[n_work++] = from;
      work_w[n_work] = w - int(kMtLag);
      work_first[n_work++] = from;
    }
  }
  for (int i = 1; i < t.n; ++i)  // ascending W
    for (int j = i; j > 0 && t.w[j - 1] > t.w[j]; --j) {
      const int tw_ = t.w[j], tf = t.first[j];
      t.w[j] = t.w[j - 1];
      t.first[j] = t.first[j - 1];
      t.w[j - 1] = tw_;
      t.first[j - 1] = tf;
    }
  t.stretch[t.n_stretch++] = 0;
  for (int pass = 0; pass < kTreeCap; ++pass) {  // the next larger 624 - W, until none is left
    const int last = t.stretch[t.n_stretch - 1];
    int best = kTreeMax;
    for (int j = 0; j < t.n; ++j) {
      const int sw = int(kMtN) - t.w[j];
      if (sw > last && sw < best) best = sw;
    }
    if (best == kTreeMax) break;
    t.stretch[t.n_stretch++] = best;
  }
  return t;
}
template <int kMax>
struct Tree {
  static constexpr TreeTables t = tree_tables(kMax);
  static constexpr int index_of(int w) {
    for (int j = 0; j < t.n; ++j)
      if (t.w[j] == w) return j;
    return -1;
  }
  static constexpr bool live(int out, int i) { return i >= 0 && t.first[i] <= out; }
  static constexpr bool made(int out, int i) { return out + t.w[i] >= int(kMtN); }
  // a made word is left as a pair exactly when its far operand is a single word
  static constexpr bool is_pair(int out, int i) {
    if (!made(out, i)) return false;
    return !is_pair(out, index_of(t.w[i] - int(kMtLag)));
  }
};
static_assert(Tree<kTreeMaxLong>::t.n < kTreeCap && Tree<kTreeMaxLong>::t.n_stretch < kTreeCap, "tree tables");
static_assert(Tree<kTreeMaxShort>::t.w[Tree<kTreeMaxShort>::t.n - 1] == int(kMtN) && Tree<kTreeMaxShort>::t.n_stretch == 7 &&
                  Tree<kTreeMaxLong>::t.n_stretch == 17 && Tree<kTreeMaxLong>::t.stretch[1] == 227 &&
                  Tree<kTreeMaxLong>::t.stretch[2] == 454 && Tree<kTreeMaxLong>::t.stretch[3] == 623 &&
                  Tree<kTreeMaxLong>::t.stretch[4] == 681 && Tree<kTreeMaxLong>::t.stretch[5] == 850 &&
                  Tree<kTreeMaxLong>::t.stretch[6] == 908 && Tree<kTreeMaxLong>::t.stretch[7] == 1077,
              "stretches of the reference-stream tree");

struct TreeSeeds {
  uint32_t seed, x1, x397, k;  // x[0], x[1], x[397], mt_multiplier()
  MtConsts mc;                 // mt_consts()
};
struct TreeState {
  uint32_t s[kTreeCap];  // per stream: the chain's word x[n + W] (seed form) or word(n + W - 624) >> 1 (made)
};
struct TreeWords {
  uint32_t p[kTreeCap], q[kTreeCap];  // this output's word of every live stream: p, or p ^ q
};

template <int kMax, int kT, int kI>
__device__ __forceinline__ uint32_t tree_single(const TreeWords &v) {
  if constexpr (Tree<kMax>::is_pair(kT, kI)) return v.p[kI] ^ v.q[kI];
  else return v.p[kI];
}

// streams kI, kI + 1, ... of output n in stretch kT
template <int kMax, int kT, int kI>
__device__ __forceinline__ void tree_eval(TreeState &st, TreeWords &v, uint32_t n, const TreeSeeds &sd) {
  using Tr = Tree<kMax>;
  if constexpr (kI < Tr::t.n) {
    if constexpr (Tr::live(kT, kI)) {
      constexpr int kW = Tr::t.w[kI];
      if constexpr (!Tr::made(kT, kI)) {
        v.p[kI] = st.s[kI];
        st.s[kI] = mt_seed_step(st.s[kI], n + static_cast<uint32_t>(kW + 1), sd.k);
      } else {
        constexpr int kNear = Tr::index_of(kW - int(kMtN) + 1), kFar = Tr::index_of(kW - int(kMtLag));
        static_assert(Tr::live(kT, kNear) && Tr::live(kT, kFar) && kNear < kI && kFar < kI, "operands come first");
        const uint32_t a1 = tree_single<kMax, kT, kNear>(v);
        const uint32_t a1h = a1 >> 1;
        const uint32_t term = mt_twist_term(st.s[kI], a1h, a1, sd.mc);
        st.s[kI] = a1h;
        if constexpr (Tr::is_pair(kT, kFar)) {
          v.p[kI] = xor3(v.p[kFar], v.q[kFar], term);
        } else {
          v.p[kI] = v.p[kFar];
          v.q[kI] = term;
        }
      }
    }
    tree_eval<kMax, kT, kI + 1>(st, v, n, sd);
  }
}

// the streams that begin at output kT (a chain at x[1] or x[397]) or change form there (word(0) >> 1)
template <int kMax, int kT, int kI>
__device__ __forceinline__ void tree_begin(TreeState &st, const TreeSeeds &sd) {
  using Tr = Tree<kMax>;
  if constexpr (kI < Tr::t.n) {
    constexpr int kW = Tr::t.w[kI];
    if constexpr (int(kMtN) - kW == kT || (kW >= int(kMtN) && kT == 0)) {
      st.s[kI] = sd.seed >> 1;
    } else if constexpr (Tr::t.first[kI] == kT) {
      static_assert(kT + kW == 1 || kT + kW == int(kMtM), "a chain begins at x[1] or x[397]");
      st.s[kI] = kT + kW == 1 ? sd.x1 : sd.x397;
    }
    tree_begin<kMax, kT, kI + 1>(st, sd);
  }
}

// outputs [stretch kS, min(P, stretch kS + 1)), then the stretches after it
template <int kMax, int kS, typename Period, typename TripDone>
__device__ __forceinline__ void tree_run_stretch(TreeState &st, const TreeSeeds &sd, uint32_t P, Period &period, TripDone &trip_done) {
  using Tr = Tree<kMax>;
  constexpr int kT = Tr::t.stretch[kS];
  if (P <= static_cast<uint32_t>(kT)) return;
  constexpr int kNext = kS + 1 < Tr::t.n_stretch ? Tr::t.stretch[kS + 1] : kMax;
  tree_begin<kMax, kT, 0>(st, sd);
  const uint32_t end = P < static_cast<uint32_t>(kNext) ? P : static_cast<uint32_t>(kNext);
  auto output = [&](uint32_t n) {
    TreeWords v;
    tree_eval<kMax, kT, 0>(st, v, n, sd);
    return tree_single<kMax, kT, Tr::t.n - 1>(v);
  };
  uint32_t j = static_cast<uint32_t>(kT);
  for (; j + 4u <= end; j += 4u) {  // four outputs per trip, written out (a loop holding a ballot is not unrolled with a remainder)
#pragma unroll
    for (uint32_t t = 0; t < 4u; ++t) period(output(j + t), j + t);
    trip_done();
  }
  for (; j < end; ++j) {
    period(output(j), j);
    trip_done();
  }
  if constexpr (kS + 1 < Tr::t.n_stretch) tree_run_stretch<kMax, kS + 1>(st, sd, P, period, trip_done);
}

// As ref_windowed_kernel (rejections and checked-divide leavers flagged for the redo launch, kTraj through the
// wave's LDS tile), for paths of up to kMax outputs.
template <int kDiv, bool kTraj, int kMax>
__global__ __launch_bounds__(kBlock) void ref_tree_kernel(const RefArgs k) {
  extern __shared__ __align__(16) float lds_table[];
  for (uint32_t i = threadIdx.x; i < k.table_len; i += kBlock) lds_table[i] = k.table_a[i];
  __syncthreads();
  constexpr bool kExactDiv = kDiv == kDivExact;
  TreeSeeds sd;
  sd.k = mt_multiplier();
  sd.mc = mt_consts();
  const uint32_t P = k.n_periods;  // <= kMax
  TrajWriter tw;
  if constexpr (kTraj) tw.init(lds_table + ((k.table_len + 3u) & ~3u) + (threadIdx.x >> 6) * kTrajTileWords, k.n_paths, P, k.traj_rows);

  auto run_path = [&](const uint32_t i) {
    sd.seed = k.seed0 + i;
    sd.x1 = mt_seed_step(sd.seed, 1u, sd.k);
    sd.x397 = mt_run_up(sd.seed, sd.k);
    TreeState st;
    float total = k.initial_capital;
    uint64_t redo_mask = 0;
    auto period = [&](uint32_t g, uint32_t j) {
      const uint32_t y = mt_temper(g, sd.mc);
      const uint64_t prod = static_cast<uint64_t>(y) * k.table_len;
      redo_mask |= __ballot(static_cast<uint32_t>(prod) < k.reject_below);
      total = compound<kExactDiv>(total, lds_table[static_cast<uint32_t>(prod >> 32)]);
      if constexpr (kDiv == kDivChecked) {
        if ((j & 7u) == 7u) redo_mask |= __ballot(!(total > k.chk_lo && total < k.chk_hi));
      }
      if constexpr (kTraj) tw.put(total);
    };
    auto trip_done = [&]() {
      if constexpr (kTraj) tw.line_check();
    };
    if constexpr (kTraj) {
      tw.put(total);
      tw.line_check();
    }
    tree_run_stretch<kMax, 0>(st, sd, P, period, trip_done);
    if (i < k.n_paths) {
      if ((redo_mask >> (threadIdx.x & 63u)) & 1u) {
        k.redo_list[atomicAdd(k.redo_count, 1u)] = i;
      } else {
        k.d_final[i] = total;
      }
    }
  };

  if constexpr (kTraj) {  // the comb of ref_windowed_kernel
    const uint32_t K = k.traj_rows, per_super = 8u / K;
    const uint32_t n_units = ((k.n_paths + kTrajSuper - 1u) / kTrajSuper) * per_super;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    for (uint32_t unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
      const uint32_t first_row = (unit / per_super) * kTrajSuper + K * (kWaves * (unit % per_super) + wave);
      tw.begin(k.d_traj, first_row);
      for (uint32_t kk = 0; kk < K; ++kk) run_path(first_row + 32u * lane + kk);
      tw.end();
    }
  } else {
    const uint32_t n_chunks = (k.n_paths + kBlock - 1u) / kBlock;
    for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) run_path(chunk * kBlock + threadIdx.x);
  }
}

// Any number of periods: the circular state in global memory, word s of lane l at workspace[s * L + l]
// (L = lanes of the launch).  Work items are the paths 0 .. n_paths - 1, or -- redo_list given -- the
// *redo_count paths the windowed kernel left over.
// kTraj: a lane that rejected outputs is at a period of its own, so every lane stores its own value (one
// 4-byte store per lane and period into its own row: the slow store shape, kept to the fallback).
template <bool kExactDiv, bool kTraj>
__global__ __launch_bounds__(kBlock) void ref_generic_kernel(const RefArgs k) {
  extern __shared__ __align__(16) float lds_table[];
  for (uint32_t i = threadIdx.x; i < k.table_len; i += kBlock) lds_table[i] = k.table_a[i];
  __syncthreads();
  const uint32_t L = gridDim.x * kBlock;
  uint32_t *const W = k.workspace + static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x;
  const uint32_t count = k.redo_list ? *k.redo_count : k.n_paths;
  const uint32_t P = k.n_periods;
  const uint32_t kmul = mt_multiplier();
  const MtConsts mconst = mt_consts();
  for (uint64_t base = static_cast<uint64_t>(blockIdx.x) * kBlock; base < count; base += L) {
    const uint64_t item = base + threadIdx.x;
    const bool active = item < count;
    const uint32_t i = !active ? 0u : (k.redo_list ? k.redo_list[item] : static_cast<uint32_t>(item));
    MtWindow w;
    w.k = kmul;
    w.mc = mconst;
    w.seed = k.seed0 + i;
    w.x397 = w.seed;
    float total = k.initial_capital;
    uint32_t need = active ? P : 0u;
    float *row = nullptr;
    if constexpr (kTraj) {
      row = k.d_traj + static_cast<size_t>(i) * (P + 1u);
      if (active) row[0] = total;
    }
    auto use = [&](uint32_t g) {
      const bool took = offer<kExactDiv>(k, lds_table, w.mc, g, total, need);
      if constexpr (kTraj) {
        if (took) row[P - need] = total;
      }
    };
    // The seed words x[0 .. 623] are never stored: as in the windowed kernel they come from chains (two
    // while j < 227, one while j < 623); only generated words x[624 + j] go to the workspace, slot j mod 624.
    // Against seeding the whole state first this saves 624 stores and 1020 loads per path (41 % of the
    // traffic of a 1000-period path, which is what bounds this kernel).
    uint32_t j = 0;
    if (__any(need != 0u)) {
      w.x397 = mt_run_up(w.seed, w.k);
      window_enter_a(w);
      // outputs 0 .. 226: all three operands are seed words
      while (j < kMtLag && __any(need != 0u)) {
        const uint32_t n_here = kMtLag - j < 4u ? kMtLag - j : 4u;
        for (uint32_t t = 0; t < n_here; ++t) {
          const uint32_t g = window_next_a(w, j + t, j + t + kMtM + 1u);
          W[static_cast<size_t>(j + t) * L] = g;
          use(g);
        }
        j += n_here;
      }
      // outputs 227 .. 623: x[j + 397] is the generated word of output j - 227; x[j], x[j + 1] stay seed words
      // until x[624] (output 0) becomes x[j + 1] at j = 623
      while (j < kMtN && __any(need != 0u)) {
        const uint32_t n_here = kMtN - j < 4u ? kMtN - j : 4u;
        uint32_t m[4];
        for (uint32_t t = 0; t < n_here; ++t) m[t] = W[static_cast<size_t>(j + t - kMtLag) * L];
        for (uint32_t t = 0; t < n_here; ++t) {
          const uint32_t g = m[t] ^ mt_twist_term(w.ah, w.a1h, w.a1, w.mc);
          w.ah = w.a1h;
          // x[j + t + 2]: a seed word, or -- the last two outputs of this stretch -- generated word 0 / 1
          w.a1 = j + t + 2u < kMtN ? mt_seed_step(w.a1, j + t + 2u, w.k) : W[static_cast<size_t>(j + t + 2u - kMtN) * L];
          w.a1h = w.a1 >> 1;
          W[static_cast<size_t>(j + t) * L] = g;
          use(g);
        }
        j += n_here;
      }
    }
    // outputs 624 ...: every operand is a generated word.  Batches of 8 (624 = 8 x 78: none straddles the wrap),
    // their 16 operands loaded before the batch (none of them is written inside it)
    uint32_t xnh = w.ah;  // x[j] >> 1: at j = 624 the stretch above leaves x[624] >> 1 here
    uint32_t s = 0;       // j mod 624
    auto wrap = [](uint32_t v) { return v >= kMtN ? v - kMtN : v; };
    while (__any(need != 0u)) {
      uint32_t n1[8], m[8];
#pragma unroll
      for (uint32_t t = 0; t < 8; ++t) {  // x[j + 1 + t], x[j + 397 + t]
        n1[t] = W[static_cast<size_t>(wrap(s + 1u + t)) * L];
        m[t] = W[static_cast<size_t>(wrap(s + kMtM + t)) * L];
      }
#pragma unroll
      for (uint32_t t = 0; t < 8; ++t) {
        const uint32_t n1h = n1[t] >> 1;
        const uint32_t g = m[t] ^ mt_twist_term(xnh, n1h, n1[t], w.mc);
        W[static_cast<size_t>(s + t) * L] = g;  // x[j + t + 624] takes the place of x[j + t]
        use(g);
        xnh = n1h;
      }
      s = wrap(s + 8u);
    }
    if (active) k.d_final[i] = total;
  }
}

// Mean and population variance of every 256 consecutive values: the per-chunk outputs of paths_kernel
// (smmc_kernels.hip, same arithmetic in the same order) for launches whose final values come from
// another kernel.  One workgroup per chunk.
__global__ __launch_bounds__(kBlock) void chunk_stats_kernel(const float *values, uint64_t n, float *d_mean, float *d_var) {
  __shared__ double slot[2 * kWaves];
  const uint64_t n_chunks = (n + kBlock - 1) / kBlock;
  for (uint64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    const uint64_t i = chunk * kBlock + threadIdx.x;
    const uint64_t left = n - chunk * kBlock;
    const double n_in = static_cast<double>(left < kBlock ? left : kBlock);
    const double dv = i < n ? static_cast<double>(values[i]) : 0.0;
    const double s1 = wave_sum(dv), s2 = wave_sum(dv * dv);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
      slot[wave] = s1;
      slot[kWaves + wave] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double t1 = slot[0], t2 = slot[kWaves];
#pragma unroll
      for (int w = 1; w < kWaves; ++w) {
        t1 += slot[w];
        t2 += slot[kWaves + w];
      }
      const double mean = t1 / n_in;
      const double var = t2 / n_in - mean * mean;
      if (d_mean) d_mean[chunk] = static_cast<float>(mean);
      if (d_var) d_var[chunk] = static_cast<float>(var > 0.0 ? var : 0.0);
    }
    __syncthreads();
  }
}

}  // namespace

// outputs the state-free kernels can generate: ref_windowed_kernel up to 454, ref_tree_kernel from there to kTreeMaxLong
constexpr uint32_t kWindowedMax = kMtN - 170u;
uint32_t ref_windowed_max_outputs() { return static_cast<uint32_t>(kTreeMaxLong); }
size_t ref_workspace_bytes(uint32_t grid) { return static_cast<size_t>(grid) * kBlock * kMtN * sizeof(uint32_t); }

size_t ref_windowed_lds_bytes(uint32_t table_len, bool traj) {
  return (static_cast<size_t>((table_len + 3u) & ~3u) + (traj ? kWaves * kTrajTileWords : 0u)) * sizeof(float);
}

namespace {
template <typename Kernel>
hipError_t launch_ref(Kernel kernel, const RefArgs &a, uint32_t grid, size_t lds, hipStream_t stream) {
  if (lds > 60u * 1024u) {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         static_cast<int>(lds));
    if (err != hipSuccess) return err;
  }
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), lds, stream, a);
  return hipGetLastError();
}
}  // namespace

namespace {
template <int kMax>
hipError_t launch_ref_tree(const RefArgs &a, int div, bool traj, uint32_t grid, size_t lds, hipStream_t stream) {
  switch (div) {
    case SMMC_DIV_FAST:
      return traj ? launch_ref(ref_tree_kernel<kDivFast, true, kMax>, a, grid, lds, stream)
                  : launch_ref(ref_tree_kernel<kDivFast, false, kMax>, a, grid, lds, stream);
    case SMMC_DIV_CHECKED:
      return traj ? launch_ref(ref_tree_kernel<kDivChecked, true, kMax>, a, grid, lds, stream)
                  : launch_ref(ref_tree_kernel<kDivChecked, false, kMax>, a, grid, lds, stream);
    default:
      return traj ? launch_ref(ref_tree_kernel<kDivExact, true, kMax>, a, grid, lds, stream)
                  : launch_ref(ref_tree_kernel<kDivExact, false, kMax>, a, grid, lds, stream);
  }
}
}  // namespace

// a.d_traj != nullptr: every value of every path as well (rows of n_periods + 1 floats)
hipError_t launch_ref_windowed(const RefArgs &a, int div, uint32_t grid, hipStream_t stream) {
  const bool traj = a.d_traj != nullptr;
  const size_t lds = ref_windowed_lds_bytes(a.table_len, traj);
  if (traj) {  // a workgroup takes 8 / K-th of a super-chunk of 2048 rows, not a chunk of 256
    if (a.traj_rows != 8u && a.traj_rows != 4u && a.traj_rows != 2u && a.traj_rows != 1u) return hipErrorInvalidValue;
    const uint32_t n_units = ((a.n_paths + kTrajSuper - 1u) / kTrajSuper) * (8u / a.traj_rows);
    grid = grid < n_units ? grid : n_units;
  }
  // SMMC_REF_KERNEL=tree (test / measurement knob): the tree form also for the lengths the hand-written one takes
  const char *env = std::getenv("SMMC_REF_KERNEL");
  const bool tree_always = env && !std::strcmp(env, "tree");
  if (a.n_periods > kWindowedMax || tree_always)
    return a.n_periods <= static_cast<uint32_t>(kTreeMaxShort) ? launch_ref_tree<kTreeMaxShort>(a, div, traj, grid, lds, stream)
                                                                : launch_ref_tree<kTreeMaxLong>(a, div, traj, grid, lds, stream);
  switch (div) {
    case SMMC_DIV_FAST:
      return traj ? launch_ref(ref_windowed_kernel<kDivFast, true>, a, grid, lds, stream)
                  : launch_ref(ref_windowed_kernel<kDivFast, false>, a, grid, lds, stream);
    case SMMC_DIV_CHECKED:
      return traj ? launch_ref(ref_windowed_kernel<kDivChecked, true>, a, grid, lds, stream)
                  : launch_ref(ref_windowed_kernel<kDivChecked, false>, a, grid, lds, stream);
    default:
      return traj ? launch_ref(ref_windowed_kernel<kDivExact, true>, a, grid, lds, stream)
                  : launch_ref(ref_windowed_kernel<kDivExact, false>, a, grid, lds, stream);
  }
}

hipError_t launch_ref_generic(const RefArgs &a, bool exact_div, uint32_t grid, hipStream_t stream) {
  const size_t lds = static_cast<size_t>(a.table_len) * sizeof(float);
  if (a.d_traj)
    return exact_div ? launch_ref(ref_generic_kernel<true, true>, a, grid, lds, stream)
                     : launch_ref(ref_generic_kernel<false, true>, a, grid, lds, stream);
  return exact_div ? launch_ref(ref_generic_kernel<true, false>, a, grid, lds, stream)
                   : launch_ref(ref_generic_kernel<false, false>, a, grid, lds, stream);
}

hipError_t launch_chunk_stats(const float *values, uint64_t n, float *d_mean, float *d_var, uint32_t grid, hipStream_t stream) {
  hipLaunchKernelGGL(chunk_stats_kernel, dim3(grid), dim3(kBlock), 0, stream, values, n, d_mean, d_var);
  return hipGetLastError();
}

}  // namespace smmc
