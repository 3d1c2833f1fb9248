/* A plain C99 caller of the C ABI, written the way INTEGRATION.md shows a binding: one engine with
 * device-resident statistics, then a group of shards streaming final values into the caller's array with
 * ONE merged record.  Prints what it got as one JSON object; tests/test_c_caller_gpu.py compares it with
 * the oracle.  argv: <mode 0|1> <n_paths> <n_periods> <seed> <first_path> <n_shards> <flags> <table file>
 * (table file: one float per line). */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "smmc.h"

static void die(const char *what) {
  fprintf(stderr, "%s: %s\n", what, smmc_last_error());
  exit(1);
}

static uint64_t fnv1a(const void *data, size_t n) {
  const unsigned char *p = (const unsigned char *)data;
  uint64_t h = 0xCBF29CE484222325ull;
  for (size_t i = 0; i < n; ++i) h = (h ^ p[i]) * 0x100000001B3ull;
  return h;
}

int main(int argc, char **argv) {
  if (argc != 9) return 2;
  smmc_sim sim;
  memset(&sim, 0, sizeof sim);
  sim.struct_size = sizeof sim;
  sim.mode = atoi(argv[1]);
  sim.n_paths = strtoull(argv[2], NULL, 10);
  sim.n_periods = (uint32_t)strtoul(argv[3], NULL, 10);
  sim.seed = strtoull(argv[4], NULL, 10);
  sim.first_path = strtoull(argv[5], NULL, 10);
  const int n_shards = atoi(argv[6]);
  sim.flags = (uint32_t)strtoul(argv[7], NULL, 10);
  sim.initial_capital = 1000.0f;
  sim.gauss_mean = 0.5f;
  sim.gauss_std = 0.83333f;
  sim.n_bins = 100;
  sim.hist_lo = 0.0f;
  sim.hist_hi = 20000.0f;
  sim.below_threshold = 1000.0f;

  static float table[1 << 16];
  uint32_t n_table = 0;
  FILE *f = fopen(argv[8], "r");
  if (!f) return 3;
  while (n_table < (1u << 16) && fscanf(f, "%f", &table[n_table]) == 1) ++n_table;
  fclose(f);

  if (smmc_abi_version() != SMMC_ABI_VERSION) return 4;
  int n_dev = 0;
  if (smmc_device_count(&n_dev) || n_dev < 1) die("no device");

  /* (1) one engine: statistics only, read back with smmc_engine_sync + a host copy the ABI makes */
  smmc_engine *e = NULL;
  if (smmc_engine_create(0, SMMC_STREAM_NEW, &e)) die("engine_create");
  if (smmc_engine_set_table(e, table, n_table)) die("set_table");
  smmc_stats one;
  uint64_t one_hist[100];
  if (smmc_engine_simulate_to_host(e, &sim, NULL, NULL, NULL, NULL, &one, one_hist)) die("simulate_to_host");
  smmc_engine_destroy(e);

  /* (2) a group of n_shards shards, all on device 0 here (a node would list 0 .. 7) */
  int devices[64];
  for (int i = 0; i < n_shards; ++i) devices[i] = 0;
  smmc_group *g = NULL;
  if (smmc_group_create(devices, n_shards, SMMC_MERGE_HOST, &g)) die("group_create");
  if (smmc_group_set_table(g, table, n_table)) die("group_set_table");
  float *host_final = (float *)malloc(sizeof(float) * (sim.n_paths ? sim.n_paths : 1));
  smmc_stats st;
  uint64_t hist[100];
  volatile int64_t progress = -1;
  if (smmc_group_simulate(g, &sim, host_final, NULL, NULL, &progress, &st, hist)) die("group_simulate");
  uint64_t first = 0, count = 0;
  if (smmc_group_shard(g, sim.n_paths, n_shards - 1, &first, &count)) die("group_shard");
  const int size = smmc_group_size(g);
  smmc_group_destroy(g);

  /* error behaviour: a status and a message, never an exit */
  sim.n_bins = SMMC_MAX_BINS + 1;
  const int rc_bad = smmc_engine_simulate_to_host(NULL, &sim, NULL, NULL, NULL, NULL, NULL, NULL);

  uint32_t min_bits, max_bits;
  memcpy(&min_bits, &st.min, 4);
  memcpy(&max_bits, &st.max, 4);
  printf("{\"group_size\": %d, \"progress\": %" PRId64 ", \"last_shard\": [%" PRIu64 ", %" PRIu64 "], "
         "\"final_fnv1a\": %" PRIu64 ", \"count\": %" PRIu64 ", \"below\": %" PRIu64 ", \"underflow\": %" PRIu64
         ", \"overflow\": %" PRIu64 ", \"sum\": %.17g, \"sumsq\": %.17g, \"min_bits\": %u, \"max_bits\": %u, "
         "\"hist_fnv1a\": %" PRIu64 ", \"one_engine_hist_fnv1a\": %" PRIu64 ", \"one_engine_count\": %" PRIu64
         ", \"one_engine_below\": %" PRIu64 ", \"one_engine_sum\": %.17g, \"rc_bad\": %d}\n",
         size, (int64_t)progress, first, count, fnv1a(host_final, sizeof(float) * sim.n_paths), st.count, st.below,
         st.underflow, st.overflow, st.sum, st.sumsq, min_bits, max_bits,
         fnv1a(hist, sizeof hist), fnv1a(one_hist, sizeof one_hist), one.count, one.below, one.sum, rc_bad);
  free(host_final);
  return 0;
}
