// launch_stubs.cpp -- CPU sanitizer builds only (make -C oracle asan / tsan): stands in for the
// HIP translation units of THIS repository (csrc/smmc_kernels.hip, smmc_ref_kernels.hip, smmc_stats_kernels.hip, smmc_vector_add.hip), whose device
// code g++ cannot compile, so that the host-side product code (csrc/smmc_capi.cpp, smmc_dropin.cpp) links
// and its no-GPU paths can run under ASan / UBSan / TSan.  Every launch reports "no device".
#include "smmc_internal.h"

namespace smmc {
hipError_t launch_values_stats(const ValuesArgs &, uint32_t, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_radix_hist(const float *, uint64_t, int, uint32_t, const SelectState *, unsigned long long *, uint32_t,
                             hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_radix_pick(int, uint32_t, SelectState *, unsigned long long *, float *, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_paths(const KernelArgs &, int, uint32_t, size_t, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_finalize(const BlockPartial *, uint32_t, smmc_stats *, uint32_t, hipStream_t, unsigned long long *, uint32_t) { return hipErrorNoDevice; }
hipError_t launch_keepdata(const KernelArgs &, bool, int, int, uint32_t, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_keepdata_comb(const KernelArgs &, bool, int, uint32_t, uint64_t, uint64_t, int, uint32_t, unsigned long long *,
                                hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_final_column(const float *, uint64_t, uint32_t, float *, uint32_t, hipStream_t) { return hipErrorNoDevice; }
size_t keepdata_comb_lds_bytes(uint32_t, int, int) { return 0; }
uint32_t keepdata_draws(uint32_t table_len) { return (table_len && table_len <= 2048u) ? 8u : 4u; }
hipError_t launch_selftest(uint32_t, uint32_t, unsigned long long *, uint32_t, hipStream_t) { return hipErrorNoDevice; }
uint32_t values_hist_copies(uint32_t) { return 1; }
size_t paths_lds_bytes(uint32_t table_len, uint32_t n_bins, int) { return (static_cast<size_t>(table_len) + n_bins) * 4u; }
size_t keepdata_lds_bytes(uint32_t, int, int, int) { return 0; }
hipError_t static_lds_bytes(size_t *bytes) { *bytes = 0; return hipSuccess; }
uint32_t ref_windowed_max_outputs() { return 1816; }
size_t ref_workspace_bytes(uint32_t grid) { return static_cast<size_t>(grid) * 256 * 624 * 4; }
size_t ref_windowed_lds_bytes(uint32_t table_len, bool) { return static_cast<size_t>(table_len) * 4; }
hipError_t launch_ref_windowed(const RefArgs &, int, uint32_t, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_ref_generic(const RefArgs &, bool, uint32_t, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_chunk_stats(const float *, uint64_t, float *, float *, uint32_t, hipStream_t) { return hipErrorNoDevice; }
size_t bm_tables_bytes(int stream) { return stream == 2 ? (1056 * 4 + 256 * 2) * 4 : (512 * 4 + 2048 * 2) * 4; }
}  // namespace smmc

// csrc/smmc_vector_add.hip (a whole C-ABI entry lives in that HIP file)
extern "C" int smmc_vector_add(float *, const float *, const float *, int64_t, double *) { return SMMC_ERR_NO_DEVICE; }
