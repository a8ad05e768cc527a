// Tuning / test switches of the library.  Every `JD_*` switch is read from the environment ONCE, when the library
// is loaded; after that it only changes through jd_set_option() (the explicit hook of the tests and A/B tools).
// No per-step function calls getenv: the launch paths read an atomic int.
#include <atomic>
#include <climits>
#include <cstdlib>
#include <cstring>

#include "jd_common.h"
#include "kernels.h"

namespace jd {

namespace {
struct OptionSlot {
  const char* name;
  std::atomic<int> value;  // INT_MIN = unset
};
// (order = enum JdOption in kernels.h)
OptionSlot g_options[OPT_COUNT] = {
    {"JD_SEP_NO_ALIAS", {INT_MIN}},      {"JD_SEP_FWD_MIN_LDS", {INT_MIN}},  {"JD_SEP_ADJ_MIN_LDS", {INT_MIN}},
    {"JD_SEP_INTERLEAVE", {INT_MIN}},    {"JD_SEP_NO_FUSION", {INT_MIN}},    {"JD_SEP_WALK", {INT_MIN}},
    {"JD_SEP_WALK_COLS", {INT_MIN}},     {"JD_SEP_WALK_ROWS", {INT_MIN}},    {"JD_SEP_WALK_ADJ_COLS", {INT_MIN}},
    {"JD_SEP_WALK_ADJ_ROWS", {INT_MIN}}, {"JD_DIRECT_FP32", {INT_MIN}},      {"JD_CONV_BLOCKS_PER_CU", {INT_MIN}},
    {"JD_POISSON_ROWS", {INT_MIN}},      {"JD_GMM_NO_HOST_STATS", {INT_MIN}}, {"JD_GMM_BLOCK_TILES", {INT_MIN}},
    {"JD_GMM_DENSE", {INT_MIN}},         {"JD_GMM_KSPLIT", {INT_MIN}},       {"JD_GMM_SCREEN_NP", {INT_MIN}},
    {"JD_GMM_SCREEN_NO_LDS_CONSTS", {INT_MIN}}, {"JD_GMM_SCREEN_DEBUG", {INT_MIN}}, {"JD_GMM_SCREEN", {INT_MIN}},
    {"JD_GMM_FUSED_BWD", {INT_MIN}},     {"JD_GMM_GATHER_TILED", {INT_MIN}}, {"JD_GMM_LSE_SCREEN", {INT_MIN}},
    {"JD_GMM_WINNER_ROWS", {INT_MIN}},   {"JD_SEP_JOINT", {INT_MIN}},        {"JD_SEP_JOINT_ROWS", {INT_MIN}},
    {"JD_SEP_JOINT_CHUNK", {INT_MIN}},   {"JD_SEP_WALK_ADJ_ALL", {INT_MIN}}, {"JD_SEP_WALK_COST33", {INT_MIN}},
    {"JD_SEP_WALK_ROWS33", {INT_MIN}},   {"JD_SEP_NO_TRIM", {INT_MIN}},      {"JD_SEP_WALK_ADJ_ROWS33", {INT_MIN}},
    {"JD_SEP_WALK_ADJ33", {INT_MIN}},    {"JD_FFT_NATIVE", {INT_MIN}},
    {"JD_DIRECT_AUTO_ALL", {INT_MIN}},  // "auto" takes the MFMA Toeplitz kernel up to 33 taps (as before round 4)
    {"JD_FFT_BATCH", {INT_MIN}},  // 0: the batched joint steps of a native FFT plan run their per-dataset calls; the calibrated
                                  // one beyond 2048 flux rows: 3 per-dataset calls, 4 per-dataset FFT launches + one tail
    {"JD_FFT_TINY", {INT_MIN}},   // longest row (points) of the one-wave generic row kernels; 0: off (default 1024)
    {"JD_FFT_POOL_IO", {INT_MIN}},  // 0: the column passes around the pooled launch of an up-sampled step move full rows
    {"JD_GMM_SORT_BLOCKS", {INT_MIN}},  // tuning: most blocks of the record sort's count / scatter launches (default: one
                                        // record segment per block up to max(2 CUs, 2^20 / K))
    {"JD_GMM_GATHER_PRELOAD", {INT_MIN}},  // 0: the gather kernel loads the optimizer step's streams behind its block barrier
};

int parse(const char* text) {
  // presence-only switches are set with any text: an empty or non-numeric value reads as 1
  char* end = nullptr;
  const long v = strtol(text, &end, 10);
  if (end == text) return 1;
  return v <= INT_MIN ? INT_MIN + 1 : v > INT_MAX ? INT_MAX : (int)v;
}

struct EnvInit {
  EnvInit() {
    for (auto& o : g_options)
      if (const char* env = getenv(o.name)) o.value.store(parse(env));
  }
} g_env_init;  // runs when the shared library is loaded
}  // namespace

bool opt_is_set(int id) { return g_options[id].value.load(std::memory_order_relaxed) != INT_MIN; }
int opt_value(int id, int unset_value) {
  const int v = g_options[id].value.load(std::memory_order_relaxed);
  return v == INT_MIN ? unset_value : v;
}

}  // namespace jd

using namespace jd;

extern "C" int jd_set_option(const char* key, const char* value) {
  JD_REQUIRE(key, "jd_set_option: key is null");
  for (auto& o : g_options)
    if (strcmp(o.name, key) == 0) {
      o.value.store(value ? parse(value) : INT_MIN);
      return JD_OK;
    }
  return fail(JD_ERR_INVALID, "jd_set_option: unknown option '%s'", key);
}

extern "C" int jd_get_option(const char* key, int* is_set, int* value) {
  JD_REQUIRE(key && is_set && value, "jd_get_option: null argument");
  for (auto& o : g_options)
    if (strcmp(o.name, key) == 0) {
      const int v = o.value.load();
      *is_set = v != INT_MIN, *value = v == INT_MIN ? 0 : v;
      return JD_OK;
    }
  return fail(JD_ERR_INVALID, "jd_get_option: unknown option '%s'", key);
}
