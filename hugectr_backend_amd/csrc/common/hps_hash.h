// Hash + synthetic-table arithmetic shared by host C++, HIP device code and the
// C oracle (oracle/hps_oracle.c re-states the same formulas independently).
//
// Nothing here comes from the reference: /root/reference ships no hashing code
// (SURVEY.md §0 — all arithmetic lives in the un-vendored libhuge_ctr_hps.so).
// The synthetic-table recipe is SURVEY.md §8(d).
#pragma once
#include <stdint.h>

#if defined(__HIP__)
#define HPS_HD __attribute__((host)) __attribute__((device)) inline __attribute__((always_inline))
#else
#define HPS_HD static inline
#endif

// splitmix64 finalizer: the one counter-based generator used everywhere.
HPS_HD uint64_t hps_mix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// Key reserved for "empty slot" in every hash structure (GPU cache buckets and
// host PS partitions).  A query for this key is legal: it is never inserted,
// always resolved through the slow path, so results stay exact.
#define HPS_EMPTY_KEY ((int64_t)0x8000000000000000ull)

// GPU cache geometry: one bucket = one 128-B line = 14 keys + 14 one-byte recency
// stamps, probed by an 8-lane group with one 16-byte load per lane.
#define HPS_BUCKET_SLOTS 14

// bucket index for a key in a table of `num_buckets` buckets (mul-shift range
// reduction on the high half of the mixed key; no 64-bit modulo on the GPU).
HPS_HD uint32_t hps_bucket_of(int64_t key, uint32_t num_buckets) {
  const uint64_t h = hps_mix64((uint64_t)key);
  return (uint32_t)(((h >> 32) * (uint64_t)num_buckets) >> 32);
}

// ---- synthetic tables --------------------------------------
// row(t,k)[j] : finite fp32 in [0.5,1) whose 23 mantissa bits are hash bits, so
// any misplaced row / lane / element shows up as a bit mismatch.
// One mix64 call yields two consecutive elements (j even: low bits, j odd: bits 32..54).
HPS_HD uint64_t hps_synth_table_base(uint64_t seed, uint32_t table) {
  return hps_mix64(seed ^ hps_mix64((uint64_t)table + 1));
}
HPS_HD uint64_t hps_synth_row_base(uint64_t table_base, int64_t key) {
  return hps_mix64(table_base + (uint64_t)key);
}
HPS_HD uint32_t hps_synth_elem_bits(uint64_t row_base, uint32_t j) {
  const uint64_t w = hps_mix64(row_base + (uint64_t)(j >> 1));
  const uint32_t m = (j & 1) ? (uint32_t)(w >> 32) : (uint32_t)w;
  return 0x3F000000u | (m & 0x007FFFFFu);
}
