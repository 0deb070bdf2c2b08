// Plain-old-data shared between host code and the HIP kernels of the GPU embedding cache.
//
// Data layout in HBM:
//   per table t:  lines[num_buckets][16] 8-byte words  one bucket = ONE 128-B line (= one L2 line):
//                     words 0..13   the bucket's 14 keys (HPS_EMPTY_KEY = free slot)
//                     word  14      recency stamps of slots 0..7, one byte each
//                     word  15      recency stamps of slots 8..13 (bytes 0..5), bytes 6..7 unused
//                 rows [num_buckets*14][D] fp32        slot s owns rows[s*D .. s*D+D)
// A key lives in exactly one bucket: hps_bucket_of(key, num_buckets) (common/hps_hash.h).
//
// Recency lives INSIDE the line the probe has to read anyway (round 2 kept 32-bit stamps in a second array: every hit
// cost a 4-B store into a 277-MB array, and the micro-benchmark tools/micro/probe_width.hip says such a store costs more
// than the probe itself: 1.1 M probes 30 us, with the store 79 us, with a 1-B store into the probed line 63 us).
// A stamp is the cache's clock — its turnover: the unique rows its lookups missed, in units of total slots / 64
// (EmbeddingCache::NextEpoch; HPS_LRU_AGE_SHIFT: units of 2^shift calls instead) — modulo kStampMod = 254 ("stamp8"; the value
// 255 marks a slot that a group of a running insert kernel owns, 254 a slot that never held a key).  A hit rewrites its
// slot's stamp only when it differs from the current unit — the comparison is free, the byte came with the keys — so a
// key that is hit call after call costs one store per unit.  age = (now8 - stamp8) mod 254; the insert
// kernel evicts the slot of greatest age and, while it is at it, pulls stamps older than kAgeSaturate units back to
// exactly that age so that they cannot wrap around and look young.
#pragma once
#include <stdint.h>

#include "../common/hps_hash.h"

namespace hps {

constexpr int kBucketSlots = HPS_BUCKET_SLOTS;  // 14 keys per bucket line
constexpr int kLineWords = 16;                   // 8-byte words per bucket line (14 keys + 2 words of stamps)
constexpr int kProbeLanes = 8;                   // lanes that share one probe: 16 B each
constexpr uint32_t kStampMod = 254;              // stamps of slots that hold (or held) a key live in [0, 254)
constexpr uint32_t kStampClaimed = 255;          // stamp byte of a slot owned by a group of the running insert kernel
constexpr uint32_t kStampFree = 254;             // stamp of the never-used slots of a fresh cache: no clock value.  (Rounds 1-3 used
                                                 // 128, a legal clock value: while the clock — or the insert stamp — read 128, every
                                                 // never-used slot looked "written in the current unit" and could not be claimed; a
                                                 // cold cache (init_ec=false) dropped its misses for one whole unit each time.)
constexpr uint32_t kAgeSaturate = 192;           // stamps older than this many units are pulled back to it by the insert kernel
constexpr int kMaxTables = 256;                  // per model
constexpr int kProbeBlockThreads = 256;

// ---- probe tiles -------------------------------------------------------------------------------------------
// The probe kernel works on tiles of kTileKeys consecutive keys of ONE table (one workgroup per tile).  Everything a
// tile hands to the later kernels of the call lives in "tile regions": region of tile b = [b*kTileKeys, (b+1)*kTileKeys)
// of the per-call arrays below, filled from the front, with the fill counts in tile_cnt[b*4 ..].  No global atomic is
// needed to build them (device-scope atomics on distinct addresses run at ~15 G/s on gfx950, on one 128-B line at
// ~0.09 G/s: tools/micro/atomic_rate.hip).
constexpr int kTileKeys = 1024;
constexpr int kTileSet = 2048;   // LDS set entries of the tile-local input dedup (load <= 0.5)
enum : int { kTileCntRepMiss = 0, kTileCntSentMiss = 1 };

// Per-call accumulator block (uint32 words in HBM, zeroed by the call's descriptor upload, copied back whole):
//   line s in [0, kStatLines)      : [0] dropped [1] inserted [2] refreshed — insert statistics, block b of the insert
//                                     kernel adds to line b % kStatLines (one line would serialise ~2,000 atomics)
//   line kStatLines + t            : [kAccUniqMiss] unique missed keys of table t   [kAccUniqHit] unique hit keys of
//                                     table t (counted only when the insertion policy needs the hit rate)
//                                     [kAccSentMiss] missed keys of table t as sent (duplicates included: statistics)
// One line = kAccStride words = 128 B = one L2 line per table.
constexpr int kAccStride = 32;
constexpr int kStatLines = 32;
enum : int { kAccUniqMiss = 0, kAccUniqHit = 1, kAccSentMiss = 2 };
constexpr int kAccWordsMax = (kStatLines + kMaxTables) * kAccStride;
inline constexpr uint32_t AccTableWord(uint32_t t, int what) { return (uint32_t)(kStatLines + t) * kAccStride + (uint32_t)what; }

// slot[] encoding produced by the probe kernel
//   >= 0   cache slot of the key's row
//   <= -2  missed: -2 - m, m = position of the key's tile representative in the tile regions (miss_key[m]);
//          its row in the miss staging is uidx_of[rep_of[m]] once the miss-unique kernel has run
//   -1     padding (CallDesc::skip_empty_keys): nothing to read, nothing to write
constexpr int32_t kSlotMiss = -1;  // also the probe's transient "not found yet" of a representative

struct TableCacheDev {
  int64_t* lines;  // [num_buckets][kLineWords]
  float* rows;
  uint32_t num_buckets;
  uint32_t dim;
  float default_value;
  uint32_t flags;  // bit0: static cache (no stamp writes, no inserts)
};

// One probe tile: keys [begin, begin + count) of the call's flat key array, all of table `table`.
struct TileDesc {
  uint64_t begin;
  uint32_t count;
  uint32_t table;
};

// Per-call work arrays of a lookup session (device pointers; passed to the kernels by value).
struct CallWork {
  const TileDesc* tiles;     // [num_tiles]
  uint32_t num_tiles;
  uint32_t call_tag;         // tags this call's entries of `set` (never 0)
  int32_t* slot;             // [N]
  uint32_t* tile_cnt;        // [num_tiles*4]
  int64_t* miss_key;         // tile regions: key of the tile's r-th missed representative
  int32_t* sent_i;           // tile regions: global index of every missed key of the tile, as sent
  int32_t* sent_m;           // tile regions: the same keys' entry m of the tile's miss list (slot[sent_i[r]] == -2 - sent_m[r])
  int32_t* rep_of;           // tile regions: m -> m of the call-wide representative of the same (table, key)
  int32_t* uidx_of;          // tile regions: valid at representatives: index in the table's unique-miss segment
  unsigned long long* set;   // open addressing, entries (call_tag << 32 | m); other tags = free
  uint64_t set_mask;
  uint32_t* acc;             // accumulator block, layout above
  int64_t* uniq_keys;        // [N] unique missed keys of table t at [key_start[t], key_start[t] + count)
  int64_t* uniq_keys_host;   // the same array in host-mapped pinned memory (the host parameter server reads it)
  uint32_t* uniq_keys_host32; // not null: the call's keys all fit 32 bits (the request was narrowed) and the host wants
                              // the unique missed keys as uint32 — half the bytes of the zero-copy stores over PCIe
  uint32_t xcd_tiles;        // 1: the probe kernel gives XCD x the x-th eighth of the tiles (kernels.hip; needs >= 8 tiles).
                             // LAST ON PURPOSE: placed between call_tag and slot (every later field 8 bytes further into
                             // the kernel-argument block, the tail's wide scalar loads no longer aligned) the probe kernel
                             // took 50-51 us instead of 43, with the order on or off — same ISA but for the offsets
                             // (profiles/round3/ab_probe_xcd_tiles.txt).  Round 6 tried the opposite direction — 8 bytes of padding
                             // in front of `set`, so that set / set_mask / acc / uniq_keys form one 32-byte aligned group of the
                             // argument block: 41.3-41.4 us against 41.5-41.8 (noise); left as it is
                             // (profiles/round6/ab_callwork_tail_alignment.txt)
};


// One lookup call.  Filled on the host (pinned), copied to the session's device buffer, then read by
// every kernel of the call.  The reference's ProcessRequest builds the same per-table pointer slices
// (model_instance_state.cpp:180-193).
struct CallDesc {
  uint32_t num_tables;
  uint32_t epoch;           // time token of the call: (recency unit << 8) | (call counter & 0xFF)
  uint32_t stamp8;          // recency unit of the call modulo 255: the stamp hits of this call leave in their slots
  uint32_t skip_empty_keys; // 1: a key equal to HPS_EMPTY_KEY is padding of the sharded exchange — no probe, no row, no list
                            // entry, slot = -1 (0: it is a key like any other, in no table by construction: default vector)
  uint64_t total_keys;      // N = sum n_t
  const int64_t* keys;      // flat, table-major, device
  const uint32_t* keys32;   // not null: the same keys narrowed to 32 bits (every key of the call is in [0, 2^32); then
                            // `keys` is not read) — halves the host->device bytes of a request's KEYS
  const uint8_t* keys24;    // not null: the keys packed at 3 bytes each, little-endian (every key of the call is in [0, 2^24))
  int64_t key_base[kMaxTables];        // keys32 / keys24 hold key - key_base[table] (frame of reference per table; 0 otherwise)
  uint64_t key_start[kMaxTables + 1];  // prefix sums of n_t
  float* out[kMaxTables];              // device pointer of table t's output slice
  uint8_t vec_ok[kMaxTables];          // 1: D%4==0 and out[t] 16-B aligned -> float4 path
  const uint32_t* dst_index;           // not null (table-sharded lookup, shard_entry.h): the row of key i goes to
                                       // out[t] + dst_index[i] * D_t instead of out[t] + (i - key_start[t]) * D_t — the keys of
                                       // this call are one owner's bucket of a larger request and every row is written
                                       // straight to its place in the request's output (possibly on another GPU, over xGMI)
};

// Second descriptor, valid after the host has sized the miss staging (per call, per chunk).
struct MissDesc {
  uint64_t useg_start[kMaxTables + 1];  // prefix sums of unique-miss counts per table (flat index space)
  uint64_t stage_off[kMaxTables];       // float offset of table t's first staged row
  uint32_t chunk_lo[kMaxTables];        // this chunk covers uidx in [chunk_lo[t], chunk_hi[t]) of table t
  uint32_t chunk_hi[kMaxTables];
};

}  // namespace hps
