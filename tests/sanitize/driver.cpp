// Sanitizer driver for the host side of the engine (the reference has no sanitizer or race-detection
// jobs; this build runs its CPU tier under ASan+UBSan and under TSan).  Built by tests/test_sanitizers.py with g++
// from the product sources csrc/common/{json,config}.cpp, csrc/ps/{thread_pool,host_table}.cpp — no GPU needed.
//
//   driver parse     configuration parser on valid / truncated / hostile inputs
//   driver table     load, duplicates, sentinel key, synthetic + sharded generation, lookups checked against a std::map
//   driver threads   concurrent Fetch from several threads while another thread upserts and reloads; pool stress
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <map>
#include <random>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "cache/key_pack.h"
#include "common/config.h"
#include "common/hps_hash.h"
#include "ps/host_table.h"
#include "ps/thread_pool.h"
#include "ps/update_source.h"

using namespace hps;

#define CHECK(c)                                                            \
  do {                                                                      \
    if (!(c)) { fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #c, __FILE__, __LINE__); return 1; } \
  } while (0)

static const char* kGoodJson = R"({"supportlonglong": true,
  "volatile_db": {"type": "hash_map", "num_partitions": 4},
  "models": [{"model": "m", "sparse_files": ["/a", "/b"], "num_of_worker_buffer_in_pool": 2,
              "embedding_vecsize_per_table": [1, 16], "maxnum_catfeature_query_per_table_per_sample": [2, 26],
              "default_value_for_each_table": [0.0, 1.5], "deployed_device_list": [0], "max_batch_size": 64,
              "gpucache": true, "hit_rate_threshold": 0.9, "gpucacheper": 0.5, "ps_direct_access": false}]})";

static int run_parse() {
  ParameterServerConfig cfg;
  CHECK(ParseParameterServerText(kGoodJson, &cfg).ok());
  CHECK(cfg.models.at("m").num_tables() == 2);
  const std::string good(kGoodJson);
  // every prefix of a valid document, and every single-byte corruption of it, must be rejected or accepted cleanly
  for (size_t n = 0; n < good.size(); n += 3) {
    ParameterServerConfig c;
    (void)ParseParameterServerText(good.substr(0, n), &c);
  }
  std::mt19937 rng(7);
  for (int it = 0; it < 400; ++it) {
    std::string s = good;
    s[rng() % s.size()] = (char)(rng() % 256);
    ParameterServerConfig c;
    (void)ParseParameterServerText(s, &c);
  }
  const char* hostile[] = {"", "{", "[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[", "{\"models\": 3}", "{\"supportlonglong\": true, \"models\": [{}]}",
                           "{\"supportlonglong\": true, \"models\": [{\"model\": \"x\", \"sparse_files\": []}]}",
                           "\"\\u00", "{\"a\": 1e999999}", "nul", "{\"supportlonglong\": \"maybe\"}"};
  for (const char* h : hostile) {
    ParameterServerConfig c;
    (void)ParseParameterServerText(h, &c);
  }
  return 0;
}

static int check_against_map(const HostTable& tb, const std::map<int64_t, size_t>& ref, const std::vector<float>& rows, uint32_t D,
                             const std::vector<int64_t>& q) {
  std::vector<float> out(q.size() * D);
  std::vector<uint8_t> found(q.size());
  tb.Fetch(q.data(), q.size(), out.data(), D, -7.f, found.data());
  for (size_t i = 0; i < q.size(); ++i) {
    auto it = ref.find(q[i]);
    CHECK((it != ref.end()) == (found[i] != 0));
    for (uint32_t j = 0; j < D; ++j) {
      const float want = it != ref.end() ? rows[it->second * D + j] : -7.f;
      CHECK(memcmp(&want, &out[i * D + j], 4) == 0);
    }
  }
  return 0;
}

static int run_table() {
  ThreadPool pool(3);
  for (uint32_t D : {1u, 3u, 16u, 128u}) {
    const size_t R = 20000;
    std::vector<int64_t> keys(R);
    std::vector<float> rows(R * D);
    std::mt19937_64 rng(D);
    for (size_t r = 0; r < R; ++r) keys[r] = (int64_t)(rng() % (R * 3)) - (int64_t)R;   // duplicates on purpose
    keys[17] = HPS_EMPTY_KEY;                                                            // the sentinel is a legal key
    for (auto& v : rows) v = (float)(rng() % 100000) * 0.25f;
    std::map<int64_t, size_t> ref;
    for (size_t r = 0; r < R; ++r) ref[keys[r]] = r;                                    // last duplicate wins
    HostTable tb("t", D, 8);
    CHECK(tb.LoadFromArrays(keys.data(), rows.data(), R, false, &pool).ok());
    CHECK(tb.size() == R && tb.has_duplicate_keys());
    std::vector<int64_t> q;
    for (int i = 0; i < 5000; ++i) q.push_back((int64_t)(rng() % (R * 4)) - (int64_t)R);
    q.push_back(HPS_EMPTY_KEY);
    q.push_back(INT64_MAX);
    if (check_against_map(tb, ref, rows, D, q)) return 1;
    // upserts: new keys and overwrites
    std::vector<int64_t> uk;
    std::vector<float> ur;
    for (int i = 0; i < 3000; ++i) {
      uk.push_back((int64_t)(rng() % (R * 6)));
      for (uint32_t j = 0; j < D; ++j) ur.push_back((float)i + (float)j * 0.5f);
    }
    CHECK(tb.Upsert(uk.data(), ur.data(), uk.size()).ok());
    for (size_t i = 0; i < uk.size(); ++i) {
      const int64_t row = tb.Find(uk[i]);
      CHECK(row >= 0);
    }
  }
  // synthetic generation, whole and sharded: the shards partition the table
  const size_t R = 30011;
  HostTable whole("w", 20, 8);
  CHECK(whole.LoadSynthetic(5, 1, 100, R, &pool).ok());
  size_t total = 0;
  for (uint32_t s = 0; s < 3; ++s) {
    HostTable part("p", 20, 8);
    CHECK(part.LoadSynthetic(5, 1, 100, R, &pool, s, 3).ok());
    total += part.size();
    for (size_t r = 0; r < part.size(); r += 97) {
      const int64_t k = part.key_at(r);
      CHECK(hps_mix64((uint64_t)k) % 3 == s);
      const int64_t wr = whole.Find(k);
      CHECK(wr >= 0 && memcmp(whole.row_at((size_t)wr), part.row_at(r), 20 * 4) == 0);
    }
  }
  CHECK(total == R);
  CHECK(!HostTable("bad", 4, 8).LoadSynthetic(1, 0, 0, 10, &pool, 5, 3).ok());
  return 0;
}

static int run_threads() {
  ThreadPool pool(4, 50);
  const uint32_t D = 32;
  const size_t R = 50000;
  HostTable tb("t", D, 8);
  CHECK(tb.LoadSynthetic(9, 0, 0, R, &pool).ok());
  std::atomic<bool> stop{false};
  std::atomic<int> bad{0};
  auto reader = [&](int seed) {
    std::mt19937_64 rng(seed);
    std::vector<int64_t> q(512);
    std::vector<float> out(q.size() * D);
    std::vector<uint8_t> found(q.size());
    while (!stop.load()) {
      for (auto& k : q) k = (int64_t)(rng() % (R + R / 10));
      tb.Fetch(q.data(), q.size(), out.data(), D, 0.f, found.data());
      for (size_t i = 0; i < q.size(); ++i)
        if ((q[i] < (int64_t)R) != (found[i] != 0) && q[i] < (int64_t)R) bad.fetch_add(1);   // original keys never disappear
    }
  };
  std::vector<std::thread> th;
  for (int i = 0; i < 3; ++i) th.emplace_back(reader, 100 + i);
  std::mt19937_64 rng(1);
  for (int round = 0; round < 30; ++round) {
    std::vector<int64_t> uk(256);
    std::vector<float> ur(uk.size() * D, (float)round);
    for (auto& k : uk) k = (int64_t)(R + rng() % (R / 10));    // new keys only: readers' expectations stay valid
    CHECK(tb.Upsert(uk.data(), ur.data(), uk.size()).ok());
    if (round % 10 == 9) CHECK(tb.LoadSynthetic(9, 0, 0, R, &pool).ok());   // full reload under the readers
    std::atomic<size_t> sum{0};
    pool.ParallelFor(1000, [&](size_t i) { sum.fetch_add(i); });
    CHECK(sum.load() == 999 * 1000 / 2);
    std::atomic<int> fired{0};
    for (int i = 0; i < 16; ++i) pool.Submit([&fired] { fired.fetch_add(1); });
    while (fired.load() < 16) std::this_thread::yield();
  }
  stop.store(true);
  for (auto& t : th) t.join();
  CHECK(bad.load() == 0);
  return 0;
}

// Bounded volatile tier in front of the row store: readers (shared side, statistics by relaxed atomics) against the
// inserts and prunes their own misses cause (exclusive side), plus online updates and a reload underneath.
static int run_tiered() {
  ThreadPool pool(4, 50);
  const uint32_t D = 8;
  const size_t R = 20000;
  HostTable tb("t", D, 4);
  HostTierOptions opt;
  opt.tiered = true;
  opt.persistent = true;
  opt.store_writable = true;
  opt.vdb.overflow_margin = 300;
  opt.vdb.overflow_policy = DatabaseOverflowPolicy::EvictLeastUsed;
  opt.vdb.overflow_resolution_target = 0.6;
  opt.vdb.initial_cache_rate = 0.02;
  opt.vdb.cache_missed_embeddings = true;
  tb.SetTierOptions(opt);
  CHECK(tb.LoadSynthetic(11, 0, 0, R, &pool).ok());
  CHECK(tb.tiered());
  HostTable plain("p", D, 4);
  CHECK(plain.LoadSynthetic(11, 0, 0, R, &pool).ok());
  std::atomic<bool> stop{false};
  std::atomic<int> bad{0};
  std::atomic<size_t> fetches{0};
  auto reader = [&](int seed) {
    std::mt19937_64 rng(seed);
    std::vector<int64_t> q(256);
    std::vector<float> out(q.size() * D), ref(q.size() * D);
    while (!stop.load()) {
      for (auto& k : q) k = (int64_t)(rng() % (R + R / 20));
      fetches.fetch_add(1);
      tb.Fetch(q.data(), q.size(), out.data(), D, -1.f, nullptr);
      plain.Fetch(q.data(), q.size(), ref.data(), D, -1.f, nullptr);
      for (size_t i = 0; i < q.size(); ++i)
        if (q[i] < (int64_t)R / 2 && memcmp(&out[i * D], &ref[i * D], D * 4) != 0) bad.fetch_add(1);   // upper half is updated below
    }
  };
  std::vector<std::thread> th;
  for (int i = 0; i < 3; ++i) th.emplace_back(reader, 200 + i);
  std::mt19937_64 rng(2);
  for (int round = 0; round < 40 || fetches.load() < 900; ++round) {
    std::vector<int64_t> uk(64);
    std::vector<float> ur(uk.size() * D, (float)round);
    for (auto& k : uk) k = (int64_t)(R / 2 + rng() % (R / 2 + 100));     // overwrites of the upper half and a few new keys
    CHECK(tb.Upsert(uk.data(), ur.data(), uk.size()).ok());
    const HostTierStats st = tb.tier_stats();
    CHECK(st.vdb.max_partition_entries <= 300);
    std::vector<int64_t> held;
    tb.DumpVolatileKeys(&held);
    CHECK(held.size() <= 4 * 300);
    if (round == 20) CHECK(tb.LoadSynthetic(11, 0, 0, R, &pool).ok());
  }
  stop.store(true);
  for (auto& t : th) t.join();
  CHECK(bad.load() == 0);
  CHECK(tb.tier_stats().vdb.overflows > 0);
  return 0;
}

// Key narrowing of the lookup's staging step (csrc/cache/key_pack.h) on buffers of exactly the size it may touch: under ASan
// a byte too many is a heap-buffer-overflow.  Task boundaries as engine.cpp cuts them: consecutive tasks pack into one array.
static int run_keypack() {
  std::mt19937_64 rng(24);
  for (size_t n : {size_t(0), size_t(1), size_t(2), size_t(3), size_t(31), size_t(32768), size_t(100001)}) {
    std::vector<int64_t> keys(n);
    for (auto& k : keys) k = (int64_t)(rng() & 0xFFFFFFu);
    std::unique_ptr<uint8_t[]> packed(new uint8_t[3 * n + (n ? 0 : 1)]);
    // three tasks of uneven size, as if three pool threads packed their ranges side by side
    const size_t cut1 = n / 3, cut2 = n / 3 + n / 2 > n ? n : n / 3 + n / 2;
    uint64_t high = 0;
    high |= hps::PackKeys24(keys.data(), cut1, packed.get());
    high |= hps::PackKeys24(keys.data() + cut2, n - cut2, packed.get() + 3 * cut2);   // out of order on purpose
    high |= hps::PackKeys24(keys.data() + cut1, cut2 - cut1, packed.get() + 3 * cut1);
    CHECK((high >> 24) == 0);
    for (size_t j = 0; j < n; ++j) CHECK(hps::UnpackKey24(packed.get() + 3 * j) == (uint32_t)keys[j]);
    std::unique_ptr<uint32_t[]> p32(new uint32_t[n + (n ? 0 : 1)]);
    CHECK((hps::PackKeys32(keys.data(), n, p32.get()) >> 32) == 0);
    for (size_t j = 0; j < n; ++j) CHECK(p32[j] == (uint32_t)keys[j]);
    if (n >= 3) {
      keys[n / 2] = (int64_t)1 << 27;
      CHECK((hps::PackKeys24(keys.data(), n, packed.get()) >> 24) != 0 && (hps::PackKeys32(keys.data(), n, p32.get()) >> 32) == 0);
      keys[n / 2] = -5;
      CHECK((hps::PackKeys24(keys.data(), n, packed.get()) >> 24) != 0 && (hps::PackKeys32(keys.data(), n, p32.get()) >> 32) != 0);
    }
    // frame of reference: ids that start high (and a negative base) narrow as offsets from the table's smallest key; a key
    // below the base wraps to a huge offset and fails every width
    for (int64_t base : {(int64_t)1 << 40, (int64_t)-1000000, INT64_MAX - (int64_t)0xFFFFFF}) {
      for (auto& k : keys) k = base + (int64_t)(rng() & 0xFFFFFFu);
      CHECK((hps::PackKeys24(keys.data(), n, packed.get(), (uint64_t)base) >> 24) == 0);
      for (size_t j = 0; j < n; ++j) CHECK(base + (int64_t)hps::UnpackKey24(packed.get() + 3 * j) == keys[j]);
      CHECK((hps::PackKeys32(keys.data(), n, p32.get(), (uint64_t)base) >> 32) == 0);
      for (size_t j = 0; j < n; ++j) CHECK(base + (int64_t)(uint64_t)p32[j] == keys[j]);
      if (n >= 1 && base > INT64_MIN + 8) {
        keys[n / 2] = base - 7;
        CHECK((hps::PackKeys32(keys.data(), n, p32.get(), (uint64_t)base) >> 32) != 0);
      }
    }
  }
  return 0;
}

// Online update source: a producer appends framed messages to a file (sometimes half a frame at a time) while the consumer
// thread follows it and upserts into a table that three reader threads are fetching from; at the end every key the producer
// wrote answers with its last row.
static int run_updates() {
  ThreadPool pool(4, 50);
  const uint32_t D = 8;
  const size_t R = 20000;
  HostTable tb("t", D, 8);
  CHECK(tb.LoadSynthetic(5, 0, 0, R, &pool).ok());
  char tmpl[] = "/tmp/hps_updates_XXXXXX";
  const int fd = mkstemp(tmpl);
  CHECK(fd >= 0);
  UpdateSourceParams up;
  up.type = UpdateSourceType::FileTail;
  up.brokers = tmpl;
  up.poll_timeout_ms = 20;
  up.max_batch_size = 100;
  up.max_commit_interval = 4;
  up.failure_backoff_ms = 1;
  std::unique_ptr<UpdateTransport> tr;
  CHECK(MakeFileTailTransport(tmpl, 1 << 20, &tr).ok());
  std::atomic<size_t> commits{0};
  std::map<int64_t, float> want;   // producer's view: key -> value every float of its row has
  {
    UpdateConsumer consumer(up, std::move(tr),
                            [&](const std::string& m, uint32_t t, uint32_t d, const int64_t* k, const float* r, size_t n) -> Status {
                              if (m != "m" || t != 0 || d != D) return Error(Code::kNotFound, "not mine");
                              return tb.Upsert(k, r, n);
                            },
                            [&](const std::set<std::string>&) { commits.fetch_add(1); });
    std::atomic<bool> stop{false};
    std::atomic<int> bad{0};
    auto reader = [&](int seed) {
      std::mt19937_64 rng(seed);
      std::vector<int64_t> q(256);
      std::vector<float> out(q.size() * D);
      std::vector<uint8_t> found(q.size());
      while (!stop.load()) {
        for (auto& k : q) k = (int64_t)(rng() % R);
        tb.Fetch(q.data(), q.size(), out.data(), D, 0.f, found.data());
        for (size_t i = 0; i < q.size(); ++i) if (!found[i]) bad.fetch_add(1);   // the table's own keys never disappear
      }
    };
    std::vector<std::thread> th;
    for (int i = 0; i < 3; ++i) th.emplace_back(reader, 7 + i);
    std::mt19937_64 rng(3);
    for (int msg = 0; msg < 40; ++msg) {
      std::vector<int64_t> k(1 + rng() % 300);
      for (auto& x : k) x = (int64_t)(rng() % (2 * R));   // half overwrite, half new keys
      std::sort(k.begin(), k.end());
      k.erase(std::unique(k.begin(), k.end()), k.end());
      std::vector<float> r(k.size() * D, (float)(msg + 1));
      for (int64_t x : k) want[x] = (float)(msg + 1);
      const std::string frame = EncodeUpdateMessage(msg % 7 == 6 ? "other" : "m", 0, D, k.data(), r.data(), k.size());
      if (msg % 7 == 6) for (int64_t x : k) want.erase(x);   // a message for another model changes nothing here: its keys are not checked
      const size_t cut = msg % 3 == 0 ? frame.size() / 2 : frame.size();
      CHECK(write(fd, frame.data(), cut) == (ssize_t)cut);
      if (cut < frame.size()) {
        std::this_thread::sleep_for(std::chrono::milliseconds(3));
        CHECK(write(fd, frame.data() + cut, frame.size() - cut) == (ssize_t)(frame.size() - cut));
      }
    }
    CHECK(consumer.Drain(20000).ok());
    stop.store(true);
    for (auto& t : th) t.join();
    CHECK(bad.load() == 0);
    const UpdateSourceStats st = consumer.stats();
    CHECK(st.messages == 40 - 5 && st.rejected_messages == 5 && st.commits >= 1 && commits.load() >= 1);
  }
  close(fd);
  // every key whose last writer was a message for "m" carries that message's value
  std::vector<int64_t> q;
  std::vector<float> expect;
  for (auto& kv : want) { q.push_back(kv.first); expect.push_back(kv.second); }
  std::vector<float> out(q.size() * D);
  std::vector<uint8_t> found(q.size());
  tb.Fetch(q.data(), q.size(), out.data(), D, -1.f, found.data());
  int wrong = 0;
  for (size_t i = 0; i < q.size(); ++i)
    if (!found[i] || out[i * D] != expect[i] || out[i * D + D - 1] != expect[i]) ++wrong;
  CHECK(wrong == 0);
  unlink(tmpl);
  unlink((std::string(tmpl) + ".offset").c_str());
  return 0;
}

int main(int argc, char** argv) {
  const std::string what = argc > 1 ? argv[1] : "all";
  int rc = 0;
  if (what == "keypack" || what == "all") rc |= run_keypack();
  if (what == "tiered" || what == "all") rc |= run_tiered();
  if (what == "parse" || what == "all") rc |= run_parse();
  if (what == "table" || what == "all") rc |= run_table();
  if (what == "threads" || what == "all") rc |= run_threads();
  if (what == "updates" || what == "all") rc |= run_updates();
  printf("%s: %s\n", what.c_str(), rc ? "FAILED" : "ok");
  return rc;
}
