/*
 * hps_oracle.c — CPU restatement of the reference's lookup path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call this
 * file.  The product (hugectr_backend_amd/) never does; it fails loudly without its HIP engine.
 *
 * PARITY UNPINNED: the arithmetic of this path lives in NVIDIA/HugeCTR's libhuge_ctr_hps.so
 * (branch `main`, unpinned: /root/reference/test/CI.DockerFile:4,11; linked at
 * /root/reference/hps_backend/CMakeLists.txt:147), which is NOT under /root/reference and
 * cannot be built here (needs CUDA, Triton headers, network).  The reference holds no numeric
 * golden vectors for the path (its CI data lives on an NVIDIA-internal volume,
 * /root/reference/.gitlab-ci.yml:70-72,87).  What the reference *does* pin — file formats,
 * request layout, output shape, default-fill rule, response parameters — is restated below and
 * checked against the structural fixtures in tests/golden/ (see tests/test_oracle_golden.py).
 *
 * What is restated (reference file:line each function follows is cited at the function):
 *   - embedding-table file format            docs/architecture.md:185-218
 *   - request layout (KEYS table-major, NUMKEYS[T])  docs/architecture.md:220-230,
 *                                            hps_backend/src/hps.cc:573-630
 *   - per-table pointer slicing              hps_backend/src/model_instance_state.cpp:177-197
 *   - lookup = hash-map find, else default   docs/hierarchical_parameter_server.md:65-78,244-246
 *   - output element count                   hps_backend/src/hps.cc:620-625
 *
 * Build: make -C oracle   (gcc -O2 -shared -fPIC -pthread)
 */
#include <errno.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* splitmix64 + synthetic table recipe.  Restated here independently of       */
/* hugectr_backend_amd/csrc/common/hps_hash.h; tests compare the two bit for bit.              */
/* ------------------------------------------------------------------------------------------ */
static uint64_t mix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

uint64_t oracle_mix64(uint64_t x) { return mix64(x); }

/* rows[k - key0][j] for k in [key0, key0+count), table `t`: finite fp32 in [0.5,1). */
void oracle_synth_rows(uint64_t seed, uint32_t t, int64_t key0, int64_t count, uint32_t D,
                       float* rows) {
  const uint64_t tb = mix64(seed ^ mix64((uint64_t)t + 1));
  for (int64_t r = 0; r < count; ++r) {
    const uint64_t rb = mix64(tb + (uint64_t)(key0 + r));
    uint32_t* dst = (uint32_t*)(rows + (size_t)r * D);
    for (uint32_t j = 0; j < D; ++j) {
      const uint64_t w = mix64(rb + (uint64_t)(j >> 1));
      const uint32_t m = (j & 1) ? (uint32_t)(w >> 32) : (uint32_t)w;
      dst[j] = 0x3F000000u | (m & 0x007FFFFFu);
    }
  }
}

/* ------------------------------------------------------------------------------------------ */
/* One embedding table = the reference's `hash_map` volatile database for one table:           */
/* key -> vector, "last wins" on duplicate keys (an upsert), unordered.                        */
/* docs/hierarchical_parameter_server.md:65-78 (lookup order), README.md:127-135 (hash_map).   */
/* ------------------------------------------------------------------------------------------ */
typedef struct oracle_table {
  uint32_t D;          /* embedding_vecsize_per_table[t]   backend.cpp:454-460 */
  int64_t R;           /* rows in the file */
  uint64_t cap_mask;   /* open addressing, capacity = power of two >= 2R */
  int64_t* slot_key;   /* cap entries; EMPTY marks free */
  int64_t* slot_row;   /* row index into `rows` */
  const float* rows;   /* R x D, file order */
  float* owned_rows;   /* non-NULL if we malloc'd rows */
  int has_empty_key;   /* the sentinel value itself may be a legal key */
  int64_t empty_key_row;
} oracle_table;

#define ORACLE_EMPTY ((int64_t)0x7FFFFFFFFFFFFFFFll)

static uint64_t oslot(int64_t key, uint64_t mask) {
  /* Fibonacci hashing — deliberately NOT the product's hash. */
  return (((uint64_t)key * 0x9E3779B97F4A7C15ull) >> 17) & mask;
}

static int table_index(oracle_table* tb, const int64_t* keys) {
  uint64_t cap = 16;
  while (cap < (uint64_t)tb->R * 2) cap <<= 1;
  tb->cap_mask = cap - 1;
  tb->slot_key = (int64_t*)malloc(cap * sizeof(int64_t));
  tb->slot_row = (int64_t*)malloc(cap * sizeof(int64_t));
  if (!tb->slot_key || !tb->slot_row) return -ENOMEM;
  for (uint64_t i = 0; i < cap; ++i) tb->slot_key[i] = ORACLE_EMPTY;
  tb->has_empty_key = 0;
  for (int64_t r = 0; r < tb->R; ++r) {
    const int64_t k = keys[r];
    if (k == ORACLE_EMPTY) { tb->has_empty_key = 1; tb->empty_key_row = r; continue; }
    uint64_t s = oslot(k, tb->cap_mask);
    while (tb->slot_key[s] != ORACLE_EMPTY && tb->slot_key[s] != k) s = (s + 1) & tb->cap_mask;
    tb->slot_key[s] = k;
    tb->slot_row[s] = r; /* duplicate key in file: last one wins */
  }
  return 0;
}

/* Build from arrays already in memory (rows are borrowed, not copied). */
oracle_table* oracle_table_from_arrays(const int64_t* keys, const float* rows, int64_t R,
                                       uint32_t D) {
  oracle_table* tb = (oracle_table*)calloc(1, sizeof(oracle_table));
  if (!tb) return NULL;
  tb->D = D; tb->R = R; tb->rows = rows;
  if (table_index(tb, keys) != 0) { free(tb->slot_key); free(tb->slot_row); free(tb); return NULL; }
  return tb;
}

static long file_size(FILE* f) {
  if (fseek(f, 0, SEEK_END) != 0) return -1;
  long n = ftell(f);
  rewind(f);
  return n;
}

/*
 * Load "<dir>/key" (R native-endian int64, no separators, any order) and "<dir>/emb_vector"
 * (R*D native-endian fp32 in the same order).  docs/architecture.md:185-218; writer recipe
 * samples/hps-triton-ensemble/01_model_training.ipynb:498-504 (struct.pack('q') / ('f')).
 */
oracle_table* oracle_table_load(const char* dir, uint32_t D) {
  char path[4096];
  snprintf(path, sizeof path, "%s/key", dir);
  FILE* fk = fopen(path, "rb");
  if (!fk) return NULL;
  snprintf(path, sizeof path, "%s/emb_vector", dir);
  FILE* fv = fopen(path, "rb");
  if (!fv) { fclose(fk); return NULL; }
  const long kb = file_size(fk), vb = file_size(fv);
  oracle_table* tb = NULL;
  int64_t* keys = NULL;
  if (kb < 0 || vb < 0 || kb % 8 != 0) goto done;
  const int64_t R = kb / 8;
  if ((int64_t)vb != R * (int64_t)D * 4) goto done; /* D must match embedding_vecsize_per_table */
  keys = (int64_t*)malloc(kb ? kb : 8);
  float* rows = (float*)malloc(vb ? vb : 4);
  if (!keys || !rows) { free(rows); goto done; }
  if ((long)fread(keys, 1, kb, fk) != kb || (long)fread(rows, 1, vb, fv) != vb) { free(rows); goto done; }
  tb = oracle_table_from_arrays(keys, rows, R, D);
  if (tb) tb->owned_rows = rows; else free(rows);
done:
  free(keys);
  fclose(fk); fclose(fv);
  return tb;
}

void oracle_table_free(oracle_table* tb) {
  if (!tb) return;
  free(tb->slot_key); free(tb->slot_row); free(tb->owned_rows); free(tb);
}

int64_t oracle_table_rows(const oracle_table* tb) { return tb->R; }
uint32_t oracle_table_dim(const oracle_table* tb) { return tb->D; }

/* row index of `key`, or -1.  The "find" of the hash_map database. */
int64_t oracle_table_find(const oracle_table* tb, int64_t key) {
  if (key == ORACLE_EMPTY) return tb->has_empty_key ? tb->empty_key_row : -1;
  uint64_t s = oslot(key, tb->cap_mask);
  while (tb->slot_key[s] != ORACLE_EMPTY) {
    if (tb->slot_key[s] == key) return tb->slot_row[s];
    s = (s + 1) & tb->cap_mask;
  }
  return -1;
}

/* out element count = sum_t D_t * n_t      hps.cc:620-625 (std::inner_product) */
int64_t oracle_output_elems(const oracle_table* const* tables, const int32_t* num_keys, int T) {
  int64_t n = 0;
  for (int t = 0; t < T; ++t) n += (int64_t)tables[t]->D * (int64_t)num_keys[t];
  return n;
}

/*
 * One table's slice: out[i*D .. i*D+D) = rows[find(key_i)] or default_value broadcast.
 * docs/hierarchical_parameter_server.md:244-246 (default_value_for_each_table when the key is in
 * no tier); order = input key order, duplicates each get a full copy.
 */
static void lookup_slice(const oracle_table* tb, const int64_t* keys, int64_t n, float dflt,
                         float* out) {
  const uint32_t D = tb->D;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t r = oracle_table_find(tb, keys[i]);
    float* dst = out + (size_t)i * D;
    if (r >= 0) memcpy(dst, tb->rows + (size_t)r * D, (size_t)D * sizeof(float));
    else for (uint32_t j = 0; j < D; ++j) dst[j] = dflt;
  }
}

/*
 * The whole request.  KEYS is flat and table-major (all keys of table 0, then table 1, ...:
 * docs/architecture.md:220-230); NUMKEYS holds T int32 (hps.cc:616-618).  Pointer slicing is
 * ModelInstanceState::ProcessRequest (model_instance_state.cpp:180-193):
 *     keys_t = keys + sum_{u<t} n_u ;  out_t = out + sum_{u<t} D_u * n_u
 * Returns the number of floats written.
 */
int64_t oracle_lookup(const oracle_table* const* tables, int T, const int64_t* keys,
                      const int32_t* num_keys, const float* default_values, float* out) {
  const int64_t* k = keys;
  float* o = out;
  for (int t = 0; t < T; ++t) {
    lookup_slice(tables[t], k, num_keys[t], default_values[t], o);
    k += num_keys[t];
    o += (size_t)tables[t]->D * (size_t)num_keys[t];
  }
  return (int64_t)(o - out);
}

/* ---- threaded variant: same arithmetic, key range of every table split over `threads`. ---- */
typedef struct { const oracle_table* tb; const int64_t* keys; int64_t n; float dflt; float* out; } job_t;
typedef struct { job_t* jobs; int njobs; int next; pthread_mutex_t mu; } queue_t;

static void* worker(void* arg) {
  queue_t* q = (queue_t*)arg;
  for (;;) {
    pthread_mutex_lock(&q->mu);
    const int j = q->next < q->njobs ? q->next++ : -1;
    pthread_mutex_unlock(&q->mu);
    if (j < 0) return NULL;
    job_t* b = &q->jobs[j];
    lookup_slice(b->tb, b->keys, b->n, b->dflt, b->out);
  }
}

int64_t oracle_lookup_mt(const oracle_table* const* tables, int T, const int64_t* keys,
                         const int32_t* num_keys, const float* default_values, float* out,
                         int threads) {
  if (threads <= 1) return oracle_lookup(tables, T, keys, num_keys, default_values, out);
  const int64_t chunk = 2048;
  int njobs = 0;
  for (int t = 0; t < T; ++t) njobs += (int)((num_keys[t] + chunk - 1) / chunk);
  job_t* jobs = (job_t*)malloc(sizeof(job_t) * (size_t)(njobs ? njobs : 1));
  const int64_t* k = keys; float* o = out; int j = 0;
  for (int t = 0; t < T; ++t) {
    const uint32_t D = tables[t]->D;
    for (int64_t s = 0; s < num_keys[t]; s += chunk) {
      const int64_t n = num_keys[t] - s < chunk ? num_keys[t] - s : chunk;
      jobs[j++] = (job_t){tables[t], k + s, n, default_values[t], o + (size_t)s * D};
    }
    k += num_keys[t]; o += (size_t)D * (size_t)num_keys[t];
  }
  queue_t q = {jobs, njobs, 0, PTHREAD_MUTEX_INITIALIZER};
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
  for (int i = 0; i < threads; ++i) pthread_create(&th[i], NULL, worker, &q);
  for (int i = 0; i < threads; ++i) pthread_join(th[i], NULL);
  free(th); free(jobs);
  return (int64_t)(o - out);
}

/*
 * Write a table in the reference's on-disk format (used to build fixtures; mirrors the
 * notebook writer samples/hps-triton-ensemble/01_model_training.ipynb:498-504).
 */
int oracle_write_table(const char* dir, const int64_t* keys, const float* rows, int64_t R,
                       uint32_t D) {
  char path[4096];
  snprintf(path, sizeof path, "%s/key", dir);
  FILE* fk = fopen(path, "wb");
  if (!fk) return -errno;
  snprintf(path, sizeof path, "%s/emb_vector", dir);
  FILE* fv = fopen(path, "wb");
  if (!fv) { fclose(fk); return -errno; }
  int rc = 0;
  if (fwrite(keys, 8, (size_t)R, fk) != (size_t)R) rc = -EIO;
  if (fwrite(rows, 4, (size_t)R * D, fv) != (size_t)R * D) rc = -EIO;
  fclose(fk); fclose(fv);
  return rc;
}
