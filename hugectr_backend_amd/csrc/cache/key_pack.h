// Narrowing a request's int64 keys while they are copied into the page-locked staging buffer (engine.cpp, lookup()): the keys
// cross PCIe at the width they need.  Each routine copies one task's keys and returns the OR of everything it saw, so that the
// caller learns from one pass whether the width was enough (a negative key has its top bits set and fails every narrow width).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#if defined(__x86_64__)
#include <emmintrin.h>
#endif

namespace hps {

// 8 bytes per key (ids without structure: hashed 64-bit ids do not narrow): a plain copy into the page-locked staging buffer,
// with NON-TEMPORAL stores — the staging buffer is written once and read by the DMA engine, never by this core: a cached
// store would first read every destination line (read-for-ownership), a third of the copy's DRAM traffic.  The caller issues
// the store fence before it hands the buffer to the copy engine (StreamFence).
inline void CopyKeys64Streaming(const int64_t* src, size_t n, int64_t* dst) {
#if defined(__x86_64__)
  size_t j = 0;
  while (j < n && ((uintptr_t)(dst + j) & 15u)) { dst[j] = src[j]; ++j; }   // up to the first 16-byte boundary
  for (; j + 8 <= n; j += 8) {
    const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + j));
    const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + j + 2));
    const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + j + 4));
    const __m128i d = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + j + 6));
    _mm_stream_si128(reinterpret_cast<__m128i*>(dst + j), a);
    _mm_stream_si128(reinterpret_cast<__m128i*>(dst + j + 2), b);
    _mm_stream_si128(reinterpret_cast<__m128i*>(dst + j + 4), c);
    _mm_stream_si128(reinterpret_cast<__m128i*>(dst + j + 6), d);
  }
  for (; j < n; ++j) dst[j] = src[j];
#else
  memcpy(dst, src, n * sizeof(int64_t));
#endif
}
inline void StreamFence() {
#if defined(__x86_64__)
  _mm_sfence();
#endif
}

// Frame of reference: a table's keys are narrowed as offsets from `base` (the table's smallest key, HostTable::min_key — feature
// ids that carry a per-table offset, or any id space that starts high, narrow like ids that start at 0); a key below the
// base wraps to a huge offset and fails the width like any wide key.
// dst[j] = (uint32_t)(src[j] - base)
inline uint64_t PackKeys32(const int64_t* src, size_t n, uint32_t* dst, uint64_t base = 0) {
  uint64_t high = 0;
  for (size_t j = 0; j < n; ++j) { const uint64_t k = (uint64_t)src[j] - base; high |= k; dst[j] = (uint32_t)k; }
  return high;
}

// 3 bytes per key, little-endian, dst[3j .. 3j+2]; writes exactly 3*n bytes.  4-byte stores 3 bytes apart, each overwriting the
// spare byte of the one before; the last key is written byte by byte (the byte behind it belongs to somebody else).
inline uint64_t PackKeys24(const int64_t* src, size_t n, uint8_t* dst, uint64_t base = 0) {
  uint64_t high = 0;
  if (n == 0) return 0;
  size_t j = 0;
  for (; j + 1 < n; ++j) {
    const uint64_t k = (uint64_t)src[j] - base;
    high |= k;
    const uint32_t v = (uint32_t)k;
    memcpy(dst + 3 * j, &v, 4);
  }
  const uint64_t k = (uint64_t)src[j] - base;
  high |= k;
  dst[3 * j] = (uint8_t)k;
  dst[3 * j + 1] = (uint8_t)(k >> 8);
  dst[3 * j + 2] = (uint8_t)(k >> 16);
  return high;
}

// what the probe kernel reads back (kernels.hip, K_P)
inline uint32_t UnpackKey24(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16); }

}  // namespace hps
