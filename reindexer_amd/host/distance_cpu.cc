// The graph builder is the one CPU-bound loop of the host library (12 minutes for 10M x 768): this unit is compiled at -O3 whatever the
// level of the build (measured, 20 000 x 768 cosine, one thread: 1596 -> 1928 inserts/s; the reference engine itself: 1939).  No float
// arithmetic is re-associated by that (no -ffast-math; -ffp-contract=off stays): the link-for-link tests against the engine hold.
#if defined(__GNUC__) && !defined(__clang__)
#pragma GCC optimize("O3")
#endif
#include "distance_cpu.h"

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace rxgpu::host {

namespace {

// (s0+s1)+(s2+s3) was already applied; v holds the 16 surviving partial sums.  16 -> 8 -> 4 -> (t0+t2)+(t1+t3).
inline float Fold16(const float* v) noexcept {
	float a[8], b[4];
	for (int j = 0; j < 8; ++j) a[j] = v[j] + v[j + 8];
	for (int j = 0; j < 4; ++j) b[j] = a[j] + a[j + 4];
	return (b[0] + b[2]) + (b[1] + b[3]);
}

enum class Op { L2, IP };

// Portable form: 64 explicit chains.  Slow, bit-identical to the vector forms below (same per-chain operation order).
template <Op op>
float DistanceScalar(const float* a, const float* b, size_t d) noexcept {
	float s[64] = {};
	const size_t end16 = d & ~size_t(15), end64 = d & ~size_t(63);
	for (size_t i = 0; i < end64; i += 64) {
		for (int l = 0; l < 64; ++l) {
			if constexpr (op == Op::L2) {
				const float df = a[i + l] - b[i + l];
				s[l] = std::fmaf(df, df, s[l]);
			} else {
				s[l] = std::fmaf(a[i + l], b[i + l], s[l]);
			}
		}
	}
	float v[16];
	for (int j = 0; j < 16; ++j) v[j] = (s[j] + s[16 + j]) + (s[32 + j] + s[48 + j]);
	size_t i = end64;
	if constexpr (op == Op::IP) {
		for (; i < end16; i += 16) {
			for (int j = 0; j < 16; ++j) v[j] = std::fmaf(a[i + j], b[i + j], v[j]);
		}
	}
	float tail = 0.0f;
	for (; i < d; ++i) {
		if constexpr (op == Op::L2) {
			const float df = a[i] - b[i];
			tail = std::fmaf(df, df, tail);
		} else {
			tail = std::fmaf(a[i], b[i], tail);
		}
	}
	return Fold16(v) + tail;
}

#if defined(__x86_64__)
// One zmm register = 16 of the 64 chains; four registers carry all of them.
template <Op op>
__attribute__((target("avx512f,fma"))) float DistanceAvx512(const float* a, const float* b, size_t d) noexcept {
	__m512 acc[4] = {_mm512_setzero_ps(), _mm512_setzero_ps(), _mm512_setzero_ps(), _mm512_setzero_ps()};
	const size_t end16 = d & ~size_t(15), end64 = d & ~size_t(63);
	for (size_t i = 0; i < end64; i += 64) {
		for (int r = 0; r < 4; ++r) {
			const __m512 x = _mm512_loadu_ps(a + i + 16 * r), y = _mm512_loadu_ps(b + i + 16 * r);
			if constexpr (op == Op::L2) {
				const __m512 df = _mm512_sub_ps(x, y);
				acc[r] = _mm512_fmadd_ps(df, df, acc[r]);
			} else {
				acc[r] = _mm512_fmadd_ps(x, y, acc[r]);
			}
		}
	}
	__m512 v = _mm512_add_ps(_mm512_add_ps(acc[0], acc[1]), _mm512_add_ps(acc[2], acc[3]));
	size_t i = end64;
	if constexpr (op == Op::IP) {
		for (; i < end16; i += 16) v = _mm512_fmadd_ps(_mm512_loadu_ps(a + i), _mm512_loadu_ps(b + i), v);
	}
	alignas(64) float lanes[16];
	_mm512_store_ps(lanes, v);
	float tail = 0.0f;
	for (; i < d; ++i) {
		if constexpr (op == Op::L2) {
			const float df = a[i] - b[i];
			tail = __builtin_fmaf(df, df, tail);
		} else {
			tail = __builtin_fmaf(a[i], b[i], tail);
		}
	}
	return Fold16(lanes) + tail;
}

// Eight ymm registers = the same 64 chains on AVX2+FMA hosts.
template <Op op>
__attribute__((target("avx2,fma"))) float DistanceAvx2(const float* a, const float* b, size_t d) noexcept {
	__m256 acc[8];
	for (auto& r : acc) r = _mm256_setzero_ps();
	const size_t end16 = d & ~size_t(15), end64 = d & ~size_t(63);
	for (size_t i = 0; i < end64; i += 64) {
		for (int r = 0; r < 8; ++r) {
			const __m256 x = _mm256_loadu_ps(a + i + 8 * r), y = _mm256_loadu_ps(b + i + 8 * r);
			if constexpr (op == Op::L2) {
				const __m256 df = _mm256_sub_ps(x, y);
				acc[r] = _mm256_fmadd_ps(df, df, acc[r]);
			} else {
				acc[r] = _mm256_fmadd_ps(x, y, acc[r]);
			}
		}
	}
	// chains j and j+16, j+32, j+48 live in acc[j/8], acc[j/8+2], acc[j/8+4], acc[j/8+6]
	__m256 lo = _mm256_add_ps(_mm256_add_ps(acc[0], acc[2]), _mm256_add_ps(acc[4], acc[6]));
	__m256 hi = _mm256_add_ps(_mm256_add_ps(acc[1], acc[3]), _mm256_add_ps(acc[5], acc[7]));
	size_t i = end64;
	if constexpr (op == Op::IP) {
		for (; i < end16; i += 16) {
			lo = _mm256_fmadd_ps(_mm256_loadu_ps(a + i), _mm256_loadu_ps(b + i), lo);
			hi = _mm256_fmadd_ps(_mm256_loadu_ps(a + i + 8), _mm256_loadu_ps(b + i + 8), hi);
		}
	}
	alignas(32) float lanes[16];
	_mm256_store_ps(lanes, lo);
	_mm256_store_ps(lanes + 8, hi);
	float tail = 0.0f;
	for (; i < d; ++i) {
		if constexpr (op == Op::L2) {
			const float df = a[i] - b[i];
			tail = __builtin_fmaf(df, df, tail);
		} else {
			tail = __builtin_fmaf(a[i], b[i], tail);
		}
	}
	return Fold16(lanes) + tail;
}
#endif

using DistFn = float (*)(const float*, const float*, size_t) noexcept;

template <Op op>
DistFn Pick() noexcept {
#if defined(__x86_64__)
	__builtin_cpu_init();
	if (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("fma")) return &DistanceAvx512<op>;
	if (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) return &DistanceAvx2<op>;
#endif
	return &DistanceScalar<op>;
}

}  // namespace

float L2SqrAvx512Order(const float* a, const float* b, size_t d) noexcept {
	static const DistFn fn = Pick<Op::L2>();
	return fn(a, b, d);
}

float InnerProductAvx512Order(const float* a, const float* b, size_t d) noexcept {
	static const DistFn fn = Pick<Op::IP>();
	return fn(a, b, d);
}

}  // namespace rxgpu::host
