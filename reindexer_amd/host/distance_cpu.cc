#include "distance_cpu.h"

namespace rxgpu::host {

#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
#define RXGPU_CPU_CLONES __attribute__((target_clones("arch=skylake-avx512", "arch=haswell", "default"), optimize("tree-vectorize")))
#else
#define RXGPU_CPU_CLONES
#endif

namespace detail {
inline float Fold16(const float* v) noexcept {
	float a[8], b[4];
	for (int j = 0; j < 8; ++j) a[j] = v[j] + v[j + 8];
	for (int j = 0; j < 4; ++j) b[j] = a[j] + a[j + 4];
	return (b[0] + b[2]) + (b[1] + b[3]);
}
}  // namespace detail

RXGPU_CPU_CLONES float L2SqrAvx512Order(const float* a, const float* b, size_t d) noexcept {
	alignas(64) float s[64];
	std::memset(s, 0, sizeof(s));
	const size_t blocks = d & ~size_t(63);
	for (size_t i = 0; i < blocks; i += 64) {
		for (int l = 0; l < 64; ++l) {
			const float df = a[i + l] - b[i + l];
			s[l] = std::fmaf(df, df, s[l]);
		}
	}
	float v[16];
	for (int j = 0; j < 16; ++j) v[j] = (s[j] + s[16 + j]) + (s[32 + j] + s[48 + j]);
	float tail = 0.0f;
	for (size_t i = blocks; i < d; ++i) {
		const float df = a[i] - b[i];
		tail = std::fmaf(df, df, tail);
	}
	return detail::Fold16(v) + tail;
}

RXGPU_CPU_CLONES float InnerProductAvx512Order(const float* a, const float* b, size_t d) noexcept {
	alignas(64) float s[64];
	std::memset(s, 0, sizeof(s));
	const size_t end16 = d & ~size_t(15);
	size_t i = 0;
	for (; i + 64 <= end16; i += 64) {
		for (int l = 0; l < 64; ++l) s[l] = std::fmaf(a[i + l], b[i + l], s[l]);
	}
	float v[16];
	for (int j = 0; j < 16; ++j) v[j] = (s[j] + s[16 + j]) + (s[32 + j] + s[48 + j]);
	for (; i < end16; i += 16) {
		for (int j = 0; j < 16; ++j) v[j] = std::fmaf(a[i + j], b[i + j], v[j]);
	}
	float tail = 0.0f;
	for (i = end16; i < d; ++i) tail = std::fmaf(a[i], b[i], tail);
	return detail::Fold16(v) + tail;
}

}  // namespace rxgpu::host
