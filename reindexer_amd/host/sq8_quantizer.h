// SQ8 scalar quantiser of the HNSW Map — the host half of the quantised search path (the device half: batch_distances_sq8, hnsw_search.hip).
// Mirrors hnswlib::Quantizer (cpp_src/core/index/float_vector/scalar_quantization/quantizer.h:11-124) and the parameter derivation of
// QuantizingParams (quantization_params.h:60-63): codes = clamp((v - minQ) / alpha, 0, 255) truncated, plus the first-order corrective
// offset of the vector that DistCalculator<uint8_t> adds to every distance (hnswlib.h:123-165).
// Float operation order matters (the offsets take part in result distances bit for bit): built with -ffp-contract=off like the rest.
#pragma once

#include <cmath>
#include <cstddef>
#include <cstdint>

#include "rx_types.h"

namespace rxgpu::host {

inline constexpr float kSq8Range = 255.f;   // type_consts.h

struct Sq8Params {
	float minQ = 0.f, maxQ = 0.f, alpha = 0.f, alpha_2 = 0.f, delta = 0.f;

	// quantization_params.h:60-63 / Quantizer::PrepareToRequantize (quantizer.h:45-53)
	static Sq8Params FromRange(float minQ, float maxQ, size_t dim) noexcept {
		Sq8Params p;
		p.minQ = minQ;
		p.maxQ = maxQ;
		p.alpha = (maxQ - minQ) / kSq8Range;
		p.alpha_2 = std::pow(p.alpha, 2.f);
		p.delta = float(0.5 * double(std::pow(minQ, 2.f)) * double(dim));
		return p;
	}
};

// Quantizer::quantize (quantizer.h:93-124).  `scale` multiplies every component first: prepareData's `norm * val` for the query of a
// quantised cosine graph (hnswalg.h:510-529); exactly 1 for stored vectors (no multiplication then).  Returns the corrective offset.
inline float Sq8Quantize(VectorMetric metric, const Sq8Params& p, const float* from, size_t dim, float scale, uint8_t* to) noexcept {
	const bool isL2 = metric == VectorMetric::L2;
	float res = 0.f, shift = 0.f;
	for (size_t i = 0; i < dim; ++i) {
		const float val = scale == 1.f ? from[i] : scale * from[i];
		float c = (val - p.minQ) / p.alpha;
		c = c < 0.f ? 0.f : (c > kSq8Range ? kSq8Range : c);
		const uint8_t code = uint8_t(c);   // the float -> uint8 conversion truncates
		const float err = val - (p.alpha * float(code) + p.minQ);
		if (isL2) {
			res += (2 * p.alpha * float(code) + err) * err;
			shift -= 2.f * p.alpha * err * float(code);
		} else {
			res += p.alpha * float(code) + err;
			shift += p.alpha * err * float(code);
		}
		to[i] = code;
	}
	if (!isL2) {
		res *= p.minQ;
		res += p.delta;
	}
	res += shift;
	return res;
}

}  // namespace rxgpu::host
