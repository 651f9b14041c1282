// SQ8 scalar quantiser of the HNSW Map — the host half of the quantised search path (the device half: batch_distances_sq8, hnsw_search.hip).
// Mirrors hnswlib::Quantizer (cpp_src/core/index/float_vector/scalar_quantization/quantizer.h:11-124) and the parameter derivation of
// QuantizingParams (quantization_params.h:60-63): codes = clamp((v - minQ) / alpha, 0, 255) truncated, plus the first-order corrective
// offset of the vector that DistCalculator<uint8_t> adds to every distance (hnswlib.h:123-165).
// Float operation order matters (the offsets take part in result distances bit for bit): built with -ffp-contract=off like the rest.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <limits>
#include <numeric>
#include <optional>
#include <stdexcept>
#include <utility>
#include <vector>

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

// hnswlib::QuantizationConfig (core/definitions/quantization_config.h:24-44): what an index definition says about SQ8
struct Sq8QuantizationConfig {
	std::optional<float> quantile;         // [0.95, 1]; unset: clamp(1 - 1 / (dim + 1), 0.95, 1) (QuantizingParams::Quantile)
	size_t sampleSize = 20'000;            // kDefaultSampleSize
	size_t quantizationThreshold = 100'000;   // kDefaultQuantizationThreshold (read by HnswIndexBase::QuantizationAvailable, not here)
};
inline constexpr int kSq8SampleBatchSize = 20;   // hnswlib/type_consts.h:14 kSampleBatchSize

// HNSWView::GetSampleIndexes (scalar_quantization/hnsw_view_iterator.h:99-113): a reservoir sample of min(sampleSize, size) internal ids drawn
// with std::rand() — the C library's generator, as in the reference, so that the same srand() state gives the same sample — in ascending order.
inline std::vector<uint32_t> Sq8SampleIndexes(size_t sampleSize, size_t size) {
	sampleSize = std::min(sampleSize, size);
	std::vector<uint32_t> take(sampleSize);
	std::iota(take.begin(), take.end(), 0u);
	for (size_t i = sampleSize; i < size; ++i) {
		const size_t j = size_t(std::rand()) % (i + 1);
		if (j < sampleSize) take[j] = uint32_t(i);
	}
	std::sort(take.begin(), take.end());
	return take;
}

// FindNthMinMax (quantization_params.h:12-44) over `count` values: n = 0.5 (1 - quantile) dataSize passes, each of which takes the smallest and
// the largest value not taken yet (first occurrence on equal values) out of the running; the last pass's pair is the answer.  dataSize is what
// the caller claims (dim x 20 even for a last, shorter batch), exactly as the reference passes it.  A flag per value instead of the
// reference's hash set of taken positions: the same walk.
inline std::pair<float, float> Sq8FindNthMinMax(const float* v, size_t count, size_t dataSize, float quantile) {
	const size_t n = size_t(0.5f * (1 - quantile) * float(dataSize));
	std::vector<uint8_t> taken(count, 0);
	float mn, mx;
	size_t cnt = 0;
	do {
		mn = std::numeric_limits<float>::max();
		mx = std::numeric_limits<float>::lowest();
		size_t minIdx = size_t(-1), maxIdx = size_t(-1);
		for (size_t i = 0; i < count; ++i) {
			if (taken[i]) continue;
			const float el = v[i];
			if (el < mn) {
				mn = el;
				minIdx = i;
			}
			if (el > mx) {
				mx = el;
				maxIdx = i;
			}
		}
		if (n > 0) {
			if (minIdx != size_t(-1)) taken[minIdx] = 1;
			if (maxIdx != size_t(-1)) taken[maxIdx] = 1;
		}
	} while (++cnt < n);
	return {mn, mx};
}

// QuantizingParams(hnsw, config) (quantization_params.h:48-66): the sample's rows in batches of 20 (HNSWView: ceil(sample / 20) batches, the last
// one shorter), per batch the n-th smallest / largest component, [minQ, maxQ] = the means over the batches (float sums in batch order).
// row(id) -> const float* of dim components for an internal id below `count`.
template <typename RowFn>
Sq8Params Sq8SampleParams(size_t count, size_t dim, const Sq8QuantizationConfig& cfg, RowFn&& row) {
	if (count == 0) throw std::runtime_error("Quantize: the index is empty");
	const float quantile = cfg.quantile ? *cfg.quantile : std::clamp(1.f - 1.f / float(dim + 1), 0.95f, 1.f);
	const std::vector<uint32_t> ids = Sq8SampleIndexes(cfg.sampleSize, count);
	std::vector<float> batch(dim * kSq8SampleBatchSize);
	float minQ = 0.f, maxQ = 0.f;
	size_t batches = 0;
	for (size_t b0 = 0; b0 < ids.size(); b0 += kSq8SampleBatchSize) {
		const size_t rows = std::min<size_t>(kSq8SampleBatchSize, ids.size() - b0);
		for (size_t r = 0; r < rows; ++r) std::copy_n(row(ids[b0 + r]), dim, batch.data() + r * dim);
		const auto [mn, mx] = Sq8FindNthMinMax(batch.data(), rows * dim, dim * kSq8SampleBatchSize, quantile);
		minQ += mn;
		maxQ += mx;
		++batches;
	}
	minQ /= float(batches);
	maxQ /= float(batches);
	return Sq8Params::FromRange(minQ, maxQ, dim);
}

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
