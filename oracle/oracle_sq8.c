/* TEST INFRASTRUCTURE ONLY — CPU restatement of the reference's SQ8 (scalar-quantised, uint8) distance path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; nothing under reindexer_amd/ does.
 * No GPU kernel consumes uint8 vectors yet: this file is the checker for the NEXT row, pinned against the real reference
 * (oracle/_ref/libref_oracle.so: ref_l2sqr_u8 / ref_ip_u8 / ref_sq8_quantize / ref_sq8_dist_*) by tests/test_sq8_oracle.py.
 *
 * Restates, for the AVX-512 dispatch level (RX_TARGET_INSTRUCTIONS=avx512, the pinned build of the float path):
 *   vector_dists::L2SqrDistance<uint8_t>         cpp_src/tools/distances/l2_dist.cc:168-199 (scalar tail :12-26)
 *   vector_dists::InnerProductDistance<uint8_t>  cpp_src/tools/distances/ip_dist.cc:163-192 (scalar tail :11-21)
 *   hnswlib::Quantizer::quantize                 cpp_src/core/index/float_vector/scalar_quantization/quantizer.h:61-124
 *   hnswlib::QuantizingParams (alpha, alpha_2, delta)  scalar_quantization/quantization_params.h:60-63
 *   hnswlib::DistCalculator<uint8_t>             cpp_src/core/index/float_vector/hnswlib/hnswlib.h:123-165, 192-197
 *
 * The integer part is exact, but NOT the reduction: the 16 int32 lane sums of the zmm accumulator are converted to float and added one
 * after another (`result += tmp[i]`), which rounds once the running sum passes 2^24 (D = 768: up to 1e8) — so the lane assignment and the
 * order of that float sum are part of the contract.  Lane j of the accumulator collects elements {2j, 2j + 1} of the low half and
 * {32 + 2j, 33 + 2j} of the high half of every 64-byte block (_mm512_cvtepu8_epi16 + _mm512_madd_epi16). */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

enum { ORC_METRIC_L2 = 0, ORC_METRIC_IP = 1, ORC_METRIC_COSINE = 2 };
#define ORC_SQ8_RANGE 255.0f /* kSq8Range, hnswlib/type_consts.h:13 */

float orc_l2sqr_u8(const uint8_t* a, const uint8_t* b, size_t d) {
	const size_t simd_end = d & ~(size_t)63;
	uint32_t lane[16];
	memset(lane, 0, sizeof(lane));
	for (size_t i = 0; i < simd_end; i += 64) {
		for (int half = 0; half < 2; ++half) {
			for (int j = 0; j < 16; ++j) {
				const size_t e = i + (size_t)half * 32 + 2 * (size_t)j;
				const int32_t d0 = (int32_t)a[e] - (int32_t)b[e], d1 = (int32_t)a[e + 1] - (int32_t)b[e + 1];
				lane[j] += (uint32_t)(d0 * d0 + d1 * d1); /* madd_epi16 then add_epi32 */
			}
		}
	}
	float result = 0;
	for (int j = 0; j < 16; ++j) result += (float)lane[j];
	int tail = 0; /* L2Sqr<uint8_t>: int accumulator */
	for (size_t i = simd_end; i < d; ++i) {
		const int t = (int)a[i] - (int)b[i];
		tail += t * t;
	}
	return result + (float)tail;
}

float orc_ip_u8(const uint8_t* a, const uint8_t* b, size_t d) {
	const size_t simd_end = d & ~(size_t)63;
	uint32_t lane[16];
	memset(lane, 0, sizeof(lane));
	for (size_t i = 0; i < simd_end; i += 64) {
		for (int half = 0; half < 2; ++half) {
			for (int j = 0; j < 16; ++j) {
				const size_t e = i + (size_t)half * 32 + 2 * (size_t)j;
				lane[j] += (uint32_t)((int32_t)a[e] * (int32_t)b[e] + (int32_t)a[e + 1] * (int32_t)b[e + 1]);
			}
		}
	}
	float result = 0;
	for (int j = 0; j < 16; ++j) result += (float)lane[j];
	int tail = 0;
	for (size_t i = simd_end; i < d; ++i) tail += (int)a[i] * (int)b[i];
	return result + (float)tail;
}

/* QuantizingParams from (minQ, maxQ): quantization_params.h:60-63 and Quantizer::PrepareToRequantize (quantizer.h:45-53) */
void orc_sq8_params(float min_q, float max_q, size_t dim, float* alpha, float* alpha_2, float* delta) {
	*alpha = (max_q - min_q) / ORC_SQ8_RANGE;
	*alpha_2 = powf(*alpha, 2.f);
	*delta = (float)(0.5 * (double)powf(min_q, 2.f) * (double)dim);
}

/* Quantizer::quantize (quantizer.h:93-124): codes + the first-order corrective offset of the vector.  `scale` multiplies every component
 * first (prepareData's `norm * val` for the query of a quantised cosine graph, hnswalg.h:510-529; 1 for stored vectors). */
float orc_sq8_quantize(int metric, size_t dim, float min_q, float alpha, float delta, const float* from, float scale, uint8_t* to) {
	const int is_l2 = metric == ORC_METRIC_L2;
	float res = 0.f, shift = 0.f;
	for (size_t i = 0; i < dim; ++i) {
		const float val = scale == 1.f ? from[i] : scale * from[i];
		float c = (val - min_q) / alpha; /* float2uint8t: clamp, then the float -> uint8 conversion truncates */
		c = c < 0.f ? 0.f : (c > ORC_SQ8_RANGE ? ORC_SQ8_RANGE : c);
		const uint8_t u8 = (uint8_t)c;
		const float err = val - (alpha * (float)u8 + min_q);
		if (is_l2) {
			res += (2 * alpha * (float)u8 + err) * err;
			shift -= 2.f * alpha * err * (float)u8;
		} else {
			res += alpha * (float)u8 + err;
			shift += alpha * err * (float)u8;
		}
		to[i] = u8;
	}
	if (!is_l2) {
		res *= min_q;
		res += delta;
	}
	res += shift;
	return res;
}

/* DistCalculator<uint8_t>::operator()(v1,id1,v2,id2) (hnswlib.h:123-145): corr_* = the stored corrective offsets, inv_norm_* = the stored
 * 1/|v| of the ORIGINAL float vectors (cosine only; pass 1 otherwise) */
float orc_sq8_dist(int metric, size_t dim, float alpha_2, const uint8_t* a, float corr_a, float inv_norm_a, const uint8_t* b, float corr_b,
				   float inv_norm_b) {
	if (metric == ORC_METRIC_L2) return alpha_2 * orc_l2sqr_u8(a, b, dim) + corr_a + corr_b;
	float dist = -(alpha_2 * orc_ip_u8(a, b, dim) + corr_a + corr_b);
	if (metric == ORC_METRIC_COSINE) {
		dist *= inv_norm_a;
		dist *= inv_norm_b;
	}
	return dist;
}

/* operator()(query, row, id) (hnswlib.h:147-165): the query's offset travels behind its codes; only the row's norm coefficient applies */
float orc_sq8_dist_query(int metric, size_t dim, float alpha_2, const uint8_t* q, float corr_q, const uint8_t* row, float corr_row,
						 float inv_norm_row) {
	if (metric == ORC_METRIC_L2) return alpha_2 * orc_l2sqr_u8(q, row, dim) + corr_q + corr_row;
	float dist = -(alpha_2 * orc_ip_u8(q, row, dim) + corr_q + corr_row);
	if (metric == ORC_METRIC_COSINE) dist *= inv_norm_row;
	return dist;
}

void orc_sq8_dist_query_many(int metric, size_t dim, float alpha_2, const uint8_t* q, float corr_q, const uint8_t* rows, const float* corr,
							 const float* inv_norms, size_t n, float* out) {
	for (size_t i = 0; i < n; ++i) {
		out[i] = orc_sq8_dist_query(metric, dim, alpha_2, q, corr_q, rows + i * dim, corr[i], inv_norms ? inv_norms[i] : 1.f);
	}
}
