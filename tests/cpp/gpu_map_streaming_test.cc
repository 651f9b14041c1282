// Engine-level scenario of the reference's own gtest, written against OUR Map classes to show that they slot in where
// hnswlib::HierarchicalNSW / hnswlib::BruteforceSearch do (same member names, argument meaning and result types):
//   cpp_src/gtests/tests/unit/hnsw_streaming_search_test.cc:129-177 (CompareRecallRatesTest): build an HNSW and a brute-force index over
//   the same N(0, 0.25) points, stream k = 500 results in batches of 50 (ef 100), and require the streaming recall to be within 0.1 of the
//   one-shot SearchKnn(k, ef = 1.1 k) recall against brute force, batches to be duplicate-free and the result sizes to be k.
// No gtest in this image: plain CHECKs, exit code 0 / 1.  Quantisation (the second half of the reference test) is not offered by the GPU
// engine (QuantizationAvailable() == false), like BruteforceSearch.
#include <cstdio>
#include <cstdlib>
#include <optional>
#include <random>
#include <unordered_set>
#include <vector>

#include "gpu_bruteforce_map.h"
#include "gpu_hnsw_map.h"

using namespace rxgpu::host;

static int g_failures = 0;
#define CHECK(cond)                                                              \
	do {                                                                         \
		if (!(cond)) {                                                           \
			std::fprintf(stderr, "CHECK failed at line %d: %s\n", __LINE__, #cond); \
			++g_failures;                                                        \
		}                                                                        \
	} while (0)

namespace {
constexpr size_t kDimension = 768, kSize = 3000, kM = 16, kEfConstruction = 200;   // sq8_test constants of the release build, smaller corpus

std::vector<float> MakePoint(std::mt19937& gen) {
	std::normal_distribution<> nd(0, 0.25);
	std::vector<float> p(kDimension);
	for (float& v : p) v = float(nd(gen));
	return p;
}

float calcRecall(const std::unordered_set<labeltype>& candidate, const std::unordered_set<labeltype>& reference) {
	float res = 0;
	for (auto id : candidate) res += reference.count(id);
	return res / float(reference.size());
}

std::vector<std::pair<float, labeltype>> BestFirst(SearchResultQueue q) {   // the reference's ToMinHeapQueue, as a sorted vector
	std::vector<std::pair<float, labeltype>> v(q.size());
	for (size_t i = q.size(); !q.empty(); q.pop()) v[--i] = q.top();
	return v;
}

void RunMetric(VectorMetric metric) {
	std::mt19937 gen(20260924u + unsigned(metric));
	GpuHnswMap hnsw(metric, kDimension, kSize, kM, kEfConstruction);
	GpuBruteforceMap bf(metric, kDimension, kSize);
	for (size_t label = 0; label < kSize; ++label) {
		const auto point = MakePoint(gen);
		bf.AddPointNoLock(ConstFloatVectorView{point.data(), kDimension}, FloatVectorId{int32_t(label), 0});
		hnsw.AddPointNoLock(ConstFloatVectorView{point.data(), kDimension}, FloatVectorId{int32_t(label), 0});
	}
	const size_t k = 500, batchSize = 50, kMaxBatches = k / batchSize + 1, efBatch = 100;
	auto query = MakePoint(gen);
	std::optional<float> normL2;
	std::vector<float> normalized(kDimension);
	const float* queryData = query.data();
	if (metric == VectorMetric::Cosine) {   // hnsw_index.cc:303-314: the caller normalises for cosine
		normL2 = 1.f / NormalizeCopyVector(query.data(), int32_t(kDimension), normalized.data());
		queryData = normalized.data();
	}
	auto session = hnsw.BeginStreamingSearch(queryData, normL2, StreamingSearchOptions{efBatch});
	std::vector<SearchResultQueue> batches;
	for (;;) {
		auto batch = hnsw.ContinueStreamingSearch(session, batchSize);
		const bool exhausted = batch.exhausted;
		if (!batch.results.empty()) batches.emplace_back(std::move(batch.results));
		if (exhausted || batches.size() >= kMaxBatches) break;
	}
	const size_t ef = size_t(1.1 * k);
	auto hnswRes = BestFirst(hnsw.SearchKnn(queryData, normL2, k, ef));
	auto bfRes = BestFirst(bf.SearchKnn(queryData, std::nullopt, k));
	CHECK(hnswRes.size() == k);
	CHECK(bfRes.size() == k);

	std::unordered_set<labeltype> hnswLabels, bfLabels;
	for (auto& p : hnswRes) CHECK(hnswLabels.insert(p.second).second);
	for (auto& p : bfRes) CHECK(bfLabels.insert(p.second).second);
	const float refRecall = calcRecall(hnswLabels, bfLabels);

	std::unordered_set<labeltype> streamLabels, bfPrefix;
	size_t taken = 0;
	float streamRecall = 0.f, prevWorst = -3.4e38f;
	for (auto& queue : batches) {
		CHECK(queue.size() <= batchSize);
		for (size_t i = 0; i < queue.size() && taken < bfRes.size(); ++i) bfPrefix.insert(bfRes[taken++].second);
		auto best = BestFirst(std::move(queue));
		for (auto& p : best) CHECK(streamLabels.insert(p.second).second);   // no element is handed out twice
		if (!best.empty()) prevWorst = best.back().first;
		streamRecall = calcRecall(streamLabels, bfPrefix);
	}
	(void)prevWorst;
	std::printf("metric %d: one-shot recall@%zu %.3f, streaming recall@%zu %.3f in %zu batches\n", int(metric), k, refRecall, bfPrefix.size(),
				streamRecall, batches.size());
	CHECK(batches.size() == kMaxBatches);
	if (streamRecall < refRecall) CHECK(refRecall - streamRecall <= 0.1f);
	// a foreign session is reported exhausted (hnswalg.h:1953-1956)
	GpuHnswMap other(metric, kDimension, 4, kM, kEfConstruction);
	CHECK(other.ContinueStreamingSearch(session, 10).exhausted);
}
}  // namespace

int main() {
	for (VectorMetric m : {VectorMetric::L2, VectorMetric::InnerProduct, VectorMetric::Cosine}) RunMetric(m);
	if (g_failures) {
		std::fprintf(stderr, "%d check(s) failed\n", g_failures);
		return 1;
	}
	std::puts("GpuMapStreamingTest: all checks passed");
	return 0;
}
