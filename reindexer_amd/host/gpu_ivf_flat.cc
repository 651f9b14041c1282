#include "gpu_ivf_flat.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>
#include <random>
#include <stdexcept>
#include <string>

#include "gpu_bruteforce_map.h"   // CalculateL2Module / NormalizeCopyVector
#include "rxgpu.h"
#include "sorted_union.h"

namespace rxgpu::host {

namespace {
[[noreturn]] void throwDevice(const char* what) { throw std::runtime_error(std::string(what) + ": " + rxgpu_last_error()); }
constexpr size_t kAssignBatch = 256;          // queries per coarse-quantiser call: one tile of the matrix-core batch path
constexpr size_t kMaxPointsPerCentroid = 256; // Clustering.h:45
constexpr int kIterations = 10;               // Level1Quantizer: cp.niter = 10 (IndexIVF.cpp:48)
constexpr float kSplitEps = 1.0f / 1024.0f;   // Clustering.cpp: EPS of split_clusters

}  // namespace

GpuIvfFlat::GpuIvfFlat(VectorMetric metric, size_t dim, size_t nlist, int device)
	: metric_(metric), dim_(dim), nlist_(nlist), device_(device), lists_(nlist) {
	if (dim_ == 0 || nlist_ == 0) throw std::logic_error("GpuIvfFlat: zero dimension or zero centroids");
	if (rxgpu_index_create(int(metric_), uint32_t(dim_), 0, device_, &dev_) != RXGPU_OK) throwDevice("GpuIvfFlat: device index creation failed");
	const int coarseMetric = metric_ == VectorMetric::L2 ? RXGPU_METRIC_L2 : RXGPU_METRIC_IP;
	if (rxgpu_index_create(coarseMetric, uint32_t(dim_), nlist_, device_, &devCentroids_) != RXGPU_OK) {
		rxgpu_index_destroy(dev_);
		dev_ = nullptr;
		throwDevice("GpuIvfFlat: centroid index creation failed");
	}
}

GpuIvfFlat::~GpuIvfFlat() {
	if (dev_) rxgpu_index_destroy(dev_);
	if (devCentroids_) rxgpu_index_destroy(devCentroids_);
}

void GpuIvfFlat::Reset() {
	rows_.clear();
	invNorms_.clear();
	ids_.clear();
	listOf_.clear();
	idToRow_.clear();
	for (auto& l : lists_) l.clear();
	centroids_.clear();
	count_ = 0;
	trained_ = false;
	if (rxgpu_index_truncate(dev_, 0) != RXGPU_OK) throwDevice("GpuIvfFlat::Reset");
}

void GpuIvfFlat::reserveRows(size_t need) {
	if (need <= capacity_) return;
	const size_t cap = std::max(need, capacity_ * 2);
	if (rxgpu_index_reserve(dev_, cap) != RXGPU_OK) throwDevice("GpuIvfFlat: device reserve failed");
	capacity_ = cap;
}

// cosine: the vector is normalised before it meets a centroid or the stored rows (IndexIVF.cpp:195-215 and the engines' cosine contract)
void GpuIvfFlat::prepareQuery(const float* x, std::vector<float>& q) const {
	q.resize(dim_);
	if (metric_ == VectorMetric::Cosine) {
		NormalizeCopyVector(x, int32_t(dim_), q.data());
	} else {
		std::memcpy(q.data(), x, dim_ * sizeof(float));
	}
}

void GpuIvfFlat::uploadCentroids() const {
	if (rxgpu_index_upload_rows(devCentroids_, 0, nlist_, centroids_.data(), nullptr) != RXGPU_OK) throwDevice("GpuIvfFlat: centroid upload failed");
}

void GpuIvfFlat::assign(const float* xPrepared, size_t n, std::vector<uint32_t>& out) const {
	out.resize(n);
	std::vector<float> dist(kAssignBatch);
	std::vector<uint32_t> cnt(kAssignBatch);
	for (size_t first = 0; first < n; first += kAssignBatch) {
		const uint32_t nq = uint32_t(std::min(kAssignBatch, n - first));
		if (rxgpu_search_knn(devCentroids_, xPrepared + first * dim_, nq, 1, dist.data(), out.data() + first, cnt.data()) != RXGPU_OK) {
			throwDevice("GpuIvfFlat: coarse assignment failed");
		}
	}
}

void GpuIvfFlat::listInsert(uint32_t list, uint32_t row) {
	auto& l = lists_[list];
	l.insert(std::lower_bound(l.begin(), l.end(), row), row);
}

void GpuIvfFlat::listErase(uint32_t list, uint32_t row) {
	auto& l = lists_[list];
	const auto it = std::lower_bound(l.begin(), l.end(), row);
	if (it == l.end() || *it != row) throw std::logic_error("GpuIvfFlat: inverted list out of sync");
	l.erase(it);
}

void GpuIvfFlat::Train(int seed) {
	if (count_ < nlist_) throw std::runtime_error("Number of training points should be at least as large as number of clusters");
	std::mt19937 rng{uint32_t(seed)};
	// the training set: every vector, or nlist * 256 of them picked at random (Clustering::train_encoded subsampling)
	std::vector<uint32_t> pick(count_);
	std::iota(pick.begin(), pick.end(), 0u);
	const size_t ns = std::min(count_, nlist_ * kMaxPointsPerCentroid);
	if (ns < count_) {
		std::shuffle(pick.begin(), pick.end(), rng);
		pick.resize(ns);
	}
	std::vector<float> pts(ns * dim_);
	for (size_t i = 0; i < ns; ++i) {
		const float* src = rows_.data() + size_t(pick[i]) * dim_;
		float* dst = pts.data() + i * dim_;
		if (metric_ == VectorMetric::Cosine) {
			NormalizeCopyVector(src, int32_t(dim_), dst);
		} else {
			std::memcpy(dst, src, dim_ * sizeof(float));
		}
	}
	const bool spherical = metric_ != VectorMetric::L2;   // IndexIVF.cpp:179-182
	auto normalise = [&](float* c) {
		double s = 0;
		for (size_t j = 0; j < dim_; ++j) s += double(c[j]) * double(c[j]);
		if (s > 0) {
			const float k = float(1.0 / std::sqrt(s));
			for (size_t j = 0; j < dim_; ++j) c[j] *= k;
		}
	};
	// initial centroids: nlist distinct training points
	std::vector<uint32_t> perm(ns);
	std::iota(perm.begin(), perm.end(), 0u);
	std::shuffle(perm.begin(), perm.end(), rng);
	centroids_.assign(nlist_ * dim_, 0.f);
	for (size_t c = 0; c < nlist_; ++c) {
		std::memcpy(centroids_.data() + c * dim_, pts.data() + size_t(perm[c]) * dim_, dim_ * sizeof(float));
		if (spherical) normalise(centroids_.data() + c * dim_);
	}
	std::vector<uint32_t> a;
	std::vector<double> sum(nlist_ * dim_);
	std::vector<size_t> hassign(nlist_);
	std::uniform_real_distribution<float> uni(0.f, 1.f);
	for (int it = 0; it < kIterations; ++it) {
		uploadCentroids();
		assign(pts.data(), ns, a);
		std::fill(sum.begin(), sum.end(), 0.0);
		std::fill(hassign.begin(), hassign.end(), size_t(0));
		for (size_t i = 0; i < ns; ++i) {
			const float* p = pts.data() + i * dim_;
			double* s = sum.data() + size_t(a[i]) * dim_;
			for (size_t j = 0; j < dim_; ++j) s[j] += double(p[j]);
			++hassign[a[i]];
		}
		for (size_t c = 0; c < nlist_; ++c) {
			if (!hassign[c]) continue;
			float* dst = centroids_.data() + c * dim_;
			const double inv = 1.0 / double(hassign[c]);
			for (size_t j = 0; j < dim_; ++j) dst[j] = float(sum[c * dim_ + j] * inv);
		}
		// split_clusters: an empty cluster takes half of a big one, both nudged apart by +-1/1024
		for (size_t ci = 0; ci < nlist_ && ns > nlist_; ++ci) {
			if (hassign[ci]) continue;
			size_t cj = 0;
			for (;; cj = (cj + 1) % nlist_) {
				const float p = (float(hassign[cj]) - 1.0f) / float(ns - nlist_);
				if (uni(rng) < p) break;
			}
			float* ni = centroids_.data() + ci * dim_;
			float* nj = centroids_.data() + cj * dim_;
			std::memcpy(ni, nj, dim_ * sizeof(float));
			for (size_t j = 0; j < dim_; ++j) {
				if (j % 2 == 0) {
					ni[j] *= 1 + kSplitEps;
					nj[j] *= 1 - kSplitEps;
				} else {
					ni[j] *= 1 - kSplitEps;
					nj[j] *= 1 + kSplitEps;
				}
			}
			hassign[ci] = hassign[cj] / 2;
			hassign[cj] -= hassign[ci];
		}
		if (spherical) {
			for (size_t c = 0; c < nlist_; ++c) normalise(centroids_.data() + c * dim_);
		}
	}
	uploadCentroids();
	trained_ = true;
	// add_with_ids of everything that sat in the flat phase (ivf_index.cc:101-103)
	for (auto& l : lists_) l.clear();
	listOf_.assign(count_, 0u);
	std::vector<float> prepared;
	std::vector<uint32_t> chunkAssign;
	constexpr size_t kChunk = 1 << 15;
	for (size_t first = 0; first < count_; first += kChunk) {
		const size_t n = std::min(kChunk, count_ - first);
		const float* src = rows_.data() + first * dim_;
		if (metric_ == VectorMetric::Cosine) {
			prepared.resize(n * dim_);
			for (size_t i = 0; i < n; ++i) NormalizeCopyVector(src + i * dim_, int32_t(dim_), prepared.data() + i * dim_);
			src = prepared.data();
		}
		assign(src, n, chunkAssign);
		for (size_t i = 0; i < n; ++i) {
			listOf_[first + i] = chunkAssign[i];
			lists_[chunkAssign[i]].push_back(uint32_t(first + i));   // rows ascend: every list stays sorted
		}
	}
}

void GpuIvfFlat::AddWithIds(size_t n, const float* x, const idx_t* ids) {
	if (n == 0) return;
	if (count_ + n > 0xFFFFFFF0ull) throw std::runtime_error("GpuIvfFlat: too many vectors");
	for (size_t i = 0; i < n; ++i) {
		if (idToRow_.count(ids[i])) throw std::logic_error("GpuIvfFlat::AddWithIds: id already present");
	}
	reserveRows(count_ + n);
	rows_.insert(rows_.end(), x, x + n * dim_);
	ids_.insert(ids_.end(), ids, ids + n);
	if (metric_ == VectorMetric::Cosine) {
		for (size_t i = 0; i < n; ++i) invNorms_.push_back(CalculateL2Module(x + i * dim_, int32_t(dim_)));
	}
	for (size_t i = 0; i < n; ++i) {
		if (!idToRow_.emplace(ids[i], uint32_t(count_ + i)).second) {   // duplicate inside the batch: roll back
			for (size_t j = 0; j < i; ++j) idToRow_.erase(ids[j]);
			rows_.resize(count_ * dim_);
			ids_.resize(count_);
			if (metric_ == VectorMetric::Cosine) invNorms_.resize(count_);
			throw std::logic_error("GpuIvfFlat::AddWithIds: duplicate id in the batch");
		}
	}
	if (rxgpu_index_upload_rows(dev_, count_, n, x, metric_ == VectorMetric::Cosine ? invNorms_.data() + count_ : nullptr) != RXGPU_OK) {
		throwDevice("GpuIvfFlat: row upload failed");
	}
	if (trained_) {
		std::vector<float> prepared;
		const float* src = x;
		if (metric_ == VectorMetric::Cosine) {
			prepared.resize(n * dim_);
			for (size_t i = 0; i < n; ++i) NormalizeCopyVector(x + i * dim_, int32_t(dim_), prepared.data() + i * dim_);
			src = prepared.data();
		}
		std::vector<uint32_t> a;
		assign(src, n, a);
		listOf_.resize(count_ + n);
		for (size_t i = 0; i < n; ++i) {
			listOf_[count_ + i] = a[i];
			lists_[a[i]].push_back(uint32_t(count_ + i));
		}
	}
	count_ += n;
}

size_t GpuIvfFlat::RemoveIds(const idx_t* ids, size_t n) {
	size_t removed = 0;
	for (size_t i = 0; i < n; ++i) {
		const auto it = idToRow_.find(ids[i]);
		if (it == idToRow_.end()) continue;
		const uint32_t row = it->second;
		const uint32_t last = uint32_t(count_ - 1);
		idToRow_.erase(it);
		if (trained_) listErase(listOf_[row], row);
		if (row != last) {   // the last vector moves into the hole (host and device alike)
			std::memcpy(rows_.data() + size_t(row) * dim_, rows_.data() + size_t(last) * dim_, dim_ * sizeof(float));
			ids_[row] = ids_[last];
			if (metric_ == VectorMetric::Cosine) invNorms_[row] = invNorms_[last];
			idToRow_[ids_[row]] = row;
			if (trained_) {
				listErase(listOf_[last], last);
				listInsert(listOf_[last], row);
				listOf_[row] = listOf_[last];
			}
			if (rxgpu_index_move_row(dev_, last, row) != RXGPU_OK) throwDevice("GpuIvfFlat: row move failed");
		}
		--count_;
		rows_.resize(count_ * dim_);
		ids_.resize(count_);
		if (metric_ == VectorMetric::Cosine) invNorms_.resize(count_);
		if (trained_) listOf_.resize(count_);
		if (rxgpu_index_truncate(dev_, count_) != RXGPU_OK) throwDevice("GpuIvfFlat: truncate failed");
		++removed;
	}
	return removed;
}

void GpuIvfFlat::coarse(const float* q, size_t nprobe, std::vector<uint32_t>& lists) const {
	const uint32_t np = uint32_t(std::min(std::max<size_t>(nprobe, 1), nlist_));
	std::vector<float> dist(np);
	lists.assign(np, 0u);
	uint32_t cnt = 0;
	if (rxgpu_search_knn(devCentroids_, q, 1, np, dist.data(), lists.data(), &cnt) != RXGPU_OK) throwDevice("GpuIvfFlat: coarse search failed");
	lists.resize(cnt);
}

std::vector<uint32_t> GpuIvfFlat::ProbedRows(const float* x, size_t nprobe) const {
	if (!trained_) throw std::logic_error("GpuIvfFlat::ProbedRows: the index is not trained");
	std::vector<float> q;
	prepareQuery(x, q);
	std::vector<uint32_t> probe;
	coarse(q.data(), nprobe, probe);
	std::vector<const std::vector<uint32_t>*> runs;
	runs.reserve(probe.size());
	for (uint32_t l : probe) runs.push_back(&lists_[l]);
	return SortedUnion(runs, count_);
}

void GpuIvfFlat::Search(const float* x, size_t k, size_t nprobe, float* distances, idx_t* labels) const {
	const float pad = metric_ == VectorMetric::L2 ? std::numeric_limits<float>::infinity() : -std::numeric_limits<float>::infinity();
	std::fill(distances, distances + k, pad);
	std::fill(labels, labels + k, idx_t(-1));
	if (k == 0 || count_ == 0) return;
	std::vector<float> q;
	prepareQuery(x, q);
	std::vector<float> dist(k);
	std::vector<uint32_t> row(k);
	uint32_t cnt = 0;
	if (!trained_) {   // the flat phase: IndexFlat::search
		if (rxgpu_search_knn(dev_, q.data(), 1, uint32_t(k), dist.data(), row.data(), &cnt) != RXGPU_OK) throwDevice("GpuIvfFlat::Search");
	} else {
		std::vector<uint32_t> probe;
		coarse(q.data(), nprobe, probe);
		std::vector<const std::vector<uint32_t>*> runs;
		runs.reserve(probe.size());
		for (uint32_t l : probe) runs.push_back(&lists_[l]);
		const std::vector<uint32_t> rows = SortedUnion(runs, count_);
		if (rows.empty()) return;
		if (rxgpu_search_knn_subset(dev_, q.data(), 1, uint32_t(k), rows.data(), rows.size(), dist.data(), row.data(), &cnt) != RXGPU_OK) {
			throwDevice("GpuIvfFlat::Search");
		}
	}
	for (uint32_t i = 0; i < cnt; ++i) {
		distances[i] = toFaiss(dist[i]);
		labels[i] = ids_[row[i]];
	}
}

void GpuIvfFlat::RangeSearch(const float* x, float radius, size_t nprobe, std::vector<float>& distances, std::vector<idx_t>& labels) const {
	distances.clear();
	labels.clear();
	if (count_ == 0) return;
	std::vector<float> q;
	prepareQuery(x, q);
	const float internal = metric_ == VectorMetric::L2 ? radius : -radius;   // similarity > radius <=> -similarity < -radius
	std::vector<uint32_t> rows;
	if (trained_) {
		std::vector<uint32_t> probe;
		coarse(q.data(), nprobe, probe);
		std::vector<const std::vector<uint32_t>*> runs;
		for (uint32_t l : probe) runs.push_back(&lists_[l]);
		rows = SortedUnion(runs, count_);
		if (rows.empty()) return;
	}
	std::vector<float> dist(1024);
	std::vector<uint32_t> row(1024);
	uint64_t total = 0;
	for (;;) {
		const int rc = trained_ ? rxgpu_search_range_subset(dev_, q.data(), internal, 0, rows.data(), rows.size(), dist.data(), row.data(), dist.size(), &total)
								: rxgpu_search_range(dev_, q.data(), internal, 0, dist.data(), row.data(), dist.size(), &total);
		if (rc == RXGPU_OK) break;
		if (rc != RXGPU_ERR_OVERFLOW) throwDevice("GpuIvfFlat::RangeSearch");
		dist.resize(total);
		row.resize(total);
	}
	distances.resize(total);
	labels.resize(total);
	for (uint64_t i = 0; i < total; ++i) {
		distances[i] = toFaiss(dist[i]);
		labels[i] = ids_[row[i]];
	}
}

}  // namespace rxgpu::host
