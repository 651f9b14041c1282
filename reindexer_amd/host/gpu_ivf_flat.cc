#include "gpu_ivf_flat.h"

#include <algorithm>
#include <atomic>
#include <exception>
#include <mutex>
#include <thread>
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
	: metric_(metric), dim_(dim), nlist_(nlist), device_(device), lists_(nlist), scan_(nlist) {
	if (dim_ == 0 || nlist_ == 0) throw std::logic_error("GpuIvfFlat: zero dimension or zero centroids");
	if (rxgpu_index_create(int(metric_), uint32_t(dim_), 0, device_, &dev_) != RXGPU_OK) throwDevice("GpuIvfFlat: device index creation failed");
	// the coarse quantiser is the flat index of the same metric (IndexFlatCosine ranks by inner product x the stored 1 / |centroid|)
	const int coarseMetric = int(metric_);
	if (rxgpu_index_create(coarseMetric, uint32_t(dim_), nlist_, device_, &devCentroids_) != RXGPU_OK) {
		rxgpu_index_destroy(dev_);
		dev_ = nullptr;
		throwDevice("GpuIvfFlat: centroid index creation failed");
	}
}

GpuIvfFlat::GpuIvfFlat(const GpuIvfFlat& o, int device)
	: metric_(o.metric_), dim_(o.dim_), nlist_(o.nlist_), device_(device), trained_(o.trained_), count_(o.count_), rows_(o.rows_), invNorms_(o.invNorms_),
	  ids_(o.ids_), listOf_(o.listOf_), idToRow_(o.idToRow_), lists_(o.lists_), scan_(o.scan_), scanPos_(o.scanPos_), centroids_(o.centroids_) {
	if (rxgpu_index_create(int(metric_), uint32_t(dim_), std::max<size_t>(count_, 1), device_, &dev_) != RXGPU_OK) throwDevice("GpuIvfFlat: device index creation failed");
	capacity_ = std::max<size_t>(count_, 1);
	if (rxgpu_index_create(int(metric_), uint32_t(dim_), nlist_, device_, &devCentroids_) != RXGPU_OK) {
		rxgpu_index_destroy(dev_);
		dev_ = nullptr;
		throwDevice("GpuIvfFlat: centroid index creation failed");
	}
	if (count_ && rxgpu_index_upload_rows(dev_, 0, count_, rows_.data(), metric_ == VectorMetric::Cosine ? invNorms_.data() : nullptr) != RXGPU_OK) {
		throwDevice("GpuIvfFlat: row upload failed");
	}
	if (trained_) uploadCentroids();
	listsDirty_ = true;
}

const float* GpuIvfFlat::VectorById(idx_t id) const {
	const auto it = idToRow_.find(id);
	if (it == idToRow_.end()) throw std::runtime_error("GpuIvfFlat: id not found");
	return rows_.data() + size_t(it->second) * dim_;
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
	for (auto& l : scan_) l.clear();
	scanPos_.clear();
	centroids_.clear();
	count_ = 0;
	trained_ = false;
	listsDirty_ = true;
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
	std::vector<float> coefs;
	if (metric_ == VectorMetric::Cosine) {   // IndexFlatCodes::add (IndexFlatCodes.cpp:33-40): norm_coefs = NormalizeVector's coefficient
		coefs.resize(nlist_);
		for (size_t c = 0; c < nlist_; ++c) coefs[c] = CalculateL2Module(centroids_.data() + c * dim_, int32_t(dim_));
	}
	if (rxgpu_index_upload_rows(devCentroids_, 0, nlist_, centroids_.data(), coefs.empty() ? nullptr : coefs.data()) != RXGPU_OK) {
		throwDevice("GpuIvfFlat: centroid upload failed");
	}
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
	listsDirty_ = true;
	auto& l = lists_[list];
	l.insert(std::lower_bound(l.begin(), l.end(), row), row);
}

void GpuIvfFlat::scanAppend(uint32_t list, uint32_t row) {
	if (scanPos_.size() <= row) scanPos_.resize(size_t(row) + 1);
	scanPos_[row] = uint32_t(scan_[list].size());
	scan_[list].push_back(row);
}

void GpuIvfFlat::listErase(uint32_t list, uint32_t row) {
	listsDirty_ = true;
	auto& l = lists_[list];
	const auto it = std::lower_bound(l.begin(), l.end(), row);
	if (it == l.end() || *it != row) throw std::logic_error("GpuIvfFlat: inverted list out of sync");
	l.erase(it);
	// DirectMap::remove_ids (Hashtable): the list's last entry takes the place of the removed one
	auto& sc = scan_[list];
	const uint32_t pos = scanPos_[row], moved = sc.back();
	sc[pos] = moved;
	scanPos_[moved] = pos;
	sc.pop_back();
}

void GpuIvfFlat::listRename(uint32_t list, uint32_t from, uint32_t to) {
	listsDirty_ = true;
	auto& l = lists_[list];
	const auto it = std::lower_bound(l.begin(), l.end(), from);
	if (it == l.end() || *it != from) throw std::logic_error("GpuIvfFlat: inverted list out of sync");
	l.erase(it);
	l.insert(std::lower_bound(l.begin(), l.end(), to), to);
	if (scanPos_.size() <= to) scanPos_.resize(size_t(to) + 1);
	scan_[list][scanPos_[from]] = to;   // the entry keeps its place in the list: only the row number behind it changed
	scanPos_[to] = scanPos_[from];
}

namespace {
// faiss::RandomGenerator (utils/random.cpp:35-51): std::mt19937 seeded with the low 32 bits; rand_int(max) = mt() % max,
// rand_float() = mt() / float(mt.max())
struct FaissRng {
	std::mt19937 mt;
	explicit FaissRng(int64_t seed) : mt(static_cast<unsigned int>(seed)) {}
	int rand_int(int max) { return int(mt() % static_cast<std::mt19937::result_type>(max)); }
	float rand_float() { return float(mt()) / float(mt.max()); }
};
// faiss::rand_perm (utils/random.cpp:184-194)
std::vector<int> faissRandPerm(size_t n, int64_t seed) {
	std::vector<int> perm(n);
	for (size_t i = 0; i < n; ++i) perm[i] = int(i);
	FaissRng rng(seed);
	for (size_t i = 0; i + 1 < n; ++i) {
		const int i2 = int(i) + rng.rand_int(int(n - i));
		std::swap(perm[i], perm[size_t(i2)]);
	}
	return perm;
}
}  // namespace

// faiss::Clustering::train_encoded as Level1Quantizer::train_q1 configures it for IndexIVFFlat (IndexIVF.cpp:43-49, 76-88; Clustering.cpp:
// 330-560): niter = 10, nredo = 1, seed 1234, <= 256 points per centroid (rand_perm subsample beyond that), initial centroids = the first
// nlist points of rand_perm(seed + 1), spherical (renormalised centroids) for inner product / cosine, compute_centroids in single-precision
// sums in data order, split_clusters with its own RandomGenerator(1234) per call.  The assignment of every iteration is the exact k = 1
// search of the coarse quantiser on the device — FAISS's own search with its BLAS shortcut off (distance_compute_blas_threshold above nx);
// with the shortcut on, the reference's result depends on the BLAS library it finds at run time.  tests/test_gpu_ivf.py holds the
// centroids to the bits of the vendored FAISS built in place (oracle/_ref/libref_ivf.so).
void GpuIvfFlat::Train(int seed) {
	if (count_ < nlist_) throw std::runtime_error("Number of training points should be at least as large as number of clusters");
	// IndexIVFFlat::train (IndexIVFFlat.cpp:54-75): cosine trains on x * (1 / |x|)
	size_t ns = count_;
	std::vector<int> pick;
	if (ns > nlist_ * kMaxPointsPerCentroid) {   // subsample_training_set (Clustering.cpp:83-137)
		pick = faissRandPerm(ns, seed);
		ns = nlist_ * kMaxPointsPerCentroid;
		pick.resize(ns);
	}
	std::vector<float> pts(ns * dim_);
	for (size_t i = 0; i < ns; ++i) {
		const size_t srcRow = pick.empty() ? i : size_t(pick[i]);
		const float* src = rows_.data() + srcRow * dim_;
		float* dst = pts.data() + i * dim_;
		if (metric_ == VectorMetric::Cosine) {
			const float k = invNorms_[srcRow];
			for (size_t j = 0; j < dim_; ++j) dst[j] = src[j] * k;
		} else {
			std::memcpy(dst, src, dim_ * sizeof(float));
		}
	}
	const bool spherical = metric_ != VectorMetric::L2;   // IndexIVF.cpp:179-182 (cosine is METRIC_INNER_PRODUCT + is_cosine)
	// fvec_renorm_L2 (utils/distances.cpp:77-86) over fvec_norm_L2sqr (utils/distances_simd.cpp:216-235): in the AVX-512 build (the SIMD level
	// every parity claim here is pinned to) the sum of squares is ONE sequential chain of fused multiply-adds
	auto renorm = [&](float* c) {
		float nr = 0.f;
		for (size_t j = 0; j < dim_; ++j) nr = std::fmaf(c[j], c[j], nr);
		if (nr > 0) {
			const float inv = float(1.0 / double(std::sqrt(nr)));
			for (size_t j = 0; j < dim_; ++j) c[j] *= inv;
		}
	};
	centroids_.assign(nlist_ * dim_, 0.f);
	if (ns == nlist_) {   // corner case: the training set is the centroid table (no post-processing, Clustering.cpp:365-383)
		std::memcpy(centroids_.data(), pts.data(), nlist_ * dim_ * sizeof(float));
	} else {
		const std::vector<int> perm = faissRandPerm(ns, int64_t(seed) + 1);
		for (size_t c = 0; c < nlist_; ++c) {
			std::memcpy(centroids_.data() + c * dim_, pts.data() + size_t(perm[c]) * dim_, dim_ * sizeof(float));
		}
		if (spherical) {
			for (size_t c = 0; c < nlist_; ++c) renorm(centroids_.data() + c * dim_);
		}
		std::vector<uint32_t> a;
		std::vector<float> hassign(nlist_);
		for (int it = 0; it < kIterations; ++it) {
			uploadCentroids();
			assign(pts.data(), ns, a);
			// compute_centroids (Clustering.cpp:153-230)
			std::fill(centroids_.begin(), centroids_.end(), 0.f);
			std::fill(hassign.begin(), hassign.end(), 0.f);
			for (size_t i = 0; i < ns; ++i) {
				const float* x = pts.data() + i * dim_;
				float* c = centroids_.data() + size_t(a[i]) * dim_;
				hassign[a[i]] += 1.0f;
				for (size_t j = 0; j < dim_; ++j) c[j] += x[j];
			}
			for (size_t ci = 0; ci < nlist_; ++ci) {
				if (hassign[ci] == 0) continue;
				const float norm = 1 / hassign[ci];
				float* c = centroids_.data() + ci * dim_;
				for (size_t j = 0; j < dim_; ++j) c[j] *= norm;
			}
			// split_clusters (Clustering.cpp:243-290): an empty cluster takes half of a big one, both nudged apart by +-1/1024
			FaissRng rng(1234);
			for (size_t ci = 0; ci < nlist_; ++ci) {
				if (hassign[ci] != 0) continue;
				size_t cj = 0;
				for (;; cj = (cj + 1) % nlist_) {
					const float p = float((double(hassign[cj]) - 1.0) / double(float(ns - nlist_)));
					const float r = rng.rand_float();
					if (r < p) break;
				}
				float* ni = centroids_.data() + ci * dim_;
				float* nj = centroids_.data() + cj * dim_;
				std::memcpy(ni, nj, dim_ * sizeof(float));
				for (size_t j = 0; j < dim_; ++j) {
					if (j % 2 == 0) {
						ni[j] = float(double(ni[j]) * (1 + double(kSplitEps)));
						nj[j] = float(double(nj[j]) * (1 - double(kSplitEps)));
					} else {
						ni[j] = float(double(ni[j]) * (1 - double(kSplitEps)));
						nj[j] = float(double(nj[j]) * (1 + double(kSplitEps)));
					}
				}
				hassign[ci] = hassign[cj] / 2;
				hassign[cj] -= hassign[ci];
			}
			if (spherical) {   // post_process_centroids
				for (size_t c = 0; c < nlist_; ++c) renorm(centroids_.data() + c * dim_);
			}
		}
	}
	uploadCentroids();
	trained_ = true;
	listsDirty_ = true;
	// add_with_ids of everything that sat in the flat phase (ivf_index.cc:101-103)
	for (auto& l : lists_) l.clear();
	for (auto& l : scan_) l.clear();
	scanPos_.assign(count_, 0u);
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
			scanAppend(chunkAssign[i], uint32_t(first + i));
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
			scanAppend(a[i], uint32_t(count_ + i));
			listsDirty_ = true;
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
				listRename(listOf_[last], last, row);
				listOf_[row] = listOf_[last];
			}
			if (rxgpu_index_move_row(dev_, last, row) != RXGPU_OK) throwDevice("GpuIvfFlat: row move failed");
		}
		--count_;
		rows_.resize(count_ * dim_);
		ids_.resize(count_);
		if (metric_ == VectorMetric::Cosine) invNorms_.resize(count_);
		if (trained_) {
			listOf_.resize(count_);
			scanPos_.resize(count_);
		}
		if (rxgpu_index_truncate(dev_, count_) != RXGPU_OK) throwDevice("GpuIvfFlat: truncate failed");
		++removed;
	}
	return removed;
}

void GpuIvfFlat::syncLists() const {
	std::lock_guard<std::mutex> lk(listsMtx_);
	if (!listsDirty_) return;
	std::vector<uint64_t> off(nlist_ + 1, 0);
	for (size_t l = 0; l < nlist_; ++l) off[l + 1] = off[l] + lists_[l].size();
	std::vector<uint32_t> rows;
	rows.reserve(off[nlist_]);
	for (const auto& l : lists_) rows.insert(rows.end(), l.begin(), l.end());
	if (rxgpu_index_set_lists(dev_, uint32_t(nlist_), off.data(), rows.data()) != RXGPU_OK) throwDevice("GpuIvfFlat: list upload failed");
	listsDirty_ = false;
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
	// one candidate more than asked for: equal distances at the k-th place are decided the way faiss's scanner decides them (below)
	const size_t kk = trained_ ? k + 1 : k;
	std::vector<float> dist(kk);
	std::vector<uint32_t> row(kk);
	uint32_t cnt = 0;
	if (!trained_) {   // the flat phase: IndexFlat::search
		if (rxgpu_search_knn(dev_, q.data(), 1, uint32_t(k), dist.data(), row.data(), &cnt) != RXGPU_OK) throwDevice("GpuIvfFlat::Search");
	} else {
		// everything on the device, whatever nprobe: no row list through the host (up to 128 probed lists not even their ids)
		syncLists();
		if (rxgpu_search_knn_lists(dev_, devCentroids_, q.data(), uint32_t(std::min<size_t>(nprobe, nlist_)), uint32_t(kk), dist.data(), row.data(), &cnt,
								   nullptr) != RXGPU_OK) {
			throwDevice("GpuIvfFlat::Search");
		}
	}
	if (trained_ && cnt > k && dist[k] == dist[k - 1]) {   // more candidates at the k-th distance than places: the scanner's order decides
		replayTies(q.data(), k, nprobe, dist[k - 1], distances, labels);
		return;
	}
	const uint32_t n = uint32_t(std::min<size_t>(cnt, k));
	for (uint32_t i = 0; i < n; ++i) {
		distances[i] = toFaiss(dist[i]);
		labels[i] = ids_[row[i]];
	}
	if (trained_) orderTies(n, distances, labels);
}

// n queries (IndexIVFFlat::search's n; ivf_index.cc:355-372 passes 1, SelectKnn callers with several keys more): the queries are independent
// launch trains, so a few host threads drive them side by side — every Search checks out its own device scratch and stream.
void GpuIvfFlat::SearchBatch(size_t n, const float* x, size_t k, size_t nprobe, float* distances, idx_t* labels) const {
	if (n == 0) return;
	const size_t workers = std::min<size_t>(n, 4);
	if (workers <= 1) {
		Search(x, k, nprobe, distances, labels);
		return;
	}
	if (trained_) syncLists();   // once, before the threads (it mutates the cached device lists)
	std::atomic<size_t> next{0};
	std::exception_ptr failure;
	std::mutex failMtx;
	std::vector<std::thread> pool;
	for (size_t w = 0; w < workers; ++w) {
		pool.emplace_back([&] {
			try {
				for (size_t q = next.fetch_add(1); q < n; q = next.fetch_add(1)) Search(x + q * dim_, k, nprobe, distances + q * k, labels + q * k);
			} catch (...) {
				std::lock_guard<std::mutex> lk(failMtx);
				if (!failure) failure = std::current_exception();
			}
		});
	}
	for (auto& t : pool) t.join();
	if (failure) std::rethrow_exception(failure);
}

// Equal distances inside a result: heap_reorder (utils/Heap.h) pops the heap top into the last free place.  The top of the CMax heap (L2)
// is the maximum by (distance, id) => ties end up by id ascending; the top of the CMin heap (inner product / cosine) is the minimum by
// (similarity, id) => ties end up by id DESCENDING.
void GpuIvfFlat::orderTies(size_t n, float* distances, idx_t* labels) const {
	for (size_t a = 0; a < n;) {
		size_t b = a + 1;
		while (b < n && distances[b] == distances[a]) ++b;
		if (b - a > 1) {
			if (metric_ == VectorMetric::L2) {
				std::sort(labels + a, labels + b);
			} else {
				std::sort(labels + a, labels + b, std::greater<idx_t>());
			}
		}
		a = b;
	}
}

// IndexIVFFlat's scanner (IndexIVFFlat.cpp, scan_codes) walks the probed lists in coarse order and every list in storage order and takes
// a vector only if it is STRICTLY better than the heap top; the top is the worst entry by (distance, id) (L2) resp. by (similarity, id)
// (inner product / cosine: among equal similarities the SMALLEST id is evicted first) — heap_replace_top compares with cmp2.  With ties
// at the k-th place the result therefore depends on the scan order.  Only the candidates at or below the k-th distance can be in the
// result or influence it, so: all of them from the device (range search over the probed rows, bound = the next float after `worst`),
// put into scan order, and the scanner's rule replayed over them.
void GpuIvfFlat::replayTies(const float* q, size_t k, size_t nprobe, float worst, float* distances, idx_t* labels) const {
	std::vector<uint32_t> probe;
	coarse(q, nprobe, probe);
	std::vector<const std::vector<uint32_t>*> runs;
	for (uint32_t l : probe) runs.push_back(&lists_[l]);
	const std::vector<uint32_t> rows = SortedUnion(runs, count_);
	const float bound = std::nextafter(worst, std::numeric_limits<float>::infinity());
	std::vector<float> dist(k + 64);
	std::vector<uint32_t> row(k + 64);
	uint64_t total = 0;
	for (;;) {
		const int rc = rxgpu_search_range_subset(dev_, q, bound, 0, rows.data(), rows.size(), dist.data(), row.data(), dist.size(), &total);
		if (rc == RXGPU_OK) break;
		if (rc != RXGPU_ERR_OVERFLOW) throwDevice("GpuIvfFlat::Search (ties)");
		dist.resize(total);
		row.resize(total);
	}
	// scan order: (position of the row's list in the probe order, position inside the list)
	std::vector<uint32_t> probeRank(nlist_, 0xFFFFFFFFu);
	for (size_t i = 0; i < probe.size(); ++i) probeRank[probe[i]] = uint32_t(i);
	struct Cand {
		uint64_t order;
		float dist;
		idx_t id;
	};
	std::vector<Cand> cands(total);
	for (uint64_t i = 0; i < total; ++i) {
		const uint32_t r = row[i];
		cands[i] = {(uint64_t(probeRank[listOf_[r]]) << 32) | scanPos_[r], dist[i], ids_[r]};
	}
	std::sort(cands.begin(), cands.end(), [](const Cand& a, const Cand& b) { return a.order < b.order; });
	const bool l2 = metric_ == VectorMetric::L2;
	// "worse" = closer to the heap top: larger internal distance; among equals the larger id (L2) resp. the smaller id (similarities)
	auto worse = [l2](const Cand& a, const Cand& b) { return a.dist != b.dist ? a.dist > b.dist : (l2 ? a.id > b.id : a.id < b.id); };
	auto heapLess = [&](const Cand& a, const Cand& b) { return worse(b, a); };   // std heap: the "largest" (= worst) on top
	std::vector<Cand> heap;
	heap.reserve(k);
	for (const Cand& c : cands) {
		if (heap.size() < k) {
			heap.push_back(c);
			std::push_heap(heap.begin(), heap.end(), heapLess);
		} else if (heap.front().dist > c.dist) {
			std::pop_heap(heap.begin(), heap.end(), heapLess);
			heap.back() = c;
			std::push_heap(heap.begin(), heap.end(), heapLess);
		}
	}
	std::sort(heap.begin(), heap.end(), [&](const Cand& a, const Cand& b) { return worse(b, a); });   // best first; ties as heap_reorder leaves them
	for (size_t i = 0; i < heap.size(); ++i) {
		distances[i] = toFaiss(heap[i].dist);
		labels[i] = heap[i].id;
	}
}

void GpuIvfFlat::RangeSearch(const float* x, float radius, size_t nprobe, std::vector<float>& distances, std::vector<idx_t>& labels) const {
	distances.clear();
	labels.clear();
	if (count_ == 0) return;
	std::vector<float> q;
	prepareQuery(x, q);
	const float internal = metric_ == VectorMetric::L2 ? radius : -radius;   // similarity > radius <=> -similarity < -radius
	if (trained_) syncLists();
	std::vector<float> dist(1024);
	std::vector<uint32_t> row(1024);
	uint64_t total = 0;
	for (;;) {   // trained: the probed lists are found, united and scanned on the device (rxgpu_search_range_lists)
		const int rc = trained_ ? rxgpu_search_range_lists(dev_, devCentroids_, q.data(), uint32_t(std::min<size_t>(std::max<size_t>(nprobe, 1), nlist_)), internal,
														   0, dist.data(), row.data(), dist.size(), &total, nullptr)
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
