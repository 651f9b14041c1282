#include "gpu_bruteforce_map.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <queue>
#include <stdexcept>
#include <string>

#include "rxgpu.h"

namespace rxgpu::host {

namespace {
[[noreturn]] void throwDevice(const char* what) { throw std::runtime_error(std::string(what) + ": " + rxgpu_last_error()); }
}  // namespace

// tools/normalize.cc:10-23: k = 1/sqrt(sum x^2), 1.0 for zero and already-unit (|1 - sum| <= 1e-5) vectors.
// The reference compiles this under an "imprecise" pragma (compiler-defined summation order); in the pinned
// oracle build it is a sequential fma chain, which is what this restates (tests/test_host_map.py re-asserts it).
float CalculateL2Module(const float* x, int32_t d) noexcept {
	float sq = 0.0f;
	for (int32_t i = 0; i < d; ++i) sq = std::fmaf(x[i], x[i], sq);
	float k = 1.0f;
	if (sq > 0.0f && std::fabs(1.0f - sq) > 0.00001f) k = float(1.0 / double(std::sqrt(sq)));
	return k;
}

float NormalizeCopyVector(const float* x, int32_t d, float* out) noexcept {
	const float k = CalculateL2Module(x, d);
	for (int32_t i = 0; i < d; ++i) out[i] = x[i] * k;
	return k;
}

GpuBruteforceMap::GpuBruteforceMap(VectorMetric metric, size_t dim, size_t maxElements, int device)
	: GpuBruteforceMap(metric, dim, maxElements, std::vector<int>{device}) {}

void GpuBruteforceMap::createDeviceIndex() {
	if (dev_) rxgpu_index_destroy(dev_);
	dev_ = nullptr;
	const int rc = devices_.size() > 1 ? rxgpu_index_create_sharded(int(metric_), uint32_t(dim_), std::max<size_t>(maxElements_, 1), uint32_t(devices_.size()),
																	devices_.data(), &dev_)
									   : rxgpu_index_create(int(metric_), uint32_t(dim_), maxElements_, device_, &dev_);
	if (rc != RXGPU_OK) throwDevice("GpuBruteforceMap: device index creation failed");
}

GpuBruteforceMap::GpuBruteforceMap(VectorMetric metric, size_t dim, size_t maxElements, std::vector<int> devices)
	: metric_(metric), dim_(dim), device_(devices.empty() ? 0 : devices[0]), devices_(std::move(devices)), maxElements_(maxElements) {
	if (devices_.empty()) throw std::logic_error("GpuBruteforceMap: empty device list");
	if (dim_ == 0 || dim_ > 65535) throw std::logic_error("GpuBruteforceMap: dimension must be in [1, 65535]");
	try {
		rows_.resize(maxElements_ * dim_);
		labels_.resize(maxElements_);
		if (metric_ == VectorMetric::Cosine) invNorms_.resize(maxElements_);
	} catch (const std::bad_alloc&) {
		throw std::runtime_error("Not enough memory: BruteforceSearch failed to allocate data");
	}
	createDeviceIndex();
}

GpuBruteforceMap::GpuBruteforceMap(const GpuBruteforceMap& other, size_t newMaxElements)
	: metric_(other.metric_),
	  dim_(other.dim_),
	  device_(other.device_),
	  devices_(other.devices_),
	  maxElements_(std::max(other.maxElements_, newMaxElements)),
	  curElementCount_(other.curElementCount_),
	  dictExternalToInternal_(other.dictExternalToInternal_) {
	try {
		rows_.resize(maxElements_ * dim_);
		labels_.resize(maxElements_);
		if (metric_ == VectorMetric::Cosine) invNorms_.resize(maxElements_);
	} catch (const std::bad_alloc&) {
		throw std::runtime_error("Not enough memory: BruteforceSearch failed to allocate data");
	}
	std::memcpy(rows_.data(), other.rows_.data(), curElementCount_ * dim_ * sizeof(float));
	std::memcpy(labels_.data(), other.labels_.data(), curElementCount_ * sizeof(labeltype));
	if (!invNorms_.empty()) std::memcpy(invNorms_.data(), other.invNorms_.data(), curElementCount_ * sizeof(float));
	labelsIdentity_ = other.labelsIdentity_;
	createDeviceIndex();
	dirtyAll_ = true;
	needSync_ = true;
}

GpuBruteforceMap::~GpuBruteforceMap() {
	if (dev_) rxgpu_index_destroy(dev_);
}

size_t GpuBruteforceMap::AllocatedMemSize() const noexcept {
	return dictExternalToInternal_.size() * (sizeof(labeltype) + sizeof(size_t) + 2 * sizeof(void*)) +
		   (MaxElements() - CurrentElementCount()) * ElementSize() + sizeof(GpuBruteforceMap);
}

size_t GpuBruteforceMap::DeviceMemSize() const noexcept { return dev_ ? rxgpu_index_device_bytes(dev_) : 0; }

const float* GpuBruteforceMap::FloatPtrByExternalLabel(labeltype label) const {
	auto it = dictExternalToInternal_.find(label);
	if (it == dictExternalToInternal_.end()) throw std::runtime_error("Label not found");
	return rows_.data() + it->second * dim_;
}

void GpuBruteforceMap::markDirty(size_t idx) {
	needSync_ = true;
	if (dirtyAll_) return;
	if (dirtyRows_.size() > 4096 && dirtyRows_.size() * 4 > curElementCount_) {   // cheaper to re-send everything
		dirtyAll_ = true;
		dirtyRows_.clear();
		return;
	}
	dirtyRows_.push_back(uint32_t(idx));
}

// bruteforce.cc:44-64
void GpuBruteforceMap::AddPointNoLock(ConstFloatVectorView vect, FloatVectorId id) {
	size_t idx;
	const labeltype label = id.AsNumber();
	auto it = dictExternalToInternal_.find(label);
	if (it != dictExternalToInternal_.end()) {
		idx = it->second;
	} else {
		if (curElementCount_ >= maxElements_) throw std::runtime_error("The number of elements exceeds the specified limit\n");
		idx = curElementCount_;
		dictExternalToInternal_[label] = idx;
		curElementCount_++;
	}
	if (metric_ == VectorMetric::Cosine) invNorms_[idx] = CalculateL2Module(vect.Data(), int32_t(dim_));   // AddNorm
	std::memcpy(rows_.data() + idx * dim_, vect.Data(), dim_ * sizeof(float));
	labels_[idx] = label;
	if (label != (labeltype(idx) << 32)) labelsIdentity_ = false;
	markDirty(idx);
}

void GpuBruteforceMap::AddPointConcurrent(ConstFloatVectorView, FloatVectorId) {
	throw std::logic_error("This brute force index does not support concurrent insertions");
}

// bruteforce.cc:70-86: swap-with-last keeps the scan order the reference's tie rule depends on
void GpuBruteforceMap::RemovePoint(labeltype curExternal) {
	auto found = dictExternalToInternal_.find(curExternal);
	if (found == dictExternalToInternal_.end()) return;
	const size_t cur = found->second;
	dictExternalToInternal_.erase(found);
	const size_t last = curElementCount_ - 1;
	if (cur != last) {
		dictExternalToInternal_[labels_[last]] = cur;
		std::memcpy(rows_.data() + cur * dim_, rows_.data() + last * dim_, dim_ * sizeof(float));
		labels_[cur] = labels_[last];
		labelsIdentity_ = false;
		if (!invNorms_.empty()) invNorms_[cur] = invNorms_[last];   // MoveNorm
		markDirty(cur);
	}
	curElementCount_--;
	needSync_ = true;
}

// bruteforce.cc:88-101
void GpuBruteforceMap::ResizeIndex(size_t newMaxElements) {
	if (newMaxElements < curElementCount_) {
		throw std::runtime_error("Cannot resize, max element is less than the current number of elements");
	}
	try {
		rows_.resize(newMaxElements * dim_);
		labels_.resize(newMaxElements);
		if (metric_ == VectorMetric::Cosine) invNorms_.resize(newMaxElements);
	} catch (const std::bad_alloc&) {
		throw std::runtime_error("Not enough memory: resizeIndex failed to allocate data");
	}
	maxElements_ = newMaxElements;
	needSync_ = true;
}

// Lazy host -> HBM synchronisation; writers hold the namespace's exclusive lock, readers may race here => mutex.
void GpuBruteforceMap::syncDevice() const {
	std::lock_guard<std::mutex> lk(syncMtx_);
	if (!needSync_) return;
	if (devices_.size() > 1 && rxgpu_index_capacity(dev_) != std::max<size_t>(maxElements_, 1)) {
		// the shard boundaries follow the capacity: a resized sharded mirror is rebuilt from the host master copy
		const_cast<GpuBruteforceMap*>(this)->createDeviceIndex();
		dirtyAll_ = true;
	} else if (devices_.size() == 1 && rxgpu_index_capacity(dev_) != maxElements_) {
		// keep the live prefix on the device when growing; shrinking below the device count needs a truncate first
		if (rxgpu_index_count(dev_) > curElementCount_) {
			if (rxgpu_index_truncate(dev_, curElementCount_) != RXGPU_OK) throwDevice("truncate");
		}
		if (rxgpu_index_reserve(dev_, maxElements_) != RXGPU_OK) throwDevice("Not enough memory: resizeIndex failed to allocate data");
	}
	const float* norms = invNorms_.empty() ? nullptr : invNorms_.data();
	// the row-id table of the hybrid fusion travels with the rows once a label is not (row << 32) any more (single-device mirror only)
	const bool withIds = !labelsIdentity_ && devices_.size() == 1;
	std::vector<int32_t> ids;
	auto uploadIds = [&](size_t first, size_t n) {
		ids.resize(n);
		for (size_t i = 0; i < n; ++i) ids[i] = int32_t(labels_[first + i] >> 32);
		if (rxgpu_index_upload_row_ids(dev_, first, n, ids.data()) != RXGPU_OK) throwDevice("row id upload failed");
	};
	auto upload = [&](size_t first, size_t n) {
		if (n == 0) return;
		if (rxgpu_index_upload_rows(dev_, first, n, rows_.data() + first * dim_, norms ? norms + first : nullptr) != RXGPU_OK) {
			throwDevice("row upload failed");
		}
		if (withIds && rowIdsOnDevice_) uploadIds(first, n);
	};
	if (withIds && !rowIdsOnDevice_) {
		if (curElementCount_) uploadIds(0, curElementCount_);
		rowIdsOnDevice_ = true;
	}
	if (dirtyAll_) {
		upload(0, curElementCount_);
	} else if (!dirtyRows_.empty()) {
		std::sort(dirtyRows_.begin(), dirtyRows_.end());
		dirtyRows_.erase(std::unique(dirtyRows_.begin(), dirtyRows_.end()), dirtyRows_.end());
		size_t i = 0;
		while (i < dirtyRows_.size()) {
			size_t j = i + 1;
			while (j < dirtyRows_.size() && dirtyRows_[j] == dirtyRows_[j - 1] + 1) ++j;
			const size_t first = dirtyRows_[i];
			if (first < curElementCount_) upload(first, std::min<size_t>(dirtyRows_[j - 1] + 1, curElementCount_) - first);
			i = j;
		}
	}
	if (rxgpu_index_truncate(dev_, curElementCount_) != RXGPU_OK) throwDevice("truncate");
	dirtyRows_.clear();
	dirtyAll_ = false;
	needSync_ = false;
}

GpuBruteforceMap::ResidentKnn GpuBruteforceMap::SearchKnnResident(const float* queryData, size_t k) const {
	if (devices_.size() > 1) throw std::logic_error("SearchKnnResident: not available on a sharded mirror");
	if (curElementCount_ == 0 || k == 0) return ResidentKnn{};
	syncDevice();
	ResidentKnn r;
	void *dd = nullptr, *dr = nullptr, *dc = nullptr;
	if (rxgpu_search_knn_resident(dev_, queryData, uint32_t(std::min<size_t>(k + 1, curElementCount_)), &dd, &dr, &dc, &r.stream, &r.entries) != RXGPU_OK) {
		throwDevice("SearchKnnResident");
	}
	r.dDist = dd;
	r.dRow = dr;
	r.dCount = dc;
	r.dRowIds = labelsIdentity_ ? nullptr : rxgpu_index_row_ids_device(dev_);
	return r;
}

// bruteforce.cc:103-127.  The kernels return the exact top-(k+1) under the (dist,row) order; the reference keeps a
// (dist,label) max-heap with STRICT admission (`dist < worst`, :121) over a scan in row order.  Both agree unless an
// exact distance tie straddles the k-th boundary; then the sequential rule is replayed over every row with
// dist <= d_k (fetched with an inclusive range scan), which is all the rule can ever look at:
//   * a row with dist < d_k is always admitted and never evicted;
//   * while fewer than k rows with dist <= d_k have been seen the worst is > d_k, so such rows are admitted;
//   * afterwards ties are rejected and every better row evicts the tie with the LARGEST LABEL.
struct GpuBruteforceMap::PendingQuery {
	const float* query;
	uint32_t kk;
	float* dist;
	uint32_t* row;
	uint32_t* count;
	bool done = false;
	int rc = 0;
	std::string error;
};

// One device round trip for every query of the batch; each caller gets the first kk_i entries of its exact top-kk_max list.
void GpuBruteforceMap::runBatch(std::vector<PendingQuery*>& batch) const {
	if (batch.size() == 1) {
		PendingQuery& p = *batch[0];
		p.rc = rxgpu_search_knn(dev_, p.query, 1, p.kk, p.dist, p.row, p.count);
		if (p.rc != RXGPU_OK) p.error = rxgpu_last_error();
		return;
	}
	uint32_t kkMax = 0;
	for (const PendingQuery* p : batch) kkMax = std::max(kkMax, p->kk);
	const size_t nq = batch.size();
	std::vector<float> queries(nq * dim_), dist(nq * kkMax);
	std::vector<uint32_t> row(nq * kkMax), count(nq);
	for (size_t i = 0; i < nq; ++i) std::copy(batch[i]->query, batch[i]->query + dim_, queries.begin() + i * dim_);
	const int rc = rxgpu_search_knn(dev_, queries.data(), uint32_t(nq), kkMax, dist.data(), row.data(), count.data());
	const std::string error = rc != RXGPU_OK ? rxgpu_last_error() : "";
	for (size_t i = 0; i < nq; ++i) {
		PendingQuery& p = *batch[i];
		p.rc = rc;
		p.error = error;
		if (rc != RXGPU_OK) continue;
		const uint32_t n = std::min(p.kk, count[i]);
		std::copy(dist.begin() + i * kkMax, dist.begin() + i * kkMax + n, p.dist);
		std::copy(row.begin() + i * kkMax, row.begin() + i * kkMax + n, p.row);
		*p.count = n;
	}
}

void GpuBruteforceMap::fetchTopK(const float* query, uint32_t kk, float* dist, uint32_t* row, uint32_t* count) const {
	PendingQuery p{query, kk, dist, row, count, false, 0, {}};
	if (!coalesce_) {
		std::vector<PendingQuery*> one{&p};
		runBatch(one);
	} else {
		std::unique_lock<std::mutex> lk(coMtx_);
		coQueue_.push_back(&p);
		while (!p.done) {
			if (coLeader_) {
				coCv_.wait(lk);
				continue;
			}
			coLeader_ = true;   // the device is idle: run everything that is queued right now (our own query included unless > 256 are ahead)
			std::vector<PendingQuery*> batch;
			while (!coQueue_.empty() && batch.size() < 256) {
				batch.push_back(coQueue_.front());
				coQueue_.pop_front();
			}
			lk.unlock();
			try {
				runBatch(batch);
			} catch (const std::exception& e) {   // e.g. bad_alloc while staging: every caller of the batch gets the error, leadership is released
				for (PendingQuery* q : batch) {
					q->rc = RXGPU_ERR_NOMEM;
					q->error = e.what();
				}
			}
			lk.lock();
			for (PendingQuery* q : batch) q->done = true;
			++coBatches_;
			coQueries_ += batch.size();
			coLeader_ = false;
			coCv_.notify_all();
		}
	}
	if (p.rc != RXGPU_OK) throw std::runtime_error("SearchKnn: " + p.error);
}

SearchResultQueue GpuBruteforceMap::SearchKnn(const float* queryData, std::optional<float>, size_t k, size_t) const {
	SearchResultQueue result;
	if (curElementCount_ == 0 || k == 0) return result;
	syncDevice();
	k = std::min(k, curElementCount_);
	const uint32_t kk = uint32_t(std::min(k + 1, curElementCount_));
	std::vector<float> dist(kk);
	std::vector<uint32_t> row(kk);
	uint32_t count = 0;
	fetchTopK(queryData, kk, dist.data(), row.data(), &count);
	ReserveQueue(result, k);
	const bool tieAcross = count > k && !(dist[k - 1] < dist[k]);
	if (!tieAcross) {
		for (size_t i = 0; i < std::min<size_t>(k, count); ++i) result.emplace(dist[i], labels_[row[i]]);
		return result;
	}
	return replayTies(queryData, k, dist[k - 1], nullptr);
}

// The tie replay of SearchKnn / SearchKnnFiltered: every row with dist <= dk (restricted to `allowedRows` when given, sorted), walked in
// scan order through the reference's admission rule.
SearchResultQueue GpuBruteforceMap::replayTies(const float* queryData, size_t k, float dk, const std::vector<uint32_t>* allowedRows) const {
	SearchResultQueue result;
	ReserveQueue(result, k);
	{
		std::lock_guard<std::mutex> lk(syncMtx_);
		++tieReplays_;
	}
	std::vector<float> rd(std::max<size_t>(4 * k, 256));
	std::vector<uint32_t> rr(rd.size());
	uint64_t total = 0;
	for (;;) {
		const int rc = rxgpu_search_range(dev_, queryData, dk, /*inclusive*/ 1, rd.data(), rr.data(), rd.size(), &total);
		if (rc == RXGPU_OK) break;
		if (rc != RXGPU_ERR_OVERFLOW) throwDevice("SearchKnn (tie replay)");
		rd.resize(total);
		rr.resize(total);
	}
	std::vector<uint32_t> order;
	order.reserve(total);
	for (uint64_t i = 0; i < total; ++i) {
		if (!allowedRows || std::binary_search(allowedRows->begin(), allowedRows->end(), rr[i])) order.push_back(uint32_t(i));
	}
	std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return rr[a] < rr[b]; });   // scan order
	std::vector<std::pair<float, labeltype>> better;   // dist < dk: permanent members
	std::priority_queue<labeltype> ties;               // dist == dk members, largest label on top
	better.reserve(k);
	for (uint32_t oi : order) {
		const float d = rd[oi];
		const labeltype lab = labels_[rr[oi]];
		const bool isTie = !(d < dk);
		if (better.size() + ties.size() < k) {
			if (isTie) {
				ties.push(lab);
			} else {
				better.emplace_back(d, lab);
			}
		} else if (!isTie) {
			ties.pop();   // heap full => worst distance is dk => the evicted pair is the tie with the largest label
			better.emplace_back(d, lab);
		}
	}
	for (auto& p : better) result.emplace(p.first, p.second);
	while (!ties.empty()) {
		result.emplace(dk, ties.top());
		ties.pop();
	}
	return result;
}

// Pre-filtered search — the caller side of `WHERE cond AND KNN(...)` (the reference post-filters the k hits on the host,
// nsselecter.cc:841-875).  Result == BruteforceSearch::SearchKnn (bruteforce.cc:103-127) over an index that holds only the points whose
// labels are in `allowed`, inserted in the same relative order.  Unknown labels are ignored.  The scan reads the allowed rows only.
SearchResultQueue GpuBruteforceMap::SearchKnnFiltered(const float* queryData, std::optional<float>, size_t k, const labeltype* allowed,
													   size_t nAllowed) const {
	SearchResultQueue result;
	if (curElementCount_ == 0 || k == 0 || nAllowed == 0) return result;
	syncDevice();
	std::vector<uint32_t> rows;
	rows.reserve(nAllowed);
	for (size_t i = 0; i < nAllowed; ++i) {
		const auto it = dictExternalToInternal_.find(allowed[i]);
		if (it != dictExternalToInternal_.end()) rows.push_back(uint32_t(it->second));
	}
	std::sort(rows.begin(), rows.end());
	rows.erase(std::unique(rows.begin(), rows.end()), rows.end());
	if (rows.empty()) return result;
	k = std::min(k, rows.size());
	const uint32_t kk = uint32_t(std::min(k + 1, rows.size()));
	std::vector<float> dist(kk);
	std::vector<uint32_t> row(kk);
	uint32_t count = 0;
	int rc;
	if (devices_.size() == 1 && rows.size() * 32 >= curElementCount_) {   // dense filter: count / 8 bytes on the wire instead of 4 per allowed row
		std::vector<uint32_t> words((curElementCount_ + 31) / 32, 0u);
		for (uint32_t r : rows) words[r >> 5] |= 1u << (r & 31);
		rc = rxgpu_search_knn_bitmap(dev_, queryData, 1, kk, words.data(), words.size(), dist.data(), row.data(), &count, nullptr);
	} else {
		rc = rxgpu_search_knn_subset(dev_, queryData, 1, kk, rows.data(), rows.size(), dist.data(), row.data(), &count);
	}
	if (rc != RXGPU_OK) throwDevice("SearchKnnFiltered");
	const bool tieAcross = count > k && !(dist[k - 1] < dist[k]);
	if (tieAcross) return replayTies(queryData, k, dist[k - 1], &rows);
	ReserveQueue(result, k);
	for (size_t i = 0; i < std::min<size_t>(k, count); ++i) result.emplace(dist[i], labels_[row[i]]);
	return result;
}

// bruteforce.cc:129-143 (radius already negated by the caller for IP / cosine, hnsw_index.cc:185)
SearchResultQueue GpuBruteforceMap::SearchRange(const float* queryData, std::optional<float>, float radius, size_t) const {
	SearchResultQueue result;
	if (curElementCount_ == 0) return result;
	syncDevice();
	std::vector<float> rd(1024);
	std::vector<uint32_t> rr(1024);
	uint64_t total = 0;
	for (;;) {
		const int rc = rxgpu_search_range(dev_, queryData, radius, /*inclusive*/ 0, rd.data(), rr.data(), rd.size(), &total);
		if (rc == RXGPU_OK) break;
		if (rc != RXGPU_ERR_OVERFLOW) throwDevice("SearchRange");
		rd.resize(total);
		rr.resize(total);
	}
	ReserveQueue(result, total);
	for (uint64_t i = 0; i < total; ++i) result.emplace(rd[i], labels_[rr[i]]);
	return result;
}

}  // namespace rxgpu::host
