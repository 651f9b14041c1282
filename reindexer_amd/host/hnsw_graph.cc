// The graph builder is the one CPU-bound loop of the host library (12 minutes for 10M x 768): this unit is compiled at -O3 whatever the
// level of the build (measured, 20 000 x 768 cosine, one thread: 1596 -> 1928 inserts/s; the reference engine itself: 1939).  No float
// arithmetic is re-associated by that (no -ffast-math; -ffp-contract=off stays): the link-for-link tests against the engine hold.
#if defined(__GNUC__) && !defined(__clang__)
#pragma GCC optimize("O3")
#endif
#include "hnsw_graph.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <stdexcept>
#include <string>
#include <thread>

#include "distance_cpu.h"
#include "gpu_bruteforce_map.h"   // CalculateL2Module
#include "numa_policy.h"

namespace rxgpu::host {

namespace {
constexpr tableint kNoEntry = std::numeric_limits<tableint>::max();

// One-byte test-and-test-and-set lock per element (a std::mutex per element would cost 40 B x 10M); waits are short except behind an
// element that is still being inserted, hence the yield.
class NodeLock {
public:
	explicit NodeLock(std::atomic<uint8_t>& f) noexcept : f_(f) {
		unsigned spins = 0;
		while (f_.exchange(1, std::memory_order_acquire)) {
			while (f_.load(std::memory_order_relaxed)) {
				if (++spins > 128) {
					std::this_thread::yield();
				} else {
					__builtin_ia32_pause();
				}
			}
		}
	}
	~NodeLock() { f_.store(0, std::memory_order_release); }
	NodeLock(const NodeLock&) = delete;
	NodeLock& operator=(const NodeLock&) = delete;

private:
	std::atomic<uint8_t>& f_;
};
}  // namespace

HnswGraph::HnswGraph(VectorMetric metric, size_t dim, size_t maxElements, size_t M, size_t efConstruction, size_t randomSeed)
	: metric_(metric),
	  dim_(dim),
	  maxElements_(maxElements),
	  M_(std::min<size_t>(M, 10'000)),          // hnswalg.h:254-262
	  maxM0_(2 * M_),
	  efConstruction_(std::max(efConstruction, M_)),
	  mult_(1.0 / std::log(1.0 * double(M_))) {
	if (dim_ == 0) throw std::logic_error("HnswGraph: zero dimension");
	try {
		ScopedInterleave pages(maxElements_ * dim_ * sizeof(float));   // rows + link lists spread over every socket's DRAM (numa_policy.h)
		vectors_.resize(maxElements_ * dim_);
		if (metric_ == VectorMetric::Cosine) invNorms_.resize(maxElements_);
		links0_.assign(maxElements_ * (1 + maxM0_), 0u);
		upper_.resize(maxElements_);
		levels_.assign(maxElements_, 0);
		labels_.assign(maxElements_, 0);
		deleted_.assign(maxElements_, 0);
		visited_.stamp.assign(maxElements_, 0);
	} catch (const std::bad_alloc&) {
		throw std::runtime_error("Not enough memory: HNSW constructor failed to allocate level0");
	}
	levelGenerator_.seed(uint32_t(randomSeed));
	randomSeed_ = randomSeed;
}

HnswGraph::HnswGraph(const HnswGraph& o, size_t newMaxElements)
	: metric_(o.metric_),
	  dim_(o.dim_),
	  maxElements_(std::max(o.maxElements_, newMaxElements)),
	  M_(o.M_),
	  maxM0_(o.maxM0_),
	  efConstruction_(o.efConstruction_),
	  mult_(o.mult_),
	  count_(o.count_),
	  numDeleted_(o.numDeleted_),
	  maxLevel_(o.maxLevel_),
	  entryPoint_(o.entryPoint_),
	  vectors_(o.vectors_),
	  invNorms_(o.invNorms_),
	  links0_(o.links0_),
	  upper_(o.upper_),
	  levels_(o.levels_),
	  labels_(o.labels_),
	  deleted_(o.deleted_),
	  labelLookup_(o.labelLookup_),
	  levelGenerator_(o.levelGenerator_) {
	visited_.stamp.assign(o.visited_.stamp.size(), 0);
	rebuildDeletedSet();   // the reference's copy constructor refills deleted_elements through initTree, in index order (hnswalg.h:492, 1265-1281)
	if (maxElements_ != o.maxElements_) {
		const size_t keep = maxElements_;
		maxElements_ = o.maxElements_;
		Resize(keep);
	}
}

void HnswGraph::Resize(size_t newMaxElements) {
	if (newMaxElements < count_) throw std::runtime_error("Cannot resize, max element is less than the current number of elements");
	try {
		ScopedInterleave pages(newMaxElements * dim_ * sizeof(float));
		vectors_.resize(newMaxElements * dim_);
		if (metric_ == VectorMetric::Cosine) invNorms_.resize(newMaxElements);
		links0_.resize(newMaxElements * (1 + maxM0_), 0u);
		upper_.resize(newMaxElements);
		levels_.resize(newMaxElements, 0);
		labels_.resize(newMaxElements, 0);
		deleted_.resize(newMaxElements, 0);
		visited_.stamp.assign(newMaxElements, 0);
		visited_.cur = 0;
		visitedPool_.clear();
		nodeLocks_.reset();
		nodeLocksSize_ = 0;
	} catch (const std::bad_alloc&) {
		throw std::runtime_error("Not enough memory: resizeIndex failed to allocate base layer");
	}
	maxElements_ = newMaxElements;
	dirtyAll_.store(true, std::memory_order_relaxed);
	if (concurrent_) EnableConcurrentInserts();
}

void HnswGraph::markDirty(tableint id) {
	if (dirtyAll_.load(std::memory_order_relaxed)) return;
	std::lock_guard<std::mutex> lk(dirtyMtx_);
	if (dirty_.size() > std::max<size_t>(count_ / 4, 4096)) {
		dirty_.clear();
		dirtyAll_.store(true, std::memory_order_relaxed);
		return;
	}
	dirty_.push_back(id);
}

bool HnswGraph::TakeDirty(std::vector<tableint>& out) {
	std::lock_guard<std::mutex> lk(dirtyMtx_);
	out.clear();
	const bool all = dirtyAll_.exchange(false, std::memory_order_relaxed);
	if (!all) {
		std::sort(dirty_.begin(), dirty_.end());
		dirty_.erase(std::unique(dirty_.begin(), dirty_.end()), dirty_.end());
		out.swap(dirty_);
	}
	dirty_.clear();
	return !all;
}

tableint HnswGraph::InternalId(labeltype label) const {
	auto it = labelLookup_.find(label);
	if (it == labelLookup_.end()) throw std::runtime_error("Label not found");
	return it->second;
}

size_t HnswGraph::AllocatedMemSize() const noexcept {
	size_t up = 0;
	for (size_t i = 0; i < count_; ++i) up += upper_[i].capacity() * sizeof(uint32_t);
	return vectors_.capacity() * sizeof(float) + links0_.capacity() * sizeof(uint32_t) + up +
		   labelLookup_.size() * (sizeof(labeltype) + sizeof(tableint) + 2 * sizeof(void*)) + sizeof(HnswGraph);
}

// DistCalculator<float>::operator()(v1,id1,v2,id2): smaller = closer; cosine multiplies by BOTH stored 1/|v| (hnswlib.h:123-145)
float HnswGraph::distIds(tableint a, tableint b) const noexcept {
	const float* va = Vector(a);
	const float* vb = Vector(b);
	if (metric_ == VectorMetric::L2) return 1.0f * L2SqrAvx512Order(va, vb, dim_) + 0.0f + 0.0f;
	float d = -(1.0f * InnerProductAvx512Order(va, vb, dim_) + 0.0f + 0.0f);
	if (metric_ == VectorMetric::Cosine) {
		d *= invNorms_[a];
		d *= invNorms_[b];
	}
	return d;
}

uint32_t* HnswGraph::list(tableint id, int level) noexcept {
	return level == 0 ? links0_.data() + size_t(id) * (1 + maxM0_) : upper_[id].data() + size_t(level - 1) * (1 + M_);
}
const uint32_t* HnswGraph::list(tableint id, int level) const noexcept { return const_cast<HnswGraph*>(this)->list(id, level); }

// level = floor(-ln(U) / ln(M)), U from std::default_random_engine — a fresh distribution object per draw, like the reference
int HnswGraph::randomLevel() {
	std::uniform_real_distribution<double> distribution(0.0, 1.0);
	const double r = -std::log(distribution(levelGenerator_)) * mult_;
	return int(r);
}

// Best-first search on one layer with beam efConstruction; returns the beam as a max-heap on distance.
// kMT: a node's list is snapshotted under its lock (the reference keeps the lock over the distance evaluations, hnswalg.h:672-676; the
// snapshot sees the same list and does not hold other inserters up).
template <bool kMT>
HnswGraph::Heap HnswGraph::searchBaseLayer(tableint ep, tableint self, int layer, Visited& vis) {
	if (++vis.cur == 0) {   // stamp wrap-around: clear and restart at 1
		std::fill(vis.stamp.begin(), vis.stamp.end(), uint16_t(0));
		vis.cur = 1;
	}
	const uint16_t stamp = vis.cur;
	uint16_t* const visitStamp = vis.stamp.data();
	Heap beam, frontier;   // beam: worst on top; frontier: keyed by -dist so the closest is on top
	beam.reserve(256);
	frontier.reserve(256);
	std::vector<uint32_t> snapshot;
	if constexpr (kMT) snapshot.resize(1 + maxM0_);
	float bound;
	if (!IsDeleted(ep)) {
		const float d = distIds(self, ep);
		beam.emplace(d, ep);
		bound = d;
		frontier.emplace(-d, ep);
	} else {
		bound = std::numeric_limits<float>::max();
		frontier.emplace(-bound, ep);
	}
	visitStamp[ep] = stamp;
	while (!frontier.empty()) {
		const Pair cur = frontier.top();
		if (-cur.first > bound && beam.size() == efConstruction_) break;
		frontier.pop();
		const uint32_t* ll;
		if constexpr (kMT) {
			NodeLock lk(nodeLocks_[cur.second]);
			const uint32_t* src = list(cur.second, layer);
			std::memcpy(snapshot.data(), src, (1 + size_t(src[0])) * sizeof(uint32_t));
			ll = snapshot.data();
		} else {
			ll = list(cur.second, layer);
		}
		const size_t size = ll[0];
		if (size) {   // the reference prefetches the same way (hnswalg.h:674-716): stamp + head of the vector of the neighbour that comes next
			__builtin_prefetch(&visitStamp[ll[1]]);
			__builtin_prefetch(Vector(ll[1]));
		}
		for (size_t j = 0; j < size; ++j) {
			const tableint cand = ll[1 + j];
			if (j + 1 < size) {
				const tableint next = ll[2 + j];
				__builtin_prefetch(&visitStamp[next]);
				const char* nv = reinterpret_cast<const char*>(Vector(next));
				__builtin_prefetch(nv);
				__builtin_prefetch(nv + 64);
			}
			if (visitStamp[cand] == stamp) continue;
			visitStamp[cand] = stamp;
			const float d = distIds(self, cand);
			if (beam.size() < efConstruction_ || bound > d) {
				frontier.emplace(-d, cand);
				if (!IsDeleted(cand)) {
					if (beam.size() < efConstruction_) {
						beam.emplace(d, cand);
					} else {
						beam.replace_top(Pair(d, cand));
					}
				}
				if (!beam.empty()) bound = beam.top().first;
			}
		}
	}
	return beam;
}

// Diversity heuristic: walk candidates closest-first, keep one only if it is closer to the base point than to every
// neighbour already kept.  In/out: a max-heap on distance (out holds -dist keys exactly like the reference re-push).
void HnswGraph::selectNeighbors(Heap& candidates, size_t M) const {
	if (candidates.size() < M) return;
	ResultHeap<Pair> closest;   // lexicographic (std::less<pair>) on (-dist, id), as in the reference
	std::vector<Pair> kept;
	while (candidates.size() > 0) {
		closest.emplace(-candidates.top().first, candidates.top().second);
		candidates.pop();
	}
	while (closest.size()) {
		if (kept.size() >= M) break;
		const Pair cur = closest.top();
		const float distToBase = -cur.first;
		closest.pop();
		bool good = true;
		for (const Pair& k : kept) {
			if (distIds(k.second, cur.second) < distToBase) {
				good = false;
				break;
			}
		}
		if (good) kept.push_back(cur);
	}
	for (const Pair& k : kept) candidates.emplace(-k.first, k.second);
}

// Link `cur` on `level` to the selected neighbours and back; returns the entry point for the next (lower) level.
// kMT: cur's own lists are covered by the lock its inserter holds for the whole insertion; every other list is rewritten under its lock
// (mutuallyConnectNewElement, hnswalg.h:1086-1095).
template <bool kMT>
tableint HnswGraph::connect(tableint cur, Heap& candidates, int level, bool isUpdate) {
	const size_t mCurMax = level ? M_ : maxM0_;
	selectNeighbors(candidates, M_);
	if (candidates.size() > M_) throw std::runtime_error("Should be not be more than M_ candidates returned by the heuristic");
	std::vector<tableint> selected;
	selected.reserve(M_);
	while (candidates.size() > 0) {
		selected.push_back(candidates.top().second);
		candidates.pop();
	}
	const tableint nextEntry = selected.back();
	{
		uint32_t* ll = list(cur, level);
		if (ll[0] != 0 && !isUpdate) throw std::runtime_error("The newly inserted element should have blank link list");
		ll[0] = uint32_t(selected.size());
		for (size_t i = 0; i < selected.size(); ++i) {
			if (level > levels_[selected[i]]) throw std::runtime_error("Trying to make a link on a non-existent level");
			ll[1 + i] = selected[i];
		}
		for (size_t i = selected.size(); i < mCurMax; ++i) ll[1 + i] = 0;   // canonical unused slots (an updated element's list may shrink)
		markDirty(cur);
	}
	for (const tableint other : selected) {
		[[maybe_unused]] std::unique_ptr<NodeLock> lk;
		if constexpr (kMT) lk = std::make_unique<NodeLock>(nodeLocks_[other]);
		uint32_t* lo = list(other, level);
		const size_t sz = lo[0];
		if (sz > mCurMax) throw std::runtime_error("Bad value of sz_link_list_other");
		if (other == cur) throw std::runtime_error("Trying to connect an element to itself");
		if (isUpdate) {   // hnswalg.h:1119-1131: an updated element may already be among the neighbour's links — then nothing changes there
			bool present = false;
			for (size_t j = 0; j < sz && !present; ++j) present = lo[1 + j] == cur;
			if (present) continue;
		}
		markDirty(other);
		if (sz < mCurMax) {
			lo[1 + sz] = cur;
			lo[0] = uint32_t(sz + 1);
		} else {
			// full: re-select among {cur} U current neighbours of `other`
			Heap pool;
			pool.emplace(distIds(cur, other), cur);
			for (size_t j = 0; j < sz; ++j) pool.emplace(distIds(lo[1 + j], other), lo[1 + j]);
			selectNeighbors(pool, mCurMax);
			uint32_t idx = 0;
			while (pool.size() > 0) {
				lo[1 + idx] = pool.top().second;
				pool.pop();
				++idx;
			}
			lo[0] = idx;
			for (size_t j = idx; j < mCurMax; ++j) lo[1 + j] = 0;   // keep unused slots canonical (the flat export is compared / uploaded verbatim)
		}
	}
	return nextEntry;
}

template <bool kMT>
tableint HnswGraph::addPoint(const float* data, labeltype label) {
	tableint cur;
	{
		[[maybe_unused]] std::unique_lock<std::mutex> lockTable;
		if constexpr (kMT) lockTable = std::unique_lock<std::mutex>(labelMtx_);
		if (auto found = labelLookup_.find(label); found != labelLookup_.end()) {
			// hnswalg.h:1709-1724: the label exists -> the element is updated in place instead of a new one being created
			const tableint existing = found->second;
			if (IsDeleted(existing)) throw std::runtime_error("Can't use addPoint to update deleted elements if replacement of deleted elements is enabled.");
			if constexpr (kMT) {
				throw std::logic_error("HnswGraph::AddPointConcurrent: in-place update of an existing label needs the graph exclusively (use AddPoint)");
			} else {
				updatePoint(data, existing);
				return existing;
			}
		}
		if (count_ >= maxElements_) throw std::runtime_error("The number of elements exceeds the specified limit");
		cur = tableint(count_);
		count_++;
		labelLookup_[label] = cur;
		if (metric_ == VectorMetric::Cosine) invNorms_[cur] = CalculateL2Module(data, int32_t(dim_));
	}

	int curLevel;
	if constexpr (kMT) {
		std::lock_guard<std::mutex> lk(generatorMtx_);
		curLevel = randomLevel();
	} else {
		curLevel = randomLevel();
	}

	// `global` stays locked only while this element creates a new top level; the element's own lock is held to the end
	[[maybe_unused]] std::unique_lock<std::mutex> tempLock;
	[[maybe_unused]] std::unique_ptr<NodeLock> lockEl;
	std::unique_ptr<Visited> pooled;
	if constexpr (kMT) {
		tempLock = std::unique_lock<std::mutex>(globalMtx_);
		lockEl = std::make_unique<NodeLock>(nodeLocks_[cur]);
	}
	levels_[cur] = curLevel;
	int maxLevelCopy;
	tableint enterCopy;
	if constexpr (kMT) {
		std::lock_guard<std::mutex> lk(entryMtx_);
		maxLevelCopy = maxLevel_;
		enterCopy = entryPoint_;
		if (curLevel <= maxLevelCopy && enterCopy != kNoEntry) tempLock.unlock();
	} else {
		maxLevelCopy = maxLevel_;
		enterCopy = entryPoint_;
	}
	tableint currObj = enterCopy;

	std::memset(list(cur, 0), 0, (1 + maxM0_) * sizeof(uint32_t));
	markDirty(cur);
	labels_[cur] = label;
	deleted_[cur] = 0;
	std::memcpy(vectors_.data() + size_t(cur) * dim_, data, dim_ * sizeof(float));
	upper_[cur].assign(size_t(curLevel) * (1 + M_), 0u);

	if (currObj != kNoEntry) {
		if (curLevel < maxLevelCopy) {
			float curDist = distIds(cur, currObj);
			std::vector<uint32_t> snapshot;
			if constexpr (kMT) snapshot.resize(1 + M_);
			for (int level = maxLevelCopy; level > curLevel; --level) {
				bool changed = true;
				while (changed) {
					changed = false;
					const uint32_t* ll;
					if constexpr (kMT) {
						NodeLock lk(nodeLocks_[currObj]);
						const uint32_t* src = list(currObj, level);
						std::memcpy(snapshot.data(), src, (1 + size_t(src[0])) * sizeof(uint32_t));
						ll = snapshot.data();
					} else {
						ll = list(currObj, level);
					}
					const int size = int(ll[0]);
					for (int i = 0; i < size; ++i) {
						const tableint cand = ll[1 + i];
						if (cand >= maxElements_) throw std::runtime_error("cand error");
						const float d = distIds(cur, cand);
						if (d < curDist) {
							curDist = d;
							currObj = cand;
							changed = true;
						}
					}
				}
			}
		}
		Visited* vis = &visited_;
		if constexpr (kMT) {
			pooled = acquireVisited();
			vis = pooled.get();
		}
		for (int level = std::min(curLevel, maxLevelCopy); level >= 0; --level) {
			Heap top = searchBaseLayer<kMT>(currObj, cur, level, *vis);
			if (IsDeleted(enterCopy)) {   // hnswalg.h:1819-1828: a deleted entry point is still offered as a neighbour
				const float d = distIds(cur, enterCopy);
				if (top.size() < efConstruction_) {
					top.emplace(d, enterCopy);
				} else if (top.top().first > d) {
					top.replace_top(Pair(d, enterCopy));
				}
			}
			currObj = connect<kMT>(cur, top, level);
		}
		if constexpr (kMT) releaseVisited(std::move(pooled));
		if (curLevel > maxLevelCopy) {
			[[maybe_unused]] std::unique_lock<std::mutex> lk;
			if constexpr (kMT) lk = std::unique_lock<std::mutex>(entryMtx_);
			entryPoint_ = cur;
			maxLevel_ = curLevel;
		}
	} else {
		[[maybe_unused]] std::unique_lock<std::mutex> lk;
		if constexpr (kMT) lk = std::unique_lock<std::mutex>(entryMtx_);
		maxLevel_ = curLevel;   // first element (hnswalg.h:1839-1848)
		entryPoint_ = curLevel > maxLevelCopy ? cur : 0;
	}
	return cur;
}

// addPoint<LockerT>(data_point, label), hnswalg.h:1401-1470 (allow_replace_deleted_ is always on: hnsw.h:72): a vacated slot is recycled first
tableint HnswGraph::AddPoint(const float* data, labeltype label) {
	if (deletedElements_.empty()) return addPoint<false>(data, label);
	const tableint id = deletedElements_.pop_front();
	numDeleted_ -= 1;
	labels_[id] = label;
	labelLookup_[label] = id;
	updatePoint(data, id);
	return id;
}

void HnswGraph::rebuildDeletedSet() {
	deletedElements_ = DeletedIdSet();
	for (size_t i = 0; i < count_; ++i) {
		if (deleted_[i]) deletedElements_.insert(tableint(i));
	}
}

// updatePoint<LockerT>(dataPointRaw, internalId, 1.0), hnswalg.h:1472-1587
void HnswGraph::updatePoint(const float* data, tableint id) {
	std::memcpy(vectors_.data() + size_t(id) * dim_, data, dim_ * sizeof(float));
	markDirty(id);
	if (metric_ == VectorMetric::Cosine) invNorms_[id] = CalculateL2Module(data, int32_t(dim_));   // AddNorm
	if (deleted_[id]) {   // unmarkDeletedInternal :1342-1361
		deleted_[id] = 0;
		if (deletedElements_.erase(id)) numDeleted_ -= 1;
	}
	const int maxLevelCopy = maxLevel_;
	const tableint entryPointCopy = entryPoint_;
	if (entryPointCopy == id && count_ == 1) return;   // the graph is this single element
	const int elemLevel = levels_[id];
	for (int layer = 0; layer <= elemLevel; ++layer) {
		CandidateIdSet sCand, sNeigh;   // reindexer::fast_hash_set<tableint>: walked in the reference's hash order
		const uint32_t* l1 = list(id, layer);
		const std::vector<tableint> listOneHop(l1 + 1, l1 + 1 + l1[0]);
		if (listOneHop.empty()) continue;
		sCand.insert(id);
		for (const tableint elOneHop : listOneHop) {
			sCand.insert(elOneHop);
			sNeigh.insert(elOneHop);   // updateNeighborProbability == 1.0: every one-hop neighbour is re-selected
			const uint32_t* l2 = list(elOneHop, layer);
			for (uint32_t j = 0; j < l2[0]; ++j) sCand.insert(l2[1 + j]);
		}
		const size_t mLayer = layer == 0 ? maxM0_ : M_;
		sNeigh.for_each([&](tableint neigh) {
			Heap candidates;
			const size_t size = sCand.count(neigh) ? sCand.size() - 1 : sCand.size();
			const size_t elementsToKeep = std::min(efConstruction_, size);
			sCand.for_each([&](tableint cand) {
				if (cand == neigh) return;
				const float distance = distIds(neigh, cand);
				if (candidates.size() < elementsToKeep) {
					candidates.emplace(distance, cand);
				} else if (distance < candidates.top().first) {
					candidates.pop();
					candidates.emplace(distance, cand);
				}
			});
			selectNeighbors(candidates, mLayer);
			uint32_t* ll = list(neigh, layer);
			const size_t candSize = candidates.size();
			ll[0] = uint32_t(candSize);
			for (size_t idx = 0; idx < candSize; ++idx) {
				ll[1 + idx] = candidates.top().second;
				candidates.pop();
			}
			for (size_t idx = candSize; idx < mLayer; ++idx) ll[1 + idx] = 0;
			markDirty(neigh);
		});
	}
	repairConnectionsForUpdate(id, entryPointCopy, elemLevel, maxLevelCopy);
}

// hnswalg.h:1589-1680
void HnswGraph::repairConnectionsForUpdate(tableint id, tableint entryPoint, int level, int maxLevel) {
	tableint currObj = entryPoint;
	if (level < maxLevel) {
		float curDist = distIds(id, currObj);
		for (int l = maxLevel; l > level; --l) {
			bool changed = true;
			while (changed) {
				changed = false;
				const uint32_t* ll = list(currObj, l);
				const int size = int(ll[0]);
				for (int i = 0; i < size; ++i) {
					const tableint cand = ll[1 + i];
					const float d = distIds(id, cand);
					if (d < curDist) {
						curDist = d;
						currObj = cand;
						changed = true;
					}
				}
			}
		}
	}
	if (level > maxLevel) throw std::runtime_error("Level of item to be updated cannot be bigger than max level");
	for (int l = level; l >= 0; --l) {
		Heap top = searchBaseLayer<false>(currObj, id, l, visited_);
		Heap filtered;   // the element itself is dropped (pushed in pop order, like the reference does)
		while (top.size() > 0) {
			if (top.top().second != id) filtered.push(top.top());
			top.pop();
		}
		// element_levels_ is used to get the level, so `top` may hold nothing but the element itself: then the level stays as it is
		if (filtered.size() > 0) {
			if (IsDeleted(entryPoint)) {
				filtered.emplace(distIds(id, entryPoint), entryPoint);
				if (filtered.size() > efConstruction_) filtered.pop();
			}
			currObj = connect<false>(id, filtered, l, true);
		}
	}
}

void HnswGraph::EnableConcurrentInserts() {
	std::lock_guard<std::mutex> lk(visitedPoolMtx_);
	if (nodeLocksSize_ == maxElements_ && nodeLocks_) return;
	nodeLocks_.reset(new std::atomic<uint8_t>[maxElements_]);
	for (size_t i = 0; i < maxElements_; ++i) nodeLocks_[i].store(0, std::memory_order_relaxed);
	nodeLocksSize_ = maxElements_;
	labelLookup_.reserve(maxElements_);
	concurrent_ = true;
}

std::unique_ptr<HnswGraph::Visited> HnswGraph::acquireVisited() {
	{
		std::lock_guard<std::mutex> lk(visitedPoolMtx_);
		if (!visitedPool_.empty()) {
			auto v = std::move(visitedPool_.back());
			visitedPool_.pop_back();
			return v;
		}
	}
	auto v = std::make_unique<Visited>();
	v->stamp.assign(maxElements_, 0);
	return v;
}

void HnswGraph::releaseVisited(std::unique_ptr<Visited> v) {
	std::lock_guard<std::mutex> lk(visitedPoolMtx_);
	visitedPool_.push_back(std::move(v));
}

tableint HnswGraph::AddPointConcurrent(const float* data, labeltype label) {
	if (!nodeLocks_ || nodeLocksSize_ != maxElements_) EnableConcurrentInserts();
	// A vacated slot is recycled first, as in the reference (hnswalg.h:1410-1421).  The reference then runs updatePoint next to other inserts
	// behind per-element data locks (ExpectConcurrentUpdates::Yes); here the recycling insert takes the graph exclusively instead: same
	// graph invariants, and slot reuse is the rare case of a bulk build.
	bool vacant = false;
	tableint id = 0;
	{
		std::lock_guard<std::mutex> lk(deletedMtx_);
		if (!deletedElements_.empty()) {
			vacant = true;
			id = deletedElements_.pop_front();
			numDeleted_ -= 1;
		}
	}
	if (!vacant) {
		std::shared_lock<std::shared_mutex> shared(updateMtx_);
		return addPoint<true>(data, label);
	}
	std::unique_lock<std::shared_mutex> exclusive(updateMtx_);
	labels_[id] = label;
	{
		std::lock_guard<std::mutex> lk(labelMtx_);
		labelLookup_[label] = id;
	}
	updatePoint(data, id);
	return id;
}

void HnswGraph::AddPoints(const float* data, const labeltype* labels, size_t n, unsigned threads) {
	if (n == 0) return;
	if (threads <= 1) {
		for (size_t i = 0; i < n; ++i) AddPoint(data + i * dim_, labels[i]);
		return;
	}
	EnableConcurrentInserts();
	size_t first = 0;
	if (count_ == 0) {   // the element that creates the entry point goes in alone
		AddPointConcurrent(data, labels[0]);
		first = 1;
	}
	std::atomic<size_t> next{first};
	std::mutex errMtx;
	std::string error;
	auto worker = [&] {
		try {
			for (;;) {
				const size_t i = next.fetch_add(1, std::memory_order_relaxed);
				if (i >= n) break;
				AddPointConcurrent(data + i * dim_, labels[i]);
			}
		} catch (const std::exception& e) {
			next.store(n, std::memory_order_relaxed);
			std::lock_guard<std::mutex> lk(errMtx);
			if (error.empty()) error = e.what();
		}
	};
	std::vector<std::thread> pool;
	pool.reserve(threads);
	for (unsigned t = 0; t < threads; ++t) pool.emplace_back(worker);
	for (auto& t : pool) t.join();
	if (!error.empty()) throw std::runtime_error(error);
}

void HnswGraph::MarkDelete(labeltype label) {
	auto it = labelLookup_.find(label);
	if (it == labelLookup_.end()) throw std::runtime_error("markDelete: Label not found: " + std::to_string(label));
	const tableint id = it->second;
	if (deleted_[id]) throw std::runtime_error("The requested to delete element is already deleted");
	deleted_[id] = 1;
	numDeleted_ += 1;
	deletedElements_.insert(id);   // markDeletedInternal :1323-1338 (allow_replace_deleted_)
	labelLookup_.erase(it);        // allow_replace_deleted_ == true in the reference's construction (hnsw.h:72)
	markDirty(id);                 // the flag travels with the next incremental patch of the device mirror (GpuHnswMap::syncDevice)
}

// ---------------------------------------------------------------------------------------------------- ANN disk cache
void HnswGraph::SaveIndex(AnnCacheWriter& writer, const std::atomic_int32_t& cancel) const {
	constexpr size_t kCancelPeriod = 0x3FFFFF;
	writer.PutVarUInt(uint64_t(maxElements_));
	writer.PutVarUInt(uint64_t(count_));
	writer.PutVarInt(int32_t(maxLevel_));
	writer.PutVarUInt(uint32_t(entryPoint_));
	writer.PutVarUInt(uint32_t(M_));
	writer.PutVarUInt(uint32_t(efConstruction_));
	for (size_t i = 0; i < count_; ++i) {
		if (((i & kCancelPeriod) == kCancelPeriod) && cancel.load(std::memory_order_relaxed)) throw std::runtime_error("HNSW index saving was canceled");
		const uint32_t* l0 = list(tableint(i), 0);
		const uint32_t n = l0[0] & 0xFFFFu;   // getListCount: the low 16 bits; the delete mark is bit 0 of byte 2 (hnswalg.h:204, 1366-1374)
		writer.PutVarUInt(uint32_t(n | (deleted_[i] ? 0x10000u : 0u)));
		for (uint32_t j = 1; j <= n; ++j) writer.PutVarUInt(uint32_t(l0[j]));
		if (deleted_[i]) {   // "We have to store full vector for the deleted items"
			writer.PutVString(std::string_view(reinterpret_cast<const char*>(Vector(tableint(i))), dim_ * sizeof(float)));
		} else {
			writer.AppendPKByID(labels_[i]);
		}
	}
	std::vector<uint32_t> block;
	for (size_t i = 0; i < count_; ++i) {
		if (((i & kCancelPeriod) == kCancelPeriod) && cancel.load(std::memory_order_relaxed)) throw std::runtime_error("HNSW index saving was canceled");
		// the reference writes the raw block (:1259-1261), slots past each list's count included; those carry no meaning (every reader
		// stops at the count), so this writer zeroes them: one graph, one byte string
		const size_t words = levels_[i] > 0 ? size_t(levels_[i]) * (1 + M_) : 0;
		block.assign(upper_[i].begin(), upper_[i].begin() + ptrdiff_t(words));
		for (size_t lv = 0; lv < size_t(std::max(levels_[i], 0)); ++lv) {
			uint32_t* ll = block.data() + lv * (1 + M_);
			for (size_t j = 1 + (ll[0] & 0xFFFFu); j <= M_; ++j) ll[j] = 0;
		}
		writer.PutVString(std::string_view(reinterpret_cast<const char*>(block.data()), words * sizeof(uint32_t)));
	}
}

void HnswGraph::LoadIndex(AnnCacheReader& reader) {
	if (count_ != 0) throw std::logic_error("HnswGraph::LoadIndex: the graph is not empty");
	try {
		loadIndex(reader);
	} catch (...) {
		Clear();   // a half-read cache leaves nothing behind
		throw;
	}
}

void HnswGraph::loadIndex(AnnCacheReader& reader) {
	const uint64_t maxElements = reader.GetVarUInt();
	const uint64_t cnt = reader.GetVarUInt();
	if (cnt > maxElements) throw std::runtime_error("Current elements count is larger than max elements count");
	const int64_t maxLevel = reader.GetVarInt();
	const uint64_t entry = reader.GetVarUInt();
	if (cnt) {
		if (entry >= cnt) throw std::runtime_error("Incorrect entrypoint node ID");
	} else if (entry != 0xFFFFFFFFull) {
		throw std::runtime_error("Unexpected entrypoint node ID for empty HNSW");
	}
	const uint64_t M = reader.GetVarUInt(), efC = reader.GetVarUInt();
	if (M != M_ || efC != efConstruction_) {
		throw std::runtime_error("HnswGraph::LoadIndex: the cache was written with M = " + std::to_string(M) + ", efConstruction = " + std::to_string(efC) +
								 ", the index is defined with " + std::to_string(M_) + " / " + std::to_string(efConstruction_));
	}
	if (maxElements >= 0xFFFFFFFFull) throw std::runtime_error("HnswGraph::LoadIndex: max elements out of range");
	if (maxElements > maxElements_) Resize(size_t(maxElements));
	const size_t levelBytes = (1 + M_) * sizeof(uint32_t);
	size_t deletedCount = 0;
	for (size_t i = 0; i < cnt; ++i) {
		uint32_t* l0 = list(tableint(i), 0);
		const uint64_t marked = reader.GetVarUInt();
		const uint32_t n = uint32_t(marked) & 0xFFFFu;
		if (n > maxM0_) throw std::runtime_error("HnswGraph::LoadIndex: a level-0 list is longer than maxM0");
		const bool isDeleted = ((marked >> 16) & 1u) != 0;
		l0[0] = n;
		for (uint32_t j = 1; j <= n; ++j) {
			const uint64_t link = reader.GetVarUInt();
			if (link >= cnt) throw std::runtime_error("HnswGraph::LoadIndex: a link points past the element count");
			l0[j] = uint32_t(link);
		}
		float* dest = vectors_.data() + i * dim_;
		labeltype label = std::numeric_limits<labeltype>::max();
		if (isDeleted) {
			const std::string_view vec = reader.GetVString();
			if (vec.size() != dim_ * sizeof(float)) throw std::runtime_error("HnswGraph::LoadIndex: a deleted element's vector has the wrong size");
			std::memcpy(dest, vec.data(), vec.size());
		} else {
			label = reader.ReadPkEncodedData(dest);
		}
		if (metric_ == VectorMetric::Cosine) invNorms_[i] = CalculateL2Module(dest, int32_t(dim_));   // AddNorm
		labels_[i] = label;
		deleted_[i] = isDeleted ? 1 : 0;
		deletedCount += isDeleted ? 1 : 0;
	}
	count_ = size_t(cnt);
	// initTree (hnswalg.h:1264-1281): deleted slots into the reuse set in id order, live labels into the lookup, levels from the list sizes
	numDeleted_ = 0;
	for (size_t i = 0; i < cnt; ++i) {
		if (deleted_[i]) {
			numDeleted_ += 1;
			deletedElements_.insert(tableint(i));
		} else {
			labelLookup_[labels_[i]] = tableint(i);
		}
		const std::string_view lists = reader.GetVString();
		if (lists.size() % levelBytes != 0) throw std::runtime_error("HnswGraph::LoadIndex: an upper-level block has a broken size");
		levels_[i] = int32_t(lists.size() / levelBytes);
		upper_[i].resize(lists.size() / sizeof(uint32_t));
		if (!lists.empty()) std::memcpy(upper_[i].data(), lists.data(), lists.size());
		for (int lv = 1; lv <= levels_[i]; ++lv) {
			const uint32_t* ll = list(tableint(i), lv);
			const uint32_t n = ll[0] & 0xFFFFu;
			if (n > M_) throw std::runtime_error("HnswGraph::LoadIndex: an upper-level list is longer than M");
			for (uint32_t j = 1; j <= n; ++j) {
				if (ll[j] >= cnt) throw std::runtime_error("HnswGraph::LoadIndex: a link points past the element count");
			}
			for (size_t j = size_t(n) + 1; j <= M_; ++j) list(tableint(i), lv)[j] = 0;   // stale slots of the writer's raw block
		}
	}
	if (numDeleted_ != deletedCount) throw std::logic_error("HnswGraph::LoadIndex: delete marks out of step");
	// third pass, once every element's level is known: a link on level lv must lead to an element that HAS a level-lv list (search, insert and
	// the device export index upper_[target] by the level they arrived on), and nothing may stand above the stored top level
	for (size_t i = 0; i < cnt; ++i) {
		if (levels_[i] > int(maxLevel)) throw std::runtime_error("HnswGraph::LoadIndex: an element stands above the stored top level");
		for (int lv = 1; lv <= levels_[i]; ++lv) {
			const uint32_t* ll = list(tableint(i), lv);
			const uint32_t n = ll[0] & 0xFFFFu;
			for (uint32_t j = 1; j <= n; ++j) {
				if (levels_[ll[j]] < lv) throw std::runtime_error("HnswGraph::LoadIndex: an upper-level link leads to an element without that level");
			}
		}
	}
	maxLevel_ = int(maxLevel);
	entryPoint_ = tableint(entry);
	if (cnt && levels_[entryPoint_] != maxLevel_) throw std::runtime_error("HnswGraph::LoadIndex: the entry point is not on the top level");
	dirty_.clear();
	dirtyAll_.store(true, std::memory_order_relaxed);   // the device mirror takes the whole graph
}

void HnswGraph::Clear() {
	count_ = 0;
	numDeleted_ = 0;
	maxLevel_ = -1;
	entryPoint_ = 0xFFFFFFFFu;
	std::fill(links0_.begin(), links0_.end(), 0u);
	for (auto& u : upper_) u.clear();
	std::fill(levels_.begin(), levels_.end(), 0);
	std::fill(labels_.begin(), labels_.end(), labeltype(0));
	std::fill(deleted_.begin(), deleted_.end(), uint8_t(0));
	labelLookup_.clear();
	deletedElements_ = DeletedIdSet();
	levelGenerator_.seed(uint32_t(randomSeed_));
	std::fill(visited_.stamp.begin(), visited_.stamp.end(), uint16_t(0));
	visited_.cur = 0;
	dirty_.clear();
	dirtyAll_.store(true, std::memory_order_relaxed);
}

void HnswGraph::ExportUpper(std::vector<uint64_t>& off, std::vector<uint32_t>& blocks) const {
	off.assign(count_ + 1, 0);
	uint64_t total = 0;
	for (size_t i = 0; i < count_; ++i) {
		off[i] = total;
		total += uint64_t(levels_[i]);
	}
	off[count_] = total;
	blocks.assign(std::max<uint64_t>(total, 1) * (1 + M_), 0u);
	for (size_t i = 0; i < count_; ++i) {
		if (levels_[i]) std::memcpy(blocks.data() + off[i] * (1 + M_), upper_[i].data(), upper_[i].size() * sizeof(uint32_t));
	}
}

}  // namespace rxgpu::host
