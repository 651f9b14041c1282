// Boundary types of the GPU float_vector engines, mirroring the reference's (so that a maintainer can swap these
// aliases for the reference's own headers when the engine is compiled inside cpp_src — see INTEGRATION.md):
//   VectorMetric            cpp_src/core/enums.h:101
//   labeltype / tableint    cpp_src/core/index/float_vector/hnswlib/type_consts.h:8-9
//   FloatVectorId           cpp_src/core/index/float_vector/float_vector_id.h:8-22   (rowId << 32 | arrayIdx)
//   ConstFloatVectorView    cpp_src/core/keyvalue/float_vector.h:12-72               (pointer + dimension)
//   SearchResultQueue       cpp_src/core/index/float_vector/hnswlib/hnsw_interface.h:14 +
//                           priority_queue.h:7-152  (max-heap of (dist,label), std::less<pair> => lexicographic)
//
// RXGPU_IN_TREE (the engine compiled inside cpp_src): the aliases below ARE the reference's own types — the Maps then
// compile against reindexer::FloatVectorId / ConstFloatVectorView / hnswlib::SearchResultQueue unchanged, which is what
// tests/test_seam_compile.py builds (HnswIndexBase<GpuBruteforceMap> / <GpuHnswMapT<...>> instantiated from the reference's hnsw_index.cc).
#pragma once

#include <cstddef>
#include <cstdint>
#include <functional>
#include <utility>
#include <vector>

#if defined(RXGPU_IN_TREE)
#include "core/enums.h"
#include "core/index/float_vector/float_vector_id.h"
#include "core/index/float_vector/hnswlib/hnsw_interface.h"
#include "core/index/float_vector/hnswlib/type_consts.h"
#include "core/keyvalue/float_vector.h"
#endif

namespace rxgpu::host {

#if defined(RXGPU_IN_TREE)
using reindexer::VectorMetric;
using reindexer::FloatVectorId;
using reindexer::ConstFloatVectorView;
using hnswlib::labeltype;
using hnswlib::tableint;
using hnswlib::SearchResultQueue;
template <class Q>
inline void ReserveQueue(Q&, size_t) noexcept {}   // hnswlib::PriorityQueue has no reserve()
#else

enum class VectorMetric { L2 = 0, InnerProduct = 1, Cosine = 2 };

using labeltype = uint64_t;
using tableint = uint32_t;

class FloatVectorId {
public:
	FloatVectorId(int32_t rowId, uint32_t arrayIdx) noexcept : value_{(uint64_t(uint32_t(rowId)) << 32) | arrayIdx} {}
	static FloatVectorId FromNumber(uint64_t v) noexcept {
		FloatVectorId id(0, 0);
		id.value_ = v;
		return id;
	}
	uint64_t AsNumber() const noexcept { return value_; }
	int32_t RowId() const noexcept { return int32_t(value_ >> 32); }
	uint32_t ArrayIndex() const noexcept { return uint32_t(value_ & 0xFFFFFFFFull); }

private:
	uint64_t value_;
};

class ConstFloatVectorView {
public:
	ConstFloatVectorView() noexcept = default;
	ConstFloatVectorView(const float* data, size_t dim) noexcept : data_(data), dim_(dim) {}
	const float* Data() const noexcept { return data_; }
	size_t Dimension() const noexcept { return dim_; }
	bool IsEmpty() const noexcept { return dim_ == 0; }

private:
	const float* data_ = nullptr;
	size_t dim_ = 0;
};
#endif   // RXGPU_IN_TREE

// Binary max-heap with the interface the reference's result consumers use (top/pop/size/empty/emplace/push,
// replace_top).  Ordering is supplied by Compare exactly like std::priority_queue.
template <class T, class Compare = std::less<T>>
class ResultHeap {
public:
	ResultHeap() = default;
	explicit ResultHeap(Compare c) : cmp_(std::move(c)) {}

	bool empty() const noexcept { return items_.empty(); }
	size_t size() const noexcept { return items_.size(); }
	const T& top() const noexcept { return items_.front(); }
	void reserve(size_t n) { items_.reserve(n); }

	void push(const T& v) {
		items_.push_back(v);
		bubbleUp(items_.size() - 1);
	}
	template <class... A>
	void emplace(A&&... a) {
		items_.emplace_back(std::forward<A>(a)...);
		bubbleUp(items_.size() - 1);
	}
	void pop() {
		items_.front() = std::move(items_.back());
		items_.pop_back();
		if (!items_.empty()) sinkDown(0);
	}
	T replace_top(T v) {
		T old = std::move(items_.front());
		items_.front() = std::move(v);
		sinkDown(0);
		return old;
	}
	void clear() noexcept { items_.clear(); }

private:
	void bubbleUp(size_t i) {
		while (i) {
			const size_t up = (i - 1) >> 1;
			if (!cmp_(items_[up], items_[i])) return;
			std::swap(items_[up], items_[i]);
			i = up;
		}
	}
	void sinkDown(size_t i) {
		const size_t n = items_.size();
		for (;;) {
			size_t big = i;
			const size_t l = 2 * i + 1, r = l + 1;
			if (l < n && cmp_(items_[big], items_[l])) big = l;
			if (r < n && cmp_(items_[big], items_[r])) big = r;
			if (big == i) return;
			std::swap(items_[big], items_[i]);
			i = big;
		}
	}
	std::vector<T> items_;
	Compare cmp_{};
};

#if !defined(RXGPU_IN_TREE)
using SearchResultQueue = ResultHeap<std::pair<float, labeltype>>;
template <class Q>
inline void ReserveQueue(Q& q, size_t n) { q.reserve(n); }
#endif

}  // namespace rxgpu::host
