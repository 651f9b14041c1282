// HopscotchIdSet — the ITERATION ORDER of the hash sets the reference keeps uint32 ids in, restated.
//
// The reference picks the slot a new point recycles as `*deleted_elements.begin()` (hnswalg.h:1410-1421) and walks its one- / two-hop
// candidate sets in hash-set order inside updatePoint (:1515-1570).  Both sets are tsl::hopscotch_sc_set<uint32_t> (vendored under
// cpp_src/vendor/hopscotch/hopscotch_hash.h) with std::hash (identity) and an std::set as overflow container:
//   deleted_elements   HashSetT<tableint>          hnswalg.h:196-197: NeighborhoodSize 30, mod_growth_policy<3/2>
//   sCand / sNeigh     reindexer::fast_hash_set    estl/fast_hash_set.h:10-13,55-76: NeighborhoodSize 62, prime_growth_policy
// both default-constructed (16 buckets requested, max load factor 0.95).  To build THE SAME graph after delete + insert cycles the product's
// builder has to recycle the same slot and meet the candidates in the same order, so the parts of hopscotch hashing that decide where a
// value sits are reproduced here operation for operation: the home bucket (growth policy), linear probing for an empty bucket
// (12 x NeighborhoodSize probes), swap_empty_bucket_closer, the overflow set and its use condition, rehash order (hopscotch_hash.h:1166-1213,
// 1355-1481).  Only what the reference's call sites use is provided: insert, erase(key), pop_front (= erase(begin())), iteration, copy.
// Pinned against the real engine link for link in tests/test_hnsw_builder.py (delete / re-insert cycles).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <set>
#include <stdexcept>
#include <vector>

namespace rxgpu::host {

// tsl::mod_growth_policy<std::ratio<3, 2>> (hopscotch_hash.h:121-147)
class ModGrowthPolicy {
public:
	explicit ModGrowthPolicy(size_t& minBucketCountInOut) noexcept {
		minBucketCountInOut = std::max<size_t>(2, minBucketCountInOut);
		bucketCount_ = minBucketCountInOut;
	}
	size_t BucketForHash(size_t hash) const noexcept { return hash % bucketCount_; }
	size_t NextBucketCount() const noexcept { return size_t(std::ceil(double(bucketCount_) * 1.5)); }

private:
	size_t bucketCount_;
};

// tsl::prime_growth_policy (hopscotch_hash.h:150-202)
class PrimeGrowthPolicy {
public:
	explicit PrimeGrowthPolicy(size_t& minBucketCountInOut) {
		const uint32_t* it = std::lower_bound(kPrimes, kPrimes + kPrimesCount, minBucketCountInOut);
		if (it == kPrimes + kPrimesCount) throw std::length_error("The map exceeds its maxmimum size.");
		iprime_ = unsigned(it - kPrimes);
		minBucketCountInOut = *it;
	}
	size_t BucketForHash(size_t hash) const noexcept { return hash % kPrimes[iprime_]; }
	size_t NextBucketCount() const {
		if (iprime_ + 1 >= kPrimesCount) throw std::length_error("The map exceeds its maxmimum size.");
		return kPrimes[iprime_ + 1];
	}

private:
	static constexpr unsigned kPrimesCount = 29;
	static constexpr uint32_t kPrimes[kPrimesCount] = {17u,		 37u,		79u,		 131u,		  257u,		   521u,		1031u,		  2053u,		6151u,		 12289u,
														24593u,	 49157u,	98317u,		 196613u,	  393241u,	   786433u,		1572869u,	  3145739u,		6291469u,	 12582917u,
														25165843u, 50331653u, 100663319u, 201326611u, 402653189u, 805306457u, 1610612741u, 3221225473u, 4294967291u};
	unsigned iprime_;
};

template <unsigned kNeighborhoodSize, class GrowthPolicy>
class HopscotchIdSet : private GrowthPolicy {
	static_assert(kNeighborhoodSize <= 62, "the neighbourhood bitmap is one 64-bit word");
	struct Bucket {
		uint64_t neighbors = 0;   // bit i: the bucket i places further holds a value whose home is this bucket
		uint32_t value = 0;
		bool full = false;
		bool overflow = false;    // some value whose home is this bucket lives in the overflow set
	};
	static constexpr size_t kInitBuckets = 16;                       // DEFAULT_INIT_BUCKETS_SIZE
	static constexpr float kMaxLoadFactor = 0.95f;                   // DEFAULT_MAX_LOAD_FACTOR
	static constexpr float kMinLoadFactorForRehash = 0.1f;           // MIN_LOAD_FACTOR_FOR_REHASH
	static constexpr size_t kMaxProbes = 12 * size_t(kNeighborhoodSize);   // MAX_PROBES_FOR_EMPTY_BUCKET

public:
	HopscotchIdSet() : HopscotchIdSet(kInitBuckets) {}

	bool empty() const noexcept { return nbElements_ == 0; }
	size_t size() const noexcept { return nbElements_; }
	size_t count(uint32_t key) const noexcept { return findBucket(key, GrowthPolicy::BucketForHash(key)) != kNone || overflow_.count(key) ? 1 : 0; }

	// the first element in iteration order: buckets in array order, then the overflow set (ascending)
	uint32_t front() const {
		for (const Bucket& b : buckets_) {
			if (b.full) return b.value;
		}
		if (overflow_.empty()) throw std::logic_error("HopscotchIdSet::front on an empty set");
		return *overflow_.begin();
	}
	// erase(begin()) (hopscotch_hash.h:904-916)
	uint32_t pop_front() {
		const uint32_t v = front();
		erase(v);
		return v;
	}
	template <typename F>
	void for_each(F&& f) const {
		for (const Bucket& b : buckets_) {
			if (b.full) f(b.value);
		}
		for (uint32_t v : overflow_) f(v);
	}

	bool insert(uint32_t key) {
		const size_t home = GrowthPolicy::BucketForHash(key);
		if (findBucket(key, home) != kNone) return false;
		if (buckets_[home].overflow && overflow_.count(key)) return false;
		insertInternal(key, home);
		return true;
	}

	size_t erase(uint32_t key) {
		const size_t home = GrowthPolicy::BucketForHash(key);
		const size_t at = findBucket(key, home);
		if (at != kNone) {
			eraseFromBucket(at, home);
			return 1;
		}
		if (buckets_[home].overflow) {
			auto it = overflow_.find(key);
			if (it != overflow_.end()) {
				overflow_.erase(it);
				nbElements_--;
				bool other = false;   // erase_from_overflow: the flag stays while another overflow value shares the home bucket
				for (uint32_t v : overflow_) other = other || GrowthPolicy::BucketForHash(v) == home;
				if (!other) buckets_[home].overflow = false;
				return 1;
			}
		}
		return 0;
	}

private:
	static constexpr size_t kNone = ~size_t(0);

	explicit HopscotchIdSet(size_t bucketCount) : GrowthPolicy(bucketCount), bucketCount_(bucketCount) {
		buckets_.resize(bucketCount_ + kNeighborhoodSize - 1);
		loadThreshold_ = size_t(float(bucketCount_) * kMaxLoadFactor);
	}

	size_t findBucket(uint32_t key, size_t home) const noexcept {
		uint64_t infos = buckets_[home].neighbors;
		for (size_t i = home; infos != 0; ++i, infos >>= 1) {
			if ((infos & 1) && buckets_[i].value == key) return i;
		}
		return kNone;
	}
	void eraseFromBucket(size_t at, size_t home) noexcept {
		buckets_[at].full = false;
		buckets_[home].neighbors ^= uint64_t(1) << (at - home);
		nbElements_--;
	}
	size_t findEmptyBucket(size_t start) const noexcept {
		const size_t limit = std::min(start + kMaxProbes, buckets_.size());
		for (; start < limit; ++start) {
			if (!buckets_[start].full) return start;
		}
		return buckets_.size();
	}
	// hopscotch_hash.h:1451-1481
	bool swapEmptyBucketCloser(size_t& emptyInOut) noexcept {
		const size_t neighborhoodStart = emptyInOut - kNeighborhoodSize + 1;
		for (size_t toCheck = neighborhoodStart; toCheck < emptyInOut; ++toCheck) {
			uint64_t infos = buckets_[toCheck].neighbors;
			size_t toSwap = toCheck;
			while (infos != 0 && toSwap < emptyInOut) {
				if (infos & 1) {
					buckets_[emptyInOut].value = buckets_[toSwap].value;
					buckets_[emptyInOut].full = true;
					buckets_[toSwap].full = false;
					buckets_[toCheck].neighbors ^= uint64_t(1) << (emptyInOut - toCheck);
					buckets_[toCheck].neighbors ^= uint64_t(1) << (toSwap - toCheck);
					emptyInOut = toSwap;
					return true;
				}
				++toSwap;
				infos >>= 1;
			}
		}
		return false;
	}
	bool willNeighborhoodChangeOnRehash(size_t check) const {
		size_t expandCount = GrowthPolicy::NextBucketCount();
		GrowthPolicy expand(expandCount);
		for (size_t i = check; i < buckets_.size() && (i - check) < kNeighborhoodSize; ++i) {
			const size_t hash = buckets_[i].value;
			if (GrowthPolicy::BucketForHash(hash) != expand.BucketForHash(hash)) return true;
		}
		return false;
	}
	// hopscotch_hash.h:1355-1390
	void insertInternal(uint32_t key, size_t home) {
		if ((nbElements_ - overflow_.size() + 1) > loadThreshold_) {
			rehashInternal(GrowthPolicy::NextBucketCount());
			home = GrowthPolicy::BucketForHash(key);
		}
		size_t empty = findEmptyBucket(home);
		if (empty < buckets_.size()) {
			do {
				if (empty - home < kNeighborhoodSize) {
					buckets_[empty].value = key;
					buckets_[empty].full = true;
					buckets_[home].neighbors ^= uint64_t(1) << (empty - home);
					nbElements_++;
					return;
				}
			} while (swapEmptyBucketCloser(empty));
		}
		const float loadF = float(nbElements_) / float(bucketCount_);
		if (loadF < kMaxLoadFactor && (loadF < kMinLoadFactorForRehash || !willNeighborhoodChangeOnRehash(home))) {
			overflow_.insert(key);
			buckets_[home].overflow = true;
			nbElements_++;
			return;
		}
		rehashInternal(GrowthPolicy::NextBucketCount());
		insertInternal(key, GrowthPolicy::BucketForHash(key));
	}
	// hopscotch_hash.h:1166-1213: a fresh table of `count` buckets; the overflow set moves over as overflow, the bucket values are re-inserted in
	// array order
	void rehashInternal(size_t count) {
		HopscotchIdSet next(count);
		if (!overflow_.empty()) {
			next.overflow_.swap(overflow_);
			next.nbElements_ += next.overflow_.size();
			for (uint32_t v : next.overflow_) next.buckets_[next.BucketForHashOf(v)].overflow = true;
		}
		for (size_t i = 0; i < buckets_.size(); ++i) {
			if (!buckets_[i].full) continue;
			const uint32_t v = buckets_[i].value;
			next.insertInternal(v, next.BucketForHashOf(v));
		}
		*this = std::move(next);
	}
	size_t BucketForHashOf(uint32_t v) const noexcept { return GrowthPolicy::BucketForHash(v); }

	std::vector<Bucket> buckets_;
	std::set<uint32_t> overflow_;
	size_t bucketCount_ = 0;
	size_t nbElements_ = 0;
	size_t loadThreshold_ = 0;
};

using DeletedIdSet = HopscotchIdSet<30, ModGrowthPolicy>;       // HierarchicalNSWImpl::deleted_elements
using CandidateIdSet = HopscotchIdSet<62, PrimeGrowthPolicy>;   // reindexer::fast_hash_set<tableint> in updatePoint

}  // namespace rxgpu::host
