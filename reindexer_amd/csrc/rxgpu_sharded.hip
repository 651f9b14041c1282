// Row-range sharding of one float_vector index over several GPUs of THIS process (BASELINE configs[3] behind the C++ seam: the Map owns a
// device list).  Every shard is an ordinary rxgpu_index on its own device; a search fans the query out, every shard runs the kernels of
// rxgpu_search_* on its rows, and the per-shard answers (kk x 8 B per query) are merged under the reference's order — (dist, GLOBAL row),
// global row = shard * shard_rows + local row, which is the scan order of BruteforceSearch::SearchKnn (bruteforce.cc:103-127) because shard
// s holds the rows [s * shard_rows, (s + 1) * shard_rows).
//
// SearchKnn (kk <= 64, the planner's case) exchanges the lists over RCCL: the index keeps ONE communicator over its distinct devices
// (ncclCommInitAll), every device's scans write their sorted lists straight into that device's send buffer, one ncclAllGather per query
// batch (inside ncclGroupStart/End, on the shards' streams: kk x 8 B x nq per shard over xGMI), knn_merge_shards on device 0, ONE D2H copy.
// One host thread enqueues everything; nothing is merged on the host.  RXGPU_SHARD_MERGE=host keeps the former path (a worker thread per
// shard, D2H per shard, host merge), which also serves range searches, pre-filtered searches, kk > 64 and shards holding fewer than kk rows.
// The one-process-per-GPU deployment (torch.distributed over the same RCCL) is reindexer_amd/sharded.py.
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <exception>
#include <functional>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include <vector>

#include <dlfcn.h>

#include <hip/hip_runtime.h>
#include "rccl_dyn.h"   // <rccl/rccl.h> for types and prototypes only: the library is opened at the first sharded index (rccl_api below)

#include "../../include/rxgpu.h"
#include "knn_kernels.hip.h"
#include "rxgpu_internal.h"

using rxgpu::set_error;

namespace rxgpu {

const RcclApi& rccl_api() {
	static RcclApi api;
	static std::once_flag once;
	std::call_once(once, [] {
		void* lib = nullptr;
		std::string tried;
		const char* env = std::getenv("RXGPU_RCCL_LIB");   // an explicit path (tests use it to provoke the fallback)
		const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
		for (const char* n : names) {
			if (!n || !*n) continue;
			lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
			if (lib) break;
			const char* e = dlerror();
			tried += std::string(tried.empty() ? "" : "; ") + n + ": " + (e ? e : "?");
			if (n == env) break;   // an explicit choice is not second-guessed
		}
		if (!lib) {
			api.why = "librccl.so could not be opened (" + tried + ")";
			return;
		}
		auto sym = [&](const char* name) -> void* {
			void* p = dlsym(lib, name);
			if (!p && api.why.empty()) api.why = std::string("librccl.so lacks ") + name;
			return p;
		};
		api.ncclCommInitAll = reinterpret_cast<decltype(api.ncclCommInitAll)>(sym("ncclCommInitAll"));
		api.ncclAllGather = reinterpret_cast<decltype(api.ncclAllGather)>(sym("ncclAllGather"));
		api.ncclGroupStart = reinterpret_cast<decltype(api.ncclGroupStart)>(sym("ncclGroupStart"));
		api.ncclGroupEnd = reinterpret_cast<decltype(api.ncclGroupEnd)>(sym("ncclGroupEnd"));
		api.ncclGetErrorString = reinterpret_cast<decltype(api.ncclGetErrorString)>(sym("ncclGetErrorString"));
	});
	return api;
}

std::shared_ptr<RcclCommSet> rccl_comm_set(const std::vector<int>& devices, std::string* why) {
	static std::mutex pool_mtx;
	static std::vector<std::shared_ptr<RcclCommSet>>* pool = new std::vector<std::shared_ptr<RcclCommSet>>();   // never torn down: no RCCL calls at exit
	const RcclApi& api = rccl_api();
	if (!api.why.empty()) {
		if (why) *why = "RCCL unavailable: " + api.why;
		return nullptr;
	}
	std::lock_guard<std::mutex> lk(pool_mtx);
	for (const auto& cs : *pool) {
		if (cs->devices == devices) return cs;
	}
	auto cs = std::make_shared<RcclCommSet>();
	cs->devices = devices;
	cs->comms.assign(devices.size(), nullptr);
	int prev = -1;
	(void)hipGetDevice(&prev);
	const ncclResult_t nr = api.ncclCommInitAll(cs->comms.data(), int(devices.size()), devices.data());
	if (prev >= 0) (void)hipSetDevice(prev);
	if (nr != ncclSuccess) {
		if (why) *why = std::string("ncclCommInitAll over ") + std::to_string(devices.size()) + " device(s): " + api.ncclGetErrorString(nr);
		return nullptr;
	}
	pool->push_back(cs);
	return cs;
}

// A small pool of worker threads per shard (the shard's device stays current on them) behind one job queue: fan-outs of concurrent
// callers queue up per shard and overlap — the single-device entry points are re-entrant (a search context and stream per call) — so
// no lock is held across a fan-out.  Searches share `call_mtx`, mutations take it exclusively (the reference's namespace lock above us
// does the same; this one only protects direct C-ABI callers).
struct ShardWorker {
	static constexpr unsigned kThreads = 4;
	std::vector<std::thread> threads;
	std::mutex mtx;
	std::condition_variable cv;
	std::deque<std::function<void()>> jobs;
	bool stop = false;

	void run() {
		std::unique_lock<std::mutex> lk(mtx);
		for (;;) {
			cv.wait(lk, [&] { return !jobs.empty() || stop; });
			if (jobs.empty()) return;   // stop requested and nothing left
			auto fn = std::move(jobs.front());
			jobs.pop_front();
			lk.unlock();
			fn();
			lk.lock();
		}
	}
	void post(std::function<void()> fn) {
		{
			std::lock_guard<std::mutex> lk(mtx);
			jobs.push_back(std::move(fn));
		}
		cv.notify_one();
	}
};

// Buffers and streams of one exchange in flight (per distinct device = RCCL rank); lanes are pooled so that concurrent callers overlap.
struct ExchangeLane {
	std::vector<hipStream_t> stream;
	std::vector<rxgpu_devbuf> d_queries, d_local, d_gathered;
	rxgpu_devbuf d_out;            // rank 0: [nq][kk] distances | [nq][kk] global rows | [nq] counts
	void* h_pinned = nullptr;      // queries on the way in, the merged lists on the way out
	size_t h_pinned_bytes = 0;
};

// The RCCL side of a sharded index: one rank per DISTINCT device; a device that holds several shards (the 1-GPU test box lists the same
// device several times) sends them as `slots` consecutive lists, devices with fewer shards pad (slot base = kInvalidRow, skipped by the merge).
struct ShardExchange {
	uint32_t nranks = 0, slots = 0;
	std::vector<int> rank_dev;
	std::shared_ptr<RcclCommSet> cs;   // the process-wide communicators over rank_dev (rccl_dyn.h)
	std::vector<uint32_t> shard_rank, shard_slot;
	uint32_t* d_slot_base = nullptr;   // on rank_dev[0]: global row base of every gathered position
	std::mutex mtx;                    // lane pool
	std::vector<ExchangeLane*> free_lanes;
	std::atomic<uint64_t> collectives{0};
};

struct ShardSet {
	std::vector<rxgpu_index*> shards;
	std::vector<ShardWorker*> workers;
	uint64_t shard_rows = 0;
	std::shared_mutex call_mtx;   // shared: searches (any number of fan-outs in flight); exclusive: uploads, moves, truncation
	ShardExchange* xch = nullptr; // null: the host-merge path (RXGPU_SHARD_MERGE=host, or RCCL is not available: merge_note says why)
	std::string merge_note;
};

namespace {

// completion latch of one fan-out (lives on the caller's stack)
struct FanOut {
	std::mutex m;
	std::condition_variable cv;
	size_t left;
	explicit FanOut(size_t n) : left(n) {}
	void arrive() {
		std::lock_guard<std::mutex> lk(m);
		if (--left == 0) cv.notify_all();
	}
	void wait() {
		std::unique_lock<std::mutex> lk(m);
		cv.wait(lk, [&] { return left == 0; });
	}
};

// runs fn on the worker pool of shard s and waits; an exception inside (bad_alloc while staging) becomes RXGPU_ERR_NOMEM
int run_on_shard(ShardSet* ss, size_t s, const std::function<int()>& fn, std::string& err) {
	int rc = RXGPU_OK;
	FanOut done(1);
	ss->workers[s]->post([&] {
		try {
			rc = fn();
			if (rc != RXGPU_OK) err = rxgpu_last_error();   // thread-local on the worker: carry it over
		} catch (const std::exception& e) {
			rc = RXGPU_ERR_NOMEM;
			err = e.what();
		}
		done.arrive();
	});
	done.wait();
	return rc;
}

// runs fn(s) for every shard concurrently; returns the first non-OK code (the message of that shard is kept)
int for_each_shard(ShardSet* ss, const std::function<int(size_t)>& fn) {
	const size_t n = ss->shards.size();
	std::vector<int> rc(n, RXGPU_OK);
	std::vector<std::string> err(n);
	FanOut done(n);
	for (size_t s = 0; s < n; ++s) {
		ss->workers[s]->post([&, s] {
			try {
				rc[s] = fn(s);
				if (rc[s] != RXGPU_OK) err[s] = rxgpu_last_error();
			} catch (const std::exception& e) {
				rc[s] = RXGPU_ERR_NOMEM;
				err[s] = e.what();
			}
			done.arrive();
		});
	}
	done.wait();
	for (size_t s = 0; s < n; ++s) {
		if (rc[s] != RXGPU_OK) {
			set_error("shard " + std::to_string(s) + ": " + err[s]);
			return rc[s];
		}
	}
	return RXGPU_OK;
}

// (dist, global row) as a strict weak order even when a distance is NaN (NaN sorts last): the comparator of the large-k path
inline bool dist_row_less(const std::pair<float, uint32_t>& a, const std::pair<float, uint32_t>& b) {
	const bool an = a.first != a.first, bn = b.first != b.first;
	if (an != bn) return bn;
	if (!an && a.first != b.first) return a.first < b.first;
	return a.second < b.second;
}

uint64_t local_count(const ShardSet* ss, size_t s, uint64_t count) {
	const uint64_t lo = uint64_t(s) * ss->shard_rows;
	return count > lo ? std::min<uint64_t>(count - lo, ss->shard_rows) : 0;
}

}  // namespace


#define SH_HIP(expr)                                                                       \
	do {                                                                                   \
		hipError_t e__ = (expr);                                                           \
		if (e__ != hipSuccess) {                                                           \
			set_error(std::string(#expr) + ": " + hipGetErrorString(e__));                 \
			return e__ == hipErrorOutOfMemory ? RXGPU_ERR_NOMEM : RXGPU_ERR_DEVICE;        \
		}                                                                                  \
	} while (0)
#define SH_NCCL(expr)                                                                      \
	do {                                                                                   \
		ncclResult_t r__ = (expr);                                                         \
		if (r__ != ncclSuccess) {                                                          \
			set_error(std::string(#expr) + ": " + rccl_api().ncclGetErrorString(r__));     \
			return RXGPU_ERR_DEVICE;                                                       \
		}                                                                                  \
	} while (0)

namespace {

struct CurrentDevice {   // restores the caller's device
	int prev = -1;
	CurrentDevice() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
	~CurrentDevice() { if (prev >= 0) (void)hipSetDevice(prev); }
};

void free_lane(ShardExchange* x, ExchangeLane* l) {
	for (uint32_t r = 0; r < x->nranks; ++r) {
		(void)hipSetDevice(x->rank_dev[r]);
		if (r < l->stream.size() && l->stream[r]) {
			(void)hipStreamSynchronize(l->stream[r]);
			(void)hipStreamDestroy(l->stream[r]);
		}
		if (r < l->d_queries.size()) l->d_queries[r].release();
		if (r < l->d_local.size()) l->d_local[r].release();
		if (r < l->d_gathered.size()) l->d_gathered[r].release();
	}
	(void)hipSetDevice(x->rank_dev[0]);
	l->d_out.release();
	if (l->h_pinned) (void)hipHostFree(l->h_pinned);
	delete l;
}

int new_lane(ShardExchange* x, ExchangeLane** out) {
	auto* l = new ExchangeLane();
	l->stream.assign(x->nranks, nullptr);
	l->d_queries.resize(x->nranks);
	l->d_local.resize(x->nranks);
	l->d_gathered.resize(x->nranks);
	for (uint32_t r = 0; r < x->nranks; ++r) {
		hipError_t e = hipSetDevice(x->rank_dev[r]);
		if (e == hipSuccess) e = hipStreamCreateWithFlags(&l->stream[r], hipStreamNonBlocking);
		if (e != hipSuccess) {
			set_error(std::string("sharded exchange: stream on device ") + std::to_string(x->rank_dev[r]) + ": " + hipGetErrorString(e));
			free_lane(x, l);
			return RXGPU_ERR_DEVICE;
		}
	}
	*out = l;
	return RXGPU_OK;
}

void exchange_destroy(ShardExchange* x) {
	if (!x) return;
	CurrentDevice cd;
	for (ExchangeLane* l : x->free_lanes) free_lane(x, l);
	if (x->d_slot_base) {
		(void)hipSetDevice(x->rank_dev[0]);
		(void)hipFree(x->d_slot_base);
	}
	delete x;
}

// One communicator over the distinct devices of the shard list, and the position -> global row base table of the merge kernel.
int exchange_create(ShardSet* ss, uint32_t n_devices, const int* devices, ShardExchange** out) {
	auto* x = new ShardExchange();
	x->shard_rank.resize(n_devices);
	x->shard_slot.resize(n_devices);
	std::vector<uint32_t> per_rank;
	for (uint32_t s = 0; s < n_devices; ++s) {
		uint32_t r = 0;
		while (r < x->rank_dev.size() && x->rank_dev[r] != devices[s]) ++r;
		if (r == x->rank_dev.size()) {
			x->rank_dev.push_back(devices[s]);
			per_rank.push_back(0);
		}
		x->shard_rank[s] = r;
		x->shard_slot[s] = per_rank[r]++;
	}
	x->nranks = uint32_t(x->rank_dev.size());
	x->slots = *std::max_element(per_rank.begin(), per_rank.end());
	CurrentDevice cd;
	std::string why;
	x->cs = rccl_comm_set(x->rank_dev, &why);
	if (!x->cs) {
		set_error(why);
		delete x;
		return RXGPU_ERR_DEVICE;
	}
	std::vector<uint32_t> base(size_t(x->nranks) * x->slots, kInvalidRow);
	for (uint32_t s = 0; s < n_devices; ++s) base[size_t(x->shard_rank[s]) * x->slots + x->shard_slot[s]] = uint32_t(s * ss->shard_rows);
	hipError_t e = hipSetDevice(x->rank_dev[0]);
	if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&x->d_slot_base), base.size() * sizeof(uint32_t));
	if (e == hipSuccess) e = hipMemcpy(x->d_slot_base, base.data(), base.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
	if (e != hipSuccess) {
		set_error(std::string("rxgpu_index_create_sharded: slot table: ") + hipGetErrorString(e));
		exchange_destroy(x);
		return RXGPU_ERR_DEVICE;
	}
	*out = x;
	return RXGPU_OK;
}

// The second half of every exchange: the per-shard lists sit in d_local of their rank's device (on that rank's lane stream) -> ONE
// ncclAllGather per query batch -> knn_merge_shards on the first device -> one D2H copy of the merged lists.
int exchange_gather_merge(ShardSet* ss, ExchangeLane* l, uint32_t nq, uint32_t kk, bool sorted, float* out_dist, uint32_t* out_row, uint32_t* out_count) {
	ShardExchange* x = ss->xch;
	const size_t list_words = size_t(2) * nq * kk;
	const size_t out_bytes = (size_t(2) * nq * kk + nq) * sizeof(uint32_t);
	{
		std::lock_guard<std::mutex> lk(x->cs->mtx);
		const RcclApi& api = rccl_api();
		SH_NCCL(api.ncclGroupStart());
		for (uint32_t r = 0; r < x->nranks; ++r) {
			const ncclResult_t nr = api.ncclAllGather(l->d_local[r].ptr, l->d_gathered[r].ptr, list_words * x->slots, ncclUint32, x->cs->comms[r], l->stream[r]);
			if (nr != ncclSuccess) {
				(void)api.ncclGroupEnd();
				set_error(std::string("ncclAllGather: ") + api.ncclGetErrorString(nr));
				return RXGPU_ERR_DEVICE;
			}
		}
		SH_NCCL(api.ncclGroupEnd());
		x->collectives.fetch_add(1, std::memory_order_relaxed);
	}
	SH_HIP(hipSetDevice(x->rank_dev[0]));
	if (int rc = l->d_out.ensure(out_bytes); rc) return rc;
	float* d_od = static_cast<float*>(l->d_out.ptr);
	uint32_t* d_or = static_cast<uint32_t*>(l->d_out.ptr) + size_t(nq) * kk;
	uint32_t* d_oc = d_or + size_t(nq) * kk;
	launch_merge_shards(static_cast<const uint32_t*>(l->d_gathered[0].ptr), x->nranks * x->slots, nq, kk, 0, d_od, d_or, d_oc, l->stream[0], x->d_slot_base, sorted);
	SH_HIP(hipGetLastError());
	SH_HIP(hipMemcpyAsync(l->h_pinned, l->d_out.ptr, out_bytes, hipMemcpyDeviceToHost, l->stream[0]));
	for (uint32_t r = 0; r < x->nranks; ++r) SH_HIP(hipStreamSynchronize(l->stream[r]));
	const auto* hp = static_cast<const uint32_t*>(l->h_pinned);
	std::memcpy(out_dist, hp, size_t(nq) * kk * sizeof(float));
	std::memcpy(out_row, hp + size_t(nq) * kk, size_t(nq) * kk * sizeof(uint32_t));
	std::memcpy(out_count, hp + size_t(2) * nq * kk, size_t(nq) * sizeof(uint32_t));
	return RXGPU_OK;
}

int ensure_pinned(ExchangeLane* l, size_t pin) {
	if (l->h_pinned_bytes < pin) {
		if (l->h_pinned) (void)hipHostFree(l->h_pinned);
		l->h_pinned = nullptr;
		l->h_pinned_bytes = 0;
		SH_HIP(hipHostMalloc(&l->h_pinned, pin, hipHostMallocDefault));
		l->h_pinned_bytes = pin;
	}
	return RXGPU_OK;
}

// SearchKnn over every shard with the exchange on the devices.  The caller holds call_mtx shared and has checked the shape
// (kk <= kMaxFusedK, every non-empty shard holds >= kk rows).
int exchange_search_knn(rxgpu_index* h, ShardSet* ss, ExchangeLane* l, const float* queries, uint32_t nq, uint32_t kk, float* out_dist,
						uint32_t* out_row, uint32_t* out_count) {
	ShardExchange* x = ss->xch;
	const size_t qbytes = size_t(nq) * h->dim * sizeof(float);
	const size_t list_words = size_t(2) * nq * kk;                 // one shard: [nq][kk] distances | [nq][kk] local rows
	const size_t local_bytes = list_words * x->slots * sizeof(uint32_t);
	const size_t out_bytes = (size_t(2) * nq * kk + nq) * sizeof(uint32_t);
	if (int rc = ensure_pinned(l, std::max(qbytes, out_bytes)); rc) return rc;
	std::memcpy(l->h_pinned, queries, qbytes);
	bool hole = x->slots * x->nranks != ss->shards.size();   // padded positions are skipped by their base; EMPTY shards need invalid lists
	std::vector<uint64_t> lc(ss->shards.size());
	for (size_t s = 0; s < ss->shards.size(); ++s) {
		lc[s] = rxgpu_index_count(ss->shards[s]);
		hole = hole || lc[s] == 0;
	}
	for (uint32_t r = 0; r < x->nranks; ++r) {
		SH_HIP(hipSetDevice(x->rank_dev[r]));
		if (int rc = l->d_queries[r].ensure(qbytes); rc) return rc;
		if (int rc = l->d_local[r].ensure(local_bytes); rc) return rc;
		if (int rc = l->d_gathered[r].ensure(local_bytes * x->nranks); rc) return rc;
		SH_HIP(hipMemcpyAsync(l->d_queries[r].ptr, l->h_pinned, qbytes, hipMemcpyHostToDevice, l->stream[r]));
		if (hole) SH_HIP(hipMemsetAsync(l->d_local[r].ptr, 0xFF, local_bytes, l->stream[r]));   // rows = kInvalidRow
		for (size_t s = 0; s < ss->shards.size(); ++s) {
			if (x->shard_rank[s] != r || lc[s] == 0) continue;
			uint32_t* dst = static_cast<uint32_t*>(l->d_local[r].ptr) + list_words * x->shard_slot[s];
			if (int rc = rxgpu_search_knn_device(ss->shards[s], l->d_queries[r].ptr, nq, kk, dst, dst + size_t(nq) * kk, nullptr, l->stream[r]); rc) return rc;
		}
	}
	return exchange_gather_merge(ss, l, nq, kk, true, out_dist, out_row, out_count);
}

// HNSW SearchKnn over every shard's own graph (SURVEY 8e "HNSW"): the searches run concurrently on the shards' worker threads — each is the
// single-device search with its re-run tiers, driven from the host by the counts alone — and leave their lists in HBM, packed into the
// shard's slot of the send buffer (HnswSink); then the same all-gather + merge as brute force (the lists are unordered sets: sorted = false).
int exchange_hnsw_search_knn(rxgpu_index* h, ShardSet* ss, ExchangeLane* l, const void* queries, const float* qcorr, const float* qnorm, uint32_t nq, uint32_t k, uint32_t ef, float* out_dist,
							 uint32_t* out_row, uint32_t* out_count) {
	(void)h;
	ShardExchange* x = ss->xch;
	const size_t list_words = size_t(2) * nq * k;
	const size_t local_bytes = list_words * x->slots * sizeof(uint32_t);
	const size_t out_bytes = (size_t(2) * nq * k + nq) * sizeof(uint32_t);
	if (int rc = ensure_pinned(l, out_bytes); rc) return rc;
	for (uint32_t r = 0; r < x->nranks; ++r) {
		SH_HIP(hipSetDevice(x->rank_dev[r]));
		if (int rc = l->d_local[r].ensure(local_bytes); rc) return rc;
		if (int rc = l->d_gathered[r].ensure(local_bytes * x->nranks); rc) return rc;
		SH_HIP(hipMemsetAsync(l->d_local[r].ptr, 0xFF, local_bytes, l->stream[r]));   // empty shards and padded slots: rows = kInvalidRow
		SH_HIP(hipStreamSynchronize(l->stream[r]));                                    // the searches write from their own streams
	}
	const int rc = for_each_shard(ss, [&](size_t s) -> int {
		if (rxgpu_index_count(ss->shards[s]) == 0) return RXGPU_OK;
		uint32_t* dst = static_cast<uint32_t*>(l->d_local[x->shard_rank[s]].ptr) + list_words * x->shard_slot[s];
		return hnsw_search_to_sink(ss->shards[s], queries, qcorr, qnorm, nq, k, ef, HnswSink{dst, dst + size_t(nq) * k, k});
	});
	if (rc != RXGPU_OK) return rc;
	return exchange_gather_merge(ss, l, nq, k, false, out_dist, out_row, out_count);
}

ExchangeLane* take_lane(ShardExchange* x) {
	std::lock_guard<std::mutex> pl(x->mtx);
	if (x->free_lanes.empty()) return nullptr;
	ExchangeLane* lane = x->free_lanes.back();
	x->free_lanes.pop_back();
	return lane;
}

void give_lane(ShardExchange* x, ExchangeLane* lane, int rc) {
	if (rc != RXGPU_OK) {   // streams may hold half an exchange: drain before the lane is reused
		for (uint32_t r = 0; r < x->nranks; ++r) {
			(void)hipSetDevice(x->rank_dev[r]);
			(void)hipStreamSynchronize(lane->stream[r]);
		}
	}
	std::lock_guard<std::mutex> pl(x->mtx);
	x->free_lanes.push_back(lane);
}

}  // namespace

void sharded_destroy(rxgpu_index* h) {
	ShardSet* ss = h->shard_set;
	if (!ss) return;
	exchange_destroy(ss->xch);
	ss->xch = nullptr;
	for (ShardWorker* w : ss->workers) {
		{
			std::lock_guard<std::mutex> lk(w->mtx);
			w->stop = true;
			w->cv.notify_all();
		}
		for (std::thread& t : w->threads) {
			if (t.joinable()) t.join();
		}
		delete w;
	}
	for (rxgpu_index* s : ss->shards) rxgpu_index_destroy(s);
	delete ss;
	h->shard_set = nullptr;
}

uint64_t sharded_device_bytes(const rxgpu_index* h) {
	uint64_t b = 0;
	for (const rxgpu_index* s : h->shard_set->shards) b += rxgpu_index_device_bytes(s);
	return b;
}

int sharded_upload_rows(rxgpu_index* h, uint64_t first_row, uint64_t n, const float* rows, const float* inv_norms) {
	ShardSet* ss = h->shard_set;
	if (n == 0) return RXGPU_OK;
	if (!rows) {
		set_error("rxgpu_index_upload_rows: rows is null");
		return RXGPU_ERR_PARAMS;
	}
	if (first_row > h->count || first_row + n > h->capacity) {
		set_error(first_row > h->count ? "rxgpu_index_upload_rows: a sharded index is filled without holes (first_row <= count)"
									   : "The number of elements exceeds the specified limit");
		return RXGPU_ERR_PARAMS;
	}
	std::unique_lock<std::shared_mutex> lk(ss->call_mtx);
	const int rc = for_each_shard(ss, [&](size_t s) -> int {
		const uint64_t lo = uint64_t(s) * ss->shard_rows, hi = lo + ss->shard_rows;
		const uint64_t a = std::max(first_row, lo), b = std::min(first_row + n, hi);
		if (a >= b) return RXGPU_OK;
		return rxgpu_index_upload_rows(ss->shards[s], a - lo, b - a, rows + (a - first_row) * h->dim, inv_norms ? inv_norms + (a - first_row) : nullptr);
	});
	if (rc == RXGPU_OK) h->count = std::max(h->count, first_row + n);
	return rc;
}

int sharded_truncate(rxgpu_index* h, uint64_t count) {
	ShardSet* ss = h->shard_set;
	if (count > h->count) {
		set_error("rxgpu_index_truncate: count exceeds the number of rows");
		return RXGPU_ERR_PARAMS;
	}
	std::unique_lock<std::shared_mutex> lk(ss->call_mtx);
	const int rc = for_each_shard(ss, [&](size_t s) -> int { return rxgpu_index_truncate(ss->shards[s], local_count(ss, s, count)); });
	if (rc == RXGPU_OK) h->count = count;
	return rc;
}

// exact top-kk of every query under (dist, global row): per-shard exact top-kk lists merged on the host
int sharded_search_knn_impl(rxgpu_index* h, const float* queries, uint32_t nq, uint32_t kk, const uint32_t* row_ids, uint64_t n_ids, float* out_dist,
							uint32_t* out_row, uint32_t* out_count) {
	ShardSet* ss = h->shard_set;
	const size_t ns = ss->shards.size();
	std::vector<std::vector<float>> sd(ns);
	std::vector<std::vector<uint32_t>> sr(ns), sc(ns);
	// a row list (pre-filtered search) is split at the shard boundaries; local ids = global - shard base
	std::vector<std::vector<uint32_t>> local_ids(ns);
	if (row_ids) {
		for (uint64_t i = 0; i < n_ids; ++i) {
			if (i && row_ids[i] <= row_ids[i - 1]) {
				set_error("rxgpu_search_knn_subset: row ids must be strictly increasing");
				return RXGPU_ERR_PARAMS;
			}
			if (row_ids[i] >= h->count) {
				set_error("rxgpu_search_knn_subset: row id out of range");
				return RXGPU_ERR_PARAMS;
			}
			const size_t s = size_t(row_ids[i] / ss->shard_rows);
			local_ids[s].push_back(uint32_t(row_ids[i] - s * ss->shard_rows));
		}
	}
	std::shared_lock<std::shared_mutex> lk(ss->call_mtx);
	if (ss->xch && !row_ids && kk <= uint32_t(kMaxFusedK)) {
		bool fits = true;   // a shard with fewer rows than kk returns a shorter list: the host path pads it
		for (rxgpu_index* sh : ss->shards) {
			const uint64_t c = rxgpu_index_count(sh);
			fits = fits && (c == 0 || c >= kk);
		}
		if (fits) {
			ShardExchange* x = ss->xch;
			CurrentDevice cd;
			ExchangeLane* lane = take_lane(x);
			if (!lane) {
				if (int rc = new_lane(x, &lane); rc) return rc;
			}
			const int rc = exchange_search_knn(h, ss, lane, queries, nq, kk, out_dist, out_row, out_count);
			give_lane(x, lane, rc);
			return rc;
		}
	}
	const int rc = for_each_shard(ss, [&](size_t s) -> int {
		sd[s].assign(size_t(nq) * kk, 0.f);
		sr[s].assign(size_t(nq) * kk, 0u);
		sc[s].assign(nq, 0u);
		if (row_ids) {
			if (local_ids[s].empty()) return RXGPU_OK;
			return rxgpu_search_knn_subset(ss->shards[s], queries, nq, kk, local_ids[s].data(), local_ids[s].size(), sd[s].data(), sr[s].data(), sc[s].data());
		}
		if (rxgpu_index_count(ss->shards[s]) == 0) return RXGPU_OK;
		return rxgpu_search_knn(ss->shards[s], queries, nq, kk, sd[s].data(), sr[s].data(), sc[s].data());
	});
	if (rc != RXGPU_OK) return rc;
	std::vector<std::pair<float, uint32_t>> all;
	for (uint32_t q = 0; q < nq; ++q) {
		all.clear();
		for (size_t s = 0; s < ns; ++s) {
			for (uint32_t j = 0; j < sc[s][q]; ++j) all.emplace_back(sd[s][size_t(q) * kk + j], uint32_t(sr[s][size_t(q) * kk + j] + s * ss->shard_rows));
		}
		const size_t take = std::min<size_t>(kk, all.size());
		std::partial_sort(all.begin(), all.begin() + take, all.end(), dist_row_less);   // lexicographic (dist, global row): the single-device order
		for (size_t j = 0; j < take; ++j) {
			out_dist[size_t(q) * kk + j] = all[j].first;
			out_row[size_t(q) * kk + j] = all[j].second;
		}
		out_count[q] = uint32_t(take);
	}
	return RXGPU_OK;
}

int sharded_search_range_impl(rxgpu_index* h, const float* query, float radius, int inclusive, const uint32_t* row_ids, uint64_t n_ids, float* out_dist,
							  uint32_t* out_row, uint64_t cap, uint64_t* out_total) {
	ShardSet* ss = h->shard_set;
	const size_t ns = ss->shards.size();
	std::vector<std::vector<float>> sd(ns);
	std::vector<std::vector<uint32_t>> sr(ns), local_ids(ns);
	std::vector<uint64_t> st(ns, 0);
	if (row_ids) {
		for (uint64_t i = 0; i < n_ids; ++i) {
			if ((i && row_ids[i] <= row_ids[i - 1]) || row_ids[i] >= h->count) {
				set_error("rxgpu_search_range_subset: row ids must be strictly increasing and below count");
				return RXGPU_ERR_PARAMS;
			}
			const size_t s = size_t(row_ids[i] / ss->shard_rows);
			local_ids[s].push_back(uint32_t(row_ids[i] - s * ss->shard_rows));
		}
	}
	std::shared_lock<std::shared_mutex> lk(ss->call_mtx);
	const int rc = for_each_shard(ss, [&](size_t s) -> int {
		uint64_t want = std::max<uint64_t>(cap, 64);
		for (int attempt = 0; attempt < 2; ++attempt) {   // a shard that overflows its buffer is asked again with the size it reported
			sd[s].resize(want);
			sr[s].resize(want);
			int r;
			if (row_ids) {
				if (local_ids[s].empty()) {
					st[s] = 0;
					return RXGPU_OK;
				}
				r = rxgpu_search_range_subset(ss->shards[s], query, radius, inclusive, local_ids[s].data(), local_ids[s].size(), sd[s].data(), sr[s].data(), want, &st[s]);
			} else {
				if (rxgpu_index_count(ss->shards[s]) == 0) {
					st[s] = 0;
					return RXGPU_OK;
				}
				r = rxgpu_search_range(ss->shards[s], query, radius, inclusive, sd[s].data(), sr[s].data(), want, &st[s]);
			}
			if (r != RXGPU_ERR_OVERFLOW) return r;
			want = st[s];
		}
		return RXGPU_ERR_OVERFLOW;
	});
	if (rc != RXGPU_OK) return rc;
	std::vector<std::pair<float, uint32_t>> all;
	for (size_t s = 0; s < ns; ++s) {
		for (uint64_t j = 0; j < st[s]; ++j) all.emplace_back(sd[s][j], uint32_t(sr[s][j] + s * ss->shard_rows));
	}
	std::sort(all.begin(), all.end(), dist_row_less);
	*out_total = all.size();
	for (size_t j = 0; j < all.size() && j < cap; ++j) {
		out_dist[j] = all[j].first;
		out_row[j] = all[j].second;
	}
	if (all.size() > cap) {
		set_error("rxgpu_search_range: more hits than the output buffer holds");
		return RXGPU_ERR_OVERFLOW;
	}
	return RXGPU_OK;
}

// rxgpu_hnsw_search_knn on a sharded handle: every shard searches its own graph (attached through rxgpu_index_shard(h, s)); the merged list
// of a query = the k best of the union of the per-shard results under (dist, global row), global row = s * shard_rows + local row.
// out_count[q] <= k entries, sorted (a superset of the single-device contract, which leaves them unordered).
int sharded_hnsw_search_knn(rxgpu_index* h, const void* queries, const float* qcorr, const float* qnorm, uint32_t nq, uint32_t k, uint32_t ef, float* out_dist, uint32_t* out_row,
							uint32_t* out_count) {
	ShardSet* ss = h->shard_set;
	const size_t ns = ss->shards.size();
	if (nq == 0) return RXGPU_OK;
	if (k == 0) {
		std::fill(out_count, out_count + nq, 0u);
		return RXGPU_OK;
	}
	std::shared_lock<std::shared_mutex> lk(ss->call_mtx);
	bool any = false;
	for (rxgpu_index* sh : ss->shards) any = any || rxgpu_index_count(sh) != 0;
	if (!any) {
		std::fill(out_count, out_count + nq, 0u);
		return RXGPU_OK;
	}
	if (ss->xch && k <= uint32_t(kMaxFusedK)) {
		ShardExchange* x = ss->xch;
		CurrentDevice cd;
		ExchangeLane* lane = take_lane(x);
		if (!lane) {
			if (int rc = new_lane(x, &lane); rc) return rc;
		}
		const int rc = exchange_hnsw_search_knn(h, ss, lane, queries, qcorr, qnorm, nq, k, ef, out_dist, out_row, out_count);
		give_lane(x, lane, rc);
		return rc;
	}
	std::vector<std::vector<float>> sd(ns);
	std::vector<std::vector<uint32_t>> sr(ns), sc(ns);
	const int rc = for_each_shard(ss, [&](size_t s) -> int {
		sd[s].assign(size_t(nq) * k, 0.f);
		sr[s].assign(size_t(nq) * k, 0u);
		sc[s].assign(nq, 0u);
		if (rxgpu_index_count(ss->shards[s]) == 0) return RXGPU_OK;
		if (qcorr) {
			return rxgpu_hnsw_search_knn_sq8(ss->shards[s], static_cast<const uint8_t*>(queries), qcorr, qnorm, nq, k, ef, sd[s].data(), sr[s].data(), sc[s].data());
		}
		return rxgpu_hnsw_search_knn(ss->shards[s], static_cast<const float*>(queries), nq, k, ef, sd[s].data(), sr[s].data(), sc[s].data());
	});
	if (rc != RXGPU_OK) return rc;
	std::vector<std::pair<float, uint32_t>> all;
	for (uint32_t q = 0; q < nq; ++q) {
		all.clear();
		for (size_t s = 0; s < ns; ++s) {
			// a shard's search clamps k to the points it holds and writes its lists with THAT stride (hnsw_search_impl: p.k = min(k, count))
			const size_t ks = std::min<uint64_t>(k, rxgpu_index_count(ss->shards[s]));
			for (uint32_t j = 0; j < sc[s][q]; ++j) all.emplace_back(sd[s][size_t(q) * ks + j], uint32_t(sr[s][size_t(q) * ks + j] + s * ss->shard_rows));
		}
		const size_t take = std::min<size_t>(k, all.size());
		std::partial_sort(all.begin(), all.begin() + take, all.end(), dist_row_less);
		for (size_t j = 0; j < take; ++j) {
			out_dist[size_t(q) * k + j] = all[j].first;
			out_row[size_t(q) * k + j] = all[j].second;
		}
		out_count[q] = uint32_t(take);
	}
	return RXGPU_OK;
}

// HierarchicalNSW::SearchRange over the shards' graphs: every shard's ef-search + closure (rxgpu_hnsw_search_range), hits concatenated with
// global rows (unordered, like the single-device call).  Overflow protocol as there: *out_total = hits counted so far.
int sharded_hnsw_search_range(rxgpu_index* h, const float* query, float radius, uint32_t ef, float* out_dist, uint32_t* out_row, uint64_t cap, uint64_t* out_total) {
	ShardSet* ss = h->shard_set;
	const size_t ns = ss->shards.size();
	if (!query || !out_total || (cap && !(out_dist && out_row))) {
		set_error("rxgpu_hnsw_search_range: null argument");
		return RXGPU_ERR_PARAMS;
	}
	*out_total = 0;
	std::vector<std::vector<float>> sd(ns);
	std::vector<std::vector<uint32_t>> sr(ns);
	std::vector<uint64_t> st(ns, 0);
	std::shared_lock<std::shared_mutex> lk(ss->call_mtx);
	const int rc = for_each_shard(ss, [&](size_t s) -> int {
		if (rxgpu_index_count(ss->shards[s]) == 0) return RXGPU_OK;
		uint64_t want = std::max<uint64_t>(cap, 256);
		for (int attempt = 0; attempt < 8; ++attempt) {   // the count an overflow reports is a lower bound: grow until the closure fits
			sd[s].resize(want);
			sr[s].resize(want);
			const int r = rxgpu_hnsw_search_range(ss->shards[s], query, radius, ef, sd[s].data(), sr[s].data(), want, &st[s]);
			if (r != RXGPU_ERR_OVERFLOW) return r;
			want = std::max<uint64_t>(want * 4, st[s] * 2);
		}
		return RXGPU_ERR_OVERFLOW;
	});
	if (rc != RXGPU_OK) return rc;
	uint64_t total = 0;
	for (size_t s = 0; s < ns; ++s) {
		for (uint64_t j = 0; j < st[s]; ++j, ++total) {
			if (total < cap) {
				out_dist[total] = sd[s][j];
				out_row[total] = uint32_t(sr[s][j] + s * ss->shard_rows);
			}
		}
	}
	*out_total = total;
	if (total > cap) {
		set_error("rxgpu_hnsw_search_range: more hits than the output buffer holds");
		return RXGPU_ERR_OVERFLOW;
	}
	return RXGPU_OK;
}

int sharded_distances(rxgpu_index* h, const float* query, const uint32_t* rows, uint32_t n, float* out_dist) {
	ShardSet* ss = h->shard_set;
	const size_t ns = ss->shards.size();
	std::vector<std::vector<uint32_t>> local(ns), where(ns);
	for (uint32_t i = 0; i < n; ++i) {
		if (rows[i] >= h->count) {
			set_error("rxgpu_distances: row out of range");
			return RXGPU_ERR_PARAMS;
		}
		const size_t s = size_t(rows[i] / ss->shard_rows);
		local[s].push_back(uint32_t(rows[i] - s * ss->shard_rows));
		where[s].push_back(i);
	}
	std::vector<std::vector<float>> sd(ns);
	std::shared_lock<std::shared_mutex> lk(ss->call_mtx);
	const int rc = for_each_shard(ss, [&](size_t s) -> int {
		if (local[s].empty()) return RXGPU_OK;
		sd[s].resize(local[s].size());
		return rxgpu_distances(ss->shards[s], query, local[s].data(), uint32_t(local[s].size()), sd[s].data());
	});
	if (rc != RXGPU_OK) return rc;
	for (size_t s = 0; s < ns; ++s) {
		for (size_t j = 0; j < where[s].size(); ++j) out_dist[where[s][j]] = sd[s][j];
	}
	return RXGPU_OK;
}

// RemovePoint's swap-with-last across shards: the row travels through the host (two devices may be involved)
int sharded_move_row(rxgpu_index* h, uint64_t from, uint64_t to) {
	ShardSet* ss = h->shard_set;
	if (from >= h->count || to >= h->count) {
		set_error("rxgpu_index_move_row: row out of range");
		return RXGPU_ERR_PARAMS;
	}
	if (from == to) return RXGPU_OK;
	const size_t sf = size_t(from / ss->shard_rows), st = size_t(to / ss->shard_rows);
	std::unique_lock<std::shared_mutex> lk(ss->call_mtx);
	std::string err;
	int rc;
	if (sf == st) {
		rc = run_on_shard(ss, sf, [&] { return rxgpu_index_move_row(ss->shards[sf], from - sf * ss->shard_rows, to - sf * ss->shard_rows); }, err);
	} else {
		std::vector<float> row(h->dim);
		float norm = 0.f;
		rc = run_on_shard(ss, sf, [&] { return rxgpu_index_download_row(ss->shards[sf], from - sf * ss->shard_rows, row.data(), &norm); }, err);
		if (rc == RXGPU_OK) {
			rc = run_on_shard(ss, st, [&] {
				return rxgpu_index_upload_rows(ss->shards[st], to - st * ss->shard_rows, 1, row.data(), h->metric == RXGPU_METRIC_COSINE ? &norm : nullptr);
			}, err);
		}
	}
	if (rc != RXGPU_OK) set_error(err);
	return rc;
}

}  // namespace rxgpu

extern "C" {

int rxgpu_index_create_sharded(int metric, uint32_t dim, uint64_t capacity, uint32_t n_devices, const int* devices, rxgpu_index** out) {
	if (!out || !devices || n_devices == 0 || n_devices > 64) {
		set_error("rxgpu_index_create_sharded: bad arguments (1..64 devices)");
		return RXGPU_ERR_PARAMS;
	}
	if (capacity == 0 || capacity >= 0xFFFFFFFFull) {
		set_error("rxgpu_index_create_sharded: capacity must be in [1, 2^32)");
		return RXGPU_ERR_PARAMS;
	}
	*out = nullptr;
	auto* ss = new rxgpu::ShardSet();
	ss->shard_rows = ((capacity + n_devices - 1) / n_devices + 31) & ~uint64_t(31);   // whole bitmap words per shard
	auto* h = new rxgpu_index();
	h->metric = metric;
	h->dim = dim;
	h->stride = (dim + 3u) & ~3u;
	h->device = devices[0];
	h->capacity = capacity;
	h->shard_set = ss;
	for (uint32_t s = 0; s < n_devices; ++s) {
		rxgpu_index* sh = nullptr;
		const uint64_t lo = uint64_t(s) * ss->shard_rows;
		const uint64_t cap = capacity > lo ? std::min<uint64_t>(capacity - lo, ss->shard_rows) : 0;
		const int rc = rxgpu_index_create(metric, dim, std::max<uint64_t>(cap, 1), devices[s], &sh);
		if (rc != RXGPU_OK) {
			rxgpu::sharded_destroy(h);
			delete h;
			return rc;
		}
		ss->shards.push_back(sh);
		auto* w = new rxgpu::ShardWorker();
		for (unsigned t = 0; t < rxgpu::ShardWorker::kThreads; ++t) {
			w->threads.emplace_back([w, dev = devices[s]] {
				(void)hipSetDevice(dev);
				w->run();
			});
		}
		ss->workers.push_back(w);
	}
	const char* mode = getenv("RXGPU_SHARD_MERGE");
	if (mode && std::strcmp(mode, "host") == 0) {
		ss->merge_note = "RXGPU_SHARD_MERGE=host";
	} else if (rxgpu::exchange_create(ss, n_devices, devices, &ss->xch) != RXGPU_OK) {
		// no device-side exchange on this node: the index works on the host-merge path (same results); said once per index, and kept for
		// rxgpu_index_shard_merge_note
		ss->xch = nullptr;
		ss->merge_note = rxgpu_last_error();
		fprintf(stderr, "rxgpu: sharded index over %u device slot(s): %s — per-shard lists are merged on the host\n", n_devices, ss->merge_note.c_str());
	}
	*out = h;
	return RXGPU_OK;
}

int rxgpu_index_shard_merge_mode(const rxgpu_index* h) { return h && h->shard_set ? (h->shard_set->xch ? 1 : 0) : -1; }
const char* rxgpu_index_shard_merge_note(const rxgpu_index* h) { return h && h->shard_set ? h->shard_set->merge_note.c_str() : ""; }
uint32_t rxgpu_index_shard_ranks(const rxgpu_index* h) { return h && h->shard_set && h->shard_set->xch ? h->shard_set->xch->nranks : 0; }
uint64_t rxgpu_index_shard_collectives(const rxgpu_index* h) {
	return h && h->shard_set && h->shard_set->xch ? h->shard_set->xch->collectives.load(std::memory_order_relaxed) : 0;
}
rxgpu_index* rxgpu_index_shard(rxgpu_index* h, uint32_t s) {
	return h && h->shard_set && s < h->shard_set->shards.size() ? h->shard_set->shards[s] : nullptr;
}
/* After the caller filled shards directly (rxgpu_index_adopt_device_rows on rxgpu_index_shard handles): count = rows held, which must
 * form a prefix of the global row space (every shard before the last non-empty one full). */
int rxgpu_index_shard_sync_count(rxgpu_index* h) {
	if (!h || !h->shard_set) {
		set_error("rxgpu_index_shard_sync_count: not a sharded index");
		return RXGPU_ERR_PARAMS;
	}
	rxgpu::ShardSet* ss = h->shard_set;
	std::unique_lock<std::shared_mutex> lk(ss->call_mtx);
	uint64_t total = 0;
	bool ended = false;
	for (rxgpu_index* sh : ss->shards) {
		const uint64_t c = rxgpu_index_count(sh);
		if (c > ss->shard_rows || (ended && c)) {
			set_error("rxgpu_index_shard_sync_count: shards must hold a prefix of the global rows (full shards, then at most one partial)");
			return RXGPU_ERR_PARAMS;
		}
		ended = ended || c < ss->shard_rows;
		total += c;
	}
	h->count = total;
	return RXGPU_OK;
}

uint32_t rxgpu_index_shard_count(const rxgpu_index* h) { return h && h->shard_set ? uint32_t(h->shard_set->shards.size()) : 0; }
uint64_t rxgpu_index_shard_rows(const rxgpu_index* h) { return h && h->shard_set ? h->shard_set->shard_rows : 0; }

}  // extern "C"
