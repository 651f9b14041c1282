// C-ABI implementation (include/rxgpu.h): index storage in HBM, search entry points, instrumentation.
// Host-side plumbing only — all arithmetic is in the kernels.
#include <unistd.h>   // environ
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>

#include "../../include/rxgpu.h"
#include "knn_kernels.hip.h"
#include "rxgpu_internal.h"

namespace rxgpu {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
}  // namespace rxgpu

using rxgpu::set_error;

namespace rxgpu {
// hipFree / hipHostFree wait for the whole device — and a resident search kernel (rxgpu_hnsw_server.hip) stays on it for up to its lifetime:
// a scratch buffer that grew on some thread's launch path stalled that thread for tens of milliseconds (measured: T = 16 planner threads over
// 10M rows, 7 of 1024 queries took the launches, the leg lasted 73 ms instead of 47).  While such a kernel may be alive, buffers that are
// replaced are retired instead and freed at the next point that waits for the device anyway (a mutation, a quiesce, index destruction),
// or when 256 MB have piled up.
std::atomic<int> g_resident_kernels{0};
namespace {
std::mutex g_retired_mtx;
std::vector<std::pair<void*, bool>> g_retired;   // (pointer, host memory?)
size_t g_retired_bytes = 0;
}  // namespace
void drain_retired() {
	std::vector<std::pair<void*, bool>> take;
	{
		std::lock_guard<std::mutex> lk(g_retired_mtx);
		take.swap(g_retired);
		g_retired_bytes = 0;
	}
	for (const auto& e : take) {
		if (e.second) {
			(void)hipHostFree(e.first);
		} else {
			(void)hipFree(e.first);
		}
	}
}
void free_or_retire(void* ptr, size_t bytes, bool host) {
	if (!ptr) return;
	if (g_resident_kernels.load(std::memory_order_acquire) <= 0) {
		if (host) {
			(void)hipHostFree(ptr);
		} else {
			(void)hipFree(ptr);
		}
		return;
	}
	bool drain = false;
	{
		std::lock_guard<std::mutex> lk(g_retired_mtx);
		g_retired.emplace_back(ptr, host);
		g_retired_bytes += bytes;
		drain = g_retired_bytes > (size_t(256) << 20);
	}
	if (drain) drain_retired();
}
// hipDeviceSynchronize for a device that may hold resident search kernels: they are told to leave first — the wait would otherwise last
// until their idle / lifetime limit
hipError_t device_wait_all(int device) {
	hnsw_servers_pause_device(device);
	const hipError_t e = hipDeviceSynchronize();
	drain_retired();
	return e;
}
}  // namespace rxgpu

#define RX_HIP(expr)                                                                                      \
	do {                                                                                                  \
		hipError_t e__ = (expr);                                                                          \
		if (e__ != hipSuccess) {                                                                          \
			set_error(std::string(#expr) + ": " + hipGetErrorString(e__));                                \
			return e__ == hipErrorOutOfMemory ? RXGPU_ERR_NOMEM : RXGPU_ERR_DEVICE;                       \
		}                                                                                                 \
	} while (0)

#define RX_CHECK(cond, code, msg) \
	do {                          \
		if (!(cond)) {            \
			set_error(msg);       \
			return code;          \
		}                         \
	} while (0)

int rxgpu_devbuf::ensure(size_t need) {
	if (need <= bytes) return RXGPU_OK;
	rxgpu::free_or_retire(ptr, bytes, false);
	ptr = nullptr;
	bytes = 0;
	const size_t want = std::max<size_t>(need, 4096);
	RX_HIP(hipMalloc(&ptr, want));
	bytes = want;
	// RXGPU_DEBUG_FILL=<byte>: every fresh scratch buffer is filled with it (a read before the first write then shows, whatever the allocator returned)
	static const int fill = [] {
		const char* e = std::getenv("RXGPU_DEBUG_FILL");
		return e ? int(std::strtol(e, nullptr, 0)) & 0xFF : -1;
	}();
	if (fill >= 0) RX_HIP(hipMemset(ptr, fill, want));
	return RXGPU_OK;
}
void rxgpu_devbuf::release() {
	rxgpu::free_or_retire(ptr, bytes, false);
	ptr = nullptr;
	bytes = 0;
}
int rxgpu_search_ctx::ensure_pinned(size_t need) {
	if (need <= h_pinned_bytes) return RXGPU_OK;
	rxgpu::free_or_retire(h_pinned, h_pinned_bytes, true);
	h_pinned = nullptr;
	h_pinned_bytes = 0;
	const size_t want = std::max<size_t>(need, 1 << 16);
	RX_HIP(hipHostMalloc(&h_pinned, want, hipHostMallocDefault));
	h_pinned_bytes = want;
	return RXGPU_OK;
}
int rxgpu_search_ctx::ensure_aux() {
	if (aux_stream) return RXGPU_OK;
	RX_HIP(hipStreamCreateWithFlags(&aux_stream, hipStreamNonBlocking));
	RX_HIP(hipStreamCreateWithFlags(&aux2_stream, hipStreamNonBlocking));
	RX_HIP(hipEventCreateWithFlags(&split_done, hipEventDisableTiming));
	RX_HIP(hipEventCreateWithFlags(&aux_done, hipEventDisableTiming));
	RX_HIP(hipEventCreateWithFlags(&main_done, hipEventDisableTiming));
	return RXGPU_OK;
}
void rxgpu_search_ctx::release() {
	if (aux_done) (void)hipEventDestroy(aux_done);
	if (main_done) (void)hipEventDestroy(main_done);
	if (split_done) (void)hipEventDestroy(split_done);
	if (aux_stream) (void)hipStreamDestroy(aux_stream);
	if (aux2_stream) (void)hipStreamDestroy(aux2_stream);
	aux_done = main_done = split_done = nullptr;
	aux_stream = aux2_stream = nullptr;
	d_queries.release();
	d_part_dist.release();
	d_part_row.release();
	d_out_dist.release();
	d_out_row.release();
	d_out_count.release();
	d_misc.release();
	d_select.release();
	d_qpad.release();
	d_qstats.release();
	d_dense.release();
	d_cand_row.release();
	d_cand_dist.release();
	d_cand_cnt.release();
	d_visited.release();
	d_helper.release();
	d_helper_bits.release();
	d_ivf.release();
	d_gcand_d.release();
	d_redo.release();
	d_top.release();
	d_subset.release();
	d_bitmap.release();
	d_tiles.release();
	if (h_pinned) (void)hipHostFree(h_pinned);
	h_pinned = nullptr;
	if (own_stream && stream) (void)hipStreamDestroy(stream);
	stream = nullptr;
}

namespace {

struct DeviceGuard {
	int prev = -1;
	bool ok = true;
	explicit DeviceGuard(int dev) {
		if (hipGetDevice(&prev) != hipSuccess) prev = -1;
		if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
	}
	~DeviceGuard() {
		if (prev >= 0) (void)hipSetDevice(prev);
	}
};

// Check out a scratch context with its own stream (host-synchronous searches).
rxgpu_search_ctx* acquire_ctx(rxgpu_index* h) {
	{
		std::lock_guard<std::mutex> lk(h->mtx);
		if (!h->free_ctx.empty()) {
			auto* c = h->free_ctx.back();
			h->free_ctx.pop_back();
			return c;
		}
	}
	auto* c = new rxgpu_search_ctx();
	if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
		delete c;
		set_error("hipStreamCreateWithFlags failed");
		return nullptr;
	}
	c->own_stream = true;
	return c;
}
void release_ctx(rxgpu_index* h, rxgpu_search_ctx* c) {
	std::lock_guard<std::mutex> lk(h->mtx);
	h->free_ctx.push_back(c);
}
// Resident contexts are per calling thread (rxgpu_search_knn_resident).  Planner threads come and go: a thread that ends hands the contexts it
// held back to the pools of the indexes that still exist — looked up by serial number, never through a pointer the thread kept — so short-lived
// threads neither pile up streams + device buffers until rxgpu_index_destroy nor leave their context to a later thread that got the same id.
std::mutex g_live_mtx;
std::map<uint64_t, rxgpu_index*> g_live_indexes;
std::atomic<uint64_t> g_index_serial{0};
struct ResidentThread {
	std::vector<uint64_t> used;
	~ResidentThread() {
		std::lock_guard<std::mutex> live(g_live_mtx);
		for (uint64_t serial : used) {
			auto it = g_live_indexes.find(serial);
			if (it == g_live_indexes.end()) continue;
			rxgpu_index* h = it->second;
			rxgpu_search_ctx* c = nullptr;
			{
				std::lock_guard<std::mutex> lk(h->resident_mtx);
				auto slot = h->resident_ctx.find(std::this_thread::get_id());
				if (slot == h->resident_ctx.end()) continue;
				c = slot->second;
				h->resident_ctx.erase(slot);
			}
			if (c) release_ctx(h, c);   // its stream orders the next user's work behind whatever this thread left running
		}
	}
};
thread_local ResidentThread t_resident;
void register_live_index(rxgpu_index* h) {
	std::lock_guard<std::mutex> live(g_live_mtx);
	h->serial = ++g_index_serial;
	g_live_indexes[h->serial] = h;
}
void unregister_live_index(rxgpu_index* h) {
	std::lock_guard<std::mutex> live(g_live_mtx);
	g_live_indexes.erase(h->serial);
}
// Scratch bound to a caller-owned stream: stream order makes reuse safe without synchronising.
rxgpu_search_ctx* stream_ctx(rxgpu_index* h, void* stream) {
	std::lock_guard<std::mutex> lk(h->mtx);
	auto it = h->stream_ctx.find(stream);
	if (it != h->stream_ctx.end()) return it->second;
	auto* c = new rxgpu_search_ctx();
	c->stream = static_cast<hipStream_t>(stream);
	c->own_stream = false;
	h->stream_ctx[stream] = c;
	return c;
}

struct ProfileScope {
	rxgpu_index* h;
	const char* name;
	hipStream_t s;
	hipEvent_t a = nullptr, b = nullptr;
	ProfileScope(rxgpu_index* h_, const char* n, hipStream_t s_) : h(h_), name(n), s(s_) {
		if (!h->profiling) return;
		if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
			a = b = nullptr;
			return;
		}
		(void)hipEventRecord(a, s);
	}
	~ProfileScope() {
		if (!a) return;
		(void)hipEventRecord(b, s);
		std::lock_guard<std::mutex> lk(h->mtx);
		h->profile[name].events.emplace_back(a, b);
	}
};

// Enqueue scan + merge for nq device-resident queries; results land in d_out_* (device).
int enqueue_knn_fused(rxgpu_index* h, rxgpu_search_ctx* c, const float* d_queries, uint32_t nq, uint32_t kk, float* d_out_dist,
					  uint32_t* d_out_row, uint32_t* d_out_count) {
	const uint32_t gridx = rxgpu::scan_grid_x(h->count, h->cus);
	const size_t part = size_t(nq) * gridx * kk;
	if (int rc = c->d_part_dist.ensure(part * sizeof(float)); rc) return rc;
	if (int rc = c->d_part_row.ensure(part * sizeof(uint32_t)); rc) return rc;
	rxgpu::ScanParams p{};
	p.rows = h->d_rows;
	p.inv_norms = h->d_inv_norms;
	p.queries = d_queries;
	p.n = h->count;
	p.stride = h->stride;
	p.dim = h->dim;
	p.kk = kk;
	p.part_dist = static_cast<float*>(c->d_part_dist.ptr);
	p.part_row = static_cast<uint32_t*>(c->d_part_row.ptr);
	{
		ProfileScope ps(h, "scan", c->stream);
		rxgpu::launch_scan(h->metric, p, nq, gridx, c->stream);
	}
	{
		ProfileScope ps(h, "merge", c->stream);
		rxgpu::launch_merge_lists(p.part_dist, p.part_row, gridx, kk, nq, d_out_dist, d_out_row, d_out_count, c->stream);
	}
	RX_HIP(hipGetLastError());
	return RXGPU_OK;
}

// ---- batched path (nq >= 2): MFMA candidate generation + exact re-score, see knn_batched.hip ----------------------
constexpr uint32_t kBatchSampleRows = 32768;
constexpr uint32_t kBatchSampleRowsBf16 = 131072;   // the sample pass is cheap on the bf16 pipe; a tighter threshold pays for the wider margin

static int batch_min_queries() {
	static const int v = [] {
		const char* e = getenv("RXGPU_BATCH_MIN");
		return e ? atoi(e) : 2;
	}();
	return v;
}

// Per-row statistics are cached on the index; recomputed (synchronously, under the index mutex) after any mutation.
int ensure_row_stats(rxgpu_index* h, hipStream_t s) {
	std::lock_guard<std::mutex> lk(h->mtx);
	if (h->stats_valid) return RXGPU_OK;
	if (!h->d_stats) RX_HIP(hipMalloc(reinterpret_cast<void**>(&h->d_stats), 2 * sizeof(unsigned int)));
	if (h->metric == RXGPU_METRIC_L2 && h->row_sq_capacity < h->count) {
		if (h->d_row_sq) (void)hipFree(h->d_row_sq);
		h->d_row_sq = nullptr;
		h->row_sq_capacity = 0;
		RX_HIP(hipMalloc(reinterpret_cast<void**>(&h->d_row_sq), std::max<uint64_t>(h->capacity, h->count) * sizeof(float)));
		h->row_sq_capacity = std::max<uint64_t>(h->capacity, h->count);
	}
	RX_HIP(hipMemsetAsync(h->d_stats, 0, 2 * sizeof(unsigned int), s));
	rxgpu::launch_row_stats(h->d_rows, h->d_inv_norms, h->count, h->stride, h->dim, h->metric == RXGPU_METRIC_L2 ? h->d_row_sq : nullptr,
							h->d_stats, h->cus, s);
	RX_HIP(hipGetLastError());
	RX_HIP(hipStreamSynchronize(s));
	h->stats_valid = true;
	return RXGPU_OK;
}

// bf16 shadow of the rows for the nomination GEMM: 2 bytes per element on top of the 4-byte rows (HBM is 288 GB: 10M x 768 costs 15.4 GB);
// rebuilt lazily after any mutation, like the row statistics.
int ensure_bf16_shadow(rxgpu_index* h, hipStream_t s) {
	std::lock_guard<std::mutex> lk(h->mtx);
	if (h->bf16_valid) return RXGPU_OK;
	const uint32_t ld = (h->dim + 63u) & ~63u;
	const uint64_t need = (std::max<uint64_t>(h->capacity, h->count) + rxgpu::kShadowTileRows - 1) / rxgpu::kShadowTileRows * rxgpu::kShadowTileRows;   // whole tiles
	if (h->bf16_capacity < need) {
		if (h->d_rows_bf16) (void)hipFree(h->d_rows_bf16);
		h->d_rows_bf16 = nullptr;
		h->bf16_capacity = 0;
		const char* e = getenv("RXGPU_SHADOW_BLOCKED");
		h->bf16_blocked = !(e && atoi(e) == 0);
		if (hipMalloc(reinterpret_cast<void**>(&h->d_rows_bf16), need * ld * sizeof(uint16_t)) != hipSuccess) {
			(void)hipGetLastError();   // not an error of the search: the caller falls back to the f32 rows
			h->d_rows_bf16 = nullptr;
			h->bf16_unavailable = true;
			return RXGPU_ERR_NOMEM;
		}
		h->bf16_capacity = need;
	}
	rxgpu::launch_to_bf16(h->d_rows, h->count, h->stride, h->dim, h->d_rows_bf16, ld, h->cus, s, 0, h->bf16_blocked);
	RX_HIP(hipGetLastError());
	RX_HIP(hipStreamSynchronize(s));
	h->bf16_valid = true;
	return RXGPU_OK;
}

static int batch_bf16_min_queries() {   // read per call (tests and A/B runs switch it)
	const char* e = getenv("RXGPU_BATCH_BF16_MIN");   // 0 disables the bf16 nomination path
	return e ? atoi(e) : 2;   // measured at 10M x 768: 4.8-5.0 ms per batch for 8..256 queries against 6.2-10.6 ms on the f32 rows
}

// Every batch (2..256 queries at a time): nomination on the bf16 MFMA pipe over the bf16 shadow (knn_batched_bf16.hip), then the same exact tail.
int enqueue_knn_batched_bf16(rxgpu_index* h, rxgpu_search_ctx* c, const float* d_queries, uint32_t q0, uint32_t cq, uint32_t kk,
							 float* d_out_dist, uint32_t* d_out_row, uint32_t* d_out_count) {
	if (int rc = ensure_bf16_shadow(h, c->stream); rc) return rc;
	const uint32_t mt = cq <= 128 ? 128 : 256;   // query-tile width of the nomination kernel
	const uint32_t ld = (h->dim + 63u) & ~63u;
	const uint32_t q_stride = ld;   // the f32 copy for the exact re-score shares the padded stride
	const uint64_t ns = std::min<uint64_t>(h->count, kBatchSampleRowsBf16);
	// nominations per query ~ kk * n / ns, times ~3 for the bf16 margin; 10x headroom, overflow falls back to the exact scan
	uint64_t cap64 = std::max<uint64_t>(4096, 10 * uint64_t(kk) * ((h->count + ns - 1) / ns));
	cap64 = std::min<uint64_t>(cap64, std::max<uint64_t>(h->count, 64));
	const uint32_t cap = uint32_t((cap64 + 63) & ~63ull);
	if (int rc = c->d_qpad.ensure(size_t(mt) * q_stride * (sizeof(float) + sizeof(uint16_t))); rc) return rc;
	if (int rc = c->d_qstats.ensure(size_t(3) * mt * sizeof(float)); rc) return rc;
	if (int rc = c->d_dense.ensure(size_t(mt) * ns * sizeof(float)); rc) return rc;
	if (int rc = c->d_cand_row.ensure(size_t(mt) * cap * sizeof(uint32_t)); rc) return rc;
	if (int rc = c->d_cand_dist.ensure(size_t(mt) * cap * sizeof(float)); rc) return rc;
	if (int rc = c->d_cand_cnt.ensure(size_t(mt) * sizeof(uint32_t)); rc) return rc;
	float* qpad = static_cast<float*>(c->d_qpad.ptr);
	uint16_t* qbf = reinterpret_cast<uint16_t*>(qpad + size_t(mt) * q_stride);
	float* q_sq = static_cast<float*>(c->d_qstats.ptr);
	float* margin = q_sq + mt;
	float* thr = q_sq + 2 * mt;
	uint32_t* cand_cnt = static_cast<uint32_t*>(c->d_cand_cnt.ptr);
	RX_HIP(hipMemsetAsync(qpad, 0, size_t(mt) * q_stride * sizeof(float), c->stream));
	RX_HIP(hipMemcpy2DAsync(qpad, q_stride * sizeof(float), d_queries + size_t(q0) * h->dim, h->dim * sizeof(float), h->dim * sizeof(float), cq,
							hipMemcpyDeviceToDevice, c->stream));
	RX_HIP(hipMemsetAsync(cand_cnt, 0, size_t(mt) * sizeof(uint32_t), c->stream));
	rxgpu::launch_to_bf16(qpad, mt, q_stride, q_stride, qbf, ld, h->cus, c->stream);
	rxgpu::launch_query_stats(h->metric, qpad, cq, mt, q_stride, h->dim, h->d_stats, q_sq, margin, true, c->stream);

	rxgpu::GemmBf16Params g{};
	g.rows = h->d_rows_bf16;
	g.blocked = (h->bf16_blocked ? 1u : 0u) | ((getenv("RXGPU_GEMM_PRIO") && atoi(getenv("RXGPU_GEMM_PRIO"))) ? 2u : 0u);   // bit 1: s_setprio around the MFMA bursts (A/B)
	g.queries = qbf;
	g.inv_norms = h->d_inv_norms;
	g.row_sq = h->d_row_sq;
	g.q_sq = q_sq;
	g.ld = ld;
	g.nq = cq;
	auto grid_for = [&](uint64_t rows) { return uint32_t(std::max<uint64_t>(1, std::min<uint64_t>((rows + 255) / 256, uint64_t(h->cus)))); };
	g.n = ns;
	g.row_step = uint32_t(std::max<uint64_t>(1, h->count / ns));   // strided sample: representative whatever the insertion order
	g.dense = static_cast<float*>(c->d_dense.ptr);
	{
		ProfileScope ps(h, "gemm_sample", c->stream);
		RX_HIP(rxgpu::launch_gemm_bf16(h->metric, rxgpu::kGemmDense, int(mt), g, grid_for(ns), c->stream));
	}
	rxgpu::launch_sample_threshold(g.dense, ns, cq, mt, kk, margin, thr, c->stream);
	g.n = h->count;
	g.row_step = 1;
	g.dense = nullptr;
	g.thr = thr;
	g.cand_row = static_cast<uint32_t*>(c->d_cand_row.ptr);
	g.cand_cnt = cand_cnt;
	g.cap = cap;
	{
		ProfileScope ps(h, "gemm", c->stream);
		RX_HIP(rxgpu::launch_gemm_bf16(h->metric, rxgpu::kGemmFilter, int(mt), g, grid_for(h->count), c->stream));
	}
	{
		ProfileScope ps(h, "rescore", c->stream);
		rxgpu::launch_rescore(h->metric, h->d_rows, h->d_inv_norms, qpad, q_stride, h->stride, h->dim, cq, cap, cand_cnt, g.cand_row,
							  static_cast<float*>(c->d_cand_dist.ptr), c->stream);
	}
	rxgpu::launch_merge(static_cast<float*>(c->d_cand_dist.ptr), g.cand_row, cap, kk, cq, d_out_dist + size_t(q0) * kk, d_out_row + size_t(q0) * kk,
						d_out_count ? d_out_count + q0 : nullptr, nullptr, 0, c->stream);
	{   // overflow fallback, gated on device
		const uint32_t gridx = rxgpu::scan_grid_x(h->count, h->cus);
		const size_t part = size_t(cq) * gridx * kk;
		if (int rc = c->d_part_dist.ensure(part * sizeof(float)); rc) return rc;
		if (int rc = c->d_part_row.ensure(part * sizeof(uint32_t)); rc) return rc;
		rxgpu::ScanParams p{};
		p.rows = h->d_rows;
		p.inv_norms = h->d_inv_norms;
		p.queries = d_queries + size_t(q0) * h->dim;
		p.n = h->count;
		p.stride = h->stride;
		p.dim = h->dim;
		p.kk = kk;
		p.part_dist = static_cast<float*>(c->d_part_dist.ptr);
		p.part_row = static_cast<uint32_t*>(c->d_part_row.ptr);
		p.gate_cnt = cand_cnt;
		p.gate_cap = cap;
		ProfileScope ps(h, "fallback_scan", c->stream);
		rxgpu::launch_scan(h->metric, p, cq, gridx, c->stream);
		rxgpu::launch_merge(p.part_dist, p.part_row, gridx * kk, kk, cq, d_out_dist + size_t(q0) * kk, d_out_row + size_t(q0) * kk,
							d_out_count ? d_out_count + q0 : nullptr, cand_cnt, cap, c->stream);
	}
	RX_HIP(hipGetLastError());
	return RXGPU_OK;
}

int enqueue_knn_batched(rxgpu_index* h, rxgpu_search_ctx* c, const float* d_queries, uint32_t nq, uint32_t kk, float* d_out_dist,
						uint32_t* d_out_row, uint32_t* d_out_count) {
	if (int rc = ensure_row_stats(h, c->stream); rc) return rc;
	const uint32_t q_stride = (h->dim + 31u) & ~31u;
	const uint64_t ns = std::min<uint64_t>(h->count, kBatchSampleRows);
	// expected nominations per query ~ kk * n / ns (plus the eps margin); 4x headroom, overflow falls back to the exact scan
	uint64_t cap64 = std::max<uint64_t>(4096, 4 * uint64_t(kk) * ((h->count + ns - 1) / ns));
	cap64 = std::min<uint64_t>(cap64, std::max<uint64_t>(h->count, 64));
	const uint32_t cap = uint32_t((cap64 + 63) & ~63ull);
	for (uint32_t q0 = 0; q0 < nq; q0 += 256) {
		const uint32_t cq = std::min<uint32_t>(256, nq - q0);
		if (batch_bf16_min_queries() > 0 && int(cq) >= batch_bf16_min_queries() && !h->bf16_unavailable) {
			const int rc = enqueue_knn_batched_bf16(h, c, d_queries, q0, cq, kk, d_out_dist, d_out_row, d_out_count);
			if (rc == RXGPU_OK) continue;
			if (!(rc == RXGPU_ERR_NOMEM && h->bf16_unavailable)) return rc;   // no room for the shadow: f32 nomination below
		}
		const int mt = cq <= 32 ? 32 : cq <= 64 ? 64 : cq <= 128 ? 128 : 256;
		if (int rc = c->d_qpad.ensure(size_t(mt) * q_stride * sizeof(float)); rc) return rc;
		if (int rc = c->d_qstats.ensure(size_t(3) * mt * sizeof(float)); rc) return rc;
		if (int rc = c->d_dense.ensure(size_t(mt) * ns * sizeof(float)); rc) return rc;
		if (int rc = c->d_cand_row.ensure(size_t(mt) * cap * sizeof(uint32_t)); rc) return rc;
		if (int rc = c->d_cand_dist.ensure(size_t(mt) * cap * sizeof(float)); rc) return rc;
		if (int rc = c->d_cand_cnt.ensure(size_t(mt) * sizeof(uint32_t)); rc) return rc;
		float* qpad = static_cast<float*>(c->d_qpad.ptr);
		float* q_sq = static_cast<float*>(c->d_qstats.ptr);
		float* margin = q_sq + mt;
		float* thr = q_sq + 2 * mt;
		uint32_t* cand_cnt = static_cast<uint32_t*>(c->d_cand_cnt.ptr);
		RX_HIP(hipMemsetAsync(qpad, 0, size_t(mt) * q_stride * sizeof(float), c->stream));
		RX_HIP(hipMemcpy2DAsync(qpad, q_stride * sizeof(float), d_queries + size_t(q0) * h->dim, h->dim * sizeof(float),
								h->dim * sizeof(float), cq, hipMemcpyDeviceToDevice, c->stream));
		RX_HIP(hipMemsetAsync(cand_cnt, 0, size_t(mt) * sizeof(uint32_t), c->stream));
		rxgpu::launch_query_stats(h->metric, qpad, cq, mt, q_stride, h->dim, h->d_stats, q_sq, margin, false, c->stream);

		rxgpu::GemmParams g{};
		g.rows = h->d_rows;
		g.inv_norms = h->d_inv_norms;
		g.row_sq = h->d_row_sq;
		g.queries = qpad;
		g.q_sq = q_sq;
		g.stride = h->stride;
		g.dim = h->dim;
		g.nq = cq;
		g.q_stride = q_stride;
		const uint32_t wg_per_cu = uint32_t(std::max<size_t>(1, std::min<size_t>(4, (160 * 1024) / rxgpu::gemm_lds_bytes(mt))));
		auto grid_for = [&](uint64_t rows) {
			const uint64_t tiles = (rows + 127) / 128;
			return uint32_t(std::max<uint64_t>(1, std::min<uint64_t>(tiles, uint64_t(h->cus) * wg_per_cu)));
		};
		// 2. sample (strided: representative whatever the insertion order)
		g.n = ns;
		g.row_step = uint32_t(std::max<uint64_t>(1, h->count / ns));
		g.dense = static_cast<float*>(c->d_dense.ptr);
		{
			ProfileScope ps(h, "gemm_sample", c->stream);
			RX_HIP(rxgpu::launch_gemm(h->metric, mt, rxgpu::kGemmDense, g, grid_for(ns), c->stream));
		}
		// 3. thresholds
		rxgpu::launch_sample_threshold(g.dense, ns, cq, mt, kk, margin, thr, c->stream);
		// 4. filter pass over the whole corpus
		g.n = h->count;
		g.row_step = 1;
		g.dense = nullptr;
		g.thr = thr;
		g.cand_row = static_cast<uint32_t*>(c->d_cand_row.ptr);
		g.cand_cnt = cand_cnt;
		g.cap = cap;
		{
			ProfileScope ps(h, "gemm", c->stream);
			RX_HIP(rxgpu::launch_gemm(h->metric, mt, rxgpu::kGemmFilter, g, grid_for(h->count), c->stream));
		}
		// 5. exact re-score, 6. exact top-kk
		{
			ProfileScope ps(h, "rescore", c->stream);
			rxgpu::launch_rescore(h->metric, h->d_rows, h->d_inv_norms, qpad, q_stride, h->stride, h->dim, cq, cap, cand_cnt, g.cand_row,
								  static_cast<float*>(c->d_cand_dist.ptr), c->stream);
		}
		rxgpu::launch_merge(static_cast<float*>(c->d_cand_dist.ptr), g.cand_row, cap, kk, cq, d_out_dist + size_t(q0) * kk,
							d_out_row + size_t(q0) * kk, d_out_count ? d_out_count + q0 : nullptr, nullptr, 0, c->stream);
		// overflow fallback, gated on device: exact fused scan only for queries with cand_cnt > cap
		{
			const uint32_t gridx = rxgpu::scan_grid_x(h->count, h->cus);
			const size_t part = size_t(cq) * gridx * kk;
			if (int rc = c->d_part_dist.ensure(part * sizeof(float)); rc) return rc;
			if (int rc = c->d_part_row.ensure(part * sizeof(uint32_t)); rc) return rc;
			rxgpu::ScanParams p{};
			p.rows = h->d_rows;
			p.inv_norms = h->d_inv_norms;
			p.queries = d_queries + size_t(q0) * h->dim;
			p.n = h->count;
			p.stride = h->stride;
			p.dim = h->dim;
			p.kk = kk;
			p.part_dist = static_cast<float*>(c->d_part_dist.ptr);
			p.part_row = static_cast<uint32_t*>(c->d_part_row.ptr);
			p.gate_cnt = cand_cnt;
			p.gate_cap = cap;
			ProfileScope ps(h, "fallback_scan", c->stream);
			rxgpu::launch_scan(h->metric, p, cq, gridx, c->stream);
			rxgpu::launch_merge(p.part_dist, p.part_row, gridx * kk, kk, cq, d_out_dist + size_t(q0) * kk, d_out_row + size_t(q0) * kk,
								d_out_count ? d_out_count + q0 : nullptr, cand_cnt, cap, c->stream);
		}
		RX_HIP(hipGetLastError());
	}
	return RXGPU_OK;
}

// bf16-pruned scan for one .. a few queries (opt-in: RXGPU_SCAN_BF16=1): 2 bytes per element from HBM instead of 4, exact result (knn_scan.hip).
static bool scan_bf16_enabled() {   // read per call: a process can switch it for A/B runs
	const char* e = getenv("RXGPU_SCAN_BF16");
	return e && atoi(e) != 0;
}
constexpr uint32_t kPrunedMaxQueries = 8;
constexpr uint32_t kPrunedCap = 4096;

int enqueue_knn_pruned(rxgpu_index* h, rxgpu_search_ctx* c, const float* d_queries, uint32_t nq, uint32_t kk, float* d_out_dist,
					   uint32_t* d_out_row, uint32_t* d_out_count) {
	if (int rc = ensure_row_stats(h, c->stream); rc) return rc;
	if (int rc = ensure_bf16_shadow(h, c->stream); rc) return rc;
	const uint32_t ld = (h->dim + 63u) & ~63u;
	const uint32_t gridx = rxgpu::scan_grid_x(h->count, h->cus);
	const uint32_t cap = uint32_t(std::min<uint64_t>(kPrunedCap, std::max<uint64_t>(64, (h->count + 63) & ~63ull)));
	if (int rc = c->d_qpad.ensure(size_t(nq) * ld * sizeof(float)); rc) return rc;
	if (int rc = c->d_qstats.ensure(size_t(2) * nq * sizeof(float)); rc) return rc;
	if (int rc = c->d_dense.ensure(size_t(nq) * h->count * sizeof(float)); rc) return rc;
	if (int rc = c->d_part_dist.ensure(size_t(nq) * gridx * kk * sizeof(float)); rc) return rc;
	if (int rc = c->d_part_row.ensure(size_t(nq) * gridx * kk * sizeof(uint32_t)); rc) return rc;
	if (int rc = c->d_top.ensure(size_t(nq) * (2 * kk + 1) * sizeof(uint32_t)); rc) return rc;
	if (int rc = c->d_cand_row.ensure(size_t(nq) * cap * sizeof(uint32_t)); rc) return rc;
	if (int rc = c->d_cand_dist.ensure(size_t(nq) * cap * sizeof(float)); rc) return rc;
	if (int rc = c->d_cand_cnt.ensure(size_t(nq) * sizeof(uint32_t)); rc) return rc;
	float* qpad = static_cast<float*>(c->d_qpad.ptr);
	float* q_sq = static_cast<float*>(c->d_qstats.ptr);
	float* margin = q_sq + nq;
	float* top_dist = static_cast<float*>(c->d_top.ptr);
	uint32_t* top_row = reinterpret_cast<uint32_t*>(top_dist + size_t(nq) * kk);
	uint32_t* top_cnt = top_row + size_t(nq) * kk;
	uint32_t* cand_cnt = static_cast<uint32_t*>(c->d_cand_cnt.ptr);
	RX_HIP(hipMemsetAsync(qpad, 0, size_t(nq) * ld * sizeof(float), c->stream));
	RX_HIP(hipMemcpy2DAsync(qpad, ld * sizeof(float), d_queries, h->dim * sizeof(float), h->dim * sizeof(float), nq, hipMemcpyDeviceToDevice, c->stream));
	RX_HIP(hipMemsetAsync(cand_cnt, 0, size_t(nq) * sizeof(uint32_t), c->stream));
	rxgpu::launch_query_stats(h->metric, qpad, nq, nq, ld, h->dim, h->d_stats, q_sq, margin, true, c->stream);
	rxgpu::ScanBf16Params p{};
	p.sp.inv_norms = h->d_inv_norms;
	p.sp.n = h->count;
	p.sp.kk = kk;
	p.sp.part_dist = static_cast<float*>(c->d_part_dist.ptr);
	p.sp.part_row = static_cast<uint32_t*>(c->d_part_row.ptr);
	p.rows16 = h->d_rows_bf16;
	p.blocked = h->bf16_blocked ? 1u : 0u;
	p.queries32 = qpad;
	p.row_sq = h->d_row_sq;
	p.q_sq = q_sq;
	p.ld = ld;
	p.approx = static_cast<float*>(c->d_dense.ptr);
	{
		ProfileScope ps(h, "scan_bf16", c->stream);
		rxgpu::launch_scan_bf16(h->metric, p, nq, gridx, c->stream);
	}
	rxgpu::launch_merge(p.sp.part_dist, p.sp.part_row, gridx * kk, kk, nq, top_dist, top_row, top_cnt, nullptr, 0, c->stream);
	{
		ProfileScope ps(h, "filter_approx", c->stream);
		rxgpu::launch_filter_approx(p.approx, h->count, top_dist, top_cnt, kk, margin, static_cast<uint32_t*>(c->d_cand_row.ptr), cand_cnt, cap, nq,
									h->cus, c->stream);
	}
	{
		ProfileScope ps(h, "rescore", c->stream);
		rxgpu::launch_rescore(h->metric, h->d_rows, h->d_inv_norms, qpad, ld, h->stride, h->dim, nq, cap, cand_cnt,
							  static_cast<uint32_t*>(c->d_cand_row.ptr), static_cast<float*>(c->d_cand_dist.ptr), c->stream);
	}
	rxgpu::launch_merge(static_cast<float*>(c->d_cand_dist.ptr), static_cast<uint32_t*>(c->d_cand_row.ptr), cap, kk, nq, d_out_dist, d_out_row,
						d_out_count, nullptr, 0, c->stream);
	{   // more rows inside the bound than the list holds (massive ties): exact scan, gated on device
		rxgpu::ScanParams e{};
		e.rows = h->d_rows;
		e.inv_norms = h->d_inv_norms;
		e.queries = d_queries;
		e.n = h->count;
		e.stride = h->stride;
		e.dim = h->dim;
		e.kk = kk;
		e.part_dist = p.sp.part_dist;
		e.part_row = p.sp.part_row;
		e.gate_cnt = cand_cnt;
		e.gate_cap = cap;
		ProfileScope ps(h, "fallback_scan", c->stream);
		rxgpu::launch_scan(h->metric, e, nq, gridx, c->stream);
		rxgpu::launch_merge(e.part_dist, e.part_row, gridx * kk, kk, nq, d_out_dist, d_out_row, d_out_count, cand_cnt, cap, c->stream);
	}
	RX_HIP(hipGetLastError());
	return RXGPU_OK;
}

// Pre-filtered search, kk <= kMaxFusedK2: gather-scan over the row list + the usual merge (rows in the lists are real rows, so the
// merge and everything downstream is unchanged).
int enqueue_knn_subset(rxgpu_index* h, rxgpu_search_ctx* c, const float* d_queries, uint32_t nq, uint32_t kk, const uint32_t* d_ids,
					   uint64_t n_ids, float* d_out_dist, uint32_t* d_out_row, uint32_t* d_out_count) {
	const uint32_t gridx = rxgpu::subset_grid_x(n_ids, h->dim, kk, h->cus);
	const size_t part = size_t(nq) * gridx * kk;
	if (int rc = c->d_part_dist.ensure(part * sizeof(float)); rc) return rc;
	if (int rc = c->d_part_row.ensure(part * sizeof(uint32_t)); rc) return rc;
	rxgpu::ScanParams p{};
	p.rows = h->d_rows;
	p.inv_norms = h->d_inv_norms;
	p.queries = d_queries;
	p.n = n_ids;
	p.stride = h->stride;
	p.dim = h->dim;
	p.kk = kk;
	p.part_dist = static_cast<float*>(c->d_part_dist.ptr);
	p.part_row = static_cast<uint32_t*>(c->d_part_row.ptr);
	{
		ProfileScope ps(h, "scan_subset", c->stream);
		rxgpu::launch_scan_subset(h->metric, p, d_ids, nq, gridx, h->cus, c->stream);
	}
	{
		ProfileScope ps(h, "merge", c->stream);
		rxgpu::launch_merge_lists(p.part_dist, p.part_row, gridx, kk, nq, d_out_dist, d_out_row, d_out_count, c->stream);
	}
	RX_HIP(hipGetLastError());
	return RXGPU_OK;
}

// Host-facing tail shared by rxgpu_search_knn_subset / _bitmap: queries on the host, the row list already in HBM.
int search_subset_host(rxgpu_index* h, rxgpu_search_ctx* c, const float* queries, uint32_t nq, uint32_t kk, const uint32_t* d_ids,
					   uint64_t n_ids, float* out_dist, uint32_t* out_row, uint32_t* out_count) {
	const uint32_t eff = uint32_t(std::min<uint64_t>(kk, n_ids));
	const size_t qbytes = size_t(nq) * h->dim * sizeof(float);
	if (int rc = c->d_queries.ensure(qbytes); rc) return rc;
	RX_HIP(hipMemcpyAsync(c->d_queries.ptr, queries, qbytes, hipMemcpyHostToDevice, c->stream));
	if (eff <= uint32_t(rxgpu::kMaxFusedK2)) {
		if (int rc = c->d_out_dist.ensure(size_t(nq) * eff * sizeof(float)); rc) return rc;
		if (int rc = c->d_out_row.ensure(size_t(nq) * eff * sizeof(uint32_t)); rc) return rc;
		if (int rc = c->d_out_count.ensure(size_t(nq) * sizeof(uint32_t)); rc) return rc;
		if (int rc = enqueue_knn_subset(h, c, static_cast<const float*>(c->d_queries.ptr), nq, eff, d_ids, n_ids,
										static_cast<float*>(c->d_out_dist.ptr), static_cast<uint32_t*>(c->d_out_row.ptr),
										static_cast<uint32_t*>(c->d_out_count.ptr));
			rc)
			return rc;
		RX_HIP(hipMemcpy2DAsync(out_dist, kk * sizeof(float), c->d_out_dist.ptr, eff * sizeof(float), eff * sizeof(float), nq,
								hipMemcpyDeviceToHost, c->stream));
		RX_HIP(hipMemcpy2DAsync(out_row, kk * sizeof(uint32_t), c->d_out_row.ptr, eff * sizeof(uint32_t), eff * sizeof(uint32_t), nq,
								hipMemcpyDeviceToHost, c->stream));
		RX_HIP(hipMemcpyAsync(out_count, c->d_out_count.ptr, size_t(nq) * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
		RX_HIP(hipStreamSynchronize(c->stream));
		return RXGPU_OK;
	}
	// large k: distances of the listed rows + radix select over (dist, position); positions -> rows; final sort of eff entries on the host
	RX_CHECK(n_ids <= (1ull << 28), RXGPU_ERR_PARAMS, "pre-filtered search with k > 128: the row list must not exceed 2^28 entries");
	if (int rc = c->d_misc.ensure(n_ids * sizeof(float)); rc) return rc;
	if (int rc = c->d_select.ensure(rxgpu::select_scratch_bytes(n_ids)); rc) return rc;
	if (int rc = c->d_out_dist.ensure(size_t(eff) * sizeof(float)); rc) return rc;
	if (int rc = c->d_out_row.ensure(size_t(eff) * sizeof(uint32_t)); rc) return rc;
	if (int rc = c->d_part_row.ensure(size_t(eff) * sizeof(uint32_t)); rc) return rc;
	std::vector<float> hd(eff);
	std::vector<uint32_t> hr(eff), order(eff);
	for (uint32_t q = 0; q < nq; ++q) {
		{
			ProfileScope ps(h, "scan_subset", c->stream);
			rxgpu::launch_distances(h->metric, h->d_rows, h->d_inv_norms, static_cast<const float*>(c->d_queries.ptr) + size_t(q) * h->dim,
									h->stride, h->dim, d_ids, uint32_t(n_ids), static_cast<float*>(c->d_misc.ptr), c->stream);
		}
		rxgpu::launch_select_smallest(static_cast<const float*>(c->d_misc.ptr), n_ids, eff, c->d_select.ptr, static_cast<float*>(c->d_out_dist.ptr),
									  static_cast<uint32_t*>(c->d_out_row.ptr), c->stream);
		rxgpu::launch_gather_u32(d_ids, static_cast<const uint32_t*>(c->d_out_row.ptr), eff, static_cast<uint32_t*>(c->d_part_row.ptr), c->stream);
		RX_HIP(hipGetLastError());
		RX_HIP(hipMemcpyAsync(hd.data(), c->d_out_dist.ptr, size_t(eff) * sizeof(float), hipMemcpyDeviceToHost, c->stream));
		RX_HIP(hipMemcpyAsync(hr.data(), c->d_part_row.ptr, size_t(eff) * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
		RX_HIP(hipStreamSynchronize(c->stream));
		std::iota(order.begin(), order.end(), 0u);
		std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return hd[a] < hd[b] || (!(hd[b] < hd[a]) && hr[a] < hr[b]); });
		for (uint32_t i = 0; i < eff; ++i) {
			out_dist[size_t(q) * kk + i] = hd[order[i]];
			out_row[size_t(q) * kk + i] = hr[order[i]];
		}
		out_count[q] = eff;
	}
	return RXGPU_OK;
}

int enqueue_knn(rxgpu_index* h, rxgpu_search_ctx* c, const float* d_queries, uint32_t nq, uint32_t kk, float* d_out_dist,
				uint32_t* d_out_row, uint32_t* d_out_count) {
	if (scan_bf16_enabled() && nq <= kPrunedMaxQueries && !h->bf16_unavailable && rxgpu::scan_bf16_supported((h->dim + 63u) & ~63u)) {
		const int rc = enqueue_knn_pruned(h, c, d_queries, nq, kk, d_out_dist, d_out_row, d_out_count);
		if (!(rc == RXGPU_ERR_NOMEM && h->bf16_unavailable)) return rc;
	}
	if (int(nq) >= batch_min_queries() && nq >= 2) return enqueue_knn_batched(h, c, d_queries, nq, kk, d_out_dist, d_out_row, d_out_count);
	return enqueue_knn_fused(h, c, d_queries, nq, kk, d_out_dist, d_out_row, d_out_count);
}

}  // namespace

extern "C" {

const char* rxgpu_last_error(void) { return rxgpu::g_err.c_str(); }
int rxgpu_abi_version(void) { return RXGPU_ABI_VERSION; }

int rxgpu_device_count(void) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) {
		set_error("hipGetDeviceCount failed");
		return RXGPU_ERR_DEVICE;
	}
	return n;
}

int rxgpu_device_arch(int device, char* name, size_t cap) {
	hipDeviceProp_t prop;
	RX_HIP(hipGetDeviceProperties(&prop, device));
	std::snprintf(name, cap, "%s", prop.gcnArchName);
	return RXGPU_OK;
}

int rxgpu_index_create(int metric, uint32_t dim, uint64_t capacity, int device, rxgpu_index** out) {
	RX_CHECK(out, RXGPU_ERR_PARAMS, "rxgpu_index_create: out is null");
	RX_CHECK(metric >= 0 && metric <= 2, RXGPU_ERR_PARAMS, "rxgpu_index_create: unknown metric");
	RX_CHECK(dim > 0 && dim <= 65535, RXGPU_ERR_PARAMS, "rxgpu_index_create: dimension must be in [1, 65535]");
	RX_CHECK(capacity < 0xFFFFFFFFull, RXGPU_ERR_PARAMS, "rxgpu_index_create: capacity must fit 32-bit rows");
	int ndev = 0;
	RX_HIP(hipGetDeviceCount(&ndev));
	RX_CHECK(device >= 0 && device < ndev, RXGPU_ERR_PARAMS, "rxgpu_index_create: no such device");
	DeviceGuard dg(device);
	RX_CHECK(dg.ok, RXGPU_ERR_DEVICE, "rxgpu_index_create: hipSetDevice failed");
	hipDeviceProp_t prop;
	RX_HIP(hipGetDeviceProperties(&prop, device));
	auto* h = new rxgpu_index();
	h->metric = metric;
	h->dim = dim;
	h->stride = (dim + 3u) & ~3u;
	if (const char* e = std::getenv("RXGPU_ROW_ALIGN")) {   // experiment: rows start on multiples of so many bytes (a 3 KB row then lies in ONE 4 KB page)
		const uint32_t align = uint32_t(std::max(0, atoi(e)));
		if (align >= 16u && (align & (align - 1u)) == 0u) h->stride = uint32_t(((uint64_t(h->stride) * 4u + align - 1u) & ~uint64_t(align - 1u)) / 4u);
	}
	h->device = device;
	h->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
	*out = h;
	if (capacity) {
		if (int rc = rxgpu_index_reserve(h, capacity); rc) {
			delete h;
			*out = nullptr;
			return rc;
		}
	}
	register_live_index(h);
	return RXGPU_OK;
}

void rxgpu_index_destroy(rxgpu_index* h) {
	if (!h) return;
	if (h->shard_set) {
		rxgpu::sharded_destroy(h);
		delete h;
		return;
	}
	unregister_live_index(h);   // from here on an ending thread leaves this index alone
	DeviceGuard dg(h->device);
	(void)rxgpu::device_wait_all(h->device);
	rxgpu::hnsw_server_destroy(h);
	for (auto& kv : h->resident_ctx) h->free_ctx.push_back(kv.second);
	for (auto* c : h->free_ctx) {
		c->release();
		delete c;
	}
	for (auto& kv : h->stream_ctx) {
		kv.second->release();
		delete kv.second;
	}
	for (auto& kv : h->profile) {
		for (auto& ev : kv.second.events) {
			(void)hipEventDestroy(ev.first);
			(void)hipEventDestroy(ev.second);
		}
	}
	if (!h->adopted) {
		if (h->d_rows) (void)hipFree(h->d_rows);
		if (h->d_inv_norms) (void)hipFree(h->d_inv_norms);
	}
	if (h->d_row_sq) (void)hipFree(h->d_row_sq);
	if (h->d_row_ids) (void)hipFree(h->d_row_ids);
	if (h->d_rows_bf16) (void)hipFree(h->d_rows_bf16);
	if (h->d_stats) (void)hipFree(h->d_stats);
	if (h->d_links0) (void)hipFree(h->d_links0);
	if (h->d_upper_off) (void)hipFree(h->d_upper_off);
	if (h->d_upper) (void)hipFree(h->d_upper);
	if (h->d_deleted) (void)hipFree(h->d_deleted);
	if (h->d_codes) (void)hipFree(h->d_codes);
	if (h->d_list_off) (void)hipFree(h->d_list_off);
	if (h->d_list_rows) (void)hipFree(h->d_list_rows);
	if (h->d_corr) (void)hipFree(h->d_corr);
	if (h->d_hnsw_stats) (void)hipFree(h->d_hnsw_stats);
	delete h;
}

int rxgpu_index_reserve(rxgpu_index* h, uint64_t capacity) {
	RX_CHECK(h, RXGPU_ERR_PARAMS, "null index");
	if (h->shard_set) {
		if (capacity == h->capacity) return RXGPU_OK;
		set_error("rxgpu_index_reserve: the capacity of a sharded index is fixed (create a new one)");
		return RXGPU_ERR_LOGIC;
	}
	RX_CHECK(!h->adopted, RXGPU_ERR_LOGIC, "rxgpu_index_reserve: storage is adopted (caller-owned)");
	RX_CHECK(capacity >= h->count, RXGPU_ERR_PARAMS, "Cannot resize, max element is less than the current number of elements");
	RX_CHECK(capacity < 0xFFFFFFFFull, RXGPU_ERR_PARAMS, "capacity must fit 32-bit rows");
	if (capacity == h->capacity) return RXGPU_OK;
	rxgpu::hnsw_server_quiesce(h);   // the resident search kernel reads what changes here
	DeviceGuard dg(h->device);
	float* nrows = nullptr;
	float* nnorm = nullptr;
	if (capacity) {
		RX_HIP(hipMalloc(reinterpret_cast<void**>(&nrows), capacity * h->stride * sizeof(float)));
		if (h->metric == RXGPU_METRIC_COSINE) {
			hipError_t e = hipMalloc(reinterpret_cast<void**>(&nnorm), capacity * sizeof(float));
			if (e != hipSuccess) {
				(void)hipFree(nrows);
				set_error("Not enough memory: failed to allocate norm coefficients");
				return RXGPU_ERR_NOMEM;
			}
		}
		if (h->count) {
			RX_HIP(hipMemcpy(nrows, h->d_rows, h->count * h->stride * sizeof(float), hipMemcpyDeviceToDevice));
			if (nnorm) RX_HIP(hipMemcpy(nnorm, h->d_inv_norms, h->count * sizeof(float), hipMemcpyDeviceToDevice));
		}
	}
	if (h->d_rows) (void)hipFree(h->d_rows);
	if (h->d_inv_norms) (void)hipFree(h->d_inv_norms);
	h->d_rows = nrows;
	h->d_inv_norms = nnorm;
	h->capacity = capacity;
	return RXGPU_OK;
}

int rxgpu_index_upload_rows(rxgpu_index* h, uint64_t first_row, uint64_t n, const float* rows, const float* inv_norms) {
	RX_CHECK(h, RXGPU_ERR_PARAMS, "null index");
	if (h->shard_set) return rxgpu::sharded_upload_rows(h, first_row, n, rows, inv_norms);
	RX_CHECK(!h->adopted, RXGPU_ERR_LOGIC, "rxgpu_index_upload_rows: storage is adopted (caller-owned)");
	if (n == 0) return RXGPU_OK;
	RX_CHECK(rows, RXGPU_ERR_PARAMS, "rxgpu_index_upload_rows: rows is null");
	RX_CHECK(first_row + n <= h->capacity, RXGPU_ERR_PARAMS, "The number of elements exceeds the specified limit");
	RX_CHECK(h->metric != RXGPU_METRIC_COSINE || inv_norms, RXGPU_ERR_PARAMS, "cosine index requires inv_norms");
	DeviceGuard dg(h->device);
	rxgpu::hnsw_server_quiesce(h);   // the resident search kernel reads what changes here
	float* dst = h->d_rows + first_row * h->stride;
	if (h->stride == h->dim) {
		RX_HIP(hipMemcpy(dst, rows, n * h->dim * sizeof(float), hipMemcpyHostToDevice));
	} else {
		RX_HIP(hipMemcpy2D(dst, h->stride * sizeof(float), rows, h->dim * sizeof(float), h->dim * sizeof(float), n, hipMemcpyHostToDevice));
	}
	if (h->metric == RXGPU_METRIC_COSINE) {
		RX_HIP(hipMemcpy(h->d_inv_norms + first_row, inv_norms, n * sizeof(float), hipMemcpyHostToDevice));
	}
	h->count = std::max(h->count, first_row + n);
	// Derived data follows the mutation incrementally (a full recompute streams the whole corpus: 4 ms per 10M x 768 rows).  The row
	// statistics are maxima entering an error BOUND, so folding the new rows in (and never shrinking on deletes) keeps them valid.
	std::lock_guard<std::mutex> lk(h->mtx);
	if (h->stats_valid) {
		if (h->metric == RXGPU_METRIC_L2 && h->row_sq_capacity < first_row + n) {
			h->stats_valid = false;
		} else {
			rxgpu::launch_row_stats(dst, h->d_inv_norms ? h->d_inv_norms + first_row : nullptr, n, h->stride, h->dim,
									h->metric == RXGPU_METRIC_L2 ? h->d_row_sq + first_row : nullptr, h->d_stats, h->cus, nullptr);
		}
	}
	if (h->bf16_valid) {
		if (h->bf16_capacity < first_row + n) {
			h->bf16_valid = false;
		} else {
			const uint32_t ld = (h->dim + 63u) & ~63u;
			rxgpu::launch_to_bf16(dst, n, h->stride, h->dim, h->d_rows_bf16, ld, h->cus, nullptr, first_row, h->bf16_blocked);
		}
	}
	RX_HIP(hipGetLastError());
	RX_HIP(hipStreamSynchronize(nullptr));
	return RXGPU_OK;
}

int rxgpu_index_adopt_device_rows(rxgpu_index* h, const void* d_rows, uint64_t n, uint32_t row_stride, const void* d_inv_norms) {
	RX_CHECK(h, RXGPU_ERR_PARAMS, "null index");
	if (h->shard_set) {
		set_error("rxgpu_index_adopt_device_rows: not available on a sharded index");
		return RXGPU_ERR_LOGIC;
	}
	RX_CHECK(d_rows || n == 0, RXGPU_ERR_PARAMS, "rxgpu_index_adopt_device_rows: d_rows is null");
	RX_CHECK(row_stride >= h->dim && row_stride % 4 == 0, RXGPU_ERR_PARAMS, "row_stride must be >= dim and a multiple of 4 floats");
	RX_CHECK((reinterpret_cast<uintptr_t>(d_rows) & 15) == 0, RXGPU_ERR_PARAMS, "d_rows must be 16-byte aligned");
	RX_CHECK(n < 0xFFFFFFFFull, RXGPU_ERR_PARAMS, "n must fit 32-bit rows");
	RX_CHECK(h->metric != RXGPU_METRIC_COSINE || d_inv_norms || n == 0, RXGPU_ERR_PARAMS, "cosine index requires d_inv_norms");
	DeviceGuard dg(h->device);
	rxgpu::hnsw_server_quiesce(h);   // the resident search kernel reads what changes here
	if (!h->adopted) {
		if (h->d_rows) (void)hipFree(h->d_rows);
		if (h->d_inv_norms) (void)hipFree(h->d_inv_norms);
	}
	h->adopted = true;
	h->d_rows = const_cast<float*>(static_cast<const float*>(d_rows));
	h->d_inv_norms = const_cast<float*>(static_cast<const float*>(d_inv_norms));
	h->stride = row_stride;
	h->capacity = n;
	h->count = n;
	h->stats_valid = false;
	h->bf16_valid = false;
	return RXGPU_OK;
}

int rxgpu_index_move_row(rxgpu_index* h, uint64_t from, uint64_t to) {
	RX_CHECK(h, RXGPU_ERR_PARAMS, "null index");
	if (h->shard_set) return rxgpu::sharded_move_row(h, from, to);
	RX_CHECK(!h->adopted, RXGPU_ERR_LOGIC, "rxgpu_index_move_row: storage is adopted (caller-owned)");
	RX_CHECK(from < h->count && to < h->count, RXGPU_ERR_PARAMS, "rxgpu_index_move_row: row out of range");
	if (from == to) return RXGPU_OK;
	DeviceGuard dg(h->device);
	rxgpu::hnsw_server_quiesce(h);   // the resident search kernel reads what changes here
	RX_HIP(hipMemcpy(h->d_rows + to * h->stride, h->d_rows + from * h->stride, h->stride * sizeof(float), hipMemcpyDeviceToDevice));
	if (h->d_inv_norms) RX_HIP(hipMemcpy(h->d_inv_norms + to, h->d_inv_norms + from, sizeof(float), hipMemcpyDeviceToDevice));
	std::lock_guard<std::mutex> lk(h->mtx);
	if (h->stats_valid && h->metric == RXGPU_METRIC_L2) {
		RX_HIP(hipMemcpy(h->d_row_sq + to, h->d_row_sq + from, sizeof(float), hipMemcpyDeviceToDevice));
	}
	if (h->bf16_valid) {
		const uint32_t ld = (h->dim + 63u) & ~63u;
		rxgpu::launch_shadow_move(h->d_rows_bf16, ld, from, to, h->bf16_blocked, nullptr);
		RX_HIP(hipGetLastError());
		RX_HIP(hipStreamSynchronize(nullptr));
	}
	return RXGPU_OK;
}

int rxgpu_index_download_row(rxgpu_index* h, uint64_t row, float* out_row, float* out_inv_norm) {
	RX_CHECK(h && out_row, RXGPU_ERR_PARAMS, "rxgpu_index_download_row: null argument");
	RX_CHECK(!h->shard_set, RXGPU_ERR_LOGIC, "rxgpu_index_download_row: single-device indexes only");
	RX_CHECK(row < h->count, RXGPU_ERR_PARAMS, "rxgpu_index_download_row: row out of range");
	DeviceGuard dg(h->device);
	RX_HIP(hipMemcpy(out_row, h->d_rows + row * h->stride, h->dim * sizeof(float), hipMemcpyDeviceToHost));
	if (out_inv_norm) {
		*out_inv_norm = 1.0f;
		if (h->d_inv_norms) RX_HIP(hipMemcpy(out_inv_norm, h->d_inv_norms + row, sizeof(float), hipMemcpyDeviceToHost));
	}
	return RXGPU_OK;
}

int rxgpu_index_truncate(rxgpu_index* h, uint64_t count) {
	RX_CHECK(h, RXGPU_ERR_PARAMS, "null index");
	if (h->shard_set) return rxgpu::sharded_truncate(h, count);
	RX_CHECK(count <= h->capacity, RXGPU_ERR_PARAMS, "rxgpu_index_truncate: count exceeds capacity");
	rxgpu::hnsw_server_quiesce(h);   // the resident search kernel reads what changes here
	// shrinking keeps the statistics (upper bounds stay upper bounds) and the shadow (rows past count are never read)
	if (count > h->count) {
		h->stats_valid = false;
		h->bf16_valid = false;
	}
	h->count = count;
	return RXGPU_OK;
}

uint64_t rxgpu_index_count(const rxgpu_index* h) { return h ? h->count : 0; }
uint64_t rxgpu_index_capacity(const rxgpu_index* h) { return h ? h->capacity : 0; }
uint32_t rxgpu_index_dim(const rxgpu_index* h) { return h ? h->dim : 0; }
uint32_t rxgpu_index_row_stride(const rxgpu_index* h) { return h ? h->stride : 0; }
int rxgpu_index_metric(const rxgpu_index* h) { return h ? h->metric : -1; }
int rxgpu_index_device(const rxgpu_index* h) { return h ? h->device : -1; }
uint64_t rxgpu_index_device_bytes(const rxgpu_index* h) {
	if (!h) return 0;
	if (h->shard_set) return rxgpu::sharded_device_bytes(h);
	return h->capacity * h->stride * sizeof(float) + (h->d_inv_norms ? h->capacity * sizeof(float) : 0);
}

int rxgpu_search_knn_device(rxgpu_index* h, const void* d_queries, uint32_t nq, uint32_t kk, void* d_out_dist, void* d_out_row,
							void* d_out_count, void* stream) {
	RX_CHECK(h, RXGPU_ERR_PARAMS, "null index");
	if (h->shard_set) {
		set_error("rxgpu_search_knn_device: not available on a sharded index");
		return RXGPU_ERR_LOGIC;
	}
	RX_CHECK(nq > 0 && d_queries && d_out_dist && d_out_row, RXGPU_ERR_PARAMS, "rxgpu_search_knn_device: null argument");
	RX_CHECK(kk > 0 && kk <= uint32_t(rxgpu::kMaxFusedK), RXGPU_ERR_PARAMS, "rxgpu_search_knn_device: kk must be in [1, 64]");
	RX_CHECK(h->count > 0, RXGPU_ERR_PARAMS, "rxgpu_search_knn_device: index is empty");
	DeviceGuard dg(h->device);
	rxgpu_search_ctx* c = stream_ctx(h, stream);
	return enqueue_knn(h, c, static_cast<const float*>(d_queries), nq, kk, static_cast<float*>(d_out_dist),
					   static_cast<uint32_t*>(d_out_row), static_cast<uint32_t*>(d_out_count));
}

// internal row -> row id table for consumers on the device (the hybrid fusion maps the scan's rows to the planner's row ids there)
int rxgpu_index_upload_row_ids(rxgpu_index* h, uint64_t first_row, uint64_t n, const int32_t* row_ids) {
	RX_CHECK(h, RXGPU_ERR_PARAMS, "null index");
	if (h->shard_set) {
		set_error("rxgpu_index_upload_row_ids: not available on a sharded index");
		return RXGPU_ERR_LOGIC;
	}
	if (n == 0) return RXGPU_OK;
	RX_CHECK(row_ids && first_row + n <= h->capacity, RXGPU_ERR_PARAMS, "rxgpu_index_upload_row_ids: rows out of range");
	DeviceGuard dg(h->device);
	if (h->row_ids_cap < h->capacity) {
		int32_t* grown = nullptr;
		RX_HIP(hipMalloc(reinterpret_cast<void**>(&grown), h->capacity * sizeof(int32_t)));
		if (h->d_row_ids) {
			(void)hipMemcpy(grown, h->d_row_ids, h->row_ids_cap * sizeof(int32_t), hipMemcpyDeviceToDevice);
			(void)hipFree(h->d_row_ids);
		}
		h->d_row_ids = grown;
		h->row_ids_cap = h->capacity;
	}
	RX_HIP(hipMemcpy(h->d_row_ids + first_row, row_ids, n * sizeof(int32_t), hipMemcpyHostToDevice));
	return RXGPU_OK;
}
const void* rxgpu_index_row_ids_device(const rxgpu_index* h) { return h && !h->shard_set ? h->d_row_ids : nullptr; }

// One query, the result LEFT IN HBM: enqueued on the calling thread's resident stream, nothing waited for.  The buffers belong to the index
// and hold this result until the SAME THREAD's next resident search on it (other threads have buffers of their own); a consumer on another
// stream orders itself behind *stream.
int rxgpu_search_knn_resident(rxgpu_index* h, const float* query, uint32_t kk, void** d_dist, void** d_row, void** d_count, void** stream,
							  uint32_t* entries) {
	RX_CHECK(h, RXGPU_ERR_PARAMS, "null index");
	RX_CHECK(query && d_dist && d_row && d_count && stream && entries, RXGPU_ERR_PARAMS, "rxgpu_search_knn_resident: null argument");
	if (h->shard_set) {
		set_error("rxgpu_search_knn_resident: not available on a sharded index");
		return RXGPU_ERR_LOGIC;
	}
	RX_CHECK(kk >= 1 && kk <= uint32_t(rxgpu::kMaxFusedK2), RXGPU_ERR_PARAMS, "rxgpu_search_knn_resident: kk must be in [1, 128]");
	RX_CHECK(h->count > 0, RXGPU_ERR_PARAMS, "rxgpu_search_knn_resident: index is empty");
	DeviceGuard dg(h->device);
	rxgpu_search_ctx* c = nullptr;
	{   // this thread's resident context (created on its first resident search; searches of one thread are sequential)
		std::lock_guard<std::mutex> lk(h->resident_mtx);
		rxgpu_search_ctx*& slot = h->resident_ctx[std::this_thread::get_id()];
		if (!slot) {
			slot = acquire_ctx(h);
			if (slot) t_resident.used.push_back(h->serial);
		}
		c = slot;
	}
	if (!c) return RXGPU_ERR_DEVICE;
	const uint32_t eff = uint32_t(std::min<uint64_t>(kk, h->count));
	const size_t qbytes = size_t(h->dim) * sizeof(float);
	if (int rc = c->d_queries.ensure(qbytes); rc) return rc;
	if (int rc = c->ensure_pinned(qbytes); rc) return rc;
	RX_HIP(hipStreamSynchronize(c->stream));   // the staging copy of the query before this one has been read (normally long ago)
	std::memcpy(c->h_pinned, query, qbytes);
	RX_HIP(hipMemcpyAsync(c->d_queries.ptr, c->h_pinned, qbytes, hipMemcpyHostToDevice, c->stream));
	if (int rc = c->d_out_dist.ensure(size_t(eff) * sizeof(float)); rc) return rc;
	if (int rc = c->d_out_row.ensure(size_t(eff) * sizeof(uint32_t)); rc) return rc;
	if (int rc = c->d_out_count.ensure(sizeof(uint32_t)); rc) return rc;
	auto* run = eff <= uint32_t(rxgpu::kMaxFusedK) ? enqueue_knn : enqueue_knn_fused;
	if (int rc = run(h, c, static_cast<const float*>(c->d_queries.ptr), 1, eff, static_cast<float*>(c->d_out_dist.ptr),
					 static_cast<uint32_t*>(c->d_out_row.ptr), static_cast<uint32_t*>(c->d_out_count.ptr));
		rc)
		return rc;
	*d_dist = c->d_out_dist.ptr;
	*d_row = c->d_out_row.ptr;
	*d_count = c->d_out_count.ptr;
	*stream = c->stream;
	*entries = eff;
	return RXGPU_OK;
}

uint32_t rxgpu_index_resident_contexts(rxgpu_index* h) {
	if (!h || h->shard_set) return 0;
	std::lock_guard<std::mutex> lk(h->resident_mtx);
	return uint32_t(h->resident_ctx.size());
}

int rxgpu_search_knn(rxgpu_index* h, const float* queries, uint32_t nq, uint32_t kk, float* out_dist, uint32_t* out_row,
					 uint32_t* out_count) {
	RX_CHECK(h, RXGPU_ERR_PARAMS, "null index");
	RX_CHECK(queries && out_dist && out_row && out_count, RXGPU_ERR_PARAMS, "rxgpu_search_knn: null argument");
	if (h->shard_set) {
		if (nq == 0) return RXGPU_OK;
		if (h->count == 0 || kk == 0) {   // bruteforce.cc:106-108, as on a single device
			std::fill(out_count, out_count + nq, 0u);
			return RXGPU_OK;
		}
		return rxgpu::sharded_search_knn_impl(h, queries, nq, kk, nullptr, 0, out_dist, out_row, out_count);
	}
	if (nq == 0) return RXGPU_OK;
	if (h->count == 0 || kk == 0) {   // bruteforce.cc:106-108
		std::fill(out_count, out_count + nq, 0u);
		return RXGPU_OK;
	}
	DeviceGuard dg(h->device);
	rxgpu_search_ctx* c = acquire_ctx(h);
	if (!c) return RXGPU_ERR_DEVICE;
	struct Rel {
		rxgpu_index* h;
		rxgpu_search_ctx* c;
		~Rel() { release_ctx(h, c); }
	} rel{h, c};
	const uint32_t eff = uint32_t(std::min<uint64_t>(kk, h->count));
	const size_t qbytes = size_t(nq) * h->dim * sizeof(float);
	if (int rc = c->d_queries.ensure(qbytes); rc) return rc;
	RX_HIP(hipMemcpyAsync(c->d_queries.ptr, queries, qbytes, hipMemcpyHostToDevice, c->stream));

	if (eff <= uint32_t(rxgpu::kMaxFusedK2)) {
		if (int rc = c->d_out_dist.ensure(size_t(nq) * eff * sizeof(float)); rc) return rc;
		if (int rc = c->d_out_row.ensure(size_t(nq) * eff * sizeof(uint32_t)); rc) return rc;
		if (int rc = c->d_out_count.ensure(size_t(nq) * sizeof(uint32_t)); rc) return rc;
		// kk <= 64: fused / batched / pruned dispatch; 64 < kk <= 128 (e.g. hybrid k = 100): the fused scan with two list entries per lane
		auto* run = eff <= uint32_t(rxgpu::kMaxFusedK) ? enqueue_knn : enqueue_knn_fused;
		if (int rc = run(h, c, static_cast<const float*>(c->d_queries.ptr), nq, eff, static_cast<float*>(c->d_out_dist.ptr),
						 static_cast<uint32_t*>(c->d_out_row.ptr), static_cast<uint32_t*>(c->d_out_count.ptr));
			rc)
			return rc;
		if (eff == kk) {
			RX_HIP(hipMemcpyAsync(out_dist, c->d_out_dist.ptr, size_t(nq) * eff * sizeof(float), hipMemcpyDeviceToHost, c->stream));
			RX_HIP(hipMemcpyAsync(out_row, c->d_out_row.ptr, size_t(nq) * eff * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
		} else {
			RX_HIP(hipMemcpy2DAsync(out_dist, kk * sizeof(float), c->d_out_dist.ptr, eff * sizeof(float), eff * sizeof(float), nq,
									hipMemcpyDeviceToHost, c->stream));
			RX_HIP(hipMemcpy2DAsync(out_row, kk * sizeof(uint32_t), c->d_out_row.ptr, eff * sizeof(uint32_t), eff * sizeof(uint32_t), nq,
									hipMemcpyDeviceToHost, c->stream));
		}
		RX_HIP(hipMemcpyAsync(out_count, c->d_out_count.ptr, size_t(nq) * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
		RX_HIP(hipStreamSynchronize(c->stream));
		return RXGPU_OK;
	}

	// large-k path: distance pass + radix select per query, final (dist,row) sort of kk entries on the host
	if (int rc = c->d_misc.ensure(h->count * sizeof(float)); rc) return rc;
	if (int rc = c->d_select.ensure(rxgpu::select_scratch_bytes(h->count)); rc) return rc;
	if (int rc = c->d_out_dist.ensure(size_t(eff) * sizeof(float)); rc) return rc;
	if (int rc = c->d_out_row.ensure(size_t(eff) * sizeof(uint32_t)); rc) return rc;
	const uint32_t gridx = rxgpu::scan_grid_x(h->count, h->cus);
	std::vector<float> hd(eff);
	std::vector<uint32_t> hr(eff), order(eff);
	for (uint32_t q = 0; q < nq; ++q) {
		{
			ProfileScope ps(h, "scan", c->stream);
			rxgpu::launch_all_distances(h->metric, h->d_rows, h->d_inv_norms, static_cast<const float*>(c->d_queries.ptr) + size_t(q) * h->dim,
										h->count, h->stride, h->dim, static_cast<float*>(c->d_misc.ptr), gridx, c->stream);
		}
		{
			ProfileScope ps(h, "select", c->stream);
			rxgpu::launch_select_smallest(static_cast<const float*>(c->d_misc.ptr), h->count, eff, c->d_select.ptr,
										  static_cast<float*>(c->d_out_dist.ptr), static_cast<uint32_t*>(c->d_out_row.ptr), c->stream);
		}
		RX_HIP(hipGetLastError());
		RX_HIP(hipMemcpyAsync(hd.data(), c->d_out_dist.ptr, size_t(eff) * sizeof(float), hipMemcpyDeviceToHost, c->stream));
		RX_HIP(hipMemcpyAsync(hr.data(), c->d_out_row.ptr, size_t(eff) * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
		RX_HIP(hipStreamSynchronize(c->stream));
		std::iota(order.begin(), order.end(), 0u);
		std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
			return hd[a] < hd[b] || (!(hd[b] < hd[a]) && hr[a] < hr[b]);
		});
		for (uint32_t i = 0; i < eff; ++i) {
			out_dist[size_t(q) * kk + i] = hd[order[i]];
			out_row[size_t(q) * kk + i] = hr[order[i]];
		}
		out_count[q] = eff;
	}
	return RXGPU_OK;
}

int rxgpu_search_knn_subset(rxgpu_index* h, const float* queries, uint32_t nq, uint32_t kk, const uint32_t* row_ids, uint64_t n_ids,
							float* out_dist, uint32_t* out_row, uint32_t* out_count) {
	RX_CHECK(h, RXGPU_ERR_PARAMS, "null index");
	RX_CHECK(queries && out_dist && out_row && out_count && (n_ids == 0 || row_ids), RXGPU_ERR_PARAMS, "rxgpu_search_knn_subset: null argument");
	if (h->shard_set) {
		if (nq == 0) return RXGPU_OK;
		RX_CHECK(kk >= 1, RXGPU_ERR_PARAMS, "rxgpu_search_knn_subset: kk must be >= 1");
		if (n_ids == 0) {
			std::fill(out_count, out_count + nq, 0u);
			return RXGPU_OK;
		}
		return rxgpu::sharded_search_knn_impl(h, queries, nq, kk, row_ids, n_ids, out_dist, out_row, out_count);
	}
	if (nq == 0) return RXGPU_OK;
	for (uint64_t i = 0; i < n_ids; ++i) {
		RX_CHECK(row_ids[i] < h->count && (i == 0 || row_ids[i - 1] < row_ids[i]), RXGPU_ERR_PARAMS,
				 "rxgpu_search_knn_subset: row_ids must be strictly increasing and below the row count");
	}
	if (n_ids == 0 || kk == 0) {
		std::fill(out_count, out_count + nq, 0u);
		return RXGPU_OK;
	}
	DeviceGuard dg(h->device);
	rxgpu_search_ctx* c = acquire_ctx(h);
	if (!c) return RXGPU_ERR_DEVICE;
	struct Rel {
		rxgpu_index* h;
		rxgpu_search_ctx* c;
		~Rel() { release_ctx(h, c); }
	} rel{h, c};
	if (int rc = c->d_subset.ensure(n_ids * sizeof(uint32_t)); rc) return rc;
	RX_HIP(hipMemcpyAsync(c->d_subset.ptr, row_ids, n_ids * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
	return search_subset_host(h, c, queries, nq, kk, static_cast<const uint32_t*>(c->d_subset.ptr), n_ids, out_dist, out_row, out_count);
}

int rxgpu_search_knn_bitmap(rxgpu_index* h, const float* queries, uint32_t nq, uint32_t kk, const uint32_t* allowed_words, uint64_t n_words,
							float* out_dist, uint32_t* out_row, uint32_t* out_count, uint64_t* out_allowed) {
	RX_CHECK(h, RXGPU_ERR_PARAMS, "null index");
	if (h->shard_set) {
		set_error("rxgpu_search_knn_bitmap: not available on a sharded index");
		return RXGPU_ERR_LOGIC;
	}
	RX_CHECK(queries && out_dist && out_row && out_count && allowed_words, RXGPU_ERR_PARAMS, "rxgpu_search_knn_bitmap: null argument");
	const uint64_t need_words = (h->count + 31) / 32;
	RX_CHECK(n_words >= need_words, RXGPU_ERR_PARAMS, "rxgpu_search_knn_bitmap: the bitmap must cover every row (ceil(count / 32) words)");
	if (out_allowed) *out_allowed = 0;
	if (nq == 0) return RXGPU_OK;
	if (h->count == 0) {
		std::fill(out_count, out_count + nq, 0u);
		return RXGPU_OK;
	}
	DeviceGuard dg(h->device);
	rxgpu_search_ctx* c = acquire_ctx(h);
	if (!c) return RXGPU_ERR_DEVICE;
	struct Rel {
		rxgpu_index* h;
		rxgpu_search_ctx* c;
		~Rel() { release_ctx(h, c); }
	} rel{h, c};
	const uint32_t tiles = rxgpu::bitmap_tiles(h->count);
	if (int rc = c->d_bitmap.ensure(need_words * sizeof(uint32_t)); rc) return rc;
	if (int rc = c->d_tiles.ensure(size_t(2) * tiles * sizeof(uint32_t) + sizeof(unsigned long long)); rc) return rc;
	uint32_t* tile_scratch = static_cast<uint32_t*>(c->d_tiles.ptr);
	unsigned long long* d_total = reinterpret_cast<unsigned long long*>(tile_scratch + size_t(2) * tiles);   // 8-byte aligned: 2 * tiles words
	RX_HIP(hipMemcpyAsync(c->d_bitmap.ptr, allowed_words, need_words * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
	{
		ProfileScope ps(h, "bitmap", c->stream);
		rxgpu::launch_bitmap_count(static_cast<const uint32_t*>(c->d_bitmap.ptr), h->count, tile_scratch, d_total, c->stream);
	}
	RX_HIP(hipGetLastError());
	unsigned long long total = 0;
	RX_HIP(hipMemcpyAsync(&total, d_total, sizeof(total), hipMemcpyDeviceToHost, c->stream));
	RX_HIP(hipStreamSynchronize(c->stream));
	if (out_allowed) *out_allowed = total;
	if (total == 0 || kk == 0) {
		std::fill(out_count, out_count + nq, 0u);
		return RXGPU_OK;
	}
	if (int rc = c->d_subset.ensure(total * sizeof(uint32_t)); rc) return rc;
	{
		ProfileScope ps(h, "bitmap", c->stream);
		rxgpu::launch_bitmap_expand(static_cast<const uint32_t*>(c->d_bitmap.ptr), h->count, tile_scratch, static_cast<uint32_t*>(c->d_subset.ptr),
									total, c->stream);
	}
	RX_HIP(hipGetLastError());
	return search_subset_host(h, c, queries, nq, kk, static_cast<const uint32_t*>(c->d_subset.ptr), total, out_dist, out_row, out_count);
}

int rxgpu_index_set_lists(rxgpu_index* h, uint32_t nlist, const uint64_t* list_off, const uint32_t* list_rows) {
	RX_CHECK(h && list_off && nlist > 0, RXGPU_ERR_PARAMS, "rxgpu_index_set_lists: null argument");
	if (h->shard_set) {
		set_error("rxgpu_index_set_lists: not available on a sharded index");
		return RXGPU_ERR_LOGIC;
	}
	RX_CHECK(list_off[0] == 0, RXGPU_ERR_PARAMS, "rxgpu_index_set_lists: offsets start at 0");
	for (uint32_t l = 0; l < nlist; ++l) RX_CHECK(list_off[l] <= list_off[l + 1], RXGPU_ERR_PARAMS, "rxgpu_index_set_lists: offsets must not decrease");
	const uint64_t total = list_off[nlist];
	RX_CHECK(total <= h->count && (total == 0 || list_rows), RXGPU_ERR_PARAMS, "rxgpu_index_set_lists: more listed rows than the index holds");
	for (uint64_t i = 0; i < total; ++i) RX_CHECK(list_rows[i] < h->count, RXGPU_ERR_PARAMS, "rxgpu_index_set_lists: row out of range");
	DeviceGuard dg(h->device);
	RX_HIP(rxgpu::device_wait_all(h->device));
	if (h->d_list_off) (void)hipFree(h->d_list_off);
	if (h->d_list_rows) (void)hipFree(h->d_list_rows);
	h->d_list_off = nullptr;
	h->d_list_rows = nullptr;
	h->nlist = 0;
	RX_HIP(hipMalloc(reinterpret_cast<void**>(&h->d_list_off), (size_t(nlist) + 1) * sizeof(uint64_t)));
	RX_HIP(hipMemcpy(h->d_list_off, list_off, (size_t(nlist) + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
	if (total) {
		RX_HIP(hipMalloc(reinterpret_cast<void**>(&h->d_list_rows), total * sizeof(uint32_t)));
		RX_HIP(hipMemcpy(h->d_list_rows, list_rows, total * sizeof(uint32_t), hipMemcpyHostToDevice));
	}
	h->nlist = nlist;
	h->lists_rows = total;
	h->lists_count = h->count;
	return RXGPU_OK;
}

namespace {
// The probed lists of one query as an ascending row list in c->d_subset, everything on the device: nprobe nearest centroids (the coarse
// quantiser's search; up to 128 lists its result never leaves HBM, wider probes fetch the list ids — nprobe words — and send them back),
// lists -> allowed-rows bitmap -> row list.  *total = rows to scan.
int ivf_probe_rows(rxgpu_index* h, rxgpu_index* coarse, rxgpu_search_ctx* c, const float* query, uint32_t nprobe, unsigned long long* total, const char* who) {
	const size_t qbytes = size_t(h->dim) * sizeof(float);
	if (int rc = c->d_queries.ensure(qbytes); rc) return rc;
	RX_HIP(hipMemcpyAsync(c->d_queries.ptr, query, qbytes, hipMemcpyHostToDevice, c->stream));
	const size_t o_lists = 0, o_dist = size_t(nprobe) * 4, o_cnt = o_dist + size_t(nprobe) * 4;
	if (int rc = c->d_ivf.ensure(o_cnt + 256); rc) return rc;
	char* ivf = static_cast<char*>(c->d_ivf.ptr);
	if (nprobe <= uint32_t(rxgpu::kMaxFusedK2)) {
		rxgpu_search_ctx* cc = stream_ctx(coarse, c->stream);
		auto* run = nprobe <= uint32_t(rxgpu::kMaxFusedK) ? enqueue_knn : enqueue_knn_fused;
		if (int rc = run(coarse, cc, static_cast<const float*>(c->d_queries.ptr), 1, nprobe, reinterpret_cast<float*>(ivf + o_dist),
						 reinterpret_cast<uint32_t*>(ivf + o_lists), reinterpret_cast<uint32_t*>(ivf + o_cnt));
			rc)
			return rc;
	} else {
		std::vector<float> cd(nprobe);
		std::vector<uint32_t> cl(nprobe);
		uint32_t cnt = 0;
		if (int rc = rxgpu_search_knn(coarse, query, 1, nprobe, cd.data(), cl.data(), &cnt); rc) return rc;
		RX_HIP(hipMemcpyAsync(ivf + o_lists, cl.data(), size_t(cnt) * 4, hipMemcpyHostToDevice, c->stream));
		RX_HIP(hipMemcpyAsync(ivf + o_cnt, &cnt, 4, hipMemcpyHostToDevice, c->stream));
		RX_HIP(hipStreamSynchronize(c->stream));   // cl / cnt live on this frame
	}
	const uint64_t need_words = (h->count + 31) / 32;
	const uint32_t tiles = rxgpu::bitmap_tiles(h->count);
	if (int rc = c->d_bitmap.ensure(need_words * sizeof(uint32_t)); rc) return rc;
	if (int rc = c->d_tiles.ensure(size_t(2) * tiles * sizeof(uint32_t) + sizeof(unsigned long long)); rc) return rc;
	uint32_t* tile_scratch = static_cast<uint32_t*>(c->d_tiles.ptr);
	unsigned long long* d_total = reinterpret_cast<unsigned long long*>(tile_scratch + size_t(2) * tiles);
	RX_HIP(hipMemsetAsync(c->d_bitmap.ptr, 0, need_words * sizeof(uint32_t), c->stream));
	{
		ProfileScope ps(h, "ivf_lists", c->stream);
		rxgpu::launch_ivf_mark_lists(reinterpret_cast<const uint32_t*>(ivf + o_lists), reinterpret_cast<const uint32_t*>(ivf + o_cnt), nprobe, h->d_list_off,
									 h->d_list_rows, static_cast<uint32_t*>(c->d_bitmap.ptr), c->stream);
		rxgpu::launch_bitmap_count(static_cast<const uint32_t*>(c->d_bitmap.ptr), h->count, tile_scratch, d_total, c->stream);
	}
	RX_HIP(hipGetLastError());
	*total = 0;
	RX_HIP(hipMemcpyAsync(total, d_total, sizeof(*total), hipMemcpyDeviceToHost, c->stream));
	RX_HIP(hipStreamSynchronize(c->stream));
	if (*total == 0) return RXGPU_OK;
	if (int rc = c->d_subset.ensure(*total * sizeof(uint32_t)); rc) return rc;
	{
		ProfileScope ps(h, "ivf_lists", c->stream);
		rxgpu::launch_bitmap_expand(static_cast<const uint32_t*>(c->d_bitmap.ptr), h->count, tile_scratch, static_cast<uint32_t*>(c->d_subset.ptr), *total,
									c->stream);
	}
	RX_HIP(hipGetLastError());
	(void)who;
	return RXGPU_OK;
}
int ivf_check(rxgpu_index* h, rxgpu_index* coarse, const char* who) {
	if (h->shard_set || coarse->shard_set) {
		set_error(std::string(who) + ": not available on a sharded index");
		return RXGPU_ERR_LOGIC;
	}
	RX_CHECK(h->nlist > 0 && h->lists_count == h->count, RXGPU_ERR_LOGIC, std::string(who) + ": inverted lists are not set / out of date");
	RX_CHECK(coarse->count == h->nlist && coarse->dim == h->dim && coarse->device == h->device, RXGPU_ERR_PARAMS,
			 std::string(who) + ": the coarse index must hold one centroid per list, same dimension, same device");
	return RXGPU_OK;
}
}  // namespace

int rxgpu_search_knn_lists(rxgpu_index* h, rxgpu_index* coarse, const float* query, uint32_t nprobe, uint32_t kk, float* out_dist,
						   uint32_t* out_row, uint32_t* out_count, uint64_t* out_scanned) {
	RX_CHECK(h && coarse && query && out_dist && out_row && out_count, RXGPU_ERR_PARAMS, "rxgpu_search_knn_lists: null argument");
	if (int rc = ivf_check(h, coarse, "rxgpu_search_knn_lists"); rc) return rc;
	if (out_scanned) *out_scanned = 0;
	*out_count = 0;
	if (h->count == 0 || kk == 0) return RXGPU_OK;
	nprobe = std::max<uint32_t>(1, std::min<uint32_t>(nprobe, h->nlist));
	DeviceGuard dg(h->device);
	rxgpu_search_ctx* c = acquire_ctx(h);
	if (!c) return RXGPU_ERR_DEVICE;
	struct Rel {
		rxgpu_index* h;
		rxgpu_search_ctx* c;
		~Rel() { release_ctx(h, c); }
	} rel{h, c};
	unsigned long long total = 0;
	if (int rc = ivf_probe_rows(h, coarse, c, query, nprobe, &total, "rxgpu_search_knn_lists"); rc) return rc;
	if (out_scanned) *out_scanned = total;
	if (total == 0) return RXGPU_OK;
	return search_subset_host(h, c, query, 1, kk, static_cast<const uint32_t*>(c->d_subset.ptr), total, out_dist, out_row, out_count);
}

namespace {
// range search over a row list that lies in c->d_subset (query in c->d_queries): the tail of rxgpu_search_range_subset
int range_subset_on_device(rxgpu_index* h, rxgpu_search_ctx* c, uint64_t n_ids, float radius, int inclusive, float* out_dist, uint32_t* out_row, uint64_t cap,
						   uint64_t* out_total, const char* who) {
	const uint64_t dcap = std::min<uint64_t>(cap, n_ids);
	if (int rc = c->d_out_dist.ensure(std::max<uint64_t>(dcap, 1) * sizeof(float)); rc) return rc;
	if (int rc = c->d_out_row.ensure(std::max<uint64_t>(dcap, 1) * sizeof(uint32_t)); rc) return rc;
	if (int rc = c->d_out_count.ensure(sizeof(unsigned long long)); rc) return rc;
	RX_HIP(hipMemsetAsync(c->d_out_count.ptr, 0, sizeof(unsigned long long), c->stream));
	{
		ProfileScope ps(h, "range_subset", c->stream);
		rxgpu::launch_range_subset(h->metric, h->d_rows, h->d_inv_norms, static_cast<const float*>(c->d_queries.ptr),
								   static_cast<const uint32_t*>(c->d_subset.ptr), n_ids, h->stride, h->dim, radius, inclusive,
								   static_cast<float*>(c->d_out_dist.ptr), static_cast<uint32_t*>(c->d_out_row.ptr), dcap,
								   static_cast<unsigned long long*>(c->d_out_count.ptr), rxgpu::scan_grid_x(n_ids, h->cus), c->stream);
	}
	RX_HIP(hipGetLastError());
	unsigned long long total = 0;
	RX_HIP(hipMemcpyAsync(&total, c->d_out_count.ptr, sizeof(total), hipMemcpyDeviceToHost, c->stream));
	RX_HIP(hipStreamSynchronize(c->stream));
	*out_total = total;
	if (total > cap) {
		set_error(std::string(who) + ": output buffer too small");
		return RXGPU_ERR_OVERFLOW;
	}
	if (total == 0) return RXGPU_OK;
	std::vector<float> hd(total);
	std::vector<uint32_t> hr(total), order(total);
	RX_HIP(hipMemcpy(hd.data(), c->d_out_dist.ptr, total * sizeof(float), hipMemcpyDeviceToHost));
	RX_HIP(hipMemcpy(hr.data(), c->d_out_row.ptr, total * sizeof(uint32_t), hipMemcpyDeviceToHost));
	std::iota(order.begin(), order.end(), 0u);
	std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return hd[a] < hd[b] || (!(hd[b] < hd[a]) && hr[a] < hr[b]); });
	for (uint64_t i = 0; i < total; ++i) {
		out_dist[i] = hd[order[i]];
		out_row[i] = hr[order[i]];
	}
	return RXGPU_OK;
}
}  // namespace

int rxgpu_search_range_lists(rxgpu_index* h, rxgpu_index* coarse, const float* query, uint32_t nprobe, float radius, int inclusive, float* out_dist,
							 uint32_t* out_row, uint64_t cap, uint64_t* out_total, uint64_t* out_scanned) {
	RX_CHECK(h && coarse && query && out_total && (cap == 0 || (out_dist && out_row)), RXGPU_ERR_PARAMS, "rxgpu_search_range_lists: null argument");
	if (int rc = ivf_check(h, coarse, "rxgpu_search_range_lists"); rc) return rc;
	if (out_scanned) *out_scanned = 0;
	*out_total = 0;
	if (h->count == 0) return RXGPU_OK;
	nprobe = std::max<uint32_t>(1, std::min<uint32_t>(nprobe, h->nlist));
	DeviceGuard dg(h->device);
	rxgpu_search_ctx* c = acquire_ctx(h);
	if (!c) return RXGPU_ERR_DEVICE;
	struct Rel {
		rxgpu_index* h;
		rxgpu_search_ctx* c;
		~Rel() { release_ctx(h, c); }
	} rel{h, c};
	unsigned long long total = 0;
	if (int rc = ivf_probe_rows(h, coarse, c, query, nprobe, &total, "rxgpu_search_range_lists"); rc) return rc;
	if (out_scanned) *out_scanned = total;
	if (total == 0) return RXGPU_OK;
	return range_subset_on_device(h, c, total, radius, inclusive, out_dist, out_row, cap, out_total, "rxgpu_search_range_lists");
}

int rxgpu_search_knn_subset_device(rxgpu_index* h, const void* d_queries, uint32_t nq, uint32_t kk, const void* d_row_ids, uint64_t n_ids,
								   void* d_out_dist, void* d_out_row, void* d_out_count, void* stream) {
	RX_CHECK(h, RXGPU_ERR_PARAMS, "null index");
	if (h->shard_set) {
		set_error("rxgpu_search_knn_subset_device: not available on a sharded index");
		return RXGPU_ERR_LOGIC;
	}
	RX_CHECK(nq > 0 && d_queries && d_row_ids && d_out_dist && d_out_row, RXGPU_ERR_PARAMS, "rxgpu_search_knn_subset_device: null argument");
	RX_CHECK(kk > 0 && kk <= uint32_t(rxgpu::kMaxFusedK2), RXGPU_ERR_PARAMS, "rxgpu_search_knn_subset_device: kk must be in [1, 128]");
	RX_CHECK(n_ids > 0 && n_ids <= h->count, RXGPU_ERR_PARAMS, "rxgpu_search_knn_subset_device: the row list must hold 1..count entries");
	DeviceGuard dg(h->device);
	rxgpu_search_ctx* c = stream_ctx(h, stream);
	return enqueue_knn_subset(h, c, static_cast<const float*>(d_queries), nq, kk, static_cast<const uint32_t*>(d_row_ids), n_ids,
							  static_cast<float*>(d_out_dist), static_cast<uint32_t*>(d_out_row), static_cast<uint32_t*>(d_out_count));
}

int rxgpu_check_row_list_device(rxgpu_index* h, const void* d_row_ids, uint64_t n_ids, void* stream, int32_t* out_ok) {
	RX_CHECK(h && out_ok && (n_ids == 0 || d_row_ids), RXGPU_ERR_PARAMS, "rxgpu_check_row_list_device: null argument");
	*out_ok = 1;
	if (n_ids == 0) return RXGPU_OK;
	DeviceGuard dg(h->device);
	rxgpu_search_ctx* c = stream_ctx(h, stream);
	if (int rc = c->d_tiles.ensure(sizeof(uint32_t)); rc) return rc;
	RX_HIP(hipMemsetAsync(c->d_tiles.ptr, 0, sizeof(uint32_t), c->stream));
	rxgpu::launch_check_row_list(static_cast<const uint32_t*>(d_row_ids), n_ids, h->count, static_cast<uint32_t*>(c->d_tiles.ptr), h->cus, c->stream);
	RX_HIP(hipGetLastError());
	uint32_t bad = 0;
	RX_HIP(hipMemcpyAsync(&bad, c->d_tiles.ptr, sizeof(bad), hipMemcpyDeviceToHost, c->stream));
	RX_HIP(hipStreamSynchronize(c->stream));
	*out_ok = bad ? 0 : 1;
	return RXGPU_OK;
}

int rxgpu_merge_shards_device(const void* d_gathered, uint32_t world, uint32_t nq, uint32_t kk, uint32_t shard_rows, void* d_out_dist,
							  void* d_out_row, void* d_out_count, void* stream) {
	RX_CHECK(d_gathered && d_out_dist && d_out_row, RXGPU_ERR_PARAMS, "rxgpu_merge_shards_device: null argument");
	RX_CHECK(world >= 1 && nq >= 1 && kk >= 1 && kk <= uint32_t(rxgpu::kMaxFusedK), RXGPU_ERR_PARAMS, "rxgpu_merge_shards_device: bad shape");
	RX_CHECK(uint64_t(world) * shard_rows <= 0xFFFFFFFEull, RXGPU_ERR_PARAMS, "rxgpu_merge_shards_device: global rows must fit 32 bits");
	rxgpu::launch_merge_shards(static_cast<const uint32_t*>(d_gathered), world, nq, kk, shard_rows, static_cast<float*>(d_out_dist),
							   static_cast<uint32_t*>(d_out_row), static_cast<uint32_t*>(d_out_count), static_cast<hipStream_t>(stream));
	RX_HIP(hipGetLastError());
	return RXGPU_OK;
}

int rxgpu_search_range(rxgpu_index* h, const float* query, float radius, int inclusive, float* out_dist, uint32_t* out_row, uint64_t cap,
					   uint64_t* out_total) {
	RX_CHECK(h, RXGPU_ERR_PARAMS, "null index");
	RX_CHECK(query && out_total && (cap == 0 || (out_dist && out_row)), RXGPU_ERR_PARAMS, "rxgpu_search_range: null argument");
	if (h->shard_set) {
		*out_total = 0;
		return rxgpu::sharded_search_range_impl(h, query, radius, inclusive, nullptr, 0, out_dist, out_row, cap, out_total);
	}
	*out_total = 0;
	if (h->count == 0) return RXGPU_OK;   // bruteforce.cc:132-134
	DeviceGuard dg(h->device);
	rxgpu_search_ctx* c = acquire_ctx(h);
	if (!c) return RXGPU_ERR_DEVICE;
	struct Rel {
		rxgpu_index* h;
		rxgpu_search_ctx* c;
		~Rel() { release_ctx(h, c); }
	} rel{h, c};
	const uint64_t dcap = std::min<uint64_t>(cap, h->count);
	if (int rc = c->d_queries.ensure(h->dim * sizeof(float)); rc) return rc;
	if (int rc = c->d_out_dist.ensure(std::max<uint64_t>(dcap, 1) * sizeof(float)); rc) return rc;
	if (int rc = c->d_out_row.ensure(std::max<uint64_t>(dcap, 1) * sizeof(uint32_t)); rc) return rc;
	if (int rc = c->d_out_count.ensure(sizeof(unsigned long long)); rc) return rc;
	RX_HIP(hipMemcpyAsync(c->d_queries.ptr, query, h->dim * sizeof(float), hipMemcpyHostToDevice, c->stream));
	RX_HIP(hipMemsetAsync(c->d_out_count.ptr, 0, sizeof(unsigned long long), c->stream));
	{
		ProfileScope ps(h, "range", c->stream);
		rxgpu::launch_range(h->metric, h->d_rows, h->d_inv_norms, static_cast<const float*>(c->d_queries.ptr), h->count, h->stride, h->dim,
							radius, inclusive, static_cast<float*>(c->d_out_dist.ptr), static_cast<uint32_t*>(c->d_out_row.ptr), dcap,
							static_cast<unsigned long long*>(c->d_out_count.ptr), rxgpu::scan_grid_x(h->count, h->cus), c->stream);
	}
	RX_HIP(hipGetLastError());
	unsigned long long total = 0;
	RX_HIP(hipMemcpyAsync(&total, c->d_out_count.ptr, sizeof(total), hipMemcpyDeviceToHost, c->stream));
	RX_HIP(hipStreamSynchronize(c->stream));
	*out_total = total;
	if (total > cap) {
		set_error("rxgpu_search_range: output buffer too small");
		return RXGPU_ERR_OVERFLOW;
	}
	if (total == 0) return RXGPU_OK;
	std::vector<float> hd(total);
	std::vector<uint32_t> hr(total), order(total);
	RX_HIP(hipMemcpy(hd.data(), c->d_out_dist.ptr, total * sizeof(float), hipMemcpyDeviceToHost));
	RX_HIP(hipMemcpy(hr.data(), c->d_out_row.ptr, total * sizeof(uint32_t), hipMemcpyDeviceToHost));
	std::iota(order.begin(), order.end(), 0u);
	std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return hd[a] < hd[b] || (!(hd[b] < hd[a]) && hr[a] < hr[b]); });
	for (uint64_t i = 0; i < total; ++i) {
		out_dist[i] = hd[order[i]];
		out_row[i] = hr[order[i]];
	}
	return RXGPU_OK;
}

int rxgpu_search_range_subset(rxgpu_index* h, const float* query, float radius, int inclusive, const uint32_t* row_ids, uint64_t n_ids,
							  float* out_dist, uint32_t* out_row, uint64_t cap, uint64_t* out_total) {
	RX_CHECK(h, RXGPU_ERR_PARAMS, "null index");
	RX_CHECK(query && out_total && (cap == 0 || (out_dist && out_row)) && (n_ids == 0 || row_ids), RXGPU_ERR_PARAMS,
			 "rxgpu_search_range_subset: null argument");
	if (h->shard_set) {
		*out_total = 0;
		if (n_ids == 0) return RXGPU_OK;
		return rxgpu::sharded_search_range_impl(h, query, radius, inclusive, row_ids, n_ids, out_dist, out_row, cap, out_total);
	}
	*out_total = 0;
	for (uint64_t i = 0; i < n_ids; ++i) {
		RX_CHECK(row_ids[i] < h->count && (i == 0 || row_ids[i - 1] < row_ids[i]), RXGPU_ERR_PARAMS,
				 "rxgpu_search_range_subset: row_ids must be strictly increasing and below the row count");
	}
	if (n_ids == 0) return RXGPU_OK;
	DeviceGuard dg(h->device);
	rxgpu_search_ctx* c = acquire_ctx(h);
	if (!c) return RXGPU_ERR_DEVICE;
	struct Rel {
		rxgpu_index* h;
		rxgpu_search_ctx* c;
		~Rel() { release_ctx(h, c); }
	} rel{h, c};
	if (int rc = c->d_queries.ensure(h->dim * sizeof(float)); rc) return rc;
	if (int rc = c->d_subset.ensure(n_ids * sizeof(uint32_t)); rc) return rc;
	RX_HIP(hipMemcpyAsync(c->d_queries.ptr, query, h->dim * sizeof(float), hipMemcpyHostToDevice, c->stream));
	RX_HIP(hipMemcpyAsync(c->d_subset.ptr, row_ids, n_ids * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
	return range_subset_on_device(h, c, n_ids, radius, inclusive, out_dist, out_row, cap, out_total, "rxgpu_search_range_subset");
}

int rxgpu_distances(rxgpu_index* h, const float* query, const uint32_t* rows, uint32_t n, float* out_dist) {
	RX_CHECK(h, RXGPU_ERR_PARAMS, "null index");
	RX_CHECK(query && (n == 0 || (rows && out_dist)), RXGPU_ERR_PARAMS, "rxgpu_distances: null argument");
	if (n == 0) return RXGPU_OK;
	if (h->shard_set) return rxgpu::sharded_distances(h, query, rows, n, out_dist);
	for (uint32_t i = 0; i < n; ++i) RX_CHECK(rows[i] < h->count, RXGPU_ERR_PARAMS, "rxgpu_distances: row out of range");
	DeviceGuard dg(h->device);
	rxgpu_search_ctx* c = acquire_ctx(h);
	if (!c) return RXGPU_ERR_DEVICE;
	struct Rel {
		rxgpu_index* h;
		rxgpu_search_ctx* c;
		~Rel() { release_ctx(h, c); }
	} rel{h, c};
	if (int rc = c->d_queries.ensure(h->dim * sizeof(float)); rc) return rc;
	if (int rc = c->d_out_row.ensure(size_t(n) * sizeof(uint32_t)); rc) return rc;
	if (int rc = c->d_out_dist.ensure(size_t(n) * sizeof(float)); rc) return rc;
	RX_HIP(hipMemcpyAsync(c->d_queries.ptr, query, h->dim * sizeof(float), hipMemcpyHostToDevice, c->stream));
	RX_HIP(hipMemcpyAsync(c->d_out_row.ptr, rows, size_t(n) * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
	{
		ProfileScope ps(h, "distances", c->stream);
		rxgpu::launch_distances(h->metric, h->d_rows, h->d_inv_norms, static_cast<const float*>(c->d_queries.ptr), h->stride, h->dim,
							static_cast<const uint32_t*>(c->d_out_row.ptr), n, static_cast<float*>(c->d_out_dist.ptr), c->stream);
	}
	RX_HIP(hipGetLastError());
	RX_HIP(hipMemcpyAsync(out_dist, c->d_out_dist.ptr, size_t(n) * sizeof(float), hipMemcpyDeviceToHost, c->stream));
	RX_HIP(hipStreamSynchronize(c->stream));
	return RXGPU_OK;
}

/* ------------------------------------------------------------------------------------------------ HNSW */

int rxgpu_hnsw_attach_graph(rxgpu_index* h, const uint32_t* links0, const uint64_t* upper_off, const uint32_t* upper, uint64_t upper_blocks,
							const uint8_t* deleted, uint32_t M, uint32_t max_m0, int32_t maxlevel, uint32_t entry, uint64_t num_deleted) {
	RX_CHECK(h, RXGPU_ERR_PARAMS, "null index");
	if (h->shard_set) {
		set_error("rxgpu_hnsw_attach_graph: a sharded index holds one graph per shard — attach to the rxgpu_index_shard(h, s) handles");
		return RXGPU_ERR_LOGIC;
	}
	const uint64_t n = h->count;
	RX_CHECK(n == 0 || (links0 && upper_off && deleted), RXGPU_ERR_PARAMS, "rxgpu_hnsw_attach_graph: null argument");
	RX_CHECK(upper_blocks == 0 || upper, RXGPU_ERR_PARAMS, "rxgpu_hnsw_attach_graph: upper is null");
	RX_CHECK(M >= 1 && max_m0 <= uint32_t(rxgpu::kHnswMaxNeighbors) && M <= max_m0, RXGPU_ERR_PARAMS,
			 "rxgpu_hnsw_attach_graph: the GPU engine supports M <= 64 (2*M <= 128)");
	RX_CHECK(n == 0 || entry < n, RXGPU_ERR_PARAMS, "rxgpu_hnsw_attach_graph: entry point out of range");
	DeviceGuard dg(h->device);
	RX_HIP(rxgpu::device_wait_all(h->device));
	// allocated for the index CAPACITY (and with headroom for upper-level blocks), so that rxgpu_hnsw_patch_graph can grow the graph in place
	auto replace = [&](auto*& dst, const void* src, size_t bytes, size_t cap_bytes) -> int {
		if (dst) (void)hipFree(dst);
		dst = nullptr;
		if (cap_bytes == 0) return RXGPU_OK;
		RX_HIP(hipMalloc(reinterpret_cast<void**>(&dst), cap_bytes));
		if (bytes) RX_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
		return RXGPU_OK;
	};
	h->graph_attached = false;
	const uint64_t rows_cap = std::max<uint64_t>(std::max<uint64_t>(h->capacity, n), 1);
	// expected upper blocks of a full index: sum over levels of cap / M^level = cap / (M - 1); twice that (and at least what is there) as headroom
	const uint64_t upper_cap = std::max<uint64_t>(2 * upper_blocks + 64, 2 * rows_cap / std::max<uint32_t>(M - 1, 1) + 64);
	const size_t row_bytes = (1 + size_t(max_m0)) * sizeof(uint32_t), blk_bytes = (1 + size_t(M)) * sizeof(uint32_t);
	if (int rc = replace(h->d_links0, links0, n * row_bytes, rows_cap * row_bytes); rc) return rc;
	if (int rc = replace(h->d_upper_off, upper_off, n ? (n + 1) * sizeof(uint64_t) : 0, (rows_cap + 1) * sizeof(uint64_t)); rc) return rc;
	if (int rc = replace(h->d_upper, upper, upper_blocks * blk_bytes, upper_cap * blk_bytes); rc) return rc;
	if (int rc = replace(h->d_deleted, deleted, n, rows_cap); rc) return rc;
	h->graph_rows_cap = rows_cap;
	h->graph_upper_cap = upper_cap;
	h->graph_upper_used = upper_blocks;
	if (!h->d_hnsw_stats) {
		RX_HIP(hipMalloc(reinterpret_cast<void**>(&h->d_hnsw_stats), 8 * sizeof(unsigned long long)));   // evals, hops, in-kernel restarts, spare, [4..7] phase cycles (RXGPU_HNSW_PHASES builds)
		RX_HIP(hipMemset(h->d_hnsw_stats, 0, 8 * sizeof(unsigned long long)));
	}
	h->graph_n = n;
	h->graph_M = M;
	h->graph_maxM0 = max_m0;
	h->graph_maxlevel = maxlevel;
	h->graph_entry = entry;
	h->graph_deleted = num_deleted;
	h->graph_attached = true;
	return RXGPU_OK;
}

int rxgpu_hnsw_patch_graph(rxgpu_index* h, uint32_t n_dirty, const uint32_t* dirty_ids, const uint32_t* links0_rows, const uint8_t* deleted_flags,
						   const int32_t* levels, const uint32_t* upper_rows, int32_t maxlevel, uint32_t entry, uint64_t num_deleted) {
	RX_CHECK(h && h->graph_attached, RXGPU_ERR_LOGIC, "rxgpu_hnsw_patch_graph: no graph attached");
	rxgpu::hnsw_server_quiesce(h);   // the resident search kernel reads what changes here
	RX_CHECK(n_dirty == 0 || (dirty_ids && links0_rows && deleted_flags && levels), RXGPU_ERR_PARAMS, "rxgpu_hnsw_patch_graph: null argument");
	const uint64_t n_new = h->count;   // the rows were uploaded first (rxgpu_index_upload_rows)
	RX_CHECK(n_new >= h->graph_n, RXGPU_ERR_LOGIC, "rxgpu_hnsw_patch_graph: the index shrank under the graph");
	RX_CHECK(n_new == 0 || entry < n_new, RXGPU_ERR_PARAMS, "rxgpu_hnsw_patch_graph: entry point out of range");
	if (n_new > h->graph_rows_cap) {
		set_error("rxgpu_hnsw_patch_graph: the graph arrays were allocated for fewer rows (re-attach the graph)");
		return RXGPU_ERR_OVERFLOW;
	}
	// staging: ids, per-node upper placement, lists — one buffer, one upload, one scatter launch
	const uint32_t M = h->graph_M, max_m0 = h->graph_maxM0;
	const size_t stride0 = 1 + size_t(max_m0), stride = 1 + size_t(M);
	std::vector<uint64_t> upper_at(n_dirty);
	std::vector<uint32_t> upper_src(n_dirty);
	uint64_t used = h->graph_upper_used, staged_blocks = 0, next_new = h->graph_n;
	for (uint32_t j = 0; j < n_dirty; ++j) {
		RX_CHECK(dirty_ids[j] < n_new && levels[j] >= 0, RXGPU_ERR_PARAMS, "rxgpu_hnsw_patch_graph: node id / level out of range");
		upper_src[j] = uint32_t(staged_blocks);
		staged_blocks += uint64_t(levels[j]);
		if (dirty_ids[j] >= h->graph_n) {   // a new node: ids ascending, every one of them listed (its blocks are appended in id order)
			RX_CHECK(dirty_ids[j] == next_new, RXGPU_ERR_PARAMS, "rxgpu_hnsw_patch_graph: new nodes must be listed in ascending id order, none skipped");
			++next_new;
			upper_at[j] = used;
			used += uint64_t(levels[j]);
		} else {
			upper_at[j] = ~uint64_t(0);
		}
	}
	RX_CHECK(next_new == n_new, RXGPU_ERR_PARAMS, "rxgpu_hnsw_patch_graph: every new node must be listed");
	RX_CHECK(staged_blocks == 0 || upper_rows, RXGPU_ERR_PARAMS, "rxgpu_hnsw_patch_graph: upper_rows is null");
	if (used > h->graph_upper_cap) {
		set_error("rxgpu_hnsw_patch_graph: upper-level storage exhausted (re-attach the graph)");
		return RXGPU_ERR_OVERFLOW;
	}
	DeviceGuard dg(h->device);
	rxgpu_search_ctx* c = acquire_ctx(h);
	if (!c) return RXGPU_ERR_DEVICE;
	struct Rel {
		rxgpu_index* h;
		rxgpu_search_ctx* c;
		~Rel() { release_ctx(h, c); }
	} rel{h, c};
	RX_HIP(rxgpu::device_wait_all(h->device));   // no search may be reading the lists while they change (the Map calls this under its writer lock)
	auto al = [](size_t v) { return (v + 255) & ~size_t(255); };
	const size_t o_ids = 0, o_at = al(o_ids + size_t(n_dirty) * 4), o_src = al(o_at + size_t(n_dirty) * 8), o_lv = al(o_src + size_t(n_dirty) * 4),
				 o_l0 = al(o_lv + size_t(n_dirty) * 4), o_up = al(o_l0 + size_t(n_dirty) * stride0 * 4), o_del = al(o_up + staged_blocks * stride * 4),
				 total = al(o_del + n_dirty);
	if (n_dirty) {
		if (int rc = c->d_misc.ensure(total); rc) return rc;
		if (int rc = c->ensure_pinned(total); rc) return rc;
		char* hp = static_cast<char*>(c->h_pinned);
		std::memcpy(hp + o_ids, dirty_ids, size_t(n_dirty) * 4);
		std::memcpy(hp + o_at, upper_at.data(), size_t(n_dirty) * 8);
		std::memcpy(hp + o_src, upper_src.data(), size_t(n_dirty) * 4);
		std::memcpy(hp + o_lv, levels, size_t(n_dirty) * 4);
		std::memcpy(hp + o_l0, links0_rows, size_t(n_dirty) * stride0 * 4);
		if (staged_blocks) std::memcpy(hp + o_up, upper_rows, staged_blocks * stride * 4);
		std::memcpy(hp + o_del, deleted_flags, n_dirty);
		char* db = static_cast<char*>(c->d_misc.ptr);
		RX_HIP(hipMemcpyAsync(db, hp, total, hipMemcpyHostToDevice, c->stream));
		rxgpu::HnswPatch p{};
		p.ids = reinterpret_cast<const uint32_t*>(db + o_ids);
		p.upper_at = reinterpret_cast<const uint64_t*>(db + o_at);
		p.upper_src = reinterpret_cast<const uint32_t*>(db + o_src);
		p.levels = reinterpret_cast<const int32_t*>(db + o_lv);
		p.src_links0 = reinterpret_cast<const uint32_t*>(db + o_l0);
		p.src_upper = reinterpret_cast<const uint32_t*>(db + o_up);
		p.src_deleted = reinterpret_cast<const uint8_t*>(db + o_del);
		p.links0 = h->d_links0;
		p.upper_off = h->d_upper_off;
		p.upper = h->d_upper;
		p.deleted = h->d_deleted;
		p.M = M;
		p.maxM0 = max_m0;
		rxgpu::launch_hnsw_patch(p, n_dirty, c->stream);
		RX_HIP(hipGetLastError());
		RX_HIP(hipStreamSynchronize(c->stream));
	}
	h->graph_upper_used = used;
	h->graph_n = n_new;
	h->graph_maxlevel = maxlevel;
	h->graph_entry = entry;
	h->graph_deleted = num_deleted;
	return RXGPU_OK;
}

int rxgpu_hnsw_update_deleted(rxgpu_index* h, const uint8_t* deleted, uint64_t num_deleted) {
	RX_CHECK(h && h->graph_attached, RXGPU_ERR_LOGIC, "rxgpu_hnsw_update_deleted: no graph attached");
	RX_CHECK(deleted || h->graph_n == 0, RXGPU_ERR_PARAMS, "rxgpu_hnsw_update_deleted: null argument");
	DeviceGuard dg(h->device);
	RX_HIP(rxgpu::device_wait_all(h->device));
	if (h->graph_n) RX_HIP(hipMemcpy(h->d_deleted, deleted, h->graph_n, hipMemcpyHostToDevice));
	h->graph_deleted = num_deleted;
	return RXGPU_OK;
}

// Rows [first_row, first_row + n) of the code table: the device side of a point added to / updated in a quantised graph (addPoint with a
// quantizer, hnswalg.h:1480-1495).  The table is allocated for the index CAPACITY (like the rows), so an upsert is a copy of its own D + 4
// bytes, not a re-upload of the table.
int rxgpu_hnsw_upload_sq8_rows(rxgpu_index* h, uint64_t first_row, uint64_t n, const uint8_t* codes, const float* corr, float alpha_2) {
	RX_CHECK(h, RXGPU_ERR_PARAMS, "null index");
	if (h->shard_set) {
		set_error("rxgpu_hnsw_upload_sq8_rows: not available on a sharded index");
		return RXGPU_ERR_LOGIC;
	}
	if (n == 0) return RXGPU_OK;
	RX_CHECK(codes && corr, RXGPU_ERR_PARAMS, "rxgpu_hnsw_upload_sq8_rows: null argument");
	RX_CHECK(first_row <= h->sq8_n && first_row + n <= h->count, RXGPU_ERR_PARAMS, "rxgpu_hnsw_upload_sq8_rows: rows follow the table without a hole and stay below count");
	DeviceGuard dg(h->device);
	const uint64_t cap = std::max<uint64_t>(h->capacity, h->count);
	if (h->sq8_cap < first_row + n) {   // first rows, or the index was reserved larger since: a new table, the old rows copied over
		RX_HIP(rxgpu::device_wait_all(h->device));
		uint8_t* nc = nullptr;
		float* nr = nullptr;
		RX_HIP(hipMalloc(reinterpret_cast<void**>(&nc), size_t(cap) * h->dim + 4));
		if (hipMalloc(reinterpret_cast<void**>(&nr), size_t(cap) * sizeof(float)) != hipSuccess) {
			(void)hipFree(nc);
			set_error("rxgpu_hnsw_upload_sq8_rows: not enough memory for the corrective offsets");
			return RXGPU_ERR_NOMEM;
		}
		if (h->sq8_n) {
			RX_HIP(hipMemcpy(nc, h->d_codes, size_t(h->sq8_n) * h->dim, hipMemcpyDeviceToDevice));
			RX_HIP(hipMemcpy(nr, h->d_corr, size_t(h->sq8_n) * sizeof(float), hipMemcpyDeviceToDevice));
		}
		if (h->d_codes) (void)hipFree(h->d_codes);
		if (h->d_corr) (void)hipFree(h->d_corr);
		h->d_codes = nc;
		h->d_corr = nr;
		h->sq8_cap = cap;
	}
	RX_HIP(hipMemcpy(h->d_codes + size_t(first_row) * h->dim, codes, size_t(n) * h->dim, hipMemcpyHostToDevice));
	RX_HIP(hipMemcpy(h->d_corr + first_row, corr, size_t(n) * sizeof(float), hipMemcpyHostToDevice));
	h->sq8_alpha2 = alpha_2;
	h->sq8_n = std::max<uint64_t>(h->sq8_n, first_row + n);
	return RXGPU_OK;
}

int rxgpu_hnsw_attach_sq8(rxgpu_index* h, const uint8_t* codes, const float* corr, uint64_t count, float alpha_2) {
	RX_CHECK(h, RXGPU_ERR_PARAMS, "null index");
	if (h->shard_set) {
		set_error("rxgpu_hnsw_attach_sq8: not available on a sharded index");
		return RXGPU_ERR_LOGIC;
	}
	RX_CHECK(count == h->count, RXGPU_ERR_PARAMS, "rxgpu_hnsw_attach_sq8: one code row per index row");
	RX_CHECK(count == 0 || (codes && corr), RXGPU_ERR_PARAMS, "rxgpu_hnsw_attach_sq8: null argument");
	DeviceGuard dg(h->device);
	RX_HIP(rxgpu::device_wait_all(h->device));
	if (h->d_codes) (void)hipFree(h->d_codes);
	if (h->d_corr) (void)hipFree(h->d_corr);
	h->d_codes = nullptr;
	h->d_corr = nullptr;
	h->sq8_n = 0;
	h->sq8_cap = 0;
	if (count) {
		// sized by the index capacity: rows added later are patched in (rxgpu_hnsw_upload_sq8_rows)
		// + 4 bytes: the word loads of the last row's last block stay inside the allocation whatever dim % 4 is
		const uint64_t cap = std::max<uint64_t>(h->capacity, count);
		RX_HIP(hipMalloc(reinterpret_cast<void**>(&h->d_codes), size_t(cap) * h->dim + 4));
		RX_HIP(hipMalloc(reinterpret_cast<void**>(&h->d_corr), size_t(cap) * sizeof(float)));
		h->sq8_cap = cap;
		RX_HIP(hipMemcpy(h->d_codes, codes, size_t(count) * h->dim, hipMemcpyHostToDevice));
		RX_HIP(hipMemcpy(h->d_corr, corr, size_t(count) * sizeof(float), hipMemcpyHostToDevice));
	}
	h->sq8_alpha2 = alpha_2;
	h->sq8_n = count;
	return RXGPU_OK;
}

// One body for both row formats: `queries` are float rows (qcorr == nullptr) or SQ8 codes with their corrective offsets and normCoefs.
// sink (sharded HNSW, rxgpu_sharded.hip): the result lists stay in HBM — packed into the shard's slot of the exchange's send buffer
// ([nq][kk] distances | [nq][kk] local rows, invalid entries past a query's count) instead of travelling to the host; only the counts
// come back (the re-run tiers are driven by them).  The stream is drained before the call returns.
static int hnsw_search_impl(rxgpu_index* h, const void* queries, const float* qcorr, const float* qnorm, uint32_t nq, uint32_t k, uint32_t ef,
							float* out_dist, uint32_t* out_row, uint32_t* out_count, const rxgpu::HnswSink* sink = nullptr, bool try_server = true);

extern "C++" {
// The A/B and test hooks of the HNSW search (RXGPU_HNSW_*), read in ONE pass over the environment per call — a dozen getenv() lookups each
// walked the whole environment, on the path of every single-query SearchKnn.  (Still the process environment: tests flip the hooks between
// calls.  Not safe against a concurrent setenv, like getenv itself.)
struct HnswKnobs {
	const char* visited = nullptr;       // RXGPU_HNSW_VISITED = bitset | hash
	int visited_log2 = -1;               // RXGPU_HNSW_VISITED_LOG2
	int visited_lds = -1;                // RXGPU_HNSW_VISITED_LDS
	int split_upload = -1;               // RXGPU_HNSW_SPLIT_UPLOAD
	int prefetch = -1;                   // RXGPU_HNSW_PREFETCH
	int lds_cand_cap = -1;               // RXGPU_HNSW_LDS_CAND_CAP
	int helper = -1;                     // RXGPU_HNSW_HELPER
	int restart_cand = -1;               // RXGPU_HNSW_RESTART_CAND
	int sorted = -1;                     // RXGPU_HNSW_SORTED
	int gcand_cap = -1;                  // RXGPU_HNSW_GCAND_CAP
	int team = -1;                       // RXGPU_HNSW_TEAM: wavefronts per search of a small launch (1 = off)
	int team_max = -1;                   // RXGPU_HNSW_TEAM_MAX: searches per launch up to which the team form is used
	int zero_copy = -1;                  // RXGPU_HNSW_ZERO_COPY = 0: small calls copy their queries / results like large ones
	int nbl = -1;                        // RXGPU_HNSW_NBL = 1: team searches fetch the link blocks of a hop's rows along with the rows (an experiment, off by default)
	int spec = -1;                       // RXGPU_HNSW_SPEC = 1: team searches also evaluate the next candidate's neighbours in the hop's distance trip (an experiment, off by default)
	int server = -1;                     // RXGPU_HNSW_SERVER = 0: single queries take a launch each (no resident kernel)
	int server_slots = -1, server_idle_us = -1, server_life_ms = -1;   // RXGPU_HNSW_SERVER_SLOTS / _IDLE_US / _LIFE_MS
	bool names_a_kernel = false;         // a hook that picks a kernel form is set: the resident kernel (one form) stands aside
};
static HnswKnobs read_hnsw_knobs() {
	HnswKnobs k;
	static const char kPrefix[] = "RXGPU_HNSW_";
	for (char** e = environ; e && *e; ++e) {
		const char* s = *e;
		if (s[0] != 'R' || std::strncmp(s, kPrefix, sizeof(kPrefix) - 1) != 0) continue;
		const char* name = s + sizeof(kPrefix) - 1;
		const char* eq = std::strchr(name, '=');
		if (!eq) continue;
		const size_t n = size_t(eq - name);
		const char* val = eq + 1;
		auto is = [&](const char* want) { return std::strlen(want) == n && std::strncmp(name, want, n) == 0; };
		if (is("VISITED")) k.visited = val;
		else if (is("VISITED_LOG2")) k.visited_log2 = atoi(val);
		else if (is("VISITED_LDS")) k.visited_lds = atoi(val);
		else if (is("SPLIT_UPLOAD")) k.split_upload = atoi(val);
		else if (is("PREFETCH")) k.prefetch = atoi(val);
		else if (is("LDS_CAND_CAP")) k.lds_cand_cap = atoi(val);
		else if (is("HELPER")) k.helper = atoi(val);
		else if (is("RESTART_CAND")) k.restart_cand = atoi(val);
		else if (is("SORTED")) k.sorted = atoi(val);
		else if (is("GCAND_CAP")) k.gcand_cap = atoi(val);
		else if (is("TEAM")) k.team = atoi(val);
		else if (is("TEAM_MAX")) k.team_max = atoi(val);
		else if (is("ZERO_COPY")) k.zero_copy = atoi(val);
		else if (is("SPEC")) k.spec = atoi(val);
		else if (is("NBL")) k.nbl = atoi(val);
		else if (is("SERVER")) k.server = atoi(val);
		else if (is("SERVER_SLOTS")) k.server_slots = atoi(val);
		else if (is("SERVER_IDLE_US")) k.server_idle_us = atoi(val);
		else if (is("SERVER_LIFE_MS")) k.server_life_ms = atoi(val);
		else continue;
		if (!is("SERVER") && !is("SERVER_SLOTS") && !is("SERVER_IDLE_US") && !is("SERVER_LIFE_MS") && !is("SPLIT_UPLOAD") && !is("HELPER") && !is("SPEC") && !is("NBL")) k.names_a_kernel = true;
	}
	return k;
}
}  // extern "C++"

// 1: the index's resident kernel answered; 0: it does not take this query (the caller launches); otherwise an error code
static int hnsw_try_server(rxgpu_index* h, const HnswKnobs& knobs, const float* query, uint32_t k, uint32_t ef, float* out_dist, uint32_t* out_row,
						   uint32_t* out_count) {
	if (knobs.server == 0 || knobs.names_a_kernel) return 0;
	rxgpu::HnswServerConfig cfg;
	if (knobs.server_slots > 0) cfg.slots = uint32_t(knobs.server_slots);
	if (knobs.server_idle_us > 0) cfg.idle_us = uint32_t(knobs.server_idle_us);
	if (knobs.server_life_ms > 0) cfg.life_ms = uint32_t(knobs.server_life_ms);
	cfg.spec = knobs.spec > 0;
	cfg.nbl = knobs.nbl > 0;
	return rxgpu::hnsw_server_search(h, cfg, query, k, ef, out_dist, out_row, out_count);
}

int rxgpu_hnsw_search_knn_posted(rxgpu_index* h, const float* query, uint32_t k, uint32_t ef, float* out_dist, uint32_t* out_row, uint32_t* out_count,
								 int32_t* served) {
	RX_CHECK(h && query && out_dist && out_row && out_count && served, RXGPU_ERR_PARAMS, "rxgpu_hnsw_search_knn_posted: null argument");
	*served = 0;
	if (h->shard_set || h->count == 0 || k == 0 || !h->graph_attached || h->graph_n != h->count) return RXGPU_OK;   // the launching call says what is wrong
	k = uint32_t(std::min<uint64_t>(k, h->count));
	if (!ef) ef = k * 3 / 2;
	if (!ef) ef = 1;
	const int rc = hnsw_try_server(h, read_hnsw_knobs(), query, k, ef, out_dist, out_row, out_count);
	if (rc == 1) {
		*served = 1;
		return RXGPU_OK;
	}
	if (rc == 2) {   // the mailbox took it and the search needs the re-run tiers: answered here by the launches, not offered to the mailbox again
		const int r2 = hnsw_search_impl(h, query, nullptr, nullptr, 1, k, ef, out_dist, out_row, out_count, nullptr, false);
		if (r2 == RXGPU_OK) *served = 1;
		return r2;
	}
	return rc;
}

int rxgpu_hnsw_server_times(rxgpu_index* h, uint64_t* device_us, uint64_t* caller_us) {
	RX_CHECK(h && device_us && caller_us, RXGPU_ERR_PARAMS, "rxgpu_hnsw_server_times: null argument");
	rxgpu::hnsw_server_times(h, device_us, caller_us);
	return RXGPU_OK;
}

int rxgpu_hnsw_server_stats(rxgpu_index* h, uint64_t* served, uint64_t* generations) {
	RX_CHECK(h && served && generations, RXGPU_ERR_PARAMS, "rxgpu_hnsw_server_stats: null argument");
	rxgpu::hnsw_server_counters(h, served, generations);
	return RXGPU_OK;
}

int rxgpu_hnsw_search_knn(rxgpu_index* h, const float* queries, uint32_t nq, uint32_t k, uint32_t ef, float* out_dist, uint32_t* out_row,
						  uint32_t* out_count) {
	RX_CHECK(h, RXGPU_ERR_PARAMS, "null index");
	RX_CHECK(queries, RXGPU_ERR_PARAMS, "rxgpu_hnsw_search_knn: null argument");
	if (h->shard_set) {   // SURVEY 8(e) "HNSW": a graph per shard, the per-shard results meet in the same all-gather + merge as brute force
		RX_CHECK(out_dist && out_row && out_count, RXGPU_ERR_PARAMS, "rxgpu_hnsw_search_knn: null argument");
		return rxgpu::sharded_hnsw_search_knn(h, queries, nullptr, nullptr, nq, k, ef, out_dist, out_row, out_count);
	}
	return hnsw_search_impl(h, queries, nullptr, nullptr, nq, k, ef, out_dist, out_row, out_count);
}

extern "C++" {
namespace rxgpu {
int hnsw_search_to_sink(rxgpu_index* shard, const void* queries, const float* qcorr, const float* qnorm, uint32_t nq, uint32_t k, uint32_t ef, const HnswSink& sink) {
	std::vector<uint32_t> counts(nq);
	if (qcorr && !(shard->d_codes && shard->sq8_n == shard->count)) {
		set_error("rxgpu_hnsw_search_knn_sq8: SQ8 codes are not attached / out of date on a shard");
		return RXGPU_ERR_LOGIC;
	}
	return hnsw_search_impl(shard, queries, qcorr, qnorm, nq, k, ef, nullptr, nullptr, counts.data(), &sink);
}
}  // namespace rxgpu
}  // extern "C++"

int rxgpu_hnsw_search_knn_sq8(rxgpu_index* h, const uint8_t* query_codes, const float* query_corr, const float* query_norm_coef, uint32_t nq,
							  uint32_t k, uint32_t ef, float* out_dist, uint32_t* out_row, uint32_t* out_count) {
	RX_CHECK(h, RXGPU_ERR_PARAMS, "null index");
	RX_CHECK(query_codes && query_corr && query_norm_coef, RXGPU_ERR_PARAMS, "rxgpu_hnsw_search_knn_sq8: null argument");
	if (h->shard_set) {   // every shard searches the code table attached to ITS handle (rxgpu_hnsw_attach_sq8 on rxgpu_index_shard(h, s)); same exchange
		RX_CHECK(out_dist && out_row && out_count, RXGPU_ERR_PARAMS, "rxgpu_hnsw_search_knn_sq8: null argument");
		for (uint32_t s = 0; s < rxgpu_index_shard_count(h); ++s) {
			const rxgpu_index* sh = rxgpu_index_shard(h, s);
			RX_CHECK(sh->count == 0 || (sh->d_codes && sh->sq8_n == sh->count), RXGPU_ERR_LOGIC, "rxgpu_hnsw_search_knn_sq8: SQ8 codes are not attached / out of date on a shard");
		}
		return rxgpu::sharded_hnsw_search_knn(h, query_codes, query_corr, query_norm_coef, nq, k, ef, out_dist, out_row, out_count);
	}
	RX_CHECK(h->count == 0 || (h->d_codes && h->sq8_n == h->count), RXGPU_ERR_LOGIC,
			 "rxgpu_hnsw_search_knn_sq8: SQ8 codes are not attached / out of date");
	return hnsw_search_impl(h, query_codes, query_corr, query_norm_coef, nq, k, ef, out_dist, out_row, out_count);
}

static int hnsw_search_impl(rxgpu_index* h, const void* queries, const float* qcorr, const float* qnorm, uint32_t nq, uint32_t k, uint32_t ef,
							float* out_dist, uint32_t* out_row, uint32_t* out_count, const rxgpu::HnswSink* sink, bool try_server) {
	const bool sq8 = qcorr != nullptr;
	const bool to_host = sink == nullptr;
	RX_CHECK(h, RXGPU_ERR_PARAMS, "null index");
	if (h->shard_set) {
		set_error("rxgpu_hnsw_search_knn_sq8: not available on a sharded index");
		return RXGPU_ERR_LOGIC;
	}
	RX_CHECK(queries && out_count && (sink || (out_dist && out_row)), RXGPU_ERR_PARAMS, "rxgpu_hnsw_search_knn: null argument");
	if (nq == 0) return RXGPU_OK;
	if (h->count == 0 || k == 0) {   // hnswalg.h:1989-1991
		RX_CHECK(to_host, RXGPU_ERR_LOGIC, "rxgpu_hnsw_search_knn: an empty shard has no list to send");
		std::fill(out_count, out_count + nq, 0u);
		return RXGPU_OK;
	}
	RX_CHECK(h->graph_attached && h->graph_n == h->count, RXGPU_ERR_LOGIC, "rxgpu_hnsw_search_knn: graph is not attached / out of date");
	k = uint32_t(std::min<uint64_t>(k, h->count));
	if (!ef) ef = k * 3 / 2;                                        // hnswalg.h:1995
	if (!ef) ef = 1;
	RX_CHECK(ef <= uint32_t(rxgpu::kHnswMaxEf), RXGPU_ERR_PARAMS, "rxgpu_hnsw_search_knn: ef must be <= 4096 on the GPU engine");
	// ef > 1024: the result heap alone takes the LDS budget of a search — the candidate heap goes to global scratch from the start
	const bool big_ef = ef > uint32_t(rxgpu::kHnswLdsCandEf);
	const HnswKnobs knobs = read_hnsw_knobs();
	// ONE query, the planner's call: through the mailbox of the index's resident kernel (rxgpu_hnsw_server.hip) — no launch on the path
	if (nq == 1 && to_host && !sq8 && try_server) {
		const int served = hnsw_try_server(h, knobs, static_cast<const float*>(queries), k, ef, out_dist, out_row, out_count);
		if (served == 1) return RXGPU_OK;
		if (served != 0 && served != 2) return served;   // (2: the search ran there and came back flagged — the tiers below answer it)
	}
	DeviceGuard dg(h->device);
	rxgpu_search_ctx* c = acquire_ctx(h);
	if (!c) return RXGPU_ERR_DEVICE;
	struct Rel {
		rxgpu_index* h;
		rxgpu_search_ctx* c;
		~Rel() { release_ctx(h, c); }
	} rel{h, c};
	const uint64_t words = (h->count + 31) / 32;
	// visited bitsets are the memory hog (N / 8 bytes per resident search): a launch gets an eighth of the free HBM for them, between 2 and
	// 16 GiB.  (A fixed 2 GiB held a 10M-node index to 1717 searches per launch — fewer than the chip keeps resident.)
	// (a handful of searches never comes near the budget: no driver call on the path of a single-query SearchKnn)
	size_t free_b = 0, total_b = 0;
	if (nq > 256 && hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
	const uint64_t visited_budget = std::min<uint64_t>(16ull << 30, std::max<uint64_t>(2ull << 30, (uint64_t(free_b) + c->d_visited.bytes) / 8));
	const uint64_t max_slots = std::max<uint64_t>(1, std::min<uint64_t>(32768, visited_budget / (words * 4)));
	// The visited set of the first pass (and of the tie re-runs) is a HASH SET sized by ef, zeroed by the search itself — not a bitset over
	// the nodes zeroed by a memset: 2^k words >= 64 ef (8192 words = 32 KB at ef = 128; a search may fill half: 4096 nodes, against the
	// 850 - 2300 it tests at 1M - 10M rows) instead of N / 8 bytes (1.25 MB per search at 10M rows: 20 GB of memset in front of a 16 384-query
	// launch).  Searches that would outgrow it come back as kHnswOverflow and take the global-heap re-run, which keeps the bitset.
	// Graphs so small that the bitset is the smaller of the two keep it.  RXGPU_HNSW_VISITED=bitset: the former path (A/B, tests).
	uint32_t vis_hash_log2 = 12;
	while ((1ull << vis_hash_log2) < 64ull * ef && vis_hash_log2 < 18) ++vis_hash_log2;
	if (knobs.visited_log2 >= 0) vis_hash_log2 = uint32_t(std::min(20, std::max(6, knobs.visited_log2)));   // test hook: force overflows
	// a handful of searches (the latency form of the kernel, at most two workgroups per CU): the same hash set in LDS, whatever the rule
	// below picks for batches — the launcher decides (launch_hnsw_nb).  RXGPU_HNSW_VISITED_LDS=0: off (A/B)
	uint32_t vis_lds_log2 = vis_hash_log2;
	if (knobs.visited_lds == 0) vis_lds_log2 = 0;
	if (knobs.visited) vis_lds_log2 = 0;   // an explicit choice of the global form (A/B, tests) stands for every launch
	{
		const char* e = knobs.visited;   // "bitset" / "hash": force one of the two (A/B, tests on small graphs)
		const bool force_hash = e && std::strcmp(e, "hash") == 0;
		// Which one by default: the hash set costs a second dependent trip on the hops where a lane's first slot is taken (measured at 1M x 768,
		// ef = 128, same box and graph: 1.28 - 1.33 M q/s against 1.43 - 1.46 M on the bitset, profiles/rd4f_hnsw_visited_ab.txt); the bitset
		// costs its memset (N / 8 bytes per query) and, once the bitsets of the searches in flight outgrow the Infinity Cache, an HBM round trip per
		// test.  The hash set takes over where one search's bitset is 16 x its hash set or more (4.2 M nodes at ef = 128).
		if ((e && std::strcmp(e, "bitset") == 0) || (!force_hash && (16ull << vis_hash_log2) > words)) vis_hash_log2 = 0;
		// in HBM the set gets twice the words (a quarter full at most): fewer second probes — 10M x 768, one graph and box, 16 384 queries:
		// 2^13 words 469 k q/s kernels only, 2^14 498 k, 2^15 497 k, 2^16 483 k, bitset 472 k (profiles/rd4j_hnsw_10m_*.json)
		if (vis_hash_log2 && knobs.visited_log2 < 0 && vis_hash_log2 < 18) vis_hash_log2 += 1;
	}
	const uint64_t vis_words = vis_hash_log2 ? (1ull << vis_hash_log2) : words;   // per search of the first pass
	const uint64_t vis_slots = vis_hash_log2 ? std::max<uint64_t>(1, std::min<uint64_t>(32768, visited_budget / (vis_words * 4))) : max_slots;
	// SQ8 queries: [codes, padded to 4 bytes][corr][normCoef] in the one query buffer
	const size_t qelem = sq8 ? sizeof(uint8_t) : sizeof(float);
	const size_t qbytes = size_t(nq) * h->dim * qelem;
	const size_t o_qcorr = (qbytes + 255) & ~size_t(255), o_qnorm = o_qcorr + ((size_t(nq) * 4 + 255) & ~size_t(255));
	if (int rc = c->d_queries.ensure(sq8 ? o_qnorm + size_t(nq) * 4 : qbytes); rc) return rc;
	if (int rc = c->d_out_dist.ensure(size_t(nq) * k * sizeof(float)); rc) return rc;
	if (int rc = c->d_out_row.ensure(size_t(nq) * k * sizeof(uint32_t)); rc) return rc;
	if (int rc = c->d_out_count.ensure(size_t(nq) * sizeof(uint32_t)); rc) return rc;
	// the bitsets of the first launch (N / 8 bytes per search: 2 GB for 16 384 searches over 1M nodes) are zeroed on a second stream and
	// enqueued BEFORE the upload of the query block (a copy from pageable memory keeps this thread until it is staged): the two overlap,
	// the launch waits for both
	bool first_zeroed = false;
	if (!big_ef && !vis_hash_log2) {
		const uint32_t cq = uint32_t(std::min<uint64_t>(vis_slots, nq));
		const size_t zero_bytes = size_t(cq) * words * 4;
		if (zero_bytes >= (size_t(8) << 20)) {
			if (int rc = c->d_visited.ensure(zero_bytes); rc) return rc;
			if (int rc = c->ensure_aux(); rc) return rc;
			RX_HIP(hipMemsetAsync(c->d_visited.ptr, 0, zero_bytes, c->aux_stream));
			RX_HIP(hipEventRecord(c->aux_done, c->aux_stream));
			first_zeroed = true;
		}
	}
	// A large batch in ONE launch is searched in two halves on two streams: the upload of the second half of the query block (a copy from
	// pageable memory keeps this thread until it is staged) runs while the first half's searches have started; the halves overlap on the
	// device like the workgroups of one launch.  RXGPU_HNSW_SPLIT_UPLOAD=0: one upload, one launch.
	// Round 6: four parts from 2048 queries on (the first launch waits for a quarter of the block, three quarters of the upload run under
	// searches), the parts alternating between the two streams; RXGPU_HNSW_SPLIT_UPLOAD = n: that many parts (0 / 1: one upload, one launch).
	bool split_upload = !big_ef && nq >= 2048 && uint64_t(nq) <= vis_slots;
	uint32_t split_parts = 4;
	if (knobs.split_upload >= 0) {
		split_upload = split_upload && knobs.split_upload > 1;
		split_parts = uint32_t(std::min(16, std::max(2, knobs.split_upload)));
	}
	const uint32_t part_q = split_upload ? (nq + split_parts - 1) / split_parts : nq;
	const uint32_t first_half = part_q;   // queries uploaded in front of the first launch
	if (split_upload) {
		if (int rc = c->ensure_aux(); rc) return rc;
	}
	// Small calls — the Map's single queries and its coalesced batches — move their queries and results through the context's PINNED buffer.
	// A copy between pageable memory and the device goes through the runtime's own staging, one call after the other whatever their
	// streams: T planner threads then queue up in the copies on both sides of a 0.5 ms kernel.  (Large batches keep the direct copies:
	// they are the bandwidth case, and the two-halves upload overlaps them with the searches.)
	const size_t up_bytes = (size_t(nq) * h->dim * qelem + 15) & ~size_t(15);
	const size_t st_corr = up_bytes, st_norm = st_corr + size_t(nq) * 4, st_count = st_norm + size_t(nq) * 4, st_dist = st_count + size_t(nq) * 4,
				 st_row = st_dist + size_t(nq) * k * 4, st_end = st_row + size_t(nq) * k * 4;
	const bool staged = !split_upload && st_end <= (size_t(1) << 20);
	const void* up_queries = queries;
	const float *up_qcorr = qcorr, *up_qnorm = qnorm;
	uint32_t* dl_count = out_count;
	float* dl_dist = out_dist;
	uint32_t* dl_row = out_row;
	// The Map's single queries and its small coalesced batches do not copy at all: the kernel reads the queries from the pinned buffer (once,
	// into LDS or registers) and writes counts and lists there — a call is ONE launch and one wait instead of a launch between four copies
	// (each an enqueue of its own on the path of a 0.5 ms search).  RXGPU_HNSW_ZERO_COPY=0: the copies (A/B).
	const bool zero_copy = staged && to_host && !sq8 && nq <= 64 && knobs.zero_copy != 0;
	char* zc_dev = nullptr;   // the pinned buffer as the device sees it
	if (staged) {
		if (int rc = c->ensure_pinned(st_end); rc) return rc;
		char* hp = static_cast<char*>(c->h_pinned);
		if (zero_copy) {
			void* dv = nullptr;
			RX_HIP(hipHostGetDevicePointer(&dv, hp, 0));
			zc_dev = static_cast<char*>(dv);
		}
		std::memcpy(hp, queries, size_t(nq) * h->dim * qelem);
		up_queries = hp;
		if (sq8) {
			std::memcpy(hp + st_corr, qcorr, size_t(nq) * 4);
			std::memcpy(hp + st_norm, qnorm, size_t(nq) * 4);
			up_qcorr = reinterpret_cast<const float*>(hp + st_corr);
			up_qnorm = reinterpret_cast<const float*>(hp + st_norm);
		}
		dl_count = reinterpret_cast<uint32_t*>(hp + st_count);
		dl_dist = reinterpret_cast<float*>(hp + st_dist);
		dl_row = reinterpret_cast<uint32_t*>(hp + st_row);
	}
	auto done_host = [&]() -> int {   // (behind the last hipStreamSynchronize) what was staged goes to the caller's arrays
		if (staged) {
			std::memcpy(out_count, dl_count, size_t(nq) * 4);
			if (to_host) {
				std::memcpy(out_dist, dl_dist, size_t(nq) * k * 4);
				std::memcpy(out_row, dl_row, size_t(nq) * k * 4);
			}
		}
		return RXGPU_OK;
	};
	void* const d_q = zero_copy ? static_cast<void*>(zc_dev) : c->d_queries.ptr;   // where the kernels read the queries
	if (!zero_copy) RX_HIP(hipMemcpyAsync(c->d_queries.ptr, up_queries, size_t(first_half) * h->dim * qelem, hipMemcpyHostToDevice, c->stream));
	// counts / lists back to the host (nothing to do when the kernels wrote them there), and the wait behind a small launch: polled — the
	// wake-up out of hipStreamSynchronize alone is tens of microseconds
	auto fetch_counts = [&]() -> int {
		if (!zero_copy) RX_HIP(hipMemcpyAsync(dl_count, c->d_out_count.ptr, size_t(nq) * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
		return RXGPU_OK;
	};
	auto fetch_lists = [&]() -> int {
		if (!zero_copy) {
			RX_HIP(hipMemcpyAsync(dl_dist, c->d_out_dist.ptr, size_t(nq) * k * sizeof(float), hipMemcpyDeviceToHost, c->stream));
			RX_HIP(hipMemcpyAsync(dl_row, c->d_out_row.ptr, size_t(nq) * k * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
		}
		return RXGPU_OK;
	};
	auto wait_stream = [&]() -> int {
		if (nq <= 256) {
			const auto t0 = std::chrono::steady_clock::now();
			hipError_t q = hipStreamQuery(c->stream);
			while (q == hipErrorNotReady && std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < 4000.0) q = hipStreamQuery(c->stream);
			if (q != hipErrorNotReady) {
				RX_HIP(q);
				return RXGPU_OK;
			}
		}
		RX_HIP(hipStreamSynchronize(c->stream));
		return RXGPU_OK;
	};
	rxgpu::HnswParams p{};
	if (sq8) {
		char* qb = static_cast<char*>(c->d_queries.ptr);
		RX_HIP(hipMemcpyAsync(qb + o_qcorr, up_qcorr, size_t(nq) * 4, hipMemcpyHostToDevice, c->stream));
		RX_HIP(hipMemcpyAsync(qb + o_qnorm, up_qnorm, size_t(nq) * 4, hipMemcpyHostToDevice, c->stream));
		p.codes = h->d_codes;
		p.corr = h->d_corr;
		p.alpha2 = h->sq8_alpha2;
		p.qcodes = reinterpret_cast<const uint8_t*>(qb);
		p.qcorr = reinterpret_cast<const float*>(qb + o_qcorr);
		p.qnorm = reinterpret_cast<const float*>(qb + o_qnorm);
	}
	p.rows = h->d_rows;
	p.inv_norms = h->d_inv_norms;
	p.links0 = h->d_links0;
	p.upper_off = h->d_upper_off;
	p.upper = h->d_upper;
	p.deleted = h->d_deleted;
	p.n = h->count;
	p.stride = h->stride;
	p.dim = h->dim;
	p.M = h->graph_M;
	p.maxM0 = h->graph_maxM0;
	p.maxlevel = h->graph_maxlevel;
	p.entry = h->graph_entry;
	p.bare = h->graph_deleted == 0;
	p.nq = nq;
	p.k = k;
	p.ef = ef;
	p.visited_words = words;
	p.vis_lds_log2 = vis_lds_log2;
	p.prefetch_links = 1;
	if (knobs.prefetch >= 0) p.prefetch_links = knobs.prefetch ? 1u : 0u;   // A/B hook
	// a handful of searches on the chip: four wavefronts share a search's distance batches (RXGPU_HNSW_TEAM=1: off, RXGPU_HNSW_TEAM_MAX: up to
	// how many searches per launch)
	p.team = knobs.team >= 0 ? uint32_t(knobs.team) : 4u;
	p.team_max = knobs.team_max >= 0 ? uint32_t(knobs.team_max) : 256u;
	p.nbl = knobs.nbl > 0 ? 1u : 0u;
	p.spec = knobs.spec > 0 ? 1u : 0u;   // off by default: measured slower at 1M x 768 (profiles/rd6sp_single.json), see hnsw_search_core.hip.h
	p.out_dist = zero_copy ? reinterpret_cast<float*>(zc_dev + st_dist) : static_cast<float*>(c->d_out_dist.ptr);
	p.out_row = zero_copy ? reinterpret_cast<uint32_t*>(zc_dev + st_row) : static_cast<uint32_t*>(c->d_out_row.ptr);
	p.out_count = zero_copy ? reinterpret_cast<uint32_t*>(zc_dev + st_count) : static_cast<uint32_t*>(c->d_out_count.ptr);
	p.stats = h->d_hnsw_stats;
	p.ef_cap = (ef + 63u) & ~63u;
	auto to_sink = [&]() -> int {   // the finished lists into the exchange's send buffer (sharded HNSW), on this search's stream, drained
		rxgpu::launch_pack_lists(p.out_dist, p.out_row, p.out_count, nq, k, sink->kk, sink->d_dist, sink->d_row, c->stream);
		RX_HIP(hipGetLastError());
		RX_HIP(hipStreamSynchronize(c->stream));
		return done_host();   // (the counts)
	};
	// typical candidate heaps stay within a few x ef.  Measured at 1M x 768, ef = 128: 512 entries overflow for a handful of queries and the
	// global-heap re-run costs more than the extra occupancy brings (1.07 M q/s at 1024 against 0.43 M at 512 and 0.86 M at 768)
	p.lds_cand_cap = ef <= 256 ? 1024u : uint32_t(rxgpu::kHnswCandLds);
	if (knobs.lds_cand_cap >= 0) {   // test hook: force the global-heap re-run
		p.lds_cand_cap = std::min<uint32_t>(uint32_t(rxgpu::kHnswCandLds), uint32_t(std::max(1, knobs.lds_cand_cap)));
	}
	// Graphs without deleted nodes, ef <= 256: both queues as one sorted list in registers (hnsw_search.hip).  A query that meets equal
	// distances there comes back as kHnswTie and takes the heap kernel, whose sift order is the reference's.
	bool use_sorted = ef <= uint32_t(p.bare ? rxgpu::kHnswSortedMaxEf : rxgpu::kHnswSortedMaxEfDel);
	uint32_t sorted_mode = 1;
	// candidate-heap entries a restarted search gets in LDS.  Measured at 1M x 768, ef = 128, 16 384 queries, ~90 restarts (profiles/
	// rd3p_restart_caps.txt, one graph, one box): 384 entries (8 KB per workgroup, 19 per CU) -> one restart overflows and the global-heap
	// launch it needs costs 2.9 ms; 600 (10 KB, 16 per CU) 11.96 ms in all; 780 12.10; 1024 12.25; no in-kernel restart (0: the tie queries
	// come back to this function and get a launch of their own) 9.92 + 2.53 = 12.46 ms.
	uint32_t sorted_restart_cap = 600;
	// Round 4: with helper workgroups beside the batch (below) an overflowing restart is no longer a launch behind the batch, and the area
	// can shrink to what lets a CU hold 20 searches instead of 15 (LDS per workgroup 10.3 -> 7.6 KB): first pass of 16 384 queries at
	// 1M x 768 11.9 -> 10.5 ms (profiles/rd4k_hnsw_1m_restart_caps.txt; without the helpers the one restart that overflows costs 2.5 ms).
	const bool helper_wanted = nq >= 2048 && !big_ef && knobs.helper != 0;
	// ... where a batch lasts long against one heap search: the overflowing searches now run beside the batch, but one that is queued late
	// still sticks out by its own length (2.5 ms at 1M x 768, where the whole batch takes 10: 1.29 M q/s with the copies at 600 entries
	// against 1.16 - 1.22 M at 256 although the first pass alone runs at 1.57 - 1.70 M; at 10M x 768: 495 k -> 586 k q/s,
	// profiles/rd4l_hnsw_*.json).  Same size rule as the hash set.
	if (helper_wanted && ef <= 128 && (16ull << 13) <= words) sorted_restart_cap = 256;
	if (knobs.restart_cand >= 0) sorted_restart_cap = std::min<uint32_t>(uint32_t(rxgpu::kHnswCandLds), uint32_t(knobs.restart_cand));
	if (knobs.sorted >= 0) {   // A/B and test hook: 0 = heaps only, 2 = list shifts through ds_bpermute instead of DPP
		sorted_mode = uint32_t(knobs.sorted);
		use_sorted = use_sorted && sorted_mode != 0;
	}
	std::vector<uint32_t> redo;
	if (big_ef) {
		redo.resize(nq);
		for (uint32_t q = 0; q < nq; ++q) redo[q] = q;
	} else {
		// Helper workgroups beside a large batch (hnsw_helper_kernel, second stream): a search that overflows its LDS heap area is queued and
		// runs with the largest LDS heap while the batch is still going, instead of as a launch of its own behind it.  RXGPU_HNSW_HELPER=0: off.
		constexpr uint32_t kHelperGroups = 64, kHelperCap = 4096;
		// ONE batch at a time has helpers, and only a batch that is one chunk.  A helper polls until ITS batch is over; HIP streams share a
		// few hardware queues, so with two callers at it helper A can sit in front of batch B's kernels while helper B sits in front of
		// batch A's — each waits for a batch that cannot start, until the helpers' wall-clock bail-out (seconds: four threads with 2300-query
		// batches measured 3 s calls, tests/test_gpu_hnsw_visited.py).  With a single set of helpers in flight nothing that spins ever waits
		// for work queued behind another spinner; the callers that come second run their overflowing searches behind their batch, as batches
		// below 2048 queries always do.  (The same inside one call: the kernels of a second chunk would queue up behind its own helpers.)
		static std::atomic<bool> helpers_in_flight{false};
		struct HelperLease {
			bool held = false;
			~HelperLease() {
				if (held) helpers_in_flight.store(false, std::memory_order_release);
			}
		} helper_lease;
		if (helper_wanted && uint64_t(nq) <= vis_slots) {
			bool expected = false;
			helper_lease.held = helpers_in_flight.compare_exchange_strong(expected, true, std::memory_order_acq_rel);
		}
		const bool use_helper = helper_lease.held;
		uint32_t* hq_words = nullptr;   // [0] entries appended, [1] stop, [2] searches of the batch that have ended, [16 ..] ids
		uint32_t helper_n = 0;
		if (use_helper) {
			const uint64_t words4 = (words + 3) & ~uint64_t(3);
			const size_t hq_bytes = (size_t(kHelperCap) + 16) * 4;
			if (int rc = c->ensure_aux(); rc) return rc;
			if (int rc = c->d_helper.ensure(hq_bytes); rc) return rc;
			if (int rc = c->d_helper_bits.ensure(size_t(kHelperGroups) * words4 * 4); rc) return rc;
			hq_words = static_cast<uint32_t*>(c->d_helper.ptr);
			RX_HIP(hipMemsetAsync(hq_words, 0, hq_bytes, c->aux_stream));
			RX_HIP(hipEventRecord(c->main_done, c->aux_stream));      // the queue is empty before the first search of the batch can append to it
			RX_HIP(hipStreamWaitEvent(c->stream, c->main_done, 0));
		}
		// The helpers spin until the batch says stop, so they are enqueued BEHIND the batch's launches: HIP streams share a few
		// hardware queues, and a helper that reached a queue in front of the batch kernel it waits for would hold that queue until its
		// wall-clock bail-out.  Launched last, the worst case is the serial one — helpers behind the batch on one queue find the stop flag and
		// only drain what was queued; on separate queues they run beside the batch as intended.
		auto launch_helpers = [&]() -> int {
			const uint64_t words4 = (words + 3) & ~uint64_t(3);
			rxgpu::HnswParams ph = p;
			ph.queries = static_cast<const float*>(d_q);
			ph.visited = static_cast<uint32_t*>(c->d_helper_bits.ptr);
			ph.visited_words = words4;
			ph.vis_hash_log2 = 0;
			ph.vis_lds_log2 = 0;
			ph.lds_cand_cap = uint32_t(rxgpu::kHnswCandLds);
			ph.sorted = 0;
			rxgpu::HnswHelper hq{hq_words, hq_words + 16, hq_words + 1, hq_words + 2, nq, kHelperCap, 100000000ull};   // (last resort: gives up after 1 s)
			rxgpu::launch_hnsw_helper(h->metric, ph, hq, kHelperGroups, c->aux_stream);
			RX_HIP(hipGetLastError());
			return RXGPU_OK;
		};
		bool helpers_launched = false;
		for (uint32_t q0 = 0; q0 < nq; q0 += uint32_t(vis_slots)) {
			const uint32_t cq = uint32_t(std::min<uint64_t>(vis_slots, nq - q0));
			if (int rc = c->d_visited.ensure(size_t(cq) * vis_words * 4); rc) return rc;
			if (!vis_hash_log2) {
				const size_t zero_bytes = size_t(cq) * words * 4;
				if (q0 == 0 && first_zeroed) {
					RX_HIP(hipStreamWaitEvent(c->stream, c->aux_done, 0));
				} else {   // (a later chunk of the same call reuses the buffer behind the chunk before it)
					RX_HIP(hipMemsetAsync(c->d_visited.ptr, 0, zero_bytes, c->stream));
				}
			}
			// searches [qa, qa + cnt) of the batch on stream st (slot = index inside this launch's visited block)
			auto launch_part = [&](uint32_t qa, uint32_t cnt, uint32_t slot0, hipStream_t st) {
				rxgpu::HnswParams pc = p;
				pc.vis_hash_log2 = vis_hash_log2;
				pc.visited_words = vis_words;
				pc.queries = reinterpret_cast<const float*>(static_cast<const char*>(d_q) + size_t(qa) * h->dim * qelem);
				if (sq8) {
					pc.qcodes = p.qcodes + size_t(qa) * h->dim;
					pc.qcorr = p.qcorr + qa;
					pc.qnorm = p.qnorm + qa;
				}
				pc.visited = static_cast<uint32_t*>(c->d_visited.ptr) + size_t(slot0) * vis_words;
				pc.out_dist = p.out_dist + size_t(qa) * k;
				pc.out_row = p.out_row + size_t(qa) * k;
				pc.out_count = p.out_count + qa;
				if (use_helper) {
					pc.helper_n = hq_words;
					pc.helper_ids = hq_words + 16;
					pc.helper_cap = kHelperCap;
					pc.q_base = qa;
				}
				if (use_sorted) {   // the list lives in registers; LDS holds only the heap area of a search that starts over (equal keys that matter)
					pc.sorted = sorted_mode;
					pc.lds_cand_cap = sorted_restart_cap;
					if (sorted_restart_cap == 0) pc.ef_cap = 0;
				}
				rxgpu::launch_hnsw_search(h->metric, pc, cnt, false, st);
			};
			ProfileScope ps(h, "hnsw", c->stream);   // (with two halves: until the main stream has waited for the second one)
			if (split_upload) {
				hipStream_t sb = c->aux2_stream;
				// whatever the first half waits for — zeroed bitsets, the empty overflow queue, the SQ8 query terms — the second half waits for too
				RX_HIP(hipEventRecord(c->split_done, c->stream));
				RX_HIP(hipStreamWaitEvent(sb, c->split_done, 0));
				launch_part(0, first_half, 0, c->stream);
				for (uint32_t qa = first_half, part = 1; qa < nq; qa += part_q, ++part) {
					const uint32_t cnt = std::min(part_q, nq - qa);
					hipStream_t st = (part & 1u) ? sb : c->stream;
					const size_t off = size_t(qa) * h->dim * qelem;
					RX_HIP(hipMemcpyAsync(static_cast<char*>(c->d_queries.ptr) + off, static_cast<const char*>(queries) + off, size_t(cnt) * h->dim * qelem, hipMemcpyHostToDevice, st));
					launch_part(qa, cnt, qa, st);
				}
				RX_HIP(hipEventRecord(c->split_done, sb));
				RX_HIP(hipStreamWaitEvent(c->stream, c->split_done, 0));
			} else {
				launch_part(q0, cq, 0, c->stream);
			}
			if (use_helper && !helpers_launched) {   // behind the first chunk's launches (all of them, for a batch that fits one chunk)
				if (int rc = launch_helpers(); rc) return rc;
				helpers_launched = true;
			}
		}
		RX_HIP(hipGetLastError());
		if (use_helper) {   // the batch is over: tell the helpers, take their results with the batch's
			RX_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(hq_words + 1), 1, 1, c->stream));
			RX_HIP(hipEventRecord(c->aux_done, c->aux_stream));
			RX_HIP(hipStreamWaitEvent(c->stream, c->aux_done, 0));
			RX_HIP(hipMemcpyAsync(&helper_n, hq_words, sizeof(helper_n), hipMemcpyDeviceToHost, c->stream));
		}
		// counts and results travel together: a batch without re-runs (the common case for a handful of queries) is done after ONE wait
		if (int rc = fetch_counts(); rc) return rc;
		if (to_host) {
			if (int rc = fetch_lists(); rc) return rc;
		}
		if (int rc = wait_stream(); rc) return rc;
		const uint32_t helper_queued = std::min<uint32_t>(helper_n, kHelperCap);
		h->hnsw_lds_reruns += helper_queued;
		std::vector<uint32_t> ties;
		bool clean = true;
		for (uint32_t q = 0; q < nq; ++q) {
			if (dl_count[q] == rxgpu::kHnswTie) ties.push_back(q);
			clean = clean && dl_count[q] != rxgpu::kHnswTie && dl_count[q] != rxgpu::kHnswOverflow;
		}
		if (clean) return to_host ? done_host() : to_sink();
		if (!ties.empty()) {   // equal keys met in the sorted list: the same queries through the reference's heaps (candidate heap in LDS)
			h->hnsw_tie_reruns += ties.size();
			if (int rc = c->d_redo.ensure(ties.size() * sizeof(uint32_t)); rc) return rc;
			RX_HIP(hipMemcpyAsync(c->d_redo.ptr, ties.data(), ties.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
			for (size_t r0 = 0; r0 < ties.size(); r0 += vis_slots) {
				const uint32_t cq = uint32_t(std::min<uint64_t>(vis_slots, ties.size() - r0));
				if (int rc = c->d_visited.ensure(size_t(cq) * vis_words * 4); rc) return rc;
				if (!vis_hash_log2) RX_HIP(hipMemsetAsync(c->d_visited.ptr, 0, size_t(cq) * words * 4, c->stream));
				rxgpu::HnswParams pc = p;
				pc.vis_hash_log2 = vis_hash_log2;
				pc.visited_words = vis_words;
				pc.queries = static_cast<const float*>(d_q);
				pc.visited = static_cast<uint32_t*>(c->d_visited.ptr);
				pc.only = static_cast<const uint32_t*>(c->d_redo.ptr) + r0;
				ProfileScope ps(h, "hnsw_ties", c->stream);
				rxgpu::launch_hnsw_search(h->metric, pc, cq, false, c->stream);
			}
			RX_HIP(hipGetLastError());
			if (int rc = fetch_counts(); rc) return rc;
			RX_HIP(hipStreamSynchronize(c->stream));
		}
		// Queries whose candidate heap outgrew its LDS area — 600 entries for a search that started over inside the sorted-list kernel, 1024 for
		// the heap kernel's first pass: once more on the heap kernel with the largest LDS heap there is and a bitset (a search that filled
		// its hash set lands here too), before the global-heap tiers.  A search with its heap in global scratch takes 5 - 7 ms at 10M x 768
		// (profiles/rd4i_hnsw_10m_*.json: ONE such query was a fifth of a 16 384-query batch), one in LDS about 1 ms.
		std::vector<uint32_t> over;
		for (uint32_t q = 0; q < nq; ++q) {
			if (dl_count[q] == rxgpu::kHnswOverflow) over.push_back(q);
		}
		const uint32_t first_cap = use_sorted ? sorted_restart_cap : p.lds_cand_cap;
		if (!over.empty() && first_cap < uint32_t(rxgpu::kHnswCandLds) && knobs.lds_cand_cap < 0) {   // (the hook forces the global tiers)
			if (int rc = c->d_redo.ensure(over.size() * sizeof(uint32_t)); rc) return rc;
			RX_HIP(hipMemcpyAsync(c->d_redo.ptr, over.data(), over.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
			for (size_t r0 = 0; r0 < over.size(); r0 += max_slots) {
				const uint32_t cq = uint32_t(std::min<uint64_t>(max_slots, over.size() - r0));
				if (int rc = c->d_visited.ensure(size_t(cq) * words * 4); rc) return rc;
				RX_HIP(hipMemsetAsync(c->d_visited.ptr, 0, size_t(cq) * words * 4, c->stream));
				rxgpu::HnswParams pc = p;
				pc.lds_cand_cap = uint32_t(rxgpu::kHnswCandLds);
				pc.vis_lds_log2 = 0;   // (a search that filled its hash set is among these)
				pc.queries = static_cast<const float*>(d_q);
				pc.visited = static_cast<uint32_t*>(c->d_visited.ptr);
				pc.only = static_cast<const uint32_t*>(c->d_redo.ptr) + r0;
				ProfileScope ps(h, "hnsw_redo", c->stream);
				rxgpu::launch_hnsw_search(h->metric, pc, cq, false, c->stream);
			}
			RX_HIP(hipGetLastError());
			if (int rc = fetch_counts(); rc) return rc;
			RX_HIP(hipStreamSynchronize(c->stream));
			// searches the helpers had queued but not finished are in `over` again: counted once
			h->hnsw_lds_reruns += over.size() > helper_queued ? over.size() - helper_queued : 0;
		}
		// ... and what still does not fit: re-run with the heap in global scratch (bounded by one entry per node)
		for (const uint32_t q : over) {
			if (dl_count[q] == rxgpu::kHnswOverflow) redo.push_back(q);
		}
	}
	// Re-runs with the candidate heap in global scratch, in two tiers: 64 K entries first (0.5 MB per search: hundreds of re-runs share one
	// launch), one entry per node — the bound that cannot overflow — only for what outgrows that.  (With the full bound from the start a
	// 10M-node index allows 13 searches per launch: 39 overflowing queries out of 16 384 cost a quarter of the whole batch.)
	uint64_t tier_cap[2] = {std::min<uint64_t>(h->count + 1, 65536), h->count + 1};
	if (knobs.gcand_cap >= 0) tier_cap[0] = std::min<uint64_t>(h->count + 1, uint64_t(std::max(1, knobs.gcand_cap)));   // test hook
	for (int tier = 0; tier < 2 && !redo.empty(); ++tier) {
		if (tier == 1 && tier_cap[1] == tier_cap[0]) break;
		const uint64_t gcap = tier_cap[tier];
		const uint64_t redo_slots = std::max<uint64_t>(1, std::min<uint64_t>(max_slots, (1ull << 30) / (gcap * 8)));
		if (int rc = c->d_redo.ensure(redo.size() * sizeof(uint32_t)); rc) return rc;
		RX_HIP(hipMemcpyAsync(c->d_redo.ptr, redo.data(), redo.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
		for (size_t r0 = 0; r0 < redo.size(); r0 += redo_slots) {
			const uint32_t cq = uint32_t(std::min<uint64_t>(redo_slots, redo.size() - r0));
			if (int rc = c->d_visited.ensure(size_t(cq) * words * 4); rc) return rc;
			if (int rc = c->d_gcand_d.ensure(size_t(cq) * gcap * sizeof(uint2)); rc) return rc;
			RX_HIP(hipMemsetAsync(c->d_visited.ptr, 0, size_t(cq) * words * 4, c->stream));
			rxgpu::HnswParams pc = p;
			pc.queries = static_cast<const float*>(d_q);
			pc.visited = static_cast<uint32_t*>(c->d_visited.ptr);
			pc.only = static_cast<const uint32_t*>(c->d_redo.ptr) + r0;
			pc.gcand = static_cast<uint2*>(c->d_gcand_d.ptr);
			pc.gcand_cap = gcap;
			ProfileScope ps(h, "hnsw_redo", c->stream);
			rxgpu::launch_hnsw_search(h->metric, pc, cq, true, c->stream);
		}
		RX_HIP(hipGetLastError());
		if (int rc = fetch_counts(); rc) return rc;
		RX_HIP(hipStreamSynchronize(c->stream));
		std::vector<uint32_t> again;
		for (const uint32_t q : redo) {
			if (dl_count[q] == rxgpu::kHnswOverflow) again.push_back(q);
		}
		redo.swap(again);
	}
	RX_CHECK(redo.empty(), RXGPU_ERR_DEVICE, "rxgpu_hnsw_search_knn: a candidate heap of one entry per node overflowed");
	if (!to_host) return to_sink();
	if (int rc = fetch_lists(); rc) return rc;
	RX_HIP(hipStreamSynchronize(c->stream));
	return done_host();
}

// ---------------------------------------------------------------------------------------------- SearchRange, expansion on the device
// hnswalg.h:2015-2070: ef-search (the kernels above), then the closure over level-0 links while dist < radius — hnsw_range_kernel, one
// launch whatever the depth of the expansion.  cap too small: RXGPU_ERR_OVERFLOW with *out_total = hits counted so far (a lower bound: the
// expansion stops growing where it cannot store) — the caller retries with more room.
static int hnsw_range_impl(rxgpu_index* h, const void* query, const float* qcorr, const float* qnorm, float radius, uint32_t ef, float* out_dist,
						   uint32_t* out_row, uint64_t cap, uint64_t* out_total) {
	const bool sq8 = qcorr != nullptr;
	RX_CHECK(h && query && out_total, RXGPU_ERR_PARAMS, "rxgpu_hnsw_search_range: null argument");
	*out_total = 0;
	RX_CHECK(cap == 0 || (out_dist && out_row), RXGPU_ERR_PARAMS, "rxgpu_hnsw_search_range: null argument");
	if (h->count == 0) return RXGPU_OK;
	const uint32_t ef_eff = ef ? ef : 1;
	const uint32_t kk = uint32_t(std::min<uint64_t>(ef_eff, h->count));
	std::vector<float> sd(kk);
	std::vector<uint32_t> sr(kk);
	uint32_t sc = 0;
	if (int rc = hnsw_search_impl(h, query, qcorr, qnorm, 1, kk, ef_eff, sd.data(), sr.data(), &sc); rc) return rc;
	DeviceGuard dg(h->device);
	rxgpu_search_ctx* c = acquire_ctx(h);
	if (!c) return RXGPU_ERR_DEVICE;
	struct Rel {
		rxgpu_index* h;
		rxgpu_search_ctx* c;
		~Rel() { release_ctx(h, c); }
	} rel{h, c};
	const uint64_t words = (h->count + 31) / 32;
	const uint64_t room = std::max<uint64_t>(cap, 1);
	const size_t qelem = sq8 ? 1 : 4;
	size_t carve = 0;
	auto take = [&carve](size_t bytes) {
		const size_t at = carve;
		carve = (carve + bytes + 255) & ~size_t(255);
		return at;
	};
	const size_t o_q = take(size_t(h->dim) * qelem + 16), o_qc = take(8), o_sd = take(size_t(kk) * 4), o_sr = take(size_t(kk) * 4),
				 o_front = take(size_t(2) * room * 4), o_total = take(8);
	if (int rc = c->d_misc.ensure(carve); rc) return rc;
	if (int rc = c->d_visited.ensure(words * 4); rc) return rc;
	if (int rc = c->d_out_dist.ensure(room * 4); rc) return rc;
	if (int rc = c->d_out_row.ensure(room * 4); rc) return rc;
	char* mb = static_cast<char*>(c->d_misc.ptr);
	RX_HIP(hipMemcpyAsync(mb + o_q, query, size_t(h->dim) * qelem, hipMemcpyHostToDevice, c->stream));
	if (sq8) {
		const float qc[2] = {*qcorr, *qnorm};
		RX_HIP(hipMemcpyAsync(mb + o_qc, qc, 8, hipMemcpyHostToDevice, c->stream));
	}
	if (sc) {
		RX_HIP(hipMemcpyAsync(mb + o_sd, sd.data(), size_t(sc) * 4, hipMemcpyHostToDevice, c->stream));
		RX_HIP(hipMemcpyAsync(mb + o_sr, sr.data(), size_t(sc) * 4, hipMemcpyHostToDevice, c->stream));
	}
	RX_HIP(hipMemsetAsync(mb + o_total, 0, 8, c->stream));
	RX_HIP(hipMemsetAsync(c->d_visited.ptr, 0, words * 4, c->stream));
	rxgpu::HnswParams p{};
	p.rows = h->d_rows;
	p.inv_norms = h->d_inv_norms;
	p.links0 = h->d_links0;
	p.deleted = h->d_deleted;
	p.n = h->count;
	p.stride = h->stride;
	p.dim = h->dim;
	p.M = h->graph_M;
	p.maxM0 = h->graph_maxM0;
	p.bare = h->graph_deleted == 0;
	p.queries = reinterpret_cast<const float*>(mb + o_q);
	if (sq8) {
		p.codes = h->d_codes;
		p.corr = h->d_corr;
		p.alpha2 = h->sq8_alpha2;
		p.qcodes = reinterpret_cast<const uint8_t*>(mb + o_q);
		p.qcorr = reinterpret_cast<const float*>(mb + o_qc);
		p.qnorm = reinterpret_cast<const float*>(mb + o_qc) + 1;
	}
	rxgpu::HnswRange r{};
	r.seed_dist = reinterpret_cast<const float*>(mb + o_sd);
	r.seed_row = reinterpret_cast<const uint32_t*>(mb + o_sr);
	r.seed_n = sc;
	r.radius = radius;
	r.visited = static_cast<uint32_t*>(c->d_visited.ptr);
	r.frontier = reinterpret_cast<uint32_t*>(mb + o_front);
	r.out_dist = static_cast<float*>(c->d_out_dist.ptr);
	r.out_row = static_cast<uint32_t*>(c->d_out_row.ptr);
	r.total = reinterpret_cast<unsigned long long*>(mb + o_total);
	r.cap = cap;
	{
		ProfileScope ps(h, "hnsw_range", c->stream);
		rxgpu::launch_hnsw_range(h->metric, p, r, c->stream);
	}
	RX_HIP(hipGetLastError());
	unsigned long long total = 0;
	RX_HIP(hipMemcpyAsync(&total, mb + o_total, 8, hipMemcpyDeviceToHost, c->stream));
	RX_HIP(hipStreamSynchronize(c->stream));
	*out_total = total;
	const uint64_t have = std::min<uint64_t>(total, cap);
	if (have) {
		RX_HIP(hipMemcpyAsync(out_dist, c->d_out_dist.ptr, have * 4, hipMemcpyDeviceToHost, c->stream));
		RX_HIP(hipMemcpyAsync(out_row, c->d_out_row.ptr, have * 4, hipMemcpyDeviceToHost, c->stream));
		RX_HIP(hipStreamSynchronize(c->stream));
	}
	if (total > cap) {
		set_error("rxgpu_hnsw_search_range: more hits than the output buffer holds");
		return RXGPU_ERR_OVERFLOW;
	}
	return RXGPU_OK;
}

int rxgpu_hnsw_search_range(rxgpu_index* h, const float* query, float radius, uint32_t ef, float* out_dist, uint32_t* out_row, uint64_t cap,
							uint64_t* out_total) {
	RX_CHECK(h, RXGPU_ERR_PARAMS, "null index");
	if (h->shard_set) return rxgpu::sharded_hnsw_search_range(h, query, radius, ef, out_dist, out_row, cap, out_total);
	RX_CHECK(h->count == 0 || (h->graph_attached && h->graph_n == h->count), RXGPU_ERR_LOGIC, "rxgpu_hnsw_search_range: graph is not attached / out of date");
	return hnsw_range_impl(h, query, nullptr, nullptr, radius, ef, out_dist, out_row, cap, out_total);
}

int rxgpu_hnsw_search_range_sq8(rxgpu_index* h, const uint8_t* query_codes, float query_corr, float query_norm_coef, float radius, uint32_t ef,
								float* out_dist, uint32_t* out_row, uint64_t cap, uint64_t* out_total) {
	RX_CHECK(h, RXGPU_ERR_PARAMS, "null index");
	RX_CHECK(!h->shard_set, RXGPU_ERR_LOGIC, "rxgpu_hnsw_search_range_sq8: not available on a sharded index");
	RX_CHECK(h->count == 0 || (h->graph_attached && h->graph_n == h->count), RXGPU_ERR_LOGIC, "rxgpu_hnsw_search_range_sq8: graph is not attached / out of date");
	RX_CHECK(h->count == 0 || (h->d_codes && h->sq8_n == h->count), RXGPU_ERR_LOGIC, "rxgpu_hnsw_search_range_sq8: SQ8 codes are not attached / out of date");
	return hnsw_range_impl(h, query_codes, &query_corr, &query_norm_coef, radius, ef, out_dist, out_row, cap, out_total);
}

// ---------------------------------------------------------------------------------------------- streaming KNN sessions
struct rxgpu_hnsw_stream {
	rxgpu_index* owner = nullptr;
	uint64_t graph_n = 0;
	uint32_t ef = 0;
	hipStream_t stream = nullptr;
	rxgpu_devbuf d_query, d_visited, d_cand, d_top, d_ext, d_state, d_out;
	uint32_t out_cap = 0;
	rxgpu::HnswStreamState host_state{};
	bool empty_graph = false;
	bool sq8 = false;          // the session runs over the SQ8 codes: d_query holds the query's codes
	float qcorr = 0.f, qnorm = 1.f;
};

static void fill_hnsw_params(const rxgpu_index* h, rxgpu::HnswParams& p) {
	p.rows = h->d_rows;
	p.inv_norms = h->d_inv_norms;
	p.links0 = h->d_links0;
	p.upper_off = h->d_upper_off;
	p.upper = h->d_upper;
	p.deleted = h->d_deleted;
	p.n = h->count;
	p.stride = h->stride;
	p.dim = h->dim;
	p.M = h->graph_M;
	p.maxM0 = h->graph_maxM0;
	p.maxlevel = h->graph_maxlevel;
	p.entry = h->graph_entry;
	p.bare = h->graph_deleted == 0;
}

static void fill_sq8_params(const rxgpu_index* h, const rxgpu_hnsw_stream* s, rxgpu::HnswParams& p) {
	if (!s->sq8) return;
	p.codes = h->d_codes;
	p.corr = h->d_corr;
	p.alpha2 = h->sq8_alpha2;
}

static void fill_stream(const rxgpu_hnsw_stream* s, rxgpu::HnswStream& d) {
	const uint64_t cap = s->graph_n;
	d.query = static_cast<const float*>(s->d_query.ptr);
	d.qcodes = s->sq8 ? static_cast<const uint8_t*>(s->d_query.ptr) : nullptr;
	d.qcorr = s->qcorr;
	d.qnorm = s->qnorm;
	d.visited = static_cast<uint32_t*>(s->d_visited.ptr);
	d.cand_d = static_cast<float*>(s->d_cand.ptr);
	d.cand_i = reinterpret_cast<uint32_t*>(d.cand_d + cap);
	d.top_d = static_cast<float*>(s->d_top.ptr);
	d.top_i = reinterpret_cast<uint32_t*>(d.top_d + 2 * cap);
	d.ext_d = static_cast<float*>(s->d_ext.ptr);
	d.ext_i = reinterpret_cast<uint32_t*>(d.ext_d + 2 * cap);
	d.cap = uint32_t(cap);
	d.ef = s->ef;
	d.state = static_cast<rxgpu::HnswStreamState*>(s->d_state.ptr);
	d.out_dist = static_cast<float*>(s->d_out.ptr);
	d.out_row = reinterpret_cast<uint32_t*>(d.out_dist + s->out_cap);
}

void rxgpu_hnsw_stream_end(rxgpu_hnsw_stream* s) {
	if (!s) return;
	DeviceGuard dg(s->owner->device);
	if (s->stream) {
		(void)hipStreamSynchronize(s->stream);
		(void)hipStreamDestroy(s->stream);
	}
	for (rxgpu_devbuf* b : {&s->d_query, &s->d_visited, &s->d_cand, &s->d_top, &s->d_ext, &s->d_state, &s->d_out}) b->release();
	delete s;
}

static int hnsw_stream_begin_impl(rxgpu_index* h, const void* query, bool sq8, float qcorr, float qnorm, uint32_t ef, rxgpu_hnsw_stream** out) {
	RX_CHECK(h && query && out, RXGPU_ERR_PARAMS, "rxgpu_hnsw_stream_begin: null argument");
	*out = nullptr;
	RX_CHECK(!h->shard_set, RXGPU_ERR_LOGIC, "rxgpu_hnsw_stream_begin: streaming sessions are per graph — not available on a sharded index");
	RX_CHECK(!sq8 || h->count == 0 || (h->d_codes && h->sq8_n == h->count), RXGPU_ERR_LOGIC,
			 "rxgpu_hnsw_stream_begin_sq8: SQ8 codes are not attached / out of date");
	DeviceGuard dg(h->device);
	auto* s = new rxgpu_hnsw_stream();
	s->owner = h;
	s->ef = ef ? ef : 100;   // kDefaultStreamingEf (hnswalg.h:1867)
	s->sq8 = sq8;
	s->qcorr = qcorr;
	s->qnorm = qnorm;
	s->graph_n = h->count;
	if (h->count == 0) {     // hnswalg.h:1880-1882: an empty graph yields a session that is exhausted at once
		s->empty_graph = true;
		*out = s;
		return RXGPU_OK;
	}
	struct Guard {
		rxgpu_hnsw_stream* s;
		~Guard() {
			if (s) rxgpu_hnsw_stream_end(s);
		}
	} guard{s};
	RX_CHECK(h->graph_attached && h->graph_n == h->count, RXGPU_ERR_LOGIC, "rxgpu_hnsw_stream_begin: graph is not attached / out of date");
	RX_HIP(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
	const uint64_t cap = h->count, words = (h->count + 31) / 32;
	const size_t qbytes = size_t(h->dim) * (sq8 ? 1 : 4);
	if (int rc = s->d_query.ensure(size_t(h->dim) * 4); rc) return rc;
	if (int rc = s->d_visited.ensure(words * 4); rc) return rc;
	if (int rc = s->d_cand.ensure(cap * 8); rc) return rc;
	if (int rc = s->d_top.ensure(cap * 16); rc) return rc;
	if (int rc = s->d_ext.ensure(cap * 16); rc) return rc;
	if (int rc = s->d_state.ensure(sizeof(rxgpu::HnswStreamState)); rc) return rc;
	RX_HIP(hipMemcpyAsync(s->d_query.ptr, query, qbytes, hipMemcpyHostToDevice, s->stream));
	RX_HIP(hipMemsetAsync(s->d_visited.ptr, 0, words * 4, s->stream));
	rxgpu::HnswParams p{};
	fill_hnsw_params(h, p);
	fill_sq8_params(h, s, p);
	rxgpu::HnswStream d{};
	fill_stream(s, d);
	rxgpu::launch_hnsw_stream(h->metric, p, d, 0, rxgpu::kStreamBegin, false, s->stream);
	RX_HIP(hipGetLastError());
	RX_HIP(hipMemcpyAsync(&s->host_state, s->d_state.ptr, sizeof(s->host_state), hipMemcpyDeviceToHost, s->stream));
	RX_HIP(hipStreamSynchronize(s->stream));
	guard.s = nullptr;
	*out = s;
	return RXGPU_OK;
}

int rxgpu_hnsw_stream_begin(rxgpu_index* h, const float* query, uint32_t ef, rxgpu_hnsw_stream** out) {
	return hnsw_stream_begin_impl(h, query, false, 0.f, 1.f, ef, out);
}

// The same session over a quantised graph (HierarchicalNSWImpl<uint8_t>): the query as prepareData leaves it (codes + corrective offset),
// every distance scaled by its normCoef — BeginStreamingSearch / ContinueStreamingSearch (hnswalg.h:1865-1975) instantiated for uint8_t.
int rxgpu_hnsw_stream_begin_sq8(rxgpu_index* h, const uint8_t* query_codes, float query_corr, float query_norm_coef, uint32_t ef, rxgpu_hnsw_stream** out) {
	return hnsw_stream_begin_impl(h, query_codes, true, query_corr, query_norm_coef, ef, out);
}

int rxgpu_hnsw_stream_continue(rxgpu_hnsw_stream* s, uint32_t batch, float* out_dist, uint32_t* out_row, uint32_t* out_count, int32_t* exhausted) {
	RX_CHECK(s && out_count && exhausted, RXGPU_ERR_PARAMS, "rxgpu_hnsw_stream_continue: null argument");
	*out_count = 0;
	*exhausted = 0;
	if (batch == 0) return RXGPU_OK;   // hnswalg.h:1956-1958
	if (s->empty_graph) {
		*exhausted = 1;
		return RXGPU_OK;
	}
	RX_CHECK(out_dist && out_row, RXGPU_ERR_PARAMS, "rxgpu_hnsw_stream_continue: null argument");
	rxgpu_index* h = s->owner;
	// the whole session must run under the caller's read lock (hnsw_interface.h:99): a mutated graph invalidates it
	RX_CHECK(h->graph_attached && h->graph_n == s->graph_n && h->count == s->graph_n, RXGPU_ERR_LOGIC,
			 "rxgpu_hnsw_stream_continue: the graph changed under the session");
	DeviceGuard dg(h->device);
	const uint32_t out_need = uint32_t(std::min<uint64_t>(batch, s->graph_n));
	if (out_need > s->out_cap) {
		if (int rc = s->d_out.ensure(size_t(out_need) * 8); rc) return rc;
		s->out_cap = out_need;
	}
	rxgpu::HnswParams p{};
	fill_hnsw_params(h, p);
	fill_sq8_params(h, s, p);
	rxgpu::HnswStream d{};
	fill_stream(s, d);
	const rxgpu::HnswStreamState& hs = s->host_state;
	const uint32_t ef_eff = std::max(s->ef, batch);
	bool lds = ef_eff <= uint32_t(rxgpu::kStreamLdsTop) && uint32_t(hs.top_n) <= uint32_t(rxgpu::kStreamLdsTop) &&
			   uint32_t(hs.ext_n) <= uint32_t(rxgpu::kStreamLdsExt) && uint32_t(hs.cand_n) + h->graph_maxM0 <= uint32_t(rxgpu::kStreamLdsCand);
	if (getenv("RXGPU_HNSW_STREAM_GLOBAL")) lds = false;   // test hook: run every call with the heaps in HBM
	int mode = rxgpu::kStreamContinue;
	for (int attempt = 0; attempt < 2; ++attempt) {
		ProfileScope ps(h, lds ? "hnsw_stream" : "hnsw_stream_global", s->stream);
		rxgpu::launch_hnsw_stream(h->metric, p, d, batch, mode, lds, s->stream);
		RX_HIP(hipGetLastError());
		RX_HIP(hipMemcpyAsync(&s->host_state, s->d_state.ptr, sizeof(s->host_state), hipMemcpyDeviceToHost, s->stream));
		RX_HIP(hipStreamSynchronize(s->stream));
		if (s->host_state.status != rxgpu::kStreamNeedGlobal) break;
		RX_CHECK(lds, RXGPU_ERR_DEVICE, "rxgpu_hnsw_stream_continue: global-heap pass asked for more room");
		lds = false;                       // outgrew LDS at a step boundary: same call, heaps in HBM, no second mergeExtras
		mode = rxgpu::kStreamResume;
	}
	RX_CHECK(s->host_state.status == rxgpu::kStreamOk, RXGPU_ERR_DEVICE, "rxgpu_hnsw_stream_continue: device-side session error");
	const uint32_t n = s->host_state.out_count;
	if (n) {
		RX_HIP(hipMemcpyAsync(out_dist, d.out_dist, size_t(n) * 4, hipMemcpyDeviceToHost, s->stream));
		RX_HIP(hipMemcpyAsync(out_row, d.out_row, size_t(n) * 4, hipMemcpyDeviceToHost, s->stream));
		RX_HIP(hipStreamSynchronize(s->stream));
	}
	*out_count = n;
	*exhausted = s->host_state.exhausted ? 1 : 0;
	return RXGPU_OK;
}

int rxgpu_hnsw_read_stats(rxgpu_index* h, uint64_t* distance_evals, uint64_t* hops) {
	RX_CHECK(h && distance_evals && hops, RXGPU_ERR_PARAMS, "rxgpu_hnsw_read_stats: null argument");
	*distance_evals = 0;
	*hops = 0;
	if (h->shard_set) {   // the sum over the shards' graphs
		for (uint32_t s = 0; s < rxgpu_index_shard_count(h); ++s) {
			uint64_t e = 0, hp = 0;
			if (int rc = rxgpu_hnsw_read_stats(rxgpu_index_shard(h, s), &e, &hp); rc) return rc;
			*distance_evals += e;
			*hops += hp;
		}
		return RXGPU_OK;
	}
	if (!h->d_hnsw_stats) return RXGPU_OK;
	DeviceGuard dg(h->device);
	unsigned long long v[4] = {0, 0, 0, 0};   // evals, hops, in-kernel restarts, distance trips of the speculative team searches
	RX_HIP(rxgpu::device_wait_all(h->device));
	RX_HIP(hipMemcpy(v, h->d_hnsw_stats, sizeof(v), hipMemcpyDeviceToHost));
	RX_HIP(hipMemset(h->d_hnsw_stats, 0, sizeof(v)));
	*distance_evals = v[0];
	*hops = v[1];
	if (std::getenv("RXGPU_HNSW_TRIPS")) std::fprintf(stderr, "[rxgpu hnsw] hops %llu evals %llu restarts %llu speculative-search distance trips %llu\n", v[1], v[0], v[2], v[3]);
	if (std::getenv("RXGPU_HNSW_PHASES")) {   // a library built with -DRXGPU_HNSW_PHASES: shader cycles of the sorted-list search by phase
		unsigned long long ph[4] = {0, 0, 0, 0};
		RX_HIP(hipMemcpy(ph, h->d_hnsw_stats + 4, sizeof(ph), hipMemcpyDeviceToHost));
		RX_HIP(hipMemset(h->d_hnsw_stats + 4, 0, sizeof(ph)));
		std::fprintf(stderr, "[rxgpu hnsw phases] hops %llu evals %llu | cycles: pop+links+visited %llu  distances %llu  inserts %llu  layer0 total %llu\n", v[1], v[0], ph[0],
					 ph[1], ph[2], ph[3]);
	}
	return RXGPU_OK;
}

int rxgpu_hnsw_read_stats4(rxgpu_index* h, uint64_t* out4) {
	RX_CHECK(h && out4, RXGPU_ERR_PARAMS, "rxgpu_hnsw_read_stats4: null argument");
	out4[0] = out4[1] = out4[2] = out4[3] = 0;
	if (h->shard_set || !h->d_hnsw_stats) return RXGPU_OK;
	DeviceGuard dg(h->device);
	unsigned long long v[4] = {0, 0, 0, 0};
	RX_HIP(rxgpu::device_wait_all(h->device));
	RX_HIP(hipMemcpy(v, h->d_hnsw_stats, sizeof(v), hipMemcpyDeviceToHost));
	RX_HIP(hipMemset(h->d_hnsw_stats, 0, sizeof(v)));
	for (int i = 0; i < 4; ++i) out4[i] = v[i];
	return RXGPU_OK;
}

int rxgpu_hnsw_read_lds_reruns(rxgpu_index* h, uint64_t* reruns) {
	RX_CHECK(h && reruns, RXGPU_ERR_PARAMS, "rxgpu_hnsw_read_lds_reruns: null argument");
	if (h->shard_set) {
		*reruns = 0;
		for (uint32_t s = 0; s < rxgpu_index_shard_count(h); ++s) {
			uint64_t v = 0;
			if (int rc = rxgpu_hnsw_read_lds_reruns(rxgpu_index_shard(h, s), &v); rc) return rc;
			*reruns += v;
		}
		return RXGPU_OK;
	}
	*reruns = h->hnsw_lds_reruns.exchange(0);
	return RXGPU_OK;
}

int rxgpu_hnsw_read_tie_reruns(rxgpu_index* h, uint64_t* reruns) {
	RX_CHECK(h && reruns, RXGPU_ERR_PARAMS, "rxgpu_hnsw_read_tie_reruns: null argument");
	if (h->shard_set) {
		*reruns = 0;
		for (uint32_t s = 0; s < rxgpu_index_shard_count(h); ++s) {
			uint64_t v = 0;
			if (int rc = rxgpu_hnsw_read_tie_reruns(rxgpu_index_shard(h, s), &v); rc) return rc;
			*reruns += v;
		}
		return RXGPU_OK;
	}
	*reruns = h->hnsw_tie_reruns.exchange(0);   // queries this library re-ran in a launch of their own ...
	if (h->d_hnsw_stats) {                      // ... and searches that started over on the heaps inside the sorted-list kernel
		DeviceGuard dg(h->device);
		unsigned long long v = 0;
		RX_HIP(rxgpu::device_wait_all(h->device));
		RX_HIP(hipMemcpy(&v, h->d_hnsw_stats + 2, sizeof(v), hipMemcpyDeviceToHost));
		RX_HIP(hipMemset(h->d_hnsw_stats + 2, 0, sizeof(v)));
		*reruns += v;
	}
	return RXGPU_OK;
}

int rxgpu_profile_enable(rxgpu_index* h, int on) {
	RX_CHECK(h, RXGPU_ERR_PARAMS, "null index");
	if (h->shard_set) {
		set_error("rxgpu_profile_enable: not available on a sharded index");
		return RXGPU_ERR_LOGIC;
	}
	if (h->shard_set) {
		set_error("rxgpu_hnsw_stream_begin: not available on a sharded index");
		return RXGPU_ERR_LOGIC;
	}
	std::lock_guard<std::mutex> lk(h->mtx);
	for (auto& kv : h->profile) {
		for (auto& ev : kv.second.events) {
			(void)hipEventDestroy(ev.first);
			(void)hipEventDestroy(ev.second);
		}
	}
	h->profile.clear();
	h->profiling = on != 0;
	return RXGPU_OK;
}

int rxgpu_profile_read(rxgpu_index* h, const char* name, uint64_t* launches, double* total_ms) {
	RX_CHECK(h && name && launches && total_ms, RXGPU_ERR_PARAMS, "rxgpu_profile_read: null argument");
	*launches = 0;
	*total_ms = 0.0;
	std::lock_guard<std::mutex> lk(h->mtx);
	auto it = h->profile.find(name);
	if (it == h->profile.end()) return RXGPU_OK;
	DeviceGuard dg(h->device);
	for (auto& ev : it->second.events) {
		RX_HIP(hipEventSynchronize(ev.second));
		float ms = 0.f;
		RX_HIP(hipEventElapsedTime(&ms, ev.first, ev.second));
		*total_ms += ms;
		++*launches;
	}
	return RXGPU_OK;
}

}  // extern "C"
