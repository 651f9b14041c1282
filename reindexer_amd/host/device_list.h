// The device list of an index, as the reference's process hands it over: an environment variable read where the reference constructs its
// Map — `Map(VectorMetric, size_t dim, size_t maxElements)` (cpp_src/core/index/float_vector/hnsw_index.cc:61-66) has no room for
// a device argument, and IndexOpts has no such field (SURVEY §8b "How a user selects the GPU engine", §8e "Host topology").
//
//     RX_GPU_VECTOR_INDEXES=3            one device
//     RX_GPU_VECTOR_INDEXES=0,1,2,3      a list — the Map range-shards its rows over it (BASELINE configs[3])
//     RX_GPU_VECTOR_INDEXES=0-7          a range; "0-3,6,7" mixes both; a device may be named more than once (several shards on one GPU)
//     unset / empty / malformed          no GPU engine (the factories keep the CPU Maps)
//
// No reference headers here: rx_seam.h (in-tree) and host_capi.cc (the test shim) share this parser.
#pragma once

#include <cstdlib>
#include <vector>

namespace rxgpu::host {

constexpr int kMaxListedDevices = 64;   // shards per index (rxgpu_index_create_sharded takes up to 64 slots)

inline std::vector<int> ParseDeviceList(const char* s) {
	std::vector<int> out;
	if (!s) return out;
	auto skip = [&] { while (*s == ' ' || *s == '\t') ++s; };
	auto number = [&](long& v) {
		skip();
		if (*s < '0' || *s > '9') return false;
		char* end = nullptr;
		v = std::strtol(s, &end, 10);
		s = end;
		skip();
		return v >= 0 && v < 1024;
	};
	skip();
	if (!*s) return out;
	for (;;) {
		long a = 0, b = 0;
		if (!number(a)) return {};
		b = a;
		if (*s == '-') {
			++s;
			if (!number(b) || b < a) return {};
		}
		for (long d = a; d <= b; ++d) {
			if (int(out.size()) == kMaxListedDevices) return {};
			out.push_back(int(d));
		}
		if (*s == ',') {
			++s;
			continue;
		}
		if (*s) return {};
		return out;
	}
}

// the devices an index definition is routed to; empty: the CPU engines
inline std::vector<int> GpuDevicesFromEnv(const char* var = "RX_GPU_VECTOR_INDEXES") { return ParseDeviceList(std::getenv(var)); }

}  // namespace rxgpu::host
