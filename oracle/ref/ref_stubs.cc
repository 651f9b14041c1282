// TEST INFRASTRUCTURE ONLY — never linked into the product library.
//
// Link-time stubs that let the reference's float_vector engines
// (/root/reference/cpp_src/core/index/float_vector/hnswlib/*, tools/distances/*,
// tools/normalize.cc, tools/cpucheck.cc) build standalone, without the rest of
// libreindexer (logger, assertion reporter, backtrace, quantization JSON).
// Declarations being satisfied:
//   tools/logger.h:14-16, tools/assertrx.h:5,18, core/definitions/quantization_config.h,
//   debug/backtrace.h:20-22.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <ostream>
#include <stdexcept>
#include <string>

#include "core/index/float_vector/hnswlib/hnsw_interface.h"
#include "core/definitions/quantization_config.h"

namespace reindexer {
namespace logger_details {
std::atomic<int> g_LogLevel{0};
void logPrintImpl(int, char*) {}
}  // namespace logger_details

[[noreturn]] void fail_throwrx(const char* assertion, const char* file, unsigned line, const char* function) noexcept(false) {
	throw std::logic_error(std::string("assertrx_throw failed: ") + assertion + " at " + file + ":" + std::to_string(line) + " " + function);
}
[[noreturn]] void fail_assertrx(const char* assertion, const char* file, unsigned line, const char* function) noexcept {
	std::fprintf(stderr, "assertrx failed: %s at %s:%u %s\n", assertion, file, line, function);
	std::abort();
}
namespace debug {
void backtrace_set_assertion_message(std::string&&) noexcept {}
void print_backtrace(std::ostream&, void*, int) {}
void print_crash_query(std::ostream&) {}
}  // namespace debug
}  // namespace reindexer

namespace hnswlib {
void QuantizationConfig::Deserialize(IReader&) {}
void QuantizationConfig::Serialize(IWriter&) const {}
}  // namespace hnswlib
