// Runtime half of the C ABI: error string, device memory, streams, events and
// hipGraph capture.  Kernels live in fss_conv.hip and fss_kernels.hip.
#include "fss_common.h"

namespace fss {
static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
}  // namespace fss

using namespace fss;

extern "C" {

const char* fss_last_error(void) { return g_last_error.c_str(); }

int fss_version(void) { return 108; }

int fss_device_count(int* count) {
  FSS_REQUIRE(count != nullptr, "null argument");
  FSS_CHECK_HIP(hipGetDeviceCount(count));
  return 0;
}

int fss_set_device(int device) {
  FSS_CHECK_HIP(hipSetDevice(device));
  return fss::pyramid_scratch_init();
}

int fss_device_name(char* buf, int buflen) {
  FSS_REQUIRE(buf != nullptr && buflen > 0, "bad buffer");
  int dev = 0;
  FSS_CHECK_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  FSS_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
  snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName,
           prop.multiProcessorCount);
  return 0;
}

int fss_malloc(void** ptr, size_t bytes) {
  FSS_REQUIRE(ptr != nullptr, "null argument");
  FSS_CHECK_HIP(hipMalloc(ptr, bytes));
  return 0;
}

int fss_free(void* ptr) {
  FSS_CHECK_HIP(hipFree(ptr));
  return 0;
}

int fss_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream) {
  FSS_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  return 0;
}

int fss_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream) {
  FSS_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return 0;
}

int fss_memset(void* dst, int value, size_t bytes, void* stream) {
  FSS_CHECK_HIP(hipMemsetAsync(dst, value, bytes, (hipStream_t)stream));
  return 0;
}

int fss_stream_create(void** stream) {
  FSS_REQUIRE(stream != nullptr, "null argument");
  hipStream_t s;
  FSS_CHECK_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = (void*)s;
  return 0;
}

int fss_stream_destroy(void* stream) {
  FSS_CHECK_HIP(hipStreamDestroy((hipStream_t)stream));
  return 0;
}

int fss_stream_sync(void* stream) {
  FSS_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

int fss_device_sync(void) {
  FSS_CHECK_HIP(hipDeviceSynchronize());
  return 0;
}

int fss_event_create(void** event) {
  FSS_REQUIRE(event != nullptr, "null argument");
  hipEvent_t e;
  FSS_CHECK_HIP(hipEventCreate(&e));
  *event = (void*)e;
  return 0;
}

int fss_event_destroy(void* event) {
  FSS_CHECK_HIP(hipEventDestroy((hipEvent_t)event));
  return 0;
}

int fss_event_record(void* event, void* stream) {
  FSS_CHECK_HIP(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
  return 0;
}

int fss_stream_wait_event(void* stream, void* event) {
  FSS_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
  return 0;
}

int fss_event_sync(void* event) {
  FSS_CHECK_HIP(hipEventSynchronize((hipEvent_t)event));
  return 0;
}

int fss_event_elapsed_ms(void* start, void* stop, float* ms) {
  FSS_REQUIRE(ms != nullptr, "null argument");
  FSS_CHECK_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return 0;
}

int fss_capture_begin(void* stream) {
  FSS_REQUIRE(stream != nullptr, "capture needs a non-null stream");
  FSS_CHECK_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
  return 0;
}

int fss_capture_end(void* stream, void** graph_exec) {
  FSS_REQUIRE(graph_exec != nullptr, "null argument");
  hipGraph_t graph = nullptr;
  FSS_CHECK_HIP(hipStreamEndCapture((hipStream_t)stream, &graph));
  hipGraphExec_t exec = nullptr;
  hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  hipGraphDestroy(graph);
  if (e != hipSuccess) {
    set_error(std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
    return 1;
  }
  *graph_exec = (void*)exec;
  return 0;
}

int fss_graph_launch(void* graph_exec, void* stream) {
  FSS_CHECK_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
  return 0;
}

int fss_graph_destroy(void* graph_exec) {
  FSS_CHECK_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
  return 0;
}

}  // extern "C"
