// rectdetect-mi355x: internal definitions of the opaque OpenCL-style handles served by the HIP runtime layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define CL_TARGET_OPENCL_VERSION 120
#define CL_USE_DEPRECATED_OPENCL_1_2_APIS
#include <CL/cl.h>

extern "C" {
#include "helper.h"
#include "oclhelper.h"
}

#define RD_MAGIC_MEM 0x52444d31u
#define RD_MAGIC_QUEUE 0x52445131u
#define RD_MAGIC_EVENT 0x52444531u

struct _cl_platform_id { int dummy; };
struct _cl_device_id { int ordinal; char name[256]; };
struct _cl_context { int refs; int ordinal; };
struct _cl_command_queue { uint32_t magic; int refs; int ordinal; hipStream_t stream; };
struct _cl_mem { uint32_t magic; int refs; void *dptr; size_t size; int host_pinned; };
struct _cl_event { uint32_t magic; int refs; hipEvent_t ev; };
struct _cl_program { int dummy; };
struct _cl_kernel { int dummy; };

// fatal-on-error convention of the reference (helper.c:31-38, oclhelper.c:113-138)
#define RD_HIP(call)                                                                                   \
  do {                                                                                                 \
    hipError_t rd_e_ = (call);                                                                         \
    if (rd_e_ != hipSuccess) exitf(-1, "%s:%d: %s failed: %s\n", __FILE__, __LINE__, #call, hipGetErrorString(rd_e_)); \
  } while (0)

namespace rdrt {
inline void *dptr(cl_mem m) {
  if (!m || m->magic != RD_MAGIC_MEM) exitf(-1, "rectdetect: invalid cl_mem handle\n");
  return m->dptr;
}
inline hipStream_t stream(cl_command_queue q) {
  if (!q || q->magic != RD_MAGIC_QUEUE) exitf(-1, "rectdetect: invalid cl_command_queue handle\n");
  return q->stream;
}
// wait for a NULL-terminated event list on the queue's stream (the reference passes events == NULL everywhere)
void wait_list(cl_command_queue q, const cl_event *events);
// NULL when events == NULL (oclhelper.c:740-743), else a new event recorded on the queue
cl_event finish_op(cl_command_queue q, const cl_event *events);
void check_launch(const char *what);
int current_device();
}  // namespace rdrt
