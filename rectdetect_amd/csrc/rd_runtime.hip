// rectdetect-mi355x: thin C-ABI HIP runtime layer that replaces the reference's OpenCL dispatch (oclhelper.c) and the
// raw OpenCL entry points its demo programs call themselves:
//   clCreateCommandQueue clReleaseCommandQueue clReleaseContext clCreateBuffer clEnqueueReadBuffer
//   clEnqueueWriteBuffer clFinish clFlush clReleaseMemObject clReleaseEvent clRetainEvent clGetEventInfo
// Handles are opaque pointers with the OpenCL type names; nothing here talks to an OpenCL ICD.
#include "rd_internal.h"
#include "rectdetect_hip.h"
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <pthread.h>

// State of this layer.  The reference keeps its plan, kernel ids and pinned-memory map in process-global variables and is not
// re-entrant (oclhelper.c:329-334, 821, 835); here one process may drive several GPUs from several threads, so the
// selected device is per thread and the shared tables are guarded.
static thread_local int g_selected = 0;
static struct _cl_device_id g_devices[16];
static int g_ndev = -1;
static pthread_once_t g_enum_once = PTHREAD_ONCE_INIT;
static pthread_mutex_t g_state_mu = PTHREAD_MUTEX_INITIALIZER;     // pinned-memory map, plan table

static void enumerate_once() {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { n = 0; (void)hipGetLastError(); }
  if (n > 16) n = 16;
  for (int i = 0; i < n; i++) {
    hipDeviceProp_t p;
    g_devices[i].ordinal = i;
    if (hipGetDeviceProperties(&p, i) == hipSuccess) snprintf(g_devices[i].name, sizeof(g_devices[i].name), "%s (%s), HIP gfx950 runtime layer", p.name, p.gcnArchName);
    else snprintf(g_devices[i].name, sizeof(g_devices[i].name), "HIP device %d", i);
  }
  g_ndev = n;
}
static void enumerate() { pthread_once(&g_enum_once, enumerate_once); }

namespace rdrt {
void wait_list(cl_command_queue q, const cl_event *events) {
  if (!events) return;
  for (int i = 0; events[i] != NULL; i++) RD_HIP(hipStreamWaitEvent(stream(q), events[i]->ev, 0));
}
cl_event finish_op(cl_command_queue q, const cl_event *events) {
  if (!events) return NULL;
  cl_event e = (cl_event)calloc(1, sizeof(*e));
  e->magic = RD_MAGIC_EVENT; e->refs = 1;
  RD_HIP(hipEventCreateWithFlags(&e->ev, hipEventDisableTiming));
  RD_HIP(hipEventRecord(e->ev, stream(q)));
  return e;
}
void check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) exitf(-1, "rectdetect: kernel launch failed in %s: %s\n", what, hipGetErrorString(e));
}
int current_device() { return g_selected; }
}  // namespace rdrt

extern "C" {

const char *rd_version(void) { return "rectdetect-mi355x 0.1 (gfx950)"; }
int rd_device_count(void) { enumerate(); return g_ndev; }
// PCI bus id ("0000:c1:00.0") of a HIP device: lets a per-GPU host process pin itself to the cores next to its GPU
// (/sys/bus/pci/devices/<id>/local_cpulist).  Returns 0 on success.
int rd_device_pci_bus_id(int ordinal, char *buf, int len) {
  enumerate();
  if (ordinal < 0 || ordinal >= g_ndev || !buf || len < 16) return -1;
  return hipDeviceGetPCIBusId(buf, len, ordinal) == hipSuccess ? 0 : -1;
}
void rd_select_device(int ordinal) { enumerate(); if (ordinal < 0 || ordinal >= g_ndev) exitf(-1, "rd_select_device: no HIP device %d\n", ordinal); g_selected = ordinal; RD_HIP(hipSetDevice(ordinal)); }
void *rd_device_alloc(size_t bytes) { void *p = NULL; RD_HIP(hipMalloc(&p, bytes ? bytes : 1)); return p; }
void rd_device_free(void *dptr) { if (dptr) RD_HIP(hipFree(dptr)); }
// pinned host memory for callers without a HIP binding of their own (frames handed over with rd_detector_enqueue(..., RD_FRAME_HOST_PINNED) travel straight from it)
void *rd_host_alloc(size_t bytes) { void *p = NULL; RD_HIP(hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault)); return p; }
void rd_host_free(void *p) { if (p) RD_HIP(hipHostFree(p)); }
void rd_upload(void *dptr, const void *host, size_t bytes) { RD_HIP(hipMemcpy(dptr, host, bytes, hipMemcpyHostToDevice)); }
void rd_download(void *host, const void *dptr, size_t bytes) { RD_HIP(hipMemcpy(host, dptr, bytes, hipMemcpyDeviceToHost)); }

// ------------------------------------------------------------------ error strings (oclhelper.c:36-111)
const char *clStrError(int c) {
  switch (c) {
  case 0: return "CL_SUCCESS";
  case -1: return "CL_DEVICE_NOT_FOUND";
  case -2: return "CL_DEVICE_NOT_AVAILABLE";
  case -4: return "CL_MEM_OBJECT_ALLOCATION_FAILURE";
  case -5: return "CL_OUT_OF_RESOURCES";
  case -6: return "CL_OUT_OF_HOST_MEMORY";
  case -30: return "CL_INVALID_VALUE";
  case -33: return "CL_INVALID_DEVICE";
  case -34: return "CL_INVALID_CONTEXT";
  case -36: return "CL_INVALID_COMMAND_QUEUE";
  case -38: return "CL_INVALID_MEM_OBJECT";
  case -48: return "CL_INVALID_KERNEL";
  case -58: return "CL_INVALID_EVENT";
  case -59: return "CL_INVALID_OPERATION";
  default: return "Unknown error";
  }
}

cl_int checkError(cl_int ret, const char *s) {
  if (ret != CL_SUCCESS) {
    if (s == NULL) exitf(-1, "%s(%d)\n", clStrError(ret), ret);
    exitf(-1, "%s : %s(%d)\n", s, clStrError(ret), ret);
  }
  return CL_SUCCESS;
}

cl_int ce(cl_int ret) {
  if (ret != CL_SUCCESS) exitf(-1, "%s\n", clStrError(ret));
  return CL_SUCCESS;
}

// ------------------------------------------------------------------ devices / contexts (oclhelper.c:143-233)
char *getDeviceName(cl_device_id device) {
  char *s = (char *)malloc(300);
  snprintf(s, 300, "%s", device ? device->name : "(null)");
  String_trim(s);
  return s;
}

int simpleGetDevices(cl_device_id *devices, int maxDevices) {
  enumerate();
  int n = g_ndev < maxDevices ? g_ndev : maxDevices;
  for (int i = 0; i < n; i++) devices[i] = &g_devices[i];
  return n;
}

cl_device_id simpleGetDevice(int did) {
  enumerate();
  if (g_ndev == 0) exitf(-1, "No platform available\n");
  if (did < 0 || did >= g_ndev) {
    if (did >= 0) fprintf(stderr, "Device %d does not exist\n", did);
    for (int i = 0; i < g_ndev; i++) fprintf(stderr, "Device %d : %s\n", i, g_devices[i].name);
    exit(-1);
  }
  return &g_devices[did];
}

cl_context simpleCreateContext(cl_device_id device) {
  if (!device) exitf(-1, "Could not create context : %s\n", clStrError(CL_INVALID_DEVICE));
  cl_context c = (cl_context)calloc(1, sizeof(*c));
  c->refs = 1; c->ordinal = device->ordinal;
  RD_HIP(hipSetDevice(device->ordinal));
  g_selected = device->ordinal;
  return c;
}

cl_int clReleaseContext(cl_context c) { if (c && --c->refs == 0) free(c); return CL_SUCCESS; }

cl_command_queue clCreateCommandQueue(cl_context c, cl_device_id d, cl_command_queue_properties props, cl_int *err) {
  (void)props;
  cl_command_queue q = (cl_command_queue)calloc(1, sizeof(*q));
  q->magic = RD_MAGIC_QUEUE; q->refs = 1; q->ordinal = d ? d->ordinal : (c ? c->ordinal : 0);
  RD_HIP(hipSetDevice(q->ordinal));
  RD_HIP(hipStreamCreateWithFlags(&q->stream, hipStreamNonBlocking));
  if (err) *err = CL_SUCCESS;
  return q;
}

cl_int clReleaseCommandQueue(cl_command_queue q) {
  if (!q || q->magic != RD_MAGIC_QUEUE) return CL_INVALID_COMMAND_QUEUE;
  if (--q->refs == 0) { RD_HIP(hipStreamSynchronize(q->stream)); RD_HIP(hipStreamDestroy(q->stream)); q->magic = 0; free(q); }
  return CL_SUCCESS;
}

cl_int clFinish(cl_command_queue q) { RD_HIP(hipStreamSynchronize(rdrt::stream(q))); return CL_SUCCESS; }
cl_int clFlush(cl_command_queue q) { (void)q; return CL_SUCCESS; }

// ------------------------------------------------------------------ buffers
cl_mem clCreateBuffer(cl_context c, cl_mem_flags flags, size_t size, void *host, cl_int *err) {
  (void)c;
  cl_mem m = (cl_mem)calloc(1, sizeof(*m));
  m->magic = RD_MAGIC_MEM; m->refs = 1; m->size = size;
  if (flags & CL_MEM_ALLOC_HOST_PTR) {
    RD_HIP(hipHostMalloc(&m->dptr, size ? size : 1, hipHostMallocDefault));
    m->host_pinned = 1;
    memset(m->dptr, 0, size);
  } else {
    RD_HIP(hipMalloc(&m->dptr, size ? size : 1));
    if ((flags & (CL_MEM_COPY_HOST_PTR | CL_MEM_USE_HOST_PTR)) && host) RD_HIP(hipMemcpy(m->dptr, host, size, hipMemcpyHostToDevice));
    else {   // fresh buffers read as zero (the detector relies on it).  hipMemset is asynchronous for device memory and
      // the queues are non-blocking streams: without the wait the fill could land after a kernel's first writes
      RD_HIP(hipMemset(m->dptr, 0, size));
      RD_HIP(hipStreamSynchronize(0));
    }
  }
  if (err) *err = CL_SUCCESS;
  return m;
}

cl_int clReleaseMemObject(cl_mem m) {
  if (!m || m->magic != RD_MAGIC_MEM) return CL_INVALID_MEM_OBJECT;
  if (--m->refs == 0) {
    if (m->host_pinned) RD_HIP(hipHostFree(m->dptr)); else RD_HIP(hipFree(m->dptr));
    m->magic = 0; free(m);
  }
  return CL_SUCCESS;
}

static cl_event make_event(cl_command_queue q) {
  cl_event e = (cl_event)calloc(1, sizeof(*e));
  e->magic = RD_MAGIC_EVENT; e->refs = 1;
  RD_HIP(hipEventCreateWithFlags(&e->ev, hipEventDisableTiming));
  RD_HIP(hipEventRecord(e->ev, rdrt::stream(q)));
  return e;
}

cl_int clEnqueueReadBuffer(cl_command_queue q, cl_mem m, cl_bool blocking, size_t off, size_t size, void *dst, cl_uint nev, const cl_event *evl, cl_event *ev) {
  if (!m || m->magic != RD_MAGIC_MEM || off + size > m->size) return CL_INVALID_VALUE;
  for (cl_uint i = 0; i < nev; i++) RD_HIP(hipStreamWaitEvent(rdrt::stream(q), evl[i]->ev, 0));
  RD_HIP(hipMemcpyAsync(dst, (char *)m->dptr + off, size, hipMemcpyDeviceToHost, rdrt::stream(q)));
  if (ev) *ev = make_event(q);
  if (blocking) RD_HIP(hipStreamSynchronize(rdrt::stream(q)));
  return CL_SUCCESS;
}

cl_int clEnqueueWriteBuffer(cl_command_queue q, cl_mem m, cl_bool blocking, size_t off, size_t size, const void *src, cl_uint nev, const cl_event *evl, cl_event *ev) {
  if (!m || m->magic != RD_MAGIC_MEM || off + size > m->size) return CL_INVALID_VALUE;
  for (cl_uint i = 0; i < nev; i++) RD_HIP(hipStreamWaitEvent(rdrt::stream(q), evl[i]->ev, 0));
  RD_HIP(hipMemcpyAsync((char *)m->dptr + off, src, size, hipMemcpyHostToDevice, rdrt::stream(q)));
  if (ev) *ev = make_event(q);
  if (blocking) RD_HIP(hipStreamSynchronize(rdrt::stream(q)));
  return CL_SUCCESS;
}

// ------------------------------------------------------------------ events
cl_int clRetainEvent(cl_event e) { if (e) e->refs++; return CL_SUCCESS; }
cl_int clReleaseEvent(cl_event e) {
  if (!e) return CL_SUCCESS;
  if (e->magic != RD_MAGIC_EVENT) return CL_INVALID_EVENT;
  if (--e->refs == 0) { RD_HIP(hipEventDestroy(e->ev)); e->magic = 0; free(e); }
  return CL_SUCCESS;
}
cl_int clGetEventInfo(cl_event e, cl_event_info what, size_t sz, void *val, size_t *ret) {
  if (!e || e->magic != RD_MAGIC_EVENT || what != CL_EVENT_COMMAND_EXECUTION_STATUS || sz < sizeof(cl_int)) return CL_INVALID_VALUE;
  hipError_t r = hipEventQuery(e->ev);
  if (r != hipSuccess && r != hipErrorNotReady) exitf(-1, "clGetEventInfo: %s\n", hipGetErrorString(r));
  *(cl_int *)val = r == hipSuccess ? CL_COMPLETE : CL_RUNNING;
  if (ret) *ret = sizeof(cl_int);
  return CL_SUCCESS;
}

// The reference polls every 15 ms (oclhelper.c:799-817); a HIP event can simply be waited for.
void waitForEvent(cl_event ev) {
  if (!ev || ev->magic != RD_MAGIC_EVENT) exitf(-1, "waitForEvent: invalid event\n");
  RD_HIP(hipEventSynchronize(ev->ev));
}

// ------------------------------------------------------------------ pinned memory (oclhelper.c:823-864)
static ArrayMap *g_pinned = NULL;
void *allocatePinnedMemory(size_t z, cl_context context, cl_command_queue queue) {
  (void)context; (void)queue;
  void *p = NULL;
  RD_HIP(hipHostMalloc(&p, z ? z : 1, hipHostMallocDefault));
  pthread_mutex_lock(&g_state_mu);
  if (!g_pinned) g_pinned = initArrayMap();
  ArrayMap_put(g_pinned, (uint64_t)(uintptr_t)p, p);
  pthread_mutex_unlock(&g_state_mu);
  return p;
}
void freePinnedMemory(void *p, cl_context context, cl_command_queue queue) {
  (void)context; (void)queue;
  pthread_mutex_lock(&g_state_mu);
  const bool known = g_pinned && ArrayMap_remove(g_pinned, (uint64_t)(uintptr_t)p) != NULL;
  pthread_mutex_unlock(&g_state_mu);
  if (!known) exitf(-1, "freePinnedMemory: unknown pointer\n");
  RD_HIP(hipHostFree(p));
}

// ------------------------------------------------------------------ kernel ids and the local-size "plan" (oclhelper.c:312-605, 821-823)
// Work-group sizes are compile-time properties of the HIP kernels, so the plan has no effect on launches.  The functions
// keep the reference's file format so that a plan.txt written by either implementation can be read by the other.
static int g_next_kernel_id = 0;
int getNextKernelID() { return __atomic_fetch_add(&g_next_kernel_id, 1, __ATOMIC_RELAXED); }

#define RD_KERNELIDMAX 1000
static struct { int valid; long long ws[3], ns; } g_plan[RD_KERNELIDMAX];

void clearPlan() { for (int i = 0; i < RD_KERNELIDMAX; i++) g_plan[i].valid = 0; }

static char *device_tag(cl_device_id device) {
  char *dn = getDeviceName(device);
  for (char *p = dn; *p; p++) { if (*p == ':') *p = ';'; if (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r') *p = '_'; }
  return dn;
}

int loadPlan(const char *fn, cl_device_id device) {
  char *tag = device_tag(device);
  size_t tl = strlen(tag);
  clearPlan();
  FILE *fp = fopen(fn, "r");
  if (!fp) { free(tag); return -1; }
  char line[1024];
  int found = 0;
  while (fgets(line, sizeof(line), fp)) {
    if (strncmp(line, tag, tl) != 0 || strncmp(line + tl, " : ", 3) != 0) continue;
    int kid; long long a, b, c, t;
    if (sscanf(line + tl, " : %d : %lld : %lld : %lld : %lld", &kid, &a, &b, &c, &t) == 5 && kid >= 0 && kid < RD_KERNELIDMAX) {
      g_plan[kid].valid = 1; g_plan[kid].ws[0] = a; g_plan[kid].ws[1] = b; g_plan[kid].ws[2] = c; g_plan[kid].ns = t;
      found = 1;
    }
  }
  fclose(fp);
  free(tag);
  return found ? 0 : -1;
}

void savePlan(const char *fn, cl_device_id device) {
  char *tag = device_tag(device);
  size_t tl = strlen(tag);
  // keep the lines of other devices
  char *keep = NULL; size_t klen = 0;
  FILE *fp = fopen(fn, "r");
  if (fp) {
    char line[1024];
    while (fgets(line, sizeof(line), fp)) {
      if (strncmp(line, tag, tl) == 0 && strncmp(line + tl, " : ", 3) == 0) continue;
      size_t l = strlen(line);
      keep = (char *)realloc(keep, klen + l + 1);
      memcpy(keep + klen, line, l + 1); klen += l;
    }
    fclose(fp);
  }
  fp = fopen(fn, "w");
  if (!fp) exitf(-1, "Couldn't open file %s for writing\n", fn);
  if (keep) fwrite(keep, 1, klen, fp);
  for (int i = 0; i < RD_KERNELIDMAX; i++)
    if (g_plan[i].valid) fprintf(fp, "%s : %d : %lld : %lld : %lld : %lld\n", tag, i, g_plan[i].ws[0], g_plan[i].ws[1], g_plan[i].ws[2], g_plan[i].ns);
  fclose(fp);
  free(keep); free(tag);
}

static long long g_prof_ws[3];
static int g_profiling = 0;
void startProfiling(size_t ws1, size_t ws2, size_t ws3) { g_prof_ws[0] = (long long)ws1; g_prof_ws[1] = (long long)ws2; g_prof_ws[2] = (long long)ws3; g_profiling = 1; }
void finishProfiling() {
  // one entry so that savePlan() produces a non-empty plan for this device and the 48-frame sweep of rect.cpp:86-101 runs once
  if (g_profiling && !g_plan[0].valid) { g_plan[0].valid = 1; g_plan[0].ws[0] = 64; g_plan[0].ws[1] = 4; g_plan[0].ws[2] = 1; g_plan[0].ns = 0; }
  g_profiling = 0;
}
void showPlan() {
  printf("%2s : %40s : %4s : %4s : %4s : %9s\n", "ID", "Kernel function name", "WS0", "WS1", "WS2", "Nano sec");
  for (int i = 0; i < RD_KERNELIDMAX; i++)
    if (g_plan[i].valid) printf("%2d : %40s : %4lld : %4lld : %4lld : %9lld\n", i, "(ahead-of-time HIP kernel)", g_plan[i].ws[0], g_plan[i].ws[1], g_plan[i].ws[2], g_plan[i].ns);
  fflush(stdout);
}

// ------------------------------------------------------------------ source-built kernels do not exist in this implementation
int simpleBuildProgram(cl_program program, cl_device_id device, const char *optionString) {
  (void)program; (void)device; (void)optionString;
  exitf(-1, "simpleBuildProgram: kernels are compiled ahead of time for gfx950; there is no OpenCL program to build\n");
  return -1;
}
void simpleSetKernelArg(cl_kernel kernel, const char *format, ...) {
  (void)kernel; (void)format;
  exitf(-1, "simpleSetKernelArg: not available - use the oclimgutil_* / oclpolyline_* / oclrect_* entry points\n");
}
cl_event runKernel1D(cl_command_queue q, cl_kernel k, int id, size_t ws1, int nev, ...) { (void)q; (void)k; (void)id; (void)ws1; (void)nev; exitf(-1, "runKernel1D: not available (ahead-of-time HIP kernels)\n"); return NULL; }
cl_event runKernel2D(cl_command_queue q, cl_kernel k, int id, size_t ws1, size_t ws2, int nev, ...) { (void)q; (void)k; (void)id; (void)ws1; (void)ws2; (void)nev; exitf(-1, "runKernel2D: not available (ahead-of-time HIP kernels)\n"); return NULL; }
cl_event runKernel1Dx(cl_command_queue q, cl_kernel k, int id, size_t ws1, const cl_event *e) { (void)q; (void)k; (void)id; (void)ws1; (void)e; exitf(-1, "runKernel1Dx: not available (ahead-of-time HIP kernels)\n"); return NULL; }
cl_event runKernel2Dx(cl_command_queue q, cl_kernel k, int id, size_t ws1, size_t ws2, const cl_event *e) { (void)q; (void)k; (void)id; (void)ws1; (void)ws2; (void)e; exitf(-1, "runKernel2Dx: not available (ahead-of-time HIP kernels)\n"); return NULL; }

}  // extern "C"
